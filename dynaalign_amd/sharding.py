"""Row-sharding of the N x N pair space over the GPUs of one node.

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI).  The inputs
are tiny (2 MB of residues at N = 100k), so every rank holds all sequences and rebuilds all
signatures itself; only the result is exchanged:

    rank p computes the upper-triangular tiles of tile rows p, p+P, p+2P, ...   (HIP kernel)
    ONE all-gather of the equally sized compact uint16 blocks                    (RCCL)
    every rank expands the gathered blocks to the dense float64 matrix          (HIP kernel)

Tile rows are dealt cyclically so that the triangular work is balanced to within one tile
row.  The compact payload (2 B/pair instead of 8) and the folded block layout (only the
upper triangle travels) are what make the gather affordable: at N = 100k, P = 8 each GPU
receives 8.8 GB instead of 70 GB.

Seeds: with no explicit seed the reference draws one from std::random_device per call; here rank 0 draws it
and broadcasts it (`shared_seed`) -- ranks hashing with different families would assemble a meaningless matrix.

The orchestration (`Plan`, `all_pairs_sharded`) is independent of where the blocks come from,
so the world_size-2 gloo tests drive it on CPU with blocks produced by the test oracle; the
product path (`mh_sharded_step`, `nw_sharded_step`) feeds it from the HIP kernels only.
"""
import torch
import torch.distributed as dist

from . import _capi

MH_TILE, NW_TILE = 128, 128    # rows per cyclic unit; the NW kernel's own tile is 64 x 64, two tile rows per unit


class Plan:
    """Which rows a rank owns and where element (i, j >= i) sits in its block / the gathered buffer.

    Folded layout (csrc/da_common.hpp ShardGeom): rank p owns tile rows t = q*world + p, q = 0..Q-1, and only
    their part right of the diagonal is valid, so local tile rows q and Q-1-q share one stored tile row of
    width W = ceil8(n) + world*tile: the first left-aligned from its diagonal tile, the second starting at column
    world*tile (both on a multiple of 8 columns).  A block is local_rows x width -- half of a full-width row block."""

    def __init__(self, n, rank, world, tile=MH_TILE):
        self.n, self.rank, self.world, self.tile = int(n), int(rank), int(world), int(tile)
        self.tiles = -(-self.n // self.tile)                       # tile rows of the pair space (T)
        self.local_tiles = -(-self.tiles // self.world)            # tile rows per rank (Q, padded)
        self.stored_tiles = (self.local_tiles + 1) // 2            # after folding (Qh)
        self.local_rows = self.stored_tiles * self.tile
        self.width = -(-self.n // 8) * 8 + self.world * self.tile
        self.back = self.world * self.tile                         # local column of global column 0 in a back-aligned row (width - ceil8(n))

    def owner(self, i):
        """rank owning global row i"""
        return (i // self.tile) % self.world

    def locate(self, i, j):
        """(rank, local row, local column) of element (i, j), valid for j >= tile start of i"""
        t = i // self.tile
        q = t // self.world
        front = q <= self.local_tiles - 1 - q
        f = q if front else self.local_tiles - 1 - q
        row = f * self.tile + i % self.tile
        col = j - t * self.tile if front else self.back + j
        return t % self.world, row, col

    def gathered_row(self, i):
        p, r, _ = self.locate(i, i)
        return p * self.local_rows + r

    def my_rows(self):
        """global rows owned by this rank (rows >= n are padding and skipped)"""
        out = []
        for q in range(self.local_tiles):
            t = q * self.world + self.rank
            out.extend(i for i in range(t * self.tile, min((t + 1) * self.tile, self.n)))
        return out

    def upper_pairs(self):
        """number of (i, j >= tile start) pairs this rank computes -- for load-balance checks"""
        tot = 0
        for q in range(self.local_tiles):
            t = q * self.world + self.rank
            if t < self.tiles:
                rows = min((t + 1) * self.tile, self.n) - t * self.tile
                tot += rows * (self.n - t * self.tile)
        return tot


def shared_seed(seed=None, group=None):
    """The hash seed every rank of the job must use.  All ranks rebuild all signatures themselves, so they have
    to draw the SAME hash family: the reference's default -- no seed, std::random_device (src/minHash.cpp:73) --
    resolved independently per process would gather blocks computed under different hash functions.  Rank 0
    resolves the seed (argument > set_option("seed") > DYNAALIGN_SEED > random_device) and broadcasts it."""
    from .similarity import _resolve_seed
    s = _resolve_seed(seed)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.tensor([s], dtype=torch.int64, device=dev)
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        s = int(t.item())
    return s & 0xFFFFFFFF


def shared_hash_family(n_hash, seed=None, group=None):
    """n_hash hash seeds (reference HashFamily, src/minHash.cpp:73-81), identical on every rank."""
    from .similarity import hash_family_seeds
    return hash_family_seeds(shared_seed(seed, group), n_hash)


def gather_blocks(gathered, local, group=None):
    """THE collective of the sharded path: every rank's block (any dtype, contiguous, the same byte size on every rank) into `gathered`
    (world x that size) -- one all_gather_into_tensor.  NCCL / RCCL and gloo have no 16-bit integer type and an all-gather only moves
    bytes, so both tensors are viewed as the widest integer type their sizes allow: a 2.8 GB block is 3.5*10^8 int64 elements instead of
    2.8*10^9 bytes -- counts stay far below 2^31 in every layer (torch, RCCL) whatever they use for them."""
    nb_l, nb_g = local.numel() * local.element_size(), gathered.numel() * gathered.element_size()
    assert local.is_contiguous() and gathered.is_contiguous() and nb_g % nb_l == 0, "blocks must be contiguous and of equal size"
    lb, gb = local.reshape(-1).view(torch.uint8), gathered.reshape(-1).view(torch.uint8)
    for dt, sz in ((torch.int64, 8), (torch.int32, 4)):
        if nb_l % sz == 0 and lb.data_ptr() % sz == 0 and gb.data_ptr() % sz == 0:
            lb, gb = lb.view(dt), gb.view(dt)
            break
    dist.all_gather_into_tensor(gb, lb, group=group)


def all_pairs_sharded(plan, local_block, gathered, finalize, group=None):
    """local_block: this rank's compact block (tensor [plan.local_rows, ld]) already computed;
    gathered: tensor [world * plan.local_rows, ld]; finalize(gathered) -> result.
    Exactly one collective: all_gather_into_tensor."""
    if plan.world > 1:
        gather_blocks(gathered, local_block, group)
    else:
        gathered.copy_(local_block)
    return finalize(gathered)


class Workspace:
    """Per-rank HBM buffers for the sharded path (allocated once, reused every step)."""

    def __init__(self, plan, device="cuda"):
        self.local = torch.zeros((plan.local_rows, plan.width), dtype=torch.int16, device=device)
        self.gathered = torch.empty((plan.world * plan.local_rows, plan.width), dtype=torch.int16, device=device)


MHWorkspace = Workspace


def _stream():
    return torch.cuda.current_stream().cuda_stream


def finalize_shards(plan, gathered, is_nw, n_hash, out):
    _capi.check(_capi.load().da_dev_finalize_shards(gathered.data_ptr(), gathered.stride(0), plan.n, plan.world,
                                                    1 if is_nw else 0, int(n_hash), out.data_ptr(), out.stride(0),
                                                    _stream()))
    return out


def mh_local_block(plan, work, planes, n_hash):
    assert plan.tile == MH_TILE
    _capi.check(_capi.load().da_dev_mh_compare_shard(planes.data_ptr(), planes.bits, plan.n, int(n_hash),
                                                     plan.rank, plan.world, work.local.data_ptr(),
                                                     work.local.stride(0), _stream()))
    return work.local


def mh_sharded_step(plan, work, planes, n_hash, out, group=None):
    """compare (this rank's tiles) -> one all-gather -> dense float64 n x n in `out` on every rank"""
    mh_local_block(plan, work, planes, n_hash)
    return all_pairs_sharded(plan, work.local, work.gathered,
                             lambda g: finalize_shards(plan, g, False, n_hash, out), group)


def value_bits(n_hash):
    """bits a match count needs (at least the byte plane)"""
    return max(8, int(n_hash).bit_length())


class PackedWorkspace(Workspace):
    """Workspace whose exchange buffers hold the MH counts in value_bits(n_hash) bits instead of 16:
    a byte plane + bit planes per rank block (da_dev_pack_shard)."""

    def __init__(self, plan, n_hash, device="cuda"):
        self.local = torch.zeros((plan.local_rows, plan.width), dtype=torch.int16, device=device)
        self.bits = value_bits(n_hash)
        self.block_bytes = int(_capi.load().da_shard_packed_bytes(plan.n, plan.world, self.bits))
        self.packed = torch.empty(self.block_bytes, dtype=torch.uint8, device=device)
        self.gathered = torch.empty(plan.world * self.block_bytes, dtype=torch.uint8, device=device)


def pack_local_block(plan, work):
    _capi.check(_capi.load().da_dev_pack_shard(work.local.data_ptr(), work.local.stride(0), plan.n, plan.world, work.bits,
                                               work.packed.data_ptr(), _stream()))
    return work.packed


def finalize_shards_packed(plan, work, gathered, n_hash, out):
    _capi.check(_capi.load().da_dev_finalize_shards_packed(gathered.data_ptr(), plan.n, plan.world, work.bits, int(n_hash),
                                                           out.data_ptr(), out.stride(0), _stream()))
    return out


def mh_sharded_step_packed(plan, work, planes, n_hash, out, group=None):
    """mh_sharded_step with the exchange in value_bits(n_hash) bits per count (work: PackedWorkspace)"""
    mh_local_block(plan, work, planes, n_hash)
    pack_local_block(plan, work)
    return all_pairs_sharded(plan, work.packed, work.gathered,
                             lambda g: finalize_shards_packed(plan, work, g, n_hash, out), group)


def nw_local_block(plan, work, ds, matrix_name="BLOSUM62", gap_open=10, gap_ext=4):
    assert plan.tile == NW_TILE
    lib = _capi.load()
    mid = lib.da_matrix_id(matrix_name.encode("latin-1"))
    if mid < 0:
        _capi.check(_capi.DA_ERR_BAD_MATRIX)
    _capi.check(lib.da_dev_nw_shard(ds.codes.data_ptr(), ds.offsets.data_ptr(), plan.n, ds.max_len, mid, int(gap_open),
                                    int(gap_ext), plan.rank, plan.world, work.local.data_ptr(), work.local.stride(0),
                                    _stream()))
    return work.local


def nw_sharded_step(plan, work, ds, out, matrix_name="BLOSUM62", gap_open=10, gap_ext=4, group=None):
    nw_local_block(plan, work, ds, matrix_name, gap_open, gap_ext)
    return all_pairs_sharded(plan, work.local, work.gathered,
                             lambda g: finalize_shards(plan, g, True, 0, out), group)


# ---- the duplicate-collapsing route on the sharded path --------------------------------------------------------------
# Byte-identical sequences have identical rows and columns of the result (csrc/api.cpp mh_full_symmetric / nw_full_symmetric).
# Every rank builds the SAME plan from the input it holds anyway (deterministic: no exchange), computes its shard of the table of
# the U unique strings, the ONE all-gather moves (U/n)^2 of the bytes, and every rank expands the table to the dense n x n
# matrix with the two streaming passes.  MinHash: the symmetric count table, sharded like the n x n problem (cyclic tile rows,
# folded upper triangle, 9-bit packed).  NW: the ORDERED square (calc is not symmetric) as row blocks of cyclic 128-row units.

class UniqueSequences:
    """the plan's unique strings as a DeviceSequences look-alike (tensor views into the plan's workspace)"""

    def __init__(self, uplan, total, max_len):
        w = uplan.work
        base = w.data_ptr()
        bo, oo = uplan.c.d_ubytes - base, uplan.c.d_uoffsets - base
        self.n = uplan.unique
        self.total, self.max_len = int(total), int(max_len)
        self.residues = w[bo:bo + max(self.total, 1)]
        self.offsets = w[oo:oo + 8 * (self.n + 1)].view(torch.int64)
        self.codes = self.residues                                    # (NW: the plan was built on the encoded residues)


def unique_table_row(r, world, unique):
    """host mirror of csrc table_row(): where row r of the unique table sits in the all-gathered row blocks of cyclic 128-row
    units (rank p computed units p, p + world, ...; every block holds ceil(ceil(unique / 128) / world) * 128 rows)"""
    if world <= 1:
        return r
    rows_local = -(-(-(-unique // NW_TILE)) // world) * NW_TILE
    t = r // NW_TILE
    return (t % world) * rows_local + (t // world) * NW_TILE + r % NW_TILE


def dedup_worth(n, unique, is_nw, n_hash=0, max_len=0, min_n=2048):
    """the rule of the single-GPU routes: few enough unique strings (NW: <= 85 %, MinHash: <= 60 %) and a shape the two expansion
    passes cover"""
    from . import device
    import os
    min_n = int(os.environ.get("DYNAALIGN_NW_DEDUP_MIN_N" if is_nw else "DYNAALIGN_MH_DEDUP_MIN_N", min_n))
    if os.environ.get("DYNAALIGN_NW_NO_DEDUP" if is_nw else "DYNAALIGN_MH_NO_DEDUP"):
        return False
    if n < min_n or unique * 100 > n * (85 if is_nw else 60):    # NW: the DP shrinks by ~(U/n)^2 of 466 ms; MinHash: pays below U = 0.63 n
        return False
    if is_nw and not (1 <= max_len <= 64):
        return False
    return device.expand_workspace_bytes(n, unique, is_nw, n_hash, max_len) > 256


def mh_unique_local(uplan, ds, k, n_hash, d_seeds, rank, world):
    """K1 + K1b on the unique strings, this rank's packed shard of their count table -> (plan, work)"""
    from . import device
    useq = UniqueSequences(uplan, ds.total, ds.max_len)
    _, planes = device.minhash_signatures(useq, k, n_hash, d_seeds)
    plan = Plan(uplan.unique, rank, world, MH_TILE)
    work = PackedWorkspace(plan, n_hash, ds.residues.device)
    mh_local_block(plan, work, planes, n_hash)
    pack_local_block(plan, work)
    return plan, work


def mh_unique_finish(uplan, plan, work, gathered, n_hash, out):
    """gathered packed shards -> symmetric count table of the unique strings -> dense float64 n x n"""
    from . import device
    table = device.shards_to_table(gathered, 0, uplan.unique, plan.world, work.bits)
    return device.expand_unique(table, uplan, False, n_hash, 0, out)


def mh_sharded_step_dedup(uplan, ds, k, n_hash, d_seeds, rank, world, out, group=None, marks=None):
    """similarityMH on `world` ranks with the duplicates collapsed; uplan = device.UniquePlan(ds.residues, ds.offsets, ...).
    marks: optional list that receives CUDA events after the local shard, the all-gather and the expansion."""
    plan, work = mh_unique_local(uplan, ds, k, n_hash, d_seeds, rank, world)
    if marks is not None:
        marks[0].record()
    if world > 1:
        gather_blocks(work.gathered, work.packed, group)
    else:
        work.gathered.copy_(work.packed)
    if marks is not None:
        marks[1].record()
    mh_unique_finish(uplan, plan, work, work.gathered, n_hash, out)
    if marks is not None:
        marks[2].record()
    return out


def nw_unique_rows_local(uplan, max_len, rank, world, matrix_name="BLOSUM62", gap_open=10, gap_ext=4, device_name="cuda"):
    """this rank's row block of the ordered unique table: cyclic 128-row units rank, rank + world, ... -> int16 [Q * 128][ld]"""
    from . import device
    U = uplan.unique
    T = -(-U // NW_TILE)
    Q = -(-T // world)
    ld = -(-U // 8) * 8
    local = torch.empty((Q * NW_TILE, ld), dtype=torch.int16, device=device_name)
    device.nw_unique_rows(uplan, max_len, matrix_name, gap_open, gap_ext, rank, world, local)     # one launch for all of the rank's units
    return local


def nw_sharded_step_dedup(uplan, max_len, rank, world, out, matrix_name="BLOSUM62", gap_open=10, gap_ext=4, group=None):
    """similarityNW on `world` ranks with the duplicates collapsed; uplan = device.UniquePlan(ds.codes, ds.offsets, ...)
    (the ENCODED residues)."""
    from . import device
    local = nw_unique_rows_local(uplan, max_len, rank, world, matrix_name, gap_open, gap_ext, out.device)
    if world > 1:
        gathered = torch.empty((world * local.shape[0], local.shape[1]), dtype=torch.int16, device=out.device)
        gather_blocks(gathered, local, group)
    else:
        gathered = local
    return device.expand_unique(gathered, uplan, True, 0, max_len, out, table_world=world)


# ---- threshold + sparsify on the shards: no N x N exchange at all -----------------------------
# Every unordered pair (i < j) is computed by exactly one rank (the owner of row i's tile row), so
#   global histogram = sum of the ranks' histograms      (ONE all-reduce of n_hash + 1 words)
#   global edge list = disjoint union of the ranks' lists (stays distributed)
# and the quantile threshold clusterbreak needs (reference R/clusterbreak.R:219-221) is exact.

def shard_histogram(plan, local, nbins):
    hist = torch.zeros(nbins, dtype=torch.int64, device=local.device)
    _capi.check(_capi.load().da_dev_shard_histogram(local.data_ptr(), local.stride(0), plan.n, plan.rank, plan.world,
                                                    int(nbins), hist.data_ptr(), _stream()))
    return hist


def shard_extract_edges(plan, local, keep, capacity):
    import numpy as np
    dev = local.device
    keep_t = torch.as_tensor(np.ascontiguousarray(keep, np.uint8)).to(dev)
    ei = torch.empty(max(capacity, 1), dtype=torch.int32, device=dev)
    ej = torch.empty(max(capacity, 1), dtype=torch.int32, device=dev)
    ev = torch.empty(max(capacity, 1), dtype=torch.int16, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    _capi.check(_capi.load().da_dev_shard_extract_edges(local.data_ptr(), local.stride(0), plan.n, plan.rank, plan.world,
                                                        keep_t.data_ptr(), keep_t.numel(), 1, ei.data_ptr(), ej.data_ptr(),
                                                        ev.data_ptr(), int(capacity), cnt.data_ptr(), _stream()))
    return ei, ej, ev, cnt


def edges_from_histograms(plan, local_hist, n_hash, thresh_p, reduce_fn, extract_fn, values=None):
    """Backend-independent part: reduce the histograms, derive threshold + keep mask, extract local edges.
    local_hist: int64 tensor [nbins] (strict upper triangle of this rank's pairs);
    reduce_fn(tensor) sums it over ranks in place; extract_fn(keep, capacity) -> (i, j, v, count).
    values[b] = the similarity bin b stands for (default: MinHash, b / n_hash); bins need not be in value order."""
    import numpy as np
    from .similarity import quantile_type7
    total = local_hist.clone()
    reduce_fn(total)
    if values is None:
        values = np.arange(n_hash + 1, dtype=np.float64) / n_hash      # src/minHash.cpp:174
    order = np.argsort(values, kind="stable")                          # order statistics need ascending VALUES
    thr = quantile_type7(total.cpu().numpy().astype(np.uint64)[order], values[order], thresh_p)
    keep = (~(values < thr)) & (values != 0.0)                         # S[S < thr] <- 0; zero weight = no edge
    mine = local_hist.cpu().numpy()
    capacity = int(mine[keep].sum()) + len(plan.my_rows())             # + this rank's diagonal entries (1.0)
    ei, ej, ev, cnt = extract_fn(keep, capacity)
    return thr, ei, ej, ev, cnt, capacity


def mh_edges_sharded(plan, work, planes, n_hash, thresh_p, group=None):
    """Device pipeline of one rank: shard compare -> local histogram -> all-reduce -> exact type-7 quantile
    -> this rank's surviving edges (i <= j, 0-based; values are match counts).  Returns
    (threshold, i, j, count_values, n_local_edges)."""
    mh_local_block(plan, work, planes, n_hash)
    hist = shard_histogram(plan, work.local, n_hash + 1)

    def reduce_fn(t):
        if plan.world > 1:
            dist.all_reduce(t, group=group)

    thr, ei, ej, ev, cnt, cap = edges_from_histograms(plan, hist, n_hash, thresh_p, reduce_fn,
                                                      lambda keep, c: shard_extract_edges(plan, work.local, keep, c))
    return thr, ei, ej, ev, cnt


def nw_code_values(max_len):
    """similarity a uint16 NW code (matches << 8 | length) stands for: matches / length (src/pairwiseSeqAlign.cpp:311)"""
    import numpy as np
    nbins = ((int(max_len) << 8) | (2 * int(max_len))) + 1
    b = np.arange(nbins)
    ln = (b & 255).astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        v = np.where(ln > 0, (b >> 8) / ln, 0.0)
    return v


def nw_edges_sharded(plan, work, ds, thresh_p, matrix_name="BLOSUM62", gap_open=10, gap_ext=4, group=None):
    """similarityNW + clusterbreak's threshold step without any N x N exchange (the NW twin of mh_edges_sharded): this
    rank's shard of alignment codes -> local histogram of the codes -> ONE all-reduce -> exact type-7 quantile of the
    ratios -> this rank's surviving edges.  Sequences of 1..64 residues (an empty one makes NaN similarities, on which R's
    quantile() stops).  Returns (threshold, i, j, codes, n_local_edges, values) -- weight of an edge = values[code]."""
    if ds.max_len > 64 or int((ds.offsets_host[1:] == ds.offsets_host[:-1]).sum()) > 0:
        raise _capi.DynaAlignError(_capi.DA_ERR_UNSUPPORTED, "sharded NW edges: sequences of 1..64 residues")
    nw_local_block(plan, work, ds, matrix_name, gap_open, gap_ext)
    values = nw_code_values(ds.max_len)
    hist = shard_histogram(plan, work.local, len(values))

    def reduce_fn(t):
        if plan.world > 1:
            dist.all_reduce(t, group=group)

    thr, ei, ej, ev, cnt, cap = edges_from_histograms(plan, hist, 0, thresh_p, reduce_fn,
                                                      lambda keep, c: shard_extract_edges(plan, work.local, keep, c), values=values)
    return thr, ei, ej, ev, cnt, values
