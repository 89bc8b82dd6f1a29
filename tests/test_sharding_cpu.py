"""CPU tests of the N>1 path: the row-sharding plan and the single all-gather, run with
world_size 2 (and 3) over gloo.  There is no CPU compute path in the product, so the ranks'
blocks are produced here by the test oracle; what is under test is the partition, the exchange
and the reassembly rule the HIP finalize kernel implements (its GPU twin is in test_gpu_sharding)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from dynaalign_amd import sharding, synth


def finalize_reference(plan, gathered, widen):
    """numpy statement of k_finalize_sharded: out[i][j] = widen(G[entry of (min(i,j), max(i,j))])"""
    n = plan.n
    iu = np.triu_indices(n)
    rows = np.empty(len(iu[0]), np.int64)
    cols = np.empty(len(iu[0]), np.int64)
    for e, (i, j) in enumerate(zip(*iu)):
        p, r, c = plan.locate(int(i), int(j))
        rows[e], cols[e] = p * plan.local_rows + r, c
    out = np.empty((n, n), np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        out[iu] = widen(gathered[rows, cols])
    out.T[iu] = out[iu]
    return out


@pytest.mark.parametrize("n,world,tile", [(1, 1, 128), (100, 2, 128), (128, 2, 128), (129, 2, 128), (1000, 8, 128),
                                          (1000, 3, 64), (64, 8, 64), (300, 2, 64), (300, 3, 64), (257, 1, 128),
                                          (100000, 8, 128)])
def test_plan_partitions_rows_exactly_once(n, world, tile):
    plans = [sharding.Plan(n, r, world, tile) for r in range(world)]
    if n <= 2000:
        owned = sorted(i for p in plans for i in p.my_rows())
        assert owned == list(range(n))
        for p in plans:
            for i in p.my_rows():
                assert p.owner(i) == p.rank
    if n <= 300:                                             # no two valid elements share a slot of the folded block
        seen = set()
        for i in range(n):
            for j in range((i // tile) * tile, n):
                p, r, c = plans[0].locate(i, j)
                assert 0 <= r < plans[0].local_rows and 0 <= c < plans[0].width
                key = (p, r, c)
                assert key not in seen
                seen.add(key)
    assert len({p.local_rows for p in plans}) == 1          # equal blocks: legal all-gather
    work = [p.upper_pairs() for p in plans]
    if n >= 50 * world * tile:                               # cyclic dealing balances the triangle
        assert max(work) / (sum(work) / world) < 1.05


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_main(rank, world, port, n, kind, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res, off = synth.h3n2_like(n, 20)
        seqs = synth.to_strings(res, off)
        tile = sharding.MH_TILE if kind == "mh" else sharding.NW_TILE
        plan = sharding.Plan(n, rank, world, tile)
        rows = plan.my_rows()
        local = np.full((plan.local_rows, plan.width), 0x7FFF, np.int16)     # poison: must never be read
        if kind == "mh":
            sig = O.signatures(seqs, 4, 64, O.seeds(12345, 64))
            for i in rows:
                t0 = (i // tile) * tile
                cnt = (sig[i][None, :] == sig[t0:]).sum(1).astype(np.uint16)
                _, r, c0 = plan.locate(i, t0)
                local[r, c0:c0 + n - t0] = cnt.view(np.int16)
            widen = lambda v: v.view(np.uint16).astype(np.float64) / 64
        else:
            for i in rows:
                rc, mt, ln, _, _ = O.nw_rows(seqs, i, i + 1)
                v = ((mt[0] << 8) | ln[0]).astype(np.uint16)
                _, r, c0 = plan.locate(i, i)
                local[r, c0:c0 + n - i] = v[i:].view(np.int16)       # only j >= i is valid for NW

            def widen(v):
                u = v.view(np.uint16).astype(np.uint32)
                return (u >> 8).astype(np.float64) / (u & 255).astype(np.float64)
        local_t = torch.from_numpy(local)
        gathered = torch.empty((world * plan.local_rows, plan.width), dtype=torch.int16)
        out = sharding.all_pairs_sharded(plan, local_t, gathered,
                                         lambda g: finalize_reference(plan, g.numpy(), widen))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,kind", [(2, 300, "mh"), (3, 333, "mh"), (2, 150, "nw")])
def test_sharded_all_gather_reassembles_the_oracle_matrix(world, n, kind):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, n, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    seqs = synth.to_strings(*synth.h3n2_like(n, 20))
    if kind == "mh":
        rc, want = O.similarity_mh(seqs, 4, 64, O.seeds(12345, 64))
    else:
        rc, want, _ = O.similarity_nw(seqs)
    assert rc == 0
    for rank, out in results:                                        # every rank ends with the full matrix
        assert np.array_equal(out.view(np.uint64), want.view(np.uint64)), rank


def _edges_rank_main(rank, world, port, n, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import dynaalign_amd  # noqa: F401  (host-side quantile lives in the library)
        n_hash, p = 64, 0.8
        seqs = synth.to_strings(*synth.h3n2_like(n, 20))
        sig = O.signatures(seqs, 4, n_hash, O.seeds(12345, n_hash))
        plan = sharding.Plan(n, rank, world, sharding.MH_TILE)
        cnt = {}
        for i in plan.my_rows():
            row = (sig[i][None, :] == sig[i:]).sum(1)
            for j, c in enumerate(row):
                cnt[(i, i + j)] = int(c)
        hist = np.bincount([c for (i, j), c in cnt.items() if j > i], minlength=n_hash + 1).astype(np.int64)

        def extract(keep, capacity):
            e = sorted((i, j, c) for (i, j), c in cnt.items() if keep[c] or i == j)
            assert len(e) == capacity
            return [x[0] for x in e], [x[1] for x in e], [x[2] for x in e], len(e)

        thr, ei, ej, ev, c, cap = sharding.edges_from_histograms(plan, torch.from_numpy(hist), n_hash, p,
                                                                 lambda t: dist.all_reduce(t), extract)
        q.put((rank, thr, list(zip(ei, ej, ev))))
    finally:
        dist.destroy_process_group()


def test_sharded_edges_all_reduce_matches_dense_threshold(built):
    """world_size 2 over gloo: one all-reduce of the histogram, then disjoint local edge lists"""
    from test_threshold_edges import reference_edges
    world, n = 2, 300
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_edges_rank_main, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    seqs = synth.to_strings(*synth.h3n2_like(n, 20))
    rc, M = O.similarity_mh(seqs, 4, 64, O.seeds(12345, 64))
    thr_w, iw, jw, ww = reference_edges(M, 0.8)
    got = sorted(e for _, _, edges in results for e in edges)
    assert {thr for _, thr, _ in results} == {thr_w}
    assert [(a, b) for a, b, _ in got] == list(zip(iw.tolist(), jw.tolist()))
    assert [c / 64 for _, _, c in got] == ww.tolist()


def _seed_rank_main(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.pop("DYNAALIGN_SEED", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import dynaalign_amd  # noqa: F401
        own = dynaalign_amd.similarity._resolve_seed(None)              # what each process would draw on its own
        s = sharding.shared_seed(None)                                  # the reference's default: no seed given
        fam = sharding.shared_hash_family(16, None)                     # a second draw: again one family for all ranks
        explicit = sharding.shared_seed(1000 + rank)                    # explicit but inconsistent seeds: rank 0's wins
        q.put((rank, own, s, fam.tolist(), explicit))
    finally:
        dist.destroy_process_group()


def test_default_seed_is_one_hash_family_for_all_ranks(built):
    """seed=None (std::random_device, reference src/minHash.cpp:73) resolved per process would make the ranks hash
    with different families; shared_seed broadcasts rank 0's draw."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_seed_rank_main, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, own0, s0, fam0, e0), (_, own1, s1, fam1, e1) = results
    assert s0 == s1 and fam0 == fam1 and len(fam0) == 16
    assert e0 == e1 == 1000
    assert 0 <= s0 < 2 ** 32


def _nw_edges_rank_main(rank, world, port, n, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import dynaalign_amd  # noqa: F401  (host-side quantile lives in the library)
        seqs = synth.to_strings(*synth.h3n2_like(n, 20))
        plan = sharding.Plan(n, rank, world, sharding.NW_TILE)
        values = sharding.nw_code_values(20)
        code = {}
        for i in plan.my_rows():
            rc, mt, ln, _, _ = O.nw_rows(seqs, i, i + 1)
            for j in range(i, n):
                code[(i, j)] = (int(mt[0][j]) << 8) | int(ln[0][j])
        hist = np.bincount([c for (i, j), c in code.items() if j > i], minlength=len(values)).astype(np.int64)

        def extract(keep, capacity):
            e = sorted((i, j, c) for (i, j), c in code.items() if keep[c] or i == j)
            assert len(e) == capacity
            return [x[0] for x in e], [x[1] for x in e], [x[2] for x in e], len(e)

        thr, ei, ej, ev, c, cap = sharding.edges_from_histograms(plan, torch.from_numpy(hist), 0, 0.8,
                                                                 lambda t: dist.all_reduce(t), extract, values=values)
        q.put((rank, thr, [(a, b, float(values[v])) for a, b, v in zip(ei, ej, ev)]))
    finally:
        dist.destroy_process_group()


def test_sharded_nw_edges_all_reduce_matches_dense_threshold(built):
    """the NW twin: bins are alignment codes whose VALUES (matches / length) are not in bin order -- the quantile runs over
    the values; world_size 2 over gloo, one all-reduce of the code histogram, disjoint local edge lists"""
    from test_threshold_edges import reference_edges
    world, n = 2, 260
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nw_edges_rank_main, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    seqs = synth.to_strings(*synth.h3n2_like(n, 20))
    rc, M, _ = O.similarity_nw(seqs)
    thr_w, iw, jw, ww = reference_edges(M, 0.8)
    got = sorted(e for _, _, edges in results for e in edges)
    assert {thr for _, thr, _ in results} == {thr_w}
    assert [(a, b) for a, b, _ in got] == list(zip(iw.tolist(), jw.tolist()))
    assert [w for _, _, w in got] == ww.tolist()


def _dup_nw_rank_main(rank, world, port, q):
    """CPU rehearsal of sharding.nw_sharded_step_dedup: the plan's numbering, the cyclic 128-row units of the ORDERED unique table,
    one all-gather, sharding.unique_table_row + the expansion rule out[i][j] = table[u(min)][u(max)] -- device kernels replaced by
    the oracle, the collective by gloo"""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.RandomState(5)
        alpha = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
        mk = lambda: "".join(map(chr, alpha[rng.randint(0, 20, rng.randint(6, 12))]))
        pool = [mk() for _ in range(40)] + ["YDYIHIYADKQDRIGWLGNT", "MYCEMNVEIQYMATKNMWNT"]
        seqs = [pool[k] for k in rng.randint(0, len(pool), 160)] + [mk() for _ in range(300)]
        rng.shuffle(seqs)
        n = len(seqs)
        # the plan: multi-copy strings first, then single-copy ones, each group in order of first occurrence (nw_kernels.hip k_dd_assign)
        first, mult = {}, {}
        for i, s_ in enumerate(seqs):
            first.setdefault(s_, i)
            mult[s_] = mult.get(s_, 0) + 1
        uniq = sorted(first, key=lambda s_: (mult[s_] == 1, first[s_]))
        uid = {s_: u for u, s_ in enumerate(uniq)}
        U = len(uniq)
        T = -(-U // sharding.NW_TILE)
        Q = -(-T // world)
        ld = -(-U // 8) * 8
        local = np.full((Q * sharding.NW_TILE, ld), 0x7FFF, np.int16)
        for qq in range(Q):
            t = qq * world + rank
            for r in range(t * sharding.NW_TILE, min((t + 1) * sharding.NW_TILE, U)):
                row = np.zeros(U, np.uint16)
                for c in range(U):
                    rc, mt, ln, _, _ = O.nw_pair(uniq[r], uniq[c])               # ORDERED: uniq[r] is sequence1
                    assert rc == 0
                    row[c] = (mt << 8) | ln
                local[qq * sharding.NW_TILE + r % sharding.NW_TILE, :U] = row.view(np.int16)
        gathered = torch.empty((world * local.shape[0], ld), dtype=torch.int16)
        dist.all_gather_into_tensor(gathered.view(torch.uint8), torch.from_numpy(local).view(torch.uint8))
        g = gathered.numpy().view(np.uint16)
        out = np.empty((n, n), np.float64)
        for i in range(n):
            for j in range(i, n):
                v = int(g[sharding.unique_table_row(uid[seqs[i]], world, U), uid[seqs[j]]])
                out[i, j] = out[j, i] = (v >> 8) / (v & 255)
        q.put((rank, out, seqs))
    finally:
        dist.destroy_process_group()


def test_duplicate_route_row_blocks_reassemble_the_oracle_matrix_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dup_nw_rank_main, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rc, want, _ = O.similarity_nw(results[0][2])
    assert rc == 0
    for rank, out, _ in results:
        assert np.array_equal(out.view(np.uint64), want.view(np.uint64)), rank
