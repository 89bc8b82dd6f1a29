"""Deterministic synthetic peptide sets for tests and bench.py.

All generators are pure functions of their seeds; the random stream is the raw
32-bit output of mt19937 (the same engine the reference uses for its hash
seeds), so the sets can be regenerated bit-for-bit anywhere.

    uniform_peptides : "S10k / S100k" -- i.i.d. uniform residues over the 20
                       standard amino acids, residue = AA[draw % 20], seed 7.
    h3n2_like        : "H100k" -- windows of 8 random 566-residue parent
                       sequences, clade mix 5124:2514:343:104:8:8:1:1 (the
                       clade table of the reference's data/h3n2sample.rda,
                       SURVEY A.4), 3 % point mutation: many near-duplicates
                       and high-Jaccard pairs, like real HA windows.
"""
import numpy as np

AA20 = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
CLADE_MIX = (5124, 2514, 343, 104, 8, 8, 1, 1)


def mt19937_raw(seed, count):
    """`count` raw uint32 outputs of std::mt19937(seed)."""
    rs = np.random.RandomState(int(seed) & 0xFFFFFFFF)
    return rs.randint(0, 2 ** 32, size=int(count), dtype=np.uint64).astype(np.uint32)


def _pack_fixed(mat):
    n, length = mat.shape
    return np.ascontiguousarray(mat).reshape(-1), np.arange(n + 1, dtype=np.int64) * length


def uniform_peptides(n, length=20, seed=7):
    """-> (residues uint8[n*length], offsets int64[n+1])"""
    draws = mt19937_raw(seed, n * length)
    return _pack_fixed(AA20[draws % 20].reshape(n, length))


def h3n2_like(n, length=20, parent_len=566, seed_parents=11, seed_samples=13, mut_percent=3):
    """-> (residues uint8[n*length], offsets int64[n+1])"""
    nclade = len(CLADE_MIX)
    parents = AA20[mt19937_raw(seed_parents, nclade * parent_len) % 20].reshape(nclade, parent_len)
    stride = 2 + 2 * length  # clade, start, then (mutate?, replacement) per residue
    d = mt19937_raw(seed_samples, n * stride).reshape(n, stride)
    cum = np.cumsum(CLADE_MIX)
    clade = np.searchsorted(cum, d[:, 0] % cum[-1], side="right")
    start = d[:, 1] % (parent_len - length + 1)
    idx = start[:, None].astype(np.int64) + np.arange(length)[None, :]
    seqs = parents[clade[:, None], idx]
    mutate = (d[:, 2::2] % 100) < mut_percent
    repl = AA20[d[:, 3::2] % 20]
    seqs = np.where(mutate, repl, seqs).astype(np.uint8)
    return _pack_fixed(seqs)


def to_strings(residues, offsets):
    b = residues.tobytes()
    return [b[offsets[i]:offsets[i + 1]].decode("latin-1") for i in range(len(offsets) - 1)]
