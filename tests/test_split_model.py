"""The identity behind the heavy / rare split of the MinHash compare (csrc/dict_kernels.hip k_hy_split, csrc/minhash_kernels.hip k_hy_fixup*), on the CPU with
the oracle's signatures: per hash function pick the `keep` most frequent signature values ("heavy"); then for every pair
    matches(i, j) = #{h: equal and heavy in column h} + #{h: equal and not heavy in column h}
whatever the choice of the heavy set -- ties at the threshold included -- and the second term is what the incidence lists enumerate: sum over the
non-heavy repeated values of m (m - 1) / 2.  (The GPU tests hold the kernels against the oracle; this one holds the arithmetic the design rests on.)"""
import numpy as np
import pytest

import oracle_lib as O


def split_counts(sig, keep, rng):
    """dense part on re-coded values (everything not heavy reads 'never equal'), rare part from per-class incidence lists; heavy set chosen with a
    random tie-break among values of equal frequency"""
    n, n_hash = sig.shape
    dense = np.zeros((n, n), np.int64)
    rare = np.zeros((n, n), np.int64)
    rare_incidences = 0
    for h in range(n_hash):
        vals, inv, cnt = np.unique(sig[:, h], return_inverse=True, return_counts=True)
        order = np.lexsort((rng.rand(len(vals)), -cnt))              # by frequency, ties in random order
        heavy = np.zeros(len(vals), bool)
        heavy[order[:keep]] = True
        heavy &= cnt >= 2                                            # a value seen once can never match
        code = np.where(heavy[inv], inv, -1)
        eq = (code[:, None] == code[None, :]) & (code[:, None] >= 0)
        dense += eq
        for v in np.nonzero(~heavy & (cnt >= 2))[0]:                 # the incidence lists: every pair inside a rare class
            members = np.nonzero(inv == v)[0]
            rare[np.ix_(members, members)] += 1
            rare_incidences += len(members) * (len(members) - 1) // 2
    return dense, rare, rare_incidences


@pytest.mark.parametrize("keep", [1, 3, 14, 254])
def test_dense_plus_rare_is_the_oracles_count(keep):
    rng = np.random.RandomState(keep)
    alpha = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    par = alpha[rng.randint(0, 20, (3, 120))]
    seqs = []
    for _ in range(260):
        p, s0 = rng.randint(0, 3), rng.randint(0, 101)
        w = par[p, s0:s0 + 20].copy()
        hit = rng.rand(20) < 0.05
        w[hit] = alpha[rng.randint(0, 20, int(hit.sum()))]
        seqs.append(w.tobytes().decode("latin-1"))
    seqs += [seqs[k] for k in rng.randint(0, 260, 40)]               # exact duplicates: classes with equal frequencies (ties)
    n_hash = 48
    seeds = O.seeds(12345, n_hash)
    sig = np.asarray(O.signatures(seqs, 4, n_hash, seeds)).reshape(len(seqs), n_hash)
    want = np.asarray(O.mh_counts(sig)).astype(np.int64)
    dense, rare, inc = split_counts(sig, keep, rng)
    total = dense + rare
    off = ~np.eye(len(seqs), dtype=bool)
    assert np.array_equal(total[off], want[off])                     # (the diagonal is forced by the kernels: src/minHash.cpp:161)
    assert inc == int(np.triu(rare, 1).sum())
    if keep == 1:
        assert inc > 0
