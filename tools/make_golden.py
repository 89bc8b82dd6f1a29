#!/usr/bin/env python3
"""Generate tests/golden/oracle_vectors.npz: seeded inputs -> CPU-oracle outputs.

The vectors come from the repo's own oracle (oracle/liborc.so), which is itself
pinned to the reference-produced known answers in survey_kats.json.  They serve
(a) as a drift detector for the oracle and (b) as committed fixtures the GPU
parity tests compare against without re-running the oracle.

    python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import oracle_lib as O  # noqa: E402
from dynaalign_amd import synth  # noqa: E402


def mixed_sequences():
    """64 sequences exercising the edge cases SURVEY 8(c) lists: L<k, L=k, repeated
    k-mers, non-AA bytes (legal for MH), empty strings, duplicates."""
    rng = np.random.RandomState(20241220)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    seqs = ["", "A", "AC", "ACD", "ACDE", "ACDEF", "AAAAAAAAAAAA", "ACACACACACAC", "acdefghik", "AC-DE FG_hi",
            "XBZ*XBZ*XBZ*", "STSIPALTAVET", "STSIPALTAVET", "TSIPALTAVETG"]
    while len(seqs) < 64:
        L = int(rng.randint(0, 31))
        seqs.append("".join(aa[i] for i in rng.randint(0, 20, L)))
    return seqs


def nw_sequences():
    """96 valid-alphabet sequences: random 20-mers, low-complexity 3-letter strings (force
    gaps), unequal lengths 0..30, B Z X * residues, duplicates, the SURVEY asymmetric pair."""
    rng = np.random.RandomState(615)
    aa = "ARNDCQEGHILKMFPSTWYVBZX*"
    seqs = ["", "A", "W", "YDYIHIYADKQDRIGWLGNT", "MYCEMNVEIQYMATKNMWNT", "RRAVELQTVAFP", "PPPSYETVMAAA",
            "TPPPSYETVMAA", "TPPASYHTVMAA", "BZX*BZX*", "********"]
    while len(seqs) < 40:
        seqs.append("".join(aa[i] for i in rng.randint(0, 20, 20)))
    while len(seqs) < 70:
        L = int(rng.randint(0, 31))
        seqs.append("".join("AGW"[i] for i in rng.randint(0, 3, L)))
    while len(seqs) < 96:
        L = int(rng.randint(0, 31))
        seqs.append("".join(aa[i] for i in rng.randint(0, 24, L)))
    return seqs


def main():
    out = {}
    mix = mixed_sequences()
    out["mh_sequences"] = np.array(mix, dtype=object).astype("U")
    for k in (1, 2, 3, 4, 5, 7):
        for n_hash in (8, 50, 500):
            seeds = O.seeds(12345, n_hash)
            sig = O.signatures(mix, k, n_hash, seeds)
            out["mh_sig_k%d_h%d" % (k, n_hash)] = sig
            out["mh_cnt_k%d_h%d" % (k, n_hash)] = O.mh_counts(sig)
    res, off = synth.uniform_peptides(256, 20, seed=7)
    u = synth.to_strings(res, off)
    out["uniform256"] = np.array(u).astype("U")
    sig = O.signatures(u, 4, 500, O.seeds(12345, 500))
    out["uniform256_sig_k4_h500"] = sig
    out["uniform256_cnt_k4_h500"] = O.mh_counts(sig)
    res, off = synth.h3n2_like(256, 20)
    hl = synth.to_strings(res, off)
    out["h3n2like256"] = np.array(hl).astype("U")
    sig = O.signatures(hl, 4, 500, O.seeds(12345, 500))
    out["h3n2like256_cnt_k4_h500"] = O.mh_counts(sig)

    nws = nw_sequences()
    out["nw_sequences"] = np.array(nws).astype("U")
    for name, go, ge in (("BLOSUM62", 10, 4), ("BLOSUM45", 10, 4), ("BLOSUM50", 12, 2), ("BLOSUM80", 0, 0),
                         ("BLOSUM90", 5, 1), ("BLOSUM100", 10, 4), ("BLOSUM62", 1, 7), ("BLOSUM62", 3, 0)):
        rc, mt, ln, sc, msg = O.nw_rows(nws, 0, None, name, go, ge)
        assert rc == 0, msg
        tag = "nw_%s_%d_%d" % (name, go, ge)
        out[tag + "_matches"] = mt.astype(np.int16)
        out[tag + "_len"] = ln.astype(np.int16)
        out[tag + "_score"] = sc
    rc, mt, ln, sc, msg = O.nw_rows(u, 0, None, "BLOSUM62", 10, 4)
    out["uniform256_nw_matches"] = mt.astype(np.int16)
    out["uniform256_nw_len"] = ln.astype(np.int16)
    out["uniform256_nw_score"] = sc
    path = os.path.join(ROOT, "tests", "golden", "oracle_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
