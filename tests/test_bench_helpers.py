"""Host-side helpers of bench.py / tools that need no GPU: the kernel-source hash that ties a PMC profile to a build, and the rule that
`roofline.traffic` is reported only from a profile measured on the running sources."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_source_hash_covers_the_kernel_sources_and_ignores_experiment_variants(tmp_path):
    import source_hash
    h = source_hash.source_hash()
    assert len(h) == 12 and int(h, 16) >= 0
    incs = source_hash.makefile_incs()
    assert {"k2_loop_p12.inc", "k2_loop_p12p.inc", "k2_loop_p16.inc", "blosum_data.inc"} <= incs and "k2_loop_p12q.inc" not in incs
    # an experiment variant dropped next to the sources (tools/k2_variants.sh writes such files) does not change the hash
    junk = os.path.join(ROOT, "dynaalign_amd", "csrc", "k2_loop_zz_test_variant.inc")
    try:
        open(junk, "w").write("// not a committed include\n")
        assert source_hash.source_hash() == h
    finally:
        os.remove(junk)


def test_traffic_is_reported_only_from_a_profile_of_the_running_sources(monkeypatch, tmp_path):
    import bench
    import source_hash
    prof = tmp_path / "profiles"
    prof.mkdir()
    cur = source_hash.source_hash()
    base = {"n": 1234, "kernels": {"k_x": {"WRITE_SIZE": 1000.0, "FETCH_SIZE": 10.0}}}
    (prof / "r99_a_pmc_summary.json").write_text(json.dumps(dict(base, source_hash=cur)))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: cur)
    t = bench.pmc_traffic("k_x", 1234)
    assert t["bytes"] == (1000.0 + 2 * 10.0) * 1024 and cur in t["source"]          # WRITE_SIZE + 2 x FETCH_SIZE (gfx950 half-count), KiB
    assert bench.pmc_traffic("k_x", 999) is None and bench.pmc_traffic("k_other", 1234) is None
    # a newer profile taken on OTHER sources wins the filename sort: stale -> no number, and the reason is said
    (prof / "r99_b_pmc_summary.json").write_text(json.dumps(dict(base, source_hash="0123456789ab")))
    t = bench.pmc_traffic("k_x", 1234)
    assert t["bytes"] is None and "STALE" in t["source"]
    # a profile without a recorded hash (rounds 1-2) is stale by definition
    (prof / "r99_c_pmc_summary.json").write_text(json.dumps(base))
    t = bench.pmc_traffic("k_x", 1234)
    assert t["bytes"] is None and "unrecorded" in t["source"]


def test_committed_pmc_profile_matches_the_committed_sources():
    """the newest committed PMC summary must have been taken on the sources in the tree -- otherwise the driver's bench line carries
    `traffic: null`; re-run tools/prof_pmc.sh after the last kernel change of a round"""
    import source_hash
    pdir = os.path.join(ROOT, "profiles")
    hashed = [f for f in sorted(os.listdir(pdir)) if f.endswith(".json") and "pmc_summary" in f
              and "source_hash" in json.load(open(os.path.join(pdir, f)))]
    newest = hashed[-1]                                            # (file names sort by round and letter: rNN_<tag>_...)
    d = json.load(open(os.path.join(pdir, newest)))
    if d.get("source_hash") != source_hash.source_hash():            # a reminder, not a gate: kernels may change before the next GPU session
        import pytest
        pytest.skip("%s was measured on other kernel sources (%s, tree %s): bench.py will print traffic null until tools/prof_pmc.sh is re-run"
                    % (newest, d.get("source_hash"), source_hash.source_hash()))
