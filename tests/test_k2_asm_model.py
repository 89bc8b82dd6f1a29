"""The generated asm of k_mh_compare_a12 (csrc/k2_loop_p12.inc) against a CPU model of its instruction stream:
register map, LDS addressing, ring bookkeeping, popcount packing, write-back order (tools/sim_k2_asm.py).
Also: the committed include is what tools/gen_k2_asm.py generates."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("tx,ty,seed", [(0, 0, 1), (5, 11, 2), (15, 15, 3), (7, 8, 4)])
def test_generated_loop_counts_what_the_definition_says(tx, ty, seed):
    import sim_k2_asm
    issued, bad = sim_k2_asm.run(tx, ty, seed)
    assert issued == 16 and bad == 0


def test_committed_include_is_the_generators_output(tmp_path):
    out = tmp_path / "k2.inc"
    env = {k: v for k, v in os.environ.items() if not k.startswith("K2ASM_")}
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_k2_asm.py"), str(out)], env=env,
                          stdout=subprocess.DEVNULL)
    assert out.read_text() == open(os.path.join(ROOT, "dynaalign_amd", "csrc", "k2_loop_p12.inc")).read()


@pytest.mark.parametrize("ntiles,ns,tx,ty", [(1, 16, 0, 0), (2, 16, 5, 11), (3, 16, 15, 15), (4, 2, 7, 8), (3, 3, 1, 2), (5, 5, 9, 4), (2, 17, 3, 3)])
def test_persistent_block_over_a_tile_sequence(ntiles, ns, tx, ty):
    """the block of k_mh_compare_p12, called ntiles times on one LDS ring: first / has-next flags, ring phase, source switch"""
    import sim_k2_asm
    issued, bad = sim_k2_asm.run_persistent(tx, ty, seed=ntiles * 100 + ns, ntiles=ntiles, ns=ns)
    assert issued == ntiles * ns and bad == 0


def test_committed_persistent_include_is_the_generators_output(tmp_path):
    out = tmp_path / "k2p.inc"
    env = {k: v for k, v in os.environ.items() if not k.startswith("K2ASM_")}
    env["K2ASM_PERSIST"] = "1"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_k2_asm.py"), str(out)], env=env,
                          stdout=subprocess.DEVNULL)
    assert out.read_text() == open(os.path.join(ROOT, "dynaalign_amd", "csrc", "k2_loop_p12p.inc")).read()


@pytest.mark.parametrize("inc", ["k2_loop_p8p.inc", "k2_loop_p8.inc"])
@pytest.mark.parametrize("ntiles,ns,tx,ty", [(1, 16, 0, 0), (2, 16, 5, 11), (3, 16, 15, 15), (4, 2, 7, 8), (3, 3, 1, 2), (5, 5, 9, 4), (2, 17, 3, 3)])
def test_8_plane_persistent_blocks(inc, ntiles, ns, tx, ty):
    """the dense half of the heavy / rare split (round 4): the persistent block on EIGHT code planes -- 32-byte LDS slots, two DMA pieces per
    wave and stage, four two-plane steps, a 24 KiB ring; k2_loop_p8p.inc is the band kernel's block, k2_loop_p8.inc the same with the stage
    loop at wave priority 2 (one tile per workgroup: called with flags first / no next only, but the tile-sequence model holds for it too)"""
    import sim_k2_asm
    path = os.path.join(ROOT, "dynaalign_amd", "csrc", inc)
    issued, bad = sim_k2_asm.run_persistent(tx, ty, seed=ntiles * 100 + ns, ntiles=ntiles, ns=ns, inc=path, planes=8, slot_bytes=32, ring=3)
    assert issued == ntiles * ns and bad == 0


@pytest.mark.parametrize("inc,prio", [("k2_loop_p8p.inc", None), ("k2_loop_p8.inc", "2")])
def test_committed_8_plane_includes_are_the_generators_output(tmp_path, inc, prio):
    out = tmp_path / "k2_8.inc"
    env = {k: v for k, v in os.environ.items() if not k.startswith("K2ASM_")}
    env.update({"K2ASM_PERSIST": "1", "K2ASM_PLANES": "8"})
    if prio:
        env["K2ASM_PRIO"] = prio
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_k2_asm.py"), str(out)], env=env, stdout=subprocess.DEVNULL)
    assert out.read_text() == open(os.path.join(ROOT, "dynaalign_amd", "csrc", inc)).read()


@pytest.mark.parametrize("ns,tx,ty", [(16, 0, 0), (1, 5, 11), (2, 15, 15), (3, 7, 8), (5, 9, 4)])
def test_16_plane_block(ns, tx, ty):
    """the block of k_mh_compare_a16: padded 80-byte slots, ring of two stages, five DMA pieces per wave and stage"""
    import sim_k2_asm
    issued, bad = sim_k2_asm.run16(tx, ty, seed=ns, ns=ns)
    assert issued == ns and bad == 0


def test_committed_16_plane_include_is_the_generators_output(tmp_path):
    out = tmp_path / "k2_16.inc"
    env = {k: v for k, v in os.environ.items() if not k.startswith("K2ASM_")}
    env["K2ASM_PLANES"] = "16"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_k2_asm.py"), str(out)], env=env,
                          stdout=subprocess.DEVNULL)
    assert out.read_text() == open(os.path.join(ROOT, "dynaalign_amd", "csrc", "k2_loop_p16.inc")).read()


@pytest.mark.parametrize("ns,tx,ty", [(16, 0, 0), (1, 5, 11), (2, 15, 15), (3, 7, 8)])
@pytest.mark.parametrize("bits", [14, 15])
def test_14_and_15_bit_blocks(ns, tx, ty, bits):
    """the shortened forms of the 16-plane block: 14 code bits = seven steps (planes 14 / 15 never read), 15 = the eighth step on
    plane 14 only"""
    import sim_k2_asm
    issued, bad = (sim_k2_asm.run14 if bits == 14 else sim_k2_asm.run15)(tx, ty, seed=ns, ns=ns)
    assert issued == ns and bad == 0


@pytest.mark.parametrize("bits", [14, 15])
def test_committed_14_15_bit_includes_are_the_generators_output(tmp_path, bits):
    out = tmp_path / "k2_x.inc"
    env = {k: v for k, v in os.environ.items() if not k.startswith("K2ASM_")}
    env["K2ASM_PLANES"] = str(bits)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_k2_asm.py"), str(out)], env=env,
                          stdout=subprocess.DEVNULL)
    assert out.read_text() == open(os.path.join(ROOT, "dynaalign_amd", "csrc", "k2_loop_p%d.inc" % bits)).read()


@pytest.mark.parametrize("ntiles,ns,tx,ty", [(1, 16, 0, 0), (3, 16, 5, 11), (4, 8, 15, 15), (3, 4, 7, 8), (2, 16, 9, 3)])
def test_inloop_store_block_over_a_tile_sequence(ntiles, ns, tx, ty):
    """the block of k_mh_compare_q12 (round 3; an experiment kept under tools/experiments/, not in the product library): counters as before; the PREVIOUS tile's 64 float64 stores issued from inside the stage
    loop carry the right addresses and values (direct + mirrored, 16 / ns pieces per stage); and the counted vmcnt waits hold in
    an in-order model of loads and stores (a stage is never read before its three DMA pieces have retired)"""
    import sim_k2_asm
    issued, bad, bad_stores = sim_k2_asm.run_inloop(tx, ty, seed=ntiles * 100 + ns, ntiles=ntiles, ns=ns)
    assert issued == ntiles * ns and bad == 0 and bad_stores == 0


def test_committed_inloop_include_is_the_generators_output(tmp_path):
    out = tmp_path / "k2q.inc"
    env = {k: v for k, v in os.environ.items() if not k.startswith("K2ASM_")}
    env["K2ASM_INLOOP"] = "1"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_k2_asm.py"), str(out)], env=env,
                          stdout=subprocess.DEVNULL)
    assert out.read_text() == open(os.path.join(ROOT, "tools", "experiments", "k2_loop_p12q.inc")).read()
