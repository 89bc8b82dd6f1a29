"""`python bench.py --gpus N` with no launcher around it must start N ranks itself: argument forwarding,
the rank environment, failure propagation, the time limit and the insufficient-device error -- with the child command stubbed,
no GPU and no torch.distributed involved.  The real two-rank run (gloo on the one-GPU box) is kept under profiles/r04_*."""
import io
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402  (imports nothing heavy at module level)


def stub(tmp_path, body):
    p = tmp_path / "child.py"
    p.write_text("import json, os, sys, time\nrank = int(os.environ['RANK'])\n" + textwrap.dedent(body))
    return [sys.executable, str(p)]


def run(n, argv, cmd, **kw):
    out, err = io.StringIO(), io.StringIO()
    rc = bench.launch_ranks(n, argv, child_cmd=cmd, device_count=kw.pop("device_count", lambda: 8), out=out, err=err, **kw)
    return rc, out.getvalue(), err.getvalue()


def test_ranks_get_their_environment_and_the_arguments_and_rank0s_line_is_relayed(tmp_path):
    cmd = stub(tmp_path, """
        rec = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")}
        rec["argv"] = sys.argv[1:]
        open(os.path.join(os.path.dirname(__file__), "rank%d.json" % rank), "w").write(json.dumps(rec))
        print("progress text of rank %d" % rank)
        print("[stderr] rank %d" % rank, file=sys.stderr)
        if rank == 0:
            print(json.dumps({"n_gpus": int(os.environ["WORLD_SIZE"]), "value": 1.0}))
    """)
    argv = ["--gpus", "3", "--steps", "2", "--warmup", "1", "--no-cpu"]
    rc, out, err = run(3, argv, cmd)
    assert rc == 0
    lines = out.strip().splitlines()
    assert json.loads(lines[-1]) == {"n_gpus": 3, "value": 1.0}          # the result is the LAST stdout line
    assert "progress text of rank 0" in out and "rank 1" not in out       # only rank 0's stdout is relayed
    assert "[rank 2] [stderr] rank 2" in err
    recs = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(3)]
    assert [r["RANK"] for r in recs] == ["0", "1", "2"] and [r["LOCAL_RANK"] for r in recs] == ["0", "1", "2"]
    assert all(r["WORLD_SIZE"] == "3" and r["MASTER_ADDR"] == "127.0.0.1" and r["argv"] == argv for r in recs)
    assert len({r["MASTER_PORT"] for r in recs}) == 1 and int(recs[0]["MASTER_PORT"]) > 0
    assert all(r["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for r in recs)


def test_a_failing_rank_fails_the_run_and_stops_the_others(tmp_path):
    cmd = stub(tmp_path, """
        if rank == 1:
            sys.exit(7)
        if rank == 0:
            print(json.dumps({"n_gpus": 2}), flush=True)
        time.sleep(120)               # ranks stuck in a collective whose peer died
    """)
    import time
    t0 = time.monotonic()
    rc, out, err = run(2, [], cmd)
    assert rc == 7 and time.monotonic() - t0 < 60
    assert "rank 1 exited with 7" in err
    assert out.strip() == "" and "NOT a result" in err                     # rank 0's line is not passed on as a result


def test_a_rank_killed_by_a_signal_is_a_failure(tmp_path):
    cmd = stub(tmp_path, """
        import signal
        if rank == 0:
            print(json.dumps({"n_gpus": 2}), flush=True)
            os.kill(os.getpid(), signal.SIGABRT)      # what torch's watchdog does on a failed / timed-out collective
        time.sleep(120)
    """)
    rc, out, err = run(2, [], cmd)
    assert rc != 0 and out.strip() == ""


def test_no_result_line_is_a_failure(tmp_path):
    rc, out, err = run(2, [], stub(tmp_path, "print('no json here')\n"))
    assert rc == 1 and "no result line" in err


def test_hung_ranks_are_stopped_at_the_time_limit(tmp_path):
    rc, out, err = run(2, [], stub(tmp_path, "time.sleep(300)\n"), timeout_s=1.0)
    assert rc == 124 and "still running" in err


def test_fewer_devices_than_ranks_is_an_error_with_rccl_and_fine_for_the_gloo_rehearsal(tmp_path, monkeypatch):
    marker = tmp_path / "started"
    cmd = stub(tmp_path, "open(%r, 'a').write('x')\nif rank == 0: print(json.dumps({'n_gpus': 2}))\n" % str(marker))
    monkeypatch.delenv("DYNAALIGN_BENCH_BACKEND", raising=False)
    rc, out, err = run(2, [], cmd, device_count=lambda: 1)
    assert rc == 3 and "needs 2 visible GPUs" in err and not marker.exists()          # nothing was started, nothing fell back to one GPU
    monkeypatch.setenv("DYNAALIGN_BENCH_BACKEND", "gloo")
    rc, out, err = run(2, [], cmd, device_count=lambda: 1)
    assert rc == 0 and marker.read_text() == "xx"


def test_gpus_n_without_a_launcher_goes_through_launch_ranks(monkeypatch):
    """main(): WORLD_SIZE unset and --gpus 4 -> build (CPU-only) then launch_ranks(4, argv), the exit code passed on; and a rank
    started with a WORLD_SIZE that contradicts --gpus refuses to run"""
    calls = {}
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.setattr(bench, "launch_ranks", lambda n, argv, **kw: calls.update(n=n, argv=argv) or 5)
    import __graft_entry__ as g
    monkeypatch.setattr(g, "build", lambda: calls.update(built=True))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 5 and calls == {"built": True, "n": 4, "argv": ["--gpus", "4", "--steps", "3"]}
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "--gpus 4 but WORLD_SIZE=2" in str(e.value.code)


def test_the_parent_process_does_not_initialise_the_gpu():
    """the launching parent imports torch only to count devices; it must not create a HIP context (on the pool, exec / fork from a
    process that initialised the GPU is refused): checked on the source -- nothing between parse() and launch_ranks touches torch.cuda
    except device_count()"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[src.index("def main():"):src.index("rank = int(os.environ.get(\"RANK\"")]
    assert "torch" not in head
    body = src[src.index("def launch_ranks("):src.index("def main():")]
    assert "torch.cuda" not in body and "os.exec" not in src and "execv" not in src
