"""bench.py argument handling for N > 1 (CPU only): `python bench.py --gpus N` must be runnable the
way the driver starts it -- without a launcher it takes the single-process partitioned path, under
torch.distributed.run (WORLD_SIZE set) the one-process-per-GPU path -- and never exits with "needs
torch.distributed.run"."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_device_list_parsing():
    import bench
    assert bench.parse_devices("", 1) == [0]
    assert bench.parse_devices("", 8) == list(range(8))
    assert bench.parse_devices("0,0", 1) == [0, 0]
    assert bench.parse_devices("3,1,2", 1) == [3, 1, 2]
    for bad in ("a,b", "0,-1", ",", "0," * 40 + "0"):
        with pytest.raises(SystemExit):
            bench.parse_devices(bad, 1)


def test_dispatch(monkeypatch):
    """Which driver a command line reaches (the drivers themselves are stubbed: no GPU here)."""
    import bench
    calls = []
    monkeypatch.setattr(bench, "run_single", lambda a: calls.append(("single", a.device_list)))
    monkeypatch.setattr(bench, "run_multi", lambda a: calls.append(("multi", a.device_list)))
    monkeypatch.setattr(bench, "run_dist", lambda a, w, r, lr: calls.append(("dist", w, r)))

    def go(argv, env=()):
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env:
            monkeypatch.setenv(k, v)
        monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
        bench.main()
        return calls[-1]

    assert go([]) == ("single", [0])
    assert go(["--gpus", "1"]) == ("single", [0])
    assert go(["--gpus", "8"]) == ("multi", list(range(8)))                 # how the driver starts it
    assert go(["--gpus", "2", "--steps", "3"]) == ("multi", [0, 1])
    assert go(["--devices", "0,0"]) == ("multi", [0, 0])                     # logical partitions
    assert go(["--gpus", "4", "--devices", "0,0,0"]) == ("multi", [0, 0, 0])
    assert go(["--devices", "0"]) == ("multi", [0])                          # explicit list: multi path
    assert go(["--gpus", "8"], env=(("WORLD_SIZE", "8"), ("RANK", "3"), ("LOCAL_RANK", "3"))) == ("dist", 8, 3)
    assert go(["--gpus", "1"], env=(("WORLD_SIZE", "1"), ("RANK", "0"))) == ("single", [0])


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful without a GPU")
def test_gpus_2_without_a_launcher_reaches_the_partitioned_path_and_fails_loudly_without_a_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "256",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "torch.distributed.run" not in r.stderr
    assert "no HIP device" in r.stderr or "libfwx" in r.stderr, r.stderr


def test_watchdog_prints_an_error_line_and_exits_with_code_3():
    """A phase that outlives its bound: ONE JSON line naming the phase (and the phases that did finish),
    flushed, then exit code 3 -- never a silent driver kill.  A phase that finishes in time prints nothing."""
    import io
    import json
    import time
    import bench
    out, codes = io.StringIO(), []
    dog = bench.Watchdog(0.3, {"metric": "m", "n_gpus": 8}, out=out, exit_fn=codes.append)
    dog.arm("warm-up step 1")
    dog.disarm()
    time.sleep(0.5)                                   # disarmed: the clock is not running
    assert out.getvalue() == "" and codes == []
    dog.arm("timed step 2 of 5")
    for _ in range(40):
        if codes:
            break
        time.sleep(0.1)
    assert codes == [3]
    lines = out.getvalue().strip().splitlines()
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["hung_phase"] == "timed step 2 of 5" and line["value"] is None and "watchdog" in line["error"]
    assert line["phases_completed"] == ["warm-up step 1"] and line["n_gpus"] == 8
    # a non-emitting rank (rank != 0 under torchrun) exits the same way without printing
    out2, codes2 = io.StringIO(), []
    dog2 = bench.Watchdog(0.2, {}, emit=False, out=out2, exit_fn=codes2.append)
    dog2.arm("x")
    for _ in range(40):
        if codes2:
            break
        time.sleep(0.1)
    assert codes2 == [3] and out2.getvalue() == ""
    # bound 0 = off: no thread
    assert bench.Watchdog(0, {})._thread is None


def test_watchdog_really_ends_a_hung_process():
    """End to end in a child process: the main thread blocks for good, the line appears, rc == 3."""
    import json
    code = ("import sys, time; sys.path.insert(0, %r); import bench; "
            "d = bench.Watchdog(0.5, {'metric': 'm'}); d.arm('timed step 1 of 1'); time.sleep(60)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3, (r.returncode, r.stderr)
    assert json.loads(r.stdout.strip())["hung_phase"] == "timed step 1 of 1"


def test_rccl_failure_falls_back_to_peer_copies_and_says_so():
    """`--exchange auto`: FWX_ERR_RCCL from fwx_matrix_create_multi -> one retry over peer copies, recorded;
    an explicit `--exchange rccl` and any other error are not swallowed."""
    import numpy as np
    import bench
    from floydwarshall_amd import engine
    made = []

    class Fake:
        def __init__(self, n, dtype, with_next, devices, exchange):
            made.append(exchange)
            if exchange != engine.FWX_XCHG_PEER and fail["status"] is not None:
                err = engine.FwxError.__new__(engine.FwxError)      # (no library needed for the message)
                RuntimeError.__init__(err, "fwx_matrix_create_multi: RCCL could not be loaded")
                err.status = fail["status"]
                raise err

    fail = {"status": engine.FWX_ERR_RCCL}
    h, info = bench.create_multi_handle(engine, 256, np.float32, False, [0, 1], "auto", factory=Fake)
    assert isinstance(h, Fake) and made == [engine.FWX_XCHG_AUTO, engine.FWX_XCHG_PEER]
    assert info["requested"] == "auto" and info["rccl_error"]["status"] == engine.FWX_ERR_RCCL
    assert info["rccl_error"]["where"] == "fwx_matrix_create_multi"
    del made[:]
    with pytest.raises(engine.FwxError):                                     # asked for RCCL: no fallback
        bench.create_multi_handle(engine, 256, np.float32, False, [0, 1], "rccl", factory=Fake)
    assert made == [engine.FWX_XCHG_RCCL]
    fail["status"] = -3                                                      # FWX_ERR_HIP: not ours to hide
    with pytest.raises(engine.FwxError):
        bench.create_multi_handle(engine, 256, np.float32, False, [0, 1], "auto", factory=Fake)
    fail["status"] = None
    del made[:]
    h, info = bench.create_multi_handle(engine, 256, np.float32, False, [0, 1], "auto", factory=Fake)
    assert made == [engine.FWX_XCHG_AUTO] and info["rccl_error"] is None


def test_the_ipc_mode_is_set_before_any_hip_call():
    """HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC: RCCL needs it on this pool) is a default of every binding of
    the package, not only of bench.py: importing the loader sets it, an explicit setting wins."""
    code = ("import os, sys; sys.path.insert(0, %r); os.environ.pop('HSA_ENABLE_IPC_MODE_LEGACY', None); "
            "import floydwarshall_amd._lib; print(os.environ['HSA_ENABLE_IPC_MODE_LEGACY'])" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.stdout.strip() == "0", r.stderr
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="1")
    code = ("import os, sys; sys.path.insert(0, %r); import floydwarshall_amd._lib; "
            "print(os.environ['HSA_ENABLE_IPC_MODE_LEGACY'])" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert r.stdout.strip() == "1", r.stderr


def test_a_hung_extra_leg_does_not_cost_the_measurement():
    """Once the timed region has produced `value`, a leg that hangs (the event-timed solve, the fused-engine
    leg) makes the watchdog print THAT line as it stands, flagged `extras_aborted`, and exit with code 0."""
    import io
    import json
    import time
    import bench
    out, codes = io.StringIO(), []
    dog = bench.Watchdog(0.3, {"metric": "m"}, out=out, exit_fn=codes.append)
    line = {"metric": "m", "value": 1.5e12, "n_gpus": 8, "exchange": {"transport": "rccl"}}
    dog.arm("the 2 timed steps")
    dog.disarm()
    dog.set_valid_line(line)
    line["exchange"]["avg_bulk_us"] = 1300.0          # the caller keeps filling the same dict
    dog.arm("fused-engine leg")
    for _ in range(40):
        if codes:
            break
        time.sleep(0.1)
    assert codes == [0]
    got = json.loads(out.getvalue().strip())
    assert got["value"] == 1.5e12 and got["exchange"]["avg_bulk_us"] == 1300.0 and "error" not in got
    assert got["extras_aborted"]["hung_phase"] == "fused-engine leg"
    assert got["extras_aborted"]["phases_completed"] == ["the 2 timed steps"]


def test_after_the_line_is_printed_a_hung_final_barrier_only_ends_the_process():
    """run_dist prints its line, then waits in a last barrier: if a rank is gone (its own watchdog ended it during an
    optional leg) that barrier never returns -- the watchdog then ends the process with code 0 and writes NOTHING
    more (the driver parses ONE line)."""
    import io
    import time
    import bench
    out, codes = io.StringIO(), []
    dog = bench.Watchdog(0.3, {"metric": "m"}, out=out, exit_fn=codes.append)
    dog.set_valid_line({"metric": "m", "value": 2.0e12})
    dog.line_printed()
    dog.arm("final barrier")
    for _ in range(40):
        if codes:
            break
        time.sleep(0.1)
    assert codes == [0] and out.getvalue() == ""


def test_digest_of_row_slabs_equals_the_digest_of_the_matrix():
    """run_dist digests the ranks' slabs in row order instead of concatenating a 1 GiB matrix: same digest."""
    import numpy as np
    import bench
    a = np.random.default_rng(5).random((37, 11)).astype(np.float32)
    assert bench.digest([a[:5], a[5:20], a[20:]]) == bench.digest(a)
    assert bench.digest([a[:5], a[5:20], a[20:36]]) != bench.digest(a)
