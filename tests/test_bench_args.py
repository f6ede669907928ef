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
