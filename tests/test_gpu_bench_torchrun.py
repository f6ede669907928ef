"""`bench.py --gpus 2` the way the driver starts it for N > 1 -- `python -m torch.distributed.run --nproc-per-node 2
... bench.py --gpus 2` -- rehearsed on the one GPU there is: both ranks on device 0, the process group on gloo
(RCCL refuses two ranks on one device; `--backend gloo` marks the line INVALID_rehearsal_backend).  Everything else
is the production path: one partition per process behind the C ABI (fwx_matrix_create_part), panels through the
exchange callback, the barrier + max-over-ranks timing, the extra legs, and the result check -- the ranks' slabs
gathered on rank 0 and digested -- which must equal the digest the single-GPU line reports for the same input."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("with_next", [False, True])
def test_torchrun_two_ranks_one_line_and_the_gathered_result(with_next):
    common = ["--size", "768", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--step-timeout", "120"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    single = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, cwd=ROOT, env=env,
                            capture_output=True, text=True, timeout=300)
    assert single.returncode == 0, single.stderr[-2000:]
    one = _line(single.stdout)
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo"] + common +
                         (["--with-next"] if with_next else []),
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-3000:]
    d = _line(two.stdout)                                   # ONE line, from rank 0
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["driver"]["name"] == "part", d["driver"]
    assert d["INVALID_rehearsal_backend"] == "gloo"
    assert d["exchange"]["ranks_in_process_group"] == 2 and d["exchange"]["steps"] >= 1
    assert d["fused_engine"]["value"] > 0
    assert d["legacy_driver"]["equals_timed_driver_bits"] is True
    # the two ranks' slabs, gathered: the same matrix the single-GPU run left in HBM
    # (its default line carries the rates-only digest and, from its next-hop leg, the next-hop digest)
    assert d["check"]["rate_digest"] == one["check"]["rate_digest"], (d["check"], one["check"])
    if with_next:
        assert d["check"]["next_digest"] == one["fused_engine_next"]["check"]["next_digest"]
