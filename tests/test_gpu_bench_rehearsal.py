"""GPU: bench.py's multi-rank path rehearsed ONCE before a driver runs it on an 8-GPU node.

`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` with FA2_BENCH_BACKEND=gloo: two ranks share
the one GPU of the test box (the launcher itself never touches the GPU, every rank is an ordinary child process), so the
branches only world > 1 reaches -- process-group init, the barriers around the timed region, the all-reduce MAX of the
elapsed time, `value` = all ranks' flops / the slowest rank's time, rank 0 printing the one JSON line, the shutdown --
execute for real.  What this cannot rehearse: the nccl (RCCL) backend itself and the ring leg (RCCL refuses two ranks on
one device): those stay "unmeasured on hardware" (DESIGN.md section 4).  No scaling figure is derived from this run."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_bench_line_over_gloo():
    env = dict(os.environ, FA2_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-ring", "--sustained-steps", "24"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE line; rank 1 prints none
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["metric"].startswith("FA2 fwd+bwd TFLOP/s at (B=4,H=16,N=8192,d=128)")
    # value = 2 ranks' flops / the slower rank's time
    flops = 14.0 * 4 * 16 * 8192 * 8192 * 128
    assert abs(out["value"] - 2 * flops / (out["ms_per_step"] * 1e-3) / 1e12) <= 0.01 * out["value"]
    assert out["roofline"]["kernel"] and "cpu_baseline" not in out and "ring" not in out
    # the clock over the window: taken per XCC (round 3 read "225 MHz" here: its two one-workgroup samples had landed on
    # different XCCs, whose s_memtime counters have different origins, once a second process moved the dispatcher's rotation)
    assert out["sustained"]["steps"] == 24 and 500.0 < out["sustained"]["mean_shader_clock_mhz"] < 2600.0
