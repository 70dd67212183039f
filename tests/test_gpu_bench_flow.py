"""bench.py's N > 1 control flow on REAL kernels, on a one-GPU box: two ranks share the GPU (VQ_BENCH_SHARE_GPU=1, collectives over
gloo) and rank 0 recomputes the whole 3840 x 4320 frame untiled and compares it byte for byte with the composite (VQ_BENCH_VERIFY=1).
RCCL itself is not exercised here (one GPU); the tiling / halo / composite logic, the double buffering and the drain are."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("ranks,halo,composite", [(2, "p2p", "gather"), (2, "allgather", "allgather"), (3, "p2p", "gather")])
def test_multi_rank_bench_flow_matches_the_untiled_frame(ranks, halo, composite):
    """3 ranks: the middle tile exchanges halos with both neighbours."""
    env = dict(os.environ, VQ_BENCH_SHARE_GPU="1", VQ_BENCH_VERIFY="1", VQ_BENCH_SPINUP="4", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "5", "--warmup", "2",
           "--no-cpu-baseline", "--halo", halo, "--composite", composite]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == ranks and d["scaling"] == "weak" and d["config"]["frame_height"] == 2160 * ranks
    assert d["verify"]["mismatching_bytes"] == 0, d["verify"]
