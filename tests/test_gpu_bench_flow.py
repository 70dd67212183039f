"""bench.py's N > 1 control flow on REAL kernels, on a one-GPU box: the ranks share the GPU (VQ_BENCH_SHARE_GPU=1: control plane over gloo,
the C ABI's RCCL calls served by tests/cpp/libmock_rccl.so through shared memory) and rank 0 recomputes the whole frame untiled and compares
it byte for byte with the composite (VQ_BENCH_VERIFY=1). RCCL itself is not exercised here (one GPU); everything the product adds around it
is: vqhip_rowtile, the halo exchange and the composite through the C ABI, the second stream / second communicator, the double buffering
and the drain — for the weak-scaling cfg3 frame and for the strong-scaling cfg5 frame (7680x4320, 256 lights, uneven 1440-row tiles)."""
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


def _run(ranks, extra_args):
    env = dict(os.environ, VQ_BENCH_SHARE_GPU="1", VQ_BENCH_VERIFY="1", VQ_BENCH_SPINUP="4", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "4", "--warmup", "1",
           "--no-cpu-baseline", "--no-second-mode"] + extra_args
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.parametrize("ranks,composite,overlap", [(2, "root", "on"), (2, "all", "off"), (3, "root", "on")])
def test_cfg3_weak_scaling_flow_matches_the_untiled_frame(ranks, composite, overlap):
    """3 ranks: the middle tile exchanges halos with both neighbours."""
    d = _run(ranks, ["--config", "cfg3", "--composite", composite, "--composite-overlap", overlap])
    assert d["n_gpus"] == ranks and d["scaling"] == "weak" and d["config"]["frame_height"] == 2160 * ranks
    assert d["verify"]["mismatching_bytes"] == 0, d["verify"]


def test_cfg5_strong_scaling_flow_matches_the_untiled_frame():
    d = _run(3, ["--config", "cfg5"])
    assert d["n_gpus"] == 3 and d["scaling"] == "strong" and d["config"]["frame_height"] == 4320 and d["config"]["tile_rows"] == 1440
    assert d["config"]["lights"] == 256 and d["verify"]["mismatching_bytes"] == 0, d["verify"]
