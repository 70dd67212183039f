"""-m gpu, SELF-ACTIVATING: runs only on a node with two or more GPUs (the round-end 1-GPU box skips it) — the first test in which the product's
row-tiled path meets the REAL RCCL between GPUs. bench.py is launched exactly as the driver launches it (torch.distributed.run, one rank per GPU,
RCCL over xGMI through the C ABI: grouped ncclSend / ncclRecv halos + composite, csrc/mgpu.hip) with VQ_BENCH_VERIFY=1: rank 0 recomputes the whole
frame untiled and compares it byte for byte with the composite. Every overlap mode — the shared-memory stand-in of the 1-GPU tests
(tests/cpp/mock_rccl.cpp) is synchronous, so the device-side ordering of two streams / two communicators is only ever exercised here — for the
weak-scaling cfg3 frame and the strong-scaling cfg5 frame. No reference analogue (one queue, SceneRendering.cpp:2507)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NGPU = torch.cuda.device_count() if torch.cuda.is_available() else 0
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(NGPU < 2, reason=f"needs >= 2 GPUs on the node (found {NGPU}): real RCCL between GPUs")]
WORLDS = sorted({2, NGPU} | ({4} if NGPU >= 4 else set()))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(ranks, extra_args):
    env = dict(os.environ, VQ_BENCH_VERIFY="1", VQ_BENCH_SPINUP="8", VQ_BENCH_SUSTAINED_S="0", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("VQ_BENCH_SHARE_GPU", None)
    env.pop("VQHIP_RCCL_LIBRARY", None)                      # the real librccl, not the stand-in
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "6", "--warmup", "2",
           "--no-cpu-baseline", "--no-second-mode"] + extra_args
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


def _check_real_rccl(d, ranks):
    r = d["rccl"]
    assert r["nranks_seen"] == ranks and r["rank_seen"] == 0, r
    assert r["version"] > 0 and "mock" not in r["library_path"], r          # ncclGetVersion of the real library; the stand-in reports 0


@pytest.mark.parametrize("overlap", ["off", "on", "two-comms", "auto"])
@pytest.mark.parametrize("ranks", WORLDS)
def test_cfg3_weak_scaling_on_real_gpus_matches_the_untiled_frame(ranks, overlap):
    d = _run(ranks, ["--config", "cfg3", "--composite", "root", "--composite-overlap", overlap, "--no-extras"])
    assert d["n_gpus"] == ranks and d["scaling"] == "weak" and d["config"]["frame_height"] == 2160 * ranks
    assert d["verify"]["mismatching_bytes"] == 0, d["verify"]
    _check_real_rccl(d, ranks)
    if overlap in ("on", "two-comms"):
        assert d["rccl"]["composite_overlap_mode"] == ("one-comm" if overlap == "on" else "two-comms") and not d["rccl"].get("fallback"), d["rccl"]


@pytest.mark.parametrize("composite", ["root", "all"])
def test_cfg5_strong_scaling_on_real_gpus_matches_the_untiled_frame(composite):
    ranks = NGPU
    d = _run(ranks, ["--config", "cfg5", "--composite", composite, "--no-extras"])
    assert d["n_gpus"] == ranks and d["scaling"] == "strong" and d["config"]["frame_height"] == 4320 and d["config"]["lights"] == 256
    assert d["verify"]["mismatching_bytes"] == 0, d["verify"]
    _check_real_rccl(d, ranks)


def test_default_invocation_on_every_gpu_reports_cfg5_strong_next_to_the_headline():
    """what the driver runs for SCALE_rNN.json at N = all GPUs: headline + cfg5_strong, both verified"""
    d = _run(NGPU, [])
    assert d["verify"]["mismatching_bytes"] == 0 and d["cfg5_strong"]["tile_rows"] == -(-4320 // NGPU) and d["cfg5_strong"]["value"] > 0
    _check_real_rccl(d, NGPU)
