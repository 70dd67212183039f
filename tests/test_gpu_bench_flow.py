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


def test_single_gpu_line_carries_the_contract_fields():
    """`python bench.py` (N = 1, defaults shortened): one JSON line with the driver's contract fields, the roofline and cpu_baseline objects,
    counter constants that belong to the current kernel sources, and the self-audit extras (engine lowering, cold start, isolated post kernels)."""
    env = dict(os.environ, VQ_BENCH_SPINUP="30")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line on stdout"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "stages", "engine_lowering", "cold_start", "frame_latency_ms", "valu_issue", "pmc_constants"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "Mpix/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 3840 * 2160 / (d["ms_per_step"] * 1e-3) / 1e6) < 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert isinstance(r["traffic"], int) and r["traffic"] > 597196800 // 2
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mpix/s" and c["sample"]
    assert d["pmc_constants"]["stale"] is False
    assert d["engine_lowering"]["fresnel_pow"] == "exp2_log2" and 0 < d["engine_lowering"]["value"] < 1.05 * d["value"]
    iso = d["stages"]["isolated"]
    assert 0 < iso["blur_x_ms"] < 0.2 and 0 < iso["blur_y_tonemap_ms"] < 0.2
    assert d["stages"]["shade_ms"] < d["ms_per_step"]
