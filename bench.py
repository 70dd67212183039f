#!/usr/bin/env python3
"""bench.py — forward-PBR hot path on MI355X (BASELINE.json metric: Mpixels/s forward-PBR @4K, 64 lights; 1/2/4/8 GPUs).

A "step" is one pass of the hot path over one synthetic frame (tile) that is already resident in HBM:
    forward lighting (point lights [+ IBL sample])   -> RGBA16F scene colour   [vqhip_forward_lighting]
    21-tap Gaussian blur X                            -> RGBA16F                 [vqhip_gaussian_blur_x]
    (N > 1) 10-row halo exchange with the neighbours                             [vqhip_exchange_blur_halos, RCCL send/recv]
    blur Y + tonemap (Reinhard + sRGB OETF)           -> RGBA8_UNORM             [vqhip_gaussian_blur_y_tonemap]
    (N > 1) composite of the RGBA8 tiles on rank 0                               [vqhip_composite_tiles, RCCL send/recv]

--config cfg3 (default; BASELINE config 3, the configuration the metric is quoted on): 3840x2160 tile per GPU, 64 point lights + the
    full-size cfg4 IBL. N > 1 is WEAK scaling: the frame is 3840 x (2160*N), one 4K tile per GPU.
--config cfg5 (BASELINE config 5): ONE 7680x4320 frame, 256 point lights (100 in the cbuffer + 156 through the extension array),
    row-tiled over the N GPUs (4320/N rows each): STRONG scaling. At N = 1 the whole 2.1 GB G-buffer is shaded by one GPU.

Every byte that crosses GPUs goes through the C ABI (include/vqhip.h, vqengine_amd/csrc/mgpu.hip); torch.distributed is the control
plane only (communicator-id broadcast, barriers, the max-over-ranks reduction of the wall time).
`value` = pixels of the whole frame * steps / max-over-ranks wall time. Prints ONE JSON line on rank 0."""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from vqengine_amd import abi, capi, synth, tiling  # noqa: E402

CONFIGS = {
    "cfg3": dict(width=3840, height=2160, lights=64, env=True, seed=0x6400, scaling="weak",
                 metric="Mpixels/s forward-PBR @4K,64 lights",
                 workload="BASELINE cfg3: 3840x2160 float4 G-buffer tile per GPU, 64 point lights + IBL sample -> RGBA16F, 21-tap blur X/Y, Reinhard+sRGB tonemap -> RGBA8"),
    "cfg5": dict(width=7680, height=4320, lights=256, env=False, seed=0x2560, scaling="strong",
                 metric="Mpixels/s forward-PBR @8K,256 lights (BASELINE cfg5, one frame row-tiled over the GPUs)",
                 workload="BASELINE cfg5: ONE 7680x4320 float4 G-buffer, 256 point lights (100 cbuffer + 156 extension) -> RGBA16F, 21-tap blur X/Y, Reinhard+sRGB tonemap -> RGBA8"),
}
HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md:35 (spec); 6290 measured copy ceiling
VALU_PEAK_TFLOPS = 157.3        # :40
SHADE_BYTES_PER_PX = 64 + 8     # 4 float4 G-buffer planes in + RGBA16F out
SPINUP_STEPS = int(os.environ.get("VQ_BENCH_SPINUP", "200"))     # untimed steady-state spin-up before the W warm-up steps (~0.25 s of GPU work)
COLD_STEPS = 20                 # the first steps after the idle set-up phase, timed on their own ("cold_start")
VALU_ISSUE_CEILING_TLIS = 66.7  # T lane-instructions/s = 133 TFLOP/s of dependent-free v_fma_f32 at steady-state clocks (scripts/ubench/valu_ceiling.hip)
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_constants.json")
PMC_SOURCES = ["vqengine_amd/csrc/shade.hip", "vqengine_amd/csrc/vq_devmath.h", "vqengine_amd/csrc/vq_sampling.h", "vqengine_amd/csrc/Makefile"]


def kernel_source_hash():
    h = hashlib.sha256()
    for f in PMC_SOURCES:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def load_pmc_constants(config, fresnel_pow):
    """Counter-derived constants of the shade kernel (HBM bytes per launch, VALU instructions per wave) cannot be measured from inside
    bench.py: they are read from profiles/pmc_constants.json, which records the sha256 of the kernel sources they were measured on
    (scripts/pmc_refresh.sh). If the sources changed since, the constants are NOT used: the fields they feed are null and `stale` is set."""
    try:
        d = json.load(open(PMC_FILE))
    except (OSError, ValueError):
        return None, {"stale": True, "why": "profiles/pmc_constants.json missing"}
    entry = d.get(f"{config}/{fresnel_pow}")
    meta = {"file": "profiles/pmc_constants.json", "kernel_sources_sha256": d.get("kernel_sources_sha256"), "measured_at_commit": d.get("measured_at_commit"),
            "profile": d.get("profile")}
    if entry is None:
        return None, dict(meta, stale=True, why=f"no entry for {config}/{fresnel_pow}")
    if d.get("kernel_sources_sha256") != kernel_source_hash():
        return None, dict(meta, stale=True, why="shade.hip / vq_devmath.h / vq_sampling.h changed since the counters were collected", now=kernel_source_hash())
    return entry, dict(meta, stale=False)


def build_ibl(ctx):
    """Load-time inputs (outside the timed region): BASELINE config 4 — 2048^2 equirect -> min-filter mips ->
    diffuse 64^2 (step 0.010) + blur + 7-mip specular 128^2, and the 1024^2 x 2048 BRDF LUT."""
    eq = torch.from_numpy(synth.equirect(2048, 2048)).cuda()
    chain, n = ctx.mip_chain(eq)
    pre = ctx.envmap_prefilter(chain, 2048, 2048, n, 64, 0.010, 128, abi.CONV_WAVE64)
    lut = ctx.brdf_lut(1024, 2048, abi.FMT_RG16F)
    torch.cuda.synchronize()
    return pre, lut


def upload_tile(cfg, frame_h, row0, row1):
    W = cfg["width"]
    gb = [torch.empty((row1 - row0, W, 4), dtype=torch.float32, device="cuda") for _ in range(4)]
    for r in range(row0, row1, 240):
        part = synth.gbuffer_rows(W, frame_h, r, min(r + 240, row1), seed=cfg["seed"])
        for k in range(4):
            gb[k][r - row0:r - row0 + part[k].shape[0]].copy_(torch.from_numpy(part[k]))
    return gb


def host_cores():
    cores = len(os.sched_getaffinity(0))
    try:                                                     # honour a cgroup CPU quota if the box has one
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(float(q) / float(p))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(cfg, env_np, pf, extra, pv, frame_h, target_s=10.0):
    """The CPU oracle (a scalar C++ port of the HLSL, OpenMP over rows) timed on the host cores on a bounded row
    band of the SAME workload. Reported baseline only — never the thing measured as `value`."""
    from tests import oracle_lib as O
    O.load()
    cores = host_cores()
    W = cfg["width"]
    env = O.host_envmap(*env_np) if env_np is not None else None
    band = 540 if cfg["lights"] <= 64 else 135               # bands of the SAME synthetic frame

    def run(row0, rows):
        gb = synth.gbuffer_rows(W, frame_h, row0, row0 + rows, seed=cfg["seed"])
        t0 = time.perf_counter()
        sc = O.forward_lighting(gb, pf, pv, abi.FMT_RGBA16F, extra_point=extra, env=env, nthreads=cores)
        x = O.blur_pass(sc, abi.FMT_RGBA16F, 0, nthreads=cores)
        y = O.blur_pass(x, abi.FMT_RGBA16F, 1, nthreads=cores)
        O.tonemap(y, abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM, nthreads=cores)
        return time.perf_counter() - t0
    run(0, 64)                                               # warm-up (thread pool, page faults)
    t, rows, k = 0.0, 0, 0
    while t < target_s and k < 64:
        t += run((k % 4) * band, band)
        rows += band
        k += 1
    return {"value": round(W * rows / t / 1e6, 4), "unit": "Mpix/s", "cores": int(cores), "kind": "port",
            "sample": f"oracle (scalar C++ port of the HLSL, OpenMP static over rows, {cores} threads) on {k} bands of {W}x{band} rows of the same "
                      f"frame ({W * rows / 1e6:.1f} Mpix): shade {cfg['lights']} lights{' + IBL' if env is not None else ''}, blur X/Y, tonemap; {t:.1f} s"}


def cpu_reference_source(cfg, env_np, pf, extra, pv, frame_h, target_s=6.0):
    """The REFERENCE'S OWN shader source (ForwardLighting.hlsl:PSMain, GaussianBlur.hlsl, Tonemapper.hlsl) run on one host core
    through oracle/_ref (oracle/ref_src/hlsl_shim.h) on rows of the same frame, when that library travelled with the tree. A second
    reported baseline next to `cpu_baseline`: scalar, single-threaded (the translated shaders keep their globals), literal IEEE."""
    from tests import oracle_lib as O, ref_lib as R
    if not R.available("shaders") or (extra is not None and not R.available("shaders_l256")):
        return None
    W = cfg["width"]
    env = O.host_envmap(*env_np) if env_np is not None else None
    rows_per = 22 if cfg["lights"] <= 64 else 4              # one blur kernel height: the band is a (small) image of its own
    t, rows, k = 0.0, 0, 0
    while t < target_s and k < 64:
        r0 = (k * 97) % (frame_h - rows_per)
        gb = synth.gbuffer_rows(W, frame_h, r0, r0 + rows_per, seed=cfg["seed"])
        t0 = time.perf_counter()
        sc = R.forward_from_gbuffer(gb, pf, pv, env=env, extra=extra).astype(np.float16).astype(np.float32)
        x = R.blur_pass(sc, 0).astype(np.float16).astype(np.float32)
        y = R.blur_pass(x, 1).astype(np.float16).astype(np.float32)
        R.tonemap(y, abi.TonemapperParams.default())
        t += time.perf_counter() - t0
        rows += rows_per
        k += 1
    return {"value": round(W * rows / t / 1e6, 4), "unit": "Mpix/s", "cores": 1, "kind": "reference",
            "sample": f"the reference's HLSL (PSMain {cfg['lights']} lights{' + IBL' if env is not None else ''}, CSMain_X/_Y, tonemapper CSMain) compiled to C++ through "
                      f"oracle/ref_src/hlsl_shim.h, 1 thread, {k} bands of {W}x{rows_per} rows of the same frame ({W * rows / 1e6:.2f} Mpix); {t:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)      # a step is ~1 ms: 0.2 s of timed GPU work; the run stays dominated by set-up and the CPU baselines
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--post", choices=["fused", "split"], default="fused",
                    help="post chain after the X blur: Y blur and tonemapper as two dispatches (split) or one kernel (fused); identical bits")
    ap.add_argument("--composite", choices=["root", "all"], default="root",
                    help="final composite of the RGBA8 tiles: on rank 0 only (the presenting GPU; it receives over its N-1 direct xGMI links) or on every rank")
    ap.add_argument("--composite-overlap", choices=["on", "off"], default="on",
                    help="on: the composite of frame n runs on a second stream / second communicator and overlaps the shading of frame n+1 (drained "
                         "inside the timed region); off: everything in one stream order")
    ap.add_argument("--fresnel-pow", choices=["product", "exp2_log2"], default="product",
                    help="pow(1 - cos, 5) of the Fresnel terms: the product x*((x*x)*(x*x)) (default, contract v4) or exp2(5*log2 x), the engine's own "
                         "DXC lowering (vqhip_set_fresnel_pow; DESIGN.md 3.2). The other mode is timed too and reported as `engine_lowering`.")
    ap.add_argument("--no-second-mode", action="store_true", help="skip the timing of the other Fresnel-pow mode")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # VQ_BENCH_SHARE_GPU=1 (debug aid for single-GPU boxes): every rank uses the visible GPUs round-robin, the control plane runs over
    # gloo and the C ABI's RCCL calls are served by tests/cpp/libmock_rccl.so (shared memory), so that the N > 1 control flow — tiles,
    # halo exchange, double-buffered composite, drain — can be exercised on real kernels without N GPUs. Never set by the driver.
    share = os.environ.get("VQ_BENCH_SHARE_GPU") == "1"
    device_ordinal = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(device_ordinal)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            os.environ["VQHIP_RCCL_LIBRARY"] = os.path.join(ROOT, "tests", "cpp", "libmock_rccl.so")
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_ordinal))
    ctx = capi.Context(device_ordinal)
    ctx.set_fresnel_pow(args.fresnel_pow == "exp2_log2")

    W, L = cfg["width"], cfg["lights"]
    frame_h = cfg["height"] * world if cfg["scaling"] == "weak" else cfg["height"]
    tl = tiling.RowTiling(W, frame_h, world, rank)
    rows = tl.tile_rows
    env = pre = lut = None
    if cfg["env"]:
        pre, lut = build_ibl(ctx)
        env = capi.make_envmap(pre["diffuse_blurred"], pre["specular"], 128, pre["spec_mips"], lut)
    pf, extra = synth.per_frame(points=synth.point_lights(L, seed=cfg["seed"]), hdri_offset=0.3 if cfg["env"] else 0.0)
    pv = synth.per_view(W, frame_h, max_env_lod=pre["spec_mips"] if pre else 0)
    gb = upload_tile(cfg, frame_h, tl.row0, tl.row1)

    # communicators of the data path (C ABI): one for the halo exchange (critical path, main stream), one for the composite
    comm_halo = comm_comp = None
    if world > 1:
        ids = [capi.comm_unique_id(), capi.comm_unique_id()] if rank == 0 else [None, None]
        dist.broadcast_object_list(ids, src=0)               # control plane: 2 x 128 bytes
        comm_halo = capi.Comm(ids[0], world, rank)
        comm_comp = capi.Comm(ids[1], world, rank)
        # wiring check before anything is timed: every rank sends rows that name their sender and their row, so a transfer that lands in the
        # wrong place, comes from the wrong neighbour or does not arrive at all is a loud failure here instead of a wrong frame later
        chk_rows = tiling.RowTiling(64, 16 * world + (world // 2), world, rank)          # uneven tiles on purpose
        t_x = torch.empty((chk_rows.tile_rows, 64, 4), dtype=torch.float16, device=ctx.device)
        t_x[:] = (torch.arange(chk_rows.row0, chk_rows.row1, device=ctx.device, dtype=torch.float32) + 1000.0 * rank).to(torch.float16)[:, None, None]
        h_t = torch.zeros((capi.HALO_ROWS, 64, 4), dtype=torch.float16, device=ctx.device) if rank > 0 else None
        h_b = torch.zeros((capi.HALO_ROWS, 64, 4), dtype=torch.float16, device=ctx.device) if rank < world - 1 else None
        comm_halo.exchange_blur_halos(t_x, abi.FMT_RGBA16F, h_t, h_b, stream=C.c_void_p(torch.cuda.current_stream(ctx.device).cuda_stream))
        t_c = torch.full((chk_rows.tile_rows, 64, 4), rank + 1, dtype=torch.uint8, device=ctx.device)
        f_c = torch.zeros((chk_rows.frame_height, 64, 4), dtype=torch.uint8, device=ctx.device)
        comm_comp.composite_tiles(t_c, abi.FMT_RGBA8_UNORM, chk_rows.frame_height, capi.ALL_RANKS, f_c, stream=C.c_void_p(torch.cuda.current_stream(ctx.device).cuda_stream))
        torch.cuda.synchronize()
        if h_t is not None:
            want = (torch.arange(chk_rows.row0 - capi.HALO_ROWS, chk_rows.row0, device=ctx.device, dtype=torch.float32) + 1000.0 * (rank - 1)).to(torch.float16)
            assert torch.equal(h_t[:, 0, 0], want), f"rank {rank}: top halo rows are not the last 10 rows of rank {rank - 1}"
        if h_b is not None:
            want = (torch.arange(chk_rows.row1, chk_rows.row1 + capi.HALO_ROWS, device=ctx.device, dtype=torch.float32) + 1000.0 * (rank + 1)).to(torch.float16)
            assert torch.equal(h_b[:, 0, 0], want), f"rank {rank}: bottom halo rows are not the first 10 rows of rank {rank + 1}"
        for k in range(world):
            r0, n = capi.rowtile(chk_rows.frame_height, world, k)
            assert bool((f_c[r0:r0 + n] == k + 1).all()), f"rank {rank}: rows of rank {k} are wrong in the composited frame"
        del t_x, h_t, h_b, t_c, f_c

    F16, R8 = abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM
    scene = [capi.empty_image(rows, W, F16, ctx.device) for _ in range(2)]
    xblur = capi.empty_image(rows, W, F16, ctx.device)
    yblur = capi.empty_image(rows, W, F16, ctx.device) if args.post == "split" else None
    sdr = [capi.empty_image(rows, W, R8, ctx.device) for _ in range(2)]
    root = 0 if args.composite == "root" else capi.ALL_RANKS
    need_frame = world > 1 and (root == capi.ALL_RANKS or rank == 0)
    frame = [torch.empty((frame_h, W, 4), dtype=torch.uint8, device=ctx.device) for _ in range(2)] if need_frame else [None, None]
    halo_top = capi.empty_image(capi.HALO_ROWS, W, F16, ctx.device) if world > 1 and rank > 0 else None
    halo_bottom = capi.empty_image(capi.HALO_ROWS, W, F16, ctx.device) if world > 1 and rank < world - 1 else None
    overlap = world > 1 and args.composite_overlap == "on"
    s_main = torch.cuda.current_stream(ctx.device)
    s_comp = torch.cuda.Stream(ctx.device) if overlap else s_main
    h_main, h_comp = C.c_void_p(s_main.cuda_stream), C.c_void_p(s_comp.cuda_stream)
    e_post = [torch.cuda.Event(), torch.cuda.Event()]
    e_comp = [None, None]

    def step(i, ev=None):
        b = i & 1
        if overlap and e_comp[b] is not None:        # sdr[b] / frame[b] were last touched by the composite of step i-2
            s_main.wait_event(e_comp[b])
        if ev:
            ev[0].record(s_main)
        ctx.forward_lighting(gb, pf, pv, out=scene[b], out_fmt=F16, extra_point=extra, env=env)
        if ev:
            ev[1].record(s_main)
        if ev and len(ev) == 5:
            ev[4].record(s_main)
        ctx.gaussian_blur_x(scene[b], F16, out=xblur)
        if world > 1:
            comm_halo.exchange_blur_halos(xblur, F16, halo_top, halo_bottom, stream=h_main)
        if args.post == "fused":
            # CSMain_Y + Tonemapper in one kernel (register-window Y pass whose store goes through the 64 KB tonemap table in
            # LDS): bit-identical to the two dispatches, BlurOutput never touches HBM. ev[2] then closes the X pass (+ halo exchange).
            if ev and len(ev) == 5:
                ev[2].record(s_main)
            ctx.gaussian_blur_y_tonemap(xblur, F16, R8, out=sdr[b], halo_top=halo_top, halo_bottom=halo_bottom)
        else:
            ctx.gaussian_blur_y(xblur, F16, out=yblur, halo_top=halo_top, halo_bottom=halo_bottom)
            if ev and len(ev) == 5:
                ev[2].record(s_main)
            ctx.tonemap(yblur, F16, R8, out=sdr[b])
        if ev and len(ev) >= 4:
            ev[3].record(s_main)
        if world > 1:
            if overlap:
                e_post[b].record(s_main)
                s_comp.wait_event(e_post[b])
            comm_comp.composite_tiles(sdr[b], R8, frame_h, root, frame[b], stream=h_comp)
            if overlap:
                e_comp[b] = torch.cuda.Event()
                e_comp[b].record(s_comp)

    def drain():
        if overlap:
            s_main.wait_stream(s_comp)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_steps, evs=None, first=0):
        barrier()
        t0 = time.perf_counter()
        for i in range(n_steps):
            step(first + i, evs[i] if evs else None)
        drain()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=ctx.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # 1. cold start: the very first steps after the idle set-up phase, timed on their own (the chip has not ramped its clocks yet)
    dt_cold = timed(COLD_STEPS)
    # 2. untimed spin-up: the chip needs ~0.2-0.3 s of sustained load to reach its steady-state clocks, so a fixed number of extra
    #    untimed steps precedes the W warm-up steps whatever W is. The timed region is still exactly K steps.
    for i in range(max(0, SPINUP_STEPS - COLD_STEPS)):
        step(i)
    drain()
    # 2b. the post kernels on their own, after the spin-up and before the warm-up: 20 back-to-back launches of each between two events, on the
    #     frame's own buffers (no event, no other kernel in between). In the frame loop each of them follows a kernel that has just filled the
    #     caches with other data, and the per-stage events sit inside the intervals they measure; both figures are reported. (Taken here rather
    #     than after the timed region: after ~0.4 s of sustained shading a burst of X passes runs 2.5-3x slower — the chip's power limiter,
    #     profiles/r2k_frame_loop.md — which says nothing about the kernel.)
    iso = None
    if args.post == "fused":
        iso = {}
        for name, fn in (("blur_x", lambda: ctx.gaussian_blur_x(scene[0], F16, out=xblur)),
                         ("blur_y_tonemap", lambda: ctx.gaussian_blur_y_tonemap(xblur, F16, R8, out=sdr[0], halo_top=halo_top, halo_bottom=halo_bottom))):
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s_main)
            for _ in range(20):
                fn()
            e1.record(s_main)
            e1.synchronize()
            iso[name] = e0.elapsed_time(e1) / 20 * 1e-3
        drain()
        barrier()
    for i in range(args.warmup):
        step(i)
    drain()
    # 3. timed region: only the dominant kernel is bracketed by HIP events (2 records per step); the per-stage timings of the
    #    HBM-bound post kernels are taken in a separate, untimed pass afterwards so their instrumentation does not sit in `value`
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(args.steps)]
    dt = timed(args.steps, evs)

    n_detail = min(args.steps, 10)
    evd = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(n_detail)]
    for i in range(n_detail):
        step(args.steps + i + (args.steps & 1), evd[i])
    drain()
    barrier()
    # the chain as a whole, without an event between its kernels: [1] after shade ... [3] after the last post kernel
    evc = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n_detail)]
    for i in range(n_detail):
        step(args.steps + i + (args.steps & 1), evc[i])
    drain()
    barrier()
    # 4. frame latency: one step at a time, nothing in flight before or after it (the throughput figure pipelines the composite)
    lat = []
    for i in range(5):
        barrier()
        t0 = time.perf_counter()
        step(i)
        drain()
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
    lat_t = torch.tensor([float(np.median(lat))], dtype=torch.float64, device=ctx.device)
    if world > 1:
        dist.all_reduce(lat_t, op=dist.ReduceOp.MAX)
    frame_latency = float(lat_t.item())

    verify = None
    if world > 1 and os.environ.get("VQ_BENCH_VERIFY") == "1":
        # debug aid: rank 0 recomputes the WHOLE frame on its own GPU (no tiles, no halos) and compares it byte for byte with the
        # composite of the last step — the row tiling + halo exchange + composite on real kernels (tests/test_gpu_bench_flow.py)
        torch.cuda.synchronize()
        last = 4 & 1
        if rank == 0:
            gb_full = upload_tile(cfg, frame_h, 0, frame_h)
            sc = ctx.forward_lighting(gb_full, pf, pv, out_fmt=F16, extra_point=extra, env=env)
            xb = ctx.gaussian_blur_x(sc, F16)
            want = ctx.gaussian_blur_y_tonemap(xb, F16, R8)
            torch.cuda.synchronize()
            verify = {"mismatching_bytes": int((want != frame[last]).sum().item()), "frame": [W, frame_h]}
            del gb_full, sc, xb, want
        dist.barrier()

    # 5. the other Fresnel-pow lowering, same invocation, same clocks: `engine_lowering` is the engine-faithful exp2(5*log2 x) form
    second = None
    if not args.no_second_mode:
        other = "exp2_log2" if args.fresnel_pow == "product" else "product"
        ctx.set_fresnel_pow(other == "exp2_log2")
        for i in range(20):
            step(i)
        drain()
        evs2 = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(args.steps)]
        dt2 = timed(args.steps, evs2)
        ctx.set_fresnel_pow(args.fresnel_pow == "exp2_log2")
        second = {"fresnel_pow": other, "value": round(W * frame_h * args.steps / dt2 / 1e6, 2), "unit": "Mpix/s", "ms_per_step": round(dt2 / args.steps * 1e3, 4),
                  "shade_ms": round(float(np.mean([e[0].elapsed_time(e[1]) for e in evs2])), 4)}

    if rank == 0:
        px_tile, px_frame = W * rows, W * frame_h
        t_shade = float(np.mean([e[0].elapsed_time(e[1]) for e in evs])) * 1e-3
        t_blur = float(np.mean([e[4].elapsed_time(e[2]) for e in evd])) * 1e-3
        t_tm = float(np.mean([e[2].elapsed_time(e[3]) for e in evd])) * 1e-3
        t_chain = float(np.mean([e[1].elapsed_time(e[3]) for e in evc])) * 1e-3
        ach = SHADE_BYTES_PER_PX * px_tile / t_shade / 1e9
        flops_px = 170 * L + 160
        pmc, pmc_meta = load_pmc_constants(args.config, args.fresnel_pow)
        if pmc_meta.get("stale"):
            print(f"bench.py: PMC constants not used: {pmc_meta.get('why')}", file=sys.stderr)
        out = {
            "metric": cfg["metric"], "value": round(px_frame * args.steps / dt / 1e6, 2), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": cfg["scaling"], "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["workload"] + ("" if world == 1 else f"; frame {W}x{frame_h} row-tiled {rows} rows per GPU, RCCL p2p halo exchange + composite on "
                                                       f"{'rank 0' if root == 0 else 'every rank'} through the C ABI"),
                       "name": args.config, "width": W, "frame_height": frame_h, "tile_rows": rows, "lights": L, "parallelism": f"rows{world}",
                       "composite_overlap": overlap, "untimed_spinup_steps": SPINUP_STEPS, "fresnel_pow": args.fresnel_pow,
                       "post": "blur X, then blur Y + tonemap in one kernel (identical bits to three dispatches)" if args.post == "fused" else "blur X, blur Y, tonemap"},
            "roofline": {"bound": "hbm", "kernel": f"k_forward_lighting<{'env' if env is not None else 'noenv'},nocasters,RGBA16F>", "achieved": round(ach, 2),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5),
                         "traffic": pmc["hbm_bytes_per_launch"] if pmc else None, "traffic_unit": "bytes/launch",
                         "traffic_source": "rocprofv3 PMC, separate passes, 2*FETCH_SIZE + WRITE_SIZE (profiles/pmc_constants.json); algorithmic = %d" % (SHADE_BYTES_PER_PX * px_tile),
                         "bytes_per_px": SHADE_BYTES_PER_PX, "ms": round(t_shade * 1e3, 4),
                         "note": f"{L}-light shading is VALU-bound by construction (SURVEY.md 8d): see valu / valu_issue"},
            "valu": {"achieved_tflops_model": round(flops_px * px_tile / t_shade / 1e12, 2), "peak": VALU_PEAK_TFLOPS,
                     "frac": round(flops_px * px_tile / t_shade / 1e12 / VALU_PEAK_TFLOPS, 4), "flops_per_px_model": flops_px},
            "pmc_constants": pmc_meta,
            "stages": {"shade_Mpix_s": round(px_tile / t_shade / 1e6, 1), "shade_ms": round(t_shade * 1e3, 4),
                       **({"blur_xy_ms": round(t_blur * 1e3, 4), "blur_xy_GBps": round(px_tile * 32 / t_blur / 1e9, 1),
                           "tonemap_ms": round(t_tm * 1e3, 4), "tonemap_GBps": round(px_tile * 12 / t_tm / 1e9, 1)} if args.post == "split" else
                          {"blur_x_ms": round(t_blur * 1e3, 4), "blur_x_GBps": round(px_tile * 16 / t_blur / 1e9, 1),
                           "blur_y_tonemap_ms": round(t_tm * 1e3, 4), "blur_y_tonemap_GBps": round(px_tile * 12 / t_tm / 1e9, 1),
                           "post_chain_ms": round(t_chain * 1e3, 4), "post_chain_GBps": round(px_tile * 28 / t_chain / 1e9, 1),
                           "post_chain_frac_of_hbm_peak": round(px_tile * 28 / t_chain / 1e9 / HBM_PEAK_GBPS, 4),
                           "post_chain_note": "from the end of the shade kernel to the end of the last post kernel inside the frame loop, no event in between"
                                              " (blur_x_ms / blur_y_tonemap_ms above come from steps that record one)",
                           "isolated": {"blur_x_ms": round(iso["blur_x"] * 1e3, 4), "blur_x_GBps": round(px_tile * 16 / iso["blur_x"] / 1e9, 1),
                                        "blur_x_frac_of_hbm_peak": round(px_tile * 16 / iso["blur_x"] / 1e9 / HBM_PEAK_GBPS, 4),
                                        "blur_y_tonemap_ms": round(iso["blur_y_tonemap"] * 1e3, 4),
                                        "blur_y_tonemap_GBps": round(px_tile * 12 / iso["blur_y_tonemap"] / 1e9, 1),
                                        "blur_y_tonemap_frac_of_hbm_peak": round(px_tile * 12 / iso["blur_y_tonemap"] / 1e9 / HBM_PEAK_GBPS, 4),
                                        "note": "20 back-to-back launches of the one kernel between two events, after the spin-up and before the warm-up steps; "
                                                "the figures above are taken inside the frame loop with an event record between the stages"}}),
                       **({"blur_x_includes": "halo exchange"} if world > 1 else {})},
            "frame_latency_ms": round(frame_latency * 1e3, 4),
            "cold_start": {"steps": COLD_STEPS, "ms_per_step": round(dt_cold / COLD_STEPS * 1e3, 4), "value": round(px_frame * COLD_STEPS / dt_cold / 1e6, 2),
                           "note": "the first steps after the idle set-up phase, before the clocks ramp; `value` is the steady-state figure"},
        }
        if pmc:
            vw, tw = pmc["valu_instr_per_wave"], pmc.get("quarter_rate_instr_per_wave", 0)
            waves = px_tile / 64.0
            out["valu_issue"] = {"achieved_T_lane_instr_s": round(vw * 64 * waves / t_shade / 1e12, 2), "ceiling": VALU_ISSUE_CEILING_TLIS,
                                 "frac": round(vw * 64 * waves / t_shade / 1e12 / VALU_ISSUE_CEILING_TLIS, 4),
                                 "frac_slot_weighted": round((vw + 3 * tw) * 64 * waves / t_shade / 1e12 / VALU_ISSUE_CEILING_TLIS, 4),
                                 "valu_instr_per_wave": vw, "quarter_rate_instr_per_wave": tw,
                                 "note": "the binding roof: VALU instructions issued per second (PMC count x live kernel time) vs the steady-state v_fma_f32 "
                                         "issue rate of the chip (scripts/ubench/valu_ceiling.hip); slot-weighted counts each quarter-rate v_rcp/v_rsq as 4 slots"}
        if second is not None:
            out["engine_lowering" if second["fresnel_pow"] == "exp2_log2" else "product_lowering"] = second
        if verify is not None:
            out["verify"] = verify
        if world == 1 and not args.no_cpu_baseline:
            env_np = (pre["diffuse_blurred"].cpu().numpy(), pre["specular"].cpu().numpy(), 128, pre["spec_mips"], lut.cpu().numpy()) if pre else None
            if args.fresnel_pow == "exp2_log2":
                from tests import oracle_lib as _O
                _O.load().vqo_set_fresnel_pow(1)             # keeps the cpu_baseline leg on the same arithmetic
            out["cpu_baseline"] = cpu_baseline(cfg, env_np, pf, extra, pv, frame_h)
            try:                                             # optional second baseline: only where oracle/_ref exists
                ref_line = cpu_reference_source(cfg, env_np, pf, extra, pv, frame_h)
                if ref_line is not None:
                    out["cpu_reference_source"] = ref_line
            except Exception as e:                           # never let the optional leg break the bench line
                out["cpu_reference_source"] = {"error": repr(e)[:200]}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        comm_halo.close()
        comm_comp.close()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
