#!/usr/bin/env python3
"""bench.py — forward-PBR hot path on MI355X (BASELINE.json metric: Mpixels/s forward-PBR @4K, 64 lights).

A "step" is one pass of BASELINE config 3 over one synthetic frame tile that is already resident in HBM:
    forward lighting (64 point lights + IBL sample)  -> RGBA16F scene colour   [vqhip_forward_lighting]
    21-tap Gaussian blur X, Y                        -> RGBA16F                 [vqhip_gaussian_blur_x/_y]
    tonemap (Reinhard + sRGB OETF)                   -> RGBA8_UNORM             [vqhip_tonemap]
N = 1: one 3840x2160 frame. N > 1 (weak scaling): the frame is 3840 x (2160*N), row-tiled one tile per GPU, with
the RCCL halo exchange before the Y blur and the composite of the RGBA8 tiles (gather on rank 0, or --composite allgather)
inside the timed region.
`value` = pixels of the whole frame / max-over-ranks wall time. Prints ONE JSON line on rank 0."""
import argparse
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

W, TILE_H, N_LIGHTS = 3840, 2160, 64
HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md:35 (spec); 6290 measured copy ceiling
VALU_PEAK_TFLOPS = 157.3        # :40
SHADE_BYTES_PER_PX = 64 + 8     # 4 float4 G-buffer planes in + RGBA16F out
SHADE_FLOPS_PER_PX = 170 * N_LIGHTS + 160
# HBM bytes per launch of the shade kernel from the PMC counters (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes,
# FETCH doubled as MI355X_MICROARCH.md §HBM prescribes for gfx950): profiles/r1l_pmc_hbm.md (calibration: r1b_pmc_hbm.md). Not measurable from inside
# bench.py; the committed figure is for exactly this workload (3840x2160, 64 lights + IBL, RGBA16F out).
SHADE_PMC_TRAFFIC_BYTES = (2 * 598175 + 64800) * 1024   # re-measured on the round's final kernel: profiles/r1l_pmc_hbm.md
# VALU instructions per wave of the same kernel from `rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES` (scripts/pmc_shade.sh, committed
# summary profiles/r1e_valu_issue_rates.md + r1g note) and the measured issue ceiling of the chip (v_fma_f32 ubench).
SPINUP_STEPS = int(os.environ.get("VQ_BENCH_SPINUP", "200"))              # untimed steady-state spin-up before the W warm-up steps (~0.28 s of GPU work)
SHADE_PMC_VALU_PER_WAVE = 5149
SHADE_TRANS_PER_WAVE = 273      # quarter-rate v_rcp_f32 / v_rsq_f32 per wave: 5 per executed light (81.7 % of 64) + ~12 in set-up / IBL
VALU_ISSUE_CEILING_TLIS = 66.7  # T lane-instructions/s = 133 TFLOP/s of dependent-free v_fma_f32 at steady-state clocks (scripts/ubench/valu_ceiling.hip;
                                # the single-shot figure of round 1a-1e, 52.7, was taken on cold clocks)


def build_ibl(ctx):
    """Load-time inputs (outside the timed region): BASELINE config 4 — 2048^2 equirect -> min-filter mips ->
    diffuse 64^2 (step 0.010) + blur + 7-mip specular 128^2, and the 1024^2 x 2048 BRDF LUT."""
    eq = torch.from_numpy(synth.equirect(2048, 2048)).cuda()
    chain, n = ctx.mip_chain(eq)
    pre = ctx.envmap_prefilter(chain, 2048, 2048, n, 64, 0.010, 128, abi.CONV_WAVE64)
    lut = ctx.brdf_lut(1024, 2048, abi.FMT_RG16F)
    torch.cuda.synchronize()
    return pre, lut


def upload_tile(frame_h, row0, row1):
    gb = [torch.empty((row1 - row0, W, 4), dtype=torch.float32, device="cuda") for _ in range(4)]
    for r in range(row0, row1, 240):
        part = synth.gbuffer_rows(W, frame_h, r, min(r + 240, row1), seed=0x6400)
        for k in range(4):
            gb[k][r - row0:r - row0 + part[k].shape[0]].copy_(torch.from_numpy(part[k]))
    return gb


def cpu_baseline(pre, lut, pf, pv, frame_h, target_s=12.0):
    """The CPU oracle (a scalar C++ port of the HLSL, OpenMP over rows) timed on the host cores on a bounded row
    band of the SAME workload. Reported baseline only — never the thing measured as `value`."""
    from tests import oracle_lib as O
    O.load()
    cores = len(os.sched_getaffinity(0))
    try:                                                     # honour a cgroup CPU quota if the box has one
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(float(q) / float(p))))
    except (OSError, ValueError):
        pass
    d_np, s_np, l_np = pre["diffuse_blurred"].cpu().numpy(), pre["specular"].cpu().numpy(), lut.cpu().numpy()
    env = O.host_envmap(d_np, s_np, 128, pre["spec_mips"], l_np)
    band = 540                                               # quarter-frame bands of the SAME synthetic frame

    def run(row0, rows):
        gb = synth.gbuffer_rows(W, frame_h, row0, row0 + rows, seed=0x6400)
        t0 = time.perf_counter()
        sc = O.forward_lighting(gb, pf, pv, abi.FMT_RGBA16F, env=env, nthreads=cores)
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
                      f"frame ({W * rows / 1e6:.1f} Mpix): shade 64 lights + IBL, blur X/Y, tonemap; {t:.1f} s"}


def cpu_reference_source(pre, lut, pf, pv, frame_h, target_s=8.0):
    """The REFERENCE'S OWN shader source (ForwardLighting.hlsl:PSMain, GaussianBlur.hlsl, Tonemapper.hlsl) run on one host core
    through oracle/_ref (oracle/ref_src/hlsl_shim.h) on rows of the same frame, when that library travelled with the tree. A second
    reported baseline next to `cpu_baseline`: scalar, single-threaded (the translated shaders keep their globals), literal IEEE."""
    from tests import oracle_lib as O, ref_lib as R
    if not R.available("shaders"):
        return None
    d_np, s_np, l_np = pre["diffuse_blurred"].cpu().numpy(), pre["specular"].cpu().numpy(), lut.cpu().numpy()
    env = O.host_envmap(d_np, s_np, 128, pre["spec_mips"], l_np)
    rows_per = 22                                           # one blur kernel height: the band is a (small) image of its own
    t, rows, k = 0.0, 0, 0
    while t < target_s and k < 64:
        gb = synth.gbuffer_rows(W, frame_h, (k * 97) % (frame_h - rows_per), (k * 97) % (frame_h - rows_per) + rows_per, seed=0x6400)
        n = gb[1][..., :3].astype(np.float64)
        gb[1][..., :3] = (n / np.linalg.norm(n, axis=-1, keepdims=True)).astype(np.float32)
        t0 = time.perf_counter()
        sc = R.forward_from_gbuffer(gb, pf, pv, env=env).astype(np.float16).astype(np.float32)
        x = R.blur_pass(sc, 0).astype(np.float16).astype(np.float32)
        y = R.blur_pass(x, 1).astype(np.float16).astype(np.float32)
        R.tonemap(y, abi.TonemapperParams.default())
        t += time.perf_counter() - t0
        rows += rows_per
        k += 1
    return {"value": round(W * rows / t / 1e6, 4), "unit": "Mpix/s", "cores": 1, "kind": "reference",
            "sample": f"the reference's HLSL (PSMain 64 lights + IBL, CSMain_X/_Y, tonemapper CSMain) compiled to C++ through oracle/ref_src/hlsl_shim.h, "
                      f"1 thread, {k} bands of {W}x{rows_per} rows of the same frame ({W * rows / 1e6:.2f} Mpix); {t:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)       # a step is ~1.4 ms: 50 + 10 keep the clocks ramped, the run stays setup-dominated
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--halo", choices=["p2p", "allgather"], default="p2p")
    ap.add_argument("--post", choices=["fused", "split"], default="fused",
                    help="post chain after the X blur: Y blur and tonemapper as two dispatches (split) or one kernel (fused); identical bits")
    ap.add_argument("--composite", choices=["gather", "allgather"], default="gather",
                    help="final composite of the RGBA8 tiles: gather on rank 0 (the presenting GPU; 1/N of the traffic, rank 0 receives over its "
                         "N-1 direct xGMI links) or all-gather on every rank")
    ap.add_argument("--fresnel-pow", choices=["product", "exp2_log2"], default="product",
                    help="pow(1 - cos, 5) of the Fresnel terms: the product x*((x*x)*(x*x)) (default, contract v4) or exp2(5*log2 x), the engine's own "
                         "DXC lowering (vqhip_set_fresnel_pow; DESIGN.md 3.2)")
    ap.add_argument("--overlap", action="store_true",
                    help="post chain of frame n on a second (high-priority) HIP stream overlapping the shading of frame n+1. Measured "
                         "+1 %% only (the 32 400-workgroup shade dispatch starves the second queue), so the default is ONE stream, "
                         "which also keeps the per-kernel event timings clean.")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # VQ_BENCH_SHARE_GPU=1 (debug aid for single-GPU boxes): every rank uses the visible GPUs round-robin and the collectives run
    # over gloo, so that the N > 1 control flow (tiles, halo exchange, double-buffered composite, drain) can be exercised on real
    # kernels without N GPUs. Never set by the driver; the product configuration is one GPU per rank over RCCL.
    share = os.environ.get("VQ_BENCH_SHARE_GPU") == "1"
    device_ordinal = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(device_ordinal)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_ordinal))
    ctx = capi.Context(device_ordinal)

    ctx.set_fresnel_pow(args.fresnel_pow == "exp2_log2")
    global SHADE_PMC_VALU_PER_WAVE
    if args.fresnel_pow == "exp2_log2":
        SHADE_PMC_VALU_PER_WAVE = 6480                # PMC count of that form (profiles/r1g_shade_valu.md, contract v3 row)
    if args.fresnel_pow == "exp2_log2":
        from tests import oracle_lib as _O
        _O.load().vqo_set_fresnel_pow(1)          # keeps the cpu_baseline leg on the same arithmetic
    frame_h = TILE_H * world
    tl = tiling.RowTiling(W, frame_h, world, rank)
    pre, lut = build_ibl(ctx)
    env = capi.make_envmap(pre["diffuse_blurred"], pre["specular"], 128, pre["spec_mips"], lut)
    pf, extra = synth.per_frame(points=synth.point_lights(N_LIGHTS, seed=0x6400), hdri_offset=0.3)
    pv = synth.per_view(W, frame_h, max_env_lod=pre["spec_mips"])
    gb = upload_tile(frame_h, tl.row0, tl.row1)

    F16, R8 = abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM
    scene = [capi.empty_image(TILE_H, W, F16, ctx.device) for _ in range(2)]
    xblur = capi.empty_image(TILE_H, W, F16, ctx.device)
    yblur = capi.empty_image(TILE_H, W, F16, ctx.device)
    sdr = [capi.empty_image(TILE_H, W, R8, ctx.device) for _ in range(2)]
    need_frame = world > 1 and (args.composite == "allgather" or rank == 0)
    frame = [torch.empty((frame_h, W, 4), dtype=torch.uint8, device=ctx.device) for _ in range(2)] if need_frame else [None, None]
    pending = [None, None]
    halo_fn = tiling.exchange_halos_p2p if args.halo == "p2p" else tiling.exchange_halos_allgather
    # Two HIP streams, like the reference's GFX + async-compute queues (SceneRendering.cpp:605-606,629): the VALU-bound
    # shading of frame n+1 runs on `s_shade` while the HBM-bound post chain (+ halo exchange + composite) of frame n runs
    # on `s_post`. Every frame still does all of its work inside the timed region; scene colour is double-buffered.
    s_shade = torch.cuda.current_stream(ctx.device)
    s_post = torch.cuda.Stream(ctx.device, priority=-1) if args.overlap else s_shade   # high priority: its short kernels slot in
    e_scene = [torch.cuda.Event(), torch.cuda.Event()]
    e_post = [None, None]

    def step(i, ev=None):
        b = i & 1
        if e_post[b] is not None and args.overlap:    # scene[b] was last read by the post chain of step i-2
            s_shade.wait_event(e_post[b])
        if ev:
            ev[0].record(s_shade)
        ctx.forward_lighting(gb, pf, pv, out=scene[b], out_fmt=F16, extra_point=extra, env=env)
        if ev:
            ev[1].record(s_shade)
        if args.overlap:
            e_scene[b].record(s_shade)
        with torch.cuda.stream(s_post):
            if args.overlap:
                s_post.wait_event(e_scene[b])
            if world > 1 and pending[b] is not None:  # composite of step i-2 must have drained before sdr[b]/frame[b] are reused
                pending[b].wait()
                pending[b] = None
            if ev and len(ev) == 5:
                ev[4].record(s_post)
            ctx.gaussian_blur_x(scene[b], F16, out=xblur)
            top = bottom = None
            if world > 1:
                top, bottom = halo_fn(xblur)
            if args.post == "fused":
                # CSMain_Y + Tonemapper in one kernel (register-window Y pass whose store goes through the 64 KB tonemap table in
                # LDS): bit-identical to the two dispatches, BlurOutput never touches HBM. ev[2] then closes the X pass only.
                if ev and len(ev) == 5:
                    ev[2].record(s_post)
                ctx.gaussian_blur_y_tonemap(xblur, F16, R8, out=sdr[b], halo_top=top, halo_bottom=bottom)
            else:
                ctx.gaussian_blur_y(xblur, F16, out=yblur, halo_top=top, halo_bottom=bottom)
                if ev and len(ev) == 5:
                    ev[2].record(s_post)
                ctx.tonemap(yblur, F16, R8, out=sdr[b])
            if ev and len(ev) == 5:
                ev[3].record(s_post)
            if world > 1:                              # all-gather on RCCL's own stream, drained two steps later
                if args.composite == "gather":
                    _, pending[b] = tiling.composite_to_root(sdr[b], out=frame[b], dst=0, async_op=True)
                else:
                    _, pending[b] = tiling.composite(sdr[b], out=frame[b], async_op=True)
            if args.overlap:
                e_post[b] = torch.cuda.Event()
                e_post[b].record(s_post)

    def drain():
        with torch.cuda.stream(s_post):
            for b in (0, 1):
                if pending[b] is not None:
                    pending[b].wait()
                    pending[b] = None
        s_shade.wait_stream(s_post)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed spin-up: the chip needs ~0.2-0.3 s of sustained load to reach its steady-state clocks (measured: the same
    # kernel runs 1.33 ms right after the idle set-up phase and 1.25 ms once ramped), so a fixed number of extra untimed steps
    # precedes the W warm-up steps whatever W is. The timed region is still exactly K steps.
    for i in range(SPINUP_STEPS):
        step(i)
    drain()
    for i in range(args.warmup):
        step(i)
    drain()
    # timed region: only the dominant kernel is bracketed by HIP events (2 records per step); the per-stage timings of the
    # HBM-bound post kernels are taken in a separate, untimed pass afterwards so their instrumentation does not sit in `value`
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, evs[i])
    drain()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=ctx.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    n_detail = min(args.steps, 10)
    evd = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(n_detail)]
    for i in range(n_detail):
        step(args.steps + i + (args.steps & 1), evd[i])
    drain()
    barrier()

    verify = None
    if world > 1 and os.environ.get("VQ_BENCH_VERIFY") == "1":
        # debug aid: rank 0 recomputes the WHOLE frame on its own GPU (no tiles, no halos) and compares it byte for byte with the
        # composite of the last step — the row tiling + halo exchange + composite on real kernels (tests/test_gpu_bench_flow.py)
        last = (args.steps + n_detail + (args.steps & 1) - 1) & 1
        if rank == 0:
            gb_full = upload_tile(frame_h, 0, frame_h)
            sc = ctx.forward_lighting(gb_full, pf, pv, out_fmt=F16, extra_point=extra, env=env)
            xb = ctx.gaussian_blur_x(sc, F16)
            want = ctx.gaussian_blur_y_tonemap(xb, F16, R8)
            torch.cuda.synchronize()
            verify = {"mismatching_bytes": int((want != frame[last]).sum().item()), "frame": [W, frame_h]}
            del gb_full, sc, xb, want
        dist.barrier()

    if rank == 0:
        px_tile, px_frame = W * TILE_H, W * frame_h
        t_shade = float(np.mean([e[0].elapsed_time(e[1]) for e in evs])) * 1e-3
        t_blur = float(np.mean([e[4].elapsed_time(e[2]) for e in evd])) * 1e-3
        t_tm = float(np.mean([e[2].elapsed_time(e[3]) for e in evd])) * 1e-3
        ach = SHADE_BYTES_PER_PX * px_tile / t_shade / 1e9
        out = {
            "metric": "Mpixels/s forward-PBR @4K,64 lights", "value": round(px_frame * args.steps / dt / 1e6, 2), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE cfg3: 3840x2160 float4 G-buffer tile per GPU, 64 point lights + IBL sample -> RGBA16F, "
                                   "21-tap blur X/Y, Reinhard+sRGB tonemap -> RGBA8" + ("" if world == 1 else f"; frame 3840x{frame_h} row-tiled, RCCL halo ({args.halo}) + composite ({args.composite}{' on rank 0' if args.composite == 'gather' else ''})"),
                       "width": W, "frame_height": frame_h, "lights": N_LIGHTS, "parallelism": f"rows{world}",
                       "streams": "2: post chain of frame n overlaps shading of frame n+1" if args.overlap else "1",
                       "untimed_spinup_steps": SPINUP_STEPS, "fresnel_pow": args.fresnel_pow,
                       "post": "blur X, then blur Y + tonemap in one kernel (identical bits to three dispatches)" if args.post == "fused" else "blur X, blur Y, tonemap"},
            "roofline": {"bound": "hbm", "kernel": "k_forward_lighting<env,nocasters,RGBA16F>", "achieved": round(ach, 2), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5), "traffic": SHADE_PMC_TRAFFIC_BYTES, "traffic_unit": "bytes/launch",
                         "traffic_source": "profiles/r1l_pmc_hbm.md (rocprofv3 PMC, separate passes, 2*FETCH_SIZE + WRITE_SIZE); algorithmic = %d" % (SHADE_BYTES_PER_PX * px_tile),
                         "bytes_per_px": SHADE_BYTES_PER_PX, "ms": round(t_shade * 1e3, 4),
                         "note": "64-light shading is VALU-bound by construction (SURVEY.md 8d): see valu"},
            "valu": {"achieved_tflops_model": round(SHADE_FLOPS_PER_PX * px_tile / t_shade / 1e12, 2), "peak": VALU_PEAK_TFLOPS,
                     "frac": round(SHADE_FLOPS_PER_PX * px_tile / t_shade / 1e12 / VALU_PEAK_TFLOPS, 4), "flops_per_px_model": SHADE_FLOPS_PER_PX},
            "valu_issue": {"achieved_T_lane_instr_s": round(SHADE_PMC_VALU_PER_WAVE * px_tile / t_shade / 1e12, 2),
                           "ceiling": VALU_ISSUE_CEILING_TLIS, "frac": round(SHADE_PMC_VALU_PER_WAVE * px_tile / t_shade / 1e12 / VALU_ISSUE_CEILING_TLIS, 4),
                           "frac_slot_weighted": round((SHADE_PMC_VALU_PER_WAVE + 3 * SHADE_TRANS_PER_WAVE) * px_tile / t_shade / 1e12 / VALU_ISSUE_CEILING_TLIS, 4),
                           "valu_instr_per_wave": SHADE_PMC_VALU_PER_WAVE, "quarter_rate_instr_per_wave": SHADE_TRANS_PER_WAVE,
                           "note": "the binding roof: VALU instructions issued per second (PMC count x live kernel time) vs the steady-state v_fma_f32 "
                                   "issue rate of the chip (scripts/ubench/valu_ceiling.hip); slot-weighted counts each quarter-rate v_rcp/v_rsq as 4 slots"},
            "stages": {"shade_Mpix_s": round(px_tile / t_shade / 1e6, 1), "shade_ms": round(t_shade * 1e3, 4),
                       **({"blur_xy_ms": round(t_blur * 1e3, 4), "blur_xy_GBps": round(px_tile * 32 / t_blur / 1e9, 1),
                           "tonemap_ms": round(t_tm * 1e3, 4), "tonemap_GBps": round(px_tile * 12 / t_tm / 1e9, 1)} if args.post == "split" else
                          {"blur_x_ms": round(t_blur * 1e3, 4), "blur_x_GBps": round(px_tile * 16 / t_blur / 1e9, 1),
                           "blur_y_tonemap_ms": round(t_tm * 1e3, 4), "blur_y_tonemap_GBps": round(px_tile * 12 / t_tm / 1e9, 1),
                           "post_algorithmic_GBps_split_equivalent": round(px_tile * 44 / (t_blur + t_tm) / 1e9, 1)})},
        }
        if verify is not None:
            out["verify"] = verify
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pre, lut, pf, pv, frame_h)
            try:                                             # optional second baseline: only where oracle/_ref exists
                ref_line = cpu_reference_source(pre, lut, pf, pv, frame_h)
                if ref_line is not None:
                    out["cpu_reference_source"] = ref_line
            except Exception as e:                           # never let the optional leg break the bench line
                out["cpu_reference_source"] = {"error": repr(e)[:200]}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
