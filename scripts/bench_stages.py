#!/usr/bin/env python3
"""Per-kernel measurement of every SURVEY.md §8 row on one MI355X: GPU time (median of HIP-event intervals on the
launch stream), algorithmic bytes / flops, fraction of the roof that bounds the kernel, and the CPU oracle timed on
a bounded sample beside it. Prints one JSON object per stage; `profiles/r1_stages.jsonl` is a committed run.

Roofs: HBM 8.0 TB/s spec (6.29 TB/s measured copy), FP32 VALU 157.3 TFLOP/s spec (105 TFLOP/s measured v_fma,
scripts/ubench/valu_ubench.hip). flop models are SURVEY.md §8(d)'s (170*L+160 per shaded pixel, ~60 per diffuse tap,
~80 per specular / LUT sample)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import oracle_lib as O  # noqa: E402
from vqengine_amd import abi, capi, synth  # noqa: E402

HBM, VALU = 8000.0, 157.3
F16, F32, R8 = abi.FMT_RGBA16F, abi.FMT_RGBA32F, abi.FMT_RGBA8_UNORM


def gpu_ms(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def cpu_s(fn):
    if os.environ.get("VQ_STAGES_NO_CPU") == "1":            # GPU-only A/B runs (scripts/bench_variants-style comparisons)
        return 0
    t0 = time.perf_counter(); fn(); return time.perf_counter() - t0


def emit(stage, units, unit_name, ms, bytes_per_unit, flops_per_unit, cpu_units, cpu_t, note=""):
    gbps = units * bytes_per_unit / (ms * 1e-3) / 1e9 if bytes_per_unit else None
    tfl = units * flops_per_unit / (ms * 1e-3) / 1e12 if flops_per_unit else None
    d = {"stage": stage, "units": units, "unit": unit_name, "gpu_ms": round(ms, 4), "M_units_per_s": round(units / (ms * 1e-3) / 1e6, 2),
         "algorithmic_GBps": round(gbps, 1) if gbps else None, "hbm_frac": round(gbps / HBM, 4) if gbps else None,
         "model_TFLOPs": round(tfl, 2) if tfl else None, "valu_frac": round(tfl / VALU, 4) if tfl else None,
         "cpu_oracle_M_units_per_s": round(cpu_units / cpu_t / 1e6, 4) if cpu_t else None, "note": note}
    print(json.dumps(d), flush=True)


def main():
    ctx = capi.Context(0)
    cores = len(os.sched_getaffinity(0))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731

    # ---- load-time IBL (cfg4): 2048^2 equirect -> mips, diffuse 64^2 step 0.010, specular 128^2 7 mips, LUT 1024^2 x 2048
    eq = synth.equirect(2048, 2048)
    eq_g = dev(eq)
    chain, n = ctx.mip_chain(eq_g)
    emit("C1 mip chain min-filter 2048^2", 2048 * 2048 // 3 * 4, "dst texel", gpu_ms(lambda: ctx.mip_chain(eq_g)), 80, 0, 0, 0,
         "includes the level-0 copy into the chain buffer")
    chain_o, _ = O.mip_chain(eq)
    taps = 6 * 64 * 64 * 629 * 158
    t_cpu = cpu_s(lambda: O.conv_diffuse(chain_o, 2048, 2048, n, 64, 0.010, abi.CONV_WAVE64, F16, t0=0, t1=96))
    emit("B1 conv diffuse 6x64^2 step 0.010 (WAVE64)", taps, "tap", gpu_ms(lambda: ctx.conv_diffuse(chain, 2048, 2048, n, 64, 0.010), reps=5, warm=1),
         0, 60, 96 * 629 * 158, t_cpu, "629 x 158 taps per texel; cpu sample = 96 texels")
    emit("B1 conv diffuse 6x64^2 step 0.010 (SEQUENTIAL)", taps, "tap",
         gpu_ms(lambda: ctx.conv_diffuse(chain, 2048, 2048, n, 64, 0.010, abi.CONV_SEQUENTIAL), reps=3, warm=1), 0, 60, 0, 0, "HLSL loop order, one lane per texel")
    stex = abi.cube_px(128, 7)
    small_o, _ = O.mip_chain(synth.equirect(256, 256))
    t_cpu = cpu_s(lambda: O.conv_specular(small_o, 256, 256, 9, 32, abi.CONV_WAVE64, F16))
    emit("B3 conv specular 128^2 7 mips", stex * 512, "sample", gpu_ms(lambda: ctx.conv_specular(chain, 2048, 2048, n, 128), reps=5, warm=1),
         0, 80, abi.cube_px(32, 5) * 512, t_cpu, "cpu sample = 32^2 5-mip cube from a 256^2 equirect")
    t_cpu = cpu_s(lambda: O.brdf_lut(1024, 2048, abi.FMT_RG16F, rows=(500, 516)))
    emit("B4 BRDF LUT 1024^2 x 2048", 1024 * 1024 * 2048, "sample", gpu_ms(lambda: ctx.brdf_lut(1024, 2048), reps=5, warm=1), 0, 80, 16 * 1024 * 2048, t_cpu,
         "cpu sample = 16 rows")
    pre = ctx.envmap_prefilter(chain, 2048, 2048, n, 64, 0.010, 128)
    lut = ctx.brdf_lut(1024, 2048)
    env = capi.make_envmap(pre["diffuse_blurred"], pre["specular"], 128, pre["spec_mips"], lut)
    env_o = O.host_envmap(pre["diffuse_blurred"].cpu().numpy(), pre["specular"].cpu().numpy(), 128, pre["spec_mips"], lut.cpu().numpy())

    # ---- shading (cfg2, cfg3, one cfg5 tile)
    for name, W, H, frame_h, L, use_env, seed in (("A1 shade cfg2 1920x1080 16 lights", 1920, 1080, 1080, 16, False, 0xC0FFEE),
                                                  ("A1 shade cfg3 3840x2160 64 lights + IBL", 3840, 2160, 2160, 64, True, 0x6400),
                                                  ("A1 shade cfg5 tile 7680x540 of 7680x4320, 256 lights", 7680, 540, 4320, 256, False, 0x25600)):
        gb = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(4)]
        for r in range(0, H, 180):
            part = synth.gbuffer_rows(W, frame_h, r, min(r + 180, H), seed=seed)
            for k in range(4):
                gb[k][r:r + part[k].shape[0]].copy_(torch.from_numpy(part[k]))
        pf, extra = synth.per_frame(points=synth.point_lights(L, seed=seed), hdri_offset=0.3)
        pv = synth.per_view(W, frame_h, max_env_lod=pre["spec_mips"])
        out = capi.empty_image(H, W, F16, ctx.device)
        ms = gpu_ms(lambda: ctx.forward_lighting(gb, pf, pv, out=out, out_fmt=F16, extra_point=extra, env=env if use_env else None))
        rows = 24
        gbc = synth.gbuffer_rows(W, frame_h, 100, 100 + rows, seed=seed)
        t_cpu = cpu_s(lambda: O.forward_lighting(gbc, pf, pv, F16, extra_point=extra, env=env_o if use_env else None, nthreads=cores))
        emit(name, W * H, "pixel", ms, 72, 170 * L + (160 if use_env else 0), W * rows, t_cpu, f"RGBA16F out; cpu sample = {rows} rows, {cores} threads")

    # ---- post chain at 4K (cfg3), reference storage formats
    img = dev(synth.hdr_image(3840, 2160).astype(np.float16))
    xb, yb = torch.empty_like(img), torch.empty_like(img)
    sdr = capi.empty_image(2160, 3840, R8, ctx.device)
    px = 3840 * 2160
    band = synth.hdr_image(3840, 128).astype(np.float16)
    emit("D2 blur X 3840x2160 RGBA16F", px, "pixel", gpu_ms(lambda: ctx.gaussian_blur_x(img, F16, out=xb)), 16, 126, 3840 * 128,
         cpu_s(lambda: O.blur_pass(band, F16, 0, nthreads=cores)), "cpu sample = 128 rows")
    emit("D2 blur Y 3840x2160 RGBA16F", px, "pixel", gpu_ms(lambda: ctx.gaussian_blur_y(xb, F16, out=yb)), 16, 126, 3840 * 128,
         cpu_s(lambda: O.blur_pass(band, F16, 1, nthreads=cores)), "cpu sample = 128 rows")
    emit("D1 tonemap 3840x2160 RGBA16F -> RGBA8 (Reinhard + sRGB)", px, "pixel", gpu_ms(lambda: ctx.tonemap(yb, F16, R8, out=sdr)), 12, 25, 3840 * 128,
         cpu_s(lambda: O.tonemap(band, F16, R8, nthreads=cores)), "cpu sample = 128 rows")
    emit("D2+D1 fused blur Y + tonemap 3840x2160", px, "pixel", gpu_ms(lambda: ctx.gaussian_blur_y_tonemap(xb, F16, R8, out=sdr)), 12, 151, 0, 0,
         "bit-identical to the two dispatches; slower at 4K, not used by bench.py")
    pq = abi.TonemapperParams(abi.COLOR_SPACE_REC_709, abi.DISPLAY_CURVE_ST2084, 200.0, 1)
    hdr = capi.empty_image(2160, 3840, F16, ctx.device)
    emit("D1 tonemap 3840x2160 RGBA16F -> RGBA16F (Rec709->2020 + ST2084)", px, "pixel", gpu_ms(lambda: ctx.tonemap(yb, F16, F16, params=pq, out=hdr)), 16, 60, 0, 0, "HDR path")
    ctx.close()


if __name__ == "__main__":
    main()
