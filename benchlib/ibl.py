"""Load-time IBL at the ENGINE'S DEFAULT sizes (Data/EngineSettings.ini:11 EnvironmentMapResolution=512 -> specular 512^2 x 9 mips; EnvironmentMap.cpp:164-165: a
2:1 4096 x 2048 .hdr -> 13 source mips; EnvironmentMapRendering.cpp:55-63,431-435), next to bench.py's `ibl_load` (BASELINE cfg4: 2048^2 -> 128^2 x 7). Timing only;
parity of this configuration: tests/test_engine_default.py."""
import time

import numpy as np
import torch

from vqengine_amd import abi, synth

from .consts import HBM_PEAK_GBPS, VALU_PEAK_TFLOPS

W0, H0, SPEC_RES0, DIFF_RES, DIFF_STEP = 4096, 2048, 512, 64, 0.010


def _hdr_file():
    """a run-length coded 4096 x 2048 .hdr: 256 distinct scanlines, repeated (the encoder is Python; the decoder's work does not depend on the repetition)"""
    rgbe = synth.float_to_rgbe(synth.equirect(W0, 256, seed=0xE9D)[..., :3])
    part = synth.hdr_file_bytes(rgbe)
    body = part[part.index(b"+X %d\n" % W0) + len(b"+X %d\n" % W0):]
    return part[:part.index(b"-Y ")] + b"-Y %d +X %d\n" % (H0, W0) + body * (H0 // 256)


def engine_default_report(ctx, stage_stats):
    data = _hdr_file()
    ctx.load_hdr(data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        img = ctx.load_hdr(data)
    torch.cuda.synchronize()
    decode_ms = (time.perf_counter() - t0) / 3 * 1e3
    chain, n = ctx.mip_chain(img)
    mips = abi.specular_mip_count(SPEC_RES0)
    out = {"workload": f"engine default: {W0}x{H0} RLE .hdr -> RGBA32F -> {n}-level min-filter chain -> diffuse irradiance 6x{DIFF_RES}^2 at step {DIFF_STEP} + blur -> "
                       f"GGX specular {SPEC_RES0}^2 x {mips} mips ({abi.cube_px(SPEC_RES0, mips)} texels x 512 taps)",
           "hdr_file_bytes": len(data), "hdr_decode_wall_ms": round(decode_ms, 3), "source_mips": int(n), "spec_mips": int(mips)}
    for key, fn in (("mip_chain_ms", lambda: ctx.mip_chain(img)),
                    ("conv_diffuse_ms", lambda: ctx.conv_diffuse(chain, W0, H0, n, DIFF_RES, DIFF_STEP, abi.CONV_SEQUENTIAL, abi.FMT_RGBA16F)),
                    ("conv_specular_ms", lambda: ctx.conv_specular(chain, W0, H0, n, SPEC_RES0, abi.CONV_SEQUENTIAL, abi.FMT_RGBA16F)),
                    ("prefilter_ms", lambda: ctx.envmap_prefilter(chain, W0, H0, n, DIFF_RES, DIFF_STEP, SPEC_RES0, abi.CONV_SEQUENTIAL))):
        st = stage_stats(fn)
        out[key] = round(st["ms"], 4)
        out[key + "_spread"] = [round(st["ms_min"], 4), round(st["ms_max"], 4)]
    taps = abi.cube_px(SPEC_RES0, mips) * 512
    out["conv_specular_Gtaps_s"] = round(taps / out["conv_specular_ms"] / 1e6, 2)
    out["note"] = ("every *_ms: >= 0.25 s spin-up of the one call, median of 7 back-to-back batches; hdr_decode_wall_ms = the call + stream sync (host walk of the run headers, upload, "
                   "one expansion kernel). The specular pass is 16x cfg4's texels (ibl_load.conv_specular_ms); the BRDF LUT does not depend on the environment (ibl_load.brdf_lut_warm_ms)")
    del img, chain
    torch.cuda.empty_cache()
    return out


def ibl_load_report(t):
    """BASELINE config 4 as the engine runs it at load time (EnvironmentMapRendering.cpp:139-486, Renderer.cpp:871-909), with the flop models
    of SURVEY.md 8d: diffuse 6x64^2 texels x 99 382 taps x ~60 flop, specular 67 M taps x ~80, LUT 1024^2 x 2048 samples x ~80."""
    model = {"conv_diffuse_ms": 6 * 64 * 64 * 99382 * 60.0, "conv_specular_ms": 67.1e6 * 80.0, "brdf_lut_warm_ms": 1024 * 1024 * 2048 * 80.0}
    out = dict(t)
    out["total_ms"] = round(t["mip_chain_ms"] + t["prefilter_ms"] + t["brdf_lut_ms"], 4)
    out["warm_total_ms"] = round(t["mip_chain_warm_ms"] + t["prefilter_warm_ms"] + t["brdf_lut_warm_ms"], 4)
    for k, fl in model.items():
        out[k.replace("_ms", "_valu_frac_model")] = round(fl / (t[k] * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 4)
    out["mip_chain_hbm_frac"] = round((2048 * 2048 * 16 * 5.0 / 3.0) / (t["mip_chain_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)   # each level read once, written once
    out["workload"] = "BASELINE cfg4: 2048^2 RGBA32F equirect -> 12-level min-filter chain, diffuse irradiance 6x64^2 at step 0.010 (99 382 taps/texel) + blur, 7-mip GGX specular 128^2, BRDF LUT 1024^2 x 2048"
    out["note"] = ("mip_chain / prefilter (diffuse + face blur + specular) / brdf_lut are the product calls as build_ibl() issues them, first use of each kernel "
                   "(code load and cold clocks included: total_ms); conv_diffuse / conv_specular / brdf_lut_warm / mip_chain_warm / prefilter_warm are the stage on its "
                   "own in back-to-back calls after a >= 0.25 s spin-up (median of 7 batches, *_ms_spread = [min, max]), like every other per-stage figure (warm_total_ms = mip chain + prefilter + "
                   "LUT of those); *_single_call_ms = ONE call without a spin-up (what these keys meant in rounds 1-3). The convolutions run in the reference's "
                   "summation order (wave-parallel taps parked in LDS, added in order: profiles/r4a_conv_ordered.md); the diffuse one is bound by L1 tag lookups")
    return out
