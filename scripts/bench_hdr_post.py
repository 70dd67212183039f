#!/usr/bin/env python3
"""The post chain outside the table path, 4K, one box: the HDR default (ST2084 on Rec.709 content: a 3x3 matrix in front of the curve, so no
64 K-entry table applies), the sRGB curve on RGBA32F images, and — for scale — the SDR default through the table. Isolated times of
the direct tonemapper, the fused Y blur + tonemap, and the whole chain (blur X + fused Y). VQHIP_LIBRARY_PATH selects another build for an A/B.
Prints one JSON line per measurement."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqengine_amd import abi, capi, synth  # noqa: E402

F16, F32, R8 = abi.FMT_RGBA16F, abi.FMT_RGBA32F, abi.FMT_RGBA8_UNORM


def timed(fn, reps=100, spin=150):
    for _ in range(spin):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps


def main():
    W, H = 3840, 2160
    ctx = capi.Context(0)
    lib = os.environ.get("VQHIP_LIBRARY_PATH", "default build")
    img16 = torch.from_numpy(synth.hdr_image(W, H).astype(np.float16)).cuda()
    img32 = img16.float()
    xb = ctx.gaussian_blur_x(img16, F16)
    hdr = abi.TonemapperParams(abi.COLOR_SPACE_REC_709, abi.DISPLAY_CURVE_ST2084, 200.0, 1)
    sdr = abi.TonemapperParams.default()
    out16 = capi.empty_image(H, W, F16, ctx.device)
    out8 = capi.empty_image(H, W, R8, ctx.device)
    sums = {}
    for what, fn, out in (
            ("tonemap direct RGBA16F->RGBA16F, ST2084 on Rec.709 (HDR default)", lambda: ctx.tonemap(img16, F16, F16, hdr, out=out16), out16),
            ("blur Y + tonemap fused RGBA16F->RGBA16F, HDR default", lambda: ctx.gaussian_blur_y_tonemap(xb, F16, F16, hdr, out=out16), out16),
            ("tonemap direct RGBA32F->RGBA8, sRGB", lambda: ctx.tonemap(img32, F32, R8, sdr, out=out8), out8),
            ("tonemap table RGBA16F->RGBA8, sRGB (SDR default)", lambda: ctx.tonemap(img16, F16, R8, sdr, out=out8), out8)):
        ms = timed(fn)
        torch.cuda.synchronize()
        sums[what] = int(out.view(torch.uint8).to(torch.int64).sum().item())
        print(json.dumps({"what": what, "lib": lib, "us": round(ms * 1e3, 2), "checksum": sums[what]}), flush=True)


if __name__ == "__main__":
    main()
