#!/usr/bin/env python3
"""Round 5: forms of the post chain at 4K on one MI355X, one JSON line per form (bench.py's stage timer: >= 0.25 s spin-up, median of 7 batches).
  * the X pass alone, the fused Y blur + tonemap alone (36-row window kernel vs the rolling-ring kernel at several strip heights), the pair X -> Y back to back
  * every form's output is compared byte for byte with the window kernel's
usage: python scripts/bench_post5.py [W H]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vqengine_amd import abi, capi, synth  # noqa: E402

F16, R8 = abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM


def main():
    W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
    ctx = capi.Context(0)
    band = synth.hdr_image(W, 270, scale=8.0).astype(np.float16)
    scene = torch.from_numpy(np.tile(band, (H // 270 + 1, 1, 1))[:H].copy()).cuda()
    xb = capi.empty_image(H, W, F16, ctx.device)
    sdr = capi.empty_image(H, W, R8, ctx.device)
    px = W * H

    def line(name, st, bytes_px, **kw):
        print(json.dumps(dict(form=name, us=round(st["ms"] * 1e3, 2), us_min=round(st["ms_min"] * 1e3, 2), us_max=round(st["ms_max"] * 1e3, 2),
                              TBps=round(px * bytes_px / st["ms"] / 1e9, 3), hbm_frac=round(px * bytes_px / st["ms"] / 1e9 / 8.0, 4), **kw)), flush=True)

    line("blur_x", bench._stage_stats(lambda: ctx.gaussian_blur_x(scene, F16, out=xb)), 16)
    ctx.set_option("blur_y_form", "window")
    ref = ctx.gaussian_blur_y_tonemap(xb, F16, R8).clone()
    line("y_window_36", bench._stage_stats(lambda: ctx.gaussian_blur_y_tonemap(xb, F16, R8, out=sdr)), 12)
    line("pair_x_then_y_window", bench._stage_stats(lambda: (ctx.gaussian_blur_x(scene, F16, out=xb), ctx.gaussian_blur_y_tonemap(xb, F16, R8, out=sdr))), 28)
    ctx.set_option("blur_y_form", None)
    for S in (24, 32, 48, 64, 96, 128, 192, 270):
        ctx.set_option("blur_y_rows", S)
        sdr.zero_()
        st = bench._stage_stats(lambda: ctx.gaussian_blur_y_tonemap(xb, F16, R8, out=sdr))
        torch.cuda.synchronize()
        line(f"y_roll_S{S}", st, 12, mismatching_bytes=int((sdr != ref).sum().item()), wave_strips=((W + 63) // 64) * ((H + S - 1) // S))
    for S in (48, 64, 96):
        ctx.set_option("blur_y_rows", S)
        line(f"pair_x_then_y_roll_S{S}", bench._stage_stats(lambda: (ctx.gaussian_blur_x(scene, F16, out=xb), ctx.gaussian_blur_y_tonemap(xb, F16, R8, out=sdr))), 28)
    ctx.close()


if __name__ == "__main__":
    main()
