#!/usr/bin/env python3
"""Round 5: forms of the post chain at 4K on one MI355X, one JSON line per form (bench.py's stage timer: >= 0.25 s spin-up, median of 7 batches).
  * the X pass alone, the fused Y blur + tonemap alone, vqhip_post_process as two kernels, and as the one-kernel chain (k_post_chain) at several strip heights
  * hbm_frac counts the 28 B/px of the two-kernel chain for every form (so that the column compares times); own_hbm_frac counts the form's own 12 B/px
  * every form's output is compared byte for byte with the two-kernel path's
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
    line("blur_y_tonemap (36-row window)", bench._stage_stats(lambda: ctx.gaussian_blur_y_tonemap(xb, F16, R8, out=sdr)), 12)
    ctx.set_option("post_form", "two")
    ref = ctx.post_process(scene, F16, R8).clone()
    line("post_process two kernels", bench._stage_stats(lambda: ctx.post_process(scene, F16, R8, out=sdr)), 28, own_bytes_px=28)
    ctx.set_option("post_form", "chain")
    strips = [int(a) for a in os.environ.get("VQ_POST_STRIPS", "0,4,6,7,8,9,12,16,17,24,34").split(",")]
    for ny, mix in [(n, 0) for n in strips]:                  # (the converted-window form of round 5, post_mix = 1, is no longer in the library)
        ctx.set_option("post_strips", ny if ny else None)
        sdr.zero_()
        st = bench._stage_stats(lambda: ctx.post_process(scene, F16, R8, out=sdr))
        torch.cuda.synchronize()
        us = st["ms"] * 1e3
        line(f"post_process chain strips={ny or 'default'} {'cvt+fmac' if mix else 'v_fma_mix'}", st, 28, own_bytes_px=12, own_hbm_frac=round(px * 12 / st["ms"] / 1e9 / 8.0, 4),
             mismatching_bytes=int((sdr != ref).sum().item()), workgroups=((W + 127) // 128) * (ny or max(1, 256 // ((W + 127) // 128))))
    ctx.close()


if __name__ == "__main__":
    main()
