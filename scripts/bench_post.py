#!/usr/bin/env python3
"""Post chain at 4K (cfg3 formats: RGBA16F scene colour -> RGBA8), steady-state clocks: the three ways the product can run it —
  split   : blur X, blur Y, tonemap (three dispatches)            algorithmic 8+8, 8+8, 8+4 = 44 B/px
  fused-y : blur X, then blur Y + tonemap in one kernel          8+8, 8+4 = 28 B/px
  one     : vqhip_post_process (k_post_fused, one kernel)        8+4 = 12 B/px
timed with HIP events over back-to-back launches after a spin-up. Prints one JSON line per variant."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqengine_amd import abi, capi, synth  # noqa: E402

F16, R8 = abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM


def timed(fn, reps=int(os.environ.get("VQ_POST_REPS", "200")), spin=int(os.environ.get("VQ_POST_SPIN", "300"))):
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
    W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
    ctx = capi.Context(0)
    img = torch.from_numpy(synth.hdr_image(W, H).astype(np.float16)).cuda()
    xb, yb = torch.empty_like(img), torch.empty_like(img)
    sdr = capi.empty_image(H, W, R8, ctx.device)
    px = W * H

    def split():
        ctx.gaussian_blur_x(img, F16, out=xb); ctx.gaussian_blur_y(xb, F16, out=yb); ctx.tonemap(yb, F16, R8, out=sdr)

    def fused_y():
        ctx.gaussian_blur_x(img, F16, out=xb); ctx.gaussian_blur_y_tonemap(xb, F16, R8, out=sdr)

    def one():
        ctx.post_process(img, F16, R8, out=sdr)
    def x_only():
        ctx.gaussian_blur_x(img, F16, out=xb)

    def ytm_only():
        ctx.gaussian_blur_y_tonemap(xb, F16, R8, out=sdr)

    def copy16():                                            # the box's copy rate on the X pass's bytes (8 B/px in, 8 B/px out)
        yb.copy_(img)
    if os.environ.get("VQ_POST_PARTS") == "1":
        for name, fn, bpp in (("copy 8+8 B/px", copy16, 16), ("blur X alone", x_only, 16), ("blur Y + tonemap alone", ytm_only, 12)):
            ms = timed(fn)
            print(json.dumps({"variant": name, "us": round(ms * 1e3, 2), "GBps_algorithmic": round(px * bpp / ms / 1e6, 1), "frac_of_8TBps": round(px * bpp / ms / 1e6 / 8000, 4)}), flush=True)
    if os.environ.get("VQ_POST_PARTS_ONLY") == "1":
        return
    ref = None
    def one_mode(m):
        def f():
            ctx.post_process(img, F16, R8, out=sdr)
        f.mode = m
        return f
    for name, fn, bpp in (("split", split, 44), ("fused-y", fused_y, 28), ("one", one_mode("1"), 12), ("one-compact-table", one_mode("1c"), 12)):
        if hasattr(fn, "mode"):
            ctx.set_option_env("VQHIP_POST_ONE_KERNEL", fn.mode)
        ms = timed(fn)
        fn(); torch.cuda.synchronize()
        if ref is None:
            ref = sdr.clone()
        same = bool(torch.equal(ref, sdr))
        ctx.set_option_env("VQHIP_POST_ONE_KERNEL", None)
        print(json.dumps({"variant": name, "size": [W, H], "us": round(ms * 1e3, 2), "algorithmic_B_per_px": bpp, "GBps_algorithmic": round(px * bpp / ms / 1e6, 1),
                          "frac_of_8TBps": round(px * bpp / ms / 1e6 / 8000, 4), "GBps_of_the_28B_chain": round(px * 28 / ms / 1e6, 1),
                          "identical_to_split": same}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
