#!/usr/bin/env python3
"""Forms of the fused Y blur + tonemap kernel (VQHIP_BLUR_Y_FORM) and of the table tonemapper (VQHIP_TONEMAP_FORM) at 4K, one box:
isolated time of each form (back-to-back launches after a spin-up), bit-equality with the round-2 form, and — for the forms named in
VQ_YFORMS_FRAME — the cfg3 frame loop (shade + blur X + this form), which is what decides (profiles/r2k_frame_loop.md).
Prints one JSON line per measurement."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqengine_amd import abi, capi, synth  # noqa: E402

F16, R8 = abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM
FORMS = os.environ.get("VQ_YFORMS", "lut64,c8,c8s,c8sw5,c12,c12s,c16,c16s").split(",")


def timed(fn, reps=200, spin=300):
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
    px = W * H
    img = torch.from_numpy(synth.hdr_image(W, H).astype(np.float16)).cuda()
    xb = ctx.gaussian_blur_x(img, F16)
    yb = ctx.gaussian_blur_y(xb, F16)
    sdr = capi.empty_image(H, W, R8, ctx.device)
    ms = timed(lambda: yb.copy_(xb))
    print(json.dumps({"what": "copy 8+8 B/px", "us": round(ms * 1e3, 2), "TBps": round(px * 16 / ms / 1e9, 3)}), flush=True)
    ms = timed(lambda: ctx.gaussian_blur_x(img, F16, out=xb))
    print(json.dumps({"what": "blur X", "us": round(ms * 1e3, 2), "frac_of_8TBps": round(px * 16 / ms / 1e6 / 8000, 4)}), flush=True)
    ref = None
    for form in FORMS:
        ctx.set_option_env("VQHIP_BLUR_Y_FORM", form)
        ms = timed(lambda: ctx.gaussian_blur_y_tonemap(xb, F16, R8, out=sdr))
        torch.cuda.synchronize()
        if ref is None:
            ref = sdr.clone()
        print(json.dumps({"what": "blur Y + tonemap", "form": form, "size": [W, H], "us": round(ms * 1e3, 2), "frac_of_8TBps": round(px * 12 / ms / 1e6 / 8000, 4),
                          "identical_to_first": bool(torch.equal(ref, sdr))}), flush=True)
    for form in ("lut64", "compact"):
        os.environ["VQHIP_TONEMAP_FORM"] = form
        ms = timed(lambda: ctx.tonemap(yb, F16, R8, out=sdr))
        print(json.dumps({"what": "tonemap RGBA16F->RGBA8", "form": form, "us": round(ms * 1e3, 2), "frac_of_8TBps": round(px * 12 / ms / 1e6 / 8000, 4)}), flush=True)
    os.environ.pop("VQHIP_TONEMAP_FORM")
    frame_forms = [f for f in os.environ.get("VQ_YFORMS_FRAME", "").split(",") if f]
    if frame_forms:
        import bench
        cfg = bench.CONFIGS["cfg3"]
        pre, lut = bench.build_ibl(ctx)
        env = capi.make_envmap(pre["diffuse_blurred"], pre["specular"], 128, pre["spec_mips"], lut)
        gb = bench.upload_tile(cfg, H, 0, H)
        pv = synth.per_view(W, H, max_env_lod=pre["spec_mips"])
        pf, extra = synth.per_frame(points=synth.point_lights(cfg["lights"], seed=cfg["seed"]), hdri_offset=0.3)
        scene = [capi.empty_image(H, W, F16, ctx.device) for _ in range(2)]
        out = [capi.empty_image(H, W, R8, ctx.device) for _ in range(2)]

        def frame(i=[0]):
            b = i[0] & 1; i[0] += 1
            ctx.forward_lighting(gb, pf, pv, out=scene[b], out_fmt=F16, extra_point=extra, env=env)
            ctx.gaussian_blur_x(scene[b], F16, out=xb)
            ctx.gaussian_blur_y_tonemap(xb, F16, R8, out=out[b])
        for rep in range(2):
            for form in frame_forms:
                ctx.set_option_env("VQHIP_BLUR_Y_FORM", form)
                ms = timed(frame, reps=200, spin=300)
                print(json.dumps({"what": "cfg3 frame loop", "form": form, "rep": rep, "ms": round(ms, 4), "Mpix_s": round(px / ms / 1e3, 1)}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
