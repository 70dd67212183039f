#!/usr/bin/env python3
"""Load-time kernels of BASELINE config 4 (2048^2 equirect -> 64^2 diffuse at step 0.010, 128^2 x 7 specular, 1024^2 x 2048 BRDF LUT), one box:
warm time of every form of each kernel (options lut_form, diffuse_form, diffuse_seq_form) and bit-equality with the first form.
Prints one JSON line per measurement."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqengine_amd import abi, capi, synth  # noqa: E402


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        out = fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps, out


def sweep(ctx, what, env, forms, fn):
    ref = None
    for form in forms:
        ctx.set_option_env(env, None if form == "default" else form)     # the round-1..3 environment knobs are context options now (vqhip_set_option)
        ms, out = timed(fn)
        out = out[0] if isinstance(out, tuple) else out
        if ref is None:
            ref = out.clone()
        print(json.dumps({"what": what, "form": form, "ms": round(ms, 4), "identical_to_first": bool(torch.equal(ref.view(torch.uint8), out.view(torch.uint8)))}), flush=True)
    ctx.set_option_env(env, None)


def main():
    ctx = capi.Context(0)
    eq = torch.from_numpy(synth.equirect(2048, 2048)).cuda()
    chain, n = ctx.mip_chain(eq)
    x = torch.empty(64 << 20, device="cuda")
    for _ in range(200):                                   # spin the clocks up
        x.mul_(1.0001)
    sweep(ctx, "brdf_lut 1024^2 x 2048", "VQHIP_LUT_FORM", ["default", "general"], lambda: ctx.brdf_lut(1024, 2048, abi.FMT_RG16F))
    sweep(ctx, "conv_diffuse 6x64^2 step 0.010 wave64", "VQHIP_DIFFUSE_FORM", ["default", "texels", "general"],
          lambda: ctx.conv_diffuse(chain, 2048, 2048, n, 64, 0.010, abi.CONV_WAVE64, abi.FMT_RGBA16F))
    sweep(ctx, "conv_diffuse 6x64^2 step 0.010 sequential (the reference's order)", "VQHIP_DIFFUSE_SEQ_FORM", ["default", "lane"],
          lambda: ctx.conv_diffuse(chain, 2048, 2048, n, 64, 0.010, abi.CONV_SEQUENTIAL, abi.FMT_RGBA16F))
    for order, tag in ((abi.CONV_SEQUENTIAL, "sequential"), (abi.CONV_WAVE64, "wave64")):
        ms, _ = timed(lambda: ctx.conv_specular(chain, 2048, 2048, n, 128, order, abi.FMT_RGBA16F))
        print(json.dumps({"what": f"conv_specular 128^2 x 7 {tag}", "form": "default", "ms": round(ms, 4)}), flush=True)
    ms, _ = timed(lambda: ctx.mip_chain(eq))
    print(json.dumps({"what": "mip_chain 2048^2 (incl. the level-0 copy)", "ms": round(ms, 4)}), flush=True)


if __name__ == "__main__":
    main()
