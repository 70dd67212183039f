#!/usr/bin/env python3
"""What the parts of the cfg3 shade kernel cost on ONE box: the same 3840x2160 G-buffer shaded with 64 lights + IBL, 64 lights alone,
IBL alone, and neither (set-up + store only); 200 back-to-back launches each after a spin-up, HIP events on the launch stream.
Prints one JSON line (committed runs: profiles/r2*_shade_parts.json)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vqengine_amd import abi, capi, synth  # noqa: E402


def main():
    cfg = bench.CONFIGS["cfg3"]
    ctx = capi.Context(0)
    W, H, L = cfg["width"], cfg["height"], cfg["lights"]
    pre, lut = bench.build_ibl(ctx)
    env = capi.make_envmap(pre["diffuse_blurred"], pre["specular"], 128, pre["spec_mips"], lut)
    gb = bench.upload_tile(cfg, H, 0, H)
    out = capi.empty_image(H, W, abi.FMT_RGBA16F, ctx.device)
    pv = synth.per_view(W, H, max_env_lod=pre["spec_mips"])
    res = {}
    for name, nl, e in (("lights+ibl", L, env), ("lights", L, None), ("ibl", 0, env), ("neither", 0, None)):
        pf, extra = synth.per_frame(points=synth.point_lights(nl, seed=cfg["seed"]) if nl else None, hdri_offset=0.3)
        run = lambda: ctx.forward_lighting(gb, pf, pv, out=out, out_fmt=abi.FMT_RGBA16F, extra_point=extra, env=e)  # noqa: E731
        for _ in range(300):
            run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200):
            run()
        b.record(); b.synchronize()
        res[name] = round(a.elapsed_time(b) / 200, 4)
    res["unit"] = "ms per launch"
    print(json.dumps(res))


if __name__ == "__main__":
    main()
