#!/usr/bin/env python3
"""A/B of the cfg3 shade kernel between two builds of libvqhip.so IN ONE PROCESS (same box, same clocks, interleaved rounds):
usage: ab_shade.py <other.so>[,<other2.so>...] [noise|coherent]. Prints one JSON line with the per-round times of all libraries."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vqengine_amd import abi, capi, synth  # noqa: E402


def main():
    others = [os.path.abspath(o) for o in sys.argv[1].split(",")]
    content = sys.argv[2] if len(sys.argv) > 2 else "noise"
    cfg = bench.CONFIGS["cfg3"]
    W, H, L = cfg["width"], cfg["height"], cfg["lights"]
    ctx_a = capi.Context(0)
    ctx_o = []
    for other in others:
        capi._lib, capi._LIB_PATH = None, other              # another binding: the other build (RTLD_LOCAL keeps them apart)
        ctx_o.append(capi.Context(0))
    pre, lut = bench.build_ibl(ctx_a)
    env = capi.make_envmap(pre["diffuse_blurred"], pre["specular"], 128, pre["spec_mips"], lut)
    gb = bench.upload_tile(cfg, H, 0, H, coherent=(content == "coherent"))
    pf, extra = synth.per_frame(points=synth.point_lights(L, seed=cfg["seed"]), hdri_offset=0.3)
    pv = synth.per_view(W, H, max_env_lod=pre["spec_mips"])
    out = capi.empty_image(H, W, abi.FMT_RGBA16F, ctx_a.device)

    def run(ctx, n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            ctx.forward_lighting(gb, pf, pv, out=out, out_fmt=abi.FMT_RGBA16F, extra_point=extra, env=env)
        b.record(); b.synchronize()
        return a.elapsed_time(b) / n
    run(ctx_a, 300)
    ta, tb = [], [[] for _ in others]
    for r in range(6):
        ta.append(round(run(ctx_a, 100), 4))
        for k, c in enumerate(ctx_o):
            tb[k].append(round(run(c, 100), 4))
    print(json.dumps({"content": content, "current_ms": ta, "current_median": float(np.median(ta)),
                      "others": {os.path.basename(o): {"ms": tb[k], "median": float(np.median(tb[k]))} for k, o in enumerate(others)}}), flush=True)


if __name__ == "__main__":
    main()
