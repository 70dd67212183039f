#!/usr/bin/env python3
"""A/B of the 4K post chain (vqhip_post_process_tile: k_post_chain) between two builds of libvqhip.so IN ONE PROCESS (same box, same clocks, interleaved rounds):
usage: ab_post.py <other.so>[,<other2.so>...]. Prints one JSON line with the per-round times of all libraries and whether the bytes agree."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqengine_amd import abi, capi, synth  # noqa: E402


def main():
    others = [os.path.abspath(o) for o in sys.argv[1].split(",")]
    W, H = 3840, 2160
    ctx_a = capi.Context(0)
    ctx_o = []
    for other in others:
        capi._lib, capi._LIB_PATH = None, other              # another binding: the other build (RTLD_LOCAL keeps them apart)
        ctx_o.append(capi.Context(0))
    band = synth.hdr_image(W, 270, seed=7).astype(np.float16)
    img = torch.from_numpy(np.tile(band, (8, 1, 1))).cuda()
    outs = [capi.empty_image(H, W, abi.FMT_RGBA8_UNORM, ctx_a.device) for _ in range(1 + len(others))]

    def run(ctx, out, n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            ctx.post_process_tile(img, abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM, out=out)
        b.record(); b.synchronize()
        return a.elapsed_time(b) / n
    run(ctx_a, outs[0], 4000)
    ta, tb = [], [[] for _ in others]
    for r in range(6):
        ta.append(round(run(ctx_a, outs[0], 600) * 1e3, 2))
        for k, c in enumerate(ctx_o):
            tb[k].append(round(run(c, outs[k + 1], 600) * 1e3, 2))
    same = [bool(torch.equal(outs[0], o)) for o in outs[1:]]
    print(json.dumps({"current_us": ta, "current_median": float(np.median(ta)),
                      "others": {os.path.basename(o): {"us": tb[k], "median": float(np.median(tb[k])), "same_bytes": same[k]} for k, o in enumerate(others)}}), flush=True)


if __name__ == "__main__":
    main()
