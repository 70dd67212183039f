#!/usr/bin/env python3
"""Round 6: the post chain at 4K on ONE box — k_post_chain (one kernel) against the two-kernel chain and its two kernels alone, 3840x2160 RGBA16F -> RGBA8 (bench.py's stage
timer); identical bytes asserted. The forms of k_post_chain's mads that profiles/r6h_post_forms.jsonl compares lived behind a temporary option (post_mads) that is gone.
usage: python scripts/bench_post6.py [out.jsonl]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vqengine_amd import abi, capi, synth  # noqa: E402

F16, R8 = abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM
ctx = capi.Context(0)
W, H = 3840, 2160
img = torch.from_numpy(synth.hdr_image(W, 540)).cuda().to(torch.float16).repeat(4, 1, 1).contiguous()
out = capi.empty_image(H, W, R8, ctx.device)
xb = capi.empty_image(H, W, F16, ctx.device)
ctx.set_option("post_form", "two")
want = ctx.post_process_tile(img, F16, R8)
ctx.set_option("post_form", "chain")
got = ctx.post_process_tile(img, F16, R8)
torch.cuda.synchronize()
assert torch.equal(got, want), "k_post_chain: bytes differ from the two-kernel chain"
rows = []
for rep in range(3):
    ctx.set_option("post_form", "chain")
    st = bench._stage_stats(lambda: ctx.post_process_tile(img, F16, R8, out=out))
    rows.append({"form": "k_post_chain", "rep": rep, "us": round(st["ms"] * 1e3, 2)})
    ctx.set_option("post_form", "two")
    st = bench._stage_stats(lambda: ctx.post_process_tile(img, F16, R8, out=out))
    rows.append({"form": "two kernels", "rep": rep, "us": round(st["ms"] * 1e3, 2)})
    sx = bench._stage_stats(lambda: ctx.gaussian_blur_x(img, F16, out=xb))
    sy = bench._stage_stats(lambda: ctx.gaussian_blur_y_tonemap(xb, F16, R8, out=out))
    rows.append({"form": "k_blur_x4 / k_blur_y_tonemap_lut alone", "rep": rep, "us": [round(sx["ms"] * 1e3, 2), round(sy["ms"] * 1e3, 2)]})
for r in rows:
    print(json.dumps(r))
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
