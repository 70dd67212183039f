#!/usr/bin/env python3
"""Round 6: forms of k_post_chain's mads on ONE box (option post_mads), 3840x2160 RGBA16F -> RGBA8, alone (bench.py's stage timer); identical bytes asserted.
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
out = [capi.empty_image(H, W, R8, ctx.device) for _ in range(3)]
ctx.set_option("post_form", "two")
want = ctx.post_process_tile(img, F16, R8)
ctx.set_option("post_form", "chain")
rows = []
for rep in range(3):
    for mads in (0, 1, 2):
        ctx.set_option("post_mads", mads)
        got = ctx.post_process_tile(img, F16, R8, out=out[mads])
        torch.cuda.synchronize()
        assert torch.equal(got, want), f"post_mads={mads}: bytes differ from the two-kernel chain"
        st = bench._stage_stats(lambda: ctx.post_process_tile(img, F16, R8, out=out[mads]))
        rows.append({"post_mads": mads, "rep": rep, "us": round(st["ms"] * 1e3, 2), "us_min": round(st["ms_min"] * 1e3, 2), "us_max": round(st["ms_max"] * 1e3, 2)})
        print(json.dumps(rows[-1]), flush=True)
ctx.set_option("post_form", "two")
st = bench._stage_stats(lambda: ctx.post_process_tile(img, F16, R8, out=out[0]))
rows.append({"form": "two kernels", "us": round(st["ms"] * 1e3, 2)})
print(json.dumps(rows[-1]))
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
