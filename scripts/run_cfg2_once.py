#!/usr/bin/env python3
"""A few launches of the BASELINE cfg2 shade kernel (1920x1080, 16 point lights, no IBL) for rocprofv3 counter passes (scripts/pmc_refresh.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vqengine_amd import abi, capi, synth  # noqa: E402

ctx = capi.Context(0)
cfg = bench.CONFIGS["cfg2"]
W, H = cfg["width"], cfg["height"]
gb = bench.upload_tile(cfg, H, 0, H)
pf, extra = synth.per_frame(points=synth.point_lights(cfg["lights"], seed=cfg["light_seed"]))
pv = synth.per_view(W, H)
out = capi.empty_image(H, W, abi.FMT_RGBA16F, ctx.device)
for _ in range(int(os.environ.get("VQ_CFG2_REPS", "12"))):
    ctx.forward_lighting(gb, pf, pv, out=out, out_fmt=abi.FMT_RGBA16F, extra_point=extra)
torch.cuda.synchronize()
