#!/usr/bin/env python3
"""A few launches of the caster workloads' shade kernel (benchlib/casters.py) for rocprofv3 passes: python scripts/run_casters_once.py cfg1|engine_max [launches]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchlib import casters  # noqa: E402
from vqengine_amd import abi, capi  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ctx = capi.Context(0)
gb, pf, pv, sm, keep, _ = casters.device_inputs(name)
w = casters.WORKLOADS[name]
img = torch.empty((w["height"], w["width"], 4), dtype=torch.float16, device="cuda")
for _ in range(n):
    ctx.forward_lighting(gb, pf, pv, out=img, out_fmt=abi.FMT_RGBA16F, shadow=sm)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(n):
    ctx.forward_lighting(gb, pf, pv, out=img, out_fmt=abi.FMT_RGBA16F, shadow=sm)
b.record(); b.synchronize()
print(name, "ms per launch", a.elapsed_time(b) / n)
ctx.close()
