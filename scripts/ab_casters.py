#!/usr/bin/env python3
"""A/B of the caster workloads (benchlib/casters.py: cfg1, engine_max; noise and coherent content) between two builds of libvqhip.so IN ONE PROCESS, interleaved rounds.
usage: ab_casters.py <other.so>"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchlib import casters  # noqa: E402
from vqengine_amd import abi, capi  # noqa: E402

other = os.path.abspath(sys.argv[1])
ctx_a = capi.Context(0)
capi._lib, capi._LIB_PATH = None, other
ctx_b = capi.Context(0)
for name in ("cfg1", "engine_max"):
    for coh in (False, True):
        gb, pf, pv, sm, keep, _ = casters.device_inputs(name, coherent=coh)
        w = casters.WORKLOADS[name]
        img = torch.empty((w["height"], w["width"], 4), dtype=torch.float16, device="cuda")
        n = 400 if name == "cfg1" else 40

        def run(ctx):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                ctx.forward_lighting(gb, pf, pv, out=img, out_fmt=abi.FMT_RGBA16F, shadow=sm)
            b.record(); b.synchronize()
            return a.elapsed_time(b) / n
        run(ctx_a); run(ctx_b)
        ta, tb = [], []
        for r in range(5):
            ta.append(run(ctx_a)); tb.append(run(ctx_b))
        print(json.dumps({"workload": name, "content": "coherent" if coh else "noise", "current_ms": round(float(np.median(ta)), 4), "other_ms": round(float(np.median(tb)), 4)}), flush=True)
        del gb, keep, img
