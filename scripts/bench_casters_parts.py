#!/usr/bin/env python3
"""Where the engine_max shade launch spends its time: the same frame with light classes switched off one by one (benchlib/casters.py; run on the GPU box).
usage: python scripts/bench_casters_parts.py [out.jsonl]"""
import copy
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from benchlib import casters  # noqa: E402
from vqengine_amd import abi, capi  # noqa: E402

ctx = capi.Context(0)
COH = os.environ.get("VQ_COHERENT") == "1"
gb, pf, pv, sm, keep, _ = casters.device_inputs("engine_max", coherent=COH)
print("content:", "coherent" if COH else "noise")
img = torch.empty((2160, 3840, 4), dtype=torch.float16, device="cuda")
rows = []


def run(tag, **kw):
    p = abi.PerFrameData.from_buffer_copy(bytes(pf))
    L = p.Lights
    for k, v in kw.items():
        if k == "dir_shadowing":
            L.directional.shadowing = v
        elif k == "dir_enabled":
            L.directional.enabled = v
        else:
            setattr(L, k, v)
    st = bench._stage_stats(lambda: ctx.forward_lighting(gb, p, pv, out=img, out_fmt=abi.FMT_RGBA16F, shadow=sm), spin_s=0.1, batches=5)
    rows.append({"content": "coherent" if COH else "noise", "case": tag, "ms": round(st["ms"], 4), "lights": casters.light_counts(p)})
    print(rows[-1], flush=True)


off = dict(numPointLights=0, numSpotLights=0, numPointCasters=0, numSpotCasters=0, dir_shadowing=0, dir_enabled=0)
run("nothing (ambient only, caster kernel)", **{**off, "dir_enabled": 1, "dir_shadowing": 1, "numSpotCasters": 0})
run("directional, no shadow", **{**off, "dir_enabled": 1})
run("directional + PCF 2048^2", **{**off, "dir_enabled": 1, "dir_shadowing": 1})
run("100 point lights", **{**off, "numPointLights": 100, "dir_enabled": 1, "dir_shadowing": 1})
run("20 spot lights", **{**off, "numSpotLights": 20, "dir_enabled": 1, "dir_shadowing": 1})
run("5 spot casters", **{**off, "numSpotCasters": 5, "dir_enabled": 1, "dir_shadowing": 1})
run("5 point casters", **{**off, "numPointCasters": 5, "dir_enabled": 1, "dir_shadowing": 1})
run("1 point caster", **{**off, "numPointCasters": 1, "dir_enabled": 1, "dir_shadowing": 1})
run("everything", )
if len(sys.argv) > 1:
    with open(sys.argv[1], "a") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
