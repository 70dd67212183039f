#!/usr/bin/env python3
"""How much slower the cfg3 shade kernel runs when the post kernels sit between its launches (one box): per-iteration time of loops of
shade alone, post alone, shade + X, shade + Y, shade + X + Y, and the shade kernel's own event interval inside the full loop."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vqengine_amd import abi, capi, synth
cfg = bench.CONFIGS["cfg3"]; ctx = capi.Context(0); W, H, L = cfg["width"], cfg["height"], cfg["lights"]
pre, lut = bench.build_ibl(ctx); env = capi.make_envmap(pre["diffuse_blurred"], pre["specular"], 128, pre["spec_mips"], lut)
gb = bench.upload_tile(cfg, H, 0, H); F16, R8 = abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM
scene = [capi.empty_image(H, W, F16, ctx.device) for _ in range(2)]; xb = capi.empty_image(H, W, F16, ctx.device); sdr = [capi.empty_image(H, W, R8, ctx.device) for _ in range(2)]
zero = torch.zeros((H, W, 4), dtype=torch.float16, device=ctx.device)
pv = synth.per_view(W, H, max_env_lod=pre["spec_mips"]); pf, extra = synth.per_frame(points=synth.point_lights(L, seed=cfg["seed"]), hdri_offset=0.3)
def shade(b): ctx.forward_lighting(gb, pf, pv, out=scene[b], out_fmt=F16, extra_point=extra, env=env)
def X(src): ctx.gaussian_blur_x(src, F16, out=xb)
def Y(b): ctx.gaussian_blur_y_tonemap(xb, F16, R8, out=sdr[b])
def run(fn, n=200):
    for i in range(300): fn(i)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(i)
    b.record(); b.synchronize(); return round(a.elapsed_time(b) / n, 4)
evs = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(600)]
def full_ev(i): evs[i % 600][0].record(); shade(i & 1); evs[i % 600][1].record(); X(scene[i & 1]); Y(i & 1)
res = {"shade": run(lambda i: shade(i & 1)), "x": run(lambda i: X(scene[i & 1])), "y": run(lambda i: Y(i & 1)),
       "shade+x": run(lambda i: (shade(i & 1), X(scene[i & 1]))), "shade+x(zeros)": run(lambda i: (shade(i & 1), X(zero))),
       "shade+y": run(lambda i: (shade(i & 1), Y(i & 1))), "shade+x+y": run(lambda i: (shade(i & 1), X(scene[i & 1]), Y(i & 1))), "shade+x+y with 2 events": run(full_ev)}
res["shade interval inside the full loop"] = round(sum(e[0].elapsed_time(e[1]) for e in evs[:200]) / 200, 4)
res["penalty shade+x+y"] = round(res["shade+x+y"] - res["shade"] - res["x"] - res["y"], 4)
print(json.dumps(res))
