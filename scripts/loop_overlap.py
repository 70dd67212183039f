#!/usr/bin/env python3
"""Does running the (HBM-bound) post chain of frame i on a second stream beside the (VALU-bound) shade kernel of frame i+1 pay? One box; per-frame time of the serial
loop and of the two-stream loop (same kernels, same buffers, double-buffered scene / sdr)."""
import ctypes as C, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vqengine_amd import abi, capi, synth
cfg = bench.CONFIGS["cfg3"]; ctx = capi.Context(0); W, H, L = cfg["width"], cfg["height"], cfg["lights"]
pre, lut = bench.build_ibl(ctx); env = capi.make_envmap(pre["diffuse_blurred"], pre["specular"], 128, pre["spec_mips"], lut)
gb = bench.upload_tile(cfg, H, 0, H); F16, R8 = abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM
scene = [capi.empty_image(H, W, F16, ctx.device) for _ in range(2)]; xb = capi.empty_image(H, W, F16, ctx.device); sdr = [capi.empty_image(H, W, R8, ctx.device) for _ in range(2)]
pv = synth.per_view(W, H, max_env_lod=pre["spec_mips"]); pf, extra = synth.per_frame(points=synth.point_lights(L, seed=cfg["seed"]), hdri_offset=0.3)
s_main = torch.cuda.current_stream(ctx.device); s_post = torch.cuda.Stream(ctx.device)
def serial(i):
    b = i & 1
    ctx.forward_lighting(gb, pf, pv, out=scene[b], out_fmt=F16, extra_point=extra, env=env)
    ctx.gaussian_blur_x(scene[b], F16, out=xb); ctx.gaussian_blur_y_tonemap(xb, F16, R8, out=sdr[b])
e_shade = [None, None]; e_post = [None, None]
def overlapped(i):
    b = i & 1
    if e_post[b] is not None: s_main.wait_event(e_post[b])           # scene[b] was read by the X pass of frame i-2
    ctx.forward_lighting(gb, pf, pv, out=scene[b], out_fmt=F16, extra_point=extra, env=env)
    e_shade[b] = torch.cuda.Event(); e_shade[b].record(s_main)
    s_post.wait_event(e_shade[b])
    with torch.cuda.stream(s_post):
        ctx.gaussian_blur_x(scene[b], F16, out=xb); ctx.gaussian_blur_y_tonemap(xb, F16, R8, out=sdr[b])
        e_post[b] = torch.cuda.Event(); e_post[b].record(s_post)
def run(fn, n=200):
    for i in range(300): fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s_main)
    for i in range(n): fn(i)
    s_main.wait_stream(s_post)
    b.record(s_main); b.synchronize(); return round(a.elapsed_time(b) / n, 4)
res = {"serial": run(serial), "post on a second stream": run(overlapped), "serial again": run(serial)}
print(json.dumps(res))
