import json, os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from vqengine_amd import abi, capi, synth
"""Shade kernel at three workgroup sizes (VQHIP_SHADE_WG = 256 / 128 / 64): cfg2 (1080p, 16 lights), cfg3 without and with IBL. One JSON line each."""
ctx = capi.Context(0)
pre, lut = bench.build_ibl(ctx)
env3 = capi.make_envmap(pre["diffuse_blurred"], pre["specular"], 128, pre["spec_mips"], lut)
for name in ("cfg2", "cfg3n", "cfg3"):
    cfg = dict(bench.CONFIGS["cfg2" if name == "cfg2" else "cfg3"])
    W, H, L = cfg["width"], cfg["height"], cfg["lights"]
    gb = bench.upload_tile(cfg, H, 0, H)
    out = capi.empty_image(H, W, abi.FMT_RGBA16F, ctx.device)
    env = env3 if name == "cfg3" else None
    pv = synth.per_view(W, H, max_env_lod=pre["spec_mips"] if env is not None else 0)
    pf, extra = synth.per_frame(points=synth.point_lights(L, seed=cfg["light_seed"]), hdri_offset=0.3 if env is not None else 0.0)
    run = lambda: ctx.forward_lighting(gb, pf, pv, out=out, out_fmt=abi.FMT_RGBA16F, extra_point=extra, env=env)
    for wg in ("256", "128", "64", "256", "128", "64", "256", "64"):
        ctx.set_option_env("VQHIP_SHADE_WG", wg)
        for _ in range(400): run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(300): run()
        b.record(); b.synchronize()
        print(json.dumps({"cfg": name, "lights": L, "wg": wg, "ms": round(a.elapsed_time(b) / 300, 4)}), flush=True)
