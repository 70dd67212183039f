#!/usr/bin/env python3
"""The cfg3 shade kernel (3840x2160, 64 point lights + the full-size IBL -> RGBA16F) on three frames of the same size:
  noise    : the BASELINE workload (synth.gbuffer_rows: white-noise normals and materials, roughness in [0.05, 1])
  coherent : synth.gbuffer_rows_coherent (terrain normals, material regions, 12 % polished regions with roughness 0 .. 0.03)
  polished : the noise frame with every roughness scaled into [0, 0.04) (the GGX EPSILON early-out range, every wave)
Kernel time = HIP events around back-to-back launches after a spin-up. One JSON line per frame; `label` names the library under test."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vqengine_amd import abi, capi, synth  # noqa: E402


def main():
    label = sys.argv[1] if len(sys.argv) > 1 else "current"
    cfgname = sys.argv[2] if len(sys.argv) > 2 else "cfg3"
    cfg = bench.CONFIGS[cfgname]
    W, H, L = cfg["width"], cfg["height"], cfg["lights"]
    if cfgname == "cfg5":
        H = 1080                                             # a quarter of the 8K frame is enough for a kernel rate
    ctx = capi.Context(0)
    env = pre = None
    if cfg["env"]:
        pre, lut = bench.build_ibl(ctx)
        env = capi.make_envmap(pre["diffuse_blurred"], pre["specular"], 128, pre["spec_mips"], lut)
    pf, extra = synth.per_frame(points=synth.point_lights(L, seed=cfg["seed"]), hdri_offset=0.3 if cfg["env"] else 0.0)
    pv = synth.per_view(W, cfg["height"], max_env_lod=pre["spec_mips"] if pre else 0)
    out = capi.empty_image(H, W, abi.FMT_RGBA16F, ctx.device)

    def upload(gen):
        gb = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(4)]
        for r in range(0, H, 240):
            part = gen(W, cfg["height"], r, min(r + 240, H), seed=cfg["seed"])
            for k in range(4):
                gb[k][r:r + part[k].shape[0]].copy_(torch.from_numpy(part[k]))
        return gb

    def polished(W_, H_, r0, r1, seed):
        p = synth.gbuffer_rows(W_, H_, r0, r1, seed=seed)
        p[1][..., 3] = (p[1][..., 3] - 0.05) * np.float32(0.04 / 0.95)
        return p
    for name, gen in (("noise", synth.gbuffer_rows), ("coherent", synth.gbuffer_rows_coherent), ("polished", polished)):
        gb = upload(gen)
        fn = lambda: ctx.forward_lighting(gb, pf, pv, out=out, out_fmt=abi.FMT_RGBA16F, extra_point=extra, env=env)      # noqa: E731
        for _ in range(250):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 100 if cfgname == "cfg3" else 30
        a.record()
        for _ in range(n):
            fn()
        b.record(); b.synchronize()
        ms = a.elapsed_time(b) / n
        rough = gb[1][..., 3]
        print(json.dumps({"lib": label, "config": cfgname, "frame": name, "shade_ms": round(ms, 4), "Mpix_s": round(W * H / ms / 1e3, 1),
                          "roughness_below_0.04_fraction": round(float((rough < 0.04).float().mean().item()), 4),
                          "checksum": int(out.view(torch.int16).to(torch.int64).sum().item())}), flush=True)
        del gb
    ctx.close()


if __name__ == "__main__":
    main()
