#!/usr/bin/env python3
"""PSMain as the engine has it (ForwardLighting.hlsl:226-380) at 3840x2160: 12 materials (up to 7 RGBA8 mip-chained maps each) + SSAO, 64 point
lights + the full-size IBL -> RGBA16F scene colour, two ways, interleaved rounds on one box:
  two calls : vqhip_gbuffer_from_materials (48 B in, 64 B out per pixel) then vqhip_forward_lighting (64 B in, 8 B out)
  fused     : vqhip_forward_lighting_from_materials (48 B in, 8 B out; the record stays in registers)
Prints one JSON line; `identical` = the two results have the same bits."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vqengine_amd import abi, capi, synth  # noqa: E402


def main():
    ctx = capi.Context(0)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    W, H, NM = 3840, 2160, 12
    pre, lut = bench.build_ibl(ctx)
    env = capi.make_envmap(pre["diffuse_blurred"], pre["specular"], 128, pre["spec_mips"], lut)
    pf, extra = synth.per_frame(points=synth.point_lights(64, seed=0x6400), hdri_offset=0.3)
    pv = synth.per_view(W, H, max_env_lod=pre["spec_mips"])
    ipd = [dev(p) for p in synth.interpolants(W, H, NM)]
    ssao = dev(synth.ssao_image(W, H))
    for same in (True, False):
        datas, texsets = synth.material_set(NM, max_dim=1024, same_size=same)
        dmats = (abi.MaterialDesc * NM)()
        keep, nmaps = [], 0
        for i, (d, ts) in enumerate(zip(datas, texsets)):
            dmats[i].data = d
            for slot, img in ts.items():
                chain_g, nm = ctx.mip_chain_rgba8(dev(img))
                keep.append(chain_g)
                setattr(dmats[i], slot, abi.Texture2D(chain_g.data_ptr(), img.shape[1], img.shape[0], nm, 0))
                nmaps += 1
        gb = tuple(torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(4))
        out2 = capi.empty_image(H, W, abi.FMT_RGBA16F, ctx.device)
        outf = capi.empty_image(H, W, abi.FMT_RGBA16F, ctx.device)

        def two():
            ctx.gbuffer_from_materials(ipd, dmats, pf.fAmbientLightingFactor, ssao, out=gb)
            ctx.forward_lighting(gb, pf, pv, out=out2, out_fmt=abi.FMT_RGBA16F, extra_point=extra, env=env)

        def fused():
            ctx.forward_lighting_from_materials(ipd, dmats, pf, pv, ssao=ssao, out=outf, out_fmt=abi.FMT_RGBA16F, extra_point=extra, env=env)

        def run(fn, n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                fn()
            b.record(); b.synchronize()
            return a.elapsed_time(b) / n
        run(two, 150)
        for waves in os.environ.get("VQ_PSMAIN_WAVES_SWEEP", "4").split(","):
            ctx.set_option_env("VQHIP_PSMAIN_WAVES", waves)
            t2, tf = [], []
            for _ in range(5):
                t2.append(round(run(two, 60), 4)); tf.append(round(run(fused, 60), 4))
            m2, mf = float(np.median(t2)), float(np.median(tf))
            print(json.dumps({"frame": [W, H], "materials": NM, "maps": nmaps, "sizes": "one per material" if same else "random per map", "lights": 64, "ibl": True,
                              "fused_min_waves_per_simd": int(waves),
                              "two_calls_ms": t2, "fused_ms": tf, "two_calls_median_ms": m2, "fused_median_ms": mf, "gain": round(1.0 - mf / m2, 4),
                              "identical": bool(torch.equal(out2.view(torch.int16), outf.view(torch.int16))),
                              "Mpix_s_two_calls": round(W * H / m2 / 1e3, 1), "Mpix_s_fused": round(W * H / mf / 1e3, 1)}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
