#!/usr/bin/env python3
"""PSMain from interpolants + materials at 3840x2160 as a TWO-STREAM band pipeline of the two kernels (round 5): the producer of band k+1 (texture latency, ~120 VGPRs)
runs on a second stream while the light loop of band k (VALU-bound, 69 VGPRs) runs on the first — two kernels may have two register counts, one kernel cannot.
Compared, interleaved on one box, with the two calls in sequence and with the fused kernel. Prints one JSON line per band count; `identical` = same bits as the two calls."""
import ctypes as C
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
    datas, texsets = synth.material_set(NM, max_dim=1024, same_size=False)
    dmats = (abi.MaterialDesc * NM)()
    keep = []
    for i, (d, ts) in enumerate(zip(datas, texsets)):
        dmats[i].data = d
        for slot, img in ts.items():
            chain_g, nm = ctx.mip_chain_rgba8(dev(img))
            keep.append(chain_g)
            setattr(dmats[i], slot, abi.Texture2D(chain_g.data_ptr(), img.shape[1], img.shape[0], nm, 0))
    gb = tuple(torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(4))
    out2 = capi.empty_image(H, W, abi.FMT_RGBA16F, ctx.device)
    outb = capi.empty_image(H, W, abi.FMT_RGBA16F, ctx.device)
    outf = capi.empty_image(H, W, abi.FMT_RGBA16F, ctx.device)
    s0 = torch.cuda.current_stream()
    s1 = torch.cuda.Stream()

    def two():
        ctx.gbuffer_from_materials(ipd, dmats, pf.fAmbientLightingFactor, ssao, out=gb)
        ctx.forward_lighting(gb, pf, pv, out=out2, out_fmt=abi.FMT_RGBA16F, extra_point=extra, env=env)

    def fused():
        ctx.forward_lighting_from_materials(ipd, dmats, pf, pv, ssao=ssao, out=outf, out_fmt=abi.FMT_RGBA16F, extra_point=extra, env=env)

    def banded(nb):
        rows = [(H * k // nb) & ~1 for k in range(nb)] + [H]
        evs = [torch.cuda.Event() for _ in range(nb)]
        s1.wait_stream(s0)
        for k in range(nb):                                  # producers, in order, on the second stream
            a, b = rows[k], rows[k + 1]
            ctx.gbuffer_from_materials([p[a:b] for p in ipd], dmats, pf.fAmbientLightingFactor, ssao[a:b], out=tuple(g[a:b] for g in gb), stream=s1)
            evs[k].record(s1)
        for k in range(nb):                                  # light loops on the first stream, each behind its band's producer
            a, b = rows[k], rows[k + 1]
            s0.wait_event(evs[k])
            ctx.forward_lighting(tuple(g[a:b] for g in gb), pf, pv, out=outb[a:b], out_fmt=abi.FMT_RGBA16F, extra_point=extra, env=env, stream=s0)

    def run(fn, n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(n):
            fn()
        b.record(); b.synchronize()
        return a.elapsed_time(b) / n
    run(two, 150)
    for nb in [int(x) for x in os.environ.get("VQ_BANDS", "2,4,8,16").split(",")]:
        t2, tf, tb = [], [], []
        for _ in range(5):
            t2.append(round(run(two, 40), 4)); tf.append(round(run(fused, 40), 4)); tb.append(round(run(lambda: banded(nb), 40), 4))
        print(json.dumps({"bands": nb, "two_calls_ms": float(np.median(t2)), "fused_ms": float(np.median(tf)), "banded_two_streams_ms": float(np.median(tb)),
                          "banded_rounds": tb, "identical": bool(torch.equal(out2.view(torch.int16), outb.view(torch.int16)))}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
