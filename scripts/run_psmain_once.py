#!/usr/bin/env python3
"""A few launches of PSMain's three forms at 4K (for rocprofv3 counter passes, scripts/pmc_psmain.sh): the producer alone, the lighting kernel alone on the
producer's G-buffer, PSMain as one kernel (option psmain_waves = VQ_PSMAIN_WAVES when set) — the workloads of bench.py's `widened` object. VQ_REPS launches each."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vqengine_amd import abi, capi, synth  # noqa: E402

W, H, NM, BAND = 3840, 2160, 12, 540
F16 = abi.FMT_RGBA16F
ctx = capi.Context(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
tile = lambda a: np.tile(a, (H // BAND,) + (1,) * (a.ndim - 1))   # noqa: E731
ipd = [dev(tile(p)) for p in synth.interpolants(W, BAND, NM)]
ssao = dev(tile(synth.ssao_image(W, BAND)))
datas, texsets = synth.material_set(NM, max_dim=1024, same_size=False)
dm, keep = (abi.MaterialDesc * NM)(), []
for i, (dd, ts) in enumerate(zip(datas, texsets)):
    dm[i].data = dd
    for slot, img in ts.items():
        chain, nm = ctx.mip_chain_rgba8(dev(img))
        keep.append(chain)
        setattr(dm[i], slot, abi.Texture2D(chain.data_ptr(), img.shape[1], img.shape[0], nm, 0))
pre, lut = bench.build_ibl(ctx)
env, spec_mips = capi.make_envmap(pre["diffuse_blurred"], pre["specular"], 128, pre["spec_mips"], lut), pre["spec_mips"]
pf, extra = synth.per_frame(points=synth.point_lights(64, seed=0x6400), hdri_offset=0.3)
pv = synth.per_view(W, H, max_env_lod=spec_mips)
gb = tuple(torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(4))
scene = capi.empty_image(H, W, F16, ctx.device)
if os.environ.get("VQ_PSMAIN_WAVES"):
    ctx.set_option("psmain_waves", os.environ["VQ_PSMAIN_WAVES"])
for k, v in [kv.split("=") for kv in os.environ.get("VQ_OPTIONS", "").split(",") if kv]:
    ctx.set_option(k, v)
for _ in range(int(os.environ.get("VQ_REPS", "3"))):
    ctx.gbuffer_from_materials(ipd, dm, 0.055, ssao, out=gb)
    ctx.forward_lighting(gb, pf, pv, out=scene, out_fmt=F16, extra_point=extra, env=env)
    ctx.forward_lighting_from_materials(ipd, dm, pf, pv, ssao=ssao, out=scene, out_fmt=F16, extra_point=extra, env=env)
torch.cuda.synchronize()
