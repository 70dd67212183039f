#!/usr/bin/env python3
"""A few launches of the widened per-frame kernels at 4K (for rocprofv3 counter passes, scripts/pmc_widened.sh): G-buffer producer, Z pre-pass normals,
SSR environment fallback — the workloads of bench.py's `widened` object. VQ_REPS launches each (default 3)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vqengine_amd import abi, capi, synth  # noqa: E402

W, H, NM, BAND = 3840, 2160, 12, 540
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
gb = tuple(torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(4))
nrm = torch.empty((H, W), dtype=torch.int32, device="cuda")
sc, depth, packed, _ = synth.ssr_surfaces(W, BAND)
scd, dpd, nmd = dev(tile(sc.astype(np.float16))), dev(tile(depth)), dev(tile(packed.view(np.int32)))
cb = synth.ssr_constants(W, H, spec_mips)
rad = capi.empty_image(H, W, abi.FMT_RGBA16F, ctx.device)
for _ in range(int(os.environ.get("VQ_REPS", "3"))):
    ctx.gbuffer_from_materials(ipd, dm, 0.055, ssao, out=gb)
    ctx.scene_normals_from_materials(ipd, dm, out=nrm)
    ctx.ssr_environment_fallback(scd, abi.FMT_RGBA16F, dpd, nmd, abi.FMT_R10G10B10A2_UNORM, cb, env, abi.FMT_RGBA16F, out=rad)
torch.cuda.synchronize()
