#!/usr/bin/env python3
"""Calibration workload for the HBM PMC counters (run under `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace`):
the 4K shade kernel once WITHOUT IBL (pure 16 B/lane streaming: known 530.8 MB read, 66.4 MB written) and once WITH IBL
(adds the cube/LUT gathers), plus a plain torch copy of 512 MiB as an external yardstick."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vqengine_amd import abi, capi, synth
ctx = capi.Context(0)
pre, lut = bench.build_ibl(ctx)
env = capi.make_envmap(pre["diffuse_blurred"], pre["specular"], 128, pre["spec_mips"], lut)
pf, extra = synth.per_frame(points=synth.point_lights(64, seed=0x6400), hdri_offset=0.3)
cfg = bench.CONFIGS["cfg3"]
pv = synth.per_view(cfg["width"], cfg["height"], max_env_lod=pre["spec_mips"])
gb = bench.upload_tile(cfg, cfg["height"], 0, cfg["height"])
out = capi.empty_image(cfg["height"], cfg["width"], abi.FMT_RGBA16F, ctx.device)
for _ in range(3):
    ctx.forward_lighting(gb, pf, pv, out=out, out_fmt=abi.FMT_RGBA16F)              # k_forward_lighting<false,false,1>
    ctx.forward_lighting(gb, pf, pv, out=out, out_fmt=abi.FMT_RGBA16F, env=env)     # k_forward_lighting<true,false,1>
a = torch.empty(512 << 20, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
