#!/usr/bin/env python3
"""A few launches of the post kernels at 4K (for rocprofv3 counter passes, scripts/pmc_post.sh): the X pass, the fused Y + tonemap pass, and the one-kernel chain
(k_post_chain) on a warmed-up chip. VQ_REPS launches each (default 5)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqengine_amd import abi, capi, synth  # noqa: E402

W, H = 3840, 2160
F16, R8 = abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM
ctx = capi.Context(0)
band = synth.hdr_image(W, 270, scale=8.0).astype(np.float16)
scene = torch.from_numpy(np.tile(band, (H // 270, 1, 1)).copy()).cuda()
xb = capi.empty_image(H, W, F16, ctx.device)
sdr = capi.empty_image(H, W, R8, ctx.device)
for _ in range(int(os.environ.get("VQ_SPIN", "200"))):     # spin-up
    ctx.gaussian_blur_x(scene, F16, out=xb)
for _ in range(int(os.environ.get("VQ_REPS", "5"))):
    ctx.gaussian_blur_x(scene, F16, out=xb)
    ctx.gaussian_blur_y_tonemap(xb, F16, R8, out=sdr)
    ctx.post_process(scene, F16, R8, out=sdr)                # default for a 4K frame: k_post_chain
torch.cuda.synchronize()
