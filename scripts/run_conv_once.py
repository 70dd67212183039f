#!/usr/bin/env python3
"""One launch of each load-time kernel of config 4 (for rocprofv3 counter passes): mips, diffuse, specular, LUT. VQ_CONV_REPS launches each,
in the order VQ_CONV_ORDER names (sequential = the reference's, default | wave64)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqengine_amd import abi, capi, synth  # noqa: E402

ctx = capi.Context(0)
eq = torch.from_numpy(synth.equirect(2048, 2048)).cuda()
chain, n = ctx.mip_chain(eq)
order = abi.CONV_WAVE64 if os.environ.get("VQ_CONV_ORDER", "sequential") == "wave64" else abi.CONV_SEQUENTIAL
for _ in range(int(os.environ.get("VQ_CONV_REPS", "2"))):
    ctx.conv_diffuse(chain, 2048, 2048, n, 64, 0.010, order, abi.FMT_RGBA16F)
    ctx.conv_specular(chain, 2048, 2048, n, 128, order, abi.FMT_RGBA16F)
    ctx.brdf_lut(1024, 2048, abi.FMT_RG16F)
torch.cuda.synchronize()
