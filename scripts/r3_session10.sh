#!/bin/bash
# Round-3 GPU session 10: what binds the diffuse convolution — counters of the fast tap, block shapes (patch + lockstep).
O=gpurun_out/r3i; mkdir -p $O
python - > $O/diffuse_shapes.jsonl 2> $O/diffuse_shapes.err <<'PY'
import json, os, sys, torch
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
from vqengine_amd import abi, capi, synth
from bench_ibl_forms import timed
ctx = capi.Context(0)
eq = torch.from_numpy(synth.equirect(2048, 2048)).cuda()
chain, n = ctx.mip_chain(eq)
x = torch.empty(64 << 20, device="cuda")
for _ in range(200): x.mul_(1.0001)
ref = None
for wpb, sync in ((4, 0), (8, 0), (8, 1), (16, 0), (16, 1), (4, 0)):
    os.environ["VQHIP_DIFFUSE_WPB"] = str(wpb); os.environ["VQHIP_DIFFUSE_SYNC"] = str(sync)
    ms, out = timed(lambda: ctx.conv_diffuse(chain, 2048, 2048, n, 64, 0.010, abi.CONV_WAVE64, abi.FMT_RGBA16F))
    if ref is None: ref = out.clone()
    print(json.dumps({"what": "conv_diffuse fast", "wpb": wpb, "sync": sync, "ms": round(ms, 4), "identical": bool(torch.equal(ref.view(torch.uint8), out.view(torch.uint8)))}), flush=True)
PY
cat $O/diffuse_shapes.jsonl; tail -3 $O/diffuse_shapes.err
bash scripts/pmc_conv.sh fast > $O/pmc_conv_fast.txt 2>&1; cat $O/pmc_conv_fast.txt
