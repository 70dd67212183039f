import sys, time, json
sys.path.insert(0, '/root/repo')
import torch
from vqengine_amd import capi, synth
ctx = capi.Context(0)
rgbe = synth.float_to_rgbe(synth.equirect(2048, 256)[..., :3])
part = synth.hdr_file_bytes(rgbe)
body = part[part.index(b"+X 2048\n") + 8:]
data = part[:part.index(b"-Y ")] + b"-Y 2048 +X 2048\n" + body * 8
for _ in range(3): ctx.load_hdr(data)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): ctx.load_hdr(data)
torch.cuda.synchronize()
print(json.dumps({"hdr_decode_2048_ms": (time.perf_counter() - t0) / 10 * 1e3, "file_bytes": len(data)}))
