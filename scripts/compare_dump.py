#!/usr/bin/env python3
"""Diff a scene-colour dump of the ENGINE (VQEngine on D3D12 — WARP or any adapter; recipe: docs/WARP_CALIBRATION.md) against this
library on the same inputs, in storage-format ulps. This is the off-box half of the north star's tolerance statement
("<= 1 ULP per channel vs the reference's D3D12-WARP render on identical G-buffer inputs"), which cannot run in the build container.

Inputs (raw little-endian files written by the patched engine, all of one frame):
  --gbuffer gb0.bin gb1.bin gb2.bin gb3.bin   RGBA32F planes, W*H*16 bytes each, dense rows: the state of PSMain at
                                              ForwardLighting.hlsl:284-293 = (P, ao) (N, roughness) (albedo, metalness) (emissive, intensity)
  --per-frame perframe.bin                    the 7120 bytes of cbuffer b1 (PerFrameData)
  --per-view perview.bin                      the 320 bytes of cbuffer b0 (PerViewLightingData)
  --scene scene.bin                           Tex_SceneColor as stored: RGBA16F, W*H*8 bytes (row padding already stripped)
  [--env env.npz]                             diffuse / specular cubes + BRDF LUT as arrays `diffuse` [6,R,R,4] f16, `specular` [px,4] f16 (mip-major),
                                              `lut` [S,S,2] f16, `spec_res0`, `spec_mips` — omit when the scene was rendered with the NullCubemap
  --size W H
  --backend hip|oracle                        hip: libvqhip.so on cuda:0 (the product); oracle: the CPU restatement (no GPU needed)
Output: one JSON object: channels compared, max ulps, histogram, the worst pixels. Exit code 1 when any channel is more than --max-ulps (1) away.
`--selftest` fabricates a dump from the oracle (plus a known 1-ulp perturbation) and runs the comparison on it."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vqengine_amd import abi  # noqa: E402


def key16(h):
    u = np.asarray(h).view(np.uint16).astype(np.int32)
    return np.where(u & 0x8000, -(u & 0x7fff), u)


def load_struct(path, cls):
    raw = open(path, "rb").read()
    if len(raw) < abi.C.sizeof(cls):
        raise SystemExit(f"{path}: {len(raw)} bytes, {cls.__name__} needs {abi.C.sizeof(cls)}")
    return cls.from_buffer_copy(raw[:abi.C.sizeof(cls)])


def shade(backend, gb, pf, pv, env_npz):
    if backend == "oracle":
        from tests import oracle_lib as O
        env = O.host_envmap(env_npz["diffuse"], env_npz["specular"], int(env_npz["spec_res0"]), int(env_npz["spec_mips"]), env_npz["lut"]) if env_npz is not None else None
        return O.forward_lighting(gb, pf, pv, abi.FMT_RGBA16F, env=env)
    import torch
    from vqengine_amd import capi
    ctx = capi.Context(0)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()      # noqa: E731
    keep, env = [], None
    if env_npz is not None:
        keep = [dev(env_npz["diffuse"]), dev(env_npz["specular"]), dev(env_npz["lut"])]
        env = capi.make_envmap(keep[0], keep[1], int(env_npz["spec_res0"]), int(env_npz["spec_mips"]), keep[2])
    out = ctx.forward_lighting([dev(g) for g in gb], pf, pv, out_fmt=abi.FMT_RGBA16F, env=env).cpu().numpy()
    ctx.close()
    return out


def compare(ours, theirs, max_ulps):
    a, b = np.asarray(ours, np.float16), np.asarray(theirs, np.float16)
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    d = np.abs(key16(a) - key16(b))
    d[nan_a & nan_b] = 0                                     # NaN == NaN whatever the payload
    d[nan_a ^ nan_b] = 1 << 16
    hist = {str(k): int((d == k).sum()) for k in range(0, 5)}
    hist[">4"] = int((d > 4).sum())
    worst = np.argsort(d, axis=None)[::-1][:8]
    rows = [{"y": int(i // (a.shape[1] * a.shape[2])), "x": int(i // a.shape[2] % a.shape[1]), "channel": int(i % a.shape[2]), "ulps": int(d.flat[i]),
             "ours": float(a.flat[i]), "engine": float(b.flat[i])} for i in worst if d.flat[i] > 0]
    rep = {"channels": int(d.size), "max_ulps": int(d.max()), "differing_fraction": float(np.mean(d > 0)), "above_limit_fraction": float(np.mean(d > max_ulps)),
           "nan_mismatches": int((nan_a ^ nan_b).sum()), "histogram_ulps": hist, "worst": rows, "limit_ulps": max_ulps, "pass": bool(d.max() <= max_ulps)}
    return rep


def selftest(tmp):
    from tests import oracle_lib as O
    from vqengine_amd import synth
    W, H = 96, 16
    gb = synth.gbuffer(W, H, seed=7)
    pf, _ = synth.per_frame(points=synth.point_lights(8, seed=7), spots=synth.spot_lights(2), directional=synth.directional_light())
    pv = synth.per_view(W, H)
    scene = O.forward_lighting(gb, pf, pv, abi.FMT_RGBA16F)
    scene.view(np.uint16)[3, 5, 1] += 1                      # a known 1-ulp difference
    for k in range(4):
        gb[k].tofile(os.path.join(tmp, f"gb{k}.bin"))
    open(os.path.join(tmp, "perframe.bin"), "wb").write(bytes(pf))
    open(os.path.join(tmp, "perview.bin"), "wb").write(bytes(pv))
    scene.tofile(os.path.join(tmp, "scene.bin"))
    return W, H


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gbuffer", nargs=4)
    ap.add_argument("--per-frame"); ap.add_argument("--per-view"); ap.add_argument("--scene"); ap.add_argument("--env")
    ap.add_argument("--size", nargs=2, type=int)
    ap.add_argument("--backend", choices=["hip", "oracle"], default="hip")
    ap.add_argument("--max-ulps", type=int, default=1)
    ap.add_argument("--selftest", action="store_true")
    a = ap.parse_args()
    if a.selftest:
        import tempfile
        tmp = tempfile.mkdtemp()
        W, H = selftest(tmp)
        a.gbuffer = [os.path.join(tmp, f"gb{k}.bin") for k in range(4)]
        a.per_frame, a.per_view, a.scene, a.size, a.backend = os.path.join(tmp, "perframe.bin"), os.path.join(tmp, "perview.bin"), os.path.join(tmp, "scene.bin"), [W, H], "oracle"
    if not (a.gbuffer and a.per_frame and a.per_view and a.scene and a.size):
        ap.error("--gbuffer, --per-frame, --per-view, --scene and --size are required")
    W, H = a.size
    gb = []
    for p in a.gbuffer:
        g = np.fromfile(p, np.float32)
        if g.size != W * H * 4:
            raise SystemExit(f"{p}: {g.size * 4} bytes, expected {W * H * 16} (RGBA32F, dense rows)")
        gb.append(g.reshape(H, W, 4))
    scene = np.fromfile(a.scene, np.float16)
    if scene.size != W * H * 4:
        raise SystemExit(f"{a.scene}: {scene.size * 2} bytes, expected {W * H * 8} (RGBA16F, row padding stripped)")
    pf, pv = load_struct(a.per_frame, abi.PerFrameData), load_struct(a.per_view, abi.PerViewLightingData)
    env = np.load(a.env) if a.env else None
    ours = shade(a.backend, gb, pf, pv, env)
    rep = compare(ours, scene.reshape(H, W, 4), a.max_ulps)
    rep["backend"] = a.backend
    print(json.dumps(rep, indent=1))
    if a.selftest:
        assert rep["max_ulps"] == 1 and rep["histogram_ulps"]["1"] == 1 and rep["worst"][0]["y"] == 3 and rep["worst"][0]["x"] == 5, rep
        print("selftest OK")
    sys.exit(0 if rep["pass"] else 1)


if __name__ == "__main__":
    main()
