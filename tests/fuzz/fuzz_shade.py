#!/usr/bin/env python3
"""Randomised parity of the headline path's shade kernel (point lights + extension array + spot lights + directional light + IBL sample), no casters:
python tests/fuzz/fuzz_shade.py [--seconds 120] [--seed 1] (needs a GPU; the oracle is the checker).

Every case draws the frame size, white-noise or surface-coherent content, light counts up to the cbuffer's 100 + an extension array, lights AT pixels, on an axis above a pixel,
with -0.0 coordinates, zero / huge / non-finite ranges and colours, the camera (also AT a pixel), environment cubes of random sizes filled with random fp16 texels (a few of them
non-finite), MaxEnvMapLODLevels (matching the cube, beyond it, fractional, negative, NaN), the diffuse-only switch, the HDRI yaw, special values and magnitudes in scattered lanes,
the output format, the reading of dot / normalize and the Fresnel power. The HIP product's bits must equal the oracle's (any NaN == any NaN). Exit status 1 if a case differed."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))          # the fuzzers import each other

from tests import oracle_lib as O  # noqa: E402
from vqengine_amd import abi, synth  # noqa: E402

F32, F16 = abi.FMT_RGBA32F, abi.FMT_RGBA16F
SPECIALS = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-30, -1e-30, 1e25, -1e25, 3e19, 1e-45, 0.5, 2.0, 0.04, 0.03999, 1.0000001, 65504.0, 7e4], np.float32)


def rand_f16(r, shape, special_rate):
    a = (r.random(shape, dtype=np.float32) * np.float32(r.choice([1.0, 8.0, 300.0]))).astype(np.float16)
    if special_rate > 0:
        flat = a.reshape(-1)
        n = max(1, int(flat.size * special_rate))
        flat[r.integers(0, flat.size, n)] = r.choice(np.array([np.inf, -np.inf, np.nan, 0.0, -0.0, 65504.0, 6e-8, -1.0], np.float16), n)
    return a


def case(seed):
    r = np.random.Generator(np.random.Philox(key=[int(seed), 0xF1]))
    W = int(r.choice([64, 100, 128, 192, 256, 321, 512, 1000]))
    H = int(r.integers(1, 6))
    coherent = bool(r.integers(0, 2))
    gb = [np.array(g, copy=True) for g in synth.gbuffer(W, H, seed=seed & 0xFFFF, coherent=coherent)]
    n_pts = int(r.choice([0, 1, 2, 5, 16, 64, 100, 103, 256]))
    pts = synth.point_lights(n_pts, seed=seed & 0xFFFFF) if n_pts else None
    spots = synth.spot_lights(int(r.choice([0, 0, 1, 3, 20])), seed=seed & 0xFFFFF)
    for sp_ in spots:                                                # narrow the cones: most of the frame outside them
        if r.random() < 0.7:
            sp_.outerConeAngle *= 0.4; sp_.innerConeAngle *= 0.4
    if r.random() < 0.3:                                             # one surface per frame: whole waves eligible for the back-facing skip
        gb[1][..., :3] = np.float32(r.choice([-1.0, 0.0, 1.0], 3))
        if r.random() < 0.5:
            gb[1][..., :3] += r.random((H, W, 3), dtype=np.float32) * np.float32(0.2)
    if r.random() < 0.4:                                             # roughness classes: polished (< 0.04: the EPSILON early-out), 0, 1, outside [0, 1]
        gb[1][..., 3] = r.choice(np.array([0.0, 0.01, 0.03999, 0.04, 0.5, 1.0, 1.0000001, -0.1, 2.0], np.float32), (H, W))
    if r.random() < 0.5:                                             # positions on a lattice: lights exactly above pixels, components that cancel
        step = np.float32(2.0 ** int(r.integers(-3, 3)))
        gb[0][..., :3] = np.round(gb[0][..., :3] / step) * step
    cam = [0.0, 10.0, -60.0]
    if r.random() < 0.15:
        y, x = int(r.integers(0, H)), int(r.integers(0, W))
        cam = [float(v) for v in gb[0][y, x, :3]]                    # the camera AT a pixel: V = normalize(0)
    for i in (r.choice(n_pts, size=min(n_pts, int(r.integers(0, 4))), replace=False) if n_pts else []):     # a few damaged lights per frame, not a fixed share of them
        u = r.random()
        l = pts[int(i)]
        if u < 0.35:                                                 # a light AT a pixel / above it on one axis
            y, x = int(r.integers(0, H)), int(r.integers(0, W))
            p = gb[0][y, x, :3].copy()
            if r.random() < 0.5:
                p[int(r.integers(0, 3))] += np.float32(r.choice([1.0, -2.0, 1e-20, 1e-42]))
            l.position.set(tuple(float(v) for v in p))
        elif u < 0.55:
            l.range = float(r.choice([0.0, -1.0, 1e-3, 3e4, 2.0 ** 30, 2.0 ** 31, 1e30, float("inf"), float("nan")]))
        elif u < 0.70:                                               # a non-finite radiance, confined by a short range so that the rest of the frame stays visible
            l.color.set(tuple(float(v) for v in r.choice(SPECIALS, 3)))
            l.range = float(np.float32(8.0 + 20.0 * r.random()))
        elif u < 0.80:
            l.brightness = float(r.choice([0.0, -0.0, -5.0, 1e38, float("inf"), float("nan")]))
            l.range = float(np.float32(8.0 + 20.0 * r.random()))
        else:
            p = [l.position.x, l.position.y, l.position.z]
            p[int(r.integers(0, 3))] = float(r.choice([-0.0, 0.0, 1e-41, 2.0 ** 41, 1e30, float("inf"), float("nan")]))
            l.position.set(tuple(p))
    direc = None
    if r.random() < 0.5:
        direc = abi.DirectionalLight()
        direc.enabled = 1
        direc.shadowing = 0
        direc.brightness = float(r.choice([0.5, 3.0, 1.0, 0.0, 2.0, 0.9, 0.1, float("inf")]))
        direc.color.set((1.0, 0.9, 0.8))
        d = r.choice(np.array([0.0, -0.0, 1.0, -1.0, 0.3, 1e-30, 1e25, np.nan], np.float32), 3) if r.random() < 0.15 else (r.random(3, dtype=np.float32) - np.float32(0.5))
        direc.lightDirection.set(tuple(float(v) for v in d))
    pf, extra = synth.per_frame(points=pts, spots=spots if len(spots) else None, directional=direc, ambient=float(r.choice([0.0, 0.055, 0.000275])),
                                hdri_offset=float(r.choice([0.0, 0.3, -2.0, 7.0, 0.9, -0.4, 3.1415927, 1e10, 100.0, float("nan")])))
    env = None
    if r.random() < 0.6:
        dres = int(r.choice([1, 2, 3, 4, 16, 64]))
        sres = int(r.choice([1, 2, 4, 8, 32, 128, 512]))
        smips = int(r.integers(1, int(np.log2(sres)) + 2))
        lsz = int(r.choice([1, 2, 3, 16, 64, 1024]))
        sp = 0.0 if r.random() < 0.8 else 0.002
        spec = np.concatenate([rand_f16(r, (6, max(1, sres >> m), max(1, sres >> m), 4), sp).reshape(-1) for m in range(smips)])
        env = dict(diffuse=rand_f16(r, (6, dres, dres, 4), sp), spec=spec, sres=sres, smips=smips, lut=rand_f16(r, (lsz, lsz, 2), sp))
    lod = float(r.choice([0.0, 1.0, float(env["smips"]) if env else 7.0, float(env["smips"] - 1) if env else 6.0, 9.0, 20.0, 0.5, -1.0, 1e9, float("nan")], p=[0.08, 0.08, 0.3, 0.2, 0.08, 0.06, 0.06, 0.06, 0.04, 0.04]))
    pv = synth.per_view(W, H, camera=tuple(cam), max_env_lod=lod, diffuse_only=int(r.random() < 0.2))
    if r.random() < 0.5:                                             # special values in scattered lanes of every plane
        for k in range(4):
            n = int(r.integers(1, 9))
            gb[k][r.integers(0, H, n), r.integers(0, W, n), r.integers(0, 4, n)] = r.choice(SPECIALS, n)
    if r.random() < 0.3:                                             # zeros in the accumulator
        gb[0][..., 3] = 0.0; gb[3][...] = 0.0
        if r.random() < 0.5:
            gb[2][..., :3] = -0.0
    return dict(W=W, H=H, gb=gb, pf=pf, pv=pv, extra=extra, env=env, n_pts=n_pts, n_spots=len(spots), fmt=int(r.choice([F32, F16])), dxc=bool(r.integers(0, 2)),
                exp2=bool(r.integers(0, 2)), coherent=coherent)


def run_case(ctx, seed, dev):
    c = case(seed)
    lib = O.load()
    what = (f"seed {seed}: {c['W']}x{c['H']} coherent {c['coherent']} fmt {c['fmt']} dxc {c['dxc']} exp2 {c['exp2']} points {c['n_pts']} spots {c['n_spots']} "
            f"dir {c['pf'].Lights.directional.enabled} env {None if c['env'] is None else (c['env']['diffuse'].shape[1], c['env']['sres'], c['env']['smips'], c['env']['lut'].shape[0])} "
            f"lod {c['pv'].MaxEnvMapLODLevels} diffuse_only {c['pv'].EnvironmentMapDiffuseOnlyIllumination}")
    ctx.set_arithmetic(c["dxc"]); lib.vqo_set_arithmetic(1 if c["dxc"] else 0)
    ctx.set_fresnel_pow(c["exp2"]); lib.vqo_set_fresnel_pow(1 if c["exp2"] else 0)
    try:
        env_o = env_g = None
        keep = []
        if c["env"] is not None:
            e = c["env"]
            env_o = O.host_envmap(e["diffuse"], e["spec"], e["sres"], e["smips"], e["lut"])
            keep = [dev(e["diffuse"]), dev(e["spec"]), dev(e["lut"])]
            env_g = abi.EnvMap(keep[0].data_ptr(), e["diffuse"].shape[1], keep[1].data_ptr(), e["sres"], e["smips"], keep[2].data_ptr(), e["lut"].shape[0])
        with np.errstate(all="ignore"):
            ref = O.forward_lighting(c["gb"], c["pf"], c["pv"], c["fmt"], extra_point=c["extra"], env=env_o)
        got = ctx.forward_lighting([dev(g) for g in c["gb"]], c["pf"], c["pv"], out_fmt=c["fmt"], extra_point=c["extra"], env=env_g).cpu().numpy()
    finally:
        ctx.set_arithmetic(False); lib.vqo_set_arithmetic(0)
        ctx.set_fresnel_pow(False); lib.vqo_set_fresnel_pow(0)
    n, idx = O.bits_equal(got, ref)
    return n, idx, what, got, ref


def main():
    import torch
    from vqengine_amd import capi
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cases", type=int, default=0, help="stop after this many cases (0: by time)")
    a = ap.parse_args()
    ctx = capi.Context(0)
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()  # noqa: E731
    t0, n, fails = time.time(), 0, []
    while (a.cases and n < a.cases) or (not a.cases and time.time() - t0 < a.seconds):
        seed = a.seed * 1000003 + n
        bad, idx, what, got, ref = run_case(ctx, seed, dev)
        if bad:
            fails.append(seed)
            y, x, ch = (int(v) for v in idx[0][:3])
            c = case(seed)
            print(f"MISMATCH {what}: {bad} channels, first at (y {y}, x {x}, c {ch}): got {got[y, x]} want {ref[y, x]}; px {[c['gb'][k][y, x].tolist() for k in range(4)]}", flush=True)
        n += 1
    print(f"fuzz_shade: {n} cases, {len(fails)} failed {fails[:20]}", flush=True)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
