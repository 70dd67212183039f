#!/usr/bin/env python3
"""Randomised parity of the light / shadow-caster path: python tests/fuzz/fuzz_casters.py [--seconds 120] [--seed 1] (needs a GPU; the oracle is the checker).

Every case draws its own light counts (0 .. the cbuffer's limits), shadow-map sizes (powers of two and not, 1 .. 300), view-projection matrices (the engine's
own, lattice-aligned orthographic ones that put taps EXACTLY on texel borders, degenerate ones), depth biases (0, negative, huge, non-finite), depth maps
quantised to a few values (so that `>` meets equality), pixel positions on a lattice, special values in a few lanes, output format, the reading of
dot / normalize and the Fresnel power — and demands the HIP product's bits == the oracle's. Prints one line per failure (with the case's seed) and a summary;
exit status 1 if any case differed. tests/test_gpu_casters.py::test_fuzz_cases_that_failed_once replays the seeds that ever failed."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))          # the fuzzers import each other

from tests import oracle_lib as O  # noqa: E402
from vqengine_amd import abi, scene, synth  # noqa: E402

F32, F16 = abi.FMT_RGBA32F, abi.FMT_RGBA16F
DIM_CHOICES = (1, 2, 3, 4, 5, 7, 8, 16, 31, 32, 33, 64, 100, 128, 255, 256, 257, 300)
SPECIALS = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-30, -1e-30, 1e25, 0.5, 2.0], np.float32)


def lattice_ortho(r, dim):
    """a row-vector-free view-projection (mul(M, float4(P, 1)), ForwardLighting.hlsl:351) that maps the x / z lattice of the frame onto texel borders of a dim^2 map"""
    m = np.zeros((4, 4), np.float32)
    span = np.float32(2.0 ** int(r.integers(3, 8)))                  # world units across the map: a power of two keeps uv exactly on k / dim
    m[0][0] = 2.0 / span; m[1][2] = 2.0 / span                       # x -> ndc x, z -> ndc y
    m[2][1] = np.float32(-1.0 / 64.0); m[2][3] = 0.5                 # y -> depth
    m[3][3] = 1.0
    m[0][3] = np.float32(r.integers(-2, 3)) / np.float32(dim)        # shifts by whole / half texels
    m[1][3] = np.float32(r.integers(-2, 3)) / np.float32(2 * dim)
    return m


def case(seed):
    r = np.random.Generator(np.random.Philox(key=[int(seed), 0xF0]))
    W = int(r.choice([64, 128, 192, 200, 256, 321, 512]))
    H = int(r.integers(1, 7))
    dims = tuple(int(r.choice(DIM_CHOICES)) for _ in range(3))
    pf, _ = scene.engine_max_frame(map_dims=(8, 8, 8))
    scene._set_shadow_dims(pf, dims)
    L = pf.Lights
    L.numPointLights = int(r.choice([0, 1, 3, 17, 100]))
    L.numSpotLights = int(r.choice([0, 1, 2, 7, 20]))
    L.numPointCasters = int(r.integers(0, 6))
    L.numSpotCasters = int(r.integers(0, 6))
    L.directional.enabled = int(r.integers(0, 2))
    L.directional.shadowing = int(r.integers(0, 2))
    maps = scene.synthetic_shadow_maps(dims, n_spot=5, n_point=5, seed=seed)
    if r.random() < 0.6:                                             # depths from a handful of values: the comparison meets equality
        q = np.float32(r.choice([2.0, 4.0, 16.0, 64.0]))
        for k in ("dir", "spot", "point"):
            maps[k] = (np.round(maps[k] * q) / q).astype(np.float32)
    if r.random() < 0.2:
        for k in ("dir", "spot", "point"):
            flat = maps[k].reshape(-1)
            idx = r.integers(0, flat.size, max(1, flat.size // 50))
            flat[idx] = r.choice(SPECIALS, idx.size)
    gb = list(synth.gbuffer(W, H, seed=seed & 0xFFFF, coherent=bool(r.integers(0, 2))))
    gb = [np.array(g, copy=True) for g in gb]
    if r.random() < 0.7:                                             # positions on a lattice (with the orthographic matrices below: taps on texel borders)
        step = np.float32(2.0 ** int(r.integers(-4, 2)))
        gb[0][..., :3] = np.round(gb[0][..., :3] / step) * step
    for i in range(5):                                               # matrices: the engine's own, lattice-aligned, or damaged
        u = r.random()
        if u < 0.45:
            scene._set_matrix(L.shadowViews[i], lattice_ortho(r, dims[1]))
        elif u < 0.55:
            L.shadowViews[i].m[int(r.integers(0, 4))][int(r.integers(0, 4))] = float(r.choice(SPECIALS))
    if r.random() < 0.5:
        scene._set_matrix(L.shadowViewDirectional, lattice_ortho(r, dims[0]))
    biases = [0.0, -1e-3, 5e-5, 9e-6, 1.0, 1e30, float("nan"), float("inf"), -0.0]
    for i in range(5):
        if r.random() < 0.5:
            L.spot_casters[i].depthBias = float(r.choice(biases))
        if r.random() < 0.5:
            L.point_casters[i].depthBias = float(r.choice(biases))
        if r.random() < 0.3:
            L.point_casters[i].range = float(r.choice([0.0, 1.0, 30.0, 1e30, float("nan"), float("inf"), -5.0]))
        if r.random() < 0.2:
            L.spot_casters[i].outerConeAngle = float(r.choice([0.0, 0.1, 1.5707964, 3.1415927, 4.0, float("nan")]))
        if r.random() < 0.2:
            L.spot_casters[i].innerConeAngle = float(r.choice([0.0, 0.1, L.spot_casters[i].outerConeAngle, 4.0]))
    if r.random() < 0.4:
        L.directional.depthBias = float(r.choice(biases))
    if L.numSpotLights and r.random() < 0.15:                        # ONE damaged spot in a few cases (a NaN direction makes every pixel NaN: nothing else is seen then)
        L.spot_lights[int(r.integers(0, L.numSpotLights))].spotDir.set(tuple(float(x) for x in r.choice(SPECIALS, 3)))
    for _ in range(int(r.integers(0, 6))):                           # a pixel AT a light, or next to it
        x, y = int(r.integers(0, W)), int(r.integers(0, H))
        src = L.point_casters[int(r.integers(0, 5))].position if r.random() < 0.5 else L.spot_casters[int(r.integers(0, 5))].position
        gb[0][y, x, :3] = np.array((src.x, src.y, src.z), np.float32) + np.float32(r.choice([0.0, 1e-30, 1e-6]))
    if r.random() < 0.5:                                             # special values in scattered lanes of every plane
        for k in range(4):
            n = int(r.integers(1, 9))
            gb[k][r.integers(0, H, n), r.integers(0, W, n), r.integers(0, 4, n)] = r.choice(SPECIALS, n)
    if r.random() < 0.3:                                             # zeros in the accumulator (the idle-wave exits must keep their signs)
        gb[0][..., 3] = 0.0; gb[3][...] = 0.0
        if r.random() < 0.5:
            gb[2][..., :3] = -0.0
    pf.fAmbientLightingFactor = float(r.choice([0.0, 0.055]))
    pv = synth.per_view(W, H)
    res = dict(W=W, H=H, dims=dims, pf=pf, pv=pv, maps=maps, gb=gb, fmt=int(r.choice([F32, F16])), dxc=bool(r.integers(0, 2)), exp2=bool(r.integers(0, 2)),
               no_maps=r.random() < 0.05, env=None, extra=None)
    # drawn AFTER everything above, so that a seed of the first runs (tests/test_gpu_casters.py replays them) still means the same frame
    if r.random() < 0.3:                                             # the <env, casters> instantiation: random fp16 cubes (tests/fuzz/fuzz_shade.py draws more of their corners)
        import fuzz_shade
        dres, sres, lsz = int(r.choice([1, 4, 16])), int(r.choice([1, 8, 64])), int(r.choice([2, 16, 64]))
        smips = int(r.integers(1, int(np.log2(sres)) + 2))
        spec = np.concatenate([fuzz_shade.rand_f16(r, (6, max(1, sres >> m), max(1, sres >> m), 4), 0.0).reshape(-1) for m in range(smips)])
        res["env"] = dict(diffuse=fuzz_shade.rand_f16(r, (6, dres, dres, 4), 0.0), spec=spec, sres=sres, smips=smips, lut=fuzz_shade.rand_f16(r, (lsz, lsz, 2), 0.0))
        pv.MaxEnvMapLODLevels = float(smips - int(r.integers(0, 2)))
        pf.fHDRIOffsetInRadians = float(r.choice([0.0, 0.3, -2.0]))
    if r.random() < 0.15:                                            # the extension array behind the cbuffer's lights
        res["extra"] = synth.point_lights(int(r.choice([3, 40])), seed=(seed >> 3) & 0xFFFFF)
    return res


def run_case(ctx, seed, dev):
    from vqengine_amd import capi
    c = case(seed)
    lib = O.load()
    L = c["pf"].Lights
    what = (f"seed {seed}: {c['W']}x{c['H']} dims {c['dims']} fmt {c['fmt']} dxc {c['dxc']} exp2 {c['exp2']} lights p{L.numPointLights} s{L.numSpotLights} "
            f"pc{L.numPointCasters} sc{L.numSpotCasters} dir {L.directional.enabled}/{L.directional.shadowing} maps {not c['no_maps']} env {c['env'] is not None} extra {0 if c['extra'] is None else len(c['extra'])}")
    ctx.set_arithmetic(c["dxc"]); lib.vqo_set_arithmetic(1 if c["dxc"] else 0)
    ctx.set_fresnel_pow(c["exp2"]); lib.vqo_set_fresnel_pow(1 if c["exp2"] else 0)
    try:
        m = c["maps"]
        keep = [dev(m[k]) for k in ("dir", "spot", "point")]
        sm_g = None if c["no_maps"] else abi.ShadowMaps(keep[0].data_ptr(), m["dims"][0], keep[1].data_ptr(), m["dims"][1], keep[2].data_ptr(), m["dims"][2])
        sm_o = None if c["no_maps"] else scene.shadow_maps_struct(m, lambda a: a.ctypes.data)
        casters = L.numPointCasters > 0 or L.numSpotCasters > 0 or (L.directional.enabled and L.directional.shadowing)
        gb_d = [dev(g) for g in c["gb"]]
        env_o = env_g = None
        if c["env"] is not None:
            e = c["env"]
            env_o = O.host_envmap(e["diffuse"], e["spec"], e["sres"], e["smips"], e["lut"])
            keep += [dev(e["diffuse"]), dev(e["spec"]), dev(e["lut"])]
            env_g = abi.EnvMap(keep[-3].data_ptr(), e["diffuse"].shape[1], keep[-2].data_ptr(), e["sres"], e["smips"], keep[-1].data_ptr(), e["lut"].shape[0])
        if c["no_maps"] and casters:                                 # casters without maps: both sides refuse (capi.hip validateLighting, vqo_forward_lighting)
            refused = [False, False]
            try:
                ctx.forward_lighting(gb_d, c["pf"], c["pv"], out_fmt=c["fmt"], shadow=None)
            except capi.VQHipError:
                refused[0] = True
            try:
                O.forward_lighting(c["gb"], c["pf"], c["pv"], c["fmt"], shadow=None)
            except AssertionError:
                refused[1] = True
            return (0 if all(refused) else 1), [], what + f" refused {refused}", None, None
        with np.errstate(all="ignore"):
            ref = O.forward_lighting(c["gb"], c["pf"], c["pv"], c["fmt"], shadow=sm_o, env=env_o, extra_point=c["extra"])
        got = ctx.forward_lighting(gb_d, c["pf"], c["pv"], out_fmt=c["fmt"], shadow=sm_g, env=env_g, extra_point=c["extra"]).cpu().numpy()
    finally:
        ctx.set_arithmetic(False); lib.vqo_set_arithmetic(0)
        ctx.set_fresnel_pow(False); lib.vqo_set_fresnel_pow(0)
    n, idx = O.bits_equal(got, ref)                                  # any NaN == any NaN, everything else bit for bit (the comparison of tests/test_gpu_parity.py)
    return n, idx, what, got, ref


def fuzz_px(seed, y, x):
    c = case(seed)
    return [c["gb"][k][y, x].tolist() for k in range(4)]


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
            if got is None:
                print(f"MISMATCH {what}", flush=True)
            else:
                y, x, ch = (int(v) for v in idx[0][:3])
                print(f"MISMATCH {what}: {bad} channels, first at (y {y}, x {x}, c {ch}): got {got[y, x]} want {ref[y, x]}; pos {fuzz_px(seed, y, x)}", flush=True)
        n += 1
    print(f"fuzz_casters: {n} cases, {len(fails)} failed {fails[:20]}", flush=True)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
