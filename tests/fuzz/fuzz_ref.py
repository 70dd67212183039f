#!/usr/bin/env python3
"""Randomised check of the ORACLE against the REFERENCE'S OWN SHADER SOURCE (oracle/_ref: the HLSL compiled for the CPU through oracle/ref_src/hlsl_shim.h) — CPU only, runs
where /root/reference was available to build oracle/_ref:   python tests/fuzz/fuzz_ref.py [--seconds 120] [--seed 1]

The cases are the GPU fuzzers' own (tests/fuzz/fuzz_shade.py, fuzz_casters.py, fuzz_post.py: same seeds, same frames). The bar is the pinning tests' (tests/ref_cases.py): the
oracle's RGBA16F / RGBA8 output within ONE unit of the storage format of what the reference's HLSL writes, wherever the reference's value is finite and the frame is not
saturated by a non-finite light; a NaN / infinity in one and a finite value in the other is reported too (the arithmetic contract regroups a few products — DESIGN.md 3 — so an
overflow may surface as inf in one and NaN in the other: those are counted, not failed). Exit status 1 if a finite channel differs by more than one unit."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from tests import oracle_lib as O, ref_lib as R  # noqa: E402
from tests.ref_cases import at_boundary, ulp16_distance  # noqa: E402
from vqengine_amd import abi, scene  # noqa: E402

F32, F16, R8 = abi.FMT_RGBA32F, abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM


def compare16(ref, got16, strict_px):
    """ref: float32 values the HLSL wrote; got16: the oracle's RGBA16F output; strict_px: [H,W] pixels where more than one unit is a FAILURE.
    -> (finite channels compared, channels above one unit, of those in strict pixels, worst, channels finite in one and not in the other, first strict offenders)"""
    ref, got = np.asarray(ref, np.float32)[..., :3], np.asarray(got16)[..., :3].astype(np.float32)
    with np.errstate(all="ignore"):
        r16 = ref.astype(np.float16).astype(np.float32)
    both = np.isfinite(r16) & np.isfinite(got)
    d = ulp16_distance(np.where(both, got, 0), np.where(both, ref, 0))
    cls = (np.isnan(r16) != np.isnan(got)) | (np.isinf(r16) != np.isinf(got))
    strict = (d > 1) & strict_px[..., None]
    return int(both.sum()), int((d > 1).sum()), int(strict.sum()), int(d.max()) if both.any() else 0, int(cls.sum()), np.argwhere(strict)[:3].tolist()


def run_shade(seed, casters):
    """Where more than one unit FAILS: the literal reading (the oracle writes the HLSL's own expression trees there, contract v5: the two sides perform the same IEEE operations)
    on pixels inside the domain the C++ shim defines — `int(roughness * MaxEnvMapLODLevels)` of a NaN / out-of-range product is undefined behaviour in C++ (D3D: NaN -> 0,
    saturating), so such pixels are only counted. The DXC reading regroups quotients into a * rcp(b) and the BRDF into fma(F, sG - kA, kA) (contract v2-v4): one-ulp differences
    that the GGX lobe of a polished pixel amplifies by up to 1 / a^2 — counted below roughness 0.3, failed above it (without environment cubes: a white-noise cube turns an ulp
    of uv into a different 1/256 filter step)."""
    import fuzz_casters
    import fuzz_shade
    c = (fuzz_casters if casters else fuzz_shade).case(seed)
    if casters and c["no_maps"]:
        return None
    lib = O.load()
    lib.vqo_set_arithmetic(1 if c["dxc"] else 0); lib.vqo_set_fresnel_pow(1)      # the shim's pow is exp2(y * log2 x): the engine-faithful lowering on both sides
    try:
        env_o = None
        if c["env"] is not None:
            e = c["env"]
            env_o = O.host_envmap(e["diffuse"], e["spec"], e["sres"], e["smips"], e["lut"])
        sm = scene.shadow_maps_struct(c["maps"], lambda a: a.ctypes.data) if casters else None
        with np.errstate(all="ignore"):
            # the reference harness takes the rasterised normal and normalises it itself (PSMain :264-266); the oracle, like the product, starts behind that line
            got = O.forward_lighting(at_boundary(c["gb"]), c["pf"], c["pv"], F16, extra_point=c["extra"], env=env_o, shadow=sm)
            try:
                ref = R.forward_from_gbuffer(c["gb"], c["pf"], c["pv"], env=env_o, shadow=sm, extra=c["extra"], reading="dxc" if c["dxc"] else "literal")
            except AssertionError:                                   # more extension lights than the reference build's raised cap (256 in all)
                return None
            rough = c["gb"][1][..., 3]
            mipf = rough * np.float32(c["pv"].MaxEnvMapLODLevels)
            defined = np.isfinite(rough) & ((env_o is None) | (np.isfinite(mipf) & (np.abs(mipf) < 2.0 ** 30)))
            strict = defined & ((rough >= 0.3) & (env_o is None) if c["dxc"] else True)
    finally:
        lib.vqo_set_arithmetic(0); lib.vqo_set_fresnel_pow(0)
    return compare16(ref, got, strict) + ("dxc" if c["dxc"] else "literal",)


def run_post(seed):
    import fuzz_post
    c = fuzz_post.case(seed)
    if c["w"] * c["h"] > 300000 or c["params"].OutputDisplayCurveEnum not in (0, 1, 2):
        return None
    img = c["img"][c["hr"]:c["hr"] + c["h"]].astype(np.float32)
    with np.errstate(all="ignore"):
        x = R.blur_pass(img, 0).astype(np.float16).astype(np.float32)
        y = R.blur_pass(x, 1).astype(np.float16).astype(np.float32)
        ref = R.tonemap(y, c["params"])
        got = O.tonemap(O.gaussian_blur(img.astype(np.float16), F16), F16, R8, params=c["params"])
    from tests.ref_cases import to_unorm8
    r8 = to_unorm8(np.nan_to_num(ref[..., :3], nan=0.0))
    ok = np.isfinite(ref[..., :3])
    d = np.abs(got[..., :3].astype(np.int32) - r8.astype(np.int32)) * ok
    return int(ok.sum()), int((d > 1).sum()), int((d > 1).sum()), int(d.max()), 0, np.argwhere(d > 1)[:3].tolist(), "post"


def _cmp16(got16, ref32):
    got = np.asarray(got16)[..., :3].astype(np.float32)
    with np.errstate(all="ignore"):
        r16 = np.asarray(ref32, np.float32)[..., :3].astype(np.float16).astype(np.float32)
    both = np.isfinite(r16) & np.isfinite(got)
    d = ulp16_distance(np.where(both, got, 0), np.where(both, ref32[..., :3], 0))
    cls = (np.isnan(r16) != np.isnan(got)) | (np.isinf(r16) != np.isinf(got))
    return int(both.sum()), int((d > 1).sum()), int((d > 1).sum()), int(d.max()) if both.any() else 0, int(cls.sum()), np.argwhere(d > 1)[:3].tolist()


def run_wide(seed):
    """FSR EASU + RCAS, the skydome and the reflections composite on finite random inputs: the oracle's RGBA16F output within one unit of the HLSL's values"""
    from vqengine_amd import synth
    r = np.random.Generator(np.random.Philox(key=[int(seed), 0xF5]))
    kind = str(r.choice(["fsr", "skydome", "reflect"]))
    with np.errstate(all="ignore"):
        if kind == "fsr":
            iw, ih = int(r.integers(2, 90)), int(r.integers(2, 60))
            ow, oh = max(1, int(iw * r.choice([1.0, 1.3, 1.5, 2.0, 3.0]))), max(1, int(ih * r.choice([1.0, 1.3, 1.5, 2.0])))
            img = (r.random((ih, iw, 4), dtype=np.float32) * np.float32(r.choice([1.0, 4.0]))).astype(np.float16)
            if r.random() < 0.3:
                img[...] = np.repeat(np.repeat(img[::4, ::4], 4, 0), 4, 1)[:ih, :iw]
            stops = float(r.choice([0.0, 0.2, 1.0, 2.0]))
            up_o = O.fsr_easu(img, F16, ow, oh)
            up_r = R.fsr_easu(img.astype(np.float32), ow, oh, R.fsr_easu_con(iw, ih, ow, oh))
            a = _cmp16(up_o, up_r)
            sh_o = O.fsr_rcas(up_o, F16, con=O.fsr_rcas_con(stops))
            sh_r = R.fsr_rcas(up_o.astype(np.float32), R.fsr_rcas_con(stops))
            b = _cmp16(sh_o, sh_r)
            return (a[0] + b[0], a[1] + b[1], a[2] + b[2], max(a[3], b[3]), a[4] + b[4], a[5] or b[5], "fsr")
        if kind == "skydome":
            W, H = int(r.choice([8, 48, 100])), int(r.integers(1, 12))
            ew, eh = [(8, 4), (64, 32), (256, 128)][int(r.integers(0, 3))]
            eq = synth.equirect(ew, eh, seed=int(r.integers(0, 1 << 20)))
            sp = scene.skydome_params(float(r.uniform(-4, 4)), float(r.uniform(-1.5, 1.5)), float(r.uniform(-4, 4)), float(r.uniform(0.2, 2.5)), W, H)
            got = O.skydome(eq, sp, np.zeros((H, W, 4), np.float16), F16)
            ref = R.skydome(eq, sp, W, H)
            res = _cmp16(got, ref)
            # an ulp of uv can move the 8-bit filter fraction of a pixel by one step next to a sun (tests/ref_cases.py "skydome"): counted, failed only beyond 0.3 % of the channels
            return res[:2] + ((res[1] if res[1] > 0.003 * max(1, res[0]) else 0),) + res[3:] + ("skydome",)
        W, H = int(r.choice([4, 64, 200])), int(r.integers(1, 9))
        mk = lambda: (r.random((H, W, 4), dtype=np.float32) * np.float32(r.choice([1.0, 100.0]))).astype(np.float16)  # noqa: E731
        refl, scn = mk(), mk()
        bv = mk() if r.random() < 0.5 else None
        if bv is not None:
            bv[..., 3] = r.choice(np.array([0.0, 1.0, 0.5, 0.25], np.float32), (H, W)).astype(np.float16)
        got = O.composite_reflections(refl, scn, F16, bv)
        ref = R.apply_reflections_bv(refl, bv, scn) if bv is not None else R.apply_reflections(refl, scn)
        return _cmp16(got, ref) + ("reflections",)


def run_psmain(seed):
    """PSMain from interpolants + a material table over mip-chained RGBA8 maps (texture sampling, uv transform, normal mapping, SSAO, alpha-masked permutation in some cases)
    and the Z pre-pass's normals: the oracle's producer + shade / scene normals against the reference's HLSL, valid pixels only. The texture sampler both sides call is the
    same statement of D3D's rules (oracle/vqo_sampling.h through ref_hooks): what is compared is everything around it."""
    from vqengine_amd import synth
    r = np.random.Generator(np.random.Philox(key=[int(seed), 0xF7]))
    W, H, NM = int(r.choice([16, 48, 96])), int(r.integers(2, 12)), int(r.choice([1, 3, 6]))
    am = bool(r.random() < 0.3)
    ip = [p.copy() for p in synth.interpolants(W, H, NM, seed=int(r.integers(0, 1 << 20)))]
    datas, texsets = synth.material_set(NM, seed=int(r.integers(0, 1 << 20)), max_dim=int(r.choice([8, 64])))
    hc = []
    for ts in texsets:
        hc.append({slot: (O.mip_chain_rgba8(img)[0], img.shape[1], img.shape[0], O.mip_chain_rgba8(img)[1]) for slot, img in ts.items()})
    mats = O.host_materials(datas, hc)
    for k in range(NM):
        mats[k].texDiffuse.reserved = abi.MATERIAL_ALPHA_MASKED if am else 0
    ssao = synth.ssao_image(W, H) if r.random() < 0.5 else None
    pf, _ = synth.per_frame(points=synth.point_lights(int(r.choice([1, 4, 12])), seed=seed & 0xFFFF), spots=synth.spot_lights(int(r.choice([1, 2])), seed=seed & 0xFFF),
                            directional=synth.directional_light() if r.random() < 0.5 else None, ambient=float(r.choice([0.0, 0.055])))
    pv = synth.per_view(W, H)
    idx0 = np.ascontiguousarray(ip[2][..., 3]).view(np.int32)
    valid = (idx0 >= 0) & (idx0 < NM)
    with np.errstate(all="ignore"):
        ipo = [p.copy() for p in ip]
        gb = O.gbuffer_from_materials(ipo, mats, pf.fAmbientLightingFactor, ssao=ssao)
        lit = O.forward_lighting(gb, pf, pv, F16).astype(np.float32)
        gone = valid & (np.ascontiguousarray(ipo[2][..., 3]).view(np.int32) == -1)
        lit[gone] = -1.0                                             # discarded fragments: the sentinel the reference harness writes
        ref = R.forward_psmain([p.copy() for p in ip], mats, pf, pv, ssao=ssao, alpha_masked=am)
        a = _cmp16(lit[valid][None].astype(np.float16), ref[valid][None])
        nr = R.prepass_normals([p.copy() for p in ip], mats, alpha_masked=am)
        no = O.scene_normals_from_materials([p.copy() for p in ip], mats, F32)
        same = int((no[valid][..., :3].view(np.uint32) != nr[valid][..., :3].view(np.uint32)).sum())
    return (a[0] + int(valid.sum()) * 3, a[1] + same, a[2] + same, a[3], a[4], a[5], "psmain_materials")


def run_ibl(seed):
    """The load-time passes on random small equirects: diffuse irradiance (the shader's default step 0.010: 99 382 taps per texel), GGX specular mips, BRDF LUT texels.
    Diffuse and LUT: within one unit, failed otherwise. Specular: counted — a tap whose uv / LOD lands an ulp to the other side of a 1/256 filter step (polynomial atan2 / asin /
    log2 against libm) moves a 512-tap mean by more (tests/golden/specular_filterstep_tail.json is that class at cfg4's size); every such channel must show that tap, else it fails."""
    from vqengine_amd import synth
    r = np.random.Generator(np.random.Philox(key=[int(seed), 0xF6]))
    kind = str(r.choice(["diffuse", "specular", "specular", "lut"]))
    with np.errstate(all="ignore"):
        if kind == "lut":
            n = 24
            xs, ys = r.integers(0, 1024, n), r.integers(0, 1024, n)
            ref = R.brdf_lut_texels(xs, ys)
            rows = {int(y): O.brdf_lut(1024, 2048, abi.FMT_RG16F, rows=(int(y), int(y) + 1))[0] for y in set(int(v) for v in ys)}
            got = np.stack([rows[int(y)][int(x)] for x, y in zip(xs, ys)]).astype(np.float32)
            g3, r3 = np.concatenate([got, got[:, :1]], 1), np.concatenate([ref, ref[:, :1]], 1)
            return _cmp16(g3.astype(np.float16), r3) + ("brdf_lut",)
        w, h = [(16, 8), (32, 16), (64, 32), (64, 64), (32, 64)][int(r.integers(0, 5))]
        eq = synth.equirect(w, h, seed=int(r.integers(0, 1 << 20))) if r.random() < 0.6 else (r.random((h, w, 4), dtype=np.float32) * np.float32(r.choice([1.0, 50.0]))).astype(np.float32)
        eq[..., 3] = 1.0
        chain, n = O.mip_chain(eq)
        if kind == "diffuse":
            res = int(r.choice([1, 2]))
            got = O.conv_diffuse(chain, w, h, n, res, 0.010, abi.CONV_SEQUENTIAL, F16)
            return _cmp16(got, R.conv_diffuse(chain, w, h, n, res)) + ("conv_diffuse",)
        res0 = int(r.choice([4, 8, 16]))
        mips = abi.specular_mip_count(res0)
        got = O.conv_specular(chain, w, h, n, res0, abi.CONV_SEQUENTIAL, F16)[0]
        ref = np.concatenate([R.conv_specular_mip(chain, w, h, n, res0 >> m, float(np.float32(m) / np.float32(mips - 1)), m).reshape(-1, 4) for m in range(mips)])
        c = _cmp16(got, ref)
        # a channel above one unit must carry its CAUSE: a tap of that texel whose 8-bit filter fraction (u, v) or LOD fraction differs between the reference's evaluation
        # (libm atan2f / asinf / log2f) and the contract's polynomials — the criterion of tests/golden/make_filterstep_tail.py. A texel without such a tap FAILS
        d = ulp16_distance(np.asarray(got)[..., :3].astype(np.float32), ref[..., :3])
        fail = 0
        for t in sorted(set(int(b[0]) for b in np.argwhere(d > 1))):
            base, loc = 0, None
            for m in range(mips):
                rr = res0 >> m
                if t < base + 6 * rr * rr:
                    q = t - base
                    loc = (m, rr, q // (rr * rr), q % rr, (q // rr) % rr)
                    break
                base += 6 * rr * rr
            m, rr, f, x, y = loc
            to, _ = O.conv_specular_taps(chain, w, h, n, res0, t)
            tr, _ = R.conv_specular_taps(chain, w, h, n, rr, float(np.float32(m) / np.float32(mips - 1)), m, int(f), int(x), int(y))
            explained = len(to) != len(tr)
            if not explained:
                fx = lambda v: np.floor(v.astype(np.float64) * 256.0 + 0.5).astype(np.int64)  # noqa: E731
                lo, lr = np.clip(to[:, 2], 0, n - 1), np.clip(tr[:, 2], 0, n - 1)
                lv = np.floor(np.minimum(lo, lr)).astype(np.int64)
                explained = bool((fx(lo) != fx(lr)).any())
                for dl in (0, 1):                                    # a trilinear tap filters BOTH levels it blends, each with its own fractions
                    Wl, Hl = np.maximum(w >> (lv + dl), 1).astype(np.float64), np.maximum(h >> (lv + dl), 1).astype(np.float64)
                    explained |= bool(((fx(to[:, 0].astype(np.float64) * Wl - 0.5) != fx(tr[:, 0].astype(np.float64) * Wl - 0.5)) |
                                       (fx(to[:, 1].astype(np.float64) * Hl - 0.5) != fx(tr[:, 1].astype(np.float64) * Hl - 0.5))).any())
            if not explained:
                fail += int((d[t] > 1).sum())
        return c[:2] + (fail,) + c[3:] + ("conv_specular",)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", default="", help="one of shade casters post wide ibl psmain")
    a = ap.parse_args()
    if not (R.available("shaders") and R.available("shaders_dxc") and R.available("shaders_l256")):
        raise SystemExit("fuzz_ref: oracle/_ref is not built (needs /root/reference: make -C oracle ref)")
    t0, n, fails, skipped = time.time(), 0, [], 0
    tot = {}
    while time.time() - t0 < a.seconds:
        seed = a.seed * 1000003 + n
        kind = a.only or ("shade", "casters", "post", "wide", "ibl", "psmain")[n % 6]
        res = (run_post(seed) if kind == "post" else run_wide(seed) if kind == "wide" else run_ibl(seed) if kind == "ibl" else run_psmain(seed) if kind == "psmain"
               else run_shade(seed, kind == "casters"))
        n += 1
        if res is None:
            skipped += 1
            continue
        ch, above, strict, worst, cls, where, reading = res
        t = tot.setdefault(reading, {"cases": 0, "channels": 0, "above_one_unit": 0, "of_those_failures": 0, "worst": 0, "finite_in_one_only": 0})
        t["cases"] += 1; t["channels"] += ch; t["above_one_unit"] += above; t["of_those_failures"] += strict; t["finite_in_one_only"] += cls; t["worst"] = max(t["worst"], worst)
        if strict:
            fails.append((kind, seed))
            print(f"FAIL {kind} ({reading}) seed {seed}: {strict} channels above one unit where the two sides perform the same operations, first at {where}", flush=True)
    for reading, t in sorted(tot.items()):
        print(f"fuzz_ref {reading}: {t}", flush=True)
    print(f"fuzz_ref: {n} cases ({skipped} skipped), failures: {fails[:20]}", flush=True)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
