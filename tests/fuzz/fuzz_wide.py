#!/usr/bin/env python3
"""Randomised parity of the kernels either side of the hot path: python tests/fuzz/fuzz_wide.py [--seconds 120] [--seed 1] (needs a GPU; the oracle is the checker).

One of, per case:
  psmain    PSMain as ONE kernel (vqhip_forward_lighting_from_materials, with or without the other render targets): random interpolants (a share of them arbitrary bit
            patterns), a random material table over mip-chained RGBA8 maps, SSAO on / off, a few lights, environment cubes in some cases == the oracle's producer + shade
  producer  vqhip_gbuffer_from_materials alone, vqhip_scene_normals_from_materials (both target formats)
  fsr       EASU (random source / target sizes, ratios from 1x to 4x, both RGBA8 and RGBA16F) followed by RCAS at a random sharpness
  skydome   random cameras, fields of view, equirect sizes and coverage planes
  hdr       run-length coded / flat .hdr files of random sizes, cut at a random byte in a third of the cases: the same image, or the same refusal
  ssr       the environment fallback of SSR's tile classification on random surfaces / cubes / thresholds, packed or float normals (seeds >= 2 000 000)
  reflect   the reflections composite, with and without the bounding-volume image;  viz: every Visualization.hlsl draw mode on every colour format
The HIP product's bits must equal the oracle's. Exit status 1 if a case differed."""
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

F32, F16, R8 = abi.FMT_RGBA32F, abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM
SPECIALS = np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 1e-45, -1e-40, 1e30, -1e30, 3.4e38, 1e-30, 0.5, 1.0, 255.0, 65536.0], np.float32)


def _materials(ctx, dev, r, n):
    datas, texsets = synth.material_set(n, seed=int(r.integers(0, 1 << 20)), max_dim=int(r.choice([8, 64, 256])), same_size=bool(r.integers(0, 2)))
    host_chains, keep = [], []
    dmats = (abi.MaterialDesc * n)()
    for i, (d, ts) in enumerate(zip(datas, texsets)):
        cs = {}
        dmats[i].data = d
        for slot, img in ts.items():
            chain_o, nm = O.mip_chain_rgba8(img)
            chain_g, _ = ctx.mip_chain_rgba8(dev(img))
            cs[slot] = (chain_o, img.shape[1], img.shape[0], nm)
            keep.append(chain_g)
            setattr(dmats[i], slot, abi.Texture2D(chain_g.data_ptr(), img.shape[1], img.shape[0], nm, 0))
        host_chains.append(cs)
    return O.host_materials(datas, host_chains), dmats, (keep, host_chains)


def _interpolants(r, W, H, n_mat):
    ip = [p.copy() for p in synth.interpolants(W, H, n_mat, seed=int(r.integers(0, 1 << 20)))]
    if r.random() < 0.4:
        idx = ip[2][..., 3].copy()
        for k in range(3):
            flat = ip[k].reshape(-1)
            sel = r.random(flat.size) < 0.03
            flat[sel] = r.integers(0, 2 ** 32, int(sel.sum()), dtype=np.uint32).view(np.float32)
            sel = r.random(flat.size) < 0.03
            flat[sel] = SPECIALS[r.integers(0, len(SPECIALS), int(sel.sum()))]
        ip[2][..., 3] = idx                                          # material indices stay valid (the product validates nothing per pixel: an index is an index)
    return ip


def _cmp(pairs):
    bad, first = 0, []
    for a, b in pairs:
        a = a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)
        b = np.asarray(b)
        if a.dtype != b.dtype:
            a = a.view(b.dtype)
        n, idx = O.bits_equal(a.reshape(b.shape), b)
        bad += n
        if n and not len(first):
            first = idx
    return bad, first


def run_case(ctx, seed, dev):
    import torch
    from vqengine_amd import capi
    r = np.random.Generator(np.random.Philox(key=[int(seed), 0xF4]))
    kind = str(r.choice(["psmain", "psmain", "producer", "fsr", "fsr", "skydome", "hdr", "hdr"] + (["ssr", "ssr", "viz", "reflect"] if seed >= 2000000 else [])))   # seeds below 2 000 000 keep their first meaning
    with np.errstate(all="ignore"):
        if kind in ("psmain", "producer"):
            W, H, NM = int(r.choice([1, 64, 130, 200, 333])), int(r.integers(1, 9)), int(r.choice([1, 3, 6, 16]))
            ip = _interpolants(r, W, H, NM)
            hmats, dmats, keep = _materials(ctx, dev, r, NM)
            ssao = synth.ssao_image(W, H) if r.random() < 0.5 else None
            ipd = [dev(p) for p in ip]
            ssd = dev(ssao) if ssao is not None else None
            what = f"seed {seed}: {kind} {W}x{H} materials {NM} ssao {ssao is not None}"
            if kind == "producer":
                ref = O.gbuffer_from_materials([p.copy() for p in ip], hmats, 0.055, ssao)
                got = ctx.gbuffer_from_materials([t.clone() for t in ipd], dmats, 0.055, ssd)
                fmt = int(r.choice([abi.FMT_R10G10B10A2_UNORM, F32]))
                nr = O.scene_normals_from_materials(ip, hmats, fmt)
                ng = ctx.scene_normals_from_materials(ipd, dmats, fmt)
                return _cmp(list(zip(got, ref)) + [(ng, nr)]) + (what + f" normals fmt {fmt}",)
            pts = synth.point_lights(int(r.choice([0, 2, 9, 64, 110])), seed=seed & 0xFFFFF)
            pf, extra = synth.per_frame(points=pts if len(pts) else None, spots=synth.spot_lights(int(r.choice([0, 2])), seed=seed & 0xFFF) or None,
                                        ambient=float(r.choice([0.0, 0.055])), hdri_offset=float(r.choice([0.0, 0.4])))
            env_o = env_g = None
            lod = 0.0
            if r.random() < 0.4:
                import fuzz_shade
                dres, sres, lsz = int(r.choice([1, 4, 16])), int(r.choice([2, 8, 64])), int(r.choice([2, 16, 64]))
                smips = int(r.integers(1, int(np.log2(sres)) + 2))
                spec = np.concatenate([fuzz_shade.rand_f16(r, (6, max(1, sres >> m), max(1, sres >> m), 4), 0.0).reshape(-1) for m in range(smips)])
                dif, lut = fuzz_shade.rand_f16(r, (6, dres, dres, 4), 0.0), fuzz_shade.rand_f16(r, (lsz, lsz, 2), 0.0)
                env_o = O.host_envmap(dif, spec, sres, smips, lut)
                kd = [dev(dif), dev(spec), dev(lut)]
                keep = (keep, kd)
                env_g = abi.EnvMap(kd[0].data_ptr(), dres, kd[1].data_ptr(), sres, smips, kd[2].data_ptr(), lsz)
                lod = float(smips)
            pv = synth.per_view(W, H, max_env_lod=lod)
            fmt = int(r.choice([F32, F16]))
            dxc = bool(r.integers(0, 2))
            ctx.set_arithmetic(dxc); O.load().vqo_set_arithmetic(1 if dxc else 0)
            try:
                gb = O.gbuffer_from_materials([p.copy() for p in ip], hmats, pf.fAmbientLightingFactor, ssao)
                ref = O.forward_lighting(gb, pf, pv, fmt, extra_point=extra, env=env_o)
                pairs = []
                if r.random() < 0.4:
                    svc, svp = synth.clip_positions(W, H)
                    out, alb, mv = ctx.forward_lighting_from_materials_mrt([t.clone() for t in ipd], dmats, pf, pv, albedo_fmt=F16, motion_fmt=abi.FMT_RG16F, sv_curr=dev(svc),
                                                                           sv_prev=dev(svp), ssao=ssd, out_fmt=fmt, extra_point=extra, env=env_g)
                    ra, rm = O.psmain_extra_targets(gb, svc, svp, F16, abi.FMT_RG16F)
                    ipo = [p.copy() for p in ip]
                    O.gbuffer_from_materials(ipo, hmats, pf.fAmbientLightingFactor, ssao)          # marks alpha-mask discards -1 in ip2.w, like the product
                    idx = np.ascontiguousarray(ipo[2][..., 3]).view(np.int32)
                    cov = (idx >= 0) & (idx < NM)
                    ra[~cov] = 0; rm[~cov] = 0                       # PSMain never runs where no fragment arrives: the targets keep their clear value (ForwardLighting.hlsl:382-389)
                    pairs = [(alb, ra), (mv, rm)]
                    what += " mrt"
                else:
                    out = ctx.forward_lighting_from_materials([t.clone() for t in ipd], dmats, pf, pv, ssao=ssd, out_fmt=fmt, extra_point=extra, env=env_g)
                return _cmp([(out, ref)] + pairs) + (what + f" fmt {fmt} dxc {dxc} lights {len(pts)} env {env_o is not None}",)
            finally:
                ctx.set_arithmetic(False); O.load().vqo_set_arithmetic(0)
        if kind == "fsr":
            iw, ih = int(r.integers(1, 200)), int(r.integers(1, 120))
            ow, oh = max(1, int(iw * r.choice([1.0, 1.3, 1.5, 2.0, 3.0, 4.0]))), max(1, int(ih * r.choice([1.0, 1.3, 1.5, 2.0, 3.0])))
            fmt = int(r.choice([R8, F16]))
            if fmt == R8:
                img = r.integers(0, 256, (ih, iw, 4), dtype=np.uint8)
                if r.random() < 0.3:
                    img[...] = np.repeat(np.repeat(r.integers(0, 256, ((ih + 7) // 8, (iw + 7) // 8, 4), dtype=np.uint8), 8, 0), 8, 1)[:ih, :iw]
            else:
                img = (r.random((ih, iw, 4), dtype=np.float32) * np.float32(r.choice([1.0, 4.0, 1000.0]))).astype(np.float16)
                if r.random() < 0.3:
                    n = int(r.integers(1, 6))
                    img[r.integers(0, ih, n), r.integers(0, iw, n), r.integers(0, 4, n)] = r.choice(np.array([np.inf, np.nan, 0.0, -0.0, -1.0, 65504.0, 6e-8], np.float16), n)
            stops = float(r.choice([0.0, 0.2, 1.0, 2.0]))
            up_o = O.fsr_easu(img, fmt, ow, oh)
            up_g = ctx.fsr_easu(dev(img), fmt, ow, oh)
            sh_o = O.fsr_rcas(up_o, fmt, con=O.fsr_rcas_con(stops))
            sh_g = ctx.fsr_rcas(up_g, fmt, con=capi.fsr_rcas_con(stops))
            return _cmp([(up_g, up_o), (sh_g, sh_o)]) + (f"seed {seed}: fsr {iw}x{ih} -> {ow}x{oh} fmt {fmt} stops {stops}",)
        if kind == "skydome":
            W, H = int(r.choice([1, 64, 200, 333])), int(r.integers(1, 20))
            ew, eh = [(8, 4), (64, 32), (256, 128), (37, 19)][int(r.integers(0, 4))]
            eq = synth.equirect(ew, eh, seed=int(r.integers(0, 1 << 20)))
            sp = scene.skydome_params(float(r.uniform(-4, 4)), float(r.uniform(-1.5, 1.5)), float(r.uniform(-4, 4)), float(r.uniform(0.1, 3.0)), W, H)
            fmt = int(r.choice([F32, F16]))
            base = (r.random((H, W, 4), dtype=np.float32)).astype(np.float16 if fmt == F16 else np.float32)
            cov = None
            if r.random() < 0.6:
                cov = [p.copy() for p in synth.interpolants(W, H, 4, seed=int(r.integers(0, 1 << 20)))]
                m = r.random((H, W)) < 0.5
                idx = np.ascontiguousarray(cov[2][..., 3]).view(np.int32).copy()
                idx[m] = -1
                cov[2][..., 3] = idx.view(np.float32)
            ref = O.skydome(eq, sp, base.copy(), fmt, cov[2] if cov is not None else None)
            got = ctx.skydome(dev(eq), sp, dev(base), fmt, [dev(p) for p in cov] if cov is not None else None)
            return _cmp([(got, ref)]) + (f"seed {seed}: skydome {W}x{H} equirect {ew}x{eh} fmt {fmt} coverage {cov is not None}",)
        if kind == "ssr":
            import fuzz_shade
            W, H = int(r.choice([1, 64, 130, 333])), int(r.integers(1, 9))
            sc, depth, packed, nf = synth.ssr_surfaces(W, H, seed=int(r.integers(0, 1 << 20)), sky_fraction=float(r.choice([0.0, 0.1, 0.9])))
            dres, sres, lsz = int(r.choice([1, 4, 16])), int(r.choice([2, 8, 64])), int(r.choice([2, 16, 64]))
            smips = int(r.integers(1, int(np.log2(sres)) + 2))
            spec = np.concatenate([fuzz_shade.rand_f16(r, (6, max(1, sres >> m), max(1, sres >> m), 4), 0.0).reshape(-1) for m in range(smips)])
            dif, lut = fuzz_shade.rand_f16(r, (6, dres, dres, 4), 0.0), fuzz_shade.rand_f16(r, (lsz, lsz, 2), 0.0)
            env_o = O.host_envmap(dif, spec, sres, smips, lut)
            kd = [dev(dif), dev(spec), dev(lut)]
            env_g = abi.EnvMap(kd[0].data_ptr(), dres, kd[1].data_ptr(), sres, smips, kd[2].data_ptr(), lsz)
            cb = synth.ssr_constants(W, max(H, 1), smips, hdri_yaw=float(r.choice([0.0, 0.3, -2.0])), roughness_threshold=float(r.choice([0.0, 0.2, 1.0])))
            sfmt = int(r.choice([F32, F16]))
            sc = sc.astype(np.float16) if sfmt == F16 else sc
            if r.random() < 0.4:
                n = int(r.integers(1, 6))
                sc[r.integers(0, H, n), r.integers(0, W, n), 3] = r.choice(np.array([0.0, 1.0, -0.0, 2.0, np.nan, 0.2, 0.19995], np.float32), n).astype(sc.dtype)
                depth[r.integers(0, H, n), r.integers(0, W, n)] = r.choice(np.array([0.0, 1.0, 0.5, np.nan, 1.0000001], np.float32), n)
            packed_normals = bool(r.integers(0, 2))
            nfmt = abi.FMT_R10G10B10A2_UNORM if packed_normals else F32
            nrm_o = packed if packed_normals else nf
            nrm_g = dev(packed.view(np.int32)) if packed_normals else dev(nf)
            ofmt = int(r.choice([F32, F16]))
            ref, rr = O.ssr_environment_fallback(sc, sfmt, depth, nrm_o, nfmt, cb, env_o, ofmt, extract_roughness=True)
            got = ctx.ssr_environment_fallback(dev(sc), sfmt, dev(depth), nrm_g, nfmt, cb, env_g, ofmt, extract_roughness=True)
            return _cmp([(got[0], ref), (got[1], rr)]) + (f"seed {seed}: ssr {W}x{H} scene fmt {sfmt} normals {nfmt} out {ofmt} env {(dres, sres, smips, lsz)}",)
        if kind == "reflect":
            W, H, fmt = int(r.choice([1, 64, 333])), int(r.integers(1, 9)), int(r.choice([F32, F16]))
            dt = np.float16 if fmt == F16 else np.float32
            mk = lambda: (r.random((H, W, 4), dtype=np.float32) * np.float32(r.choice([1.0, 100.0, 6e4]))).astype(dt)  # noqa: E731
            refl, scn, bv = mk(), mk(), (mk() if r.random() < 0.5 else None)
            if bv is not None:
                bv[..., 3] = r.choice(np.array([0.0, 1.0, 0.5, 0.25], np.float32), (H, W)).astype(dt)
            if r.random() < 0.4:
                n = int(r.integers(1, 6))
                refl[r.integers(0, H, n), r.integers(0, W, n), r.integers(0, 4, n)] = r.choice(np.array([np.inf, -np.inf, np.nan, -0.0, 0.0], np.float32), n).astype(dt)
            want = O.composite_reflections(refl, scn, fmt, bv)
            got = ctx.composite_reflections(dev(refl), dev(scn), fmt, dev(bv) if bv is not None else None)
            return _cmp([(got, want)]) + (f"seed {seed}: reflections composite {W}x{H} fmt {fmt} bounding volumes {bv is not None}",)
        if kind == "viz":
            W, H = int(r.choice([1, 64, 333])), int(r.integers(1, 9))
            fmt = int(r.choice([F32, F16, R8]))
            if fmt == R8:
                img = r.integers(0, 256, (H, W, 4), dtype=np.uint8)
            else:
                img = (r.random((H, W, 4), dtype=np.float32) * np.float32(r.choice([1.0, 5.0, 2000.0])) - np.float32(r.choice([0.0, 0.5]))).astype(np.float16 if fmt == F16 else np.float32)
            vp = abi.VizParams(int(r.integers(0, 12)), int(r.integers(0, 2)), float(r.choice([0.0, 1.0, 10.0, 100.0])))
            want = O.visualize(img, fmt, vp)
            got = ctx.visualize(dev(img), fmt, vp)
            return _cmp([(got, want)]) + (f"seed {seed}: visualize {W}x{H} fmt {fmt} mode {vp.iDrawMode} unpack {vp.iUnpackNormals} strength {vp.fInputStrength}",)
        # hdr
        w, h = int(r.choice([1, 7, 8, 9, 64, 300, 32767 // 40])), int(r.integers(1, 12))
        rgbe = synth.float_to_rgbe((r.random((h, w, 3), dtype=np.float32) * np.float32(r.choice([1.0, 100.0, 6e4]))).astype(np.float32))
        if r.random() < 0.5:
            rgbe[:, : w // 2] = rgbe[:, :1]                          # long runs
        data = synth.hdr_file_bytes(rgbe)
        cut = None
        if r.random() < 0.33:
            cut = int(r.integers(0, len(data)))
            data = data[:cut]
        what = f"seed {seed}: hdr {w}x{h} bytes {len(data)} cut {cut}"
        try:
            ref = O.hdr_decode(data)
        except ValueError:
            ref = None
        try:
            got = ctx.load_hdr(data)
            torch.cuda.synchronize()
        except (capi.VQHipError, ValueError):
            got = None
        if ref is None or got is None:
            return (0 if (ref is None) == (got is None) else 1), [], what + f" refused oracle {ref is None} product {got is None}"
        return _cmp([(got, ref)]) + (what,)


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
        bad, idx, what = run_case(ctx, seed, dev)
        if bad:
            fails.append(seed)
            print(f"MISMATCH {what}: {bad} channels, first at {np.asarray(idx).tolist()[:2]}", flush=True)
        n += 1
    print(f"fuzz_wide: {n} cases, {len(fails)} failed {fails[:20]}", flush=True)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
