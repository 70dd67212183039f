"""Cases shared by the three users of the reference-output fixtures (tests/golden/ref_outputs.npz):
  * tests/golden/make_ref_fixtures.py  — runs each case through oracle/_ref (the reference's own sources; this container only)
                                         and stores the reference's OUTPUT next to a checksum of the case's inputs;
  * tests/test_ref_fixtures.py (CPU)   — the oracle on the same inputs vs the stored reference output;
  * tests/test_ref_fixtures.py (gpu)   — the HIP product through the C ABI vs the stored reference output.
Inputs are rebuilt from seeds (vqengine_amd.synth), so only outputs are stored; a checksum guards against input drift.
Every case: build() -> inputs, ref(inputs) -> array (needs oracle/_ref), oracle(inputs) -> array, product(ctx, inputs) -> array,
tol = (median, p99, worst, floor) of the relative error, or ("ulp16", max_ulps, max_fraction) / ("u8", ...) / "exact"."""
import ctypes as C
import hashlib

import numpy as np

from tests import oracle_lib as O
from vqengine_amd import abi, synth
from vqengine_amd import scene as scene_mod


def checksum(inputs):
    h = hashlib.sha256()

    def feed(x):
        if isinstance(x, np.ndarray):
            h.update(np.ascontiguousarray(x).tobytes())
        elif isinstance(x, (C.Structure, C.Array)):
            h.update(bytes(x))
        elif isinstance(x, dict):
            for k in sorted(x):
                h.update(str(k).encode()); feed(x[k])
        elif isinstance(x, (list, tuple)):
            for v in x:
                feed(v)
        else:
            h.update(repr(x).encode())
    feed(inputs)
    return h.hexdigest()[:16]


def rel_stats(got, ref, floor):
    r = np.abs(got.astype(np.float64) - ref) / np.maximum(np.abs(ref), floor)
    return float(np.median(r)), float(np.quantile(r, 0.99)), float(r.max())


def check(name, got, ref, tol):
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    if tol == "exact":
        assert np.array_equal(got, ref), name
        return
    if tol[0] == "ulp16":                       # got: stored halfs; ref: float values the reference wrote to its RGBA16F UAV
        with np.errstate(over="ignore"):
            r16 = ref.astype(np.float16)

        def key(h):
            u = h.view(np.uint16).astype(np.int32)
            return np.where(u & 0x8000, -(u & 0x7fff), u)
        d = np.abs(key(got) - key(r16))
        assert d.max() <= tol[1] and np.mean(d != 0) <= tol[2], (name, int(d.max()), float(np.mean(d != 0)))
        return
    if tol[0] == "u8":
        r8 = np.empty(ref.shape, np.uint8)
        rc = np.ascontiguousarray(ref, np.float32)
        O.load().vqo_f32_to_unorm8(rc.ctypes.data, r8.ctypes.data, rc.size)
        d = np.abs(got.astype(np.int32) - r8.astype(np.int32))
        assert d.max() <= tol[1] and np.mean(d != 0) <= tol[2], (name, int(d.max()), float(np.mean(d != 0)))
        return
    assert np.isfinite(got).all(), name
    med, p99, worst = rel_stats(got, ref, tol[3])
    assert med <= tol[0] and p99 <= tol[1] and worst <= tol[2], (name, (med, p99, worst), tol)


# ---------------------------------------------------------------------------------------------------------------------
# shared input builders
# ---------------------------------------------------------------------------------------------------------------------
def small_env():
    eq = synth.equirect(64, 32)
    chain, n = O.mip_chain(eq)
    pre = O.envmap_prefilter(chain, 64, 32, n, 8, 0.1, 16, abi.CONV_WAVE64)
    lut = O.brdf_lut(32, 64, abi.FMT_RG16F)
    return {"diffuse": pre["diffuse_blurred"], "specular": pre["specular"], "spec_res0": 16, "spec_mips": pre["spec_mips"], "lut": lut}


def shadow_scene():
    rng = np.random.default_rng(11)
    pf, _ = synth.per_frame(points=synth.point_lights(3), directional=synth.directional_light(shadowing=1))
    L = pf.Lights
    spots = synth.spot_lights(2, seed=77)
    L.numSpotCasters = 2
    for i in range(2):
        L.spot_casters[i] = spots[i]
    pc = synth.point_lights(1, seed=99)
    pc[0].depthBias = 5e-5
    L.numPointCasters = 1
    L.point_casters[0] = pc[0]

    def mat(scale, tz):
        m = abi.matrix()
        m.m[0][0] = scale; m.m[2][1] = scale; m.m[1][2] = -0.02; m.m[3][2] = tz; m.m[3][3] = 1.0
        return m
    L.shadowViewDirectional = mat(1 / 60.0, 0.5)
    L.shadowViews[0] = mat(1 / 45.0, 0.45)
    L.shadowViews[1] = mat(1 / 70.0, 0.55)
    dmap = rng.random((64, 64), dtype=np.float32) * 0.2 + 0.4
    dmap[:, 32:] = 1.0
    smap = rng.random((5, 32, 32), dtype=np.float32) * 0.3 + 0.35
    smap[:, 16:, :] = 1.0
    pmap = rng.random((5, 6, 16, 16), dtype=np.float32) * 0.5 + 0.05
    pf.f2DirectionalLightShadowMapDimensions = abi.float2(64.0, 64.0)
    pf.f2SpotLightShadowMapDimensions = abi.float2(32.0, 32.0)
    pf.f2PointLightShadowMapDimensions = abi.float2(16.0, 16.0)
    return pf, {"dir": dmap, "spot": smap, "point": pmap}


def unit_normal_gbuffer(w, h, seed):
    gb = [g.copy() for g in synth.gbuffer(w, h, seed=seed)]
    n = gb[1][..., :3].astype(np.float64)
    gb[1][..., :3] = (n / np.linalg.norm(n, axis=-1, keepdims=True)).astype(np.float32)   # see test_ref_pinning.py
    return gb


def host_env(e):
    return O.host_envmap(e["diffuse"], e["specular"], e["spec_res0"], e["spec_mips"], e["lut"]) if e is not None else None


def host_shadow(s):
    return abi.ShadowMaps(s["dir"].ctypes.data, 64, s["spot"].ctypes.data, 32, s["point"].ctypes.data, 16) if s is not None else None


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def dev_env(e, keep):
    from vqengine_amd import capi
    if e is None:
        return None
    d, s, l = _dev(e["diffuse"]), _dev(e["specular"]), _dev(e["lut"])
    keep += [d, s, l]
    return capi.make_envmap(d, s, e["spec_res0"], e["spec_mips"], l)


def dev_shadow(s, keep):
    if s is None:
        return None
    d, sp, p = _dev(s["dir"]), _dev(s["spot"]), _dev(s["point"])
    keep += [d, sp, p]
    return abi.ShadowMaps(d.data_ptr(), 64, sp.data_ptr(), 32, p.data_ptr(), 16)


def hdr_scene(w, h, seed):
    rng = np.random.default_rng(seed)
    img = (rng.random((h, w, 4), dtype=np.float32) ** 3) * rng.choice(np.array([0.05, 1.0, 8.0, 60.0], np.float32), (h, w, 1))
    img[..., 3] = rng.random((h, w), dtype=np.float32)
    img[1, 2, :3] = 0.0
    return img.astype(np.float16)


# ---------------------------------------------------------------------------------------------------------------------
# cases
# ---------------------------------------------------------------------------------------------------------------------
class Case:
    def __init__(self, name, build, ref, oracle, product, tol):
        self.name, self.build, self.ref, self.oracle, self.product, self.tol = name, build, ref, oracle, product, tol


CASES = []
STAT = (3e-7, 3e-5, 6e-3, 1e-5)


def _forward_case(kind):
    W, H = 64, 32

    def build():
        gb = unit_normal_gbuffer(W, H, 5)
        env = sh = None
        pv = synth.per_view(W, H)
        if kind == "ambient":
            pf, _ = synth.per_frame()
        elif kind == "points64":
            pf, _ = synth.per_frame(points=synth.point_lights(64))
        elif kind == "spots8":
            pf, _ = synth.per_frame(spots=synth.spot_lights(8))
        elif kind == "directional":
            pf, _ = synth.per_frame(directional=synth.directional_light())
        elif kind in ("mixed_env", "env_diffuse_only"):
            env = small_env()
            pf, _ = synth.per_frame(points=synth.point_lights(12), spots=synth.spot_lights(3), directional=synth.directional_light(), hdri_offset=0.7)
            pv = synth.per_view(W, H, max_env_lod=env["spec_mips"] - 1, diffuse_only=int(kind == "env_diffuse_only"))
        else:
            pf, sh = shadow_scene()
        return {"gb": gb, "pf": pf, "pv": pv, "env": env, "shadow": sh}

    def ref(i):
        from tests import ref_lib as R
        return R.forward_from_gbuffer(i["gb"], i["pf"], i["pv"], env=host_env(i["env"]), shadow=host_shadow(i["shadow"]))

    def oracle(i):
        return O.forward_lighting(i["gb"], i["pf"], i["pv"], abi.FMT_RGBA32F, env=host_env(i["env"]), shadow=host_shadow(i["shadow"]))

    def product(ctx, i):
        keep = []
        out = ctx.forward_lighting([_dev(g) for g in i["gb"]], i["pf"], i["pv"], out_fmt=abi.FMT_RGBA32F, env=dev_env(i["env"], keep),
                                   shadow=dev_shadow(i["shadow"], keep))
        return out.cpu().numpy()
    # casters: a PCF tap / range test on its threshold flips a pixel by 1/25 or 1/20 of one light -> a few loose pixels
    tol = (3e-7, 2e-3, 0.5, 1e-5) if kind == "casters_pcf" else STAT
    return Case("forward_" + kind, build, ref, oracle, product, tol)


for _k in ("ambient", "points64", "spots8", "directional", "mixed_env", "env_diffuse_only", "casters_pcf"):
    CASES.append(_forward_case(_k))


def _psmain_case():
    W, H, NM = 48, 32, 5

    def build():
        ip = synth.interpolants(W, H, NM)
        datas, texsets = synth.material_set(NM, max_dim=64)
        env = small_env()
        pf, _ = synth.per_frame(points=synth.point_lights(10, seed=3), spots=synth.spot_lights(2, seed=3), directional=synth.directional_light(),
                                hdri_offset=-0.4)
        pv = synth.per_view(W, H, max_env_lod=env["spec_mips"] - 1)
        return {"ip": ip, "datas": datas, "tex": texsets, "ssao": synth.ssao_image(W, H), "env": env, "pf": pf, "pv": pv}

    def host_mats(i):
        hc = []
        for ts in i["tex"]:
            cs = {}
            for slot, img in ts.items():
                chain, n = O.mip_chain_rgba8(img)
                cs[slot] = (chain, img.shape[1], img.shape[0], n)
            hc.append(cs)
        return O.host_materials(i["datas"], hc)

    def valid(i):
        idx = i["ip"][2][..., 3].view(np.int32)
        return (idx >= 0) & (idx < NM)

    def ref(i):
        from tests import ref_lib as R
        out = R.forward_psmain(i["ip"], host_mats(i), i["pf"], i["pv"], ssao=i["ssao"], env=host_env(i["env"]))
        return out[valid(i)]

    def oracle(i):
        gb = O.gbuffer_from_materials(i["ip"], host_mats(i), i["pf"].fAmbientLightingFactor, ssao=i["ssao"])
        return O.forward_lighting(gb, i["pf"], i["pv"], abi.FMT_RGBA32F, env=host_env(i["env"]))[valid(i)]

    def product(ctx, i):
        keep = []
        dm = (abi.MaterialDesc * NM)()
        for k, (d, ts) in enumerate(zip(i["datas"], i["tex"])):
            dm[k].data = d
            for slot, img in ts.items():
                chain, n = ctx.mip_chain_rgba8(_dev(img))
                keep.append(chain)
                setattr(dm[k], slot, abi.Texture2D(chain.data_ptr(), img.shape[1], img.shape[0], n, 0))
        gb = ctx.gbuffer_from_materials([_dev(p) for p in i["ip"]], dm, i["pf"].fAmbientLightingFactor, _dev(i["ssao"]))
        out = ctx.forward_lighting(gb, i["pf"], i["pv"], out_fmt=abi.FMT_RGBA32F, env=dev_env(i["env"], keep))
        return out.cpu().numpy()[valid(i)]
    return Case("psmain_textured", build, ref, oracle, product, (3e-7, 1e-4, 6e-3, 1e-5))


CASES.append(_psmain_case())


def _lut_case():
    rows = [0, 20, 77, 512, 1023]
    xs = np.concatenate([[0, 1, 1022, 1023], np.arange(5, 1024, 41)]).astype(np.int32)

    def build():
        return {"rows": rows, "xs": xs}

    def ref(i):
        from tests import ref_lib as R
        return np.stack([R.brdf_lut_texels(xs, np.full_like(xs, y)) for y in rows])

    def oracle(i):
        return np.stack([O.brdf_lut(1024, 2048, abi.FMT_RG32F, rows=(y, y + 1))[0][xs] for y in rows])

    def product(ctx, i):
        lut = ctx.brdf_lut(1024, 2048, abi.FMT_RG32F).cpu().numpy()
        return np.stack([lut[y][xs] for y in rows])
    return Case("brdf_lut_1024x2048_rows", build, ref, oracle, product, (1e-6, 1e-2, 5e-2, 1e-4))    # rows 0, 20: the sqrt(x/x) corner
    # (rows >= 64 alone: median 1e-7, max 1e-6 — tests/test_ref_pinning.py asserts that split)


CASES.append(_lut_case())


def _conv_cases():
    def build():
        eq = synth.equirect(64, 32)
        chain, n = O.mip_chain(eq)
        return {"chain": chain, "n": n}

    def ref_d(i):
        from tests import ref_lib as R
        return R.conv_diffuse(i["chain"], 64, 32, i["n"], 3)[..., :3]

    def ref_s(i):
        from tests import ref_lib as R
        mips = abi.specular_mip_count(16)
        return np.concatenate([R.conv_specular_mip(i["chain"], 64, 32, i["n"], 16 >> m, float(np.float32(m) / np.float32(mips - 1)), m).reshape(-1, 4)
                               for m in range(mips)])[:, :3]
    CASES.append(Case("conv_diffuse_step0.010", build, ref_d,
                      lambda i: O.conv_diffuse(i["chain"], 64, 32, i["n"], 3, 0.010, abi.CONV_SEQUENTIAL, abi.FMT_RGBA32F)[..., :3],
                      lambda ctx, i: ctx.conv_diffuse(_dev(i["chain"]), 64, 32, i["n"], 3, 0.010, abi.CONV_SEQUENTIAL, abi.FMT_RGBA32F).cpu().numpy()[..., :3],
                      (3e-6, 1e-4, 1e-4, 1e-4)))
    CASES.append(Case("conv_diffuse_step0.010_wave64", build, ref_d,
                      lambda i: O.conv_diffuse(i["chain"], 64, 32, i["n"], 3, 0.010, abi.CONV_WAVE64, abi.FMT_RGBA32F)[..., :3],
                      lambda ctx, i: ctx.conv_diffuse(_dev(i["chain"]), 64, 32, i["n"], 3, 0.010, abi.CONV_WAVE64, abi.FMT_RGBA32F).cpu().numpy()[..., :3],
                      (5e-5, 5e-4, 5e-4, 1e-4)))
    CASES.append(Case("conv_specular_16", build, ref_s,
                      lambda i: O.conv_specular(i["chain"], 64, 32, i["n"], 16, abi.CONV_SEQUENTIAL, abi.FMT_RGBA32F)[0][:, :3],
                      lambda ctx, i: ctx.conv_specular(_dev(i["chain"]), 64, 32, i["n"], 16, abi.CONV_SEQUENTIAL, abi.FMT_RGBA32F)[0].cpu().numpy()[:, :3],
                      (5e-6, 2e-3, 2e-2, 1e-4)))       # mip 0 (Roughness 0) is the sqrt(x/x) corner


_conv_cases()


def _post_cases():
    def build_img():
        return {"img": hdr_scene(40, 24, 9)}
    for d in (0, 1):
        def ref(i, d=d):
            from tests import ref_lib as R
            return R.blur_pass(i["img"].astype(np.float32), d)[..., :3]

        def product(ctx, i, d=d):
            f = ctx.gaussian_blur_x if d == 0 else ctx.gaussian_blur_y
            return f(_dev(i["img"]), abi.FMT_RGBA16F).cpu().numpy()[..., :3]
        CASES.append(Case(f"blur_{'xy'[d]}_rgba16f", build_img, ref, lambda i, d=d: O.blur_pass(i["img"], abi.FMT_RGBA16F, d)[..., :3], product,
                          ("ulp16", 1, 3e-3)))
    for curve, space, gamma in ((abi.DISPLAY_CURVE_SRGB, abi.COLOR_SPACE_REC_709, 1), (abi.DISPLAY_CURVE_SRGB, abi.COLOR_SPACE_REC_709, 0),
                                (abi.DISPLAY_CURVE_ST2084, abi.COLOR_SPACE_REC_709, 1), (abi.DISPLAY_CURVE_ST2084, abi.COLOR_SPACE_REC_2020, 1),
                                (abi.DISPLAY_CURVE_LINEAR, abi.COLOR_SPACE_REC_709, 1)):
        def build(curve=curve, space=space, gamma=gamma):
            return {"img": hdr_scene(40, 24, 10), "p": abi.TonemapperParams(space, curve, 200.0, gamma)}

        def ref(i):
            from tests import ref_lib as R
            return R.tonemap(i["img"].astype(np.float32), i["p"])
        st = curve == abi.DISPLAY_CURVE_ST2084
        CASES.append(Case(f"tonemap_{curve}_{space}_{gamma}_f32", build, ref,
                          lambda i: O.tonemap(i["img"], abi.FMT_RGBA16F, abi.FMT_RGBA32F, params=i["p"]),
                          lambda ctx, i: ctx.tonemap(_dev(i["img"]), abi.FMT_RGBA16F, abi.FMT_RGBA32F, params=i["p"]).cpu().numpy(),
                          (2e-7, 3e-5 if st else 2e-6, 1e-4 if st else 2e-5, 1e-4)))
        if curve == abi.DISPLAY_CURVE_SRGB:
            CASES.append(Case(f"tonemap_{curve}_{space}_{gamma}_unorm8", build, ref,
                              lambda i: O.tonemap(i["img"], abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM, params=i["p"]),
                              lambda ctx, i: ctx.tonemap(_dev(i["img"]), abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM, params=i["p"]).cpu().numpy(),
                              ("u8", 1, 3e-3)))

    def build_sky():
        return {"eq": synth.equirect(128, 64), "sp": scene_mod.skydome_params(2.5, 0.6, 1.1, 1.0, 48, 27)}

    def ref_sky(i):
        from tests import ref_lib as R
        return R.skydome(i["eq"], i["sp"], 48, 27)

    def prod_sky(ctx, i):
        import torch
        color = torch.zeros((27, 48, 4), dtype=torch.float32, device="cuda")
        ctx.skydome(_dev(i["eq"]), i["sp"], color, abi.FMT_RGBA32F)
        return color.cpu().numpy()
    CASES.append(Case("skydome", build_sky, ref_sky, lambda i: O.skydome(i["eq"], i["sp"], np.zeros((27, 48, 4), np.float32), abi.FMT_RGBA32F),
                      prod_sky, (1e-7, 5e-3, 5e-2, 1e-3)))        # an ulp of uv moves the 8-bit filter fraction of a few pixels by one step
    for mode in range(10):
        def build_viz(mode=mode):
            img = hdr_scene(16, 12, 12).astype(np.float32)
            img[..., 0] = np.clip(img[..., 0], 0, 1)
            return {"img": img, "p": abi.VizParams(mode, mode & 1, 2.5)}

        def ref_viz(i):
            from tests import ref_lib as R
            return R.visualize(i["img"], i["p"])
        CASES.append(Case(f"visualize_mode{mode}", build_viz, ref_viz, lambda i: O.visualize(i["img"], abi.FMT_RGBA32F, i["p"]),
                          lambda ctx, i: ctx.visualize(_dev(i["img"]), abi.FMT_RGBA32F, i["p"]).cpu().numpy(), (1e-7, 2e-5, 2e-3, 1e-6)))


_post_cases()


def _fsr_con_case():
    sizes = [(1280, 720, 1920, 1080), (2560, 1440, 3840, 2160), (1477, 831, 1920, 1080), (1, 1, 1, 1), (3840, 2160, 3840, 2160), (640, 360, 1137, 777)]
    stops = [0.0, 0.2, 0.25, 0.87, 1.0, 1.5, 2.0, 3.3]

    def build():
        return {"sizes": sizes, "stops": stops}

    def ref(i):
        from tests import ref_lib as R
        return np.concatenate([R.fsr_easu_con(*s) for s in sizes] + [R.fsr_rcas_con(s) for s in stops])

    def oracle(i):
        return np.concatenate([O.fsr_easu_con(*s) for s in sizes] + [O.fsr_rcas_con(s) for s in stops])

    def product(ctx, i):
        from vqengine_amd import capi
        return np.concatenate([np.array(list(capi.fsr_easu_con(*s)), np.uint32) for s in sizes] +
                              [np.array(list(capi.fsr_rcas_con(s)), np.uint32) for s in stops])
    return Case("fsr_constant_blocks", build, ref, oracle, product, "exact")


CASES.append(_fsr_con_case())


def _ccon(a):
    return (C.c_uint32 * len(a))(*[int(v) for v in a])


def _fsr_filter_cases():
    def image(iw, ih, seed):
        rng = np.random.default_rng(seed)
        img = rng.random((ih, iw, 4), dtype=np.float32)
        yy, xx = np.mgrid[0:ih, 0:iw]
        img[..., 0] = 0.5 + 0.5 * np.sin(xx * 0.7 + yy * 0.3)
        img[ih // 3:, :, 1] = (xx[ih // 3:] > iw // 2) * 0.9
        img[2, 3, :3] = 0.0
        img[..., 3] = 1.0
        return img
    for iw, ih, ow, oh in ((48, 27, 72, 41), (33, 17, 66, 34), (17, 9, 64, 33)):
        def build(iw=iw, ih=ih, ow=ow, oh=oh):
            return {"img": image(iw, ih, iw * 131 + oh), "out": (ow, oh), "con": O.fsr_easu_con(iw, ih, ow, oh)}

        def ref(i):
            from tests import ref_lib as R
            return R.fsr_easu(i["img"], i["out"][0], i["out"][1], i["con"]).view(np.uint32)
        CASES.append(Case(f"fsr_easu_{iw}x{ih}_to_{ow}x{oh}", build, ref,
                          lambda i: O.fsr_easu(i["img"], abi.FMT_RGBA32F, i["out"][0], i["out"][1], abi.FMT_RGBA32F, con=i["con"])[..., :3].copy().view(np.uint32),
                          lambda ctx, i: ctx.fsr_easu(_dev(i["img"]), abi.FMT_RGBA32F, i["out"][0], i["out"][1], abi.FMT_RGBA32F,
                                                      con=_ccon(i["con"])).cpu().numpy()[..., :3].copy().view(np.uint32), "exact"))
    for stops in (0.0, 0.2, 1.3):
        def build_r(stops=stops):
            return {"img": image(45, 31, 77), "con": O.fsr_rcas_con(stops)}

        def ref_r(i):
            from tests import ref_lib as R
            return R.fsr_rcas(i["img"], i["con"]).view(np.uint32)
        CASES.append(Case(f"fsr_rcas_stops{stops}", build_r, ref_r,
                          lambda i: O.fsr_rcas(i["img"], abi.FMT_RGBA32F, abi.FMT_RGBA32F, con=i["con"])[..., :3].copy().view(np.uint32),
                          lambda ctx, i: ctx.fsr_rcas(_dev(i["img"]), abi.FMT_RGBA32F, abi.FMT_RGBA32F, con=_ccon(i["con"])).cpu().numpy()[..., :3].copy().view(np.uint32),
                          "exact"))


_fsr_filter_cases()


def _mip_cases():
    def build_f():
        rng = np.random.default_rng(64032)
        return {"l0": (rng.random((32, 64, 4), dtype=np.float32) * 20).astype(np.float32)}

    def build_u():
        rng = np.random.default_rng(12816)
        return {"l0": rng.integers(0, 256, (128, 16, 4), dtype=np.uint8)}

    def ref(i):
        from tests import ref_lib as R
        return np.concatenate([lv.reshape(-1, 4) for lv in R.mip_chain(i["l0"])])

    def n_px(i):                 # texels of the levels the reference function produces (both dimensions >= 2 going in)
        h, w = i["l0"].shape[:2]
        n = 0
        while h >= 2 and w >= 2:
            h, w = h // 2, w // 2
            n += h * w
        return n

    def cut(chain, i):
        h, w = i["l0"].shape[:2]
        return chain[w * h: w * h + n_px(i)]
    CASES.append(Case("mipimage_min_rgba32f", build_f, lambda i: ref(i).view(np.uint32), lambda i: cut(O.mip_chain(i["l0"])[0], i).view(np.uint32),
                      lambda ctx, i: cut(ctx.mip_chain(_dev(i["l0"]))[0].cpu().numpy(), i).view(np.uint32), "exact"))
    CASES.append(Case("mipimage_box_rgba8", build_u, ref, lambda i: cut(O.mip_chain_rgba8(i["l0"])[0], i),
                      lambda ctx, i: cut(ctx.mip_chain_rgba8(_dev(i["l0"]))[0].cpu().numpy(), i), "exact"))


_mip_cases()
BY_NAME = {c.name: c for c in CASES}
