"""GPU parity: every HIP kernel, called through the C ABI (libvqhip.so), against the CPU oracle on the same
seeded inputs. Bar: BIT-EXACT in every output format (fp32 before storage rounding, and the reference's
RGBA16F / RG16F / RGBA8 storage formats) — which implies the north-star tolerance of <= 1 ULP per channel in
the reference storage format with margin. NaNs compare equal to NaNs."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import oracle_lib as O
from vqengine_amd import abi, capi, synth

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def assert_bits(gpu_t, ref_np, what):
    got = gpu_t.cpu().numpy()
    n, idx = O.bits_equal(got, ref_np)
    assert n == 0, f"{what}: {n} mismatching elements of {got.size}; first at {idx.tolist()}: " \
                   f"gpu={[got[tuple(i)] for i in idx]} ref={[ref_np[tuple(i)] for i in idx]}"


# ---------------------------------------------------------------------------------------------------
# small shared IBL inputs (equirect -> mips -> prefilter -> LUT), computed once by the ORACLE and once by the GPU
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def env_small(ctx):
    eq = synth.equirect(128, 64)
    chain_o, n = O.mip_chain(eq)
    chain_g, n_g = ctx.mip_chain(dev(eq))
    assert n == n_g
    pre_o = O.envmap_prefilter(chain_o, 128, 64, n, 16, 0.05, 32, abi.CONV_SEQUENTIAL)
    pre_g = ctx.envmap_prefilter(chain_g, 128, 64, n, 16, 0.05, 32, abi.CONV_SEQUENTIAL)
    lut_o = O.brdf_lut(64, 128, abi.FMT_RG16F)
    lut_g = ctx.brdf_lut(64, 128, abi.FMT_RG16F)
    return dict(eq=eq, n=n, chain_o=chain_o, chain_g=chain_g, pre_o=pre_o, pre_g=pre_g, lut_o=lut_o, lut_g=lut_g)


def test_mip_chain_min(env_small):
    assert_bits(env_small["chain_g"], env_small["chain_o"], "min-filter mip chain")


def test_brdf_lut(ctx, env_small):
    assert_bits(env_small["lut_g"], env_small["lut_o"], "BRDF LUT RG16F 64^2 x128")
    assert_bits(ctx.brdf_lut(32, 256, abi.FMT_RG32F), O.brdf_lut(32, 256, abi.FMT_RG32F), "BRDF LUT RG32F 32^2 x256")


def test_brdf_lut_reference_size_rows(ctx):
    """1024^2 x 2048 (the reference's size, Renderer.cpp:895-900,1026-1032): full LUT on the GPU, 6 rows checked."""
    lut = ctx.brdf_lut(1024, 2048, abi.FMT_RG16F).cpu().numpy()
    for y in (0, 1, 511, 777, 1023):
        ref = O.brdf_lut(1024, 2048, abi.FMT_RG16F, rows=(y, y + 1))
        n, idx = O.bits_equal(lut[y:y + 1], ref)
        assert n == 0, (y, n, idx)
    f = lut.astype(np.float32)
    assert np.isfinite(f).all() and (f >= 0).all() and (f[..., 0] + f[..., 1] <= 1.001).all()


def test_ibl_reference_sizes_cfg4(ctx):
    """BASELINE config 4 at full size: 2048^2 equirect -> 12-level min-filter chain, diffuse 6x64^2 at step 0.010 (99 382 taps per
    texel), 7-mip specular 128^2. The GPU computes everything; the oracle checks the whole mip chain, 96 diffuse texels spread
    over the six faces (each a full 99 382-tap integral) and the complete specular cube."""
    eq = synth.equirect(2048, 2048)
    chain_g, n = ctx.mip_chain(dev(eq))
    chain_o, n_o = O.mip_chain(eq)
    assert n == n_o == 12
    assert_bits(chain_g, chain_o, "2048^2 mip chain")
    for order in (abi.CONV_SEQUENTIAL, abi.CONV_WAVE64):             # the reference's order (default) and the optional 64-lane order
        diff_g = ctx.conv_diffuse(chain_g, 2048, 2048, n, 64, 0.010, order, abi.FMT_RGBA16F).cpu().numpy().reshape(-1, 4)
        for t0 in (0, 4095, 4096 + 1234, 3 * 4096 + 4000, 5 * 4096 + 64 * 63, 6 * 4096 - 16):
            ref = O.conv_diffuse(chain_o, 2048, 2048, n, 64, 0.010, order, abi.FMT_RGBA16F, t0=t0, t1=t0 + 16).reshape(-1, 4)
            n_bad, idx = O.bits_equal(diff_g[t0:t0 + 16], ref[t0:t0 + 16])
            assert n_bad == 0, (order, t0, n_bad, idx)
        spec_g, mips = ctx.conv_specular(chain_g, 2048, 2048, n, 128, order, abi.FMT_RGBA16F)
        spec_o, mips_o = O.conv_specular(chain_o, 2048, 2048, n, 128, order, abi.FMT_RGBA16F)
        assert mips == mips_o == 7
        assert_bits(spec_g, spec_o, f"specular 128^2 x 7 mips from the 2048^2 equirect, order {order}")
    f = diff_g.astype(np.float32)
    assert np.isfinite(f).all() and (f[:, :3] >= 0).all() and np.all(f[:, 3] == 1.0)


def test_cfg4_env_matches_golden(ctx):
    """The WHOLE of BASELINE config 4 (2048^2 equirect -> 6x64^2 diffuse irradiance at 99 382 taps per texel -> blur -> 128^2 x 7-mip
    specular; 1024^2 x 2048-sample BRDF LUT) computed by the HIP product == tests/golden/cfg4_env.npz, the same thing computed by the CPU
    oracle (tests/golden/make_cfg4_env.py), bit for bit in the reference's storage formats: every texel, not a sample of them."""
    from tests import ref_cases
    g = ref_cases.cfg4_env()
    eq = synth.equirect(2048, 2048)
    chain, n = ctx.mip_chain(dev(eq))
    pre = ctx.envmap_prefilter(chain, 2048, 2048, n, 64, 0.010, 128)
    assert pre["spec_mips"] == g["spec_mips"] == 7
    assert_bits(pre["diffuse_blurred"], g["diffuse"], "cfg4 diffuse irradiance (blurred) 6x64^2")
    assert_bits(pre["specular"], g["specular"], "cfg4 specular 128^2 x 7 mips")
    assert_bits(ctx.brdf_lut(1024, 2048, abi.FMT_RG16F), g["lut"], "cfg4 BRDF LUT 1024^2 x 2048")


def test_envmap_prefilter(env_small):
    for k in ("diffuse_unblurred", "diffuse_blurred", "specular"):
        assert_bits(env_small["pre_g"][k], env_small["pre_o"][k], f"prefilter {k}")


@pytest.mark.parametrize("order", [abi.CONV_SEQUENTIAL, abi.CONV_WAVE64])
@pytest.mark.parametrize("fmt", [abi.FMT_RGBA32F, abi.FMT_RGBA16F])
def test_conv_orders(ctx, env_small, order, fmt):
    e = env_small
    d_g = ctx.conv_diffuse(e["chain_g"], 128, 64, e["n"], 8, 0.05, order, fmt)
    d_o = O.conv_diffuse(e["chain_o"], 128, 64, e["n"], 8, 0.05, order, fmt)
    assert_bits(d_g, d_o, f"conv_diffuse order={order} fmt={fmt}")
    s_g, _ = ctx.conv_specular(e["chain_g"], 128, 64, e["n"], 16, order, fmt)
    s_o, _ = O.conv_specular(e["chain_o"], 128, 64, e["n"], 16, order, fmt)
    assert_bits(s_g, s_o, f"conv_specular order={order} fmt={fmt}")


# ---------------------------------------------------------------------------------------------------
# forward lighting
# ---------------------------------------------------------------------------------------------------
def _envs(e):
    pre_o, pre_g = e["pre_o"], e["pre_g"]
    env_o = O.host_envmap(pre_o["diffuse_blurred"], pre_o["specular"], 32, pre_o["spec_mips"], e["lut_o"])
    env_g = capi.make_envmap(pre_g["diffuse_blurred"], pre_g["specular"], 32, pre_g["spec_mips"], e["lut_g"])
    return env_o, env_g


@pytest.mark.parametrize("fmt", [abi.FMT_RGBA32F, abi.FMT_RGBA16F])
def test_forward_point_lights_cfg2_shape(ctx, fmt):
    """BASELINE cfg2 shape at reduced size: 16 point lights, no IBL, no casters, ragged width (not a multiple of 256)."""
    W, H = 333, 61
    gb = synth.gbuffer(W, H)
    pf, extra = synth.per_frame(points=synth.point_lights(16))
    pv = synth.per_view(W, H)
    ref = O.forward_lighting(gb, pf, pv, fmt)
    got = ctx.forward_lighting([dev(g) for g in gb], pf, pv, out_fmt=fmt)
    assert_bits(got, ref, f"forward 16 point lights fmt={fmt}")


def test_forward_all_light_types_and_ibl(ctx, env_small):
    """cfg3 shape at reduced size: 64 point + 6 spot + directional + IBL (specular mips, LUT, diffuse cube)."""
    W, H = 256, 48
    gb = synth.gbuffer(W, H, seed=0x6400)
    pf, extra = synth.per_frame(points=synth.point_lights(64, seed=0x6400), spots=synth.spot_lights(6),
                                directional=synth.directional_light(), hdri_offset=0.3)
    env_o, env_g = _envs(env_small)
    pv = synth.per_view(W, H, max_env_lod=env_small["pre_o"]["spec_mips"])
    for fmt in (abi.FMT_RGBA32F, abi.FMT_RGBA16F):
        ref = O.forward_lighting(gb, pf, pv, fmt, env=env_o)
        got = ctx.forward_lighting([dev(g) for g in gb], pf, pv, out_fmt=fmt, env=env_g)
        assert_bits(got, ref, f"forward all lights + IBL fmt={fmt}")
    pv.EnvironmentMapDiffuseOnlyIllumination = 1          # GFXSettings.Reflections == SSR path, SceneRendering.cpp:464
    ref = O.forward_lighting(gb, pf, pv, abi.FMT_RGBA32F, env=env_o)
    got = ctx.forward_lighting([dev(g) for g in gb], pf, pv, out_fmt=abi.FMT_RGBA32F, env=env_g)
    assert_bits(got, ref, "forward diffuse-only IBL")


def test_forward_cube_seams_and_directions(ctx, env_small):
    """IBL-only shading with normals/view vectors aimed at cube edges and corners (seamless filtering paths)."""
    rng = np.random.default_rng(5)
    n = 4096
    d = rng.choice([-1.0, 1.0], (n, 3)) * np.where(rng.random((n, 3)) < 0.5, 1.0, rng.random((n, 3)))
    d += rng.normal(0, 0.02, (n, 3))
    gb = [np.zeros((1, n, 4), np.float32) for _ in range(4)]
    gb[0][0, :, :3] = rng.normal(0, 1, (n, 3)); gb[0][0, :, 3] = 0.03
    gb[1][0, :, :3] = d / np.linalg.norm(d, axis=1, keepdims=True); gb[1][0, :, 3] = rng.random(n)
    gb[2][0] = rng.random((n, 4)); gb[3][0] = 0
    pf, _ = synth.per_frame(hdri_offset=0.0)
    env_o, env_g = _envs(env_small)
    pv = synth.per_view(n, 1, camera=(0.0, 0.0, 0.0), max_env_lod=env_small["pre_o"]["spec_mips"])
    ref = O.forward_lighting(gb, pf, pv, abi.FMT_RGBA32F, env=env_o)
    got = ctx.forward_lighting([dev(g) for g in gb], pf, pv, out_fmt=abi.FMT_RGBA32F, env=env_g)
    assert_bits(got, ref, "forward IBL cube seams")


def test_forward_extra_point_lights_cfg5_shape(ctx):
    """cfg5 shape: 256 point lights = 100 in the cbuffer array + 156 through the extension array."""
    W, H = 192, 16
    gb = synth.gbuffer(W, H, seed=0x25600)
    pf, extra = synth.per_frame(points=synth.point_lights(256, seed=0x25600))
    assert pf.Lights.numPointLights == 100 and len(extra) == 156
    pv = synth.per_view(W, H)
    ref = O.forward_lighting(gb, pf, pv, abi.FMT_RGBA16F, extra_point=extra)
    got = ctx.forward_lighting([dev(g) for g in gb], pf, pv, out_fmt=abi.FMT_RGBA16F, extra_point=extra)
    assert_bits(got, ref, "forward 256 point lights")


def _shadow_setup(W, H):
    """Default-scene-like caster set (cfg1 substitute): shadowing directional + 2 spot casters + 1 point
    caster with synthetic shadow maps: half-plane occluders so both lit and shadowed PCF taps occur."""
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

    def mat(scale, tz):       # simple "orthographic" light-space matrix: xz plane -> clip xy, y -> depth
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
    return pf, dmap, smap, pmap


def test_forward_range_cull_boundary(ctx):
    """`if (D < range)` (Lighting.hlsl:318) is evaluated in the kernel as dd < rangeSq with a host-made exact threshold: pixels placed
    within a few ulps of the range sphere of every light must be lit / culled exactly like the oracle's sqrt-and-compare."""
    r = np.random.default_rng(1234)
    ranges = np.concatenate([r.uniform(0.5, 200.0, 12), [1.0, 2.0, 1e-3, 3e4]]).astype(np.float32)
    K = 48
    W, H = 2 * K + 1, len(ranges)
    gb = [g.copy() for g in synth.gbuffer(W, H, seed=0xB0B)]
    pts = synth.point_lights(len(ranges), seed=0xB0B)
    for y, rg in enumerate(ranges):
        pts[y].position.x, pts[y].position.y, pts[y].position.z = 0.0, 0.0, 0.0
        pts[y].range = float(rg)
        x = np.float32(rg)
        xs = [x]
        for _ in range(K):
            xs.append(np.nextafter(xs[-1], np.float32(np.inf), dtype=np.float32))
        lo = [x]
        for _ in range(K):
            lo.append(np.nextafter(lo[-1], np.float32(0), dtype=np.float32))
        row = np.array(lo[:0:-1] + xs, np.float32)                      # K below, range itself, K above
        gb[0][y, :, 0] = row; gb[0][y, :, 1] = 0.0; gb[0][y, :, 2] = 0.0
        gb[1][y, :, :3] = (-1.0, 0.0, 0.0)                               # facing the light at the origin
    pf, extra = synth.per_frame(points=pts)
    pv = synth.per_view(W, H)
    ref = O.forward_lighting(gb, pf, pv, abi.FMT_RGBA32F)
    got = ctx.forward_lighting([dev(g) for g in gb], pf, pv, out_fmt=abi.FMT_RGBA32F)
    assert_bits(got, ref, "range-cull boundary")
    # the rows really straddle the boundary: the own light's contribution switches off somewhere along each row
    for y in range(H):
        lit_first, lit_last = ref[y, 0, :3].sum(), ref[y, -1, :3].sum()
        assert lit_first != lit_last


def test_forward_shadow_casters_pcf(ctx):
    W, H = 200, 40
    gb = synth.gbuffer(W, H, seed=0x5AD0)
    pf, dmap, smap, pmap = _shadow_setup(W, H)
    pv = synth.per_view(W, H)
    sm_o = abi.ShadowMaps(dmap.ctypes.data, 64, smap.ctypes.data, 32, pmap.ctypes.data, 16)
    dg, sg, pg = dev(dmap), dev(smap), dev(pmap)
    sm_g = abi.ShadowMaps(dg.data_ptr(), 64, sg.data_ptr(), 32, pg.data_ptr(), 16)
    ref = O.forward_lighting(gb, pf, pv, abi.FMT_RGBA32F, shadow=sm_o)
    got = ctx.forward_lighting([dev(g) for g in gb], pf, pv, out_fmt=abi.FMT_RGBA32F, shadow=sm_g)
    assert_bits(got, ref, "forward with PCF shadow casters")
    unshadowed = O.forward_lighting(gb, synth.per_frame(points=synth.point_lights(3), directional=synth.directional_light())[0], pv, abi.FMT_RGBA32F)
    assert (np.abs(ref[..., :3] - unshadowed[..., :3]) > 1e-6).mean() > 0.05, "shadow maps had no effect: the PCF path was not exercised"


def test_forward_edge_cases(ctx):
    """Degenerate inputs: zero lights, zero normal (NaN propagation), light exactly at the pixel, roughness 0/1, 1x1 image."""
    pf0, _ = synth.per_frame()
    pv = synth.per_view(1, 1)
    gb = [np.zeros((1, 1, 4), np.float32) for _ in range(4)]
    gb[1][0, 0] = (0, 1, 0, 0.5); gb[2][0, 0] = (0.5, 0.25, 0.125, 0.0); gb[0][0, 0, 3] = 0.05; gb[3][0, 0] = (1, 2, 3, 0.5)
    got = ctx.forward_lighting([dev(g) for g in gb], pf0, pv, out_fmt=abi.FMT_RGBA32F).cpu().numpy()
    np.testing.assert_array_equal(got[0, 0], np.array([0.5 * 0.05 + 0.5, 0.25 * 0.05 + 1.0, 0.125 * 0.05 + 1.5, 0.5], np.float32))
    W = 64
    gb = synth.gbuffer(W, 2, seed=3)
    pts = synth.point_lights(4, seed=3)
    gb[1][0, 0, :3] = 0.0                                   # zero normal -> normalize gives NaN
    gb[0][0, 1, :3] = (pts[0].position.x, pts[0].position.y, pts[0].position.z)   # D == 0
    gb[1][0, 2, 3] = 0.0; gb[1][0, 3, 3] = 1.0              # roughness extremes
    gb[2][0, 4, 3] = 1.0; gb[2][0, 5, 3] = 0.0              # metalness extremes
    pf, _ = synth.per_frame(points=pts)
    pv = synth.per_view(W, 2)
    ref = O.forward_lighting(gb, pf, pv, abi.FMT_RGBA32F)
    got = ctx.forward_lighting([dev(g) for g in gb], pf, pv, out_fmt=abi.FMT_RGBA32F)
    assert_bits(got, ref, "forward edge cases")
    # zero normal: normalize() yields NaN but every use goes through saturate()/max(0,.) which map NaN -> 0 (HLSL semantics)
    assert np.isfinite(ref[0, 0, :3]).all()
    # D == 0: rcp(0) = inf and 0*inf = NaN propagate exactly as in the HLSL
    assert np.isnan(ref[0, 1, :3]).all()


def test_forward_full_size_properties(ctx):
    """BASELINE cfg3 size 3840x2160, 64 lights: (i) a 24-row crop is bit-exact vs the oracle; (ii) additivity of lights
    in a size-independent form: shading with lights A then B separately sums to shading with A+B within fp32 rounding;
    (iii) RGBA16F output == RNE(fp32 output)."""
    W, H = 3840, 2160
    rows = (1000, 1024)
    gb_crop = synth.gbuffer_rows(W, H, rows[0], rows[1], seed=0x6400)
    pts = synth.point_lights(64, seed=0x6400)
    pf, _ = synth.per_frame(points=pts)
    pv = synth.per_view(W, H)
    gb_full = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(4)]
    for r0 in range(0, H, 240):
        part = synth.gbuffer_rows(W, H, r0, r0 + 240, seed=0x6400)
        for k in range(4):
            gb_full[k][r0:r0 + 240].copy_(torch.from_numpy(part[k]))
    out32 = ctx.forward_lighting(gb_full, pf, pv, out_fmt=abi.FMT_RGBA32F)
    ref = O.forward_lighting(gb_crop, pf, pv, abi.FMT_RGBA32F)
    assert_bits(out32[rows[0]:rows[1]], ref, "4K crop")
    out16 = ctx.forward_lighting(gb_full, pf, pv, out_fmt=abi.FMT_RGBA16F)
    assert torch.equal(out16.view(torch.int16), out32.to(torch.float16).view(torch.int16))
    pfA, _ = synth.per_frame(points=[pts[i] for i in range(32)], ambient=0.055)
    pfB, _ = synth.per_frame(points=[pts[i] for i in range(32, 64)])
    a = ctx.forward_lighting(gb_full, pfA, pv, out_fmt=abi.FMT_RGBA32F)
    b = ctx.forward_lighting(gb_full, pfB, pv, out_fmt=abi.FMT_RGBA32F)
    base = ctx.forward_lighting(gb_full, synth.per_frame()[0], pv, out_fmt=abi.FMT_RGBA32F)
    s = (a[..., :3].double() + b[..., :3].double() - base[..., :3].double())
    rel = ((s - out32[..., :3].double()).abs() / out32[..., :3].double().abs().clamp_min(1e-6))
    # pow(1 - dot(H,V), 5.0) is the product x*((x*x)*(x*x)) (contract v4): a dot product that rounds a hair above 1 gives
    # a tiny negative product, not the NaN that exp2(5*log2(x)) — contracts v1-v3, and the engine's own compile — produces (~2e-5 of the pixels)
    assert torch.isfinite(out32).all()
    assert rel.max().item() < 2e-5, rel.max().item()


def test_forward_cfg3_full_frame_with_ibl(ctx):
    """The bench's own dispatch — 3840x2160, 64 point lights + the full-size cfg4 IBL, k_forward_lighting<env,nocasters,RGBA16F> — against
    the oracle on five 40-row bands spread over the frame (768 000 pixels), bit for bit, plus the 256-light cfg5 shape on one band."""
    from tests import ref_cases
    W, H = 3840, 2160
    g = ref_cases.cfg4_env()
    keep = []
    env_g = ref_cases.dev_env(g, keep)
    env_o = ref_cases.host_env(g)
    pf, _ = synth.per_frame(points=synth.point_lights(64, seed=0x6400), hdri_offset=0.3)
    pv = synth.per_view(W, H, max_env_lod=g["spec_mips"])
    gb_full = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(4)]
    for r0 in range(0, H, 240):
        part = synth.gbuffer_rows(W, H, r0, r0 + 240, seed=0x6400)
        for k in range(4):
            gb_full[k][r0:r0 + 240].copy_(torch.from_numpy(part[k]))
    out = ctx.forward_lighting(gb_full, pf, pv, out_fmt=abi.FMT_RGBA16F, env=env_g)
    for r0 in (0, 517, 1060, 1603, 2120):
        gb = synth.gbuffer_rows(W, H, r0, r0 + 40, seed=0x6400)
        assert_bits(out[r0:r0 + 40], O.forward_lighting(gb, pf, pv, abi.FMT_RGBA16F, env=env_o), f"cfg3 rows {r0}..{r0 + 40}")
    del gb_full, out
    W5, H5 = 7680, 4320
    pf5, extra = synth.per_frame(points=synth.point_lights(256, seed=0x2560))
    pv5 = synth.per_view(W5, H5)
    gb = synth.gbuffer_rows(W5, H5, 3000, 3024, seed=0x2560)
    got = ctx.forward_lighting([dev(x) for x in gb], pf5, pv5, out_fmt=abi.FMT_RGBA16F, extra_point=extra)
    assert_bits(got, O.forward_lighting(gb, pf5, pv5, abi.FMT_RGBA16F, extra_point=extra), "cfg5 band, 256 lights")


# ---------------------------------------------------------------------------------------------------
# post chain
# ---------------------------------------------------------------------------------------------------
POST_FORMS = [None, ("chain", 0), ("chain", 1), ("chain", 5), ("chain", 40), ("two", 0)]      # (post_form, post_strips): default / the one-kernel chain at any size, with
                                                                                               # one / few / more row strips than fit / the two-kernel path


@pytest.mark.parametrize("form", POST_FORMS)
@pytest.mark.parametrize("shape", [(2160, 3840), (97, 301), (64, 64), (33, 1000), (540, 257), (1440, 2560), (32, 64), (31, 700), (1, 1), (53, 129), (300, 128)])
def test_post_process_one_call_equals_three_dispatches(ctx, shape, form, set_opt):
    """vqhip_post_process against the oracle's three passes, bit for bit: 4K, sizes that are no multiple of the tiles, small images, negative and NaN inputs.
    Default = the one-kernel chain (k_post_chain: X, Y and the tonemapper; frames of >= 2^20 pixels) or blur X into the context's scratch, then blur Y + tonemap
    in one kernel; both forms at every size through the option post_form, the chain with several strip heights (post_strips)."""
    h, w = shape
    if form is not None:
        set_opt("post_form", form[0])
        if form[1]:
            set_opt("post_strips", form[1])
    if form is not None and form[1] and h * w > 3000 * 2000:
        pytest.skip("strip sweeps on the smaller frames")
    img = synth.hdr_image(w, h, seed=h * 7 + w).astype(np.float16)
    img[h // 3, w // 2, 0] = np.float16(-3.5)          # negative / NaN colours
    img[h // 2, w // 3, 1] = np.float16(np.nan)
    img[0, 0, 2] = np.float16(-0.0)
    img[h - 1, w - 1, 0] = np.float16(6.0e4)
    with np.errstate(all="ignore"):
        want = O.tonemap(O.gaussian_blur(img, abi.FMT_RGBA16F), abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM)
    got = ctx.post_process(dev(img), abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM)
    assert_bits(got, want, f"post_process {shape} {form}")
    if h <= 100:
        for p in (abi.TonemapperParams(0, abi.DISPLAY_CURVE_SRGB, 200.0, 0), abi.TonemapperParams(1, abi.DISPLAY_CURVE_ST2084, 200.0, 1),
                  abi.TonemapperParams(0, abi.DISPLAY_CURVE_ST2084, 200.0, 1), abi.TonemapperParams(0, abi.DISPLAY_CURVE_LINEAR, 200.0, 1)):
            with np.errstate(all="ignore"):
                want = O.tonemap(O.gaussian_blur(img, abi.FMT_RGBA16F), abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM, params=p)
            assert_bits(ctx.post_process(dev(img), abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM, params=p), want, f"post_process {shape} curve {p.OutputDisplayCurveEnum}")
        with np.errstate(all="ignore"):
            want16 = O.tonemap(O.gaussian_blur(img, abi.FMT_RGBA16F), abi.FMT_RGBA16F, abi.FMT_RGBA16F)
        assert_bits(ctx.post_process(dev(img), abi.FMT_RGBA16F, abi.FMT_RGBA16F), want16, "post_process HDR target (three dispatches)")
        assert_bits(ctx.post_process(dev(img), abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM, blur=False), O.tonemap(img, abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM), "no blur")


@pytest.mark.parametrize("form", [None, ("chain", 0), ("chain", 3), ("two", 0)])
@pytest.mark.parametrize("geom", [(192, 270, 10), (700, 135, 10), (130, 27, 12), (3840, 270, 10), (257, 64, 16)])
def test_post_process_tile_with_scene_colour_halos(ctx, geom, form, set_opt):
    """vqhip_post_process_tile: a row tile of a frame plus the neighbouring tiles' SCENE-COLOUR rows == the same rows of the untiled chain (oracle), bit for bit —
    the one-kernel chain filters the halo rows in X itself, the two-kernel path filters them into the context's scratch. Each halo is the LAST `hr` rows of its own
    exact-size allocation (no read may leave it); tiles at the top / bottom border of the frame (one halo NULL) and RGBA32F / HDR targets (two-kernel path) included."""
    W, rows, hr = geom
    if form is not None:
        set_opt("post_form", form[0])
        if form[1]:
            set_opt("post_strips", form[1])
    F16, F32, R8 = abi.FMT_RGBA16F, abi.FMT_RGBA32F, abi.FMT_RGBA8_UNORM
    img = synth.hdr_image(W, rows + 2 * hr, seed=0x7113 + W).astype(np.float16)
    img[hr + rows // 2, W // 2, 1] = np.float16(-2.0)
    with np.errstate(all="ignore"):
        full = O.tonemap(O.gaussian_blur(img, F16), F16, R8)
    top, mid, bottom = dev(img[:hr].copy()), dev(img[hr:hr + rows].copy()), dev(img[hr + rows:].copy())
    assert_bits(ctx.post_process_tile(mid, F16, R8, halo_top=top, halo_bottom=bottom), full[hr:hr + rows], f"interior tile {geom} {form}")
    # top tile of the frame: rows [0, rows) of img[hr:], the frame border above (clamp), the next tile's rows below
    with np.errstate(all="ignore"):
        full_t = O.tonemap(O.gaussian_blur(img[hr:], F16), F16, R8)
        full_b = O.tonemap(O.gaussian_blur(img[:hr + rows], F16), F16, R8)
    assert_bits(ctx.post_process_tile(mid, F16, R8, halo_bottom=bottom), full_t[:rows], f"top tile {geom} {form}")
    assert_bits(ctx.post_process_tile(mid, F16, R8, halo_top=top), full_b[hr:], f"bottom tile {geom} {form}")
    if W <= 300:
        with np.errstate(all="ignore"):
            hdr = O.tonemap(O.gaussian_blur(img, F16), F16, F16)
            f32 = O.tonemap(O.gaussian_blur(img.astype(np.float32), F32), F32, R8)
        assert_bits(ctx.post_process_tile(mid, F16, F16, halo_top=top, halo_bottom=bottom), hdr[hr:hr + rows], "HDR target")
        assert_bits(ctx.post_process_tile(dev(img[hr:hr + rows].astype(np.float32)), F32, R8, halo_top=dev(img[:hr].astype(np.float32)),
                                          halo_bottom=dev(img[hr + rows:].astype(np.float32))), f32[hr:hr + rows], "RGBA32F scene colour")


@pytest.mark.parametrize("rows", [270, 135, 27])
def test_blur_y_halo_at_the_end_of_an_allocation(ctx, rows):
    """Tile heights that are no multiple of the 16-row groups of the Y kernels (270 = 2160/8, 135): the register window of the last
    group extends past the 10 halo rows the filter needs. Those reads must stay inside the caller's 10-row halo buffer: here each
    halo is the LAST 10 rows of its own exact-size allocation and the result must still equal the untiled blur (all three Y kernels)."""
    W = 192
    img = synth.hdr_image(W, rows + 20, seed=0x4A10).astype(np.float16)
    full = O.blur_pass(img, abi.FMT_RGBA16F, 1)
    top = dev(img[:10].copy())
    mid = dev(img[10:10 + rows].copy())
    bottom = dev(img[10 + rows:].copy())
    got = ctx.gaussian_blur_y(mid, abi.FMT_RGBA16F, halo_top=top, halo_bottom=bottom)
    assert_bits(got, full[10:10 + rows], f"blur Y {rows}-row tile with exact-size halos")
    sdr = ctx.gaussian_blur_y_tonemap(mid, abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM, halo_top=top, halo_bottom=bottom)
    assert_bits(sdr, O.tonemap(full[10:10 + rows], abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM), "fused blur Y + tonemap (table kernel or LDS tile)")
    hdr = ctx.gaussian_blur_y_tonemap(mid, abi.FMT_RGBA16F, abi.FMT_RGBA16F, halo_top=top, halo_bottom=bottom)
    assert_bits(hdr, O.tonemap(full[10:10 + rows], abi.FMT_RGBA16F, abi.FMT_RGBA16F), "fused blur Y + tonemap, LDS-tile kernel")


@pytest.mark.parametrize("fmt", [abi.FMT_RGBA32F, abi.FMT_RGBA16F])
@pytest.mark.parametrize("shape", [(97, 301), (1, 5), (64, 64), (40, 1), (7, 2051)])
def test_blur(ctx, fmt, shape):
    h, w = shape
    img = synth.hdr_image(w, h).astype(O._NP[fmt][0])
    x_o = O.blur_pass(img, fmt, 0)
    x_g = ctx.gaussian_blur_x(dev(img), fmt)
    assert_bits(x_g, x_o, f"blur X {shape} fmt={fmt}")
    y_o = O.blur_pass(x_o, fmt, 1)
    y_g = ctx.gaussian_blur_y(x_g, fmt)
    assert_bits(y_g, y_o, f"blur Y {shape} fmt={fmt}")
    assert_bits(ctx.gaussian_blur(dev(img), fmt), y_o, f"blur XY {shape} fmt={fmt}")
    assert (y_o[..., 3] == 1).all()


def test_blur_y_row_tiles_with_halos(ctx):
    """Row-tiled Y pass (multi-GPU mode): tiles blurred with neighbour halos == the full-frame blur."""
    fmt = abi.FMT_RGBA16F
    h, w, tiles = 96, 130, 3
    x = O.blur_pass(synth.hdr_image(w, h).astype(np.float16), fmt, 0)
    full = O.blur_pass(x, fmt, 1)
    th = h // tiles
    xg = dev(x)
    for t in range(tiles):
        top = xg[t * th - 10:t * th].contiguous() if t > 0 else None
        bot = xg[(t + 1) * th:(t + 1) * th + 10].contiguous() if t < tiles - 1 else None
        got = ctx.gaussian_blur_y(xg[t * th:(t + 1) * th].contiguous(), fmt, halo_top=top, halo_bottom=bot)
        assert_bits(got, full[t * th:(t + 1) * th], f"tile {t}")
        ref_t = O.blur_pass(x[t * th:(t + 1) * th], fmt, 1, halo_top=x[t * th - 10:t * th] if t > 0 else None,
                            halo_bottom=x[(t + 1) * th:(t + 1) * th + 10] if t < tiles - 1 else None)
        n, _ = O.bits_equal(ref_t, full[t * th:(t + 1) * th])
        assert n == 0


@pytest.mark.parametrize("in_fmt", [abi.FMT_RGBA32F, abi.FMT_RGBA16F])
@pytest.mark.parametrize("out_fmt", [abi.FMT_RGBA8_UNORM, abi.FMT_RGBA16F, abi.FMT_RGBA32F])
@pytest.mark.parametrize("curve", [abi.DISPLAY_CURVE_SRGB, abi.DISPLAY_CURVE_ST2084, abi.DISPLAY_CURVE_LINEAR, 7])
def test_tonemap(ctx, in_fmt, out_fmt, curve):
    img = synth.hdr_image(257, 33, scale=50.0).astype(O._NP[in_fmt][0])
    img[0, :8, 0] = (0.0, 1e-6, 0.0031308, 0.0031309, 1.0, 65504.0, 0.18, 0.5)     # curve knee + extremes
    for gamma, cs in ((1, abi.COLOR_SPACE_REC_709), (0, abi.COLOR_SPACE_REC_2020)):
        p = abi.TonemapperParams(cs, curve, 200.0, gamma)
        ref = O.tonemap(img, in_fmt, out_fmt, p)
        got = ctx.tonemap(dev(img), in_fmt, out_fmt, p)
        assert_bits(got, ref, f"tonemap in={in_fmt} out={out_fmt} curve={curve} gamma={gamma}")


def test_post_chain_full_size_properties(ctx):
    """cfg3-sized post chain (3840x2160 RGBA16F): blur of a constant image is that constant times the summed weights;
    a 32-row band of blur->tonemap matches the oracle bit-exactly; tonemap output is monotone in input."""
    W, H = 3840, 2160
    img = torch.from_numpy(synth.hdr_image(W, H).astype(np.float16)).cuda()
    blurred = ctx.gaussian_blur(img, abi.FMT_RGBA16F)
    sdr = ctx.tonemap(blurred, abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM)
    band = (1200, 1232)
    src = img[band[0] - 10:band[1] + 10].cpu().numpy()
    x_o = O.blur_pass(src, abi.FMT_RGBA16F, 0)
    y_o = O.blur_pass(x_o, abi.FMT_RGBA16F, 1)[10:-10]
    assert_bits(blurred[band[0]:band[1]], y_o, "4K blur band")
    assert_bits(sdr[band[0]:band[1]], O.tonemap(y_o, abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM), "4K tonemap band")
    const = torch.full((64, 512, 4), 0.75, dtype=torch.float16, device="cuda")
    cb = ctx.gaussian_blur(const, abi.FMT_RGBA16F).float()
    assert (cb[..., :3] - 0.75).abs().max().item() < 2e-3 and (cb[..., 3] == 1).all()
    ramp = torch.linspace(0, 8, 4096, device="cuda").half().reshape(1, 4096, 1).repeat(1, 1, 4).contiguous()
    tm = ctx.tonemap(ramp, abi.FMT_RGBA16F, abi.FMT_RGBA32F)[0, :, 0]
    assert (tm[1:] >= tm[:-1]).all()


# ---------------------------------------------------------------------------------------------------
# C-ABI error behaviour (the reference asserts/logs; the ABI returns negative codes and never falls back)
# ---------------------------------------------------------------------------------------------------
def test_abi_errors(ctx):
    lib = ctx.lib
    g = synth.gbuffer(8, 2)
    pf, _ = synth.per_frame()
    pv = synth.per_view(8, 2)
    t = [dev(x) for x in g]
    with pytest.raises(capi.VQHipError) as e:
        pf.Lights.numPointLights = 101
        ctx.forward_lighting(t, pf, pv)
    assert e.value.code == abi.VQHIP_ERR_INVALID_ARG and "light count" in str(e.value)
    pf.Lights.numPointLights = 0
    pf.Lights.numSpotCasters = 1
    with pytest.raises(capi.VQHipError):
        ctx.forward_lighting(t, pf, pv)                       # casters without shadow maps
    pf.Lights.numSpotCasters = 0
    out8 = torch.empty((2, 8, 4), dtype=torch.uint8, device="cuda")
    gb = abi.GBuffer(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), 8, 2, 8)
    rc = lib.vqhip_forward_lighting(ctx._h, None, C.byref(gb), C.byref(pf), C.byref(pv), None, 0, None, None, out8.data_ptr(), 8, abi.FMT_RGBA8_UNORM)
    assert rc == abi.VQHIP_ERR_UNSUPPORTED
    rc = lib.vqhip_tonemap(ctx._h, None, None, out8.data_ptr(), 8, 2, C.byref(abi.TonemapperParams.default()), abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM)
    assert rc == abi.VQHIP_ERR_INVALID_ARG
    assert lib.vqhip_forward_lighting(None, None, None, None, None, None, 0, None, None, None, 0, 0) == abi.VQHIP_ERR_INVALID_ARG


# ---------------------------------------------------------------------------------------------------
# committed golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py from the oracle)
# ---------------------------------------------------------------------------------------------------
def test_hip_path_reproduces_golden_fixtures(ctx):
    import os
    from tests.golden import make_golden as G
    gdir = os.path.dirname(os.path.abspath(G.__file__))
    eq = synth.equirect(64, 32)
    chain, n = ctx.mip_chain(dev(eq))
    pre = ctx.envmap_prefilter(chain, 64, 32, n, 8, 0.1, 16)
    lut = ctx.brdf_lut(32, 64, abi.FMT_RG16F)
    fx = np.load(os.path.join(gdir, "ibl_small.npz"))
    assert_bits(chain[64 * 32:], fx["mip_tail"], "golden mip_tail")
    for k in ("diffuse_unblurred", "diffuse_blurred", "specular"):
        assert_bits(pre[k], fx[k], f"golden {k}")
    assert_bits(lut, fx["lut"], "golden lut")
    assert_bits(ctx.conv_diffuse(chain, 64, 32, n, 4, 0.1, abi.CONV_SEQUENTIAL, abi.FMT_RGBA16F), fx["diffuse_sequential_4"], "golden sequential diffuse")
    W, H, gb, pf, extra = G.shade_inputs()
    fx = np.load(os.path.join(gdir, "shade_small.npz"))
    g = [dev(x) for x in gb]
    assert_bits(ctx.forward_lighting(g, pf, synth.per_view(W, H), out_fmt=abi.FMT_RGBA32F), fx["noenv_rgba32f"], "golden shade noenv")
    env = capi.make_envmap(pre["diffuse_blurred"], pre["specular"], 16, pre["spec_mips"], lut)
    assert_bits(ctx.forward_lighting(g, pf, synth.per_view(W, H, max_env_lod=pre["spec_mips"]), out_fmt=abi.FMT_RGBA16F, env=env), fx["env_rgba16f"], "golden shade env")
    fx = np.load(os.path.join(gdir, "post_small.npz"))
    bl = ctx.gaussian_blur(dev(G.post_inputs()), abi.FMT_RGBA16F)
    assert_bits(bl, fx["blur_rgba16f"], "golden blur")
    assert_bits(ctx.tonemap(bl, abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM), fx["sdr_rgba8"], "golden sdr")
    assert_bits(ctx.tonemap(bl, abi.FMT_RGBA16F, abi.FMT_RGBA16F, abi.TonemapperParams(0, abi.DISPLAY_CURVE_ST2084, 200.0, 1)), fx["pq_rgba16f"], "golden pq")


@pytest.mark.parametrize("huge_ranges", [True, False])
def test_forward_fuzz_adversarial_inputs(ctx, env_small, huge_ranges):
    """Fuzz: G-buffer fields drawn from random BIT PATTERNS (all exponents, denormals, infs, NaNs) mixed with normal
    values, lights at degenerate distances (0, 1e-25, 1e25). Exercises the slow (IEEE) fall-backs of the proven-range
    fast reciprocals in add_point_light and every NaN/inf propagation path; must still match the oracle bit-for-bit.
    huge_ranges: ranges beyond 2^30 switch the whole light set to the IEEE loop (FrameConstants::pointFastOK); without them the
    fast loop runs and single pixels are redone (validity minimum below 2^-40, roughness outside [0.04, 1])."""
    rng = np.random.default_rng(77)
    W, H = 1024, 12
    gb = synth.gbuffer(W, H, seed=0xF022)
    for k in range(4):
        bits = rng.integers(0, 2 ** 32, gb[k].shape, dtype=np.uint32).view(np.float32)
        sel = rng.random(gb[k].shape) < 0.12
        gb[k] = np.where(sel, bits, gb[k]).astype(np.float32)
    gb[1][:, ::7, 3] = rng.choice(np.array([0.0, 1.0, 1.0000001, -1e-9, 2.0, 1e-20], np.float32), gb[1][:, ::7, 3].shape)   # roughness edge values
    pts = synth.point_lights(10, seed=0xF022)
    pts[1].position.set(gb[0][3, 100, :3]); pts[1].range = 1e9                     # D == 0 for one pixel
    pts[2].position.set(gb[0][5, 200, :3] + np.float32(1e-25)); pts[2].range = 1e9 # D^2 below 2^-60
    big = 3.0e38 if huge_ranges else 1.0e9
    pts[3].position.set((1e25, 1e25, -1e25)); pts[3].range = big                   # D^2 overflows
    pts[4].position.set((1e12, 0, 0)); pts[4].range = big                          # D^2 = 1e24 > 2^60
    pts[5].brightness = float("inf")
    pts[6].color.set((float("nan"), 1.0, -1.0))
    pts[7].range = float("nan")
    pf, _ = synth.per_frame(points=pts, spots=synth.spot_lights(2, seed=3), directional=synth.directional_light(), hdri_offset=0.3)
    env_o, env_g = _envs(env_small)
    pv = synth.per_view(W, H, max_env_lod=env_small["pre_o"]["spec_mips"])
    for env_pair in ((None, None), (env_o, env_g)):
        with np.errstate(all="ignore"):
            ref = O.forward_lighting(gb, pf, pv, abi.FMT_RGBA32F, env=env_pair[0])
        got = ctx.forward_lighting([dev(g) for g in gb], pf, pv, out_fmt=abi.FMT_RGBA32F, env=env_pair[1])
        assert_bits(got, ref, "fuzz forward lighting")
    assert np.isnan(ref).any() and np.isfinite(ref).any()


@pytest.mark.parametrize("fmt", [abi.FMT_RGBA32F, abi.FMT_RGBA16F])
@pytest.mark.parametrize("out_fmt", [abi.FMT_RGBA8_UNORM, abi.FMT_RGBA16F, abi.FMT_RGBA32F])
def test_fused_blur_y_tonemap_equals_two_dispatches(ctx, fmt, out_fmt):
    """vqhip_gaussian_blur_y_tonemap == vqhip_gaussian_blur_y then vqhip_tonemap == oracle, incl. row tiles with halos."""
    h, w = 75, 203
    x = O.blur_pass(synth.hdr_image(w, h, scale=30.0).astype(O._NP[fmt][0]), fmt, 0)
    for p in (abi.TonemapperParams.default(), abi.TonemapperParams(abi.COLOR_SPACE_REC_709, abi.DISPLAY_CURVE_ST2084, 300.0, 1)):
        ref = O.tonemap(O.blur_pass(x, fmt, 1), fmt, out_fmt, p)
        xg = dev(x)
        assert_bits(ctx.gaussian_blur_y_tonemap(xg, fmt, out_fmt, params=p), ref, "fused full image")
        assert_bits(ctx.tonemap(ctx.gaussian_blur_y(xg, fmt), fmt, out_fmt, params=p), ref, "two dispatches")
        t0, t1 = 25, 50
        got = ctx.gaussian_blur_y_tonemap(xg[t0:t1].contiguous(), fmt, out_fmt, params=p, halo_top=xg[t0 - 10:t0].contiguous(), halo_bottom=xg[t1:t1 + 10].contiguous())
        assert_bits(got, ref[t0:t1], "fused tile with halos")


@pytest.mark.parametrize("out_fmt", [abi.FMT_RGBA8_UNORM, abi.FMT_RGBA16F])
def test_tonemap_table_path(ctx, out_fmt):
    """Images >= 65536 px in RGBA16F take the 65536-entry table kernel (k_tonemap_lut) whenever the curve does not mix
    channels; it must equal the oracle for EVERY half bit pattern (all 65536 appear in the image) and every parameter set,
    and the channel-mixing case (ST2084 on Rec.709 content) must still take the direct kernel."""
    allh = np.arange(65536, dtype=np.uint16).view(np.float16)
    img = np.empty((300, 256, 4), np.float16)
    rng = np.random.default_rng(8)
    for c in range(4):
        img[..., c] = np.concatenate([rng.permutation(allh), rng.choice(allh, 300 * 256 - 65536)]).reshape(300, 256)
    cases = [abi.TonemapperParams(0, abi.DISPLAY_CURVE_SRGB, 200.0, 1), abi.TonemapperParams(1, abi.DISPLAY_CURVE_SRGB, 200.0, 0),
             abi.TonemapperParams(1, abi.DISPLAY_CURVE_ST2084, 450.0, 1), abi.TonemapperParams(0, abi.DISPLAY_CURVE_ST2084, 200.0, 1),
             abi.TonemapperParams(0, abi.DISPLAY_CURVE_LINEAR, 200.0, 1), abi.TonemapperParams(0, 5, 200.0, 1)]
    g = dev(img)
    for p in cases:
        with np.errstate(all="ignore"):
            ref = O.tonemap(img, abi.FMT_RGBA16F, out_fmt, p)
        assert_bits(ctx.tonemap(g, abi.FMT_RGBA16F, out_fmt, params=p), ref, f"table tonemap curve={p.OutputDisplayCurveEnum} cs={p.ContentColorSpaceEnum}")
