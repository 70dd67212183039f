"""GPU parity of the round-6 forms of the spot-light and shadow-caster path (vq_shade.h: spot_geometry / spot_light with its wave-level skip, the separable 5 x 5 PCF,
the unrolled cube PCF), bit-exact against the CPU oracle through the C ABI — on the two workloads bench.py times (benchlib/casters.py: real view-projection matrices with a
perspective w, shadow maps of the engine's sizes) and on inputs chosen to reach every exit: lit and idle waves, accumulators holding zeros, degenerate cones and directions,
non-finite colours, a pixel AT the light, maps that are not powers of two, w = 0, both readings of dot / normalize and both Fresnel powers."""
import numpy as np
import pytest

from tests import oracle_lib as O
from tests.test_gpu_parity import assert_bits, dev
from vqengine_amd import abi, scene, synth

pytestmark = pytest.mark.gpu
F32, F16 = abi.FMT_RGBA32F, abi.FMT_RGBA16F


def _maps(dims, n_spot=5, n_point=5, seed=0x5AD0):
    m = scene.synthetic_shadow_maps(dims, n_spot=n_spot, n_point=n_point, seed=seed)
    return m, scene.shadow_maps_struct(m, lambda a: a.ctypes.data)


def _dev_maps(m, keep):
    t = [dev(m[k]) for k in ("dir", "spot", "point")]
    keep += t
    return abi.ShadowMaps(t[0].data_ptr(), m["dims"][0], t[1].data_ptr(), m["dims"][1], t[2].data_ptr(), m["dims"][2])


def _check(ctx, gb, pf, pv, m, what, fmts=(F32,)):
    keep = []
    sm_g, sm_o = _dev_maps(m, keep), scene.shadow_maps_struct(m, lambda a: a.ctypes.data)
    for fmt in fmts:
        with np.errstate(all="ignore"):
            ref = O.forward_lighting(gb, pf, pv, fmt, shadow=sm_o)
        got = ctx.forward_lighting([dev(g) for g in gb], pf, pv, out_fmt=fmt, shadow=sm_g)
        assert_bits(got, ref, f"{what} fmt={fmt}")
    return ref


@pytest.mark.parametrize("coherent", [False, True])
def test_cfg1_default_scene_bands(ctx, coherent):
    """BASELINE cfg1 as bench.py times it: the Default scene's casters with their own view-projection matrices (90-degree perspective frusta: w != 1), 2048^2 / 1024^2 maps."""
    W, H = 1280, 720
    pf, maps = scene.default_scene_frame()
    pv = synth.per_view(W, H)
    gen = synth.gbuffer_rows_coherent if coherent else synth.gbuffer_rows
    lit = 0.0
    for r0 in (120, 352, 600):
        gb = gen(W, H, r0, r0 + 12, seed=0xC0FFEE)
        ref = _check(ctx, gb, pf, pv, maps, f"cfg1 rows {r0} coherent={coherent}", (F16,))
        base = gb[2][..., :3] * gb[0][..., 3:4] + gb[3][..., :3] * gb[3][..., 3:4]
        lit = max(lit, float((np.abs(ref[..., :3].astype(np.float32) - base) > 1e-3).mean()))
    assert lit > 0.2, "the lights never reached the bands: nothing was exercised"


@pytest.mark.parametrize("coherent", [False, True])
def test_engine_max_bands(ctx, coherent):
    """100 point + 20 spot lights, 5 + 5 + 1 casters at the engine's map sizes (the `engine_max` object of the bench line), rows across the frame."""
    W, H = 3840, 2160
    pf, maps = scene.engine_max_frame()
    pv = synth.per_view(W, H)
    gen = synth.gbuffer_rows_coherent if coherent else synth.gbuffer_rows
    for r0 in (8, 1080, 2100):
        gb = gen(W, H, r0, r0 + 4, seed=0x6400)
        _check(ctx, gb, pf, pv, maps, f"engine_max rows {r0} coherent={coherent}", (F16,))


def _spot_fuzz_frame():
    """A coherent frame lit by spots only: most waves are idle for most spots (the wave-level skip), the cones' borders cut through waves."""
    W, H = 768, 24
    gb = synth.gbuffer(W, H, seed=0x5107, coherent=True)
    spots = synth.spot_lights(8, seed=0x5107)
    return W, H, gb, spots


@pytest.mark.parametrize("arith_dxc", [False, True])
@pytest.mark.parametrize("pow_exp2", [False, True])
def test_spot_lights_every_exit(ctx, arith_dxc, pow_exp2):
    W, H, gb, spots = _spot_fuzz_frame()
    rng = np.random.default_rng(0x5107)
    # accumulators with zeros: no ambient / emissive in some rows (+0), albedo -0 in others (-0 + ... ), so an idle spot must not flip a sign
    gb[0][0:4, :, 3] = 0.0; gb[3][0:4] = 0.0
    gb[2][2:4, :, :3] = -0.0
    gb[3][3, :, :3] = -0.0; gb[3][3, :, 3] = 1.0
    gb[2][5, ::7, 0] = np.inf; gb[2][5, 3::7, 2] = np.nan                # non-finite BRDF in some lanes of otherwise idle waves
    gb[1][6, ::11, :3] = np.nan; gb[1][7, ::13, :3] = 0.0                 # NaN / zero normals
    gb[1][8, ::5, 3] = rng.choice(np.array([0.0, 1.0, 1.0000001, -1e-9, 2.0, 0.03], np.float32), gb[1][8, ::5, 3].shape)
    gb[0][9, 100, :3] = (spots[0].position.x, spots[0].position.y, spots[0].position.z)          # a pixel AT the light: D = 0
    gb[0][9, 200, :3] = np.array((spots[1].position.x, spots[1].position.y, spots[1].position.z), np.float32) + np.float32(1e-25)
    gb[0][10, ::17, 1] = 1e25; gb[0][10, 5::17, 0] = np.nan; gb[0][10, 9::17, 2] = np.inf
    spots[2].spotDir.set((0.0, 0.0, 0.0))                                 # normalize -> NaN
    spots[3].innerConeAngle = spots[3].outerConeAngle                     # outer - inner = 0: the quotient's divisor is not normal
    spots[4].innerConeAngle = spots[4].outerConeAngle + 0.1               # inner > outer
    spots[5].brightness = float("inf")                                    # color * brightness not finite: never skipped
    spots[6].color.set((float("nan"), 1.0, -1.0))
    spots[7].spotDir.set((1e-30, -1e-30, 0.0))                            # a direction whose squared length is below the fast range
    ctx.set_arithmetic(arith_dxc); O.load().vqo_set_arithmetic(1 if arith_dxc else 0)
    ctx.set_fresnel_pow(pow_exp2); O.load().vqo_set_fresnel_pow(1 if pow_exp2 else 0)
    try:
        for n in (8, 2):                                                  # 2: only well-formed spots (the pure skip / lit forms)
            pf, _ = synth.per_frame(spots=(abi.SpotLight * n)(*[spots[i] for i in ((0, 1) if n == 2 else range(8))]), ambient=0.0)
            pv = synth.per_view(W, H)
            with np.errstate(all="ignore"):
                ref = O.forward_lighting(gb, pf, pv, F32)
            got = ctx.forward_lighting([dev(g) for g in gb], pf, pv, out_fmt=F32)
            assert_bits(got, ref, f"spot fuzz n={n} dxc={arith_dxc} exp2={pow_exp2}")
        assert np.isnan(ref).any() and np.isfinite(ref).any()
    finally:
        ctx.set_arithmetic(False); O.load().vqo_set_arithmetic(0)
        ctx.set_fresnel_pow(False); O.load().vqo_set_fresnel_pow(0)


@pytest.mark.parametrize("dims", [(64, 32, 16), (100, 48, 20), (1, 1, 1), (257, 1000, 129)])
def test_casters_odd_map_sizes_and_degenerate_views(ctx, dims):
    """Maps that are not powers of two take the integer-modulo wrap; 1 x 1 maps; a view matrix with w = 0 / negative w / NaN; a point caster AT a pixel (major axis 0:
    the cube fetch's reciprocal is redone with the checked operations); casters whose illumination is idle in whole waves."""
    W, H = 512, 12
    gb = synth.gbuffer(W, H, seed=0xCA57, coherent=True)
    pf, _ = scene.engine_max_frame(map_dims=(64, 32, 16))
    maps, _ = _maps(dims, seed=0xCA57)
    scene._set_shadow_dims(pf, dims)
    L = pf.Lights
    L.numPointLights = 3
    L.numSpotLights = 4
    gb[0][1, 50, :3] = (L.point_casters[0].position.x, L.point_casters[0].position.y, L.point_casters[0].position.z)
    gb[0][1, 60, :3] = np.array((L.point_casters[1].position.x, L.point_casters[1].position.y, L.point_casters[1].position.z), np.float32) + np.float32(1e-30)
    gb[0][2, ::9, 0] = np.nan; gb[0][2, 4::9, 1] = np.inf
    gb[0][3:5, :, 3] = 0.0; gb[3][3:5] = 0.0; gb[2][4, :, :3] = -0.0      # zeros in the accumulator
    L.point_casters[2].range = 30.0                                       # a range that cuts through the frame
    L.point_casters[3].range = float("nan")
    L.shadowViews[1].m[0][3] = 0.0; L.shadowViews[1].m[1][3] = 0.0; L.shadowViews[1].m[2][3] = 0.0; L.shadowViews[1].m[3][3] = 0.0      # w = 0 everywhere
    L.shadowViews[2].m[3][3] = -50.0                                      # negative / sign-changing w
    L.shadowViews[3].m[2][2] = float("nan")
    pv = synth.per_view(W, H)
    _check(ctx, gb, pf, pv, maps, f"casters dims={dims}", (F32, F16))


def test_casters_dxc_reading_and_exp2_pow(ctx):
    W, H = 640, 8
    pf, maps = scene.engine_max_frame(map_dims=(128, 64, 32))
    pv = synth.per_view(W, H)
    gb = synth.gbuffer(W, H, seed=0xD0C)
    for dxc, p5 in ((True, False), (False, True), (True, True)):
        ctx.set_arithmetic(dxc); O.load().vqo_set_arithmetic(1 if dxc else 0)
        ctx.set_fresnel_pow(p5); O.load().vqo_set_fresnel_pow(1 if p5 else 0)
        try:
            _check(ctx, gb, pf, pv, maps, f"casters dxc={dxc} exp2={p5}", (F16,))
        finally:
            ctx.set_arithmetic(False); O.load().vqo_set_arithmetic(0)
            ctx.set_fresnel_pow(False); O.load().vqo_set_fresnel_pow(0)


FUZZ_SEEDS_THAT_FAILED_ONCE = [1001926, 1003254, 1003275, 1003389, 1003901, 1004453, 1004987, 1005360]       # an albedo of 1e25: F0 and kA finite, their product not (px.skipOK)


@pytest.mark.parametrize("seed", FUZZ_SEEDS_THAT_FAILED_ONCE + [1000003, 1000040, 1000085])
def test_fuzz_cases_that_failed_once(ctx, seed):
    """tests/fuzz/fuzz_casters.py draws light counts, map sizes, matrices, biases, lattice positions and special values per case; the seeds that ever differed are replayed here
    (and three ordinary ones, among them a frame with casters and NO maps, which both sides refuse)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz"))
    import fuzz_casters
    bad, idx, what, got, ref = fuzz_casters.run_case(ctx, seed, dev)
    assert bad == 0, f"{what}: {bad} channels, first at {np.asarray(idx).tolist()}"


@pytest.mark.parametrize("arith_dxc", [False, True])
def test_huge_albedo_in_a_wave_that_skips_lights(ctx, arith_dxc):
    """Finite F0 and kA whose PRODUCT overflows (albedo 1e25, metalness between 0 and 1): the BRDF of such a lane is -inf, the reference's b * (cb * +0) is NaN for a light that
    faces away or whose cone misses the pixel — the lane must keep its wave out of the back-facing skip of the point-light loop and of the idle-wave exit of the spot lights."""
    W, H = 512, 6
    gb = [np.array(g, copy=True) for g in synth.gbuffer(W, H, seed=0xA1BE, coherent=True)]
    gb[1][..., :3] = np.float32((0.0, 1.0, 0.0))                          # one surface: every wave is eligible for the skip
    gb[1][..., 3] = 0.5
    for y, x, ch in ((0, 5, 0), (1, 70, 1), (2, 200, 2), (3, 300, 0), (4, 450, 1)):
        gb[2][y, x, ch] = 1e25
        gb[2][y, x, 3] = 0.4
    gb[2][5, 100, :3] = 3e19; gb[2][5, 100, 3] = 0.5                      # just below the overflow: finite b, the skip's result and the full evaluation agree
    points = synth.point_lights(24, seed=0xA1BE)
    for i in range(0, 24, 2):
        points[i].position.y = -30.0                                      # below the surface: NdotL = +0 in every lane
    pf, _ = synth.per_frame(points=points, spots=synth.spot_lights(8, seed=0xA1BE), ambient=0.055)
    pv = synth.per_view(W, H)
    ctx.set_arithmetic(arith_dxc); O.load().vqo_set_arithmetic(1 if arith_dxc else 0)
    try:
        with np.errstate(all="ignore"):
            ref = O.forward_lighting(gb, pf, pv, F32)
        got = ctx.forward_lighting([dev(g) for g in gb], pf, pv, out_fmt=F32)
        assert_bits(got, ref, f"huge albedo dxc={arith_dxc}")
        assert np.isnan(ref[0, 5, :3]).any() and np.isfinite(ref[0, 6, :3]).all()
    finally:
        ctx.set_arithmetic(False); O.load().vqo_set_arithmetic(0)
