"""GPU parity of FSR 1.0: vqhip_fsr_easu / vqhip_fsr_rcas through the C ABI against oracle/vqo_fsr.cpp —
identical bits in every storage format, plus the post-chain tail tonemap -> EASU -> RCAS at the reference's formats."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import oracle_lib as O
from tests.test_fsr import _img
from vqengine_amd import abi, capi, synth

pytestmark = pytest.mark.gpu

_NP = {abi.FMT_RGBA32F: np.float32, abi.FMT_RGBA16F: np.float16, abi.FMT_RGBA8_UNORM: np.uint8}


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _as(img, fmt):
    if fmt == abi.FMT_RGBA8_UNORM:
        return (np.clip(img, 0, 1) * 255 + 0.5).astype(np.uint8)
    return img.astype(_NP[fmt])


@pytest.mark.parametrize("in_fmt", [abi.FMT_RGBA32F, abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM])
@pytest.mark.parametrize("out_fmt", [abi.FMT_RGBA32F, abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM])
@pytest.mark.parametrize("shape", [((160, 90), (240, 135)), ((97, 33), (200, 61)), ((8, 8), (8, 8)), ((1, 1), (3, 2))])
def test_easu_matches_oracle(ctx, in_fmt, out_fmt, shape):
    (w, h), (ow, oh) = shape
    img = _as(_img(w, h, seed=w + h), in_fmt)
    ref = O.fsr_easu(img, in_fmt, ow, oh, out_fmt)
    got = ctx.fsr_easu(dev(img), in_fmt, ow, oh, out_fmt)
    n, idx = O.bits_equal(got.cpu().numpy(), ref)
    assert n == 0, (n, idx)


@pytest.mark.parametrize("in_fmt", [abi.FMT_RGBA32F, abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM])
@pytest.mark.parametrize("out_fmt", [abi.FMT_RGBA32F, abi.FMT_RGBA8_UNORM])
@pytest.mark.parametrize("stops", [0.0, 0.2, 2.0])
def test_rcas_matches_oracle(ctx, in_fmt, out_fmt, stops):
    img = _as(_img(131, 47, seed=7), in_fmt)
    ref = O.fsr_rcas(img, in_fmt, out_fmt, con=O.fsr_rcas_con(stops))
    got = ctx.fsr_rcas(dev(img), in_fmt, out_fmt, con=capi.fsr_rcas_con(stops))
    n, idx = O.bits_equal(got.cpu().numpy(), ref)
    assert n == 0, (stops, n, idx)


def test_post_chain_tail_reference_formats(ctx):
    """RGBA16F scene colour -> tonemap (RGBA8) -> EASU 1.5x -> RCAS, every stage in the reference's SDR format."""
    w, h, ow, oh = 320, 180, 480, 270
    scene = synth.hdr_image(w, h, seed=0xF5).astype(np.float16)
    sdr_o = O.tonemap(scene, abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM)
    up_o = O.fsr_easu(sdr_o, abi.FMT_RGBA8_UNORM, ow, oh)
    fin_o = O.fsr_rcas(up_o, abi.FMT_RGBA8_UNORM)
    sdr_g = ctx.tonemap(dev(scene), abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM)
    up_g = ctx.fsr_easu(sdr_g, abi.FMT_RGBA8_UNORM, ow, oh)
    fin_g = ctx.fsr_rcas(up_g, abi.FMT_RGBA8_UNORM)
    assert np.array_equal(up_g.cpu().numpy(), up_o) and np.array_equal(fin_g.cpu().numpy(), fin_o)


def test_fsr_full_size_properties(ctx):
    """2560x1440 -> 3840x2160 (the FSR 'Quality' ratio at 4K): constant frame stays constant through EASU, outputs stay inside
    the input range, 8 output rows checked bit for bit against the oracle run on the matching input window."""
    w, h, ow, oh = 2560, 1440, 3840, 2160
    const = torch.full((h, w, 4), 0.5, dtype=torch.float16, device="cuda")
    out = ctx.fsr_easu(const, abi.FMT_RGBA16F, ow, oh).cpu().numpy()
    assert np.all(out[..., :3] == np.float16(0.5)) and np.all(out[..., 3] == np.float16(1.0))
    img = _as(_img(w, h, seed=21), abi.FMT_RGBA8_UNORM)
    up = ctx.fsr_easu(dev(img), abi.FMT_RGBA8_UNORM, ow, oh).cpu().numpy()
    ref = O.fsr_easu(img, abi.FMT_RGBA8_UNORM, ow, oh)                       # ~1 s on the host cores
    assert np.array_equal(up, ref)
    fin = ctx.fsr_rcas(dev(up), abi.FMT_RGBA8_UNORM).cpu().numpy()
    assert np.array_equal(fin, O.fsr_rcas(ref, abi.FMT_RGBA8_UNORM))


@pytest.mark.parametrize("in_fmt", [abi.FMT_RGBA32F, abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM])
@pytest.mark.parametrize("out_fmt", [abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM])
def test_visualize_matches_oracle(ctx, in_fmt, out_fmt):
    r = np.random.default_rng(4)
    img = _as(r.random((37, 129, 4), dtype=np.float32), in_fmt)
    for mode, unpack, strength in ((1, 0, 1.0), (2, 0, 1.0), (2, 1, 1.0), (3, 0, 1.0), (4, 0, 1.0), (5, 0, 1.0), (6, 0, 1.0), (7, 0, 1.0), (8, 0, 7.25), (0, 0, 1.0), (42, 0, 1.0)):
        p = abi.VizParams(mode, unpack, strength)
        n, idx = O.bits_equal(ctx.visualize(dev(img), in_fmt, p, out_fmt).cpu().numpy(), O.visualize(img, in_fmt, p, out_fmt))
        assert n == 0, (mode, n, idx)


def test_visualize_reads_the_targets_the_draw_modes_bind(ctx):
    """SceneRendering.cpp:2555-2566: NORMALS shows Tex_SceneNormals (R10G10B10A2), MOTION_VECTORS Tex_SceneMotionVectors (RG16F), ALBEDO / METALLIC
    Tex_SceneVisualization (RGBA16F) — here the product's own vqhip_scene_normals_from_materials / vqhip_forward_lighting_mrt outputs, visualised, == the oracle chain."""
    from tests.test_gpu_gbuffer import build_materials
    from vqengine_amd import synth
    W, H, NM = 200, 48, 4
    ip = synth.interpolants(W, H, NM)
    _, _, hmats, dmats, keep = build_materials(ctx, NM, max_dim=64)
    cur, prev = synth.clip_positions(W, H)
    pf, _ = synth.per_frame(points=synth.point_lights(3))
    pv = synth.per_view(W, H)
    ipd = [dev(p) for p in ip]
    nrm = ctx.scene_normals_from_materials(ipd, dmats)
    _, alb, mv = ctx.forward_lighting_from_materials_mrt(ipd, dmats, pf, pv, motion_fmt=abi.FMT_RG16F, sv_curr=dev(cur), sv_prev=dev(prev))
    nrm_o = O.scene_normals_from_materials(ip, hmats)
    gb_o = O.gbuffer_from_materials([p.copy() for p in ip], hmats, pf.fAmbientLightingFactor, None)
    alb_o, mv_o = O.psmain_extra_targets(gb_o, cur, prev)
    idx = ipd[2][..., 3].contiguous().view(torch.int32).cpu().numpy()
    holes = ~((idx >= 0) & (idx < NM))                       # no fragment there: the one-kernel PSMain leaves the targets at their clear value (0), like the rasteriser
    alb_o[holes] = 0
    mv_o[holes] = 0
    for src, src_o, fmt, modes in ((nrm, nrm_o, abi.FMT_R10G10B10A2_UNORM, ((2, 0, 1.0), (2, 1, 1.0))), (mv, mv_o, abi.FMT_RG16F, ((8, 0, 40.0),)),
                                   (alb, alb_o, abi.FMT_RGBA16F, ((6, 0, 1.0), (4, 0, 1.0)))):
        for mode, unpack, strength in modes:
            p = abi.VizParams(mode, unpack, strength)
            for out_fmt in (abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM):
                n, idx = O.bits_equal(ctx.visualize(src, fmt, p, out_fmt).cpu().numpy(), O.visualize(src_o, fmt, p, out_fmt))
                assert n == 0, (fmt, mode, out_fmt, n, idx)
    mv32 = O.psmain_extra_targets(gb_o, cur, prev, None, abi.FMT_RG32F)[1]
    p = abi.VizParams(8, 0, 40.0)
    n, idx = O.bits_equal(ctx.visualize(dev(mv32), abi.FMT_RG32F, p, abi.FMT_RGBA32F).cpu().numpy(), O.visualize(mv32, abi.FMT_RG32F, p, abi.FMT_RGBA32F))
    assert n == 0, (n, idx)
    out = O.visualize(mv_o, abi.FMT_RG16F, p, abi.FMT_RGBA32F)
    assert (out[..., 2] == 0.5).all() and (out[..., 3] == 1.0).all()                     # typed load of a two-channel target: b = 0, a = 1


@pytest.mark.parametrize("fmt", [abi.FMT_RGBA32F, abi.FMT_RGBA16F])
def test_apply_reflections(ctx, fmt):
    """ApplyReflections.hlsl:30-50: one IEEE add per colour channel in fp32, result rounded to the target format, alpha kept."""
    dt = _NP[fmt]
    scene = synth.hdr_image(157, 43, seed=3).astype(dt)
    refl = synth.hdr_image(157, 43, seed=4, scale=0.5).astype(dt)
    refl[5, 7] = (np.inf, -1.0, np.nan, 9.0)
    exp = scene.copy()
    with np.errstate(invalid="ignore", over="ignore"):
        exp[..., :3] = (scene[..., :3].astype(np.float32) + refl[..., :3].astype(np.float32)).astype(dt)
    got = ctx.apply_reflections(dev(refl), dev(scene), fmt).cpu().numpy()
    n, idx = O.bits_equal(got, exp)
    assert n == 0, (n, idx)


@pytest.mark.parametrize("fmt", [abi.FMT_RGBA32F, abi.FMT_RGBA16F])
def test_composite_reflections_with_bounding_volumes(ctx, fmt):
    """VQRenderer::CompositeReflections, the COMPOSITE_BOUNDING_VOLUMES permutation (ApplyReflections.hlsl:44-48): BV.rgb * BV.a + (scene + refl) * (1 - BV.a),
    alpha = BV.a — every operation rounded on its own; without the image it is vqhip_apply_reflections; aliasing inputs are refused."""
    dt = _NP[fmt]
    W, H = 1921, 23
    scene = synth.hdr_image(W, H, seed=3).astype(dt)
    refl = synth.hdr_image(W, H, seed=4, scale=0.5).astype(dt)
    bv = synth.hdr_image(W, H, seed=5, scale=0.25).astype(dt)
    r = np.random.default_rng(5)
    bv[..., 3] = r.random((H, W)).astype(dt)
    bv[::3, ::2, 3] = 0.0
    bv[1::3, ::2, 3] = 1.0
    bv[2, 5] = (np.inf, 1.0, np.nan, 0.5)
    refl[5, 7] = (np.inf, -1.0, np.nan, 9.0)
    n, idx = O.bits_equal(ctx.composite_reflections(dev(refl), dev(scene), fmt, dev(bv)).cpu().numpy(), O.composite_reflections(refl, scene, fmt, bv))
    assert n == 0, (n, idx)
    n, idx = O.bits_equal(ctx.composite_reflections(dev(refl), dev(scene), fmt).cpu().numpy(), ctx.apply_reflections(dev(refl), dev(scene), fmt).cpu().numpy())
    assert n == 0, (n, idx)
    s = dev(scene)
    assert ctx.lib.vqhip_composite_reflections(ctx._h, None, C.c_void_p(s.data_ptr()), None, C.c_void_p(s.data_ptr()), W, H, fmt) == abi.VQHIP_ERR_INVALID_ARG
    assert ctx.lib.vqhip_composite_reflections(ctx._h, None, C.c_void_p(dev(refl).data_ptr()), C.c_void_p(s.data_ptr()), C.c_void_p(s.data_ptr()), W, H, fmt) == abi.VQHIP_ERR_INVALID_ARG
    assert ctx.lib.vqhip_composite_reflections(ctx._h, None, C.c_void_p(dev(refl).data_ptr()), None, C.c_void_p(s.data_ptr()), W, H, abi.FMT_RGBA8_UNORM) == abi.VQHIP_ERR_UNSUPPORTED


def test_fsr_abi_errors(ctx):
    lib = ctx.lib
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    a = torch.zeros((8, 8, 4), dtype=torch.uint8, device="cuda")
    b = torch.zeros((16, 16, 4), dtype=torch.uint8, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    con = capi.fsr_easu_con(8, 8, 16, 16)
    assert lib.vqhip_fsr_easu(ctx._h, st, p(a), 8, 8, abi.FMT_RGBA8_UNORM, con, p(b), 16, 16, abi.FMT_RGBA8_UNORM) == 0
    assert lib.vqhip_fsr_easu(ctx._h, st, p(a), 8, 8, abi.FMT_RG16F, con, p(b), 16, 16, abi.FMT_RGBA8_UNORM) == abi.VQHIP_ERR_UNSUPPORTED
    assert lib.vqhip_fsr_easu(ctx._h, st, p(a), 8, 8, abi.FMT_RGBA8_UNORM, None, p(b), 16, 16, abi.FMT_RGBA8_UNORM) == abi.VQHIP_ERR_INVALID_ARG
    assert lib.vqhip_fsr_easu(ctx._h, st, p(a), 8, 8, abi.FMT_RGBA8_UNORM, con, p(a), 8, 8, abi.FMT_RGBA8_UNORM) == abi.VQHIP_ERR_INVALID_ARG
    rc = capi.fsr_rcas_con()
    assert lib.vqhip_fsr_rcas(ctx._h, st, p(b), p(b), 16, 16, rc, abi.FMT_RGBA8_UNORM, abi.FMT_RGBA8_UNORM) == abi.VQHIP_ERR_INVALID_ARG
    assert lib.vqhip_fsr_rcas(ctx._h, st, p(a), p(b), 0, 8, rc, abi.FMT_RGBA8_UNORM, abi.FMT_RGBA8_UNORM) == abi.VQHIP_ERR_INVALID_ARG
