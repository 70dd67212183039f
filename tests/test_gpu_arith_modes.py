"""-m gpu: the HIP product in the DXC reading (vqhip_set_arithmetic(VQHIP_ARITH_DXC), with either Fresnel power) bit for bit against the oracle in the same
mode — forward lighting with every light type + IBL + shadow casters, the G-buffer producer, PSMain as one kernel, SSR's environment fallback — and
within one RGBA16F ulp of the DXC build of the reference's own HLSL on the four BASELINE-shape bands (tests/golden/ref_outputs_dxc.npz)."""
import os

import numpy as np
import pytest

from tests import oracle_lib as O
from tests import ref_cases
from tests.test_arith_modes import FIX, MAX_FRACTION, TAGS, boundary_gbuffer, dxc_mode
from tests.test_ref_readings import distance
from vqengine_amd import abi, capi, synth

pytestmark = pytest.mark.gpu
dev = ref_cases._dev
F16 = abi.FMT_RGBA16F


def assert_bits(got, ref, what):
    n, idx = O.bits_equal(got.cpu().numpy() if hasattr(got, "cpu") else got, ref)
    assert n == 0, f"{what}: {n} mismatching elements, first {idx.tolist()}"


@pytest.mark.parametrize("fresnel", [True, False])
def test_forward_lighting_all_light_types(ctx, fresnel):
    W, H = 256, 24
    e = ref_cases.small_env()
    keep = []
    gb = synth.gbuffer(W, H, seed=0xD8C)
    gb[1][0, :4, 3] = [0.0, 0.01, 0.039, 1.0]                                       # polished pixels: the EPSILON-select form of the loop
    pf, extra = synth.per_frame(points=synth.point_lights(40, seed=0xD8C), spots=synth.spot_lights(3, seed=0xD8C), directional=synth.directional_light(), hdri_offset=0.3)
    pv = synth.per_view(W, H, max_env_lod=e["spec_mips"])
    with dxc_mode(ctx, fresnel):
        want = O.forward_lighting(gb, pf, pv, F16, extra_point=extra, env=ref_cases.host_env(e))
        got = ctx.forward_lighting([dev(g) for g in gb], pf, pv, out_fmt=F16, extra_point=extra, env=ref_cases.dev_env(e, keep))
        want32 = O.forward_lighting(gb, pf, pv, abi.FMT_RGBA32F, extra_point=extra)
        got32 = ctx.forward_lighting([dev(g) for g in gb], pf, pv, out_fmt=abi.FMT_RGBA32F, extra_point=extra)
    assert_bits(got, want, f"forward lighting, DXC reading, exp2/log2 Fresnel {fresnel}")
    assert_bits(got32, want32, "forward lighting without IBL, RGBA32F, DXC reading")
    lit = ctx.forward_lighting([dev(g) for g in gb], pf, pv, out_fmt=abi.FMT_RGBA32F, extra_point=extra)
    assert not np.array_equal(lit.cpu().numpy(), got32.cpu().numpy())              # the mode was switched back, and it is a different function


def test_shadow_casters_and_degenerate_geometry(ctx):
    """the Default-scene band (directional + spot casters, PCF) and lights exactly above a pixel on an axis (the IEEE redo of the loop) in the DXC reading"""
    build, _, _ = ref_cases.DXC_SCENES["cfg1_default_1280x16"]
    inp = build()
    keep = []
    with dxc_mode(ctx):
        gb = boundary_gbuffer(inp)
        want = O.forward_lighting(gb, inp["pf"], inp["pv"], F16, shadow=ref_cases.host_shadow_dims(inp["shadow"]))
        got = ctx.forward_lighting([dev(g) for g in gb], inp["pf"], inp["pv"], out_fmt=F16, shadow=ref_cases.dev_shadow_dims(inp["shadow"], keep))
    assert_bits(got, want, "cfg1 Default-scene band, DXC reading")
    W, H = 128, 8
    gb = synth.gbuffer(W, H, seed=5)
    pts = synth.point_lights(6, seed=5)
    for k in range(6):                                          # light k sits exactly above pixel (0, k) on the y axis: a zero component of Lw - P
        pts[k].position.set((gb[0][0, k, 0], gb[0][0, k, 1] + 3.0, gb[0][0, k, 2]))
    pf, extra = synth.per_frame(points=pts)
    pv = synth.per_view(W, H)
    with dxc_mode(ctx, False):
        assert_bits(ctx.forward_lighting([dev(g) for g in gb], pf, pv, out_fmt=abi.FMT_RGBA32F), O.forward_lighting(gb, pf, pv, abi.FMT_RGBA32F), "axis-aligned lights, DXC reading")


def test_producer_and_fused_psmain(ctx):
    W, H, NM = 192, 24, 5
    ip = synth.interpolants(W, H, NM, seed=0xD8C)
    datas, texsets = synth.material_set(NM, seed=0xD8C, max_dim=64)
    dmats, keep, chains = (abi.MaterialDesc * NM)(), [], []
    for i, (d, ts) in enumerate(zip(datas, texsets)):
        dmats[i].data = d
        cs = {}
        for slot, img in ts.items():
            chain_g, nm = ctx.mip_chain_rgba8(dev(img))
            keep.append(chain_g)
            setattr(dmats[i], slot, abi.Texture2D(chain_g.data_ptr(), img.shape[1], img.shape[0], nm, 0))
            cs[slot] = (chain_g.cpu().numpy(), img.shape[1], img.shape[0], nm)
        chains.append(cs)
    hm = O.host_materials(datas, chains)
    ssao = synth.ssao_image(W, H)
    e = ref_cases.small_env()
    pf, extra = synth.per_frame(points=synth.point_lights(12, seed=0xD8C), hdri_offset=0.3)
    pv = synth.per_view(W, H, max_env_lod=e["spec_mips"])
    ipd = [dev(p) for p in ip]
    with dxc_mode(ctx):
        gb_o = O.gbuffer_from_materials(ip, hm, pf.fAmbientLightingFactor, ssao)
        gb_g = ctx.gbuffer_from_materials(ipd, dmats, pf.fAmbientLightingFactor, dev(ssao))
        for k in range(4):
            assert_bits(gb_g[k], gb_o[k], f"G-buffer plane {k}, DXC reading")
        want = O.forward_lighting(gb_o, pf, pv, F16, extra_point=extra, env=ref_cases.host_env(e))
        got = ctx.forward_lighting_from_materials(ipd, dmats, pf, pv, ssao=dev(ssao), out_fmt=F16, extra_point=extra, env=ref_cases.dev_env(e, keep))
    assert_bits(got, want, "PSMain as one kernel, DXC reading")
    lit = ctx.gbuffer_from_materials(ipd, dmats, pf.fAmbientLightingFactor, dev(ssao))
    assert not np.array_equal(lit[1].cpu().numpy(), gb_g[1].cpu().numpy())


def test_ssr_environment_fallback(ctx):
    W, H = 320, 20
    e = ref_cases.small_env()
    keep = []
    scene, depth, packed, _ = synth.ssr_surfaces(W, H, seed=0xD8C)
    scene = scene.astype(np.float16)
    cb = synth.ssr_constants(W, H, e["spec_mips"])
    with dxc_mode(ctx):
        want = O.ssr_environment_fallback(scene, F16, depth, packed, abi.FMT_R10G10B10A2_UNORM, cb, ref_cases.host_env(e), abi.FMT_RGBA32F)
        got = ctx.ssr_environment_fallback(dev(scene), F16, dev(depth), dev(packed.view(np.int32)), abi.FMT_R10G10B10A2_UNORM, cb, ref_cases.dev_env(e, keep), abi.FMT_RGBA32F)
    assert_bits(got, want, "SSR environment fallback, DXC reading")


@pytest.mark.parametrize("tag", TAGS)
def test_product_in_dxc_mode_matches_the_dxc_build_of_the_reference(ctx, tag):
    fixtures = np.load(FIX)
    build, _, _ = ref_cases.DXC_SCENES[tag]
    inp = build()
    keep = []
    with dxc_mode(ctx):
        gb = boundary_gbuffer(inp)
        sh = ref_cases.dev_shadow_dims(inp["shadow"], keep) if inp["shadow"] is not None else None
        scene = ctx.forward_lighting([dev(g) for g in gb], inp["pf"], inp["pv"], out_fmt=F16, extra_point=inp["extra"], env=ref_cases.dev_env(inp["env"], keep), shadow=sh)
        want = O.forward_lighting(gb, inp["pf"], inp["pv"], F16, extra_point=inp["extra"], env=ref_cases.host_env(inp["env"]),
                                  shadow=ref_cases.host_shadow_dims(inp["shadow"]) if inp["shadow"] is not None else None)
    assert_bits(scene, want, f"{tag}: product vs oracle, DXC reading")
    d = distance(scene.cpu().numpy()[..., :3], fixtures[tag + "/scene"])
    assert d["max"] <= 1 and d["frac_gt0"] <= MAX_FRACTION and d["nonfinite_mismatch"] == 0, d


def test_set_arithmetic_and_set_option_argument_checks(ctx):
    assert ctx.lib.vqhip_set_arithmetic(ctx._h, 7) == abi.VQHIP_ERR_INVALID_ARG
    assert ctx.lib.vqhip_set_option(ctx._h, b"no_such_key", b"1") == abi.VQHIP_ERR_INVALID_ARG and b"unknown key" in ctx.lib.vqhip_last_error(ctx._h)
    assert ctx.lib.vqhip_set_option(ctx._h, b"shade_wg", b"100") == abi.VQHIP_ERR_INVALID_ARG
    assert ctx.lib.vqhip_set_option(ctx._h, b"lut_form", b"bogus") == abi.VQHIP_ERR_INVALID_ARG
    assert ctx.lib.vqhip_set_option(ctx._h, None, b"1") == abi.VQHIP_ERR_INVALID_ARG
    assert ctx.lib.vqhip_set_option(ctx._h, b"post_one_kernel", b"1") == abi.VQHIP_ERR_INVALID_ARG      # a form removed in round 4 is an unknown key
    for k, v in (("shade_wg", "128"), ("blur_y_wgs", "300"), ("diffuse_form", "texels"), ("diffuse_seq_form", "lane"), ("lut_form", "general"), ("psmain_waves", "6")):
        ctx.set_option(k, v)
        ctx.set_option(k, None)
        ctx.set_option(k, "default")
