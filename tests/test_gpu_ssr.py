"""-m gpu: vqhip_ssr_environment_fallback (csrc/ssr.hip) through the C ABI against the CPU oracle, bit for bit — 1280 x 720 and 3840 x 2160 frames against
the BASELINE config-4 cube, both normal formats, both scene-colour formats, pitched buffers, the extracted roughness, non-finite inputs, the
argument checks. The reference's own outputs for bands of these frames: tests/test_ref_fixtures.py (cases ssr_env_fallback_*)."""
import numpy as np
import pytest
import torch

from tests import oracle_lib as O
from tests import ref_cases
from vqengine_amd import abi, capi, synth

pytestmark = pytest.mark.gpu
dev = ref_cases._dev


def assert_bits(got, ref, what):
    n, idx = O.bits_equal(got.cpu().numpy() if hasattr(got, "cpu") else got, ref)
    assert n == 0, f"{what}: {n} mismatching elements, first {idx.tolist()}"


@pytest.fixture(scope="module")
def cfg4(ctx):
    e = ref_cases.cfg4_env()
    keep = []
    return {"e": e, "henv": ref_cases.host_env(e), "denv": ref_cases.dev_env(e, keep), "keep": keep}


@pytest.mark.parametrize("W,H", [(1280, 720), (3840, 2160)])
def test_full_frames_match_oracle(ctx, cfg4, W, H):
    scene, depth, packed, _ = synth.ssr_surfaces(W, H, seed=0x5500 + W)
    scene = scene.astype(np.float16)
    cb = synth.ssr_constants(W, H, cfg4["e"]["spec_mips"])
    want, r8 = O.ssr_environment_fallback(scene, abi.FMT_RGBA16F, depth, packed, abi.FMT_R10G10B10A2_UNORM, cb, cfg4["henv"], abi.FMT_RGBA16F, extract_roughness=True)
    got, g8 = ctx.ssr_environment_fallback(dev(scene), abi.FMT_RGBA16F, dev(depth), dev(packed.view(np.int32)), abi.FMT_R10G10B10A2_UNORM, cb, cfg4["denv"],
                                           abi.FMT_RGBA16F, extract_roughness=True)
    assert_bits(got, want, f"ssr fallback {W}x{H}")
    assert_bits(g8, r8, f"extracted roughness {W}x{H}")
    assert (want[..., :3].astype(np.float32).sum(-1) > 0).mean() > 0.6


@pytest.mark.parametrize("scene_fmt", [abi.FMT_RGBA16F, abi.FMT_RGBA32F])
@pytest.mark.parametrize("normal_fmt", [abi.FMT_R10G10B10A2_UNORM, abi.FMT_RGBA32F])
@pytest.mark.parametrize("out_fmt", [abi.FMT_RGBA16F, abi.FMT_RGBA32F])
def test_formats(ctx, scene_fmt, normal_fmt, out_fmt):
    W, H = 333, 37                                           # not a multiple of the 256-lane workgroup
    e = ref_cases.small_env()
    keep = []
    denv, henv = ref_cases.dev_env(e, keep), ref_cases.host_env(e)
    scene, depth, packed, n01 = synth.ssr_surfaces(W, H, seed=77)
    scene = scene.astype(np.float16 if scene_fmt == abi.FMT_RGBA16F else np.float32)
    normals = packed if normal_fmt == abi.FMT_R10G10B10A2_UNORM else n01
    cb = synth.ssr_constants(W, H, e["spec_mips"], hdri_yaw=-1.1, roughness_threshold=0.35)
    want = O.ssr_environment_fallback(scene, scene_fmt, depth, normals, normal_fmt, cb, henv, out_fmt)
    got = ctx.ssr_environment_fallback(dev(scene), scene_fmt, dev(depth), dev(normals.view(np.int32) if normals.dtype == np.uint32 else normals), normal_fmt, cb, denv, out_fmt)
    assert_bits(got, want, f"ssr fallback formats {scene_fmt}/{normal_fmt}/{out_fmt}")


def test_fresnel_pow_mode_and_mip_counts(ctx):
    """vqhip_set_fresnel_pow applies (FresnelWithRoughness, BRDF.hlsl:155); envMapSpecularIrradianceCubemapMipLevelCount below the cube's mip count scales the lod"""
    W, H = 200, 16
    e = ref_cases.small_env()
    keep = []
    denv, henv = ref_cases.dev_env(e, keep), ref_cases.host_env(e)
    scene, depth, packed, _ = synth.ssr_surfaces(W, H, seed=78)
    scene = scene.astype(np.float16)
    for mips in range(1, e["spec_mips"] + 1):
        cb = synth.ssr_constants(W, H, mips)
        for mode in (1, 0):
            O.load().vqo_set_fresnel_pow(mode); ctx.set_fresnel_pow(bool(mode))
            try:
                want = O.ssr_environment_fallback(scene, abi.FMT_RGBA16F, depth, packed, abi.FMT_R10G10B10A2_UNORM, cb, henv, abi.FMT_RGBA32F)
                got = ctx.ssr_environment_fallback(dev(scene), abi.FMT_RGBA16F, dev(depth), dev(packed.view(np.int32)), abi.FMT_R10G10B10A2_UNORM, cb, denv, abi.FMT_RGBA32F)
            finally:
                O.load().vqo_set_fresnel_pow(0); ctx.set_fresnel_pow(False)
            assert_bits(got, want, f"ssr fallback, {mips} mips, fresnel pow mode {mode}")


def test_nonfinite_and_degenerate_inputs(ctx):
    """NaN / inf roughness and depth, zero normals (0.5 decodes to a near-zero vector: normalize of ~0), depth exactly 1 and just below: same bits as the oracle"""
    W, H = 256, 8
    e = ref_cases.small_env()
    keep = []
    denv, henv = ref_cases.dev_env(e, keep), ref_cases.host_env(e)
    scene, depth, packed, n01 = synth.ssr_surfaces(W, H, seed=79)
    scene = scene.astype(np.float32)
    scene[0, :8, 3] = [np.nan, np.inf, -np.inf, 0.0, -0.0, 1.0, 0.2, np.nextafter(np.float32(0.2), np.float32(0))]
    depth[1, :6] = [np.nan, np.inf, -1.0, 1.0, np.nextafter(np.float32(1), np.float32(0)), 0.0]
    n01[2, :4, :3] = [[0.5, 0.5, 0.5], [0, 0, 0], [1, 1, 1], [np.nan, 0.5, 0.5]]
    cb = synth.ssr_constants(W, H, e["spec_mips"])
    with np.errstate(all="ignore"):
        want = O.ssr_environment_fallback(scene, abi.FMT_RGBA32F, depth, n01, abi.FMT_RGBA32F, cb, henv, abi.FMT_RGBA32F)
    got = ctx.ssr_environment_fallback(dev(scene), abi.FMT_RGBA32F, dev(depth), dev(n01), abi.FMT_RGBA32F, cb, denv, abi.FMT_RGBA32F)
    g, w = got.cpu().numpy(), want
    same = (g.view(np.uint32) == w.view(np.uint32)) | (np.isnan(g) & np.isnan(w))
    assert same.all(), np.argwhere(~same)[:5]


def test_pitched_buffers(ctx):
    W, H, P = 300, 12, 320
    e = ref_cases.small_env()
    keep = []
    denv, henv = ref_cases.dev_env(e, keep), ref_cases.host_env(e)
    scene, depth, packed, _ = synth.ssr_surfaces(W, H, seed=80)
    scene = scene.astype(np.float16)
    cb = synth.ssr_constants(W, H, e["spec_mips"])
    want = O.ssr_environment_fallback(scene, abi.FMT_RGBA16F, depth, packed, abi.FMT_R10G10B10A2_UNORM, cb, henv, abi.FMT_RGBA16F)

    def pitched(a, fill):
        out = torch.full((H, P) + tuple(a.shape[2:]), fill, dtype=a.dtype, device="cuda")
        out[:, :W] = a
        return out
    sc, dp, nm = pitched(dev(scene), 7.0), pitched(dev(depth), 0.5), pitched(dev(packed.view(np.int32)), 0)
    out = torch.full((H, P, 4), -1.0, dtype=torch.float16, device="cuda")
    rc = ctx.lib.vqhip_ssr_environment_fallback(ctx._h, None, sc.data_ptr(), abi.FMT_RGBA16F, P, dp.data_ptr(), P, nm.data_ptr(), abi.FMT_R10G10B10A2_UNORM, P,
                                                W, H, cb, denv, out.data_ptr(), abi.FMT_RGBA16F, P, None)
    assert rc == 0, ctx.lib.vqhip_last_error(ctx._h)
    torch.cuda.synchronize()
    assert_bits(out[:, :W].contiguous(), want, "pitched ssr fallback")
    assert (out[:, W:] == -1.0).all()


def test_argument_checks(ctx):
    W, H = 64, 4
    e = ref_cases.small_env()
    keep = []
    denv = ref_cases.dev_env(e, keep)
    scene, depth, packed, _ = synth.ssr_surfaces(W, H)
    sc, dp, nm = dev(scene.astype(np.float16)), dev(depth), dev(packed.view(np.int32))
    out = torch.empty((H, W, 4), dtype=torch.float16, device="cuda")
    cb = synth.ssr_constants(W, H, e["spec_mips"])

    def call(scene_fmt=abi.FMT_RGBA16F, normal_fmt=abi.FMT_R10G10B10A2_UNORM, out_fmt=abi.FMT_RGBA16F, cb=cb, env=denv, pitch=0, scene=sc):
        return ctx.lib.vqhip_ssr_environment_fallback(ctx._h, None, scene.data_ptr() if scene is not None else None, scene_fmt, pitch, dp.data_ptr(), 0, nm.data_ptr(), normal_fmt, 0,
                                                      W, H, cb, env, out.data_ptr(), out_fmt, 0, None)
    assert call() == 0
    assert call(scene=None) == abi.VQHIP_ERR_INVALID_ARG
    assert call(scene_fmt=abi.FMT_RGBA8_UNORM) == abi.VQHIP_ERR_UNSUPPORTED
    assert call(normal_fmt=abi.FMT_RGBA16F) == abi.VQHIP_ERR_UNSUPPORTED
    assert call(out_fmt=abi.FMT_R10G10B10A2_UNORM) == abi.VQHIP_ERR_UNSUPPORTED
    assert call(pitch=W - 1) == abi.VQHIP_ERR_INVALID_ARG
    assert call(cb=synth.ssr_constants(W, H, e["spec_mips"] + 1)) == abi.VQHIP_ERR_INVALID_ARG and b"MipLevelCount" in ctx.lib.vqhip_last_error(ctx._h)
    assert call(cb=synth.ssr_constants(W, H, 0)) == abi.VQHIP_ERR_INVALID_ARG
    noenv = abi.EnvMap(None, 0, None, 0, 0, None, 0)
    assert call(env=noenv) == abi.VQHIP_ERR_INVALID_ARG
    torch.cuda.synchronize()
