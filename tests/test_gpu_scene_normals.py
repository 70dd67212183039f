"""-m gpu: vqhip_scene_normals_from_materials (csrc/gbuffer.hip:k_scene_normals_from_materials) — the pixel shader of the Z pre-pass,
Shaders/DepthPrePass.hlsl:PSMain :153-171 — through the C ABI against oracle/vqo_gbuffer.cpp:scene_normal_pixel, word for word: Tex_SceneNormals as the
R10G10B10A2_UNORM words the engine stores and as unquantised float4, both arithmetic readings, the alpha-masked permutation, adversarial interpolants,
pitched output, and the chain it exists for: its output is the `g_normal` of vqhip_ssr_environment_fallback. The reference's own shader on the same
inputs: tests/test_ref_fixtures.py (cases prepass_normals, prepass_normals_alpha_masked)."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import oracle_lib as O
from tests import ref_cases
from tests.test_gpu_gbuffer import build_materials, dev
from vqengine_amd import abi, synth

pytestmark = pytest.mark.gpu


def words(t):
    return t.cpu().numpy().view(np.uint32)


def unpack(w):
    return np.stack([(w & 1023), (w >> 10) & 1023, (w >> 20) & 1023], -1).astype(np.float32) / np.float32(1023.0)


@pytest.mark.parametrize("shape", [(1920, 1080), (333, 127), (130, 3), (1, 1), (2, 257)])
def test_scene_normals_match_oracle(ctx, shape):
    W, H = shape
    ip = synth.interpolants(W, H, 6)
    _, _, hmats, dmats, keep = build_materials(ctx, 6)
    ipd = [dev(p) for p in ip]
    want = O.scene_normals_from_materials(ip, hmats)
    got = ctx.scene_normals_from_materials(ipd, dmats)
    assert np.array_equal(words(got), want), f"{int((words(got) != want).sum())} of {want.size} words differ"
    want32 = O.scene_normals_from_materials(ip, hmats, abi.FMT_RGBA32F)
    n, idx = O.bits_equal(ctx.scene_normals_from_materials(ipd, dmats, abi.FMT_RGBA32F).cpu().numpy(), want32)
    assert n == 0, (n, idx)
    assert torch.equal(ipd[2].cpu().view(torch.int32), torch.from_numpy(ip[2]).view(torch.int32)), "the coverage plane must not be modified by the pre-pass"
    covered = (want >> 30) == 3
    idx = ip[2][..., 3].view(np.int32)
    assert np.array_equal(covered, (idx >= 0) & (idx < 6)) and ((want == 0) | covered).all()


def test_unbiased_normal_map_fetch_and_agreement_with_the_gbuffer(ctx):
    """the pre-pass fetches the normal map WITHOUT normalMapMipBias (DepthPrePass.hlsl:164): for materials whose bias is 0 the stored normal is the G-buffer's
    Surface.N packed, (N + 1) * 0.5; for biased materials on minified pixels it is not (so the test would notice a SampleBias)."""
    W, H, NM = 640, 360, 6
    ip = synth.interpolants(W, H, NM)
    datas, _, hmats, dmats, keep = build_materials(ctx, NM)
    ipd = [dev(p) for p in ip]
    n32 = ctx.scene_normals_from_materials(ipd, dmats, abi.FMT_RGBA32F).cpu().numpy()
    gb = ctx.gbuffer_from_materials(ipd, dmats, 0.05, None)
    packed_from_gb = ((gb[1][..., :3] + 1.0) * 0.5).cpu().numpy()
    idx = ip[2][..., 3].view(np.int32)
    bias = np.array([d.normalMapMipBias for d in datas], np.float32)
    assert (bias == 0).any() and (bias != 0).any()
    valid = (idx >= 0) & (idx < NM)
    unbiased = valid & (bias[np.clip(idx, 0, NM - 1)] == 0)
    biased = valid & ~unbiased
    assert np.array_equal(n32[unbiased][:, :3].view(np.uint32), packed_from_gb[unbiased].view(np.uint32))
    assert (n32[biased][:, :3] != packed_from_gb[biased]).any()


def test_alpha_masked_permutation_discards_the_same_fragments_as_the_lighting_pass(ctx):
    c = [c for c in ref_cases.CASES if c.name == "prepass_normals_alpha_masked"][0]
    i = c.build()
    got = c.product(ctx, i)
    assert np.array_equal(got, c.oracle(i))
    lit = [c2 for c2 in ref_cases.CASES if c2.name == "psmain_alpha_masked"][0]
    # the lighting producer marks its discards -1 in ip2.w: the same pixels hold the clear value here
    keep = []
    dm = (abi.MaterialDesc * 5)()
    for k, (d, ts) in enumerate(zip(i["datas"], i["tex"])):
        dm[k].data = d
        for slot, img in ts.items():
            chain, n = ctx.mip_chain_rgba8(dev(img))
            keep.append(chain)
            setattr(dm[k], slot, abi.Texture2D(chain.data_ptr(), img.shape[1], img.shape[0], n, 0))
        dm[k].texDiffuse.reserved = abi.MATERIAL_ALPHA_MASKED
    ipd = [dev(p) for p in i["ip"]]
    ctx.gbuffer_from_materials(ipd, dm, 0.05, None)
    after = ipd[2].cpu().numpy()[..., 3].view(np.int32)
    before = i["ip"][2][..., 3].view(np.int32)
    valid = (before >= 0) & (before < 5)
    assert np.array_equal((got >> 30) == 3, valid & (after != -1))
    assert (valid & (after == -1)).any() and lit is not None


def test_both_arithmetic_readings(ctx):
    """vqhip_set_arithmetic(DXC): normalize = v * rsqrt(dot), FMA-chain dot inside UnpackNormal — bit-exact against the oracle in the same mode"""
    W, H = 512, 128
    ip = synth.interpolants(W, H, 6, seed=0x77)
    _, _, hmats, dmats, keep = build_materials(ctx, 6)
    lit = O.scene_normals_from_materials(ip, hmats, abi.FMT_RGBA32F)
    ctx.set_arithmetic(True); O.load().vqo_set_arithmetic(1)
    try:
        want = O.scene_normals_from_materials(ip, hmats, abi.FMT_RGBA32F)
        got = ctx.scene_normals_from_materials([dev(p) for p in ip], dmats, abi.FMT_RGBA32F).cpu().numpy()
    finally:
        ctx.set_arithmetic(False); O.load().vqo_set_arithmetic(0)
    n, idx = O.bits_equal(got, want)
    assert n == 0, (n, idx)
    assert (want != lit).any() and np.abs(want - lit).max() < 1e-6


def test_adversarial_interpolants(ctx):
    """random bit patterns, NaN / inf / zero normals and tangents, huge uv: same words as the oracle (NaN -> 0 in the UNORM conversion)"""
    W, H, NM = 192, 64, 6
    r = np.random.default_rng(7)
    ip = [p.copy() for p in synth.interpolants(W, H, NM)]
    idx = ip[2][..., 3].copy()
    specials = np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 1e-45, -1e-40, 1e30, -1e30, 3.4e38, 1e-30, 0.5, 1.0, 255.0, 65536.0], np.float32)
    for k in range(3):
        flat = ip[k].reshape(-1)
        sel = r.random(flat.size) < 0.15
        flat[sel] = r.integers(0, 2 ** 32, int(sel.sum()), dtype=np.uint32).view(np.float32)
        sel = r.random(flat.size) < 0.10
        flat[sel] = specials[r.integers(0, len(specials), int(sel.sum()))]
    ip[2][..., 3] = idx
    _, _, hmats, dmats, keep = build_materials(ctx, NM)
    with np.errstate(all="ignore"):
        want = O.scene_normals_from_materials(ip, hmats)
        want32 = O.scene_normals_from_materials(ip, hmats, abi.FMT_RGBA32F)
    ipd = [dev(p) for p in ip]
    assert np.array_equal(words(ctx.scene_normals_from_materials(ipd, dmats)), want)
    n, idx = O.bits_equal(ctx.scene_normals_from_materials(ipd, dmats, abi.FMT_RGBA32F).cpu().numpy(), want32)
    assert n == 0, (n, idx)


def test_feeds_the_ssr_fallback(ctx):
    """Z pre-pass normals -> SSR environment fallback: the product chain == the oracle chain (the normals target is what ClassifyReflectionTiles.hlsl:80 loads)"""
    W, H, NM = 320, 96, 6
    ip = synth.interpolants(W, H, NM)
    _, _, hmats, dmats, keep = build_materials(ctx, NM)
    e = ref_cases.small_env()
    denv, henv = ref_cases.dev_env(e, keep), ref_cases.host_env(e)
    scene, depth, _, _ = synth.ssr_surfaces(W, H, seed=3)
    scene = scene.astype(np.float16)
    cb = synth.ssr_constants(W, H, e["spec_mips"])
    n_o = O.scene_normals_from_materials(ip, hmats)
    n_g = ctx.scene_normals_from_materials([dev(p) for p in ip], dmats)
    want = O.ssr_environment_fallback(scene, abi.FMT_RGBA16F, depth, n_o, abi.FMT_R10G10B10A2_UNORM, cb, henv, abi.FMT_RGBA16F)
    got = ctx.ssr_environment_fallback(dev(scene), abi.FMT_RGBA16F, dev(depth), n_g, abi.FMT_R10G10B10A2_UNORM, cb, denv, abi.FMT_RGBA16F)
    n, idx = O.bits_equal(got.cpu().numpy(), want)
    assert n == 0, (n, idx)
    assert np.abs(unpack(n_o)[(n_o >> 30) == 3] * 2 - 1).max() <= 1.0 + 2e-3


def test_pitched_output_and_argument_checks(ctx):
    W, H, P = 100, 7, 128
    ip = synth.interpolants(W, H, 3)
    _, _, hmats, dmats, keep = build_materials(ctx, 3)
    ipd = [dev(p) for p in ip]
    want = O.scene_normals_from_materials(ip, hmats)
    out = torch.full((H, P), -1, dtype=torch.int32, device="cuda")
    inter = abi.Interpolants(ipd[0].data_ptr(), ipd[1].data_ptr(), ipd[2].data_ptr(), W, H, W)

    def call(o, fmt, pitch, n=3, mats=dmats):
        return ctx.lib.vqhip_scene_normals_from_materials(ctx._h, None, C.byref(inter), mats, n, o, fmt, pitch)
    assert call(out.data_ptr(), abi.FMT_R10G10B10A2_UNORM, P) == 0
    torch.cuda.synchronize()
    assert np.array_equal(words(out[:, :W].contiguous()), want) and (out[:, W:] == -1).all()
    assert call(None, abi.FMT_R10G10B10A2_UNORM, 0) == abi.VQHIP_ERR_INVALID_ARG
    assert call(out.data_ptr(), abi.FMT_RGBA16F, 0) == abi.VQHIP_ERR_UNSUPPORTED and b"outFmt" in ctx.lib.vqhip_last_error(ctx._h)
    assert call(out.data_ptr(), abi.FMT_R10G10B10A2_UNORM, W - 1) == abi.VQHIP_ERR_INVALID_ARG
    assert call(out.data_ptr(), abi.FMT_R10G10B10A2_UNORM, P, n=10 ** 6) == abi.VQHIP_ERR_INVALID_ARG
    assert call(out.data_ptr(), abi.FMT_R10G10B10A2_UNORM, P, n=2, mats=None) == abi.VQHIP_ERR_INVALID_ARG
    # no materials at all: nothing is covered, the target holds the clear value
    assert call(out.data_ptr(), abi.FMT_R10G10B10A2_UNORM, P, n=0, mats=None) == 0
    torch.cuda.synchronize()
    assert (out[:, :W] == 0).all()


def test_full_4k_frame_properties_and_bands(ctx):
    """3840 x 2160, 12 materials: whole-frame properties (covered <=> valid material index; the stored word == the float4 output quantised; unit normals to UNORM10
    precision) and three 32-row bands word for word against the oracle (bands start on even rows: the quad derivatives see the same neighbours)."""
    W, H, NM, BAND = 3840, 2160, 12, 540
    band = synth.interpolants(W, BAND, NM)
    ip = [np.tile(p, (H // BAND, 1, 1)) for p in band]
    _, _, hmats, dmats, keep = build_materials(ctx, NM, max_dim=512)
    ipd = [dev(p) for p in ip]
    w = words(ctx.scene_normals_from_materials(ipd, dmats))
    f = ctx.scene_normals_from_materials(ipd, dmats, abi.FMT_RGBA32F).cpu().numpy()
    idx = ip[2][..., 3].view(np.int32)
    covered = (idx >= 0) & (idx < NM)
    assert np.array_equal((w >> 30) == 3, covered) and (w[~covered] == 0).all()
    assert np.array_equal(w, ref_cases.pack_r10g10b10a2(f))
    n = unpack(w)[covered] * 2.0 - 1.0
    assert np.abs(np.linalg.norm(n.astype(np.float64), axis=-1) - 1.0).max() < 4e-3
    for r0 in (0, 1052, 2128):
        sub = [p[r0:r0 + 32] for p in ip]
        assert np.array_equal(w[r0:r0 + 32], O.scene_normals_from_materials(sub, hmats)), r0
