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
