"""Light gizmo composite (Unlit.hlsl:PSMain over the engine's coverage plane; SURVEY.md §8(f).2, SceneRendering.cpp:1787-1819)."""
import ctypes as C

import numpy as np
import pytest

from tests import oracle_lib as O
from tests import ref_lib as R
from vqengine_amd import abi, synth


def _scene(w=96, h=40, seed=4):
    rng = np.random.default_rng(seed)
    ip = [p.copy() for p in synth.interpolants(w, h, 4)]
    idx = ip[2][..., 3].view(np.int32)
    idx[5:12, 10:30] = -2                      # gizmo 0
    idx[20:33, 50:61] = -3                     # gizmo 1
    idx[0, 0] = -66                            # gizmo 64: beyond the table -> untouched
    idx[1, 1] = -(2 + 5)                       # gizmo 5 with only 3 colours passed -> untouched
    colors = [(1.0, 0.9, 0.1, 1.0), (0.2, 0.4, 1.5, 0.5), (9.0, 9.0, 9.0, 9.0)]
    base = rng.random((h, w, 4), dtype=np.float32)
    return ip, idx.copy(), colors, base


def test_oracle_unlit_composite_closed_form():
    ip, idx, colors, base = _scene()
    for fmt, dt in ((abi.FMT_RGBA32F, np.float32), (abi.FMT_RGBA16F, np.float16)):
        img = base.astype(dt)
        out = O.unlit_composite(ip[2], colors, img.copy(), fmt)
        want = img.copy()
        want[idx == -2] = np.asarray(colors[0], dt)
        want[idx == -3] = np.asarray(colors[1], dt)
        assert np.array_equal(out.view(np.uint16 if dt == np.float16 else np.uint32), want.view(np.uint16 if dt == np.float16 else np.uint32))
        assert (idx == -2).sum() == 140 and np.array_equal(out[0, 0], img[0, 0]) and np.array_equal(out[1, 1], img[1, 1])


@pytest.mark.skipif(not R.available("shaders"), reason="oracle/_ref is built only where /root/reference exists")
def test_reference_unlit_psmain_returns_the_cbuffer_colour():
    lib = R.load()
    for c in ((1.0, 0.9, 0.1, 1.0), (0.0, -2.5, 3e4, 0.25)):
        a, o = np.asarray(c, np.float32), np.zeros(4, np.float32)
        lib.vqref_unlit_color(a.ctypes.data, o.ctypes.data)
        assert np.array_equal(a.view(np.uint32), o.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", [abi.FMT_RGBA32F, abi.FMT_RGBA16F])
def test_hip_unlit_composite_matches_oracle(ctx, fmt):
    import torch
    ip, idx, colors, base = _scene(w=333, h=77)
    dt = np.float32 if fmt == abi.FMT_RGBA32F else np.float16
    img = base.astype(dt)
    want = O.unlit_composite(ip[2], colors, img.copy(), fmt)
    dev = [torch.from_numpy(p).cuda() for p in ip]
    got = ctx.unlit_composite(dev, colors, torch.from_numpy(img.copy()).cuda(), fmt).cpu().numpy()
    n, where = O.bits_equal(got, want)
    assert n == 0, (n, where)
    # sky, gizmos and geometry partition the frame: the skydome leaves gizmo pixels alone and vice versa
    lib, st = ctx.lib, None
    cov = abi.Interpolants(dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), 333, 77, 333)
    col = torch.zeros((77, 333, 4), dtype=torch.float32, device="cuda")
    arr = (abi.float4 * 65)()
    assert lib.vqhip_unlit_composite(ctx._h, st, C.byref(cov), C.cast(arr, C.c_void_p), 65, col.data_ptr(), 333, 77, 333, abi.FMT_RGBA32F) == abi.VQHIP_ERR_INVALID_ARG
    assert lib.vqhip_unlit_composite(ctx._h, st, None, C.cast(arr, C.c_void_p), 1, col.data_ptr(), 333, 77, 333, abi.FMT_RGBA32F) == abi.VQHIP_ERR_INVALID_ARG
    assert lib.vqhip_unlit_composite(ctx._h, st, C.byref(cov), C.cast(arr, C.c_void_p), 1, col.data_ptr(), 333, 77, 333, abi.FMT_RGBA8_UNORM) == abi.VQHIP_ERR_UNSUPPORTED
    assert lib.vqhip_unlit_composite(ctx._h, st, C.byref(cov), None, 0, col.data_ptr(), 333, 77, 333, abi.FMT_RGBA32F) == abi.VQHIP_OK
