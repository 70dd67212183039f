"""Row pitch: every entry point that takes `row_pitch_px` must give the same pixels for a padded layout as for the dense one
and must not touch the padding (the reference's render targets are pitched resources)."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.test_gpu_gbuffer import build_materials, dev
from vqengine_amd import abi, capi, synth

pytestmark = pytest.mark.gpu


def _padded(arr, pitch, fill):
    h, w = arr.shape[:2]
    out = np.full((h, pitch) + arr.shape[2:], fill, arr.dtype)
    out[:, :w] = arr
    return out


def test_forward_lighting_pitched_planes(ctx):
    W, H, GP, OP = 203, 37, 211, 209
    gb = synth.gbuffer(W, H, seed=0x917C)
    pf, extra = synth.per_frame(points=synth.point_lights(9, seed=0x917C))
    pv = synth.per_view(W, H)
    dense = ctx.forward_lighting([dev(g) for g in gb], pf, pv, out_fmt=abi.FMT_RGBA16F).cpu().numpy()
    gpad = [dev(_padded(g, GP, np.float32(np.nan))) for g in gb]
    out = torch.full((H, OP, 4), 7.0, dtype=torch.float16, device="cuda")
    g = abi.GBuffer(gpad[0].data_ptr(), gpad[1].data_ptr(), gpad[2].data_ptr(), gpad[3].data_ptr(), W, H, GP)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = ctx.lib.vqhip_forward_lighting(ctx._h, st, C.byref(g), C.byref(pf), C.byref(pv), None, 0, None, None, C.c_void_p(out.data_ptr()), OP, abi.FMT_RGBA16F)
    assert rc == 0, ctx.lib.vqhip_last_error(ctx._h)
    o = out.cpu().numpy()
    assert np.array_equal(o[:, :W].view(np.uint16), dense.view(np.uint16))
    assert np.all(o[:, W:] == np.float16(7.0))                       # padding untouched
    # pitch smaller than the width is rejected
    g.row_pitch_px = W - 1
    assert ctx.lib.vqhip_forward_lighting(ctx._h, st, C.byref(g), C.byref(pf), C.byref(pv), None, 0, None, None, C.c_void_p(out.data_ptr()), OP, abi.FMT_RGBA16F) == abi.VQHIP_ERR_INVALID_ARG


def test_constant_ring_back_to_back_calls_see_their_own_constants(ctx):
    """600 forward-lighting calls enqueued without any host synchronisation, each with different cbuffer contents (the light's
    brightness and position change per call), wrap the 32-slot constant ring many times while uploads run on the context's copy
    stream. Every result must equal the one obtained with a device synchronisation after every call."""
    W, H, N = 96, 16, 600
    gb = [dev(g) for g in synth.gbuffer(W, H, seed=0xA11)]
    pts = synth.point_lights(3, seed=0xA11)
    pv = synth.per_view(W, H)
    frames = []
    for i in range(N):
        pts[0].brightness = 100.0 + i
        pts[1].position.x = -40.0 + 0.1 * i
        pf, _ = synth.per_frame(points=pts)
        frames.append(pf)
    outs = [ctx.forward_lighting(gb, frames[i], pv, out_fmt=abi.FMT_RGBA32F) for i in range(N)]      # no sync in between
    torch.cuda.synchronize()
    got = torch.stack(outs).cpu().numpy()
    for i in list(range(0, N, 37)) + [N - 1]:
        ref = ctx.forward_lighting(gb, frames[i], pv, out_fmt=abi.FMT_RGBA32F)
        torch.cuda.synchronize()
        assert np.array_equal(got[i].view(np.uint32), ref.cpu().numpy().view(np.uint32)), i
    assert not np.array_equal(got[0], got[1])                     # the constants really differed


def test_gbuffer_producer_and_skydome_pitched(ctx):
    import math
    from vqengine_amd import scene
    W, H, NM, IP, OP = 150, 31, 4, 163, 157
    ip = synth.interpolants(W, H, NM)
    datas, host_chains, hmats, dmats, keep = build_materials(ctx, NM, max_dim=64)
    dense = [t.cpu().numpy() for t in ctx.gbuffer_from_materials([dev(p) for p in ip], dmats, 0.055, None)]
    ipad = []
    for k, p in enumerate(ip):
        q = _padded(p, IP, np.float32(0.0))
        if k == 2:
            q[:, W:, 3] = np.full((H, IP - W), -1, np.int32).view(np.float32)     # padding pixels: "no geometry"
        ipad.append(dev(q))
    outs = [torch.full((H, OP, 4), 5.0, dtype=torch.float32, device="cuda") for _ in range(4)]
    inter = abi.Interpolants(ipad[0].data_ptr(), ipad[1].data_ptr(), ipad[2].data_ptr(), W, H, IP)
    gb = abi.GBuffer(outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(), W, H, OP)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert ctx.lib.vqhip_gbuffer_from_materials(ctx._h, st, C.byref(inter), dmats, NM, 0.055, None, C.byref(gb)) == 0
    for k in range(4):
        o = outs[k].cpu().numpy()
        # the quad derivative of the last column differs: dense has no x-partner for an odd last column, the pitched layout
        # neither (partner index is taken inside the image width) -> identical everywhere
        assert np.array_equal(o[:, :W].view(np.uint32), dense[k].view(np.uint32)), k
        assert np.all(o[:, W:] == 5.0)
    # skydome into a pitched colour target with pitched coverage planes
    eq = dev(synth.equirect(64, 32))
    sp = scene.skydome_params(0.3, 0.1, 0.0, 60.0 * math.pi / 180.0, W, H)
    dense_c = ctx.skydome(eq, sp, torch.zeros((H, W, 4), dtype=torch.float16, device="cuda"), abi.FMT_RGBA16F, coverage_ip=[dev(p) for p in ip]).cpu().numpy()
    col = torch.full((H, OP, 4), 3.0, dtype=torch.float16, device="cuda")
    col[:, :W] = 0
    cov = abi.Interpolants(ipad[0].data_ptr(), ipad[1].data_ptr(), ipad[2].data_ptr(), W, H, IP)
    assert ctx.lib.vqhip_skydome(ctx._h, st, C.c_void_p(eq.data_ptr()), 64, 32, C.byref(sp), C.byref(cov), C.c_void_p(col.data_ptr()), W, H, OP, abi.FMT_RGBA16F) == 0
    c = col.cpu().numpy()
    assert np.array_equal(c[:, :W].view(np.uint16), dense_c.view(np.uint16))
    assert np.all(c[:, W:] == np.float16(3.0))
