"""GPU parity of the G-buffer producer: vqhip_gbuffer_from_materials and vqhip_mip_chain_box_rgba8,
called through the C ABI, against oracle/vqo_gbuffer.cpp on the same seeded inputs. Bar: identical bits in all four planes."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import oracle_lib as O
from vqengine_amd import abi, capi, synth

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def build_materials(ctx, n, seed=0x3A7, max_dim=256):
    """Level-0 textures -> mip chains on BOTH sides (oracle on the host, vqhip_mip_chain_box_rgba8 on the GPU), compared
    bit for bit, then the two material tables."""
    datas, texsets = synth.material_set(n, seed=seed, max_dim=max_dim)
    host_chains, dev_keep = [], []
    dmats = (abi.MaterialDesc * n)()
    for i, (d, ts) in enumerate(zip(datas, texsets)):
        cs = {}
        dmats[i].data = d
        for slot, img in ts.items():
            chain_o, nm = O.mip_chain_rgba8(img)
            chain_g, nm_g = ctx.mip_chain_rgba8(dev(img))
            assert nm == nm_g
            assert np.array_equal(chain_g.cpu().numpy(), chain_o), (i, slot)
            cs[slot] = (chain_o, img.shape[1], img.shape[0], nm)
            dev_keep.append(chain_g)
            setattr(dmats[i], slot, abi.Texture2D(chain_g.data_ptr(), img.shape[1], img.shape[0], nm, 0))
        host_chains.append(cs)
    return datas, host_chains, O.host_materials(datas, host_chains), dmats, dev_keep


def check(ctx, W, H, n_mat, with_ssao, seed=0x1A7E, max_dim=256):
    ip = synth.interpolants(W, H, n_mat, seed=seed)
    datas, host_chains, hmats, dmats, keep = build_materials(ctx, n_mat, max_dim=max_dim)
    ssao = synth.ssao_image(W, H) if with_ssao else None
    ref = O.gbuffer_from_materials(ip, hmats, 0.055, ssao)
    got = ctx.gbuffer_from_materials([dev(p) for p in ip], dmats, 0.055, dev(ssao) if with_ssao else None)
    torch.cuda.synchronize()
    for k in range(4):
        g = got[k].cpu().numpy()
        n, idx = O.bits_equal(g, ref[k])
        assert n == 0, f"gb{k}: {n} mismatches of {g.size}; first {idx.tolist()} gpu={[g[tuple(i)] for i in idx]} ref={[ref[k][tuple(i)] for i in idx]}"
    return ip, ref, got


@pytest.mark.parametrize("shape", [(640, 360), (333, 127), (130, 3), (1, 1), (2, 257)])
@pytest.mark.parametrize("with_ssao", [False, True])
def test_gbuffer_producer_matches_oracle(ctx, shape, with_ssao):
    check(ctx, shape[0], shape[1], 6, with_ssao)


def test_gbuffer_producer_many_materials_per_wave(ctx):
    """16 materials in 97x61 blocks + stray indices: waves waterfall over several material records."""
    ip, ref, got = check(ctx, 512, 256, 16, True, max_dim=128)
    idx = np.ascontiguousarray(ip[2][..., 3]).view(np.int32)
    assert len(np.unique(idx)) >= 17


def test_gbuffer_producer_fuzz_adversarial_inputs(ctx):
    """Interpolants made of random bit patterns, NaN / inf / denormals, zero and huge uv, zero normals and tangents: the HIP
    kernel and the oracle must still agree bit for bit (NaN == NaN)."""
    W, H, NM = 192, 64, 6
    r = np.random.default_rng(99)
    ip = [p.copy() for p in synth.interpolants(W, H, NM)]
    idx = ip[2][..., 3].copy()
    specials = np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 1e-45, -1e-40, 1e30, -1e30, 3.4e38, 1e-30, 0.5, 1.0, 255.0, 65536.0], np.float32)
    for k in range(3):
        flat = ip[k].reshape(-1)
        sel = r.random(flat.size) < 0.15
        flat[sel] = r.integers(0, 2 ** 32, int(sel.sum()), dtype=np.uint32).view(np.float32)        # arbitrary bit patterns
        sel = r.random(flat.size) < 0.10
        flat[sel] = specials[r.integers(0, len(specials), int(sel.sum()))]
    ip[1][10:14, :, :3] = 0.0                                                                         # zero normals
    ip[2][20:24, :, :3] = 0.0                                                                         # zero tangents
    ip[2][..., 3] = idx                                                                               # keep material indices valid
    datas, host_chains, hmats, dmats, keep = build_materials(ctx, NM, max_dim=64)
    ref = O.gbuffer_from_materials(ip, hmats, 0.055, None)
    got = ctx.gbuffer_from_materials([dev(p) for p in ip], dmats, 0.055, None)
    for k in range(4):
        g = got[k].cpu().numpy()
        n, where = O.bits_equal(g, ref[k])
        assert n == 0, f"gb{k}: {n} mismatches; first {where.tolist()} gpu={[g[tuple(i)] for i in where]} ref={[ref[k][tuple(i)] for i in where]}"
    assert np.isnan(ref[1]).any()                      # the fuzz really reached the NaN paths


def test_gbuffer_producer_feeds_forward_lighting(ctx):
    """Producer -> shade chain on the GPU equals the oracle's chain (planes handed over without leaving HBM)."""
    W, H = 320, 200
    ip, ref, got = check(ctx, W, H, 5, False)
    pf, extra = synth.per_frame(points=synth.point_lights(8))
    pv = synth.per_view(W, H)
    out_g = ctx.forward_lighting(got, pf, pv, out_fmt=abi.FMT_RGBA16F)
    out_o = O.forward_lighting(ref, pf, pv, abi.FMT_RGBA16F)
    n, idx = O.bits_equal(out_g.cpu().numpy(), out_o)
    assert n == 0, (n, idx)


def test_hip_path_reproduces_gbuffer_golden_fixture(ctx):
    """tests/golden/gbuffer_small.npz (committed, made by tests/golden/make_golden.py) without touching the oracle."""
    import os
    from tests.golden import make_golden as G
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(G.__file__)), "gbuffer_small.npz"))
    ip, datas, texsets, ssao = G.gbuffer_inputs()
    dmats = (abi.MaterialDesc * len(datas))()
    keep = []
    for i, (d, ts) in enumerate(zip(datas, texsets)):
        dmats[i].data = d
        for slot, img in ts.items():
            chain, nm = ctx.mip_chain_rgba8(dev(img))
            keep.append(chain)
            setattr(dmats[i], slot, abi.Texture2D(chain.data_ptr(), img.shape[1], img.shape[0], nm, 0))
    first = next(iter(texsets[0].values()))
    assert np.array_equal(keep[0].cpu().numpy()[first.shape[0] * first.shape[1]:], fx["mat0_first_chain_tail"])
    got = ctx.gbuffer_from_materials([dev(p) for p in ip], dmats, 0.055, dev(ssao))
    for k in range(4):
        n, idx = O.bits_equal(got[k].cpu().numpy(), fx[f"gb{k}"])
        assert n == 0, (k, n, idx)


@pytest.mark.parametrize("fmt", [abi.FMT_RGBA32F, abi.FMT_RGBA16F])
@pytest.mark.parametrize("shape", [(640, 360), (333, 61), (1, 1)])
def test_skydome_matches_oracle(ctx, fmt, shape):
    """vqhip_skydome against the oracle: full-screen sky and sky composited over a lit frame."""
    import math
    from vqengine_amd import scene
    W, H = shape
    eq = synth.equirect(256, 128)
    sp = scene.skydome_params(0.9, -0.35, 0.5, 65.0 * math.pi / 180.0, W, H)
    dt = np.float32 if fmt == abi.FMT_RGBA32F else np.float16
    ref = O.skydome(eq, sp, np.zeros((H, W, 4), dt), fmt, None)
    got = ctx.skydome(dev(eq), sp, torch.zeros((H, W, 4), dtype=torch.float32 if fmt == abi.FMT_RGBA32F else torch.float16, device="cuda"), fmt)
    n, idx = O.bits_equal(got.cpu().numpy(), ref)
    assert n == 0, (n, idx)
    ip = synth.interpolants(W, H, 3)
    base = synth.hdr_image(W, H).astype(dt)
    ref = O.skydome(eq, sp, base.copy(), fmt, ip[2])
    got = ctx.skydome(dev(eq), sp, dev(base), fmt, coverage_ip=[dev(p) for p in ip])
    n, idx = O.bits_equal(got.cpu().numpy(), ref)
    assert n == 0, (n, idx)


@pytest.mark.parametrize("rle", [True, False])
def test_hdr_ingest_matches_oracle(ctx, rle):
    """vqhip_hdr_decode_rgba32f against the oracle: every exponent (denormal scale factors included), runs,
    zero exponents; then the min-filter mip chain built on the decoded level 0."""
    h, w = 64, 256
    r = np.random.default_rng(11)
    rgbe = synth.float_to_rgbe(synth.equirect(w, h)[..., :3])
    rgbe[5] = r.integers(0, 256, (w, 4), dtype=np.uint8)
    rgbe[6, :, 3] = np.arange(w, dtype=np.uint8)               # all 256 exponents
    rgbe[7, :100] = (1, 2, 3, 200)
    data = synth.hdr_file_bytes(rgbe, rle=rle)
    ref = O.hdr_decode(data)
    got = ctx.load_hdr(data)
    n, idx = O.bits_equal(got.cpu().numpy(), ref)
    assert n == 0, (n, idx)
    chain_g, nm = ctx.mip_chain(got)
    chain_o, nm_o = O.mip_chain(ref)
    assert nm == nm_o
    n, idx = O.bits_equal(chain_g.cpu().numpy(), chain_o)
    assert n == 0, (n, idx)


def _hdr_with_groups(rgbe, r, style):
    """A run-length coded .hdr whose byte planes are cut into groups the way `style` says (every coding stb_image accepts, not only the tidy one of synth._rle_plane):
    'ones' = literal groups of one byte; 'max' = runs of 127 / literal groups of 128 wherever the data allows; 'mixed' = random group lengths, equal bytes coded as
    runs OR literals at random (a run of length 1 and 2 included)."""
    h, w = rgbe.shape[:2]
    head = b"#?RGBE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w)
    body = bytearray()
    for y in range(h):
        body += bytes((2, 2, w >> 8, w & 255))
        for k in range(4):
            row, i = rgbe[y, :, k], 0
            while i < w:
                same = 1
                while i + same < w and same < 127 and row[i + same] == row[i]:
                    same += 1
                if style == "ones":
                    n, run = 1, False
                elif style == "max":
                    run = same >= 2
                    n = same if run else min(128, w - i)
                else:
                    run = bool(r.integers(0, 2))
                    n = int(r.integers(1, same + 1)) if run else int(r.integers(1, min(128, w - i) + 1))
                if run:
                    body += bytes((128 + n, int(row[i])))
                else:
                    body += bytes((n,)) + row[i:i + n].tobytes()
                i += n
    return head + bytes(body)


@pytest.mark.parametrize("style", ["ones", "max", "mixed"])
@pytest.mark.parametrize("shape", [(3, 8), (5, 9), (2, 255), (7, 300), (3, 2048), (2, 4097), (1, 19000)])
def test_hdr_run_expansion_on_the_gpu(ctx, shape, style):
    """The GPU run expansion (hdri.hip:k_hdr_expand: a workgroup per scanline, a wave per byte plane) against the oracle for every group structure stb_image accepts,
    widths from the smallest run-length coded one to the widest the LDS path takes, long constant stretches (runs of 127) and noise (literal groups of 128)."""
    h, w = shape
    r = np.random.default_rng(h * 1000 + w)
    rgbe = r.integers(0, 256, (h, w, 4), dtype=np.uint8)
    rgbe[:, w // 3: w // 3 + min(w // 2, 700)] = (7, 7, 9, 130)          # constant stretch: runs
    rgbe[0, :, 3] = 0                                                   # zero exponents
    data = _hdr_with_groups(rgbe, r, style)
    ref = O.hdr_decode(data)
    got = ctx.load_hdr(data)
    n, idx = O.bits_equal(got.cpu().numpy(), ref)
    assert n == 0, (n, idx)


def test_hdr_corrupt_files_get_the_oracles_verdict(ctx):
    """Truncation at every byte of a small run-length coded file and single corrupted count bytes: the C ABI accepts / refuses exactly the files the oracle
    accepts / refuses (the host walk of the run headers validates what the host expansion of round 4 validated), and what it accepts decodes to the same bits."""
    r = np.random.default_rng(5)
    rgbe = r.integers(0, 256, (3, 40, 4), dtype=np.uint8)
    rgbe[1, 5:30] = (1, 2, 3, 140)
    good = synth.hdr_file_bytes(rgbe)
    w, h, off = capi.hdr_parse_header(good)
    cases = [good[:n] for n in range(off, len(good))]
    for pos in range(off, len(good), 3):
        for v in (0, 129, 255, 128, 2):
            cases.append(good[:pos] + bytes((v,)) + good[pos + 1:])
    n_ok = 0
    for data in cases:
        try:
            ref = O.hdr_decode(data)
        except ValueError:
            ref = None
        try:
            got = ctx.load_hdr(data).cpu().numpy()
        except Exception:
            got = None
        assert (ref is None) == (got is None), (len(data), ref is None, got is None)
        if ref is not None:
            n_ok += 1
            assert O.bits_equal(got, ref)[0] == 0
    assert n_ok > 10


def test_hdr_ingest_errors(ctx):
    rgbe = synth.float_to_rgbe(synth.equirect(32, 4)[..., :3])
    good = synth.hdr_file_bytes(rgbe)
    out = torch.empty((4, 32, 4), dtype=torch.float32, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = C.c_void_p(out.data_ptr())
    lib = ctx.lib
    assert lib.vqhip_hdr_decode_rgba32f(ctx._h, st, good, len(good), p, 32, 4) == 0
    assert lib.vqhip_hdr_decode_rgba32f(ctx._h, st, good, len(good), p, 31, 4) == abi.VQHIP_ERR_INVALID_ARG
    assert lib.vqhip_hdr_decode_rgba32f(ctx._h, st, good[:-3], len(good) - 3, p, 32, 4) == abi.VQHIP_ERR_INVALID_ARG
    assert b"truncated" in lib.vqhip_last_error(ctx._h)
    bad = good.replace(b"#?RADIANCE", b"#?RADIANCX", 1)
    assert lib.vqhip_hdr_decode_rgba32f(ctx._h, st, bad, len(bad), p, 32, 4) == abi.VQHIP_ERR_INVALID_ARG
    w, h, off = capi.hdr_parse_header(good)
    corrupt = good[:off + 4] + bytes((128,)) + good[off + 5:]
    assert lib.vqhip_hdr_decode_rgba32f(ctx._h, st, corrupt, len(corrupt), p, 32, 4) == abi.VQHIP_ERR_INVALID_ARG


def test_skydome_abi_errors(ctx):
    lib = ctx.lib
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    eq = torch.zeros((8, 16, 4), dtype=torch.float32, device="cuda")
    col = torch.zeros((4, 8, 4), dtype=torch.float16, device="cuda")
    sp = abi.SkydomeParams()
    p = C.c_void_p
    assert lib.vqhip_skydome(ctx._h, st, p(eq.data_ptr()), 16, 8, C.byref(sp), None, p(col.data_ptr()), 8, 4, 8, abi.FMT_RGBA16F) == 0
    assert lib.vqhip_skydome(ctx._h, st, None, 16, 8, C.byref(sp), None, p(col.data_ptr()), 8, 4, 8, abi.FMT_RGBA16F) == abi.VQHIP_ERR_INVALID_ARG
    assert lib.vqhip_skydome(ctx._h, st, p(eq.data_ptr()), 16, 8, C.byref(sp), None, p(col.data_ptr()), 8, 4, 8, abi.FMT_RGBA8_UNORM) == abi.VQHIP_ERR_UNSUPPORTED
    cov = abi.Interpolants(eq.data_ptr(), eq.data_ptr(), eq.data_ptr(), 9, 4, 9)
    assert lib.vqhip_skydome(ctx._h, st, p(eq.data_ptr()), 16, 8, C.byref(sp), C.byref(cov), p(col.data_ptr()), 8, 4, 8, abi.FMT_RGBA16F) == abi.VQHIP_ERR_INVALID_ARG


def test_gbuffer_producer_abi_errors(ctx):
    lib = ctx.lib
    W, H = 8, 4
    ip = [dev(p) for p in synth.interpolants(W, H, 1)]
    out = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(4)]
    inter = abi.Interpolants(ip[0].data_ptr(), ip[1].data_ptr(), ip[2].data_ptr(), W, H, W)
    gb = abi.GBuffer(out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), W, H, W)
    mats = (abi.MaterialDesc * 1)()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.vqhip_gbuffer_from_materials(ctx._h, st, C.byref(inter), mats, 1, 0.1, None, C.byref(gb)) == 0
    assert lib.vqhip_gbuffer_from_materials(ctx._h, st, None, mats, 1, 0.1, None, C.byref(gb)) == abi.VQHIP_ERR_INVALID_ARG
    assert lib.vqhip_gbuffer_from_materials(ctx._h, st, C.byref(inter), mats, lib.vqhip_max_materials() + 1, 0.1, None, C.byref(gb)) == abi.VQHIP_ERR_INVALID_ARG
    mats[0].texDiffuse = abi.Texture2D(ip[0].data_ptr(), 8, 8, 9, 0)          # 8x8 has 4 levels, not 9
    assert lib.vqhip_gbuffer_from_materials(ctx._h, st, C.byref(inter), mats, 1, 0.1, None, C.byref(gb)) == abi.VQHIP_ERR_INVALID_ARG
    assert b"mip count" in lib.vqhip_last_error(ctx._h)
    bad = abi.GBuffer(out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), W + 1, H, W + 1)
    mats[0].texDiffuse = abi.Texture2D(None, 0, 0, 0, 0)
    assert lib.vqhip_gbuffer_from_materials(ctx._h, st, C.byref(inter), mats, 1, 0.1, None, C.byref(bad)) == abi.VQHIP_ERR_INVALID_ARG
    npot = torch.zeros((12 * 8 * 2, 4), dtype=torch.uint8, device="cuda")
    assert lib.vqhip_mip_chain_box_rgba8(ctx._h, st, C.c_void_p(npot.data_ptr()), 12, 8, 2) == abi.VQHIP_ERR_UNSUPPORTED
    assert lib.vqhip_max_materials() >= 128


def test_gbuffer_producer_full_size_properties(ctx):
    """3840x2160 with 12 materials: size-independent properties (no oracle run at this size) — no-geometry pixels are
    all-zero records, P passes through bit-exactly, |Surface.N| == 1 within fp32, parameters stay in range — plus 16 rows
    checked bit for bit against the oracle."""
    W, H, NM = 3840, 2160, 12
    ip = synth.interpolants(W, H, NM)
    datas, host_chains, hmats, dmats, keep = build_materials(ctx, NM)
    ssao = synth.ssao_image(W, H)
    got = ctx.gbuffer_from_materials([dev(p) for p in ip], dmats, 0.055, dev(ssao))
    g = [t.cpu().numpy() for t in got]
    idx = np.ascontiguousarray(ip[2][..., 3]).view(np.int32)
    geo = (idx >= 0) & (idx < NM)
    for k in range(4):
        assert np.all(g[k][~geo] == 0)
    assert np.array_equal(g[0][geo][:, :3], ip[0][geo][:, :3])
    nlen = np.sqrt((g[1][geo][:, :3].astype(np.float64) ** 2).sum(-1))
    assert np.abs(nlen - 1).max() < 1e-5
    assert (g[1][geo][:, 3] >= 0).all() and (g[1][geo][:, 3] <= 1.0).all()
    assert (g[2][geo] >= 0).all() and (g[2][geo] <= 1.0).all()
    assert (g[0][geo][:, 3] >= 0).all() and (g[0][geo][:, 3] <= 0.055).all()
    # 16 rows against the oracle: rows [y0, y0+16) need their quad partners, so cut on an even row
    y0 = 1000
    ipc = [p[y0:y0 + 16] for p in ip]
    ref = O.gbuffer_from_materials(ipc, hmats, 0.055, None)
    got_c = ctx.gbuffer_from_materials([dev(p) for p in ipc], dmats, 0.055, None)
    for k in range(4):
        n, idx2 = O.bits_equal(got_c[k].cpu().numpy(), ref[k])
        assert n == 0, (k, n, idx2)
