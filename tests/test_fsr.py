"""CPU tests of the FSR 1.0 oracle (oracle/vqo_fsr.cpp ⇔ ffx_fsr1.h FsrEasuF / FsrRcasF as dispatched at
SceneRendering.cpp:2695-2784): an independent numpy restatement in IEEE binary32 (numpy rounds every operation to float32
and never contracts, i.e. the contract's arithmetic) that must agree BIT FOR BIT, closed-form properties, constant blocks."""
import numpy as np
import pytest

from tests import oracle_lib as O
from vqengine_amd import abi, capi, synth

F = np.float32


def _u(a):
    return np.ascontiguousarray(a, F).view(np.uint32)


def _f(a):
    return np.ascontiguousarray(a, np.uint32).view(F)


def lo_rcp(a): return _f(np.uint32(0x7ef07ebb) - _u(a))
def lo_rsq(a): return _f(np.uint32(0x5f347d74) - (_u(a) >> np.uint32(1)))
def med_rcp(a):
    b = _f(np.uint32(0x7ef19fff) - _u(a))
    return b * (-b * a + F(2.0))


def sat(x): return np.minimum(np.maximum(x, F(0)), F(1))


def _tex(img, x, y):
    h, w = img.shape[:2]
    return img[np.clip(y, 0, h - 1), np.clip(x, 0, w - 1), :3]


def easu_np(img, out_w, out_h, con):
    """img float32 [h,w,>=3]; returns float32 [out_h,out_w,3]."""
    c = _f(con)
    iy, ix = np.meshgrid(np.arange(out_h), np.arange(out_w), indexing="ij")
    ppx = ix.astype(F) * c[0] + c[2]
    ppy = iy.astype(F) * c[1] + c[3]
    fpx, fpy = np.floor(ppx), np.floor(ppy)
    ppx, ppy = ppx - fpx, ppy - fpy
    fx, fy = fpx.astype(np.int64), fpy.astype(np.int64)
    T = lambda dx, dy: _tex(img, fx + dx, fy + dy)
    b, cc = T(0, -1), T(1, -1)
    e, f, g, h = T(-1, 0), T(0, 0), T(1, 0), T(2, 0)
    i, j, k, l = T(-1, 1), T(0, 1), T(1, 1), T(2, 1)
    n, o = T(0, 2), T(1, 2)
    L = lambda t: t[..., 2] * F(0.5) + (t[..., 0] * F(0.5) + t[..., 1])
    bL, cL, eL, fL, gL, hL, iL, jL, kL, lL, nL, oL = map(L, (b, cc, e, f, g, h, i, j, k, l, n, o))
    dirx = np.zeros_like(ppx); diry = np.zeros_like(ppx); ln = np.zeros_like(ppx)

    def eset(w, lA, lB, lC, lD, lE):
        nonlocal dirx, diry, ln
        dc, cb = lD - lC, lC - lB
        lenX = lo_rcp(np.maximum(np.abs(dc), np.abs(cb)))
        dirX = lD - lB
        dirx = dirx + dirX * w
        lenX = sat(np.abs(dirX) * lenX)
        lenX = lenX * lenX
        ln = ln + lenX * w
        ec, ca = lE - lC, lC - lA
        lenY = lo_rcp(np.maximum(np.abs(ec), np.abs(ca)))
        dirY = lE - lA
        diry = diry + dirY * w
        lenY = sat(np.abs(dirY) * lenY)
        lenY = lenY * lenY
        ln = ln + lenY * w

    one = F(1.0)
    with np.errstate(over="ignore", invalid="ignore"):
        eset((one - ppx) * (one - ppy), bL, eL, fL, gL, jL)
        eset(ppx * (one - ppy), cL, fL, gL, hL, kL)
        eset((one - ppx) * ppy, fL, iL, jL, kL, nL)
        eset(ppx * ppy, gL, jL, kL, lL, oL)
        dirR = dirx * dirx + diry * diry
        zro = dirR < F(1.0 / 32768.0)
        dirR = np.where(zro, one, lo_rsq(dirR))
        dirx = np.where(zro, one, dirx)
        dirx, diry = dirx * dirR, diry * dirR
        ln = ln * F(0.5)
        ln = ln * ln
        stretch = (dirx * dirx + diry * diry) * lo_rcp(np.maximum(np.abs(dirx), np.abs(diry)))
        len2x = one + (stretch - one) * ln
        len2y = one + F(-0.5) * ln
        lob = F(0.5) + F((1.0 / 4.0 - 0.04) - 0.5) * ln
        clp = lo_rcp(lob)
        mn4 = np.minimum(np.minimum(f, np.minimum(g, j)), k)
        mx4 = np.maximum(np.maximum(f, np.maximum(g, j)), k)
        aC = np.zeros(ppx.shape + (3,), F); aW = np.zeros_like(ppx)
        for (ox, oy, t) in ((0, -1, b), (1, -1, cc), (-1, 1, i), (0, 1, j), (0, 0, f), (-1, 0, e), (1, 1, k), (2, 1, l), (2, 0, h), (1, 0, g), (1, 2, o), (0, 2, n)):
            offx, offy = F(ox) - ppx, F(oy) - ppy
            vx = (offx * dirx) + (offy * diry)
            vy = (offx * (-diry)) + (offy * dirx)
            vx, vy = vx * len2x, vy * len2y
            d2 = np.minimum(vx * vx + vy * vy, clp)
            wB = F(2.0 / 5.0) * d2 + F(-1.0)
            wA = lob * d2 + F(-1.0)
            wB, wA = wB * wB, wA * wA
            wB = F(25.0 / 16.0) * wB + F(-(25.0 / 16.0 - 1.0))
            w = wB * wA
            aC = aC + t * w[..., None]
            aW = aW + w
        r = one / aW
        return np.minimum(mx4, np.maximum(mn4, aC * r[..., None]))


def rcas_np(img, con):
    h, w = img.shape[:2]
    pad = np.zeros((h + 2, w + 2, 3), F)
    pad[1:-1, 1:-1] = img[..., :3]
    b, d, e, f, hh = pad[:-2, 1:-1], pad[1:-1, :-2], pad[1:-1, 1:-1], pad[1:-1, 2:], pad[2:, 1:-1]
    mn4 = np.minimum(np.minimum(b, np.minimum(d, f)), hh)
    mx4 = np.maximum(np.maximum(b, np.maximum(d, f)), hh)
    four, one = F(4.0), F(1.0)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        hitMin = mn4 * (one / (four * mx4))
        hitMax = (one - mx4) * (one / (four * mn4 + F(-4.0)))
        lobe3 = np.fmax(-hitMin, hitMax)
        lobe = np.fmax(F(-0.1875), np.fmin(np.fmax(lobe3[..., 0], np.fmax(lobe3[..., 1], lobe3[..., 2])), F(0.0))) * _f(con)[0]
        rcpL = med_rcp(four * lobe + one)
        l3 = lobe[..., None]
        return (l3 * b + l3 * d + l3 * hh + l3 * f + e) * rcpL[..., None]


def _img(w, h, seed=3):
    r = np.random.default_rng(seed)
    base = synth.hdr_image(w, h, seed=seed, scale=1.0)[..., :3]
    img = np.clip(base / (1 + base), 0, 1).astype(F)                       # tonemapped-like [0,1]
    rows = img[h // 3: h // 3 + 4]
    rows[...] = r.random(rows.shape, dtype=F)                              # noise rows
    img[:, w // 2: w // 2 + 3] = F(0.9)                                    # a hard vertical edge
    img[h // 2:, : w // 4] = F(0.0)                                        # black block (mx4 = 0 -> inf/NaN limiters inside RCAS)
    out = np.ones((h, w, 4), F)
    out[..., :3] = img
    return out


@pytest.mark.parametrize("scale", [(1.5, 1.5), (1.3, 1.7), (2.0, 2.0), (1.0, 1.0)])
def test_easu_oracle_bit_exact_with_numpy_restatement(scale):
    w, h = 64, 40
    img = _img(w, h)
    ow, oh = int(w * scale[0]), int(h * scale[1])
    con = O.fsr_easu_con(w, h, ow, oh)
    got = O.fsr_easu(img, abi.FMT_RGBA32F, ow, oh, con=con)
    ref = easu_np(img, ow, oh, con)
    n, idx = O.bits_equal(got[..., :3], ref)
    assert n == 0, (n, idx)
    assert np.all(got[..., 3] == 1.0)


def test_rcas_oracle_bit_exact_with_numpy_restatement():
    img = _img(96, 50, seed=5)
    for stops in (0.0, 0.2, 1.0, 2.0):
        con = O.fsr_rcas_con(stops)
        got = O.fsr_rcas(img, abi.FMT_RGBA32F, con=con)
        ref = rcas_np(img, con)
        n, idx = O.bits_equal(got[..., :3], ref)
        assert n == 0, (stops, n, idx)


def test_constant_blocks():
    """FsrEasuCon / FsrRcasCon (ffx_fsr1.h:156-203, :662-674): product host functions == oracle == closed form."""
    lib = capi.load_library()
    for (iw, ih, ow, oh) in ((1280, 720, 1920, 1080), (2560, 1440, 3840, 2160), (1477, 831, 1920, 1080)):
        con = np.array(list(capi.fsr_easu_con(iw, ih, ow, oh)), np.uint32)
        assert np.array_equal(con, O.fsr_easu_con(iw, ih, ow, oh))
        c = con.view(F)
        assert c[0] == F(iw) * (F(1) / F(ow)) and c[2] == F(0.5) * F(iw) * (F(1) / F(ow)) - F(0.5)
        assert c[4] == F(1) / F(iw) and c[7] == -(F(1) / F(ih)) and c[13] == F(4) * (F(1) / F(ih)) and con[14] == 0 == con[15] and c[12] == 0
    for stops in (0.0, 0.2, 1.0, 2.0):
        con = np.array(list(capi.fsr_rcas_con(stops)), np.uint32)
        assert np.array_equal(con, O.fsr_rcas_con(stops))
        s = con[:1].view(F)[0]
        assert abs(float(s) - 2.0 ** -stops) < 1e-7
        # ffx_a.h:482-550 packs the CPU-side half by TRUNCATION (pinned against the reference's own table in test_ref_pinning.py)
        hb = ((int(s.view(np.uint32)) >> 23) - 112 << 10) + ((int(s.view(np.uint32)) & 0x7fffff) >> 13)
        assert con[1] == (hb | (hb << 16)) and con[2] == 0 == con[3]
    assert capi.fsr_rcas_con(0.2)[1] == 0x3af63af6
    assert lib is not None


def test_easu_properties():
    """Constant image -> the same constant (dering clamp makes it exact); output inside the local min/max of the 2x2 footprint."""
    const = np.empty((20, 30, 4), F); const[...] = (0.25, 0.5, 0.75, 1.0)
    out = O.fsr_easu(const, abi.FMT_RGBA32F, 45, 30)
    assert np.all(out == np.array([0.25, 0.5, 0.75, 1.0], F))
    img = _img(48, 32, seed=9)
    out = O.fsr_easu(img, abi.FMT_RGBA32F, 96, 64)
    assert np.isfinite(out).all() and out[..., :3].min() >= img[..., :3].min() and out[..., :3].max() <= img[..., :3].max()
    # storage formats: RGBA8 in -> RGBA8 out, RGBA16F -> RGBA16F run and stay in range
    img8 = (img * 255 + 0.5).astype(np.uint8)
    o8 = O.fsr_easu(img8, abi.FMT_RGBA8_UNORM, 72, 48)
    assert o8.dtype == np.uint8 and np.all(o8[..., 3] == 255)
    o16 = O.fsr_rcas(img.astype(np.float16), abi.FMT_RGBA16F)
    assert o16.dtype == np.float16 and np.all(o16[..., 3] == 1.0)


def test_visualization_modes_closed_form():
    """Visualization.hlsl:34-120 — every draw mode against its closed form (numpy float32, same operation order)."""
    r = np.random.default_rng(8)
    img = r.random((9, 13, 4), dtype=F)
    img[0, 0] = (1.0, 0.0, 0.5, 0.25)
    V = lambda m, u=0, s=1.0: O.visualize(img, abi.FMT_RGBA32F, abi.VizParams(m, u, s))
    lib = O.load(); lib.vqo_pow.restype = O.C.c_float; lib.vqo_pow.argtypes = [O.C.c_float, O.C.c_float]
    d = V(1)
    exp = np.array([[lib.vqo_pow(float(v), 500.0) for v in row] for row in img[..., 0]], F)
    assert np.array_equal(d[..., 0], exp) and np.array_equal(d[..., 1], exp) and np.array_equal(d[..., 3], img[..., 3])
    np.testing.assert_allclose(d[1:, :, 0], img[1:, :, 0].astype(np.float64) ** 500, rtol=2e-3, atol=1e-37)     # 500*log2 amplifies the log error
    assert d[0, 0, 0] == 1.0
    assert np.array_equal(V(2, 0)[..., :3], (img[..., :3] - F(0.5)) * F(2) * F(0) + F(1) * img[..., :3])
    assert np.array_equal(V(2, 1)[..., :3], (img[..., :3] - F(0.5)) * F(2) * F(1) + F(0) * img[..., :3])
    for m in (3, 4):
        assert np.array_equal(V(m)[..., :3], np.repeat(img[..., 3:4], 3, -1))
    assert np.array_equal(V(5)[..., :3], np.repeat(img[..., 0:1], 3, -1))
    for m in (6, 7):
        assert np.array_equal(V(m)[..., :3], img[..., :3])
    mv = V(8, 0, 3.5)
    assert np.array_equal(mv[..., 0], (img[..., 0] * F(0.5)) * F(3.5) + F(0.5)) and np.array_equal(mv[..., 1], (img[..., 1] * F(-0.5)) * F(3.5) + F(0.5))
    assert np.all(mv[..., 2] == 0.5)
    for m in (0, 9, -1):
        assert np.all(V(m)[..., :3] == np.array([1, 0, 1], F))


def test_rcas_sharpens_an_edge_and_keeps_flat_regions():
    img = np.ones((16, 32, 4), F); img[..., :3] = 0.25; img[:, 16:, :3] = 0.75
    out = O.fsr_rcas(img, abi.FMT_RGBA32F, con=O.fsr_rcas_con(0.0))
    flat = out[4:12, 4:12, :3]
    assert np.abs(flat - 0.25).max() < 2e-3                                 # APrxMedRcp error only
    assert out[8, 15, 0] < 0.25 and out[8, 16, 0] > 0.75                     # undershoot / overshoot across the edge


def test_visualization_reads_packed_normals_and_two_channel_targets():
    """vqo_visualize on the formats the draw modes' SRVs have (SceneRendering.cpp:2555-2566): R10G10B10A2_UNORM decodes c / 1023 (alpha / 3), RG16F / RG32F read
    (r, g, 0, 1) — each == the RGBA32F path on the decoded values, and the decode == the SSR fallback's (one correctly rounded quotient)."""
    r = np.random.default_rng(9)
    q = r.integers(0, 2 ** 32, (7, 11), dtype=np.uint32)
    dec = np.stack([(q & 1023), (q >> 10) & 1023, (q >> 20) & 1023], -1).astype(np.float64) / 1023.0
    dec4 = np.concatenate([dec, ((q >> 30).astype(np.float64) / 3.0)[..., None]], -1).astype(F)          # float64 quotient rounded once == the IEEE float quotient here
    for p in (abi.VizParams(2, 0, 1.0), abi.VizParams(2, 1, 1.0), abi.VizParams(3, 0, 1.0)):
        assert np.array_equal(O.visualize(q, abi.FMT_R10G10B10A2_UNORM, p, abi.FMT_RGBA32F), O.visualize(dec4, abi.FMT_RGBA32F, p, abi.FMT_RGBA32F))
    mv = (r.random((7, 11, 2), dtype=F) - F(0.5)) * F(0.1)
    full = np.concatenate([mv, np.zeros((7, 11, 1), F), np.ones((7, 11, 1), F)], -1)
    p = abi.VizParams(8, 0, 25.0)
    assert np.array_equal(O.visualize(mv, abi.FMT_RG32F, p, abi.FMT_RGBA32F), O.visualize(full, abi.FMT_RGBA32F, p, abi.FMT_RGBA32F))
    h = mv.astype(np.float16)
    fullh = np.concatenate([h.astype(F), np.zeros((7, 11, 1), F), np.ones((7, 11, 1), F)], -1)
    assert np.array_equal(O.visualize(h, abi.FMT_RG16F, p), O.visualize(fullh, abi.FMT_RGBA32F, p, abi.FMT_RGBA16F))
