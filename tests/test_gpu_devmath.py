"""The product's device arithmetic (vq_devmath.h) must agree BIT-FOR-BIT with the oracle's lowering table
(oracle/vqo_math.h) — that is what makes every kernel comparable bit-exactly. Element-wise sweep over random,
structured and special inputs through the test-only probe library tests/probe/libvqprobe.so."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu
PROBE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libvqprobe.so")

SPECIAL = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 2.0, 1e-45, -1e-45, 1.17549435e-38, 1.17549421e-38, 3.4028235e38, -3.4028235e38,
                    np.inf, -np.inf, np.nan, 0.70710677, 0.70710683, 0.99999994, 1.0000001, 127.99999, 128.0, -126.0, -126.00001, -149.0,
                    0.0031308, 1048576.0, 1048577.0, 65504.0, 65519.996, 65520.0, 5.9604645e-8, 2.9802322e-8, 2.9802326e-8, 0.4142135623730950,
                    2.414213562373095, 0.999, 255.0, 0.49999997, 2147483520.0, -2147483648.0, 3e9], np.float32)


@pytest.fixture(scope="module")
def probe(ctx):
    lib = C.CDLL(PROBE)
    lib.vqprobe_math.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]

    def run(fn, a, b=None):
        ta = torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
        tb = torch.from_numpy(np.ascontiguousarray(b, np.float32)).cuda() if b is not None else None
        out = torch.empty_like(ta)
        rc = lib.vqprobe_math(fn, ta.data_ptr(), tb.data_ptr() if tb is not None else None, out.data_ptr(), ta.numel(), None)
        assert rc == 0
        torch.cuda.synchronize()
        return out.cpu().numpy()
    return run


def inputs(kind, n=1 << 21, seed=0):
    r = np.random.default_rng(seed)
    if kind == "bits":       # uniformly random bit patterns: every exponent, denormals, NaNs, infs
        return r.integers(0, 2 ** 32, n, dtype=np.uint32).view(np.float32)
    if kind == "unit":
        return r.uniform(-1.0000001, 1.0000001, n).astype(np.float32)
    if kind == "pos":
        return np.exp(r.uniform(-90, 90, n)).astype(np.float32)
    if kind == "angle":
        return r.uniform(-20, 20, n).astype(np.float32)
    if kind == "exp":
        return r.uniform(-160, 140, n).astype(np.float32)
    raise KeyError(kind)


CASES = [(0, "log2", ["bits", "pos", "unit"]), (1, "exp2", ["bits", "exp", "unit"]), (3, "sin", ["bits", "angle", "unit"]),
         (4, "cos", ["bits", "angle", "unit"]), (5, "tan", ["angle", "unit"]), (6, "asin", ["bits", "unit"]), (7, "acos", ["bits", "unit"]),
         (9, "rcp", ["bits", "pos"]), (10, "sqrt", ["bits", "pos"]), (11, "rsqrt", ["bits", "pos"])]


@pytest.mark.parametrize("fn,name,kinds", CASES, ids=[c[1] for c in CASES])
def test_unary(probe, fn, name, kinds):
    for k in kinds:
        x = np.concatenate([inputs(k, seed=fn), SPECIAL, -SPECIAL])
        n, idx = O.bits_equal(probe(fn, x), O.math_array(fn, x))
        assert n == 0, f"{name}[{k}]: {n} mismatches, e.g. x={x[idx[:, 0]]} gpu={probe(fn, x)[idx[:, 0]]} cpu={O.math_array(fn, x)[idx[:, 0]]}"


def test_binary(probe):
    r = np.random.default_rng(3)
    x = np.concatenate([inputs("unit", seed=1) * 0.5 + 0.5, np.repeat(SPECIAL, len(SPECIAL))])
    y = np.concatenate([np.full(1 << 21, 5.0, np.float32), np.tile(SPECIAL, len(SPECIAL))])
    for (fn, name) in ((2, "pow"), (8, "atan2")):
        n, idx = O.bits_equal(probe(fn, x, y), O.math_array(fn, x, y))
        assert n == 0, f"{name}: {n} mismatches at x={x[idx[:, 0]]} y={y[idx[:, 0]]}"
    a, b = r.normal(size=1 << 20).astype(np.float32), r.normal(size=1 << 20).astype(np.float32)
    n, idx = O.bits_equal(probe(8, a, b), O.math_array(8, a, b))
    assert n == 0, f"atan2 random: {n}"
    m = inputs("pos", 1 << 20, 9); e = r.uniform(0.1, 3.0, 1 << 20).astype(np.float32)
    n, idx = O.bits_equal(probe(2, m, e), O.math_array(2, m, e))
    assert n == 0, f"pow random: {n}"


def test_storage_conversions(probe):
    lib = O.load()
    x = np.concatenate([inputs("bits", seed=7), inputs("pos", seed=8), SPECIAL, -SPECIAL])
    h = np.empty(x.size, np.uint16)
    lib.vqo_f32_to_f16(x.ctypes.data, h.ctypes.data, x.size)
    back = np.empty(x.size, np.float32)
    lib.vqo_f16_to_f32(h.ctypes.data, back.ctypes.data, x.size)
    n, idx = O.bits_equal(probe(12, x), back)
    assert n == 0, f"fp16 round trip: {n} mismatches at {x[idx[:, 0]]}"
    u8 = np.empty(x.size, np.uint8)
    lib.vqo_f32_to_unorm8(x.ctypes.data, u8.ctypes.data, x.size)
    assert np.array_equal(probe(13, x), u8.astype(np.float32))


@pytest.mark.parametrize("which,name", [(0, "rcp == 1.0f/x"), (1, "sqrt_ == IEEE sqrtf"), (2, "saturate == select form"),
                                        (3, "rcp_newton correct whenever its result is normal"),
                                        (4, "sqrt_newton correct on [2^-100, FLT_MAX]"),
                                        (5, "rsqrt_cr == (float)(1.0 / sqrt((double)x)), the DXC reading's correctly rounded Rsqrt"),
                                        (6, "rsqrt_cr_fast (v_rsq_f32 seed + binary64 second-order correction) correct on [2^-100, 2^100]"),
                                        (7, "sqrt_rcp_newton: the reciprocal refined from the root's v_rsq_f32 seed == 1.0f / sqrtf(x) on [2^-100, 2^100]"),
                                        (8, "sqrt_rcp_newton: the root == IEEE sqrtf on [2^-100, 2^100]")])
def test_fast_paths_exhaustive(ctx, which, name):
    """The product's fast exact primitives against the plain IEEE forms for ALL 2^32 float bit patterns."""
    lib = C.CDLL(PROBE)
    lib.vqprobe_exhaustive.restype = C.c_longlong
    lib.vqprobe_exhaustive.argtypes = [C.c_int, C.POINTER(C.c_uint32)]
    first = C.c_uint32(0)
    n = lib.vqprobe_exhaustive(which, C.byref(first))
    assert n == 0, f"{name}: {n} mismatching inputs, first 0x{first.value:08x}"


def test_fdiv_rcp_exhaustive_significands(ctx):
    """fdiv_rcp (q = a*r, one exact-residual correction with the correctly rounded reciprocal r) == IEEE a / b for ALL 2^23 x 2^23
    significand pairs (7.0e13 quotients, ~20 s of GPU time): the quotients of normalize() in the light loop are the reference's."""
    lib = C.CDLL(PROBE)
    lib.vqprobe_fdiv_exhaustive.restype = C.c_longlong
    lib.vqprobe_fdiv_exhaustive.argtypes = [C.c_uint32] * 4 + [C.POINTER(C.c_uint32)]
    first = (C.c_uint32 * 2)()
    n = lib.vqprobe_fdiv_exhaustive(0, 1 << 23, 0, 1 << 23, first)
    assert n == 0, f"{n} mismatching significand pairs, e.g. a=1+{first[0]}*2^-23, b=1+{first[1]}*2^-23"


@pytest.mark.parametrize("dxc", [0, 1])
def test_normalize_fast_form_on_adversarial_vectors(ctx, dxc):
    """vq_devmath.h:normalize_lit — root + reciprocal from one v_rsq_f32, three fdiv_rcp quotients, one integer guard — against the oracle's plain IEEE statement on
    vectors built to sit on every edge of the guard: components at and around 2^-78 and 2^-102 (the underflow limit of the corrected quotient), +-0 (a zero keeps its sign),
    denormals, dot products at and around 2^-100 / 2^100, all-ones significands (the hard case of the reciprocal), inf / NaN, the zero vector; and 2 M random vectors over
    the whole exponent range. Bit for bit (NaN == NaN), both readings."""
    lib = C.CDLL(PROBE)
    lib.vqprobe_normalize.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    r = np.random.default_rng(123 + dxc)
    mags = np.concatenate([np.float32(2.0) ** np.arange(-149, 128, dtype=np.float32), np.array([0.0, np.inf, np.nan], np.float32),
                           np.nextafter(np.float32(2.0) ** np.arange(-110, -70), np.float32(0)), np.nextafter(np.float32(2.0) ** np.arange(-60, 60), np.float32(0))]).astype(np.float32)
    a = mags[r.integers(0, mags.size, (400_000, 3))] * r.choice(np.array([1.0, -1.0], np.float32), (400_000, 3))
    a[::7, 1] = 0.0; a[1::7, 2] = -0.0; a[2::7, 0] = a[2::7, 1]                       # zero components of either sign, equal components
    u = r.integers(0, 2 ** 32, (2_000_000, 3), dtype=np.uint32).view(np.float32)        # arbitrary bit patterns
    g = (r.standard_normal((600_000, 3)) * 10.0 ** r.uniform(-30, 30, (600_000, 1))).astype(np.float32)
    v = np.ascontiguousarray(np.concatenate([a, u, g]), np.float32)
    want = np.empty_like(v)
    O.load().vqo_normalize_lit_array.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    O.load().vqo_set_arithmetic(dxc)
    try:
        with np.errstate(all="ignore"):
            O.load().vqo_normalize_lit_array(v.ctypes.data, want.ctypes.data, v.shape[0])
    finally:
        O.load().vqo_set_arithmetic(0)
    dv = torch.from_numpy(v).cuda()
    out = torch.empty_like(dv)
    assert lib.vqprobe_normalize(dv.data_ptr(), out.data_ptr(), v.shape[0], dxc, None) == 0
    torch.cuda.synchronize()
    n, idx = O.bits_equal(out.cpu().numpy(), want)
    assert n == 0, (n, idx, v[idx[:, 0]][:3])
