"""CPU tests of the oracle's lowering table (oracle/vqo_math.h): accuracy of every transcendental against numpy
float64, IEEE behaviour of the basic ops, special cases, and the storage conversions against numpy's own."""
import numpy as np
import pytest

from tests import oracle_lib as O

N = 400_000


def ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.maximum(np.spacing(np.abs(ref32)).astype(np.float64), 1e-45)
    return np.abs(got.astype(np.float64) - ref64) / ulp


@pytest.fixture(scope="module")
def rng():
    return np.random.default_rng(1234)


def test_basic_ops_are_ieee(rng):
    x = np.exp(rng.uniform(-80, 80, N)).astype(np.float32)
    assert np.array_equal(O.math_array(9, x), (np.float32(1.0) / x))                   # rcp: correctly rounded 1/x
    assert np.array_equal(O.math_array(10, x), np.sqrt(x))                              # sqrt: correctly rounded
    assert np.array_equal(O.math_array(11, x), np.float32(1.0) / np.sqrt(x))            # rsqrt = rcp(sqrt(x))


def test_log2_exp2_pow(rng):
    x = np.exp(rng.uniform(-85, 85, N)).astype(np.float32)
    assert ulp_err(O.math_array(0, x), np.log2(x.astype(np.float64))).max() < 2.2
    x = rng.uniform(0.5, 2.0, N).astype(np.float32)                                      # near 1: no cancellation (mantissa centred on 1)
    assert np.abs(O.math_array(0, x).astype(np.float64) - np.log2(x.astype(np.float64))).max() < 1.2e-7
    assert ulp_err(O.math_array(0, x), np.log2(x.astype(np.float64))).max() < 2.2
    x = rng.uniform(-126, 128, N).astype(np.float32)
    assert ulp_err(O.math_array(1, x), np.exp2(x.astype(np.float64))).max() < 1.6
    x = rng.uniform(0, 1, N).astype(np.float32)
    p = O.math_array(2, x, np.full_like(x, 5.0))
    rel = np.abs(p.astype(np.float64) - x.astype(np.float64) ** 5) / np.maximum(x.astype(np.float64) ** 5, 1e-30)
    assert rel[x > 1e-3].max() < 4e-6                                                    # Fresnel pow(1-c, 5)
    p = O.math_array(2, x, np.full_like(x, np.float32(1 / 2.4)))
    assert np.abs(p - x.astype(np.float64) ** (1 / 2.4)).max() < 3e-7                    # sRGB OETF pow


def test_log2_exp2_special_cases():
    x = np.array([0.0, -0.0, -1.0, np.inf, np.nan, 1.0, 2.0, 1e-45, 1.17549435e-38], np.float32)
    r = O.math_array(0, x)
    assert r[0] == -np.inf and r[1] == -np.inf and np.isnan(r[2]) and r[3] == np.inf and np.isnan(r[4])
    assert r[5] == 0.0 and r[6] == 1.0 and abs(r[7] + 149) < 1e-5 and abs(r[8] + 126) < 1e-5
    x = np.array([128.0, 127.99999, -126.0, -126.0001, -np.inf, np.inf, np.nan, 0.0, 1.0, -1.0, 10.0], np.float32)
    r = O.math_array(1, x)
    assert r[0] == np.inf and np.isfinite(r[1]) and r[2] == np.float32(2.0 ** -126) and r[3] == 0 and r[4] == 0 and r[5] == np.inf
    assert np.isnan(r[6]) and r[7] == 1 and r[8] == 2 and r[9] == 0.5 and r[10] == 1024
    # HLSL pow semantics from exp2(y*log2 x): pow(0, 5) = 0, pow(negative, 5) = NaN, pow(1, y) = 1
    p = O.math_array(2, np.array([0.0, -0.5, 1.0, 2.0], np.float32), np.array([5, 5, 7.5, 10], np.float32))
    assert p[0] == 0 and np.isnan(p[1]) and p[2] == 1 and p[3] == 1024


def test_trig(rng):
    x = rng.uniform(-50, 50, N).astype(np.float32)
    assert np.abs(O.math_array(3, x) - np.sin(x.astype(np.float64))).max() < 1.5e-7
    assert np.abs(O.math_array(4, x) - np.cos(x.astype(np.float64))).max() < 1.5e-7
    x = rng.uniform(-1.5, 1.5, N).astype(np.float32)
    assert ulp_err(O.math_array(5, x), np.tan(x.astype(np.float64))).max() < 4.5
    x = rng.uniform(-1, 1, N).astype(np.float32)
    assert ulp_err(O.math_array(6, x), np.arcsin(x.astype(np.float64))).max() < 3.0
    assert ulp_err(O.math_array(7, x), np.arccos(x.astype(np.float64))).max() < 2.0
    y, xx = rng.normal(size=N).astype(np.float32), rng.normal(size=N).astype(np.float32)
    assert np.abs(O.math_array(8, y, xx) - np.arctan2(y.astype(np.float64), xx.astype(np.float64))).max() < 4e-7
    # quadrants / axes of atan2 and the domain of asin/acos/sincos
    a = O.math_array(8, np.array([0, 0, 1, -1, 0, 1, -1], np.float32), np.array([1, -1, 0, 0, 0, -1, -1], np.float32))
    assert a[0] == 0 and a[1] == np.float32(np.pi) and a[2] == np.float32(np.pi / 2) and a[3] == -np.float32(np.pi / 2) and a[4] == 0
    assert abs(a[5] - 3 * np.pi / 4) < 1e-6 and abs(a[6] + 3 * np.pi / 4) < 1e-6
    assert np.isnan(O.math_array(6, np.array([1.0000001, -2, np.nan], np.float32))).all()
    assert np.isnan(O.math_array(7, np.array([1.0000001, -2, np.nan], np.float32))).all()
    assert np.isnan(O.math_array(3, np.array([np.inf, 2e6, np.nan], np.float32))).all()
    assert O.math_array(7, np.array([1.0, -1.0, 0.0], np.float32)).tolist() == [0.0, np.float32(np.pi), np.float32(np.pi / 2)]


def test_fp16_conversion_matches_numpy(rng):
    lib = O.load()
    v = np.concatenate([(rng.normal(size=N) * 10 ** rng.uniform(-9, 6, N)), rng.integers(0, 2 ** 32, N, dtype=np.uint32).view(np.float32).astype(np.float64),
                        [0, -0.0, 65504, 65519.99, 65520, 1e-8, 5.96e-8, 2.98e-8, 2.9802322e-8, 2.9802326e-8, np.inf, -np.inf, 6.1e-5, 6.0e-5]]).astype(np.float32)
    out = np.empty(v.size, np.uint16)
    lib.vqo_f32_to_f16(v.ctypes.data, out.ctypes.data, v.size)
    with np.errstate(over="ignore", invalid="ignore"):
        ref = v.astype(np.float16).view(np.uint16)
    ok = (out == ref) | np.isnan(v)
    assert ok.all(), v[~ok][:5]
    assert (np.isnan(out.view(np.float16)) == np.isnan(v)).all()
    h = np.arange(65536, dtype=np.uint16)
    f = np.empty(65536, np.float32)
    lib.vqo_f16_to_f32(h.ctypes.data, f.ctypes.data, 65536)
    ref = h.view(np.float16).astype(np.float32)
    assert ((f.view(np.uint32) == ref.view(np.uint32)) | np.isnan(ref)).all()


def test_unorm8_conversion():
    lib = O.load()
    v = np.array([-1, 0, 0.5 / 255, 0.49999 / 255, 1.49999 / 255, 1.5001 / 255, 0.5, 1, 2, np.nan, np.inf, -np.inf, 254.5 / 255, 254.4999 / 255], np.float32)
    out = np.empty(v.size, np.uint8)
    lib.vqo_f32_to_unorm8(v.ctypes.data, out.ctypes.data, v.size)
    assert out.tolist() == [0, 0, 1, 0, 1, 2, 128, 255, 255, 0, 255, 0, 255, 254]
