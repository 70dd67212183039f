"""CPU: the SCALING half of the light loop's quotient proof. The GPU proves fdiv_rcp(a, b, RN(1/b)) == RN(a / b) for all 2^23 x 2^23 significand pairs at unit scale
(tests/test_gpu_devmath.py); what the scale of the operands adds is underflow — of the first product q0 = a * r or of the residual a - b q0 (a multiple of ulp(b) ulp(q0)).
vq_shade.h:add_point_light claims neither happens for the numerators its granularity conditions allow: |a| >= 2^-63 for a component of L - P (D <= 2^30), >= 2^-93 for a
component of Wo + Wi (|Hs| <= 2). Checked here in exact rational arithmetic (fractions), with binary32 rounding done by hand: the claimed ranges are exact, and where the residual's quantum
falls below 2^-149 the sequence does go wrong — the conditions are needed, not decoration."""
import random
from fractions import Fraction

import numpy as np


def rn32(x):
    """round a Fraction to the nearest binary32 (ties to even), denormals included; returns a Fraction"""
    if x == 0:
        return Fraction(0)
    s = -1 if x < 0 else 1
    x = abs(x)
    e = x.numerator.bit_length() - x.denominator.bit_length()
    if Fraction(2) ** e > x:
        e -= 1
    e = max(e, -126)                                         # denormal range: fixed quantum 2^-149
    q = Fraction(2) ** (e - 23)
    n = x / q
    f = n.numerator // n.denominator
    r = n - f
    if r > Fraction(1, 2) or (r == Fraction(1, 2) and f % 2 == 1):
        f += 1
    return s * f * q


def fma32(a, b, c):
    return rn32(a * b + c)


def fdiv_rcp(a, b):
    r = rn32(1 / b)                                          # the correctly rounded reciprocal (sqrt_rcp_newton / rcp_newton deliver exactly this)
    q0 = rn32(a * r)
    return fma32(fma32(-b, q0, a), r, q0)


def f32(rng, lo_exp, hi_exp):
    m = rng.getrandbits(23)
    return Fraction((1 << 23) | m, 1 << 23) * Fraction(2) ** rng.randint(lo_exp, hi_exp)


def test_rounding_helper_matches_numpy():
    rng = random.Random(1)
    for _ in range(2000):
        a, b = f32(rng, -60, 60), f32(rng, -60, 60)
        assert float(rn32(a / b)) == float(np.float32(float(a)) / np.float32(float(b)))
    assert rn32(Fraction(3, 2) * Fraction(2) ** -149) == Fraction(2) ** -148        # tie in the denormal range -> even


def test_quotients_of_the_light_loop_are_exact_at_the_allowed_scales():
    rng = random.Random(2)
    for _ in range(3000):
        D = f32(rng, -40, 29)                                # D = |L - P| in [2^-40, 2^30)
        a = f32(rng, -63, -63 + rng.choice([0, 0, 1, 5, 40])) * rng.choice([1, -1])
        if abs(a) <= D * 2:                                  # a component never exceeds the length (rounding aside)
            assert fdiv_rcp(a, D) == rn32(a / D), (a, D)
    for _ in range(3000):
        Hl = f32(rng, -40, 0)                                # |Wo + Wi| in [2^-40, 2]
        a = f32(rng, -93, -93 + rng.choice([0, 0, 1, 10, 60])) * rng.choice([1, -1])
        if abs(a) <= Hl * 2:
            assert fdiv_rcp(a, Hl) == rn32(a / Hl), (a, Hl)
    for _ in range(3000):                                    # vq_devmath.h:normalize_lit: |a| >= 2^-78, |v| = sqrt(dd) in [2^-50, 2^48]
        D = f32(rng, -50, 47)
        a = f32(rng, -78, -78 + rng.choice([0, 0, 2, 30, 100])) * rng.choice([1, -1])
        if abs(a) <= D * 2:
            assert fdiv_rcp(a, D) == rn32(a / D), (a, D)
    assert fdiv_rcp(Fraction(0), f32(rng, -10, 10)) == 0     # a zero component stays zero


def test_the_conditions_are_needed():
    """below the allowed scales the sequence does fail: with a normal quotient but |a| < 2^-103 the residual a - b q0 (a multiple of 2^-46 |a|) is no longer representable,
    the correction works on a rounded residual and the result lands on the wrong neighbour now and then"""
    rng = random.Random(3)
    bad = 0
    for _ in range(6000):
        D = f32(rng, -40, -20)
        a = f32(rng, -125, -106)                             # q0 = a / D in [2^-105, 2^-66]: normal; residual quantum 2^-171 .. 2^-152: below 2^-149
        if abs(a) <= D:
            bad += fdiv_rcp(a, D) != rn32(a / D)
    assert bad > 0
