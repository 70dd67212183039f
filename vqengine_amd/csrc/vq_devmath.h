// vq_devmath.h — device-side float32 arithmetic of the vqhip kernels (gfx950).
//
// The kernels are bit-exact against the CPU oracle, which is only possible if every HLSL intrinsic is
// lowered to operations that both a CPU and CDNA4 evaluate identically: IEEE add/sub/mul/fma,
// correctly rounded 1/x and sqrt, and explicit polynomial kernels for the transcendentals (the
// hardware v_exp/v_log/v_sin/v_rcp/v_rsq approximations are NOT used on value paths: their bits
// are not reproducible off-chip). The lowering table is specified in DESIGN.md §"Arithmetic
// contract"; this file is the product's own implementation of it (it does not include or link
// anything from oracle/). Build flags that matter: -ffp-contract=off (no implicit FMA) and
// -fhip-fp32-correctly-rounded-divide-sqrt (the hipcc default).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vqd {

#define VQD __device__ __forceinline__
#define VQHD __host__ __device__ __forceinline__     // also callable from the C-ABI translation unit (same IEEE operations on the host)

VQHD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// Correctly rounded reciprocal RN(1/b). Fast path: v_rcp_f32 (1 ulp) + one Newton step, i.e. Markstein's
// r' = fma(fma(-b,r,1), r, r); checked EXHAUSTIVELY on gfx950 against IEEE 1.0f/b over all 2^32 inputs
// (scripts/ubench/valu_ubench.hip, tests/test_gpu_devmath.py::test_rcp_exhaustive): whenever the result is a
// normal number it is the correctly rounded quotient. Zero / denormal / inf / NaN results (denormal or huge
// inputs, 0, inf, NaN) take the IEEE division expansion.
VQD float rcp_newton(float b) {                              // == RN(1/b) whenever the RESULT is a normal number
    const float r = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
VQD bool is_normal(float r) { return __builtin_amdgcn_classf(r, 0x108); }      // 0x108 = {-normal, +normal}
VQD float rcp(float b) {
    float r = rcp_newton(b);
    if (__builtin_expect(!is_normal(r), 0)) r = 1.0f / b;
    return r;
}
// Correctly rounded square root. Fast path: v_rsq_f32 seed, s = x*y, one exact-residual correction
// s' = fma(fma(-s,s,x), y/2, s); EXHAUSTIVELY equal to IEEE sqrtf for every x in [2^-100, FLT_MAX] on gfx950
// (below 2^-100 the residual goes denormal). Everything else (0, tiny, inf, NaN, negative) takes the IEEE expansion.
VQD float sqrt_newton(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    const float s = x * y, h = 0.5f * y;
    const float r = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(r, h, s);
}
VQD bool sqrt_fast_ok(float x) { return (x >= 0x1p-100f) & (x <= 3.4028234663852886e38f); }
// D = RN(sqrt(x)) AND r = RN(1/D) from ONE quarter-rate instruction: the reciprocal is refined from the v_rsq_f32 seed the root already fetched — one Markstein step on
// y against D whose residual carries a bias just above 2^-48: a Newton step always underestimates (by y e^2), and for the one significand family where that lands exactly on a
// rounding midpoint (D = 2^n (2 - 2^-23): 1/D = 2^-n-1 (1 + 2^-24 + 2^-48 + ...)) the bias turns e = 2^-24 into 2^-24 + 2^-47 and the tie into the right neighbour.
// EXHAUSTIVELY equal to (sqrt_newton(x), rcp_newton(sqrt_newton(x))) — i.e. to the correctly rounded pair — for every x in [2^-100, 2^100] on gfx950
// (tests/test_gpu_devmath.py; scripts/ubench/rcp_from_rsq.hip scans the alternatives: without the bias 200 inputs differ, with 1.5 x 2^-48 another 100). 3 full-rate VALU
// replace v_rcp_f32 (4 issue slots) + 2 fma.
VQD float sqrt_rcp_newton(float x, float* rD) {
    const float y = __builtin_amdgcn_rsqf(x);
    const float s = x * y, h = 0.5f * y;
    const float D = __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
    const float e = __builtin_fmaf(-D, y, 1.0f) + 0x1.000002p-48f;
    *rD = __builtin_fmaf(e, y, y);
    return D;
}
VQD bool sqrt_rcp_fast_ok(float x) { return (x >= 0x1p-100f) & (x <= 0x1p100f); }
VQD float sqrt_(float x) {
    float s = sqrt_newton(x);
    if (__builtin_expect(!sqrt_fast_ok(x), 0)) s = __builtin_sqrtf(x);
    return s;
}
// RN(x^-1/2) as the oracle and the shim define it, (float)(1.0 / sqrt((double)x)). Fast path in BINARY32: v_rsq_f32 seed y (1 ulp), the residual e = 1 - x y^2 to ~2^-47
// (x*y = t + d exactly: d = fma(x, y, -t) recovers the product's rounding error) and the second-order correction y + y (e/2 + 3/8 e^2) in exactly this grouping.
// EXHAUSTIVELY equal to the definition for every x in [2^-100, 2^100] on gfx950 (tests/probe/devmath_probe.hip `rsqrt_cr`, tests/test_gpu_devmath.py); the scan of the
// alternatives (scripts/ubench/rsqrt_fp32.hip, profiles/r4r_rsqrt_fp32.md): a first-order step mis-rounds 100 inputs whatever its bias, other groupings of the second-order
// term 100, dropping d 218 million. 8 full-rate VALU; rounds 3-4 used a binary64 tail (~15 issue slots). Outside the domain (0, denormal, inf, NaN, negative): the definition itself.
VQD float rsqrt_cr_fast(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    const float t = x * y, d = __builtin_fmaf(x, y, -t);
    const float e = __builtin_fmaf(-d, y, __builtin_fmaf(-t, y, 1.0f));
    return __builtin_fmaf(__builtin_fmaf(0.375f * e, e, 0.5f * e), y, y);
}
VQD bool rsqrt_cr_fast_ok(float x) { return (x >= 0x1p-100f) & (x <= 0x1p100f); }
VQD float rsqrt_cr(float x) {
    float r = rsqrt_cr_fast(x);
    if (__builtin_expect(!rsqrt_cr_fast_ok(x), 0)) r = (float)(1.0 / __builtin_sqrt((double)x));
    return r;
}
// IEEE-754 correctly rounded quotient (v_div_scale/v_div_fmas/v_div_fixup under -fhip-fp32-correctly-rounded-divide-sqrt). Used where
// the reference's x/x must be EXACTLY 1 (ImportanceSampleGGX at roughness 0, BRDF.hlsl:222): a*rcp(b) gives 1 - 2^-24 there.
VQD float fdiv_(float a, float b) { return a / b; }
// The same quotient from an already available correctly rounded reciprocal r = RN(1/b): q = a*r, one exact-residual correction
// q' = fma(fma(-b,q,a), r, q) (Markstein). 3 VALU per quotient instead of the ~10 of the IEEE expansion when several numerators share
// one divisor (normalize: 3 components). Equal to a/b for ALL 2^23 x 2^23 significand pairs — checked exhaustively on gfx950
// (tests/probe/devmath_probe.hip `fdiv`, tests/test_gpu_devmath.py::test_fdiv_rcp_exhaustive_significands) — whenever nothing underflows:
// callers guarantee |a| in [2^-78, 2^100], b and r normal (the residual a - b*q is a multiple of ulp(b)*ulp(q) ~ 2^-47 |a|).
// NOT valid for a = -0 (gives +0), denormals, inf/NaN: hot loops test their operands once and redo the light with fdiv_ otherwise.
VQD float fdiv_rcp(float a, float b, float r) {
    const float q = a * r;
    return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
}
// Arithmetic policies for hot loops: Fast runs reciprocals / square roots unchecked and accumulates ONE validity
// flag; when it drops (operand outside the validated fast range — rare) the caller redoes the whole block with
// IEEE. Results are bit-identical to rcp() / sqrt_() either way.
VQD bool fdiv_rcp_ok(float a) { const float m = __builtin_fabsf(a); return (m >= 0x1p-78f) & (m <= 0x1p100f); }
struct RcpFast {
    static constexpr bool kGgxDenomAboveEps = false;         // policies that make no claim about the GGX denominator keep its EPSILON test
    bool ok = true;
    VQD float operator()(float b) { const float r = rcp_newton(b); ok = ok & is_normal(r); return r; }
    VQD float sqrt(float x) { ok = ok & sqrt_fast_ok(x); return sqrt_newton(x); }
    VQD float sqrt_rcp(float x, float* r) { ok = ok & sqrt_rcp_fast_ok(x); return sqrt_rcp_newton(x, r); }   // D = sqrt(x) and *r = 1/D, both correctly rounded
    VQD float div(float a, float b, float r) { ok = ok & fdiv_rcp_ok(a); return fdiv_rcp(a, b, r); }       // r = (*this)(b)
    VQD float rsqrt(float x) { ok = ok & rsqrt_cr_fast_ok(x); return rsqrt_cr_fast(x); }                  // DXC reading only
};
struct RcpIEEE {
    static constexpr bool kGgxDenomAboveEps = false;
    VQD float operator()(float b) const { return 1.0f / b; }
    VQD float sqrt(float x) const { return __builtin_sqrtf(x); }
    VQD float sqrt_rcp(float x, float* r) const { const float D = __builtin_sqrtf(x); *r = 1.0f / D; return D; }
    VQD float div(float a, float b, float) const { return a / b; }
    VQD float rsqrt(float x) const { return (float)(1.0 / __builtin_sqrt((double)x)); }
};
VQD float rsqrt(float x) {                                   // rcp(sqrt(x)), both correctly rounded: from one v_rsq_f32 inside sqrt_rcp_newton's proven domain
    float r;
    (void)sqrt_rcp_newton(x, &r);
    if (__builtin_expect(!sqrt_rcp_fast_ok(x), 0)) r = rcp(sqrt_(x));
    return r;
}
VQD float div_(float a, float b) { return a * rcp(b); }     // HLSL a / b
VQD float max_(float a, float b) { return __builtin_fmaxf(a, b); }
VQD float min_(float a, float b) { return __builtin_fminf(a, b); }
// saturate: clamp to [0,1], NaN -> 0, -0 -> +0. v_med3_f32(x,0,1) equals the select form (x>0 ? (x<1 ? x : 1) : 0) for ALL
// 2^32 inputs on gfx950 (exhaustive: scripts/ubench/sqrt_ubench.hip, tests/test_gpu_devmath.py).
VQD float saturate(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }
VQHD float abs_(float x) { return __builtin_fabsf(x); }
VQHD float qnan() { return __builtin_bit_cast(float, 0x7fc00000u); }

struct f3 { float x, y, z; };
VQD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
VQD f3 add(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
VQD f3 sub(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
VQD f3 mul(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
VQD f3 mul(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
VQD f3 neg(f3 a) { return mk3(-a.x, -a.y, -a.z); }
VQD float dot(f3 a, f3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
VQD f3 normalize(f3 v) { return mul(v, rsqrt(dot(v, v))); }
VQD float length(f3 v) { return sqrt_(dot(v, v)); }
VQD f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
VQD float lerp(float a, float b, float t) { return fma_(t, b - a, a); }       // a + t*(b-a) as one mad
VQD f3 reflect(f3 i, f3 n) { float t = 2.0f * dot(n, i); return mk3(i.x - n.x * t, i.y - n.y * t, i.z - n.z * t); }
// The HLSL AS WRITTEN (arithmetic contract v5): every product and sum rounded on its own, left to right, a/b the IEEE
// quotient. Used for everything ForwardLighting.hlsl:PSMain evaluates once per pixel and, inside the light loops, for the chain that
// feeds the GGX denominator and the range cull — where one ulp of an operand is worth tens of RGBA16F ulps of a highlight pixel.
VQD float dot_lit(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
VQD float length_lit(f3 v) { return sqrt_(dot_lit(v, v)); }
VQD f3 div_lit(f3 v, float l) { return mk3(fdiv_(v.x, l), fdiv_(v.y, l), fdiv_(v.z, l)); }
// normalize(v) AS WRITTEN = one IEEE quotient per component by length(v). Fast form: the root and its reciprocal from one v_rsq_f32 (sqrt_rcp_newton), the three
// quotients by fdiv_rcp (3 VALU each) — equal to the IEEE operations whenever dot(v,v) lies in [2^-100, 2^96] and no component is a non-zero number below 2^-78
// (exhaustive proofs above); a zero component keeps its sign (0 * r would lose a -0). One test per vector (integer tricks: (bits << 1) - 1 maps +-0 to 0xffffffff and
// orders the magnitudes); anything else — NaN, inf, denormal components, a zero vector — takes the plain IEEE operations. ~37 VALU instead of ~58.
VQD f3 normalize_lit(f3 v) {
    const float dd = dot_lit(v, v);
    float r;
    const float D = sqrt_rcp_newton(dd, &r);
    const uint32_t bx = __float_as_uint(v.x), by = __float_as_uint(v.y), bz = __float_as_uint(v.z);
    const uint32_t m = min((bx << 1) - 1u, min((by << 1) - 1u, (bz << 1) - 1u));                  // smallest magnitude, zeros excluded
    const bool ok = (m >= ((0x18800000u << 1) - 1u)) & ((__float_as_uint(dd) - 0x0d800000u) <= (0x6f800000u - 0x0d800000u));   // 2^-78; dd in [2^-100, 2^96] (NaN / negative: out):
                                                                                                                                  // every quotient >= 2^-78 / 2^48 = 2^-126 is normal
    f3 q;
    q.x = __uint_as_float(__float_as_uint(fdiv_rcp(v.x, D, r)) | (bx & 0x80000000u));
    q.y = __uint_as_float(__float_as_uint(fdiv_rcp(v.y, D, r)) | (by & 0x80000000u));
    q.z = __uint_as_float(__float_as_uint(fdiv_rcp(v.z, D, r)) | (bz & 0x80000000u));
    if (__builtin_expect(!ok, 0)) q = div_lit(v, sqrt_(dd));
    return q;
}
VQD float lerp_lit(float a, float b, float t) { return a + t * (b - a); }
VQD f3 reflect_lit(f3 i, f3 n) { const float t = 2.0f * dot_lit(n, i); return mk3(i.x - t * n.x, i.y - t * n.y, i.z - t * n.z); }

// ---- the SECOND READING of dot / normalize / length / reflect (vqhip_set_arithmetic): what DXC's HLOperationLower emits — DXIL Dot3 as the
// FMA chain (dot() above), normalize(v) = v * Rsqrt(dot(v, v)) with a CORRECTLY ROUNDED rsqrt, length = sqrt of that dot. The CPU checker restates it (oracle/, test infrastructure);
// the reference's own HLSL in this reading: oracle/ref_src/hlsl_shim.h VQ_SHIM_DXC.
// AR = 0: the literal reading (the *_lit functions above), AR = 1: the DXC reading. lerp, mul(v, M) and the quotients are the same in both.
template <int AR> VQD float dot_r(f3 a, f3 b) { return AR ? dot(a, b) : dot_lit(a, b); }
template <int AR> VQD float length_r(f3 v) { return sqrt_(dot_r<AR>(v, v)); }
template <int AR> VQD f3 normalize_r(f3 v) { if (AR) return mul(v, rsqrt_cr(dot(v, v))); return normalize_lit(v); }
template <int AR> VQD f3 reflect_r(f3 i, f3 n) { const float t = 2.0f * dot_r<AR>(n, i); return mk3(i.x - t * n.x, i.y - t * n.y, i.z - t * n.z); }
// the same choice at run time (wave-uniform flag) for per-pixel code outside the light loops (gbuffer.hip, ssr.hip)
VQD float dot_rt(f3 a, f3 b, bool dxc) { return dxc ? dot(a, b) : dot_lit(a, b); }
VQD float length_rt(f3 v, bool dxc) { return sqrt_(dot_rt(v, v, dxc)); }
VQD f3 normalize_rt(f3 v, bool dxc) { return dxc ? mul(v, rsqrt_cr(dot(v, v))) : normalize_lit(v); }
VQD f3 reflect_rt(f3 i, f3 n, bool dxc) { const float t = 2.0f * dot_rt(n, i, dxc); return mk3(i.x - t * n.x, i.y - t * n.y, i.z - t * n.z); }

// float -> int: truncation, NaN -> 0, saturating
// Branch-free (round 6; the if-ladder compiled to three nested exec-mask branches per conversion): the median clamps to the two largest binary32 integers that fit
// (a NaN comes out as the minimum and is replaced below), the conversion of the clamped value is exact truncation. Same result for every float.
VQD int f2i_trunc(float x) {
    const int r = (int)__builtin_amdgcn_fmed3f(x, -2147483520.0f, 2147483520.0f);
    return (x == x) ? r : 0;
}
VQD int f2i_floor(float x) { return f2i_trunc(__builtin_floorf(x)); }

// log2 of a positive NORMAL number given its bit pattern: mantissa centred on 1 by integer arithmetic
// (m in [sqrt(1/2), sqrt(2))), log2(x) = fma(f, Q(f), e) with the degree-8 polynomial of the arithmetic contract.
VQD float log2_normal_bits(uint32_t u, int ebias) {
    const uint32_t up = u - 0x3f3504f3u;                        // bits(0.70710677f)
    const int e = ebias + ((int32_t)up >> 23);
    const float f = __uint_as_float(u - (up & 0xff800000u)) - 1.0f;
    float q = 0x1.08baeap-3f;
    q = fma_(q, f, -0x1.abe534p-3f);
    q = fma_(q, f,  0x1.b8c15cp-3f);
    q = fma_(q, f, -0x1.e8ced8p-3f);
    q = fma_(q, f,  0x1.26d980p-2f);
    q = fma_(q, f, -0x1.715f9ap-2f);
    q = fma_(q, f,  0x1.ec73bep-2f);
    q = fma_(q, f, -0x1.71546cp-1f);
    q = fma_(q, f,  0x1.715476p+0f);
    return fma_(f, q, (float)e);
}
VQD float log2_(float x) {
    uint32_t u = __float_as_uint(x);
    int e = 0;
    if (u < 0x00800000u) { u = __float_as_uint(x * 8388608.0f); e = -23; }           // +denormal (and +0, handled below)
    float r = log2_normal_bits(u, e);
    // special cases, same precedence as the reference restatement: NaN, negative, zero, +inf
    if (x == __builtin_inff()) r = x;
    if (x == 0.0f) r = -__builtin_inff();
    if (x < 0.0f) r = qnan();
    if (!(x == x)) r = x;
    return r;
}

// exp2: n = nearest integer, 2^f ~ 1 + f*P(f) on [-0.5,0.5]; >= 128 -> inf, < -126 -> 0
VQD float exp2_(float x) {
    float n = __builtin_rintf(x);                          // v_rndne_f32: round half to even
    float f = x - n;
    float p = 1.535336188319500E-4f;
    p = fma_(p, f, 1.339887440266574E-3f);
    p = fma_(p, f, 9.618437357674640E-3f);
    p = fma_(p, f, 5.550332471162809E-2f);
    p = fma_(p, f, 2.402264791363012E-1f);
    p = fma_(p, f, 6.931472028550421E-1f);
    float r = fma_(p, f, 1.0f);
    int ni = (int)max_(min_(n, 128.0f), -127.0f);       // n is an integer-valued float in [-126,128] for in-range x
    if (ni > 127) { r = r * 2.0f; ni = 127; }
    if (ni < -126) ni = -126;
    float out = r * __uint_as_float((uint32_t)(ni + 127) << 23);
    if (x >= 128.0f) out = __builtin_inff();
    if (x < -126.0f) out = 0.0f;
    if (!(x == x)) out = x;
    return out;
}
VQD float pow_(float x, float y) { return exp2_(y * log2_(x)); }
// pow(1 - cos, 5.0) of the three Fresnel terms as FXC's mul-only pattern, square-and-multiply: x * ((x*x) * (x*x)) (contract v4 — a
// deliberate choice: the engine's DXC flags give exp2(5*log2 x)). Three full-rate multiplies instead of a log2 and an
// exp2 polynomial, and no NaN for a base that rounding pushed a hair below zero.
VQD float pow5(float x) { const float x2 = x * x; return x * (x2 * x2); }
// The engine-compile form (vqhip_set_fresnel_pow(EXP2_LOG2)): pow_(x, 5.0f) == exp2_(5 * log2_(x)), where x = 1 - max(0, cos) is either
// 0, negative by a rounding hair, or in [2^-24, 1]. On [2^-24, 1] none of log2_/exp2_'s special cases can fire (5*log2(x) >= -120), so the
// same arithmetic runs without their selects; anything else (rare) takes the general routine. Bit-identical to pow_(x, 5.0f) for every x.
VQD float pow5_explog(float x) {
    if (__builtin_expect(!(x >= 5.9604644775390625e-8f && x <= 1.0f), 0)) return pow_(x, 5.0f);
    const float t = 5.0f * log2_normal_bits(__float_as_uint(x), 0);
    const float n = __builtin_rintf(t);
    const float g = t - n;
    float q = 1.535336188319500E-4f;
    q = fma_(q, g, 1.339887440266574E-3f);
    q = fma_(q, g, 9.618437357674640E-3f);
    q = fma_(q, g, 5.550332471162809E-2f);
    q = fma_(q, g, 2.402264791363012E-1f);
    q = fma_(q, g, 6.931472028550421E-1f);
    const float s = fma_(q, g, 1.0f);
    return s * __uint_as_float((uint32_t)((int)n + 127) << 23);
}
// pow_(x, y) for x in [0,1] known to be +0 or a positive NORMAL number and y > 0 with y*log2(x) >= -126 (e.g. UNORM8 data,
// y = 2.2): the same operations as pow_ with the special-case selects that cannot trigger removed — identical bits.
VQD float pow_unit(float x, float y) {
    const float t = y * log2_normal_bits(__float_as_uint(x), 0);
    const float n = __builtin_rintf(t), f = t - n;
    float p = 1.535336188319500E-4f;
    p = fma_(p, f, 1.339887440266574E-3f);
    p = fma_(p, f, 9.618437357674640E-3f);
    p = fma_(p, f, 5.550332471162809E-2f);
    p = fma_(p, f, 2.402264791363012E-1f);
    p = fma_(p, f, 6.931472028550421E-1f);
    const float r = fma_(p, f, 1.0f) * __uint_as_float((uint32_t)((int)n + 127) << 23);
    return x == 0.0f ? 0.0f : r;
}

// sin/cos: octant reduction with a 3-part pi/4, Cephes kernels; |x| > 2^20 or non-finite -> NaN
VQHD void sincos_(float x, float* s, float* c) {
    float ax = abs_(x);
    bool bad = !(ax <= 1048576.0f);
    float axc = bad ? 0.0f : ax;
    int j = (int)(axc * 1.27323954473516f);
    j = (j + 1) & ~1;
    float y = (float)j;
    float r = ((axc - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    float z = r * r;
    float ps = -1.9515295891E-4f;
    ps = fma_(ps, z, 8.3321608736E-3f);
    ps = fma_(ps, z, -1.6666654611E-1f);
    float sn = fma_(ps * z, r, r);
    float pc = 2.443315711809948E-5f;
    pc = fma_(pc, z, -1.388731625493765E-3f);
    pc = fma_(pc, z, 4.166664568298827E-2f);
    float cs = fma_(pc * z, z, fma_(-0.5f, z, 1.0f));
    int q = (j >> 1) & 3;
    float ss = (q & 1) ? cs : sn;
    float cc = (q & 1) ? sn : cs;
    if (q == 2 || q == 3) ss = -ss;
    if (q == 1 || q == 2) cc = -cc;
    if (x < 0.0f) ss = -ss;
    *s = bad ? qnan() : ss;
    *c = bad ? qnan() : cc;
}
VQD float tan_(float x) { float s, c; sincos_(x, &s, &c); return div_(s, c); }

VQD float asin_poly(float x, float z) {
    float p = 4.2163199048E-2f;
    p = fma_(p, z, 2.4181311049E-2f);
    p = fma_(p, z, 4.5470025998E-2f);
    p = fma_(p, z, 7.4953002686E-2f);
    p = fma_(p, z, 1.6666752422E-1f);
    return fma_(p * z, x, x);
}
VQD float asin_(float x) {
    float a = abs_(x);
    float r;
    if (a > 0.5f) {
        float z = 0.5f * (1.0f - a);
        float s = sqrt_(z);
        float t = asin_poly(s, z);
        r = 1.5707963267948966192f - (t + t);
    } else {
        r = asin_poly(a, a * a);
    }
    r = (x < 0.0f) ? -r : r;
    return (a <= 1.0f) ? r : qnan();
}
VQD float acos_(float x) {
    float r;
    if (x < -0.5f)     { float z = 0.5f * (1.0f + x); float s = sqrt_(z); float t = asin_poly(s, z); r = 3.14159265358979323846f - (t + t); }
    else if (x > 0.5f) { float z = 0.5f * (1.0f - x); float s = sqrt_(z); float t = asin_poly(s, z); r = t + t; }
    else               { r = 1.5707963267948966192f - asin_poly(x, x * x); }
    return (abs_(x) <= 1.0f) ? r : qnan();
}
VQD float atan_(float xx) {
    float x = abs_(xx);
    float y;
    if (x > 2.414213562373095f)       { y = 1.5707963267948966192f; x = -rcp(x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483096f; x = div_(x - 1.0f, x + 1.0f); }
    else                              { y = 0.0f; }
    float z = x * x;
    float p = 8.05374449538e-2f;
    p = fma_(p, z, -1.38776856032E-1f);
    p = fma_(p, z,  1.99777106478E-1f);
    p = fma_(p, z, -3.33329491539E-1f);
    y = y + fma_(p * z, x, x);
    return (xx < 0.0f) ? -y : y;
}
VQD float atan2_(float y, float x) {
    const float PI_F = 3.14159265358979323846f, PIO2_F = 1.5707963267948966192f;
    if (!(x == x) || !(y == y)) return qnan();
    if (x == 0.0f) { if (y > 0.0f) return PIO2_F; if (y < 0.0f) return -PIO2_F; return 0.0f; }
    if (y == 0.0f) return (x < 0.0f) ? PI_F : 0.0f;
    float w = 0.0f;
    if (x < 0.0f) w = (y < 0.0f) ? -PI_F : PI_F;
    return w + atan_(div_(y, x));
}

// ---- storage -----------------------------------------------------------------------------------
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

VQD float4 load_rgba16f(const void* base, size_t idx) {
    h4 v = ((const h4*)base)[idx];
    return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
}
// fp32 -> fp16, round-to-nearest-even of the ALREADY ROUNDED fp32 value. The empty asm pins the fp32 value in a
// VGPR: without it LLVM folds `(half)(a*b)` into v_fma_mixlo_f16, which rounds the exact product once to fp16 and
// differs from RNE16(RNE32(a*b)) in rare double-rounding cases (seen: 1 texel of 32736 in the specular prefilter).
VQD _Float16 to_f16(float f) { asm volatile("" : "+v"(f)); return (_Float16)f; }
VQD void store_rgba16f(void* base, size_t idx, float4 c) {
    h4 v; v.x = to_f16(c.x); v.y = to_f16(c.y); v.z = to_f16(c.z); v.w = to_f16(c.w);
    ((h4*)base)[idx] = v;
}
VQD uint32_t unorm8(float f) { return (uint32_t)(int)(saturate(f) * 255.0f + 0.5f); }
VQD void store_rgba8(void* base, size_t idx, float4 c) {
    ((uint32_t*)base)[idx] = unorm8(c.x) | (unorm8(c.y) << 8) | (unorm8(c.z) << 16) | (unorm8(c.w) << 24);
}
template <int FMT> VQD float4 load_px(const void* base, size_t idx) {
    if (FMT == 0) return ((const float4*)base)[idx];
    return load_rgba16f(base, idx);
}
template <int FMT> VQD void store_px(void* base, size_t idx, float4 c) {
    if (FMT == 0) ((float4*)base)[idx] = c;
    else if (FMT == 1) store_rgba16f(base, idx, c);
    else if (FMT == 2) store_rgba8(base, idx, c);
    else if (FMT == 3) { h2 v; v.x = to_f16(c.x); v.y = to_f16(c.y); ((h2*)base)[idx] = v; }
    else ((float2*)base)[idx] = make_float2(c.x, c.y);
}

} // namespace vqd
