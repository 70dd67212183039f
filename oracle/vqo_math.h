// vqo_math.h — scalar float32 "lowering table" of the HLSL intrinsics used by the hot path.
//
// ORACLE / TEST INFRASTRUCTURE ONLY. Nothing under vqengine_amd/ may include, link or call this.
// PARITY: the ALGORITHM built from these intrinsics is pinned against the reference's own HLSL run on the CPU (oracle/_ref,
// see vqo_oracle.cpp's header); the BITS of the intrinsics are not: the reference (vilbeyli/VQEngine) has no numeric tests,
// golden images or known-answer vectors for this path and D3D12/WARP/DXC cannot run here, so what
// DXC's HL->DXIL lowering + a driver's JIT make of them is not observable. This file fixes ONE legal D3D-precision lowering per intrinsic; the HIP kernels
// implement the same lowering independently (vqengine_amd/csrc/vq_devmath.h) and must match these
// bits exactly.
//
// Lowering rules (each cites the HLSL construct it stands for):
//   a + b, a - b, a * b      IEEE-754 binary32, round-to-nearest-even, NO implicit contraction into FMA
//                            (compile with -ffp-contract=off). Where an expression tree uses a fused multiply-add
//                            ("mad", as D3D compilers/drivers emit for a*b+c) it is WRITTEN as fma_() — see the
//                            lighting functions in vqo_oracle.cpp and DESIGN.md §3 "expression trees".
//   a / b                    a * rcp(b), rcp(b) = correctly rounded 1/b.  (What D3D drivers emit for
//                            HLSL '/'; within the D3D 1-ULP 'div' allowance per factor.)
//   sqrt(x)                  correctly rounded.
//   rsqrt(x)                 rcp(sqrt(x)).
//   dot(a,b)                 left-to-right FMA chain: fma(az,bz, fma(ay,by, ax*bx)) (DXIL dot2/3/4 are
//                            opaque intrinsics; GPUs evaluate them as mul + mad chains). mul(vec,matrix)
//                            and mul(matrix,vec) are dots of the same form.
//   normalize(v)             v * rsqrt(dot(v,v))          (DXC lowers normalize this way)
//   length(v)                sqrt(dot(v,v))
//   lerp(a,b,t)              fma(t, b-a, a)               (a + t*(b-a) as one mad)
//   reflect(i,n)             i - (2*dot(n,i))*n           written as i - n*(2*dot(n,i)) per component
//   cross(a,b)               (ay*bz - az*by, az*bx - ax*bz, ax*by - ay*bx), no FMA
//   saturate(x)              min(max(x,0),1) with NaN -> 0
//   max/min                  a NaN operand is dropped (maxNum / minNum, DXIL FMax / FMin); of two zeros min returns -0, max returns +0 whatever their order —
//                            IEEE 754-2019 minimum / maximum, what v_min_f32 / v_max_f32 do (libm's fminf / fmaxf return the FIRST operand on that tie:
//                            rounds 1-5 called them, tests/fuzz/fuzz_wide.py found FSR frames where the order shows)
//   pow(x, 2)                x*x          (DXC HLOperationLower: only the literal exponent 2 becomes a mul
//                                          outside FXC-compat mode — recalled from DXC sources, unverifiable here)
//   pow(x, y)                exp2(y * log2(x))   => pow(0,y>0) = 0, pow(neg,y) = NaN
//   exp2/log2/sin/cos/tan/asin/acos/atan2   the explicit polynomial algorithms below (classic
//                            Cephes-style single-precision kernels, evaluated with explicit FMA Horner
//                            steps; accuracy vs libm double is measured in tests/test_oracle_math.py).
//   (int)x                   truncation toward zero; NaN -> 0; saturating.
//
// Contract v5 (round 2): the table above is the FAST-MATH lowering; it is now used only where the result is insensitive to it. Everything
// ForwardLighting.hlsl:PSMain evaluates once per pixel, and inside the light loops the chain that feeds the GGX denominator and the range
// cull (Lw - P, its length, Wi, H, dot(N,H), nh2*(a2-1)+1), is evaluated AS WRITTEN: products and sums rounded one by one, left to right,
// a/b = the IEEE quotient (fdiv_), dot / length / normalize / lerp / reflect / mul(v,M) in their textbook expansion (the *_lit functions
// below) — the evaluation the reference's own HLSL gets when it is run through oracle/ref_src/hlsl_shim.h. Reason: at roughness < 0.2 a
// 1-ulp change of dot(N,H) moves a highlight pixel by tens of RGBA16F ulps (measured on the BASELINE-shape fixtures: up to 62), so only
// the same rounding sequence keeps the product within 1 storage ulp of the reference source everywhere.
#ifndef VQO_MATH_H
#define VQO_MATH_H

#include <cmath>
#include <cstdint>
#include <cstring>

namespace vqo {

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float    u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

static inline float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
static inline float rcp(float b) { return 1.0f / b; }
static inline float div_(float a, float b) { return a * rcp(b); }
static inline float fdiv_(float a, float b) { return a / b; }          // IEEE-754 correctly rounded quotient (load-time passes, see below)
static inline float sqrt_(float x) { return __builtin_sqrtf(x); }
static inline float rsqrt(float x) { return rcp(sqrt_(x)); }
static inline float max_(float a, float b) { if (a != a) return b; if (b != b) return a; if (a == b) return __builtin_signbit(a) ? b : a; return a > b ? a : b; }
static inline float min_(float a, float b) { if (a != a) return b; if (b != b) return a; if (a == b) return __builtin_signbit(a) ? a : b; return a < b ? a : b; }
static inline float saturate(float x) { return (x > 0.0f) ? ((x < 1.0f) ? x : 1.0f) : 0.0f; }   // NaN -> 0
static inline float abs_(float x) { return __builtin_fabsf(x); }

static inline f3 add(f3 a, f3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
static inline f3 sub(f3 a, f3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static inline f3 mul(f3 a, f3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
static inline f3 mul(f3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
static inline f3 neg(f3 a) { return { -a.x, -a.y, -a.z }; }
static inline float dot(f3 a, f3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
static inline float dot4(f4 a, f4 b) { return fma_(a.w, b.w, fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x))); }
static inline f3 normalize(f3 v) { return mul(v, rsqrt(dot(v, v))); }
// The HLSL AS WRITTEN, for the ill-conditioned chain of the lighting functions (contract v5) — in one of the TWO READINGS the HLSL
// leaves open (vqo_set_arithmetic / vqhip_set_arithmetic; oracle/ref_src/hlsl_shim.h builds the reference's sources in both):
//   literal (default): dot = x*x' + y*y' + z*z' left to right with every product and sum rounded on its own, normalize(v) = v / length(v)
//                      with one IEEE division per component;
//   dxc              : what DXC's HLOperationLower emits — DXIL Dot3 as the FMA chain fma(az,bz, fma(ay,by, ax*bx)), normalize(v) =
//                      v * Rsqrt(dot(v,v)) with a CORRECTLY ROUNDED rsqrt, length = sqrt of that dot, reflect through that dot. lerp, mul(v, M)
//                      and the quotients stay as written in both. The DXC reading's pow = exp2(y * log2 x) is the separate switch vqo_set_fresnel_pow.
inline int g_arith_dxc = 0;
static inline float rsqrt_cr(float x) { return (float)(1.0 / __builtin_sqrt((double)x)); }   // RN(x^-1/2): the double result is far inside half a binary32 ulp of the true value except for a
                                                                                              // vanishing set of inputs, none of which exists (tests/test_oracle_math.py checks all 2^24 x 2 significands exactly)
static inline float dot_lit(f3 a, f3 b) { return g_arith_dxc ? dot(a, b) : (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float dot4_lit(f4 a, f4 b) { return g_arith_dxc ? dot4(a, b) : ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }
static inline f3 div_lit(f3 v, float l) { return { fdiv_(v.x, l), fdiv_(v.y, l), fdiv_(v.z, l) }; }
static inline float length_lit(f3 v) { return sqrt_(dot_lit(v, v)); }
static inline f3 normalize_lit(f3 v) { return g_arith_dxc ? mul(v, rsqrt_cr(dot(v, v))) : div_lit(v, length_lit(v)); }
static inline float lerp_lit(float a, float b, float t) { return a + t * (b - a); }
static inline f3 reflect_lit(f3 i, f3 n) { const float t = 2.0f * dot_lit(n, i); return { i.x - t * n.x, i.y - t * n.y, i.z - t * n.z }; }
static inline float length(f3 v) { return sqrt_(dot(v, v)); }
static inline f3 cross(f3 a, f3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
static inline float lerp(float a, float b, float t) { return fma_(t, b - a, a); }
static inline f3 reflect(f3 i, f3 n) { float t = 2.0f * dot(n, i); return { i.x - n.x * t, i.y - n.y * t, i.z - n.z * t }; }

// float -> int, truncation toward zero, NaN -> 0, saturating (HLSL (int)x; also used for texel addressing)
static inline int f2i_trunc(float x) {
    if (!(x == x)) return 0;
    if (x >=  2147483520.0f) return  2147483520;
    if (x <= -2147483520.0f) return -2147483520;
    return (int)x;
}
static inline int f2i_floor(float x) { return f2i_trunc(__builtin_floorf(x)); }

// ---------------------------------------------------------------------------------------------
// log2(x): x = m * 2^e with m in [sqrt(1/2), sqrt(2)) obtained by integer arithmetic on the bit pattern
//   (u' = u - bits(sqrt(1/2)); e = u' >> 23 (arithmetic); m = bits(u - (u' & 0xff800000))), f = m - 1,
//   log2(1+f) = f * Q(f), Q = degree-8 near-minimax fit of log2(1+f)/f on [sqrt(1/2)-1, sqrt(2)-1]
//   (max error 3.8e-8; fitted by tests-independent least-squares/Lawson iteration, coefficients below are exact
//   binary32 values), log2(x) = fma(f, Q(f), e).  <= 2.1 ULP vs float64 (tests/test_oracle_math.py); because the
//   mantissa is centred on 1 there is no cancellation near x = 1.  x<0 -> NaN, x==0 -> -inf, denormals are scaled.
// ---------------------------------------------------------------------------------------------
static inline float log2_(float x) {
    if (!(x == x)) return x;
    if (x < 0.0f) return u2f(0x7fc00000u);
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    int e = 0;
    uint32_t u = f2u(x);
    if (u < 0x00800000u) { x = x * 8388608.0f; u = f2u(x); e = -23; }     // denormal
    const uint32_t up = u - 0x3f3504f3u;                                   // bits(0.70710677f)
    e += (int32_t)up >> 23;
    const float m = u2f(u - (up & 0xff800000u));
    const float f = m - 1.0f;
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

// exp2(x): n = rint(x) (round half to even), f = x - n in [-0.5, 0.5], 2^f ~ 1 + f*P(f) (Cephes exp2f), scale by 2^n.
// x >= 128 -> +inf; x < -126 -> 0 (denormal results flush to zero like D3D); NaN -> NaN.
static inline float exp2_(float x) {
    if (!(x == x)) return x;
    if (x >= 128.0f) return INFINITY;
    if (x < -126.0f) return 0.0f;
    float n = __builtin_rintf(x);                 // round-half-even (v_rndne_f32): f = x - n in [-0.5, 0.5]
    float f = x - n;
    float p = 1.535336188319500E-4f;
    p = fma_(p, f, 1.339887440266574E-3f);
    p = fma_(p, f, 9.618437357674640E-3f);
    p = fma_(p, f, 5.550332471162809E-2f);
    p = fma_(p, f, 2.402264791363012E-1f);
    p = fma_(p, f, 6.931472028550421E-1f);
    float r = fma_(p, f, 1.0f);
    int ni = (int)n;                              // in [-126, 128]
    if (ni > 127) { r = r * 2.0f; ni = 127; }     // f rounding can give n = 128 for x just below 128
    return r * u2f((uint32_t)(ni + 127) << 23);
}

static inline float pow_(float x, float y) { return exp2_(y * log2_(x)); }
// pow(1 - cos, 5.0) of the three Fresnel terms (BRDF.hlsl:135, :155, :274) is evaluated as the product x * ((x*x) * (x*x)): FXC's
// "mul-only pattern" for a literal integral exponent (square-and-multiply, up to 3 multiplies for a scalar), which DXC reproduces only
// in FXC-compatibility mode (lib/HLSL/HLOperationLower.cpp: TranslatePowImpl, isFXCCompatMode; otherwise only exponent 2 becomes a
// multiply). The engine's own flags do NOT enable that mode, its binary evaluates exp2(5*log2 x) — contract v4 takes the product
// deliberately: more accurate on [0,1], inside D3D's pow tolerance, 23 fewer instructions per light, and finite
// where rounding pushes the base a hair below zero (dot(H,V) = 1 + ulp), where the log/exp form is NaN. v1-v3 used exp2(5*log2 x).
static inline float pow5_(float x) { const float x2 = x * x; return x * (x2 * x2); }

// sin/cos: Cody-Waite reduction by pi/4 octants (3-part constant), Cephes sinf/cosf kernels on
// [-pi/4, pi/4]. |x| > 2^20 or non-finite -> NaN.
static inline void sincos_(float x, float* s, float* c) {
    float ax = abs_(x);
    if (!(ax <= 1048576.0f)) { *s = *c = u2f(0x7fc00000u); return; }
    int j = (int)(ax * 1.27323954473516f);        // 4/pi
    j = (j + 1) & ~1;
    float y = (float)j;
    float r = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
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
    float ss, cc;
    switch (q) {
        case 0:  ss =  sn; cc =  cs; break;
        case 1:  ss =  cs; cc = -sn; break;
        case 2:  ss = -sn; cc = -cs; break;
        default: ss = -cs; cc =  sn; break;
    }
    *s = (x < 0.0f) ? -ss : ss;
    *c = cc;
}
static inline float sin_(float x) { float s, c; sincos_(x, &s, &c); return s; }
static inline float cos_(float x) { float s, c; sincos_(x, &s, &c); return c; }
static inline float tan_(float x) { float s, c; sincos_(x, &s, &c); return div_(s, c); }

// asin on [-1,1] (Cephes asinf): |x| > 0.5 uses pi/2 - 2*asin(sqrt((1-|x|)/2)). |x| > 1 -> NaN.
static inline float asin_poly(float x, float z) {   // x + x*z*P(z)
    float p = 4.2163199048E-2f;
    p = fma_(p, z, 2.4181311049E-2f);
    p = fma_(p, z, 4.5470025998E-2f);
    p = fma_(p, z, 7.4953002686E-2f);
    p = fma_(p, z, 1.6666752422E-1f);
    return fma_(p * z, x, x);
}
static inline float asin_(float x) {
    float a = abs_(x);
    if (!(a <= 1.0f)) return u2f(0x7fc00000u);
    float r;
    if (a > 0.5f) {
        float z = 0.5f * (1.0f - a);
        float s = sqrt_(z);
        float t = asin_poly(s, z);
        r = 1.5707963267948966192f - (t + t);
    } else {
        r = asin_poly(a, a * a);
    }
    return (x < 0.0f) ? -r : r;
}
static inline float acos_(float x) {
    if (!(abs_(x) <= 1.0f)) return u2f(0x7fc00000u);
    if (x < -0.5f) { float z = 0.5f * (1.0f + x); float s = sqrt_(z); float t = asin_poly(s, z); return 3.14159265358979323846f - (t + t); }
    if (x >  0.5f) { float z = 0.5f * (1.0f - x); float s = sqrt_(z); float t = asin_poly(s, z); return t + t; }
    return 1.5707963267948966192f - asin_poly(x, x * x);
}

// atan (Cephes atanf): reduce with tan(3pi/8), tan(pi/8); odd polynomial in x.
static inline float atan_(float xx) {
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
// atan2(y,x) with HLSL/C quadrant conventions; atan2(0,0) = 0.
static inline float atan2_(float y, float x) {
    if (!(x == x) || !(y == y)) return u2f(0x7fc00000u);
    const float PI_F = 3.14159265358979323846f, PIO2_F = 1.5707963267948966192f;
    if (x == 0.0f) { if (y > 0.0f) return PIO2_F; if (y < 0.0f) return -PIO2_F; return 0.0f; }
    if (y == 0.0f) return (x < 0.0f) ? PI_F : 0.0f;
    float w = 0.0f;
    if (x < 0.0f) w = (y < 0.0f) ? -PI_F : PI_F;
    return w + atan_(div_(y, x));
}

// ---------------------------------------------------------------------------------------------
// storage conversions
// ---------------------------------------------------------------------------------------------
// fp32 -> fp16 round-to-nearest-even, denormals kept, overflow -> inf, NaN -> quiet NaN.
static inline uint16_t f32_to_f16(float f) {
    uint32_t u = f2u(f);
    uint32_t sign = (u >> 16) & 0x8000u;
    uint32_t a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return (uint16_t)(sign | (a > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);            // >= 65520 rounds to inf
    if (a < 0x33000001u) return (uint16_t)sign;                          // <= 2^-25 rounds to 0
    int e = (int)(a >> 23) - 127;
    uint32_t m = (a & 0x007fffffu) | 0x00800000u;
    int shift = (e < -14) ? (13 + (-14 - e)) : 13;                       // bits to drop
    uint32_t h = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (h & 1u))) h += 1;
    uint32_t out;
    if (e < -14) out = h;                                                // denormal (may carry into normal)
    else         out = ((uint32_t)(e + 15) << 10) + (h - 0x400u);        // h has the implicit bit at 0x400
    return (uint16_t)(sign | out);
}
static inline float f16_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0) return u2f(sign);
        float v = (float)m * 5.9604644775390625e-8f;                     // m * 2^-24
        return (sign ? -v : v);
    }
    if (e == 31) return u2f(sign | 0x7f800000u | (m << 13));
    return u2f(sign | ((e + 112u) << 23) | (m << 13));
}
// fp32 -> UNORM8: NaN -> 0, clamp to [0,1], scale by 255, add 0.5, truncate (D3D FLOAT -> UNORM rule).
static inline uint8_t f32_to_unorm8(float f) {
    float c = saturate(f);
    return (uint8_t)(int)(c * 255.0f + 0.5f);
}

} // namespace vqo
#endif
