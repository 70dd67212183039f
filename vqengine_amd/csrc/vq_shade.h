// vq_shade.h — the per-pixel body of the forward-PBR lighting pass as a device function: ForwardLighting.hlsl:PSMain :289-380 evaluated for
// one G-buffer record. Shared by the two kernels that run it — k_forward_lighting (shade.hip: the G-buffer planes in)
// and k_forward_from_materials (gbuffer.hip: the G-buffer producer and this body in one kernel, the record never leaves registers).
// One lane per pixel, light records read through the scalar cache (wave-uniform index), every light-invariant term hoisted to per-pixel setup.
//
// Arithmetic follows the contract in DESIGN.md §3 (intrinsic lowering: vq_devmath.h). Contract v5: everything evaluated once per
// pixel, and per light the chain into the GGX denominator and the range cull (Lw - P, its length, Wi, H, dot(N,H), nh2*(a2-1)+1), is
// the HLSL AS WRITTEN (products / sums rounded one by one, IEEE quotients: the *_lit functions) — one ulp there is tens of RGBA16F
// ulps of a highlight pixel. The insensitive rest of the light loop keeps the regrouped trees of contract v2-v4 (scalar factors of
// vector products gathered, a*b+c as one mad, the three divisions of D*G/denom merged into one reciprocal, 1/(D*D) = (1/D)^2):
// Every function here takes the READING of dot / normalize / length / reflect as the template parameter AR (vq_devmath.h: 0 = literal, the default;
// 1 = DXC: FMA-chain dot, normalize = v * correctly rounded rsqrt(dot) — vqhip_set_arithmetic); AR = 0 compiles to exactly the round-3 code.
//   Shaders/BRDF.hlsl:65-79,82-97,118-121,132-136,152-161,163-207
//   Shaders/Lighting.hlsl:29-32,57-73,110-174,177-272,308-395
//   Shaders/ForwardLighting.hlsl:284-380
#pragma once
#include "vq_internal.h"
#include "vq_devmath.h"
#include "vq_sampling.h"

namespace {
using namespace vqd;


constexpr float PI_      = 3.14159265359f;    // ShadingMath.hlsl:25
constexpr float EPSILON_ = 0.000000000001f;   // BRDF.hlsl:21

#ifndef VQ_SHADE_WAVES
#define VQ_SHADE_WAVES 1
#endif

// Per-pixel state: BRDF_Surface (BRDF.hlsl:50-58) + everything in BRDF() that does not depend on the light.
struct Pixel {
    f3 P, V, Wo, Nraw, Nn, albedo, F0, omF0, kA;
    float roughness, metalness, omm;      // omm = 1 - metalness
    float NdotV4;                         // 4 * saturate(dot(N, Wo))
    float G1V;                            // Geometry_Smiths_SchlickGGX(N, Wo, roughness)
    float k, omk;                         // k = (roughness+1)^2/8, omk = 1-k
    float a2, a2m1;                       // GGX alpha^2, alpha^2 - 1
    float a2G1V;                          // a2 * G1V: the light-independent part of the merged D*G numerator
    bool p5ExpLog;                        // wave-uniform: Fresnel pow as exp2(5*log2 x) instead of the product (vqhip_set_fresnel_pow). Kept a RUN-TIME
                                          // flag on purpose: with the mode as a template parameter (no branch in the light loop) the same arithmetic ran
                                          // 9 % slower on the same box (profiles/r2c_shade_variants.md) — the scheduler's choice for the longer block
    bool fastOK;                          // roughness in [0,1] and a finite Wo: precondition of the unchecked fast reciprocals (add_point_light)
    bool skipOK;                          // F0 / kA below 2^40 (a finite BRDF whatever the light): this lane may take part in the skips of lights that add b * 0
};

VQD f3 ld3(const VQ_float3& v) { return mk3(v.x, v.y, v.z); }
VQD float min3abs(f3 v) { return __builtin_fminf(__builtin_fminf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), __builtin_fabsf(v.z)); }   // one v_min3_f32 with |.| modifiers

// (bits << 1) - 1: the magnitude of a float as an unsigned integer with +-0 mapped to 0xffffffff (a zero is always an acceptable operand below)
VQD uint32_t mag_or_top(float v) { return (__float_as_uint(v) << 1) - 1u; }

template <int AR>
VQD void setup_pixel(Pixel& px, float4 g0, float4 g1, float4 g2, f3 cam, int negZeroAxes = 0) {
    px.P = mk3(g0.x, g0.y, g0.z);
    px.Nraw = mk3(g1.x, g1.y, g1.z);
    px.roughness = g1.w;
    px.albedo = mk3(g2.x, g2.y, g2.z);
    px.metalness = g2.w;
    // once per pixel: as written (contract v5)
    px.V = normalize_r<AR>(sub(cam, px.P));                    // ForwardLighting.hlsl:285
    px.Wo = normalize_r<AR>(px.V);                             // BRDF.hlsl:166
    px.Nn = normalize_r<AR>(px.Nraw);                          // :167
    px.F0 = mk3(lerp_lit(0.04f, px.albedo.x, px.metalness), lerp_lit(0.04f, px.albedo.y, px.metalness), lerp_lit(0.04f, px.albedo.z, px.metalness));   // :178
    px.omF0 = mk3(1.0f - px.F0.x, 1.0f - px.F0.y, 1.0f - px.F0.z);
    px.omm = 1.0f - px.metalness;
    const float invPI = rcp(PI_);
    px.kA = mk3((px.omm * px.albedo.x) * invPI, (px.omm * px.albedo.y) * invPI, (px.omm * px.albedo.z) * invPI);     // (1-metal)*albedo/PI
    const float NdotV = saturate(dot_r<AR>(px.Nn, px.Wo));   // :171
    px.NdotV4 = 4.0f * NdotV;
    const float rp1 = px.roughness + 1.0f;                   // Geometry_Smiths_SchlickGGX :92-96
    px.k = (rp1 * rp1) * 0.125f;                             // / 8.0f: exact either way
    px.omk = 1.0f - px.k;
    const float NV = max_(0.0f, dot_r<AR>(px.Nn, px.Wo));
    px.G1V = fdiv_(NV, (NV * px.omk + px.k) + 0.0001f);
    const float a = px.roughness * px.roughness;             // NormalDistributionGGX :74-75
    px.a2 = a * a;
    px.a2m1 = px.a2 - 1.0f;
    px.a2G1V = px.a2 * px.G1V;
    // preconditions of the unchecked fast path of add_point_light that depend on the pixel only: roughness in [0, 1] (below 0.04 the GGX
    // EPSILON early-out may fire: such a wave takes the loop form that keeps the early-out as a select, RcpTrustEps) and a finite Wo (with a
    // finite Wi it makes Wo + Wi free of NaN, which the min3 test there cannot see)
    px.fastOK = (px.roughness >= 0.0f) & (px.roughness <= 1.0f) & (dot_r<AR>(px.Wo, px.Wo) <= 4.0f);       // the comparison is false for a NaN component
    // ... and the GRANULARITY conditions that make every quotient of the light loop free of underflow without a test per light (add_point_light): each coordinate of P
    // is 0 or has magnitude in [2^-40, 2^40] (the host checks the same for every light position: pointFastOK), each component of Wo is 0 or >= 2^-40; where some light
    // has a -0.0 coordinate, P must not hold +0.0 on that axis ((-0) - (+0) is the one difference that yields -0, which the fast quotient would turn into +0)
    {
        const uint32_t lo = (0x2b800000u << 1) - 1u;                                                           // 2^-40
        const uint32_t pm = min(mag_or_top(px.P.x), min(mag_or_top(px.P.y), mag_or_top(px.P.z)));
        const uint32_t wm = min(mag_or_top(px.Wo.x), min(mag_or_top(px.Wo.y), mag_or_top(px.Wo.z)));
        const float pmax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(px.P.x), __builtin_fabsf(px.P.y)), __builtin_fabsf(px.P.z));
        bool ok = (pm >= lo) & (wm >= lo) & (pmax <= 0x1p40f);                                                 // false for a NaN / inf coordinate (fmax drops a NaN: caught by the others below)
        ok &= (px.P.x == px.P.x) & (px.P.y == px.P.y) & (px.P.z == px.P.z);
        if (negZeroAxes) {                                                                                     // wave-uniform, almost never taken
            const int z = (__float_as_uint(px.P.x) == 0u ? 1 : 0) | (__float_as_uint(px.P.y) == 0u ? 2 : 0) | (__float_as_uint(px.P.z) == 0u ? 4 : 0);
            ok &= (z & negZeroAxes) == 0;
        }
        px.fastOK &= ok;
    }
    // the skip of lights that add b * (cb * +0) (add_point_light<.., true>, spot_light) needs a FINITE BRDF whatever the light. b = fma(F, sG - kA, kA) with
    // |F| <= 2 |F0| + 1 and sG = D G / denom <= 1e12 * 4 / 1e-4 < 2^56 for roughness in [0, 1] (D <= 1 / EPSILON, each G1 <= 2, denom >= 1e-4): with
    // |F0| and |kA| below 2^40 the product stays under 2^98. Finite F0 and kA alone are NOT enough — an albedo of 1e25 keeps both finite and overflows F * kA
    // (found by tests/fuzz/fuzz_casters.py: the reference's inf * 0 = NaN against a skipped light)
    px.skipOK = ((__builtin_fabsf(px.F0.x) + __builtin_fabsf(px.F0.y) + __builtin_fabsf(px.F0.z)) +
                 (__builtin_fabsf(px.kA.x) + __builtin_fabsf(px.kA.y) + __builtin_fabsf(px.kA.z))) < 0x1p40f;
}

// BRDF(s, Wi, V), BRDF.hlsl:163-194. As written: H = normalize(Wo + Wi) (IEEE quotients through rc.div), NdotH, nh2*(a2-1)+1.
// Regrouped (contract v2-v4): fma(F, sG - kA, kA) with sG = (D*G)*rcp(denom) and the per-pixel kA = ((1-metal)*albedo)*rcp(PI).
// `rc` is the reciprocal / sqrt / quotient policy (vq_devmath.h).
// hh = |Wo + Wi|^2 in the reading AR (dot / dot_lit): handed in by the hot loop, which also folds it into its validity minimum
template <int AR, class R>
VQD f3 brdf_t(const Pixel& px, f3 Wi, R& rc, f3 Hs, float hh) {
    f3 H;                                                    // normalize(Wo + Wi) :168
    if (AR) {
        H = mul(Hs, rc.rsqrt(hh));                           // DXC reading: one correctly rounded rsqrt, three products
    } else {
        float rH;
        const float Hl = rc.sqrt_rcp(hh, &rH);               // length(Wo + Wi) and its reciprocal from one v_rsq_f32 (vq_devmath.h:sqrt_rcp_newton)
        H = mk3(rc.div(Hs.x, Hl, rH), rc.div(Hs.y, Hl, rH), rc.div(Hs.z, Hl, rH));
    }
    const float NdotH = saturate(dot_r<AR>(px.Nn, H));       // :169
    const float dNL = dot(px.Nn, Wi);
    const float NdotL = saturate(dNL);
    // Fresnel_Schlick(H, V, F0) :132-136
    const float x5 = 1.0f - max_(0.0f, dot(H, px.V));
    const float p5 = px.p5ExpLog ? pow5_explog(x5) : pow5(x5);   // default: x*((x*x)*(x*x)), FXC's mul-only pattern (contract v4)
    const f3 F = mk3(fma_(px.omF0.x, p5, px.F0.x), fma_(px.omF0.y, p5, px.F0.y), fma_(px.omF0.z, p5, px.F0.z));
    // D*G/denom with the three divisions merged into one (contract v3):
    //   D = a2/(PI t^2) (NormalDistributionGGX :65-79; 1 when PI t^2 < EPSILON), G = G1V * NL/(NL(1-k)+k+1e-4) (Geometry_Smith :118-121)
    //   sG = ((a2*G1V) * NL) * rcp((PI t^2 * gL) * denom)      [ (G1V*NL) * rcp(gL*denom) on the EPSILON branch ]
    const float NL = max_(0.0f, dNL);
    const float gL = fma_(NL, px.omk, px.k) + 0.0001f;
    const float nh2 = NdotH * NdotH;
    const float t = nh2 * px.a2m1 + 1.0f;                    // :77 as written: the product is rounded before the sum
    const float dd = PI_ * (t * t);
    const float den = max_(px.NdotV4 * NdotL, 0.0001f);
    float sG;
    if constexpr (R::kGgxDenomAboveEps) {
        sG = (px.a2G1V * NL) * rc((dd * gL) * den);          // `if (denom < EPSILON) return 1` (:76) cannot trigger: proven below (RcpTrust)
    } else {
        const bool eps = dd < EPSILON_;
        const float num = (eps ? px.G1V : px.a2G1V) * NL;
        const float d3 = eps ? gL * den : (dd * gL) * den;
        sG = num * rc(d3);
    }
    // Id + Is = (1-F)*kA + F*sG regrouped as kA + F*(sG - kA) (contract v3)
    return mk3(fma_(F.x, sG - px.kA.x, px.kA.x), fma_(F.y, sG - px.kA.y, px.kA.y), fma_(F.z, sG - px.kA.z, px.kA.z));
}
template <int AR, class R>
VQD f3 brdf_t(const Pixel& px, f3 Wi, R& rc) {
    const f3 Hs = add(px.Wo, Wi);
    return brdf_t<AR>(px, Wi, rc, Hs, AR ? dot(Hs, Hs) : dot_lit(Hs, Hs));
}

// acc + b * (cb * w): cb = l.color * l.brightness, w = attenuation [* cone] * NdotL (one mad per channel)
VQD f3 lit(f3 acc, f3 b, f3 cb, float w) { return mk3(fma_(b.x, cb.x * w, acc.x), fma_(b.y, cb.y * w, acc.y), fma_(b.z, cb.z * w, acc.z)); }
VQD f3 light_cb(const VQ_float3& color, float brightness) { return mk3(color.x * brightness, color.y * brightness, color.z * brightness); }

// CalculatePointLightIllumination, Lighting.hlsl:308-322 (general form, shadow casters)
template <int AR, class R>
VQD f3 point_light_t(const Pixel& px, f3 lpos, float range, f3 cb, f3 acc, R& rc) {
    const f3 d = sub(lpos, px.P);
    const float dd = dot_r<AR>(d, d);
    float rD;                                                // one reciprocal for the quotients of normalize(Lw - P) and for AttenuationBRDF :29-32
    const float D = rc.sqrt_rcp(dd, &rD);                    // length(Lw - P) as written; the literal normalize() shares the sqrt
    if (D < range) {
        const f3 Wi = AR ? mul(d, rc.rsqrt(dd)) : mk3(rc.div(d.x, D, rD), rc.div(d.y, D, rD), rc.div(d.z, D, rD));
        const float NdotL = saturate(dot(px.Nraw, Wi));
        const float w = (rD * rD) * NdotL;                   // 1/(D*D) as (1/D)*(1/D) (contract v3)
        return lit(acc, brdf_t<AR>(px, Wi, rc), cb, w);
    }
    return acc;
}
template <int AR>
VQD f3 point_light(const Pixel& px, const VQ_PointLight& l) {       // per-op validity flag; used for the <= 5 casters
    const f3 zero = mk3(0.0f, 0.0f, 0.0f), cb = light_cb(l.color, l.brightness);
    RcpFast fast;
    f3 r = point_light_t<AR>(px, ld3(l.position), l.range, cb, zero, fast);
    if (__builtin_expect(!fast.ok, 0)) { RcpIEEE ieee; r = point_light_t<AR>(px, ld3(l.position), l.range, cb, zero, ieee); }
    return r;
}

// Hot-loop form: I = CalculatePointLightIllumination(..., acc = I). All reciprocals / square roots use the unchecked
// fast sequences (RcpTrust); their validity is PROVEN from range tests instead of being checked per operation, and the tests are
// folded into ONE comparison per pixel after the loop (vmin, below):
//   pixel  : roughness in [0.04,1]  (px.fastOK + the wave-uniform `eps` test of k_forward_lighting)  =>  k in [1/8,1/2], 1-k in [1/2,7/8],
//              a2 in [2.5e-6,1], hence
//              gL = fma(NL,1-k,k)+1e-4 in [0.125, 1.4]   (NL = max(0,.) <= 1+eps, NaN -> 0)
//              pi t^2 in [1.9e-11, pi] (t = nh2*(a2-1) + 1, product and sum each rounded — contract v5 — in [a2 - 2^-24, 1], nh2 saturated)
//              denom = max(4 NdotV NdotL, 1e-4) in [1e-4, 4]
//              => the merged reciprocal's operand (pi t^2 * gL) * denom in [2.4e-16, 17.6]: operand and result normal
//   frame  : every coordinate of every light position is +0 or has magnitude in [2^-40, 2^40], every rangeSq <= 2^60 (host: FrameConstants::pointFastOK)
//   pixel  : every coordinate of P is 0 or has magnitude in [2^-40, 2^40]; every component of Wo is 0 or >= 2^-40; no (-0) - (+0) on any axis (setup_pixel)
//            => a component of Lw-P is +0 or has magnitude >= 2^-63 (the difference of two such floats is a multiple of the smaller one's ulp)
//   light  : dd = |Lw-P|^2 in [2^-80, 2^60) (upper: the cull, a NaN / inf dd fails it; lower: vmin) => D in [2^-40, 2^30]: root and reciprocal inside sqrt_rcp_newton's
//            proven domain; a quotient d.c / D is +0 or has magnitude >= 2^-93: normal, and its residual d.c - D q (a multiple of 2^-46 |d.c| >= 2^-109) is exact
//   light  : a component of Wo+Wi is +0 (x + (-x), or 0 + 0: never -0 because d.c is never -0), or — with Wo.c >= 2^-40 — has magnitude >= 2^-64, or — with Wo.c = 0 —
//            equals Wi.c (>= 2^-93); hh = |Wo+Wi|^2 in [2^-80, ~4] (vmin) => the quotients Hs.c / |Hs| are normal with exact residuals (>= 2^-139) as well
//   => the corrected quotients d/D, Hs/|Hs| (fdiv_rcp: exhaustively equal to IEEE division when nothing underflows; +0 / D = +0 either way) are the IEEE quotients
// all inside the exhaustively validated domains of rcp_newton / sqrt_newton (vq_devmath.h). A failed test (NaN inputs,
// degenerate geometry, roughness outside [0,1], a range beyond 2^30) redoes the pixel's point-light loop with IEEE operations
// (k_forward_lighting); where both are valid the two give identical bits, so the redo changes only what was invalid.
struct RcpTrust {
    // roughness >= 0.04 (every lane of the wave: `eps` in k_forward_lighting) => a2 >= 2.56e-6, t = RN(RN(nh2*(a2-1)) + 1) >= a2 - 2^-24 >= 2.5e-6
    // for nh2 in [0,1] => pi t^2 >= 1.9e-11 > EPSILON (1e-12): the GGX early-out never fires on this path
    static constexpr bool kGgxDenomAboveEps = true;
    VQD float operator()(float b) const { return rcp_newton(b); }
    VQD float sqrt(float x) const { return sqrt_newton(x); }
    VQD float sqrt_rcp(float x, float* r) const { return sqrt_rcp_newton(x, r); }
    VQD float div(float a, float b, float r) const { return fdiv_rcp(a, b, r); }
    VQD float rsqrt(float x) const { return rsqrt_cr_fast(x); }      // DXC reading: dd and hh lie in [2^-80, 2^60], inside its validated domain [2^-100, 2^100]
};
// The same fast sequences for a wave that holds a pixel of roughness < 0.04 (polished metal, a2 down to 0): there pi t^2 can fall below EPSILON
// and `if (denom < EPSILON) return 1` (BRDF.hlsl:76) must stay — as a select, like the IEEE form. The reciprocal's operand is then
// gL * denom in [1.25e-5, 5.6] on the early-out branch and (pi t^2 * gL) * denom >= 1e-12 * 0.125 * 1e-4 otherwise: still normal, so the
// validity proof above carries over with roughness in [0, 1] (k in [1/8, 1/2] as before). Where the early-out cannot fire the select form
// and RcpTrust give identical bits, so which of the two a wave runs is a pure speed choice (+3 VALU per light).
struct RcpTrustEps : RcpTrust { static constexpr bool kGgxDenomAboveEps = false; };
// `vmin` collects the smallest dd and hh over the lights that passed the range cull (ONE v_min3 per light; rounds 2-3 tracked the six components: three);
// the caller compares it with 2^-80 ONCE after the loop and, when the test fails, redoes the pixel's whole point-light loop with IEEE
// operations (k_forward_lighting) — the accumulator needs no copy per light and the loop carries no validity masks.
// dd <= 2^60 follows from dd < rangeSq <= 2^60 (FrameConstants::pointFastOK, host); a NaN / inf dd fails the cull like the reference's D < range.
// SKIP (wave-uniform choice of the caller): a light that faces away from every lane that passed the cull — NdotL = saturate(dot(N, Wi)) = +0 —
// adds b * (cb * +0) = +-0 to the accumulator when b and cb are finite, which changes nothing unless the accumulator holds a zero (the sign of
// -0 + +0). `izmin` = min |component| of I is kept per lane; when no lane has NdotL > 0 or a zero in I the BRDF (~70 of the ~108 VALU of a
// light) is skipped for the wave. Finite b: px.skipOK (finite F0, kA) and the proven ranges above; finite cb: FrameConstants::pointSkipOK;
// the product form of the Fresnel power only (exp2(5 log2 x) is NaN for the x = -6e-8 that a dot product rounding above 1 yields).
// On surface-coherent content about half the lights are behind the surface of a whole wave; white-noise normals never take this form.
template <int AR, class RC, bool SKIP>
VQD void add_point_light(const Pixel& px, const vqk::DevPointLight& l, f3& I, float& vmin, float& izmin) {
    const f3 lpos = mk3(l.px, l.py, l.pz), cb = mk3(l.cbx, l.cby, l.cbz);
    const f3 d = sub(lpos, px.P);
    const float dd = dot_r<AR>(d, d);                        // as written: D decides the range cull
    if (dd < l.rangeSq) {                                    // == (length(Lw - P) < l.range), exactly (host-made threshold): culled lights
        RC rc;                                               // need no square root; wave-coherent (execz skip)
        float rD;
        const float D = rc.sqrt_rcp(dd, &rD);                // dd in [2^-80, 2^60]: inside the proven domain of sqrt_rcp_newton
        const f3 Wi = AR ? mul(d, rc.rsqrt(dd)) : mk3(fdiv_rcp(d.x, D, rD), fdiv_rcp(d.y, D, rD), fdiv_rcp(d.z, D, rD));    // (Lw - P) / length(Lw - P) | (Lw - P) * rsqrt
        const f3 Hs = add(px.Wo, Wi);
        const float dNL = dot(px.Nraw, Wi);
        const float hh = AR ? dot(Hs, Hs) : dot_lit(Hs, Hs);                 // the operand of brdf_t's root
        vmin = __builtin_fminf(__builtin_fminf(vmin, dd), hh);               // one v_min3_f32, in front of the skip like the three it replaces (behind it the skip form ran 4 % slower)
        if (SKIP) { if (__builtin_amdgcn_ballot_w64((dNL > 0.0f) | !(izmin > 0.0f)) == 0) return; }
        const float NdotL = saturate(dNL);
        const float w = (rD * rD) * NdotL;
        const f3 b = brdf_t<AR>(px, Wi, rc, Hs, hh);
        I = lit(I, b, cb, w);
        if (SKIP) izmin = min3abs(I);
    }
}
template <int AR, class RC, bool SKIP>
VQD void point_light_loop(const Pixel& px, const vqk::DevPointLight* pts, int nP, f3& I, float& vmin) {
    float izmin = SKIP ? min3abs(I) : 0.0f;
    for (int p = 0; p < nP; ++p) add_point_light<AR, RC, SKIP>(px, pts[p], I, vmin, izmin);
}

// SpotlightIntensity :57-73 + CalculateSpotLightIllumination :323-333 (no range cull), round 6.
// What the HLSL writes three times is evaluated once:
//   * pd = normalize(P - l.position) is -Wi BIT FOR BIT (a - b = -(b - a) in round-to-nearest, x / D and x * r are odd in x, and the dot product of a negated
//     vector is the negated dot product in either reading), so dot(pd, sd) = -dot(Wi, sd): one normalize instead of two;
//   * sd = normalize(l.spotDir), l.color * l.brightness and 1 / (outer - inner) do not depend on the pixel: formed on the host with the shader's operations
//     (FrameConstants::spot, capi.hip);
//   * length(Lw - P), its reciprocal and the quotients of the two normalizes come from the exactly rounded fast sequences (policy R, vq_devmath.h) with one validity flag
//     per light (RcpFast), redone with IEEE operations when the flag drops — identical bits where both are valid.
// spot_geometry: Wi and w = (cone / D^2) * NdotL, the light's scalar weight (:330-332).
template <int AR, class R>
VQD void spot_geometry(const Pixel& px, const VQ_SpotLight& l, const vqk::DevSpotLight& ds, R& rc, f3& Wi, float& w) {
    const f3 d = sub(ld3(l.position), px.P);
    const float dd = dot_r<AR>(d, d);
    float rD;
    const float D = rc.sqrt_rcp(dd, &rD);                    // length(l.position - P) and 1 / it
    Wi = AR ? mul(d, rc.rsqrt(dd)) : mk3(rc.div(d.x, D, rD), rc.div(d.y, D, rD), rc.div(d.z, D, rD));
    const float theta = acos_(-dot_r<AR>(Wi, mk3(ds.sdx, ds.sdy, ds.sdz)));      // acos(dot(normalize(P - l.position), normalize(l.spotDir))) :60-61
    const float a = theta - l.innerConeAngle, den = l.outerConeAngle - l.innerConeAngle;
    const float q = (ds.flags & 1) ? rc.div(a, den, ds.rConeDen) : fdiv_(a, den);  // wave-uniform choice; the IEEE quotient either way
    float cone = 1.0f - q;                                   // :71
    cone = (theta <= l.innerConeAngle) ? 1.0f : cone;        // :67
    cone = (theta > l.outerConeAngle) ? 0.0f : cone;         // :63
    const float NdotL = saturate(dot(px.Nraw, Wi));
    w = (cone * (rD * rD)) * NdotL;
}
// the light's contribution added to acc, IEEE operations throughout (the redo path, and the reading of the oracle)
template <int AR>
VQD f3 spot_light_ieee(const Pixel& px, const VQ_SpotLight& l, const vqk::DevSpotLight& ds, f3 acc, f3* WiOut) {
    RcpIEEE rc;
    f3 Wi; float w;
    spot_geometry<AR>(px, l, ds, rc, Wi, w);
    if (WiOut) *WiOut = Wi;
    return lit(acc, brdf_t<AR>(px, Wi, rc), mk3(ds.cbx, ds.cby, ds.cbz), w);
}
// Fast form. `mayContribute` (out): false when the WAVE skipped the BRDF because no lane can change its accumulator — every lane has w == +0 (outside the cone or
// facing away; rD is finite: the flag), a BRDF that is finite whatever the light (px.skipOK, px.fastOK: roughness in [0, 1], finite F0 / kA / Wo, hh in the proven domain — the
// reasoning of add_point_light<.., SKIP>), finite color * brightness (ds.flags bit 1) and — `accNoZero` — an accumulator without a zero component, the one value that
// adding b * (cb * +0) = +-0 could change (-0 + +0 = +0). A spot light covers a few per cent of a frame: most waves take this exit, at the cost of the geometry alone.
template <int AR>
VQD f3 spot_light(const Pixel& px, const VQ_SpotLight& l, const vqk::DevSpotLight& ds, f3 acc, bool laneSkipOK, bool* mayContribute, f3* WiOut = nullptr) {
    RcpFast fast;
    f3 Wi; float w;
    spot_geometry<AR>(px, l, ds, fast, Wi, w);
    if (WiOut) *WiOut = Wi;
    const f3 Hs = add(px.Wo, Wi);
    const float hh = AR ? dot(Hs, Hs) : dot_lit(Hs, Hs);
    const bool idle = laneSkipOK & fast.ok & sqrt_rcp_fast_ok(hh) & (w == 0.0f);
    if (((ds.flags & 2) != 0) & !px.p5ExpLog & (__builtin_amdgcn_ballot_w64(!idle) == 0)) { *mayContribute = false; return acc; }
    *mayContribute = true;
    f3 r = lit(acc, brdf_t<AR>(px, Wi, fast, Hs, hh), mk3(ds.cbx, ds.cby, ds.cbz), w);
    if (__builtin_expect(!fast.ok, 0)) r = spot_light_ieee<AR>(px, l, ds, acc, WiOut);
    return r;
}

// CalculateDirectionalLightIllumination :334-345; Wi = normalize(-l.lightDirection) comes from the host (FrameConstants::dirWi)
template <int AR>
VQD f3 directional_light(const Pixel& px, const VQ_DirectionalLight& l, f3 Wi) {
    const float NdotL = saturate(dot(px.Nraw, Wi));
    const f3 cb = light_cb(l.color, l.brightness), zero = mk3(0.0f, 0.0f, 0.0f);
    RcpFast fast;
    f3 r = lit(zero, brdf_t<AR>(px, Wi, fast), cb, NdotL);
    if (__builtin_expect(!fast.ok, 0)) { RcpIEEE rc; r = lit(zero, brdf_t<AR>(px, Wi, rc), cb, NdotL); }
    return r;
}

VQD f3 mul_v_m3(f3 v, float c, float s) {     // mul(v, GetHDRIRotationMatrix) with m = {c,0,s; 0,1,0; -s,0,c}, as written (zero terms kept)
    return mk3((v.x * c + v.y * 0.0f) + v.z * -s,
               (v.x * 0.0f + v.y * 1.0f) + v.z * 0.0f,
               (v.x * s + v.y * 0.0f) + v.z * c);
}

// CalculateEnvironmentMapIllumination(+_DiffuseOnly), Lighting.hlsl:348-395 ; EnvironmentBRDF BRDF.hlsl:196-207
template <int AR>
VQD f3 environment(const Pixel& px, const vqk::FrameConstants* fc) {
    const float sn = fc->hdriSin, cs = fc->hdriCos;          // sin / cos(-fHDRIOffsetInRadians), correctly rounded, from the host (capi.hip)
    const float NdotV = saturate(dot_r<AR>(px.Nraw, px.V));  // everything here runs once per pixel: as written (contract v5)
    const f3 N = mul_v_m3(px.Nraw, cs, sn);
    const float4 irr = sample_cube_rgba16f(fc->env.diffuse_cube, fc->env.diffuse_res, N);
    f3 spec = mk3(0, 0, 0); float2 sb = make_float2(0, 0);
    if (!fc->perView.EnvironmentMapDiffuseOnlyIllumination) {
        const f3 R = mul_v_m3(reflect_r<AR>(neg(px.V), px.Nraw), cs, sn);
        const int maxLod = f2i_trunc(fc->perView.MaxEnvMapLODLevels);
        int mip = f2i_trunc(px.roughness * (float)maxLod);
        mip = min(max(mip, 0), fc->env.spec_mips - 1);
        // texel offset of level `mip` in the mip-major cube: sum_{m<mip} 6*(res0>>m)^2; for a power-of-two res0 (every mip
        // exact) that is 8*(res0^2 - (res0>>mip)^2)
        const int res0 = fc->env.spec_res0, rm = res0 >> mip;
        uint32_t off;
        if ((res0 & (res0 - 1)) == 0) off = 8u * (uint32_t)(res0 * res0 - rm * rm);
        else { off = 0; for (int m = 0; m < mip; ++m) { const uint32_t r = (uint32_t)(res0 >> m); off += 6u * r * r; } }
        const float4 sp = sample_cube_rgba16f((const h4*)fc->env.specular_cube + off, rm, R);
        spec = mk3(sp.x, sp.y, sp.z);
        sb = sample_2d_rg16f_clamp(fc->env.brdf_lut, fc->env.lut_size, fc->env.lut_size, NdotV, px.roughness);
    }
    const float p5 = px.p5ExpLog ? pow5_explog(1.0f - NdotV) : pow5(1.0f - NdotV);   // FresnelWithRoughness :152-156
    const float omr = 1.0f - px.roughness;
    const f3 Ks = mk3(px.F0.x + (max_(omr, px.F0.x) - px.F0.x) * p5, px.F0.y + (max_(omr, px.F0.y) - px.F0.y) * p5, px.F0.z + (max_(omr, px.F0.z) - px.F0.z) * p5);
    const f3 Kd = mk3((1.0f - Ks.x) * px.omm, (1.0f - Ks.y) * px.omm, (1.0f - Ks.z) * px.omm);
    const f3 diffuse = mk3(irr.x * px.albedo.x, irr.y * px.albedo.y, irr.z * px.albedo.z);
    const f3 specular = mk3(spec.x * (Ks.x * sb.x + sb.y), spec.y * (Ks.y * sb.x + sb.y), spec.z * (Ks.z * sb.x + sb.y));
    return mk3(Kd.x * diffuse.x + specular.x, Kd.y * diffuse.y + specular.y, Kd.z * diffuse.z + specular.z);
}

VQD float4 mul_M_v(const VQ_matrix& M, f3 P) {     // HLSL mul(M, float4(P,1)) == row vector * M_cpu
    float o[4];
    for (int j = 0; j < 4; ++j) o[j] = ((P.x * M.m[0][j] + P.y * M.m[1][j]) + P.z * M.m[2][j]) + 1.0f * M.m[3][j];   // as written
    return make_float4(o[0], o[1], o[2], o[3]);
}

// OmnidirectionalShadowTestPCF, Lighting.hlsl:110-174; SAMPLE_OFFSET_DIRS_NORMALIZED :123-131.
// Round 6: the 20 taps are unrolled with their offsets as literals and fetched TOGETHER — all addresses first, then the loads, then the comparisons — where the rolled
// loop waited for each texel before it formed the next address (three scalar loads of the offset tables, eight branches and one full memory latency per tap). The point
// fetch of a cube texel is branch-free: the reciprocal of the major axis from the unchecked Markstein sequence with ONE validity flag for the 20 taps (a denormal / zero /
// non-finite major axis redoes them with the checked operations — identical bits where both are valid), the texel index as (int)med3(floor(c), 0, dim - 1) (== the clamp of
// the saturating truncation for every float, NaN -> 0 included). The count of lit taps is an integer, / 20 the corrected product with RN(1 / 20).
VQD int texel_index(float c, float fdimm1) { return (int)__builtin_amdgcn_fmed3f(__builtin_floorf(c), 0.0f, fdimm1); }
template <bool CHECKED>
VQD uint32_t cube_point_offset(int dim, float fdim, float fdimm1, f3 d, bool& ok) {          // dword index of the texel fetch_cube_point reads
    const float ax = abs_(d.x), ay = abs_(d.y), az = abs_(d.z);
    const bool isZ = (az >= ax) & (az >= ay);
    const bool isY = !isZ & (ay >= ax);
    const bool isX = !(isZ | isY);
    const float major = isZ ? d.z : (isY ? d.y : d.x);
    const float ma = isZ ? az : (isY ? ay : ax);
    const bool neg = major < 0.0f;
    const float sc0 = isX ? -d.z : d.x;
    const float tc0 = isY ? d.z : -d.y;
    const float sc = (neg & !isY) ? -sc0 : sc0;
    const float tc = (neg & isY) ? -tc0 : tc0;
    const int face = (isZ ? 4 : (isY ? 2 : 0)) + (neg ? 1 : 0);
    float r;
    if (CHECKED) r = rcp(ma); else { r = rcp_newton(ma); ok = ok & is_normal(r); }
    const float su = (sc * r) * 0.5f + 0.5f, sv = (tc * r) * 0.5f + 0.5f;
    const int x = texel_index(su * fdim, fdimm1), y = texel_index(sv * fdim, fdimm1);
    return (uint32_t)((face * dim + y) * dim + x);
}
#define PA 0.5773502691896258f
#define PB 0.7071067811865475f
__device__ const float kOmniDX[20] = {  PA,  PA, -PA, -PA,  PA,  PA, -PA, -PA,  PB,  PB, -PB, -PB,  PB, -PB,  PB, -PB,  0,  0,  0,  0 };     // the rare checked redo indexes these
__device__ const float kOmniDY[20] = {  PA, -PA, -PA,  PA,  PA, -PA, -PA,  PA,  PB, -PB, -PB,  PB,  0,  0,  0,  0,  PB, -PB, -PB,  PB };
__device__ const float kOmniDZ[20] = {  PA,  PA,  PA,  PA, -PA, -PA, -PA, -PA,  0,  0,  0,  0,  PB,  PB, -PB, -PB,  PB,  PB, -PB, -PB };
template <int AR>
VQD float omni_pcf(const float* cubeArr, int dim, int index, f3 Lw, float farPlane, float depthBias, float viewDist) {
    constexpr float DX[20] = {  PA,  PA, -PA, -PA,  PA,  PA, -PA, -PA,  PB,  PB, -PB, -PB,  PB, -PB,  PB, -PB,  0,  0,  0,  0 };
    constexpr float DY[20] = {  PA, -PA, -PA,  PA,  PA, -PA, -PA,  PA,  PB, -PB, -PB,  PB,  0,  0,  0,  0,  PB, -PB, -PB,  PB };
    constexpr float DZ[20] = {  PA,  PA,  PA,  PA, -PA, -PA, -PA, -PA,  0,  0,  0,  0,  PB,  PB, -PB, -PB,  PB,  PB, -PB, -PB };
    const float diskRadius = (1.0f + fdiv_(viewDist, farPlane)) * 0.125f;
    const float* cube = cubeArr + (size_t)index * 6 * dim * dim;
    const float lenLw = length_r<AR>(Lw);
    const float fdim = (float)dim, fdimm1 = (float)(dim - 1);
    uint32_t off[20];
    bool ok = true;
    #pragma unroll
    for (int i = 0; i < 20; ++i)
        off[i] = cube_point_offset<false>(dim, fdim, fdimm1, mk3(-(Lw.x + DX[i] * diskRadius), -(Lw.y + DY[i] * diskRadius), -(Lw.z + DZ[i] * diskRadius)), ok);
    if (__builtin_expect(!ok, 0)) {
        #pragma unroll 1
        for (int i = 0; i < 20; ++i)
            off[i] = cube_point_offset<true>(dim, fdim, fdimm1, mk3(-(Lw.x + kOmniDX[i] * diskRadius), -(Lw.y + kOmniDY[i] * diskRadius), -(Lw.z + kOmniDZ[i] * diskRadius)), ok);
    }
    float closest[20];
    #pragma unroll
    for (int i = 0; i < 20; ++i) closest[i] = cube[off[i]];
    int count = 0;
    #pragma unroll
    for (int i = 0; i < 20; ++i) count += (lenLw > (closest[i] * farPlane + depthBias) + 0.001f) ? 1 : 0;
    return 1.0f - fdiv_rcp((float)count, 20.0f, 0.05f);
}
#undef PA
#undef PB
// ShadowTestPCF :177-218 (useTanBias) / ShadowTestPCF_Directional :222-272 (raw bias). Round 6: the 25 taps of the 5 x 5 kernel share 5 column and 5 row
// coordinates — tap (x, y) samples (u + x * tx, v + y * ty), each coordinate a function of x or of y alone — so the POINT_WRAP address arithmetic
// (floor(coord * dim), wrap) runs 10 times instead of 50 and a tap is one add + one load + one compare; a power-of-two map wraps with an AND instead of
// two integer divisions. The sum of 25 zeros and ones is exact in any order: counted in an integer. shadow / 25 through the corrected product with RN(1 / 25)
// (fdiv_rcp: equal to the IEEE quotient for every normal operand pair, +0 / 25 = +0).
VQD float pcf_2d(const float* slice, int dim, float2 smDims, float4 lsp, float bias) {
    const f3 p = mk3(fdiv_(lsp.x, lsp.w), fdiv_(lsp.y, lsp.w), fdiv_(lsp.z, lsp.w));
    if (p.x < -1.0f || p.x > 1.0f || p.y < -1.0f || p.y > 1.0f || p.z < 0.0f || p.z > 1.0f) return 0.0f;
    const float tx = rcp(smDims.x), ty = rcp(smDims.y);
    const float u = 0.5f + p.x * 0.5f, v = 0.5f + p.y * -0.5f;
    const float ref = p.z - bias;
    const float fdim = (float)dim;
    const bool pot = (dim & (dim - 1)) == 0;                 // wave-uniform
    uint32_t col[5], row[5];                                 // byte offsets of the five columns / rows
    #pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int ix = f2i_floor((u + (float)(k - 2) * tx) * fdim), iy = f2i_floor((v + (float)(k - 2) * ty) * fdim);
        const int wx = pot ? (ix & (dim - 1)) : wrapi(ix, dim), wy = pot ? (iy & (dim - 1)) : wrapi(iy, dim);
        col[k] = (uint32_t)wx * 4u;
        row[k] = (uint32_t)wy * (uint32_t)dim * 4u;          // a slice is at most 2^32 bytes (dim <= 32768)
    }
    int count = 0;
    // The five columns are five CONSECUTIVE texels unless the kernel straddles the map's wrap or a rounding of u + x * tx skips one: then a row of the kernel is 20
    // contiguous bytes and goes out as one 16-byte + one 4-byte load (4-byte aligned: the hardware's unaligned-access mode) instead of five — the pass is bound by L1 tag
    // lookups on incoherent positions (every lane's tap is its own cache line: 25 lookups per pixel and caster; now ~11), profiles/r6t_pcf_rows.md. Wave-uniform choice.
    const bool rowsContiguous = (col[1] == col[0] + 4u) & (col[2] == col[0] + 8u) & (col[3] == col[0] + 12u) & (col[4] == col[0] + 16u);
    if (__builtin_amdgcn_ballot_w64(!rowsContiguous) == 0) {
        typedef float fl4 __attribute__((ext_vector_type(4)));
        struct __attribute__((packed, aligned(4))) U4 { fl4 v; };
        #pragma unroll
        for (int y = 0; y < 5; ++y) {
            const char* r = (const char*)slice + (row[y] + col[0]);
            const fl4 a = ((const U4*)r)->v;
            const float b = *(const float*)(r + 16);
            count += ((ref > a.x) ? 1 : 0) + ((ref > a.y) ? 1 : 0) + ((ref > a.z) ? 1 : 0) + ((ref > a.w) ? 1 : 0) + ((ref > b) ? 1 : 0);
        }
    } else {
        #pragma unroll
        for (int x = 0; x < 5; ++x)
            #pragma unroll
            for (int y = 0; y < 5; ++y) {
                const float closest = *(const float*)((const char*)slice + (row[y] + col[x]));
                count += (ref > closest) ? 1 : 0;
            }
    }
    return 1.0f - fdiv_rcp((float)count, 25.0f, 0.04f);
}

// PSMain :289-380 for ONE pixel whose G-buffer record is (g0, g1, g2, g3): returns float4(I_total, roughness) (:380).
// The wave-uniform choices inside (forms of the point-light loop) are taken over the lanes that are active at the call.
template <bool HAS_ENV, bool HAS_CASTERS, int AR = 0>
VQD float4 shade_pixel(const float4 g0, const float4 g1, const float4 g2, const float4 g3, const vqk::FrameConstants* fc) {
    Pixel px;
    const f3 cam = ld3(fc->perView.CameraPosition);
    px.p5ExpLog = fc->pow5ExpLog != 0;
    setup_pixel<AR>(px, g0, g1, g2, cam, fc->pointNegZeroAxes);
    const float ao = g0.w;
    // illumination accumulators, ForwardLighting.hlsl:290-293: diffuse*ao + emissive*intensity, as written
    f3 I = mk3(px.albedo.x * ao + g3.x * g3.w, px.albedo.y * ao + g3.y * g3.w, px.albedo.z * ao + g3.z * g3.w);

    if (HAS_ENV) I = add(I, environment<AR>(px, fc));                                             // :299-306

    // non-shadowing point lights :310-313 — point_lights[0..numPointLights) followed by the extension array, packed by
    // the host into 32-byte records {position, range, color*brightness}
    const vqk::DevPointLight* pts = (const vqk::DevPointLight*)(fc + 1);
    const int nP = fc->numPointAll;
    // Fast loop: unchecked reciprocal / sqrt sequences whose validity is established once per pixel (vmin, see add_point_light). The rare
    // pixel that fails (roughness outside [0, 1], a light exactly above the pixel on an axis, a range beyond 2^30, NaN inputs) is redone
    // from the accumulator's value before the loop with IEEE operations; where both are valid the two paths agree bit for bit.
    // (A software-pipelined prefetch of the next 32-byte record was measured and is not used: 1.034 ms vs 1.022 ms, profiles/r2c_shade_variants.md.)
    const f3 I0 = I;
    const bool fast = px.fastOK & (fc->pointFastOK != 0);
    float vmin = 0.0f;
    if (fast) {
        vmin = __builtin_inff();
        // wave-uniform choice among four forms of the same loop (identical bits wherever more than one applies):
        //   eps  : some lane has roughness < 0.04 -> the GGX EPSILON early-out stays in, as a select (RcpTrustEps)
        //   skip : every lane has a finite BRDF and a normal within 60 degrees of the first lane's (a surface, not noise) -> lights behind
        //          the surface of the whole wave cost the cull, the normalize and one dot product instead of the BRDF
        const bool eps = __builtin_amdgcn_ballot_w64(px.roughness < 0.04f) != 0;
        const f3 n0 = mk3(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, px.Nraw.x))),
                          __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, px.Nraw.y))),
                          __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, px.Nraw.z))));
        const bool laneSkipOK = px.skipOK & (dot(px.Nraw, n0) > 0.5f);
        const bool skip = (fc->pointSkipOK != 0) & !px.p5ExpLog & (__builtin_amdgcn_ballot_w64(!laneSkipOK) == 0);
        if (skip) { if (eps) point_light_loop<AR, RcpTrustEps, true>(px, pts, nP, I, vmin); else point_light_loop<AR, RcpTrust, true>(px, pts, nP, I, vmin); }
        else      { if (eps) point_light_loop<AR, RcpTrustEps, false>(px, pts, nP, I, vmin); else point_light_loop<AR, RcpTrust, false>(px, pts, nP, I, vmin); }
    }
    if (__builtin_expect(!(vmin >= 0x1p-80f), 0)) {
        I = I0;
        RcpIEEE ieee;
        for (int p = 0; p < nP; ++p) I = point_light_t<AR>(px, mk3(pts[p].px, pts[p].py, pts[p].pz), pts[p].range, mk3(pts[p].cbx, pts[p].cby, pts[p].cbz), I, ieee);
    }
    const VQ_SceneLighting& L = fc->perFrame.Lights;
    const int nS = L.numSpotLights;
    // the wave-level skip of spot lights that cannot change the accumulator (spot_light) needs a lane whose BRDF is finite whatever the light, and an accumulator without zeros
    const bool spotLaneOK = px.skipOK & px.fastOK;
    float izmin = min3abs(I);
    for (int s = 0; s < nS; ++s) {                                                                // :314-317
        bool hit;
        I = spot_light<AR>(px, L.spot_lights[s], fc->spot[s], I, spotLaneOK & (izmin > 0.0f), &hit);
        if (hit) izmin = min3abs(I);
    }

    if (HAS_CASTERS) {
        const int nPC = L.numPointCasters;
        if (nPC > 0) {
            const float viewDist = length_r<AR>(sub(px.P, cam));                              // :325 (pcfData.viewDistanceOfPixel): the same for every caster
            for (int pc = 0; pc < nPC; ++pc) {                                                // :321-339
                const VQ_PointLight& l = L.point_casters[pc];
                const f3 Lw = sub(ld3(l.position), px.P);
                const float D = length_r<AR>(Lw);
                if (D < l.range) {
                    const f3 c = point_light<AR>(px, l);
                    const float sh = omni_pcf<AR>(fc->sm.point, fc->sm.point_dim, pc, Lw, l.range, l.depthBias, viewDist);
                    I = mk3(fma_(c.x, sh, I.x), fma_(c.y, sh, I.y), fma_(c.z, sh, I.z));
                }
            }
            izmin = min3abs(I);
        }
        const int nSC = L.numSpotCasters;
        for (int sc = 0; sc < nSC; ++sc) {                                                    // :342-356
            const VQ_SpotLight& l = L.spot_casters[sc];
            // illumination of the light from a zero accumulator, then * shadow factor. When no lane of the wave is lit (spot_light's exit: every lane w == +0, finite BRDF)
            // the illumination is (+0, +0, +0) in every lane and I + 0 * sh = I for the finite sh of pcf_2d unless I holds a zero (izmin): the PCF is skipped with it
            bool hit;
            f3 Ln;                                                                            // normalize(l.position - P) :348 == the Wi of the light's illumination
            const f3 c = spot_light<AR>(px, l, fc->spot[VQ_NUM_LIGHTS__SPOT + sc], mk3(0.0f, 0.0f, 0.0f), spotLaneOK & (izmin > 0.0f), &hit, &Ln);
            if (!hit) continue;
            const float NdotL = saturate(dot_r<AR>(px.Nraw, Ln));
            const float4 lsp = mul_M_v(L.shadowViews[sc], px.P);
            const float bias = l.depthBias * tan_(acos_(NdotL));
            const float sh = pcf_2d(fc->sm.spot + (size_t)sc * fc->sm.spot_dim * fc->sm.spot_dim, fc->sm.spot_dim,
                                    make_float2(fc->perFrame.f2SpotLightShadowMapDimensions.x, fc->perFrame.f2SpotLightShadowMapDimensions.y), lsp, bias);
            I = mk3(fma_(c.x, sh, I.x), fma_(c.y, sh, I.y), fma_(c.z, sh, I.z));
            izmin = min3abs(I);
        }
    }
    {                                                                                         // :360-377
        const VQ_DirectionalLight& l = L.directional;
        if (l.enabled) {
            float sh = 1.0f;
            if (HAS_CASTERS && l.shadowing) {
                const float4 lsp = mul_M_v(L.shadowViewDirectional, px.P);
                sh = pcf_2d(fc->sm.directional, fc->sm.dir_dim,
                            make_float2(fc->perFrame.f2DirectionalLightShadowMapDimensions.x, fc->perFrame.f2DirectionalLightShadowMapDimensions.y), lsp, l.depthBias);
            }
            const f3 c = directional_light<AR>(px, l, mk3(fc->dirWi[0], fc->dirWi[1], fc->dirWi[2]));
            I = mk3(fma_(c.x, sh, I.x), fma_(c.y, sh, I.y), fma_(c.z, sh, I.z));
        }
    }
    return make_float4(I.x, I.y, I.z, px.roughness);                                          // :380
}

} // namespace
