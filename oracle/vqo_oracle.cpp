// vqo_oracle.cpp — CPU restatement of VQEngine's forward-PBR / IBL / convolution / post path.
//
// ORACLE / TEST INFRASTRUCTURE ONLY. Only tests/, __graft_entry__.smoke() and bench.py's
// `cpu_baseline` leg may load this library (as the checker / reported CPU baseline). The product
// (vqengine_amd/, libvqhip.so) never includes, links or calls anything in oracle/.
//
// PARITY: the reference holds no golden vectors / KATs for this path and its D3D12 renderer cannot be
// built here (Windows/DXC/WARP only). What CAN run here is the reference's shader SOURCE: oracle/_ref (Makefile target `ref`)
// compiles ForwardLighting / BRDF / Lighting / ShadingMath / CubemapConvolution / GaussianBlur / Tonemapper / HDR / Skydome /
// Visualization / ApplyReflections .hlsl from where they lie, through oracle/ref_src/hlsl_shim.h, and runs them on the CPU.
//   PINNED  (tests/test_ref_pinning.py live, tests/golden/ref_outputs.npz on the GPU box): the ALGORITHM of every pass in this
//           file — constants, branches, operand roles, loop bounds, sample sequences — against that run, to about one ulp in
//           the median, with a tail explained by binary32 conditioning (numbers: DESIGN.md §5).
//   NOT PINNED: the bits a DXC + driver compile of the same HLSL produces (fast-math regrouping, approximate intrinsics: the
//           arithmetic contract of vqo_math.h picks one legal outcome), texture filtering and rasteriser interpolation (no source
//           in the reference: vqo_sampling.h restates D3D's rules; the _ref harness uses the same statement).
// Intrinsic lowering: vqo_math.h; sampling: vqo_sampling.h.
//
// Every function cites the reference file:line it follows (paths relative to the VQEngine tree).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <omp.h>

#include "../include/vqhip.h"
#include "vqo_math.h"
#include "vqo_sampling.h"

using namespace vqo;

namespace {

// vqo_set_fresnel_pow: 0 = the product x*((x*x)*(x*x)) (contract v4, default), 1 = exp2(5*log2 x) (the engine's DXC flags; v1-v3)
int g_pow5ExpLog = 0;
inline float fresnel_pow5(float x) { return g_pow5ExpLog ? pow_(x, 5.0f) : pow5_(x); }

const float PI_         = 3.14159265359f;   // Shaders/ShadingMath.hlsl:25
const float TWO_PI_     = 6.28318530718f;   // :26
const float PI_OVER_TWO = 1.5707963268f;    // :27
const float EPSILON_    = 0.000000000001f;  // Shaders/BRDF.hlsl:21

struct Surface {            // BRDF_Surface, Shaders/BRDF.hlsl:50-58
    f3 N; float roughness; f3 diffuseColor; float metalness; float emissiveIntensity; f3 emissiveColor;
};

inline f3 to3(const VQ_float3& v) { return { v.x, v.y, v.z }; }

// ---- Shaders/BRDF.hlsl ------------------------------------------------------------------------
// NormalDistributionGGX, BRDF.hlsl:65-79  (pow(.,2) -> x*x)
inline float NormalDistributionGGX(float NdotH, float roughness) {
    const float a = roughness * roughness;
    const float a2 = a * a;
    const float nh2 = NdotH * NdotH;
    const float t = fma_(nh2, a2 - 1.0f, 1.0f);                    // (nh2 * (a2 - 1) + 1) as one mad
    const float denom = PI_ * (t * t);
    if (denom < EPSILON_) return 1.0f;
    return div_(a2, denom);
}
// Geometry_Smiths_SchlickGGX, BRDF.hlsl:82-97
inline float Geometry_Smiths_SchlickGGX(f3 N, f3 V, float roughness) {
    const float rp1 = roughness + 1.0f;
    const float k = div_(rp1 * rp1, 8.0f);
    const float NV = max_(0.0f, dot(N, V));
    const float denom = fma_(NV, 1.0f - k, k) + 0.0001f;              // (NV*(1-k) + k) as one mad
    return div_(NV, denom);
}
// Geometry_Smiths_SchlickGGX_EnvironmentMap, BRDF.hlsl:100-115
inline float Geometry_Smiths_SchlickGGX_EnvironmentMap(f3 N, f3 V, float roughness) {
    const float k = div_(roughness * roughness, 2.0f);
    const float NV = max_(0.0f, dot(N, V));
    const float denom = fma_(NV, 1.0f - k, k) + 0.0001f;              // (NV*(1-k) + k) as one mad
    return div_(NV, denom);
}
// Geometry_Smith, BRDF.hlsl:118-121 (its 'k' argument is the roughness: BRDF() passes roughness, :184)
inline float Geometry_Smith(f3 N, f3 V, f3 L, float k) {
    return Geometry_Smiths_SchlickGGX(N, V, k) * Geometry_Smiths_SchlickGGX(N, L, k);
}
// GeometryEnvironmentMap, BRDF.hlsl:124-129
inline float GeometryEnvironmentMap(f3 N, f3 V, f3 L, float k) {
    float geomNV = Geometry_Smiths_SchlickGGX_EnvironmentMap(N, V, k);
    float geomNL = Geometry_Smiths_SchlickGGX_EnvironmentMap(N, L, k);
    return geomNV * geomNL;
}
// Fresnel_Schlick, BRDF.hlsl:132-136 — called as Fresnel_Schlick(H, V, F0) (:183): "N" is H, V is the caller's V
inline f3 Fresnel_Schlick(f3 N, f3 V, f3 F0) {
    const float p = fresnel_pow5(1.0f - max_(0.0f, dot(N, V)));
    return { fma_(1.0f - F0.x, p, F0.x), fma_(1.0f - F0.y, p, F0.y), fma_(1.0f - F0.z, p, F0.z) };      // F0 + (1-F0)*p, mad
}
// FresnelWithRoughness, BRDF.hlsl:152-156
inline f3 FresnelWithRoughness(float cosTheta, f3 F0, float roughness) {     // per pixel: as written (contract v5)
    const float p = fresnel_pow5(1.0f - cosTheta);
    const float omr = 1.0f - roughness;
    return { F0.x + (max_(omr, F0.x) - F0.x) * p, F0.y + (max_(omr, F0.y) - F0.y) * p, F0.z + (max_(omr, F0.z) - F0.z) * p };
}
// F_LambertDiffuse, BRDF.hlsl:158-161
inline f3 F_LambertDiffuse(f3 kd) { return { div_(kd.x, PI_), div_(kd.y, PI_), div_(kd.z, PI_) }; }

// BRDF, BRDF.hlsl:163-194 (contract v5). Two kinds of expressions:
//  (1) AS WRITTEN — everything that depends on the pixel only (Wo, N, NdotV, F0, k, G1(N,V), a2: the product hoists them out of the light
//      loop) and, per light, the chain into the GGX denominator: H = normalize(Wo + Wi), NdotH = saturate(dot(N,H)), nh2*(a2-1)+1.
//      At roughness < 0.2 that denominator cancels down to a2 ~ 1e-5..1e-3, so one ulp of NdotH is up to several per cent of D: only the
//      reference's own rounding sequence keeps a highlight pixel within one RGBA16F ulp of the reference's output.
//  (2) REGROUPED (contract v3/v4, what a fast-math D3D compile may emit; each insensitive to an ulp of its inputs):
//      specular = D*F*G/denom          ->  F * sG,  sG = (D*G) * rcp(denom)           (scalar)
//      diffuse  = (1-F)*(1-metal)*albedo/PI -> (1-F) * kA,  kA = ((1-metal)*albedo) * rcp(PI)   (light independent)
//      result   = Id + Is              ->  fma(F, sG - kA, kA)
inline f3 BRDF(const Surface& s, f3 Wi, f3 V) {
    // (1) per pixel, as written
    const f3 Wo = normalize_lit(V);                                          // :166
    const f3 N = normalize_lit(s.N);                                         // :167
    const float NdotV = saturate(dot_lit(N, Wo));                            // :171
    const f3 albedo = s.diffuseColor;
    const float roughness = s.roughness, metalness = s.metalness;
    const f3 F0 = { lerp_lit(0.04f, albedo.x, metalness), lerp_lit(0.04f, albedo.y, metalness), lerp_lit(0.04f, albedo.z, metalness) };   // :178
    const float rp1 = roughness + 1.0f;
    const float k = fdiv_(rp1 * rp1, 8.0f);                                  // :91
    const float NV = max_(0.0f, dot_lit(N, Wo));
    const float G1V = fdiv_(NV, (NV * (1.0f - k) + k) + 0.0001f);            // Geometry_Smiths_SchlickGGX(N, Wo, roughness) :92-95
    const float a = roughness * roughness, a2 = a * a;                       // :74-75
    // (1) per light, as written: the half vector and the GGX denominator
    const f3 H = normalize_lit(add(Wo, Wi));                                 // :168
    const float NdotH = saturate(dot_lit(N, H));                             // :169
    const float t = (NdotH * NdotH) * (a2 - 1.0f) + 1.0f;                    // :77  nh2 * (a2 - 1) + 1
    const float dd = PI_ * (t * t);                                          //      PI * pow(., 2)
    // (2) regrouped
    const float NdotL = saturate(dot(N, Wi));
    const f3 F = Fresnel_Schlick(H, V, F0);
    // D*G/denom with its three divisions merged into one (contract v3; fast-math arcp + reassoc):
    //   D  = a2 / (PI t^2)                      NormalDistributionGGX :65-79 (returns 1 when PI t^2 < EPSILON)
    //   G  = G1(N,V) * NL / (NL(1-k) + k + 1e-4) Geometry_Smith :118-121, Geometry_Smiths_SchlickGGX :82-97
    //   sG = ((a2*G1V) * NL) * rcp((PI t^2 * gL) * denom)          [ (G1V*NL) * rcp(gL*denom) on the EPSILON branch ]
    const float NL = max_(0.0f, dot(N, Wi));
    const float gL = fma_(NL, 1.0f - k, k) + 0.0001f;
    const float denom = max_((4.0f * NdotV) * NdotL, 0.0001f);
    const float sG = (dd < EPSILON_) ? (G1V * NL) * rcp(gL * denom) : ((a2 * G1V) * NL) * rcp((dd * gL) * denom);
    const float omm = 1.0f - metalness, invPI = rcp(PI_);
    const f3 kA = { (omm * albedo.x) * invPI, (omm * albedo.y) * invPI, (omm * albedo.z) * invPI };
    // Id + Is = (1-F)*kA + F*sG regrouped as kA + F*(sG - kA): one subtraction and one mad per channel (contract v3)
    return { fma_(F.x, sG - kA.x, kA.x), fma_(F.y, sG - kA.y, kA.y), fma_(F.z, sG - kA.z, kA.z) };
}
// EnvironmentBRDF, BRDF.hlsl:196-207
inline f3 EnvironmentBRDF(float NdotV, float roughness, float metallic, f3 diffuseColor, f3 diffuseIrradiance, f3 preFilteredSpecular, f2 F0ScaleBias) {
    // once per pixel: every expression as written (contract v5)
    const f3 F0 = { lerp_lit(0.04f, diffuseColor.x, metallic), lerp_lit(0.04f, diffuseColor.y, metallic), lerp_lit(0.04f, diffuseColor.z, metallic) };
    const f3 Ks = FresnelWithRoughness(NdotV, F0, roughness);
    const float omm = 1.0f - metallic;
    const f3 Kd = { (1.0f - Ks.x) * omm, (1.0f - Ks.y) * omm, (1.0f - Ks.z) * omm };
    const f3 diffuse = mul(diffuseIrradiance, diffuseColor);
    const f3 specular = { preFilteredSpecular.x * (Ks.x * F0ScaleBias.x + F0ScaleBias.y),
                          preFilteredSpecular.y * (Ks.y * F0ScaleBias.x + F0ScaleBias.y),
                          preFilteredSpecular.z * (Ks.z * F0ScaleBias.x + F0ScaleBias.y) };
    return { Kd.x * diffuse.x + specular.x, Kd.y * diffuse.y + specular.y, Kd.z * diffuse.z + specular.z };
}

// ---- Shaders/ShadingMath.hlsl -----------------------------------------------------------------
// DirectionToEquirectUV, ShadingMath.hlsl:70-80
inline f2 DirectionToEquirectUV(f3 v) {
    f2 uv = { atan2_(v.z, v.x), asin_(-v.y) };
    uv.x = div_(uv.x, -TWO_PI_); uv.y = div_(uv.y, PI_);
    uv.x += 0.5f; uv.y += 0.5f;
    return uv;
}
// RadicalInverse_VdC, ShadingMath.hlsl:87-95
inline float RadicalInverse_VdC(uint32_t bits) {
    bits = (bits << 16u) | (bits >> 16u);
    bits = ((bits & 0x55555555u) << 1u) | ((bits & 0xAAAAAAAAu) >> 1u);
    bits = ((bits & 0x33333333u) << 2u) | ((bits & 0xCCCCCCCCu) >> 2u);
    bits = ((bits & 0x0F0F0F0Fu) << 4u) | ((bits & 0xF0F0F0F0u) >> 4u);
    bits = ((bits & 0x00FF00FFu) << 8u) | ((bits & 0xFF00FF00u) >> 8u);
    return (float)bits * 2.3283064365386963e-10f;
}
// Hammersley, ShadingMath.hlsl:119-127
inline f2 Hammersley(uint32_t i, uint32_t count) { return { div_((float)i, (float)count), RadicalInverse_VdC(i) }; }

// ImportanceSampleGGX, BRDF.hlsl:217-238
inline f3 ImportanceSampleGGX(f2 Xi, f3 N, float roughness) {
    const float a = roughness * roughness;
    const float phi = (2.0f * PI_) * Xi.x;
    const float cosTheta = sqrt_(fdiv_(1.0f - Xi.y, 1.0f + (a * a - 1.0f) * Xi.y));
    const float sinTheta = sqrt_(1.0f - cosTheta * cosTheta);
    float sp, cp; sincos_(phi, &sp, &cp);
    f3 H = { cp * sinTheta, sp * sinTheta, cosTheta };
    const f3 up = abs_(N.z) < 0.999f ? f3{ 0, 0, 1 } : f3{ 1, 0, 0 };
    const f3 tangent = normalize(cross(up, N));
    const f3 bitangent = cross(N, tangent);
    const f3 sample = { (tangent.x * H.x + bitangent.x * H.y) + N.x * H.z,
                        (tangent.y * H.x + bitangent.y * H.y) + N.y * H.z,
                        (tangent.z * H.x + bitangent.z * H.y) + N.z * H.z };
    return normalize(sample);
}
// IntegrateBRDF, BRDF.hlsl:239-283
inline f2 IntegrateBRDF(float NdotV, float roughness, int count) {
    f3 V = { sqrt_(1.0f - NdotV * NdotV), 0.0f, NdotV };
    float F0Scale = 0, F0Bias = 0;
    const f3 N = { 0, 0, 1 };
    for (uint32_t i = 0; i < (uint32_t)count; ++i) {
        const f2 Xi = Hammersley(i, (uint32_t)count);
        const f3 H = ImportanceSampleGGX(Xi, N, roughness);
        const f3 L = normalize(reflect(neg(V), H));
        const float NdotL = max_(L.z, 0.0f);
        const float NdotH = max_(H.z, 0.0f);
        const float VdotH = max_(dot(V, H), 0.0f);
        if (NdotL > 0.0f) {
            const float G = GeometryEnvironmentMap(N, V, L, roughness);
            const float G_Vis = max_(div_(G * VdotH, NdotH * NdotV), 0.0001f);
            const float Fc = fresnel_pow5(1.0f - VdotH);
            F0Scale += (1.0f - Fc) * G_Vis;
            F0Bias += Fc * G_Vis;
        }
    }
    return { div_(F0Scale, (float)count), div_(F0Bias, (float)count) };
}

// ---- Shaders/Lighting.hlsl --------------------------------------------------------------------
inline float AttenuationBRDF(float dist) { return rcp(dist * dist); }   // Lighting.hlsl:29-32

// SpotlightIntensity, Lighting.hlsl:57-73
inline float SpotlightIntensity(const VQ_SpotLight& l, f3 worldPos) {
    const f3 pixelDir = normalize_lit(sub(worldPos, to3(l.position)));       // as written (contract v5): acos near 1 amplifies every ulp
    const f3 spotDir = normalize_lit(to3(l.spotDir));
    const float theta = acos_(dot_lit(pixelDir, spotDir));
    if (theta > l.outerConeAngle) return 0.0f;
    if (theta <= l.innerConeAngle) return 1.0f;
    return 1.0f - fdiv_(theta - l.innerConeAngle, l.outerConeAngle - l.innerConeAngle);
}

struct ShadowTestPCFData { f4 lightSpacePos; float depthBias, NdotL, viewDistanceOfPixel; };   // Lighting.hlsl:79-87

// OmnidirectionalShadowTestPCF, Lighting.hlsl:110-174 (BIAS at :143 is computed but unused there)
inline float OmnidirectionalShadowTestPCF(const ShadowTestPCFData& d, const float* cubeArr, int dim, int index, f3 Lw, float farPlane) {
    const float f3_ = 0.5773502691896258f, f2_ = 0.7071067811865475f;
    const f3 DIRS[20] = {
        { f3_, f3_, f3_ }, { f3_, -f3_, f3_ }, { -f3_, -f3_, f3_ }, { -f3_, f3_, f3_ },
        { f3_, f3_, -f3_ }, { f3_, -f3_, -f3_ }, { -f3_, -f3_, -f3_ }, { -f3_, f3_, -f3_ },
        { f2_, f2_, 0 }, { f2_, -f2_, 0 }, { -f2_, -f2_, 0 }, { -f2_, f2_, 0 },
        { f2_, 0, f2_ }, { -f2_, 0, f2_ }, { f2_, 0, -f2_ }, { -f2_, 0, -f2_ },
        { 0, f2_, f2_ }, { 0, -f2_, f2_ }, { 0, -f2_, -f2_ }, { 0, f2_, -f2_ } };
    float shadow = 0.0f;
    const float diskRadius = (1.0f + fdiv_(d.viewDistanceOfPixel, farPlane)) * fdiv_(1.0f, 8.0f);
    const float* cube = cubeArr + (size_t)index * 6 * dim * dim;
    const float lenLw = length_lit(Lw);
    for (int i = 0; i < 20; ++i) {
        const f3 sv = { -(Lw.x + DIRS[i].x * diskRadius), -(Lw.y + DIRS[i].y * diskRadius), -(Lw.z + DIRS[i].z * diskRadius) };
        const float closest = fetch_cube_point(cube, dim, sv) * farPlane;
        shadow += (lenLw > (closest + d.depthBias) + 0.001f) ? 1.0f : 0.0f;
    }
    shadow = fdiv_(shadow, 20.0f);
    return 1.0f - shadow;
}
// ShadowTestPCF, Lighting.hlsl:177-218
inline float ShadowTestPCF(const ShadowTestPCFData& d, const float* arr, int dim, f2 smDims, int index) {
    const f3 p = { fdiv_(d.lightSpacePos.x, d.lightSpacePos.w), fdiv_(d.lightSpacePos.y, d.lightSpacePos.w), fdiv_(d.lightSpacePos.z, d.lightSpacePos.w) };
    if (p.x < -1.0f || p.x > 1.0f || p.y < -1.0f || p.y > 1.0f || p.z < 0.0f || p.z > 1.0f) return 0.0f;
    const float BIAS = d.depthBias * tan_(acos_(d.NdotL));
    float shadow = 0.0f;
    const f2 texel = { rcp(smDims.x), rcp(smDims.y) };
    const f2 uv = { 0.5f + p.x * 0.5f, 0.5f + p.y * -0.5f };
    const float* slice = arr + (size_t)index * dim * dim;
    for (int x = -2; x <= 2; ++x)
        for (int y = -2; y <= 2; ++y) {
            const float closest = fetch_point_wrap(slice, dim, uv.x + (float)x * texel.x, uv.y + (float)y * texel.y);
            shadow += (p.z - BIAS > closest) ? 1.0f : 0.0f;
        }
    shadow = fdiv_(shadow, 25.0f);
    return 1.0f - shadow;
}
// ShadowTestPCF_Directional, Lighting.hlsl:222-272 (uses the raw depthBias, :263)
inline float ShadowTestPCF_Directional(const ShadowTestPCFData& d, const float* map, int dim, f2 smDims) {
    const f3 p = { fdiv_(d.lightSpacePos.x, d.lightSpacePos.w), fdiv_(d.lightSpacePos.y, d.lightSpacePos.w), fdiv_(d.lightSpacePos.z, d.lightSpacePos.w) };
    if (p.x < -1.0f || p.x > 1.0f || p.y < -1.0f || p.y > 1.0f || p.z < 0.0f || p.z > 1.0f) return 0.0f;
    float shadow = 0.0f;
    const f2 texel = { rcp(smDims.x), rcp(smDims.y) };
    const f2 uv = { 0.5f + p.x * 0.5f, 0.5f + p.y * -0.5f };
    for (int x = -2; x <= 2; ++x)
        for (int y = -2; y <= 2; ++y) {
            const float closest = fetch_point_wrap(map, dim, uv.x + (float)x * texel.x, uv.y + (float)y * texel.y);
            shadow += (p.z - d.depthBias > closest) ? 1.0f : 0.0f;
        }
    shadow = fdiv_(shadow, 25.0f);
    return 1.0f - shadow;
}

// Light illumination functions, Lighting.hlsl:308-345. Each returns BRDF * radiance * NdotL with the scalar factors
// gathered:  radiance * NdotL = (l.color * l.brightness) * w  with the per-light colour cb = color*brightness (a
// loop-invariant a compiler hoists; the product's HOST side precomputes it with the same IEEE multiply) and the
// scalar w = attenuation [* cone] * NdotL.  Result = acc + b * (cb * w) per channel (one mad).
inline f3 light_cb(const VQ_float3& color, float brightness) { return { color.x * brightness, color.y * brightness, color.z * brightness }; }
// acc + b * (cb * w): the caller's `I_total += ...` is fused into the final mad (acc = 0 gives the plain product)
inline f3 lit(f3 acc, f3 b, f3 cb, float w) { return { fma_(b.x, cb.x * w, acc.x), fma_(b.y, cb.y * w, acc.y), fma_(b.z, cb.z * w, acc.z) }; }

// CalculatePointLightIllumination, Lighting.hlsl:308-322
inline f3 CalculatePointLightIllumination(const VQ_PointLight& l, const Surface& s, f3 P, f3 V, f3 acc = { 0, 0, 0 }) {
    const f3 Lw = to3(l.position);
    const f3 d = sub(Lw, P);
    const float D = length_lit(d);                           // as written (contract v5): D decides the range cull, Wi feeds the GGX denominator
    const f3 Wi = g_arith_dxc ? normalize_lit(d) : div_lit(d, D);   // normalize(Lw - P): literal (Lw - P) / length(Lw - P), one IEEE quotient per component (shares D); dxc: its own rsqrt
    const float rD = rcp(D);
    const float NdotL = saturate(dot(s.N, Wi));
    const float w = (rD * rD) * NdotL;                       // AttenuationBRDF: 1/(D*D) as (1/D)*(1/D) (contract v3, insensitive)
    if (D < l.range) return lit(acc, BRDF(s, Wi, V), light_cb(l.color, l.brightness), w);
    return acc;
}
// CalculateSpotLightIllumination, Lighting.hlsl:323-333 (no range cull)
inline f3 CalculateSpotLightIllumination(const VQ_SpotLight& l, const Surface& s, f3 P, f3 V, f3 acc = { 0, 0, 0 }) {
    const f3 d = sub(to3(l.position), P);
    const float D = length_lit(d);
    const f3 Wi = g_arith_dxc ? normalize_lit(d) : div_lit(d, D);
    const float rD = rcp(D);
    const float cone = SpotlightIntensity(l, P);
    const float NdotL = saturate(dot(s.N, Wi));
    const float w = (cone * (rD * rD)) * NdotL;
    return lit(acc, BRDF(s, Wi, V), light_cb(l.color, l.brightness), w);
}
// CalculateDirectionalLightIllumination, Lighting.hlsl:334-345
inline f3 CalculateDirectionalLightIllumination(const VQ_DirectionalLight& l, const Surface& s, f3 V) {
    const f3 Wi = normalize_lit(neg(to3(l.lightDirection)));
    const float NdotL = saturate(dot(s.N, Wi));
    return lit({ 0, 0, 0 }, BRDF(s, Wi, V), light_cb(l.color, l.brightness), NdotL);
}

// GetHDRIRotationMatrix, Lighting.hlsl:348-358; mul(v, m) = row vector times matrix
struct M3 { float m[3][3]; };
// The matrix depends on the frame only (the shader's own TODO: "pass m with cbuffer"): cos / sin are taken CORRECTLY ROUNDED (double
// libm rounded to float) — the product's host side does the same and hands the kernel the two numbers. The contract's polynomial cos_
// is 1 ulp off at the bench's offset 0.3, which moved the 8-bit filter fraction of two taps next to a sun in the cfg3 band (3 RGBA16F ulps).
inline M3 GetHDRIRotationMatrix(float offs) {
    const float c = (float)std::cos((double)-offs), s = (float)std::sin((double)-offs);
    return { { { c, 0, s }, { 0, 1, 0 }, { -s, 0, c } } };
}
inline f3 mul_v_m(f3 v, const M3& m) {          // as written: products and sums rounded one by one, left to right (contract v5)
    return { (v.x * m.m[0][0] + v.y * m.m[1][0]) + v.z * m.m[2][0],
             (v.x * m.m[0][1] + v.y * m.m[1][1]) + v.z * m.m[2][1],
             (v.x * m.m[0][2] + v.y * m.m[1][2]) + v.z * m.m[2][2] };
}
inline size_t cube_mip_offset_halfs(int res0, int mip) {   // packed [mip][6][r][r] RGBA16F
    size_t off = 0;
    for (int m = 0; m < mip; ++m) { size_t r = (size_t)(res0 >> m); off += 6 * r * r * 4; }
    return off;
}
// CalculateEnvironmentMapIllumination, Lighting.hlsl:360-380 ; _DiffuseOnly :382-395
inline f3 CalculateEnvironmentMapIllumination(const Surface& s, f3 V, int MAX_REFLECTION_LOD, const vqhip_envmap& env, float hdriOffset, bool diffuseOnly) {
    const M3 m = GetHDRIRotationMatrix(hdriOffset);
    const float NdotV = saturate(dot_lit(s.N, V));
    const f3 N = mul_v_m(s.N, m);
    const f4 irr = sample_cube_rgba16f((const uint16_t*)env.diffuse_cube, env.diffuse_res, N);
    if (diffuseOnly)
        return EnvironmentBRDF(NdotV, s.roughness, s.metalness, s.diffuseColor, { irr.x, irr.y, irr.z }, { 0, 0, 0 }, { 0, 0 });
    const f3 R = mul_v_m(reflect_lit(neg(V), s.N), m);
    int MIP_LEVEL = f2i_trunc(s.roughness * (float)MAX_REFLECTION_LOD);
    if (MIP_LEVEL < 0) MIP_LEVEL = 0;
    if (MIP_LEVEL > env.spec_mips - 1) MIP_LEVEL = env.spec_mips - 1;           // sampler clamps the LOD
    const uint16_t* spec = (const uint16_t*)env.specular_cube + cube_mip_offset_halfs(env.spec_res0, MIP_LEVEL);
    const f4 sp = sample_cube_rgba16f(spec, env.spec_res0 >> MIP_LEVEL, R);
    const f2 F0ScaleBias = sample_2d_rg16f_clamp((const uint16_t*)env.brdf_lut, env.lut_size, env.lut_size, NdotV, s.roughness);
    return EnvironmentBRDF(NdotV, s.roughness, s.metalness, s.diffuseColor, { irr.x, irr.y, irr.z }, { sp.x, sp.y, sp.z }, F0ScaleBias);
}

inline f4 mul_M_v(const VQ_matrix& M, f4 v) {   // HLSL mul(M, v) with column-major cbuffer == v * M_cpu
    f4 r;
    float* o = &r.x;
    for (int j = 0; j < 4; ++j)
        o[j] = ((v.x * M.m[0][j] + v.y * M.m[1][j]) + v.z * M.m[2][j]) + v.w * M.m[3][j];      // as written (contract v5)
    return r;
}

// ---- Shaders/ForwardLighting.hlsl:PSMain :284-380 for one G-buffer pixel ---------------------------
inline f4 ShadePixel(f4 g0, f4 g1, f4 g2, f4 g3, const VQ_PerFrameData& F, const VQ_PerViewLightingData& Vw,
                     const VQ_PointLight* extra, int nExtra, const vqhip_envmap* env, const vqhip_shadowmaps* sm) {
    Surface S;
    S.N = { g1.x, g1.y, g1.z }; S.roughness = g1.w;
    S.diffuseColor = { g2.x, g2.y, g2.z }; S.metalness = g2.w;
    S.emissiveColor = { g3.x, g3.y, g3.z }; S.emissiveIntensity = g3.w;
    const float ao = g0.w;
    const f3 P = { g0.x, g0.y, g0.z };                                   // :284
    const f3 cam = to3(Vw.CameraPosition);
    const f3 V = normalize_lit(sub(cam, P));                              // :285
    f3 I = { S.diffuseColor.x * ao + S.emissiveColor.x * S.emissiveIntensity,          // :290-293 as written
             S.diffuseColor.y * ao + S.emissiveColor.y * S.emissiveIntensity,
             S.diffuseColor.z * ao + S.emissiveColor.z * S.emissiveIntensity };
    if (env) {                                                            // :299-306 (NULL == NullCubemap: adds 0)
        const f3 e = CalculateEnvironmentMapIllumination(S, V, f2i_trunc(Vw.MaxEnvMapLODLevels), *env, F.fHDRIOffsetInRadians,
                                                         Vw.EnvironmentMapDiffuseOnlyIllumination != 0);
        I = add(I, e);
    }
    const VQ_SceneLighting& L = F.Lights;
    for (int p = 0; p < L.numPointLights; ++p) I = CalculatePointLightIllumination(L.point_lights[p], S, P, V, I);         // :310-313 (I_total += ...)
    for (int p = 0; p < nExtra; ++p)           I = CalculatePointLightIllumination(extra[p], S, P, V, I);                   // extension (vqhip.h)
    for (int s = 0; s < L.numSpotLights; ++s)  I = CalculateSpotLightIllumination(L.spot_lights[s], S, P, V, I);            // :314-317
    for (int pc = 0; pc < L.numPointCasters; ++pc) {                      // :321-339
        const VQ_PointLight& l = L.point_casters[pc];
        const f3 Lw = sub(to3(l.position), P);
        const float D = length_lit(Lw);
        if (D < l.range) {
            const f3 Ln = normalize_lit(Lw);
            ShadowTestPCFData d{};
            d.depthBias = l.depthBias;
            d.NdotL = saturate(dot_lit(S.N, Ln));
            d.viewDistanceOfPixel = length_lit(sub(P, cam));
            const f3 c = CalculatePointLightIllumination(l, S, P, V);
            const float sh = OmnidirectionalShadowTestPCF(d, sm->point, sm->point_dim, pc, Lw, l.range);
            I = { fma_(c.x, sh, I.x), fma_(c.y, sh, I.y), fma_(c.z, sh, I.z) };
        }
    }
    for (int sc = 0; sc < L.numSpotCasters; ++sc) {                       // :342-356
        const VQ_SpotLight& l = L.spot_casters[sc];
        const f3 Ln = normalize_lit(sub(to3(l.position), P));
        ShadowTestPCFData d{};
        d.depthBias = l.depthBias;
        d.NdotL = saturate(dot_lit(S.N, Ln));
        d.lightSpacePos = mul_M_v(L.shadowViews[sc], { P.x, P.y, P.z, 1.0f });
        d.viewDistanceOfPixel = length_lit(sub(P, cam));
        const f3 c = CalculateSpotLightIllumination(l, S, P, V);
        const float sh = ShadowTestPCF(d, sm->spot, sm->spot_dim, { F.f2SpotLightShadowMapDimensions.x, F.f2SpotLightShadowMapDimensions.y }, sc);
        I = { fma_(c.x, sh, I.x), fma_(c.y, sh, I.y), fma_(c.z, sh, I.z) };
    }
    {                                                                     // :360-377
        const VQ_DirectionalLight& l = L.directional;
        if (l.enabled) {
            float ShadowingFactor = 1.0f;
            if (l.shadowing) {
                ShadowTestPCFData d{};
                const f3 Ln = normalize_lit(neg(to3(l.lightDirection)));
                d.lightSpacePos = mul_M_v(L.shadowViewDirectional, { P.x, P.y, P.z, 1.0f });
                d.NdotL = saturate(dot_lit(S.N, Ln));
                d.depthBias = l.depthBias;
                ShadowingFactor = ShadowTestPCF_Directional(d, sm->directional, sm->dir_dim,
                                      { F.f2DirectionalLightShadowMapDimensions.x, F.f2DirectionalLightShadowMapDimensions.y });
            }
            const f3 c = CalculateDirectionalLightIllumination(l, S, V);
            I = { fma_(c.x, ShadowingFactor, I.x), fma_(c.y, ShadowingFactor, I.y), fma_(c.z, ShadowingFactor, I.z) };
        }
    }
    return { I.x, I.y, I.z, S.roughness };                                // :380
}

// ---- storage helpers ----------------------------------------------------------------------------
inline int fmt_bpp(int fmt) {
    switch (fmt) { case VQHIP_FMT_RGBA32F: return 16; case VQHIP_FMT_RGBA16F: return 8; case VQHIP_FMT_RGBA8_UNORM: return 4;
                   case VQHIP_FMT_RG16F: return 4; case VQHIP_FMT_RG32F: return 8; }
    return 0;
}
inline void store_px(void* base, size_t idx, int fmt, f4 c) {
    switch (fmt) {
        case VQHIP_FMT_RGBA32F: { float* p = (float*)base + idx * 4; p[0] = c.x; p[1] = c.y; p[2] = c.z; p[3] = c.w; } break;
        case VQHIP_FMT_RGBA16F: { uint16_t* p = (uint16_t*)base + idx * 4; p[0] = f32_to_f16(c.x); p[1] = f32_to_f16(c.y); p[2] = f32_to_f16(c.z); p[3] = f32_to_f16(c.w); } break;
        case VQHIP_FMT_RGBA8_UNORM: { uint8_t* p = (uint8_t*)base + idx * 4; p[0] = f32_to_unorm8(c.x); p[1] = f32_to_unorm8(c.y); p[2] = f32_to_unorm8(c.z); p[3] = f32_to_unorm8(c.w); } break;
        case VQHIP_FMT_RG32F: { float* p = (float*)base + idx * 2; p[0] = c.x; p[1] = c.y; } break;
        case VQHIP_FMT_RG16F: { uint16_t* p = (uint16_t*)base + idx * 2; p[0] = f32_to_f16(c.x); p[1] = f32_to_f16(c.y); } break;
    }
}
inline f4 load_px(const void* base, size_t idx, int fmt) {
    if (fmt == VQHIP_FMT_RGBA32F) { const float* p = (const float*)base + idx * 4; return { p[0], p[1], p[2], p[3] }; }
    const uint16_t* p = (const uint16_t*)base + idx * 4; return load_rgba16f(p);
}

// KERNEL_WEIGHTS for KERNEL_RANGE == 11, Shaders/GaussianBlur.hlsl:109-111
const float KERNEL_WEIGHTS[11] = { 0.224716f, 0.191756f, 0.119146f, 0.053897f, 0.017746f, 0.004252f, 0.000741f, 0.000094f, 0.000009f, 0.000001f, 0.0f };

} // namespace

// =================================================================================================
// C entry points (loaded with ctypes by tests/ and bench.py's cpu_baseline leg)
// =================================================================================================
extern "C" {

int vqo_has_fma(void) { return __builtin_cpu_supports("fma") ? 1 : 0; }
int vqo_max_threads(void) { return omp_get_max_threads(); }

// scalar probes for tests/test_oracle_math.py
float vqo_log2(float x) { return log2_(x); }
float vqo_exp2(float x) { return exp2_(x); }
float vqo_pow(float x, float y) { return pow_(x, y); }
float vqo_sin(float x) { return sin_(x); }
float vqo_cos(float x) { return cos_(x); }
float vqo_tan(float x) { return tan_(x); }
float vqo_asin(float x) { return asin_(x); }
float vqo_acos(float x) { return acos_(x); }
float vqo_atan2(float y, float x) { return atan2_(y, x); }
float vqo_rcp(float x) { return rcp(x); }
float vqo_sqrt(float x) { return sqrt_(x); }
void vqo_math_array(int fn, const float* a, const float* b, float* out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        float x = a[i], y = b ? b[i] : 0.0f, r = 0;
        switch (fn) { case 0: r = log2_(x); break; case 1: r = exp2_(x); break; case 2: r = pow_(x, y); break; case 3: r = sin_(x); break;
                      case 4: r = cos_(x); break; case 5: r = tan_(x); break; case 6: r = asin_(x); break; case 7: r = acos_(x); break;
                      case 8: r = atan2_(x, y); break; case 9: r = rcp(x); break; case 10: r = sqrt_(x); break; case 11: r = rsqrt(x); break; }
        out[i] = r;
    }
}
void vqo_f32_to_f16(const float* in, uint16_t* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = f32_to_f16(in[i]); }
void vqo_f16_to_f32(const uint16_t* in, float* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = f16_to_f32(in[i]); }
void vqo_f32_to_unorm8(const float* in, uint8_t* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = f32_to_unorm8(in[i]); }

void vqo_brdf(const float* N, float roughness, const float* albedo, float metal, const float* Wi, const float* V, float* out3) {
    Surface s{}; s.N = { N[0], N[1], N[2] }; s.roughness = roughness; s.diffuseColor = { albedo[0], albedo[1], albedo[2] }; s.metalness = metal;
    f3 r = BRDF(s, { Wi[0], Wi[1], Wi[2] }, { V[0], V[1], V[2] });
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
void vqo_cube_texel_dir(int face, int x, int y, int res, float* out3) { f3 d = cube_texel_dir(face, x, y, res); out3[0] = d.x; out3[1] = d.y; out3[2] = d.z; }
int  vqo_cube_face_uv(const float* d, float* uv) { return cube_face_uv({ d[0], d[1], d[2] }, &uv[0], &uv[1]); }
void vqo_cube_edge_neighbor(int f, int i, int j, int N, int* out3) { cube_edge_neighbor(f, i, j, N, &out3[0], &out3[1], &out3[2]); }
void vqo_sample_cube_rgba16f(const uint16_t* cube, int N, const float* dir, float* out4) {
    f4 c = sample_cube_rgba16f(cube, N, { dir[0], dir[1], dir[2] }); out4[0] = c.x; out4[1] = c.y; out4[2] = c.z; out4[3] = c.w;
}
void vqo_sample_cube_lod_rgba16f(const uint16_t* cube, int res0, int nMips, const float* dir, float lod, float* out4) {
    f4 c = sample_cube_lod_rgba16f(cube, res0, nMips, { dir[0], dir[1], dir[2] }, lod); out4[0] = c.x; out4[1] = c.y; out4[2] = c.z; out4[3] = c.w;
}
void vqo_sample_2d_rg16f_clamp(const uint16_t* tex, int W, int H, float u, float v, float* out2) { f2 r = sample_2d_rg16f_clamp(tex, W, H, u, v); out2[0] = r.x; out2[1] = r.y; }
void vqo_sample_material_tex(const vqhip_texture2d* t, const float* uv, const float* ddx, const float* ddy, float bias, float* out4) {
    f4 c = sample_material_tex(*t, { uv[0], uv[1] }, { ddx[0], ddx[1] }, { ddy[0], ddy[1] }, bias); out4[0] = c.x; out4[1] = c.y; out4[2] = c.z; out4[3] = c.w;
}
float vqo_fetch_r8_point_wrap(const uint8_t* tex, int W, int H, float u, float v) { return fetch_r8_point_wrap(tex, W, H, u, v); }
void vqo_direction_to_equirect_uv(const float* d, float* uv) { f2 r = DirectionToEquirectUV({ d[0], d[1], d[2] }); uv[0] = r.x; uv[1] = r.y; }
void vqo_sample_equirect_lod(const float* chain, int w0, int h0, int nMips, float u, float v, float lod, float* out4) {
    f4 c = sample_equirect_lod(chain, w0, h0, nMips, u, v, lod); out4[0] = c.x; out4[1] = c.y; out4[2] = c.z; out4[3] = c.w;
}

// ForwardLighting.hlsl:PSMain :284-380 over a G-buffer (all pointers HOST memory here; `env`/`sm` members too)
int vqo_forward_lighting(const vqhip_gbuffer* gb, const VQ_PerFrameData* perFrame, const VQ_PerViewLightingData* perView,
                         const VQ_PointLight* extra, int nExtra, const vqhip_envmap* env, const vqhip_shadowmaps* sm,
                         void* out, int out_pitch, int outFmt, int nthreads) {
    if (!gb || !perFrame || !perView || !out) return -1;
    if (outFmt != VQHIP_FMT_RGBA32F && outFmt != VQHIP_FMT_RGBA16F) return -3;
    {   // casters without their maps: refused like the product refuses them (capi.hip: "shadow casters present but sm is NULL"), not dereferenced
        const auto& L = perFrame->Lights;
        const bool dirCaster = L.directional.enabled && L.directional.shadowing;
        if ((L.numPointCasters > 0 || L.numSpotCasters > 0 || dirCaster) && !sm) return -1;
        if (sm && ((L.numPointCasters > 0 && (!sm->point || sm->point_dim <= 0)) || (L.numSpotCasters > 0 && (!sm->spot || sm->spot_dim <= 0)) ||
                   (dirCaster && (!sm->directional || sm->dir_dim <= 0)))) return -1;
    }
    const int W = gb->width, H = gb->height, pitch = gb->row_pitch_px;
    const f4* g0 = (const f4*)gb->gb0; const f4* g1 = (const f4*)gb->gb1; const f4* g2 = (const f4*)gb->gb2; const f4* g3 = (const f4*)gb->gb3;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    #pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t i = (size_t)y * pitch + x;
            f4 c = ShadePixel(g0[i], g1[i], g2[i], g3[i], *perFrame, *perView, extra, nExtra, env, sm);
            store_px(out, (size_t)y * out_pitch + x, outFmt, c);
        }
    return 0;
}

// The other render targets of PSMain (ForwardLighting.hlsl:PSOutput :57-68), per pixel of a G-buffer:
//   outAlbedo  = float4(Surface.diffuseColor, Surface.metalness)                                                   :383  (the G-buffer's gb2, include/vqhip.h)
//   outMotion  = float2(svPositionCurr.xy / svPositionCurr.w - svPositionPrev.xy / svPositionPrev.w)               :387
// svCurr / svPrev: float4 planes (PSInput :49-52), svPitch pixels per row; either output may be NULL; outputs are tightly packed.
int vqo_psmain_extra_targets(const vqhip_gbuffer* gb, const float* svCurr, const float* svPrev, int svPitch,
                             void* outAlbedo, int albedoFmt, void* outMotion, int motionFmt) {
    if (!gb) return -1;
    if (outAlbedo && albedoFmt != VQHIP_FMT_RGBA32F && albedoFmt != VQHIP_FMT_RGBA16F) return -3;
    if (outMotion && ((motionFmt != VQHIP_FMT_RG32F && motionFmt != VQHIP_FMT_RG16F) || !svCurr || !svPrev)) return -3;
    const int W = gb->width, H = gb->height, pitch = gb->row_pitch_px;
    const f4* g2 = (const f4*)gb->gb2;
    const f4* c4 = (const f4*)svCurr; const f4* p4 = (const f4*)svPrev;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            if (outAlbedo) store_px(outAlbedo, (size_t)y * W + x, albedoFmt, g2[(size_t)y * pitch + x]);
            if (outMotion) {
                const f4 c = c4[(size_t)y * svPitch + x], p = p4[(size_t)y * svPitch + x];
                store_px(outMotion, (size_t)y * W + x, motionFmt, { c.x / c.w - p.x / p.w, c.y / c.w - p.y / p.w, 0.0f, 0.0f });
            }
        }
    return 0;
}

// GaussianBlur.hlsl:CSMain_X :120-151 / CSMain_Y :155-187. dir 0 = X, 1 = Y.
// halo_top/halo_bottom: optional rows outside the tile for the Y pass (row-tiled multi-GPU mode).
int vqo_gaussian_blur_pass(const void* in, void* out, int W, int H, int fmt, int dir,
                           const void* halo_top, const void* halo_bottom, int halo_rows, int nthreads) {
    if (fmt != VQHIP_FMT_RGBA32F && fmt != VQHIP_FMT_RGBA16F) return -3;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    #pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            f3 acc = { 0, 0, 0 };
            for (int it = 0; it < 21; ++it) {
                const int off = it - 10, ki = off < 0 ? -off : off;
                f4 s;
                if (dir == 0) {
                    int sx = x + off; sx = sx < 0 ? 0 : (sx > W - 1 ? W - 1 : sx);
                    s = load_px(in, (size_t)y * W + sx, fmt);
                } else {
                    int sy = y + off;
                    if (sy < 0 && halo_top)              s = load_px(halo_top, (size_t)(halo_rows + sy) * W + x, fmt);
                    else if (sy > H - 1 && halo_bottom)  s = load_px(halo_bottom, (size_t)(sy - H) * W + x, fmt);
                    else { sy = sy < 0 ? 0 : (sy > H - 1 ? H - 1 : sy); s = load_px(in, (size_t)sy * W + x, fmt); }
                }
                const float w = KERNEL_WEIGHTS[ki];
                acc = { fma_(s.x, w, acc.x), fma_(s.y, w, acc.y), fma_(s.z, w, acc.z) };   // OutRGB += rgb * KERNEL_WEIGHTS[i], one mad (contract v2)
            }
            store_px(out, (size_t)y * W + x, fmt, { acc.x, acc.y, acc.z, 1.0f });
        }
    return 0;
}

// Tonemapper.hlsl:CSMain :110-151 (+ HDR.hlsl:76-80, :88-97, :110-119)
int vqo_tonemap(const void* in, void* out, int W, int H, const VQ_TonemapperParams* p, int inFmt, int outFmt, int nthreads) {
    if (inFmt != VQHIP_FMT_RGBA32F && inFmt != VQHIP_FMT_RGBA16F) return -3;
    if (outFmt != VQHIP_FMT_RGBA32F && outFmt != VQHIP_FMT_RGBA16F && outFmt != VQHIP_FMT_RGBA8_UNORM) return -3;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    const size_t n = (size_t)W * H;
    #pragma omp parallel for schedule(static) num_threads(nthreads)
    for (size_t i = 0; i < n; ++i) {
        const f4 c = load_px(in, i, inFmt);
        float rgb[3] = { c.x, c.y, c.z }, o[3] = { 0, 0, 0 };
        switch (p->OutputDisplayCurveEnum) {
            case VQ_DISPLAY_CURVE_SRGB:
                for (int k = 0; k < 3; ++k) {
                    float t = div_(rgb[k], rgb[k] + 1.0f);                                   // Tonemap_Reinhard :24-27
                    if (p->ToggleGammaCorrection)                                            // LinearToSRGB HDR.hlsl:76-80
                        t = (t < 0.0031308f) ? 12.92f * t : 1.055f * pow_(abs_(t), (float)(1.0 / 2.4)) - 0.055f;
                    o[k] = t;
                }
                break;
            case VQ_DISPLAY_CURVE_ST2084: {
                const float HDR_Scalar = div_(p->DisplayReferenceBrightnessLevel, 10000.0f);
                float v[3] = { rgb[0], rgb[1], rgb[2] };
                if (p->ContentColorSpaceEnum == VQ_COLOR_SPACE_REC_709) {                    // Rec709ToRec2020 HDR.hlsl:88-97, mul(M, v)
                    const float M[3][3] = { { 0.627402f, 0.329292f, 0.043306f }, { 0.069095f, 0.919544f, 0.011360f }, { 0.016394f, 0.088028f, 0.895578f } };
                    for (int r = 0; r < 3; ++r) v[r] = fma_(M[r][2], rgb[2], fma_(M[r][1], rgb[1], M[r][0] * rgb[0]));
                }
                const float m1 = (float)(2610.0 / 4096.0 / 4), m2 = (float)(2523.0 / 4096.0 * 128), c1 = (float)(3424.0 / 4096.0),
                            c2 = (float)(2413.0 / 4096.0 * 32), c3 = (float)(2392.0 / 4096.0 * 32);
                for (int k = 0; k < 3; ++k) {                                                // LinearToST2084 HDR.hlsl:110-119
                    const float cp = pow_(abs_(v[k] * HDR_Scalar), m1);
                    o[k] = pow_(div_(c1 + c2 * cp, 1.0f + c3 * cp), m2);
                }
            } break;
            case VQ_DISPLAY_CURVE_LINEAR: o[0] = rgb[0]; o[1] = rgb[1]; o[2] = rgb[2]; break;
            default: o[0] = 1; o[1] = 1; o[2] = 0; break;
        }
        store_px(out, i, outFmt, { o[0], o[1], o[2], c.w });
    }
    return 0;
}

// CubemapConvolution.hlsl:CSMain_BRDFIntegration :225-240
int vqo_brdf_lut(void* outRG, int size, int samples, int fmt, int nthreads) {
    if (fmt != VQHIP_FMT_RG16F && fmt != VQHIP_FMT_RG32F) return -3;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    #pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
    for (int y = 0; y < size; ++y)
        for (int x = 0; x < size; ++x) {
            const float u = div_((float)x + 0.5f, (float)size), v = div_((float)y + 0.5f, (float)size);
            const f2 r = IntegrateBRDF(u, v, samples);
            store_px(outRG, (size_t)y * size + x, fmt, { r.x, r.y, 0, 0 });
        }
    return 0;
}
// rows [y0,y1) only — lets tests sample the 1024^2 x 2048 LUT without computing all of it
int vqo_brdf_lut_rows(void* outRG, int size, int samples, int fmt, int y0, int y1, int nthreads) {
    if (fmt != VQHIP_FMT_RG16F && fmt != VQHIP_FMT_RG32F) return -3;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    #pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int y = y0; y < y1; ++y)
        for (int x = 0; x < size; ++x) {
            const float u = div_((float)x + 0.5f, (float)size), v = div_((float)y + 0.5f, (float)size);
            const f2 r = IntegrateBRDF(u, v, samples);
            store_px(outRG, (size_t)(y - y0) * size + x, fmt, { r.x, r.y, 0, 0 });
        }
    return 0;
}

// VQ_DXGI_UTILS::MipImage 16-byte branch, Source/Renderer/Resources/DXGIUtils.cpp:289-317, applied level by
// level as TextureManager::GenerateMips does (TextureManager.cpp:643-738). `chain` holds level 0.
int vqo_mip_level_count(int w, int h) { return mip_level_count(w, h); }
size_t vqo_mip_chain_floats(int w0, int h0, int nMips) { return mip_offset_floats(w0, h0, nMips); }
int vqo_mip_chain_min_rgba32f(float* chain, int w0, int h0, int nMips) {
    for (int l = 1; l < nMips; ++l) {
        const int sw = mip_dim(w0, l - 1), sh = mip_dim(h0, l - 1), dw = mip_dim(w0, l), dh = mip_dim(h0, l);
        const float* src = chain + mip_offset_floats(w0, h0, l - 1);
        float* dst = chain + mip_offset_floats(w0, h0, l);
        for (int y = 0; y < dh; ++y)
            for (int x = 0; x < dw; ++x) {
                const int x0 = 2 * x, y0 = 2 * y, x1 = (2 * x + 1 < sw) ? 2 * x + 1 : sw - 1, y1 = (2 * y + 1 < sh) ? 2 * y + 1 : sh - 1;
                for (int ch = 0; ch < 3; ++ch) {
                    const float a = src[((size_t)y0 * sw + x0) * 4 + ch], b = src[((size_t)y0 * sw + x1) * 4 + ch],
                                c = src[((size_t)y1 * sw + x0) * 4 + ch], d = src[((size_t)y1 * sw + x1) * 4 + ch];
                    // min(a, min(b, min(c, d))) with std::min(p, q) = (q < p) ? q : p, literally: a NaN or a zero in FIRST position wins (NaN: no comparison is true;
                    // +0 against -0: neither is less). Rounds 1-5 wrote (p < q) ? p : q, which differs exactly there (tests/fuzz/fuzz_ibl.py, round 6)
                    const float cd = d < c ? d : c, bcd = cd < b ? cd : b;
                    dst[((size_t)y * dw + x) * 4 + ch] = bcd < a ? bcd : a;
                }
                dst[((size_t)y * dw + x) * 4 + 3] = 1.0f;
            }
    }
    return 0;
}

int vqo_specular_mip_count(int res0) { return mip_level_count(res0, res0) - 1; }   // EnvironmentMapRendering.cpp:63
size_t vqo_cube_halfs(int res0, int nMips) { return cube_mip_offset_halfs(res0, nMips); }

// fp32 sequence of the float-accumulated loop `for (x = 0; x < limit; x += step)` (CubemapConvolution.hlsl:132-136)
static std::vector<float> loop_sequence(float limit, float step) {
    std::vector<float> v;
    for (float x = 0.0f; x < limit; x += step) v.push_back(x);
    return v;
}
int vqo_loop_count(float limit, float step) { return (int)loop_sequence(limit, step).size(); }

// PSMain_DiffuseIrradiance, CubemapConvolution.hlsl:112-163, for one texel direction.
// order 0: HLSL loop order (phi outer, theta inner, one accumulator).
// order 1 (WAVE64): phi index k goes to lane k % 64; each lane accumulates its (phi, all theta) taps in
//          order; the 64 partial sums are combined by the butterfly s[l] += s[l ^ m], m = 32,16,8,4,2,1.
static f3 diffuse_irradiance_texel(f3 dirIn, const float* chain, int w0, int h0, int nMips,
                                   const std::vector<float>& phis, const std::vector<float>& thetas, int order) {
    const f3 N = normalize(dirIn);
    f3 up = { 0, 1, 0 };
    const f3 right = normalize(cross(up, N));
    up = normalize(cross(N, right));
    const int nLanes = order == 1 ? 64 : 1;
    f3 acc[64]; for (int l = 0; l < 64; ++l) acc[l] = { 0, 0, 0 };
    std::vector<float> sinT(thetas.size()), cosT(thetas.size());
    for (size_t t = 0; t < thetas.size(); ++t) sincos_(thetas[t], &sinT[t], &cosT[t]);
    for (size_t k = 0; k < phis.size(); ++k) {
        float sinPhi, cosPhi; sincos_(phis[k], &sinPhi, &cosPhi);
        f3& a = acc[k % nLanes];
        for (size_t t = 0; t < thetas.size(); ++t) {
            const float sinTheta = sinT[t], cosTheta = cosT[t];
            const f3 ts = { sinTheta * cosPhi, sinTheta * sinPhi, cosTheta };
            f3 sv = { (ts.x * right.x + ts.y * up.x) + ts.z * N.x, (ts.x * right.y + ts.y * up.y) + ts.z * N.y, (ts.x * right.z + ts.y * up.z) + ts.z * N.z };
            sv = normalize(sv);
            const f2 uv = DirectionToEquirectUV(sv);
            const f4 c = sample_equirect_lod(chain, w0, h0, nMips, uv.x, uv.y, 3.0f);
            a = { a.x + (c.x * cosTheta) * sinTheta, a.y + (c.y * cosTheta) * sinTheta, a.z + (c.z * cosTheta) * sinTheta };
        }
    }
    if (order == 1)
        for (int m = 32; m >= 1; m >>= 1) {
            f3 t[64];
            for (int l = 0; l < 64; ++l) t[l] = add(acc[l], acc[l ^ m]);
            for (int l = 0; l < 64; ++l) acc[l] = t[l];
        }
    const float numSamples = (float)(phis.size() * thetas.size());       // numSamples += 1.0f per tap: exact below 2^24
    const float rn = rcp(numSamples);
    return { (PI_ * acc[0].x) * rn, (PI_ * acc[0].y) * rn, (PI_ * acc[0].z) * rn };
}

// all 6 faces (EnvironmentMapRendering.cpp:221-240); optionally only texels [t0,t1) of the 6*res*res list
int vqo_conv_diffuse(const float* chain, int w0, int h0, int nMips, int res, float step, int order, void* outCube, int fmt,
                     long t0, long t1, int nthreads) {
    if (fmt != VQHIP_FMT_RGBA32F && fmt != VQHIP_FMT_RGBA16F) return -3;
    const std::vector<float> phis = loop_sequence(TWO_PI_, step), thetas = loop_sequence(PI_OVER_TWO, step);
    const long total = 6L * res * res;
    if (t1 < 0 || t1 > total) t1 = total;
    if (t0 < 0) t0 = 0;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    #pragma omp parallel for schedule(dynamic, 8) num_threads(nthreads)
    for (long t = t0; t < t1; ++t) {
        const int f = (int)(t / ((long)res * res)), y = (int)((t / res) % res), x = (int)(t % res);
        const f3 r = diffuse_irradiance_texel(cube_texel_dir(f, x, y, res), chain, w0, h0, nMips, phis, thetas, order);
        store_px(outCube, (size_t)t, fmt, { r.x, r.y, r.z, 1.0f });
    }
    return 0;
}

// PSMain_SpecularIrradiance, CubemapConvolution.hlsl:168-223 for one texel.
// order 1 (WAVE64): sample i goes to lane i % 64 (8 samples per lane, increasing i); prefilteredColor and
// totalWeight partial sums are combined by the same butterfly as the diffuse pass.
// optional recorder (per thread) of the equirect fetches of one texel: (uv.x, uv.y, lod) per executed sample, in sample order — tests/golden/make_filterstep_tail.py
struct SpecTapRecorder { float* out; int cap, n; };
static thread_local SpecTapRecorder* t_specTaps = nullptr;
static f3 specular_irradiance_texel(f3 dirIn, float Roughness, float dimX, float dimY, const float* chain, int w0, int h0, int nMips, int order) {
    const f3 N = normalize(dirIn);
    const f3 V = N;
    const uint32_t NUM_SAMPLES = 512;
    const int nLanes = order == 1 ? 64 : 1;
    f4 acc[64]; for (int l = 0; l < 64; ++l) acc[l] = { 0, 0, 0, 0 };     // rgb = prefilteredColor, w = totalWeight
    for (uint32_t i = 0; i < NUM_SAMPLES; ++i) {
        const f2 Xi = Hammersley(i, NUM_SAMPLES);
        const f3 H = ImportanceSampleGGX(Xi, N, Roughness);
        const f3 L = reflect(neg(V), H);
        const float NdotL = saturate(dot(N, L));
        if (NdotL > 0.0f) {
            const float NdotH = saturate(dot(N, H));
            const float HdotV = saturate(dot(H, V));
            const float D = NormalDistributionGGX(NdotH, Roughness);
            const float pdf = div_(D * NdotH, 4.0f * HdotV);
            const float fOmegaS = rcp(max_((float)NUM_SAMPLES * pdf, 0.00001f));
            const float fOmegaP = div_(4.0f * PI_, (6.0f * dimX) * dimY);
            const float fMipLevel = (Roughness == 0.0f) ? 0.0f : max_(0.5f * log2_(div_(fOmegaS, fOmegaP)) + -1.0f, 0.0f);
            const f2 uv = DirectionToEquirectUV(L);
            if (t_specTaps) { SpecTapRecorder* r = t_specTaps; if (r->n < r->cap) { float* o = r->out + 3 * (size_t)r->n; o[0] = uv.x; o[1] = uv.y; o[2] = fMipLevel; } ++r->n; }
            const f4 c = sample_equirect_lod(chain, w0, h0, nMips, uv.x, uv.y, fMipLevel);
            f4& a = acc[i % nLanes];
            a = { a.x + c.x * NdotL, a.y + c.y * NdotL, a.z + c.z * NdotL, a.w + NdotL };
        }
    }
    if (order == 1)
        for (int m = 32; m >= 1; m >>= 1) {
            f4 t[64];
            for (int l = 0; l < 64; ++l) t[l] = { acc[l].x + acc[l ^ m].x, acc[l].y + acc[l ^ m].y, acc[l].z + acc[l ^ m].z, acc[l].w + acc[l ^ m].w };
            for (int l = 0; l < 64; ++l) acc[l] = t[l];
        }
    const float rw = rcp(max_(acc[0].w, 0.0001f));
    return { acc[0].x * rw, acc[0].y * rw, acc[0].z * rw };
}

// all mips and faces (EnvironmentMapRendering.cpp:413-464); packed [mip][6][r][r]
int vqo_conv_specular(const float* chain, int w0, int h0, int nMips, int specRes0, int order, void* outCubeMips, int fmt, int nthreads) {
    if (fmt != VQHIP_FMT_RGBA32F && fmt != VQHIP_FMT_RGBA16F) return -3;
    const int MIPS = mip_level_count(specRes0, specRes0) - 1;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    size_t base = 0;
    for (int mip = 0; mip < MIPS; ++mip) {
        const int r = specRes0 >> mip;
        const float Roughness = div_((float)mip, (float)(MIPS - 1));                 // :432
        const long total = 6L * r * r;
        #pragma omp parallel for schedule(dynamic, 8) num_threads(nthreads)
        for (long t = 0; t < total; ++t) {
            const int f = (int)(t / ((long)r * r)), y = (int)((t / r) % r), x = (int)(t % r);
            const f3 c = specular_irradiance_texel(cube_texel_dir(f, x, y, r), Roughness, (float)w0, (float)h0, chain, w0, h0, nMips, order);
            store_px(outCubeMips, base + (size_t)t, fmt, { c.x, c.y, c.z, 1.0f });
        }
        base += (size_t)total;
    }
    return 0;
}

// The same pass for a LIST of texels of the mip-major cube (flat indices into [mip][face][y][x]): lets the tests sample a 512^2 x 9-mip cube (the engine's default size,
// Data/EngineSettings.ini:11) without evaluating its 2.1 M texels on the CPU. out: n pixels in `fmt`.
int vqo_conv_specular_texels(const float* chain, int w0, int h0, int nMips, int specRes0, int order, const int64_t* texels, int n, void* out, int fmt, int nthreads) {
    if (fmt != VQHIP_FMT_RGBA32F && fmt != VQHIP_FMT_RGBA16F) return -3;
    const int MIPS = mip_level_count(specRes0, specRes0) - 1;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    int bad = 0;
    #pragma omp parallel for schedule(dynamic, 8) num_threads(nthreads)
    for (int k = 0; k < n; ++k) {
        int64_t t = texels[k];
        int mip = 0;
        while (mip < MIPS && t >= 6LL * (specRes0 >> mip) * (specRes0 >> mip)) { t -= 6LL * (specRes0 >> mip) * (specRes0 >> mip); ++mip; }
        if (mip >= MIPS || t < 0) { bad = 1; continue; }
        const int r = specRes0 >> mip;
        const float Roughness = div_((float)mip, (float)(MIPS - 1));                 // :432
        const int f = (int)(t / ((int64_t)r * r)), y = (int)((t / r) % r), x = (int)(t % r);
        const f3 c = specular_irradiance_texel(cube_texel_dir(f, x, y, r), Roughness, (float)w0, (float)h0, chain, w0, h0, nMips, order);
        store_px(out, (size_t)k, fmt, { c.x, c.y, c.z, 1.0f });
    }
    return bad ? -1 : 0;
}

// the equirect fetches of ONE texel (flat index into the mip-major cube), sequential order: taps[k] = (uv.x, uv.y, lod); returns their number, out3 = the texel's rgb (fp32)
int vqo_conv_specular_taps(const float* chain, int w0, int h0, int nMips, int specRes0, int64_t texel, float* taps, int cap, float* out3) {
    const int MIPS = mip_level_count(specRes0, specRes0) - 1;
    int mip = 0;
    int64_t t = texel;
    while (mip < MIPS && t >= 6LL * (specRes0 >> mip) * (specRes0 >> mip)) { t -= 6LL * (specRes0 >> mip) * (specRes0 >> mip); ++mip; }
    if (mip >= MIPS || t < 0) return -1;
    const int r = specRes0 >> mip;
    const int f = (int)(t / ((int64_t)r * r)), y = (int)((t / r) % r), x = (int)(t % r);
    SpecTapRecorder rec = { taps, cap, 0 };
    t_specTaps = &rec;
    const f3 c = specular_irradiance_texel(cube_texel_dir(f, x, y, r), div_((float)mip, (float)(MIPS - 1)), (float)w0, (float)h0, chain, w0, h0, nMips, 0);
    t_specTaps = nullptr;
    out3[0] = c.x; out3[1] = c.y; out3[2] = c.z;
    return rec.n;
}

// VQRenderer::PreFilterEnvironmentMap, EnvironmentMapRendering.cpp:139-486: diffuse -> blur X,Y per face -> specular.
int vqo_envmap_prefilter(const float* chain, int w0, int h0, int nMips, int diffuseRes, float diffuseStep, int specRes0, int order,
                         void* diffuse_unblurred, void* diffuse_blurred, void* specular, int nthreads) {
    std::vector<uint16_t> tmpDiff, tmpBlur((size_t)diffuseRes * diffuseRes * 4);
    void* diff = diffuse_unblurred;
    if (!diff) { tmpDiff.resize((size_t)6 * diffuseRes * diffuseRes * 4); diff = tmpDiff.data(); }
    int rc = vqo_conv_diffuse(chain, w0, h0, nMips, diffuseRes, diffuseStep, order, diff, VQHIP_FMT_RGBA16F, 0, -1, nthreads);
    if (rc) return rc;
    const size_t faceHalfs = (size_t)diffuseRes * diffuseRes * 4;
    for (int face = 0; face < 6; ++face) {
        vqo_gaussian_blur_pass((const uint16_t*)diff + face * faceHalfs, tmpBlur.data(), diffuseRes, diffuseRes, VQHIP_FMT_RGBA16F, 0, nullptr, nullptr, 0, nthreads);
        vqo_gaussian_blur_pass(tmpBlur.data(), (uint16_t*)diffuse_blurred + face * faceHalfs, diffuseRes, diffuseRes, VQHIP_FMT_RGBA16F, 1, nullptr, nullptr, 0, nthreads);
    }
    return vqo_conv_specular(chain, w0, h0, nMips, specRes0, order, specular, VQHIP_FMT_RGBA16F, nthreads);
}

// SSR's environment-map fallback: per pixel, ClassifyReflectionTiles.hlsl:ClassifyTiles :146-152 —
//   if (is_reflective_surface && !is_glossy_reflection) intersection_output.xyz = SampleEnvironmentMap(...) (:78-94); g_intersection_output = it —
// and g_extracted_roughness of CSMain :196. IsReflectiveSurface :59-63 (depth < 1), FFX_DNSR_Reflections_IsGlossyReflection Common.hlsl:108-110,
// FFX_DNSR_Reflections_ScreenSpaceToViewSpace -> InvProjectPosition Common.hlsl:98-104,116-118. Every expression as written (contract v5): the
// whole function runs once per pixel. mul(M, v) reads the row-major XMMATRIX column-major: the row vector v times M_cpu;
// mul(g_envMapRotation, float3) truncates the float4x4 to its upper-left 3x3. UNORM10 decodes as c / 1023 (IEEE quotient).
int vqo_ssr_environment_fallback(const void* scene, int sceneFmt, int scenePitch, const float* depth, int depthPitch, const void* normals, int normalFmt, int normalPitch,
                                 int W, int H, const VQ_SSSRConstants* cb, const vqhip_envmap* env, void* out, int outFmt, int outPitch, uint8_t* outRoughness, int nthreads) {
    if (!scene || !depth || !normals || !cb || !env || !out) return -1;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    if (!scenePitch) scenePitch = W;
    if (!depthPitch) depthPitch = W;
    if (!normalPitch) normalPitch = W;
    if (!outPitch) outPitch = W;
    auto mulM = [](const VQ_matrix& M, f4 v) {
        f4 r;
        float* o = &r.x;
        for (int j = 0; j < 4; ++j) o[j] = ((v.x * M.m[0][j] + v.y * M.m[1][j]) + v.z * M.m[2][j]) + v.w * M.m[3][j];
        return r;
    };
    const float mipCount = (float)cb->envMapSpecularIrradianceCubemapMipLevelCount;
    #pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float roughness = load_px(scene, (size_t)y * scenePitch + x, sceneFmt).w;                      // CSMain :193
            const float z = depth[(size_t)y * depthPitch + x];
            if (outRoughness) outRoughness[(size_t)y * W + x] = f32_to_unorm8(roughness);                         // :196
            f4 result = { 0, 0, 0, 0 };
            if ((z < 1.0f) && !(roughness < cb->roughnessThreshold)) {
                f3 n01;
                if (normalFmt == VQHIP_FMT_RGBA32F) { const float* q = (const float*)normals + ((size_t)y * normalPitch + x) * 4; n01 = { q[0], q[1], q[2] }; }
                else { const uint32_t q = ((const uint32_t*)normals)[(size_t)y * normalPitch + x];
                       n01 = { fdiv_((float)(q & 1023u), 1023.0f), fdiv_((float)((q >> 10) & 1023u), 1023.0f), fdiv_((float)((q >> 20) & 1023u), 1023.0f) }; }
                const float u = ((float)x + 0.5f) * cb->inverseBufferDimensions[0], v = ((float)y + 0.5f) * cb->inverseBufferDimensions[1];   // :79
                const f3 wn = normalize_lit({ 2.0f * n01.x - 1.0f, 2.0f * n01.y - 1.0f, 2.0f * n01.z - 1.0f });                                  // :80
                const float cy = 1.0f - v;                                                                                                       // Common.hlsl:99-100
                const f4 pr = mulM(cb->invProjection, { 2.0f * u - 1.0f, 2.0f * cy - 1.0f, z, 1.0f });
                const f3 ray = { fdiv_(pr.x, pr.w), fdiv_(pr.y, pr.w), fdiv_(pr.z, pr.w) };
                const f3 dirV = normalize_lit(ray);                                                                                              // :84
                const f4 nv4 = mulM(cb->view, { wn.x, wn.y, wn.z, 0.0f });                                                                       // :85
                const f3 nV = { nv4.x, nv4.y, nv4.z };
                const f3 Rv = reflect_lit(dirV, nV);                                                                                             // :86
                const f4 rw = mulM(cb->invView, { Rv.x, Rv.y, Rv.z, 0.0f });                                                                     // :87
                const VQ_matrix& R = cb->envMapRotation;
                const f3 d = { (rw.x * R.m[0][0] + rw.y * R.m[1][0]) + rw.z * R.m[2][0], (rw.x * R.m[0][1] + rw.y * R.m[1][1]) + rw.z * R.m[2][1],
                               (rw.x * R.m[0][2] + rw.y * R.m[1][2]) + rw.z * R.m[2][2] };
                const f4 pre = sample_cube_lod_rgba16f((const uint16_t*)env->specular_cube, env->spec_res0, env->spec_mips, d, roughness * (mipCount - 1.0f));   // :89
                const float NdotV = saturate(dot_lit(nV, neg(dirV)));                                                                            // :90
                const f2 sb = sample_2d_rg16f_clamp((const uint16_t*)env->brdf_lut, env->lut_size, env->lut_size, NdotV, roughness);             // :92
                const f3 c = EnvironmentBRDF(NdotV, roughness, 1.0f, { 0, 0, 0 }, { 0, 0, 0 }, { pre.x, pre.y, pre.z }, sb);                     // :93
                result = { c.x, c.y, c.z, 0.0f };
            }
            store_px(out, (size_t)y * outPitch + x, outFmt, result);                                                                             // :153
        }
    return 0;
}

// Skydome.hlsl:VSMain/PSMain :39-56 as drawn at SceneRendering.cpp:1822-1850. CubemapLookDirection =
// normalize(position) is linear on each face of the camera-centred cube, so its interpolant is parallel to the pixel's view ray:
// dir = forward + (ndc.x*tanHalfFovX)*right + (ndc.y*tanHalfFovY)*up (one mad per term), then PSMain:
// uv = DirectionToEquirectUV(normalize(dir)); SampleLevel(TRILINEAR_WRAP, uv, 0) == bilinear WRAP of level 0; alpha 1.
// coverage_ip2 (nullable): plane ip2 of vqhip_interpolants — only pixels with material index < 0 are written.
int vqo_skydome(const float* equirect0, int w0, int h0, const VQ_SkydomeParams* sp, const float* coverage_ip2, int cov_pitch,
                void* color, int W, int H, int pitch, int fmt, int nthreads) {
    if (!equirect0 || !sp || !color) return -1;
    if (fmt != VQHIP_FMT_RGBA32F && fmt != VQHIP_FMT_RGBA16F) return -3;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    #pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            if (coverage_ip2) { int32_t idx; std::memcpy(&idx, coverage_ip2 + ((size_t)y * cov_pitch + x) * 4 + 3, 4); if (idx >= 0) continue; }
            const float nx = div_(2.0f * ((float)x + 0.5f), (float)W) - 1.0f, ny = 1.0f - div_(2.0f * ((float)y + 0.5f), (float)H);
            const float a = nx * sp->tanHalfFovX, b = ny * sp->tanHalfFovY;
            const f3 d = { fma_(b, sp->up.x, fma_(a, sp->right.x, sp->forward.x)),
                           fma_(b, sp->up.y, fma_(a, sp->right.y, sp->forward.y)),
                           fma_(b, sp->up.z, fma_(a, sp->right.z, sp->forward.z)) };
            const f2 uv = DirectionToEquirectUV(normalize(d));
            const f4 c = sample_2d_rgba32f_wrap(equirect0, w0, h0, uv.x, uv.y);
            store_px(color, (size_t)y * pitch + x, fmt, { c.x, c.y, c.z, 1.0f });
        }
    return 0;
}

void vqo_set_fresnel_pow(int expLog) { g_pow5ExpLog = expLog ? 1 : 0; }
// 0 = literal reading of dot / normalize / length / reflect (default), 1 = the DXC reading (vqo_math.h)
void vqo_set_arithmetic(int dxc) { g_arith_dxc = dxc ? 1 : 0; }
int vqo_get_arithmetic(void) { return g_arith_dxc; }
// normalize() of n 3-vectors in the current reading (tests: Surface.N = normalize(In.WorldSpaceNormal) at the G-buffer boundary, ForwardLighting.hlsl:264)
void vqo_normalize_lit_array(const float* v, float* out, size_t n) {
    for (size_t i = 0; i < n; ++i) { const f3 r = normalize_lit({ v[3 * i], v[3 * i + 1], v[3 * i + 2] }); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; }
}
// rsqrt_cr's definition (float)(1.0 / sqrt((double)x)) against the same quotient in x87 extended precision (64-bit significand: its rounding to binary32
// is the correct rounding unless the true value lies within 2^-63 of a rounding boundary) for EVERY significand and both exponent parities — rsqrt of
// 4^k x is 2^-k rsqrt(x) exactly, so two binades cover all normal inputs. Returns the number of disagreements (expected 0).
long vqo_rsqrt_cr_check(void) {
    long bad = 0;
    #pragma omp parallel for reduction(+ : bad) schedule(static)
    for (long i = 0; i < (1L << 24); ++i) {
        const float x = u2f(0x3f800000u + (uint32_t)i);                         // [1, 4): two binades
        const float want = (float)(1.0L / sqrtl((long double)x));
        if (f2u(rsqrt_cr(x)) != f2u(want)) ++bad;
    }
    return bad;
}
void vqo_rsqrt_cr_array(const float* x, float* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = rsqrt_cr(x[i]); }

// Unlit.hlsl:PSMain :58-61 (light gizmo meshes, SceneRendering.cpp:1787-1819) over the engine's coverage plane: ip2.w == -(2+k) -> colors[k]
int vqo_unlit_composite(const float* coverage_ip2, int cov_pitch, const float* colors, int numColors, void* color, int W, int H, int pitch, int fmt) {
    if (!coverage_ip2 || !color || (numColors > 0 && !colors)) return -1;
    if (fmt != VQHIP_FMT_RGBA32F && fmt != VQHIP_FMT_RGBA16F) return -3;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int32_t idx; std::memcpy(&idx, coverage_ip2 + ((size_t)y * cov_pitch + x) * 4 + 3, 4);
            if (idx > -2) continue;
            const long k = -2L - (long)idx;
            if (k >= numColors) continue;
            store_px(color, (size_t)y * pitch + x, fmt, { colors[4 * k], colors[4 * k + 1], colors[4 * k + 2], colors[4 * k + 3] });
        }
    return 0;
}

} // extern "C"
