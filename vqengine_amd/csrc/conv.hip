// conv.hip — load-time image-based-lighting kernels for gfx950:
//   k_brdf_lut      == Shaders/CubemapConvolution.hlsl:CSMain_BRDFIntegration :225-240 + BRDF.hlsl:IntegrateBRDF :239-283
//   k_mip_min       == VQ_DXGI_UTILS::MipImage 16-byte branch, Source/Renderer/Resources/DXGIUtils.cpp:289-317
//   k_conv_diffuse  == CubemapConvolution.hlsl:PSMain_DiffuseIrradiance :112-163
//   k_conv_specular == CubemapConvolution.hlsl:PSMain_SpecularIrradiance :168-223
// These are VALU/transcendental-bound, their buffers are cache resident. The cube
// rasterisation (VS/GS + 6 draws, EnvironmentMapRendering.cpp:221-240,413-464) is replaced by the closed
// form texel -> direction (vq_sampling.h:cube_texel_dir).
// Summation order: SEQUENTIAL = the HLSL loop order, one lane per texel; WAVE64 = one wave per texel,
// lane l takes taps l, l+64, ... in order and the 64 partial sums are combined with the xor butterfly
// 32,16,8,4,2,1 (wavefront shuffles). The oracle implements both orders; see DESIGN.md.
#include "vq_internal.h"
#include "vq_devmath.h"
#include "vq_sampling.h"
#include <cstdlib>
#include <cstring>

using namespace vqd;

namespace {

constexpr float PI_      = 3.14159265359f;
constexpr float TWO_PI_  = 6.28318530718f;
constexpr float EPSILON_ = 0.000000000001f;

VQD float2 DirectionToEquirectUV(f3 v) {                        // ShadingMath.hlsl:70-80
    float ux = atan2_(v.z, v.x), uy = asin_(-v.y);
    ux = div_(ux, -TWO_PI_); uy = div_(uy, PI_);
    return make_float2(ux + 0.5f, uy + 0.5f);
}
VQD float RadicalInverse_VdC(uint32_t bits) {                   // ShadingMath.hlsl:87-95
    bits = __builtin_bitreverse32(bits);                        // the five swap steps == a full 32-bit reversal
    return (float)bits * 2.3283064365386963e-10f;
}
VQD float NormalDistributionGGX(float NdotH, float roughness) { // BRDF.hlsl:65-79
    const float a = roughness * roughness;
    const float a2 = a * a;
    const float nh2 = NdotH * NdotH;
    const float t = fma_(nh2, a2 - 1.0f, 1.0f);                 // one mad (contract v2 tree of BRDF.hlsl:76)
    const float denom = PI_ * (t * t);
    if (denom < EPSILON_) return 1.0f;
    return div_(a2, denom);
}
VQD float G1_env(f3 N, f3 V, float roughness) {                 // Geometry_Smiths_SchlickGGX_EnvironmentMap, BRDF.hlsl:100-115
    const float k = div_(roughness * roughness, 2.0f);
    const float NV = max_(0.0f, dot(N, V));
    return div_(NV, fma_(NV, 1.0f - k, k) + 0.0001f);           // (NV*(1-k) + k) as one mad
}
// ImportanceSampleGGX, BRDF.hlsl:217-238, with sin/cos(phi) supplied by the caller — in its two halves: the tangent-space half vector, a function
// of the sample and the roughness only (:219-227), and its rotation into the frame of N (:229-237). Kernels whose samples / roughness are shared by
// a whole block evaluate the first half once per block (k_brdf_lut, k_conv_specular_all); the composition is the function as written.
VQD f3 ggx_sample_tangent(float Xiy, float sinPhi, float cosPhi, float roughness) {
    const float a = roughness * roughness;
    const float cosTheta = sqrt_(fdiv_(1.0f - Xiy, 1.0f + (a * a - 1.0f) * Xiy));
    const float sinTheta = sqrt_(1.0f - cosTheta * cosTheta);
    return mk3(cosPhi * sinTheta, sinPhi * sinTheta, cosTheta);
}
struct TangentFrame { f3 tangent, bitangent; };
VQD TangentFrame tangent_frame(f3 N) {
    const f3 up = abs_(N.z) < 0.999f ? mk3(0, 0, 1) : mk3(1, 0, 0);
    TangentFrame fr;
    fr.tangent = normalize(cross(up, N));
    fr.bitangent = cross(N, fr.tangent);
    return fr;
}
VQD f3 tangent_to_world(f3 H, const TangentFrame& fr, f3 N) {
    const f3 s = mk3((fr.tangent.x * H.x + fr.bitangent.x * H.y) + N.x * H.z,
                     (fr.tangent.y * H.x + fr.bitangent.y * H.y) + N.y * H.z,
                     (fr.tangent.z * H.x + fr.bitangent.z * H.y) + N.z * H.z);
    return normalize(s);
}
VQD f3 ImportanceSampleGGX(float Xiy, float sinPhi, float cosPhi, f3 N, float roughness) {
    return tangent_to_world(ggx_sample_tangent(Xiy, sinPhi, cosPhi, roughness), tangent_frame(N), N);
}

// ---- BRDF integration LUT ----------------------------------------------------------------------
// Block = 256 texels of one row, i.e. ONE roughness. The half vector of sample i, ImportanceSampleGGX(Xi_i, N = +Z, roughness)
// (BRDF.hlsl:217-238: an IEEE division, two square roots, the tangent frame, a normalize — more than half of IntegrateBRDF's
// arithmetic), depends on the row and the sample only, not on the texel: the block computes each H ONCE (lane j takes samples j,
// j + 256 of a 512-sample chunk — the same function on the same inputs, hence the same bits as the per-texel evaluation) and
// every lane reads it back as an LDS broadcast. Per texel and sample that leaves reflect, one normalize, G, the Fresnel power
// and the two accumulations. G1_env(N, V), the light-independent half of G, is hoisted by hand (LLVM does not move the
// reciprocal's rare-path branch out of the loop).
//
// One sample of IntegrateBRDF (BRDF.hlsl:250-281) for a texel with V = (vx, 0, vz), vz = NdotV. FAST drops what cannot matter when
//   (a) |H|^2 in [0.98, 1.02] and H.z >= 2^-60   (checked per sample when the chunk's table is built: block-uniform)
//   (b) NdotV in [2^-60, 1], |V|^2 in [0.98, 1.02] (checked per lane; a lane outside takes the general form for every chunk)
// hold — always, for the sizes and sample counts the engine uses; the general form is what runs otherwise, and where both apply
// they give identical bits:
//   * Lraw = reflect(-V, H) then has |Lraw|^2 in [0.9, 1.2]: normalize's square root and reciprocal are inside the exhaustively
//     validated domains of sqrt_newton / rcp_newton (vq_devmath.h), no range tests needed; L is finite
//   * dot(N, L) with N = (0, 0, 1) is fma(1, L.z, fma(0, L.y, 0 * L.x)) = L.z + (+-0) = L.z for the L.z > 0 it is used with and
//     finite L.x, L.y: G1_env(N, L)'s NV is L.z itself, and L.x, L.y are never formed
//   * the operand of G1_env(N, L)'s reciprocal, fma(NV, 1-k, k) + 1e-4 with k = roughness^2 / 2 in [0, 1/2] and NV in (0, 1 + 2^-22],
//     lies in [1e-4, 1.51]; that of G_Vis's, NdotH * NdotV, in [2^-120, 1.01]: both reciprocals are normal numbers
//   * the V.y = 0 terms of dot(V, H) and dot(H, -V) add +-0 to a partial sum that is followed by +- vz * H.z != 0: dropping them
//     changes at most the sign of a zero that the next fma absorbs
template <bool FAST>
VQD void lut_sample(const float4 hv, f3 V, float NdotV, float k, float omk, float G1V, float roughness, int p5ExpLog, float& F0Scale, float& F0Bias) {
    const f3 H = mk3(hv.x, hv.y, hv.z);
    const float NdotH = hv.w;                                   // max(H.z, 0), from the table
    if (FAST) {
        const float t = 2.0f * fma_(H.z, -V.z, H.x * -V.x);     // reflect(-V, H): i - n * (2 dot(n, i)), i = -V
        const f3 Lr = mk3(-V.x - H.x * t, -(H.y * t), -V.z - H.z * t);
        float r; (void)sqrt_rcp_newton(dot(Lr, Lr), &r);     // == rcp_newton(sqrt_newton(.)) bit for bit, from one quarter-rate instruction (vq_devmath.h)
        const float Lz = Lr.z * r;
        if (max_(Lz, 0.0f) > 0.0f) {
            const float VdotH = max_(fma_(V.z, H.z, V.x * H.x), 0.0f);
            const float G1L = Lz * rcp_newton(fma_(Lz, omk, k) + 0.0001f);
            const float G = G1V * G1L;
            const float G_Vis = max_((G * VdotH) * rcp_newton(NdotH * NdotV), 0.0001f);
            const float Fc = p5ExpLog ? pow5_explog(1.0f - VdotH) : pow5(1.0f - VdotH);
            F0Scale += (1.0f - Fc) * G_Vis;
            F0Bias += Fc * G_Vis;
        }
    } else {
        const f3 N = mk3(0.0f, 0.0f, 1.0f);
        const f3 L = normalize(reflect(neg(V), H));
        const float NdotL = max_(L.z, 0.0f);
        const float VdotH = max_(dot(V, H), 0.0f);
        if (NdotL > 0.0f) {
            const float G = G1V * G1_env(N, L, roughness);
            const float G_Vis = max_(div_(G * VdotH, NdotH * NdotV), 0.0001f);
            const float Fc = p5ExpLog ? pow5_explog(1.0f - VdotH) : pow5(1.0f - VdotH);
            F0Scale += (1.0f - Fc) * G_Vis;
            F0Bias += Fc * G_Vis;
        }
    }
}

template <int FMT>
__global__ __launch_bounds__(256) void k_brdf_lut(void* __restrict__ out, int size, int samples, int p5ExpLog, int allowFast) {
    __shared__ float4 sH[512];
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    const float NdotV = div_((float)x + 0.5f, (float)size);     // CubemapConvolution.hlsl:233-236
    const float roughness = div_((float)y + 0.5f, (float)size);
    const f3 V = mk3(sqrt_(1.0f - NdotV * NdotV), 0.0f, NdotV);
    const f3 N = mk3(0.0f, 0.0f, 1.0f);
    const float rcount = rcp((float)samples);
    const float G1V = G1_env(N, V, roughness);
    const float k = div_(roughness * roughness, 2.0f), omk = 1.0f - k;      // G1_env's k, 1 - k
    const float vv = dot(V, V);
    const bool laneBad = !allowFast | !((NdotV >= 0x1p-60f) & (NdotV <= 1.0f) & (vv >= 0.98f) & (vv <= 1.02f) & (k >= 0.0f) & (k <= 0.5f));
    float F0Scale = 0.0f, F0Bias = 0.0f;
    for (int base = 0; base < samples; base += 512) {
        __syncthreads();
        bool bad = laneBad;
        for (int j = threadIdx.x; j < 512; j += 256) {
            const uint32_t i = (uint32_t)(base + j);
            const float Xix = (float)i * rcount;                // Hammersley, ShadingMath.hlsl:119-127
            float sp, cp; sincos_((2.0f * PI_) * Xix, &sp, &cp);
            const f3 H = ImportanceSampleGGX(RadicalInverse_VdC(i), sp, cp, N, roughness);
            sH[j] = make_float4(H.x, H.y, H.z, max_(H.z, 0.0f));
            const float hh = dot(H, H);
            if (base + j < samples) bad |= !((hh >= 0.98f) & (hh <= 1.02f) & (H.z >= 0x1p-60f));
        }
        const bool general = __syncthreads_or(bad) != 0;        // block-uniform choice of the loop form (also the barrier behind the table)
        const int n = min(512, samples - base);
        if (!general) for (int j = 0; j < n; ++j) lut_sample<true>(sH[j], V, NdotV, k, omk, G1V, roughness, p5ExpLog, F0Scale, F0Bias);
        else          for (int j = 0; j < n; ++j) lut_sample<false>(sH[j], V, NdotV, k, omk, G1V, roughness, p5ExpLog, F0Scale, F0Bias);
    }
    if (x < size) store_px<FMT>(out, (size_t)y * size + x, make_float4(F0Scale * rcount, F0Bias * rcount, 0, 0));
}

// ---- min-filter mip ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mip_min(const float4* __restrict__ src, float4* __restrict__ dst, int sw, int sh, int dw, int dh) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    const int x0 = 2 * x, y0 = 2 * y, x1 = min(2 * x + 1, sw - 1), y1 = min(2 * y + 1, sh - 1);
    const float4 a = src[(size_t)y0 * sw + x0], b = src[(size_t)y0 * sw + x1], c = src[(size_t)y1 * sw + x0], d = src[(size_t)y1 * sw + x1];
    auto mn = [](float p, float q) { return q < p ? q : p; };   // std::min
    dst[(size_t)y * dw + x] = make_float4(mn(a.x, mn(b.x, mn(c.x, d.x))), mn(a.y, mn(b.y, mn(c.y, d.y))), mn(a.z, mn(b.z, mn(c.z, d.z))), 1.0f);
}

VQD float wave_xor_add(float v, int m) { return v + __shfl_xor(v, m, 64); }

// ---- diffuse irradiance ------------------------------------------------------------------------------
// One tap of PSMain_DiffuseIrradiance (:146-157) from the rotated tangent-space sample vector: normalize, DirectionToEquirectUV,
// SampleLevel(uv, 3). General form: every operation with its range tests and special cases (vq_devmath.h / vq_sampling.h).
VQD f3 diffuse_tap_general(f3 sv, const float4* chain, int w0, int h0, int nMips) {
    sv = normalize(sv);
    const float2 uv = DirectionToEquirectUV(sv);
    const float4 c = sample_equirect_lod(chain, w0, h0, nMips, uv.x, uv.y, 3.0f);                                // mipLevel = 3 :155-157
    return mk3(c.x, c.y, c.z);
}
// Fast form of the same tap: the SAME operations in the same order with (a) the range tests of sqrt_/rcp dropped where the operand's
// range is known, (b) the branches of atan_/asin_ turned into selects around ONE reciprocal / square root (the untaken side's operand
// substituted before the operation: -1 * rcp(x) is -rcp(x), the square root of a lane that does not use it is discarded), (c) the mip
// level, its base and size resolved by the host (lod 3.0 has fraction 0: one level, vq_sampling.h:280), 32-bit byte offsets.
// Preconditions, established outside the tap loop (k_conv_diffuse): |sv|^2 in [0.9, 1.1] (orthonormal frame, unit table entries), level
// dimensions powers of two. What the selects cannot express sets `special`, and the caller redoes the tap of the WHOLE WAVE with the general
// form (wave-uniform branch, rare): a zero or sub-2^-100 x or a zero z (atan2_'s axis cases; keeps y * rcp(x) finite with rcp(x) normal),
// |y| >= 1 (asin_'s pole / NaN cases; keeps 0.5 (1 - |y|) >= 2^-25 inside sqrt_newton's domain). Where it applies it returns the bits of
// diffuse_tap_general (tests/test_gpu_conv_forms.py compares whole cubes of the two forms).
constexpr float kInvNegTwoPi = 1.0f / -TWO_PI_;                  // == rcp(-TWO_PI_): rcp is the correctly rounded quotient, as is the constant division
constexpr float kInvPi       = 1.0f / PI_;
// DirectionToEquirectUV (ShadingMath.hlsl:70-80) of a direction of length ~1, branch-free: the SAME operations as atan2_ / asin_ / div_ with their branches turned into
// selects around ONE reciprocal / square root (see diffuse_tap_fetch). What the selects cannot express sets `special` (the caller redoes the tap in the general form): a zero
// or sub-2^-100 x or a zero z (atan2_'s axis cases; keeps y * rcp(x) finite with rcp(x) normal), |y| >= 1 or NaN (asin_'s pole / NaN cases).
VQD void equirect_uv_fast(f3 sv, float& u, float& v, bool& special) {
    // atan2_(sv.z, sv.x)
    const float ay = sv.z, axx = sv.x;
    special |= !(abs_(axx) >= 0x1p-100f) | (ay == 0.0f);
    const float q = ay * rcp_newton(axx);                                                                         // div_(y, x)
    const float aq = abs_(q);
    const bool big = aq > 2.414213562373095f, mid = aq > 0.4142135623730950f;                                     // mid includes big
    const float xr0 = (big ? -1.0f : aq - 1.0f) * rcp_newton(big ? aq : aq + 1.0f);                               // -rcp(x) | div_(x - 1, x + 1)
    const float xr = mid ? xr0 : aq;
    const float y0 = big ? 1.5707963267948966192f : (mid ? 0.7853981633974483096f : 0.0f);
    const float z = xr * xr;
    float p = 8.05374449538e-2f;
    p = fma_(p, z, -1.38776856032E-1f);
    p = fma_(p, z,  1.99777106478E-1f);
    p = fma_(p, z, -3.33329491539E-1f);
    float at = y0 + fma_(p * z, xr, xr);
    at = (q < 0.0f) ? -at : at;
    const float PI_F = 3.14159265358979323846f;
    const float w = (axx < 0.0f) ? ((ay < 0.0f) ? -PI_F : PI_F) : 0.0f;
    const float ux = w + at;
    // asin_(-sv.y)
    const float sx = -sv.y, sa = abs_(sx);
    special |= !(sa < 1.0f);
    const bool sbig = sa > 0.5f;
    const float sz = sbig ? 0.5f * (1.0f - sa) : sa * sa;
    const float ss = sbig ? sqrt_newton(sz) : sa;
    const float st = asin_poly(ss, sz);
    float uy = sbig ? 1.5707963267948966192f - (st + st) : st;
    uy = (sx < 0.0f) ? -uy : uy;
    u = ux * kInvNegTwoPi + 0.5f; v = uy * kInvPi + 0.5f;                                             // ShadingMath.hlsl:76-79
}
struct DiffuseLevel { const char* tex; int W, H, rowShift; float W256, H256; const char* rec; };   // the sampled level: base, size, log2(W) + 4 (byte offset of a row), 256 W, 256 H
// the tap in two halves: address arithmetic + the four gathers, then the blend (a loop that requests tap t+1 before it blends tap t was measured: 93
// VGPRs, 5 waves per SIMD instead of 7, 7.74 against 7.61 ms — the gathers are throughput-, not latency-bound; profiles/r3i_conv_kernels.md)
struct TapTexels { float4 c00, c10, c01, c11; float wx, wy; };
template <bool REC = false>
VQD TapTexels diffuse_tap_fetch(f3 sv, const DiffuseLevel& lv, bool& special) {
    { float r; (void)sqrt_rcp_newton(dot(sv, sv), &r); sv = mul(sv, r); }                                         // normalize: v * rcp(sqrt(dot)), one quarter-rate instruction
    float u, v;
    equirect_uv_fast(sv, u, v, special);
    // sample_2d_rgba32f_wrap_t<POT = true> on the resolved level; u, v in [-0.01, 1.01]: the float -> int conversions are in range
    // fixed8 (vq_sampling.h): floor((u * W - 0.5) * 256 + 0.5). W is a power of two: u * (256 W) is exact and scaling by 256 commutes with the
    // rounding of the subtraction, so RN(u * W - 0.5) * 256 == fma(u, 256 W, -128) — one operation instead of three, the same value
    const int fx = (int)__builtin_floorf(fma_(u, lv.W256, -128.0f) + 0.5f);
    const int fy = (int)__builtin_floorf(fma_(v, lv.H256, -128.0f) + 0.5f);
    const int ix = fx >> 8, iy = fy >> 8;
    const float wx = (float)(fx & 255) * 0.00390625f, wy = (float)(fy & 255) * 0.00390625f;
    TapTexels tt;
    tt.wx = wx; tt.wy = wy;
    if (REC) {                                               // k_diffuse_records: the 2 x 2 texels (rgb) of footprint (x, y), wrap applied, as 48 contiguous bytes
        const uint32_t idx = ((uint32_t)(iy & (lv.H - 1)) << (lv.rowShift - 4)) + (uint32_t)(ix & (lv.W - 1));
        const char* r = lv.rec + __umul24(idx, 48u);
        const float4 a = *(const float4*)r, b = *(const float4*)(r + 16), c = *(const float4*)(r + 32);
        tt.c00 = make_float4(a.x, a.y, a.z, 0); tt.c10 = make_float4(a.w, b.x, b.y, 0); tt.c01 = make_float4(b.z, b.w, c.x, 0); tt.c11 = make_float4(c.y, c.z, c.w, 0);
        return tt;
    }
    const uint32_t x0 = (uint32_t)(ix & (lv.W - 1)) << 4, x1 = (uint32_t)((ix + 1) & (lv.W - 1)) << 4;
    const uint32_t r0 = (uint32_t)(iy & (lv.H - 1)) << lv.rowShift, r1 = (uint32_t)((iy + 1) & (lv.H - 1)) << lv.rowShift;
    tt.c00 = *(const float4*)(lv.tex + (r0 + x0)); tt.c10 = *(const float4*)(lv.tex + (r0 + x1));
    tt.c01 = *(const float4*)(lv.tex + (r1 + x0)); tt.c11 = *(const float4*)(lv.tex + (r1 + x1));
    return tt;
}
VQD f3 diffuse_tap_blend(const TapTexels& q) {
    const float wx = q.wx, wy = q.wy;
    const float w00 = (1.0f - wx) * (1.0f - wy), w10 = wx * (1.0f - wy), w01 = (1.0f - wx) * wy, w11 = wx * wy;    // blend4
    return mk3(fma_(w11, q.c11.x, fma_(w01, q.c01.x, fma_(w10, q.c10.x, w00 * q.c00.x))),
               fma_(w11, q.c11.y, fma_(w01, q.c01.y, fma_(w10, q.c10.y, w00 * q.c00.y))),
               fma_(w11, q.c11.z, fma_(w01, q.c01.z, fma_(w10, q.c10.z, w00 * q.c00.z))));
}
template <bool REC = false>
VQD f3 diffuse_tap_fast(f3 sv, const DiffuseLevel& lv, bool& special) { return diffuse_tap_blend(diffuse_tap_fetch<REC>(sv, lv, special)); }
// Footprint records of the sampled level: a tap's 2 x 2 texels are 48 contiguous bytes (three 16-byte gathers from one record instead of four
// 12-byte gathers from two rows): the texture-address unit was the busiest unit of the texel form (79 %), 7.6 -> 6.6 ms with the 3 us pre-pass
// included (profiles/r3i_conv_kernels.md). rec[3 i .. 3 i + 2] = the rgb of texels (x, y), (x+1, y), (x, y+1), (x+1, y+1) of footprint i = y W + x, WRAP applied
__global__ __launch_bounds__(256) void k_diffuse_records(const float4* __restrict__ lvl, int W, int H, float4* __restrict__ rec) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int x1 = (x + 1) & (W - 1), y1 = (y + 1) & (H - 1);
    const float4 c00 = lvl[(size_t)y * W + x], c10 = lvl[(size_t)y * W + x1], c01 = lvl[(size_t)y1 * W + x], c11 = lvl[(size_t)y1 * W + x1];
    float4* r = rec + 3 * ((size_t)y * W + x);
    r[0] = make_float4(c00.x, c00.y, c00.z, c10.x); r[1] = make_float4(c10.y, c10.z, c01.x, c01.y); r[2] = make_float4(c01.z, c11.x, c11.y, c11.z);
}

constexpr int kDiffuseOrderedSteps = 2;                         // 16-tap steps per round of k_conv_diffuse_ordered (measured: scripts/bench_ibl_forms.py)
VQD bool near_one(float x) { return (x >= 0.99f) & (x <= 1.01f); }       // false for NaN

// phis/thetas: the fp32 sequences of the float-accumulated loops (CubemapConvolution.hlsl:132-136), built on the host.
// FAST: the level is a power-of-two image and the fast tap may run (launch_conv_diffuse_tables); the kernel still checks per block that the
// theta table holds unit (sin, cos) pairs, per texel that (right, up, N) is an orthonormal frame and per phi that (sin, cos) is a unit pair —
// a lane failing any of these runs every tap in the general form.
template <bool WAVE, int FMT, int FAST>
__global__ __launch_bounds__(256) void k_conv_diffuse(const float4* __restrict__ chain, int w0, int h0, int nMips, int res,
                                                      const float* __restrict__ phis, int nPhi, const float* __restrict__ thetas, int nTheta,
                                                      void* __restrict__ out, DiffuseLevel lv) {
    extern __shared__ float lds[];                               // (sinT, cosT)[nTheta]
    float2* scT = (float2*)lds;
    bool tabBad = false;
    for (int t = threadIdx.x; t < nTheta; t += 256) {
        float s, c; sincos_(thetas[t], &s, &c); scT[t] = make_float2(s, c);
        tabBad |= !near_one(fma_(s, s, c * c));
    }
    const bool fastBlock = FAST && __syncthreads_or(tabBad) == 0;
    if (!FAST) __syncthreads();
    const int lane = threadIdx.x & 63;
    const long total = 6L * res * res;
    const long texel = WAVE ? ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) : ((long)blockIdx.x * 256 + threadIdx.x);
    if (texel >= total) return;
    const int f = (int)(texel / ((long)res * res)), y = (int)((texel / res) % res), x = (int)(texel % res);
    const f3 N = normalize(cube_texel_dir(f, x, y, res));        // :114
    f3 up = mk3(0, 1, 0);
    const f3 right = normalize(cross(up, N));                    // :122
    up = normalize(cross(N, right));                             // :124
    const bool frameOK = near_one(dot(N, N)) && near_one(dot(right, right)) && near_one(dot(up, up)) &&
                         abs_(dot(N, right)) <= 0.01f && abs_(dot(N, up)) <= 0.01f && abs_(dot(right, up)) <= 0.01f;
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    for (int k = WAVE ? lane : 0; k < nPhi; k += (WAVE ? 64 : 1)) {
        float sinPhi, cosPhi; sincos_(phis[k], &sinPhi, &cosPhi);
        const bool laneFast = fastBlock && frameOK && near_one(fma_(sinPhi, sinPhi, cosPhi * cosPhi));
        auto sample_vec = [&](float2 sc) {                                                                        // :146-152
            const f3 ts = mk3(sc.x * cosPhi, sc.x * sinPhi, sc.y);
            return mk3((ts.x * right.x + ts.y * up.x) + ts.z * N.x, (ts.x * right.y + ts.y * up.y) + ts.z * N.y,
                       (ts.x * right.z + ts.y * up.z) + ts.z * N.z);
        };
        float2 scNext = scT[0];
        for (int t = 0; t < nTheta; ++t) {
            const float2 sc = scNext;
            scNext = scT[t + 1 < nTheta ? t + 1 : t];           // the next (sin, cos) pair is requested a whole tap ahead of its first use
            const float sinTheta = sc.x, cosTheta = sc.y;
            const f3 sv = sample_vec(sc);
            f3 c;
            if (FAST) {
                bool special = !laneFast;
                c = FAST == 2 ? diffuse_tap_fast<true>(sv, lv, special) : diffuse_tap_fast<false>(sv, lv, special);
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(special) != 0, 0)) c = diffuse_tap_general(sv, chain, w0, h0, nMips);
            } else {
                c = diffuse_tap_general(sv, chain, w0, h0, nMips);
            }
            ax = ax + (c.x * cosTheta) * sinTheta; ay = ay + (c.y * cosTheta) * sinTheta; az = az + (c.z * cosTheta) * sinTheta;
        }
    }
    if (WAVE) {
        #pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { ax = wave_xor_add(ax, m); ay = wave_xor_add(ay, m); az = wave_xor_add(az, m); }
        if (lane != 0) return;
    }
    const float rn = rcp((float)((long)nPhi * nTheta));          // numSamples :158,162
    store_px<FMT>(out, (size_t)texel, make_float4((PI_ * ax) * rn, (PI_ * ay) * rn, (PI_ * az) * rn, 1.0f));
}

// ---- diffuse irradiance in the reference's summation order, at wave speed -------------------------------------
// `irradiance += tap` of CubemapConvolution.hlsl:132-163 is one chain of nPhi * nTheta dependent additions per texel (phi outer, theta inner). The chain
// is cheap — three additions per tap against ~140 VALU for evaluating the tap — so evaluation and summation are split: a block owns 16 texels; in one
// step its 256 lanes evaluate 16 CONSECUTIVE taps (flattened index phi * nTheta + theta) of each of the 16 texels and park the weighted rgb in LDS as
// buf[tap][texel]; after a round of S steps one wave — lane = (texel, channel) — adds the round's 16 S taps to its accumulator IN TAP ORDER (one 4-byte
// LDS read + one v_add per 16 evaluated taps). Rounds are double-buffered, one barrier per round; the adding wave rotates and the accumulators live in
// LDS between rounds. Same taps, same order of additions as the one-lane-per-texel form (k_conv_diffuse<false,...>) and the oracle: identical bits.
// The block's texels are a 4 x 4 patch of a face (neighbouring texels sample neighbouring equirect texels: 8.23 against 8.40 ms for 16 texels of a row).
// The kernel is bound by the L1 tag-lookup rate, like the WAVE64 form (profiles/r4a_conv_ordered.md: TCP_TOTAL_CACHE_ACCESSES per CU == the kernel's clocks
// in both): the TA coalesces within the 4 lanes of a quad only, and a quad of 4 consecutive thetas (0.57 degrees apart: 0.8 texels of the 0.70-degree rows
// of the sampled level) touches 2.8 64-byte lines per gather against 2.15 for the WAVE64 form's 4 consecutive phis (0.57 sin(theta) degrees apart) —
// that, not the additions (+4 % VALU), is the 1.26 x between the two orders. S = 2 steps per round measured best (1: 8.44, 2: 8.29, 4: 9.07 ms).
// LDS: buf 2 x 16 x (16 S + 1) float4, acc 64 floats, (sin, cos) tables of theta and phi.
template <int FMT, int FAST>
__global__ __launch_bounds__(256) void k_conv_diffuse_ordered(const float4* __restrict__ chain, int w0, int h0, int nMips, int res,
                                                              const float* __restrict__ phis, int nPhi, const float* __restrict__ thetas, int nTheta,
                                                              void* __restrict__ out, DiffuseLevel lv, int patch) {
    constexpr int S = kDiffuseOrderedSteps, TPR = 16 * S;        // taps per texel per round
    extern __shared__ float4 lds4[];
    constexpr int BUF = 16 * (TPR + 1);                          // float4s of one tap buffer
    float4* buf = lds4;                                          // [2][BUF]
    float*  acc = (float*)(buf + 2 * BUF);                       // [64]: (texel, channel)
    float2* scT = (float2*)(acc + 64);                           // [nTheta]
    float2* scP = scT + nTheta;                                  // [nPhi]
    bool tabBad = false;
    for (int t = threadIdx.x; t < nTheta + nPhi; t += 256) {
        float s, c; sincos_(t < nTheta ? thetas[t] : phis[t - nTheta], &s, &c); scT[t] = make_float2(s, c);
        tabBad |= !near_one(fma_(s, s, c * c));
    }
    if (threadIdx.x < 64) acc[threadIdx.x] = 0.0f;
    const bool fastBlock = (FAST && __syncthreads_or(tabBad) == 0);
    if (!FAST) __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // a wave = 4 texels x 16 consecutive taps; buf[texel][tap] with the texel stride padded by one slot: the 16 lanes of a 128-bit write pass cover
    // 256 contiguous bytes and the adder's lanes (texel, channel) fall into distinct banks — no LDS bank conflict (PMC: SQ_LDS_BANK_CONFLICT 0)
    const int xi  = wave * 4 + (lane >> 4);                      // texel of the block
    const int j16 = lane & 15;                                   // tap slot of a 16-tap step
    const int wr0 = xi * (TPR + 1) + j16;
    // texel list: 4 x 4 patches of a face when the resolution allows it (patch != 0), else 16 consecutive texels of the row-major list
    const long total = 6L * res * res;
    auto block_texel = [&](int i) -> long {
        if (!patch) return (long)blockIdx.x * 16 + i;
        const int pr = res >> 2;                                 // patches per row
        const int f = (int)(blockIdx.x / (unsigned)(pr * pr)), pi = (int)(blockIdx.x % (unsigned)(pr * pr));
        return ((long)f * res + ((pi / pr) * 4 + (i >> 2))) * res + ((pi % pr) * 4 + (i & 3));
    };
    const long texel = block_texel(xi);
    const bool live = texel < total;
    const long tx = live ? texel : total - 1;
    const int f = (int)(tx / ((long)res * res)), y = (int)((tx / res) % res), x = (int)(tx % res);
    const f3 N = normalize(cube_texel_dir(f, x, y, res));        // :114
    f3 up = mk3(0, 1, 0);
    const f3 right = normalize(cross(up, N));                    // :122
    up = normalize(cross(N, right));                             // :124
    const bool frameOK = near_one(dot(N, N)) && near_one(dot(right, right)) && near_one(dot(up, up)) &&
                         abs_(dot(N, right)) <= 0.01f && abs_(dot(N, up)) <= 0.01f && abs_(dot(right, up)) <= 0.01f;
    const bool laneFast = fastBlock && frameOK;
    const int totalTaps = nPhi * nTheta;
    const int nRounds = (totalTaps + TPR - 1) / TPR;
    int k = j16 / nTheta, t = j16 % nTheta;                      // the lane's tap = (phi k, theta t); it advances by 16 per step
    float2 scPhi = scP[k < nPhi ? k : 0];
    for (int r = 0; r < nRounds; ++r) {
        float4* b = buf + (r & 1) * BUF;
        #pragma unroll
        for (int s = 0; s < S; ++s) {
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (k < nPhi) {
                const float2 sc = scT[t];
                const float sinTheta = sc.x, cosTheta = sc.y, sinPhi = scPhi.x, cosPhi = scPhi.y;
                const f3 ts = mk3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);                                 // :146-152
                const f3 sv = mk3((ts.x * right.x + ts.y * up.x) + ts.z * N.x, (ts.x * right.y + ts.y * up.y) + ts.z * N.y,
                                  (ts.x * right.z + ts.y * up.z) + ts.z * N.z);
                f3 c;
                if (FAST) {
                    bool special = !laneFast;
                    c = FAST == 2 ? diffuse_tap_fast<true>(sv, lv, special) : diffuse_tap_fast<false>(sv, lv, special);
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(special) != 0, 0)) c = diffuse_tap_general(sv, chain, w0, h0, nMips);
                } else {
                    c = diffuse_tap_general(sv, chain, w0, h0, nMips);
                }
                v = make_float4((c.x * cosTheta) * sinTheta, (c.y * cosTheta) * sinTheta, (c.z * cosTheta) * sinTheta, 0.0f);
                t += 16;
                if (t >= nTheta) {
                    do { t -= nTheta; ++k; } while (t >= nTheta);
                    scPhi = scP[k < nPhi ? k : 0];
                }
            }
            b[wr0 + s * 16] = v;
        }
        __syncthreads();
        if (wave == (r & 3)) {                                   // this round's adder: lane = 4 * texel + channel
            const float* bf = (const float*)b;
            float a = acc[lane];
            const int n = totalTaps - r * TPR;
            bf += (lane >> 2) * ((TPR + 1) * 4) + (lane & 3);
            if (n >= TPR) {
                #pragma unroll
                for (int j = 0; j < TPR; ++j) a = a + bf[j * 4];
            } else {
                for (int j = 0; j < n; ++j) a = a + bf[j * 4];
            }
            acc[lane] = a;
        }
    }
    __syncthreads();
    if (threadIdx.x < 64 && (lane & 3) == 0) {                   // lane = 4 * texel + channel, as the adders left it
        const long to = block_texel(lane >> 2);
        if (to < total) {
            const float rn = rcp((float)((long)nPhi * nTheta));  // numSamples :158,162
            store_px<FMT>(out, (size_t)to, make_float4((PI_ * acc[lane]) * rn, (PI_ * acc[lane + 1]) * rn, (PI_ * acc[lane + 2]) * rn, 1.0f));
        }
    }
}

// One executed sample of PSMain_SpecularIrradiance (CubemapConvolution.hlsl:197-215) behind `NdotL > 0`: the filtered environment colour at L.
// General form: every operation with its range tests and special cases.
VQD float4 specular_tap_general(f3 L, float NdotH, float HdotV, float Roughness, float fOmegaP, const float4* chain, int w0, int h0, int nMips) {
    const float D = NormalDistributionGGX(NdotH, Roughness);
    const float pdf = div_(D * NdotH, 4.0f * HdotV);
    const float fOmegaS = rcp(max_(512.0f * pdf, 0.00001f));
    const float fMipLevel = (Roughness == 0.0f) ? 0.0f : max_(0.5f * log2_(div_(fOmegaS, fOmegaP)) + -1.0f, 0.0f);
    const float2 uv = DirectionToEquirectUV(L);
    return sample_equirect_lod(chain, w0, h0, nMips, uv.x, uv.y, fMipLevel);
}
// Fast form (round 6; the kernel issued ~304 VALU per sample, 0.75 of the issue ceiling: profiles/r6n_pmc_conv.txt): the SAME operations on the same values with
//   * the three reciprocals unchecked (rcp_newton) and log2 without its special cases (log2_normal_bits), one validity flag: a reciprocal that is not a normal number, a
//     log2 operand that is not a positive normal number -> `special`;
//   * DirectionToEquirectUV branch-free (equirect_uv_fast);
//   * the trilinear WRAP fetch for a power-of-two chain (wave-uniform precondition): level offsets from a table built once per block (LDS), rows by shifts,
//     fixed8(u * W - 0.5) as floor(fma(u, 256 W, -128) + 0.5) (exact for a power-of-two W: diffuse_tap_fetch), 32-bit byte offsets from the scalar chain base.
// The caller redoes the sample of the whole wave in the general form when any lane raised `special`; where both apply they give identical bits
// (tests/test_gpu_conv_forms.py compares whole cubes of the two forms; the oracle comparisons run the fast form).
struct SpecChain { const char* chain; int lw0, lh0, nMips; float rOmegaP; const uint32_t* lvOff; };    // log2 of the level-0 size; rcp(fOmegaP); byte offsets of the levels (LDS)
VQD float4 spec_bilinear_pot(const SpecChain& sc, int level, float u, float v) {
    const int lw = max(sc.lw0 - level, 0), lh = max(sc.lh0 - level, 0);
    const int W = 1 << lw, H = 1 << lh;
    const int fx = (int)__builtin_floorf(fma_(u, (float)(W << 8), -128.0f) + 0.5f);
    const int fy = (int)__builtin_floorf(fma_(v, (float)(H << 8), -128.0f) + 0.5f);
    const int ix = fx >> 8, iy = fy >> 8;
    const float wx = (float)(fx & 255) * 0.00390625f, wy = (float)(fy & 255) * 0.00390625f;
    const uint32_t x0 = (uint32_t)(ix & (W - 1)) << 4, x1 = (uint32_t)((ix + 1) & (W - 1)) << 4;
    const uint32_t r0 = sc.lvOff[level] + ((uint32_t)(iy & (H - 1)) << (lw + 4)), r1 = sc.lvOff[level] + ((uint32_t)((iy + 1) & (H - 1)) << (lw + 4));
    return blend4(*(const float4*)(sc.chain + (r0 + x0)), *(const float4*)(sc.chain + (r0 + x1)), *(const float4*)(sc.chain + (r1 + x0)), *(const float4*)(sc.chain + (r1 + x1)), wx, wy);
}
VQD float4 specular_tap_fast(f3 L, float NdotH, float HdotV, float Roughness, const SpecChain& sc, bool& special) {
    // NormalDistributionGGX :65-79
    const float a = Roughness * Roughness, a2 = a * a, nh2 = NdotH * NdotH;
    const float t = fma_(nh2, a2 - 1.0f, 1.0f);
    const float denom = PI_ * (t * t);
    const bool eps = denom < EPSILON_;
    const float rd = rcp_newton(denom);
    special |= !eps & !is_normal(rd);
    const float D = eps ? 1.0f : a2 * rd;
    const float rv = rcp_newton(4.0f * HdotV);
    const float pdf = (D * NdotH) * rv;
    const float rs = rcp_newton(max_(512.0f * pdf, 0.00001f));           // fOmegaS
    special |= !(is_normal(rv) && is_normal(rs));
    const float x = rs * sc.rOmegaP;                                      // div_(fOmegaS, fOmegaP)
    special |= !((x >= 0x1p-126f) & (x <= 3.4028234663852886e38f));
    const float lod = (Roughness == 0.0f) ? 0.0f : max_(0.5f * log2_normal_bits(__float_as_uint(x), 0) + -1.0f, 0.0f);
    float u, v;
    equirect_uv_fast(L, u, v, special);
    // sample_equirect_lod_t<POT = true>
    const float maxl = (float)(sc.nMips - 1);
    const float l = (lod > 0.0f) ? ((lod < maxl) ? lod : maxl) : 0.0f;
    const int fl = (int)__builtin_floorf(l * 256.0f + 0.5f);             // l in [0, nMips - 1]: in range
    int lo = fl >> 8;
    float f = (float)(fl & 255) * 0.00390625f;
    if (lo >= sc.nMips - 1) { lo = sc.nMips - 1; f = 0.0f; }
    const float4 c0 = spec_bilinear_pot(sc, lo, u, v);
    if (f == 0.0f) return c0;
    const float4 c1 = spec_bilinear_pot(sc, lo + 1, u, v);
    const float g = 1.0f - f;
    return make_float4(fma_(f, c1.x, g * c0.x), fma_(f, c1.y, g * c0.y), fma_(f, c1.z, g * c0.z), fma_(f, c1.w, g * c0.w));
}
// ---- the one-sample form of a block (round 6) ---------------------------------------------------------------------------------------------------------
// At roughness 0 (mip 0: three quarters of the cube's texels) ImportanceSampleGGX returns the SAME half vector for every one of the 512 samples: cosTheta = sqrt((1 - y) /
// (1 - y)) = 1 exactly (IEEE quotient, BRDF.hlsl:222), sinTheta = 0, so the tangent-space vector is (+-0, +-0, 1) and tangent_to_world forms
// (t.c * +-0 + b.c * +-0) + N.c * 1 = N.c for every component with a finite frame and N.c != 0 — whatever the signs of the zeros. All 512 samples then fetch the same
// texels with the same weights, and `prefilteredColor += c * NdotL; totalWeight += NdotL` adds the same four values 512 times. Checked, not assumed: the block verifies that
// every entry of its table is (+-0, +-0, 1) (spec_table_is_axis: the polar half of every entry evaluated as written, without the 512 sincos) and every texel has a finite
// frame and no zero component of N; it then evaluates ONE sample per texel and performs the 512 additions in the reference's order on registers (identical operands, identical
// order: identical bits). cfg4's specular pass 0.409 -> 0.206 ms, the engine default's (512^2 x 9) 6.2 -> 3.06 ms (profiles/r6o_specular_one_sample.md); what
// remains is the 25 % of the texels with roughness > 0 (8 gathers per sample: two levels) and, per mip-0 block, the chain of 512 dependent additions.
VQD bool spec_axis_entry(float4 ht) { return (ht.x == 0.0f) & (ht.y == 0.0f) & (ht.z == 1.0f); }
VQD bool spec_texel_allows_one_sample(f3 N, const TangentFrame& fr) {
    const float m = (abs_(fr.tangent.x) + abs_(fr.tangent.y) + abs_(fr.tangent.z)) + (abs_(fr.bitangent.x) + abs_(fr.bitangent.y) + abs_(fr.bitangent.z));
    return (m < __builtin_inff()) & (N.x != 0.0f) & (N.y != 0.0f) & (N.z != 0.0f) & (abs_(N.x) + abs_(N.y) + abs_(N.z) < __builtin_inff());   // false for NaN
}
// the 512 tangent-space half vectors of (sample, roughness) into LDS (k_conv_specular_all / _ordered: once per block)
VQD void spec_build_table(float4* sHt, float Roughness) {
    for (uint32_t i = threadIdx.x; i < 512u; i += 256) {
        const float Xix = div_((float)i, 512.0f);
        float sp, cp; sincos_((2.0f * PI_) * Xix, &sp, &cp);
        const f3 Ht = ggx_sample_tangent(RadicalInverse_VdC(i), sp, cp, Roughness);
        sHt[i] = make_float4(Ht.x, Ht.y, Ht.z, 0.0f);
    }
}
// Roughness == 0: is every entry of that table (+-0, +-0, 1)? Only the polar half of ggx_sample_tangent is evaluated, as written — cosTheta = sqrt((1 - y) / (1 + (a^2 - 1) y)),
// sinTheta = sqrt(1 - cosTheta^2) — and compared with 1 and 0: (cos, sin)(phi) are finite for an argument in [0, 2 pi) (sincos_), so cos(phi) * 0 and sin(phi) * 0 are zeros.
// Entry 0 is built in full (the one-sample form evaluates it). Returns this lane's verdict over its two entries.
VQD bool spec_table_is_axis(float4* sHt, float Roughness) {
    bool axis = true;
    const float a = Roughness * Roughness;
    for (uint32_t i = threadIdx.x; i < 512u; i += 256) {
        const float Xiy = RadicalInverse_VdC(i);
        const float cosTheta = sqrt_(fdiv_(1.0f - Xiy, 1.0f + (a * a - 1.0f) * Xiy));
        const float sinTheta = sqrt_(1.0f - cosTheta * cosTheta);
        axis &= (cosTheta == 1.0f) & (sinTheta == 0.0f);
    }
    if (threadIdx.x == 0) {
        float sp, cp; sincos_((2.0f * PI_) * div_(0.0f, 512.0f), &sp, &cp);
        const f3 Ht = ggx_sample_tangent(RadicalInverse_VdC(0u), sp, cp, Roughness);
        sHt[0] = make_float4(Ht.x, Ht.y, Ht.z, 0.0f);
        axis &= spec_axis_entry(sHt[0]);
    }
    return axis;
}
constexpr int kSpecMaxLevels = 16;
// level offsets (bytes) of a power-of-two chain into LDS; returns whether the fast tap may run (power-of-two level 0, <= 16 levels, below 4 GB) — block-uniform
VQD bool spec_chain_setup(uint32_t* lvOff, int w0, int h0, int nMips) {
    const bool pot = is_pot2(w0, h0) && nMips <= kSpecMaxLevels && nMips >= 1 && (size_t)w0 * h0 <= ((size_t)1 << 26);
    if (pot && threadIdx.x < (unsigned)nMips) lvOff[threadIdx.x] = chain_level_offset<true>(w0, h0, (int)threadIdx.x) << 4;
    return pot;
}

// One sample of the specular pass for a texel with normal N (= V) and tangent frame fr, from the tangent-space half vector ht: (c * NdotL, NdotL), or zeros for a
// sample the shader skips (NdotL <= 0). The wave-level choice between the fast and the general form is taken over the lanes active at the call.
VQD float4 spec_sample(float4 ht, f3 N, const TangentFrame& fr, float Roughness, bool fastOK, const SpecChain& sc, float fOmegaP, const float4* chain, int w0, int h0, int nMips) {
    const f3 V = N;                                          // PSMain_SpecularIrradiance :172-174
    const f3 H = tangent_to_world(mk3(ht.x, ht.y, ht.z), fr, N);
    const f3 L = reflect(neg(V), H);
    const float NdotL = saturate(dot(N, L));
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (NdotL > 0.0f) {
        const float NdotH = saturate(dot(N, H));
        const float HdotV = saturate(dot(H, V));
        float4 c;
        if (fastOK) {
            bool special = false;
            c = specular_tap_fast(L, NdotH, HdotV, Roughness, sc, special);
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(special) != 0, 0)) c = specular_tap_general(L, NdotH, HdotV, Roughness, fOmegaP, chain, w0, h0, nMips);
        } else c = specular_tap_general(L, NdotH, HdotV, Roughness, fOmegaP, chain, w0, h0, nMips);
        v = make_float4(c.x * NdotL, c.y * NdotL, c.z * NdotL, NdotL);
    }
    return v;
}
// ---- specular prefilter -------------------------------------------------------------------------------
// All mips of the prefiltered cube in ONE launch, WAVE64 order (one wave per texel): the 7 launches of the per-mip form leave the chip almost
// empty for the five small mips (24 ... 1 536 waves). Block = 4 consecutive texels of the mip-major cube; every mip holds a multiple of 4 texels
// (6 r^2, r >= 2), so a block lies in one mip = one roughness: the 512 tangent-space half vectors of (sample, roughness) — sincos_, the Van der Corput
// reversal, an IEEE division and two square roots per sample — are built once per block into LDS (2 per lane instead of 8 per lane and texel)
// and the tangent frame of N once per texel. Everything else is the per-mip kernel's arithmetic on the same values: identical bits.
template <int FMT>
__global__ __launch_bounds__(256) void k_conv_specular_all(const float4* __restrict__ chain, int w0, int h0, int nMips, int res0, int MIPS, void* __restrict__ out, int allowFast) {
    __shared__ float4 sHt[512];
    __shared__ uint32_t sLvOff[kSpecMaxLevels];
    const bool fastOK = spec_chain_setup(sLvOff, w0, h0, nMips) && allowFast;
    const uint32_t NUM_SAMPLES = 512;
    const int lane = threadIdx.x & 63;
    const long T0 = (long)blockIdx.x * 4;
    int mip = 0, res = res0; long base = 0;
    while (mip < MIPS - 1 && T0 >= base + 6L * res * res) { base += 6L * res * res; ++mip; res >>= 1; }
    const float Roughness = div_((float)mip, (float)(MIPS - 1));                       // EnvironmentMapRendering.cpp:432
    // Roughness == 0 (mip 0): the cheap proof that the table is (+-0, +-0, 1) throughout; everything else, and a block that fails it: the table itself
    const bool axisTable = (Roughness == 0.0f && allowFast) ? (__syncthreads_and(spec_table_is_axis(sHt, Roughness)) != 0) : false;
    if (!axisTable) { __syncthreads(); spec_build_table(sHt, Roughness); __syncthreads(); }
    const long texel = T0 + (threadIdx.x >> 6) - base;                                  // within the mip
    if (texel >= 6L * res * res) return;
    const int f = (int)(texel / ((long)res * res)), y = (int)((texel / res) % res), x = (int)(texel % res);
    const f3 N = normalize(cube_texel_dir(f, x, y, res));
    const f3 V = N;
    const TangentFrame fr = tangent_frame(N);
    const float fOmegaP = div_(4.0f * PI_, (6.0f * (float)w0) * (float)h0);            // :203 with TextureDimensionsLOD0 = equirect dims (:433-434)
    const SpecChain sc = { (const char*)chain, 31 - __builtin_clz(w0 | 1), 31 - __builtin_clz(h0 | 1), nMips, rcp(fOmegaP), sLvOff };
    float ax = 0.0f, ay = 0.0f, az = 0.0f, aw = 0.0f;
    const bool oneSample = axisTable && spec_texel_allows_one_sample(N, fr);            // wave-uniform: a wave is one texel
    for (uint32_t i = (uint32_t)lane; i < NUM_SAMPLES; i += 64u) {
        if (oneSample && i >= 64u) {                                                     // the lane's other seven samples are its first one again: add it seven more times
            const float px = ax, py = ay, pz = az, pw = aw;
            for (int k = 0; k < 7; ++k) { ax = ax + px; ay = ay + py; az = az + pz; aw = aw + pw; }
            break;
        }
        const float4 ht = sHt[oneSample ? 0u : i];               // one-sample form: only entry 0 of the table was built (every entry is the same vector up to the signs of its zeros)
        const f3 H = tangent_to_world(mk3(ht.x, ht.y, ht.z), fr, N);
        const f3 L = reflect(neg(V), H);
        const float NdotL = saturate(dot(N, L));
        if (NdotL > 0.0f) {
            const float NdotH = saturate(dot(N, H));
            const float HdotV = saturate(dot(H, V));
            float4 c;
            if (fastOK) {
                bool special = false;
                c = specular_tap_fast(L, NdotH, HdotV, Roughness, sc, special);
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(special) != 0, 0)) c = specular_tap_general(L, NdotH, HdotV, Roughness, fOmegaP, chain, w0, h0, nMips);
            } else c = specular_tap_general(L, NdotH, HdotV, Roughness, fOmegaP, chain, w0, h0, nMips);
            ax = ax + c.x * NdotL; ay = ay + c.y * NdotL; az = az + c.z * NdotL; aw = aw + NdotL;
        }
    }
    #pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { ax = wave_xor_add(ax, m); ay = wave_xor_add(ay, m); az = wave_xor_add(az, m); aw = wave_xor_add(aw, m); }
    if (lane != 0) return;
    const float rw = rcp(max_(aw, 0.0001f));
    store_px<FMT>(out, (size_t)(base + texel), make_float4(ax * rw, ay * rw, az * rw, 1.0f));
}

// The specular prefilter in the reference's summation order (`prefilteredColor += ...; totalWeight += NdotL` over i = 0 ... 511, CubemapConvolution.hlsl:
// 186-219), every mip in one launch: the k_conv_diffuse_ordered scheme. A block owns 8 consecutive texels of the mip-major cube (every mip holds a multiple
// of 8 texels, so a block has one roughness and builds the 512 tangent-space half vectors once, as k_conv_specular_all does); per step its 256 lanes evaluate
// 32 consecutive samples of each texel into LDS, a sample the shader skips (NdotL <= 0) parks +0 (x + 0 == x for every accumulator value that can occur:
// the sums start at +0 and never become -0), and after a round of 64 samples one wave — lane = (texel, channel), 32 lanes — adds them in sample order.
template <int FMT>
__global__ __launch_bounds__(256) void k_conv_specular_ordered(const float4* __restrict__ chain, int w0, int h0, int nMips, int res0, int MIPS, void* __restrict__ out, int allowFast) {
    constexpr int S = 2, TPR = 32 * S;
    constexpr uint32_t NUM_SAMPLES = 512;
    __shared__ float4 sHt[512];
    __shared__ uint32_t sLvOff[kSpecMaxLevels];
    const bool fastOK = spec_chain_setup(sLvOff, w0, h0, nMips) && allowFast;
    __shared__ float4 buf[2][TPR * 8];
    __shared__ float acc[32];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int xi = lane >> 3;                                    // texel of the block
    const int j32 = wave * 8 + (lane & 7);                       // sample slot of a 32-sample step
    const long T0 = (long)blockIdx.x * 8;
    int mip = 0, res = res0; long base = 0;
    while (mip < MIPS - 1 && T0 >= base + 6L * res * res) { base += 6L * res * res; ++mip; res >>= 1; }
    const float Roughness = div_((float)mip, (float)(MIPS - 1));                       // EnvironmentMapRendering.cpp:432
    if (threadIdx.x < 32) acc[threadIdx.x] = 0.0f;
    const long texel = T0 + xi - base;                                                  // within the mip
    const bool live = texel < 6L * res * res;
    const long tx = live ? texel : 0;
    const int f = (int)(tx / ((long)res * res)), y = (int)((tx / res) % res), x = (int)(tx % res);
    const f3 N = normalize(cube_texel_dir(f, x, y, res));
    const TangentFrame fr = tangent_frame(N);
    const float fOmegaP = div_(4.0f * PI_, (6.0f * (float)w0) * (float)h0);            // :203 with TextureDimensionsLOD0 = equirect dims (:433-434)
    const SpecChain sc = { (const char*)chain, 31 - __builtin_clz(w0 | 1), 31 - __builtin_clz(h0 | 1), nMips, rcp(fOmegaP), sLvOff };
    auto sample = [&](uint32_t i) { return spec_sample(sHt[i], N, fr, Roughness, fastOK, sc, fOmegaP, chain, w0, h0, nMips); };
    // block-uniform: every entry of the table is (+-0, +-0, 1) and every texel of the block qualifies -> one sample per texel, 512 additions on registers (see above)
    // Roughness == 0 (mip 0): the cheap proof that the table is (+-0, +-0, 1) throughout (spec_table_is_axis); everything else, and a block that fails it: the table itself
    const bool oneSample = (Roughness == 0.0f && allowFast) &&
                           __syncthreads_and(spec_table_is_axis(sHt, Roughness) && (!live || spec_texel_allows_one_sample(N, fr))) != 0;
    if (!oneSample) { __syncthreads(); spec_build_table(sHt, Roughness); __syncthreads(); }
    if (oneSample) {
        if (wave != 0) return;
        if ((lane & 7) == 0) buf[0][xi] = sample(0);             // lane 8 xi: texel xi
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (lane < 32) {                                         // lane = 4 * texel + channel
            const float p = ((const float*)buf[0])[lane];
            float a = 0.0f;
            for (int j = 0; j < (int)NUM_SAMPLES; ++j) a = a + p;
            acc[lane] = a;
        }
        __builtin_amdgcn_wave_barrier();
        const long to = T0 + (lane >> 2);
        if (lane < 32 && (lane & 3) == 0 && to - base < 6L * res * res) {
            const float rw = rcp(max_(acc[lane + 3], 0.0001f));
            store_px<FMT>(out, (size_t)to, make_float4(acc[lane] * rw, acc[lane + 1] * rw, acc[lane + 2] * rw, 1.0f));
        }
        return;
    }
    for (int r = 0; r < (int)NUM_SAMPLES / TPR; ++r) {
        float4* b = buf[r & 1];
        #pragma unroll
        for (int s = 0; s < S; ++s) {
            b[(s * 32 + j32) * 8 + xi] = sample((uint32_t)(r * TPR + s * 32 + j32));
        }
        __syncthreads();
        if (wave == (r & 3) && lane < 32) {                      // this round's adder: lane = 4 * texel + channel
            const float* bf = (const float*)b + lane;
            float a = acc[lane];
            #pragma unroll
            for (int j = 0; j < TPR; ++j) a = a + bf[j * 32];
            acc[lane] = a;
        }
    }
    __syncthreads();
    const long texelOut = T0 + (lane >> 2);                                             // the adder's lane = 4 * texel + channel
    if (threadIdx.x < 32 && (lane & 3) == 0 && texelOut - base < 6L * res * res) {
        const float rw = rcp(max_(acc[lane + 3], 0.0001f));
        store_px<FMT>(out, (size_t)texelOut, make_float4(acc[lane] * rw, acc[lane + 1] * rw, acc[lane + 2] * rw, 1.0f));
    }
}

} // namespace

namespace vqk {

// ---- Skydome.hlsl:39-56: sky colour for the pixels no geometry covers ---------------------------
// One lane per pixel; HBM-bound in the worst case (all-sky frame: 8 B/pixel written, equirect taps cache-resident).
template <int FMT>
__global__ __launch_bounds__(256) void k_skydome(const float4* __restrict__ eq0, int w0, int h0, VQ_SkydomeParams sp,
                                                 const float4* __restrict__ cov, int covPitch, void* __restrict__ color, int W, int H, int pitch) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    if (cov && __float_as_int(cov[(size_t)y * covPitch + x].w) >= 0) return;
    const float nx = div_(2.0f * ((float)x + 0.5f), (float)W) - 1.0f, ny = 1.0f - div_(2.0f * ((float)y + 0.5f), (float)H);
    const float a = nx * sp.tanHalfFovX, b = ny * sp.tanHalfFovY;
    const f3 d = mk3(fma_(b, sp.up.x, fma_(a, sp.right.x, sp.forward.x)),
                     fma_(b, sp.up.y, fma_(a, sp.right.y, sp.forward.y)),
                     fma_(b, sp.up.z, fma_(a, sp.right.z, sp.forward.z)));
    const float2 uv = DirectionToEquirectUV(normalize(d));
    const float4 c = sample_2d_rgba32f_wrap(eq0, w0, h0, uv.x, uv.y);
    store_px<FMT>(color, (size_t)y * pitch + x, make_float4(c.x, c.y, c.z, 1.0f));
}

hipError_t launch_skydome(hipStream_t s, const float4* eq0, int w0, int h0, const VQ_SkydomeParams& sp, const float4* cov, int covPitch,
                          void* color, int W, int H, int pitch, int fmt) {
    dim3 grid((W + 255) / 256, H);
    if (fmt == VQHIP_FMT_RGBA32F) hipLaunchKernelGGL((k_skydome<0>), grid, dim3(256), 0, s, eq0, w0, h0, sp, cov, covPitch, color, W, H, pitch);
    else                          hipLaunchKernelGGL((k_skydome<1>), grid, dim3(256), 0, s, eq0, w0, h0, sp, cov, covPitch, color, W, H, pitch);
    return hipGetLastError();
}

// Unlit.hlsl:PSMain :58-61 over the engine's coverage plane: pixels covered by light gizmo k (ip2.w == -(2+k)) get its colour
template <int FMT>
__global__ __launch_bounds__(256) void k_unlit_composite(const float4* __restrict__ cov, int covPitch, UnlitColors cols, int n,
                                                         void* __restrict__ color, int W, int H, int pitch) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int idx = __float_as_int(cov[(size_t)y * covPitch + x].w);
    if (idx > -2) return;
    const long k = -2L - (long)idx;
    if (k >= n) return;
    store_px<FMT>(color, (size_t)y * pitch + x, cols.c[k]);
}
hipError_t launch_unlit_composite(hipStream_t s, const float4* cov, int covPitch, const UnlitColors& cols, int n, void* color, int W, int H, int pitch, int fmt) {
    dim3 grid((W + 255) / 256, H);
    if (fmt == VQHIP_FMT_RGBA32F) hipLaunchKernelGGL((k_unlit_composite<0>), grid, dim3(256), 0, s, cov, covPitch, cols, n, color, W, H, pitch);
    else                          hipLaunchKernelGGL((k_unlit_composite<1>), grid, dim3(256), 0, s, cov, covPitch, cols, n, color, W, H, pitch);
    return hipGetLastError();
}

hipError_t launch_brdf_lut(hipStream_t s, void* out, int size, int samples, int fmt, int p5ExpLog, const Options& opt) {
    dim3 grid((size + 255) / 256, size);
    const int allowFast = opt.lutForm != 1;                     // option "lut_form" = "general": the shared-H kernel with every range test left in
    if (fmt == VQHIP_FMT_RG16F) hipLaunchKernelGGL((k_brdf_lut<3>), grid, dim3(256), 0, s, out, size, samples, p5ExpLog, allowFast);
    else                        hipLaunchKernelGGL((k_brdf_lut<4>), grid, dim3(256), 0, s, out, size, samples, p5ExpLog, allowFast);
    return hipGetLastError();
}

hipError_t launch_mip_min(hipStream_t s, const float4* src, float4* dst, int sw, int sh, int dw, int dh) {
    hipLaunchKernelGGL(k_mip_min, dim3((dw + 255) / 256, dh), dim3(256), 0, s, src, dst, sw, sh, dw, dh);
    return hipGetLastError();
}

// phis = device array [nPhi], thetas = device array [nTheta], packed by the caller right behind each other
template <bool WAVE, int FMT>
static void launch_conv_diffuse_form(int fast, dim3 grid, size_t lds, hipStream_t s, const float4* chain, int w0, int h0, int nMips, int res,
                                     const float* phis, int nPhi, const float* thetas, int nTheta, void* out, const DiffuseLevel& lv) {
    if (fast == 2)  hipLaunchKernelGGL((k_conv_diffuse<WAVE, FMT, 2>), grid, dim3(256), lds, s, chain, w0, h0, nMips, res, phis, nPhi, thetas, nTheta, out, lv);
    else if (fast)  hipLaunchKernelGGL((k_conv_diffuse<WAVE, FMT, 1>), grid, dim3(256), lds, s, chain, w0, h0, nMips, res, phis, nPhi, thetas, nTheta, out, lv);
    else            hipLaunchKernelGGL((k_conv_diffuse<WAVE, FMT, 0>), grid, dim3(256), lds, s, chain, w0, h0, nMips, res, phis, nPhi, thetas, nTheta, out, lv);
}
static void diffuse_level(int w0, int h0, int nMips, int* level, int* W, int* H) {
    *level = nMips - 1 < 3 ? nMips - 1 : 3;                     // SampleLevel(uv, 3): lod clamped to the chain, fraction 0 -> one level (sample_equirect_lod_t)
    const int w = w0 >> *level, h = h0 >> *level;
    *W = w < 1 ? 1 : w; *H = h < 1 ? 1 : h;
}
// Bytes of the footprint-record buffer launch_conv_diffuse_tables can use for this chain (0: the level it samples is no power-of-two image, or
// too large for 24-bit record indices: the records do not apply)
size_t conv_diffuse_record_bytes(int w0, int h0, int nMips) {
    int level, W, H; diffuse_level(w0, h0, nMips, &level, &W, &H);
    if (((W & (W - 1)) | (H & (H - 1))) != 0 || (size_t)W * H >= ((size_t)1 << 24)) return 0;
    return (size_t)W * H * 48;
}
// recBuf: NULL, or a device buffer of conv_diffuse_record_bytes() bytes the launch may overwrite (the 2 x 2 footprints of the sampled level)
hipError_t launch_conv_diffuse_tables(hipStream_t s, const float4* chain, int w0, int h0, int nMips, int res,
                                      const float* phis, int nPhi, const float* thetas, int nTheta, int order, void* out, int fmt, void* recBuf, const Options& opt) {
    const long total = 6L * res * res;
    const size_t lds = (size_t)nTheta * 2 * sizeof(float);
    // the level SampleLevel(uv, 3) reads (sample_equirect_lod_t: lod clamped to the chain, fraction 0 -> one level) and whether the fast tap applies
    const int level = nMips - 1 < 3 ? nMips - 1 : 3;
    auto dim = [](int d0, int l) { const int d = d0 >> l; return d < 1 ? 1 : d; };
    size_t offPx = 0;
    for (int l = 0; l < level; ++l) offPx += (size_t)dim(w0, l) * dim(h0, l);
    DiffuseLevel lv;
    lv.tex = (const char*)(chain + offPx); lv.W = dim(w0, level); lv.H = dim(h0, level);
    lv.rowShift = 4; while ((1 << (lv.rowShift - 4)) < lv.W) ++lv.rowShift;
    lv.W256 = 256.0f * (float)lv.W; lv.H256 = 256.0f * (float)lv.H;
    // option "diffuse_form" = "general": every tap with its range tests and branches (the round-1/2 kernel)
    int fast = ((lv.W & (lv.W - 1)) | (lv.H & (lv.H - 1))) == 0 && (size_t)lv.W * lv.H * 16 < (1ull << 31) && opt.diffuseForm != 2 ? 1 : 0;
    lv.rec = nullptr;
    if (fast && recBuf && opt.diffuseForm != 1) {               // "texels": four gathers from the level itself (the first fast form; A/B, tests)
        hipLaunchKernelGGL(k_diffuse_records, dim3((lv.W + 255) / 256, lv.H), dim3(256), 0, s, (const float4*)lv.tex, lv.W, lv.H, (float4*)recBuf);
        lv.rec = (const char*)recBuf; fast = 2;
    }
    if (order == VQHIP_CONV_WAVE64) {
        dim3 grid((unsigned)((total + 3) / 4));
        if (fmt == VQHIP_FMT_RGBA32F) launch_conv_diffuse_form<true, 0>(fast, grid, lds, s, chain, w0, h0, nMips, res, phis, nPhi, thetas, nTheta, out, lv);
        else                          launch_conv_diffuse_form<true, 1>(fast, grid, lds, s, chain, w0, h0, nMips, res, phis, nPhi, thetas, nTheta, out, lv);
        return hipGetLastError();
    }
    // SEQUENTIAL (the reference's order): wave-parallel evaluation + ordered per-texel additions (k_conv_diffuse_ordered) while the (sin, cos) tables
    // of both loops fit in LDS next to the tap buffers; the one-lane-per-texel kernel otherwise (steps below ~0.002) or when asked for (A/B, tests)
    constexpr int S = kDiffuseOrderedSteps;
    const size_t ldsOrd = (size_t)2 * 16 * (16 * S + 1) * sizeof(float4) + 64 * sizeof(float) + ((size_t)nTheta + nPhi) * sizeof(float2);
    if (ldsOrd <= 64 * 1024 && (long)nPhi * nTheta < (1L << 30) && opt.diffuseSeqForm != 1) {   // option "diffuse_seq_form" = "lane": one lane per texel
        const int patch = (res % 4 == 0) ? 1 : 0;
        dim3 grid((unsigned)((total + 15) / 16));
        #define VQ_ORD(FMT_, FAST_) hipLaunchKernelGGL((k_conv_diffuse_ordered<FMT_, FAST_>), grid, dim3(256), ldsOrd, s, chain, w0, h0, nMips, res, phis, nPhi, thetas, nTheta, out, lv, patch)
        if (fmt == VQHIP_FMT_RGBA32F) { if (fast == 2) VQ_ORD(0, 2); else if (fast) VQ_ORD(0, 1); else VQ_ORD(0, 0); }
        else                          { if (fast == 2) VQ_ORD(1, 2); else if (fast) VQ_ORD(1, 1); else VQ_ORD(1, 0); }
        #undef VQ_ORD
        return hipGetLastError();
    }
    dim3 grid((unsigned)((total + 255) / 256));
    if (fmt == VQHIP_FMT_RGBA32F) launch_conv_diffuse_form<false, 0>(fast, grid, lds, s, chain, w0, h0, nMips, res, phis, nPhi, thetas, nTheta, out, lv);
    else                          launch_conv_diffuse_form<false, 1>(fast, grid, lds, s, chain, w0, h0, nMips, res, phis, nPhi, thetas, nTheta, out, lv);
    return hipGetLastError();
}

// every mip of the res0 cube (res0 a power of two >= 4: mips res0 ... 2) in one launch, either order; `out` = the mip-major cube
hipError_t launch_conv_specular_all(hipStream_t s, const float4* chain, int w0, int h0, int nMips, int res0, int MIPS, int order, void* out, int fmt, const Options& opt) {
    const int allowFast = opt.specularForm != 1;               // option "specular_form" = "general": every sample with its range tests and branches
    long total = 0;
    for (int m = 0; m < MIPS; ++m) { const long r = res0 >> m; total += 6 * r * r; }
    if (order == VQHIP_CONV_WAVE64) {
        dim3 grid((unsigned)((total + 3) / 4));
        if (fmt == VQHIP_FMT_RGBA32F) hipLaunchKernelGGL((k_conv_specular_all<0>), grid, dim3(256), 0, s, chain, w0, h0, nMips, res0, MIPS, out, allowFast);
        else                          hipLaunchKernelGGL((k_conv_specular_all<1>), grid, dim3(256), 0, s, chain, w0, h0, nMips, res0, MIPS, out, allowFast);
    } else {
        dim3 grid((unsigned)((total + 7) / 8));
        if (fmt == VQHIP_FMT_RGBA32F) hipLaunchKernelGGL((k_conv_specular_ordered<0>), grid, dim3(256), 0, s, chain, w0, h0, nMips, res0, MIPS, out, allowFast);
        else                          hipLaunchKernelGGL((k_conv_specular_ordered<1>), grid, dim3(256), 0, s, chain, w0, h0, nMips, res0, MIPS, out, allowFast);
    }
    return hipGetLastError();
}

} // namespace vqk
