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
// ImportanceSampleGGX, BRDF.hlsl:217-238, with sin/cos(phi) supplied by the caller
VQD f3 ImportanceSampleGGX(float Xiy, float sinPhi, float cosPhi, f3 N, float roughness) {
    const float a = roughness * roughness;
    const float cosTheta = sqrt_(fdiv_(1.0f - Xiy, 1.0f + (a * a - 1.0f) * Xiy));
    const float sinTheta = sqrt_(1.0f - cosTheta * cosTheta);
    const f3 H = mk3(cosPhi * sinTheta, sinPhi * sinTheta, cosTheta);
    const f3 up = abs_(N.z) < 0.999f ? mk3(0, 0, 1) : mk3(1, 0, 0);
    const f3 tangent = normalize(cross(up, N));
    const f3 bitangent = cross(N, tangent);
    const f3 s = mk3((tangent.x * H.x + bitangent.x * H.y) + N.x * H.z,
                     (tangent.y * H.x + bitangent.y * H.y) + N.y * H.z,
                     (tangent.z * H.x + bitangent.z * H.y) + N.z * H.z);
    return normalize(s);
}

// ---- BRDF integration LUT ----------------------------------------------------------------------
// Block = 256 texels of one row; the per-sample (sin phi, cos phi, Xi.y) table is shared through LDS
// (every lane reads the same entry: LDS broadcast). 4 KB chunks of 512 samples bound the LDS use.
template <int FMT>
__global__ __launch_bounds__(256) void k_brdf_lut(void* __restrict__ out, int size, int samples, int p5ExpLog) {
    __shared__ float sSin[512], sCos[512], sXy[512];
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    const float NdotV = div_((float)x + 0.5f, (float)size);     // CubemapConvolution.hlsl:233-236
    const float roughness = div_((float)y + 0.5f, (float)size);
    const f3 V = mk3(sqrt_(1.0f - NdotV * NdotV), 0.0f, NdotV);
    const f3 N = mk3(0.0f, 0.0f, 1.0f);
    const float rcount = rcp((float)samples);
    float F0Scale = 0.0f, F0Bias = 0.0f;
    for (int base = 0; base < samples; base += 512) {
        __syncthreads();
        for (int j = threadIdx.x; j < 512; j += 256) {
            const uint32_t i = (uint32_t)(base + j);
            const float Xix = (float)i * rcount;                // Hammersley, ShadingMath.hlsl:119-127
            float sp, cp; sincos_((2.0f * PI_) * Xix, &sp, &cp);
            sSin[j] = sp; sCos[j] = cp; sXy[j] = RadicalInverse_VdC(i);
        }
        __syncthreads();
        const int n = min(512, samples - base);
        for (int j = 0; j < n; ++j) {                           // IntegrateBRDF, BRDF.hlsl:250-281
            const f3 H = ImportanceSampleGGX(sXy[j], sSin[j], sCos[j], N, roughness);
            const f3 L = normalize(reflect(neg(V), H));
            const float NdotL = max_(L.z, 0.0f);
            const float NdotH = max_(H.z, 0.0f);
            const float VdotH = max_(dot(V, H), 0.0f);
            if (NdotL > 0.0f) {
                const float G = G1_env(N, V, roughness) * G1_env(N, L, roughness);
                const float G_Vis = max_(div_(G * VdotH, NdotH * NdotV), 0.0001f);
                const float Fc = p5ExpLog ? pow5_explog(1.0f - VdotH) : pow5(1.0f - VdotH);
                F0Scale += (1.0f - Fc) * G_Vis;
                F0Bias += Fc * G_Vis;
            }
        }
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
// phis/thetas: the fp32 sequences of the float-accumulated loops (CubemapConvolution.hlsl:132-136), built on the host.
template <bool WAVE, int FMT>
__global__ __launch_bounds__(256) void k_conv_diffuse(const float4* __restrict__ chain, int w0, int h0, int nMips, int res,
                                                      const float* __restrict__ phis, int nPhi, const float* __restrict__ thetas, int nTheta,
                                                      void* __restrict__ out) {
    extern __shared__ float lds[];                               // sinT[nTheta], cosT[nTheta]
    float* sinT = lds; float* cosT = lds + nTheta;
    for (int t = threadIdx.x; t < nTheta; t += 256) { float s, c; sincos_(thetas[t], &s, &c); sinT[t] = s; cosT[t] = c; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long total = 6L * res * res;
    const long texel = WAVE ? ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) : ((long)blockIdx.x * 256 + threadIdx.x);
    if (texel >= total) return;
    const int f = (int)(texel / ((long)res * res)), y = (int)((texel / res) % res), x = (int)(texel % res);
    const f3 N = normalize(cube_texel_dir(f, x, y, res));        // :114
    f3 up = mk3(0, 1, 0);
    const f3 right = normalize(cross(up, N));                    // :122
    up = normalize(cross(N, right));                             // :124
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    for (int k = WAVE ? lane : 0; k < nPhi; k += (WAVE ? 64 : 1)) {
        float sinPhi, cosPhi; sincos_(phis[k], &sinPhi, &cosPhi);
        for (int t = 0; t < nTheta; ++t) {
            const float sinTheta = sinT[t], cosTheta = cosT[t];
            const f3 ts = mk3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);                                  // :146-150
            f3 sv = mk3((ts.x * right.x + ts.y * up.x) + ts.z * N.x, (ts.x * right.y + ts.y * up.y) + ts.z * N.y,
                        (ts.x * right.z + ts.y * up.z) + ts.z * N.z);                                           // :152
            sv = normalize(sv);
            const float2 uv = DirectionToEquirectUV(sv);
            const float4 c = sample_equirect_lod(chain, w0, h0, nMips, uv.x, uv.y, 3.0f);                        // mipLevel = 3 :155-157
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

// ---- specular prefilter -------------------------------------------------------------------------------
template <bool WAVE, int FMT>
__global__ __launch_bounds__(256) void k_conv_specular(const float4* __restrict__ chain, int w0, int h0, int nMips, int res, int mip, int MIPS,
                                                       void* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long total = 6L * res * res;
    const long texel = WAVE ? ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) : ((long)blockIdx.x * 256 + threadIdx.x);
    if (texel >= total) return;
    const int f = (int)(texel / ((long)res * res)), y = (int)((texel / res) % res), x = (int)(texel % res);
    const float Roughness = div_((float)mip, (float)(MIPS - 1));                       // EnvironmentMapRendering.cpp:432
    const f3 N = normalize(cube_texel_dir(f, x, y, res));
    const f3 V = N;
    const float fOmegaP = div_(4.0f * PI_, (6.0f * (float)w0) * (float)h0);            // :203 with TextureDimensionsLOD0 = equirect dims (:433-434)
    const uint32_t NUM_SAMPLES = 512;
    float ax = 0.0f, ay = 0.0f, az = 0.0f, aw = 0.0f;
    for (uint32_t i = WAVE ? (uint32_t)lane : 0u; i < NUM_SAMPLES; i += (WAVE ? 64u : 1u)) {
        const float Xix = div_((float)i, (float)NUM_SAMPLES);
        float sp, cp; sincos_((2.0f * PI_) * Xix, &sp, &cp);
        const f3 H = ImportanceSampleGGX(RadicalInverse_VdC(i), sp, cp, N, Roughness);
        const f3 L = reflect(neg(V), H);
        const float NdotL = saturate(dot(N, L));
        if (NdotL > 0.0f) {
            const float NdotH = saturate(dot(N, H));
            const float HdotV = saturate(dot(H, V));
            const float D = NormalDistributionGGX(NdotH, Roughness);
            const float pdf = div_(D * NdotH, 4.0f * HdotV);
            const float fOmegaS = rcp(max_((float)NUM_SAMPLES * pdf, 0.00001f));
            const float fMipLevel = (Roughness == 0.0f) ? 0.0f : max_(0.5f * log2_(div_(fOmegaS, fOmegaP)) + -1.0f, 0.0f);
            const float2 uv = DirectionToEquirectUV(L);
            const float4 c = sample_equirect_lod(chain, w0, h0, nMips, uv.x, uv.y, fMipLevel);
            ax = ax + c.x * NdotL; ay = ay + c.y * NdotL; az = az + c.z * NdotL; aw = aw + NdotL;
        }
    }
    if (WAVE) {
        #pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { ax = wave_xor_add(ax, m); ay = wave_xor_add(ay, m); az = wave_xor_add(az, m); aw = wave_xor_add(aw, m); }
        if (lane != 0) return;
    }
    const float rw = rcp(max_(aw, 0.0001f));
    store_px<FMT>(out, (size_t)texel, make_float4(ax * rw, ay * rw, az * rw, 1.0f));
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

hipError_t launch_brdf_lut(hipStream_t s, void* out, int size, int samples, int fmt, int p5ExpLog) {
    dim3 grid((size + 255) / 256, size);
    if (fmt == VQHIP_FMT_RG16F) hipLaunchKernelGGL((k_brdf_lut<3>), grid, dim3(256), 0, s, out, size, samples, p5ExpLog);
    else                        hipLaunchKernelGGL((k_brdf_lut<4>), grid, dim3(256), 0, s, out, size, samples, p5ExpLog);
    return hipGetLastError();
}

hipError_t launch_mip_min(hipStream_t s, const float4* src, float4* dst, int sw, int sh, int dw, int dh) {
    hipLaunchKernelGGL(k_mip_min, dim3((dw + 255) / 256, dh), dim3(256), 0, s, src, dst, sw, sh, dw, dh);
    return hipGetLastError();
}

// phis = device array [nPhi], thetas = device array [nTheta], packed by the caller right behind each other
hipError_t launch_conv_diffuse_tables(hipStream_t s, const float4* chain, int w0, int h0, int nMips, int res,
                                      const float* phis, int nPhi, const float* thetas, int nTheta, int order, void* out, int fmt) {
    const long total = 6L * res * res;
    const size_t lds = (size_t)nTheta * 2 * sizeof(float);
    if (order == VQHIP_CONV_WAVE64) {
        dim3 grid((unsigned)((total + 3) / 4));
        if (fmt == VQHIP_FMT_RGBA32F) hipLaunchKernelGGL((k_conv_diffuse<true, 0>), grid, dim3(256), lds, s, chain, w0, h0, nMips, res, phis, nPhi, thetas, nTheta, out);
        else                          hipLaunchKernelGGL((k_conv_diffuse<true, 1>), grid, dim3(256), lds, s, chain, w0, h0, nMips, res, phis, nPhi, thetas, nTheta, out);
    } else {
        dim3 grid((unsigned)((total + 255) / 256));
        if (fmt == VQHIP_FMT_RGBA32F) hipLaunchKernelGGL((k_conv_diffuse<false, 0>), grid, dim3(256), lds, s, chain, w0, h0, nMips, res, phis, nPhi, thetas, nTheta, out);
        else                          hipLaunchKernelGGL((k_conv_diffuse<false, 1>), grid, dim3(256), lds, s, chain, w0, h0, nMips, res, phis, nPhi, thetas, nTheta, out);
    }
    return hipGetLastError();
}

hipError_t launch_conv_specular(hipStream_t s, const float4* chain, int w0, int h0, int nMips, int res, int mip, int MIPS,
                                int order, void* out, int fmt) {
    const long total = 6L * res * res;
    if (order == VQHIP_CONV_WAVE64) {
        dim3 grid((unsigned)((total + 3) / 4));
        if (fmt == VQHIP_FMT_RGBA32F) hipLaunchKernelGGL((k_conv_specular<true, 0>), grid, dim3(256), 0, s, chain, w0, h0, nMips, res, mip, MIPS, out);
        else                          hipLaunchKernelGGL((k_conv_specular<true, 1>), grid, dim3(256), 0, s, chain, w0, h0, nMips, res, mip, MIPS, out);
    } else {
        dim3 grid((unsigned)((total + 255) / 256));
        if (fmt == VQHIP_FMT_RGBA32F) hipLaunchKernelGGL((k_conv_specular<false, 0>), grid, dim3(256), 0, s, chain, w0, h0, nMips, res, mip, MIPS, out);
        else                          hipLaunchKernelGGL((k_conv_specular<false, 1>), grid, dim3(256), 0, s, chain, w0, h0, nMips, res, mip, MIPS, out);
    }
    return hipGetLastError();
}

} // namespace vqk
