// vqo_gbuffer.cpp — CPU restatement of the surface-assembly half of ForwardLighting.hlsl:PSMain (:226-287)
// and of MipImage's 4-byte branch ("G-buffer producer").
//
// ORACLE / TEST INFRASTRUCTURE ONLY (see vqo_oracle.cpp). PARITY: the arithmetic of PSMain's surface assembly is pinned
// against the reference's own ForwardLighting.hlsl run on the CPU (oracle/_ref, tests/test_ref_pinning.py::test_forward_lighting_psmain_
// with_material_textures); the texture FETCHES (filtering, LOD selection, derivatives) have no source in the reference and stay a
// restatement of D3D's rules. MipImage's 4-byte branch is pinned BIT-EXACT against the reference's own DXGIUtils.cpp (oracle/_ref/libvqref_mip.so).
//
// Contract additions of this file (DESIGN.md "G-buffer producer"):
//   * UNORM8 texels are filtered as the integers 0..255 and the filtered value is scaled once by rcp(255)
//     (the contract's a/b = a*rcp(b)). With 8-bit weights the bilinear blend of bytes is EXACT in binary32
//     (weights are multiples of 2^-16, sums <= 255 need 24 bits), so only the trilinear lerp and the final
//     scale round. (D3D11.3 §7.18.8 lets filtering run at fixed-point texel precision.)
//   * Material textures are R8G8B8A8_UNORM mip chains (TextureManager.cpp:590). AnisoSampler is
//     ANISOTROPIC_WRAP with MaxAnisotropy = 0 (RootSignatures.cpp:111,149) and LinearSampler TRILINEAR_WRAP:
//     both are evaluated as isotropic trilinear WRAP filtering, 8-bit fractions as in vqo_sampling.h.
//   * Implicit derivatives are the fine 2x2-quad differences of the *transformed* uv:
//       ddx = uv(x|1, y) - uv(x&~1, y),  ddy = uv(x, y|1) - uv(x, y&~1);
//     when the quad neighbour is outside the image or belongs to another material index the derivative
//     is 0 (hardware would extrapolate the triangle's plane through helper lanes; the planes do not hold it).
//   * LOD = 0.5*log2(max(|ddx*(W,H)|^2, |ddy*(W,H)|^2)) + bias, clamped to [0, mips-1] (D3D11.3 §7.18.11,
//     isotropic); -inf / NaN -> 0.
//   * Everything else keeps the HLSL's literal operation order (this is not one of the v2 lighting functions).
#include <cstdlib>
#include <cstring>
#include <vector>
#include <omp.h>

#include "../include/vqhip.h"
#include "vqo_math.h"
#include "vqo_sampling.h"

using namespace vqo;

namespace {

// LightingConstantBufferData.h:116-124
inline bool has_bit(int cfg, int bit) { return (cfg & (1 << bit)) > 0; }

// ShadingMath.hlsl:44-52
// Evaluated once per pixel: AS WRITTEN (contract v5) — the normal it returns steers the cube-map taps of the lighting pass, whose 8-bit
// filter fractions turn an ulp of the direction into a step of 1/256 of a texel difference.
inline f3 UnpackNormal(f3 S, f3 worldNormal, f3 worldTangent) {
    S = normalize_lit(f3{ S.x * 2.0f - 1.0f, S.y * 2.0f - 1.0f, S.z * 2.0f - 1.0f });
    const float nt = dot_lit(worldNormal, worldTangent);
    const f3 T = normalize_lit(sub(worldTangent, f3{ nt * worldNormal.x, nt * worldNormal.y, nt * worldNormal.z }));
    const f3 N = normalize_lit(worldNormal);
    const f3 B = normalize_lit(cross(T, N));
    // mul(SampledNormal, float3x3(T, B, N)): row vector times matrix with rows T, B, N
    return { (S.x * T.x + S.y * B.x) + S.z * N.x,
             (S.x * T.y + S.y * B.y) + S.z * N.y,
             (S.x * T.z + S.y * B.z) + S.z * N.z };
}

inline f3 SRGBToLinear(f3 c) { return { pow_(c.x, 2.2f), pow_(c.y, 2.2f), pow_(c.z, 2.2f) }; }   // ShadingMath.hlsl:65

struct Planes { const float* ip0; const float* ip1; const float* ip2; int W, H, pitch; };

inline int mat_index(const Planes& p, int x, int y) {
    int32_t i; std::memcpy(&i, p.ip2 + ((size_t)y * p.pitch + x) * 4 + 3, 4); return i;
}
inline f2 uv_transformed(const Planes& p, int x, int y, const VQ_MaterialData& m) {     // ForwardLighting.hlsl:226
    const size_t o = ((size_t)y * p.pitch + x) * 4;
    return { p.ip0[o + 3] * m.uvScaleOffset.x + m.uvScaleOffset.z, p.ip1[o + 3] * m.uvScaleOffset.y + m.uvScaleOffset.w };
}

// implicit derivatives of the transformed uv over the pixel's 2x2 quad (contract above): 0 where the neighbour is off the image or another material
inline void quad_derivatives(const Planes& in, int x, int y, int idx, const VQ_MaterialData& m, f2* ddx, f2* ddy) {
    *ddx = { 0, 0 }; *ddy = { 0, 0 };
    const int xa = x & ~1, xb = x | 1, ya = y & ~1, yb = y | 1;
    if (xb < in.W && mat_index(in, xa, y) == idx && mat_index(in, xb, y) == idx) {
        const f2 a = uv_transformed(in, xa, y, m), b = uv_transformed(in, xb, y, m);
        *ddx = { b.x - a.x, b.y - a.y };
    }
    if (yb < in.H && mat_index(in, x, ya) == idx && mat_index(in, x, yb) == idx) {
        const f2 a = uv_transformed(in, x, ya, m), b = uv_transformed(in, x, yb, m);
        *ddy = { b.x - a.x, b.y - a.y };
    }
}

// DepthPrePass.hlsl:PSMain :153-171 for one pixel: the packed surface normal (SurfaceN + 1) * 0.5 that the Z pre-pass writes to Tex_SceneNormals — the
// `g_normal` of SSR (vqo_ssr_environment_fallback) and of FFX-CACAO. Same surface normal as ForwardLighting.hlsl:265-267, but the normal map is fetched
// with Sample (no normalMapMipBias, :163) and the diffuse map only for the alpha test of the "_AlphaMasked" permutation (:157-161).
// Returns false for a pixel without geometry or a discarded fragment (the target keeps its clear value 0, SceneRendering.cpp:1289-1300).
bool scene_normal_pixel(const Planes& in, int x, int y, const vqhip_material* mats, int nMats, float* n01) {
    const int idx = mat_index(in, x, y);
    if (idx < 0 || idx >= nMats) return false;
    const vqhip_material& mt = mats[idx];
    const VQ_MaterialData& m = mt.data;
    const size_t o = ((size_t)y * in.pitch + x) * 4;
    const f2 uv = uv_transformed(in, x, y, m);                                          // :155
    f2 ddx, ddy;
    quad_derivatives(in, x, y, idx, m, &ddx, &ddy);
    if (mt.texDiffuse.reserved & VQHIP_MATERIAL_ALPHA_MASKED) {                         // #if ENABLE_ALPHA_MASK :157-161
        const int TEX_CFG = f2i_trunc(m.textureConfig);
        const f4 AlbedoAlpha = sample_material_tex(mt.texDiffuse, uv, ddx, ddy, 0.0f);
        if (has_bit(TEX_CFG, 0) && AlbedoAlpha.w < 0.01f) return false;
    }
    const f4 Normal4 = sample_material_tex(mt.texNormals, uv, ddx, ddy, 0.0f);          // :164
    const f3 N = normalize_lit(f3{ in.ip1[o], in.ip1[o + 1], in.ip1[o + 2] });          // :165
    const f3 T = normalize_lit(f3{ in.ip2[o], in.ip2[o + 1], in.ip2[o + 2] });          // :166
    const f3 Nrm = { Normal4.x, Normal4.y, Normal4.z };
    const f3 S = (length_lit(Nrm) < 0.01f) ? N : UnpackNormal(Nrm, N, T);               // :167
    n01[0] = (S.x + 1.0f) * 0.5f; n01[1] = (S.y + 1.0f) * 0.5f; n01[2] = (S.z + 1.0f) * 0.5f;   // :168
    return true;
}

// returns true when the fragment is discarded (ENABLE_ALPHA_MASK permutation, ForwardLighting.hlsl:237-240)
bool gbuffer_pixel(const Planes& in, int x, int y, const vqhip_material* mats, int nMats, float ambient,
                   const vqhip_ssao* ssao, float* o0, float* o1, float* o2, float* o3) {
    const int idx = mat_index(in, x, y);
    if (idx < 0 || idx >= nMats) {
        for (int k = 0; k < 4; ++k) o0[k] = o1[k] = o2[k] = o3[k] = 0.0f;
        return false;
    }
    const vqhip_material& mt = mats[idx];
    const VQ_MaterialData& m = mt.data;
    const size_t o = ((size_t)y * in.pitch + x) * 4;

    const f2 uv = uv_transformed(in, x, y, m);
    f2 ddx, ddy;
    quad_derivatives(in, x, y, idx, m, &ddx, &ddy);
    const int TEX_CFG = f2i_trunc(m.textureConfig);                                   // :227

    f4 AlbedoAlpha   = sample_material_tex(mt.texDiffuse,        uv, ddx, ddy, 0.0f); // :229
    const f4 Normal4 = sample_material_tex(mt.texNormals,        uv, ddx, ddy, m.normalMapMipBias);
    const f4 Emis4   = sample_material_tex(mt.texEmissive,       uv, ddx, ddy, 0.0f);
    const float Metalness = sample_material_tex(mt.texMetalness, uv, ddx, ddy, 0.0f).x;
    const float Roughness = sample_material_tex(mt.texRoughness, uv, ddx, ddy, 0.0f).x;
    const f4 ORM     = sample_material_tex(mt.texOcclRoughMetal, uv, ddx, ddy, 0.0f);
    const float LocalAO   = sample_material_tex(mt.texLocalAO,   uv, ddx, ddy, 0.0f).x;

    // :237-240  #if ENABLE_ALPHA_MASK (the "_AlphaMasked" PSO permutation == VQHIP_MATERIAL_ALPHA_MASKED): discard
    if ((mt.texDiffuse.reserved & VQHIP_MATERIAL_ALPHA_MASKED) && has_bit(TEX_CFG, 0) && AlbedoAlpha.w < 0.01f) {
        for (int k = 0; k < 4; ++k) o0[k] = o1[k] = o2[k] = o3[k] = 0.0f;
        return true;
    }
    const f3 Albedo   = SRGBToLinear({ AlbedoAlpha.x, AlbedoAlpha.y, AlbedoAlpha.z });     // :243
    const f3 Emissive = SRGBToLinear({ Emis4.x, Emis4.y, Emis4.z });                       // :244

    float ao = ambient;                                                                // :247
    const f3 mdiff = { m.diffuse.x, m.diffuse.y, m.diffuse.z }, memis = { m.emissiveColor.x, m.emissiveColor.y, m.emissiveColor.z };
    const f3 diffuseColor  = has_bit(TEX_CFG, 0) ? mul(Albedo, mdiff) : mdiff;         // :249
    const f3 emissiveColor = has_bit(TEX_CFG, 7) ? mul(Emissive, memis) : memis;       // :250
    float roughness = m.roughness, metalness = m.metalness;                            // :252-253

    const f3 N = normalize_lit(f3{ in.ip1[o], in.ip1[o + 1], in.ip1[o + 2] });         // :265
    const f3 T = normalize_lit(f3{ in.ip2[o], in.ip2[o + 1], in.ip2[o + 2] });         // :266
    const f3 Nrm = { Normal4.x, Normal4.y, Normal4.z };
    const f3 SurfN = (length_lit(Nrm) < 0.01f) ? N : UnpackNormal(Nrm, N, T);          // :267

    if (has_bit(TEX_CFG, 2)) ao *= LocalAO;                                            // :269
    if (has_bit(TEX_CFG, 4)) roughness *= Roughness;                                   // :270
    if (has_bit(TEX_CFG, 5)) metalness *= Metalness;                                   // :271
    if (has_bit(TEX_CFG, 8)) { roughness *= ORM.y; metalness *= ORM.z; }               // :272-277

    // :280-281  ScreenSpaceUV = (In.position.xy + 0.5) / ScreenDimensions, PointSampler = POINT_WRAP
    if (ssao && ssao->texels) {
        const float su = div_(((float)x + 0.5f) + 0.5f, (float)in.W), sv = div_(((float)y + 0.5f) + 0.5f, (float)in.H);
        ao *= fetch_r8_point_wrap((const uint8_t*)ssao->texels, ssao->width, ssao->height, su, sv);   // vqo_sampling.h
    }

    o0[0] = in.ip0[o]; o0[1] = in.ip0[o + 1]; o0[2] = in.ip0[o + 2]; o0[3] = ao;      // :284
    o1[0] = SurfN.x; o1[1] = SurfN.y; o1[2] = SurfN.z; o1[3] = roughness;
    o2[0] = diffuseColor.x; o2[1] = diffuseColor.y; o2[2] = diffuseColor.z; o2[3] = metalness;
    o3[0] = emissiveColor.x; o3[1] = emissiveColor.y; o3[2] = emissiveColor.z; o3[3] = m.emissiveIntensity;   // :251
    return false;
}

} // namespace

extern "C" {

// all pointers are HOST pointers here (the structs are shared with the product's ABI for convenience)
int vqo_gbuffer_from_materials(const vqhip_interpolants* in, const vqhip_material* materials, int numMaterials,
                               float fAmbientLightingFactor, const vqhip_ssao* ssao, const vqhip_gbuffer* out, int nthreads) {
    if (!in || !out || (numMaterials > 0 && !materials)) return -1;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    const Planes p = { (const float*)in->ip0, (const float*)in->ip1, (const float*)in->ip2, in->width, in->height, in->row_pitch_px };
    float* g0 = (float*)out->gb0; float* g1 = (float*)out->gb1; float* g2 = (float*)out->gb2; float* g3 = (float*)out->gb3;
    std::vector<uint8_t> discarded((size_t)in->width * in->height, 0);
    #pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int y = 0; y < in->height; ++y)
        for (int x = 0; x < in->width; ++x) {
            const size_t q = ((size_t)y * out->row_pitch_px + x) * 4;
            discarded[(size_t)y * in->width + x] = gbuffer_pixel(p, x, y, materials, numMaterials, fAmbientLightingFactor, ssao, g0 + q, g1 + q, g2 + q, g3 + q);
        }
    // discarded fragments read as "no geometry" afterwards: index -1 in the coverage plane (after the pass: quad partners needed the old index)
    float* ip2 = (float*)in->ip2;
    for (int y = 0; y < in->height; ++y)
        for (int x = 0; x < in->width; ++x)
            if (discarded[(size_t)y * in->width + x]) { const int32_t m1 = -1; std::memcpy(ip2 + ((size_t)y * in->row_pitch_px + x) * 4 + 3, &m1, 4); }
    return 0;
}

// DepthPrePass.hlsl:PSMain over the interpolant planes: out = Tex_SceneNormals, R10G10B10A2_UNORM (one uint32 per pixel, r in bits 0-9, alpha 1 -> 3;
// RenderResources.cpp:185-197, PipelineStateObjects.cpp:1634-1635) or RGBA32F holding the unquantised float4(SurfaceN, 1); tightly packed rows.
// float -> UNORM n: trunc(saturate(c) * (2^n - 1) + 0.5), NaN -> 0 (D3D11.3 §3.2.3.6, the rule of f32_to_unorm8).
int vqo_scene_normals_from_materials(const vqhip_interpolants* in, const vqhip_material* materials, int numMaterials, void* out, int outFmt, int nthreads) {
    if (!in || !out || (numMaterials > 0 && !materials)) return -1;
    if (outFmt != VQHIP_FMT_R10G10B10A2_UNORM && outFmt != VQHIP_FMT_RGBA32F) return -3;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    const Planes p = { (const float*)in->ip0, (const float*)in->ip1, (const float*)in->ip2, in->width, in->height, in->row_pitch_px };
    auto un10 = [](float c) -> uint32_t { const float s = c > 0.0f ? (c < 1.0f ? c : 1.0f) : 0.0f; return (uint32_t)(int)(s * 1023.0f + 0.5f); };
    #pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int y = 0; y < in->height; ++y)
        for (int x = 0; x < in->width; ++x) {
            float n[3] = { 0, 0, 0 };
            const bool covered = scene_normal_pixel(p, x, y, materials, numMaterials, n);
            const size_t q = (size_t)y * in->width + x;
            if (outFmt == VQHIP_FMT_RGBA32F) { float* d = (float*)out + q * 4; d[0] = n[0]; d[1] = n[1]; d[2] = n[2]; d[3] = covered ? 1.0f : 0.0f; }
            else ((uint32_t*)out)[q] = covered ? (un10(n[0]) | (un10(n[1]) << 10) | (un10(n[2]) << 20) | (3u << 30)) : 0u;
        }
    return 0;
}

// VQ_DXGI_UTILS::MipImage 4-byte branch, DXGIUtils.cpp:264-285: per channel (a+b+c+d)/4, integer division.
size_t vqo_mip_chain_texels(int w0, int h0, int nMips) { return tex_level_offset_px(w0, h0, nMips); }
int vqo_mip_chain_box_rgba8(uint8_t* chain, int w0, int h0, int nMips) {
    for (int l = 1; l < nMips; ++l) {
        const int sw = mip_dim(w0, l - 1), sh = mip_dim(h0, l - 1), dw = mip_dim(w0, l), dh = mip_dim(h0, l);
        const uint8_t* src = chain + tex_level_offset_px(w0, h0, l - 1) * 4;
        uint8_t* dst = chain + tex_level_offset_px(w0, h0, l) * 4;
        for (int y = 0; y < dh; ++y)
            for (int x = 0; x < dw; ++x) {
                const int x0 = 2 * x, y0 = 2 * y, x1 = (2 * x + 1 < sw) ? 2 * x + 1 : sw - 1, y1 = (2 * y + 1 < sh) ? 2 * y + 1 : sh - 1;
                for (int ch = 0; ch < 4; ++ch) {
                    const unsigned s = src[((size_t)y0 * sw + x0) * 4 + ch] + src[((size_t)y0 * sw + x1) * 4 + ch] +
                                       src[((size_t)y1 * sw + x0) * 4 + ch] + src[((size_t)y1 * sw + x1) * 4 + ch];
                    dst[((size_t)y * dw + x) * 4 + ch] = (uint8_t)(s / 4);
                }
            }
    }
    return 0;
}

float vqo_unorm8_to_float(int c) { return unorm8_to_float((float)(uint8_t)c); }

} // extern "C"
