// gbuffer.hip — SURVEY.md §8(f).1: the G-buffer producer.
//
//   k_gbuffer_from_materials : the surface-assembly half of ForwardLighting.hlsl:PSMain (:226-287) for every
//                              pixel of the interpolant planes: uv transform, seven material-map fetches
//                              (software trilinear WRAP on RGBA8 mip chains), SRGBToLinear, Has*Map() selection,
//                              UnpackNormal, ORM / AO / SSAO multiplies -> the four float4 planes of vqhip_gbuffer.
//   k_mip_box_rgba8          : VQ_DXGI_UTILS::MipImage 4-byte branch (DXGIUtils.cpp:264-285).
//
// Mapping. A wave owns a 32x2 pixel strip with quad-major lanes (lanes 4q..4q+3 = the 2x2 pixel quad q), so the
// implicit derivatives of Texture2D.Sample are two DPP quad swaps (lane^1 = horizontal, lane^2 = vertical
// neighbour) and every row of the strip is still one contiguous 512-byte read per plane. Materials are
// wave-coherent in practice, so the wave "waterfalls" over the distinct material indices it holds
// (readfirstlane): inside one iteration the material record and its seven texture descriptors are wave-uniform
// (SGPRs), null maps are skipped by a scalar branch, and the loop runs once for almost every wave.
//
// Roofline: 48 B in + 64 B out = 112 B/pixel of HBM traffic (textures are cache-resident); with all seven maps
// bound ~600-900 VALU/pixel (6 pow for the two sRGB decodes), i.e. roughly balanced between HBM and VALU.
//
// Arithmetic/sampling contract: DESIGN.md "G-buffer producer".
#include "vq_internal.h"
#include "vq_sampling.h"

namespace vqk {
using namespace vqd;

struct Tex { const uint8_t* p; int w, h, mips; };

VQD int wrap_fast(int i, int n, bool pot) { return pot ? (i & (n - 1)) : wrapi(i, n); }

// bilinear WRAP of one RGBA8 level in byte units (exact: 8-bit weights x 8-bit texels fit binary32)
VQD float4 sample_2d_rgba8_wrap(const uint8_t* tex, int W, int H, float u, float v) {
    int ix, iy; float wx, wy;
    fixed8(u * (float)W - 0.5f, &ix, &wx);
    fixed8(v * (float)H - 0.5f, &iy, &wy);
    const bool potx = (W & (W - 1)) == 0, poty = (H & (H - 1)) == 0;
    const int x0 = wrap_fast(ix, W, potx), x1 = wrap_fast(ix + 1, W, potx), y0 = wrap_fast(iy, H, poty), y1 = wrap_fast(iy + 1, H, poty);
    const uint32_t* t = (const uint32_t*)tex;
    const uint32_t a = t[(size_t)y0 * W + x0], b = t[(size_t)y0 * W + x1], c = t[(size_t)y1 * W + x0], d = t[(size_t)y1 * W + x1];
    auto dec = [](uint32_t q) { return make_float4((float)(q & 255u), (float)((q >> 8) & 255u), (float)((q >> 16) & 255u), (float)(q >> 24)); };
    return blend4(dec(a), dec(b), dec(c), dec(d), wx, wy);
}

VQD size_t tex_level_offset_px(int w0, int h0, int level) {
    size_t off = 0;
    for (int l = 0; l < level; ++l) off += (size_t)mip_dim(w0, l) * mip_dim(h0, l);
    return off;
}

// Texture2D.Sample / SampleBias: isotropic trilinear WRAP, LOD from the quad derivatives
VQD float4 sample_material_tex(const vqhip_texture2d& t, float2 uv, float2 ddx, float2 ddy, float bias) {
    const float W = (float)t.width, H = (float)t.height;
    const float dXx = ddx.x * W, dXy = ddx.y * H, dYx = ddy.x * W, dYy = ddy.y * H;
    const float rx = fma_(dXy, dXy, dXx * dXx), ry = fma_(dYy, dYy, dYx * dYx);
    const float lod = 0.5f * log2_(max_(rx, ry)) + bias;
    const float maxl = (float)(t.mips - 1);
    const float l = (lod > 0.0f) ? ((lod < maxl) ? lod : maxl) : 0.0f;
    const int fl = f2i_floor(l * 256.0f + 0.5f);
    int lo = fl >> 8;
    float f = (float)(fl & 255) * 0.00390625f;
    if (lo >= t.mips - 1) { lo = t.mips - 1; f = 0.0f; }
    const uint8_t* base = (const uint8_t*)t.texels;
    const float4 a = sample_2d_rgba8_wrap(base + tex_level_offset_px(t.width, t.height, lo) * 4, mip_dim(t.width, lo), mip_dim(t.height, lo), uv.x, uv.y);
    float4 r = a;
    if (f != 0.0f) {
        const float4 b = sample_2d_rgba8_wrap(base + tex_level_offset_px(t.width, t.height, lo + 1) * 4, mip_dim(t.width, lo + 1), mip_dim(t.height, lo + 1), uv.x, uv.y);
        const float g = 1.0f - f;
        r = make_float4(fma_(f, b.x, g * a.x), fma_(f, b.y, g * a.y), fma_(f, b.z, g * a.z), fma_(f, b.w, g * a.w));
    }
    const float s = 0.0039215688593685627f;     // RN(1/255) = rcp(255.0f)
    return make_float4(r.x * s, r.y * s, r.z * s, r.w * s);
}

VQD f3 UnpackNormal(f3 S, f3 worldNormal, f3 worldTangent) {          // ShadingMath.hlsl:44-52
    S = normalize(mk3(S.x * 2.0f - 1.0f, S.y * 2.0f - 1.0f, S.z * 2.0f - 1.0f));
    const f3 T = normalize(sub(worldTangent, mul(worldNormal, dot(worldNormal, worldTangent))));
    const f3 N = normalize(worldNormal);
    const f3 B = normalize(cross(T, N));
    return mk3(fma_(S.z, N.x, fma_(S.y, B.x, S.x * T.x)),
               fma_(S.z, N.y, fma_(S.y, B.y, S.x * T.y)),
               fma_(S.z, N.z, fma_(S.y, B.z, S.x * T.z)));
}

VQD bool has_bit(int cfg, int bit) { return (cfg & (1 << bit)) > 0; }   // LightingConstantBufferData.h:116-124

__global__ __launch_bounds__(256) void k_gbuffer_from_materials(GbufArgs a) {
    const GbufConstants* __restrict__ gc = a.gc;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = (blockIdx.x * 4 + wave) * 32 + ((lane >> 2) << 1) + (lane & 1);
    const int y = blockIdx.y * 2 + ((lane >> 1) & 1);
    const bool inside = (x < a.width) & (y < a.height);
    const size_t o = (size_t)y * a.pitch + x;

    float4 i0 = make_float4(0, 0, 0, 0), i1 = i0, i2 = i0;
    int idx = -1;
    if (inside) {
        i0 = a.ip0[o]; i1 = a.ip1[o]; i2 = a.ip2[o];
        idx = __float_as_int(i2.w);
        if (idx >= gc->numMaterials) idx = -1;
    }
    // quad neighbours (raw uv + material index): lane^1 horizontal, lane^2 vertical
    const float hu = __shfl_xor(i0.w, 1), hv = __shfl_xor(i1.w, 1); const int hidx = __shfl_xor(idx, 1);
    const float vu = __shfl_xor(i0.w, 2), vv = __shfl_xor(i1.w, 2); const int vidx = __shfl_xor(idx, 2);

    float4 o0 = make_float4(0, 0, 0, 0), o1 = o0, o2 = o0, o3 = o0;

    bool todo = idx >= 0;
    for (;;) {
        const unsigned long long pending = __builtin_amdgcn_ballot_w64(todo);
        if (pending == 0) break;
        // wave-uniform material of this pass = the index held by the first pending lane
        const int mi = __builtin_amdgcn_readlane(idx, __builtin_ctzll(pending));
        if (todo && idx == mi) {
            const vqhip_material& mt = gc->mats[mi];
            const VQ_MaterialData& m = mt.data;
            const float sx = m.uvScaleOffset.x, sy = m.uvScaleOffset.y, ox = m.uvScaleOffset.z, oy = m.uvScaleOffset.w;
            const float2 uv = make_float2(i0.w * sx + ox, i1.w * sy + oy);                  // ForwardLighting.hlsl:226
            float2 ddx = make_float2(0, 0), ddy = make_float2(0, 0);
            if (hidx == mi) {
                const float nu = hu * sx + ox, nv = hv * sy + oy;
                ddx = (lane & 1) ? make_float2(uv.x - nu, uv.y - nv) : make_float2(nu - uv.x, nv - uv.y);
            }
            if (vidx == mi) {
                const float nu = vu * sx + ox, nv = vv * sy + oy;
                ddy = (lane & 2) ? make_float2(uv.x - nu, uv.y - nv) : make_float2(nu - uv.x, nv - uv.y);
            }
            const int TEX_CFG = f2i_trunc(m.textureConfig);                                // :227
            const float4 z4 = make_float4(0, 0, 0, 0);
            const float4 AlbedoAlpha = mt.texDiffuse.texels        ? sample_material_tex(mt.texDiffuse,        uv, ddx, ddy, 0.0f) : z4;   // :229-235
            const float4 Normal4     = mt.texNormals.texels        ? sample_material_tex(mt.texNormals,        uv, ddx, ddy, m.normalMapMipBias) : z4;
            const float4 Emis4       = mt.texEmissive.texels       ? sample_material_tex(mt.texEmissive,       uv, ddx, ddy, 0.0f) : z4;
            const float Metalness    = mt.texMetalness.texels      ? sample_material_tex(mt.texMetalness,      uv, ddx, ddy, 0.0f).x : 0.0f;
            const float Roughness    = mt.texRoughness.texels      ? sample_material_tex(mt.texRoughness,      uv, ddx, ddy, 0.0f).x : 0.0f;
            const float4 ORM         = mt.texOcclRoughMetal.texels ? sample_material_tex(mt.texOcclRoughMetal, uv, ddx, ddy, 0.0f) : z4;
            const float LocalAO      = mt.texLocalAO.texels        ? sample_material_tex(mt.texLocalAO,        uv, ddx, ddy, 0.0f).x : 0.0f;

            float ao = gc->ambient;                                                        // :247
            const f3 mdiff = mk3(m.diffuse.x, m.diffuse.y, m.diffuse.z), memis = mk3(m.emissiveColor.x, m.emissiveColor.y, m.emissiveColor.z);
            f3 diffuseColor = mdiff, emissiveColor = memis;
            if (has_bit(TEX_CFG, 0))                                                       // :243,249  SRGBToLinear = pow(c, 2.2)
                diffuseColor = mul(mk3(pow_(AlbedoAlpha.x, 2.2f), pow_(AlbedoAlpha.y, 2.2f), pow_(AlbedoAlpha.z, 2.2f)), mdiff);
            if (has_bit(TEX_CFG, 7))                                                       // :244,250
                emissiveColor = mul(mk3(pow_(Emis4.x, 2.2f), pow_(Emis4.y, 2.2f), pow_(Emis4.z, 2.2f)), memis);
            float roughness = m.roughness, metalness = m.metalness;                        // :252-253

            const f3 N = normalize(mk3(i1.x, i1.y, i1.z));                                 // :265
            const f3 T = normalize(mk3(i2.x, i2.y, i2.z));                                 // :266
            const f3 Nrm = mk3(Normal4.x, Normal4.y, Normal4.z);
            const f3 SurfN = (length(Nrm) < 0.01f) ? N : UnpackNormal(Nrm, N, T);          // :267

            if (has_bit(TEX_CFG, 2)) ao *= LocalAO;                                        // :269
            if (has_bit(TEX_CFG, 4)) roughness *= Roughness;                               // :270
            if (has_bit(TEX_CFG, 5)) metalness *= Metalness;                               // :271
            if (has_bit(TEX_CFG, 8)) { roughness *= ORM.y; metalness *= ORM.z; }           // :272-277

            if (gc->ssao.texels) {                                                         // :280-281, POINT_WRAP
                const float su = div_(((float)x + 0.5f) + 0.5f, (float)a.width), sv = div_(((float)y + 0.5f) + 0.5f, (float)a.height);
                // coordinate snapped to 8 fractional bits before the floor (these coordinates sit exactly on texel borders)
                const int tx = wrapi(f2i_floor((su * (float)gc->ssao.width) * 256.0f + 0.5f) >> 8, gc->ssao.width);
                const int ty = wrapi(f2i_floor((sv * (float)gc->ssao.height) * 256.0f + 0.5f) >> 8, gc->ssao.height);
                ao *= (float)((const uint8_t*)gc->ssao.texels)[(size_t)ty * gc->ssao.width + tx] * 0.0039215688593685627f;
            }
            o0 = make_float4(i0.x, i0.y, i0.z, ao);                                        // :284
            o1 = make_float4(SurfN.x, SurfN.y, SurfN.z, roughness);
            o2 = make_float4(diffuseColor.x, diffuseColor.y, diffuseColor.z, metalness);
            o3 = make_float4(emissiveColor.x, emissiveColor.y, emissiveColor.z, m.emissiveIntensity);   // :251
            todo = false;
        }
    }
    if (inside) {
        const size_t q = (size_t)y * a.outPitch + x;
        a.gb0[q] = o0; a.gb1[q] = o1; a.gb2[q] = o2; a.gb3[q] = o3;
    }
}

hipError_t launch_gbuffer_from_materials(hipStream_t s, const GbufArgs& a) {
    dim3 grid((a.width + 127) / 128, (a.height + 1) / 2);
    hipLaunchKernelGGL(k_gbuffer_from_materials, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---- MipImage, 4-byte branch (DXGIUtils.cpp:264-285): each channel (a+b+c+d)/4, integer division ---------
__global__ __launch_bounds__(256) void k_mip_box_rgba8(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int sw, int sh, int dw, int dh) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= dw || y >= dh) return;
    const int x0 = 2 * x, y0 = 2 * y, x1 = min(2 * x + 1, sw - 1), y1 = min(2 * y + 1, sh - 1);
    const uint32_t a = src[(size_t)y0 * sw + x0], b = src[(size_t)y0 * sw + x1], c = src[(size_t)y1 * sw + x0], d = src[(size_t)y1 * sw + x1];
    // two channels per 32-bit lane-op: even bytes and odd bytes in 16-bit fields
    const uint32_t ev = (a & 0x00ff00ffu) + (b & 0x00ff00ffu) + (c & 0x00ff00ffu) + (d & 0x00ff00ffu);
    const uint32_t od = ((a >> 8) & 0x00ff00ffu) + ((b >> 8) & 0x00ff00ffu) + ((c >> 8) & 0x00ff00ffu) + ((d >> 8) & 0x00ff00ffu);
    dst[(size_t)y * dw + x] = ((ev >> 2) & 0x00ff00ffu) | (((od >> 2) & 0x00ff00ffu) << 8);
}

hipError_t launch_mip_box_rgba8(hipStream_t s, const void* src, void* dst, int sw, int sh, int dw, int dh) {
    dim3 grid((dw + 255) / 256, dh);
    hipLaunchKernelGGL(k_mip_box_rgba8, grid, dim3(256), 0, s, (const uint32_t*)src, (uint32_t*)dst, sw, sh, dw, dh);
    return hipGetLastError();
}

} // namespace vqk
