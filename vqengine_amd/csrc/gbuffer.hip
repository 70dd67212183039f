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
// (readlane of the first pending lane): inside one iteration the material record and its seven texture
// descriptors are wave-uniform (SGPRs), null maps are skipped by a scalar branch, and the loop runs once for
// almost every wave.
//
// Sampling. The filter footprint (mip level + fraction, the four tap offsets and weights of both levels)
// depends only on (texture size, mip count, bias, uv, derivatives): it is computed once per distinct size and
// reused by every map of the material that shares it (a scalar compare of SGPR descriptors) — material sets are
// normally authored at one resolution, so seven maps cost one or two footprints. Power-of-two textures wrap by
// AND and find their mip offset in closed form; horizontally adjacent taps are fetched as one 8-byte load.
//
// Roofline: 48 B in + 64 B out = 112 B/pixel of HBM traffic (textures are cache-resident). Texture-less
// materials stream at the HBM roof; with all seven maps bound the kernel is bound by VALU + L1 gathers.
//
// Arithmetic/sampling contract: DESIGN.md "G-buffer producer".
#include "vq_shade.h"          // vq_internal.h, vq_devmath.h, vq_sampling.h + the lighting body of the fused kernel
#include "vq_mrt.h"

#include <cstdlib>
#ifndef VQ_PSMAIN_WAVES_DEFAULT
#define VQ_PSMAIN_WAVES_DEFAULT 6      // measured at 4K, 12 materials + 64 lights + IBL: 4 / 5 / 6 / 7 waves -> 1.535 / 1.58 / 1.528 / 1.63 ms (profiles/r4t_psmain_waves.jsonl, after the round-4
                                       // normalize_lit; round 3 measured 1.532 / 1.529 / 1.552 and kept 5)
#endif
namespace vqk {
using namespace vqd;

struct __attribute__((packed, aligned(4))) TexelPair { uint32_t a, b; };

// One level of a footprint: texel offsets (from the chain base) of the two rows, the two columns, the 4 weights.
struct LevelTaps { uint32_t row0, row1; int x0, x1; float w00, w10, w01, w11; };
struct Footprint { LevelTaps l0, l1; float f; };


template <bool POT> VQD LevelTaps level_taps(int w0, int h0, int level, float u, float v) {
    const int W = mip_dim(w0, level), H = mip_dim(h0, level);
    int ix, iy; float wx, wy;
    fixed8(u * (float)W - 0.5f, &ix, &wx);
    fixed8(v * (float)H - 0.5f, &iy, &wy);
    LevelTaps t;
    int y0, y1;
    if (POT) { t.x0 = ix & (W - 1); t.x1 = (ix + 1) & (W - 1); y0 = iy & (H - 1); y1 = (iy + 1) & (H - 1); }
    else     { t.x0 = wrapi(ix, W); t.x1 = wrapi(ix + 1, W); y0 = wrapi(iy, H); y1 = wrapi(iy + 1, H); }
    const uint32_t base = chain_level_offset<POT>(w0, h0, level);    // vq_sampling.h: closed form for power-of-two chains
    t.row0 = base + __umul24(y0, W);                              // dims < 2^24: full-rate v_mad_u32_u24
    t.row1 = base + __umul24(y1, W);
    t.w00 = (1.0f - wx) * (1.0f - wy); t.w10 = wx * (1.0f - wy); t.w01 = (1.0f - wx) * wy; t.w11 = wx * wy;   // == blend4's weights
    return t;
}

// Texture2D.Sample / SampleBias: isotropic trilinear WRAP, LOD from the quad derivatives
template <bool POT> VQD Footprint make_footprint(int w, int h, int mips, float2 uv, float2 ddx, float2 ddy, float bias) {
    const float W = (float)w, H = (float)h;
    const float dXx = ddx.x * W, dXy = ddx.y * H, dYx = ddy.x * W, dYy = ddy.y * H;
    const float rx = fma_(dXy, dXy, dXx * dXx), ry = fma_(dYy, dYy, dYx * dYx);
    const float lod = 0.5f * log2_(max_(rx, ry)) + bias;
    const float maxl = (float)(mips - 1);
    const float l = (lod > 0.0f) ? ((lod < maxl) ? lod : maxl) : 0.0f;
    const int fl = f2i_floor(l * 256.0f + 0.5f);
    int lo = fl >> 8;
    Footprint fp;
    fp.f = (float)(fl & 255) * 0.00390625f;
    if (lo >= mips - 1) { lo = mips - 1; fp.f = 0.0f; }
    fp.l0 = level_taps<POT>(w, h, lo, uv.x, uv.y);
    fp.l1 = level_taps<POT>(w, h, min(lo + 1, mips - 1), uv.x, uv.y);     // unused (weight 0) when f == 0
    return fp;
}

VQD float4 dec8(uint32_t q) { return make_float4((float)(q & 255u), (float)((q >> 8) & 255u), (float)((q >> 16) & 255u), (float)(q >> 24)); }

// bilinear blend of one level in byte units (exact: 8-bit weights x 8-bit texels fit binary32); same FMA chain as blend4
struct Taps4 { uint32_t a, b, c, d; };
template <bool PAIRS> VQD Taps4 load_taps(const uint32_t* __restrict__ t, const LevelTaps& k) {
    Taps4 q;
    // 32-bit byte offsets from the wave-uniform chain pointer (a chain is < 4 GB): global_load with SGPR base + VGPR offset
    const char* b = (const char*)t;
    if (PAIRS) {                                                     // x1 == x0 + 1 in every lane: two 8-byte loads
        const TexelPair p0 = *(const TexelPair*)(b + ((k.row0 + (uint32_t)k.x0) << 2)), p1 = *(const TexelPair*)(b + ((k.row1 + (uint32_t)k.x0) << 2));
        q.a = p0.a; q.b = p0.b; q.c = p1.a; q.d = p1.b;
    } else {
        q.a = *(const uint32_t*)(b + ((k.row0 + (uint32_t)k.x0) << 2)); q.b = *(const uint32_t*)(b + ((k.row0 + (uint32_t)k.x1) << 2));
        q.c = *(const uint32_t*)(b + ((k.row1 + (uint32_t)k.x0) << 2)); q.d = *(const uint32_t*)(b + ((k.row1 + (uint32_t)k.x1) << 2));
    }
    return q;
}
VQD float4 blend_taps(const Taps4& q, const LevelTaps& k) {
    const float4 c00 = dec8(q.a), c10 = dec8(q.b), c01 = dec8(q.c), c11 = dec8(q.d);
    float4 r;
    r.x = fma_(k.w11, c11.x, fma_(k.w01, c01.x, fma_(k.w10, c10.x, k.w00 * c00.x)));
    r.y = fma_(k.w11, c11.y, fma_(k.w01, c01.y, fma_(k.w10, c10.y, k.w00 * c00.y)));
    r.z = fma_(k.w11, c11.z, fma_(k.w01, c01.z, fma_(k.w10, c10.z, k.w00 * c00.z)));
    r.w = fma_(k.w11, c11.w, fma_(k.w01, c01.w, fma_(k.w10, c10.w, k.w00 * c00.w)));     // alpha: read by the ENABLE_ALPHA_MASK discard test (:237-240)
    return r;
}

VQD float4 fetch(const void* texels, const Footprint& fp) {
    const uint32_t* t = (const uint32_t*)texels;
    const bool two = __builtin_amdgcn_ballot_w64(fp.f != 0.0f) != 0;                              // some lane blends two levels
    const bool wraps = (fp.l0.x1 != fp.l0.x0 + 1) | (two & (fp.l1.x1 != fp.l1.x0 + 1));
    const bool pairs = __builtin_amdgcn_ballot_w64(wraps) == 0;                                   // no lane wraps in x
    // all loads of both levels are issued before the first use (one memory round trip per map)
    Taps4 q0, q1 = { 0, 0, 0, 0 };
    if (pairs) { q0 = load_taps<true>(t, fp.l0);  if (two) q1 = load_taps<true>(t, fp.l1); }
    else       { q0 = load_taps<false>(t, fp.l0); if (two) q1 = load_taps<false>(t, fp.l1); }
    float4 r = blend_taps(q0, fp.l0);
    if (two) {
        const float4 b = blend_taps(q1, fp.l1);
        const float f = fp.f, g = 1.0f - f;                          // f == 0: fma(0, b, 1*a) == a exactly
        r = make_float4(fma_(f, b.x, g * r.x), fma_(f, b.y, g * r.y), fma_(f, b.z, g * r.z), fma_(f, b.w, g * r.w));
    }
    const float s = 0.0039215688593685627f;                          // RN(1/255) = rcp(255.0f)
    return make_float4(r.x * s, r.y * s, r.z * s, r.w * s);
}

// once per pixel: as written (contract v5) — this normal steers the cube-map taps of the lighting pass
VQD f3 UnpackNormal(f3 S, f3 worldNormal, f3 worldTangent, bool dxc) {          // ShadingMath.hlsl:44-52; dxc: the reading of dot / normalize (vq_devmath.h)
    S = normalize_rt(mk3(S.x * 2.0f - 1.0f, S.y * 2.0f - 1.0f, S.z * 2.0f - 1.0f), dxc);
    const float nt = dot_rt(worldNormal, worldTangent, dxc);
    const f3 T = normalize_rt(sub(worldTangent, mk3(nt * worldNormal.x, nt * worldNormal.y, nt * worldNormal.z)), dxc);
    const f3 N = normalize_rt(worldNormal, dxc);
    const f3 B = normalize_rt(cross(T, N), dxc);
    return mk3((S.x * T.x + S.y * B.x) + S.z * N.x,
               (S.x * T.y + S.y * B.y) + S.z * N.y,
               (S.x * T.z + S.y * B.z) + S.z * N.z);
}

VQD bool has_bit(int cfg, int bit) { return (cfg & (1 << bit)) > 0; }   // LightingConstantBufferData.h:116-124

// Samples the maps of one (wave-uniform) material, reusing the footprint between maps of equal size / bias.
struct MaterialSampler {
    float2 uv, ddx, ddy;
    Footprint fp;
    int cw = -1, ch = -1, cm = -1; float cb = 0.0f;
    VQD float4 sample(const vqhip_texture2d& t, float bias) {
        if (!t.texels) return make_float4(0, 0, 0, 0);                // null SRV
        if (t.width != cw || t.height != ch || t.mips != cm || bias != cb) {
            const bool pot = ((t.width & (t.width - 1)) | (t.height & (t.height - 1))) == 0;
            fp = pot ? make_footprint<true>(t.width, t.height, t.mips, uv, ddx, ddy, bias)
                     : make_footprint<false>(t.width, t.height, t.mips, uv, ddx, ddy, bias);
            cw = t.width; ch = t.height; cm = t.mips; cb = bias;
        }
        return fetch(t.texels, fp);
    }
};

// The G-buffer record of pixel (x, y) — the state of PSMain at ForwardLighting.hlsl:284-293 — for the lane's pixel of the wave's 32x2 strip;
// all-zero for a pixel without geometry or a discarded fragment. Every lane of the wave must call it (quad swaps, ballots).
// PREPASS: the Z pre-pass's pixel shader instead (DepthPrePass.hlsl:PSMain :153-171): only g1 = float4((SurfaceN + 1) * 0.5, 1) is produced (0 = not covered);
// the diffuse map is fetched for the alpha test of the "_AlphaMasked" permutation alone, the normal map WITHOUT normalMapMipBias (:164 is Sample), the
// coverage plane is left as it is (the lighting pass repeats the discard itself, ForwardLighting.hlsl:237-240).
struct Record { float4 g0, g1, g2, g3; bool covered; };    // covered: a fragment of some material survived to the end of PSMain at this pixel
template <bool PREPASS = false>
VQD Record produce_record(const GbufArgs& a, int x, int y, bool inside, int lane) {
    const GbufConstants* __restrict__ gc = a.gc;
    Record rec;
    rec.g0 = rec.g1 = rec.g2 = rec.g3 = make_float4(0, 0, 0, 0);
    rec.covered = false;
    const uint32_t o = (__umul24(y, a.pitch) + (uint32_t)x) << 4;      // 32-bit byte offsets: planes are < 4 GB (checked by the C ABI)

    float4 i0 = make_float4(0, 0, 0, 0), i1 = i0, i2 = i0;
    int idx = -1;
    if (inside) {
        // the interpolant planes are a stream (48 B per pixel, read once per kernel): non-temporal, so that they do not evict the material maps / LUT / cubes from the XCD L2s
        typedef float v4f __attribute__((ext_vector_type(4)));
        auto nt = [](const void* p) { const v4f v = __builtin_nontemporal_load((const v4f*)p); return make_float4(v.x, v.y, v.z, v.w); };
        i0 = nt((const char*)a.ip0 + o); i1 = nt((const char*)a.ip1 + o); i2 = nt((const char*)a.ip2 + o);
        idx = __float_as_int(i2.w);
        if (idx >= gc->numMaterials) idx = -1;
    }
    // quad neighbours (raw uv + material index): lane^1 horizontal, lane^2 vertical
    const float hu = __shfl_xor(i0.w, 1), hv = __shfl_xor(i1.w, 1); const int hidx = __shfl_xor(idx, 1);
    const float vu = __shfl_xor(i0.w, 2), vv = __shfl_xor(i1.w, 2); const int vidx = __shfl_xor(idx, 2);

    // idx < 0 (no geometry): the record stays all-zero

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
            MaterialSampler ms;
            ms.uv = make_float2(i0.w * sx + ox, i1.w * sy + oy);                            // ForwardLighting.hlsl:226
            ms.ddx = make_float2(0, 0); ms.ddy = make_float2(0, 0);
            if (hidx == mi) {
                const float nu = hu * sx + ox, nv = hv * sy + oy;
                ms.ddx = (lane & 1) ? make_float2(ms.uv.x - nu, ms.uv.y - nv) : make_float2(nu - ms.uv.x, nv - ms.uv.y);
            }
            if (vidx == mi) {
                const float nu = vu * sx + ox, nv = vv * sy + oy;
                ms.ddy = (lane & 2) ? make_float2(ms.uv.x - nu, ms.uv.y - nv) : make_float2(nu - ms.uv.x, nv - ms.uv.y);
            }
            const int TEX_CFG = f2i_trunc(m.textureConfig);                                // :227
            // :229-235. A map whose Has*Map() bit is clear is fetched by the HLSL but never used: skipped here.
            // The normal map has no HasNormalMap() test (:267) and is always fetched. One rolled loop over the seven descriptors
            // (wave-uniform trip, scalar switch) keeps a single copy of the sampling code in the instruction cache.
            float4 AlbedoAlpha = make_float4(0, 0, 0, 0), Emis4 = AlbedoAlpha, ORM = AlbedoAlpha, Normal4 = AlbedoAlpha;
            float Metalness = 0.0f, Roughness = 0.0f, LocalAO = 0.0f;
            const vqhip_texture2d* slots = &mt.texDiffuse;             // t0,t1,t2,t4,t5,t6,t7 are consecutive in vqhip_material
            #pragma unroll 1
            for (int k = 0; k < (PREPASS ? 2 : 7); ++k) {
                const int bit = (0x2845710 >> (4 * k)) & 15;           // Has*Map bit of slot k: 0,1,7,5,4,8,2
                if (k != 1 && !has_bit(TEX_CFG, bit)) continue;
                if (PREPASS && k == 0 && !(mt.texDiffuse.reserved & VQHIP_MATERIAL_ALPHA_MASKED)) continue;      // DepthPrePass.hlsl:157-161: #if ENABLE_ALPHA_MASK
                const float4 r = ms.sample(slots[k], (k == 1 && !PREPASS) ? m.normalMapMipBias : 0.0f);
                switch (k) {
                    case 0: AlbedoAlpha = r; break;
                    case 1: Normal4 = r; break;
                    case 2: Emis4 = r; break;
                    case 3: Metalness = r.x; break;
                    case 4: Roughness = r.x; break;
                    case 5: ORM = r; break;
                    default: LocalAO = r.x; break;
                }
            }

            // :237-240  #if ENABLE_ALPHA_MASK  if (HasDiffuseMap(TEX_CFG) && AlbedoAlpha.a < 0.01f) discard;
            // the fragment is gone: zero record, and the pixel's index in the coverage plane becomes "no geometry" (quad partners keep the
            // index they read before, like the helper lanes of a discarded fragment keep serving derivatives)
            if ((mt.texDiffuse.reserved & VQHIP_MATERIAL_ALPHA_MASKED) && has_bit(TEX_CFG, 0) && AlbedoAlpha.w < 0.01f) {
                if (!PREPASS) ((float*)((char*)a.ip2 + o))[3] = __int_as_float(-1);            // the record stays all-zero
                todo = false;
                continue;
            }

            float ao = gc->ambient;                                                        // :247
            const f3 mdiff = mk3(m.diffuse.x, m.diffuse.y, m.diffuse.z), memis = mk3(m.emissiveColor.x, m.emissiveColor.y, m.emissiveColor.z);
            f3 diffuseColor = mdiff, emissiveColor = memis;
            if (has_bit(TEX_CFG, 0))                                                       // :243,249  SRGBToLinear = pow(c, 2.2)
                diffuseColor = mul(mk3(pow_unit(AlbedoAlpha.x, 2.2f), pow_unit(AlbedoAlpha.y, 2.2f), pow_unit(AlbedoAlpha.z, 2.2f)), mdiff);   // filtered UNORM8: [0,1], normal or 0
            if (has_bit(TEX_CFG, 7))                                                       // :244,250
                emissiveColor = mul(mk3(pow_unit(Emis4.x, 2.2f), pow_unit(Emis4.y, 2.2f), pow_unit(Emis4.z, 2.2f)), memis);
            float roughness = m.roughness, metalness = m.metalness;                        // :252-253

            const bool dxc = gc->arithDxc != 0;
            const f3 N = normalize_rt(mk3(i1.x, i1.y, i1.z), dxc);                         // :265
            const f3 Nrm = mk3(Normal4.x, Normal4.y, Normal4.z);
            f3 SurfN = N;                                                                  // :267  length(Normal) < 0.01 ? N : UnpackNormal(...)
            const bool unpack = !(length_rt(Nrm, dxc) < 0.01f);
            if (__builtin_amdgcn_ballot_w64(unpack) != 0) {                                // a real branch: most waves of a normal-map-less material skip it
                const f3 T = normalize_rt(mk3(i2.x, i2.y, i2.z), dxc);                     // :266
                const f3 U = UnpackNormal(Nrm, N, T, dxc);
                if (unpack) SurfN = U;
            }

            if (has_bit(TEX_CFG, 2)) ao *= LocalAO;                                        // :269
            if (has_bit(TEX_CFG, 4)) roughness *= Roughness;                               // :270
            if (has_bit(TEX_CFG, 5)) metalness *= Metalness;                               // :271
            if (has_bit(TEX_CFG, 8)) { roughness *= ORM.y; metalness *= ORM.z; }           // :272-277

            if (!PREPASS && gc->ssao.texels) {                                             // :280-281, POINT_WRAP
                const float su = div_(((float)x + 0.5f) + 0.5f, (float)a.width), sv = div_(((float)y + 0.5f) + 0.5f, (float)a.height);
                // coordinate snapped to 8 fractional bits before the floor (these coordinates sit exactly on texel borders)
                // su, sv are in (0, 1] so the texel index is in [0, dim]: WRAP is one conditional subtract (== the modulo)
                int tx = f2i_floor((su * (float)gc->ssao.width) * 256.0f + 0.5f) >> 8, ty = f2i_floor((sv * (float)gc->ssao.height) * 256.0f + 0.5f) >> 8;
                tx = tx >= gc->ssao.width ? tx - gc->ssao.width : tx;
                ty = ty >= gc->ssao.height ? ty - gc->ssao.height : ty;
                ao *= (float)((const uint8_t*)gc->ssao.texels)[(uint32_t)ty * (uint32_t)gc->ssao.width + (uint32_t)tx] * 0.0039215688593685627f;
            }
            if (PREPASS) {                                                                // DepthPrePass.hlsl:168-169 (the rest of this body is dead code here)
                rec.g1 = make_float4((SurfN.x + 1.0f) * 0.5f, (SurfN.y + 1.0f) * 0.5f, (SurfN.z + 1.0f) * 0.5f, 1.0f);
                rec.covered = true;
                todo = false;
                continue;
            }
            rec.g0 = make_float4(i0.x, i0.y, i0.z, ao);                                  // :284
            rec.g1 = make_float4(SurfN.x, SurfN.y, SurfN.z, roughness);
            rec.g2 = make_float4(diffuseColor.x, diffuseColor.y, diffuseColor.z, metalness);
            rec.g3 = make_float4(emissiveColor.x, emissiveColor.y, emissiveColor.z, m.emissiveIntensity);   // :251
            rec.covered = true;
            todo = false;
        }
    }
    return rec;
}

// lane -> pixel of the wave's 32x2 strip, quad-major (lanes 4q..4q+3 = the 2x2 pixel quad q)
VQD void strip_pixel(int& x, int& y, int& lane) {
    lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    x = (blockIdx.x * 4 + wave) * 32 + ((lane >> 2) << 1) + (lane & 1);
    y = blockIdx.y * 2 + ((lane >> 1) & 1);
}

__global__ __launch_bounds__(256, 4) void k_gbuffer_from_materials(GbufArgs a) {      // 4 waves per SIMD = 128 VGPRs (5 spilled): 17 % faster than the 137 the compiler takes uncapped (profiles/r4p_gbuffer_waves.jsonl)
    int x, y, lane;
    strip_pixel(x, y, lane);
    const bool inside = (x < a.width) & (y < a.height);
    const Record r = produce_record<false>(a, x, y, inside, lane);
    if (!inside) return;
    const uint32_t q = (__umul24(y, a.outPitch) + (uint32_t)x) << 4;
    *(float4*)((char*)a.gb0 + q) = r.g0; *(float4*)((char*)a.gb1 + q) = r.g1; *(float4*)((char*)a.gb2 + q) = r.g2; *(float4*)((char*)a.gb3 + q) = r.g3;
}

// The Z pre-pass's colour target (DepthPrePass.hlsl:PSMain; VQRenderer::RenderDepthPrePass, SceneRendering.cpp:1264-1360): Tex_SceneNormals, R10G10B10A2_UNORM
// (float -> UNORM n: trunc(saturate(c) * (2^n - 1) + 0.5)) or the unquantised float4; pixels nothing is drawn on keep the clear value 0 (:1289-1300).
template <int OUTFMT>
__global__ __launch_bounds__(256) void k_scene_normals_from_materials(GbufArgs a, void* out) {
    int x, y, lane;
    strip_pixel(x, y, lane);
    const bool inside = (x < a.width) & (y < a.height);
    const float4 n = produce_record<true>(a, x, y, inside, lane).g1;
    if (!inside) return;
    const size_t q = (size_t)y * a.outPitch + x;
    if (OUTFMT == VQHIP_FMT_RGBA32F) { ((float4*)out)[q] = n; return; }
    const uint32_t r = (uint32_t)(int)(saturate(n.x) * 1023.0f + 0.5f), g = (uint32_t)(int)(saturate(n.y) * 1023.0f + 0.5f), b = (uint32_t)(int)(saturate(n.z) * 1023.0f + 0.5f);
    ((uint32_t*)out)[q] = n.w != 0.0f ? (r | (g << 10) | (b << 20) | (3u << 30)) : 0u;
}

// PSMain as the engine has it (ForwardLighting.hlsl:226-380): the producer and the lighting body (vq_shade.h) in ONE kernel — the 64-byte record
// stays in registers instead of making a 128 B/pixel round trip through HBM. Bit-identical to vqhip_gbuffer_from_materials followed by
// vqhip_forward_lighting (the record is the same fp32 values either way; pixels without geometry shade the all-zero record like the two calls do).
template <bool HAS_ENV, bool HAS_CASTERS, int OUTFMT, int WAVES, int AR>
__global__ __launch_bounds__(256, WAVES) void k_forward_from_materials(GbufArgs a, const FrameConstants* fc, void* out, int outPitch, MrtArgs mrt) {
    int x, y, lane;
    strip_pixel(x, y, lane);
    const bool inside = (x < a.width) & (y < a.height);
    const Record r = produce_record<false>(a, x, y, inside, lane);
    if (!inside) return;
    // SV_TARGET1 / the motion vectors exist only where a fragment reaches :382-389: pixels without geometry and alpha-mask discards keep the targets' clear
    // values, as under the rasteriser (the colour target is still written for them: the all-zero record shaded, like the two calls do)
    if (r.covered) write_extra_targets(mrt, x, y, r.g2);
    const float4 c = shade_pixel<HAS_ENV, HAS_CASTERS, AR>(r.g0, r.g1, r.g2, r.g3, fc);
    store_px<OUTFMT>(out, (size_t)y * outPitch + x, c);
}


hipError_t launch_gbuffer_from_materials(hipStream_t s, const GbufArgs& a) {
    dim3 grid((a.width + 127) / 128, (a.height + 1) / 2);
    hipLaunchKernelGGL(k_gbuffer_from_materials, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_scene_normals_from_materials(hipStream_t s, const GbufArgs& a, void* out, int outFmt) {
    dim3 grid((a.width + 127) / 128, (a.height + 1) / 2);
    if (outFmt == VQHIP_FMT_RGBA32F) hipLaunchKernelGGL(k_scene_normals_from_materials<VQHIP_FMT_RGBA32F>, grid, dim3(256), 0, s, a, out);
    else                             hipLaunchKernelGGL(k_scene_normals_from_materials<VQHIP_FMT_R10G10B10A2_UNORM>, grid, dim3(256), 0, s, a, out);
    return hipGetLastError();
}

hipError_t launch_forward_from_materials(hipStream_t s, const GbufArgs& a, const FrameConstants* fc, bool hasEnv, bool hasCasters, void* out, int outPitch, int outFmt, int arithDxc, const Options& opt, const MrtArgs& mrt) {
    dim3 grid((a.width + 127) / 128, (a.height + 1) / 2);
    // register budget of the fused kernel: the producer half peaks at ~120 VGPRs (4 waves per SIMD), the lighting half needs 67; VQHIP_PSMAIN_WAVES = 5 / 6
    // caps the kernel at 96 / 80 VGPRs (the producer half then spills a little, the 64-light loop runs at higher occupancy)
    const int wv = opt.psmainWaves ? opt.psmainWaves : VQ_PSMAIN_WAVES_DEFAULT;      // option "psmain_waves"
#define FFM(E, C, F) do { if (arithDxc) hipLaunchKernelGGL((k_forward_from_materials<E, C, F, VQ_PSMAIN_WAVES_DEFAULT, 1>), grid, dim3(256), 0, s, a, fc, out, outPitch, mrt); /* DXC reading: one occupancy form */ \
                          else if (wv >= 6) hipLaunchKernelGGL((k_forward_from_materials<E, C, F, 6, 0>), grid, dim3(256), 0, s, a, fc, out, outPitch, mrt); \
                          else if (wv == 5) hipLaunchKernelGGL((k_forward_from_materials<E, C, F, 5, 0>), grid, dim3(256), 0, s, a, fc, out, outPitch, mrt); \
                          else hipLaunchKernelGGL((k_forward_from_materials<E, C, F, 4, 0>), grid, dim3(256), 0, s, a, fc, out, outPitch, mrt); } while (0)
#define FFM2(E, C) do { if (outFmt == VQHIP_FMT_RGBA32F) FFM(E, C, 0); else FFM(E, C, 1); } while (0)
    if (hasEnv) { if (hasCasters) FFM2(true, true); else FFM2(true, false); }
    else        { if (hasCasters) FFM2(false, true); else FFM2(false, false); }
#undef FFM2
#undef FFM
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
