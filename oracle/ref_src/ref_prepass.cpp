// ref_prepass.cpp — harness that RUNS the reference's own DepthPrePass.hlsl:PSMain (:153-171, with Lighting.hlsl / ShadingMath.hlsl:UnpackNormal) on the CPU:
// the pixel shader of the Z pre-pass, whose colour target is Tex_SceneNormals — the packed surface normals that SSR (`g_normal`) and FFX-CACAO read.
// The source is read where it lies under /root/reference, rewritten syntactically by hlsl2cpp.py into oracle/_ref/gen/ and compiled here against
// hlsl_shim.h. Part of oracle/_ref/libvqref_shaders.so and, with -DENABLE_ALPHA_MASK=1 (the "_AlphaMasked" Z-pre-pass PSOs, PipelineStateObjects.cpp:1600-1660),
// of libvqref_shaders_am.so. TEST INFRASTRUCTURE: pins vqo_gbuffer.cpp:scene_normal_pixel; never loaded by the product.
//   * every arithmetic statement of PSMain is the reference's source; the texture FETCHES are the oracle's sampling contract (ref_hooks.cpp), with the
//     implicit-derivative rule of vqo_gbuffer.cpp — as in ref_forward.cpp;
//   * the render target's float -> R10G10B10A2_UNORM conversion is fixed-function: the caller applies it (tests/ref_lib.py), this returns PSMain's float4.
#include <cstdint>
#include <cstring>

#include "../../include/vqhip.h"
#include "../vqo_math.h"
#include "../vqo_sampling.h"
#include "ref_hooks.h"

#if ENABLE_ALPHA_MASK
#define discard do { vqref::g_ctx.discarded = true; return float4(0.0f); } while (0)
#endif

namespace hlsl {
namespace zpp {
#define VQ_GPU 1
#include "DepthPrePass.hlsl"
#undef f2
#undef f3
} // namespace zpp
} // namespace hlsl

namespace {
using namespace hlsl;
using namespace hlsl::zpp;
using namespace vqref;

void bindTex(Texture2D& t, const vqhip_texture2d& d) { t.res = &d; t.kind = d.texels ? kTexMaterial : kTexNull; }
int32_t matIndex(const float* ip2, size_t o) { int32_t i; std::memcpy(&i, ip2 + o + 3, 4); return i; }
} // namespace

extern "C" {

// DepthPrePass.hlsl:PSMain over an image of interpolants + material table. out = RGBA32F [H][W][4] = the float4 PSMain returns; pixels without geometry and
// discarded fragments hold the target's clear value 0 (SceneRendering.cpp:1289-1300), so out.w tells covered (1) from uncovered (0).
int vqref_prepass_normals(const vqhip_interpolants* in, const vqhip_material* mats, int nMats, float* out) {
    if (!in || !out) return -1;
    const float* ip0 = (const float*)in->ip0; const float* ip1 = (const float*)in->ip1; const float* ip2 = (const float*)in->ip2;
    const int W = in->width, H = in->height, P = in->row_pitch_px;
    g_ctx.env = nullptr; g_ctx.sm = nullptr;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t o = ((size_t)y * P + x) * 4;
            float* dst = out + ((size_t)y * W + x) * 4;
            dst[0] = dst[1] = dst[2] = dst[3] = 0.0f;
            const int idx = matIndex(ip2, o);
            if (idx < 0 || idx >= nMats) continue;
            const vqhip_material& mt = mats[idx];
            cbPerObject.materialData.uvScaleOffset = float4(mt.data.uvScaleOffset.x, mt.data.uvScaleOffset.y, mt.data.uvScaleOffset.z, mt.data.uvScaleOffset.w);
            cbPerObject.materialData.textureConfig = mt.data.textureConfig;
            cbPerObject.materialData.normalMapMipBias = mt.data.normalMapMipBias;         // present in the cbuffer; PSMain does not read it (:164 is Sample, not SampleBias)
            bindTex(texDiffuse, mt.texDiffuse); bindTex(texNormals, mt.texNormals);
            auto uvAt = [&](int xx, int yy) {
                const size_t q = ((size_t)yy * P + xx) * 4;
                const float2 uv = float2(ip0[q + 3], ip1[q + 3]) * cbPerObject.materialData.uvScaleOffset.xy + cbPerObject.materialData.uvScaleOffset.zw;
                return vqo::f2{ uv.x, uv.y };
            };
            g_ctx.ddx = { 0, 0 }; g_ctx.ddy = { 0, 0 };
            const int xa = x & ~1, xb = x | 1, ya = y & ~1, yb = y | 1;
            if (xb < W && matIndex(ip2, ((size_t)y * P + xa) * 4) == idx && matIndex(ip2, ((size_t)y * P + xb) * 4) == idx) {
                const vqo::f2 a = uvAt(xa, y), b = uvAt(xb, y); g_ctx.ddx = { b.x - a.x, b.y - a.y };
            }
            if (yb < H && matIndex(ip2, ((size_t)ya * P + x) * 4) == idx && matIndex(ip2, ((size_t)yb * P + x) * 4) == idx) {
                const vqo::f2 a = uvAt(x, ya), b = uvAt(x, yb); g_ctx.ddy = { b.x - a.x, b.y - a.y };
            }
            PSInput In;
            In.position = float4((float)x + 0.5f, (float)y + 0.5f, 0.0f, 1.0f);
            In.WorldSpacePosition = float3(ip0[o], ip0[o + 1], ip0[o + 2]);
            In.WorldSpaceNormal = float3(ip1[o], ip1[o + 1], ip1[o + 2]);
            In.WorldSpaceTangent = float3(ip2[o], ip2[o + 1], ip2[o + 2]);
            In.uv = float2(ip0[o + 3], ip1[o + 3]);
            g_ctx.discarded = false;
            const float4 r = PSMain(In);
            if (g_ctx.discarded) continue;
            dst[0] = r.x; dst[1] = r.y; dst[2] = r.z; dst[3] = r.w;
        }
    return 0;
}

} // extern "C"
