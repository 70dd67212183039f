// ref_hooks.cpp — texture fetch hooks of the CPU-run reference shaders (see ref_hooks.h). TEST INFRASTRUCTURE.
#include "ref_hooks.h"

namespace vqref { Ctx g_ctx;
TapRecorder* g_tapRecorder = nullptr; }

namespace hlsl {
using namespace vqref;

float4 vqref_sample_2d(const Texture2D& t, const SamplerState&, float2 uv, int mode, float arg) {
    switch (t.kind) {
    case kTexMaterial: {
        const vqhip_texture2d* tx = (const vqhip_texture2d*)t.res;
        if (mode == kSampleLevel) {      // explicit LOD = the level as the bias of a unit (one texel per pixel) footprint
            const float w = (float)tx->width, h = (float)tx->height;
            const vqo::f4 r = vqo::sample_material_tex(*tx, { uv.x, uv.y }, { 1.0f / w, 0 }, { 0, 1.0f / h }, arg);
            return float4(r.x, r.y, r.z, r.w);
        }
        const vqo::f4 r = vqo::sample_material_tex(*tx, { uv.x, uv.y }, g_ctx.ddx, g_ctx.ddy, mode == kSampleBias ? arg : 0.0f);
        return float4(r.x, r.y, r.z, r.w);
    }
    case kTexSSAO: {
        const vqhip_ssao* s = (const vqhip_ssao*)t.res;
        const float a = vqo::fetch_r8_point_wrap((const uint8_t*)s->texels, s->width, s->height, uv.x, uv.y);
        return float4(a, a, a, a);
    }
    case kTexOne: return float4(1, 1, 1, 1);
    case kTexLUT: {
        const vqhip_envmap* e = g_ctx.env;
        const vqo::f2 r = vqo::sample_2d_rg16f_clamp((const uint16_t*)e->brdf_lut, e->lut_size, e->lut_size, uv.x, uv.y);
        return float4(r.x, r.y, 0, 0);
    }
    case kTexShadowDir: {
        const float d = vqo::fetch_point_wrap(g_ctx.sm->directional, g_ctx.sm->dir_dim, uv.x, uv.y);
        return float4(d, d, d, d);
    }
    case kTexEquirect: {
        const EquirectChain* c = (const EquirectChain*)t.res;
        if (g_tapRecorder) {
            TapRecorder* r = g_tapRecorder;
            if (r->n < r->cap) { float* o = r->out + 3 * (size_t)r->n; o[0] = uv.x; o[1] = uv.y; o[2] = mode == kSampleLevel ? arg : 0.0f; }
            ++r->n;
        }
        const vqo::f4 r = vqo::sample_equirect_lod(c->chain, c->w0, c->h0, c->nMips, uv.x, uv.y, mode == kSampleLevel ? arg : 0.0f);
        return float4(r.x, r.y, r.z, r.w);
    }
    default: return float4(0, 0, 0, 0);                                  // null SRV
    }
}
float4 vqref_sample_cube(const TextureCube& t, const SamplerState&, float3 d, int mode, float arg) {
    const vqhip_envmap* e = g_ctx.env;
    if (t.kind == kCubeDiffuse) {                                         // single-mip cube: every LOD is level 0
        const vqo::f4 r = vqo::sample_cube_rgba16f((const uint16_t*)e->diffuse_cube, e->diffuse_res, { d.x, d.y, d.z });
        return float4(r.x, r.y, r.z, r.w);
    }
    if (t.kind == kCubeSpecular) {                                        // SampleLevel on the mip-major specular cube, lod clamped by the sampler; an integral
        const float lod = mode == kSampleLevel ? arg : 0.0f;              // lod (Lighting.hlsl:375) reads one level, a fractional one (SSR, ClassifyReflectionTiles.hlsl:89) two
        const vqo::f4 r = vqo::sample_cube_lod_rgba16f((const uint16_t*)e->specular_cube, e->spec_res0, e->spec_mips, { d.x, d.y, d.z }, lod);
        return float4(r.x, r.y, r.z, r.w);
    }
    return float4(0, 0, 0, 0);
}
float4 vqref_sample_2d_array(const Texture2DArray& t, const SamplerState&, float3 uvw) {
    if (t.kind != kArrSpot) return float4(0, 0, 0, 0);
    const int dim = g_ctx.sm->spot_dim;
    const float d = vqo::fetch_point_wrap(g_ctx.sm->spot + (size_t)(int)uvw.z * dim * dim, dim, uvw.x, uvw.y);
    return float4(d, d, d, d);
}
float4 vqref_sample_cube_array(const TextureCubeArray& t, const SamplerState&, float4 dw) {
    if (t.kind != kArrPoint) return float4(0, 0, 0, 0);
    const int dim = g_ctx.sm->point_dim;
    const float d = vqo::fetch_cube_point(g_ctx.sm->point + (size_t)(int)dw.w * 6 * dim * dim, dim, { dw.x, dw.y, dw.z });
    return float4(d, d, d, d);
}
float4 vqref_load_2d(const Texture2D& t, int x, int y, int) {
    if (t.kind != kTexImage) return float4(0, 0, 0, 0);
    const Image* im = (const Image*)t.res;
    if (x < 0 || y < 0 || x >= im->width || y >= im->height) return float4(0, 0, 0, 0);      // out-of-bounds loads return 0
    const float* p = im->rgba + ((size_t)y * im->width + x) * 4;
    return float4(p[0], p[1], p[2], p[3]);
}
// Gather with a clamp sampler: the 2x2 footprint whose top-left texel is floor(uv*size - 0.5), texels clamped to the image
float4 vqref_gather_2d(const Texture2D& t, const SamplerState&, float2 uv, int ch) {
    if (t.kind != kTexImage) return float4(0, 0, 0, 0);
    const Image* im = (const Image*)t.res;
    const int x0 = vqo::f2i_floor(uv.x * (float)im->width - 0.5f), y0 = vqo::f2i_floor(uv.y * (float)im->height - 0.5f);
    auto tex = [&](int x, int y) {
        x = x < 0 ? 0 : (x > im->width - 1 ? im->width - 1 : x); y = y < 0 ? 0 : (y > im->height - 1 ? im->height - 1 : y);
        return im->rgba[((size_t)y * im->width + x) * 4 + ch];
    };
    return float4(tex(x0, y0 + 1), tex(x0 + 1, y0 + 1), tex(x0 + 1, y0), tex(x0, y0));
}
uint f32tof16(float f) { return vqo::f32_to_f16(f); }
float f16tof32(uint h) { return vqo::f16_to_f32((uint16_t)h); }
void vqref_dims_2d(const Texture2D& t, uint* w, uint* h) {
    if (t.kind == kTexImage) { const Image* im = (const Image*)t.res; *w = (uint)im->width; *h = (uint)im->height; }
    else { *w = 0; *h = 0; }
}

} // namespace hlsl
