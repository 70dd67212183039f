// ref_convolution.cpp — runs the reference's CubemapConvolution.hlsl (PSMain_DiffuseIrradiance, PSMain_SpecularIrradiance,
// CSMain_BRDFIntegration, with BRDF.hlsl / ShadingMath.hlsl) on the CPU. Same construction and caveats as ref_forward.cpp;
// part of oracle/_ref/libvqref_shaders.so. TEST INFRASTRUCTURE.
//   * The shader's arithmetic (loops, sample generation, weights, mip selection) is the reference's source.
//   * In.CubemapLookDirection is the rasteriser's interpolation of the unit cube's positions: the harness supplies the texel-centre
//     direction of vqo::cube_texel_dir (the oracle's statement of CubemapUtility::CalculateViewMatrix + a 90-degree projection).
//   * texEquirectEnvironmentMap.SampleLevel is the oracle's trilinear-WRAP fetch of the min-filtered mip chain.
#include <vector>

#include "ref_hooks.h"

namespace hlsl {
namespace conv {
#include "CubemapConvolution.hlsl"
} // namespace conv
} // namespace hlsl

namespace {
using namespace hlsl;
using namespace hlsl::conv;
using namespace vqref;
EquirectChain g_chain;
}

extern "C" {

// out: RGBA32F [6][res][res][4]; only texels t in [t0, t1) of the face-major list are written (t1 < 0: all)
int vqref_conv_diffuse(const float* chain, int w0, int h0, int nMips, int res, float* out, int t0, int t1) {
    if (!chain || !out) return -1;
    g_chain = { chain, w0, h0, nMips };
    texEquirectEnvironmentMap.res = &g_chain; texEquirectEnvironmentMap.kind = kTexEquirect;
    const int total = 6 * res * res;
    if (t1 < 0 || t1 > total) t1 = total;
    for (int t = t0; t < t1; ++t) {
        const int f = t / (res * res), y = (t / res) % res, x = t % res;
        const vqo::f3 d = vqo::cube_texel_dir(f, x, y, res);
        GSOut In;
        In.CubemapLookDirection = float3(d.x, d.y, d.z);
        In.layer = (uint)f;
        const float4 c = PSMain_DiffuseIrradiance(In);
        float* p = out + (size_t)t * 4;
        p[0] = c.x; p[1] = c.y; p[2] = c.z; p[3] = c.w;
    }
    return 0;
}

// one mip of the specular cube: out RGBA32F [6][res][res][4]. Roughness / TextureDimensionsLOD0 are CBufferPS, set by the caller
// as EnvironmentMapRendering.cpp:432-440 does (mip / (MIPS-1); the equirect's level-0 size).
int vqref_conv_specular(const float* chain, int w0, int h0, int nMips, int res, float roughness, float dimX, float dimY, int mip, float* out) {
    if (!chain || !out) return -1;
    g_chain = { chain, w0, h0, nMips };
    texEquirectEnvironmentMap.res = &g_chain; texEquirectEnvironmentMap.kind = kTexEquirect;
    Roughness = roughness; TextureDimensionsLOD0 = float2(dimX, dimY); MIP = mip;
    for (int t = 0; t < 6 * res * res; ++t) {
        const int f = t / (res * res), y = (t / res) % res, x = t % res;
        const vqo::f3 d = vqo::cube_texel_dir(f, x, y, res);
        GSOut In;
        In.CubemapLookDirection = float3(d.x, d.y, d.z);
        In.layer = (uint)f;
        const float4 c = PSMain_SpecularIrradiance(In);
        float* p = out + (size_t)t * 4;
        p[0] = c.x; p[1] = c.y; p[2] = c.z; p[3] = c.w;
    }
    return 0;
}

// the same pass for n texels (face, x, y) of ONE mip: out [n][4]
int vqref_conv_specular_texels(const float* chain, int w0, int h0, int nMips, int res, float roughness, float dimX, float dimY, int mip,
                               const int* faces, const int* xs, const int* ys, int n, float* out) {
    if (!chain || !out) return -1;
    g_chain = { chain, w0, h0, nMips };
    texEquirectEnvironmentMap.res = &g_chain; texEquirectEnvironmentMap.kind = kTexEquirect;
    Roughness = roughness; TextureDimensionsLOD0 = float2(dimX, dimY); MIP = mip;
    for (int k = 0; k < n; ++k) {
        if (faces[k] < 0 || faces[k] > 5 || xs[k] < 0 || ys[k] < 0 || xs[k] >= res || ys[k] >= res) return -1;
        const vqo::f3 d = vqo::cube_texel_dir(faces[k], xs[k], ys[k], res);
        GSOut In;
        In.CubemapLookDirection = float3(d.x, d.y, d.z);
        In.layer = (uint)faces[k];
        const float4 c = PSMain_SpecularIrradiance(In);
        float* p = out + (size_t)k * 4;
        p[0] = c.x; p[1] = c.y; p[2] = c.z; p[3] = c.w;
    }
    return 0;
}

// the equirect fetches of ONE texel of the specular pass as the reference's code forms them: taps[k] = (uv.x, uv.y, lod) of the k-th SampleLevel call (samples with
// NdotL <= 0 make none); returns the number of calls (<= 512), out4 = the texel's result
int vqref_conv_specular_taps(const float* chain, int w0, int h0, int nMips, int res, float roughness, float dimX, float dimY, int mip, int face, int x, int y,
                             float* taps, int cap, float* out4) {
    vqref::TapRecorder rec = { taps, cap, 0 };
    vqref::g_tapRecorder = &rec;
    const int rc = vqref_conv_specular_texels(chain, w0, h0, nMips, res, roughness, dimX, dimY, mip, &face, &x, &y, 1, out4);
    vqref::g_tapRecorder = nullptr;
    return rc ? rc : rec.n;
}

// CSMain_BRDFIntegration for n texels (xs[i], ys[i]) of ITS 1024 x 1024 image with ITS 2048 samples: out [n][2]
int vqref_brdf_lut_texels(const int* xs, const int* ys, int n, float* out) {
    static std::vector<float2> img(1024 * 1024);
    texBRDFLUT.data = img.data(); texBRDFLUT.width = 1024; texBRDFLUT.height = 1024;
    for (int i = 0; i < n; ++i) {
        if (xs[i] < 0 || ys[i] < 0 || xs[i] >= 1024 || ys[i] >= 1024) return -1;
        CSMain_BRDFIntegration(uint3((uint)xs[i], (uint)ys[i], 0));
        const float2 v = img[(size_t)ys[i] * 1024 + xs[i]];
        out[2 * i] = v.x; out[2 * i + 1] = v.y;
    }
    return 0;
}

} // extern "C"
