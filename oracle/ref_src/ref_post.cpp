// ref_post.cpp — runs the reference's post / compose shaders on the CPU: GaussianBlur.hlsl (CSMain_X, CSMain_Y),
// Tonemapper.hlsl (CSMain, with HDR.hlsl), Skydome.hlsl (PSMain), Visualization.hlsl (CSMain), ApplyReflections.hlsl (CSMain).
// Same construction and caveats as ref_forward.cpp; part of oracle/_ref/libvqref_shaders.so. TEST INFRASTRUCTURE.
// Images cross this boundary as RGBA32F VALUES: the storage-format conversion of a UAV store (RNE to fp16, UNORM8 quantisation)
// is fixed-function hardware, the tests apply the oracle's conversion to these values before comparing stored texels.
#include <cstring>
#include <vector>

#include "ref_hooks.h"

namespace hlsl {
namespace blur {
#include "GaussianBlur.hlsl"
}
namespace tonemap {
#include "Tonemapper.hlsl"
}
#undef _SHADING_MATH_H
namespace sky {
#include "Skydome.hlsl"
}
namespace reflections {
#include "ApplyReflections.hlsl"
}
#define COMPOSITE_BOUNDING_VOLUMES 1          // "[PSO] ApplyReflectionsAndBoundingVolumes" (ApplyReflections.cpp:82-86): the same source, second permutation
namespace reflections_bv {
#include "ApplyReflections.hlsl"
}
#undef COMPOSITE_BOUNDING_VOLUMES
namespace viz {
#include "Visualization.hlsl"
}
namespace unlit {
#include "Unlit.hlsl"
}
} // namespace hlsl

using namespace hlsl;
using namespace vqref;

extern "C" {

// one separable pass of the 21-tap blur: direction 0 = CSMain_X, 1 = CSMain_Y. in/out RGBA32F [H][W][4]
int vqref_blur_pass(const float* in, int W, int H, int direction, float* out) {
    if (!in || !out) return -1;
    const Image src{ in, W, H };
    blur::texColorInput.res = &src; blur::texColorInput.kind = kTexImage;
    std::vector<float4> dst((size_t)W * H);
    blur::texColorOutput.data = dst.data(); blur::texColorOutput.width = W; blur::texColorOutput.height = H;
    blur::iImageSize = int2(W, H);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const uint3 id((uint)x, (uint)y, 0), z(0, 0, 0);
            if (direction == 0) blur::CSMain_X(z, z, id); else blur::CSMain_Y(z, z, id);
        }
    std::memcpy(out, dst.data(), sizeof(float4) * dst.size());
    return 0;
}

int vqref_tonemap(const float* in, int W, int H, const VQ_TonemapperParams* p, float* out) {
    if (!in || !out || !p) return -1;
    const Image src{ in, W, H };
    tonemap::texColorInput.res = &src; tonemap::texColorInput.kind = kTexImage;
    std::vector<float4> dst((size_t)W * H);
    tonemap::texColorOutput.data = dst.data(); tonemap::texColorOutput.width = W; tonemap::texColorOutput.height = H;
    tonemap::ContentColorSpaceEnum = p->ContentColorSpaceEnum; tonemap::OutputDisplayCurveEnum = p->OutputDisplayCurveEnum;
    tonemap::DisplayReferenceBrightnessLevel = p->DisplayReferenceBrightnessLevel; tonemap::ToggleGammaCorrection = p->ToggleGammaCorrection;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) { const uint3 z(0, 0, 0); tonemap::CSMain(z, z, uint3((uint)x, (uint)y, 0)); }
    std::memcpy(out, dst.data(), sizeof(float4) * dst.size());
    return 0;
}

// Skydome PSMain per pixel. CubemapLookDirection (the rasteriser's interpolant of the cube's normalised vertex positions) is
// supplied as the pixel's view ray from the camera basis, the same statement the oracle uses (vqo_skydome).
int vqref_skydome(const float* equirect0, int w0, int h0, const VQ_SkydomeParams* sp, int W, int H, float* out) {
    if (!equirect0 || !sp || !out) return -1;
    const EquirectChain chain{ equirect0, w0, h0, 1 };
    sky::texEquirectEnvironmentMap.res = &chain; sky::texEquirectEnvironmentMap.kind = kTexEquirect;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float nx = vqo::div_(2.0f * ((float)x + 0.5f), (float)W) - 1.0f, ny = 1.0f - vqo::div_(2.0f * ((float)y + 0.5f), (float)H);
            const float a = nx * sp->tanHalfFovX, b = ny * sp->tanHalfFovY;
            sky::PSInput in;
            in.CubemapLookDirection = float3(vqo::fma_(b, sp->up.x, vqo::fma_(a, sp->right.x, sp->forward.x)),
                                             vqo::fma_(b, sp->up.y, vqo::fma_(a, sp->right.y, sp->forward.y)),
                                             vqo::fma_(b, sp->up.z, vqo::fma_(a, sp->right.z, sp->forward.z)));
            const float4 c = sky::PSMain(in);
            float* o = out + ((size_t)y * W + x) * 4;
            o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = c.w;
        }
    return 0;
}

int vqref_visualize(const float* in, int W, int H, const VQ_VizParams* p, float* out) {
    if (!in || !out || !p) return -1;
    const Image src{ in, W, H };
    viz::texIn.res = &src; viz::texIn.kind = kTexImage;
    std::vector<float4> dst((size_t)W * H);
    viz::texOut.data = dst.data(); viz::texOut.width = W; viz::texOut.height = H;
    viz::iDrawMode = p->iDrawMode; viz::iUnpackNormals = p->iUnpackNormals; viz::fInputStrength = p->fInputStrength;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) { const uint3 z(0, 0, 0); viz::CSMain(z, z, uint3((uint)x, (uint)y, 0)); }
    std::memcpy(out, dst.data(), sizeof(float4) * dst.size());
    return 0;
}

// Unlit.hlsl: VSMain carries the cbuffer colour to PSMain, PSMain returns it. One invocation for cbuffer colour `rgba`.
void vqref_unlit_color(const float* rgba, float* out4) {
    unlit::color = float4(rgba[0], rgba[1], rgba[2], rgba[3]);
    unlit::VSInput v;
    const unlit::PSInput in = unlit::VSMain(v, 0);
    const float4 c = unlit::PSMain(in);
    out4[0] = c.x; out4[1] = c.y; out4[2] = c.z; out4[3] = c.w;
}

// scene (RGBA32F values, in place) += reflections
int vqref_apply_reflections(const float* refl, float* scene, int W, int H) {
    if (!refl || !scene) return -1;
    const Image src{ refl, W, H };
    reflections::TexReflectionRadiance.res = &src; reflections::TexReflectionRadiance.kind = kTexImage;
    reflections::TexSceneColor.data = (float4*)scene; reflections::TexSceneColor.width = W; reflections::TexSceneColor.height = H;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) { const uint3 z(0, 0, 0); reflections::CSMain(z, z, uint3((uint)x, (uint)y, 0)); }
    return 0;
}

// the COMPOSITE_BOUNDING_VOLUMES permutation: scene = BV.rgb * BV.a + (scene + reflections) * (1 - BV.a), alpha = BV.a
int vqref_apply_reflections_bv(const float* refl, const float* bv, float* scene, int W, int H) {
    if (!refl || !bv || !scene) return -1;
    const Image src{ refl, W, H }, bvI{ bv, W, H };
    reflections_bv::TexReflectionRadiance.res = &src; reflections_bv::TexReflectionRadiance.kind = kTexImage;
    reflections_bv::TexBoundingVolumes.res = &bvI; reflections_bv::TexBoundingVolumes.kind = kTexImage;
    reflections_bv::TexSceneColor.data = (float4*)scene; reflections_bv::TexSceneColor.width = W; reflections_bv::TexSceneColor.height = H;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) { const uint3 z(0, 0, 0); reflections_bv::CSMain(z, z, uint3((uint)x, (uint)y, 0)); }
    return 0;
}

} // extern "C"
