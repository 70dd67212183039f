// ref_hooks.h — what the fixed-function hardware / the descriptor tables provide around one shader invocation of the
// reference's HLSL when it runs on the CPU (oracle/_ref/libvqref_shaders.so). TEST INFRASTRUCTURE, see hlsl_shim.h.
// Texture FETCHES have no source in the reference: ref_hooks.cpp implements them with the oracle's sampling contract
// (vqo_sampling.h); which fetch a Texture* object performs is selected by its `kind`.
#pragma once
#include "../../include/vqhip.h"
#include "../vqo_math.h"
#include "../vqo_sampling.h"
#include "hlsl_shim.h"

namespace vqref {

enum TexKind {
    kTexNull = 0,
    kTexMaterial,      // res = const vqhip_texture2d* : RGBA8_UNORM mip chain, trilinear WRAP, implicit LOD from ctx.ddx/ddy
    kTexSSAO,          // res = const vqhip_ssao*      : R8_UNORM, POINT_WRAP
    kTexOne,           // constant 1 (the engine's white default texture)
    kTexLUT,           // ctx.env->brdf_lut RG16F, bilinear CLAMP
    kTexShadowDir,     // ctx.sm->directional R32F, POINT_WRAP
    kTexEquirect,      // res = const EquirectChain*   : RGBA32F mip chain, trilinear WRAP, explicit LOD
    kTexImage,         // res = const Image*           : Load / operator[] of a float4 image (compute shaders)
    kCubeDiffuse, kCubeSpecular, kArrSpot, kArrPoint
};
struct EquirectChain { const float* chain; int w0, h0, nMips; };
// optional recorder of the equirect fetches (uv.x, uv.y, lod per SampleLevel call, in call order): tests list the taps of a texel as the REFERENCE'S code formed them
struct TapRecorder { float* out; int cap, n; };
extern TapRecorder* g_tapRecorder;
struct Image { const float* rgba; int width, height; };       // RGBA32F row-major; out-of-range loads return 0 (D3D)

struct Ctx {
    vqo::f2 ddx{ 0, 0 }, ddy{ 0, 0 };
    const vqhip_envmap* env = nullptr;
    const vqhip_shadowmaps* sm = nullptr;
    bool discarded = false;             // set by the `discard` statement of an ENABLE_ALPHA_MASK build (ref_forward.cpp)
};
extern Ctx g_ctx;

} // namespace vqref
