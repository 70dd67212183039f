// vq_internal.h — shared declarations between the C-ABI translation unit (capi.hip) and the kernel
// translation units. Not part of the public boundary (that is include/vqhip.h).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/vqhip.h"

namespace vqk {

// roctx range around the host side of an entry point, named after the GPU marker the reference opens at the call it replaces
// (SCOPED_GPU_MARKER, e.g. SceneRendering.cpp:1630 "RenderSceneColor"), so a rocprofv3 --marker-trace timeline reads like a PIX capture of
// the engine. The marker library (librocprofiler-sdk-roctx / libroctx64) is bound at run time; absent library or VQHIP_ROCTX=0: no-op.
struct Range {
    explicit Range(const char* name);
    ~Range();
    Range(const Range&) = delete;
    Range& operator=(const Range&) = delete;
    bool on;
};

// Tuning / A-B options of one context (vqhip_set_option, include/vqhip.h). Read by the launchers from the context the call came through — never
// from the process environment: getenv racing a host setenv is undefined behaviour in glibc and the engine is heavily threaded. 0 / "" = the default.
struct Options {
    int shadeWg = 0;            // "shade_wg"          : 64 | 128 | 256 lanes per workgroup of the shade kernel (default: by frame size)
    int psmainWaves = 0;        // "psmain_waves"      : 4 | 5 | 6 waves per SIMD of the fused PSMain kernel
    int blurYWgs = 0;           // "blur_y_wgs"        : workgroups of k_blur_y_tonemap_lut
    int postForm = 0;           // "post_form"         : 1 "two" = blur X, then blur Y + tonemap (two kernels) whatever the frame; 2 "chain" = k_post_chain whatever the size
                                //                       (default: k_post_chain for RGBA16F -> RGBA8 frames of >= 2^20 pixels with a per-channel display curve, else two kernels)
    int postStrips = 0;         // "post_strips"       : row strips per column strip of k_post_chain (default: CUs / column strips — one workgroup per CU)
    int lutForm = 0;            // "lut_form"          : 1 "general" (every range test left in)
    int diffuseForm = 0;        // "diffuse_form"      : 1 "texels", 2 "general" (default: footprint records)
    int specularForm = 0;       // "specular_form"     : 1 "general" (default: the branch-free sample with one validity flag)
    int diffuseSeqForm = 0;     // "diffuse_seq_form"  : 1 "lane" (default: k_conv_diffuse_ordered)
};

// Device-resident per-call constant block == the cbuffers b0/b1 of ForwardLighting.hlsl:76-77 plus the
// resource descriptors that replace its SRV tables. Uploaded once per vqhip_forward_lighting call into a
// slot of the context's constant ring (the analogue of the reference's DynamicBufferHeap bump allocation,
// SceneRendering.cpp:432-434,455-457). `extra` lights follow the struct in the same slot.
struct alignas(16) FrameConstants {
    VQ_PerFrameData        perFrame;       // 7120 B
    VQ_PerViewLightingData perView;        // 320 B
    vqhip_envmap           env;            // device pointers
    vqhip_shadowmaps       sm;             // device pointers
    int32_t                hasEnv;
    int32_t                numPointAll;    // numPointLights + numExtraPoint
    int32_t                pow5ExpLog;     // vqhip_set_fresnel_pow: 0 = product (default), 1 = exp2(5*log2 x)
    int32_t                pointFastOK;    // every point light's rangeSq <= 2^60 (or never lit) and every coordinate of its position is 0 or has magnitude in [2^-40, 2^40]:
                                           // preconditions of the unchecked light loop (vq_shade.h)
    float                  hdriSin, hdriCos;   // sin / cos(-fHDRIOffsetInRadians): frame-uniform, taken CORRECTLY ROUNDED on the host (double libm
                                               // rounded to float, capi.hip; contract v5 — the oracle does the same), not the contract's polynomial
    int32_t                pointSkipOK;    // every point light's color*brightness is finite: precondition of the back-facing-light skip (shade.hip)
    int32_t                pointNegZeroAxes;   // bit c: some point light has the coordinate -0.0 on axis c (then a pixel whose P has +0.0 there takes the IEEE loop: (-0) - (+0) = -0)
    // What the spot / directional lights hand to every pixel alike, formed once on the host with the operations the shader would use (round 6):
    // spot[0..20) belong to Lights.spot_lights, spot[20..25) to Lights.spot_casters
    struct alignas(16) DevSpotLight {
        float sdx, sdy, sdz;       // normalize(l.spotDir) in the context's reading (Lighting.hlsl:60): IEEE quotients by the length | v * correctly rounded rsqrt
        float rConeDen;            // RN(1 / (outerConeAngle - innerConeAngle)), the divisor of the penumbra quotient (:71)
        float cbx, cby, cbz;       // l.color * l.brightness (:330)
        int32_t flags;             // bit 0: outer - inner and its reciprocal are normal numbers (the penumbra quotient may use the corrected product);
                                   // bit 1: color * brightness is finite (a light that adds w = +0 times a finite BRDF may be skipped)
    } spot[VQ_NUM_LIGHTS__SPOT + VQ_NUM_SHADOWING_LIGHTS__SPOT];
    float                  dirWi[3];           // normalize(-directional.lightDirection) in the context's reading (Lighting.hlsl:337)
    int32_t                pad6[1];
    // DevPointLight pts[numPointAll] follows
};
using DevSpotLight = FrameConstants::DevSpotLight;
// Non-shadowing point lights as the hot loop reads them (one s_load_dwordx8 per light): point_lights[0..numPointLights)
// followed by the extension array, with the loop-invariant product color*brightness formed once on the host (IEEE
// multiply, identical to the in-shader product).
// `rangeSq` is the exact threshold of the range cull in squared distance: the smallest float t with sqrtf(t) >= range, so that
// (sqrt(dd) < range) == (dd < rangeSq) for every dd (sqrt is correctly rounded, hence monotonic) and culled lights need no sqrt.
struct alignas(16) DevPointLight { float px, py, pz, range; float cbx, cby, cbz, rangeSq; };
static constexpr int    kMaxExtraPointLights = 1024;
static constexpr size_t kConstSlotBytes = (sizeof(FrameConstants) + (VQ_NUM_LIGHTS__POINT + kMaxExtraPointLights) * sizeof(DevPointLight) + 255) & ~(size_t)255;

// the draw's other render targets (vqhip_psmain_targets, include/vqhip.h): pointers NULL = target not bound (wave-uniform tests in the kernels)
struct MrtArgs {
    void* albedo; void* motion;                     // SV_TARGET1 (ForwardLighting.hlsl:383), motion vectors (:387)
    const float4* svCurr; const float4* svPrev;     // PSInput :49-52
    int albedoPitch, motionPitch, svPitch;
    int albedoF32, motionF32;                       // storage: 0 = RGBA16F / RG16F (the reference's formats), 1 = RGBA32F / RG32F
};
struct ShadeArgs {
    const float4* gb0; const float4* gb1; const float4* gb2; const float4* gb3;
    void* out;
    const FrameConstants* fc;   // device
    int width, height, pitch, outPitch;
    int arithDxc;               // vqhip_set_arithmetic: 0 literal reading, 1 DXC reading (selects the kernel instantiation)
    MrtArgs mrt;                // last: the kernels without extra targets keep their argument offsets
};

// G-buffer producer (§8f.1): per-call constants in a ring slot — cbPerObject.materialData + descriptor tables of every
// material referenced by the interpolant planes.
struct alignas(16) GbufConstants {
    float      ambient;          // cbPerFrame.fAmbientLightingFactor
    int32_t    numMaterials;
    vqhip_ssao ssao;             // device pointer or NULL
    int32_t    arithDxc;         // vqhip_set_arithmetic: the reading of dot / normalize / length in the producer (wave-uniform run-time flag)
    int32_t    pad[1];
    vqhip_material mats[1];      // numMaterials entries
};
static constexpr int kMaxMaterials = (int)((kConstSlotBytes - offsetof(GbufConstants, mats)) / sizeof(vqhip_material));
struct GbufArgs {
    const float4* ip0; const float4* ip1; const float4* ip2;
    float4* gb0; float4* gb1; float4* gb2; float4* gb3;
    const GbufConstants* gc;     // device
    int width, height, pitch, outPitch;
};
hipError_t launch_gbuffer_from_materials(hipStream_t s, const GbufArgs& a);
hipError_t launch_scene_normals_from_materials(hipStream_t s, const GbufArgs& a, void* out, int outFmt);   // a.gb* unused, a.outPitch = pitch of `out`
struct FrameConstants;
hipError_t launch_forward_from_materials(hipStream_t s, const GbufArgs& a, const FrameConstants* fc, bool hasEnv, bool hasCasters, void* out, int outPitch, int outFmt, int arithDxc, const Options& opt, const MrtArgs& mrt);
hipError_t launch_mip_box_rgba8(hipStream_t s, const void* src, void* dst, int sw, int sh, int dw, int dh);
hipError_t launch_skydome(hipStream_t s, const float4* eq0, int w0, int h0, const VQ_SkydomeParams& sp, const float4* cov, int covPitch,
                          void* color, int W, int H, int pitch, int fmt);

struct UnlitColors { float4 c[VQHIP_MAX_UNLIT_COLORS]; };
hipError_t launch_unlit_composite(hipStream_t s, const float4* cov, int covPitch, const UnlitColors& cols, int n, void* color, int W, int H, int pitch, int fmt);

// HDRI ingest (§8f.3, hdri.hip): host-side header parse / run expansion (0 or -1 with *err set), device conversion
int hdr_parse_header(const uint8_t* f, size_t n, int* w, int* h, size_t* off, const char** err);
int hdr_expand_rgbe(const uint8_t* f, size_t n, size_t off, int w, int h, uint8_t* rgbe, const char** err);
hipError_t launch_rgbe_to_rgba32f(hipStream_t s, const void* rgbe, void* out, size_t n);
int hdr_walk_runs(const uint8_t* f, size_t n, size_t off, int w, int h, uint32_t* planeOff, const char** err);
bool hdr_expand_fits(const uint32_t* planeOff, int w, int h, int ldsLimit, int* encCap, int* pitch, int* ldsBytes);
hipError_t launch_hdr_expand(hipStream_t s, const void* file, const void* planeOff, void* out, int w, int h, int encCap, int pitch, int ldsBytes);
hipError_t launch_downsize_box(hipStream_t s, const void* src, void* dst, int sw, int sh, int k);

// FSR 1.0 (§8f.4, fsr.hip)
hipError_t launch_fsr_easu(hipStream_t s, const void* in, int inW, int inH, int inFmt, const uint32_t* con16, void* out, int outW, int outH, int outFmt);
hipError_t launch_fsr_rcas(hipStream_t s, const void* in, void* out, int W, int H, const uint32_t* con4, int inFmt, int outFmt);
hipError_t launch_visualize(hipStream_t s, const void* in, void* out, int W, int H, const VQ_VizParams& p, int inFmt, int outFmt);
hipError_t launch_apply_reflections(hipStream_t s, const void* refl, const void* boundingVolumes, void* scene, int W, int H, int fmt);   // boundingVolumes NULL: plain permutation

// SSR environment fallback (§8f.4, ssr.hip): what the kernel reads of VQ_SSSRConstants + the resource descriptors, by value
struct SsrArgs {
    const void* scene; const float* depth; const void* normals; void* out; uint8_t* outRoughness;
    int width, height, scenePitch, depthPitch, normalPitch, outPitch;
    VQ_matrix invProj, view, invView;
    float rot[3][3];                      // upper-left 3x3 of envMapRotation
    float invDimX, invDimY, roughnessThreshold, mipCount;
    int pow5ExpLog, arithDxc;
    vqhip_envmap env;
};
hipError_t launch_ssr_env_fallback(hipStream_t s, const SsrArgs& a, int sceneFmt, int normalFmt, int outFmt);

// launchers (each returns the hipError_t of the launch)
hipError_t launch_forward_lighting(hipStream_t s, const ShadeArgs& a, bool hasEnv, bool hasCasters, int outFmt, const Options& opt);
hipError_t launch_blur_x(hipStream_t s, const void* in, void* out, int W, int H, int fmt, const Options& opt);
hipError_t launch_blur_y(hipStream_t s, const void* in, void* out, const void* haloTop, const void* haloBottom, int haloRows, int W, int H, int fmt);
bool tonemap_uses_lut(const VQ_TonemapperParams& p, int inFmt, int outFmt, size_t nPixels);
bool blur_y_tonemap_uses_lut(const VQ_TonemapperParams& p, int blurFmt, int outFmt, size_t nPixels);
hipError_t launch_tonemap_lut_build(hipStream_t s, void* table, const VQ_TonemapperParams& p, int outFmt);
hipError_t launch_tonemap(hipStream_t s, const void* in, void* out, int W, int H, const VQ_TonemapperParams& p, int inFmt, int outFmt, const void* lutTable, const Options& opt);
hipError_t launch_blur_y_tonemap(hipStream_t s, const void* in, void* out, const void* haloTop, const void* haloBottom, int haloRows, int W, int H,
                                 const VQ_TonemapperParams& p, int fmt, int outFmt, const void* lutTable, const Options& opt);
bool post_chain_applies(const VQ_TonemapperParams& p, int inFmt, int outFmt, int W, int H, const Options& opt);
hipError_t launch_post_chain(hipStream_t s, const void* in, void* out, const void* haloTop, const void* haloBottom, int haloRows, int W, int H,
                             const void* lutTable, int nCUs, const Options& opt);
hipError_t launch_brdf_lut(hipStream_t s, void* out, int size, int samples, int fmt, int pow5ExpLog, const Options& opt);
hipError_t launch_mip_min(hipStream_t s, const float4* src, float4* dst, int sw, int sh, int dw, int dh);
size_t conv_diffuse_record_bytes(int w0, int h0, int nMips);
hipError_t launch_conv_diffuse_tables(hipStream_t s, const float4* chain, int w0, int h0, int nMips, int res,
                                      const float* phis, int nPhi, const float* thetas, int nTheta, int order, void* out, int fmt, void* recBuf, const Options& opt);
hipError_t launch_conv_specular_all(hipStream_t s, const float4* chain, int w0, int h0, int nMips, int res0, int MIPS, int order, void* out, int fmt, const Options& opt);

} // namespace vqk
