/*
 * vqhip.h — C ABI of the MI355X-native offscreen PBR-shading / IBL / post path.
 *
 * This is the drop-in boundary: plain C, pointers + sizes, no torch / HIP types.
 * Every entry point cites the reference call it replaces (paths relative to the VQEngine tree).
 *
 * Conventions
 *   - every `const void* / void*` image argument is a DEVICE pointer owned by the caller;
 *   - every `VQ_*` struct pointer is a HOST pointer (the reference fills these on the CPU and
 *     bump-allocates them into an upload heap: Source/Renderer/Rendering/SceneRendering.cpp:429-467);
 *   - every call enqueues on `stream` (a hipStream_t passed as void*; NULL = the null stream) and
 *     returns without synchronising, mirroring "record into a command list";
 *   - return value: VQHIP_OK (0) or a negative vqhip_status; vqhip_last_error() gives the text.
 *     (The reference returns void and asserts/logs: e.g. EnvironmentMapRendering.cpp:139, Renderer.cpp:871.)
 *   - images are dense row-major, `row_pitch_px` pixels between rows where stated, otherwise == width.
 *     An image (or row tile) has at most 65 535 rows: several kernels map rows to grid.y; taller inputs fail with VQHIP_ERR_HIP.
 *
 * Struct layouts are byte-for-byte those of Shaders/LightingConstantBufferData.h (namespace
 * VQ_SHADER_DATA); the static asserts below pin the offsets derived in SURVEY.md §8(b).
 */
#ifndef VQHIP_H
#define VQHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__cplusplus)
#define VQHIP_STATIC_ASSERT(c, m) static_assert(c, m)
#define VQHIP_ALIGNAS(n) alignas(n)
#else
#define VQHIP_STATIC_ASSERT(c, m) _Static_assert(c, m)
#define VQHIP_ALIGNAS(n) _Alignas(n)
#endif

#define VQHIP_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------------
 * status / formats
 * ---------------------------------------------------------------------------------------------- */
typedef enum vqhip_status {
    VQHIP_OK                = 0,
    VQHIP_ERR_INVALID_ARG   = -1,
    VQHIP_ERR_HIP           = -2,   /* a HIP runtime call failed; see vqhip_last_error()            */
    VQHIP_ERR_UNSUPPORTED   = -3,   /* format / parameter combination not implemented              */
    VQHIP_ERR_NO_DEVICE     = -4,   /* no gfx950 device visible — there is NO CPU fallback          */
    VQHIP_ERR_RCCL          = -5    /* RCCL could not be loaded or a collective call failed (row-tiled multi-GPU mode) */
} vqhip_status;

/* Storage formats of the reference render targets:
 *   scene colour / blur / env cubemaps : DXGI_FORMAT_R16G16B16A16_FLOAT  (RenderResources.cpp:40,144-161,221-243;
 *                                                                          EnvironmentMapRendering.cpp:32-53)
 *   SDR tonemapper output              : DXGI_FORMAT_R8G8B8A8_UNORM      (RenderResources.cpp:41,245-261)
 *   BRDF integration LUT               : DXGI_FORMAT_R16G16_FLOAT        (Renderer.cpp:1026-1032)
 *   HDRI equirect                      : DXGI_FORMAT_R32G32B32A32_FLOAT  (TextureManager.cpp:590-592)
 * RGBA32F / RG32F variants are offered for every output so parity can also be judged before the
 * storage rounding. fp32 -> fp16 is round-to-nearest-even, fp32 -> UNORM8 is trunc(sat(x)*255 + 0.5). */
typedef enum vqhip_format {
    VQHIP_FMT_RGBA32F     = 0,
    VQHIP_FMT_RGBA16F     = 1,
    VQHIP_FMT_RGBA8_UNORM = 2,
    VQHIP_FMT_RG16F       = 3,
    VQHIP_FMT_RG32F       = 4,
    VQHIP_FMT_R10G10B10A2_UNORM = 5   /* Tex_SceneNormals (RenderResources.cpp:185-197): output of vqhip_scene_normals_from_materials, input of vqhip_ssr_environment_fallback */
} vqhip_format;

/* ------------------------------------------------------------------------------------------------
 * VQ_SHADER_DATA mirror  (Shaders/LightingConstantBufferData.h:39-186, Shaders/VQPlatform.h:21-41)
 * ---------------------------------------------------------------------------------------------- */
#define VQ_NUM_LIGHTS__POINT                 100   /* LightingConstantBufferData.h:39 */
#define VQ_NUM_LIGHTS__SPOT                  20    /* :40 */
#define VQ_NUM_SHADOWING_LIGHTS__POINT       5     /* :42 */
#define VQ_NUM_SHADOWING_LIGHTS__SPOT        5     /* :43 */
#define VQ_NUM_SHADOWING_LIGHTS__DIRECTIONAL 1     /* :44 */

typedef struct VQ_float2 { float x, y; } VQ_float2;
typedef struct VQ_float3 { float x, y, z; } VQ_float3;
typedef struct VQ_float4 { float x, y, z, w; } VQ_float4;
/* DirectX::XMMATRIX: 4 rows of 4 floats, 16-byte aligned, row-major on the CPU. HLSL reads a cbuffer
 * `matrix` column-major, so HLSL `mul(M, v)` == row-vector v * M_cpu  (SURVEY.md §8b). */
typedef struct VQ_matrix { VQHIP_ALIGNAS(16) float m[4][4]; } VQ_matrix;

typedef struct VQ_PointLight {          /* LightingConstantBufferData.h:50-61 */
    VQ_float3 position;    float range;
    VQ_float3 color;       float brightness;
    VQ_float3 attenuation; float depthBias;
} VQ_PointLight;

typedef struct VQ_SpotLight {           /* :63-78 (its "48 bytes" comment is wrong: 64) */
    VQ_float3 position;    float outerConeAngle;
    VQ_float3 color;       float brightness;
    VQ_float3 spotDir;     float depthBias;
    float innerConeAngle;  float range; float dummy1; float dummy2;
} VQ_SpotLight;

typedef struct VQ_DirectionalLight {    /* :80-90 */
    VQ_float3 lightDirection; float brightness;
    VQ_float3 color;          float depthBias;
    int32_t shadowing;        int32_t enabled;
} VQ_DirectionalLight;

typedef struct VQ_SceneLighting {       /* :92-109 */
    int32_t numPointLights, numSpotLights, numPointCasters, numSpotCasters;
    VQ_DirectionalLight directional;
    VQ_matrix     shadowViewDirectional;
    VQ_PointLight point_lights [VQ_NUM_LIGHTS__POINT];
    VQ_PointLight point_casters[VQ_NUM_SHADOWING_LIGHTS__POINT];
    VQ_SpotLight  spot_lights  [VQ_NUM_LIGHTS__SPOT];
    VQ_SpotLight  spot_casters [VQ_NUM_SHADOWING_LIGHTS__SPOT];
    VQ_matrix     shadowViews  [VQ_NUM_SHADOWING_LIGHTS__SPOT];
} VQ_SceneLighting;

typedef struct VQ_PerFrameData {        /* :164-172 ; filled at SceneRendering.cpp:429-450 */
    VQ_SceneLighting Lights;
    VQ_float2 f2PointLightShadowMapDimensions;
    VQ_float2 f2SpotLightShadowMapDimensions;
    VQ_float2 f2DirectionalLightShadowMapDimensions;
    float fAmbientLightingFactor;
    float fHDRIOffsetInRadians;
} VQ_PerFrameData;

typedef struct VQ_PerViewLightingData { /* :173-186 ; filled at SceneRendering.cpp:452-467 */
    VQ_matrix matView, matViewToWorld, matProjInverse;
    VQ_float4 WorldFrustumPlanes[6];
    VQ_float3 CameraPosition; float MaxEnvMapLODLevels;
    VQ_float2 ScreenDimensions;
    int32_t   EnvironmentMapDiffuseOnlyIllumination;
    float     pad1;
} VQ_PerViewLightingData;

typedef struct VQ_MaterialData {        /* :126-143 ; Material::GetCBufferData Material.h:120-127 */
    VQHIP_ALIGNAS(16) VQ_float3 diffuse; float alpha;
    VQ_float3 emissiveColor;  float emissiveIntensity;
    VQ_float3 specular;       float normalMapMipBias;
    VQ_float4 uvScaleOffset;
    float roughness, metalness, displacement, textureConfig;
} VQ_MaterialData;

/* Tonemapper.hlsl:98-104 cbuffer == first 16 bytes of FPostProcessParameters::FTonemapper
 * (Source/Engine/PostProcess/PostProcess.h:84-91). Enums: Source/Renderer/Rendering/HDR.h:78-95. */
typedef enum VQ_EColorSpace   { VQ_COLOR_SPACE_REC_709 = 0, VQ_COLOR_SPACE_REC_2020 = 1 } VQ_EColorSpace;
typedef enum VQ_EDisplayCurve { VQ_DISPLAY_CURVE_SRGB = 0, VQ_DISPLAY_CURVE_ST2084 = 1, VQ_DISPLAY_CURVE_LINEAR = 2 } VQ_EDisplayCurve;
typedef struct VQ_TonemapperParams {
    int32_t ContentColorSpaceEnum;            /* default REC_709 */
    int32_t OutputDisplayCurveEnum;           /* default sRGB    */
    float   DisplayReferenceBrightnessLevel;  /* default 200.0f  */
    int32_t ToggleGammaCorrection;            /* default 1       */
} VQ_TonemapperParams;

/* GaussianBlur.hlsl:65-68 cbuffer == FPostProcessParameters::FBlurParams (PostProcess.h:92-96). */
typedef struct VQ_BlurParams { int32_t iImageSizeX, iImageSizeY; } VQ_BlurParams;

VQHIP_STATIC_ASSERT(sizeof(VQ_PointLight) == 48, "PointLight");
VQHIP_STATIC_ASSERT(offsetof(VQ_PointLight, range) == 12 && offsetof(VQ_PointLight, color) == 16 &&
                    offsetof(VQ_PointLight, brightness) == 28 && offsetof(VQ_PointLight, attenuation) == 32 &&
                    offsetof(VQ_PointLight, depthBias) == 44, "PointLight offsets");
VQHIP_STATIC_ASSERT(sizeof(VQ_SpotLight) == 64, "SpotLight");
VQHIP_STATIC_ASSERT(offsetof(VQ_SpotLight, outerConeAngle) == 12 && offsetof(VQ_SpotLight, spotDir) == 32 &&
                    offsetof(VQ_SpotLight, depthBias) == 44 && offsetof(VQ_SpotLight, innerConeAngle) == 48 &&
                    offsetof(VQ_SpotLight, range) == 52, "SpotLight offsets");
VQHIP_STATIC_ASSERT(sizeof(VQ_DirectionalLight) == 40, "DirectionalLight");
VQHIP_STATIC_ASSERT(offsetof(VQ_DirectionalLight, shadowing) == 32 && offsetof(VQ_DirectionalLight, enabled) == 36, "DirectionalLight offsets");
VQHIP_STATIC_ASSERT(sizeof(VQ_SceneLighting) == 7088, "SceneLighting");
VQHIP_STATIC_ASSERT(offsetof(VQ_SceneLighting, directional) == 16 && offsetof(VQ_SceneLighting, shadowViewDirectional) == 64 &&
                    offsetof(VQ_SceneLighting, point_lights) == 128 && offsetof(VQ_SceneLighting, point_casters) == 4928 &&
                    offsetof(VQ_SceneLighting, spot_lights) == 5168 && offsetof(VQ_SceneLighting, spot_casters) == 6448 &&
                    offsetof(VQ_SceneLighting, shadowViews) == 6768, "SceneLighting offsets");
VQHIP_STATIC_ASSERT(sizeof(VQ_PerFrameData) == 7120, "PerFrameData");
VQHIP_STATIC_ASSERT(offsetof(VQ_PerFrameData, f2PointLightShadowMapDimensions) == 7088 &&
                    offsetof(VQ_PerFrameData, f2SpotLightShadowMapDimensions) == 7096 &&
                    offsetof(VQ_PerFrameData, f2DirectionalLightShadowMapDimensions) == 7104 &&
                    offsetof(VQ_PerFrameData, fAmbientLightingFactor) == 7112 &&
                    offsetof(VQ_PerFrameData, fHDRIOffsetInRadians) == 7116, "PerFrameData offsets");
VQHIP_STATIC_ASSERT(sizeof(VQ_PerViewLightingData) == 320, "PerViewLightingData");
VQHIP_STATIC_ASSERT(offsetof(VQ_PerViewLightingData, WorldFrustumPlanes) == 192 && offsetof(VQ_PerViewLightingData, CameraPosition) == 288 &&
                    offsetof(VQ_PerViewLightingData, MaxEnvMapLODLevels) == 300 && offsetof(VQ_PerViewLightingData, ScreenDimensions) == 304 &&
                    offsetof(VQ_PerViewLightingData, EnvironmentMapDiffuseOnlyIllumination) == 312, "PerViewLightingData offsets");
VQHIP_STATIC_ASSERT(sizeof(VQ_MaterialData) == 80, "MaterialData");
VQHIP_STATIC_ASSERT(offsetof(VQ_MaterialData, uvScaleOffset) == 48 && offsetof(VQ_MaterialData, roughness) == 64 &&
                    offsetof(VQ_MaterialData, textureConfig) == 76, "MaterialData offsets");
VQHIP_STATIC_ASSERT(sizeof(VQ_TonemapperParams) == 16, "TonemapperParams");
VQHIP_STATIC_ASSERT(sizeof(VQ_BlurParams) == 8, "BlurParams");

/* ------------------------------------------------------------------------------------------------
 * resource descriptors (replace the SRV tables bound at SceneRendering.cpp:1687-1717)
 * ---------------------------------------------------------------------------------------------- */

/* Image-based-lighting inputs of ForwardLighting.hlsl:93-95 (t10 texEnvMapDiff, t11 texEnvMapSpec,
 * t12 texBRDFIntegral). Layouts:
 *   diffuse_cube  : [6][diffuse_res][diffuse_res] RGBA16F (Tex_IrradianceDiffBlurred, 1 mip; EnvironmentMapRendering.cpp:32-43)
 *   specular_cube : mip-major, [mip][6][res>>mip][res>>mip] RGBA16F, densely packed, spec_mips levels
 *                   (Tex_IrradianceSpec; MIPS = CalculateMipLevelCount(res,res) - 1, EnvironmentMapRendering.cpp:55-63)
 *   brdf_lut      : [lut_size][lut_size] RG16F, u = NdotV along x, v = roughness along y (Renderer.cpp:1026-1032)
 * Face order +X,-X,+Y,-Y,+Z,-Z (CubemapUtility.h:26-36). */
typedef struct vqhip_envmap {
    const void* diffuse_cube;   int32_t diffuse_res;
    const void* specular_cube;  int32_t spec_res0;  int32_t spec_mips;
    const void* brdf_lut;       int32_t lut_size;
} vqhip_envmap;

/* Shadow maps of ForwardLighting.hlsl:97-99 as linear R32F arrays (depth targets are D32; point
 * casters store distance/far, Lighting.hlsl:161-163):
 *   directional : [dir_dim][dir_dim]                 (2048 in the reference, SceneRendering.cpp:441)
 *   spot        : [VQ_NUM_SHADOWING_LIGHTS__SPOT][spot_dim][spot_dim]      (1024, :440)
 *   point       : [VQ_NUM_SHADOWING_LIGHTS__POINT][6][point_dim][point_dim] (1024, :439)
 * Any pointer may be NULL when the matching caster count is 0 / directional.shadowing == 0. */
typedef struct vqhip_shadowmaps {
    const float* directional;  int32_t dir_dim;
    const float* spot;         int32_t spot_dim;
    const float* point;        int32_t point_dim;
} vqhip_shadowmaps;

/* Linear float4 G-buffer = the state of ForwardLighting.hlsl:PSMain at :284-293.
 * Four SoA planes of float4, row-major:
 *   gb0 = (P.xyz world position, ao)        ao = fAmbientLightingFactor*localAO*ssao   (:247,269,281)
 *   gb1 = (Surface.N.xyz raw (not renormalised), Surface.roughness)                     (:267,272-277)
 *   gb2 = (Surface.diffuseColor.rgb linear, Surface.metalness)                          (:249,253)
 *   gb3 = (Surface.emissiveColor.rgb, Surface.emissiveIntensity)                        (:250-251) */
typedef struct vqhip_gbuffer {
    const void* gb0; const void* gb1; const void* gb2; const void* gb3;
    int32_t width, height, row_pitch_px;
} vqhip_gbuffer;

/* Output of the load-time environment-map prefilter (FEnvironmentMapRenderingResources,
 * EnvironmentMapRendering.h:30-64): caller-allocated device buffers, same layouts as vqhip_envmap. */
typedef struct vqhip_envmap_out {
    void* diffuse_unblurred;    /* Tex_IrradianceDiff        [6][dres][dres] RGBA16F (optional, may be NULL) */
    void* diffuse_blurred;      /* Tex_IrradianceDiffBlurred [6][dres][dres] RGBA16F                          */
    void* blur_tmp;             /* Tex_BlurTemp              [dres][dres]    RGBA16F (scratch)                */
    void* specular;             /* Tex_IrradianceSpec        mip-major RGBA16F                                */
} vqhip_envmap_out;

/* ---- the G-buffer producer's inputs -------------------------------------------- */

/* One material texture = the reference's DXGI_FORMAT_R8G8B8A8_UNORM Texture2D with its full mip chain
 * (TextureManager.cpp:590; mips by VQ_DXGI_UTILS::MipImage's 4-byte branch, DXGIUtils.cpp:264-285).
 * texels : device pointer, level 0 first, levels densely packed, `mips` levels, level l is
 *          max(1,width>>l) x max(1,height>>l). NULL == the null SRV bound for a missing map
 *          (Renderer_Resources.cpp:385-387): every sample returns 0. */
typedef struct vqhip_texture2d {
    const void* texels;
    int32_t width, height, mips, reserved;   /* reserved: 0, except vqhip_material.texDiffuse.reserved = material flags (below) */
} vqhip_texture2d;
/* vqhip_material.texDiffuse.reserved bit 0: the material is drawn with the "_AlphaMasked" PSO permutation, i.e. PSMain compiled with
 * ENABLE_ALPHA_MASK (PipelineStateObjects.cpp:1477,1571) — chosen by Material::IsAlphaMasked (Material.cpp:39: an alpha-mask map is
 * bound, or the diffuse map uses its alpha channel). Then `if (HasDiffuseMap(TEX_CFG) && AlbedoAlpha.a < 0.01f) discard;`
 * (ForwardLighting.hlsl:237-240): vqhip_gbuffer_from_materials gives such a pixel an all-zero record AND rewrites its material index in
 * ip2.w to -1 (in place), so that it reads as "no geometry" to vqhip_skydome / vqhip_unlit_composite, like a discarded fragment. */
#define VQHIP_MATERIAL_ALPHA_MASKED 1

/* cbPerObject.materialData + the descriptor table t0..t7 of ForwardLighting.hlsl:84-92 that
 * AssetLoader.cpp:406-420 fills per material (t3 texAlphaMask and t8 texHeightmap are not read by
 * the non-tessellated, non-alpha-masked PSMain and are omitted). */
typedef struct vqhip_material {
    VQ_MaterialData data;
    vqhip_texture2d texDiffuse;         /* t0 */
    vqhip_texture2d texNormals;         /* t1 */
    vqhip_texture2d texEmissive;        /* t2 */
    vqhip_texture2d texMetalness;       /* t4 */
    vqhip_texture2d texRoughness;       /* t5 */
    vqhip_texture2d texOcclRoughMetal;  /* t6 */
    vqhip_texture2d texLocalAO;         /* t7 */
} vqhip_material;

VQHIP_STATIC_ASSERT(sizeof(vqhip_texture2d) == 24, "vqhip_texture2d");
VQHIP_STATIC_ASSERT(sizeof(vqhip_material) == 256 && offsetof(vqhip_material, texDiffuse) == 80 &&
                    offsetof(vqhip_material, texLocalAO) == 224, "vqhip_material");

/* The rasteriser's output that PSMain consumes (struct PSInput, ForwardLighting.hlsl:42-53), one record
 * per pixel in three SoA float4 planes (device pointers, row_pitch_px pixels per row):
 *   ip0 = (WorldSpacePosition.xyz, uv.x)
 *   ip1 = (WorldSpaceNormal.xyz,   uv.y)
 *   ip2 = (WorldSpaceTangent.xyz,  asfloat(int32 materialIndex))   materialIndex < 0: no geometry
 * SV_POSITION.xy is implied: (x + 0.5, y + 0.5). Producing these planes (VS + rasteriser, or a
 * visibility-buffer resolve) stays with the caller. */
typedef struct vqhip_interpolants {
    const void* ip0; const void* ip1; const void* ip2;
    int32_t width, height, row_pitch_px;
} vqhip_interpolants;

/* texScreenSpaceAO (t9): R8_UNORM [height][width] (RenderResources.cpp:358-371), or NULL when SSAO is
 * off (the reference then clears the target to 1.0, SceneRendering.cpp:1543-1553). */
typedef struct vqhip_ssao {
    const void* texels; int32_t width, height;
} vqhip_ssao;

typedef struct vqhip_ctx vqhip_ctx;

/* summation order of the convolution integrals (DESIGN.md "Convolution order"):
 *   SEQUENTIAL : every texel's taps are accumulated in the HLSL loop order (CubemapConvolution.hlsl:132-159,183-219): the
 *                reference's result. Default. The taps are evaluated wave-parallel and parked in LDS; one lane per (texel,
 *                channel) adds them in order (conv.hip:k_conv_diffuse_ordered / k_conv_specular_ordered).
 *   WAVE64     : one 64-lane wave per texel; lane l accumulates taps l, l+64, ... in order, then a fixed
 *                xor-butterfly (32,16,8,4,2,1) combines the 64 partial sums: a better-conditioned sum that is up to 2 RGBA16F
 *                ulps from the reference's on BASELINE config 4, 1.26 x faster for the diffuse integral. The oracle implements both. */
typedef enum vqhip_conv_order { VQHIP_CONV_SEQUENTIAL = 0, VQHIP_CONV_WAVE64 = 1 } vqhip_conv_order;

/* ------------------------------------------------------------------------------------------------
 * entry points
 * ---------------------------------------------------------------------------------------------- */

/* Replaces VQRenderer::Initialize device/queue creation (Source/Renderer/Renderer.cpp:204) for this path:
 * binds the context to HIP device `device_ordinal`. Fails with VQHIP_ERR_NO_DEVICE when no GPU is visible. */
VQHIP_API int  vqhip_create(int device_ordinal, vqhip_ctx** out_ctx);
VQHIP_API void vqhip_destroy(vqhip_ctx* ctx);
VQHIP_API const char* vqhip_last_error(const vqhip_ctx* ctx);   /* ctx may be NULL: process-wide last error */
VQHIP_API int  vqhip_abi_version(void);                          /* == VQHIP_ABI_VERSION */

/* How pow(1 - cos, 5.0) of the three Fresnel terms (BRDF.hlsl:135, :155, :274) is evaluated by vqhip_forward_lighting and
 * vqhip_brdf_lut issued through this context (contract v4):
 *   VQHIP_FRESNEL_POW_PRODUCT   (default) x*((x*x)*(x*x)) — FXC's / DXC -Gec's mul-only pattern: 21 % fewer instructions in the
 *                               light loop, finite where dot(H,V) rounds a hair above 1;
 *   VQHIP_FRESNEL_POW_EXP2_LOG2 exp2(5*log2(x)) — what DXC emits with the engine's own flags (no -Gec): NaN for a negative base,
 *                               values within ~3 binary32 ulps of the product elsewhere (the contract of rounds v1-v3). */
typedef enum vqhip_fresnel_pow { VQHIP_FRESNEL_POW_PRODUCT = 0, VQHIP_FRESNEL_POW_EXP2_LOG2 = 1 } vqhip_fresnel_pow;
VQHIP_API int  vqhip_set_fresnel_pow(vqhip_ctx* ctx, vqhip_fresnel_pow mode);
/* The READING of the HLSL intrinsics whose lowering the source does not fix — dot, normalize, length, reflect — in every later
 * vqhip_forward_lighting / vqhip_gbuffer_from_materials / vqhip_forward_lighting_from_materials / vqhip_ssr_environment_fallback through this context
 * (the reference's own HLSL is built in both readings: oracle/ref_src/hlsl_shim.h):
 *   VQHIP_ARITH_LITERAL (default) products and sums rounded one by one, left to right; normalize(v) = v / length(v), one IEEE quotient per component;
 *   VQHIP_ARITH_DXC     what DXC's HLOperationLower emits (Source/Renderer/Pipeline/ShaderCompileUtils.cpp:53-56 selects no flag that changes it): DXIL Dot3 as
 *                       the FMA chain fma(az,bz, fma(ay,by, ax*bx)), normalize(v) = v * Rsqrt(dot(v,v)) with a correctly rounded rsqrt, length = sqrt of that dot.
 * Together with VQHIP_FRESNEL_POW_EXP2_LOG2 this is the second build of the reference's sources (tests/golden/ref_outputs_dxc.npz): each mode is within one
 * RGBA16F ulp of ITS reading; the two readings lie up to 15 ulps apart on highlight pixels. A maintainer who calibrates against D3D12 WARP
 * (docs/WARP_CALIBRATION.md) selects whichever WARP turns out to follow. */
/* Tuning / A-B options of a context, read by the launchers at call time (never from the process environment: getenv racing a host setenv is
 * undefined behaviour, and VQEngine records on many threads, SceneRendering.cpp:563-706). value NULL, "" or "default" restores the default. Every
 * form selected here gives the bits of the default form (tests/test_gpu_conv_forms.py, tests/test_gpu_round3.py); unknown keys / values: VQHIP_ERR_INVALID_ARG.
 *   shade_wg 64|128|256 · psmain_waves 4|5|6 · blur_y_wgs n · post_form two|chain · post_strips n · lut_form general ·
 *   specular_form general · diffuse_form records|texels|general · diffuse_seq_form ordered|lane
 * (the non-default values are the general / fallback forms of the same kernels; the measured-and-rejected kernel forms of rounds 2-5 — the compact tonemap
 * tables, the per-sample LUT and per-mip specular kernels, the persistent X pass, the rolling-ring Y pass — are not in the library: docs/HISTORY.md).
 * psmain_waves applies to the literal reading only (the DXC-reading instantiation of the fused PSMain kernel has one register cap). post_form: "two" = vqhip_post_process[_tile] always runs blur X, then blur Y + tonemap (two kernels); "chain" = the one-kernel chain whatever the frame size
 * (default: the chain for RGBA16F -> RGBA8 frames of >= 2^20 pixels and a per-channel display curve, the two kernels otherwise). */
VQHIP_API int  vqhip_set_option(vqhip_ctx* ctx, const char* key, const char* value);
typedef enum vqhip_arithmetic { VQHIP_ARITH_LITERAL = 0, VQHIP_ARITH_DXC = 1 } vqhip_arithmetic;
VQHIP_API int  vqhip_set_arithmetic(vqhip_ctx* ctx, vqhip_arithmetic mode);
#define VQHIP_ABI_VERSION 3   /* 3 (round 5): + vqhip_post_process_tile; options blur_x_wgs / blur_y_form / blur_y_rows removed, post_form / post_strips added.
                               * 2 (round 4): + vqhip_set_arithmetic, vqhip_set_option, vqhip_ssr_environment_fallback, VQHIP_FMT_R10G10B10A2_UNORM; conv order default SEQUENTIAL;
                               * later in round 4, additions only: vqhip_forward_lighting_mrt, vqhip_forward_lighting_from_materials_mrt, vqhip_scene_normals_from_materials,
                               * vqhip_composite_reflections; vqhip_visualize reads R10G10B10A2 / RG16F / RG32F inputs */

/* Replaces VQRenderer::RenderSceneColor's lit draw loop (SceneRendering.cpp:1619-1785, hot part :1730-1784)
 * == ForwardLighting.hlsl:PSMain :289-380 evaluated for every pixel of the G-buffer.
 *   perFrame / perView : the cbuffers b0 / b1 of ForwardLighting.hlsl:76-77.
 *   extraPoint[numExtraPoint] : extension — point lights beyond the 100-slot cbuffer array (BASELINE cfg5);
 *                               accumulated right after point_lights[0..numPointLights) in index order.
 *   env  : NULL => NullCubemap / NullTex2D path (SceneRendering.cpp:1698-1709): IBL contributes 0.
 *   sm   : NULL allowed iff numPointCasters == numSpotCasters == 0 and !directional.shadowing.
 *   out  : float4(I_total, roughness) per pixel (ForwardLighting.hlsl:380), outFmt RGBA16F (reference RT0,
 *          PipelineStateObjects.cpp:1454) or RGBA32F; out_row_pitch_px pixels per row. */
VQHIP_API int vqhip_forward_lighting(vqhip_ctx* ctx, void* stream,
        const vqhip_gbuffer* gbuf,
        const VQ_PerFrameData* perFrame, const VQ_PerViewLightingData* perView,
        const VQ_PointLight* extraPoint, int numExtraPoint,
        const vqhip_envmap* env, const vqhip_shadowmaps* sm,
        void* out, int out_row_pitch_px, vqhip_format outFmt);

/* The other render targets of the same draw: the PSO permutations OUTPUT_ALBEDO / OUTPUT_MOTION_VECTORS of ForwardLighting.hlsl
 * (PipelineStateObjects.cpp:1499-1504,1547-1563; PSOutput :57-68; written at :382-389; bound by RenderSceneColor, SceneRendering.cpp:1647-1680):
 *   albedo_metallic : SV_TARGET1 = float4(Surface.diffuseColor, Surface.metalness) (:383) — Tex_SceneVisualization, RGBA16F (RenderResources.cpp:171-175)
 *                     | RGBA32F; NULL = the permutation without OUTPUT_ALBEDO
 *   motion_vectors  : float2(svPositionCurr.xy / svPositionCurr.w - svPositionPrev.xy / svPositionPrev.w) (:387) — Tex_SceneMotionVectors, RG16F
 *                     (RenderResources.cpp:178-182) | RG32F; NULL = the permutation without OUTPUT_MOTION_VECTORS
 *   svPositionCurr / svPositionPrev : two more planes of the rasteriser's output (PSInput :49-52, TEXCOORD1 / TEXCOORD2): the interpolated clip-space
 *                     positions mul(matWorldViewProj, v) and mul(matWorldViewProjPrev, v) of TransformVertex :172-185, float4 per pixel, sv_pitch_px
 *                     pixels per row (0 = width). Required iff motion_vectors is set.
 * Every pixel of the frame is written from the planes as given (a rasteriser leaves uncovered pixels at the clear value: give them equal
 * positions with w = 1, or composite afterwards). Device pointers; pitches in pixels (0 = width). */
typedef struct vqhip_psmain_targets {
    void*       albedo_metallic;  vqhip_format albedo_fmt;  int32_t albedo_pitch_px;
    void*       motion_vectors;   vqhip_format motion_fmt;  int32_t motion_pitch_px;
    const void* svPositionCurr;   const void*  svPositionPrev;  int32_t sv_pitch_px;  int32_t pad_;
} vqhip_psmain_targets;
VQHIP_STATIC_ASSERT(sizeof(vqhip_psmain_targets) == 56, "vqhip_psmain_targets");

/* vqhip_forward_lighting + the extra targets, in the same kernel (the G-buffer record is in registers: + 8 B/pixel written for SV_TARGET1,
 * + 32 B read and 4 B written per pixel for the motion vectors). targets == NULL or both outputs NULL: exactly vqhip_forward_lighting. */
VQHIP_API int vqhip_forward_lighting_mrt(vqhip_ctx* ctx, void* stream,
        const vqhip_gbuffer* gb,
        const VQ_PerFrameData* perFrame, const VQ_PerViewLightingData* perView,
        const VQ_PointLight* extraPoint, int numExtraPoint,
        const vqhip_envmap* env, const vqhip_shadowmaps* sm,
        void* out, int out_row_pitch_px, vqhip_format outFmt, const vqhip_psmain_targets* targets);

/* Replaces the GaussianBlur.hlsl CSMain_X / CSMain_Y dispatches (EnvironmentMapRendering.cpp:279-373,
 * SceneRendering.cpp:2582-2638): 21-tap separable Gaussian, clamp-to-edge, alpha := 1.
 * fmt (RGBA16F = reference, or RGBA32F) applies to in, tmp and out alike; the intermediate is rounded
 * to `fmt` between the passes exactly like the reference's Tex_BlurTemp / BlurIntermediate. */
VQHIP_API int vqhip_gaussian_blur(vqhip_ctx* ctx, void* stream, const void* in, void* tmp, void* out,
        const VQ_BlurParams* params, vqhip_format fmt);
VQHIP_API int vqhip_gaussian_blur_x(vqhip_ctx* ctx, void* stream, const void* in, void* out,
        const VQ_BlurParams* params, vqhip_format fmt);
/* Y pass over a row tile. halo_top = the `halo_rows` rows just above row 0 of `in` (row -halo_rows first),
 * halo_bottom = the rows just below the tile; NULL => that side is the image border (clamp).
 * halo_rows must be 0 (both NULL) or >= 10 (= KERNEL_RANGE-1, GaussianBlur.hlsl:54-55). Used by the
 * row-tiled multi-GPU mode where neighbours exchange X-blurred rows over RCCL. */
VQHIP_API int vqhip_gaussian_blur_y(vqhip_ctx* ctx, void* stream, const void* in, void* out,
        const void* halo_top, const void* halo_bottom, int halo_rows,
        const VQ_BlurParams* params, vqhip_format fmt);

/* Fused form of the last two dispatches of the post chain when the blur is enabled: CSMain_Y (GaussianBlur.hlsl:155-187,
 * SceneRendering.cpp:2613-2638) immediately followed by Tonemapper.hlsl:CSMain (:2640-2656). The blurred value is rounded
 * to `blurFmt` exactly as if it had been stored to BlurOutput and re-read, then tonemapped in registers: identical bits to
 * vqhip_gaussian_blur_y + vqhip_tonemap, one image round trip through HBM less. */
VQHIP_API int vqhip_gaussian_blur_y_tonemap(vqhip_ctx* ctx, void* stream, const void* in, void* out,
        const void* halo_top, const void* halo_bottom, int halo_rows,
        const VQ_BlurParams* blurParams, const VQ_TonemapperParams* tonemapParams, vqhip_format blurFmt, vqhip_format outFmt);

/* Replaces the Tonemapper.hlsl:CSMain dispatch (SceneRendering.cpp:2640-2656).
 * inFmt RGBA16F|RGBA32F; outFmt RGBA8_UNORM (SDR swapchain path) | RGBA16F (HDR path) | RGBA32F.
 * Table cache: the calls that tonemap through a 65 536-entry table (this one, vqhip_gaussian_blur_y_tonemap, vqhip_post_process; RGBA16F input, a curve that
 * does not mix channels) keep the tables of the FOUR most recent (parameters, output format) sets in the context. A fifth set rebuilds the least recently used
 * table on the calling stream, behind an event recorded after that table's last reader — no host wait, legal under stream capture — unless the table has been
 * read from more than one stream: then the call waits for the whole device once (hipDeviceSynchronize: a host stall, and an error under stream capture).
 * An application that animates tonemapper parameters from several streams should serialise those calls on one stream. */
VQHIP_API int vqhip_tonemap(vqhip_ctx* ctx, void* stream, const void* in, void* out, int width, int height,
        const VQ_TonemapperParams* params, vqhip_format inFmt, vqhip_format outFmt);

/* Replaces the blur + tonemapper part of VQRenderer::RenderPostProcess as ONE call (SceneRendering.cpp:2579-2656: "BlurCS" { CSMain_X, CSMain_Y }
 * when bEnableGaussianBlur, then "TonemapperCS"): sceneColor -> out. BlurIntermediate lives in a scratch buffer of the context; for the
 * reference's formats (RGBA16F scene colour, RGBA8 SDR target) and a display curve that does not mix channels (sRGB, LINEAR, ST2084 on Rec.2020
 * content) the Y pass and the tonemapper are one kernel and BlurOutput never exists; for frames of >= 2^20 pixels the X pass joins them (k_post_chain: 8 B read +
 * 4 B written per pixel, BlurIntermediate never exists either). Identical bits to vqhip_gaussian_blur_x -> _y ->
 * vqhip_tonemap (the intermediate roundings to the blur format are reproduced). enableGaussianBlur == 0: the tonemapper alone. */
VQHIP_API int vqhip_post_process(vqhip_ctx* ctx, void* stream, const void* sceneColor, void* out, int width, int height,
        const VQ_TonemapperParams* tonemapParams, int enableGaussianBlur, vqhip_format inFmt, vqhip_format outFmt);
/* The same chain (blur enabled) over ONE ROW TILE of a frame that is split between GPUs. halo_top = the `halo_rows` SCENE-COLOUR rows just above
 * row 0 of `sceneColor` (row -halo_rows first), halo_bottom = the rows just below the tile, in `inFmt`; NULL => that side is the image border (clamp).
 * halo_rows must be 0 (both NULL) or >= 10. The X pass is purely horizontal, so the neighbour's shaded rows are filtered in X here like the tile's own and the
 * Y window reaches them: neighbours exchange 10 rows of scene colour (vqhip_exchange_blur_halos moves rows of any RGBA16F / RGBA32F image) right after the
 * shade kernel instead of 10 X-blurred rows after the X pass — the same bytes, one dependency earlier. Identical bits to the untiled frame. */
VQHIP_API int vqhip_post_process_tile(vqhip_ctx* ctx, void* stream, const void* sceneColor, void* out,
        const void* halo_top, const void* halo_bottom, int halo_rows, int width, int height,
        const VQ_TonemapperParams* tonemapParams, vqhip_format inFmt, vqhip_format outFmt);

/* Replaces VQRenderer::ComputeBRDFIntegrationLUT (Renderer.cpp:871-909) == CubemapConvolution.hlsl:
 * CSMain_BRDFIntegration :225-240. Reference values: size 1024, samples 2048, RG16F. */
VQHIP_API int vqhip_brdf_lut(vqhip_ctx* ctx, void* stream, void* outRG, int size, int samples, vqhip_format fmt);

/* Replaces VQ_DXGI_UTILS::MipImage's 16-byte branch (Source/Renderer/Resources/DXGIUtils.cpp:289-317) as
 * driven by TextureManager::GenerateMips (TextureManager.cpp:643-738): per-channel MIN of each 2x2 block,
 * alpha := 1. `mips` = device buffer holding level 0 (w0 x h0 RGBA32F) followed by space for all further
 * levels, densely packed; levels 1..nMips-1 are written. vqhip_mip_chain_bytes() sizes the buffer and
 * vqhip_mip_level_count() == Image::CalculateMipLevelCount == floor(log2(max(w,h)))+1. */
VQHIP_API int    vqhip_mip_level_count(int w, int h);
VQHIP_API size_t vqhip_mip_chain_bytes(int w0, int h0, int nMips);
VQHIP_API size_t vqhip_mip_level_offset_bytes(int w0, int h0, int level);
VQHIP_API int    vqhip_mip_chain_min_rgba32f(vqhip_ctx* ctx, void* stream, void* mips, int w0, int h0, int nMips);

/* Cube-map helpers: number of mips of the specular cube == CalculateMipLevelCount(res,res) - 1
 * (EnvironmentMapRendering.cpp:63) and byte sizes of the packed RGBA16F cubes. */
VQHIP_API int    vqhip_specular_mip_count(int spec_res0);
VQHIP_API size_t vqhip_cube_bytes(int res0, int nMips, vqhip_format fmt);

/* Replaces the diffuse-irradiance draws (EnvironmentMapRendering.cpp:181-277) ==
 * CubemapConvolution.hlsl:PSMain_DiffuseIrradiance :112-163 over all 6 faces.
 * equirect_mips = RGBA32F mip chain (layout of vqhip_mip_chain_min_rgba32f); the shader samples mip 3
 * with a TRILINEAR_WRAP sampler (RootSignatures.cpp:402). step = INTEGRATION_STEP_DIFFUSE_IRRADIANCE
 * (0.050 / 0.025 / 0.010, PipelineStateObjects.cpp:1298-1306).
 * When the sampled mip is a power-of-two image the call first rewrites it, on `stream`, as 48-byte footprint records in a buffer the context
 * owns (3 x the mip: 3 MB for a 2048^2 chain) and the taps gather from those; the buffer is rewritten by every call, and a call on another
 * stream waits (hipStreamWaitEvent) for the previous call's kernel before it does so. */
VQHIP_API int vqhip_conv_diffuse(vqhip_ctx* ctx, void* stream, const void* equirect_mips, int w0, int h0, int nMips,
        int diffuseRes, float step, vqhip_conv_order order, void* outCube, vqhip_format fmt);

/* Replaces the specular-prefilter draws (EnvironmentMapRendering.cpp:386-472) ==
 * CubemapConvolution.hlsl:PSMain_SpecularIrradiance :168-223 for every (mip, face); roughness =
 * mip/(MIPS-1), TextureDimensionsLOD0 = (w0,h0) of the equirect (:431-435). */
VQHIP_API int vqhip_conv_specular(vqhip_ctx* ctx, void* stream, const void* equirect_mips, int w0, int h0, int nMips,
        int specRes0, vqhip_conv_order order, void* outCubeMips, vqhip_format fmt);

/* Replaces VQRenderer::PreFilterEnvironmentMap (EnvironmentMapRendering.cpp:139-486): diffuse
 * convolution -> per-face blur X,Y -> specular mips, all outputs RGBA16F like the reference. */
VQHIP_API int vqhip_envmap_prefilter(vqhip_ctx* ctx, void* stream, const void* equirect_mips, int w0, int h0, int nMips,
        int diffuseRes, float diffuseStep, int specRes0, vqhip_conv_order order, const vqhip_envmap_out* out);

/* ---- G-buffer producer ----------------------------------------------------------
 * Replaces the surface-assembly half of ForwardLighting.hlsl:PSMain (:226-287): uv transform, the seven
 * material-map fetches, SRGBToLinear (ShadingMath.hlsl:65), Has*Map() selection
 * (LightingConstantBufferData.h:116-124), UnpackNormal (ShadingMath.hlsl:44-52), the ORM / AO / SSAO
 * multiplies — for every pixel of the interpolant planes at once, each pixel using
 * materials[materialIndex] (the reference binds one material per draw: SceneRendering.cpp:1744-1750).
 * Writes the four planes of `out` (their pointers are written through despite the const in
 * vqhip_gbuffer); pixels without geometry get all-zero records, and so do the fragments an alpha-masked material discards
 * (VQHIP_MATERIAL_ALPHA_MASKED: their index in in->ip2 is overwritten with -1).
 *   materials : HOST array (copied; numMaterials <= vqhip_max_materials())
 *   fAmbientLightingFactor : cbPerFrame.fAmbientLightingFactor (:247)
 *   ssao : NULL or a descriptor with texels == NULL => factor 1.0
 * Sampler s2 is ANISOTROPIC_WRAP with MaxAnisotropy = 0 (RootSignatures.cpp:111,149), s0 TRILINEAR_WRAP:
 * both are evaluated as isotropic trilinear WRAP with the LOD of the pixel quad's uv differences
 * (DESIGN.md "G-buffer producer"). */
VQHIP_API int vqhip_max_materials(void);
VQHIP_API int vqhip_gbuffer_from_materials(vqhip_ctx* ctx, void* stream,
        const vqhip_interpolants* in, const vqhip_material* materials, int numMaterials,
        float fAmbientLightingFactor, const vqhip_ssao* ssao, const vqhip_gbuffer* out);

/* PSMain as the engine runs it (Shaders/ForwardLighting.hlsl:226-380; the lit draws of VQRenderer::RenderSceneColor,
 * SceneRendering.cpp:1619-1760): vqhip_gbuffer_from_materials and vqhip_forward_lighting in ONE kernel. The 64-byte G-buffer record of a
 * pixel never leaves registers (128 B/pixel of HBM traffic less than the two calls); the result is bit-identical to the two calls with
 * fAmbientLightingFactor = perFrame->fAmbientLightingFactor (the same cbuffer field PSMain reads, :247). Pixels without geometry and
 * discarded fragments shade the all-zero record exactly like the two calls do — composite the sky over them with vqhip_skydome
 * (alpha-masked discards are marked -1 in in->ip2.w as by vqhip_gbuffer_from_materials). Arguments: as the two calls. */
VQHIP_API int vqhip_forward_lighting_from_materials(vqhip_ctx* ctx, void* stream,
        const vqhip_interpolants* in, const vqhip_material* materials, int numMaterials, const vqhip_ssao* ssao,
        const VQ_PerFrameData* perFrame, const VQ_PerViewLightingData* perView,
        const VQ_PointLight* extraPoint, int numExtraPoint,
        const vqhip_envmap* env, const vqhip_shadowmaps* sm,
        void* out, int out_row_pitch_px, vqhip_format outFmt);
/* the same with the extra render targets of the draw (vqhip_psmain_targets above); bit-identical to vqhip_gbuffer_from_materials + vqhip_forward_lighting_mrt
 * on every pixel a fragment covers. Pixels WITHOUT geometry and alpha-mask discards are not written in albedo_metallic / motion_vectors — they keep the
 * values the caller cleared the targets to, as under the rasteriser (PSMain never runs there; the G-buffer form above cannot know and writes them all). */
VQHIP_API int vqhip_forward_lighting_from_materials_mrt(vqhip_ctx* ctx, void* stream,
        const vqhip_interpolants* in, const vqhip_material* materials, int numMaterials, const vqhip_ssao* ssao,
        const VQ_PerFrameData* perFrame, const VQ_PerViewLightingData* perView,
        const VQ_PointLight* extraPoint, int numExtraPoint,
        const vqhip_envmap* env, const vqhip_shadowmaps* sm,
        void* out, int out_row_pitch_px, vqhip_format outFmt, const vqhip_psmain_targets* targets);

/* Replaces the pixel shader of the Z pre-pass (VQRenderer::RenderDepthPrePass, SceneRendering.cpp:1264-1360; Shaders/DepthPrePass.hlsl:PSMain :153-171) for
 * every pixel of the interpolant planes: Tex_SceneNormals, the packed surface normals that SSR (`g_normal` of vqhip_ssr_environment_fallback) and
 * FFX-CACAO read —
 *     float4((SurfaceN + 1) * 0.5, 1),   SurfaceN = length(Normal) < 0.01 ? normalize(WorldSpaceNormal) : UnpackNormal(Normal, N, T)
 * the surface normal of ForwardLighting.hlsl:265-267, except that the normal map is fetched with Sample — no normalMapMipBias (:164) — and the diffuse map
 * only for the alpha test of the "_AlphaMasked" PSOs (VQHIP_MATERIAL_ALPHA_MASKED, :157-161). Pixels without geometry and discarded fragments keep the
 * target's clear value 0 (SceneRendering.cpp:1289-1300). The coverage plane in->ip2.w is NOT modified (the lighting pass repeats the discard itself).
 *   out    : outFmt R10G10B10A2_UNORM — the reference's format (RenderResources.cpp:185-197, PipelineStateObjects.cpp:1634-1635): one uint32 per pixel,
 *            r in bits 0-9, g 10-19, b 20-29, alpha 30-31 (1 -> 3); float -> UNORM n = trunc(saturate(c) * (2^n - 1) + 0.5) —
 *            or RGBA32F holding the unquantised float4 (w = 1 covered / 0 not). out_row_pitch_px pixels per row (0 = width).
 *   materials : HOST array, as for vqhip_gbuffer_from_materials; vqhip_set_arithmetic selects the reading of normalize / dot here too.
 * The depth half of the pre-pass (the rasteriser's z) stays with the caller, like the interpolant planes. */
VQHIP_API int vqhip_scene_normals_from_materials(vqhip_ctx* ctx, void* stream,
        const vqhip_interpolants* in, const vqhip_material* materials, int numMaterials,
        void* out, vqhip_format outFmt, int out_row_pitch_px);

/* Replaces VQ_DXGI_UTILS::MipImage's 4-byte branch (DXGIUtils.cpp:264-285) as driven by
 * TextureManager::GenerateMips: each channel = (sum of the 2x2 block) / 4, integer division.
 * Same buffer convention as vqhip_mip_chain_min_rgba32f with 4-byte texels; w0, h0 powers of two. */
VQHIP_API size_t vqhip_mip_chain_bytes_rgba8(int w0, int h0, int nMips);
VQHIP_API int    vqhip_mip_chain_box_rgba8(vqhip_ctx* ctx, void* stream, void* mips, int w0, int h0, int nMips);

/* ---- skydome ----------------------------------------------------------------------
 * Replaces the "Draw Environment Map" draw of VQRenderer::RenderSceneColor (SceneRendering.cpp:1822-1850) ==
 * Skydome.hlsl:VSMain/PSMain (:39-56): every pixel no geometry covers gets
 * float4(texEquirectEnvironmentMap.SampleLevel(TRILINEAR_WRAP, DirectionToEquirectUV(normalize(dir)), 0).rgb, 1).
 * The cube mesh is centred on the sky camera (position 0, yaw = camera yaw + HDRIYawOffset, pitch = camera pitch,
 * projection of the main camera: Scene.cpp:573-584) and CubemapLookDirection = normalize(position) is linear on
 * every cube face, so its perspective-correct interpolant at a pixel is parallel to the pixel's view ray:
 *   dir = forward + (ndc.x * tanHalfFovX) * right + (ndc.y * tanHalfFovY) * up,
 *   ndc = (2(x+.5)/W - 1, 1 - 2(y+.5)/H)          (VQ_SkydomeParams holds the sky camera's world-space basis).
 *   equirect_level0 : RGBA32F w0 x h0 (level 0 of the HDRI chain; SRV_HDREnvironment, EnvironmentMap.cpp:142-209)
 *   coverage        : NULL => every pixel is sky; else only pixels whose ip2.w material index is < 0 are written
 *   color           : scene colour target (RGBA16F reference format, or RGBA32F), written in place. */
typedef struct VQ_SkydomeParams {
    VQ_float3 right;   float tanHalfFovX;
    VQ_float3 up;      float tanHalfFovY;
    VQ_float3 forward; float pad;
} VQ_SkydomeParams;
VQHIP_API int vqhip_skydome(vqhip_ctx* ctx, void* stream, const void* equirect_level0, int w0, int h0,
        const VQ_SkydomeParams* params, const vqhip_interpolants* coverage,
        void* color, int width, int height, int row_pitch_px, vqhip_format fmt);

/* Light gizmo meshes — the other half of SURVEY.md §8(f).2 — are drawn by the reference with Unlit.hlsl (SceneRendering.cpp:1787-1819):
 * VSMain transforms the mesh, PSMain (:58-61) returns the light's colour for every covered, depth-tested pixel. Rasterisation stays
 * with the engine; what crosses the boundary is its result, in the same plane the skydome reads: a pixel whose ip2.w index is
 * -(2+k) is covered by gizmo k (index -1 = sky, >= 0 = material). Those pixels receive colors[k] (all four channels, as PSMain
 * returns float4(color)); every other pixel is left untouched. colors is a HOST array of numColors <= VQHIP_MAX_UNLIT_COLORS entries
 * (FFrameConstantBufferUnlit::color of each FLightRenderData, SceneRendering.cpp:1799). */
#define VQHIP_MAX_UNLIT_COLORS 64
VQHIP_API int vqhip_unlit_composite(vqhip_ctx* ctx, void* stream, const vqhip_interpolants* coverage,
        const VQ_float4* colors, int numColors, void* color, int width, int height, int row_pitch_px, vqhip_format fmt);

/* ---- HDRI ingest -------------------------------------------------------------------
 * Replaces Image::LoadFromFile -> stbi_loadf for Radiance .hdr files (call site TextureManager.cpp:566; the decoder
 * itself lives in stb_image inside the un-vendored Libs/VQUtils submodule — its published algorithm, stb_image.h
 * v2.2x stbi__hdr_load / stbi__hdr_convert, is what is restated): header "#?RADIANCE" | "#?RGBE", a
 * "FORMAT=32-bit_rle_rgbe" line, blank line, "-Y <height> +X <width>", then per scanline either new-style RLE
 * (02 02 hi lo + four run-length coded byte planes) or flat RGBE quadruples (width < 8, width >= 32768, or no
 * 02 02 marker); pixel = (r,g,b) * 2^(e-136), (0,0,0) when e == 0, alpha := 1.
 * The host parses the header and walks the run headers (count bytes only: where every byte plane of every scanline starts, and whether the file is sound); the
 * encoded bytes go to the GPU as they are, and one kernel expands the runs (a workgroup per scanline, a wave per byte plane, through LDS) and converts RGBE -> RGBA32F
 * straight into level 0 of the chain that vqhip_mip_chain_min_rgba32f completes. Flat files and scanlines wider than ~19 000 pixels: expansion on the host.
 *   file / bytes : HOST memory holding the whole .hdr file
 * Truncated or corrupt run data returns VQHIP_ERR_INVALID_ARG (stb_image reads zeros past the end of the file). */
VQHIP_API int vqhip_hdr_parse_header(const void* file, size_t bytes, int* width, int* height, size_t* data_offset);
VQHIP_API int vqhip_hdr_decode_rgba32f(vqhip_ctx* ctx, void* stream, const void* file, size_t bytes,
        void* out_rgba32f, int width, int height);
/* Replaces Image::CreateResizedImage as CreateEnvironmentMapTextureFromHiResAndSaveToDisk uses it (EnvironmentMap.cpp:142-209, call :167):
 * when the HDRI of the chosen resolution is missing, the next higher one (8k / 4k / 2k of LookupResolutionX/Y :163-164) is downsized to it.
 * The resampler itself (stb_image_resize inside the un-vendored Libs/VQUtils submodule, version and filter not pinned by the reference tree)
 * is NOT restated — PARITY UNPINNED for this entry point: every target texel is the plain mean of its k x k source block, k = width /
 * out_width = height / out_height an integer (2, 4, 8 for the engine's table) — which is also what the engine's alternative branch (:183-200,
 * repeated CreateHalfResolutionFromImage) converges to. Any other ratio returns VQHIP_ERR_UNSUPPORTED. RGBA32F device images, alpha := 1. */
VQHIP_API int vqhip_hdr_downsize_rgba32f(vqhip_ctx* ctx, void* stream, const void* in_rgba32f, int width, int height,
        void* out_rgba32f, int out_width, int out_height);

/* ---- FidelityFX Super Resolution 1.0 (post chain tail) ------------------------------
 * Replace the FSR-EASU and FSR-RCAS dispatches of VQRenderer::RenderPostProcess (SceneRendering.cpp:2695-2784):
 * Shaders/AMDFidelityFX.hlsl:FSR_EASU_CSMain / FSR_RCAS_CSMain, compiled without FSR_FP16
 * (PipelineStateObjects.cpp:1366-1374) == FsrEasuF / FsrRcasF of Shaders/AMDFidelityFX/FSR1.0/ffx_fsr1.h.
 * The constant blocks are exactly the reference's cbuffers: FFSR1_EASU::EASUConstantBlock[16] and
 * FFSR1_RCAS::RCASConstantBlock[4] (PostProcess.h:114-130), filled by vqhip_fsr_easu_con / vqhip_fsr_rcas_con ==
 * FsrEasuCon / FsrRcasCon on the CPU (PostProcess.cpp:39-79; default RCASSharpnessStops 0.2).
 *   EASU : in (inW x inH, the tonemapper output: RGBA8_UNORM SDR or RGBA16F HDR; RGBA32F also accepted) -> out (outW x outH)
 *   RCAS : in/out of the same size. Alpha is undefined in the reference (RWTexture2D<float3>): 1 is written. */
VQHIP_API void vqhip_fsr_easu_con(uint32_t con[16], float inputViewportW, float inputViewportH,
        float inputSizeW, float inputSizeH, float outputW, float outputH);
VQHIP_API void vqhip_fsr_rcas_con(uint32_t con[4], float sharpnessStops);
VQHIP_API int vqhip_fsr_easu(vqhip_ctx* ctx, void* stream, const void* in, int inW, int inH, vqhip_format inFmt,
        const uint32_t con[16], void* out, int outW, int outH, vqhip_format outFmt);
VQHIP_API int vqhip_fsr_rcas(vqhip_ctx* ctx, void* stream, const void* in, void* out, int width, int height,
        const uint32_t con[4], vqhip_format inFmt, vqhip_format outFmt);

/* Replaces the debug-visualisation dispatch of RenderPostProcess (SceneRendering.cpp:2541-2576) == Visualization.hlsl:CSMain
 * :34-120. VQ_VizParams == FPostProcessParameters::FVizualizationParams (the cbuffer :26-31); iDrawMode uses the SHADER's
 * numbering (:71-80): 1 DEPTH pow(r,500), 2 NORMALS, 3 ROUGHNESS / 4 METALLIC (alpha), 5 AO (red), 6 ALBEDO / 7 REFLECTIONS
 * (rgb), 8 MOTION_VECTORS; anything else (including 0) writes magenta. Output alpha = input alpha. The caller binds the
 * image the reference's switch selects (:2555-2565) in the format that target has: inFmt RGBA8_UNORM | RGBA16F | RGBA32F (scene colour, Tex_SceneVisualization of
 * vqhip_forward_lighting_mrt, reflections), R10G10B10A2_UNORM (Tex_SceneNormals of vqhip_scene_normals_from_materials: c / 1023, alpha / 3), RG16F | RG32F
 * (Tex_SceneMotionVectors: the missing channels read 0 and 1 like a typed SRV load); single-channel sources are passed expanded to (r,0,0,1).
 * outFmt RGBA8_UNORM | RGBA16F | RGBA32F. */
/* Replaces ApplyReflectionsPass::RecordCommands (ApplyReflections.cpp:45-80) == ApplyReflections.hlsl:CSMain :30-50 without
 * COMPOSITE_BOUNDING_VOLUMES: sceneColor.rgb += reflectionRadiance.rgb, alpha (roughness) kept; in place on the scene colour.
 * (The producer of the reflection radiance, FidelityFX SSSR + denoiser, is out of scope.) fmt: RGBA16F | RGBA32F for both. */
VQHIP_API int vqhip_apply_reflections(vqhip_ctx* ctx, void* stream, const void* reflectionRadiance, void* sceneColor,
        int width, int height, vqhip_format fmt);
/* Replaces VQRenderer::CompositeReflections (SceneRendering.cpp:2362-2403): ApplyReflectionsPass in the permutation its parameters select
 * (ApplyReflections.cpp:62-68). boundingVolumes == NULL: vqhip_apply_reflections. Otherwise "[PSO] ApplyReflectionsAndBoundingVolumes"
 * (COMPOSITE_BOUNDING_VOLUMES, ApplyReflections.hlsl:44-48): Tex_SceneColorBoundingVolumes (the light-bounds target of RenderLightBounds,
 * same format as the scene colour) is blended over the sum,
 *     rgb = BV.rgb * BV.a + (scene.rgb + reflection.rgb) * (1 - BV.a),   alpha = BV.a,
 * every product and sum rounded on its own, as written. In place on sceneColor; the inputs must not alias it. fmt: RGBA16F | RGBA32F for all three. */
VQHIP_API int vqhip_composite_reflections(vqhip_ctx* ctx, void* stream, const void* reflectionRadiance, const void* boundingVolumes, void* sceneColor,
        int width, int height, vqhip_format fmt);

/* ---- SSR's consumption of the specular cube + BRDF LUT ---------------------------
 * == FFX_SSSRConstants (Source/Renderer/Rendering/RenderPass/ScreenSpaceReflections.h:43-65), the cbuffer `Constants` of
 * Shaders/ScreenSpaceReflections/Common.hlsl:26-48, byte for byte (XMMATRIX: row-major, 16-aligned), as VQRenderer::RenderReflections fills it
 * (SceneRendering.cpp:2221-2242). */
typedef struct VQ_SSSRConstants {
    VQ_matrix invViewProjection, projection, invProjection, view, invView, prevViewProjection, envMapRotation;
    uint32_t bufferDimensions[2];
    float    inverseBufferDimensions[2];
    float    temporalStabilityFactor, depthBufferThickness, roughnessThreshold, varianceThreshold;
    uint32_t frameIndex, maxTraversalIntersections, minTraversalOccupancy, mostDetailedMip, samplesPerQuad,
             temporalVarianceGuidedTracingEnabled, envMapSpecularIrradianceCubemapMipLevelCount;
    uint32_t pad_[1];
} VQ_SSSRConstants;
VQHIP_STATIC_ASSERT(sizeof(VQ_SSSRConstants) == 512, "FFX_SSSRConstants is 7 matrices + 15 dwords, padded to 16");
VQHIP_STATIC_ASSERT(offsetof(VQ_SSSRConstants, bufferDimensions) == 448 && offsetof(VQ_SSSRConstants, roughnessThreshold) == 472 &&
                    offsetof(VQ_SSSRConstants, envMapSpecularIrradianceCubemapMipLevelCount) == 504, "FFX_SSSRConstants layout");

/* Replaces the part of the "FFX DNSR ClassifyTiles" dispatch (ScreenSpaceReflections.cpp:262-276) that consumes the prefiltered specular cube and
 * the BRDF LUT: for every pixel, ClassifyReflectionTiles.hlsl:ClassifyTiles :146-152 + SampleEnvironmentMap :78-94 —
 *     radiance = (depth < 1 && !(roughness < roughnessThreshold)) ? float4(SampleEnvironmentMap(pixel, roughness, mipCount), 0) : 0
 * i.e. surfaces too rough for a traced ray get the environment's prefiltered reflection: view-space reflect of the view ray about the G-buffer normal,
 * g_environment_map.SampleLevel(envMapRotation * R_world, roughness * (mipCount - 1)) — a FRACTIONAL level: trilinear between cube mips, seamless
 * (csrc/vq_sampling.h:sample_cube_lod_rgba16f) — times EnvironmentBRDF(NdotV, roughness, metallic 1, ...) from the LUT (BRDF.hlsl:196-207).
 * The ray classification / ray list / denoiser tile list of the same dispatch (wave intrinsics, atomics) and the traced rays are FidelityFX SSSR: out of scope.
 *   sceneColorRoughness : g_roughness   == the scene colour whose alpha is the roughness (ForwardLighting.hlsl:380), RGBA16F | RGBA32F
 *   depth               : g_depth_buffer == mip 0 of the depth hierarchy, R32F, NDC z (far plane 1)
 *   normals             : g_normal      == Tex_SceneNormals, R10G10B10A2_UNORM (one uint32 per pixel, r in bits 0-9) | RGBA32F holding the same [0,1] values
 *   cb                  : host pointer; invProjection, view, invView, envMapRotation, inverseBufferDimensions, roughnessThreshold and
 *                         envMapSpecularIrradianceCubemapMipLevelCount are read (mipCount must be in [1, env->spec_mips])
 *   env                 : specular_cube / spec_res0 / spec_mips and brdf_lut / lut_size are read (device pointers)
 *   outRadiance         : g_intersection_output, RGBA16F (TexRadiance, ScreenSpaceReflections.cpp:129) | RGBA32F; alpha = 0
 *   outExtractedRoughness: g_extracted_roughness (CSMain :196), R8_UNORM, one byte per pixel, row pitch = width; may be NULL
 * Pitches in pixels (0 = width). */
VQHIP_API int vqhip_ssr_environment_fallback(vqhip_ctx* ctx, void* stream,
        const void* sceneColorRoughness, vqhip_format sceneFmt, int scenePitchPx,
        const float* depth, int depthPitchPx,
        const void* normals, vqhip_format normalFmt, int normalPitchPx,
        int width, int height, const VQ_SSSRConstants* cb, const vqhip_envmap* env,
        void* outRadiance, vqhip_format outFmt, int outPitchPx, uint8_t* outExtractedRoughness);

typedef struct VQ_VizParams { int32_t iDrawMode; int32_t iUnpackNormals; float fInputStrength; } VQ_VizParams;
VQHIP_API int vqhip_visualize(vqhip_ctx* ctx, void* stream, const void* in, void* out, int width, int height,
        const VQ_VizParams* params, vqhip_format inFmt, vqhip_format outFmt);

/* ---- one frame row-tiled over the GPUs of a node ------------------------------------------------------
 * No reference analogue (VQEngine renders on one adapter; the limit being exceeded is the single command queue of
 * SceneRendering.cpp:2507). One process per GPU; rank r owns rows [row0, row0+rows) of the frame and only that part of the
 * G-buffer, lights / env maps / LUT are replicated. vqhip_forward_lighting, vqhip_gaussian_blur_x and the tonemapper need
 * no communication; the two exchanges below are the whole multi-GPU data path:
 *
 *   vqhip_forward_lighting(tile) -> vqhip_exchange_blur_halos(scene colour) -> vqhip_post_process_tile(tile, halo_top, halo_bottom) -> vqhip_composite_tiles
 *   or, pass by pass:
 *   vqhip_gaussian_blur_x(tile) -> vqhip_exchange_blur_halos(X-blurred) -> vqhip_gaussian_blur_y[_tonemap](tile, halo_top, halo_bottom) -> vqhip_composite_tiles
 *
 * A vqhip_comm wraps an RCCL communicator (one rank per GPU, xGMI point-to-point). RCCL is loaded at run time
 * (librccl.so.1, or the library named by $VQHIP_RCCL_LIBRARY): single-GPU hosts never touch it. Like every other entry point
 * the two exchanges enqueue on `stream` and return without synchronising.
 *   vqhip_comm_unique_id : rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the other ranks out of band
 *   vqhip_comm_create    : every rank, collectively (ncclCommInitRank on the calling thread's current HIP device)
 *   vqhip_comm_adopt     : wrap an ncclComm_t the host already owns (not destroyed by vqhip_comm_destroy) */
#define VQHIP_HALO_ROWS      10        /* KERNEL_RANGE - 1, Shaders/GaussianBlur.hlsl:54-55 */
#define VQHIP_COMM_ID_BYTES  128       /* sizeof(ncclUniqueId) */
#define VQHIP_ALL_RANKS      (-1)
typedef struct vqhip_comm vqhip_comm;
/* rows of rank `rank`: frame_height / world each, the first frame_height % world ranks one more. world > 1 needs >= 10 rows per tile. */
VQHIP_API int  vqhip_rowtile(int frame_height, int world, int rank, int* row0, int* rows);
VQHIP_API int  vqhip_comm_unique_id(void* id128);
VQHIP_API int  vqhip_comm_create(const void* id128, int world, int rank, vqhip_comm** out_comm);
VQHIP_API int  vqhip_comm_adopt(void* nccl_comm, int world, int rank, vqhip_comm** out_comm);
VQHIP_API void vqhip_comm_destroy(vqhip_comm* comm);
/* ncclCommAbort: ends the transfers in flight on the communicator (a watchdog's way out of an exchange that does not complete) and frees
 * it like vqhip_comm_destroy; collective in effect — every rank must abort and build a new communicator before it exchanges again. */
VQHIP_API int  vqhip_comm_abort(vqhip_comm* comm);
/* What the communicator itself reports (read back from RCCL, not from the arguments of vqhip_comm_create): size and rank as RCCL sees
 * them (ncclCommCount / ncclCommUserRank), the library's version (ncclGetVersion; 0 = the test stand-in) and the path it was loaded from. */
typedef struct vqhip_comm_info {
    int32_t world, rank;                 /* as given to vqhip_comm_create / vqhip_comm_adopt */
    int32_t nranks_seen, rank_seen;      /* as RCCL reports them; -1 when the library lacks the query */
    int32_t rccl_version;                /* ncclGetVersion: e.g. 22105; -1 when unavailable */
    int32_t reserved;
    char    library_path[232];           /* dladdr of ncclSend */
} vqhip_comm_info;
VQHIP_API int  vqhip_comm_query(const vqhip_comm* comm, vqhip_comm_info* out_info);
/* Diagnostic: ONE grouped ncclSend + ncclRecv of `bytes` bytes from this rank to ITSELF, enqueued on `stream` like the exchanges below (RCCL serves
 * a self-addressed pair as a device copy). A single-GPU host can so check that the point-to-point entry points of the RCCL it bound
 * (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd, stream semantics, byte counts) work from inside libvqhip.so before a multi-GPU frame
 * depends on them. src and dst: device buffers of `bytes` bytes that do not overlap. */
VQHIP_API int  vqhip_comm_loopback(vqhip_comm* comm, void* stream, const void* src, void* dst, size_t bytes);
/* Exchange 1. xblur_tile: this rank's X-blurred tile (tile_rows x width, row_pitch_px pixels per row, fmt RGBA16F | RGBA32F).
 * Sends its first 10 rows to rank-1 and its last 10 rows to rank+1 and receives theirs into halo_top / halo_bottom (dense
 * 10 x width buffers, the layout vqhip_gaussian_blur_y takes; ignored — may be NULL — at the frame's top / bottom edge).
 * ONE message per neighbour and direction whatever the pitch: rows of a pitched tile are first packed into a dense staging block
 * owned by the communicator (hipMemcpy2DAsync on `stream`), so both sides always post one send and one receive of 10*width pixels. */
VQHIP_API int  vqhip_exchange_blur_halos(vqhip_comm* comm, void* stream, const void* xblur_tile, int width, int tile_rows,
        int row_pitch_px, vqhip_format fmt, void* halo_top, void* halo_bottom);
/* Exchange 2. tile: this rank's finished tile (dense rows, fmt e.g. RGBA8_UNORM). root >= 0: only that rank receives the
 * frame (frame_height x width, dense; NULL elsewhere) — the GPU that presents, like the reference's one swap chain;
 * root == VQHIP_ALL_RANKS: every rank receives it. Transfers are grouped point-to-point messages straight into the frame. */
VQHIP_API int  vqhip_composite_tiles(vqhip_comm* comm, void* stream, const void* tile, int width, int frame_height,
        vqhip_format fmt, int root, void* frame);

#ifdef __cplusplus
}
#endif
#endif /* VQHIP_H */
