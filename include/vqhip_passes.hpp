// vqhip_passes.hpp — host-side C++ mirror of the reference's render-pass interface for the hot path.
//
// VQEngine's passes derive IRenderPass / RenderPassBase (Source/Renderer/Rendering/RenderPass/RenderPass.h:44-89) and take
// nested POD `FResourceCollection` / `FDrawParameters` structs (idiom: ApplyReflections.h:29-39, static_cast in
// RecordCommands: ApplyReflections.cpp:45-50). The four functions on the hot path are NOT behind that interface in the
// reference — they are VQRenderer members:
//     RenderSceneColor          Source/Renderer/Rendering/SceneRendering.cpp:1619
//     RenderPostProcess         SceneRendering.cpp:2507
//     PreFilterEnvironmentMap   Source/Renderer/Rendering/EnvironmentMapRendering.cpp:139
//     ComputeBRDFIntegrationLUT Source/Renderer/Renderer.cpp:871
// These adaptors give them the IRenderPass shape so a maintainer can register them next to the other 8 passes
// (Renderer.cpp:577-585) and call RecordCommands() where the D3D12 code recorded command lists. "Recording" here means
// enqueueing HIP kernels on a stream through the C ABI (include/vqhip.h); nothing else is linked.
// Header-only, C++17, needs the HIP runtime only for buffer allocation (hipMalloc/hipFree).
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstddef>
#include <string>
#include <vector>
#include "vqhip.h"

// The interface the adaptors derive from:
//   * inside the engine: define VQHIP_ENGINE_RENDERPASS_H to the include path of the engine's own RenderPass.h (e.g.
//     -DVQHIP_ENGINE_RENDERPASS_H='"Renderer/Rendering/RenderPass/RenderPass.h"'). The adaptors then derive from the ENGINE'S ::IRenderPass,
//     implement its CollectPSOCreationParameters() (RenderPass.h:58) as "no PSOs: kernels are compiled ahead of time", and can be stored
//     in VQRenderer::mRenderPasses (std::vector<std::shared_ptr<IRenderPass>>, Renderer.h:403) next to the other passes;
//   * stand-alone (tests, tools): the stand-in interface below with the same five virtuals.
// tests/cpp/test_passes_engine.cpp compiles the first form against a stand-in of the engine header (tests/cpp/mock_engine/).
#ifdef VQHIP_ENGINE_RENDERPASS_H
#include VQHIP_ENGINE_RENDERPASS_H
namespace vqhip {
using IRenderPassResourceCollection = ::IRenderPassResourceCollection;      // RenderPass.h:27
using IRenderPassDrawParameters = ::IRenderPassDrawParameters;              // RenderPass.h:29
class IRenderPass : public ::IRenderPass {                                  // RenderPass.h:44-59
public:
    std::vector<FPSOCreationTaskParameters> CollectPSOCreationParameters() override { return {}; }   // :58 — nothing to compile at load time
    int LastStatus() const { return mStatus; }      // the reference asserts/logs; here the last vqhip_status is kept
protected:
    int mStatus = VQHIP_OK;
};
} // namespace vqhip
#else
namespace vqhip {
struct IRenderPassResourceCollection {};            // RenderPass.h:27
struct IRenderPassDrawParameters {};                // RenderPass.h:29

class IRenderPass {                                 // RenderPass.h:44-59 (CollectPSOCreationParameters only exists in the engine build above)
public:
    virtual ~IRenderPass() = default;
    virtual bool Initialize() = 0;
    virtual void Destroy() = 0;
    virtual void OnCreateWindowSizeDependentResources(unsigned Width, unsigned Height, const IRenderPassResourceCollection* pRscParameters = nullptr) = 0;
    virtual void OnDestroyWindowSizeDependentResources() = 0;
    virtual void RecordCommands(const IRenderPassDrawParameters* pDrawParameters = nullptr) = 0;
    int LastStatus() const { return mStatus; }      // the reference asserts/logs; here the last vqhip_status is kept
protected:
    int mStatus = VQHIP_OK;
};
} // namespace vqhip
#endif

namespace vqhip {

class RenderPassBase : public IRenderPass {         // RenderPass.h:65-89: holds the renderer; here the vqhip context
protected:
    explicit RenderPassBase(vqhip_ctx* Ctx) : mCtx(Ctx) {}
    vqhip_ctx* mCtx;
    static void* Alloc(size_t bytes) { void* p = nullptr; return hipMalloc(&p, bytes) == hipSuccess ? p : nullptr; }
    static void Free(void*& p) { if (p) { (void)hipFree(p); p = nullptr; } }
};

// ---------------------------------------------------------------------------------------------------------------
// Forward lighting == VQRenderer::RenderSceneColor (SceneRendering.cpp:1619-1851, lit draws :1730-1784).
// Owns the scene-colour target (Tex_SceneColor, RGBA16F: RenderResources.cpp:40,144-161).
// ---------------------------------------------------------------------------------------------------------------
class HipForwardLightingPass : public RenderPassBase {
public:
    struct FResourceCollection : public IRenderPassResourceCollection {};
    struct FDrawParameters : public IRenderPassDrawParameters {
        void* Stream = nullptr;                                  // replaces ID3D12GraphicsCommandList* pCmd
        vqhip_gbuffer GBuffer = {};                              // replaces the rasterised PSInput + material textures
        // Alternative input: rasteriser output + materials; when pInterpolants != nullptr the pass first runs
        // vqhip_gbuffer_from_materials (PSMain :226-287) into its own G-buffer planes and GBuffer above is ignored.
        const vqhip_interpolants* pInterpolants = nullptr;
        const vqhip_material* pMaterials = nullptr;              // cbPerObject.materialData + SRV tables of every material on screen
        int NumMaterials = 0;
        const vqhip_ssao* pScreenSpaceAO = nullptr;              // Tex_AmbientOcclusion or nullptr (SSAO off)
        const VQ_PerFrameData* pPerFrame = nullptr;              // == cbPerFrame  (SceneRendering.cpp:429-450)
        const VQ_PerViewLightingData* pPerView = nullptr;        // == cbPerView   (:452-467)
        const VQ_PointLight* pExtraPointLights = nullptr;        // extension: lights beyond NUM_LIGHTS__POINT
        int NumExtraPointLights = 0;
        const vqhip_envmap* pEnvironmentMap = nullptr;           // nullptr == bDrawEnvironmentMap false (NullCubemapSRV, :1698-1709)
        const vqhip_shadowmaps* pShadowMaps = nullptr;
        // the draw's other render targets (SceneRendering.cpp:1640-1641,1662-1663 -> PSO permutations OUTPUT_ALBEDO / OUTPUT_MOTION_VECTORS)
        bool bUseVisualizationRenderTarget = false;              // SV_TARGET1 = (albedo, metalness) into Tex_SceneVisualization (RGBA16F)
        bool bRenderMotionVectors = false;                       // motion vectors into Tex_SceneMotionVectors (RG16F); needs the two planes below
        const void* pSvPositionCurr = nullptr;                   // PSInput.svPositionCurr / svPositionPrev (ForwardLighting.hlsl:49-52): float4 per pixel, tightly packed
        const void* pSvPositionPrev = nullptr;
    };
    explicit HipForwardLightingPass(vqhip_ctx* Ctx) : RenderPassBase(Ctx) {}
    ~HipForwardLightingPass() override { OnDestroyWindowSizeDependentResources(); }
    bool Initialize() override { return mCtx != nullptr; }
    void Destroy() override { OnDestroyWindowSizeDependentResources(); }
    void OnCreateWindowSizeDependentResources(unsigned Width, unsigned Height, const IRenderPassResourceCollection* = nullptr) override {
        OnDestroyWindowSizeDependentResources();
        mWidth = Width; mHeight = Height;
        mSceneColor = Alloc((size_t)Width * Height * 8);
    }
    void OnDestroyWindowSizeDependentResources() override {
        Free(mSceneColor); Free(mSceneVisualization); Free(mSceneMotionVectors);
        for (void*& g : mGB) Free(g);
        mWidth = mHeight = 0;
    }
    void RecordCommands(const IRenderPassDrawParameters* pDrawParameters = nullptr) override {
        const FDrawParameters* p = static_cast<const FDrawParameters*>(pDrawParameters);
        if (!p || !mSceneColor) { mStatus = VQHIP_ERR_INVALID_ARG; return; }
        vqhip_gbuffer gb = p->GBuffer;
        if (p->pInterpolants) {
            if (!p->pPerFrame) { mStatus = VQHIP_ERR_INVALID_ARG; return; }
            for (void*& g : mGB) if (!g) g = Alloc((size_t)mWidth * mHeight * 16);     // planes allocated on first use
            gb = { mGB[0], mGB[1], mGB[2], mGB[3], (int32_t)mWidth, (int32_t)mHeight, (int32_t)mWidth };
            mStatus = vqhip_gbuffer_from_materials(mCtx, p->Stream, p->pInterpolants, p->pMaterials, p->NumMaterials,
                                                   p->pPerFrame->fAmbientLightingFactor, p->pScreenSpaceAO, &gb);
            if (mStatus != VQHIP_OK) return;
        }
        vqhip_psmain_targets t = {};
        if (p->bUseVisualizationRenderTarget) {
            if (!mSceneVisualization) mSceneVisualization = Alloc((size_t)mWidth * mHeight * 8);
            t.albedo_metallic = mSceneVisualization; t.albedo_fmt = VQHIP_FMT_RGBA16F;
        }
        if (p->bRenderMotionVectors) {
            if (!mSceneMotionVectors) mSceneMotionVectors = Alloc((size_t)mWidth * mHeight * 4);
            t.motion_vectors = mSceneMotionVectors; t.motion_fmt = VQHIP_FMT_RG16F;
            t.svPositionCurr = p->pSvPositionCurr; t.svPositionPrev = p->pSvPositionPrev;
        }
        mStatus = vqhip_forward_lighting_mrt(mCtx, p->Stream, &gb, p->pPerFrame, p->pPerView, p->pExtraPointLights, p->NumExtraPointLights,
                                             p->pEnvironmentMap, p->pShadowMaps, mSceneColor, (int)mWidth, VQHIP_FMT_RGBA16F, &t);
    }
    void* GetSceneColor() const { return mSceneColor; }          // RGBA16F, width*height
    void* GetSceneVisualization() const { return mSceneVisualization; }   // RGBA16F (albedo, metalness); nullptr until a draw asked for it
    void* GetSceneMotionVectors() const { return mSceneMotionVectors; }   // RG16F; nullptr until a draw asked for it
    unsigned Width() const { return mWidth; }
    unsigned Height() const { return mHeight; }
private:
    void* mSceneColor = nullptr;
    void* mSceneVisualization = nullptr;                         // Tex_SceneVisualization (RenderResources.cpp:171-175)
    void* mSceneMotionVectors = nullptr;                         // Tex_SceneMotionVectors (RenderResources.cpp:178-182)
    void* mGB[4] = { nullptr, nullptr, nullptr, nullptr };       // G-buffer planes of the §8f.1 producer path
    unsigned mWidth = 0, mHeight = 0;
};

// ---------------------------------------------------------------------------------------------------------------
// Post-process == VQRenderer::RenderPostProcess (SceneRendering.cpp:2507-2788): [blur X,Y :2582-2638] -> tonemapper :2640-2656.
// Owns BlurIntermediate / BlurOutput (RGBA16F, RenderResources.cpp:221-243) and TonemapperOut (RGBA8 SDR / RGBA16F HDR, :245-261).
// Returns the buffer holding the final image like the reference returns ID3D12Resource*.
// ---------------------------------------------------------------------------------------------------------------
class HipPostProcessPass : public RenderPassBase {
public:
    struct FResourceCollection : public IRenderPassResourceCollection {};
    struct FDrawParameters : public IRenderPassDrawParameters {
        void* Stream = nullptr;
        const void* pSceneColor = nullptr;                       // RGBA16F input (HipForwardLightingPass::GetSceneColor())
        VQ_TonemapperParams TonemapperParams = { VQ_COLOR_SPACE_REC_709, VQ_DISPLAY_CURVE_SRGB, 200.0f, 1 };   // FTonemapper defaults, PostProcess.h:84-91
        bool bEnableGaussianBlur = false;                        // FPostProcessParameters::bEnableGaussianBlur (PostProcess.h:166); compiled out in the reference (:2526)
        bool bHDR = false;                                       // selects the RGBA16F tonemapper target
        // Row-tiled multi-GPU frame (no reference analogue): this pass was sized for ONE ROW TILE of the frame
        // (vqhip_rowtile) and pSceneColor is that tile. With pComm set the pass exchanges 10 halo rows with the tiles above / below — rows of SCENE COLOUR right
        // behind the shade kernel on the SDR path (one call filters X, Y and tonemaps), X-blurred rows before the Y blur on the HDR path — and afterwards
        // composites the finished tiles into pCompositeFrame on CompositeRoot
        // (VQHIP_ALL_RANKS: on every rank). pCompositeFrame: FrameHeight x Width texels of the output format, or nullptr on ranks that
        // do not receive the frame.
        vqhip_comm* pComm = nullptr;                             // its size and this process's rank are read from it (vqhip_comm_query)
        int FrameHeight = 0;
        int CompositeRoot = 0;
        void* pCompositeFrame = nullptr;
    };
    explicit HipPostProcessPass(vqhip_ctx* Ctx) : RenderPassBase(Ctx) {}
    ~HipPostProcessPass() override { OnDestroyWindowSizeDependentResources(); }
    bool Initialize() override { return mCtx != nullptr; }
    void Destroy() override { OnDestroyWindowSizeDependentResources(); }
    void OnCreateWindowSizeDependentResources(unsigned Width, unsigned Height, const IRenderPassResourceCollection* = nullptr) override {
        OnDestroyWindowSizeDependentResources();
        mWidth = Width; mHeight = Height;
        const size_t px = (size_t)Width * Height;
        mTonemapperOut = Alloc(px * 8);                          // BlurIntermediate / BlurOutput (16 B/px) are allocated on first use of the HDR path only: the SDR path
                                                                 // never touches them (one kernel for frames of >= 2^20 pixels; below that, or for a curve that mixes channels,
                                                                 // vqhip_post_process_tile blurs into the CONTEXT's scratch buffer — one more reason why a context
                                                                 // serves one thread and one stream at a time)
        mHaloTop = Alloc((size_t)VQHIP_HALO_ROWS * Width * 8); mHaloBottom = Alloc((size_t)VQHIP_HALO_ROWS * Width * 8);   // row-tiled mode only
    }
    void OnDestroyWindowSizeDependentResources() override {
        Free(mBlurIntermediate); Free(mBlurOutput); Free(mTonemapperOut); Free(mHaloTop); Free(mHaloBottom); mWidth = mHeight = 0;
        mBlurIntermediate = mBlurOutput = mTonemapperOut = mHaloTop = mHaloBottom = nullptr;
    }
    void RecordCommands(const IRenderPassDrawParameters* pDrawParameters = nullptr) override {
        const FDrawParameters* p = static_cast<const FDrawParameters*>(pDrawParameters);
        if (!p || !p->pSceneColor || !mTonemapperOut) { mStatus = VQHIP_ERR_INVALID_ARG; return; }
        mOutFormat = p->bHDR ? VQHIP_FMT_RGBA16F : VQHIP_FMT_RGBA8_UNORM;
        // Row-tiled mode: size and rank come from the communicator itself, and the tile geometry is validated BEFORE anything is enqueued on
        // it — a rank that fails here has not posted a send its neighbours would wait for.
        int world = 1, rank = 0;
        if (p->pComm) {
            vqhip_comm_info info;
            mStatus = vqhip_comm_query(p->pComm, &info);
            if (mStatus != VQHIP_OK) return;
            world = info.world; rank = info.rank;
            int row0 = 0, rows = 0;
            mStatus = vqhip_rowtile(p->FrameHeight, world, rank, &row0, &rows);
            if (mStatus == VQHIP_OK && rows != (int)mHeight) mStatus = VQHIP_ERR_INVALID_ARG;                 // the pass must have been sized for this tile
            if (mStatus == VQHIP_OK && (p->CompositeRoot == VQHIP_ALL_RANKS || p->CompositeRoot == rank) && !p->pCompositeFrame) mStatus = VQHIP_ERR_INVALID_ARG;
            if (mStatus != VQHIP_OK) return;
        }
        if (p->bEnableGaussianBlur && !p->bHDR) {
            // SDR: CSMain_X, CSMain_Y and the tonemapper (SceneRendering.cpp:2582-2656) are ONE call — one kernel for frames of >= 2^20 pixels (neither BlurIntermediate
            // nor BlurOutput touches HBM), identical bits to the separate dispatches. Row-tiled mode: exchange 1 moves the 10 boundary rows of SCENE COLOUR right behind
            // the shade kernel (the X pass is horizontal: the neighbour's rows are filtered here like the tile's own).
            const void* top = nullptr; const void* bottom = nullptr; int haloRows = 0;
            if (p->pComm) {
                mStatus = vqhip_exchange_blur_halos(p->pComm, p->Stream, p->pSceneColor, (int)mWidth, (int)mHeight, (int)mWidth, VQHIP_FMT_RGBA16F, mHaloTop, mHaloBottom);
                if (mStatus != VQHIP_OK) return;
                top = rank > 0 ? mHaloTop : nullptr; bottom = rank < world - 1 ? mHaloBottom : nullptr; haloRows = (top || bottom) ? VQHIP_HALO_ROWS : 0;
            }
            mStatus = vqhip_post_process_tile(mCtx, p->Stream, p->pSceneColor, mTonemapperOut, top, bottom, haloRows, (int)mWidth, (int)mHeight, &p->TonemapperParams,
                                              VQHIP_FMT_RGBA16F, mOutFormat);
        } else if (p->bEnableGaussianBlur) {
            // HDR (RGBA16F out): CSMain_X -> BlurIntermediate, CSMain_Y -> BlurOutput, tonemapper — the fused Y + tonemap kernel on the LDS tile is slower than two
            // dispatches there, so it keeps them.
            const VQ_BlurParams bp = { (int32_t)mWidth, (int32_t)mHeight };                                      // FBlurParams, PostProcess.h:92-96
            if (!mBlurIntermediate) { const size_t px = (size_t)mWidth * mHeight; mBlurIntermediate = Alloc(px * 8); mBlurOutput = Alloc(px * 8); }
            if (!mBlurIntermediate || !mBlurOutput) { mStatus = VQHIP_ERR_HIP; return; }
            mStatus = vqhip_gaussian_blur_x(mCtx, p->Stream, p->pSceneColor, mBlurIntermediate, &bp, VQHIP_FMT_RGBA16F);
            if (mStatus != VQHIP_OK) return;
            const void* top = nullptr; const void* bottom = nullptr; int haloRows = 0;
            if (p->pComm) {                                      // exchange 1: the 10 boundary rows of the X-blurred tile (RCCL send/recv on p->Stream)
                mStatus = vqhip_exchange_blur_halos(p->pComm, p->Stream, mBlurIntermediate, (int)mWidth, (int)mHeight, (int)mWidth, VQHIP_FMT_RGBA16F, mHaloTop, mHaloBottom);
                if (mStatus != VQHIP_OK) return;
                top = rank > 0 ? mHaloTop : nullptr; bottom = rank < world - 1 ? mHaloBottom : nullptr; haloRows = (top || bottom) ? VQHIP_HALO_ROWS : 0;
            }
            mStatus = vqhip_gaussian_blur_y(mCtx, p->Stream, mBlurIntermediate, mBlurOutput, top, bottom, haloRows, &bp, VQHIP_FMT_RGBA16F);
            if (mStatus != VQHIP_OK) return;
            mStatus = vqhip_tonemap(mCtx, p->Stream, mBlurOutput, mTonemapperOut, (int)mWidth, (int)mHeight, &p->TonemapperParams, VQHIP_FMT_RGBA16F, mOutFormat);
        } else {
            mStatus = vqhip_tonemap(mCtx, p->Stream, p->pSceneColor, mTonemapperOut, (int)mWidth, (int)mHeight, &p->TonemapperParams, VQHIP_FMT_RGBA16F, mOutFormat);
        }
        if (mStatus == VQHIP_OK && p->pComm)                     // exchange 2: the finished tiles -> the frame on the presenting rank
            mStatus = vqhip_composite_tiles(p->pComm, p->Stream, mTonemapperOut, (int)mWidth, p->FrameHeight, mOutFormat, p->CompositeRoot, p->pCompositeFrame);
    }
    void* GetOutput() const { return mTonemapperOut; }
    vqhip_format GetOutputFormat() const { return mOutFormat; }
private:
    void* mHaloTop = nullptr; void* mHaloBottom = nullptr;
    void* mBlurIntermediate = nullptr; void* mBlurOutput = nullptr; void* mTonemapperOut = nullptr;
    vqhip_format mOutFormat = VQHIP_FMT_RGBA8_UNORM;
    unsigned mWidth = 0, mHeight = 0;
};

// ---------------------------------------------------------------------------------------------------------------
// Environment-map prefilter == FEnvironmentMapRenderingResources::CreateRenderingResources (EnvironmentMapRendering.cpp:20-106)
// + VQRenderer::PreFilterEnvironmentMap (:139-486) + ComputeBRDFIntegrationLUT (Renderer.cpp:871-909).
// "Window size" plays no role; resources depend on the HDRI and the two resolutions.
// ---------------------------------------------------------------------------------------------------------------
class HipEnvMapPrefilterPass : public RenderPassBase {
public:
    struct FResourceCollection : public IRenderPassResourceCollection {
        int DiffuseIrradianceCubemapResolution = 64;             // EnvironmentMap.cpp:214
        int SpecularMapMip0Resolution = 128;                     // gfx.EnvironmentMapResolution (Settings.h:48)
        int HDRIWidth = 0, HDRIHeight = 0;
    };
    struct FDrawParameters : public IRenderPassDrawParameters {
        void* Stream = nullptr;
        const void* pEquirectRGBA32F = nullptr;                  // device, HDRIWidth x HDRIHeight float4 (level 0 only; mips are generated here)
        float DiffuseIntegrationStep = 0.010f;                   // INTEGRATION_STEP_DIFFUSE_IRRADIANCE, PipelineStateObjects.cpp:1298-1306
        vqhip_conv_order Order = VQHIP_CONV_SEQUENTIAL;                // the reference's summation order (CubemapConvolution.hlsl:132-163,186-219)
        bool bComputeBRDFLUT = true;                             // LoadDefaultResources does this once (Renderer.cpp:934)
    };
    explicit HipEnvMapPrefilterPass(vqhip_ctx* Ctx) : RenderPassBase(Ctx) {}
    ~HipEnvMapPrefilterPass() override { DestroyRenderingResources(); }
    bool Initialize() override { return mCtx != nullptr; }
    void Destroy() override { DestroyRenderingResources(); }
    void OnCreateWindowSizeDependentResources(unsigned, unsigned, const IRenderPassResourceCollection* pRsc = nullptr) override {
        const FResourceCollection* r = static_cast<const FResourceCollection*>(pRsc);
        if (!r) { mStatus = VQHIP_ERR_INVALID_ARG; return; }
        DestroyRenderingResources();
        mRsc = *r;
        mNumMips = vqhip_mip_level_count(r->HDRIWidth, r->HDRIHeight);
        mSpecMips = vqhip_specular_mip_count(r->SpecularMapMip0Resolution);
        const size_t face = (size_t)r->DiffuseIrradianceCubemapResolution * r->DiffuseIrradianceCubemapResolution * 8;
        mChain = Alloc(vqhip_mip_chain_bytes(r->HDRIWidth, r->HDRIHeight, mNumMips));
        mDiff = Alloc(6 * face); mDiffBlurred = Alloc(6 * face); mBlurTemp = Alloc(face);
        mSpec = Alloc(vqhip_cube_bytes(r->SpecularMapMip0Resolution, mSpecMips, VQHIP_FMT_RGBA16F));
        mLUT = Alloc((size_t)1024 * 1024 * 4);                   // 1024^2 RG16F, Renderer.cpp:1026-1032
    }
    void OnDestroyWindowSizeDependentResources() override { DestroyRenderingResources(); }
    void RecordCommands(const IRenderPassDrawParameters* pDrawParameters = nullptr) override {
        const FDrawParameters* p = static_cast<const FDrawParameters*>(pDrawParameters);
        if (!p || !p->pEquirectRGBA32F || !mChain) { mStatus = VQHIP_ERR_INVALID_ARG; return; }
        const size_t l0 = (size_t)mRsc.HDRIWidth * mRsc.HDRIHeight * 16;
        if (hipMemcpyAsync(mChain, p->pEquirectRGBA32F, l0, hipMemcpyDeviceToDevice, (hipStream_t)p->Stream) != hipSuccess) { mStatus = VQHIP_ERR_HIP; return; }
        mStatus = vqhip_mip_chain_min_rgba32f(mCtx, p->Stream, mChain, mRsc.HDRIWidth, mRsc.HDRIHeight, mNumMips);   // TextureManager.cpp:590-592,643-738
        if (mStatus != VQHIP_OK) return;
        const vqhip_envmap_out out = { mDiff, mDiffBlurred, mBlurTemp, mSpec };
        mStatus = vqhip_envmap_prefilter(mCtx, p->Stream, mChain, mRsc.HDRIWidth, mRsc.HDRIHeight, mNumMips, mRsc.DiffuseIrradianceCubemapResolution,
                                         p->DiffuseIntegrationStep, mRsc.SpecularMapMip0Resolution, p->Order, &out);
        if (mStatus != VQHIP_OK || !p->bComputeBRDFLUT) return;
        mStatus = vqhip_brdf_lut(mCtx, p->Stream, mLUT, 1024, 2048, VQHIP_FMT_RG16F);                             // Renderer.cpp:895-900
    }
    // what RenderSceneColor binds at SceneRendering.cpp:1698-1709
    vqhip_envmap GetEnvironmentMap() const { return vqhip_envmap{ mDiffBlurred, mRsc.DiffuseIrradianceCubemapResolution, mSpec, mRsc.SpecularMapMip0Resolution, mSpecMips, mLUT, 1024 }; }
    int GetNumSpecularIrradianceCubemapLODLevels() const { return mSpecMips; }   // -> PerViewLightingData::MaxEnvMapLODLevels (SceneRendering.cpp:463)
private:
    void DestroyRenderingResources() { Free(mChain); Free(mDiff); Free(mDiffBlurred); Free(mBlurTemp); Free(mSpec); Free(mLUT); }   // :108
    FResourceCollection mRsc;
    int mNumMips = 0, mSpecMips = 0;
    void* mChain = nullptr; void* mDiff = nullptr; void* mDiffBlurred = nullptr; void* mBlurTemp = nullptr; void* mSpec = nullptr; void* mLUT = nullptr;
};

// ---------------------------------------------------------------------------------------------------------------
// SSR environment fallback == the part of ScreenSpaceReflectionsPass::RecordCommands' "FFX DNSR ClassifyTiles" dispatch
// (ScreenSpaceReflections.cpp:262-276) that consumes the specular cube + BRDF LUT: pixels too rough for a traced ray get
// SampleEnvironmentMap (ClassifyReflectionTiles.hlsl:78-94,146-153). FResourceParameters / FDrawParameters keep the fields of
// ScreenSpaceReflections.h:33-75 this part reads (textures as device pointers, SRVs as the vqhip_envmap). Owns TexRadiance
// (RGBA16F, :129) and TexExtractedRoughness (R8_UNORM, :135); the radiance is what vqhip_apply_reflections adds to the scene colour
// when no ray was traced (the traced rays, the denoiser: FidelityFX SSSR, out of scope).
// ---------------------------------------------------------------------------------------------------------------
class HipSSREnvironmentFallbackPass : public RenderPassBase {
public:
    struct FResourceParameters : public IRenderPassResourceCollection {};
    struct FDrawParameters : public IRenderPassDrawParameters {
        void* Stream = nullptr;
        VQ_SSSRConstants ffxCBuffer = {};                        // == FDrawParameters::ffxCBuffer, filled as RenderReflections does (SceneRendering.cpp:2221-2242)
        const void* TexSceneColorRoughness = nullptr;            // RGBA16F scene colour, alpha = roughness (HipForwardLightingPass::GetSceneColor())
        const float* TexDepthHierarchy = nullptr;                // mip 0 of Tex_DownsampledSceneDepth, R32F
        const void* TexNormals = nullptr;                        // Tex_SceneNormals, R10G10B10A2_UNORM
        const vqhip_envmap* SRVEnvironmentSpecularIrradianceCubemap_BRDFIntegrationLUT = nullptr;   // the two SRVs of :268-269 (HipEnvMapPrefilterPass::GetEnvironmentMap())
    };
    explicit HipSSREnvironmentFallbackPass(vqhip_ctx* Ctx) : RenderPassBase(Ctx) {}
    ~HipSSREnvironmentFallbackPass() override { OnDestroyWindowSizeDependentResources(); }
    bool Initialize() override { return mCtx != nullptr; }
    void Destroy() override { OnDestroyWindowSizeDependentResources(); }
    void OnCreateWindowSizeDependentResources(unsigned Width, unsigned Height, const IRenderPassResourceCollection* = nullptr) override {
        OnDestroyWindowSizeDependentResources();
        mWidth = Width; mHeight = Height;
        mRadiance = Alloc((size_t)Width * Height * 8); mExtractedRoughness = Alloc((size_t)Width * Height);
    }
    void OnDestroyWindowSizeDependentResources() override { Free(mRadiance); Free(mExtractedRoughness); mWidth = mHeight = 0; }
    void RecordCommands(const IRenderPassDrawParameters* pDrawParameters = nullptr) override {
        const FDrawParameters* p = static_cast<const FDrawParameters*>(pDrawParameters);
        if (!p || !mRadiance) { mStatus = VQHIP_ERR_INVALID_ARG; return; }
        mStatus = vqhip_ssr_environment_fallback(mCtx, p->Stream, p->TexSceneColorRoughness, VQHIP_FMT_RGBA16F, 0, p->TexDepthHierarchy, 0,
                                                 p->TexNormals, VQHIP_FMT_R10G10B10A2_UNORM, 0, (int)mWidth, (int)mHeight, &p->ffxCBuffer,
                                                 p->SRVEnvironmentSpecularIrradianceCubemap_BRDFIntegrationLUT, mRadiance, VQHIP_FMT_RGBA16F, 0, (uint8_t*)mExtractedRoughness);
    }
    void* GetRadiance() const { return mRadiance; }                      // RGBA16F
    void* GetExtractedRoughness() const { return mExtractedRoughness; }  // R8_UNORM
private:
    void* mRadiance = nullptr; void* mExtractedRoughness = nullptr;
    unsigned mWidth = 0, mHeight = 0;
};

// ---------------------------------------------------------------------------------------------------------------
// Z pre-pass colour target == the pixel-shader half of VQRenderer::RenderDepthPrePass (SceneRendering.cpp:1264-1360, DepthPrePass.hlsl:PSMain): owns
// Tex_SceneNormals (R10G10B10A2_UNORM, RenderResources.cpp:185-197) — HipSSREnvironmentFallbackPass::FDrawParameters::TexNormals. The depth target of the
// same draws is the rasteriser's and stays with the caller, like the interpolant planes.
// ---------------------------------------------------------------------------------------------------------------
class HipDepthPrePassNormals : public RenderPassBase {
public:
    struct FDrawParameters : public IRenderPassDrawParameters {
        void* Stream = nullptr;
        const vqhip_interpolants* pInterpolants = nullptr;       // the rasterised PSInput (DepthPrePass.hlsl:38-49: the same five attributes the lit draws get)
        const vqhip_material* pMaterials = nullptr;              // cbPerObject.materialData + texDiffuse / texNormals of every material on screen
        int NumMaterials = 0;
    };
    explicit HipDepthPrePassNormals(vqhip_ctx* Ctx) : RenderPassBase(Ctx) {}
    ~HipDepthPrePassNormals() override { OnDestroyWindowSizeDependentResources(); }
    bool Initialize() override { return mCtx != nullptr; }
    void Destroy() override { OnDestroyWindowSizeDependentResources(); }
    void OnCreateWindowSizeDependentResources(unsigned Width, unsigned Height, const IRenderPassResourceCollection* = nullptr) override {
        OnDestroyWindowSizeDependentResources();
        mWidth = Width; mHeight = Height;
        mSceneNormals = Alloc((size_t)Width * Height * 4);
    }
    void OnDestroyWindowSizeDependentResources() override { Free(mSceneNormals); mWidth = mHeight = 0; }
    void RecordCommands(const IRenderPassDrawParameters* pDrawParameters = nullptr) override {
        const FDrawParameters* p = static_cast<const FDrawParameters*>(pDrawParameters);
        if (!p || !mSceneNormals) { mStatus = VQHIP_ERR_INVALID_ARG; return; }
        mStatus = vqhip_scene_normals_from_materials(mCtx, p->Stream, p->pInterpolants, p->pMaterials, p->NumMaterials, mSceneNormals, VQHIP_FMT_R10G10B10A2_UNORM, (int)mWidth);
    }
    void* GetSceneNormals() const { return mSceneNormals; }      // R10G10B10A2_UNORM, width*height uint32
private:
    void* mSceneNormals = nullptr;
    unsigned mWidth = 0, mHeight = 0;
};

} // namespace vqhip
