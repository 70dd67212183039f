// TEST INFRASTRUCTURE — a stand-in for the ENGINE'S header Source/Renderer/Rendering/RenderPass/RenderPass.h, which cannot be compiled
// here (it pulls the D3D12 pipeline headers). Only what include/vqhip_passes.hpp touches when VQHIP_ENGINE_RENDERPASS_H is defined:
// the two empty parameter bases and the pure interface with the engine's six virtuals (RenderPass.h:27-59), global namespace like the
// engine's. tests/cpp/test_passes_engine.cpp derives the adaptors from THIS ::IRenderPass and stores them the way VQRenderer stores its
// passes (std::vector<std::shared_ptr<IRenderPass>>, Renderer.h:403).
#pragma once
#include <vector>

struct FPSOCreationTaskParameters { int unused = 0; };      // PipelineStateObjects.h in the engine

struct IRenderPassResourceCollection {};
struct IRenderPassDrawParameters {};

class IRenderPass {
public:
    virtual ~IRenderPass() = 0;
    virtual bool Initialize() = 0;
    virtual void Destroy() = 0;
    virtual void OnCreateWindowSizeDependentResources(unsigned Width, unsigned Height, const IRenderPassResourceCollection* pRscParameters = nullptr) = 0;
    virtual void OnDestroyWindowSizeDependentResources() = 0;
    virtual void RecordCommands(const IRenderPassDrawParameters* pDrawParameters = nullptr) = 0;
    virtual std::vector<FPSOCreationTaskParameters> CollectPSOCreationParameters() = 0;
};
inline IRenderPass::~IRenderPass() {}
