// test_passes_engine.cpp — include/vqhip_passes.hpp built the way the ENGINE builds it: VQHIP_ENGINE_RENDERPASS_H points at (a stand-in of)
// the engine's RenderPass.h, so the adaptors derive from the engine's own ::IRenderPass and live in the container VQRenderer keeps its passes
// in (Renderer.h:403, filled at Renderer.cpp:577-585). Host-only checks (no GPU work): the types convert, the virtuals dispatch through the
// engine's base pointer, CollectPSOCreationParameters() reports nothing to compile, RecordCommands(nullptr) reports instead of crashing.
#define VQHIP_ENGINE_RENDERPASS_H "mock_engine/RenderPass.h"
#include <cstdio>
#include <memory>
#include <vector>
#include "vqhip_passes.hpp"

int main() {
    static_assert(std::is_base_of<::IRenderPass, vqhip::HipForwardLightingPass>::value, "adaptors derive from the engine's IRenderPass");
    static_assert(std::is_base_of<::IRenderPass, vqhip::HipPostProcessPass>::value && std::is_base_of<::IRenderPass, vqhip::HipEnvMapPrefilterPass>::value, "");
    std::vector<std::shared_ptr<::IRenderPass>> mRenderPasses;                       // VQRenderer::mRenderPasses
    mRenderPasses.push_back(std::make_shared<vqhip::HipForwardLightingPass>(nullptr));
    mRenderPasses.push_back(std::make_shared<vqhip::HipPostProcessPass>(nullptr));
    mRenderPasses.push_back(std::make_shared<vqhip::HipEnvMapPrefilterPass>(nullptr));
    mRenderPasses.push_back(std::make_shared<vqhip::HipSSREnvironmentFallbackPass>(nullptr));
    mRenderPasses.push_back(std::make_shared<vqhip::HipDepthPrePassNormals>(nullptr));
    for (std::shared_ptr<::IRenderPass>& pPass : mRenderPasses) {                    // the loop of Renderer.cpp:590-593
        if (pPass->Initialize()) { std::fprintf(stderr, "Initialize() must fail without a context\n"); return 1; }
        if (!pPass->CollectPSOCreationParameters().empty()) return 2;
        pPass->RecordCommands(nullptr);                                              // reports VQHIP_ERR_INVALID_ARG, never crashes
    }
    if (std::static_pointer_cast<vqhip::HipPostProcessPass>(mRenderPasses[1])->LastStatus() != VQHIP_ERR_INVALID_ARG) return 3;
    std::printf("engine-interface passes OK\n");
    return 0;
}
