// test_passes_engine.cpp — include/vqhip_passes.hpp built the way the ENGINE builds it: VQHIP_ENGINE_RENDERPASS_H points at (a stand-in of)
// the engine's RenderPass.h, so the adaptors derive from the engine's own ::IRenderPass and live in the container VQRenderer keeps its passes
// in (Renderer.h:403, filled at Renderer.cpp:577-585). Host-only checks (no GPU work): the types convert, the virtuals dispatch through the
// engine's base pointer, CollectPSOCreationParameters() reports nothing to compile, RecordCommands(nullptr) reports instead of crashing.
// Two builds: tests/cpp/Makefile (the GPU box, no reference tree) takes the stand-in below; oracle/Makefile `_ref/test_passes_engine_ref` (where
// /root/reference exists) passes -DVQHIP_ENGINE_RENDERPASS_H='"Renderer/Rendering/RenderPass/RenderPass.h"' = the reference's REAL header, and links the
// reference's RenderPass.cpp for the interface's destructors (tests/test_engine_header.py).
#ifndef VQHIP_ENGINE_RENDERPASS_H
#define VQHIP_ENGINE_RENDERPASS_H "mock_engine/RenderPass.h"
#endif
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
#ifdef VQHIP_ENGINE_HEADER_IS_THE_REFERENCES
    static_assert(NUM_RENDER_PASSES == 8 && !std::is_constructible<RenderPassBase>::value, "the reference's RenderPass.h:31-43,70-71");   // names only the real header has
    std::printf("compiled against the reference's RenderPass.h\n");
#endif
    std::printf("engine-interface passes OK\n");
    return 0;
}
