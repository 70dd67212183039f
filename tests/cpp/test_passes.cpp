// test_passes.cpp — drives the IRenderPass-shaped C++ adaptors (include/vqhip_passes.hpp) exactly the way a VQEngine
// maintainer would from VQRenderer: create window-size resources, fill FDrawParameters, RecordCommands().
// Inputs/outputs are raw files in a directory given on the command line (written/checked by tests/test_gpu_passes.py).
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "vqhip_passes.hpp"

static std::vector<char> readFile(const std::string& p) {
    FILE* f = fopen(p.c_str(), "rb"); if (!f) { fprintf(stderr, "cannot open %s\n", p.c_str()); exit(2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<char> b(n); if (fread(b.data(), 1, n, f) != (size_t)n) exit(2); fclose(f); return b;
}
static void writeDev(const std::string& p, const void* dev, size_t bytes) {
    std::vector<char> h(bytes);
    if (hipMemcpy(h.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "copy back failed\n"); exit(3); }
    FILE* f = fopen(p.c_str(), "wb"); fwrite(h.data(), 1, bytes, f); fclose(f);
}
static void* upload(const std::vector<char>& h) {
    void* d = nullptr;
    if (hipMalloc(&d, h.size()) != hipSuccess || hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "upload failed\n"); exit(3); }
    return d;
}
#define CHECK(pass) do { if ((pass).LastStatus() != VQHIP_OK) { fprintf(stderr, "%s failed: %d %s\n", #pass, (pass).LastStatus(), vqhip_last_error(ctx)); return 4; } } while (0)

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: test_passes <dir> <W> <H> <eqW> <eqH>\n"); return 1; }
    const std::string dir = argv[1];
    const int W = atoi(argv[2]), H = atoi(argv[3]), eqW = atoi(argv[4]), eqH = atoi(argv[5]);
    vqhip_ctx* ctx = nullptr;
    if (vqhip_create(0, &ctx) != VQHIP_OK) { fprintf(stderr, "vqhip_create: %s\n", vqhip_last_error(nullptr)); return 3; }
    hipStream_t stream; (void)hipStreamCreate(&stream);

    // --- load time: environment map (VQEngine::LoadEnvironmentMap -> CreateRenderingResources + PreFilterEnvironmentMap)
    vqhip::HipEnvMapPrefilterPass envPass(ctx);
    envPass.Initialize();
    vqhip::HipEnvMapPrefilterPass::FResourceCollection envRsc;
    envRsc.DiffuseIrradianceCubemapResolution = 8; envRsc.SpecularMapMip0Resolution = 16; envRsc.HDRIWidth = eqW; envRsc.HDRIHeight = eqH;
    envPass.OnCreateWindowSizeDependentResources(0, 0, &envRsc);
    void* dEq = upload(readFile(dir + "/equirect.bin"));
    vqhip::HipEnvMapPrefilterPass::FDrawParameters envDraw;
    envDraw.Stream = stream; envDraw.pEquirectRGBA32F = dEq; envDraw.DiffuseIntegrationStep = 0.1f;
    envPass.RecordCommands(&envDraw);
    CHECK(envPass);
    const vqhip_envmap env = envPass.GetEnvironmentMap();

    // --- per frame: RenderSceneColor -> RenderPostProcess
    vqhip::HipForwardLightingPass lighting(ctx);
    vqhip::HipPostProcessPass post(ctx);
    lighting.Initialize(); post.Initialize();
    lighting.OnCreateWindowSizeDependentResources(W, H);
    post.OnCreateWindowSizeDependentResources(W, H);
    std::vector<char> pf = readFile(dir + "/perframe.bin"), pv = readFile(dir + "/perview.bin");
    if (pf.size() != sizeof(VQ_PerFrameData) || pv.size() != sizeof(VQ_PerViewLightingData)) { fprintf(stderr, "cbuffer size mismatch\n"); return 2; }
    ((VQ_PerViewLightingData*)pv.data())->MaxEnvMapLODLevels = (float)envPass.GetNumSpecularIrradianceCubemapLODLevels();   // SceneRendering.cpp:463
    vqhip::HipForwardLightingPass::FDrawParameters ld;
    ld.Stream = stream;
    ld.GBuffer = vqhip_gbuffer{ upload(readFile(dir + "/gb0.bin")), upload(readFile(dir + "/gb1.bin")), upload(readFile(dir + "/gb2.bin")), upload(readFile(dir + "/gb3.bin")), W, H, W };
    ld.pPerFrame = (const VQ_PerFrameData*)pf.data();
    ld.pPerView = (const VQ_PerViewLightingData*)pv.data();
    ld.pEnvironmentMap = &env;
    lighting.RecordCommands(&ld);
    CHECK(lighting);
    vqhip::HipPostProcessPass::FDrawParameters pd;
    pd.Stream = stream; pd.pSceneColor = lighting.GetSceneColor(); pd.bEnableGaussianBlur = true;
    post.RecordCommands(&pd);
    CHECK(post);
    if (hipStreamSynchronize(stream) != hipSuccess) { fprintf(stderr, "stream sync failed\n"); return 3; }
    writeDev(dir + "/scene_rgba16f.bin", lighting.GetSceneColor(), (size_t)W * H * 8);
    writeDev(dir + "/sdr_rgba8.bin", post.GetOutput(), (size_t)W * H * 4);
    writeDev(dir + "/diffuse_blurred.bin", env.diffuse_cube, (size_t)6 * 8 * 8 * 8);
    // --- §8f.4: SSR's environment fallback on the frame just lit (ssr_cb.bin present): scene colour (alpha = roughness) + depth + normals -> radiance
    if (FILE* probe = fopen((dir + "/ssr_cb.bin").c_str(), "rb")) {
        fclose(probe);
        std::vector<char> cb = readFile(dir + "/ssr_cb.bin");
        if (cb.size() != sizeof(VQ_SSSRConstants)) { fprintf(stderr, "ssr_cb.bin size\n"); return 2; }
        vqhip::HipSSREnvironmentFallbackPass ssr(ctx);
        ssr.Initialize();
        ssr.OnCreateWindowSizeDependentResources(W, H);
        vqhip::HipSSREnvironmentFallbackPass::FDrawParameters sd;
        sd.Stream = stream;
        sd.ffxCBuffer = *(const VQ_SSSRConstants*)cb.data();
        sd.TexSceneColorRoughness = lighting.GetSceneColor();
        sd.TexDepthHierarchy = (const float*)upload(readFile(dir + "/ssr_depth.bin"));
        sd.TexNormals = upload(readFile(dir + "/ssr_normals.bin"));
        sd.SRVEnvironmentSpecularIrradianceCubemap_BRDFIntegrationLUT = &env;
        ssr.RecordCommands(&sd);
        CHECK(ssr);
        if (hipStreamSynchronize(stream) != hipSuccess) return 3;
        writeDev(dir + "/ssr_radiance_rgba16f.bin", ssr.GetRadiance(), (size_t)W * H * 8);
        writeDev(dir + "/ssr_roughness_r8.bin", ssr.GetExtractedRoughness(), (size_t)W * H);
        ssr.RecordCommands(nullptr);
        if (ssr.LastStatus() != VQHIP_ERR_INVALID_ARG) return 5;
        ssr.Destroy();
    }
    // --- §8f.1 path: rasteriser planes + a material table (texture pointers in materials.bin are NULL: texture-less
    // materials) -> producer -> lighting inside the same pass
    if (FILE* probe = fopen((dir + "/ip0.bin").c_str(), "rb")) {
        fclose(probe);
        std::vector<char> mats = readFile(dir + "/materials.bin");
        if (mats.size() % sizeof(vqhip_material) != 0) { fprintf(stderr, "materials.bin size\n"); return 2; }
        vqhip_interpolants ip = { upload(readFile(dir + "/ip0.bin")), upload(readFile(dir + "/ip1.bin")), upload(readFile(dir + "/ip2.bin")), W, H, W };
        ld.pInterpolants = &ip;
        ld.pMaterials = (const vqhip_material*)mats.data();
        ld.NumMaterials = (int)(mats.size() / sizeof(vqhip_material));
        // the Z pre-pass's colour target on the same planes (RenderDepthPrePass): Tex_SceneNormals
        {
            vqhip::HipDepthPrePassNormals zpp(ctx);
            zpp.Initialize();
            zpp.OnCreateWindowSizeDependentResources(W, H);
            vqhip::HipDepthPrePassNormals::FDrawParameters zd;
            zd.Stream = stream; zd.pInterpolants = &ip; zd.pMaterials = ld.pMaterials; zd.NumMaterials = ld.NumMaterials;
            zpp.RecordCommands(&zd);
            CHECK(zpp);
            if (hipStreamSynchronize(stream) != hipSuccess) return 3;
            writeDev(dir + "/scene_normals_r10g10b10a2.bin", zpp.GetSceneNormals(), (size_t)W * H * 4);
            zpp.RecordCommands(nullptr);
            if (zpp.LastStatus() != VQHIP_ERR_INVALID_ARG) return 5;
            zpp.Destroy();
        }
        // the draw's other render targets (sv_curr.bin present): RenderSceneColor with bUseVisualizationRenderTarget + bRenderMotionVectors (SceneRendering.cpp:1640-1663)
        bool mrt = false;
        if (FILE* sv = fopen((dir + "/sv_curr.bin").c_str(), "rb")) {
            fclose(sv); mrt = true;
            ld.bUseVisualizationRenderTarget = true; ld.bRenderMotionVectors = true;
            ld.pSvPositionCurr = upload(readFile(dir + "/sv_curr.bin")); ld.pSvPositionPrev = upload(readFile(dir + "/sv_prev.bin"));
        }
        lighting.RecordCommands(&ld);
        CHECK(lighting);
        if (hipStreamSynchronize(stream) != hipSuccess) return 3;
        writeDev(dir + "/scene_ip_rgba16f.bin", lighting.GetSceneColor(), (size_t)W * H * 8);
        if (mrt) {
            writeDev(dir + "/scene_viz_rgba16f.bin", lighting.GetSceneVisualization(), (size_t)W * H * 8);
            writeDev(dir + "/scene_mv_rg16f.bin", lighting.GetSceneMotionVectors(), (size_t)W * H * 4);
            ld.pSvPositionCurr = nullptr;                   // motion vectors without the clip positions: refused, not crashed
            lighting.RecordCommands(&ld);
            if (lighting.LastStatus() != VQHIP_ERR_INVALID_ARG) return 5;
            ld.bRenderMotionVectors = false; ld.bUseVisualizationRenderTarget = false;
        }
    }
    // --- row-tiled mode of the post pass through the real RCCL: a world of one rank (the box has one GPU). No halos, the composite of the
    // single tile is the tile: the frame must equal the SDR image written above.
    if (argc > 6 && std::string(argv[6]) == "rccl") {
        char id[VQHIP_COMM_ID_BYTES];
        vqhip_comm* comm = nullptr;
        if (vqhip_comm_unique_id(id) != VQHIP_OK || vqhip_comm_create(id, 1, 0, &comm) != VQHIP_OK) { fprintf(stderr, "comm: %s\n", vqhip_last_error(nullptr)); return 6; }
        void* frame = nullptr;
        if (hipMalloc(&frame, (size_t)W * H * 4) != hipSuccess) return 3;
        pd.pComm = comm; pd.FrameHeight = H; pd.CompositeRoot = 0; pd.pCompositeFrame = frame;
        post.RecordCommands(&pd);
        CHECK(post);
        if (hipStreamSynchronize(stream) != hipSuccess) return 3;
        writeDev(dir + "/frame_rgba8.bin", frame, (size_t)W * H * 4);
        pd.pComm = nullptr;
        vqhip_comm_destroy(comm);
        (void)hipFree(frame);
    }
    // error behaviour: RecordCommands without parameters reports, never crashes
    lighting.RecordCommands(nullptr);
    if (lighting.LastStatus() != VQHIP_ERR_INVALID_ARG) return 5;
    lighting.Destroy(); post.Destroy(); envPass.Destroy();
    vqhip_destroy(ctx);
    printf("passes OK\n");
    return 0;
}
