// ref_ssr.cpp — runs the reference's SSR environment-map fallback on the CPU: SampleEnvironmentMap, IsReflectiveSurface
// (Shaders/ScreenSpaceReflections/ClassifyReflectionTiles.hlsl:59-63,78-94), FFX_DNSR_Reflections_IsGlossyReflection / InvProjectPosition
// (ScreenSpaceReflections/Common.hlsl:98-118) and EnvironmentBRDF (BRDF.hlsl:196-207) as hlsl2cpp.py generates them. The rest of ClassifyTiles
// (ray list, tile list: wave intrinsics and UAV atomics — FidelityFX SSSR, out of scope) is cut from the generated file; the three lines of it that
// produce g_intersection_output (:146-153) are restated here, the way ref_forward.cpp stands in for the rasteriser around PSMain.
// Part of oracle/_ref/libvqref_shaders.so. TEST INFRASTRUCTURE.
// Images cross this boundary as RGBA32F VALUES (roughness = scene.w, normals = the UNORM10-decoded [0,1] values, depth as a float plane); the cube and the
// LUT are fetched through ref_hooks.cpp (vqo_sampling.h: the sampling contract is the oracle's, fixed-function hardware has no source in the reference).
#include <cstring>
#include <vector>

#include "ref_hooks.h"

namespace hlsl {
namespace ssr {
#include "ScreenSpaceReflections/ClassifyReflectionTiles.hlsl"
}
} // namespace hlsl

using namespace hlsl;
using namespace vqref;

namespace {
matrix toMatrix(const VQ_matrix& h) { matrix M; for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) M.m[r][c] = h.m[c][r]; return M; }   // column-major read of the row-major XMMATRIX
}

extern "C" {

// scene / normals: RGBA32F [H][W][4]; depth: R32F [H][W]; out: RGBA32F [H][W][4] = g_intersection_output
int vqref_ssr_environment_fallback(const float* scene, const float* depth, const float* normals, int W, int H, const VQ_SSSRConstants* cb, const vqhip_envmap* env, float* out) {
    if (!scene || !depth || !normals || !cb || !env || !out) return -1;
    std::vector<float> depth4((size_t)W * H * 4, 0.0f);
    for (size_t i = 0; i < (size_t)W * H; ++i) depth4[i * 4] = depth[i];
    const Image sceneI{ scene, W, H }, depthI{ depth4.data(), W, H }, normI{ normals, W, H };
    ssr::g_roughness.res = &sceneI; ssr::g_roughness.kind = kTexImage;
    ssr::g_depth_buffer.res = &depthI; ssr::g_depth_buffer.kind = kTexImage;
    ssr::g_normal.res = &normI; ssr::g_normal.kind = kTexImage;
    ssr::g_environment_map.kind = kCubeSpecular;
    ssr::texBRDFIntegrationLUT.kind = kTexLUT;
    g_ctx.env = env;
    ssr::g_inv_view_proj = toMatrix(cb->invViewProjection); ssr::g_proj = toMatrix(cb->projection); ssr::g_inv_proj = toMatrix(cb->invProjection);
    ssr::g_view = toMatrix(cb->view); ssr::g_inv_view = toMatrix(cb->invView); ssr::g_prev_view_proj = toMatrix(cb->prevViewProjection);
    ssr::g_envMapRotation = toMatrix(cb->envMapRotation);
    ssr::g_buffer_dimensions = uint2(cb->bufferDimensions[0], cb->bufferDimensions[1]);
    ssr::g_inv_buffer_dimensions = float2(cb->inverseBufferDimensions[0], cb->inverseBufferDimensions[1]);
    ssr::g_roughness_threshold = cb->roughnessThreshold;
    ssr::g_env_map_mip_count = cb->envMapSpecularIrradianceCubemapMipLevelCount;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const uint2 dispatch_thread_id((uint)x, (uint)y);
            const float roughness = ssr::g_roughness.Load(int3(dispatch_thread_id, 0)).w;                         // CSMain :193
            // ClassifyTiles :109,112,146-153
            const bool is_reflective_surface = ssr::IsReflectiveSurface(int2(x, y), roughness);
            const bool is_glossy_reflection = ssr::FFX_DNSR_Reflections_IsGlossyReflection(roughness);
            float4 intersection_output = float4(0.0f);
            if (is_reflective_surface && !is_glossy_reflection)
                intersection_output.xyz = ssr::SampleEnvironmentMap(dispatch_thread_id, roughness, ssr::g_env_map_mip_count);
            float* o = out + ((size_t)y * W + x) * 4;
            o[0] = intersection_output.x; o[1] = intersection_output.y; o[2] = intersection_output.z; o[3] = intersection_output.w;
        }
    return 0;
}

} // extern "C"
