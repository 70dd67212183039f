// ssr.hip — SSR's environment-map fallback for gfx950:
//   k_ssr_env_fallback == Shaders/ScreenSpaceReflections/ClassifyReflectionTiles.hlsl:SampleEnvironmentMap :78-94 under the condition of
//                         ClassifyTiles :146-152, + g_extracted_roughness of CSMain :196; InvProjectPosition == Common.hlsl:98-104,
//                         EnvironmentBRDF == BRDF.hlsl:196-207, FresnelWithRoughness == BRDF.hlsl:152-156
// One lane per pixel; every expression runs once per pixel and is evaluated AS WRITTEN (contract v5, DESIGN.md §3: products and sums rounded one by
// one, left to right; a / b = the IEEE quotient; dot / normalize / reflect / mul in their textbook expansion). HBM-bound where the whole frame takes
// the fallback: 8 (scene colour, for its alpha) + 4 (depth) + 4 (normals) bytes read, 8 (+1) written per pixel; the cube mips and the LUT are cache resident.
#include "vq_internal.h"
#include "vq_devmath.h"
#include "vq_sampling.h"

using namespace vqd;

namespace vqk {

namespace {

// mul(M_hlsl, float4(v, w)) with the cbuffer's column-major read of a row-major XMMATRIX == the row vector (v, w) times M_cpu,
// components summed left to right as the shim's / DXC's unfused expansion does
VQD float4 mul_M_v4(const VQ_matrix& M, float x, float y, float z, float w) {
    float o[4];
    #pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = ((x * M.m[0][j] + y * M.m[1][j]) + z * M.m[2][j]) + w * M.m[3][j];
    return make_float4(o[0], o[1], o[2], o[3]);
}

template <int SCENEFMT, int NORMFMT, int OUTFMT>
__global__ __launch_bounds__(256) void k_ssr_env_fallback(SsrArgs a) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= a.width) return;
    // CSMain :193: roughness = g_roughness.Load(...).w
    float roughness;
    if (SCENEFMT == VQHIP_FMT_RGBA32F) roughness = ((const float4*)a.scene)[(size_t)y * a.scenePitch + x].w;
    else                               roughness = (float)((const _Float16*)a.scene)[((size_t)y * a.scenePitch + x) * 4 + 3];
    const float z = a.depth[(size_t)y * a.depthPitch + x];
    if (a.outRoughness) a.outRoughness[(size_t)y * a.width + x] = (uint8_t)unorm8(roughness);                      // :196, R8_UNORM store
    float4 result = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    // ClassifyTiles :146-152: is_reflective_surface (depth < far plane 1, IsReflectiveSurface :59-63) && !is_glossy_reflection (roughness < threshold, Common.hlsl:108-110)
    if ((z < 1.0f) && !(roughness < a.roughnessThreshold)) {
        f3 n01;
        if (NORMFMT == VQHIP_FMT_RGBA32F) { const float4 n = ((const float4*)a.normals)[(size_t)y * a.normalPitch + x]; n01 = mk3(n.x, n.y, n.z); }
        else { const uint32_t q = ((const uint32_t*)a.normals)[(size_t)y * a.normalPitch + x];                   // UNORM10 -> float: c / 1023, correctly rounded
               n01 = mk3(fdiv_((float)(q & 1023u), 1023.0f), fdiv_((float)((q >> 10) & 1023u), 1023.0f), fdiv_((float)((q >> 20) & 1023u), 1023.0f)); }
        const float u = ((float)x + 0.5f) * a.invDimX, v = ((float)y + 0.5f) * a.invDimY;                         // :79
        const bool dxc = a.arithDxc != 0;                                                                         // the reading of dot / normalize / reflect (vq_devmath.h)
        const f3 wn = normalize_rt(mk3(2.0f * n01.x - 1.0f, 2.0f * n01.y - 1.0f, 2.0f * n01.z - 1.0f), dxc);     // :80
        // FFX_DNSR_Reflections_ScreenSpaceToViewSpace == InvProjectPosition(coord, g_inv_proj), Common.hlsl:98-104,116-118
        const float cy = 1.0f - v;
        const float px = 2.0f * u - 1.0f, py = 2.0f * cy - 1.0f;
        const float4 pr = mul_M_v4(a.invProj, px, py, z, 1.0f);
        const f3 ray = mk3(fdiv_(pr.x, pr.w), fdiv_(pr.y, pr.w), fdiv_(pr.z, pr.w));
        const f3 dirV = normalize_rt(ray, dxc);                                                                       // :84
        const float4 nv4 = mul_M_v4(a.view, wn.x, wn.y, wn.z, 0.0f);                                              // :85
        const f3 nV = mk3(nv4.x, nv4.y, nv4.z);
        const f3 Rv = reflect_rt(dirV, nV, dxc);                                                                      // :86
        const float4 rw4 = mul_M_v4(a.invView, Rv.x, Rv.y, Rv.z, 0.0f);                                          // :87
        // mul(g_envMapRotation, float3): the float4x4 truncates to its upper-left 3x3 (HLSL's implicit truncation)
        const f3 d = mk3((rw4.x * a.rot[0][0] + rw4.y * a.rot[1][0]) + rw4.z * a.rot[2][0],
                         (rw4.x * a.rot[0][1] + rw4.y * a.rot[1][1]) + rw4.z * a.rot[2][1],
                         (rw4.x * a.rot[0][2] + rw4.y * a.rot[1][2]) + rw4.z * a.rot[2][2]);
        const float lod = roughness * (a.mipCount - 1.0f);                                                        // :89
        const float4 pre = sample_cube_lod_rgba16f(a.env.specular_cube, a.env.spec_res0, a.env.spec_mips, d, lod);
        const float NdotV = saturate(dot_rt(nV, neg(dirV), dxc));                                                     // :90
        const float2 sb = sample_2d_rg16f_clamp(a.env.brdf_lut, a.env.lut_size, a.env.lut_size, NdotV, roughness);   // :92, level 0
        // EnvironmentBRDF(NdotV, roughness, metallic = 1, diffuseColor = 0, diffuseIrradiance = 0, pre, sb), BRDF.hlsl:196-207, as written
        const float F0 = lerp_lit(0.04f, 0.0f, 1.0f);
        const float p5 = a.pow5ExpLog ? pow5_explog(1.0f - NdotV) : pow5(1.0f - NdotV);                           // FresnelWithRoughness :152-156
        const float Ks = F0 + (max_(1.0f - roughness, F0) - F0) * p5;
        const float Kd = (1.0f - Ks) * (1.0f - 1.0f);
        const float diffuse = 0.0f * 0.0f;
        const float k = Ks * sb.x + sb.y;
        result = make_float4(Kd * diffuse + pre.x * k, Kd * diffuse + pre.y * k, Kd * diffuse + pre.z * k, 0.0f);
    }
    store_px<OUTFMT>(a.out, (size_t)y * a.outPitch + x, result);                                                  // :153
}

template <int SCENEFMT, int NORMFMT> hipError_t launch_out(hipStream_t s, const SsrArgs& a, int outFmt) {
    dim3 grid((a.width + 255) / 256, a.height);
    if (outFmt == VQHIP_FMT_RGBA32F) hipLaunchKernelGGL((k_ssr_env_fallback<SCENEFMT, NORMFMT, VQHIP_FMT_RGBA32F>), grid, dim3(256), 0, s, a);
    else                             hipLaunchKernelGGL((k_ssr_env_fallback<SCENEFMT, NORMFMT, VQHIP_FMT_RGBA16F>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

} // namespace

hipError_t launch_ssr_env_fallback(hipStream_t s, const SsrArgs& a, int sceneFmt, int normalFmt, int outFmt) {
    if (sceneFmt == VQHIP_FMT_RGBA32F)
        return normalFmt == VQHIP_FMT_RGBA32F ? launch_out<VQHIP_FMT_RGBA32F, VQHIP_FMT_RGBA32F>(s, a, outFmt) : launch_out<VQHIP_FMT_RGBA32F, VQHIP_FMT_R10G10B10A2_UNORM>(s, a, outFmt);
    return normalFmt == VQHIP_FMT_RGBA32F ? launch_out<VQHIP_FMT_RGBA16F, VQHIP_FMT_RGBA32F>(s, a, outFmt) : launch_out<VQHIP_FMT_RGBA16F, VQHIP_FMT_R10G10B10A2_UNORM>(s, a, outFmt);
}

} // namespace vqk
