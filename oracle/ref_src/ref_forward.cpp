// ref_forward.cpp — harness that RUNS the reference's own ForwardLighting.hlsl:PSMain (with BRDF.hlsl, Lighting.hlsl,
// ShadingMath.hlsl, LightingConstantBufferData.h) on the CPU: the sources are read where they lie under /root/reference,
// rewritten syntactically by hlsl2cpp.py into oracle/_ref/gen/ and compiled here against hlsl_shim.h.
// Output: part of oracle/_ref/libvqref_shaders.so (git-ignored). TEST INFRASTRUCTURE: used by tests/golden/make_ref_fixtures.py
// and tests/test_ref_pinning.py to pin the oracle's restatement (vqo_oracle.cpp, vqo_gbuffer.cpp); never loaded by the product.
//
// What is the reference's and what is this file's:
//   * every arithmetic statement between the texture fetches and the returned colour is the reference's source;
//   * texture FETCHES are fixed-function hardware without source in the reference: the hooks below implement them with the
//     oracle's sampling contract (vqo_sampling.h), including the implicit-derivative rule of vqo_gbuffer.cpp;
//   * the cbuffer fill (host struct -> HLSL struct) follows the D3D packing: `matrix` is column_major, so the HLSL
//     M[r][c] is the host's m[c][r] (the host writes XMMATRIX rows).
#include <cstdint>
#include <cstring>

#include "../../include/vqhip.h"
#include "../vqo_math.h"
#include "../vqo_sampling.h"
#include "ref_hooks.h"

#if ENABLE_ALPHA_MASK
// the "_AlphaMasked" PSO permutation (PipelineStateObjects.cpp:1571) — built as its own library, libvqref_shaders_am.so. HLSL's `discard`
// ends the invocation: here it flags the pixel and returns from PSMain (the only function that uses it, ForwardLighting.hlsl:239).
#define discard do { vqref::g_ctx.discarded = true; return PSOutput{}; } while (0)
#endif

namespace hlsl {
// ---- the reference's shader, verbatim apart from hlsl2cpp.py's syntactic rewrites ------------------------------------
namespace fwd {
#define VQ_GPU 1
#include "ForwardLighting.hlsl"
#undef f2
#undef f3
} // namespace fwd
} // namespace hlsl

namespace {
using namespace hlsl;
using namespace hlsl::fwd;
using namespace vqref;

float3 v3(const VQ_float3& a) { return float3(a.x, a.y, a.z); }
matrix toMatrix(const VQ_matrix& h) { matrix M; for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) M.m[r][c] = h.m[c][r]; return M; }
PointLight toPoint(const VQ_PointLight& l) {
    PointLight o; o.position = v3(l.position); o.range = l.range; o.color = v3(l.color); o.brightness = l.brightness;
    o.attenuation = v3(l.attenuation); o.depthBias = l.depthBias; return o;
}
SpotLight toSpot(const VQ_SpotLight& l) {
    SpotLight o; o.position = v3(l.position); o.outerConeAngle = l.outerConeAngle; o.color = v3(l.color); o.brightness = l.brightness;
    o.spotDir = v3(l.spotDir); o.depthBias = l.depthBias; o.innerConeAngle = l.innerConeAngle; o.range = l.range; o.dummy1 = l.dummy1; o.dummy2 = l.dummy2;
    return o;
}
void fillFrame(const VQ_PerFrameData& F, const VQ_PerViewLightingData& V) {
    const VQ_SceneLighting& L = F.Lights;
    SceneLighting& o = cbPerFrame.Lights;
    o.numPointLights = L.numPointLights; o.numSpotLights = L.numSpotLights; o.numPointCasters = L.numPointCasters; o.numSpotCasters = L.numSpotCasters;
    o.directional.lightDirection = v3(L.directional.lightDirection); o.directional.brightness = L.directional.brightness;
    o.directional.color = v3(L.directional.color); o.directional.depthBias = L.directional.depthBias;
    o.directional.shadowing = L.directional.shadowing; o.directional.enabled = L.directional.enabled;
    o.shadowViewDirectional = toMatrix(L.shadowViewDirectional);
    for (int i = 0; i < VQ_NUM_LIGHTS__POINT; ++i) o.point_lights[i] = toPoint(L.point_lights[i]);
    for (int i = 0; i < VQ_NUM_SHADOWING_LIGHTS__POINT; ++i) o.point_casters[i] = toPoint(L.point_casters[i]);
    for (int i = 0; i < VQ_NUM_LIGHTS__SPOT; ++i) o.spot_lights[i] = toSpot(L.spot_lights[i]);
    for (int i = 0; i < VQ_NUM_SHADOWING_LIGHTS__SPOT; ++i) { o.spot_casters[i] = toSpot(L.spot_casters[i]); o.shadowViews[i] = toMatrix(L.shadowViews[i]); }
    cbPerFrame.f2PointLightShadowMapDimensions = float2(F.f2PointLightShadowMapDimensions.x, F.f2PointLightShadowMapDimensions.y);
    cbPerFrame.f2SpotLightShadowMapDimensions = float2(F.f2SpotLightShadowMapDimensions.x, F.f2SpotLightShadowMapDimensions.y);
    cbPerFrame.f2DirectionalLightShadowMapDimensions = float2(F.f2DirectionalLightShadowMapDimensions.x, F.f2DirectionalLightShadowMapDimensions.y);
    cbPerFrame.fAmbientLightingFactor = F.fAmbientLightingFactor;
    cbPerFrame.fHDRIOffsetInRadians = F.fHDRIOffsetInRadians;
    cbPerView.matView = toMatrix(V.matView); cbPerView.matViewToWorld = toMatrix(V.matViewToWorld); cbPerView.matProjInverse = toMatrix(V.matProjInverse);
    for (int i = 0; i < 6; ++i) cbPerView.WorldFrustumPlanes[i] = float4(V.WorldFrustumPlanes[i].x, V.WorldFrustumPlanes[i].y, V.WorldFrustumPlanes[i].z, V.WorldFrustumPlanes[i].w);
    cbPerView.CameraPosition = v3(V.CameraPosition); cbPerView.MaxEnvMapLODLevels = V.MaxEnvMapLODLevels;
    cbPerView.ScreenDimensions = float2(V.ScreenDimensions.x, V.ScreenDimensions.y);
    cbPerView.EnvironmentMapDiffuseOnlyIllumination = V.EnvironmentMapDiffuseOnlyIllumination; cbPerView.pad1 = V.pad1;
}
void fillMaterial(const VQ_MaterialData& m) {
    MaterialData& o = cbPerObject.materialData;
    o.diffuse = v3(m.diffuse); o.alpha = m.alpha; o.emissiveColor = v3(m.emissiveColor); o.emissiveIntensity = m.emissiveIntensity;
    o.specular = v3(m.specular); o.normalMapMipBias = m.normalMapMipBias;
    o.uvScaleOffset = float4(m.uvScaleOffset.x, m.uvScaleOffset.y, m.uvScaleOffset.z, m.uvScaleOffset.w);
    o.roughness = m.roughness; o.metalness = m.metalness; o.displacement = m.displacement; o.textureConfig = m.textureConfig;
}
void bindScene(const vqhip_envmap* env, const vqhip_shadowmaps* sm) {
    g_ctx.env = env; g_ctx.sm = sm;
    texEnvMapDiff.kind = env ? kCubeDiffuse : kTexNull; texEnvMapSpec.kind = env ? kCubeSpecular : kTexNull; texBRDFIntegral.kind = env ? kTexLUT : kTexNull;
    texDirectionalLightShadowMap.kind = sm && sm->directional ? kTexShadowDir : kTexNull;
    texSpotLightShadowMaps.kind = sm && sm->spot ? kArrSpot : kTexNull;
    texPointLightShadowMaps.kind = sm && sm->point ? kArrPoint : kTexNull;
}
void bindTex(Texture2D& t, const vqhip_texture2d& d) { t.res = &d; t.kind = d.texels ? kTexMaterial : kTexNull; }
int32_t matIndex(const float* ip2, size_t o) { int32_t i; std::memcpy(&i, ip2 + o + 3, 4); return i; }
} // namespace

extern "C" {

// PSMain over an image of interpolants (include/vqhip.h vqhip_interpolants) + material table: the reference's whole pixel
// shader, i.e. what the oracle splits into vqo_gbuffer_from_materials + vqo_forward_lighting. out = RGBA32F [H][W][4].
// Pixels whose material index is outside the table get zeros (no geometry; not a reference concept).
// The permutation with the other render targets (the "mrt" build: -DOUTPUT_ALBEDO=1 -DOUTPUT_MOTION_VECTORS=1, PipelineStateObjects.cpp:1547-1563):
//   svCurr / svPrev : float4 planes = PSInput.svPositionCurr / svPositionPrev (:49-52), svPitch pixels per row
//   outAlbedo [H][W][4], outMotion [H][W][2] : PSOutput.albedo_metallic / motion_vectors as float32 (the render targets' fp16 rounding is the caller's)
static int psmainImage(const vqhip_interpolants* in, const vqhip_material* mats, int nMats, const vqhip_ssao* ssao,
                       const VQ_PerFrameData* pf, const VQ_PerViewLightingData* pv, const vqhip_envmap* env, const vqhip_shadowmaps* sm, float* out,
                       const float* svCurr, const float* svPrev, int svPitch, float* outAlbedo, float* outMotion);
int vqref_forward_psmain(const vqhip_interpolants* in, const vqhip_material* mats, int nMats, const vqhip_ssao* ssao,
                         const VQ_PerFrameData* pf, const VQ_PerViewLightingData* pv, const vqhip_envmap* env,
                         const vqhip_shadowmaps* sm, float* out) {
    return psmainImage(in, mats, nMats, ssao, pf, pv, env, sm, out, nullptr, nullptr, 0, nullptr, nullptr);
}
int vqref_forward_psmain_mrt(const vqhip_interpolants* in, const vqhip_material* mats, int nMats, const vqhip_ssao* ssao,
                             const VQ_PerFrameData* pf, const VQ_PerViewLightingData* pv, const vqhip_envmap* env, const vqhip_shadowmaps* sm, float* out,
                             const float* svCurr, const float* svPrev, int svPitch, float* outAlbedo, float* outMotion) {
#if PS_OUTPUT_ALBEDO_METALLIC && PS_OUTPUT_MOTION_VECTORS
    if (!svCurr || !svPrev || !outAlbedo || !outMotion) return -1;
    return psmainImage(in, mats, nMats, ssao, pf, pv, env, sm, out, svCurr, svPrev, svPitch, outAlbedo, outMotion);
#else
    return -3;                                              // this build is a permutation without the extra targets
#endif
}
static int psmainImage(const vqhip_interpolants* in, const vqhip_material* mats, int nMats, const vqhip_ssao* ssao,
                       const VQ_PerFrameData* pf, const VQ_PerViewLightingData* pv, const vqhip_envmap* env, const vqhip_shadowmaps* sm, float* out,
                       const float* svCurr, const float* svPrev, int svPitch, float* outAlbedo, float* outMotion) {
    if (!in || !pf || !pv || !out) return -1;
    fillFrame(*pf, *pv);
    bindScene(env, sm);
    static const vqhip_ssao none{};
    texScreenSpaceAO.res = ssao ? ssao : &none;
    texScreenSpaceAO.kind = ssao && ssao->texels ? kTexSSAO : kTexOne;    // the engine binds a white texture when SSAO is off
    const float* ip0 = (const float*)in->ip0; const float* ip1 = (const float*)in->ip1; const float* ip2 = (const float*)in->ip2;
    const int W = in->width, H = in->height, P = in->row_pitch_px;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t o = ((size_t)y * P + x) * 4;
            float* dst = out + ((size_t)y * W + x) * 4;
            const int idx = matIndex(ip2, o);
            float* dstA = outAlbedo ? outAlbedo + ((size_t)y * W + x) * 4 : nullptr;
            float* dstM = outMotion ? outMotion + ((size_t)y * W + x) * 2 : nullptr;
            if (idx < 0 || idx >= nMats) {
                dst[0] = dst[1] = dst[2] = dst[3] = 0.0f;
                if (dstA) dstA[0] = dstA[1] = dstA[2] = dstA[3] = 0.0f;
                if (dstM) dstM[0] = dstM[1] = 0.0f;                                  // the targets' clear value (SceneRendering.cpp:1659-1668)
                continue;
            }
            const vqhip_material& mt = mats[idx];
            fillMaterial(mt.data);
            bindTex(texDiffuse, mt.texDiffuse); bindTex(texNormals, mt.texNormals); bindTex(texEmissive, mt.texEmissive);
            bindTex(texMetalness, mt.texMetalness); bindTex(texRoughness, mt.texRoughness);
            bindTex(texOcclRoughMetal, mt.texOcclRoughMetal); bindTex(texLocalAO, mt.texLocalAO);
            // implicit derivatives of the transformed uv over the 2x2 quad (contract of vqo_gbuffer.cpp): the uv expression is PSMain's
            auto uvAt = [&](int xx, int yy) {
                const size_t q = ((size_t)yy * P + xx) * 4;
                const float2 uv = float2(ip0[q + 3], ip1[q + 3]) * cbPerObject.materialData.uvScaleOffset.xy + cbPerObject.materialData.uvScaleOffset.zw;
                return vqo::f2{ uv.x, uv.y };
            };
            g_ctx.ddx = { 0, 0 }; g_ctx.ddy = { 0, 0 };
            const int xa = x & ~1, xb = x | 1, ya = y & ~1, yb = y | 1;
            if (xb < W && matIndex(ip2, ((size_t)y * P + xa) * 4) == idx && matIndex(ip2, ((size_t)y * P + xb) * 4) == idx) {
                const vqo::f2 a = uvAt(xa, y), b = uvAt(xb, y); g_ctx.ddx = { b.x - a.x, b.y - a.y };
            }
            if (yb < H && matIndex(ip2, ((size_t)ya * P + x) * 4) == idx && matIndex(ip2, ((size_t)yb * P + x) * 4) == idx) {
                const vqo::f2 a = uvAt(x, ya), b = uvAt(x, yb); g_ctx.ddy = { b.x - a.x, b.y - a.y };
            }
            PSInput In;
            In.position = float4((float)x + 0.5f, (float)y + 0.5f, 0.0f, 1.0f);     // SV_Position: pixel centre
            In.WorldSpacePosition = float3(ip0[o], ip0[o + 1], ip0[o + 2]);
            In.WorldSpaceNormal = float3(ip1[o], ip1[o + 1], ip1[o + 2]);
            In.WorldSpaceTangent = float3(ip2[o], ip2[o + 1], ip2[o + 2]);
            In.uv = float2(ip0[o + 3], ip1[o + 3]);
#if PS_OUTPUT_MOTION_VECTORS
            if (svCurr) {
                const float* c = svCurr + ((size_t)y * svPitch + x) * 4; const float* p = svPrev + ((size_t)y * svPitch + x) * 4;
                In.svPositionCurr = float4(c[0], c[1], c[2], c[3]); In.svPositionPrev = float4(p[0], p[1], p[2], p[3]);
            }
#endif
            g_ctx.discarded = false;
            const PSOutput r = PSMain(In);
            if (g_ctx.discarded) { dst[0] = dst[1] = dst[2] = dst[3] = -1.0f; continue; }      // sentinel: no colour is negative
            dst[0] = r.color.x; dst[1] = r.color.y; dst[2] = r.color.z; dst[3] = r.color.w;
#if PS_OUTPUT_ALBEDO_METALLIC && PS_OUTPUT_MOTION_VECTORS
            if (dstA) { dstA[0] = r.albedo_metallic.x; dstA[1] = r.albedo_metallic.y; dstA[2] = r.albedo_metallic.z; dstA[3] = r.albedo_metallic.w; }
            if (dstM) { dstM[0] = r.motion_vectors.x; dstM[1] = r.motion_vectors.y; }
#endif
        }
    return 0;
}

// PSMain driven from a G-buffer (the product's boundary for the lighting half): each pixel becomes a texture-less material
// (textureConfig 0, null SRVs) whose constants are the G-buffer values, ao arrives as fAmbientLightingFactor with a white SSAO
// texture, the normal as the interpolated WorldSpaceNormal. out = RGBA32F [H][W][4].
//   extra / nExtra: point lights beyond the cbuffer's 100 (the product's extension array, include/vqhip.h). Only the build with the
//   cap raised (-DNUM_LIGHTS__POINT=256, libvqref_shaders_l256.so) has room for them: they continue point_lights[] at index 100.
int vqref_forward_from_gbuffer(const vqhip_gbuffer* gb, const VQ_PerFrameData* pf, const VQ_PerViewLightingData* pv,
                               const VQ_PointLight* extra, int nExtra, const vqhip_envmap* env, const vqhip_shadowmaps* sm, float* out) {
    if (!gb || !pf || !pv || !out) return -1;
    if (nExtra < 0 || (nExtra > 0 && (!extra || pf->Lights.numPointLights != VQ_NUM_LIGHTS__POINT || VQ_NUM_LIGHTS__POINT + nExtra > NUM_LIGHTS__POINT))) return -2;
    fillFrame(*pf, *pv);
    for (int i = 0; i < nExtra; ++i) cbPerFrame.Lights.point_lights[VQ_NUM_LIGHTS__POINT + i] = toPoint(extra[i]);
    cbPerFrame.Lights.numPointLights += nExtra;
    bindScene(env, sm);
    texScreenSpaceAO.kind = kTexOne;
    Texture2D* mts[] = { &texDiffuse, &texNormals, &texEmissive, &texMetalness, &texRoughness, &texOcclRoughMetal, &texLocalAO };
    for (Texture2D* t : mts) { t->res = nullptr; t->kind = kTexNull; }
    g_ctx.ddx = { 0, 0 }; g_ctx.ddy = { 0, 0 };
    const float* g0 = (const float*)gb->gb0; const float* g1 = (const float*)gb->gb1; const float* g2 = (const float*)gb->gb2; const float* g3 = (const float*)gb->gb3;
    const int W = gb->width, H = gb->height, P = gb->row_pitch_px;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t o = ((size_t)y * P + x) * 4;
            VQ_MaterialData m{};
            m.diffuse = { g2[o], g2[o + 1], g2[o + 2] }; m.metalness = g2[o + 3];
            m.emissiveColor = { g3[o], g3[o + 1], g3[o + 2] }; m.emissiveIntensity = g3[o + 3];
            m.roughness = g1[o + 3]; m.uvScaleOffset = { 1, 1, 0, 0 }; m.textureConfig = 0.0f;
            fillMaterial(m);
            cbPerFrame.fAmbientLightingFactor = g0[o + 3];
            PSInput In;
            In.position = float4((float)x + 0.5f, (float)y + 0.5f, 0.0f, 1.0f);
            In.WorldSpacePosition = float3(g0[o], g0[o + 1], g0[o + 2]);
            In.WorldSpaceNormal = float3(g1[o], g1[o + 1], g1[o + 2]);
            In.WorldSpaceTangent = float3(1, 0, 0);
            In.uv = float2(0, 0);
            const PSOutput r = PSMain(In);
            float* dst = out + ((size_t)y * W + x) * 4;
            dst[0] = r.color.x; dst[1] = r.color.y; dst[2] = r.color.z; dst[3] = r.color.w;
        }
    return 0;
}

// single functions of BRDF.hlsl / Lighting.hlsl / ShadingMath.hlsl for known-answer checks
void vqref_brdf(const float* N, float roughness, const float* albedo, float metalness, const float* Wi, const float* V, float* out3) {
    BRDF_Surface s = BRDF_Surface{};
    s.N = float3(N[0], N[1], N[2]); s.roughness = roughness; s.diffuseColor = float3(albedo[0], albedo[1], albedo[2]); s.metalness = metalness;
    const float3 r = BRDF(s, float3(Wi[0], Wi[1], Wi[2]), float3(V[0], V[1], V[2]));
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
void vqref_integrate_brdf(float NdotV, float roughness, int samples, float* out2) {
    const float2 r = IntegrateBRDF(NdotV, roughness, samples);
    out2[0] = r.x; out2[1] = r.y;
}
void vqref_importance_sample_ggx(float xi0, float xi1, const float* N, float roughness, float* out3) {
    const float3 r = ImportanceSampleGGX(float2(xi0, xi1), float3(N[0], N[1], N[2]), roughness);
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
void vqref_hammersley(uint32_t i, uint32_t n, float* out2) { const float2 r = Hammersley(i, n); out2[0] = r.x; out2[1] = r.y; }
void vqref_direction_to_equirect_uv(const float* d, float* out2) { const float2 r = DirectionToEquirectUV(float3(d[0], d[1], d[2])); out2[0] = r.x; out2[1] = r.y; }
void vqref_unpack_normal(const float* s, const float* n, const float* t, float* out3) {
    const float3 r = UnpackNormal(float3(s[0], s[1], s[2]), float3(n[0], n[1], n[2]), float3(t[0], t[1], t[2]));
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}

} // extern "C"
