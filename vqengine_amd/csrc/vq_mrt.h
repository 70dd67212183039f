// vq_mrt.h — the lit draw's other render targets (ForwardLighting.hlsl:PSOutput :57-68, written at :382-389): shared by k_forward_lighting (shade.hip)
// and k_forward_from_materials (gbuffer.hip). Both tests are wave-uniform (kernel arguments); with neither target bound nothing is loaded or stored.
#pragma once
#include "vq_internal.h"
#include "vq_devmath.h"

namespace vqk {

// g2 = (Surface.diffuseColor, Surface.metalness): the G-buffer plane IS SV_TARGET1's value (:383). Motion vectors :387, as written: four IEEE quotients, two differences.
VQD void write_extra_targets(const MrtArgs& m, int x, int y, float4 g2) {
    using namespace vqd;
    if (m.albedo) {
        const size_t i = (size_t)y * m.albedoPitch + x;
        if (m.albedoF32) store_px<VQHIP_FMT_RGBA32F>(m.albedo, i, g2); else store_px<VQHIP_FMT_RGBA16F>(m.albedo, i, g2);
    }
    if (m.motion) {
        const size_t j = (size_t)y * m.svPitch + x;
        const float4 c = m.svCurr[j], p = m.svPrev[j];
        const float4 mv = make_float4(fdiv_(c.x, c.w) - fdiv_(p.x, p.w), fdiv_(c.y, c.w) - fdiv_(p.y, p.w), 0.0f, 0.0f);
        const size_t i = (size_t)y * m.motionPitch + x;
        if (m.motionF32) store_px<VQHIP_FMT_RG32F>(m.motion, i, mv); else store_px<VQHIP_FMT_RG16F>(m.motion, i, mv);
    }
}

} // namespace vqk
