// shade.hip — forward-PBR lighting kernel for gfx950: ForwardLighting.hlsl:PSMain :289-380 evaluated per G-buffer pixel (SURVEY.md §8a rows
// A1-A7). One lane per pixel, 256-lane workgroups (64 for frames under 4 Mpixel), float4 SoA plane loads (16 B/lane, fully coalesced); the per-pixel body is vq_shade.h.
#include "vq_shade.h"
#include "vq_mrt.h"
#include <cstdlib>

namespace {

template <bool HAS_ENV, bool HAS_CASTERS, int OUTFMT, int AR, bool MRT>
__global__ __launch_bounds__(256, VQ_SHADE_WAVES) void k_forward_lighting(vqk::ShadeArgs a) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= a.width) return;
    const size_t i = (size_t)y * a.pitch + x;
    // the G-buffer record is read once and never again: non-temporal loads keep the 531 MB stream of a 4K frame from evicting the BRDF LUT and the cubes from the XCD's L2
    typedef float v4f __attribute__((ext_vector_type(4)));
    auto ntload = [](const float4* p) { const v4f v = __builtin_nontemporal_load((const v4f*)p); return make_float4(v.x, v.y, v.z, v.w); };
    const float4 g0 = ntload(&a.gb0[i]), g1 = ntload(&a.gb1[i]), g2 = ntload(&a.gb2[i]), g3 = ntload(&a.gb3[i]);
    if (MRT) vqk::write_extra_targets(a.mrt, x, y, g2);      // SV_TARGET1 / motion vectors of the same draw (vqhip_forward_lighting_mrt): its own instantiation,
                                                             // the kernel without them is instruction for instruction what it was
    const vqk::FrameConstants* fc = a.fc;
    const float4 c = shade_pixel<HAS_ENV, HAS_CASTERS, AR>(g0, g1, g2, g3, fc);
    store_px<OUTFMT>(a.out, (size_t)y * a.outPitch + x, c);
}

template <bool E, bool C, int AR, bool MRT>
hipError_t launch_k(hipStream_t s, const vqk::ShadeArgs& a, int outFmt, dim3 grid, int wg) {
    if (outFmt == VQHIP_FMT_RGBA32F) hipLaunchKernelGGL((k_forward_lighting<E, C, 0, AR, MRT>), grid, dim3(wg), 0, s, a);
    else                             hipLaunchKernelGGL((k_forward_lighting<E, C, 1, AR, MRT>), grid, dim3(wg), 0, s, a);
    return hipGetLastError();
}
template <bool E, bool C>
hipError_t launch_fmt(hipStream_t s, const vqk::ShadeArgs& a, int outFmt, dim3 grid) {
    const int wg = grid.z; grid.z = 1;
    const bool mrt = a.mrt.albedo || a.mrt.motion;
    if (a.arithDxc)                                          // the DXC reading of dot / normalize (vqhip_set_arithmetic): its own instantiation
        return mrt ? launch_k<E, C, 1, true>(s, a, outFmt, grid, wg) : launch_k<E, C, 1, false>(s, a, outFmt, grid, wg);
    return mrt ? launch_k<E, C, 0, true>(s, a, outFmt, grid, wg) : launch_k<E, C, 0, false>(s, a, outFmt, grid, wg);
}

} // namespace

namespace vqk {
hipError_t launch_forward_lighting(hipStream_t s, const ShadeArgs& a, bool hasEnv, bool hasCasters, int outFmt, const Options& opt) {
    // Workgroup = wg consecutive pixels of one row (grid.z carries wg to launch_fmt). 256 for large frames; frames under 4 Mpixel (1080p: 32 400 waves,
    // ~4 rounds of the chip at 8 waves per SIMD) run faster with finer-grained dispatch, which shortens the fill and the tail, and a width like 1920 is no multiple of 256:
    // cfg2 0.0778 / 0.0739 / 0.0730 ms at 256 / 128 / 64 lanes, cfg3 0.919 / 0.921 / 0.924 (profiles/r4z_shade_wg.jsonl; round 3 chose 128 on r3t_shade_wg.jsonl).
    // Option "shade_wg" overrides.
    int wg = (size_t)a.width * a.height < ((size_t)4 << 20) ? 64 : 256;
    if (opt.shadeWg == 64 || opt.shadeWg == 128 || opt.shadeWg == 256) wg = opt.shadeWg;
    dim3 grid((a.width + wg - 1) / wg, a.height, wg);
    if (hasEnv) return hasCasters ? launch_fmt<true, true>(s, a, outFmt, grid) : launch_fmt<true, false>(s, a, outFmt, grid);
    return hasCasters ? launch_fmt<false, true>(s, a, outFmt, grid) : launch_fmt<false, false>(s, a, outFmt, grid);
}
} // namespace vqk
