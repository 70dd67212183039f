// ref_fsr.cpp — runs the reference's FSR 1.0 compute shaders on the CPU: Shaders/AMDFidelityFX.hlsl (FSR_EASU_CSMain, FSR_RCAS_CSMain,
// compiled WITHOUT FSR_FP16 as PipelineStateObjects.cpp:1366-1374 does) with AMD's ffx_a.h (A_GPU + A_HLSL) and ffx_fsr1.h
// (FsrEasuF, FsrRcasF), read where they lie and rewritten syntactically by hlsl2cpp.py. Part of oracle/_ref/libvqref_shaders.so.
// TEST INFRASTRUCTURE, same construction and caveats as ref_forward.cpp. The 64-lane workgroups and their 8x8 remap (ARmp8x8) run
// exactly as dispatched: ceil(W/16) x ceil(H/16) groups, 4 pixels per lane; stores outside the image are dropped like UAV writes.
// GatherRed/Green/Blue with the clamp sampler is the fixed-function part (ref_hooks.cpp). Images are RGBA32F values.
#include <vector>

#include "ref_hooks.h"

namespace hlsl {
namespace easu {
#define FSR_EASU_CS 1
#include "AMDFidelityFX.hlsl"
#undef FSR_EASU_CS
} // namespace easu
namespace rcas {
#undef A_GPU
#undef A_HLSL
#undef FSR_EASU_F
#define FSR_RCAS_CS 1
#include "AMDFidelityFX.hlsl"
} // namespace rcas
} // namespace hlsl

using namespace hlsl;
using namespace vqref;

extern "C" {

// in: RGBA32F [inH][inW][4]; con: the 16 dwords of FsrEasuCon; out: RGB written to RGBA32F [outH][outW][4] (alpha untouched)
int vqref_fsr_easu(const float* in, int inW, int inH, const uint32_t* con, float* out, int outW, int outH) {
    if (!in || !con || !out) return -1;
    const Image src{ in, inW, inH };
    easu::FSRInputTexture.res = &src; easu::FSRInputTexture.kind = kTexImage;
    std::vector<float3> dst((size_t)outW * outH);
    easu::FSROutputTexture.data = dst.data(); easu::FSROutputTexture.width = outW; easu::FSROutputTexture.height = outH;
    easu::FSRConst0 = uint4(con[0], con[1], con[2], con[3]);   easu::FSRConst1 = uint4(con[4], con[5], con[6], con[7]);
    easu::FSRConst2 = uint4(con[8], con[9], con[10], con[11]); easu::FSRConst3 = uint4(con[12], con[13], con[14], con[15]);
    for (int gy = 0; gy < (outH + 15) / 16; ++gy)
        for (int gx = 0; gx < (outW + 15) / 16; ++gx)
            for (uint lane = 0; lane < 64; ++lane) easu::FSR_EASU_CSMain(uint3(lane, 0, 0), uint3((uint)gx, (uint)gy, 0));
    for (size_t i = 0; i < dst.size(); ++i) { out[4 * i] = dst[i].x; out[4 * i + 1] = dst[i].y; out[4 * i + 2] = dst[i].z; }
    return 0;
}

int vqref_fsr_rcas(const float* in, int W, int H, const uint32_t* con, float* out) {
    if (!in || !con || !out) return -1;
    const Image src{ in, W, H };
    rcas::RCASInputTexture.res = &src; rcas::RCASInputTexture.kind = kTexImage;
    std::vector<float3> dst((size_t)W * H);
    rcas::RCASOutputTexture.data = dst.data(); rcas::RCASOutputTexture.width = W; rcas::RCASOutputTexture.height = H;
    rcas::RCASConst0 = uint4(con[0], con[1], con[2], con[3]);
    for (int gy = 0; gy < (H + 15) / 16; ++gy)
        for (int gx = 0; gx < (W + 15) / 16; ++gx)
            for (uint lane = 0; lane < 64; ++lane) rcas::FSR_RCAS_CSMain(uint3(lane, 0, 0), uint3((uint)gx, (uint)gy, 0));
    for (size_t i = 0; i < dst.size(); ++i) { out[4 * i] = dst[i].x; out[4 * i + 1] = dst[i].y; out[4 * i + 2] = dst[i].z; }
    return 0;
}

} // extern "C"
