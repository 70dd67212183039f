// post.hip — post-process kernels for gfx950:
//   separable 21-tap Gaussian blur == Shaders/GaussianBlur.hlsl:CSMain_X :120-151 / CSMain_Y :155-187
//   tonemapper                     == Shaders/Tonemapper.hlsl:CSMain :110-151 (+ HDR.hlsl:76-80,88-97,110-119)
// Both are HBM-bound: one read + one write of the image per pass. The reference's
// "naive" blur re-reads 21 texels per output from the texture cache (PipelineStateObjects.cpp:1321);
// here the X pass stages a row segment + halo in LDS once, and the Y pass keeps a 36-row register
// window per column (16 outputs per lane) so each input row is fetched 36/16 times from L2, once from HBM.
// Accumulation order is the HLSL's: kernelIt = 0..20 i.e. offset -10..+10, acc = acc + rgb*w (no FMA).
#include "vq_internal.h"
#include "vq_devmath.h"

using namespace vqd;

namespace {

// KERNEL_WEIGHTS for KERNEL_RANGE == 11 (KERNEL_DIMENSION 21), GaussianBlur.hlsl:30-32,109-111
__device__ const float kW[11] = { 0.224716f, 0.191756f, 0.119146f, 0.053897f, 0.017746f, 0.004252f, 0.000741f, 0.000094f, 0.000009f, 0.000001f, 0.0f };
constexpr int R = 10;    // KERNEL_RANGE_MINUS1

template <int FMT>
__global__ __launch_bounds__(256) void k_blur_x(const void* __restrict__ in, void* __restrict__ out, int W, int H) {
    __shared__ float4 tile[256 + 2 * R];
    const int y = blockIdx.y;
    const int x0 = blockIdx.x * 256;
    const int t = threadIdx.x;
    const size_t row = (size_t)y * W;
    for (int i = t; i < 256 + 2 * R; i += 256) {
        int sx = x0 - R + i;
        sx = min(max(sx, 0), W - 1);                        // clamp(sampleCoord.x, 0, iImageSize.x - 1) :143
        tile[i] = load_px<FMT>(in, row + sx);
    }
    __syncthreads();
    const int x = x0 + t;
    if (x >= W) return;                                     // early out :129
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    #pragma unroll
    for (int it = 0; it < 21; ++it) {
        const int off = it - R;
        const float w = kW[off < 0 ? -off : off];
        const float4 s = tile[t + it];
        ax = ax + s.x * w; ay = ay + s.y * w; az = az + s.z * w;
    }
    store_px<FMT>(out, row + x, make_float4(ax, ay, az, 1.0f));
}

// Y pass: block = 64 columns x 4 row groups, each lane produces ROWS outputs of one column.
template <int FMT, int ROWS>
__global__ __launch_bounds__(256) void k_blur_y(const void* __restrict__ in, void* __restrict__ out,
                                                const void* __restrict__ haloTop, const void* __restrict__ haloBottom, int haloRows, int W, int H) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int yBase = (blockIdx.y * 4 + (threadIdx.x >> 6)) * ROWS;
    if (x >= W || yBase >= H) return;
    float wx[ROWS + 2 * R], wy[ROWS + 2 * R], wz[ROWS + 2 * R];
    #pragma unroll
    for (int i = 0; i < ROWS + 2 * R; ++i) {
        int sy = yBase - R + i;
        float4 s;
        if (sy < 0 && haloTop)              s = load_px<FMT>(haloTop, (size_t)(haloRows + sy) * W + x);
        else if (sy > H - 1 && haloBottom)  s = load_px<FMT>(haloBottom, (size_t)(sy - H) * W + x);
        else { sy = min(max(sy, 0), H - 1); s = load_px<FMT>(in, (size_t)sy * W + x); }      // clamp :178
        wx[i] = s.x; wy[i] = s.y; wz[i] = s.z;
    }
    #pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        if (yBase + r >= H) break;
        float ax = 0.0f, ay = 0.0f, az = 0.0f;
        #pragma unroll
        for (int it = 0; it < 21; ++it) {
            const int off = it - R;
            const float w = kW[off < 0 ? -off : off];
            ax = ax + wx[r + it] * w; ay = ay + wy[r + it] * w; az = az + wz[r + it] * w;
        }
        store_px<FMT>(out, (size_t)(yBase + r) * W + x, make_float4(ax, ay, az, 1.0f));
    }
}

// ---- tonemapper ------------------------------------------------------------------------------------
VQD float reinhard_srgb(float c, int gamma) {
    float t = div_(c, c + 1.0f);                                              // Tonemap_Reinhard, Tonemapper.hlsl:24-27
    if (gamma)                                                                // LinearToSRGB, HDR.hlsl:76-80
        t = (t < 0.0031308f) ? 12.92f * t : 1.055f * pow_(abs_(t), (float)(1.0 / 2.4)) - 0.055f;
    return t;
}
VQD float st2084(float c) {                                                   // LinearToST2084, HDR.hlsl:110-119
    const float m1 = (float)(2610.0 / 4096.0 / 4), m2 = (float)(2523.0 / 4096.0 * 128), c1 = (float)(3424.0 / 4096.0),
                c2 = (float)(2413.0 / 4096.0 * 32), c3 = (float)(2392.0 / 4096.0 * 32);
    const float cp = pow_(abs_(c), m1);
    return pow_(div_(c1 + c2 * cp, 1.0f + c3 * cp), m2);
}

template <int INFMT, int OUTFMT>
__global__ __launch_bounds__(256) void k_tonemap(const void* __restrict__ in, void* __restrict__ out, size_t n, VQ_TonemapperParams p) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 c = load_px<INFMT>(in, i);
    float ox, oy, oz;
    switch (p.OutputDisplayCurveEnum) {                                       // Tonemapper.hlsl:120-148
        case VQ_DISPLAY_CURVE_SRGB:
            ox = reinhard_srgb(c.x, p.ToggleGammaCorrection); oy = reinhard_srgb(c.y, p.ToggleGammaCorrection); oz = reinhard_srgb(c.z, p.ToggleGammaCorrection);
            break;
        case VQ_DISPLAY_CURVE_ST2084: {
            const float s = div_(p.DisplayReferenceBrightnessLevel, 10000.0f);
            float vx = c.x, vy = c.y, vz = c.z;
            if (p.ContentColorSpaceEnum == VQ_COLOR_SPACE_REC_709) {          // Rec709ToRec2020, HDR.hlsl:88-97
                vx = fma_(0.043306f, c.z, fma_(0.329292f, c.y, 0.627402f * c.x));
                vy = fma_(0.011360f, c.z, fma_(0.919544f, c.y, 0.069095f * c.x));
                vz = fma_(0.895578f, c.z, fma_(0.088028f, c.y, 0.016394f * c.x));
            }
            ox = st2084(vx * s); oy = st2084(vy * s); oz = st2084(vz * s);
        } break;
        case VQ_DISPLAY_CURVE_LINEAR: ox = c.x; oy = c.y; oz = c.z; break;
        default: ox = 1.0f; oy = 1.0f; oz = 0.0f; break;
    }
    store_px<OUTFMT>(out, i, make_float4(ox, oy, oz, c.w));                   // alpha passes through :150
}

} // namespace

namespace vqk {

hipError_t launch_blur_x(hipStream_t s, const void* in, void* out, int W, int H, int fmt) {
    dim3 grid((W + 255) / 256, H);
    if (fmt == VQHIP_FMT_RGBA32F) hipLaunchKernelGGL((k_blur_x<0>), grid, dim3(256), 0, s, in, out, W, H);
    else                          hipLaunchKernelGGL((k_blur_x<1>), grid, dim3(256), 0, s, in, out, W, H);
    return hipGetLastError();
}

hipError_t launch_blur_y(hipStream_t s, const void* in, void* out, const void* haloTop, const void* haloBottom, int haloRows, int W, int H, int fmt) {
    constexpr int ROWS = 16;
    dim3 grid((W + 63) / 64, (H + 4 * ROWS - 1) / (4 * ROWS));
    if (fmt == VQHIP_FMT_RGBA32F) hipLaunchKernelGGL((k_blur_y<0, ROWS>), grid, dim3(256), 0, s, in, out, haloTop, haloBottom, haloRows, W, H);
    else                          hipLaunchKernelGGL((k_blur_y<1, ROWS>), grid, dim3(256), 0, s, in, out, haloTop, haloBottom, haloRows, W, H);
    return hipGetLastError();
}

hipError_t launch_tonemap(hipStream_t s, const void* in, void* out, int W, int H, const VQ_TonemapperParams& p, int inFmt, int outFmt) {
    const size_t n = (size_t)W * H;
    dim3 grid((unsigned)((n + 255) / 256));
#define TM(I, O) hipLaunchKernelGGL((k_tonemap<I, O>), grid, dim3(256), 0, s, in, out, n, p)
    if (inFmt == VQHIP_FMT_RGBA32F) {
        if (outFmt == VQHIP_FMT_RGBA32F) TM(0, 0); else if (outFmt == VQHIP_FMT_RGBA16F) TM(0, 1); else TM(0, 2);
    } else {
        if (outFmt == VQHIP_FMT_RGBA32F) TM(1, 0); else if (outFmt == VQHIP_FMT_RGBA16F) TM(1, 1); else TM(1, 2);
    }
#undef TM
    return hipGetLastError();
}

} // namespace vqk
