// post.hip — post-process kernels for gfx950:
//   separable 21-tap Gaussian blur == Shaders/GaussianBlur.hlsl:CSMain_X :120-151 / CSMain_Y :155-187
//   tonemapper                     == Shaders/Tonemapper.hlsl:CSMain :110-151 (+ HDR.hlsl:76-80,88-97,110-119)
// Both are HBM-bound: one read + one write of the image per pass. The reference's
// "naive" blur re-reads 21 texels per output from the texture cache (PipelineStateObjects.cpp:1321);
// here the X pass stages a row segment + halo in LDS once, and the Y pass keeps a 36-row register
// window per column (16 outputs per lane) so each input row is fetched 36/16 times from L2, once from HBM.
// Accumulation order is the HLSL's: kernelIt = 0..20 i.e. offset -10..+10; each `OutRGB += rgb * w` is one mad,
// acc = fma(rgb, w, acc) (arithmetic contract v2: halves the VALU work of the two HBM-bound passes).
#include <cstring>
#include <type_traits>
#include "vq_internal.h"
#include "vq_devmath.h"

using namespace vqd;

namespace {

// KERNEL_WEIGHTS for KERNEL_RANGE == 11 (KERNEL_DIMENSION 21), GaussianBlur.hlsl:30-32,109-111
__device__ const float kW[11] = { 0.224716f, 0.191756f, 0.119146f, 0.053897f, 0.017746f, 0.004252f, 0.000741f, 0.000094f, 0.000009f, 0.000001f, 0.0f };
constexpr int R = 10;    // KERNEL_RANGE_MINUS1
constexpr size_t kCompactOffset = 65536;     // compact tonemap table behind the RGBA8 byte table in the context's 128 KB table slot
#ifndef VQ_BLUR_Y_FORM_DEFAULT
#define VQ_BLUR_Y_FORM_DEFAULT "lut64"     // measured (profiles/r3a_yforms.jsonl): the compact forms are bit-identical but 8-12 us slower at 4K
#endif

// two adjacent pixels with ONE 16-byte store (RGBA16F; idx even) / two float4 stores
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int FMT> VQD void store_px2(void* base, size_t idx, float4 a, float4 b) {
    if (FMT == 1) {
        h8 v;
        v[0] = to_f16(a.x); v[1] = to_f16(a.y); v[2] = to_f16(a.z); v[3] = to_f16(a.w);
        v[4] = to_f16(b.x); v[5] = to_f16(b.y); v[6] = to_f16(b.z); v[7] = to_f16(b.w);
        *(h8*)((h4*)base + idx) = v;
    } else { store_px<FMT>(base, idx, a); store_px<FMT>(base, idx + 1, b); }
}

// X pass, 4 consecutive pixels per lane: a 256-lane workgroup covers a 1024-pixel row segment whose 1044 input pixels
// are staged once in LDS in the storage format; each lane reads a 24-pixel register window (6 LDS reads per output
// instead of 21). One padding pixel after every 4 puts lane i's window element k at 5i + k + k/4: the stride-5-pixel
// (10-dword) ds_read_b64 pattern is conflict-free within each 32-lane group.
template <int FMT>
__global__ __launch_bounds__(256) void k_blur_x4(const void* __restrict__ in, void* __restrict__ out, int W, int H) {
    constexpr int PXB = (FMT == 0) ? 16 : 8;
    constexpr int NPX = 1024 + 2 * R;
    __shared__ __attribute__((aligned(16))) unsigned char tile[(NPX + NPX / 4 + 4) * PXB];
    const int y = blockIdx.y, x0 = blockIdx.x * 1024, t = threadIdx.x;
    const size_t row = (size_t)y * W;
    for (int i = t; i < NPX; i += 256) {
        const int sx = min(max(x0 - R + i, 0), W - 1);      // clamp(sampleCoord.x, 0, iImageSize.x - 1) :143
        const int idx = i + (i >> 2);
        if (FMT == 0) ((float4*)tile)[idx] = ((const float4*)in)[row + sx];
        else          ((h4*)tile)[idx] = ((const h4*)in)[row + sx];
    }
    __syncthreads();
    const int xb = x0 + 4 * t;
    if (xb >= W) return;                                    // early out :129
    float wx[24], wy[24], wz[24];
    #pragma unroll
    for (int k = 0; k < 24; ++k) {
        const int p = 4 * t + k;
        const float4 s = load_px<FMT>(tile, (size_t)(p + (p >> 2)));
        wx[k] = s.x; wy[k] = s.y; wz[k] = s.z;
    }
    float4 res[4];
    #pragma unroll
    for (int j = 0; j < 4; ++j) {
        float ax = 0.0f, ay = 0.0f, az = 0.0f;
        #pragma unroll
        for (int it = 0; it < 21; ++it) {
            const int off = it - R;
            const float w = kW[off < 0 ? -off : off];
            ax = fma_(wx[j + it], w, ax); ay = fma_(wy[j + it], w, ay); az = fma_(wz[j + it], w, az);
        }
        res[j] = make_float4(ax, ay, az, 1.0f);
    }
    if (xb + 3 < W && ((row + xb) & 1) == 0) {               // the lane's 4 pixels are 32 contiguous bytes (RGBA16F): two 16-byte stores
        #pragma unroll
        for (int j = 0; j < 4; j += 2) store_px2<FMT>(out, row + xb + j, res[j], res[j + 1]);
    } else {
        #pragma unroll
        for (int j = 0; j < 4; ++j) if (xb + j < W) store_px<FMT>(out, row + xb + j, res[j]);
    }
}

// Same arithmetic as k_blur_x4, software-pipelined: a persistent workgroup walks over 1024-pixel row segments and issues the
// global loads of segment n+1 (5 pixels per lane, kept in registers) before it filters segment n out of LDS, so the HBM
// latency of the next tile hides behind the 252 mads of the current one instead of adding to them.
template <int FMT>
__global__ __launch_bounds__(256) void k_blur_x4p(const void* __restrict__ in, void* __restrict__ out, int W, int H, int segsPerRow, int nSeg) {
    constexpr int PXB = (FMT == 0) ? 16 : 8;
    constexpr int NPX = 1024 + 2 * R;
    using px_t = typename std::conditional<FMT == 0, float4, h4>::type;
    __shared__ __attribute__((aligned(16))) unsigned char tile[(NPX + NPX / 4 + 4) * PXB];
    const int t = threadIdx.x;
    px_t pre[5];
    auto fetch = [&](int seg) {
        const int y = seg / segsPerRow, x0 = (seg - y * segsPerRow) * 1024;
        const size_t row = (size_t)y * W;
        #pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int i = t + 256 * j;
            if (i < NPX) pre[j] = ((const px_t*)in)[row + min(max(x0 - R + i, 0), W - 1)];      // clamp(sampleCoord.x, 0, iImageSize.x - 1) :143
        }
    };
    int seg = blockIdx.x;
    if (seg < nSeg) fetch(seg);
    for (; seg < nSeg; seg += gridDim.x) {
        #pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int i = t + 256 * j;
            if (i < NPX) ((px_t*)tile)[i + (i >> 2)] = pre[j];
        }
        __syncthreads();
        if (seg + (int)gridDim.x < nSeg) fetch(seg + gridDim.x);
        const int y = seg / segsPerRow, x0 = (seg - y * segsPerRow) * 1024;
        const size_t row = (size_t)y * W;
        const int xb = x0 + 4 * t;
        if (xb < W) {                                       // early out :129
            float wx[24], wy[24], wz[24];
            #pragma unroll
            for (int k = 0; k < 24; ++k) {
                const float4 s = load_px<FMT>(tile, (size_t)(5 * t + k + (k >> 2)));     // slot of pixel 4t + k: constant offsets from one address
                wx[k] = s.x; wy[k] = s.y; wz[k] = s.z;
            }
            float4 res[4];
            #pragma unroll
            for (int j = 0; j < 4; ++j) {
                float ax = 0.0f, ay = 0.0f, az = 0.0f;
                #pragma unroll
                for (int it = 0; it < 21; ++it) {
                    const int off = it - R;
                    const float w = kW[off < 0 ? -off : off];
                    ax = fma_(wx[j + it], w, ax); ay = fma_(wy[j + it], w, ay); az = fma_(wz[j + it], w, az);
                }
                res[j] = make_float4(ax, ay, az, 1.0f);
            }
            if (xb + 3 < W && ((row + xb) & 1) == 0) {       // the lane's 4 pixels are 32 contiguous bytes (RGBA16F): two 16-byte stores instead of
                #pragma unroll                               // four 8-byte ones (-2.4 us at 4K, profiles/r2e_post_chain.md)
                for (int j = 0; j < 4; j += 2) store_px2<FMT>(out, row + xb + j, res[j], res[j + 1]);
            } else {
                #pragma unroll
                for (int j = 0; j < 4; ++j) if (xb + j < W) store_px<FMT>(out, row + xb + j, res[j]);      // early out :129
            }
        }
        __syncthreads();
    }
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
        // window rows beyond the 10 the filter reaches (tile heights that are no multiple of ROWS) feed no valid output: their index is
        // clamped into the halo buffer so that no row outside the caller's haloRows-row allocation is ever touched
        if (sy < 0 && haloTop)              s = load_px<FMT>(haloTop, (size_t)(haloRows + max(sy, -haloRows)) * W + x);
        else if (sy > H - 1 && haloBottom)  s = load_px<FMT>(haloBottom, (size_t)min(sy - H, haloRows - 1) * W + x);
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
            ax = fma_(wx[r + it], w, ax); ay = fma_(wy[r + it], w, ay); az = fma_(wz[r + it], w, az);
        }
        store_px<FMT>(out, (size_t)(yBase + r) * W + x, make_float4(ax, ay, az, 1.0f));
    }
}

// ---- tonemapper ------------------------------------------------------------------------------------
// pow_(x, y) == exp2_(y * log2_(x)) for the operands the display curves produce. When x is a positive normal number and t = y * log2(x)
// lies in [-126, 127.5) none of the special cases of log2_ / exp2_ (zero, denormal, negative, inf, NaN base; overflow, underflow, the
// 2^128 split) can fire, and the routines reduce to the two polynomials: the same operations on the same values, ~28 instead of ~75
// VALU. Anything else takes pow_ itself (a branch no lane enters in practice). The direct-arithmetic tonemapper — RGBA32F images, and
// the HDR default ST2084 on Rec.709 content, whose 3x3 matrix rules the 64 K-entry table out — spends 3 to 6 of these per pixel.
VQD float pow_pn(float x, float y) {
    const float t = y * log2_normal_bits(__float_as_uint(x), 0);
    if (__builtin_expect(!((x >= 0x1p-126f) & (x <= 3.4028234663852886e38f) & (t >= -126.0f) & (t < 127.5f)), 0)) return pow_(x, y);
    const float n = __builtin_rintf(t), f = t - n;
    float q = 1.535336188319500E-4f;
    q = fma_(q, f, 1.339887440266574E-3f);
    q = fma_(q, f, 9.618437357674640E-3f);
    q = fma_(q, f, 5.550332471162809E-2f);
    q = fma_(q, f, 2.402264791363012E-1f);
    q = fma_(q, f, 6.931472028550421E-1f);
    return fma_(q, f, 1.0f) * __uint_as_float((uint32_t)((int)n + 127) << 23);
}
VQD float reinhard_srgb(float c, int gamma) {
    float t = div_(c, c + 1.0f);                                              // Tonemap_Reinhard, Tonemapper.hlsl:24-27
    if (gamma) {                                                              // LinearToSRGB, HDR.hlsl:76-80
        const bool lin = t < 0.0031308f;                                      // the power of a lane on the linear segment is discarded: give it a base of 1
        const float pw = pow_pn(lin ? 1.0f : abs_(t), (float)(1.0 / 2.4));
        t = lin ? 12.92f * t : 1.055f * pw - 0.055f;
    }
    return t;
}
VQD float st2084(float c) {                                                   // LinearToST2084, HDR.hlsl:110-119
    const float m1 = (float)(2610.0 / 4096.0 / 4), m2 = (float)(2523.0 / 4096.0 * 128), c1 = (float)(3424.0 / 4096.0),
                c2 = (float)(2413.0 / 4096.0 * 32), c3 = (float)(2392.0 / 4096.0 * 32);
    const float a = abs_(c);
    const bool zero = a == 0.0f;                                              // black: pow_(0, m1) = exp2_(-inf) = 0
    const float pw = pow_pn(zero ? 1.0f : a, m1);
    const float cp = zero ? 0.0f : pw;
    return pow_pn(div_(c1 + c2 * cp, 1.0f + c3 * cp), m2);
}

VQD float4 tonemap_px(const float4 c, const VQ_TonemapperParams& p) {
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
    return make_float4(ox, oy, oz, c.w);                                      // alpha passes through :150
}

template <int INFMT, int OUTFMT>
__global__ __launch_bounds__(256) void k_tonemap(const void* __restrict__ in, void* __restrict__ out, size_t n, VQ_TonemapperParams p) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    store_px<OUTFMT>(out, i, tonemap_px(load_px<INFMT>(in, i), p));
}

// Fused Y blur + tonemapper: the blurred pixel is rounded to the blur format (as if stored to BlurOutput and
// re-read) and tonemapped in registers — one image write + read less than the two dispatches. Same bits.
// Tile: 64 columns x TR output rows per 256-lane workgroup; the TR+20 input rows are staged ONCE in LDS in the storage
// format (8 B/px for RGBA16F), then lane (column c, row group g) produces TR/4 consecutive outputs from a register window
// read out of LDS (wave = 64 adjacent columns of one row: conflict-free ds_read_b64/b128). Compared with the pure
// register-window Y pass this has 4x more lanes and 4x shorter serial chains, which the 3 pow() per pixel need.
template <int FMT, int OUTFMT, int TR, bool TM = true>
__global__ __launch_bounds__(256) void k_blur_y_tonemap(const void* __restrict__ in, void* __restrict__ out,
                                                        const void* __restrict__ haloTop, const void* __restrict__ haloBottom, int haloRows,
                                                        int W, int H, VQ_TonemapperParams p) {
    constexpr int ROWS = TR / 4;                              // outputs per lane
    constexpr int PXB = (FMT == 0) ? 16 : 8;
    __shared__ __attribute__((aligned(16))) unsigned char tile[(TR + 2 * R) * 64 * PXB];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + c;
    const int y0 = blockIdx.y * TR;
    const int xc = min(x, W - 1);
    for (int i = g; i < TR + 2 * R; i += 4) {                 // 4 waves stream the rows, 64 contiguous pixels each
        int sy = y0 - R + i;
        const void* src = in; size_t idx;
        if (sy < 0 && haloTop)             { src = haloTop;    idx = (size_t)(haloRows + max(sy, -haloRows)) * W + xc; }      // never outside the halo buffer
        else if (sy > H - 1 && haloBottom) { src = haloBottom; idx = (size_t)min(sy - H, haloRows - 1) * W + xc; }
        else                               { sy = min(max(sy, 0), H - 1); idx = (size_t)sy * W + xc; }            // clamp :178
        if (FMT == 0) ((float4*)tile)[i * 64 + c] = ((const float4*)src)[idx];
        else          ((h4*)tile)[i * 64 + c] = ((const h4*)src)[idx];
    }
    __syncthreads();
    const int yBase = y0 + g * ROWS;
    if (x >= W || yBase >= H) return;
    float wx[ROWS + 2 * R], wy[ROWS + 2 * R], wz[ROWS + 2 * R];
    #pragma unroll
    for (int i = 0; i < ROWS + 2 * R; ++i) {
        const float4 s = load_px<FMT>(tile, (size_t)(g * ROWS + i) * 64 + c);
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
            ax = fma_(wx[r + it], w, ax); ay = fma_(wy[r + it], w, ay); az = fma_(wz[r + it], w, az);
        }
        float4 b = make_float4(ax, ay, az, 1.0f);
        if (!TM) { store_px<OUTFMT>(out, (size_t)(yBase + r) * W + x, b); continue; }                     // plain CSMain_Y (OUTFMT == FMT)
        if (FMT == 1) b = make_float4((float)to_f16(ax), (float)to_f16(ay), (float)to_f16(az), 1.0f);      // BlurOutput is RGBA16F
        store_px<OUTFMT>(out, (size_t)(yBase + r) * W + x, tonemap_px(b, p));
    }
}

// ---- tonemapper through a table: RGBA16F has only 65536 values per channel --------------------------------------
// When the display curve does not mix channels (sRGB / LINEAR, or ST2084 on Rec.2020 content: no 3x3 matrix) the
// tonemapped STORAGE value of a channel is a pure function of its 16 input bits. k_tonemap_lut_build evaluates the
// contract arithmetic (reinhard_srgb / st2084 above, then the UNORM8 or fp16 store conversion) once for all 65536 half
// bit patterns; k_tonemap_lut keeps the table in LDS (64 KB u8 / 128 KB u16) and turns the kernel into a pure
// HBM stream (8 B in, 4-8 B out per pixel) instead of 3 x exp2(log2()) per pixel. Identical bits by construction.
VQD float tonemap_channel(float c, const VQ_TonemapperParams& p) {
    switch (p.OutputDisplayCurveEnum) {
        case VQ_DISPLAY_CURVE_SRGB:   return reinhard_srgb(c, p.ToggleGammaCorrection);
        case VQ_DISPLAY_CURVE_ST2084: return st2084(c * div_(p.DisplayReferenceBrightnessLevel, 10000.0f));   // Rec.2020 content only
        default:                      return c;                                                                 // LINEAR
    }
}
VQD float half_bits_to_float(uint32_t h) { return (float)__builtin_bit_cast(_Float16, (uint16_t)h); }
VQD uint32_t float_to_half_bits(float f) { return (uint32_t)__builtin_bit_cast(uint16_t, to_f16(f)); }

template <int OUTFMT>
__global__ __launch_bounds__(256) void k_tonemap_lut_build(void* __restrict__ table, VQ_TonemapperParams p) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;        // 65536 lanes
    const float r = tonemap_channel(half_bits_to_float(i), p);
    if (OUTFMT == 2) ((uint8_t*)table)[i] = (uint8_t)unorm8(r);
    else             ((uint16_t*)table)[i] = (uint16_t)float_to_half_bits(r);
}

template <int OUTFMT>
__global__ __launch_bounds__(1024) void k_tonemap_lut(const uint2* __restrict__ in, void* __restrict__ out, size_t n, const void* __restrict__ table) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int TBYTES = (OUTFMT == 2) ? 65536 : 131072;
    for (int i = threadIdx.x * 16; i < TBYTES; i += 1024 * 16) *(uint4*)(lds + i) = *(const uint4*)((const unsigned char*)table + i);
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (size_t)gridDim.x * 1024) {
        const uint2 v = in[i];                                // 4 halfs: x | y<<16, z | w<<16
        const uint32_t hx = v.x & 0xffffu, hy = v.x >> 16, hz = v.y & 0xffffu, hw = v.y >> 16;
        const float alpha = half_bits_to_float(hw);           // alpha passes through the same store conversion as k_tonemap
        if (OUTFMT == 2) {
            const uint8_t* t = lds;
            ((uint32_t*)out)[i] = (uint32_t)t[hx] | ((uint32_t)t[hy] << 8) | ((uint32_t)t[hz] << 16) | (unorm8(alpha) << 24);
        } else {
            const uint16_t* t = (const uint16_t*)lds;
            ((uint2*)out)[i] = make_uint2((uint32_t)t[hx] | ((uint32_t)t[hy] << 16), (uint32_t)t[hz] | (float_to_half_bits(alpha) << 16));
        }
    }
}

// ---- Y blur + tonemap through the table, one kernel (RGBA16F in, RGBA8 out) ---------------------------------------------
// The register-window Y pass (k_blur_y) with the tonemapper folded into its store: the blurred value is rounded to fp16
// exactly like the store to BlurOutput, and its 16 bits index the 64 KB table of k_tonemap_lut_build held in LDS. BlurOutput
// (8 B written + 8 B read per pixel) never exists. Persistent 512-lane workgroups (2 per CU: 128 KB of LDS, 4 waves/SIMD)
// walk over 64-column x 128-row tiles so the table is loaded 512 times, not once per tile. Identical bits to
// vqhip_gaussian_blur_y + vqhip_tonemap.
template <int ROWS>
__global__ __launch_bounds__(512) void k_blur_y_tonemap_lut(const void* __restrict__ in, void* __restrict__ out, const void* __restrict__ haloTop,
                                                            const void* __restrict__ haloBottom, int haloRows, int W, int H,
                                                            const void* __restrict__ table, int tilesX, int nTiles) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    for (int i = threadIdx.x * 16; i < 65536; i += 512 * 16) *(uint4*)(lds + i) = *(const uint4*)((const unsigned char*)table + i);
    __syncthreads();
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave index as an SGPR: the row logic of the window becomes scalar code
    for (int tile = blockIdx.x; tile < nTiles; tile += gridDim.x) {
        const int ty = tile / tilesX, tx = tile - ty * tilesX;
        const int x = tx * 64 + (threadIdx.x & 63);
        const int yBase = (ty * 8 + wv) * ROWS;
        if (x >= W || yBase >= H) continue;
        float wx[ROWS + 2 * R], wy[ROWS + 2 * R], wz[ROWS + 2 * R];
        #pragma unroll
        for (int i = 0; i < ROWS + 2 * R; ++i) {
            int sy = yBase - R + i;
            float4 s;
            const h4* rowp;                                           // wave-uniform row source: image / halo / clamp (GaussianBlur.hlsl:178); never outside the halo buffers
            if (sy < 0 && haloTop)              rowp = (const h4*)haloTop + (size_t)(haloRows + max(sy, -haloRows)) * W;
            else if (sy > H - 1 && haloBottom)  rowp = (const h4*)haloBottom + (size_t)min(sy - H, haloRows - 1) * W;
            else                                rowp = (const h4*)in + (size_t)min(max(sy, 0), H - 1) * W;
            s = load_px<1>(rowp, (size_t)x);
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
                ax = fma_(wx[r + it], w, ax); ay = fma_(wy[r + it], w, ay); az = fma_(wz[r + it], w, az);
            }
            const uint32_t hx = float_to_half_bits(ax), hy = float_to_half_bits(ay), hz = float_to_half_bits(az);   // == the BlurOutput store
            ((uint32_t*)out)[(size_t)(yBase + r) * W + x] = (uint32_t)lds[hx] | ((uint32_t)lds[hy] << 8) | ((uint32_t)lds[hz] << 16) | (255u << 24);   // alpha 1 -> 255
        }
    }
}


// ---- the same kernel without the 64 KB table: COMPACT tonemap table (8 KB) + 16-byte stores --------------------------------------------
// The RGBA8 store value of a channel is a step function of its half code: 65 536 codes take 256 values, and inside a group of 32
// consecutive codes (one 32nd of a binade) a display curve changes value at most once. k_tonemap_lut_compact turns the full table
// T[65536] (k_tonemap_lut_build: the contract arithmetic evaluated for every half) into 2 048 entries
//     entry(g) = v0 | v1 << 8 | p << 16        with   T[32 g + i] == (i >= p ? v1 : v0)   for i = 0..31
// (p = 32: constant group). A group that takes a third value gets p = 64; a pixel that meets one reads the full table in memory, so
// the result is T[code] for EVERY code and every curve — identical bits to the table kernels by construction, verified over all
// 65 536 codes by tests/test_gpu_parity.py::test_tonemap_compact_table. With 8 KB instead of 64 KB of LDS the workgroups are small
// and plentiful (one per 64 x 4*ROWS tile, like the X pass), the table fill is 16 MB instead of 33 MB per 4K frame, and occupancy is
// set by VGPRs. ST16: the wave's 4-byte pixels of four output rows go through a wave-private 1 KB LDS staging block and leave as ONE
// 16-byte store per lane (4 adjacent pixels of one row) instead of four 4-byte stores.
__global__ __launch_bounds__(256) void k_tonemap_lut_compact(const uint8_t* __restrict__ table, uint32_t* __restrict__ compact) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;        // 2 048 lanes
    const uint4 a = ((const uint4*)table)[2 * g], b = ((const uint4*)table)[2 * g + 1];
    const uint32_t w[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
    const uint32_t v0 = w[0] & 0xffu;
    uint32_t v1 = v0, p = 32;
    bool ok = true;
    for (int i = 1; i < 32; ++i) {
        const uint32_t t = (w[i >> 2] >> (8 * (i & 3))) & 0xffu;
        if (p == 32) { if (t != v0) { p = i; v1 = t; } }
        else if (t != v1) ok = false;
    }
    compact[g] = v0 | (v1 << 8) | ((ok ? p : 64u) << 16);
}
VQD uint32_t compact_lookup(const uint32_t* ctab, uint32_t h, uint32_t& flags) {
    const uint32_t e = ctab[h >> 5];
    flags |= e;
    return ((h & 31u) >= (e >> 16)) ? ((e >> 8) & 0xffu) : (e & 0xffu);
}

// Standalone tonemapper through the compact table (RGBA16F -> RGBA8): 4 pixels per lane — two 16-byte loads, one 16-byte store — and 8 KB of
// LDS per 256-lane workgroup instead of 64 KB per 1024-lane one. Alpha passes through the UNORM8 store conversion like k_tonemap.
__global__ __launch_bounds__(256) void k_tonemap_c(const uint4* __restrict__ in, uint4* __restrict__ out, size_t nQuads, size_t nPixels,
                                                   const uint32_t* __restrict__ compact, const uint8_t* __restrict__ table) {
    __shared__ __attribute__((aligned(16))) uint32_t ctab[2048];
    ((uint4*)ctab)[threadIdx.x] = ((const uint4*)compact)[threadIdx.x]; ((uint4*)ctab)[threadIdx.x + 256] = ((const uint4*)compact)[threadIdx.x + 256];
    __syncthreads();
    auto px = [&](uint32_t lo, uint32_t hi) -> uint32_t {      // lo = x | y << 16, hi = z | w << 16
        const uint32_t hx = lo & 0xffffu, hy = lo >> 16, hz = hi & 0xffffu;
        uint32_t flags = 0;
        uint32_t v = compact_lookup(ctab, hx, flags) | (compact_lookup(ctab, hy, flags) << 8) | (compact_lookup(ctab, hz, flags) << 16);
        if (__builtin_expect((flags & (64u << 16)) != 0, 0)) v = (uint32_t)table[hx] | ((uint32_t)table[hy] << 8) | ((uint32_t)table[hz] << 16);
        return v | (unorm8(half_bits_to_float(hi >> 16)) << 24);
    };
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nQuads; q += (size_t)gridDim.x * 256) {
        const uint4 a = in[2 * q], b = in[2 * q + 1];
        out[q] = make_uint4(px(a.x, a.y), px(a.z, a.w), px(b.x, b.y), px(b.z, b.w));
    }
    if (blockIdx.x == 0 && threadIdx.x < (nPixels & 3)) {      // the last 1-3 pixels of an image whose size is no multiple of 4
        const size_t i = nQuads * 4 + threadIdx.x;
        const uint2 v = ((const uint2*)in)[i];
        ((uint32_t*)out)[i] = px(v.x, v.y);
    }
}

template <int ROWS, bool ST16, int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_blur_y_tonemap_c(const void* __restrict__ in, void* __restrict__ out, const void* __restrict__ haloTop,
                                                          const void* __restrict__ haloBottom, int haloRows, int W, int H,
                                                          const uint32_t* __restrict__ compact, const uint8_t* __restrict__ table) {
    __shared__ __attribute__((aligned(16))) uint32_t ctab[2048];
    __shared__ __attribute__((aligned(16))) uint32_t stage[ST16 ? 4 * 4 * 64 : 4];
    const uint4 t0 = ((const uint4*)compact)[threadIdx.x], t1 = ((const uint4*)compact)[threadIdx.x + 256];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * 64, x = x0 + lane, xc = min(x, W - 1);
    const int yBase = (blockIdx.y * 4 + wv) * ROWS;
    float wx[ROWS + 2 * R], wy[ROWS + 2 * R], wz[ROWS + 2 * R];
    if (yBase < H) {                                          // wave-uniform; the window's loads are in flight while the table is staged
        #pragma unroll
        for (int i = 0; i < ROWS + 2 * R; ++i) {
            const int sy = yBase - R + i;
            const h4* rowp;                                   // wave-uniform row source: image / halo / clamp (GaussianBlur.hlsl:178); never outside the halo buffers
            if (sy < 0 && haloTop)              rowp = (const h4*)haloTop + (size_t)(haloRows + max(sy, -haloRows)) * W;
            else if (sy > H - 1 && haloBottom)  rowp = (const h4*)haloBottom + (size_t)min(sy - H, haloRows - 1) * W;
            else                                rowp = (const h4*)in + (size_t)min(max(sy, 0), H - 1) * W;
            const float4 s = load_px<1>(rowp, (size_t)xc);
            wx[i] = s.x; wy[i] = s.y; wz[i] = s.z;
        }
    }
    ((uint4*)ctab)[threadIdx.x] = t0; ((uint4*)ctab)[threadIdx.x + 256] = t1;
    __syncthreads();
    if (yBase >= H) return;
    const bool wide = ST16 && (x0 + 64 <= W) && ((W & 3) == 0);      // wave-uniform: the tile's rows are whole 16-byte groups
    uint32_t* myStage = stage + (ST16 ? wv * 256 : 0);
    uint32_t o[4];
    #pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        float ax = 0.0f, ay = 0.0f, az = 0.0f;
        #pragma unroll
        for (int it = 0; it < 21; ++it) {
            const int off = it - R;
            const float w = kW[off < 0 ? -off : off];
            ax = fma_(wx[r + it], w, ax); ay = fma_(wy[r + it], w, ay); az = fma_(wz[r + it], w, az);
        }
        const uint32_t hx = float_to_half_bits(ax), hy = float_to_half_bits(ay), hz = float_to_half_bits(az);   // == the BlurOutput store
        uint32_t flags = 0;
        uint32_t px = compact_lookup(ctab, hx, flags) | (compact_lookup(ctab, hy, flags) << 8) | (compact_lookup(ctab, hz, flags) << 16) | (255u << 24);   // alpha 1 -> 255
        if (__builtin_expect((flags & (64u << 16)) != 0, 0))  // a group with more than one step: the full table
            px = (uint32_t)table[hx] | ((uint32_t)table[hy] << 8) | ((uint32_t)table[hz] << 16) | (255u << 24);
        if (!ST16) { if (x < W && yBase + r < H) ((uint32_t*)out)[(size_t)(yBase + r) * W + x] = px; continue; }
        o[r & 3] = px;
        if ((r & 3) != 3) continue;
        const int rb = yBase + r - 3;                         // first of the four rows in o[]
        if (wide && rb + 3 < H) {
            #pragma unroll
            for (int k = 0; k < 4; ++k) myStage[k * 64 + lane] = o[k];
            __builtin_amdgcn_wave_barrier();                  // LDS operations of one wave execute in order: the b128 read below sees the four writes
            const uint4 v = *(const uint4*)(myStage + (lane >> 4) * 64 + (lane & 15) * 4);
            __builtin_amdgcn_wave_barrier();
            *(uint4*)((uint32_t*)out + (size_t)(rb + (lane >> 4)) * W + x0 + (lane & 15) * 4) = v;
        } else {
            #pragma unroll
            for (int k = 0; k < 4; ++k) if (x < W && rb + k < H) ((uint32_t*)out)[(size_t)(rb + k) * W + x] = o[k];
        }
    }
}

// ---- the whole post chain in ONE kernel: CSMain_X -> CSMain_Y -> Tonemapper (RGBA16F scene colour in, RGBA8 out) -----------------------
// EXPERIMENTAL, opt-in (VQHIP_POST_ONE_KERNEL=1 / 1c): bit-identical to the dispatches, 12 B/px of HBM traffic instead of 28, but SLOWER at 4K — 60 us against
// 51 us for blur X + fused blur Y/tonemap (profiles/r3g_post_one_kernel.md): ~212 instructions per pixel at 2 waves per SIMD and two barriers per step
// issue at ~53 % of the ceiling. BlurIntermediate and BlurOutput never exist in HBM.
// A 256-lane workgroup owns a strip of TW = 128 columns and a segment of rows, and walks down it RI = 8 input rows per step:
//   1. the 8 x 148 raw pixels of the step (prefetched into registers one step ahead) are converted ONCE to fp32 and staged in LDS as three planes:
//      lane (row r, t) reads its 24-pixel window as 6 aligned ds_read_b128 per channel, filters 4 consecutive outputs, rounds them to fp16
//      exactly like the store to BlurIntermediate and writes them into a 28-row ring of X-blurred rows;
//   2. lane (column c, group g) streams the 24 ring rows under its 4 output rows through 12 accumulators (each output still sums its 21 taps in the
//      HLSL's order, one mad per tap), rounds to fp16 like the store to BlurOutput and looks the 16 bits up in the tonemap table.
// Rows / columns outside the image are clamped when the raw pixels are loaded (CSMain_X :143, CSMain_Y :178), which commutes with the row-wise X
// pass. Outputs lag the input by 20 rows; two barriers per step.
//   * one workgroup per (strip, row segment), no persistence: 510 workgroups at 4K = ONE wave of workgroups at two per CU (540 cost +50 %);
//   * the ring of X-blurred rows holds the BlurIntermediate texels themselves — packed RGBA16F, 8 B per pixel: a column lane fetches its 24-row
//     window as 24 conflict-free ds_read_b64 (round 2's k_post_fused: 72 ds_read_b32 from fp32 planes, 145 KB of LDS, 8 waves per CU, 97-100 us);
//   * LDS per workgroup: 14.6 KB raw step (fp32 planes: the X pass reads 24-pixel windows as aligned ds_read_b128) + 28.7 KB ring + the tonemap
//     table: its positive half (32 KB; negative / NaN-signed codes read the table in memory) -> 75 KB, two workgroups per CU; or the compact table
//     (8 KB, LUTMODE 1) -> 51 KB, three per CU.
// Identical bits to the dispatches: every output still sums its 21 taps in the HLSL's order, one mad per tap, and both intermediate images are
// rounded to fp16 exactly where the reference stores them.
namespace pc2 {
constexpr int TW = 128, RI = 8, RING = 28, NPX = TW + 2 * R, NPXP = (NPX + 3) & ~3;      // 148 -> 152
constexpr int RAW_FLOATS = RI * 3 * NPXP, RING_PX = RING * TW;
constexpr int LUT_BYTES[2] = { 32768, 8192 };
constexpr int lds_bytes(int lutmode) { return RAW_FLOATS * 4 + RING_PX * 8 + LUT_BYTES[lutmode]; }
}
template <int LUTMODE>
__global__ __launch_bounds__(256) void k_post_chain2(const h4* __restrict__ in, uint32_t* __restrict__ out, int W, int H, const uint8_t* __restrict__ table,
                                                     int strips, int segRows) {
    using namespace pc2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* raw = (float*)lds;                               // [RI][3][NPXP]
    uint2* ring = (uint2*)(raw + RAW_FLOATS);               // [RING][TW] packed RGBA16F
    unsigned char* lut = (unsigned char*)(ring + RING_PX);
    const int tid = threadIdx.x;
    {
        const unsigned char* src = LUTMODE == 0 ? table : table + kCompactOffset;
        for (int i = tid * 16; i < LUT_BYTES[LUTMODE]; i += 256 * 16) *(uint4*)(lut + i) = *(const uint4*)(src + i);
    }
    const float w0 = 0.224716f, w1 = 0.191756f, w2 = 0.119146f, w3 = 0.053897f, w4 = 0.017746f, w5 = 0.004252f, w6 = 0.000741f, w7 = 0.000094f,
                w8 = 0.000009f, w9 = 0.000001f, w10 = 0.0f;                                      // KERNEL_WEIGHTS, GaussianBlur.hlsl:109-111
    const float wt[21] = { w10, w9, w8, w7, w6, w5, w4, w3, w2, w1, w0, w1, w2, w3, w4, w5, w6, w7, w8, w9, w10 };   // offset -10 .. +10
    const int seg = blockIdx.x / strips, strip = blockIdx.x - seg * strips;
    const int x0 = strip * TW, y0 = seg * segRows;
    const int rows = min(segRows, H - y0);
    const int nIter = (rows + 2 * R + RI - 1) / RI;
    h4 pre[5];
    auto fetch = [&](int it) {                              // raw rows j = it*RI .. +7 of the segment: image row y0 - 10 + j, columns x0 - 10 + p (clamped :143,:178)
        #pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int e = tid + 256 * k;
            if (e < RI * NPX) {
                const int r = e / NPX, p = e - r * NPX;
                const int y = min(max(y0 - R + it * RI + r, 0), H - 1), x = min(max(x0 - R + p, 0), W - 1);
                pre[k] = in[(size_t)y * W + x];
            }
        }
    };
    fetch(0);
    for (int it = 0; it < nIter; ++it) {
        // 1a. stage the prefetched raw pixels as fp32 planes
        #pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int e = tid + 256 * k;
            if (e < RI * NPX) {
                const int r = e / NPX, p = e - r * NPX;
                float* dst = raw + (r * 3) * NPXP + p;
                dst[0] = (float)pre[k].x; dst[NPXP] = (float)pre[k].y; dst[2 * NPXP] = (float)pre[k].z;
            }
        }
        __syncthreads();                                    // A: raw visible; the Y pass of the previous step is done with the ring rows X now overwrites
        if (it + 1 < nIter) fetch(it + 1);
        // 1b. X pass: lane (r, t) -> outputs 4t .. 4t+3 of raw row r -> ring row (it*RI + r) % RING as packed RGBA16F (== the store to BlurIntermediate)
        {
            const int r = tid >> 5, t = tid & 31;
            uint32_t hx[4], hy[4], hz[4];
            #pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float4* src = (const float4*)(raw + (r * 3 + ch) * NPXP) + t;
                float v[24];
                #pragma unroll
                for (int g = 0; g < 6; ++g) { const float4 q = src[g]; v[4 * g] = q.x; v[4 * g + 1] = q.y; v[4 * g + 2] = q.z; v[4 * g + 3] = q.w; }
                #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float acc = 0.0f;
                    #pragma unroll
                    for (int k = 0; k < 21; ++k) acc = fma_(v[j + k], wt[k], acc);
                    const uint32_t h = float_to_half_bits(acc);
                    if (ch == 0) hx[j] = h; else if (ch == 1) hy[j] = h; else hz[j] = h;
                }
            }
            const int slot = (it * RI + r) % RING;
            uint4* dst = (uint4*)(ring + slot * TW + 4 * t);
            dst[0] = make_uint4(hx[0] | (hy[0] << 16), hz[0] | 0x3c000000u, hx[1] | (hy[1] << 16), hz[1] | 0x3c000000u);      // alpha := 1
            dst[1] = make_uint4(hx[2] | (hy[2] << 16), hz[2] | 0x3c000000u, hx[3] | (hy[3] << 16), hz[3] | 0x3c000000u);
        }
        __syncthreads();                                    // B: the new ring rows are visible
        // 2. Y pass + tonemap: lane (c, g) -> output rows ob .. ob+3 of column c, ob = it*RI - 20 + 4g
        {
            const int c = tid & (TW - 1), g = tid >> 7;
            const int ob = it * RI - 2 * R + 4 * g;
            const int x = x0 + c;
            if (ob + 3 >= 0 && ob < rows && x < W) {
                float ax[4] = { 0, 0, 0, 0 }, ay[4] = { 0, 0, 0, 0 }, az[4] = { 0, 0, 0, 0 };
                int slot = (ob + RING * 4) % RING;          // ring row of X-blurred row j = ob (ob >= -20)
                #pragma unroll
                for (int i = 0; i < 24; ++i) {
                    const uint2 q = ring[slot * TW + c];
                    const float vx = half_bits_to_float(q.x & 0xffffu), vy = half_bits_to_float(q.x >> 16), vz = half_bits_to_float(q.y & 0xffffu);
                    #pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        const int k = i - o;                // tap index of output o for window row i
                        if (k >= 0 && k <= 20) { ax[o] = fma_(vx, wt[k], ax[o]); ay[o] = fma_(vy, wt[k], ay[o]); az[o] = fma_(vz, wt[k], az[o]); }
                    }
                    slot = slot + 1 == RING ? 0 : slot + 1;
                }
                #pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const int oy = ob + o;
                    if (oy < 0 || oy >= rows) continue;
                    const uint32_t bx = float_to_half_bits(ax[o]), by = float_to_half_bits(ay[o]), bz = float_to_half_bits(az[o]);   // == the BlurOutput store
                    uint32_t px;
                    if (LUTMODE == 0) {
                        const uint32_t tx = bx < 0x8000u ? lut[bx] : table[bx], ty = by < 0x8000u ? lut[by] : table[by], tz = bz < 0x8000u ? lut[bz] : table[bz];
                        px = tx | (ty << 8) | (tz << 16) | (255u << 24);
                    } else {
                        uint32_t flags = 0;
                        px = compact_lookup((const uint32_t*)lut, bx, flags) | (compact_lookup((const uint32_t*)lut, by, flags) << 8) |
                             (compact_lookup((const uint32_t*)lut, bz, flags) << 16) | (255u << 24);
                        if (__builtin_expect((flags & (64u << 16)) != 0, 0)) px = (uint32_t)table[bx] | ((uint32_t)table[by] << 8) | ((uint32_t)table[bz] << 16) | (255u << 24);
                    }
                    out[(size_t)(y0 + oy) * W + x] = px;     // alpha 1 -> 255
                }
            }
        }
    }
}

} // namespace

namespace vqk {

// CSMain_X + CSMain_Y + Tonemapper as ONE kernel (k_post_chain2) when the table path applies; `table` = the tonemap table of (p, RGBA8).
bool post_chain_fusable(const VQ_TonemapperParams& p, int inFmt, int outFmt, int W, int H) {
    return blur_y_tonemap_uses_lut(p, inFmt, outFmt, (size_t)W * H) && W >= 64 && H >= 32;
}
hipError_t launch_post_chain2(hipStream_t s, const void* in, void* out, int W, int H, const void* table, bool compactLut, const Options& opt) {
    const int strips = (W + pc2::TW - 1) / pc2::TW;
    int nseg = 512 / strips;                                 // at most two workgroups per CU in ONE wave of workgroups (a second, partial wave costs +50 %)
    if (nseg > (H + 31) / 32) nseg = (H + 31) / 32;          // segments of at least 32 rows: each re-reads 20 halo rows
    if (nseg < 1) nseg = 1;
    if (opt.postSegments > 0) nseg = opt.postSegments;
    const int segRows = (H + nseg - 1) / nseg;
    nseg = (H + segRows - 1) / segRows;
    const int ldsBytes = pc2::lds_bytes(compactLut ? 1 : 0);
    hipError_t e = compactLut ? hipFuncSetAttribute((const void*)k_post_chain2<1>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes)
                              : hipFuncSetAttribute((const void*)k_post_chain2<0>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes);
    if (e != hipSuccess) return e;
    if (compactLut) hipLaunchKernelGGL((k_post_chain2<1>), dim3(strips * nseg), dim3(256), ldsBytes, s, (const h4*)in, (uint32_t*)out, W, H, (const uint8_t*)table, strips, segRows);
    else            hipLaunchKernelGGL((k_post_chain2<0>), dim3(strips * nseg), dim3(256), ldsBytes, s, (const h4*)in, (uint32_t*)out, W, H, (const uint8_t*)table, strips, segRows);
    return hipGetLastError();
}
// Which form of the X pass runs is chosen for the FRAME, not for the kernel alone (profiles/r2k_frame_loop.md): the software-pipelined persistent
// form is the fastest kernel in isolation (25.9 us at 4K with 1 024 workgroups, 27.3 with 2 048), but with many workgroups in flight on real image
// data it makes the chip throttle, and the shade kernel that follows it runs 2-13 % slower. Option "blur_x_wgs" overrides the default for tuning:
// 0 = one workgroup per 1024-pixel segment (k_blur_x4), n > 0 = n persistent workgroups (k_blur_x4p).
hipError_t launch_blur_x(hipStream_t s, const void* in, void* out, int W, int H, int fmt, const Options& opt) {
    const int segsPerRow = (W + 1023) / 1024, nSeg = segsPerRow * H;
    int want = opt.blurXWgs;
    if (want <= 0 && H > 65535) want = 1024;               // grid.y is limited to 65 535: taller images take the persistent form
    if (want > 0) {
        const int wgs = nSeg < want ? nSeg : want;
        if (fmt == VQHIP_FMT_RGBA32F) hipLaunchKernelGGL((k_blur_x4p<0>), dim3(wgs), dim3(256), 0, s, in, out, W, H, segsPerRow, nSeg);
        else                          hipLaunchKernelGGL((k_blur_x4p<1>), dim3(wgs), dim3(256), 0, s, in, out, W, H, segsPerRow, nSeg);
    } else {
        dim3 grid(segsPerRow, H);
        if (fmt == VQHIP_FMT_RGBA32F) hipLaunchKernelGGL((k_blur_x4<0>), grid, dim3(256), 0, s, in, out, W, H);
        else                          hipLaunchKernelGGL((k_blur_x4<1>), grid, dim3(256), 0, s, in, out, W, H);
    }
    return hipGetLastError();
}

hipError_t launch_blur_y(hipStream_t s, const void* in, void* out, const void* haloTop, const void* haloBottom, int haloRows, int W, int H, int fmt) {
#ifndef VQ_BLUR_Y_TR
#define VQ_BLUR_Y_TR 0      // A/B at 4K RGBA16F (scripts/bench_variants.sh): register window 31 us, LDS tile TR=16/32/64: 41/36/49 us
#endif
#if VQ_BLUR_Y_TR
    constexpr int TR = VQ_BLUR_Y_TR;                          // LDS-tiled: (TR+20) rows x 64 px staged once, TR/4 outputs per lane
    dim3 grid((W + 63) / 64, (H + TR - 1) / TR);
    const VQ_TonemapperParams none = {};
    if (fmt == VQHIP_FMT_RGBA32F) hipLaunchKernelGGL((k_blur_y_tonemap<0, 0, TR, false>), grid, dim3(256), 0, s, in, out, haloTop, haloBottom, haloRows, W, H, none);
    else                          hipLaunchKernelGGL((k_blur_y_tonemap<1, 1, TR, false>), grid, dim3(256), 0, s, in, out, haloTop, haloBottom, haloRows, W, H, none);
#else
    constexpr int ROWS = 16;                                  // register-window variant
    dim3 grid((W + 63) / 64, (H + 4 * ROWS - 1) / (4 * ROWS));
    if (fmt == VQHIP_FMT_RGBA32F) hipLaunchKernelGGL((k_blur_y<0, ROWS>), grid, dim3(256), 0, s, in, out, haloTop, haloBottom, haloRows, W, H);
    else                          hipLaunchKernelGGL((k_blur_y<1, ROWS>), grid, dim3(256), 0, s, in, out, haloTop, haloBottom, haloRows, W, H);
#endif
    return hipGetLastError();
}

static bool perChannelCurve(const VQ_TonemapperParams& p) {
    return p.OutputDisplayCurveEnum == VQ_DISPLAY_CURVE_SRGB || p.OutputDisplayCurveEnum == VQ_DISPLAY_CURVE_LINEAR ||
           (p.OutputDisplayCurveEnum == VQ_DISPLAY_CURVE_ST2084 && p.ContentColorSpaceEnum != VQ_COLOR_SPACE_REC_709);
}
// the table path applies: RGBA16F in, a curve that does not mix channels, an 8-bit or fp16 target and enough pixels to pay for the table load
bool tonemap_uses_lut(const VQ_TonemapperParams& p, int inFmt, int outFmt, size_t nPixels) {
    return perChannelCurve(p) && inFmt == VQHIP_FMT_RGBA16F && (outFmt == VQHIP_FMT_RGBA8_UNORM || outFmt == VQHIP_FMT_RGBA16F) && nPixels >= (size_t)1 << 16;
}
bool blur_y_tonemap_uses_lut(const VQ_TonemapperParams& p, int blurFmt, int outFmt, size_t nPixels) {
    return perChannelCurve(p) && blurFmt == VQHIP_FMT_RGBA16F && outFmt == VQHIP_FMT_RGBA8_UNORM && nPixels >= (size_t)1 << 16;
}
hipError_t launch_tonemap_lut_build(hipStream_t s, void* table, const VQ_TonemapperParams& p, int outFmt) {
    if (outFmt == VQHIP_FMT_RGBA8_UNORM) {
        hipLaunchKernelGGL((k_tonemap_lut_build<2>), dim3(256), dim3(256), 0, s, table, p);
        // the compact form of the byte table (2 048 entries) sits behind it in the same 128 KB slot
        hipLaunchKernelGGL(k_tonemap_lut_compact, dim3(8), dim3(256), 0, s, (const uint8_t*)table, (uint32_t*)((unsigned char*)table + kCompactOffset));
    } else hipLaunchKernelGGL((k_tonemap_lut_build<1>), dim3(256), dim3(256), 0, s, table, p);
    return hipGetLastError();
}

// lutTable: NULL, or the table of (p, outFmt) built by launch_tonemap_lut_build (the context caches it per parameter set, capi.hip)
hipError_t launch_tonemap(hipStream_t s, const void* in, void* out, int W, int H, const VQ_TonemapperParams& p, int inFmt, int outFmt, const void* lutTable, const Options& opt) {
    const size_t n = (size_t)W * H;
    if (lutTable && tonemap_uses_lut(p, inFmt, outFmt, n)) {
        if (outFmt == VQHIP_FMT_RGBA8_UNORM) {
            // option "tonemap_form" = "compact": k_tonemap_c; default: the 64 KB-table kernel (19.4 vs 22.1 us at 4K)
            if (opt.tonemapCompact && (((uintptr_t)in | (uintptr_t)out) & 15) == 0) {
                const size_t nQuads = n / 4;
                const size_t wgs = (nQuads + 255) / 256;
                hipLaunchKernelGGL(k_tonemap_c, dim3((unsigned)(wgs < 4096 ? (wgs ? wgs : 1) : 4096)), dim3(256), 0, s, (const uint4*)in, (uint4*)out, nQuads, n,
                                   (const uint32_t*)((const unsigned char*)lutTable + kCompactOffset), (const uint8_t*)lutTable);
            } else
                hipLaunchKernelGGL((k_tonemap_lut<2>), dim3(512), dim3(1024), 65536, s, (const uint2*)in, out, n, lutTable);
        } else {
            // > 64 KB of dynamic LDS needs the opt-in; it is a per-device function attribute and cheap, so it is simply set on every launch
            hipError_t e = hipFuncSetAttribute((const void*)k_tonemap_lut<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((k_tonemap_lut<1>), dim3(256), dim3(1024), 131072, s, (const uint2*)in, out, n, lutTable);
        }
        return hipGetLastError();
    }
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

hipError_t launch_blur_y_tonemap(hipStream_t s, const void* in, void* out, const void* haloTop, const void* haloBottom, int haloRows, int W, int H,
                                 const VQ_TonemapperParams& p, int fmt, int outFmt, const void* lutTable, const Options& opt) {
    if (lutTable && blur_y_tonemap_uses_lut(p, fmt, outFmt, (size_t)W * H)) {
        // Form of the kernel (all bit-identical): "cN" / "cNs" = compact table, N output rows per lane (8, 12, 16), s = 16-byte stores through
        // the LDS staging block; "lut64" = the round-2 kernel with the 64 KB table. Option "blur_y_form" overrides the default for tuning.
        const char* form = opt.blurYForm;
        if (!*form) form = VQ_BLUR_Y_FORM_DEFAULT;
        if (form[0] == 'c') {
            const int rows = std::atoi(form + 1);
            const bool st16 = std::strchr(form, 's') != nullptr && ((uintptr_t)out & 15) == 0;
            const uint32_t* compact = (const uint32_t*)((const unsigned char*)lutTable + kCompactOffset);
            const int tilesXc = (W + 63) / 64;
#define BYC(N, S, WV) hipLaunchKernelGGL((k_blur_y_tonemap_c<N, S, WV>), dim3(tilesXc, (H + 4 * N - 1) / (4 * N)), dim3(256), 0, s, in, out, haloTop, haloBottom, haloRows, W, H, \
                                     compact, (const uint8_t*)lutTable)
            const char* wq = std::strchr(form, 'w');          // "c8sw6": at least 6 waves per SIMD (register budget 80)
            const int wv = wq ? std::atoi(wq + 1) : 0;
            if (rows == 8)       { if (wv >= 6) { if (st16) BYC(8, true, 6); else BYC(8, false, 6); } else if (wv == 5) { if (st16) BYC(8, true, 5); else BYC(8, false, 5); } else { if (st16) BYC(8, true, 4); else BYC(8, false, 4); } }
            else if (rows == 12) { if (wv >= 5) { if (st16) BYC(12, true, 5); else BYC(12, false, 5); } else { if (st16) BYC(12, true, 4); else BYC(12, false, 4); } }
            else                 { if (wv >= 5) { if (st16) BYC(16, true, 5); else BYC(16, false, 5); } else { if (st16) BYC(16, true, 4); else BYC(16, false, 4); } }
#undef BYC
            return hipGetLastError();
        }
        const int tilesX = (W + 63) / 64, tilesY = (H + 127) / 128, nTiles = tilesX * tilesY;
        int wgs = 512;                                        // two 64 KB tables per CU
        if (opt.blurYWgs > 0) wgs = opt.blurYWgs;             // tuning knob, like "blur_x_wgs"
        hipLaunchKernelGGL((k_blur_y_tonemap_lut<16>), dim3(nTiles < wgs ? nTiles : wgs), dim3(512), 0, s, in, out, haloTop, haloBottom, haloRows, W, H,
                           lutTable, tilesX, nTiles);
        return hipGetLastError();
    }
#ifndef VQ_FUSED_TR
#define VQ_FUSED_TR 16
#endif
    constexpr int TR = VQ_FUSED_TR;                           // output rows per workgroup (TR/4 per lane); LDS (TR+20) rows x 64 px
    dim3 grid((W + 63) / 64, (H + TR - 1) / TR);
#define BT(F, O) hipLaunchKernelGGL((k_blur_y_tonemap<F, O, TR>), grid, dim3(256), 0, s, in, out, haloTop, haloBottom, haloRows, W, H, p)
    if (fmt == VQHIP_FMT_RGBA32F) {
        if (outFmt == VQHIP_FMT_RGBA32F) BT(0, 0); else if (outFmt == VQHIP_FMT_RGBA16F) BT(0, 1); else BT(0, 2);
    } else {
        if (outFmt == VQHIP_FMT_RGBA32F) BT(1, 0); else if (outFmt == VQHIP_FMT_RGBA16F) BT(1, 1); else BT(1, 2);
    }
#undef BT
    return hipGetLastError();
}

} // namespace vqk
