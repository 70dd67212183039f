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

// ---- Y blur + tonemap through the table, ROLLING window (round 5) -----------------------------------------------------------------------
// The same arithmetic and the same table as k_blur_y_tonemap_lut, another schedule. A wave owns a 64-column strip of S output rows and walks down
// it one row at a time: the 21 rows the filter reaches live in a 32-row register ring in the STORAGE format (two dwords per row: the fp16 -> fp32
// conversion is the operand conversion of v_fma_mix_f32, exact, so there is no convert instruction and the ring costs 64 VGPRs), and the 11 other
// slots of the ring are loads IN FLIGHT: at step j the wave issues the load of row j + 21 into the slot row j - 11 has just left, then runs the
// 63 mads of row j. Loads and mads of ONE wave overlap for the whole strip — the 36-row-window form loaded, waited, computed, stored, in phases that
// all 4 096 resident waves went through together — and an input row is read (S + 20) / S times instead of 36 / 16 = 2.25 times.
// Ring slot of input row (y0 - 10 + i) is i & 31; the step loop is unrolled 32-fold so that every ring index is a compile-time register.
// acc = fma((float)half(lo / hi 16 bits of h), w, acc): ONE v_fma_mix_f32 — the fp16 operand is converted exactly inside the instruction, the product and the
// sum are rounded once to fp32 like v_fma_f32 of the converted value (same MODE denormal fields as v_cvt_f32_f16 + v_fma_f32: identical bits). Written as asm
// because the compiler, left alone, converts every ring row once and keeps an fp32 ring (96 VGPRs + 3 converts per row instead of 64 + 0).
VQD float fma_mix_lo(uint32_t h, float w, float acc) { asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(h), "s"(w)); return acc; }
VQD float fma_mix_hi(uint32_t h, float w, float acc) { asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(h), "s"(w)); return acc; }
__global__ __launch_bounds__(512, 2) void k_blur_y_tonemap_roll(const void* __restrict__ in, void* __restrict__ out, const void* __restrict__ haloTop,
                                                               const void* __restrict__ haloBottom, int haloRows, int W, int H,
                                                               const void* __restrict__ table, int stripsX, int nStrips, int S) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int strip = blockIdx.x * 8 + wv;
    const bool valid = strip < nStrips;                     // wave-uniform; an idle wave still helps to fill the table and meets the barrier
    const int sy = strip / stripsX, sx = strip - sy * stripsX;
    const int x = sx * 64 + (threadIdx.x & 63);
    const int xc = min(x, W - 1);
    const int y0 = sy * S;
    const int rows = min(S, H - y0);                        // output rows of this strip (> 0 when valid)
    // wave-uniform row source of input row r: image / halo / clamp (GaussianBlur.hlsl:178); never outside the halo buffers. Written as scalar selects and
    // the step loop below kept free of branches (but the one wave-uniform exit): with control flow inside the unrolled body the compiler's wait-count
    // insertion loses track of the loads in flight at every join and drains them all (s_waitcnt vmcnt(0)) — measured: 35 us instead of the window kernel's 27.
    const int lastRow = y0 + rows - 1 + R;                  // the last input row any output of this strip reads
    // (masks instead of selects: the compiler turns a chain of wave-uniform selects back into branches)
    const uint64_t aIn = (uint64_t)in, dTop = haloTop ? (uint64_t)haloTop - aIn : 0, dBot = haloBottom ? (uint64_t)haloBottom - aIn : 0;
    const int hasTop = haloTop ? -1 : 0, hasBot = haloBottom ? -1 : 0;
    auto load_row = [&](int r) -> uint2 {
        r = min(r, lastRow);                                // the ring runs 11 rows ahead: past the end of the strip it re-reads the last row instead of branching
        const int mT = (r >> 31) & hasTop, mB = ((H - 1 - r) >> 31) & hasBot;          // all ones when the row comes from the top / bottom halo
        const int rowIn = min(max(r, 0), H - 1), rowTop = haloRows + max(r, -haloRows), rowBot = min(r - H, haloRows - 1);
        const int row = rowIn + (mT & (rowTop - rowIn)) + (mB & (rowBot - rowIn));
        const uint64_t base = aIn + ((uint64_t)(int64_t)mT & dTop) + ((uint64_t)(int64_t)mB & dBot);
        typedef uint32_t u2v __attribute__((ext_vector_type(2)));
        typedef const u2v __attribute__((address_space(1))) * gptr;               // an address built from integers is a FLAT pointer unless it says otherwise
        const u2v v = *(gptr)(base + (((uint64_t)row * (uint64_t)W + (uint64_t)xc) << 3));
        return make_uint2(v.x, v.y);
    };
    // one memory round trip before the first row: the 31 ring loads and the wave's share of the table go out together, the table is parked in LDS, barrier
    uint2 ring[32];
    if (valid) {
        #pragma unroll
        for (int i = 0; i < 31; ++i) ring[i] = load_row(y0 - R + i);
    }
    for (int i = threadIdx.x * 16; i < 65536; i += 512 * 16) *(uint4*)(lds + i) = *(const uint4*)((const unsigned char*)table + i);
    __syncthreads();
    if (!valid) return;
    // lanes beyond the image (x >= W) load column W - 1, compute the same value as lane W - 1 and store it to the same pixel: a benign duplicate instead of a predicate
    uint32_t* __restrict__ dst = (uint32_t*)out + (size_t)y0 * W + xc;
    for (int jb = 0; jb < rows; jb += 32) {
        #pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int j = jb + u;
            if (j >= rows) return;                          // wave-uniform exit
            ring[(u + 31) & 31] = load_row(y0 - R + j + 31);                  // row j + 21 into the slot of row j - 11
            float ax = 0.0f, ay = 0.0f, az = 0.0f;
            #pragma unroll
            for (int it = 0; it < 21; ++it) {               // kernelIt = 0..20: the HLSL's order
                const int off = it - R;
                const float w = kW[off < 0 ? -off : off];
                const uint2 v = ring[(u + it) & 31];
                ax = fma_mix_lo(v.x, w, ax); ay = fma_mix_hi(v.x, w, ay); az = fma_mix_lo(v.y, w, az);
            }
            const uint32_t hx = float_to_half_bits(ax), hy = float_to_half_bits(ay), hz = float_to_half_bits(az);   // == the BlurOutput store
            dst[(size_t)j * W] = (uint32_t)lds[hx] | ((uint32_t)lds[hy] << 8) | ((uint32_t)lds[hz] << 16) | (255u << 24);   // alpha 1 -> 255
        }
    }
}


} // namespace

namespace vqk {

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
    } else hipLaunchKernelGGL((k_tonemap_lut_build<1>), dim3(256), dim3(256), 0, s, table, p);
    return hipGetLastError();
}

// lutTable: NULL, or the table of (p, outFmt) built by launch_tonemap_lut_build (the context caches it per parameter set, capi.hip)
hipError_t launch_tonemap(hipStream_t s, const void* in, void* out, int W, int H, const VQ_TonemapperParams& p, int inFmt, int outFmt, const void* lutTable, const Options& opt) {
    const size_t n = (size_t)W * H;
    if (lutTable && tonemap_uses_lut(p, inFmt, outFmt, n)) {
        if (outFmt == VQHIP_FMT_RGBA8_UNORM) {
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
        if (opt.blurYForm != 1) {                             // rolling window (default); "blur_y_form" = "window" selects the 36-row-window kernel below
            int S = opt.blurYRows > 0 ? opt.blurYRows : 64;
            if (S < 12) S = 12;
            const int stripsX = (W + 63) / 64, stripsY = (H + S - 1) / S, nStrips = stripsX * stripsY;
            hipLaunchKernelGGL(k_blur_y_tonemap_roll, dim3((nStrips + 7) / 8), dim3(512), 0, s, in, out, haloTop, haloBottom, haloRows, W, H, lutTable, stripsX, nStrips, S);
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
