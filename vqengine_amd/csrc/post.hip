// post.hip — post-process kernels for gfx950:
//   separable 21-tap Gaussian blur == Shaders/GaussianBlur.hlsl:CSMain_X :120-151 / CSMain_Y :155-187
//   tonemapper                     == Shaders/Tonemapper.hlsl:CSMain :110-151 (+ HDR.hlsl:76-80,88-97,110-119)
// SURVEY.md §8d expected both to be HBM-bound (one read + one write of the image per pass); on gfx950 they are bound by VALU issue (63 mads per pixel and
// pass on fp16 operands: profiles/r5g_post_forms.md). The reference's "naive" blur re-reads 21 texels per output from the texture cache
// (PipelineStateObjects.cpp:1321); here the X pass stages a row segment + halo in LDS once, the Y pass keeps a 36-row register window per column
// (16 outputs per lane) so each input row is fetched 36/16 times from L2, once from HBM, and k_post_chain (round 5) runs both passes and the tonemapper in
// one kernel over an LDS ring of X-blurred rows: 8 B read + 4 B written per pixel.
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
typedef float v2f __attribute__((ext_vector_type(2)));
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
__global__ __launch_bounds__(256) void k_blur_x4(const void* __restrict__ in, void* __restrict__ out, int W, int H, int segsPerRow) {
    constexpr int PXB = (FMT == 0) ? 16 : 8;
    constexpr int NPX = 1024 + 2 * R;
    __shared__ __attribute__((aligned(16))) unsigned char tile[(NPX + NPX / 4 + 4) * PXB];
    const int y = blockIdx.x / segsPerRow, x0 = (blockIdx.x - y * segsPerRow) * 1024, t = threadIdx.x;      // one workgroup per (row, segment): a 1-D grid, so the height is not limited by grid.y
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
    v2f wxy[24]; float wz[24];                               // (x, y) as a register pair: one v_pk_fma_f32 per tap for the two channels (round 6, as in k_post_chain)
    #pragma unroll
    for (int k = 0; k < 24; ++k) {
        const int p = 4 * t + k;
        const float4 s = load_px<FMT>(tile, (size_t)(p + (p >> 2)));
        wxy[k] = v2f{ s.x, s.y }; wz[k] = s.z;
        // the converted texel as three fp32 registers: left alone, the compiler folds each conversion into the four mads that use it (v_fma_mix_f32, 252 per lane), and
        // v_fma_mix_f32 issues no faster than v_cvt_f32_f16 on gfx950 while v_fmac_f32 with a literal weight issues faster (scripts/ubench/mix_rate.hip): 72 conversions +
        // 252 v_fmac_f32 — 31.6 against 33.1 us at 4K on one box (profiles/r5g_post_forms.md); round 6: (x, y) as a register pair, 84 v_pk_fma_f32 + 84 v_fmac_f32: 29.6 us
        if (FMT == 1) asm("" : "+v"(wxy[k]), "+v"(wz[k]));
    }
    float4 res[4];
    #pragma unroll
    for (int j = 0; j < 4; ++j) {
        v2f a = { 0.0f, 0.0f }; float az = 0.0f;
        #pragma unroll
        for (int it = 0; it < 21; ++it) {
            const int off = it - R;
            const float w = kW[off < 0 ? -off : off];
            a = __builtin_elementwise_fma(wxy[j + it], v2f{ w, w }, a); az = fma_(wz[j + it], w, az);
        }
        res[j] = make_float4(a.x, a.y, az, 1.0f);
    }
    if (xb + 3 < W && ((row + xb) & 1) == 0) {               // the lane's 4 pixels are 32 contiguous bytes (RGBA16F): two 16-byte stores
        #pragma unroll
        for (int j = 0; j < 4; j += 2) store_px2<FMT>(out, row + xb + j, res[j], res[j + 1]);
    } else {
        #pragma unroll
        for (int j = 0; j < 4; ++j) if (xb + j < W) store_px<FMT>(out, row + xb + j, res[j]);
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
template <int FMT, int OUTFMT, int TR>
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
        float wx[ROWS + 2 * R], wy[ROWS + 2 * R], wz[ROWS + 2 * R];     // (the (x, y)-pair / v_pk_fma_f32 form of k_blur_x4 costs this kernel its occupancy: 115 VGPRs, 41.3 against 27.0 us)
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

// ---- the whole chain in ONE kernel (round 5): CSMain_X -> CSMain_Y -> tonemapper, RGBA16F scene colour in, RGBA8 out ---------------------------------
// BlurIntermediate never exists either: 8 B read + 4 B written per pixel instead of 28. The same arithmetic as k_blur_x4 -> k_blur_y_tonemap_lut (each
// X-blurred texel is rounded to fp16 exactly like the store to BlurIntermediate, each Y-blurred one like the store to BlurOutput, whose 16 bits index the
// 64 KB tonemap table), another schedule:
//   * one 1 024-lane workgroup per CU owns a 64-column strip of S output rows and walks down it 32 rows per iteration. LDS = the tonemap table (64 KB) + a
//     ring of 84 X-blurred rows of 64 texels — since round 6 as the fp32 VALUES of the stored halfs, (x, y) pairs of 8 B in the padded slots + a z plane of 4 B
//     (162 640 of the CU's 163 840 bytes): the Y waves filter what they read without converting it (18 conversions per pixel less; 6 more in the X waves);
//   * waves 0-7 are the X waves, waves 8-15 the Y waves (two of each per SIMD), and they work on DIFFERENT iterations: while the X waves filter the 32 input
//     rows of iteration i into the ring, the Y waves filter the 32 output rows whose windows iteration i - 1 completed. The two halves meet at ONE barrier per
//     iteration; in between, the LDS phases (window reads) of one half run under the mad phases of the other — with all 16 waves in the same stage
//     (first form of this kernel: profiles/r5d_post_forms.md) the LDS and the VALU took turns and the kernel was no faster than the two it replaces;
//   * X waves: a quarter-wave filters one 64-texel row, 4 adjacent texels per lane. The 84 input texels are parked raw in the row's ring slot (slot of texel
//     p = p + p/4: the stride-5 ds_read_b64 pattern of k_blur_x4; the row stride of 112 slots puts the second row of a 32-lane group on the other half of the
//     banks), every lane reads its 24-texel window (6 LDS reads per output), converts it, runs the 252 mads and writes its 4 results (rounded to fp16, kept as fp32)
//     over the raw texels — the LDS operations of one wave complete in order. The global loads of iteration i + 1 are issued before the mads of iteration i;
//   * Y waves: 64 columns x 4 rows per wave; a lane reads its 24-row column window out of the ring (an 8-byte and a 4-byte read per row; six group bases per plane,
//     the rows of a group at immediate offsets: the ring's wrap never falls inside a group of 4), 252 mads, three table lookups and one 4-byte store per output.
//     In iteration 0 they have no rows yet and fill the table.
// The mads: Window4 below (round 6: converted window, v_pk_fma_f32 on (x, y) + v_fmac_f32 on z; round 5: 252 v_fma_mix_f32 on the packed halfs).
// Halos (row-tiled frames): halo_top / halo_bottom are SCENE-COLOUR rows here — the X pass is purely horizontal, so the rows the neighbour tile shaded are filtered
// in X like the tile's own and the Y window reaches them: 10 rows per side, the same byte count as the X-blurred halos of the two-kernel path.
// An 8-byte LDS read that stays ONE ds_read_b64 (2 LDS cycles per wave): left alone, the compiler merges two reads at nearby offsets of one base into ds_read2_b64
// (8 cycles). Round 6: an ordinary load from an address-space-3 pointer whose integer BASE is laundered in place through an empty asm before every load — the compiler
// cannot prove two such loads related, so it does not merge them, keeps the row / slot offsets in the instruction's immediate field (no address arithmetic per read: 158
// v_add_u32 per window pair before), and tracks their completion itself (s_waitcnt lgkmcnt(n) as the data is needed). Round 5 issued the reads from inline asm whose
// results the compiler believed ready at once and waited in a separate asm: correct only as long as the register allocator moved none of them in between.
typedef uint32_t u2v __attribute__((ext_vector_type(2)));
#define LDS_AS __attribute__((address_space(3)))
// `a` is a byte address in the LDS aperture (lds_addr() of a __shared__ pointer + offsets); it is laundered IN PLACE, so consecutive reads from one base register cost no copy
template <class T, int OFF> VQD T lds_load(uint32_t& a) {
    asm volatile("" : "+v"(a));
    return *(const LDS_AS T*)(uintptr_t)(a + OFF);
}
VQD uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const LDS_AS void*)p; }
template <int T> struct XWindow {                             // window texel t of the X stage: slot 5 li + t + t / 4
    static VQD void read(uint32_t (&lo)[24], uint32_t (&hi)[24], uint32_t& base) {
        XWindow<T - 1>::read(lo, hi, base);
        const u2v v = lds_load<u2v, (T + (T >> 2)) * 8>(base);
        lo[T] = v.x; hi[T] = v.y;
    }
};
template <> struct XWindow<-1> { static VQD void read(uint32_t (&)[24], uint32_t (&)[24], uint32_t&) {} };
// The 63 mads of each of a lane's 4 outputs: kernelIt = 0..20, the HLSL's order (GaussianBlur.hlsl:138-150 / :173-185). Round 6: the 24-texel window is converted ONCE
// (72 v_cvt_f32_f16), then every tap is ONE v_pk_fma_f32 on the (x, y) register pair (weights in SGPR pairs: VOP3P takes no literal; VGPR weight pairs cost 12 registers and
// spilled) and ONE v_fmac_f32 with a literal weight on z — 72 + 84 + 84 instructions where round 5 issued 252 v_fma_mix_f32 on the packed halfs (fp16 -> fp32 inside the
// instruction): 60.4 -> 52.6 us at 4K on one box, identical bits (profiles/r6h_post_chain_forms.md). The same IEEE operations in the same order either way.
VQD float half_lo(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(u & 0xffffu)); }
VQD float half_hi(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(u >> 16)); }
struct Window4 {
    uint32_t wl[24], wh[24];                                  // as read: x | y << 16, z | a << 16
    v2f xy[24]; float z[24];
    VQD void convert() {
        #pragma unroll
        for (int t = 0; t < 24; ++t) {
            xy[t] = v2f{ half_lo(wl[t]), half_hi(wl[t]) }; z[t] = half_lo(wh[t]);
            asm("" : "+v"(xy[t]), "+v"(z[t]));                // keep the conversion apart: left alone the compiler folds it back into v_fma_mix_f32
        }
    }
    VQD void filter(int j, float& ax, float& ay, float& az) const {
        v2f a = { 0.0f, 0.0f };
        az = 0.0f;
        #pragma unroll
        for (int it = 0; it < 21; ++it) {
            const int off = it - R;
            const float wt = kW[off < 0 ? -off : off];
            a = __builtin_elementwise_fma(xy[j + it], v2f{ wt, wt }, a);
            az = fma_(z[j + it], wt, az);
        }
        ax = a.x; ay = a.y;
    }
};
constexpr int PC_C = 64;                                      // columns of a strip
constexpr int PC_RS = 32;                                     // rows per iteration
constexpr int PC_RING = 2 * PC_RS + 2 * R;                    // 84 rows: the 32 the X waves write + the 52 the Y waves read
constexpr int PC_NPX = PC_C + 2 * R;                          // 84 input texels per row
constexpr int PC_ROWB = 112 * 8;                              // bytes per ring row: 84 texels + one pad after every 4 = 105 slots, rounded up to 16 mod 32 slots
constexpr int PC_TABLE = 65536;
constexpr int PC_ZROWB = 65 * 4;                              // bytes per row of the ring's z plane: 64 floats + 1 (the four rows a wave writes at once land on different banks)
constexpr int PC_ZBASE = PC_TABLE + PC_RING * PC_ROWB;
constexpr int PC_LDS = PC_ZBASE + PC_RING * PC_ZROWB;         // 162 640 of the CU's 163 840 bytes
static_assert(PC_NPX + PC_NPX / 4 <= PC_ROWB / 8 && (PC_ROWB / 8) % 32 == 16, "ring row");
static_assert(PC_LDS <= 160 * 1024, "LDS of one CU");
__global__ __launch_bounds__(1024) void k_post_chain(const void* __restrict__ in, void* __restrict__ out, const void* __restrict__ haloTop,
                                                     const void* __restrict__ haloBottom, int haloRows, int W, int H,
                                                     const void* __restrict__ table, int stripsX, int stripsY, int S, int xcdBands) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t ldsA = lds_addr(lds);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // strip (sx, sy) of workgroup b. Workgroup b runs on XCD b % 8 and every XCD has its own L2: with xcdBands = 8 / stripsY > 0 an XCD owns ONE band of rows
    // and a contiguous range of its column strips, so that the 20 halo columns two neighbouring strips share are read through one L2.
    int sx, sy;
    if (xcdBands > 0) {
        const int xcd = blockIdx.x & 7, m = blockIdx.x >> 3;
        sy = xcd % stripsY;
        sx = (xcd / stripsY) * (stripsX / xcdBands) + m;
    } else { sx = blockIdx.x / stripsY; sy = blockIdx.x - sx * stripsY; }
    const int x0 = sx * PC_C, y0 = sy * S;
    const int rows = min(S, H - y0);                          // output rows of this strip (> 0)
    const int nX = (rows + 2 * R + PC_RS - 1) / PC_RS;        // X iterations; the Y waves run one iteration behind
    if (wv < 8) {
        // ================= X waves: row rr of the iteration, texels 4 li .. 4 li + 3 of the strip; staging texels li + 16 j =================
        const int li = lane & 15, rr = 4 * wv + (lane >> 4);
        uint32_t colOff[6], slotOff[6];
        #pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int p = li + 16 * j;
            colOff[j] = (uint32_t)min(max(x0 - R + p, 0), W - 1) * 8u;         // clamp(sampleCoord.x, 0, iImageSize.x - 1) :143
            slotOff[j] = (uint32_t)(p + (p >> 2)) * 8u;
        }
        const bool stage5 = li + 80 < PC_NPX;                                   // the sixth staging texel exists for li < 4
        u2v pre[6];
        auto fetch = [&](int k) {
            const int r = y0 - R + PC_RS * k + rr;                              // image row of this quarter-wave (clamp :178, or the neighbour tile's rows)
            const unsigned char* rowp;
            if (r < 0 && haloTop)             rowp = (const unsigned char*)haloTop + (size_t)(haloRows + max(r, -haloRows)) * W * 8;
            else if (r > H - 1 && haloBottom) rowp = (const unsigned char*)haloBottom + (size_t)min(r - H, haloRows - 1) * W * 8;
            else                              rowp = (const unsigned char*)in + (size_t)min(max(r, 0), H - 1) * W * 8;
            #pragma unroll
            for (int j = 0; j < 6; ++j) if (j < 5 || stage5) pre[j] = *(const u2v*)(rowp + colOff[j]);
        };
        fetch(0);
        int pb = 0;                                           // (32 k) mod 84: ring slot of the iteration's first row
        for (int k = 0; k <= nX; ++k) {
            if (k < nX) {
                int phys = pb + rr; if (phys >= PC_RING) phys -= PC_RING;
                const uint32_t rowB = PC_TABLE + (uint32_t)phys * PC_ROWB, zRowB = PC_ZBASE + (uint32_t)phys * PC_ZROWB;
                #pragma unroll
                for (int j = 0; j < 6; ++j) if (j < 5 || stage5) *(u2v*)(lds + rowB + slotOff[j]) = pre[j];
                __builtin_amdgcn_wave_barrier();
                Window4 w;                              // the 24-texel window: x | y << 16, z | a << 16
                uint32_t wbase = ldsA + rowB + (uint32_t)(5 * li) * 8u;
                XWindow<23>::read(w.wl, w.wh, wbase);
                if (k + 1 < nX) fetch(k + 1);                 // in flight during the mads below
                w.convert();
                #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float ax, ay, az;
                    w.filter(j, ax, ay, az);
                    // == the store to BlurIntermediate (RGBA16F): rounded to fp16, and kept as the fp32 value of that half — the Y waves filter without converting
                    *(v2f*)(lds + rowB + (uint32_t)(5 * li + j) * 8u) = v2f{ (float)to_f16(ax), (float)to_f16(ay) };
                    *(float*)(lds + zRowB + (uint32_t)(4 * li + j) * 4u) = (float)to_f16(az);
                }
                pb += PC_RS; if (pb >= PC_RING) pb -= PC_RING;
            }
            __syncthreads();
        }
    } else {
        // ================= Y waves: column `lane` of the strip, rows 4 g .. 4 g + 3 of the iteration's 32 outputs =================
        const int g = wv - 8;
        const uint32_t colB = PC_TABLE + (uint32_t)(lane + (lane >> 2)) * 8u, zColB = PC_ZBASE + (uint32_t)lane * 4u;
        const bool xOk = x0 + lane < W;
        uint32_t* __restrict__ dstCol = (uint32_t*)out + (size_t)y0 * W + (xOk ? x0 + lane : 0);
        for (int i = (tid - 512) * 16; i < PC_TABLE; i += 512 * 16) *(uint4*)(lds + i) = *(const uint4*)((const unsigned char*)table + i);
        __syncthreads();                                      // iteration 0: no rows yet
        int pb = PC_RING - 2 * R;                             // (32 (k - 1) - 20) mod 84 for k = 1: ring slot of the first row the iteration's first output reads
        for (int k = 1; k <= nX; ++k) {
            const int oRel = PC_RS * (k - 1) - 2 * R + 4 * g;     // first output row of this wave, relative to y0 (a multiple of 4; negative in the first iteration)
            if (oRel >= 0 && oRel < rows) {
                int q = pb + 4 * g; if (q >= PC_RING) q -= PC_RING;
                Window4 w;                                    // the ring holds fp32 values: nothing to convert
                #pragma unroll
                for (int g4 = 0; g4 < 6; ++g4) {              // q and the ring length are multiples of 4: four consecutive window rows never straddle the ring's wrap
                    int ph = q + 4 * g4; if (ph >= PC_RING) ph -= PC_RING;
                    uint32_t axy = ldsA + colB + (uint32_t)ph * PC_ROWB, az_ = ldsA + zColB + (uint32_t)ph * PC_ZROWB;
                    w.xy[4 * g4 + 0] = lds_load<v2f, 0>(axy);            w.z[4 * g4 + 0] = lds_load<float, 0>(az_);
                    w.xy[4 * g4 + 1] = lds_load<v2f, PC_ROWB>(axy);      w.z[4 * g4 + 1] = lds_load<float, PC_ZROWB>(az_);
                    w.xy[4 * g4 + 2] = lds_load<v2f, 2 * PC_ROWB>(axy);  w.z[4 * g4 + 2] = lds_load<float, 2 * PC_ZROWB>(az_);
                    w.xy[4 * g4 + 3] = lds_load<v2f, 3 * PC_ROWB>(axy);  w.z[4 * g4 + 3] = lds_load<float, 3 * PC_ZROWB>(az_);
                }
                #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (oRel + j >= rows) break;              // wave-uniform
                    float ax, ay, az;
                    w.filter(j, ax, ay, az);
                    const uint32_t hx = float_to_half_bits(ax), hy = float_to_half_bits(ay), hz = float_to_half_bits(az);   // == the BlurOutput store
                    const uint32_t px = (uint32_t)lds[hx] | ((uint32_t)lds[hy] << 8) | ((uint32_t)lds[hz] << 16) | (255u << 24);   // alpha 1 -> 255
                    if (xOk) dstCol[(size_t)(oRel + j) * W] = px;
                }
            }
            pb += PC_RS; if (pb >= PC_RING) pb -= PC_RING;
            __syncthreads();
        }
    }
}

} // namespace

namespace vqk {

hipError_t launch_blur_x(hipStream_t s, const void* in, void* out, int W, int H, int fmt, const Options& opt) {
    const int segsPerRow = (W + 1023) / 1024;
    if ((long long)segsPerRow * H > 0x7fffffffLL) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(segsPerRow * H));
    if (fmt == VQHIP_FMT_RGBA32F) hipLaunchKernelGGL((k_blur_x4<0>), grid, dim3(256), 0, s, in, out, W, H, segsPerRow);
    else                          hipLaunchKernelGGL((k_blur_x4<1>), grid, dim3(256), 0, s, in, out, W, H, segsPerRow);
    return hipGetLastError();
}

hipError_t launch_blur_y(hipStream_t s, const void* in, void* out, const void* haloTop, const void* haloBottom, int haloRows, int W, int H, int fmt) {
    constexpr int ROWS = 16;                                  // register-window variant
    dim3 grid((W + 63) / 64, (H + 4 * ROWS - 1) / (4 * ROWS));
    if (fmt == VQHIP_FMT_RGBA32F) hipLaunchKernelGGL((k_blur_y<0, ROWS>), grid, dim3(256), 0, s, in, out, haloTop, haloBottom, haloRows, W, H);
    else                          hipLaunchKernelGGL((k_blur_y<1, ROWS>), grid, dim3(256), 0, s, in, out, haloTop, haloBottom, haloRows, W, H);
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
        const int tilesX = (W + 63) / 64, tilesY = (H + 127) / 128, nTiles = tilesX * tilesY;
        int wgs = 512;                                        // two 64 KB tables per CU
        if (opt.blurYWgs > 0) wgs = opt.blurYWgs;             // tuning knob
        hipLaunchKernelGGL((k_blur_y_tonemap_lut<16>), dim3(nTiles < wgs ? nTiles : wgs), dim3(512), 0, s, in, out, haloTop, haloBottom, haloRows, W, H,
                           lutTable, tilesX, nTiles);
        return hipGetLastError();
    }
    constexpr int TR = 16;                                    // output rows per workgroup (TR/4 per lane); LDS (TR+20) rows x 64 px
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

// The chain kernel applies: RGBA16F scene colour, RGBA8 target, a display curve that does not mix channels, and a frame large enough to give every CU a strip
bool post_chain_applies(const VQ_TonemapperParams& p, int inFmt, int outFmt, int W, int H, const Options& opt) {
    if (opt.postForm == 1 || !perChannelCurve(p) || inFmt != VQHIP_FMT_RGBA16F || outFmt != VQHIP_FMT_RGBA8_UNORM) return false;
    return opt.postForm == 2 || (long long)W * H >= (1 << 20);
}
// haloTop / haloBottom: SCENE-COLOUR rows of the neighbouring tiles (NULL: image border)
hipError_t launch_post_chain(hipStream_t s, const void* in, void* out, const void* haloTop, const void* haloBottom, int haloRows, int W, int H,
                             const void* lutTable, int nCUs, const Options& opt) {
    const int stripsX = (W + PC_C - 1) / PC_C;
    int stripsY = opt.postStrips > 0 ? opt.postStrips : (nCUs > stripsX ? nCUs / stripsX : 1);      // one workgroup per CU, in ONE round
    const int maxY = (H + 43) / 44;                           // strips shorter than 44 rows filter more halo rows than rows of their own
    if (stripsY > maxY) stripsY = maxY;
    if (stripsY < 1) stripsY = 1;
    const int S = (H + stripsY - 1) / stripsY;
    stripsY = (H + S - 1) / S;
    const int xcdBands = (8 % stripsY == 0 && stripsX % (8 / stripsY) == 0) ? 8 / stripsY : 0;     // XCD-aware strip order (see the kernel) when the strips divide evenly
    hipError_t e = hipFuncSetAttribute((const void*)k_post_chain, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS);      // > 64 KB of LDS needs the opt-in
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_post_chain, dim3((unsigned)(stripsX * stripsY)), dim3(1024), PC_LDS, s, in, out, haloTop, haloBottom, haloRows, W, H, lutTable,
                       stripsX, stripsY, S, xcdBands);
    return hipGetLastError();
}

} // namespace vqk
