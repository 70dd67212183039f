// What does the ACCESS PATTERN of the fused Y blur + tonemap kernel cost, without its arithmetic? 3840x2160 RGBA16F in (8 B/px), RGBA8 out
// (4 B/px): 99.5 MB per launch, every kernel launched back to back after a spin-up, best / median time. The output is a trivial function of
// the input (so nothing is optimised away), the question is only how fast each load / store shape streams:
//   row1   : grid-stride over pixels, 1 px per lane  — 8-byte load, 4-byte store            (k_tonemap_lut's shape)
//   row2   : 2 px per lane                            — 16-byte load, 8-byte store
//   row4   : 4 px per lane                            — two 16-byte loads, 16-byte store
//   colR   : column strips: a wave owns 64 columns x R rows: R 8-byte loads per lane issued up front, then R 4-byte stores   (window shape, no halo)
//   colRh  : the same with the (R+20)-row window of the Y pass (halo rows re-read from L2): the kernel's real load pattern
//   colRhs : colRh whose stores leave as 16 bytes per lane through a wave-private LDS transpose
//   colRhp : colRh software-pipelined: a persistent wave walks down its strip, R new rows per step (the 20 carried rows stay in registers),
//            the loads of step n+1 in flight while step n is "computed" (FMA loop of the real kernel's length when WORK=1)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
#define DEV __device__ __forceinline__
DEV uint32_t squash(uint2 v) { return (v.x & 0xffu) | ((v.x >> 8) & 0xff00u) | ((v.y & 0xffu) << 16) | (v.y & 0xff000000u); }

__global__ __launch_bounds__(256) void row1(const uint2* in, uint32_t* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = squash(in[i]);
}
__global__ __launch_bounds__(256) void row2(const uint4* in, uint2* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n / 2; i += (size_t)gridDim.x * 256) { const uint4 v = in[i]; out[i] = make_uint2(squash(make_uint2(v.x, v.y)), squash(make_uint2(v.z, v.w))); }
}
__global__ __launch_bounds__(256) void row4(const uint4* in, uint4* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n / 4; i += (size_t)gridDim.x * 256) {
        const uint4 a = in[2 * i], b = in[2 * i + 1];
        out[i] = make_uint4(squash(make_uint2(a.x, a.y)), squash(make_uint2(a.z, a.w)), squash(make_uint2(b.x, b.y)), squash(make_uint2(b.z, b.w)));
    }
}
// column strips; HALO = 0 / 10; ST16: 16-byte stores through LDS
template <int R, int HALO, bool ST16>
__global__ __launch_bounds__(256) void colR(const uint2* __restrict__ in, uint32_t* __restrict__ out, int W, int H) {
    __shared__ uint32_t stage[ST16 ? 4 * 256 : 4];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * 64, x = x0 + lane, y0 = (blockIdx.y * 4 + wv) * R;
    if (y0 >= H) return;
    uint2 w[R + 2 * HALO];
    #pragma unroll
    for (int i = 0; i < R + 2 * HALO; ++i) w[i] = in[(size_t)min(max(y0 - HALO + i, 0), H - 1) * W + x];
    uint32_t o[4];
    #pragma unroll
    for (int r = 0; r < R; ++r) {
        uint2 v = w[r + HALO];
        if (HALO) { v.x ^= w[r].x & w[r + 2 * HALO].x & 1u; }
        const uint32_t p = squash(v);
        if (!ST16) { if (y0 + r < H) out[(size_t)(y0 + r) * W + x] = p; continue; }
        o[r & 3] = p;
        if ((r & 3) != 3) continue;
        uint32_t* st = stage + wv * 256;
        #pragma unroll
        for (int k = 0; k < 4; ++k) st[k * 64 + lane] = o[k];
        __builtin_amdgcn_wave_barrier();
        const uint4 q = *(const uint4*)(st + (lane >> 4) * 64 + (lane & 15) * 4);
        __builtin_amdgcn_wave_barrier();
        const int yy = y0 + r - 3 + (lane >> 4);
        if (yy < H) *(uint4*)(out + (size_t)yy * W + x0 + (lane & 15) * 4) = q;
    }
}
// persistent walker: each wave owns a strip of 64 columns and SEG rows, R new rows per step, rows of step n+1 requested before step n is consumed
template <int R, int WORK>
__global__ __launch_bounds__(256) void colWalk(const uint2* __restrict__ in, uint32_t* __restrict__ out, int W, int H, int seg, float wgt) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x = blockIdx.x * 64 + lane, y0 = (blockIdx.y * 4 + wv) * seg;
    if (y0 >= H) return;
    const int y1 = min(y0 + seg, H);
    uint2 carry[20], cur[R], nxt[R];
    #pragma unroll
    for (int i = 0; i < 20; ++i) carry[i] = in[(size_t)min(max(y0 - 10 + i, 0), H - 1) * W + x];
    #pragma unroll
    for (int i = 0; i < R; ++i) cur[i] = in[(size_t)min(y0 + 10 + i, H - 1) * W + x];
    for (int y = y0; y < y1; y += R) {
        #pragma unroll
        for (int i = 0; i < R; ++i) nxt[i] = in[(size_t)min(y + R + 10 + i, H - 1) * W + x];       // next step's new rows: in flight during this step
        #pragma unroll
        for (int r = 0; r < R; ++r) {
            uint2 v = (r < 10) ? carry[r + 10] : cur[r - 10];
            if (WORK) {                                                                             // 63 dependent-ish mads like the 21 taps x 3 channels
                float ax = 0, ay = 0, az = 0;
                #pragma unroll
                for (int t = 0; t < 21; ++t) {
                    const uint2 s = (r + t < 20) ? carry[r + t] : cur[r + t - 20];
                    ax = __builtin_fmaf(__uint_as_float(s.x), wgt, ax); ay = __builtin_fmaf(__uint_as_float(s.y), wgt, ay); az = __builtin_fmaf(__uint_as_float(s.x ^ s.y), wgt, az);
                }
                v.x ^= (__float_as_uint(ax) ^ __float_as_uint(ay) ^ __float_as_uint(az)) & 1u;
            }
            if (y + r < y1) out[(size_t)(y + r) * W + x] = squash(v);
        }
        #pragma unroll
        for (int i = 0; i < 20; ++i) carry[i] = (i + R < 20) ? carry[i + R] : cur[i + R - 20];
        #pragma unroll
        for (int i = 0; i < R; ++i) cur[i] = nxt[i];
    }
}

template <class F> static void bench(const char* name, F launch, double bytes) {
    const int reps = 60;
    std::vector<hipEvent_t> ev(reps + 1);
    for (auto& e : ev) (void)hipEventCreate(&e);
    for (int r = 0; r < 40; ++r) launch();
    (void)hipEventRecord(ev[0]);
    for (int r = 0; r < reps; ++r) { launch(); (void)hipEventRecord(ev[r + 1]); }
    (void)hipDeviceSynchronize();
    std::vector<float> ms(reps);
    for (int r = 0; r < reps; ++r) (void)hipEventElapsedTime(&ms[r], ev[r], ev[r + 1]);
    std::sort(ms.begin(), ms.end());
    printf("{\"pattern\": \"%s\", \"best_us\": %.2f, \"median_us\": %.2f, \"TBps_median\": %.3f, \"frac_of_8TBps\": %.3f}\n", name, ms[0] * 1e3, ms[reps / 2] * 1e3,
           bytes / ms[reps / 2] / 1e9, bytes / ms[reps / 2] / 1e9 / 8.0);
    fflush(stdout);
}
int main() {
    const int W = 3840, H = 2160; const size_t n = (size_t)W * H;
    uint2* in; uint32_t* out;
    (void)hipMalloc(&in, n * 8); (void)hipMalloc(&out, n * 4);
    (void)hipMemset(in, 0x3c, n * 8);
    const double bytes = (double)n * 12;
    for (int wgs : { 2048, 4096, 8192, 16384 }) {
        char nm[64];
        snprintf(nm, sizeof nm, "row1 wgs=%d", wgs); bench(nm, [&] { hipLaunchKernelGGL(row1, dim3(wgs), dim3(256), 0, 0, in, out, n); }, bytes);
        snprintf(nm, sizeof nm, "row2 wgs=%d", wgs); bench(nm, [&] { hipLaunchKernelGGL(row2, dim3(wgs), dim3(256), 0, 0, (const uint4*)in, (uint2*)out, n); }, bytes);
        snprintf(nm, sizeof nm, "row4 wgs=%d", wgs); bench(nm, [&] { hipLaunchKernelGGL(row4, dim3(wgs), dim3(256), 0, 0, (const uint4*)in, (uint4*)out, n); }, bytes);
    }
#define COL(R, HALO, S, NAME) bench(NAME, [&] { hipLaunchKernelGGL((colR<R, HALO, S>), dim3(W / 64, (H + 4 * R - 1) / (4 * R)), dim3(256), 0, 0, in, out, W, H); }, bytes)
    COL(8, 0, false, "col8"); COL(16, 0, false, "col16"); COL(16, 0, true, "col16 st16"); COL(32, 0, false, "col32"); COL(32, 0, true, "col32 st16");
    COL(8, 10, false, "col8h"); COL(16, 10, false, "col16h"); COL(16, 10, true, "col16h st16"); COL(32, 10, false, "col32h"); COL(32, 10, true, "col32h st16");
#define WALK(R, WORK, SEG, NAME) bench(NAME, [&] { hipLaunchKernelGGL((colWalk<R, WORK>), dim3(W / 64, (H + 4 * SEG - 1) / (4 * SEG)), dim3(256), 0, 0, in, out, W, H, SEG, 0.5f); }, bytes)
    WALK(10, 0, 30, "walk10 seg30"); WALK(10, 0, 60, "walk10 seg60"); WALK(10, 0, 120, "walk10 seg120"); WALK(10, 0, 270, "walk10 seg270");
    WALK(10, 1, 30, "walk10 seg30 +mads"); WALK(10, 1, 60, "walk10 seg60 +mads"); WALK(10, 1, 120, "walk10 seg120 +mads"); WALK(10, 1, 270, "walk10 seg270 +mads");
    WALK(16, 1, 64, "walk16 seg64 +mads"); WALK(16, 1, 128, "walk16 seg128 +mads"); WALK(8, 1, 64, "walk8 seg64 +mads"); WALK(8, 1, 136, "walk8 seg136 +mads");
    return 0;
}
