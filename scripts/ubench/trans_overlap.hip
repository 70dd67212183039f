// Do quarter-rate transcendental instructions (v_rcp_f32 / v_rsq_f32) overlap with full-rate VALU work on gfx950, or do they occupy the same issue
// slots? Three loops at 8 waves per SIMD, launched back to back at steady-state clocks: F = 24 independent v_fma_f32 per iteration, T = 8 independent
// v_rcp_f32 per iteration, M = both interleaved. If M takes T + F the units are one pipe (slot-weighted counting is right); if M ~ max(T, F)
// transcendental work can hide behind mads (or the reverse).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define FMA8(a, b) x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b); \
                   x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
#define RCP4 t0 = __builtin_amdgcn_rcpf(t0); t1 = __builtin_amdgcn_rcpf(t1); t2 = __builtin_amdgcn_rcpf(t2); t3 = __builtin_amdgcn_rcpf(t3);
template <int MODE>
__global__ void k(float* out, float a, float b, int iters) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float t0 = x0 + 1.5f, t1 = x0 + 2.5f, t2 = x0 + 3.5f, t3 = x0 + 4.5f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { FMA8(a, b) FMA8(a, b) FMA8(a, b) }
        if (MODE == 1) { RCP4 RCP4 }
        if (MODE == 2) { RCP4 FMA8(a, b) FMA8(a, b) RCP4 FMA8(a, b) }
        if (MODE == 3) { t0 = __builtin_amdgcn_rcpf(t0); FMA8(a, b) t1 = __builtin_amdgcn_rcpf(t1); FMA8(a, b) t2 = __builtin_amdgcn_rcpf(t2); FMA8(a, b) }      // 3 rcp + 24 fma
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + t0 + t1 + t2 + t3;
}
template <int MODE> double run(float* out) {
    const int blocks = 256 * 8, threads = 256, iters = 4000, reps = 30;
    std::vector<hipEvent_t> ev(reps + 1);
    for (auto& e : ev) (void)hipEventCreate(&e);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f, iters);
    (void)hipEventRecord(ev[0]);
    for (int r = 0; r < reps; ++r) { hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f, iters); (void)hipEventRecord(ev[r + 1]); }
    (void)hipDeviceSynchronize();
    std::vector<float> ms(reps);
    for (int r = 0; r < reps; ++r) (void)hipEventElapsedTime(&ms[r], ev[r], ev[r + 1]);
    std::sort(ms.begin(), ms.end());
    return ms[reps / 2];
}
int main() {
    float* out; (void)hipMalloc(&out, 1 << 26);
    run<0>(out);
    const double f = run<0>(out), t = run<1>(out), m = run<2>(out), m3 = run<3>(out);
    printf("{\"fma24_ms\": %.4f, \"rcp8_ms\": %.4f, \"rcp8_fma24_interleaved_ms\": %.4f, \"sum\": %.4f, \"rcp3_fma24_ms\": %.4f, \"rcp_slots\": %.2f}\n", f, t, m, f + t, m3, (t / 8) / (f / 24));
    return 0;
}
