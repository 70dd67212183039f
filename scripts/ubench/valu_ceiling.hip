// Steady-state VALU issue ceiling of the chip: dependent-free v_fma_f32 (8 independent chains per lane, 8 waves per SIMD),
// the same kernel launched 60 times back to back so the clocks are ramped; prints best / median lane-instructions per second.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
__global__ void k_fma(float* out, float a, float b, int iters) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
        x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
        x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
int main() {
    float* out; (void)hipMalloc(&out, 1 << 26);
    const int blocks = 256 * 8, threads = 256, iters = 20000, reps = 60;
    std::vector<hipEvent_t> ev(reps + 1);
    for (auto& e : ev) (void)hipEventCreate(&e);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f, iters);
    (void)hipEventRecord(ev[0]);
    for (int r = 0; r < reps; ++r) { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f, iters); (void)hipEventRecord(ev[r + 1]); }
    (void)hipDeviceSynchronize();
    std::vector<float> ms(reps);
    for (int r = 0; r < reps; ++r) (void)hipEventElapsedTime(&ms[r], ev[r], ev[r + 1]);
    std::sort(ms.begin(), ms.end());
    const double li = (double)blocks * threads * iters * 8;      // lane-instructions per launch
    printf("v_fma_f32 steady state: best %.2f T lane-instr/s (%.1f TFLOP/s), median %.2f T lane-instr/s (%.1f TFLOP/s), worst %.2f\n",
           li / ms[0] / 1e9, 2 * li / ms[0] / 1e9, li / ms[reps / 2] / 1e9, 2 * li / ms[reps / 2] / 1e9, li / ms[reps - 1] / 1e9);
    return 0;
}
