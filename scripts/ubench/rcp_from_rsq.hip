// rcp_from_rsq.hip — exhaustive experiment (all x in [2^-100, 2^100]): can the reciprocal of D = sqrt(x) that the light loop needs next to D (normalize(L - P): D and
// 1/D; vq_shade.h) be refined from the v_rsq_f32 seed the square root already fetched, instead of from a second quarter-rate instruction (v_rcp_f32)?
//   reference : rcp_newton(D) = v_rcp_f32 + one Markstein step (== RN(1/D), proven exhaustively: tests/test_gpu_devmath.py)
//   A         : r = fma(fma(-D, y, 1), y, y)                       y = v_rsq_f32(x)              [2 fma]
//   B         : A, then one more Markstein step on A's result                                     [4 fma]
//   C         : second-order step  e = fma(-D, y, 1), r = fma(fma(e, e, e), y, y)                  [3 fma]
//   P / M     : A from the seed moved one ulp up / down (integer add on the bit pattern)           [1 int + 2 fma]
//   Q         : e from y, correction applied to the seed moved one ulp up: fma(e', y+, y+) with e' = fma(-D, y+, 1)  (== P); Q2: fma(e, y+, y) mixed
// Prints the number of x for which each candidate differs from the reference.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o scripts/ubench/rcp_from_rsq scripts/ubench/rcp_from_rsq.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "../../vqengine_amd/csrc/vq_devmath.h"
using namespace vqd;

__device__ __forceinline__ float newton(float D, float y) { return __builtin_fmaf(__builtin_fmaf(-D, y, 1.0f), y, y); }

__global__ void k(uint32_t base, unsigned long long* bad) {
    const uint32_t u = base + blockIdx.x * blockDim.x + threadIdx.x;
    const float x = __uint_as_float(u);
    if (!(x >= 0x1p-100f && x <= 0x1p100f)) return;
    const float y = __builtin_amdgcn_rsqf(x);
    const float D = sqrt_newton(x);
    const float ref = rcp_newton(D);
    const float yp = __uint_as_float(__float_as_uint(y) + 1u), ym = __uint_as_float(__float_as_uint(y) - 1u);
    const float e = __builtin_fmaf(-D, y, 1.0f);
    float c[10];
    c[0] = newton(D, y);
    c[1] = newton(D, c[0]);
    c[2] = __builtin_fmaf(__builtin_fmaf(e, e, e), y, y);
    c[3] = newton(D, yp);
    c[4] = newton(D, ym);
    c[5] = __builtin_fmaf(e, yp, y);
    c[6] = __builtin_fmaf(e + 0x1p-47f, y, y);          // biased residual: Newton always underestimates (by y e^2); 2^-47 = 2^-24 ulp of the result
    c[7] = __builtin_fmaf(e + 0x1.000002p-48f, y, y);
    c[8] = __builtin_fmaf(e + 0x1.8p-48f, y, y);
    c[9] = __builtin_fmaf(__builtin_fmaf(-D, y, 1.0f + 0x1p-23f) - 0x1p-23f, y, y);
    for (int i = 0; i < 10; ++i) if (__float_as_uint(c[i]) != __float_as_uint(ref)) { if (atomicAdd(&bad[i], 1ull) == 0 && i == 6) { bad[11] = u; bad[12] = __float_as_uint(y); bad[13] = __float_as_uint(D); bad[14] = __float_as_uint(ref); bad[15] = __float_as_uint(c[6]); } }
    atomicAdd(&bad[10], 1ull);
}
int main() {
    unsigned long long* d; unsigned long long h[16] = {};
    (void)hipMalloc(&d, sizeof(h)); (void)hipMemset(d, 0, sizeof(h));
    for (uint32_t hi = 0; hi < 128; ++hi) hipLaunchKernelGGL(k, dim3((1u << 24) / 256), dim3(256), 0, 0, hi << 24, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("{\"inputs\": %llu, \"A_2fma\": %llu, \"B_4fma\": %llu, \"C_second_order\": %llu, \"P_seed_plus_1ulp\": %llu, \"M_seed_minus_1ulp\": %llu, \"Q2_mixed\": %llu, \"bias_2^-47\": %llu, \"bias_just_above_2^-48\": %llu, \"bias_1.5x2^-48\": %llu, \"x9\": %llu}\n",
           h[10], h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9]);
    printf("first bad of bias_2^-47: x=%08llx y=%08llx D=%08llx ref=%08llx got=%08llx\n", h[11], h[12], h[13], h[14], h[15]);
    return 0;
}
