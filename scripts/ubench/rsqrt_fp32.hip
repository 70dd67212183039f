// rsqrt_fp32.hip — exhaustive experiment (all x in [2^-100, 2^100]): the DXC reading's Rsqrt, RN((double)1 / sqrt((double)x)) (vq_devmath.h:rsqrt_cr — what the oracle computes),
// from the v_rsq_f32 seed with binary32 operations only (today: a binary64 second-order tail, ~14 issue slots).
//   t = x*y, d = fma(x, y, -t)   (x*y exactly = t + d)       e = fma(-d, y, fma(-t, y, 1))   (= 1 - x y^2 to ~2^-47)
//   R(b) : r = fma(e + b, y/2, y)        first order + a bias b (the Newton step underestimates by ~3/8 y e^2)
//   S(b) : r = fma(fma(0.375f*e, e, 0.5f*e) + b, y, y)        second order
// Prints the number of x for which each candidate differs.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o scripts/ubench/rsqrt_fp32 scripts/ubench/rsqrt_fp32.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void k(uint32_t base, unsigned long long* bad) {
    const uint32_t u = base + blockIdx.x * blockDim.x + threadIdx.x;
    const float x = __uint_as_float(u);
    if (!(x >= 0x1p-100f && x <= 0x1p100f)) return;
    const float ref = (float)(1.0 / __builtin_sqrt((double)x));
    const float y = __builtin_amdgcn_rsqf(x);
    const float t = x * y, d = __builtin_fmaf(x, y, -t);
    const float e = __builtin_fmaf(-d, y, __builtin_fmaf(-t, y, 1.0f));
    const float h = 0.5f * y;
    const float bias[6] = { 0.0f, 0x1p-49f, 0x1.000002p-48f, 0x1p-47f, 0x1p-46f, 0x1p-45f };
    float c[12];
    for (int i = 0; i < 6; ++i) c[i] = __builtin_fmaf(e + bias[i], h, y);
    const float e2 = __builtin_fmaf(0.375f * e, e, 0.5f * e);
    for (int i = 0; i < 6; ++i) c[6 + i] = __builtin_fmaf(e2 + 0.5f * bias[i], y, y);
    // cheaper groupings of the second-order step (no bias): V1 e * fma(.375, e, .5) as one product; V2 the same on the residual WITHOUT the product-error term d;
    // V3 fma(e, y/2 * fma(.75, e, 1), y); V4 second order on e without d, original grouping
    const float g = __builtin_fmaf(0.375f, e, 0.5f);
    const float v1 = __builtin_fmaf(e * g, y, y);
    const float e0 = __builtin_fmaf(-t, y, 1.0f);
    const float v2 = __builtin_fmaf(e0 * __builtin_fmaf(0.375f, e0, 0.5f), y, y);
    const float v3 = __builtin_fmaf(e, h * __builtin_fmaf(0.75f, e, 1.0f), y);
    const float v4 = __builtin_fmaf(__builtin_fmaf(0.375f * e0, e0, 0.5f * e0), y, y);
    if (__float_as_uint(v1) != __float_as_uint(ref)) atomicAdd(&bad[14], 1ull);
    if (__float_as_uint(v2) != __float_as_uint(ref)) atomicAdd(&bad[15], 1ull);
    if (__float_as_uint(v3) != __float_as_uint(ref)) atomicAdd(&bad[16], 1ull);
    if (__float_as_uint(v4) != __float_as_uint(ref)) atomicAdd(&bad[17], 1ull);
    for (int i = 0; i < 12; ++i) if (__float_as_uint(c[i]) != __float_as_uint(ref)) atomicAdd(&bad[i], 1ull);
    if (__float_as_uint(y) != __float_as_uint(ref)) atomicAdd(&bad[12], 1ull);
    atomicAdd(&bad[13], 1ull);
}
int main() {
    unsigned long long* d; unsigned long long h[18] = {};
    (void)hipMalloc(&d, sizeof(h)); (void)hipMemset(d, 0, sizeof(h));
    for (uint32_t hi = 0; hi < 128; ++hi) hipLaunchKernelGGL(k, dim3((1u << 24) / 256), dim3(256), 0, 0, hi << 24, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("{\"inputs\": %llu, \"seed_alone\": %llu, \"first_order_bias[0,2^-49,>2^-48,2^-47,2^-46,2^-45]\": [%llu, %llu, %llu, %llu, %llu, %llu], \"second_order_same_biases_halved\": [%llu, %llu, %llu, %llu, %llu, %llu]}\n",
           h[13], h[12], h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11]);
    printf("{\"V1_e_times_fma\": %llu, \"V2_V1_without_product_error_term\": %llu, \"V3_fma_e_h_times_fma\": %llu, \"V4_second_order_without_product_error_term\": %llu}\n", h[14], h[15], h[16], h[17]);
    return 0;
}
