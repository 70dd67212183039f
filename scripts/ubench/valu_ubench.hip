// Microbenchmarks that decide the shade-kernel optimisation direction (run on the GPU box):
//  1. v_fma_f32 vs v_pk_fma_f32 issue rate (is packed fp32 a lever on gfx950?)
//  2. exhaustive validation of candidate correctly-rounded rcp / sqrt sequences against IEEE 1/x, sqrt(x)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ void k_fma(float* out, float a, float b, int iters) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
        x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
        x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_pkfma(float* out, float a, float b, int iters) {
    v2f va = {a, a}, vb = {b, b};
    v2f x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    for (int i = 0; i < iters; ++i) {
        x0 = __builtin_elementwise_fma(x0, va, vb); x1 = __builtin_elementwise_fma(x1, va, vb);
        x2 = __builtin_elementwise_fma(x2, va, vb); x3 = __builtin_elementwise_fma(x3, va, vb);
    }
    v2f s = x0 + x1 + x2 + x3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
__global__ void k_rcp_rate(float* out, float a, int iters) {   // quarter-rate check
    float x0 = threadIdx.x + 1.5f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    for (int i = 0; i < iters; ++i) { x0 = __builtin_amdgcn_rcpf(x0) + a; x1 = __builtin_amdgcn_rcpf(x1) + a; x2 = __builtin_amdgcn_rcpf(x2) + a; x3 = __builtin_amdgcn_rcpf(x3) + a; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}

__device__ __forceinline__ float fast_rcp(float b) {
    float r = __builtin_amdgcn_rcpf(b);
    float e = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    return r;
}
__device__ __forceinline__ float fast_rcp1(float b) {   // one Newton step + residual correction (Markstein form)
    float r = __builtin_amdgcn_rcpf(b);
    float e = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    return r;
}
__device__ __forceinline__ float fast_sqrt(float x) {
    float y = __builtin_amdgcn_rsqf(x);
    float g = x * y, h = 0.5f * y;
    float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g); h = __builtin_fmaf(h, r, h);
    float d = __builtin_fmaf(-g, g, x);
    g = __builtin_fmaf(d, h, g);
    return g;
}
// which: 0 rcp 2-step, 1 rcp 1-step, 2 sqrt. Counts mismatches over bit patterns [base, base+n) by result class.
__global__ void k_exhaust(int which, uint32_t base, unsigned long long* counts, uint32_t* examples) {
    uint32_t u = base + blockIdx.x * blockDim.x + threadIdx.x;
    float x = __uint_as_float(u);
    float ref, got;
    if (which == 2) { ref = __builtin_sqrtf(x); got = fast_sqrt(x); } else { ref = 1.0f / x; got = (which == 0) ? fast_rcp(x) : fast_rcp1(x); }
    bool same = (__float_as_uint(ref) == __float_as_uint(got)) || (ref != ref && got != got);
    if (!same) {
        bool refNormal = __builtin_amdgcn_classf(ref, 0x108) && __builtin_amdgcn_classf(x, 0x108);   // +-normal result AND input
        bool gotNormal = __builtin_amdgcn_classf(got, 0x108);
        int bucket = refNormal ? (gotNormal ? 0 : 1) : (gotNormal ? 2 : 3);
        unsigned long long k = atomicAdd(&counts[bucket], 1ull);
        if (bucket == 0 && k < 16) examples[k] = u;
        if (bucket == 2 && k < 16) examples[16 + k] = u;
    }
}

int main() {
    float* out; hipMalloc(&out, 1 << 26);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8, threads = 256, iters = 20000;
    auto timeit = [&](auto launch, const char* name, double flops) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-10s %.3f ms  %.1f Gop-lane/s  (%.1f TFLOP/s)\n", name, ms, flops / 2 / ms / 1e6, flops / ms / 1e9);
    };
    double lanes = (double)blocks * threads;
    timeit([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f, iters); }, "v_fma", lanes * iters * 8 * 2);
    timeit([&] { hipLaunchKernelGGL(k_pkfma, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f, iters); }, "v_pk_fma", lanes * iters * 8 * 2);
    timeit([&] { hipLaunchKernelGGL(k_rcp_rate, dim3(blocks), dim3(threads), 0, 0, out, 0.5f, iters); }, "v_rcp+add", lanes * iters * 4 * 2);
    unsigned long long* counts; uint32_t* ex;
    hipMalloc(&counts, 64); hipMalloc(&ex, 32 * 4);
    for (int which = 0; which < 3; ++which) {
        hipMemset(counts, 0, 64); hipMemset(ex, 0, 128);
        for (uint32_t hi = 0; hi < 256; ++hi)
            hipLaunchKernelGGL(k_exhaust, dim3((1u << 24) / 256), dim3(256), 0, 0, which, hi << 24, counts, ex);
        hipDeviceSynchronize();
        unsigned long long c[4]; uint32_t x[32];
        hipMemcpy(c, counts, 32, hipMemcpyDeviceToHost); hipMemcpy(x, ex, 128, hipMemcpyDeviceToHost);
        printf("exhaust which=%d: mismatch normal->normal %llu, normal-ref/abnormal-got %llu, abnormal-ref/normal-got %llu, both abnormal %llu\n", which, c[0], c[1], c[2], c[3]);
        for (int i = 0; i < 16 && i < (int)c[0]; ++i) printf("  nn example 0x%08x\n", x[i]);
        for (int i = 0; i < 16 && i < (int)c[2]; ++i) printf("  an example 0x%08x\n", x[16 + i]);
    }
    return 0;
}
