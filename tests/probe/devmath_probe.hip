// devmath_probe.hip — TEST INFRASTRUCTURE: exposes the product's device math (vqengine_amd/csrc/vq_devmath.h,
// vq_sampling.h) element-wise so tests can compare it bit-for-bit against the CPU oracle's lowering table, and
// runs exhaustive (all 2^32 bit patterns) checks of the product's fast paths against the plain IEEE forms.
#include <hip/hip_runtime.h>
#include "../../vqengine_amd/csrc/vq_devmath.h"
#include "../../vqengine_amd/csrc/vq_sampling.h"
using namespace vqd;

__global__ void k_probe(int fn, const float* a, const float* b, float* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b ? b[i] : 0.0f, r = 0.0f, s, c;
    switch (fn) {
        case 0: r = log2_(x); break;   case 1: r = exp2_(x); break;   case 2: r = pow_(x, y); break;
        case 3: sincos_(x, &s, &c); r = s; break;                     case 4: sincos_(x, &s, &c); r = c; break;
        case 5: r = tan_(x); break;    case 6: r = asin_(x); break;   case 7: r = acos_(x); break;
        case 8: r = atan2_(x, y); break; case 9: r = rcp(x); break;   case 10: r = sqrt_(x); break; case 11: r = rsqrt(x); break;
        case 12: r = (float)to_f16(x); break;                         // fp32 -> fp16 -> fp32 round trip
        case 13: r = (float)unorm8(x); break;
        case 14: r = max_(x, y); break; case 15: r = min_(x, y); break; case 16: r = saturate(x); break;
        case 17: r = (float)f2i_floor(x); break; case 18: r = (float)f2i_trunc(x); break;
        case 19: r = rsqrt_cr(x); break;
    }
    out[i] = r;
}
extern "C" __attribute__((visibility("default"))) int vqprobe_math(int fn, const float* a, const float* b, float* out, size_t n, void* stream) {
    hipLaunchKernelGGL(k_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, fn, a, b, out, n);
    return (int)hipGetLastError();
}

// normalize() of n float3 vectors in the reading `dxc` (0: as written = v / length(v), the guarded fast form of vq_devmath.h:normalize_lit; 1: v * rsqrt_cr(dot))
__global__ void k_normalize(const float* v, float* out, size_t n, int dxc) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 r = normalize_rt(mk3(v[3 * i], v[3 * i + 1], v[3 * i + 2]), dxc != 0);
    out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
}
extern "C" __attribute__((visibility("default"))) int vqprobe_normalize(const float* v, float* out, size_t n, int dxc, void* stream) {
    hipLaunchKernelGGL(k_normalize, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, v, out, n, dxc);
    return (int)hipGetLastError();
}

// which: 0 rcp() vs 1.0f/x | 1 sqrt_() vs IEEE sqrtf | 2 saturate() vs the select form | 3/4 the unchecked fast paths
// inside their validated domains (rcp_newton: normal result; sqrt_newton: x in [2^-100, FLT_MAX]) | 5 rsqrt_cr() vs (float)(1.0 / sqrt((double)x)) | 6 rsqrt_cr_fast in [2^-100, 2^100]
// | 7 / 8 sqrt_rcp_newton: 1 / sqrtf(x) as two IEEE operations (the reciprocal OF THE ROUNDED ROOT) and the root, x in [2^-100, 2^100]
__global__ void k_exhaust(int which, uint32_t base, unsigned long long* bad, uint32_t* first) {
    const uint32_t u = base + blockIdx.x * blockDim.x + threadIdx.x;
    const float x = __uint_as_float(u);
    float ref, got;
    bool inDomain = true;
    switch (which) {
        case 0:  ref = 1.0f / x; got = rcp(x); break;
        case 1:  ref = __builtin_sqrtf(x); got = sqrt_(x); break;
        case 2:  ref = (x > 0.0f) ? ((x < 1.0f) ? x : 1.0f) : 0.0f; got = saturate(x); break;
        case 3:  ref = 1.0f / x; got = rcp_newton(x); inDomain = is_normal(got); break;
        case 5:  ref = (float)(1.0 / __builtin_sqrt((double)x)); got = rsqrt_cr(x); break;                       // the DXC reading's Rsqrt: definition vs product, all inputs
        case 6:  ref = (float)(1.0 / __builtin_sqrt((double)x)); got = rsqrt_cr_fast(x); inDomain = rsqrt_cr_fast_ok(x); break;   // the unchecked fast sequence inside its domain
        case 7:  { float r; const float D = sqrt_rcp_newton(x, &r); ref = 1.0f / __builtin_sqrtf(x); got = r; inDomain = sqrt_rcp_fast_ok(x) && D == __builtin_sqrtf(x); } break;   // the reciprocal of the root from the root's own v_rsq seed
        case 8:  { float r; ref = __builtin_sqrtf(x); got = sqrt_rcp_newton(x, &r); inDomain = sqrt_rcp_fast_ok(x); } break;                                                        // ... and the root it returns
        default: ref = __builtin_sqrtf(x); got = sqrt_newton(x); inDomain = sqrt_fast_ok(x); break;
    }
    if (inDomain && __float_as_uint(ref) != __float_as_uint(got) && !(ref != ref && got != got)) { if (atomicAdd(bad, 1ull) == 0) *first = u; }
}
extern "C" __attribute__((visibility("default"))) long long vqprobe_exhaustive(int which, uint32_t* first_bad) {
    unsigned long long* d; uint32_t* f;
    if (hipMalloc(&d, 8) != hipSuccess || hipMalloc(&f, 4) != hipSuccess) return -1;
    (void)hipMemset(d, 0, 8); (void)hipMemset(f, 0, 4);
    for (uint32_t hi = 0; hi < 256; ++hi) hipLaunchKernelGGL(k_exhaust, dim3((1u << 24) / 256), dim3(256), 0, 0, which, hi << 24, d, f);
    unsigned long long h = 0;
    if (hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    (void)hipMemcpy(first_bad, f, 4, hipMemcpyDeviceToHost);
    (void)hipFree(d); (void)hipFree(f);
    return (long long)h;
}

// fdiv_rcp(a, b, rcp(b)) vs the IEEE quotient a / b for EVERY pair of significands: a = 1.ma, b = 1.mb, ma, mb in [0, 2^23). Rounding of a
// quotient depends on the significands only (scaling by powers of two is exact while nothing under/overflows), so the 2^46 pairs are a
// proof for all normal-range operands. One thread per mb, looping over a slice of ma; `mb0`/`nb` select the divisors of one launch.
__global__ void k_fdiv_exhaust(uint32_t mb0, uint32_t ma0, uint32_t na, unsigned long long* bad, uint32_t* first) {
    const uint32_t mb = mb0 + blockIdx.x * blockDim.x + threadIdx.x;
    const float b = __uint_as_float(0x3f800000u | mb);
    const float r = rcp_newton(b);
    unsigned long long nbad = 0; uint32_t fa = 0;
    for (uint32_t ma = ma0; ma < ma0 + na; ++ma) {
        const float a = __uint_as_float(0x3f800000u | ma);
        const float got = fdiv_rcp(a, b, r), ref = a / b;
        if (__float_as_uint(got) != __float_as_uint(ref)) { if (!nbad) fa = ma; ++nbad; }
    }
    if (nbad) { if (atomicAdd(bad, nbad) == 0) { first[0] = fa; first[1] = mb; } }
}
// returns the number of mismatching pairs among ma in [ma0, ma0+na) x mb in [mb0, mb0+nb); first_bad = {ma, mb} of one of them
extern "C" __attribute__((visibility("default"))) long long vqprobe_fdiv_exhaustive(uint32_t mb0, uint32_t nb, uint32_t ma0, uint32_t na, uint32_t* first_bad) {
    unsigned long long* d; uint32_t* f;
    if (hipMalloc(&d, 8) != hipSuccess || hipMalloc(&f, 8) != hipSuccess) return -1;
    (void)hipMemset(d, 0, 8); (void)hipMemset(f, 0, 8);
    for (uint32_t o = 0; o < nb; o += 1u << 17) {                      // launches of 2^17 divisors (~0.3 s each for all 2^23 numerators)
        const uint32_t n = (nb - o) < (1u << 17) ? (nb - o) : (1u << 17);
        hipLaunchKernelGGL(k_fdiv_exhaust, dim3(n / 256), dim3(256), 0, 0, mb0 + o, ma0, na, d, f);
        if (hipDeviceSynchronize() != hipSuccess) return -2;
    }
    unsigned long long h = 0;
    if (hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    (void)hipMemcpy(first_bad, f, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d); (void)hipFree(f);
    return (long long)h;
}
