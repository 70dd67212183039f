// Issue rate of the instructions a packed-fp16 blur window can be filtered with on gfx950 (round 5): v_fma_f32, v_fma_mix_f32 (fp16 operand converted inside the
// instruction), v_cvt_f32_f16, v_fma_mixlo_f16, v_pk_fma_f32. 8 independent chains per lane, 2048 workgroups of 256 lanes (8 waves per SIMD), 60 launches after a spin-up.
// Prints T lane-instructions/s. usage: ./mix_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHAINS(OP) for (int i = 0; i < iters; ++i) { OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7) }
#define DECL float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; uint32_t h = 0x3c003800u + threadIdx.x;
#define FIN out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
__global__ void k_fma(float* out, float w, int iters) { DECL
#define OPF(x) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(h), "s"(w));
    CHAINS(OPF) FIN }
__global__ void k_mix_lo(float* out, float w, int iters) { DECL
#define OPM(x) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(x) : "v"(h), "s"(w));
    CHAINS(OPM) FIN }
__global__ void k_mix_hi(float* out, float w, int iters) { DECL
#define OPH(x) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x) : "v"(h), "s"(w));
    CHAINS(OPH) FIN }
__global__ void k_mix_vv(float* out, float w, int iters) { DECL float wv = w + threadIdx.x;
#define OPV(x) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(x) : "v"(h), "v"(wv));
    CHAINS(OPV) FIN }
__global__ void k_mix_f32only(float* out, float w, int iters) { DECL float wv = w + threadIdx.x;       // v_fma_mix_f32 with all three operands fp32
#define OPN(x) asm volatile("v_fma_mix_f32 %0, %1, %2, %0" : "+v"(x) : "v"(h), "v"(wv));
    CHAINS(OPN) FIN }
__global__ void k_cvt(float* out, float w, int iters) { DECL
#define OPC(x) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(x) : "v"(h));
    CHAINS(OPC) FIN }
__global__ void k_cvt_sdwa(float* out, float w, int iters) { DECL
#define OPS(x) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(x) : "v"(h));
    CHAINS(OPS) FIN }
__global__ void k_fma_1v(float* out, float w, int iters) { DECL     // one VGPR source: x = x * s + s
#define OP1(x) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "s"(w));
    CHAINS(OP1) FIN }
__global__ void k_fmac_vv(float* out, float w, int iters) { DECL float wv = w + threadIdx.x; float hf = __uint_as_float(h);    // VOP2: x += h * wv
#define OP2(x) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(hf), "v"(wv));
    CHAINS(OP2) FIN }
__global__ void k_fmac_sv(float* out, float w, int iters) { DECL float hf = __uint_as_float(h);    // VOP2: x += s * h
#define OP3(x) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "s"(w), "v"(hf));
    CHAINS(OP3) FIN }
__global__ void k_add(float* out, float w, int iters) { DECL float hf = __uint_as_float(h);
#define OP4(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(hf));
    CHAINS(OP4) FIN }
__global__ void k_mul_s(float* out, float w, int iters) { DECL
#define OP5(x) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x) : "s"(w));
    CHAINS(OP5) FIN }
__global__ void k_fma_3v(float* out, float w, int iters) { DECL float wv = w + threadIdx.x; float hf = __uint_as_float(h);    // VOP3, three VGPR sources
#define OP6(x) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(hf), "v"(wv));
    CHAINS(OP6) FIN }
__global__ void k_mov(float* out, float w, int iters) { DECL float hf = __uint_as_float(h);
#define OP7(x) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(hf));
    CHAINS(OP7) FIN }
__global__ void k_add_inline(float* out, float w, int iters) { DECL     // inline constant 1.0
#define OPA(x) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(x));
    CHAINS(OPA) FIN }
__global__ void k_add_literal(float* out, float w, int iters) { DECL     // 32-bit literal
#define OPB(x) asm volatile("v_add_f32 %0, 0x38d1b717, %0" : "+v"(x));
    CHAINS(OPB) FIN }
__global__ void k_fma_inline(float* out, float w, int iters) { DECL float hf = __uint_as_float(h);    // VOP3 with an inline constant
#define OPD(x) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(x) : "v"(hf));
    CHAINS(OPD) FIN }
__global__ void k_fma_neg(float* out, float w, int iters) { DECL float hf = __uint_as_float(h); float wv = w + threadIdx.x;   // VOP3 with a neg modifier
#define OPE(x) asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(x) : "v"(hf), "v"(wv));
    CHAINS(OPE) FIN }
__global__ void k_max_clamp(float* out, float w, int iters) { DECL
#define OPG(x) asm volatile("v_max_f32_e64 %0, %0, %0 clamp" : "+v"(x));
    CHAINS(OPG) FIN }
__global__ void k_cmp_vcc(float* out, float w, int iters) { DECL float hf = __uint_as_float(h);
#define OPI(x) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" : : "v"(x), "v"(hf) : "vcc");
    CHAINS(OPI) FIN }
__global__ void k_cmp_sgpr_dst(float* out, float w, int iters) { DECL float hf = __uint_as_float(h); unsigned long long m;
#define OPJ(x) asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(x), "v"(hf));
    CHAINS(OPJ) FIN }
__global__ void k_cndmask(float* out, float w, int iters) { DECL float hf = __uint_as_float(h);
#define OPK(x) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(hf) : "vcc");
    CHAINS(OPK) FIN }
__global__ void k_rsq(float* out, float w, int iters) { DECL
#define OPL(x) asm volatile("v_rsq_f32_e32 %0, %0" : "+v"(x));
    CHAINS(OPL) FIN }
__global__ void k_rcp(float* out, float w, int iters) { DECL
#define OPQ(x) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(x));
    CHAINS(OPQ) FIN }
__global__ void k_min3(float* out, float w, int iters) { DECL float hf = __uint_as_float(h); float wv = w + threadIdx.x;
#define OPR(x) asm volatile("v_min3_f32 %0, |%0|, |%1|, |%2|" : "+v"(x) : "v"(hf), "v"(wv));
    CHAINS(OPR) FIN }
__global__ void k_rsq_fma_mix(float* out, float w, int iters) { DECL float hf = __uint_as_float(h); float wv = w + threadIdx.x;    // 1 rsq + 7 fma (all VGPR): does the transcendental overlap?
    for (int i = 0; i < iters; ++i) {
        asm volatile("v_rsq_f32_e32 %0, %0" : "+v"(x0)); OP6(x1) OP6(x2) OP6(x3) OP6(x4) OP6(x5) OP6(x6) OP6(x7) }
    FIN }
__global__ void k_cvt_f16(float* out, float w, int iters) { DECL
#define OPT(x) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(x));
    CHAINS(OPT) FIN }
__global__ void k_lshl_or(float* out, float w, int iters) { DECL
#define OPU(x) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(x) : "v"(h));
    CHAINS(OPU) FIN }
__global__ void k_max(float* out, float w, int iters) { DECL float hf = __uint_as_float(h);
#define OPa(x) asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(x) : "v"(hf));
    CHAINS(OPa) FIN }
__global__ void k_max0(float* out, float w, int iters) { DECL
#define OPb(x) asm volatile("v_max_f32_e32 %0, 0, %0" : "+v"(x));
    CHAINS(OPb) FIN }
__global__ void k_mul_clamp(float* out, float w, int iters) { DECL
#define OPc(x) asm volatile("v_mul_f32_e64 %0, 1.0, %0 clamp" : "+v"(x));
    CHAINS(OPc) FIN }
__global__ void k_add_clamp(float* out, float w, int iters) { DECL float hf = __uint_as_float(h);
#define OPd(x) asm volatile("v_add_f32_e64 %0, %1, %0 clamp" : "+v"(x) : "v"(hf));
    CHAINS(OPd) FIN }
__global__ void k_sub(float* out, float w, int iters) { DECL float hf = __uint_as_float(h);
#define OPe(x) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(x) : "v"(hf));
    CHAINS(OPe) FIN }
__global__ void k_med3(float* out, float w, int iters) { DECL float hf = __uint_as_float(h);
#define OPf(x) asm volatile("v_med3_f32 %0, %0, 0, 1.0" : "+v"(x));
    CHAINS(OPf) FIN }
__global__ void k_mul_abs(float* out, float w, int iters) { DECL float hf = __uint_as_float(h);
#define OPg(x) asm volatile("v_mul_f32_e64 %0, |%1|, %0" : "+v"(x) : "v"(hf));
    CHAINS(OPg) FIN }
__global__ void k_cndmask2(float* out, float w, int iters) { DECL float hf = __uint_as_float(h);
    asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" : : "v"(x0), "v"(hf) : "vcc");
#define OPh(x) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(hf));
    CHAINS(OPh) FIN }
__global__ void k_and(float* out, float w, int iters) { DECL
#define OPi(x) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(x) : "v"(h));
    CHAINS(OPi) FIN }
__global__ void k_addu(float* out, float w, int iters) { DECL
#define OPj(x) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(x) : "v"(h));
    CHAINS(OPj) FIN }
__global__ void k_fmamk(float* out, float w, int iters) { DECL float hf = __uint_as_float(h);
#define OPk(x) asm volatile("v_fmac_f32_e32 %0, 0x3e661e7b, %1" : "+v"(x) : "v"(hf));
    CHAINS(OPk) FIN }
__global__ void k_mov_s(float* out, float w, int iters) { DECL
#define OPl(x) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "s"(w));
    CHAINS(OPl) FIN }
__global__ void k_pkfma(float* out, float w, int iters) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, a = {w, w}, b = {w, 1.0f};
    for (int i = 0; i < iters; ++i) {
#define OPP(x) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
        OPP(x0) OPP(x1) OPP(x2) OPP(x3) OPP(x0) OPP(x1) OPP(x2) OPP(x3) }
    v2f s = x0 + x1 + x2 + x3; out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y; }
int main() {
    float* out; hipMalloc(&out, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 2048, threads = 256, iters = 4000, reps = 60;
    auto run = [&](auto k, const char* name) {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, 0.5f, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, 0.5f, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)blocks * threads * iters * 8.0 * reps;
        printf("{\"instr\": \"%s\", \"T_lane_instr_per_s\": %.2f, \"cycles_per_wave_instr_at_2.4GHz_per_SIMD\": %.2f}\n", name, n / (ms * 1e-3) / 1e12, 2.4e9 * 1024 * 64 / (n / (ms * 1e-3)));
    };
    run(k_fma, "v_fma_f32 (sgpr weight)"); run(k_mix_lo, "v_fma_mix_f32 lo half (sgpr weight)"); run(k_mix_hi, "v_fma_mix_f32 hi half (sgpr weight)");
    run(k_mix_vv, "v_fma_mix_f32 lo half (vgpr weight)"); run(k_mix_f32only, "v_fma_mix_f32 all fp32"); run(k_cvt, "v_cvt_f32_f16"); run(k_cvt_sdwa, "v_cvt_f32_f16_sdwa WORD_1");
    run(k_pkfma, "v_pk_fma_f32 (instructions; 2 fma each)");
    run(k_fma_1v, "v_fma_f32 x = x*s+s (one VGPR source)"); run(k_fmac_vv, "v_fmac_f32 VOP2 v,v"); run(k_fmac_sv, "v_fmac_f32 VOP2 s,v"); run(k_add, "v_add_f32 VOP2"); run(k_mul_s, "v_mul_f32 VOP2 s,v");
    run(k_fma_3v, "v_fma_f32 three VGPR sources"); run(k_mov, "v_mov_b32");
    run(k_add_inline, "v_add_f32 inline constant 1.0"); run(k_add_literal, "v_add_f32 32-bit literal"); run(k_fma_inline, "v_fma_f32 v,v,1.0"); run(k_fma_neg, "v_fma_f32 -v,v,v");
    run(k_max_clamp, "v_max_f32_e64 clamp"); run(k_cmp_vcc, "v_cmp_gt_f32_e32 vcc"); run(k_cmp_sgpr_dst, "v_cmp_gt_f32_e64 sgpr pair dst"); run(k_cndmask, "v_cndmask_b32 vcc");
    run(k_rsq, "v_rsq_f32"); run(k_rcp, "v_rcp_f32"); run(k_min3, "v_min3_f32 |v|,|v|,|v|"); run(k_rsq_fma_mix, "1 v_rsq_f32 + 7 v_fma_f32 (vvv) per 8"); run(k_cvt_f16, "v_cvt_f16_f32");
    run(k_lshl_or, "v_lshl_or_b32 v, 8, v");
    run(k_max, "v_max_f32_e32 v,v"); run(k_max0, "v_max_f32_e32 0,v"); run(k_mul_clamp, "v_mul_f32_e64 1.0,v clamp"); run(k_add_clamp, "v_add_f32_e64 v,v clamp"); run(k_sub, "v_sub_f32_e32 v,v");
    run(k_med3, "v_med3_f32 v,0,1.0"); run(k_mul_abs, "v_mul_f32_e64 |v|,v"); run(k_cndmask2, "v_cndmask_b32 v,v,vcc"); run(k_and, "v_and_b32 v,v"); run(k_addu, "v_add_u32 v,v");
    run(k_fmamk, "v_fmac_f32_e32 literal,v"); run(k_mov_s, "v_mov_b32 v, s");
    return 0;
}
