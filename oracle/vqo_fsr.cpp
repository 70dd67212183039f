// vqo_fsr.cpp — CPU restatement of FidelityFX Super Resolution 1.0 as VQEngine dispatches it after the tonemapper
// (SceneRendering.cpp:2695-2784; Shaders/AMDFidelityFX.hlsl:FSR_EASU_CSMain / FSR_RCAS_CSMain compiled WITHOUT FSR_FP16,
// PipelineStateObjects.cpp:1366-1374; algorithm in Shaders/AMDFidelityFX/FSR1.0/ffx_fsr1.h:FsrEasuCon :156-203,
// FsrEasuF :315-437 with FsrEasuSetF :275-313 and FsrEasuTapF :239-273, FsrRcasCon :662-674, FsrRcasF :684-770;
// bit-trick approximations ffx_a.h:1843-1845). SURVEY.md §8(f).4.
//
// ORACLE / TEST INFRASTRUCTURE ONLY (see vqo_oracle.cpp). PARITY: FsrEasuCon / FsrRcasCon and the CPU half packing are PINNED BIT-EXACT
// against the reference's own C++ path (ffx_a.h + ffx_fsr1.h with A_CPU compiled into oracle/_ref/libvqref_fsr.so, as PostProcess.cpp:21-75
// does); Visualization.hlsl is pinned through the HLSL shim (oracle/_ref/libvqref_shaders.so). The EASU / RCAS FILTERS are PINNED BIT-EXACT too:
// AMDFidelityFX.hlsl + ffx_a.h (A_GPU/A_HLSL) + ffx_fsr1.h run on the CPU as dispatched (oracle/ref_src/ref_fsr.cpp); only the gather / load
// addressing is the harness's (fixed-function).
//
// Contract: every operation is a separate IEEE binary32 operation in the order the header writes it (a*b+c is a multiply
// then an add), HLSL rcp() = correctly rounded 1/x, the APrx* approximations are integer bit tricks and exact by
// construction, min/max are IEEE minNum/maxNum, saturate clamps NaN to 0. Texture2D.GatherRed/Green/Blue with a clamp
// sampler at the positions FsrEasuCon builds resolves to the integer texels fp + {0,1} (the gather point sits exactly
// between texel centres), clamped to the image. UNORM8 texels decode as c*rcp(255); the stores round like every other
// pass (RNE fp16 / trunc(sat*255+.5)); RWTexture2D<float3> leaves alpha undefined in the reference: 1 is written.
#include <cmath>
#include <cstdint>
#include <omp.h>

#include "../include/vqhip.h"
#include "vqo_math.h"

using namespace vqo;

namespace {

inline float APrxLoRcp(float a) { return u2f(0x7ef07ebbu - f2u(a)); }                       // ffx_a.h:1843
inline float APrxMedRcp(float a) { const float b = u2f(0x7ef19fffu - f2u(a)); return b * (-b * a + 2.0f); }   // :1844
inline float APrxLoRsq(float a) { return u2f(0x5f347d74u - (f2u(a) >> 1)); }                // :1845
inline float min3(float a, float b, float c) { return min_(a, min_(b, c)); }
inline float max3(float a, float b, float c) { return max_(a, max_(b, c)); }

struct Img { const void* p; int w, h, fmt; };
inline f3 texel(const Img& im, int x, int y) {                                              // clamp-to-edge load, rgb only
    x = x < 0 ? 0 : (x > im.w - 1 ? im.w - 1 : x);
    y = y < 0 ? 0 : (y > im.h - 1 ? im.h - 1 : y);
    const size_t i = (size_t)y * im.w + x;
    if (im.fmt == VQHIP_FMT_RGBA32F) { const float* q = (const float*)im.p + i * 4; return { q[0], q[1], q[2] }; }
    if (im.fmt == VQHIP_FMT_RGBA16F) { const uint16_t* q = (const uint16_t*)im.p + i * 4; return { f16_to_f32(q[0]), f16_to_f32(q[1]), f16_to_f32(q[2]) }; }
    const uint8_t* q = (const uint8_t*)im.p + i * 4;
    const float s = rcp(255.0f);
    return { (float)q[0] * s, (float)q[1] * s, (float)q[2] * s };
}
inline void store(void* base, size_t i, int fmt, f3 c) {
    if (fmt == VQHIP_FMT_RGBA32F) { float* q = (float*)base + i * 4; q[0] = c.x; q[1] = c.y; q[2] = c.z; q[3] = 1.0f; }
    else if (fmt == VQHIP_FMT_RGBA16F) { uint16_t* q = (uint16_t*)base + i * 4; q[0] = f32_to_f16(c.x); q[1] = f32_to_f16(c.y); q[2] = f32_to_f16(c.z); q[3] = f32_to_f16(1.0f); }
    else { uint8_t* q = (uint8_t*)base + i * 4; q[0] = f32_to_unorm8(c.x); q[1] = f32_to_unorm8(c.y); q[2] = f32_to_unorm8(c.z); q[3] = 255; }
}

// FsrEasuSetF, ffx_fsr1.h:275-313
inline void EasuSet(float& dirx, float& diry, float& len, float ppx, float ppy, int which, float lA, float lB, float lC, float lD, float lE) {
    float w = 0.0f;
    if (which == 0) w = (1.0f - ppx) * (1.0f - ppy);
    if (which == 1) w = ppx * (1.0f - ppy);
    if (which == 2) w = (1.0f - ppx) * ppy;
    if (which == 3) w = ppx * ppy;
    const float dc = lD - lC, cb = lC - lB;
    float lenX = max_(abs_(dc), abs_(cb));
    lenX = APrxLoRcp(lenX);
    const float dirX = lD - lB;
    dirx = dirx + dirX * w;
    lenX = saturate(abs_(dirX) * lenX);
    lenX = lenX * lenX;
    len = len + lenX * w;
    const float ec = lE - lC, ca = lC - lA;
    float lenY = max_(abs_(ec), abs_(ca));
    lenY = APrxLoRcp(lenY);
    const float dirY = lE - lA;
    diry = diry + dirY * w;
    lenY = saturate(abs_(dirY) * lenY);
    lenY = lenY * lenY;
    len = len + lenY * w;
}
// FsrEasuTapF, ffx_fsr1.h:239-273
inline void EasuTap(f3& aC, float& aW, float offx, float offy, float dirx, float diry, float lenx, float leny, float lob, float clp, f3 c) {
    float vx = (offx * dirx) + (offy * diry);
    float vy = (offx * (-diry)) + (offy * dirx);
    vx = vx * lenx; vy = vy * leny;
    float d2 = vx * vx + vy * vy;
    d2 = min_(d2, clp);
    float wB = (float)(2.0 / 5.0) * d2 + -1.0f;
    float wA = lob * d2 + -1.0f;
    wB = wB * wB;
    wA = wA * wA;
    wB = (float)(25.0 / 16.0) * wB + (float)(-(25.0 / 16.0 - 1.0));
    const float w = wB * wA;
    aC = { aC.x + c.x * w, aC.y + c.y * w, aC.z + c.z * w };
    aW = aW + w;
}

// FsrEasuF, ffx_fsr1.h:315-437
f3 Easu(const Img& im, int ipx, int ipy, const uint32_t* con) {
    float ppx = (float)ipx * u2f(con[0]) + u2f(con[2]);
    float ppy = (float)ipy * u2f(con[1]) + u2f(con[3]);
    const float fpx = __builtin_floorf(ppx), fpy = __builtin_floorf(ppy);
    ppx = ppx - fpx; ppy = ppy - fpy;
    const int fx = f2i_trunc(fpx), fy = f2i_trunc(fpy);
    //    b c
    //  e f g h
    //  i j k l
    //    n o
    const f3 b = texel(im, fx, fy - 1), c = texel(im, fx + 1, fy - 1);
    const f3 e = texel(im, fx - 1, fy), f = texel(im, fx, fy), g = texel(im, fx + 1, fy), h = texel(im, fx + 2, fy);
    const f3 i = texel(im, fx - 1, fy + 1), j = texel(im, fx, fy + 1), k = texel(im, fx + 1, fy + 1), l = texel(im, fx + 2, fy + 1);
    const f3 n = texel(im, fx, fy + 2), o = texel(im, fx + 1, fy + 2);
    auto luma = [](f3 t) { return t.z * 0.5f + (t.x * 0.5f + t.y); };       // B*0.5 + (R*0.5 + G)
    const float bL = luma(b), cL = luma(c), eL = luma(e), fL = luma(f), gL = luma(g), hL = luma(h), iL = luma(i), jL = luma(j),
                kL = luma(k), lL = luma(l), nL = luma(n), oL = luma(o);
    float dirx = 0.0f, diry = 0.0f, len = 0.0f;
    EasuSet(dirx, diry, len, ppx, ppy, 0, bL, eL, fL, gL, jL);
    EasuSet(dirx, diry, len, ppx, ppy, 1, cL, fL, gL, hL, kL);
    EasuSet(dirx, diry, len, ppx, ppy, 2, fL, iL, jL, kL, nL);
    EasuSet(dirx, diry, len, ppx, ppy, 3, gL, jL, kL, lL, oL);
    const float d2x = dirx * dirx, d2y = diry * diry;
    float dirR = d2x + d2y;
    const bool zro = dirR < (float)(1.0 / 32768.0);
    dirR = APrxLoRsq(dirR);
    dirR = zro ? 1.0f : dirR;
    dirx = zro ? 1.0f : dirx;
    dirx = dirx * dirR; diry = diry * dirR;
    len = len * 0.5f;
    len = len * len;
    const float stretch = (dirx * dirx + diry * diry) * APrxLoRcp(max_(abs_(dirx), abs_(diry)));
    const float len2x = 1.0f + (stretch - 1.0f) * len, len2y = 1.0f + -0.5f * len;
    const float lob = 0.5f + (float)((1.0 / 4.0 - 0.04) - 0.5) * len;
    const float clp = APrxLoRcp(lob);
    const f3 mn4 = { min_(min3(f.x, g.x, j.x), k.x), min_(min3(f.y, g.y, j.y), k.y), min_(min3(f.z, g.z, j.z), k.z) };
    const f3 mx4 = { max_(max3(f.x, g.x, j.x), k.x), max_(max3(f.y, g.y, j.y), k.y), max_(max3(f.z, g.z, j.z), k.z) };
    f3 aC = { 0, 0, 0 }; float aW = 0.0f;
    EasuTap(aC, aW,  0.0f - ppx, -1.0f - ppy, dirx, diry, len2x, len2y, lob, clp, b);
    EasuTap(aC, aW,  1.0f - ppx, -1.0f - ppy, dirx, diry, len2x, len2y, lob, clp, c);
    EasuTap(aC, aW, -1.0f - ppx,  1.0f - ppy, dirx, diry, len2x, len2y, lob, clp, i);
    EasuTap(aC, aW,  0.0f - ppx,  1.0f - ppy, dirx, diry, len2x, len2y, lob, clp, j);
    EasuTap(aC, aW,  0.0f - ppx,  0.0f - ppy, dirx, diry, len2x, len2y, lob, clp, f);
    EasuTap(aC, aW, -1.0f - ppx,  0.0f - ppy, dirx, diry, len2x, len2y, lob, clp, e);
    EasuTap(aC, aW,  1.0f - ppx,  1.0f - ppy, dirx, diry, len2x, len2y, lob, clp, k);
    EasuTap(aC, aW,  2.0f - ppx,  1.0f - ppy, dirx, diry, len2x, len2y, lob, clp, l);
    EasuTap(aC, aW,  2.0f - ppx,  0.0f - ppy, dirx, diry, len2x, len2y, lob, clp, h);
    EasuTap(aC, aW,  1.0f - ppx,  0.0f - ppy, dirx, diry, len2x, len2y, lob, clp, g);
    EasuTap(aC, aW,  1.0f - ppx,  2.0f - ppy, dirx, diry, len2x, len2y, lob, clp, o);
    EasuTap(aC, aW,  0.0f - ppx,  2.0f - ppy, dirx, diry, len2x, len2y, lob, clp, n);
    const float r = rcp(aW);
    return { min_(mx4.x, max_(mn4.x, aC.x * r)), min_(mx4.y, max_(mn4.y, aC.y * r)), min_(mx4.z, max_(mn4.z, aC.z * r)) };
}

// FsrRcasF, ffx_fsr1.h:684-770 (FSR_RCAS_DENOISE and FSR_RCAS_PASSTHROUGH_ALPHA are not defined by the reference)
f3 Rcas(const Img& im, int x, int y, const uint32_t* con) {
    const f3 b = texel(im, x, y - 1), d = texel(im, x - 1, y), e = texel(im, x, y), f = texel(im, x + 1, y), h = texel(im, x, y + 1);
    // Texture2D.Load outside the resource returns 0 — but RCAS runs on an image whose size equals the dispatch, and the
    // reference's loads at -1 / size hit the zero border: restated as zero, not clamp.
    auto ring = [&](int px, int py, f3 v) -> f3 { return (px < 0 || py < 0 || px >= im.w || py >= im.h) ? f3{ 0, 0, 0 } : v; };
    const f3 B = ring(x, y - 1, b), D = ring(x - 1, y, d), F = ring(x + 1, y, f), H = ring(x, y + 1, h);
    const float mn4R = min_(min3(B.x, D.x, F.x), H.x), mn4G = min_(min3(B.y, D.y, F.y), H.y), mn4B = min_(min3(B.z, D.z, F.z), H.z);
    const float mx4R = max_(max3(B.x, D.x, F.x), H.x), mx4G = max_(max3(B.y, D.y, F.y), H.y), mx4B = max_(max3(B.z, D.z, F.z), H.z);
    const float peakCx = 1.0f, peakCy = -1.0f * 4.0f;
    const float hitMinR = mn4R * rcp(4.0f * mx4R), hitMinG = mn4G * rcp(4.0f * mx4G), hitMinB = mn4B * rcp(4.0f * mx4B);
    const float hitMaxR = (peakCx - mx4R) * rcp(4.0f * mn4R + peakCy), hitMaxG = (peakCx - mx4G) * rcp(4.0f * mn4G + peakCy),
                hitMaxB = (peakCx - mx4B) * rcp(4.0f * mn4B + peakCy);
    const float lobeR = max_(-hitMinR, hitMaxR), lobeG = max_(-hitMinG, hitMaxG), lobeB = max_(-hitMinB, hitMaxB);
    const float lobe = max_(-(float)(0.25 - (1.0 / 16.0)), min_(max3(lobeR, lobeG, lobeB), 0.0f)) * u2f(con[0]);
    const float rcpL = APrxMedRcp(4.0f * lobe + 1.0f);
    return { (lobe * B.x + lobe * D.x + lobe * H.x + lobe * F.x + e.x) * rcpL,
             (lobe * B.y + lobe * D.y + lobe * H.y + lobe * F.y + e.y) * rcpL,
             (lobe * B.z + lobe * D.z + lobe * H.z + lobe * F.z + e.z) * rcpL };
}

} // namespace

extern "C" {

// FsrEasuCon, ffx_fsr1.h:156-203 (A_CPU: ARcpF1(a) = 1.0f/a)
void vqo_fsr_easu_con(uint32_t* con, float inVpX, float inVpY, float inSzX, float inSzY, float outX, float outY) {
    con[0] = f2u(inVpX * (1.0f / outX));
    con[1] = f2u(inVpY * (1.0f / outY));
    con[2] = f2u(0.5f * inVpX * (1.0f / outX) - 0.5f);
    con[3] = f2u(0.5f * inVpY * (1.0f / outY) - 0.5f);
    con[4] = f2u(1.0f / inSzX);
    con[5] = f2u(1.0f / inSzY);
    con[6] = f2u(1.0f * (1.0f / inSzX));
    con[7] = f2u(-1.0f * (1.0f / inSzY));
    con[8] = f2u(-1.0f * (1.0f / inSzX));
    con[9] = f2u(2.0f * (1.0f / inSzY));
    con[10] = f2u(1.0f * (1.0f / inSzX));
    con[11] = f2u(2.0f * (1.0f / inSzY));
    con[12] = f2u(0.0f * (1.0f / inSzX));
    con[13] = f2u(4.0f * (1.0f / inSzY));
    con[14] = con[15] = 0;
}
// The CPU-side float -> half packing of ffx_a.h:482-550 (AU1_AH1_AF1, two 512-entry base/shift tables indexed by sign+exponent),
// restated in closed form: it TRUNCATES (no round-to-nearest), flushes everything below the smallest half denormal to a signed
// zero and maps overflow, inf and NaN to +-65504. PINNED: equal to the reference's table for all 2^32 inputs
// (tests/test_ref_pinning.py against oracle/_ref/libvqref_fsr.so, which compiles the reference's header itself).
uint32_t vqo_ffx_half_bits(float f) {
    const uint32_t u = f2u(f), s = (u >> 16) & 0x8000u, e = (u >> 23) & 0xffu, m = u & 0x7fffffu;
    if (e < 103) return s;
    if (e < 113) return s + (1u << (e - 103)) + (m >> (126 - e));
    if (e < 143) return s + ((e - 112) << 10) + (m >> 13);
    return s + 0x7bffu;
}
// FsrRcasCon, ffx_fsr1.h:662-674: con[0] = exp2(-stops); con[1] = the same value as two packed (truncated) halves.
// PINNED against the reference's own FsrRcasCon / FsrEasuCon (oracle/_ref, tests/golden/ref_fsr_con.json).
void vqo_fsr_rcas_con(uint32_t* con, float sharpnessStops) {
    const float s = std::exp2(-sharpnessStops);
    const uint32_t hbits = vqo_ffx_half_bits(s);
    con[0] = f2u(s); con[1] = hbits | (hbits << 16); con[2] = 0; con[3] = 0;
}

int vqo_fsr_easu(const void* in, int inW, int inH, int inFmt, const uint32_t* con, void* out, int outW, int outH, int outFmt, int nthreads) {
    if (!in || !out || !con) return -1;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    const Img im = { in, inW, inH, inFmt };
    #pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int y = 0; y < outH; ++y)
        for (int x = 0; x < outW; ++x) store(out, (size_t)y * outW + x, outFmt, Easu(im, x, y, con));
    return 0;
}
// Visualization.hlsl:CSMain :34-120 (dispatched at SceneRendering.cpp:2541-2576). pow(x, 500) = exp2(500*log2(x)).
int vqo_visualize(const void* in, void* out, int W, int H, const VQ_VizParams* p, int inFmt, int outFmt, int nthreads) {
    if (!in || !out || !p) return -1;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    #pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t i = (size_t)y * W + x;
            f4 t;
            if (inFmt == VQHIP_FMT_RGBA32F) { const float* q = (const float*)in + i * 4; t = { q[0], q[1], q[2], q[3] }; }
            else if (inFmt == VQHIP_FMT_RGBA16F) { const uint16_t* q = (const uint16_t*)in + i * 4; t = { f16_to_f32(q[0]), f16_to_f32(q[1]), f16_to_f32(q[2]), f16_to_f32(q[3]) }; }
            else if (inFmt == VQHIP_FMT_RG16F) { const uint16_t* q = (const uint16_t*)in + i * 2; t = { f16_to_f32(q[0]), f16_to_f32(q[1]), 0.0f, 1.0f }; }   // Tex_SceneMotionVectors: missing channels read (0, 1)
            else if (inFmt == VQHIP_FMT_RG32F) { const float* q = (const float*)in + i * 2; t = { q[0], q[1], 0.0f, 1.0f }; }
            else if (inFmt == VQHIP_FMT_R10G10B10A2_UNORM) {                                                                     // Tex_SceneNormals: UNORM n -> float = c / (2^n - 1)
                const uint32_t q = ((const uint32_t*)in)[i];
                t = { fdiv_((float)(q & 1023u), 1023.0f), fdiv_((float)((q >> 10) & 1023u), 1023.0f), fdiv_((float)((q >> 20) & 1023u), 1023.0f), fdiv_((float)(q >> 30), 3.0f) };
            }
            else { const uint8_t* q = (const uint8_t*)in + i * 4; const float s = rcp(255.0f); t = { q[0] * s, q[1] * s, q[2] * s, q[3] * s }; }
            f3 o;
            switch (p->iDrawMode) {
                case 1: { const float d = pow_(t.x, 500.0f); o = { d, d, d }; } break;
                case 2: {
                    const float u = (float)p->iUnpackNormals, k = (float)(1 - p->iUnpackNormals);
                    o = { ((t.x - 0.5f) * 2.0f) * u + k * t.x, ((t.y - 0.5f) * 2.0f) * u + k * t.y, ((t.z - 0.5f) * 2.0f) * u + k * t.z };
                } break;
                case 3: case 4: o = { t.w, t.w, t.w }; break;
                case 5: o = { t.x, t.x, t.x }; break;
                case 6: case 7: o = { t.x, t.y, t.z }; break;
                case 8: o = { (t.x * 0.5f) * p->fInputStrength + 0.5f, (t.y * -0.5f) * p->fInputStrength + 0.5f, 0.0f + 0.5f }; break;
                default: o = { 1.0f, 0.0f, 1.0f }; break;
            }
            if (outFmt == VQHIP_FMT_RGBA32F) { float* q = (float*)out + i * 4; q[0] = o.x; q[1] = o.y; q[2] = o.z; q[3] = t.w; }
            else if (outFmt == VQHIP_FMT_RGBA16F) { uint16_t* q = (uint16_t*)out + i * 4; q[0] = f32_to_f16(o.x); q[1] = f32_to_f16(o.y); q[2] = f32_to_f16(o.z); q[3] = f32_to_f16(t.w); }
            else { uint8_t* q = (uint8_t*)out + i * 4; q[0] = f32_to_unorm8(o.x); q[1] = f32_to_unorm8(o.y); q[2] = f32_to_unorm8(o.z); q[3] = f32_to_unorm8(t.w); }
        }
    return 0;
}

int vqo_fsr_rcas(const void* in, void* out, int W, int H, const uint32_t* con, int inFmt, int outFmt, int nthreads) {
    if (!in || !out || !con) return -1;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    const Img im = { in, W, H, inFmt };
    #pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) store(out, (size_t)y * W + x, outFmt, Rcas(im, x, y, con));
    return 0;
}

} // extern "C"
