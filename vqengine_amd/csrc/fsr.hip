// fsr.hip — SURVEY.md §8(f).4: FidelityFX Super Resolution 1.0, the tail of VQEngine's post chain
// (SceneRendering.cpp:2695-2784; Shaders/AMDFidelityFX.hlsl:FSR_EASU_CSMain / FSR_RCAS_CSMain, FP32 path).
//
//   k_fsr_easu : FsrEasuF  (Shaders/AMDFidelityFX/FSR1.0/ffx_fsr1.h:315-437, FsrEasuSetF :275-313, FsrEasuTapF :239-273)
//   k_fsr_rcas : FsrRcasF  (:684-770)
//
// One lane per output pixel, 256-lane row segments. The 12 (EASU) / 5 (RCAS) taps are plain clamped loads: the gather
// points FsrEasuCon builds sit exactly between texel centres, so GatherRed/Green/Blue resolve to integer texels. Both
// kernels are VALU-bound (~450 / ~90 operations per pixel against 4-16 B of traffic); the input (a 1440p-class image)
// stays in L2/MALL. Arithmetic: every header operation is its own IEEE operation (no contraction), rcp() correctly
// rounded, the APrx* approximations are integer bit tricks (ffx_a.h:1843-1845).
#include "vq_internal.h"
#include "vq_devmath.h"

namespace vqk {
using namespace vqd;

VQD float APrxLoRcp(float a) { return __uint_as_float(0x7ef07ebbu - __float_as_uint(a)); }
VQD float APrxMedRcp(float a) { const float b = __uint_as_float(0x7ef19fffu - __float_as_uint(a)); return b * (-b * a + 2.0f); }
VQD float APrxLoRsq(float a) { return __uint_as_float(0x5f347d74u - (__float_as_uint(a) >> 1)); }
VQD float min3(float a, float b, float c) { return min_(a, min_(b, c)); }
VQD float max3(float a, float b, float c) { return max_(a, max_(b, c)); }

template <int FMT> VQD f3 texel_at(const void* __restrict__ p, uint32_t i) {
    if (FMT == VQHIP_FMT_RGBA32F) { const float4 q = ((const float4*)p)[i]; return mk3(q.x, q.y, q.z); }
    if (FMT == VQHIP_FMT_RGBA16F) { const float4 q = load_rgba16f(p, i); return mk3(q.x, q.y, q.z); }
    const uint32_t q = ((const uint32_t*)p)[i];
    const float s = 0.0039215688593685627f;                     // rcp(255.0f)
    return mk3((float)(q & 255u) * s, (float)((q >> 8) & 255u) * s, (float)((q >> 16) & 255u) * s);
}
template <int FMT> VQD f3 texel_clamp(const void* __restrict__ p, int W, int H, int x, int y) {
    x = min(max(x, 0), W - 1); y = min(max(y, 0), H - 1);
    return texel_at<FMT>(p, __umul24(y, W) + (uint32_t)x);
}
template <int FMT> VQD void store_rgb1(void* __restrict__ p, uint32_t i, f3 c) {
    if (FMT == VQHIP_FMT_RGBA32F) ((float4*)p)[i] = make_float4(c.x, c.y, c.z, 1.0f);
    else if (FMT == VQHIP_FMT_RGBA16F) store_rgba16f(p, i, make_float4(c.x, c.y, c.z, 1.0f));
    else store_rgba8(p, i, make_float4(c.x, c.y, c.z, 1.0f));
}

struct EasuCon { uint32_t c[16]; };
struct RcasCon { uint32_t c[4]; };

VQD void easu_set(float& dirx, float& diry, float& len, float w, float lA, float lB, float lC, float lD, float lE) {
    const float dc = lD - lC, cb = lC - lB;
    float lenX = APrxLoRcp(max_(abs_(dc), abs_(cb)));
    const float dirX = lD - lB;
    dirx = dirx + dirX * w;
    lenX = saturate(abs_(dirX) * lenX);
    lenX = lenX * lenX;
    len = len + lenX * w;
    const float ec = lE - lC, ca = lC - lA;
    float lenY = APrxLoRcp(max_(abs_(ec), abs_(ca)));
    const float dirY = lE - lA;
    diry = diry + dirY * w;
    lenY = saturate(abs_(dirY) * lenY);
    lenY = lenY * lenY;
    len = len + lenY * w;
}
VQD void easu_tap(f3& aC, float& aW, float offx, float offy, float dirx, float diry, float lenx, float leny, float lob, float clp, f3 c) {
    float vx = (offx * dirx) + (offy * diry);
    float vy = (offx * (-diry)) + (offy * dirx);
    vx = vx * lenx; vy = vy * leny;
    float d2 = vx * vx + vy * vy;
    d2 = min_(d2, clp);
    float wB = 0.4f * d2 + -1.0f;
    float wA = lob * d2 + -1.0f;
    wB = wB * wB;
    wA = wA * wA;
    wB = 1.5625f * wB + -0.5625f;
    const float w = wB * wA;
    aC = mk3(aC.x + c.x * w, aC.y + c.y * w, aC.z + c.z * w);
    aW = aW + w;
}

template <int INFMT, int OUTFMT>
__global__ __launch_bounds__(256) void k_fsr_easu(const void* __restrict__ in, int inW, int inH, EasuCon con, void* __restrict__ out, int outW, int outH) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= outW) return;
    float ppx = (float)x * __uint_as_float(con.c[0]) + __uint_as_float(con.c[2]);
    float ppy = (float)y * __uint_as_float(con.c[1]) + __uint_as_float(con.c[3]);
    const float fpx = __builtin_floorf(ppx), fpy = __builtin_floorf(ppy);
    ppx = ppx - fpx; ppy = ppy - fpy;
    const int fx = f2i_trunc(fpx), fy = f2i_trunc(fpy);
    f3 b, c, e, f, g, h, i, j, k, l, n, o;
    if (fx >= 1 && fy >= 1 && fx + 2 < inW && fy + 2 < inH) {    // whole 4x4 footprint inside the image: one base index, no clamps
        const uint32_t p0 = __umul24(fy, inW) + (uint32_t)fx, W1 = (uint32_t)inW;
        b = texel_at<INFMT>(in, p0 - W1);     c = texel_at<INFMT>(in, p0 - W1 + 1);
        e = texel_at<INFMT>(in, p0 - 1);      f = texel_at<INFMT>(in, p0);          g = texel_at<INFMT>(in, p0 + 1);      h = texel_at<INFMT>(in, p0 + 2);
        i = texel_at<INFMT>(in, p0 + W1 - 1); j = texel_at<INFMT>(in, p0 + W1);     k = texel_at<INFMT>(in, p0 + W1 + 1); l = texel_at<INFMT>(in, p0 + W1 + 2);
        n = texel_at<INFMT>(in, p0 + 2 * W1); o = texel_at<INFMT>(in, p0 + 2 * W1 + 1);
    } else {
        b = texel_clamp<INFMT>(in, inW, inH, fx, fy - 1); c = texel_clamp<INFMT>(in, inW, inH, fx + 1, fy - 1);
        e = texel_clamp<INFMT>(in, inW, inH, fx - 1, fy); f = texel_clamp<INFMT>(in, inW, inH, fx, fy);
        g = texel_clamp<INFMT>(in, inW, inH, fx + 1, fy); h = texel_clamp<INFMT>(in, inW, inH, fx + 2, fy);
        i = texel_clamp<INFMT>(in, inW, inH, fx - 1, fy + 1); j = texel_clamp<INFMT>(in, inW, inH, fx, fy + 1);
        k = texel_clamp<INFMT>(in, inW, inH, fx + 1, fy + 1); l = texel_clamp<INFMT>(in, inW, inH, fx + 2, fy + 1);
        n = texel_clamp<INFMT>(in, inW, inH, fx, fy + 2); o = texel_clamp<INFMT>(in, inW, inH, fx + 1, fy + 2);
    }
    #define VQ_LUMA(t) ((t).z * 0.5f + ((t).x * 0.5f + (t).y))
    const float bL = VQ_LUMA(b), cL = VQ_LUMA(c), eL = VQ_LUMA(e), fL = VQ_LUMA(f), gL = VQ_LUMA(g), hL = VQ_LUMA(h), iL = VQ_LUMA(i),
                jL = VQ_LUMA(j), kL = VQ_LUMA(k), lL = VQ_LUMA(l), nL = VQ_LUMA(n), oL = VQ_LUMA(o);
    #undef VQ_LUMA
    float dirx = 0.0f, diry = 0.0f, len = 0.0f;
    easu_set(dirx, diry, len, (1.0f - ppx) * (1.0f - ppy), bL, eL, fL, gL, jL);
    easu_set(dirx, diry, len, ppx * (1.0f - ppy), cL, fL, gL, hL, kL);
    easu_set(dirx, diry, len, (1.0f - ppx) * ppy, fL, iL, jL, kL, nL);
    easu_set(dirx, diry, len, ppx * ppy, gL, jL, kL, lL, oL);
    const float d2x = dirx * dirx, d2y = diry * diry;
    float dirR = d2x + d2y;
    const bool zro = dirR < 3.0517578125e-05f;                  // 1/32768
    dirR = APrxLoRsq(dirR);
    dirR = zro ? 1.0f : dirR;
    dirx = zro ? 1.0f : dirx;
    dirx = dirx * dirR; diry = diry * dirR;
    len = len * 0.5f;
    len = len * len;
    const float stretch = (dirx * dirx + diry * diry) * APrxLoRcp(max_(abs_(dirx), abs_(diry)));
    const float len2x = 1.0f + (stretch - 1.0f) * len, len2y = 1.0f + -0.5f * len;
    const float lob = 0.5f + -0.29f * len;                      // (1/4 - 0.04) - 0.5, rounded to binary32 like AF1_()
    const float clp = APrxLoRcp(lob);
    const f3 mn4 = mk3(min_(min3(f.x, g.x, j.x), k.x), min_(min3(f.y, g.y, j.y), k.y), min_(min3(f.z, g.z, j.z), k.z));
    const f3 mx4 = mk3(max_(max3(f.x, g.x, j.x), k.x), max_(max3(f.y, g.y, j.y), k.y), max_(max3(f.z, g.z, j.z), k.z));
    f3 aC = mk3(0.0f, 0.0f, 0.0f); float aW = 0.0f;
    easu_tap(aC, aW,  0.0f - ppx, -1.0f - ppy, dirx, diry, len2x, len2y, lob, clp, b);
    easu_tap(aC, aW,  1.0f - ppx, -1.0f - ppy, dirx, diry, len2x, len2y, lob, clp, c);
    easu_tap(aC, aW, -1.0f - ppx,  1.0f - ppy, dirx, diry, len2x, len2y, lob, clp, i);
    easu_tap(aC, aW,  0.0f - ppx,  1.0f - ppy, dirx, diry, len2x, len2y, lob, clp, j);
    easu_tap(aC, aW,  0.0f - ppx,  0.0f - ppy, dirx, diry, len2x, len2y, lob, clp, f);
    easu_tap(aC, aW, -1.0f - ppx,  0.0f - ppy, dirx, diry, len2x, len2y, lob, clp, e);
    easu_tap(aC, aW,  1.0f - ppx,  1.0f - ppy, dirx, diry, len2x, len2y, lob, clp, k);
    easu_tap(aC, aW,  2.0f - ppx,  1.0f - ppy, dirx, diry, len2x, len2y, lob, clp, l);
    easu_tap(aC, aW,  2.0f - ppx,  0.0f - ppy, dirx, diry, len2x, len2y, lob, clp, h);
    easu_tap(aC, aW,  1.0f - ppx,  0.0f - ppy, dirx, diry, len2x, len2y, lob, clp, g);
    easu_tap(aC, aW,  1.0f - ppx,  2.0f - ppy, dirx, diry, len2x, len2y, lob, clp, o);
    easu_tap(aC, aW,  0.0f - ppx,  2.0f - ppy, dirx, diry, len2x, len2y, lob, clp, n);
    const float r = rcp(aW);
    const f3 pix = mk3(min_(mx4.x, max_(mn4.x, aC.x * r)), min_(mx4.y, max_(mn4.y, aC.y * r)), min_(mx4.z, max_(mn4.z, aC.z * r)));
    store_rgb1<OUTFMT>(out, __umul24(y, outW) + (uint32_t)x, pix);
}

template <int INFMT, int OUTFMT>
__global__ __launch_bounds__(256) void k_fsr_rcas(const void* __restrict__ in, void* __restrict__ out, int W, int H, RcasCon con) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const f3 z = mk3(0.0f, 0.0f, 0.0f);                          // Texture2D.Load outside the resource returns 0
    const f3 B = y > 0 ? texel_clamp<INFMT>(in, W, H, x, y - 1) : z, D = x > 0 ? texel_clamp<INFMT>(in, W, H, x - 1, y) : z;
    const f3 e = texel_clamp<INFMT>(in, W, H, x, y);
    const f3 F = x + 1 < W ? texel_clamp<INFMT>(in, W, H, x + 1, y) : z, Hh = y + 1 < H ? texel_clamp<INFMT>(in, W, H, x, y + 1) : z;
    const float mn4R = min_(min3(B.x, D.x, F.x), Hh.x), mn4G = min_(min3(B.y, D.y, F.y), Hh.y), mn4B = min_(min3(B.z, D.z, F.z), Hh.z);
    const float mx4R = max_(max3(B.x, D.x, F.x), Hh.x), mx4G = max_(max3(B.y, D.y, F.y), Hh.y), mx4B = max_(max3(B.z, D.z, F.z), Hh.z);
    const float hitMinR = mn4R * rcp(4.0f * mx4R), hitMinG = mn4G * rcp(4.0f * mx4G), hitMinB = mn4B * rcp(4.0f * mx4B);
    const float hitMaxR = (1.0f - mx4R) * rcp(4.0f * mn4R + -4.0f), hitMaxG = (1.0f - mx4G) * rcp(4.0f * mn4G + -4.0f),
                hitMaxB = (1.0f - mx4B) * rcp(4.0f * mn4B + -4.0f);
    const float lobeR = max_(-hitMinR, hitMaxR), lobeG = max_(-hitMinG, hitMaxG), lobeB = max_(-hitMinB, hitMaxB);
    const float lobe = max_(-0.1875f, min_(max3(lobeR, lobeG, lobeB), 0.0f)) * __uint_as_float(con.c[0]);
    const float rcpL = APrxMedRcp(4.0f * lobe + 1.0f);
    const f3 pix = mk3((lobe * B.x + lobe * D.x + lobe * Hh.x + lobe * F.x + e.x) * rcpL,
                       (lobe * B.y + lobe * D.y + lobe * Hh.y + lobe * F.y + e.y) * rcpL,
                       (lobe * B.z + lobe * D.z + lobe * Hh.z + lobe * F.z + e.z) * rcpL);
    store_rgb1<OUTFMT>(out, __umul24(y, W) + (uint32_t)x, pix);
}

// ---- Visualization.hlsl:CSMain :34-120 (debug draw modes, SceneRendering.cpp:2541-2576): a per-pixel switch, HBM-bound ----
template <int INFMT, int OUTFMT>
__global__ __launch_bounds__(256) void k_visualize(const void* __restrict__ in, void* __restrict__ out, uint32_t n, VQ_VizParams p) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 t;
    if (INFMT == VQHIP_FMT_RGBA32F) t = ((const float4*)in)[i];
    else if (INFMT == VQHIP_FMT_RGBA16F) t = load_rgba16f(in, i);
    else if (INFMT == VQHIP_FMT_RG16F) { const h2 q = ((const h2*)in)[i]; t = make_float4((float)q.x, (float)q.y, 0.0f, 1.0f); }        // Tex_SceneMotionVectors: missing channels read (0, 1)
    else if (INFMT == VQHIP_FMT_RG32F) { const float2 q = ((const float2*)in)[i]; t = make_float4(q.x, q.y, 0.0f, 1.0f); }
    else if (INFMT == VQHIP_FMT_R10G10B10A2_UNORM) {                                                                                     // Tex_SceneNormals: c / (2^n - 1), correctly rounded
        const uint32_t q = ((const uint32_t*)in)[i];
        t = make_float4(fdiv_((float)(q & 1023u), 1023.0f), fdiv_((float)((q >> 10) & 1023u), 1023.0f), fdiv_((float)((q >> 20) & 1023u), 1023.0f), fdiv_((float)(q >> 30), 3.0f));
    }
    else { const uint32_t q = ((const uint32_t*)in)[i]; const float s = 0.0039215688593685627f;
           t = make_float4((float)(q & 255u) * s, (float)((q >> 8) & 255u) * s, (float)((q >> 16) & 255u) * s, (float)(q >> 24) * s); }
    f3 o;
    switch (p.iDrawMode) {
        case 1: { const float d = pow_(t.x, 500.0f); o = mk3(d, d, d); } break;
        case 2: {
            const float u = (float)p.iUnpackNormals, k = (float)(1 - p.iUnpackNormals);
            o = mk3(((t.x - 0.5f) * 2.0f) * u + k * t.x, ((t.y - 0.5f) * 2.0f) * u + k * t.y, ((t.z - 0.5f) * 2.0f) * u + k * t.z);
        } break;
        case 3: case 4: o = mk3(t.w, t.w, t.w); break;
        case 5: o = mk3(t.x, t.x, t.x); break;
        case 6: case 7: o = mk3(t.x, t.y, t.z); break;
        case 8: o = mk3((t.x * 0.5f) * p.fInputStrength + 0.5f, (t.y * -0.5f) * p.fInputStrength + 0.5f, 0.0f + 0.5f); break;
        default: o = mk3(1.0f, 0.0f, 1.0f); break;
    }
    const float4 r = make_float4(o.x, o.y, o.z, t.w);
    if (OUTFMT == VQHIP_FMT_RGBA32F) ((float4*)out)[i] = r;
    else if (OUTFMT == VQHIP_FMT_RGBA16F) store_rgba16f(out, i, r);
    else store_rgba8(out, i, r);
}
// ---- ApplyReflections.hlsl:CSMain :30-50: scene.rgb += reflection.rgb, alpha kept; in place, HBM-bound (24 B/pixel RGBA16F) ----
// BV: the COMPOSITE_BOUNDING_VOLUMES permutation (:44-48; "[PSO] ApplyReflectionsAndBoundingVolumes", ApplyReflections.cpp:82-86): the light-bounds image is
// blended over the sum by its alpha, which also replaces the scene's alpha; as written — two products and one sum per channel, each rounded (32 B/pixel).
template <int FMT, bool BV>
__global__ __launch_bounds__(256) void k_apply_reflections(const void* __restrict__ refl, const void* __restrict__ bv, void* scene, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 r = load_px<FMT>(refl, i), s = load_px<FMT>(scene, i);
    float4 o = make_float4(s.x + r.x, s.y + r.y, s.z + r.z, s.w);
    if (BV) {
        const float4 b = load_px<FMT>(bv, i);
        const float k = 1.0f - b.w;
        o = make_float4(b.x * b.w + o.x * k, b.y * b.w + o.y * k, b.z * b.w + o.z * k, b.w);
    }
    store_px<FMT>(scene, i, o);
}
hipError_t launch_apply_reflections(hipStream_t s, const void* refl, const void* bv, void* scene, int W, int H, int fmt) {
    const uint32_t n = (uint32_t)W * (uint32_t)H;
    const dim3 grid((n + 255) / 256);
    if (bv) {
        if (fmt == VQHIP_FMT_RGBA32F) hipLaunchKernelGGL((k_apply_reflections<0, true>), grid, dim3(256), 0, s, refl, bv, scene, n);
        else                          hipLaunchKernelGGL((k_apply_reflections<1, true>), grid, dim3(256), 0, s, refl, bv, scene, n);
    } else {
        if (fmt == VQHIP_FMT_RGBA32F) hipLaunchKernelGGL((k_apply_reflections<0, false>), grid, dim3(256), 0, s, refl, bv, scene, n);
        else                          hipLaunchKernelGGL((k_apply_reflections<1, false>), grid, dim3(256), 0, s, refl, bv, scene, n);
    }
    return hipGetLastError();
}

template <int INFMT> static hipError_t viz_out(hipStream_t s, const void* in, void* out, uint32_t n, const VQ_VizParams& p, int outFmt) {
    dim3 grid((n + 255) / 256);
    switch (outFmt) {
        case VQHIP_FMT_RGBA32F: hipLaunchKernelGGL((k_visualize<INFMT, VQHIP_FMT_RGBA32F>), grid, dim3(256), 0, s, in, out, n, p); break;
        case VQHIP_FMT_RGBA16F: hipLaunchKernelGGL((k_visualize<INFMT, VQHIP_FMT_RGBA16F>), grid, dim3(256), 0, s, in, out, n, p); break;
        default:                hipLaunchKernelGGL((k_visualize<INFMT, VQHIP_FMT_RGBA8_UNORM>), grid, dim3(256), 0, s, in, out, n, p); break;
    }
    return hipGetLastError();
}
hipError_t launch_visualize(hipStream_t s, const void* in, void* out, int W, int H, const VQ_VizParams& p, int inFmt, int outFmt) {
    const uint32_t n = (uint32_t)W * (uint32_t)H;
    switch (inFmt) {
        case VQHIP_FMT_RGBA32F: return viz_out<VQHIP_FMT_RGBA32F>(s, in, out, n, p, outFmt);
        case VQHIP_FMT_RGBA16F: return viz_out<VQHIP_FMT_RGBA16F>(s, in, out, n, p, outFmt);
        case VQHIP_FMT_RG16F:   return viz_out<VQHIP_FMT_RG16F>(s, in, out, n, p, outFmt);
        case VQHIP_FMT_RG32F:   return viz_out<VQHIP_FMT_RG32F>(s, in, out, n, p, outFmt);
        case VQHIP_FMT_R10G10B10A2_UNORM: return viz_out<VQHIP_FMT_R10G10B10A2_UNORM>(s, in, out, n, p, outFmt);
        default:                return viz_out<VQHIP_FMT_RGBA8_UNORM>(s, in, out, n, p, outFmt);
    }
}

template <int INFMT> static hipError_t easu_out(hipStream_t s, const void* in, int inW, int inH, const EasuCon& con, void* out, int outW, int outH, int outFmt) {
    dim3 grid((outW + 255) / 256, outH);
    switch (outFmt) {
        case VQHIP_FMT_RGBA32F: hipLaunchKernelGGL((k_fsr_easu<INFMT, VQHIP_FMT_RGBA32F>), grid, dim3(256), 0, s, in, inW, inH, con, out, outW, outH); break;
        case VQHIP_FMT_RGBA16F: hipLaunchKernelGGL((k_fsr_easu<INFMT, VQHIP_FMT_RGBA16F>), grid, dim3(256), 0, s, in, inW, inH, con, out, outW, outH); break;
        default:                hipLaunchKernelGGL((k_fsr_easu<INFMT, VQHIP_FMT_RGBA8_UNORM>), grid, dim3(256), 0, s, in, inW, inH, con, out, outW, outH); break;
    }
    return hipGetLastError();
}
hipError_t launch_fsr_easu(hipStream_t s, const void* in, int inW, int inH, int inFmt, const uint32_t* con16, void* out, int outW, int outH, int outFmt) {
    EasuCon con; for (int i = 0; i < 16; ++i) con.c[i] = con16[i];
    switch (inFmt) {
        case VQHIP_FMT_RGBA32F: return easu_out<VQHIP_FMT_RGBA32F>(s, in, inW, inH, con, out, outW, outH, outFmt);
        case VQHIP_FMT_RGBA16F: return easu_out<VQHIP_FMT_RGBA16F>(s, in, inW, inH, con, out, outW, outH, outFmt);
        default:                return easu_out<VQHIP_FMT_RGBA8_UNORM>(s, in, inW, inH, con, out, outW, outH, outFmt);
    }
}
template <int INFMT> static hipError_t rcas_out(hipStream_t s, const void* in, void* out, int W, int H, const RcasCon& con, int outFmt) {
    dim3 grid((W + 255) / 256, H);
    switch (outFmt) {
        case VQHIP_FMT_RGBA32F: hipLaunchKernelGGL((k_fsr_rcas<INFMT, VQHIP_FMT_RGBA32F>), grid, dim3(256), 0, s, in, out, W, H, con); break;
        case VQHIP_FMT_RGBA16F: hipLaunchKernelGGL((k_fsr_rcas<INFMT, VQHIP_FMT_RGBA16F>), grid, dim3(256), 0, s, in, out, W, H, con); break;
        default:                hipLaunchKernelGGL((k_fsr_rcas<INFMT, VQHIP_FMT_RGBA8_UNORM>), grid, dim3(256), 0, s, in, out, W, H, con); break;
    }
    return hipGetLastError();
}
hipError_t launch_fsr_rcas(hipStream_t s, const void* in, void* out, int W, int H, const uint32_t* con4, int inFmt, int outFmt) {
    RcasCon con; for (int i = 0; i < 4; ++i) con.c[i] = con4[i];
    switch (inFmt) {
        case VQHIP_FMT_RGBA32F: return rcas_out<VQHIP_FMT_RGBA32F>(s, in, out, W, H, con, outFmt);
        case VQHIP_FMT_RGBA16F: return rcas_out<VQHIP_FMT_RGBA16F>(s, in, out, W, H, con, outFmt);
        default:                return rcas_out<VQHIP_FMT_RGBA8_UNORM>(s, in, out, W, H, con, outFmt);
    }
}

} // namespace vqk
