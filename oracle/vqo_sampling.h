// vqo_sampling.h — software texture sampling semantics used by the oracle.
//
// ORACLE / TEST INFRASTRUCTURE ONLY (see vqo_math.h header). PARITY UNPINNED (fixed-function hardware, no source in the reference): D3D12/WARP filtering
// bits are not observable here; this fixes one D3D-conformant behaviour:
//   * texel coordinates converted to fixed point with 8 fractional bits (D3D11.3 functional spec
//     §7.18.7 "Fixed point texture coordinates"), linear weights = those 8-bit fractions;
//   * bilinear = sum of 4 weighted taps, weights (1-wx)(1-wy).. are exact in fp32, accumulated as the
//     FMA chain fma(w11,c11, fma(w01,c01, fma(w10,c10, w00*c00)));
//   * trilinear (fractional LOD) = (1-f)*lo + f*hi with f quantised to 8 bits, hi skipped when f == 0;
//   * cube maps: major-axis face select (z over y over x on ties, as v_cubeid), seamless bilinear — a
//     tap that leaves the face through ONE edge is fetched from the adjacent face's edge texel with the
//     same along-edge position; a tap that leaves through a corner is replaced by the mean of the other
//     three taps of the footprint;
//   * samplers: ClampedLinearSampler s3 = TRILINEAR_CLAMP (Source/Renderer/Pipeline/RootSignatures.cpp:150),
//     PointSampler s1 = POINT_WRAP (:148), convolution Sampler = TRILINEAR_WRAP (:402).
#ifndef VQO_SAMPLING_H
#define VQO_SAMPLING_H

#include "../include/vqhip.h"
#include "vqo_math.h"

namespace vqo {

// Cube face bases == CubemapUtility::CalculateViewMatrix (Source/Renderer/Resources/CubemapUtility.cpp:40-49)
// with LookAtLH: forward F, up U (the 'up' argument), right R = cross(U, F).
// face order +X,-X,+Y,-Y,+Z,-Z (CubemapUtility.h:26-36).
struct FaceBasis { int F[3], U[3], R[3]; };
static const FaceBasis kFace[6] = {
    { { 1, 0, 0}, {0, 1, 0}, { 0, 0,-1} },   // +X  up = +Y
    { {-1, 0, 0}, {0, 1, 0}, { 0, 0, 1} },   // -X  up = +Y
    { { 0, 1, 0}, {0, 0,-1}, { 1, 0, 0} },   // +Y  up = -Z (VEC3_BACK)
    { { 0,-1, 0}, {0, 0, 1}, { 1, 0, 0} },   // -Y  up = +Z (VEC3_FORWARD)
    { { 0, 0, 1}, {0, 1, 0}, { 1, 0, 0} },   // +Z  up = +Y
    { { 0, 0,-1}, {0, 1, 0}, {-1, 0, 0} },   // -Z  up = +Y
};

// texel (x,y) of a res x res face -> direction through the texel centre:
// the cube mesh is [-1,1]^3 (MeshGenerator.h:227-250), VSMain_PerFace passes the local position
// through (CubemapConvolution.hlsl:63-73), so dir = F + u*R + v*U, u = 2(x+.5)/W - 1, v = 1 - 2(y+.5)/H.
static inline f3 cube_texel_dir(int face, int x, int y, int res) {
    float inv = rcp((float)res);
    float u = (2.0f * ((float)x + 0.5f)) * inv - 1.0f;
    float v = 1.0f - (2.0f * ((float)y + 0.5f)) * inv;
    const FaceBasis& b = kFace[face];
    // components of F,R,U are 0/+-1: products and sums with zeros are exact
    return { (float)b.F[0] + u * (float)b.R[0] + v * (float)b.U[0],
             (float)b.F[1] + u * (float)b.R[1] + v * (float)b.U[1],
             (float)b.F[2] + u * (float)b.R[2] + v * (float)b.U[2] };
}

// direction -> (face, sc/ma, tc/ma) with tc pointing DOWN the texture (row index direction).
static inline int cube_face_uv(f3 d, float* su, float* sv) {
    float ax = abs_(d.x), ay = abs_(d.y), az = abs_(d.z);
    int face; float sc, tc, ma;
    if (az >= ax && az >= ay) { ma = az; if (d.z < 0.0f) { face = 5; sc = -d.x; tc = -d.y; } else { face = 4; sc =  d.x; tc = -d.y; } }
    else if (ay >= ax)        { ma = ay; if (d.y < 0.0f) { face = 3; sc =  d.x; tc = -d.z; } else { face = 2; sc =  d.x; tc =  d.z; } }
    else                      { ma = ax; if (d.x < 0.0f) { face = 1; sc =  d.z; tc = -d.y; } else { face = 0; sc = -d.z; tc = -d.y; } }
    float r = rcp(ma);
    *su = (sc * r) * 0.5f + 0.5f;
    *sv = (tc * r) * 0.5f + 0.5f;
    return face;
}

// Resolve a tap (i,j) that left face `f` through exactly one edge onto the neighbouring face.
// Integer arithmetic in units of 1/N: texel centre i has coordinate a = 2i+1-N along R, b = N-(2j+1) along U.
static inline void cube_edge_neighbor(int f, int i, int j, int N, int* nf, int* ni, int* nj) {
    const FaceBasis& b = kFace[f];
    int a  = 2 * i + 1 - N;       // along R (sc)
    int bb = N - (2 * j + 1);     // along U (= -tc)
    int axis[3]; int sgn;         // the edge we crossed: +-R or +-U
    if (i < 0)       { sgn = -1; for (int k = 0; k < 3; ++k) axis[k] = b.R[k]; a = -N; }
    else if (i >= N) { sgn =  1; for (int k = 0; k < 3; ++k) axis[k] = b.R[k]; a =  N; }
    else if (j < 0)  { sgn =  1; for (int k = 0; k < 3; ++k) axis[k] = b.U[k]; bb =  N; }
    else             { sgn = -1; for (int k = 0; k < 3; ++k) axis[k] = b.U[k]; bb = -N; }
    int q[3];
    for (int k = 0; k < 3; ++k) q[k] = N * b.F[k] + a * b.R[k] + bb * b.U[k];     // point on the shared edge
    int g = -1;
    for (int t = 0; t < 6; ++t) {
        if (kFace[t].F[0] == sgn * axis[0] && kFace[t].F[1] == sgn * axis[1] && kFace[t].F[2] == sgn * axis[2]) { g = t; break; }
    }
    const FaceBasis& n = kFace[g];
    int a2 = q[0] * n.R[0] + q[1] * n.R[1] + q[2] * n.R[2];
    int b2 = q[0] * n.U[0] + q[1] * n.U[1] + q[2] * n.U[2];
    int ii = (a2 >= N) ? N - 1 : (a2 <= -N) ? 0 : (a2 + N - 1) / 2;
    int jj = (b2 >= N) ? 0 : (b2 <= -N) ? N - 1 : (N - 1 - b2) / 2;
    *nf = g; *ni = ii; *nj = jj;
}

// fixed-point split of a texel-space coordinate: x = u*N - 0.5 -> integer texel + 8-bit fraction
static inline void fixed8(float x, int* ix, float* w) {
    int fx = f2i_floor(x * 256.0f + 0.5f);
    *ix = fx >> 8;                                  // arithmetic shift == floor division
    *w = (float)(fx & 255) * 0.00390625f;
}

static inline f4 load_rgba16f(const uint16_t* p) { return { f16_to_f32(p[0]), f16_to_f32(p[1]), f16_to_f32(p[2]), f16_to_f32(p[3]) }; }

static inline f4 blend4(f4 c00, f4 c10, f4 c01, f4 c11, float wx, float wy) {
    float w00 = (1.0f - wx) * (1.0f - wy), w10 = wx * (1.0f - wy), w01 = (1.0f - wx) * wy, w11 = wx * wy;
    f4 r;
    r.x = fma_(w11, c11.x, fma_(w01, c01.x, fma_(w10, c10.x, w00 * c00.x)));
    r.y = fma_(w11, c11.y, fma_(w01, c01.y, fma_(w10, c10.y, w00 * c00.y)));
    r.z = fma_(w11, c11.z, fma_(w01, c01.z, fma_(w10, c10.z, w00 * c00.z)));
    r.w = fma_(w11, c11.w, fma_(w01, c01.w, fma_(w10, c10.w, w00 * c00.w)));
    return r;
}

// Seamless bilinear sample of one mip of an RGBA16F cube stored [6][N][N][4].
static inline f4 sample_cube_rgba16f(const uint16_t* cube, int N, f3 dir) {
    float su, sv;
    int f = cube_face_uv(dir, &su, &sv);
    int ix, iy; float wx, wy;
    fixed8(su * (float)N - 0.5f, &ix, &wx);
    fixed8(sv * (float)N - 0.5f, &iy, &wy);
    f4 c[4]; bool ok[4];
    for (int t = 0; t < 4; ++t) {
        int i = ix + (t & 1), j = iy + (t >> 1);
        bool ox = (i < 0 || i >= N), oy = (j < 0 || j >= N);
        ok[t] = true;
        if (!ox && !oy) {
            c[t] = load_rgba16f(cube + (((size_t)f * N + j) * N + i) * 4);
        } else if (ox && oy) {
            ok[t] = false; c[t] = { 0, 0, 0, 0 };
        } else {
            int nf, ni, nj;
            cube_edge_neighbor(f, i, j, N, &nf, &ni, &nj);
            c[t] = load_rgba16f(cube + (((size_t)nf * N + nj) * N + ni) * 4);
        }
    }
    for (int t = 0; t < 4; ++t) {
        if (!ok[t]) {   // corner: mean of the other three taps, in tap order
            f4 s = { 0, 0, 0, 0 }; bool first = true;
            for (int k = 0; k < 4; ++k) {
                if (k == t) continue;
                if (first) { s = c[k]; first = false; }
                else { s.x += c[k].x; s.y += c[k].y; s.z += c[k].z; s.w += c[k].w; }
            }
            const float third = 0.333333343267440796f;
            c[t] = { s.x * third, s.y * third, s.z * third, s.w * third };
        }
    }
    return blend4(c[0], c[1], c[2], c[3], wx, wy);
}

// Bilinear CLAMP sample of an RG16F 2D texture [H][W][2] (BRDF LUT).
static inline f2 sample_2d_rg16f_clamp(const uint16_t* tex, int W, int H, float u, float v) {
    int ix, iy; float wx, wy;
    fixed8(u * (float)W - 0.5f, &ix, &wx);
    fixed8(v * (float)H - 0.5f, &iy, &wy);
    int x0 = ix < 0 ? 0 : (ix > W - 1 ? W - 1 : ix), x1 = ix + 1 < 0 ? 0 : (ix + 1 > W - 1 ? W - 1 : ix + 1);
    int y0 = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy), y1 = iy + 1 < 0 ? 0 : (iy + 1 > H - 1 ? H - 1 : iy + 1);
    auto ld = [&](int x, int y) -> f4 { const uint16_t* p = tex + ((size_t)y * W + x) * 2; return { f16_to_f32(p[0]), f16_to_f32(p[1]), 0, 0 }; };
    f4 r = blend4(ld(x0, y0), ld(x1, y0), ld(x0, y1), ld(x1, y1), wx, wy);
    return { r.x, r.y };
}

// Bilinear WRAP sample of one RGBA32F level [H][W][4] (equirect mip).
static inline f4 sample_2d_rgba32f_wrap(const float* tex, int W, int H, float u, float v) {
    int ix, iy; float wx, wy;
    fixed8(u * (float)W - 0.5f, &ix, &wx);
    fixed8(v * (float)H - 0.5f, &iy, &wy);
    auto wrap = [](int i, int n) { int m = i % n; return m < 0 ? m + n : m; };
    int x0 = wrap(ix, W), x1 = wrap(ix + 1, W), y0 = wrap(iy, H), y1 = wrap(iy + 1, H);
    auto ld = [&](int x, int y) -> f4 { const float* p = tex + ((size_t)y * W + x) * 4; return { p[0], p[1], p[2], p[3] }; };
    return blend4(ld(x0, y0), ld(x1, y0), ld(x0, y1), ld(x1, y1), wx, wy);
}

// mip chain helpers (dense RGBA32F chain: level 0 first)
static inline int mip_level_count(int w, int h) { int m = w > h ? w : h; int n = 1; while (m > 1) { m >>= 1; ++n; } return n; }
static inline int mip_dim(int d0, int level) { int d = d0 >> level; return d < 1 ? 1 : d; }
static inline size_t mip_offset_floats(int w0, int h0, int level) {
    size_t off = 0;
    for (int l = 0; l < level; ++l) off += (size_t)mip_dim(w0, l) * mip_dim(h0, l) * 4;
    return off;
}

// SampleLevel(uv, lod) on the equirect chain with a TRILINEAR_WRAP sampler.
static inline f4 sample_equirect_lod(const float* chain, int w0, int h0, int nMips, float u, float v, float lod) {
    float maxl = (float)(nMips - 1);
    float l = (lod > 0.0f) ? ((lod < maxl) ? lod : maxl) : 0.0f;     // NaN -> 0
    int fl = f2i_floor(l * 256.0f + 0.5f);
    int lo = fl >> 8;
    float f = (float)(fl & 255) * 0.00390625f;
    if (lo >= nMips - 1) { lo = nMips - 1; f = 0.0f; }
    f4 a = sample_2d_rgba32f_wrap(chain + mip_offset_floats(w0, h0, lo), mip_dim(w0, lo), mip_dim(h0, lo), u, v);
    if (f == 0.0f) return a;
    f4 b = sample_2d_rgba32f_wrap(chain + mip_offset_floats(w0, h0, lo + 1), mip_dim(w0, lo + 1), mip_dim(h0, lo + 1), u, v);
    float g = 1.0f - f;
    return { fma_(f, b.x, g * a.x), fma_(f, b.y, g * a.y), fma_(f, b.z, g * a.z), fma_(f, b.w, g * a.w) };
}

// SampleLevel(dir, lod) on a mip-major RGBA16F cube with a MIN_MAG_MIP_LINEAR sampler (ScreenSpaceReflections.cpp:632-648, used by
// ClassifyReflectionTiles.hlsl:89 with the fractional lod roughness * (mip_count - 1)): lod clamped to the chain, 8-bit level fraction, the two
// seamless bilinear fetches blended as fma(f, hi, (1-f)*lo) — the rule of sample_equirect_lod above. An integral lod reads one level.
static inline size_t cube_mip_offset_halfs_(int res0, int mip) {
    size_t off = 0;
    for (int m = 0; m < mip; ++m) { const size_t r = (size_t)(res0 >> m); off += 6 * r * r * 4; }
    return off;
}
static inline f4 sample_cube_lod_rgba16f(const uint16_t* cube, int res0, int nMips, f3 dir, float lod) {
    const float maxl = (float)(nMips - 1);
    const float l = (lod > 0.0f) ? ((lod < maxl) ? lod : maxl) : 0.0f;     // NaN -> 0
    const int fl = f2i_floor(l * 256.0f + 0.5f);
    int lo = fl >> 8;
    float f = (float)(fl & 255) * 0.00390625f;
    if (lo >= nMips - 1) { lo = nMips - 1; f = 0.0f; }
    const f4 a = sample_cube_rgba16f(cube + cube_mip_offset_halfs_(res0, lo), res0 >> lo, dir);
    if (f == 0.0f) return a;
    const f4 b = sample_cube_rgba16f(cube + cube_mip_offset_halfs_(res0, lo + 1), res0 >> (lo + 1), dir);
    const float g = 1.0f - f;
    return { fma_(f, b.x, g * a.x), fma_(f, b.y, g * a.y), fma_(f, b.z, g * a.z), fma_(f, b.w, g * a.w) };
}

// ---- material textures (RGBA8_UNORM mip chains, vqo_gbuffer.cpp header) and shadow maps (R32F) ----
static inline float unorm8_to_float(float c) { return c * rcp(255.0f); }

static inline size_t tex_level_offset_px(int w0, int h0, int level) {
    size_t off = 0;
    for (int l = 0; l < level; ++l) off += (size_t)mip_dim(w0, l) * mip_dim(h0, l);
    return off;
}

// bilinear WRAP of one RGBA8 level, in byte units (exact)
static inline f4 sample_2d_rgba8_wrap(const uint8_t* tex, int W, int H, float u, float v) {
    int ix, iy; float wx, wy;
    fixed8(u * (float)W - 0.5f, &ix, &wx);
    fixed8(v * (float)H - 0.5f, &iy, &wy);
    auto wrap = [](int i, int n) { int m = i % n; return m < 0 ? m + n : m; };
    const int x0 = wrap(ix, W), x1 = wrap(ix + 1, W), y0 = wrap(iy, H), y1 = wrap(iy + 1, H);
    auto ld = [&](int x, int y) -> f4 {
        const uint8_t* p = tex + ((size_t)y * W + x) * 4;
        return { (float)p[0], (float)p[1], (float)p[2], (float)p[3] };
    };
    return blend4(ld(x0, y0), ld(x1, y0), ld(x0, y1), ld(x1, y1), wx, wy);
}

// Texture2D.Sample / SampleBias with the quad derivatives ddx, ddy (already in uv units)
static inline f4 sample_material_tex(const vqhip_texture2d& t, f2 uv, f2 ddx, f2 ddy, float bias) {
    if (!t.texels) return { 0, 0, 0, 0 };                                   // null SRV
    const float W = (float)t.width, H = (float)t.height;
    const f2 dX = { ddx.x * W, ddx.y * H }, dY = { ddy.x * W, ddy.y * H };
    const float rx = fma_(dX.y, dX.y, dX.x * dX.x), ry = fma_(dY.y, dY.y, dY.x * dY.x);
    const float lod = 0.5f * log2_(max_(rx, ry)) + bias;
    const float maxl = (float)(t.mips - 1);
    const float l = (lod > 0.0f) ? ((lod < maxl) ? lod : maxl) : 0.0f;      // NaN, -inf -> 0
    const int fl = f2i_floor(l * 256.0f + 0.5f);
    int lo = fl >> 8;
    float f = (float)(fl & 255) * 0.00390625f;
    if (lo >= t.mips - 1) { lo = t.mips - 1; f = 0.0f; }
    const uint8_t* base = (const uint8_t*)t.texels;
    const f4 a = sample_2d_rgba8_wrap(base + tex_level_offset_px(t.width, t.height, lo) * 4, mip_dim(t.width, lo), mip_dim(t.height, lo), uv.x, uv.y);
    f4 r = a;
    if (f != 0.0f) {
        const f4 b = sample_2d_rgba8_wrap(base + tex_level_offset_px(t.width, t.height, lo + 1) * 4, mip_dim(t.width, lo + 1), mip_dim(t.height, lo + 1), uv.x, uv.y);
        const float g = 1.0f - f;
        r = { fma_(f, b.x, g * a.x), fma_(f, b.y, g * a.y), fma_(f, b.z, g * a.z), fma_(f, b.w, g * a.w) };
    }
    return { unorm8_to_float(r.x), unorm8_to_float(r.y), unorm8_to_float(r.z), unorm8_to_float(r.w) };
}

// texScreenSpaceAO.Sample(PointSampler, uv) of an R8_UNORM image (ForwardLighting.hlsl:280-281; POINT_WRAP): texel = floor of
// the coordinate snapped to 8 fractional bits (D3D11.3 §7.18.7) — the shader's coordinates (x+1)/W sit exactly on texel
// borders, the snap makes the choice (texel x+1, wrapping at the right/bottom edge) rounding-proof
static inline float fetch_r8_point_wrap(const uint8_t* tex, int W, int H, float u, float v) {
    auto wrap = [](int i, int n) { int q = i % n; return q < 0 ? q + n : q; };
    const int tx = wrap(f2i_floor((u * (float)W) * 256.0f + 0.5f) >> 8, W);
    const int ty = wrap(f2i_floor((v * (float)H) * 256.0f + 0.5f) >> 8, H);
    return unorm8_to_float((float)tex[(size_t)ty * W + tx]);
}
// point-sampled (POINT_WRAP) fetch of one R32F 2D array slice: texel = floor(uv * dim) wrapped
static inline float fetch_point_wrap(const float* slice, int dim, float u, float v) {
    int x = f2i_floor(u * (float)dim), y = f2i_floor(v * (float)dim);
    x %= dim; if (x < 0) x += dim; y %= dim; if (y < 0) y += dim;
    return slice[(size_t)y * dim + x];
}
// point-sampled fetch of an R32F cube [6][dim][dim]
static inline float fetch_cube_point(const float* cube, int dim, f3 dir) {
    float su, sv; int f = cube_face_uv(dir, &su, &sv);
    int x = f2i_floor(su * (float)dim), y = f2i_floor(sv * (float)dim);
    x = x < 0 ? 0 : (x > dim - 1 ? dim - 1 : x); y = y < 0 ? 0 : (y > dim - 1 ? dim - 1 : y);
    return cube[((size_t)f * dim + y) * dim + x];
}

} // namespace vqo
#endif
