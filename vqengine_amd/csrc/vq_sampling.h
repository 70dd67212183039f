// vq_sampling.h — software texture filtering for the vqhip kernels (no texture units on this path:
// the inputs are linear buffers, so D3D's sampler behaviour is restated in ALU).
//
// Contract (DESIGN.md §"Sampling contract"): 8-bit fixed-point filter fractions, bilinear as the FMA
// chain fma(w11,c11, fma(w01,c01, fma(w10,c10, w00*c00))) with exact weights, trilinear =
// fma(f, hi, (1-f)*lo) with an 8-bit f, seamless cube edges via the adjacent face's edge texel,
// cube corners = mean of the three remaining taps.
// Samplers restated: ClampedLinearSampler (TRILINEAR_CLAMP, Source/Renderer/Pipeline/RootSignatures.cpp:150),
// PointSampler (POINT_WRAP, :148), convolution sampler (TRILINEAR_WRAP, :402).
#pragma once
#include "vq_devmath.h"

namespace vqd {

// ---- cube geometry: faces +X,-X,+Y,-Y,+Z,-Z (CubemapUtility.h:26-36), LookAtLH bases of
// CubemapUtility.cpp:40-49: F forward, U up, R = cross(U,F). Packed as 2-bit signed components.
// rows: F, U, R ; columns x,y,z
__device__ const int8_t kFaceBasis[6][9] = {
    {  1, 0, 0,   0, 1, 0,   0, 0,-1 },
    { -1, 0, 0,   0, 1, 0,   0, 0, 1 },
    {  0, 1, 0,   0, 0,-1,   1, 0, 0 },
    {  0,-1, 0,   0, 0, 1,   1, 0, 0 },
    {  0, 0, 1,   0, 1, 0,   1, 0, 0 },
    {  0, 0,-1,   0, 1, 0,  -1, 0, 0 } };
VQD void face_basis(int f, int F[3], int U[3], int R[3]) {
    for (int k = 0; k < 3; ++k) { F[k] = kFaceBasis[f][k]; U[k] = kFaceBasis[f][3 + k]; R[k] = kFaceBasis[f][6 + k]; }
}

// texel centre (x,y) of a res^2 face -> direction F + u*R + v*U, u = 2(x+.5)/res - 1, v = 1 - 2(y+.5)/res
// (cube mesh [-1,1]^3, MeshGenerator.h:227-250; VSMain_PerFace CubemapConvolution.hlsl:63-73)
VQD f3 cube_texel_dir(int face, int x, int y, int res) {
    float inv = rcp((float)res);
    float u = (2.0f * ((float)x + 0.5f)) * inv - 1.0f;
    float v = 1.0f - (2.0f * ((float)y + 0.5f)) * inv;
    switch (face) {
        case 0:  return mk3( 1.0f,  v, -u);
        case 1:  return mk3(-1.0f,  v,  u);
        case 2:  return mk3( u,  1.0f, -v);
        case 3:  return mk3( u, -1.0f,  v);
        case 4:  return mk3( u,  v,  1.0f);
        default: return mk3(-u,  v, -1.0f);
    }
}

// direction -> face and [0,1] face coordinates (sv grows down the rows); ties: z over y over x
VQD int cube_face_uv(f3 d, float* su, float* sv) {
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

// tap (i,j) outside face f through ONE edge -> (face, i, j) of the texel across that edge.
// Works on the integer lattice of texel centres scaled by N (centre i <-> 2i+1-N).
VQD void cube_edge_neighbor(int f, int i, int j, int N, int* nf, int* ni, int* nj) {
    int F[3], U[3], R[3];
    face_basis(f, F, U, R);
    int a = 2 * i + 1 - N, b = N - (2 * j + 1);
    int ax[3];
    if (i < 0)       { a = -N; for (int k = 0; k < 3; ++k) ax[k] = -R[k]; }
    else if (i >= N) { a =  N; for (int k = 0; k < 3; ++k) ax[k] =  R[k]; }
    else if (j < 0)  { b =  N; for (int k = 0; k < 3; ++k) ax[k] =  U[k]; }
    else             { b = -N; for (int k = 0; k < 3; ++k) ax[k] = -U[k]; }
    int q[3];
    for (int k = 0; k < 3; ++k) q[k] = N * F[k] + a * R[k] + b * U[k];
    // the face whose forward axis is ax: +X=0,-X=1,+Y=2,-Y=3,+Z=4,-Z=5
    int g = ax[0] ? (ax[0] > 0 ? 0 : 1) : ax[1] ? (ax[1] > 0 ? 2 : 3) : (ax[2] > 0 ? 4 : 5);
    int F2[3], U2[3], R2[3];
    face_basis(g, F2, U2, R2);
    int a2 = q[0] * R2[0] + q[1] * R2[1] + q[2] * R2[2];
    int b2 = q[0] * U2[0] + q[1] * U2[1] + q[2] * U2[2];
    *nf = g;
    *ni = (a2 >= N) ? N - 1 : (a2 <= -N) ? 0 : (a2 + N - 1) / 2;
    *nj = (b2 >= N) ? 0 : (b2 <= -N) ? N - 1 : (N - 1 - b2) / 2;
}

VQD void fixed8(float x, int* ix, float* w) {
    int fx = f2i_floor(x * 256.0f + 0.5f);
    *ix = fx >> 8;
    *w = (float)(fx & 255) * 0.00390625f;
}

VQD float4 blend4(float4 c00, float4 c10, float4 c01, float4 c11, float wx, float wy) {
    float w00 = (1.0f - wx) * (1.0f - wy), w10 = wx * (1.0f - wy), w01 = (1.0f - wx) * wy, w11 = wx * wy;
    float4 r;
    r.x = fma_(w11, c11.x, fma_(w01, c01.x, fma_(w10, c10.x, w00 * c00.x)));
    r.y = fma_(w11, c11.y, fma_(w01, c01.y, fma_(w10, c10.y, w00 * c00.y)));
    r.z = fma_(w11, c11.z, fma_(w01, c01.z, fma_(w10, c10.z, w00 * c00.z)));
    r.w = fma_(w11, c11.w, fma_(w01, c01.w, fma_(w10, c10.w, w00 * c00.w)));
    return r;
}

// seamless bilinear fetch from one mip [6][N][N] RGBA16F of a cube
VQD float4 sample_cube_rgba16f(const void* cube, int N, f3 dir) {
    float su, sv;
    int f = cube_face_uv(dir, &su, &sv);
    int ix, iy; float wx, wy;
    fixed8(su * (float)N - 0.5f, &ix, &wx);
    fixed8(sv * (float)N - 0.5f, &iy, &wy);
    float4 c[4];
    if (ix >= 0 && iy >= 0 && ix + 1 < N && iy + 1 < N) {            // interior footprint: the common case
        size_t base = ((size_t)f * N + iy) * N + ix;
        c[0] = load_rgba16f(cube, base);     c[1] = load_rgba16f(cube, base + 1);
        c[2] = load_rgba16f(cube, base + N); c[3] = load_rgba16f(cube, base + N + 1);
    } else {
        int missing = -1;
        for (int t = 0; t < 4; ++t) {
            int i = ix + (t & 1), j = iy + (t >> 1);
            bool ox = (i < 0 || i >= N), oy = (j < 0 || j >= N);
            if (!ox && !oy)      c[t] = load_rgba16f(cube, ((size_t)f * N + j) * N + i);
            else if (ox && oy) { c[t] = make_float4(0, 0, 0, 0); missing = t; }
            else { int nf, ni, nj; cube_edge_neighbor(f, i, j, N, &nf, &ni, &nj); c[t] = load_rgba16f(cube, ((size_t)nf * N + nj) * N + ni); }
        }
        if (missing >= 0) {                                           // corner: mean of the other three, in tap order
            float4 s = make_float4(0, 0, 0, 0); bool first = true;
            for (int k = 0; k < 4; ++k) {
                if (k == missing) continue;
                if (first) { s = c[k]; first = false; }
                else { s.x += c[k].x; s.y += c[k].y; s.z += c[k].z; s.w += c[k].w; }
            }
            const float third = 0.333333343267440796f;
            float4 m = make_float4(s.x * third, s.y * third, s.z * third, s.w * third);
            if (missing == 0) c[0] = m; else if (missing == 1) c[1] = m; else if (missing == 2) c[2] = m; else c[3] = m;
        }
    }
    return blend4(c[0], c[1], c[2], c[3], wx, wy);
}

// bilinear CLAMP fetch of an RG16F [H][W] texture
VQD float2 sample_2d_rg16f_clamp(const void* tex, int W, int H, float u, float v) {
    int ix, iy; float wx, wy;
    fixed8(u * (float)W - 0.5f, &ix, &wx);
    fixed8(v * (float)H - 0.5f, &iy, &wy);
    int x0 = min(max(ix, 0), W - 1), x1 = min(max(ix + 1, 0), W - 1);
    int y0 = min(max(iy, 0), H - 1), y1 = min(max(iy + 1, 0), H - 1);
    const h2* t = (const h2*)tex;
    h2 a = t[(size_t)y0 * W + x0], b = t[(size_t)y0 * W + x1], c = t[(size_t)y1 * W + x0], d = t[(size_t)y1 * W + x1];
    float4 r = blend4(make_float4((float)a.x, (float)a.y, 0, 0), make_float4((float)b.x, (float)b.y, 0, 0),
                      make_float4((float)c.x, (float)c.y, 0, 0), make_float4((float)d.x, (float)d.y, 0, 0), wx, wy);
    return make_float2(r.x, r.y);
}

// ---- RGBA32F mip chain (equirect) ------------------------------------------------------------
VQD int mip_dim(int d0, int level) { int d = d0 >> level; return d < 1 ? 1 : d; }
VQD size_t mip_offset_px(int w0, int h0, int level) {
    size_t off = 0;
    for (int l = 0; l < level; ++l) off += (size_t)mip_dim(w0, l) * mip_dim(h0, l);
    return off;
}
VQD int wrapi(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }

VQD float4 sample_2d_rgba32f_wrap(const float4* tex, int W, int H, float u, float v) {
    int ix, iy; float wx, wy;
    fixed8(u * (float)W - 0.5f, &ix, &wx);
    fixed8(v * (float)H - 0.5f, &iy, &wy);
    int x0 = wrapi(ix, W), x1 = wrapi(ix + 1, W), y0 = wrapi(iy, H), y1 = wrapi(iy + 1, H);
    return blend4(tex[(size_t)y0 * W + x0], tex[(size_t)y0 * W + x1], tex[(size_t)y1 * W + x0], tex[(size_t)y1 * W + x1], wx, wy);
}

// SampleLevel(uv, lod) with TRILINEAR_WRAP on a dense chain (level 0 first)
VQD float4 sample_equirect_lod(const float4* chain, int w0, int h0, int nMips, float u, float v, float lod) {
    float maxl = (float)(nMips - 1);
    float l = (lod > 0.0f) ? ((lod < maxl) ? lod : maxl) : 0.0f;
    int fl = f2i_floor(l * 256.0f + 0.5f);
    int lo = fl >> 8;
    float f = (float)(fl & 255) * 0.00390625f;
    if (lo >= nMips - 1) { lo = nMips - 1; f = 0.0f; }
    float4 a = sample_2d_rgba32f_wrap(chain + mip_offset_px(w0, h0, lo), mip_dim(w0, lo), mip_dim(h0, lo), u, v);
    if (f == 0.0f) return a;
    float4 b = sample_2d_rgba32f_wrap(chain + mip_offset_px(w0, h0, lo + 1), mip_dim(w0, lo + 1), mip_dim(h0, lo + 1), u, v);
    float g = 1.0f - f;
    return make_float4(fma_(f, b.x, g * a.x), fma_(f, b.y, g * a.y), fma_(f, b.z, g * a.z), fma_(f, b.w, g * a.w));
}

// ---- point sampling (shadow maps) ----------------------------------------------------------------
VQD float fetch_point_wrap(const float* slice, int dim, float u, float v) {
    int x = wrapi(f2i_floor(u * (float)dim), dim), y = wrapi(f2i_floor(v * (float)dim), dim);
    return slice[(size_t)y * dim + x];
}
VQD float fetch_cube_point(const float* cube, int dim, f3 dir) {
    float su, sv; int f = cube_face_uv(dir, &su, &sv);
    int x = min(max(f2i_floor(su * (float)dim), 0), dim - 1), y = min(max(f2i_floor(sv * (float)dim), 0), dim - 1);
    return cube[((size_t)f * dim + y) * dim + x];
}

} // namespace vqd
