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
    // Branch-free form of the 6-way table (only selects and sign flips, so the values are those of the if/else ladder):
    //   face  +X(0)  -X(1)  +Y(2)  -Y(3)  +Z(4)  -Z(5)
    //   sc     -z     +z     +x     +x     +x     -x
    //   tc     -y     -y     +z     -z     -y     -y
    const float ax = abs_(d.x), ay = abs_(d.y), az = abs_(d.z);
    const bool isZ = (az >= ax) & (az >= ay);
    const bool isY = !isZ & (ay >= ax);
    const bool isX = !(isZ | isY);
    const float major = isZ ? d.z : (isY ? d.y : d.x);
    const float ma = isZ ? az : (isY ? ay : ax);
    const bool neg = major < 0.0f;
    const float sc0 = isX ? -d.z : d.x;
    const float tc0 = isY ? d.z : -d.y;
    const float sc = (neg & !isY) ? -sc0 : sc0;
    const float tc = (neg & isY) ? -tc0 : tc0;
    const int face = (isZ ? 4 : (isY ? 2 : 0)) + (neg ? 1 : 0);
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

// The same adjacency as a 24-entry table, derived at COMPILE TIME from the face bases by probing the lattice construction
// above with two taps per (face, exit edge): along an edge the neighbour's texel runs linearly with slope +-1 while its other
// coordinate sits on the neighbour's border row/column. Entry bits: [2:0] neighbour face, [3] ni is the fixed coordinate
// (nj runs), [4] fixed coordinate = N-1 (else 0), [5] running coordinate is flipped (N-1-r). Exit edges: 0 i<0, 1 i>=N, 2 j<0, 3 j>=N.
namespace edge_tab {
constexpr int8_t kB[6][9] = {
    {  1, 0, 0,   0, 1, 0,   0, 0,-1 }, { -1, 0, 0,   0, 1, 0,   0, 0, 1 }, {  0, 1, 0,   0, 0,-1,   1, 0, 0 },
    {  0,-1, 0,   0, 0, 1,   1, 0, 0 }, {  0, 0, 1,   0, 1, 0,   1, 0, 0 }, {  0, 0,-1,   0, 1, 0,  -1, 0, 0 } };
struct NB { int f, i, j; };
constexpr NB neighbor(int f, int i, int j, int N) {
    int a = 2 * i + 1 - N, b = N - (2 * j + 1);
    int ax[3] = { 0, 0, 0 };
    if (i < 0)       { a = -N; for (int k = 0; k < 3; ++k) ax[k] = -kB[f][6 + k]; }
    else if (i >= N) { a =  N; for (int k = 0; k < 3; ++k) ax[k] =  kB[f][6 + k]; }
    else if (j < 0)  { b =  N; for (int k = 0; k < 3; ++k) ax[k] =  kB[f][3 + k]; }
    else             { b = -N; for (int k = 0; k < 3; ++k) ax[k] = -kB[f][3 + k]; }
    int q[3] = { 0, 0, 0 };
    for (int k = 0; k < 3; ++k) q[k] = N * kB[f][k] + a * kB[f][6 + k] + b * kB[f][3 + k];
    const int g = ax[0] ? (ax[0] > 0 ? 0 : 1) : ax[1] ? (ax[1] > 0 ? 2 : 3) : (ax[2] > 0 ? 4 : 5);
    const int a2 = q[0] * kB[g][6] + q[1] * kB[g][7] + q[2] * kB[g][8];
    const int b2 = q[0] * kB[g][3] + q[1] * kB[g][4] + q[2] * kB[g][5];
    NB r = { g, (a2 >= N) ? N - 1 : (a2 <= -N) ? 0 : (a2 + N - 1) / 2, (b2 >= N) ? 0 : (b2 <= -N) ? N - 1 : (N - 1 - b2) / 2 };
    return r;
}
constexpr uint32_t entry(int f, int e) {
    const int N = 4;
    const int i0 = e == 0 ? -1 : e == 1 ? N : 0, j0 = e == 2 ? -1 : e == 3 ? N : 0;
    const int i1 = e < 2 ? i0 : 1, j1 = e < 2 ? 1 : j0;
    const NB p = neighbor(f, i0, j0, N), q = neighbor(f, i1, j1, N);
    const bool fixedIsI = (p.i == q.i);
    const int fixedV = fixedIsI ? p.i : p.j, run0 = fixedIsI ? p.j : p.i;
    return (uint32_t)p.f | ((uint32_t)fixedIsI << 3) | ((uint32_t)(fixedV != 0) << 4) | ((uint32_t)(run0 != 0) << 5);
}
// 24 six-bit entries packed into three 64-bit words (8 entries each), indexed by f*4 + e
constexpr uint64_t word(int w) {
    uint64_t r = 0;
    for (int k = 0; k < 8; ++k) { const int idx = w * 8 + k; r |= (uint64_t)entry(idx >> 2, idx & 3) << (8 * k); }
    return r;
}
constexpr uint64_t kW0 = word(0), kW1 = word(1), kW2 = word(2);
} // namespace edge_tab

// tap (i,j) of face f that left the face through exactly ONE edge -> texel (nf, ni, nj) of the adjacent face (table form of
// cube_edge_neighbor; ~12 integer operations)
VQD uint32_t cube_edge_entry(int f, int e) {
    const int idx = f * 4 + e;
    const uint64_t w = idx < 8 ? edge_tab::kW0 : (idx < 16 ? edge_tab::kW1 : edge_tab::kW2);
    return (uint32_t)(w >> (8 * (idx & 7))) & 63u;
}
// r = the tap's coordinate ALONG the edge it left through (j for an x edge, i for a y edge)
VQD void cube_edge_apply(uint32_t en, int r, int N, int* nf, int* ni, int* nj) {
    const int run = (en & 32u) ? N - 1 - r : r;
    const int fix = (en & 16u) ? N - 1 : 0;
    *nf = (int)(en & 7u);
    *ni = (en & 8u) ? fix : run;
    *nj = (en & 8u) ? run : fix;
}
VQD void cube_edge_lookup(int f, int i, int j, int N, int* nf, int* ni, int* nj) {
    const bool ox = (i < 0) | (i >= N);
    cube_edge_apply(cube_edge_entry(f, ox ? (i < 0 ? 0 : 1) : (j < 0 ? 2 : 3)), ox ? j : i, N, nf, ni, nj);
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
    // One branch-free tap path for interior and edge footprints alike: on the 64^2 / <=128^2 cubes of the reference nearly
    // every wave holds a lane whose footprint crosses a face edge, so a separate interior fast path only ever ran IN ADDITION
    // to the general one. A tap outside through ONE edge comes from the adjacent face (table lookup, selected by predicate);
    // a tap outside through a corner is "missing" (rare: handled in the only divergent branch). (An interior fast path in front of it was measured:
    // +1.6 % on coherent content, -0.8 % on the BASELINE frame, profiles/r4*; not in the source.)
    float4 c[4];
    int missing = -1;
    // the two columns of a footprint can only leave the face on the same side (ix < 0: left, else right), likewise the rows:
    // two table entries per sample serve all four taps
    const uint32_t enx = cube_edge_entry(f, ix < 0 ? 0 : 1), eny = cube_edge_entry(f, iy < 0 ? 2 : 3);
    uint32_t idx[4];
    #pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int i = ix + (t & 1), j = iy + (t >> 1);
        const bool ox = (i < 0) | (i >= N), oy = (j < 0) | (j >= N);
        int nf, ni, nj;
        cube_edge_apply(ox ? enx : eny, ox ? j : i, N, &nf, &ni, &nj); // meaningful only when exactly one of ox, oy holds
        const bool one = ox != oy;
        nf = one ? nf : f;
        ni = one ? ni : min(max(i, 0), N - 1);                        // identity inside the face; any valid address for a corner
        nj = one ? nj : min(max(j, 0), N - 1);
        if (ox & oy) missing = t;
        idx[t] = (uint32_t)((nf * N + nj) * N + ni);
    }
    // Round 6: the two texels of a footprint row are neighbours in memory unless the row crosses a face edge — then they are ONE 16-byte load (8-byte aligned: the hardware's
    // unaligned-access mode) instead of two 8-byte gathers. On incoherent directions every lane's gather is its own cache line, and the pass costs L1 tag lookups, not bytes.
    typedef _Float16 h8u __attribute__((ext_vector_type(8)));
    struct __attribute__((packed, aligned(8))) H8 { h8u v; };
    #pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t a = idx[2 * r], b = idx[2 * r + 1];
        if (b == a + 1u) {
            const h8u v = ((const H8*)((const h4*)cube + a))->v;
            c[2 * r] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
            c[2 * r + 1] = make_float4((float)v[4], (float)v[5], (float)v[6], (float)v[7]);
        } else {
            c[2 * r] = load_rgba16f(cube, (size_t)a);
            c[2 * r + 1] = load_rgba16f(cube, (size_t)b);
        }
    }
    if (missing >= 0) {                                               // corner: mean of the other three, in tap order
        const float4 a = missing == 0 ? c[1] : c[0], b = missing <= 1 ? c[2] : c[1], d = missing <= 2 ? c[3] : c[2];
        const float third = 0.333333343267440796f;
        const float4 m = make_float4(((a.x + b.x) + d.x) * third, ((a.y + b.y) + d.y) * third, ((a.z + b.z) + d.z) * third, ((a.w + b.w) + d.w) * third);
        if (missing == 0) c[0] = m; else if (missing == 1) c[1] = m; else if (missing == 2) c[2] = m; else c[3] = m;
    }
    return blend4(c[0], c[1], c[2], c[3], wx, wy);
}

// SampleLevel(dir, lod) on a mip-major RGBA16F cube ([mip][6][r][r], r = res0 >> mip) with a MIN_MAG_MIP_LINEAR sampler (CLAMP addressing has no
// effect on a cube: edges are seamless): the sampler clamps lod to [0, nMips-1] (NaN -> 0), the level fraction is an 8-bit fixed-point value
// like the in-level filter fractions — floor(lod * 256 + 0.5) / 256, the rule of sample_equirect_lod_t below and of D3D11.3 functional spec
// 7.18.10 "LOD fraction: at least 8 bits" — and the two seamless bilinear fetches blend as fma(f, hi, (1 - f) * lo); a zero fraction reads one
// level only. Used by SSR's environment fallback (ClassifyReflectionTiles.hlsl:91: roughness * (mip_count - 1) is fractional);
// tests/ref64_sampling.py bounds it against exact float64 trilinear weights.
VQD uint32_t cube_mip_offset_texels(int res0, int mip) {
    if ((res0 & (res0 - 1)) == 0) { const int rm = res0 >> mip; return 8u * (uint32_t)(res0 * res0 - rm * rm); }   // 6 * sum_{m<mip} 4^-m res0^2
    uint32_t off = 0;
    for (int m = 0; m < mip; ++m) { const uint32_t r = (uint32_t)(res0 >> m); off += 6u * r * r; }
    return off;
}
VQD float4 sample_cube_lod_rgba16f(const void* cube, int res0, int nMips, f3 dir, float lod) {
    const float maxl = (float)(nMips - 1);
    const float l = (lod > 0.0f) ? ((lod < maxl) ? lod : maxl) : 0.0f;
    const int fl = f2i_floor(l * 256.0f + 0.5f);
    int lo = fl >> 8;
    float f = (float)(fl & 255) * 0.00390625f;
    if (lo >= nMips - 1) { lo = nMips - 1; f = 0.0f; }
    const float4 a = sample_cube_rgba16f((const h4*)cube + cube_mip_offset_texels(res0, lo), res0 >> lo, dir);
    if (f == 0.0f) return a;
    const float4 b = sample_cube_rgba16f((const h4*)cube + cube_mip_offset_texels(res0, lo + 1), res0 >> (lo + 1), dir);
    const float g = 1.0f - f;
    return make_float4(fma_(f, b.x, g * a.x), fma_(f, b.y, g * a.y), fma_(f, b.z, g * a.z), fma_(f, b.w, g * a.w));
}

// bilinear CLAMP fetch of an RG16F [H][W] texture
VQD float2 sample_2d_rg16f_clamp(const void* tex, int W, int H, float u, float v) {
    int ix, iy; float wx, wy;
    fixed8(u * (float)W - 0.5f, &ix, &wx);
    fixed8(v * (float)H - 0.5f, &iy, &wy);
    int x0 = min(max(ix, 0), W - 1), x1 = min(max(ix + 1, 0), W - 1);
    int y0 = min(max(iy, 0), H - 1), y1 = min(max(iy + 1, 0), H - 1);
    const h2* t = (const h2*)tex;
    h2 a, b, c, d;
    if (x1 == x0 + 1) {                                        // not clamped at a border: the two texels of a row as one 8-byte load (see sample_cube_rgba16f)
        typedef _Float16 h4u __attribute__((ext_vector_type(4)));
        struct __attribute__((packed, aligned(4))) H4 { h4u v; };
        const h4u r0 = ((const H4*)(t + (size_t)y0 * W + x0))->v, r1 = ((const H4*)(t + (size_t)y1 * W + x0))->v;
        a.x = r0[0]; a.y = r0[1]; b.x = r0[2]; b.y = r0[3]; c.x = r1[0]; c.y = r1[1]; d.x = r1[2]; d.y = r1[3];
    } else {
        a = t[(size_t)y0 * W + x0]; b = t[(size_t)y0 * W + x1]; c = t[(size_t)y1 * W + x0]; d = t[(size_t)y1 * W + x1];
    }
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

// POT = both dimensions are powers of two (every level then is, too): WRAP is an AND instead of an integer modulo (~20 VALU each)
template <bool POT> VQD float4 sample_2d_rgba32f_wrap_t(const float4* tex, int W, int H, float u, float v) {
    int ix, iy; float wx, wy;
    fixed8(u * (float)W - 0.5f, &ix, &wx);
    fixed8(v * (float)H - 0.5f, &iy, &wy);
    int x0, x1, y0, y1;
    if (POT) { x0 = ix & (W - 1); x1 = (ix + 1) & (W - 1); y0 = iy & (H - 1); y1 = (iy + 1) & (H - 1); }
    else     { x0 = wrapi(ix, W); x1 = wrapi(ix + 1, W); y0 = wrapi(iy, H); y1 = wrapi(iy + 1, H); }
    const uint32_t r0 = __umul24(y0, W), r1 = __umul24(y1, W);       // level dimensions are < 2^24
    return blend4(tex[r0 + (uint32_t)x0], tex[r0 + (uint32_t)x1], tex[r1 + (uint32_t)x0], tex[r1 + (uint32_t)x1], wx, wy);
}
VQD bool is_pot2(int w, int h) { return ((w & (w - 1)) | (h & (h - 1))) == 0; }
VQD float4 sample_2d_rgba32f_wrap(const float4* tex, int W, int H, float u, float v) {
    return is_pot2(W, H) ? sample_2d_rgba32f_wrap_t<true>(tex, W, H, u, v) : sample_2d_rgba32f_wrap_t<false>(tex, W, H, u, v);
}

// offset (in texels) of level l of a dense chain whose level k is max(1,w>>k) x max(1,h>>k). Power-of-two sizes in closed form:
// levels 0..m (m = log2 of the short side) shrink by 4, the tail (short side clamped to 1) by 2:
//   sum_{k<l1} wh/4^k = (4wh - wh/4^(l1-1))/3 (exact; /3 by the inverse of 3 mod 2^32), sum_{k=m+1}^{l-1} L>>k = (L>>m) - (L>>(l-1))
template <bool POT> VQD uint32_t chain_level_offset(int w, int h, int l) {
    if (POT) {
        const int m = min(31 - __builtin_clz(w), 31 - __builtin_clz(h)), L = max(w, h);
        const int l1 = min(l, m + 1);
        const uint32_t wh = (uint32_t)w * (uint32_t)h;
        uint32_t off = l1 >= 1 ? (4u * wh - (wh >> (2 * l1 - 2))) * 0xAAAAAAABu : 0u;
        if (l > m + 1) off += (uint32_t)((L >> m) - (L >> (l - 1)));
        return off;
    }
    uint32_t off = 0;
    for (int k = 0; k < l; ++k) off += (uint32_t)mip_dim(w, k) * (uint32_t)mip_dim(h, k);
    return off;
}

// SampleLevel(uv, lod) with TRILINEAR_WRAP on a dense chain (level 0 first)
template <bool POT> VQD float4 sample_equirect_lod_t(const float4* chain, int w0, int h0, int nMips, float u, float v, float lod) {
    float maxl = (float)(nMips - 1);
    float l = (lod > 0.0f) ? ((lod < maxl) ? lod : maxl) : 0.0f;
    int fl = f2i_floor(l * 256.0f + 0.5f);
    int lo = fl >> 8;
    float f = (float)(fl & 255) * 0.00390625f;
    if (lo >= nMips - 1) { lo = nMips - 1; f = 0.0f; }
    float4 a = sample_2d_rgba32f_wrap_t<POT>(chain + chain_level_offset<POT>(w0, h0, lo), mip_dim(w0, lo), mip_dim(h0, lo), u, v);
    if (f == 0.0f) return a;
    float4 b = sample_2d_rgba32f_wrap_t<POT>(chain + chain_level_offset<POT>(w0, h0, lo + 1), mip_dim(w0, lo + 1), mip_dim(h0, lo + 1), u, v);
    float g = 1.0f - f;
    return make_float4(fma_(f, b.x, g * a.x), fma_(f, b.y, g * a.y), fma_(f, b.z, g * a.z), fma_(f, b.w, g * a.w));
}
VQD float4 sample_equirect_lod(const float4* chain, int w0, int h0, int nMips, float u, float v, float lod) {
    return is_pot2(w0, h0) ? sample_equirect_lod_t<true>(chain, w0, h0, nMips, u, v, lod) : sample_equirect_lod_t<false>(chain, w0, h0, nMips, u, v, lod);
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
