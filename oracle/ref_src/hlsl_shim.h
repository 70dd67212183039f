// hlsl_shim.h — the subset of HLSL's types and intrinsics that the reference's shaders on this path use, as C++17,
// so that oracle/ref_src/hlsl2cpp.py's output (the reference's OWN shader sources, read where they lie under
// /root/reference/Shaders and rewritten only syntactically) compiles with g++ into oracle/_ref/libvqref_shaders.so.
//
// TEST INFRASTRUCTURE (oracle side). Arithmetic model of the shim = "HLSL as written, IEEE binary32, no fast-math":
//   * every operator is one correctly rounded binary32 operation, evaluated in source order (-ffp-contract=off);
//   * dot(a,b) = a.x*b.x + a.y*b.y + ... left to right; length = sqrt(dot); normalize(v) = v / length(v) per component;
//   * pow(x,y) = exp2(y*log2(x)) (how DXC lowers it; negative bases give NaN like on a GPU), exp2/log2/sin/... = libm;
//   * saturate/min/max drop NaN like the DXIL ops (fmin/fmax); lerp(a,b,t) = a + t*(b-a); reflect(i,n) = i - 2*dot(n,i)*n.
// What a GPU compiler does on top of that (fast-math regrouping, approximate transcendentals) is exactly what the
// oracle's arithmetic contract pins down by choice; the comparison of the two is therefore made with a tolerance
// (tests/test_ref_pinning.py), and what it pins is the ALGORITHM: constants, branches, operand order, loop bounds.
// Texture sampling has no source in the reference (fixed-function hardware): the Texture* objects below forward to hooks
// that the harness (ref_shaders.cpp) implements with the oracle's sampling contract.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#ifdef VQ_SHIM_DXC
#include "../vqo_math.h"      // the contract's exp2_ / log2_ for the DXC reading of pow
#endif

namespace hlsl {

typedef unsigned int uint;
typedef float half;

struct float2; struct float3; struct float4;

// ---- swizzle proxies (live inside the vector's union; N = components of the owning vector) ----------------------------
template <int N, int A, int B> struct sw2 {
    float d[N];
    operator float2() const;
    sw2& operator=(const float2& v);
    sw2& operator=(const sw2& o);
};
template <int N, int A, int B, int C> struct sw3 {
    float d[N];
    operator float3() const;
    sw3& operator=(const float3& v);
    sw3& operator=(const sw3& o);
    sw3& operator+=(const float3& v);
    sw3& operator/=(float s) { d[A] /= s; d[B] /= s; d[C] /= s; return *this; }
};
template <int N, int A, int B, int C, int D> struct sw4 {
    float d[N];
    operator float4() const;
};

struct float2 {
    union {
        struct { float x, y; };
        struct { float r, g; };
        float d[2];
        sw2<2, 0, 1> xy; sw2<2, 1, 0> yx; sw2<2, 0, 0> xx; sw2<2, 1, 1> yy; sw2<2, 0, 1> rg;
        sw3<2, 0, 0, 0> xxx; sw3<2, 1, 1, 1> yyy;
    };
    float2() : d{ 0, 0 } {}
    float2(float a, float b) : d{ a, b } {}
    explicit float2(float s) : d{ s, s } {}
    float2(const float2& o) : d{ o.d[0], o.d[1] } {}
    float2& operator=(const float2& o) { d[0] = o.d[0]; d[1] = o.d[1]; return *this; }
    float& operator[](int i) { return d[i]; }
    const float& operator[](int i) const { return d[i]; }
};
struct float3 {
    union {
        struct { float x, y, z; };
        struct { float r, g, b; };
        float d[3];
        sw2<3, 0, 1> xy; sw2<3, 0, 2> xz; sw2<3, 1, 2> yz; sw2<3, 0, 1> rg;
        sw3<3, 0, 1, 2> xyz; sw3<3, 0, 1, 2> rgb; sw3<3, 0, 0, 0> xxx; sw3<3, 0, 0, 0> rrr; sw3<3, 2, 1, 0> zyx; sw3<3, 1, 1, 1> yyy; sw3<3, 2, 2, 2> zzz;
        sw2<3, 0, 0> xx; sw2<3, 1, 1> yy; sw2<3, 2, 2> zz;
    };
    float3() : d{ 0, 0, 0 } {}
    float3(float a, float b, float c) : d{ a, b, c } {}
    explicit float3(float s) : d{ s, s, s } {}
    float3(const float2& v, float c) : d{ v.d[0], v.d[1], c } {}
    float3(const float3& o) : d{ o.d[0], o.d[1], o.d[2] } {}
    float3& operator=(const float3& o) { d[0] = o.d[0]; d[1] = o.d[1]; d[2] = o.d[2]; return *this; }
    float& operator[](int i) { return d[i]; }
    const float& operator[](int i) const { return d[i]; }
};
struct float4 {
    union {
        struct { float x, y, z, w; };
        struct { float r, g, b, a; };
        float d[4];
        sw2<4, 0, 1> xy; sw2<4, 2, 3> zw; sw2<4, 0, 1> rg; sw2<4, 0, 2> xz;
        sw3<4, 0, 1, 2> xyz; sw3<4, 0, 1, 2> rgb; sw3<4, 0, 0, 0> xxx; sw3<4, 0, 0, 0> rrr; sw3<4, 3, 3, 3> aaa; sw3<4, 3, 3, 3> www;
        sw4<4, 0, 1, 2, 3> xyzw; sw4<4, 0, 1, 2, 3> rgba; sw4<4, 0, 1, 3, 3> xyww;
    };
    float4() : d{ 0, 0, 0, 0 } {}
    float4(float a, float b, float c, float e) : d{ a, b, c, e } {}
    explicit float4(float s) : d{ s, s, s, s } {}
    float4(const float3& v, float e) : d{ v.d[0], v.d[1], v.d[2], e } {}
    float4(const float2& v, float c, float e) : d{ v.d[0], v.d[1], c, e } {}
    float4(const float2& a, const float2& b) : d{ a.d[0], a.d[1], b.d[0], b.d[1] } {}
    float4(const float4& o) : d{ o.d[0], o.d[1], o.d[2], o.d[3] } {}
    float4& operator=(const float4& o) { for (int i = 0; i < 4; ++i) d[i] = o.d[i]; return *this; }
    float& operator[](int i) { return d[i]; }
    const float& operator[](int i) const { return d[i]; }
};
typedef float2 half2; typedef float3 half3; typedef float4 half4;

template <int N, int A, int B> sw2<N, A, B>::operator float2() const { return float2(d[A], d[B]); }
template <int N, int A, int B> sw2<N, A, B>& sw2<N, A, B>::operator=(const float2& v) { d[A] = v.x; d[B] = v.y; return *this; }
template <int N, int A, int B> sw2<N, A, B>& sw2<N, A, B>::operator=(const sw2& o) { const float2 v = o; return *this = v; }
template <int N, int A, int B, int C> sw3<N, A, B, C>::operator float3() const { return float3(d[A], d[B], d[C]); }
template <int N, int A, int B, int C> sw3<N, A, B, C>& sw3<N, A, B, C>::operator=(const float3& v) { d[A] = v.x; d[B] = v.y; d[C] = v.z; return *this; }
template <int N, int A, int B, int C> sw3<N, A, B, C>& sw3<N, A, B, C>::operator=(const sw3& o) { const float3 v = o; return *this = v; }
template <int N, int A, int B, int C> sw3<N, A, B, C>& sw3<N, A, B, C>::operator+=(const float3& v) { d[A] += v.x; d[B] += v.y; d[C] += v.z; return *this; }
template <int N, int A, int B, int C, int D> sw4<N, A, B, C, D>::operator float4() const { return float4(d[A], d[B], d[C], d[D]); }

// ---- integer vectors (plain; only the members the shaders touch) ------------------------------------------------------
struct int2; struct int3; struct int4; struct uint2; struct uint3; struct uint4;
template <class V, class T, int N, int A, int B> struct isw2 { T d[N]; operator V() const { return V(d[A], d[B]); } };
struct int2 {
    union { struct { int x, y; }; int d[2]; isw2<int2, int, 2, 0, 1> xy; };
    int2() : d{ 0, 0 } {}
    int& operator[](int i) { return d[i]; }
    const int& operator[](int i) const { return d[i]; }
    int2(int a, int b) : d{ a, b } {}
    explicit int2(int s) : d{ s, s } {}
    explicit int2(const uint2& o); explicit int2(const float2& o);
    explicit operator float2() const { return float2((float)d[0], (float)d[1]); }
};
struct int3 {
    union { struct { int x, y, z; }; int d[3]; isw2<int2, int, 3, 0, 1> xy; };
    int3() : d{ 0, 0, 0 } {}
    int& operator[](int i) { return d[i]; }
    const int& operator[](int i) const { return d[i]; }
    int3(int a, int b, int c) : d{ a, b, c } {}
    int3(const int2& v, int c) : d{ v.d[0], v.d[1], c } {}
    int3(const uint2& v, int c);
    explicit int3(int s) : d{ s, s, s } {}
    explicit int3(const uint3& o); explicit int3(const float3& o);
    explicit operator float3() const { return float3((float)d[0], (float)d[1], (float)d[2]); }
};
struct int4 {
    union { struct { int x, y, z, w; }; int d[4]; isw2<int2, int, 4, 0, 1> xy; isw2<int2, int, 4, 2, 3> zw; };
    int4() : d{ 0, 0, 0, 0 } {}
    int& operator[](int i) { return d[i]; }
    const int& operator[](int i) const { return d[i]; }
    int4(int a, int b, int c, int e) : d{ a, b, c, e } {}
    explicit int4(int s) : d{ s, s, s, s } {}
    explicit int4(const uint4& o); explicit int4(const float4& o);
    explicit operator float4() const { return float4((float)d[0], (float)d[1], (float)d[2], (float)d[3]); }
};
struct uint2 {
    union { struct { uint x, y; }; uint d[2]; isw2<uint2, uint, 2, 0, 1> xy; };
    uint2() : d{ 0, 0 } {}
    uint& operator[](int i) { return d[i]; }
    const uint& operator[](int i) const { return d[i]; }
    uint2(uint a, uint b) : d{ a, b } {}
    explicit uint2(uint s) : d{ s, s } {}
    explicit uint2(const int2& o) : d{ (uint)o.d[0], (uint)o.d[1] } {}
    explicit uint2(const float2& o);
    explicit operator float2() const { return float2((float)d[0], (float)d[1]); }
};
// DispatchThreadID.xy must convert to both uint2 and int2 (texture indexing takes either)
struct usw_xy { uint d[3]; operator uint2() const { return uint2(d[0], d[1]); } operator int2() const { return int2((int)d[0], (int)d[1]); } };
struct uint3 {
    union { struct { uint x, y, z; }; uint d[3]; usw_xy xy; };
    uint3() : d{ 0, 0, 0 } {}
    uint& operator[](int i) { return d[i]; }
    const uint& operator[](int i) const { return d[i]; }
    uint3(uint a, uint b, uint c) : d{ a, b, c } {}
    explicit uint3(uint s) : d{ s, s, s } {}
    explicit uint3(const int3& o) : d{ (uint)o.d[0], (uint)o.d[1], (uint)o.d[2] } {}
    explicit uint3(const float3& o);
    explicit operator float3() const { return float3((float)d[0], (float)d[1], (float)d[2]); }
};
struct uint4 {
    union { struct { uint x, y, z, w; }; uint d[4]; isw2<uint2, uint, 4, 0, 1> xy; isw2<uint2, uint, 4, 2, 3> zw; };
    uint4() : d{ 0, 0, 0, 0 } {}
    uint& operator[](int i) { return d[i]; }
    const uint& operator[](int i) const { return d[i]; }
    uint4(uint a, uint b, uint c, uint e) : d{ a, b, c, e } {}
    explicit uint4(uint s) : d{ s, s, s, s } {}
    explicit uint4(const int4& o) : d{ (uint)o.d[0], (uint)o.d[1], (uint)o.d[2], (uint)o.d[3] } {}
    explicit uint4(const float4& o);
    explicit operator float4() const { return float4((float)d[0], (float)d[1], (float)d[2], (float)d[3]); }
};
inline int2::int2(const uint2& o) : d{ (int)o.d[0], (int)o.d[1] } {}
inline int3::int3(const uint3& o) : d{ (int)o.d[0], (int)o.d[1], (int)o.d[2] } {}
inline int3::int3(const uint2& v, int c) : d{ (int)v.d[0], (int)v.d[1], c } {}
inline int4::int4(const uint4& o) : d{ (int)o.d[0], (int)o.d[1], (int)o.d[2], (int)o.d[3] } {}
inline int2::int2(const float2& o) : d{ (int)o.d[0], (int)o.d[1] } {}
inline int3::int3(const float3& o) : d{ (int)o.d[0], (int)o.d[1], (int)o.d[2] } {}
inline int4::int4(const float4& o) : d{ (int)o.d[0], (int)o.d[1], (int)o.d[2], (int)o.d[3] } {}
inline uint2::uint2(const float2& o) : d{ (uint)o.d[0], (uint)o.d[1] } {}
inline uint3::uint3(const float3& o) : d{ (uint)o.d[0], (uint)o.d[1], (uint)o.d[2] } {}
inline uint4::uint4(const float4& o) : d{ (uint)o.d[0], (uint)o.d[1], (uint)o.d[2], (uint)o.d[3] } {}
inline float2 to_float(const uint2& v) { return float2((float)v.x, (float)v.y); }
inline float2 to_float(const int2& v) { return float2((float)v.x, (float)v.y); }
#define VQ_HLSL_IOPS(V, T, N)                                                                                          \
    inline V operator+(const V& a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }    \
    inline V operator-(const V& a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }    \
    inline V operator*(const V& a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.d[i]; return r; }    \
    inline V operator&(const V& a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] & b.d[i]; return r; }    \
    inline V operator|(const V& a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] | b.d[i]; return r; }    \
    inline V operator^(const V& a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] ^ b.d[i]; return r; }    \
    inline V operator<<(const V& a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] << b.d[i]; return r; }  \
    inline V operator>>(const V& a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] >> b.d[i]; return r; }  \
    inline V operator+(const V& a, T b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b; return r; }                \
    inline V operator-(const V& a, T b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b; return r; }                \
    inline V operator*(const V& a, T b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b; return r; }                \
    inline V operator&(const V& a, T b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] & b; return r; }                \
    inline V operator|(const V& a, T b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] | b; return r; }                \
    inline V operator<<(const V& a, T b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] << b; return r; }              \
    inline V operator>>(const V& a, T b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] >> b; return r; }              \
    inline V operator-(T a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a - b.d[i]; return r; }                \
    inline V operator~(const V& a) { V r; for (int i = 0; i < N; ++i) r.d[i] = ~a.d[i]; return r; }                        \
    inline V min(const V& a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] < b.d[i] ? a.d[i] : b.d[i]; return r; } \
    inline V max(const V& a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] > b.d[i] ? a.d[i] : b.d[i]; return r; }
#define VQ_HLSL_ICMP(V, N) \
    inline V operator!=(const V& a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] != b.d[i]; return r; } \
    inline V operator==(const V& a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] == b.d[i]; return r; }
VQ_HLSL_ICMP(int2, 2) VQ_HLSL_ICMP(int3, 3) VQ_HLSL_ICMP(int4, 4) VQ_HLSL_ICMP(uint2, 2) VQ_HLSL_ICMP(uint3, 3) VQ_HLSL_ICMP(uint4, 4)
#undef VQ_HLSL_ICMP
inline int2 abs(const int2& a) { int2 r; for (int i = 0; i < 2; ++i) r.d[i] = a.d[i] < 0 ? -a.d[i] : a.d[i]; return r; }
inline int3 abs(const int3& a) { int3 r; for (int i = 0; i < 3; ++i) r.d[i] = a.d[i] < 0 ? -a.d[i] : a.d[i]; return r; }
inline int4 abs(const int4& a) { int4 r; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] < 0 ? -a.d[i] : a.d[i]; return r; }
#define VQ_HLSL_SEL(Cnd, V, N) inline V select(const Cnd& c, const V& a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = c.d[i] ? a.d[i] : b.d[i]; return r; }
VQ_HLSL_SEL(uint2, uint2, 2) VQ_HLSL_SEL(uint3, uint3, 3) VQ_HLSL_SEL(uint4, uint4, 4)
VQ_HLSL_SEL(uint2, float2, 2) VQ_HLSL_SEL(uint3, float3, 3) VQ_HLSL_SEL(uint4, float4, 4)
#undef VQ_HLSL_SEL
VQ_HLSL_IOPS(int2, int, 2) VQ_HLSL_IOPS(int3, int, 3) VQ_HLSL_IOPS(int4, int, 4)
VQ_HLSL_IOPS(uint2, uint, 2) VQ_HLSL_IOPS(uint3, uint, 3) VQ_HLSL_IOPS(uint4, uint, 4)
#undef VQ_HLSL_IOPS

// ---- component-wise arithmetic ------------------------------------------------------------------------------------------
#define VQ_HLSL_OPS(V, N)                                                                                           \
    inline V operator+(const V& a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; } \
    inline V operator-(const V& a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; } \
    inline V operator*(const V& a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.d[i]; return r; } \
    inline V operator/(const V& a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] / b.d[i]; return r; } \
    inline V operator+(const V& a, float b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b; return r; }         \
    inline V operator-(const V& a, float b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b; return r; }         \
    inline V operator*(const V& a, float b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b; return r; }         \
    inline V operator/(const V& a, float b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] / b; return r; }         \
    inline V operator+(float a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a + b.d[i]; return r; }         \
    inline V operator-(float a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a - b.d[i]; return r; }         \
    inline V operator*(float a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a * b.d[i]; return r; }         \
    inline V operator/(float a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = a / b.d[i]; return r; }         \
    inline V operator-(const V& a) { V r; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }                     \
    inline V& operator+=(V& a, const V& b) { for (int i = 0; i < N; ++i) a.d[i] += b.d[i]; return a; }                  \
    inline V& operator-=(V& a, const V& b) { for (int i = 0; i < N; ++i) a.d[i] -= b.d[i]; return a; }                  \
    inline V& operator*=(V& a, const V& b) { for (int i = 0; i < N; ++i) a.d[i] *= b.d[i]; return a; }                  \
    inline V& operator/=(V& a, const V& b) { for (int i = 0; i < N; ++i) a.d[i] /= b.d[i]; return a; }                  \
    inline V& operator*=(V& a, float b) { for (int i = 0; i < N; ++i) a.d[i] *= b; return a; }                          \
    inline V& operator/=(V& a, float b) { for (int i = 0; i < N; ++i) a.d[i] /= b; return a; }                          \
    inline V& operator+=(V& a, float b) { for (int i = 0; i < N; ++i) a.d[i] += b; return a; }
VQ_HLSL_OPS(float2, 2)
VQ_HLSL_OPS(float3, 3)
VQ_HLSL_OPS(float4, 4)
#undef VQ_HLSL_OPS

// ---- scalar intrinsics --------------------------------------------------------------------------------------------------
// FMax / FMin: a NaN operand is dropped; of two zeros max returns +0 and min -0 whatever their order (IEEE 754-2019 maximum / minimum, what a GPU's v_max_f32 / v_min_f32 do).
// libm's fmaxf / fminf return their FIRST operand on that tie — an artefact of the host library, not of the shader (round 6)
inline float vq_fmax(float a, float b) { if (a != a) return b; if (b != b) return a; if (a == b) return __builtin_signbit(a) ? b : a; return a > b ? a : b; }
inline float vq_fmin(float a, float b) { if (a != a) return b; if (b != b) return a; if (a == b) return __builtin_signbit(a) ? a : b; return a < b ? a : b; }
inline float saturate(float x) { return vq_fmin(vq_fmax(x, 0.0f), 1.0f); }
inline float max(float a, float b) { return vq_fmax(a, b); }
inline float min(float a, float b) { return vq_fmin(a, b); }
inline int max(int a, int b) { return a > b ? a : b; }
inline int min(int a, int b) { return a < b ? a : b; }
inline uint max(uint a, uint b) { return a > b ? a : b; }
inline uint min(uint a, uint b) { return a < b ? a : b; }
inline float max(float a, int b) { return vq_fmax(a, (float)b); }      // HLSL promotes the int: max(L.z, 0)
inline float min(float a, int b) { return vq_fmin(a, (float)b); }
inline float max(int a, float b) { return vq_fmax((float)a, b); }
inline float min(int a, float b) { return vq_fmin((float)a, b); }
inline int abs(int x) { return x < 0 ? -x : x; }
inline int clamp(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
inline float trunc(float x) { return truncf(x); }
inline float clamp(float x, float lo, float hi) { return vq_fmin(vq_fmax(x, lo), hi); }
inline float abs(float x) { return fabsf(x); }
inline float sqrt(float x) { return sqrtf(x); }
#ifdef VQ_SHIM_DXC
inline float rsqrt(float x) { return (float)(1.0 / std::sqrt((double)x)); }   // DXIL Rsqrt, correctly rounded (one rounding, not 1/RN(sqrt))
#else
inline float rsqrt(float x) { return 1.0f / sqrtf(x); }
#endif
inline float rcp(float x) { return 1.0f / x; }
inline float sin(float x) { return sinf(x); }
inline float cos(float x) { return cosf(x); }
inline float tan(float x) { return tanf(x); }
inline float acos(float x) { return acosf(x); }
inline float asin(float x) { return asinf(x); }
inline float atan2(float y, float x) { return atan2f(y, x); }
inline float exp2(float x) { return exp2f(x); }
inline float log2(float x) { return log2f(x); }
inline float exp(float x) { return expf(x); }
inline float log(float x) { return logf(x); }
inline float floor(float x) { return floorf(x); }
inline float frac(float x) { return x - floorf(x); }
#ifdef VQ_SHIM_DXC
inline float pow(float x, float y) { return vqo::exp2_(y * vqo::log2_(x)); }    // DXIL Exp(y * Log(x)) with the contract's exp2 / log2 (vqo_math.h) in place of libm's
#else
inline float pow(float x, float y) { return exp2f(y * log2f(x)); }          // DXC: pow -> exp2(y * log2(x))
#endif
inline float lerp(float a, float b, float t) { return a + t * (b - a); }
inline float step(float e, float x) { return x >= e ? 1.0f : 0.0f; }
inline float sign(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }
inline float asfloat(uint u) { float f; std::memcpy(&f, &u, 4); return f; }
inline uint asuint(float f) { uint u; std::memcpy(&u, &f, 4); return u; }
inline float asfloat(int i) { float f; std::memcpy(&f, &i, 4); return f; }
inline float asfloat(float f) { return f; }
inline uint asuint(uint u) { return u; }
inline uint asuint(int i) { return (uint)i; }
inline int asint(float f) { int i; std::memcpy(&i, &f, 4); return i; }
inline int asint(uint u) { return (int)u; }
uint f32tof16(float f);          // RNE, defined by the harness that needs it (vqo::f32_to_f16)
float f16tof32(uint h);

// The two READINGS of the intrinsics whose lowering the HLSL does not fix:
//   literal (default): dot = products and sums rounded one by one, left to right; normalize(v) = v / length(v) (IEEE quotients)
//   VQ_SHIM_DXC      : what DXC's HLOperationLower emits — DXIL Dot2/3/4 evaluated as an FMA chain fma(az,bz, fma(ay,by, ax*bx)) (how GPU
//                      back ends expand the intrinsic), normalize(v) = v * rsqrt(dot(v,v)) (TranslateNormalize: Dot -> Rsqrt -> FMul) with a
//                      correctly rounded rsqrt, pow = exp2(y * log2 x) with the contract's exp2 / log2; SURVEY.md §8c's list
#ifdef VQ_SHIM_DXC
#define VQ_SHIM_DOT_STEP(x, y, s) __builtin_fmaf((x), (y), (s))
#define VQ_SHIM_NORMALIZE(a) ((a) * rsqrt(dot((a), (a))))
#else
#define VQ_SHIM_DOT_STEP(x, y, s) ((s) + (x) * (y))
#define VQ_SHIM_NORMALIZE(a) ((a) / length(a))
#endif
#define VQ_HLSL_MAP1(V, N, FN) inline V FN(const V& a) { V r; for (int i = 0; i < N; ++i) r.d[i] = FN(a.d[i]); return r; }
#define VQ_HLSL_MAP2(V, N, FN) inline V FN(const V& a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = FN(a.d[i], b.d[i]); return r; }
#define VQ_HLSL_VEC(V, N)                                                                                            \
    VQ_HLSL_MAP1(V, N, rcp) VQ_HLSL_MAP1(V, N, rsqrt) VQ_HLSL_MAP1(V, N, saturate) VQ_HLSL_MAP1(V, N, abs) VQ_HLSL_MAP1(V, N, sqrt) VQ_HLSL_MAP1(V, N, floor)              \
    VQ_HLSL_MAP1(V, N, frac) VQ_HLSL_MAP1(V, N, exp2) VQ_HLSL_MAP1(V, N, log2) VQ_HLSL_MAP1(V, N, sin) VQ_HLSL_MAP1(V, N, cos) \
    VQ_HLSL_MAP2(V, N, max) VQ_HLSL_MAP2(V, N, min) VQ_HLSL_MAP2(V, N, pow)                                               \
    inline V pow(const V& a, float b) { V r; for (int i = 0; i < N; ++i) r.d[i] = pow(a.d[i], b); return r; }             \
    inline V max(const V& a, float b) { V r; for (int i = 0; i < N; ++i) r.d[i] = max(a.d[i], b); return r; }             \
    inline V max(float a, const V& b) { V r; for (int i = 0; i < N; ++i) r.d[i] = max(a, b.d[i]); return r; }             \
    inline V min(const V& a, float b) { V r; for (int i = 0; i < N; ++i) r.d[i] = min(a.d[i], b); return r; }             \
    inline V clamp(const V& a, float lo, float hi) { V r; for (int i = 0; i < N; ++i) r.d[i] = clamp(a.d[i], lo, hi); return r; } \
    inline V clamp(const V& a, const V& lo, const V& hi) { V r; for (int i = 0; i < N; ++i) r.d[i] = clamp(a.d[i], lo.d[i], hi.d[i]); return r; } \
    inline V lerp(const V& a, const V& b, float t) { V r; for (int i = 0; i < N; ++i) r.d[i] = lerp(a.d[i], b.d[i], t); return r; } \
    inline V lerp(const V& a, const V& b, const V& t) { V r; for (int i = 0; i < N; ++i) r.d[i] = lerp(a.d[i], b.d[i], t.d[i]); return r; } \
    inline float dot(const V& a, const V& b) { float s = a.d[0] * b.d[0]; for (int i = 1; i < N; ++i) s = VQ_SHIM_DOT_STEP(a.d[i], b.d[i], s); return s; } \
    inline float length(const V& a) { return sqrtf(dot(a, a)); }                                                           \
    inline V normalize(const V& a) { return VQ_SHIM_NORMALIZE(a); }                                                        \
    inline V reflect(const V& i, const V& n) { return i - 2.0f * dot(n, i) * n; }
VQ_HLSL_VEC(float2, 2)
VQ_HLSL_VEC(float3, 3)
VQ_HLSL_VEC(float4, 4)
#undef VQ_HLSL_VEC
#undef VQ_HLSL_MAP1
#undef VQ_HLSL_MAP2
struct bool3 { bool x, y, z; };
inline bool3 operator<(const float3& a, float b) { return { a.x < b, a.y < b, a.z < b }; }
inline bool3 operator>(const float3& a, float b) { return { a.x > b, a.y > b, a.z > b }; }
inline float3 select(const bool3& c, const float3& a, const float3& b) { return float3(c.x ? a.x : b.x, c.y ? a.y : b.y, c.z ? a.z : b.z); }
inline float3 lerp(const float3& a, float b, float t) { return float3(lerp(a.x, b, t), lerp(a.y, b, t), lerp(a.z, b, t)); }
inline float3 cross(const float3& a, const float3& b) { return float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

// ---- matrices (m[row][col], HLSL element order) -------------------------------------------------------------------------
struct float3x3 {
    float m[3][3];
    float3x3() : m{} {}
    float3x3(const float3& r0, const float3& r1, const float3& r2) : m{ { r0.x, r0.y, r0.z }, { r1.x, r1.y, r1.z }, { r2.x, r2.y, r2.z } } {}
    float3x3(float a, float b, float c, float d, float e, float f, float g, float h, float i) : m{ { a, b, c }, { d, e, f }, { g, h, i } } {}
    float* operator[](int r) { return m[r]; }
    const float* operator[](int r) const { return m[r]; }
};
typedef float3x3 half3x3;
struct float4x4 {
    float m[4][4];
    float4x4() : m{} {}
    float* operator[](int r) { return m[r]; }
    const float* operator[](int r) const { return m[r]; }
};
typedef float4x4 matrix;
// mul(vector, matrix): row vector times matrix;  mul(matrix, vector): matrix times column vector
inline float3 mul(const float3& v, const float3x3& M) {
    float3 r;
    for (int c = 0; c < 3; ++c) r.d[c] = v.x * M.m[0][c] + v.y * M.m[1][c] + v.z * M.m[2][c];
    return r;
}
inline float3 mul(const float3x3& M, const float3& v) {
    float3 r;
    for (int i = 0; i < 3; ++i) r.d[i] = M.m[i][0] * v.x + M.m[i][1] * v.y + M.m[i][2] * v.z;
    return r;
}
inline float4 mul(const float4x4& M, const float4& v) {
    float4 r;
    for (int i = 0; i < 4; ++i) r.d[i] = M.m[i][0] * v.x + M.m[i][1] * v.y + M.m[i][2] * v.z + M.m[i][3] * v.w;
    return r;
}
// mul(float4x4, float3): HLSL truncates the matrix to its upper-left 3x3 (implicit truncation, warning X3206) — ClassifyReflectionTiles.hlsl:89
inline float3 mul(const float4x4& M, const float3& v) {
    float3 r;
    for (int i = 0; i < 3; ++i) r.d[i] = M.m[i][0] * v.x + M.m[i][1] * v.y + M.m[i][2] * v.z;
    return r;
}
inline float4 mul(const float4& v, const float4x4& M) {
    float4 r;
    for (int c = 0; c < 4; ++c) r.d[c] = v.x * M.m[0][c] + v.y * M.m[1][c] + v.z * M.m[2][c] + v.w * M.m[3][c];
    return r;
}

inline float2 asfloat(const uint2& u) { return float2(asfloat(u.x), asfloat(u.y)); }
inline float3 asfloat(const uint3& u) { return float3(asfloat(u.x), asfloat(u.y), asfloat(u.z)); }
inline float4 asfloat(const uint4& u) { return float4(asfloat(u.x), asfloat(u.y), asfloat(u.z), asfloat(u.w)); }
inline uint2 asuint(const float2& f) { return uint2(asuint(f.x), asuint(f.y)); }
inline uint3 asuint(const float3& f) { return uint3(asuint(f.x), asuint(f.y), asuint(f.z)); }
inline uint4 asuint(const float4& f) { return uint4(asuint(f.x), asuint(f.y), asuint(f.z), asuint(f.w)); }

inline uint3 operator!=(const float3& a, const uint3& b) { return uint3(a.x != (float)b.x, a.y != (float)b.y, a.z != (float)b.z); }
inline uint4 operator!=(const float4& a, const uint4& b) { return uint4(a.x != (float)b.x, a.y != (float)b.y, a.z != (float)b.z, a.w != (float)b.w); }

// ---- resources: fixed-function sampling is delegated to the harness -----------------------------------------------------
struct SamplerState { int id = 0; };
enum { kSampleImplicit = 0, kSampleBias = 1, kSampleLevel = 2 };
struct Texture2D;  struct TextureCube;  struct Texture2DArray;  struct TextureCubeArray;
float4 vqref_sample_2d(const Texture2D& t, const SamplerState& s, float2 uv, int mode, float arg);
float4 vqref_sample_cube(const TextureCube& t, const SamplerState& s, float3 dir, int mode, float arg);
float4 vqref_sample_2d_array(const Texture2DArray& t, const SamplerState& s, float3 uvw);
float4 vqref_sample_cube_array(const TextureCubeArray& t, const SamplerState& s, float4 dirw);
float4 vqref_load_2d(const Texture2D& t, int x, int y, int mip);
float4 vqref_gather_2d(const Texture2D& t, const SamplerState& s, float2 uv, int channel);   // D3D order: (0,1) (1,1) (1,0) (0,0)
void   vqref_dims_2d(const Texture2D& t, uint* w, uint* h);
struct Texture2D {
    const void* res = nullptr; int kind = 0;
    float4 Sample(const SamplerState& s, float2 uv) const { return vqref_sample_2d(*this, s, uv, kSampleImplicit, 0.0f); }
    float4 SampleBias(const SamplerState& s, float2 uv, float bias) const { return vqref_sample_2d(*this, s, uv, kSampleBias, bias); }
    float4 SampleLevel(const SamplerState& s, float2 uv, float lod) const { return vqref_sample_2d(*this, s, uv, kSampleLevel, lod); }
    float4 Load(int3 p) const { return vqref_load_2d(*this, p.x, p.y, p.z); }
    float4 GatherRed(const SamplerState& s, float2 uv, int2) const { return vqref_gather_2d(*this, s, uv, 0); }
    float4 GatherGreen(const SamplerState& s, float2 uv, int2) const { return vqref_gather_2d(*this, s, uv, 1); }
    float4 GatherBlue(const SamplerState& s, float2 uv, int2) const { return vqref_gather_2d(*this, s, uv, 2); }
    float4 operator[](uint2 p) const { return vqref_load_2d(*this, (int)p.x, (int)p.y, 0); }
    float4 operator[](int2 p) const { return vqref_load_2d(*this, p.x, p.y, 0); }
    float4 operator[](const usw_xy& p) const { return vqref_load_2d(*this, (int)p.d[0], (int)p.d[1], 0); }
    void GetDimensions(uint& w, uint& h) const { vqref_dims_2d(*this, &w, &h); }
};
// Texture2D<float>: Load / operator[] return the red channel
struct Texture2DF : Texture2D {
    float Load(int3 p) const { return vqref_load_2d(*this, p.x, p.y, p.z).x; }
    float operator[](uint2 p) const { return vqref_load_2d(*this, (int)p.x, (int)p.y, 0).x; }
    float operator[](int2 p) const { return vqref_load_2d(*this, p.x, p.y, 0).x; }
};
// RWBuffer<T>: declared by shaders whose appends are cut away (hlsl2cpp.py CUTS); never dereferenced with data == nullptr
#define globallycoherent
template <class T> struct RWBuffer { T* data = nullptr; T dummy{}; T& operator[](uint i) { return data ? data[i] : dummy; } };
struct TextureCube {
    const void* res = nullptr; int kind = 0;
    float4 Sample(const SamplerState& s, float3 d) const { return vqref_sample_cube(*this, s, d, kSampleImplicit, 0.0f); }
    float4 SampleLevel(const SamplerState& s, float3 d, float lod) const { return vqref_sample_cube(*this, s, d, kSampleLevel, lod); }
};
struct Texture2DArray {
    const void* res = nullptr; int kind = 0;
    float4 Sample(const SamplerState& s, float3 uvw) const { return vqref_sample_2d_array(*this, s, uvw); }
};
struct TextureCubeArray {
    const void* res = nullptr; int kind = 0;
    float4 Sample(const SamplerState& s, float4 dirw) const { return vqref_sample_cube_array(*this, s, dirw); }
};
// RWTexture2D<T>: a plain row-major image the harness owns
template <class T> struct RWTexture2D {
    T* data = nullptr; int width = 0, height = 0;
    T dummy;
    T& at(long x, long y) { return (x < 0 || y < 0 || x >= width || y >= height) ? dummy : data[(size_t)y * width + x]; }
    T& operator[](uint2 p) { return at(p.x, p.y); }
    T& operator[](int2 p) { return at(p.x, p.y); }
    T& operator[](const usw_xy& p) { return at(p.d[0], p.d[1]); }
    void GetDimensions(uint& w, uint& h) const { w = (uint)width; h = (uint)height; }
};

} // namespace hlsl
