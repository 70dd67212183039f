// hdri.hip — SURVEY.md §8(f).3: Radiance .hdr (RGBE) ingest.
//
// Replaces Image::LoadFromFile -> stbi_loadf (call site Source/Renderer/Resources/TextureManager.cpp:566). The decoder
// the reference uses is stb_image's (Libs/VQUtils submodule, not vendored): stbi__hdr_test / stbi__hdr_load /
// stbi__hdr_convert of stb_image.h v2.2x, whose published behaviour is restated here:
//   * first line "#?RADIANCE" or "#?RGBE"; header lines until an empty line, one of them "FORMAT=32-bit_rle_rgbe";
//     then "-Y <height> +X <width>" (any other orientation is rejected, as stb does);
//   * width < 8 or >= 32768: flat RGBE quadruples; otherwise each scanline starts 02 02 hi lo (hi<<8|lo == width, hi < 0x80)
//     followed by four run-length coded byte planes (count > 128: run of count-128 copies; else count literals; a count of
//     0 or one overrunning the scanline is corrupt); a first scanline without the 02 02 marker means the whole image is flat;
//   * pixel = (r,g,b) * 2^(e - 136) (ldexp(1, e-(128+8))), (0,0,0) for e == 0, alpha = 1.
// Split (round 5): the host parses the header and WALKS the run headers of every scanline — count bytes only, no pixel is touched: that finds where every byte
// plane of every scanline starts (a scanline's length is only known by walking its runs) and validates the file; the encoded bytes go to the GPU as they are and
// k_hdr_expand expands them: one workgroup per scanline parks the scanline's encoded bytes in LDS, one wave per byte plane expands its runs (a run or literal group per
// iteration, its up to 128 bytes written by the lanes in parallel) into four plane rows in LDS, then every lane converts whole pixels (RGBE -> RGBA32F, 16-byte coalesced
// stores) straight into level 0 of the mip chain. Scanlines too wide for the LDS (> 19 000 pixels) and flat (not run-length coded) files take the round-4 path:
// expansion on the host, k_rgbe_to_rgba32f on the GPU.
#include <cstdlib>
#include <cstring>
#include "vq_internal.h"

namespace vqk {

namespace {
struct Cursor {
    const uint8_t* p; size_t n, i;
    bool eof() const { return i >= n; }
    int get() { return i < n ? p[i++] : -1; }
};
// stbi__hdr_gettoken: reads up to the next '\n' (not included), at most 1023 chars kept
bool getToken(Cursor& c, char* buf, size_t cap) {
    size_t len = 0;
    if (c.eof()) return false;
    for (;;) {
        const int ch = c.get();
        if (ch < 0 || ch == '\n') break;
        if (len + 1 < cap) buf[len++] = (char)ch;
    }
    buf[len] = 0;
    return true;
}
} // namespace

int hdr_parse_header(const uint8_t* f, size_t n, int* w, int* h, size_t* off, const char** err) {
    Cursor c = { f, n, 0 };
    char tok[1024];
    if (!getToken(c, tok, sizeof tok) || (std::strcmp(tok, "#?RADIANCE") != 0 && std::strcmp(tok, "#?RGBE") != 0)) { *err = "hdr: not a Radiance file (missing #?RADIANCE / #?RGBE)"; return -1; }
    bool valid = false;
    for (;;) {
        if (!getToken(c, tok, sizeof tok)) { *err = "hdr: truncated header"; return -1; }
        if (tok[0] == 0) break;
        if (std::strcmp(tok, "FORMAT=32-bit_rle_rgbe") == 0) valid = true;
    }
    if (!valid) { *err = "hdr: unsupported format (FORMAT=32-bit_rle_rgbe expected)"; return -1; }
    if (!getToken(c, tok, sizeof tok)) { *err = "hdr: missing resolution line"; return -1; }
    if (std::strncmp(tok, "-Y ", 3) != 0) { *err = "hdr: unsupported data layout (-Y expected)"; return -1; }
    char* p = tok + 3;
    const long hh = std::strtol(p, &p, 10);
    while (*p == ' ') ++p;
    if (std::strncmp(p, "+X ", 3) != 0) { *err = "hdr: unsupported data layout (+X expected)"; return -1; }
    p += 3;
    const long ww = std::strtol(p, nullptr, 10);
    if (ww <= 0 || hh <= 0 || ww > (1 << 24) || hh > (1 << 24)) { *err = "hdr: bad image dimensions"; return -1; }
    *w = (int)ww; *h = (int)hh; *off = c.i;
    return 0;
}

// Expands the pixel data into w*h RGBE quadruples (host).
int hdr_expand_rgbe(const uint8_t* f, size_t n, size_t off, int w, int h, uint8_t* rgbe, const char** err) {
    Cursor c = { f, n, off };
    const size_t px = (size_t)w * h;
    auto flat = [&](size_t firstPixel) -> int {
        const size_t need = (px - firstPixel) * 4;
        if (c.n - c.i < need) { *err = "hdr: truncated pixel data"; return -1; }
        std::memcpy(rgbe + firstPixel * 4, c.p + c.i, need);
        c.i += need;
        return 0;
    };
    if (w < 8 || w >= 32768) return flat(0);
    for (int j = 0; j < h; ++j) {
        const int c1 = c.get(), c2 = c.get(), len = c.get();
        if (len < 0) { *err = "hdr: truncated pixel data"; return -1; }
        if (c1 != 2 || c2 != 2 || (len & 0x80)) {
            // not run-length encoded: these three bytes + the next one are the first pixel of flat data. stb_image takes this
            // path with (i,j) = (1,0), i.e. it is only meaningful on the first scanline; later scanlines are corrupt.
            if (j != 0) { *err = "hdr: scanline without run-length marker after the first"; return -1; }
            const int c4 = c.get();
            if (c4 < 0) { *err = "hdr: truncated pixel data"; return -1; }
            rgbe[0] = (uint8_t)c1; rgbe[1] = (uint8_t)c2; rgbe[2] = (uint8_t)len; rgbe[3] = (uint8_t)c4;
            return flat(1);
        }
        const int lo = c.get();
        if (lo < 0) { *err = "hdr: truncated pixel data"; return -1; }
        if (((len << 8) | lo) != w) { *err = "hdr: invalid decoded scanline length"; return -1; }
        uint8_t* row = rgbe + (size_t)j * w * 4;
        for (int k = 0; k < 4; ++k) {
            int i = 0, nleft;
            while ((nleft = w - i) > 0) {
                int count = c.get();
                if (count < 0) { *err = "hdr: truncated pixel data"; return -1; }
                if (count > 128) {
                    const int value = c.get();
                    if (value < 0) { *err = "hdr: truncated pixel data"; return -1; }
                    count -= 128;
                    if (count == 0 || count > nleft) { *err = "hdr: corrupt run"; return -1; }
                    for (int z = 0; z < count; ++z) row[(size_t)(i++) * 4 + k] = (uint8_t)value;
                } else {
                    if (count == 0 || count > nleft) { *err = "hdr: corrupt run"; return -1; }
                    if (c.n - c.i < (size_t)count) { *err = "hdr: truncated pixel data"; return -1; }
                    for (int z = 0; z < count; ++z) row[(size_t)(i++) * 4 + k] = c.p[c.i++];
                }
            }
        }
    }
    return 0;
}

// The run headers of a run-length coded image, walked without expanding anything: planeOff[4 j + k] = offset (from the start of the file) of byte plane k of
// scanline j, planeOff[4 h] = the end of the data. Validates exactly what hdr_expand_rgbe validates. Returns 1 when the image is not (entirely) run-length coded
// (flat data: the caller expands on the host), 0 on success, -1 on a corrupt file.
int hdr_walk_runs(const uint8_t* f, size_t n, size_t off, int w, int h, uint32_t* planeOff, const char** err) {
    if (w < 8 || w >= 32768 || n > 0xfffffff0u) return 1;
    size_t i = off;
    for (int j = 0; j < h; ++j) {
        if (i > n || n - i < 3) { *err = "hdr: truncated pixel data"; return -1; }
        if (f[i] != 2 || f[i + 1] != 2 || (f[i + 2] & 0x80)) {
            if (j != 0) { *err = "hdr: scanline without run-length marker after the first"; return -1; }
            return 1;                                          // flat data: hdr_expand_rgbe
        }
        if (n - i < 4) { *err = "hdr: truncated pixel data"; return -1; }
        if (((f[i + 2] << 8) | f[i + 3]) != w) { *err = "hdr: invalid decoded scanline length"; return -1; }
        i += 4;
        for (int k = 0; k < 4; ++k) {
            planeOff[4 * (size_t)j + k] = (uint32_t)i;
            int left = w;
            while (left > 0) {
                if (i >= n) { *err = "hdr: truncated pixel data"; return -1; }
                int count = f[i++];
                if (count > 128) {
                    if (i >= n) { *err = "hdr: truncated pixel data"; return -1; }
                    count -= 128;
                    if (count > left) { *err = "hdr: corrupt run"; return -1; }
                    i += 1;
                } else {
                    if (count == 0 || count > left) { *err = "hdr: corrupt run"; return -1; }
                    if (n - i < (size_t)count) { *err = "hdr: truncated pixel data"; return -1; }
                    i += (size_t)count;
                }
                left -= count;
            }
        }
    }
    planeOff[4 * (size_t)h] = (uint32_t)i;
    return 0;
}

__device__ __forceinline__ float4 rgbe_to_float4(uint32_t q) {                       // stbi__hdr_convert, req_comp == 4: rgb * ldexp(1, e - 136), zero when e == 0, alpha 1
    const int e = (int)(q >> 24);
    float4 c = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    if (e != 0) {
        const float f1 = __builtin_ldexpf(1.0f, e - 136);       // exact, denormal below e = 10
        c.x = (float)(q & 255u) * f1; c.y = (float)((q >> 8) & 255u) * f1; c.z = (float)((q >> 16) & 255u) * f1;
    }
    return c;
}

// One workgroup per scanline, 256 lanes = one wave per byte plane. LDS: the scanline's encoded bytes (from the 4-byte aligned address below its first plane), then
// four plane rows of `pitch` bytes. file: the whole .hdr file on the device, 4-byte aligned; planeOff: hdr_walk_runs' table.
__global__ __launch_bounds__(256) void k_hdr_expand(const uint8_t* __restrict__ file, const uint32_t* __restrict__ planeOff, float4* __restrict__ out, int w, int encCap, int pitch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int j = blockIdx.x, t = threadIdx.x;
    const uint32_t o0 = planeOff[4 * (size_t)j], o4 = planeOff[4 * (size_t)j + 4];
    const uint32_t a0 = o0 & ~3u;                             // the copy starts at an aligned dword
    const int nDw = (int)((o4 - a0 + 3) >> 2);
    for (int i = t; i < nDw; i += 256) ((uint32_t*)lds)[i] = ((const uint32_t*)(file + a0))[i];
    __syncthreads();
    unsigned char* planes = lds + encCap;
    {
        const int k = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
        unsigned char* dst = planes + k * pitch;
        int src = (int)(planeOff[4 * (size_t)j + k] - a0);    // wave-uniform
        int done = 0;
        while (done < w) {                                    // one run or literal group per iteration: validated by the host walk
            int count = lds[src];
            const bool run = count > 128;
            if (run) count -= 128;
            if (lane < count) dst[done + lane] = lds[run ? src + 1 : src + 1 + lane];
            if (lane + 64 < count) dst[done + lane + 64] = lds[run ? src + 1 : src + 65 + lane];
            src += run ? 2 : 1 + count;
            done += count;
        }
    }
    __syncthreads();
    for (int x = t; x < w; x += 256) {
        const uint32_t q = (uint32_t)planes[x] | ((uint32_t)planes[pitch + x] << 8) | ((uint32_t)planes[2 * pitch + x] << 16) | ((uint32_t)planes[3 * pitch + x] << 24);
        out[(size_t)j * w + x] = rgbe_to_float4(q);
    }
}
// encoded bytes of the longest scanline (+ alignment slack) and the LDS the kernel needs for it; false when it does not fit
bool hdr_expand_fits(const uint32_t* planeOff, int w, int h, int ldsLimit, int* encCap, int* pitch, int* ldsBytes) {
    uint32_t longest = 0;
    for (int j = 0; j < h; ++j) { const uint32_t len = planeOff[4 * (size_t)j + 4] - (planeOff[4 * (size_t)j] & ~3u); if (len > longest) longest = len; }
    *encCap = (int)((longest + 3 + 15) & ~15u);
    *pitch = (w + 15) & ~15;
    *ldsBytes = *encCap + 4 * *pitch;
    return *ldsBytes <= ldsLimit;                            // the device's own limit (hipDeviceAttributeMaxSharedMemoryPerBlock), not a constant of gfx950
}
hipError_t launch_hdr_expand(hipStream_t s, const void* file, const void* planeOff, void* out, int w, int h, int encCap, int pitch, int ldsBytes) {
    hipError_t e = hipFuncSetAttribute((const void*)k_hdr_expand, hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_hdr_expand, dim3((unsigned)h), dim3(256), ldsBytes, s, (const uint8_t*)file, (const uint32_t*)planeOff, (float4*)out, w, encCap, pitch);
    return hipGetLastError();
}

// stbi__hdr_convert over expanded RGBE quadruples (the host-expansion path)
__global__ __launch_bounds__(256) void k_rgbe_to_rgba32f(const uint32_t* __restrict__ rgbe, float4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        out[i] = rgbe_to_float4(rgbe[i]);
    }
}

// HDRI downsize by an integer factor k in both directions (8k -> 4k / 2k / 1k: k = 2, 4, 8): every output texel is the mean of its k x k
// block. One lane per output texel; the block is summed row by row, left to right (a fixed fp32 order the oracle repeats), then scaled
// by 1/(k*k) (a power of two for the engine's resolutions: exact). Alpha := 1. HBM-bound: 16 B * k*k in, 16 B out per texel.
__global__ __launch_bounds__(256) void k_downsize_box(const float4* __restrict__ src, float4* __restrict__ dst, int sw, int dw, int dh, int k, float inv) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dw || y >= dh) return;
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    for (int j = 0; j < k; ++j) {
        const float4* row = src + (size_t)(y * k + j) * sw + (size_t)x * k;
        for (int i = 0; i < k; ++i) { const float4 t = row[i]; ax = ax + t.x; ay = ay + t.y; az = az + t.z; }
    }
    dst[(size_t)y * dw + x] = make_float4(ax * inv, ay * inv, az * inv, 1.0f);
}
hipError_t launch_downsize_box(hipStream_t s, const void* src, void* dst, int sw, int sh, int k) {
    const int dw = sw / k, dh = sh / k;
    hipLaunchKernelGGL(k_downsize_box, dim3((dw + 255) / 256, dh), dim3(256), 0, s, (const float4*)src, (float4*)dst, sw, dw, dh, k, 1.0f / (float)(k * k));
    return hipGetLastError();
}

hipError_t launch_rgbe_to_rgba32f(hipStream_t s, const void* rgbe, void* out, size_t n) {
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_rgbe_to_rgba32f, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s, (const uint32_t*)rgbe, (float4*)out, n);
    return hipGetLastError();
}

} // namespace vqk
