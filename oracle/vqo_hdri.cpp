// vqo_hdri.cpp — CPU restatement of the Radiance .hdr (RGBE) decode the reference reaches through
// Image::LoadFromFile -> stbi_loadf (call site Source/Renderer/Resources/TextureManager.cpp:566), SURVEY.md §8(f).3.
//
// ORACLE / TEST INFRASTRUCTURE ONLY (see vqo_oracle.cpp). PARITY UNPINNED: the decoder is stb_image's, inside the
// un-vendored Libs/VQUtils submodule (version not pinned by the reference; the algorithm below is the one published in
// stb_image.h v2.2x: stbi__hdr_test, stbi__hdr_load, stbi__hdr_convert); the reference ships no .hdr fixture with known
// decoded values. One deliberate difference: data that ends early is an error here (stb_image reads zeros past EOF).
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Reader {
    const uint8_t* d; size_t n; size_t pos = 0;
    bool more() const { return pos < n; }
    int byte() { return pos < n ? (int)d[pos++] : -1; }
    // stbi__hdr_gettoken: one header line without its '\n'
    bool line(std::string& s) {
        s.clear();
        if (!more()) return false;
        while (more()) { const char c = (char)d[pos++]; if (c == '\n') break; if (s.size() < 1023) s.push_back(c); }
        return true;
    }
};

// stbi__hdr_convert with req_comp == 4
inline void convert(const uint8_t* in, float* out) {
    if (in[3] != 0) {
        const float f1 = std::ldexp(1.0f, (int)in[3] - (128 + 8));
        out[0] = in[0] * f1; out[1] = in[1] * f1; out[2] = in[2] * f1;
    } else {
        out[0] = out[1] = out[2] = 0.0f;
    }
    out[3] = 1.0f;
}

} // namespace

extern "C" {

// returns 0 and fills w,h,data offset; <0 on a malformed header
int vqo_hdr_parse_header(const uint8_t* file, size_t n, int* w, int* h, size_t* off) {
    Reader r{ file, n };
    std::string s;
    if (!r.line(s) || (s != "#?RADIANCE" && s != "#?RGBE")) return -1;
    bool fmt = false;
    for (;;) {
        if (!r.line(s)) return -2;
        if (s.empty()) break;
        if (s == "FORMAT=32-bit_rle_rgbe") fmt = true;
    }
    if (!fmt) return -3;
    if (!r.line(s) || s.compare(0, 3, "-Y ") != 0) return -4;
    char* end = nullptr;
    const long hh = std::strtol(s.c_str() + 3, &end, 10);
    while (*end == ' ') ++end;
    if (std::strncmp(end, "+X ", 3) != 0) return -4;
    const long ww = std::strtol(end + 3, nullptr, 10);
    if (ww <= 0 || hh <= 0) return -5;
    *w = (int)ww; *h = (int)hh; *off = r.pos;
    return 0;
}

// out: w*h*4 floats (RGBA32F, alpha 1). returns 0, or <0 on corrupt / truncated data
int vqo_hdr_decode_rgba32f(const uint8_t* file, size_t n, float* out, int w, int h) {
    int fw, fh; size_t off;
    if (vqo_hdr_parse_header(file, n, &fw, &fh, &off) || fw != w || fh != h) return -1;
    Reader r{ file, n, off };
    std::vector<uint8_t> px((size_t)w * h * 4);
    size_t flatFrom = (size_t)-1;                       // index of the first pixel stored flat, if any
    if (w < 8 || w >= 32768) flatFrom = 0;
    for (int y = 0; y < h && flatFrom == (size_t)-1; ++y) {
        int hdr[4];
        for (int k = 0; k < 3; ++k) { hdr[k] = r.byte(); if (hdr[k] < 0) return -2; }
        if (hdr[0] != 2 || hdr[1] != 2 || (hdr[2] & 0x80)) {
            if (y != 0) return -3;
            hdr[3] = r.byte(); if (hdr[3] < 0) return -2;
            for (int k = 0; k < 4; ++k) px[k] = (uint8_t)hdr[k];
            flatFrom = 1;
            break;
        }
        hdr[3] = r.byte(); if (hdr[3] < 0) return -2;
        if (((hdr[2] << 8) | hdr[3]) != w) return -4;
        for (int plane = 0; plane < 4; ++plane) {
            int x = 0;
            while (x < w) {
                int count = r.byte(); if (count < 0) return -2;
                const bool run = count > 128;
                if (run) count -= 128;
                if (count == 0 || count > w - x) return -5;
                if (run) {
                    const int v = r.byte(); if (v < 0) return -2;
                    for (int z = 0; z < count; ++z, ++x) px[((size_t)y * w + x) * 4 + plane] = (uint8_t)v;
                } else {
                    for (int z = 0; z < count; ++z, ++x) { const int v = r.byte(); if (v < 0) return -2; px[((size_t)y * w + x) * 4 + plane] = (uint8_t)v; }
                }
            }
        }
    }
    if (flatFrom != (size_t)-1) {
        for (size_t i = flatFrom * 4; i < px.size(); ++i) { const int v = r.byte(); if (v < 0) return -2; px[i] = (uint8_t)v; }
    }
    for (size_t i = 0; i < (size_t)w * h; ++i) convert(&px[i * 4], out + i * 4);
    return 0;
}

// Integer-ratio downsize (vqhip_hdr_downsize_rgba32f; EnvironmentMap.cpp:142-209 — the reference's resampler is stb_image_resize in the absent
// submodule: PARITY UNPINNED, plain k x k mean restated): block summed row by row, left to right, scaled by 1/(k*k), alpha 1.
int vqo_hdr_downsize_rgba32f(const float* in, int w, int h, float* out, int ow, int oh) {
    if (!in || !out || w <= 0 || h <= 0 || ow <= 0 || oh <= 0) return -1;
    const int k = w / ow;
    if (k < 1 || ow * k != w || oh * k != h) return -3;
    const float inv = 1.0f / (float)(k * k);
    for (int y = 0; y < oh; ++y)
        for (int x = 0; x < ow; ++x) {
            float a[3] = { 0, 0, 0 };
            for (int j = 0; j < k; ++j)
                for (int i = 0; i < k; ++i) {
                    const float* p = in + ((size_t)(y * k + j) * w + (size_t)x * k + i) * 4;
                    a[0] = a[0] + p[0]; a[1] = a[1] + p[1]; a[2] = a[2] + p[2];
                }
            float* o = out + ((size_t)y * ow + x) * 4;
            o[0] = a[0] * inv; o[1] = a[1] * inv; o[2] = a[2] * inv; o[3] = 1.0f;
        }
    return 0;
}

} // extern "C"
