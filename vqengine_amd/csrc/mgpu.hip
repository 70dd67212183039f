// mgpu.hip — row-tiled multi-GPU frame behind the C ABI (the reference has no analogue: it renders on one adapter,
// Source/Renderer/Rendering/SceneRendering.cpp:2507 is the single-GPU post chain this mode cuts into row tiles).
//
// One process per GPU. Shade, X blur and tonemap need no communication; exactly two exchanges touch the data path:
//   1. blur halo : the Y pass needs KERNEL_RANGE-1 = 10 X-blurred rows of each vertical neighbour (GaussianBlur.hlsl:54-55,176-180):
//                  one grouped ncclSend/ncclRecv pair per neighbour — every GPU pair of an MI355X node has its own xGMI link;
//   2. composite : the RGBA8 tiles are collected on the presenting rank (or on every rank) by grouped point-to-point transfers, so the
//                  root receives over its world-1 direct links at once instead of around a ring.
// Both are enqueued on the caller's stream (RCCL stream semantics) and return without synchronising, like every other vqhip_* call.
//
// RCCL is bound at run time (dlopen of librccl.so.1, or of $VQHIP_RCCL_LIBRARY): libvqhip.so keeps depending on the HIP runtime only,
// single-GPU hosts never load RCCL, and a host that already holds an ncclComm_t (vqhip_comm_adopt) shares whichever copy it loaded.
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include <rccl/rccl.h>
#include "vq_internal.h"

namespace vqk { int fail_global(int code, const std::string& msg); }

namespace {

struct Rccl {
    void* lib = nullptr;
    bool hostBuffers = false;        // tests/cpp/mock_rccl.cpp on a box without a GPU: the "device" pointers are host memory
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* env = std::getenv("VQHIP_RCCL_LIBRARY");
        const char* names[] = { env, "librccl.so.1", "librccl.so" };
        for (const char* n : names) {
            if (!n || !*n) continue;
            r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
            r.error = std::string("dlopen(") + n + "): " + dlerror();
            if (n == env) return;                          // an explicit override that fails is an error, not a reason to look elsewhere
        }
        if (!r.lib) return;
        bool ok = true;
        auto sym = [&](const char* name) { void* p = dlsym(r.lib, name); if (!p) { ok = false; r.error = std::string("RCCL symbol missing: ") + name; } return p; };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        r.hostBuffers = dlsym(r.lib, "vqmock_rccl_host_buffers") != nullptr && ((int (*)())dlsym(r.lib, "vqmock_rccl_host_buffers"))() != 0;
        if (!ok) { dlclose(r.lib); r.lib = nullptr; }
    });
    return r;
}

int failRccl(const char* what, ncclResult_t e) {
    Rccl& r = rccl();
    return vqk::fail_global(VQHIP_ERR_RCCL, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(e) : "RCCL error") + " (" + std::to_string((int)e) + ")");
}
#define RCCL_TRY(call) do { ncclResult_t e_ = (call); if (e_ != ncclSuccess) return failRccl(#call, e_); } while (0)

size_t bytesPerPixel(int fmt) {
    switch (fmt) { case VQHIP_FMT_RGBA32F: return 16; case VQHIP_FMT_RGBA16F: return 8; case VQHIP_FMT_RGBA8_UNORM: return 4;
                   case VQHIP_FMT_RG16F: return 4; case VQHIP_FMT_RG32F: return 8; }
    return 0;
}

} // namespace

struct vqhip_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    bool owned = false;
};

namespace {

// rows [r0, r0+n) of an image whose rows are `pitchB` bytes apart and `rowB` bytes long: one message when dense, one per row otherwise
int sendRows(Rccl& r, vqhip_comm* c, hipStream_t st, const char* base, int r0, int n, size_t rowB, size_t pitchB, int peer) {
    if (rowB == pitchB) { RCCL_TRY(r.Send(base + (size_t)r0 * pitchB, (size_t)n * rowB, ncclUint8, peer, c->comm, st)); return VQHIP_OK; }
    for (int y = 0; y < n; ++y) RCCL_TRY(r.Send(base + (size_t)(r0 + y) * pitchB, rowB, ncclUint8, peer, c->comm, st));
    return VQHIP_OK;
}
int recvRows(Rccl& r, vqhip_comm* c, hipStream_t st, char* base, int r0, int n, size_t rowB, size_t pitchB, int peer) {
    if (rowB == pitchB) { RCCL_TRY(r.Recv(base + (size_t)r0 * pitchB, (size_t)n * rowB, ncclUint8, peer, c->comm, st)); return VQHIP_OK; }
    for (int y = 0; y < n; ++y) RCCL_TRY(r.Recv(base + (size_t)(r0 + y) * pitchB, rowB, ncclUint8, peer, c->comm, st));
    return VQHIP_OK;
}

} // namespace

extern "C" {

int vqhip_rowtile(int frame_height, int world, int rank, int* row0, int* rows) {
    if (frame_height <= 0 || world <= 0 || rank < 0 || rank >= world || !row0 || !rows)
        return vqk::fail_global(VQHIP_ERR_INVALID_ARG, "vqhip_rowtile: bad arguments");
    const int q = frame_height / world, rem = frame_height % world;           // the first `rem` ranks own one row more
    *rows = q + (rank < rem ? 1 : 0);
    *row0 = rank * q + (rank < rem ? rank : rem);
    if (world > 1 && q < VQHIP_HALO_ROWS)
        return vqk::fail_global(VQHIP_ERR_INVALID_ARG, "vqhip_rowtile: tiles must be at least 10 rows tall (the blur halo comes from the direct neighbour only)");
    return VQHIP_OK;
}

int vqhip_comm_unique_id(void* id128) {
    if (!id128) return vqk::fail_global(VQHIP_ERR_INVALID_ARG, "vqhip_comm_unique_id: id is NULL");
    Rccl& r = rccl();
    if (!r.lib) return vqk::fail_global(VQHIP_ERR_RCCL, "RCCL is not available: " + r.error);
    static_assert(sizeof(ncclUniqueId) == VQHIP_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    RCCL_TRY(r.GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    return VQHIP_OK;
}

int vqhip_comm_create(const void* id128, int world, int rank, vqhip_comm** out) {
    if (!id128 || !out || world <= 0 || rank < 0 || rank >= world) return vqk::fail_global(VQHIP_ERR_INVALID_ARG, "vqhip_comm_create: bad arguments");
    Rccl& r = rccl();
    if (!r.lib) return vqk::fail_global(VQHIP_ERR_RCCL, "RCCL is not available: " + r.error);
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    vqhip_comm* c = new vqhip_comm;
    c->world = world; c->rank = rank; c->owned = true;
    const ncclResult_t e = r.CommInitRank(&c->comm, world, id, rank);        // collective over all ranks, on the calling thread's current device
    if (e != ncclSuccess) { delete c; return failRccl("ncclCommInitRank", e); }
    *out = c;
    return VQHIP_OK;
}

int vqhip_comm_adopt(void* nccl_comm, int world, int rank, vqhip_comm** out) {
    if (!nccl_comm || !out || world <= 0 || rank < 0 || rank >= world) return vqk::fail_global(VQHIP_ERR_INVALID_ARG, "vqhip_comm_adopt: bad arguments");
    Rccl& r = rccl();
    if (!r.lib) return vqk::fail_global(VQHIP_ERR_RCCL, "RCCL is not available: " + r.error);
    vqhip_comm* c = new vqhip_comm;
    c->comm = (ncclComm_t)nccl_comm; c->world = world; c->rank = rank; c->owned = false;
    *out = c;
    return VQHIP_OK;
}

void vqhip_comm_destroy(vqhip_comm* c) {
    if (!c) return;
    if (c->owned && c->comm && rccl().CommDestroy) rccl().CommDestroy(c->comm);
    delete c;
}

int vqhip_exchange_blur_halos(vqhip_comm* c, void* stream, const void* xblur_tile, int width, int tile_rows, int row_pitch_px,
                              vqhip_format fmt, void* halo_top, void* halo_bottom) {
    if (!c || !xblur_tile || width <= 0 || tile_rows <= 0 || row_pitch_px < width) return vqk::fail_global(VQHIP_ERR_INVALID_ARG, "vqhip_exchange_blur_halos: bad arguments");
    if (fmt != VQHIP_FMT_RGBA16F && fmt != VQHIP_FMT_RGBA32F) return vqk::fail_global(VQHIP_ERR_UNSUPPORTED, "vqhip_exchange_blur_halos: fmt must be RGBA16F or RGBA32F");
    const bool up = c->rank > 0, down = c->rank < c->world - 1;
    if ((up && !halo_top) || (down && !halo_bottom)) return vqk::fail_global(VQHIP_ERR_INVALID_ARG, "vqhip_exchange_blur_halos: a halo buffer of an inner tile edge is NULL");
    if ((up || down) && tile_rows < VQHIP_HALO_ROWS) return vqk::fail_global(VQHIP_ERR_INVALID_ARG, "vqhip_exchange_blur_halos: tile shorter than the halo");
    if (!up && !down) return VQHIP_OK;
    vqk::Range range_("BlurHaloExchange");
    Rccl& r = rccl();
    hipStream_t st = (hipStream_t)stream;
    const size_t bpp = bytesPerPixel(fmt), rowB = (size_t)width * bpp, pitchB = (size_t)row_pitch_px * bpp;
    const char* src = (const char*)xblur_tile;
    RCCL_TRY(r.GroupStart());
    int rc = VQHIP_OK;
    // my first 10 rows are the bottom halo of the tile above, my last 10 rows the top halo of the tile below; halo buffers are dense
    if (up)   { rc = sendRows(r, c, st, src, 0, VQHIP_HALO_ROWS, rowB, pitchB, c->rank - 1);
                if (rc == VQHIP_OK) rc = recvRows(r, c, st, (char*)halo_top, 0, VQHIP_HALO_ROWS, rowB, rowB, c->rank - 1); }
    if (down && rc == VQHIP_OK) { rc = sendRows(r, c, st, src, tile_rows - VQHIP_HALO_ROWS, VQHIP_HALO_ROWS, rowB, pitchB, c->rank + 1);
                if (rc == VQHIP_OK) rc = recvRows(r, c, st, (char*)halo_bottom, 0, VQHIP_HALO_ROWS, rowB, rowB, c->rank + 1); }
    const ncclResult_t e = r.GroupEnd();
    if (rc != VQHIP_OK) return rc;
    if (e != ncclSuccess) return failRccl("ncclGroupEnd", e);
    return VQHIP_OK;
}

int vqhip_composite_tiles(vqhip_comm* c, void* stream, const void* tile, int width, int frame_height, vqhip_format fmt, int root, void* frame) {
    if (!c || !tile || width <= 0 || frame_height <= 0 || root < VQHIP_ALL_RANKS || root >= c->world)
        return vqk::fail_global(VQHIP_ERR_INVALID_ARG, "vqhip_composite_tiles: bad arguments");
    const size_t bpp = bytesPerPixel(fmt);
    if (!bpp) return vqk::fail_global(VQHIP_ERR_UNSUPPORTED, "vqhip_composite_tiles: unknown format");
    vqk::Range range_("CompositeTiles");
    const bool receiver = root == VQHIP_ALL_RANKS || root == c->rank;
    if (receiver && !frame) return vqk::fail_global(VQHIP_ERR_INVALID_ARG, "vqhip_composite_tiles: frame is NULL on a receiving rank");
    Rccl& r = rccl();
    hipStream_t st = (hipStream_t)stream;
    const size_t rowB = (size_t)width * bpp;
    std::vector<int> row0(c->world), rows(c->world);
    for (int k = 0; k < c->world; ++k) { const int rc = vqhip_rowtile(frame_height, c->world, k, &row0[k], &rows[k]); if (rc != VQHIP_OK && c->world > 1) return rc; }
    if (receiver) {                                         // own tile: a device-to-device copy on the same stream
        char* dst = (char*)frame + (size_t)row0[c->rank] * rowB;
        const size_t n = (size_t)rows[c->rank] * rowB;
        if (dst != (const char*)tile) {
            if (r.hostBuffers) std::memcpy(dst, tile, n);
            else { const hipError_t he = hipMemcpyAsync(dst, tile, n, hipMemcpyDeviceToDevice, st);
                   if (he != hipSuccess) return vqk::fail_global(VQHIP_ERR_HIP, std::string("vqhip_composite_tiles: hipMemcpyAsync: ") + hipGetErrorString(he)); }
        }
    }
    if (c->world == 1) return VQHIP_OK;
    RCCL_TRY(r.GroupStart());
    int rc = VQHIP_OK;
    for (int k = 0; k < c->world && rc == VQHIP_OK; ++k) {
        if (k == c->rank) continue;
        if (root == VQHIP_ALL_RANKS || root == k)          // k receives my tile
            rc = sendRows(r, c, st, (const char*)tile, 0, rows[c->rank], rowB, rowB, k);
        if (rc == VQHIP_OK && receiver)                     // I receive k's tile straight into its place in the frame
            rc = recvRows(r, c, st, (char*)frame, row0[k], rows[k], rowB, rowB, k);
    }
    const ncclResult_t e = r.GroupEnd();
    if (rc != VQHIP_OK) return rc;
    if (e != ncclSuccess) return failRccl("ncclGroupEnd", e);
    return VQHIP_OK;
}

} // extern "C"
