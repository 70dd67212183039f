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
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;               // optional queries (vqhip_comm_query)
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
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
        r.CommAbort = (decltype(r.CommAbort))dlsym(r.lib, "ncclCommAbort");
        r.GetVersion = (decltype(r.GetVersion))dlsym(r.lib, "ncclGetVersion");
        r.CommCount = (decltype(r.CommCount))dlsym(r.lib, "ncclCommCount");
        r.CommUserRank = (decltype(r.CommUserRank))dlsym(r.lib, "ncclCommUserRank");
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
    // dense staging block for the boundary rows of a PITCHED tile (2 x 10 rows): the message structure must not depend on the local pitch —
    // RCCL matches sends and receives one to one, in posting order, with equal counts, and the neighbour always posts ONE receive per halo
    char* stage = nullptr; size_t stageBytes = 0;
    hipEvent_t stageFree = nullptr; bool stageUsed = false;  // recorded after the sends that read the block: the next pack waits for it
};

namespace {

// a dense [n][rowB] block at c->stage + offset holding rows [r0, r0+n) of a tile whose rows are pitchB bytes apart
int packRows(Rccl& r, vqhip_comm* c, hipStream_t st, const char* base, int r0, int n, size_t rowB, size_t pitchB, size_t offset) {
    if (r.hostBuffers) { for (int y = 0; y < n; ++y) std::memcpy(c->stage + offset + (size_t)y * rowB, base + (size_t)(r0 + y) * pitchB, rowB); return VQHIP_OK; }
    const hipError_t e = hipMemcpy2DAsync(c->stage + offset, rowB, base + (size_t)r0 * pitchB, pitchB, rowB, (size_t)n, hipMemcpyDeviceToDevice, st);
    return e == hipSuccess ? VQHIP_OK : vqk::fail_global(VQHIP_ERR_HIP, std::string("vqhip_exchange_blur_halos: hipMemcpy2DAsync: ") + hipGetErrorString(e));
}
int ensureStage(Rccl& r, vqhip_comm* c, hipStream_t st, size_t bytes) {
    if (c->stageUsed && !r.hostBuffers) {                   // the previous exchange's sends may still be reading the block (possibly on another stream)
        const hipError_t e = hipStreamWaitEvent(st, c->stageFree, 0);
        if (e != hipSuccess) return vqk::fail_global(VQHIP_ERR_HIP, std::string("vqhip_exchange_blur_halos: hipStreamWaitEvent: ") + hipGetErrorString(e));
    }
    if (c->stage && c->stageBytes >= bytes) return VQHIP_OK;
    c->stageBytes = 0;                                       // the old block goes: a failed allocation below must not leave its size behind
    if (r.hostBuffers) { std::free(c->stage); c->stage = (char*)std::malloc(bytes); if (!c->stage) return vqk::fail_global(VQHIP_ERR_HIP, "staging allocation failed"); }
    else {
        if (c->stage) { (void)hipEventSynchronize(c->stageFree); (void)hipFree(c->stage); c->stage = nullptr; }
        hipError_t e = hipMalloc((void**)&c->stage, bytes);
        if (e == hipSuccess && !c->stageFree) e = hipEventCreateWithFlags(&c->stageFree, hipEventDisableTiming);
        if (e != hipSuccess) return vqk::fail_global(VQHIP_ERR_HIP, std::string("vqhip_exchange_blur_halos: staging allocation: ") + hipGetErrorString(e));
    }
    c->stageBytes = bytes;
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
    if (c->stage) { if (rccl().hostBuffers) std::free(c->stage); else { if (c->stageFree) (void)hipEventSynchronize(c->stageFree); (void)hipFree(c->stage); } }
    if (c->stageFree) (void)hipEventDestroy(c->stageFree);
    delete c;
}

int vqhip_comm_abort(vqhip_comm* c) {
    if (!c) return vqk::fail_global(VQHIP_ERR_INVALID_ARG, "vqhip_comm_abort: comm is NULL");
    Rccl& r = rccl();
    ncclResult_t e = ncclSuccess;
    if (c->owned && c->comm) { e = r.CommAbort ? r.CommAbort(c->comm) : (r.CommDestroy ? r.CommDestroy(c->comm) : ncclSuccess); c->comm = nullptr; }
    vqhip_comm_destroy(c);                                  // the staging block and its event (c->comm is gone: no second destroy)
    return e == ncclSuccess ? VQHIP_OK : failRccl("ncclCommAbort", e);
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
    // my first 10 rows are the bottom halo of the tile above, my last 10 rows the top halo of the tile below; halo buffers are dense.
    // One send and one receive of 10 dense rows per neighbour: a pitched tile's rows are packed first.
    const size_t haloB = (size_t)VQHIP_HALO_ROWS * rowB;
    const char* sendUp = src;
    const char* sendDown = src + (size_t)(tile_rows - VQHIP_HALO_ROWS) * pitchB;
    const bool pitched = rowB != pitchB;
    if (pitched) {
        int rc = ensureStage(r, c, st, 2 * haloB);
        if (rc == VQHIP_OK && up)   rc = packRows(r, c, st, src, 0, VQHIP_HALO_ROWS, rowB, pitchB, 0);
        if (rc == VQHIP_OK && down) rc = packRows(r, c, st, src, tile_rows - VQHIP_HALO_ROWS, VQHIP_HALO_ROWS, rowB, pitchB, haloB);
        if (rc != VQHIP_OK) return rc;
        sendUp = c->stage; sendDown = c->stage + haloB;
    }
    RCCL_TRY(r.GroupStart());
    ncclResult_t e = ncclSuccess;
    if (up)                      { e = r.Send(sendUp, haloB, ncclUint8, c->rank - 1, c->comm, st);
                                   if (e == ncclSuccess) e = r.Recv(halo_top, haloB, ncclUint8, c->rank - 1, c->comm, st); }
    if (down && e == ncclSuccess) { e = r.Send(sendDown, haloB, ncclUint8, c->rank + 1, c->comm, st);
                                   if (e == ncclSuccess) e = r.Recv(halo_bottom, haloB, ncclUint8, c->rank + 1, c->comm, st); }
    const ncclResult_t eg = r.GroupEnd();
    if (e != ncclSuccess) return failRccl("ncclSend / ncclRecv (blur halos)", e);
    if (eg != ncclSuccess) return failRccl("ncclGroupEnd", eg);
    if (pitched && !r.hostBuffers) {
        const hipError_t he = hipEventRecord(c->stageFree, st);
        if (he != hipSuccess) return vqk::fail_global(VQHIP_ERR_HIP, std::string("vqhip_exchange_blur_halos: hipEventRecord: ") + hipGetErrorString(he));
        c->stageUsed = true;
    }
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
    ncclResult_t e = ncclSuccess;
    for (int k = 0; k < c->world && e == ncclSuccess; ++k) {
        if (k == c->rank) continue;
        if (root == VQHIP_ALL_RANKS || root == k)          // k receives my tile: one message
            e = r.Send(tile, (size_t)rows[c->rank] * rowB, ncclUint8, k, c->comm, st);
        if (e == ncclSuccess && receiver)                   // I receive k's tile straight into its place in the frame
            e = r.Recv((char*)frame + (size_t)row0[k] * rowB, (size_t)rows[k] * rowB, ncclUint8, k, c->comm, st);
    }
    const ncclResult_t eg = r.GroupEnd();
    if (e != ncclSuccess) return failRccl("ncclSend / ncclRecv (composite)", e);
    if (eg != ncclSuccess) return failRccl("ncclGroupEnd", eg);
    return VQHIP_OK;
}

int vqhip_comm_loopback(vqhip_comm* c, void* stream, const void* src, void* dst, size_t bytes) {
    if (!c || !c->comm || !src || !dst || bytes == 0) return vqk::fail_global(VQHIP_ERR_INVALID_ARG, "vqhip_comm_loopback: bad arguments");
    Rccl& r = rccl();
    hipStream_t st = (hipStream_t)stream;
    RCCL_TRY(r.GroupStart());                               // a send and a receive addressed to the own rank, posted together
    ncclResult_t e = r.Send(src, bytes, ncclUint8, c->rank, c->comm, st);
    if (e == ncclSuccess) e = r.Recv(dst, bytes, ncclUint8, c->rank, c->comm, st);
    const ncclResult_t eg = r.GroupEnd();
    if (e != ncclSuccess) return failRccl("ncclSend / ncclRecv (loopback)", e);
    if (eg != ncclSuccess) return failRccl("ncclGroupEnd", eg);
    return VQHIP_OK;
}

int vqhip_comm_query(const vqhip_comm* c, vqhip_comm_info* out) {
    if (!c || !out) return vqk::fail_global(VQHIP_ERR_INVALID_ARG, "vqhip_comm_query: NULL argument");
    Rccl& r = rccl();
    std::memset(out, 0, sizeof(*out));
    out->world = c->world; out->rank = c->rank;
    out->nranks_seen = out->rank_seen = out->rccl_version = -1;
    int v = 0;
    if (r.GetVersion && r.GetVersion(&v) == ncclSuccess) out->rccl_version = v;
    if (r.CommCount && c->comm && r.CommCount(c->comm, &v) == ncclSuccess) out->nranks_seen = v;
    if (r.CommUserRank && c->comm && r.CommUserRank(c->comm, &v) == ncclSuccess) out->rank_seen = v;
    Dl_info di;
    if (r.Send && dladdr((void*)r.Send, &di) && di.dli_fname) std::snprintf(out->library_path, sizeof(out->library_path), "%s", di.dli_fname);
    return VQHIP_OK;
}

} // extern "C"
