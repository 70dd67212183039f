// capi.hip — the C ABI of include/vqhip.h: argument validation, the per-call constant ring (the
// analogue of the reference's DynamicBufferHeap upload heap) and kernel dispatch. There is no CPU
// fallback anywhere in this library: without a gfx950 device vqhip_create() fails and every other
// entry point needs a context.
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <atomic>
#include <string>
#include <thread>
#include <vector>
#include "vq_internal.h"
#include "vq_devmath.h"      // vqd::sincos_ is __host__ __device__: frame-uniform trigonometry is evaluated here, once

using namespace vqk;

struct vqhip_ctx {
    int device = 0;
    int nCUs = 256;
    int ldsPerBlock = 64 * 1024;   // hipDeviceAttributeMaxSharedMemoryPerBlock (163 840 on gfx950): what a kernel may opt into, queried once in vqhip_create
    static constexpr int kSlots = 32;
    char* hostRing = nullptr;      // pinned
    char* devRing = nullptr;
    hipEvent_t slotEvent[kSlots] = {};
    hipStream_t copyStream = nullptr;          // constants are uploaded on their own stream: the DMA of call n+1 overlaps the kernels of call n
    hipEvent_t copyEvent[kSlots] = {};
    bool slotBusy[kSlots] = {};
    int nextSlot = 0;
    void* scratch = nullptr; size_t scratchBytes = 0;
    // footprint records of the diffuse convolution (conv.hip:k_diffuse_records): rewritten by every vqhip_conv_diffuse; `recFree` is recorded behind the
    // kernel that reads them, and the next call's stream waits for it before it overwrites them (calls may come on different streams)
    void* rec = nullptr; size_t recBytes = 0; hipEvent_t recFree = nullptr; bool recUsed = false;
    int pow5ExpLog = 0;            // vqhip_set_fresnel_pow
    int arithDxc = 0;              // vqhip_set_arithmetic
    vqk::Options opt;              // vqhip_set_option
    // 65536-entry tonemap tables (post.hip:k_tonemap_lut), cached per (TonemapperParams, output format): the table is built once
    // per parameter set instead of once per frame. Streams that HIT a cached table only wait for the event of its build. A table is replaced
    // only when a fifth parameter set shows up: the least recently used one goes (a hit counts as a use). Every use records the slot's `lastUse` event
    // behind the kernel that read the table; the rebuild makes ITS stream wait for that event — no host stall, legal under stream capture. Only a table
    // that has been read from MORE THAN ONE stream (one event cannot cover readers on several streams) is replaced after a device-wide wait.
    static constexpr int kLuts = 4;
    struct TonemapLut { void* table = nullptr; VQ_TonemapperParams key{}; int outFmt = -1; bool valid = false;
                        hipEvent_t built = nullptr, lastUse = nullptr; hipStream_t lastStream = nullptr; bool used = false, multiStream = false, everBuilt = false;
                        uint64_t lastUseTick = 0; } lut[kLuts];
    uint64_t lutTick = 0;
    // one thread at a time per context: entry points detect a second thread inside the same context and refuse it
    std::atomic<uint64_t> owner{0};            // token of the thread inside an entry point (0: nobody); acquired by CAS, cleared by the outermost guard
    int ownerDepth = 0;                        // nesting depth: touched by the owner only
    uint64_t generation = 0;                   // distinguishes this context from an earlier one at the same address (vqhip_last_error)
    std::string lastError;
};

namespace {

thread_local std::string g_lastError;
thread_local const vqhip_ctx* g_lastErrorCtx = nullptr;     // the context this thread's most recent failure belongs to (vqhip_last_error) ...
thread_local uint64_t g_lastErrorGen = 0;                   // ... and its generation: a later context at the same address does not inherit the message
std::atomic<uint64_t> g_ctxGeneration{1};

int fail(vqhip_ctx* ctx, int code, const std::string& msg) {
    g_lastError = msg;
    g_lastErrorCtx = ctx;
    g_lastErrorGen = ctx ? ctx->generation : 0;
    if (ctx) ctx->lastError = msg;
    return code;
}
// a call refused because ANOTHER thread is inside the context: the message stays with the refused thread, the context is not written to
int failRefused(const vqhip_ctx* ctx, const std::string& msg) {
    g_lastError = msg;
    g_lastErrorCtx = ctx;
    g_lastErrorGen = ctx ? ctx->generation : 0;
    return VQHIP_ERR_INVALID_ARG;
}
int failHip(vqhip_ctx* ctx, hipError_t e, const char* what) {
    return fail(ctx, VQHIP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return failHip(ctx, e_, #call); } while (0)

bool isImageFmt(int f) { return f == VQHIP_FMT_RGBA32F || f == VQHIP_FMT_RGBA16F; }

// Acquire the next constant-ring slot (waits for the kernel that last read it, if still in flight).
int acquireSlot(vqhip_ctx* ctx, int* slot) {
    const int s = ctx->nextSlot;
    ctx->nextSlot = (s + 1) % vqhip_ctx::kSlots;
    if (ctx->slotBusy[s]) { HIP_TRY(ctx, hipEventSynchronize(ctx->slotEvent[s])); ctx->slotBusy[s] = false; }
    *slot = s;
    return VQHIP_OK;
}
int commitSlot(vqhip_ctx* ctx, int slot, size_t bytes, hipStream_t st) {
    // The slot is free (acquireSlot waited for its last reader), so the copy need not queue behind the caller's stream: it
    // runs on the context's copy stream right away and the caller's stream only waits for its completion event.
    HIP_TRY(ctx, hipMemcpyAsync(ctx->devRing + (size_t)slot * kConstSlotBytes, ctx->hostRing + (size_t)slot * kConstSlotBytes, bytes, hipMemcpyHostToDevice, ctx->copyStream));
    HIP_TRY(ctx, hipEventRecord(ctx->copyEvent[slot], ctx->copyStream));
    HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->copyEvent[slot], 0));
    return VQHIP_OK;
}
int releaseSlot(vqhip_ctx* ctx, int slot, hipStream_t st) {
    HIP_TRY(ctx, hipEventRecord(ctx->slotEvent[slot], st));
    ctx->slotBusy[slot] = true;
    return VQHIP_OK;
}

int ensureScratch(vqhip_ctx* ctx, size_t bytes) {
    if (ctx->scratchBytes >= bytes) return VQHIP_OK;
    if (ctx->scratch) { HIP_TRY(ctx, hipDeviceSynchronize()); HIP_TRY(ctx, hipFree(ctx->scratch)); ctx->scratch = nullptr; ctx->scratchBytes = 0; }
    HIP_TRY(ctx, hipMalloc(&ctx->scratch, bytes));
    ctx->scratchBytes = bytes;
    return VQHIP_OK;
}

// The tonemap table of (p, outFmt), ready to be read by work enqueued on `st` after this call: a cached table makes `st` wait for the
// event of its build (it may have happened on another stream); a miss rebuilds the LEAST RECENTLY USED slot on `st` — behind the slot's last-use
// event when all its readers were on one stream, after a device-wide wait when they were on several (releaseTonemapLut keeps the record).
int acquireTonemapLut(vqhip_ctx* ctx, hipStream_t st, const VQ_TonemapperParams& p, int outFmt, int* slotOut) {
    int victim = 0;
    for (int i = 0; i < vqhip_ctx::kLuts; ++i) {
        auto& L = ctx->lut[i];
        if (L.valid && L.outFmt == outFmt && std::memcmp(&L.key, &p, sizeof(p)) == 0) {
            HIP_TRY(ctx, hipStreamWaitEvent(st, L.built, 0));
            L.lastUseTick = ++ctx->lutTick;
            *slotOut = i;
            return VQHIP_OK;
        }
        if (!L.valid) { if (ctx->lut[victim].valid) victim = i; }
        else if (ctx->lut[victim].valid && L.lastUseTick < ctx->lut[victim].lastUseTick) victim = i;
    }
    auto& L = ctx->lut[victim];
    // Whatever state an earlier failure left the slot in (valid or not), the rebuild must queue behind every kernel that may still read the table and
    // behind the last build that wrote it: `used` / `lastUse` / `built` survive an invalidation and are only reset once the NEW build is enqueued.
    if (L.used) {
        if (L.multiStream) HIP_TRY(ctx, hipDeviceSynchronize());            // readers on several streams: one event does not cover them (documented in vqhip.h)
        else HIP_TRY(ctx, hipStreamWaitEvent(st, L.lastUse, 0));              // the rebuild queues behind the last kernel that read the table
    } else if (L.everBuilt) {
        HIP_TRY(ctx, hipStreamWaitEvent(st, L.built, 0));                     // built (possibly on another stream) and never read: write after write
    }
    L.valid = false;
    hipError_t e = launch_tonemap_lut_build(st, L.table, p, outFmt);
    if (e != hipSuccess) return failHip(ctx, e, "tonemap table build launch");
    HIP_TRY(ctx, hipEventRecord(L.built, st));
    L.key = p; L.outFmt = outFmt; L.valid = true; L.everBuilt = true; L.used = false; L.multiStream = false; L.lastStream = nullptr; L.lastUseTick = ++ctx->lutTick;
    *slotOut = victim;
    return VQHIP_OK;
}
// called behind the kernel that read table `slot` on `st`
int releaseTonemapLut(vqhip_ctx* ctx, hipStream_t st, int slot) {
    auto& L = ctx->lut[slot];
    if (L.used && L.lastStream != st) L.multiStream = true;
    HIP_TRY(ctx, hipEventRecord(L.lastUse, st));
    L.used = true; L.lastStream = st;
    return VQHIP_OK;
}

// RAII marker of "this thread is inside an entry point of ctx". A second THREAD entering the same context while one is inside is refused
// (VQHIP_ERR_INVALID_ARG) instead of corrupting the constant ring / table cache; nested calls of the same thread (vqhip_post_process ->
// vqhip_gaussian_blur_x) pass. The context may migrate between threads, it just cannot be shared at the same time.
// Ownership is ONE atomic word holding a per-thread token: a thread enters by CAS 0 -> its token, so there is no window in which another thread can
// read a stale owner; a nested call recognises its own token and bumps a depth counter only the owner touches; the outermost guard clears the word.
uint64_t threadToken() {
    static std::atomic<uint64_t> next{1};
    thread_local const uint64_t t = next.fetch_add(1, std::memory_order_relaxed);
    return t;
}
struct CtxGuard {
    vqhip_ctx* c; bool ok;
    explicit CtxGuard(vqhip_ctx* ctx) : c(ctx), ok(true) {
        if (!c) return;
        const uint64_t me = threadToken();
        if (c->owner.load(std::memory_order_acquire) == me) { ++c->ownerDepth; return; }      // nested call of the owner
        uint64_t expected = 0;
        if (!c->owner.compare_exchange_strong(expected, me, std::memory_order_acq_rel)) { ok = false; c = nullptr; return; }
        c->ownerDepth = 1;
    }
    ~CtxGuard() { if (c && --c->ownerDepth == 0) c->owner.store(0, std::memory_order_release); }
};
#define CTX_GUARD(ctx, who) CtxGuard guard_(ctx); if (!guard_.ok) return failRefused(ctx, std::string(who) + ": the context is in use on another thread (one thread at a time per vqhip_ctx)")

int mipDim(int d0, int l) { int d = d0 >> l; return d < 1 ? 1 : d; }

// Smallest float t >= 0 with sqrtf(t) >= range: `length(Lw - P) < range` (Lighting.hlsl:318) <=> `dot(d,d) < t` exactly, because the
// correctly rounded sqrt is monotonic. NaN range -> NaN (never lit); range <= 0 -> 0 (never lit); +inf -> +inf (every finite distance).
float rangeCullThreshold(float range) {
    if (!(range == range)) return range;
    if (range <= 0.0f) return 0.0f;
    if (range == INFINITY) return INFINITY;
    const double r2 = (double)range * (double)range;
    float t = r2 >= (double)FLT_MAX ? FLT_MAX : (float)r2;
    while (t > 0.0f && sqrtf(t) >= range) t = nextafterf(t, 0.0f);           // now sqrtf(t) < range (or t == 0)
    while (sqrtf(t) < range) { if (t == FLT_MAX) return INFINITY; t = nextafterf(t, INFINITY); }
    return t;
}

} // namespace

namespace vqk { int fail_global(int code, const std::string& msg) { return fail(nullptr, code, msg); } }     // for mgpu.hip (no context)

namespace vqk {
namespace {
struct Roctx { int (*push)(const char*) = nullptr; int (*pop)() = nullptr; };
const Roctx& roctx() {
    static const Roctx r = [] {
        Roctx x;
        const char* env = std::getenv("VQHIP_ROCTX");
        if (env && env[0] == '0') return x;
        for (const char* n : { "librocprofiler-sdk-roctx.so.1", "libroctx64.so.4", "libroctx64.so" }) {
            if (void* lib = dlopen(n, RTLD_NOW | RTLD_LOCAL)) {
                x.push = (int (*)(const char*))dlsym(lib, "roctxRangePushA");
                x.pop = (int (*)())dlsym(lib, "roctxRangePop");
                if (x.push && x.pop) return x;
                x = Roctx{};
            }
        }
        return x;
    }();
    return r;
}
} // namespace
Range::Range(const char* name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
Range::~Range() { if (on) roctx().pop(); }
} // namespace vqk

extern "C" {

int vqhip_abi_version(void) { return VQHIP_ABI_VERSION; }

// the calling thread's own most recent failure when it belongs to `ctx` (race-free, and the only record of a refused concurrent call), else the context's
const char* vqhip_last_error(const vqhip_ctx* ctx) { return (!ctx || (g_lastErrorCtx == ctx && g_lastErrorGen == ctx->generation)) ? g_lastError.c_str() : ctx->lastError.c_str(); }

int vqhip_create(int device_ordinal, vqhip_ctx** out_ctx) {
    if (!out_ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "vqhip_create: out_ctx is NULL");
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(nullptr, VQHIP_ERR_NO_DEVICE, "vqhip_create: no HIP device visible (this library has no CPU fallback)");
    if (device_ordinal < 0 || device_ordinal >= n) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "vqhip_create: device ordinal out of range");
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device_ordinal));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, VQHIP_ERR_NO_DEVICE, std::string("vqhip_create: device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    HIP_TRY(nullptr, hipSetDevice(device_ordinal));
    vqhip_ctx* ctx = new vqhip_ctx();
    ctx->device = device_ordinal;
    ctx->nCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    {
        int lds = 0;
        if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device_ordinal) == hipSuccess && lds > 0) ctx->ldsPerBlock = lds;
    }
    ctx->generation = g_ctxGeneration.fetch_add(1, std::memory_order_relaxed);
    const size_t ringBytes = kConstSlotBytes * vqhip_ctx::kSlots;
    if ((e = hipHostMalloc((void**)&ctx->hostRing, ringBytes, hipHostMallocDefault)) != hipSuccess ||
        (e = hipMalloc((void**)&ctx->devRing, ringBytes)) != hipSuccess) {
        int rc = failHip(nullptr, e, "vqhip_create: constant ring allocation");
        vqhip_destroy(ctx);
        return rc;
    }
    for (int i = 0; i < vqhip_ctx::kLuts; ++i)
        if ((e = hipMalloc(&ctx->lut[i].table, 131072)) != hipSuccess || (e = hipEventCreateWithFlags(&ctx->lut[i].built, hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&ctx->lut[i].lastUse, hipEventDisableTiming)) != hipSuccess) { int rc = failHip(nullptr, e, "vqhip_create: tonemap tables"); vqhip_destroy(ctx); return rc; }
    for (int i = 0; i < vqhip_ctx::kSlots; ++i)
        if ((e = hipEventCreateWithFlags(&ctx->slotEvent[i], hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&ctx->copyEvent[i], hipEventDisableTiming)) != hipSuccess) { int rc = failHip(nullptr, e, "hipEventCreate"); vqhip_destroy(ctx); return rc; }
    if ((e = hipStreamCreateWithFlags(&ctx->copyStream, hipStreamNonBlocking)) != hipSuccess) { int rc = failHip(nullptr, e, "hipStreamCreate"); vqhip_destroy(ctx); return rc; }
    *out_ctx = ctx;
    return VQHIP_OK;
}

void vqhip_destroy(vqhip_ctx* ctx) {
    if (!ctx) return;
    if (g_lastErrorCtx == ctx) g_lastErrorCtx = nullptr;     // this thread's message dies with the context (other threads: the generation check)
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (int i = 0; i < vqhip_ctx::kSlots; ++i) { if (ctx->slotEvent[i]) (void)hipEventDestroy(ctx->slotEvent[i]); if (ctx->copyEvent[i]) (void)hipEventDestroy(ctx->copyEvent[i]); }
    if (ctx->copyStream) (void)hipStreamDestroy(ctx->copyStream);
    if (ctx->hostRing) (void)hipHostFree(ctx->hostRing);
    if (ctx->devRing) (void)hipFree(ctx->devRing);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->rec) (void)hipFree(ctx->rec);
    if (ctx->recFree) (void)hipEventDestroy(ctx->recFree);
    for (int i = 0; i < vqhip_ctx::kLuts; ++i) {
        if (ctx->lut[i].table) (void)hipFree(ctx->lut[i].table);
        if (ctx->lut[i].built) (void)hipEventDestroy(ctx->lut[i].built);
        if (ctx->lut[i].lastUse) (void)hipEventDestroy(ctx->lut[i].lastUse);
    }
    delete ctx;
}

// validation of the lighting half shared by vqhip_forward_lighting and vqhip_forward_lighting_from_materials; *casters = shadow casters present
static int validateLighting(vqhip_ctx* ctx, const char* who, const VQ_PerFrameData* perFrame, const VQ_PerViewLightingData* perView,
                            const VQ_PointLight* extraPoint, int numExtraPoint, const vqhip_envmap* env, const vqhip_shadowmaps* sm, vqhip_format outFmt, bool* casters) {
    const std::string w(who);
    if (!perFrame || !perView) return fail(ctx, VQHIP_ERR_INVALID_ARG, w + ": NULL argument");
    if (!isImageFmt(outFmt)) return fail(ctx, VQHIP_ERR_UNSUPPORTED, w + ": outFmt must be RGBA32F or RGBA16F");
    const VQ_SceneLighting& L = perFrame->Lights;
    if (L.numPointLights < 0 || L.numPointLights > VQ_NUM_LIGHTS__POINT || L.numSpotLights < 0 || L.numSpotLights > VQ_NUM_LIGHTS__SPOT ||
        L.numPointCasters < 0 || L.numPointCasters > VQ_NUM_SHADOWING_LIGHTS__POINT || L.numSpotCasters < 0 || L.numSpotCasters > VQ_NUM_SHADOWING_LIGHTS__SPOT)
        return fail(ctx, VQHIP_ERR_INVALID_ARG, w + ": light count exceeds the cbuffer array (LightingConstantBufferData.h:39-44)");
    if (numExtraPoint < 0 || numExtraPoint > kMaxExtraPointLights || (numExtraPoint > 0 && !extraPoint))
        return fail(ctx, VQHIP_ERR_INVALID_ARG, w + ": bad extraPoint / numExtraPoint");
    *casters = L.numPointCasters > 0 || L.numSpotCasters > 0 || (L.directional.enabled && L.directional.shadowing);
    if (*casters) {
        if (!sm) return fail(ctx, VQHIP_ERR_INVALID_ARG, w + ": shadow casters present but sm is NULL");
        if ((L.numPointCasters > 0 && (!sm->point || sm->point_dim <= 0)) || (L.numSpotCasters > 0 && (!sm->spot || sm->spot_dim <= 0)) ||
            (L.directional.enabled && L.directional.shadowing && (!sm->directional || sm->dir_dim <= 0)))
            return fail(ctx, VQHIP_ERR_INVALID_ARG, w + ": missing shadow map for a caster");
        // the PCF kernels address a slice with 32-bit byte / texel offsets (vq_shade.h: pcf_2d, omni_pcf): 2-D maps up to 32768^2, cube faces up to 16384^2
        if ((L.numPointCasters > 0 && sm->point_dim > 16384) || (L.numSpotCasters > 0 && sm->spot_dim > 32768) ||
            (L.directional.enabled && L.directional.shadowing && sm->dir_dim > 32768))
            return fail(ctx, VQHIP_ERR_UNSUPPORTED, w + ": shadow map larger than 32768^2 (2-D) / 16384^2 (cube face)");
    }
    if (env) {
        if (!env->diffuse_cube || env->diffuse_res <= 0) return fail(ctx, VQHIP_ERR_INVALID_ARG, w + ": env->diffuse_cube missing");
        if (!perView->EnvironmentMapDiffuseOnlyIllumination &&
            (!env->specular_cube || env->spec_res0 <= 0 || env->spec_mips <= 0 || (env->spec_res0 >> (env->spec_mips - 1)) < 1 || !env->brdf_lut || env->lut_size <= 0))
            return fail(ctx, VQHIP_ERR_INVALID_ARG, w + ": env specular cube / BRDF LUT missing");
    }
    return VQHIP_OK;
}
// the lighting pass's constant block (cbuffers b0 / b1 + descriptors + the packed point-light records) into a ring slot; returns its byte size
static size_t fillFrameConstants(vqhip_ctx* ctx, int slot, const VQ_PerFrameData* perFrame, const VQ_PerViewLightingData* perView,
                                 const VQ_PointLight* extraPoint, int numExtraPoint, const vqhip_envmap* env, const vqhip_shadowmaps* sm) {
    const VQ_SceneLighting& L = perFrame->Lights;
    FrameConstants* fc = (FrameConstants*)(ctx->hostRing + (size_t)slot * kConstSlotBytes);
    std::memset(fc, 0, sizeof(FrameConstants));
    fc->perFrame = *perFrame;
    fc->perView = *perView;
    if (env) fc->env = *env;
    if (sm) fc->sm = *sm;
    fc->hasEnv = env ? 1 : 0;
    fc->pow5ExpLog = ctx->pow5ExpLog;
    // GetHDRIRotationMatrix, Lighting.hlsl:348-358: a per-frame constant (the shader's own TODO: "pass m with cbuffer") — sin / cos are taken
    // correctly rounded (double libm rounded to float), like the oracle; the contract's polynomial cos is 1 ulp off at e.g. 0.3 rad
    fc->hdriSin = (float)std::sin((double)-perFrame->fHDRIOffsetInRadians);
    fc->hdriCos = (float)std::cos((double)-perFrame->fHDRIOffsetInRadians);
    // pack the non-shadowing point lights for the hot loop: cbuffer array first, then the extension array, in index order
    DevPointLight* pts = (DevPointLight*)(fc + 1);
    const int nPts = L.numPointLights + numExtraPoint;
    int32_t pointFastOK = 1, pointSkipOK = 1, negZeroAxes = 0;
    auto coordOK = [](float c) { const float m = std::fabs(c); return c == 0.0f || (m >= 0x1p-40f && m <= 0x1p40f); };      // false for NaN / inf
    for (int i = 0; i < nPts; ++i) {
        const VQ_PointLight& l = i < L.numPointLights ? L.point_lights[i] : extraPoint[i - L.numPointLights];
        pts[i].px = l.position.x; pts[i].py = l.position.y; pts[i].pz = l.position.z; pts[i].range = l.range;
        pts[i].cbx = l.color.x * l.brightness; pts[i].cby = l.color.y * l.brightness; pts[i].cbz = l.color.z * l.brightness;   // l.color * l.brightness (Lighting.hlsl:317)
        pts[i].rangeSq = rangeCullThreshold(l.range);
        if (pts[i].rangeSq > 0x1p60f) pointFastOK = 0;                           // false for a NaN threshold (never lit)
        // the light loop proves its fast quotients from the GRANULARITY of the coordinates (vq_shade.h:add_point_light): a non-zero Lw - P then has magnitude >= 2^-63
        if (!(coordOK(l.position.x) && coordOK(l.position.y) && coordOK(l.position.z))) pointFastOK = 0;
        if (l.position.x == 0.0f && std::signbit(l.position.x)) negZeroAxes |= 1;
        if (l.position.y == 0.0f && std::signbit(l.position.y)) negZeroAxes |= 2;
        if (l.position.z == 0.0f && std::signbit(l.position.z)) negZeroAxes |= 4;
        if (!(std::isfinite(pts[i].cbx) && std::isfinite(pts[i].cby) && std::isfinite(pts[i].cbz))) pointSkipOK = 0;
    }
    // spot / directional lights: the wave-uniform parts of SpotlightIntensity / CalculateDirectionalLightIllumination, with the shader's own operations
    // (this translation unit is built with -ffp-contract=off: every product and sum below is rounded on its own)
    const bool dxc = ctx->arithDxc != 0;
    auto normalizeAsShader = [dxc](const VQ_float3& v, float* o) {
        if (dxc) {                                                               // DXC reading: v * rsqrt(dot(v, v)), FMA-chain dot, correctly rounded rsqrt (vq_devmath.h:rsqrt_cr)
            const float dd = std::fma(v.z, v.z, std::fma(v.y, v.y, v.x * v.x));
            const float r = (float)(1.0 / std::sqrt((double)dd));
            o[0] = v.x * r; o[1] = v.y * r; o[2] = v.z * r;
        } else {                                                                 // literal reading: one IEEE quotient per component by length(v)
            const float dd = (v.x * v.x + v.y * v.y) + v.z * v.z;
            const float D = std::sqrt(dd);
            o[0] = v.x / D; o[1] = v.y / D; o[2] = v.z / D;
        }
    };
    for (int i = 0; i < VQ_NUM_LIGHTS__SPOT + VQ_NUM_SHADOWING_LIGHTS__SPOT; ++i) {
        const bool caster = i >= VQ_NUM_LIGHTS__SPOT;
        if (i >= (caster ? VQ_NUM_LIGHTS__SPOT + L.numSpotCasters : L.numSpotLights)) continue;
        const VQ_SpotLight& l = caster ? L.spot_casters[i - VQ_NUM_LIGHTS__SPOT] : L.spot_lights[i];
        DevSpotLight& ds = fc->spot[i];
        float sd[3];
        normalizeAsShader(l.spotDir, sd);
        ds.sdx = sd[0]; ds.sdy = sd[1]; ds.sdz = sd[2];
        const float den = l.outerConeAngle - l.innerConeAngle;
        ds.rConeDen = 1.0f / den;
        ds.cbx = l.color.x * l.brightness; ds.cby = l.color.y * l.brightness; ds.cbz = l.color.z * l.brightness;
        ds.flags = ((std::isnormal(den) && std::isnormal(ds.rConeDen)) ? 1 : 0) | ((std::isfinite(ds.cbx) && std::isfinite(ds.cby) && std::isfinite(ds.cbz)) ? 2 : 0);
    }
    {
        const VQ_float3 nd = { -L.directional.lightDirection.x, -L.directional.lightDirection.y, -L.directional.lightDirection.z };
        normalizeAsShader(nd, fc->dirWi);
    }
    fc->pointFastOK = pointFastOK;
    fc->pointSkipOK = pointSkipOK;
    fc->pointNegZeroAxes = negZeroAxes;
    fc->numPointAll = nPts;
    return sizeof(FrameConstants) + (size_t)nPts * sizeof(DevPointLight);
}

// vqhip_psmain_targets -> kernel arguments; NULL / no output bound: all-zero (the kernels then touch nothing)
static int fillMrt(vqhip_ctx* ctx, const char* who, const vqhip_psmain_targets* t, int width, MrtArgs* m) {
    std::memset(m, 0, sizeof(*m));
    if (!t || (!t->albedo_metallic && !t->motion_vectors)) return VQHIP_OK;
    const std::string w(who);
    if (t->albedo_metallic) {
        if (t->albedo_fmt != VQHIP_FMT_RGBA16F && t->albedo_fmt != VQHIP_FMT_RGBA32F) return fail(ctx, VQHIP_ERR_UNSUPPORTED, w + ": albedo_fmt must be RGBA16F or RGBA32F");
        m->albedoPitch = t->albedo_pitch_px ? t->albedo_pitch_px : width;
        if (m->albedoPitch < width) return fail(ctx, VQHIP_ERR_INVALID_ARG, w + ": albedo_pitch_px below the width");
        m->albedo = t->albedo_metallic; m->albedoF32 = t->albedo_fmt == VQHIP_FMT_RGBA32F;
    }
    if (t->motion_vectors) {
        if (t->motion_fmt != VQHIP_FMT_RG16F && t->motion_fmt != VQHIP_FMT_RG32F) return fail(ctx, VQHIP_ERR_UNSUPPORTED, w + ": motion_fmt must be RG16F or RG32F");
        if (!t->svPositionCurr || !t->svPositionPrev) return fail(ctx, VQHIP_ERR_INVALID_ARG, w + ": motion_vectors needs svPositionCurr and svPositionPrev");
        m->motionPitch = t->motion_pitch_px ? t->motion_pitch_px : width;
        m->svPitch = t->sv_pitch_px ? t->sv_pitch_px : width;
        if (m->motionPitch < width || m->svPitch < width) return fail(ctx, VQHIP_ERR_INVALID_ARG, w + ": motion_pitch_px / sv_pitch_px below the width");
        m->motion = t->motion_vectors; m->motionF32 = t->motion_fmt == VQHIP_FMT_RG32F;
        m->svCurr = (const float4*)t->svPositionCurr; m->svPrev = (const float4*)t->svPositionPrev;
    }
    return VQHIP_OK;
}

int vqhip_forward_lighting(vqhip_ctx* ctx, void* stream, const vqhip_gbuffer* gb,
        const VQ_PerFrameData* perFrame, const VQ_PerViewLightingData* perView,
        const VQ_PointLight* extraPoint, int numExtraPoint,
        const vqhip_envmap* env, const vqhip_shadowmaps* sm,
        void* out, int out_row_pitch_px, vqhip_format outFmt) {
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "forward_lighting: ctx is NULL");
    CTX_GUARD(ctx, "forward_lighting");
    return vqhip_forward_lighting_mrt(ctx, stream, gb, perFrame, perView, extraPoint, numExtraPoint, env, sm, out, out_row_pitch_px, outFmt, nullptr);
}

int vqhip_forward_lighting_mrt(vqhip_ctx* ctx, void* stream, const vqhip_gbuffer* gb,
        const VQ_PerFrameData* perFrame, const VQ_PerViewLightingData* perView,
        const VQ_PointLight* extraPoint, int numExtraPoint,
        const vqhip_envmap* env, const vqhip_shadowmaps* sm,
        void* out, int out_row_pitch_px, vqhip_format outFmt, const vqhip_psmain_targets* targets) {
    vqk::Range range_("RenderSceneColor");
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "forward_lighting: ctx is NULL");
    CTX_GUARD(ctx, "forward_lighting");
    if (!gb || !perFrame || !perView || !out) return fail(ctx, VQHIP_ERR_INVALID_ARG, "forward_lighting: NULL argument");
    if (!gb->gb0 || !gb->gb1 || !gb->gb2 || !gb->gb3) return fail(ctx, VQHIP_ERR_INVALID_ARG, "forward_lighting: NULL G-buffer plane");
    if (gb->width <= 0 || gb->height <= 0 || gb->row_pitch_px < gb->width || out_row_pitch_px < gb->width)
        return fail(ctx, VQHIP_ERR_INVALID_ARG, "forward_lighting: bad dimensions / pitch");
    bool casters = false;
    int rc = validateLighting(ctx, "forward_lighting", perFrame, perView, extraPoint, numExtraPoint, env, sm, outFmt, &casters);
    if (rc) return rc;
    MrtArgs mrt;
    if ((rc = fillMrt(ctx, "forward_lighting", targets, gb->width, &mrt)) != VQHIP_OK) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    int slot;
    rc = acquireSlot(ctx, &slot);
    if (rc) return rc;
    const size_t bytes = fillFrameConstants(ctx, slot, perFrame, perView, extraPoint, numExtraPoint, env, sm);
    rc = commitSlot(ctx, slot, bytes, st);
    if (rc) return rc;
    ShadeArgs a;
    a.mrt = mrt;
    a.gb0 = (const float4*)gb->gb0; a.gb1 = (const float4*)gb->gb1; a.gb2 = (const float4*)gb->gb2; a.gb3 = (const float4*)gb->gb3;
    a.out = out;
    a.fc = (const FrameConstants*)(ctx->devRing + (size_t)slot * kConstSlotBytes);
    a.width = gb->width; a.height = gb->height; a.pitch = gb->row_pitch_px; a.outPitch = out_row_pitch_px;
    a.arithDxc = ctx->arithDxc;
    hipError_t e = launch_forward_lighting(st, a, env != nullptr, casters, outFmt, ctx->opt);
    if (e != hipSuccess) return failHip(ctx, e, "forward_lighting launch");
    return releaseSlot(ctx, slot, st);
}

int vqhip_gaussian_blur_x(vqhip_ctx* ctx, void* stream, const void* in, void* out, const VQ_BlurParams* p, vqhip_format fmt) {
    vqk::Range range_("BlurX");
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "gaussian_blur_x: ctx is NULL");
    CTX_GUARD(ctx, "gaussian_blur_x");
    if (!in || !out || !p || p->iImageSizeX <= 0 || p->iImageSizeY <= 0) return fail(ctx, VQHIP_ERR_INVALID_ARG, "gaussian_blur_x: bad argument");
    if (!isImageFmt(fmt)) return fail(ctx, VQHIP_ERR_UNSUPPORTED, "gaussian_blur_x: fmt must be RGBA32F or RGBA16F");
    if (in == out) return fail(ctx, VQHIP_ERR_INVALID_ARG, "gaussian_blur_x: in-place blur is not supported");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = launch_blur_x((hipStream_t)stream, in, out, p->iImageSizeX, p->iImageSizeY, fmt, ctx->opt);
    return e == hipSuccess ? VQHIP_OK : failHip(ctx, e, "blur_x launch");
}

int vqhip_gaussian_blur_y(vqhip_ctx* ctx, void* stream, const void* in, void* out, const void* halo_top, const void* halo_bottom, int halo_rows,
                          const VQ_BlurParams* p, vqhip_format fmt) {
    vqk::Range range_("BlurY");
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "gaussian_blur_y: ctx is NULL");
    CTX_GUARD(ctx, "gaussian_blur_y");
    if (!in || !out || !p || p->iImageSizeX <= 0 || p->iImageSizeY <= 0) return fail(ctx, VQHIP_ERR_INVALID_ARG, "gaussian_blur_y: bad argument");
    if (!isImageFmt(fmt)) return fail(ctx, VQHIP_ERR_UNSUPPORTED, "gaussian_blur_y: fmt must be RGBA32F or RGBA16F");
    if (in == out) return fail(ctx, VQHIP_ERR_INVALID_ARG, "gaussian_blur_y: in-place blur is not supported");
    if ((halo_top || halo_bottom) && halo_rows < 10) return fail(ctx, VQHIP_ERR_INVALID_ARG, "gaussian_blur_y: halo_rows must be >= 10 (KERNEL_RANGE-1)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = launch_blur_y((hipStream_t)stream, in, out, halo_top, halo_bottom, halo_rows, p->iImageSizeX, p->iImageSizeY, fmt);
    return e == hipSuccess ? VQHIP_OK : failHip(ctx, e, "blur_y launch");
}

int vqhip_gaussian_blur_y_tonemap(vqhip_ctx* ctx, void* stream, const void* in, void* out, const void* halo_top, const void* halo_bottom, int halo_rows,
                                  const VQ_BlurParams* p, const VQ_TonemapperParams* tm, vqhip_format blurFmt, vqhip_format outFmt) {
    vqk::Range range_("BlurY+TonemapperCS");
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "gaussian_blur_y_tonemap: ctx is NULL");
    CTX_GUARD(ctx, "gaussian_blur_y_tonemap");
    if (!in || !out || !p || !tm || p->iImageSizeX <= 0 || p->iImageSizeY <= 0) return fail(ctx, VQHIP_ERR_INVALID_ARG, "gaussian_blur_y_tonemap: bad argument");
    if (!isImageFmt(blurFmt)) return fail(ctx, VQHIP_ERR_UNSUPPORTED, "gaussian_blur_y_tonemap: blurFmt must be RGBA32F or RGBA16F");
    if (!isImageFmt(outFmt) && outFmt != VQHIP_FMT_RGBA8_UNORM) return fail(ctx, VQHIP_ERR_UNSUPPORTED, "gaussian_blur_y_tonemap: outFmt must be RGBA32F, RGBA16F or RGBA8_UNORM");
    if (in == out) return fail(ctx, VQHIP_ERR_INVALID_ARG, "gaussian_blur_y_tonemap: in-place is not supported");
    if ((halo_top || halo_bottom) && halo_rows < 10) return fail(ctx, VQHIP_ERR_INVALID_ARG, "gaussian_blur_y_tonemap: halo_rows must be >= 10");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int slot = -1;
    if (blur_y_tonemap_uses_lut(*tm, blurFmt, outFmt, (size_t)p->iImageSizeX * p->iImageSizeY)) { const int rc = acquireTonemapLut(ctx, (hipStream_t)stream, *tm, outFmt, &slot); if (rc) return rc; }
    hipError_t e = launch_blur_y_tonemap((hipStream_t)stream, in, out, halo_top, halo_bottom, halo_rows, p->iImageSizeX, p->iImageSizeY, *tm, blurFmt, outFmt,
                                         slot >= 0 ? ctx->lut[slot].table : nullptr, ctx->opt);
    if (e != hipSuccess) return failHip(ctx, e, "blur_y_tonemap launch");
    return slot >= 0 ? releaseTonemapLut(ctx, (hipStream_t)stream, slot) : VQHIP_OK;
}

int vqhip_gaussian_blur(vqhip_ctx* ctx, void* stream, const void* in, void* tmp, void* out, const VQ_BlurParams* p, vqhip_format fmt) {
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "gaussian_blur: ctx is NULL");
    CTX_GUARD(ctx, "gaussian_blur");
    if (!tmp) return fail(ctx, VQHIP_ERR_INVALID_ARG, "gaussian_blur: tmp is NULL");
    int rc = vqhip_gaussian_blur_x(ctx, stream, in, tmp, p, fmt);
    if (rc) return rc;
    return vqhip_gaussian_blur_y(ctx, stream, tmp, out, nullptr, nullptr, 0, p, fmt);
}

int vqhip_tonemap(vqhip_ctx* ctx, void* stream, const void* in, void* out, int width, int height,
                  const VQ_TonemapperParams* p, vqhip_format inFmt, vqhip_format outFmt) {
    vqk::Range range_("TonemapperCS");
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "tonemap: ctx is NULL");
    CTX_GUARD(ctx, "tonemap");
    if (!in || !out || !p || width <= 0 || height <= 0) return fail(ctx, VQHIP_ERR_INVALID_ARG, "tonemap: bad argument");
    if (!isImageFmt(inFmt)) return fail(ctx, VQHIP_ERR_UNSUPPORTED, "tonemap: inFmt must be RGBA32F or RGBA16F");
    if (!isImageFmt(outFmt) && outFmt != VQHIP_FMT_RGBA8_UNORM) return fail(ctx, VQHIP_ERR_UNSUPPORTED, "tonemap: outFmt must be RGBA32F, RGBA16F or RGBA8_UNORM");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int slot = -1;
    if (tonemap_uses_lut(*p, inFmt, outFmt, (size_t)width * height)) { const int rc = acquireTonemapLut(ctx, (hipStream_t)stream, *p, outFmt, &slot); if (rc) return rc; }
    hipError_t e = launch_tonemap((hipStream_t)stream, in, out, width, height, *p, inFmt, outFmt, slot >= 0 ? ctx->lut[slot].table : nullptr, ctx->opt);
    if (e != hipSuccess) return failHip(ctx, e, "tonemap launch");
    return slot >= 0 ? releaseTonemapLut(ctx, (hipStream_t)stream, slot) : VQHIP_OK;
}

// sceneColor -> out for one row tile; halo_top / halo_bottom are SCENE-COLOUR rows (NULL: image border)
static int postProcessTile(vqhip_ctx* ctx, void* stream, const void* sceneColor, void* out, const void* halo_top, const void* halo_bottom, int halo_rows,
                           int width, int height, const VQ_TonemapperParams* tm, vqhip_format inFmt, vqhip_format outFmt) {
    const hipStream_t st = (hipStream_t)stream;
    if (post_chain_applies(*tm, inFmt, outFmt, width, height, ctx->opt)) {                      // X, Y and the tonemapper in one kernel: no intermediate at all
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        int slot = -1;
        const int rc = acquireTonemapLut(ctx, st, *tm, outFmt, &slot);
        if (rc) return rc;
        hipError_t e = launch_post_chain(st, sceneColor, out, halo_top, halo_bottom, halo_rows, width, height, ctx->lut[slot].table, ctx->nCUs, ctx->opt);
        if (e != hipSuccess) return failHip(ctx, e, "post chain launch");
        return releaseTonemapLut(ctx, st, slot);
    }
    // blur X into a BlurIntermediate held in the context's scratch buffer (the tile, then the halo rows behind it), then blur Y + tonemap in one kernel
    // (or on the LDS-tile kernel: HDR / RGBA32F targets)
    const size_t bpp = inFmt == VQHIP_FMT_RGBA32F ? 16 : 8;
    const size_t tileBytes = (size_t)width * height * bpp, haloBytes = (size_t)width * halo_rows * bpp;
    const int rc0 = ensureScratch(ctx, tileBytes + 2 * haloBytes);
    if (rc0) return rc0;
    char* xt = (char*)ctx->scratch; char* xTop = xt + tileBytes; char* xBot = xTop + haloBytes;
    const VQ_BlurParams bp = { width, height };
    int rc = vqhip_gaussian_blur_x(ctx, stream, sceneColor, xt, &bp, inFmt);
    if (rc) return rc;
    const VQ_BlurParams hp = { width, halo_rows };
    if (halo_top && (rc = vqhip_gaussian_blur_x(ctx, stream, halo_top, xTop, &hp, inFmt))) return rc;
    if (halo_bottom && (rc = vqhip_gaussian_blur_x(ctx, stream, halo_bottom, xBot, &hp, inFmt))) return rc;
    return vqhip_gaussian_blur_y_tonemap(ctx, stream, xt, out, halo_top ? xTop : nullptr, halo_bottom ? xBot : nullptr, halo_rows, &bp, tm, inFmt, outFmt);
}

static int postProcessCheck(vqhip_ctx* ctx, const char* who, const void* sceneColor, const void* out, int width, int height, const VQ_TonemapperParams* tm,
                            vqhip_format inFmt, vqhip_format outFmt) {
    const std::string w = who;
    if (!sceneColor || !out || !tm || width <= 0 || height <= 0) return fail(ctx, VQHIP_ERR_INVALID_ARG, w + ": bad argument");
    if (!isImageFmt(inFmt)) return fail(ctx, VQHIP_ERR_UNSUPPORTED, w + ": inFmt must be RGBA32F or RGBA16F");
    if (!isImageFmt(outFmt) && outFmt != VQHIP_FMT_RGBA8_UNORM) return fail(ctx, VQHIP_ERR_UNSUPPORTED, w + ": outFmt must be RGBA32F, RGBA16F or RGBA8_UNORM");
    if (sceneColor == out) return fail(ctx, VQHIP_ERR_INVALID_ARG, w + ": in-place is not supported");
    return VQHIP_OK;
}

int vqhip_post_process(vqhip_ctx* ctx, void* stream, const void* sceneColor, void* out, int width, int height,
                       const VQ_TonemapperParams* tm, int enableGaussianBlur, vqhip_format inFmt, vqhip_format outFmt) {
    vqk::Range range_("RenderPostProcess");
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "post_process: ctx is NULL");
    CTX_GUARD(ctx, "post_process");
    const int rc = postProcessCheck(ctx, "post_process", sceneColor, out, width, height, tm, inFmt, outFmt);
    if (rc) return rc;
    if (!enableGaussianBlur) return vqhip_tonemap(ctx, stream, sceneColor, out, width, height, tm, inFmt, outFmt);
    return postProcessTile(ctx, stream, sceneColor, out, nullptr, nullptr, 0, width, height, tm, inFmt, outFmt);
}

int vqhip_post_process_tile(vqhip_ctx* ctx, void* stream, const void* sceneColor, void* out, const void* halo_top, const void* halo_bottom, int halo_rows,
                            int width, int height, const VQ_TonemapperParams* tm, vqhip_format inFmt, vqhip_format outFmt) {
    vqk::Range range_("RenderPostProcess(tile)");
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "post_process_tile: ctx is NULL");
    CTX_GUARD(ctx, "post_process_tile");
    const int rc = postProcessCheck(ctx, "post_process_tile", sceneColor, out, width, height, tm, inFmt, outFmt);
    if (rc) return rc;
    if ((halo_top || halo_bottom) && halo_rows < 10) return fail(ctx, VQHIP_ERR_INVALID_ARG, "post_process_tile: halo_rows must be >= 10");
    if (!halo_top && !halo_bottom) halo_rows = 0;
    return postProcessTile(ctx, stream, sceneColor, out, halo_top, halo_bottom, halo_rows, width, height, tm, inFmt, outFmt);
}

int vqhip_set_option(vqhip_ctx* ctx, const char* key, const char* value) {
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "set_option: ctx is NULL");
    CTX_GUARD(ctx, "set_option");
    if (!key) return fail(ctx, VQHIP_ERR_INVALID_ARG, "set_option: key is NULL");
    const std::string k = key, v = (value && std::strcmp(value, "default")) ? value : "";
    vqk::Options& o = ctx->opt;
    auto num = [&](int* dst, std::initializer_list<int> allowed) -> int {       // "" -> 0 (default); otherwise a non-negative integer (from `allowed` when given)
        if (v.empty()) { *dst = 0; return VQHIP_OK; }
        char* end = nullptr;
        const long n = std::strtol(v.c_str(), &end, 10);
        bool ok = end && !*end && n >= 0 && n <= (1 << 24);
        if (ok && allowed.size()) { ok = false; for (int a : allowed) ok |= a == (int)n; }
        if (!ok) return fail(ctx, VQHIP_ERR_INVALID_ARG, "set_option: bad value '" + v + "' for " + k);
        *dst = (int)n; return VQHIP_OK;
    };
    auto pick = [&](int* dst, std::initializer_list<const char*> names) -> int {  // names[i] selects value i + 1
        if (v.empty()) { *dst = 0; return VQHIP_OK; }
        int i = 1;
        for (const char* n : names) { if (v == n) { *dst = i; return VQHIP_OK; } ++i; }
        return fail(ctx, VQHIP_ERR_INVALID_ARG, "set_option: bad value '" + v + "' for " + k);
    };
    if (k == "shade_wg") return num(&o.shadeWg, { 0, 64, 128, 256 });
    if (k == "psmain_waves") return num(&o.psmainWaves, { 0, 4, 5, 6 });
    if (k == "blur_y_wgs") return num(&o.blurYWgs, {});
    if (k == "post_form") return pick(&o.postForm, { "two", "chain" });
    if (k == "post_strips") return num(&o.postStrips, {});
    if (k == "lut_form") return pick(&o.lutForm, { "general" });
    if (k == "specular_form") return pick(&o.specularForm, { "general" });
    if (k == "diffuse_form") { if (v == "records") { o.diffuseForm = 0; return VQHIP_OK; } return pick(&o.diffuseForm, { "texels", "general" }); }
    if (k == "diffuse_seq_form") { if (v == "ordered") { o.diffuseSeqForm = 0; return VQHIP_OK; } return pick(&o.diffuseSeqForm, { "lane" }); }
    return fail(ctx, VQHIP_ERR_INVALID_ARG, "set_option: unknown key '" + k + "'");
}
int vqhip_set_arithmetic(vqhip_ctx* ctx, vqhip_arithmetic mode) {
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "set_arithmetic: ctx is NULL");
    CTX_GUARD(ctx, "set_arithmetic");
    if (mode != VQHIP_ARITH_LITERAL && mode != VQHIP_ARITH_DXC) return fail(ctx, VQHIP_ERR_INVALID_ARG, "set_arithmetic: unknown mode");
    ctx->arithDxc = mode == VQHIP_ARITH_DXC ? 1 : 0;
    return VQHIP_OK;
}
int vqhip_set_fresnel_pow(vqhip_ctx* ctx, vqhip_fresnel_pow mode) {
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "set_fresnel_pow: ctx is NULL");
    CTX_GUARD(ctx, "set_fresnel_pow");
    if (mode != VQHIP_FRESNEL_POW_PRODUCT && mode != VQHIP_FRESNEL_POW_EXP2_LOG2) return fail(ctx, VQHIP_ERR_INVALID_ARG, "set_fresnel_pow: unknown mode");
    ctx->pow5ExpLog = mode == VQHIP_FRESNEL_POW_EXP2_LOG2 ? 1 : 0;
    return VQHIP_OK;
}

int vqhip_brdf_lut(vqhip_ctx* ctx, void* stream, void* outRG, int size, int samples, vqhip_format fmt) {
    vqk::Range range_("CreateBRDFIntegralLUT");
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "brdf_lut: ctx is NULL");
    CTX_GUARD(ctx, "brdf_lut");
    if (!outRG || size <= 0 || samples <= 0) return fail(ctx, VQHIP_ERR_INVALID_ARG, "brdf_lut: bad argument");
    if (fmt != VQHIP_FMT_RG16F && fmt != VQHIP_FMT_RG32F) return fail(ctx, VQHIP_ERR_UNSUPPORTED, "brdf_lut: fmt must be RG16F or RG32F");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = launch_brdf_lut((hipStream_t)stream, outRG, size, samples, fmt, ctx->pow5ExpLog, ctx->opt);
    return e == hipSuccess ? VQHIP_OK : failHip(ctx, e, "brdf_lut launch");
}

int vqhip_mip_level_count(int w, int h) { int m = w > h ? w : h; if (m < 1) return 0; int n = 1; while (m > 1) { m >>= 1; ++n; } return n; }
size_t vqhip_mip_level_offset_bytes(int w0, int h0, int level) {
    size_t off = 0;
    for (int l = 0; l < level; ++l) off += (size_t)mipDim(w0, l) * mipDim(h0, l) * 16;
    return off;
}
size_t vqhip_mip_chain_bytes(int w0, int h0, int nMips) { return vqhip_mip_level_offset_bytes(w0, h0, nMips); }

int vqhip_mip_chain_min_rgba32f(vqhip_ctx* ctx, void* stream, void* mips, int w0, int h0, int nMips) {
    vqk::Range range_("GenerateMips");
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "mip_chain: ctx is NULL");
    CTX_GUARD(ctx, "mip_chain");
    if (!mips || w0 <= 0 || h0 <= 0 || nMips <= 0 || nMips > vqhip_mip_level_count(w0, h0)) return fail(ctx, VQHIP_ERR_INVALID_ARG, "mip_chain: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (int l = 1; l < nMips; ++l) {
        const float4* src = (const float4*)((const char*)mips + vqhip_mip_level_offset_bytes(w0, h0, l - 1));
        float4* dst = (float4*)((char*)mips + vqhip_mip_level_offset_bytes(w0, h0, l));
        hipError_t e = launch_mip_min((hipStream_t)stream, src, dst, mipDim(w0, l - 1), mipDim(h0, l - 1), mipDim(w0, l), mipDim(h0, l));
        if (e != hipSuccess) return failHip(ctx, e, "mip_min launch");
    }
    return VQHIP_OK;
}

// ---- G-buffer producer --------------------------------------------------------------
size_t vqhip_mip_chain_bytes_rgba8(int w0, int h0, int nMips) { return vqhip_mip_level_offset_bytes(w0, h0, nMips) / 4; }

int vqhip_mip_chain_box_rgba8(vqhip_ctx* ctx, void* stream, void* mips, int w0, int h0, int nMips) {
    vqk::Range range_("GenerateMips");
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "mip_chain_box_rgba8: ctx is NULL");
    CTX_GUARD(ctx, "mip_chain_box_rgba8");
    if (!mips || w0 <= 0 || h0 <= 0 || nMips <= 0 || nMips > vqhip_mip_level_count(w0, h0)) return fail(ctx, VQHIP_ERR_INVALID_ARG, "mip_chain_box_rgba8: bad argument");
    if ((w0 & (w0 - 1)) || (h0 & (h0 - 1))) return fail(ctx, VQHIP_ERR_UNSUPPORTED, "mip_chain_box_rgba8: w0 and h0 must be powers of two");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (int l = 1; l < nMips; ++l) {
        const void* src = (const char*)mips + vqhip_mip_level_offset_bytes(w0, h0, l - 1) / 4;
        void* dst = (char*)mips + vqhip_mip_level_offset_bytes(w0, h0, l) / 4;
        hipError_t e = launch_mip_box_rgba8((hipStream_t)stream, src, dst, mipDim(w0, l - 1), mipDim(h0, l - 1), mipDim(w0, l), mipDim(h0, l));
        if (e != hipSuccess) return failHip(ctx, e, "mip_box_rgba8 launch");
    }
    return VQHIP_OK;
}

int vqhip_max_materials(void) { return kMaxMaterials; }

static bool badTexture(const vqhip_texture2d& t) {
    return t.texels && (t.width <= 0 || t.height <= 0 || t.mips <= 0 || t.mips > vqhip_mip_level_count(t.width, t.height));
}

static int validateProducer(vqhip_ctx* ctx, const char* who, const vqhip_interpolants* in, const vqhip_material* materials, int numMaterials, const vqhip_ssao* ssao) {
    const std::string w(who);
    if (!in || !in->ip0 || !in->ip1 || !in->ip2) return fail(ctx, VQHIP_ERR_INVALID_ARG, w + ": NULL plane");
    if (in->width <= 0 || in->height <= 0 || in->row_pitch_px < in->width) return fail(ctx, VQHIP_ERR_INVALID_ARG, w + ": bad dimensions / pitch");
    if ((uint64_t)in->row_pitch_px * in->height * 16u >= (1ull << 32) || in->row_pitch_px >= (1 << 24) || in->height >= (1 << 24))
        return fail(ctx, VQHIP_ERR_UNSUPPORTED, w + ": a plane must be smaller than 4 GiB (32-bit offsets in the kernel)");
    if (numMaterials < 0 || numMaterials > kMaxMaterials || (numMaterials > 0 && !materials))
        return fail(ctx, VQHIP_ERR_INVALID_ARG, w + ": bad materials / numMaterials (see vqhip_max_materials)");
    for (int i = 0; i < numMaterials; ++i) {
        const vqhip_material& m = materials[i];
        if (badTexture(m.texDiffuse) || badTexture(m.texNormals) || badTexture(m.texEmissive) || badTexture(m.texMetalness) ||
            badTexture(m.texRoughness) || badTexture(m.texOcclRoughMetal) || badTexture(m.texLocalAO))
            return fail(ctx, VQHIP_ERR_INVALID_ARG, w + ": material " + std::to_string(i) + " has a texture with bad dimensions / mip count");
    }
    if (ssao && ssao->texels && (ssao->width <= 0 || ssao->height <= 0)) return fail(ctx, VQHIP_ERR_INVALID_ARG, w + ": bad ssao dimensions");
    return VQHIP_OK;
}
static size_t fillGbufConstants(vqhip_ctx* ctx, int slot, const vqhip_material* materials, int numMaterials, float ambient, const vqhip_ssao* ssao) {
    GbufConstants* gc = (GbufConstants*)(ctx->hostRing + (size_t)slot * kConstSlotBytes);
    std::memset(gc, 0, offsetof(GbufConstants, mats));
    gc->ambient = ambient;
    gc->numMaterials = numMaterials;
    gc->arithDxc = ctx->arithDxc;
    if (ssao && ssao->texels) gc->ssao = *ssao;
    if (numMaterials > 0) std::memcpy(gc->mats, materials, (size_t)numMaterials * sizeof(vqhip_material));
    return offsetof(GbufConstants, mats) + (size_t)numMaterials * sizeof(vqhip_material);
}

int vqhip_gbuffer_from_materials(vqhip_ctx* ctx, void* stream, const vqhip_interpolants* in, const vqhip_material* materials, int numMaterials,
                                 float fAmbientLightingFactor, const vqhip_ssao* ssao, const vqhip_gbuffer* out) {
    vqk::Range range_("Geometry");                       // :1723 (surface assembly half of the lit draws)
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "gbuffer_from_materials: ctx is NULL");
    CTX_GUARD(ctx, "gbuffer_from_materials");
    if (!out || !out->gb0 || !out->gb1 || !out->gb2 || !out->gb3) return fail(ctx, VQHIP_ERR_INVALID_ARG, "gbuffer_from_materials: NULL plane");
    int rc = validateProducer(ctx, "gbuffer_from_materials", in, materials, numMaterials, ssao);
    if (rc) return rc;
    if (out->width != in->width || out->height != in->height || out->row_pitch_px < in->width)
        return fail(ctx, VQHIP_ERR_INVALID_ARG, "gbuffer_from_materials: bad dimensions / pitch");
    if ((uint64_t)out->row_pitch_px * out->height * 16u >= (1ull << 32) || out->row_pitch_px >= (1 << 24))
        return fail(ctx, VQHIP_ERR_UNSUPPORTED, "gbuffer_from_materials: a plane must be smaller than 4 GiB (32-bit offsets in the kernel)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    int slot;
    rc = acquireSlot(ctx, &slot);
    if (rc) return rc;
    const size_t bytes = fillGbufConstants(ctx, slot, materials, numMaterials, fAmbientLightingFactor, ssao);
    rc = commitSlot(ctx, slot, bytes, st);
    if (rc) return rc;
    GbufArgs a;
    a.ip0 = (const float4*)in->ip0; a.ip1 = (const float4*)in->ip1; a.ip2 = (const float4*)in->ip2;
    a.gb0 = (float4*)out->gb0; a.gb1 = (float4*)out->gb1; a.gb2 = (float4*)out->gb2; a.gb3 = (float4*)out->gb3;
    a.gc = (const GbufConstants*)(ctx->devRing + (size_t)slot * kConstSlotBytes);
    a.width = in->width; a.height = in->height; a.pitch = in->row_pitch_px; a.outPitch = out->row_pitch_px;
    hipError_t e = launch_gbuffer_from_materials(st, a);
    if (e != hipSuccess) return failHip(ctx, e, "gbuffer_from_materials launch");
    return releaseSlot(ctx, slot, st);
}

// The colour target of the Z pre-pass (DepthPrePass.hlsl:PSMain :153-171; VQRenderer::RenderDepthPrePass, SceneRendering.cpp:1264-1360): Tex_SceneNormals.
int vqhip_scene_normals_from_materials(vqhip_ctx* ctx, void* stream, const vqhip_interpolants* in, const vqhip_material* materials, int numMaterials,
                                       void* out, vqhip_format outFmt, int out_row_pitch_px) {
    vqk::Range range_("RenderDepthPrePass");             // SceneRendering.cpp:1273
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "scene_normals_from_materials: ctx is NULL");
    CTX_GUARD(ctx, "scene_normals_from_materials");
    if (!out) return fail(ctx, VQHIP_ERR_INVALID_ARG, "scene_normals_from_materials: out is NULL");
    if (outFmt != VQHIP_FMT_R10G10B10A2_UNORM && outFmt != VQHIP_FMT_RGBA32F)
        return fail(ctx, VQHIP_ERR_UNSUPPORTED, "scene_normals_from_materials: outFmt must be R10G10B10A2_UNORM or RGBA32F");
    int rc = validateProducer(ctx, "scene_normals_from_materials", in, materials, numMaterials, nullptr);
    if (rc) return rc;
    const int pitch = out_row_pitch_px ? out_row_pitch_px : in->width;
    if (pitch < in->width) return fail(ctx, VQHIP_ERR_INVALID_ARG, "scene_normals_from_materials: bad output pitch");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    int slot;
    rc = acquireSlot(ctx, &slot);
    if (rc) return rc;
    const size_t bytes = fillGbufConstants(ctx, slot, materials, numMaterials, 0.0f, nullptr);
    rc = commitSlot(ctx, slot, bytes, st);
    if (rc) return rc;
    GbufArgs a;
    a.ip0 = (const float4*)in->ip0; a.ip1 = (const float4*)in->ip1; a.ip2 = (const float4*)in->ip2;
    a.gb0 = a.gb1 = a.gb2 = a.gb3 = nullptr;
    a.gc = (const GbufConstants*)(ctx->devRing + (size_t)slot * kConstSlotBytes);
    a.width = in->width; a.height = in->height; a.pitch = in->row_pitch_px; a.outPitch = pitch;
    hipError_t e = launch_scene_normals_from_materials(st, a, out, outFmt);
    if (e != hipSuccess) return failHip(ctx, e, "scene_normals_from_materials launch");
    return releaseSlot(ctx, slot, st);
}

// PSMain as the engine runs it (ForwardLighting.hlsl:226-380, the lit draws of RenderSceneColor, SceneRendering.cpp:1619-1760): interpolants +
// materials in, scene colour out, one kernel — the G-buffer record stays in registers. Two ring slots: the producer's constants and the lighting's.
int vqhip_forward_lighting_from_materials(vqhip_ctx* ctx, void* stream, const vqhip_interpolants* in, const vqhip_material* materials, int numMaterials,
        const vqhip_ssao* ssao, const VQ_PerFrameData* perFrame, const VQ_PerViewLightingData* perView,
        const VQ_PointLight* extraPoint, int numExtraPoint, const vqhip_envmap* env, const vqhip_shadowmaps* sm,
        void* out, int out_row_pitch_px, vqhip_format outFmt) {
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "forward_lighting_from_materials: ctx is NULL");
    CTX_GUARD(ctx, "forward_lighting_from_materials");
    return vqhip_forward_lighting_from_materials_mrt(ctx, stream, in, materials, numMaterials, ssao, perFrame, perView, extraPoint, numExtraPoint, env, sm,
                                                     out, out_row_pitch_px, outFmt, nullptr);
}

int vqhip_forward_lighting_from_materials_mrt(vqhip_ctx* ctx, void* stream, const vqhip_interpolants* in, const vqhip_material* materials, int numMaterials,
        const vqhip_ssao* ssao, const VQ_PerFrameData* perFrame, const VQ_PerViewLightingData* perView,
        const VQ_PointLight* extraPoint, int numExtraPoint, const vqhip_envmap* env, const vqhip_shadowmaps* sm,
        void* out, int out_row_pitch_px, vqhip_format outFmt, const vqhip_psmain_targets* targets) {
    vqk::Range range_("RenderSceneColor");
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "forward_lighting_from_materials: ctx is NULL");
    CTX_GUARD(ctx, "forward_lighting_from_materials");
    if (!out) return fail(ctx, VQHIP_ERR_INVALID_ARG, "forward_lighting_from_materials: out is NULL");
    int rc = validateProducer(ctx, "forward_lighting_from_materials", in, materials, numMaterials, ssao);
    if (rc) return rc;
    if (out_row_pitch_px < in->width) return fail(ctx, VQHIP_ERR_INVALID_ARG, "forward_lighting_from_materials: bad output pitch");
    bool casters = false;
    rc = validateLighting(ctx, "forward_lighting_from_materials", perFrame, perView, extraPoint, numExtraPoint, env, sm, outFmt, &casters);
    if (rc) return rc;
    MrtArgs mrt;
    if ((rc = fillMrt(ctx, "forward_lighting_from_materials", targets, in->width, &mrt)) != VQHIP_OK) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    int slotG, slotF;
    if ((rc = acquireSlot(ctx, &slotG)) != VQHIP_OK || (rc = acquireSlot(ctx, &slotF)) != VQHIP_OK) return rc;
    const size_t bytesG = fillGbufConstants(ctx, slotG, materials, numMaterials, perFrame->fAmbientLightingFactor, ssao);     // cbPerFrame.fAmbientLightingFactor, :247
    const size_t bytesF = fillFrameConstants(ctx, slotF, perFrame, perView, extraPoint, numExtraPoint, env, sm);
    if ((rc = commitSlot(ctx, slotG, bytesG, st)) != VQHIP_OK || (rc = commitSlot(ctx, slotF, bytesF, st)) != VQHIP_OK) return rc;
    GbufArgs a;
    a.ip0 = (const float4*)in->ip0; a.ip1 = (const float4*)in->ip1; a.ip2 = (const float4*)in->ip2;
    a.gb0 = a.gb1 = a.gb2 = a.gb3 = nullptr;
    a.gc = (const GbufConstants*)(ctx->devRing + (size_t)slotG * kConstSlotBytes);
    a.width = in->width; a.height = in->height; a.pitch = in->row_pitch_px; a.outPitch = 0;
    hipError_t e = launch_forward_from_materials(st, a, (const FrameConstants*)(ctx->devRing + (size_t)slotF * kConstSlotBytes), env != nullptr, casters,
                                                 out, out_row_pitch_px, outFmt, ctx->arithDxc, ctx->opt, mrt);
    if (e != hipSuccess) return failHip(ctx, e, "forward_lighting_from_materials launch");
    if ((rc = releaseSlot(ctx, slotG, st)) != VQHIP_OK) return rc;
    return releaseSlot(ctx, slotF, st);
}

// ---- skydome ------------------------------------------------------------------------
int vqhip_skydome(vqhip_ctx* ctx, void* stream, const void* equirect_level0, int w0, int h0, const VQ_SkydomeParams* params,
                  const vqhip_interpolants* coverage, void* color, int width, int height, int row_pitch_px, vqhip_format fmt) {
    vqk::Range range_("EnvironmentMap");                 // SceneRendering.cpp:1824
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "skydome: ctx is NULL");
    CTX_GUARD(ctx, "skydome");
    if (!equirect_level0 || !params || !color || w0 <= 0 || h0 <= 0 || width <= 0 || height <= 0 || row_pitch_px < width)
        return fail(ctx, VQHIP_ERR_INVALID_ARG, "skydome: bad argument");
    if (!isImageFmt(fmt)) return fail(ctx, VQHIP_ERR_UNSUPPORTED, "skydome: fmt must be RGBA32F or RGBA16F");
    if (coverage && (!coverage->ip2 || coverage->width != width || coverage->height != height || coverage->row_pitch_px < width))
        return fail(ctx, VQHIP_ERR_INVALID_ARG, "skydome: coverage planes do not match the colour target");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = launch_skydome((hipStream_t)stream, (const float4*)equirect_level0, w0, h0, *params,
                                  coverage ? (const float4*)coverage->ip2 : nullptr, coverage ? coverage->row_pitch_px : 0,
                                  color, width, height, row_pitch_px, fmt);
    return e == hipSuccess ? VQHIP_OK : failHip(ctx, e, "skydome launch");
}

int vqhip_unlit_composite(vqhip_ctx* ctx, void* stream, const vqhip_interpolants* coverage, const VQ_float4* colors, int numColors,
                          void* color, int width, int height, int row_pitch_px, vqhip_format fmt) {
    vqk::Range range_("Lights");                         // :1790
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "unlit_composite: ctx is NULL");
    CTX_GUARD(ctx, "unlit_composite");
    if (!coverage || !coverage->ip2 || !color || width <= 0 || height <= 0 || row_pitch_px < width || numColors < 0 || (numColors > 0 && !colors))
        return fail(ctx, VQHIP_ERR_INVALID_ARG, "unlit_composite: bad argument");
    if (numColors > VQHIP_MAX_UNLIT_COLORS) return fail(ctx, VQHIP_ERR_INVALID_ARG, "unlit_composite: more than VQHIP_MAX_UNLIT_COLORS gizmos");
    if (coverage->width != width || coverage->height != height || coverage->row_pitch_px < width)
        return fail(ctx, VQHIP_ERR_INVALID_ARG, "unlit_composite: coverage planes do not match the colour target");
    if (!isImageFmt(fmt)) return fail(ctx, VQHIP_ERR_UNSUPPORTED, "unlit_composite: fmt must be RGBA32F or RGBA16F");
    if (numColors == 0) return VQHIP_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    UnlitColors cols;
    for (int i = 0; i < numColors; ++i) cols.c[i] = make_float4(colors[i].x, colors[i].y, colors[i].z, colors[i].w);
    hipError_t e = launch_unlit_composite((hipStream_t)stream, (const float4*)coverage->ip2, coverage->row_pitch_px, cols, numColors,
                                          color, width, height, row_pitch_px, fmt);
    return e == hipSuccess ? VQHIP_OK : failHip(ctx, e, "unlit_composite launch");
}

// ---- HDRI ingest --------------------------------------------------------------------
int vqhip_hdr_parse_header(const void* file, size_t bytes, int* width, int* height, size_t* data_offset) {
    if (!file || !width || !height) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "hdr_parse_header: NULL argument");
    const char* err = nullptr; size_t off = 0;
    if (hdr_parse_header((const uint8_t*)file, bytes, width, height, &off, &err)) return fail(nullptr, VQHIP_ERR_INVALID_ARG, err);
    if (data_offset) *data_offset = off;
    return VQHIP_OK;
}

int vqhip_hdr_decode_rgba32f(vqhip_ctx* ctx, void* stream, const void* file, size_t bytes, void* out_rgba32f, int width, int height) {
    vqk::Range range_("LoadHDRI");
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "hdr_decode: ctx is NULL");
    CTX_GUARD(ctx, "hdr_decode");
    if (!file || !out_rgba32f) return fail(ctx, VQHIP_ERR_INVALID_ARG, "hdr_decode: NULL argument");
    const char* err = nullptr; int w = 0, h = 0; size_t off = 0;
    if (hdr_parse_header((const uint8_t*)file, bytes, &w, &h, &off, &err)) return fail(ctx, VQHIP_ERR_INVALID_ARG, err);
    if (w != width || h != height) return fail(ctx, VQHIP_ERR_INVALID_ARG, "hdr_decode: width/height do not match the file header");
    const size_t px = (size_t)w * h;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    // run-length coded files: the host walks the run headers (offsets of every byte plane, validation), the GPU expands the runs and converts (hdri.hip:k_hdr_expand)
    {
        std::vector<uint32_t> planeOff(4 * (size_t)h + 1);
        const int wr = hdr_walk_runs((const uint8_t*)file, bytes, off, w, h, planeOff.data(), &err);
        if (wr < 0) return fail(ctx, VQHIP_ERR_INVALID_ARG, err);
        int encCap = 0, pitch = 0, ldsBytes = 0;
        if (wr == 0 && hdr_expand_fits(planeOff.data(), w, h, ctx->ldsPerBlock, &encCap, &pitch, &ldsBytes)) {
            const size_t used = planeOff[4 * (size_t)h], fileDev = (used + 4 + 255) & ~(size_t)255, tabBytes = planeOff.size() * 4;
            const int rc = ensureScratch(ctx, fileDev + tabBytes);
            if (rc) return rc;
            // the offset table is a pageable host vector: whatever happens below, the stream is drained before this scope (and the vector) ends
            hipError_t e = hipMemcpyAsync(ctx->scratch, file, used, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipMemcpyAsync((char*)ctx->scratch + fileDev, planeOff.data(), tabBytes, hipMemcpyHostToDevice, st);
            const bool copied = e == hipSuccess;
            if (copied) e = launch_hdr_expand(st, ctx->scratch, (char*)ctx->scratch + fileDev, out_rgba32f, w, h, encCap, pitch, ldsBytes);
            const hipError_t es = hipStreamSynchronize(st);  // load-time call: the offset table and the scratch buffer are free again on return
            if (!copied) return failHip(ctx, e, "hdr_decode upload");
            if (e == hipSuccess && es == hipSuccess) return VQHIP_OK;
            if (es != hipSuccess) return failHip(ctx, es, "hdr_expand");
            (void)hipGetLastError();                         // the launch was refused (e.g. the LDS opt-in on a part with less LDS): the host expansion below still decodes the file
        }
    }
    // flat files, scanlines too wide for the LDS: expansion on the host, conversion on the GPU
    std::vector<uint8_t> rgbe(px * 4);
    if (hdr_expand_rgbe((const uint8_t*)file, bytes, off, w, h, rgbe.data(), &err)) return fail(ctx, VQHIP_ERR_INVALID_ARG, err);
    int rc = ensureScratch(ctx, px * 4);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch, rgbe.data(), px * 4, hipMemcpyHostToDevice, st));
    hipError_t e = launch_rgbe_to_rgba32f(st, ctx->scratch, out_rgba32f, px);
    if (e != hipSuccess) return failHip(ctx, e, "rgbe_to_rgba32f launch");
    HIP_TRY(ctx, hipStreamSynchronize(st));      // load-time call: the staging vector and the scratch buffer are free again on return
    return VQHIP_OK;
}

// ---- FSR 1.0 ------------------------------------------------------------------------
static uint32_t fbits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

// FsrEasuCon, ffx_fsr1.h:156-203, as called on the CPU by FFSR1_EASU::UpdateEASUConstantBlock (PostProcess.cpp:47-79)
int vqhip_hdr_downsize_rgba32f(vqhip_ctx* ctx, void* stream, const void* in, int width, int height, void* out, int out_width, int out_height) {
    vqk::Range range_("DownsizeHDRI");
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "hdr_downsize: ctx is NULL");
    CTX_GUARD(ctx, "hdr_downsize");
    if (!in || !out || width <= 0 || height <= 0 || out_width <= 0 || out_height <= 0) return fail(ctx, VQHIP_ERR_INVALID_ARG, "hdr_downsize: bad argument");
    if (in == out) return fail(ctx, VQHIP_ERR_INVALID_ARG, "hdr_downsize: in-place is not supported");
    const int k = width / out_width;
    if (k < 1 || out_width * k != width || out_height * k != height)
        return fail(ctx, VQHIP_ERR_UNSUPPORTED, "hdr_downsize: only integer ratios, the same in x and y, are implemented (the engine's 8k/4k/2k/1k table); "
                                                "general resampling (stb_image_resize, absent from the reference tree) is not");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (k == 1) { HIP_TRY(ctx, hipMemcpyAsync(out, in, (size_t)width * height * 16, hipMemcpyDeviceToDevice, (hipStream_t)stream)); return VQHIP_OK; }
    hipError_t e = launch_downsize_box((hipStream_t)stream, in, out, width, height, k);
    return e == hipSuccess ? VQHIP_OK : failHip(ctx, e, "downsize launch");
}

void vqhip_fsr_easu_con(uint32_t con[16], float inVpW, float inVpH, float inSzW, float inSzH, float outW, float outH) {
    const float rOutW = 1.0f / outW, rOutH = 1.0f / outH, rInW = 1.0f / inSzW, rInH = 1.0f / inSzH;
    con[0] = fbits(inVpW * rOutW);                 con[1] = fbits(inVpH * rOutH);
    con[2] = fbits(0.5f * inVpW * rOutW - 0.5f);   con[3] = fbits(0.5f * inVpH * rOutH - 0.5f);
    con[4] = fbits(rInW);                          con[5] = fbits(rInH);
    con[6] = fbits(1.0f * rInW);                   con[7] = fbits(-1.0f * rInH);
    con[8] = fbits(-1.0f * rInW);                  con[9] = fbits(2.0f * rInH);
    con[10] = fbits(1.0f * rInW);                  con[11] = fbits(2.0f * rInH);
    con[12] = fbits(0.0f * rInW);                  con[13] = fbits(4.0f * rInH);
    con[14] = con[15] = 0;
}
// FsrRcasCon, ffx_fsr1.h:662-674 (FFSR1_RCAS::UpdateRCASConstantBlock, PostProcess.cpp:39-45)
void vqhip_fsr_rcas_con(uint32_t con[4], float sharpnessStops) {
    const float s = exp2f(-sharpnessStops);
    // ffx_a.h:482-550 packs the CPU-side half by TRUNCATION (table lookup on sign+exponent, mantissa shifted right), flushing
    // sub-denormals to zero and saturating overflow / inf / NaN to 65504 — not round-to-nearest: 0.2 stops gives 0x3af6, not 0x3af7
    const uint32_t u = fbits(s), sg = (u >> 16) & 0x8000u, e = (u >> 23) & 0xffu, m = u & 0x7fffffu;
    const uint32_t hb = e < 103 ? sg : e < 113 ? sg + (1u << (e - 103)) + (m >> (126 - e)) : e < 143 ? sg + ((e - 112) << 10) + (m >> 13) : sg + 0x7bffu;
    con[0] = fbits(s); con[1] = hb | (hb << 16); con[2] = 0; con[3] = 0;
}

static bool isColorFmt(int f) { return f == VQHIP_FMT_RGBA32F || f == VQHIP_FMT_RGBA16F || f == VQHIP_FMT_RGBA8_UNORM; }

int vqhip_fsr_easu(vqhip_ctx* ctx, void* stream, const void* in, int inW, int inH, vqhip_format inFmt, const uint32_t con[16],
                   void* out, int outW, int outH, vqhip_format outFmt) {
    vqk::Range range_("FSR-EASU CS");
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "fsr_easu: ctx is NULL");
    CTX_GUARD(ctx, "fsr_easu");
    if (!in || !out || !con || inW <= 0 || inH <= 0 || outW <= 0 || outH <= 0 || inW >= (1 << 24) || outW >= (1 << 24))
        return fail(ctx, VQHIP_ERR_INVALID_ARG, "fsr_easu: bad argument");
    if (!isColorFmt(inFmt) || !isColorFmt(outFmt)) return fail(ctx, VQHIP_ERR_UNSUPPORTED, "fsr_easu: formats must be RGBA8_UNORM, RGBA16F or RGBA32F");
    if (in == out) return fail(ctx, VQHIP_ERR_INVALID_ARG, "fsr_easu: in-place is not supported");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = launch_fsr_easu((hipStream_t)stream, in, inW, inH, inFmt, con, out, outW, outH, outFmt);
    return e == hipSuccess ? VQHIP_OK : failHip(ctx, e, "fsr_easu launch");
}

int vqhip_fsr_rcas(vqhip_ctx* ctx, void* stream, const void* in, void* out, int width, int height, const uint32_t con[4],
                   vqhip_format inFmt, vqhip_format outFmt) {
    vqk::Range range_("FSR-RCAS CS");
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "fsr_rcas: ctx is NULL");
    CTX_GUARD(ctx, "fsr_rcas");
    if (!in || !out || !con || width <= 0 || height <= 0 || width >= (1 << 24)) return fail(ctx, VQHIP_ERR_INVALID_ARG, "fsr_rcas: bad argument");
    if (!isColorFmt(inFmt) || !isColorFmt(outFmt)) return fail(ctx, VQHIP_ERR_UNSUPPORTED, "fsr_rcas: formats must be RGBA8_UNORM, RGBA16F or RGBA32F");
    if (in == out) return fail(ctx, VQHIP_ERR_INVALID_ARG, "fsr_rcas: in-place is not supported");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = launch_fsr_rcas((hipStream_t)stream, in, out, width, height, con, inFmt, outFmt);
    return e == hipSuccess ? VQHIP_OK : failHip(ctx, e, "fsr_rcas launch");
}

int vqhip_visualize(vqhip_ctx* ctx, void* stream, const void* in, void* out, int width, int height, const VQ_VizParams* params,
                    vqhip_format inFmt, vqhip_format outFmt) {
    vqk::Range range_("RenderPostProcess_DebugViz");      // :2543
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "visualize: ctx is NULL");
    CTX_GUARD(ctx, "visualize");
    if (!in || !out || !params || width <= 0 || height <= 0 || (uint64_t)width * height >= (1ull << 28)) return fail(ctx, VQHIP_ERR_INVALID_ARG, "visualize: bad argument");
    // inputs: whatever the draw mode's SRV holds (SceneRendering.cpp:2555-2566) — colour formats, Tex_SceneNormals (R10G10B10A2), Tex_SceneMotionVectors (RG16F | RG32F)
    const bool inOk = isColorFmt(inFmt) || inFmt == VQHIP_FMT_R10G10B10A2_UNORM || inFmt == VQHIP_FMT_RG16F || inFmt == VQHIP_FMT_RG32F;
    if (!inOk || !isColorFmt(outFmt)) return fail(ctx, VQHIP_ERR_UNSUPPORTED, "visualize: inFmt must be RGBA8_UNORM, RGBA16F, RGBA32F, RG16F, RG32F or R10G10B10A2_UNORM; outFmt RGBA8_UNORM, RGBA16F or RGBA32F");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = launch_visualize((hipStream_t)stream, in, out, width, height, *params, inFmt, outFmt);
    return e == hipSuccess ? VQHIP_OK : failHip(ctx, e, "visualize launch");
}

int vqhip_apply_reflections(vqhip_ctx* ctx, void* stream, const void* reflectionRadiance, void* sceneColor, int width, int height, vqhip_format fmt) {
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "apply_reflections: ctx is NULL");
    CTX_GUARD(ctx, "apply_reflections");
    return vqhip_composite_reflections(ctx, stream, reflectionRadiance, nullptr, sceneColor, width, height, fmt);
}

// VQRenderer::CompositeReflections (SceneRendering.cpp:2362-2403): ApplyReflectionsPass in the permutation its SRVBoundingVolumes selects (ApplyReflections.cpp:62-68)
int vqhip_composite_reflections(vqhip_ctx* ctx, void* stream, const void* reflectionRadiance, const void* boundingVolumes, void* sceneColor, int width, int height, vqhip_format fmt) {
    vqk::Range range_("CompositeReflections");            // :2374
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "composite_reflections: ctx is NULL");
    CTX_GUARD(ctx, "composite_reflections");
    if (!reflectionRadiance || !sceneColor || width <= 0 || height <= 0 || (uint64_t)width * height >= (1ull << 28)) return fail(ctx, VQHIP_ERR_INVALID_ARG, "composite_reflections: bad argument");
    if (!isImageFmt(fmt)) return fail(ctx, VQHIP_ERR_UNSUPPORTED, "composite_reflections: fmt must be RGBA32F or RGBA16F");
    if (boundingVolumes == sceneColor || reflectionRadiance == sceneColor) return fail(ctx, VQHIP_ERR_INVALID_ARG, "composite_reflections: the inputs must not alias the scene colour");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = launch_apply_reflections((hipStream_t)stream, reflectionRadiance, boundingVolumes, sceneColor, width, height, fmt);
    return e == hipSuccess ? VQHIP_OK : failHip(ctx, e, "composite_reflections launch");
}

int vqhip_ssr_environment_fallback(vqhip_ctx* ctx, void* stream, const void* sceneColorRoughness, vqhip_format sceneFmt, int scenePitchPx,
                                   const float* depth, int depthPitchPx, const void* normals, vqhip_format normalFmt, int normalPitchPx,
                                   int width, int height, const VQ_SSSRConstants* cb, const vqhip_envmap* env,
                                   void* outRadiance, vqhip_format outFmt, int outPitchPx, uint8_t* outExtractedRoughness) {
    vqk::Range range_("FFX DNSR ClassifyTiles");
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "ssr_environment_fallback: ctx is NULL");
    CTX_GUARD(ctx, "ssr_environment_fallback");
    if (!sceneColorRoughness || !depth || !normals || !cb || !env || !outRadiance || width <= 0 || height <= 0 || (uint64_t)width * height >= (1ull << 28))
        return fail(ctx, VQHIP_ERR_INVALID_ARG, "ssr_environment_fallback: bad argument");
    if (!isImageFmt(sceneFmt) || !isImageFmt(outFmt)) return fail(ctx, VQHIP_ERR_UNSUPPORTED, "ssr_environment_fallback: scene colour and radiance must be RGBA32F or RGBA16F");
    if (normalFmt != VQHIP_FMT_R10G10B10A2_UNORM && normalFmt != VQHIP_FMT_RGBA32F) return fail(ctx, VQHIP_ERR_UNSUPPORTED, "ssr_environment_fallback: normals must be R10G10B10A2_UNORM or RGBA32F");
    if (!env->specular_cube || !env->brdf_lut || env->spec_res0 <= 0 || env->spec_mips <= 0 || env->lut_size <= 0) return fail(ctx, VQHIP_ERR_INVALID_ARG, "ssr_environment_fallback: env needs the specular cube and the BRDF LUT");
    const uint32_t mc = cb->envMapSpecularIrradianceCubemapMipLevelCount;
    if (mc < 1 || mc > (uint32_t)env->spec_mips) return fail(ctx, VQHIP_ERR_INVALID_ARG, "ssr_environment_fallback: envMapSpecularIrradianceCubemapMipLevelCount must be in [1, env->spec_mips]");
    auto pitch = [&](int p) { return p ? p : width; };
    if (pitch(scenePitchPx) < width || pitch(depthPitchPx) < width || pitch(normalPitchPx) < width || pitch(outPitchPx) < width) return fail(ctx, VQHIP_ERR_INVALID_ARG, "ssr_environment_fallback: pitch < width");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    SsrArgs a;
    a.scene = sceneColorRoughness; a.depth = depth; a.normals = normals; a.out = outRadiance; a.outRoughness = outExtractedRoughness;
    a.width = width; a.height = height; a.scenePitch = pitch(scenePitchPx); a.depthPitch = pitch(depthPitchPx); a.normalPitch = pitch(normalPitchPx); a.outPitch = pitch(outPitchPx);
    a.invProj = cb->invProjection; a.view = cb->view; a.invView = cb->invView;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) a.rot[i][j] = cb->envMapRotation.m[i][j];
    a.invDimX = cb->inverseBufferDimensions[0]; a.invDimY = cb->inverseBufferDimensions[1];
    a.roughnessThreshold = cb->roughnessThreshold; a.mipCount = (float)mc;
    a.pow5ExpLog = ctx->pow5ExpLog; a.arithDxc = ctx->arithDxc; a.env = *env;
    hipError_t e = launch_ssr_env_fallback((hipStream_t)stream, a, sceneFmt, normalFmt, outFmt);
    return e == hipSuccess ? VQHIP_OK : failHip(ctx, e, "ssr_environment_fallback launch");
}

int vqhip_specular_mip_count(int spec_res0) { return vqhip_mip_level_count(spec_res0, spec_res0) - 1; }
size_t vqhip_cube_bytes(int res0, int nMips, vqhip_format fmt) {
    const size_t bpp = fmt == VQHIP_FMT_RGBA32F ? 16 : 8;
    size_t px = 0;
    for (int m = 0; m < nMips; ++m) { size_t r = (size_t)(res0 >> m); px += 6 * r * r; }
    return px * bpp;
}

static int checkChain(vqhip_ctx* ctx, const void* chain, int w0, int h0, int nMips, const char* who) {
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, std::string(who) + ": ctx is NULL");
    if (!chain || w0 <= 0 || h0 <= 0 || nMips <= 0 || nMips > vqhip_mip_level_count(w0, h0)) return fail(ctx, VQHIP_ERR_INVALID_ARG, std::string(who) + ": bad equirect chain");
    return VQHIP_OK;
}

int vqhip_conv_diffuse(vqhip_ctx* ctx, void* stream, const void* equirect_mips, int w0, int h0, int nMips,
                       int diffuseRes, float step, vqhip_conv_order order, void* outCube, vqhip_format fmt) {
    vqk::Range range_("DiffuseIrradianceCubemap");
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "conv_diffuse: ctx is NULL");
    CTX_GUARD(ctx, "conv_diffuse");                          // before anything writes the context's error string
    int rc = checkChain(ctx, equirect_mips, w0, h0, nMips, "conv_diffuse");
    if (rc) return rc;
    if (!outCube || diffuseRes <= 0 || !(step > 0.0f)) return fail(ctx, VQHIP_ERR_INVALID_ARG, "conv_diffuse: bad argument");
    if (!isImageFmt(fmt)) return fail(ctx, VQHIP_ERR_UNSUPPORTED, "conv_diffuse: fmt must be RGBA32F or RGBA16F");
    if (order != VQHIP_CONV_SEQUENTIAL && order != VQHIP_CONV_WAVE64) return fail(ctx, VQHIP_ERR_INVALID_ARG, "conv_diffuse: bad order");
    // fp32 sequences of `for (phi = 0; phi < TWO_PI; phi += step)` / `for (theta = 0; theta < PI_OVER_TWO; theta += step)`
    // (CubemapConvolution.hlsl:132-136): plain IEEE adds, identical on any host.
    const size_t maxFloats = kConstSlotBytes / sizeof(float);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    int slot;
    rc = acquireSlot(ctx, &slot);
    if (rc) return rc;
    float* tab = (float*)(ctx->hostRing + (size_t)slot * kConstSlotBytes);
    size_t n = 0; int nPhi = 0, nTheta = 0;
    for (float phi = 0.0f; phi < 6.28318530718f; phi += step) { if (n >= maxFloats) return fail(ctx, VQHIP_ERR_UNSUPPORTED, "conv_diffuse: step too small"); tab[n++] = phi; ++nPhi; }
    for (float th = 0.0f; th < 1.5707963268f; th += step) { if (n >= maxFloats) return fail(ctx, VQHIP_ERR_UNSUPPORTED, "conv_diffuse: step too small"); tab[n++] = th; ++nTheta; }
    if ((size_t)nTheta * 2 * sizeof(float) > 60 * 1024) return fail(ctx, VQHIP_ERR_UNSUPPORTED, "conv_diffuse: step too small for the LDS theta table");
    rc = commitSlot(ctx, slot, n * sizeof(float), st);
    if (rc) return rc;
    const float* dtab = (const float*)(ctx->devRing + (size_t)slot * kConstSlotBytes);
    const size_t recNeed = conv_diffuse_record_bytes(w0, h0, nMips);
    if (recNeed) {
        if (ctx->recUsed) HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->recFree, 0));
        if (ctx->recBytes < recNeed) {
            if (ctx->rec) { HIP_TRY(ctx, hipDeviceSynchronize()); HIP_TRY(ctx, hipFree(ctx->rec)); ctx->rec = nullptr; ctx->recBytes = 0; }
            HIP_TRY(ctx, hipMalloc(&ctx->rec, recNeed));
            ctx->recBytes = recNeed;
        }
        if (!ctx->recFree) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->recFree, hipEventDisableTiming));
    }
    hipError_t e = launch_conv_diffuse_tables(st, (const float4*)equirect_mips, w0, h0, nMips, diffuseRes, dtab, nPhi, dtab + nPhi, nTheta, order, outCube, fmt,
                                              recNeed ? ctx->rec : nullptr, ctx->opt);
    if (e != hipSuccess) return failHip(ctx, e, "conv_diffuse launch");
    if (recNeed) { HIP_TRY(ctx, hipEventRecord(ctx->recFree, st)); ctx->recUsed = true; }
    return releaseSlot(ctx, slot, st);
}

int vqhip_conv_specular(vqhip_ctx* ctx, void* stream, const void* equirect_mips, int w0, int h0, int nMips,
                        int specRes0, vqhip_conv_order order, void* outCubeMips, vqhip_format fmt) {
    vqk::Range range_("SpecularIrradianceCubemap");
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "conv_specular: ctx is NULL");
    CTX_GUARD(ctx, "conv_specular");
    int rc = checkChain(ctx, equirect_mips, w0, h0, nMips, "conv_specular");
    if (rc) return rc;
    const int MIPS = vqhip_specular_mip_count(specRes0);
    if (!outCubeMips || specRes0 < 4 || (specRes0 & (specRes0 - 1)) || MIPS < 2) return fail(ctx, VQHIP_ERR_INVALID_ARG, "conv_specular: specRes0 must be a power of two >= 4");
    if (!isImageFmt(fmt)) return fail(ctx, VQHIP_ERR_UNSUPPORTED, "conv_specular: fmt must be RGBA32F or RGBA16F");
    if (order != VQHIP_CONV_SEQUENTIAL && order != VQHIP_CONV_WAVE64) return fail(ctx, VQHIP_ERR_INVALID_ARG, "conv_specular: bad order");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = launch_conv_specular_all((hipStream_t)stream, (const float4*)equirect_mips, w0, h0, nMips, specRes0, MIPS, order, outCubeMips, fmt, ctx->opt);
    return e == hipSuccess ? VQHIP_OK : failHip(ctx, e, "conv_specular launch");
}

int vqhip_envmap_prefilter(vqhip_ctx* ctx, void* stream, const void* equirect_mips, int w0, int h0, int nMips,
                           int diffuseRes, float diffuseStep, int specRes0, vqhip_conv_order order, const vqhip_envmap_out* out) {
    vqk::Range range_("RenderEnvironmentMapCubeFaces");
    if (!ctx) return fail(nullptr, VQHIP_ERR_INVALID_ARG, "envmap_prefilter: ctx is NULL");
    CTX_GUARD(ctx, "envmap_prefilter");
    if (!out || !out->diffuse_blurred || !out->blur_tmp || !out->specular) return fail(ctx, VQHIP_ERR_INVALID_ARG, "envmap_prefilter: missing output buffer");
    void* diff = out->diffuse_unblurred;
    const size_t faceBytes = (size_t)diffuseRes * diffuseRes * 8;
    if (!diff) { int rc = ensureScratch(ctx, 6 * faceBytes); if (rc) return rc; diff = ctx->scratch; }
    int rc = vqhip_conv_diffuse(ctx, stream, equirect_mips, w0, h0, nMips, diffuseRes, diffuseStep, order, diff, VQHIP_FMT_RGBA16F);   // :181-277
    if (rc) return rc;
    VQ_BlurParams bp = { diffuseRes, diffuseRes };
    for (int face = 0; face < 6; ++face) {                                   // :279-373
        rc = vqhip_gaussian_blur_x(ctx, stream, (const char*)diff + face * faceBytes, out->blur_tmp, &bp, VQHIP_FMT_RGBA16F);
        if (rc) return rc;
        rc = vqhip_gaussian_blur_y(ctx, stream, out->blur_tmp, (char*)out->diffuse_blurred + face * faceBytes, nullptr, nullptr, 0, &bp, VQHIP_FMT_RGBA16F);
        if (rc) return rc;
    }
    return vqhip_conv_specular(ctx, stream, equirect_mips, w0, h0, nMips, specRes0, order, out->specular, VQHIP_FMT_RGBA16F);             // :386-472
}

} // extern "C"
