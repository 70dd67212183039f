// mock_rccl.cpp — TEST INFRASTRUCTURE: a stand-in for librccl.so that moves the bytes of ncclSend / ncclRecv through POSIX shared
// memory between the processes of ONE machine. It exists because the row-tiled multi-GPU path of the product
// (vqengine_amd/csrc/mgpu.hip: vqhip_exchange_blur_halos, vqhip_composite_tiles) calls RCCL through the C ABI, and the boxes tests run on
// have at most one GPU (RCCL refuses two ranks on one device) or none. Loaded through $VQHIP_RCCL_LIBRARY; never shipped, never used
// by bench.py outside its single-GPU debug mode (VQ_BENCH_SHARE_GPU).
//   * no GPU visible (this container): buffers are HOST memory, bytes move with memcpy           (tests/test_mgpu_mock.py)
//   * a GPU is visible: buffers are DEVICE memory and the transfers are ASYNCHRONOUS like RCCL's (round 5): ncclGroupEnd records an event on the caller's stream,
//     hands the group to the communicator's worker thread and makes the stream wait (hipStreamWaitValue32 on a signal word) — it returns at once. The worker waits
//     for the event (the data the group sends has been produced), moves the bytes through the mailbox with copies on its own non-blocking stream, then writes the
//     signal word from that stream. Two streams / two communicators therefore progress in whatever order their events fire: a missing hipStreamWaitEvent between
//     the post kernel and the composite, or between the composite and the next frame's kernels, shows up as wrong bytes (tests/test_gpu_bench_flow.py), which a
//     mock that drains the stream at every call cannot show. Devices without stream wait-value support (hipDeviceAttributeCanUseStreamWaitValue) and
//     $VQMOCK_RCCL_SYNC=1 fall back to the synchronous form of rounds 1-4 (drain the stream, copy, return); vqmock_rccl_async() says which.
//     (tests/test_gpu_bench_flow.py: N ranks sharing one GPU)
// Semantics kept from RCCL: sends / receives between GroupStart and GroupEnd are posted together and progress concurrently (no ordering
// deadlock), messages between one ordered pair of ranks match ONE TO ONE in posting order and a send whose byte count differs from the
// receive it meets ABORTS the process (real RCCL hangs or corrupts there: ten row-sized sends against one ten-row receive must not pass).
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <hip/hip_runtime_api.h>

extern "C" {
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef struct MockComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
}

namespace {
constexpr int kMaxRanks = 8;
constexpr size_t kSlotBytes = 4u << 20;                    // one in-flight chunk per ordered pair
struct Mailbox { std::atomic<uint64_t> produced, consumed; std::atomic<uint64_t> bytes; std::atomic<uint64_t> total; char pad[32]; };
struct Shared { std::atomic<int> arrived; char pad[60]; Mailbox box[kMaxRanks][kMaxRanks]; };   // payload slots follow
size_t sharedBytes() { return sizeof(Shared) + (size_t)kMaxRanks * kMaxRanks * kSlotBytes; }
char* slot(Shared* s, int src, int dst) { return (char*)(s + 1) + ((size_t)src * kMaxRanks + dst) * kSlotBytes; }
bool haveGpu() { static int n = -1; if (n < 0) { int c = 0; n = (hipGetDeviceCount(&c) == hipSuccess && c > 0) ? 1 : 0; } return n == 1; }

struct Op { bool send; char* buf; size_t bytes, done; int peer; hipStream_t st; };
struct Group { std::vector<Op> ops; std::vector<hipEvent_t> ready; uint32_t seq; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local MockComm* g_comm = nullptr;
}

struct MockComm {
    Shared* sh; int world, rank; char name[64];
    // asynchronous mode (a GPU is visible and supports stream wait-value)
    bool async = false; int device = 0;
    uint32_t* signal = nullptr;                              // hipMallocSignalMemory word: the worker writes the sequence number of the last finished group
    uint32_t seq = 0;
    hipStream_t copyStream = nullptr;
    std::thread worker; std::mutex mu; std::condition_variable cv; std::deque<Group> queue; std::atomic<bool> stop{false}; bool busy = false; std::condition_variable idle;
};

namespace {
thread_local hipStream_t t_copyStream = nullptr;             // set inside a worker thread: copies go through the communicator's own stream
void copyIn(void* dst, const void* src, size_t n) {
    if (!haveGpu()) { std::memcpy(dst, src, n); return; }
    if (t_copyStream) { (void)hipMemcpyAsync(dst, src, n, hipMemcpyDefault, t_copyStream); (void)hipStreamSynchronize(t_copyStream); }
    else (void)hipMemcpy(dst, src, n, hipMemcpyDefault);
}
size_t elemSize(ncclDataType_t t) { switch (t) { case 0: case 1: return 1; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; case 6: case 9: return 2; } return 1; }

// one non-blocking step of an operation; returns true when it has finished
bool progress(MockComm* c, Op& op) {
    if (op.done == op.bytes) return true;
    const int src = op.send ? c->rank : op.peer, dst = op.send ? op.peer : c->rank;
    Mailbox& m = c->sh->box[src][dst];
    if (op.send) {
        if (m.produced.load(std::memory_order_acquire) != m.consumed.load(std::memory_order_acquire)) return false;     // slot still full
        const size_t n = std::min(kSlotBytes, op.bytes - op.done);
        copyIn(slot(c->sh, src, dst), op.buf + op.done, n);
        if (op.done == 0) m.total.store(op.bytes, std::memory_order_relaxed);       // size of the whole message, checked by the receive it meets
        m.bytes.store(n, std::memory_order_relaxed);
        m.produced.fetch_add(1, std::memory_order_release);
        op.done += n;
    } else {
        if (m.produced.load(std::memory_order_acquire) == m.consumed.load(std::memory_order_acquire)) return false;     // nothing there yet
        const size_t n = m.bytes.load(std::memory_order_relaxed);
        if (op.done == 0 && m.total.load(std::memory_order_relaxed) != op.bytes) {
            std::fprintf(stderr, "mock_rccl: rank %d posted a receive of %zu bytes from rank %d, the matching send has %llu bytes (RCCL requires equal counts)\n",
                         c->rank, op.bytes, op.peer, (unsigned long long)m.total.load());
            std::abort();
        }
        if (n > op.bytes - op.done) { std::fprintf(stderr, "mock_rccl: message larger than the posted receive\n"); std::abort(); }
        copyIn(op.buf + op.done, slot(c->sh, src, dst), n);
        m.consumed.fetch_add(1, std::memory_order_release);
        op.done += n;
    }
    return op.done == op.bytes;
}
// the exchange itself: per ordered pair only the oldest unfinished operation may progress (messages match in posting order)
void exchange(MockComm* c, std::vector<Op>& ops) {
    for (bool all = false; !all;) {
        if (c->stop.load(std::memory_order_acquire)) return;  // ncclCommAbort: the transfers in flight end where they are
        all = true;
        bool blockedSend[kMaxRanks] = {}, blockedRecv[kMaxRanks] = {};
        for (Op& op : ops) {
            if (op.done == op.bytes && op.bytes) continue;
            bool& blocked = op.send ? blockedSend[op.peer] : blockedRecv[op.peer];
            if (blocked) { all = false; continue; }
            if (!progress(c, op)) { blocked = true; all = false; }
            else if (op.done != op.bytes) { blocked = true; all = false; }
        }
        if (!all) sched_yield();
    }
}
void workerMain(MockComm* c) {
    (void)hipSetDevice(c->device);
    t_copyStream = c->copyStream;
    for (;;) {
        Group g;
        {
            std::unique_lock<std::mutex> lk(c->mu);
            c->cv.wait(lk, [&] { return c->stop.load() || !c->queue.empty(); });
            if (c->stop.load() || c->queue.empty()) break;
            g = std::move(c->queue.front());
            c->queue.pop_front();
            c->busy = true;
        }
        for (hipEvent_t e : g.ready) { (void)hipEventSynchronize(e); (void)hipEventDestroy(e); }      // what the group sends has been produced
        exchange(c, g.ops);
        (void)hipStreamWriteValue32(c->copyStream, c->signal, g.seq, 0);                               // behind the copies of this group on the same stream
        (void)hipStreamSynchronize(c->copyStream);
        { std::lock_guard<std::mutex> lk(c->mu); c->busy = false; }
        c->idle.notify_all();
    }
    (void)hipStreamWriteValue32(c->copyStream, c->signal, 0xffffffffu, 0);                              // destroy / abort: whatever still waits on this communicator is released
    (void)hipStreamSynchronize(c->copyStream);
}
void startAsync(MockComm* c) {
    const char* env = std::getenv("VQMOCK_RCCL_SYNC");
    if (!haveGpu() || (env && env[0] == '1')) return;
    int can = 0;
    if (hipGetDevice(&c->device) != hipSuccess || hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, c->device) != hipSuccess || !can) return;
    if (hipExtMallocWithFlags((void**)&c->signal, 8, hipMallocSignalMemory) != hipSuccess) return;
    // The worker's stream must never share a hardware queue with a stream of the caller: a caller's stream that sits in hipStreamWaitValue32 blocks its hardware queue, and
    // the copies + the signal write that would release it would queue up behind it (HIP multiplexes streams onto GPU_MAX_HW_QUEUES = 4 queues per priority level;
    // the two-communicator mode has five streams). Streams of another priority level come from another pool of queues, and nothing on this one ever waits.
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (hipMemset(c->signal, 0, 8) != hipSuccess || hipStreamCreateWithPriority(&c->copyStream, hipStreamNonBlocking, greatest) != hipSuccess) { (void)hipFree(c->signal); c->signal = nullptr; return; }
    c->async = true;
    c->worker = std::thread(workerMain, c);
}
void stopAsync(MockComm* c, bool drain) {                    // drain: ncclCommDestroy finishes what was posted; ncclCommAbort does not
    if (!c->async) return;
    if (drain) { std::unique_lock<std::mutex> lk(c->mu); c->idle.wait(lk, [&] { return c->queue.empty() && !c->busy; }); }
    { std::lock_guard<std::mutex> lk(c->mu); c->stop.store(true, std::memory_order_release); }
    c->cv.notify_all();
    c->worker.join();
    (void)hipStreamDestroy(c->copyStream);
    (void)hipFree(c->signal);
}
void runGroup() {
    MockComm* c = g_comm;
    if (!c) { g_ops.clear(); return; }
    if (c->async) {
        Group g;
        g.ops = std::move(g_ops);
        g_ops.clear();
        std::vector<hipStream_t> streams;
        for (const Op& op : g.ops) { bool seen = false; for (hipStream_t s : streams) seen |= s == op.st; if (!seen) streams.push_back(op.st); }
        for (hipStream_t s : streams) { hipEvent_t e; (void)hipEventCreateWithFlags(&e, hipEventDisableTiming); (void)hipEventRecord(e, s); g.ready.push_back(e); }
        g.seq = ++c->seq;
        const uint32_t seq = g.seq;
        { std::lock_guard<std::mutex> lk(c->mu); c->queue.push_back(std::move(g)); }
        c->cv.notify_one();
        for (hipStream_t s : streams) (void)hipStreamWaitValue32(s, c->signal, seq, hipStreamWaitValueGte, 0xffffffffu);      // later work on the stream waits for the group; the host does not
        return;
    }
    if (haveGpu()) for (const Op& op : g_ops) (void)hipStreamSynchronize(op.st);       // synchronous form: the data was produced on the caller's stream
    exchange(c, g_ops);
    g_ops.clear();
}
}

extern "C" {
__attribute__((visibility("default"))) int vqmock_rccl_host_buffers() { return haveGpu() ? 0 : 1; }
__attribute__((visibility("default"))) int vqmock_rccl_async(ncclComm_t c) { return c && c->async ? 1 : 0; }
__attribute__((visibility("default"))) const char* ncclGetErrorString(ncclResult_t) { return "mock_rccl error"; }
__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::memset(id, 0, sizeof(*id));
    std::snprintf(id->internal, sizeof(id->internal), "/vqmock_%d_%ld", (int)getpid(), (long)random());
    const int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)sharedBytes()) != 0) return 2;
    close(fd);                                              // zero-filled: counters start at 0
    return 0;
}
__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank) {
    if (world > kMaxRanks) return 4;
    const int fd = shm_open(id.internal, O_RDWR, 0600);
    if (fd < 0) return 2;
    void* p = mmap(nullptr, sharedBytes(), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 2;
    MockComm* c = new MockComm();
    c->sh = (Shared*)p; c->world = world; c->rank = rank;
    std::memcpy(c->name, id.internal, sizeof(c->name) - 1);      // the name written by ncclGetUniqueId is < 40 characters
    startAsync(c);
    c->sh->arrived.fetch_add(1);
    while (c->sh->arrived.load() < world) sched_yield();    // collective, like the real one
    if (rank == 0) shm_unlink(c->name);                     // every rank has it mapped: the name can go
    *out = c;
    return 0;
}
__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t c) { if (c) { stopAsync(c, true); munmap(c->sh, sharedBytes()); delete c; } return 0; }
__attribute__((visibility("default"))) ncclResult_t ncclCommAbort(ncclComm_t c) { if (c) { stopAsync(c, false); munmap(c->sh, sharedBytes()); delete c; } return 0; }
__attribute__((visibility("default"))) ncclResult_t ncclGetVersion(int* v) { *v = 0; return 0; }          // 0 = this stand-in
__attribute__((visibility("default"))) ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { *n = c->world; return 0; }
__attribute__((visibility("default"))) ncclResult_t ncclCommUserRank(const ncclComm_t c, int* r) { *r = c->rank; return 0; }
__attribute__((visibility("default"))) ncclResult_t ncclGroupStart() { ++g_depth; return 0; }
__attribute__((visibility("default"))) ncclResult_t ncclGroupEnd() { if (--g_depth == 0) runGroup(); return 0; }
__attribute__((visibility("default"))) ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
    g_comm = c; g_ops.push_back({ true, (char*)buf, count * elemSize(t), 0, peer, st });
    if (g_depth == 0) runGroup();
    return 0;
}
__attribute__((visibility("default"))) ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
    g_comm = c; g_ops.push_back({ false, (char*)buf, count * elemSize(t), 0, peer, st });
    if (g_depth == 0) runGroup();
    return 0;
}
}
