// scratch.hip -- temporaries of the one-shot entries (qdas_shift_sum: its tables; qdas_das_lut: misfit counter and partial images; qdas_greens: distance tables,
// bounds, sorted copies, tap list; qdas_convd's FFT path: filter spectrum and twiddles).
//
// They used to come from the device's stream-ordered pool (hipMallocAsync / hipFreeAsync).  On this platform that pool does not mix with callers that hipMalloc /
// hipFree around the calls: tools/repro/stale_pool.hip -- forty lines of plain HIP: a pool temporary written by one kernel and read by the next, staged buffers
// allocated and freed per call -- has the second call of a fresh process read ZEROS in 4 of 30 processes; 0 of 30 with the staged buffers kept, 0 of 29 with the
// temporaries from hipMalloc (profiles/r05/stale_read_repro.txt).  The gateway test had found it as `shiftsum` images that were wrong in every sample.
// So: one ARENA per (device, stream), hipMalloc'ed, kept.  Calls on one stream run in stream order, so the next call may reuse the arena as soon as it is
// issued; a call that needs more than the arena holds gets blocks of its own (released -- after a stream synchronisation -- when the stream's next call
// begins, and the arena is regrown to what the call needed in total, up to 8 x 64 MiB); a single request above 64 MiB never stays: such a block is freed when its call
// returns (which then waits for it).  The arena's mutex is held for the life of a Scratch object: one one-shot call at a time per (device, stream).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <memory>
#include <mutex>
#include <vector>
#include "qdas_kernels.h"

namespace qdas {

namespace {
// (QDAS_SCRATCH_ARENA_MAX_MB: read per call -- the tests set it to 0 to send every temporary down the per-call path)
size_t arena_max() { const char *e = getenv("QDAS_SCRATCH_ARENA_MAX_MB"); return (size_t)(e && atoll(e) >= 0 ? atoll(e) : 64) << 20; }
struct Arena {
    int dev = 0;
    hipStream_t s = nullptr;
    void *base = nullptr;
    size_t cap = 0, want = 0;
    std::vector<void *> extra;                           // blocks of calls that overflowed the arena: freed when the next call on the stream begins
    std::mutex mu;                                       // one call at a time per (device, stream)
};
std::mutex g_mu;
std::vector<std::unique_ptr<Arena>> g_arenas;

Arena *arena_of(int dev, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &a : g_arenas) if (a->dev == dev && a->s == s) return a.get();
    g_arenas.emplace_back(new Arena());
    g_arenas.back()->dev = dev; g_arenas.back()->s = s;
    return g_arenas.back().get();
}
size_t up256(size_t v) { return (v + 255) / 256 * 256; }
}  // namespace

Scratch::Scratch(hipStream_t s) : stream_(s) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    Arena *a = arena_of(dev, s);
    a->mu.lock();
    arena_ = a;
    if (!a->extra.empty() || a->want > a->cap) {         // the last call outgrew the arena: wait for it, release its blocks, regrow
        (void)hipStreamSynchronize(s);
        for (void *p : a->extra) (void)hipFree(p);
        a->extra.clear();
        if (a->want > a->cap) {
            if (a->base) (void)hipFree(a->base);
            a->base = nullptr; a->cap = 0;
            if (hipMalloc(&a->base, a->want) == hipSuccess) a->cap = a->want; else { (void)hipGetLastError(); a->base = nullptr; }
        }
    }
}

void *Scratch::get(size_t bytes) {
    Arena *a = (Arena *)arena_;
    const size_t b = up256(bytes ? bytes : 1);
    void *p = nullptr;
    const size_t ARENA_MAX = arena_max();
    if (b > ARENA_MAX) {                                 // never kept
        if (hipMalloc(&p, b) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        big_.push_back(p);
        return p;
    }
    if (off_ + b <= a->cap) { p = (char *)a->base + off_; off_ += b; return p; }
    off_ += b;
    // the arena regrows to the call's real total (clamped to ARENA_MAX, a call whose SMALL requests add up to more overflowed on every call -- a stream
    // synchronisation, hipFree and hipMalloc per call: the churn the arena exists to remove); a total beyond 8 x ARENA_MAX is not kept: that call pattern keeps the
    // per-call blocks, and each such call begins by waiting for the stream's previous one (include/qdas.h: one-shot entries may block)
    if (off_ > a->want && off_ <= 8 * ARENA_MAX) a->want = off_;
    if (hipMalloc(&p, b) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    a->extra.push_back(p);
    return p;
}

Scratch::~Scratch() {
    Arena *a = (Arena *)arena_;
    if (!big_.empty()) {
        (void)hipStreamSynchronize(stream_);
        for (void *p : big_) (void)hipFree(p);
    }
    a->mu.unlock();
}

// qdas_device_trim: arenas of streams that are idle are released (a stream that was destroyed leaves its arena behind until then)
void scratch_trim() {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &a : g_arenas) {
        if (!a->mu.try_lock()) continue;
        int prev = -1;
        const bool sw = hipGetDevice(&prev) == hipSuccess && prev != a->dev && hipSetDevice(a->dev) == hipSuccess;
        (void)hipDeviceSynchronize();
        for (void *p : a->extra) (void)hipFree(p);
        a->extra.clear();
        if (a->base) (void)hipFree(a->base);
        a->base = nullptr; a->cap = 0; a->want = 0;
        if (sw) (void)hipSetDevice(prev);
        (void)hipGetLastError();
        a->mu.unlock();
    }
}

}  // namespace qdas
