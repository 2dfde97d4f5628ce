// sharded.hip -- ONE host thread, several HIP devices: the multi-device entry of the C ABI (include/qdas.h qdas_plan_*_sharded).
//
// The reference drives one gpuDevice per MATLAB process (reference README.md:232) and has no multi-device path; SURVEY.md
// section 8b/8e ask for `qdas_plan_create_sharded(..., ndev)` + gather so that a single-process caller (MATLAB through the MEX
// gateway, any C program) reaches all GPUs of a node.  Layout, exactly as the per-process layout of qups_amd/dist.py: pixels are
// split into ndev contiguous slabs of the linear pixel index; geometry and channel data are REPLICATED; every device beamforms
// its slab with an ordinary plan (qdas_plan_create with i_begin / i_count); the slabs are concatenated into the caller's y.
// No collective library is needed for that: the data path is
//     x: caller -> devices[0] (H2D if the caller's memory is host memory) -> SCATTER + ALL-GATHER with peer copies over the distinct
//        devices: xGMI is a point-to-point mesh (every GPU has its own link to every other GPU of the node), so the frame is cut into
//        U-1 pieces, device u pulls piece u from the root over the link (0,u), then every device pulls the other pieces from their
//        holders over the links (u,v) -- all links carry 1/(U-1) of the frame at the same time: 2 |x| / ((U-1) B_link) instead of the
//        log2(U) |x| / B_link of a tree (each pull runs on its own stream of the destination device, ordered with events only)
//     y: slab g -> peer copy into its place in y on devices[0] (-> one D2H for host callers)
// and the host thread never blocks before the final wait.  A device starts its kernel when its replica is complete: every pixel sums
// over ALL (receiver, transmit) traces, and summing partial apertures as the pieces arrive would change the summation order -- the
// sharded image is bit-identical to the single-plan image, by construction and by test.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
extern "C" int qdas_internal_upload(void *dst, const void *src, size_t bytes);      // qdas_api.hip: host -> device through pinned staging

#include "../../include/qdas.h"

void qdas_internal_set_error(const char *msg);         // qdas_api.hip: the thread-local message behind qdas_last_error()

namespace {

struct Shard {
    int device = 0;
    qdas_plan *plan = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t x_ready = nullptr, done = nullptr;
    uint64_t i_begin = 0, i_count = 0;
    void *x = nullptr;                                  // this device's replica of the frame (null: shares the root's)
    void *y = nullptr;                                  // slab output, i_count x [N] x [M]
    std::vector<void *> owned;                          // device copies of geometry made for this shard
    qdas_desc d{};                                      // this shard's problem (its own device copies of the constant inputs)
    // replication (holders only: the first shard of every distinct device): one pull stream per source holder, an event per pull,
    // and the event that says "my own piece has arrived" (what the other holders' pulls of that piece wait for)
    std::vector<hipStream_t> pull_stream;
    std::vector<hipEvent_t> pull_done;
    hipEvent_t piece_ready = nullptr;
};

// the calling thread's current device is restored on every return path
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    DeviceGuard() { ok = hipGetDevice(&prev) == hipSuccess; }
    ~DeviceGuard() { if (ok) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

int failf(int code, const char *fmt, const char *a = "", const char *b = "") {
    char buf[512];
    snprintf(buf, sizeof buf, fmt, a, b);
    qdas_internal_set_error(buf);
    return code;
}
#define SHIP(call)                                                                                   \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) return failf(QDAS_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

size_t real_size(int dtype) { return dtype == QDAS_F64 ? 8 : 4; }
size_t data_size(int dtype) { return dtype == QDAS_F64 ? 16 : (dtype == QDAS_F32 ? 8 : 4); }
size_t apod_real_size(int dtype) { return dtype == QDAS_F64 ? 8 : (dtype == QDAS_F32 ? 4 : 2); }

uint64_t bcast_numel(const uint64_t *st, const qdas_sizes &z) {
    const uint64_t dims[5] = {z.I1, z.I2, z.I3, z.N, z.M};
    uint64_t n = 1;
    for (int k = 0; k < 5; ++k) if (st[k]) n += (dims[k] - 1) * st[k];
    return n;
}

}  // namespace

struct qdas_sharded_plan {
    qdas_desc d{};
    std::vector<Shard> sh;
    std::vector<uint64_t> acs;
    uint64_t I = 0, oN = 1, oM = 1;
    size_t x_bytes = 0, ds = 0;
    void *y_root = nullptr;                             // host callers: the gathered image on devices[0]
    void *x_root = nullptr;                             // host callers: the uploaded frame on devices[0]
    std::vector<int> holder;                            // shard index of the first shard on each distinct device (holder[0] = 0)
    std::vector<int> uidx;                              // shard -> index into holder
    bool executed = false;                              // (the `done` events of a previous frame exist)
    bool mirror = false;                                // mirror slabs (QDAS_PLAN_MIRROR_SLAB): shard g = columns of the first half + their mirror images
    ~qdas_sharded_plan() {
        DeviceGuard guard;
        for (Shard &s : sh) {
            (void)hipSetDevice(s.device);
            (void)hipDeviceSynchronize();               // nothing of this plan may still be in flight when its buffers go
            if (s.plan) qdas_plan_destroy(s.plan);
            for (hipStream_t ps : s.pull_stream) if (ps) (void)hipStreamDestroy(ps);
            for (hipEvent_t pe : s.pull_done) if (pe) (void)hipEventDestroy(pe);
            if (s.piece_ready) (void)hipEventDestroy(s.piece_ready);
            for (void *p : s.owned) (void)hipFree(p);
            if (s.x) (void)hipFree(s.x);
            if (s.y) (void)hipFree(s.y);
            if (s.x_ready) (void)hipEventDestroy(s.x_ready);
            if (s.done) (void)hipEventDestroy(s.done);
            if (s.stream) (void)hipStreamDestroy(s.stream);
        }
        if (!sh.empty()) (void)hipSetDevice(sh[0].device);
        if (y_root) (void)hipFree(y_root);
        if (x_root) (void)hipFree(x_root);
    }
};

extern "C" int qdas_plan_create_sharded(qdas_sharded_plan **out, const qdas_desc *desc, int ndev, const int *devices) {
    if (!out || !desc) return failf(QDAS_EINVAL, "null argument");
    *out = nullptr;
    if (ndev < 1 || ndev > 64) return failf(QDAS_EINVAL, "qdas_plan_create_sharded: ndev must be in 1..64");
    if (desc->i_begin || desc->i_count || desc->y_ld) return failf(QDAS_EINVAL, "qdas_plan_create_sharded: the library chooses the pixel slabs (i_begin, i_count, y_ld must be 0)");
    int have = 0;
    SHIP(hipGetDeviceCount(&have));
    DeviceGuard guard;
    qdas_sharded_plan *sp = new qdas_sharded_plan();
    auto bail = [&](int code) { delete sp; return code; };
    sp->d = *desc;
    const qdas_sizes &z = desc->sz;
    sp->I = z.I1 * z.I2 * z.I3;
    sp->oN = (z.flag & QDAS_FLAG_KEEP_RX) ? z.N : 1;
    sp->oM = (z.flag & QDAS_FLAG_KEEP_TX) ? z.M : 1;
    sp->ds = data_size(z.dtype);
    sp->x_bytes = (size_t)z.T * z.N * z.M * sp->ds;
    sp->acs.assign(desc->acstride ? desc->acstride : nullptr, desc->acstride ? desc->acstride + 6 * (1 + z.S) : nullptr);
    int root_dev = devices ? devices[0] : 0;
    if (desc->mem == QDAS_MEM_DEVICE && desc->device >= 0 && devices && desc->device != devices[0])
        return bail(failf(QDAS_EINVAL, "qdas_plan_create_sharded: device-resident inputs must live on devices[0]"));
    sp->sh.resize(ndev);
    // sizes of the constant inputs (as qdas_plan_create computes them)
    const size_t rs = real_size(z.dtype);
    uint64_t apod_elems = 0;
    for (uint64_t s = 0; s < z.S && desc->acstride; ++s) {
        const uint64_t *a = desc->acstride + 6 * (1 + s);
        const uint64_t end = a[5] + bcast_numel(a, z);
        if (end > apod_elems) apod_elems = end;
    }
    const size_t ael = desc->apod_real ? apod_real_size(z.dtype) : sp->ds;
    const uint64_t cinv_elems = desc->acstride ? bcast_numel(desc->acstride, z) : 1;
    for (int g = 0; g < ndev; ++g) {
        Shard &s = sp->sh[g];
        s.device = devices ? devices[g] : g;
        if (s.device < 0 || s.device >= have) return bail(failf(QDAS_EINVAL, "qdas_plan_create_sharded: device ordinal out of range"));
        s.i_begin = sp->I * (uint64_t)g / (uint64_t)ndev;
        s.i_count = sp->I * (uint64_t)(g + 1) / (uint64_t)ndev - s.i_begin;
        hipError_t e = hipSetDevice(s.device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.x_ready, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
        if (e != hipSuccess) return bail(failf(QDAS_EHIP, "qdas_plan_create_sharded: stream / event creation: %s", hipGetErrorString(e)));
        qdas_desc d = *desc;
        if (s.device != root_dev || desc->mem == QDAS_MEM_HOST) d.plan_flags &= ~QDAS_PLAN_COPY_INPUTS;   // (this shard's inputs are plan-owned copies already)
        d.device = s.device;
        d.i_begin = s.i_begin; d.i_count = s.i_count; d.y_ld = 0;
        d.acstride = sp->acs.empty() ? nullptr : sp->acs.data();
        d.mem = QDAS_MEM_DEVICE;                         // shards always run on device-resident frames (the sharded plan moves the data)
        // constant inputs: host arrays are uploaded per device; device arrays (on devices[0]) are replicated with peer copies
        auto bring = [&](const void *src, size_t bytes, const void **dst) -> int {
            *dst = src;
            if (!src || !bytes) return QDAS_OK;
            if (desc->mem == QDAS_MEM_DEVICE && s.device == root_dev) return QDAS_OK;
            void *p = nullptr;
            hipError_t e2 = hipMalloc(&p, bytes);
            if (e2 != hipSuccess) return failf(QDAS_ENOMEM, "qdas_plan_create_sharded: hipMalloc: %s", hipGetErrorString(e2));
            s.owned.push_back(p);
            e2 = desc->mem == QDAS_MEM_HOST ? (hipError_t)qdas_internal_upload(p, src, bytes) : hipMemcpyPeer(p, s.device, src, root_dev, bytes);
            if (e2 != hipSuccess) return failf(QDAS_EHIP, "qdas_plan_create_sharded: replicating inputs: %s", hipGetErrorString(e2));
            *dst = p;
            return QDAS_OK;
        };
        int rc = 0;
        if (sp->I && z.N && z.M) {
            if ((rc = bring(desc->Pi, 3 * sp->I * rs, &d.Pi)) || (rc = bring(desc->Pr, 3 * z.N * rs, &d.Pr)) || (rc = bring(desc->Pv, 4 * z.M * rs, &d.Pv)) ||
                (rc = bring(desc->Nv, 3 * z.M * rs, &d.Nv)) || (rc = bring(desc->cinv, cinv_elems * rs, &d.cinv)) ||
                (rc = bring(desc->apod, apod_elems * ael, &d.apod)) || (rc = bring(desc->rx_normals, desc->rx_apod_kind ? 3 * z.N * rs : 0, &d.rx_normals)))
                return bail(rc);
        }
        s.d = d;                                         // (plans are created below, once the slab layout is decided)
        // replica of the frame: one per distinct device (a device listed twice shares it); the root needs one only for host callers.
        // QDAS_SHARDED_FORCE_REPLICAS=1 (tests): every shard is its own holder even on a shared device, so that the replication
        // machinery -- pull streams, events, pieces -- runs and is checked on a single GPU.
        const bool force = getenv("QDAS_SHARDED_FORCE_REPLICAS") != nullptr;
        int first = g;
        if (!force) for (int q = 0; q < g; ++q) if (sp->sh[q].device == s.device) { first = q; break; }
        if (first == g) sp->holder.push_back(g);
        sp->uidx.push_back(0);
        for (size_t k = 0; k < sp->holder.size(); ++k) if (sp->holder[k] == first) sp->uidx[g] = (int)k;
        if (first == g && !(g == 0 && desc->mem == QDAS_MEM_DEVICE) && sp->x_bytes) {
            e = hipSetDevice(s.device);
            if (e == hipSuccess) e = hipMalloc(&s.x, sp->x_bytes);
            if (e != hipSuccess) return bail(failf(QDAS_ENOMEM, "qdas_plan_create_sharded: frame replica: %s", hipGetErrorString(e)));
        }
    }
    if (desc->mem == QDAS_MEM_HOST) {
        hipError_t e = hipSetDevice(root_dev);
        if (e == hipSuccess) e = hipMalloc(&sp->y_root, (size_t)sp->I * sp->oN * sp->oM * sp->ds + 16);
        if (e != hipSuccess) return bail(failf(QDAS_ENOMEM, "qdas_plan_create_sharded: gather buffer: %s", hipGetErrorString(e)));
    }
    // ---- the shards' plans.  First choice: MIRROR SLABS -- shard g takes columns [c0, c1) of the first half of the image and their mirror
    //      images (QDAS_PLAN_MIRROR_SLAB), so that the lateral-mirror mode of the fused kernel survives the sharding; every shard's plan must
    //      accept it (a mirror-symmetric geometry, 'DAS', an even number of columns ...), else plain contiguous slabs.
    auto make_plans = [&](bool mirror) -> int {
        for (int g = 0; g < ndev; ++g) {
            Shard &s = sp->sh[g];
            if (mirror) {
                const uint64_t h = z.I2 / 2, c0 = h * (uint64_t)g / (uint64_t)ndev, c1 = h * (uint64_t)(g + 1) / (uint64_t)ndev;
                s.i_begin = c0 * z.I1; s.i_count = (c1 - c0) * z.I1;
            } else {
                s.i_begin = sp->I * (uint64_t)g / (uint64_t)ndev;
                s.i_count = sp->I * (uint64_t)(g + 1) / (uint64_t)ndev - s.i_begin;
            }
            if (!s.i_count) continue;
            qdas_desc d = s.d;
            d.i_begin = s.i_begin; d.i_count = s.i_count; d.y_ld = 0;
            if (mirror) d.plan_flags |= QDAS_PLAN_MIRROR_SLAB;
            int rc = qdas_plan_create(&s.plan, &d);
            if (rc) return rc;
            hipError_t e = hipSetDevice(s.device);
            if (e == hipSuccess) e = hipMalloc(&s.y, (size_t)s.i_count * (mirror ? 2 : 1) * sp->oN * sp->oM * sp->ds + 16);
            if (e != hipSuccess) return failf(QDAS_ENOMEM, "qdas_plan_create_sharded: slab buffer: %s", hipGetErrorString(e));
        }
        return QDAS_OK;
    };
    auto drop_plans = [&]() {
        for (Shard &s : sp->sh) {
            (void)hipSetDevice(s.device);
            if (s.plan) { qdas_plan_destroy(s.plan); s.plan = nullptr; }
            if (s.y) { (void)hipFree(s.y); s.y = nullptr; }
        }
    };
    {
        const bool try_mirror = ndev > 1 && sp->oN == 1 && sp->oM == 1 && z.I3 == 1 && z.I2 >= 2 && z.I2 % 2 == 0 && sp->I && z.N && z.M
                                && !(desc->plan_flags & QDAS_PLAN_NO_MIRROR) && desc->kernel != QDAS_KERNEL_GENERIC && !getenv("QDAS_NO_MIRROR");
        int rc = try_mirror ? make_plans(true) : QDAS_EUNSUPPORTED;
        if (rc == QDAS_OK) sp->mirror = true;
        else {
            drop_plans();
            if (try_mirror && rc != QDAS_EUNSUPPORTED && rc != QDAS_EINVAL) return bail(rc);
            if ((rc = make_plans(false))) return bail(rc);
        }
    }
    // ---- replication plumbing: a pull stream + event per (destination holder, source holder), direct access between every pair
    //      of distinct devices (peer copies between devices without it are staged through the host)
    const int U = (int)sp->holder.size();
    for (int u = 0; u < U; ++u) {
        Shard &hu = sp->sh[sp->holder[u]];
        hipError_t e = hipSetDevice(hu.device);
        if (e != hipSuccess) return bail(failf(QDAS_EHIP, "qdas_plan_create_sharded: hipSetDevice: %s", hipGetErrorString(e)));
        for (int v = 0; v < U; ++v) {
            const int pd = sp->sh[sp->holder[v]].device;
            if (pd == hu.device) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, hu.device, pd) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(pd, 0);
            (void)hipGetLastError();                     // (already enabled is fine)
        }
        if (u == 0 || U < 2) continue;
        hu.pull_stream.assign(U, nullptr);
        hu.pull_done.assign(U, nullptr);
        e = hipEventCreateWithFlags(&hu.piece_ready, hipEventDisableTiming);
        for (int v = 0; v < U && e == hipSuccess; ++v) {
            if (v == u) continue;
            e = hipStreamCreateWithFlags(&hu.pull_stream[v], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&hu.pull_done[v], hipEventDisableTiming);
        }
        if (e != hipSuccess) return bail(failf(QDAS_EHIP, "qdas_plan_create_sharded: pull stream / event creation: %s", hipGetErrorString(e)));
    }
    *out = sp;
    return QDAS_OK;
}

extern "C" int qdas_plan_execute_sharded(qdas_sharded_plan *sp, const void *x, void *y, void *stream) {
    if (!sp || !y) return failf(QDAS_EINVAL, "null argument");
    const qdas_sizes &z = sp->d.sz;
    const int G = (int)sp->sh.size();
    const int root_dev = sp->sh[0].device;
    const bool host = sp->d.mem == QDAS_MEM_HOST;
    const size_t ybytes = (size_t)sp->I * sp->oN * sp->oM * sp->ds;
    if (sp->I == 0 || z.N == 0 || z.M == 0 || z.T == 0) {                // empty sum: zeros
        if (!ybytes) return QDAS_OK;
        if (host) { memset(y, 0, ybytes); return QDAS_OK; }
        DeviceGuard guard0;
        SHIP(hipSetDevice(root_dev));
        SHIP(hipMemsetAsync(y, 0, ybytes, (hipStream_t)stream));
        return QDAS_OK;
    }
    if (!x) return failf(QDAS_EINVAL, "null data");
    DeviceGuard guard;
    // ---- the frame on the root device
    Shard &r0 = sp->sh[0];
    const void *x0 = x;
    SHIP(hipSetDevice(root_dev));
    if (host) {
        void *dst = r0.x;                                // (the root owns a replica for host callers)
        if (!dst) return failf(QDAS_EINVAL, "qdas_plan_execute_sharded: internal: no root replica");
        SHIP(hipMemcpyAsync(dst, x, sp->x_bytes, hipMemcpyHostToDevice, r0.stream));
        x0 = dst;
    } else if (stream != (void *)r0.stream) {            // order behind the caller's stream on the root device
        SHIP(hipEventRecord(r0.done, (hipStream_t)stream));
        SHIP(hipStreamWaitEvent(r0.stream, r0.done, 0));
    }
    SHIP(hipEventRecord(r0.x_ready, r0.stream));
    // ---- replicate: scatter + all-gather over the holders (the first shard of every distinct device).  Piece p (p = 1 .. U-1, cut at
    //      256-byte boundaries) belongs to holder p: it pulls it from the root, everybody else pulls it from holder p.
    const int U = (int)sp->holder.size();
    std::vector<const void *> xr(U, nullptr);
    xr[0] = x0;
    if (U >= 2) {
        const size_t K = (size_t)U - 1;
        auto cut = [&](size_t p) { return p >= K ? sp->x_bytes : (sp->x_bytes / K * p) & ~(size_t)255; };
        for (int u = 1; u < U; ++u) {                    // scatter
            Shard &dst = sp->sh[sp->holder[u]];
            if (!dst.x) return failf(QDAS_EINVAL, "qdas_plan_execute_sharded: internal: missing replica buffer");
            xr[u] = dst.x;
            const size_t b = cut((size_t)u - 1), e = cut((size_t)u);
            SHIP(hipSetDevice(dst.device));
            hipStream_t ps = dst.pull_stream[0];
            SHIP(hipStreamWaitEvent(ps, r0.x_ready, 0));
            // the replica's last readers: the previous frame's kernels on this device, and the other holders' pulls of its piece --
            // every shard's `done` event lies behind both (a shard launches only after all its pulls)
            if (sp->executed) for (int g = 0; g < G; ++g) if (sp->sh[g].plan) SHIP(hipStreamWaitEvent(ps, sp->sh[g].done, 0));
            if (e > b) SHIP(hipMemcpyPeerAsync((char *)dst.x + b, dst.device, (const char *)x0 + b, root_dev, e - b, ps));
            SHIP(hipEventRecord(dst.piece_ready, ps));
        }
        for (int u = 1; u < U; ++u) {                    // all-gather
            Shard &dst = sp->sh[sp->holder[u]];
            SHIP(hipSetDevice(dst.device));
            for (int v = 1; v < U; ++v) {
                if (v == u) continue;
                Shard &src = sp->sh[sp->holder[v]];
                const size_t b = cut((size_t)v - 1), e = cut((size_t)v);
                hipStream_t ps = dst.pull_stream[v];
                SHIP(hipStreamWaitEvent(ps, src.piece_ready, 0));
                SHIP(hipStreamWaitEvent(ps, dst.piece_ready, 0));     // (orders this pull behind the scatter's wait for the previous frame)
                if (e > b) SHIP(hipMemcpyPeerAsync((char *)dst.x + b, dst.device, (const char *)src.x + b, src.device, e - b, ps));
                SHIP(hipEventRecord(dst.pull_done[v], ps));
            }
            SHIP(hipStreamWaitEvent(dst.stream, dst.piece_ready, 0));
            for (int v = 1; v < U; ++v) if (v != u) SHIP(hipStreamWaitEvent(dst.stream, dst.pull_done[v], 0));
            SHIP(hipEventRecord(dst.x_ready, dst.stream));
        }
    }
    std::vector<const void *> xs(G, nullptr);
    for (int g = 0; g < G; ++g) {
        const int u = sp->uidx[g];
        xs[g] = xr[u];
        if (g != sp->holder[u]) {                        // a later shard on the same device: its stream waits for the device's replica
            SHIP(hipSetDevice(sp->sh[g].device));
            SHIP(hipStreamWaitEvent(sp->sh[g].stream, sp->sh[sp->holder[u]].x_ready, 0));
        }
    }
    // ---- beamform the slabs and gather them into y (on the root device)
    void *ydst = host ? sp->y_root : y;
    for (int g = 0; g < G; ++g) {
        Shard &s = sp->sh[g];
        if (!s.plan) continue;
        SHIP(hipSetDevice(s.device));
        int rc = qdas_plan_execute(s.plan, xs[g], s.y, (void *)s.stream);
        if (rc) return rc;
        // slab planes (i_count x [N] x [M]) -> their rows of the I x [N] x [M] image
        const size_t planes = (size_t)sp->oN * sp->oM;
        if (sp->mirror) {                                // [slab A | slab B] -> pixels [i_begin, +i_count) and their mirror images [I - i_begin - i_count, +i_count)
            const size_t nb = (size_t)s.i_count * sp->ds;
            char *dA = (char *)ydst + s.i_begin * sp->ds, *dB = (char *)ydst + (sp->I - s.i_begin - s.i_count) * sp->ds;
            if (s.device == root_dev) {
                SHIP(hipMemcpyAsync(dA, s.y, nb, hipMemcpyDeviceToDevice, s.stream));
                SHIP(hipMemcpyAsync(dB, (const char *)s.y + nb, nb, hipMemcpyDeviceToDevice, s.stream));
            } else {
                SHIP(hipMemcpyPeerAsync(dA, root_dev, s.y, s.device, nb, s.stream));
                SHIP(hipMemcpyPeerAsync(dB, root_dev, (const char *)s.y + nb, s.device, nb, s.stream));
            }
        } else if (s.device == root_dev)
            SHIP(hipMemcpy2DAsync((char *)ydst + s.i_begin * sp->ds, (size_t)sp->I * sp->ds, s.y, (size_t)s.i_count * sp->ds,
                                  (size_t)s.i_count * sp->ds, planes, hipMemcpyDeviceToDevice, s.stream));
        else
            for (size_t p = 0; p < planes; ++p)
                SHIP(hipMemcpyPeerAsync((char *)ydst + ((size_t)p * sp->I + s.i_begin) * sp->ds, root_dev,
                                        (const char *)s.y + p * (size_t)s.i_count * sp->ds, s.device, (size_t)s.i_count * sp->ds, s.stream));
        SHIP(hipEventRecord(s.done, s.stream));
    }
    // ---- join: the caller's stream (device callers) or the host (host callers) waits for every slab
    SHIP(hipSetDevice(root_dev));
    if (host) {
        for (int g = 1; g < G; ++g) if (sp->sh[g].plan) SHIP(hipStreamWaitEvent(r0.stream, sp->sh[g].done, 0));
        SHIP(hipMemcpyAsync(y, sp->y_root, ybytes, hipMemcpyDeviceToHost, r0.stream));
        SHIP(hipStreamSynchronize(r0.stream));
    } else {
        for (int g = 0; g < G; ++g) if (sp->sh[g].plan) SHIP(hipStreamWaitEvent((hipStream_t)stream, sp->sh[g].done, 0));
    }
    sp->executed = true;
    return QDAS_OK;
}

extern "C" int qdas_plan_sharded_info(const qdas_sharded_plan *sp, int shard, int *device, uint64_t *i_begin, uint64_t *i_count, int *kernel) {
    if (!sp) return failf(QDAS_EINVAL, "null plan");
    if (shard < 0) { if (device) *device = (int)sp->sh.size(); return QDAS_OK; }      // shard = -1: number of shards in *device
    if (shard >= (int)sp->sh.size()) return failf(QDAS_EINVAL, "shard index out of range");
    const Shard &s = sp->sh[shard];
    if (device) *device = s.device;
    if (i_begin) *i_begin = s.i_begin;
    if (i_count) *i_count = s.i_count;
    if (kernel) *kernel = s.plan ? qdas_plan_kernel(s.plan) : 0;
    return QDAS_OK;
}

// 1: mirror slabs -- shard g beamforms [i_begin, +i_count) AND the mirror images of those columns, pixels [I - i_begin - i_count, +i_count)
extern "C" int qdas_plan_sharded_mirror(const qdas_sharded_plan *sp) { return sp && sp->mirror ? 1 : 0; }

extern "C" void qdas_plan_destroy_sharded(qdas_sharded_plan *sp) { delete sp; }
