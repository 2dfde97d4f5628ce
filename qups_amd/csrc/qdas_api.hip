// qdas_api.hip -- C ABI of libqdas.so (see include/qdas.h for the contract and the reference
// call sites each entry replaces).  Host-side only: validation, plan construction (device
// copies of geometry, folded apodization table, stride tables), kernel selection and launch.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/qdas.h"
#include "qdas_kernels.h"
#include "jit.h"

using namespace qdas;

// ------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(call)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) return fail(QDAS_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

extern "C" const char *qdas_last_error(void) { return g_err.c_str(); }
void qdas_internal_set_error(const char *msg) { g_err = msg ? msg : ""; }      // for the library's other translation units (sharded.hip)
extern "C" int qdas_version(void) { return QDAS_VERSION; }

extern "C" int qdas_device_info(int device, char *name, size_t name_len, int *cu_count, int *clock_khz,
                                uint64_t *hbm_bytes) {
    if (device < 0) HIPCHK(hipGetDevice(&device));
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, device));
    if (name && name_len) { snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName); }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (clock_khz) *clock_khz = p.clockRate;
    if (hbm_bytes) *hbm_bytes = (uint64_t)p.totalGlobalMem;
    return QDAS_OK;
}

// Every entry that works on a particular device switches to it for the duration of the call only: the calling thread's current
// device is restored on every return path (a MEX gateway or a plain C caller keeps issuing its own work where it was).
struct DeviceGuard {
    int prev = -1;
    bool restore = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        if (dev < 0) return;
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) { err = hipSetDevice(dev); restore = err == hipSuccess; }
    }
    ~DeviceGuard() { if (restore) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// ------------------------------------------------------------------------------------ plan
static size_t real_size(int dtype) { return dtype == QDAS_F64 ? 8 : 4; }            // geometry / time type
static size_t data_size(int dtype) { return dtype == QDAS_F64 ? 16 : (dtype == QDAS_F32 ? 8 : 4); }  // complex sample
static size_t apod_real_size(int dtype) { return dtype == QDAS_F64 ? 8 : (dtype == QDAS_F32 ? 4 : 2); }

struct qdas_plan {
    qdas_desc d{};
    uint64_t I = 0, i_count = 0, y_ld = 0, oN = 1, oM = 1;
    double cinv0 = 0.0;                       // first entry of the sound-speed array (what `delays` uses, kern/das_spec.m:377)
    int device = 0;
    int kernel = QDAS_KERNEL_GENERIC;
    std::vector<void *> owned;                // device allocations made by the plan
    GenericParams gp{};
    TileParams tp{};
    TileConfig tc{};
    unsigned ntiles = 0, tile_cols = 0;
    bool no_fallback = false;                 // the probe found no tile whose delay spread exceeds the LDS window
    double misfit_frac = 0.0;                 // fraction of the chosen footprint's tiles that do not fit it
    bool wtab_real = false;                   // the folded weight table has no imaginary part
    // reciprocity fold (fold.hip; tile_params.h `fold`): the plan's folded copy of a frame (T x N x M complex64, upper triangle written per frame)
    // and the N x M weight table the fold pass applies (null: ones) -- the folded kernels themselves carry no table
    void *fold_buf = nullptr;
    const void *fold_wtab = nullptr;
    // tolerance mode (QDAS_PLAN_APPROX_SYMMETRY): bounds [samples] of the delay error the symmetry modes in use commit; 0 = exact symmetry, -1 = mode not in use
    double mirror_bound = -1.0, recip_bound = -1.0;
    bool prefolded = false;                   // QDAS_PLAN_PREFOLDED: execute is handed the folded frame (qdas_fold) -- no fold pass, no fold buffer
    // fp16 reciprocal data: the frame is folded into a complex64 copy (fold.hip, fp16 in) and beamformed by an fp32 PREFOLDED child plan into a complex64
    // image, which is rounded to the plan's complex32 output: the folded fp32 kernels serve fp16 data (C3 with fp16 data 23.7 -> 15 ms)
    qdas_plan *f16_child = nullptr;
    void *y32 = nullptr;
    bool fb2_ok = false;                      // frames of a sequence may share launches, 4 or 2 at a time (decided at plan creation)
    bool fb4_off = false;                     // ... but at most pairwise (QDAS_NO_FB4)
    bool fold2_ok = false;                    // folded data: two frames may share a launch
    void *fold_buf2 = nullptr;                // ... and the folded copy of the second one (made at the first stream of frames)
    // A lateral-mirror plan runs one frame per launch (its second window set is taken).  For a STREAM of frames four frames per launch share
    // more (index + weights of four traces instead of two): such plans keep a twin without the mirror mode, made at the first stream.
    qdas_plan *frames_twin = nullptr;
    bool twin_tried = false;
    uint32_t *fallback = nullptr;             // device: [0] = count, [1..ntiles]
    size_t jit_lds = 0;                       // dynamic LDS of the specialised kernel
    int jit_mb = 0, jit_w = 0;                // its transmits per stage, samples per window
    int hint_mb = 0, hint_w = 0;              // stage shape a plan-specialised build should take instead of the prebuilt configuration's (plan_stage_shape)
    hipFunction_t jit_fn = nullptr;           // plan-specialised kernel (QDAS_PLAN_JIT, jit.hip); null: prebuilt instantiation
    std::string jit_tag;                      // "jit <hash>" when the plan runs a hiprtc-specialised kernel (QDAS_PLAN_JIT)
    bool prep2 = false, prep4 = false;        // the two- / four-frame instantiations have been resolved (qdas_plan_execute_frames)
    bool timing = false;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float last_ms = 0.f;
    // host staging (mem == HOST)
    void *dx = nullptr, *dy = nullptr;
    size_t x_bytes = 0, y_bytes = 0;
    // second staging set + copy stream: frame f+1 is uploaded while frame f is beamformed (host-resident frame sequences)
    void *dx2 = nullptr, *dy2 = nullptr;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ex[2] = {nullptr, nullptr}, ek[2] = {nullptr, nullptr};

    ~qdas_plan() {
        if (frames_twin) qdas_plan_destroy(frames_twin);
        if (f16_child) qdas_plan_destroy(f16_child);
        for (void *p : owned) (void)hipFree(p);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        for (int k = 0; k < 2; ++k) { if (ex[k]) (void)hipEventDestroy(ex[k]); if (ek[k]) (void)hipEventDestroy(ek[k]); }
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
    }
};

static int dev_alloc(qdas_plan *pl, void **out, size_t bytes) {
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
    if (e != hipSuccess) return fail(QDAS_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    pl->owned.push_back(p);
    *out = p;
    return QDAS_OK;
}

// Product of the pixel-dependent apodization arrays of a plan, broadcast to one contiguous I1*I2*I3 x E array (E = N, M or 1) in the
// arrays' own element type: what the fused kernel indexes as [pixel + I * stage element].  (The reference multiplies the S arrays per
// pair in the data precision, src/bf.cu:118-120; here the pixel-dependent factors are multiplied once per plan, in fp32, rounded once.)
struct ApodFold {
    const void *base;
    uint64_t st[QDAS_MAX_APOD][5];      // per array: strides over I1, I2, I3, the element; offset (in elements of the array type)
    uint64_t I1, I2, I3, E;
    int ns;
};
template <class T, bool CPLX> __global__ void apod_fold_kernel(ApodFold f, T *out, uint64_t nel) {
    const uint64_t I = f.I1 * f.I2 * f.I3;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nel; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = q % I, e = q / I;
        const uint64_t i1 = i % f.I1, i2 = (i / f.I1) % f.I2, i3 = i / (f.I1 * f.I2);
        float wr = 1.f, wi = 0.f;
        for (int s = 0; s < f.ns; ++s) {
            const uint64_t k = f.st[s][4] + i1 * f.st[s][0] + i2 * f.st[s][1] + i3 * f.st[s][2] + e * f.st[s][3];
            if constexpr (CPLX) {
                const float ar = (float)((const T *)f.base)[2 * k], ai = (float)((const T *)f.base)[2 * k + 1];
                const float nr = wr * ar - wi * ai, ni = wr * ai + wi * ar;
                wr = nr; wi = ni;
            } else wr *= (float)((const T *)f.base)[k];
        }
        if constexpr (CPLX) { out[2 * q] = (T)wr; out[2 * q + 1] = (T)wi; }
        else out[q] = (T)wr;
    }
}

// Host -> device, synchronous, for the library's own (mostly small) uploads: staged in pinned memory and written by a KERNEL -- whatever sits between a
// copy-engine (or CPU) write into freshly allocated memory and the kernel launched behind it (see csrc/scratch.hip for what was seen on this platform), a
// write issued by the shader engines goes through the same translation and caches as the reads that follow it.
__global__ void __launch_bounds__(256) upload_kernel(unsigned char *__restrict__ dst, const unsigned char *__restrict__ src, size_t bytes) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, n16 = bytes / 16;
    if (((uintptr_t)dst & 15u) == 0) {
        if (i < n16) ((uint4 *)dst)[i] = ((const uint4 *)src)[i];
        if (i < bytes - 16 * n16) dst[16 * n16 + i] = src[16 * n16 + i];
    } else {
        for (size_t b = 16 * i; b < 16 * i + 16 && b < bytes; ++b) dst[b] = src[b];
    }
}
extern "C" int qdas_internal_upload(void *dst, const void *src, size_t bytes) {
    constexpr size_t CAP = 1u << 20;
    if (!bytes) return (int)hipSuccess;
    // (larger uploads go through the same pinned buffer, one MiB at a time -- ADVICE r5: they fell back to a plain hipMemcpy, the call the stale read of round 5
    //  involved; root cause unknown, tools/repro keeps reproducing it on fresh boxes -- so no upload of the library takes that path any more)
    static std::mutex mu;
    static void *pin = nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!pin && hipHostMalloc(&pin, CAP, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { pin = nullptr; (void)hipGetLastError(); return (int)hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice); }
    hipError_t e = hipSuccess;
    for (size_t at = 0; at < bytes && e == hipSuccess; at += CAP) {
        const size_t nb = bytes - at < CAP ? bytes - at : CAP;
        memcpy(pin, (const unsigned char *)src + at, nb);
        upload_kernel<<<(unsigned)((nb + 4095) / 4096), 256, 0, nullptr>>>((unsigned char *)dst + at, (const unsigned char *)pin, nb);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);      // (the pinned buffer is rewritten by the next piece)
    }
    return (int)e;
}
static inline hipError_t upload(void *dst, const void *src, size_t bytes) { return (hipError_t)qdas_internal_upload(dst, src, bytes); }

// device copy of a caller array (host -> new device buffer; device -> used in place, or -- QDAS_PLAN_COPY_INPUTS -- copied
// into a plan-owned buffer so that the plan outlives the caller's arrays)
static int import_array(qdas_plan *pl, const void *src, size_t bytes, int mem, const void **out) {
    const bool own = (pl->d.plan_flags & QDAS_PLAN_COPY_INPUTS) != 0;
    if ((mem == QDAS_MEM_DEVICE && !own) || bytes == 0 || !src) { *out = src; return QDAS_OK; }
    void *p;
    int rc = dev_alloc(pl, &p, bytes);
    if (rc) return rc;
    if (mem == QDAS_MEM_DEVICE) HIPCHK(hipMemcpy(p, src, bytes, hipMemcpyDeviceToDevice));
    else HIPCHK(upload(p, src, bytes));
    *out = p;
    return QDAS_OK;
}

// host copy of (part of) a caller array
static int fetch_host(const void *src, size_t bytes, int mem, void *dst) {
    if (mem == QDAS_MEM_DEVICE) HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    else memcpy(dst, src, bytes);
    return QDAS_OK;
}

// Largest pixel pitch [m] along I1 (3 sampled columns) and across columns (3 sampled rows) of the scan: the inputs of the
// tiled kernel's shape model (choose_tile_shape below).
static int scan_pitch(const qdas_desc *desc, double *gz_out, double *gc_out) {
    const qdas_sizes &z = desc->sz;
    const uint64_t ncols = z.I2 * z.I3;
    const size_t rs = z.dtype == QDAS_F64 ? 8 : 4, ps = 3 * rs;          // bytes per coordinate / per pixel position (real(prec))
    const unsigned char *Pi = (const unsigned char *)desc->Pi;
    double gz = 0.0, gc = 0.0;
    std::vector<unsigned char> buf;
    auto dist = [&](uint64_t i, uint64_t j) {                             // distance between pixels i and j of buf
        double d[3];
        for (int k = 0; k < 3; ++k)
            d[k] = rs == 8 ? ((const double *)buf.data())[3 * i + k] - ((const double *)buf.data())[3 * j + k]
                           : (double)((const float *)buf.data())[3 * i + k] - (double)((const float *)buf.data())[3 * j + k];
        return std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    };
    if (z.I1 > 1) {
        buf.resize(ps * z.I1);
        const uint64_t cs[3] = {0, ncols / 2, ncols - 1};
        for (int k = 0; k < 3; ++k) {
            int rc = fetch_host(Pi + ps * z.I1 * cs[k], ps * z.I1, desc->mem, buf.data());
            if (rc) return rc;
            for (uint64_t i = 0; i + 1 < z.I1; ++i) { const double d = dist(i, i + 1); if (d > gz) gz = d; }
        }
    }
    if (z.I2 > 1) {
        buf.resize(ps * ncols);
        const uint64_t rws[3] = {0, z.I1 / 2, z.I1 - 1};
        for (int k = 0; k < 3; ++k) {
            if (desc->mem == QDAS_MEM_DEVICE)
                HIPCHK(hipMemcpy2D(buf.data(), ps, Pi + ps * rws[k], ps * z.I1, ps, ncols, hipMemcpyDeviceToHost));
            else
                for (uint64_t c = 0; c < ncols; ++c) memcpy(&buf[ps * c], Pi + ps * (rws[k] + z.I1 * c), ps);
            for (uint64_t c = 0; c + 1 < ncols; ++c) {
                if ((c + 1) % z.I2 == 0) continue;         // slice boundary of a 3-D scan
                const double d = dist(c, c + 1);
                if (d > gc) gc = d;
            }
        }
    }
    *gz_out = (gz == gz) ? gz : 0.0;                     // NaN pixels: no information
    *gc_out = (gc == gc) ? gc : 0.0;
    return QDAS_OK;
}

static float half_to_float(uint16_t h) {
    const uint32_t s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023;
    float v;
    if (e == 0) v = std::ldexp((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = std::ldexp((float)(m + 1024), (int)e - 25);
    return s ? -v : v;
}

static int validate(const qdas_desc *d) {
    const qdas_sizes &z = d->sz;
    if (z.dtype < QDAS_F64 || z.dtype > QDAS_F16) return fail(QDAS_EINVAL, "Unrecognized input precision %d", z.dtype);
    const int interp = z.flag & QDAS_FLAG_INTERP_MASK;
    if (interp > 5) return fail(QDAS_EINVAL, "QUPS:das_spec:UnrecognizedInput: Unrecognized interpolation flag %d: must be one of "
                                "{'nearest', 'linear', 'cubic', 'lanczos3'}.", interp);
    if (z.flag & ~63) return fail(QDAS_EINVAL, "Invalid beamformer flag 0x%x.", z.flag);
    if (z.S > QDAS_MAX_APOD) return fail(QDAS_EUNSUPPORTED, "at most %d apodization arrays are supported (got %llu)", QDAS_MAX_APOD,
                                         (unsigned long long)z.S);
    if (!(d->fs > 0.0) || !std::isfinite(d->fs)) return fail(QDAS_EINVAL, "Undefined sampling rate.");
    if (!std::isfinite(d->fmod)) return fail(QDAS_EINVAL, "modulation frequency must be finite");
    const uint64_t I = z.I1 * z.I2 * z.I3;
    if (d->i_begin > I || (d->i_count && d->i_begin + d->i_count > I))
        return fail(QDAS_EINVAL, "pixel shard [%llu, +%llu) exceeds the image (%llu pixels)", (unsigned long long)d->i_begin,
                    (unsigned long long)d->i_count, (unsigned long long)I);
    if (I && z.N && z.M) {
        if (!d->Pi || !d->Pr || !d->Pv || !d->Nv || !d->cinv || !d->acstride) return fail(QDAS_EINVAL, "null geometry / stride pointer");
        if (z.S && !d->apod) return fail(QDAS_EINVAL, "S > 0 but apod is null");
        if (d->rx_apod_kind < QDAS_RXAPOD_NONE || d->rx_apod_kind > QDAS_RXAPOD_FNUMBER_ORIENTED)
            return fail(QDAS_EINVAL, "Unrecognized generated receive apodization %d", d->rx_apod_kind);
        if ((d->rx_apod_kind == QDAS_RXAPOD_ACCEPTANCE || d->rx_apod_kind == QDAS_RXAPOD_COSINE || d->rx_apod_kind == QDAS_RXAPOD_FNUMBER_ORIENTED)
            && !d->rx_normals) return fail(QDAS_EINVAL, "generated receive apodization needs the element normals (rx_normals)");
        if (d->rx_apod_kind && !(d->rx_apod_p[0] == d->rx_apod_p[0])) return fail(QDAS_EINVAL, "generated receive apodization: NaN parameter");
    }
    if (d->mem != QDAS_MEM_HOST && d->mem != QDAS_MEM_DEVICE) return fail(QDAS_EINVAL, "bad mem kind %d", d->mem);
    if (z.T > 0x7fffffffull) return fail(QDAS_EUNSUPPORTED, "T must be < 2^31");
    return QDAS_OK;
}

// size (elements) of a broadcastable I1xI2xI3xNxM array from its stride row
static uint64_t bcast_numel(const uint64_t *st, const qdas_sizes &z) {
    const uint64_t dims[5] = {z.I1, z.I2, z.I3, z.N, z.M};
    uint64_t n = 1;
    for (int k = 0; k < 5; ++k) if (st[k]) n += (dims[k] - 1) * st[k];
    return n;
}

// Shape of the tiled kernel's tiles and waves (das_tile_impl.h).
//  * tile footprint (64x16, 32x32, 16x64 or 8x128 pixels of I1 x columns): every candidate is PROBED -- the kernel's own
//    prologue runs for all tiles and counts those whose delay spread does not fit the LDS window; only footprints with the
//    fewest misfits are considered (a misfit tile is redone by the generic kernel at >10x the cost).
//  * wave footprint inside the tile (2^w x 2^(6-w) pixels): an LDS access group is 32 lanes; it is conflict-free when the
//    lanes read <= 32 consecutive samples, i.e. (group depth) x (delay samples per pixel of depth) + (group columns) x
//    (lateral gradient) <= 32.  The shallowest-needed wave wins; ties go to the deeper wave (longer store runs).
// QDAS_TILE_Z / QDAS_WAVE_Z override the choice (profiling, tests).
template <typename F>
static int choose_tile_shape(qdas_plan *pl, const qdas_desc *desc, F &&set_grid) {
    TileParams &t = pl->tp;
    const qdas_sizes &z = desc->sz;
    auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    int env_tz = 0, env_wz = 0;
    if (const char *e = getenv("QDAS_TILE_Z")) { const int v = atoi(e); if (v == 8 || v == 16 || v == 32 || v == 64) env_tz = v; }
    if (const char *e = getenv("QDAS_WAVE_Z")) { const int v = atoi(e); if (v == 4 || v == 8 || v == 16 || v == 32 || v == 64) env_wz = v; }
    double gz = 0.0, gc = 0.0;
    int rc = scan_pitch(desc, &gz, &gc);
    if (rc) return rc;
    const double per = 2.0 * std::fabs(t.cinv_fs);      // delay samples per metre of pixel pitch (both legs)
    const double spz = per * gz, spc = 0.35 * per * gc;  // per pixel of depth / per column (obliquity-weighted)
    // misfit count of each footprint
    double fbn[7] = {0, 0, 0, 0, 0, 0, 0}, best_fb = 2.0;   // misfit FRACTION of the footprint's tiles
    for (int l = 6; l >= 3; --l) {
        if (env_tz && (1 << l) != env_tz) continue;
        set_grid(l);
        TileParams p = t;
        p.probe = 1; p.wz_log2 = l; p.x = nullptr; p.y = nullptr; p.wtab = nullptr; p.apix = nullptr;
        HIPCHK(hipMemset(pl->fallback, 0, sizeof(uint32_t)));
        HIPCHK(launch_tile(p, z.dtype, pl->ntiles, nullptr));
        uint32_t cnt = 0;
        HIPCHK(hipMemcpy(&cnt, pl->fallback, sizeof(uint32_t), hipMemcpyDeviceToHost));
        fbn[l] = (double)cnt / (double)(pl->ntiles ? pl->ntiles : 1);
        if (fbn[l] < best_fb) best_fb = fbn[l];
    }
    int best_t = -1, best_w = -1;
    double best_c = 1e300;
    for (int l = 6; l >= 3; --l) {
        if (env_tz && (1 << l) != env_tz) continue;
        if (fbn[l] > best_fb + 1e-9) continue;
        for (int w = l; w >= 2 && w >= l - 4; --w) {
            if (env_wz && (1 << w) != env_wz && !(env_wz > (1 << l) && w == l)) continue;
            const int gd = w >= 5 ? 32 : (1 << w), gcn = 32 / gd;     // depth x columns of one 32-lane access group
            const double span = gd * spz + gcn * spc;
            const double c = span <= 32.0 ? 1.0 : span / 32.0;
            if (c < best_c - 1e-9) { best_c = c; best_t = l; best_w = w; }
        }
    }
    if (best_t < 0) { best_t = env_tz ? lg(env_tz) : 6; best_w = best_t; }
    // the window fit depends on the geometry only: a footprint without misfits never needs the per-frame fallback pass
    pl->no_fallback = fbn[best_t] == 0.0;
    pl->misfit_frac = fbn[best_t];
    HIPCHK(hipMemset(pl->fallback, 0, sizeof(uint32_t)));
    set_grid(best_t);
    t.wz_log2 = best_w;
    t.probe = 0;
    return QDAS_OK;
}

// Lateral-mirror symmetry (tile_params.h `mir`): is pixel column I2-1-c the mirror image (x -> -x) of column c, bit for bit?  `dev`: the largest
// distance |mirror(p') - p| over the pixels, as float bits (non-negative floats order like their bit patterns): what the tolerance mode bounds
__global__ void mirror_check_kernel(const float *Pi, uint64_t I1, uint64_t I2, uint32_t *bad, uint32_t *dev) {
    const uint64_t half = (I2 + 1) / 2, n = I1 * half;
    float worst = 0.f;
    bool exact = true;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i1 = q % I1, c = q / I1;
        const float *a = Pi + 3 * (i1 + I1 * c), *b = Pi + 3 * (i1 + I1 * (I2 - 1 - c));
        if (!(a[0] == -b[0] && a[1] == b[1] && a[2] == b[2])) {      // (NaN coordinates: not symmetric, in either mode)
            exact = false;
            const float dx = a[0] + b[0], dy = a[1] - b[1], dz = a[2] - b[2];
            const float d = sqrtf(dx * dx + dy * dy + dz * dz);
            worst = (d == d) ? fmaxf(worst, d) : INFINITY;
        }
    }
    if (!exact) { *bad = 1u; atomicMax(dev, __float_as_uint(worst)); }
}
// a per-pixel map (I1 x I2, contiguous): c[i1, I2-1-col] == c[i1, col] bit for bit?
__global__ void mirror_map_check_kernel(const uint32_t *c, uint64_t I1, uint64_t I2, uint32_t *bad) {
    const uint64_t n = I1 * (I2 / 2);
    bool same = true;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i1 = q % I1, col = q / I1;
        if (c[i1 + I1 * col] != c[i1 + I1 * (I2 - 1 - col)]) same = false;
    }
    if (!same) *bad = 1u;
}
static int mirror_symmetric_map(const float *dmap, uint64_t I1, uint64_t I2, bool *yes) {
    *yes = false;
    uint32_t *flag = nullptr, res = 1;
    HIPCHK(hipMalloc((void **)&flag, sizeof(uint32_t)));
    hipError_t e = hipMemset(flag, 0, sizeof(uint32_t));
    if (e == hipSuccess) {
        const uint64_t n = I1 * (I2 / 2);
        mirror_map_check_kernel<<<(unsigned)std::min<uint64_t>((n + 255) / 256 + 1, 4096), 256, 0, 0>>>((const uint32_t *)dmap, I1, I2, flag);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(&res, flag, sizeof(uint32_t), hipMemcpyDeviceToHost);
    }
    (void)hipFree(flag);
    HIPCHK(e);
    *yes = res == 0;
    return QDAS_OK;
}
// receivers, transmits (positions, normals, t0) and pixel columns mirror-symmetric about x = 0?  fp32 geometry.
// Exact mode (`tol` < 0): bit for bit.  Tolerance mode (QDAS_PLAN_APPROX_SYMMETRY, `tol` >= 0 in SAMPLES): positions may deviate -- distance is
// 1-Lipschitz in either end point, so |tau(p', N-1-n, M-1-m) - tau(p, n, m)| * fs <= cinv * fs * (2 max|M(p') - p| + max|M(r') - r| + max|M(v') - v|) =: bound,
// and the mode is taken when bound <= tol; normals and start times (which enter the plane-wave delay and the focused-wave sign) must still match exactly.
static int mirror_symmetric(const qdas_desc *desc, const float *dPi, bool *yes, double tol = -1.0, double cinv_fs = 0.0, double *bound = nullptr) {
    const qdas_sizes &z = desc->sz;
    *yes = false;
    if (bound) *bound = 0.0;
    std::vector<float> hr(3 * z.N), hv(4 * z.M), hn(3 * z.M);
    int rc;
    if ((rc = fetch_host(desc->Pr, hr.size() * 4, desc->mem, hr.data()))) return rc;
    if ((rc = fetch_host(desc->Pv, hv.size() * 4, desc->mem, hv.data()))) return rc;
    if ((rc = fetch_host(desc->Nv, hn.size() * 4, desc->mem, hn.data()))) return rc;
    double dr = 0.0, dv = 0.0;                          // largest deviation of a receiver / transmit position from its mirror partner's image
    auto dist3 = [](const float *a, const float *b) { const double dx = (double)a[0] + (double)b[0], dy = (double)a[1] - (double)b[1], dz = (double)a[2] - (double)b[2]; return sqrt(dx * dx + dy * dy + dz * dz); };
    for (uint64_t n = 0; n < z.N; ++n) {
        const float *a = &hr[3 * n], *b = &hr[3 * (z.N - 1 - n)];
        if (!(a[0] == -b[0] && a[1] == b[1] && a[2] == b[2])) {
            if (tol < 0) return QDAS_OK;
            const double d = dist3(a, b);
            if (!(d == d)) return QDAS_OK;
            dr = std::max(dr, d);
        }
    }
    for (uint64_t m = 0; m < z.M; ++m) {
        const float *a = &hv[4 * m], *b = &hv[4 * (z.M - 1 - m)], *c = &hn[3 * m], *d = &hn[3 * (z.M - 1 - m)];
        if (!(a[3] == b[3] && c[0] == -d[0] && c[1] == d[1] && c[2] == d[2])) return QDAS_OK;       // start times and normals: exact in both modes
        if (!(a[0] == -b[0] && a[1] == b[1] && a[2] == b[2])) {
            if (tol < 0) return QDAS_OK;
            const double e = dist3(a, b);
            if (!(e == e)) return QDAS_OK;
            dv = std::max(dv, e);
        }
    }
    if (desc->rx_apod_kind && desc->rx_normals) {       // a generated receive apodization: the element normals as well
        std::vector<float> hx(3 * z.N);
        if ((rc = fetch_host(desc->rx_normals, hx.size() * 4, desc->mem, hx.data()))) return rc;
        for (uint64_t n = 0; n < z.N; ++n) {
            const float *a = &hx[3 * n], *b = &hx[3 * (z.N - 1 - n)];
            if (!(a[0] == -b[0] && a[1] == b[1] && a[2] == b[2])) return QDAS_OK;
        }
    }
    uint32_t *flag = nullptr;
    HIPCHK(hipMalloc(&flag, 2 * sizeof(uint32_t)));
    hipError_t e = hipMemset(flag, 0, 2 * sizeof(uint32_t));
    uint32_t res[2] = {1u, 0x7f800000u};
    if (e == hipSuccess) {
        const uint64_t n = z.I1 * ((z.I2 + 1) / 2);
        mirror_check_kernel<<<(unsigned)std::min<uint64_t>((n + 255) / 256, 4096), 256, 0, 0>>>(dPi, z.I1, z.I2, flag, flag + 1);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(res, flag, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost);
    }
    (void)hipFree(flag);
    HIPCHK(e);
    float dpf;
    memcpy(&dpf, &res[1], 4);
    const double dp = res[0] ? (double)dpf : 0.0;
    if (tol < 0) { *yes = res[0] == 0; return QDAS_OK; }
    const double b = cinv_fs * (2.0 * dp + dr + dv);
    if (bound) *bound = b;
    *yes = b == b && b <= tol;
    return QDAS_OK;
}

// ------------------------------------------------------------------------------------ plan creation
// qdas_plan_create in steps.  WHAT a plan runs -- kernel, launch configuration, symmetry modes, and why not -- is decided by the pure functions of
// plan_modes.h (testable without a GPU: tests/modes/enumerate_modes.cpp); the steps below gather the facts those functions ask for (host comparisons of
// the geometry, the device's mirror check, allocations, window-fit probes) and carry out their decisions (uploads, tables, kernel builds).
static modes::Switches read_switches() {
    modes::Switches sw;
    auto on = [](const char *n) { return getenv(n) != nullptr; };
    sw.no_sym = on("QDAS_NO_SYM"); sw.no_fold = on("QDAS_NO_FOLD"); sw.no_jit = on("QDAS_NO_JIT"); sw.no_mirror = on("QDAS_NO_MIRROR"); sw.no_mirq = on("QDAS_NO_MIRQ");
    sw.no_narrow = on("QDAS_NO_NARROW"); sw.no_role_swap = on("QDAS_NO_ROLE_SWAP"); sw.no_bpix = on("QDAS_NO_BPIX"); sw.no_w64 = on("QDAS_NO_W64");
    sw.no_mirror_wpix = on("QDAS_NO_MIRROR_WPIX"); sw.no_mirror_wpix32 = on("QDAS_NO_MIRROR_WPIX32"); sw.no_side_split = on("QDAS_NO_SIDE_SPLIT"); sw.no_wide = on("QDAS_NO_WIDE");
    sw.no_fb2 = on("QDAS_NO_FB2"); sw.no_fb4 = on("QDAS_NO_FB4"); sw.no_fold16 = on("QDAS_NO_FOLD16");
    if (const char *e = getenv("QDAS_SYM_TOL")) { const double v = atof(e); if (v >= 0.0 && v <= 0.5) sw.sym_tol = v; }
    if (const char *e = getenv("QDAS_KSPLIT")) { const int v = atoi(e); if (v >= 1 && v <= 8) sw.ksplit = v; }
    if (const char *e = getenv("QDAS_KSPLIT_M")) { const int v = atoi(e); if (v >= 1 && v <= 8) sw.ksplit_m = v; }
    return sw;
}

// working state of one qdas_plan_create
struct PlanBuild {
    const qdas_desc *desc = nullptr;
    modes::Switches sw;
    modes::Request rq;
    modes::Symmetry sy;
    size_t ael = 0;                         // bytes per apodization entry
    int txkind = 0;                         // distance | signed distance | plane wave
    std::vector<float> host_tab;            // the folded N x M table as uploaded ([stage element + stages * block element])
    uint64_t kN_eff = 0;                    // stage elements after a side split
    modes::BuildOutcome outcome;            // what the build steps did (plan_modes.h derive_launch_shape: the model the finished plan is checked against)
};

// sizes, slab, output planes; QDAS_PLAN_MIRROR_SLAB validation
static int plan_init(qdas_plan *pl, const qdas_desc *desc) {
    pl->d = *desc;
    const qdas_sizes &z = pl->d.sz;
    pl->I = z.I1 * z.I2 * z.I3;
    pl->i_count = desc->i_count ? desc->i_count : pl->I - desc->i_begin;
    pl->y_ld = desc->y_ld ? desc->y_ld : pl->i_count;
    pl->oN = (z.flag & QDAS_FLAG_KEEP_RX) ? z.N : 1;
    pl->oM = (z.flag & QDAS_FLAG_KEEP_TX) ? z.M : 1;
    if (pl->y_ld < pl->i_count) return fail(QDAS_EINVAL, "y_ld smaller than the pixel count");
    if (desc->plan_flags & QDAS_PLAN_MIRROR_SLAB) {
        const bool ok = z.I3 == 1 && z.I2 % 2 == 0 && z.I1 && desc->i_begin % z.I1 == 0 && pl->i_count % z.I1 == 0 && pl->i_count
                        && desc->i_begin + pl->i_count <= pl->I / 2 && !(z.flag & (QDAS_FLAG_KEEP_RX | QDAS_FLAG_KEEP_TX)) && desc->kernel != QDAS_KERNEL_GENERIC;
        if (!ok) return fail(QDAS_EINVAL, "QDAS_PLAN_MIRROR_SLAB: the slab must be whole columns of the first half of an even number of columns (I3 == 1, 'DAS')");
        if (!desc->y_ld) pl->y_ld = 2 * pl->i_count;     // y holds slab A and its mirror image
        else if (desc->y_ld < 2 * pl->i_count) return fail(QDAS_EINVAL, "QDAS_PLAN_MIRROR_SLAB: y_ld smaller than 2 * i_count (y holds slab A and its mirror image)");
    }
    return QDAS_OK;
}

// stride tables (host pointer, reference kern/das_spec.m:257-260) + device copies of the constant inputs -> the generic kernel's parameter block
static int plan_import_inputs(qdas_plan *pl, const qdas_desc *desc, PlanBuild &b) {
    const qdas_sizes &z = pl->d.sz;
    const int dt = z.dtype;
    GenericParams &g = pl->gp;
    int rc;
    memcpy(g.cst, desc->acstride, sizeof g.cst);
    memset(g.ast, 0, sizeof g.ast);
    memcpy(g.ast, desc->acstride + 6, sizeof(uint64_t) * 6 * z.S);
    const size_t rs = real_size(dt);
    uint64_t apod_elems = 0;
    for (uint64_t s = 0; s < z.S; ++s) {
        const uint64_t end = g.ast[6 * s + 5] + bcast_numel(&g.ast[6 * s], z);
        if (end > apod_elems) apod_elems = end;
    }
    b.ael = desc->apod_real ? apod_real_size(dt) : data_size(dt);
    const uint64_t cinv_elems = bcast_numel(g.cst, z);
    if ((rc = import_array(pl, desc->Pi, 3 * pl->I * rs, desc->mem, &g.Pi))) return rc;
    if ((rc = import_array(pl, desc->Pr, 3 * z.N * rs, desc->mem, &g.Pr))) return rc;
    if ((rc = import_array(pl, desc->Pv, 4 * z.M * rs, desc->mem, &g.Pv))) return rc;
    if ((rc = import_array(pl, desc->Nv, 3 * z.M * rs, desc->mem, &g.Nv))) return rc;
    if ((rc = import_array(pl, desc->cinv, cinv_elems * rs, desc->mem, &g.cinv))) return rc;
    if ((rc = import_array(pl, desc->apod, apod_elems * b.ael, desc->mem, &g.apod))) return rc;
    g.T = z.T; g.N = z.N; g.M = z.M; g.I1 = z.I1; g.I2 = z.I2; g.I3 = z.I3;
    g.i_begin = desc->i_begin; g.i_count = pl->i_count; g.y_ld = pl->y_ld;
    // the reference passes [fs, fmod] in the kernel's real type (src/bf.cu:57-58)
    g.fs = dt == QDAS_F64 ? desc->fs : (double)(float)desc->fs;
    g.fmod = dt == QDAS_F64 ? desc->fmod : (double)(float)desc->fmod;
    g.S = (int32_t)z.S; g.flag = z.flag; g.VS = z.VS; g.DV = z.DV; g.apod_real = desc->apod_real;
    g.gen_kind = desc->rx_apod_kind; g.gen_p0 = desc->rx_apod_p[0]; g.gen_p1 = desc->rx_apod_p[1]; g.rxn = nullptr;
    if (g.gen_kind && desc->rx_normals && (rc = import_array(pl, desc->rx_normals, 3 * z.N * rs, desc->mem, &g.rxn))) return rc;
    g.tile_list = nullptr; g.blocks_per_tile = 0; g.tile_cols = 0; g.tiles_z = 0;
    if (dt == QDAS_F64) { if ((rc = fetch_host(desc->cinv, sizeof(double), desc->mem, &pl->cinv0))) return rc; }
    else { float c32; if ((rc = fetch_host(desc->cinv, sizeof(float), desc->mem, &c32))) return rc; pl->cinv0 = (double)c32; }
    return QDAS_OK;
}

// FACT (plan_modes.h Facts::recip_*): do the transmit elements equal the receive elements, with one start time?  Host comparison of the fp32 geometry.
static int gather_recip_facts(const qdas_desc *desc, modes::Facts *f) {
    const qdas_sizes &z = desc->sz;
    std::vector<float> hr(3 * z.N), hv(4 * z.M);
    int rc;
    if ((rc = fetch_host(desc->Pr, hr.size() * 4, desc->mem, hr.data()))) return rc;
    if ((rc = fetch_host(desc->Pv, hv.size() * 4, desc->mem, hv.data()))) return rc;
    f->recip_known = true; f->recip_one_t0 = true; f->recip_exact = true; f->recip_finite = true; f->recip_dev = 0.0;
    for (uint64_t m = 0; m < z.M; ++m) {
        if (memcmp(&hv[4 * m + 3], &hv[3], 4) != 0) { f->recip_one_t0 = false; break; }                 // one start time: exact in both modes
        if (memcmp(&hv[4 * m], &hr[3 * m], 12) != 0) {
            f->recip_exact = false;
            const double dx = (double)hv[4 * m] - hr[3 * m], dy = (double)hv[4 * m + 1] - hr[3 * m + 1], dz = (double)hv[4 * m + 2] - hr[3 * m + 2];
            const double dd = sqrt(dx * dx + dy * dy + dz * dz);      // tolerance mode: the largest distance between a transmit element and "its" receive element
            if (!(dd == dd)) f->recip_finite = false; else f->recip_dev = std::max(f->recip_dev, dd);
        }
    }
    return QDAS_OK;
}

// steps 1 + 2 of plan_modes.h: the request, then the symmetry modes -- gathering each fact the resolver asks for
static int plan_resolve_modes(qdas_plan *pl, const qdas_desc *desc, PlanBuild &b) {
    const qdas_sizes &z = pl->d.sz;
    b.rq = modes::analyze_request(*desc, b.sw);
    modes::Facts f;
    f.cinv0 = pl->cinv0;
    int rc;
    for (;;) {
        const int need = modes::resolve_symmetry(*desc, pl->i_count, b.rq, f, b.sw, &b.sy);
        if (need == modes::NEED_NOTHING) break;
        if (need == modes::NEED_RECIP) { if ((rc = gather_recip_facts(desc, &f))) return rc; }
        else if (need == modes::NEED_FOLD_BUF) {          // the plan's folded copy of a frame: without the memory for it, the plan simply does not fold
            void *fbuf = nullptr;
            f.fold_buf_known = true;
            if (hipMalloc(&fbuf, (size_t)z.T * z.N * z.M * 8) != hipSuccess) { (void)hipGetLastError(); f.fold_buf_ok = false; }
            else { pl->owned.push_back(fbuf); pl->fold_buf = fbuf; f.fold_buf_ok = true; }
        } else if (need == modes::NEED_MIRROR) {
            bool yes = false;
            double bound = 0.0;
            if ((rc = mirror_symmetric(desc, (const float *)pl->gp.Pi, &yes, b.sy.sym_tol, pl->cinv0 * desc->fs, &bound))) return rc;
            // (a sound-speed map: exact mode only -- plan_modes.h leaves sym_tol unset for maps --, and the map itself mirror-symmetric bit for bit)
            if (yes && b.rq.cmap && (rc = mirror_symmetric_map((const float *)pl->gp.cinv + pl->gp.cst[5], z.I1, z.I2, &yes))) return rc;
            f.mirror_known = true; f.mirror_yes = yes; f.mirror_bound = bound;
        }
    }
    if (b.sy.prefolded_refused)
        return fail(QDAS_EUNSUPPORTED, "QDAS_PLAN_PREFOLDED: a folded frame can only be beamformed by a reciprocal fp32 'DAS' plan without apodization arrays "
                                       "(transmit elements == receive elements bit for bit, one t0, N == M >= 2; weights belong to qdas_fold)");
    pl->prefolded = b.sy.prefolded;
    pl->tc = b.sy.tc();
    if (desc->kernel == QDAS_KERNEL_TILED && !b.sy.eligible) return fail(QDAS_EUNSUPPORTED, "%s", b.sy.why);
    pl->kernel = (b.sy.eligible && desc->kernel != QDAS_KERNEL_GENERIC) ? QDAS_KERNEL_TILED : QDAS_KERNEL_GENERIC;
    return QDAS_OK;
}

// roles swapped: stage elements = transmits (positions + {t0, normal} records), block elements = receivers (zero normals, zero start times)
static int upload_swapped_geometry(qdas_plan *pl, const qdas_desc *desc, uint64_t E, int shift) {      // E stage elements; element e reads transmit e >> shift
    const qdas_sizes &z = pl->d.sz;
    TileParams &t = pl->tp;
    std::vector<float> hr(3 * z.N), hv(4 * z.M), hn(3 * z.M);
    int rc;
    if ((rc = fetch_host(desc->Pr, hr.size() * 4, desc->mem, hr.data()))) return rc;
    if ((rc = fetch_host(desc->Pv, hv.size() * 4, desc->mem, hv.data()))) return rc;
    if ((rc = fetch_host(desc->Nv, hn.size() * 4, desc->mem, hn.data()))) return rc;
    std::vector<float> spos(3 * E), sst(4 * E), bpos(4 * z.N), bnrm(3 * z.N, 0.f);
    for (uint64_t e = 0; e < E; ++e) {
        const uint64_t m = e >> shift;
        for (int k = 0; k < 3; ++k) { spos[3 * e + k] = hv[4 * m + k]; sst[4 * e + 1 + k] = hn[3 * m + k]; }
        sst[4 * e] = hv[4 * m + 3];
    }
    for (uint64_t n = 0; n < z.N; ++n) { for (int k = 0; k < 3; ++k) bpos[4 * n + k] = hr[3 * n + k]; bpos[4 * n + 3] = 0.f; }
    const void *d0, *d1, *d2, *d3;
    if ((rc = import_array(pl, spos.data(), spos.size() * 4, QDAS_MEM_HOST, &d0))) return rc;
    if ((rc = import_array(pl, sst.data(), sst.size() * 4, QDAS_MEM_HOST, &d1))) return rc;
    if ((rc = import_array(pl, bpos.data(), bpos.size() * 4, QDAS_MEM_HOST, &d2))) return rc;
    if ((rc = import_array(pl, bnrm.data(), bnrm.size() * 4, QDAS_MEM_HOST, &d3))) return rc;
    t.Pr = (const float *)d0; t.St = (const float *)d1; t.Pv = (const float *)d2; t.Nv = (const float *)d3;
    return QDAS_OK;
}

// tile grid: (1 << tz_log2) pixels of I1 x tile_cols columns (columns = I2*I3 flattened)
static void plan_set_grid(qdas_plan *pl, int tzl) {
    TileParams &t = pl->tp;
    const qdas_sizes &z = pl->d.sz;
    t.tz_log2 = tzl;
    pl->tile_cols = ((unsigned)pl->tc.waves * 64u) >> tzl;
    const uint64_t col0 = pl->d.i_begin / z.I1, col1 = t.mir == 1 ? (z.I2 + 1) / 2 - 1 : (pl->d.i_begin + pl->i_count - 1) / z.I1;   // (mirror mode: the first half of the columns)
    t.tiles_z = (uint32_t)((z.I1 + (1u << tzl) - 1) >> tzl);
    t.tile_x0 = (uint32_t)(col0 / pl->tile_cols);
    t.tiles_x = (uint32_t)(col1 / pl->tile_cols) - t.tile_x0 + 1;
    pl->ntiles = t.tiles_z * t.tiles_x;
}

// the tiled kernel's parameter block from the resolved modes (everything except weights and the probed shape)
static int plan_setup_tile_params(qdas_plan *pl, const qdas_desc *desc, PlanBuild &b) {
    const qdas_sizes &z = pl->d.sz;
    const GenericParams &g = pl->gp;
    TileParams &t = pl->tp;
    int rc;
    t.Pi = (const float *)g.Pi; t.Pr = (const float *)g.Pr; t.Pv = (const float *)g.Pv; t.Nv = (const float *)g.Nv; t.St = nullptr;
    t.T = z.T; t.N = z.N; t.M = z.M; t.I1 = z.I1; t.I2 = z.I2; t.I3 = z.I3;
    t.i_begin = desc->i_begin; t.i_count = pl->i_count;
    const bool tp = z.flag & QDAS_FLAG_TPOSE;
    t.strN = tp ? z.T * z.M : z.T;                  // reference src/bf.cu:100
    t.strM = tp ? z.T : z.T * z.N;
    b.txkind = z.VS ? (z.DV ? 0 : 1) : 2;           // distance | signed distance | plane wave
    t.kindB = b.txkind; t.kindS = 0;
    if (b.sy.swap) {                                // roles swapped: stage elements = transmits, block elements = receivers
        if ((rc = upload_swapped_geometry(pl, desc, z.M, 0))) return rc;
        t.N = z.M; t.M = z.N;
        std::swap(t.strN, t.strM);
        t.kindB = 0; t.kindS = b.txkind;
    }
    t.fs = g.fs; t.fmod = g.fmod;
    t.cinv_fs = pl->cinv0 * g.fs;
    t.cinv_pix = b.rq.cmap ? (const float *)g.cinv + g.cst[5] : nullptr;
    t.flag = z.flag; t.VS = z.VS; t.DV = z.DV; t.sym = b.sy.sym; t.big = b.sy.big; t.fold = b.sy.rfold;
    t.mir = b.sy.mir ? (b.sy.mslab ? 2 : 1) : 0;
    unsigned max_tiles = 0;
    for (int l = 3; l <= 6; ++l) { plan_set_grid(pl, l); if (pl->ntiles > max_tiles) max_tiles = pl->ntiles; }
    void *fb;
    if ((rc = dev_alloc(pl, &fb, sizeof(uint32_t) * (max_tiles + 1)))) return rc;
    pl->fallback = (uint32_t *)fb;
    t.fallback_list = pl->fallback; t.fallback_cap = max_tiles;
    t.probe = 0; t.wz_log2 = 6; t.ksplit = 1; t.part = nullptr; t.nfr = 1; t.x_fstride = 0; t.y_fstride = 0;
    t.syn = b.rq.syn ? 1 : 0; t.y_ld = pl->y_ld;
    t.bf = b.rq.bfm ? 1 : 0;
    t.bf_pn = (z.flag & QDAS_FLAG_TPOSE) ? z.M : 1;      // plane nm = the data's aperture order (src/bf.cu:100,135)
    t.bf_pm = (z.flag & QDAS_FLAG_TPOSE) ? 1 : z.N;
    t.wtab = nullptr; t.apix = nullptr; t.apix_real = desc->apod_real;
    t.gen_kind = g.gen_kind; t.gen_p0 = g.gen_p0; t.gen_p1 = g.gen_p1; t.rxn = (const float *)g.rxn;
    t.bpix = nullptr;
    return QDAS_OK;
}

// product of the arrays `take` selects, broadcast to I x E entries (E = M: indexed by the transmit; N: by the receiver; 1: pixel-only)
template <class Take>
static int fold_pixel_arrays(qdas_plan *pl, const qdas_desc *desc, size_t ael, int side, Take take, void **out) {
    const qdas_sizes &z = pl->d.sz;
    const GenericParams &g = pl->gp;
    const int dt = z.dtype;
    ApodFold f;
    memset(&f, 0, sizeof f);
    f.base = g.apod; f.I1 = z.I1; f.I2 = z.I2; f.I3 = z.I3;
    f.E = side == 2 ? 1 : side == 1 ? z.M : z.N;
    for (uint64_t s = 0; s < z.S; ++s) {
        if (!take(s)) continue;
        const uint64_t *a = &g.ast[6 * s];
        uint64_t *q = f.st[f.ns++];
        q[0] = a[0]; q[1] = a[1]; q[2] = a[2]; q[3] = side == 2 ? 0 : side == 1 ? (z.M > 1 ? a[4] : 0) : (z.N > 1 ? a[3] : 0); q[4] = a[5];
    }
    const uint64_t nel = z.I1 * z.I2 * z.I3 * f.E;
    void *fb;
    int frc = dev_alloc(pl, &fb, nel * ael);
    if (frc) return frc;
    const unsigned nb = (unsigned)std::min<uint64_t>((nel + 255) / 256, 1u << 20);
    if (desc->apod_real) {
        if (dt == QDAS_F32) apod_fold_kernel<float, false><<<nb, 256, 0, 0>>>(f, (float *)fb, nel);
        else apod_fold_kernel<_Float16, false><<<nb, 256, 0, 0>>>(f, (_Float16 *)fb, nel);
    } else {
        if (dt == QDAS_F32) apod_fold_kernel<float, true><<<nb, 256, 0, 0>>>(f, (float *)fb, nel);
        else apod_fold_kernel<_Float16, true><<<nb, 256, 0, 0>>>(f, (_Float16 *)fb, nel);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(0));
    *out = fb;
    return QDAS_OK;
}

// the pixel-dependent apodization of the plan: used in place, or multiplied into one plan-owned I x [N | M | 1] array (two with per-pair weights)
static int plan_pixel_weights(qdas_plan *pl, const qdas_desc *desc, PlanBuild &b) {
    const qdas_sizes &z = pl->d.sz;
    const GenericParams &g = pl->gp;
    TileParams &t = pl->tp;
    const modes::Request &rq = b.rq;
    int rc;
    if (rq.bpix_mode) {                                // transmit side (+ pixel-only arrays) -> stage weight; receive side -> per-pair weight (real fp32)
        void *fa_ = nullptr, *fb_ = nullptr;
        auto on_rx = [&](uint64_t s) { return g.ast[6 * s + 3] != 0 && z.N > 1; };
        if ((rc = fold_pixel_arrays(pl, desc, b.ael, 1, [&](uint64_t s) { return !on_rx(s); }, &fa_))) return rc;
        if ((rc = fold_pixel_arrays(pl, desc, b.ael, 0, on_rx, &fb_))) return rc;
        t.apix = (const unsigned char *)fa_; t.bpix = (const float *)fb_;
    } else if (rq.pix_fold) {
        void *fb = nullptr;
        if ((rc = fold_pixel_arrays(pl, desc, b.ael, rq.pix_only ? 2 : rq.pix_is_tx ? 1 : 0, [&](uint64_t s) { return rq.is_pix[s]; }, &fb))) return rc;
        t.apix = (const unsigned char *)fb;
    } else if (rq.pix_arr >= 0) {
        t.apix = (const unsigned char *)g.apod + g.ast[6 * rq.pix_arr + 5] * b.ael;
    }
    t.apix_pixel_only = rq.pix_only ? 1 : 0;
    t.act_bytes = ((t.apix || t.gen_kind) && z.dtype != QDAS_F64) ? (uint32_t)(8 * (t.N + 1)) : 0u;      // (fp64 data: the plain list of stages)
    return QDAS_OK;
}

// the pixel-independent apodization arrays folded into one N x M table (complex128 for fp64 data, else complex64; folded plans hand it to the fold pass)
static int plan_weight_table(qdas_plan *pl, const qdas_desc *desc, PlanBuild &b) {
    const qdas_sizes &z = pl->d.sz;
    const GenericParams &g = pl->gp;
    TileParams &t = pl->tp;
    const int dt = z.dtype;
    const size_t ael = b.ael;
    int rc;
    if (z.S > b.rq.npix && dt == QDAS_F64) {             // fp64 data: the table in double (complex128 entries)
        std::vector<double> tab(2 * z.N * z.M);
        for (size_t k = 0; k < z.N * z.M; ++k) { tab[2 * k] = 1.0; tab[2 * k + 1] = 0.0; }
        for (uint64_t s = 0; s < z.S; ++s) {
            if (b.rq.is_pix[s]) continue;                 // (the pixel x receiver array is applied per stage: das_tile_impl.h TileCfg::W64)
            const uint64_t *st = &g.ast[6 * s];
            const uint64_t nel = bcast_numel(st, z);
            std::vector<double> raw(nel * (desc->apod_real ? 1 : 2));
            if ((rc = fetch_host((const unsigned char *)desc->apod + st[5] * ael, nel * ael, desc->mem, raw.data()))) return rc;
            for (uint64_t m = 0; m < z.M; ++m)
                for (uint64_t n = 0; n < z.N; ++n) {
                    const uint64_t k = n * st[3] + m * st[4];
                    const double ar = desc->apod_real ? raw[k] : raw[2 * k], ai = desc->apod_real ? 0.0 : raw[2 * k + 1];
                    double &tr = tab[2 * (n + z.N * m)], &ti = tab[2 * (n + z.N * m) + 1];
                    const double nr = tr * ar - ti * ai, ni = tr * ai + ti * ar;
                    tr = nr; ti = ni;
                }
        }
        void *dtab;
        if ((rc = dev_alloc(pl, &dtab, tab.size() * sizeof(double)))) return rc;
        hipError_t e = upload(dtab, tab.data(), tab.size() * sizeof(double));
        if (e != hipSuccess) return fail(QDAS_EHIP, "hipMemcpy(wtab): %s", hipGetErrorString(e));
        t.wtab = dtab;
    } else if (z.S > b.rq.npix && dt != QDAS_F64) {
        std::vector<float> &tab = b.host_tab;
        tab.assign(2 * z.N * z.M, 0.f);
        for (size_t k = 0; k < z.N * z.M; ++k) { tab[2 * k] = 1.f; tab[2 * k + 1] = 0.f; }
        for (uint64_t s = 0; s < z.S; ++s) {
            if (b.rq.is_pix[s]) continue;
            const uint64_t *st = &g.ast[6 * s];
            const uint64_t nel = bcast_numel(st, z);
            std::vector<unsigned char> raw(nel * ael);
            if ((rc = fetch_host((const unsigned char *)desc->apod + st[5] * ael, nel * ael, desc->mem, raw.data()))) return rc;
            for (uint64_t m = 0; m < z.M; ++m)
                for (uint64_t n = 0; n < z.N; ++n) {
                    const uint64_t k = n * st[3] + m * st[4];
                    float ar, ai = 0.f;
                    if (desc->apod_real) ar = dt == QDAS_F32 ? ((const float *)raw.data())[k] : half_to_float(((const uint16_t *)raw.data())[k]);
                    else if (dt == QDAS_F32) { ar = ((const float *)raw.data())[2 * k]; ai = ((const float *)raw.data())[2 * k + 1]; }
                    else { ar = half_to_float(((const uint16_t *)raw.data())[2 * k]); ai = half_to_float(((const uint16_t *)raw.data())[2 * k + 1]); }
                    const size_t q = b.sy.swap ? (m + z.M * n) : (n + z.N * m);       // [stage element + stages * block element]
                    float &tr = tab[2 * q], &ti = tab[2 * q + 1];
                    const float nr = tr * ar - ti * ai, ni = tr * ai + ti * ar;
                    tr = nr; ti = ni;
                }
        }
        void *dtab;
        if ((rc = dev_alloc(pl, &dtab, tab.size() * sizeof(float)))) return rc;
        hipError_t e = upload(dtab, tab.data(), tab.size() * sizeof(float));
        if (e != hipSuccess) return fail(QDAS_EHIP, "hipMemcpy(wtab): %s", hipGetErrorString(e));
        t.wtab = dtab;
    }
    if (t.fold && t.wtab) { pl->fold_wtab = t.wtab; t.wtab = nullptr; }      // (folded data: the fold pass applies the table, trace by trace -- it need not be symmetric in any way)
    pl->wtab_real = !b.host_tab.empty();               // every entry of the table real?  (apodization windows usually are: hiprtc builds then accumulate with ONE packed FMA per sample)
    for (size_t k = 1; k < b.host_tab.size(); k += 2) if (b.host_tab[k] != 0.f) { pl->wtab_real = false; break; }
    return QDAS_OK;
}

// FACT: the folded table equals its own mirror image, w[n,m] == w[N-1-n,M-1-m] bit for bit (receive and transmit windows are)
static bool table_mirror_symmetric(const qdas_sizes &z, const std::vector<float> &tab, bool swap) {
    for (uint64_t m = 0; m < z.M; ++m)
        for (uint64_t n = 0; n < z.N; ++n) {
            const size_t q = swap ? (m + z.M * n) : (n + z.N * m), q2 = swap ? ((z.M - 1 - m) + z.M * (z.N - 1 - n)) : ((z.N - 1 - n) + z.N * (z.M - 1 - m));
            if (memcmp(&tab[2 * q], &tab[2 * q2], 8) != 0) return false;
        }
    return true;
}

// step 3 of plan_modes.h: the chain of window-fit probes; each step = one choose_tile_shape (four footprints on the device)
static int plan_probe_chain(qdas_plan *pl, const qdas_desc *desc, PlanBuild &b) {
    const qdas_sizes &z = pl->d.sz;
    TileParams &t = pl->tp;
    modes::ProbeState ps;
    ps.dtype = z.dtype; ps.sym = t.sym; ps.rfold = t.fold; ps.mir = t.mir; ps.narrow = 0; ps.bpix = b.rq.bpix_mode;
    ps.set_tc(b.sy.tc_sym, b.sy.tc_narrow, b.sy.tc_fb, b.sy.tc_mirq, b.sy.tc_fold);
    if (t.fold && pl->fold_buf) HIPCHK(hipMemset(pl->fold_buf, 0, (size_t)z.T * z.N * z.M * 8));      // (the lower triangle is never written: zeros, not garbage, where a border window reaches into it)
    const bool has_table = !b.host_tab.empty();
    const bool tsym = (t.mir && has_table && !t.fold) ? table_mirror_symmetric(z, b.host_tab, b.sy.swap) : true;
    auto set_grid = [&](int tzl) { plan_set_grid(pl, tzl); };
    auto probe = [&](const modes::ProbeState &s, bool *fit) -> int {
        t.mir = s.mir; t.narrow = s.narrow;
        pl->tc = s.tc();
        const int rc = choose_tile_shape(pl, desc, set_grid);
        *fit = pl->no_fallback;
        return rc;
    };
    int err = 0;
    (void)modes::run_probe_chain(ps, b.sw, has_table, tsym, probe, &err);
    if (err) return err;
    t.mir = ps.mir; t.narrow = ps.narrow;                // (the chain's last probe ran the final state: tc, grid and shape are the plan's)
    b.sy.mir = ps.mir != 0;
    b.outcome.mir = ps.mir; b.outcome.narrow = ps.narrow;
    return QDAS_OK;
}

// Focused transmits whose focal planes cut through the image: the delay flips sign there (src/bf.cu:106-108), so the tiles a plane
// crosses fit no window and would go to the generic kernel -- with a walking aperture that is half the image.  Second attempt:
// every transmit listed twice, once per side of its plane (tile_params.h kindS == 3); kept if fewer tiles misfit.
static int plan_side_split(qdas_plan *pl, const qdas_desc *desc, PlanBuild &b) {
    const qdas_sizes &z = pl->d.sz;
    TileParams &t = pl->tp;
    b.kN_eff = b.sy.kN;
    if (!modes::side_split_applicable(*desc, b.rq, b.sy, pl->no_fallback, b.txkind, b.sy.wtb != 0, b.sw)) return QDAS_OK;
    int rc;
    const TileParams keep = t;
    const double keep_frac = pl->misfit_frac;
    const unsigned keep_ntiles = pl->ntiles, keep_cols = pl->tile_cols;
    const bool tp = z.flag & QDAS_FLAG_TPOSE;
    const uint64_t E = 2 * z.M;                          // stage elements: (transmit, side)
    if ((rc = upload_swapped_geometry(pl, desc, E, 1))) return rc;
    t.N = E; t.M = z.N;
    t.strN = tp ? z.T : z.T * z.N;                       // stage element = transmit, block element = receiver
    t.strM = tp ? z.T * z.M : z.T;
    t.kindB = 0; t.kindS = 3; t.stage_shift = 1;
    t.gen_kind = t.apix ? 6 : 5; t.gen_p0 = t.gen_p1 = 0.0; t.rxn = nullptr;      // (6: the side rule times the plan's pixel x transmit / pixel-only array)
    t.act_bytes = (uint32_t)(8 * (E + 1));
    if (t.wtab) {                                        // the table in the new element order: [(2m + side) + 2M * n]
        std::vector<float> tab2(2 * E * z.N);
        for (uint64_t n = 0; n < z.N; ++n)
            for (uint64_t e = 0; e < E; ++e) {
                const uint64_t m = e >> 1, q = b.sy.swap ? (m + z.M * n) : (n + z.N * m);
                tab2[2 * (e + E * n)] = b.host_tab[2 * q]; tab2[2 * (e + E * n) + 1] = b.host_tab[2 * q + 1];
            }
        void *dtab2;
        if ((rc = dev_alloc(pl, &dtab2, tab2.size() * sizeof(float)))) return rc;
        hipError_t e2 = upload(dtab2, tab2.data(), tab2.size() * sizeof(float));
        if (e2 != hipSuccess) return fail(QDAS_EHIP, "hipMemcpy(wtab): %s", hipGetErrorString(e2));
        t.wtab = dtab2;
    }
    auto set_grid = [&](int tzl) { plan_set_grid(pl, tzl); };
    if ((rc = choose_tile_shape(pl, desc, set_grid))) return rc;
    if (pl->misfit_frac < keep_frac) { b.kN_eff = E; b.outcome.side_split = true; }
    else {                                               // no better: the plan as it was
        t = keep;
        pl->misfit_frac = keep_frac; pl->no_fallback = keep_frac == 0.0; pl->ntiles = keep_ntiles; pl->tile_cols = keep_cols;
        HIPCHK(hipMemset(pl->fallback, 0, sizeof(uint32_t)));
    }
    return QDAS_OK;
}

// Tiles that still do not fit (a pixel grid coarser than about lambda/2 -- volumes, previews --, steep delay gradients): fp32 plans
// try the 384-sample windows of launch configuration 14 (16 transmits per stage, same LDS image); kept if fewer tiles misfit.
// (a reciprocal plan gives up its mode for them: an image on the generic kernel costs ten times more than the shared index work saves)
static int plan_wide_windows(qdas_plan *pl, const qdas_desc *desc, PlanBuild &b) {
    const qdas_sizes &z = pl->d.sz;
    TileParams &t = pl->tp;
    if (!modes::wide_applicable(z.dtype, pl->no_fallback, b.rq.bfm, t.big, t.narrow, pl->prefolded, t.N, t.M, t.strN, t.strM, t.act_bytes != 0,
                                t.wtab || (t.fold && pl->fold_wtab), b.sw)) return QDAS_OK;      // (a folded plan's table comes back into the kernel there)
    int rc;
    const TileParams keep = t;
    const TileConfig keep_tc = pl->tc;
    const double keep_frac = pl->misfit_frac;
    const unsigned keep_ntiles = pl->ntiles, keep_cols = pl->tile_cols;
    t.narrow = 2; t.sym = 0;
    if (t.fold) { t.fold = 0; t.wtab = pl->fold_wtab; }      // (the wide-window configuration is a general-mode kernel on the frame as it is)
    pl->tc = tile_config(z.dtype, 0, 2);
    auto set_grid = [&](int tzl) { plan_set_grid(pl, tzl); };
    if ((rc = choose_tile_shape(pl, desc, set_grid))) return rc;
    b.outcome.wide = pl->misfit_frac < keep_frac;
    if (!b.outcome.wide) {
        t = keep; pl->tc = keep_tc;
        pl->misfit_frac = keep_frac; pl->no_fallback = keep_frac == 0.0; pl->ntiles = keep_ntiles; pl->tile_cols = keep_cols;
        HIPCHK(hipMemset(pl->fallback, 0, sizeof(uint32_t)));
    }
    return QDAS_OK;
}

// Stage shape of plan-specialised builds.  The prebuilt configurations stage 16 transmits x 2 window sets x 192 samples (general-mode lateral-mirror plans:
// BASELINE C2, plane-wave compounding) or 32 transmits x 192 samples (one window set: general mode, the reciprocity fold without the mirror mode); fp16 data: 384 samples.  A
// hiprtc build takes any shape: when every tile of some footprint fits ONE-KiB windows (128 samples of fp32, 256 of fp16 data), 32 x 2 resp. 64 x 1 of them -- the same 64 KiB
// per buffer (the one-set plans: 64 instead of 48), HALF the stages (barriers, receive-delay evaluations, weight loads, DMA issue) -- is the better shape (C2 1.70 -> 1.61 ms;
// C3 without any symmetry -6 %, the fold alone -3 %; round 6, fp16 data incl. plans with a pixel x receiver weight: BASELINE C5 1.73 -> 1.59 ms, same box).  Asked with the
// prebuilt probe kernels (tile_params.h probe_w); if the build fails later the plan runs the prebuilt kernel on the footprint chosen here (the narrower window fits the wider).
static int plan_stage_shape(qdas_plan *pl, const qdas_desc *desc, PlanBuild &b) {
    const qdas_sizes &z = pl->d.sz;
    TileParams &t = pl->tp;
    const bool jit_on = (desc->plan_flags & QDAS_PLAN_JIT) && !b.sw.no_jit;
    const bool f16 = z.dtype == QDAS_F16;
    // (fp32 frames with a pixel x receiver weight took this shape in round 6 as well: one accumulator per pixel -- TileCfg::ONEACC -- made room for the weighted totals, and a build
    //  that still needs scratch is rebuilt with the plain pair loop, jit_get_kernel_nospill: C3 with a generated f/1.5 mask 13.3 -> see profiles/r06.
    //  fp16 data: the two-window-set plans only -- one set of 64 transmits measured SLOWER on BASELINE C5 without the mirror mode, 2.37 -> 2.80 ms: M = 96 is 1.5 such blocks)
    if (!(jit_on && (z.dtype == QDAS_F32 || f16) && !t.big && !t.bf && !t.syn && !t.bpix && t.narrow == 0 && !t.stage_shift && !t.cinv_pix
          && !(f16 && (t.sym || !t.mir)) && pl->no_fallback && !getenv("QDAS_NO_STAGE_SHAPE") && !getenv("QDAS_JIT_MB") && !getenv("QDAS_JIT_W"))) return QDAS_OK;
    int mb = 0, sets = 1;
    if (t.mir && !t.sym) { mb = 32; sets = 2; }                          // two window sets: a pixel and its mirror image
    else if (!t.mir && (!t.sym || t.fold)) { mb = 64; sets = 1; }        // one window set
    if (!mb || t.M < (uint64_t)mb) return QDAS_OK;
    const int w1k = f16 ? 256 : 128;                                     // samples of a one-KiB window
    // LDS image of the specialised build: the header of the prebuilt configuration (+ the stage weights of the longer stages) + 2 buffers x sets x mb windows x 1 KiB
    const int pixw = t.act_bytes ? 1 : 0;
    const size_t hdr = tile_lds_bytes(z.dtype, t.sym, t.N, t.M, 0, pixw, t.wtab ? 1 : 0, 0, t.fold) - tile_config(z.dtype, t.sym, 0, t.mir ? 2 : 1, 0, t.fold).lds_bytes;
    if (hdr + (t.wtab ? (size_t)2 * 2 * (size_t)mb * 8 : 0) + (size_t)2 * sets * mb * 1024 > (size_t)158 * 1024) return QDAS_OK;
    int rc;
    const TileParams keep = t;
    const double keep_frac = pl->misfit_frac;
    const unsigned keep_ntiles = pl->ntiles, keep_cols = pl->tile_cols;
    const bool keep_nf = pl->no_fallback;
    t.probe_w = w1k;
    auto set_grid = [&](int tzl) { plan_set_grid(pl, tzl); };
    if ((rc = choose_tile_shape(pl, desc, set_grid))) return rc;
    const bool fits = pl->no_fallback;
    t.probe_w = 0;
    if (fits) { pl->hint_mb = mb; pl->hint_w = w1k; }
    else {
        t = keep;
        pl->misfit_frac = keep_frac; pl->no_fallback = keep_nf; pl->ntiles = keep_ntiles; pl->tile_cols = keep_cols;
        HIPCHK(hipMemset(pl->fallback, 0, sizeof(uint32_t)));
    }
    return QDAS_OK;
}

// step 4 of plan_modes.h: workgroups per tile (+ the partial images of a split aperture)
static int plan_split_aperture(qdas_plan *pl, PlanBuild &b) {
    const qdas_sizes &z = pl->d.sz;
    TileParams &t = pl->tp;
    int ncu = 0, rc;
    HIPCHK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, pl->device));
    const unsigned cus = ncu > 0 ? (unsigned)ncu : 256u;
    int mb_eff = pl->hint_mb ? pl->hint_mb : pl->tc.mb;             // (transmits per stage of the kernel the plan means to run: plan_stage_shape)
    int mb_grp = pl->hint_mb ? pl->hint_mb : pl->tc.mb;             // (... and of the transmit-block groups of a two-dimensional split)
    if (const char *e = getenv("QDAS_JIT_MB")) { const int mb = atoi(e); if (mb >= 2 && mb % pl->tc.waves == 0 && (b.desc->plan_flags & QDAS_PLAN_JIT) && !b.sw.no_jit) { mb_eff = mb; mb_grp = mb; } }      // (tuning builds, plan_jit)
    t.ksplit = modes::choose_ksplit(pl->ntiles, cus, z.M, mb_eff, t.sym != 0, b.kN_eff, t.act_bytes != 0, t.syn != 0, b.sw);
    t.ksplit_m = modes::choose_ksplit_m(&t.ksplit, t.M, mb_grp, t.sym != 0, t.act_bytes != 0, t.syn != 0, z.dtype, b.sw);
    if (t.ksplit > 1 && !t.bf) {
        void *pb;
        if ((rc = dev_alloc(pl, &pb, sizeof(float) * 2 * (size_t)t.ksplit * 4 * pl->i_count * (t.mir == 2 ? 2 : 1)))) return rc;   // x4: up to four frames per launch (fp64 data: one complex128 frame -- fits as well)
        t.part = (float2 *)pb;
    }
    return QDAS_OK;
}

// dynamic LDS of a plan-specialised build: header (Tile::setup) + window buffers (or the prologue's scratch, which aliases them)
static size_t jit_tile_lds(const JitSpec &k, uint64_t tN, uint64_t tM, uint32_t act_bytes, bool wtab) {
    const size_t MX = std::min<size_t>(tM > tN ? tM : tN, QDAS_PROLOGUE_CHUNK);
    const size_t off_act = (((((2 * tM + tN) * 4 + 15) & ~(size_t)15) + 16 * tN + 7 * tM * 4) + 15) & ~(size_t)15;   // Tile::setup
    const size_t off_wst = off_act + (((size_t)act_bytes + 15) & ~(size_t)15);
    const size_t hdr = (off_wst + (wtab ? (size_t)k.nbuf * (2 * (size_t)k.mb * 8 + 16) : 0) + 15) & ~(size_t)15;
    size_t body = (size_t)k.nbuf * k.mb * (k.fold ? (k.mirq ? 2 : 1) : k.mirq ? 4 : (k.sym || k.mir) ? 2 : 1) * k.w * (k.dtype == QDAS_F16 ? 4 : 8);
    const size_t scratch = 2 * (size_t)k.waves * MX * 4 + 1024;
    if (body < scratch) body = scratch;
    return hdr + body;
}

// "No kernel spills" is a property the library enforces, not a sentence: a build that uses scratch memory (the long-stage two-window-set fp32 shapes are at the
// register limit -- delay kinds, a split aperture, roles swapped move them by a register or two) is rebuilt with the plain instead of the software-pipelined
// pair loop (16 tap registers less; same-box A/B < 1 %: profiles/r06/oneacc_ab.txt).  QDAS_JIT_SPEC_LOG=<file> logs the final spec of every build:
// tests/test_jit.py rebuilds those (tests/jit_kernels.txt) without a device and fails on a spilled register.
static std::string jit_get_kernel_nospill(JitSpec &k, int device, hipFunction_t *fn, std::string *key) {
    // (roles swapped -- stage elements with their own delay kind and {t0, normal} records -- are known to need the plain loop: asked for up front, so that no spilling build is made at all)
    if (!k.plain && k.mir && !k.sym && k.dtype == QDAS_F32 && k.mb >= 32 && k.has_st) k.plain = 1;
    std::string err = jit_get_kernel(k, device, fn, key);
    auto scratch_of = [](hipFunction_t f) -> int {
        int scratch = 0;
        if (hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, f) != hipSuccess) { (void)hipGetLastError(); return 0; }
        return scratch;
    };
    if (err.empty() && !k.plain && k.mir && !k.sym && k.dtype == QDAS_F32 && k.mb >= 32 && scratch_of(*fn) > 0) {
        k.plain = 1;
        hipFunction_t fn2 = nullptr;
        std::string key2;
        const std::string err2 = jit_get_kernel(k, device, &fn2, &key2);
        if (err2.empty()) { *fn = fn2; *key = key2; } else k.plain = 0;
    }
    // A build that STILL uses scratch memory is not used at all (round 6).  Not a matter of speed: the pair loops read their taps with inline-asm ds_read
    // instructions and wait for them by hand (s_waitcnt lgkmcnt further down); a register the compiler spills between the two is stored BEFORE its
    // data has arrived.  Fuzz seed 126301 with hiprtc builds forced found such a build (roles swapped onto two transmits, 16-element stages of 384-sample
    // windows, N = 2 as a constant: 352 spilled registers): images that differed from run to run by 0.8 % of the peak in the pixels of one wave.  The plan
    // keeps the prebuilt / built-on-demand kernel of its shape (checked for spills at build time: tests/test_build_regs.py, tile_variant_check).
    if (err.empty()) {
        const int sc = scratch_of(*fn);
        if (sc > 0 && !getenv("QDAS_JIT_ALLOW_SCRATCH")) {
            *fn = nullptr;
            return "hiprtc build uses " + std::to_string(sc) + " bytes of scratch memory per lane (spilled registers): not used";
        }
    }
    if (err.empty()) {
        if (const char *lf = getenv("QDAS_JIT_SPEC_LOG")) {
            if (FILE *f = fopen(lf, "a")) { fprintf(f, "%s\n", jit_spec_string(k).c_str()); fclose(f); }
        }
    }
    return err;
}

// QDAS_PLAN_JIT: the tiled kernel compiled for this plan's sizes (jit.hip).  A failure is not an error: the plan keeps its prebuilt kernel and
// qdas_last_error() says why -- unless the plan's mode exists as a hiprtc build only: *remake = the plan flag to add for a second attempt without it.
static int plan_jit(qdas_plan *pl, const qdas_desc *desc, int *remake) {
    const qdas_sizes &z = pl->d.sz;
    const int dt = z.dtype;
    *remake = 0;
    g_err.clear();
    if (!((desc->plan_flags & QDAS_PLAN_JIT) && pl->kernel == QDAS_KERNEL_TILED && !pl->tp.bf && !getenv("QDAS_NO_JIT"))) return QDAS_OK;
    const TileParams &t = pl->tp;
    JitSpec k{};
    k.interp = (z.flag & 7) == 4 ? 1 : (z.flag & 7); k.dtype = dt; k.fmod = t.fmod != 0.0; k.wtab = t.wtab != nullptr; k.sym = t.sym; k.big = t.big;
    const int narrow = (t.sym && dt == QDAS_F32 && t.narrow) ? 1 : (!t.sym && dt == QDAS_F32 && t.narrow == 2) ? 2 : 0;
    const int mirq = (t.sym && t.mir) ? 1 : 0;
    const Cfg &cg = CFGS[cfg_index(dt, t.sym, t.fold ? 1 : (t.mir ? 2 : 1), narrow, mirq, t.fold)];      // (fb = 2 selects the two-window-set configuration of a general-mode mirror plan; folded plans: one frame)
    k.mir = t.sym ? 0 : t.mir; k.mirq = mirq; k.mslab = t.mir == 2; k.fold = t.fold;
    k.waves = cg.waves; k.mb = cg.mb; k.w = cg.w; k.nbuf = cg.nbuf;
    k.N = t.N; k.M = t.M; k.T = t.T; k.I1 = t.I1; k.strN = t.strN; k.strM = t.strM;
    k.kindB = t.kindB; k.kindS = t.kindS; k.tzl = t.tz_log2; k.wzl = t.wz_log2; k.ksplit = t.ksplit;
    k.gen_kind = t.gen_kind; k.has_apix = t.apix != nullptr; k.apix_real = t.apix_real; k.syn = t.syn;
    k.has_st = t.St != nullptr; k.has_cinv_pix = t.cinv_pix != nullptr;
    k.wreal = (t.wtab && pl->wtab_real && !getenv("QDAS_NO_WREAL")) ? 1 : 0;
    // tuning: a specialised build may use another number of transmits per stage than the prebuilt configuration (its register
    // budget is smaller); the LDS image grows with it
    // reciprocal mode: the specialised kernel has the registers for 32-transmit stages (half the stages, barriers and per-stage
    // delay evaluations of the prebuilt 16-transmit configuration) whenever the 64 windows of a buffer stay within the 16-bit
    // immediate offsets of the LDS reads (C3: 30.9 -> 29.2 ms)
    if (t.sym && !mirq && !t.fold && z.M % 32 == 0 && 2 * 32 * k.w * (dt == QDAS_F16 ? 4 : 8) <= 65536 && !getenv("QDAS_JIT_NO_MB32")) k.mb = 32;
    if (pl->hint_mb && pl->hint_w) { k.mb = pl->hint_mb; k.w = pl->hint_w; }      // (plan_stage_shape: the tiles were probed for this window)
    if (const char *e = getenv("QDAS_JIT_MB")) {
        const int mb = atoi(e);
        if (mb >= 2 && mb % k.waves == 0 && (!t.sym || t.fold || z.M % (uint64_t)mb == 0) && (!mirq || t.fold)) k.mb = mb;
    }
    if (const char *e = getenv("QDAS_JIT_NBUF")) { const int nb = atoi(e); if (nb >= 2 && nb <= 4) k.nbuf = nb; }
    if (const char *e = getenv("QDAS_JIT_W")) { const int wv = atoi(e); if (wv >= 64 && wv <= 1024 && wv % 64 == 0) k.w = wv; }      // (experiments: tiles that do not fit go to the generic kernel)
    pl->jit_lds = dt == QDAS_F64 ? 0 : jit_tile_lds(k, t.N, t.M, t.act_bytes, t.wtab != nullptr);       // (fp64 data: the prebuilt configuration's own LDS image, das_tile.hip)
    std::string key;
    std::string err = pl->jit_lds > (size_t)160 * 1024 ? std::string("LDS image too large for the requested configuration")
                                                       : jit_get_kernel_nospill(k, pl->device, &pl->jit_fn, &key);
    if (err.empty()) { pl->jit_tag = "jit " + key; pl->jit_mb = k.mb; pl->jit_w = k.w; return QDAS_OK; }
    pl->jit_fn = nullptr; g_err = "QDAS_PLAN_JIT: " + err + " -- using the prebuilt kernel";
    const bool unfolded = dt == QDAS_F32 && t.sym && !t.fold;                  // (likewise: the general kernels then)
    if ((dt == QDAS_F32 && t.mir && !t.sym && (t.apix || t.gen_kind)) || unfolded)      // exists only as a hiprtc build: the same plan without the mode
        *remake = unfolded ? QDAS_PLAN_NO_RECIPROCAL : QDAS_PLAN_NO_MIRROR;
    return QDAS_OK;
}

// fp16 reciprocal data on the folded fp32 kernels: an fp32 PREFOLDED child plan over the same geometry and slab; this plan folds each frame into a
// complex64 copy (its weight table applied on the way), the child beamforms it, the complex64 image is rounded to complex32.  If anything of that is
// not available (memory, a child that does not fit) the plan keeps its own fp16 reciprocal kernels.
static void plan_f16_child(qdas_plan *pl, const qdas_desc *desc) {
    const qdas_sizes &z = pl->d.sz;
    if (!(pl->kernel == QDAS_KERNEL_TILED && z.dtype == QDAS_F16 && pl->tp.sym && !pl->tp.syn && !pl->tp.bf && !(desc->plan_flags & QDAS_PLAN_NO_FOLD) && !getenv("QDAS_NO_FOLD")
          && !getenv("QDAS_NO_FOLD16") && z.N >= 2 && z.N <= 65535)) return;
    qdas_desc d = *desc;
    const GenericParams &gg = pl->gp;
    uint64_t acs[6] = {gg.cst[0], gg.cst[1], gg.cst[2], gg.cst[3], gg.cst[4], gg.cst[5]};
    d.sz.dtype = QDAS_F32; d.sz.S = 0;
    d.Pi = gg.Pi; d.Pr = gg.Pr; d.Pv = gg.Pv; d.Nv = gg.Nv; d.cinv = gg.cinv; d.apod = nullptr; d.rx_normals = nullptr; d.rx_apod_kind = 0; d.acstride = acs;
    d.mem = QDAS_MEM_DEVICE; d.device = pl->device; d.y_ld = 0;
    d.plan_flags = (d.plan_flags | QDAS_PLAN_PREFOLDED) & ~(QDAS_PLAN_COPY_INPUTS | QDAS_PLAN_NO_FOLD);
    const std::string keep = g_err;
    qdas_plan *child = nullptr;
    void *fb = nullptr, *yb = nullptr;
    if (qdas_plan_create(&child, &d) == QDAS_OK && child && child->kernel == QDAS_KERNEL_TILED && child->prefolded
        && hipMalloc(&fb, (size_t)z.T * z.N * z.M * 8) == hipSuccess && hipMalloc(&yb, (size_t)child->y_ld * 8 + 16) == hipSuccess
        && hipMemset(fb, 0, (size_t)z.T * z.N * z.M * 8) == hipSuccess) {
        pl->owned.push_back(fb); pl->owned.push_back(yb);
        pl->fold_buf = fb; pl->y32 = yb; pl->f16_child = child;
        pl->fold_wtab = pl->tp.wtab;               // (this plan's N x M table, float2: applied by the fold pass)
    } else {
        (void)hipGetLastError();
        if (fb) (void)hipFree(fb);
        if (yb) (void)hipFree(yb);
        if (child) qdas_plan_destroy(child);
    }
    g_err = keep;
}

// the instantiation this plan launches for one frame: libqdas.so carries it, or it is built now (das_tile_cfg.h tile_prebuilt, jit.hip lazy_tile_launch) --
// never inside an execute.  *no_compiler: the variant is missing and cannot be built (no libhiprtc.so, QDAS_NO_LAZY); g_err says which one it was.
static void plan_resolve_kernel(qdas_plan *pl, bool *no_compiler) {
    *no_compiler = false;
    if (!(pl->kernel == QDAS_KERNEL_TILED && !pl->jit_fn && !pl->f16_child)) return;      // (fp16 reciprocal data: the fp32 child plan resolved its own)
    std::string built;
    TileParams t1 = pl->tp;
    t1.nfr = 1;
    const hipError_t pe = prepare_tile(t1, pl->d.sz.dtype, pl->ntiles, &built);
    if (pe == hipErrorSharedObjectInitFailed) { *no_compiler = true; return; }
    if (pe != hipSuccess) (void)hipGetLastError();      // (anything else is the launch's to report)
    if (!built.empty()) pl->jit_tag = "built on demand " + built;
}

// step 5 of plan_modes.h (frames per launch of a stream), host staging, and what the plan must not keep
static int plan_finish(qdas_plan *pl, const qdas_desc *desc, const PlanBuild &b) {
    const qdas_sizes &z = pl->d.sz;
    const int dt = z.dtype;
    int rc;
    modes::PlanShape ps;
    ps.dtype = dt; ps.tiled = pl->kernel == QDAS_KERNEL_TILED; ps.sym = pl->tp.sym; ps.fold = pl->tp.fold; ps.mir = pl->tp.mir; ps.narrow = pl->tp.narrow; ps.big = pl->tp.big;
    ps.bf = pl->tp.bf; ps.syn = pl->tp.syn; ps.stage_shift = pl->tp.stage_shift; ps.has_apix = pl->tp.apix != nullptr; ps.has_wtab = pl->tp.wtab != nullptr;
    ps.has_bpix = pl->tp.bpix != nullptr; ps.gen_kind = pl->tp.gen_kind; ps.fmod = pl->tp.fmod; ps.N = pl->tp.N; ps.M = pl->tp.M; ps.mem_device = desc->mem == QDAS_MEM_DEVICE;
    const modes::StreamModes sm = modes::stream_modes(ps, z.N, z.M, b.sw);
    pl->fb2_ok = sm.fb2_ok; pl->fold2_ok = sm.fold2_ok; pl->fb4_off = sm.fb4_off;
    if (desc->mem == QDAS_MEM_HOST) {                   // staging buffers for x / y
        pl->x_bytes = (size_t)z.T * z.N * z.M * data_size(dt);
        pl->y_bytes = (size_t)pl->y_ld * pl->oN * pl->oM * data_size(dt);
        if ((rc = dev_alloc(pl, &pl->dx, pl->x_bytes))) return rc;
        if ((rc = dev_alloc(pl, &pl->dy, pl->y_bytes))) return rc;
    }
    // the plan keeps no pointer into caller memory it does not need: host arrays were copied; device arrays are used in place
    // (g.* / tp.*: they must stay valid for the life of the plan unless QDAS_PLAN_COPY_INPUTS made plan-owned copies)
    pl->mirror_bound = b.sy.mirror_bound; pl->recip_bound = b.sy.recip_bound;
    if (pl->fold_buf && !pl->f16_child && !(pl->kernel == QDAS_KERNEL_TILED && pl->tp.fold)) {      // (the plan left the fold after all: wide windows, strides, generic kernel)
        for (size_t k = 0; k < pl->owned.size(); ++k) if (pl->owned[k] == pl->fold_buf) { pl->owned.erase(pl->owned.begin() + (long)k); break; }
        (void)hipFree(pl->fold_buf);
        pl->fold_buf = nullptr;
    }
    pl->d.Pi = pl->d.Pr = pl->d.Pv = pl->d.Nv = pl->d.apod = pl->d.cinv = pl->d.rx_normals = nullptr;
    pl->d.acstride = nullptr;
    return QDAS_OK;
}

namespace qdas { modes::LaunchShape launch_shape(const TileParams &P, int dtype, bool jit); }      // das_tile.hip: the launcher's view of a parameter block

// The tile prologue's tables (window bases, extents, tile statistics: tile_params.h pro_tab) depend on the geometry only: computed ONCE per plan by the probe
// kernels -- the same prologue code, writing to a plan-owned buffer -- and loaded by every workgroup of every execute instead of being recomputed
// ((M + N) wave-wide reductions per wave and workgroup: 25 % of BASELINE C5's kernel time, ~10 % of C2's).  QDAS_NO_PRO_TAB=1: as before.
static int plan_cache_prologue(qdas_plan *pl) {
    TileParams &t = pl->tp;
    t.pro_tab = nullptr; t.pro_out = nullptr;
    if (getenv("QDAS_NO_PRO_TAB") || !pl->ntiles) return QDAS_OK;
    t.pro_mask = 0;
    const bool with_mask = t.act_bytes != 0 && !getenv("QDAS_NO_PRO_MASK");      // (a stage list: the tile's activity mask rides behind the statistics, plan_cache_activity)
    const size_t stride = 2 * ((size_t)t.M + t.N) + 8 + (with_mask ? (t.N + 31) / 32 : 0), bytes = (size_t)pl->ntiles * stride * sizeof(float);
    if (bytes > (1ull << 30)) return QDAS_OK;           // (thousands of elements x tens of thousands of tiles: not worth a GiB)
    void *buf = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess) { (void)hipGetLastError(); return QDAS_OK; }      // (no memory for it: the kernels compute their tables themselves)
    pl->owned.push_back(buf);
    TileParams p = t;
    p.probe = 1; p.x = nullptr; p.y = nullptr; p.wtab = nullptr; p.apix = nullptr; p.pro_out = (float *)buf; p.pro_mask = with_mask ? 1 : 0;
    HIPCHK(hipMemsetAsync(buf, 0, bytes, 0));
    const hipError_t e = launch_tile(p, pl->d.sz.dtype, pl->ntiles, nullptr);
    if (e != hipSuccess) { (void)hipGetLastError(); return QDAS_OK; }      // (a configuration without a probe kernel: as before)
    HIPCHK(hipStreamSynchronize(nullptr));
    HIPCHK(hipMemset(pl->fallback, 0, sizeof(uint32_t)));      // (the probe re-listed the misfit tiles: the frame kernels list them per launch)
    t.pro_tab = (const float *)buf;
    t.pro_mask = with_mask ? 3 : 0;                     // (3: room for the masks, not yet written -- plan_cache_activity, once the plan's kernel is resolved)
    return QDAS_OK;
}

// The tiles' activity masks (tile_params.h pro_mask): ONE launch of the plan's own kernel in mask mode -- every tile tests all its stage elements with exactly
// the rule the frames use (array / generated / side weights, the mirror image's share) and stores the mask; the frame kernels then load it instead of
// re-loading up to N weights per lane, workgroup and execute.  Any failure: the plan runs without the tables.
static void plan_cache_activity(qdas_plan *pl) {
    TileParams &t = pl->tp;
    if (pl->kernel != QDAS_KERNEL_TILED || t.pro_mask != 3) return;
    TileParams p = t;
    p.nfr = 1; p.x = nullptr; p.y = nullptr; p.ksplit = 1; p.ksplit_m = 1; p.part = nullptr; p.pro_mask = 2; p.pro_out = const_cast<float *>(t.pro_tab);
    const std::string keep = g_err;
    hipError_t e = launch_tile(p, pl->d.sz.dtype, pl->ntiles, nullptr, pl->jit_fn, pl->jit_lds);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e == hipSuccess && !pl->no_fallback) e = hipMemset(pl->fallback, 0, sizeof(uint32_t));
    g_err = keep;
    if (e == hipSuccess) t.pro_mask = 1;
    else { (void)hipGetLastError(); t.pro_mask = 0; t.pro_tab = nullptr; }      // (the table's stride counted the masks: without them, no table)
}

// the tiled kernel's side of a plan: parameter block, weights, probed shape, split aperture -- then checked against the model of plan_modes.h
static int plan_build_tiled(qdas_plan *pl, const qdas_desc *desc, PlanBuild &b) {
    int rc;
    if ((rc = plan_setup_tile_params(pl, desc, b))) return rc;
    if ((rc = plan_pixel_weights(pl, desc, b))) return rc;
    if ((rc = plan_weight_table(pl, desc, b))) return rc;
    if ((rc = plan_probe_chain(pl, desc, b))) return rc;
    if ((rc = plan_side_split(pl, desc, b))) return rc;
    if ((rc = plan_wide_windows(pl, desc, b))) return rc;
    if ((rc = plan_stage_shape(pl, desc, b))) return rc;
    if ((rc = plan_split_aperture(pl, b))) return rc;
    if ((rc = plan_cache_prologue(pl))) return rc;
    b.outcome.ksplit = pl->tp.ksplit;
    const modes::LaunchShape model = modes::derive_launch_shape(*desc, b.rq, b.sy, b.outcome), built = launch_shape(pl->tp, pl->d.sz.dtype, false);
    if (const char *f = modes::shape_mismatch(model, built))
        return fail(QDAS_EINVAL, "internal: the built plan differs from the mode model of plan_modes.h in '%s' -- please report the descriptor", f);
    return QDAS_OK;
}

static int plan_create_impl(qdas_plan *pl, qdas_plan **out, const qdas_desc *desc, bool *replaced) {
    *replaced = false;
    int rc = plan_init(pl, desc);
    if (rc) return rc;
    const qdas_sizes &z = pl->d.sz;
    DeviceGuard guard(desc->device);
    if (guard.err != hipSuccess) return fail(QDAS_EHIP, "hipSetDevice(%d): %s", desc->device, hipGetErrorString(guard.err));
    { hipError_t e = hipGetDevice(&pl->device); if (e != hipSuccess) return fail(QDAS_EHIP, "hipGetDevice: %s", hipGetErrorString(e)); }
    if (pl->I == 0 || z.N == 0 || z.M == 0) return QDAS_OK;   // empty problem: execute() just zero-fills
    if ((desc->plan_flags & QDAS_PLAN_PREFOLDED) && z.S > 0)      // (a prefolded plan runs no fold pass: nothing would apply the table -- include/qdas.h QDAS_PLAN_PREFOLDED)
        return fail(QDAS_EUNSUPPORTED, "QDAS_PLAN_PREFOLDED: apodization arrays belong to the fold (qdas_fold_desc.wtab), not to the plan that is handed folded frames");
    PlanBuild b;
    b.desc = desc;
    b.sw = read_switches();
    if ((rc = plan_import_inputs(pl, desc, b))) return rc;
    if ((rc = plan_resolve_modes(pl, desc, b))) return rc;
    if (pl->kernel == QDAS_KERNEL_TILED && (rc = plan_build_tiled(pl, desc, b))) return rc;
    const bool mslab = (desc->plan_flags & QDAS_PLAN_MIRROR_SLAB) != 0;
    if (mslab && !(pl->kernel == QDAS_KERNEL_TILED && pl->tp.mir == 2))
        return fail(QDAS_EUNSUPPORTED, "QDAS_PLAN_MIRROR_SLAB: the lateral-mirror mode is not available for this problem (geometry not mirror-symmetric "
                                       "about x = 0, weights / modes it does not take, or tiles that do not fit the staging windows): use plain slabs");
    int remake = 0;
    if ((rc = plan_jit(pl, desc, &remake))) return rc;
    if (remake) {                                       // the plan's mode exists only as a hiprtc build, which failed: the same plan without the mode
        qdas_desc d2 = *desc;
        d2.plan_flags |= remake;
        const std::string keep = g_err;
        *replaced = true;
        const int rc2 = qdas_plan_create(out, &d2);
        if (rc2 == QDAS_OK) g_err = keep;
        return rc2;
    }
    if (pl->prefolded && pl->kernel == QDAS_KERNEL_TILED && !pl->no_fallback)
        return fail(QDAS_EUNSUPPORTED, "QDAS_PLAN_PREFOLDED: tiles of this image do not fit the staging windows (they would be redone from the unfolded frame)");
    plan_f16_child(pl, desc);
    bool no_compiler = false;
    plan_resolve_kernel(pl, &no_compiler);
    if (no_compiler) {                                  // re-made on the generic kernel (identical semantics, 5-12x slower), or an error when the caller insisted on the fused kernel
        const std::string why = g_err;
        if (desc->kernel == QDAS_KERNEL_TILED || mslab || pl->prefolded) return fail(QDAS_EUNSUPPORTED, "%s", why.c_str());
        qdas_desc d2 = *desc;
        d2.kernel = QDAS_KERNEL_GENERIC;
        *replaced = true;
        const int rc2 = qdas_plan_create(out, &d2);
        if (rc2 == QDAS_OK) g_err = why + " -- using the generic kernel";
        return rc2;
    }
    plan_cache_activity(pl);
    return plan_finish(pl, desc, b);
}

extern "C" int qdas_plan_create(qdas_plan **out, const qdas_desc *desc) {
    if (!out || !desc) return fail(QDAS_EINVAL, "null argument");
    *out = nullptr;
    int rc = validate(desc);
    if (rc) return rc;
    qdas_plan *pl = new qdas_plan();
    bool replaced = false;
    rc = plan_create_impl(pl, out, desc, &replaced);
    if (rc || replaced) { delete pl; return rc; }      // (replaced: *out holds the plan that was made in this one's stead)
    *out = pl;
    return QDAS_OK;
}

extern "C" void qdas_plan_destroy(qdas_plan *pl) {
    if (!pl) return;
    DeviceGuard guard(pl->device);
    delete pl;
}

extern "C" int qdas_plan_kernel(const qdas_plan *pl) { return pl ? pl->kernel : 0; }

extern "C" int qdas_plan_fallback_tiles(const qdas_plan *pl, uint64_t *n) {
    if (!pl || !n) return fail(QDAS_EINVAL, "null argument");
    if (pl->f16_child) return qdas_plan_fallback_tiles(pl->f16_child, n);
    *n = 0;
    if (pl->kernel != QDAS_KERNEL_TILED || !pl->fallback) return QDAS_OK;
    uint32_t c = 0;
    HIPCHK(hipMemcpy(&c, pl->fallback, sizeof c, hipMemcpyDeviceToHost));
    *n = c;
    return QDAS_OK;
}

extern "C" int qdas_plan_tile_shape(const qdas_plan *pl, int *tile_z, int *tile_cols, int *wave_z, int *ksplit) {
    if (!pl || !tile_z || !tile_cols) return fail(QDAS_EINVAL, "null argument");
    if (pl->f16_child) return qdas_plan_tile_shape(pl->f16_child, tile_z, tile_cols, wave_z, ksplit);
    const bool tiled = pl->kernel == QDAS_KERNEL_TILED;
    *tile_z = tiled ? (1 << pl->tp.tz_log2) : 0;
    *tile_cols = tiled ? (int)pl->tile_cols : 0;
    if (wave_z) *wave_z = tiled ? (1 << pl->tp.wz_log2) : 0;
    if (ksplit) *ksplit = tiled ? (int)pl->tp.ksplit : 0;
    return QDAS_OK;
}

extern "C" int qdas_plan_mirror(const qdas_plan *pl) { if (pl && pl->f16_child) return qdas_plan_mirror(pl->f16_child); return pl && pl->kernel == QDAS_KERNEL_TILED && pl->tp.mir ? 1 : 0; }

extern "C" int qdas_plan_reciprocal(const qdas_plan *pl) { return pl && pl->kernel == QDAS_KERNEL_TILED && pl->tp.sym ? 1 : 0; }

extern "C" int qdas_plan_symmetry_bound(const qdas_plan *pl, double *mirror_samples, double *reciprocal_samples) {
    if (!pl) return fail(QDAS_EINVAL, "null plan");
    if (pl->f16_child) return qdas_plan_symmetry_bound(pl->f16_child, mirror_samples, reciprocal_samples);
    const bool tiled = pl->kernel == QDAS_KERNEL_TILED;
    if (mirror_samples) *mirror_samples = (tiled && pl->tp.mir) ? pl->mirror_bound : -1.0;
    if (reciprocal_samples) *reciprocal_samples = (tiled && pl->tp.sym) ? pl->recip_bound : -1.0;
    return QDAS_OK;
}

extern "C" int qdas_plan_folded(const qdas_plan *pl) { if (pl && pl->f16_child) return 1; return pl && pl->kernel == QDAS_KERNEL_TILED && pl->tp.fold ? 1 : 0; }

// ---- the reciprocity fold as an entry of its own (qdas.h): for hosts that fold once and hand folded frames to QDAS_PLAN_PREFOLDED plans
extern "C" int qdas_fold(const qdas_fold_desc *d, const void *x, void *xs, void *stream) {
    if (!d || !x || !xs) return fail(QDAS_EINVAL, "null argument");
    if (d->dtype != QDAS_F32 && d->dtype != QDAS_F16) return fail(QDAS_EINVAL, "qdas_fold: complex64 or complex32 (fp16) frames");
    if (d->N > 65535) return fail(QDAS_EUNSUPPORTED, "qdas_fold: at most 65535 elements");
    DeviceGuard guard(d->device);
    HIPCHK(guard.err);
    HIPCHK(launch_fold(x, xs, d->wtab, d->T, d->N, d->strN ? d->strN : d->T, d->strM ? d->strM : d->T * d->N, (hipStream_t)stream, d->dtype == QDAS_F16));
    return QDAS_OK;
}

extern "C" int qdas_plan_kernel_name(const qdas_plan *pl, char *buf, size_t len) {
    if (!pl || !buf || !len) return fail(QDAS_EINVAL, "null argument");
    if (pl->f16_child) {                                // "das_tile_kernel<interp=3,f16>f32,sym,fold,...": fp16 data on the folded fp32 kernels
        char tmp[256];
        int rc = qdas_plan_kernel_name(pl->f16_child, tmp, sizeof tmp);
        if (rc) return rc;
        std::string nm(tmp);
        const size_t at = nm.find(",f32");
        if (at != std::string::npos) nm.replace(at, 4, ",f16>f32");
        if (pl->fold_wtab && nm.find(",wtab") == std::string::npos) { const size_t f = nm.find(",fold"); if (f != std::string::npos) nm.insert(f + 5, ",wtab"); }
        snprintf(buf, len, "%s", nm.c_str());
        return QDAS_OK;
    }
    const qdas_sizes &z = pl->d.sz;
    const char *dts = z.dtype == QDAS_F64 ? "f64" : (z.dtype == QDAS_F32 ? "f32" : "f16");
    if (pl->kernel == QDAS_KERNEL_TILED) {
        const TileParams &t = pl->tp;
        snprintf(buf, len, "das_tile_kernel<interp=%d,%s%s%s%s%s%s%s,mb=%d,W=%d> [%s]", z.flag & 7, dts, t.sym ? ",sym" : "", t.fold ? ",fold" : "", t.fmod != 0.0 ? ",fmod" : "",
                 (t.wtab || (t.fold && pl->fold_wtab)) ? ",wtab" : "", t.big ? ",big" : "", t.mir ? ",mirror" : ((t.St && !t.syn) ? ",roles swapped" : ""), pl->jit_fn ? pl->jit_mb : pl->tc.mb, (pl->jit_fn && pl->jit_w) ? pl->jit_w : pl->tc.window, pl->jit_tag.empty() ? "prebuilt" : pl->jit_tag.c_str());
    } else snprintf(buf, len, "das_generic_kernel<interp=%d,%s>", z.flag & 7, dts);
    return QDAS_OK;
}

extern "C" int qdas_plan_set_timing(qdas_plan *pl, int enable) {
    if (!pl) return fail(QDAS_EINVAL, "null plan");
    pl->timing = enable != 0;
    if (pl->timing && !pl->e0) { HIPCHK(hipEventCreate(&pl->e0)); HIPCHK(hipEventCreate(&pl->e1)); }
    return QDAS_OK;
}

extern "C" int qdas_plan_last_kernel_ms(const qdas_plan *pl, float *ms) {
    if (!pl || !ms) return fail(QDAS_EINVAL, "null argument");
    *ms = pl->last_ms;
    return QDAS_OK;
}

static int hip_rc(hipError_t e) { HIPCHK(e); return QDAS_OK; }

// One frame -- or, for the tiled kernel outside the reciprocal mode, TWO frames in one launch (nf == 2): frame 1 lives at
// x + x_fstride bytes / y + y_fstride elements and shares tap indices and weights with frame 0 (das_tile_impl.h "FB2").
static int run_frame(qdas_plan *pl, const void *x, void *y, hipStream_t s, int nf = 1, uint64_t x_fstride = 0, uint64_t y_fstride = 0) {
    const qdas_sizes &z = pl->d.sz;
    if (pl->f16_child && nf == 1) {                     // fp16 reciprocal data: fold into complex64, the folded fp32 kernels, round the image to complex32
        const TileParams &c = pl->f16_child->tp;
        HIPCHK(launch_fold(x, pl->fold_buf, pl->fold_wtab, z.T, z.N, c.strN, c.strM, s, 1));
        int rc = run_frame(pl->f16_child, pl->fold_buf, pl->y32, s);
        if (rc) return rc;
        HIPCHK(launch_y32_to_y16(pl->y32, y, pl->f16_child->y_ld, s));
        return QDAS_OK;
    }
    if (pl->kernel == QDAS_KERNEL_TILED) {
        TileParams t = pl->tp;
        t.x = x; t.y = y;
        t.nfr = nf; t.x_fstride = x_fstride; t.y_fstride = y_fstride;
        if (t.fold && pl->prefolded) {                  // (the caller folded the frames: qdas_fold)
            if (!pl->no_fallback) return fail(QDAS_EUNSUPPORTED, "QDAS_PLAN_PREFOLDED: tiles of this image do not fit the staging windows and would need the unfolded frame");
        } else if (t.fold) {                            // the reciprocity fold of this frame (fold.hip): one pass over HBM, then the fused kernel on the folded copy
            HIPCHK(launch_fold(x, pl->fold_buf, pl->fold_wtab, z.T, z.N, t.strN, t.strM, s));
            t.x = pl->fold_buf;
            if (nf == 2) {                              // (two frames per launch: the second frame's folded copy; the kernel reads it x_fstride BYTES after the first)
                if (!pl->fold_buf2) return fail(QDAS_EINVAL, "internal: no second fold buffer");
                HIPCHK(launch_fold((const char *)x + x_fstride, pl->fold_buf2, pl->fold_wtab, z.T, z.N, t.strN, t.strM, s));
                t.x_fstride = (uint64_t)((intptr_t)pl->fold_buf2 - (intptr_t)pl->fold_buf);      // (modulo 2^64: the kernel adds it to the base pointer)
            }
        }
        if (t.syn) {                                    // planes are accumulated with atomics: start from zero
            const size_t ds = data_size(z.dtype);        // (only this plan's pixels of every plane: y_ld may span a full-size buffer)
            for (int f = 0; f < nf; ++f)
                HIPCHK(hipMemset2DAsync((char *)y + (size_t)f * y_fstride * ds, (size_t)pl->y_ld * ds, 0, (size_t)pl->i_count * ds, pl->oN * pl->oM, s));
        }
        // (frames that share a launch run the prebuilt two- / four-frame kernels also in plans with a hiprtc build: shared tap indices and
        //  weights are worth more than the specialisation -- C2: 1.49 ms per frame in pairs, 1.64 ms one by one on the hiprtc build)
        const hipFunction_t jf = nf == 1 ? pl->jit_fn : nullptr;
        if (pl->no_fallback) return hip_rc(launch_tile(t, z.dtype, pl->ntiles, s, jf, pl->jit_lds));     // one launch per frame (+ the reduce of a split aperture)
        HIPCHK(hipMemsetAsync(pl->fallback, 0, sizeof(uint32_t), s));
        HIPCHK(launch_tile(t, z.dtype, pl->ntiles, s, jf, pl->jit_lds));
        // tiles whose delay window overflowed LDS are redone by the generic kernel; the launch is
        // sized for the worst case and exits immediately for ids >= the device-side count
        GenericParams g = pl->gp;
        g.x = x; g.y = y; g.y_ld = pl->y_ld;
        g.tile_list = pl->fallback;
        g.tile_cols = pl->tile_cols;
        g.tile_zl = (uint32_t)pl->tp.tz_log2;
        g.blocks_per_tile = (64 * pl->tc.waves + 255) / 256;
        g.tiles_z = pl->tp.tiles_z;
        HIPCHK(launch_generic(g, z.dtype, pl->ntiles * g.blocks_per_tile, s));
        for (int f = 1; f < nf; ++f) {                  // the same misfit tiles of the other frames of the launch
            g.x = (const char *)x + (size_t)f * x_fstride; g.y = (char *)y + (size_t)f * y_fstride * data_size(z.dtype);
            HIPCHK(launch_generic(g, z.dtype, pl->ntiles * g.blocks_per_tile, s));
        }
    } else {
        GenericParams g = pl->gp;
        g.x = x; g.y = y;
        HIPCHK(launch_generic(g, z.dtype, (unsigned)((pl->i_count + 255) / 256), s));
    }
    return QDAS_OK;
}

// One-time preparations of a STREAM of F frames (twin plan, second folded copy, frame-sharing instantiations, host staging): everything that allocates,
// compiles or synchronises.  qdas_plan_execute_frames runs it at the plan's first stream; qdas_plan_prepare_frames lets a caller run it ahead of time, so that
// the stream calls themselves only enqueue work (graph capture, latency-critical first frames).  Idempotent.
static int prepare_stream(qdas_plan *pl, uint64_t F, hipStream_t s) {
    const qdas_sizes &z = pl->d.sz;
    if (pl->I == 0 || z.N == 0 || z.M == 0 || pl->i_count == 0 || z.T == 0 || F < 2) return QDAS_OK;
    // a stream of >= 4 device-resident frames through a general-mode lateral-mirror plan: four frames per launch on the plan's twin
    if (F >= 4 && pl->kernel == QDAS_KERNEL_TILED && pl->tp.mir == 1 && !pl->tp.sym && !pl->tp.apix && !pl->tp.gen_kind && pl->d.mem == QDAS_MEM_DEVICE
        && !getenv("QDAS_NO_FB4") && !getenv("QDAS_NO_FB2") && !getenv("QDAS_NO_FRAMES_TWIN") && !pl->twin_tried) {
        pl->twin_tried = true;
        qdas_desc d = pl->d;                            // (the plan's own device copies / the caller's device arrays: both stay valid for the plan's life)
        const GenericParams &g = pl->gp;
        uint64_t acs[6 * (1 + QDAS_MAX_APOD)];
        memcpy(acs, g.cst, sizeof g.cst); memcpy(acs + 6, g.ast, sizeof(uint64_t) * 6 * z.S);
        d.Pi = g.Pi; d.Pr = g.Pr; d.Pv = g.Pv; d.Nv = g.Nv; d.cinv = g.cinv; d.apod = g.apod; d.rx_normals = g.rxn; d.acstride = acs;
        d.mem = QDAS_MEM_DEVICE; d.device = pl->device;
        d.plan_flags = (d.plan_flags | QDAS_PLAN_NO_MIRROR) & ~(QDAS_PLAN_COPY_INPUTS | QDAS_PLAN_JIT | QDAS_PLAN_MIRROR_SLAB);
        const std::string keep = g_err;
        if (qdas_plan_create(&pl->frames_twin, &d) != QDAS_OK) pl->frames_twin = nullptr;
        else if (!(pl->frames_twin->fb2_ok && !pl->frames_twin->fb4_off)) { qdas_plan_destroy(pl->frames_twin); pl->frames_twin = nullptr; }
        g_err = keep;
    }
    if (pl->frames_twin && F >= 4) return prepare_stream(pl->frames_twin, F, s);
    // folded data: frame pairs share a launch; the second folded copy is made now (once)
    if (pl->fold2_ok && !pl->prefolded && !pl->fold_buf2) {
        const size_t fb = (size_t)z.T * z.N * z.M * 8;
        void *p2 = nullptr;
        if (hipMalloc(&p2, fb) == hipSuccess) {
            pl->owned.push_back(p2);
            if (hipMemsetAsync(p2, 0, fb, s) == hipSuccess) pl->fold_buf2 = p2;
        } else (void)hipGetLastError();
        if (!pl->fold_buf2) pl->fold2_ok = false;       // (no memory for it: one frame per launch)
    }
    // the frame-sharing instantiations are resolved (built on demand when libqdas.so does not carry them) here, not inside the frame loop; without a
    // compiler the stream runs one frame per launch
    if ((pl->fb2_ok || pl->fold2_ok) && pl->kernel == QDAS_KERNEL_TILED) {
        const std::string keep = g_err;
        for (int nf = 2; nf <= 4; nf += 2) {
            bool &done = nf == 2 ? pl->prep2 : pl->prep4;
            if (done || (nf == 4 && (F < 4 || pl->fb4_off || pl->tp.fold))) continue;
            done = true;
            TileParams t = pl->tp;
            t.nfr = nf;
            if (prepare_tile(t, z.dtype, pl->ntiles, nullptr) == hipErrorSharedObjectInitFailed) {
                if (nf == 2) { pl->fb2_ok = false; pl->fold2_ok = false; } else pl->fb4_off = true;
            }
            (void)hipGetLastError();
        }
        g_err = keep;
    }
    if (pl->d.mem == QDAS_MEM_HOST && !pl->copy_stream) {      // host frames: a second staging set + copy stream (frame f+1 is uploaded while f is beamformed)
        int rc;
        if ((rc = dev_alloc(pl, &pl->dx2, pl->x_bytes)) || (rc = dev_alloc(pl, &pl->dy2, pl->y_bytes))) return rc;
        HIPCHK(hipStreamCreateWithFlags(&pl->copy_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            HIPCHK(hipEventCreateWithFlags(&pl->ex[k], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&pl->ek[k], hipEventDisableTiming));
        }
    }
    return QDAS_OK;
}

extern "C" int qdas_plan_prepare_frames(qdas_plan *pl, uint64_t F) {
    if (!pl) return fail(QDAS_EINVAL, "null plan");
    DeviceGuard guard(pl->device);
    HIPCHK(guard.err);
    int rc = prepare_stream(pl, F, nullptr);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(nullptr));
    return QDAS_OK;
}

// host-resident frames: upload f+1 on the copy stream while f is beamformed
static int stream_host_frames(qdas_plan *pl, const void *x, void *y, uint64_t F, uint64_t x_stride, uint64_t y_stride, hipStream_t s) {
    const size_t ds = data_size(pl->d.sz.dtype);
    void *dxs[2] = {pl->dx, pl->dx2}, *dys[2] = {pl->dy, pl->dy2};
    hipStream_t sc = pl->copy_stream;
    HIPCHK(hipMemcpyAsync(dxs[0], x, pl->x_bytes, hipMemcpyHostToDevice, sc));
    HIPCHK(hipEventRecord(pl->ex[0], sc));
    for (uint64_t f = 0; f < F; ++f) {
        const int b = (int)(f & 1), nb = b ^ 1;
        HIPCHK(hipStreamWaitEvent(s, pl->ex[b], 0));
        int rc = run_frame(pl, dxs[b], dys[b], s);
        if (rc) return rc;
        HIPCHK(hipEventRecord(pl->ek[b], s));
        if (f + 1 < F) {                                // the other buffer's last reader was frame f-1
            if (f >= 1) HIPCHK(hipStreamWaitEvent(sc, pl->ek[nb], 0));
            HIPCHK(hipMemcpyAsync(dxs[nb], (const char *)x + (f + 1) * x_stride * ds, pl->x_bytes, hipMemcpyHostToDevice, sc));
            HIPCHK(hipEventRecord(pl->ex[nb], sc));
        }
        HIPCHK(hipMemcpyAsync((char *)y + f * y_stride * ds, dys[b], pl->y_bytes, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipStreamSynchronize(sc));
    return QDAS_OK;
}

static int stop_timer(qdas_plan *pl, hipStream_t s) {
    if (!pl->timing) return QDAS_OK;
    HIPCHK(hipEventRecord(pl->e1, s));
    HIPCHK(hipEventSynchronize(pl->e1));
    HIPCHK(hipEventElapsedTime(&pl->last_ms, pl->e0, pl->e1));
    return QDAS_OK;
}

extern "C" int qdas_plan_execute_frames(qdas_plan *pl, const void *x, void *y, uint64_t F, uint64_t x_stride,
                                        uint64_t y_stride, void *stream) {
    if (!pl) return fail(QDAS_EINVAL, "null plan");
    if (F == 0) return QDAS_OK;
    if (!y) return fail(QDAS_EINVAL, "null output");
    const qdas_sizes &z = pl->d.sz;
    const size_t ds = data_size(z.dtype);
    hipStream_t s = (hipStream_t)stream;
    DeviceGuard guard(pl->device);
    HIPCHK(guard.err);
    const size_t ybytes = (size_t)pl->y_ld * pl->oN * pl->oM * ds;
    if (pl->I == 0 || z.N == 0 || z.M == 0 || pl->i_count == 0 || z.T == 0) {      // empty sum: zeros
        for (uint64_t f = 0; f < F && ybytes; ++f) {
            char *yf = (char *)y + f * y_stride * ds;
            if (pl->d.mem == QDAS_MEM_HOST) memset(yf, 0, ybytes);
            else HIPCHK(hipMemsetAsync(yf, 0, ybytes, s));
        }
        return QDAS_OK;
    }
    if (!x) return fail(QDAS_EINVAL, "null data");
    if (pl->timing) HIPCHK(hipEventRecord(pl->e0, s));
    int rc = prepare_stream(pl, F, s);                  // (a no-op after the plan's first stream, or after qdas_plan_prepare_frames)
    if (rc) return rc;
    if (F >= 4 && pl->frames_twin) {
        // the WHOLE stream goes through the twin (groups of four, then a pair, then a single frame on its own kernels): every frame of
        // one call is summed in the same order -- round 3 sent the tail through the mirror kernel.  The images equal
        // qdas_plan_execute's to fp32 re-association, not bit for bit (another kernel): include/qdas.h says so.
        pl->frames_twin->timing = false;
        rc = qdas_plan_execute_frames(pl->frames_twin, x, y, F, x_stride, y_stride, stream);
        return rc ? rc : stop_timer(pl, s);
    }
    // frame pairs share one launch (device-resident data, tiled kernel; QDAS_NO_FB2 disables); folded data: the caller's folded frames as they lie
    // (prefolded plans), or the plan's two folded copies
    bool pairs_ok = pl->fb2_ok && x_stride * ds < (1ull << 40);
    if (pl->fold2_ok && F >= 2) pairs_ok = pl->prefolded ? x_stride * ds < (1ull << 40) : pl->fold_buf2 != nullptr;
    if (pl->d.mem == QDAS_MEM_HOST && F >= 2) {
        rc = stream_host_frames(pl, x, y, F, x_stride, y_stride, s);
        return rc ? rc : stop_timer(pl, s);
    }
    for (uint64_t f = 0; f < F; ++f) {
        const char *xf = (const char *)x + f * x_stride * ds;
        char *yf = (char *)y + f * y_stride * ds;
        if (pairs_ok && f + 1 < F) {                    // four (else two) frames share a launch
            const int nf = (f + 3 < F && !pl->fb4_off && !pl->tp.fold) ? 4 : 2;
            rc = run_frame(pl, xf, yf, s, nf, x_stride * ds, y_stride);
            if (rc) return rc;
            f += nf - 1;
            continue;
        }
        if (pl->d.mem == QDAS_MEM_HOST) {
            HIPCHK(hipMemcpyAsync(pl->dx, xf, pl->x_bytes, hipMemcpyHostToDevice, s));
            rc = run_frame(pl, pl->dx, pl->dy, s);
            if (rc) return rc;
            HIPCHK(hipMemcpyAsync(yf, pl->dy, pl->y_bytes, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
        } else {
            rc = run_frame(pl, xf, yf, s);
            if (rc) return rc;
        }
    }
    return stop_timer(pl, s);
}

extern "C" int qdas_plan_execute(qdas_plan *pl, const void *x, void *y, void *stream) {
    return qdas_plan_execute_frames(pl, x, y, 1, 0, 0, stream);
}

extern "C" int qdas_plan_delays(qdas_plan *pl, void *tau, void *stream) {
    if (!pl || !tau) return fail(QDAS_EINVAL, "null argument");
    const qdas_sizes &z = pl->d.sz;
    if (pl->I == 0 || z.N == 0 || z.M == 0 || pl->i_count == 0) return QDAS_OK;
    DeviceGuard guard(pl->device);
    HIPCHK(guard.err);
    hipStream_t s = (hipStream_t)stream;
    const size_t rs = real_size(z.dtype);
    const double cinv = pl->cinv0;                     // (read at plan creation: the caller's arrays are not touched after it)
    GenericParams g = pl->gp;
    g.y_ld = pl->i_count;
    const size_t bytes = (size_t)pl->i_count * z.N * z.M * rs;
    if (pl->d.mem == QDAS_MEM_HOST) {
        void *dt;
        HIPCHK(hipMalloc(&dt, bytes));
        hipError_t e = launch_delays(g, z.dtype == QDAS_F64 ? 0 : 1, dt, cinv, s);
        if (e == hipSuccess) e = hipMemcpyAsync(tau, dt, bytes, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        (void)hipFree(dt);
        if (e != hipSuccess) return fail(QDAS_EHIP, "delays: %s", hipGetErrorString(e));
    } else {
        HIPCHK(launch_delays(g, z.dtype == QDAS_F64 ? 0 : 1, tau, cinv, s));
    }
    return QDAS_OK;
}

// ------------------------------------------------------------------------------------ one-shot entries
static int one_shot(const qdas_sizes *sz, int dtype, void *y, const void *Pi, const void *Pr, const void *Pv, const void *Nv,
                    const void *a, const void *cinv, const uint64_t *acs, const void *x, double fs, double fmod, void *stream) {
    if (!sz) return fail(QDAS_EINVAL, "null sizes");
    qdas_desc d{};
    d.sz = *sz;
    d.sz.dtype = dtype;
    d.fs = fs; d.fmod = fmod;
    d.Pi = Pi; d.Pr = Pr; d.Pv = Pv; d.Nv = Nv; d.apod = a; d.cinv = cinv; d.acstride = acs;
    d.mem = QDAS_MEM_DEVICE; d.device = -1; d.kernel = QDAS_KERNEL_AUTO;
    qdas_plan *pl = nullptr;
    int rc = qdas_plan_create(&pl, &d);
    if (rc) return rc;
    rc = qdas_plan_execute(pl, x, y, stream);
    if (!rc) { hipError_t e = hipStreamSynchronize((hipStream_t)stream); if (e != hipSuccess) rc = fail(QDAS_EHIP, "sync: %s", hipGetErrorString(e)); }
    qdas_plan_destroy(pl);
    return rc;
}

extern "C" int qdas_DAS(const qdas_sizes *sz, void *y, const double *Pi, const double *Pr, const double *Pv, const double *Nv,
                        const void *a, const double *cinv, const uint64_t *acs, const void *x, const double tv[2], void *stream) {
    if (!tv) return fail(QDAS_EINVAL, "null tvars");
    return one_shot(sz, QDAS_F64, y, Pi, Pr, Pv, Nv, a, cinv, acs, x, tv[0], tv[1], stream);
}
extern "C" int qdas_DASf(const qdas_sizes *sz, void *y, const float *Pi, const float *Pr, const float *Pv, const float *Nv,
                         const void *a, const float *cinv, const uint64_t *acs, const void *x, const float tv[2], void *stream) {
    if (!tv) return fail(QDAS_EINVAL, "null tvars");
    return one_shot(sz, QDAS_F32, y, Pi, Pr, Pv, Nv, a, cinv, acs, x, tv[0], tv[1], stream);
}
extern "C" int qdas_DASh(const qdas_sizes *sz, void *y, const float *Pi, const float *Pr, const float *Pv, const float *Nv,
                         const void *a, const float *cinv, const uint64_t *acs, const void *x, const float tv[2], void *stream) {
    if (!tv) return fail(QDAS_EINVAL, "null tvars");
    return one_shot(sz, QDAS_F16, y, Pi, Pr, Pv, Nv, a, cinv, acs, x, tv[0], tv[1], stream);
}

static int delays_one_shot(const qdas_sizes *sz, int dtype, void *tau, const void *Pi, const void *Pr, const void *Pv,
                           const void *Nv, double cinv, void *stream) {
    if (!sz || !tau) return fail(QDAS_EINVAL, "null argument");
    GenericParams g{};
    g.Pi = Pi; g.Pr = Pr; g.Pv = Pv; g.Nv = Nv;
    g.N = sz->N; g.M = sz->M; g.I1 = sz->I1; g.I2 = sz->I2; g.I3 = sz->I3;
    g.i_begin = 0; g.i_count = sz->I1 * sz->I2 * sz->I3; g.y_ld = g.i_count;
    g.VS = sz->VS; g.DV = sz->DV;
    if (g.i_count == 0 || g.N == 0 || g.M == 0) return QDAS_OK;
    if (!Pi || !Pr || !Pv || !Nv) return fail(QDAS_EINVAL, "null geometry pointer");
    HIPCHK(launch_delays(g, dtype, tau, cinv, (hipStream_t)stream));
    return QDAS_OK;
}
extern "C" int qdas_delays(const qdas_sizes *sz, double *tau, const double *Pi, const double *Pr, const double *Pv,
                           const double *Nv, double cinv, void *stream) {
    return delays_one_shot(sz, 0, tau, Pi, Pr, Pv, Nv, cinv, stream);
}
extern "C" int qdas_delaysf(const qdas_sizes *sz, float *tau, const float *Pi, const float *Pr, const float *Pv,
                            const float *Nv, float cinv, void *stream) {
    return delays_one_shot(sz, 1, tau, Pi, Pr, Pv, Nv, cinv, stream);
}

// Split-delay tables in lateral-mirror mode: tau[i', E-1-e] == tau[i, e] bit for bit for both tables (i' = the pixel of column I2-1-c in the same row)?  The tables
// are data of the call, so the test runs per call (one pass over both tables: 0.27 GB at BASELINE C2's size), ahead of the probe that is read back anyway.
__global__ void __launch_bounds__(256) lut_mirror_check_kernel(const uint32_t *ta, uint64_t Ea, const uint32_t *tb, uint64_t Eb, uint64_t I1, uint64_t I2, uint32_t *bad) {
    // blockIdx.y: element e of table a (e < Ea) or b; blockIdx.x / threads: the pixels of the first half of the columns, four rows of depth per lane when I1 allows
    const uint64_t I = I1 * I2, half = (I2 + 1) / 2;
    const uint64_t e = blockIdx.y;
    const uint32_t *t = e < Ea ? ta : tb;
    const uint64_t E = e < Ea ? Ea : Eb, f = e < Ea ? e : e - Ea;
    const uint32_t *p = t + I * f, *q = t + I * (E - 1 - f);
    bool same = true;
    if ((I1 & 3) == 0 && (((uintptr_t)t) & 15) == 0) {
        const uint64_t r4 = I1 / 4, n4 = r4 * half;
        for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n4; k += (uint64_t)gridDim.x * blockDim.x) {
            const uint64_t c = k / r4, r = k - c * r4;
            const uint4 a = ((const uint4 *)(p + I1 * c))[r], b = ((const uint4 *)(q + I1 * (I2 - 1 - c)))[r];
            if (a.x != b.x || a.y != b.y || a.z != b.z || a.w != b.w) same = false;
        }
    } else {
        const uint64_t n = I1 * half;
        for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
            const uint64_t c = k / I1, r = k - c * I1;
            if (p[r + I1 * c] != q[r + I1 * (I2 - 1 - c)]) same = false;
        }
    }
    if (!same) *bad = 1u;
}

// what the last call with these very arguments found (footprint; tables their own mirror images): the next call launches the kernel straight away and CHECKS
// afterwards -- the kernel's own prologue counts the tiles that do not fit, the symmetry test runs ahead of it on the stream -- instead of probing first
namespace {
struct LutMemo { uint64_t I, I1, N, M, T; const void *rx, *tx; int flag, dtype, level; bool fm; };
std::mutex g_lut_memo_mu;
std::vector<LutMemo> g_lut_memo;
bool lut_memo_same(const LutMemo &a, const LutMemo &b) { return a.I == b.I && a.I1 == b.I1 && a.N == b.N && a.M == b.M && a.T == b.T && a.rx == b.rx && a.tx == b.tx && a.flag == b.flag && a.dtype == b.dtype && a.fm == b.fm; }
int lut_memo_get(const LutMemo &k) { std::lock_guard<std::mutex> lk(g_lut_memo_mu); for (const auto &m : g_lut_memo) if (lut_memo_same(m, k)) return m.level; return -1; }
void lut_memo_put(const LutMemo &k, int level) {      // level < 0: forget
    std::lock_guard<std::mutex> lk(g_lut_memo_mu);
    for (size_t i = 0; i < g_lut_memo.size(); ++i) if (lut_memo_same(g_lut_memo[i], k)) { g_lut_memo.erase(g_lut_memo.begin() + (long)i); break; }
    if (level >= 0) { LutMemo m = k; m.level = level; g_lut_memo.push_back(m); if (g_lut_memo.size() > 16) g_lut_memo.erase(g_lut_memo.begin()); }
}
}  // namespace

// ------------------------------------------------------------------------------------ split-delay flavour
// The split-delay flavour through the tiled kernel (fp32 / fp16 data).  Returns -1 when the launch was made, +1 when the problem has to run on
// das_lut_kernel (double precision / kept dimensions / pixel-dependent weights / a tile whose delay spread does not fit the
// LDS window for any footprint / QDAS_LUT_GENERIC=1), 0 on a HIP error.
static thread_local std::string g_lut_last;          // the kernel of this thread's last qdas_das_lut (qdas_das_lut_last_kernel)
extern "C" int qdas_das_lut_last_kernel(char *buf, size_t len) {
    if (!buf || !len) return fail(QDAS_EINVAL, "null argument");
    snprintf(buf, len, "%s", g_lut_last.c_str());
    return QDAS_OK;
}
static int lut_tiled(const qdas_lut_desc *d, const void *x, void *y, hipStream_t s) {
    const int dt = d->dtype;
    const bool keep_rx = d->flag & QDAS_FLAG_KEEP_RX, keep_tx = d->flag & QDAS_FLAG_KEEP_TX;
    if ((dt != QDAS_F32 && dt != QDAS_F16) || (keep_rx && keep_tx) || getenv("QDAS_LUT_GENERIC")) return 1;
    // one kept dimension (fp32; weights: none, or a pixel x receiver array with the receive dimension kept): the kernel's 'SYN' mode -- one output plane per STAGE element; keeping the transmit
    // dimension swaps the roles of the two tables (as 'MUL' does for geometry-driven plans)
    const bool keep = keep_rx || keep_tx;
    // weights: none; one complex fp32 N x M table (wstride {0, 1, N}); or -- the usual receive apodization -- an I x N array in the
    // data precision, real or complex (wstride {1, I, 0}), which the kernel applies per stage like the plans' pixel x receiver arrays
    const bool w_tab = d->w && d->wstride[0] == 0 && !d->w_real && dt == QDAS_F32 && d->wstride[1] == 1 && d->wstride[2] == d->N;
    const bool w_pix = d->w && !w_tab && d->wstride[0] == 1 && d->wstride[1] == d->I && d->wstride[2] == 0 && d->N > 1;
    // ... or an I x M array (a weight per pixel and TRANSMIT: scanline-style transmit apodization): the same with the roles of the two
    // tables swapped (full sum only)
    const bool w_pixm = d->w && !w_tab && !w_pix && !keep && d->wstride[0] == 1 && d->wstride[1] == 0 && d->wstride[2] == d->I && d->M > 1;
    if (d->w && !w_tab && !w_pix && !w_pixm) return 1;
    if (keep && (dt != QDAS_F32 || w_tab || (w_pix && keep_tx))) return 1;
    if (d->T < 8 || d->N >= (1ull << 20) || d->M >= (1ull << 20)) return 1;
    if (tile_lds_bytes(dt, 0, d->N > d->M ? d->N : d->M, d->N > d->M ? d->N : d->M, 0, (w_pix || w_pixm) ? 1 : 0, w_tab ? 1 : 0) > tile_lds_limit(0)) return 1;
    const bool tp = d->flag & QDAS_FLAG_TPOSE;
    uint64_t strN = tp ? d->T * d->M : d->T, strM = tp ? d->T : d->T * d->N;
    uint64_t kN = d->N, kM = d->M;
    const void *tab_s = d->tau_rx, *tab_b = d->tau_tx;                // stage / block tables
    const TileConfig tc = tile_config(dt, 0);
    // roles of the two tables: stage = receive table unless the transmit dimension is kept, the weights are per (pixel, transmit), or -- full
    // sum without pixel weights -- swapping leaves at least a quarter fewer stages (few transmits: qdas_plan_create has the same rule)
    bool swap = keep_tx || w_pixm;
    if (!keep && !w_pix && !w_pixm && !w_tab && !getenv("QDAS_NO_ROLE_SWAP")) {      // (an N x M weight table is laid out for the usual roles)
        const uint64_t mb = (uint64_t)tc.mb;
        if (4 * d->M * ((d->N + mb - 1) / mb) < 3 * d->N * ((d->M + mb - 1) / mb)) swap = true;
    }
    if (swap) { std::swap(strN, strM); std::swap(kN, kM); std::swap(tab_s, tab_b); }
    if ((kN * strN + (uint64_t)tc.mb * strM) * data_size(dt) + 65536 >= (1ull << 31)) return 1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    // misfit counter of THIS call, from the arena of the caller's stream (one call at a time per stream holds it): concurrent calls (other host
    // threads, other streams of the device) never share it, so no probe result can be cleared or read by another call
    Scratch scratch(s);                                  // (this stream's arena, csrc/scratch.hip)
    uint32_t *counter = (uint32_t *)scratch.get(64);
    if (!counter) return 1;
    const bool shaped = d->I1 && d->I1 < d->I && d->I % d->I1 == 0;    // (a per-pixel array needs the true image shape: no ragged rows)
    if ((w_pix || w_pixm) && !shaped && d->I1 != d->I) return 1;
    TileParams t{};
    t.x = x; t.y = y; t.wtab = w_tab ? d->w : nullptr;
    if (w_pix || w_pixm) { t.apix = d->w; t.apix_real = d->w_real; }
    t.T = d->T; t.N = kN; t.M = kM;
    t.act_bytes = t.apix ? (uint32_t)(8 * (t.N + 1)) : 0u;
    t.syn = keep ? 1 : 0;
    const uint64_t I1 = shaped ? d->I1 : (d->I1 >= d->I ? d->I : 64);
    t.I1 = I1; t.I2 = (d->I + I1 - 1) / I1; t.I3 = 1;
    t.i_begin = 0; t.i_count = d->I; t.y_ld = d->I;
    t.strN = strN; t.strM = strM;
    t.fs = 1.0; t.cinv_fs = 0.0; t.fmod = d->omega / 6.283185307179586476925;
    t.flag = d->flag & (7 | QDAS_FLAG_TPOSE);
    t.nfr = 1; t.ksplit = 1;
    t.lut_tx = (const float *)tab_b; t.lut_rx = (const float *)tab_s;
    t.fallback_list = counter; t.fallback_cap = 0;
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    // ---- lateral-mirror mode (round 6; VERDICT r5 item 5: `bfDAS` within 1.25x of `DAS`).  Tables whose mirror images are their own -- a centred scan under a symmetric
    // probe and sequence: what bfDASLUT computes for every BASELINE configuration -- let a pixel and its mirror image share tap index and weights exactly as geometry-
    // driven plans do.  Checked per call, bit for bit, on the device (one pass over both tables, read back with the probe's counter); the kernel is the
    // plan-specialised two-window-set build with 32-transmit stages of 128-sample windows, so it exists when hiprtc does and when every tile of some footprint fits those
    // windows; anything else falls through to the general table-driven kernel below.  fp32 data, full sum, no weights.
    if (dt == QDAS_F32 && !keep && !d->w && shaped && t.I2 >= 2 && kM >= 32 && !getenv("QDAS_LUT_NO_MIRROR") && !getenv("QDAS_NO_MIRROR") && !getenv("QDAS_NO_JIT")) {
        uint32_t hc[2] = {1u, 1u};
        int found = -1;
        unsigned nt2 = 0;
        const uint64_t halfc = (t.I2 + 1) / 2;
        const LutMemo mk{d->I, t.I1, d->N, d->M, d->T, d->tau_rx, d->tau_tx, d->flag, dt, -1, t.fmod != 0.0};
        const int memo = lut_memo_get(mk);
        auto set_grid = [&](int l) {
            t.tz_log2 = l; t.wz_log2 = 3;
            const unsigned cols = ((unsigned)tc.waves * 64u) >> l;
            t.tiles_z = (uint32_t)((t.I1 + (1u << l) - 1) >> l);
            t.tile_x0 = 0;
            t.tiles_x = (uint32_t)((halfc + cols - 1) / cols);      // (the tiles of the first half of the columns: their mirror images have the same delays)
            nt2 = t.tiles_z * t.tiles_x;
        };
        bool ok = hipMemsetAsync(counter, 0, 2 * sizeof(uint32_t), s) == hipSuccess;
        if (ok) {
            const uint64_t rows = (t.I1 & 3) == 0 ? t.I1 / 4 : t.I1;
            const unsigned gx = (unsigned)std::min<uint64_t>((rows * halfc + 255) / 256, 1024);
            lut_mirror_check_kernel<<<dim3(gx, (unsigned)(kN + kM)), 256, 0, s>>>((const uint32_t *)tab_s, kN, (const uint32_t *)tab_b, kM, t.I1, t.I2, counter + 1);
            ok = hipGetLastError() == hipSuccess;
        }
        // the launch of a resolved footprint: `verify` = no probe went before it (a remembered footprint): the kernel's own fit test and the symmetry flag are read afterwards
        auto run_mirror = [&](int l, bool verify) -> int {      // -1 launched and good, 0 HIP error, 1 not taken
            set_grid(l);
            JitSpec k{};
            k.interp = (d->flag & 7) == 4 ? 1 : (d->flag & 7); k.dtype = dt; k.fmod = t.fmod != 0.0; k.lut = 1; k.mir = 1;
            k.waves = tc.waves; k.mb = 32; k.w = 128; k.nbuf = 2;
            k.N = kN; k.M = kM; k.T = d->T; k.I1 = t.I1; k.strN = strN; k.strM = strM; k.tzl = l; k.wzl = t.wz_log2;
            unsigned ks2 = 1;
            while (ks2 * 2 <= (unsigned)std::min<uint64_t>(8, kN) && (uint64_t)nt2 * ks2 < (uint64_t)ncu) ks2 *= 2;
            void *part2 = ks2 > 1 ? scratch.get(sizeof(float) * 2 * (size_t)ks2 * d->I) : nullptr;
            if (ks2 > 1 && !part2) ks2 = 1;
            k.ksplit = ks2;
            const size_t lds = jit_tile_lds(k, kN, kM, 0, false);
            hipFunction_t fn = nullptr;
            std::string key;
            const std::string keep_err = g_err;
            if (lds > (size_t)160 * 1024) return 1;
            // (the resolved kernel of a spec is remembered: jit_get_kernel hashes the source AND every embedded header per call -- 0.1 ms, twice a C1-sized call's kernel)
            {
                static std::mutex mu;
                struct Hit { JitSpec k; int dev; hipFunction_t fn; std::string key; };
                static std::vector<Hit> hits;
                bool have = false;
                { std::lock_guard<std::mutex> lk(mu); for (const Hit &h : hits) if (h.dev == dev && !memcmp(&h.k, &k, sizeof k)) { fn = h.fn; key = h.key; have = true; break; } }
                if (have && !fn) return 1;               // (remembered: no compiler, or a build that would spill -- the general flavour, without asking again)
                if (!have) {
                    const JitSpec asked = k;
                    const bool failed = !jit_get_kernel_nospill(k, dev, &fn, &key).empty();      // (no compiler: not an error)
                    if (failed) { fn = nullptr; g_err = keep_err; (void)hipGetLastError(); }
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        hits.push_back(Hit{asked, dev, fn, key});
                        if (hits.size() > 64) hits.erase(hits.begin());
                    }
                    if (failed) return 1;
                }
            }
            t.probe = 0; t.probe_w = 0; t.mir = 1; t.ksplit = ks2; t.part = (float2 *)part2;
            if (hipMemsetAsync(counter, 0, sizeof(uint32_t), s) != hipSuccess) return 0;
            if (launch_tile(t, dt, nt2, s, fn, lds) != hipSuccess) return 0;
            if (verify) {
                if (hipMemcpyAsync(hc, counter, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return 0;
                if (hc[0] != 0 || hc[1] != 0) return 1;          // the tables changed under the same pointers: redone below, every pixel rewritten
            }
            g_lut_last = "tiled,mirror,mb=32,W=128 [jit " + key + "]";
            return -1;
        };
        if (ok && memo >= 3 && memo <= 6) {
            const int r = run_mirror(memo, true);
            if (r <= 0) return r;
            lut_memo_put(mk, -1);
            ok = hipMemsetAsync(counter + 1, 0, sizeof(uint32_t), s) == hipSuccess;      // (the flag stays valid for the loop below only if it was clean)
            if (ok && hc[1] != 0) ok = false;
        }
        for (int l = 6; ok && l >= 3 && found < 0; --l) {
            set_grid(l);
            t.probe = 1; t.probe_w = 128; t.mir = 0; t.ksplit = 1; t.part = nullptr;
            ok = hipMemsetAsync(counter, 0, sizeof(uint32_t), s) == hipSuccess && launch_tile(t, dt, nt2, s) == hipSuccess
                 && hipMemcpyAsync(hc, counter, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
            if (!ok || hc[1] != 0) break;                           // (tables that are not their own mirror images: the general kernel)
            if (hc[0] == 0) found = l;
        }
        t.probe = 0; t.probe_w = 0;
        if (!ok) (void)hipGetLastError();
        if (ok && found >= 0 && hc[1] == 0) {
            const int r = run_mirror(found, false);
            if (r <= 0) { if (r < 0) lut_memo_put(mk, found); return r; }
        }
        t.mir = 0; t.ksplit = 1; t.part = nullptr;
    }
    // footprint: the deepest tile (of 64, 32, 16, 8 pixels of I1) whose delay spreads all fit the window; the tables are data
    // of this call, so the fit is probed per call (prologue-only launches)
    int best = -1;
    unsigned ntiles = 0;
    for (int l = 6; l >= 3 && best < 0; --l) {
        t.tz_log2 = l; t.wz_log2 = l < 3 ? l : 3;
        const unsigned cols = ((unsigned)tc.waves * 64u) >> l;
        t.tiles_z = (uint32_t)((t.I1 + (1u << l) - 1) >> l);
        t.tile_x0 = 0;
        t.tiles_x = (uint32_t)((t.I2 + cols - 1) / cols);
        ntiles = t.tiles_z * t.tiles_x;
        t.probe = 1;
        if (hipMemsetAsync(counter, 0, sizeof(uint32_t), s) != hipSuccess) return 0;
        if (launch_tile(t, dt, ntiles, s) != hipSuccess) return 0;
        uint32_t cnt = 1;
        if (hipMemcpyAsync(&cnt, counter, sizeof(uint32_t), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return 0;
        if (cnt == 0) best = l;
    }
    if (best < 0) return 1;
    t.probe = 0;
    // too few tiles for the GPU: several workgroups per tile, each summing a range of receivers (as plans do)
    unsigned ks = 1;
    const unsigned cap = (unsigned)std::min<uint64_t>(8, kN);
    while (ks * 2 <= cap && (uint64_t)ntiles * ks < (uint64_t)ncu) ks *= 2;
    void *part = nullptr;
    if (ks > 1) {
        part = scratch.get(sizeof(float) * 2 * (size_t)ks * d->I);
        if (!part) ks = 1;
    }
    t.ksplit = ks; t.part = (float2 *)part;
    if (keep && hipMemsetAsync(y, 0, d->I * kN * sizeof(float2), s) != hipSuccess) return 0;   // planes are accumulated with atomics
    if (hipMemsetAsync(counter, 0, sizeof(uint32_t), s) != hipSuccess) return 0;
    const hipError_t e = launch_tile(t, dt, ntiles, s);
    if (e == hipErrorSharedObjectInitFailed) { (void)hipGetLastError(); return 1; }      // (a variant that is built on demand, and no compiler at hand: the any-shape kernel)
    g_lut_last = "tiled";
    return e == hipSuccess ? -1 : 0;     // (the probe of this footprint found no misfit on these very tables: every tile is written)
}

extern "C" int qdas_das_lut(const qdas_lut_desc *d, const void *x, void *y, void *stream) {
    if (!d || !y) return fail(QDAS_EINVAL, "null argument");
    if (d->dtype < QDAS_F64 || d->dtype > QDAS_F16) return fail(QDAS_EINVAL, "Unrecognized input precision %d", d->dtype);
    if ((d->flag & 7) > 5) return fail(QDAS_EINVAL, "Interp option not recognized: %d", d->flag & 7);
    const uint64_t oN = (d->flag & QDAS_FLAG_KEEP_RX) ? d->N : 1, oM = (d->flag & QDAS_FLAG_KEEP_TX) ? d->M : 1;
    hipStream_t s = (hipStream_t)stream;
    if (d->I == 0) return QDAS_OK;
    if (d->N == 0 || d->M == 0 || d->T == 0) {
        HIPCHK(hipMemsetAsync(y, 0, d->I * oN * oM * data_size(d->dtype), s));
        return QDAS_OK;
    }
    if (!x || !d->tau_rx || !d->tau_tx) return fail(QDAS_EINVAL, "null data / delay table");
    {   // fp32, full sum, no pixel-dependent weights: the fused tiled kernel with table-driven delays (das_tile_impl.h "LUT")
        const int rc = lut_tiled(d, x, y, s);
        if (rc <= 0) return rc < 0 ? QDAS_OK : fail(QDAS_EHIP, "das_lut: tiled launch failed");
    }
    g_lut_last = "generic";
    LutParams p{};
    p.tau_rx = d->tau_rx; p.tau_tx = d->tau_tx; p.w = d->w; p.x = x; p.y = y;
    p.T = d->T; p.N = d->N; p.M = d->M; p.I = d->I;
    p.wst[0] = d->wstride[0]; p.wst[1] = d->wstride[1]; p.wst[2] = d->wstride[2];
    p.omega = d->omega; p.flag = d->flag; p.w_real = d->w_real;
    HIPCHK(launch_lut(p, d->dtype, s));
    return QDAS_OK;
}

extern "C" int qdas_wsinterpd(const qdas_wsinterpd_desc *d, void *y, void *stream) {
    if (!d || !y) return fail(QDAS_EINVAL, "null argument");
    if (d->dtype < QDAS_F64 || d->dtype > QDAS_F16) return fail(QDAS_EINVAL, "Unrecognized input precision %d", d->dtype);
    if ((d->flag & 7) > 5 || (d->flag & ~7)) return fail(QDAS_EINVAL, "Interp option not recognized: %d", d->flag);
    if (d->ndim < 1 || d->ndim > 8) return fail(QDAS_EINVAL, "wsinterpd: 1..8 dimensions");
    if (d->xstride[0] != 0) return fail(QDAS_EINVAL, "wsinterpd: xstride[0] must be 0 (dimension 0 is the sampling dimension)");
    WsParams p{};
    p.t = d->t; p.w = d->w; p.x = d->x; p.y = y;
    p.T = d->T; p.x_tstride = d->x_tstride ? d->x_tstride : 1; p.nd = d->ndim;
    p.n_out = 1; p.n_sum = 1;
    for (int k = 0; k < d->ndim; ++k) {
        p.size[k] = d->size[k]; p.tst[k] = d->tstride[k]; p.xst[k] = d->xstride[k]; p.wst[k] = d->wstride[k]; p.sum[k] = d->sum[k] ? 1 : 0;
        if (p.sum[k]) { p.n_sum *= d->size[k]; p.any_sum = 1; } else p.n_out *= d->size[k];
    }
    p.omega = d->omega; p.extrap = d->extrap; p.flag = d->flag; p.w_real = d->w_real;
    for (int k = 0; k < d->ndim; ++k) {                  // the summed dimensions, compacted (size-1 dimensions do not advance anything)
        if (!p.sum[k] || d->size[k] <= 1) continue;
        if (d->size[k] > 0xffffffffull) return fail(QDAS_EUNSUPPORTED, "wsinterpd: a summed dimension of more than 2^32 - 1 elements");
        p.ssz[p.nsd] = (uint32_t)d->size[k]; p.sts[p.nsd] = d->tstride[k]; p.sxs[p.nsd] = d->xstride[k]; p.sws[p.nsd] = d->wstride[k];
        ++p.nsd;
    }
    // kept dimensions in decode order: lane_dim first (default: the first kept dimension with more than one element), then the others ascending;
    // y dense column-major over the kept dimensions unless the caller gave strides
    {
        bool ygiven = false;
        for (int k = 0; k < d->ndim; ++k) if (d->ystride[k] != 0) ygiven = true;
        int64_t acc = 1;
        for (int k = 0; k < d->ndim; ++k) {
            if (p.sum[k]) { p.yst[k] = 0; continue; }
            p.yst[k] = ygiven ? d->ystride[k] : acc;
            acc *= (int64_t)d->size[k];
        }
        int lane = d->lane_dim;
        if (lane < 0 || lane >= d->ndim || p.sum[lane] || d->size[lane] <= 1) {
            lane = -1;
            for (int k = 0; k < d->ndim && lane < 0; ++k) if (!p.sum[k] && d->size[k] > 1) lane = k;
            if (lane < 0) for (int k = 0; k < d->ndim && lane < 0; ++k) if (!p.sum[k]) lane = k;
        }
        if (lane < 0) { lane = 0; }                     // (every dimension summed: one output; dimension 0 then has size 1 in the output)
        p.nkd = 0; p.n_rest = 1;
        if (!p.sum[lane]) p.kord[p.nkd++] = lane;
        // (the others by ascending memory stride of x -- dimension 0: the sample stride; where x broadcasts: of t --, so that a second lane dimension and
        //  consecutive workgroups stay close in memory)
        auto skey = [&](int k) -> uint64_t {
            const uint64_t xs = k == 0 ? p.x_tstride : (uint64_t)(d->xstride[k] < 0 ? -d->xstride[k] : d->xstride[k]);
            if (xs && (k == 0 || d->size[k] > 1)) return xs;
            const uint64_t ts = (uint64_t)(d->tstride[k] < 0 ? -d->tstride[k] : d->tstride[k]);
            return (1ull << 62) + (ts ? ts : (1ull << 61) + (uint64_t)k);
        };
        for (int k = 0; k < d->ndim; ++k) if (!p.sum[k] && k != lane) { p.kord[p.nkd++] = k; p.n_rest *= d->size[k]; }
        for (int a = 1; a < p.nkd; ++a)                 // (insertion sort of at most 7 entries)
            for (int b = a; b > 1 && skey(p.kord[b]) < skey(p.kord[b - 1]); --b) std::swap(p.kord[b], p.kord[b - 1]);
        if (p.nkd == 0) {                               // all summed: a single output -- decode nothing (a pseudo dimension of size 1)
            p.kord[0] = 0; p.nkd = 1;
            static_assert(sizeof(p.size) / sizeof(p.size[0]) == 8, "");
            // dimension 0 is summed here: give the lane loop a size-1 view of it through a spare slot
            if (d->ndim < 8) { p.size[d->ndim] = 1; p.tst[d->ndim] = p.xst[d->ndim] = p.wst[d->ndim] = 0; p.yst[d->ndim] = 0; p.kord[0] = d->ndim; }
            else return fail(QDAS_EUNSUPPORTED, "wsinterpd: all 8 dimensions summed");
        }
        p.n_lane = p.size[p.kord[0]];
        p.lane2 = 0;
        if (p.nkd >= 2 && (p.n_lane < 2048 || p.n_lane % 256 != 0) && p.n_lane * p.size[p.kord[1]] < (1ull << 31)) {   // a short (or ragged) fastest dimension: the lanes also cover the next one -- full workgroups
            p.lane2 = 1;
            p.n_lane *= p.size[p.kord[1]];
            p.n_rest /= p.size[p.kord[1]] ? p.size[p.kord[1]] : 1;
        }
    }
    {   // the lean streaming kernel (wsinterpd.hip interpd_stream_kernel): plain sampling, dimension 0 among the block-level dimensions, 32-bit extents
        bool ok = !p.any_sum && !d->w && d->omega == 0.0 && p.nkd >= 2 && !getenv("QDAS_WS_GENERAL");
        const int nl = p.lane2 ? 2 : 1;
        for (int k = 0; k < nl && ok; ++k) if (p.kord[k] == 0) ok = false;
        if (ok && p.nkd <= nl) ok = false;
        auto ext = [&](const int64_t *st) { uint64_t e = 0; for (int k = 0; k < d->ndim; ++k) if (d->size[k] > 1) e += (uint64_t)(st[k] < 0 ? -st[k] : st[k]) * (d->size[k] - 1); return e; };
        if (ok) {
            for (int k = 0; k < d->ndim; ++k) if (p.tst[k] < 0 || p.xst[k] < 0 || p.yst[k] < 0) ok = false;
            const uint64_t ex = ext(p.xst) + (uint64_t)p.x_tstride * (d->T ? d->T - 1 : 0);
            if (ext(p.tst) >= (1ull << 31) || ex >= (1ull << 31) || ext(p.yst) >= (1ull << 31) || d->size[0] >= (1ull << 31) || d->T >= (1ull << 24)) ok = false;
        }
        p.stream_ok = ok ? 1 : 0;
    }
    // one summed dimension along which x is contiguous (a record in torch order -- last dimension fastest -- summed over that dimension): the lanes of a
    // wave take the terms of ONE output and add up across the wave (wsinterpd.hip wsinterpd_lanesum_kernel) instead of one output per lane, every tap a
    // 64-lane gather with one lane per memory row
    p.lanesum_ok = (p.nsd == 1 && p.sxs[0] == 1 && p.ssz[0] >= 16 && p.ssz[0] < (1u << 31) && d->T > 1 && p.x_tstride > 1 && p.n_out < (1ull << 32) && !getenv("QDAS_WS_NO_LANESUM")) ? 1 : 0;
    if (p.n_out == 0) return QDAS_OK;
    if (p.n_out >= (1ull << 39)) return fail(QDAS_EUNSUPPORTED, "wsinterpd: too many outputs for one launch");
    hipStream_t s = (hipStream_t)stream;
    if (p.n_sum == 0 || d->T == 0) {                     // empty sums / empty record
        if (p.n_sum == 0 || p.any_sum || d->extrap == 0.0) { HIPCHK(hipMemsetAsync(y, 0, p.n_out * data_size(d->dtype), s)); return QDAS_OK; }
    }
    if (!d->t || (!d->x && d->T)) return fail(QDAS_EINVAL, "null data / delay pointer");
    HIPCHK(launch_wsinterpd(p, d->dtype, s));
    return QDAS_OK;
}

extern "C" int qdas_greens(const qdas_greens_desc *d, void *y, void *stream) {
    if (!d || !y) return fail(QDAS_EINVAL, "null argument");
    if (d->dtype != QDAS_F64 && d->dtype != QDAS_F32) return fail(QDAS_EINVAL, "greens: datatype must be double or single");
    if (d->interp < 0 || d->interp > 5) return fail(QDAS_EINVAL, "Interp option not recognized: %d", d->interp);
    if (d->En < 1 || d->Em < 1) return fail(QDAS_EINVAL, "greens: element subdivisions must be >= 1");
    if (!(d->fs > 0) || !(d->fsr > 0) || !(d->R0 >= 0)) return fail(QDAS_EINVAL, "greens: fs, fsr must be positive and R0 non-negative");
    if (d->N > 65535 || d->M > 65535) return fail(QDAS_EUNSUPPORTED, "greens: at most 65535 receivers / transmitters");
    hipStream_t s = (hipStream_t)stream;
    DeviceGuard guard(d->device);
    HIPCHK(guard.err);
    if (d->S == 0 || d->N == 0 || d->M == 0) return QDAS_OK;
    if (d->I == 0 || d->T == 0) {
        HIPCHK(hipMemsetAsync(y, 0, d->S * d->N * d->M * data_size(d->dtype), s));
        return QDAS_OK;
    }
    if (!d->Ps || !d->a || !d->Pr || !d->Pv || !d->x) return fail(QDAS_EINVAL, "null scatterer / element / waveform pointer");
    GreensParams p{};
    p.Ps = d->Ps; p.a = d->a; p.Pr = d->Pr; p.Pv = d->Pv; p.x = d->x; p.y = y;
    p.S = d->S; p.T = d->T; p.N = d->N; p.M = d->M; p.I = d->I;
    p.En = d->En; p.Em = d->Em; p.interp = d->interp;
    p.s0 = d->s0; p.t0 = d->t0; p.fs = d->fs; p.fsr = d->fsr; p.cinv = d->cinv; p.R0 = d->R0;
    HIPCHK(launch_greens(p, d->dtype, s));
    return QDAS_OK;
}

// ---- pre-processing (pre.hip)
namespace qdas {
struct PrePlan;
int pre_create(PrePlan **out, uint64_t T, uint64_t K, uint64_t N, int in_type, double fs, double t0, double fd);
void pre_destroy(PrePlan *p);
int pre_execute(PrePlan *p, const void *x, void *y, hipStream_t s);
bool pre_one_pass(const PrePlan *p);
}
struct qdas_pre_plan { qdas::PrePlan *p; int device; };

extern "C" int qdas_pre_plan_create(qdas_pre_plan **out, const qdas_pre_desc *d) {
    if (!out || !d) return fail(QDAS_EINVAL, "null argument");
    *out = nullptr;
    if (d->in_type != QDAS_PRE_F32 && d->in_type != QDAS_PRE_I16) return fail(QDAS_EINVAL, "pre: input type must be fp32 or int16");
    const uint64_t N = d->Nfft ? d->Nfft : d->T;
    if (N >= (1ull << 31) || d->K >= (1ull << 31)) return fail(QDAS_EUNSUPPORTED, "pre: transform length / trace count too large");
    if (d->fdown != 0.0 && !(d->fs > 0)) return fail(QDAS_EINVAL, "Undefined sampling rate.");
    DeviceGuard guard(d->device);
    HIPCHK(guard.err);
    qdas_pre_plan *pl = new qdas_pre_plan();
    { hipError_t e = hipGetDevice(&pl->device); if (e != hipSuccess) { delete pl; return fail(QDAS_EHIP, "hipGetDevice: %s", hipGetErrorString(e)); } }
    const int rc = qdas::pre_create(&pl->p, d->T, d->K, d->Nfft, d->in_type, d->fs, d->t0, d->fdown);
    if (rc) { delete pl; return fail(rc == 2 ? QDAS_ENOMEM : QDAS_EHIP, "pre: hipFFT plan / workspace creation failed"); }
    *out = pl;
    return QDAS_OK;
}

extern "C" int qdas_pre_execute(qdas_pre_plan *pl, const void *x, void *y, void *stream) {
    if (!pl || !y) return fail(QDAS_EINVAL, "null argument");
    DeviceGuard guard(pl->device);
    HIPCHK(guard.err);
    const int rc = qdas::pre_execute(pl->p, x, y, (hipStream_t)stream);
    if (rc) return fail(QDAS_EHIP, "pre: hipFFT execution failed (%d)", rc);
    return QDAS_OK;
}

extern "C" int qdas_pre_plan_one_pass(const qdas_pre_plan *pl) { return pl && qdas::pre_one_pass(pl->p) ? 1 : 0; }

extern "C" void qdas_pre_plan_destroy(qdas_pre_plan *pl) {
    if (!pl) return;
    DeviceGuard guard(pl->device);
    qdas::pre_destroy(pl->p);
    delete pl;
}

// ---- transmit synthesis (shiftsum.hip)
extern "C" int qdas_shift_sum(const qdas_shift_desc *d, const void *x, void *y, void *stream) {
    if (!d || !y) return fail(QDAS_EINVAL, "null argument");
    if (d->dtype != QDAS_F64 && d->dtype != QDAS_F32) return fail(QDAS_EUNSUPPORTED, "shift_sum: datatype must be double or single");
    if ((d->flag & 7) > 5 || (d->flag & ~7)) return fail(QDAS_EINVAL, "Interp option not recognized: %d", d->flag);
    if (!d->cplx && d->w && !d->w_real) return fail(QDAS_EINVAL, "shift_sum: real data take real weights");
    if (d->To == 0 || d->N == 0 || d->Mo == 0 || d->F == 0) return QDAS_OK;
    if (!d->shift || (!x && d->T > (uint64_t)(d->tpad > 0 ? d->tpad : 0) && d->M)) return fail(QDAS_EINVAL, "null data / shift pointer");
    if (d->T >= (1ull << 31) || d->To >= (1ull << 31)) return fail(QDAS_EUNSUPPORTED, "shift_sum: at most 2^31 - 1 samples per trace");
    if (d->N > 65535 || ((d->Mo + 7) / 8) * d->F > 65535) return fail(QDAS_EUNSUPPORTED, "shift_sum: too many receivers / synthesised transmits x frames for one launch");
    if (d->M * d->Mo >= (1ull << 31)) return fail(QDAS_EUNSUPPORTED, "shift_sum: too many (element, transmit) pairs");
    DeviceGuard guard(d->device);
    HIPCHK(guard.err);
    ShiftParams p{};
    if (d->tpad < 0 || (uint64_t)d->tpad > d->T) return fail(QDAS_EINVAL, "shift_sum: tpad must lie in [0, T]");
    p.x = x; p.y = y; p.T = d->T; p.To = d->To; p.N = d->N; p.M = d->M; p.Mo = d->Mo; p.F = d->F;
    p.Tx = d->T - (uint64_t)d->tpad;
    if (d->M == 0) {                                   // an empty sum
        const size_t bytes = (size_t)d->To * d->N * d->Mo * d->F * (d->dtype == QDAS_F64 ? 8 : 4) * (d->cplx ? 2 : 1);
        HIPCHK(hipMemsetAsync(y, 0, bytes, (hipStream_t)stream));
        return QDAS_OK;
    }
    HIPCHK(launch_shift_sum(p, d->dtype, d->cplx ? 1 : 0, d->flag & 7, d->shift, d->w, d->w_real, (hipStream_t)stream));
    return QDAS_OK;
}

// ---- batched 1-D convolution (conv.hip)
extern "C" uint64_t qdas_convd_len(uint64_t M, uint64_t N, int shape) {
    if (M == 0 || N == 0) return 0;
    switch (shape) {
        case QDAS_CONV_FULL:  return M + N - 1;
        case QDAS_CONV_SAME:  return M;
        case QDAS_CONV_VALID: return M >= N ? M - N + 1 : 0;
        case QDAS_CONV_CAUSAL: return M;
        default: return 0;
    }
}

extern "C" int qdas_convd(const qdas_convd_desc *d, const void *x, const void *y, void *z, void *stream) {
    if (!d) return fail(QDAS_EINVAL, "null argument");
    if (d->dtype != QDAS_F64 && d->dtype != QDAS_F32 && d->dtype != QDAS_F16) return fail(QDAS_EINVAL, "convd: datatype must be double, single or half");
    if (d->shape != QDAS_CONV_FULL && d->shape != QDAS_CONV_SAME && d->shape != QDAS_CONV_VALID && d->shape != QDAS_CONV_CAUSAL)
        return fail(QDAS_EINVAL, "convd: shape must be one of {'full', 'same', 'valid'}");
    if (d->bcast & ~15) return fail(QDAS_EINVAL, "convd: unknown broadcast bits");
    if (d->y_real && !d->cplx) return fail(QDAS_EINVAL, "convd: y_real describes complex data with real taps (cplx must be 1)");
    if (d->M >= (1ull << 31) || d->N >= (1ull << 31)) return fail(QDAS_EUNSUPPORTED, "convd: at most 2^31 - 1 samples along the convolved dimension");
    const uint64_t L = qdas_convd_len(d->M, d->N, d->shape);
    if (L == 0 || d->C == 0 || d->S == 0) return QDAS_OK;
    if (!x || !y || !z) return fail(QDAS_EINVAL, "null data pointer");
    const uint64_t ncb = (d->C + 63) / 64;
    if (d->S * (d->C == 1 ? 1 : ncb) >= (1ull << 31) || (L + 15) / 16 > 65535ull * (d->C == 1 ? 64 : 1))
        return fail(QDAS_EUNSUPPORTED, "convd: too many slices / outputs for one launch");
    DeviceGuard guard(d->device);
    HIPCHK(guard.err);
    ConvParams p{};
    p.x = x; p.y = y; p.z = z;
    p.C = d->C; p.M = d->M; p.N = d->N; p.L = L; p.S = d->S;
    p.off = (d->shape == QDAS_CONV_FULL || d->shape == QDAS_CONV_CAUSAL) ? 0 : d->shape == QDAS_CONV_VALID ? (int64_t)d->N - 1 : (int64_t)(d->N - 1 - (d->N - 1) / 2);
    const uint64_t Cx = (d->bcast & QDAS_CONV_X_ONE_COLUMN) ? 1 : d->C, Cy = (d->bcast & QDAS_CONV_Y_ONE_COLUMN) ? 1 : d->C;
    p.xcs = Cx == 1 && d->C > 1 ? 0 : 1; p.xts = Cx; p.xss = (d->bcast & QDAS_CONV_X_ONE_SLICE) ? 0 : Cx * d->M;
    p.ycs = Cy == 1 && d->C > 1 ? 0 : 1; p.yts = Cy; p.yss = (d->bcast & QDAS_CONV_Y_ONE_SLICE) ? 0 : Cy * d->N;
    // long filters on complex64 traces, ONE filter for all of them, time contiguous (ChannelData.filter's band-pass): FFT convolution with the trace
    // resident in LDS (pre.hip fftconv_launch) -- from QDAS_CONV_FFT_MIN_TAPS taps on (default 128: it overtakes the direct kernel at about 120 taps on the C3 record, profiles/r04/convd_fft_time.txt)
    {
        static const bool no_fft = getenv("QDAS_CONV_NO_FFT") != nullptr;
        uint64_t min_taps = 128;
        if (const char *e = getenv("QDAS_CONV_FFT_MIN_TAPS")) { const long long v = atoll(e); if (v >= 2) min_taps = (uint64_t)v; }
        const bool one_filter = d->S == 1 || (d->bcast & QDAS_CONV_Y_ONE_SLICE), every_trace = d->S == 1 || !(d->bcast & QDAS_CONV_X_ONE_SLICE);
        if (!no_fft && d->dtype == QDAS_F32 && d->cplx && d->C == 1 && one_filter && every_trace && d->N >= min_taps) {
            const int rc = fftconv_launch(x, y, d->y_real ? 1 : 0, z, d->M, d->N, d->S, (uint64_t)p.off, L, (hipStream_t)stream);
            if (rc == 0) return QDAS_OK;
            if (rc == 2) return fail(QDAS_EHIP, "convd: the FFT convolution kernel failed to launch");
        }
    }
    HIPCHK(launch_conv(p, d->dtype, d->cplx ? 1 : 0, d->y_real ? 1 : 0, (hipStream_t)stream));
    return QDAS_OK;
}

// ---- layout conversion for row-major hosts (layout.hip)
extern "C" int qdas_permute3(const void *in, void *out, uint64_t A, uint64_t B, uint64_t C, int elem_bytes, void *stream) {
    if (A == 0 || B == 0 || C == 0) return QDAS_OK;
    if (!in || !out) return fail(QDAS_EINVAL, "null argument");
    if (elem_bytes != 2 && elem_bytes != 4 && elem_bytes != 8 && elem_bytes != 16) return fail(QDAS_EINVAL, "permute3: element size must be 2, 4, 8 or 16 bytes");
    if (B > 65535 || (A + 63) / 64 > 65535) return fail(QDAS_EUNSUPPORTED, "permute3: too many slices for one launch");
    HIPCHK(launch_permute3(in, out, A, B, C, elem_bytes, (hipStream_t)stream));
    return QDAS_OK;
}

// ---- device staging for host callers of the device-pointer entries (include/qdas.h; the MEX gateway's host-array path)
// Staging buffers are RECYCLED: a gateway call allocates its arguments and frees them again, and on this platform hipMalloc / hipFree around calls that
// use temporaries of the stream-ordered pool had the second call of a fresh process read stale memory (tools/repro/stale_pool.hip: 4 of 30 boxes; 0 of 30
// with the staged buffers kept; csrc/scratch.hip removes the pool side as well).  A buffer that stays mapped does not do that -- and hipMalloc / hipFree
// (synchronising, ~100 us each) leave the call path.  Freed buffers are kept per device and size class (256-byte steps below 1 MiB, 1 MiB steps above),
// at most 4 GiB in all (QDAS_STAGING_CACHE_MB; a gateway that stages a 1.5 GB record per call keeps that buffer too); qdas_device_trim releases them.
namespace {
struct StagingCache {
    std::mutex mu;
    struct Item { void *p; size_t bytes; int dev; };
    std::vector<Item> free_list;
    std::vector<Item> live;                               // (what qdas_device_malloc handed out: size class and device of a pointer)
    size_t cached = 0;
};
StagingCache &staging() { static StagingCache c; return c; }
size_t size_class(size_t bytes) { const size_t step = bytes < (1u << 20) ? 256 : (1u << 20); return (std::max<size_t>(bytes, 1) + step - 1) / step * step; }
}  // namespace

extern "C" int qdas_device_malloc(void **p, size_t bytes, int device) {
    if (!p) return fail(QDAS_EINVAL, "null argument");
    *p = nullptr;
    DeviceGuard guard(device);
    HIPCHK(guard.err);
    int dev = device;
    if (dev < 0) HIPCHK(hipGetDevice(&dev));
    const size_t cls = size_class(bytes);
    StagingCache &c = staging();
    {
        std::lock_guard<std::mutex> lk(c.mu);
        for (size_t k = c.free_list.size(); k-- > 0;)
            if (c.free_list[k].bytes == cls && c.free_list[k].dev == dev) {
                *p = c.free_list[k].p;
                c.cached -= cls;
                c.live.push_back(c.free_list[k]);
                c.free_list.erase(c.free_list.begin() + (long)k);
                return QDAS_OK;
            }
    }
    hipError_t e = hipMalloc(p, cls);
    if (e != hipSuccess) {                                // (out of memory: give the cache back and try once more)
        (void)hipGetLastError();
        qdas_device_trim();
        e = hipMalloc(p, cls);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return fail(QDAS_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
    std::lock_guard<std::mutex> lk(c.mu);
    c.live.push_back({*p, cls, dev});
    return QDAS_OK;
}
extern "C" int qdas_device_free(void *p, int device) {
    if (!p) return QDAS_OK;
    StagingCache &c = staging();
    {
        std::lock_guard<std::mutex> lk(c.mu);
        for (size_t k = 0; k < c.live.size(); ++k)
            if (c.live[k].p == p) {
                const StagingCache::Item it = c.live[k];
                c.live.erase(c.live.begin() + (long)k);
                static const size_t cap = [] { const char *e = getenv("QDAS_STAGING_CACHE_MB"); return (size_t)(e && atoll(e) >= 0 ? atoll(e) : 4096) << 20; }();
                if (c.cached + it.bytes <= cap) {
                    // hipFree waits for the device before it unmaps; a RECYCLED buffer must be as idle as a freed one: the next qdas_device_malloc of
                    // this size class hands it out at once, and a kernel the caller launched on a non-blocking stream may still be reading or writing it -- the
                    // next owner's upload (null stream: not ordered with hipStreamNonBlocking streams) would race with it.  One device-wide wait, outside the lock.
                    c.mu.unlock();
                    hipError_t se;
                    { DeviceGuard guard(it.dev); se = guard.err != hipSuccess ? guard.err : getenv("QDAS_DEVICE_FREE_NO_SYNC") ? hipSuccess : hipDeviceSynchronize(); }      // (the switch: tests show the hazard with it)
                    c.mu.lock();
                    if (se != hipSuccess) { (void)hipGetLastError(); c.live.push_back(it); return fail(QDAS_EHIP, "qdas_device_free: hipDeviceSynchronize: %s", hipGetErrorString(se)); }
                    c.free_list.push_back(it);
                    c.cached += it.bytes;
                    return QDAS_OK;
                }
                break;
            }
    }
    DeviceGuard guard(device);
    HIPCHK(guard.err);
    HIPCHK(hipFree(p));
    return QDAS_OK;
}
extern "C" int qdas_device_trim(void) {
    scratch_trim();                                       // (the one-shot entries' arenas of idle streams)
    StagingCache &c = staging();
    std::vector<StagingCache::Item> drop;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        drop.swap(c.free_list);
        c.cached = 0;
    }
    int rc = QDAS_OK;
    for (const auto &it : drop) {
        DeviceGuard guard(it.dev);
        if (guard.err != hipSuccess || hipFree(it.p) != hipSuccess) { (void)hipGetLastError(); rc = QDAS_EHIP; }
    }
    return rc;
}
extern "C" int qdas_device_copy(void *dst, const void *src, size_t bytes, int kind, int device) {
    if (!bytes) return QDAS_OK;
    if (!dst || !src || kind < 0 || kind > 2) return fail(QDAS_EINVAL, "qdas_device_copy: null pointer or unknown kind");
    DeviceGuard guard(device);
    HIPCHK(guard.err);
    if (kind == 0) HIPCHK(upload(dst, src, bytes));
    else HIPCHK(hipMemcpy(dst, src, bytes, kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice));
    return QDAS_OK;
}
