/* qdas.h -- C ABI of libqdas.so: MI355X-native delay-and-sum beamforming engine.
 *
 * Drop-in boundary for ONE reference call site: the device launch inside
 * `das_spec` (reference kern/das_spec.m:279-306 builds the kernel object and
 * uploads the size constants; kern/das_spec.m:372 is the per-frame launch
 *
 *     y{f} = k.feval(yg, Pi, Pr, Pv, Nv, apod, cinv, [cstride, astride],
 *                    x(:,:,:,f), [fs, fmod]);
 *
 * of `DAS` / `DASf` / `DASh`, reference src/bf.cu:144-172), plus the `delays`
 * variant (kern/das_spec.m:377, src/bf.cu:209-298) and -- as the "next" row --
 * the split-delay launch of `wsinterpd2[f|h]` (kern/wsinterpd2.m:236,
 * src/interpd.cu:449-476) used by bfDAS/bfDASLUT.  The steps either side of
 * the path have their own entry points further down: `qdas_greens`
 * (src/greens.cu), `qdas_shift_sum` (UltrasoundSystem.focusTx), `qdas_pre_*` (ChannelData.hilbert / downmix), `qdas_convd`
 * (src/convd.cu), and `qdas_permute3` for row-major hosts.
 *
 * Everything is plain C: pointers, sizes, no torch / HIP types in signatures
 * (`stream` is a `hipStream_t` passed as `void*`; NULL = the default stream).
 * All arrays use MATLAB (column-major) memory order exactly as the reference
 * kernel ABI does.  Functions return 0 on success, a QDAS_E* code otherwise;
 * `qdas_last_error()` returns a thread-local message (the MEX shim maps it to
 * mexErrMsgIdAndTxt("QUPS:das_spec:...")).  The library never frees or keeps
 * caller memory beyond a call, except device copies it makes itself.
 */
#ifndef QDAS_H
#define QDAS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 100: rounds 1-3.  101 (round 5): qdas_wsinterpd_desc carries ystride[8] / lane_dim / reserved (added in round 4 without a bump), QDAS_PLAN_PREFOLDED rejects
 * apodization arrays.  102 (round 5): the device staging entries recycle their buffers (qdas_device_trim), one-shot entries take temporaries from kept arenas.
 * 103 (round 6): qdas_device_free waits for the device before a buffer can be handed out again (it used to be caller's business, unstated).  EVERY descriptor of this header must be ZERO-INITIALISED by the caller (memset / `= {0}`) before its fields are set: fields added by
 * later versions then read as "default", and a caller compiled against an older header must check qdas_version() against the QDAS_VERSION it was built with. */
#define QDAS_VERSION 103

/* ---- data precision: the reference's kernel postfix (kern/das_spec.m:218-222) */
#define QDAS_F64 0 /* 'DAS'  : double2 data/apod/y, double geometry + time            */
#define QDAS_F32 1 /* 'DASf' : float2  data/apod/y, float  geometry + time            */
#define QDAS_F16 2 /* 'DASh' : half2   data/apod/y (passed as ushort2), float geometry */

/* ---- QUPS_BF_FLAG bit layout (reference kern/das_spec.m:198-213, src/bf.cu:100,126,129) */
#define QDAS_INTERP_NEAREST  0
#define QDAS_INTERP_LINEAR   1
#define QDAS_INTERP_CUBIC    2 /* Catmull-Rom = MATLAB interp1 'cubic' (the CPU path)   */
#define QDAS_INTERP_LANCZOS3 3 /* window a = 2, 4 taps (src/interpd.cu:116-150)          */
#define QDAS_INTERP_LINEAR4  4 /* alias of linear (src/interpd.cu:163)                   */
#define QDAS_INTERP_CUBIC_DEV 5 /* extension: the Horner lines the CUDA/OpenCL code executes
                                   (src/interpd.cu:103-106) -- NOT Catmull-Rom; see DESIGN.md */
#define QDAS_FLAG_INTERP_MASK 7
#define QDAS_FLAG_KEEP_RX 8   /* 'SYN' | 'BF' : keep the receive dimension   */
#define QDAS_FLAG_KEEP_TX 16  /* 'MUL' | 'BF' : keep the transmit dimension  */
#define QDAS_FLAG_TPOSE   32  /* data is T x M x N instead of T x N x M      */

/* ---- where caller pointers live */
#define QDAS_MEM_HOST   0
#define QDAS_MEM_DEVICE 1

/* ---- kernel selection (QDAS_KERNEL_AUTO picks the tiled kernel when eligible)
 * Eligible: 'DAS' with any data precision -- fp64 data: pixel-independent apodization and / or one pixel x receiver (pixel-only) array, scalar sound speed --,
 * 'SYN' / 'MUL' / 'BF' with fp32 data; scalar sound speed or a full per-pixel map; any number of pixel-independent apodization arrays
 * (folded into an N x M table) plus pixel-dependent arrays of ONE side: I x N (pixel x receiver), or I x 1 x M (pixel x transmit: 'DAS' / 'MUL';
 * the roles of the two apertures are swapped), with any I-only arrays (spatial weights / region-of-interest masks) -- several of them, or arrays
 * that broadcast over a pixel dimension (I1 x 1 x 1 x N), are multiplied into one plan-owned array at plan creation --, or one generated receive
 * rule (rx_apod_kind); fp32 data with real weights and no remodulation also takes a transmit-side AND a receive-side pixel array together
 * (the receive-side product is applied per pair);
 * N and M up to about 1000 each.  Everything else runs the generic kernel with identical semantics; qdas_last_error() after a
 * QDAS_KERNEL_TILED request says why a plan is not eligible. */
#define QDAS_KERNEL_AUTO    0
#define QDAS_KERNEL_GENERIC 1 /* one pixel per lane, any mode / broadcast shape        */
#define QDAS_KERNEL_TILED   2 /* LDS-staged pixel tiles; error if the case is ineligible */

/* ---- generated receive apodization w(pixel, receiver), r = Pi - Pr(n), nhat = element normal:
 *  ACCEPTANCE     (reference src/UltrasoundSystem.m:5355-5373 apAcceptanceAngle): w = [ r.nhat/|r| >= p[0] ],  p[0] = cosd(theta)
 *  COSINE         (:5414-5428 apCosineAngle): w = cos(min(pi/2, p[0]*acos(clip(r.nhat/|r|, -1, 1)))),        p[0] = 90/theta
 *  FNUMBER_PLANAR (:5251-5256 apApertureGrowth, planar array): d = x_n - x_i, z = z_i: w = [z > p[0]*|2d|]*[|2d| < p[1]],  p = {f, Dmax}
 *  FNUMBER_ORIENTED (:5244-5250,5255-5256, non-planar array, elements rotated about y): d = r_x*n_z - r_z*n_x, z = |r_x*n_x + r_z*n_z| */
#define QDAS_RXAPOD_NONE             0
#define QDAS_RXAPOD_ACCEPTANCE       1
#define QDAS_RXAPOD_COSINE           2
#define QDAS_RXAPOD_FNUMBER_PLANAR   3
#define QDAS_RXAPOD_FNUMBER_ORIENTED 4

/* ---- plan flags (qdas_desc.plan_flags) */
#define QDAS_PLAN_NO_RECIPROCAL 1 /* never use the reciprocal mode of the tiled kernel (transmit elements == receive elements):
                                     the general kernel that every apodized / non-FSA acquisition runs                        */
#define QDAS_PLAN_JIT           2 /* compile the tiled kernel for THIS plan's sizes with hiprtc: N, M, T, strides, delay kinds,
                                     tile shape and modes become constants, the reference's const-compile specialisation
                                     (src/UltrasoundSystem.m:5626-5748 getDASConstCudaDef, src/sizes.cu:17-52).  Cached on disk
                                     (QDAS_CACHE_DIR, default ~/.cache/qdas); falls back to the prebuilt kernel with a message
                                     in qdas_last_error() if hiprtc is unavailable                                            */

#define QDAS_PLAN_COPY_INPUTS   4 /* desc.mem == QDAS_MEM_DEVICE: copy Pi, Pr, Pv, Nv, apod, cinv, rx_normals into plan-owned device
                                     buffers instead of reading the caller's arrays in place.  For hosts whose arrays do not
                                     outlive the call that created the plan (the MEX gateway's persistent handles: MATLAB frees
                                     the gpuArrays of das_spec's workspace when it returns)                                  */

#define QDAS_PLAN_NO_MIRROR     8 /* never use the lateral-mirror mode of the tiled kernel (scan, array and sequence mirror-symmetric
                                     about x = 0, detected from the geometry: tap index and weights are shared by a pixel and its
                                     mirror image)                                                                            */

#define QDAS_PLAN_MIRROR_SLAB   16 /* a pixel slab AND its mirror image in one plan (multi-GPU jobs that keep the lateral-mirror mode):
                                     i_begin / i_count describe slab A -- whole columns [c0, c1) of the FIRST half of an even number of
                                     columns I2 (I3 == 1) --; the plan also beamforms the mirror-image columns [I2 - c1, I2 - c0) and
                                     writes them behind slab A: y[0 .. i_count) = slab A, y[i_count .. 2 i_count) = slab B, both in
                                     natural pixel order ('DAS' only; y holds 2 i_count pixels).  QDAS_EUNSUPPORTED when the geometry
                                     is not mirror-symmetric or the mode is not available for the problem: use plain slabs then.  */

#define QDAS_PLAN_NO_FOLD       32 /* reciprocal plans with fp32 data: do NOT fold the frame (qdas_plan_folded) -- gather both traces of every
                                     unordered pair per pixel and share only tap index and weights between them, as rounds 1-3 did.  The fold
                                     costs a plan-owned copy of the frame (T x N x M complex64) and one streaming pass per frame            */

#define QDAS_PLAN_APPROX_SYMMETRY 64 /* take the reciprocal / lateral-mirror modes also when the geometry is symmetric only WITHIN A TOLERANCE: element and
                                     pixel POSITIONS may deviate from their partners' (mirror) images as long as the bound on the delay error this commits,
                                     cinv fs (2 max|M(p') - p| + max|M(r') - r| + max|M(v') - v|) resp. 2 cinv fs max|r_m - v_m|, stays <= 1e-5 sample
                                     (QDAS_SYM_TOL=<samples> overrides; normals and start times must match exactly).  Off by default: the modes then need
                                     bit-exact symmetry.  1e-5 sample is 0.8 nm of path at 20 MHz / 1540 m/s -- sub-ulp deviations of coordinates of a
                                     few mm; a delay error d sample shows up as a relative image error of about 2 pi (fc / fs) d: the image tolerance
                                     of 1e-4 allows ~6e-5 sample, i.e. ~5 nm, NOT micrometres -- a calibrated probe is simply not symmetric.
                                     qdas_plan_symmetry_bound reports the bounds of the modes in use                                          */

#define QDAS_PLAN_PREFOLDED     128 /* the frames handed to execute ARE folded frames (qdas_fold: complex64, upper triangle n <= m; weights applied there): the plan
                                     runs no fold pass and owns no folded copy.  For hosts that fold once per acquisition and replicate the folded frame (half
                                     the bytes) to several devices.  Needs a reciprocal fp32 'DAS' problem without apodization arrays whose tiles all fit the
                                     staging windows, else QDAS_EUNSUPPORTED                                                                       */

/* ---- LIFETIME of caller memory.  Host arrays (QDAS_MEM_HOST) are copied at qdas_plan_create and never touched again.  Device
 *      arrays (QDAS_MEM_DEVICE) are used IN PLACE: Pi, Pr, Pv, Nv, apod, cinv and rx_normals must stay allocated and unchanged
 *      until qdas_plan_destroy -- unless the plan was created with QDAS_PLAN_COPY_INPUTS.  acstride is read at creation only.
 *      x / y belong to the caller and are only accessed by the execute call they are passed to (asynchronously on `stream` for
 *      device memory: keep them alive until the stream has passed the call).  Every entry restores the calling thread's
 *      current HIP device before it returns. */

/* ---- error codes */
#define QDAS_OK            0
#define QDAS_EINVAL        1 /* bad argument / inconsistent sizes (message says which)  */
#define QDAS_EUNSUPPORTED  2 /* valid request the build cannot serve                    */
#define QDAS_EHIP          3 /* HIP runtime error (message carries hipGetErrorString)   */
#define QDAS_ENOMEM        4

/* Size/flag constants: the reference's constant-memory symbols QUPS_{T,N,M,I,I1,I2,I3,S},
 * QUPS_{VS,DV}, QUPS_BF_FLAG (reference src/sizes.cu:17-52, src/bf.cu:45-47;
 * uploaded at kern/das_spec.m:294-298). */
typedef struct qdas_sizes {
    uint64_t T;          /* fast-time samples per trace                                  */
    uint64_t N, M;       /* receivers, transmits                                         */
    uint64_t I1, I2, I3; /* image size; I = I1*I2*I3, I1 fastest                         */
    uint64_t S;          /* number of apodization arrays (0 = none; reference passes {1}) */
    int32_t  flag;       /* QUPS_BF_FLAG                                                  */
    int32_t  VS;         /* 1: virtual-source model, 0: plane-wave model                  */
    int32_t  DV;         /* 1: diverging wave (distance always positive)                  */
    int32_t  dtype;      /* QDAS_F64 | QDAS_F32 | QDAS_F16                                */
} qdas_sizes;

/* Full description of one beamforming problem (everything except the data). */
typedef struct qdas_desc {
    qdas_sizes sz;
    double fs, fmod;      /* [fs, fmod] == tvars (src/bf.cu:57-58)                        */
    /* geometry, real(prec) -- float for QDAS_F16 (kern/das_spec.m:356) */
    const void *Pi;       /* 3 x I    pixel positions                                      */
    const void *Pr;       /* 3 x N    receiver positions                                   */
    const void *Pv;       /* 4 x M    (virtual) source positions, row 4 = t0 (das_spec.m:361) */
    const void *Nv;       /* 3 x M    transmit normals                                     */
    const void *apod;     /* concatenated apodization arrays (das_spec.m:344-345); complex(prec)
                             unless apod_real != 0; may be NULL iff S == 0                 */
    const void *cinv;     /* 1/c, real(prec), broadcastable I1 x I2 x I3 x N x M           */
    const uint64_t *acstride; /* HOST pointer, 6*(1+S) entries: [cstride(6), astride(6 x S)]
                             element strides for dims (I1,I2,I3,N,M) -- 0 where singleton -- and
                             the array's base offset in entry 6 (reference das_spec.m:257-260) */
    int32_t mem;          /* QDAS_MEM_*: where Pi..cinv AND x / y of execute() live        */
    int32_t apod_real;    /* extension: apod buffer holds real(prec) weights               */
    int32_t kernel;       /* QDAS_KERNEL_*                                                 */
    int32_t device;       /* HIP device ordinal, -1 = current device                       */
    /* pixel shard (multi-GPU: every rank builds the same desc with its own slab;
       single GPU: i_begin = 0, i_count = 0 meaning "all I") */
    uint64_t i_begin;     /* first linear pixel index of this plan                         */
    uint64_t i_count;     /* number of pixels (0 = I - i_begin)                            */
    uint64_t y_ld;        /* pixels between consecutive [n|m] planes of y (0 = i_count): lets a
                             shard write straight into a full-size I x [N] x [M] buffer    */
    /* extension: receive apodization GENERATED in the kernel from the geometry instead of a
       materialised I1 x I2 x I3 x N array; multiplies the apodization arrays.  rx_normals: 3 x N element
       normals, real(prec) like Pr (float for QDAS_F16); same memory kind as Pr.                            */
    int32_t  rx_apod_kind;   /* QDAS_RXAPOD_*                                                             */
    int32_t  plan_flags;     /* QDAS_PLAN_* bits (0 = defaults)                                           */
    double   rx_apod_p[2];   /* parameters, see QDAS_RXAPOD_*                                             */
    const void *rx_normals;  /* may be NULL for QDAS_RXAPOD_NONE / QDAS_RXAPOD_FNUMBER_PLANAR             */
} qdas_desc;

typedef struct qdas_plan qdas_plan; /* opaque: device copies of geometry + strides + kernel choice */

/* ---- plan API: the reusable [k, PRE_ARGS, POST_ARGS] handle of the reference
 *      (kern/das_spec.m:72-81,387-390) */
int  qdas_plan_create(qdas_plan **plan, const qdas_desc *desc);
/* Beamform ONE frame: x is T x N x M (or T x M x N with QDAS_FLAG_TPOSE) complex(prec);
 * y is i_count x [1|N] x [1|M] complex(prec) with plane stride y_ld; y is fully
 * overwritten.  Asynchronous on `stream` when desc.mem == QDAS_MEM_DEVICE. */
int  qdas_plan_execute(qdas_plan *plan, const void *x, void *y, void *stream);
/* F frames in one call (kern/das_spec.m:371-373 host loop): frame f uses
 * x + f*x_stride and y + f*y_stride (strides in complex elements).  Device-resident frames of a tiled plan share launches
 * (four / two per launch: tap index and interpolation weights are evaluated once per group), so a frame's image equals
 * qdas_plan_execute's to fp32 re-association, not bit for bit, and every frame of ONE call is summed in the same order.
 * A general-mode lateral-mirror plan streams F >= 4 frames through a twin plan without the mode, created at the first such
 * call (device allocations and probe launches of a plan creation, synchronous, ~plan memory x2): that first call is not
 * asynchronous and not graph-capturable; set QDAS_NO_FRAMES_TWIN=1 to keep streams on the plan's own kernel. */
int  qdas_plan_execute_frames(qdas_plan *plan, const void *x, void *y, uint64_t F,
                              uint64_t x_stride, uint64_t y_stride, void *stream);
/* Everything a stream of F frames does ONCE per plan -- the twin plan above; for a reciprocity-folded plan the second folded copy of a frame (T x N x M
 * complex64) that lets two frames share a launch; the frame-sharing kernel instantiations libqdas.so does not carry (built through hiprtc, ~2 s each);
 * the second staging set of host-resident frames -- done now, synchronously.  Optional: qdas_plan_execute_frames does the same at the plan's first
 * stream, which makes THAT call synchronous and not graph-capturable; after qdas_plan_prepare_frames(plan, F) every stream of up to F frames only enqueues. */
int  qdas_plan_prepare_frames(qdas_plan *plan, uint64_t F);
/* 'delays' (src/bf.cu:209-298, kern/das_spec.m:377): tau is i_count x N x M real(prec),
 * tau = cinv(1) * (dv + dr); no t0, like the reference. */
int  qdas_plan_delays(qdas_plan *plan, void *tau, void *stream);
void qdas_plan_destroy(qdas_plan *plan);
/* which kernel a plan resolved to (QDAS_KERNEL_GENERIC | QDAS_KERNEL_TILED) and, after an
 * execute, how many pixel tiles fell back to the generic kernel (oversize delay window) */
int  qdas_plan_kernel(const qdas_plan *plan);
int  qdas_plan_fallback_tiles(const qdas_plan *plan, uint64_t *ntiles);
/* shapes a QDAS_KERNEL_TILED plan chose from the scan's delay gradient and size: a workgroup tile of
 * tile_z pixels of I1 (64 | 32 | 16 | 8) x tile_cols columns of I2*I3, made of waves of wave_z x
 * (64 / wave_z) pixels, and ksplit workgroups per tile that each sum a slice of the aperture (> 1 when the
 * image or pixel slab has too few tiles to fill the GPU); wave_z / ksplit may be NULL; all 0 for a
 * generic-kernel plan */
int  qdas_plan_tile_shape(const qdas_plan *plan, int *tile_z, int *tile_cols, int *wave_z, int *ksplit);
/* 1 when a QDAS_KERNEL_TILED plan runs in reciprocal mode (transmit elements == receive elements, one t0: every unordered
 * transmit/receive pair is indexed and weighted once), else 0 */
int  qdas_plan_reciprocal(const qdas_plan *plan);
/* 1 when a reciprocal plan beamforms the RECIPROCITY-FOLDED frame (fp32 data; QDAS_PLAN_NO_FOLD): every execute first adds the two traces of
 * each unordered transmit/receive pair, xs[:,n,m] = w[n,m] x[:,n,m] + w[m,n] x[:,m,n] (n <= m; one pass over HBM into a plan-owned copy of the
 * frame, pixel-independent apodization applied on the way), and the fused kernel then walks the upper triangle only */
int  qdas_plan_folded(const qdas_plan *plan);
/* The reciprocity fold on its own: xs[:,n,m] = w[n,m] x[:,n,m] + w[m,n] x[:,m,n] for n < m, xs[:,n,n] = w[n,n] x[:,n,n]; traces of T samples at (n strN + m strM)
 * samples in x AND xs (0: T, T N); x complex64 (QDAS_F32) or complex32 (QDAS_F16), xs complex64 always; wtab: device N x N complex64 [n + N m] or NULL (ones);
 * only the upper triangle of xs is written (zero-fill it once).  Device pointers; asynchronous on `stream`. */
typedef struct qdas_fold_desc {
    uint64_t T, N;
    uint64_t strN, strM;
    int32_t  dtype;       /* QDAS_F32 | QDAS_F16: type of x */
    int32_t  device;      /* HIP device ordinal, -1 = current */
    const void *wtab;
} qdas_fold_desc;
int  qdas_fold(const qdas_fold_desc *d, const void *x, void *xs, void *stream);
/* bounds [samples] of the delay error committed by the lateral-mirror / reciprocal mode of the plan: 0 = exact symmetry, > 0 = accepted within
 * the tolerance of QDAS_PLAN_APPROX_SYMMETRY, -1 = that mode is not in use (either pointer may be NULL) */
int  qdas_plan_symmetry_bound(const qdas_plan *plan, double *mirror_samples, double *reciprocal_samples);
/* 1 when a QDAS_KERNEL_TILED plan runs in lateral-mirror mode (QDAS_PLAN_NO_MIRROR), else 0 */
int  qdas_plan_mirror(const qdas_plan *plan);
/* human-readable name of the kernel a plan launches for one frame, e.g.
 *   "das_tile_kernel<interp=3,f32,sym,mb=16,W=128> [prebuilt]"   |   "... [jit 5f0c...]"  (QDAS_PLAN_JIT, hiprtc build)   |
 *   "das_generic_kernel<interp=2,f64>"
 * Written NUL-terminated into buf (truncated to len). */
int  qdas_plan_kernel_name(const qdas_plan *plan, char *buf, size_t len);
/* time of the last execute()'s kernels in ms measured with hipEvents on its stream
 * (enabled by qdas_plan_set_timing(plan, 1); synchronises the stream) */
int  qdas_plan_set_timing(qdas_plan *plan, int enable);
int  qdas_plan_last_kernel_ms(const qdas_plan *plan, float *ms);

/* ---- one host thread, several devices: the image is split into ndev contiguous slabs of the linear pixel
 *      index, one per entry of devices[] (NULL: 0 .. ndev-1; an ordinal may repeat: several streams on one device); geometry and
 *      channel data are replicated, every device beamforms its slab with an ordinary plan, the slabs are concatenated into y.
 *      desc->mem == QDAS_MEM_DEVICE: the constant inputs and x / y live on devices[0]; x is replicated with peer copies over the xGMI mesh --
 *      scatter (device u pulls piece u of the frame from devices[0]) + all-gather (every device pulls the other pieces from their holders:
 *      all links carry 1/(U-1) of the frame at once) --, slabs (mirror slabs when the geometry allows: a slab of the first half of the
 *      columns AND its mirror image per device) come back with peer copies; asynchronous on `stream` (a stream of devices[0]).
 *      desc->mem == QDAS_MEM_HOST: x is uploaded once to devices[0] and replicated from there; y is downloaded; synchronous.
 *      desc->i_begin / i_count / y_ld must be 0; y is I x [1|N] x [1|M].  The reference has no multi-device path (one gpuDevice
 *      per MATLAB process, README.md:232): this is what lets its single-process host reach a whole node. */
typedef struct qdas_sharded_plan qdas_sharded_plan;
int  qdas_plan_create_sharded(qdas_sharded_plan **plan, const qdas_desc *desc, int ndev, const int *devices);
int  qdas_plan_execute_sharded(qdas_sharded_plan *plan, const void *x, void *y, void *stream);
/* shard < 0: the number of shards in *device.  Else the shard's device, pixel slab and kernel (QDAS_KERNEL_*; 0: empty slab). */
int  qdas_plan_sharded_info(const qdas_sharded_plan *plan, int shard, int *device, uint64_t *i_begin, uint64_t *i_count, int *kernel);
/* 1 when the plan runs MIRROR SLABS: shard g then owns the slab qdas_plan_sharded_info reports (whole columns of the first half of the image)
 * AND its mirror image, the pixels [I - i_begin - i_count, I - i_begin); 0: plain contiguous slabs */
int  qdas_plan_sharded_mirror(const qdas_sharded_plan *plan);
void qdas_plan_destroy_sharded(qdas_sharded_plan *plan);

/* ---- one-shot entries shaped like the reference kernels' argument lists
 *      (device pointers; sizes struct replaces the constant-memory symbols).
 *      DAS  <-> src/bf.cu:144-151, DASf <-> :153-161, DASh <-> :164-171 */
int qdas_DAS (const qdas_sizes *sz, void *y, const double *Pi, const double *Pr, const double *Pv,
              const double *Nv, const void *a, const double *cinv, const uint64_t *acstride_host,
              const void *x, const double tvars[2], void *stream);
int qdas_DASf(const qdas_sizes *sz, void *y, const float *Pi, const float *Pr, const float *Pv,
              const float *Nv, const void *a, const float *cinv, const uint64_t *acstride_host,
              const void *x, const float tvars[2], void *stream);
int qdas_DASh(const qdas_sizes *sz, void *y, const float *Pi, const float *Pr, const float *Pv,
              const float *Nv, const void *a, const float *cinv, const uint64_t *acstride_host,
              const void *x, const float tvars[2], void *stream);
/* delays <-> src/bf.cu:255-298, delaysf <-> :209-252 (tau is I x N x M) */
int qdas_delays (const qdas_sizes *sz, double *tau, const double *Pi, const double *Pr,
                 const double *Pv, const double *Nv, double cinv, void *stream);
int qdas_delaysf(const qdas_sizes *sz, float *tau, const float *Pi, const float *Pr,
                 const float *Pv, const float *Nv, float cinv, void *stream);

/* ---- split-delay flavour ("next" row): what bfDASLUT ->
 *      ChannelData.sample2sep -> wsinterpd2 computes (reference src/UltrasoundSystem.m:4641-4660,
 *      src/ChannelData.m:1431-1445, kern/wsinterpd2.m:236, src/interpd.cu:344-396) with owned
 *      outputs instead of float atomics:
 *        y[i,(n),(m)] = sum w[i,n,m] * exp(j*omega*s) * sample(x[:,n,m], s),
 *        s = tau_rx[i,n] + tau_tx[i,m]        (sample units, already (tau - t0)*fs)
 *      non-finite s are skipped (src/interpd.cu:390). */
typedef struct qdas_lut_desc {
    uint64_t T, N, M, I;      /* I pixels (any shape, flattened)                            */
    int32_t  flag;            /* interp bits + QDAS_FLAG_KEEP_RX/TX (+ TPOSE for the data)   */
    int32_t  dtype;           /* QDAS_F64 | QDAS_F32 | QDAS_F16 (delays: double|float|float) */
    double   omega;           /* imag(omega) = 2*pi*fmod/fs (src/ChannelData.m:1439)         */
    const void *tau_rx;       /* I x N real, device                                          */
    const void *tau_tx;       /* I x M real, device                                          */
    const void *w;            /* complex(prec) weights or NULL; strides below               */
    uint64_t wstride[3];      /* element strides of w for (i, n, m); 0 where singleton       */
    int32_t  w_real;          /* weights are real(prec)                                      */
    int32_t  reserved;
    uint64_t I1;              /* size of the fastest pixel dimension (the image is I1 x I/I1; 0: unknown) -- lets fp32 full-sum
                                 calls run on the fused tiled kernel, whose tiles must be compact in depth */
} qdas_lut_desc;
int qdas_das_lut(const qdas_lut_desc *d, const void *x, void *y, void *stream);
/* which kernel served the calling thread's last qdas_das_lut (diagnostic, like qdas_plan_kernel_name): "tiled,mirror [jit <key>]" (tables that are their own
 * lateral mirror images, checked bit for bit per call: a pixel and its image share tap index and weights -- fp32 full sums without weights), "tiled", "generic" */
int qdas_das_lut_last_kernel(char *buf, size_t len);

/* ---- General single-delay flavour: weighted, phase-rotated sampling over an N-D broadcast index space.  Replaces the launches of
 *      wsinterpd[f|h] (reference kern/wsinterpd.m:221-236, kernel src/interpd.cu:295-342) and interpd[f|h] (kern/interpd.m,
 *      src/interpd.cu:169-192) -- what ChannelData.sample (src/ChannelData.m:1230-1336) and focusTx run on:
 *        y[kept] = sum over summed dims of  w[j] * exp(i*omega*t[j]) * sample(x[:, j], t[j])        (t in samples, 0-based)
 *      The index space has ndim <= 8 dimensions of sizes size[]; dimension 0 is the sampling dimension (size[0] samples of t
 *      against T samples of x).  t, w and the trace bases of x address it through element strides (0 = broadcast): the reference's
 *      matching / outer dimension classification (kern/wsinterpd.m:70-93) reduces to these.  y is dense, column-major over the kept
 *      dimensions (summed ones have size 1).  Infinite t are skipped; samples outside the record give `extrap` (NaN allowed; sums
 *      omit NaN like the reference's sum(..., 'omitnan')).  All pointers are device pointers of `dtype` (t: double | float | float). */
typedef struct qdas_wsinterpd_desc {
    uint64_t T;               /* samples per trace of x                                             */
    uint64_t x_tstride;       /* element stride between consecutive samples of a trace (1: contiguous) */
    int32_t  ndim;            /* 1..8                                                               */
    int32_t  flag;            /* QDAS_INTERP_*                                                      */
    int32_t  dtype;           /* QDAS_F64 | QDAS_F32 | QDAS_F16                                     */
    int32_t  w_real;          /* weights are real(prec)                                             */
    uint64_t size[8];
    int64_t  tstride[8], xstride[8], wstride[8];   /* xstride[0] must be 0 (the sampling dimension indexes t, not x) */
    uint8_t  sum[8];          /* 1: summed dimension                                                */
    double   omega;           /* imag(omega)                                                        */
    double   extrap;          /* value of out-of-record samples                                     */
    const void *t, *w, *x;    /* w may be NULL                                                      */
    /* extension (round 4): the layout of y.  ystride all 0: dense column-major over the kept dimensions (the reference's output).  Else element
       strides per dimension (summed dimensions ignored): lets a row-major / arbitrarily strided x be sampled IN PLACE into a y of the same layout
       -- together with x_tstride / xstride no layout pass is needed.  `lane_dim`: the kept dimension the lanes of a wave run along (choose the
       one along which x -- or, where x broadcasts, t -- is contiguous); -1, a summed dimension or a dimension of one element (so also the 0 of a
       zero-initialised descriptor whose first dimension is a singleton): the first kept dimension with more than one element.                   */
    int64_t  ystride[8];
    int32_t  lane_dim;
    int32_t  reserved;
} qdas_wsinterpd_desc;
int qdas_wsinterpd(const qdas_wsinterpd_desc *d, void *y, void *stream);

/* ---- misc */
/* ---- Point-scatterer channel-data simulator: replaces the kernels greens / greensf
 * (reference src/greens.cu:88-121, body :8-86) launched by UltrasoundSystem.greens
 * (src/UltrasoundSystem.m:681-718: k.feval(x, ps, as, pn, pv, kn, sb, blocks, [t0k t0x fso fsr cinv R0], [E E], flagnum)).
 *   y[s,n,m] = 1/fsr * sum_i sum_ne sum_me a_i * sample(x, fsr*(s - (cinv*(r1+r2) + t0 - s0)*fs)) / (max(r1,R0)*max(r2,R0))
 * All pointers are DEVICE pointers; real = double (QDAS_F64) or float (QDAS_F32), complex = interleaved pairs.
 * The reference's `sb` / `blocks` arguments (per-scatterer sample windows for culling) are not needed: the kernel
 * derives the windows itself.  R0 == 0: no propagation loss (the reference's CPU branch, :797-803).
 * Two kernels, the same sum: per (entry, sample), and -- fp32 data, an integer fsr, at least QDAS_GREENS_TRAIN_MIN (1024) entries per trace -- impulse
 * trains: the interpolation fraction is one per entry, so every entry adds its K weighted amplitudes to K fixed-point trains (64-bit integer LDS atomics:
 * the result does not depend on the order of arrival, bit-reproducible) and one dense convolution with the waveform per block of samples follows.  The two
 * agree to fp32 re-association (~1e-6 of the peak); an entry whose fp32 delay lies on a tap boundary may land on either side (csrc/greens.hip). */
typedef struct qdas_greens_desc {
    uint64_t S;            /* output samples per trace                          (QUPS_S) */
    uint64_t T;            /* samples of the waveform x                          (QUPS_T) */
    uint64_t N, M, I;      /* receivers, transmitters, scatterers                         */
    int32_t  En, Em;       /* sub-apertures per receive / transmit element       (E)      */
    int32_t  interp;       /* QDAS_INTERP_*                                      (iflag)  */
    int32_t  dtype;        /* QDAS_F64 | QDAS_F32                                         */
    double   s0;           /* time of output sample 0 [s]                        (t0k)    */
    double   t0;           /* time of waveform sample 0 [s]                      (t0x)    */
    double   fs;           /* output sampling frequency [Hz]                     (fso)    */
    double   fsr;          /* waveform sampling frequency / fs                            */
    double   cinv;         /* 1 / sound speed                                             */
    double   R0;           /* minimum distance of the 1/(r1 r2) loss; 0 = no loss         */
    const void *Ps;        /* 3 x I   scatterer positions, real                           */
    const void *a;         /* I       scatterer amplitudes, complex                       */
    const void *Pr;        /* 3 x N x En receive (sub-)element positions, real            */
    const void *Pv;        /* 3 x M x Em transmit (sub-)element positions, real           */
    const void *x;         /* T       waveform samples, complex                           */
    int32_t  device;       /* HIP device ordinal, -1 = current                            */
    int32_t  reserved;
} qdas_greens_desc;
int qdas_greens(const qdas_greens_desc *desc, void *y /* S x N x M complex */, void *stream);

/* ---- Transmit synthesis (UltrasoundSystem.focusTx, reference src/UltrasoundSystem.m:3374-3503): delay-and-sum over the transmit ELEMENTS of a
 * full-synthetic-aperture record,
 *     y[t', n, m'] = sum_m  w[m, m'] * x(t' + shift[m, m'],  n, m),      t' = 0 .. To-1,
 * the call the reference makes as sample2sep(chd.time, -tau, interp, apd, mdim) (:3498 -> kern/wsinterpd2.m -> src/interpd.cu:344-396) with the
 * positions written as the record's own time grid plus one offset per (element, synthesised transmit) [samples; shift = -tau * fs after the
 * reference's re-basing of the time axis, :3457-3463].  qdas_wsinterpd computes the same numbers from a materialised position array; this entry
 * uses the structure (per-pair tap offset and weights, LDS-staged windows, uniform skips of zero weights; csrc/shiftsum.hip).
 * x: T x N x M x F, y: To x N x Mo x F column-major DEVICE arrays of `dtype` (QDAS_F64 | QDAS_F32), complex or real; shift: M x Mo real(dtype),
 * w: M x Mo real(dtype) or complex(dtype) or NULL (ones) -- DEVICE arrays, m fastest.  Real data take real weights.
 * Edge rule as in every kernel: a term counts iff all its taps lie in [0, T) and its position is >= 0. */
typedef struct qdas_shift_desc {
    uint64_t T, To, N, M, Mo, F;
    int32_t  flag;      /* interpolation, bits 0-2 of QUPS_BF_FLAG */
    int32_t  dtype;     /* QDAS_F64 | QDAS_F32                     */
    int32_t  cplx;      /* samples are interleaved complex         */
    int32_t  w_real;    /* w holds real(dtype) weights             */
    int32_t  device;    /* HIP device ordinal, -1 = current        */
    int32_t  tpad;      /* the last tpad of the T samples of a trace are ZEROS that are not stored: x holds T - tpad samples per trace (that is its stride);
                         * a tap in the tail counts as an in-range zero -- ChannelData.zeropad in front of the sampling (src/UltrasoundSystem.m:3479) without the copy */
    const void *shift;
    const void *w;
} qdas_shift_desc;
int qdas_shift_sum(const qdas_shift_desc *desc, const void *x, void *y, void *stream);

/* ---- Batched 1-D convolution along one dimension (band-pass / matched filtering of the traces in front of DAS).
 * Replaces the kernels conv / convf / convc / convcf (reference src/convd.cu:95-127,130-146) launched by kern/convd.m:150-199
 * (kern.feval(x, y, z, sizes) with the constant L0).  x: C x M x S, y: C x N x S, z: C x L x S, column-major, all DEVICE pointers,
 * real or interleaved complex of `dtype`; C = product of the dimensions in front of the convolved one, S = of those behind it:
 *   z[c,l,s] = sum_i x[c,i,s] * y[c, l + off - i, s]
 * 'full': L = M+N-1, off = 0; 'same': L = M, off = N-1 - floor((N-1)/2); 'valid': L = max(M-N+1, 0), off = N-1
 * (kern/convd.m:103-114).  A singleton column / slice dimension of x or y is broadcast (bits of `bcast`) instead of being
 * replicated as the reference does (kern/convd.m:75-84).  QDAS_F16: the half-precision twins convh / convch (src/convd.cu:141,153) --
 * products and sums in fp32, rounded once at the store (the reference accumulates in half).
 * Long filters: complex64 traces with C == 1 (time contiguous), one filter for all slices, N >= QDAS_CONV_FFT_MIN_TAPS (128) and M + N - 1 <= 8192 take an
 * FFT convolution with the trace resident in LDS (csrc/pre.hip): the same outputs to fp32 rounding (a few 1e-6 of the largest), O(L log L) per trace. */
#define QDAS_CONV_FULL  0
#define QDAS_CONV_SAME  1
#define QDAS_CONV_VALID 2
#define QDAS_CONV_CAUSAL 3         /* extension: L = M, off = 0 -- the first M samples of the full convolution = MATLAB filter(b, 1, x), what
                                      ChannelData.filter applies with an FIR digitalFilter (reference src/ChannelData.m:857-880)            */
#define QDAS_CONV_X_ONE_COLUMN 1   /* x is 1 x M x (S | 1) */
#define QDAS_CONV_X_ONE_SLICE  2   /* x is (C | 1) x M x 1 */
#define QDAS_CONV_Y_ONE_COLUMN 4
#define QDAS_CONV_Y_ONE_SLICE  8
typedef struct qdas_convd_desc {
    uint64_t C, M, N, S;
    int32_t  dtype;    /* QDAS_F64 | QDAS_F32 | QDAS_F16         */
    int32_t  cplx;     /* 0: real data, 1: interleaved complex   */
    int32_t  shape;    /* QDAS_CONV_*                            */
    int32_t  bcast;    /* QDAS_CONV_{X,Y}_ONE_{COLUMN,SLICE}     */
    int32_t  device;   /* HIP device ordinal, -1 = current       */
    int32_t  y_real;   /* extension (cplx == 1): y holds REAL taps of `dtype` -- a real filter on complex traces costs half the multiplies
                          of the promoted product the reference forms (kern/convd.m:259-268 makes both operands complex) */
} qdas_convd_desc;
uint64_t qdas_convd_len(uint64_t M, uint64_t N, int shape);    /* L */
int qdas_convd(const qdas_convd_desc *desc, const void *x, const void *y, void *z, void *stream);

/* ---- Recursive (IIR) filtering along time: what ChannelData.filter applies with an IIR digitalFilter (reference src/ChannelData.m:857-888:
 * filter(D, x) -- a cascade of second-order sections, each the direct-form II transposed recursion -- then t0 -= filtord(D) / fs).
 * x, y: T x K DEVICE arrays of `dtype` (QDAS_F64 | QDAS_F32), real or interleaved complex, time contiguous (K = every other dimension flattened); y may be x.
 * sos: HOST array, nsec x 6 row-major [b0 b1 b2 a0 a1 a2] per section (MATLAB's D.Coefficients / scipy's sos); gain: overall scale (0 = 1).
 * One read and one write of the record whatever the order (iir.hip); the state is kept in double. */
typedef struct qdas_iir_desc {
    uint64_t T, K;
    int32_t  nsec;     /* 1 .. 16 second-order sections */
    int32_t  dtype;    /* QDAS_F64 | QDAS_F32           */
    int32_t  cplx;     /* samples are interleaved complex */
    int32_t  device;   /* HIP device ordinal, -1 = current */
    double   gain;
    const double *sos;
} qdas_iir_desc;
int qdas_iir(const qdas_iir_desc *desc, const void *x, void *y, void *stream);

/* ---- Temporaries of the stream entries (qdas_shift_sum, qdas_das_lut, qdas_greens, qdas_convd's FFT path): taken from an arena the library keeps per (device,
 * stream).  One such call at a time runs per (device, stream) -- a second thread on the same stream waits --, and a call MAY BLOCK the host: when the stream's
 * previous call outgrew the arena (the next call waits for it, then regrows the arena to what that call needed, up to 512 MiB) or asks for a single temporary above
 * 64 MiB (freed, after a stream synchronisation, when the call returns).  Otherwise the calls only enqueue work.  qdas_device_trim releases idle arenas. */

/* ---- Device staging for HOST callers of the device-pointer entries above (qdas_delays*, qdas_das_lut, qdas_wsinterpd, qdas_greens, qdas_convd,
 * qdas_pre_execute ...): the reference reaches those kernels with gpuArrays (kern/wsinterpd2.m:236, src/UltrasoundSystem.m:681-718, kern/convd.m:150-199),
 * a MEX gateway built WITHOUT the mxGPUArray API has host arrays only and must not need the HIP headers -- it allocates, copies and frees through
 * these three (mex/qdas_mex.c dev_in / dev_out).  device: HIP ordinal, -1 = current.  qdas_device_copy is synchronous; kind 0: host -> device,
 * 1: device -> host, 2: device -> device. */
int qdas_device_malloc(void **p, size_t bytes, int device);
int qdas_device_free(void *p, int device);      /* (buffers of qdas_device_malloc are kept for reuse, at most 4 GiB -- QDAS_STAGING_CACHE_MB --: a call-per-launch gateway does not map / unmap.
                                                 *  Like hipFree, qdas_device_free WAITS for the buffer's device -- every stream, blocking or not -- before the buffer can be handed
                                                 *  out again: work the caller launched on any stream may still be using it when it is freed.) */
int qdas_device_trim(void);                      /* ... and released here */
int qdas_device_copy(void *dst, const void *src, size_t bytes, int kind, int device);

/* ---- Layout conversion for row-major hosts (numpy / torch; no reference counterpart: MATLAB arrays are column-major already and
 * the ABI follows the reference's memory order, e.g. kern/das_spec.m:367-372 passes x(:,:,:,f) as it lies in memory).
 * out[c][b][a] = in[a][b][c] for a row-major A x B x C array of elem_bytes-sized elements (2 | 4 | 8 | 16): the column-major
 * image of T x N x M channel data or I1 x I2 x N delay tables.  Device pointers, in != out. */
int qdas_permute3(const void *in, void *out, uint64_t A, uint64_t B, uint64_t C, int elem_bytes, void *stream);

/* ---- Pre-processing in front of the DAS path: real RF traces -> analytic channel data, optionally downmixed.
 * Replaces ChannelData.hilbert (reference src/ChannelData.m:935-966: fft to N points along time, weights
 * [1; 2...; 1 + mod(N,2); 0...], ifft) and ChannelData.downmix (src/ChannelData.m:757-766: data .* exp(-2i*pi*fc*time))
 * applied after it -- one kernel, two real traces per complex transform, forward and inverse FFT stages LDS to LDS (pre.hip).
 * x: T x K real traces (K = N*M*F, device pointer) as fp32 or int16; y: Nfft x K complex64 (device).
 * Real input is half (fp32) / a quarter (int16) of the bytes of complex64 channel data on the PCIe link. */
#define QDAS_PRE_F32 0
#define QDAS_PRE_I16 1
typedef struct qdas_pre_desc {
    uint64_t T;        /* samples per input trace                                   */
    uint64_t K;        /* number of traces                                          */
    uint64_t Nfft;     /* transform length = output samples per trace (0 = T)       */
    int32_t  in_type;  /* QDAS_PRE_F32 | QDAS_PRE_I16                               */
    int32_t  device;   /* HIP device ordinal, -1 = current                          */
    double   fs, t0;   /* sampling frequency, time of sample 0 (downmix only)       */
    double   fdown;    /* downmix frequency [Hz]; 0 = no downmix                    */
} qdas_pre_desc;
typedef struct qdas_pre_plan qdas_pre_plan;
int  qdas_pre_plan_create(qdas_pre_plan **plan, const qdas_pre_desc *desc);
int  qdas_pre_execute(qdas_pre_plan *plan, const void *x, void *y, void *stream);
void qdas_pre_plan_destroy(qdas_pre_plan *plan);
/* 1: the plan runs the one-pass kernel (a trace pair resident in LDS: Nfft = 2^a 3^b 5^c 7^d 11^e 13^f <= 8192 whose stages fit 1024 threads; HBM sees the input once and
 * the output once); 0: hipFFT passes (any other length, or QDAS_PRE_HIPFFT=1 in the environment) */
int  qdas_pre_plan_one_pass(const qdas_pre_plan *plan);

/* ---- Variants of the fused kernel that are built on demand.  libqdas.so carries the instantiations the BASELINE configurations and frame streams
 * launch (csrc/das_tile_cfg.h TILE_PREBUILT); any other point of the template's matrix -- launch configuration `cfg` (das_tile_cfg.h) x interpolator
 * flag x remodulation x weight table -- is compiled by hiprtc when a plan first needs it (qdas_plan_create; ~2 s, then cached in memory and under
 * $QDAS_CACHE_DIR | ~/.cache/qdas), as the reference compiles its kernels per UltrasoundSystem (src/UltrasoundSystem.m:5527-5625).  This entry
 * builds one variant ahead of time (deployment, test sessions; `python -m qups_amd.warm` runs it over a list with one process per core).
 * Returns 0: built or already cached, 1: failed (qdas_last_error), 2: libqdas.so carries it, 3: no such variant.  Needs no device.
 * Without libhiprtc.so (or with QDAS_NO_LAZY=1) a plan that needs such a variant runs the generic kernel; qdas_last_error() after
 * qdas_plan_create says so, and a QDAS_KERNEL_TILED request fails with QDAS_EUNSUPPORTED. */
int  qdas_kernel_variant_build(int cfg, int interp, int fmod, int wtab);
/* 1: libqdas.so carries the variant, 0: it is built on demand, -1: no such variant */
int  qdas_kernel_variant_prebuilt(int cfg, int interp, int fmod, int wtab);

const char *qdas_last_error(void);
int  qdas_version(void);
/* device properties the host side reports next to measurements */
int  qdas_device_info(int device, char *name, size_t name_len, int *cu_count, int *clock_khz,
                      uint64_t *hbm_bytes);

#ifdef __cplusplus
}
#endif
#endif /* QDAS_H */
