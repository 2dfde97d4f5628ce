/* qdas_mex.c -- thin MEX gateway from MATLAB to libqdas.so (C ABI in include/qdas.h).
 *
 * MATLAB is not available in this repository's environment; this is the binding a QUPS maintainer builds,
 *
 *     mex -R2018a -I<repo>/include -L<repo>/qups_amd -lqdas mex/qdas_mex.c -output bin/qdas_mex
 *     mex -R2018a -DQDAS_MEX_GPU -I... -L... -lqdas -lmwgpu mex/qdas_mex.c -output bin/qdas_mex      % gpuArray inputs / outputs
 *
 * and what tests/test_mex_gateway.py compiles against a small stand-in for mex.h (tests/fake_mex/: a syntax / ABI / logic check of
 * THIS file, not a MATLAB).  It replaces ONE call site of the reference: the kernel object + constant upload + per-frame launch
 * inside das_spec (kern/das_spec.m:279-306, :367-378), including the reusable handle the reference returns to callers that stream
 * frames, [k, PRE_ARGS, POST_ARGS] (kern/das_spec.m:72-81, 387-390):
 *
 *   y = qdas_mex(sizes, Pi, Pr, Pv, Nv, apod, cinv, acstride, x, tvars)               one call = create + execute + destroy
 *   h = qdas_mex('create', sizes, Pi, Pr, Pv, Nv, apod, cinv, acstride, tvars [, opts]) plan handle (uint64 scalar), kept until
 *   y = qdas_mex('execute', h, x)                                                       'destroy' / clear mex; x: T x N x M x F
 *       qdas_mex('prepare', h, F)                                                       do now what the plan's first stream of F frames would do once (qdas_plan_prepare_frames)
 *   s = qdas_mex('info', h)                                                             char: kernel name, shards
 *       qdas_mex('destroy' [, h])
 *
 * The other launch sites of the path, one command each (stateless; arguments in the order of the reference's own k.feval lists):
 *   tau = qdas_mex('delays', sizes, Pi, Pr, Pv, Nv, cinv)                 kern/das_spec.m:376-377  k.feval(yg, Pi, Pr, Pv, Nv, cinv(1))       -> I x N x M real(prec)
 *   tau = qdas_mex('delays', h)                                           the same from a live plan (its geometry is on the device already)
 *   y   = qdas_mex('lut', lsizes, w, x, t1, t2, wstride, omega)           kern/wsinterpd2.m:236    k.feval(y, w, x, t1, t2, sizes, iflags, strides, flagnum, imag(omega))
 *         as bfDASLUT -> sample2sep calls it (src/UltrasoundSystem.m:4641-4660): lsizes = [T N M I flag dtype I1 w_real], t1 = receive table I x N, t2 = transmit
 *         table I x M (samples), w = [] or weights with element strides wstride = uint64 [si sn sm] (0 where singleton); y is I x [1|N] x [1|M] complex(prec)
 *   y   = qdas_mex('wsinterpd', wsz, size, strides, sumdims, w, x, t, tvars)   kern/wsinterpd.m:221-236  k.feval(y, w, x, t, sizes, iflags, strides, flagnum, imag(omega))
 *         the general single-delay launch (ChannelData.sample, rectifyt0): wsz = [T x_tstride ndim flag dtype w_real], size = the ndim sizes of the broadcast index
 *         space (dimension 1 = the sampled one), strides = int64 3 x ndim element strides of [t; x; w] (0 where singleton; the trace bases of x: row 2, entry 1 is 0),
 *         sumdims = 1 x ndim flags, tvars = [imag(omega) extrapval]; y is dense column-major over the kept dimensions (summed ones singleton), complex(prec)
 *   y   = qdas_mex('shiftsum', ssz, x, shift, w)                               src/UltrasoundSystem.m:3498 (focusTx: sample2sep(chd.time, -tau, interp, apd, mdim))
 *         ssz = [T To N M Mo F flag dtype cplx w_real]; x: T x N x M x F, shift: M x Mo real(prec) samples, w: M x Mo real or complex(prec) or []; y: To x N x Mo x F
 *   y   = qdas_mex('greens', gsizes, ps, as, pn, pv, x, tvars)            src/UltrasoundSystem.m:681-718  k.feval(x, ps, as, pn, pv, kn, sb, blocks, [t0k t0x fso fsr cinv R0], [E E], flagnum)
 *         gsizes = [S T N M I En Em flagnum dtype], tvars = [t0k t0x fso fsr cinv R0]; the sb / blocks culling tables are not needed; y is S x N x M complex
 *   z   = qdas_mex('convd', csizes, x, y)                                 kern/convd.m:150-199     kern.feval(x, y, z, sizes)
 *         csizes = [C M N S dtype cplx shape bcast y_real] (include/qdas.h qdas_convd_desc); z is C x L x S
 *   y   = qdas_mex('hilbert', psizes, x, tvars)                           src/ChannelData.m:935-966 (+ downmix :757-766)
 *         psizes = [T K Nfft in_type], tvars = [fs t0 fdown]; x real single or int16 T x K; y is Nfft x K complex single
 * Host arrays are staged through device memory by the gateway (qdas_device_malloc / _copy / _free: no HIP headers needed); with -DQDAS_MEX_GPU gpuArrays
 * pass as device pointers and the result is a gpuArray.
 *
 *   sizes    : numeric row [T N M I1 I2 I3 S flag VS DV dtype]  (dtype: 0 double, 1 single, 2 half)   -- any real numeric class;
 *              the QUPS_* constants of kern/das_spec.m:294-298.  The one-call form takes a 12th entry F (frames).
 *   Pi..Nv   : real(prec) arrays laid out 3 x I, 3 x N, 4 x M (row 4 = t0, kern/das_spec.m:361), 3 x M
 *   apod     : complex(prec) column (concatenated arrays, kern/das_spec.m:344-345) or [] ; cinv: real(prec) array
 *   acstride : uint64 6 x (1+S)  [cstride, astride]                                     (kern/das_spec.m:257-260)
 *   tvars    : [fs fmod], any real numeric class
 *   opts     : struct, all fields optional: devices (row of device ordinals: a multi-GPU plan, qdas_plan_create_sharded),
 *              jit (true: QDAS_PLAN_JIT), reciprocal (false: QDAS_PLAN_NO_RECIPROCAL), mirror (false: QDAS_PLAN_NO_MIRROR), fold (false: QDAS_PLAN_NO_FOLD),
 *              approx_symmetry (true: QDAS_PLAN_APPROX_SYMMETRY), kernel (0 auto | 1 generic | 2 tiled)
 *   x, y     : complex(prec); half precision travels as uint16 arrays with a leading dimension of 2 (re, im), as the reference
 *              passes ushort2 (src/bf.cu:164-171)
 * Host arrays use QDAS_MEM_HOST (the library stages them through HBM).  With -DQDAS_MEX_GPU, gpuArray arguments are passed as
 * device pointers (QDAS_MEM_DEVICE, mxGPUArray API) and y is returned as a gpuArray: no gather() of the 1.5 GB frame.  A plan is
 * created for ONE memory kind (that of Pi); 'execute' refuses an x of the other kind.
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "mex.h"
#ifdef QDAS_MEX_GPU
#include "gpu/mxGPUArray.h"
#endif
#include "qdas.h"

#define QDAS_MEX_MAX_PLANS 64

typedef struct {
    int used;
    qdas_plan *plan;
    qdas_sharded_plan *splan;
    qdas_sizes sz;
    int mem;                                            /* QDAS_MEM_* of the plan */
} plan_slot;

static plan_slot g_slots[QDAS_MEX_MAX_PLANS];
static int g_nlive = 0, g_atexit = 0;

static void destroy_slot(plan_slot *s) {
    if (!s->used) return;
    if (s->plan) qdas_plan_destroy(s->plan);
    if (s->splan) qdas_plan_destroy_sharded(s->splan);
    memset(s, 0, sizeof *s);
    if (--g_nlive == 0 && mexIsLocked()) mexUnlock();
}
static void destroy_all(void) { for (int k = 0; k < QDAS_MEX_MAX_PLANS; ++k) destroy_slot(&g_slots[k]); }

/* element k of a real numeric array of any class, as double (sizes / tvars / opts may arrive as double, single or integers) */
static double num_at(const mxArray *a, mwSize k, const char *what) {
    if (!a || mxIsComplex(a) || k >= mxGetNumberOfElements(a)) mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "%s: real numeric array with more than %d entries expected.", what, (int)k);
    const void *p = mxGetData(a);
    switch (mxGetClassID(a)) {
        case mxDOUBLE_CLASS: return ((const double *)p)[k];
        case mxSINGLE_CLASS: return (double)((const float *)p)[k];
        case mxINT64_CLASS:  return (double)((const int64_t *)p)[k];
        case mxUINT64_CLASS: return (double)((const uint64_t *)p)[k];
        case mxINT32_CLASS:  return (double)((const int32_t *)p)[k];
        case mxUINT32_CLASS: return (double)((const uint32_t *)p)[k];
        case mxLOGICAL_CLASS: return (double)((const unsigned char *)p)[k];
        default: mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "%s: unsupported numeric class.", what);
    }
    return 0.0;
}

static mxClassID class_of(int dtype) { return dtype == QDAS_F64 ? mxDOUBLE_CLASS : (dtype == QDAS_F32 ? mxSINGLE_CLASS : mxUINT16_CLASS); }

/* data pointer of an argument; *is_dev tells whether it is a gpuArray (only with QDAS_MEX_GPU) */
typedef struct { const void *ptr; int is_dev;
#ifdef QDAS_MEX_GPU
    const mxGPUArray *g;
#endif
} argptr;

static argptr arg_data(const mxArray *a) {
    argptr r;
    memset(&r, 0, sizeof r);
    if (!a || mxIsEmpty(a)) return r;
#ifdef QDAS_MEX_GPU
    if (mxIsGPUArray(a)) { r.g = mxGPUCreateFromMxArray(a); r.ptr = mxGPUGetDataReadOnly(r.g); r.is_dev = 1; return r; }
#endif
    r.ptr = mxGetData(a);
    return r;
}
static void arg_release(argptr *r) {
#ifdef QDAS_MEX_GPU
    if (r->g) mxGPUDestroyGPUArray(r->g);
#endif
    (void)r;
}

static void read_sizes(const mxArray *sz, qdas_sizes *z, uint64_t *F) {
    const mwSize n = mxGetNumberOfElements(sz);
    if (n < 11) mexErrMsgIdAndTxt("QUPS:das_spec:sizes", "sizes must have at least 11 entries [T N M I1 I2 I3 S flag VS DV dtype].");
    z->T = (uint64_t)num_at(sz, 0, "sizes");  z->N = (uint64_t)num_at(sz, 1, "sizes");  z->M = (uint64_t)num_at(sz, 2, "sizes");
    z->I1 = (uint64_t)num_at(sz, 3, "sizes"); z->I2 = (uint64_t)num_at(sz, 4, "sizes"); z->I3 = (uint64_t)num_at(sz, 5, "sizes");
    z->S = (uint64_t)num_at(sz, 6, "sizes");  z->flag = (int32_t)num_at(sz, 7, "sizes");
    z->VS = (int32_t)num_at(sz, 8, "sizes");  z->DV = (int32_t)num_at(sz, 9, "sizes");  z->dtype = (int32_t)num_at(sz, 10, "sizes");
    if (F) *F = n >= 12 ? (uint64_t)num_at(sz, 11, "sizes") : 1;
}

/* create a plan from prhs[0..8] = sizes, Pi, Pr, Pv, Nv, apod, cinv, acstride, tvars (+ optional opts); returns the slot index */
static int create_plan(int nrhs, const mxArray *prhs[]) {
    if (nrhs < 9) mexErrMsgIdAndTxt("QUPS:das_spec:nargin", "qdas_mex('create', ...) expects sizes, Pi, Pr, Pv, Nv, apod, cinv, acstride, tvars [, opts].");
    int slot = -1;
    for (int k = 0; k < QDAS_MEX_MAX_PLANS; ++k) if (!g_slots[k].used) { slot = k; break; }
    if (slot < 0) mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "too many live plans (destroy some).");
    qdas_desc d;
    memset(&d, 0, sizeof d);
    read_sizes(prhs[0], &d.sz, NULL);
    if (mxGetClassID(prhs[7]) != mxUINT64_CLASS || mxGetNumberOfElements(prhs[7]) < 6 * (1 + d.sz.S))
        mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "acstride must be uint64 with 6 x (1 + S) entries.");
    argptr a[6];
    const int idx[6] = {1, 2, 3, 4, 5, 6};
    for (int k = 0; k < 6; ++k) a[k] = arg_data(prhs[idx[k]]);
    d.Pi = a[0].ptr; d.Pr = a[1].ptr; d.Pv = a[2].ptr; d.Nv = a[3].ptr; d.apod = a[4].ptr; d.cinv = a[5].ptr;
    const int dev = a[0].is_dev;
    for (int k = 1; k < 6; ++k)
        if (a[k].ptr && a[k].is_dev != dev) mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "Pi, Pr, Pv, Nv, apod, cinv must all be gpuArrays or all be host arrays.");
    d.acstride = (const uint64_t *)mxGetData(prhs[7]);       /* always a host array */
    d.fs = num_at(prhs[8], 0, "tvars"); d.fmod = num_at(prhs[8], 1, "tvars");
    d.mem = dev ? QDAS_MEM_DEVICE : QDAS_MEM_HOST;
    d.device = -1; d.kernel = QDAS_KERNEL_AUTO;
    /* gpuArray inputs: the handle outlives this call but the caller's gpuArrays do not (das_spec's workspace is freed when it
       returns `k = h`), and the mxGPUArray views are released below -- the plan must own device copies (qdas.h, LIFETIME) */
    if (dev) d.plan_flags |= QDAS_PLAN_COPY_INPUTS;
    int ndev = 0, devices[64];
    if (nrhs >= 10 && !mxIsEmpty(prhs[9])) {
        const mxArray *o = prhs[9], *f;
        if (!mxIsStruct(o)) mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "opts must be a struct.");
        if ((f = mxGetField(o, 0, "jit")) && num_at(f, 0, "opts.jit") != 0) d.plan_flags |= QDAS_PLAN_JIT;
        if ((f = mxGetField(o, 0, "reciprocal")) && num_at(f, 0, "opts.reciprocal") == 0) d.plan_flags |= QDAS_PLAN_NO_RECIPROCAL;
        if ((f = mxGetField(o, 0, "mirror")) && num_at(f, 0, "opts.mirror") == 0) d.plan_flags |= QDAS_PLAN_NO_MIRROR;
        if ((f = mxGetField(o, 0, "fold")) && num_at(f, 0, "opts.fold") == 0) d.plan_flags |= QDAS_PLAN_NO_FOLD;
        if ((f = mxGetField(o, 0, "approx_symmetry")) && num_at(f, 0, "opts.approx_symmetry") != 0) d.plan_flags |= QDAS_PLAN_APPROX_SYMMETRY;
        if ((f = mxGetField(o, 0, "kernel"))) d.kernel = (int32_t)num_at(f, 0, "opts.kernel");
        if ((f = mxGetField(o, 0, "devices"))) {
            ndev = (int)mxGetNumberOfElements(f);
            if (ndev > 64) mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "at most 64 devices.");
            for (int k = 0; k < ndev; ++k) devices[k] = (int)num_at(f, (mwSize)k, "opts.devices");
        }
    }
    plan_slot *s = &g_slots[slot];
    int rc;
    if (ndev > 0) { d.device = devices[0]; rc = qdas_plan_create_sharded(&s->splan, &d, ndev, devices); }
    else rc = qdas_plan_create(&s->plan, &d);
    for (int k = 0; k < 6; ++k) arg_release(&a[k]);
    if (rc) mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "%s", qdas_last_error());
    s->used = 1; s->sz = d.sz; s->mem = d.mem;
    if (g_nlive++ == 0) { if (!g_atexit) { mexAtExit(destroy_all); g_atexit = 1; } mexLock(); }   /* plans outlive the call */
    return slot;
}

/* y = execute(slot, x): F frames of T x N x M.  Returns NULL on success, else the message to raise -- so that the one-call form
 * can release its plan before raising (mexErrMsgIdAndTxt does not return). */
static char g_msg[600];
#define QFAIL(...) do { snprintf(g_msg, sizeof g_msg, __VA_ARGS__); return g_msg; } while (0)
static const char *execute_plan(plan_slot *s, const mxArray *xa, uint64_t F_hint, mxArray **out) {
    const qdas_sizes *z = &s->sz;
    const uint64_t per = z->T * z->N * z->M, I = z->I1 * z->I2 * z->I3;
    const uint64_t oN = (z->flag & QDAS_FLAG_KEEP_RX) ? z->N : 1, oM = (z->flag & QDAS_FLAG_KEEP_TX) ? z->M : 1;
    const int half = z->dtype == QDAS_F16;
    *out = NULL;
    if (!half && per && !mxIsComplex(xa)) QFAIL("x must be complex.");
    if (per && mxGetClassID(xa) != class_of(z->dtype)) QFAIL("x does not have the plan's precision.");
    const uint64_t nel = (uint64_t)mxGetNumberOfElements(xa) / (half ? 2 : 1);
    const uint64_t F = per ? nel / per : (F_hint ? F_hint : 1);
    if (per && (F == 0 || F * per != nel)) QFAIL("x must be T x N x M x F for the plan's T, N, M.");
    argptr x = arg_data(xa);
    if (per && x.is_dev != (s->mem == QDAS_MEM_DEVICE)) { arg_release(&x); QFAIL("x must be a %s array for this plan.", s->mem == QDAS_MEM_DEVICE ? "gpuArray" : "host"); }
    mxArray *ya;
    void *yp;
    mwSize dims[5], nd;
    if (half) { dims[0] = 2; dims[1] = (mwSize)I; dims[2] = (mwSize)oN; dims[3] = (mwSize)oM; dims[4] = (mwSize)F; nd = 5; }
    else { dims[0] = (mwSize)I; dims[1] = (mwSize)oN; dims[2] = (mwSize)oM; dims[3] = (mwSize)F; nd = 4; }
#ifdef QDAS_MEX_GPU
    mxGPUArray *yg = NULL;
    if (s->mem == QDAS_MEM_DEVICE) {
        yg = mxGPUCreateGPUArray(nd, dims, class_of(z->dtype), half ? mxREAL : mxCOMPLEX, MX_GPU_DO_NOT_INITIALIZE);
        yp = mxGPUGetData(yg);
        ya = NULL;
    } else
#endif
    {
        ya = mxCreateNumericArray(nd, dims, class_of(z->dtype), half ? mxREAL : mxCOMPLEX);
        yp = mxGetData(ya);
    }
    int rc = 0;
    if (s->splan) {
        const size_t ds = z->dtype == QDAS_F64 ? 16 : (z->dtype == QDAS_F32 ? 8 : 4);
        for (uint64_t f = 0; f < F && !rc; ++f)
            rc = qdas_plan_execute_sharded(s->splan, (const char *)x.ptr + f * per * ds, (char *)yp + f * I * oN * oM * ds, NULL);
    } else
        rc = qdas_plan_execute_frames(s->plan, x.ptr, yp, F, per, I * oN * oM, NULL);
    arg_release(&x);
#ifdef QDAS_MEX_GPU
    if (yg) { ya = mxGPUCreateMxArrayOnGPU(yg); mxGPUDestroyGPUArray(yg); }
#endif
    if (rc) { if (ya) mxDestroyArray(ya); QFAIL("%s", qdas_last_error()); }
    *out = ya;
    return NULL;
}

static plan_slot *slot_of(const mxArray *h) {
    const double v = num_at(h, 0, "handle");
    const int k = (int)v - 1;
    if (v != (double)(k + 1) || k < 0 || k >= QDAS_MEX_MAX_PLANS || !g_slots[k].used) mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "invalid or destroyed plan handle.");
    return &g_slots[k];
}

/* ---- the stateless commands: device-pointer entries of the C ABI behind host arrays (staged here) or gpuArrays (passed through) */
typedef struct { const void *ptr; void *owned; argptr a; } devarg;
#define QDAS_MEX_MAX_DEVARGS 8
static devarg g_da[QDAS_MEX_MAX_DEVARGS];
static int g_nda = 0;
static void *g_out_dev = NULL;                          /* device image of a host output */
static void release_devargs(void) {
    for (int k = 0; k < g_nda; ++k) { if (g_da[k].owned) qdas_device_free(g_da[k].owned, -1); arg_release(&g_da[k].a); }
    g_nda = 0;
    if (g_out_dev) { qdas_device_free(g_out_dev, -1); g_out_dev = NULL; }
}
/* raise after releasing what the command staged (mexErrMsgIdAndTxt does not return) */
#define CFAIL(...) do { release_devargs(); mexErrMsgIdAndTxt("QUPS:das_spec:qdas", __VA_ARGS__); } while (0)

/* device pointer of an input of `bytes` bytes ([] -> NULL); *any_dev collects whether some argument was a gpuArray */
static const void *dev_in(const mxArray *a, size_t bytes, const char *what, int *any_dev) {
    if (g_nda >= QDAS_MEX_MAX_DEVARGS) CFAIL("too many array arguments.");
    devarg *d = &g_da[g_nda++];
    memset(d, 0, sizeof *d);
    d->a = arg_data(a);
    if (!d->a.ptr) return NULL;
    const size_t have = (size_t)mxGetNumberOfElements(a) * (mxGetClassID(a) == mxDOUBLE_CLASS || mxGetClassID(a) == mxINT64_CLASS || mxGetClassID(a) == mxUINT64_CLASS ? 8 :
                        mxGetClassID(a) == mxSINGLE_CLASS || mxGetClassID(a) == mxINT32_CLASS || mxGetClassID(a) == mxUINT32_CLASS ? 4 :
                        mxGetClassID(a) == mxINT16_CLASS || mxGetClassID(a) == mxUINT16_CLASS ? 2 : 1) * (mxIsComplex(a) ? 2 : 1);
    if (have < bytes) CFAIL("%s: %llu bytes expected, the array holds %llu (class / complexity / size mismatch).", what, (unsigned long long)bytes, (unsigned long long)have);
    if (d->a.is_dev) { *any_dev = 1; d->ptr = d->a.ptr; return d->ptr; }
    if (qdas_device_malloc(&d->owned, bytes, -1) || qdas_device_copy(d->owned, d->a.ptr, bytes, 0, -1)) CFAIL("%s", qdas_last_error());
    d->ptr = d->owned;
    return d->ptr;
}

/* output array of `nd` dims: a gpuArray when the inputs were (GPU build), else a host array with a device image in g_out_dev; returns the device pointer */
#ifdef QDAS_MEX_GPU
static mxGPUArray *g_out_gpu = NULL;
#endif
static void *dev_out(mwSize nd, const mwSize *dims, mxClassID cls, int cplx, int on_dev, size_t bytes, mxArray **host) {
    *host = NULL;
#ifdef QDAS_MEX_GPU
    if (on_dev) { g_out_gpu = mxGPUCreateGPUArray(nd, dims, cls, cplx ? mxCOMPLEX : mxREAL, MX_GPU_DO_NOT_INITIALIZE); return mxGPUGetData(g_out_gpu); }
#endif
    (void)on_dev;
    *host = mxCreateNumericArray(nd, dims, cls, cplx ? mxCOMPLEX : mxREAL);
    if (qdas_device_malloc(&g_out_dev, bytes, -1)) { mxDestroyArray(*host); *host = NULL; CFAIL("%s", qdas_last_error()); }
    return g_out_dev;
}
/* after the launch: the host image is fetched (synchronises) ... */
static int fetch_out(int rc, mxArray *host, size_t bytes) {
    if (!rc && host && g_out_dev) rc = qdas_device_copy(mxGetData(host), g_out_dev, bytes, 1, -1);
    return rc;
}
/* ... and everything staged is released; rc != 0 raises with the library's message */
static mxArray *conclude(int rc, mxArray *host) {
    mxArray *ret = host;
#ifdef QDAS_MEX_GPU
    if (g_out_gpu) { if (!rc) ret = mxGPUCreateMxArrayOnGPU(g_out_gpu); mxGPUDestroyGPUArray(g_out_gpu); g_out_gpu = NULL; }
#endif
    if (rc) { if (host) mxDestroyArray(host); CFAIL("%s", qdas_last_error()); }
    release_devargs();
    return ret;
}
static mxArray *finish(int rc, mxArray *host, size_t bytes) { return conclude(fetch_out(rc, host, bytes), host); }
static size_t cbytes(int dtype) { return dtype == QDAS_F64 ? 16 : (dtype == QDAS_F32 ? 8 : 4); }     /* complex sample */
static size_t rbytes(int dtype) { return dtype == QDAS_F64 ? 8 : 4; }                                   /* geometry / time (half data: single) */

/* tau = qdas_mex('delays', sizes, Pi, Pr, Pv, Nv, cinv) | qdas_mex('delays', h) -- kern/das_spec.m:376-377, src/bf.cu:209-298 */
static mxArray *cmd_delays(int nrhs, const mxArray *prhs[]) {
    mxArray *host;
    if (nrhs == 1) {                                    /* a live plan */
        plan_slot *s = slot_of(prhs[0]);
        if (!s->plan) mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "'delays' needs a single-device plan.");
        const qdas_sizes *z = &s->sz;
        const mwSize dims[3] = {(mwSize)(z->I1 * z->I2 * z->I3), (mwSize)z->N, (mwSize)z->M};
        const size_t bytes = (size_t)dims[0] * dims[1] * dims[2] * rbytes(z->dtype);
        void *tau = dev_out(3, dims, z->dtype == QDAS_F64 ? mxDOUBLE_CLASS : mxSINGLE_CLASS, 0, s->mem == QDAS_MEM_DEVICE, bytes, &host);
        return finish(qdas_plan_delays(s->plan, tau, NULL), host, bytes);
    }
    if (nrhs != 6) mexErrMsgIdAndTxt("QUPS:das_spec:nargin", "qdas_mex('delays', sizes, Pi, Pr, Pv, Nv, cinv) or qdas_mex('delays', h)");
    qdas_sizes z;
    read_sizes(prhs[0], &z, NULL);
    if (z.dtype == QDAS_F16) z.dtype = QDAS_F32;        /* (delays of half data are single: kern/das_spec.m:354-357) */
    const size_t rs = rbytes(z.dtype), I = (size_t)(z.I1 * z.I2 * z.I3);
    int dev = 0;
    const void *Pi = dev_in(prhs[1], 3 * I * rs, "Pi", &dev), *Pr = dev_in(prhs[2], 3 * z.N * rs, "Pr", &dev);
    const void *Pv = dev_in(prhs[3], 4 * z.M * rs, "Pv", &dev), *Nv = dev_in(prhs[4], 3 * z.M * rs, "Nv", &dev);
    const double cinv = num_at(prhs[5], 0, "cinv");    /* cinv(1), as the reference passes it */
    const mwSize dims[3] = {(mwSize)I, (mwSize)z.N, (mwSize)z.M};
    const size_t bytes = I * z.N * z.M * rs;
    void *tau = dev_out(3, dims, z.dtype == QDAS_F64 ? mxDOUBLE_CLASS : mxSINGLE_CLASS, 0, dev, bytes, &host);
    const int rc = z.dtype == QDAS_F64 ? qdas_delays(&z, (double *)tau, (const double *)Pi, (const double *)Pr, (const double *)Pv, (const double *)Nv, cinv, NULL)
                                       : qdas_delaysf(&z, (float *)tau, (const float *)Pi, (const float *)Pr, (const float *)Pv, (const float *)Nv, (float)cinv, NULL);
    return finish(rc, host, bytes);
}

/* y = qdas_mex('lut', lsizes, w, x, t1, t2, wstride, omega) -- kern/wsinterpd2.m:236 as bfDASLUT -> sample2sep reaches it */
static mxArray *cmd_lut(int nrhs, const mxArray *prhs[]) {
    if (nrhs != 7) mexErrMsgIdAndTxt("QUPS:das_spec:nargin", "qdas_mex('lut', lsizes, w, x, t1, t2, wstride, omega)");
    qdas_lut_desc d;
    memset(&d, 0, sizeof d);
    const mxArray *ls = prhs[0];
    d.T = (uint64_t)num_at(ls, 0, "lsizes"); d.N = (uint64_t)num_at(ls, 1, "lsizes"); d.M = (uint64_t)num_at(ls, 2, "lsizes"); d.I = (uint64_t)num_at(ls, 3, "lsizes");
    d.flag = (int32_t)num_at(ls, 4, "lsizes"); d.dtype = (int32_t)num_at(ls, 5, "lsizes");
    d.I1 = mxGetNumberOfElements(ls) > 6 ? (uint64_t)num_at(ls, 6, "lsizes") : 0;
    d.w_real = mxGetNumberOfElements(ls) > 7 ? (int32_t)num_at(ls, 7, "lsizes") : 0;
    d.omega = num_at(prhs[6], 0, "omega");
    if (d.dtype < QDAS_F64 || d.dtype > QDAS_F16) mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "lsizes(6): dtype must be 0 (double), 1 (single) or 2 (half).");
    const size_t cs = cbytes(d.dtype), ts = rbytes(d.dtype), ws = d.w_real ? cs / 2 : cs;
    const uint64_t oN = (d.flag & QDAS_FLAG_KEEP_RX) ? d.N : 1, oM = (d.flag & QDAS_FLAG_KEEP_TX) ? d.M : 1;
    int dev = 0;
    if (!mxIsEmpty(prhs[1])) {
        if (mxGetClassID(prhs[5]) != mxUINT64_CLASS || mxGetNumberOfElements(prhs[5]) < 3) mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "wstride must be uint64 [si sn sm].");
        memcpy(d.wstride, mxGetData(prhs[5]), 3 * sizeof(uint64_t));
        const uint64_t nel = 1 + (d.I ? d.I - 1 : 0) * d.wstride[0] + (d.N ? d.N - 1 : 0) * d.wstride[1] + (d.M ? d.M - 1 : 0) * d.wstride[2];
        d.w = dev_in(prhs[1], (size_t)nel * ws, "w", &dev);
    }
    const void *x = dev_in(prhs[2], (size_t)(d.T * d.N * d.M) * cs, "x", &dev);
    d.tau_rx = dev_in(prhs[3], (size_t)(d.I * d.N) * ts, "t1", &dev);
    d.tau_tx = dev_in(prhs[4], (size_t)(d.I * d.M) * ts, "t2", &dev);
    const int half = d.dtype == QDAS_F16;
    mwSize dims[4], nd = 0;
    if (half) dims[nd++] = 2;
    dims[nd++] = (mwSize)d.I; dims[nd++] = (mwSize)oN; dims[nd++] = (mwSize)oM;
    const size_t bytes = (size_t)(d.I * oN * oM) * cs;
    mxArray *host;
    void *y = dev_out(nd, dims, class_of(d.dtype), !half, dev, bytes, &host);
    return finish(qdas_das_lut(&d, x, y, NULL), host, bytes);
}

/* y = qdas_mex('wsinterpd', wsz, size, strides, sumdims, w, x, t, tvars) -- kern/wsinterpd.m:221-236, src/interpd.cu:295-342 */
static mxArray *cmd_wsinterpd(int nrhs, const mxArray *prhs[]) {
    if (nrhs != 8) mexErrMsgIdAndTxt("QUPS:das_spec:nargin", "qdas_mex('wsinterpd', wsz, size, strides, sumdims, w, x, t, tvars)");
    qdas_wsinterpd_desc d;
    memset(&d, 0, sizeof d);
    const mxArray *z = prhs[0];
    d.T = (uint64_t)num_at(z, 0, "wsz"); d.x_tstride = (uint64_t)num_at(z, 1, "wsz"); d.ndim = (int32_t)num_at(z, 2, "wsz"); d.flag = (int32_t)num_at(z, 3, "wsz");
    d.dtype = (int32_t)num_at(z, 4, "wsz"); d.w_real = (int32_t)num_at(z, 5, "wsz"); d.lane_dim = -1;
    if (d.ndim < 1 || d.ndim > 8) mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "wsz(3): 1 to 8 dimensions.");
    if (d.dtype < QDAS_F64 || d.dtype > QDAS_F16) mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "wsz(5): dtype must be 0 (double), 1 (single) or 2 (half).");
    if (mxGetNumberOfElements(prhs[1]) < (size_t)d.ndim || mxGetNumberOfElements(prhs[2]) < 3 * (size_t)d.ndim || mxGetNumberOfElements(prhs[3]) < (size_t)d.ndim)
        mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "size, strides (3 x ndim: t, x, w) and sumdims must cover ndim dimensions.");
    uint64_t nt = 1, nx = 1, nw = 1, ny = 1;               /* extents of t, x (trace bases), w, y in elements */
    for (int k = 0; k < d.ndim; ++k) {
        d.size[k] = (uint64_t)num_at(prhs[1], (mwSize)k, "size");
        d.tstride[k] = (int64_t)num_at(prhs[2], (mwSize)(3 * k), "strides"); d.xstride[k] = (int64_t)num_at(prhs[2], (mwSize)(3 * k + 1), "strides");
        d.wstride[k] = (int64_t)num_at(prhs[2], (mwSize)(3 * k + 2), "strides");
        d.sum[k] = num_at(prhs[3], (mwSize)k, "sumdims") != 0;
        if (d.tstride[k] < 0 || d.xstride[k] < 0 || d.wstride[k] < 0) mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "strides must be non-negative (column-major MATLAB arrays).");
        if (d.size[k]) { nt += (d.size[k] - 1) * (uint64_t)d.tstride[k]; nx += (d.size[k] - 1) * (uint64_t)d.xstride[k]; nw += (d.size[k] - 1) * (uint64_t)d.wstride[k]; }
        if (!d.sum[k]) ny *= d.size[k];
    }
    d.omega = num_at(prhs[7], 0, "tvars"); d.extrap = num_at(prhs[7], 1, "tvars");
    const size_t cs = cbytes(d.dtype), ts = rbytes(d.dtype), ws = d.w_real ? cs / 2 : cs;
    int dev = 0;
    d.w = mxIsEmpty(prhs[4]) ? NULL : dev_in(prhs[4], (size_t)nw * ws, "w", &dev);
    d.x = dev_in(prhs[5], (size_t)(nx - 1 + (d.T ? (d.T - 1) * d.x_tstride + 1 : 0)) * cs, "x", &dev);
    d.t = dev_in(prhs[6], (size_t)nt * ts, "t", &dev);
    const int half = d.dtype == QDAS_F16;
    mwSize dims[9], nd = 0;
    if (half) dims[nd++] = 2;
    for (int k = 0; k < d.ndim; ++k) dims[nd++] = (mwSize)(d.sum[k] ? 1 : d.size[k]);
    const size_t bytes = (size_t)ny * cs;
    mxArray *host;
    void *y = dev_out(nd, dims, class_of(d.dtype), !half, dev, bytes, &host);
    return finish(qdas_wsinterpd(&d, y, NULL), host, bytes);
}

/* y = qdas_mex('shiftsum', ssz, x, shift, w) -- UltrasoundSystem.focusTx, src/UltrasoundSystem.m:3374-3503 (:3498) */
static mxArray *cmd_shiftsum(int nrhs, const mxArray *prhs[]) {
    if (nrhs != 4) mexErrMsgIdAndTxt("QUPS:das_spec:nargin", "qdas_mex('shiftsum', ssz, x, shift, w)");
    qdas_shift_desc d;
    memset(&d, 0, sizeof d);
    const mxArray *z = prhs[0];
    d.T = (uint64_t)num_at(z, 0, "ssz"); d.To = (uint64_t)num_at(z, 1, "ssz"); d.N = (uint64_t)num_at(z, 2, "ssz"); d.M = (uint64_t)num_at(z, 3, "ssz");
    d.Mo = (uint64_t)num_at(z, 4, "ssz"); d.F = (uint64_t)num_at(z, 5, "ssz"); d.flag = (int32_t)num_at(z, 6, "ssz"); d.dtype = (int32_t)num_at(z, 7, "ssz");
    d.cplx = (int32_t)num_at(z, 8, "ssz"); d.w_real = mxGetNumberOfElements(z) > 9 ? (int32_t)num_at(z, 9, "ssz") : 1;
    d.tpad = mxGetNumberOfElements(z) > 10 ? (int32_t)num_at(z, 10, "ssz") : 0;      /* zeros behind the record that are not stored (include/qdas.h) */
    d.device = -1;
    if (d.dtype != QDAS_F64 && d.dtype != QDAS_F32) mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "shiftsum: datatype must be double or single");
    const size_t rs = rbytes(d.dtype), es = rs * (d.cplx ? 2 : 1);
    int dev = 0;
    const void *x = dev_in(prhs[1], (size_t)(d.T * d.N * d.M * d.F) * es, "x", &dev);
    d.shift = dev_in(prhs[2], (size_t)(d.M * d.Mo) * rs, "shift", &dev);
    d.w = mxIsEmpty(prhs[3]) ? NULL : dev_in(prhs[3], (size_t)(d.M * d.Mo) * (d.w_real ? rs : 2 * rs), "w", &dev);
    const mwSize dims[4] = {(mwSize)d.To, (mwSize)d.N, (mwSize)d.Mo, (mwSize)d.F};
    const size_t bytes = (size_t)(d.To * d.N * d.Mo * d.F) * es;
    mxArray *host;
    void *y = dev_out(4, dims, d.dtype == QDAS_F64 ? mxDOUBLE_CLASS : mxSINGLE_CLASS, d.cplx, dev, bytes, &host);
    return finish(qdas_shift_sum(&d, x, y, NULL), host, bytes);
}

/* y = qdas_mex('greens', gsizes, ps, as, pn, pv, x, tvars) -- src/UltrasoundSystem.m:681-718, src/greens.cu:88-121 */
static mxArray *cmd_greens(int nrhs, const mxArray *prhs[]) {
    if (nrhs != 7) mexErrMsgIdAndTxt("QUPS:das_spec:nargin", "qdas_mex('greens', gsizes, ps, as, pn, pv, x, tvars)");
    qdas_greens_desc d;
    memset(&d, 0, sizeof d);
    const mxArray *g = prhs[0], *tv = prhs[6];
    d.S = (uint64_t)num_at(g, 0, "gsizes"); d.T = (uint64_t)num_at(g, 1, "gsizes"); d.N = (uint64_t)num_at(g, 2, "gsizes"); d.M = (uint64_t)num_at(g, 3, "gsizes");
    d.I = (uint64_t)num_at(g, 4, "gsizes"); d.En = (int32_t)num_at(g, 5, "gsizes"); d.Em = (int32_t)num_at(g, 6, "gsizes");
    d.interp = (int32_t)num_at(g, 7, "gsizes"); d.dtype = (int32_t)num_at(g, 8, "gsizes");
    d.s0 = num_at(tv, 0, "tvars"); d.t0 = num_at(tv, 1, "tvars"); d.fs = num_at(tv, 2, "tvars"); d.fsr = num_at(tv, 3, "tvars"); d.cinv = num_at(tv, 4, "tvars"); d.R0 = num_at(tv, 5, "tvars");
    d.device = -1;
    if (d.dtype != QDAS_F64 && d.dtype != QDAS_F32) mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "greens: datatype must be double or single");
    if (d.En < 1 || d.Em < 1) mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "greens: element subdivisions must be >= 1");
    const size_t rs = rbytes(d.dtype), cs = cbytes(d.dtype);
    int dev = 0;
    d.Ps = dev_in(prhs[1], 3 * (size_t)d.I * rs, "ps", &dev);
    d.a = dev_in(prhs[2], (size_t)d.I * cs, "as", &dev);
    d.Pr = dev_in(prhs[3], 3 * (size_t)d.N * (size_t)d.En * rs, "pn", &dev);
    d.Pv = dev_in(prhs[4], 3 * (size_t)d.M * (size_t)d.Em * rs, "pv", &dev);
    d.x = dev_in(prhs[5], (size_t)d.T * cs, "x", &dev);
    const mwSize dims[3] = {(mwSize)d.S, (mwSize)d.N, (mwSize)d.M};
    const size_t bytes = (size_t)(d.S * d.N * d.M) * cs;
    mxArray *host;
    void *y = dev_out(3, dims, d.dtype == QDAS_F64 ? mxDOUBLE_CLASS : mxSINGLE_CLASS, 1, dev, bytes, &host);
    return finish(qdas_greens(&d, y, NULL), host, bytes);
}

/* z = qdas_mex('convd', csizes, x, y) -- kern/convd.m:150-199, src/convd.cu:95-146 */
static mxArray *cmd_convd(int nrhs, const mxArray *prhs[]) {
    if (nrhs != 3) mexErrMsgIdAndTxt("QUPS:das_spec:nargin", "qdas_mex('convd', csizes, x, y)");
    qdas_convd_desc d;
    memset(&d, 0, sizeof d);
    const mxArray *c = prhs[0];
    d.C = (uint64_t)num_at(c, 0, "csizes"); d.M = (uint64_t)num_at(c, 1, "csizes"); d.N = (uint64_t)num_at(c, 2, "csizes"); d.S = (uint64_t)num_at(c, 3, "csizes");
    d.dtype = (int32_t)num_at(c, 4, "csizes"); d.cplx = (int32_t)num_at(c, 5, "csizes"); d.shape = (int32_t)num_at(c, 6, "csizes");
    d.bcast = mxGetNumberOfElements(c) > 7 ? (int32_t)num_at(c, 7, "csizes") : 0;
    d.y_real = mxGetNumberOfElements(c) > 8 ? (int32_t)num_at(c, 8, "csizes") : 0;
    d.device = -1;
    if (d.dtype < QDAS_F64 || d.dtype > QDAS_F16) mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "csizes(5): dtype must be 0 (double), 1 (single) or 2 (half).");
    const size_t es = (d.dtype == QDAS_F64 ? 8 : d.dtype == QDAS_F32 ? 4 : 2), xs = es * (d.cplx ? 2 : 1), ys = es * ((d.cplx && !d.y_real) ? 2 : 1);
    const uint64_t L = qdas_convd_len(d.M, d.N, d.shape);
    const uint64_t xC = (d.bcast & QDAS_CONV_X_ONE_COLUMN) ? 1 : d.C, xS = (d.bcast & QDAS_CONV_X_ONE_SLICE) ? 1 : d.S;
    const uint64_t yC = (d.bcast & QDAS_CONV_Y_ONE_COLUMN) ? 1 : d.C, yS = (d.bcast & QDAS_CONV_Y_ONE_SLICE) ? 1 : d.S;
    int dev = 0;
    const void *x = dev_in(prhs[1], (size_t)(xC * d.M * xS) * xs, "x", &dev);
    const void *y = dev_in(prhs[2], (size_t)(yC * d.N * yS) * ys, "y", &dev);
    const int half = d.dtype == QDAS_F16;
    mwSize dims[4], nd = 0;
    if (half && d.cplx) dims[nd++] = 2;                 /* (half complex travels as uint16 pairs, like the DAS data) */
    dims[nd++] = (mwSize)d.C; dims[nd++] = (mwSize)L; dims[nd++] = (mwSize)d.S;
    const size_t bytes = (size_t)(d.C * L * d.S) * xs;
    mxArray *host;
    void *z = dev_out(nd, dims, class_of(d.dtype), d.cplx && !half, dev, bytes, &host);
    return finish(qdas_convd(&d, x, y, z, NULL), host, bytes);
}

/* y = qdas_mex('hilbert', psizes, x, tvars) -- src/ChannelData.m:935-966 (+ downmix :757-766): plan, one execute, destroy */
static mxArray *cmd_hilbert(int nrhs, const mxArray *prhs[]) {
    if (nrhs != 3) mexErrMsgIdAndTxt("QUPS:das_spec:nargin", "qdas_mex('hilbert', psizes, x, tvars)");
    qdas_pre_desc d;
    memset(&d, 0, sizeof d);
    const mxArray *p = prhs[0], *tv = prhs[2];
    d.T = (uint64_t)num_at(p, 0, "psizes"); d.K = (uint64_t)num_at(p, 1, "psizes"); d.Nfft = (uint64_t)num_at(p, 2, "psizes");
    d.in_type = mxGetNumberOfElements(p) > 3 ? (int32_t)num_at(p, 3, "psizes") : QDAS_PRE_F32;
    d.device = -1;
    d.fs = num_at(tv, 0, "tvars"); d.t0 = num_at(tv, 1, "tvars"); d.fdown = num_at(tv, 2, "tvars");
    if (d.in_type != QDAS_PRE_F32 && d.in_type != QDAS_PRE_I16) mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "psizes(4): in_type must be 0 (single) or 1 (int16).");
    if (mxIsComplex(prhs[1]) || mxGetClassID(prhs[1]) != (d.in_type == QDAS_PRE_I16 ? mxINT16_CLASS : mxSINGLE_CLASS))
        mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "x must be real %s for this in_type.", d.in_type == QDAS_PRE_I16 ? "int16" : "single");
    const uint64_t Nf = d.Nfft ? d.Nfft : d.T;
    int dev = 0;
    const void *x = dev_in(prhs[1], (size_t)(d.T * d.K) * (d.in_type == QDAS_PRE_I16 ? 2 : 4), "x", &dev);
    const mwSize dims[2] = {(mwSize)Nf, (mwSize)d.K};
    const size_t bytes = (size_t)(Nf * d.K) * 8;
    mxArray *host;
    void *y = dev_out(2, dims, mxSINGLE_CLASS, 1, dev, bytes, &host);
    qdas_pre_plan *pl = NULL;
    int rc = qdas_pre_plan_create(&pl, &d);
    if (!rc) rc = qdas_pre_execute(pl, x, y, NULL);
    rc = fetch_out(rc, host, bytes);                    /* (host output: synchronises before the plan goes; gpuArray output: the plan's hipFree does) */
    if (pl) qdas_pre_plan_destroy(pl);
    return conclude(rc, host);
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
#ifdef QDAS_MEX_GPU
    mxInitGPU();
#endif
    if (nlhs > 1) mexErrMsgIdAndTxt("QUPS:das_spec:nargout", "qdas_mex returns at most one output.");
    if (nrhs >= 1 && mxIsChar(prhs[0])) {
        char cmd[32];
        if (mxGetString(prhs[0], cmd, sizeof cmd)) mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "unreadable command.");
        if (!strcmp(cmd, "create")) {
            const int slot = create_plan(nrhs - 1, prhs + 1);
            plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
            *(uint64_t *)mxGetData(plhs[0]) = (uint64_t)(slot + 1);
        } else if (!strcmp(cmd, "execute")) {
            if (nrhs != 3) mexErrMsgIdAndTxt("QUPS:das_spec:nargin", "qdas_mex('execute', h, x)");
            const char *err = execute_plan(slot_of(prhs[1]), prhs[2], 0, &plhs[0]);
            if (err) mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "%s", err);
        } else if (!strcmp(cmd, "prepare")) {
            if (nrhs != 3) mexErrMsgIdAndTxt("QUPS:das_spec:nargin", "qdas_mex('prepare', h, F)");
            plan_slot *s = slot_of(prhs[1]);
            if (s->plan && qdas_plan_prepare_frames(s->plan, (uint64_t)num_at(prhs[2], 0, "F"))) mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "%s", qdas_last_error());
        } else if (!strcmp(cmd, "info")) {
            if (nrhs != 2) mexErrMsgIdAndTxt("QUPS:das_spec:nargin", "qdas_mex('info', h)");
            plan_slot *s = slot_of(prhs[1]);
            char buf[512];
            if (s->plan) {
                char kn[256];
                qdas_plan_kernel_name(s->plan, kn, sizeof kn);
                snprintf(buf, sizeof buf, "%s; reciprocal=%d", kn, qdas_plan_reciprocal(s->plan));
            } else {
                int n = 0;
                qdas_plan_sharded_info(s->splan, -1, &n, NULL, NULL, NULL);
                snprintf(buf, sizeof buf, "sharded over %d device slab(s)", n);
            }
            plhs[0] = mxCreateString(buf);
        } else if (!strcmp(cmd, "delays")) { plhs[0] = cmd_delays(nrhs - 1, prhs + 1);
        } else if (!strcmp(cmd, "lut")) { plhs[0] = cmd_lut(nrhs - 1, prhs + 1);
        } else if (!strcmp(cmd, "wsinterpd")) { plhs[0] = cmd_wsinterpd(nrhs - 1, prhs + 1);
        } else if (!strcmp(cmd, "shiftsum")) { plhs[0] = cmd_shiftsum(nrhs - 1, prhs + 1);
        } else if (!strcmp(cmd, "greens")) { plhs[0] = cmd_greens(nrhs - 1, prhs + 1);
        } else if (!strcmp(cmd, "convd")) { plhs[0] = cmd_convd(nrhs - 1, prhs + 1);
        } else if (!strcmp(cmd, "hilbert")) { plhs[0] = cmd_hilbert(nrhs - 1, prhs + 1);
        } else if (!strcmp(cmd, "destroy")) {
            if (nrhs >= 2) destroy_slot(slot_of(prhs[1])); else destroy_all();
        } else mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "unknown command '%s'.", cmd);
        return;
    }
    /* one call: create + execute + destroy */
    if (nrhs != 10) mexErrMsgIdAndTxt("QUPS:das_spec:nargin", "qdas_mex expects 10 inputs (or a command string first).");
    qdas_sizes z;
    uint64_t F = 1;
    read_sizes(prhs[0], &z, &F);
    const mxArray *cargs[9] = {prhs[0], prhs[1], prhs[2], prhs[3], prhs[4], prhs[5], prhs[6], prhs[7], prhs[9]};
    const int slot = create_plan(9, cargs);
    const char *err = execute_plan(&g_slots[slot], prhs[8], F, &plhs[0]);
    destroy_slot(&g_slots[slot]);
    if (err) mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "%s", err);
}
