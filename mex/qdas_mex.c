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
 *   s = qdas_mex('info', h)                                                             char: kernel name, shards
 *       qdas_mex('destroy' [, h])
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
