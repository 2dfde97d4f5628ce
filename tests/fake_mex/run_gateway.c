/* run_gateway.c -- drives mex/qdas_mex.c's mexFunction through its sub-commands over the fake MEX runtime and the REAL libqdas.so,
 * and checks the results against the C ABI called directly (TEST INFRASTRUCTURE; needs a GPU): create / execute / info / destroy, multi-device plans,
 * and the stateless commands of the other launch sites (delays, lut, greens, convd, hilbert).  Prints "fake-MEX gateway OK". */
#include <math.h>
#include <setjmp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mex.h"
#include "qdas.h"

extern jmp_buf fake_mex_jmp;
extern char fake_mex_last_id[128], fake_mex_last_msg[1024];
void fake_mex_run_atexit(void);
mxArray *fake_mex_struct(int n, const char **names, mxArray **vals);

enum { T = 300, N = 8, M = 8, I1 = 70, I2 = 6, I = I1 * I2 };
#define CHECK(c) do { if (!(c)) { printf("FAILED line %d: %s (last error: %s / %s)\n", __LINE__, #c, fake_mex_last_id, fake_mex_last_msg); return 1; } } while (0)

static mxArray *arr(mwSize a, mwSize b, mxClassID cls, int cplx) { const mwSize d[2] = {a, b}; return mxCreateNumericArray(2, d, cls, cplx ? mxCOMPLEX : mxREAL); }
static int call(int nlhs, mxArray **out, int nrhs, const mxArray **in) {     /* 0: returned, 1: raised */
    if (setjmp(fake_mex_jmp)) return 1;
    mexFunction(nlhs, out, nrhs, in);
    return 0;
}

static void *dup_dev(const void *h, size_t bytes) {      /* device copy of a host array (qdas_device_*: what the gateway itself stages with) */
    void *p = NULL;
    if (qdas_device_malloc(&p, bytes, -1) || qdas_device_copy(p, h, bytes, 0, -1)) return NULL;
    return p;
}

int main(void) {
    /* a small full-synthetic-aperture problem: 8-element array, 70 x 6 pixels, random data */
    mxArray *sizes = arr(1, 12, mxDOUBLE_CLASS, 0), *Pi = arr(3, I, mxSINGLE_CLASS, 0), *Pr = arr(3, N, mxSINGLE_CLASS, 0), *Pv = arr(4, M, mxSINGLE_CLASS, 0),
            *Nv = arr(3, M, mxSINGLE_CLASS, 0), *apod = arr(0, 0, mxSINGLE_CLASS, 1), *cinv = arr(1, 1, mxSINGLE_CLASS, 0), *acs = arr(6, 1, mxUINT64_CLASS, 0),
            *tv = arr(1, 2, mxSINGLE_CLASS, 0);                                  /* (single tvars: any real class is accepted) */
    const double sz[12] = {T, N, M, I1, I2, 1, 0, 2 /* cubic */, 1, 1, 1 /* single */, 2 /* F */};
    memcpy(mxGetData(sizes), sz, sizeof sz);
    float *pi = (float *)mxGetData(Pi), *pr = (float *)mxGetData(Pr), *pv = (float *)mxGetData(Pv), *nv = (float *)mxGetData(Nv);
    for (int n = 0; n < N; ++n) { pr[3 * n] = (n - 3.5f) * 3e-4f; pr[3 * n + 1] = 0; pr[3 * n + 2] = 0; }
    for (int m = 0; m < M; ++m) { memcpy(pv + 4 * m, pr + 3 * m, 12); pv[4 * m + 3] = 0; nv[3 * m] = 0; nv[3 * m + 1] = 0; nv[3 * m + 2] = 1; }
    for (int c = 0; c < I2; ++c) for (int r = 0; r < I1; ++r) { float *p = pi + 3 * (r + I1 * c); p[0] = (c - 2.5f) * 3e-4f; p[1] = 0; p[2] = 4e-3f + r * 7.7e-5f; }
    ((float *)mxGetData(cinv))[0] = 1.0f / 1540.0f;
    ((float *)mxGetData(tv))[0] = 20e6f; ((float *)mxGetData(tv))[1] = 0.f;
    const mwSize xd[4] = {T, N, M, 2};
    mxArray *x = mxCreateNumericArray(4, xd, mxSINGLE_CLASS, mxCOMPLEX);
    float *xp = (float *)mxGetData(x);
    srand(7);
    for (size_t k = 0; k < (size_t)2 * T * N * M * 2; ++k) xp[k] = (float)rand() / RAND_MAX - 0.5f;

    mxArray *out[1] = {NULL};
    /* 1. one call */
    const mxArray *a1[10] = {sizes, Pi, Pr, Pv, Nv, apod, cinv, acs, x, tv};
    CHECK(call(1, out, 10, a1) == 0);
    mxArray *y_one = out[0];
    CHECK(mxGetNumberOfElements(y_one) == (size_t)I * 2 && mxIsComplex(y_one) && mxGetClassID(y_one) == mxSINGLE_CLASS && !mexIsLocked());
    /* 2. persistent plan: create, info, execute twice, destroy */
    mxArray *cmd_create = mxCreateString("create"), *cmd_exec = mxCreateString("execute"), *cmd_info = mxCreateString("info"), *cmd_destroy = mxCreateString("destroy");
    const mxArray *a2[10] = {cmd_create, sizes, Pi, Pr, Pv, Nv, apod, cinv, acs, tv};
    CHECK(call(1, out, 10, a2) == 0);
    mxArray *h = out[0];
    CHECK(mxGetClassID(h) == mxUINT64_CLASS && *(uint64_t *)mxGetData(h) >= 1 && mexIsLocked());
    const mxArray *a3[2] = {cmd_info, h};
    CHECK(call(1, out, 2, a3) == 0);
    char info[512];
    CHECK(mxGetString(out[0], info, sizeof info) == 0 && strstr(info, "das_") != NULL);
    printf("plan: %s\n", info);
    mxArray *cmd_prepare = mxCreateString("prepare"), *nfr = arr(1, 1, mxDOUBLE_CLASS, 0);
    ((double *)mxGetData(nfr))[0] = 2;
    const mxArray *a3b[3] = {cmd_prepare, h, nfr};
    CHECK(call(0, out, 3, a3b) == 0);                                             /* the one-time work of the first stream, ahead of it */
    const mxArray *a4[3] = {cmd_exec, h, x};
    CHECK(call(1, out, 3, a4) == 0);
    mxArray *y_a = out[0];
    CHECK(call(1, out, 3, a4) == 0);
    mxArray *y_b = out[0];
    CHECK(memcmp(mxGetData(y_a), mxGetData(y_one), sizeof(float) * 2 * I * 2) == 0 && memcmp(mxGetData(y_b), mxGetData(y_one), sizeof(float) * 2 * I * 2) == 0);
    /* 3. the same frames through the C ABI directly */
    qdas_desc d;
    memset(&d, 0, sizeof d);
    d.sz.T = T; d.sz.N = N; d.sz.M = M; d.sz.I1 = I1; d.sz.I2 = I2; d.sz.I3 = 1; d.sz.flag = 2; d.sz.VS = 1; d.sz.DV = 1; d.sz.dtype = QDAS_F32;
    d.fs = 20e6f; d.Pi = pi; d.Pr = pr; d.Pv = pv; d.Nv = nv; d.cinv = mxGetData(cinv); d.acstride = (const uint64_t *)mxGetData(acs);
    d.mem = QDAS_MEM_HOST; d.device = -1;
    qdas_plan *pl = NULL;
    CHECK(qdas_plan_create(&pl, &d) == 0);
    float *yd = (float *)malloc(sizeof(float) * 2 * I * 2);
    CHECK(qdas_plan_execute_frames(pl, xp, yd, 2, (uint64_t)T * N * M, I, NULL) == 0);
    qdas_plan_destroy(pl);
    CHECK(memcmp(yd, mxGetData(y_one), sizeof(float) * 2 * I * 2) == 0);
    double e = 0;
    for (int k = 0; k < 2 * I * 2; ++k) e += fabs(yd[k]);
    CHECK(e > 0);
    /* 4. a multi-device plan (two slabs on device 0) and the JIT flag through opts */
    mxArray *devs = arr(1, 2, mxDOUBLE_CLASS, 0), *jit = arr(1, 1, mxLOGICAL_CLASS, 0);
    ((unsigned char *)mxGetData(jit))[0] = 0;
    const char *fn[2] = {"devices", "jit"};
    mxArray *fv[2] = {devs, jit};
    mxArray *opts = fake_mex_struct(2, fn, fv);
    const mxArray *a5[11] = {cmd_create, sizes, Pi, Pr, Pv, Nv, apod, cinv, acs, tv, opts};
    CHECK(call(1, out, 11, a5) == 0);
    mxArray *h2 = out[0];
    const mxArray *a6[3] = {cmd_exec, h2, x};
    CHECK(call(1, out, 3, a6) == 0);
    CHECK(memcmp(mxGetData(out[0]), mxGetData(y_one), sizeof(float) * 2 * I * 2) == 0);
    /* 5. errors come back as QUPS:das_spec:* identifiers, not crashes */
    CHECK(call(1, out, 3, a1) == 1 && strstr(fake_mex_last_id, "QUPS:das_spec:nargin"));
    mxArray *xbad = arr(5, 1, mxSINGLE_CLASS, 1);
    const mxArray *a7[3] = {cmd_exec, h, xbad};
    CHECK(call(1, out, 3, a7) == 1 && strstr(fake_mex_last_msg, "T x N x M"));
    const mxArray *a8[2] = {cmd_destroy, h};
    CHECK(call(0, out, 2, a8) == 0);
    CHECK(call(1, out, 3, a4) == 1 && strstr(fake_mex_last_msg, "destroyed"));      /* stale handle */
    CHECK(mexIsLocked());                                                             /* h2 is still alive */
    /* 6. the other launch sites of the path -- 'delays', 'lut', 'greens', 'convd', 'hilbert' -- against the C ABI called directly on device arrays, bit for bit */
    {
        /* 6a. delays: one-shot and from the live plan h2's sibling (a fresh single-device plan) */
        mxArray *cmd_delays = mxCreateString("delays"), *sz11 = arr(1, 11, mxDOUBLE_CLASS, 0);
        memcpy(mxGetData(sz11), sz, 11 * sizeof(double));
        const mxArray *b1[7] = {cmd_delays, sz11, Pi, Pr, Pv, Nv, cinv};
        CHECK(call(1, out, 7, b1) == 0);
        mxArray *tau_m = out[0];
        CHECK(mxGetClassID(tau_m) == mxSINGLE_CLASS && !mxIsComplex(tau_m) && mxGetNumberOfElements(tau_m) == (size_t)I * N * M);
        void *dPi = dup_dev(pi, sizeof(float) * 3 * I), *dPr = dup_dev(pr, sizeof(float) * 3 * N), *dPv = dup_dev(pv, sizeof(float) * 4 * M), *dNv = dup_dev(nv, sizeof(float) * 3 * M), *dtau = NULL;
        CHECK(dPi && dPr && dPv && dNv && qdas_device_malloc(&dtau, sizeof(float) * I * N * M, -1) == 0);
        CHECK(qdas_delaysf(&d.sz, (float *)dtau, (const float *)dPi, (const float *)dPr, (const float *)dPv, (const float *)dNv, 1.0f / 1540.0f, NULL) == 0);
        float *tau_d = (float *)malloc(sizeof(float) * I * N * M);
        CHECK(qdas_device_copy(tau_d, dtau, sizeof(float) * I * N * M, 1, -1) == 0);
        CHECK(memcmp(tau_d, mxGetData(tau_m), sizeof(float) * I * N * M) == 0 && tau_d[5] > 0);
        const mxArray *b2[10] = {cmd_create, sizes, Pi, Pr, Pv, Nv, apod, cinv, acs, tv};
        CHECK(call(1, out, 10, b2) == 0);
        mxArray *h3 = out[0];
        const mxArray *b3[2] = {cmd_delays, h3};
        CHECK(call(1, out, 2, b3) == 0);
        CHECK(memcmp(tau_d, mxGetData(out[0]), sizeof(float) * I * N * M) == 0);
        const mxArray *b4[2] = {cmd_destroy, h3};
        CHECK(call(0, out, 2, b4) == 0);
        /* 6b. lut: bfDASLUT's tables = the delays above split into a receive and a transmit leg (any split works for the comparison), in samples */
        mxArray *cmd_lut = mxCreateString("lut"), *ls = arr(1, 8, mxDOUBLE_CLASS, 0), *t1 = arr(I, N, mxSINGLE_CLASS, 0), *t2 = arr(I, M, mxSINGLE_CLASS, 0),
                *w = arr(I, N, mxSINGLE_CLASS, 0), *wst = arr(1, 3, mxUINT64_CLASS, 0), *om = arr(1, 1, mxDOUBLE_CLASS, 0), *x1 = mxCreateNumericArray(3, xd, mxSINGLE_CLASS, mxCOMPLEX);
        memcpy(mxGetData(x1), xp, sizeof(float) * 2 * T * N * M);
        float *t1p = (float *)mxGetData(t1), *t2p = (float *)mxGetData(t2), *wp = (float *)mxGetData(w);
        for (int n = 0; n < N; ++n) for (int i = 0; i < I; ++i) { t1p[i + I * n] = 0.5f * tau_d[i + I * (n + N * n)] * 20e6f; wp[i + I * n] = (float)((i + 3 * n) % 7) / 6.0f; }
        for (int m = 0; m < M; ++m) for (int i = 0; i < I; ++i) t2p[i + I * m] = 0.5f * tau_d[i + I * (m + N * m)] * 20e6f;
        ((uint64_t *)mxGetData(wst))[0] = 1; ((uint64_t *)mxGetData(wst))[1] = I; ((uint64_t *)mxGetData(wst))[2] = 0;
        void *dx = dup_dev(xp, sizeof(float) * 2 * T * N * M), *dt1 = dup_dev(t1p, sizeof(float) * I * N), *dt2 = dup_dev(t2p, sizeof(float) * I * M), *dw = dup_dev(wp, sizeof(float) * I * N), *dy = NULL;
        CHECK(dx && dt1 && dt2 && dw && qdas_device_malloc(&dy, sizeof(float) * 2 * I * N, -1) == 0);
        float *yl = (float *)malloc(sizeof(float) * 2 * I * N);
        for (int variant = 0; variant < 3; ++variant) {          /* full sum without weights | full sum with I x N weights | receive dimension kept */
            const double lsv[8] = {T, N, M, I, 2 /* cubic */ + (variant == 2 ? QDAS_FLAG_KEEP_RX : 0), 1, I1, 1 /* real weights */};
            memcpy(mxGetData(ls), lsv, sizeof lsv);
            const mxArray *c1[8] = {cmd_lut, ls, variant ? w : apod /* [] */, x1, t1, t2, wst, om};
            CHECK(call(1, out, 8, c1) == 0);
            const size_t on = variant == 2 ? N : 1;
            CHECK(mxIsComplex(out[0]) && mxGetNumberOfElements(out[0]) == (size_t)I * on);
            qdas_lut_desc ld;
            memset(&ld, 0, sizeof ld);
            ld.T = T; ld.N = N; ld.M = M; ld.I = I; ld.flag = (int32_t)lsv[4]; ld.dtype = QDAS_F32; ld.tau_rx = dt1; ld.tau_tx = dt2; ld.I1 = I1;
            if (variant) { ld.w = dw; ld.wstride[0] = 1; ld.wstride[1] = I; ld.w_real = 1; }
            CHECK(qdas_das_lut(&ld, dx, dy, NULL) == 0 && qdas_device_copy(yl, dy, sizeof(float) * 2 * I * on, 1, -1) == 0);
            CHECK(memcmp(yl, mxGetData(out[0]), sizeof(float) * 2 * I * on) == 0);
            double el = 0;
            for (size_t k = 0; k < 2 * (size_t)I * on; ++k) el += fabs(yl[k]);
            CHECK(el > 0);
        }
        /* 6c. greens: 40 scatterers in front of the array, a 64-sample pulse */
        enum { GS = 300, GT = 64, GI = 40 };
        mxArray *cmd_greens = mxCreateString("greens"), *gs = arr(1, 9, mxDOUBLE_CLASS, 0), *ps = arr(3, GI, mxSINGLE_CLASS, 0), *as = arr(GI, 1, mxSINGLE_CLASS, 1),
                *pn = arr(3, N, mxSINGLE_CLASS, 0), *pvg = arr(3, M, mxSINGLE_CLASS, 0), *xg = arr(GT, 1, mxSINGLE_CLASS, 1), *gtv = arr(1, 6, mxDOUBLE_CLASS, 0);
        const double gsv[9] = {GS, GT, N, M, GI, 1, 1, 1 /* linear */, 1 /* single */}, gtvv[6] = {2e-6, -1e-6, 20e6, 1.0, 1.0 / 1540.0, 1e-3};
        memcpy(mxGetData(gs), gsv, sizeof gsv); memcpy(mxGetData(gtv), gtvv, sizeof gtvv);
        float *psp = (float *)mxGetData(ps), *asp = (float *)mxGetData(as), *xgp = (float *)mxGetData(xg);
        for (int i = 0; i < GI; ++i) { psp[3 * i] = (i % 8 - 3.5f) * 4e-4f; psp[3 * i + 1] = 0; psp[3 * i + 2] = 3e-3f + (i / 8) * 1.5e-3f; asp[2 * i] = 1.0f + 0.1f * i; asp[2 * i + 1] = 0.05f * i; }
        for (int k = 0; k < GT; ++k) { const float e = expf(-(k - 32) * (k - 32) / 128.0f); xgp[2 * k] = e * cosf(1.5f * k); xgp[2 * k + 1] = e * sinf(1.5f * k); }
        memcpy(mxGetData(pn), pr, sizeof(float) * 3 * N);
        for (int m = 0; m < M; ++m) memcpy((float *)mxGetData(pvg) + 3 * m, pv + 4 * m, 12);
        const mxArray *g1[8] = {cmd_greens, gs, ps, as, pn, pvg, xg, gtv};
        CHECK(call(1, out, 8, g1) == 0);
        CHECK(mxIsComplex(out[0]) && mxGetNumberOfElements(out[0]) == (size_t)GS * N * M);
        qdas_greens_desc gd;
        memset(&gd, 0, sizeof gd);
        gd.S = GS; gd.T = GT; gd.N = N; gd.M = M; gd.I = GI; gd.En = gd.Em = 1; gd.interp = 1; gd.dtype = QDAS_F32;
        gd.s0 = gtvv[0]; gd.t0 = gtvv[1]; gd.fs = gtvv[2]; gd.fsr = gtvv[3]; gd.cinv = gtvv[4]; gd.R0 = gtvv[5]; gd.device = -1;
        void *dgy = NULL;
        gd.Ps = dup_dev(psp, sizeof(float) * 3 * GI); gd.a = dup_dev(asp, sizeof(float) * 2 * GI); gd.Pr = dup_dev(pr, sizeof(float) * 3 * N);
        gd.Pv = dup_dev(mxGetData(pvg), sizeof(float) * 3 * M); gd.x = dup_dev(xgp, sizeof(float) * 2 * GT);
        CHECK(gd.Ps && gd.a && gd.Pr && gd.Pv && gd.x && qdas_device_malloc(&dgy, sizeof(float) * 2 * GS * N * M, -1) == 0);
        float *yg = (float *)malloc(sizeof(float) * 2 * GS * N * M);
        CHECK(qdas_greens(&gd, dgy, NULL) == 0 && qdas_device_copy(yg, dgy, sizeof(float) * 2 * GS * N * M, 1, -1) == 0);
        CHECK(memcmp(yg, mxGetData(out[0]), sizeof(float) * 2 * GS * N * M) == 0);
        double eg = 0;
        for (int k = 0; k < 2 * GS * N * M; ++k) eg += fabs(yg[k]);
        CHECK(eg > 0);
        /* 6d. convd: a 17-tap real filter along the time axis of one frame, 'same' */
        enum { CN = 17 };
        mxArray *cmd_convd = mxCreateString("convd"), *cs = arr(1, 9, mxDOUBLE_CLASS, 0), *taps = arr(CN, 1, mxSINGLE_CLASS, 0);
        const double csv[9] = {1, T, CN, N * M, 1 /* single */, 1 /* complex */, QDAS_CONV_SAME, QDAS_CONV_Y_ONE_SLICE, 1 /* real taps */};
        memcpy(mxGetData(cs), csv, sizeof csv);
        float *tp = (float *)mxGetData(taps);
        for (int k = 0; k < CN; ++k) tp[k] = (k % 3 - 1) * 0.25f + 0.1f * k;
        const mxArray *v1[4] = {cmd_convd, cs, x1, taps};
        CHECK(call(1, out, 4, v1) == 0);
        CHECK(mxIsComplex(out[0]) && mxGetNumberOfElements(out[0]) == (size_t)T * N * M);
        qdas_convd_desc cd;
        memset(&cd, 0, sizeof cd);
        cd.C = 1; cd.M = T; cd.N = CN; cd.S = N * M; cd.dtype = QDAS_F32; cd.cplx = 1; cd.shape = QDAS_CONV_SAME; cd.bcast = QDAS_CONV_Y_ONE_SLICE; cd.device = -1; cd.y_real = 1;
        void *dtaps = dup_dev(tp, sizeof(float) * CN), *dz = NULL;
        CHECK(dtaps && qdas_device_malloc(&dz, sizeof(float) * 2 * T * N * M, -1) == 0);
        float *zc = (float *)malloc(sizeof(float) * 2 * T * N * M);
        CHECK(qdas_convd(&cd, dx, dtaps, dz, NULL) == 0 && qdas_device_copy(zc, dz, sizeof(float) * 2 * T * N * M, 1, -1) == 0);
        CHECK(memcmp(zc, mxGetData(out[0]), sizeof(float) * 2 * T * N * M) == 0 && (zc[100] != 0 || zc[101] != 0));
        /* 6e. hilbert (+ downmix): real traces -> analytic, downmixed channel data */
        mxArray *cmd_hilbert = mxCreateString("hilbert"), *psz = arr(1, 4, mxDOUBLE_CLASS, 0), *xr = arr(T, N * M, mxSINGLE_CLASS, 0), *ptv = arr(1, 3, mxDOUBLE_CLASS, 0);
        const double pszv[4] = {T, N * M, 0, QDAS_PRE_F32}, ptvv[3] = {20e6, 1e-6, 5e6};
        memcpy(mxGetData(psz), pszv, sizeof pszv); memcpy(mxGetData(ptv), ptvv, sizeof ptvv);
        float *xrp = (float *)mxGetData(xr);
        for (int k = 0; k < T * N * M; ++k) xrp[k] = xp[2 * k];
        const mxArray *p1[4] = {cmd_hilbert, psz, xr, ptv};
        CHECK(call(1, out, 4, p1) == 0);
        CHECK(mxIsComplex(out[0]) && mxGetNumberOfElements(out[0]) == (size_t)T * N * M);
        qdas_pre_desc pd;
        memset(&pd, 0, sizeof pd);
        pd.T = T; pd.K = N * M; pd.in_type = QDAS_PRE_F32; pd.device = -1; pd.fs = ptvv[0]; pd.t0 = ptvv[1]; pd.fdown = ptvv[2];
        qdas_pre_plan *pp = NULL;
        void *dxr = dup_dev(xrp, sizeof(float) * T * N * M);
        CHECK(dxr && qdas_pre_plan_create(&pp, &pd) == 0 && qdas_pre_execute(pp, dxr, dz, NULL) == 0 && qdas_device_copy(zc, dz, sizeof(float) * 2 * T * N * M, 1, -1) == 0);
        qdas_pre_plan_destroy(pp);
        CHECK(memcmp(zc, mxGetData(out[0]), sizeof(float) * 2 * T * N * M) == 0 && (zc[100] != 0 || zc[101] != 0));
        /* 6e'. wsinterpd (the general single-delay launch): x T x N x M sampled at t (I x N, broadcast over the transmits), weights 1 x N, summed over the receivers */
        {
            mxArray *cmd_ws = mxCreateString("wsinterpd"), *wsz = arr(1, 6, mxDOUBLE_CLASS, 0), *wsize = arr(1, 3, mxDOUBLE_CLASS, 0), *wstr = arr(3, 3, mxINT64_CLASS, 0),
                    *wsum = arr(1, 3, mxDOUBLE_CLASS, 0), *ww = arr(1, N, mxSINGLE_CLASS, 0), *wtv = arr(1, 2, mxDOUBLE_CLASS, 0);
            const double wszv[6] = {T, 1, 3, 2 /* cubic */, 1 /* single */, 1 /* real weights */}, wsizev[3] = {I, N, M}, wsumv[3] = {0, 1, 0}, wtvv[2] = {0.3, 0.0};
            const int64_t wstrv[9] = {1, 0, 0, /* dim 1: t, x, w */ I, T, 1, /* dim 2 */ 0, (int64_t)T * N, 0 /* dim 3 */};
            memcpy(mxGetData(wsz), wszv, sizeof wszv); memcpy(mxGetData(wsize), wsizev, sizeof wsizev); memcpy(mxGetData(wsum), wsumv, sizeof wsumv);
            memcpy(mxGetData(wstr), wstrv, sizeof wstrv); memcpy(mxGetData(wtv), wtvv, sizeof wtvv);
            for (int n = 0; n < N; ++n) ((float *)mxGetData(ww))[n] = 0.5f + 0.1f * n;
            const mxArray *w1[9] = {cmd_ws, wsz, wsize, wstr, wsum, ww, x1, t1, wtv};
            CHECK(call(1, out, 9, w1) == 0);
            CHECK(mxIsComplex(out[0]) && mxGetNumberOfElements(out[0]) == (size_t)I * M);
            qdas_wsinterpd_desc wd;
            memset(&wd, 0, sizeof wd);
            wd.T = T; wd.x_tstride = 1; wd.ndim = 3; wd.flag = 2; wd.dtype = QDAS_F32; wd.w_real = 1; wd.lane_dim = -1;
            for (int k = 0; k < 3; ++k) { wd.size[k] = (uint64_t)wsizev[k]; wd.tstride[k] = wstrv[3 * k]; wd.xstride[k] = wstrv[3 * k + 1]; wd.wstride[k] = wstrv[3 * k + 2]; wd.sum[k] = wsumv[k] != 0; }
            wd.omega = 0.3; wd.extrap = 0.0;
            void *dww = dup_dev(mxGetData(ww), sizeof(float) * N);
            wd.t = dt1; wd.x = dx; wd.w = dww;
            CHECK(dww && qdas_wsinterpd(&wd, dz, NULL) == 0 && qdas_device_copy(zc, dz, sizeof(float) * 2 * I * M, 1, -1) == 0);
            CHECK(memcmp(zc, mxGetData(out[0]), sizeof(float) * 2 * I * M) == 0 && (zc[10] != 0 || zc[11] != 0));
            CHECK(qdas_device_free(dww, -1) == 0);
            /* 6e''. shiftsum (focusTx): 3 synthesised transmits from the M elements of one frame */
            enum { MO = 3, TO = 280 };
            mxArray *cmd_ss = mxCreateString("shiftsum"), *ssz = arr(1, 10, mxDOUBLE_CLASS, 0), *sh = arr(M, MO, mxSINGLE_CLASS, 0), *sw = arr(M, MO, mxSINGLE_CLASS, 0);
            const double sszv[10] = {T, TO, N, M, MO, 1, 2 /* cubic */, 1 /* single */, 1 /* complex */, 1 /* real weights */};
            memcpy(mxGetData(ssz), sszv, sizeof sszv);
            for (int k = 0; k < M * MO; ++k) { ((float *)mxGetData(sh))[k] = 3.25f + 0.7f * (k % 5); ((float *)mxGetData(sw))[k] = (k % 4) ? 1.0f - 0.05f * k : 0.0f; }
            const mxArray *s1[5] = {cmd_ss, ssz, x1, sh, sw};
            CHECK(call(1, out, 5, s1) == 0);
            CHECK(mxIsComplex(out[0]) && mxGetNumberOfElements(out[0]) == (size_t)TO * N * MO);
            qdas_shift_desc sd;
            memset(&sd, 0, sizeof sd);
            sd.T = T; sd.To = TO; sd.N = N; sd.M = M; sd.Mo = MO; sd.F = 1; sd.flag = 2; sd.dtype = QDAS_F32; sd.cplx = 1; sd.w_real = 1; sd.device = -1;
            void *dsh = dup_dev(mxGetData(sh), sizeof(float) * M * MO), *dsw = dup_dev(mxGetData(sw), sizeof(float) * M * MO);
            sd.shift = dsh; sd.w = dsw;
            CHECK(dsh && dsw && qdas_shift_sum(&sd, dx, dz, NULL) == 0 && qdas_device_copy(zc, dz, sizeof(float) * 2 * TO * N * MO, 1, -1) == 0);
            { const float *g = (const float *)mxGetData(out[0]); long nd = 0, first = -1; for (long k = 0; k < 2L * TO * N * MO; ++k) if (memcmp(&g[k], &zc[k], 4)) { if (first < 0) first = k; ++nd; }
              if (nd) { printf("shiftsum: %ld of %ld floats differ, first at %ld: gateway %.9g, C ABI %.9g\n", nd, 2L * TO * N * MO, first, g[first], zc[first]);
                float *z2 = (float *)malloc(sizeof(float) * 2 * TO * N * MO);
                qdas_shift_sum(&sd, dx, dz, NULL); qdas_device_copy(z2, dz, sizeof(float) * 2 * TO * N * MO, 1, -1);
                mxArray *o2[1]; call(1, o2, 5, s1);
                printf("  again: C ABI %s its first result (%.9g), gateway %s its first result (%.9g); second results %s\n", memcmp(z2, zc, sizeof(float) * 2 * TO * N * MO) ? "DIFFERS from" : "equals", z2[first],
                       memcmp(mxGetData(o2[0]), g, sizeof(float) * 2 * TO * N * MO) ? "DIFFERS from" : "equals", ((float *)mxGetData(o2[0]))[first], memcmp(mxGetData(o2[0]), z2, sizeof(float) * 2 * TO * N * MO) ? "differ" : "agree"); } }
            CHECK(memcmp(zc, mxGetData(out[0]), sizeof(float) * 2 * TO * N * MO) == 0 && (zc[100] != 0 || zc[101] != 0));
            CHECK(qdas_device_free(dsh, -1) == 0 && qdas_device_free(dsw, -1) == 0);
        }
        /* 6f. their error paths: wrong argument counts, a wrong class, a short array */
        CHECK(call(1, out, 3, v1) == 1 && strstr(fake_mex_last_id, "QUPS:das_spec:nargin"));
        const mxArray *p2[4] = {cmd_hilbert, psz, x1 /* complex */, ptv};
        CHECK(call(1, out, 4, p2) == 1 && strstr(fake_mex_last_msg, "real"));
        const mxArray *c2[8] = {cmd_lut, ls, apod, xbad, t1, t2, wst, om};
        CHECK(call(1, out, 8, c2) == 1 && strstr(fake_mex_last_msg, "bytes expected"));
        void *frees[] = {dPi, dPr, dPv, dNv, dtau, dx, dt1, dt2, dw, dy, (void *)gd.Ps, (void *)gd.a, (void *)gd.Pr, (void *)gd.Pv, (void *)gd.x, dgy, dtaps, dz, dxr};
        for (size_t k = 0; k < sizeof frees / sizeof frees[0]; ++k) CHECK(qdas_device_free(frees[k], -1) == 0);
        printf("delays / lut / greens / convd / hilbert / wsinterpd / shiftsum through the gateway: bit-identical to the C ABI\n");
    }
    fake_mex_run_atexit();                                                            /* 'clear mex' */
    CHECK(!mexIsLocked());
    printf("fake-MEX gateway OK\n");
    return 0;
}
