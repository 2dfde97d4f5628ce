/* run_gateway.c -- drives mex/qdas_mex.c's mexFunction through its sub-commands over the fake MEX runtime and the REAL libqdas.so,
 * and checks the results against the C ABI called directly (TEST INFRASTRUCTURE; needs a GPU).  Prints "fake-MEX gateway OK". */
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
    fake_mex_run_atexit();                                                            /* 'clear mex' */
    CHECK(!mexIsLocked());
    printf("fake-MEX gateway OK\n");
    return 0;
}
