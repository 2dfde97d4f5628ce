/* fake_mex_runtime.c -- malloc-backed implementation of the few mx / mex functions declared in tests/fake_mex/mex.h (TEST
 * INFRASTRUCTURE: lets the gateway's mexFunction run against the real libqdas.so without MATLAB).  Errors longjmp back to the
 * driver like MATLAB's own mexErrMsgIdAndTxt never returns. */
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mex.h"

struct mxArray_tag {
    mxClassID cls; int cplx; mwSize ndim, dims[8]; void *data;
    int nfields; const char *fnames[8]; mxArray *fvals[8];
};
jmp_buf fake_mex_jmp;
char fake_mex_last_id[128], fake_mex_last_msg[1024];
static int g_locked = 0;
static void (*g_atexit_fn)(void) = NULL;

static size_t elsize(mxClassID c) {
    switch (c) { case mxDOUBLE_CLASS: case mxINT64_CLASS: case mxUINT64_CLASS: return 8; case mxSINGLE_CLASS: case mxINT32_CLASS: case mxUINT32_CLASS: return 4;
                 case mxINT16_CLASS: case mxUINT16_CLASS: case mxCHAR_CLASS: return 2; default: return 1; }
}
mxClassID mxGetClassID(const mxArray *a) { return a->cls; }
void *mxGetData(const mxArray *a) { return a->data; }
size_t mxGetNumberOfElements(const mxArray *a) { size_t n = 1; for (mwSize k = 0; k < a->ndim; ++k) n *= a->dims[k]; return n; }
int mxIsEmpty(const mxArray *a) { return mxGetNumberOfElements(a) == 0; }
int mxIsComplex(const mxArray *a) { return a->cplx; }
int mxIsChar(const mxArray *a) { return a->cls == mxCHAR_CLASS; }
int mxIsStruct(const mxArray *a) { return a->cls == mxSTRUCT_CLASS; }
int mxGetString(const mxArray *a, char *buf, mwSize buflen) {
    const size_t n = mxGetNumberOfElements(a);
    if (n + 1 > buflen) return 1;
    for (size_t k = 0; k < n; ++k) buf[k] = (char)((const uint16_t *)a->data)[k];
    buf[n] = 0;
    return 0;
}
mxArray *mxGetField(const mxArray *a, mwIndex i, const char *name) {
    (void)i;
    for (int k = 0; k < a->nfields; ++k) if (!strcmp(a->fnames[k], name)) return a->fvals[k];
    return NULL;
}
mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID cls, mxComplexity c) {
    mxArray *a = (mxArray *)calloc(1, sizeof *a);
    a->cls = cls; a->cplx = c == mxCOMPLEX; a->ndim = ndim;
    size_t n = 1;
    for (mwSize k = 0; k < ndim; ++k) { a->dims[k] = dims[k]; n *= dims[k]; }
    a->data = calloc(n ? n : 1, elsize(cls) * (a->cplx ? 2 : 1));
    return a;
}
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity c) { const mwSize d[2] = {m, n}; return mxCreateNumericArray(2, d, cls, c); }
mxArray *mxCreateString(const char *s) {
    const mwSize d[2] = {1, strlen(s)};
    mxArray *a = mxCreateNumericArray(2, d, mxCHAR_CLASS, mxREAL);
    for (size_t k = 0; k < d[1]; ++k) ((uint16_t *)a->data)[k] = (uint16_t)(unsigned char)s[k];
    return a;
}
void mxDestroyArray(mxArray *a) { if (a) { free(a->data); free(a); } }
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(fake_mex_last_msg, sizeof fake_mex_last_msg, fmt, ap);
    va_end(ap);
    snprintf(fake_mex_last_id, sizeof fake_mex_last_id, "%s", id);
    longjmp(fake_mex_jmp, 1);
}
void mexLock(void) { g_locked = 1; }
void mexUnlock(void) { g_locked = 0; }
int mexIsLocked(void) { return g_locked; }
int mexAtExit(void (*fn)(void)) { g_atexit_fn = fn; return 0; }
void fake_mex_run_atexit(void) { if (g_atexit_fn) g_atexit_fn(); }
/* helper for the driver: a struct with up to 8 fields */
mxArray *fake_mex_struct(int n, const char **names, mxArray **vals) {
    mxArray *a = (mxArray *)calloc(1, sizeof *a);
    a->cls = mxSTRUCT_CLASS; a->ndim = 2; a->dims[0] = a->dims[1] = 1; a->nfields = n;
    for (int k = 0; k < n; ++k) { a->fnames[k] = names[k]; a->fvals[k] = vals[k]; }
    return a;
}
