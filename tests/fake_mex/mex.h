/* tests/fake_mex/mex.h -- a STAND-IN for MATLAB's mex.h, for testing mex/qdas_mex.c only.
 *
 * There is no MATLAB (hence no mex.h) in this repository's environment.  This header declares exactly the mx / mex functions the
 * gateway uses, with the signatures of the documented C Matrix API (interleaved-complex, -R2018a); fake_mex_runtime.c implements
 * them over malloc so that tests/test_mex_gateway.py can compile the gateway (syntax / ABI check) and drive mexFunction through
 * create / execute / destroy against the real libqdas.so (logic check).  It is NOT a MATLAB and proves nothing about MATLAB's
 * own behaviour. */
#ifndef QDAS_FAKE_MEX_H
#define QDAS_FAKE_MEX_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef size_t mwSize;
typedef size_t mwIndex;
typedef struct mxArray_tag mxArray;
typedef enum { mxUNKNOWN_CLASS = 0, mxCELL_CLASS, mxSTRUCT_CLASS, mxLOGICAL_CLASS, mxCHAR_CLASS, mxVOID_CLASS, mxDOUBLE_CLASS, mxSINGLE_CLASS,
               mxINT8_CLASS, mxUINT8_CLASS, mxINT16_CLASS, mxUINT16_CLASS, mxINT32_CLASS, mxUINT32_CLASS, mxINT64_CLASS, mxUINT64_CLASS } mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX } mxComplexity;

mxClassID mxGetClassID(const mxArray *a);
void *mxGetData(const mxArray *a);
size_t mxGetNumberOfElements(const mxArray *a);
int mxIsEmpty(const mxArray *a);
int mxIsComplex(const mxArray *a);
int mxIsChar(const mxArray *a);
int mxIsStruct(const mxArray *a);
int mxGetString(const mxArray *a, char *buf, mwSize buflen);
mxArray *mxGetField(const mxArray *a, mwIndex i, const char *name);
mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID cls, mxComplexity c);
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity c);
mxArray *mxCreateString(const char *s);
void mxDestroyArray(mxArray *a);
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...);
void mexLock(void);
void mexUnlock(void);
int mexIsLocked(void);
int mexAtExit(void (*fn)(void));
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
#ifdef __cplusplus
}
#endif
#endif
