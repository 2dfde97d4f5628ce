/* tests/fake_mex/gpu/mxGPUArray.h -- STAND-IN for the Parallel Computing Toolbox header, prototypes only (compile check of the
 * -DQDAS_MEX_GPU branch of mex/qdas_mex.c; nothing implements them here). */
#ifndef QDAS_FAKE_MXGPUARRAY_H
#define QDAS_FAKE_MXGPUARRAY_H
#include "mex.h"
typedef struct mxGPUArray_tag mxGPUArray;
typedef enum { MX_GPU_DO_NOT_INITIALIZE = 0, MX_GPU_INITIALIZE_VALUES = 1 } mxGPUInitialize;
int mxInitGPU(void);
int mxIsGPUArray(const mxArray *a);
const mxGPUArray *mxGPUCreateFromMxArray(const mxArray *a);
const void *mxGPUGetDataReadOnly(const mxGPUArray *g);
void *mxGPUGetData(mxGPUArray *g);
mxGPUArray *mxGPUCreateGPUArray(mwSize ndims, const mwSize *dims, mxClassID cls, mxComplexity c, mxGPUInitialize init);
mxArray *mxGPUCreateMxArrayOnGPU(const mxGPUArray *g);
void mxGPUDestroyGPUArray(const mxGPUArray *g);
#endif
