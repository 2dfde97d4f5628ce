/* qdas_mex.c -- thin MEX gateway from MATLAB to libqdas.so (C ABI in include/qdas.h).
 *
 * NOT compiled in this repository's environment (no MATLAB / mex.h here); it is the binding a QUPS maintainer adds:
 *
 *     mex -R2018a -I<repo>/include -L<repo>/qups_amd -lqdas mex/qdas_mex.c -output bin/qdas_mex
 *
 * It replaces ONE call site of the reference: the per-frame CUDA launch inside das_spec
 *     y{f} = k.feval(yg, Pi, Pr, Pv, Nv, apod, cinv, [cstride, astride], x(:,:,:,f), [fs, fmod]);   (kern/das_spec.m:372)
 * together with the kernel construction / constant upload before it (kern/das_spec.m:279-306).
 *
 * MATLAB usage (see INTEGRATION.md for the das_spec.m patch):
 *     y = qdas_mex(sizes, Pi, Pr, Pv, Nv, apod, cinv, acstride, x, [fs fmod])
 *   sizes    : int64/double row [T N M I1 I2 I3 S flag VS DV dtype F]   (dtype: 0 double, 1 single, 2 halfT-as-uint16 pairs)
 *   Pi..Nv   : real(prec) arrays laid out 3xI, 3xN, 4xM (row 4 = t0), 3xM   (host arrays, interleaved-complex API)
 *   apod     : complex(prec) column (concatenated arrays) or [] ; cinv: real(prec) array
 *   acstride : uint64 6 x (1+S)                                           (kern/das_spec.m:257-260)
 *   x        : complex(prec) T x N x M x F ;  y: complex(prec) I x [N] x [M] x F
 * Host buffers are passed with QDAS_MEM_HOST: the library stages them through HBM itself.  (With the Parallel Computing
 * Toolbox, mxGPUArray device pointers can be passed with QDAS_MEM_DEVICE instead; omitted here.)
 */
#include <string.h>
#include "mex.h"
#include "qdas.h"

static double scalar_at(const mxArray *a, mwSize k) { return mxGetPr(a)[k]; }

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    if (nrhs != 10) mexErrMsgIdAndTxt("QUPS:das_spec:nargin", "qdas_mex expects 10 inputs.");
    const mxArray *sz = prhs[0];
    if (mxGetNumberOfElements(sz) < 12) mexErrMsgIdAndTxt("QUPS:das_spec:sizes", "sizes must have 12 entries.");
    qdas_desc d;
    memset(&d, 0, sizeof d);
    d.sz.T = (uint64_t)scalar_at(sz, 0);  d.sz.N = (uint64_t)scalar_at(sz, 1);  d.sz.M = (uint64_t)scalar_at(sz, 2);
    d.sz.I1 = (uint64_t)scalar_at(sz, 3); d.sz.I2 = (uint64_t)scalar_at(sz, 4); d.sz.I3 = (uint64_t)scalar_at(sz, 5);
    d.sz.S = (uint64_t)scalar_at(sz, 6);  d.sz.flag = (int32_t)scalar_at(sz, 7);
    d.sz.VS = (int32_t)scalar_at(sz, 8);  d.sz.DV = (int32_t)scalar_at(sz, 9);  d.sz.dtype = (int32_t)scalar_at(sz, 10);
    const uint64_t F = (uint64_t)scalar_at(sz, 11);
    d.Pi = mxGetData(prhs[1]); d.Pr = mxGetData(prhs[2]); d.Pv = mxGetData(prhs[3]); d.Nv = mxGetData(prhs[4]);
    d.apod = mxIsEmpty(prhs[5]) ? NULL : mxGetData(prhs[5]);
    d.cinv = mxGetData(prhs[6]);
    d.acstride = (const uint64_t *)mxGetData(prhs[7]);
    const double *tv = mxGetPr(prhs[9]);
    d.fs = tv[0]; d.fmod = tv[1];
    d.mem = QDAS_MEM_HOST; d.device = -1; d.kernel = QDAS_KERNEL_AUTO;

    qdas_plan *plan = NULL;
    if (qdas_plan_create(&plan, &d)) mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "%s", qdas_last_error());

    const uint64_t I = d.sz.I1 * d.sz.I2 * d.sz.I3;
    const uint64_t oN = (d.sz.flag & QDAS_FLAG_KEEP_RX) ? d.sz.N : 1, oM = (d.sz.flag & QDAS_FLAG_KEEP_TX) ? d.sz.M : 1;
    const mwSize dims[4] = {(mwSize)I, (mwSize)oN, (mwSize)oM, (mwSize)F};
    const mxClassID cls = d.sz.dtype == QDAS_F64 ? mxDOUBLE_CLASS : (d.sz.dtype == QDAS_F32 ? mxSINGLE_CLASS : mxUINT16_CLASS);
    plhs[0] = mxCreateNumericArray(4, dims, cls, mxCOMPLEX);
    const int rc = qdas_plan_execute_frames(plan, mxGetData(prhs[8]), mxGetData(plhs[0]), F,
                                            d.sz.T * d.sz.N * d.sz.M, I * oN * oM, NULL);
    qdas_plan_destroy(plan);
    if (rc) mexErrMsgIdAndTxt("QUPS:das_spec:qdas", "%s", qdas_last_error());
}
