/* das_ref.c -- CPU oracle for the QUPS DAS hot path (TEST INFRASTRUCTURE ONLY).
 *
 * Plain C (+ optional OpenMP) restatement of the reference's CPU branch
 * kern/das_spec.m:391-560 with the interpolator definitions of
 * src/interpd.cu:68-150; the body (with per-line citations) is in
 * das_ref_body.inc, instantiated for float and double.  It is used
 *   - by tests/ as a second, independently written checker of the numpy oracle
 *     (das_oracle.py) and of the stride-table ABI, and
 *   - by bench.py's `cpu_baseline` leg (kind "port": the reference's CPU path is
 *     MATLAB and cannot run here).
 * The product (qups_amd/) never links or calls this.
 *
 * Parity pinning: see the header of das_oracle.py ("cubic"/"lanczos3" are
 * unpinned by the reference itself).
 *
 * Build: make -C oracle   (gcc -O3 -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct das_ref_sizes {
    uint64_t T, N, M, I, I1, I2, I3, S;   /* QUPS_T ... QUPS_S (src/sizes.cu) */
    int32_t flag;                         /* QUPS_BF_FLAG: bits 0-2 interp, 3 keep_rx, 4 keep_tx, 5 tpose */
    int32_t VS, DV;                       /* QUPS_VS, QUPS_DV */
} das_ref_sizes;

#define R float
#define FN(name) name##_f32
#define FLOORR floorf
#define SQRTR sqrtf
#define COPYSIGNR copysignf
#include "das_ref_body.inc"
#undef R
#undef FN
#undef FLOORR
#undef SQRTR
#undef COPYSIGNR

#define R double
#define FN(name) name##_f64
#define FLOORR floor
#define SQRTR sqrt
#define COPYSIGNR copysign
#include "das_ref_body.inc"
#undef R
#undef FN
#undef FLOORR
#undef SQRTR
#undef COPYSIGNR

int das_ref_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
