// layout.hip -- column-major image of a row-major A x B x C array: out[c][b][a] = in[a][b][c].  No reference counterpart (MATLAB
// arrays already are column-major); the Python mirror uses it to hand row-major torch / numpy data (T x N x M channel data,
// I1 x I2 x N delay tables) to the C ABI, which follows the reference's memory order.  64 x 64 tiles of (a, c) through LDS:
// reads coalesced along c, writes coalesced along a.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qdas {

template <typename E>
__global__ void __launch_bounds__(256) permute3_kernel(const E *__restrict__ in, E *__restrict__ out, uint64_t A, uint64_t B, uint64_t C) {
    __shared__ E tile[64][65];
    const uint64_t b = blockIdx.z;
    const uint64_t a0 = (uint64_t)blockIdx.y * 64, c0 = (uint64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
        const uint64_t a = a0 + r, c = c0 + tx;
        if (a < A && c < C) tile[r][tx] = in[(a * B + b) * C + c];
    }
    __syncthreads();
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
        const uint64_t c = c0 + r, a = a0 + tx;
        if (a < A && c < C) out[(c * B + b) * A + a] = tile[tx][r];
    }
}

hipError_t launch_permute3(const void *in, void *out, uint64_t A, uint64_t B, uint64_t C, int elem_bytes, hipStream_t s) {
    if (A == 0 || B == 0 || C == 0) return hipSuccess;
    if (B > 65535 || (A + 63) / 64 > 65535) return hipErrorInvalidValue;
    const dim3 g((unsigned)((C + 63) / 64), (unsigned)((A + 63) / 64), (unsigned)B);
    switch (elem_bytes) {
        case 2:  permute3_kernel<uint16_t><<<g, 256, 0, s>>>((const uint16_t *)in, (uint16_t *)out, A, B, C); break;
        case 4:  permute3_kernel<uint32_t><<<g, 256, 0, s>>>((const uint32_t *)in, (uint32_t *)out, A, B, C); break;
        case 8:  permute3_kernel<uint2><<<g, 256, 0, s>>>((const uint2 *)in, (uint2 *)out, A, B, C); break;
        case 16: permute3_kernel<uint4><<<g, 256, 0, s>>>((const uint4 *)in, (uint4 *)out, A, B, C); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace qdas
