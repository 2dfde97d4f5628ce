// pre.hip -- the step in front of the DAS path: real RF traces -> analytic (complex) channel data, optionally downmixed.
//
// Replaces ChannelData.hilbert (reference src/ChannelData.m:935-966: fft along time to N points, weights
// w = [1; 2 x (floor(N/2)-1); 1 + mod(N,2); 0 ...], ifft) fused with ChannelData.downmix (src/ChannelData.m:757-766:
// data .* exp(-2i*pi*fc*time)), so that real (fp32 or int16) traces can be uploaded -- half / a quarter of the bytes of
// complex64 channel data over PCIe -- and turned into the beamformer's input on the device.  The FFTs are hipFFT plans
// (a library FFT, like the reference's MATLAB fft); the packing, weighting and downmix kernels are HIP.
//
// For real input, ifft(w .* fft(x)) = x + i*H(x): the real part IS the (padded / truncated) input, only the imaginary part
// needs an inverse transform, and it is real: H(x) = C2R of (-i sgn(f) X(f)) over the half spectrum.  So the pass is
// pad -> R2C -> half-spectrum weights -> C2R -> interleave (x, H(x)) [* phasor]: two half-size real transforms instead of a real
// forward + a full complex inverse one (whose length-2816 decomposition costs hipFFT two extra transposes).
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <stdint.h>

namespace qdas {

// real traces (T x K, fp32 or int16) -> zero-padded / truncated fp32 (N x K)
template <typename TI>
__global__ void __launch_bounds__(256) pre_pad_kernel(const TI *x, float *xr, uint64_t T, uint64_t N, uint64_t K) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N * K) return;
    const uint64_t t = i % N, k = i / N;
    xr[i] = t < T ? (float)x[t + T * k] : 0.f;
}

// half spectrum X (N/2+1 bins per trace) -> spectrum of the Hilbert transform, scaled by the 1/N of the inverse transform:
// -i*sgn(f)*X(f)/N for 0 < f < N/2, 0 at DC and (even N) at Nyquist -- the imaginary part of what the reference's weights
// [1; 2...; 1 + mod(N,2); 0...] produce (src/ChannelData.m:961-963)
__global__ void __launch_bounds__(256) pre_spectrum_kernel(float2 *half, uint64_t N, uint64_t K, float scale) {
    const uint64_t H = N / 2 + 1;
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= H * K) return;
    const uint64_t f = i % H;
    const bool zero = (f == 0) || ((N & 1) == 0 && f == N / 2);
    const float2 v = half[i];
    half[i] = zero ? make_float2(0.f, 0.f) : make_float2(v.y * scale, -v.x * scale);
}

// y = (x, H(x)) times the downmix phasor exp(-2j pi fd (t0 + t/fs)) (fd == 0: none)
__global__ void __launch_bounds__(256) pre_finish_kernel(const float *xr, const float *hr, float2 *y, uint64_t N, uint64_t K, double fd, double t0, double fs) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N * K) return;
    float2 v = make_float2(xr[i], hr[i]);
    if (fd != 0.0) {
        const double cyc = fd * (t0 + (double)(i % N) / fs);           // cycles; reduced in fp64 before the fp32 sincos
        const float ph = (float)(cyc - floor(cyc));
        const float c = __builtin_amdgcn_cosf(ph), s = -__builtin_amdgcn_sinf(ph);
        v = make_float2(v.x * c - v.y * s, v.x * s + v.y * c);
    }
    y[i] = v;
}

struct PrePlan {
    uint64_t T, K, N;
    int in_type;          // 0: fp32, 1: int16
    double fs, t0, fd;
    hipfftHandle r2c = 0, c2r = 0;
    float *xr = nullptr;  // N x K real (padded input)
    float *hr = nullptr;  // N x K real (its Hilbert transform)
    float2 *half = nullptr;
    bool have = false;
};

int pre_create(PrePlan **out, uint64_t T, uint64_t K, uint64_t N, int in_type, double fs, double t0, double fd) {
    PrePlan *p = new PrePlan();
    p->T = T; p->K = K; p->N = N ? N : T; p->in_type = in_type; p->fs = fs; p->t0 = t0; p->fd = fd;
    if (p->N == 0 || K == 0) { *out = p; return 0; }
    int n[1] = {(int)p->N};
    if (hipfftPlanMany(&p->r2c, 1, n, nullptr, 1, (int)p->N, nullptr, 1, (int)(p->N / 2 + 1), HIPFFT_R2C, (int)K) != HIPFFT_SUCCESS) { delete p; return 1; }
    if (hipfftPlanMany(&p->c2r, 1, n, nullptr, 1, (int)(p->N / 2 + 1), nullptr, 1, (int)p->N, HIPFFT_C2R, (int)K) != HIPFFT_SUCCESS) { hipfftDestroy(p->r2c); delete p; return 1; }
    if (hipMalloc(&p->xr, sizeof(float) * p->N * K) != hipSuccess || hipMalloc(&p->hr, sizeof(float) * p->N * K) != hipSuccess ||
        hipMalloc(&p->half, sizeof(float2) * (p->N / 2 + 1) * K) != hipSuccess) {
        hipfftDestroy(p->r2c); hipfftDestroy(p->c2r); if (p->xr) (void)hipFree(p->xr); if (p->hr) (void)hipFree(p->hr); delete p; return 2;
    }
    p->have = true;
    *out = p;
    return 0;
}

void pre_destroy(PrePlan *p) {
    if (!p) return;
    if (p->have) { hipfftDestroy(p->r2c); hipfftDestroy(p->c2r); (void)hipFree(p->xr); (void)hipFree(p->hr); (void)hipFree(p->half); }
    delete p;
}

int pre_execute(PrePlan *p, const void *x, void *y, hipStream_t s) {
    if (!p->have) return 0;
    const uint64_t NK = p->N * p->K;
    const unsigned g = (unsigned)((NK + 255) / 256);
    if (p->in_type == 1) pre_pad_kernel<int16_t><<<g, 256, 0, s>>>((const int16_t *)x, p->xr, p->T, p->N, p->K);
    else pre_pad_kernel<float><<<g, 256, 0, s>>>((const float *)x, p->xr, p->T, p->N, p->K);
    if (hipfftSetStream(p->r2c, s) != HIPFFT_SUCCESS || hipfftSetStream(p->c2r, s) != HIPFFT_SUCCESS) return 1;
    if (hipfftExecR2C(p->r2c, p->xr, (hipfftComplex *)p->half) != HIPFFT_SUCCESS) return 1;
    const uint64_t HK = (p->N / 2 + 1) * p->K;
    pre_spectrum_kernel<<<(unsigned)((HK + 255) / 256), 256, 0, s>>>(p->half, p->N, p->K, 1.0f / (float)p->N);
    if (hipfftExecC2R(p->c2r, (hipfftComplex *)p->half, p->hr) != HIPFFT_SUCCESS) return 1;
    pre_finish_kernel<<<g, 256, 0, s>>>(p->xr, p->hr, (float2 *)y, p->N, p->K, p->fd, p->t0, p->fs);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

}  // namespace qdas
