// das_tile.hip -- the fused, LDS-staged delay-and-sum kernel for gfx950 (MI355X).
//
// Replaces the reference launch of `DASf` / `DASh` (reference src/bf.cu:153-171, body
// src/bf.cu:49-142) for the bulk case: sum over both apertures ('DAS'), scalar sound speed,
// apodization that does not depend on the pixel (folded by the host into one N x M table).
// Everything else is served by das_generic.hip.
//
// Design (MI355X-first, not a re-tiling of the reference's one-thread-per-pixel loop):
//
//  * A workgroup owns a TILE of 64 (fast image axis I1 = depth) x TX (columns) pixels.  A wave's
//    64 lanes are 64 consecutive depth pixels, so for any trace (n, m) the lanes read
//    neighbouring fast-time samples.
//  * Time of flight is separable: tau*fs + off = a(i,m) + b(i,n).  The prologue computes, in
//    fp64, the tile-wide integer window bases A[m] <= a, B[n] <= b and extents; afterwards each
//    lane only carries the small fp32 residuals ra = a - A[m], rb = b - B[n] (exact to ~1e-5
//    sample; the reference's fp32 tau carries ~1e-4 sample at tau*fs ~ 2000).  The per-pair
//    address is then ONE add:  tr = ra[m] + rb,  tap index = (uint)tr inside the staged window.
//  * For every (receiver n, block of MB transmits) the workgroup stages MB fast-time WINDOWS
//    (W samples starting at A[m]+B[n]) of the channel data into LDS, coalesced along fast time
//    and double-buffered against the compute of the previous stage; taps are then gathered from
//    LDS with ds_read_b64 (fp32) / ds_read_b32 (fp16), not from global memory.
//  * All resident workgroups walk the traces in the same order, so the channel data streams
//    from HBM about once per "round" of tiles and is otherwise served by L2 / Infinity Cache.
//  * A stage whose windows lie completely inside [0, T) takes a branch-free path; stages that
//    touch the ends of the record take the checked path (edge rule of SURVEY.md section 8 a5).
//  * A tile whose delay spread does not fit W appends itself to a fallback list and is
//    processed by the generic kernel afterwards -- results never depend on the geometry being
//    "image like".
//  * Lanczos weights: even/plain polynomials (lanczos_poly.h), no transcendentals; fp16 data is
//    accumulated in fp32 (the reference accumulates in half2, src/bf.cu:170).
#include "qdas_device.h"
#include "qdas_kernels.h"
#include "lanczos_poly.h"

namespace qdas {

constexpr int TZ = 64;            // pixels along I1 per tile == wave width
constexpr int WAVES = 4;          // waves per workgroup
constexpr int THREADS = WAVES * 64;

template <int INTERP> struct tapinfo {
    static constexpr int K = interp_taps(INTERP);
    // offset folded into a(i,m) so that floor(a + b) is the FIRST tap:
    //   nearest: round(tau) = floor(tau + 1/2); linear: floor(tau); 4-tap: floor(tau) - 1
    static constexpr double OFF = (INTERP == 0) ? 0.5 : (K == 2 ? 0.0 : -1.0);
    // lowest admissible value of (tau*fs + OFF): tau >= 0 AND first tap >= 0
    static constexpr float LO = (INTERP == 0) ? 0.5f : 0.0f;
};

template <int D> __device__ __forceinline__ float horner(const float (&c)[D + 1], float t) {
    float r = c[D];
#pragma unroll
    for (int k = D - 1; k >= 0; --k) r = fmaf(r, t, c[k]);
    return r;
}

template <int INTERP> __device__ __forceinline__ void tile_weights(float u, float w[4]) {
    if constexpr (INTERP == 3) {
        constexpr float ein[QDAS_LANCZOS_DIN + 1] = QDAS_LANCZOS_EIN;
        constexpr float pout[QDAS_LANCZOS_DOUT + 1] = QDAS_LANCZOS_POUT;
        const float v = 1.0f - u;
        w[0] = horner<QDAS_LANCZOS_DOUT>(pout, u);
        w[1] = horner<QDAS_LANCZOS_DIN>(ein, u * u);
        w[2] = horner<QDAS_LANCZOS_DIN>(ein, v * v);
        w[3] = horner<QDAS_LANCZOS_DOUT>(pout, v);
    } else {
        interp_weights<INTERP, float>(u, w);
    }
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// LDS sample -> fp32 complex
__device__ __forceinline__ cplx<float> lds_ld(const float2 *p) { const float2 v = *p; return {v.x, v.y}; }
__device__ __forceinline__ cplx<float> lds_ld(const uint32_t *p) {
    const uint32_t v = *p;
    return {__half2float(__ushort_as_half((unsigned short)(v & 0xffffu))),
            __half2float(__ushort_as_half((unsigned short)(v >> 16)))};
}
__device__ __forceinline__ float2   zero_of(const float2 *)   { return make_float2(0.f, 0.f); }
__device__ __forceinline__ uint32_t zero_of(const uint32_t *) { return 0u; }

template <int INTERP, typename ST, bool FMOD, bool WTAB, int CPW, int MB, int W>
__global__ void __launch_bounds__(THREADS)
das_tile_kernel(const TileParams P) {
    constexpr int K = tapinfo<INTERP>::K;
    constexpr int TX = WAVES * CPW;
    constexpr int WPW = MB / WAVES;           // windows staged per wave
    constexpr int CH = W / 64;                // 64-sample chunks per window
    static_assert(MB % WAVES == 0 && W % 64 == 0, "staging split");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t M = (uint32_t)P.M, N = (uint32_t)P.N;
    const long T = (long)P.T;
    int   *Abase = (int *)smem;                       // [M]
    float *Aext  = (float *)(Abase + M);              // [M]
    int   *Bbase = (int *)(Aext + M);                 // [N]
    float *Bext  = (float *)(Bbase + N);              // [N]
    const uint32_t hdr = ((M + N) * 8 + 15) & ~15u;
    ST *win = (ST *)(smem + hdr);                     // [2][MB][W]
    float *part = (float *)(smem + hdr);              // prologue scratch, aliases the windows

    // ---- which tile (XCD-aware: consecutive tile ids -> same XCD, dispatch is round-robin mod 8)
    const uint32_t nb = gridDim.x;
    uint32_t bid = blockIdx.x;
    {
        const uint32_t q = nb / 8, r = nb % 8, xcd = bid % 8, k = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;   // bijective remap
    }
    const uint32_t tz = bid % P.tiles_z, txi = P.tile_x0 + bid / P.tiles_z;
    const uint32_t tile_id = tz + P.tiles_z * txi;

    // ---- my pixels: lane -> depth, (wave, c) -> column.  Out-of-image lanes are clamped onto a
    //      real pixel (keeps them inside the tile's delay window) and masked at the store.
    const uint64_t ncols = P.I2 * P.I3, i_end = P.i_begin + P.i_count;
    const uint64_t i1 = (uint64_t)tz * TZ + lane;
    const uint64_t i1c = i1 < P.I1 ? i1 : P.I1 - 1;
    double px[CPW], py[CPW], pz[CPW];
    bool ok[CPW];
    uint64_t ipix[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const uint64_t col = (uint64_t)txi * TX + wave * CPW + c;
        const uint64_t colc = col < ncols ? col : ncols - 1;
        const uint64_t i = i1c + P.I1 * colc;
        const uint64_t ig = i1 + P.I1 * col;
        ok[c] = (i1 < P.I1) && (col < ncols) && (ig >= P.i_begin) && (ig < i_end);
        ipix[c] = ig;
        px[c] = P.Pi[3 * i]; py[c] = P.Pi[3 * i + 1]; pz[c] = P.Pi[3 * i + 2];
    }
    const double cf = P.cinv_fs, fs = P.fs;
    const bool VS = P.VS, DV = P.DV;

    auto a_of = [&](int c, uint32_t m) -> double {       // (tau_tx*fs - t0*fs + OFF), reference src/bf.cu:104-108,114
        const double rx = px[c] - (double)P.Pv[4 * m], ry = py[c] - (double)P.Pv[4 * m + 1], rz = pz[c] - (double)P.Pv[4 * m + 2];
        const double dot = rx * (double)P.Nv[3 * m] + ry * (double)P.Nv[3 * m + 1] + rz * (double)P.Nv[3 * m + 2];
        double dv = dot;
        if (VS) { const double len = sqrt(rx * rx + ry * ry + rz * rz); dv = DV ? len : copysign(len, dot); }
        return dv * cf - (double)P.Pv[4 * m + 3] * fs + tapinfo<INTERP>::OFF;
    };
    auto b_of = [&](int c, uint32_t n) -> double {       // tau_rx*fs, reference src/bf.cu:110
        const double rx = px[c] - (double)P.Pr[3 * n], ry = py[c] - (double)P.Pr[3 * n + 1], rz = pz[c] - (double)P.Pr[3 * n + 2];
        return sqrt(rx * rx + ry * ry + rz * rz) * cf;
    };

    // ---- prologue: tile-wide window bases / extents per transmit and per receiver
    const uint32_t MX = M > N ? M : N;
    for (uint32_t m = 0; m < M; ++m) {
        float mn = INFINITY, mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < CPW; ++c) { const float a = (float)a_of(c, m); mn = fminf(mn, a); mx = fmaxf(mx, a); if (!(a == a)) mx = INFINITY; }
        mn = wave_min(mn); mx = wave_max(mx);
        if (lane == 0) { part[wave * MX + m] = mn; part[(WAVES + wave) * MX + m] = mx; }
    }
    __syncthreads();
    for (uint32_t m = tid; m < M; m += THREADS) {
        float mn = part[m], mx = part[WAVES * MX + m];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) { mn = fminf(mn, part[w * MX + m]); mx = fmaxf(mx, part[(WAVES + w) * MX + m]); }
        const float fl = floorf(mn) - 1.0f;              // margin: (float)a may have rounded up
        Abase[m] = (fabsf(fl) < 1.0e9f) ? (int)fl : 0;
        Aext[m] = (fabsf(fl) < 1.0e9f) ? (mx - fl) + 0.01f : INFINITY;
    }
    __syncthreads();
    for (uint32_t n = 0; n < N; ++n) {
        float mn = INFINITY, mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < CPW; ++c) { const float b = (float)b_of(c, n); mn = fminf(mn, b); mx = fmaxf(mx, b); if (!(b == b)) mx = INFINITY; }
        mn = wave_min(mn); mx = wave_max(mx);
        if (lane == 0) { part[wave * MX + n] = mn; part[(WAVES + wave) * MX + n] = mx; }
    }
    __syncthreads();
    float emax = 0.f;
    for (uint32_t n = tid; n < N; n += THREADS) {
        float mn = part[n], mx = part[WAVES * MX + n];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) { mn = fminf(mn, part[w * MX + n]); mx = fmaxf(mx, part[(WAVES + w) * MX + n]); }
        const float fl = floorf(mn) - 1.0f;
        const bool fin = fabsf(fl) < 1.0e9f;
        Bbase[n] = fin ? (int)fl : 0;
        const float e = fin ? (mx - fl) + 0.01f : INFINITY;
        Bext[n] = e;
        emax = fmaxf(emax, e);
    }
    float amax = 0.f;
    for (uint32_t m = tid; m < M; m += THREADS) amax = fmaxf(amax, Aext[m]);
    __syncthreads();                                   // part[] is free again
    emax = wave_max(emax); amax = wave_max(amax);
    if (lane == 0) { part[wave] = emax; part[WAVES + wave] = amax; }
    __syncthreads();
    {
        float e = part[0], a = part[WAVES];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) { e = fmaxf(e, part[w]); a = fmaxf(a, part[WAVES + w]); }
        // every lane's last tap must be inside the staged window: floor(tr) + K - 1 <= W - 1
        if (!(a + e + (float)K <= (float)W)) {
            if (tid == 0) {
                const uint32_t slot = atomicAdd(&P.fallback_list[0], 1u);
                if (slot < P.fallback_cap) P.fallback_list[1 + slot] = tile_id;
            }
            return;                                    // uniform exit: generic kernel takes this tile
        }
    }
    __syncthreads();

    // ---- main loop over stages (mb = transmit block, n = receiver; n is the inner index)
    const uint32_t nmb = (M + MB - 1) / MB;
    const uint32_t nstage = nmb * N;
    float accx[CPW], accy[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) { accx[c] = 0.f; accy[c] = 0.f; }
    float ra[CPW][MB];
    const ST *__restrict__ xg = (const ST *)P.x;
    ST stg[WPW][CH];                                   // staging registers (global -> LDS)

    auto stage_load = [&](uint32_t st) {               // issue the global loads of stage st
        const uint32_t n = st % N, m0 = (st / N) * MB;
        const int bn = Bbase[n];
#pragma unroll
        for (int r = 0; r < WPW; ++r) {
            const uint32_t j = wave + WAVES * r, m = m0 + j;
            const bool mok = m < M;
            const long ws = (long)(mok ? Abase[m] : 0) + bn;
            const long base = (long)n * (long)P.strN + (long)(mok ? m : 0) * (long)P.strM;
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const long s = ws + q * 64 + lane;
                stg[r][q] = (mok && s >= 0 && s < T) ? xg[base + s] : zero_of(xg);
            }
        }
    };
    auto stage_store = [&](int buf) {                  // registers -> LDS window buffer
#pragma unroll
        for (int r = 0; r < WPW; ++r) {
            const uint32_t j = wave + WAVES * r;
#pragma unroll
            for (int q = 0; q < CH; ++q) win[(buf * MB + j) * W + q * 64 + lane] = stg[r][q];
        }
    };

    stage_load(0);
    stage_store(0);
    __syncthreads();

    for (uint32_t st = 0; st < nstage; ++st) {
        const uint32_t n = st % N, m0 = (st / N) * MB;
        const int buf = st & 1;
        if (st + 1 < nstage) stage_load(st + 1);       // in flight during the compute below

        if (n == 0) {                                  // new transmit block: refresh the tx residuals
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                const uint32_t m = m0 + j < M ? m0 + j : M - 1;
                const double A = (double)Abase[m];
#pragma unroll
                for (int c = 0; c < CPW; ++c) ra[c][j] = (float)(a_of(c, m) - A);
            }
        }
        const int bn = Bbase[n];
        const float en = Bext[n];
        float rb[CPW];
#pragma unroll
        for (int c = 0; c < CPW; ++c) rb[c] = (float)(b_of(c, n) - (double)bn);

        // is every window of this stage strictly inside the record?  (uniform)
        bool interior = true;
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const uint32_t m = m0 + j;
            if (m < M) {
                const long ws = (long)Abase[m] + bn;
                interior = interior && (ws >= 1) && ((float)ws + Aext[m] + en + (float)K < (float)T);
            }
        }
        const ST *wb = win + (size_t)buf * MB * W;

#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const uint32_t m = m0 + j;
            if (m >= M) break;
            float wr = 1.f, wi = 0.f;
            if constexpr (WTAB) {
                const float2 wt = ((const float2 *)P.wtab)[n + (size_t)N * m];
                wr = wt.x; wi = wt.y;
                if (wr == 0.f && wi == 0.f) continue;   // zero weight: skip (reference src/bf.cu:122,126)
            }
            const long ws = (long)Abase[m] + bn;
            const float lo = tapinfo<INTERP>::LO - (float)ws;     // validity bounds in window-relative units
            const float hi = (float)(T - K + 1 - ws);
            float phc = 0.f, fcyc = 0.f;
            if constexpr (FMOD) {                       // phase (cycles) = fmod*tau, tau = (tr + ws - OFF)/fs
                fcyc = (float)(P.fmod / fs);
                const double p0 = ((double)ws - tapinfo<INTERP>::OFF) * (P.fmod / fs);
                phc = (float)(p0 - floor(p0));
            }
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                const float tr = ra[c][j] + rb[c];
                const uint32_t idx = (uint32_t)tr;       // tr >= ~2 by construction
                const ST *tp = wb + j * W + idx;
                float vx, vy;
                if constexpr (K == 1) {
                    const cplx<float> s0 = lds_ld(tp);
                    vx = s0.x; vy = s0.y;
                } else {
                    float w[4];
                    tile_weights<INTERP>(tr - (float)idx, w);
                    vx = 0.f; vy = 0.f;
#pragma unroll
                    for (int k = 0; k < K; ++k) { const cplx<float> s = lds_ld(tp + k); vx = fmaf(w[k], s.x, vx); vy = fmaf(w[k], s.y, vy); }
                }
                if (!interior) { const bool v = (tr >= lo) && (tr < hi); vx = v ? vx : 0.f; vy = v ? vy : 0.f; }
                if constexpr (FMOD) {                   // reference src/bf.cu:117
                    const float ph = fmaf(tr, fcyc, phc);
                    const float cs = __builtin_amdgcn_cosf(ph), sn = __builtin_amdgcn_sinf(ph);
                    const float tx = vx * cs - vy * sn; vy = vx * sn + vy * cs; vx = tx;
                }
                if constexpr (WTAB) {
                    accx[c] = fmaf(wr, vx, fmaf(-wi, vy, accx[c]));
                    accy[c] = fmaf(wr, vy, fmaf(wi, vx, accy[c]));
                } else { accx[c] += vx; accy[c] += vy; }
            }
        }

        if (st + 1 < nstage) stage_store(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: y[i] = pix  (reference src/bf.cu:140); lanes = consecutive i -> coalesced
#pragma unroll
    for (int c = 0; c < CPW; ++c)
        if (ok[c]) st((ST *)P.y, (size_t)(ipix[c] - P.i_begin), cplx<float>{accx[c], accy[c]});
}

// ------------------------------------------------------------------------------------------
constexpr int CFG_CPW = 2, CFG_MB = 16, CFG_W = 192;

TileConfig tile_config(int dtype, int /*interp*/) {
    TileConfig c;
    c.tile_cols = WAVES * CFG_CPW;
    c.mb = CFG_MB;
    c.window = CFG_W;
    c.threads = THREADS;
    c.lds_bytes = (size_t)2 * CFG_MB * CFG_W * (dtype == 2 ? 4 : 8);
    return c;
}

template <int INTERP, typename ST>
static hipError_t launch_tile_i(const TileParams &P, unsigned ntiles, size_t lds, hipStream_t s) {
    const bool fm = P.fmod != 0.0, wt = P.wtab != nullptr;
    const dim3 g(ntiles), b(THREADS);
#define QDAS_LAUNCH(FM, WT)                                                                              \
    do {                                                                                                 \
        auto kfn = das_tile_kernel<INTERP, ST, FM, WT, CFG_CPW, CFG_MB, CFG_W>;                          \
        hipError_t e = hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                   \
        kfn<<<g, b, lds, s>>>(P);                                                                        \
    } while (0)
    if (fm && wt) QDAS_LAUNCH(true, true);
    else if (fm)  QDAS_LAUNCH(true, false);
    else if (wt)  QDAS_LAUNCH(false, true);
    else          QDAS_LAUNCH(false, false);
#undef QDAS_LAUNCH
    return hipGetLastError();
}

hipError_t launch_tile(const TileParams &P, int dtype, unsigned ntiles, hipStream_t s) {
    if (ntiles == 0) return hipSuccess;
    const TileConfig c = tile_config(dtype, P.flag & 7);
    const size_t MX = P.M > P.N ? P.M : P.N;
    const size_t hdr = (((P.M + P.N) * 8) + 15) & ~(size_t)15;
    size_t body = c.lds_bytes;
    if (body < 2 * WAVES * MX * 4) body = 2 * WAVES * MX * 4;   // prologue scratch aliases the windows
    const size_t lds = hdr + body;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const int interp = P.flag & 7;
#define QDAS_DT(I)                                                                       \
    (dtype == 2 ? launch_tile_i<I, uint32_t>(P, ntiles, lds, s) : launch_tile_i<I, float2>(P, ntiles, lds, s))
    switch (interp) {
        case 0: return QDAS_DT(0);
        case 1: case 4: return QDAS_DT(1);
        case 2: return QDAS_DT(2);
        case 3: return QDAS_DT(3);
        case 5: return QDAS_DT(5);
    }
#undef QDAS_DT
    return hipErrorInvalidValue;
}

}  // namespace qdas
