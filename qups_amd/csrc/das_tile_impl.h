#pragma once
// das_tile_impl.h -- the fused, LDS-staged delay-and-sum kernel for gfx950 (MI355X).
//
// Replaces the reference launch of `DASf` / `DASh` (reference src/bf.cu:153-171, body src/bf.cu:49-142) for the bulk of the
// work: 'DAS' (sum over both apertures) with fp32 / fp16 / fp64 data, 'SYN' / 'MUL' / 'BF' (kept apertures) with fp32 data; scalar
// sound speed or a per-pixel map; pixel-independent apodization (folded by the host into one N x M table) plus one pixel x
// stage-element array (pixel x receiver; pixel x transmit with the roles swapped; pixel only) or generated rule.  Everything
// else is served by das_generic.hip.  Instantiated per launch configuration in
// das_tile_{f32,sym,f16,...}.hip (dispatch in das_tile.hip) and, with the plan's sizes as constants, by hiprtc (jit.hip).
//
// Design (MI355X-first, not a re-tiling of the reference's one-thread-per-pixel loop):
//
//  * A workgroup (16 waves) owns a TILE of 1024 pixels, 2^t (fast image axis I1 = depth) x 1024/2^t columns; a wave covers
//    2^w x 64/2^w of them.  The plan probes the tile footprint (largest that fits the LDS window) and picks the wave footprint
//    from an LDS bank model (8 x 8 pixels on a lambda/4 grid: the 32 lanes of an access group read <= 32 consecutive samples).
//    Two consecutive transmits (m, m+1) of the same pixel ride in the two halves of packed-fp32 (v_pk_*_f32) instructions.
//  * Time of flight is separable: tau*fs + off = a(i,m) + b(i,n).  The prologue (tile_prologue.h) computes tile-wide integer
//    window bases A[m] <= a, B[n] <= b and extents; afterwards each lane only carries the small fp32 residuals of the fp64
//    delays, ra = a - A[m] - 1/2, rb = b - B[n] (exact to ~1e-5 sample; the reference's fp32 tau carries ~1e-4 sample at
//    tau*fs ~ 2000).  Per pair (tile_pairs.h):
//        t = ra[m] + rb;  k = rint(t) (magic-number add);  s = t - k  in [-1/2, 1/2];
//        first tap = window[k], weights = even/odd polynomials in s.
//    "Block" elements (the MB transmits of a stage) and the "stage" element (its receiver) each have a delay kind {distance,
//    signed distance, plane wave}; for 'MUL' -- and for a full sum with few transmits, or with a pixel x transmit weight -- the
//    host swaps the two apertures' roles (stage element = transmit, block = 32 receivers).
//  * For every STAGE (receiver n, block of MB transmits) the workgroup stages MB fast-time WINDOWS (W samples starting at
//    A[m]+B[n]) of the channel data into LDS with LDS-DMA (tile_staging.h: buffer_load_dwordx4 ... lds: no VGPR round trip, no
//    ds_write; out-of-buffer lanes deliver 0), coalesced along fast time and double-buffered against the compute of the previous
//    stage.  Taps are gathered with ds_read_b64 / ds_read_b32 from inline asm with immediate (window, tap) offsets (tile_taps.h).
//  * A stage's SECOND window set holds, depending on the mode: the mirror traces x[:,m,n] (reciprocal mode, SYM: Pv == Pr, so
//    tau(n,m) == tau(m,n) and index + weights serve both traces of an unordered pair), or the same traces of the NEXT FRAME
//    (FB2 / FB4: index + weights serve two / four frames).
//  * All resident workgroups walk the traces in the same order (columns-fastest tile order, XCD-aware remap), so the channel
//    data is served by L2 to all but the first of the tiles of a depth band.
//  * Tiles whose windows all lie inside [0, T) run a branch-free loop; tiles that touch the ends of the record run the checked
//    loop (edge rule of SURVEY.md section 8 a5).  A tile whose delay spread does not fit W appends itself to a fallback list
//    and is processed by the generic kernel afterwards -- results never depend on the geometry being "image like".
//  * A pixel x receiver weight (an I1 x I2 x I3 x N array, or a rule evaluated from the geometry: qdas.h QDAS_RXAPOD_*) does
//    not depend on the transmit: it multiplies the stage's partial sum once per (pixel, receiver), loaded one stage ahead; the tile's
//    STAGE LIST holds only the receivers that carry weight somewhere in the tile (plan_stages), and a wave whose 64 weights are all
//    zero skips the stage's gathers altogether.
//  * Pixel-independent weights: the stage's MB table entries go through LDS, fetched one stage ahead by one wave (TileCfg::WST).
//  * fp64 data: geometry, delays, weights and sums in double, one transmit at a time (tile_pairs.h pairs_f64).
//  * 'SYN' / 'MUL': a stage belongs to one plane of the output; its sum is added with non-returning fp32 atomics.
//  * Few tiles (pixel slab of a multi-GPU job, small image): ksplit workgroups per tile, each a slice of the aperture,
//    partial images reduced in a fixed order.
//  * Cold fp64 code (transmit-block refresh, generated apodization) is called, not inlined (tile_util.h): no instantiation
//    uses scratch memory (tests/test_build_regs.py).
#include "tile_params.h"
#ifndef QDAS_DBG_SYNC
#define QDAS_DBG_SYNC 0           // race hunting (hiprtc builds: QDAS_JIT_DEFINES=QDAS_DBG_SYNC=<bits>): extra waits / barriers in the stage loop, see run()
#endif
#ifndef QDAS_ONEACC_PLAIN
#define QDAS_ONEACC_PLAIN 0      // tuning builds: the plain (not software-pipelined) pair loop for the long-stage two-window-set fp32 builds
#endif
#ifndef QDAS_F16_PIPE
#define QDAS_F16_PIPE 0          // tuning builds (QDAS_JIT_DEFINES): the software-pipelined pair loop for fp16 two-window-set kernels too
#endif
#include "tile_util.h"
#include "tile_taps.h"
#include "tile_hooks.h"
#include "das_tile_cfg.h"

namespace qdas {

typedef __attribute__((address_space(3))) void lds_void;

// Compile-time configuration of one instantiation.
//   WAVES waves per workgroup, MB transmits per stage, W samples per window, NBUF window buffers (NBUF-1 stages of LDS-DMA in
//   flight).  FB2 / FB4: two / four FRAMES per launch (x + f*x_fstride -> y + f*y_fstride): the other window sets of a stage hold
//   the same traces of the next frames, so tap index and weights -- which depend on the geometry only -- are computed once for all
//   of them (the reference launches one kernel per frame, kern/das_spec.m:371).  BIG: re-base the DMA descriptors along the
//   receiver walk (transposed fp32 frames beyond 2 GiB).  LUT: the delays come from host-supplied tables (qdas_das_lut).
//   FOLD: the channel data are RECIPROCITY-FOLDED (fold.hip; reciprocal plans, fp32): xs[:,n,m] = w[n,m] x[:,n,m] + w[m,n] x[:,m,n] for n < m,
//   xs[:,n,n] = w[n,n] x[:,n,n] -- tau(n,m) == tau(m,n), and interpolation is linear in the data, so the two traces of an unordered pair are
//   added ONCE per frame (one streaming pass over HBM) instead of being gathered and weighted separately for every pixel: the stage loop
//   walks the upper triangle n <= m only, with ONE window set (FOLD && MIRQ: two -- my pixel's trace (n, m) and, for its lateral mirror image,
//   (N-1-m, N-1-n), which lies in the upper triangle too).  Pixel-independent weights ride in the fold pass: such kernels carry no table.
template <int INTERP_, typename ST_, bool FMOD_, bool WTAB_, bool SYM_, bool FB2_, bool FB4_, int WAVES_, int MB_, int W_, int NBUF_, bool BIG_, bool LUT_, bool BF_ = false, bool MIRQ_ = false, bool FOLD_ = false>
struct TileCfg {
    static constexpr int INTERP = INTERP_, WAVES = WAVES_, MB = MB_, W = W_, NBUF = NBUF_;
    using ST = ST_;
    static constexpr bool FMOD = FMOD_, WTAB = WTAB_, SYM = SYM_, FB2 = FB2_, FB4 = FB4_, BIG = BIG_, LUT = LUT_;
    static constexpr bool BF = BF_;                  // keep both aperture dimensions: every pair's sample goes to its own output plane
    static constexpr bool FBX = FB2 || FB4;          // more than one frame per launch
    // reciprocal mode AND lateral-mirror mode (tile_params.h `mir`): tau(n,m) == tau(m,n) == tau'(N-1-n,N-1-m) == tau'(N-1-m,N-1-n) -- four
    // window sets per stage: {x[:,n,m], x[:,m,n]} for my pixel, {x[:,N-1-n,N-1-m], x[:,N-1-m,N-1-n]} for its mirror image, ONE tap index
    // and ONE set of weights for all four
    static constexpr bool MIRQ = MIRQ_;
    static constexpr bool FOLD = FOLD_;              // reciprocity-folded data (above): upper triangle only, no reciprocal window set
    static constexpr bool FOLDQ = FOLD_ && MIRQ_;    // ... in lateral-mirror mode: window set 0 = my pixel's trace, set 1 = its mirror image's
    // (folded data, TWO FRAMES per launch: FOLD && FB2 -- window sets {frame 0, frame 1}; with the mirror mode {f0 mine, f0 image, f1 mine, f1 image}:
    //  one tap index and one set of weights for two folded traces of two frames, i.e. for EIGHT products of the reference's loop)
    static constexpr bool QUAD = FB4 || (MIRQ && !FOLD) || (FOLDQ && FB2);   // four window sets: the pair loop makes two passes over one index / weight evaluation
    static constexpr bool TWO = (SYM && !FOLD) || FBX || FOLDQ;   // (at least) two window sets per stage: direct + (reciprocal | next frame | mirror image)
    // long-stage two-window-set fp32 builds (hiprtc: 32 transmits x 2 sets, plan_stage_shape): ONE accumulator per pixel (my pixel, its mirror image / the next frame), as
    // MIRQ and FOLDQ run -- the second accumulator of each is the register pair the 16 residual pairs of such a stage leave no room for (round 5's builds kept one residual
    // pair in scratch and re-loaded it in every stage behind an s_waitcnt vmcnt(0) that also drained the LDS-DMA: profiles/r05/kernel_regs_hiprtc.txt)
    // (likewise the one general-mode variant that is two registers short: fp32, lanczos3 weights, remodulation AND a weight table -- built on demand, found by
    //  tests/test_jit.py::test_bench_and_suite_hiprtc_kernels_do_not_spill in round 6)
    static constexpr bool ONEACC = (FB2 && !SYM && sizeof(ST_) == 8 && MB_ >= 32) || (!TWO && sizeof(ST_) == 8 && FMOD_ && WTAB_ && INTERP_ == 3 && !BF_ && !LUT_ && !BIG_);
    static constexpr int NHP = QUAD ? 2 : 1;         // passes of the pair loop: one per frame pair (MIRQ: my pixel, its mirror image)
    static constexpr int NFR = FB4 ? 4 : (FB2 ? 2 : 1);   // frames per launch
    static constexpr int NW = QUAD ? 4 * MB : (TWO ? 2 * MB : MB);     // windows per LDS buffer
    static constexpr int K = tapinfo<INTERP>::K;
    static constexpr int THREADS = WAVES * 64;
    static constexpr int WPW = FB4 ? 1 : MB / WAVES; // windows staged per wave and window set
    static constexpr int SB = (int)sizeof(ST);       // bytes per complex sample
    static constexpr bool F32 = (SB == 8);
    static constexpr bool F64 = (SB == 16);          // double data: geometry, delays, weights and sums in fp64 (tile_pairs.h pairs_f64)
    // instantiations that may run a pixel x receiver weight: their stage list holds the ACTIVE receivers only
#ifdef QDAS_JIT
    static constexpr bool JITB = true;               // plan-specialised build: constants folded, registers to spare
#else
    static constexpr bool JITB = false;
#endif
#ifdef QDAS_NO_ACT
    static constexpr bool ACT = false;               // (A/B builds)
#else
    static constexpr bool ACT = !F64 && !SYM && !BIG && !BF_ && !(FB2 && F32 && !JITB) && !FB4;     // (fp32 frames sharing a launch keep the plain list: no register for it)
    // a pixel x receiver weight in lateral-mirror mode: the mirror image of a pixel has its OWN weight (fp16 two-window-set kernels)
    static constexpr bool WMIR = ACT && FB2;
    static constexpr bool W64 = F64 && !FMOD_;       // fp64 data: a pixel x receiver (or pixel-only) weight ARRAY, plain list of stages (the remodulation variants have no registers for it)
#endif
    // pixel-independent weights (N x M table) reach the pair loop through LDS: the 32 (reciprocal mode: 2 x 32) table entries of a stage are
    // fetched one stage ahead by a single wave and read back with broadcast ds_reads -- as scalar loads inside the pair loop they cost a
    // scalar-memory latency per transmit pair and drained the LDS queue with it (lgkmcnt counts both)
    static constexpr bool WST = WTAB_ && !F64 && !BF_ && !FB4;      // (four frames per launch: no registers to spare -- scalar loads as before)
    static constexpr int WSTB = WST ? NBUF_ * (2 * MB_ * 8 + 16) : 0;     // [NBUF][{direct, mirror}][MB] float2 + {non-zero masks} per buffer
    // a weight per (pixel, BLOCK element) on top of the stage weight: 16 more registers -- the 16-transmit wide-window configuration has them
    static constexpr bool BPIX = F32 && !SYM_ && !FB2_ && !FB4_ && !BF_ && !LUT_ && !BIG_ && !WTAB_ && !FMOD_ && MB_ == 16 && W_ == 384;
    using GT = std::conditional_t<F64, double, float>;   // type of the geometry tables (the reference casts them to the data precision, kern/das_spec.m:244)
    static constexpr int WB = W * SB;                // bytes per window
    static constexpr int PB = 1024;                  // bytes per full DMA piece (one wave-instruction x 16 B)
    static constexpr int PCS = (WB + PB - 1) / PB;   // pieces per window; the last one may use fewer lanes
    static constexpr int NDMA = WPW * PCS * (QUAD ? (FB4 ? 2 : 4) : TWO ? 2 : 1);    // DMA instructions per wave and stage
    static_assert(!(SYM && FBX && !FOLD_) && !(FB2 && FB4) && !(FOLD_ && FB4), "reciprocal mode runs one frame per launch (folded data: one or two)");
    static_assert(!MIRQ || FOLD || (SYM && MB_ == WAVES_ && 4 * MB_ * W_ * (int)sizeof(ST_) <= 65536), "reciprocal + lateral-mirror mode: one window per wave and set, immediate LDS offsets");
    static_assert(!FOLD || (SYM && !WTAB_ && !BIG && !LUT && !BF_ && sizeof(ST_) == 8), "folded data: reciprocal plans, fp32, weights folded into the data");
    static_assert(!FOLDQ || (MB_ % WAVES_ == 0 && (FB2 ? 4 : 2) * MB_ * W_ * (int)sizeof(ST_) <= 65536), "folded data + lateral-mirror mode: two (two frames: four) window sets within the immediate LDS offsets");
    static_assert(!BIG || (!SYM && !FBX), "the re-basing general kernel runs one frame per launch");
    static_assert(!(FOLDQ && FB2) || MB_ == WAVES_, "folded data + mirror mode, two frames: one window per wave and set");
    static_assert(!LUT || (!SYM && !FB4 && !BIG && (!FB2 || JITB)), "table-driven delays: general mode, one frame per launch (hiprtc builds: the lateral-mirror mode's two window sets)");
    static_assert(!BF || (!SYM && !FBX && !BIG && !LUT && sizeof(ST_) == 8), "'BF': general mode, fp32 data, one frame per launch");
    static_assert(!F64 || (!SYM && !FBX && !BIG && !LUT && !BF), "fp64 data: the 'DAS' sum (optionally remodulated / with a weight table), one frame per launch");
    static_assert(FB4 ? (2 * MB == WAVES) : (MB % WAVES == 0 && MB % 2 == 0), "staging split");
    static_assert(WB % 16 == 0, "window must be a whole number of 16-byte lanes");
};

// One workgroup's view of its tile: uniform state (LDS carve-up, tile index, aperture share), the lane's pixel, and the running
// state of the stage loop.  The member functions are defined in tile_prologue.h / tile_staging.h / tile_pairs.h / below; all
// are inlined into the kernel, the object never leaves registers.
template <class C> struct Tile {
    using ST = typename C::ST;
    const TileParams &P;
    hooks::Timer timer;
    // ---- uniform
    int tid, lane, wave;
    uint32_t M, N;                                   // block / stage elements ("transmits" / "receivers")
    int T;
    uint64_t strN, strM;                             // trace strides in samples
    int kindB, kindS;
    using GT = typename C::GT;
    struct rec64 { double x, y, z; int b, pad; };     // fp64 twin of the receiver record {window base B, position}
    int *Abase; float *Aext, *Bext; float4 *nrec; rec64 *nrec64; GT *PvL, *NvL; ST *win; float *part; uint32_t win_off;
    unsigned char *wst; uint32_t wst_off;            // stage weights (TileCfg::WST), and their LDS byte address
    uint2 *act;                                      // [N + 1] {receiver, its window base B} of the receivers with a non-zero weight somewhere in the tile, then {count, -} (pixel x receiver weights)
    uint32_t split, S, tile_id;
    uint32_t Sm, sm;                   // transmit-block groups of a two-dimensional split (tile_params.h ksplit_m) and mine; 1, 0 otherwise
    double fs, symC; int symCi;
    bool tile_interior;
    int need_b;                        // bytes of a staged window that some lane of the tile can touch (<= WB; tile_prologue.h): the LDS-DMA moves no more
    // ---- this lane's pixel
    static constexpr uint32_t NOT_MINE = 0xffffffffu;
    uint32_t pofs;                                   // my pixel's offset in this plan's slab, or NOT_MINE (lane outside the image / the slab; slabs stay below 2^32 - 1 pixels)
    __device__ __forceinline__ bool in_shard() const { return pofs != NOT_MINE; }
    // the lane id straight from the hardware (two instructions), for the per-stage code (LDS-DMA, stage weights): as a variable it -- and
    // lane * 16 -- would be two registers carried through every pair loop, which the register-bound instantiations do not have
    static __device__ __forceinline__ uint32_t lane_now() {
        uint32_t l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    }
    uint64_t ipx;                                    // (clamped) linear pixel index: row of per-pixel arrays / delay tables
    uint64_t ipx2;                                   // lateral-mirror mode with a pixel x receiver weight array: the same of my pixel's mirror image (TileCfg::WMIR)
    GT px, py, pz;
    double cf;                                       // samples per metre (scalar sound speed or this pixel's entry of the map)
    // ---- stage loop
    uint32_t n_lo, n_hi, nstage;
    uint32_t nact; bool use_act;                     // stages per transmit block / whether they come from act[] (else: receivers n_lo ... in order)
    v2f acc, acc1, acc2, acc3;                       // independent partial sums: no back-to-back dependent packed FMAs
    v2f ra[C::MB / 2];                               // block residuals a - A[m] - 1/2 of transmits (2p, 2p+1), packed
    v2f bw[C::BPIX ? C::MB / 2 : 1]; bool bp;        // pixel x block-element weights of the same transmits (TileParams::bpix), and whether the plan has them
    double rad[C::F64 ? C::MB : 1];                  // fp64 data: the same residuals, one double per transmit
    double dacc[4];                                  // fp64 data: two independent complex partial sums {re, im, re, im}
    v2f tot[C::NFR];                                 // weighted totals per frame when a pixel x receiver weight is applied
    double dtot[2], w64[2];                          // fp64 data: the same in double (weighted total, this stage's weight); instantiations without remodulation only
    bool wpix, syn;
    // ---- LDS-DMA staging (tile_staging.h)
    int wb[C::WPW], wb2[C::WPW];
    int qo[C::MIRQ ? (C::FOLD ? 2 * C::WPW : 4) : 1];   // MIRQ: absolute byte offsets of this wave's window(s) in the four (folded data: two) traces of the stage at the DMA front (without B[n])
    uint32_t soff, soff2;
    __amdgpu_buffer_rsrc_t rsD, rsM;
    uint64_t offD, offM, xbytes;
    int fa, fb;

    __device__ __forceinline__ Tile(const TileParams &p) : P(p) {}

    __device__ __forceinline__ void setup(unsigned char *smem);          // LDS carve-up, tile / pixel of this lane
    __device__ __forceinline__ uint32_t locate(uint32_t &i1, uint32_t &col, bool first, bool image = false);
    template <bool PROBE> __device__ __forceinline__ bool prologue();    // window bases + fit verdict           (tile_prologue.h)
    __device__ __forceinline__ void plan_stages();                       // this workgroup's share of the aperture
    template <bool CHECK> __device__ __forceinline__ void run();         // the stage loop
    __device__ __forceinline__ void epilogue();

    // aperture share
    __device__ __forceinline__ uint32_t blk(uint32_t r) const {          // first transmit of my r-th block (>= M: exhausted)
        if constexpr (C::SYM) return (r * S + ((r & 1u) ? S - 1u - split : split)) * C::MB;
        else if constexpr (C::ACT) return (r * Sm + sm) * C::MB;
        else return r * C::MB;
    }
    __device__ __forceinline__ uint32_t nlim(uint32_t m0) const { return C::SYM ? (m0 + C::MB < N ? m0 + C::MB : N) : n_hi; }

    // delays (fp64)
    // fp32 seed (the raw v_sqrt_f32, 1 ulp: the IEEE-exact sqrtf expands to ~20 instructions) + one Newton step in fp64 (rel. error ~1e-14)
    static __device__ __forceinline__ double dsqrt(double d2) {
        const float s0 = __builtin_amdgcn_sqrtf((float)d2);
        const double sd = (double)s0;
        const double r = __builtin_fma(-sd, sd, d2);
        return __builtin_fma(r, (double)(0.5f * __builtin_amdgcn_rcpf(fmaxf(s0, 1.0e-30f))), sd);
    }
    __device__ __forceinline__ double a_of(uint32_t m, const GT *gPv, const GT *gNv) const;
    __device__ __forceinline__ double b_at(GT ex, GT ey, GT ez) const;
    __device__ __forceinline__ double s_at(uint32_t n, GT ex, GT ey, GT ez) const;
    __device__ __forceinline__ bool on_side(uint32_t n, GT ex, GT ey, GT ez) const;
    // geometry tables in the plan's real type (TileParams carries them as float pointers)
    __device__ __forceinline__ const GT *geo_Pi() const { return (const GT *)P.Pi; }
    __device__ __forceinline__ const GT *geo_Pr() const { return (const GT *)P.Pr; }
    __device__ __forceinline__ const GT *geo_Pv() const { return (const GT *)P.Pv; }
    __device__ __forceinline__ const GT *geo_Nv() const { return (const GT *)P.Nv; }
    struct wraw { uint32_t a, b; };                                      // pixel x receiver weight of a stage element as loaded (raw bits)
    __device__ __forceinline__ wraw wload_raw(uint32_t n, bool image = false) const;
    __device__ __forceinline__ v2f wconv(wraw r) const;
    __device__ __forceinline__ v2f wload(uint32_t n) const { return wconv(wload_raw(n)); }

    // staging (tile_staging.h)
    __device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rs(uint64_t o, uint64_t extra) const;
    __device__ __forceinline__ int wjr(int r) const;
    __device__ __forceinline__ void dma_block(uint32_t m0);
    __device__ __forceinline__ void stage_dma(int bn, int buf);

    // pair loops (tile_pairs.h)
    template <bool CHECK, bool TAILV, bool WZ = true> __device__ __forceinline__ void pairs_plain(uint32_t n, uint32_t m0, int bn, float rb, uint32_t cbase, float phB, uint32_t wbase, uint64_t wmask, uint64_t xmask);
    __device__ __forceinline__ void pairs_pipelined(float rb, uint32_t cbase);
    template <bool CHECK, bool TAILV> __device__ __forceinline__ void pairs_f64(uint32_t n, uint32_t m0, int bn, double rb, uint32_t cbase);
    __device__ __forceinline__ void frame_sums(v2f (&Sf)[4]) const {
        if constexpr (C::FOLDQ && C::FB2) { Sf[0] = acc; Sf[1] = acc1; Sf[2] = acc2; Sf[3] = acc3; }   // {frame 0: my pixel, its image; frame 1: my pixel, its image}
        else if constexpr (C::MIRQ) { Sf[0] = acc + acc1; Sf[1] = acc2 + acc3; }       // my pixel, its mirror image
        else if constexpr (C::FB4) { Sf[0] = acc; Sf[1] = acc1; Sf[2] = acc2; Sf[3] = acc3; }
        else if constexpr (C::FB2) { Sf[0] = acc + acc1; Sf[1] = acc2 + acc3; }
        else Sf[0] = (acc + acc1) + (acc2 + acc3);
    }
};

}  // namespace qdas

#include "tile_prologue.h"
#include "tile_staging.h"
#include "tile_pairs.h"

namespace qdas {

// ------------------------------------------------------------------------------------------------- which tile, which pixel
// (row, column) of this lane's pixel and its offset in the plan's slab.  Called by setup() and -- in the register-tight reciprocal
// kernels -- AGAIN by the epilogue: ~25 mostly scalar instructions once per tile, instead of a register held through the stage loop.
// image: the offset of the MIRROR IMAGE of my pixel (lateral-mirror modes; NOT_MINE for the centre column, which is its own image).
template <class C> __device__ __forceinline__ uint32_t Tile<C>::locate(uint32_t &i1, uint32_t &col, bool first, bool image) {
    // ---- which tile (XCD-aware: consecutive tile ids -> same XCD; dispatch is round-robin mod 8)
    const uint32_t nb = gridDim.x;
    uint32_t bid = blockIdx.x;
    {
        const uint32_t q = nb / 8, r = nb % 8, xcd = bid % 8, k = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;   // bijective remap
    }
    // columns fastest: the tiles an XCD runs concurrently sit at the SAME depth band, so they stage (nearly) the same window of
    // every trace and the XCD's L2 serves all but the first of them.  split = slowest index: the workgroups an XCD runs
    // concurrently work on the SAME slice of the aperture of neighbouring tiles
    const uint32_t ntile = P.tiles_x * P.tiles_z;
    const uint32_t sp = bid / ntile;
    bid -= sp * ntile;
    // DEEPEST band first: with an acceptance-angle / f-number mask the deep tiles have the most active receivers -- the longest
    // jobs start first and the shallow ones fill in behind them (a CU runs one workgroup at a time; images have 2-4 tiles per CU)
#ifdef QDAS_SHALLOW_FIRST
    const uint32_t tz = bid / P.tiles_x, txi = P.tile_x0 + bid % P.tiles_x;
#else
    const uint32_t tz = P.tiles_z - 1u - bid / P.tiles_x, txi = P.tile_x0 + bid % P.tiles_x;
#endif
    if (first) { S = QSPEC(KSPLIT, P.ksplit); split = sp; tile_id = tz + P.tiles_z * txi; }

    // ---- my pixel.  Tile and wave footprints (uniform, chosen by the plan from the scan's delay gradient; qdas_api.hip
    //      choose_tile_shape): the tile is (1 << tzl) pixels of I1 x (1024 >> tzl) columns -- as deep as the LDS window allows;
    //      inside it a wave covers (1 << wzl) x (64 >> wzl) pixels -- as shallow as needed for the 32 lanes of an LDS access group
    //      to read <= 32 consecutive samples (conflict-free), e.g. 8 x 8 when the delay advances 2 samples per pixel of depth.
    //      Out-of-image lanes are clamped onto a real pixel (keeps them inside the tile's delay window) and masked at the store.
    const uint64_t I1 = QSPEC(I1, P.I1);
    uint64_t ncols = P.I2 * P.I3;
    asm volatile("" : "+s"(ncols));                   // opaque: what the epilogue's call derives from it is made THERE, not carried through the stage loop (in a vector register pair)
    const uint64_t i_end = P.i_begin + P.i_count;
    const int tzl = QSPEC(TZL, P.tz_log2), wzl = QSPEC(WZL, P.wz_log2);
    const uint32_t wave_z = (uint32_t)wave & ((1u << (tzl - wzl)) - 1u), wave_c = (uint32_t)wave >> (tzl - wzl);
    // (depth index, column: may lie outside the image -- clamped for the delays, masked at the store)
    if (first) {
        i1 = (tz << tzl) + (wave_z << wzl) + (uint32_t)(lane & ((1 << wzl) - 1));
        col = txi * ((uint32_t)(C::WAVES * 64) >> tzl) + (wave_c << (6 - wzl)) + (uint32_t)(lane >> wzl);
    } else {                                          // (the epilogue's calls: recomputed THERE from an opaque lane id, not carried through the stage loop in registers -- or scratch)
        uint32_t ln = (uint32_t)lane;
        asm volatile("" : "+v"(ln));
        i1 = (tz << tzl) + (wave_z << wzl) + (ln & ((1u << wzl) - 1u));
        col = txi * ((uint32_t)(C::WAVES * 64) >> tzl) + (wave_c << (6 - wzl)) + (ln >> wzl);
    }
    const bool mirror = C::MIRQ || (C::FB2 && QSPEC(MIR, P.mir));
    const uint64_t ncols_mine = mirror ? (ncols + 1) / 2 : ncols;      // lateral-mirror mode: the tiles cover the first half of the columns
    uint64_t ig = (uint64_t)i1 + I1 * (uint64_t)col;
    bool inside = ((uint64_t)i1 < I1) && ((uint64_t)col < ncols_mine) && (ig >= P.i_begin) && (ig < i_end);
    if constexpr (C::MIRQ || C::FB2) {
        if (image) {                                  // (from the epilogue)
            const uint64_t col2 = ncols - 1 - (uint64_t)col;
            inside = inside && mirror && col2 != (uint64_t)col;
            ig = (uint64_t)i1 + I1 * col2;
            // (mirror slab: slab B = the images of slab A's columns lies behind slab A in y)
            if (QSPEC(MIR, P.mir) == 2) ig = ig - (I1 * ncols - i_end) + i_end;
        }
    }
    uint32_t po = inside ? (uint32_t)(ig - P.i_begin) : NOT_MINE;
    asm volatile("" : "+v"(po));                      // opaque: ONE register carries "mine?" and "where" (never re-derived from the 64-bit (row, column) pair)
    return po;
}

template <class C> __device__ __forceinline__ void Tile<C>::setup(unsigned char *smem) {
    tid = threadIdx.x; lane = tid & 63;
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: keeps everything derived from it in SGPRs
    M = QSPEC(M, (uint32_t)P.M); N = QSPEC(N, (uint32_t)P.N);
    T = QSPEC(T, (int)P.T);
    strN = QSPEC(STRN, P.strN); strM = QSPEC(STRM, P.strM);
    kindB = QSPEC(KINDB, P.kindB); kindS = QSPEC(KINDS, P.kindS);
    fs = P.fs;
    Abase = (int *)smem;                              // [M]
    Aext  = (float *)(Abase + M);                     // [M]
    Bext  = Aext + M;                                 // [N]
    nrec = (float4 *)(smem + (((2 * M + N) * 4 + 15) & ~15u));   // [N] per receiver {window base B (int bits), x, y, z}: ONE broadcast read per stage
    nrec64 = (rec64 *)nrec;                           // (fp64 data: 32-byte records)
    constexpr uint32_t RECB = C::F64 ? 32u : 16u, GB = (uint32_t)sizeof(GT);
    PvL   = (GT *)((unsigned char *)nrec + RECB * N);   // [4M] (virtual) sources + t0
    NvL   = PvL + 4 * M;                              // [3M] transmit normals
    const uint32_t actb = P.act_bytes;               // (only plans with a pixel x receiver weight pay for the stage list)
    const uint32_t off_act = ((((2 * M + N) * 4 + 15) & ~15u) + RECB * N + 7 * M * GB + 15) & ~15u;
    const uint32_t off_wst = off_act + ((actb + 15) & ~15u);
    act   = (uint2 *)(smem + off_act);                // [N + 1]
    wst   = smem + off_wst;
    wst_off = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned char *)smem) + off_wst;
    const uint32_t hdr = (off_wst + (uint32_t)C::WSTB + 15) & ~15u;
    win = (ST *)(smem + hdr);                         // [NBUF][NW][W]
    part = (float *)(smem + hdr);                     // prologue scratch, aliases the windows
    win_off = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned char *)smem) + hdr;

    uint32_t i1, col;
    pofs = locate(i1, col, true);
    const uint64_t I1 = QSPEC(I1, P.I1), ncols = (C::MIRQ || (C::FB2 && QSPEC(MIR, P.mir))) ? (P.I2 * P.I3 + 1) / 2 : P.I2 * P.I3;
    ipx = ((uint64_t)i1 < I1 ? (uint64_t)i1 : I1 - 1) + I1 * ((uint64_t)col < ncols ? (uint64_t)col : ncols - 1);
    ipx2 = ipx;
    if constexpr (C::WMIR) { if (QSPEC(MIR, P.mir)) ipx2 = ((uint64_t)i1 < I1 ? (uint64_t)i1 : I1 - 1) + I1 * (P.I2 * P.I3 - 1 - ((uint64_t)col < ncols ? (uint64_t)col : ncols - 1)); }
    cf = P.cinv_fs;
    if constexpr (C::LUT) {                            // delays from host tables (tau_tx: I x M, tau_rx: I x N, in samples; table-driven plans cover [0, I))
        px = py = pz = 0.f;
        const uint64_t i_end = P.i_begin + P.i_count;
        ipx = ipx < i_end ? ipx : i_end - 1;
    } else {
        px = geo_Pi()[3 * ipx]; py = geo_Pi()[3 * ipx + 1]; pz = geo_Pi()[3 * ipx + 2];
        if constexpr (!C::MIRQ) {                      // (reciprocal + lateral-mirror plans have a scalar sound speed: cf stays uniform -- two registers the four window sets need)
            if (QSPEC(HAS_CINV_PIX, P.cinv_pix != nullptr)) cf = (double)P.cinv_pix[ipx] * fs;
        }
    }
    // reciprocal mode: a(i,m) = b(i,m) + C with C = OFF - t0*fs, so A[m] := B[m] + floor(C)
    symC = 0.0; symCi = 0;
    if constexpr (C::SYM) { symC = tapinfo<C::INTERP>::OFF - (double)geo_Pv()[3] * fs; symCi = (int)floor(symC); }
    xbytes = (uint64_t)(N >> (C::ACT ? P.stage_shift : 0)) * M * (uint64_t)T * C::SB;
}

// ------------------------------------------------------------------------------------------------- delays
// Delay model of the BLOCK elements (the MB "transmits" of a stage) and of the STAGE elements (its "receiver"):
// 0 = distance, 1 = signed distance (focused wave: copysign by the normal, src/bf.cu:106-108), 2 = plane wave (dot product).
// 'DAS' / 'SYN': block = transmits (kind from VS / DV), stage = receivers (distance).  'MUL' (keep the transmit dimension) runs
// the same kernel with the roles swapped by the host: block = receivers, stage = transmits -- tables, strides and kinds swap.
template <class C> __device__ __forceinline__ double Tile<C>::a_of(uint32_t m, const GT *gPv, const GT *gNv) const {
    // tau_tx*fs - t0*fs + OFF, reference src/bf.cu:104-108,114
    if constexpr (C::LUT) return (double)P.lut_tx[ipx + (P.i_begin + P.i_count) * m] + tapinfo<C::INTERP>::OFF;
    const double rx = (double)px - (double)gPv[4 * m], ry = (double)py - (double)gPv[4 * m + 1], rz = (double)pz - (double)gPv[4 * m + 2];
    const double dot = kindB ? rx * (double)gNv[3 * m] + ry * (double)gNv[3 * m + 1] + rz * (double)gNv[3 * m + 2] : 0.0;
    double dv = dot;
    if (kindB != 2) { const double len = dsqrt(rx * rx + ry * ry + rz * rz); dv = kindB == 0 ? len : copysign(len, dot); }
    return dv * cf - (double)gPv[4 * m + 3] * fs + tapinfo<C::INTERP>::OFF;
}
template <class C> __device__ __forceinline__ double Tile<C>::b_at(GT ex, GT ey, GT ez) const {      // tau_rx*fs, reference src/bf.cu:110
    GT qx = px, qy = py, qz = pz;
    asm volatile("" : "+v"(qx), "+v"(qy), "+v"(qz));      // opaque: the fp64 images of the pixel are re-made per call (3 conversions) instead of living in 6 registers through the stage loop
    const double rx = (double)qx - (double)ex, ry = (double)qy - (double)ey, rz = (double)qz - (double)ez;
    return dsqrt(rx * rx + ry * ry + rz * rz) * cf;
}
// delay of STAGE element n at (ex,ey,ez): a receiver (kind 0), or -- roles swapped -- a transmit with {t0, normal} in P.St (scalar loads)
template <class C> __device__ __forceinline__ double Tile<C>::s_at(uint32_t n, GT ex, GT ey, GT ez) const {
    if constexpr (C::LUT) return (double)P.lut_rx[ipx + (P.i_begin + P.i_count) * n];
    if (C::F64 || !QSPEC(HAS_ST, P.St != nullptr)) return b_at(ex, ey, ez);
    GT qx = px, qy = py, qz = pz;
    asm volatile("" : "+v"(qx), "+v"(qy), "+v"(qz));
    const double rx = (double)qx - (double)ex, ry = (double)qy - (double)ey, rz = (double)qz - (double)ez;
    const double dot = kindS ? rx * (double)P.St[4 * n + 1] + ry * (double)P.St[4 * n + 2] + rz * (double)P.St[4 * n + 3] : 0.0;
    double dv = dot;
    if (kindS != 2) { const double len = dsqrt(rx * rx + ry * ry + rz * rz); dv = kindS == 0 ? len : ((C::ACT && kindS == 3) ? ((n & 1u) ? len : -len) : copysign(len, dot)); }
    return dv * cf - (double)P.St[4 * n] * fs;
}

// ---- per-(pixel, receiver) apodization (optional; uniform runtime switch, nothing of it in the pair loop)
// Two halves: the LOAD (raw bits, requested a stage ahead) and the CONVERSION to fp32 (done where the weight is used).  As one
// expression the conversion was scheduled right behind the load: every stage then waited out a full global-memory latency.
// kindS == 3: is my pixel on the side of the focal plane that stage element n (transmit n >> 1, side n & 1) stands for?  The sign test of
// src/bf.cu:107 (copysign: a zero dot product counts as behind the plane), evaluated in fp64 like the delay itself.
template <class C> __device__ __forceinline__ bool Tile<C>::on_side(uint32_t n, GT ex, GT ey, GT ez) const {
    const double rx = (double)px - (double)ex, ry = (double)py - (double)ey, rz = (double)pz - (double)ez;
    const double dot = rx * (double)P.St[4 * n + 1] + ry * (double)P.St[4 * n + 2] + rz * (double)P.St[4 * n + 3];
    return (__builtin_signbit(dot) != 0) == ((n & 1u) == 0u);
}
// image: the weight of my pixel's MIRROR IMAGE and the mirrored stage element N-1-n (lateral-mirror mode; array weights only: a generated
// rule has the same value there, the host checks that the element normals are mirror-symmetric too)
template <class C> __device__ __forceinline__ typename Tile<C>::wraw Tile<C>::wload_raw(uint32_t n, bool image) const {
    const int gen_kind = QSPEC(GEN_KIND, P.gen_kind);
    uint32_t na = n;                                  // the stage element's entry of the weight array
    if constexpr (C::ACT && !C::LUT) {
        if constexpr (C::FMOD && C::WTAB) {           // (at the register limit: the side rule alone; qdas_api.hip does not ask for 6 here)
            if (gen_kind == 5) { const float4 e = nrec[n]; return wraw{on_side(n, e.y, e.z, e.w) ? 0x3f800000u : 0u, 0u}; }
        } else if (gen_kind == 5 || gen_kind == 6) {  // 6: the side rule times a pixel x transmit (or pixel-only) array, transmit n >> 1
            const float4 e = nrec[n];
            const bool on = on_side(n, e.y, e.z, e.w);
            if (gen_kind == 5 || !on) return wraw{on ? 0x3f800000u : 0u, 0u};
            na = n >> 1;
        }
    }
    bool generated = !C::SYM && gen_kind;
    if constexpr (C::ACT && !C::LUT && !(C::FMOD && C::WTAB)) { if (gen_kind == 6) generated = false; }
    if (generated) {                                  // qdas.h QDAS_RXAPOD_*: element from the LDS record (never in reciprocal mode)
        const float4 e = nrec[n];
        return wraw{__float_as_uint(rx_apod_generated(gen_kind, P.gen_p0, P.gen_p1, px, py, pz, e.y, e.z, e.w, P.rxn, n)), 0u};
    }
    uint64_t k = ipx + (P.apix_pixel_only ? 0ull : P.I1 * P.I2 * P.I3 * na);
    if constexpr (C::WMIR) { if (image) k = ipx2 + (P.apix_pixel_only ? 0ull : P.I1 * P.I2 * P.I3 * (uint64_t)(N - 1u - na)); }
    if (QSPEC(APIX_REAL, P.apix_real)) {
        if constexpr (C::F32) return wraw{((const uint32_t *)P.apix)[k], 0u};
        else {
            // the 16 bits land in the low half of a register whose upper half is left UNDEFINED: as `(uint32_t)ushort` the zero extension was a
            // v_and_b32 right behind the load -- an s_waitcnt vmcnt and a full global-memory latency in every stage head (round 6: read off the
            // ISA of BASELINE C5's kernel; phase timers 17-30 % of a workgroup's life).  wconv() reads the low half only (v_cvt_f32_f16).
            typedef _Float16 h2raw __attribute__((ext_vector_type(2)));
            h2raw hv;
            hv.x = ((const _Float16 *)P.apix)[k];
            return wraw{__builtin_bit_cast(uint32_t, hv), 0u};
        }
    } else {
        if constexpr (C::F32) { const uint2 v = ((const uint2 *)P.apix)[k]; return wraw{v.x, v.y}; }
        else return wraw{((const uint32_t *)P.apix)[k], 0u};
    }
}
template <class C> __device__ __forceinline__ v2f Tile<C>::wconv(wraw r) const {
    bool generated = !C::SYM && QSPEC(GEN_KIND, P.gen_kind);
    if constexpr (C::ACT && !C::LUT && !(C::FMOD && C::WTAB)) { if (QSPEC(GEN_KIND, P.gen_kind) == 6) generated = false; }
    if (generated || (C::F32 && QSPEC(APIX_REAL, P.apix_real))) return (v2f){__uint_as_float(r.a), 0.f};
    if (QSPEC(APIX_REAL, P.apix_real)) return (v2f){__half2float(__ushort_as_half((unsigned short)r.a)), 0.f};
    if constexpr (C::F32) return (v2f){__uint_as_float(r.a), __uint_as_float(r.b)};
    else return half2_to_v2f(r.a);
}

// ------------------------------------------------------------------------------------------------- aperture share
// ksplit workgroups per tile when the image has too few tiles to fill the GPU; their partial sums are added in a fixed order by
// tile_reduce_kernel.  Reciprocal mode: every S-th transmit block, dealt out boustrophedon (0..S-1, S-1..0, ...) because block kb
// pairs with 16(kb+1) receivers -- the triangular work is balanced; otherwise: a contiguous range of receivers, all transmit blocks.
template <class C> __device__ __forceinline__ void Tile<C>::plan_stages() {
    n_lo = C::SYM ? 0u : (uint32_t)((uint64_t)N * split / S);
    n_hi = (uint32_t)((uint64_t)N * (split + 1) / S);
    Sm = 1u; sm = 0u;
    acc = acc1 = acc2 = acc3 = (v2f){0.f, 0.f};
    dacc[0] = dacc[1] = dacc[2] = dacc[3] = 0.0;
#pragma unroll
    for (int f = 0; f < C::NFR; ++f) tot[f] = (v2f){0.f, 0.f};
    // weights from an I x N array, or generated from the geometry (fp32 frames with such a weight do not share launches: their single-frame
    // kernel has the stage list of the active receivers instead, and the two-frame kernels have no registers for the weight bookkeeping)
    wpix = !C::F64 && !C::SYM && !C::BF && !(C::FBX && C::F32 && !C::WMIR) && (QSPEC(HAS_APIX, P.apix != nullptr) || QSPEC(GEN_KIND, P.gen_kind) != 0);
    if constexpr (C::W64) wpix = P.apix != nullptr;      // (fp64 data: arrays only, checked by the host)
    dtot[0] = dtot[1] = 0.0; w64[0] = 1.0; w64[1] = 0.0;
    bp = false;
    if constexpr (C::BPIX) bp = P.bpix != nullptr;
    syn = !C::SYM && !C::BF && C::F32 && QSPEC(SYN, P.syn);          // keep the stage dimension: one output plane per stage element
    fa = C::FB4 ? __builtin_amdgcn_readfirstlane(wave / C::MB) : 0;   // window sets this wave stages: (0, 1) in general; four frames: (0, 2) / (1, 3)
    fb = C::FB4 ? fa + 2 : 1;
    soff = soff2 = 0; offD = offM = 0;
    rsD = __builtin_amdgcn_make_buffer_rsrc((void *)P.x, 0, 0, 0x00020000); rsM = rsD;
    // Pixel x receiver weights (an acceptance-angle or f-number mask, say): receivers whose weight is zero for EVERY pixel of the tile
    // -- typically more than half of them -- are dropped from the tile's stage list: no staging, no barrier, no delay evaluation.
    // (With one window buffer in flight a stage that every wave skips still costs a full LDS-DMA latency.)
    nact = n_hi - n_lo; use_act = false;
    if constexpr (C::ACT) {
        if (wpix) {
            use_act = true;
            // a split aperture: the workgroups of a tile take every S-th receiver instead of contiguous ranges -- the receivers that carry
            // weight form a band (acceptance angle, f-number), which a contiguous split hands to ONE of the workgroups
            const bool wmode = P.pro_mask == 2;        // (plan creation's mask pass: ALL stage elements, whatever split the kernel was built for)
            if (wmode) { n_lo = 0; n_hi = N; }
            const bool inter = !wmode && S > 1 && !QSPEC(SYN, P.syn);
            // two-dimensional split (tile_params.h ksplit_m): my transmit-block group, and my class of stage elements among S / Sm
            if (inter && P.ksplit_m > 1u) { Sm = P.ksplit_m; sm = split % Sm; }
            const uint32_t Sn = S / Sm, sn = split / Sm;
            const uint32_t a_first = inter ? sn : n_lo, a_step = inter ? Sn : 1u;
            if (inter) { n_lo = 0; n_hi = N; }
            const uint32_t a_cnt = inter ? (N > sn ? (N - sn + Sn - 1) / Sn : 0u) : n_hi - n_lo;
            uint32_t *flg = (uint32_t *)part;          // prologue scratch (the windows are not in use yet)
            // the plan's cached activity mask of this tile (tile_params.h pro_mask): loaded instead of found
            const uint32_t *cmask = (P.pro_tab && P.pro_mask == 1) ? (const uint32_t *)(P.pro_tab + (size_t)(tile_id - P.tiles_z * P.tile_x0) * (2 * ((size_t)M + N) + 8 + (N + 31) / 32) + 2 * ((size_t)M + N) + 8) : nullptr;
            for (uint32_t k = tid; k < (N + 31) / 32; k += C::THREADS) flg[k] = cmask ? cmask[k] : 0u;
            __syncthreads();
            if (cmask) {}
            else if (QSPEC(GEN_KIND, P.gen_kind) != 0) {   // generated weights: arithmetic only (ONE call site of the out-of-line rule)
                for (uint32_t k = 0; k < a_cnt; ++k) {
                    const uint32_t n = a_first + a_step * k;
                    const v2f w = wload(n);
                    const bool any = __ballot(!(w.x == 0.f && w.y == 0.f)) != 0ull;
                    if (any && lane == 0) atomicOr(&flg[n >> 5], 1u << (n & 31u));
                }
            } else {                                   // array weights: four loads in flight
                for (uint32_t k0 = 0; k0 < a_cnt; k0 += 4) {
                    wraw r4[4];
                    uint32_t n4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { n4[q] = a_first + a_step * (k0 + q < a_cnt ? k0 + q : a_cnt - 1); r4[q] = wload_raw(n4[q]); }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const v2f w = wconv(r4[q]);
                        bool nz = !(w.x == 0.f && w.y == 0.f);
                        if constexpr (C::WMIR) {         // (lateral-mirror mode: the stage also serves the mirror images of my pixels at receiver N-1-n)
                            if (QSPEC(MIR, P.mir)) { const v2f w2 = wconv(wload_raw(n4[q], true)); nz = nz || !(w2.x == 0.f && w2.y == 0.f); }
                        }
                        const bool any = __ballot(nz) != 0ull;
                        if (any && lane == 0 && k0 + q < a_cnt) atomicOr(&flg[n4[q] >> 5], 1u << (n4[q] & 31u));
                    }
                }
            }
            __syncthreads();
            if (wmode) {                                // plan creation (qdas_api.hip plan_cache_activity): the tile's mask over ALL stage elements goes to the plan's table; nothing else runs
                uint32_t *o = (uint32_t *)(P.pro_out + (size_t)(tile_id - P.tiles_z * P.tile_x0) * (2 * ((size_t)M + N) + 8 + (N + 31) / 32) + 2 * ((size_t)M + N) + 8);
                for (uint32_t k = tid; k < (N + 31) / 32; k += C::THREADS) o[k] = flg[k];
                nact = 0; nstage = 0;
                return;
            }
            if (tid == 0) {
                uint32_t cnt = 0;
                for (uint32_t k = 0; k < a_cnt; ++k) {
                    const uint32_t n = a_first + a_step * k;
                    if ((flg[n >> 5] >> (n & 31u)) & 1u) act[cnt++] = make_uint2(n, __float_as_uint(nrec[n].x));
                }
                act[N] = make_uint2(cnt, 0u);
            }
            __syncthreads();
            nact = (uint32_t)__builtin_amdgcn_readfirstlane((int)act[N].x);
            __syncthreads();                           // (flg aliases the first window buffer)
        }
    }
    nstage = 0;
    if (use_act) { for (uint32_t r = 0; blk(r) < M; ++r) nstage += nact; }
    else { for (uint32_t r = 0; blk(r) < M; ++r) nstage += nlim(blk(r)) - n_lo; }
}

// ------------------------------------------------------------------------------------------------- the stage loop
// Stages: mb = transmit block, n = receiver; n is the inner index.  CHECK: the tile touches the ends of the record.
template <class C> template <bool CHECK> __device__ __forceinline__ void Tile<C>::run() {
    constexpr int NBUF = C::NBUF, NDMA = C::NDMA;
    // the k-th stage of a transmit block works on receiver nsel(k); a block has klim(m0) stages
    auto nsel = [&](uint32_t k) -> uint32_t { if constexpr (C::ACT) { if (use_act) return (uint32_t)__builtin_amdgcn_readfirstlane((int)act[k < N ? k : N - 1].x); } return n_lo + k; };   // (uniform: scalar register)
    auto klim = [&](uint32_t mm) -> uint32_t { if constexpr (C::ACT) { if (use_act) return nact; } return nlim(mm) - n_lo; };
    v2f wcur = {1.f, 0.f}, wcur2 = {1.f, 0.f};      // (wcur2: the weight of my pixel's mirror image, TileCfg::WMIR)
    wraw wnext_r = {0x3f800000u, 0u}, wnext_r2 = {0x3f800000u, 0u};
    const uint32_t n_first = nstage ? nsel(0) : n_lo;
    bool wmir = false;                                 // array weights in lateral-mirror mode: two weights per stage
    if constexpr (C::WMIR) wmir = wpix && QSPEC(MIR, P.mir) && QSPEC(HAS_APIX, P.apix != nullptr) && QSPEC(GEN_KIND, P.gen_kind) == 0;
    // fp64 data: the weight of (my pixel, stage element n) as a double pair, straight from the array (real or complex128)
    auto wload64 = [&](uint32_t nn, double &wr, double &wi) {
        const uint64_t k = ipx + (P.apix_pixel_only ? 0ull : P.I1 * P.I2 * P.I3 * (uint64_t)nn);
        if (P.apix_real) { wr = ((const double *)P.apix)[k]; wi = 0.0; }
        else { const double2 v = ((const double2 *)P.apix)[k]; wr = v.x; wi = v.y; }
    };
    double w64n[2] = {1.0, 0.0};                     // the next stage's weight, requested a stage ahead
    if constexpr (C::W64) { if (wpix) wload64(n_first, w64[0], w64[1]); }
    else if (wpix) { wcur = wload(n_first); wcur2 = wcur; if constexpr (C::WMIR) { if (wmir) wcur2 = wconv(wload_raw(n_first, true)); } }
    float tbc = 0.f, tbn = 0.f;                        // LUT: this / the next stage's receive delay of my pixel
    const uint64_t Ilut = P.i_begin + P.i_count;
    if constexpr (C::LUT) tbc = P.lut_rx[ipx + Ilut * n_first];
    uint32_t pr = 0, pk = 0, pm0 = blk(0);             // stage the DMA front is at (NBUF-1 stages ahead)
    dma_block(pm0);
    // The receiver of the stage at the DMA front and its window base B[n] travel in VGPRs, loaded one stage before they are needed:
    // every LDS read of a stage is issued BEFORE the stage's LDS-DMA in program order (the compiler orders a later LDS read behind
    // the DMA's vmcnt), and nothing waits for an LDS round trip just to form the next address (a stage list with gaps keeps
    // {receiver, base} in one table entry).
    auto rec_base = [&](uint32_t k) -> float { if constexpr (C::F64) return __int_as_float(nrec64[k].b); else return nrec[k].x; };
    float vbn; uint32_t vpn;
    auto front_entry = [&](uint32_t kk) {
        if constexpr (C::ACT) {
            if (use_act) { const uint2 e = act[kk < N ? kk : N - 1]; vpn = e.x; vbn = __uint_as_float(e.y); return; }
        }
        const uint32_t q = n_lo + kk;
        vpn = q; vbn = rec_base(q < N ? q : N - 1);
    };
    front_entry(0);
    // staging of the stage at the front, in two steps: its LDS read, then the DMA itself (the stage loop puts its global loads between them)
    int dbn = 0; uint32_t dpn = 0;
    auto dma_prep = [&]() {
        dbn = __builtin_amdgcn_readfirstlane(__float_as_int(vbn));
        dpn = (uint32_t)__builtin_amdgcn_readfirstlane((int)vpn);
        front_entry(pk + 1 == klim(pm0) ? 0u : pk + 1);             // of the stage after this one
    };
    auto dma_go = [&](int buf) {
        if constexpr (C::ACT) {                         // (stage lists with gaps: no running offset)
            if (use_act) {
                soff = ((dpn >> P.stage_shift) - (n_lo >> P.stage_shift)) * (uint32_t)strN * (uint32_t)C::SB;
                if constexpr (C::WMIR) { if (QSPEC(MIR, P.mir)) soff2 = (n_hi - 1u - dpn) * (uint32_t)strN * (uint32_t)C::SB; }   // (mirrored receiver N-1-n above the descriptor's base N - n_hi)
            }
        }
        stage_dma(dbn, buf);
        if (++pk == klim(pm0)) { pk = 0; pm0 = blk(++pr); dma_block(pm0); }
    };
    auto dma_next = [&](int buf) { dma_prep(); dma_go(buf); };
    // stage weights (TileCfg::WST): wave 0 fetches the stage's MB table entries w[n, m0 .. m0+MB-1] (wave 1, reciprocal mode: the mirror
    // entries w[m0 .., n]) one stage ahead -- requested with the stage head's global loads, written to LDS after the pair loop
    constexpr uint32_t WBUF = 2u * C::MB * 8u + 16u;
    float2 wnx = {0.f, 0.f};
    auto wst_load = [&](uint32_t nn, uint32_t mm0) {
        if constexpr (C::WST) {
            const uint32_t ln = lane_now(), mm = mm0 + ln;
            if (wave == 0 && ln < (uint32_t)C::MB) wnx = mm < M ? ((const float2 *)P.wtab)[nn + (size_t)N * mm] : make_float2(0.f, 0.f);
            if constexpr (C::SYM) { if (wave == 1 && ln < (uint32_t)C::MB) wnx = ((const float2 *)P.wtab)[mm + (size_t)N * nn]; }
        }
    };
    auto wst_store = [&](int b) {
        if constexpr (C::WST) {
            if (wave == 0 || (C::SYM && wave == 1)) {
                const uint32_t ln = lane_now();
                const bool nz = ln < (uint32_t)C::MB && !(wnx.x == 0.f && wnx.y == 0.f);
                // (64-bit masks: the long-stage one-set builds stage 64 transmits -- round 5 kept 32 bits here, and a zero weight among the first 32 of such a stage
                //  sent the zero tests of transmits 32..63 to shifted-out bits: found by the fuzz soak of round 6, seed 50160)
                const uint64_t mask = __ballot(nz);
                unsigned char *q = wst + (uint32_t)b * WBUF + (wave ? C::MB * 8 : 0);
                if (ln < (uint32_t)C::MB) ((float2 *)q)[ln] = wnx;
                if (ln == 0) ((uint64_t *)(wst + (uint32_t)b * WBUF + 2 * C::MB * 8))[wave ? 1 : 0] = mask;
            }
        }
    };
    if (nstage) { wst_load(n_first, blk(0)); wst_store(0); }
    if constexpr ((QDAS_DBG_SYNC & 16) != 0) __syncthreads();
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b)
        if ((uint32_t)b < nstage) dma_next(b);
    // counted wait: everything but the newest (NBUF-2) stages has landed; then publish to the workgroup
    if (nstage >= (uint32_t)(NBUF - 1)) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NBUF - 2) * NDMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    int buf = 0;
    uint32_t cr = 0, k = 0, n = n_first, m0 = blk(0);
    for (uint32_t st = 0; st < nstage; ++st) {
        timer.mark(0);
        if constexpr ((QDAS_DBG_SYNC & 1) != 0) __syncthreads();
        const bool more = st + (NBUF - 1) < nstage;
        // receiver of the next stage: with one stage of staging in flight that is where the DMA front stands (no LDS round trip)
        const uint32_t n_next = (NBUF == 2 && C::ACT) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)vpn) : nsel(k + 1 == klim(m0) ? 0u : k + 1);
        const bool skip = wpix && (__ballot(wcur.x != 0.f || wcur.y != 0.f || (C::WMIR && (wcur2.x != 0.f || wcur2.y != 0.f))) == 0ull);   // whole wave weightless: no gathers
        // {B[n], receiver position}: one broadcast LDS read (fp64 data: two), issued ahead of the DMA
        int rec_b; GT rec_x, rec_y, rec_z;
        if constexpr (C::F64) { const rec64 r = nrec64[n]; rec_b = r.b; rec_x = r.x; rec_y = r.y; rec_z = r.z; }
        else { const float4 r = nrec[n]; rec_b = __float_as_int(r.x); rec_x = r.y; rec_y = r.z; rec_z = r.w; }
        float phB = 0.f;                               // remodulation: frac(B[n]*fmod/fs) (tile_prologue.h)
        if constexpr (C::FMOD && !C::F64) {              // (fp64 data: the pair loop forms the whole phase in double)
            phB = Bext[n];
            if constexpr (C::MIRQ) phB = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(phB)));   // (uniform: a scalar register -- the one vector register the four window sets are short of)
        }
        uint64_t wmask = ~0ull, xmask = ~0ull;                   // which of this stage's table entries are non-zero (zero weights are skipped, src/bf.cu:122,126)
        if constexpr (C::WST) {
            const uint4 mk = *(const uint4 *)(wst + (uint32_t)buf * WBUF + 2 * C::MB * 8);
            wmask = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)mk.x) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)mk.y) << 32);
            if constexpr (C::SYM) xmask = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)mk.z) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)mk.w) << 32);
        }
        // The next stage's staging is issued at the start of this stage by the younger half of the waves and AFTER the pair
        // loop by the older half: the hardware favours older waves, they finish their pair loop early and would only wait at
        // the barrier -- their (scalar-heavy) issue phase then overlaps the younger waves' arithmetic instead of everybody's.
        const bool dma_late = C::SYM && !hooks::no_late_dma && wave < C::WAVES / 2;
        const bool dma_now = !hooks::no_stage_dma && more && !dma_late;
        if (dma_now) dma_prep();
        // next stage's pixel weight / table delay: global loads, requested AFTER every LDS read of the stage's head (the compiler puts
        // an s_waitcnt vmcnt(0) in front of an LDS read that follows them -- it cannot tell them from the LDS-DMA it has to order
        // LDS reads behind: each stage then waited out a global-memory latency) and BEFORE this stage's DMA (so that the counted
        // end-of-stage wait covers them)
        if constexpr (C::W64) { if (wpix && st + 1 < nstage) wload64(n_next, w64n[0], w64n[1]); }
        else if (wpix && st + 1 < nstage) { wnext_r = wload_raw(n_next); if constexpr (C::WMIR) { if (wmir) wnext_r2 = wload_raw(n_next, true); } }
        if constexpr (C::LUT) { if (st + 1 < nstage) tbn = P.lut_rx[ipx + Ilut * n_next]; }
        if (st + 1 < nstage) wst_load(n_next, k + 1 == klim(m0) ? blk(cr + 1) : m0);
        if (dma_now) dma_go((buf + NBUF - 1) % NBUF);            // lands during the next NBUF-1 stages
        if constexpr ((QDAS_DBG_SYNC & 32) != 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if constexpr ((QDAS_DBG_SYNC & 2) != 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __syncthreads(); }
        timer.mark(1);
        if constexpr (C::F64) {
            if (k == 0) {                              // new transmit block: refresh the block residuals (cold: once per N stages)
#pragma unroll
                for (int p = 0; p < C::MB; ++p) {
                    const uint32_t ma = m0 + p < M ? m0 + p : M - 1;
                    rad[p] = block_residual64(px, py, pz, cf, fs, kindB, (lds_cdouble *)PvL, (lds_cdouble *)NvL, ma, Abase[ma], tapinfo<C::INTERP>::OFF);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else if (k == 0) {                           // new transmit block: refresh the block residuals (cold: once per N stages)
#pragma unroll
            for (int p = 0; p < C::MB / 2; ++p) {
                const uint32_t ma = m0 + 2 * p < M ? m0 + 2 * p : M - 1, mb = m0 + 2 * p + 1 < M ? m0 + 2 * p + 1 : M - 1;
                if constexpr (C::SYM) {                // a - A - 1/2 = (b - B) + frac(C) - 1/2, from the receiver records
                    const float4 ea = nrec[ma], eb = nrec[mb];
                    const double fc = symC - (double)symCi - 0.5;
                    ra[p] = (v2f){(float)(b_at(ea.y, ea.z, ea.w) - (double)__float_as_int(ea.x) + fc),
                                  (float)(b_at(eb.y, eb.z, eb.w) - (double)__float_as_int(eb.x) + fc)};
                } else if constexpr (C::LUT)
                    ra[p] = (v2f){(float)(a_of(ma, nullptr, nullptr) - ((double)Abase[ma] + 0.5)), (float)(a_of(mb, nullptr, nullptr) - ((double)Abase[mb] + 0.5))};
                else
                    ra[p] = (v2f){block_residual(px, py, pz, cf, fs, kindB, (lds_cfloat *)PvL, (lds_cfloat *)NvL, ma, Abase[ma], tapinfo<C::INTERP>::OFF),
                                  block_residual(px, py, pz, cf, fs, kindB, (lds_cfloat *)PvL, (lds_cfloat *)NvL, mb, Abase[mb], tapinfo<C::INTERP>::OFF)};
                __builtin_amdgcn_sched_barrier(0);      // one pair at a time: keeps this cold block from inflating the register budget
            }
            if constexpr (C::BPIX) {
                if (!bp) {                             // (no such weight: ones -- the pair loop multiplies unconditionally: a uniform branch there
#pragma unroll                                         //  makes the compiler clone the unrolled loop and spill 270 registers)
                    for (int p = 0; p < C::MB / 2; ++p) bw[p] = (v2f){1.f, 1.f};
                } else {                               // this block's pixel x block-element weights (cold: once per N stages)
                    const uint64_t Itot = P.I1 * P.I2 * P.I3;
                    uint32_t ip32 = (uint32_t)ipx;         // (uniform row pointer + 32-bit lane offset: the host keeps such plans below 2^30 pixels)
                    asm volatile("" : "+v"(ip32));
#pragma unroll
                    for (int p = 0; p < C::MB / 2; ++p) {
                        const uint32_t ma = m0 + 2 * p < M ? m0 + 2 * p : M - 1, mb = m0 + 2 * p + 1 < M ? m0 + 2 * p + 1 : M - 1;
                        const float *rowa = P.bpix + Itot * ma, *rowb = P.bpix + Itot * mb;
                        asm volatile("" : "+s"(rowa), "+s"(rowb));
                        bw[p] = (v2f){rowa[ip32], rowb[ip32]};
                        __builtin_amdgcn_sched_barrier(0);  // one pair at a time (as above)
                    }
                }
            }
        }
        timer.mark(2);
        if constexpr ((QDAS_DBG_SYNC & 4) != 0) __syncthreads();
        if constexpr (C::F64) {                        // everything in fp64; the low word of (t + 1.5*2^52) IS rint(t)
            const int bn = rec_b;
            const double rbd = s_at(n, rec_x, rec_y, rec_z) - (double)bn;
            const uint32_t cbase = win_off + (uint32_t)buf * (C::NW * C::WB);
            bool skip64 = false;                       // (a stage whose weight is zero for the whole wave reads no sample: src/bf.cu:122,126)
            if constexpr (C::W64) skip64 = wpix && __ballot(w64[0] != 0.0 || w64[1] != 0.0) == 0ull;
            if (skip64) {}
            else if (m0 + C::MB <= M) pairs_f64<CHECK, false>(n, m0, bn, rbd, cbase);
            else                      pairs_f64<CHECK, true>(n, m0, bn, rbd, cbase);
        } else if (!skip) {
            const int bn = rec_b;
            const float rb = C::LUT ? tbc - (float)bn : hooks::fake_rx_delay ? (float)(lane * 2 + 3) + 0.37f * (float)(n & 7)
                                                                            : (float)(s_at(n, rec_x, rec_y, rec_z) - (double)bn);
            // LDS byte address of a tap = bits(t + MAGIC)*SB + cbase + (window, tap) immediate
            const uint32_t cbase = win_off + (uint32_t)buf * (C::NW * C::WB) - (MAGIC_BITS * (uint32_t)C::SB);
            // full block (reciprocal mode: block entirely above the diagonal): check-free; else the tail / diagonal variant
            if (C::SYM ? (n < m0 && (!C::FOLD || m0 + C::MB <= M)) : (m0 + C::MB <= M)) {
                if constexpr (C::TWO && (C::F32 || C::SYM || QDAS_F16_PIPE) && C::K == 4 && !(CHECK || C::FMOD || C::WTAB) && !hooks::no_pipeline && !((QDAS_ONEACC_PLAIN || C::INTERP == 3) && C::ONEACC))      // (Lanczos weights + double-buffered taps + 16 residual pairs do not fit 128 registers: the plain loop, same-box 27.95 -> 28.2 ms on C3 without reciprocity, profiles/r06/oneacc_ab.txt)
                    pairs_pipelined(rb, cbase);
                else
                {
                    constexpr uint64_t FULL = C::MB >= 64 ? ~0ull : ((1ull << (C::MB & 63)) - 1ull);
                    bool done = false;
                    if constexpr (C::WST) {
                        if (wmask == FULL && (!C::SYM || xmask == FULL)) {          // no zero weight in this stage: the loop without zero tests
                            pairs_plain<CHECK, false, false>(n, m0, bn, rb, cbase, phB, wst_off + (uint32_t)buf * WBUF, wmask, xmask);
                            done = true;
                        }
                    }
                    if (!done) pairs_plain<CHECK, false>(n, m0, bn, rb, cbase, phB, wst_off + (uint32_t)buf * WBUF, wmask, xmask);
                }
            } else {
                pairs_plain<CHECK, true>(n, m0, bn, rb, cbase, phB, wst_off + (uint32_t)buf * WBUF, wmask, xmask);
            }
        }
        if constexpr ((QDAS_DBG_SYNC & 8) != 0) __syncthreads();
        if constexpr (!hooks::no_fair_prio) __builtin_amdgcn_s_setprio(3);     // stage epilogue / next preamble at full priority (tile_pairs.h)
        if (!hooks::no_stage_dma && more && dma_late) dma_next((buf + NBUF - 1) % NBUF);
        if (st + 1 < nstage) wst_store((buf + 1) % NBUF);          // (the barrier below publishes it)
        timer.mark(3);
        // stage st+1 must have landed (all but the NBUF-2 newest DMA groups), all my LDS reads are done
        if (!hooks::no_stage_barrier) {
            if (more) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((NBUF - 2) * NDMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        timer.mark(4);
        timer.stage_done();
        buf = (buf + 1 == NBUF) ? 0 : buf + 1;
        if (syn) {                                     // 'SYN' (src/bf.cu:131-133): the stage's sum over its transmits joins plane n of y.
            // Non-returning fp32 atomics: the planes are zero-filled by the host, the transmit blocks of one (pixel, n) are
            // visited in order by this lane only (and receiver ranges of a split aperture are disjoint) -> deterministic.
            v2f Sf[4];
            frame_sums(Sf);
#pragma unroll
            for (int f = 0; f < C::NFR; ++f) {
                if (wpix) Sf[f] = (v2f){wcur.x * Sf[f].x - wcur.y * Sf[f].y, wcur.x * Sf[f].y + wcur.y * Sf[f].x};
                if (in_shard()) {
                    float2 *plane = (float2 *)P.y + (size_t)f * P.y_fstride + (size_t)n * P.y_ld;     // uniform: stays in SGPRs;
                    asm volatile("" : "+s"(plane));                // opaque, so that no per-frame 64-bit lane address is hoisted out of the stage loop
                    uint32_t po = pofs;
                    asm volatile("" : "+v"(po));                   // (likewise: pofs*8 as a 64-bit loop invariant would cost two registers for the whole loop)
                    float *q = (float *)(plane + po);
                    unsafeAtomicAdd(q, Sf[f].x); unsafeAtomicAdd(q + 1, Sf[f].y);
                }
            }
            if (wpix) { if constexpr (!C::F32) asm volatile("" : "+v"(wnext_r.a)); wcur = wconv(wnext_r); }   // (fp16 weights: converted here, a stage after the load)
            acc = acc1 = acc2 = acc3 = (v2f){0.f, 0.f};
        } else if (C::W64 && wpix) {                   // fp64 data: the stage's sum times the stage element's weight joins the weighted total
            if constexpr (C::W64) {
                const double sr = dacc[0] + dacc[2], si = dacc[1] + dacc[3];
                dtot[0] += w64[0] * sr - w64[1] * si; dtot[1] += w64[0] * si + w64[1] * sr;
                dacc[0] = dacc[1] = dacc[2] = dacc[3] = 0.0;
                w64[0] = w64n[0]; w64[1] = w64n[1];
            }
        } else if (wpix) {                             // weight the stage's partial sum (the weight does not depend on m)
            v2f Sf[4];
            frame_sums(Sf);
#pragma unroll
            for (int f = 0; f < C::NFR; ++f) {
                const v2f wf = (C::WMIR && f == 1) ? wcur2 : wcur;      // (lateral-mirror mode: the second sum is the mirror image's)
                tot[f] += (v2f){wf.x * Sf[f].x - wf.y * Sf[f].y, wf.x * Sf[f].y + wf.y * Sf[f].x};
            }
            acc = acc1 = acc2 = acc3 = (v2f){0.f, 0.f};
            if constexpr (!C::F32) asm volatile("" : "+v"(wnext_r.a));      // (fp16 weights: converted here, a stage after the load)
            wcur = wconv(wnext_r);
            wcur2 = wcur;
            if constexpr (C::WMIR) { if (wmir) { asm volatile("" : "+v"(wnext_r2.a)); wcur2 = wconv(wnext_r2); } }
        }
        if constexpr (C::LUT) tbc = tbn;
        if (++k == klim(m0)) { k = 0; m0 = blk(++cr); }
        n = n_next;
    }
}

// ---- epilogue: y[i] = pix  (reference src/bf.cu:140); one image per frame of the launch
template <class C> __device__ __forceinline__ void Tile<C>::epilogue() {
    if (syn || C::BF) return;                          // every stage already added its share to its plane / stored its pairs
    if constexpr (C::F64) {                            // (one frame per launch; the partial images of a split aperture are complex128: [split][pixel])
        if (pofs != NOT_MINE) {
            uint32_t po = pofs;
            asm volatile("" : "+v"(po));
            ST *base = S > 1 ? (ST *)P.part + (size_t)split * P.i_count : (ST *)P.y;
            asm volatile("" : "+s"(base));
            if (C::W64 && wpix) st(base, (size_t)po, cplx<double>{dtot[0], dtot[1]});
            else st(base, (size_t)po, cplx<double>{dacc[0] + dacc[2], dacc[1] + dacc[3]});
        }
    } else {
        v2f res[4];
        frame_sums(res);
        uint32_t po = pofs;
        if constexpr (C::SYM) { uint32_t i1, col; po = locate(i1, col, false); }      // (nothing in the reciprocal stage loop needs it: not kept alive)
        if constexpr (C::FOLDQ && C::FB2) {           // two frames of folded data in mirror mode: res = {f0 mine, f0 image, f1 mine, f1 image}
            uint32_t i1, col;
            const uint32_t po2 = locate(i1, col, false, true);
            const uint32_t pos[2] = {po, po2};
#pragma unroll
            for (int f = 0; f < 2; ++f) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    uint32_t q = pos[h];
                    if (q == NOT_MINE) continue;
                    asm volatile("" : "+v"(q));
                    const v2f r = res[2 * f + h];
                    if (S > 1) { float2 *base = P.part + ((size_t)split * 2 + f) * (P.i_count << (QSPEC(MIR, P.mir) == 2 ? 1 : 0)); asm volatile("" : "+s"(base)); base[q] = make_float2(r.x, r.y); }
                    else { ST *base = (ST *)P.y + (size_t)f * P.y_fstride; asm volatile("" : "+s"(base)); st(base, (size_t)q, cplx<float>{r.x, r.y}); }
                }
            }
            return;
        }
        if constexpr (C::FB2 || C::MIRQ) {
            if (C::MIRQ || QSPEC(MIR, P.mir)) {          // lateral-mirror modes: the second sum belongs to the mirror image of my pixel
                uint32_t i1, col;
                const uint32_t po2 = locate(i1, col, false, true);
                const uint32_t pos[2] = {po, po2};
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    uint32_t q = pos[f];
                    if (q == NOT_MINE) continue;
                    asm volatile("" : "+v"(q));
                    v2f r = res[f];
                    if constexpr (!C::MIRQ) { if (wpix) r = tot[f]; }       // (a pixel x receiver weight: the weighted totals)
                    if (S > 1) { float2 *base = P.part + (size_t)split * (P.i_count << (QSPEC(MIR, P.mir) == 2 ? 1 : 0)); asm volatile("" : "+s"(base)); base[q] = make_float2(r.x, r.y); }
                    else { ST *base = (ST *)P.y; asm volatile("" : "+s"(base)); st(base, (size_t)q, cplx<float>{r.x, r.y}); }
                }
                return;
            }
        }
        if (po != NOT_MINE) {
            asm volatile("" : "+v"(po));
#pragma unroll
            for (int f = 0; f < C::NFR; ++f) {
                const v2f r = wpix ? tot[f] : res[f];
                // partial images of a split aperture are laid out [split][frame][pixel]
                // (uniform bases made opaque: no 64-bit lane address is formed before the stage loop and carried through it)
                if (S > 1) { float2 *base = P.part + ((size_t)split * C::NFR + f) * P.i_count; asm volatile("" : "+s"(base)); base[po] = make_float2(r.x, r.y); }
                else { ST *base = (ST *)P.y + (size_t)f * P.y_fstride; asm volatile("" : "+s"(base)); st(base, (size_t)po, cplx<float>{r.x, r.y}); }
            }
        }
    }
}

// The whole kernel.  PROBE: plan-time variant that stops after the window-fit test.
template <class C, bool PROBE> __device__ __forceinline__ void das_tile_body(const TileParams &P, unsigned char *smem) {
    Tile<C> t(P);
    t.timer.begin();
    t.setup(smem);
    if (!t.template prologue<PROBE>()) return;         // uniform exit: misfit tile (generic kernel takes it) or probe
    if constexpr (!PROBE) {
        t.timer.prologue_done();
        t.plan_stages();
        if (P.pro_mask == 2) return;                    // (uniform: the mask pass of plan creation ends here)
        if (t.tile_interior) t.template run<false>(); else t.template run<true>();
        t.timer.finish(t.lane, t.wave, C::WAVES);
        t.epilogue();
    }
}

// PSZ / BPC: bytes per lane and DMA piece (16) and workgroups per CU the register budget is sized for -- kept in the kernel's
// name so that profiles of different rounds list the same kernels.  PROBE is a kernel of its own name: profiles of
// das_tile_kernel<..., false> hold full frames only.
template <int INTERP, typename ST, bool FMOD, bool WTAB, bool SYM, bool FB2, bool FB4, int WAVES, int MB, int W, int NBUF, int PSZ, int BPC, bool PROBE, bool BIG = false, bool LUT = false, bool BFM = false, bool MIRQ = false, bool FOLD = false>
__global__ void __launch_bounds__(WAVES * 64, WAVES * BPC / 4)
das_tile_kernel(const TileParams P) {
    static_assert(PSZ == 16, "16-byte DMA pieces");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    das_tile_body<TileCfg<INTERP, ST, FMOD, WTAB, SYM, FB2, FB4, WAVES, MB, W, NBUF, BIG, LUT, BFM, MIRQ, FOLD>, PROBE>(P, smem);
}

#ifndef __HIPCC_RTC__
}  // namespace qdas
#include "jit.h"
namespace qdas {
// $QDAS_KERNEL_CENSUS=<file>: every distinct prebuilt instantiation a process launches is appended to <file> (das_tile.hip; tools/kernel_census.py)
void tile_census(int ci, int interp, int sample_bytes, bool fm, bool wt, bool probe);
bool tile_prepare_only();          // das_tile.hip: this thread is resolving a plan's kernels (TilePrepare), not launching
template <int INTERP, typename ST, int CI>
static hipError_t launch_tile_i(const TileParams &P, unsigned ntiles, size_t lds, hipStream_t s) {
    constexpr Cfg G = CFGS[CI];
    constexpr bool FOLD = (CI >= 17 && CI <= 21);       // folded data: with the lateral-mirror mode (narrow / wide windows), without; 20 / 21: two frames per launch
    constexpr bool MIRQ = (CI == 15 || CI == 16 || CI == 17 || CI == 18 || CI == 20);
    constexpr bool SYM = (CI == 1 || CI == 7 || CI == 8 || MIRQ || FOLD), FB2 = (CI == 3 || CI == 4 || CI == 20 || CI == 21), FB4 = (CI == 5 || CI == 6), BIG = (CI == 9), LUT = (CI == 10 || CI == 11), BFM = (CI == 12);
    const bool fm = P.fmod != 0.0, wt = P.wtab != nullptr;
    const dim3 g(ntiles * (P.probe ? 1u : P.ksplit)), b(G.waves * 64);
    tile_census(CI, INTERP, (int)sizeof(ST), fm, wt, P.probe != 0);
#define QDAS_LAUNCH(FM, WT) QDAS_LAUNCH_P(FM, WT, false)
    // instantiations libqdas.so does not carry (das_tile_cfg.h tile_prebuilt) are built on demand from the same template arguments (jit.hip lazy_tile_launch);
    // tile_prepare_only(): resolve the kernel, launch nothing (plan creation: the first execute of a plan never compiles)
#define QDAS_LAUNCH_P(FM, WT, PR)                                                                        \
    do {                                                                                                 \
        if constexpr (tile_prebuilt(CI, INTERP, FM, WT, PR)) {                                           \
            if (tile_prepare_only()) return hipSuccess;                                                  \
            auto kfn = das_tile_kernel<INTERP, ST, FM, WT, SYM, FB2, FB4, G.waves, G.mb, G.w, G.nbuf, G.psz, G.bpc, PR, BIG && !PR, LUT, BFM && !PR, MIRQ && !PR, FOLD && !PR>; \
            hipError_t e = hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e;                                                               \
            kfn<<<g, b, lds, s>>>(P);                                                                    \
        } else {                                                                                         \
            return lazy_tile_launch(LazySpec{INTERP, (int)sizeof(ST), FM, WT, CI, PR}, P, g.x, b.x, lds, s, tile_prepare_only()); \
        }                                                                                                \
    } while (0)
    if (P.probe) {
        if constexpr (FB2 || FB4) return hipErrorInvalidValue;    // the window fit does not depend on the frame count: probes use one frame
        else { QDAS_LAUNCH_P(false, false, true); return hipGetLastError(); }
    }
    if constexpr (sizeof(ST) == 16) {                  // fp64 data: no generated weight rule, no kept dimension (qdas_api.hip)
        if ((P.apix && fm) || P.gen_kind || P.syn) return hipErrorInvalidValue;   // (a weight array: the instantiations without remodulation, TileCfg::W64)
        if (fm && wt) QDAS_LAUNCH(true, true);         // (grid: ntiles * ksplit workgroups, as for the other data types)
        else if (fm)  QDAS_LAUNCH(true, false);
        else if (wt)  QDAS_LAUNCH(false, true);
        else          QDAS_LAUNCH(false, false);
    } else if constexpr (FOLD) {                      // (pixel-independent weights were applied by the fold pass: no table here)
        if (wt) return hipErrorInvalidValue;
        if (fm) QDAS_LAUNCH(true, false);
        else    QDAS_LAUNCH(false, false);
    } else if constexpr (MIRQ) {                      // (a weight table must be mirror-symmetric too: checked by the host)
        if (fm && wt) QDAS_LAUNCH(true, true);
        else if (fm)  QDAS_LAUNCH(true, false);
        else if (wt)  QDAS_LAUNCH(false, true);
        else          QDAS_LAUNCH(false, false);
    } else if constexpr (SYM) {
        if (fm && wt) QDAS_LAUNCH(true, true);
        else if (fm)  QDAS_LAUNCH(true, false);
        else if (wt)  QDAS_LAUNCH(false, true);
        else          QDAS_LAUNCH(false, false);
    } else if constexpr (FB4 && sizeof(ST) == 8) {
        // four fp32 frames per launch: 4 x 8 tap registers per transmit pair leave no room for the per-sample post-processing of
        // remodulation / weight tables or for the per-frame totals of a pixel x receiver weight -- those plans share launches
        // pairwise (qdas_api.hip), so that no instantiation needs scratch memory
        if (fm || wt || P.apix || P.gen_kind) return hipErrorInvalidValue;
        QDAS_LAUNCH(false, false);
    } else if constexpr (FB2 && sizeof(ST) == 8) {
        if (P.apix || P.gen_kind) return hipErrorInvalidValue;     // (fp32 frames with a pixel x receiver weight run one per launch: qdas_api.hip)
        if (fm && wt) QDAS_LAUNCH(true, true);
        else if (fm)  QDAS_LAUNCH(true, false);
        else if (wt)  QDAS_LAUNCH(false, true);
        else          QDAS_LAUNCH(false, false);
    } else {
        if (fm && wt) QDAS_LAUNCH(true, true);
        else if (fm)  QDAS_LAUNCH(true, false);
        else if (wt)  QDAS_LAUNCH(false, true);
        else          QDAS_LAUNCH(false, false);
    }
#undef QDAS_LAUNCH
#undef QDAS_LAUNCH_P
    return hipGetLastError();
}
#endif

}  // namespace qdas
