// plan_modes.h -- MODE RESOLUTION of a DAS plan as data: which kernel, which launch configuration, which symmetry modes -- and why not.
//
// Pure host code: no HIP, no allocation, no environment reads, no device probes.  qdas_plan_create (qdas_api.hip) gathers the FACTS these
// functions ask for (host comparisons of the geometry, the device's mirror check, whether the fold buffer could be allocated, the outcome of
// the window-fit probes) and executes what they decide; tests/modes/enumerate_modes.cpp compiles this header with g++ -- no GPU, no libqdas.so --
// and walks >= 10 000 descriptors x facts x probe outcomes through it: every draw must end in a launch configuration that launch_legal()
// accepts, or name the reason it does not run fused (the 715-line qdas_plan_create re-derived ~10 interacting mode bits in
// place -- `mir && big`, twin-plan tail frames were found there -- and none of it was testable without a GPU).
//
// Order of decisions (each a function below; "fact" = something only the caller can find out):
//   1. analyze_request      what the descriptor asks for: mode ('DAS' / 'SYN' / 'MUL' / 'BF'), sound-speed map, classification of the apodization arrays
//   2. resolve_symmetry     reciprocal mode + reciprocity fold (facts: Pv == Pr?, one t0?, fold buffer), roles of the apertures, lateral-mirror
//                           candidate -> mode (fact: the geometry's mirror check), LDS header and DMA stride limits (re-basing configuration)
//   3. ProbeChain           the sequence of launch configurations whose window fit is probed on the device (fact per step: misfit tiles?) and
//                           what is given up when tiles do not fit (narrow windows -> wide, mirror mode -> plain)
//   4. choose_ksplit        workgroups per tile
//   5. stream_modes         frames per launch of a stream
//   6. launch_legal         the launcher's own admission rules (das_tile.hip launch_tile calls THIS: one source of truth)
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <algorithm>
#include "../../include/qdas.h"
#include "das_tile_cfg.h"

#ifndef QDAS_MAX_APOD
#define QDAS_MAX_APOD 6
#endif

namespace qdas {

// ---- LDS budget of the launch configurations (das_tile_impl.h Tile::setup) -- pure, used by the launcher and by the resolver
struct TileConfig { int waves; int mb; int window; size_t lds_bytes; int threads; };

// fb = 2: the two-window-set configuration of the general mode (frames sharing a launch; lateral-mirror mode)
inline TileConfig tile_config(int dtype, int sym, int narrow = 0, int fb = 1, int mirq = 0, int fold = 0) {
    const Cfg &g = CFGS[cfg_index(dtype, sym, fb, narrow, mirq, fold)];
    TileConfig c;
    c.waves = g.waves;
    c.mb = g.mb;
    c.window = g.w;
    c.threads = g.waves * 64;
    const int sets = (fold && sym && dtype == 1) ? (mirq ? 2 : 1) * (fb == 2 ? 2 : 1) : (mirq && sym) ? 4 : (sym || fb == 2) ? 2 : 1;      // window sets per buffer
    c.lds_bytes = (size_t)g.nbuf * g.mb * sets * g.w * (dtype == 2 ? 4 : dtype == 0 ? 16 : 8);
    return c;
}

// dynamic LDS of one workgroup (pixw: a pixel x receiver weight -> the tile keeps a stage list; wtab: stage weights staged per stage)
inline size_t tile_lds_bytes(int dtype, int sym, uint64_t N, uint64_t M, int narrow = 0, int pixw = 0, int wtab = 0, int mirq = 0, int fold = 0, int fb = 1) {
    const Cfg &g = CFGS[cfg_index(dtype, sym, fold ? fb : 1, narrow, mirq, fold)];
    const TileConfig c = tile_config(dtype, sym, narrow, fold ? fb : 1, mirq, fold);
    const size_t MX = std::min<size_t>(M > N ? M : N, QDAS_PROLOGUE_CHUNK);
    // (geometry tables in the plan's real type: 32-byte receiver records and 8-byte table entries for fp64 data -- Tile::setup)
    const size_t off_act = (((((2 * M + N) * 4 + 15) & ~(size_t)15) + (dtype == 0 ? 32 : 16) * N + 7 * M * (dtype == 0 ? 8 : 4)) + 15) & ~(size_t)15;
    const size_t off_wst = off_act + (((pixw ? 8 * (N + 1) : 0) + 15) & ~(size_t)15);
    const size_t hdr = (off_wst + ((wtab && dtype != 0) ? (size_t)g.nbuf * (2 * (size_t)g.mb * 8 + 16) : 0) + 15) & ~(size_t)15;     // Tile::setup
    size_t body = c.lds_bytes;
    const size_t scratch = 2 * (size_t)g.waves * MX * 4 + 1024;   // prologue scratch aliases the windows
    if (body < scratch) body = scratch;
    return hdr + body;
}
inline size_t tile_lds_limit(int sym) { return (size_t)(160 * 1024) / CFGS[sym ? 1 : 0].bpc; }

namespace modes {

inline size_t data_size(int dtype) { return dtype == QDAS_F64 ? 16 : (dtype == QDAS_F32 ? 8 : 4); }

// ---- the library's environment switches, read ONCE per plan by the caller (qdas_api.hip read_switches); tests enumerate them
struct Switches {
    bool no_sym = false, no_fold = false, no_jit = false, no_mirror = false, no_mirq = false, no_narrow = false, no_role_swap = false, no_bpix = false,
         no_w64 = false, no_mirror_wpix = false, no_mirror_wpix32 = false, no_side_split = false, no_wide = false, no_fb2 = false, no_fb4 = false, no_fold16 = false;
    double sym_tol = -1.0;      // QDAS_SYM_TOL in [0, 0.5], else < 0
    int ksplit = 0;             // QDAS_KSPLIT in 1..8, else 0
    int ksplit_m = 0;           // QDAS_KSPLIT_M in 1..8 (transmit-block groups of a two-dimensional split), else 0
};

// ---- 1. what the descriptor asks for
struct Request {
    bool eligible = true;
    const char *why = "";       // why the fused kernel does not take it (when !eligible)
    bool mul = false, syn = false, bfm = false, cmap = false;
    int pix_arr = -1;           // first pixel-dependent apodization array (-1: none)
    bool pix_is_tx = false, pix_only = false, pix_fold = false, bpix_mode = false;
    bool is_pix[QDAS_MAX_APOD] = {};
    uint64_t npix = 0;
    bool dep_rx = false, dep_tx = false;
};

inline Request analyze_request(const qdas_desc &d, const Switches &sw) {
    Request r;
    const qdas_sizes &z = d.sz;
    const int dt = z.dtype;
    const uint64_t *cst = d.acstride, *ast = d.acstride + 6;
    // modes: 'DAS' (sum both apertures), and with fp32 data 'SYN' (keep the receive dimension: a plane per receiver)
    // and 'MUL' (keep the transmit dimension: the same kernel with the roles of the two apertures swapped)
    r.mul = (z.flag & QDAS_FLAG_KEEP_TX) && !(z.flag & QDAS_FLAG_KEEP_RX);
    r.syn = ((z.flag & QDAS_FLAG_KEEP_RX) && !(z.flag & QDAS_FLAG_KEEP_TX)) || r.mul;      // one output plane per STAGE element
    // 'BF' (both dimensions kept, fp32 data): the same stage loop, every pair's weighted sample stored to its own plane
    r.bfm = (z.flag & QDAS_FLAG_KEEP_TX) && (z.flag & QDAS_FLAG_KEEP_RX);
    r.eligible = (!r.syn && !r.bfm) || dt == QDAS_F32;
    r.why = "tiled kernel needs the 'DAS' mode (or fp32 data and 'SYN' / 'MUL' / 'BF')";
    auto no = [&](const char *w) { r.eligible = false; r.why = w; };
    // fp64 data (das_tile_impl.h "F64"): the plain sum with pixel-independent weights, scalar sound speed, no remodulation
    if (r.eligible && dt == QDAS_F64 && d.rx_apod_kind) no("tiled kernel, fp64 data: a generated receive apodization needs the generic kernel");
    // sound speed: a scalar, or a full per-pixel map (contiguous I1 x I2 x I3, no aperture dependence): the delay stays separable
    if (r.eligible && (cst[0] || cst[1] || cst[2] || cst[3] || cst[4])) {
        r.cmap = !cst[3] && !cst[4] && (cst[0] == 1 || z.I1 == 1) && (cst[1] == z.I1 || z.I2 == 1) && (cst[2] == z.I1 * z.I2 || z.I3 == 1);
        if (!r.cmap) no("tiled kernel needs a scalar sound speed or a full per-pixel map without aperture dependence");
        else if (dt == QDAS_F64) no("tiled kernel, fp64 data: a sound-speed map needs the generic kernel");
    }
    // apodization arrays: pixel-independent ones fold into an N x M table; ONE array may be a full I1 x I2 x I3 x [N] array
    // (contiguous pixel strides, no transmit dependence) -- it is applied per (pixel, receiver) by the tiled kernel
    // A full I1 x I2 x I3 x 1 x M array -- a weight per (pixel, TRANSMIT): scanline / multiline / parallelogram transmit apodization of
    // focused sequences -- is the same thing with the roles of the apertures swapped (stage element = transmit): 'DAS' only.
    // Several pixel-dependent arrays of ONE family -- receive side (I x N, I), transmit side (I x M, I) -- and arrays that broadcast over some
    // pixel dimension (a weight per depth and receiver: I1 x 1 x 1 x N) are multiplied into one plan-owned I x [N | M] array (apod_fold_kernel).
    // A receive-side and a transmit-side pixel array together, or an I x N x M array: generic kernel.
    const uint64_t I = z.I1 * z.I2 * z.I3;
    bool direct = true;
    for (uint64_t s = 0; s < z.S && r.eligible; ++s) {
        const uint64_t *a = &ast[6 * s];
        if (!a[0] && !a[1] && !a[2]) continue;
        const bool dn = a[3] && z.N > 1, dm = a[4] && z.M > 1;
        if (dn && dm) { no("tiled kernel: an apodization array over pixels x receivers x transmits needs the generic kernel"); break; }
        const bool pixstr = (a[0] == 1 || z.I1 == 1) && (a[1] == z.I1 || z.I2 == 1) && (a[2] == z.I1 * z.I2 || z.I3 == 1);
        if (!pixstr || (dn && a[3] != I) || (dm && a[4] != I)) direct = false;
        r.is_pix[s] = true; ++r.npix; r.dep_rx |= dn; r.dep_tx |= dm;
        if (r.pix_arr < 0) r.pix_arr = (int)s;
    }
    if (r.eligible && r.npix) {
        if (r.dep_rx && r.dep_tx) {
            // a transmit-side rule AND a receive-side mask (multiline x acceptance angle): the transmit is the stage element with its weight, the
            // receive-side product is a second weight per (pixel, block element) applied per pair -- launch configuration 14 (das_tile_impl.h BPIX):
            // fp32 data, real weights, plain 'DAS', no remodulation; pixel-independent arrays must belong to one aperture (they join that side's product)
            bool ok = dt == QDAS_F32 && d.apod_real && !r.syn && !r.bfm && d.fmod == 0.0 && I < (1ull << 30) && !sw.no_bpix;
            for (uint64_t s = 0; s < z.S && ok; ++s) {
                const uint64_t *a = &ast[6 * s];
                if (!r.is_pix[s] && a[3] && z.N > 1 && a[4] && z.M > 1) ok = false;
            }
            if (ok) { r.bpix_mode = true; r.pix_is_tx = true; for (uint64_t s = 0; s < z.S; ++s) r.is_pix[s] = true; r.npix = z.S; }
            else no("tiled kernel: pixel x receiver and pixel x transmit apodization arrays together run fused for fp32 data, real weights, 'DAS', no remodulation, no N x M array only");
        }
        else if (dt == QDAS_F64 && !(r.npix == 1 && direct && !r.dep_tx && d.fmod == 0.0 && !r.syn && !r.bfm && !r.mul && !sw.no_w64)) {
            // fp64 data: ONE pixel x receiver (or pixel-only) array, used in place, plain 'DAS', no remodulation (das_tile_impl.h TileCfg::W64)
            no("tiled kernel, fp64 data: only a single pixel x receiver / pixel-only apodization array without remodulation runs fused");
        }
        else if (r.dep_tx) {
            if ((!r.syn || r.mul) && !r.bfm) r.pix_is_tx = true;           // ('MUL': the transmit is the stage element anyway)
            else no("tiled kernel: a pixel x transmit apodization array with 'SYN' / 'BF' needs the generic kernel");
        } else if (!r.dep_rx && z.N > 1) {
            if (!r.bfm && !r.mul) r.pix_only = true;                     // a spatial weight / ROI mask
            else no("tiled kernel: a pixel-only apodization array needs the generic kernel");
        }
        r.pix_fold = r.eligible && (r.npix > 1 || !direct || r.bpix_mode);
    }
    if (r.eligible && d.rx_apod_kind && r.pix_arr >= 0) no("tiled kernel: a generated receive apodization and a pixel-dependent array need the generic kernel");
    if (r.eligible && r.bfm && (r.pix_arr >= 0 || d.rx_apod_kind)) no("tiled kernel: 'BF' with a pixel x receiver apodization needs the generic kernel");
    if (r.eligible && r.mul && ((r.pix_arr >= 0 && !r.pix_is_tx) || d.rx_apod_kind)) no("tiled kernel: 'MUL' with a pixel x receiver apodization needs the generic kernel");
    return r;
}

// ---- 2. symmetry modes, roles, strides
// Facts about the geometry that only the caller can establish.  `known` bits say which have been gathered: resolve_symmetry returns a NEED_* code
// for the first missing one it depends on, the caller gathers it and calls again -- the function itself stays pure.
struct Facts {
    // host comparison of Pv against Pr (fp32 geometry): every transmit shares the first one's t0; every transmit position equals "its" receiver's bit for
    // bit; else the largest distance |r_m - v_m| [m] (NaN-free: a NaN distance counts as "not reciprocal")
    bool recip_known = false, recip_one_t0 = false, recip_exact = false, recip_finite = true;
    double recip_dev = 0.0;
    double cinv0 = 0.0;                   // first entry of the sound-speed array (as float for fp32 / fp16 plans)
    // the plan's folded copy of a frame could be allocated
    bool fold_buf_known = false, fold_buf_ok = false;
    // mirror_symmetric() of the geometry at the tolerance resolve_symmetry asked for (exact: sym_tol < 0)
    bool mirror_known = false, mirror_yes = false;
    double mirror_bound = 0.0;
};
enum { NEED_NOTHING = 0, NEED_RECIP = 1, NEED_FOLD_BUF = 2, NEED_MIRROR = 3 };

struct Symmetry {
    bool eligible = true;
    const char *why = "";
    int sym = 0, rfold = 0, big = 0;
    bool prefolded = false;               // the plan takes folded frames (QDAS_PLAN_PREFOLDED honoured)
    bool prefolded_refused = false;       // ... was asked for and cannot be: QDAS_EUNSUPPORTED
    bool swap = false, mir = false, mslab = false;
    bool mir_asked = false;               // the mirror fact was needed (the caller ran the device check)
    double sym_tol = -1.0, recip_bound = 0.0, mirror_bound = 0.0;
    int tc_sym = 0, tc_narrow = 0, tc_fb = 1, tc_mirq = 0, tc_fold = 0;      // arguments of tile_config() for the plan's first configuration
    int pixw = 0, wtb = 0;
    uint64_t kN = 0, kM = 0;              // stage / block element counts (receivers / transmits, or swapped)
    int dtype = QDAS_F32;
    TileConfig tc() const { return tile_config(dtype, tc_sym, tc_narrow, tc_fb, tc_mirq, tc_fold); }
};

// returns NEED_NOTHING when `out` is final, else the fact to gather (out is then partial)
inline int resolve_symmetry(const qdas_desc &d, uint64_t i_count, const Request &rq, const Facts &f, const Switches &sw, Symmetry *out) {
    Symmetry s;
    const qdas_sizes &z = d.sz;
    const int dt = z.dtype;
    const uint64_t I = z.I1 * z.I2 * z.I3;
    s.dtype = dt;
    s.eligible = rq.eligible; s.why = rq.why;
    const bool jit_on = (d.plan_flags & QDAS_PLAN_JIT) && !sw.no_jit;
    // tolerance mode of the symmetry tests (QDAS_PLAN_APPROX_SYMMETRY; bound in SAMPLES, default 1e-5, QDAS_SYM_TOL overrides): < 0 = exact only
    if ((d.plan_flags & QDAS_PLAN_APPROX_SYMMETRY) && !rq.cmap) s.sym_tol = (sw.sym_tol >= 0.0 && sw.sym_tol <= 0.5) ? sw.sym_tol : 1.0e-5;
    const bool want_prefolded = (d.plan_flags & QDAS_PLAN_PREFOLDED) != 0;
    // reciprocal mode (das_tile_impl.h "SYM"): a full-synthetic-aperture acquisition whose transmit elements are the receive
    // elements and share one t0 has tau(n,m) == tau(m,n); detected from the geometry itself, bit-exactly.
    // RECIPROCITY FOLD (fold.hip, das_tile_impl.h TileCfg::FOLD; fp32 data): interpolation is linear in the data, so the two traces of an unordered
    // pair are ADDED once per frame -- one streaming pass over HBM, pixel-independent weights applied on the way -- and the fused kernel walks the
    // upper triangle of the folded frame: half the staging, gathers and multiply-accumulates.  QDAS_PLAN_NO_FOLD / QDAS_NO_FOLD=1: the reciprocal
    // mode as it was (both traces gathered, tap index and weights shared).
    if (rq.eligible && !rq.syn && !rq.bfm && (dt == QDAS_F32 || dt == QDAS_F16) && z.VS && z.DV && z.N == z.M && rq.pix_arr < 0 && !d.rx_apod_kind
        && !(d.plan_flags & QDAS_PLAN_NO_RECIPROCAL) && !sw.no_sym) {
        if (!f.recip_known) { *out = s; return NEED_RECIP; }
        s.sym = f.recip_one_t0 ? 1 : 0;
        if (s.sym && !f.recip_exact && (s.sym_tol < 0 || !f.recip_finite)) s.sym = 0;
        // |tau(n,m) - tau(m,n)| fs <= cinv fs (|r_n - v_n| + |r_m - v_m|) <= 2 cinv fs max|r - v|
        s.recip_bound = 2.0 * f.cinv0 * d.fs * (f.recip_exact ? 0.0 : f.recip_dev);
        if (s.sym && !(s.recip_bound <= (s.sym_tol < 0 ? 0.0 : s.sym_tol))) s.sym = 0;
        s.rfold = s.sym && dt == QDAS_F32 && z.N >= 2 && z.N <= 65535 && ((!(d.plan_flags & QDAS_PLAN_NO_FOLD) && !sw.no_fold) || want_prefolded)
                  && tile_lds_bytes(dt, 1, z.N, z.M, 0, 0, 0, 0, 1) <= tile_lds_limit(1);
        s.prefolded = want_prefolded && s.rfold;
        if (s.rfold && !want_prefolded) {                 // the plan's folded copy of a frame: without the memory for it, the plan simply does not fold
            if (!f.fold_buf_known) { *out = s; return NEED_FOLD_BUF; }
            if (!f.fold_buf_ok) s.rfold = 0;
        }
        if (s.sym && !s.rfold && (z.M % tile_config(dt, 1).mb != 0 || tile_lds_bytes(dt, 1, z.N, z.M) > tile_lds_limit(1))) s.sym = 0;
        // fp32 data without the fold (QDAS_PLAN_NO_FOLD): that reciprocal mode exists as a plan-specialised (hiprtc) build only -- its prebuilt
        // instantiations were pruned in round 4 --; without QDAS_PLAN_JIT such a plan runs the general kernels
        if (s.sym && !s.rfold && dt == QDAS_F32 && !jit_on) s.sym = 0;
    }
    if (want_prefolded && !s.rfold) { s.prefolded_refused = true; *out = s; return NEED_NOTHING; }
    // Roles of the two apertures (das_tile_impl.h): a stage = one STAGE element x a block of 32 BLOCK elements.  'DAS' / 'SYN': stage =
    // receiver, block = transmits; 'MUL': swapped.  The full sum may run either way, and runs swapped when that gives fewer, fuller
    // stages: plane-wave compounding with a handful of angles (N = 128, M = 9: 128 stages of 9 transmits -> 36 stages of 32 receivers).
    s.swap = rq.mul || rq.pix_is_tx;
    if (rq.eligible && !rq.syn && !rq.bfm && !s.sym && dt != QDAS_F64 && rq.pix_arr < 0 && !d.rx_apod_kind && !sw.no_role_swap) {
        const uint64_t mb = (uint64_t)tile_config(dt, 0).mb;
        if (4 * z.M * ((z.N + mb - 1) / mb) < 3 * z.N * ((z.M + mb - 1) / mb)) s.swap = true;     // (at least a quarter fewer stages: measured break-even, PW31 on 128 elements)
    }
    // Lateral-mirror mode (tile_params.h `mir`): a scan, an array and a sequence that are mirror-symmetric about x = 0 have
    // tau(pixel', N-1-n, M-1-m) == tau(pixel, n, m) bit for bit: tap index and interpolation weights serve a pixel and its mirror image.
    // Detected from the geometry itself (a fact); the whole image in one plan (or a mirror slab), plain 'DAS'; a per-pixel sound-speed map must be
    // mirror-symmetric itself, bit for bit (part of the same fact: a pixel and its image then share `cinv` as they share everything else) -- general mode
    // only: the reciprocal + mirror kernels keep the sound speed uniform (two registers their four window sets need), a reciprocal plan with a map stays reciprocal.
    // A reciprocal plan that is also mirror-symmetric runs FOUR window sets per stage (launch configurations 15 / 16; folded data: two, 17 / 18):
    // that kernel addresses the frame with one descriptor: frames below 2 GiB.
    // fp16 data: a pixel x receiver weight rides along (an I x N array, a pixel-only array, or a generated rule); fp32 data: as a hiprtc build only.
    s.mslab = (d.plan_flags & QDAS_PLAN_MIRROR_SLAB) != 0;
    const bool mir_plain = z.S == 0 && !d.rx_apod_kind;
    // pixel-independent weights only (folded into an N x M table): fine when the TABLE is mirror-symmetric (checked once it is built: ProbeChain)
    const bool mir_tab = z.S > 0 && rq.npix == 0 && !d.rx_apod_kind && dt != QDAS_F64;
    const bool jit_asked = jit_on && !sw.no_mirror_wpix32;
    const bool mir_wpix = (dt == QDAS_F16 || (dt == QDAS_F32 && jit_asked)) && !s.sym && !s.swap && !rq.bpix_mode && z.S == rq.npix
                          && ((rq.pix_arr >= 0 && !rq.pix_is_tx && !d.rx_apod_kind) || (rq.pix_arr < 0 && d.rx_apod_kind >= 1 && d.rx_apod_kind <= 4)) && !sw.no_mirror_wpix;
    if (rq.eligible && !rq.syn && !rq.bfm && (dt == QDAS_F32 || dt == QDAS_F16) && (mir_plain || mir_wpix || mir_tab) && !(rq.cmap && s.sym) && z.I3 == 1 && z.I2 >= 2
        && z.N >= 2 && ((d.i_begin == 0 && i_count == I) || s.mslab) && !(d.plan_flags & QDAS_PLAN_NO_MIRROR) && !sw.no_mirror
        && (!s.sym || ((uint64_t)z.T * z.N * z.M * data_size(dt) + 65536 < (1ull << 31) && (s.rfold || z.M % 16 == 0) && !sw.no_mirq
                       && tile_lds_bytes(dt, 1, z.N, z.M, 1, 0, (z.S > 0 && !s.rfold) ? 1 : 0, 1, s.rfold) <= tile_lds_limit(1)))) {
        s.mir_asked = true;
        if (!f.mirror_known) { *out = s; return NEED_MIRROR; }
        s.mir = f.mirror_yes;
        s.mirror_bound = f.mirror_bound;
    }
    // stage / block element counts of the kernel: receivers / transmits, or swapped
    s.kN = s.swap ? z.M : z.N; s.kM = s.swap ? z.N : z.M;
    s.tc_sym = s.sym; s.tc_narrow = 0; s.tc_fb = s.mir ? 2 : 1; s.tc_mirq = 0; s.tc_fold = s.rfold;
    s.pixw = (rq.pix_arr >= 0 || d.rx_apod_kind) ? 1 : 0;      // a pixel x receiver weight: the tile keeps a stage list (das_tile_impl.h plan_stages)
    s.wtb = (z.S > rq.npix && !s.rfold) ? 1 : 0;                // pixel-independent arrays: an N x M table staged per stage in LDS (folded data: applied by the fold pass)
    auto no = [&](const char *w) { s.eligible = false; s.why = w; };
    if (s.eligible && tile_lds_bytes(dt, s.sym, s.kN, s.kM, 0, s.pixw, s.wtb) > tile_lds_limit(s.sym)) no("tiled kernel: N + M too large for the LDS header");
    if (s.eligible && z.T < 8) no("tiled kernel needs T >= 8");
    // (a lane keeps its pixel's offset in the plan's slab in 32 bits, 0xffffffff = "not mine": das_tile_impl.h Tile::pofs)
    if (s.eligible && i_count >= 0xffffffffull) no("tiled kernel: more than 2^32 - 2 pixels in one plan (shard the image)");
    {   // LDS-DMA offsets are 32-bit and signed: inside one transmit block, N receivers + mb transmits + a window must stay below
        // 2^31 bytes.  The reciprocal kernel re-bases its descriptors whenever a running offset reaches 2^30 (its mirror traces
        // walk the whole frame), so there only one trace stride and the span of a block have to stay below 2^30.
        uint64_t strM = (z.flag & QDAS_FLAG_TPOSE) ? z.T : z.T * z.N;
        uint64_t strN = (z.flag & QDAS_FLAG_TPOSE) ? z.T * z.M : z.T;
        if (s.swap) std::swap(strM, strN);
        const uint64_t slack = 65536;
        const uint64_t smax = strM > strN ? strM : strN;
        if (s.sym && (uint64_t)(tile_config(dt, 1, 0, 1, 0, s.rfold).mb + 1) * smax * data_size(dt) + slack >= (1ull << 30)) {
            s.sym = 0; s.rfold = 0; s.prefolded = false;
            if (want_prefolded) { s.prefolded_refused = true; *out = s; return NEED_NOTHING; }
            s.tc_sym = 0; s.tc_narrow = 0; s.tc_fb = s.mir ? 2 : 1; s.tc_mirq = 0; s.tc_fold = 0;     // (a mirror-symmetric plan keeps the two-window-set configuration of the general mode)
            s.wtb = (z.S > rq.npix) ? 1 : 0;
            if (s.eligible && tile_lds_bytes(dt, 0, z.N, z.M, 0, s.pixw, s.wtb) > tile_lds_limit(0)) no("tiled kernel: N + M too large for the LDS header");
        }
        if (s.eligible && !s.sym && (s.kN * strN + (uint64_t)s.tc().mb * strM) * data_size(dt) + slack >= (1ull << 31)) {
            // fp32, one frame per launch: the re-basing instantiation of the general kernel (launch configuration 9)
            // (the re-basing instantiation is a plain general-mode kernel: no lateral-mirror mode there -- the mirrored window set's offsets are
            //  32-bit offsets of the same magnitude; launch_tile rejects mir && big)
            if (dt == QDAS_F32 && !rq.bfm && ((uint64_t)tile_config(dt, 0).mb * strM + strN) * data_size(dt) + slack < (1ull << 30)) {
                s.big = 1; s.mir = false; s.tc_sym = 0; s.tc_narrow = 0; s.tc_fb = 1; s.tc_mirq = 0; s.tc_fold = 0;
            } else no("tiled kernel: trace strides too large for 32-bit DMA offsets (transposed data of more than 2 GiB)");
        }
    }
    *out = s;
    return NEED_NOTHING;
}

// ---- 3. the probe chain: which launch configurations have their window fit probed, and what the plan gives up when tiles do not fit.
// One step = one choose_tile_shape() of the caller (four footprints probed on the device); its outcome = "some footprint has no misfit tile".
struct ProbeState {
    int dtype = QDAS_F32, sym = 0, rfold = 0, mir = 0 /* 0 | 1 | 2 (slab) */, narrow = 0;
    bool bpix = false;
    int tc_sym = 0, tc_narrow = 0, tc_fb = 1, tc_mirq = 0, tc_fold = 0;
    TileConfig tc() const { return tile_config(dtype, tc_sym, tc_narrow, tc_fb, tc_mirq, tc_fold); }
    void set_tc(int s, int n = 0, int fb = 1, int mq = 0, int fo = 0) { tc_sym = s; tc_narrow = n; tc_fb = fb; tc_mirq = mq; tc_fold = fo; }
};

// probe(state) -> true when some tile footprint has no misfit tile (pl->no_fallback).  Returns the final outcome of the last probe.
// table_mirror_symmetric: the folded N x M weight table equals its own mirror image (only asked when it matters).
template <class Probe>
inline bool run_probe_chain(ProbeState &t, const Switches &sw, bool has_table, bool table_mirror_symmetric, Probe &&probe, int *err) {
    const int dt = t.dtype;
    *err = 0;
    auto run = [&]() -> bool { bool ok = false; const int e = probe(t, &ok); if (e) *err = e; return ok; };
    if (t.mir && has_table && !t.rfold && !table_mirror_symmetric) { t.mir = 0; t.set_tc(t.sym); }      // (reciprocal plans: the narrow configuration is chosen just below)
    bool fit;
    if (t.rfold) {
        // folded data: with the lateral-mirror mode two window sets -- 32 x 128 samples when every tile of some footprint fits them (launch
        // configuration 17), else 16 x 192 (18) --; when that leaves misfit tiles too (a misfit tile is redone by the generic kernel, which knows
        // nothing of mirror images) or without the mode: one set of 32 x 192 samples (19), misfit tiles to the generic kernel as ever
        fit = false;
        if (t.mir) {
            t.narrow = sw.no_narrow ? 0 : 1;
            t.set_tc(1, t.narrow, 1, 1, 1);
            fit = run(); if (*err) return false;
            if (t.narrow && !fit) { t.narrow = 0; t.set_tc(1, 0, 1, 1, 1); fit = run(); if (*err) return false; }
            if (!fit) t.mir = 0;
        }
        if (!t.mir) { t.narrow = 0; t.set_tc(1, 0, 1, 0, 1); fit = run(); if (*err) return false; }
        return fit;
    }
    // reciprocal mode: first the 128-sample-window configuration (less staging traffic); it is kept only if some footprint
    // has no misfit tile at all -- otherwise the 192-sample configuration
    t.narrow = (t.sym && dt == QDAS_F32 && !sw.no_narrow) ? 1 : 0;
    if (t.sym && t.mir && dt == QDAS_F32 && !t.narrow) t.mir = 0;        // (the four-set configuration has 128-sample windows)
    if (t.narrow) t.set_tc(1, 1, 1, t.mir ? 1 : 0);
    else if (t.sym && t.mir) t.set_tc(1, 0, 1, 1);
    if (t.bpix) { t.narrow = 2; t.set_tc(0, 2); }      // (the configuration that applies a per-pair pixel weight)
    fit = run(); if (*err) return false;
    if (t.mir && !fit) {                 // (a misfit tile is redone by the generic kernel, which knows nothing of mirror images;
        t.mir = 0;                       //  reciprocal plans: the four-set configuration exists with the narrow windows only)
        if (t.narrow == 1) t.set_tc(1, 1); else t.set_tc(t.sym);
        fit = run(); if (*err) return false;
    }
    if (t.narrow == 1 && !fit) { t.narrow = 0; t.set_tc(1, 0); fit = run(); if (*err) return false; }
    return fit;
}

// Focused transmits whose focal planes cut through the image: second attempt with every transmit listed twice, once per side of its plane
inline bool side_split_applicable(const qdas_desc &d, const Request &rq, const Symmetry &s, bool no_fallback, int txkind, bool has_wtab_in_kernel, const Switches &sw) {
    const qdas_sizes &z = d.sz;
    const int dt = z.dtype;
    return !no_fallback && txkind == 1 && !rq.syn && !rq.bfm && !s.sym && !s.big && (rq.pix_arr < 0 || ((rq.pix_is_tx || rq.pix_only) && !(d.fmod != 0.0 && has_wtab_in_kernel)))
           && !d.rx_apod_kind && (dt == QDAS_F32 || dt == QDAS_F16) && z.M < (1u << 15) && tile_lds_bytes(dt, 0, 2 * z.M, z.N, 0, 1, has_wtab_in_kernel ? 1 : 0) <= tile_lds_limit(0) && !sw.no_side_split;
}

// Tiles that still do not fit: fp32 plans try the 384-sample windows of launch configuration 14 (16 transmits per stage, same LDS image)
inline bool wide_applicable(int dtype, bool no_fallback, bool bfm, int big, int narrow, bool prefolded, uint64_t tN, uint64_t tM, uint64_t strN, uint64_t strM,
                            bool stage_list, bool table_in_kernel, const Switches &sw) {
    return !no_fallback && dtype == QDAS_F32 && !bfm && !big && narrow != 2 && !sw.no_wide && !prefolded
           && tile_lds_bytes(dtype, 0, tN, tM, 2, stage_list ? 1 : 0, table_in_kernel ? 1 : 0) <= tile_lds_limit(0)
           && (tN * strN + (uint64_t)tile_config(dtype, 0, 2).mb * strM) * data_size(dtype) + 65536 < (1ull << 31);
}

// ---- 4. workgroups per tile: too few tiles for the GPU (a pixel slab of a multi-GPU job, a small image) -> several workgroups per tile, each summing a
// slice of the aperture until every CU has a workgroup.  Plans with a pixel x stage-element weight: the tiles' stage lists differ in length by the mask, so
// the longest tile sets the kernel time unless there are many more workgroups than CUs: up to 8 per CU, each taking every ks-th receiver, as long as a
// workgroup keeps at least 16 candidate stage elements.
inline unsigned choose_ksplit(unsigned ntiles, unsigned cus, uint64_t M, int mb, bool sym, uint64_t kN_eff, bool stage_list, bool syn, const Switches &sw) {
    const uint64_t nmb = (M + (uint64_t)mb - 1) / (uint64_t)mb;
    const unsigned cap = (unsigned)std::min<uint64_t>(8, sym ? nmb : kN_eff);
    unsigned ks = 1;
    while (ks * 2 <= cap && (uint64_t)ntiles * ks < (uint64_t)cus) ks *= 2;
    if (stage_list && !syn)
        while (ks * 2 <= cap && (uint64_t)ntiles * ks < 8ull * cus && kN_eff / (ks * 2) >= 16) ks *= 2;
    if (sw.ksplit >= 1 && sw.ksplit <= 8 && (unsigned)sw.ksplit <= cap) ks = (unsigned)sw.ksplit;
    return ks;
}

// Plans with a stage list (pixel x stage-element weights): the split is two-dimensional (tile_params.h ksplit_m) -- first over the TRANSMIT BLOCKS, so that a
// workgroup refreshes its block residuals once or twice instead of once per handful of stages, then over interleaved classes of stage elements.  Returns the
// number of transmit-block groups and adjusts *ksplit to groups x classes (<= 8).  QDAS_KSPLIT (a forced split) keeps the one-dimensional split unless
// QDAS_KSPLIT_M says otherwise.
inline unsigned choose_ksplit_m(uint32_t *ksplit, uint64_t tM, int mb, bool sym, bool stage_list, bool syn, int dtype, const Switches &sw) {
    if (!stage_list || syn || sym || dtype == QDAS_F64 || *ksplit <= 1) return 1;
    const unsigned nblk = (unsigned)((tM + (uint64_t)mb - 1) / (uint64_t)mb);
    if (sw.ksplit_m >= 1) {                              // forced: must divide the split
        const unsigned f = (unsigned)sw.ksplit_m;
        return (f <= nblk && *ksplit % f == 0) ? f : 1;
    }
    if (sw.ksplit >= 1 || nblk < 2) return 1;
    const unsigned sm = nblk < 8u ? nblk : 8u;           // a group per transmit block (at most 8) ...
    unsigned sn = *ksplit / sm;                          // ... times as many classes of stage elements as the chosen split leaves room for
    if (sn < 1) sn = 1;
    *ksplit = sm * sn;
    return sm;
}

// ---- 5. frames per launch of a stream
struct StreamModes { bool fb2_ok, fold2_ok, fb4_off; };
struct PlanShape {          // what the finished plan looks like (the fields launch_legal and stream_modes read)
    int dtype = QDAS_F32;
    bool tiled = false;
    int sym = 0, fold = 0, mir = 0, narrow = 0, big = 0, bf = 0, syn = 0, stage_shift = 0;
    bool has_apix = false, has_wtab = false, has_bpix = false;
    int gen_kind = 0;
    double fmod = 0.0;
    uint64_t N = 0, M = 0;
    bool mem_device = true;
};
inline StreamModes stream_modes(const PlanShape &p, uint64_t zN, uint64_t zM, const Switches &sw) {
    StreamModes m;
    const int dt = p.dtype;
    m.fb2_ok = p.tiled && dt != QDAS_F64 && !p.mir && !p.stage_shift && p.narrow != 2 && !(dt == QDAS_F32 && (p.has_apix || p.gen_kind)) && !p.bf && !p.sym && !p.big && p.mem_device && !sw.no_fb2;
    // folded data: TWO frames per launch (launch configurations 20 / 21 -- tap index and weights serve two folded traces of two frames, in mirror mode four);
    // the mirror mode needs the 128-sample windows for it (four window sets), and the plan a second folded copy of a frame
    m.fold2_ok = p.tiled && p.fold && (!p.mir || p.narrow) && !p.syn && p.mem_device && !sw.no_fb2
                 && tile_lds_bytes(dt, 1, zN, zM, p.narrow, 0, 0, p.mir ? 1 : 0, 1, 2) <= tile_lds_limit(1);
    // four frames per launch: not for fp32 plans with remodulation, a weight table or a pixel x receiver weight (das_tile_impl.h launch_tile_i)
    m.fb4_off = sw.no_fb4 || (dt == QDAS_F32 && p.tiled && (p.fmod != 0.0 || p.has_wtab || p.has_apix || p.gen_kind));
    return m;
}

// ---- 6. the launcher's admission rules (das_tile.hip launch_tile): nullptr = legal, else the rule that refuses
struct LaunchShape {
    int dtype = QDAS_F32;
    int sym = 0, fold = 0, mir = 0, narrow_raw = 0, big = 0, bf = 0, syn = 0, stage_shift = 0, probe = 0, nfr = 1;
    bool lut = false, has_wtab = false, has_apix = false, has_bpix = false, has_part = false, jit = false;
    int gen_kind = 0;
    double fmod = 0.0;
    uint32_t act_bytes = 0, ksplit = 1;
    uint64_t N = 0, M = 0;
};
struct LaunchChoice { int narrow = 0, fold = 0, mirq = 0, nf = 1, nfr = 1; bool probe_f32sym = false; size_t lds = 0; int cfg = 0; };

inline const char *launch_legal(const LaunchShape &P, LaunchChoice *c) {
    const int dtype = P.dtype, sym = P.sym ? 1 : 0;
    LaunchChoice ch;
    if (sym && dtype != 1 && dtype != 2) return "reciprocal mode: fp32 / fp16 data only";
    if (dtype == 0) {                                    // fp64 data: one frame, one workgroup per tile, plain 'DAS' sum, prebuilt kernels
        if (P.lut || P.bf || P.syn || P.big || P.nfr > 1 || (P.jit && P.probe) || (!P.probe && (P.ksplit < 1 || (P.ksplit > 1 && !P.has_part)))) return "fp64 data: plain 'DAS', one frame per launch";
        ch.lds = tile_lds_bytes(0, 0, P.N, P.M, 0);
        if (ch.lds > tile_lds_limit(0)) return "fp64 data: LDS image too large";
        ch.cfg = 13;
        if (c) *c = ch;
        return nullptr;
    }
    const int narrow = (sym && dtype == 1 && P.narrow_raw) ? 1 : (!sym && dtype == 1 && P.narrow_raw == 2) ? 2 : 0;     // window variant (das_tile_cfg.h)
    if (narrow == 2 && (P.lut || P.bf || P.big || (!P.probe && P.nfr > 1))) return "384-sample windows: one frame, no table-driven delays / 'BF' / re-basing";
    if (P.has_bpix && (narrow != 2 || P.has_wtab || P.fmod != 0.0 || P.syn || sym)) return "per-pair pixel weights: the 384-sample configuration without table / remodulation only";
    if (P.act_bytes != 0 && P.act_bytes != 8 * (P.N + 1)) return "stage list: 8 (N + 1) bytes";
    // fp32 reciprocal plans WITHOUT the fold (QDAS_PLAN_NO_FOLD) exist as hiprtc builds only; the plan-time probes of every fp32 reciprocal plan run the
    // probe kernels of the folded configurations (the same prologue; 128- or 192-sample windows)
    const int fold = (sym && P.fold && dtype == 1) ? 1 : 0;   // reciprocity-folded data (launch configurations 17 ... 21)
    if (P.fold && (!fold || P.has_wtab)) return "folded data: reciprocal fp32 plans, the table belongs to the fold pass";
    const bool probe_f32sym = P.probe && sym && dtype == 1;
    if (sym && dtype == 1 && !fold && !P.jit && !P.probe) return "unfolded fp32 reciprocal mode: hiprtc builds only";
    const int mirq = (sym && P.mir && !P.probe) ? 1 : 0;   // reciprocal + lateral-mirror mode: four window sets (launch configurations 15 / 16; folded data: two)
    const size_t lds = probe_f32sym ? tile_lds_bytes(dtype, 1, P.N, P.M, narrow, 0, 0, narrow ? 1 : 0, 1)
                                    : tile_lds_bytes(dtype, sym, P.N, P.M, narrow, P.act_bytes ? 1 : 0, P.has_wtab ? 1 : 0, mirq, fold, (fold && P.nfr == 2) ? 2 : 1);   // (two frames of folded data: launch configurations 20 / 21)
    if (lds > tile_lds_limit(sym)) return "LDS image too large";
    if (!P.probe && (P.ksplit < 1 || (P.ksplit > 1 && !P.has_part && !P.bf))) return "split aperture without partial images";
    // frames per launch; lateral-mirror plans (one frame) run the two-window-set instantiations as well
    const int nfr = P.probe ? 1 : (P.nfr > 1 ? P.nfr : 1);
    if (P.mir && !P.probe && ((nfr != 1 && !(fold && nfr == 2)) || P.big || P.bf || (P.lut && !P.jit) || P.syn || ((P.has_apix || P.gen_kind) && ((dtype != 2 && !P.jit) || sym || P.stage_shift || P.has_bpix))
                              || (!sym && narrow) || (sym && dtype == 1 && !narrow && !fold)))
        return "lateral-mirror mode: one frame (folded data: two), no re-basing / 'BF' / 'SYN' / tables of delays, pixel weights for fp16 data or hiprtc builds, narrow windows when reciprocal";
    const int nf = (P.mir && !P.probe && !sym) ? 2 : nfr;
    if ((nf != 1 && nf != 2 && nf != 4) || (nf > 1 && ((sym && !(fold && nf == 2)) || P.big))) return "frames per launch: 1, 2 or 4; reciprocal plans one (folded: two); no re-basing";     // (folded data: two frames may share a launch)
    // (table-driven delays in lateral-mirror mode -- symmetric tables, qdas_das_lut checks them per call -- exist as a hiprtc build only: round 6)
    if (P.lut && (sym || (nf != 1 && !(P.mir && P.jit && nf == 2 && dtype == 1 && !P.syn)) || (dtype != 1 && dtype != 2) || (P.syn && dtype != 1))) return "table-driven delays: general mode, one frame";
    if (P.jit && (P.probe || nfr != 1 || (P.lut && !P.mir) || P.bf)) return "hiprtc builds: one frame, no probe / 'BF'; table-driven delays in lateral-mirror mode only";
    if (P.bf && (sym || nf != 1 || dtype != 1 || P.lut || P.big || P.has_apix || P.gen_kind)) return "'BF': fp32 general mode without pixel weights";
    ch.narrow = narrow; ch.fold = fold; ch.mirq = mirq; ch.nf = nf; ch.nfr = nfr; ch.probe_f32sym = probe_f32sym; ch.lds = lds;
    ch.cfg = cfg_index(dtype, sym, 1, narrow, mirq, fold);
    if (c) *c = ch;
    return nullptr;
}

// ---- the MODEL of a built plan: what the launcher will be handed, derived from the decisions alone.  qdas_plan_create checks its parameter block against
// this after the build (a mismatch is an internal error: the model and the code have drifted), and the GPU-less enumeration runs launch_legal on it.
struct BuildOutcome {
    int mir = 0, narrow = 0;              // the probe chain's final state
    bool side_split = false, wide = false;
    unsigned ksplit = 1;
};
inline LaunchShape derive_launch_shape(const qdas_desc &d, const Request &rq, const Symmetry &sy, const BuildOutcome &o) {
    LaunchShape L;
    const qdas_sizes &z = d.sz;
    const int dt = z.dtype;
    L.dtype = dt; L.sym = sy.sym; L.fold = sy.rfold; L.big = sy.big; L.mir = o.mir; L.narrow_raw = o.narrow;
    L.bf = rq.bfm ? 1 : 0; L.syn = rq.syn ? 1 : 0;
    uint64_t tN = sy.swap ? z.M : z.N, tM = sy.swap ? z.N : z.M;
    const bool apix = rq.bpix_mode || rq.pix_fold || rq.pix_arr >= 0;
    const bool table = z.S > rq.npix;
    L.has_apix = apix; L.has_bpix = rq.bpix_mode; L.gen_kind = d.rx_apod_kind;
    L.has_wtab = table && !L.fold;
    if (o.side_split) { tN = 2 * z.M; tM = z.N; L.stage_shift = 1; L.gen_kind = apix ? 6 : 5; }
    L.act_bytes = ((apix || L.gen_kind) && dt != QDAS_F64) ? (uint32_t)(8 * (tN + 1)) : 0u;
    if (o.wide) { L.narrow_raw = 2; L.sym = 0; if (L.fold) { L.fold = 0; L.has_wtab = table; } }
    L.N = tN; L.M = tM; L.ksplit = o.ksplit; L.has_part = o.ksplit > 1 && !L.bf;
    L.fmod = dt == QDAS_F64 ? d.fmod : (double)(float)d.fmod;
    return L;
}
// the first field in which two launch shapes differ (nullptr: none); `jit`, `probe`, `nfr` are the launch's, not the plan's
inline const char *shape_mismatch(const LaunchShape &a, const LaunchShape &b) {
#define QDAS_CMP(f) if (a.f != b.f) return #f;
    QDAS_CMP(dtype) QDAS_CMP(sym) QDAS_CMP(fold) QDAS_CMP(mir) QDAS_CMP(narrow_raw) QDAS_CMP(big) QDAS_CMP(bf) QDAS_CMP(syn) QDAS_CMP(stage_shift) QDAS_CMP(lut)
    QDAS_CMP(has_wtab) QDAS_CMP(has_apix) QDAS_CMP(has_bpix) QDAS_CMP(has_part) QDAS_CMP(gen_kind) QDAS_CMP(fmod) QDAS_CMP(act_bytes) QDAS_CMP(ksplit) QDAS_CMP(N) QDAS_CMP(M)
#undef QDAS_CMP
    return nullptr;
}

}  // namespace modes
}  // namespace qdas
