// enumerate_modes.cpp -- GPU-less table test of the plan's mode resolution (qups_amd/csrc/plan_modes.h).  TEST INFRASTRUCTURE.
//
//   g++ -std=c++17 -O1 -I qups_amd/csrc tests/modes/enumerate_modes.cpp -o enumerate_modes && ./enumerate_modes [descriptors]
//
// Draws descriptors with the distributions of the GPU fuzz (tests/test_gpu_fuzz.py _draw: sequences, aperture sizes, apodization shapes, modes, shards,
// precisions) plus what the fuzz cannot afford on a device -- frames beyond 2 GiB, transposed data, 256 x 256 apertures, every plan flag and environment
// switch -- and, per descriptor, every combination of the FACTS qdas_plan_create would have to gather on a device: reciprocal geometry (exact / within a
// tolerance / not), fold buffer allocated or not, mirror-symmetric or not, weight table mirror-symmetric or not, every outcome of the window-fit probes
// (each probe of the chain fits or does not), side split / wide windows better or not.  Every draw must
//   * end in a launch configuration that launch_legal() -- the launcher's own admission rules -- accepts, for one frame and for the frame-sharing
//     launches stream_modes() admits, or
//   * name the reason the fused kernel does not take it (Request / Symmetry::why), or refuse QDAS_PLAN_PREFOLDED,
// and must satisfy the invariants the modes rely on (no mirror mode on the re-basing configuration, fold only with the reciprocal mode, ...).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "plan_modes.h"

using namespace qdas;
using namespace qdas::modes;

struct Rng {                                            // splitmix64: the draws are the same on every host
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 12345) {}
    uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
    uint64_t below(uint64_t n) { return next() % n; }
    bool one_in(uint64_t n) { return below(n) == 0; }
    template <class T, size_t K> T pick(const T (&a)[K]) { return a[below(K)]; }
};

struct Drawn {
    qdas_desc d;
    uint64_t acs[6 * (1 + QDAS_MAX_APOD)];
    Switches sw;
    uint64_t i_count;
    std::string text;
};

static void draw(uint64_t seed, Drawn &D) {
    Rng r(seed);
    memset(&D.d, 0, sizeof D.d);
    memset(D.acs, 0, sizeof D.acs);
    D.sw = Switches();
    qdas_desc &d = D.d;
    qdas_sizes &z = d.sz;
    static const int Ns[] = {1, 2, 3, 7, 16, 17, 32, 33, 48, 64, 128, 256, 320}, Ms[] = {1, 2, 5, 9, 16, 31, 32, 33, 40, 96, 128, 256, 520};
    static const uint64_t Ts[] = {7, 8, 300, 2048, 2816, 4096, 10240, 65536};
    const int seq = (int)r.below(4);                    // FSA | PW | DV | FC
    z.VS = seq != 1; z.DV = (seq == 0 || seq == 2);
    z.N = (uint64_t)r.pick(Ns); z.M = (uint64_t)r.pick(Ms);
    if (seq == 0 || r.one_in(6)) z.M = z.N;             // full synthetic aperture: M == N (the reciprocal-mode candidates)
    z.T = r.pick(Ts);
    z.I1 = 1 + r.below(1100); z.I2 = 1 + r.below(1100); z.I3 = r.one_in(8) ? 1 + r.below(6) : 1;
    if (r.one_in(40)) { z.I1 = 70000; z.I2 = 70000; }   // > 2^32 pixels
    z.dtype = (int)(r.below(8) == 0 ? QDAS_F64 : r.below(3) == 0 ? QDAS_F16 : QDAS_F32);
    const int interp = (int)r.below(6) % 6;
    const int fun = (int)r.below(8);                    // 0..3 DAS, 4 SYN, 5 MUL, 6 BF, 7 DAS
    z.flag = (interp == 4 ? 1 : interp) | (fun == 4 ? QDAS_FLAG_KEEP_RX : fun == 5 ? QDAS_FLAG_KEEP_TX : fun == 6 ? (QDAS_FLAG_KEEP_RX | QDAS_FLAG_KEEP_TX) : 0) | (r.one_in(3) ? QDAS_FLAG_TPOSE : 0);
    d.fs = 20e6; d.fmod = r.one_in(3) ? 2.5e6 : 0.0;
    d.apod_real = r.one_in(2);
    d.acstride = D.acs;
    d.mem = r.one_in(5) ? QDAS_MEM_HOST : QDAS_MEM_DEVICE;
    d.kernel = r.one_in(6) ? QDAS_KERNEL_TILED : QDAS_KERNEL_AUTO;
    const uint64_t I = z.I1 * z.I2 * z.I3;
    // sound speed: scalar | full map | a map with aperture dependence | a broadcast map
    switch (r.below(8)) {
        case 0: D.acs[0] = 1; D.acs[1] = z.I1; D.acs[2] = z.I1 * z.I2; break;
        case 1: D.acs[0] = 1; D.acs[1] = z.I1; D.acs[2] = z.I1 * z.I2; D.acs[3] = I; break;
        case 2: D.acs[0] = 1; break;
        default: break;
    }
    // apodization arrays: any of the singleton patterns the reference sweeps (test/USTest.m:333-337), in element strides
    const uint64_t S = r.below(4) == 0 ? 0 : r.below(QDAS_MAX_APOD + 1);
    z.S = S;
    uint64_t off = 0;
    for (uint64_t s = 0; s < S; ++s) {
        uint64_t *a = &D.acs[6 * (1 + s)];
        const bool p1 = r.one_in(3), p2 = r.one_in(3), p3 = r.one_in(4), pn = r.one_in(2), pm = r.one_in(3);
        uint64_t st = 1;
        if (p1) { a[0] = st; st *= z.I1; }
        if (p2) { a[1] = st; st *= z.I2; }
        if (p3) { a[2] = st; st *= z.I3; }
        if (pn) { a[3] = st; st *= z.N; }
        if (pm) { a[4] = st; st *= z.M; }
        a[5] = off; off += st;
    }
    if (r.one_in(5)) d.rx_apod_kind = 1 + (int)r.below(4);
    // slab
    d.i_begin = 0; D.i_count = I;
    if (r.one_in(4) && I >= 3) { d.i_begin = I / 3; D.i_count = I - I / 3 - I / 4; }
    d.i_count = D.i_count == I ? 0 : D.i_count;
    int pf = 0;
    if (r.one_in(3)) pf |= QDAS_PLAN_JIT;
    if (r.one_in(8)) pf |= QDAS_PLAN_NO_RECIPROCAL;
    if (r.one_in(8)) pf |= QDAS_PLAN_NO_MIRROR;
    if (r.one_in(8)) pf |= QDAS_PLAN_NO_FOLD;
    if (r.one_in(6)) pf |= QDAS_PLAN_APPROX_SYMMETRY;
    if (r.one_in(12)) pf |= QDAS_PLAN_PREFOLDED;
    if (r.one_in(10) && z.I3 == 1 && z.I2 % 2 == 0 && z.I2 >= 4) {        // a mirror slab: whole columns of the first half
        pf |= QDAS_PLAN_MIRROR_SLAB;
        d.i_begin = z.I1 * (z.I2 / 8); D.i_count = z.I1 * (z.I2 / 4 > 0 ? z.I2 / 4 : 1); d.i_count = D.i_count;
    }
    d.plan_flags = pf;
    Switches &sw = D.sw;
    if (r.one_in(3)) {                                  // a third of the draws with some switches thrown
        sw.no_sym = r.one_in(6); sw.no_fold = r.one_in(6); sw.no_jit = r.one_in(6); sw.no_mirror = r.one_in(6); sw.no_mirq = r.one_in(6); sw.no_narrow = r.one_in(6);
        sw.no_role_swap = r.one_in(6); sw.no_bpix = r.one_in(6); sw.no_w64 = r.one_in(6); sw.no_mirror_wpix = r.one_in(6); sw.no_mirror_wpix32 = r.one_in(6);
        sw.no_side_split = r.one_in(6); sw.no_wide = r.one_in(6); sw.no_fb2 = r.one_in(6); sw.no_fb4 = r.one_in(6);
        if (r.one_in(4)) sw.ksplit = 1 + (int)r.below(8);
        if (r.one_in(6)) sw.sym_tol = 0.05;
    }
    char buf[512];
    snprintf(buf, sizeof buf, "seed %llu: seq %d dtype %d T %llu N %llu M %llu I %llu x %llu x %llu flag 0x%x S %llu gen %d fmod %g pf 0x%x slab %llu+%llu",
             (unsigned long long)seed, seq, z.dtype, (unsigned long long)z.T, (unsigned long long)z.N, (unsigned long long)z.M, (unsigned long long)z.I1, (unsigned long long)z.I2,
             (unsigned long long)z.I3, z.flag, (unsigned long long)z.S, d.rx_apod_kind, d.fmod, pf, (unsigned long long)d.i_begin, (unsigned long long)D.i_count);
    D.text = buf;
}

static long g_fail = 0;
#define EXPECT(c, ...) do { if (!(c)) { if (g_fail++ < 20) { printf("FAILED %s:%d  %s\n   ", __FILE__, __LINE__, #c); printf(__VA_ARGS__); printf("\n"); } } } while (0)

int main(int argc, char **argv) {
    const uint64_t ndesc = argc > 1 ? strtoull(argv[1], nullptr, 10) : 12000;
    std::map<std::string, long> reasons;
    std::map<int, long> cfgs;
    long resolved = 0, named = 0, refused = 0, walks = 0;
    for (uint64_t seed = 0; seed < ndesc; ++seed) {
        Drawn D;
        draw(seed, D);
        const qdas_desc &d = D.d;
        const qdas_sizes &z = d.sz;
        const int dt = z.dtype;
        const Request rq = analyze_request(d, D.sw);
        // ---- invariants of the request
        EXPECT(!(rq.mul && rq.bfm) && (!rq.mul || rq.syn), "%s", D.text.c_str());
        EXPECT(rq.eligible || (rq.why && *rq.why), "%s", D.text.c_str());
        EXPECT(!rq.bpix_mode || (dt == QDAS_F32 && d.apod_real && d.fmod == 0.0), "%s", D.text.c_str());
        EXPECT(!(rq.eligible && dt != QDAS_F32 && (rq.syn || rq.bfm)), "%s", D.text.c_str());
        // ---- every combination of geometry facts
        for (int rf = 0; rf < 4; ++rf) for (int fb = 0; fb < 2; ++fb) for (int mf = 0; mf < 2; ++mf) {
            Facts f;
            f.cinv0 = 1.0 / 1540.0;
            Symmetry sy;
            bool asked_recip = false, asked_fold = false, asked_mirror = false;
            int guard = 0;
            for (;;) {
                const int need = resolve_symmetry(d, D.i_count, rq, f, D.sw, &sy);
                if (need == NEED_NOTHING) break;
                EXPECT(++guard < 6, "resolve_symmetry does not converge: %s", D.text.c_str());
                if (guard >= 6) break;
                if (need == NEED_RECIP) { asked_recip = true; f.recip_known = true; f.recip_one_t0 = rf != 3; f.recip_exact = rf == 0; f.recip_finite = true; f.recip_dev = rf == 1 ? 1e-10 : 1e-6; }
                else if (need == NEED_FOLD_BUF) { asked_fold = true; f.fold_buf_known = true; f.fold_buf_ok = fb == 0; }
                else if (need == NEED_MIRROR) { asked_mirror = true; f.mirror_known = true; f.mirror_yes = mf == 0; f.mirror_bound = mf == 0 ? 0.0 : 1.0; }
            }
            if ((!asked_recip && rf) || (!asked_fold && fb) || (!asked_mirror && mf)) continue;      // (a fact nobody asked for: the same walk as its first value)
            ++walks;
            // ---- invariants of the symmetry decision
            EXPECT(!sy.rfold || sy.sym, "fold without the reciprocal mode: %s", D.text.c_str());
            EXPECT(!sy.prefolded || sy.rfold, "prefolded without fold: %s", D.text.c_str());
            EXPECT(!sy.sym || ((dt == QDAS_F32 || dt == QDAS_F16) && z.N == z.M && z.VS && z.DV && !rq.syn && !rq.bfm), "reciprocal mode on %s", D.text.c_str());
            EXPECT(!sy.sym || rf != 3, "reciprocal mode with several start times: %s", D.text.c_str());
            EXPECT(!(sy.sym && rf >= 1) || (sy.sym_tol >= 0 && sy.recip_bound <= sy.sym_tol && (d.plan_flags & QDAS_PLAN_APPROX_SYMMETRY)), "reciprocal mode beyond its tolerance: %s", D.text.c_str());
            EXPECT(!(sy.sym && rf == 0) || sy.recip_bound == 0.0, "%s", D.text.c_str());
            EXPECT(!(sy.mir && sy.big), "mirror mode on the re-basing configuration: %s", D.text.c_str());
            EXPECT(!sy.mir || (mf == 0 && !rq.syn && !rq.bfm && !(rq.cmap && sy.sym) && z.I3 == 1 && !(d.plan_flags & QDAS_PLAN_NO_MIRROR)), "mirror mode on %s", D.text.c_str());
            EXPECT(!sy.big || (dt == QDAS_F32 && !sy.sym), "re-basing configuration on %s", D.text.c_str());
            EXPECT(!(sy.rfold && !sy.prefolded && fb == 1), "folds without a fold buffer: %s", D.text.c_str());
            EXPECT(sy.eligible || (sy.why && *sy.why), "%s", D.text.c_str());
            if (sy.prefolded_refused) { ++refused; EXPECT(d.plan_flags & QDAS_PLAN_PREFOLDED, "%s", D.text.c_str()); continue; }
            EXPECT(!(d.plan_flags & QDAS_PLAN_PREFOLDED) || sy.prefolded, "PREFOLDED neither honoured nor refused: %s", D.text.c_str());
            if (!sy.eligible || d.kernel == QDAS_KERNEL_GENERIC) { ++named; ++reasons[sy.why]; continue; }
            // ---- the build: every outcome of the probe chain (up to three probes: bits of `po`), table symmetric or not, side split / wide better or not
            const bool has_table = z.S > rq.npix && dt != QDAS_F64;
            const int txkind = z.VS ? (z.DV ? 0 : 1) : 2;
            for (int po = 0; po < 8; ++po) for (int ts = 0; ts < (has_table ? 2 : 1); ++ts) for (int sw2 = 0; sw2 < 4; ++sw2) {
                ProbeState ps;
                ps.dtype = dt; ps.sym = sy.sym; ps.rfold = sy.rfold; ps.mir = sy.mir ? (sy.mslab ? 2 : 1) : 0; ps.bpix = rq.bpix_mode;
                ps.set_tc(sy.tc_sym, sy.tc_narrow, sy.tc_fb, sy.tc_mirq, sy.tc_fold);
                int nprobe = 0, err = 0;
                auto probe = [&](const ProbeState &s, bool *fit) -> int {
                    // the probe launch itself must be legal (the plan-time probe kernels: prologue only)
                    LaunchShape L;
                    L.dtype = dt; L.sym = s.sym; L.fold = s.rfold; L.mir = s.mir; L.narrow_raw = s.narrow; L.big = sy.big; L.bf = rq.bfm; L.syn = rq.syn; L.probe = 1;
                    L.N = sy.kN; L.M = sy.kM; L.has_bpix = false;
                    L.act_bytes = ((rq.pix_arr >= 0 || d.rx_apod_kind) && dt != QDAS_F64) ? (uint32_t)(8 * (sy.kN + 1)) : 0u;
                    const char *why = launch_legal(L, nullptr);
                    EXPECT(!why, "probe launch refused (%s): %s", why, D.text.c_str());
                    const TileConfig tc = s.tc();
                    EXPECT(tc.mb > 0 && tc.window > 0, "%s", D.text.c_str());
                    *fit = nprobe < 3 ? ((po >> nprobe) & 1) != 0 : true;
                    ++nprobe;
                    return 0;
                };
                const bool fit = run_probe_chain(ps, D.sw, has_table, ts == 0, probe, &err);
                EXPECT(nprobe >= 1 && nprobe <= 3, "%d probes: %s", nprobe, D.text.c_str());
                if (nprobe < 3 && (po >> nprobe)) continue;                    // (outcomes of probes that did not run: the same walk)
                EXPECT(!(ps.mir && !fit), "mirror mode kept although tiles misfit: %s", D.text.c_str());
                EXPECT(!(ps.mir && has_table && !sy.rfold && ts == 1), "mirror mode with an asymmetric weight table: %s", D.text.c_str());
                BuildOutcome o;
                o.mir = ps.mir; o.narrow = ps.narrow;
                Symmetry sy2 = sy;
                sy2.mir = ps.mir != 0;
                uint64_t kN_eff = sy.kN;
                uint64_t tN = sy.kN, tM = sy.kM;
                uint64_t strM = (z.flag & QDAS_FLAG_TPOSE) ? z.T : z.T * z.N, strN = (z.flag & QDAS_FLAG_TPOSE) ? z.T * z.M : z.T;
                if (sy.swap) std::swap(strM, strN);
                if (side_split_applicable(d, rq, sy2, fit, txkind, sy.wtb != 0, D.sw)) {
                    if (sw2 & 1) { o.side_split = true; kN_eff = 2 * z.M; tN = 2 * z.M; tM = z.N; strN = (z.flag & QDAS_FLAG_TPOSE) ? z.T : z.T * z.N; strM = (z.flag & QDAS_FLAG_TPOSE) ? z.T * z.M : z.T; }
                } else if (sw2 & 1) continue;
                const bool stage_list = o.side_split || ((rq.bpix_mode || rq.pix_fold || rq.pix_arr >= 0 || d.rx_apod_kind) && dt != QDAS_F64);
                const bool table_in_kernel = (z.S > rq.npix) && !sy.rfold;
                const bool fit2 = fit || (o.side_split && (sw2 & 1) && false);   // (a side split that is kept has FEWER misfits, not necessarily none: wide may follow)
                if (wide_applicable(dt, fit2, rq.bfm, sy.big, ps.narrow, sy.prefolded, tN, tM, strN, strM, stage_list, table_in_kernel || (sy.rfold && z.S > rq.npix), D.sw)) {
                    if (sw2 & 2) o.wide = true;
                } else if (sw2 & 2) continue;
                TileConfig tc = o.wide ? tile_config(dt, 0, 2) : ps.tc();
                for (unsigned ntiles : {1u, 40u, 300u, 5000u}) {
                    o.ksplit = choose_ksplit(ntiles, 256, z.M, tc.mb, sy.sym && !o.wide, kN_eff, stage_list, rq.syn, D.sw);
                    EXPECT(o.ksplit >= 1 && o.ksplit <= 8, "ksplit %u: %s", o.ksplit, D.text.c_str());
                    LaunchShape L = derive_launch_shape(d, rq, sy2, o);
                    // which builds exist: the unfolded fp32 reciprocal mode and fp32 mirror plans with pixel weights are hiprtc builds (the plan asked for one, or was re-made without the mode)
                    const bool jit_on = (d.plan_flags & QDAS_PLAN_JIT) && !D.sw.no_jit && !L.bf;
                    L.jit = jit_on;
                    LaunchChoice ch;
                    const char *why = launch_legal(L, &ch);
                    EXPECT(!why, "one-frame launch refused (%s): %s [sym %d fold %d mir %d narrow %d big %d wide %d split %d ks %u]", why, D.text.c_str(), L.sym, L.fold, L.mir, L.narrow_raw, L.big,
                           (int)o.wide, (int)o.side_split, o.ksplit);
                    if (why) continue;
                    EXPECT(ch.cfg >= 0 && ch.cfg < 22 && ch.lds <= tile_lds_limit(L.sym), "cfg %d lds %zu: %s", ch.cfg, ch.lds, D.text.c_str());
                    ++cfgs[L.big ? 9 : L.bf ? 12 : ch.cfg];          // (cfg_index does not encode the re-basing / 'BF' instantiations)
                    // ---- streams: every frame-sharing launch stream_modes admits must be legal (prebuilt kernels: no hiprtc build shares frames)
                    PlanShape p;
                    p.dtype = dt; p.tiled = true; p.sym = L.sym; p.fold = L.fold; p.mir = L.mir; p.narrow = L.narrow_raw; p.big = L.big; p.bf = L.bf; p.syn = L.syn; p.stage_shift = L.stage_shift;
                    p.has_apix = L.has_apix; p.has_wtab = L.has_wtab; p.has_bpix = L.has_bpix; p.gen_kind = L.gen_kind; p.fmod = L.fmod; p.N = L.N; p.M = L.M; p.mem_device = d.mem == QDAS_MEM_DEVICE;
                    const StreamModes sm = stream_modes(p, z.N, z.M, D.sw);
                    EXPECT(!(sm.fb2_ok && sm.fold2_ok), "both stream modes: %s", D.text.c_str());
                    for (int nf = 2; nf <= 4; nf += 2) {
                        if (!((sm.fb2_ok && (nf == 2 || !sm.fb4_off)) || (sm.fold2_ok && nf == 2))) continue;
                        LaunchShape L2 = L;
                        L2.nfr = nf; L2.jit = false;
                        L2.has_part = o.ksplit > 1 && !L.bf;
                        const char *w2 = launch_legal(L2, nullptr);
                        EXPECT(!w2, "%d-frame launch refused (%s): %s", nf, w2, D.text.c_str());
                        if (!w2) ++cfgs[cfg_index(dt, L2.sym, nf, 0, L2.mir && L2.sym, L2.fold)];
                    }
                    ++resolved;
                }
            }
        }
    }
    printf("enumerated %llu descriptors, %ld walks of the geometry facts: %ld plans resolved to a legal launch configuration, %ld named a reason, %ld refused QDAS_PLAN_PREFOLDED\n",
           (unsigned long long)ndesc, walks, resolved, named, refused);
    printf("launch configurations reached:");
    for (auto &c : cfgs) printf(" %d:%ld", c.first, c.second);
    printf("\nreasons named:\n");
    for (auto &q : reasons) printf("  %6ld  %s\n", q.second, q.first.c_str());
    if (g_fail) { printf("%ld FAILED expectations\n", g_fail); return 1; }
    printf("mode enumeration OK\n");
    return 0;
}
