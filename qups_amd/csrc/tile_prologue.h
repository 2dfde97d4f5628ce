// tile_prologue.h -- prologue of the tiled kernel (internal; included by das_tile_impl.h): tile-wide integer window bases A[m], B[n]
// and extents of the separable delay tau*fs + off = a(i,m) + b(i,n), the window-fit verdict, and the LDS copies of the geometry.
#pragma once

namespace qdas {

// Returns false when the workgroup is done: the tile's delay spread does not fit the LDS window (it has appended itself to the
// fallback list; the generic kernel takes it), or PROBE (plan-time footprint selection only wants the verdict).
template <class C> template <bool PROBE> __device__ __forceinline__ bool Tile<C>::prologue() {
    constexpr int WAVES = C::WAVES, THREADS = C::THREADS, K = C::K, INTERP = C::INTERP;
    constexpr bool SYM = C::SYM, LUT = C::LUT;
    // the geometry tables are read from global memory here and from their LDS copies in the main loop: a vector-memory load there
    // would sit behind the stage's LDS-DMA in the in-order vmcnt queue and expose the DMA latency every stage (measured: 15 of 64 ms)
    using GT = typename C::GT;
    const GT *gPv = geo_Pv(), *gNv = geo_Nv(), *gPr = geo_Pr();
    // the per-wave partial extrema go through LDS scratch (aliasing the windows) in CHUNKS of PCH elements: 1000-element apertures
    // (matrix arrays) would otherwise need more scratch than a CU has
    constexpr uint32_t PCH = QDAS_PROLOGUE_CHUNK;
    const uint32_t MX = (M > N ? M : N) < PCH ? (M > N ? M : N) : PCH;
    const uint64_t Ilut = P.i_begin + P.i_count;
    const bool has_st = QSPEC(HAS_ST, P.St != nullptr);
    float a_lo = INFINITY, a_hi = -INFINITY, a_ext = 0.f;            // per-thread partials of tile-wide stats
    // the plan's cached tables of this tile (tile_params.h pro_tab): loaded instead of computed; pro_out (plan creation, probe kernels): computed AND stored
    const size_t pro_stride = 2 * ((size_t)M + N) + 8 + (P.pro_mask ? (N + 31) / 32 : 0);
    const uint32_t tile_slot = tile_id - P.tiles_z * P.tile_x0;
    const float *ptab = nullptr;
    if constexpr (!PROBE && !LUT) { if (P.pro_tab) ptab = P.pro_tab + (size_t)tile_slot * pro_stride; }
    float b_lo = INFINITY, b_hi = -INFINITY, b_ext = 0.f;
    if (ptab) {
        if constexpr (!SYM) {
            for (uint32_t m = tid; m < M; m += THREADS) { Abase[m] = __float_as_int(ptab[2 * m]); Aext[m] = ptab[2 * m + 1]; }
        }
        for (uint32_t n = tid; n < N; n += THREADS) {
            const float bb = ptab[2 * ((size_t)M + n)], e = ptab[2 * ((size_t)M + n) + 1];
            if constexpr (C::F64) nrec64[n] = rec64{gPr[3 * n], gPr[3 * n + 1], gPr[3 * n + 2], __float_as_int(bb), 0};
            else nrec[n] = make_float4(bb, gPr[3 * n], gPr[3 * n + 1], gPr[3 * n + 2]);
            Bext[n] = e;
            if constexpr (SYM) { Abase[n] = __float_as_int(bb) + symCi; Aext[n] = e + 1.0f; }
        }
        const float *st6 = ptab + 2 * ((size_t)M + N);
        a_lo = st6[0]; b_lo = st6[1]; a_hi = st6[2]; b_hi = st6[3]; a_ext = st6[4]; b_ext = st6[5];
    } else {
    // The window bases / extents only need the delays to a small fraction of a sample: fp32 estimates with an explicit error
    // margin (DLT, below) -- a quarter of the fp64 cost.  Focused transmits keep fp64: their delay flips sign with
    // (Pi - Pv).Nv (copysign, src/bf.cu:107) and the two precisions must agree on the sign of a dot product that may be ~0.
    const bool pro32 = !LUT && kindB != 1 && kindS != 1 && kindS != 3;
    const bool one_sided = C::ACT && !LUT && kindS == 3;     // stage elements that stand for ONE side of a focal plane (tile_params.h): the lanes
                                                           // of the other side stay out of the element's window base (NaN = "not mine" below)
    const float cf32 = (float)cf, fs32 = (float)fs;
    auto a_est = [&](uint32_t m) -> float {
        if (!pro32) return (float)a_of(m, gPv, gNv);
        const float rx = (float)(px - gPv[4 * m]), ry = (float)(py - gPv[4 * m + 1]), rz = (float)(pz - gPv[4 * m + 2]);
        const float d = kindB != 2 ? __builtin_amdgcn_sqrtf(rx * rx + ry * ry + rz * rz) : rx * (float)gNv[3 * m] + ry * (float)gNv[3 * m + 1] + rz * (float)gNv[3 * m + 2];
        return d * cf32 - (float)gPv[4 * m + 3] * fs32 + (float)tapinfo<INTERP>::OFF;
    };
    auto b_est = [&](uint32_t n) -> float {
        if constexpr (LUT) return P.lut_rx[ipx + Ilut * n];
        if (kindS == 1) return (float)s_at(n, gPr[3 * n], gPr[3 * n + 1], gPr[3 * n + 2]);
        if constexpr (C::ACT) {                          // (only the single-frame general kernels run such plans)
            if (kindS == 3) return on_side(n, gPr[3 * n], gPr[3 * n + 1], gPr[3 * n + 2]) ? (float)s_at(n, gPr[3 * n], gPr[3 * n + 1], gPr[3 * n + 2]) : __builtin_nanf("");
        }
        const float rx = (float)(px - gPr[3 * n]), ry = (float)(py - gPr[3 * n + 1]), rz = (float)(pz - gPr[3 * n + 2]);
        if (!has_st) return __builtin_amdgcn_sqrtf(rx * rx + ry * ry + rz * rz) * cf32;
        if (kindS == 0) return __builtin_amdgcn_sqrtf(rx * rx + ry * ry + rz * rz) * cf32 - P.St[4 * n] * fs32;
        return (rx * P.St[4 * n + 1] + ry * P.St[4 * n + 2] + rz * P.St[4 * n + 3]) * cf32 - P.St[4 * n] * fs32;
    };
    // |fp32 estimate - fp64 delay| <= ~4e-7 * (|distance*cf| + |t0*fs|), and |distance*cf| <= |a| + |t0*fs| + 1: 1e-6 is generous
    auto margin = [](float mn, float mx, float t0fs) -> float { return 1.0e-6f * (fmaxf(fabsf(mn), fabsf(mx)) + 2.0f * fabsf(t0fs) + 2.0f); };
    // (four elements per pass: independent reduction chains overlap)
    auto minmax4 = [&](float (&v)[4], uint32_t e0, uint32_t cnt, uint32_t c0, bool neutral) {
        float lo[4], hi[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {                   // a NaN delay poisons the tile's extent -- unless it means "this lane is not the element's"
            const bool ok = v[q] == v[q];
            lo[q] = ok ? v[q] : INFINITY;
            hi[q] = ok ? v[q] : (neutral ? -INFINITY : INFINITY);
        }
        wave_minmax63x4(lo, hi);
        if (lane == 63) {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (e0 + q < cnt) { part[wave * MX + (e0 - c0) + q] = lo[q]; part[(WAVES + wave) * MX + (e0 - c0) + q] = hi[q]; }
        }
    };
    if constexpr (!SYM) {
      for (uint32_t c0 = 0; c0 < M; c0 += PCH) {
        const uint32_t c1 = c0 + PCH < M ? c0 + PCH : M;
        for (uint32_t m = c0; m < c1; m += 4) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = a_est(m + q < M ? m + q : M - 1);
            minmax4(v, m, c1, c0, false);
        }
        __syncthreads();
        for (uint32_t m = c0 + tid; m < c1; m += THREADS) {
            float mn = part[m - c0], mx = part[WAVES * MX + m - c0];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) { mn = fminf(mn, part[w * MX + m - c0]); mx = fmaxf(mx, part[(WAVES + w) * MX + m - c0]); }
            const float dlt = margin(mn, mx, LUT ? 0.f : (float)gPv[4 * m + 3] * fs32);
            const float fl = floorf(mn - dlt) - 1.0f;        // margin: the estimate may lie above the true minimum
            const bool fin = fabsf(fl) < 1.0e9f;
            const float e = fin ? ((mx + dlt) - fl) + 0.01f : INFINITY;
            Abase[m] = fin ? (int)fl : 0;
            Aext[m] = e;
            a_lo = fminf(a_lo, fl); a_hi = fmaxf(a_hi, fl + e); a_ext = fmaxf(a_ext, e);
        }
        __syncthreads();
      }
    }
    for (uint32_t c0 = 0; c0 < N; c0 += PCH) {
      const uint32_t c1 = c0 + PCH < N ? c0 + PCH : N;
      for (uint32_t n = c0; n < c1; n += 4) {
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = b_est(n + q < N ? n + q : N - 1);
        minmax4(v, n, c1, c0, one_sided);
      }
      __syncthreads();
      for (uint32_t n = c0 + tid; n < c1; n += THREADS) {
        float mn = part[n - c0], mx = part[WAVES * MX + n - c0];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) { mn = fminf(mn, part[w * MX + n - c0]); mx = fmaxf(mx, part[(WAVES + w) * MX + n - c0]); }
        float t0fs = 0.f;
        if constexpr (SYM) t0fs = (float)gPv[3] * fs32;
        else if constexpr (!LUT) t0fs = has_st ? P.St[4 * n] * fs32 : 0.f;
        const bool nobody = one_sided && mx < mn;        // no pixel of the tile on this element's side: never staged (its weights are all zero)
        if (nobody) { mn = 0.f; mx = 0.f; }
        const float dlt = margin(mn, mx, t0fs);
        const float fl = floorf(mn - dlt) - 1.0f;
        const bool fin = fabsf(fl) < 1.0e9f;
        const float e = nobody ? 0.0f : (fin ? ((mx + dlt) - fl) + 0.01f : INFINITY);
        if constexpr (LUT) nrec[n] = make_float4(__int_as_float(fin ? (int)fl : 0), 0.f, 0.f, 0.f);
        else if constexpr (C::F64) nrec64[n] = rec64{gPr[3 * n], gPr[3 * n + 1], gPr[3 * n + 2], fin ? (int)fl : 0, 0};
        else nrec[n] = make_float4(__int_as_float(fin ? (int)fl : 0), gPr[3 * n], gPr[3 * n + 1], gPr[3 * n + 2]);
        Bext[n] = e;
        if (!nobody) { b_lo = fminf(b_lo, fl); b_hi = fmaxf(b_hi, fl + e); b_ext = fmaxf(b_ext, e); }
        if constexpr (SYM) {                             // a - A = (b - B) + frac(C) in [1, Bext + 1)
            Abase[n] = (fin ? (int)fl : 0) + symCi;
            Aext[n] = e + 1.0f;
            a_lo = fminf(a_lo, fl + (float)symCi); a_hi = fmaxf(a_hi, fl + (float)symCi + e + 1.0f); a_ext = fmaxf(a_ext, e + 1.0f);
        }
      }
      __syncthreads();                                 // part[] is free again
    }
    a_lo = wave_min(a_lo); b_lo = wave_min(b_lo);
    a_hi = wave_max(a_hi); b_hi = wave_max(b_hi); a_ext = wave_max(a_ext); b_ext = wave_max(b_ext);
    if (lane == 0) { float *q = part + wave * 8; q[0] = a_lo; q[1] = b_lo; q[2] = a_hi; q[3] = b_hi; q[4] = a_ext; q[5] = b_ext; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const float *q = part + w * 8;
        a_lo = fminf(a_lo, q[0]); b_lo = fminf(b_lo, q[1]); a_hi = fmaxf(a_hi, q[2]); b_hi = fmaxf(b_hi, q[3]);
        a_ext = fmaxf(a_ext, q[4]); b_ext = fmaxf(b_ext, q[5]);
    }
    }       // (!ptab)
    if constexpr (PROBE && !LUT) {                     // plan creation: this tile's tables go to the plan's buffer (the LDS copies are complete: the loops above end in barriers)
        if (P.pro_out) {
            float *o = P.pro_out + (size_t)tile_slot * pro_stride;
            if constexpr (!SYM) { for (uint32_t m = tid; m < M; m += THREADS) { o[2 * m] = __int_as_float(Abase[m]); o[2 * m + 1] = Aext[m]; } }
            for (uint32_t n = tid; n < N; n += THREADS) {
                float bb;
                if constexpr (C::F64) bb = __int_as_float(nrec64[n].b); else bb = nrec[n].x;
                o[2 * ((size_t)M + n)] = bb; o[2 * ((size_t)M + n) + 1] = Bext[n];
            }
            if (tid == 0) { float *q = o + 2 * ((size_t)M + N); q[0] = a_lo; q[1] = b_lo; q[2] = a_hi; q[3] = b_hi; q[4] = a_ext; q[5] = b_ext; }
        }
    }
    // every lane's last tap (+1 for the rint/floor ambiguity at exact integers) must be inside the staged window
    if (!(a_ext + b_ext + (float)(K + 1) <= (float)((PROBE && P.probe_w > 0) ? P.probe_w : C::W))) {
        if (tid == 0 && split == 0) {
            const uint32_t slot = atomicAdd(&P.fallback_list[0], 1u);
            if (slot < P.fallback_cap) P.fallback_list[1 + slot] = tile_id;
        }
        return false;
    }
    if constexpr (PROBE) return false;
    // every window of every stage strictly inside the record?  (uniform) -> branch-free loop
    tile_interior = (a_lo + b_lo >= 1.0f) && (a_hi + b_hi + (float)(K + 1) < (float)T);
    // every tap of every lane lies in [0, a_ext + b_ext + K + 1) samples of its window (the fit test above): samples beyond are never read -- nor staged
#ifdef QDAS_DMA_TRIM
    need_b = __builtin_amdgcn_readfirstlane((((int)(a_ext + b_ext) + K + 3) * (int)C::SB + 15) & ~15);
    if (need_b > (int)C::WB) need_b = (int)C::WB;
#else
    need_b = (int)C::WB;
#endif
    if constexpr (C::FMOD && !C::F64) {                // remodulation phase constants (cycles) of the window bases, tile_pairs.h
        const double f = P.fmod / fs;
        for (uint32_t m = tid; m < M; m += THREADS) { const double c = ((double)Abase[m] + 0.5 - tapinfo<INTERP>::OFF) * f; Aext[m] = (float)(c - floor(c)); }
        for (uint32_t n = tid; n < N; n += THREADS) { const double c = (double)__float_as_int(nrec[n].x) * f; Bext[n] = (float)(c - floor(c)); }
    }
    if constexpr (!LUT) {
        for (uint32_t k = tid; k < 4 * M; k += THREADS) PvL[k] = gPv[k];
        for (uint32_t k = tid; k < 3 * M; k += THREADS) NvL[k] = gNv[k];
    }
    __syncthreads();
    return true;
}

}  // namespace qdas
