// tile_staging.h -- LDS-DMA staging of the tiled kernel (internal; included by das_tile_impl.h).
//
// One buffer descriptor per TRANSMIT BLOCK, based at trace (rx n_lo, tx m0) (mirror: (rx m0, tx 0)): window j starts
// (j*strM + A[m0+j] + B[n]) samples after it.  A lane moves 16 bytes; a wave-instruction 1 KiB (only 16-byte pieces give a
// contiguous LDS image: a 12-byte piece still advances 16 bytes per lane -- measured with tools/scratch/dma12.hip).  Offsets before
// the base wrap to >= num_records and, like offsets past the end of x, deliver 0.  Samples outside [0, T) of a trace but inside x
// read the neighbouring trace: they are only ever touched by lanes that the checked loop masks out (select, not multiply).
// Everything that does not depend on the receiver -- A[m], the window's trace offset -- is folded into one scalar per window when
// the block starts (dma_block).  A stage then costs two scalar adds per window: offset = soff (running receiver offset, 32-bit by
// the plan-time check) + wb[r] + B[n]*SB.
// Reciprocal mode walks the whole frame inside one transmit block (the mirror "transmits", or the receivers of transposed data):
// its running offsets are kept below 2^30 by re-basing the descriptor when they get there (uniform, rare).  The general kernels
// keep one descriptor per block (plan-time check: the walk stays below 2^31 bytes); transposed fp32 frames beyond that run the BIG
// instantiation (launch configuration 9), which re-bases as well.
#pragma once

namespace qdas {

constexpr uint32_t DMA_REBASE = 1u << 30;

// descriptor based `o` bytes into the frame (+ `extra`: the frame itself); records beyond the frame read as zeros
template <class C> __device__ __forceinline__ __amdgpu_buffer_rsrc_t Tile<C>::make_rs(uint64_t o, uint64_t extra) const {
    const uint64_t rem = xbytes > o ? xbytes - o : 0;   // (m0 >= M when the split is exhausted: nothing more is issued)
    return __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)P.x + (rem ? o + extra : 0)), 0, rem > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)rem, 0x00020000);
}

// window index (within its set) this wave stages in round r: windows wave + WAVES*r; with four frames the lower / upper half of
// the waves take frames (0, 2) / (1, 3) of window wave % MB
template <class C> __device__ __forceinline__ int Tile<C>::wjr(int r) const {
    return C::FB4 ? __builtin_amdgcn_readfirstlane(wave % C::MB) : __builtin_amdgcn_readfirstlane(wave + C::WAVES * r);
}

template <class C> __device__ __forceinline__ void Tile<C>::dma_block(uint32_t m0) {
    constexpr int SB = C::SB;
    if constexpr (C::MIRQ) {
        // reciprocal + lateral-mirror mode: ONE descriptor over the whole frame (the host keeps such plans below 2 GiB) and absolute byte
        // offsets; wave j stages window j (transmit m = m0 + j) of all four window sets.  At receiver n the four traces are
        //   set 0: (rx n, tx m)   set 1: (rx m, tx n)   set 2: (rx N-1-n, tx N-1-m)   set 3: (rx N-1-m, tx N-1-n)
        // -- every block starts at n = 0; sets 0 / 1 walk up by one receiver / transmit stride per stage, sets 2 / 3 walk down
        rsD = make_rs(0, 0);
        if constexpr (C::FOLD && C::FB2) rsM = make_rs(0, P.x_fstride);      // the folded copy of the second frame: the same offsets
        if constexpr (C::FOLD) {
            // folded data: my pixel's trace (rx n, tx m) -- walking up one receiver stride per stage -- and its mirror image's
            // (rx N-1-m, tx N-1-n) -- the upper-triangle twin of (N-1-n, N-1-m) --, walking DOWN one transmit stride per stage
#pragma unroll
            for (int r = 0; r < C::WPW; ++r) {
                const uint32_t m = m0 + (uint32_t)wjr(r);
                const uint32_t mc = m < M ? m : M - 1;
                const long am = (long)__builtin_amdgcn_readfirstlane(Abase[mc]);
                qo[2 * r] = (int)(((long)mc * (long)strM + am) * SB);
                qo[2 * r + 1] = (int)(((long)(N - 1 - mc) * (long)strN + (long)(N - 1) * (long)strM + am) * SB);
            }
            return;
        }
        const uint32_t m = m0 + (uint32_t)wjr(0);
        const uint32_t mc = m < M ? m : M - 1;
        const long am = (long)__builtin_amdgcn_readfirstlane(Abase[mc]);
        qo[0] = (int)(((long)mc * (long)strM + am) * SB);
        qo[1] = (int)(((long)mc * (long)strN + am) * SB);
        qo[2] = (int)(((long)(N - 1 - mc) * (long)strM + (long)(N - 1) * (long)strN + am) * SB);
        qo[3] = (int)(((long)(N - 1 - mc) * (long)strN + (long)(N - 1) * (long)strM + am) * SB);
        return;
    }
    const uint64_t o = ((uint64_t)m0 * strM + (uint64_t)(n_lo >> (C::ACT ? P.stage_shift : 0)) * strN) * SB;
    if constexpr (C::SYM || C::BIG) offD = o;
    rsD = make_rs(o, (uint64_t)fa * P.x_fstride);
    soff = 0;
#pragma unroll
    for (int r = 0; r < C::WPW; ++r) {
        const int j = wjr(r);
        const uint32_t m = m0 + j;
        const int am = __builtin_amdgcn_readfirstlane(Abase[m < M ? m : M - 1]);
        wb[r] = am * SB + (int)((long)j * (long)strM * SB);
        if constexpr (C::SYM && !C::FOLD) wb2[r] = am * SB + (int)((long)j * (long)strN * SB);
    }
    if constexpr (C::FBX) rsM = make_rs(o, (uint64_t)fb * P.x_fstride);   // the same traces of the next frame (four frames: of frame fb)
    if constexpr (C::FB2) {
        if (QSPEC(MIR, P.mir)) {
            // lateral-mirror mode: the second window set holds the traces (rx N-1-n, tx M-1-m) of the SAME frame.  Descriptor at the
            // lowest trace this block touches -- (rx N - n_hi, tx M - m0 - MB), clamped to transmit 0 in the last block --; window j sits
            // (mt(j) - mlo) transmit strides and (n_hi - 1 - n) receiver strides after it (the receiver offset runs DOWN, soff2)
            const uint32_t mlo = m0 + (uint32_t)C::MB <= M ? M - m0 - (uint32_t)C::MB : 0u;
            const uint64_t oM = ((uint64_t)mlo * strM + (uint64_t)(N - n_hi) * strN) * SB;
            rsM = make_rs(oM, 0);
            soff2 = (n_hi - 1u - n_lo) * (uint32_t)strN * (uint32_t)SB;
#pragma unroll
            for (int r = 0; r < C::WPW; ++r) {
                const int j = wjr(r);
                const uint32_t m = m0 + j;
                const int am = __builtin_amdgcn_readfirstlane(Abase[m < M ? m : M - 1]);
                const uint32_t mt = m < M ? M - 1u - m : 0u;
                wb2[r] = am * SB + (int)((long)(mt - mlo) * (long)strM * SB);
            }
        }
    }
    if constexpr (C::SYM && !C::FOLD) {               // mirror traces x[:, rx = m0 + j, tx = n] (reciprocal mode starts every block at n = 0)
        offM = (uint64_t)m0 * strN * SB;
        rsM = make_rs(offM, 0);
        soff2 = 0;
    }
}

// stage (receiver with window base bn = B[n], current DMA transmit block) into window buffer `buf`
template <class C> __device__ __forceinline__ void Tile<C>::stage_dma(int bn, int buf) {
    constexpr int SB = C::SB, WB = C::WB, PB = C::PB, PCS = C::PCS, NW = C::NW, MB = C::MB;
    const int bs = bn * SB;
    const int l16 = (int)(lane_now() * 16u);          // (byte offset of this lane's 16 bytes in a piece; see Tile::lane_now)
    if constexpr (C::FOLDQ) {
#pragma unroll
        for (int r = 0; r < C::WPW; ++r) {
            const int j = wjr(r);
#pragma unroll
            for (int f = 0; f < (C::FB2 ? 2 : 1); ++f) {           // (two frames: window sets {f0 mine, f0 image, f1 mine, f1 image})
#pragma unroll
                for (int k = 0; k < 2; ++k) {
#pragma unroll
                    for (int q = 0; q < PCS; ++q) {
                        lds_void *dst = (lds_void *)((unsigned char *)win + ((buf * NW + (2 * f + k) * MB + j) * WB + q * PB));
#ifdef QDAS_DMA_SAME_SRC                               // (measurement builds, profiles/r06/dma_ab_c3.txt: every window from ONE hot source address -- the DMA's issue and LDS-write side without its memory side)
                        if (l16 < need_b - q * PB)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(f ? rsM : rsD, dst, 16, l16, (j & 1) * PB + q * PB, 0, 0);
#else
                        if (l16 < need_b - q * PB)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(f ? rsM : rsD, dst, 16, l16, qo[2 * r + k] + bs + q * PB, 0, 0);
#endif
                    }
                }
            }
            qo[2 * r] += (int)((uint32_t)strN * SB); qo[2 * r + 1] -= (int)((uint32_t)strM * SB);
        }
        return;
    } else if constexpr (C::MIRQ) {
        const int j = wjr(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int q = 0; q < PCS; ++q) {
                lds_void *dst = (lds_void *)((unsigned char *)win + ((buf * NW + k * MB + j) * WB + q * PB));
                if (l16 < need_b - q * PB)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, dst, 16, l16, qo[k] + bs + q * PB, 0, 0);
            }
        }
        qo[0] += (int)((uint32_t)strN * SB); qo[1] += (int)((uint32_t)strM * SB);
        qo[2] -= (int)((uint32_t)strN * SB); qo[3] -= (int)((uint32_t)strM * SB);
        return;
    }
#pragma unroll
    for (int r = 0; r < C::WPW; ++r) {
        const int j = wjr(r);
        const int so = (int)soff + wb[r] + bs;
#pragma unroll
        for (int q = 0; q < (hooks::one_dma_piece ? 1 : PCS); ++q) {
            lds_void *dst = (lds_void *)((unsigned char *)win + ((buf * NW + fa * MB + j) * WB + q * PB));
            if (l16 < need_b - q * PB)             // trailing partial piece: upper lanes masked off
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, dst, 16, l16, so + q * PB, 0, 0);
        }
    }
    if constexpr (C::FBX) {                           // next frame: same offsets, other descriptor, another window set
        const bool mirror = C::FB2 && QSPEC(MIR, P.mir);   // (lateral-mirror mode: the mirrored traces of the same frame, their own offsets)
#pragma unroll
        for (int r = 0; r < C::WPW; ++r) {
            const int j = wjr(r);
            const int so = mirror ? (int)soff2 + wb2[r] + bs : (int)soff + wb[r] + bs;
#pragma unroll
            for (int q = 0; q < PCS; ++q) {
                lds_void *dst = (lds_void *)((unsigned char *)win + ((buf * NW + fb * MB + j) * WB + q * PB));
                if (l16 < need_b - q * PB)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsM, dst, 16, l16, so + q * PB, 0, 0);
            }
        }
    }
    soff += (uint32_t)strN * SB;                      // next receiver, same transmit block
    if constexpr (C::FB2) { if (QSPEC(MIR, P.mir)) soff2 -= (uint32_t)strN * SB; }      // (its mirror image: one receiver down)
    if constexpr (C::SYM || C::BIG) {
        if (soff >= DMA_REBASE) {
            offD += soff; soff = 0; rsD = make_rs(offD, (uint64_t)fa * P.x_fstride);
            // (a second window set that shares `soff` -- the same traces of another frame: launch configuration 21, two folded frames
            //  without the mirror mode -- moves with it; the lateral-mirror set walks its own offset, soff2, from its own base)
            if constexpr (C::FBX) { if (!(C::FB2 && QSPEC(MIR, P.mir))) rsM = make_rs(offD, (uint64_t)fb * P.x_fstride); }
        }
    }
    if constexpr (C::SYM && !C::FOLD) {               // same window start A[m] + B[n] in the mirror trace
#pragma unroll
        for (int r = 0; r < C::WPW; ++r) {
            const int j = __builtin_amdgcn_readfirstlane(wave + C::WAVES * r);
            const int so = (int)soff2 + wb2[r] + bs;
#pragma unroll
            for (int q = 0; q < PCS; ++q) {
                lds_void *dst = (lds_void *)((unsigned char *)win + ((buf * NW + MB + j) * WB + q * PB));
                if (l16 < need_b - q * PB)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsM, dst, 16, l16, so + q * PB, 0, 0);
            }
        }
        soff2 += (uint32_t)strM * SB;                 // next "transmit" n of the mirror traces
        if (soff2 >= DMA_REBASE) { offM += soff2; soff2 = 0; rsM = make_rs(offM, 0); }
    }
}

}  // namespace qdas
