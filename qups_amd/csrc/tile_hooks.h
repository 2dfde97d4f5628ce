// tile_hooks.h -- the ONE place where measurement builds hook into the tiled kernel (internal).
//
// The product (libqdas.so, and every hiprtc build) compiles this header with QDAS_ABL == 0 and QDAS_PROF == 0: every member of
// `hooks` is then a compile-time `false` / an empty inline function and the kernel contains nothing of them.  Profiling builds
// (tools/ablate.sh, tools/phase_timers.py: unity build with -DQDAS_ABL=<bits> / -DQDAS_PROF=1) switch single costs off to
// attribute the kernel time (profiles/ablation_r01.txt) or time the phases of a stage with s_memtime.
#pragma once
#include "tile_rtc.h"

#ifndef QDAS_ABL
#define QDAS_ABL 0
#endif
#ifndef QDAS_PROF
#define QDAS_PROF 0
#endif

#if QDAS_PROF
__device__ unsigned long long qdas_prof_buf[2 * 8 * 8192];
extern "C" int qdas_debug_read_prof(unsigned long long *dst, size_t n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(qdas_prof_buf), n * sizeof(unsigned long long));
}
#endif

namespace qdas {

struct hooks {
    static constexpr unsigned abl = QDAS_ABL;
    static constexpr bool no_stage_dma     = (abl & 1) != 0;     // skip the LDS-DMA of every stage
    static constexpr bool fake_rx_delay    = (abl & 2) != 0;     // synthetic receive residual instead of the fp64 delay
    static constexpr bool no_tap_reads     = (abl & 4) != 0;     // taps from registers instead of LDS
    static constexpr bool trivial_weights  = (abl & 8) != 0;     // no interpolation polynomials
    static constexpr bool no_stage_barrier = (abl & 16) != 0;    // no end-of-stage wait / barrier
    static constexpr bool one_dma_piece    = (abl & 64) != 0;    // one DMA piece per window
    static constexpr bool linear_taps      = (abl & 128) != 0;   // conflict-free synthetic tap addresses
    static constexpr bool no_pipeline      = (abl & 256) != 0;   // plain pair loop where the software-pipelined one would run
    static constexpr bool no_late_dma      = (abl & 1024) != 0;  // every wave issues the next stage's DMA before its pair loop
    static constexpr bool no_fair_prio     = (abl & 2048) != 0;  // no s_setprio staircase in the pair loops
    static constexpr bool product = abl == 0 && QDAS_PROF == 0;

    // phase timers of waves 0 and 15 of every workgroup (tools/phase_timers.py)
    struct Timer {
#if QDAS_PROF
        unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, start, t[5];
        static __device__ __forceinline__ unsigned long long tick() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long v = __builtin_readcyclecounter();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            return v;
        }
        __device__ __forceinline__ void begin() { start = tick(); }
        __device__ __forceinline__ void prologue_done() { pt[0] = tick() - start; }
        __device__ __forceinline__ void mark(int k) { t[k] = tick(); }                       // 0..4 inside a stage
        __device__ __forceinline__ void stage_done() { pt[1] += t[1] - t[0]; pt[2] += t[2] - t[1]; pt[3] += t[3] - t[2]; pt[4] += t[4] - t[3]; pt[6] += 1; }
        __device__ __forceinline__ void finish(int lane, int wave, int waves) {
            pt[5] = tick() - start;
            if (lane == 0 && (wave == 0 || wave == waves - 1) && blockIdx.x < 8192) {
                unsigned long long *o = qdas_prof_buf + ((size_t)blockIdx.x * 2 + (wave ? 1 : 0)) * 8;
                for (int k = 0; k < 8; ++k) o[k] = pt[k];
            }
        }
#else
        __device__ __forceinline__ void begin() {}
        __device__ __forceinline__ void prologue_done() {}
        __device__ __forceinline__ void mark(int) {}
        __device__ __forceinline__ void stage_done() {}
        __device__ __forceinline__ void finish(int, int, int) {}
#endif
    };
};

}  // namespace qdas
