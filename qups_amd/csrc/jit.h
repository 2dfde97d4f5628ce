// jit.h -- plan-specialised builds of the tiled kernel with hiprtc (internal; implementation and rationale in jit.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "tile_params.h"
#include "das_tile_cfg.h"

namespace qdas {

// Everything that becomes a constant of the compiled kernel: the template parameters of the instantiation and the plan's sizes.
struct JitSpec {
    int interp, dtype, fmod, wtab, sym, big;            // instantiation (one frame per launch, geometry-driven delays)
    int waves, mb, w, nbuf;                             // launch configuration (das_tile_cfg.h)
    uint64_t N, M, T, I1, strN, strM;                   // stage / block elements, record length, fastest image dimension, trace strides
    int kindB, kindS, tzl, wzl;                         // delay kinds, tile / wave footprint
    unsigned ksplit;
    int gen_kind, has_apix, apix_real, syn, has_st, has_cinv_pix;
    int mir;                                            // lateral-mirror mode: the two-window-set instantiation (TileCfg::FB2), one frame
    int mirq;                                           // reciprocal + lateral-mirror mode: four window sets (TileCfg::MIRQ)
    int mslab;                                          // ... of a mirror slab (tile_params.h mir == 2)
    int wreal;                                          // the weight table is real: the weighted accumulation is one packed FMA per sample
    int fold;                                           // reciprocity-folded data (TileCfg::FOLD; with mirq: two window sets)
    int lut;                                            // table-driven delays (qdas_das_lut) in lateral-mirror mode: exists as a hiprtc build only (round 6)
    int plain;                                          // the plain instead of the software-pipelined pair loop (QDAS_ONEACC_PLAIN: plan_jit's second attempt when a build uses scratch)
};

// One point of the template's matrix as the prebuilt instantiations name it: built on demand when libqdas.so does not carry it (jit.hip)
struct LazySpec { int interp, sample_bytes, fm, wt, ci, probe; };
std::string lazy_tile_source(const LazySpec &k);
hipError_t lazy_tile_launch(const LazySpec &k, const TileParams &P, unsigned grid, unsigned block, size_t lds, hipStream_t s, bool prepare_only);
void lazy_tile_reset();
const std::string &lazy_tile_last();
std::string jit_compile_source(const std::string &src, std::vector<char> *code, std::string *key_out, bool use_disk);
std::string jit_get_kernel_source(const std::string &src, int device, hipFunction_t *fn, std::string *key_out);

std::string jit_source(const JitSpec &k);
std::string jit_spec_string(const JitSpec &k);      // every field, "raw:name=value,..." (QDAS_JIT_SPEC_LOG; qdas_debug_jit_compile parses it back)
// "" on success, else the reason (hiprtc missing, compile log, ...)
std::string jit_compile(const JitSpec &k, std::vector<char> *code, std::string *key_out, bool use_disk = true);
std::string jit_get_kernel(const JitSpec &k, int device, hipFunction_t *fn, std::string *key_out);
hipError_t jit_launch(hipFunction_t fn, const TileParams &P, unsigned grid, unsigned block, size_t lds, hipStream_t s);

}  // namespace qdas
