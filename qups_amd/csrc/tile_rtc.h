// tile_rtc.h -- what the device headers need from the toolchain, for both ways they are compiled: by hipcc into libqdas.so
// (prebuilt instantiations) and by hiprtc at plan creation (QDAS_PLAN_JIT: the plan's sizes as constants, jit.hip).  hiprtc
// pre-includes the HIP device runtime but has no <stdint.h> / <math.h>.
#pragma once
#ifdef __HIPCC_RTC__
typedef unsigned long long uint64_t;
typedef long long int64_t;
typedef unsigned int uint32_t;
typedef int int32_t;
typedef unsigned short uint16_t;
typedef unsigned long long uintptr_t;
#ifndef INFINITY
#define INFINITY __builtin_inff()
#endif
#else
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#endif

// Plan-specialised builds (jit.hip) define QDAS_JIT and one QDAS_SPEC_<NAME> per specialised quantity; everywhere else the
// value is read from the parameter block at run time.  The reference does the same with -DQUPS_<NAME>=... against __constant__
// symbols (reference src/sizes.cu:17-52, src/UltrasoundSystem.m:5727-5746).
#ifdef QDAS_JIT
#define QSPEC(NAME, RUNTIME) (QDAS_SPEC_##NAME)
#else
#define QSPEC(NAME, RUNTIME) (RUNTIME)
#endif
