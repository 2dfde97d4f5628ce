#!/bin/bash
# Every plan-specialised (hiprtc) kernel the bench workloads of tools/profile_all.sh build, as complete "raw:" specs (QDAS_JIT_SPEC_LOG, csrc/jit.hip jit_spec_string):
#   tools/jit_specs_collect.sh > tests/jit_kernels.txt     (on a GPU box: the specs depend on probed tile shapes)
# tests/test_jit.py rebuilds each of them WITHOUT a device and fails on spilled VGPRs / scratch.
L=$(mktemp); export QDAS_JIT_SPEC_LOG=$L
Q="--no-cpu --no-traffic --no-general --steps 2 --warmup 1"
b() { python bench.py "$@" $Q > /dev/null 2>&1; }
b; b --workload c2; b --workload c5; b --workload c1
b --no-fold; b --frames 4; QDAS_NO_MIRROR=1 b; QDAS_NO_MIRROR=1 b --no-fold; QDAS_NO_MIRROR=1 b --workload c2; QDAS_NO_MIRROR=1 b --workload c1
b --no-reciprocal; QDAS_NO_MIRROR=1 b --no-reciprocal; b --prec halfT; b --fmod 5e6; b --window-apod; b --rx-apod fnumber:1.5; b --prec halfT --rx-apod fnumber:1.5
QDAS_NO_MIRROR=1 b --workload c5; b --workload c5 --prec single; b --workload c5 --gen-apod; b --workload c2 --prec double; b --workload c2 --prec double --fmod 5e6; b --workload c2 --window-apod; b --workload c2 --prec halfT
b --workload pw9; b --workload c1f; b --workload c1 --tx-apod multiline; b --workload c1 --tx-apod multiline --rx-apod acceptance:30 --rx-apod-array; b --workload c2 --rx-apod fnumber:1.5
echo "# hiprtc kernels of the bench workloads (tools/jit_specs_collect.sh on a GPU box); one complete JitSpec per line"
sort -u $L
