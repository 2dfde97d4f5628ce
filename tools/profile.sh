#!/bin/bash
# rocprofv3 evidence for one bench.py workload: kernel-trace/stats + separate PMC passes (never combined with
# sys/hip/hsa tracing).  Run ON the GPU box through gpurun from the repo root:
#   tools/profile.sh <tag> [bench args...]      -> gpurun_out/prof_<tag>/  (+ summary.txt)
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="env QDAS_BENCH_CHILD=1 python $REPO/bench.py --steps 6 --warmup 1 --no-cpu --no-traffic --no-general $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.err
pass() { # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o p -- $BENCH > /dev/null 2> $OUT/pmc_$name.err
}
pass clk  GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
pass inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass lds  SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F64
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass l2   TCC_HIT_sum TCC_MISS_sum
python $REPO/tools/profile_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
