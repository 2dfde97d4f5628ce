#!/bin/bash
# Same-box A/B of the headline kernel's LDS-DMA: hiprtc builds through QDAS_JIT_DEFINES / QDAS_JIT_MB / QDAS_JIT_NBUF, interleaved rounds, kernel + fold ms.
#   tools/dma_ab.sh [rounds] > profiles/r06/dma_ab_c3.txt       (on a GPU box)
R=${1:-2}
run() { # label env...
  local label=$1; shift
  v=$(env "$@" QDAS_BENCH_CHILD=1 python bench.py --steps 6 --warmup 2 --no-cpu --no-traffic --no-general 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])" 2>/dev/null)
  echo "$label|$v"
}
for r in $(seq $R); do
  run "shipped (32 tx x 2 sets x 128 samples, 2 buffers)" QDAS_X=0
  run "no LDS-DMA at all (QDAS_ABL=1)" QDAS_JIT_DEFINES=QDAS_ABL=1
  run "DMA issued, every window from one hot source address (QDAS_DMA_SAME_SRC: no memory side)" QDAS_JIT_DEFINES=QDAS_DMA_SAME_SRC=1
  run "DMA trimmed to the bytes the tile can touch (QDAS_DMA_TRIM)" QDAS_JIT_DEFINES=QDAS_DMA_TRIM=1
  run "no end-of-stage wait / barrier (QDAS_ABL=16)" QDAS_JIT_DEFINES=QDAS_ABL=16
  run "no DMA and no barrier (QDAS_ABL=17)" QDAS_JIT_DEFINES=QDAS_ABL=17
  run "no priority staircase (QDAS_ABL=2048)" QDAS_JIT_DEFINES=QDAS_ABL=2048
  run "16-transmit stages, 2 buffers (QDAS_JIT_MB=16)" QDAS_JIT_MB=16
  run "16-transmit stages, 3 buffers (QDAS_JIT_MB=16 QDAS_JIT_NBUF=3)" QDAS_JIT_MB=16 QDAS_JIT_NBUF=3
  run "plain instead of pipelined pair loop (QDAS_ABL=256)" QDAS_JIT_DEFINES=QDAS_ABL=256
done | sort -s -t'|' -k1,1 | awk -F'|' '{a[$1]=a[$1]" "$2} END{for(k in a) printf "%-100s %s\n", k, a[k]}' | sort
