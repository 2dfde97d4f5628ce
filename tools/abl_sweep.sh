#!/bin/bash
# Same-box ablation sweep of the plan-specialised (hiprtc) kernel: tools/abl_sweep.sh <rounds> "<bench args>" <QDAS_JIT_DEFINES value> ...
# ("-" = the product build).  Interleaved rounds; prints kernel ms per variant.  Run on the GPU box through gpurun.
R=$1; ARGS=$2; shift 2
for r in $(seq $R); do
  for D in "$@"; do
    echo -n "$D "
    if [ "$D" = "-" ]; then unset QDAS_JIT_DEFINES; else export QDAS_JIT_DEFINES="$D"; fi
    QDAS_BENCH_CHILD=1 python bench.py --steps 6 --warmup 2 --no-cpu --no-traffic --no-general $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])"
  done
done | sort | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print k, a[k]}' | sort
