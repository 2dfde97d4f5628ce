#!/bin/bash
# Interleaved A/B of libqdas builds over several workloads on ONE box: tools/ab_w.sh "<workloads>" <rounds> libA.so libB.so ...
WL=$1; R=$2; shift 2
for w in $WL; do
  for r in $(seq $R); do
    for L in "$@"; do
      echo -n "$w $(basename $L) "
      QDAS_LIB=$PWD/$L python bench.py --workload $w --steps 5 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])"
    done
  done
done | sort | awk '{a[$1" "$2]=a[$1" "$2]" "$3} END{for(k in a) print k, a[k]}' | sort
