#!/bin/bash
# Interleaved A/B of libqdas builds on ONE GPU box (cross-call variance is ~4%): tools/ab.sh <rounds> libA.so libB.so ...
R=$1; shift
for r in $(seq $R); do
  for L in "$@"; do
    echo -n "$(basename $L) "
    QDAS_LIB=$PWD/$L python bench.py --steps 5 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])"
  done
done | sort | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print k, a[k]}'
