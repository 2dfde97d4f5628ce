#!/bin/bash
# Grid over tile / wave footprints of the tiled kernel on ONE box: tools/shape_grid.sh <workload> [rounds]
W=$1; R=${2:-1}
for r in $(seq $R); do
  for tz in 64 32 16 8; do for wz in 64 32 16 8 4; do
    [ $wz -gt $tz ] && continue
    [ $((tz / wz)) -gt 16 ] && continue
    echo -n "$W tz$tz wz$wz "
    QDAS_TILE_Z=$tz QDAS_WAVE_Z=$wz python bench.py --workload $W --steps 5 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['config']['fallback_tiles'])"
  done; done
  echo -n "$W auto "
  python bench.py --workload $W --steps 5 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['config']['fallback_tiles'], d['config'].get('tile'))"
done
