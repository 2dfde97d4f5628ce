cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do
for cfg in "fold_only::QDAS_NO_MIRROR=1" "mirror_only:--no-reciprocal:" "no_symmetry:--no-reciprocal:QDAS_NO_MIRROR=1" "headline::"; do
  IFS=: read name args envs <<< "$cfg"
  for ss in 0 1; do
    e="$envs"; [ $ss = 1 ] && e="$e QDAS_NO_STAGE_SHAPE=1"
    ms=$(env $e timeout 300 python bench.py $args --no-cpu --no-traffic --no-general --steps 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['kernel'][:70])")
    echo "$name no_stage_shape=$ss $ms"
  done
done; done
