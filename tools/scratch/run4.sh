cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for w in c1 c2 c5; do timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu 2>/dev/null | tail -1 > gpurun_out/bench_$w.json; cat gpurun_out/bench_$w.json | cut -c1-400; done
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_c3.json; cut -c1-300 gpurun_out/bench_c3.json
