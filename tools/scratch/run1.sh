set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "transmit_slabs or pixel_shards" 2>&1 | tail -5
QDAS_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/bench_share2.log 2>&1
tail -3 gpurun_out/bench_share2.log
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu 2>&1 | tail -1
