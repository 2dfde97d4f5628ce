cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_convd.py -q -m gpu -x 2>&1 | tail -15
timeout 600 python tools/convd_time.py > gpurun_out/convd_time.txt 2>&1; cat gpurun_out/convd_time.txt
