#!/bin/bash
# full visit: gpu tests, smoke, bench with CPU baseline (the driver's end-of-round sequence)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -q -p no:cacheprovider > $R/gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $R/gpurun_out/pytest_gpu.log
timeout -k 10 200 python __graft_entry__.py smoke > $R/gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $R/gpurun_out/smoke.log
timeout -k 10 500 python bench.py > $R/gpurun_out/bench_full.log 2>&1; echo "bench rc=$?"; grep '^{' $R/gpurun_out/bench_full.log | tail -1
