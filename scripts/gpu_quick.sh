#!/bin/bash
# quick GPU visit: parity tests, then the bench under rocprofv3 --kernel-trace (no CPU baseline leg)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/prof
timeout -k 10 400 python -m pytest tests -m gpu -q -p no:cacheprovider > $R/gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 $R/gpurun_out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 99; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu "$@" > $R/gpurun_out/prof_bench.log 2>&1
echo "rocprof rc=$?"
grep '^{' $R/gpurun_out/prof_bench.log | tail -1
