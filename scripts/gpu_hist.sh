#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/prof_hist
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_hist -o bench -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --offsets hist > $R/gpurun_out/bench_hist.log 2>&1
echo "rc=$?"; grep '^{' $R/gpurun_out/bench_hist.log | tail -1 | cut -c1-600
