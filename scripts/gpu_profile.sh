#!/bin/bash
# rocprofv3 kernel trace of the bench (no PMC in this pass).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu "$@" > $R/gpurun_out/prof_bench.log 2>&1
echo "rocprof rc=$?"
tail -2 $R/gpurun_out/prof_bench.log
find $R/gpurun_out/prof -name "*stats*" | head
