#!/bin/bash
# rocprofv3 kernel trace of the bench's minibatch leg; prints the launch timeline of a few minibatch steps
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/mbtrace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/mbtrace -o mb -- python3 $R/bench.py --no-cpu --trained-steps 0 --steps 5 --blocks 1 > $R/gpurun_out/mbtrace.log 2>&1
f=$(find $R/gpurun_out/mbtrace -name "*kernel_trace.csv" | head -1)
python3 $R/scripts/mb_timeline.py $f
rm -rf $R/gpurun_out/mbtrace
