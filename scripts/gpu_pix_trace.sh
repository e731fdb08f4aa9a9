#!/bin/bash
# rocprofv3 kernel trace of the pixel-kernel micro-benchmark; env passes through
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pixtrace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pixtrace -o pix -- python3 $R/scripts/pix_bench.py --launches 30 $PIXARGS > $R/gpurun_out/pixtrace.log 2>&1
f=$(find $R/gpurun_out/pixtrace -name "*kernel_trace.csv" | head -1)
python3 $R/scripts/prof_summary.py $f | grep "ksmogn\|kernel  "
rm -rf $R/gpurun_out/pixtrace
