#!/bin/bash
# Kernel trace of a minibatch fit through Model.run (20 000 iterations of 10 x 512 on the c2 data set, no checkpoint files):
# what the single-launch minibatch step costs over a whole fit, from the initial to the converged parameter regime.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/fittrace
RATE_ITERS=20000 RATE_NO_CKPT=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fittrace -o fit -- python3 $R/scripts/run_rate.py > $O/fit_traced.log 2>&1
echo "rc=$?"
f=$(find $O/fittrace -name "*kernel_trace.csv" | head -1)
python3 $R/scripts/prof_summary.py $f > $O/kernel_trace_summary_fit.txt
rm -rf $O/fittrace
tail -2 $O/fit_traced.log
grep "minibatch\|kernel " $O/kernel_trace_summary_fit.txt
