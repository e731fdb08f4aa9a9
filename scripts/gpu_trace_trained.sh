#!/bin/bash
# Kernel trace of bench.py including its trained-regime leg; summaries of the initial regime (all launches before the
# fit) are in gpu_r03_profiles.sh -- here the LAST launches of each kernel (= after --trained-steps steps of fitting).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUTDIR:-trace_trained}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o bench -- python3 $R/bench.py --no-cpu --steps 20 --blocks 3 --trained-steps ${TRAINED:-4000} > $O/bench_traced.json 2> $O/bench_traced.err
echo "trace rc=$?"
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 $R/scripts/prof_summary.py $f --last 60 > $O/kernel_trace_summary_trained.txt
rm -rf $O/trace
cat $O/kernel_trace_summary_trained.txt
python3 -c "
import json,sys
d=json.loads([l for l in open('$O/bench_traced.json') if l.startswith('{')][-1])
print('step_ms', d['ms_per_step'], 'trained', d.get('trained_regime'))"
