#!/bin/bash
# Tail effect of the fused launch: c2-shaped steps whose tile count is / is not a whole number of rounds on 2048 wave slots
# (6144 tiles = 384 x 1024 units; 6250 = 400 x 1000; 6400 = 400 x 1024; 8192 = 512 x 1024), kernel-only durations from a trace.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for nf in "384 1024" "400 1000" "400 1024" "512 1024" "256 1024"; do
  set -- $nf
  rm -rf $R/gpurun_out/round
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/round -o t -- python3 $R/bench.py --quick --aois $1 --frames $2 --steps 20 --blocks 3 > $R/gpurun_out/round.log 2>&1
  f=$(find $R/gpurun_out/round -name "*kernel_trace.csv" | head -1)
  echo "== $1 x $2 = $(( $1 * $2 )) units, $(( ($1 * $2 + 63) / 64 )) tiles: $(grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/round.log | head -1)"
  python3 $R/scripts/prof_summary.py $f | grep -E "pixel_unit|sample_locals_tail"
done
rm -rf $R/gpurun_out/round
