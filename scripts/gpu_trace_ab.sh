#!/bin/bash
# rocprofv3 kernel trace of `bench.py --quick` for each "PERSIST ROWS" pair given as arguments (same box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  set -- $cfg
  tag=p$1r$2
  rm -rf $R/gpurun_out/prof_$tag
  TAPQIR_AMD_PERSIST=$1 TAPQIR_AMD_ROWS=$2 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_$tag -o bench -- python3 $R/bench.py --quick --steps 10 --blocks 3 > $R/gpurun_out/prof_$tag.log 2>&1
  echo "== $tag rc=$?"
  f=$(find $R/gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1)
  python3 $R/scripts/prof_summary.py $f | grep -v "glimpse\|interleave\|image_stats" > $R/gpurun_out/trace_$tag.txt
  cat $R/gpurun_out/trace_$tag.txt
  rm -rf $R/gpurun_out/prof_$tag
done
