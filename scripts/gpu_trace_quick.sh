#!/bin/bash
# Kernel trace of `bench.py --quick` (headline steps only) for each environment setting given as an argument
# ("NAME=VALUE[,NAME=VALUE...]" or "-"): median kernel durations of the step's launches.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  envs=""
  [ "$cfg" != "-" ] && envs=$(echo $cfg | tr ',' ' ')
  O=$R/gpurun_out/tq_$(echo $cfg | tr -c 'A-Za-z0-9_\n' '_')
  rm -rf $O; mkdir -p $O
  for e in $envs; do export $e; done
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o bench -- python3 $R/bench.py --quick --steps 20 --blocks 3 ${BENCHARGS:-} > $O/bench.json 2> $O/bench.err
  for e in $envs; do unset ${e%%=*}; done
  f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
  echo "== $cfg: $(python3 -c "import json;print(json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1])['ms_per_step'])")"
  python3 $R/scripts/prof_summary.py $f --last 60 | grep -E "kernel |tail_kernel|pixel_unit|sample_locals_kernel|unit_rows|il2"
  rm -rf $O/trace
done
