#!/bin/bash
# PMC passes over the bench (each --pmc set in its own run, with --kernel-trace only).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
if [ "${1:-}" = "list" ]; then rocprofv3 -L > $R/gpurun_out/pmc/counters.txt 2>&1; grep -c "" $R/gpurun_out/pmc/counters.txt; exit 0; fi
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc -o pass$i -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu > $R/gpurun_out/pmc/pass$i.log 2>&1
  echo "pass$i ($set) rc=$?"
done
ls $R/gpurun_out/pmc
