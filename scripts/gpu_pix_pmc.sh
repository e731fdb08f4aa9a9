#!/bin/bash
# PMC passes over the pixel-kernel micro-benchmark (each counter set in its own run, --kernel-trace only)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pixpmc
mkdir -p $OUT
python3 $R/scripts/pix_bench.py $PIXARGS
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT -o pass$i -- python3 $R/scripts/pix_bench.py --launches 4 $PIXARGS > $OUT/pass$i.log 2>&1
  echo "pass$i ($set) rc=$?"
done
python3 $R/scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
grep -A40 "il2" $OUT/summary.txt | head -120
