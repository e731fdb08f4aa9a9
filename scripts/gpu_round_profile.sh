#!/bin/bash
# Round profile: (1) rocprofv3 --kernel-trace --stats of the default bench command, (2) PMC passes over the
# pixel-kernel micro-benchmark, each counter set in its own run with --kernel-trace only (FETCH_SIZE and
# WRITE_SIZE do not fit one pass).  Summaries land in gpurun_out/round/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/round
mkdir -p $OUT/prof $OUT/pmc
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu > $OUT/bench.log 2>&1
echo "kernel-trace rc=$?"
grep '^{' $OUT/bench.log | tail -1 > $OUT/bench.json
python3 $R/scripts/prof_summary.py $OUT/prof/bench_kernel_trace.csv > $OUT/kernel_trace_summary.txt 2>&1
cat $OUT/kernel_trace_summary.txt
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_INSTS_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc -o pass$i -- python3 $R/scripts/pix_bench.py --launches 4 > $OUT/pmc/pass$i.log 2>&1
  echo "pmc pass$i ($set) rc=$?"
done
python3 $R/scripts/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.txt 2>&1
grep -A16 "il2" $OUT/pmc_summary.txt | head -60
