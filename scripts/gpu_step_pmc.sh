#!/bin/bash
# PMC passes over a short run of the bench itself (all kernels of the step; each counter set in its own run,
# --kernel-trace only).  Output: gpurun_out/steppmc/summary.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${STEPOUT:-steppmc}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT -o pass$i -- python3 $R/bench.py --quick --steps 4 --blocks 1 --warmup 2 ${STEPARGS:-} > $OUT/pass$i.log 2>&1
  echo "pass$i ($set) rc=$?"
done
python3 $R/scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +2M -delete
grep -A16 "${STEPGREP:-^tq_unit_rows\|^tq_sample_locals_tail}" $OUT/summary.txt
