#!/bin/bash
# Round-2 evidence: kernel trace + stats of the bench, PMC passes of both forms of the backward pixel kernel, bench lines
# of the other BASELINE configs.  Outputs under gpurun_out/r02/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats of the default bench (no CPU leg)
rm -rf $O/trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 $R/bench.py --no-cpu --steps 10 --blocks 3 --trained-steps 0 > $O/bench_traced.json 2> $O/bench_traced.err
echo "trace rc=$?"
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 $R/scripts/prof_summary.py $f > $O/kernel_trace_summary.txt
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/trace
# 2. PMC passes, both forms
for form in 0 1; do
  export TAPQIR_AMD_PERSIST=$form
  PIXARGS="" bash $R/scripts/gpu_pix_pmc.sh "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU" > $O/pixpmc_form$form.log 2>&1
  cp $R/gpurun_out/pixpmc/summary.txt $O/pmc_summary_form$form.txt
done
unset TAPQIR_AMD_PERSIST
cd $R
python3 scripts/make_pmc_traffic.py $O/pmc_traffic.json 2 14 400000 $O/pmc_summary_form0.txt $O/pmc_summary_form1.txt > /dev/null
# 2b. PMC passes over the step itself (every kernel of a c2 step, fused pixel + per-unit launch included)
STEPOUT=steppmc bash $R/scripts/gpu_step_pmc.sh > $O/steppmc.log 2>&1
cp $R/gpurun_out/steppmc/summary.txt $O/pmc_summary_step.txt
cd $R
# 3. bench lines
timeout -k 10 400 python3 bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"
timeout -k 10 300 python3 bench.py --config c3 --no-cpu --trained-steps 0 > $O/bench_c3shard.json 2>/dev/null; echo "c3 rc=$?"
timeout -k 10 300 python3 bench.py --config c4 --no-cpu --trained-steps 0 > $O/bench_c4.json 2>/dev/null; echo "c4 rc=$?"
timeout -k 10 300 python3 bench.py --config c5 --no-cpu --trained-steps 0 > $O/bench_c5shard.json 2>/dev/null; echo "c5 rc=$?"
timeout -k 10 300 python3 bench.py --offsets hist --no-cpu --trained-steps 0 > $O/bench_c2_hist50.json 2>/dev/null; echo "hist rc=$?"
timeout -k 10 300 python3 bench.py --force-dist --quick > $O/bench_c2_forcedist.json 2>/dev/null; echo "forcedist rc=$?"
ls -la $O
