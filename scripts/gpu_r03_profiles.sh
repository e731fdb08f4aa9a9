#!/bin/bash
# Round-3 evidence, every file from runs of bench.py itself: kernel traces (initial regime, trained regime, minibatch with a 50-value
# offset histogram), PMC passes over the step (c2 initial + trained regime, the other configs), PMC passes of the stand-alone
# log-prob kernel, bench lines.  Outputs under gpurun_out/r03/; copy what is to be judged into profiles/ (r03_*).
# SECTIONS="trace pmc pix bench" selects parts.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
S=${SECTIONS:-trace pmc pix bench}
cd /tmp && export TMPDIR=/tmp
trace() {  # name, bench args...
  local name=$1; shift
  rm -rf $O/trace_$name
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -o bench -- python3 $R/bench.py "$@" > $O/bench_traced_$name.json 2> $O/bench_traced_$name.err
  echo "trace $name rc=$?"
  cp $(find $O/trace_$name -name "*kernel_trace.csv" | head -1) $O/kernel_trace_$name.csv
  cp $(find $O/trace_$name -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$name.csv
  rm -rf $O/trace_$name
}
if [[ $S == *trace* ]]; then
  trace c2 --quick --steps 20 --blocks 5
  python3 $R/scripts/prof_summary.py $O/kernel_trace_c2.csv > $O/kernel_trace_summary.txt
  trace c2_trained --quick --steps 20 --blocks 5 --pre-steps 4000
  python3 $R/scripts/prof_summary.py $O/kernel_trace_c2_trained.csv --last 100 > $O/kernel_trace_summary_trained.txt
  trace c2_hist50 --offsets hist --no-cpu --trained-steps 0 --steps 10 --blocks 3
  python3 $R/scripts/prof_summary.py $O/kernel_trace_c2_hist50.csv > $O/kernel_trace_summary_hist50.txt
  trace c3 --quick --config c3 --steps 10 --blocks 3
  python3 $R/scripts/make_trace_json.py $O/kernel_trace.json c2=$O/kernel_trace_c2.csv c2_trained=$O/kernel_trace_c2_trained.csv:100 \
      c2_hist50=$O/kernel_trace_c2_hist50.csv c3=$O/kernel_trace_c3.csv
  rm -f $O/kernel_trace_c2_trained.csv $O/kernel_trace_c2_hist50.csv $O/kernel_trace_c3.csv
  find $O -name "kernel_trace_*.csv" -size +8M -delete
fi
if [[ $S == *pmc* ]]; then
  STEPOUT=r03/steppmc_c2 bash $R/scripts/gpu_step_pmc.sh > $O/steppmc_c2.log 2>&1; cp $O/steppmc_c2/summary.txt $O/pmc_summary_step.txt
  STEPOUT=r03/steppmc_trained STEPARGS="--pre-steps 4000" bash $R/scripts/gpu_step_pmc.sh > $O/steppmc_trained.log 2>&1; cp $O/steppmc_trained/summary.txt $O/pmc_summary_step_trained.txt
  STEPOUT=r03/steppmc_c3 STEPARGS="--config c3" bash $R/scripts/gpu_step_pmc.sh > $O/steppmc_c3.log 2>&1; cp $O/steppmc_c3/summary.txt $O/pmc_summary_step_c3shard.txt
  STEPOUT=r03/steppmc_c4 STEPARGS="--config c4" bash $R/scripts/gpu_step_pmc.sh > $O/steppmc_c4.log 2>&1; cp $O/steppmc_c4/summary.txt $O/pmc_summary_step_c4.txt
  STEPOUT=r03/steppmc_c5 STEPARGS="--config c5" bash $R/scripts/gpu_step_pmc.sh > $O/steppmc_c5.log 2>&1; cp $O/steppmc_c5/summary.txt $O/pmc_summary_step_c5shard.txt
  STEPOUT=r03/steppmc_hist STEPARGS="--offsets hist --profile-minibatch 100" bash $R/scripts/gpu_step_pmc.sh > $O/steppmc_hist.log 2>&1; cp $O/steppmc_hist/summary.txt $O/pmc_summary_step_hist50.txt  # (full-batch steps + 100 minibatch steps 10 x 512)
  cd $R
  python3 scripts/make_step_traffic.py $O/pmc_step_traffic.json c2=$O/pmc_summary_step.txt:400000 c2_trained=$O/pmc_summary_step_trained.txt:400000 \
      c3=$O/pmc_summary_step_c3shard.txt:1600000 \
      c4=$O/pmc_summary_step_c4.txt:800000:tq_sample_locals_tail_kernel,tq_xtalk_il_kernel,tq_unit_rows_kernel \
      c5=$O/pmc_summary_step_c5shard.txt:500000:tq_sample_locals_tail_kernel,tq_ksmogn_il2_kernel,tq_unit_rows_kernel > $O/pmc_step_traffic.log
  python3 scripts/make_pmc_valu.py $O/pmc_valu.json c2:$O/pmc_summary_step.txt:tq_pixel_unit_kernel:400000 c2_sampling:$O/pmc_summary_step.txt:tq_sample_locals_tail_kernel:400000 \
      c2_trained_sampling:$O/pmc_summary_step_trained.txt:tq_sample_locals_tail_kernel:400000 \
      c4:$O/pmc_summary_step_c4.txt:tq_xtalk_il_kernel:400000 c5:$O/pmc_summary_step_c5shard.txt:tq_ksmogn_il2_kernel:500000 hist:$O/pmc_summary_step_hist50.txt:tq_ksmogn_il2m_kernel:400000 > /dev/null 2> $O/pmc_valu.err
  rm -rf $O/steppmc_*
  cd /tmp
fi
if [[ $S == *pix* ]]; then
  PIXARGS="" TAPQIR_AMD_PERSIST=0 bash $R/scripts/gpu_pix_pmc.sh "FETCH_SIZE" "WRITE_SIZE" > $O/pixpmc.log 2>&1
  cp $R/gpurun_out/pixpmc/summary.txt $O/pmc_summary_logprob.txt
  cd $R; python3 scripts/make_pmc_traffic.py $O/pmc_traffic.json 2 14 400000 $O/pmc_summary_logprob.txt > /dev/null; cd /tmp
fi
if [[ $S == *bench* ]]; then
  cd $R
  timeout -k 10 500 python3 bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"
  timeout -k 10 300 python3 bench.py --config c1 --no-cpu --trained-steps 0 > $O/bench_c1.json 2>/dev/null; echo "c1 rc=$?"
  timeout -k 10 300 python3 bench.py --config c3 --no-cpu --trained-steps 0 > $O/bench_c3shard.json 2>/dev/null; echo "c3 rc=$?"
  timeout -k 10 300 python3 bench.py --config c4 --no-cpu --trained-steps 0 > $O/bench_c4.json 2>/dev/null; echo "c4 rc=$?"
  timeout -k 10 300 python3 bench.py --config c5 --no-cpu --trained-steps 0 > $O/bench_c5shard.json 2>/dev/null; echo "c5 rc=$?"
  timeout -k 10 300 python3 bench.py --offsets hist --no-cpu --trained-steps 0 > $O/bench_c2_hist50.json 2>/dev/null; echo "hist rc=$?"
  timeout -k 10 300 python3 bench.py --force-dist --quick > $O/bench_c2_forcedist.json 2>/dev/null; echo "forcedist rc=$?"
fi
ls -la $O
