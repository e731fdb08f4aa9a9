#!/bin/bash
# gpurun_out/r03/* (scripts/gpu_r03_profiles.sh) -> profiles/r03_*: what the bench line and DESIGN.md cite
set -eu
R=$(cd $(dirname $0)/.. && pwd)
O=$R/gpurun_out/r03; P=$R/profiles
cp $O/kernel_stats_c2.csv $P/r03_kernel_stats.csv
cp $O/kernel_stats_c2_trained.csv $P/r03_kernel_stats_trained.csv
cp $O/kernel_stats_c2_hist50.csv $P/r03_kernel_stats_hist50.csv
for f in kernel_trace_summary.txt kernel_trace_summary_trained.txt kernel_trace_summary_hist50.txt kernel_trace.json pmc_summary_step.txt \
         pmc_summary_step_trained.txt pmc_summary_step_c3shard.txt pmc_summary_step_c4.txt pmc_summary_step_c5shard.txt pmc_summary_step_hist50.txt \
         pmc_step_traffic.json pmc_traffic.json pmc_summary_logprob.txt pmc_valu.json; do
  [ -f $O/$f ] && cp $O/$f $P/r03_$f
done
for c in c1 c2 c3shard c4 c5shard c2_hist50 c2_forcedist; do
  [ -s $O/bench_$c.json ] && grep '^{' $O/bench_$c.json | tail -1 > $P/r03_bench_${c}_1gpu.json
done
# the summaries' own paths are those of the GPU box: name the committed files instead
python3 - <<PY
import json
p = "$P/r03_pmc_step_traffic.json"
d = json.load(open(p))
names = {"c2": "r03_pmc_summary_step.txt", "c2_trained": "r03_pmc_summary_step_trained.txt", "c3": "r03_pmc_summary_step_c3shard.txt",
         "c4": "r03_pmc_summary_step_c4.txt", "c5": "r03_pmc_summary_step_c5shard.txt"}
for c, v in d["configs"].items():
    v["summary"] = "profiles/" + names.get(c, "")
json.dump(d, open(p, "w"), indent=1)
PY
ls $P | grep r03
