#!/bin/bash
# The bench lines of the round (no profiler): c2 (default), c3 / c5 per-GPU shards, c4, c2 with an offset histogram, c2 through
# the sharded launch sequence.  Outputs under gpurun_out/r02/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
timeout -k 10 400 python3 bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"
timeout -k 10 300 python3 bench.py --config c3 --no-cpu --trained-steps 0 > $O/bench_c3shard.json 2>/dev/null; echo "c3 rc=$?"
timeout -k 10 300 python3 bench.py --config c4 --no-cpu --trained-steps 0 > $O/bench_c4.json 2>/dev/null; echo "c4 rc=$?"
timeout -k 10 300 python3 bench.py --config c5 --no-cpu --trained-steps 0 > $O/bench_c5shard.json 2>/dev/null; echo "c5 rc=$?"
timeout -k 10 300 python3 bench.py --offsets hist --no-cpu --trained-steps 0 > $O/bench_c2_hist50.json 2>/dev/null; echo "hist rc=$?"
timeout -k 10 300 python3 bench.py --force-dist --quick > $O/bench_c2_forcedist.json 2>/dev/null; echo "forcedist rc=$?"
