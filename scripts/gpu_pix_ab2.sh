#!/bin/bash
# same-box A/B of pixel-kernel variants, several alternating rounds: arguments = "LIB PERSIST REM" triples
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3 4; do
  for cfg in "$@"; do
    read lib per rem <<< "$cfg"
    echo -n "$lib persist=$per rem=$rem: "
    TAPQIR_AMD_LIB=$R/tapqir_amd/$lib TAPQIR_AMD_PERSIST=$per TAPQIR_AMD_PERSIST_REM=$rem timeout -k 10 200 python $R/scripts/pix_bench.py --launches 100 2>&1 | grep "bwd=1" | sed 's/pixel kernel K=2 P=14 units=400000 //'
  done
done
