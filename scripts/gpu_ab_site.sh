#!/bin/bash
# same-box A/B of the local-sampling kernel for library builds given as arguments
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
  for lib in "$@"; do
    echo -n "$lib: "
    TAPQIR_AMD_LIB=$R/tapqir_amd/$lib timeout -k 10 200 python $R/scripts/site_bench.py 30 2>&1 | grep -E "ms/step|sample_locals" | tr '\n' ' '
    echo
  done
done
