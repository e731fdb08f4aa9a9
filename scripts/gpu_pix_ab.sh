#!/bin/bash
# same-box A/B of the pixel kernel (HIP-event timing, scripts/pix_bench.py): arguments = "LIB PERSIST" pairs
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
  for cfg in "$@"; do
    lib=${cfg% *}; per=${cfg#* }
    echo -n "$lib persist=$per: "
    TAPQIR_AMD_LIB=$R/tapqir_amd/$lib TAPQIR_AMD_PERSIST=$per timeout -k 10 200 python $R/scripts/pix_bench.py --launches 50 2>&1 | grep "bwd=1" | sed 's/pixel kernel K=2 P=14 //'
  done
done
