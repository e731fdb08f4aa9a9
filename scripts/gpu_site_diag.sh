#!/bin/bash
# Where the local guide-sampling kernel spends its time: the same kernel with one part of the per-site terms compiled out
# (diagnostic builds of tq_cosmos.hip: -DTQ_DIAG_NO_BETAGRAD / _NO_BETALP / _NO_GAMMAGRAD), at the initial parameters
# (scripts/site_breakdown.py) and in the trained regime (scripts/site_trained.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in ${VARIANTS:-base NO_BETAGRAD NO_BETALP NO_GAMMAGRAD}; do
  lib=tapqir_amd/libtapqir_hip_diag_$v.so
  [ -f $lib ] || continue
  echo "== $v"
  [ -n "${SKIP_INIT:-}" ] || TAPQIR_AMD_LIB=$lib timeout -k 10 200 python3 scripts/site_breakdown.py 2>/dev/null | grep "^init"
  TAPQIR_AMD_LIB=$lib timeout -k 10 300 python3 scripts/site_trained.py 2>/dev/null | grep "^trained\|^size"
done
