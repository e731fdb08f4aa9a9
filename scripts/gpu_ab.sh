#!/bin/bash
# A/B two builds of the library on the same box: bench without CPU leg, 3 rounds interleaved
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do
  for lib in "$@"; do
    TAPQIR_AMD_LIB=$R/tapqir_amd/$lib timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$lib', 'step_ms=%.4f'%d['ms_per_step'], 'pix_bwd_ms=%.4f'%d['roofline']['avg_launch_ms'], 'pix_fwd_ms=%.4f'%d['roofline']['forward_only']['avg_launch_ms'], 'mb_ms=%.4f'%d['minibatch_10x512']['ms_per_step'])"
  done
done
