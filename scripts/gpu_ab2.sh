#!/bin/bash
# A/B two builds of the library on the same box (TAPQIR_AMD_LIB): full bench line without CPU / trained legs, 3 rounds interleaved
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TAPQIR_AMD_SITE_ROWS=6
for round in 1 2 3; do
  for lib in "$@"; do
    TAPQIR_AMD_LIB=$R/tapqir_amd/$lib timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu --trained-steps ${TRAINED:-0} 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d['roofline']
print('$lib', 'step_ms=%.4f'%d['ms_per_step'], 'fused_ms=%.4f'%r.get('step_kernel',{}).get('avg_launch_ms',float('nan')), 'pix_bwd_ms=%.4f'%r.get('logprob_kernel',r)['avg_launch_ms'], 'mb_ms=%.4f'%d['minibatch_10x512']['ms_per_step'], 'trained=%.4f'%d.get('trained_regime',{}).get('ms_per_step',float('nan')), 'fuse=',r.get('step_kernel',{}).get('kernel','')[-40:])"
  done
done
