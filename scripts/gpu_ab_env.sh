#!/bin/bash
# A/B of environment switches on the same box and the same library: each argument is "NAME=VALUE[,NAME=VALUE...]" (or "-" for
# the defaults); full bench line without the CPU leg, 3 rounds interleaved.  TRAINED=4000 adds the trained-regime leg.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do
  for cfg in "$@"; do
    envs=""
    [ "$cfg" != "-" ] && envs=$(echo $cfg | tr ',' ' ')
    env $envs timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu --trained-steps ${TRAINED:-0} 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d['roofline']
print('$cfg', 'step_ms=%.4f'%d['ms_per_step'], 'mb_ms=%.4f'%d['minibatch_10x512']['ms_per_step'], 'trained=%.4f'%d.get('trained_regime',{}).get('ms_per_step',float('nan')), 'dominant_ms=%.4f'%r['avg_launch_ms'])"
  done
done
