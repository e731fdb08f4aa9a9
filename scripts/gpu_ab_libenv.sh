#!/bin/bash
# A/B of (library, environment) pairs on one box: arguments "lib.so[:NAME=VALUE[,NAME=VALUE]]"; 3 rounds interleaved.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do
  for cfg in "$@"; do
    lib=${cfg%%:*}; envs=""
    [ "$cfg" != "$lib" ] && envs=$(echo ${cfg#*:} | tr ',' ' ')
    env TAPQIR_AMD_LIB=$R/tapqir_amd/$lib $envs timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu --trained-steps ${TRAINED:-0} 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$cfg', 'step_ms=%.4f'%d['ms_per_step'], 'mb_ms=%.4f'%d['minibatch_10x512']['ms_per_step'], 'trained=%.4f'%d.get('trained_regime',{}).get('ms_per_step',float('nan')))"
  done
done
