#!/bin/bash
# A/B an environment toggle on one box: usage gpu_ab_env.sh VAR val1 val2 ...
set -u
var=$1; shift
for round in 1 2 3; do
  for v in "$@"; do
    env $var=$v timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$var=$v', 'step_ms=%.4f'%d['ms_per_step'], 'pix_bwd_ms=%.4f'%d['roofline']['avg_launch_ms'], 'pix_fwd_ms=%.4f'%d['roofline']['forward_only']['avg_launch_ms'], 'mb_ms=%.4f'%d['minibatch_10x512']['ms_per_step'])"
  done
done
