#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench.  A step that is KILLED (timeout) stops the visit.
set -u
mkdir -p gpurun_out
run() {  # run <seconds> <logfile> <cmd...>
  local secs=$1 log=$2; shift 2
  timeout -k 10 "$secs" "$@" > "$log" 2>&1
  local rc=$?
  echo "[$(date +%T)] rc=$rc :: $*"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed -> stopping"; exit 99; fi
  return 0
}
python -c "import torch; print(torch.cuda.get_device_name(0), torch.cuda.device_count())"
run 420 gpurun_out/pytest_gpu.log python -m pytest tests -m gpu -q -p no:cacheprovider
tail -5 gpurun_out/pytest_gpu.log
run 200 gpurun_out/smoke.log python __graft_entry__.py smoke
tail -2 gpurun_out/smoke.log
run 400 gpurun_out/bench.log python bench.py --steps 10 --warmup 2 "$@"
tail -3 gpurun_out/bench.log
