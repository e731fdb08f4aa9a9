#!/bin/bash
# GPU visit: parity tests (stop on failure), then same-box A/B timing of library builds given as arguments
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $R/gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -6 $R/gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then exit 98; fi
bash $R/scripts/gpu_ab.sh "$@"
