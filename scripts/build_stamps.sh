#!/bin/bash
# Diagnostic builds of the library with the phase stamps of tq_minibatch_kernel compiled in for ONE workgroup:
#   scripts/build_stamps.sh 5 300   ->  tapqir_amd/libtapqir_hip_stamps_5.so, ..._300.so  (read them with
#   TAPQIR_AMD_LIB=... STAMPS=<blk> python scripts/mb_dev_time.py)
set -e
cd "$(dirname "$0")/.."
python -m tapqir_amd.build >/dev/null
for blk in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -DTQ_MB_STAMPS=$blk -DTQ_MB_STAMPS_SITES=${SITES:-0} -c tapqir_amd/csrc/tq_cosmos.hip \
    -o tapqir_amd/build/tq_cosmos_stamps_$blk${SITES:+_s$SITES}.o &
done
wait
for blk in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tapqir_amd/libtapqir_hip_stamps_$blk${SITES:+_s$SITES}.so tapqir_amd/build/tq_ksmogn.o \
    tapqir_amd/build/tq_xtalk.o tapqir_amd/build/tq_cosmos_stamps_$blk${SITES:+_s$SITES}.o tapqir_amd/build/tq_glimpse.o tapqir_amd/build/tq_aux.o
done
