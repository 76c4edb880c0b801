#!/bin/bash
# Phase profile of the BDF kernel.  Step 1 (here, no GPU needed):  bash tools/bdf_phase_profile.sh build
#                                   Step 2 (on the GPU box):        bash tools/bdf_phase_profile.sh run
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/ivp_amd/csrc
if [ "${1:-build}" = build ]; then
  make -C $C -j8 > /dev/null
  mkdir -p $R/exp_libs
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DIVP_FAST=0 -ffp-contract=off -DIVP_PHASE_PROF \
      -I$R/include -c $C/rk_bdf.hip -o $R/exp_libs/rk_bdf_strict_prof.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $C/rk_strict.o $C/rk_fast.o $C/rk_strict_h.o $C/rk_fast_h.o $C/rk_group_strict.o \
      $C/rk_group_fast.o $R/exp_libs/rk_bdf_strict_prof.o $C/rk_bdf_fast.o $C/rk_bdf_strict_occ2.o $C/rk_bdf_fast_occ2.o $C/ivp_capi.o $C/ivp_log.o \
      $C/log_gather.o $C/ivp_jit.o -o $R/exp_libs/libivp_hip_prof.so -L/opt/rocm/lib -lhiprtc -Wl,-rpath,/opt/rocm/lib
  echo built $R/exp_libs/libivp_hip_prof.so
else
  python3 -c "import torch"
  python3 $R/tools/bdf_phase_profile.py
  B=256 IVP_TUNE_BDF_LPW=1 python3 $R/tools/bdf_phase_profile.py
fi
