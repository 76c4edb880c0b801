#!/bin/bash
# Where a large-n BDF step spends its cycles (one wavefront per trajectory, bdf_group.h): builds rk_group.hip with
# -DIVP_PHASE_CLOCKS into build_clk/libivp_hip_clk.so (a MEASURING library: never shipped, never loaded by the tests) --
# run this part in the build container, after `make -C ivp_amd/csrc`:
#   bash tools/phase_clocks_large_n.sh build
# -- and on the GPU box run one N = 100 system through it; the kernel prints shader-clock cycles per phase:
#   bash tools/phase_clocks_large_n.sh run [variant] [decay100|dense64]     (variant 2: factors in LDS, 1: in global memory)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
    cd $R/ivp_amd/csrc && mkdir -p $R/build_clk
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DIVP_FAST=0 -DIVP_PHASE_CLOCKS -ffp-contract=off \
        -c rk_group.hip -o $R/build_clk/rk_group_strict.o
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 rk_strict.o rk_fast.o rk_strict_h.o rk_fast_h.o $R/build_clk/rk_group_strict.o rk_group_fast.o \
        rk_bdf_strict.o rk_bdf_fast.o rk_bdf_strict_occ2.o rk_bdf_fast_occ2.o ivp_capi.o ivp_log.o log_gather.o ivp_jit.o \
        -o $R/build_clk/libivp_hip_clk.so -L/opt/rocm/lib -lhiprtc -Wl,-rpath,/opt/rocm/lib
    rm -f $R/build_clk/*.o
else
    IVP_AMD_LIB=$R/build_clk/libivp_hip_clk.so python3 $R/tools/time_large_n_one.py 1 ${2:-2} ${3:-decay100} 2>&1 | tail -3
fi
