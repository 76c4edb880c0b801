#!/bin/bash
# Register / scratch / occupancy table of every kernel in libivp_hip.so's translation units (hipcc remarks, no GPU needed).
# usage: tools/kernel_resources.sh > profiles/rNN_kernel_resources.txt
TOOLS="$(cd "$(dirname "$0")" && pwd)"
cd "$TOOLS/../ivp_amd/csrc" || exit 1
COMMON="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off -Rpass-analysis=kernel-resource-usage -c -o /dev/null"
run() {  # name, source, flags...
    local name=$1 src=$2; shift 2
    /opt/rocm/bin/hipcc $COMMON "$@" "$src" 2>&1 | python3 "$TOOLS/kernel_resources.py" "$name"
}
run strict      rk_kernels.hip -DIVP_FAST=0 &
run fma         rk_kernels.hip -DIVP_FAST=1 &
run strict_res  rk_kernels.hip -DIVP_FAST=0 -DIVP_HOIST=1 -DIVP_MIN_WAVES=1 &
run fma_res     rk_kernels.hip -DIVP_FAST=1 -DIVP_HOIST=1 -DIVP_MIN_WAVES=1 &
wait
run bdf_strict  rk_bdf.hip -DIVP_FAST=0 &
run bdf_fma     rk_bdf.hip -DIVP_FAST=1 &
run group_strict rk_group.hip -DIVP_FAST=0 &
run group_fma   rk_group.hip -DIVP_FAST=1 &
wait
