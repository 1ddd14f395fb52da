#!/bin/bash
# does s_setprio around the MFMA block of syrk_mfma8_kernel<128> move the n = 60 000 dense solve?
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/lat_probe.hip -o build/lat_probe 2>/dev/null && build/lat_probe | tail -12
for p in 0 1 2 0 1 2; do echo "GSLAM_HIP_SYRK_PRIO=$p"; GSLAM_HIP_SYRK_PRIO=$p timeout 300 python tools/c5_solve_probe.py 60000 3 2>&1 | grep "n = "; done
