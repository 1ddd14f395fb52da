#!/bin/bash
# orb_fast_cells: one tile per workgroup against persistent workgroups (GSLAM_HIP_ORB_PERSIST = 0 one tile per workgroup, 3 tile loop
# without prefetch, 1 tile loop with the next tile in flight); LIBS = library builds under build/ab to compare.
# gpurun -- bash tools/r6_persist_ab.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
LIBS=${LIBS:-"n7 n8"}
for lib in $LIBS; do
  L=$R/build/ab/libgslam_hip_$lib.so; [ $lib = lib ] && L=$R/gslam_amd/lib/libgslam_hip.so
  for pv in 1 3; do
    echo "tests $lib persist=$pv: $(GSLAM_HIP_LIB=$L GSLAM_HIP_ORB_PERSIST=$pv timeout 900 python -m pytest tests/test_orb_gpu.py tests/test_orb_adversarial_gpu.py -x -q -m gpu 2>&1 | tail -1)"
  done
done
for rep in 1 2; do
for lib in $LIBS; do
  L=$R/build/ab/libgslam_hip_$lib.so; [ $lib = lib ] && L=$R/gslam_amd/lib/libgslam_hip.so
  for v in "0 0" "0 2500" "3 0" "1 0"; do
    set -- $v
    for n in 400 1000; do
      echo "$lib persist=$1 ldspad=$2 frames=$n: $(GSLAM_HIP_ORB_LDSPAD=$2 GSLAM_HIP_ORB_PERSIST=$1 GSLAM_HIP_LIB=$L timeout 120 python $R/tools/orb_perf.py $n 2>&1 | grep -E 'extract|orb_fast_cells' | tail -2 | tr -s ' ' | tr '\n' '|')"
    done
  done
done
done
