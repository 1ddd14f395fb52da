#!/bin/bash
# ORB parity tests + per-kernel timing of the built library against build/ab/libgslam_hip_prev.so (gpurun -- bash tools/r6_ab.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_orb_gpu.py tests/test_orb_adversarial_gpu.py tests/test_orb_stream_gpu.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2 3; do
for v in new prev; do
  lib=$R/gslam_amd/lib/libgslam_hip.so; [ $v = prev ] && lib=$R/build/ab/libgslam_hip_prev.so
  [ -f $lib ] || continue
  for n in 400 1000; do
    echo "$v frames=$n: $(GSLAM_HIP_LIB=$lib timeout 120 python $R/tools/orb_perf.py $n 2>&1 | grep -E 'extract|orb_fast_cells|orb_describe' | tail -3 | tr -s ' ' | tr '\n' '|')"
  done
done
done
[ -n "$VALU" ] && bash tools/valu_ab.sh
