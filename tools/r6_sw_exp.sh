#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python tools/r6_sw_exp.py 2>&1 | grep -v amdgpu.ids > $O/r6_sw_exp.log
cat $O/r6_sw_exp.log
GSLAM_HIP_ORB_PLANE_SW=1 timeout 900 python -m pytest tests/test_orb_gpu.py tests/test_orb_adversarial_gpu.py -x -q -m gpu -k "quadtree or slam or plane" > $O/r6_sw_tests.log 2>&1; echo "sw tests rc=$?"; tail -3 $O/r6_sw_tests.log
