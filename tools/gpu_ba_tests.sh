#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
t0=$(date +%s)
timeout 2400 python -m pytest tests/test_ba_order_gpu.py tests/test_cr_solver.py tests/test_full_configs_gpu.py tests/test_ba_gpu.py tests/test_graph_gpu.py tests/test_calib_gpu.py -x -q -m gpu > $O/r6_tests2.log 2>&1; echo "tests2 rc=$? $(( $(date +%s) - t0 )) s"
tail -8 $O/r6_tests2.log
