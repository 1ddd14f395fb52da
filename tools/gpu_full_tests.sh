#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
t0=$(date +%s)
timeout 3000 python -m pytest tests -x -q -m gpu > $O/r6_tests_all.log 2>&1; echo "all gpu tests rc=$? $(( $(date +%s) - t0 )) s"
tail -6 $O/r6_tests_all.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
