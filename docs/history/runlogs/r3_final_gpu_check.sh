mkdir -p gpurun_out
F1="tests/test_pg_gpu.py tests/test_graph_gpu.py tests/test_bsparse.py tests/test_ransac_gpu.py tests/test_bow_gpu.py"
timeout 220 python -m pytest $F1 -m gpu -q --tb=short > gpurun_out/t1.log 2>&1; echo "t1 rc=$?" > gpurun_out/rc.log
timeout 70 python tools/host_call_probe.py > gpurun_out/probe_new.json 2> gpurun_out/probe_new.err; echo "probe_new rc=$?" >> gpurun_out/rc.log
GSLAM_HIP_LIB=build/ab/libgslam_hip_old.so timeout 70 python tools/host_call_probe.py > gpurun_out/probe_old.json 2> gpurun_out/probe_old.err; echo "probe_old rc=$?" >> gpurun_out/rc.log
GSLAM_HIP_PG_ARENA=0 timeout 70 python tools/host_call_probe.py > gpurun_out/probe_noarena.json 2> gpurun_out/probe_noarena.err; echo "probe_noarena rc=$?" >> gpurun_out/rc.log
IG=""; for f in $F1; do IG="$IG --ignore=$f"; done
timeout 300 python -m pytest tests -m gpu -q --tb=short $IG > gpurun_out/t2.log 2>&1; echo "t2 rc=$?" >> gpurun_out/rc.log
cat gpurun_out/rc.log; tail -3 gpurun_out/t1.log; tail -3 gpurun_out/t2.log; cat gpurun_out/probe_new.json gpurun_out/probe_old.json gpurun_out/probe_noarena.json
