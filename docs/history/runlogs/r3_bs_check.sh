mkdir -p gpurun_out
timeout 60 python -m pytest tests/test_bsparse.py tests/test_pg_gpu.py -m gpu -q --tb=short > gpurun_out/bs_t.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/bs_t.log
timeout 40 python tools/host_call_probe.py > gpurun_out/probe_bs.json 2> gpurun_out/probe_bs.err; echo "probe rc=$?"; cat gpurun_out/probe_bs.json
