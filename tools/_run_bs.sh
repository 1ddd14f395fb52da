cd /root/repo
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_bsparse.py tests/test_pg_gpu.py -q -x -m gpu -s -k "bsparse or numeric or block_sparse or 6000" --durations=8 2>&1 | tail -25
timeout 400 python tools/pg_large_probe.py 700 1500 5000 20000 2>&1 | tee gpurun_out/pg_block_sparse.txt | cut -c1-700
