set -x
cd /root/repo
timeout 600 python -m pytest tests/test_bsparse.py tests/test_pg_gpu.py -q -x -m gpu -s 2>&1 | tail -30
