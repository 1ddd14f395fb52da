cd /root/repo
mkdir -p gpurun_out
LIGHT="--no-cpu-baseline --no-ba --no-bow --no-c3 --no-c5 --no-host-fed --no-range --no-all-pairs-full --steps 40"
for ov in 0 1 0 1; do
  GSLAM_HIP_ORB_SELECT_OVERLAP=$ov timeout 200 python bench.py $LIGHT 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['extra']['kernels']
print('overlap $ov', d['value'], d['ms_per_step'], {n:round(v.get('ms_per_step',v.get('total_ms',0)),3) if isinstance(v,dict) else v for n,v in k.items()})
"
done
GSLAM_HIP_ORB_SELECT_OVERLAP=1 timeout 300 python -m pytest tests/test_orb_gpu.py -q -x -m gpu 2>&1 | tail -3
timeout 300 python -m pytest tests/test_bsparse.py tests/test_pg_gpu.py -q -x -m gpu -k "bsparse or numeric or agree or block_sparse or 6000" 2>&1 | tail -3
