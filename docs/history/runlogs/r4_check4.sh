# pyramid placement A/B: fused into fast_cells (1), stand-alone serial (0), stand-alone AHEAD on its own stream (2); parity under 2
mkdir -p gpurun_out
for f in 1 0 2 1 0 2; do GSLAM_HIP_ORB_FUSE_PYRAMID=$f python tools/orb_perf.py 400 2>&1 | grep -h "extract\|orb_" | tail -5 | tr '\n' '|'; echo " fuse=$f"; done
T="tests/test_orb_gpu.py tests/test_orb_adversarial_gpu.py tests/test_orb_stream_gpu.py"
GSLAM_HIP_ORB_FUSE_PYRAMID=2 timeout 300 python -m pytest $T -m gpu -q --tb=short -x 2>&1 | tail -4
timeout 200 python -m pytest tests/test_orb_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT}
GSLAM_HIP_ORB_FUSE_PYRAMID=0 timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/prof_f0 -- python $R/tools/orb_perf.py 400 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for path in glob.glob("$R/gpurun_out/prof_f0/**/*_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        for key in ("fast_cells_kernel", "resize_kernel"):
            if key in row["Kernel_Name"]:
                acc[key][row["Counter_Name"]] += float(row["Counter_Value"])
for k, d in acc.items():
    print("fuse=0", k, "VALU per wave %.1f" % (d["SQ_INSTS_VALU"] / d["SQ_WAVES"]), "waves", d["SQ_WAVES"], "valu", d["SQ_INSTS_VALU"])
PY
rm -rf $R/gpurun_out/prof_f0
