# SQ_INSTS_VALU per wave of orb_fast_cells / orb_describe, the built library against build/ab/libgslam_hip_prev.so (gpurun: bash tools/valu_ab.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for v in new prev; do
  lib=$R/gslam_amd/lib/libgslam_hip.so; [ $v = prev ] && lib=$R/build/ab/libgslam_hip_prev.so
  rm -rf $O/prof_ab_$v
  GSLAM_HIP_LIB=$lib timeout 100 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/prof_ab_$v -- python $R/tools/orb_perf.py 100 > /dev/null 2>&1
done
python - <<PY
import csv,glob
for v in ("new","prev"):
    acc={}
    for path in glob.glob("$O/prof_ab_%s/**/*_counter_collection.csv"%v, recursive=True):
        for row in csv.DictReader(open(path)):
            n=row["Kernel_Name"]
            k="fast_cells" if "fast_cells_kernel" in n else ("describe" if "describe_kernel" in n else None)
            if k is None: continue
            a=acc.setdefault(k,{"SQ_INSTS_VALU":0.0,"SQ_WAVES":0.0})
            if row["Counter_Name"] in a: a[row["Counter_Name"]]+=float(row["Counter_Value"])
    for k,a in acc.items():
        if a["SQ_WAVES"]: print(v,k,"VALU/wave %.1f"%(a["SQ_INSTS_VALU"]/a["SQ_WAVES"]))
PY
rm -rf $O/prof_ab_new $O/prof_ab_prev
