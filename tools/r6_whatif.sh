#!/bin/bash
# orb_fast_cells stage by stage: build/ab/libgslam_hip_w<mask>.so are the library with -DGH_ORB_WHATIF=<mask> (orb.hip: bit 0 no pass 2,
# bit 1 no pass 1 (hence no pass 2 either), bit 2 no cell stage; wrong results, timing only).  gpurun -- bash tools/r6_whatif.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
MASKS=${MASKS:-"0 1 2 4 5 6 7"}
for rep in 1 2; do
for m in $MASKS; do
  lib=$R/gslam_amd/lib/libgslam_hip.so; [ $m != 0 ] && lib=$R/build/ab/libgslam_hip_w$m.so
  echo "mask $m: $(GSLAM_HIP_LIB=$lib timeout 120 python $R/tools/orb_perf.py 400 2>&1 | grep -E 'orb_fast_cells|orb_describe|orb_select' | tr -s ' ' | tr '\n' '|')"
done
done
for m in $MASKS; do
  lib=$R/gslam_amd/lib/libgslam_hip.so; [ $m != 0 ] && lib=$R/build/ab/libgslam_hip_w$m.so
  rm -rf $O/prof_wi_$m
  GSLAM_HIP_LIB=$lib timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $O/prof_wi_$m -- python $R/tools/orb_perf.py 100 > /dev/null 2>&1
  python - <<PY
import csv,glob
acc={}
for path in glob.glob("$O/prof_wi_$m/**/*_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "fast_cells_kernel" not in row["Kernel_Name"]: continue
        acc[row["Counter_Name"]]=acc.get(row["Counter_Name"],0.0)+float(row["Counter_Value"])
w=acc.get("SQ_WAVES",0)
if w: print("mask $m per wave: VALU %.1f SALU %.1f LDS %.1f"%(acc["SQ_INSTS_VALU"]/w,acc["SQ_INSTS_SALU"]/w,acc["SQ_INSTS_LDS"]/w))
PY
  rm -rf $O/prof_wi_$m
done
