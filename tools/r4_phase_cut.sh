#!/bin/bash
# Per-phase cost of orb_fast_cells: four extra builds of the library in which the tile function returns after the tile load /
# pass 1 + queue / pass 2 / the fused resize (-DGH_FAST_CUT=1..4), timed with tools/orb_perf.py and counted with SQ_INSTS_VALU.
# Build here (authoring container): bash tools/r4_phase_cut.sh build ; on the GPU box: bash tools/r4_phase_cut.sh run
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
if [ "$1" = build ]; then
  mkdir -p build/ab
  for c in 1 2 3 4; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -Iinclude -DGH_FAST_CUT=$c -c gslam_amd/csrc/orb.hip -o build/ab/orb_cut$c.o 2>/dev/null
    objs=$(ls build/obj/*.o | grep -v "/orb.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/libgslam_hip_cut$c.so $objs build/ab/orb_cut$c.o -lpthread -ldl -lrt
  done
  ls -la build/ab/*.so
  exit 0
fi
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in 1 2 3 4 0; do
  lib=$R/build/ab/libgslam_hip_cut$c.so; [ $c = 0 ] && lib=$R/gslam_amd/lib/libgslam_hip.so
  GSLAM_HIP_LIB=$lib timeout 60 python $R/tools/orb_perf.py 400 > $O/cut_$c.txt 2>&1 || true
  GSLAM_HIP_LIB=$lib timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/prof_cut_$c -- python $R/tools/orb_perf.py 400 > $O/prof_cut_$c.log 2>&1 || true
  echo "cut $c: $(grep -h 'orb_fast_cells' $O/cut_$c.txt | tail -1)"
done
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob("$O/prof_cut_*/**/*_counter_collection.csv", recursive=True):
    v = re.search(r"prof_cut_(\d+)", path).group(1)
    for row in csv.DictReader(open(path)):
        if "fast_cells_kernel" in row["Kernel_Name"]:
            a = acc[v][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
for v, d in sorted(acc.items()):
    w = d["SQ_WAVES"][0] / d["SQ_WAVES"][1]
    print("cut %s: per wave" % v, {c: round(x / n / w, 1) for c, (x, n) in sorted(d.items()) if c != "SQ_WAVES"})
PY
find $O/prof_cut_* -type f ! -name "*.csv" -delete 2>/dev/null
find $O/prof_cut_* -name "*kernel_trace.csv" -delete 2>/dev/null
