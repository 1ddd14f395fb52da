cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cr_trace -- $GRAFT_REPO_ROOT/build/cr_stamp_probe > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/cr_trace/*/*kernel_trace.csv | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last solve: take last 25 kernels
rows=rows[-24:]
t0=int(rows[0]['Start_Timestamp'])
prev_end=None
for r in rows:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    name=r['Kernel_Name'][:40]
    gap = (s-prev_end)/1000 if prev_end else 0
    print(f"{name:42s} start {(s-t0)/1000:8.1f} dur {(e-s)/1000:7.1f} gap {gap:6.1f} grid {r.get('Grid_Size_X','?')} wg {r.get('Workgroup_Size_X','?')}")
    prev_end=e
PY
