# round-4: occupancy sensitivity of orb_fast_cells (extra dynamic LDS per workgroup -> 6 / 5 / 4 workgroups per CU) and the SQ counter picture
mkdir -p gpurun_out
for v in 0 1; do for pad in 0 7000 14000; do
  GSLAM_HIP_ORB_PASS1=$v GSLAM_HIP_ORB_LDSPAD=$pad timeout 60 python tools/orb_perf.py 400 > gpurun_out/occ_${v}_$pad.txt 2>&1
  echo "pass1=$v pad=$pad: $(grep -h 'orb_fast_cells' gpurun_out/occ_${v}_$pad.txt | tail -1) | $(grep -h 'extract' gpurun_out/occ_${v}_$pad.txt | tail -1)"
done; done
bash tools/orb_counters.sh "0 1" > gpurun_out/orb_counters.txt 2>&1
cat gpurun_out/orb_counters.txt
