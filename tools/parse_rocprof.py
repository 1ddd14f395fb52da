#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (gpurun_out/prof_*) into the tracked summaries under profiles/.

  python tools/parse_rocprof.py <round-tag> <frames-per-launch> ["<profiled command>"]
    gpurun_out/prof_stats/**/_kernel_stats.csv      -> profiles/<tag>_kernel_stats.csv   (rocprofv3 --kernel-trace --stats)
    gpurun_out/prof_fetch/**/_counter_collection.csv + prof_write/** -> profiles/pmc_traffic.json
    gpurun_out/prof_sq/**/_counter_collection.csv                    -> profiles/sq_counters.json

HBM traffic per launch follows MI355X_MICROARCH.md "HBM": separate --pmc passes for FETCH_SIZE and WRITE_SIZE
(they do not fit one pass), both in KiB units; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide
(16 B/lane) coalesced read, so hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  WRITE_SIZE is uncalibrated
per the guide and is reported as is.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {"fast_cells_kernel": "orb_fast_cells", "resize_kernel": "orb_resize", "select_kernel": "orb_select",
         "describe_kernel": "orb_describe", "describe_pipe_kernel": "orb_describe", "bf_match_pairs_kernel": "bf_match_pairs", "bf_match_pairs_mfma_kernel": "bf_match_pairs_mfma", "synth_kernel": "synth_frames",
         "syrk_mfma_kernel": "ba_syrk", "panel_step_kernel": "ba_panel_step", "potf2_inv_kernel": "ba_potf2", "trsm_inv_kernel": "ba_trsm",
         "schur_blocks_kernel": "ba_schur_blocks", "schur_reduce_kernel": "ba_schur_reduce",
         "lin_kernel": "ba_lin", "lin_cams_reduce_kernel": "ba_lin_cams_reduce",
         "fwd_step_kernel": "ba_trsv_fwd", "bwd_step_inv_kernel": "ba_trsv_bwd",
         "potrf_flow_kernel": "ba_potrf_flow", "bwd_chain_kernel": "ba_trsv_bwd (single launch)",
         "cr_factor_kernel": "ba_cr_factor", "cr_panels_kernel": "ba_cr_panels (+ ba_cr_inverse)",
         "cr_update_kernel": "ba_cr_update (+ ba_cr_backprep)", "cr_back_kernel": "ba_cr_back",
         "schur_blocks_init_kernel": "ba_schur_blocks (+ seed of S)",
         "cr_border_update_kernel": "ba_cr_border_update", "cr_border_syrk_kernel": "ba_cr_border_syrk",
         "cr_border_syrk_reduce_kernel": "ba_cr_border_syrk_reduce", "cr_border_gather_kernel": "ba_cr_border_gather",
         "cr_border_back_kernel": "ba_cr_border_back", "cr_border_yh_kernel": "ba_cr_border_yh"}


def short(name):
    for k, v in NAMES.items():
        if k in name:
            return v
    return None


def newest(paths):
    """gpurun merges every call's files into gpurun_out/: only the latest collection counts."""
    return sorted(paths, key=os.path.getmtime)[-1:]


# Kernels launched on ONE shape per step: dispatches of any other grid (e.g. the all-pairs stress launches of the matcher,
# 8x the work of a step launch) are left out instead of being averaged in.  orb_fast_cells is launched once per pyramid
# level (8 shapes per step): its per-launch figure is the average over all of them, like its duration.
SINGLE_SHAPE = {"bf_match_pairs", "bf_match_pairs_mfma", "orb_select", "orb_describe"}


def counters(pattern, counter):
    rows_by = defaultdict(list)
    for path in newest(glob.glob(os.path.join(ROOT, "gpurun_out", pattern, "**", "*_counter_collection.csv"), recursive=True)):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] != counter:
                continue
            s = short(row["Kernel_Name"])
            if s:
                rows_by[s].append((row.get("Grid_Size", "?"), float(row["Counter_Value"])))
    out = {}
    for k, rows in rows_by.items():
        grid = None
        if k in SINGLE_SHAPE:
            cnt = defaultdict(int)
            for g, _ in rows:
                cnt[g] += 1
            grid = max(cnt, key=lambda g: cnt[g])
            rows = [r for r in rows if r[0] == grid]
        out[k] = (sum(v for _, v in rows) / len(rows), len(rows), grid)
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    stats = newest(glob.glob(os.path.join(ROOT, "gpurun_out", "prof_stats", "**", "*_kernel_stats.csv"), recursive=True))
    if stats:
        rows = list(csv.reader(open(stats[0])))
        with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
            cmd = sys.argv[3] if len(sys.argv) > 3 else f"python bench.py --frames {frames} --no-cpu-baseline"
            f.write(f"# rocprofv3 --kernel-trace --stats --output-format csv -- {cmd}   (1x MI355X)\n")
            csv.writer(f).writerows(rows)
        print("wrote", f"profiles/{tag}_kernel_stats.csv")
    fetch = counters("prof_fetch", "FETCH_SIZE")
    write = counters("prof_write", "WRITE_SIZE")
    traffic = {}
    for k in sorted(set(fetch) | set(write)):
        f_kib = fetch.get(k, (0.0, 0, None))[0]
        w_kib = write.get(k, (0.0, 0, None))[0]
        traffic[k] = {"frames_per_launch": frames, "launches_sampled": fetch.get(k, (0, 0, None))[1],
                      "grid_size": fetch.get(k, (0, 0, None))[2],
                      "FETCH_SIZE_KiB_avg": round(f_kib, 1), "WRITE_SIZE_KiB_avg": round(w_kib, 1),
                      "hbm_bytes_per_launch": int((2.0 * f_kib + w_kib) * 1024),
                      "correction": "2 x FETCH_SIZE (gfx950 wide-read under-count) + WRITE_SIZE, KiB -> bytes"}
    # instruction mix per launch (rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES)
    sq = {}
    for cname in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVES"):
        for k, (avg, n, grid) in counters("prof_sq", cname).items():
            sq.setdefault(k, {"frames_per_launch": frames, "launches_sampled": n, "grid_size": grid})[cname + "_per_launch"] = round(avg, 1)
    if sq:
        json.dump(sq, open(os.path.join(out_dir, "sq_counters.json"), "w"), indent=1)
        print("wrote profiles/sq_counters.json")
        for k, v in sq.items():
            w = v.get("SQ_WAVES_per_launch", 0) or 1
            print("  %-18s VALU/wave %.0f  SALU/wave %.0f  LDS/wave %.0f" %
                  (k, v.get("SQ_INSTS_VALU_per_launch", 0) / w, v.get("SQ_INSTS_SALU_per_launch", 0) / w,
                   v.get("SQ_INSTS_LDS_per_launch", 0) / w))
    if traffic:
        json.dump(traffic, open(os.path.join(out_dir, "pmc_traffic.json"), "w"), indent=1)
        print("wrote profiles/pmc_traffic.json")
        for k, v in traffic.items():
            print("  %-18s fetch %.0f KiB  write %.0f KiB  -> %.1f MB/launch" %
                  (k, v["FETCH_SIZE_KiB_avg"], v["WRITE_SIZE_KiB_avg"], v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
