"""Build check (no GPU): register / scratch budgets of the hot kernels in the shipped libgslam_hip.so, read from the code
objects' metadata.  The kernels are tuned against these numbers (occupancy of the VALU-bound ORB
kernels, 256-register budget of the single-launch factorisation, no spills to scratch anywhere on the hot path); a
compiler or flag change that breaks one of them should fail here, on the CPU, before it shows up as a slower bench."""
import os
import re
import subprocess

import pytest

from test_build_float_mode import LIB, LLVM, _code_objects

# kernel-name fragment -> (max VGPRs + AGPRs, max bytes of scratch per work-item)
BUDGET = {
    "fast_cells_kernel": (72, 0),        # packed-16-bit variant: >= 6 waves per SIMD beside 24.5 KB of LDS per workgroup
    "fast_cells_all_kernel": (72, 0),
    "describe_kernelILi13ELb0ELb1E": (56, 0),   # the 30-bin table mode (default), h-pass of the blur on MFMA: 9 waves per SIMD, 9.5 KB of LDS
    "describe_kernelILi13ELb0ELb0E": (48, 0),   # ... with the blur on the VALU (GSLAM_HIP_ORB_DESC_MFMA=0): 8 workgroups per CU with 20.1 KB of LDS
    "describe_kernelILi19ELb1ELb0E": (96, 0),   # continuous steering (optional mode: 45 x 45 patch, 38 KB of LDS per workgroup)
    "describe_pipe_kernel": (96, 0),            # the shipped default (software pipeline over 8 keypoints per wave): 5 waves per SIMD, no scratch
    "select_kernel": (128, 0),
    "resize_kernel": (64, 0),
    "bf_match_pairs_kernel": (64, 0),    # 8 waves per SIMD
    "bf_match_split_kernel": (64, 0),
    "bf_match_pairs_mfma_kernel": (256, 0),
    "potrf_flow_kernel": (256, 0),       # 512-thread workgroups: two waves per SIMD, everything in registers
    "bwd_chain_kernel": (128, 0),
    "syrk_mfma_kernel": (256, 0),
    # eight waves per 128 x 128 tile, four waves per SIMD: 128 registers; the potf2 + inverse of the next diagonal block that
    # ONE workgroup of the launch runs (potf2_fused, written for 256 registers) spills a few values there -- the MFMA loops of
    # the tiles hold everything in registers (no scratch instruction between the loop labels of the ISA)
    "syrk_mfma8_kernel": (128, 64),
    "lin_kernel": (192, 0),
    "schur_blocks_kernel": (256, 0),
    # band solver (chol_cr.hip): the factor kernel is a 512-thread workgroup (256 registers per wave: its tiles are loaded in
    # two stages for exactly that), the update kernel must keep two workgroups per CU, nothing may touch scratch
    "cr_factor_kernel": (256, 0),
    "cr_update_kernel": (256, 0),
    "cr_back_kernel": (96, 0),
    "marginalize_pairs_kernel": (512, 0),   # (not hot: one wave per camera pair, 36 + 60 doubles live; no scratch)
    "slam_cells_wave_kernel": (80, 0),      # quadtree mode: one wave per cell, six workgroups (24 waves) per CU beside 23.9 KB of LDS
    "gr_schur_cam_kernel": (256, 0),        # self-calibration's Schur slot (one thread per landmark, 27 + 27 doubles live)
    "bow_words_kernel": (64, 0),
    "pack_results_kernel": (32, 0),
}


def _kernels():
    rows = {}
    for i, co in enumerate(_code_objects(LIB)):
        p = "/tmp/gslam_build_res_%d_%d.elf" % (os.getpid(), i)
        with open(p, "wb") as f:
            f.write(co)
        try:
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", p], capture_output=True, text=True, check=True).stdout
        finally:
            os.remove(p)
        cur = {}
        for line in notes.splitlines():
            m = re.match(r"\s+(?:- )?\.(\w+):\s+(\S+)\s*$", line)
            if not m:
                continue
            k, v = m.groups()
            if k == "agpr_count" and "name" in cur:  # first key of the next kernel's record
                rows[cur["name"]] = cur
                cur = {}
            if k in ("name", "vgpr_count", "agpr_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"):
                cur[k] = v
        if "name" in cur:
            rows[cur["name"]] = cur
    return rows


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-readelf")), reason="needs the ROCm llvm-readelf")
def test_hot_kernels_stay_within_their_register_and_scratch_budgets():
    assert os.path.exists(LIB), "build the library first (make lib)"
    rows = _kernels()
    assert len(rows) >= 60
    for frag, (max_regs, max_scratch) in BUDGET.items():
        # the mangled name holds <length><identifier>; a fragment may carry the first template arguments (I...E)
        hits = [r for n, r in rows.items() if re.search(r"\d+%s([A-Z]|$)" % frag, n) or (frag.endswith("E") and frag in n)]
        assert hits, "kernel %s not found in the library" % frag
        for r in hits:
            regs = int(r["vgpr_count"]) + int(r.get("agpr_count", 0))
            assert regs <= max_regs, "%s: %d registers (budget %d)" % (r["name"], regs, max_regs)
            assert int(r["vgpr_spill_count"]) == 0 or max_scratch > 0, "%s spills vector registers" % r["name"]
            assert int(r["private_segment_fixed_size"]) <= max_scratch, "%s: %s bytes of scratch" % (r["name"], r["private_segment_fixed_size"])


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-readelf")), reason="needs the ROCm llvm-readelf")
def test_swar_fast_cells_fits_eight_workgroups_per_cu():
    """The shipped orb_fast_cells (SWAR pass 1, fast_cells_kernel<true, 1>) is latency sensitive: 6 -> 7 -> 8 workgroups per
    CU measured 5.26 -> 4.82 -> 4.33 ms per 8 launches (profiles/orb_pass1_ab_r04.txt).  Eight 256-thread workgroups need
    at most 64 VGPRs (8 waves per SIMD) and 160 KB / 8 = 20480 bytes of LDS each."""
    rows = _kernels()
    # two variants: <.., PLANE = false> is the default mode's kernel, <.., PLANE = true> writes the score plane of the quadtree mode
    hits = [r for n, r in rows.items() if "fast_cells_kernelILb1ELi1E" in n]
    assert len(hits) == 2, [n for n in rows if "fast_cells" in n]
    for r in hits:
        assert int(r["vgpr_count"]) + int(r.get("agpr_count", 0)) <= 64, r
        assert int(r["group_segment_fixed_size"]) <= 20480, r
        assert int(r["vgpr_spill_count"]) == 0 and int(r["private_segment_fixed_size"]) == 0, r


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-readelf")), reason="needs the ROCm llvm-readelf")
def test_quadtree_mode_kernels_keep_their_occupancy():
    """Round 5b: the continuous-steering orb_describe with its blur on MFMA (describe_kernel<19, true, true>) holds 8 workgroups
    per CU (18.7 KB of LDS instead of 38 KB, at most 64 VGPRs) and does not spill; the score-plane cell kernel keeps 7 (cells up
    to 32 x 32) / 5 (up to 40 x 40) workgroups per CU (profiles/orb_slam_mode_r05.txt)."""
    rows = _kernels()
    d19 = [r for n, r in rows.items() if "describe_kernelILi19ELb1ELb1E" in n]
    assert len(d19) == 1, [n for n in rows if "describe_kernel" in n]
    assert int(d19[0]["vgpr_count"]) <= 64 and int(d19[0]["group_segment_fixed_size"]) <= 20480, d19[0]
    assert int(d19[0]["private_segment_fixed_size"]) == 0, d19[0]
    pc = {n: r for n, r in rows.items() if "slam_cells_plane_kernel" in n}
    assert len(pc) == 2, list(pc)
    for n, r in pc.items():
        big = "ILb1E" in n
        assert int(r["group_segment_fixed_size"]) <= 160 * 1024 // (5 if big else 7), (n, r)
        assert int(r["private_segment_fixed_size"]) == 0, (n, r)
