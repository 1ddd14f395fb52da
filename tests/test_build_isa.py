"""Build check (no GPU): quarter-rate integer instructions inside the loops of the hot ORB kernels, read from the shipped gfx950
code objects.  v_mul_lo_u32, v_mul_hi_*, v_mad_u64_u32 / v_mad_i64_i32 and the v_rcp_iflag_f32 of an integer division issue at a
quarter of the rate of the 24-bit multiplier; compilers produce them from innocent source (a division by 7, a size_t row
offset, a constant times 0x10001).  Taking them out of orb_describe and orb_fast_cells was worth 25 % / 4 % of their VALU
instructions: this test keeps them out."""
import os
import re
import subprocess

import pytest

from test_build_float_mode import LIB, LLVM, _code_objects

SLOW = re.compile(r"v_(mul_lo_u32|mul_hi_u32|mul_hi_i32|mad_u64_u32|mad_i64_i32|rcp_iflag_f32)")

# kernel (mangled-name fragment) -> most slow integer instructions allowed inside loops
BUDGET = {
    "fast_cells_kernelILb1ELi1E": 0,     # the shipped FAST / NMS / pyramid kernel
    "describe_kernelILi13ELb0E": 0,      # table-mode descriptors
    "resize_kernel": 0,
    "slam_cells_wave_kernel": 0,         # quadtree mode
}


def _loops_and_lines(body):
    lines = [l.strip() for l in body.splitlines() if l.strip()]
    addr = {}
    for n, l in enumerate(lines):
        m = re.search(r"// ([0-9A-F]+):", l)
        if m:
            addr[int(m.group(1), 16)] = n
    base = min(addr)
    loops = []
    for n, l in enumerate(lines):
        if l.startswith("s_cbranch") or l.startswith("s_branch"):
            m, cur = re.search(r"\+0x([0-9a-f]+)>", l), re.search(r"// ([0-9A-F]+):", l)
            if m and cur:
                tgt = base + int(m.group(1), 16)
                if tgt < int(cur.group(1), 16) and tgt in addr:
                    loops.append((addr[tgt], n))
    return lines, loops


def _scan():
    found = {}
    for i, co in enumerate(_code_objects(LIB)):
        path = "/tmp/gslam_isa_%d_%d.elf" % (os.getpid(), i)
        with open(path, "wb") as f:
            f.write(co)
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout
        os.remove(path)
        for m in re.finditer(r"<(_Z\S+)>:\n(.*?)\n\n", dis, re.S):
            for frag in BUDGET:
                if frag in m.group(1):
                    lines, loops = _loops_and_lines(m.group(2))
                    found[frag] = [l.split("//")[0].strip() for n, l in enumerate(lines)
                                   if SLOW.match(l) and any(a <= n <= b for a, b in loops)]
    return found


def test_no_quarter_rate_integer_arithmetic_in_the_loops_of_the_orb_kernels():
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    found = _scan()
    for frag, budget in BUDGET.items():
        assert frag in found, "kernel %s not found in the shipped code objects" % frag
        assert len(found[frag]) <= budget, (frag, found[frag])


def test_sliding_window_prefetch_registers_are_not_copied_between_load_and_wait():
    """fast_plane_sw_kernel (round-6 experiment) loads its prefetched rows by inline asm and waits for them by hand
    (`s_waitcnt vmcnt(4)`): the compiler believes the destination registers are valid at once, so a register copy (or a spill)
    between the load and the wait would read them before the data lands.  In the shipped code object every `global_load_dword vD, vD,
    s[..]` (destination == address register: the inline-asm form) must be followed by no VALU / DS instruction that READS vD before
    the next hand-placed `s_waitcnt vmcnt(4)`."""
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    body = None
    for i, co in enumerate(_code_objects(LIB)):
        path = "/tmp/gslam_isa_sw_%d_%d.elf" % (os.getpid(), i)
        with open(path, "wb") as f:
            f.write(co)
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout
        os.remove(path)
        m = re.search(r"<(_Z\S*fast_plane_sw_kernel\S*)>:\n(.*?)\n\n", dis, re.S)
        if m:
            body = m.group(2)
    assert body is not None, "fast_plane_sw_kernel not found in the shipped code objects"
    lines = [l.split("//")[0].strip() for l in body.splitlines() if l.strip()]
    pending = {}  # register -> line of the asm load that owns it
    checked = 0
    for n, l in enumerate(lines):
        m = re.match(r"global_load_dword (v\d+), (v\d+), s\[", l)
        if m and m.group(1) == m.group(2):
            pending[m.group(1)] = n
            continue
        if l.startswith("s_waitcnt vmcnt(4)") or l.startswith("s_waitcnt vmcnt(0)"):
            checked += len(pending)
            pending.clear()
            continue
        if l.startswith("s_") or not pending:
            continue
        ops = l.split(None, 1)[1] if " " in l else ""
        srcs = ops.split(",")[1:] if not l.startswith(("ds_write", "global_store", "v_cmp")) else ops.split(",")
        for reg in pending:
            for src in srcs:
                assert not re.search(r"\b%s\b" % reg, src), "line %d reads %s before its load (line %d) was waited for: %s" % (n, reg, pending[reg], l)
            m2 = re.search(r"v\[(\d+):(\d+)\]", ",".join(srcs))
            if m2:
                assert not (int(m2.group(1)) <= int(reg[1:]) <= int(m2.group(2))), (n, l)
    assert checked >= 8, "expected the prologue's and the loop's prefetch sets (got %d registers)" % checked
