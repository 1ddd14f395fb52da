#!/usr/bin/env python3
"""Static resource survey of every kernel in gslam_amd/lib/libgslam_hip.so, read from the bundled gfx950 code objects (no GPU
needed): VGPRs / AGPRs / SGPRs, spills, scratch bytes per work-item, static LDS, maximum workgroup size, fp16 denormal mode.

  python tools/kernel_resources.py > profiles/kernel_resources_r03.txt
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_build_float_mode import LIB, LLVM, _code_objects  # noqa: E402

KEYS = ("name", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
        "group_segment_fixed_size", "max_flat_workgroup_size")


def main():
    rows = {}
    for co in _code_objects(LIB):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(co)
            f.flush()
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True).stdout
            secs = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-S", "-W", f.name], capture_output=True, text=True).stdout
            syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "-W", f.name], capture_output=True, text=True).stdout
        m = re.search(r"\.rodata\s+PROGBITS\s+([0-9a-f]+)\s+([0-9a-f]+)", secs)
        denorm = {}
        if m:
            addr, off = int(m.group(1), 16), int(m.group(2), 16)
            for line in syms.splitlines():
                p = line.split()
                if len(p) >= 8 and p[-1].endswith(".kd"):
                    (rsrc1,) = struct.unpack_from("<I", co, off + int(p[1], 16) - addr + 48)
                    denorm[p[-1][:-3]] = (rsrc1 >> 18) & 3
        cur = {}
        for line in notes.splitlines():
            mm = re.match(r"\s+(?:- )?\.(\w+):\s+(\S+)\s*$", line)
            if not mm:
                continue
            k, v = mm.groups()
            if k == "agpr_count" and "name" in cur:
                rows[cur["name"]] = cur
                cur = {}
            if k in KEYS:
                cur[k] = v
        if "name" in cur:
            rows[cur["name"]] = cur
        for n, d in denorm.items():
            if n in rows:
                rows[n]["denorm16"] = d
    print("# python tools/kernel_resources.py -- %d kernels in gslam_amd/lib/libgslam_hip.so (gfx950 code objects)" % len(rows))
    print("%-52s %5s %5s %5s %7s %7s %8s %7s %6s %8s" % ("kernel", "vgpr", "agpr", "sgpr", "v-spill", "s-spill", "scratch", "lds", "wg", "denorm16"))
    out = []
    for n, r in rows.items():
        dem = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0].replace("void ", "")
        if "rocprim" in dem:
            dem = "rocprim::" + dem.split("::")[-1]
        out.append("%-52s %5s %5s %5s %7s %7s %8s %7s %6s %8s" % (dem[:52], r.get("vgpr_count"), r.get("agpr_count"), r.get("sgpr_count"),
                                                               r.get("vgpr_spill_count"), r.get("sgpr_spill_count"),
                                                               r.get("private_segment_fixed_size"), r.get("group_segment_fixed_size"),
                                                               r.get("max_flat_workgroup_size"), r.get("denorm16", "?")))
    for line in sorted(set(out)):
        print(line)


if __name__ == "__main__":
    main()
