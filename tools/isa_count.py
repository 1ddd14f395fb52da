#!/usr/bin/env python3
"""Static instruction census of kernels in libgslam_hip.so (no GPU): per kernel whose mangled name contains a fragment, the
number of VALU / MFMA / LDS / vector-memory / scalar instructions in its code, and optionally the disassembly.
usage: tools/isa_count.py <name-fragment> [--dump DIR] [--lib PATH]
(static counts: loops count once -- use them to compare two formulations of straight-line code, not as trip-weighted totals)"""
import os
import re
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_build_float_mode import LIB, LLVM, _code_objects  # noqa: E402


def kernels(lib, frag):
    out = {}
    for i, co in enumerate(_code_objects(lib)):
        if frag.encode() not in co:
            continue
        p = "/tmp/isa_count_%d_%d.elf" % (os.getpid(), i)
        open(p, "wb").write(co)
        try:
            txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", p], capture_output=True, text=True, check=True).stdout
        finally:
            os.remove(p)
        for b in re.split(r"\n(?=[0-9a-f]+ <)", txt):
            m = re.match(r"[0-9a-f]+ <([^>]+)>:", b)
            if m and frag in m.group(1):
                out[m.group(1)] = b
    return out


def census(body):
    ins = [l.split()[0] for l in body.split("\n")[1:] if re.match(r"\s+[a-z]", l)]
    c = Counter(ins)
    cls = lambda pred: sum(v for k, v in c.items() if pred(k))
    return {
        "total": len(ins),
        "valu": cls(lambda k: k.startswith("v_") and not k.startswith(("v_mfma", "v_accvgpr"))),
        "mfma": cls(lambda k: k.startswith("v_mfma")),
        "lds": cls(lambda k: k.startswith("ds_")),
        "vmem": cls(lambda k: k.startswith(("global_", "buffer_", "flat_", "scratch_"))),
        "salu": cls(lambda k: k.startswith("s_")),
    }, c


if __name__ == "__main__":
    frag = sys.argv[1]
    lib = LIB
    dump = None
    if "--lib" in sys.argv:
        lib = sys.argv[sys.argv.index("--lib") + 1]
    if "--dump" in sys.argv:
        dump = sys.argv[sys.argv.index("--dump") + 1]
        os.makedirs(dump, exist_ok=True)
    for name, body in sorted(kernels(lib, frag).items()):
        cen, c = census(body)
        print(name, cen)
        if "--top" in sys.argv:
            print("   ", c.most_common(25))
        if dump:
            open(os.path.join(dump, re.sub(r"\W", "_", name)[:120] + ".s"), "w").write(body)
