"""Build check (no GPU): the kernels of orb.hip must be compiled with fp16 denormals PRESERVED -- orb_fast_cells compares bytes
held as fp16 denormals with v_pk_minimum3_f16 / v_pk_maximum3_f16 (gslam_amd/csrc/orb.hip: fast_score16_pk), and a build with
-fgpu-flush-denormals-to-zero / -ffast-math would silently zero every arc score.  Reads FLOAT_DENORM_MODE_16_64
(compute_pgm_rsrc1 bits 18-19 of the kernel descriptor) of every *fast_cells* kernel in the shipped libgslam_hip.so."""
import os
import re
import shutil
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gslam_amd", "lib", "libgslam_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(path):
    """every gfx950 code object bundled into the library (.hip_fatbin holds one bundle per translation unit)"""
    data = open(path, "rb").read()
    out = []
    for m in re.finditer(re.escape(MAGIC), data):
        base = m.start()
        (n,) = struct.unpack_from("<Q", data, base + len(MAGIC))
        p = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                out.append(data[base + off:base + off + size])
    return out


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-readelf")), reason="needs the ROCm llvm-readelf")
def test_fast_cells_kernels_keep_fp16_denormals(tmp_path):
    assert os.path.exists(LIB), "build the library first (make lib)"
    seen = 0
    for i, co in enumerate(_code_objects(LIB)):
        if b"fast_cells" not in co:
            continue
        f = tmp_path / f"co{i}.elf"
        f.write_bytes(co)
        secs = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-S", "-W", str(f)], capture_output=True, text=True, check=True).stdout
        syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "-W", str(f)], capture_output=True, text=True, check=True).stdout
        m = re.search(r"\.rodata\s+PROGBITS\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", secs)
        assert m, "code object without .rodata"
        addr, off = int(m.group(1), 16), int(m.group(2), 16)
        for line in syms.splitlines():
            parts = line.split()
            if len(parts) >= 8 and parts[-1].endswith(".kd") and "fast_cells" in parts[-1]:
                kd = co[off + int(parts[1], 16) - addr:][:64]
                (rsrc1,) = struct.unpack_from("<I", kd, 48)
                assert (rsrc1 >> 18) & 3 == 3, f"{parts[-1]}: FLOAT_DENORM_MODE_16_64 = {(rsrc1 >> 18) & 3}, fp16 denormals are flushed"
                seen += 1
    assert seen >= 4, f"expected the fast_cells kernel descriptors (both score paths, both launch shapes), found {seen}"
