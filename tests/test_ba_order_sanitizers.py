"""The host code of the camera ordering (gslam_amd/csrc/ba_order.hip: threads, index arithmetic over caller-supplied arrays) under
AddressSanitizer + UndefinedBehaviorSanitizer on the CPU -- GPU sanitizers are not available on this pool.  Builds
tools/order_asan.cpp with g++ (the file is host-only C++ once __HIP_PLATFORM_AMD__ is defined) and runs 40 random graphs."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_camera_order_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    if shutil.which("g++") is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("needs g++ and the HIP headers")
    exe = str(tmp_path / "order_asan")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "gslam_amd", "csrc"), "-I/opt/rocm/include",
           "-x", "c++", os.path.join(ROOT, "gslam_amd", "csrc", "ba_order.hip"), os.path.join(ROOT, "tools", "order_asan.cpp"), "-o", exe, "-lpthread"]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("this g++ has no sanitizer runtime")
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", GSLAM_HIP_HOST_THREADS="8"))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert r.stdout.count("status 0") == 40
