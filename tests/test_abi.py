"""The C-ABI library loads and exports every symbol include/gslam_hip.h declares; the ctypes table and
the header agree; record layouts match the reference's (no compute calls: runs without a GPU)."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "gslam_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gh_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    from gslam_amd import hip
    names = _header_functions()
    assert len(names) >= 30
    assert sorted(hip.SIGNATURES) == names, "gslam_amd/hip.py SIGNATURES and include/gslam_hip.h disagree"
    assert hip.bind(strict=False) == []
    for n in names:
        assert getattr(hip.lib, n) is not None
    assert hip.lib.gh_abi_version() == 2  # (bumped in round 5: gh_graph_problem grew two fields in round 4, new entry points)


def test_record_layouts():
    from gslam_amd import hip
    from gslam_amd.orb import KP_DTYPE
    assert C.sizeof(hip.KeyPoint) == 28 == KP_DTYPE.itemsize  # sizeof(GSLAM::KeyPoint), Map.h:122-195
    assert C.sizeof(hip.OrbParams) == 16
    assert C.sizeof(hip.ProfEntry) == 64
    assert hip.BaSummary.trace_cost.offset % 8 == 0
    o = hip.BaOptions()
    hip.lib.gh_ba_default_options(C.byref(o))
    assert (o.huber_delta, o.max_iterations) == (0.01, 500)  # OptimzeConfig defaults, Optimizer.h:174-182
    assert (o.initial_radius, o.function_tolerance, o.min_relative_decrease) == (1e4, 1e-6, 1e-3)
    p = hip.OrbParams()
    hip.lib.gh_orb_default_params(C.byref(p))
    assert (p.n_features, p.n_levels, p.ini_th_fast, p.min_th_fast) == (1000, 8, 20, 7)


def test_reference_layout_sizes_if_reference_built():
    import oracle_lib
    if not oracle_lib.have_reference():
        import pytest
        pytest.skip("oracle/_ref not built")
    ref = oracle_lib.load_reference()
    assert ref.lib.ref_sizeof_keypoint() == 28
    assert ref.lib.ref_sizeof_se3() == 56 and ref.lib.ref_sizeof_sim3() == 64


def test_no_gpu_means_loud_failure():
    """Without a device gh_ctx_create must fail (there is no CPU fallback).  On a GPU box it succeeds."""
    import torch
    from gslam_amd import hip
    if torch.cuda.is_available():
        c = hip.Context(0)
        c.close()
        return
    h = C.c_void_p()
    assert hip.lib.gh_ctx_create(0, C.byref(h)) != 0 and not h.value
    try:
        hip.Context(0)
        raise AssertionError("Context() must raise without a GPU")
    except hip.GslamHipError:
        pass


def test_product_never_imports_oracle():
    """The product path (package + plugins + csrc) must not reference oracle/ in any way."""
    bad = []
    for base in ("gslam_amd",):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".hip", ".h", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"oracle_lib|liboracle|#include\s+\"[^\"]*oracle/|import oracle|from oracle", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_library_override_for_ab_measurements(tmp_path):
    """GSLAM_HIP_LIB points the ctypes mirror at another build of the same library (tools/host_call_probe.py measures two
    builds on one box that way); a path that does not exist fails loudly, like a missing library."""
    import shutil
    import subprocess
    import sys
    from gslam_amd import hip
    other = tmp_path / "libgslam_hip_other.so"
    shutil.copy(os.path.join(ROOT, "gslam_amd", "lib", "libgslam_hip.so"), other)
    code = "from gslam_amd import hip; print(hip.LIB_PATH, hip.lib.gh_abi_version())"
    env = dict(os.environ, GSLAM_HIP_LIB=str(other), PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=120)
    assert r.returncode == 0 and r.stdout.split() == [str(other), "2"], r.stderr[-400:]
    env["GSLAM_HIP_LIB"] = str(tmp_path / "missing.so")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=120)
    assert r.returncode != 0 and "ImportError" in r.stderr
    assert hip.LIB_PATH.endswith(os.path.join("gslam_amd", "lib", "libgslam_hip.so"))
