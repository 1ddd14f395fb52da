"""ctypes binding of include/gslam_hip.h (libgslam_hip.so).  Plumbing only: torch supplies device
memory / streams, this module passes raw device pointers through the C ABI.

Fails loudly (ImportError) when the HIP library has not been built: there is no fallback path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSLAM_HIP_LIB: another build of the same library (A/B measurements of two builds on one box: tools/host_call_probe.py)
LIB_PATH = os.environ.get("GSLAM_HIP_LIB") or os.path.join(_HERE, "lib", "libgslam_hip.so")


class GslamHipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: run `make lib` (or __graft_entry__.build()). "
            "gslam_amd has no CPU fallback.")
    return C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)


lib = _load()

GH_OK = 0
GH_BA_MAX_TRACE = 512


class KeyPoint(C.Structure):
    """Layout-identical to GSLAM::KeyPoint (GSLAM/core/Map.h:122-195)."""
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int32), ("class_id", C.c_int32)]


class OrbParams(C.Structure):
    _fields_ = [("n_features", C.c_int32), ("n_levels", C.c_int32), ("ini_th_fast", C.c_int32),
                ("min_th_fast", C.c_int32)]


class OrbStreamResult(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("offsets", C.POINTER(C.c_int32)), ("kps", C.c_void_p), ("desc", C.c_void_p),
                ("gpu_ms", C.c_float)]


class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double)]


class BaProblem(C.Structure):
    _fields_ = [("n_cams", C.c_int32), ("n_points", C.c_int32), ("n_obs", C.c_int32),
                ("cam_pose", C.c_void_p), ("cam_dof", C.c_void_p), ("point_xyz", C.c_void_p),
                ("point_free", C.c_void_p), ("obs_cam", C.c_void_p), ("obs_point", C.c_void_p),
                ("obs_xy", C.c_void_p), ("obs_info", C.c_void_p)]


class PgProblem(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("frame_sim3", C.c_void_p), ("frame_dof", C.c_void_p),
                ("n_se3", C.c_int32), ("se3_first", C.c_void_p), ("se3_second", C.c_void_p), ("se3_meas", C.c_void_p),
                ("se3_info", C.c_void_p),
                ("n_sim3", C.c_int32), ("sim3_first", C.c_void_p), ("sim3_second", C.c_void_p), ("sim3_meas", C.c_void_p),
                ("sim3_info", C.c_void_p),
                ("n_gps", C.c_int32), ("gps_frame", C.c_void_p), ("gps_meas", C.c_void_p), ("gps_info", C.c_void_p)]


class GraphProblem(C.Structure):
    _fields_ = [("pg", PgProblem), ("n_xyz", C.c_int32), ("xyz", C.c_void_p), ("xyz_free", C.c_void_p),
                ("n_idp", C.c_int32), ("idp_host", C.c_void_p), ("idp_anchor", C.c_void_p), ("idp_rho", C.c_void_p),
                ("idp_free", C.c_void_p), ("n_obs", C.c_int32), ("obs_kind", C.c_void_p), ("obs_point", C.c_void_p),
                ("obs_frame", C.c_void_p), ("obs_xy", C.c_void_p), ("obs_info", C.c_void_p), ("projection", C.c_int32),
                ("obs_bearing", C.c_void_p), ("intrinsics", C.c_void_p), ("intrinsics_free", C.c_int32)]


class BaOptions(C.Structure):
    _fields_ = [("huber_delta", C.c_double), ("max_iterations", C.c_int32), ("initial_radius", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("min_relative_decrease", C.c_double), ("verbose", C.c_int32), ("deterministic", C.c_int32)]


class BaSummary(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("accepted", C.c_int32), ("termination", C.c_int32),
                ("initial_cost", C.c_double), ("final_cost", C.c_double), ("solve_ms_total", C.c_double),
                ("total_ms", C.c_double), ("trace_len", C.c_int32),
                ("trace_cost", C.c_double * GH_BA_MAX_TRACE), ("trace_radius", C.c_double * GH_BA_MAX_TRACE),
                ("trace_accepted", C.c_uint8 * GH_BA_MAX_TRACE)]


# Every symbol include/gslam_hip.h declares.  tests/test_abi.py checks header <-> this table <-> .so.
_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
SIGNATURES = {
    "gh_abi_version": (C.c_int, []),
    "gh_ctx_create": (C.c_int, [_i, C.POINTER(_vp)]),
    "gh_ctx_destroy": (None, [_vp]),
    "gh_last_error": (C.c_char_p, [_vp]),
    "gh_ctx_set_stream": (C.c_int, [_vp, _vp]),
    "gh_ctx_use_own_stream": (C.c_int, [_vp]),
    "gh_ctx_stream": (_vp, [_vp]),
    "gh_ctx_sync": (C.c_int, [_vp]),
    "gh_device_info": (C.c_int, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_sz), C.c_char_p, _i]),
    "gh_dev_alloc": (C.c_int, [_vp, _sz, C.POINTER(_vp)]),
    "gh_dev_free": (C.c_int, [_vp, _vp]),
    "gh_dev_upload": (C.c_int, [_vp, _vp, _vp, _sz]),
    "gh_dev_download": (C.c_int, [_vp, _vp, _vp, _sz]),
    "gh_dev_memset": (C.c_int, [_vp, _vp, _i, _sz]),
    "gh_ctx_trim": (C.c_int, [_vp]),
    "gh_magic_div": (C.c_uint32, [C.c_uint32, C.c_uint32]),
    "gh_ctx_set_ba_solver": (C.c_int, [_vp, _i]),
    "gh_ctx_last_ba_solver": (C.c_int, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "gh_ctx_last_ba_order": (C.c_int, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "gh_ba_camera_order": (C.c_int, [_vp, _vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gh_ctx_last_ba_border_points": (C.c_int, [_vp]),
    "gh_prof_enable": (C.c_int, [_vp, _i]),
    "gh_prof_collect": (C.c_int, [_vp, C.POINTER(ProfEntry), _i, C.POINTER(_i)]),
    "gh_bf_match_dev": (C.c_int, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "gh_bf_match_host": (C.c_int, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "gh_bf_match_pairs_dev": (C.c_int, [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "gh_bf_match_bytes_dev": (C.c_int, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "gh_bf_match_bytes_host": (C.c_int, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "gh_bf_match_pairs_bytes_dev": (C.c_int, [_vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "gh_bf_match_pairs_popc_dev": (C.c_int, [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "gh_bf_match_pairs_mfma_dev": (C.c_int, [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "gh_bf_match_band_pairs_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, C.c_float, _vp, _vp, _vp]),
    "gh_match_mask_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "gh_bf_valu_probe": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "gh_valu_issue_probe": (C.c_int, [_vp, _i, C.POINTER(C.c_double), C.c_char_p, _i]),
    "gh_orb_default_params": (None, [C.POINTER(OrbParams)]),
    "gh_orb_plan_create": (C.c_int, [_vp, _i, _i, _i, C.POINTER(OrbParams), C.POINTER(_vp)]),
    "gh_orb_plan_destroy": (None, [_vp]),
    "gh_orb_plan_set_pattern": (C.c_int, [_vp, _vp]),
    "gh_orb_plan_set_steering": (C.c_int, [_vp, _i]),
    "gh_orb_plan_set_distribution": (C.c_int, [_vp, _i]),
    "gh_orb_stream_plan": (_vp, [_vp]),
    "gh_orb_plan_level": (C.c_int, [_vp, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "gh_orb_plan_device_bytes": (_sz, [_vp]),
    "gh_orb_extract_dev": (C.c_int, [_vp, _vp, _i, _sz, _i, _vp, _vp, _vp]),
    "gh_orb_extract_host": (C.c_int, [_vp, _vp, _i, _vp, _vp, C.POINTER(C.c_int32)]),
    "gh_bgr_to_gray_dev": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _vp, _i]),
    "gh_bgr_to_gray_batch_dev": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _sz, _i, _vp, _i, _sz]),
    "gh_orb_stream_create": (C.c_int, [_vp, _i, _i, _i, _i, _sz, _i, _i, C.POINTER(OrbParams), C.POINTER(_vp)]),
    "gh_orb_stream_destroy": (None, [_vp]),
    "gh_orb_stream_staging": (C.c_int, [_vp, C.POINTER(_vp)]),
    "gh_orb_stream_submit": (C.c_int, [_vp, _vp, _i, C.POINTER(C.c_int64)]),
    "gh_orb_stream_poll": (C.c_int, [_vp, C.c_int64, C.POINTER(_i)]),
    "gh_orb_stream_collect": (C.c_int, [_vp, C.c_int64, C.POINTER(OrbStreamResult)]),
    "gh_host_alloc_pinned": (C.c_int, [_vp, _sz, C.POINTER(_vp)]),
    "gh_host_free_pinned": (C.c_int, [_vp, _vp]),
    "gh_orb_plan_debug_counters": (C.c_int, [_vp, _i, _vp]),
    "gh_orb_debug_level": (C.c_int, [_vp, _i, _i, _vp]),
    "gh_synth_frames_dev": (C.c_int, [_vp, _vp, _i, _i, _i, _sz, _i, _i, C.c_uint32]),
    "gh_comm_unique_id": (C.c_int, [_vp]),
    "gh_comm_create_rccl": (C.c_int, [_vp, _i, _i, _vp, C.POINTER(_vp)]),
    "gh_comm_create_ipc": (C.c_int, [_vp, _i, _i, C.c_char_p, C.POINTER(_vp)]),
    "gh_comm_destroy": (None, [_vp]),
    "gh_comm_rank": (C.c_int, [_vp]),
    "gh_comm_world": (C.c_int, [_vp]),
    "gh_comm_buffer": (C.c_int, [_vp, _sz, C.POINTER(_vp)]),
    "gh_allgather": (C.c_int, [_vp, _vp, _vp, _sz]),
    "gh_allgather_features": (C.c_int, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gh_allgather_matches": (C.c_int, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gh_comm_wait": (C.c_int, [_vp]),
    "gh_comm_status": (C.c_int, [_vp]),
    "gh_bow_vocab_create": (C.c_int, [_vp, _i, _i, _i, _i, C.c_uint32, _vp, _vp, C.POINTER(_vp)]),
    "gh_bow_vocab_create_bytes": (C.c_int, [_vp, _i, _i, _i, _i, C.c_uint32, _vp, _vp, _i, C.POINTER(_vp)]),
    "gh_bow_vocab_create_f32": (C.c_int, [_vp, _i, _i, _i, _i, C.c_uint32, _vp, _vp, _i, C.POINTER(_vp)]),
    "gh_bow_vocab_destroy": (None, [_vp]),
    "gh_bow_transform_dev": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gh_bow_transform_host": (C.c_int, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_int32)]),
    "gh_bow_score_dev": (C.c_int, [_vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp]),
    "gh_bow_score_host": (C.c_int, [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    "gh_undist_plan_create": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, C.POINTER(_vp)]),
    "gh_undist_plan_destroy": (None, [_vp]),
    "gh_undistort_dev": (C.c_int, [_vp, _vp, _i, _i, _sz, _vp, _sz, _i]),
    "gh_undistort_host": (C.c_int, [_vp, _vp, _i, _vp, _i]),
    "gh_ransac_estimate": (C.c_int, [_vp, _i, _vp, _vp, _i, C.c_double, C.c_uint64, _vp, _vp, C.POINTER(_i)]),
    "gh_ransac_estimate_conf": (C.c_int, [_vp, _i, _vp, _vp, _i, C.c_double, C.c_double, C.c_uint64, _vp, _vp, C.POINTER(_i),
                                          C.POINTER(_i)]),
    "gh_ransac_estimate_ex": (C.c_int, [_vp, _i, _vp, _vp, _i, C.c_double, C.c_double, C.c_uint64, _i, _vp, _vp, C.POINTER(_i),
                                          C.POINTER(_i)]),
    "gh_triangulate": (C.c_int, [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp]),
    "gh_ba_default_options": (None, [C.POINTER(BaOptions)]),
    "gh_ba_solve": (C.c_int, [_vp, C.POINTER(BaProblem), C.POINTER(BaOptions), C.POINTER(BaSummary)]),
    "gh_ba_graph_create": (C.c_int, [_vp, C.POINTER(BaProblem), C.POINTER(BaOptions), C.POINTER(_vp)]),
    "gh_ba_graph_destroy": (None, [_vp]),
    "gh_ba_graph_update": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gh_ba_graph_solve": (C.c_int, [_vp, C.POINTER(BaOptions), C.POINTER(BaSummary)]),
    "gh_ba_graph_read": (C.c_int, [_vp, _vp, _vp]),
    "gh_ba_pnp": (C.c_int, [_vp, _vp, _vp, _i, _vp, _i, C.POINTER(BaOptions), _vp, C.POINTER(BaSummary)]),
    "gh_pg_solve": (C.c_int, [_vp, C.POINTER(PgProblem), C.POINTER(BaOptions), C.POINTER(BaSummary)]),
    "gh_graph_solve": (C.c_int, [_vp, C.POINTER(GraphProblem), C.POINTER(BaOptions), C.POINTER(BaSummary)]),
    "gh_align_sim3": (C.c_int, [_vp, _vp, _vp, _i, _i, _vp, _vp, C.POINTER(C.c_double), C.POINTER(_i)]),
    "gh_ba_marginalize": (C.c_int, [_vp, _vp, C.c_double, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, C.POINTER(C.c_int32)]),
    "gh_potrf_solve_dev": (C.c_int, [_vp, _vp, _i, _i, _vp, C.POINTER(_i)]),
    "gh_band_solve_dev": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, C.POINTER(_i)]),
    "gh_arrow_solve_dev": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _vp, C.POINTER(_i)]),
    "gh_cr_compact_layout": (C.c_int, [_i, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "gh_arrow_solve_compact_dev": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _vp, C.POINTER(_i)]),
    "gh_cr_border_structure": (C.c_size_t, [_i, _i, _i, _vp, _vp, C.c_size_t]),
    "gh_bs_symbolic": (C.c_int, [_i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _i]),
    "gh_bs_solve_host": (C.c_int, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, C.c_double, _i, _i, _vp, C.POINTER(_i)]),
}


def bind(strict=True):
    """Attach restype/argtypes; with strict=True a missing export is an error."""
    missing = []
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing and strict:
        raise ImportError(f"libgslam_hip.so lacks symbols declared in gslam_hip.h: {missing}")
    return missing


MISSING = bind(strict=False)

# the struct layouts mirrored above are those of ABI version 2 (include/gslam_hip.h: GH_ABI_VERSION): a library built from
# another header version must not be driven through them
ABI_VERSION = 2
if "gh_abi_version" not in MISSING and lib.gh_abi_version() != ABI_VERSION:
    raise ImportError(f"{LIB_PATH} reports ABI version {lib.gh_abi_version()}, this mirror was written for {ABI_VERSION}: rebuild (make lib)")


class Context:
    """Owns a gh_ctx.  `stream` (int hipStream_t) lets kernels run on the caller's torch stream."""

    def __init__(self, device=0, stream=None):
        h = C.c_void_p()
        st = lib.gh_ctx_create(int(device), C.byref(h))
        if st != GH_OK:
            raise GslamHipError(f"gh_ctx_create(device={device}) failed with status {st}: no usable HIP device "
                                "(gslam_amd has no CPU fallback)")
        self.h = h
        if stream is not None:
            self.check(lib.gh_ctx_set_stream(self.h, C.c_void_p(int(stream))))

    def check(self, st):
        if st != GH_OK:
            raise GslamHipError(f"status {st}: {lib.gh_last_error(self.h).decode(errors='replace')}")

    def sync(self):
        self.check(lib.gh_ctx_sync(self.h))

    def close(self):
        if self.h:
            lib.gh_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_info(self):
        cu, clk, mem = C.c_int(), C.c_int(), C.c_size_t()
        name = C.create_string_buffer(128)
        self.check(lib.gh_device_info(self.h, C.byref(cu), C.byref(clk), C.byref(mem), name, 128))
        return {"cu_count": cu.value, "clock_khz": clk.value, "hbm_bytes": mem.value, "name": name.value.decode()}

    def valu_issue_probes(self):
        """{instruction class: measured wave-instructions / s over the whole device}."""
        out = {}
        op = 0
        while True:
            r = C.c_double()
            name = C.create_string_buffer(32)
            if lib.gh_valu_issue_probe(self.h, op, C.byref(r), name, 32) != GH_OK:
                break
            out[name.value.decode()] = r.value
            op += 1
        return out

    def trim(self):
        """gh_ctx_trim: give back scratch, pinned staging and the solver arenas (re-grown on demand)."""
        self.check(lib.gh_ctx_trim(self.h))

    def set_ba_solver(self, solver):
        """0 / "auto", 1 / "dense", 2 / "band": linear solver of the reduced camera system (gh_ctx_set_ba_solver)."""
        code = {"auto": 0, "dense": 1, "band": 2}.get(solver, solver)
        self.check(lib.gh_ctx_set_ba_solver(self.h, int(code)))

    def last_ba_solver(self):
        """("dense" | "band" | "arrow" | None, tiles per superblock, camera span of the band part) of the last BA solve on this context."""
        t, sp = C.c_int(), C.c_int()
        code = lib.gh_ctx_last_ba_solver(self.h, C.byref(t), C.byref(sp))
        return {0: None, 1: "dense", 2: "band", 3: "arrow"}[code], t.value, sp.value

    def last_ba_order(self):
        """(border cameras, reordered) of the last BA solve: gh_ctx_last_ba_order."""
        b, r = C.c_int(), C.c_int()
        lib.gh_ctx_last_ba_order(self.h, C.byref(b), C.byref(r))
        return b.value, bool(r.value)

    def last_ba_border_points(self):
        """long-range points that formed the arrowhead border of the last BA solve (0: camera border or none)"""
        return int(lib.gh_ctx_last_ba_border_points(self.h))

    def prof_enable(self, on=True):
        self.check(lib.gh_prof_enable(self.h, 1 if on else 0))

    def prof_collect(self):
        buf = (ProfEntry * 64)()
        n = C.c_int()
        self.check(lib.gh_prof_collect(self.h, buf, 64, C.byref(n)))
        return {buf[i].name.decode(): {"launches": int(buf[i].launches), "total_ms": float(buf[i].total_ms)}
                for i in range(n.value)}
