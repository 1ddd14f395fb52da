"""Drop-in boundary on the GPU box: a real GSLAM host (build/plugin_host, compiled against the GSLAM
headers in the authoring container) loads libgslam_optimizer.so / libgslam_featuredetector.so through
GSLAM::Optimizer::create() / Registry / dlopen and drives them with GSLAM's own containers.  Results
must equal the C-ABI path (bit-exact for ORB/BF, tolerance for BA)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle_lib
from gslam_amd.ba_synth import make_graph

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "build", "plugin_host")
LIBDIR = os.path.join(ROOT, "gslam_amd", "lib")


def _need_host():
    if not (os.path.exists(HOST) and os.path.exists(os.path.join(LIBDIR, "libgslam_optimizer.so"))):
        # the plugins need the GSLAM headers at build time; a tree built where /root/reference is absent cannot have them
        pytest.skip("build/plugin_host or the plugin .so files are missing: run `make plugins` in the authoring "
                    "container (they travel to the GPU box as built artefacts)")


def _run(args, extra_env=None):
    env = dict(os.environ)
    env.update(extra_env or {})
    env["LD_LIBRARY_PATH"] = LIBDIR + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([HOST] + [str(a) for a in args], capture_output=True, text=True, timeout=300, env=env)
    return r


def test_optimizer_plugin_optimize_matches_oracle(tmp_path, oracle):
    _need_host()
    g = make_graph(10, 200, n_obs_per_point=4, seed=21)
    g["point_free"] = np.ones(200, np.uint8)
    g["point_free"][:5] = 0
    inp, out = tmp_path / "graph.bin", tmp_path / "out.bin"
    nc, npt, no = len(g["cam_pose"]), len(g["point_xyz"]), len(g["obs_cam"])
    with open(inp, "wb") as f:
        f.write(np.array([nc, npt, no, 0, 30, 1], np.int32).tobytes())
        f.write(struct.pack("d", 0.01))
        for k, dt in (("cam_pose", np.float64), ("cam_dof", np.int32), ("point_xyz", np.float64),
                      ("point_free", np.uint8), ("obs_cam", np.int32), ("obs_point", np.int32),
                      ("obs_xy", np.float64)):
            f.write(np.ascontiguousarray(g[k], dtype=dt).tobytes())
    r = _run(["ba", LIBDIR, inp, out])
    assert r.returncode == 0, r.stdout + r.stderr
    assert "optimize=1 optimizePose(unsupported)=0" in r.stdout
    raw = open(out, "rb").read()
    assert struct.unpack("i", raw[:4])[0] == 1
    poses = np.frombuffer(raw, np.float64, nc * 7, 4).reshape(nc, 7)
    pts = np.frombuffer(raw, np.float64, npt * 3, 4 + nc * 56).reshape(npt, 3)
    po, xo, so, _ = oracle.ba_solve(g, oracle_lib.ba_options(max_iterations=30))
    assert np.abs(poses - po).max() < 1e-7 and np.abs(pts - xo).max() < 1e-7
    assert np.array_equal(pts[:5], g["point_xyz"][:5]) and np.array_equal(poses[0], g["cam_pose"][0])


def test_optimizer_plugin_magin(tmp_path, oracle):
    """Optimizer::magin (GSLAM/core/Optimizer.h:230-232) through Optimizer::create(): the SE3 edges equal the oracle's
    (oracle_ba_marginalize, default OptimizerHIP.MaginMinShared = 15), their measurements are T_first^-1 T_second by the
    reference's own SE3 algebra, the observations are gone, and the converted graph runs through optimize() again."""
    _need_host()
    g = make_graph(24, 1500, n_obs_per_point=5, seed=23)
    inp, out = tmp_path / "graph.bin", tmp_path / "out.bin"
    nc, npt, no = len(g["cam_pose"]), len(g["point_xyz"]), len(g["obs_cam"])
    with open(inp, "wb") as f:
        f.write(np.array([nc, npt, no, 0, 30, 0], np.int32).tobytes())
        f.write(struct.pack("d", 0.01))
        for k, dt in (("cam_pose", np.float64), ("cam_dof", np.int32), ("point_xyz", np.float64),
                      ("obs_cam", np.int32), ("obs_point", np.int32), ("obs_xy", np.float64)):
            f.write(np.ascontiguousarray(g[k], dtype=dt).tobytes())
    r = _run(["magin", LIBDIR, inp, out])
    assert r.returncode == 0, r.stdout + r.stderr
    assert "magin=1" in r.stdout and "observations_left=0" in r.stdout and "posegraph_optimize=1" in r.stdout, r.stdout
    worst = float(r.stdout.split("worst_rotation_residual=")[1].split()[0])
    assert worst < 1e-5, r.stdout
    raw = open(out, "rb").read()
    ok, ne = struct.unpack("ii", raw[:8])
    fo, so, no_, lamo = oracle.ba_marginalize(g, huber=0.01, min_shared=15)
    assert ok == 1 and ne == len(fo) > 20
    rec = np.dtype([("ij", np.int32, 2), ("m", np.float64, 7), ("info", np.float64, 36)])
    ed = np.frombuffer(raw, rec, ne, 8)
    assert np.array_equal(ed["ij"][:, 0], fo) and np.array_equal(ed["ij"][:, 1], so)
    scale = np.abs(lamo).reshape(ne, -1).max(axis=1)[:, None]
    assert (np.abs(ed["info"] - lamo.reshape(ne, 36)) <= 1e-11 * scale).all()
    poses = np.asarray(g["cam_pose"], dtype=np.float64)
    for e in range(0, ne, 7):   # T_i^-1 T_j
        i, j = fo[e], so[e]
        qi, ti, qj, tj = poses[i, :4], poses[i, 4:], poses[j, :4], poses[j, 4:]
        qic = np.array([-qi[0], -qi[1], -qi[2], qi[3]])
        def qmul(a, b):
            return np.array([a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1], a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0],
                             a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3], a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2]])
        q = qmul(qic, qj)
        t = qmul(qmul(qic, np.array([*(tj - ti), 0.0])), qi)[:3]
        m = ed["m"][e]
        sgn = np.sign(np.dot(m[:4], q))
        assert np.abs(m[:4] * sgn - q).max() < 1e-12 and np.abs(m[4:] - t).max() < 1e-11


@pytest.mark.parametrize("model,dof_flags", [("OpenCV", 1 | 2 | 4 | 8), ("PinHole", 1 | 2 | 4 | 64)])
def test_optimizer_plugin_self_calibration(tmp_path, ctx, model, dof_flags):
    """BundleGraph::camera + cameraDOF (GSLAM/core/Optimizer.h:86-100,169-171) through Optimizer::create(): the host turns
    pixels into CameraAnchors with the reference's own Camera::UnProject of a WRONG camera, optimize() must hand back the
    camera the pixels were made with (0.1 px of noise) -- and the same camera gh_graph_solve finds on the pixels directly.
    A PinHole camera ignores the distortion flags."""
    _need_host()
    from gslam_amd import posegraph
    from gslam_amd.ba import default_options
    from gslam_amd.pg_synth import make_landmark_graph, with_camera
    truth, start, dof, base = make_landmark_graph(n_frames=10, n_xyz=300, n_idp=0, kind="se3", seed=17, noise=0.0, perturb=0.02,
                                                  point_perturb=0.03, obs_per_point=6)
    cam_true = np.array([260.0, 258.0, 318.0, 242.0, -0.05, 0.01, 8e-4, -5e-4, 0.0])  # (a wide lens: the scene fills 640 x 480)
    cam_start = cam_true * np.array([1.04, 0.97, 1.02, 0.98, 0.9, 1.2, 1, 1, 1])
    if model == "PinHole":
        cam_true[4:] = 0.0
        cam_start[4:] = 0.0
    free = 0b001111 if model == "PinHole" else 0b111111
    prob = with_camera(base, cam_true, cam_start, free, pixel_noise=0.1, seed=4)  # (noise: the solve ends on the function tolerance)
    kind, point, frame, px, _ = prob["obs"]
    inside = (px[:, 0] >= 0) & (px[:, 0] < 640) & (px[:, 1] >= 0) & (px[:, 1] < 480)  # (what a 640 x 480 sensor sees)
    kind, point, frame, px = kind[inside], point[inside], frame[inside], px[inside]
    prob["obs"] = (kind, point, frame, px, None)
    assert inside.sum() > 800
    params = np.concatenate([[640.0, 480.0], cam_start[:4] if model == "PinHole" else cam_start])
    inp, out = tmp_path / "calib.bin", tmp_path / "out.bin"
    nc, npt, no = len(start), len(prob["xyz"][0]), len(kind)
    with open(inp, "wb") as f:
        f.write(np.array([nc, npt, no, dof_flags, 100, len(params)], np.int32).tobytes())
        f.write(struct.pack("d", 0.0))
        f.write(params.tobytes())
        f.write(np.ascontiguousarray(start[:, :7]).tobytes())
        f.write(np.ascontiguousarray(dof, np.int32).tobytes())
        f.write(np.ascontiguousarray(prob["xyz"][0]).tobytes())
        f.write(np.ascontiguousarray(frame, np.int32).tobytes())
        f.write(np.ascontiguousarray(point, np.int32).tobytes())
        f.write(np.ascontiguousarray(px).tobytes())
    r = _run(["calib", LIBDIR, inp, out])
    assert r.returncode == 0 and "calib=1" in r.stdout, r.stdout + r.stderr
    raw = open(out, "rb").read()
    ok, n = struct.unpack("ii", raw[:8])
    got = np.frombuffer(raw, np.float64, n, 8)
    roundtrip = struct.unpack("d", raw[8 + 8 * n:16 + 8 * n])[0]
    assert ok == 1 and n == len(params) and np.array_equal(got[:2], [640.0, 480.0]) and roundtrip < 0.05
    cam = np.zeros(9)
    cam[:n - 2] = got[2:]
    # the anchors carry the error of the reference's 5-step UnProject (measured above, in pixels): recovery to that level
    assert np.allclose(cam[:4], cam_true[:4], rtol=4e-3, atol=20 * roundtrip), (cam, cam_true)
    assert np.allclose(cam[4:6], cam_true[4:6], atol=1e-2) and np.array_equal(cam[6:], cam_start[6:])
    o = default_options()
    o.huber_delta, o.max_iterations = 0.0, 100
    S, xyz, rho, cam_abi, sm, st = posegraph.solve_graph(ctx, start, dof, prob, o)
    assert st in (0, 4) and np.allclose(cam[:4], cam_abi[:4], rtol=2e-3, atol=20 * roundtrip) and np.allclose(cam[4:6], cam_abi[4:6], atol=5e-3)
    poses = np.frombuffer(raw, np.float64, nc * 7, 16 + 8 * n).reshape(nc, 7)
    assert np.allclose(poses, S[:, :7], atol=2e-3)


def test_optimizer_plugin_resident_graph_update_path(tmp_path, oracle):
    """The plugin keeps the graph on the device between optimize() calls (gh_ba_graph_*): a first call on the same
    topology with other values, then the checked call through the update path, must equal the oracle / the one-shot path."""
    _need_host()
    g = make_graph(10, 200, n_obs_per_point=4, seed=21)
    inp = tmp_path / "graph.bin"
    nc, npt = 10, 200
    _write_graph(inp, g, 30, 0.01)
    res = []
    for warm in (False, True):
        out = tmp_path / f"out_{int(warm)}.bin"
        r = _run(["ba", LIBDIR, inp, out], {"GSLAM_HOST_WARM_GRAPH": "1"} if warm else None)
        assert r.returncode == 0, r.stdout + r.stderr
        assert ("warm_optimize=1" in r.stdout) == warm
        raw = open(out, "rb").read()
        res.append((np.frombuffer(raw, np.float64, nc * 7, 4).copy(), np.frombuffer(raw, np.float64, npt * 3, 4 + nc * 56).copy()))
    assert res[0][0].tobytes() == res[1][0].tobytes() and res[0][1].tobytes() == res[1][1].tobytes()
    po, xo, so, _ = oracle.ba_solve(g, oracle_lib.ba_options(max_iterations=30))
    assert np.abs(res[1][0].reshape(nc, 7) - po).max() < 1e-7 and np.abs(res[1][1].reshape(npt, 3) - xo).max() < 1e-7


def _write_graph(path, g, max_it, huber):
    nc, npt, no = len(g["cam_pose"]), len(g["point_xyz"]), len(g["obs_cam"])
    with open(path, "wb") as f:
        f.write(np.array([nc, npt, no, 0, max_it, 0], np.int32).tobytes())
        f.write(struct.pack("d", huber))
        for k, dt in (("cam_pose", np.float64), ("cam_dof", np.int32), ("point_xyz", np.float64),
                      ("obs_cam", np.int32), ("obs_point", np.int32), ("obs_xy", np.float64)):
            f.write(np.ascontiguousarray(g[k], dtype=dt).tobytes())


def test_optimizer_plugin_sim3_scale_and_bad_anchor(tmp_path, oracle):
    """Keyframes are SIM3 (Optimizer.h:116-119).  The pinhole residual does not depend on the scale, so keyframes with
    s = 2.5 must give the SAME (R, t, points) as s = 1, keep s, and the sum of squared reprojection errors evaluated
    by the host through the reference's own SIM3 operators (scale included) must equal the oracle's cost at the
    result.  A measurement with z = 0 is a caller bug: optimize() returns false instead of guessing z = 1."""
    _need_host()
    g = make_graph(8, 150, n_obs_per_point=4, seed=33)
    inp = tmp_path / "graph.bin"
    _write_graph(inp, g, 25, 0.0)
    nc, npt = 8, 150
    res = {}
    for scale in (1.0, 2.5):
        out = tmp_path / f"out_{scale}.bin"
        r = _run(["ba", LIBDIR, inp, out, scale])
        assert r.returncode == 0, r.stdout + r.stderr
        raw = open(out, "rb").read()
        poses = np.frombuffer(raw, np.float64, nc * 7, 4).reshape(nc, 7)
        pts = np.frombuffer(raw, np.float64, npt * 3, 4 + nc * 56).reshape(npt, 3)
        kv = dict(tok.split("=") for line in r.stdout.splitlines() if line.startswith("ref_sim3_ssq") for tok in line.split())
        assert float(kv["scale_min"]) == float(kv["scale_max"]) == scale
        res[scale] = (poses, pts, float(kv["ref_sim3_ssq"]))
    assert res[1.0][0].tobytes() == res[2.5][0].tobytes() and res[1.0][1].tobytes() == res[2.5][1].tobytes()
    po, xo, so, _ = oracle.ba_solve(g, oracle_lib.ba_options(huber=0.0, max_iterations=25))
    assert np.abs(res[2.5][0] - po).max() < 1e-7 and np.abs(res[2.5][1] - xo).max() < 1e-7
    for scale in (1.0, 2.5):
        assert abs(0.5 * res[scale][2] - so.final_cost) <= 1e-7 * so.final_cost
    r = _run(["ba", LIBDIR, inp, tmp_path / "bad.bin", 1.0, 17])
    assert r.returncode == 3 and "optimize=0" in r.stdout
    r = _run(["ba", LIBDIR, inp, tmp_path / "bad2.bin", -1.0])
    assert r.returncode == 3 and "optimize=0" in r.stdout


def test_optimizer_plugin_pnp(tmp_path, oracle):
    _need_host()
    g = make_graph(3, 150, n_obs_per_point=3, seed=11, noise=0.0, outlier_frac=0.0, perturb=False)
    sel = g["obs_cam"] == 1
    X = g["point_xyz_gt"][g["obs_point"][sel]]
    m = g["obs_xy"][sel]
    truth = g["cam_pose_gt"][1]
    start = oracle.se3_retract(truth, np.array([0.05, -0.04, 0.03, 0.01, -0.02, 0.015]))
    inp, out = tmp_path / "pnp.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("i", len(X)))
        f.write(X.astype(np.float64).tobytes() + m.astype(np.float64).tobytes() + start.tobytes())
    r = _run(["pnp", LIBDIR, inp, out])
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(out, "rb").read()
    pose = np.frombuffer(raw, np.float64, 7, 4)
    info = np.frombuffer(raw, np.float64, 36, 4 + 56).reshape(6, 6)
    assert np.abs(pose - truth).max() < 1e-9
    assert np.allclose(info, info.T) and np.all(np.linalg.eigvalsh(info) > 0)


def test_optimizer_plugin_pnp_under_sphere_projection(tmp_path, oracle):
    """optimizePnP with OptimzeConfig::cameraProjectionType = PROJECTION_SPHERE (Optimizer.h:58-61,174-176,202-207): the
    measurements are bearings (here incl. ones BEHIND the z = 1 plane's reach: a 200-degree field of view), the pose comes
    back through the general graph solver with the tangent-plane residual.  Against the ground truth, against the oracle's
    graph solver on the same one-keyframe problem, and the information matrix against an independent numpy J^T J."""
    _need_host()
    rng = np.random.default_rng(4)
    n = 160
    truth = np.array([0.0, 0.0, 0.0, 1.0, 0.3, -0.2, 0.1])
    truth[:4] = [0.05, -0.08, 0.02, 1.0]
    truth[:4] /= np.linalg.norm(truth[:4])
    from gslam_amd.ba_synth import quat_to_R
    R = quat_to_R(truth[None, :4])[0]
    # points all around the camera (wide field of view: z in the camera frame may be <= 0)
    d = rng.normal(size=(n, 3))
    d[: n // 2, 2] = np.abs(d[: n // 2, 2]) + 0.5
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    Xc = d * rng.uniform(2, 9, (n, 1))
    X = Xc @ R.T + truth[4:]
    m = d * rng.uniform(0.5, 2.0, (n, 1))  # bearings of any length
    assert (Xc[:, 2] <= 0).sum() > 10
    start = oracle.se3_retract(truth, np.array([0.05, -0.04, 0.03, 0.01, -0.02, 0.015]))
    inp, out = tmp_path / "pnp.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("i", n))
        f.write(X.astype(np.float64).tobytes() + m.astype(np.float64).tobytes() + start.tobytes())
    r = _run(["pnp", LIBDIR, inp, out, 1])
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(out, "rb").read()
    pose = np.frombuffer(raw, np.float64, 7, 4)
    info = np.frombuffer(raw, np.float64, 36, 4 + 56).reshape(6, 6)
    assert np.abs(pose - truth).max() < 1e-8
    # the oracle's general graph solver on the same problem
    problem = {"xyz": (X, np.zeros(n, np.uint8)), "projection": "sphere",
               "obs": (np.zeros(n, np.int32), np.arange(n, dtype=np.int32), np.zeros(n, np.int32), d, None)}
    fo, _, _, so, st = oracle.graph_solve(np.r_[start, 1.0][None], np.array([63], np.int32), problem,
                                          oracle_lib.ba_options(huber=0.01, max_iterations=500))
    assert np.abs(fo[0, :7] - pose).max() < 1e-8 and abs(fo[0, 7] - 1.0) < 1e-15
    # information = sum J^T J of the tangent-plane residuals w.r.t. T_wc <- T_wc exp(delta), independent float64 restatement
    def resid(p7):
        Rp = quat_to_R(p7[None, :4])[0]
        u = (X - p7[4:]) @ Rp
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        return u
    J = np.zeros((n, 3, 6))
    h = 1e-6
    for c in range(6):
        e = np.zeros(6); e[c] = h
        J[:, :, c] = (resid(oracle.se3_retract(pose, e)) - resid(oracle.se3_retract(pose, -e))) / (2 * h)
    # project onto the tangent plane at the measured bearing: P = I - b b^T (J^T P J == sum over any orthonormal tangent basis)
    P = np.eye(3)[None] - d[:, :, None] * d[:, None, :]
    exp_info = np.einsum("nac,nab,nbd->cd", J, P, J)
    assert np.allclose(info, exp_info, rtol=1e-5, atol=1e-7 * np.abs(exp_info).max())


@pytest.mark.parametrize("channels", [1, 3])
def test_featuredetector_plugin_matches_oracle(tmp_path, oracle, channels):
    _need_host()
    w, h, K = 640, 480, 800
    gray = oracle.synth_frame(w, h, 4242)
    if channels == 1:
        img, expect_gray = gray, gray
    else:
        rng = np.random.default_rng(5)
        img = np.stack([gray, np.roll(gray, 3, axis=1), rng.integers(0, 256, gray.shape, dtype=np.uint8)], axis=2)
        expect_gray = oracle.bgr_to_gray(img)
    inp, out = tmp_path / "img.raw", tmp_path / "out.bin"
    img.tofile(inp)
    r = _run(["orb", LIBDIR, w, h, channels, inp, out, K])
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(out, "rb").read()
    ok, n, okm, nm = struct.unpack("4i", raw[:16])
    ek, ed = oracle.orb_extract(expect_gray, K)
    assert ok == 1 and n == len(ek)
    kps = np.frombuffer(raw, oracle_lib.KP_DTYPE, n, 16)
    desc = np.frombuffer(raw, np.uint8, n * 32, 16 + n * 28).reshape(n, 32)
    assert kps.tobytes() == ek.tobytes() and np.array_equal(desc, ed)
    matches = np.frombuffer(raw, np.int32, nm * 2, 16 + n * 60).reshape(nm, 2)
    # self-match with cross-check: every row matches itself unless an identical earlier row exists
    e = oracle.bf_match(ed, ed)
    keep = oracle.match_mask(e[0], e[1], e[2], e[0], n, 100, 0, 1, 1)
    exp_matches = np.stack([np.nonzero(keep)[0], e[0][keep == 1]], axis=1).astype(np.int32)
    assert okm == 1 and np.array_equal(matches, exp_matches)


def test_featuredetector_plugin_orbslam_mode(tmp_path, oracle):
    """svar FeatureDetectorHIP.Distribution = 1 + FeatureDetectorHIP.Steering = 1: ORB-SLAM's cell / quadtree distribution
    and continuous steering behind FeatureDetector::detectAndCompute, against the oracle's steps 4', 5', 6', 8'."""
    _need_host()
    w, h, K = 752, 480, 1200
    gray = oracle.synth_frame(w, h, 777)
    inp, out = tmp_path / "img.raw", tmp_path / "out.bin"
    gray.tofile(inp)
    r = _run(["orb", LIBDIR, w, h, 1, inp, out, K],
             {"GSLAM_HOST_SVAR": "FeatureDetectorHIP.Steering=1;FeatureDetectorHIP.Distribution=1"})
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(out, "rb").read()
    ok, n, okm, nm = struct.unpack("4i", raw[:16])
    oracle.orb_set_distribution(1)
    oracle.orb_set_steer(1)
    try:
        ek, ed = oracle.orb_extract(gray, K)
    finally:
        oracle.orb_set_distribution(0)
        oracle.orb_set_steer(0)
    e0, _ = oracle.orb_extract(gray, K)
    assert ok == 1 and n == len(ek)
    kps = np.frombuffer(raw, oracle_lib.KP_DTYPE, n, 16)
    desc = np.frombuffer(raw, np.uint8, n * 32, 16 + n * 28).reshape(n, 32)
    assert kps.tobytes() == ek.tobytes() and np.array_equal(desc, ed)
    assert kps.tobytes() != e0[:n].tobytes()  # (the options did reach the plugin)


@pytest.mark.parametrize("desc_bytes", [32, 64, 40, -64])
def test_vocabulary_plugin_equals_reference_base_class(tmp_path, oracle, desc_bytes):
    """libgslam_vocabulary.so (VocabularyHIP) vs GSLAM::Vocabulary itself, both inside the GSLAM host process:
    BowVector and FeatureVector maps must compare equal (operator== on the std::maps, i.e. bit-exact floats)."""
    _need_host()
    from gslam_amd import bow_synth
    if desc_bytes < 0:  # a float (L2) vocabulary of -desc_bytes dimensions: l2generic
        voc = bow_synth.make_float_vocabulary(k=8, L=3, dims=-desc_bytes, seed=5)
        desc = bow_synth.float_features_near_words(voc, 1500, seed=8)
    else:
        voc = bow_synth.make_vocabulary(k=10, L=4, seed=5, desc_bytes=desc_bytes)  # hamming32 / hamming64 / hamming8x
        desc = bow_synth.features_near_words(voc, 1500, seed=8)
    gb, df, out = tmp_path / "voc.gbow", tmp_path / "desc.raw", tmp_path / "out.bin"
    open(gb, "wb").write(bow_synth.to_gbow_bytes(voc))
    desc.tofile(df)
    r = _run(["bow", LIBDIR, gb, df, 1500, 2, out])
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bow gpu==reference:1" in r.stdout
    assert "database vectors, mismatches=0" in r.stdout  # scoreVocabularyBatch == the reference's own score()
    raw = open(out, "rb").read()
    same, nwords, nnodes, ncpu = struct.unpack("4i", raw[:16])
    e = oracle.bow_transform(voc, desc, 2)
    assert same == 1 and nwords == ncpu == len(e[3])
    rec = np.frombuffer(raw, np.dtype([("id", "<u8"), ("v", "<f4")]), nwords, 16)
    assert np.array_equal(rec["id"], e[3]) and rec["v"].tobytes() == e[4].tobytes()


def test_vocabulary_plugin_overloads_and_large_images(tmp_path, oracle):
    """The std::vector<TinyMat> overload, the single-feature transform -> WordId and an image of 40 000 features (more than
    the 16384 one library call sorts: chunked descent + the reference's own accumulation) equal the base class; the failure
    counter stays 0."""
    _need_host()
    from gslam_amd import bow_synth
    voc = bow_synth.make_vocabulary(k=10, L=3, seed=9)
    gb = tmp_path / "voc.gbow"
    open(gb, "wb").write(bow_synth.to_gbow_bytes(voc))
    for n in (700, 40000):
        df, out = tmp_path / f"desc{n}.raw", tmp_path / f"out{n}.bin"
        desc = np.concatenate([bow_synth.features_near_words(voc, n - n // 5, seed=n), oracle_lib.random_descriptors(n // 5, 0xAB + n)])
        desc.tofile(df)
        r = _run(["bow", LIBDIR, gb, df, n, 1, out])
        assert r.returncode == 0, r.stdout + r.stderr
        assert "bow gpu==reference:1" in r.stdout and "list overload equal=1 single-feature word mismatches=0 failures=0" in r.stdout


@pytest.mark.parametrize("channels", [1, 3])
def test_undistorter_hip_equals_reference_class(tmp_path, channels):
    """UndistorterHIP (gslam_amd/plugin/UndistorterHIP.h) vs GSLAM::Undistorter in the same GSLAM host process."""
    _need_host()
    rng = np.random.default_rng(channels)
    img = rng.integers(0, 256, (240, 320, channels), dtype=np.uint8)
    f = tmp_path / "img.raw"
    img.tofile(f)
    r = _run(["undist", LIBDIR, channels, f])
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches=0" in r.stdout


def test_application_plugin_on_the_message_bus(tmp_path, oracle):
    """BASELINE configs[0] plumbing: the `orbhip` application plugin is loaded the way GSLAM's launcher loads apps
    (Registry::load -> gslam.apps.orbhip, shared Messenger), receives "dataset/frame", publishes "orbhip/curframe"
    with keypoints + descriptors deposited through MapFrame::setKeyPoints.  Outputs equal the oracle bit for bit."""
    _need_host()
    if not os.path.exists(os.path.join(LIBDIR, "libgslam_orbhip.so")):
        pytest.skip("libgslam_orbhip.so not built")
    w, h, n, K = 640, 480, 5, 1000
    frames = np.stack([oracle.synth_frame(w, h, 0x5EED0000 + i) for i in range(n)])
    # make consecutive frames overlap so that matching is non-trivial: frame i = crop of a wider panorama
    pano = np.concatenate([oracle.synth_frame(w, h, 77), oracle.synth_frame(w, h, 78)], axis=1)
    frames = np.stack([pano[:, 32 * i:32 * i + w] for i in range(n)])
    fin, out = tmp_path / "frames.raw", tmp_path / "out.bin"
    frames.tofile(fin)
    r = _run(["app", LIBDIR, w, h, n, fin, out, K])
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(out, "rb").read()
    n_out, n_msgs = struct.unpack("2i", raw[:8])
    assert n_out == n and n_msgs == n
    off = 8
    prev = None
    for i in range(n):
        fid, nk, nm = struct.unpack("3i", raw[off:off + 12])
        off += 12
        ek, ed = oracle.orb_extract(frames[i], K)
        assert fid == i + 1 and nk == len(ek)
        kps = np.frombuffer(raw, oracle_lib.KP_DTYPE, nk, off)
        off += nk * 28
        desc = np.frombuffer(raw, np.uint8, nk * 32, off).reshape(nk, 32)
        off += nk * 32
        assert kps.tobytes() == ek.tobytes() and np.array_equal(desc, ed)
        if prev is None:
            assert nm == 0
        else:
            f = oracle.bf_match(ed, prev)
            b = oracle.bf_match(prev, ed)
            keep = oracle.match_mask(f[0], f[1], f[2], b[0], len(prev), 100, 0, 1, 1)
            assert nm == int(keep.sum()) and nm > 100  # 32 px = one cell shift: many exact re-detections
        prev = ed


@pytest.mark.parametrize("model", [0, 1, 2])
def test_estimator_plugin_through_estimator_create(tmp_path, oracle, model):
    """GSLAM::Estimator::create() loads libgslam_estimator.so (createEstimatorInstance, Estimator.h:42-53,175-191);
    findHomography / findAffine2D / findFundamental return the oracle's model and inlier mask bit for bit."""
    _need_host()
    if not os.path.exists(os.path.join(LIBDIR, "libgslam_estimator.so")):
        pytest.skip("libgslam_estimator.so not built")
    from test_ransac_oracle import _corr
    n, thr = 900, 2.0 if model < 2 else 1.0
    P, Q, inl, _ = _corr(model, n, 0.3, 40 + model, 0.3)
    fin, out = tmp_path / "pts.raw", tmp_path / "out.bin"
    np.ascontiguousarray(np.c_[P, Q], dtype=np.float64).tofile(fin)
    r = _run(["est", LIBDIR, model, n, fin, thr, out])
    assert r.returncode == 0, r.stdout + r.stderr
    assert "EstimatorHIP ok=1 unsupported_paths=0" in r.stdout
    raw = open(out, "rb").read()
    ok, nm = struct.unpack("2i", raw[:8])
    m = np.frombuffer(raw, np.float64, 9, 8)
    mask = np.frombuffer(raw, np.uint8, nm, 8 + 72)
    em, emask, ecnt, _ = oracle.ransac_conf(model, P, Q, thr, 0.99, seed=1)  # the host passes confidence 0.99
    ms = 6 if model == 1 else 9
    assert ok == 1 and nm == n and np.array_equal(mask, emask) and m[:ms].tobytes() == em[:ms].tobytes()
    # the sampling flags of EstimatorMethod (Estimator.h:86-89) through the same virtuals: `X | NOSAMPLE`, bare `LMEDS`
    o = 8 + 72 + nm
    for sampling in (2, 1):
        ok2, nm2 = struct.unpack_from("2i", raw, o)
        m2 = np.frombuffer(raw, np.float64, 9, o + 8)
        mask2 = np.frombuffer(raw, np.uint8, nm2, o + 80)
        o += 80 + nm2
        if sampling == 1 and model == 2:  # LMEDS == F8_Point: findFundamental cannot be asked for it through `method`
            assert ok2 == 0
            continue
        xm, xmask, xcnt, _ = oracle.estimate_ex(model, P, Q, thr, sampling, confidence=0.99, seed=1)
        if xcnt == 0:  # (the least-squares fit over 30 % outliers may have nobody within the threshold: "no model", false)
            assert ok2 == 0, sampling
            continue
        assert ok2 == 1 and nm2 == n and np.array_equal(mask2, xmask) and m2[:ms].tobytes() == xm[:ms].tobytes(), sampling


@pytest.mark.parametrize("model", [4, 5, 6, 7, 8])
def test_estimator_plugin_remaining_solvers(tmp_path, oracle, model):
    """findEssentialMatrix / findSIM3 / findPlane / findPnP / trianglate through GSLAM::Estimator::create()
    (GSLAM/core/Estimator.h:118-169): the plugin's answers equal the oracle's (RANSAC part bit for bit; findPnP's
    motion-only refinement on the inliers to 1e-8)."""
    _need_host()
    from test_ransac_gpu import _cases
    from test_ransac_oracle import _rot
    fin, out = tmp_path / "pts.raw", tmp_path / "out.bin"
    if model == 4:
        P, Q, thr = _cases()[4]
        n = len(P)
        np.ascontiguousarray(np.c_[P, Q], dtype=np.float64).tofile(fin)
        r = _run(["est", LIBDIR, 4, n, fin, thr, out])
        assert r.returncode == 0, r.stdout + r.stderr
        raw = open(out, "rb").read()
        ok, nm = struct.unpack("2i", raw[:8])
        m = np.frombuffer(raw, np.float64, 9, 8)
        mask = np.frombuffer(raw, np.uint8, nm, 8 + 72)
        em, emask, _, _ = oracle.ransac_conf(4, P, Q, thr, 0.99, seed=1)
        assert ok == 1 and np.array_equal(mask, emask) and m.tobytes() == em[:9].tobytes()
        return
    if model == 8:
        from gslam_amd.ba_synth import _quat_from_R
        rng = np.random.default_rng(3)
        R, t = _rot([0.1, 1, 0.2], 0.2), np.array([-0.8, 0.02, 0.05])
        pose = np.r_[_quat_from_R(R[None])[0], t]
        X = np.c_[rng.uniform(-2, 2, (40, 2)), rng.uniform(3, 10, 40)]
        X2 = X @ R.T + t
        d1, d2 = X / X[:, 2:3], X2 / X2[:, 2:3]
        d2[7] = R @ d1[7]
        with open(fin, "wb") as f:
            f.write(pose.tobytes() + np.ascontiguousarray(np.c_[d1, d2]).tobytes())
        r = _run(["est3", LIBDIR, 8, 40, fin, 0.0, out])
        assert r.returncode == 0, r.stdout + r.stderr
        raw = open(out, "rb").read()
        ok, nm, nv = struct.unpack("3i", raw[:12])
        pts = np.frombuffer(raw, np.float64, nv, 12).reshape(-1, 3)
        mask = np.frombuffer(raw, np.uint8, nm, 12 + 8 * nv)
        for i in range(40):
            e, eok = oracle.triangulate(pose, d1[i], d2[i])
            assert bool(mask[i]) == eok and (not eok or pts[i].tobytes() == e.tobytes())
        assert mask.sum() == 39 and np.abs(pts[0] - X[0]).max() < 1e-9
        return
    P, Q, thr = _cases()[model]
    n = len(P)
    rows = np.zeros((n, 6))
    rows[:, :3] = P
    rows[:, 3:3 + Q.shape[1]] = Q
    rows.tofile(fin)
    r = _run(["est3", LIBDIR, model, n, fin, thr, out])
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(out, "rb").read()
    ok, nm, nv = struct.unpack("3i", raw[:12])
    m = np.frombuffer(raw, np.float64, nv, 12)
    mask = np.frombuffer(raw, np.uint8, nm, 12 + 8 * nv)
    em, emask, ecnt, _ = oracle.ransac_conf(model, P, Q, thr, 0.99, seed=1)  # the host passes confidence 0.99
    assert ok == 1 and np.array_equal(mask, emask)
    if model == 5:
        assert m.tobytes() == em[:8].tobytes()
    elif model == 6:
        # the normal comes back through SE3's quaternion (z axis of the plane pose): rounding of that round trip only
        assert np.abs(m[:3] - em[:3]).max() < 1e-12 and np.abs(m[3:6] + em[3] * em[:3]).max() < 1e-12
    else:  # world -> camera pose after the refinement on the inliers
        R = em[:9].reshape(3, 3)
        from gslam_amd.ba_synth import _quat_from_R
        start = np.r_[_quat_from_R(R.T[None])[0], -R.T @ em[9:12]]  # T_wc
        inl = emask.astype(bool)
        po, so, _, rc = oracle.ba_pnp(P[inl], Q[inl], start, opts=oracle_lib.ba_options(huber=thr, max_iterations=30))
        from gslam_amd.ba_synth import quat_to_R
        Rwc = quat_to_R(po[None, :4])[0]
        qcw = _quat_from_R(Rwc.T[None])[0]
        tcw = -Rwc.T @ po[4:]
        got_q = m[:4] * np.sign(m[3]) if m[3] != 0 else m[:4]
        exp_q = qcw * np.sign(qcw[3]) if qcw[3] != 0 else qcw
        assert np.abs(got_q - exp_q).max() < 1e-8 and np.abs(m[4:7] - tcw).max() < 1e-8
