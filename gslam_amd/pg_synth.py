"""Synthetic pose graphs for tests / bench: a camera trajectory (SIM3 keyframes T_wc) on a loop, odometry edges between
neighbours, loop-closure edges across the loop, optional GPS priors.  Measurements follow GSLAM/core/Optimizer.h:127-148:
SE3Edge / SIM3Edge measurement = S_first^-1 * S_second, GPSEdge measurement = SE3 of the frame."""
import numpy as np


def _qmul(a, b):
    return np.array([a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1], a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2],
                     a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0], a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2]])


def _qrot(q, p):
    u = 2.0 * np.cross(q[:3], p)
    return p + q[3] * u + np.cross(q[:3], u)


def sim3_mul(a, b):
    return np.concatenate([_qmul(a[:4], b[:4]), a[4:7] + _qrot(a[:4], a[7] * b[4:7]), [a[7] * b[7]]])


def sim3_inv(a):
    qc = np.array([-a[0], -a[1], -a[2], a[3]])
    return np.concatenate([qc, -(1.0 / a[7]) * _qrot(qc, a[4:7]), [1.0 / a[7]]])


def _quat_from_rotvec(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.array([0.5 * w[0], 0.5 * w[1], 0.5 * w[2], 1.0])
    return np.concatenate([np.sin(0.5 * th) * w / th, [np.cos(0.5 * th)]])


def make_pose_graph(n_frames=40, n_loops=8, kind="sim3", seed=1, noise=0.0, perturb=0.05, scale_drift=0.0, gps_every=0,
                    with_info=False):
    """Returns (truth n x 8, start n x 8, dof n, problem dict in the layout gslam_amd.posegraph.solve takes)."""
    rng = np.random.default_rng(seed)
    truth = np.zeros((n_frames, 8))
    for i in range(n_frames):
        a = 2 * np.pi * i / n_frames
        pos = np.array([6 * np.cos(a), 6 * np.sin(a), 0.5 * np.sin(3 * a)])
        q = _quat_from_rotvec(np.array([0.1 * np.sin(a), 0.15 * np.cos(2 * a), a + np.pi / 2]))
        truth[i] = np.concatenate([q, pos, [1.0 if kind != "sim3" else np.exp(scale_drift * np.sin(a))]])

    def rel(i, j):
        m = sim3_mul(sim3_inv(truth[i]), truth[j])
        if noise > 0:
            d = rng.normal(size=7) * noise
            m = sim3_mul(m, np.concatenate([_quat_from_rotvec(d[3:6]), d[:3], [np.exp(d[6] if kind == "sim3" else 0.0)]]))
        return m

    pairs = [(i, i + 1) for i in range(n_frames - 1)]
    for k in range(n_loops):
        i = int(rng.integers(0, n_frames // 3))
        pairs.append((i, n_frames - 1 - int(rng.integers(0, n_frames // 3))))
    pairs.append((n_frames - 1, 0))
    first = np.array([p[0] for p in pairs], np.int32)
    second = np.array([p[1] for p in pairs], np.int32)
    meas = np.stack([rel(i, j) for i, j in pairs])
    problem = {}

    def spd(dim):
        a = rng.normal(size=(dim, dim)) * 0.2
        return (np.eye(dim) * (1.0 + rng.random()) + a @ a.T).reshape(-1)

    if kind == "sim3":
        problem["sim3"] = (first, second, meas, np.stack([spd(7) for _ in pairs]) if with_info else None)
    else:
        problem["se3"] = (first, second, meas[:, :7].copy(), np.stack([spd(6) for _ in pairs]) if with_info else None)
    if kind == "mixed":  # half of the loop closures as SIM3 edges on top
        h = len(pairs) // 2
        problem["sim3"] = (first[h:], second[h:], meas[h:], None)
    if gps_every:
        fr = np.arange(0, n_frames, gps_every, dtype=np.int32)
        gm = truth[fr, :7].copy()
        if noise > 0:
            gm[:, 4:7] += rng.normal(size=(len(fr), 3)) * noise
        problem["gps"] = (fr, gm, np.stack([spd(6) for _ in fr]) if with_info else None)
    start = truth.copy()
    for i in range(1, n_frames):
        d = rng.normal(size=7) * perturb
        if kind != "sim3":
            d[6] = 0.0
        start[i] = sim3_mul(truth[i], np.concatenate([_quat_from_rotvec(d[3:6]), d[:3], [np.exp(d[6])]]))
    dof = np.full(n_frames, 127 if kind in ("sim3", "mixed") else 63, np.int32)
    dof[0] = 0  # gauge
    return truth, start, dof, problem


def make_landmark_graph(n_frames=8, n_xyz=30, n_idp=30, kind="se3", seed=1, noise=0.0, perturb=0.03, point_perturb=0.05,
                        obs_per_point=4, pose_edges=False, with_info=False, observe_host=True, outliers=0.0,
                        projection="pinhole"):
    """A general BundleGraph (GSLAM/core/Optimizer.h:150-172): keyframes on the loop of make_pose_graph looking up at a
    cloud of landmarks, `n_xyz` of them as world points, `n_idp` as inverse-depth points anchored in a host keyframe;
    pinhole observations m = (x / z, y / z) in the observing camera; optionally the odometry / loop edges on top.
    Returns (truth frames, start frames, dof, problem) with problem = the pose-edge dict of make_pose_graph plus
      "xyz": (points n x 3 START values, free mask), "idp": (host, anchor n x 3, rho START values, free mask),
      "obs": (kind, point, frame, xy n x 2, info n x 4 | None), "truth_xyz", "truth_rho".
    projection="sphere" (PROJECTION_SPHERE): anchors and measurements are unit bearings (xy becomes n x 3), rho an inverse
    range, and "projection": "sphere" is set in the problem."""
    sphere = projection == "sphere"
    rng = np.random.default_rng(seed)
    truth, start, dof, problem = make_pose_graph(n_frames, n_loops=2, kind="sim3" if kind == "sim3" else "se3", seed=seed,
                                                 noise=noise, perturb=perturb)
    if not pose_edges:
        problem = {}
    if kind != "sim3":
        truth[:, 7] = 1.0
        start[:, 7] = 1.0

    def cam_coords(S, X):
        qc = np.array([-S[0], -S[1], -S[2], S[3]])
        return _qrot(qc, X - S[4:7]) / S[7]

    def sample_point():
        return np.array([rng.uniform(-4, 4), rng.uniform(-4, 4), rng.uniform(5, 11)])

    okind, opoint, oframe, oxy = [], [], [], []
    xyz = np.zeros((n_xyz, 3))
    for p in range(n_xyz):
        xyz[p] = sample_point()
        frames = rng.choice(n_frames, size=min(obs_per_point, n_frames), replace=False)
        for j in frames:
            Xc = cam_coords(truth[j], xyz[p])
            assert Xc[2] > 0.5
            okind.append(0); opoint.append(p); oframe.append(int(j)); oxy.append(Xc / np.linalg.norm(Xc) if sphere else Xc[:2] / Xc[2])
    host = np.zeros(n_idp, np.int32)
    anchor = np.zeros((n_idp, 3))
    rho = np.zeros(n_idp)
    for p in range(n_idp):
        X = sample_point()
        frames = rng.choice(n_frames, size=min(obs_per_point, n_frames), replace=False)
        host[p] = int(frames[0])
        Xh = cam_coords(truth[host[p]], X)
        anchor[p] = Xh / np.linalg.norm(Xh) if sphere else [Xh[0] / Xh[2], Xh[1] / Xh[2], 1.0]
        rho[p] = 1.0 / (np.linalg.norm(Xh) if sphere else Xh[2])
        for j in (frames if observe_host else frames[1:]):
            Xc = cam_coords(truth[j], X)
            okind.append(1); opoint.append(p); oframe.append(int(j)); oxy.append(Xc / np.linalg.norm(Xc) if sphere else Xc[:2] / Xc[2])
    oxy = np.array(oxy)
    if noise > 0:
        oxy = oxy + rng.normal(size=oxy.shape) * noise
    if outliers > 0:
        bad = rng.random(len(oxy)) < outliers
        oxy[bad] += rng.normal(size=(int(bad.sum()), oxy.shape[1])) * 0.2
    if sphere:
        oxy /= np.linalg.norm(oxy, axis=1, keepdims=True)
        problem["projection"] = "sphere"
    info = None
    if with_info:
        a = rng.normal(size=(len(oxy), 2, 2)) * 0.2
        info = (np.eye(2)[None] * (1.0 + rng.random((len(oxy), 1, 1))) + a @ a.transpose(0, 2, 1)).reshape(-1, 4)
    problem["xyz"] = (xyz + rng.normal(size=xyz.shape) * point_perturb, np.ones(n_xyz, np.uint8))
    problem["idp"] = (host, anchor, rho * np.exp(rng.normal(size=n_idp) * point_perturb), np.ones(n_idp, np.uint8))
    problem["obs"] = (np.array(okind, np.int32), np.array(opoint, np.int32), np.array(oframe, np.int32), oxy, info)
    problem["truth_xyz"], problem["truth_rho"] = xyz, rho
    dof = np.full(n_frames, 127 if kind == "sim3" else 63, np.int32)
    dof[0] = 0  # gauge: the first keyframe is fixed ...
    if n_frames > 1 and not pose_edges:
        dof[1] &= ~1  # ... and one translation component of the second (the scale of a monocular reconstruction)
    return truth, start, dof, problem


def opencv_project(cam, x, y):
    """GSLAM's OpenCV camera model on normalised coordinates (GSLAM/core/Camera.h:386-407; pinhole = zero distortion):
    cam = (fx, fy, cx, cy, k1, k2, p1, p2, k3) -> pixel (U, V).  numpy restatement used by the generators and tests."""
    fx, fy, cx, cy, k1, k2, p1, p2, k3 = cam
    r2 = x * x + y * y
    rad = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3))
    xd = x * rad + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
    yd = y * rad + p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y
    return cx + fx * xd, cy + fy * yd


def with_camera(problem, cam_true, cam_start, free_mask, pixel_noise=0.0, seed=0):
    """BundleGraph::camera + cameraDOF (GSLAM/core/Optimizer.h:86-100,169-171): the normalised observations of a
    make_landmark_graph problem become PIXELS of `cam_true`; the solve starts from `cam_start` and estimates the parameters
    whose bit is set in free_mask (bit i = parameter i of fx fy cx cy k1 k2 p1 p2 k3).  Information blocks (if any) are kept
    as they are (now in pixel units)."""
    kind, point, frame, xy, info = problem["obs"]
    U, V = opencv_project(np.asarray(cam_true, float), xy[:, 0], xy[:, 1])
    px = np.stack([U, V], axis=1)
    if pixel_noise > 0:
        px = px + np.random.default_rng(seed).normal(size=px.shape) * pixel_noise
    out = dict(problem)
    out["obs"] = (kind, point, frame, px, info)
    out["intrinsics"] = (np.asarray(cam_start, float).copy(), int(free_mask))
    return out
