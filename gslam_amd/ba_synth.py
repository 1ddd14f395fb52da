"""Synthetic bundle-adjustment graphs: cameras on a noisy helix looking inward, points
uniform in a box, each point observed by its `n_obs_per_point` nearest cameras, pixel noise
sigma = 1 px at f = 500 (0.002 normalised), 5 % outliers (x50), initial poses perturbed 1 deg / 1 %,
points 1 %; first camera fixed.  Input generator for tests and bench.py — numpy/scipy only.

Layout mirrors GSLAM::BundleGraph (GSLAM/core/Optimizer.h:150-172): keyframes = T_wc as
[qx qy qz qw tx ty tz] + dof bitmask, mappoints = xyz, mappointObserves = (pointId, frameId,
normalised (x, y) on the z = 1 plane).
"""
import numpy as np

KF_SE3 = 63


def _quat_from_R(R):
    """Rotation matrices (N,3,3) -> quaternions (N,4) [x y z w]."""
    N = R.shape[0]
    q = np.zeros((N, 4))
    tr = R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]
    for i in range(N):
        m = R[i]
        if tr[i] > 0:
            s = np.sqrt(tr[i] + 1.0) * 2
            q[i] = [(m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s, 0.25 * s]
        elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
            s = np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
            q[i] = [0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s, (m[2, 1] - m[1, 2]) / s]
        elif m[1, 1] > m[2, 2]:
            s = np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
            q[i] = [(m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s, (m[0, 2] - m[2, 0]) / s]
        else:
            s = np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
            q[i] = [(m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s, (m[1, 0] - m[0, 1]) / s]
    return q


def quat_to_R(q):
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((q.shape[0], 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def _quat_mul(a, b):
    ax, ay, az, aw = a.T
    bx, by, bz, bw = b.T
    return np.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz], axis=1)


def make_graph(n_cams, n_points, n_obs_per_point=6, seed=1, noise=0.002, outlier_frac=0.05, perturb=True):
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(seed)
    ang = np.linspace(0, 2 * np.pi * max(1.0, n_cams / 200.0), n_cams, endpoint=False)
    radius = 10.0 + 0.3 * rng.standard_normal(n_cams)
    pos = np.stack([radius * np.cos(ang), radius * np.sin(ang),
                    np.linspace(-2, 2, n_cams) + 0.2 * rng.standard_normal(n_cams)], axis=1)
    zc = -pos / np.linalg.norm(pos, axis=1, keepdims=True)
    up = np.array([0.0, 0.0, 1.0])
    xc = np.cross(up, zc)
    xc /= np.linalg.norm(xc, axis=1, keepdims=True)
    yc = np.cross(zc, xc)
    R = np.stack([xc, yc, zc], axis=2)  # columns = camera axes in the world
    q_gt = _quat_from_R(R)
    pts_gt = rng.uniform(-3, 3, size=(n_points, 3))
    k = min(n_obs_per_point, n_cams)
    _, nn = cKDTree(pos).query(pts_gt, k=k)
    nn = nn.reshape(n_points, k)
    obs_point = np.repeat(np.arange(n_points, dtype=np.int32), k)
    obs_cam = nn.reshape(-1).astype(np.int32)
    Xc = np.einsum("nji,nj->ni", R[obs_cam], pts_gt[obs_point] - pos[obs_cam])
    assert (Xc[:, 2] > 0.5).all()
    xy = Xc[:, :2] / Xc[:, 2:3]
    sig = np.full(len(xy), noise)
    sig[rng.random(len(xy)) < outlier_frac] *= 50
    xy = xy + rng.standard_normal(xy.shape) * sig[:, None]
    poses_gt = np.concatenate([q_gt, pos], axis=1)
    poses = poses_gt.copy()
    pts = pts_gt.copy()
    if perturb:
        w = rng.standard_normal((n_cams, 3)) * np.deg2rad(1.0) / np.sqrt(3)
        th = np.linalg.norm(w, axis=1, keepdims=True)
        dq = np.concatenate([np.sin(th / 2) * w / th, np.cos(th / 2)], axis=1)
        poses[:, :4] = _quat_mul(poses[:, :4], dq)
        poses[:, 4:] += rng.standard_normal((n_cams, 3)) * 0.01 * 10.0 / np.sqrt(3)
        pts += rng.standard_normal(pts.shape) * 0.01 * 6.0 / np.sqrt(3)
        poses[0] = poses_gt[0]
    dof = np.full(n_cams, KF_SE3, np.int32)
    dof[0] = 0  # first camera fixed (UPDATE_KF_NONE)
    if n_cams > 1:
        dof[1] = KF_SE3 & ~1  # second camera: local-x translation frozen -> fixes the monocular scale gauge
    return {
        "cam_pose": np.ascontiguousarray(poses), "cam_dof": dof, "point_xyz": np.ascontiguousarray(pts),
        "obs_cam": obs_cam, "obs_point": obs_point, "obs_xy": np.ascontiguousarray(xy),
        "cam_pose_gt": poses_gt, "point_xyz_gt": pts_gt,
    }
