"""Seeded synthetic bundle-adjustment scenes of the shapes BASELINE.json names.

Contract of the data follows what BundleAdjuster::adjust packs (BundleAdjuster.cpp:17-97):
  poses      (Nc,6)  angle-axis*angle, translation; world->camera p_c = R p_w + t (BundleAdjuster.h:24)
  intrinsics (Nc,6)  fx fy cx cy k1 k2                                   (Camera.h:119)
  points     (Np,3)
  observations landmark-major, integer pixel coordinates cast to double   (datatypes.h:12, BundleAdjuster.cpp:83-84)
Projection model = PinholeCamera::project / ReprojectionError (Camera.h:59-76, BundleAdjuster.h:27-58):
ADDITIVE radial term, the same scalar added to x and y.
"""
import numpy as np


def rodrigues(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    n = w / th
    K = np.array([[0, -n[2], n[1]], [n[2], 0, -n[0]], [-n[1], n[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def rot_to_angle_axis(R):
    """Eigen::AngleAxisd::fromRotationMatrix equivalent for well-conditioned rotations."""
    c = (np.trace(R) - 1) / 2
    th = np.arccos(np.clip(c, -1, 1))
    if th < 1e-12:
        return np.zeros(3)
    ax = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (2 * np.sin(th))
    return ax * th


def project(pose, intr, X):
    """Vectorised reference projection.  pose (...,6) intr (...,6) X (...,3) -> (...,2), depth."""
    w, t = pose[..., :3], pose[..., 3:]
    th2 = (w * w).sum(-1, keepdims=True)
    th = np.sqrt(np.maximum(th2, 1e-300))
    n = w / th
    c, s = np.cos(th), np.sin(th)
    p_rod = X * c + np.cross(n, X) * s + n * ((n * X).sum(-1, keepdims=True) * (1 - c))
    p_small = X + np.cross(w, X)
    p = np.where(th2 > np.finfo(np.float64).eps, p_rod, p_small) + t
    x, y = p[..., 0] / p[..., 2], p[..., 1] / p[..., 2]
    r = x * x + y * y
    d = intr[..., 4] * r + intr[..., 5] * r * r
    u = intr[..., 0] * (x + d) + intr[..., 2]
    v = intr[..., 1] * (y + d) + intr[..., 3]
    return np.stack([u, v], -1), p[..., 2]


def make_scene(n_cams, n_points, obs_per_point=10, seed=2024, noise_px=0.5, integer_obs=True,
               width=512, height=336, perturb=(0.01, 0.05, 0.05), focal_factor=1.2):
    """Cameras on a ring (radius 10, small height variation) looking at the origin; points in a
    radius-3 ball; fx = fy = 1.2*max(w,h) (SequentialReconstructor.h:261), k1 = k2 = 0.
    Returns a dict with ground truth, the perturbed initial estimate and the observations."""
    rng = np.random.default_rng(seed)
    f = focal_factor * max(width, height)
    ang = 2 * np.pi * np.arange(n_cams) / n_cams
    C = np.stack([10 * np.cos(ang), 1.5 * np.sin(3 * ang), 10 * np.sin(ang)], 1)
    poses = np.zeros((n_cams, 6))
    for i in range(n_cams):
        z = -C[i] / np.linalg.norm(C[i])                    # optical axis towards the origin
        x = np.cross([0.0, 1.0, 0.0], z); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z])                              # world -> camera
        poses[i, :3] = rot_to_angle_axis(R)
        poses[i, 3:] = -R @ C[i]
    intr = np.tile([f, f, width / 2, height / 2, 0.0, 0.0], (n_cams, 1)).astype(np.float64)
    d = rng.standard_normal((n_points, 3))
    pts = d / np.linalg.norm(d, axis=1, keepdims=True) * (3 * rng.random((n_points, 1)) ** (1 / 3))

    k = min(obs_per_point, n_cams)
    obs_cam = np.empty((n_points, k), np.int32)
    # visibility: positive depth and inside the frame; fall back to any camera if too few
    chunk = max(1, 2_000_000 // n_cams)
    for s in range(0, n_points, chunk):
        P = pts[s:s + chunk]
        uv, depth = project(poses[None, :, :], intr[None, :, :], P[:, None, :])
        vis = (depth > 0.1) & (uv[..., 0] >= 1) & (uv[..., 0] < width - 1) & (uv[..., 1] >= 1) & (uv[..., 1] < height - 1)
        score = rng.random(vis.shape) + vis * 2.0            # visible cameras first, random order
        pick = np.argsort(-score, axis=1)[:, :k]
        obs_cam[s:s + chunk] = np.sort(pick, axis=1)
    obs_pt = np.repeat(np.arange(n_points, dtype=np.int32), k)
    obs_cam = obs_cam.reshape(-1)
    uv, _ = project(poses[obs_cam], intr[obs_cam], pts[obs_pt])
    uv = uv + noise_px * rng.standard_normal(uv.shape)
    if integer_obs:
        uv = np.trunc(uv)                                    # FeatureDetector.cpp:28-29: int truncation

    p0, i0, x0 = poses.copy(), intr.copy(), pts.copy()
    p0[1:, :3] += perturb[0] * rng.standard_normal((n_cams - 1, 3))
    p0[2:, 3:] += perturb[1] * rng.standard_normal((max(0, n_cams - 2), 3))  # cam0 fixed, cam1 t fixed
    x0 += perturb[2] * rng.standard_normal(x0.shape)
    return {"poses_gt": poses, "intr_gt": intr, "points_gt": pts,
            "poses": p0, "intrinsics": i0, "points": x0,
            "obs_uv": np.ascontiguousarray(uv), "obs_cam": np.ascontiguousarray(obs_cam),
            "obs_pt": np.ascontiguousarray(obs_pt)}


def poses_to_34(poses):
    """(n,6) angle-axis * angle, translation -> (n,12) rows of [R | t] (world -> camera), the
    layout of imgIdx2camPose that the landmark validity sweep reads."""
    out = np.zeros((len(poses), 3, 4))
    for i, p in enumerate(np.asarray(poses, np.float64)):
        out[i, :, :3] = rodrigues(p[:3])
        out[i, :, 3] = p[3:]
    return out.reshape(len(poses), 12)


def make_validity_case(n_cams, n_points, obs_per_point=10, seed=7, defect_rate=0.15):
    """Input of the landmark validity sweep with every kind of defect the reference tests for:
    ragged tracks (0 .. obs_per_point observations), observations displaced around the 4-px L1
    threshold, landmarks pushed behind some of their cameras, narrow-baseline tracks (angle
    around the 1-degree threshold) and a few cameras with radial distortion."""
    sc = make_scene(n_cams, n_points, obs_per_point=obs_per_point, seed=seed)
    rng = np.random.default_rng(seed + 1000)
    k = min(obs_per_point, n_cams)
    poses, intr, pts = sc["poses_gt"].copy(), sc["intr_gt"].copy(), sc["points_gt"].copy()
    intr[rng.random(n_cams) < 0.2, 4:] = rng.normal(0, 1e-3, (1, 2))
    cam = sc["obs_cam"].reshape(n_points, k).copy()
    # narrow-baseline tracks: every observation from a run of neighbouring cameras
    narrow = rng.random(n_points) < defect_rate / 2
    start = rng.integers(0, n_cams, n_points)
    width = rng.integers(1, 4, n_points)
    cam[narrow] = (start[narrow, None] + rng.integers(0, 4, (narrow.sum(), k)) % width[narrow, None]) % n_cams
    # a few landmarks far out along a viewing ray (tiny triangulation angle) or behind cameras
    far = rng.random(n_points) < defect_rate / 3
    pts[far] *= rng.choice([40.0, 400.0, -4.0], (far.sum(), 1))
    uv, depth = project(poses[cam.reshape(-1)], intr[cam.reshape(-1)], np.repeat(pts, k, axis=0))
    uv = uv + 0.5 * rng.standard_normal(uv.shape)
    bump = rng.random(len(uv)) < defect_rate
    uv[bump] += rng.uniform(-5, 5, (bump.sum(), 2))
    xy = np.trunc(np.nan_to_num(uv, nan=0.0, posinf=1e6, neginf=-1e6)).astype(np.int64).clip(-2**31 + 1, 2**31 - 1).astype(np.int32)
    # ragged tracks
    length = np.where(rng.random(n_points) < defect_rate, rng.integers(0, 4, n_points), rng.integers(2, k + 1, n_points))
    sel = (np.arange(k)[None, :] < length[:, None]).reshape(-1)
    pt_off = np.concatenate([[0], np.cumsum(length)]).astype(np.int32)
    return {"poses34": poses_to_34(poses), "intrinsics": intr, "points": pts, "pt_off": pt_off,
            "obs_cam": np.ascontiguousarray(cam.reshape(-1)[sel].astype(np.int32)),
            "obs_xy": np.ascontiguousarray(xy[sel])}


def make_always_invalid_scene(seed=7):
    """A scene whose EVERY LM step is invalid, exactly and in any floating-point arithmetic (tests of the invalid-step counter): the last
    camera has fx = fy = 0 -- its projection is the constant (cx, cy), every Jacobian entry of its observations is exactly zero -- and one
    extra landmark is seen by that camera only, so that landmark's 3 x 3 block is exactly zero; with min_lm_diagonal = 0 the damped
    block stays zero whatever the radius, the block solve fails, and the step cannot be computed.  The other cameras and landmarks are
    an ordinary scene (the gradient is far from zero, so the loop does not end on the gradient test)."""
    sc = make_scene(6, 60, obs_per_point=3, seed=seed)
    sc = dict(sc)
    last = sc["poses"].shape[0] - 1
    intr = sc["intrinsics"].copy()
    intr[last, 0] = intr[last, 1] = 0.0
    sc["intrinsics"] = intr
    sc["points"] = np.concatenate([sc["points"], [[0.1, -0.2, 0.3]]])
    j = sc["points"].shape[0] - 1
    sc["obs_pt"] = np.concatenate([sc["obs_pt"], [j, j]]).astype(np.int32)
    sc["obs_cam"] = np.concatenate([sc["obs_cam"], [last, last]]).astype(np.int32)
    sc["obs_uv"] = np.concatenate([sc["obs_uv"], [[100.0, 90.0], [101.0, 91.0]]])
    return sc
