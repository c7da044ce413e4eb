"""Host-side mirror (Python) of the reference's BundleAdjuster over the C ABI.

  BundleAdjuster::adjust(features, landmarks, imgIdx2camPose, imgIdx2camIntrinsics, imgIdxOrder)
      BundleAdjuster.h:80-84 / BundleAdjuster.cpp:11-188
`solve_flat` is the flat form (what adjust packs, :17-97); `BundleAdjuster.adjust` packs and
unpacks exactly as the reference does, including its quirks (axis/(angle+1e-6) on unpack,
:161-170; integer feature coordinates, :83-84).  The C++ form of the same adapter is
reconstructor_amd/host/HipBundleAdjuster.h.  All arithmetic of the solve runs in librcn.so.
"""
import ctypes as C

import numpy as np

from . import _lib

TERMINATION = {1: "CONVERGENCE (function tolerance)", 2: "CONVERGENCE (gradient tolerance)",
               3: "CONVERGENCE (parameter tolerance)", 4: "CONVERGENCE (trust region radius)",
               5: "NO_CONVERGENCE (max iterations)", 6: "FAILURE"}


def default_options(ctx, n_cams):
    o = _lib.BaOptions()
    ctx.lib.rcn_ba_default_options(int(n_cams), C.byref(o))
    return o


def summary_dict(s):
    d = {k: getattr(s, k) for k, _ in s._fields_ if k != "cost_trace"}
    d["cost_trace"] = np.array(s.cost_trace[:min(160, s.iterations + 1)])
    return d


def solve_flat(ctx, poses, intrinsics, points, obs_uv, obs_cam, obs_pt, options=None, allow_failure=False):
    """Returns (poses, intrinsics, points, summary); inputs are copied, not modified.
    allow_failure: a solve that ends in RCN_BA_FAILURE (RCN_ERR_NUMERIC; the summary is filled) returns instead of raising."""
    poses = np.array(poses, np.float64, order="C")
    intr = np.array(intrinsics, np.float64, order="C")
    pts = np.array(points, np.float64, order="C").reshape(-1, 3)
    uv = np.ascontiguousarray(obs_uv, np.float64)
    cam = np.ascontiguousarray(obs_cam, np.int32)
    pt = np.ascontiguousarray(obs_pt, np.int32)
    pb = _lib.BaProblem(poses.shape[0], pts.shape[0], cam.shape[0], 0,
                        poses.ctypes.data, intr.ctypes.data, pts.ctypes.data if pts.size else None,
                        uv.ctypes.data if uv.size else None, cam.ctypes.data if cam.size else None,
                        pt.ctypes.data if pt.size else None)
    o = options if options is not None else default_options(ctx, poses.shape[0])
    s = _lib.BaSummary()
    rc = ctx.lib.rcn_ba_solve(ctx.h, C.byref(pb), C.byref(o), C.byref(s))
    if not (allow_failure and rc == -6 and s.termination == 6):
        ctx.check(rc)
    return poses, intr, pts, summary_dict(s)


def solve_scene(ctx, scene, options=None, allow_failure=False):
    return solve_flat(ctx, scene["poses"], scene["intrinsics"], scene["points"], scene["obs_uv"],
                      scene["obs_cam"], scene["obs_pt"], options, allow_failure)


def _rot_to_angle_axis(R):
    c = (np.trace(R) - 1.0) / 2.0
    th = np.arccos(np.clip(c, -1.0, 1.0))
    if th < 1e-12:
        return np.zeros(3)
    ax = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (2.0 * np.sin(th))
    return ax * th


def _angle_axis_to_rot(angle, axis):
    """Eigen::AngleAxisd(angle, axis).toRotationMatrix() for a possibly non-unit axis."""
    c, s = np.cos(angle), np.sin(angle)
    x, y, z = axis
    t = 1 - c
    return np.array([[t * x * x + c, t * x * y - s * z, t * x * z + s * y],
                     [t * x * y + s * z, t * y * y + c, t * y * z - s * x],
                     [t * x * z - s * y, t * y * z + s * x, t * z * z + c]])


class BundleAdjuster:
    """Same call shape as the reference class (BundleAdjuster.h:76-85)."""

    def __init__(self, ctx=None, device=0):
        self.ctx = ctx if ctx is not None else _lib.Context(device)
        self.last_summary = None

    def adjust(self, features, landmarks, img_idx2cam_pose, img_idx2cam_intrinsics, img_idx_order):
        """features[img][feat] -> (x, y) integer pixel coords (or objects with .featCoord.x/.y);
        landmarks: list of dicts/objects with x, y, z and triangulatedFeatures = [(imgIdx, featIdx)];
        img_idx2cam_pose: {img: 4x4 world->camera}; img_idx2cam_intrinsics: {img: [fx,fy,cx,cy,k1,k2]}.
        Updates landmarks, poses and intrinsics in place; returns {global img idx: local idx}."""
        order = list(img_idx_order)
        g2l = {g: l for l, g in enumerate(order)}                       # BundleAdjuster.cpp:34-63
        nc = len(order)
        poses = np.zeros((nc, 6)); intr = np.zeros((nc, 6))
        for l, g in enumerate(order):
            intr[l] = np.asarray(img_idx2cam_intrinsics[g], np.float64)[:6]
            T = np.asarray(img_idx2cam_pose[g], np.float64)
            poses[l, :3] = _rot_to_angle_axis(T[:3, :3])
            poses[l, 3:] = T[:3, 3]
        get = (lambda lm, k: lm[k]) if isinstance(landmarks[0], dict) else getattr
        pts = np.array([[get(lm, "x"), get(lm, "y"), get(lm, "z")] for lm in landmarks], np.float64)
        uv, cam, pt = [], [], []
        for j, lm in enumerate(landmarks):                              # :74-97, landmark-major
            for (img, feat) in get(lm, "triangulatedFeatures"):
                f = features[img][feat]
                xy = (f.featCoord.x, f.featCoord.y) if hasattr(f, "featCoord") else (f[0], f[1])
                uv.append((float(xy[0]), float(xy[1]))); cam.append(g2l[img]); pt.append(j)
        P, I, X, s = solve_flat(self.ctx, poses, intr, pts, np.array(uv).reshape(-1, 2),
                                np.array(cam, np.int32), np.array(pt, np.int32))
        self.last_summary = s
        for j, lm in enumerate(landmarks):                              # :150-155
            if isinstance(lm, dict):
                lm["x"], lm["y"], lm["z"] = X[j]
            else:
                lm.x, lm.y, lm.z = X[j]
        for l, g in enumerate(order):                                   # :157-185
            ang = float(np.sqrt((P[l, :3] ** 2).sum()))
            axis = P[l, :3] / (ang + 1e-6)                               # the reference's non-unit axis
            T = np.eye(4)
            T[:3, :3] = _angle_axis_to_rot(ang, axis)
            T[:3, 3] = P[l, 3:]
            img_idx2cam_pose[g] = T
            img_idx2cam_intrinsics[g] = I[l].copy()
        return g2l


def poses34_from_angle_axis(poses):
    """[R | t] rows as the reference's pipeline holds them after adjust(): the unpack of BundleAdjuster.cpp:157-185,
    non-unit axis w / (angle + 1e-6) included."""
    poses = np.asarray(poses, np.float64).reshape(-1, 6)
    out = np.zeros((len(poses), 12))
    for l, p in enumerate(poses):
        ang = float(np.sqrt((p[:3] ** 2).sum()))
        R = _angle_axis_to_rot(ang, p[:3] / (ang + 1e-6))
        out[l] = np.concatenate([R, p[3:, None]], 1).reshape(-1)
    return out


class BaSession:
    """rcn_ba_session: the incremental loop's problem resident in HBM across its N - 2 global solves
    (SequentialReconstructor.cpp:1040-1094).  Cameras / landmarks / observations are appended; solve(), validity()
    and remove_outliers() work on the device arrays."""

    def __init__(self, ctx):
        self.ctx = ctx
        h = C.c_void_p()
        ctx.check(ctx.lib.rcn_ba_session_create(ctx.h, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.rcn_ba_session_destroy(self.h)
            self.h = None

    def add_camera(self, pose6, intr6):
        p, k = np.ascontiguousarray(pose6, np.float64), np.ascontiguousarray(intr6, np.float64)
        idx = C.c_int32()
        self.ctx.check(self.ctx.lib.rcn_ba_session_add_camera(self.h, p.ctypes.data, k.ctypes.data, C.byref(idx)))
        return idx.value

    def add_points(self, xyz):
        xyz = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
        first = C.c_int32()
        self.ctx.check(self.ctx.lib.rcn_ba_session_add_points(self.h, len(xyz), xyz.ctypes.data if len(xyz) else None, C.byref(first)))
        return first.value

    def add_observations(self, pt, cam, xy):
        pt, cam = np.ascontiguousarray(pt, np.int32), np.ascontiguousarray(cam, np.int32)
        xy = np.ascontiguousarray(xy, np.int32).reshape(-1, 2)
        self.ctx.check(self.ctx.lib.rcn_ba_session_add_observations(self.h, len(pt), pt.ctypes.data if len(pt) else None,
                                                                    cam.ctypes.data if len(pt) else None, xy.ctypes.data if len(pt) else None))

    def counts(self):
        nc, npts, no = C.c_int32(), C.c_int32(), C.c_int64()
        self.ctx.check(self.ctx.lib.rcn_ba_session_counts(self.h, C.byref(nc), C.byref(npts), C.byref(no)))
        return nc.value, npts.value, no.value

    def cameras(self, poses=None, intr=None):
        """Returns (poses, intrinsics); arrays given are written into the session first."""
        nc = self.counts()[0]
        pi = None if poses is None else np.ascontiguousarray(poses, np.float64)
        ki = None if intr is None else np.ascontiguousarray(intr, np.float64)
        po, ko = np.zeros((nc, 6)), np.zeros((nc, 6))
        self.ctx.check(self.ctx.lib.rcn_ba_session_cameras(self.h, po.ctypes.data, ko.ctypes.data,
                                                           pi.ctypes.data if pi is not None else None, ki.ctypes.data if ki is not None else None))
        return po, ko

    def graph(self):
        no = self.counts()[2]
        pt, cam, xy = np.zeros(no, np.int32), np.zeros(no, np.int32), np.zeros((no, 2), np.int32)
        self.ctx.check(self.ctx.lib.rcn_ba_session_graph(self.h, pt.ctypes.data, cam.ctypes.data, xy.ctypes.data))
        return pt, cam, xy

    def points(self):
        x = np.zeros((self.counts()[1], 3))
        self.ctx.check(self.ctx.lib.rcn_ba_session_read_points(self.h, x.ctypes.data))
        return x

    def solve(self, options=None):
        s = _lib.BaSummary()
        self.ctx.check(self.ctx.lib.rcn_ba_session_solve(self.h, C.byref(options) if options is not None else None, C.byref(s)))
        return summary_dict(s)

    def validity(self, poses34=None, max_err=4.0, min_angle=1.0):
        """checkLandmarkValidity on the device arrays; returns (inlier flags, observations erased)."""
        if poses34 is None:
            poses34 = poses34_from_angle_axis(self.cameras()[0])
        p = np.ascontiguousarray(poses34, np.float64)
        inl = np.zeros(max(1, self.counts()[1]), np.uint8)
        n_in, n_er = C.c_int32(), C.c_int32()
        self.ctx.check(self.ctx.lib.rcn_ba_session_validity(self.h, p.ctypes.data, float(max_err), float(min_angle), inl.ctypes.data, C.byref(n_in), C.byref(n_er)))
        return inl[:self.counts()[1]].astype(bool), n_er.value

    def remove_outliers(self):
        npts = self.counts()[1]
        new_idx = np.zeros(max(1, npts), np.int32)
        n = C.c_int32()
        self.ctx.check(self.ctx.lib.rcn_ba_session_remove_outliers(self.h, new_idx.ctypes.data, C.byref(n)))
        return new_idx[:npts], n.value


def smoke(ctx):
    """Small solve on the GPU checked against nothing but itself converging (used by
    __graft_entry__.smoke together with the oracle comparison there)."""
    from . import synth_ba
    sc = synth_ba.make_scene(12, 300, seed=5)
    P, I, X, s = solve_scene(ctx, sc)
    assert s["final_rms_px"] < 1.0 < s["initial_rms_px"], s
    return sc, (P, I, X, s)
