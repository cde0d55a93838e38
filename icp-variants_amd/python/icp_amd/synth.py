"""Seeded synthetic workloads shaped like the reference's datasets.

The real ETH Apartment scans and TUM freiburg1_xyz frames are not available offline, so bench.py and
the tests use a ray-cast box-room scene:
  * laser_scan / eth_like_pair : tilting-laser pattern 344 tilt steps x 1077 beams = 370 488 points
    (ETH "Apartment" scan size, reference README.md:45-46), clouds in the global frame, analytic plane
    normals flipped toward the sensor (stand-in for PCL k=5 normals, PointCloud.h:44-55), initial
    perturbation scaled by 0.1 exactly as main.cpp:420-429 scales the benchmark pose.
  * depth_frame / rgbd_pair : organised pinhole depth images with MINF holes and uint8 RGBA albedo,
    depth quantised to 1/5000 m like VirtualSensor.h:119-124 (TUM RGB-D format).
Everything is numpy float64 ray math rounded once to float32; generators are np.random.Generator
(MT19937) with explicit seeds, so tests and bench see identical inputs on every box.
"""
import numpy as np

MINF = np.float32(-np.inf)

ROOM_MIN = np.array([-3.0, -4.0, 0.0])
ROOM_MAX = np.array([3.0, 4.0, 2.6])
# six axis-aligned boxes (min, max) standing in the room
BOXES = np.array([
    [[-2.6, -3.6, 0.0], [-1.6, -2.2, 0.9]],
    [[1.2, -3.8, 0.0], [2.8, -3.0, 2.0]],
    [[2.2, 0.5, 0.0], [2.9, 2.5, 0.75]],
    [[-2.9, 1.0, 0.0], [-2.3, 3.0, 1.8]],
    [[-0.6, 2.9, 0.0], [0.9, 3.7, 0.45]],
    [[-0.4, -0.9, 0.0], [0.5, -0.1, 0.72]],
])


def _rng(seed):
    return np.random.Generator(np.random.MT19937(int(seed)))


def rot_xyz(ax, ay, az):
    """R = Rx(ax) * Ry(ay) * Rz(az) -- the composition main.cpp:422-425 uses."""
    ca, sa, cb, sb, cg, sg = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    Rx = np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]])
    Ry = np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
    Rz = np.array([[cg, -sg, 0], [sg, cg, 0], [0, 0, 1]])
    return Rx @ Ry @ Rz


def make_pose(angles, t):
    T = np.eye(4)
    T[:3, :3] = rot_xyz(*angles)
    T[:3, 3] = t
    return T


def raycast(origin, dirs):
    """Cast rays origin + t*dirs into the scene. Returns (t, normal, surface_id); normal faces the ray."""
    o = np.asarray(origin, np.float64)
    d = np.asarray(dirs, np.float64)
    n = len(d)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        # room: we are inside -> exit parameter of the slab intersection
        t1 = (ROOM_MIN - o) * inv
        t2 = (ROOM_MAX - o) * inv
        tfar = np.where(np.isnan(np.maximum(t1, t2)), np.inf, np.maximum(t1, t2))
        axis = np.argmin(tfar, axis=1)
        t_hit = tfar[np.arange(n), axis]
        normal = np.zeros((n, 3))
        normal[np.arange(n), axis] = -np.sign(d[np.arange(n), axis])
        sid = axis.astype(np.int32)                       # 0,1,2 room faces by axis
        for b, (bmin, bmax) in enumerate(BOXES):
            a1 = (bmin - o) * inv
            a2 = (bmax - o) * inv
            lo = np.minimum(a1, a2)
            hi = np.maximum(a1, a2)
            lo = np.where(np.isnan(lo), -np.inf, lo)
            hi = np.where(np.isnan(hi), np.inf, hi)
            tn = lo.max(axis=1)
            tf = hi.min(axis=1)
            hit = (tn <= tf) & (tn > 1e-9) & (tn < t_hit)
            ax = np.argmax(lo, axis=1)
            idx = np.nonzero(hit)[0]
            t_hit[idx] = tn[idx]
            normal[idx] = 0.0
            normal[idx, ax[idx]] = -np.sign(d[idx, ax[idx]])
            sid[idx] = 3 + b
    return t_hit, normal, sid


def albedo(points, sid):
    """Deterministic uint8 RGBA texture: per-surface base colour x checker x smooth gradient."""
    p = np.asarray(points, np.float64)
    base = np.array([[200, 180, 160], [170, 200, 210], [150, 150, 150], [220, 90, 70], [80, 160, 220], [90, 200, 110],
                     [230, 210, 80], [180, 100, 200], [120, 120, 220]], np.float64)
    c = base[np.clip(sid, 0, len(base) - 1)]
    chk = ((np.floor(p[:, 0] * 2.5) + np.floor(p[:, 1] * 2.5) + np.floor(p[:, 2] * 2.5)) % 2) * 0.35 + 0.65
    grad = 0.75 + 0.25 * np.sin(p[:, 0] * 1.7 + p[:, 1] * 0.9 + p[:, 2] * 2.3)
    rgb = np.clip(c * (chk * grad)[:, None], 0, 255).astype(np.uint8)
    out = np.empty((len(p), 4), np.uint8)
    out[:, :3] = rgb
    out[:, 3] = 255
    return out


def laser_scan(sensor_pose, seed, n_tilt=344, n_beam=1077, sigma=0.01):
    """One tilting-laser scan (270 deg fan x 180 deg tilt). Returns points, normals, rgba in the GLOBAL frame."""
    rng = _rng(seed)
    fan = np.deg2rad(np.linspace(-135.0, 135.0, n_beam))
    tilt = np.deg2rad(np.linspace(-90.0, 90.0, n_tilt, endpoint=False))
    cf, sf = np.cos(fan), np.sin(fan)
    dirs = np.empty((n_tilt, n_beam, 3))
    for i, a in enumerate(tilt):                     # fan in the sensor x-y plane, tilted about the x axis
        ca, sa = np.cos(a), np.sin(a)
        dirs[i, :, 0] = cf
        dirs[i, :, 1] = sf * ca
        dirs[i, :, 2] = sf * sa
    dirs = dirs.reshape(-1, 3)
    R, t = sensor_pose[:3, :3], sensor_pose[:3, 3]
    dw = dirs @ R.T
    th, nrm, sid = raycast(t, dw)
    th = th + rng.normal(0.0, sigma, size=th.shape)
    pts = t[None, :] + dw * th[:, None]
    return pts.astype(np.float32), nrm.astype(np.float32), albedo(pts, sid)


def scan_pose(k, seed=0xE7A0):
    """Pose of scan k: chained small motions (|t| ~ 0.4 m, yaw <= 10 deg) from a fixed start."""
    T = make_pose((0.0, 0.0, 0.3), (-0.5, -1.0, 1.1))
    for j in range(1, k + 1):
        r = _rng(seed + 7919 * j)
        yaw = np.deg2rad(r.uniform(-10, 10))
        ang = r.uniform(0, 2 * np.pi)
        step = make_pose((np.deg2rad(r.uniform(-1, 1)), np.deg2rad(r.uniform(-1, 1)), yaw),
                         (0.4 * np.cos(ang), 0.4 * np.sin(ang), r.uniform(-0.03, 0.03)))
        T = T @ step
        T[:3, 3] = np.clip(T[:3, 3], ROOM_MIN + 0.8, ROOM_MAX - 0.8)
    return T


def perturbation(seed, scale=0.1):
    """Benchmark-style initial perturbation, scaled like main.cpp:420-429 (angles and translation x 0.1)."""
    r = _rng(seed)
    ang = np.deg2rad(r.uniform(-30, 30, size=3)) * scale
    tr = r.uniform(-1.0, 1.0, size=3) * scale
    return make_pose(ang, tr)


def apply_pose(T, pts, nrm=None):
    """PointCloud::change_pose (PointCloud.h:277-282) in float64, rounded once."""
    p = (np.asarray(pts, np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    if nrm is None:
        return p
    return p, (np.asarray(nrm, np.float64) @ T[:3, :3].T).astype(np.float32)


def eth_like_pair(k=0, n_tilt=344, n_beam=1077, seed=0xE7A0, sigma=0.01):
    """Config 2: target = scan k, source = scan k+1 moved by the scaled perturbation. ICP must undo it.

    Returns dict(src_pts, src_nrm, src_rgba, tgt_pts, tgt_nrm, tgt_rgba, perturb (4x4), gt = perturb^-1).
    """
    tp, tn, tc = laser_scan(scan_pose(k, seed), seed + k, n_tilt, n_beam, sigma)
    sp, sn, sc = laser_scan(scan_pose(k + 1, seed), seed + k + 1, n_tilt, n_beam, sigma)
    Tp = perturbation(seed + 100003 * (k + 1))
    sp2, sn2 = apply_pose(Tp, sp, sn)
    return dict(src_pts=sp2, src_nrm=sn2, src_rgba=sc, tgt_pts=tp, tgt_nrm=tn, tgt_rgba=tc,
                perturb=Tp, gt=np.linalg.inv(Tp), src_unperturbed=sp)


def depth_frame(cam_pose, K, width, height, seed, hole_frac=0.05, quantize=True, sigma=0.0):
    """Organised pinhole depth image of the scene, in the CAMERA frame (row-major, width*height points).

    Holes are MINF points and normals (PointCloud.h:104-106); depth is quantised to 1/5000 m
    (VirtualSensor.h:119-124).  Camera looks along +z, x right, y down.
    """
    rng = _rng(seed)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    u, v = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    dc = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], axis=-1).reshape(-1, 3)   # z = 1 rays
    R, t = cam_pose[:3, :3], cam_pose[:3, 3]
    dw = dc @ R.T
    th, nrm_w, sid = raycast(t, dw)          # th is depth along z because |dc_z| = 1
    depth = th + (rng.normal(0.0, sigma, size=th.shape) if sigma > 0 else 0.0)
    if quantize:
        depth = np.round(depth * 5000.0) / 5000.0
    pts_c = (dc * depth[:, None]).astype(np.float32)
    nrm_c = (nrm_w @ R).astype(np.float32)    # world -> camera: R^T n
    rgba = albedo(t[None, :] + dw * th[:, None], sid)
    holes = rng.random(len(depth)) < hole_frac
    pts_c[holes] = MINF
    nrm_c[holes] = MINF
    return pts_c, nrm_c, rgba


def camera_pose(k, seed=0x7A11):
    """Hand-held camera trajectory: looks roughly along +y of the room, small motion between frames."""
    base = np.eye(4)
    base[:3, :3] = np.array([[1, 0, 0], [0, 0, 1], [0, -1, 0]], float)   # cam z -> world +y, cam y -> world -z
    base[:3, 3] = [-0.3, -2.8, 1.3]
    T = base
    for j in range(1, k + 1):
        r = _rng(seed + 104729 * j)
        step = make_pose(np.deg2rad(r.uniform(-1.5, 1.5, size=3)), r.uniform(-0.04, 0.04, size=3))
        T = T @ step
    return T


def rgbd_pair(k=0, width=640, height=480, K=None, seed=0x7A11, hole_frac=0.05):
    """Configs 3/5: target = frame k (organised, camera-k frame); source = frame k+1 in ITS camera frame.

    gt maps source-camera coordinates into target-camera coordinates (what ICP should recover from identity).
    """
    if K is None:
        K = np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]])   # VirtualSensor.h:44-46
    K = np.asarray(K, np.float64)
    Tk, Tk1 = camera_pose(k, seed), camera_pose(k + 1, seed)
    tp, tn, tc = depth_frame(Tk, K, width, height, seed + 2 * k, hole_frac)
    sp, sn, sc = depth_frame(Tk1, K, width, height, seed + 2 * k + 1, hole_frac)
    gt = np.linalg.inv(Tk) @ Tk1
    return dict(src_pts=sp, src_nrm=sn, src_rgba=sc, tgt_pts=tp, tgt_nrm=tn, tgt_rgba=tc, gt=gt,
                K=K.astype(np.float32), width=width, height=height)


def compact_valid(pts, nrm, rgba):
    """keepOriginalSize=false filtering of PointCloud(depth...) (PointCloud.h:148-163): drop invalid pixels."""
    ok = np.isfinite(pts).all(axis=1) & np.isfinite(nrm).all(axis=1)
    return pts[ok], nrm[ok], rgba[ok]
