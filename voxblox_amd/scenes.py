"""Deterministic synthetic depth-camera inputs for the BASELINE.json configs.

Spec: SURVEY.md §8(d).  Pinhole camera, optical axis = camera +z, pixel ray
dir_C = normalize((u+0.5-cx)/f, (v+0.5-cy)/f, 1) (config 3 uses Kinect intrinsics and
(u-cx)/f), points emitted row-major (v outer, u inner), colours
(u mod 256, v mod 256, 40*surface_id, 255).  Analytic intersections follow the formulas
of the reference's simulation objects (simulation/objects.h: Plane :229-253, Cube
:144-201, Sphere :65-98, Cylinder :300-395) restated for numpy; everything is computed
in float64 and rounded once to float32.  No files, no RNG except the config-3 pixel
dropout (numpy MT19937 seed 0).
"""
import numpy as np

W, H = 640, 480


def _quat_yaw_y(theta):
    """Rotation by theta about world y as (w, x, y, z)."""
    return np.array([np.cos(theta / 2.0), 0.0, np.sin(theta / 2.0), 0.0], np.float64)


def quat_to_R(q):
    w, x, y, z = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _quat_from_R(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s]
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s]
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s]
    q = np.array(q)
    return q / np.linalg.norm(q)


def pixel_dirs(f=320.0, cx=None, cy=None, half_pixel=True, width=W, height=H):
    """Unit ray directions in the camera frame, shape (H*W, 3), row-major; plus (u, v)."""
    u = np.arange(width, dtype=np.float64)
    v = np.arange(height, dtype=np.float64)
    if cx is None:
        cx = width / 2.0
    if cy is None:
        cy = height / 2.0
    off = 0.5 if half_pixel else 0.0
    uu, vv = np.meshgrid(u, v)  # (H, W): v outer, u inner
    d = np.stack([(uu + off - cx) / f, (vv + off - cy) / f, np.ones_like(uu)], -1).reshape(-1, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return d, uu.reshape(-1).astype(np.int64), vv.reshape(-1).astype(np.int64)


def _colors(u, v, surface_id):
    c = np.empty((u.shape[0], 4), np.uint8)
    c[:, 0] = u % 256
    c[:, 1] = v % 256
    c[:, 2] = (40 * surface_id) % 256
    c[:, 3] = 255
    return c


def _box_interior_exit(o, d, lo, hi):
    """Distance along unit rays d from interior point o to the box [lo,hi]; face id 0..5."""
    with np.errstate(divide="ignore", invalid="ignore"):
        t_hi = np.where(d > 0, (hi - o) / d, np.inf)
        t_lo = np.where(d < 0, (lo - o) / d, np.inf)
    t_axis = np.minimum(t_hi, t_lo)
    axis = np.argmin(t_axis, axis=1)
    t = t_axis[np.arange(d.shape[0]), axis]
    face = 2 * axis + (d[np.arange(d.shape[0]), axis] > 0)
    return t, face


def _sphere_hit(o, d, c, r):
    oc = o - c
    b = (d * oc).sum(1)
    disc = b * b - ((oc * oc).sum() - r * r)
    t = np.where(disc >= 0, -b - np.sqrt(np.maximum(disc, 0)), np.inf)
    return np.where(t > 0, t, np.inf)


def _cylinder_y_hit(o, d, c, r, h):
    """Finite cylinder with axis along world y, centre c, radius r, height h (side + caps)."""
    ox, oz = o[0] - c[0], o[2] - c[2]
    a = d[:, 0] ** 2 + d[:, 2] ** 2
    b = d[:, 0] * ox + d[:, 2] * oz
    cc = ox * ox + oz * oz - r * r
    disc = b * b - a * cc
    with np.errstate(divide="ignore", invalid="ignore"):
        ts = np.where((disc >= 0) & (a > 0), (-b - np.sqrt(np.maximum(disc, 0))) / a, np.inf)
    y = o[1] + ts * d[:, 1]
    ts = np.where((ts > 0) & (np.abs(y - c[1]) <= h / 2), ts, np.inf)
    best = ts
    for ycap in (c[1] - h / 2, c[1] + h / 2):
        with np.errstate(divide="ignore", invalid="ignore"):
            tc = np.where(d[:, 1] != 0, (ycap - o[1]) / d[:, 1], np.inf)
        px = o[0] + tc * d[:, 0] - c[0]
        pz = o[2] + tc * d[:, 2] - c[2]
        tc = np.where((tc > 0) & (px * px + pz * pz <= r * r), tc, np.inf)
        best = np.minimum(best, tc)
    return best


def plane_frame(voxel_size=0.10):
    """BASELINE config 1: a wall at z_C = 3 m, identity pose, f = 320."""
    d, u, v = pixel_dirs(320.0)
    t = 3.0 / d[:, 2]
    pts = (d * t[:, None]).astype(np.float32)
    pose = (np.zeros(3, np.float32), np.array([1, 0, 0, 0], np.float32))
    return pose, pts, _colors(u, v, 1)


ROOM_LO = np.array([-3.0, -2.0, -3.5])
ROOM_HI = np.array([3.0, 2.0, 3.5])


def room_pose(k, n_frames=100):
    th = 2.0 * np.pi * k / n_frames
    pos = np.array([0.3 + 0.5 * np.sin(th), -0.2, -1.0 + 0.5 * np.cos(th)])
    return pos, _quat_yaw_y(th)


def room_frame(k, n_frames=100, f=320.0, width=W, height=H):
    """BASELINE config 2/4: box room sweep; ~28 % of the returns exceed 5 m."""
    pos, q = room_pose(k, n_frames)
    d, u, v = pixel_dirs(f, width=width, height=height, cx=width / 2.0, cy=height / 2.0)
    dG = d @ quat_to_R(q).T
    t, face = _box_interior_exit(pos, dG, ROOM_LO, ROOM_HI)
    pts = (d * t[:, None]).astype(np.float32)
    return (pos.astype(np.float32), q.astype(np.float32)), pts, _colors(u, v, face + 1)


def room_sensor_frame(sensor, step, n_steps=25, f=320.0, width=W, height=H):
    """BASELINE config 5: four cameras at yaw 0/90/180/270 deg from (+-0.5, -0.2, +-0.5)."""
    th = np.pi / 2 * sensor + 2.0 * np.pi * step / (4.0 * n_steps)
    base = [(0.5, 0.5), (0.5, -0.5), (-0.5, -0.5), (-0.5, 0.5)][sensor % 4]
    pos = np.array([base[0], -0.2, base[1]])
    q = _quat_yaw_y(th)
    d, u, v = pixel_dirs(f, width=width, height=height, cx=width / 2.0, cy=height / 2.0)
    dG = d @ quat_to_R(q).T
    t, face = _box_interior_exit(pos, dG, ROOM_LO, ROOM_HI)
    pts = (d * t[:, None]).astype(np.float32)
    return (pos.astype(np.float32), q.astype(np.float32)), pts, _colors(u, v, face + 1)


def cow_and_lady_like_frame(k, n_frames=200, dropout=0.10):
    """BASELINE config 3: room + sphere + cylinder, Kinect intrinsics, 2 m orbit, 10 % dropout."""
    th = 2.0 * np.pi * k / n_frames
    target = np.array([0.5, -0.2, 0.9])
    pos = target + 2.0 * np.array([np.sin(th), 0.0, -np.cos(th)])
    pos = np.clip(pos, ROOM_LO + 0.3, ROOM_HI - 0.3)
    zc = target - pos
    zc /= np.linalg.norm(zc)
    yc = np.array([0.0, 1.0, 0.0])
    xc = np.cross(yc, zc)
    xc /= np.linalg.norm(xc)
    yc = np.cross(zc, xc)
    R = np.stack([xc, yc, zc], 1)
    q = _quat_from_R(R)
    d, u, v = pixel_dirs(525.0, 319.5, 239.5, half_pixel=False)
    dG = d @ quat_to_R(q).T
    t_room, face = _box_interior_exit(pos, dG, ROOM_LO, ROOM_HI)
    t_sph = _sphere_hit(pos, dG, np.array([0.0, -0.5, 1.0]), 0.5)
    t_cyl = _cylinder_y_hit(pos, dG, np.array([1.0, 0.0, 0.8]), 0.25, 1.7)
    t = np.minimum(np.minimum(t_room, t_sph), t_cyl)
    sid = np.where(t == t_sph, 7, np.where(t == t_cyl, 8, face + 1))
    pts = (d * t[:, None]).astype(np.float32)
    col = _colors(u, v, sid)
    keep = np.random.RandomState(k).random_sample(pts.shape[0]) >= dropout
    return (pos.astype(np.float32), q.astype(np.float32)), pts[keep], col[keep]
