"""Seeded synthetic Fast-YCB / HO-3D shaped streams.

Produces, per object, exactly what the reference's Dataset* sources hand to the tracker
(App. B of SURVEY.md): depth frames (f32 metres, H x W), forward optical flow frames
(CV_32FC2 grid 1 or CV_16SC2 S10.5 grid 4), segmentation masks and 6D pose measurements delivered
on the reference's 5 fps / 6-frame-delay schedule
(src/roft-lib/src/DatasetImageSegmentationDelayed.cpp:42-63), plus ground truth.

Objects are boxes (ray/box intersection is analytic, so frames are generated with a handful of
vectorised torch ops on CPU or GPU); the matching render mesh is a subdivided box of ~8k
vertices / ~16k triangles like the reference's YCB meshes (src/roft-lib/meshes/DOPE/*.obj).
torch is used here only as an array library.
"""
import math
from dataclasses import dataclass, field

import numpy as np
import torch

FLOW_S16C2 = 11  # OpenCV type codes stored in the reference's .float flow files
FLOW_F32C2 = 13

CRACKER_BOX_HALF_EXTENTS = (0.082, 0.1065, 0.036)  # 003_cracker_box.obj extents / 2
# extents (m) in the spirit of the five Fast-YCB objects of the reference (cracker box, sugar box, mustard bottle,
# tomato soup can, potted meat can; src/roft-lib/meshes/DOPE/*.obj); boxes stand in for the meshes (BASELINE config #3)
FAST_YCB_HALF_EXTENTS = [CRACKER_BOX_HALF_EXTENTS, (0.046, 0.088, 0.019), (0.048, 0.096, 0.033), (0.034, 0.051, 0.034),
                         (0.051, 0.042, 0.029)]


@dataclass
class Camera:
    width: int
    height: int
    fx: float
    fy: float
    cx: float
    cy: float

    @staticmethod
    def shape_a():  # 640x480, config/config_ho3d.cfg
        return Camera(640, 480, 614.7142806307731, 614.7142806307731, 320.0, 240.0)

    @staticmethod
    def shape_b():  # 1280x720, config/config_fast_ycb.cfg:5-10
        return Camera(1280, 720, 1229.4285612615463, 1229.4285612615463, 640.0, 360.0)

    def scaled(self, s):
        return Camera(self.width // s, self.height // s, self.fx / s, self.fy / s, self.cx / s, self.cy / s)


def box_mesh(half_extents, n=36):
    """Subdivided box: 6 faces x (n+1)^2 vertices, 12 n^2 triangles (n=36 -> 8214 / 15552)."""
    hx, hy, hz = half_extents
    verts, tris = [], []
    lin = np.linspace(-1.0, 1.0, n + 1)
    a, b = np.meshgrid(lin, lin, indexing="ij")
    a, b = a.ravel(), b.ravel()
    idx = np.arange((n + 1) * (n + 1)).reshape(n + 1, n + 1)
    q = np.stack([idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()], 1)
    face_tris = np.concatenate([q[:, [0, 1, 2]], q[:, [0, 2, 3]]], 0)
    for axis in range(3):
        for sgn in (-1.0, 1.0):
            p = np.zeros((a.size, 3))
            p[:, axis] = sgn
            p[:, (axis + 1) % 3] = a
            p[:, (axis + 2) % 3] = b
            base = sum(v.shape[0] for v in verts)
            verts.append(p * np.array([hx, hy, hz]))
            tris.append(face_tris + base)
    return np.concatenate(verts).astype(np.float32), np.concatenate(tris).astype(np.int32)


def quat_mul(a, b):
    w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3]
    x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2]
    y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1]
    z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]
    return np.array([w, x, y, z])


def quat_exp(r):
    n = np.linalg.norm(r)
    if n == 0.0:
        return np.array([1.0, 0.0, 0.0, 0.0])
    return np.concatenate([[math.cos(n / 2)], math.sin(n / 2) * r / n])


def quat_to_rot(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


@dataclass
class Trajectory:
    """Ground truth per frame: position x, quaternion q (w,x,y,z), twist [v_O, w] with v_O the
    velocity of the object point instantaneously at the camera origin (the state of the
    reference's velocity filter, cf. CartesianQuaternionMeasurement.cpp:410)."""
    x: np.ndarray
    q: np.ndarray
    twist: np.ndarray


def loop_index(k, period):
    """Image index of frame k of a stream whose images repeat with `period` (make_stream(..., period=P) holds frames
    0 .. P, frame P showing the pose of frame 0 again): k for k <= P, then 1 .. P over and over."""
    return k if k <= period else ((k - 1) % period) + 1


def make_periodic_trajectory(seed, period, dt=1.0 / 30.0, speed=1.0):
    """period + 1 poses of a closed motion (pose[period] == pose[0]): sinusoids whose frequencies are multiples of
    1 / (period dt) for the position, a product of two oscillating rotations for the orientation.  A stream built on it
    can be tracked for any number of frames by cycling through its images (loop_index)."""
    rng = np.random.default_rng(seed)
    Tp = period * dt
    x0 = np.array([rng.uniform(-0.12, 0.12), rng.uniform(-0.08, 0.08), rng.uniform(0.6, 0.85)])
    A = rng.uniform(0.03, 0.08, 3) * speed
    m = rng.integers(1, 3, 3)
    ph = rng.uniform(0, 2 * math.pi, 3)
    ax1 = rng.normal(size=3)
    ax1 /= np.linalg.norm(ax1)
    ax2 = rng.normal(size=3)
    ax2 /= np.linalg.norm(ax2)
    a1, a2 = rng.uniform(0.3, 0.6) * speed, rng.uniform(0.2, 0.4) * speed
    ph2 = rng.uniform(0, 2 * math.pi)
    q0 = quat_exp(rng.normal(0, 0.6, 3))

    def pose(t):
        x = x0 + A * np.sin(2 * math.pi * m * t / Tp + ph)
        q = quat_mul(quat_exp(a1 * math.sin(2 * math.pi * t / Tp) * ax1),
                     quat_mul(quat_exp(a2 * math.sin(4 * math.pi * t / Tp + ph2) * ax2), q0))
        return x, q / np.linalg.norm(q)

    xs, qs, tw = [], [], []
    for k in range(period + 1):
        x, q = pose((k % period) * dt)
        xn, qn = pose(((k % period) + 1e-4) * dt)
        dq = quat_mul(qn, q * np.array([1.0, -1.0, -1.0, -1.0]))
        w = 2.0 * dq[1:] / (1e-4 * dt)
        xd = (xn - x) / (1e-4 * dt)
        xs.append(x)
        qs.append(q)
        tw.append(np.concatenate([xd - np.cross(w, x), w]))
    return Trajectory(np.array(xs), np.array(qs), np.array(tw))


def make_trajectory(seed, n_frames, dt=1.0 / 30.0, speed=1.0):
    rng = np.random.default_rng(seed)
    x0 = np.array([rng.uniform(-0.12, 0.12), rng.uniform(-0.08, 0.08), rng.uniform(0.6, 0.85)])
    A = rng.uniform(0.03, 0.08, 3) * speed
    f = rng.uniform(0.2, 0.6, 3)
    ph = rng.uniform(0, 2 * math.pi, 3)
    w0 = rng.uniform(0.3, 1.2, 3) * rng.choice([-1.0, 1.0], 3) * speed
    fw = rng.uniform(0.15, 0.5, 3)
    phw = rng.uniform(0, 2 * math.pi, 3)
    q = quat_exp(rng.normal(0, 0.6, 3))

    def omega(t):
        return w0 * np.sin(2 * math.pi * fw * t + phw)

    xs, qs, tw = [], [], []
    for k in range(n_frames):
        t = k * dt
        x = x0 + A * np.sin(2 * math.pi * f * t + ph)
        xd = A * 2 * math.pi * f * np.cos(2 * math.pi * f * t + ph)
        w = omega(t)
        xs.append(x)
        qs.append(q / np.linalg.norm(q))
        tw.append(np.concatenate([xd - np.cross(w, x), w]))
        q = quat_mul(quat_exp(omega(t + dt / 2) * dt), q)
    return Trajectory(np.array(xs), np.array(qs), np.array(tw))


@dataclass
class Stream:
    camera: Camera
    flow_type: int
    flow_grid: int
    flow_scale: float
    half_extents: tuple
    depth: torch.Tensor       # [F, H, W] f32
    flow: torch.Tensor        # [F, H/g, W/g, 2] f32 | i16; frame 0 is absent (zeros, flow_valid[0]=0)
    flow_valid: np.ndarray    # [F] bool
    mask_gt: torch.Tensor     # [F, H, W] u8 {0,255}
    mask_delivery: np.ndarray  # [F] int: index into mask_gt delivered at that frame, -1 = none
    pose_valid: np.ndarray    # [F] bool
    pose_meas: np.ndarray     # [F, 7] x, q(wxyz) (delayed content)
    gt: Trajectory = None
    dt: float = 1.0 / 30.0
    mesh: tuple = field(default=None, repr=False)
    gray: torch.Tensor = None  # [F, H, W] u8 textured intensity image (only with with_gray=True)
    period: int = None         # looping stream (make_stream(..., period=P)): images 0 .. P, image P shows pose 0 again

    @property
    def n_frames(self):
        return self.depth.shape[0]

    def image(self, k):
        """Index into depth / flow / gt of the image shown at frame k of the sequence (k itself unless the stream
        loops; the schedules mask_delivery / pose_valid / pose_meas / flow_valid are indexed by k)."""
        return loop_index(k, self.period) if self.period else k


def _render_box(cam, half, x, R, device):
    """Ray/box slab intersection for a batch of poses.  x: [F,3], R: [F,3,3] (numpy).
    Returns depth Z of the box surface [F, H, W] (inf = miss)."""
    u = torch.arange(cam.width, device=device, dtype=torch.float64)
    v = torch.arange(cam.height, device=device, dtype=torch.float64)
    dx = ((u - cam.cx) / cam.fx)[None, :].expand(cam.height, cam.width)
    dy = ((v - cam.cy) / cam.fy)[:, None].expand(cam.height, cam.width)
    d = torch.stack([dx, dy, torch.ones_like(dx)], -1)            # [H,W,3] camera-frame ray, d.z = 1
    Rt = torch.as_tensor(np.ascontiguousarray(np.swapaxes(R, 1, 2)), device=device, dtype=torch.float64)  # R^T
    xt = torch.as_tensor(x, device=device, dtype=torch.float64)
    o = -torch.einsum("fij,fj->fi", Rt, xt)                        # [F,3]
    dl = torch.einsum("hwj,fij->fhwi", d, Rt)                      # R^T d
    h = torch.as_tensor(half, device=device, dtype=torch.float64)
    inv = 1.0 / torch.where(dl.abs() < 1e-12, torch.full_like(dl, 1e-12), dl)
    o = o[:, None, None, :]
    t1 = (-h - o) * inv
    t2 = (h - o) * inv
    tn = torch.minimum(t1, t2).amax(-1)
    tf = torch.maximum(t1, t2).amin(-1)
    hit = (tn <= tf) & (tn > 1e-3)
    return torch.where(hit, tn, torch.full_like(tn, float("inf")))


def _texture(P, scale):
    """Smooth procedural texture attached to 3-D surface points P [..., 3] (metres): a few sinusoids of
    wavelength ~scale, so that brightness is constant along the motion of a surface point."""
    k = 2.0 * math.pi / scale
    t = (torch.sin(k * (1.00 * P[..., 0] + 0.37 * P[..., 1] + 0.21 * P[..., 2])) +
         torch.sin(k * (-0.43 * P[..., 0] + 0.91 * P[..., 1] + 0.55 * P[..., 2]) * 1.31 + 1.0) +
         torch.sin(k * (0.29 * P[..., 0] - 0.47 * P[..., 1] + 1.07 * P[..., 2]) * 0.77 + 2.0) +
         0.6 * torch.sin(k * (0.83 * P[..., 0] + 0.59 * P[..., 1] - 0.35 * P[..., 2]) * 2.3 + 0.5))
    return 128.0 + 30.0 * t


def make_stream(seed, n_frames, camera=None, flow_type=FLOW_F32C2, half_extents=CRACKER_BOX_HALF_EXTENTS,
                device="cpu", background_z=1.5, mask_period=6, pose_period=6, depth_noise=1e-3,
                depth_dropout=0.02, flow_invalid=0.005, pose_noise_x=0.005, pose_noise_rot=math.radians(2.0),
                pose_outlier_prob=0.10, pose_drop_prob=0.03, mask_dilate=1, speed=1.0, mesh_n=36, chunk=8,
                with_gray=False, period=None, n_schedule=None):
    """period = P: a looping stream -- P + 1 image frames of a closed motion (n_frames is ignored) whose delivery
    schedules (mask_delivery, pose_valid, pose_meas) cover n_schedule frames; frame k of the sequence shows image
    loop_index(k, P), mask_delivery already holds image indices."""
    cam = camera or Camera.shape_a()
    dt = 1.0 / 30.0
    if period:
        n_frames = period + 1
        gt = make_periodic_trajectory(seed, period, dt, speed)
    else:
        gt = make_trajectory(seed, n_frames, dt, speed)
    n_sched = n_schedule if (period and n_schedule) else n_frames
    li = (lambda k: loop_index(k, period)) if period else (lambda k: k)
    rng = np.random.default_rng(seed + 7919)
    device = torch.device(device)
    g = torch.Generator(device=device).manual_seed(seed)   # seeded per (seed, device type)
    H, W = cam.height, cam.width
    grid = 4 if flow_type == FLOW_S16C2 else 1
    scale = 32.0 if flow_type == FLOW_S16C2 else 1.0
    Rall = np.stack([quat_to_rot(q) for q in gt.q])

    depth = torch.empty(n_frames, H, W, device=device, dtype=torch.float32)
    mask_gt = torch.empty(n_frames, H, W, device=device, dtype=torch.uint8)
    flow = torch.zeros(n_frames, H // grid, W // grid, 2, device=device,
                       dtype=torch.int16 if flow_type == FLOW_S16C2 else torch.float32)
    u = torch.arange(W, device=device, dtype=torch.float64)[None, None, :]
    v = torch.arange(H, device=device, dtype=torch.float64)[None, :, None]
    gray = torch.empty(n_frames, H, W, device=device, dtype=torch.uint8) if with_gray else None
    for k0 in range(0, n_frames, chunk):
        k1 = min(n_frames, k0 + chunk)
        f = k1 - k0
        zb = _render_box(cam, half_extents, gt.x[k0:k1], Rall[k0:k1], device)
        hit = torch.isfinite(zb)
        z_clean = torch.where(hit, zb, torch.full_like(zb, background_z))
        mask_gt[k0:k1] = hit.to(torch.uint8) * 255
        noise = torch.randn(f, H, W, generator=g, dtype=torch.float32, device=device) * depth_noise
        drop = torch.rand(f, H, W, generator=g, device=device) < depth_dropout
        depth[k0:k1] = torch.where(drop, torch.zeros((), device=device), z_clean.float() + noise)
        if with_gray:
            Pc = torch.stack([(u - cam.cx) / cam.fx * z_clean, (v - cam.cy) / cam.fy * z_clean, z_clean], -1)
            Rf = torch.as_tensor(Rall[k0:k1], device=device)
            xf = torch.as_tensor(gt.x[k0:k1], device=device)[:, None, None, :]
            Pobj = torch.einsum("fhwj,fji->fhwi", Pc - xf, Rf)            # R^T (P - x)
            tex = torch.where(hit, _texture(Pobj, 0.035), _texture(Pc, 0.12))
            gray[k0:k1] = tex.clamp(0, 255).round().to(torch.uint8)

        # forward flow of frame k pixels towards frame k+1 (stored as flow[k+1]): back-project with
        # the clean depth of frame k, move object points rigidly with the GT motion, re-project.
        kk = np.arange(k0, min(k1, n_frames - 1))
        if len(kk):
            zp, hp = z_clean[:len(kk)], hit[:len(kk)]
            P = torch.stack([(u - cam.cx) / cam.fx * zp, (v - cam.cy) / cam.fy * zp, zp], -1)   # [f,H,W,3]
            Rp = torch.as_tensor(Rall[kk], device=device)
            Rk = torch.as_tensor(Rall[kk + 1], device=device)
            xp = torch.as_tensor(gt.x[kk], device=device)[:, None, None, :]
            xk = torch.as_tensor(gt.x[kk + 1], device=device)[:, None, None, :]
            M = torch.einsum("fij,fkj->fik", Rk, Rp)                 # R_{k+1} R_k^T
            Pn = torch.einsum("fhwj,fij->fhwi", P - xp, M) + xk
            un = cam.fx * Pn[..., 0] / Pn[..., 2] + cam.cx
            vn = cam.fy * Pn[..., 1] / Pn[..., 2] + cam.cy
            fl = torch.stack([torch.where(hp, un - u, torch.zeros_like(un)),
                              torch.where(hp, vn - v, torch.zeros_like(vn))], -1).float()
            if flow_type == FLOW_S16C2:
                fl = fl[:, grid // 2::grid, grid // 2::grid]
                fl = torch.clamp(torch.round(fl * scale), -32768, 32767).to(torch.int16)
            else:
                bad = torch.rand(len(kk), H, W, generator=g, device=device)
                fl = torch.where((bad < flow_invalid / 2)[..., None], torch.full_like(fl, float("nan")), fl)
                fl = torch.where(((bad >= flow_invalid / 2) & (bad < flow_invalid))[..., None],
                                 torch.full_like(fl, 1e10), fl)
            flow[kk + 1] = fl
        del zb, hit, z_clean

    if mask_dilate > 0:  # crude segmentation noise: the network mask is a bit fatter than GT
        k = 2 * mask_dilate + 1
        for k0 in range(0, n_frames, 32):
            m = mask_gt[k0:k0 + 32]
            mask_gt[k0:k0 + 32] = torch.nn.functional.max_pool2d(m[:, None].float(), k, 1, mask_dilate)[:, 0].to(torch.uint8)

    # delivery schedules: frame h delivers the content of frame max(h - D, 0) iff (h - D) % D == 0
    mask_delivery = np.full(n_sched, -1, np.int64)
    pose_valid = np.zeros(n_sched, bool)
    pose_meas = np.zeros((n_sched, 7))
    pose_meas[:, 3] = 1.0
    for h in range(n_sched):
        if mask_period > 0 and (h - mask_period) % mask_period == 0:
            mask_delivery[h] = li(max(h - mask_period, 0))
        if pose_period > 0 and (h - pose_period) % pose_period == 0:
            src = li(max(h - pose_period, 0))
            if h > 0 and rng.uniform() < pose_drop_prob:
                continue  # dropped detection (all-zero row in poses.txt = invalid)
            outlier = h > 0 and rng.uniform() < pose_outlier_prob
            sx = 0.05 if outlier else pose_noise_x
            sr = math.radians(30.0) if outlier else pose_noise_rot
            xm = gt.x[src] + rng.normal(0, sx, 3)
            qm = quat_mul(quat_exp(rng.normal(0, sr / math.sqrt(3), 3)), gt.q[src])
            pose_valid[h] = True
            pose_meas[h, :3] = xm
            pose_meas[h, 3:] = qm / np.linalg.norm(qm)

    flow_valid = np.ones(n_sched, bool)
    flow_valid[0] = False
    return Stream(cam, flow_type, grid, scale, tuple(half_extents), depth, flow, flow_valid, mask_gt, mask_delivery,
                  pose_valid, pose_meas, gt, dt, box_mesh(half_extents, mesh_n), gray, period)


def initial_pose_from_stream(stream):
    """The reference initialises the filter from the first DOPE pose (test/test.sh:120-123)."""
    m = np.zeros(13)
    m[6:9] = stream.pose_meas[0, :3]
    m[9:13] = stream.pose_meas[0, 3:]
    return m
