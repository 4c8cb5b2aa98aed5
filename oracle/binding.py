"""ctypes binding of the CPU oracle (oracle/libroft_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py -- never by the product package `roft_amd`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libroft_oracle.so")

FLOW_S16C2 = 11
FLOW_F32C2 = 13
MEAS_NONE, MEAS_VELOCITY, MEAS_POSE, MEAS_POSE_VELOCITY = 0, 1, 2, 3


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libroft_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class Camera(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("fx", C.c_double), ("fy", C.c_double),
                ("cx", C.c_double), ("cy", C.c_double)]


class Flow(C.Structure):
    _fields_ = [("data", C.c_void_p), ("type", C.c_int), ("cols", C.c_int), ("rows", C.c_int),
                ("grid", C.c_int), ("scale", C.c_float), ("valid", C.c_int)]


class UT(C.Structure):
    _fields_ = [("alpha", C.c_double), ("beta", C.c_double), ("kappa", C.c_double)]


class Mesh(C.Structure):
    _fields_ = [("verts", C.c_void_p), ("n_verts", C.c_int), ("tris", C.c_void_p), ("n_tris", C.c_int),
                ("tri_flip", C.c_void_p), ("closed", C.c_int)]


class TrackerConfig(C.Structure):
    _fields_ = [("cam", Camera), ("sample_time", C.c_double), ("ut", UT),
                ("p_mean0", C.c_double * 13), ("p_cov0_diag", C.c_double * 12),
                ("v_mean0", C.c_double * 6), ("v_cov0_diag", C.c_double * 6),
                ("p_sigma_ang_vel", C.c_double * 3), ("p_psd_lin_acc", C.c_double * 3),
                ("v_q_diag", C.c_double * 6),
                ("p_meas_cov_v", C.c_double * 3), ("p_meas_cov_w", C.c_double * 3),
                ("p_meas_cov_x", C.c_double * 3), ("p_meas_cov_q", C.c_double * 3),
                ("v_meas_cov_flow", C.c_double * 2),
                ("depth_maximum", C.c_double), ("subsampling_radius", C.c_double),
                ("flow_weighting", C.c_int), ("use_pose", C.c_int), ("use_pose_resync", C.c_int),
                ("use_velocity", C.c_int), ("outlier_rejection", C.c_int),
                ("flow_aided_segmentation", C.c_int), ("mask_frames_between", C.c_int),
                ("pose_frames_between", C.c_int), ("stamped_masks", C.c_int)]


class Frame(C.Structure):
    _fields_ = [("dt", C.c_double), ("depth", C.c_void_p), ("flow", Flow), ("mask", C.c_void_p),
                ("pose_valid", C.c_int), ("pose_x", C.c_double * 3), ("pose_q", C.c_double * 4),
                ("stamp", C.c_double), ("mask_stamp", C.c_double)]


class FrameResult(C.Structure):
    _fields_ = [("pose", C.c_double * 13), ("pose_cov", C.c_double * 144), ("twist", C.c_double * 6),
                ("twist_cov", C.c_double * 36), ("n_flow_points", C.c_int),
                ("outlier_selected", C.c_int), ("outlier_L", C.c_double * 2),
                ("n_ukf_corrections", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        dp = C.POINTER(C.c_double)
        L.ro_flow_measurement.restype = C.c_int
        L.ro_flow_measurement.argtypes = [C.POINTER(Camera), C.c_void_p, C.c_void_p, C.POINTER(Flow),
                                          C.c_double, C.c_float, C.c_double, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_void_p]
        L.ro_kf_predict.argtypes = [C.c_void_p] * 5
        L.ro_skf_correct.restype = C.c_int
        L.ro_skf_correct.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_void_p, C.c_void_p]
        L.ro_mask_propagate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(Flow), C.c_int, C.c_int,
                                        C.c_void_p]
        L.ro_mask_binarise.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.ro_pose_process_noise.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
        L.ro_ukf_predict.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.POINTER(UT),
                                     C.c_void_p, C.c_void_p]
        L.ro_ukf_correct.restype = C.c_int
        L.ro_ukf_correct.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                     C.POINTER(UT), C.c_void_p, C.c_void_p]
        L.ro_render_depth.argtypes = [C.POINTER(Mesh), C.c_void_p, C.c_void_p, C.POINTER(Camera), C.c_int,
                                      C.c_void_p]
        L.ro_depth_likelihood.restype = C.c_double
        L.ro_depth_likelihood.argtypes = [C.POINTER(Camera), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                          C.POINTER(C.c_long)]
        L.ro_tracker_default_config.argtypes = [C.POINTER(TrackerConfig), C.c_int, C.c_int]
        L.ro_tracker_create.restype = C.c_void_p
        L.ro_tracker_create.argtypes = [C.POINTER(TrackerConfig), C.POINTER(Mesh)]
        L.ro_tracker_destroy.argtypes = [C.c_void_p]
        L.ro_tracker_step.restype = C.c_int
        L.ro_tracker_step.argtypes = [C.c_void_p, C.POINTER(Frame), C.POINTER(FrameResult)]
        L.ro_tracker_mask.restype = C.c_void_p
        L.ro_tracker_mask.argtypes = [C.c_void_p]
        L.ro_mesh_classify.restype = C.c_int
        L.ro_mesh_classify.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.ro_render_depth_mode.argtypes = [C.POINTER(Mesh), C.c_void_p, C.c_void_p, C.POINTER(Camera), C.c_int, C.c_void_p, C.c_int]
        L.ro_tracker_shadow_render.argtypes = [C.c_void_p, C.c_int]
        L.ro_tracker_shadow_L.restype = C.c_int
        L.ro_tracker_shadow_L.argtypes = [C.c_void_p, C.c_void_p]
        for name in ("ro_add", "ro_adds"):
            f = getattr(L, name)
            f.restype = C.c_double
            f.argtypes = [C.c_void_p] * 5 + [C.c_int]
        L.ro_auc.restype = C.c_double
        L.ro_auc.argtypes = [C.c_void_p, C.c_int]
        L.ro_optical_flow.restype = C.c_int
        L.ro_optical_flow.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
        L.ro_flow_to_s16_grid4.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.ro_jacobi_eig.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ro_inverse.restype = C.c_int
        L.ro_inverse.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.ro_quat_boxplus.argtypes = [C.c_void_p] * 3
        L.ro_quat_diff.argtypes = [C.c_void_p] * 3
        L.ro_quat_to_axis_angle.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def camera(width, height, fx, fy, cx, cy):
    return Camera(width, height, fx, fy, cx, cy)


def make_flow(arr, width, valid=True):
    """arr: (rows, cols, 2) float32 or int16 numpy array (kept alive by the caller)."""
    if arr is None:
        return Flow(None, FLOW_F32C2, 0, 0, 1, 1.0, 0)
    assert arr.flags["C_CONTIGUOUS"] and arr.ndim == 3 and arr.shape[2] == 2
    if arr.dtype == np.int16:
        typ, scale = FLOW_S16C2, 32.0
    else:
        assert arr.dtype == np.float32
        typ, scale = FLOW_F32C2, 1.0
    rows, cols = arr.shape[:2]
    return Flow(arr.ctypes.data, typ, cols, rows, width // cols, scale, 1 if valid else 0)


def flow_measurement(cam, prev_mask, prev_depth, flow_arr, dt, radius=35.0, depth_max=2.0):
    L = lib()
    H, W = prev_mask.shape
    cap = H * W // 2 + 16
    uv = np.zeros((cap, 2), np.int32)
    y = np.zeros(2 * cap)
    Hm = np.zeros((2 * cap, 6))
    fl = make_flow(flow_arr, W)
    prev_mask = np.ascontiguousarray(prev_mask, np.uint8)
    prev_depth = np.ascontiguousarray(prev_depth, np.float32)
    n = L.ro_flow_measurement(C.byref(cam), _p(prev_mask), _p(prev_depth), C.byref(fl), dt,
                              np.float32(radius), depth_max, cap, _p(uv), _p(y), _p(Hm))
    assert n >= 0
    return n, uv[:n].copy(), y[:2 * n].copy(), Hm[:2 * n].copy()


def kf_predict(x, P, qdiag):
    x, P, qdiag = _f64(x), _f64(P), _f64(qdiag)
    xo, Po = np.zeros(6), np.zeros((6, 6))
    lib().ro_kf_predict(_p(x), _p(P), _p(qdiag), _p(xo), _p(Po))
    return xo, Po


def skf_correct(x, P, y, Hm, rdiag=(1.0, 1.0), reweight=True):
    x, P, y, Hm, rd = _f64(x), _f64(P), _f64(y), _f64(Hm), _f64(rdiag)
    n = y.size // 2
    xo, Po = np.zeros(6), np.zeros((6, 6))
    rc = lib().ro_skf_correct(_p(x), _p(P), n, _p(y), _p(Hm), _p(rd), int(reweight), _p(xo), _p(Po))
    return rc, xo, Po


def mask_propagate(mask, flow_arrs, frames_between=6):
    mask = np.ascontiguousarray(mask, np.uint8).copy()
    H, W = mask.shape
    arr = (Flow * max(1, len(flow_arrs)))()
    for i, f in enumerate(flow_arrs):
        arr[i] = make_flow(f, W)
    scratch = np.zeros(H * W, np.int32)
    lib().ro_mask_propagate(_p(mask), W, H, arr, len(flow_arrs), frames_between, _p(scratch))
    return mask


def process_noise(psd, sig_w, T):
    Q = np.zeros((9, 9))
    lib().ro_pose_process_noise(_p(_f64(psd)), _p(_f64(sig_w)), T, _p(Q))
    return Q


def ut_weights(n, ut=(1.0, 2.0, 0.0)):
    """{c, wm_0, wc_0, w_i} of the unscented transform over an augmented dimension n (conformance kit)."""
    out = np.zeros(4)
    lib().ro_ut_weights(C.c_int(n), C.byref(UT(*ut)), _p(out))
    return out


def sigma_points(mean, P, Qn, ut=(1.0, 2.0, 0.0)):
    """The sigma set the oracle draws for Gaussian(9, 1 quaternion) + r noise dof: (13 + r) x (2 (12 + r) + 1)."""
    Qn = _f64(np.atleast_2d(Qn))
    r = Qn.shape[0]
    sp = np.zeros((13 + r, 2 * (12 + r) + 1))
    n = lib().ro_sigma_points(_p(_f64(mean)), _p(_f64(P)), _p(Qn), C.c_int(r), C.byref(UT(*ut)), _p(sp))
    assert n == sp.shape[1]
    return sp


def ukf_predict(mean, P, Q, T, ut=(1.0, 2.0, 0.0)):
    mean, P, Q = _f64(mean), _f64(P), _f64(Q)
    mo, Po = np.zeros(13), np.zeros((12, 12))
    u = UT(*ut)
    lib().ro_ukf_predict(_p(mean), _p(P), _p(Q), T, C.byref(u), _p(mo), _p(Po))
    return mo, Po


def ukf_correct(mean, P, mtype, meas, rdiag, ut=(1.0, 2.0, 0.0)):
    mean, P, meas, rdiag = _f64(mean), _f64(P), _f64(meas), _f64(rdiag)
    mo, Po = np.zeros(13), np.zeros((12, 12))
    u = UT(*ut)
    rc = lib().ro_ukf_correct(_p(mean), _p(P), mtype, _p(meas), _p(rdiag), C.byref(u), _p(mo), _p(Po))
    return rc, mo, Po


def make_mesh(verts, tris):
    verts = np.ascontiguousarray(verts, np.float32)
    tris = np.ascontiguousarray(tris, np.int32)
    flip = np.zeros(max(tris.shape[0], 1), np.uint8)
    closed = lib().ro_mesh_classify(verts.ctypes.data, verts.shape[0], tris.ctypes.data, tris.shape[0], flip.ctypes.data) if tris.shape[0] else 0
    m = Mesh(verts.ctypes.data, verts.shape[0], tris.ctypes.data, tris.shape[0], flip.ctypes.data, closed)
    m._keep = (verts, tris, flip)
    return m


def mesh_classify(verts, tris):
    """(closed, flip[n_tris]) of oracle/ro_meshclass.c: is the mesh a closed orientable surface, which triangles are wound
    clockwise seen from outside."""
    verts = np.ascontiguousarray(verts, np.float32)
    tris = np.ascontiguousarray(tris, np.int32)
    flip = np.zeros(max(tris.shape[0], 1), np.uint8)
    closed = lib().ro_mesh_classify(verts.ctypes.data, verts.shape[0], tris.ctypes.data, tris.shape[0], flip.ctypes.data)
    return bool(closed), flip[:tris.shape[0]]


def render_depth(mesh, x, q, cam, divider):
    x, q = _f64(x), _f64(q)
    tile = np.zeros((cam.height // divider, cam.width // divider), np.float32)
    lib().ro_render_depth(C.byref(mesh), _p(x), _p(q), C.byref(cam), divider, _p(tile))
    return tile


RENDER_CONTRACT, RENDER_V1, RENDER_GL = 0, 1, 2


def render_depth_mode(mesh, x, q, cam, divider, mode):
    """The rasteriser in the arithmetic of rounds 1 - 4 (RENDER_V1) or with the numerics of the reference's GL pipeline
    (RENDER_GL): only for bounding the distance between those and the contract (ro_render.c)."""
    x, q = _f64(x), _f64(q)
    tile = np.zeros((cam.height // divider, cam.width // divider), np.float32)
    lib().ro_render_depth_mode(C.byref(mesh), _p(x), _p(q), C.byref(cam), divider, _p(tile), mode)
    return tile


def depth_likelihood(cam, depth, mask, tile, divider):
    depth = np.ascontiguousarray(depth, np.float32)
    mask = np.ascontiguousarray(mask, np.uint8)
    tile = np.ascontiguousarray(tile, np.float32)
    ns = C.c_long(0)
    L = lib().ro_depth_likelihood(C.byref(cam), _p(depth), _p(mask), _p(tile), divider, C.byref(ns))
    return L, ns.value


def default_config(width, height):
    cfg = TrackerConfig()
    lib().ro_tracker_default_config(C.byref(cfg), width, height)
    return cfg


class Tracker:
    """One ROFTFilter instance (one object)."""

    def __init__(self, cfg, verts, tris):
        self.cfg = cfg
        self._mesh = make_mesh(verts, tris)
        self._h = lib().ro_tracker_create(C.byref(cfg), C.byref(self._mesh))
        self.H, self.W = cfg.cam.height, cfg.cam.width

    def step(self, dt, depth, flow_arr, mask, pose, stamp=0.0, mask_stamp=0.0):
        """pose: None or (x[3], q[4]).  stamp / mask_stamp: only with cfg.stamped_masks.  Returns a FrameResult."""
        depth = np.ascontiguousarray(depth, np.float32)
        fr = Frame()
        fr.dt = dt
        fr.stamp, fr.mask_stamp = stamp, mask_stamp
        fr.depth = depth.ctypes.data
        fr.flow = make_flow(flow_arr, self.W)
        keep = [depth, flow_arr]
        if mask is not None:
            mask = np.ascontiguousarray(mask, np.uint8)
            keep.append(mask)
            fr.mask = mask.ctypes.data
        else:
            fr.mask = None
        if pose is not None:
            fr.pose_valid = 1
            fr.pose_x = (C.c_double * 3)(*pose[0])
            fr.pose_q = (C.c_double * 4)(*pose[1])
        res = FrameResult()
        rc = lib().ro_tracker_step(self._h, C.byref(fr), C.byref(res))
        if rc != 0:
            raise RuntimeError("ro_tracker_step failed: %d" % rc)
        return res

    def shadow_render(self, on=True):
        """Every outlier test also scores its alternatives on RENDER_V1 and RENDER_GL renders (shadow_L)."""
        lib().ro_tracker_shadow_render(self._h, 1 if on else 0)

    def shadow_L(self):
        """(L_v1[2], L_gl[2]) of the last step's outlier test, or None."""
        out = np.zeros(4)
        if not lib().ro_tracker_shadow_L(self._h, _p(out)):
            return None
        return out[:2].copy(), out[2:].copy()

    def mask(self):
        ptr = lib().ro_tracker_mask(self._h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(self.H, self.W)).copy()

    def close(self):
        if self._h:
            lib().ro_tracker_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def add(R_est, t_est, R_gt, t_gt, pts):
    a = [_f64(v) for v in (R_est, t_est, R_gt, t_gt, pts)]
    return lib().ro_add(*[_p(v) for v in a], a[4].shape[0])


def adds(R_est, t_est, R_gt, t_gt, pts):
    a = [_f64(v) for v in (R_est, t_est, R_gt, t_gt, pts)]
    return lib().ro_adds(*[_p(v) for v in a], a[4].shape[0])


def auc(distances):
    d = _f64(distances)
    return lib().ro_auc(_p(d), d.size)


def optical_flow(prev, cur, levels=3, radius=3, iterations=3, det_min=100.0):
    prev = np.ascontiguousarray(prev, np.uint8)
    cur = np.ascontiguousarray(cur, np.uint8)
    H, W = prev.shape
    flow = np.zeros((H, W, 2), np.float32)
    rc = lib().ro_optical_flow(_p(prev), _p(cur), W, H, levels, radius, iterations, np.float32(det_min), _p(flow))
    assert rc == 0
    return flow


def flow_to_s16_grid4(flow):
    flow = np.ascontiguousarray(flow, np.float32)
    H, W = flow.shape[:2]
    out = np.zeros((H // 4, W // 4, 2), np.int16)
    lib().ro_flow_to_s16_grid4(_p(flow), W, H, _p(out))
    return out
