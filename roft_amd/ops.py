"""Operator-level host API over the C ABI: one function per reference operator, numpy in / out.

Names follow the reference's classes (hsp-iit/roft `src/roft-lib`):
  flow_measurement  <- ImageOpticalFlowMeasurement<T>::freeze
  kf_predict        <- bfl::KFPrediction(SpatialVelocityModel)
  skf_correct       <- SKFCorrection::correctStep
  mask_propagate    <- ImageSegmentationOFAidedSource<T>::map + cv::remap
  ukf_predict       <- bfl::UKFPrediction(CartesianQuaternionModel)
  ukf_correct       <- ROFT::UKFCorrection::correctStep(CartesianQuaternionMeasurement)
  render_depth      <- SICAD::superimpose(..., depth)
  depth_likelihood  <- ROFTFilter::pick_best_alternative (inner loop)
  optical_flow      <- ImageOpticalFlowNVOF::step_frame (the product contract; the algorithm is this project's own)
All of them run on the GPU; none has a CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _lib as L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def make_flow(arr, width):
    assert arr.flags["C_CONTIGUOUS"] and arr.ndim == 3 and arr.shape[2] == 2
    if arr.dtype == np.int16:
        typ, scale = L.FLOW_S16C2, 32.0
    elif arr.dtype == np.float32:
        typ, scale = L.FLOW_F32C2, 1.0
    else:
        raise TypeError("flow must be int16 (CV_16SC2) or float32 (CV_32FC2)")
    rows, cols = arr.shape[:2]
    return L.Flow(arr.ctypes.data, typ, cols, rows, width // cols, scale, 1)


def flow_measurement(cam, prev_mask, prev_depth, flow_arr, dt, radius=35.0, depth_max=2.0):
    H, W = prev_mask.shape
    cap = H * W // max(int(radius), 1) + 16
    uv = np.zeros((cap, 2), np.int32)
    y = np.zeros(2 * cap)
    Hm = np.zeros((2 * cap, 6))
    n = C.c_int(0)
    fl = make_flow(flow_arr, W)
    prev_mask = np.ascontiguousarray(prev_mask, np.uint8)
    prev_depth = np.ascontiguousarray(prev_depth, np.float32)
    L.check(L.lib().roft_flow_measurement(C.byref(cam), _p(prev_mask), _p(prev_depth), C.byref(fl), dt,
                                          np.float32(radius), depth_max, cap, _p(uv), _p(y), _p(Hm), C.byref(n)))
    n = n.value
    return n, uv[:n].copy(), y[:2 * n].copy(), Hm[:2 * n].copy()


def kf_predict(x, P, qdiag):
    x, P, qdiag = _f64(x), _f64(P), _f64(qdiag)
    xo, Po = np.zeros(6), np.zeros((6, 6))
    L.check(L.lib().roft_kf_predict(_p(x), _p(P), _p(qdiag), _p(xo), _p(Po)))
    return xo, Po


def skf_correct(x, P, y, Hm, rdiag=(1.0, 1.0), reweight=True):
    x, P, y, Hm, rd = _f64(x), _f64(P), _f64(y), _f64(Hm), _f64(rdiag)
    n = y.size // 2
    xo, Po = np.zeros(6), np.zeros((6, 6))
    st = C.c_int(0)
    L.check(L.lib().roft_skf_correct(_p(x), _p(P), n, _p(y), _p(Hm), _p(rd), int(reweight), _p(xo), _p(Po),
                                     C.byref(st)))
    return st.value, xo, Po


def skf_correct_points(cam, dt, x, P, uv, z, flow_xy, rdiag=(1.0, 1.0), reweight=True):
    """SKFCorrection::correctStep fed with the kept flow points (the engine's form: H rows rebuilt on the device)."""
    x, P, rd = _f64(x), _f64(P), _f64(rdiag)
    uv = np.ascontiguousarray(uv, np.int32)
    z = np.ascontiguousarray(z, np.float32)
    fxy = np.ascontiguousarray(flow_xy, np.float32)
    n = z.size
    xo, Po = np.zeros(6), np.zeros((6, 6))
    st = C.c_int(0)
    L.check(L.lib().roft_skf_correct_points(C.byref(cam), dt, _p(x), _p(P), n, _p(uv), _p(z), _p(fxy), _p(rd),
                                            int(reweight), _p(xo), _p(Po), C.byref(st)))
    return st.value, xo, Po


def mask_propagate(mask, flow_arrs, frames_between=6):
    mask = np.ascontiguousarray(mask, np.uint8).copy()
    H, W = mask.shape
    arr = (L.Flow * max(1, len(flow_arrs)))()
    for i, f in enumerate(flow_arrs):
        arr[i] = make_flow(f, W)
    L.check(L.lib().roft_mask_propagate(_p(mask), W, H, arr, len(flow_arrs), frames_between))
    return mask


def process_noise(psd, sig_w, T):
    Q = np.zeros((9, 9))
    L.check(L.lib().roft_pose_process_noise(_p(_f64(psd)), _p(_f64(sig_w)), T, _p(Q)))
    return Q


def ukf_predict(mean, P, Q, T, ut=(1.0, 2.0, 0.0)):
    mean, P, Q = _f64(mean), _f64(P), _f64(Q)
    mo, Po = np.zeros(13), np.zeros((12, 12))
    u = L.UT(*ut)
    L.check(L.lib().roft_ukf_predict(_p(mean), _p(P), _p(Q), T, C.byref(u), _p(mo), _p(Po)))
    return mo, Po


def ukf_correct(mean, P, mtype, meas, rdiag, ut=(1.0, 2.0, 0.0)):
    mean, P, meas, rdiag = _f64(mean), _f64(P), _f64(meas), _f64(rdiag)
    mo, Po = np.zeros(13), np.zeros((12, 12))
    u = L.UT(*ut)
    st = C.c_int(0)
    L.check(L.lib().roft_ukf_correct(_p(mean), _p(P), mtype, _p(meas), _p(rdiag), C.byref(u), _p(mo), _p(Po),
                                     C.byref(st)))
    return st.value, mo, Po


def make_mesh(verts, tris):
    verts = np.ascontiguousarray(verts, np.float32)
    tris = np.ascontiguousarray(tris, np.int32)
    m = L.Mesh(verts.ctypes.data, verts.shape[0], tris.ctypes.data, tris.shape[0])
    m._keep = (verts, tris)
    return m


def mesh_classify(verts, tris):
    """(closed, flip[n_tris]): is the mesh a closed orientable surface, which triangles are wound clockwise seen from outside
    (roft_mesh_classify: host code, the classification roft_object_add applies; rules in oracle/ro_meshclass.c)."""
    m = make_mesh(verts, tris)
    flip = np.zeros(max(m.n_tris, 1), np.uint8)
    closed = C.c_int(-1)
    L.check(L.lib().roft_mesh_classify(C.byref(m), _p(flip), C.byref(closed)))
    return bool(closed.value), flip[:m.n_tris]


def render_depth(mesh, x, q, cam, divider):
    x, q = _f64(x), _f64(q)
    tile = np.zeros((cam.height // divider, cam.width // divider), np.float32)
    L.check(L.lib().roft_render_depth(C.byref(mesh), _p(x), _p(q), C.byref(cam), divider, _p(tile)))
    return tile


def depth_likelihood(cam, depth, mask, tile, divider):
    depth = np.ascontiguousarray(depth, np.float32)
    mask = np.ascontiguousarray(mask, np.uint8)
    tile = np.ascontiguousarray(tile, np.float32)
    Lv = C.c_double(0.0)
    ns = C.c_long(0)
    L.check(L.lib().roft_depth_likelihood(C.byref(cam), _p(depth), _p(mask), _p(tile), divider, C.byref(Lv),
                                          C.byref(ns)))
    return Lv.value, ns.value


def outlier_test(cam, divider, depth, mask, mesh, x2, q2, bands=0, vertex_cache=True, window_pixels=0, tiles=True, split=None):
    """ROFTFilter::pick_best_alternative (ROFTFilter.cpp:467-621) on the engine's own kernels (features_kernel,
    outlier_fused_kernel, the deciding pose chain segment).  x2 (2, 3), q2 (2, 4): the two alternatives.
    split: the workgroups of an alternative share its triangles (True) / the rows of its window (False); None: the library's choice.
    Returns (L[2], samples[2], selected, tiles (2, H/d, W/d) or None)."""
    depth = np.ascontiguousarray(depth, np.float32)
    mask = np.ascontiguousarray(mask, np.uint8)
    x2, q2 = _f64(np.asarray(x2).reshape(6)), _f64(np.asarray(q2).reshape(8))
    Lv = np.zeros(2, np.float64)
    ns = np.zeros(2, np.int64)
    sel = C.c_int(-2)
    t = np.zeros((2, cam.height // divider, cam.width // divider), np.float32) if tiles else None
    # (split travels with the call -- roft_outlier_test_split -- not through a process-wide switch)
    L.check(L.lib().roft_outlier_test_split(C.byref(cam), divider, _p(depth), _p(mask), C.byref(mesh), _p(x2), _p(q2), bands,
                                            1 if vertex_cache else 0, window_pixels, -1 if split is None else (1 if split else 0),
                                            _p(Lv), _p(ns), C.byref(sel), _p(t) if tiles else None))
    return Lv, ns, sel.value, t


def of_params(levels=3, radius=3, iterations=3, det_min=100.0):
    p = L.OFParams()
    L.check(L.lib().roft_default_of_params(C.byref(p)))
    p.levels, p.radius, p.iterations, p.det_min = levels, radius, iterations, det_min
    return p


def optical_flow(prev_gray, cur_gray, flow_type=L.FLOW_F32C2, **kw):
    """Forward flow of `prev_gray` pixels towards `cur_gray` (u8, H x W).  CV_32FC2: float (H, W, 2); CV_16SC2: int16
    (H/4, W/4, 2) S10.5 -- the two products of ImageOpticalFlowNVOF.cpp:19-80."""
    prev = np.ascontiguousarray(prev_gray, np.uint8)
    cur = np.ascontiguousarray(cur_gray, np.uint8)
    if prev.shape != cur.shape or prev.ndim != 2:
        raise ValueError("two gray images of the same shape expected")
    H, W = prev.shape
    out = np.zeros((H, W, 2), np.float32) if flow_type == L.FLOW_F32C2 else np.zeros((H // 4, W // 4, 2), np.int16)
    p = of_params(**kw)
    L.check(L.lib().roft_optical_flow(_p(prev), _p(cur), W, H, C.byref(p), flow_type, _p(out)))
    return out


class FlowProducer:
    """Batched device-resident producer (roft_flow_producer_*): `run` takes lists of device pointers."""

    def __init__(self, width, height, max_pairs, flow_type=L.FLOW_F32C2, device=0, **kw):
        self._h = C.c_void_p()
        self.width, self.height, self.max_pairs, self.flow_type = width, height, max_pairs, flow_type
        p = of_params(**kw)
        L.check(L.lib().roft_flow_producer_create(width, height, max_pairs, C.byref(p), flow_type, device, C.byref(self._h)))

    def run(self, prev_ptrs, cur_ptrs, out_ptrs):
        n = len(prev_ptrs)
        arr = C.c_void_p * n
        L.check(L.lib().roft_flow_producer_run(self._h, arr(*prev_ptrs), arr(*cur_ptrs), arr(*out_ptrs), n))

    def sync(self):
        L.check(L.lib().roft_flow_producer_sync(self._h))

    @property
    def stream(self):
        return L.lib().roft_flow_producer_stream(self._h)

    def close(self):
        if self._h:
            L.lib().roft_flow_producer_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
