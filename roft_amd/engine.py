"""Batched ROFT filtering engine: host-side mirror of ROFT::ROFTFilter for many objects at once.

Thin wrapper over the C ABI (include/roft_engine.h): `ROFTFilterBatch` plays the role the
reference's `ROFTFilter` plays for one object (src/roft-lib/include/ROFT/ROFTFilter.h:38-194) --
construct with the same parameters, feed one frame at a time -- but every object of the batch
advances in the same kernel launches and all filter state stays in HBM.
"""
import ctypes as C

import numpy as np

from . import _lib as L


def default_config(width, height, flow_type=L.FLOW_F32C2, max_objects=64, device=0, max_batch_frames=1):
    cfg = L.Config()
    L.check(L.lib().roft_default_config(C.byref(cfg), width, height, flow_type))
    cfg.max_objects = max_objects
    cfg.device = device
    cfg.max_batch_frames = max_batch_frames
    return cfg


def default_object():
    o = L.ObjectDesc()
    L.check(L.lib().roft_default_object(C.byref(o)))
    return o


def aligned_batches(first, last, T, period, phase=0):
    """[first, last) in consecutive batches of at most T frames that END with a pose-arrival frame (frames f with
    f % period == phase: the delayed pose source delivers every `period` frames).  Inside the engine a pose arrival hands
    the pose chain from one belief lineage -- one lane -- to the other: with the arrival as the last frame of
    a batch, every batch is one lineage's ordinary steps followed by the other's re-sync replay, the replay of batch b
    overlaps the ordinary steps of batch b + 1 on the other lane, and no launch of a lane carries both.  Measured against
    full batches of 8 cut anywhere (A/B in one box): +4 % at 20 steps, +5 % at 60, +3.5 % at 240."""
    out = []
    k = first
    while k < last:
        end = k + ((phase - k) % period) + 1 if period <= T else k + T
        while period <= T and (end - k) % period == 0 and end + period - k <= T:   # (whole periods, as many as a batch holds)
            end += period
        end = min(end, last, k + T)
        out.append((k, end - k))
        k = end
    return out



class ROFTFilterBatch:
    def __init__(self, cfg):
        L.require_device()
        self.cfg = cfg
        self._h = C.c_void_p()
        L.check(L.lib().roft_engine_create(C.byref(cfg), C.byref(self._h)))
        self.n_objects = 0
        self._keep = []
        self._inputs = None
        self.W, self.H = cfg.cam.width, cfg.cam.height

    def add_object(self, desc, verts, tris):
        verts = np.ascontiguousarray(verts, np.float32)
        tris = np.ascontiguousarray(tris, np.int32)
        desc.mesh = L.Mesh(verts.ctypes.data, verts.shape[0], tris.ctypes.data, tris.shape[0])
        oid = C.c_int(-1)
        L.check(L.lib().roft_object_add(self._h, C.byref(desc), C.byref(oid)))
        self.n_objects += 1
        self._inputs = (L.FrameInput * self.n_objects)()
        return oid.value

    def submit(self, frames):
        """frames: one dict per object with keys depth, flow, mask, pose (None or (x, q)), dt,
        mem_kind; depth/flow/mask are numpy arrays (HOST) or integer device addresses (DEVICE)."""
        self._fill(frames)
        L.check(L.lib().roft_frame_submit(self._h, self._inputs, self.n_objects))

    def _fill(self, frames):
        assert len(frames) == self.n_objects
        keep = []
        for i, f in enumerate(frames):
            fi = self._inputs[i]
            kind = f.get("mem_kind", L.MEM_HOST)
            fi.mem_kind = kind
            fi.dt = f.get("dt", 0.0)
            fi.stamp = f.get("stamp", 0.0)            # only read with cfg.stamped_masks
            fi.mask_stamp = f.get("mask_stamp", 0.0)
            for key in ("depth", "flow", "mask"):
                v = f.get(key)
                if v is None:
                    setattr(fi, key, None)
                elif kind == L.MEM_DEVICE or isinstance(v, int):
                    setattr(fi, key, int(v))   # raw address (device pointer, or a host pointer kept alive by the caller)
                else:
                    v = np.ascontiguousarray(v)
                    keep.append(v)
                    setattr(fi, key, v.ctypes.data)
            pose = f.get("pose")
            if pose is not None:
                fi.pose_valid = 1
                fi.pose_x = (C.c_double * 3)(*pose[0])
                fi.pose_q = (C.c_double * 4)(*pose[1])
            else:
                fi.pose_valid = 0
        self._keep = keep

    def build_inputs(self, frames):
        """Pre-build the ctypes input array of one frame (see submit) for submit_raw."""
        saved = self._inputs
        self._inputs = (L.FrameInput * self.n_objects)()
        self._fill(frames)
        arr, keep = self._inputs, self._keep
        self._inputs = saved
        return arr, keep

    def submit_raw(self, inputs):
        L.check(L.lib().roft_frame_submit(self._h, inputs, self.n_objects))

    def build_batch(self, frames_list):
        """ctypes input array of a batch: frames_list[t] = one dict per object (see submit), t = 0 .. T-1."""
        T = len(frames_list)
        arr = (L.FrameInput * (self.n_objects * T))()
        keep = []
        saved = self._inputs
        for t, frames in enumerate(frames_list):
            self._inputs = (L.FrameInput * self.n_objects)()
            self._fill(frames)
            for i in range(self.n_objects):
                arr[t * self.n_objects + i] = self._inputs[i]
            keep.append(self._keep)
        self._inputs = saved
        return arr, keep, T

    def submit_batch(self, frames_list):
        """A batch of consecutive frames (at most cfg.max_batch_frames), roft_frames_submit."""
        arr, keep, T = self.build_batch(frames_list)
        self._keep = keep
        L.check(L.lib().roft_frames_submit(self._h, arr, self.n_objects, T))

    def submit_batch_raw(self, arr, T):
        L.check(L.lib().roft_frames_submit(self._h, arr, self.n_objects, T))

    def retain_frames(self):
        return L.lib().roft_engine_retain_frames(self._h)

    def stats(self):
        st = L.EngineStats()
        L.check(L.lib().roft_engine_get_stats(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in L.EngineStats._fields_}

    def batch_trace(self, n=64):
        """The engine's record of its last (at most 64) batches, oldest first: scheduling decisions and host times
        (roft_batch_trace in include/roft_engine.h)."""
        arr = (L.BatchTrace * n)()
        got = C.c_int(0)
        L.check(L.lib().roft_engine_get_batch_trace(self._h, arr, n, C.byref(got)))
        return [{k: getattr(arr[i], k) for k, _ in L.BatchTrace._fields_} for i in range(got.value)]

    def step(self):
        L.check(L.lib().roft_step(self._h))

    def sync(self):
        L.check(L.lib().roft_sync(self._h))

    def state(self, obj):
        pose, P, tw, Pv = np.zeros(13), np.zeros((12, 12)), np.zeros(6), np.zeros((6, 6))
        L.check(L.lib().roft_get_state(self._h, obj, pose.ctypes.data, P.ctypes.data, tw.ctypes.data, Pv.ctypes.data))
        return pose, P, tw, Pv

    def outputs(self):
        outs = (L.ObjectOutput * self.n_objects)()
        L.check(L.lib().roft_get_outputs(self._h, outs, self.n_objects))
        return outs

    def mask(self, obj):
        m = np.zeros((self.H, self.W), np.uint8)
        L.check(L.lib().roft_get_mask(self._h, obj, m.ctypes.data))
        return m

    def enable_log(self, n_frames):
        L.check(L.lib().roft_engine_enable_log(self._h, n_frames))

    def get_log(self, first, n):
        outs = (L.ObjectOutput * (n * self.n_objects))()
        L.check(L.lib().roft_engine_get_log(self._h, first, n, outs))
        pose = np.zeros((n, self.n_objects, 13))
        twist = np.zeros((n, self.n_objects, 6))
        npts = np.zeros((n, self.n_objects), np.int64)
        sel = np.zeros((n, self.n_objects), np.int64)
        for f in range(n):
            for o in range(self.n_objects):
                r = outs[f * self.n_objects + o]
                pose[f, o] = r.pose[:]
                twist[f, o] = r.twist[:]
                npts[f, o] = r.n_flow_points
                sel[f, o] = r.outlier_selected
        return pose, twist, npts, sel

    def get_log_rows(self, first, n):
        """[n, n_objects, 19] float64: pose(13) | twist(6) per object-frame, as the reference logs them."""
        rows = np.zeros((n, self.n_objects, 19))
        L.check(L.lib().roft_engine_get_log_rows(self._h, first, n, rows.ctypes.data))
        return rows

    def stream(self):
        return L.lib().roft_engine_stream(self._h)

    def enable_timing(self, level=2):
        """0 off, 1 only the flow measurement kernel, 2 every launch group."""
        L.check(L.lib().roft_engine_enable_timing(self._h, int(level)))

    def timing(self):
        n = C.c_int(0)
        names = C.POINTER(C.c_char_p)()
        ms = C.POINTER(C.c_float)()
        launches = C.POINTER(C.c_int)()
        L.check(L.lib().roft_engine_get_timing(self._h, C.byref(n), C.byref(names), C.byref(ms), C.byref(launches)))
        return {names[i].decode(): (ms[i], launches[i]) for i in range(n.value)}

    def close(self):
        if self._h:
            L.lib().roft_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
