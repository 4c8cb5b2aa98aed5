"""ctypes binding of libroft_hip.so (the C ABI declared in include/roft_engine.h).

There is no CPU path: if the library is missing it is built with hipcc, and if that fails -- or if
a compute entry point is called without a HIP device -- an exception is raised.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO = os.environ.get("ROFT_LIB_SO") or os.path.join(CSRC, "libroft_hip.so")   # (override: A/B runs of two builds)

OK = 0
FLOW_S16C2 = 11
FLOW_F32C2 = 13
MEAS_NONE, MEAS_VELOCITY, MEAS_POSE, MEAS_POSE_VELOCITY = 0, 1, 2, 3
MEM_HOST, MEM_DEVICE = 0, 1
RETAIN_FRAMES = 16        # default configuration (roft_engine_retain_frames otherwise)
MAX_BATCH_FRAMES = 8
MAX_FLOW_CHASE = 30


class RoftError(RuntimeError):
    pass


class Camera(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("fx", C.c_double), ("fy", C.c_double),
                ("cx", C.c_double), ("cy", C.c_double)]


class Flow(C.Structure):
    _fields_ = [("data", C.c_void_p), ("type", C.c_int), ("cols", C.c_int), ("rows", C.c_int),
                ("grid", C.c_int), ("scale", C.c_float), ("valid", C.c_int)]


class UT(C.Structure):
    _fields_ = [("alpha", C.c_double), ("beta", C.c_double), ("kappa", C.c_double)]


class OFParams(C.Structure):
    _fields_ = [("levels", C.c_int), ("radius", C.c_int), ("iterations", C.c_int), ("det_min", C.c_float)]


class Mesh(C.Structure):
    _fields_ = [("verts", C.c_void_p), ("n_verts", C.c_int), ("tris", C.c_void_p), ("n_tris", C.c_int)]


class Config(C.Structure):
    _fields_ = [("cam", Camera), ("flow_type", C.c_int), ("flow_grid", C.c_int), ("flow_scale", C.c_float),
                ("sample_time", C.c_double), ("ut", UT), ("depth_maximum", C.c_double),
                ("subsampling_radius", C.c_double), ("flow_weighting", C.c_int), ("use_pose", C.c_int),
                ("use_pose_resync", C.c_int), ("use_velocity", C.c_int), ("outlier_rejection", C.c_int),
                ("flow_aided_segmentation", C.c_int), ("mask_frames_between", C.c_int),
                ("pose_frames_between", C.c_int), ("stamped_masks", C.c_int), ("max_objects", C.c_int), ("ukf_cholesky_guard", C.c_double),
                ("ukf_cholesky_guard_bilinear", C.c_double),
                ("device", C.c_int), ("max_batch_frames", C.c_int), ("mask_workgroups_per_object", C.c_int),
                ("outlier_bands_per_alternative", C.c_int)]


class ObjectDesc(C.Structure):
    _fields_ = [("p_mean0", C.c_double * 13), ("p_cov0_diag", C.c_double * 12), ("v_mean0", C.c_double * 6),
                ("v_cov0_diag", C.c_double * 6), ("p_sigma_ang_vel", C.c_double * 3),
                ("p_psd_lin_acc", C.c_double * 3), ("v_q_diag", C.c_double * 6),
                ("p_meas_cov_v", C.c_double * 3), ("p_meas_cov_w", C.c_double * 3),
                ("p_meas_cov_x", C.c_double * 3), ("p_meas_cov_q", C.c_double * 3),
                ("v_meas_cov_flow", C.c_double * 2), ("mesh", Mesh)]


class FrameInput(C.Structure):
    _fields_ = [("dt", C.c_double), ("depth", C.c_void_p), ("flow", C.c_void_p), ("mask", C.c_void_p),
                ("pose_valid", C.c_int), ("pose_x", C.c_double * 3), ("pose_q", C.c_double * 4),
                ("mem_kind", C.c_int), ("stamp", C.c_double), ("mask_stamp", C.c_double)]


class EngineStats(C.Structure):
    _fields_ = [("frames", C.c_longlong), ("batches", C.c_longlong), ("launches", C.c_longlong),
                ("event_ops", C.c_longlong), ("h2d_bytes", C.c_longlong), ("h2d_copies", C.c_longlong)]


class BatchTrace(C.Structure):
    _fields_ = [("batch", C.c_int), ("frames", C.c_int), ("steady", C.c_int), ("throttled", C.c_int), ("handoff", C.c_int),
                ("early_lanes", C.c_int), ("outlier_parts_halved", C.c_int), ("launches", C.c_int), ("event_ops", C.c_int),
                ("t_submit_us", C.c_double), ("submit_us", C.c_double), ("wait_us", C.c_double), ("step_us", C.c_double),
                ("t_done_us", C.c_double)]


class ObjectOutput(C.Structure):
    _fields_ = [("pose", C.c_double * 13), ("twist", C.c_double * 6), ("n_flow_points", C.c_int),
                ("outlier_selected", C.c_int), ("outlier_L", C.c_double * 2)]


# every symbol include/roft_engine.h declares
ABI_SYMBOLS = [
    "roft_last_error_string", "roft_device_count", "roft_flow_measurement", "roft_kf_predict",
    "roft_skf_correct", "roft_skf_correct_points", "roft_mask_propagate", "roft_pose_process_noise", "roft_ukf_predict",
    "roft_ukf_correct", "roft_mesh_classify", "roft_render_depth", "roft_depth_likelihood", "roft_outlier_test", "roft_outlier_test_split", "roft_default_config",
    "roft_default_object", "roft_engine_create", "roft_engine_destroy", "roft_object_add",
    "roft_frame_submit", "roft_frames_submit", "roft_engine_retain_frames", "roft_engine_get_stats", "roft_step", "roft_sync", "roft_get_state", "roft_get_outputs", "roft_get_mask",
    "roft_engine_enable_log", "roft_engine_get_log", "roft_engine_get_log_rows", "roft_engine_stream", "roft_engine_enable_timing",
    "roft_engine_get_timing", "roft_engine_get_batch_trace", "roft_default_of_params", "roft_optical_flow", "roft_flow_producer_create",
    "roft_flow_producer_destroy", "roft_flow_producer_run", "roft_flow_producer_sync", "roft_flow_producer_stream",
    "roft_debug_plan", "roft_debug_get_dbg", "roft_debug_probe_streams", "roft_debug_sector_rate",
    "roft_host_alloc", "roft_host_free", "roft_host_is_pinned", "roft_debug_get_residency", "roft_debug_outlier_split",
]


def build(force=False):
    """Compile libroft_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "roft_engine.h"))
    stale = (not os.path.exists(SO)) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs)
    if force or stale:
        jobs = str(min(8, os.cpu_count() or 1))
        r = subprocess.run(["make", "-C", CSRC, "-j", jobs, "libroft_hip.so"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RoftError("building libroft_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        build()
    L = C.CDLL(SO)
    vp, ip, dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)
    L.roft_last_error_string.restype = C.c_char_p
    L.roft_device_count.restype = C.c_int
    L.roft_flow_measurement.argtypes = [C.POINTER(Camera), vp, vp, C.POINTER(Flow), C.c_double, C.c_float,
                                        C.c_double, C.c_int, vp, vp, vp, ip]
    L.roft_kf_predict.argtypes = [vp] * 5
    L.roft_skf_correct.argtypes = [vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, ip]
    L.roft_skf_correct_points.argtypes = [C.POINTER(Camera), C.c_double, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, ip]
    L.roft_mask_propagate.argtypes = [vp, C.c_int, C.c_int, C.POINTER(Flow), C.c_int, C.c_int]
    L.roft_pose_process_noise.argtypes = [vp, vp, C.c_double, vp]
    L.roft_ukf_predict.argtypes = [vp, vp, vp, C.c_double, C.POINTER(UT), vp, vp]
    L.roft_ukf_correct.argtypes = [vp, vp, C.c_int, vp, vp, C.POINTER(UT), vp, vp, ip]
    L.roft_render_depth.argtypes = [C.POINTER(Mesh), vp, vp, C.POINTER(Camera), C.c_int, vp]
    L.roft_depth_likelihood.argtypes = [C.POINTER(Camera), vp, vp, vp, C.c_int, dp, C.POINTER(C.c_long)]
    L.roft_outlier_test.argtypes = [C.POINTER(Camera), C.c_int, vp, vp, C.POINTER(Mesh), vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, ip, vp]
    L.roft_outlier_test_split.argtypes = [C.POINTER(Camera), C.c_int, vp, vp, C.POINTER(Mesh), vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, ip, vp]
    L.roft_engine_get_batch_trace.argtypes = [vp, C.POINTER(BatchTrace), C.c_int, ip]
    L.roft_default_config.argtypes = [C.POINTER(Config), C.c_int, C.c_int, C.c_int]
    L.roft_default_object.argtypes = [C.POINTER(ObjectDesc)]
    L.roft_engine_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.roft_engine_destroy.argtypes = [vp]
    L.roft_object_add.argtypes = [vp, C.POINTER(ObjectDesc), ip]
    L.roft_frame_submit.argtypes = [vp, C.POINTER(FrameInput), C.c_int]
    L.roft_frames_submit.argtypes = [vp, C.POINTER(FrameInput), C.c_int, C.c_int]
    L.roft_engine_retain_frames.argtypes = [vp]
    L.roft_engine_get_stats.argtypes = [vp, C.POINTER(EngineStats)]
    L.roft_step.argtypes = [vp]
    L.roft_sync.argtypes = [vp]
    L.roft_get_state.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    L.roft_get_outputs.argtypes = [vp, C.POINTER(ObjectOutput), C.c_int]
    L.roft_get_mask.argtypes = [vp, C.c_int, vp]
    L.roft_engine_enable_log.argtypes = [vp, C.c_int]
    L.roft_engine_get_log.argtypes = [vp, C.c_int, C.c_int, C.POINTER(ObjectOutput)]
    L.roft_engine_get_log_rows.argtypes = [vp, C.c_int, C.c_int, vp]
    L.roft_engine_stream.restype = vp
    L.roft_engine_stream.argtypes = [vp]
    L.roft_engine_enable_timing.argtypes = [vp, C.c_int]
    L.roft_engine_get_timing.argtypes = [vp, ip, C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.POINTER(C.c_float)),
                                         C.POINTER(ip)]
    L.roft_default_of_params.argtypes = [C.POINTER(OFParams)]
    L.roft_optical_flow.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(OFParams), C.c_int, vp]
    L.roft_flow_producer_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(OFParams), C.c_int, C.c_int, C.POINTER(vp)]
    L.roft_flow_producer_destroy.argtypes = [vp]
    L.roft_flow_producer_run.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_int]
    L.roft_flow_producer_sync.argtypes = [vp]
    L.roft_flow_producer_stream.restype = vp
    L.roft_flow_producer_stream.argtypes = [vp]
    L.roft_host_alloc.restype = vp
    L.roft_host_alloc.argtypes = [C.c_size_t]
    L.roft_host_free.restype = None
    L.roft_host_free.argtypes = [vp]
    L.roft_host_is_pinned.argtypes = [vp]
    for name in ABI_SYMBOLS:
        if name.startswith("roft_debug_") and not hasattr(L, name):
            continue   # (an older build loaded through ROFT_LIB_SO for an A/B run: diagnostics only; tests/test_abi_cpu.py checks the in-tree library has them all)
        f = getattr(L, name)
        if name not in ("roft_last_error_string", "roft_engine_stream", "roft_flow_producer_stream", "roft_host_alloc", "roft_host_free"):
            f.restype = C.c_int
    _lib = L
    return L


def check(rc):
    if rc != OK:
        raise RoftError("libroft_hip error %d: %s" % (rc, lib().roft_last_error_string().decode()))


def require_device():
    if lib().roft_device_count() <= 0:
        raise RoftError("no HIP device visible: roft_amd has no CPU fallback")
