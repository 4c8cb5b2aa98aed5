"""The C-ABI shared library: loads, exports every symbol include/roft_engine.h declares, and has no
CPU fallback (compute entry points fail loudly without a HIP device)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from roft_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "roft_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(roft_[a-z_0-9]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    L.build()
    lib = L.lib()
    decl = declared_symbols()
    assert len(decl) >= 25
    for name in decl:
        assert hasattr(lib, name), "libroft_hip.so does not export %s" % name
    assert sorted(L.ABI_SYMBOLS) == decl


def test_struct_layouts_match_header_sizes():
    # sizes the C compiler gives the ABI structs (guards the ctypes mirrors)
    import subprocess
    import tempfile
    src = '#include <stdio.h>\n#include "roft_engine.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(roft_camera), sizeof(roft_flow), sizeof(roft_config), sizeof(roft_object_desc), sizeof(roft_frame_input), sizeof(roft_object_output), sizeof(roft_batch_trace), sizeof(roft_engine_stats));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(d, "s")]).split()]
    assert sizes == [C.sizeof(L.Camera), C.sizeof(L.Flow), C.sizeof(L.Config), C.sizeof(L.ObjectDesc),
                     C.sizeof(L.FrameInput), C.sizeof(L.ObjectOutput), C.sizeof(L.BatchTrace), C.sizeof(L.EngineStats)]
    # (the trailing ints of roft_config share one alignment slot: a field missing from the mirror would not change the size)
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "roft_engine.h"\nint main(){printf("%zu %zu %zu %zu\\n", offsetof(roft_config, device), '
           'offsetof(roft_config, max_batch_frames), offsetof(roft_config, mask_workgroups_per_object), '
           'offsetof(roft_config, outlier_bands_per_alternative));return 0;}')
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "o.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "o.c"), "-o", os.path.join(d, "o")])
        offs = [int(x) for x in subprocess.check_output([os.path.join(d, "o")]).split()]
    assert offs == [L.Config.device.offset, L.Config.max_batch_frames.offset, L.Config.mask_workgroups_per_object.offset,
                    L.Config.outlier_bands_per_alternative.offset]


def test_defaults_follow_the_reference_config():
    """config/config_fast_ycb.cfg + test/test.sh:70-71 defaults."""
    lib = L.lib()
    cfg = L.Config()
    assert lib.roft_default_config(C.byref(cfg), 1280, 720, L.FLOW_S16C2) == 0
    assert (cfg.flow_grid, cfg.flow_scale) == (4, 32.0)
    assert abs(cfg.cam.fx - 1229.4285612615463) < 1e-9 and cfg.cam.cx == 640.0 and cfg.cam.cy == 360.0
    assert abs(cfg.sample_time - 1 / 30) < 1e-12 and (cfg.ut.alpha, cfg.ut.beta, cfg.ut.kappa) == (1.0, 2.0, 0.0)
    assert cfg.depth_maximum == 2.0 and cfg.subsampling_radius == 35.0 and cfg.flow_weighting == 1
    assert (cfg.use_pose, cfg.use_pose_resync, cfg.use_velocity, cfg.outlier_rejection, cfg.flow_aided_segmentation) == (1,) * 5
    assert cfg.mask_frames_between == 6 and cfg.pose_frames_between == 6
    o = L.ObjectDesc()
    assert lib.roft_default_object(C.byref(o)) == 0
    assert list(o.p_cov0_diag) == [1e-3] * 12 and list(o.v_q_diag) == [0.1] * 6
    assert list(o.p_meas_cov_v) == [0.1] * 3 and list(o.p_meas_cov_w) == [1e-4] * 3
    assert list(o.p_meas_cov_x) == [1e-3] * 3 and list(o.p_meas_cov_q) == [1e-4] * 3
    assert list(o.v_meas_cov_flow) == [1.0, 1.0] and o.p_mean0[9] == 1.0


def test_no_cpu_fallback():
    lib = L.lib()
    if lib.roft_device_count() > 0:
        pytest.skip("a HIP device is present")
    from roft_amd import ops
    with pytest.raises(L.RoftError):
        ops.kf_predict(np.zeros(6), np.eye(6), np.ones(6))
    with pytest.raises(L.RoftError):
        ops.mask_propagate(np.zeros((64, 64), np.uint8), [])
    from roft_amd import engine as E
    with pytest.raises(L.RoftError):
        E.ROFTFilterBatch(E.default_config(640, 480))
    assert b"no HIP device" in lib.roft_last_error_string() or b"CPU" in lib.roft_last_error_string()


def test_argument_validation_without_device():
    lib = L.lib()
    assert lib.roft_default_config(None, 640, 480, L.FLOW_F32C2) == -1
    assert lib.roft_pose_process_noise(None, None, 0.1, None) == -1
    Q = np.zeros((9, 9))
    a = np.ones(3)
    assert lib.roft_pose_process_noise(a.ctypes.data, a.ctypes.data, 0.5, Q.ctypes.data) == 0
    assert Q[0, 0] == 0.5 and Q[3, 3] == 1.0 and Q[0, 6] == 0.125


def _plan(cfg, pose_valid):
    lib = L.lib()
    n = len(pose_valid)
    pv = (C.c_int * n)(*[int(v) for v in pose_valid])
    ns, nc, out = (C.c_int * n)(), (C.c_int * n)(), (C.c_int * n)()
    slots = (C.c_int * (n * 10))()
    lib.roft_debug_plan.argtypes = [C.POINTER(L.Config), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    assert lib.roft_debug_plan(C.byref(cfg), pv, n, ns, nc, out, slots) == 0
    return list(ns), list(nc), list(out), np.array(list(slots)).reshape(n, 10)


def test_frame_program_matches_the_oracle_state_machine(oracle):
    """Host logic, no GPU: the per-frame UKF program (re-sync replays, outlier step, velocity deque) built by
    the engine has exactly as many corrections per frame as the oracle's ROFTFilter restatement performs."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import util
    for seed, drop in ((71, 0.0), (72, 0.5)):
        st = util.stream(seed, 40, scale=8, pose_drop_prob=drop)
        for over in (dict(), dict(use_pose_resync=0), dict(outlier_rejection=0), dict(use_pose=0, use_pose_resync=0, outlier_rejection=0)):
            ref = util.run_oracle_tracker(oracle, st, **over)
            cfg = L.Config()
            L.lib().roft_default_config(C.byref(cfg), 640, 480, L.FLOW_F32C2)
            for k, v in over.items():
                setattr(cfg, k, v)
            ns, nc, out, slots = _plan(cfg, st.pose_valid)
            from oracle import binding as ob
            cfg_o = util.oracle_config(ob, st, **over)
            trk = ob.Tracker(cfg_o, *st.mesh)
            want = []
            for k in range(40):
                depth, flow, mask, pose = util.frame_inputs(st, k)
                want.append(trk.step(st.dt, depth, flow, mask, pose).n_ukf_corrections)
            trk.close()
            assert nc == want, (seed, over)
            tested = [k for k in range(40) if out[k] >= 0]
            assert tested == [k for k, r in enumerate(ref) if r["sel"] >= 0]
    # steady state with re-sync: a pose frame replays the D + 1 = 7 buffered twists, oldest first
    cfg = L.Config()
    L.lib().roft_default_config(C.byref(cfg), 640, 480, L.FLOW_F32C2)
    pv = [k % 6 == 0 for k in range(20)]
    ns, nc, out, slots = _plan(cfg, pv)
    assert ns[12] == 7 and nc[12] == 8 and out[12] == 0
    assert list(slots[12][:7]) == [6, 7, 8, 9, 10, 11, 12]
    assert ns[13] == 1 and nc[13] == 1 and out[13] == -1 and slots[13][0] == 13
