"""Host-side logic on CPU: synthetic stream generator, delivery schedules, object sharding over ranks
(world_size 2, gloo)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from roft_amd import parallel, synth

import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stream_shapes_and_schedule():
    st = util.stream(51, 14, scale=4)
    cam = st.camera
    assert st.depth.shape == (14, cam.height, cam.width) and st.depth.dtype.is_floating_point
    assert st.flow.shape == (14, cam.height, cam.width, 2)
    assert set(np.unique(st.mask_gt.numpy())) <= {0, 255}
    # DatasetImageSegmentationDelayed.cpp:42-63: frame h delivers frame max(h-6, 0) iff (h-6) % 6 == 0
    assert list(st.mask_delivery) == [0, -1, -1, -1, -1, -1, 0, -1, -1, -1, -1, -1, 6, -1]
    assert st.pose_valid[0] and not st.pose_valid[1:6].any()
    assert not st.flow_valid[0] and st.flow_valid[1:].all()
    bad = ~np.isfinite(st.flow[1:].numpy()) | (np.abs(st.flow[1:].numpy()) > 1e9)
    assert 0 < bad.mean() < 0.02                      # a few invalid flow entries (NaN / 1e10)
    assert 0.005 < (st.depth.numpy() == 0).mean() < 0.05


def test_stream_is_deterministic_and_s16_quantised():
    a = synth.make_stream(52, 3, synth.Camera.shape_a().scaled(4))
    b = synth.make_stream(52, 3, synth.Camera.shape_a().scaled(4))
    assert np.array_equal(a.depth.numpy(), b.depth.numpy()) and np.array_equal(a.pose_meas, b.pose_meas)
    s = synth.make_stream(52, 3, synth.Camera.shape_a().scaled(4), flow_type=synth.FLOW_S16C2)
    assert s.flow.dtype.is_floating_point is False and s.flow.shape[1:3] == (s.camera.height // 4, s.camera.width // 4)
    f32 = a.flow[1].numpy()[2::4, 2::4]
    ok = np.isfinite(f32).all(-1) & (np.abs(f32) < 1e9).all(-1)
    assert np.abs(s.flow[1].numpy()[ok] / 32.0 - f32[ok]).max() <= 1 / 64 + 1e-6


def test_flow_moves_mask_onto_next_mask():
    """Sanity of the generator itself: GT flow carries frame k-1's silhouette onto frame k's."""
    st = synth.make_stream(53, 3, synth.Camera.shape_a().scaled(2), flow_invalid=0.0, mask_dilate=0)
    m0, m1 = st.mask_gt[0].numpy() > 0, st.mask_gt[1].numpy() > 0
    fl = st.flow[1].numpy()
    vs, us = np.nonzero(m0)
    tx = np.round(us + fl[vs, us, 0]).astype(int).clip(0, m1.shape[1] - 1)
    ty = np.round(vs + fl[vs, us, 1]).astype(int).clip(0, m1.shape[0] - 1)
    assert m1[ty, tx].mean() > 0.97


def test_box_mesh_is_closed_and_sized_like_ycb():
    v, t = synth.box_mesh(synth.CRACKER_BOX_HALF_EXTENTS, 36)
    assert v.shape == (8214, 3) and t.shape == (15552, 3)
    assert np.allclose(np.abs(v).max(0), synth.CRACKER_BOX_HALF_EXTENTS)


def test_shard_objects_partitions():
    for n, g in ((64, 8), (64, 1), (5, 2), (3, 4), (16, 3)):
        parts = [parallel.shard_objects(n, r, g) for r in range(g)]
        flat = [i for p in parts for i in p]
        assert flat == list(range(n))
        assert max(len(p) for p in parts) == -(-n // g)
    assert parallel.weak_objects(4, 2) == [8, 9, 10, 11]


WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch
from roft_amd import parallel
rank, local, world = parallel.init("gloo")
mine = parallel.shard_objects(5, rank, world)
assert parallel.weak_objects(3, rank) == [3 * rank + i for i in range(3)]
parallel.barrier()
t = parallel.max_over_ranks(1.0 + rank)
assert t == float(world), t
rec = torch.full((3, 19), float(rank))
allrec = parallel.gather_records(rec)
assert allrec.shape == (3 * world, 19) and float(allrec[:, 0].sum()) == 3.0 * sum(range(world))
# uneven shards of engine-log-shaped records ([object, frame, 19]: pose 13 | twist 6), as bench.py gathers them:
# 5 objects over 2 ranks = 3 + 2; the gathered block is in global object order
n_frames = 4
rows = torch.zeros((len(mine), n_frames, 19), dtype=torch.float64)
for i, gid in enumerate(mine):
    rows[i] = gid * 100.0 + torch.arange(n_frames, dtype=torch.float64)[:, None] + torch.arange(19, dtype=torch.float64)[None, :] / 100.0
allrows = parallel.gather_records(rows)
assert allrows.shape == (5, n_frames, 19)
for gid in range(5):
    assert float(allrows[gid, 2, 7]) == gid * 100.0 + 2 + 0.07
# every object is owned by exactly one rank
own = torch.zeros(5)
own[mine] = 1
import torch.distributed as dist
dist.all_reduce(own)
assert bool((own == 1).all())
# shared scene: the ingest rank's frames reach every rank in place
depth = torch.arange(2 * 6 * 8, dtype=torch.float32).reshape(2, 6, 8) if rank == 0 else torch.zeros(2, 6, 8)
flow = torch.full((2, 6, 8, 2), 0.25) if rank == 0 else torch.zeros(2, 6, 8, 2)
parallel.broadcast_frames([depth, flow], src=0)
assert bool((depth == torch.arange(2 * 6 * 8, dtype=torch.float32).reshape(2, 6, 8)).all()) and bool((flow == 0.25).all())
print("rank", rank, "ok", mine)
'''


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "rank 0 ok [0, 1, 2]" in outs[0] and "rank 1 ok [3, 4]" in outs[1]


def test_rmse_metrics():
    from scipy.spatial.transform import Rotation
    from roft_amd import metrics
    rng = np.random.default_rng(8)
    n = 30
    x = rng.normal(size=(n, 3))
    assert abs(metrics.rmse_cartesian_3d(x, x + [0.03, 0.0, 0.04]) - 5.0) < 1e-12        # 5 cm
    q = Rotation.random(n, random_state=1)
    d = Rotation.from_rotvec(np.tile([0.0, np.radians(10.0), 0.0], (n, 1)))
    qa = q.as_quat()[:, [3, 0, 1, 2]]
    qb = (d * q).as_quat()[:, [3, 0, 1, 2]]
    assert abs(metrics.rmse_angular(qa, qb) - 10.0) < 1e-9
    assert abs(metrics.rmse_angular(qa, -qb) - 10.0) < 1e-9                              # double cover
    tw = rng.normal(size=(n, 6))
    v = metrics.object_velocity_from_twist(tw, x)
    assert np.allclose(v[:, :3], tw[:, :3] + np.cross(tw[:, 3:], x)) and np.array_equal(v[:, 3:], tw[:, 3:])
    assert abs(metrics.rmse_linear_velocity(tw[:, :3], tw[:, :3] + [0.01, 0, 0]) - 1.0) < 1e-12
    assert abs(metrics.rmse_angular_velocity(tw[:, 3:], tw[:, 3:] + [0, np.radians(2.0), 0]) - 2.0) < 1e-12
    assert metrics.time_metrics([10, 20, 40, 33, 34]) == (27.4, 2)
    # the synthetic ground-truth twist is the origin-form twist: moving the pole recovers the body velocity
    st = util.stream(60, 12, scale=8)
    body = metrics.object_velocity_from_twist(st.gt.twist, st.gt.x)
    fd = np.gradient(st.gt.x, st.dt, axis=0)
    assert np.abs(body[2:-2, :3] - fd[2:-2]).max() < 0.02


def test_metric_class_of_the_references_evaluation():
    """roft_amd.metrics.Metric = the interface of evaluation/metrics.py (which cannot be imported here: pyquaternion): every
    metric name, per object and pooled over 'ALL', against values worked out by hand and against the functions pinned by the
    bop_pose_error fixtures."""
    from roft_amd import io, metrics as M
    rng = np.random.default_rng(4)
    assert set(M.Metric.NAMES) == {"rmse_cartesian_3d", "rmse_cartesian_x", "rmse_cartesian_y", "rmse_cartesian_z", "rmse_angular",
                                   "rmse_linear_velocity", "rmse_angular_velocity", "max_linear_velocity", "max_angular_velocity",
                                   "add", "adi", "time", "excess_33_ms"}
    with pytest.raises(ValueError):
        M.Metric("nope")
    # poses: reference at rest, signal 1 cm off in x and rotated by 10 deg about z on every second row
    n = 6
    ref = np.zeros((n, 7)); ref[:, 3:] = [0.0, 0.0, 1.0, 0.0]
    sig = ref.copy(); sig[:, 0] = 0.01; sig[::2, 6] = np.radians(10.0)
    assert np.isclose(M.Metric("rmse_cartesian_3d").evaluate("a", ref, sig, None), 1.0)
    assert np.isclose(M.Metric("rmse_cartesian_x").evaluate("a", ref, sig, None), 1.0)
    assert M.Metric("rmse_cartesian_y").evaluate("a", ref, sig, None) == 0.0 == M.Metric("rmse_cartesian_z").evaluate("a", ref, sig, None)
    assert np.isclose(M.Metric("rmse_angular").evaluate("a", ref, sig, None), 10.0 * np.sqrt(0.5))
    # the same through the quaternion form used by bench.py
    q = lambda rows: np.array([io.axis_angle_to_quat(r[3:6], r[6]) for r in rows])
    assert np.isclose(M.Metric("rmse_angular").evaluate("a", ref, sig, None), M.rmse_angular(q(ref), q(sig)))
    # velocities
    vr = rng.normal(size=(n, 6)); vs = vr.copy(); vs[:, 0] += 0.02; vs[:, 5] -= np.radians(3.0)
    assert np.isclose(M.Metric("rmse_linear_velocity").evaluate("a", vr, vs, None), 2.0)
    assert np.isclose(M.Metric("rmse_angular_velocity").evaluate("a", vr, vs, None), 3.0)
    assert np.isclose(M.Metric("max_linear_velocity").evaluate("a", vr, vs, None), np.linalg.norm(vr[:, :3], axis=1).max())
    assert np.isclose(M.Metric("max_angular_velocity").evaluate("a", vr, vs, None), np.degrees(np.linalg.norm(vr[:, 3:], axis=1).max()))
    # times
    t = np.array([[10.0, 1.0], [40.0, 2.0], [33.0, 0.0], [34.0, 0.0]])
    assert M.Metric("time").evaluate("a", None, None, t) == 29.25 and M.Metric("excess_33_ms").evaluate("a", None, None, t) == 2.0
    # 'ALL': rows of all objects pooled
    two = {"a": ref, "b": ref[:2]}
    two_s = {"a": sig, "b": ref[:2]}
    pooled = np.sqrt((n * 1.0) / (n + 2))
    assert np.isclose(M.Metric("rmse_cartesian_3d").evaluate("ALL", two, two_s, None), pooled)
    assert M.Metric("time").evaluate("ALL", None, None, {"a": t, "b": t[:1]}) == np.mean([10.0, 40.0, 33.0, 34.0, 10.0])
    # ADD / ADD-S AUC: per object and pooled, against the functions the bop_pose_error fixtures pin
    pts = {"a": rng.normal(size=(200, 3)) * 0.05, "b": rng.normal(size=(150, 3)) * 0.03}
    poses = {}
    for name in pts:
        r = np.zeros((5, 7)); r[:, :3] = rng.normal(size=(5, 3)) * 0.1 + [0, 0, 0.7]
        ax = rng.normal(size=(5, 3)); r[:, 3:6] = ax / np.linalg.norm(ax, axis=1, keepdims=True); r[:, 6] = rng.uniform(0, 2, 5)
        s = r.copy(); s[:, :3] += rng.normal(size=(5, 3)) * 0.01; s[:, 6] += rng.normal(size=5) * 0.05
        poses[name] = (r, s)
    for ad, f in (("add", M.add), ("adi", M.adds)):
        m = M.Metric(ad, auc_points=pts)
        want_all = []
        for name, (r, s) in poses.items():
            d = [f(M.Metric._rot(s[k, 3:]), s[k, :3], M.Metric._rot(r[k, 3:]), r[k, :3], pts[name]) for k in range(5)]
            want_all += d
            assert np.isclose(m.evaluate(name, r, s, None), M.auc(np.array(d)))
        ref_d = {k: v[0] for k, v in poses.items()}
        sig_d = {k: v[1] for k, v in poses.items()}
        assert np.isclose(m.evaluate("ALL", ref_d, sig_d, None), M.auc(np.array(want_all)))
        assert 0.0 < m.evaluate("ALL", ref_d, sig_d, None) <= 100.0


def test_results_table_tool(tmp_path, capsys):
    """tools/evaluate_results.py on a results tree written in the reference's log format: per-object rows and the pooled ALL row,
    the dropped velocity columns of the pose log, the pole displacement of the linear velocity, the optical-flow time added."""
    import importlib.util
    import json
    from roft_amd import io
    rng = np.random.default_rng(8)
    results, dataset = tmp_path / "results", tmp_path / "dataset"
    n = 30
    for name, off in (("a", 0.01), ("b", 0.02)):
        (results / name).mkdir(parents=True)
        (dataset / name / "gt").mkdir(parents=True)
        gt = np.zeros((n + 5, 7)); gt[:, :3] = rng.normal(size=(n + 5, 3)) * 0.05 + [0, 0, 0.7]; gt[:, 3:] = [0, 0, 1, 0.3]
        np.savetxt(str(dataset / name / "gt" / "poses.txt"), gt)
        w = rng.normal(size=(n + 5, 3)) * 0.2
        gv = np.concatenate([rng.normal(size=(n + 5, 3)) * 0.1, w], 1)
        np.savetxt(str(dataset / name / "gt" / "velocities.txt"), gv)
        io.write_obj(str(dataset / name / "model.obj"), rng.normal(size=(60, 3)) * 0.04, np.array([[0, 1, 2]]))
        est = gt[:n].copy(); est[:, 0] += off
        np.savetxt(str(results / name / "pose_estimate.txt"), np.concatenate([np.zeros((n, 6)), est], 1))
        # the filter reports the velocity of the point at the camera origin: v_O = v - w x r
        vo = gv[:n].copy(); vo[:, :3] = gv[:n, :3] - np.cross(gv[:n, 3:], gt[:n, :3])
        np.savetxt(str(results / name / "velocity_estimate.txt"), vo)
        np.savetxt(str(results / name / "execution_times.txt"), np.stack([np.full(n, 2.0), np.zeros(n)], 1))
    spec = importlib.util.spec_from_file_location("evaluate_results", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "evaluate_results.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    assert ev.main(["--results", str(results), "--dataset", str(dataset), "--of-ms", "3", "--json", str(tmp_path / "t.json")]) == 0
    out = capsys.readouterr().out.strip().splitlines()
    assert out[0].startswith("| object | rmse_cartesian_3d (cm) | rmse_angular (deg) | add (AUC %) | adi (AUC %) | rmse_linear_velocity (cm/s)")
    assert [line.split("|")[1].strip() for line in out[2:]] == ["a", "b", "ALL"]
    t = json.load(open(str(tmp_path / "t.json")))
    assert np.isclose(t["rmse_cartesian_3d"]["a"], 1.0) and np.isclose(t["rmse_cartesian_3d"]["b"], 2.0)
    assert np.isclose(t["rmse_cartesian_3d"]["ALL"], np.sqrt((1.0 + 4.0) / 2.0))
    assert abs(t["rmse_angular"]["ALL"]) < 1e-5
    assert t["rmse_linear_velocity"]["a"] < 1e-9 and t["rmse_angular_velocity"]["b"] < 1e-9      # the pole was moved back
    assert t["time"]["ALL"] == 5.0 and t["excess_33_ms"]["ALL"] == 0.0
    assert 80.0 < t["add"]["b"] < t["add"]["a"] <= 100.0 and t["adi"]["a"] >= t["add"]["a"]


def test_batch_planners_cover_the_frames_once_and_end_with_the_pose_arrivals():
    """roft_amd.engine.aligned_batches (the batches a recorded sequence is cut into: each ends with a pose-arrival frame where
    a batch can hold a period) and bench.py's split_batches: every frame exactly once, in order, no batch above T; the
    driver's shape (5 warm-up + 20 timed frames, T = 8, arrivals every 6 frames) is 2, 6, 6, 6."""
    import importlib.util
    from roft_amd import engine as E
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_planner", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert [t for _, t in E.aligned_batches(5, 25, 8, 6)] == [2, 6, 6, 6]
    assert [t for _, t in E.aligned_batches(0, 5, 8, 6)] == [1, 4]
    rng = np.random.default_rng(3)
    for _ in range(200):
        first = int(rng.integers(0, 40)); last = first + int(rng.integers(1, 90)); T = int(rng.integers(1, 9))
        period = int(rng.integers(1, 13)); phase = int(rng.integers(0, period))
        for plan in (E.aligned_batches(first, last, T, period, phase), bench.split_batches(first, last, T), bench.split_batches(first, last, T, ramp=True)):
            k = first
            for k0, t in plan:
                assert k0 == k and 1 <= t <= T
                k += t
            assert k == last
        plan = E.aligned_batches(first, last, T, period, phase)
        if period <= T:
            for k0, t in plan[:-1]:   # (the last batch ends where the frames end)
                assert (k0 + t - 1) % period == phase, (first, last, T, period, phase, plan)


def test_bench_refuses_a_world_size_other_than_gpus():
    """`--gpus N` is a statement about the job: under a launcher whose WORLD_SIZE differs bench.py exits non-zero with the
    reason (before it imports torch) instead of printing a line whose n_gpus is not what the caller asked for."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr and "--gpus 4" in r.stderr
    assert r.stdout.strip() == ""


def test_bench_gpus_n_starts_its_own_ranks_and_relays_their_exit_code():
    """`python bench.py --gpus 2` without a launcher starts torch.distributed.run itself (a child process, two ranks).  Here,
    without a GPU, the ranks stop at their device check: the parent relays the failure and the reason reaches stderr."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    # (whichever rank reports first: the launcher stops the other one as soon as one has failed)
    assert "needs GPU" in r.stderr and ("rank 0" in r.stderr or "rank 1" in r.stderr)


def _canned_bench_record(world, n_windows=5, n_batches=40):
    """A bench record with every block filled the way a long run fills it (many windows, long batch traces, prose notes,
    per-rank blocks) -- the input of bench.headline()."""
    batch = dict(frames=8, steady=True, throttled=False, handoff=True, early_lanes=2, outlier_parts_halved=False, launches=21,
                 submit_us=123.4, wait_us=0.1, step_us=301.2, submitted_at_ms=0.1234, done_at_ms=1.2345)
    win = dict(value=1.0954321e6, ms_per_step=0.0584321, elapsed_ms=1.1686, elapsed_ms_this_rank=1.16, host_enqueue_ms_per_step=0.04,
               host_cpu=3, loadavg_1min=0.5, seed_base=4000, launches_per_frame=2.7, event_ops_per_frame=1.1, batches=[batch] * n_batches)
    rf = dict(kernel="flow_measure_kernel", bound="hbm", achieved=1514.123456, peak=8000.0, unit="GB/s", frac=0.189265432,
              frac_on_sample_bytes=0.0751234, traffic=45.8e6, traffic_raw_counter=33.3e6, traffic_source="profiles/r06_pmc_k1.json: " + "x" * 300,
              traffic_over_algorithmic=0.92, algorithmic_bytes_per_object_frame=155362.0, algorithmic_bytes_formula="y" * 300,
              object_frames_per_launch=320.0, avg_launch_us=32.83, launches=12, measured_copy_GBs=5280.0,
              frac_of_measured_random_sector_rate=0.33, note="z" * 1500,
              kernel_span=dict(avg_us=27.1, launches=12, achieved=1830.0, frac=0.229, note="n" * 400),
              alone=dict(avg_launch_us=18.6, launches=4, object_frames_per_launch=320.0, frac=0.279, note="n" * 300))
    other = {k: dict(declared_bytes=1e5, per="object-frame", avg_us_per_launch_group=95.0, launch_groups=4, declared_bytes_per_launch_group=5.1e7,
                     achieved_GBs=536.0, frac_of_hbm_peak=0.067, fetch_size_bytes_per_dispatch=14.4e6, write_size_bytes_per_dispatch=8.1e6)
             for k in ("mask_frames", "features", "outlier_fused")}
    other["note"] = "o" * 900
    cpu = dict(value=659.47, unit="object-frames/s", cores=1, kind="port", sample="8 objects x 32 frames of the same 640x480 streams, " + "s" * 200,
               cpu_model="AMD EPYC 9575F 64-Core Processor", ms_per_object_frame=1.5164)
    out = {"metric": "tracker frames/sec per object (640x480) + ADD-S vs CPU ref", "value": 1.0954321e6, "unit": "object-frames/s",
           "n_gpus": world, "steps": 20, "warmup": 5, "ms_per_step": 0.0584321, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic", "value_is": "v" * 200, "runs": [1.07e6, 1.08e6, 1.0954321e6, 1.1e6, 1.11e6][:n_windows],
           "value_min": 1.07e6, "value_max": 1.11e6, "value_first_window": 1.07e6, "host_enqueue_ms_per_step": 0.04,
           "host_state": {"cpu": 3, "loadavg_1min": 0.5, "cpus_online": 64}, "batches": [batch] * n_batches, "windows": [win] * n_windows,
           "instrumented_window": win, "inputs": "resident in HBM", "timed_frames_first_touch": True, "method": {"note": "m" * 2000},
           "config": {"workload": "BASELINE config #4: 640x480, CV_32FC2 flow grid 1, 64 objects in total, 8 per GPU (sharded by object, "
                                  "no data-path collective; result rows all-gathered over RCCL at N > 1), masks+poses at 5 fps with 6-frame delay, "
                                  "flow-aided masks, re-sync and outlier rejection on, frames submitted in batches of at most 8",
                      "objects_per_gpu": 64 // world, "objects_total": 64, "width": 640, "height": 480, "batch_frames": 8,
                      "timed_batches": [4, 8, 8], "ranks": world, "backend": "nccl"},
           "ranks": None, "shared_scene": None, "frames_per_sec_per_object": 17114.0, "launches_per_frame": 2.7, "event_ops_per_frame": 1.1,
           "roofline": rf, "roofline_other": other, "cpu_baseline": cpu if world == 1 else None,
           "cpu_baseline_multicore": dict(value=6492.0, cores=64, kind="port", sample="c" * 300) if world == 1 else None,
           "speedup_vs_cpu_1core": 1660.43 if world == 1 else None, "speedup_vs_cpu_multicore": 168.7 if world == 1 else None,
           "value_pcie_inclusive": 1.14e4, "value_pcie_inclusive_shared_scene": 2.4e5, "value_pcie_inclusive_in_place": 5.4e4,
           "value_pcie_inclusive_shared_scene_in_place": 4.4e5, "pcie_inclusive": {"note": "p" * 800}, "value_cold": 1.083e6,
           "cold_run": {"note": "c" * 400}, "live_latency": dict(median_us=212.0, p99_us=402.0, note="l" * 200),
           "adds_vs_gt_mm": {"mean": 7.9, "auc": 92.1, "objects": 64, "note": "a" * 100},
           "adds_vs_cpu_ref_mm": {"mean": 1.2e-9, "max": 6.4e-9, "objects_per_rank": [1] * world},
           "rmse_vs_gt": {"position_cm": 0.5}, "pipeline": "q" * 400, "kernels_post_run_breakdown": {"ukf_chain": dict(total_ms=1.0, marks=3, avg_us=333.0)},
           "dominant_kernel": "ukf_chain", "stream_generation_s": 3.2}
    if world > 1:
        out["ranks"] = {"world_size": world, "ranks_seen_by_all_reduce": world, "objects_per_gpu": [64 // world] * world,
                        "first_object_of_rank": list(range(0, 64, 64 // world)),
                        "devices": [dict(rank=r, local_rank=r, device=r, name="AMD Instinct MI355X", uuid="GPU-%032x" % r) for r in range(world)],
                        "window_ms_per_rank": [[1.1] * n_windows] * world, "per_gpu_rate_of_the_median_window": 1.3e5, "note": "r" * 300}
    return out


@pytest.mark.parametrize("world", [1, 2, 8])
def test_bench_line_is_small(world, tmp_path, capsys):
    """The driver keeps an 8 KB tail of bench.py's stdout and parses its last line (round 5's 20.7 KB line came back as
    `parsed: null`): the line is < 4 KB of strict JSON with the contract's keys, whatever the run recorded; the full
    record lands in the side file the line names."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    out = _canned_bench_record(world)
    out["roofline"]["traffic"] = float("nan")   # a non-finite number must not make the line non-strict JSON
    detail = str(tmp_path / "bench_detail.json")
    text = bench.emit(out, detail)
    printed = capsys.readouterr().out.strip().splitlines()
    assert printed[-1] == text and "\n" not in text
    assert len(text) < bench.LINE_LIMIT, len(text)
    line = json.loads(text, parse_constant=lambda c: pytest.fail("non-strict JSON constant " + c))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == world and line["config"]["ranks"] == world and "workload" in line["config"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"])
    assert line["roofline"]["traffic"] is None
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-5
    if world == 1:
        assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"])
    else:
        assert line["ranks"]["world_size"] == world and len(line["ranks"]["objects_per_gpu"]) == world
    full = json.load(open(detail))
    assert len(full["windows"]) == 5 and full["method"]["note"].startswith("m")   # nothing measured is lost
    # a tree the run cannot write to costs the side file, not the line
    text2 = bench.emit(out, "/proc/nonexistent/bench_detail.json")
    assert len(text2) < bench.LINE_LIMIT and "detail" not in json.loads(text2)
