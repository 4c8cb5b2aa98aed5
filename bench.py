#!/usr/bin/env python3
"""bench.py -- tracker throughput of the MI355X-native ROFT engine on synthetic Fast-YCB-shaped streams.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it with
torch.distributed.run (one rank per GPU).  Rank 0 prints ONE JSON line.

Workload = BASELINE.json config #4: 640x480 frames, CV_32FC2 grid-1 optical flow, 64 independently tracked objects
in total, block-sharded over the GPUs (64 / 32 / 16 / 8 per GPU at N = 1 / 2 / 4 / 8: strong scaling; objects
never exchange data, so there is no data-path collective -- at N > 1 the ranks all-gather the per-object result rows over
RCCL inside the timed region).  All reference features on (flow-aided masks, Laplacian re-weighting, 5 fps / 6-frame
delayed masks and poses, pose re-sync, depth-render outlier rejection).  One "step" = one camera frame for every object
= ROFTFilter::filtering_step x n_objects; frames are handed to the engine in batches of --batch frames
(roft_frames_submit).  Inputs (depth, flow, masks) are resident in HBM before the timed region; the PCIe-inclusive rates
(HOST inputs) are measured in the same run and reported beside `value`, never as it.

Round 5: the timed region is run --windows times (default 5) in one invocation, every time on a FRESH engine and on the workload's
streams generated anew into FRESH buffers (W warm-up steps, barrier + synchronize, exactly K timed steps, synchronize + barrier, max over ranks), with
no timing machinery of any kind inside it; `value` is the MEDIAN window (a real run: its ms_per_step, host time and batch
trace are the top-level ones), `runs` lists all of them.  A 20-step window is 1.3 ms of GPU time in which one host thread
enqueues ~70 launches: one window can be hit by anything that happens on the box in that millisecond, five cannot.  The
roofline kernel is timed (HIP event pair on its own dispatch + device clock span) in one MORE window of the same shape,
`instrumented_window`, which is never `value`.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E datasheet peak (/opt/skills/guides/MI355X_MICROARCH.md)


# ---------------------------------------------------------------------------------------------------------------
# cpu_baseline legs (the only place bench.py touches oracle/): the oracle's ROFTFilter restatement on host cores
# ---------------------------------------------------------------------------------------------------------------
def cpu_worker(sample, sync, idx):
    """`bench.py --cpu-worker <sample.npz> <sync dir> <i>`: one oracle ROFTFilter instance (one object) on one host
    core, like one tracker process of the reference (one compute thread per process, src/roft/src/main.cpp:421-424).
    No GPU, no torch.  Frame loading is excluded from the time like the reference does (ROFTFilter.cpp:267-270,372-384)."""
    from oracle import binding as ob
    z = np.load(sample)
    depth, flow, masks = z["depth"], z["flow"], z["masks"]
    flow_valid, mask_delivery, pose_valid, pose_meas = z["flow_valid"], z["mask_delivery"], z["pose_valid"], z["pose_meas"]
    cam = z["cam"]
    W, H = int(cam[0]), int(cam[1])
    cfg = ob.default_config(640 if W in (640, 320, 160) else 1280, H)
    cfg.cam.width, cfg.cam.height = W, H
    cfg.cam.fx, cfg.cam.fy, cfg.cam.cx, cfg.cam.cy = [float(v) for v in cam[2:6]]
    for i in range(13):
        cfg.p_mean0[i] = float(z["p_mean0"][i])
    trk = ob.Tracker(cfg, z["verts"], z["tris"])
    dt = float(z["dt"])
    n = depth.shape[0]
    open(os.path.join(sync, "ready_%d" % idx), "w").close()
    go = os.path.join(sync, "go")
    while not os.path.exists(go):
        time.sleep(0.002)
    spent = 0.0
    t_first = time.time()
    for k in range(n):
        mi = int(mask_delivery[k])
        pose = (pose_meas[k, :3], pose_meas[k, 3:]) if pose_valid[k] else None
        t1 = time.perf_counter()
        trk.step(dt, depth[k], flow[k] if flow_valid[k] else None, masks[mi] if mi >= 0 else None, pose)
        spent += time.perf_counter() - t1
    t_last = time.time()
    trk.close()
    print(json.dumps(dict(frames=n, tracker_s=spent, t_first=t_first, t_last=t_last)))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_multicore(streams, n_sample, n_frames, n_objects_total):
    """One oracle process per object over min(n_objects, nproc) cores, all running at the same time.  The
    processes track copies of the first n_sample objects' streams (written once to /dev/shm)."""
    from roft_amd import synth
    n_workers = max(1, min(n_objects_total, os.cpu_count() or 1))
    tmp = tempfile.mkdtemp(prefix="roft_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    procs = []
    try:
        for o in range(n_sample):
            st = streams[o]
            used = sorted(set(int(m) for m in st.mask_delivery[:n_frames] if m >= 0))
            remap = {m: i for i, m in enumerate(used)}
            cam = st.camera
            np.savez(os.path.join(tmp, "s%d.npz" % o), depth=st.depth[:n_frames].cpu().numpy(), flow=st.flow[:n_frames].cpu().numpy(),
                     masks=st.mask_gt[used].cpu().numpy(), flow_valid=st.flow_valid[:n_frames],
                     mask_delivery=np.array([remap.get(int(m), -1) for m in st.mask_delivery[:n_frames]]),
                     pose_valid=st.pose_valid[:n_frames], pose_meas=st.pose_meas[:n_frames],
                     cam=np.array([cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy]),
                     p_mean0=synth.initial_pose_from_stream(st), verts=st.mesh[0], tris=st.mesh[1], dt=st.dt)
        env = dict(os.environ, OMP_NUM_THREADS="1")
        for w in range(n_workers):
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker",
                                           os.path.join(tmp, "s%d.npz" % (w % n_sample)), tmp, str(w)],
                                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env))
        deadline = time.time() + 120.0
        while time.time() < deadline and not all(os.path.exists(os.path.join(tmp, "ready_%d" % w)) for w in range(n_workers)):
            if any(p.poll() is not None for p in procs):
                break
            time.sleep(0.01)
        open(os.path.join(tmp, "go"), "w").close()
        res = []
        for p in procs:
            out, err = p.communicate(timeout=600)
            if p.returncode != 0:
                raise RuntimeError("cpu baseline worker failed: " + err.decode()[-400:])
            res.append(json.loads(out.decode().strip().splitlines()[-1]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for f in os.listdir(tmp):
            os.unlink(os.path.join(tmp, f))
        os.rmdir(tmp)
    frames = sum(r["frames"] for r in res)
    wall = max(r["t_last"] for r in res) - min(r["t_first"] for r in res)
    flags = "gcc -O2 (oracle/Makefile), one process per object, OMP_NUM_THREADS=1"
    return dict(value=frames / wall, unit="object-frames/s", cores=n_workers, kind="port",
                sample="%d concurrent oracle processes (one per object, copies of the first %d streams) x %d frames of the "
                       "640x480 workload; wall clock from the first to the last tracker call" % (n_workers, n_sample, n_frames),
                cpu_model=cpu_model(), host_cores=os.cpu_count(), flags=flags,
                mean_ms_per_object_frame=1e3 * sum(r["tracker_s"] for r in res) / frames)


# ---------------------------------------------------------------------------------------------------------------
# The line the driver parses.  The driver keeps an 8 KB tail of stdout: the line is a fixed set of numbers and short
# identifiers (no prose, no per-window traces), strict JSON, < 4 KB at any N.  Everything else the run measured goes
# to a side file whose path the line names (`detail`).  tests/test_host_cpu.py::test_bench_line_is_small holds this.
# ---------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 4096

_TOP_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
             "vs_baseline", "dtype", "data", "inputs", "runs", "value_cold", "speedup_vs_cpu_1core", "speedup_vs_cpu_multicore",
             "value_pcie_inclusive", "value_pcie_inclusive_shared_scene", "value_pcie_inclusive_in_place",
             "value_pcie_inclusive_shared_scene_in_place", "launches_per_frame", "dominant_kernel")
_CONFIG_KEYS = ("workload", "objects_per_gpu", "objects_total", "width", "height", "batch_frames", "ranks", "backend")
_ROOFLINE_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_on_sample_bytes", "traffic",
                  "traffic_over_algorithmic", "algorithmic_bytes_per_object_frame", "object_frames_per_launch",
                  "avg_launch_us", "launches", "measured_copy_GBs", "frac_of_measured_random_sector_rate")
_CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "cpu_model")


def _finite(x, digits=6):
    """Numbers of the line: finite floats rounded to `digits` significant digits, anything non-finite -> None (strict JSON)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, (float, np.floating)):
        x = float(x)
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (digits, x)) if digits else x
    if isinstance(x, np.integer):
        return int(x)
    if isinstance(x, dict):
        return {k: _finite(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v, digits) for v in x]
    return str(x)


def headline(out, detail_path=None):
    """The compact record of a run: a whitelist of `out`'s numbers (see the block comment above)."""
    line = {k: out.get(k) for k in _TOP_KEYS if k in out}
    cfg = out.get("config") or {}
    line["config"] = {k: cfg[k] for k in _CONFIG_KEYS if k in cfg}
    rf = out.get("roofline")
    if rf:
        r = {k: rf.get(k) for k in _ROOFLINE_KEYS if k in rf}
        if rf.get("kernel_span"):
            r["kernel_span_frac"] = rf["kernel_span"].get("frac")
            r["kernel_span_us"] = rf["kernel_span"].get("avg_us")
        if rf.get("alone"):
            r["alone_frac"] = rf["alone"].get("frac")
            r["alone_us"] = rf["alone"].get("avg_launch_us")
        line["roofline"] = r
    else:
        line["roofline"] = None
    ro = out.get("roofline_other")
    if ro:
        line["roofline_other"] = {k: {"us": v.get("avg_us_per_launch_group"), "frac": v.get("frac_of_hbm_peak")}
                                  for k, v in ro.items() if isinstance(v, dict)}
    cb = out.get("cpu_baseline")
    line["cpu_baseline"] = {k: cb.get(k) for k in _CPU_KEYS if k in cb} if cb else None
    cm = out.get("cpu_baseline_multicore")
    if cm and "value" in cm:
        line["cpu_baseline_multicore"] = {"value": cm["value"], "cores": cm["cores"], "kind": cm.get("kind", "port")}
    for k in ("adds_vs_cpu_ref_mm", "adds_vs_gt_mm"):
        v = out.get(k)
        if v:
            line[k] = {kk: vv for kk, vv in v.items() if kk in ("mean", "max", "auc", "objects")}
    rk = out.get("ranks")
    if rk:
        line["ranks"] = {"world_size": rk.get("world_size"), "seen": rk.get("ranks_seen_by_all_reduce"),
                         "objects_per_gpu": rk.get("objects_per_gpu")}
    ss = out.get("shared_scene")
    if ss:
        line["shared_scene"] = {"broadcast_MB_per_step": ss.get("broadcast_MB_per_step")}
    lv = out.get("live_latency")
    if lv:
        line["live_latency_us"] = {"median": lv.get("median_us"), "p99": lv.get("p99_us")}
    if detail_path:
        line["detail"] = detail_path
    line = _finite(line)
    if line.get("cpu_baseline") and isinstance(line["cpu_baseline"].get("sample"), str):
        line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:120]
    return line


def emit(out, detail_path):
    """Full record -> `detail_path` (best effort: a read-only tree must not cost the run its line); the compact line -> stdout,
    as the LAST line of the process."""
    written = None
    if detail_path:
        try:
            with open(detail_path, "w") as f:
                json.dump(_finite(out, 0), f, allow_nan=False, indent=1)   # (digits 0: as measured)
            written = os.path.relpath(detail_path, ROOT) if os.path.abspath(detail_path).startswith(ROOT) else detail_path
        except OSError as ex:
            sys.stderr.write("bench.py: could not write %s: %s\n" % (detail_path, ex))
    text = json.dumps(headline(out, written), allow_nan=False, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:   # cannot happen with the whitelist above; if it ever does, drop the optional blocks, keep the contract keys
        slim = headline(out, written)
        for k in ("roofline_other", "live_latency_us", "adds_vs_gt_mm", "runs", "shared_scene"):
            slim.pop(k, None)
        text = json.dumps(slim, allow_nan=False, separators=(",", ":"))
    sys.stdout.flush()
    print(text, flush=True)
    return text


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--json-out", default=os.path.join(ROOT, "bench_detail.json"),
                   help="the full record of the run (windows, batch traces, method, notes); the stdout line carries the numbers only")
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=60)
    p.add_argument("--warmup", type=int, default=12)
    p.add_argument("--batch", type=int, default=8, help="frames per roft_frames_submit (1..8)")
    p.add_argument("--k1-windows", type=int, default=3,
                   help="instrumented windows (same shape as the timed ones, never `value`) whose launches of the roofline kernel are pooled")
    p.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                   help="strong: --objects in total, block-sharded over the GPUs (BASELINE config #4); weak: --objects per GPU")
    p.add_argument("--objects", "--objects-per-gpu", dest="objects", type=int, default=64)
    p.add_argument("--shape", default="A", choices=["A", "B"])
    p.add_argument("--flow", default="f32", choices=["f32", "s16"])
    p.add_argument("--cpu-sample-objects", type=int, default=8)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--host-inputs", action="store_true",
                   help="hand the engine HOST buffers in the main run (PCIe-inclusive rate; never the headline value)")
    p.add_argument("--pcie-frames", type=int, default=48,
                   help="timed frames of each run of the two PCIe-inclusive legs (three runs each, the median is reported; 0: skip them)")
    p.add_argument("--no-extras", action="store_true", help="skip the cold run and the live per-frame latency measurement")
    p.add_argument("--shared-scene", action="store_true",
                   help="all objects look at ONE camera stream (SURVEY 8e ii): rank 0 ingests depth + flow and broadcasts every batch "
                        "of frames to the other ranks inside the timed region (RCCL); never the headline configuration")
    p.add_argument("--dump-rows", default="", help="rank 0 saves the gathered [objects, steps, 19] result rows of the timed region (.npy)")
    p.add_argument("--rehearsal-ms", type=float, default=400.0,
                   help="process warm-up before the W warm-up steps: scratch engines track the same W + K frames (untimed, results "
                        "discarded) for at least this long -- kernels loaded, allocations sized, input pages touched by the "
                        "tracker's own access pattern, device out of its idle power state under the tracker's own load.  The "
                        "timed K steps then run on a fresh engine.  0: off (see also `value_cold` in the output)")
    p.add_argument("--clock-warm-ms", type=float, default=0.0,
                   help="milliseconds of unrelated GPU work (GEMMs + copies) before the warm-up steps (round 2's warm-up; off)")
    p.add_argument("--no-align", action="store_true",
                   help="cut the frames into full batches wherever they fall instead of ending every batch with a pose-arrival frame")
    p.add_argument("--no-ramp", action="store_true", help="(with --no-align) full batches from the first timed frame on, the short one last")
    p.add_argument("--splits", default="", help="explicit batch sizes of the timed frames, e.g. 2,6,6,6 (must sum to --steps)")
    p.add_argument("--mask-workgroups", type=int, default=0,
                   help="roft_config::mask_workgroups_per_object (0: the engine's choice, (CUs - CUs / 4) / objects)")
    p.add_argument("--outlier-bands", type=int, default=0,
                   help="roft_config::outlier_bands_per_alternative (0: the engine's choice, CUs / (2 x objects))")
    p.add_argument("--no-kernel-timing", action="store_true",
                   help="skip the instrumented window (roofline kernel's event pair) and the per-kernel breakdown behind it")
    p.add_argument("--windows", type=int, default=5,
                   help="timed windows per invocation (each: fresh engine, the streams generated anew into fresh buffers, W warm-up + K timed steps); "
                        "`value` is the median window")
    return p.parse_args()


def split_batches(first, last, T, ramp=False):
    """[first, last) in consecutive batches of at most T frames.  ramp: the pipeline is empty at `first` -- the batch
    that is not full ((last - first) mod T frames) goes first, so that the velocity and pose chains start after a short
    batch's image chains instead of a whole batch's; measured against 1, 2, 4, ... ramps, which cost more in launches
    than they save in latency."""
    out = []
    k = first
    r = (last - first) % T
    if ramp and r:
        out.append((k, r))
        k += r
    while k < last:
        t = min(T, last - k)
        out.append((k, t))
        k += t
    return out


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process (torch.distributed.run, one rank
    per GPU, rendezvous on 127.0.0.1) before this process has imported torch or touched a GPU, relay the child's output
    (rank 0's JSON line) and exit with its code -- never exec.  Under a launcher (WORLD_SIZE set) the world size must be
    the one --gpus states: a silent mismatch would print a line whose n_gpus is not what the caller asked for."""
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != args.gpus:
            sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks; start it with "
                             "--nproc-per-node %d (or drop the launcher: `python bench.py --gpus %d` starts the ranks itself)\n"
                             % (args.gpus, ws, args.gpus, args.gpus))
            raise SystemExit(2)
        return
    if args.gpus <= 1:
        return
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    child = subprocess.Popen(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))
    try:
        rc = child.wait()
    except KeyboardInterrupt:
        child.terminate()
        rc = child.wait()
    raise SystemExit(rc)


def main():
    args = parse()
    self_launch(args, sys.argv[1:])
    import torch
    import torch.distributed as dist
    from roft_amd import _lib as L
    from roft_amd import engine as E
    from roft_amd import metrics, parallel, synth

    rank, local_rank, world = parallel.env_rank()
    # ROFT_BENCH_DEVICE / ROFT_BENCH_BACKEND exist only to exercise the N > 1 code path on a one-GPU box
    # (several ranks on cuda:0 over gloo); the driver's multi-GPU runs use LOCAL_RANK and RCCL.
    dev_index = int(os.environ.get("ROFT_BENCH_DEVICE", local_rank))
    if "ROFT_BENCH_DEVICE" in os.environ and world > 1:
        # several PROCESSES on one GPU: an engine counts the CUs its early pose lanes may occupy over its own objects only
        # (engine_step.hip, early_lanes) -- off, the lanes then wait for velocity-filter workgroups that are resident
        os.environ.setdefault("ROFT_EARLY_LANES", "0")
    backend = os.environ.get("ROFT_BENCH_BACKEND", "nccl")
    if dev_index >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d needs GPU %d, this node shows %d (one rank per GPU; ROFT_BENCH_DEVICE=0 "
                         "ROFT_BENCH_BACKEND=gloo puts every rank on GPU 0 to exercise the N > 1 path on a one-GPU box)"
                         % (rank, dev_index, torch.cuda.device_count()))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if backend == "nccl":
        parallel.init("nccl")   # RCCL: barrier, max-over-ranks time, all-gather of the result rows
        red_dev = dev
    else:
        parallel.init(backend)
        red_dev = "cpu"
    local_rank = dev_index

    T = max(1, min(args.batch, L.MAX_BATCH_FRAMES))
    if args.scaling == "strong":
        total_obj = args.objects
        my_objects = parallel.shard_objects(total_obj, rank, world)
    else:
        total_obj = args.objects * world
        my_objects = parallel.weak_objects(args.objects, rank)
    n_obj = len(my_objects)
    if n_obj == 0:
        raise SystemExit("bench.py: rank %d owns no object (%d objects over %d ranks)" % (rank, total_obj, world))
    n_windows = max(1, args.windows)
    n_extra = 0 if args.no_kernel_timing else int(os.environ.get("ROFT_BENCH_EXTRA_FRAMES", "24"))   # frames behind the instrumented window: per-kernel breakdown
    n_timed_end = args.warmup + args.steps
    n_frames = n_timed_end   # frames of a window's stream set (the instrumented set carries n_extra more)
    cam = synth.Camera.shape_a() if args.shape == "A" else synth.Camera.shape_b()
    ftype = synth.FLOW_F32C2 if args.flow == "f32" else synth.FLOW_S16C2

    # The streams of a window stay resident in HBM while it runs (depth 4 B + mask 1 B per pixel, flow per grid cell): refuse a
    # K + W that cannot fit instead of running the box out of memory.
    g = 1 if args.flow == "f32" else 4
    flow_frame_bytes = (cam.width // g) * (cam.height // g) * (8 if args.flow == "f32" else 4)
    per_frame = cam.width * cam.height * 5 + flow_frame_bytes
    # (alive at once: window 0's set, kept for the accuracy figures, + the running window's -- or the instrumented one with its
    #  n_extra frames; the rehearsal's set is freed before the first window)
    need = per_frame * (2 * n_timed_end + n_extra) * n_obj
    free_b, _total_b = torch.cuda.mem_get_info(dev)
    if need > 0.8 * free_b:
        raise SystemExit("bench.py: %d frames x %d objects of synthetic input (two sets alive) need %.0f GB of HBM, %.0f GB are free; "
                         "lower --steps / --warmup / --objects" % (n_timed_end, n_obj, need / 1e9, free_b / 1e9))

    # ---- synthetic streams, generated on the GPU and left resident in HBM
    gen_s = [0.0]

    def make_streams(seed_base, n_fr, zero_remote=True):
        """The rank's objects' streams: stream seed = seed_base + global object index (seed_base 4000 = 1000 x config #4:
        the workload; other bases: same shapes and object models, different motion, noise and images)."""
        t_g = time.time()
        if args.shared_scene:
            # one scene for every object of every rank: all ranks build the same stream (masks, poses and the mesh stay
            # local), but only the ingest rank keeps its images -- the others receive them batch by batch (broadcast_frames)
            sc = synth.make_stream(seed_base, n_fr, cam, flow_type=ftype, device=dev)
            if world > 1 and rank != 0 and zero_remote:
                sc.depth.zero_()
                sc.flow.zero_()
            torch.cuda.synchronize()
            gen_s[0] += time.time() - t_g
            return [sc] * n_obj, sc
        out = []
        for gid in my_objects:
            scale = 0.8 + 0.4 * (((gid % 64) * 7) % 10) / 9.0
            half = tuple(h * scale for h in synth.CRACKER_BOX_HALF_EXTENTS)
            out.append(synth.make_stream(seed_base + gid, n_fr, cam, flow_type=ftype, half_extents=half, device=dev))
        torch.cuda.synchronize()
        gen_s[0] += time.time() - t_g
        return out, None

    def new_engine(max_objects):
        cfg = E.default_config(cam.width, cam.height, ftype, max_objects=max_objects, device=local_rank, max_batch_frames=T)
        cfg.cam.fx, cfg.cam.fy, cfg.cam.cx, cfg.cam.cy = cam.fx, cam.fy, cam.cx, cam.cy
        cfg.mask_workgroups_per_object = args.mask_workgroups
        cfg.outlier_bands_per_alternative = args.outlier_bands
        return cfg, E.ROFTFilterBatch(cfg)

    def add_objects(eng, sts):
        for st in sts:
            d = E.default_object()
            m0 = synth.initial_pose_from_stream(st)
            for i in range(13):
                d.p_mean0[i] = m0[i]
            eng.add_object(d, *st.mesh)

    def frame_dict(st, k, src, kind):
        mi = st.mask_delivery[k]
        pose = (st.pose_meas[k, :3], st.pose_meas[k, 3:]) if st.pose_valid[k] else None
        return dict(depth=src["depth"][k].data_ptr(), flow=src["flow"][k].data_ptr() if st.flow_valid[k] else None,
                    mask=src["mask"][mi].data_ptr() if mi >= 0 else None, pose=pose, dt=st.dt, mem_kind=kind)

    def host_copies(sts):   # pinned host copies of the streams: the boundary then pays the PCIe transfer
        if not args.host_inputs:
            return None
        seen = {}
        for st in sts:
            if id(st) not in seen:
                seen[id(st)] = dict(depth=st.depth.cpu().pin_memory(), flow=st.flow.cpu().pin_memory(), mask=st.mask_gt.cpu().pin_memory())
        return [seen[id(st)] for st in sts]

    def build(eng, k0, t, sts, hst=None):
        frames_list = []
        for k in range(k0, k0 + t):
            frames_list.append([frame_dict(st, k, hst[o] if hst else dict(depth=st.depth, flow=st.flow, mask=st.mask_gt),
                                           L.MEM_HOST if hst else L.MEM_DEVICE) for o, st in enumerate(sts)])
        return eng.build_batch(frames_list)

    cfg0 = E.default_config(cam.width, cam.height, ftype)
    period = int(cfg0.pose_frames_between)

    def plan_batches(first, last, ramp=False):
        if args.no_align or period <= 0:
            return split_batches(first, last, T, ramp=ramp)
        return E.aligned_batches(first, last, T, period)

    warm_splits = plan_batches(0, args.warmup)
    if args.splits:
        sizes = [int(x) for x in args.splits.split(",")]
        if sum(sizes) != args.steps or max(sizes) > T or min(sizes) < 1:
            raise SystemExit("bench.py: --splits must sum to --steps with entries in 1..--batch")
        starts = np.cumsum([args.warmup] + sizes[:-1])
        timed_splits = list(zip([int(x) for x in starts], sizes))
    else:
        timed_splits = plan_batches(args.warmup, n_timed_end, ramp=not args.no_ramp)
    extra_splits = plan_batches(n_timed_end, n_timed_end + n_extra)

    bcast_bytes = [0]

    def run(eng, batches, splits=None, scene=None):
        for i, (arr, _keep, t) in enumerate(batches):
            if scene is not None and world > 1 and splits is not None:
                k0 = splits[i][0]
                frames = [scene.depth[k0:k0 + t], scene.flow[k0:k0 + t]]
                if backend == "nccl":
                    parallel.broadcast_frames(frames, src=0)
                else:   # (gloo, the one-GPU exercise of this path: staged through the host)
                    host_frames = [f.cpu() for f in frames]
                    parallel.broadcast_frames(host_frames, src=0)
                    for f, h in zip(frames, host_frames):
                        f.copy_(h)
                torch.cuda.synchronize()   # the engine's streams are not torch's: the frames must have landed
                bcast_bytes[0] += sum(f.numel() * f.element_size() for f in frames)
            eng.submit_batch_raw(arr, t)
            eng.step()

    barrier = parallel.barrier
    if args.rehearsal_ms > 0:
        # Process warm-up, once, before the first window: scratch engines track a DISJOINT stream set (seeds 20000 +) of the
        # same shapes and batch cuts -- kernels loaded, allocations sized, the device out of its idle power state under the
        # tracker's own load.  No frame of any window is read by it.
        reh_streams, _rs = make_streams(20000, n_timed_end, zero_remote=False)   # (nothing broadcasts in the rehearsal)
        reh_host = host_copies(reh_streams)
        t_w = time.perf_counter()
        while True:
            _c, scratch = new_engine(n_obj)
            add_objects(scratch, reh_streams)
            for k0, t in warm_splits + timed_splits:
                arr, _keep, tt = build(scratch, k0, t, reh_streams, reh_host)
                scratch.submit_batch_raw(arr, tt)
                scratch.step()
            scratch.sync()
            scratch.close()
            if (time.perf_counter() - t_w) * 1e3 >= args.rehearsal_ms:
                break
        del reh_streams, reh_host
    if args.clock_warm_ms > 0:
        # not tracker work and not timed: brings the device out of its idle power state (see --clock-warm-ms)
        wa = torch.randn(4096, 4096, device=dev, dtype=torch.float32)
        wb = torch.empty(256 << 20, device=dev, dtype=torch.uint8)
        wc = torch.empty_like(wb)
        t_w = time.perf_counter()
        while (time.perf_counter() - t_w) * 1e3 < args.clock_warm_ms:
            for _ in range(4):
                wa = (wa @ wa).mul_(1.0 / 4096.0)
                wc.copy_(wb)
            torch.cuda.synchronize()
        del wa, wb, wc
    gplan = None
    if world > 1:
        # Nothing of the exchange happens for the first time inside a window: the shard sizes are exchanged here, once, and
        # one all-gather of the timed region's shape (and, for a shared scene, one broadcast of a batch's shape) runs before
        # any clock starts -- the first collective of a process builds its communicator's channels and can cost more than
        # a whole window.
        gplan = parallel.gather_plan(n_obj, (args.steps, 19), red_dev)
        parallel.gather_records(torch.zeros((n_obj, args.steps, 19), dtype=torch.float64), red_dev, gplan)
        if args.shared_scene:
            t_max = max(t for _k0, t in timed_splits)
            dummy = [torch.zeros((t_max, cam.height, cam.width), dtype=torch.float32, device=dev),
                     torch.zeros((t_max, cam.height // g, cam.width // g, 2), dtype=torch.float32 if args.flow == "f32" else torch.int16, device=dev)]
            if backend != "nccl":
                dummy = [d.cpu() for d in dummy]
            parallel.broadcast_frames(dummy, src=0)
            del dummy
        torch.cuda.synchronize()

    def host_spin():
        # a few milliseconds of spinning bring the host thread's core to its working clock before the window in which it enqueues
        # ~70 launches in a millisecond (one run in twenty showed the host side of those launches four to five times slower)
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < 3e-3:
            pass

    def timed_window(seed_base, instrument=0, extra=0):
        """One timed region on a fresh engine and a fresh stream set: W warm-up steps, barrier + synchronize, EXACTLY K timed
        steps, engine sync (+ the all-gather of the result rows at N > 1), synchronize + barrier, max over ranks.  instrument:
        roft_engine_enable_timing level (0 for every window that can become `value`).  Returns the record of the window and
        what the caller keeps (engine, streams, gathered rows)."""
        sts, scene = make_streams(seed_base, n_timed_end + extra)
        hst = host_copies(sts)
        cfg, eng = new_engine(n_obj)
        add_objects(eng, sts)
        eng.enable_log(n_timed_end + extra)
        warm_b = [build(eng, k0, t, sts, hst) for k0, t in warm_splits]
        timed_b = [build(eng, k0, t, sts, hst) for k0, t in timed_splits]
        if instrument:
            eng.enable_timing(instrument)
        barrier()
        host_spin()
        run(eng, warm_b, warm_splits, scene)
        eng.sync()
        torch.cuda.synchronize()
        if instrument:
            # (the marks of the warm-up steps are not the timed region's -- nor part of a ROFT_DUMP_MARKS timeline)
            dump_path = os.environ.pop("ROFT_DUMP_MARKS", None)
            eng.timing()
            if dump_path:
                os.environ["ROFT_DUMP_MARKS"] = dump_path
        bcast_bytes[0] = 0
        stats0 = eng.stats()
        nb0 = stats0["batches"]
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t0_us = time.clock_gettime(time.CLOCK_MONOTONIC) * 1e6   # (the engine's batch trace is on this clock: std::chrono::steady_clock)
        run(eng, timed_b, timed_splits, scene)
        host_enqueue = time.perf_counter() - t0   # host side of the loop (frame programs + launches), GPU still running
        host_cpu = os.sched_getcpu() if hasattr(os, "sched_getcpu") else -1
        eng.sync()
        gathered = None
        if world > 1:
            # the only exchange of the job: every rank's per-object result rows (pose 13 | twist 6 per object-frame)
            rows = torch.from_numpy(eng.get_log_rows(args.warmup, args.steps)).transpose(0, 1).contiguous()   # [obj, frame, 19]
            gathered = parallel.gather_records(rows, red_dev, gplan)
        torch.cuda.synchronize()
        barrier()
        elapsed_own = time.perf_counter() - t0
        elapsed = parallel.max_over_ranks(elapsed_own, red_dev)
        stats1 = eng.stats()
        trace = [b for b in eng.batch_trace() if b["batch"] >= nb0]
        rec = {
            "value": total_obj * args.steps / elapsed,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "elapsed_ms": 1e3 * elapsed,
            "elapsed_ms_this_rank": 1e3 * elapsed_own,
            "host_enqueue_ms_per_step": 1e3 * host_enqueue / args.steps,
            "host_cpu": host_cpu,
            "loadavg_1min": os.getloadavg()[0],
            "seed_base": seed_base,
            "launches_per_frame": (stats1["launches"] - stats0["launches"]) / max(stats1["frames"] - stats0["frames"], 1),
            "event_ops_per_frame": (stats1["event_ops"] - stats0["event_ops"]) / max(stats1["frames"] - stats0["frames"], 1),
            # per batch of the window: the engine's scheduling decisions (a function of the batch index), the host's time in
            # the submit and step calls, and when the host saw the batch complete, relative to the start of the window
            "batches": [dict(frames=b["frames"], steady=b["steady"], throttled=b["throttled"], handoff=b["handoff"],
                             early_lanes=b["early_lanes"], outlier_parts_halved=b["outlier_parts_halved"], launches=b["launches"],
                             submit_us=round(b["submit_us"], 1), wait_us=round(b["wait_us"], 1), step_us=round(b["step_us"], 1),
                             submitted_at_ms=round((b["t_submit_us"] - t0_us) * 1e-3, 4),
                             done_at_ms=round((b["t_done_us"] - t0_us) * 1e-3, 4) if b["t_done_us"] else None) for b in trace],
        }
        return rec, dict(eng=eng, streams=sts, scene=scene, host=hst, gathered=gathered, cfg=cfg, bcast=bcast_bytes[0])

    # ---- the timed windows
    windows = []
    keep0 = None
    for w in range(n_windows):
        # Every window tracks the CANONICAL streams of config #4 (seeds 4000 + object index), generated anew into fresh
        # buffers: no frame has been read by anything when its window reads it, and all windows measure the SAME workload -- the
        # spread of `runs` is the machine's, not the data's.  (Round 5 measured windows on other seeds first: the same generator
        # with seeds 5000 + / 6000 + tracks 15 % slower at 64 objects and 40 % slower at 16 -- different motion, different work --,
        # which made the windows incomparable.  ROFT_BENCH_WINDOW_SEEDS=1: window i on seeds 4000 + 1000 i.)
        rec, keep = timed_window(4000 + (1000 * w if os.environ.get("ROFT_BENCH_WINDOW_SEEDS") == "1" else 0))
        windows.append(rec)
        if w == 0:
            # window 0 tracks the canonical streams (seeds 4000 +): accuracy figures, --dump-rows.  Its results are read now and the
            # engine is closed like the others' -- a tracker whose run is over does not sit next to the next one
            keep0 = keep
            pose_log, twist_log, npts_log, sel_log = keep["eng"].get_log(0, n_timed_end)
            rows0 = None if keep["gathered"] is not None else np.ascontiguousarray(keep["eng"].get_log_rows(args.warmup, args.steps).transpose(1, 0, 2))
        keep["eng"].close()
        if w != 0:
            del keep
    order = sorted(range(n_windows), key=lambda i: windows[i]["value"])
    med = windows[order[(n_windows - 1) // 2]]   # the median window (the lower one of an even count): a run that happened
    value = med["value"]
    streams, scene, host, gathered, cfg = (keep0[k] for k in ("streams", "scene", "host", "gathered", "cfg"))
    bcast_total = keep0["bcast"]

    # ---- one more window of the same shape WITH the roofline kernel's event pair (never `value`), and behind it the
    #      per-kernel breakdown over the next frames of the same streams (a marker event after every launch group)
    kernels = {}
    k1_live = None
    inst = None
    if not args.no_kernel_timing:
        # (the window holds four launches of the roofline kernel, each next to whatever the other chains are doing at that moment:
        #  24 - 47 us between invocations in round 5.  --k1-windows instrumented windows, fresh engine and streams each, pool their
        #  launches; the last one carries the breakdown frames)
        full_marks = os.environ.get("ROFT_BENCH_FULL_TIMING") == "1"
        n_inst = 1 if full_marks else max(1, args.k1_windows)
        inst_seed = 4000 + (1000 * n_windows if os.environ.get("ROFT_BENCH_WINDOW_SEEDS") == "1" else 0)
        k1_ms = k1_cnt = span_ms = span_cnt = 0
        inst_values = []
        for wi in range(n_inst):
            last = wi == n_inst - 1
            inst, keep_i = timed_window(inst_seed, instrument=(2 if full_marks else 1), extra=n_extra if last else 0)
            e_i = keep_i["eng"]
            tm = e_i.timing()
            # the kernel that takes the flow measurements: the mask + measurement chain (one launch per batch), or -- images too large
            # for it, ROFT_MASK_FUSED=0 -- the stand-alone measurement kernel
            k1_name = "mask_flow_chain" if "mask_flow_chain" in tm else "flow_measure"
            k1_ms += tm[k1_name][0]
            k1_cnt += tm[k1_name][1]
            if k1_name + "_span" in tm:
                span_ms += tm[k1_name + "_span"][0]
                span_cnt += tm[k1_name + "_span"][1]
            inst_values.append(inst["value"])
            if not last:
                e_i.close()
                del keep_i
        inst["values_of_all_instrumented_windows"] = inst_values
        k1_live = dict(total_ms=k1_ms, launches=k1_cnt, avg_us=1e3 * k1_ms / max(k1_cnt, 1), kernel=k1_name, instrumented_windows=n_inst)
        if span_cnt:
            # the same launches on the device's own 100 MHz clock: first workgroup in -> last workgroup out
            k1_live["span_avg_us"] = 1e3 * span_ms / span_cnt
            k1_live["span_launches"] = span_cnt
        if n_extra > 0:
            e_i.enable_timing(2)
            extra_b = [build(e_i, k0, t, keep_i["streams"], keep_i["host"]) for k0, t in extra_splits]
            run(e_i, extra_b, extra_splits, keep_i["scene"])
            e_i.sync()
            for name, (ms, cnt) in e_i.timing().items():
                if not name.endswith("_span"):
                    kernels[name] = dict(total_ms=ms, marks=cnt, avg_us=1e3 * ms / max(cnt, 1))
            e_i.enable_timing(0)
        npts_inst = e_i.get_log(0, n_timed_end)[2]
        inst_streams = keep_i["streams"]
        e_i.close()
        del keep_i

    # ---- accuracy of window 0 (the canonical streams): ADD-S vs ground truth for EVERY object of every rank, ADD-S vs the CPU
    #      reference path on a sample that holds objects of every rank
    pts_cache = {}

    def model_points(st):
        # 500 model points per object model, drawn with a seed of the model's own (every rank, whatever objects it holds and in
        # whatever order it asks, scores a model on the same points)
        key = st.half_extents
        if key not in pts_cache:
            v = st.mesh[0].astype(np.float64)
            seed = int(sum(round(h * 1e6) * (i + 1) for i, h in enumerate(key))) & 0x7FFFFFFF
            pts_cache[key] = v[np.random.default_rng(seed).choice(len(v), 500, replace=False)]
        return pts_cache[key]

    sl = slice(args.warmup, n_timed_end)
    adds_gt_local = []
    for o in range(n_obj):
        st = streams[o]
        est = np.concatenate([pose_log[:, o, 6:9], pose_log[:, o, 9:13]], 1)
        gt = np.concatenate([st.gt.x, st.gt.q], 1)
        adds_gt_local.append(metrics.trajectory_adds(est[sl], gt[sl], model_points(st)))
    n_sample = min(args.cpu_sample_objects, n_obj)
    # ADD-S vs the CPU path at N > 1: every rank runs the oracle on its first ceil(sample / world) objects (25 frames of one
    # object are 40 ms of one core) -- untimed: `cpu_baseline` is an N = 1 measurement
    adds_cpu_local = None
    if world > 1 and not args.no_cpu_baseline:
        from oracle import binding as ob
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import util
        adds_cpu_local = []
        for o in range(min(n_obj, max(1, -(-args.cpu_sample_objects // world)))):
            st = streams[o]
            ref = util.run_oracle_tracker(ob, st, n_timed_end)
            ref_pose = np.stack([np.concatenate([r["pose"][6:9], r["pose"][9:13]]) for r in ref])
            est = np.concatenate([pose_log[:, o, 6:9], pose_log[:, o, 9:13]], 1)
            adds_cpu_local.append(metrics.trajectory_adds(est[:n_timed_end], ref_pose, model_points(st)))
    ranks_info = None
    if world > 1:
        props = torch.cuda.get_device_properties(dev)
        mine = dict(rank=rank, local_rank=local_rank, device=dev_index, device_name=props.name,
                    device_uuid=str(getattr(props, "uuid", "")), objects=n_obj, first_object=int(my_objects[0]),
                    adds_gt=[a.tolist() for a in adds_gt_local],
                    adds_cpu=[a.tolist() for a in adds_cpu_local] if adds_cpu_local is not None else None,
                    windows_ms_this_rank=[w_["elapsed_ms_this_rank"] for w_ in windows])
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        seen = torch.ones(1, device=red_dev)
        dist.all_reduce(seen)   # one contribution per rank that is really in the communicator
        ranks_info = dict(all=allr, seen=int(seen.item()))

    if rank != 0:
        if world > 1:
            dist.barrier()  # rank 0 is timing the CPU baseline and the PCIe legs
            dist.destroy_process_group()
        return

    if gathered is not None:
        assert tuple(gathered.shape) == (total_obj, args.steps, 19), gathered.shape
    if args.dump_rows:
        rows_out = gathered.cpu().numpy() if gathered is not None else rows0
        np.save(args.dump_rows, rows_out)
    if world > 1:
        # cpu_baseline, the PCIe-inclusive legs and the extras are N = 1 measurements (the other ranks wait at a barrier)
        args.no_cpu_baseline = True
        args.pcie_frames = 0
        args.no_extras = True

    if ranks_info is not None:
        adds_gt = np.concatenate([np.asarray(a) for r in ranks_info["all"] for a in r["adds_gt"]])
        adds_gt_objects = sum(len(r["adds_gt"]) for r in ranks_info["all"])
    else:
        adds_gt = np.concatenate(adds_gt_local)
        adds_gt_objects = n_obj
    # RMSE metrics of evaluation/metrics.py on this rank's sample (position cm, orientation deg, velocities with the
    # pole moved to the object, evaluate.py:514-521)
    est_x = np.concatenate([pose_log[sl, o, 6:9] for o in range(n_sample)])
    est_q = np.concatenate([pose_log[sl, o, 9:13] for o in range(n_sample)])
    gt_x = np.concatenate([streams[o].gt.x[sl] for o in range(n_sample)])
    gt_q = np.concatenate([streams[o].gt.q[sl] for o in range(n_sample)])
    est_tw = metrics.object_velocity_from_twist(np.concatenate([twist_log[sl, o] for o in range(n_sample)]), gt_x)
    gt_tw = metrics.object_velocity_from_twist(np.concatenate([streams[o].gt.twist[sl] for o in range(n_sample)]), gt_x)
    rmse = {"position_cm": metrics.rmse_cartesian_3d(gt_x, est_x), "orientation_deg": metrics.rmse_angular(gt_q, est_q),
            "linear_velocity_cm_s": metrics.rmse_linear_velocity(gt_tw[:, :3], est_tw[:, :3]),
            "angular_velocity_deg_s": metrics.rmse_angular_velocity(gt_tw[:, 3:], est_tw[:, 3:])}
    cpu = None
    cpu_multi = None
    adds_cpu = None
    if not args.no_cpu_baseline:
        # CPU baseline: the oracle's ROFTFilter restatement on host cores, one object after the other on
        # ONE core (the reference runs one compute thread per tracker process, main.cpp:421-424).
        # Frame loading/generation is excluded like the reference does (ROFTFilter.cpp:267-270,372-384).
        from oracle import binding as ob
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import util
        cpu_time = 0.0
        cpu_frames = 0
        dists = []
        for o in range(n_sample):
            st = streams[o]
            ocfg = util.oracle_config(ob, st)
            trk = ob.Tracker(ocfg, *st.mesh)
            depth = st.depth.cpu().numpy()
            flow = st.flow.cpu().numpy()
            masks = st.mask_gt.cpu().numpy()
            ref_pose = np.zeros((n_frames, 7))
            for k in range(n_frames):
                mi = st.mask_delivery[k]
                pose = (st.pose_meas[k, :3], st.pose_meas[k, 3:]) if st.pose_valid[k] else None
                t1 = time.perf_counter()
                r = trk.step(st.dt, depth[k], flow[k] if st.flow_valid[k] else None, masks[mi] if mi >= 0 else None, pose)
                cpu_time += time.perf_counter() - t1
                cpu_frames += 1
                ref_pose[k, :3] = r.pose[6:9]
                ref_pose[k, 3:] = r.pose[9:13]
            trk.close()
            est = np.concatenate([pose_log[:, o, 6:9], pose_log[:, o, 9:13]], 1)
            dists.append(metrics.trajectory_adds(est, ref_pose, model_points(st)))
        adds_cpu = np.concatenate(dists)
        cpu = dict(value=cpu_frames / cpu_time, unit="object-frames/s", cores=1, kind="port",
                   sample="%d objects x %d frames of the same %dx%d streams, oracle/ ROFTFilter restatement "
                          "(gcc -O2, incl. CPU rasteriser), sequential on one host core; host: %d cores" %
                          (n_sample, n_frames, cam.width, cam.height, os.cpu_count()),
                   cpu_model=cpu_model(), ms_per_object_frame=1e3 * cpu_time / cpu_frames)
        try:
            cpu_multi = cpu_baseline_multicore(streams, n_sample, min(n_frames, 36), total_obj)
        except Exception as ex:   # the 1-core figure stands on its own
            cpu_multi = dict(error=str(ex)[:300])

    # ---- PCIe-inclusive rates: the same tracker fed with pinned HOST buffers (upload inside the submit call).
    #      (i) one depth + flow + mask stream per object; (ii) the shared-scene form of config #4: every
    #      object points at the same depth and flow image, which the engine uploads once per frame.
    pcie = None
    # (on the longest stream set of the run: the instrumented window's, W + K + 24 frames, else window 0's)
    long_streams = inst_streams if (inst is not None and not args.shared_scene) else streams
    n_long = int(long_streams[0].depth.shape[0])
    if args.pcie_frames > 0 and n_long >= T + 2:
        def pcie_leg(shared, in_place=False):
            n_run = min(n_long, T + args.pcie_frames)
            sts = [long_streams[0]] * n_obj if shared else long_streams
            src = {}
            for st in (sts[:1] if shared else sts):
                src[id(st)] = dict(depth=st.depth[:n_run].cpu().pin_memory(), flow=st.flow[:n_run].cpu().pin_memory(),
                                   mask=st.mask_gt[:n_run].cpu().pin_memory())
            # shared scene: the masks stay per object (distinct host buffers -> one upload per object and mask frame)
            own_masks = [st.mask_gt[:n_run].cpu().pin_memory() for st in sts] if shared else None
            _c, e2 = new_engine(n_obj)
            add_objects(e2, sts)
            batches = []
            for k0, t in split_batches(0, n_run, T):
                fl = []
                for k in range(k0, k0 + t):
                    row = []
                    for o, st in enumerate(sts):
                        s = dict(src[id(st)])
                        if own_masks:
                            s["mask"] = own_masks[o]
                        # in_place: the same pinned buffers handed over as ROFT_MEM_DEVICE -- read over the bus where the kernels
                        # touch them, nothing staged (the buffers outlive the run: the retention contract of DEVICE inputs)
                        row.append(frame_dict(st, k, s, L.MEM_DEVICE if in_place else L.MEM_HOST))
                    fl.append(row)
                batches.append(e2.build_batch(fl))
            e2.submit_batch_raw(batches[0][0], batches[0][2])   # first batch: allocations, first touch of the pinned pages
            e2.step()
            e2.sync()
            s0 = e2.stats()
            t1 = time.perf_counter()
            for arr, _keep, t in batches[1:]:
                e2.submit_batch_raw(arr, t)
                e2.step()
            e2.sync()
            dt_ = time.perf_counter() - t1
            s1 = e2.stats()
            e2.close()
            frames = s1["frames"] - s0["frames"]
            return dict(value=n_obj * frames / dt_, unit="object-frames/s", frames=frames, ms_per_step=1e3 * dt_ / frames,
                        h2d_GB_per_s=(s1["h2d_bytes"] - s0["h2d_bytes"]) / dt_ / 1e9,
                        h2d_MB_per_step=(s1["h2d_bytes"] - s0["h2d_bytes"]) / frames / 1e6)
        def median_leg(shared, in_place=False):
            runs = sorted((pcie_leg(shared, in_place) for _ in range(3)), key=lambda r: r["value"])
            med = dict(runs[1])
            med["runs"] = [r["value"] for r in runs]
            return med
        pcie = dict(per_object_streams=median_leg(False), shared_scene=median_leg(True),
                    per_object_streams_in_place=median_leg(False, True), shared_scene_in_place=median_leg(True, True),
                    note="pinned host buffers.  per_object_streams / shared_scene: handed over as ROFT_MEM_HOST, copied to the device by the "
                         "submit call before it returns (every byte crosses the bus); *_in_place: the same buffers handed over as "
                         "ROFT_MEM_DEVICE and read in place -- only the sectors the kernels touch cross the bus, nothing is staged "
                         "(h2d_* are 0 there by construction).  per_object_streams: %d x (depth + flow [+ mask]) per frame; shared_scene: "
                         "one depth + flow for all objects, masks per object; each leg: median of three runs of %d timed frames" % (n_obj, args.pcie_frames))

    # ---- extras: the same timed sequence from a cold device, and the tracker used live (one frame at a time, state read
    #      back before the next frame is submitted)
    value_cold = None
    live = None
    k1_alone = None
    if not args.no_extras:
        cold_host = [0.0]

        def timed_sequence(e2, sts, k_warm, k_timed):
            def bat(k0, t):
                return e2.build_batch([[frame_dict(st, k, dict(depth=st.depth, flow=st.flow, mask=st.mask_gt), L.MEM_DEVICE) for st in sts]
                                       for k in range(k0, k0 + t)])
            wb = [bat(k0, t) for k0, t in warm_splits]
            tb = [bat(k0, t) for k0, t in timed_splits]
            for arr, _keep, t in wb:
                e2.submit_batch_raw(arr, t)
                e2.step()
            e2.sync()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for arr, _keep, t in tb:
                e2.submit_batch_raw(arr, t)
                e2.step()
            cold_host[0] = time.perf_counter() - t1
            e2.sync()
            torch.cuda.synchronize()
            return time.perf_counter() - t1
        # (the device has idled through the CPU baseline and the host work above: no clock warm-up precedes this run)
        # fresh frames as well: a third stream set (seeds + 2000, same shapes) that nothing has read before this run
        cold_streams, _cs = make_streams(4000, n_timed_end, zero_remote=False)   # (the canonical workload, generated anew: comparable with `value`)
        torch.cuda.synchronize()
        time.sleep(1.0)
        _c, e2 = new_engine(n_obj)
        add_objects(e2, cold_streams)
        dt_cold = timed_sequence(e2, cold_streams, args.warmup, args.steps)
        e2.close()
        del cold_streams
        value_cold = dict(value=n_obj * args.steps / dt_cold, ms_per_step=1e3 * dt_cold / args.steps, host_enqueue_ms_per_step=1e3 * cold_host[0] / args.steps,
                          note="a timed sequence of the same shape on a fresh engine and on FRESH frames (the workload's streams generated "
                               "once more into fresh buffers, read by nothing before) after the device has idled (CPU baseline, host work, 1 s sleep), "
                               "no rehearsal of any kind and nothing of this shape run for seconds: what a short burst from an idle GPU gets")
        # the roofline kernel with the device to itself: the same frames, every batch waited for before the next one is
        # submitted (its launch then overlaps nothing but the tail of its own batch's mask chain)
        _c, e3 = new_engine(n_obj)
        add_objects(e3, streams)
        e3.enable_timing(1)
        for k0, t in warm_splits + timed_splits:
            arr, _keep, tt = e3.build_batch([[frame_dict(st, k, dict(depth=st.depth, flow=st.flow, mask=st.mask_gt), L.MEM_DEVICE) for st in streams]
                                             for k in range(k0, k0 + t)])
            e3.submit_batch_raw(arr, tt)
            e3.step()
            e3.sync()
        tm3 = e3.timing()
        ms_alone, n_alone = tm3["mask_flow_chain" if "mask_flow_chain" in tm3 else "flow_measure"]
        e3.close()
        k1_alone = dict(avg_launch_us=1e3 * ms_alone / max(n_alone, 1), launches=n_alone,
                        object_frames_per_launch=n_obj * (args.warmup + args.steps) / max(n_alone, 1))
        # live: ROFTFilter::filtering_step followed by a reader of the estimate, one object, nothing in flight across frames
        n_live = min(n_long, 72)
        cfg1 = E.default_config(cam.width, cam.height, ftype, max_objects=1, device=local_rank, max_batch_frames=1)
        cfg1.cam.fx, cfg1.cam.fy, cfg1.cam.cx, cfg1.cam.cy = cam.fx, cam.fy, cam.cx, cam.cy
        e1 = E.ROFTFilterBatch(cfg1)
        add_objects(e1, long_streams[:1])
        st = long_streams[0]
        ins = [e1.build_inputs([frame_dict(st, k, dict(depth=st.depth, flow=st.flow, mask=st.mask_gt), L.MEM_DEVICE)]) for k in range(n_live)]
        lat = np.zeros(n_live)
        for k in range(n_live):
            t1 = time.perf_counter()
            e1.submit_raw(ins[k][0])
            e1.step()
            e1.state(0)
            lat[k] = time.perf_counter() - t1
        e1.close()
        w = lat[12:] * 1e6
        pf = np.array([bool(st.pose_valid[k]) for k in range(12, n_live)])
        live = dict(median_us=float(np.median(w)), p99_us=float(np.percentile(w, 99)),
                    median_us_pose_frames=float(np.median(w[pf])) if pf.any() else None,
                    median_us_other_frames=float(np.median(w[~pf])) if (~pf).any() else None, frames=int(len(w)),
                    note="one object, %dx%d: submit + step + get_state per frame, nothing in flight across frames (device inputs)" %
                         (cam.width, cam.height))

    # ---- roofline of the masked flow + depth measurement kernel (north_star's target kernel)
    roofline = None
    if k1_live:
        e = 8 if ftype == synth.FLOW_F32C2 else 4
        plane_bytes = cam.width * cam.height // 8
        # algorithmic bytes per object-frame with tile culling declared: the obj bit plane
        # (the whole mask, 1 bit/px) + one depth and one flow sample per candidate + the kept records
        nl = npts_inst[args.warmup:n_timed_end]   # (of the instrumented window: the launches that were timed)
        ran = nl >= 0
        # mask pixels of the frame the measurement reads (the previous frame's propagated mask ~ its ground-truth mask)
        mask_px = np.mean([float((st.mask_gt[max(args.warmup - 1, 0):n_timed_end - 1] > 0).sum().item()) / args.steps for st in inst_streams])
        cand = mask_px / float(int(cfg.subsampling_radius))
        n_kept = float(np.mean(nl[ran])) if ran.any() else 0.0
        # Declared bytes (culled form `mask + rho W H (4 + e / g^2)`, rho = fraction of TILES touched): the tile the
        # memory system moves is the 64-byte sector, so a sampled depth (4 B) and a sampled flow element (8 / 4 B) are declared as
        # the sector each of them lives in -- candidates are 35 mask pixels apart, 140 B of depth and 280 B of flow: no two share
        # a sector -- + the obj bit plane (the whole mask, 1 bit / pixel) + the kept records.  `sample_bytes` is the figure rounds
        # 2 - 4 declared (4 + e bytes per candidate): what an ideal gather engine would move, a third of what any HBM system can.
        sample_bytes_per_obj = plane_bytes + cand * (4 + e) + n_kept * 20
        bytes_per_obj = plane_bytes + cand * (64 + 64) + n_kept * 20
        obj_frames_per_launch = n_obj * args.steps * k1_live.get("instrumented_windows", 1) / max(k1_live["launches"], 1)
        bytes_per_launch = bytes_per_obj * obj_frames_per_launch
        dur_s = k1_live["avg_us"] * 1e-6
        achieved = bytes_per_launch / dur_s / 1e9
        dense_per_obj = cam.width * cam.height * 5 + flow_frame_bytes
        # HBM traffic of this kernel: a separate rocprofv3 --pmc pass of this same command, committed under profiles/
        # (see profiles/README.md for the gfx950 counting caveats); only quoted when the workload matches that pass
        traffic, traffic_raw, traffic_source = None, None, None
        for tag in ("r06", "r05", "r04", "r03", "r02"):
            pmc_path = os.path.join(ROOT, "profiles", "%s_pmc_k1.json" % tag)
            if not os.path.exists(pmc_path):
                continue
            pm = json.load(open(pmc_path))
            if (pm.get("objects"), pm.get("shape"), pm.get("flow"), pm.get("batch")) == (n_obj, args.shape, args.flow, T):
                # raw: the FETCH_SIZE counter as rocprofv3 reports it; corrected: + the bit-plane bytes once more, because
                # on gfx950 FETCH_SIZE tallies the 128-byte requests of a wide coalesced stream at 64 bytes
                # (MI355X_MICROARCH.md, HBM) -- the plane read is such a stream, the 4- and 8-byte gathers are left as counted
                traffic_raw = pm["fetch_bytes_per_object_frame_raw"] * obj_frames_per_launch
                traffic = pm["fetch_bytes_per_object_frame"] * obj_frames_per_launch
                traffic_source = "profiles/%s_pmc_k1.json: %s" % (tag, pm.get("source", ""))
                break
        # the 100 % mark measured on this box next to the datasheet figure: a device-to-device copy of 1 GiB,
        # read + write bytes over the best of five
        copy_gbs = None
        try:
            src_t = torch.empty(1 << 30, device=dev, dtype=torch.uint8)
            dst_t = torch.empty_like(src_t)
            best = None
            for _ in range(6):
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                dst_t.copy_(src_t)
                ev1.record()
                torch.cuda.synchronize()
                ms = ev0.elapsed_time(ev1)
                best = ms if best is None else min(best, ms)
            copy_gbs = 2.0 * (1 << 30) / (best * 1e-3) / 1e9
            del src_t, dst_t
        except Exception:   # (not enough free memory next to the streams: the datasheet figure stands alone)
            copy_gbs = None
        # what this box gives SCATTERED 64-byte sectors (the depth and flow samples of K1 are 35 mask pixels apart: one sector
        # each, whatever the order): roft_debug_sector_rate -- 16 M reads at random sector-aligned offsets of a 2 GiB buffer,
        # best of four launches (tools/micro/sector_probe.hip is the stand-alone version with the skeleton of K1's launch)
        sector_rate = None
        try:
            import ctypes as C
            rate = C.c_double(0.0)
            L.check(L.lib().roft_debug_sector_rate(int(torch.cuda.current_device()), C.byref(rate)))
            sector_rate = float(rate.value)
        except Exception:   # (not enough free memory next to the streams)
            sector_rate = None
        gather_sectors = 2.0 * cand * obj_frames_per_launch       # one depth + one flow sector per candidate
        roofline = dict(kernel="flow_measure_kernel", bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                        measured_copy_GBs=copy_gbs, frac_of_measured_copy=(achieved / copy_gbs) if copy_gbs else None,
                        gather_sectors_per_launch=gather_sectors, gather_Gsectors_per_s=gather_sectors / dur_s / 1e9,
                        measured_random_Gsectors_per_s=(sector_rate / 1e9) if sector_rate else None,
                        frac_of_measured_random_sector_rate=(gather_sectors / dur_s / sector_rate) if sector_rate else None,
                        frac=achieved / HBM_PEAK_GBS, traffic=traffic, traffic_raw_counter=traffic_raw, traffic_source=traffic_source,
                        frac_on_fetched_bytes=(traffic / dur_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                        algorithmic_bytes_per_launch=bytes_per_launch, algorithmic_bytes_per_object_frame=bytes_per_obj,
                        algorithmic_bytes_formula="W H / 8 (obj bit plane) + candidates x (64 + 64) (the depth sector and the flow sector of every "
                                                  "candidate: SURVEY 8d's tiles touched, tile = 64-byte sector) + kept x 20 (records)",
                        traffic_over_algorithmic=(traffic / bytes_per_launch) if traffic else None,
                        frac_on_sample_bytes=sample_bytes_per_obj * obj_frames_per_launch / dur_s / 1e9 / HBM_PEAK_GBS,
                        sample_bytes_per_object_frame=sample_bytes_per_obj,
                        object_frames_per_launch=obj_frames_per_launch, avg_launch_us=k1_live["avg_us"],
                        launches=k1_live["launches"],
                        dense_bytes_equivalent_per_object_frame=dense_per_obj,
                        note="declared bytes: mask bit plane + the 64-byte sectors of the sampled depth / flow values + kept records (SURVEY 8d, "
                             "tiles touched); one launch measures every (frame, object) of a batch; the duration is measured live with a HIP event "
                             "pair on the kernel's own dispatch, in a window of the same shape as the timed ones (instrumented_window), while the other "
                             "chains run on their streams.  The north-star target of 0.70 is not met: the kernel touches ~135 KB per object-frame "
                             "instead of the dense 4 MB, a launch moves ~45 MB, and at that size it is bound by the latency of its three dependent "
                             "round trips (plane -> candidates -> gathers -> records) and by the RATE at which the memory system serves scattered "
                             "sectors (measured_random_Gsectors_per_s: roft_debug_sector_rate on this box, x 64 B = 0.4 of the streaming peak), not "
                             "by streaming bandwidth; frac_on_sample_bytes is the same launch priced on 12 bytes per candidate instead of its two "
                             "sectors (the figure of rounds 2 - 4)")
    if roofline is not None and k1_live is not None and k1_live.get("span_avg_us"):
        span_s = k1_live["span_avg_us"] * 1e-6
        roofline["kernel_span"] = dict(
            avg_us=k1_live["span_avg_us"], launches=k1_live["span_launches"], achieved=bytes_per_launch / span_s / 1e9,
            frac=bytes_per_launch / span_s / 1e9 / HBM_PEAK_GBS,
            note="the same timed launches on the device's own 100 MHz clock (every workgroup leaves its start and end: first "
                 "workgroup in -> last workgroup out), which is the duration a kernel trace reports; the HIP event pair of "
                 "`avg_launch_us` reads 2 - 6 us more per launch (under rocprofv3 too: profiles/README.md). `frac` above stays "
                 "on the event pair")
    if roofline is not None and k1_alone is not None and k1_alone["launches"]:
        roofline["alone"] = dict(k1_alone, frac=bytes_per_obj * k1_alone["object_frames_per_launch"] / (k1_alone["avg_launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                 note="the same kernel with every batch waited for before the next one is submitted (nothing of another "
                                      "batch on the device): what the launch costs by itself; `frac` above is the figure of record")
    # ---- the other kernels of the path, priced (SURVEY 8d "other kernels -- reported"): declared bytes per launch group, the
    #      duration of the group in the 24 frames behind the timed region (HIP event marks, all chains running), GB/s, and the
    #      HBM bytes of the same kernels from the committed --pmc passes (FETCH_SIZE / WRITE_SIZE per dispatch)
    roofline_other = None
    if kernels:
        e_b = 8 if ftype == synth.FLOW_F32C2 else 4
        plane_b = cam.width * cam.height // 8
        mask_px_o = float(np.mean([float((st.mask_gt[:n_timed_end] > 0).sum().item()) / max(n_timed_end, 1) for st in streams]))
        n_feat = mask_px_o / 2.0
        flows_per_frame = (period - 1 + min(period, 6)) / float(period) if period > 0 else 1.0   # one flow per frame, `period` on the frame a mask arrives
        nv = float(np.mean([st.mesh[0].shape[0] for st in streams]))
        nt = float(np.mean([st.mesh[1].shape[0] for st in streams]))
        declared = {
            # source plane (both candidates are fetched: the last propagated mask and the delivered one) + flow at the mask's
            # pixels + the zeroed slot of the frame after + the OR flush (read-modify-write of the touched words)
            "mask_frames": dict(per="object-frame", bytes=2 * plane_b + e_b * mask_px_o * flows_per_frame / (g * g) + plane_b + mask_px_o / 2.0,
                                mark="mask_chain", frames_per_group=float(np.mean([t for _k0, t in extra_splits])) if extra_splits else 1.0),
            # plane + one depth sample per feature + the (pixel, depth) list written
            "features": dict(per="object and pose frame", bytes=plane_b + 4 * n_feat + 8 * n_feat, mark="features", frames_per_group=1.0),
            # per alternative: vertices + triangle indices + the feature list
            "outlier_fused": dict(per="object and test", bytes=2 * (12 * nv + 12 * nt + 8 * n_feat), mark="outlier_render_likelihood", frames_per_group=1.0),
        }
        pmc = {}
        pmc_tag = next((t for t in ("r06", "r05", "r04") if os.path.exists(os.path.join(ROOT, "profiles", "%s_pmc_FETCH_SIZE.csv" % t))), "r04")
        for cname in ("FETCH_SIZE", "WRITE_SIZE"):
            path = os.path.join(ROOT, "profiles", "%s_pmc_%s.csv" % (pmc_tag, cname))
            if os.path.exists(path):
                import csv as _csv
                for r in _csv.DictReader(open(path)):
                    pmc[(r["kernel"].split("::")[-1].split("<")[0], cname)] = (float(r["mean_value_per_dispatch"]) * 1024.0, int(r["dispatches"]))
        kmap = {"mask_frames": "mask_frame_kernel", "features": "features_kernel", "outlier_fused": "outlier_fused_kernel"}
        roofline_other = {}
        for name, dsc in declared.items():
            mk = kernels.get(dsc["mark"])
            ent = dict(declared_bytes=dsc["bytes"], per=dsc["per"])
            if mk and mk["marks"]:
                group_bytes = dsc["bytes"] * n_obj * dsc["frames_per_group"]
                ent.update(avg_us_per_launch_group=mk["avg_us"], launch_groups=mk["marks"], declared_bytes_per_launch_group=group_bytes,
                           achieved_GBs=group_bytes / (mk["avg_us"] * 1e-6) / 1e9, frac_of_hbm_peak=group_bytes / (mk["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS)
            for cname in ("FETCH_SIZE", "WRITE_SIZE"):
                if (kmap[name], cname) in pmc:
                    ent[cname.lower() + "_bytes_per_dispatch"] = pmc[(kmap[name], cname)][0]
            roofline_other[name] = ent
        roofline_other["note"] = ("declared = algorithmic bytes of the kernel (DESIGN.md section 5); durations = HIP event marks around the launch "
                                  "group in the frames behind the timed region, all chains running (mask_frames: the T frame kernels of a batch + "
                                  "mask_general); fetch / write = profiles/%s_pmc_{FETCH,WRITE}_SIZE.csv (separate rocprofv3 --pmc passes, bytes " % pmc_tag +
                                  "per dispatch: one frame of all objects for mask_frame_kernel).  None of these kernels is bandwidth-bound: they "
                                  "are chains of dependent round trips over a few hundred KB per object (latency), the rasteriser is bound by "
                                  "VALU issue on its CU; the CU x us budget of the pipeline is profiles/r04_cu_budget.csv / r05_residency_budget.csv")
    dominant = max(kernels.items(), key=lambda kv: kv[1]["total_ms"])[0] if kernels else None
    vals = sorted(w_["value"] for w_ in windows)
    # accuracy at N > 1 (window 0): ADD-S vs the CPU path over a sample with objects of EVERY rank
    if ranks_info is not None and all(r["adds_cpu"] is not None for r in ranks_info["all"]):
        adds_cpu = np.concatenate([np.asarray(a) for r in ranks_info["all"] for a in r["adds_cpu"]])
        adds_cpu_objects = [len(r["adds_cpu"]) for r in ranks_info["all"]]
    else:
        adds_cpu_objects = [n_sample] if adds_cpu is not None else None

    out = {
        "metric": "tracker frames/sec per object (640x480) + ADD-S vs CPU ref",
        "value": value,
        "unit": "object-frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": med["ms_per_step"],
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        # ---- how `value` came about: every window of this invocation (value = the median one), and the median window's own
        #      host time, host state and batch trace, so that a slow run explains itself
        "value_is": "median of %d timed windows of exactly %d steps each (fresh engine + the workload's streams generated anew into fresh "
                    "buffers per window, no timing instrumentation inside)" % (n_windows, args.steps),
        "runs": [w_["value"] for w_ in windows],
        "value_min": vals[0],
        "value_max": vals[-1],
        "value_first_window": windows[0]["value"],
        "host_enqueue_ms_per_step": med["host_enqueue_ms_per_step"],
        "host_state": {"cpu": med["host_cpu"], "loadavg_1min": med["loadavg_1min"], "cpus_online": os.cpu_count()},
        "batches": med["batches"],
        "windows": windows,
        "instrumented_window": inst,
        "inputs": "host (PCIe-inclusive)" if args.host_inputs else "resident in HBM",
        "timed_frames_first_touch": True,
        "method": {"windows": n_windows, "rehearsal_ms": args.rehearsal_ms,
                   "rehearsal_streams": "disjoint from every window's (seeds 20000 +, same shapes and batch cuts), once, before the first window" if args.rehearsal_ms > 0 else None,
                   "window_streams": "every window (and the instrumented one): the canonical streams of config #4, seeds 4000 + object index "
                                     "(SURVEY 8d), generated anew into fresh buffers -- first touch by the window, same workload in every window",
                   "host_spin_ms": 3.0, "host_spin": "before the W warm-up steps of every window (the window follows them behind one barrier + synchronize)",
                   "batch_cuts": "explicit --splits" if args.splits else ("full batches" if args.no_align else "batches end with the pose-arrival frame"),
                   "scheduling": "a function of the batch index since the engine was last idle (roft_batch_trace::steady), not of host timing",
                   "note": "no frame of a window is read by anything in this process before the window reads it; the W warm-up steps "
                           "run on the window's engine right before its timed steps; nothing records an event inside a window"},
        "config": {"workload": "BASELINE config #4: %dx%d, %s flow grid %d, %d objects in total, %s "
                               "(sharded by object, no data-path collective; result rows all-gathered over RCCL at N > 1), "
                               "masks+poses at 5 fps with 6-frame delay, flow-aided masks, re-sync and outlier rejection on, "
                               "frames submitted in batches of at most %d" %
                               (cam.width, cam.height, "CV_32FC2" if ftype == synth.FLOW_F32C2 else "CV_16SC2",
                                cfg.flow_grid, total_obj,
                                "%d per GPU" % n_obj if args.scaling == "strong" else "%d per GPU (weak scaling)" % n_obj, T),
                   "objects_per_gpu": n_obj, "objects_total": total_obj, "width": cam.width, "height": cam.height,
                   "batch_frames": T, "timed_batches": [t for _k0, t in timed_splits], "ranks": world, "backend": backend},
        # ---- N > 1: who was there (one entry per rank of the communicator), what each owned, and each rank's OWN time per window
        #      (before the max over ranks) -- a SCALE line can be checked against the per-GPU points of the object sweep
        "ranks": ({"world_size": world, "ranks_seen_by_all_reduce": ranks_info["seen"],
                   "objects_per_gpu": [r["objects"] for r in ranks_info["all"]],
                   "first_object_of_rank": [r["first_object"] for r in ranks_info["all"]],
                   "devices": [dict(rank=r["rank"], local_rank=r["local_rank"], device=r["device"], name=r["device_name"], uuid=r["device_uuid"]) for r in ranks_info["all"]],
                   "window_ms_per_rank": [r["windows_ms_this_rank"] for r in ranks_info["all"]],
                   "per_gpu_rate_of_the_median_window": value / world,
                   "note": "per_gpu_rate = value / n_gpus: what ONE GPU tracks with objects_per_gpu objects -- compare with the N = 1 run at --objects <that number> (profiles/*object_sweep_20.json)"}
                  if ranks_info is not None else None),
        "shared_scene": ({"broadcast_MB_per_step": bcast_total / args.steps / 1e6, "ingest_rank": 0,
                          "note": "every object of every rank tracks in one camera stream; rank 0's depth + flow frames are "
                                  "broadcast batch by batch inside the timed region"} if args.shared_scene else None),
        "frames_per_sec_per_object": 1e3 / med["ms_per_step"],
        "launches_per_frame": med["launches_per_frame"],
        "event_ops_per_frame": med["event_ops_per_frame"],
        "roofline": roofline,
        "roofline_other": roofline_other,
        "cpu_baseline": cpu,
        "cpu_baseline_multicore": cpu_multi,
        "speedup_vs_cpu_1core": (value / cpu["value"]) if cpu else None,
        "speedup_vs_cpu_multicore": (value / cpu_multi["value"]) if cpu_multi and "value" in cpu_multi else None,
        "value_pcie_inclusive": pcie["per_object_streams"]["value"] if pcie else None,
        "value_pcie_inclusive_shared_scene": pcie["shared_scene"]["value"] if pcie else None,
        "value_pcie_inclusive_in_place": pcie["per_object_streams_in_place"]["value"] if pcie else None,
        "value_pcie_inclusive_shared_scene_in_place": pcie["shared_scene_in_place"]["value"] if pcie else None,
        "pcie_inclusive": pcie,
        "value_cold": value_cold["value"] if value_cold else None,
        "cold_run": value_cold,
        "live_latency": live,
        "adds_vs_gt_mm": {"mean": 1e3 * float(adds_gt.mean()), "auc": metrics.auc(adds_gt), "objects": adds_gt_objects,
                          "note": "window 0 (the canonical streams), every object of every rank"},
        "adds_vs_cpu_ref_mm": ({"mean": 1e3 * float(adds_cpu.mean()), "max": 1e3 * float(adds_cpu.max()), "objects_per_rank": adds_cpu_objects}
                               if adds_cpu is not None else None),
        "rmse_vs_gt": rmse,
        "pipeline": "four HIP streams per engine (mask frames / velocity chain / two pose lanes): one mask kernel per frame (small "
                    "workgroups, nothing persistent), one per-object kernel per batch for the velocity filter and per segment for a pose "
                    "lane, twists handed from the velocity filter to the lanes frame by frame in bursts, up to 5 batches in flight",
        "kernels_post_run_breakdown": kernels,
        "dominant_kernel": dominant,
        "stream_generation_s": gen_s[0],
    }
    emit(out, args.json_out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) >= 5 and sys.argv[1] == "--cpu-worker":
        cpu_worker(sys.argv[2], sys.argv[3], int(sys.argv[4]))
    else:
        main()
