"""The N > 1 path of bench.py executed for real: two ranks (two processes, both on cuda:0, gloo for the barrier / the
max-over-ranks time / the all-gather of the result rows -- the driver's multi-GPU runs use one GPU per rank and RCCL) track
the block-sharded objects of BASELINE config #4's layout; the [objects, steps, 19] block rank 0 gathers inside the timed
region must equal, bit for bit, what one rank tracking all the objects logs (objects never exchange data: SURVEY 8e)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def run_bench(tmp_path, n_ranks, tag, port, extra=(), launcher=True, cpu_ref=False):
    out = str(tmp_path / ("rows_%s.npy" % tag))
    detail = str(tmp_path / ("detail_%s.json" % tag))
    args = ["bench.py", "--gpus", str(n_ranks), "--steps", "14", "--warmup", "4", "--objects", "5", "--windows", "3",
            "--pcie-frames", "0", "--no-extras", "--no-kernel-timing", "--rehearsal-ms", "0", "--dump-rows", out, "--json-out", detail] + list(extra)
    if not cpu_ref:
        args.append("--no-cpu-baseline")
    env = dict(os.environ, ROFT_BENCH_DEVICE="0", ROFT_BENCH_BACKEND="gloo")
    if n_ranks > 1 and launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
               "--master-addr", "127.0.0.1", "--master-port", str(port)] + args
    else:
        cmd = [sys.executable] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    # the LAST line of stdout is the compact record the driver parses (< 4 KB, strict JSON); the full record (windows, batch
    # traces, ranks) is the side file it names -- the tests below read the full record and check the line against it
    text = r.stdout.strip().splitlines()[-1]
    assert len(text) < 4096, len(text)
    line = json.loads(text, parse_constant=lambda c: pytest.fail("non-strict JSON constant " + c))
    full = json.load(open(detail))
    assert line["detail"] == detail
    for k in ("metric", "unit", "n_gpus", "steps", "warmup", "scaling", "dtype", "data", "higher_is_better"):
        assert line[k] == full[k], k
    assert abs(line["value"] - full["value"]) <= 1e-5 * full["value"] and abs(line["ms_per_step"] - full["ms_per_step"]) <= 1e-5 * full["ms_per_step"]
    assert line["config"]["objects_per_gpu"] == full["config"]["objects_per_gpu"] and line["config"]["ranks"] == n_ranks
    assert (line.get("ranks") is None) == (n_ranks == 1)
    if n_ranks > 1:
        assert line["ranks"]["world_size"] == n_ranks and line["ranks"]["seen"] == n_ranks
    return full, np.load(out)


def test_two_ranks_gather_what_one_rank_logs(tmp_path):
    one, rows1 = run_bench(tmp_path, 1, "one", 0)
    two, rows2 = run_bench(tmp_path, 2, "two", 29517, cpu_ref=True)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    # the line explains itself: every window's value (value = their median), the batch trace of the median window, who was there
    for line in (one, two):
        assert len(line["runs"]) == 3 and line["value"] == sorted(line["runs"])[1] and line["value_min"] <= line["value"] <= line["value_max"]
        assert len(line["windows"]) == 3 and all(len(w["batches"]) == len(line["config"]["timed_batches"]) for w in line["windows"])
        assert all(b["done_at_ms"] is not None and b["done_at_ms"] > 0 and b["steady"] == 0 for b in line["batches"])
        assert abs(line["ms_per_step"] * line["value"] - 1e3 * 5) < 1e-6 * 5e3
    assert one["ranks"] is None
    rk = two["ranks"]
    assert rk["world_size"] == 2 and rk["ranks_seen_by_all_reduce"] == 2 and rk["objects_per_gpu"] == [3, 2] and rk["first_object_of_rank"] == [0, 3]
    assert len(rk["window_ms_per_rank"]) == 2 and all(len(w) == 3 for w in rk["window_ms_per_rank"])
    assert abs(rk["per_gpu_rate_of_the_median_window"] - two["value"] / 2) < 1e-9 * two["value"]
    # accuracy at N > 1: ADD-S vs ground truth over the objects of BOTH ranks, ADD-S vs the CPU path on objects of both
    assert two["adds_vs_gt_mm"]["objects"] == 5 and one["adds_vs_gt_mm"]["objects"] == 5
    assert abs(two["adds_vs_gt_mm"]["mean"] - one["adds_vs_gt_mm"]["mean"]) < 1e-9
    assert two["adds_vs_cpu_ref_mm"]["objects_per_rank"] == [3, 2] and two["adds_vs_cpu_ref_mm"]["max"] < 1e-3
    assert two["config"]["objects_total"] == 5 and two["config"]["objects_per_gpu"] == 3     # block partition 3 + 2
    assert two["scaling"] == "strong" and two["cpu_baseline"] is None
    assert rows1.shape == rows2.shape == (5, 14, 19)
    assert np.array_equal(rows1, rows2)
    assert np.isfinite(rows2).all() and np.abs(rows2[:, :, 9:13]).max() <= 1.0 + 1e-12


def test_shared_scene_broadcast_delivers_the_ingest_ranks_frames(tmp_path):
    """--shared-scene: every object of both ranks tracks in ONE camera stream; rank 1 starts with zeroed images and only has
    what rank 0 broadcasts batch by batch inside the timed region (SURVEY 8e ii) -- and logs what one rank holding the
    stream itself logs."""
    one, rows1 = run_bench(tmp_path, 1, "s1", 0, ["--shared-scene"])
    two, rows2 = run_bench(tmp_path, 2, "s2", 29519, ["--shared-scene"])
    assert two["shared_scene"]["broadcast_MB_per_step"] > 3.0      # 640x480: 1.2 MB depth + 2.5 MB flow per frame
    assert one["shared_scene"]["broadcast_MB_per_step"] == 0.0
    assert np.array_equal(rows1, rows2)
    assert np.array_equal(rows2[0], rows2[4])                      # the same scene and object model for every tracker


def test_gpus_n_without_a_launcher_starts_the_ranks_itself(tmp_path):
    """`python bench.py --gpus 2`, no torch.distributed.run around it: bench.py starts the two ranks as a child process (before
    it touches the GPU), relays rank 0's line -- n_gpus 2, the block partition, the gathered rows of both ranks."""
    env_clean = {k: os.environ[k] for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK") if k in os.environ}
    for k in env_clean:
        del os.environ[k]
    try:
        two, rows2 = run_bench(tmp_path, 2, "self", 0, launcher=False)
    finally:
        os.environ.update(env_clean)
    assert two["n_gpus"] == 2 and two["config"]["ranks"] == 2
    assert two["config"]["objects_total"] == 5 and two["config"]["objects_per_gpu"] == 3
    assert two["timed_frames_first_touch"] is True
    assert rows2.shape == (5, 14, 19) and np.isfinite(rows2).all()
    one, rows1 = run_bench(tmp_path, 1, "self1", 0)
    assert np.array_equal(rows1, rows2)


def test_the_drivers_command_prints_one_small_parseable_line(tmp_path):
    """`python3 bench.py --gpus 1 --steps 20 --warmup 5` -- exactly what the driver runs and records as BENCH_rNN.json -- with every
    leg on (CPU baselines, PCIe legs, cold run, instrumented windows): stdout is ONE line, under 4 KB, strict JSON, with the
    contract's keys, `roofline` and `cpu_baseline` filled in; the side file holds the rest.  (Round 5's line was 20.7 KB and came
    back as `parsed: null`.)"""
    detail = str(tmp_path / "detail.json")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--json-out", detail],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1 and len(r.stdout.encode()) < 4096, (len(lines), len(r.stdout))
    tail = r.stdout.encode()[-8192:].decode()       # what the driver keeps
    d = json.loads(tail.strip().splitlines()[-1], parse_constant=lambda c: pytest.fail("non-strict JSON constant " + c))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["unit"] == "object-frames/s" and d["dtype"] == "f64"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["data"] == "synthetic"
    assert d["config"]["objects_total"] == 64 and d["config"]["width"] == 640 and d["config"]["height"] == 480 and "workload" in d["config"]
    assert abs(d["value"] * d["ms_per_step"] - 64e3) < 1e-3 * 64e3 and len(d["runs"]) == 5
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and 0.02 < rf["frac"] < 0.7
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5 and rf["avg_launch_us"] > 5 and rf["launches"] >= 4
    cb = d["cpu_baseline"]
    assert cb["cores"] == 1 and cb["kind"] == "port" and cb["unit"] == "object-frames/s" and 100 < cb["value"] < 1e5 and cb["sample"]
    assert d["value"] > 50 * cb["value"]
    assert d["adds_vs_cpu_ref_mm"]["max"] < 1e-3
    full = json.load(open(detail))
    assert len(full["windows"]) == 5 and full["roofline"]["note"] and full["cpu_baseline_multicore"]
