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


def run_bench(tmp_path, n_ranks, tag, port):
    out = str(tmp_path / ("rows_%s.npy" % tag))
    args = ["bench.py", "--gpus", str(n_ranks), "--steps", "14", "--warmup", "4", "--objects", "5", "--no-cpu-baseline",
            "--pcie-frames", "0", "--no-extras", "--no-kernel-timing", "--clock-warm-ms", "0", "--dump-rows", out]
    env = dict(os.environ, ROFT_BENCH_DEVICE="0", ROFT_BENCH_BACKEND="gloo")
    if n_ranks > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
               "--master-addr", "127.0.0.1", "--master-port", str(port)] + args
    else:
        cmd = [sys.executable] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line), np.load(out)


def test_two_ranks_gather_what_one_rank_logs(tmp_path):
    one, rows1 = run_bench(tmp_path, 1, "one", 0)
    two, rows2 = run_bench(tmp_path, 2, "two", 29517)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["objects_total"] == 5 and two["config"]["objects_per_gpu"] == 3     # block partition 3 + 2
    assert two["scaling"] == "strong" and two["cpu_baseline"] is None
    assert rows1.shape == rows2.shape == (5, 14, 19)
    assert np.array_equal(rows1, rows2)
    assert np.isfinite(rows2).all() and np.abs(rows2[:, :, 9:13]).max() <= 1.0 + 1e-12
