#!/usr/bin/env python3
"""Times the optical-flow producer (roft_flow_producer_*) on textured synthetic pairs resident in HBM.

usage: python tools/bench_flow_producer.py [--pairs 64] [--shape A|B] [--flow f32|s16] [--reps 10]
Prints one JSON line: pairs/s, ms per batch, arithmetic of the Lucas-Kanade level-0..L-1 kernels.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from roft_amd import _lib as L
from roft_amd import ops, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=64)
    ap.add_argument("--shape", default="A")
    ap.add_argument("--flow", default="f32")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--levels", type=int, default=3)
    ap.add_argument("--radius", type=int, default=3)
    ap.add_argument("--iterations", type=int, default=3)
    args = ap.parse_args()
    L.require_device()
    cam = synth.Camera.shape_a() if args.shape == "A" else synth.Camera.shape_b()
    W, H = cam.width, cam.height
    n = args.pairs
    # a handful of distinct textured pairs, repeated to fill the batch
    base = [synth.make_stream(900 + i, 2, cam, with_gray=True, device="cuda").gray for i in range(min(n, 8))]
    prev = [base[i % len(base)][0].clone() for i in range(n)]
    cur = [base[i % len(base)][1].clone() for i in range(n)]
    ft = L.FLOW_F32C2 if args.flow == "f32" else L.FLOW_S16C2
    out = [torch.zeros((H, W, 2), dtype=torch.float32, device="cuda") if ft == L.FLOW_F32C2 else
           torch.zeros((H // 4, W // 4, 2), dtype=torch.int16, device="cuda") for _ in range(n)]
    torch.cuda.synchronize()
    fp = ops.FlowProducer(W, H, n, ft, levels=args.levels, radius=args.radius, iterations=args.iterations)
    pp, cc, oo = [t.data_ptr() for t in prev], [t.data_ptr() for t in cur], [t.data_ptr() for t in out]
    for _ in range(2):
        fp.run(pp, cc, oo)
    fp.sync()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        fp.run(pp, cc, oo)
    fp.sync()
    dt = (time.perf_counter() - t0) / args.reps
    fp.close()
    taps = (2 * args.radius + 1) ** 2
    px = sum((W >> l) * (H >> l) for l in range(args.levels))
    n1 = 2 * args.radius + 1
    # G taps (6 flop) + per iteration: horizontal pass (n1 + 1 rows x n1 lerps of 3 flop), vertical pass + residual + b
    # accumulation (8 flop per tap), update
    flops_px = taps * 6 + args.iterations * ((n1 + 1) * n1 * 3 + taps * 8 + 12)
    print(json.dumps(dict(workload="%dx%d, %d pairs, %s, L%d r%d it%d" % (W, H, n, args.flow, args.levels, args.radius, args.iterations),
                          pairs_per_s=n / dt, ms_per_batch=1e3 * dt, us_per_pair=1e6 * dt / n,
                          gflop_per_pair=1e-9 * px * flops_px, tflops=1e-12 * n * px * flops_px / dt,
                          io_bytes_per_pair=W * H * (2 + (8 if ft == L.FLOW_F32C2 else 0.25)))))


if __name__ == "__main__":
    main()
