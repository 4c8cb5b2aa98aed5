#!/usr/bin/env python3
"""How far is the render contract (oracle/ro_render.c, what the HIP rasteriser reproduces bit for bit) from (a) the arithmetic
rounds 1 - 4 rendered with and (b) the numerics of the reference's OpenGL pipeline (window-z interpolation, 24-bit depth test,
the float linearisation of shader_model.frag:33-51 with near 0.001 / far 1000, top-left rule)?  CPU only, oracle only.

Every outlier test of the oracle tracker on the workloads of BASELINE configs #3 - #5 (SURVEY 8d streams, same seeds as
tools/run_baseline_configs.py) also scores its two alternatives on renders in those two arithmetics (ro_tracker_shadow_render);
the tracker itself goes on with the contract's decision, so every test compares the three renderers on identical inputs.
Reported per config and mode: tests, decisions that differ from the contract's (ROFTFilter.cpp:581-583: L0 > 2 L1), max and
mean |dL| / L, and how close the closest test sits to the threshold.

usage: python tools/render_gap.py [--frames3 98] [--frames4 98] [--objects4 64] [--frames5 600] [--out profiles/r06_render_gap.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np

from roft_amd import synth
import util
from oracle import binding as ob


def study(streams, n_frames):
    rows = []
    for st in streams:
        cfg = util.oracle_config(ob, st)
        trk = ob.Tracker(cfg, *st.mesh)
        trk.shadow_render(True)
        for k in range(n_frames):
            depth, flow, mask, pose = util.frame_inputs(st, k)
            r = trk.step(st.dt, depth, flow, mask, pose)
            if r.outlier_selected >= 0:
                sh = trk.shadow_L()
                rows.append((r.outlier_L[0], r.outlier_L[1], sh[0][0], sh[0][1], sh[1][0], sh[1][1]))
        trk.close()
    return np.array(rows).reshape(-1, 6)


def summarise(rows):
    out = {"tests": int(len(rows))}
    if not len(rows):
        return out
    L = rows[:, 0:2]
    finite = np.all(L < 1e300, axis=1)
    sel = L[:, 0] > 2.0 * L[:, 1]
    out["vel_only_chosen"] = int(sel.sum())
    ratio = L[finite, 0] / (2.0 * L[finite, 1])
    out["closest_ratio_to_threshold"] = float(np.min(np.abs(ratio - 1.0))) if finite.any() else None
    for name, c in (("v1", 2), ("gl", 4)):
        M = rows[:, c:c + 2]
        selm = M[:, 0] > 2.0 * M[:, 1]
        ok = finite & np.all(M < 1e300, axis=1)
        rel = np.abs(M[ok] - L[ok]) / L[ok]
        out[name] = {"decisions_flipped": int((selm != sel).sum()),
                     "flipped_fraction": float((selm != sel).mean()),
                     "max_rel_dL": float(rel.max()) if rel.size else None,
                     "mean_rel_dL": float(rel.mean()) if rel.size else None,
                     "max_abs_dL_mm": float(1e3 * np.abs(M[ok] - L[ok]).max()) if ok.any() else None,
                     "tests_with_a_sample_set_that_differs": int((np.all(M < 1e300, axis=1) != finite).sum())}
    return out


def object_stream(which, o, n_frames):
    """Object o of BASELINE config #`which` (the streams of tools/run_baseline_configs.py / bench.py, on the CPU)."""
    if which == 3:
        return synth.make_stream(3000 + o, n_frames, synth.Camera.shape_b(), flow_type=synth.FLOW_S16C2, half_extents=synth.FAST_YCB_HALF_EXTENTS[o])
    if which == 4:
        scale = 0.8 + 0.4 * (((o % 64) * 7) % 10) / 9.0   # (bench.py's extents)
        half = tuple(h * scale for h in synth.CRACKER_BOX_HALF_EXTENTS)
        return synth.make_stream(4000 + o, n_frames, synth.Camera.shape_a(), flow_type=synth.FLOW_F32C2, half_extents=half)
    # config #5: 60 looping images, the delivery schedules run on for the whole length
    return synth.make_stream(5000 + o, 0, synth.Camera.shape_b(), flow_type=synth.FLOW_S16C2, half_extents=synth.FAST_YCB_HALF_EXTENTS[o % 5],
                             period=60, n_schedule=n_frames)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--frames3", type=int, default=98)
    p.add_argument("--frames4", type=int, default=98)
    p.add_argument("--objects4", type=int, default=64)
    p.add_argument("--frames5", type=int, default=600)
    p.add_argument("--objects5", type=int, default=16)
    p.add_argument("--out", default="")
    a = p.parse_args()
    rep = {"what": "oracle tracker with shadow renders: every outlier test scored on the contract's render, on the arithmetic of "
                   "rounds 1 - 4 (v1) and on the GL pipeline's numerics (gl); tools/render_gap.py"}
    all_rows = []
    for which, nf, no in ((3, a.frames3, 5), (4, a.frames4, a.objects4), (5, a.frames5, a.objects5)):
        if nf <= 0:
            continue
        t0 = time.time()
        rows = []
        for o in range(no):   # one object at a time: the streams of config #5 are 60 images of 1280x720 each
            st = object_stream(which, o, nf)
            rows.append(study([st], nf))
            del st
        rows = np.concatenate(rows) if rows else np.zeros((0, 6))
        all_rows.append(rows)
        rep["config_%d" % which] = dict(summarise(rows), objects=no, frames=nf, seconds=round(time.time() - t0, 1))
        print("config #%d: %s" % (which, json.dumps(rep["config_%d" % which])), flush=True)
    rep["all"] = summarise(np.concatenate(all_rows))
    print(json.dumps(rep["all"]))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(rep, f, indent=1)


if __name__ == "__main__":
    main()
