"""The render contract (oracle/ro_render.c, reproduced bit for bit by the HIP rasteriser) against the two renderers it is NOT:
the arithmetic rounds 1 - 4 rendered with (RENDER_V1) and the numerics of the reference's OpenGL pipeline (RENDER_GL: window z
interpolated in screen space, 24-bit depth test, the float linearisation of src/roft-lib/shader/shader_model.frag:33-51 with
near 0.001 / far 1000, top-left rule).  CPU only; the full-size study is tools/render_gap.py -> profiles/r06_render_gap.json
(configs #3 - #5), this is its bounded sample, with the failure criteria of VERDICT r05 #4 / ADVICE r05:
more than 0.1 % of the outlier decisions (ROFTFilter.cpp:581-583) differing from the contract's fails the suite."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from roft_amd import synth
import util
from oracle import binding as ob


def ulps(a, b):
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    return np.abs(ia - ib)


def test_v1_and_contract_depths_differ_by_a_few_ulp():
    """Round 5 changed the contract's arithmetic (one reciprocal per vertex, one quotient per pixel).  Same coverage up to pixel
    centres that sit on an edge to the last bit of the projection, depths within a few float ulp -- at z = 0.7 m an ulp is 6e-8 m."""
    cam = synth.Camera.shape_a()
    ocam = util.oracle_camera(ob, cam)
    worst = 0
    for seed in (34, 35, 36):
        st = util.stream(seed, 2, scale=1, mesh_n=24)
        mesh = ob.make_mesh(*st.mesh)
        for k in range(2):
            a = ob.render_depth(mesh, st.gt.x[k], st.gt.q[k], ocam, 2)
            assert np.array_equal(a, ob.render_depth_mode(mesh, st.gt.x[k], st.gt.q[k], ocam, 2, ob.RENDER_CONTRACT))
            b = ob.render_depth_mode(mesh, st.gt.x[k], st.gt.q[k], ocam, 2, ob.RENDER_V1)
            both = (a > 0) & (b > 0)
            assert both.sum() > 2000
            assert ((a > 0) != (b > 0)).sum() <= 4            # (a centre on an edge to the last bit)
            u = ulps(a[both], b[both])
            # interior pixels: a few ulp; a pixel whose nearest triangle changes hands at a shared edge: still < 1e-5 m
            assert np.percentile(u, 99) <= 8, np.percentile(u, 99)
            assert np.abs(a[both] - b[both]).max() < 1e-5
            worst = max(worst, int(np.percentile(u, 99)))
    assert worst >= 0


def test_gl_numerics_quantise_depth_to_hundredths_of_a_millimetre():
    """What the reference's shader returns after linearising a float gl_FragCoord.z: at z ~ 0.5 - 0.9 m one ulp of window z is
    0.015 - 0.05 mm of depth.  The contract's exact eye Z differs from it by that much and no more; coverage differs only where a
    pixel centre sits exactly on an edge (top-left rule against inclusive edges)."""
    cam = synth.Camera.shape_a()
    ocam = util.oracle_camera(ob, cam)
    st = util.stream(34, 1, scale=1, mesh_n=24)
    mesh = ob.make_mesh(*st.mesh)
    a = ob.render_depth(mesh, st.gt.x[0], st.gt.q[0], ocam, 2)
    g = ob.render_depth_mode(mesh, st.gt.x[0], st.gt.q[0], ocam, 2, ob.RENDER_GL)
    both = (a > 0) & (g > 0)
    assert both.sum() > 2000 and ((a > 0) != (g > 0)).sum() <= 4
    d = np.abs(a[both] - g[both])
    z = float(a[both].mean())
    ulp_mm = 1e3 * z * z / 0.001 * 2.0 ** -24       # dZ = Z^2 / near * d(z_w), d(z_w) = one ulp below 1.0
    assert 1e3 * np.median(d) < 1.5 * ulp_mm and 1e3 * np.percentile(d, 99) < 4 * ulp_mm, (np.median(d), ulp_mm)
    assert 1e3 * d.max() < 0.25                      # (a pixel whose nearest surface changes hands under the 24-bit test)
    assert len(np.unique(g[both])) < len(np.unique(a[both]))   # quantised: fewer distinct depths than the exact render


def test_outlier_decisions_do_not_depend_on_the_renderers_last_bits():
    """A bounded sample of tools/render_gap.py: objects of configs #4 and #3, every outlier test scored on all three renders."""
    import render_gap
    rows = [render_gap.study([render_gap.object_stream(4, o, 37)], 37) for o in (0, 1, 2, 3)]
    rows.append(render_gap.study([render_gap.object_stream(3, 0, 19)], 19))
    rows = np.concatenate(rows)
    s = render_gap.summarise(rows)
    assert s["tests"] >= 24
    for mode, tol in (("v1", 1e-5), ("gl", 5e-3)):
        assert s[mode]["flipped_fraction"] <= 1e-3, s
        assert s[mode]["max_rel_dL"] < tol, s
        assert s[mode]["tests_with_a_sample_set_that_differs"] == 0
    # the closest test is far from the threshold compared with what a renderer moves L by
    assert s["closest_ratio_to_threshold"] > 10 * s["gl"]["max_rel_dL"]


def test_full_size_study_is_committed_and_clean():
    """profiles/r06_render_gap.json (tools/render_gap.py on configs #3 - #5 at their test sizes): <= 0.1 % of the decisions flip."""
    path = os.path.join(ROOT, "profiles", "r06_render_gap.json")
    if not os.path.exists(path):
        pytest.skip("study not collected yet")
    rep = json.load(open(path))
    assert rep["all"]["tests"] > 2000
    for mode in ("v1", "gl"):
        assert rep["all"][mode]["flipped_fraction"] <= 1e-3, rep["all"][mode]
