"""Meshes for the closed-surface classification and the back-face rule of the render contract (oracle/ro_meshclass.c,
roft_amd/csrc/mesh_class.hip): name -> (verts float32 [n, 3], tris int32 [m, 3], closed?)."""
import numpy as np

from roft_amd import synth


def box(n=6, half=(0.08, 0.10, 0.035)):
    v, t = synth.box_mesh(half, n)
    return np.ascontiguousarray(v, np.float32), np.ascontiguousarray(t, np.int32)


def zoo(n=6):
    v, t = box(n)
    out = {}
    out["box"] = (v, t, True)
    out["box_reversed"] = (v, np.ascontiguousarray(t[:, ::-1]), True)                      # inside out: closed, every flip inverted
    rng = np.random.default_rng(5)
    t_mixed = t.copy()
    sel = rng.random(len(t)) < 0.5
    t_mixed[sel] = t_mixed[sel][:, [0, 2, 1]]
    out["box_random_windings"] = (v, t_mixed, True)                                        # orientable, inconsistently wound
    out["box_open"] = (v, np.ascontiguousarray(t[3:]), False)                              # three triangles missing
    out["box_duplicate_triangle"] = (v, np.ascontiguousarray(np.concatenate([t, t[:1]])), False)   # an edge in three triangles
    t_deg = t.copy()
    t_deg[7, 2] = t_deg[7, 1]
    out["box_degenerate_triangle"] = (v, t_deg, False)
    # every triangle with corners of its own (what an OBJ with per-face normals looks like): closed after welding
    out["box_unwelded"] = (np.ascontiguousarray(v[t].reshape(-1, 3)), np.arange(3 * len(t), dtype=np.int32).reshape(-1, 3), True)
    # two components: a box and a smaller inside-out one next to it
    v2 = (v * 0.5 + np.array([0.2, 0.0, 0.0], np.float32)).astype(np.float32)
    out["two_components"] = (np.concatenate([v, v2]), np.ascontiguousarray(np.concatenate([t, t[:, ::-1] + len(v)]).astype(np.int32)), True)
    # the real projective plane on six vertices: every edge in exactly two triangles, not orientable
    rp2 = np.array([[1, 2, 3], [1, 2, 4], [1, 3, 5], [1, 4, 6], [1, 5, 6], [2, 3, 6], [2, 4, 5], [2, 5, 6], [3, 4, 5], [3, 4, 6]], np.int32) - 1
    out["projective_plane"] = (rng.normal(size=(6, 3)).astype(np.float32) * 0.05, rp2, False)
    # a torus (genus 1: closed and orientable, not simply connected) with every second quad split the other way, and the same
    # surface without one ring of quads (open)
    nu, nv_, R0, r0 = 4 * n, 2 * n, 0.07, 0.025
    uu, vv = np.meshgrid(np.arange(nu) * 2 * np.pi / nu, np.arange(nv_) * 2 * np.pi / nv_, indexing="ij")
    tv = np.stack([(R0 + r0 * np.cos(vv)) * np.cos(uu), (R0 + r0 * np.cos(vv)) * np.sin(uu), r0 * np.sin(vv)], -1).reshape(-1, 3).astype(np.float32)
    tt = []
    for i in range(nu):
        for j in range(nv_):
            a, b, c, d = i * nv_ + j, ((i + 1) % nu) * nv_ + j, ((i + 1) % nu) * nv_ + (j + 1) % nv_, i * nv_ + (j + 1) % nv_
            tt += [[a, b, c], [a, c, d]] if (i + j) % 2 == 0 else [[a, b, d], [b, c, d]]
    tt = np.array(tt, np.int32)
    out["torus"] = (tv, tt, True)
    out["torus_open"] = (tv, np.ascontiguousarray(tt[2 * nv_:]), False)
    # a hollow box: an outer shell and an inner one wound towards the cavity (two closed components, one inside the other)
    vin = (v * 0.6).astype(np.float32)
    out["hollow_box"] = (np.concatenate([v, vin]), np.ascontiguousarray(np.concatenate([t, t[:, ::-1] + len(v)]).astype(np.int32)), True)
    v_nan = v.copy()
    v_nan[3, 1] = np.nan
    out["box_nan_vertex"] = (v_nan, t, False)
    return out
