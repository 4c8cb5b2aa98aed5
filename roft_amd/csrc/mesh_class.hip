// mesh_class.hip -- closed-surface classification of an object's mesh and the walk order of its triangles (host code only).
//
// Reference: the reference draws every triangle of the mesh with the depth test LESS and no culling (SICAD.cpp:271-272), so the
// rendered value is the nearest surface along a pixel's ray.  Seen from outside, the nearest surface of a closed mesh faces the
// camera: the render contract (oracle/ro_render.c, round 6) therefore leaves the triangles that face away out -- for meshes THIS
// classification accepts, and only while every vertex is in front of the near plane.  The rules are stated in
// oracle/ro_meshclass.c; oracle and engine implement them independently and tests/test_parity_gpu.py compares the two on closed,
// open, inside-out, non-manifold and multi-component meshes.
#include "mesh_class.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

namespace roft {

bool classify_mesh(const float* verts, int n_verts, const int32_t* tris, int n_tris, std::vector<uint8_t>& flip)
{
    flip.assign((size_t)std::max(n_tris, 0), 0);
    if (n_verts <= 0 || n_tris <= 0) return false;
    // 1. weld vertices with equal coordinates (-0 == +0) to the smallest index of their group
    struct P { float x, y, z; };
    std::vector<P> pos((size_t)n_verts);
    for (int i = 0; i < n_verts; ++i) {
        pos[i] = {verts[3 * i] + 0.0f, verts[3 * i + 1] + 0.0f, verts[3 * i + 2] + 0.0f};
        if (std::isnan(pos[i].x) || std::isnan(pos[i].y) || std::isnan(pos[i].z)) return false;
    }
    std::vector<int> by_pos((size_t)n_verts), id((size_t)n_verts);
    std::iota(by_pos.begin(), by_pos.end(), 0);
    std::sort(by_pos.begin(), by_pos.end(), [&](int a, int b) {
        if (pos[a].x != pos[b].x) return pos[a].x < pos[b].x;
        if (pos[a].y != pos[b].y) return pos[a].y < pos[b].y;
        if (pos[a].z != pos[b].z) return pos[a].z < pos[b].z;
        return a < b;
    });
    for (int i = 0; i < n_verts; ++i) {
        const int v = by_pos[i];
        if (i > 0) {
            const int u = by_pos[i - 1];
            if (pos[u].x == pos[v].x && pos[u].y == pos[v].y && pos[u].z == pos[v].z) { id[v] = id[u]; continue; }
        }
        id[v] = v;
    }
    // 2. three distinct corners per triangle; every undirected edge in exactly two triangles
    struct E { int lo, hi, slot, fwd; };   // slot = 3 * triangle + edge; fwd: the triangle walks lo -> hi
    std::vector<E> edges((size_t)n_tris * 3);
    for (int t = 0; t < n_tris; ++t) {
        int v[3];
        for (int k = 0; k < 3; ++k) {
            const int32_t raw = tris[3 * t + k];
            if (raw < 0 || raw >= n_verts) return false;
            v[k] = id[raw];
        }
        if (v[0] == v[1] || v[1] == v[2] || v[0] == v[2]) return false;
        for (int k = 0; k < 3; ++k) {
            const int a = v[k], b = v[(k + 1) % 3];
            edges[(size_t)3 * t + k] = {std::min(a, b), std::max(a, b), 3 * t + k, a < b ? 1 : 0};
        }
    }
    std::sort(edges.begin(), edges.end(), [](const E& a, const E& b) {
        if (a.lo != b.lo) return a.lo < b.lo;
        if (a.hi != b.hi) return a.hi < b.hi;
        return a.slot < b.slot;
    });
    std::vector<int> nb((size_t)n_tris * 3);   // across edge `slot`: neighbour triangle * 2 + (both walk the edge the same way)
    for (size_t i = 0; i < edges.size(); i += 2) {
        if (i + 1 >= edges.size() || edges[i].lo != edges[i + 1].lo || edges[i].hi != edges[i + 1].hi) return false;
        if (i + 2 < edges.size() && edges[i + 2].lo == edges[i].lo && edges[i + 2].hi == edges[i].hi) return false;
        const int same = edges[i].fwd == edges[i + 1].fwd ? 1 : 0;
        nb[edges[i].slot] = (edges[i + 1].slot / 3) * 2 + same;
        nb[edges[i + 1].slot] = (edges[i].slot / 3) * 2 + same;
    }
    // 3. orientation per connected component, from its lowest triangle
    std::vector<int> comp((size_t)n_tris, -1), queue;
    queue.reserve((size_t)n_tris);
    int n_comp = 0;
    for (int seed = 0; seed < n_tris; ++seed) {
        if (comp[seed] >= 0) continue;
        size_t head = queue.size();
        queue.push_back(seed);
        comp[seed] = n_comp;
        flip[seed] = 0;
        while (head < queue.size()) {
            const int t = queue[head++];
            for (int k = 0; k < 3; ++k) {
                const int u = nb[(size_t)3 * t + k] >> 1;
                const uint8_t want = flip[t] ^ (uint8_t)(nb[(size_t)3 * t + k] & 1);
                if (comp[u] < 0) { comp[u] = n_comp; flip[u] = want; queue.push_back(u); }
                else if (flip[u] != want) { flip.assign((size_t)n_tris, 0); return false; }
            }
        }
        ++n_comp;
    }
    // 4. outward: the signed volume of every component is positive
    std::vector<double> vol((size_t)n_comp, 0.0);
    for (int t = 0; t < n_tris; ++t) {
        const float* a = verts + 3 * (size_t)tris[3 * t];
        const float* b = verts + 3 * (size_t)tris[3 * t + 1];
        const float* c = verts + 3 * (size_t)tris[3 * t + 2];
        const double cx = (double)b[1] * c[2] - (double)b[2] * c[1], cy = (double)b[2] * c[0] - (double)b[0] * c[2], cz = (double)b[0] * c[1] - (double)b[1] * c[0];
        const double det = ((double)a[0] * cx + (double)a[1] * cy) + (double)a[2] * cz;
        vol[comp[t]] += (flip[t] ? -det : det) / 6.0;
    }
    for (double v : vol)
        if (!(v != 0.0) || !std::isfinite(v)) { flip.assign((size_t)n_tris, 0); return false; }
    for (int t = 0; t < n_tris; ++t)
        if (vol[comp[t]] < 0.0) flip[t] ^= 1;
    return true;
}

void facing_coherent_order(const float* verts, const int32_t* tris, int n_tris, const std::vector<uint8_t>& flip, std::vector<int32_t>& order)
{
    order.resize((size_t)std::max(n_tris, 0));
    std::vector<int> key((size_t)std::max(n_tris, 0));
    for (int t = 0; t < n_tris; ++t) {
        const float* a = verts + 3 * (size_t)tris[3 * t];
        const float* b = verts + 3 * (size_t)tris[3 * t + 1];
        const float* c = verts + 3 * (size_t)tris[3 * t + 2];
        const double e1[3] = {(double)b[0] - a[0], (double)b[1] - a[1], (double)b[2] - a[2]}, e2[3] = {(double)c[0] - a[0], (double)c[1] - a[1], (double)c[2] - a[2]};
        double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        if (flip[t]) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
        // cube face of the outward normal, then the quadrant of the face it points into
        int ax = 0;
        if (std::fabs(n[1]) > std::fabs(n[ax])) ax = 1;
        if (std::fabs(n[2]) > std::fabs(n[ax])) ax = 2;
        const int face = 2 * ax + (n[ax] < 0.0 ? 1 : 0), u = (ax + 1) % 3, v = (ax + 2) % 3;
        key[t] = face * 4 + (n[u] < 0.0 ? 1 : 0) * 2 + (n[v] < 0.0 ? 1 : 0);
        order[t] = t;
    }
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return key[a] < key[b]; });
}

void prepare_mesh(const float* verts, int n_verts, const int32_t* tris, int n_tris, PreparedMesh& out)
{
    std::vector<uint8_t> flip;
    out.closed = classify_mesh(verts, n_verts, tris, n_tris, flip);
    out.reordered.clear();
    out.flip.clear();
    if (!out.closed) return;
    std::vector<int32_t> order;
    facing_coherent_order(verts, tris, n_tris, flip, order);
    out.reordered.resize((size_t)3 * n_tris);
    out.flip.resize((size_t)n_tris);
    for (int k = 0; k < n_tris; ++k) {
        const int32_t t = order[k];
        for (int q = 0; q < 3; ++q) out.reordered[(size_t)3 * k + q] = tris[(size_t)3 * t + q];
        out.flip[k] = flip[t];
    }
}

}  // namespace roft
