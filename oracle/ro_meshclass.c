/*
 * ro_meshclass.c -- CPU oracle (test infrastructure): is a triangle mesh a closed, orientable surface, and which of its
 * triangles are wound inwards?
 *
 * Why: the render contract (ro_render.c) draws, like the reference (depth test LESS, no culling, SICAD.cpp:271-272), the
 * NEAREST surface along every pixel's ray.  Seen from outside, the nearest surface of a closed mesh is a triangle that faces
 * the camera, so the triangles that face away need not be scan-converted at all -- half of the rasteriser's work.  Since round 6
 * the contract says so explicitly (and the HIP rasteriser follows it bit for bit): for a mesh this file classifies as CLOSED,
 * rendered with every vertex in front of the near plane, triangles facing away from the camera are not drawn.  In exact
 * arithmetic that is the reference's image; in float arithmetic a pixel centre within rounding of a silhouette or of an edge
 * shared by two front faces can be claimed by a back face only -- tools/render_gap.py counts those pixels (RO_RENDER_GL and
 * RO_RENDER_V1 draw both faces).  Meshes that are open, non-manifold or non-orientable are drawn whole, as before.
 *
 * The classification (the engine's roft_object_add restates it; both follow this specification):
 *  1. weld: vertices with equal coordinates (-0 == +0) are one vertex (the smallest index of the group); a NaN -> not closed;
 *  2. every triangle has three distinct welded vertices, every undirected edge belongs to exactly two triangles -- else not closed;
 *  3. per connected component (triangles joined by edges), starting at its lowest triangle with flip = 0: a neighbour that
 *     walks the shared edge in the SAME direction gets the opposite flip, one that walks it the other way the same flip; a
 *     contradiction -> not closed (non-orientable);
 *  4. signed volume of the component V = sum_t (flip_t ? -1 : +1) v0 . (v1 x v2) / 6 in double, triangles in index order;
 *     V < 0: all flips of the component are inverted; V == 0 or not finite -> not closed;
 *  5. flip_t = 1 then means: triangle t is wound clockwise seen from outside.
 * Facing test in the rasteriser: with the pin-hole projection of the contract (x right, y down, z forward) a counter-clockwise-
 * from-outside triangle that faces the camera has NEGATIVE screen area; a triangle is drawn iff (area < 0) != flip_t.
 */
#include "roft_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; int idx; } weld_rec;
static int weld_cmp(const void* pa, const void* pb)
{
    const weld_rec* a = (const weld_rec*)pa; const weld_rec* b = (const weld_rec*)pb;
    if (a->x != b->x) return a->x < b->x ? -1 : 1;
    if (a->y != b->y) return a->y < b->y ? -1 : 1;
    if (a->z != b->z) return a->z < b->z ? -1 : 1;
    return a->idx < b->idx ? -1 : (a->idx > b->idx ? 1 : 0);
}
typedef struct { int lo, hi, tri, fwd; } edge_rec;   /* fwd: the triangle walks lo -> hi */
static int edge_cmp(const void* pa, const void* pb)
{
    const edge_rec* a = (const edge_rec*)pa; const edge_rec* b = (const edge_rec*)pb;
    if (a->lo != b->lo) return a->lo < b->lo ? -1 : 1;
    if (a->hi != b->hi) return a->hi < b->hi ? -1 : 1;
    return a->tri < b->tri ? -1 : (a->tri > b->tri ? 1 : 0);
}

int ro_mesh_classify(const float* verts, int n_verts, const int32_t* tris, int n_tris, uint8_t* flip)
{
    if (n_verts <= 0 || n_tris <= 0) return 0;
    int ok = 1;
    weld_rec* wr = (weld_rec*)malloc(sizeof(weld_rec) * (size_t)n_verts);
    int* id = (int*)malloc(sizeof(int) * (size_t)n_verts);
    edge_rec* er = (edge_rec*)malloc(sizeof(edge_rec) * (size_t)n_tris * 3);
    int* nb = (int*)malloc(sizeof(int) * (size_t)n_tris * 3);       /* neighbour across edge e of triangle t: tri * 2 + same_direction */
    int* comp = (int*)malloc(sizeof(int) * (size_t)n_tris);
    int* queue = (int*)malloc(sizeof(int) * (size_t)n_tris);
    double* vol = NULL;
    for (int i = 0; i < n_verts; i++) {
        wr[i].x = verts[3 * i] + 0.0f; wr[i].y = verts[3 * i + 1] + 0.0f; wr[i].z = verts[3 * i + 2] + 0.0f;   /* (-0 + 0 = +0) */
        wr[i].idx = i;
        if (!(wr[i].x == wr[i].x) || !(wr[i].y == wr[i].y) || !(wr[i].z == wr[i].z)) ok = 0;
    }
    if (!ok) goto done;
    qsort(wr, (size_t)n_verts, sizeof(weld_rec), weld_cmp);
    for (int i = 0; i < n_verts; i++) {
        if (i > 0 && wr[i].x == wr[i - 1].x && wr[i].y == wr[i - 1].y && wr[i].z == wr[i - 1].z) id[wr[i].idx] = id[wr[i - 1].idx];
        else id[wr[i].idx] = wr[i].idx;
    }
    for (int t = 0; t < n_tris && ok; t++) {
        int v[3];
        for (int k = 0; k < 3; k++) {
            const int32_t raw = tris[3 * t + k];
            if (raw < 0 || raw >= n_verts) { ok = 0; break; }
            v[k] = id[raw];
        }
        if (!ok) break;
        if (v[0] == v[1] || v[1] == v[2] || v[0] == v[2]) { ok = 0; break; }
        for (int k = 0; k < 3; k++) {
            const int a = v[k], b = v[(k + 1) % 3];
            edge_rec* e = er + 3 * (size_t)t + k;
            e->lo = a < b ? a : b; e->hi = a < b ? b : a; e->tri = 3 * t + k; e->fwd = a < b;
        }
    }
    if (!ok) goto done;
    qsort(er, (size_t)n_tris * 3, sizeof(edge_rec), edge_cmp);
    for (size_t i = 0; i < (size_t)n_tris * 3; i += 2) {
        if (i + 1 >= (size_t)n_tris * 3 || er[i].lo != er[i + 1].lo || er[i].hi != er[i + 1].hi) { ok = 0; break; }
        if (i + 2 < (size_t)n_tris * 3 && er[i + 2].lo == er[i].lo && er[i + 2].hi == er[i].hi) { ok = 0; break; }
        const int same = er[i].fwd == er[i + 1].fwd;
        nb[er[i].tri] = (er[i + 1].tri / 3) * 2 + same;
        nb[er[i + 1].tri] = (er[i].tri / 3) * 2 + same;
    }
    if (!ok) goto done;
    {
        int n_comp = 0;
        for (int t = 0; t < n_tris; t++) comp[t] = -1;
        for (int seed = 0; seed < n_tris && ok; seed++) {
            if (comp[seed] >= 0) continue;
            int head = 0, tail = 0;
            queue[tail++] = seed; comp[seed] = n_comp; flip[seed] = 0;
            while (head < tail && ok) {
                const int t = queue[head++];
                for (int k = 0; k < 3; k++) {
                    const int u = nb[3 * t + k] >> 1, want = flip[t] ^ (nb[3 * t + k] & 1);
                    if (comp[u] < 0) { comp[u] = n_comp; flip[u] = (uint8_t)want; queue[tail++] = u; }
                    else if (flip[u] != want) { ok = 0; break; }
                }
            }
            n_comp++;
        }
        if (!ok) goto done;
        vol = (double*)calloc((size_t)n_comp, sizeof(double));
        for (int t = 0; t < n_tris; t++) {
            const float* a = verts + 3 * (size_t)tris[3 * t]; const float* b = verts + 3 * (size_t)tris[3 * t + 1]; const float* c = verts + 3 * (size_t)tris[3 * t + 2];
            const double cx = (double)b[1] * c[2] - (double)b[2] * c[1], cy = (double)b[2] * c[0] - (double)b[0] * c[2], cz = (double)b[0] * c[1] - (double)b[1] * c[0];
            const double det = ((double)a[0] * cx + (double)a[1] * cy) + (double)a[2] * cz;
            vol[comp[t]] += (flip[t] ? -det : det) / 6.0;
        }
        for (int c = 0; c < n_comp; c++)
            if (!(vol[c] != 0.0) || !isfinite(vol[c])) ok = 0;
        if (ok)
            for (int t = 0; t < n_tris; t++)
                if (vol[comp[t]] < 0.0) flip[t] ^= 1;
    }
done:
    if (!ok) memset(flip, 0, (size_t)n_tris);
    free(wr); free(id); free(er); free(nb); free(comp); free(queue); free(vol);
    return ok;
}
