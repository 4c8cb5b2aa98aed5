/*
 * ro_render.c -- CPU oracle (test infrastructure): mesh depth render + masked depth likelihood.
 *
 * The reference renders with OpenGL (SICAD::superimpose, src/roft-lib/src/SICAD.cpp:924-1066;
 * projection :1634-1637; view :1723-1735 with the OpenGL->camera flip of ROFTFilter.cpp:198;
 * fragment shader src/roft-lib/shader/shader_model.frag:30-52).  GL is not available here, so this
 * is a restatement of the *contract* of that pipeline, not of a GPU driver's rasteriser:
 *   - tile of (W/d) x (H/d) pixels, intrinsics divided by d (ROFTFilter.cpp:191-197);
 *   - pixel (i, j) (column i, row j from the top) is sampled at its centre: the ray through
 *     u = i + 0.5, v = j + 0.5 with u = fx X / Z + cx, v = fy Y / Z + cy (this is what the
 *     projection matrix + vertical flip at SICAD.cpp:1052 amount to);
 *   - value = eye-space Z of the nearest surface (depth test LESS, both faces, no culling,
 *     SICAD.cpp:271-272), perspective-correct (1/Z interpolated linearly in screen space, which
 *     is what linearising gl_FragCoord.z yields); background = 0;
 *   - model matrix = translation * rotation(angle, axis) in float (SICAD.cpp:986-989).
 * Deviations (documented, only silhouette pixels are affected): edges are inclusive instead of
 * GL's top-left rule; triangles with a vertex at Z <= near (0.001) are dropped instead of clipped;
 * depth-buffer quantisation is not modelled.
 *
 * All per-pixel arithmetic is IEEE float with a fixed operation order (build with
 * -ffp-contract=off) so that an independent implementation following the same spec can be
 * compared bit for bit.  The order is chosen for the fewest divisions (round 5: an IEEE division is ~10
 * instructions on the GPU and the rasteriser is bound by instruction issue): one reciprocal per
 * vertex (u = (fx X) (1/Z) + cx), and the perspective-correct depth of a covered pixel as ONE quotient,
 *   z = 1 / sum_k b_k / z_k,  b_k = w_k / area   ==   area z0 z1 z2 / (w0 z1 z2 + w1 z0 z2 + w2 z0 z1),
 * instead of three reciprocals per triangle and four divisions per pixel -- the same contract (GL's own
 * interpolation is not specified to the bit either), different last bits than rounds 1 - 4 rendered (RO_RENDER_V1 below keeps that
 * arithmetic; tests/test_render_gap_cpu.py bounds the difference: <= 8 ulp at the 99th percentile, no outlier decision moves).
 * Range of the one-quotient form: num = area z0 z1 z2 and den = sum w_k z_i z_j are products of a screen area (pixels^2, <= ~1e6
 * on a 1280 x 720 tile, >= ~1e-9 for a triangle that still covers a pixel centre) and two or three depths in METRES (the unit of
 * the depth images and of the 0.001 near plane: 0.001 .. ~10): 1e-18 .. 1e9, far inside float's normal range; a mesh given in
 * millimetres would be wrong for the near plane and the depth gate long before it is wrong here (z^3 ~ 1e9, num <= 1e15).
 *
 * Likelihood: ROFTFilter::pick_best_alternative  src/roft-lib/src/ROFTFilter.cpp:553-577.
 */
#include "roft_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* The rasteriser in three arithmetic modes.  RO_RENDER_CONTRACT is the render contract (what the HIP kernel reproduces bit for
 * bit); the other two exist ONLY to bound how far that contract sits from (a) what rounds 1 - 4 rendered and (b) what the
 * reference's OpenGL pipeline computes -- tests/test_render_gap_cpu.py, DESIGN.md section 3.  Never used by the product.
 *
 * RO_RENDER_V1: the formulation of rounds 1 - 4 (u = (fx X) / Z + cx; three reciprocals per triangle, normalised barycentric
 *   weights, z = 1 / sum b_k / z_k).
 * RO_RENDER_GL: the numerics of the reference's pipeline as far as the GL specification fixes them:
 *   - per vertex z_ndc = ((f + n) / (f - n) * Z - 2 f n / (f - n)) / Z in float (the third row of SICAD.cpp:1634-1637's
 *     projection applied to the eye-space point (X, -Y, -Z, 1); near 0.001, far 1000), window z = 0.5 z_ndc + 0.5;
 *   - window z interpolated LINEARLY in screen space (it is affine there; plane equation in double), handed to the fragment
 *     shader as a float gl_FragCoord.z;
 *   - depth test LESS (SICAD.cpp:271-272) on the window z quantised to the 24-bit normalised depth buffer the reference
 *     allocates (GL_DEPTH_COMPONENT, SICAD.cpp:260), first fragment wins ties;
 *   - the fragment's output is shader_model.frag:33-51 in float: z = 2 z_w - 1; (2 n f) / (f + n - z (f - n))
 *     -- at Z = 0.7 m one float ulp of z_w is 0.03 mm of depth;
 *   - coverage by the top-left rule (a pixel centre exactly on an edge belongs to the triangle only if the edge is a left edge,
 *     or a horizontal top edge, in image orientation).
 *   Not modelled (implementation-defined in GL): sub-pixel snapping of vertex positions, near-plane clipping of triangles
 *   that cross Z = 0.001 (dropped here like in the contract; no tracked object comes within a millimetre of the camera). */
static void render_mode(const ro_mesh* mesh, const double x[3], const double q[4], const ro_camera* cam, int divider,
                        float* tile, int mode)
{
    const int w = cam->width / divider, h = cam->height / divider;
    const float fx = (float)(cam->fx / divider), fy = (float)(cam->fy / divider);
    const float cx = (float)(cam->cx / divider), cy = (float)(cam->cy / divider);
    double Rd[9];
    ro_quat_to_rotmat(q, Rd);
    float R[9], t[3];
    for (int i = 0; i < 9; i++) R[i] = (float)Rd[i];
    for (int i = 0; i < 3; i++) t[i] = (float)x[i];

    const size_t npix = (size_t)w * h;
    for (size_t i = 0; i < npix; i++) tile[i] = INFINITY;
    /* contract, closed mesh (ro_meshclass.c): triangles that face away are not drawn -- provided every vertex is in front of the
     * near plane (a camera inside the surface would have vertices behind it) */
    const uint8_t* flip = NULL;
    uint8_t* flip_own = NULL;
    if (mode == RO_RENDER_CONTRACT) {
        if (mesh->closed > 0) flip = mesh->tri_flip;
        else if (mesh->closed < 0) {
            flip_own = (uint8_t*)malloc((size_t)(mesh->n_tris > 0 ? mesh->n_tris : 1));
            if (ro_mesh_classify(mesh->verts, mesh->n_verts, mesh->tris, mesh->n_tris, flip_own)) flip = flip_own;
        }
    }
    uint32_t* zq = NULL;   /* GL: the quantised depth buffer */
    if (mode == RO_RENDER_GL) {
        zq = (uint32_t*)malloc(sizeof(uint32_t) * npix);
        for (size_t i = 0; i < npix; i++) zq[i] = 0xFFFFFFu;   /* glClear: depth 1.0 */
    }
    const float gl_near = 0.001f, gl_far = 1000.0f;
    const float gl_a = (gl_far + gl_near) / (gl_far - gl_near), gl_b = (2.0f * (gl_far * gl_near)) / (gl_far - gl_near);

    /* camera-frame vertices and their projections */
    float* cam_z = (float*)malloc(sizeof(float) * mesh->n_verts);
    float* sx = (float*)malloc(sizeof(float) * mesh->n_verts);
    float* sy = (float*)malloc(sizeof(float) * mesh->n_verts);
    float* zw = (float*)malloc(sizeof(float) * mesh->n_verts);
    for (int i = 0; i < mesh->n_verts; i++) {
        const float* p = mesh->verts + (size_t)3 * i;
        float X = ((R[0] * p[0] + R[1] * p[1]) + R[2] * p[2]) + t[0];
        float Y = ((R[3] * p[0] + R[4] * p[1]) + R[5] * p[2]) + t[1];
        float Z = ((R[6] * p[0] + R[7] * p[1]) + R[8] * p[2]) + t[2];
        cam_z[i] = Z;
        zw[i] = 0.0f;
        if (!(Z > 0.001f)) flip = NULL;
        if (Z > 0.001f) {
            if (mode == RO_RENDER_CONTRACT) {
                const float iZ = 1.0f / Z;
                sx[i] = (fx * X) * iZ + cx;
                sy[i] = (fy * Y) * iZ + cy;
            } else {
                sx[i] = (fx * X) / Z + cx;
                sy[i] = (fy * Y) / Z + cy;
                /* clip z = -(f+n)/(f-n) * (-Z) - 2fn/(f-n), clip w = Z */
                const float z_ndc = (gl_a * Z - gl_b) / Z;
                zw[i] = 0.5f * z_ndc + 0.5f;
            }
        } else {
            sx[i] = sy[i] = 0.0f;
        }
    }

    for (int k = 0; k < mesh->n_tris; k++) {
        const int32_t* tri = mesh->tris + (size_t)3 * k;
        const int i0 = tri[0], i1 = tri[1], i2 = tri[2];
        const float z0 = cam_z[i0], z1 = cam_z[i1], z2 = cam_z[i2];
        if (!(z0 > 0.001f && z1 > 0.001f && z2 > 0.001f)) continue;
        const float x0 = sx[i0], y0 = sy[i0], x1 = sx[i1], y1 = sy[i1], x2 = sx[i2], y2 = sy[i2];
        const float area = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0);
        if (area == 0.0f || !(area == area)) continue;
        if (flip && ((area < 0.0f) == (flip[k] != 0))) continue;   /* faces away */
        float minx = fminf(x0, fminf(x1, x2)), maxx = fmaxf(x0, fmaxf(x1, x2));
        float miny = fminf(y0, fminf(y1, y2)), maxy = fmaxf(y0, fmaxf(y1, y2));
        /* pixel centres i + 0.5 inside [min, max] */
        float fi0 = ceilf(minx - 0.5f), fi1 = floorf(maxx - 0.5f);
        float fj0 = ceilf(miny - 0.5f), fj1 = floorf(maxy - 0.5f);
        if (fi0 < 0.0f) fi0 = 0.0f;
        if (fj0 < 0.0f) fj0 = 0.0f;
        if (fi1 > (float)(w - 1)) fi1 = (float)(w - 1);
        if (fj1 > (float)(h - 1)) fj1 = (float)(h - 1);
        if (!(fi0 <= fi1) || !(fj0 <= fj1)) continue;
        const int ia = (int)fi0, ib = (int)fi1, ja = (int)fj0, jb = (int)fj1;
        const float p12 = z1 * z2, p02 = z0 * z2, p01 = z0 * z1;
        const float num = area * (z0 * p12);
        const float iz0 = 1.0f / z0, iz1 = 1.0f / z1, iz2 = 1.0f / z2;   /* (V1) */
        /* (GL) top-left rule: with the weights oriented so that the interior is w > 0, edge k is w_k = a_k px + b_k py + c_k;
         * a centre ON the edge is covered iff a_k > 0 (left edge) or a_k == 0 and b_k > 0 (top edge, image rows grow downwards) */
        const float sgn = (area > 0.0f) ? 1.0f : -1.0f;
        const float ea[3] = {-sgn * (y2 - y1), -sgn * (y0 - y2), -sgn * (y1 - y0)};
        const float eb[3] = {sgn * (x2 - x1), sgn * (x0 - x2), sgn * (x1 - x0)};
        int owns[3];
        for (int e = 0; e < 3; e++) owns[e] = (ea[e] > 0.0f) || (ea[e] == 0.0f && eb[e] > 0.0f);
        for (int j = ja; j <= jb; j++) {
            const float py = (float)j + 0.5f;
            for (int i = ia; i <= ib; i++) {
                const float px = (float)i + 0.5f;
                /* edge functions; w_k is the weight of vertex k */
                float w0 = (x2 - x1) * (py - y1) - (y2 - y1) * (px - x1);
                float w1 = (x0 - x2) * (py - y2) - (y0 - y2) * (px - x2);
                float w2 = (x1 - x0) * (py - y0) - (y1 - y0) * (px - x0);
                int inside = (area > 0.0f) ? (w0 >= 0.0f && w1 >= 0.0f && w2 >= 0.0f)
                                           : (w0 <= 0.0f && w1 <= 0.0f && w2 <= 0.0f);
                if (!inside) continue;
                float* dst = tile + (size_t)j * w + i;
                if (mode == RO_RENDER_CONTRACT) {
                    float den = (w0 * p12 + w1 * p02) + w2 * p01;
                    float z = num / den;
                    if (!(z > 0.0f)) continue;
                    if (z < *dst) *dst = z;
                } else if (mode == RO_RENDER_V1) {
                    float b0 = w0 / area, b1 = w1 / area, b2 = w2 / area;
                    float iz = (b0 * iz0 + b1 * iz1) + b2 * iz2;
                    float z = 1.0f / iz;
                    if (!(z > 0.0f)) continue;
                    if (z < *dst) *dst = z;
                } else {
                    if ((w0 == 0.0f && !owns[0]) || (w1 == 0.0f && !owns[1]) || (w2 == 0.0f && !owns[2])) continue;
                    /* the plane equation of window z evaluated in double (fixed-function interpolators carry more than the
                     * depth buffer's precision), delivered to the shader as a float gl_FragCoord.z */
                    const double dw0 = ((double)x2 - x1) * ((double)py - y1) - ((double)y2 - y1) * ((double)px - x1);
                    const double dw1 = ((double)x0 - x2) * ((double)py - y2) - ((double)y0 - y2) * ((double)px - x2);
                    const double dw2 = ((double)x1 - x0) * ((double)py - y0) - ((double)y1 - y0) * ((double)px - x0);
                    const float z_w = (float)((dw0 * zw[i0] + dw1 * zw[i1] + dw2 * zw[i2]) / (dw0 + dw1 + dw2));
                    if (!(z_w >= 0.0f && z_w <= 1.0f)) continue;   /* clipped by the depth range */
                    const uint32_t qz = (uint32_t)floor((double)z_w * 16777215.0 + 0.5);
                    uint32_t* zd = zq + (size_t)j * w + i;
                    if (!(qz < *zd)) continue;   /* GL_LESS */
                    *zd = qz;
                    const float zn = z_w * 2.0f - 1.0f;
                    *dst = (2.0f * gl_near * gl_far) / (gl_far + gl_near - zn * (gl_far - gl_near));
                }
            }
        }
    }
    for (size_t i = 0; i < npix; i++)
        if (tile[i] == INFINITY) tile[i] = 0.0f;
    free(cam_z);
    free(sx);
    free(sy);
    free(zw);
    free(zq);
    free(flip_own);
}

void ro_render_depth(const ro_mesh* mesh, const double x[3], const double q[4],
                     const ro_camera* cam, int divider, float* tile)
{
    render_mode(mesh, x, q, cam, divider, tile, RO_RENDER_CONTRACT);
}

void ro_render_depth_mode(const ro_mesh* mesh, const double x[3], const double q[4],
                          const ro_camera* cam, int divider, float* tile, int mode)
{
    render_mode(mesh, x, q, cam, divider, tile, mode);
}

double ro_depth_likelihood(const ro_camera* cam, const float* depth, const uint8_t* mask,
                           const float* tile, int divider, long* samples_out)
{
    const int W = cam->width, H = cam->height, w = W / divider;
    double error = 0.0;
    long samples = 0;
    size_t rank = 0;
    /* cv::findNonZero order, every second entry (`k += 2`, ROFTFilter.cpp:556) */
    for (int v = 0; v < H; v++) {
        for (int u = 0; u < W; u++) {
            if (mask[(size_t)v * W + u] == 0) continue;
            if ((rank & 1) == 0) {
                float d = depth[(size_t)v * W + u];
                float r = tile[(size_t)(v / divider) * w + (u / divider)];
                if ((d > 0) && (d < 2.0) && (r != 0.0)) {
                    error += fabsf(d - r);
                    samples++;
                }
            }
            rank++;
        }
    }
    if (samples_out) *samples_out = samples;
    if (samples == 0) return DBL_MAX;
    /* outlier_rejection_gain_ is a `const bool` (ROFTFilter.h:64): 0.01 -> true -> 1.0 */
    return (error / samples) / 1.0;
}
