// k_flow.hip -- masked optical-flow + depth measurement of the velocity filter (gfx950).
//
// Reference: ImageOpticalFlowMeasurement<T>::freeze  include/ROFT/ImageOpticalFlowMeasurement.hpp:231-283
//   C      = cv::findNonZero(previous_segmentation_)            (row-major list)
//   cand   = C[0], C[R], C[2R], ...   R = subsampling radius (float-accumulated index, exact < 2^24)
//   keep   = flow finite and |.| < 1e9 (OpticalFlowUtilities.h:19-22) and 0 < Z < depth_maximum
//   y_i    = flow / scale;  H_i = T * [interaction matrix rows]  (:272-282)
//
// MI355X design: one workgroup per object.  The previous frame's `obj` bit plane (W*H/8 bytes) is
// staged in LDS once; row popcounts + a block scan give the row-major rank of every set bit, so
// candidate c is located by a binary search over the row prefix and a word walk -- no full-image
// pass over depth or flow: only the ~N_mask/R candidate pixels are gathered from HBM.
// Output order equals the reference's (candidates in rank order, invalid ones dropped).
#include "plane_rank.h"

namespace roft {

__device__ __forceinline__ void flow_at(const void* data, const DevFlowFmt& f, int row, int col, float& dx,
                                        float& dy)
{
    size_t idx = ((size_t)row * (size_t)f.cols + (size_t)col);
    if (f.type == ROFT_FLOW_S16C2) {
        short2 p = reinterpret_cast<const short2*>(data)[idx];
        dx = (float)p.x / f.scale;
        dy = (float)p.y / f.scale;
    } else {
        float2 p = reinterpret_cast<const float2*>(data)[idx];
        dx = p.x / f.scale;
        dy = p.y / f.scale;
    }
}

__device__ __forceinline__ bool is_flow_valid(float fx, float fy)
{
    return !isnan(fx) && !isnan(fy) && fabs((double)fx) < 1e9 && fabs((double)fy) < 1e9;
}

constexpr int kFlowThreads = 512;

// dynamic LDS: plane words [wpr*H] | rowpref [H+1]
__global__ __launch_bounds__(kFlowThreads) void flow_measure_kernel(EngineArrays a, double depth_max, int radius,
                                                                   int mask_finish)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_wave[17];
    const int obj = blockIdx.x;
    const FrameCtrl& c = a.ctrl[obj];
    ObjState& st = a.state[obj];
    // the mask stage's per-object bookkeeping rides along (saves a launch per frame); this kernel reads
    // none of the fields it touches
    if (mask_finish && threadIdx.x == 0) mask_bookkeeping(c, st);
    if (!c.vel_stage) {
        if (threadIdx.x == 0) st.n_flow_points = -1;
        return;
    }
    const int W = a.cam.W, H = a.cam.H, wpr = a.cam.wpr;
    uint32_t* s_plane = reinterpret_cast<uint32_t*>(smem);
    int* s_rowpref = reinterpret_cast<int*>(smem + ((a.plane_words * 4 + 15) & ~(size_t)15));

    // 1+2. stage the previous frame's obj plane in LDS, row popcounts -> exclusive row prefix
    const int M = stage_plane(a.planes + plane_offset(a, obj, c.slot_prev, 1), a.plane_words, H, wpr, s_plane, s_rowpref,
                              s_wave);
    const int C = (M + radius - 1) / radius;

    // 3. candidates, blocked assignment so that the compaction keeps rank order
    FlowRec* cand = a.cand + (size_t)obj * a.cand_cap;
    const int per = (C + blockDim.x - 1) / blockDim.x;
    const int c_begin = min(C, (int)threadIdx.x * per), c_end = min(C, c_begin + per);
    const float* depth = c.depth_prev;
    int n_valid = 0;
    for (int ci = c_begin; ci < c_end; ++ci) {
        int u, v;
        select_rank(s_plane, s_rowpref, H, wpr, ci * radius, u, v);

        const float z = depth[(size_t)v * W + u];
        float dx, dy;
        flow_at(c.flow[0], a.ffmt, v / a.ffmt.grid, u / a.ffmt.grid, dx, dy);
        const bool ok = is_flow_valid(dx, dy) && z > 0 && (double)z < depth_max;
        FlowRec r;
        r.u = ok ? u : -1; r.v = v; r.z = z; r.dx = dx; r.dy = dy;
        cand[ci] = r;
        n_valid += ok ? 1 : 0;
    }
    int total;
    int pos = block_exclusive_scan(n_valid, s_wave, &total);
    FlowRec* recs = a.recs + (size_t)obj * a.cand_cap;
    for (int ci = c_begin; ci < c_end; ++ci) {
        const FlowRec r = cand[ci];
        if (r.u >= 0) recs[pos++] = r;
    }
    if (threadIdx.x == 0) st.n_flow_points = total;
}

void launch_flow_measure(const EngineArrays& a, double depth_max, int radius, bool mask_finish, hipStream_t s)
{
    const size_t lds = plane_lds_bytes(a.plane_words, a.cam.H);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(flow_measure_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        attr_set = true;
    }
    hipLaunchKernelGGL(flow_measure_kernel, dim3(a.n_obj), dim3(kFlowThreads), lds, s, a, depth_max, radius,
                       mask_finish ? 1 : 0);
}

// ---- records -> (uv, y, H) exactly as the reference assembles them (hpp:258-283) -------------
__global__ __launch_bounds__(256) void expand_yh_kernel(const FlowRec* recs, const int* n, DevCamera cam, double dt,
                                                        int32_t* uv, double* y, double* Hm, int cap)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = min(*n, cap);
    if (i >= N) return;
    const FlowRec r = recs[i];
    uv[2 * i] = r.u;
    uv[2 * i + 1] = r.v;
    y[2 * i] = (double)r.dx;
    y[2 * i + 1] = (double)r.dy;
    const double z = (double)r.z;
    const double uu = (r.u - cam.cx);
    const double vv = (r.v - cam.cy);
    double* h = Hm + (size_t)12 * i;
    h[0] = (cam.fx / z) * dt;
    h[1] = 0.0 * dt;
    h[2] = (-uu / z) * dt;
    h[3] = (-uu * vv / cam.fy) * dt;
    h[4] = (cam.fx + uu * uu / cam.fx) * dt;
    h[5] = (-vv * cam.fx / cam.fy) * dt;
    h[6] = 0.0 * dt;
    h[7] = (cam.fy / z) * dt;
    h[8] = (-vv / z) * dt;
    h[9] = (-(cam.fy + vv * vv / cam.fy)) * dt;
    h[10] = (vv * uu / cam.fx) * dt;
    h[11] = (uu * cam.fy / cam.fx) * dt;
}

void launch_expand_yh(const FlowRec* recs, const int* n, DevCamera cam, double dt, int32_t* uv, double* y, double* H,
                      int cap, hipStream_t s)
{
    hipLaunchKernelGGL(expand_yh_kernel, dim3((cap + 255) / 256), dim3(256), 0, s, recs, n, cam, dt, uv, y, H, cap);
}

}  // namespace roft
