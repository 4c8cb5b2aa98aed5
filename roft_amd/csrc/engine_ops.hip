// engine_ops.hip -- the operator-level entry points of include/roft_engine.h (section 1): each runs the engine's own kernels on a
// private one-object context on device 0.
#include "engine_internal.h"

// =================================================================================================
// operator level: one-object context on device 0
// =================================================================================================
namespace {

struct OpCtx {
    std::mutex mu;
    hipStream_t stream = nullptr;
    Arrays arr;
    int W = 0, H = 0, ftype = 0, fgrid = 0, radius = 0;
    DevBuf<unsigned char> b0, b1, b2, b3, b4, b5, bflip;  // generic scratch
    bool ready = false;

    int prepare(const roft_camera& cam, int ftype_, int fgrid_, float fscale, int radius_)
    {
        if (roft_device_count() <= 0) return fail(ROFT_ERR_DEVICE, "no HIP device (libroft_hip has no CPU path)");
        if (int rc = check_geometry(cam.width, cam.height)) return rc;
        HIP_TRY(hipSetDevice(0));
        (void)hipGetLastError();   // a stale error of another library on this thread is not this call's
        if (!stream) HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        DevFlowFmt ff;
        ff.type = ftype_;
        ff.grid = std::max(fgrid_, 1);
        ff.cols = cam.width / ff.grid;
        ff.rows = cam.height / ff.grid;
        ff.scale = fscale;
        if (!ready || W != cam.width || H != cam.height || radius != radius_) {
            if (int rc = arr.alloc(1, 1, make_cam(cam), ff, std::max(radius_, 1))) return rc;
            W = cam.width; H = cam.height; radius = radius_;
            ready = true;
        }
        arr.a.cam = make_cam(cam);
        arr.a.ffmt = ff;
        arr.a.n_obj = 1;
        arr.a.T = 1;
        ObjState st;
        init_state(st);
        HIP_TRY(hipMemcpyAsync(arr.state.p, &st, sizeof(st), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return ROFT_OK;
    }
};

OpCtx& op()
{
    static OpCtx c;
    return c;
}

int upload_ctrl(OpCtx& c, const FrameCtrl& fc)
{
    HIP_TRY(hipMemcpyAsync(c.arr.ctrl.p, &fc, sizeof(fc), hipMemcpyHostToDevice, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));  // fc lives on the caller's stack
    return ROFT_OK;
}

template <class T>
int to_dev(DevBuf<unsigned char>& b, const T* src, size_t count, hipStream_t s)
{
    HIP_TRY(b.ensure(std::max<size_t>(count * sizeof(T), 16)));
    if (count) HIP_TRY(hipMemcpyAsync(b.p, src, count * sizeof(T), hipMemcpyHostToDevice, s));
    return ROFT_OK;
}

}  // namespace

extern "C" {

int roft_flow_measurement(const roft_camera* cam, const uint8_t* prev_mask, const float* prev_depth,
                          const roft_flow* flow, double dt, float radius, double depth_max, int capacity,
                          int32_t* uv, double* y, double* H, int* n_out)
{
    if (!cam || !prev_mask || !prev_depth || !flow || !flow->data || !n_out) return fail(ROFT_ERR_INVALID, "null argument");
    OpCtx& c = op();
    std::lock_guard<std::mutex> lk(c.mu);
    const int r = (int)(size_t)radius;
    if (r <= 0) return fail(ROFT_ERR_INVALID, "radius must be >= 1");
    if (int rc = c.prepare(*cam, flow->type, flow->grid, flow->scale, r)) return rc;
    const size_t npix = (size_t)cam->width * cam->height;
    if (int rc = to_dev(c.b0, prev_mask, npix, c.stream)) return rc;
    if (int rc = to_dev(c.b1, prev_depth, npix, c.stream)) return rc;
    if (int rc = to_dev(c.b2, (const unsigned char*)flow->data, flow_bytes(c.arr.a.ffmt), c.stream)) return rc;
    FrameCtrl fc;
    clear_ctrl(fc);
    fc.dt = dt;
    fc.has_new_mask = 1;
    fc.new_mask = c.b0.p;
    fc.slot_prev = kSlotNew;  // the ingested planes are "the previous frame's mask"
    fc.slot_cur = 0;
    fc.depth_prev = reinterpret_cast<const float*>(c.b1.p);
    fc.flow[0] = c.b2.p;
    fc.vel_stage = 1;
    if (int rc = upload_ctrl(c, fc)) return rc;
    launch_mask_ingest(c.arr.a, 0, c.stream);
    launch_flow_measure(c.arr.a, depth_max, r, c.stream);
    int n = 0;
    HIP_TRY(hipMemcpyAsync(&n, c.arr.a.npts, sizeof(int), hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    HIP_TRY(hipGetLastError());
    *n_out = n;
    if (n > capacity) return fail(ROFT_ERR_CAPACITY, "more flow points than the caller's capacity");
    if (n > 0 && uv && y && H) {
        HIP_TRY(c.b3.ensure(sizeof(int32_t) * 2 * n));
        HIP_TRY(c.b4.ensure(sizeof(double) * 2 * n));
        HIP_TRY(c.b5.ensure(sizeof(double) * 12 * n));
        launch_expand_yh(c.arr.a.recs, c.arr.a.npts, c.arr.a.cam, dt, reinterpret_cast<int32_t*>(c.b3.p),
                         reinterpret_cast<double*>(c.b4.p), reinterpret_cast<double*>(c.b5.p), n, c.stream);
        HIP_TRY(hipMemcpyAsync(uv, c.b3.p, sizeof(int32_t) * 2 * n, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipMemcpyAsync(y, c.b4.p, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipMemcpyAsync(H, c.b5.p, sizeof(double) * 12 * n, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipStreamSynchronize(c.stream));
    }
    // leave the one-object context clean for the next call
    ObjState st;
    init_state(st);
    HIP_TRY(hipMemcpy(c.arr.state.p, &st, sizeof(st), hipMemcpyHostToDevice));
    return ROFT_OK;
}

static int op_simple_prepare(OpCtx& c)
{
    if (roft_device_count() <= 0) return fail(ROFT_ERR_DEVICE, "no HIP device (libroft_hip has no CPU path)");
    HIP_TRY(hipSetDevice(0));
    (void)hipGetLastError();   // a stale error of another library on this thread is not this call's
    if (!c.stream) HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    return ROFT_OK;
}

int roft_kf_predict(const double x[6], const double P[36], const double Qdiag[6], double x_out[6], double P_out[36])
{
    if (!x || !P || !Qdiag || !x_out || !P_out) return fail(ROFT_ERR_INVALID, "null argument");
    OpCtx& c = op();
    std::lock_guard<std::mutex> lk(c.mu);
    if (int rc = op_simple_prepare(c)) return rc;
    double in[48];
    std::memcpy(in, x, 48);
    std::memcpy(in + 6, P, 288);
    std::memcpy(in + 42, Qdiag, 48);
    if (int rc = to_dev(c.b0, in, 48, c.stream)) return rc;
    HIP_TRY(c.b1.ensure(sizeof(double) * 42));
    double* d = reinterpret_cast<double*>(c.b0.p);
    double* o = reinterpret_cast<double*>(c.b1.p);
    launch_kf_predict(d, d + 6, d + 42, o, o + 6, c.stream);
    double out[42];
    HIP_TRY(hipMemcpyAsync(out, o, sizeof(out), hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    std::memcpy(x_out, out, 48);
    std::memcpy(P_out, out + 6, 288);
    return ROFT_OK;
}

int roft_skf_correct(const double x_pred[6], const double P_pred[36], int N, const double* y, const double* H,
                     const double Rdiag[2], int reweight, double x_out[6], double P_out[36], int* status_out)
{
    if (!x_pred || !P_pred || !Rdiag || !x_out || !P_out || (N > 0 && (!y || !H))) return fail(ROFT_ERR_INVALID, "null argument");
    OpCtx& c = op();
    std::lock_guard<std::mutex> lk(c.mu);
    if (int rc = op_simple_prepare(c)) return rc;
    double in[44];
    std::memcpy(in, x_pred, 48);
    std::memcpy(in + 6, P_pred, 288);
    in[42] = Rdiag[0]; in[43] = Rdiag[1];
    if (int rc = to_dev(c.b0, in, 44, c.stream)) return rc;
    const int n = std::max(N, 0);
    if (int rc = to_dev(c.b1, y, (size_t)2 * n, c.stream)) return rc;
    if (int rc = to_dev(c.b2, H, (size_t)12 * n, c.stream)) return rc;
    HIP_TRY(c.b3.ensure(sizeof(double) * 3 * std::max(n, 1)));
    HIP_TRY(c.b4.ensure(sizeof(double) * 44));
    double* d = reinterpret_cast<double*>(c.b0.p);
    double* o = reinterpret_cast<double*>(c.b4.p);
    launch_skf_arrays(d, d + 6, N, reinterpret_cast<double*>(c.b1.p), reinterpret_cast<double*>(c.b2.p), d + 42, reweight,
                      reinterpret_cast<double*>(c.b3.p), o, o + 6, reinterpret_cast<int*>(o + 42), c.stream);
    double out[44];
    HIP_TRY(hipMemcpyAsync(out, o, sizeof(out), hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    HIP_TRY(hipGetLastError());
    std::memcpy(x_out, out, 48);
    std::memcpy(P_out, out + 6, 288);
    if (status_out) std::memcpy(status_out, out + 42, sizeof(int));
    return ROFT_OK;
}

int roft_skf_correct_points(const roft_camera* cam, double dt, const double x_pred[6], const double P_pred[36], int N,
                            const int32_t* uv, const float* z, const float* flow_xy, const double Rdiag[2], int reweight,
                            double x_out[6], double P_out[36], int* status_out)
{
    if (!cam || !x_pred || !P_pred || !Rdiag || !x_out || !P_out || (N > 0 && (!uv || !z || !flow_xy)))
        return fail(ROFT_ERR_INVALID, "null argument");
    OpCtx& c = op();
    std::lock_guard<std::mutex> lk(c.mu);
    if (int rc = op_simple_prepare(c)) return rc;
    double in[44];
    std::memcpy(in, x_pred, 48);
    std::memcpy(in + 6, P_pred, 288);
    in[42] = Rdiag[0]; in[43] = Rdiag[1];
    if (int rc = to_dev(c.b0, in, 44, c.stream)) return rc;
    const int n = std::max(N, 0);
    std::vector<FlowRec> recs(n);
    for (int i = 0; i < n; ++i) recs[i] = FlowRec{uv[2 * i], uv[2 * i + 1], z[i], flow_xy[2 * i], flow_xy[2 * i + 1]};
    if (int rc = to_dev(c.b1, recs.data(), (size_t)n, c.stream)) return rc;
    HIP_TRY(c.b3.ensure(sizeof(double) * 3 * std::max(n, 1)));
    HIP_TRY(c.b4.ensure(sizeof(double) * 44));
    double* d = reinterpret_cast<double*>(c.b0.p);
    double* o = reinterpret_cast<double*>(c.b4.p);
    launch_skf_records(d, d + 6, N, reinterpret_cast<const FlowRec*>(c.b1.p), make_cam(*cam), dt, d + 42, reweight,
                       reinterpret_cast<double*>(c.b3.p), o, o + 6, reinterpret_cast<int*>(o + 42), c.stream);
    double out[44];
    HIP_TRY(hipMemcpyAsync(out, o, sizeof(out), hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));   // (also keeps `recs` alive until its upload has been read)
    HIP_TRY(hipGetLastError());
    std::memcpy(x_out, out, 48);
    std::memcpy(P_out, out + 6, 288);
    if (status_out) std::memcpy(status_out, out + 42, sizeof(int));
    return ROFT_OK;
}

int roft_mask_propagate(uint8_t* mask, int W, int H, const roft_flow* flows, int n_flows, int frames_between)
{
    if (!mask || (n_flows > 0 && !flows)) return fail(ROFT_ERR_INVALID, "null argument");
    int start = 0;
    if (frames_between > 0) start = std::max(0, n_flows - frames_between);
    const int used = n_flows - start;
    if (used > kMaxFlowHist) return fail(ROFT_ERR_INVALID, "more than ROFT_MAX_FLOW_CHASE flow frames per propagation are not supported");
    OpCtx& c = op();
    std::lock_guard<std::mutex> lk(c.mu);
    roft_camera cam{W, H, 1.0, 1.0, 0.0, 0.0};
    const roft_flow* f0 = used > 0 ? &flows[start] : nullptr;
    if (int rc = c.prepare(cam, f0 ? f0->type : ROFT_FLOW_F32C2, f0 ? f0->grid : 1, f0 ? f0->scale : 1.0f, 35)) return rc;
    const size_t npix = (size_t)W * H;
    const size_t fb = flow_bytes(c.arr.a.ffmt);
    if (int rc = to_dev(c.b0, mask, npix, c.stream)) return rc;
    HIP_TRY(c.b1.ensure(fb * std::max(used, 1)));
    FrameCtrl fc;
    clear_ctrl(fc);
    for (int j = 0; j < used; ++j) {
        const roft_flow& f = flows[start + j];
        if (f.type != f0->type || f.cols != f0->cols || f.rows != f0->rows || !f.data)
            return fail(ROFT_ERR_INVALID, "all flow frames must share one format");
        HIP_TRY(hipMemcpyAsync(c.b1.p + fb * j, f.data, fb, hipMemcpyHostToDevice, c.stream));
        fc.flow[used - 1 - j] = c.b1.p + fb * j;  // [0] = newest
    }
    fc.has_new_mask = 1;
    fc.new_mask = c.b0.p;
    fc.force_mode = 3;
    fc.n_hist = used;
    fc.slot_prev = 1;
    fc.slot_cur = 0;
    fc.flow_valid = 0;
    // decide_mode() uses fbuf_n + flow_valid as the number of buffered flows: state carried in = `used` buffered flows
    MaskRec rec0[2];
    std::memset(rec0, 0, sizeof(rec0));
    rec0[0].fbuf_n = used;
    HIP_TRY(hipMemcpyAsync(c.arr.mrec.p, rec0, sizeof(rec0), hipMemcpyHostToDevice, c.stream));
    // the chain kernel ORs into a zeroed destination (inside the engine the frame before leaves it zeroed)
    HIP_TRY(hipMemsetAsync(c.arr.a.planes + plane_offset(c.arr.a, 0, 0, 0), 0, sizeof(uint32_t) * 2 * c.arr.a.plane_words, c.stream));
    if (int rc = upload_ctrl(c, fc)) return rc;
    c.arr.a.mrec_carry = c.arr.mrec.p;   // row 0: rec0[0]
    launch_mask_reset(c.arr.a, c.stream);
    launch_mask_ingest(c.arr.a, 0, c.stream);
    launch_mask_chain(c.arr.a, frames_between, 1, 1u, c.stream);
    HIP_TRY(c.b2.ensure(npix));
    launch_planes_to_mask(c.arr.a.planes + plane_offset(c.arr.a, 0, 0, 0), c.arr.a.planes + plane_offset(c.arr.a, 0, 0, 1),
                          (int)npix, c.b2.p, c.stream);
    HIP_TRY(hipMemcpyAsync(mask, c.b2.p, npix, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    HIP_TRY(hipGetLastError());
    return ROFT_OK;
}

int roft_pose_process_noise(const double psd[3], const double sig_w[3], double T, double Q[81])
{
    if (!psd || !sig_w || !Q) return fail(ROFT_ERR_INVALID, "null argument");
    // parameter packing only (CartesianQuaternionModel.cpp:127-141); the filter kernels build Q(T) themselves
    std::memset(Q, 0, sizeof(double) * 81);
    for (int i = 0; i < 3; ++i) {
        Q[i * 9 + i] = psd[i] * T;
        Q[(3 + i) * 9 + (3 + i)] = sig_w[i];
        Q[(6 + i) * 9 + (6 + i)] = psd[i] * (std::pow(T, 3.0) / 3.0);
        Q[i * 9 + (6 + i)] = psd[i] * (std::pow(T, 2.0) / 2.0);
        Q[(6 + i) * 9 + i] = psd[i] * (std::pow(T, 2.0) / 2.0);
    }
    return ROFT_OK;
}

static int op_ukf(const double mean[13], const double P[144], const double* Q81, double T, int type, const double* meas,
                  const double* Rdiag, const roft_ut_params* ut, double mean_out[13], double P_out[144], int* status)
{
    OpCtx& c = op();
    std::lock_guard<std::mutex> lk(c.mu);
    roft_camera cam{64, 64, 1.0, 1.0, 0.0, 0.0};
    if (!c.ready) { if (int rc = c.prepare(cam, ROFT_FLOW_F32C2, 1, 1.0f, 35)) return rc; }
    else { if (int rc = op_simple_prepare(c)) return rc; }
    ObjState* st = new ObjState();
    init_state(*st);
    std::memcpy(st->belief[B_CORR].mean, mean, sizeof(double) * 13);
    std::memcpy(st->belief[B_CORR].cov, P, sizeof(double) * 144);
    ObjParams prm;
    std::memset(&prm, 0, sizeof(prm));
    FrameCtrl fc;
    clear_ctrl(fc);
    fc.dt = T;
    fc.n_steps = 1;
    StepDesc& sd = fc.steps[0];
    sd.op = 1;
    sd.src = B_CORR;
    if (Q81) {
        if (int rc = to_dev(c.b0, Q81, 81, c.stream)) { delete st; return rc; }
        prm.q_override = reinterpret_cast<const double*>(c.b0.p);
        sd.do_predict = 1;
        sd.n_corr = 0;
        sd.dst[0] = B_SPARE;
    } else {
        sd.do_predict = 0;
        sd.n_corr = 1;
        sd.type[0] = type;
        sd.dst[0] = B_SPARE;
        sd.twist_slot = 0;
        int k = 0;
        const bool has_vel = (type == ROFT_MEAS_VELOCITY || type == ROFT_MEAS_POSE_VELOCITY);
        const bool has_pose = (type == ROFT_MEAS_POSE || type == ROFT_MEAS_POSE_VELOCITY);
        if (has_vel) {
            for (int i = 0; i < 6; ++i) st->twist_hist[0][i] = meas[i];
            for (int i = 0; i < 3; ++i) prm.R_v[i] = Rdiag[k++];
            for (int i = 0; i < 3; ++i) prm.R_w[i] = Rdiag[k++];
        }
        if (has_pose) {
            const double* pm = meas + (has_vel ? 6 : 0);
            for (int i = 0; i < 3; ++i) fc.pose_x[i] = pm[i];
            for (int i = 0; i < 4; ++i) fc.pose_q[i] = pm[3 + i];
            for (int i = 0; i < 3; ++i) prm.R_x[i] = Rdiag[k++];
            for (int i = 0; i < 3; ++i) prm.R_q[i] = Rdiag[k++];
        }
    }
    hipError_t err = hipMemcpyAsync(c.arr.state.p, st, sizeof(ObjState), hipMemcpyHostToDevice, c.stream);
    if (err == hipSuccess) err = hipMemcpyAsync(c.arr.params.p, &prm, sizeof(prm), hipMemcpyHostToDevice, c.stream);
    if (err == hipSuccess) err = hipStreamSynchronize(c.stream);
    if (err != hipSuccess) { delete st; HIP_TRY(err); }
    c.arr.a.n_obj = 1;
    if (int rc = upload_ctrl(c, fc)) { delete st; return rc; }
    launch_ukf_chain(c.arr.a, *ut, true, 0, c.stream);
    err = hipMemcpyAsync(st, c.arr.state.p, sizeof(ObjState), hipMemcpyDeviceToHost, c.stream);
    if (err == hipSuccess) err = hipStreamSynchronize(c.stream);
    if (err == hipSuccess) err = hipGetLastError();
    if (err == hipSuccess) {
        std::memcpy(mean_out, st->belief[B_SPARE].mean, sizeof(double) * 13);
        std::memcpy(P_out, st->belief[B_SPARE].cov, sizeof(double) * 144);
        if (status) *status = st->lane[0].ukf_status & 0xF;
    }
    delete st;
    HIP_TRY(err);
    return ROFT_OK;
}

int roft_ukf_predict(const double mean[13], const double P[144], const double Q[81], double T, const roft_ut_params* ut,
                     double mean_out[13], double P_out[144])
{
    if (!mean || !P || !Q || !ut || !mean_out || !P_out) return fail(ROFT_ERR_INVALID, "null argument");
    return op_ukf(mean, P, Q, T, 0, nullptr, nullptr, ut, mean_out, P_out, nullptr);
}

int roft_ukf_correct(const double mean[13], const double P[144], int type, const double* meas, const double* Rdiag,
                     const roft_ut_params* ut, double mean_out[13], double P_out[144], int* status_out)
{
    if (!mean || !P || !ut || !mean_out || !P_out) return fail(ROFT_ERR_INVALID, "null argument");
    if (type != ROFT_MEAS_NONE && (!meas || !Rdiag)) return fail(ROFT_ERR_INVALID, "null measurement");
    if (type < ROFT_MEAS_NONE || type > ROFT_MEAS_POSE_VELOCITY) return fail(ROFT_ERR_INVALID, "bad measurement type");
    return op_ukf(mean, P, nullptr, 0.0, type, meas, Rdiag, ut, mean_out, P_out, status_out);
}

// The engine's outlier test on a one-object context: features of (depth, mask) buffered by features_kernel, both
// alternatives rendered and scored by outlier_fused_kernel, the decision taken by the pose chain segment that follows -- the
// three launches roft_step enqueues at a pose arrival.  depth / mask may be null (render only: no samples).
static int op_outlier(const roft_camera* cam, int divider, const float* depth, const uint8_t* mask, const roft_mesh* mesh,
                      const double* x2 /*2x3*/, const double* q2 /*2x4*/, const OutlierLaunchOpts& o_in, double L_out[2],
                      long samples_out[2], int* selected_out, float* tiles_out)
{
    OpCtx& c = op();
    std::lock_guard<std::mutex> lk(c.mu);
    if (int rc = c.prepare(*cam, ROFT_FLOW_F32C2, 1, 1.0f, 35)) return rc;
    EngineArrays& a = c.arr.a;
    a.cam.divider = divider;
    a.tile_w = cam->width / divider;
    a.tile_h = cam->height / divider;
    const size_t npix = (size_t)cam->width * cam->height, tpix = (size_t)a.tile_w * a.tile_h;
    if (int rc = c.arr.ensure_zmerge(1, tpix)) return rc;
    auto restore = [&]() {   // default tile geometry of this context
        a.cam = make_cam(*cam);
        a.tile_w = cam->width / a.cam.divider;
        a.tile_h = cam->height / a.cam.divider;
        a.max_verts = a.max_tris = 0;
    };
    for (size_t i = 0; i < (size_t)3 * mesh->n_tris; ++i)
        if (mesh->tris[i] < 0 || mesh->tris[i] >= mesh->n_verts) return fail(ROFT_ERR_INVALID, "mesh: a triangle refers to a vertex outside the vertex array");
    PreparedMesh pm;
    prepare_mesh(mesh->verts, mesh->n_verts, mesh->tris, mesh->n_tris, pm);
    if (int rc = to_dev(c.b0, mesh->verts, (size_t)3 * mesh->n_verts, c.stream)) return rc;
    if (int rc = to_dev(c.b1, pm.tris(mesh->tris), (size_t)3 * mesh->n_tris, c.stream)) return rc;
    if (pm.closed)
        if (int rc = to_dev(c.bflip, pm.flip.data(), (size_t)mesh->n_tris, c.stream)) return rc;
    ObjParams prm;
    std::memset(&prm, 0, sizeof(prm));
    prm.verts = reinterpret_cast<const float*>(c.b0.p);
    prm.tris = reinterpret_cast<const int32_t*>(c.b1.p);
    prm.tri_flip = pm.closed ? reinterpret_cast<const uint8_t*>(c.bflip.p) : nullptr;
    prm.n_verts = mesh->n_verts;
    prm.n_tris = mesh->n_tris;
    a.max_verts = mesh->n_verts;
    a.max_tris = mesh->n_tris;
    FrameCtrl fc;
    clear_ctrl(fc);
    fc.n_steps = 1;        // (walked already: the segment below only decides)
    fc.outlier_step = 0;
    fc.cur_slot = B_LIN0;
    fc.lane = 0;
    fc.feat_read = 0;
    if (depth && mask) {
        std::vector<uint8_t> zero;
        if (int rc = to_dev(c.b2, mask, npix, c.stream)) return rc;
        if (int rc = to_dev(c.b3, depth, npix, c.stream)) return rc;
        fc.has_new_mask = 1;
        fc.new_mask = c.b2.p;
        fc.slot_cur = kSlotNew;
        fc.depth_cur = reinterpret_cast<const float*>(c.b3.p);
        fc.feat_write = 0;
    }
    ObjState* st = new ObjState();
    init_state(*st);
    st->lane[0].pending_frame = 0;
    st->lane[0].pc_frame = 0;
    st->lane[0].pc_step = 1;
    for (int k = 0; k < 2; ++k) {
        PoseBelief& b = st->belief[b_alt(0, k)];
        for (int i = 0; i < 3; ++i) b.mean[6 + i] = x2[3 * k + i];
        for (int i = 0; i < 4; ++i) b.mean[9 + i] = q2[4 * k + i];
    }
    hipError_t err = hipMemcpyAsync(c.arr.state.p, st, sizeof(ObjState), hipMemcpyHostToDevice, c.stream);
    if (err == hipSuccess) err = hipMemcpyAsync(c.arr.params.p, &prm, sizeof(prm), hipMemcpyHostToDevice, c.stream);
    if (err == hipSuccess) err = hipStreamSynchronize(c.stream);
    if (err != hipSuccess) { delete st; restore(); HIP_TRY(err); }
    if (int rc = upload_ctrl(c, fc)) { delete st; restore(); return rc; }
    OutlierLaunchOpts o = o_in;
    if (tiles_out) {
        err = c.b4.ensure(sizeof(float) * 2 * tpix);
        if (err == hipSuccess) err = hipMemsetAsync(c.b4.p, 0, sizeof(float) * 2 * tpix, c.stream);
        if (err != hipSuccess) { delete st; restore(); HIP_TRY(err); }
        o.tile_dump = reinterpret_cast<float*>(c.b4.p);
    }
    if (depth && mask) {
        launch_mask_ingest(a, 0, c.stream);
        launch_features(a, c.stream);
    }
    launch_outlier(a, 0, c.stream, nullptr, &o);
    roft_ut_params ut{1.0, 2.0, 0.0};
    launch_ukf_chain(a, ut, false, 0, c.stream);   // decision (ROFTFilter.cpp:581-583) as the engine's next segment takes it
    err = hipMemcpyAsync(st, c.arr.state.p, sizeof(ObjState), hipMemcpyDeviceToHost, c.stream);
    if (err == hipSuccess && tiles_out) err = hipMemcpyAsync(tiles_out, c.b4.p, sizeof(float) * 2 * tpix, hipMemcpyDeviceToHost, c.stream);
    if (err == hipSuccess) err = hipStreamSynchronize(c.stream);
    if (err == hipSuccess) err = hipGetLastError();
    if (err == hipSuccess) {
        for (int k = 0; k < 2; ++k) {
            if (L_out) L_out[k] = st->lane[0].outlier_L[k];
            if (samples_out) samples_out[k] = (long)st->lane[0].outlier_cnt[k];
        }
        if (selected_out) *selected_out = st->lane[0].outlier_selected;
    }
    delete st;
    restore();
    HIP_TRY(err);
    return ROFT_OK;
}

int roft_mesh_classify(const roft_mesh* mesh, uint8_t* flip_out, int* closed_out)
{
    if (!mesh || !mesh->verts || !mesh->tris || mesh->n_verts <= 0 || mesh->n_tris <= 0 || !closed_out) return fail(ROFT_ERR_INVALID, "bad argument");
    std::vector<uint8_t> flip;
    *closed_out = classify_mesh(mesh->verts, mesh->n_verts, mesh->tris, mesh->n_tris, flip) ? 1 : 0;
    if (flip_out) std::memcpy(flip_out, flip.data(), (size_t)mesh->n_tris);
    return ROFT_OK;
}

int roft_render_depth(const roft_mesh* mesh, const double x[3], const double q[4], const roft_camera* cam, int divider,
                      float* tile)
{
    if (!mesh || !mesh->verts || !mesh->tris || mesh->n_verts <= 0 || mesh->n_tris <= 0 || !x || !q || !cam || !tile || divider <= 0)
        return fail(ROFT_ERR_INVALID, "bad argument");
    const size_t tpix = (size_t)(cam->width / divider) * (cam->height / divider);
    std::vector<float> tiles(2 * tpix);
    const double x2[6] = {x[0], x[1], x[2], x[0], x[1], x[2]};
    const double q2[8] = {q[0], q[1], q[2], q[3], q[0], q[1], q[2], q[3]};
    OutlierLaunchOpts o;
    if (int rc = op_outlier(cam, divider, nullptr, nullptr, mesh, x2, q2, o, nullptr, nullptr, nullptr, tiles.data())) return rc;
    std::memcpy(tile, tiles.data(), sizeof(float) * tpix);
    return ROFT_OK;
}

int roft_outlier_test(const roft_camera* cam, int divider, const float* depth, const uint8_t* mask, const roft_mesh* mesh,
                      const double x[6], const double q[8], int bands, int vertex_cache, int window_pixels, double L_out[2],
                      long samples_out[2], int* selected_out, float* tiles_out)
{
    if (!cam || !depth || !mask || !mesh || !mesh->verts || !mesh->tris || mesh->n_verts <= 0 || mesh->n_tris <= 0 || !x || !q ||
        divider <= 0 || bands < 0 || bands > kMaxOutlierParts || window_pixels < 0)
        return fail(ROFT_ERR_INVALID, "bad argument");
    return roft_outlier_test_split(cam, divider, depth, mask, mesh, x, q, bands, vertex_cache, window_pixels, -1, L_out, samples_out, selected_out, tiles_out);
}

int roft_outlier_test_split(const roft_camera* cam, int divider, const float* depth, const uint8_t* mask, const roft_mesh* mesh,
                            const double x[6], const double q[8], int bands, int vertex_cache, int window_pixels, int split, double L_out[2],
                            long samples_out[2], int* selected_out, float* tiles_out)
{
    if (!cam || !depth || !mask || !mesh || !mesh->verts || !mesh->tris || mesh->n_verts <= 0 || mesh->n_tris <= 0 || !x || !q ||
        divider <= 0 || bands < 0 || bands > kMaxOutlierParts || window_pixels < 0)
        return fail(ROFT_ERR_INVALID, "bad argument");
    OutlierLaunchOpts o;
    o.parts = bands;
    o.no_vertex_cache = vertex_cache ? 0 : 1;
    o.window_pixels = window_pixels;
    o.split = split < 0 ? -1 : (split ? 1 : 0);   // (this call only: nothing process-wide changes)
    return op_outlier(cam, divider, depth, mask, mesh, x, q, o, L_out, samples_out, selected_out, tiles_out);
}

int roft_depth_likelihood(const roft_camera* cam, const float* depth, const uint8_t* mask, const float* tile, int divider,
                          double* L_out, long* samples_out)
{
    if (!cam || !depth || !mask || !tile || !L_out || divider <= 0) return fail(ROFT_ERR_INVALID, "bad argument");
    OpCtx& c = op();
    std::lock_guard<std::mutex> lk(c.mu);
    if (int rc = c.prepare(*cam, ROFT_FLOW_F32C2, 1, 1.0f, 35)) return rc;
    c.arr.a.cam.divider = divider;
    c.arr.a.tile_w = cam->width / divider;
    c.arr.a.tile_h = cam->height / divider;
    const size_t npix = (size_t)cam->width * cam->height;
    const size_t tpix = (size_t)c.arr.a.tile_w * c.arr.a.tile_h;
    if (tpix * 2 > c.arr.zbuf.n) HIP_TRY(c.arr.zbuf.ensure(tpix * 2));
    c.arr.a.zbuf = c.arr.zbuf.p;
    if (int rc = to_dev(c.b0, mask, npix, c.stream)) return rc;
    if (int rc = to_dev(c.b1, depth, npix, c.stream)) return rc;
    // tile -> z-buffer bit pattern (0 = background -> +inf), used for both alternatives
    std::vector<uint32_t> zb(tpix * 2);
    for (size_t i = 0; i < tpix; ++i) {
        uint32_t bits;
        std::memcpy(&bits, &tile[i], 4);
        if (tile[i] == 0.0f) bits = 0x7F800000u;
        zb[i] = bits;
        zb[tpix + i] = bits;
    }
    HIP_TRY(hipMemcpyAsync(c.arr.zbuf.p, zb.data(), zb.size() * 4, hipMemcpyHostToDevice, c.stream));
    FrameCtrl fc;
    clear_ctrl(fc);
    fc.has_new_mask = 1;
    fc.new_mask = c.b0.p;
    fc.slot_cur = kSlotNew;
    fc.depth_cur = reinterpret_cast<const float*>(c.b1.p);
    fc.feat_write = 0;
    fc.feat_read = 0;
    fc.outlier_step = 0;
    if (int rc = upload_ctrl(c, fc)) return rc;
    {
        ObjState st0;
        init_state(st0);
        st0.lane[0].pending_frame = 0;   // the test of frame 0 is pending
        HIP_TRY(hipMemcpyAsync(c.arr.state.p, &st0, sizeof(st0), hipMemcpyHostToDevice, c.stream));
        HIP_TRY(hipStreamSynchronize(c.stream));
    }
    launch_mask_ingest(c.arr.a, 0, c.stream);
    launch_features(c.arr.a, c.stream);
    // likelihood only (the z-buffers are already filled)
    launch_outlier_only(c.arr.a, c.stream);
    ObjState* st = new ObjState();
    hipError_t err = hipMemcpyAsync(st, c.arr.state.p, sizeof(ObjState), hipMemcpyDeviceToHost, c.stream);
    if (err == hipSuccess) err = hipStreamSynchronize(c.stream);
    if (err == hipSuccess) err = hipGetLastError();
    if (err == hipSuccess) {
        *L_out = st->lane[0].outlier_L[0];
        if (samples_out) *samples_out = (long)st->lane[0].outlier_cnt[0];
    }
    delete st;
    // restore the default tile geometry of this context
    c.arr.a.cam = make_cam(*cam);
    c.arr.a.tile_w = cam->width / c.arr.a.cam.divider;
    c.arr.a.tile_h = cam->height / c.arr.a.cam.divider;
    c.ready = false;  // zbuf may have been re-sized: force a clean re-allocation next time
    HIP_TRY(err);
    return ROFT_OK;
}

}  // extern "C"
