/*
 * ro_tracker.c -- CPU oracle (test infrastructure): the per-frame orchestration of ROFT.
 *
 * Restates ROFT::ROFTFilter  src/roft-lib/src/ROFTFilter.cpp
 *   initialization_step :216-237, filtering_step :255-452 (steps numbered below as in that
 *   function), pick_best_alternative :467-621, buffer_outlier_rejection_features :624-646,
 *   correct_outlier_rejection :649-676;
 * the state machines of the sources/models it drives:
 *   ImageSegmentationOFAidedSource<T>::step_frame   include/ROFT/ImageSegmentationOFAidedSource.hpp:127-231
 *   ImageSegmentationMeasurement::freeze            src/roft-lib/src/ImageSegmentationMeasurement.cpp:30-75
 *   ImageOpticalFlowMeasurement<T>::freeze          include/ROFT/ImageOpticalFlowMeasurement.hpp:167-294
 *   CartesianQuaternionMeasurement::freeze          src/roft-lib/src/CartesianQuaternionMeasurement.cpp:92-348
 * The caller plays the role of the Dataset* sources: it hands over, per frame, the depth image,
 * the flow frame (or none), a newly delivered mask (or none) and a newly delivered pose (or none),
 * on the reference's 5 fps / 6-frame-delay schedule (DatasetImageSegmentationDelayed.cpp:42-63).
 */
#include "roft_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define VEL_BUF_MAX 64

typedef struct {
    double mean[13];
    double cov[144];
} pose_belief;

struct ro_tracker {
    ro_tracker_config cfg;
    ro_mesh mesh;
    float* mesh_verts;
    int32_t* mesh_tris;
    uint8_t* mesh_flip;
    int W, H, divider;
    size_t npix;

    /* beliefs */
    double v_mean[6], v_cov[36];
    pose_belief p_corr, p_pred, buffered_belief;

    /* ImageSegmentationOFAidedSource state */
    uint8_t* of_mask;      /* mask_ (raw values) */
    int seg_available;     /* segmentation_available_ */
    int of_first_frame;    /* is_first_frame_ */
    void** flow_buf;       /* copies of the buffered flow frames (oldest first) */
    ro_flow* flow_buf_desc;
    int flow_buf_n, flow_buf_cap;
    int32_t* map_scratch;
    /* OpticalFlowQueueHandler (window 30, hpp:90-92) of the stamped source */
    void* fq_data[30];
    ro_flow fq_desc[30];
    double fq_stamp[30];
    int fq_n;

    /* ImageSegmentationMeasurement state */
    uint8_t* seg_bin; /* segmentation_ (binarised) */
    int seg_meas_available;

    /* ImageOpticalFlowMeasurement state */
    float* prev_depth;
    uint8_t* prev_seg;
    int flow_first_frame;
    int32_t* uv;
    double* y;
    double* Hm;
    int capacity;

    /* CartesianQuaternionMeasurement state */
    double vel_buf[VEL_BUF_MAX][6];
    int vel_buf_n;
    double last_lin[3], last_ang[3];
    double last_pose_x[3], last_pose_q[4];
    int is_pose, is_first_velocity_in;
    int meas_type;
    double meas[13];

    /* outlier rejection */
    float* buffered_depth;
    uint8_t* buffered_seg;
    int features_initialized;
    float* tiles; /* 2 tiles */
    int shadow, shadow_valid; /* ro_tracker_shadow_render */
    double shadow_L[4];

    /* current frame */
    const float* cur_depth;
    ro_frame_result* res;
};

void ro_tracker_default_config(ro_tracker_config* c, int width, int height)
{
    /* defaults of config/config_fast_ycb.cfg / config_ho3d.cfg with the overrides of
     * test/test.sh:70-71 (SIGMA_ANG_VEL, P_COV_Q) */
    memset(c, 0, sizeof(*c));
    c->cam.width = width;
    c->cam.height = height;
    if (width == 640) {
        c->cam.fx = c->cam.fy = 614.7142806307731;
        c->cam.cx = 320.0; c->cam.cy = 240.0;
    } else {
        c->cam.fx = c->cam.fy = 1229.4285612615463;
        c->cam.cx = 640.0; c->cam.cy = 360.0;
    }
    c->sample_time = 1.0 / 30.0;
    c->ut.alpha = 1.0; c->ut.beta = 2.0; c->ut.kappa = 0.0;
    c->p_mean0[9] = 1.0;
    for (int i = 0; i < 12; i++) c->p_cov0_diag[i] = 1e-3;
    for (int i = 0; i < 6; i++) { c->v_cov0_diag[i] = 1e-3; c->v_q_diag[i] = 0.1; }
    for (int i = 0; i < 3; i++) {
        c->p_sigma_ang_vel[i] = 1.0;
        c->p_psd_lin_acc[i] = 1.0;
        c->p_meas_cov_v[i] = 0.1;
        c->p_meas_cov_w[i] = 1e-4;
        c->p_meas_cov_x[i] = 1e-3;
        c->p_meas_cov_q[i] = 1e-4;
    }
    c->v_meas_cov_flow[0] = c->v_meas_cov_flow[1] = 1.0;
    c->depth_maximum = 2.0;
    c->subsampling_radius = 35.0;
    c->flow_weighting = 1;
    c->use_pose = c->use_pose_resync = c->use_velocity = 1;
    c->outlier_rejection = 1;
    c->flow_aided_segmentation = 1;
    c->mask_frames_between = 6;
    c->pose_frames_between = 6;
}

ro_tracker* ro_tracker_create(const ro_tracker_config* cfg, const ro_mesh* mesh)
{
    ro_tracker* t = (ro_tracker*)calloc(1, sizeof(ro_tracker));
    t->cfg = *cfg;
    t->W = cfg->cam.width;
    t->H = cfg->cam.height;
    t->npix = (size_t)t->W * t->H;
    t->divider = (t->W == 640) ? 2 : 4; /* ROFTFilter.cpp:191-193 */
    if (mesh && mesh->n_verts > 0) {
        t->mesh_verts = (float*)malloc(sizeof(float) * 3 * mesh->n_verts);
        memcpy(t->mesh_verts, mesh->verts, sizeof(float) * 3 * mesh->n_verts);
        t->mesh_tris = (int32_t*)malloc(sizeof(int32_t) * 3 * mesh->n_tris);
        memcpy(t->mesh_tris, mesh->tris, sizeof(int32_t) * 3 * mesh->n_tris);
        t->mesh.verts = t->mesh_verts; t->mesh.n_verts = mesh->n_verts;
        t->mesh.tris = t->mesh_tris; t->mesh.n_tris = mesh->n_tris;
        /* closed surface?  (ro_meshclass.c: the render then leaves out the triangles that face away) -- once per tracker */
        t->mesh_flip = (uint8_t*)malloc((size_t)mesh->n_tris);
        t->mesh.closed = ro_mesh_classify(t->mesh_verts, mesh->n_verts, t->mesh_tris, mesh->n_tris, t->mesh_flip);
        t->mesh.tri_flip = t->mesh_flip;
    }
    t->of_mask = (uint8_t*)calloc(t->npix, 1);
    t->seg_bin = (uint8_t*)calloc(t->npix, 1);
    t->prev_seg = (uint8_t*)calloc(t->npix, 1);
    t->buffered_seg = (uint8_t*)calloc(t->npix, 1);
    t->prev_depth = (float*)calloc(t->npix, sizeof(float));
    t->buffered_depth = (float*)calloc(t->npix, sizeof(float));
    t->map_scratch = (int32_t*)malloc(sizeof(int32_t) * t->npix);
    t->capacity = (int)(t->npix / 2 + 16);
    t->uv = (int32_t*)malloc(sizeof(int32_t) * 2 * t->capacity);
    t->y = (double*)malloc(sizeof(double) * 2 * t->capacity);
    t->Hm = (double*)malloc(sizeof(double) * 12 * t->capacity);
    t->tiles = (float*)malloc(sizeof(float) * 2 * (t->W / t->divider) * (t->H / t->divider));
    t->flow_buf_cap = 32;
    t->flow_buf = (void**)calloc(t->flow_buf_cap, sizeof(void*));
    t->flow_buf_desc = (ro_flow*)calloc(t->flow_buf_cap, sizeof(ro_flow));

    /* initialization_step (ROFTFilter.cpp:216-237) */
    memcpy(t->v_mean, cfg->v_mean0, sizeof(t->v_mean));
    memcpy(t->p_corr.mean, cfg->p_mean0, sizeof(t->p_corr.mean));
    for (int i = 0; i < 6; i++) t->v_cov[i * 6 + i] = cfg->v_cov0_diag[i];
    for (int i = 0; i < 12; i++) t->p_corr.cov[i * 12 + i] = cfg->p_cov0_diag[i];
    t->buffered_belief = t->p_corr;
    t->p_pred = t->p_corr;
    t->of_first_frame = 1;
    t->flow_first_frame = 1;
    return t;
}

void ro_tracker_destroy(ro_tracker* t)
{
    if (!t) return;
    for (int i = 0; i < t->flow_buf_cap; i++) free(t->flow_buf[i]);
    for (int i = 0; i < 30; i++) free(t->fq_data[i]);
    free(t->flow_buf); free(t->flow_buf_desc);
    free(t->mesh_verts); free(t->mesh_tris); free(t->mesh_flip);
    free(t->of_mask); free(t->seg_bin); free(t->prev_seg); free(t->buffered_seg);
    free(t->prev_depth); free(t->buffered_depth); free(t->map_scratch);
    free(t->uv); free(t->y); free(t->Hm); free(t->tiles);
    free(t);
}

const uint8_t* ro_tracker_mask(const ro_tracker* t) { return t->seg_bin; }

void ro_tracker_shadow_render(ro_tracker* t, int on) { t->shadow = on; t->shadow_valid = 0; }
int ro_tracker_shadow_L(const ro_tracker* t, double out[4])
{
    if (!t->shadow_valid) return 0;
    for (int i = 0; i < 4; i++) out[i] = t->shadow_L[i];
    return 1;
}

static size_t flow_bytes(const ro_flow* f)
{
    return (size_t)f->cols * f->rows * 2 * (f->type == RO_FLOW_S16C2 ? sizeof(int16_t) : sizeof(float));
}

static void flow_buf_push(ro_tracker* t, const ro_flow* f)
{
    /* flow_buffer_.push_back(flow.clone()) (hpp:205-209).  The reference's vector is unbounded; only the last
     * mask_frames_between entries can ever be used when that number is known (hpp:239-245), so older ones are dropped
     * here.  When it is unknown (<= 0) every buffered flow is used: the buffer grows. */
    const int keep = t->cfg.mask_frames_between;
    if (keep > 0 && t->flow_buf_n == keep) {
        void* oldest = t->flow_buf[0];
        memmove(t->flow_buf, t->flow_buf + 1, sizeof(void*) * (keep - 1));
        memmove(t->flow_buf_desc, t->flow_buf_desc + 1, sizeof(ro_flow) * (keep - 1));
        t->flow_buf[keep - 1] = oldest;
        t->flow_buf_n--;
    }
    if (t->flow_buf_n == t->flow_buf_cap) {
        const int cap = t->flow_buf_cap * 2;
        t->flow_buf = (void**)realloc(t->flow_buf, sizeof(void*) * cap);
        t->flow_buf_desc = (ro_flow*)realloc(t->flow_buf_desc, sizeof(ro_flow) * cap);
        for (int i = t->flow_buf_cap; i < cap; i++) t->flow_buf[i] = NULL;
        t->flow_buf_cap = cap;
    }
    int k = t->flow_buf_n++;
    t->flow_buf[k] = realloc(t->flow_buf[k], flow_bytes(f));
    memcpy(t->flow_buf[k], f->data, flow_bytes(f));
    t->flow_buf_desc[k] = *f;
    t->flow_buf_desc[k].data = t->flow_buf[k];
}

static int mask_is_empty(const uint8_t* m, size_t n)
{
    for (size_t i = 0; i < n; i++)
        if (m[i]) return 0;
    return 1;
}

/* ImageSegmentationOFAidedSource<T>::step_frame (hpp:127-231), wait_source_initialization=false */
static void of_aided_step(ro_tracker* t, const ro_frame* f)
{
    int valid_segmentation = (f->mask != NULL);
    const uint8_t* mask = f->mask;

    if (!t->seg_available && valid_segmentation) {
        t->seg_available = 1;
        memcpy(t->of_mask, mask, t->npix);
        valid_segmentation = 0;
    }
    if (valid_segmentation) {
        if (mask_is_empty(mask, t->npix)) {
            valid_segmentation = 0;
            if (t->cfg.mask_frames_between <= 0) t->flow_buf_n = 0;
        }
    }
    int valid_flow = f->flow.valid && !t->of_first_frame;
    if (valid_flow) flow_buf_push(t, &f->flow);

    if (valid_segmentation) {
        memcpy(t->of_mask, mask, t->npix);
        ro_mask_propagate(t->of_mask, t->W, t->H, t->flow_buf_desc, t->flow_buf_n,
                          t->cfg.mask_frames_between, t->map_scratch);
        t->flow_buf_n = 0;
    } else if (valid_flow && t->seg_available) {
        t->of_mask[0] = 0;
        ro_flow one = f->flow;
        ro_mask_propagate(t->of_mask, t->W, t->H, &one, 1, t->cfg.mask_frames_between, t->map_scratch);
    }
    t->of_first_frame = 0;
}

/* OpticalFlowQueueHandler::add_flow (OpticalFlowQueueHandler.cpp:18-26) */
static void flow_queue_add(ro_tracker* t, const ro_flow* f, double stamp)
{
    if (t->fq_n == 30) {
        void* oldest = t->fq_data[0];
        memmove(t->fq_data, t->fq_data + 1, sizeof(void*) * 29);
        memmove(t->fq_desc, t->fq_desc + 1, sizeof(ro_flow) * 29);
        memmove(t->fq_stamp, t->fq_stamp + 1, sizeof(double) * 29);
        t->fq_data[29] = oldest;
        t->fq_n = 29;
    }
    int k = t->fq_n++;
    t->fq_data[k] = realloc(t->fq_data[k], flow_bytes(f));
    memcpy(t->fq_data[k], f->data, flow_bytes(f));
    t->fq_desc[k] = *f;
    t->fq_desc[k].data = t->fq_data[k];
    t->fq_stamp[k] = stamp;
}

/* ImageSegmentationOFAidedSourceStamped<T>::step_frame (…Stamped.hpp:153-268), wait_source_initialization=false,
 * no source feedback.  get_buffer_region (OpticalFlowQueueHandler.cpp:29-58): the flows stored AFTER the entry whose
 * stamp matches the mask stamp within 1 ms (`abs` taken as the floating-point absolute value). */
static void of_aided_stamped_step(ro_tracker* t, const ro_frame* f)
{
    int valid_segmentation = (f->mask != NULL);
    const uint8_t* mask = f->mask;
    if (!t->seg_available && valid_segmentation) {
        t->seg_available = 1;
        memcpy(t->of_mask, mask, t->npix);
        valid_segmentation = 0;
    }
    if (valid_segmentation && mask_is_empty(mask, t->npix)) valid_segmentation = 0;
    int valid_flow = f->flow.valid && !t->of_first_frame;
    if (valid_flow) flow_queue_add(t, &f->flow, f->stamp);

    if (valid_segmentation) {
        memcpy(t->of_mask, mask, t->npix);
        int first = -1;
        for (int i = 0; i < t->fq_n; i++)
            if (fabs(t->fq_stamp[i] - f->mask_stamp) < 1e-3) { first = i + 1; break; }
        int n = (first >= 0) ? t->fq_n - first : 0;
        if (n > 0) {
            ro_mask_propagate(t->of_mask, t->W, t->H, t->fq_desc + first, n, t->cfg.mask_frames_between, t->map_scratch);
        } else if (f->flow.valid) {
            t->of_mask[0] = 0;
            ro_flow one = f->flow;
            ro_mask_propagate(t->of_mask, t->W, t->H, &one, 1, t->cfg.mask_frames_between, t->map_scratch);
        }
    } else if (valid_flow && t->seg_available) {
        t->of_mask[0] = 0;
        ro_flow one = f->flow;
        ro_mask_propagate(t->of_mask, t->W, t->H, &one, 1, t->cfg.mask_frames_between, t->map_scratch);
    }
    t->of_first_frame = 0;
}

/* ImageSegmentationMeasurement::freeze */
static int segmentation_freeze(ro_tracker* t, const ro_frame* f)
{
    if (t->cfg.flow_aided_segmentation) {
        if (t->cfg.stamped_masks) of_aided_stamped_step(t, f);
        else of_aided_step(t, f);
        if (t->seg_available) {
            t->seg_meas_available = 1;
            ro_mask_binarise(t->of_mask, t->seg_bin, t->npix);
        }
    } else if (f->mask) {
        t->seg_meas_available = 1;
        ro_mask_binarise(f->mask, t->seg_bin, t->npix);
    }
    return t->seg_meas_available;
}

/* ImageOpticalFlowMeasurement<T>::freeze(ExceptStepSource); returns validity, *N on success */
static int flow_freeze(ro_tracker* t, const ro_frame* f, int* N)
{
    *N = -1;
    if (!t->seg_meas_available) return 0;
    int flow_available = f->flow.valid;
    if (!flow_available || t->flow_first_frame) {
        memcpy(t->prev_depth, f->depth, sizeof(float) * t->npix);
        memcpy(t->prev_seg, t->seg_bin, t->npix);
        t->flow_first_frame = 0;
        return 0;
    }
    *N = ro_flow_measurement(&t->cfg.cam, t->prev_seg, t->prev_depth, &f->flow, f->dt,
                             (float)(size_t)t->cfg.subsampling_radius, t->cfg.depth_maximum,
                             t->capacity, t->uv, t->y, t->Hm);
    memcpy(t->prev_depth, f->depth, sizeof(float) * t->npix);
    memcpy(t->prev_seg, t->seg_bin, t->npix);
    return flow_available;
}

static void buffer_features(ro_tracker* t)
{
    memcpy(t->buffered_depth, t->cur_depth, sizeof(float) * t->npix);
    memcpy(t->buffered_seg, t->seg_bin, t->npix);
}

static void set_meas_velocity(ro_tracker* t)
{
    t->meas_type = RO_MEAS_VELOCITY;
    memcpy(t->meas, t->last_lin, sizeof(double) * 3);
    memcpy(t->meas + 3, t->last_ang, sizeof(double) * 3);
}

static void set_meas_pose_velocity(ro_tracker* t)
{
    t->meas_type = RO_MEAS_POSE_VELOCITY;
    memcpy(t->meas, t->last_lin, sizeof(double) * 3);
    memcpy(t->meas + 3, t->last_ang, sizeof(double) * 3);
    memcpy(t->meas + 6, t->last_pose_x, sizeof(double) * 3);
    memcpy(t->meas + 9, t->last_pose_q, sizeof(double) * 4);
}

static void vel_buf_pop_front(ro_tracker* t)
{
    memmove(t->vel_buf[0], t->vel_buf[1], sizeof(double) * 6 * (t->vel_buf_n - 1));
    t->vel_buf_n--;
}

static void vel_buf_push(ro_tracker* t, const double* v6)
{
    if (t->vel_buf_n == VEL_BUF_MAX) vel_buf_pop_front(t);
    memcpy(t->vel_buf[t->vel_buf_n++], v6, sizeof(double) * 6);
}

/* CartesianQuaternionMeasurement::freeze(Standard) (cpp:176-347) */
static int meas_freeze_standard(ro_tracker* t, const ro_frame* f)
{
    if (t->cfg.use_velocity) {
        t->is_first_velocity_in = 1;
        memcpy(t->last_lin, t->v_mean, sizeof(double) * 3);
        memcpy(t->last_ang, t->v_mean + 3, sizeof(double) * 3);
    }
    t->is_pose = 0;
    if (t->cfg.use_pose) {
        t->is_pose = f->pose_valid;
        if (t->is_pose) {
            memcpy(t->last_pose_x, f->pose_x, sizeof(double) * 3);
            memcpy(t->last_pose_q, f->pose_q, sizeof(double) * 4);
        }
    }
    if (t->is_first_velocity_in && t->is_pose) {
        set_meas_pose_velocity(t);
        vel_buf_push(t, t->meas);
    } else if (t->is_first_velocity_in) {
        set_meas_velocity(t);
        vel_buf_push(t, t->meas);
    } else if (t->is_pose) {
        t->meas_type = RO_MEAS_POSE;
        memcpy(t->meas, t->last_pose_x, sizeof(double) * 3);
        memcpy(t->meas + 3, t->last_pose_q, sizeof(double) * 4);
    } else {
        t->meas_type = RO_MEAS_NONE;
        return 0;
    }
    return 1;
}

/* CartesianQuaternionMeasurement::freeze(PopBufferedMeasurement) (cpp:97-154) */
static int meas_freeze_pop(ro_tracker* t)
{
    if (t->cfg.pose_frames_between > 0)
        while (t->vel_buf_n > t->cfg.pose_frames_between + 1) vel_buf_pop_front(t);
    if (t->vel_buf_n == 0) {
        vel_buf_push(t, t->meas); /* measurement_.col(0).head<6>() */
        return 0;
    }
    memcpy(t->last_lin, t->vel_buf[0], sizeof(double) * 3);
    memcpy(t->last_ang, t->vel_buf[0] + 3, sizeof(double) * 3);
    vel_buf_pop_front(t);
    if (t->is_pose) {
        set_meas_pose_velocity(t);
        t->is_pose = 0;
    } else
        set_meas_velocity(t);
    return 1;
}

/* CartesianQuaternionMeasurement::freeze(RepeatOnlyVelocity) (cpp:156-174) */
static void meas_freeze_repeat_velocity(ro_tracker* t)
{
    if (t->is_first_velocity_in) set_meas_velocity(t);
}

static void p_predict(ro_tracker* t, double dt, const pose_belief* in, pose_belief* out)
{
    double Q[81];
    ro_pose_process_noise(t->cfg.p_psd_lin_acc, t->cfg.p_sigma_ang_vel, dt, Q);
    pose_belief tmp;
    ro_ukf_predict(in->mean, in->cov, Q, dt, &t->cfg.ut, tmp.mean, tmp.cov);
    *out = tmp;
}

static void p_correct(ro_tracker* t, const pose_belief* pred, pose_belief* corr)
{
    double Rdiag[12];
    int k = 0;
    if (t->meas_type == RO_MEAS_VELOCITY || t->meas_type == RO_MEAS_POSE_VELOCITY) {
        for (int i = 0; i < 3; i++) Rdiag[k++] = t->cfg.p_meas_cov_v[i];
        for (int i = 0; i < 3; i++) Rdiag[k++] = t->cfg.p_meas_cov_w[i];
    }
    if (t->meas_type == RO_MEAS_POSE || t->meas_type == RO_MEAS_POSE_VELOCITY) {
        for (int i = 0; i < 3; i++) Rdiag[k++] = t->cfg.p_meas_cov_x[i];
        for (int i = 0; i < 3; i++) Rdiag[k++] = t->cfg.p_meas_cov_q[i];
    }
    pose_belief tmp;
    ro_ukf_correct(pred->mean, pred->cov, t->meas_type, t->meas, Rdiag, &t->cfg.ut, tmp.mean, tmp.cov);
    *corr = tmp;
    if (t->res) t->res->n_ukf_corrections++;
}

/* correct_outlier_rejection + pick_best_alternative */
static void correct_outlier_rejection(ro_tracker* t, const pose_belief* pred, int use_buffered,
                                      pose_belief* out)
{
    pose_belief alt[2];
    p_correct(t, pred, &alt[0]);
    meas_freeze_repeat_velocity(t);
    p_correct(t, pred, &alt[1]);

    const float* depth = use_buffered ? t->buffered_depth : t->cur_depth;
    const uint8_t* seg = use_buffered ? t->buffered_seg : t->seg_bin;
    const size_t tile_px = (size_t)(t->W / t->divider) * (t->H / t->divider);
    double L[2];
    for (int a = 0; a < 2; a++) {
        ro_render_depth(&t->mesh, alt[a].mean + 6, alt[a].mean + 9, &t->cfg.cam, t->divider,
                        t->tiles + a * tile_px);
        L[a] = ro_depth_likelihood(&t->cfg.cam, depth, seg, t->tiles + a * tile_px, t->divider, NULL);
    }
    t->shadow_valid = 0;
    if (t->shadow) {
        float* tmp = (float*)malloc(sizeof(float) * tile_px);
        for (int m = 0; m < 2; m++)
            for (int a = 0; a < 2; a++) {
                ro_render_depth_mode(&t->mesh, alt[a].mean + 6, alt[a].mean + 9, &t->cfg.cam, t->divider, tmp,
                                     m == 0 ? RO_RENDER_V1 : RO_RENDER_GL);
                t->shadow_L[2 * m + a] = ro_depth_likelihood(&t->cfg.cam, depth, seg, tmp, t->divider, NULL);
            }
        free(tmp);
        t->shadow_valid = 1;
    }
    int selected = (L[0] > 2.0 * L[1]) ? 1 : 0;
    if (t->res) {
        t->res->outlier_selected = selected;
        t->res->outlier_L[0] = L[0];
        t->res->outlier_L[1] = L[1];
    }
    *out = alt[selected];
}

int ro_tracker_step(ro_tracker* t, const ro_frame* f, ro_frame_result* out)
{
    ro_frame_result local;
    if (!out) out = &local;
    memset(out, 0, sizeof(*out));
    out->n_flow_points = -1;
    out->outlier_selected = -1;
    t->res = out;
    t->cur_depth = f->depth;

    int data_in = (f->depth != NULL);
    if (!data_in) return -1; /* "cannot continue without a continuous depth stream" (:261-266) */

    /* 4: segmentation freeze; 5: flow freeze */
    data_in &= segmentation_freeze(t, f);
    int N = -1;
    data_in &= flow_freeze(t, f, &N);
    out->n_flow_points = N;

    /* 6: velocity filter */
    if (data_in) {
        double vm[6], vc[36], pm[6], pc[36];
        memcpy(vm, t->v_mean, sizeof(vm));
        memcpy(vc, t->v_cov, sizeof(vc));
        ro_kf_predict(t->v_mean, t->v_cov, t->cfg.v_q_diag, pm, pc);
        ro_skf_correct(pm, pc, N, t->y, t->Hm, t->cfg.v_meas_cov_flow, t->cfg.flow_weighting,
                       t->v_mean, t->v_cov);
        if (N < 3) { /* check_observability (hpp:361-366) -> restore (:297-301) */
            memcpy(t->v_mean, vm, sizeof(vm));
            memcpy(t->v_cov, vc, sizeof(vc));
        }
    }

    /* 8: buffer features on the first frame */
    if (t->cfg.use_pose_resync && !t->features_initialized) {
        if (!t->seg_meas_available) return -2;
        buffer_features(t);
        t->features_initialized = 1;
    }

    /* 9: pose prediction */
    p_predict(t, f->dt, &t->p_corr, &t->p_pred);

    /* 10-12 */
    if (meas_freeze_standard(t, f)) {
        if (t->meas_type == RO_MEAS_POSE_VELOCITY) {
            if (t->cfg.use_pose_resync) {
                pose_belief copy = t->buffered_belief;
                t->buffered_belief = t->p_corr;
                t->p_corr = copy;
                while (meas_freeze_pop(t)) {
                    p_predict(t, f->dt, &t->p_corr, &t->p_pred);
                    if (t->cfg.outlier_rejection && t->meas_type == RO_MEAS_POSE_VELOCITY)
                        correct_outlier_rejection(t, &t->p_pred, 1, &t->p_corr);
                    else
                        p_correct(t, &t->p_pred, &t->p_corr);
                }
                buffer_features(t);
            } else {
                if (t->cfg.outlier_rejection)
                    correct_outlier_rejection(t, &t->p_pred, 0, &t->p_corr);
                else
                    p_correct(t, &t->p_pred, &t->p_corr);
            }
        } else
            p_correct(t, &t->p_pred, &t->p_corr);
    } else
        t->p_corr = t->p_pred;

    memcpy(out->pose, t->p_corr.mean, sizeof(out->pose));
    memcpy(out->pose_cov, t->p_corr.cov, sizeof(out->pose_cov));
    memcpy(out->twist, t->v_mean, sizeof(out->twist));
    memcpy(out->twist_cov, t->v_cov, sizeof(out->twist_cov));
    t->res = NULL;
    return 0;
}
