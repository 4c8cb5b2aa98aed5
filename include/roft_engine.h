/*
 * roft_engine.h -- C ABI of the MI355X-native ROFT filtering engine (libroft_hip.so).
 *
 * This is the drop-in boundary for the per-frame filtering hot path of hsp-iit/roft
 * (`src/roft-lib`).  Plain C types only: pointers, sizes, POD structs.  Every entry point returns
 * an int status (ROFT_OK == 0, negative = error; nothing throws across the ABI) and
 * roft_last_error_string() describes the last failure of the calling thread.
 * Citations are relative to the reference checkout (hsp-iit/roft v1.2.1).
 *
 * Two levels:
 *  (1) operator-level entry points -- one per reference operator, host buffers in / host buffers
 *      out, used by the C++ facade classes in include/ROFT/ and by the parity tests:
 *        roft_flow_measurement   <- ImageOpticalFlowMeasurement<T>::freeze
 *                                   include/ROFT/ImageOpticalFlowMeasurement.hpp:231-283
 *        roft_kf_predict         <- bfl::KFPrediction over SpatialVelocityModel
 *                                   src/roft-lib/src/SpatialVelocityModel.cpp:15-27
 *        roft_skf_correct        <- SKFCorrection::correctStep  src/roft-lib/src/SKFCorrection.cpp:37-153
 *        roft_mask_propagate     <- ImageSegmentationOFAidedSource<T>::map + cv::remap
 *                                   include/ROFT/ImageSegmentationOFAidedSource.hpp:215,225,234-281
 *        roft_ukf_predict        <- bfl::UKFPrediction over CartesianQuaternionModel::motion
 *                                   src/roft-lib/src/CartesianQuaternionModel.cpp:86-141
 *        roft_ukf_correct        <- ROFT::UKFCorrection::correctStep over CartesianQuaternionMeasurement
 *                                   src/roft-lib/src/UKFCorrection.cpp:54-133,
 *                                   src/roft-lib/src/CartesianQuaternionMeasurement.cpp:357-487
 *        roft_render_depth       <- SICAD::superimpose(poses, ..., depth)  src/roft-lib/src/SICAD.cpp:924-1066
 *        roft_depth_likelihood   <- ROFTFilter::pick_best_alternative inner loop
 *                                   src/roft-lib/src/ROFTFilter.cpp:553-577
 *        roft_outlier_test       <- ROFTFilter::pick_best_alternative  src/roft-lib/src/ROFTFilter.cpp:467-621
 *  (2) the batched engine -- ROFTFilter::filtering_step (src/roft-lib/src/ROFTFilter.cpp:255-452)
 *      for many objects at once with all filter state resident in HBM:
 *        roft_engine_create / roft_object_add / roft_frame_submit | roft_frames_submit / roft_step / roft_get_state.
 *      roft_frames_submit hands over a BATCH of consecutive frames (a recorded sequence, or a live source read a few
 *      frames at a time): the engine then runs each of its three per-object chains (masks, velocity, pose) over the
 *      whole batch in one persistent kernel instead of a handful of launches per frame.
 */
#ifndef ROFT_ENGINE_H
#define ROFT_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ROFT_OK 0
#define ROFT_ERR_INVALID (-1)  /* bad argument */
#define ROFT_ERR_DEVICE (-2)   /* HIP runtime error / no device */
#define ROFT_ERR_CAPACITY (-3) /* caller buffer too small */
#define ROFT_ERR_STATE (-4)    /* call order violated */

/* OpenCV matrix type codes of the reference's .float flow files
 * (src/roft-lib/src/OpticalFlowUtilities.cpp:38-62) */
#define ROFT_FLOW_S16C2 11 /* CV_16SC2: S10.5 fixed point, scale 32, grid 4 (NVOF 1.0) */
#define ROFT_FLOW_F32C2 13 /* CV_32FC2: float pixels, grid 1 (NVOF 2.0) */

/* measurement types of CartesianQuaternionMeasurement (CartesianQuaternionMeasurement.h:54) */
#define ROFT_MEAS_NONE 0
#define ROFT_MEAS_VELOCITY 1
#define ROFT_MEAS_POSE 2
#define ROFT_MEAS_POSE_VELOCITY 3

/* memory kind of image pointers handed to the engine.  ROFT_MEM_DEVICE: memory the GPU can address as it is -- device or
 * managed memory, or pinned host memory that is mapped into the device's address space at the same address (hipHostMalloc,
 * hipHostRegister: zero-copy reads over the bus).  The pointers are looked up (hipPointerGetAttributes) on the first submit of
 * an engine: unregistered, pageable host memory declared as device memory is refused with ROFT_ERR_INVALID instead of
 * faulting on the GPU; later submits trust the caller (the look-up costs microseconds per pointer). */
#define ROFT_MEM_HOST 0
#define ROFT_MEM_DEVICE 1

/* frames per roft_frames_submit call (roft_config::max_batch_frames) */
#define ROFT_MAX_BATCH_FRAMES 8
/* optical-flow frames one mask can be chased through: the 30-entry queue of the time-stamped source
 * (OpticalFlowQueueHandler.cpp:18-26, at most 29 follow the matching entry) and the bound on "all buffered flows" of
 * ImageSegmentationOFAidedSource with an unknown number of frames between masks (hpp:186-198, 239-245) */
#define ROFT_MAX_FLOW_CHASE 30

typedef struct {
    int width, height;
    double fx, fy, cx, cy;
} roft_camera;

typedef struct {
    const void* data; /* rows*cols interleaved (dx,dy), row-major */
    int type;         /* ROFT_FLOW_* */
    int cols, rows;
    int grid;    /* image width / cols (DatasetImageOpticalFlow.cpp:46) */
    float scale; /* 32 for S16C2 else 1 (DatasetImageOpticalFlow.cpp:48-50) */
    int valid;
} roft_flow;

typedef struct {
    double alpha, beta, kappa;
} roft_ut_params;

typedef struct {
    const float* verts; /* n_verts x 3 (object frame, metres) */
    int n_verts;
    const int32_t* tris; /* n_tris x 3 */
    int n_tris;
} roft_mesh;

const char* roft_last_error_string(void);
/* number of visible HIP devices (0 when there is none); never fails */
int roft_device_count(void);

/* ---- (1) operator level: host buffers, device 0 ----------------------------------------- */

/* uv: 2 ints per kept point (u, v); y: 2N; H: 2N x 6 row-major; *n_out = N.
 * ROFT_ERR_CAPACITY if more than `capacity` points are kept. */
int roft_flow_measurement(const roft_camera* cam, const uint8_t* prev_mask, const float* prev_depth,
                          const roft_flow* flow, double dt, float radius, double depth_max,
                          int capacity, int32_t* uv, double* y, double* H, int* n_out);

int roft_kf_predict(const double x[6], const double P[36], const double Qdiag[6], double x_out[6],
                    double P_out[36]);

/* *status_out: 0 corrected, 1 measurement empty (corr = pred, SKFCorrection.cpp:61-69) */
int roft_skf_correct(const double x_pred[6], const double P_pred[36], int N, const double* y,
                     const double* H, const double Rdiag[2], int reweight, double x_out[6],
                     double P_out[36], int* status_out);

/* The same correction fed with the kept flow points themselves -- the form the engine's velocity filter consumes (H rows
 * rebuilt on the device from pixel, depth and camera): uv 2N ints (u, v), z N floats, flow_xy 2N floats (pixels, i.e.
 * already divided by the flow scale). */
int roft_skf_correct_points(const roft_camera* cam, double dt, const double x_pred[6], const double P_pred[36], int N,
                            const int32_t* uv, const float* z, const float* flow_xy, const double Rdiag[2],
                            int reweight, double x_out[6], double P_out[36], int* status_out);

/* mask (W*H u8) is propagated in place through flows[0..n_flows) (chronological); only the last
 * `frames_between` flows are used when frames_between > 0. */
int roft_mask_propagate(uint8_t* mask, int W, int H, const roft_flow* flows, int n_flows,
                        int frames_between);

int roft_pose_process_noise(const double psd_lin_acc[3], const double sigma_ang_vel[3], double T,
                            double Q[81]);
int roft_ukf_predict(const double mean[13], const double P[144], const double Q[81], double T,
                     const roft_ut_params* ut, double mean_out[13], double P_out[144]);
/* meas: [v w] | [x q] | [v w x q] with q = (w,x,y,z); Rdiag in the same order.
 * *status_out: 0 corrected, 1 no measurement, 2 singular innovation covariance (corr = pred) */
int roft_ukf_correct(const double mean[13], const double P[144], int type, const double* meas,
                     const double* Rdiag, const roft_ut_params* ut, double mean_out[13],
                     double P_out[144], int* status_out);

/* Is the mesh a closed orientable surface (host code, no device needed)?  *closed_out = 1 and flip_out[n_tris] (optional) = 1 for
 * every triangle wound clockwise seen from outside, else 0 and flip_out zeroed.  The reference draws every triangle (depth test
 * LESS, no culling: src/roft-lib/src/SICAD.cpp:271-272) and reads the NEAREST surface back; seen from outside, the nearest
 * surface of a closed mesh faces the camera, so the renders below leave the triangles of a closed mesh that face away out (while
 * every vertex is in front of the near plane) -- half the scan conversion.  Open, non-manifold or non-orientable meshes are drawn
 * whole.  Rules: oracle/ro_meshclass.c. */
int roft_mesh_classify(const roft_mesh* mesh, uint8_t* flip_out, int* closed_out);
/* tile: (H/divider) x (W/divider) float, 0 = background.  Drawn by the rasteriser of the engine's own outlier test
 * (outlier_fused_kernel: projected vertices and the depth window in LDS). */
int roft_render_depth(const roft_mesh* mesh, const double x[3], const double q[4],
                      const roft_camera* cam, int divider, float* tile);
/* *L_out = mean |depth - render| over every second mask pixel, DBL_MAX if no sample -- for a tile rendered elsewhere
 * (e.g. by the reference's SICAD); the engine itself never materialises a tile, see roft_outlier_test. */
int roft_depth_likelihood(const roft_camera* cam, const float* depth, const uint8_t* mask,
                          const float* tile, int divider, double* L_out, long* samples_out);
/* ROFTFilter::pick_best_alternative (src/roft-lib/src/ROFTFilter.cpp:467-621) for one object, on exactly the three launches
 * the engine enqueues at a pose arrival: the outlier-rejection features of (depth, mask) are buffered (:624-646), the two
 * alternatives x[0..2] q[0..3] (pose + velocity correction) and x[3..5] q[4..7] (velocity only) are rendered at
 * (W/divider) x (H/divider) and scored against every second mask pixel with 0 < depth < 2 (:553-577), and the decision
 * L_0 > 2 L_1 -> 1 (:581-583) is taken.  bands: workgroups one alternative is split over, 1..8 (0: by the CUs to spare, as
 * the engine does); vertex_cache 0: project the vertices per triangle instead of once into LDS; window_pixels > 0 caps
 * the LDS depth window (a larger window is rendered in strips).  None of the three changes a bit of the depths.
 * tiles_out (optional): 2 x (H/divider) x (W/divider) floats, the renders as the kernel drew them, 0 = background.
 * L_out: DBL_MAX when an alternative has no sample. */
int roft_outlier_test(const roft_camera* cam, int divider, const float* depth, const uint8_t* mask, const roft_mesh* mesh,
                      const double x[6], const double q[8], int bands, int vertex_cache, int window_pixels, double L_out[2],
                      long samples_out[2], int* selected_out, float* tiles_out);
/* The same with the way the workgroups of an alternative share its work chosen FOR THIS CALL: split 1 = its triangles (windows
 * merged through memory), 0 = only the rows of its window, -1 = the library's choice.  No result depends on it. */
int roft_outlier_test_split(const roft_camera* cam, int divider, const float* depth, const uint8_t* mask, const roft_mesh* mesh,
                            const double x[6], const double q[8], int bands, int vertex_cache, int window_pixels, int split,
                            double L_out[2], long samples_out[2], int* selected_out, float* tiles_out);

/* ---- (2) batched engine --------------------------------------------------------------------- */

typedef struct roft_engine roft_engine;

/* Mirrors the keys of config/config_fast_ycb.cfg consumed by ROFTFilter's constructor
 * (src/roft-lib/src/ROFTFilter.cpp:32-201, wiring src/roft/src/main.cpp:286-325). */
typedef struct {
    roft_camera cam;
    int flow_type;  /* ROFT_FLOW_* of every flow frame of this engine */
    int flow_grid;
    float flow_scale;
    double sample_time;
    roft_ut_params ut;
    double depth_maximum;       /* measurement_model.velocity.depth_maximum */
    double subsampling_radius;  /* measurement_model.velocity.subsampling_radius */
    int flow_weighting;         /* measurement_model.velocity.weight_flow */
    int use_pose, use_pose_resync, use_velocity; /* measurement_model.use_* */
    int outlier_rejection;                       /* outlier_rejection.enable */
    int flow_aided_segmentation;                 /* segmentation_dataset.flow_aided */
    /* original_fps / desired_fps of the mask source = segm_frames_between_iterations_: a new mask is chased through the
     * last mask_frames_between buffered flows; <= 0 = unknown: through ALL flows buffered since the last mask
     * (ImageSegmentationOFAidedSource.hpp:186-198, 239-245; at most ROFT_MAX_FLOW_CHASE, else roft_frames_submit fails
     * with ROFT_ERR_CAPACITY).  Values above ROFT_MAX_FLOW_CHASE are refused. */
    int mask_frames_between;
    int pose_frames_between;                     /* original_fps / desired_fps of the pose source */
    /* 1: masks come from a live source and carry the time stamp of the image they were computed on; a new mask is
     * propagated through the flows stored after the flow with that stamp (time-stamp keyed queue of the last 30 flows,
     * i.e. up to 29 flows; only the last mask_frames_between of them when that is > 0),
     * ImageSegmentationOFAidedSourceStamped.hpp:153-268 + OpticalFlowQueueHandler.cpp.  Needs
     * roft_frame_input::stamp / mask_stamp.  0: the frame-counting ImageSegmentationOFAidedSource. */
    int stamped_masks;
    int max_objects;
    /* Square root the sigma points are drawn from.  The reference (bfl) uses U sqrt(S) of the eigen-decomposition
     * of the covariance.  Any square root reproduces the first two moments; the choice only shows in fourth-order
     * terms: of the quaternion kinematics, ~ (var(theta) + T^2 var(omega))^2, in the prediction, and of the
     * quaternion measurement and the bilinear term w x r of the velocity measurement, ~ var(omega) var(x), in the
     * correction.  While these are small the engine draws the sigma points from the (much cheaper) Cholesky factor
     * and falls back to the eigen-decomposition otherwise:
     *   prediction:  max var(theta) + T^2 max var(omega) <= ukf_cholesky_guard
     *   correction:  max var(theta) <= ukf_cholesky_guard  and  max var(omega) max var(x) <= ukf_cholesky_guard_bilinear
     * Measured effect on the trajectories at the defaults (4e-4, 8e-3): <= 1e-10 (m, m/s, rad/s), DESIGN.md.
     * ukf_cholesky_guard = 0: always the eigen-decomposition; ukf_cholesky_guard_bilinear = 0: always for the
     * correction. */
    double ukf_cholesky_guard;
    double ukf_cholesky_guard_bilinear;
    int device;                                  /* HIP device ordinal */
    /* largest number of frames one roft_frames_submit may carry (1 .. ROFT_MAX_BATCH_FRAMES; 0 means 1).  Sizes the
     * engine's rings and, for zero-copy DEVICE inputs, the retention window (roft_engine_retain_frames). */
    int max_batch_frames;
    /* Bands (workgroups) the mask bit plane of an object is split over in the kernel that propagates the masks of one frame
     * (1..8; 0 = chosen by the engine: bands of ~20 image rows, a third of that on frames that deliver a mask).  A band is a
     * small workgroup -- four waves, a few KB of LDS -- that lives for one frame: nothing is persistent and no workgroup ever
     * waits for another one, so any number of engines and processes may share a device.  The propagation is an order-free OR:
     * the number of bands changes no bit of the masks (tests/test_batch_gpu.py).  (Rounds 2 - 3 walked a batch in one
     * persistent kernel whose workgroups met at a barrier in device memory and had to be resident together; the value then
     * had to be 1 for engines sharing a device.) */
    int mask_workgroups_per_object;
    /* Workgroups one alternative of an outlier test is rendered by: horizontal bands of the object's window, 1 .. 8; 0 = chosen
     * by the engine: from the device (CUs / (2 x max_objects), 2 for 64 objects on 256 CUs), and half of that while the host runs
     * ahead of the device by the whole in-flight bound, i.e. in the steady state of a long sequence (fewer bands occupy fewer CUs
     * for longer: with 64 objects one band tracks 5 % more object-frames/s in long runs and 2.5 % fewer in a 20-frame burst,
     * DESIGN.md section 4).  The likelihood's sums are exact (integer), so the band count changes no result. */
    int outlier_bands_per_alternative;
} roft_config;

typedef struct {
    double p_mean0[13];     /* v w x q(wxyz): initial_condition.pose */
    double p_cov0_diag[12];
    double v_mean0[6];      /* initial_condition.velocity */
    double v_cov0_diag[6];
    double p_sigma_ang_vel[3]; /* kinematic_model.pose.sigma_angular */
    double p_psd_lin_acc[3];   /* kinematic_model.pose.sigma_linear  */
    double v_q_diag[6];        /* kinematic_model.velocity.{sigma_linear, sigma_angular} */
    double p_meas_cov_v[3], p_meas_cov_w[3], p_meas_cov_x[3], p_meas_cov_q[3];
    double v_meas_cov_flow[2];
    roft_mesh mesh;            /* host pointers; copied */
} roft_object_desc;

/* One object's inputs for one frame.  Image pointers are HOST or DEVICE memory according to
 * mem_kind.  DEVICE buffers are used in place (zero copy): previous depth, the buffered flows a new mask is
 * chased through and the outlier-rejection features of ROFTFilter.cpp:624-646 are references into them, and
 * the engine keeps several frames in flight.  A DEVICE buffer handed over for frame k must therefore stay
 * valid and unmodified until the submit call that carries frame k + roft_engine_retain_frames() has returned (that
 * call blocks until every frame that can still read it has finished on the GPU).  ROFT_RETAIN_FRAMES is that number
 * for the default configuration (max_batch_frames 1, mask_frames_between 6).  A flow that outlives the window
 * because later flows were dropped (the reference clones every buffered flow) is copied into engine memory before
 * the window closes.  HOST buffers are copied into the engine's own ring before the submit call returns (the call
 * waits for the copies: the caller may re-use a HOST buffer as soon as it has returned); identical HOST pointers
 * within one frame -- a scene shared by several objects -- are uploaded once.  The ring holds
 * roft_engine_retain_frames() frames and grows, in 32 MiB pieces of device memory, to the bytes the largest frame
 * staged so far needed: retain x (depth + flow + mask) x objects when every object has images of its own (64 objects at
 * 640x480 CV_32FC2, batches of 8: 48 x 239 MB = 11.5 GB), retain x (one depth + flow + the masks) for a shared scene. */
#define ROFT_RETAIN_FRAMES 16
typedef struct {
    double dt;            /* RGB stamp delta; <= 0 means cfg.sample_time */
    const float* depth;   /* H x W metres, 0 = invalid; required */
    const void* flow;     /* flow frame or NULL when absent (first frame) */
    const uint8_t* mask;  /* newly delivered mask or NULL */
    int pose_valid;       /* newly delivered pose measurement? */
    double pose_x[3];
    double pose_q[4];     /* (w,x,y,z) */
    int mem_kind;
    double stamp;         /* stamped_masks: RGB time stamp of this frame (s) */
    double mask_stamp;    /* stamped_masks: time stamp of the image `mask` was computed on */
} roft_frame_input;

typedef struct {
    double pose[13];   /* v w x q */
    double twist[6];   /* v_O, w */
    int n_flow_points; /* N of the velocity stage, -1 if it did not run */
    int outlier_selected; /* -1 no test this frame, 0 pose+velocity kept, 1 velocity-only chosen */
    double outlier_L[2];
} roft_object_output;

int roft_default_config(roft_config* cfg, int width, int height, int flow_type);
int roft_default_object(roft_object_desc* obj);

int roft_engine_create(const roft_config* cfg, roft_engine** out);
int roft_engine_destroy(roft_engine* e);
int roft_object_add(roft_engine* e, const roft_object_desc* desc, int* obj_id);

/* inputs: one entry per object, in obj_id order.  Enqueues uploads and builds the frame program.  On an error
 * nothing of the frame has been consumed: the call can be repeated with corrected inputs. */
int roft_frame_submit(roft_engine* e, const roft_frame_input* inputs, int n_inputs);
/* A batch of n_frames consecutive frames (1 .. roft_config::max_batch_frames): inputs[t * n_objects + obj]. */
int roft_frames_submit(roft_engine* e, const roft_frame_input* inputs, int n_objects, int n_frames);
/* Enqueues every kernel of ROFTFilter::filtering_step for all objects and all submitted frames; returns without
 * waiting. */
int roft_step(roft_engine* e);
/* retention window of zero-copy DEVICE inputs in frames (see roft_frame_input) */
int roft_engine_retain_frames(const roft_engine* e);
int roft_sync(roft_engine* e);
/* Blocks until the last step has finished.  Any of the output pointers may be NULL. */
int roft_get_state(roft_engine* e, int obj_id, double pose13[13], double P12[144], double twist6[6],
                   double Pv[36]);
/* outputs of the last finished step for all objects (n_objects entries) */
int roft_get_outputs(roft_engine* e, roft_object_output* outs, int n_outs);
/* current propagated, binarised mask (H*W u8, {0,255}) of one object -> host buffer */
int roft_get_mask(roft_engine* e, int obj_id, uint8_t* mask_out);

/* Device-side log of the per-frame outputs (what ROFTFilter logs per frame, ROFTFilter.cpp:386-394):
 * a ring of n_frames x n_objects records written by the step itself, read back in one copy. */
int roft_engine_enable_log(roft_engine* e, int n_frames);
int roft_engine_get_log(roft_engine* e, int first_frame, int n_frames, roft_object_output* outs);
/* the same log as the reference writes it: rows[(f * n_objects + obj) * 19 ..] = pose(13: v w x q) | twist(6)
 * (`pose_estimate` + `velocity_estimate`, ROFTFilter.cpp:386-394) -- the per-object records a multi-GPU job gathers */
int roft_engine_get_log_rows(roft_engine* e, int first_frame, int n_frames, double* rows);

/* work enqueued since roft_engine_create */
typedef struct {
    long long frames;          /* frames stepped */
    long long batches;         /* roft_step calls */
    long long launches;        /* kernel launches + memsets + copies enqueued by roft_step */
    long long event_ops;       /* cross-stream waits + explicit event records enqueued by roft_step */
    long long h2d_bytes;       /* bytes of HOST inputs uploaded by the submit calls */
    long long h2d_copies;      /* ... in this many copies (images of consecutive frames that are consecutive in host memory go in one) */
} roft_engine_stats;
int roft_engine_get_stats(roft_engine* e, roft_engine_stats* out);

/* What the engine decided for, and the host spent on, each of the last batches (a ring of 64): a slow run explains itself.
 * The scheduling mode of a batch is a function of the batch INDEX, the object count and the number of engines this process holds on
 * the device (lanes are released early only by the ONLY engine of the device: a count taken at the submit, never a timing): `steady` = at least <batches in flight> batches have
 * been stepped since the engine was last idle (creation / roft_sync / anything that reads results); bursts release the pose
 * lanes early (`handoff`, `early_lanes`) and spread an outlier test over all the CUs to spare, steady batches halve that
 * (`outlier_parts_halved`); `early_lanes` is a bit mask: 1 / 2 = pose lane 0 / 1 released behind the batch's control blocks (its
 * objects start with the first step of a re-sync replay, whose twist an earlier batch published), 4 = both lanes because the
 * device has CUs to spare.  `throttled` is the MEASURED counterpart (the submit call had to wait for the in-flight bound) and
 * steers nothing.  Times: host steady clock in microseconds; t_done_us is when the HOST observed the batch complete (inside a
 * later submit call or roft_sync, which waits for the batches one by one in order), 0 while it has not. */
typedef struct {
    int batch;                 /* index since roft_engine_create */
    int frames;
    int steady, throttled, handoff, early_lanes, outlier_parts_halved;
    int launches, event_ops;   /* enqueued by roft_step for this batch */
    double t_submit_us;        /* entry of roft_frames_submit */
    double submit_us;          /* inside roft_frames_submit (frame programs, uploads) ... */
    double wait_us;            /* ... of which blocked on the in-flight bound */
    double step_us;            /* inside roft_step (launches) */
    double t_done_us;
} roft_batch_trace;
/* the last min(capacity, 64, batches so far) batches, oldest first */
int roft_engine_get_batch_trace(roft_engine* e, roft_batch_trace* out, int capacity, int* n_out);

/* HIP stream the engine enqueues on (as void*), for timing with hipEvents */
void* roft_engine_stream(roft_engine* e);

/* Kernel timing with HIP events on the engine's streams, accumulated over the steps since the last
 * roft_engine_get_timing.  enable: 0 off, 1 only flow_measure_kernel (a start / stop event pair bound to its dispatch, what
 * bench.py keeps on inside its timed region), 2 every launch group (adds ~10 event records per batch).
 * With timing on, the flow measurement's launches (the first 64 between two roft_engine_get_timing calls) are also timed
 * on the device's own 100 MHz clock -- every workgroup leaves its start and end, the entry "flow_measure_span" is first
 * workgroup in -> last workgroup out, the duration a kernel trace reports.  roft_engine_enable_timing creates its events
 * and sends one event-paired dispatch down the velocity stream before it returns (call it outside a timed region).
 * names/ms arrays are owned by the engine. */
int roft_engine_enable_timing(roft_engine* e, int enable);
int roft_engine_get_timing(roft_engine* e, int* n_out, const char*** names_out, const float** ms_out,
                           const int** launches_out);

/* ---- (2b) pinned host memory for images that are read in place ---------------------------------------------------- *
 * Image buffers handed over as ROFT_MEM_HOST are copied to the device by the submit call: every byte of depth, flow and mask
 * crosses the bus although the kernels read a few hundred KB of a frame (the pixels of the object's mask).  Buffers that live in
 * pinned, device-mapped host memory can be handed over as ROFT_MEM_DEVICE instead -- under the retention contract of DEVICE
 * inputs (roft_frame_input) -- and are then read in place: only the sectors the kernels touch cross the bus and nothing is
 * staged.  roft_host_alloc returns such memory from a pool (blocks are recycled by size; the first allocation of a size pins
 * pages, ~0.1 ms per MB), NULL when there is no device or the allocation fails: use ordinary memory and ROFT_MEM_HOST then.
 * The class facade (include/ROFT/Compat.h) allocates its image-sized buffers this way and ROFT::ROFTFilter keeps the frames of
 * the retention window alive, so the reference's executable reads its images from disk straight into memory the GPU reads. */
void* roft_host_alloc(size_t bytes);
void roft_host_free(void* p);              /* p from roft_host_alloc; NULL is ignored */
int roft_host_is_pinned(const void* p);    /* 1 when p lies inside a live block of the pool */

/* ---- (3) optical-flow producer (replaces the reference's NVIDIA-hardware flow source) ---------------- *
 * ROFT consumes pre-computed flow frames produced by cv::cuda::NvidiaOpticalFlow_{1_0,2_0}
 * (src/roft-lib/src/ImageOpticalFlowNVOF.cpp:100-159, tools/nvof/dumper/src/main.cpp:40-146).  MI355X has no
 * fixed-function flow unit: this is a dense pyramidal Lucas-Kanade on the CUs producing the same two products,
 * CV_32FC2 at grid 1 (NVOF 2.0 shape) or CV_16SC2 S10.5 at grid 4 (NVOF 1.0 shape), forward flow of the PREVIOUS
 * frame's pixels. */
typedef struct {
    int levels;      /* pyramid levels (1..6); width and height must be multiples of 2^(levels-1) */
    int radius;      /* window half size: (2r+1)^2 taps */
    int iterations;  /* Gauss-Newton iterations per level */
    float det_min;   /* pixels whose structure tensor determinant is below this keep the coarser estimate */
} roft_of_params;

int roft_default_of_params(roft_of_params* p);

/* host buffers: prev/cur gray u8 (H x W).  out_type ROFT_FLOW_F32C2 -> float[H][W][2];
 * ROFT_FLOW_S16C2 -> int16[H/4][W/4][2]. */
int roft_optical_flow(const uint8_t* prev_gray, const uint8_t* cur_gray, int W, int H, const roft_of_params* p,
                      int out_type, void* flow_out);

/* batched, device-resident producer: n image pairs per call, asynchronous on its own stream */
typedef struct roft_flow_producer roft_flow_producer;
int roft_flow_producer_create(int W, int H, int max_pairs, const roft_of_params* p, int out_type, int device,
                              roft_flow_producer** out);
int roft_flow_producer_destroy(roft_flow_producer* fp);
/* prev/cur/out: host arrays of n DEVICE pointers (u8 images in, flow frames out) */
int roft_flow_producer_run(roft_flow_producer* fp, const uint8_t* const* prev, const uint8_t* const* cur,
                           void* const* out, int n_pairs);
int roft_flow_producer_sync(roft_flow_producer* fp);
void* roft_flow_producer_stream(roft_flow_producer* fp);

/* ---- (4) diagnostics (tests and profiling tools; not needed by an integration) --------------------------------- */
/* The per-frame program builder without a device: the host-side mirror of CartesianQuaternionMeasurement::freeze's
 * Standard / PopBufferedMeasurement / RepeatOnlyVelocity state machine (cpp:92-348) and of the re-sync loop of
 * ROFTFilter::filtering_step (ROFTFilter.cpp:327-367) over n_frames pose-validity flags: per frame the number of UKF
 * steps, of corrections, the index of the step followed by the outlier test (-1 none) and the twist-ring slots replayed
 * (slots: n_frames x 10, -1 padded).  Any output may be NULL. */
int roft_debug_plan(const roft_config* cfg, const int* pose_valid, int n_frames, int* n_steps, int* n_corrections,
                    int* outlier, int* slots);
/* Which of the engine's HIP streams delay each other at the dispatch level (streams that the runtime mapped onto one hardware
 * queue): out[a * 5 + b] = microseconds until a one-workgroup kernel on stream b completes while stream a is placing a grid
 * larger than the device; ~15 = independent, >= 80 = queued behind it.  Order: pose lane 0, pose lane 1, velocity chain, mask
 * chain, upload stream.  Takes ~20 ms; not for a hot loop. */
int roft_debug_probe_streams(roft_engine* e, double out[25]);
/* The rate (sectors per second) at which the device serves scattered 64-byte sectors: 16 M reads at random sector-aligned offsets
 * of a 2 GiB scratch buffer, best of four launches -- the roofline of a gather-bound kernel such as the flow measurement (its
 * depth and flow samples are one sector each).  Allocates and frees 2 GiB of device memory; ~30 ms. */
int roft_debug_sector_rate(int device, double* sectors_per_second);
/* (100 MHz ticks, workgroups) per kernel the workgroups spent resident on the device since the last call, read and cleared; order:
 * mask_frame, mask_ingest, mask_general, flow_measure, skf_chain, features, ukf_chain, outlier_fused.  Only filled by libraries
 * built with -DROFT_RESIDENCY (tools/residency_budget.py: the CU x us budget of the pipeline). */
int roft_debug_get_residency(roft_engine* e, unsigned long long out[32]);
/* (A/B experiments on a whole engine only -- process-wide, so not for a process whose threads run other engines; a single
 * operator-level test takes the choice as an argument: roft_outlier_test_split.)
 * How the workgroups of one alternative of an outlier test share its work, for every test launched by this process from now on:
 * 1 = they split its TRIANGLES (windows merged in memory, the last workgroup scores; rows as well when the window does not fit
 * the LDS in one piece), 0 = they split only the ROWS of its window, -1 = the library's choice (1).  The results do not depend
 * on it (tests/test_parity_gpu.py). */
int roft_debug_outlier_split(int mode);
/* phase counters of one object's last kernels (only filled by libraries built with a -DROFT_*_PROFILE switch) */
int roft_debug_get_dbg(roft_engine* e, int obj_id, long long out[32]);

#ifdef __cplusplus
}
#endif
#endif
