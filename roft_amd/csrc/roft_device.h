// roft_device.h -- device-visible data layout of the MI355X ROFT engine (gfx950 only).
//
// Everything the per-frame hot path touches lives in HBM in these structures; the host only
// uploads one small FrameCtrl block per object and frame and enqueues a fixed sequence of kernels per BATCH of
// frames (roft_frames_submit: 1 .. kMaxBatch frames; per-object chain kernels loop over the frames of the batch),
// so a frame needs no device->host round trip and a batch needs ~8 launches whatever its length.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/roft_engine.h"

namespace roft {

constexpr int kWave = 64;              // CDNA wavefront
constexpr int kNumLin = 2;             // belief lineages per object = lanes of the pose chain (see BeliefSlot)
constexpr int kNumBelief = 9;          // pose belief slots per object
constexpr int kMaxBatch = ROFT_MAX_BATCH_FRAMES;   // frames per roft_frames_submit
constexpr int kTwistRing = 64;         // twist history ring (velocity deque of the measurement model + frames in flight)
constexpr int kMaxFlowHist = ROFT_MAX_FLOW_CHASE;  // flows a new mask can be chased through (30-entry queue of the
                                       // time-stamped source, OpticalFlowQueueHandler.cpp:18-26; "all buffered" otherwise)
constexpr int kMaxOutlierParts = 8;    // workgroups one alternative of an outlier test is rendered by (horizontal bands)
constexpr int kMaxSteps = 10;          // UKF steps per frame (re-sync replays <= pose_frames_between + 1 <= 9)
constexpr int kPlaneSlots = 64;        // mask bit-plane ring per object (> frames in flight + one batch + 1)
constexpr int kFeatRing = 12;          // buffered outlier-rejection feature sets per object (>= kMaxBatch + 2; the host
                                       // waits before it re-uses a set an unfinished batch still reads)

// Pose belief slots.  The reference's re-sync swaps two Gaussians at every pose arrival -- `buffered_belief_ <-
// p_corr_belief_`, `p_corr_belief_ <- old buffered_belief_`, then replays the buffered velocities on the latter
// (ROFTFilter.cpp:331-350) -- so a filter with re-sync carries TWO interleaved belief lineages that never read each
// other: the one that is p_corr_belief_ between pose arrivals 2j and 2j+1, and the one that is between 2j+1 and 2j+2
// (it sits in buffered_belief_ meanwhile).  Here each lineage lives in one of the slots B_LIN0 / B_LIN1; the swap is a
// role change decided on the host (FrameCtrl::cur_slot), never a copy.  The pose chain has two LANES (HIP streams,
// FrameCtrl::lane) with their own scratch slots, cursors and z-buffers; a lineage is walked by the lane that owns its
// slot, so the re-sync replay of one lineage runs next to the ordinary steps of the other -- of the same batch and of
// the neighbouring ones -- which halves the serial UKF work per frame on the pose chain.
enum BeliefSlot { B_LIN0 = 0, B_LIN1 = 1, B_PRED = 2 /* + lane */, B_ALT0 = 4 /* + 2 lane */, B_ALT1 = 5 /* + 2 lane */,
                  B_SPARE = 8, B_CORR = B_LIN0 /* operator level: one lineage */ };
__host__ __device__ inline int b_alt(int lane, int k) { return B_ALT0 + 2 * lane + k; }

struct DevCamera {
    int W, H;
    int wpr;       // 32-bit words per bit-plane row
    int divider;   // render divider (ROFTFilter.cpp:191-193)
    double fx, fy, cx, cy;
};

struct DevFlowFmt {
    int type;   // ROFT_FLOW_*
    int cols, rows, grid;
    float scale;
};

struct PoseBelief {
    double mean[13];   // v w x q(wxyz)
    double cov[144];   // 12 x 12 row-major
};

struct ObjParams {
    double sigma_ang_vel[3];
    double psd_lin_acc[3];
    double v_q[6];
    double R_v[3], R_w[3], R_x[3], R_q[3];
    double r_flow[2];
    const float* verts;
    const int32_t* tris;
    int n_verts, n_tris;
    const double* q_override;  // operator level only: explicit 9x9 process noise (else Q(T) from the PSDs)
    // closed-surface classification of the mesh (mesh_class.h): tri_flip[t] = 1: triangle t of `tris` is wound clockwise seen
    // from outside; null: the mesh is not a closed orientable surface and every triangle is drawn.  (`tris` of a closed mesh is
    // in the facing-coherent walk order, tri_flip follows it.)
    const uint8_t* tri_flip;
};

// compact flow measurement record (what the SKF kernel consumes)
struct FlowRec {
    int u, v;
    float z, dx, dy;
};

// pose chain state of one lane of one object
struct PoseLane {
    int pc_frame, pc_step; // cursor inside the current batch (ukf_chain_kernel)
    int pending_frame;     // frame of the batch whose outlier test is pending between two pose chain segments, -1 none
    int outlier_selected;  // of the lane's last frame
    double outlier_L[2];
    double outlier_cnt[2];
    int ukf_status;
    int n_parts[2];        // partial sums each alternative of the last outlier test left (outlier_fused_kernel)
    // their partial sums of |depth - render| and sample counts, per alternative.  The sums are EXACT: every float term is
    // split into two integers (units of 2^-32 m and 2^-64 m, LikelihoodSum below), so that the likelihood does not depend on
    // how the samples are distributed over bands, strips, threads and waves -- the band count may follow the load
    long long part_hi[2][kMaxOutlierParts], part_lo[2][kMaxOutlierParts];
    double part_cnt[2][kMaxOutlierParts];
};

struct ObjState {
    double v_mean[6];
    double v_cov[36];
    PoseBelief belief[kNumBelief];
    double twist_hist[kTwistRing][6];
    // frame index + 1 of the twist a ring slot holds, published by the velocity filter AFTER the six values (agent-coherent
    // stores, tag last): the frame-granular hand-over to a pose lane that runs next to it (EngineArrays::handoff)
    int twist_tag[kTwistRing];
    PoseLane lane[kNumLin];
    int n_flow_points;     // N of the velocity stage of the last frame (-1: did not run)
    int n_feat[kFeatRing]; // buffered outlier-rejection samples (rank-even mask pixels) per feature ring slot
    int skf_status;
    // warm start of the covariance eigen-decomposition per lineage slot: [0] prediction input, [1] correction input
    double warm_V[kNumLin][2][144];
    int warm_age[kNumLin][2];  // 0 = no basis yet; a cold start is forced every kWarmRefresh uses
    long long dbg[32];     // phase cycle counters of the last ukf_step (ROFT_UKF_PROFILE builds only)
};

constexpr int kWarmRefresh = 32;

// one UKF launch for one object
struct StepDesc {
    int op;          // 0 = nop
    int src;         // belief slot read
    int do_predict;  // run the prediction (result also stored to B_PRED)
    int n_corr;      // 0: dst0 <- prediction; 1 or 2 corrections
    int type[2];     // ROFT_MEAS_*
    int dst[2];      // belief slots written
    int twist_slot;  // twist_hist index used as the velocity measurement
    int pad_;
};

struct alignas(16) FrameCtrl {
    double dt;
    const float* depth_prev;
    const float* depth_cur;
    const void* flow[kMaxFlowHist];  // [0] = this frame's flow, [j] = frame k-j
    const uint8_t* new_mask;         // raw u8 mask delivered this frame (device) or null
    int flow_valid;                  // this frame's flow exists and it is not the first frame
    int has_new_mask;
    int first_mask;                  // first mask ever: initialisation, not "new" (hpp:169-178)
    int stamped;                     // time-stamped source (…Stamped.hpp): n_region replaces the device-side flow count
    int n_region;                    // flows stored after the flow whose stamp matches the new mask's (0: none / no match)
    int n_hist;                      // valid entries of flow[]
    int slot_prev, slot_cur;         // bit-plane ring slots
    int vel_stage;                   // run the velocity stage this frame
    int twist_slot;                  // twist_hist slot written this frame
    int feat_write;                  // feature ring slot this frame's features are buffered into, -1: none
                                     // (first frame ROFTFilter.cpp:313-322, pose re-sync frames :353, and the
                                     // current-frame features of the outlier test without re-sync)
    int feat_read;                   // feature ring slot this frame's outlier test reads
    int n_steps;
    double pose_x[3];
    double pose_q[4];
    StepDesc steps[kMaxSteps];
    int outlier_step;                // index of the step followed by render + likelihood (-1 none)
    int force_mode;                  // operator level: force the mask mode (0 = decide on device)
    int frame_idx;                   // engine frame counter (row of the output log)
    int lane;                        // pose chain lane (stream) that walks this frame's steps (see BeliefSlot)
    int cur_slot;                    // B_LIN0 / B_LIN1: slot of p_corr_belief_ during this frame
    int pad_[2];
};

// Mask chain record of one frame of the batch and one object (k_mask.hip).  Row t + 1 of EngineArrays::mrec belongs to
// frame t; row 0 carries the state in from the batch before.
struct MaskRec {
    int mode;        // 0 copy, 1 propagate the last mask through this frame's flow, 2 new mask through the buffered flows
    int src_slot;    // plane slot of the source mask
    int n_flows;     // flows the source is chased through
    int src_binary;  // the source has no pixel of value 1 (nz plane == obj plane): order-free OR-scatter, only the obj
                     // plane of the result is written; otherwise the general, map-based path writes both planes
    int fbuf_n;      // flows buffered AFTER this frame (OF-aided source)
    int binary;      // == src_binary: the propagated mask after this frame is binary
    int new_count;   // non-zero pixels of the mask delivered with this frame (mask_ingest_kernel)
    int new_ones;    // ... of value 1 among them
};

// ---- launch wrappers (defined in the k_*.hip files) -------------------------------------------

// Order-free sum of non-negative float terms: a term x (clamped to 256 -- metres of depth difference) is x = hi 2^-32 + lo 2^-64
// with integers hi < 2^40, lo < 2^32, exactly for every float >= 2^-41; integer sums are associative, so any distribution of
// the terms over threads gives the same two totals (2^19.8 terms at most: hi < 2^60, lo < 2^52).
struct LikelihoodSum {
    long long hi = 0, lo = 0;
    __host__ __device__ void add(float x)
    {
        const double xs = fmin((double)x, 256.0) * 4294967296.0;
        const double fl = floor(xs);
        hi += (long long)fl;
        lo += (long long)((xs - fl) * 4294967296.0);
    }
    __host__ __device__ static double value(long long hi, long long lo) { return (double)hi * (1.0 / 4294967296.0) + (double)lo * (1.0 / 4294967296.0 / 4294967296.0); }
};

struct EngineArrays {
    int n_obj;
    DevCamera cam;
    DevFlowFmt ffmt;
    ObjParams* params;       // [n_obj]
    ObjState* state;         // [n_obj]
    int T;                   // frames of the current batch
    FrameCtrl* ctrl;         // [T][n_obj] (current batch)
    uint32_t* planes;        // [n_obj][kPlaneSlotsTotal][2][wpr*H]   (nz plane, obj plane)
    // The mask chain's per-batch tables exist twice (batch parity): the preparation of batch b + 1 -- control block
    // upload, counter reset, ingest of the delivered masks -- runs on a stream of its own while the chain of batch b
    // is still walking its frames.
    MaskRec* mrec;           // [kMaxBatch + 1][n_obj] of this batch (row t + 1: frame t; row 0 is never written)
    const MaskRec* mrec_carry;  // [n_obj] state after the last frame of the batch before (a row of the other table), or
                             // row 0 of this one (zeros) for the first batch
    int slot_new;            // plane slots receiving the masks ingested by this batch: slot_new + frame of the batch
    int slot_prev0;          // ring slot of the mask BEFORE frame 0 of the batch (frame t reads slot_prev0 + t, mod the ring:
                             // the flow measurement starts its plane loads without waiting for the control block), -1: unknown
    unsigned* mask_general;  // [n_obj] bit t: frame t of the batch is left to mask_general_kernel (three-valued source);
                             // set by the frame kernels, read and cleared by mask_general_kernel
    int32_t* map;            // [n_obj][W*H] scatter map of the general (non-binary) mask path, all-zero between frames
    FlowRec* cand;           // [T][n_obj][cand_cap] candidate scratch
    FlowRec* recs;           // [T][n_obj][cand_cap] kept flow records
    int* npts;               // [T][n_obj] N of the velocity stage (-1: did not run)
    double* norms;           // [n_obj][3 * cand_cap] SKF scratch (innovations + norms when N > LDS capacity)
    uint32_t* feat_pix;      // [n_obj][kFeatRing][feat_cap] buffered feature pixel (v << 16 | u)
    float* feat_depth;       // [n_obj][kFeatRing][feat_cap]
    uint32_t* zbuf;          // [2][tile_h*tile_w] float bits, +inf = empty: z-buffers of the operator-level likelihood
    // Outlier test with the TRIANGLES of an alternative split over several workgroups (k_render.hip): every workgroup leaves its
    // window in a slab of its own, [lane][zmerge_slabs: object * parts + part][alternative][zmerge_stride] float bits, zcount
    // counts the arrivals and the workgroup that arrives last merges the slabs.
    uint32_t* zmerge;
    int* zcount;             // [lane][object][alternative][band], zero between two tests
    size_t zmerge_stride;    // >= tile_w * tile_h
    size_t zmerge_slabs;     // slabs per lane (>= objects * workgroups per alternative, else the rows are split instead)
    int cand_cap, feat_cap;
    size_t plane_words;      // wpr*H
    int tile_w, tile_h;
    int max_tris, max_verts; // largest mesh among the objects
    double ukf_chol_guard;      // roft_config::ukf_cholesky_guard (0: always the eigen-decomposition)
    double ukf_chol_guard_bil;  // roft_config::ukf_cholesky_guard_bilinear
    roft_object_output* out_log;  // [log_cap][n_obj] per-frame outputs, or null
    int log_cap;
    int mask_wgs;            // roft_config::mask_workgroups_per_object (0: bands of ~20 image rows)
    int outlier_parts;       // roft_config::outlier_bands_per_alternative (0: by the device's CU count)
    int* dev_error;          // one word of pinned host memory (or null): ROFT_DEV_ERROR_* raised by a kernel that gave up
    // Frame-granular hand-over velocity filter -> pose lanes: the lanes' kernels of a batch are released by
    // the command processor as soon as every workgroup of the batch's velocity filter is resident (skf_started, a running
    // count the host waits on with hipStreamWaitValue64) and take each twist when its tag appears, instead of starting behind
    // the filter's last frame.  0: the lanes run behind the velocity chain's completion event (tags are still published).
    int handoff;
    unsigned long long* skf_started;   // running count of velocity-filter workgroups that have started (or null)
    unsigned long long* residency;     // [16][2] (100 MHz ticks, workgroups) a kernel's workgroups spent resident: only written by
                                       // libraries built with -DROFT_RESIDENCY (tools/residency_budget.py)
    unsigned long long* k1_span;  // [T][n_obj][2] of THIS launch of the flow measurement, or null: 100 MHz wall clock at which each
                                  // workgroup started and ended (timing runs only: the launch's span on the device's own clock)
};

// CU residency accounting (-DROFT_RESIDENCY builds only): every workgroup adds the time between its first instruction and the end
// of its thread 0 to its kernel's counter -- what a workgroup really occupies its share of a CU for, early exits included.
enum ResidencyKernel { RK_MASK_FRAME = 0, RK_MASK_INGEST, RK_MASK_GENERAL, RK_FLOW_MEASURE, RK_SKF_CHAIN, RK_FEATURES, RK_UKF_CHAIN, RK_OUTLIER,
                       RK_MASK_FRAME_EMPTY /* the workgroups of mask_frame_kernel that found no pixel in their band (counted here INSTEAD of RK_MASK_FRAME) */,
                       RK_MASK_FRESH, RK_MASK_FRESH_EMPTY /* the same two for the two-wave workgroups of the frames that deliver a mask */, RK_COUNT };
#ifdef ROFT_RESIDENCY
struct ResidencyTimer {
    unsigned long long* p;
    long long t0;
    __device__ ResidencyTimer(unsigned long long* base, int kid) : p(base ? base + 2 * kid : nullptr), t0(wall_clock64()) {}
    __device__ ~ResidencyTimer()
    {
        if (p && threadIdx.x == 0) { atomicAdd(p, (unsigned long long)(wall_clock64() - t0)); atomicAdd(p + 1, 1ull); }
    }
    __device__ void rekind(unsigned long long* base, int kid) { if (base) p = base + 2 * kid; }
};
#define ROFT_RESIDENT(a, kid) ResidencyTimer roft_resident_timer_((a).residency, (kid))
#define ROFT_RESIDENT_AS(a, kid) roft_resident_timer_.rekind((a).residency, (kid))
#else
#define ROFT_RESIDENT(a, kid) do {} while (0)
#define ROFT_RESIDENT_AS(a, kid) do {} while (0)
#endif

constexpr int ROFT_DEV_ERROR_TWIST_WAIT = 2;     // ukf_chain_kernel: a twist it was told to wait for was not published within two seconds
constexpr int ROFT_DEV_ERROR_MASK_BARRIER = 1;   // (rounds 2 - 3: the persistent mask chain's barrier in memory; no kernel raises it any more)

#define ROFT_LDS __attribute__((address_space(3)))

// LDS address of an object in DYNAMIC LDS, pinned in a scalar register.  The base of `extern __shared__` memory is not
// a link-time constant: left to itself the compiler re-reads it from a table in memory wherever it is used -- a scalar
// load and an s_waitcnt in front of every LDS atomic of the scatter loops.
__device__ __forceinline__ ROFT_LDS uint32_t* pin_lds(uint32_t* p)
{
    uint32_t off = (uint32_t)(uintptr_t)(ROFT_LDS uint32_t*)p;
    asm volatile("" : "+s"(off));
    return (ROFT_LDS uint32_t*)(uintptr_t)off;
}

// Control block -> LDS with one 16-byte load per thread (threads 0 .. sizeof(FrameCtrl)/16 - 1; the caller's barrier
// follows): read field by field from global memory, the compiler sinks every load to its first use and the kernel pays
// one memory latency per field.
__device__ inline void stage_ctrl(FrameCtrl* s_c, const FrameCtrl& g)
{
    constexpr int n16 = (int)(sizeof(FrameCtrl) / 16);
    for (int i = threadIdx.x; i < n16; i += blockDim.x)
        reinterpret_cast<uint4*>(s_c)[i] = reinterpret_cast<const uint4*>(&g)[i];
}

// control block of frame t of the batch
__device__ inline const FrameCtrl& frame_ctrl(const EngineArrays& a, int t, int obj)
{
    return a.ctrl[(size_t)t * a.n_obj + obj];
}

// row of the device-side output log a frame writes (null when logging is off)
__device__ inline roft_object_output* log_row(const EngineArrays& a, const FrameCtrl& c, int obj)
{
    if (!a.out_log) return nullptr;
    return a.out_log + (size_t)(c.frame_idx % a.log_cap) * a.n_obj + obj;
}

constexpr int kSlotNew = kPlaneSlots;       // first plane slot receiving ingested masks (EngineArrays::slot_new)
constexpr int kPlaneSlotsTotal = kPlaneSlots + 2 * kMaxBatch;   // ring + two sets of ingest slots (batch parity)

// Mask mode of a frame (ImageSegmentationOFAidedSource::step_frame, hpp:169-226), decided on the device
// because it depends on whether the newly delivered mask is empty:
// 0 copy (no flow, no usable new mask); 1 propagate the last mask through this frame's flow with
// mask(0,0) forced to 0 (hpp:221-226); 2 new mask chased through the buffered flows (hpp:211-219).
// fbuf_n = flows buffered before this frame, new_count = non-zero pixels of the mask delivered with it.
__device__ inline int decide_mode(const FrameCtrl& c, int new_slot, int fbuf_n, int new_count, int frames_between,
                                  int& src_slot, int& n_flows)
{
    const int n_avail = fbuf_n + (c.flow_valid ? 1 : 0);
    int mode;
    if (c.force_mode == 3) {  // operator level: map() + remap() of the given mask through n flows
        src_slot = new_slot;
        n_flows = n_avail;
        mode = 2;
    } else if (c.has_new_mask && !c.first_mask && new_count > 0) {
        src_slot = new_slot;
        if (c.stamped && c.n_region <= 0) {
            // …Stamped.hpp:229-236: no flow after the mask's stamp in the queue -> the NEW mask through this frame's
            // flow only, mask(0,0) forced to 0 (mode 1 semantics on the new mask)
            n_flows = 1;
            return c.flow_valid ? 1 : 0;
        }
        n_flows = c.stamped ? c.n_region : n_avail;
        mode = 2;
    } else {
        src_slot = (c.has_new_mask && c.first_mask) ? new_slot : c.slot_prev;
        n_flows = 1;
        return c.flow_valid ? 1 : 0;
    }
    // map(): only the last frames_between flows of the region when that number is known (hpp:239-245)
    if (frames_between > 0 && n_flows > frames_between) n_flows = frames_between;
    if (n_flows > c.n_hist) n_flows = c.n_hist;   // (the host refuses frames whose history it cannot supply)
    return mode;
}

// flows buffered after a frame (flow_buffer_ of the OF-aided source): a consumed new mask clears the buffer
// (hpp:218), an EMPTY new mask clears it only when the number of frames between masks is unknown (hpp:192-196)
__device__ inline int next_fbuf(const FrameCtrl& c, int fbuf_n, int new_count, int mode, int frames_between)
{
    if (c.has_new_mask && !c.first_mask && new_count == 0 && frames_between <= 0) fbuf_n = 0;
    int n = fbuf_n + (c.flow_valid ? 1 : 0);
    if (n > kMaxFlowHist) n = kMaxFlowHist;
    return (mode == 2) ? 0 : n;
}

__host__ __device__ inline size_t plane_offset(const EngineArrays& a, int obj, int slot, int which)
{
    return (((size_t)obj * kPlaneSlotsTotal + slot) * 2 + which) * a.plane_words;
}

// frame t's new masks -> plane slot slot_new + t, their pixel counts -> mrec row t + 1 (zeroed before: mask_reset_tables)
void launch_mask_ingest(const EngineArrays& a, int t, hipStream_t s, hipEvent_t stop = nullptr);
// the batch's control blocks (pinned staging -> a.ctrl) and the ingest of every mask it delivers in ONE launch; the ingest counters
// must be zero (they are: mask_general_kernel's final launch leaves them so).  false: not launched (too many delivering frames)
bool launch_ctrl_ingest(const void* staging, const EngineArrays& a, size_t n16, unsigned new_mask_frames, hipStream_t s, hipEvent_t stop = nullptr);
// Zeroes what the ingest kernels accumulate into: the counters of mrec rows 1 .. T (mask_general is cleared by its only
// reader, mask_general_kernel: this reset may run while the chain before still sets bits).
// (Inside the engine the control block upload kernel does this; the operator-level entry points call it.)
void launch_mask_reset(const EngineArrays& a, hipStream_t s);
__device__ inline void mask_reset_tables(const EngineArrays& a, size_t i)   // thread i of a grid of >= (T + 1) * n_obj threads
{
    if (i < (size_t)a.T * a.n_obj) {
        MaskRec& r = a.mrec[(size_t)a.n_obj + i];
        r.new_count = 0;
        r.new_ones = 0;
    }
}
// Mask chain of the batch (after the reset and the ingest of its new masks): one launch per frame (binary masks; many small
// workgroups per object, nothing persistent) and one kernel for the frames of objects with three-valued masks.
// new_mask_frames: bit t = some object receives a mask in frame t (those frames are split finer).  Returns the number of
// launches.
int launch_mask_chain(const EngineArrays& a, int frames_between, int flow_aided, unsigned new_mask_frames, hipStream_t s, hipEvent_t stop = nullptr,
                      hipEvent_t stop_early = nullptr);   // stop_early: completes with the masks of frames 0 .. T - 2
void launch_planes_to_mask(const uint32_t* nz, const uint32_t* ob, int npix, uint8_t* mask, hipStream_t s);
// `stop` / `start` (optional): HIP events bound to the kernel's own dispatch (hipExtLaunchKernelGGL) -- they complete
// with the kernel, without the extra barrier packet and host call of a hipEventRecord behind it.
// Flow measurement of every (frame, object) of the batch in one launch.
void launch_flow_measure(const EngineArrays& a, double depth_max, int radius, hipStream_t s,
                         hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
// Velocity filter: one workgroup per object walks the frames of the batch.
void launch_skf_chain(const EngineArrays& a, int reweight, hipStream_t s, hipEvent_t stop = nullptr);
// operator level: explicit (y, H) arrays, x_pred/P_pred in, x/P out (all device pointers)
void launch_skf_arrays(const double* x_pred, const double* P_pred, int N, const double* y, const double* H,
                       const double* Rdiag, int reweight, double* norms, double* x_out, double* P_out, int* status,
                       hipStream_t s);
// same, fed with compact flow records through the accessor the engine uses (parity tests)
void launch_skf_records(const double* x_pred, const double* P_pred, int N, const FlowRec* recs, DevCamera cam, double dt,
                        const double* Rdiag, int reweight, double* norms, double* x_out, double* P_out, int* status,
                        hipStream_t s);
void launch_kf_predict(const double* x, const double* P, const double* qdiag, double* xo, double* Po, hipStream_t s);
// Pose chain segment: every object runs its UKF steps from its cursor up to and including its next outlier-rejection
// step (then launch_outlier and another segment follow) or to the end of the batch.
// Of every object only the frames of lane `lin` are walked.
void launch_ukf_chain(const EngineArrays& a, roft_ut_params ut, bool first_segment, int lin, hipStream_t s, hipEvent_t stop = nullptr);
void launch_features(const EngineArrays& a, hipStream_t s, hipEvent_t stop = nullptr, unsigned feat_frames = 0);   // after the mask chain of the batch; feat_frames: bit t = frame t buffers features (0: all frames)
// Operator-level overrides of the outlier test's launch shape (roft_render_depth / roft_outlier_test: the parity tests drive
// the engine's kernel through every configuration); the engine passes none.
struct OutlierLaunchOpts {
    int parts = 0;            // horizontal bands per alternative (0: by the CUs to spare, -d: that count / d)
    int no_vertex_cache = 0;  // project the vertices per triangle instead of once into LDS
    int window_pixels = 0;    // cap of the LDS depth window in pixels (> 0: forces the strip path for larger windows)
    float* tile_dump = nullptr;   // [2][tile_h][tile_w], zero-filled: receives the rendered window of both alternatives
    int split = -1;           // several workgroups per alternative share its TRIANGLES (1) or only its window's rows (0); -1: the default (triangles)
};
// render + likelihood of the pending tests of a lane (the decision is the first thing the next pose chain segment does)
void launch_outlier(const EngineArrays& a, int lin, hipStream_t s, hipEvent_t stop = nullptr, const OutlierLaunchOpts* opts = nullptr);
void set_outlier_split(int mode);   // process-wide override of OutlierLaunchOpts::split (-1: none)
void launch_outlier_only(const EngineArrays& a, hipStream_t s);  // operator level: likelihood + decision on filled z-buffers

// multiProcessorCount of the calling thread's current device (cached per device ordinal; 256 on MI355X)
int device_cu_count();
// hipFuncAttributeMaxDynamicSharedMemorySize once per (kernel, device): the attribute belongs to the kernel's code object
// on ONE device, and a process may own engines on several
hipError_t set_max_dynamic_lds(const void* func, int bytes);

// operator-level helpers on raw device buffers (used by the C ABI operator entry points)
void launch_expand_yh(const FlowRec* recs, const int* n, DevCamera cam, double dt, int32_t* uv, double* y,
                      double* H, int cap, hipStream_t s);

}  // namespace roft
