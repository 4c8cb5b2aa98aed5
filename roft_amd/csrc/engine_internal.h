// engine_internal.h -- what the translation units of the host side of libroft_hip.so share: device buffers, the arrays of an
// engine, the host-side mirrors of the reference's source / measurement state machines (Sched), the engine object itself and
// the few functions that cross a file boundary.  Not installed, not part of the ABI (that is include/roft_engine.h).
//
//   engine.hip          error string, pinned host pool, defaults, stream sets, roft_engine_create / destroy, roft_object_add
//   engine_submit.hip   roft_frames_submit: the frame programs (build_pose_program), HOST staging, the control blocks of a batch
//   engine_step.hip     roft_step / roft_sync: the four-stream launch graph of a batch (step_batch) and its timing marks
//   engine_results.hip  state, outputs, log, masks, timing and batch-trace readers
//   engine_ops.hip      the operator-level entry points (one-object context: roft_flow_measurement ... roft_outlier_test)
//   engine_debug.hip    roft_debug_* (diagnostics and experiments)
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "roft_device.h"
#include "mesh_class.h"


namespace roft {
namespace host {

// sets the calling thread's error string (roft_last_error_string) and returns `code` (engine.hip)
int fail(int code, const std::string& msg);


#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess)                                                                              \
            return fail(ROFT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));               \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    hipError_t ensure(size_t count, bool zero = false)
    {
        if (count <= n && p) return hipSuccess;
        release();
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T));
        if (e != hipSuccess) { p = nullptr; return e; }
        n = count;
        if (zero) e = hipMemset(p, 0, std::max<size_t>(count, 1) * sizeof(T));
        return e;
    }
};

inline size_t flow_bytes(const DevFlowFmt& f)
{
    return (size_t)f.cols * f.rows * 2 * (f.type == ROFT_FLOW_S16C2 ? sizeof(int16_t) : sizeof(float));
}

inline DevCamera make_cam(const roft_camera& c)
{
    DevCamera d;
    d.W = c.width;
    d.H = c.height;
    d.wpr = c.width / 32;
    d.divider = (c.width == 640) ? 2 : 4;  // ROFTFilter.cpp:191-193
    d.fx = c.fx; d.fy = c.fy; d.cx = c.cx; d.cy = c.cy;
    return d;
}

inline int check_geometry(int W, int H)
{
    if (W <= 0 || H <= 0 || (W % 32) != 0 || (((size_t)W * H) % 64) != 0)
        return fail(ROFT_ERR_INVALID, "image width must be a multiple of 32 and width*height a multiple of 64");
    if ((size_t)W * H >= (1u << 24))
        return fail(ROFT_ERR_INVALID, "width*height must be < 2^24 (float-accumulated sampling index, hpp:237)");
    // (No bound from the LDS: the mask frames work on windows of a band's rows, the flow measurement and the feature kernel read
    //  planes that do not fit the LDS -- beyond ~1.1 Mpixel -- from memory, the general mask path lists its groups in pieces.
    //  The reference scans any cv::Mat, ImageOpticalFlowMeasurement.hpp:231-256.)
    return ROFT_OK;
}

// Device arrays for n objects of one geometry
struct Arrays {
    EngineArrays a{};
    DevBuf<ObjParams> params;
    DevBuf<ObjState> state;
    DevBuf<FrameCtrl> ctrl;
    DevBuf<uint32_t> planes;
    DevBuf<int32_t> map;
    DevBuf<FlowRec> cand, recs;
    DevBuf<double> norms;
    DevBuf<int> npts;
    DevBuf<MaskRec> mrec;
    DevBuf<unsigned> mask_general;
    DevBuf<uint32_t> feat_pix;
    DevBuf<float> feat_depth;
    DevBuf<uint32_t> zbuf;
    DevBuf<uint32_t> zmerge;   // merge slabs of the outlier test (EngineArrays::zmerge)
    DevBuf<int> zcount;
    // (re)allocates the merge slabs for n objects and tiles of tpix pixels: enough for the automatic band count at any number of
    // objects up to n (objects * bands <= max(n, CUs / 2)); a caller who asks for more bands than that gets the row split
    int ensure_zmerge(int n, size_t tpix)
    {
        const size_t slabs = std::max<size_t>((size_t)n, std::min<size_t>((size_t)n * kMaxOutlierParts, (size_t)std::max(device_cu_count() / 2, 1)));
        const size_t need = (size_t)kNumLin * slabs * 2 * tpix;
        if (need > zmerge.n || !zmerge.p || tpix != a.zmerge_stride || slabs != a.zmerge_slabs) {
            HIP_TRY(zmerge.ensure(need));
            a.zmerge_stride = tpix;
            a.zmerge_slabs = slabs;
        }
        HIP_TRY(zcount.ensure((size_t)kNumLin * n * 2 * kMaxOutlierParts, true));
        a.zmerge = zmerge.p;
        a.zcount = zcount.p;
        return ROFT_OK;
    }
    DevBuf<roft_object_output> log;
    DevBuf<unsigned long long> skf_started, residency;

    int alloc(int n_obj, int T, const DevCamera& cam, const DevFlowFmt& ffmt, int radius)
    {
        a.n_obj = n_obj;
        a.T = 1;
        a.cam = cam;
        a.ffmt = ffmt;
        a.plane_words = (size_t)cam.wpr * cam.H;
        const size_t npix = (size_t)cam.W * cam.H;
        a.cand_cap = ((int)((npix + radius - 1) / std::max(radius, 1)) + 9) & ~1;   // even: rows of a.cand stay 8-byte aligned
        a.feat_cap = (int)(npix / 2 + 8);
        a.tile_w = cam.W / cam.divider;
        a.tile_h = cam.H / cam.divider;
        HIP_TRY(params.ensure(n_obj, true));
        HIP_TRY(state.ensure(n_obj, true));
        HIP_TRY(ctrl.ensure((size_t)n_obj * T, true));
        HIP_TRY(planes.ensure((size_t)n_obj * kPlaneSlotsTotal * 2 * a.plane_words, true));
        HIP_TRY(mrec.ensure((size_t)2 * n_obj * (kMaxBatch + 1), true));   // two tables (batch parity)
        HIP_TRY(mask_general.ensure(n_obj, true));
        HIP_TRY(map.ensure((size_t)n_obj * npix, true));
        HIP_TRY(cand.ensure((size_t)n_obj * T * a.cand_cap));
        HIP_TRY(recs.ensure((size_t)n_obj * T * a.cand_cap));
        HIP_TRY(npts.ensure((size_t)n_obj * T, true));
        HIP_TRY(norms.ensure((size_t)n_obj * 3 * a.cand_cap));
        HIP_TRY(feat_pix.ensure((size_t)n_obj * kFeatRing * a.feat_cap));
        HIP_TRY(feat_depth.ensure((size_t)n_obj * kFeatRing * a.feat_cap));
        HIP_TRY(zbuf.ensure((size_t)2 * a.tile_w * a.tile_h));   // operator level only (roft_depth_likelihood)
        if (int rc = ensure_zmerge(n_obj, (size_t)a.tile_w * a.tile_h)) return rc;
        a.params = params.p; a.state = state.p; a.ctrl = ctrl.p; a.planes = planes.p; a.map = map.p;
        a.cand = cand.p; a.recs = recs.p; a.norms = norms.p; a.npts = npts.p; a.mrec = mrec.p;
        a.mask_general = mask_general.p;
        a.mrec_carry = mrec.p; a.slot_new = kSlotNew; a.slot_prev0 = -1; a.feat_pix = feat_pix.p; a.feat_depth = feat_depth.p;
        a.zbuf = zbuf.p;
        a.out_log = nullptr;
        a.log_cap = 0;
        a.max_tris = 0;
        a.max_verts = 0;
        a.ukf_chol_guard = 0.0;
        a.ukf_chol_guard_bil = 0.0;
        a.mask_wgs = 0;
        a.outlier_parts = 0;
        a.dev_error = nullptr;
        a.k1_span = nullptr;
        HIP_TRY(skf_started.ensure(1, true));
        HIP_TRY(residency.ensure(32, true));
        a.residency = residency.p;
        a.skf_started = nullptr;   // (the batched engine sets it; the operator level runs its kernels one after the other)
        a.handoff = 0;
        return ROFT_OK;
    }
};

inline void init_state(ObjState& st)
{
    std::memset(&st, 0, sizeof(st));
    for (PoseLane& pl : st.lane) { pl.pending_frame = -1; pl.outlier_selected = -1; }
    st.n_flow_points = -1;
}

inline void clear_ctrl(FrameCtrl& c)
{
    std::memset(&c, 0, sizeof(c));
    c.outlier_step = -1;
    c.feat_write = c.feat_read = -1;
}



struct FlowEntry {
    const void* ptr;
    int frame;   // frame index the flow was delivered with
    int owned;   // index into HostObject::owned when the engine holds its own copy, else -1
};

// Schedule-driven mirrors of the reference's source / measurement-model state machines.  Trivially copyable: a submit
// call works on the live copy and restores the snapshot taken at its start if it fails, so a failed call consumes
// nothing.
struct Sched {
    int frame_idx = 0;
    bool seg_available = false;        // ImageSegmentationOFAidedSource::segmentation_available_
    bool of_first_frame = true;        // ...::is_first_frame_
    bool flow_first_frame = true;      // ImageOpticalFlowMeasurement::is_first_frame_
    bool features_initialized = false; // ROFTFilter::outlier_rejection_features_initialized_
    int feat_slot = 0;                 // feature ring slot holding the buffered outlier-rejection features
    int feat_next = 0;                 // next ring slot to write
    int feat_use[kFeatRing];           // last batch that reads or writes each feature ring slot (-1: never used)
    int n_hist = 0;
    FlowEntry hist[kMaxFlowHist];      // last valid flows, newest first
    int n_stamps = 0;
    double stamps[30];                 // stamped source: RGB stamps of the last 30 valid flows, oldest first
    int n_vel = 0;
    int vel_buf[kTwistRing];           // twist_hist slots (CartesianQuaternionMeasurement::buffer_velocities_), oldest first
    int last_meas_slot = 0;            // slot of measurement_.head<6>()
    int cur_slot = 0;                  // B_LIN0 / B_LIN1: slot holding p_corr_belief_ (the other one holds buffered_belief_)
    int own[kNumLin] = {0, 1};         // pose chain lane that walks each of the two slots (always different lanes)
    int last_touch[kNumLin] = {-1, -1};   // last batch whose pose chain reads or writes each slot
    int flows_since_mask = 0;          // upper bound of the flows buffered since the last delivered mask
    const float* depth_prev = nullptr;
    Sched() { for (int& u : feat_use) u = -1; }
};

struct OwnedFlow {
    DevBuf<unsigned char> buf;
    int last_ref_frame = -1;   // last frame whose control block references the copy
};

struct HostObject {
    Sched s;
    int stepped_slot = 0, stepped_lane = 0;   // slot holding p_corr_belief_ after the last stepped frame, and its lane
    std::vector<OwnedFlow*> owned;   // engine copies of flows that outlived the zero-copy retention window
    DevBuf<float> verts;
    DevBuf<int32_t> tris;
    DevBuf<uint8_t> tri_flip;   // closed meshes only (mesh_class.h)
    ~HostObject() { for (auto* o : owned) delete o; }
};

// Device copies of HOST inputs: a ring of `retain` frame slots, each a bump allocator over chunks of device memory that
// are allocated when a frame first needs them and kept (a slot grows to the largest frame it ever held: 64 objects with
// their own 640x480 depth + CV_32FC2 flow + mask streams need 239 MB per slot, a shared scene 7 MB + the masks); identical
// host pointers within a frame (a scene shared by several objects) share one upload.
struct StageFrame {
    std::vector<DevBuf<unsigned char>*> chunks;
    size_t cur = 0, used = 0;   // bump pointer: chunk index, bytes used of it
    std::vector<std::pair<const void*, void*>> seen;
    StageFrame() = default;
    StageFrame(StageFrame&&) = default;
    StageFrame(const StageFrame&) = delete;
    ~StageFrame() { for (auto* c : chunks) delete c; }
};
constexpr size_t kStageChunk = (size_t)32 << 20;


}  // namespace host
}  // namespace roft

using namespace roft;
using namespace roft::host;

// The HIP streams of the engines of this process.  The runtime maps streams onto a few hardware queues in the order in
// which they are created; streams created after others were destroyed can end up sharing queues, and the chains of such
// an engine then run one after the other (measured: the second engine of a process tracked at a third of the rate of
// the first).  So a set of streams is created once per device and priority mode, handed to one engine at a time and
// never destroyed.
struct StreamSet {
    hipStream_t mask = nullptr, vel = nullptr, pose[kNumLin] = {nullptr, nullptr}, up = nullptr;
    int device = 0;
    bool priorities = true;
    bool in_use = false;   // handed to an engine
    bool parked = false;   // its busy streams share a hardware queue: kept alive (it shifts the runtime's round robin), handed out only
                           // when the device's cap of sets is reached
    int conflicts = 0;     // pairs of busy streams on one hardware queue when the set was created (-1: not probed)
};

struct GatherItem { const void* src; void* dst; size_t bytes; };   // one small pinned HOST image -> its staging copy
constexpr int kGatherCap = 8192;                                    // items per batch (8 frames x 1024 objects)
constexpr size_t kGatherMaxBytes = (size_t)2 << 20;                 // larger images go through the copy engine

struct roft_engine {
    roft_config cfg{};
    Arrays arr;
    // Three in-order chains per batch, one HIP stream each (ROFT_ONE_STREAM=1 puts them on one stream):
    hipStream_t stream = nullptr;       // mask chain: FrameCtrl upload, mask chain kernel, features
    hipStream_t vel_stream = nullptr;   // velocity chain: flow measurement, velocity filter
    hipStream_t pose_stream[kNumLin] = {nullptr, nullptr};  // pose chain, one stream per lane (BeliefSlot): UKF segments, outlier rejection
    hipStream_t up_stream = nullptr;    // uploads of HOST inputs and the copies of aged-out flows
    struct StreamSet* streams = nullptr;   // the pooled set the four above come from
    // Batches in flight.  The image chains of batch b+1 do not depend on the pose chain of batch b, so they run ahead
    // of it.  The lead is bounded on the host: the submit call of batch b returns only when batch b - lead has ended
    // (its pose chain, which implies its other chains).  Rings are sized for it:
    //   batch ring (device FrameCtrl blocks, staging, events) kBatchRing > lead;
    //   plane ring kPlaneSlots > lead * T + T + 1;  twist ring kTwistRing > lead * T + pose_frames_between + 2;
    //   feature ring kFeatRing >= T + 2 (re-use is ordered by feat_use);
    //   caller buffers / HOST staging: retain = hist_cap + lead * T + 2 frames.
    static constexpr int kBatchRing = 8;
    int T_max = 1;        // cfg.max_batch_frames
    int lead = 6;         // batches
    int hist_cap = 6;     // flows kept per object
    int retain = ROFT_RETAIN_FRAMES;
    DevBuf<FrameCtrl> dctrl[kBatchRing];
    FrameCtrl* stage[kBatchRing] = {};     // pinned staging blocks
    // Small HOST images in PINNED memory (the per-object masks of a delivery: 64 buffers of 300 KB) are not copied one
    // hipMemcpyAsync each but fetched by ONE kernel over the bus (engine_submit.hip, gather_copy_kernel): what to fetch, per batch
    std::vector<GatherItem> gather;                    // collected by stage_host during a submit
    GatherItem* gather_tab[kBatchRing] = {};           // pinned tables the kernel reads (kGatherCap entries each; allocated on first use)
    hipEvent_t ev_up[kBatchRing] = {};     // uploads of the batch on the device
    hipEvent_t ev_ctrl[kBatchRing] = {};   // FrameCtrl blocks of the batch on the device (and the mask chain of the batch before)
    hipEvent_t ev_mask[kBatchRing] = {};   // mask chain kernel of the batch complete
    hipEvent_t ev_part[kBatchRing] = {};   // the masks of the batch's frames 0 .. T - 2 complete (what its flow measurements read)
    hipEvent_t ev_prep[kBatchRing] = {};   // control blocks + ingested masks of the batch on the device (prepared on the upload stream)
    hipEvent_t ev_feat[kBatchRing] = {};   // features of the batch complete
    hipEvent_t ev_vel[kBatchRing] = {};    // the batch's velocity chain complete (velocity filter AND the feature kernel behind it)
    hipEvent_t ev_skf[kBatchRing] = {};    // twists of the batch complete (the velocity filter alone: what a pose lane waits for)
    hipEvent_t ev_done[kBatchRing][kNumLin] = {};   // pose chain of the batch complete (per lane)
    bool done_used[kBatchRing][kNumLin] = {};       // ... the lane had work in that batch
    bool multi = false;
    // ROFT_HOST_PROF=1: host time of the sections of the submit call / roft_step, printed by roft_engine_destroy
    bool host_prof = false;
    double hp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long hp_batches = 0;
    std::vector<HostObject*> objs;
    std::vector<Sched> backup;
    std::vector<ObjParams> h_params;
    std::vector<StageFrame> staging;       // [retain]
    ObjState* state_host = nullptr;   // pinned landing block of roft_get_state (velocity belief + corrected pose belief)
    int* dev_error = nullptr;         // pinned word a kernel raises when it gives up (EngineArrays::dev_error)
    // the submitted, not yet stepped batch
    bool submitted = false;
    int cur_T = 0;
    int n_segments[kNumLin] = {1, 1};     // pose chain segments per lane (1 + outlier tests of the busiest object)
    bool lin_any[kNumLin] = {false, false};   // some object has a frame on the lane in the batch
    // per lane of the submitted batch: objects with a frame on the lane, and how many of them START with a step whose twist was
    // published by an EARLIER batch (the first step of a re-sync replay reads the twist of pose_frames_between frames ago): such a
    // lane can run its first segment -- and the outlier test behind it -- before this batch's velocity filter exists (step_batch)
    int lane_objs[kNumLin] = {0, 0}, lane_old_first[kNumLin] = {0, 0};
    int relabel_wait[kNumLin] = {-1, -1};     // batch of the OTHER lane this lane's launches must follow (slots that changed lanes)
    bool any_feat = false, any_feat_now = false, had_uploads = false;
    unsigned feat_frames = 0;              // bit t: some object buffers outlier-rejection features in frame t of the batch
    unsigned new_mask_frames = 0;   // bit t: some object receives a mask in frame t of the batch
    int prev_T = 0;                 // frames of the batch stepped before
    int batch_counter = 0, frame_counter = 0;
    int completed_batches = 0, completed_frames = 0;
    int batch_end_frame[kBatchRing] = {};
    roft_engine_stats stats{};
    bool device_pointers_checked = false;   // ROFT_MEM_DEVICE inputs are looked up once, on the first submit
    bool throttled = false;   // MEASURED, diagnostics only (roft_batch_trace): the submit of the current batch had to wait for the in-flight bound
    // Scheduling mode of a batch, a function of the batch INDEX alone (round 5; rounds 3 - 4 keyed it on `throttled`, a host
    // timing, so that the launch graph itself differed from run to run): a batch is "steady" when at least `lead` batches have
    // been stepped since the engine was last idle (creation, roft_sync and everything that calls it), i.e. from the batch on
    // whose submit call may have to wait for the in-flight bound.  Bursts (fewer batches between two syncs) favour latency:
    // lanes released early, outlier tests on all the CUs to spare; steady batches favour occupancy.
    int idle_mark = 0;        // batch_counter when the engine was last known idle
    bool steady = false;      // mode of the batch being stepped
    bool alone_on_device = true;   // no other engine of this process holds a stream set on the device (asked at every submit: a count, not a timing)
    bool wait_value_ok = true;     // hipDeviceAttributeCanUseStreamWaitValue
    // trace of the last kTraceRing batches (roft_engine_get_batch_trace)
    static constexpr int kTraceRing = 64;
    roft_batch_trace trace[kTraceRing] = {};
    double cur_submit_t0 = 0.0, cur_submit_us = 0.0, cur_wait_us = 0.0;
    // Frame-granular hand-over velocity filter -> pose lanes (EngineArrays::handoff).  handoff_mode: 0 never, 1 while the host is
    // not throttled by the in-flight bound (bursts: the pipeline is filling or draining and latency is what counts), 2 always.
    int handoff_mode = 1;
    // ROFT_PREP_AHEAD / ROFT_MASK_PART_GATE, read when the engine is created: 0 never, 1 the default rule (a function of batch index
    // and object count: step_batch), 2 always, 3 whenever the batch index allows it whatever the object count.  No setting changes a result.
    int prep_mode = 1, part_mode = 1;
    int feat_mask_mode = 1;     // features kernel on the mask stream: 0 never, 1 at most one object per sixteen CUs, 2 always (ROFT_FEAT_ON_MASK)
    int lanes_wait_skf = 1;     // pose lanes without hand-over wait for the velocity filter's own event (ROFT_LANES_WAIT_SKF=0: for the features too)
    bool feat_dep_in_batch = false;        // an outlier test of the batch reads features buffered by a frame of the same batch
    unsigned long long skf_total = 0;      // velocity-filter workgroups launched so far (the value the lanes' gates wait for)
    bool vel_used[kBatchRing] = {};        // the batch's velocity chain ended with ev_vel (wait_batch waits for it as well)
    bool feat_used[kBatchRing] = {};       // the batch's feature kernel ran on the mask stream and ended with ev_feat (wait_batch waits for it as well)
    std::vector<int> feat_batch;           // [objects][kFeatRing] batch that last wrote each feature set (-1: none)
    // timing
    bool timing = false;
    int timing_level = 2;   // 1: only flow_measure_kernel (two events per batch), 2: every launch group
    std::vector<hipEvent_t> tev;
    std::vector<std::string> tnames_s;
    std::vector<const char*> tnames;
    std::vector<float> tms;
    std::vector<int> tlaunches;
    std::vector<int> tmark;    // kernel id per event interval (-1 = chain start)
    std::vector<int> tstream;  // stream of each mark (0 mask chain, 1 / 3 pose lanes, 2 velocity chain, 4 upload / preparation)
    // the flow measurement's launches on the device's own clock (timing runs): per launch and workgroup the 100 MHz wall clock at
    // its start and end, kSpanLaunches launches between two roft_engine_get_timing() calls (later ones are not stamped)
    static constexpr int kSpanLaunches = 64;
    DevBuf<unsigned long long> k1_span;
    std::vector<int> span_wgs;   // workgroups of each stamped launch
};

namespace roft {
namespace host {

inline double host_now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define HP_MARK(e, slot, t) do { if ((e)->host_prof) { const double _n = host_now_us(); (e)->hp_acc[slot] += _n - (t); (t) = _n; } } while (0)


}  // namespace host
}  // namespace roft

// ---- functions that cross a file boundary ---------------------------------------------------------------------------
// engine.hip
const std::string& last_error();   // the calling thread's error string
int check_dev_error(roft_engine* e);   // a kernel gave up (EngineArrays::dev_error): sticky
int wait_batch(roft_engine* e, int b, bool* waited = nullptr);   // blocks until batch b (and every earlier one) has ended on the GPU
bool alone_on_device(const StreamSet* mine);
__global__ void probe_blocker_kernel(long long ticks);
__global__ void probe_tiny_kernel(int* p);
__global__ void probe_sectors_kernel(const unsigned* buf, unsigned sector_mask, unsigned salt, unsigned* sink);
// engine_submit.hip
bool build_pose_program(const roft_config& cfg, Sched& o, const roft_frame_input& in, FrameCtrl& c);
// engine_step.hip
int step_batch(roft_engine* e);
