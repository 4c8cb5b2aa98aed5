// engine.hip -- host side of libroft_hip.so: the C ABI of include/roft_engine.h.
//
// (2) the batched engine mirrors ROFTFilter::filtering_step (src/roft-lib/src/ROFTFilter.cpp:255-452)
//     as a per-frame *program*: the data-dependent control flow of the reference that depends only
//     on the delivery schedule (is a pose / mask / flow present this frame, how many buffered
//     velocities are replayed by the re-sync, ROFTFilter.cpp:327-367 and
//     CartesianQuaternionMeasurement.cpp:92-348) is resolved here on the host into one FrameCtrl
//     block per object; everything that depends on image content or filter state (is the new mask
//     empty, N < 3, the outlier decision, which matrix square root) is resolved inside the kernels.  A frame
//     is therefore one small H2D copy plus three in-order chains of batched launches (mask, velocity and pose
//     chain, one HIP stream each, roft_step) with several frames in flight and no D2H sync.
// (1) the operator-level entry points run the same kernels on a private one-object context.
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

using namespace roft;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

}  // namespace

namespace roft {
// shared with flow_producer.hip
int set_last_error(int code, const std::string& msg) { return fail(code, msg); }

int device_cu_count()
{
    static std::mutex mu;
    static std::vector<int> cus;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 256;
    std::lock_guard<std::mutex> lk(mu);
    if ((int)cus.size() <= dev) cus.resize(dev + 1, 0);
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

hipError_t set_max_dynamic_lds(const void* func, int bytes)
{
    static std::mutex mu;
    static std::vector<std::pair<const void*, int>> done;   // (kernel, device) pairs already raised to >= bytes
    static std::vector<int> done_bytes;
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev)) return e;
    std::lock_guard<std::mutex> lk(mu);
    for (size_t i = 0; i < done.size(); ++i)
        if (done[i].first == func && done[i].second == dev) {
            if (done_bytes[i] >= bytes) return hipSuccess;
            const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e == hipSuccess) done_bytes[i] = bytes;
            return e;
        }
    const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) { done.emplace_back(func, dev); done_bytes.push_back(bytes); }
    return e;
}
}  // namespace roft

namespace {

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

size_t flow_bytes(const DevFlowFmt& f)
{
    return (size_t)f.cols * f.rows * 2 * (f.type == ROFT_FLOW_S16C2 ? sizeof(int16_t) : sizeof(float));
}

DevCamera make_cam(const roft_camera& c)
{
    DevCamera d;
    d.W = c.width;
    d.H = c.height;
    d.wpr = c.width / 32;
    d.divider = (c.width == 640) ? 2 : 4;  // ROFTFilter.cpp:191-193
    d.fx = c.fx; d.fy = c.fy; d.cx = c.cx; d.cy = c.cy;
    return d;
}

int check_geometry(int W, int H)
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

void init_state(ObjState& st)
{
    std::memset(&st, 0, sizeof(st));
    for (PoseLane& pl : st.lane) { pl.pending_frame = -1; pl.outlier_selected = -1; }
    st.n_flow_points = -1;
}

void clear_ctrl(FrameCtrl& c)
{
    std::memset(&c, 0, sizeof(c));
    c.outlier_step = -1;
    c.feat_write = c.feat_read = -1;
}

}  // namespace

// =================================================================================================
// batched engine
// =================================================================================================

// FrameCtrl upload without the copy engine: a kernel reads the pinned (device-visible) staging block and
// writes the device copy, so the control blocks of a batch travel in-order on the compute queue instead of
// through an SDMA copy with its cross-engine signalling.
// Control blocks of a batch: pinned host staging -> device, and the reset of what the batch's mask chain accumulates
// into (ingest counters, the bits of the frames left to mask_general_kernel) on the way.  (a.ctrl, a.mrec: this batch's.)
__global__ void ctrl_upload_kernel(const uint4* __restrict__ src, EngineArrays a, size_t n16, int reset)
{
    uint4* dst = reinterpret_cast<uint4*>(a.ctrl);
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = i0; i < n16; i += stride) dst[i] = src[i];
    if (reset)
        for (size_t i = i0; i < (size_t)a.T * a.n_obj; i += stride) roft::mask_reset_tables(a, i);
}

// Diagnostics (roft_debug_probe_streams): a dispatch that cannot be placed completely keeps its hardware queue busy until its
// last workgroup is placed; a one-workgroup kernel on another stream completes at once unless the two streams share a queue.
__global__ void probe_blocker_kernel(long long ticks)
{
    extern __shared__ unsigned char smem[];
    smem[threadIdx.x] = 0;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void probe_tiny_kernel(int* p) { if (threadIdx.x == 0) *p = 1; }

// Diagnostics (roft_debug_sector_rate): every thread reads eight words at hashed, 64-byte-aligned offsets of a large buffer --
// the access pattern of the flow measurement's depth and flow gathers, without anything else.
__global__ __launch_bounds__(1024) void probe_sectors_kernel(const unsigned* buf, unsigned sector_mask, unsigned salt, unsigned* sink)
{
    const unsigned t = blockIdx.x * 1024u + threadIdx.x;
    unsigned acc = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        unsigned x = t * 8u + (unsigned)j + salt;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        acc += buf[(size_t)(x & sector_mask) * 16];
    }
    if (acc == 0x12345678u) sink[t] = acc;   // (never: keeps the loads)
}

namespace {

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

}  // namespace

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
    hipEvent_t ev_up[kBatchRing] = {};     // uploads of the batch on the device
    hipEvent_t ev_ctrl[kBatchRing] = {};   // FrameCtrl blocks of the batch on the device (and the mask chain of the batch before)
    hipEvent_t ev_mask[kBatchRing] = {};   // mask chain kernel of the batch complete
    hipEvent_t ev_part[kBatchRing] = {};   // the masks of the batch's frames 0 .. T - 2 complete (what its flow measurements read)
    hipEvent_t ev_prep[kBatchRing] = {};   // control blocks + ingested masks of the batch on the device (prepared on the upload stream)
    hipEvent_t ev_feat[kBatchRing] = {};   // features of the batch complete
    hipEvent_t ev_vel[kBatchRing] = {};    // twists of the batch complete
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
    bool feat_dep_in_batch = false;        // an outlier test of the batch reads features buffered by a frame of the same batch
    unsigned long long skf_total = 0;      // velocity-filter workgroups launched so far (the value the lanes' gates wait for)
    bool vel_used[kBatchRing] = {};        // the batch's velocity chain ended with ev_vel (wait_batch waits for it as well)
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

static inline double host_now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define HP_MARK(e, slot, t) do { if ((e)->host_prof) { const double _n = host_now_us(); (e)->hp_acc[slot] += _n - (t); (t) = _n; } } while (0)

// a kernel gave up (EngineArrays::dev_error): sticky -- the filter state of the objects is no longer what the reference
// would hold
static int check_dev_error(roft_engine* e)
{
    const int code = e->dev_error ? *reinterpret_cast<volatile int*>(e->dev_error) : 0;
    if (code == 0) return ROFT_OK;
    if (code == ROFT_DEV_ERROR_TWIST_WAIT)
        return fail(ROFT_ERR_DEVICE, "a pose lane waited two seconds for a twist of the velocity filter it runs next to and gave up "
                                     "(frame-granular hand-over; ROFT_HANDOFF=0 disables it): the results of the batch are invalid");
    return fail(ROFT_ERR_DEVICE, "a kernel reported error " + std::to_string(code));
}

// blocks until batch b (and therefore every earlier one) has ended on the GPU
static int wait_batch(roft_engine* e, int b, bool* waited = nullptr)
{
    if (waited) *waited = false;
    if (b < e->completed_batches || b >= e->batch_counter) return ROFT_OK;
    // (with the frame-granular hand-over a pose lane can end before the features kernel behind the velocity filter does)
    if (e->vel_used[b % roft_engine::kBatchRing]) {
        if (waited && hipEventQuery(e->ev_vel[b % roft_engine::kBatchRing]) == hipErrorNotReady) *waited = true;
        (void)hipGetLastError();
        HIP_TRY(hipEventSynchronize(e->ev_vel[b % roft_engine::kBatchRing]));
    }
    for (int l = 0; l < kNumLin; ++l)
        if (e->done_used[b % roft_engine::kBatchRing][l]) {
            if (waited && hipEventQuery(e->ev_done[b % roft_engine::kBatchRing][l]) == hipErrorNotReady) *waited = true;
            (void)hipGetLastError();   // (hipErrorNotReady is not an error of this call)
            HIP_TRY(hipEventSynchronize(e->ev_done[b % roft_engine::kBatchRing][l]));
        }
    {
        roft_batch_trace& tr = e->trace[b % roft_engine::kTraceRing];
        if (tr.batch == b && tr.t_done_us == 0.0) tr.t_done_us = host_now_us();
    }
    e->completed_batches = b + 1;
    e->completed_frames = e->batch_end_frame[b % roft_engine::kBatchRing];
    return check_dev_error(e);
}

extern "C" {

const char* roft_last_error_string(void) { return g_last_error.c_str(); }

int roft_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- pinned host memory pool (roft_engine.h section 2b) ----
extern "C++" {
namespace {
struct HostPool {
    std::mutex mu;
    std::map<uintptr_t, size_t> live;                       // blocks handed out: start -> bytes
    std::map<size_t, std::vector<void*>> spare;             // recycled blocks by size
    size_t spare_bytes = 0;
};
HostPool& host_pool() { static HostPool* p = new HostPool(); return *p; }   // (never destroyed: buffers of static objects may outlive main)
constexpr size_t kHostPoolGranule = (size_t)64 << 10;
constexpr size_t kHostPoolSpareCap = (size_t)512 << 20;   // recycled bytes kept before blocks go back to the runtime
}  // namespace
}  // extern "C++"

void* roft_host_alloc(size_t bytes)
{
    if (bytes == 0) return nullptr;
    static const int n_dev = roft_device_count();
    if (n_dev <= 0) return nullptr;
    const size_t sz = (bytes + kHostPoolGranule - 1) / kHostPoolGranule * kHostPoolGranule;
    HostPool& hp = host_pool();
    {
        std::lock_guard<std::mutex> lk(hp.mu);
        auto it = hp.spare.find(sz);
        if (it != hp.spare.end() && !it->second.empty()) {
            void* p = it->second.back();
            it->second.pop_back();
            hp.spare_bytes -= sz;
            hp.live[reinterpret_cast<uintptr_t>(p)] = sz;
            return p;
        }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, sz, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess || !p) { (void)hipGetLastError(); return nullptr; }
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess || dp != p) {   // (must be addressable by the GPU at the same address)
        (void)hipGetLastError();
        (void)hipHostFree(p);
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(hp.mu);
    hp.live[reinterpret_cast<uintptr_t>(p)] = sz;
    return p;
}

void roft_host_free(void* p)
{
    if (!p) return;
    HostPool& hp = host_pool();
    size_t sz = 0;
    {
        std::lock_guard<std::mutex> lk(hp.mu);
        auto it = hp.live.find(reinterpret_cast<uintptr_t>(p));
        if (it == hp.live.end()) return;
        sz = it->second;
        hp.live.erase(it);
        if (hp.spare_bytes + sz <= kHostPoolSpareCap) {
            hp.spare[sz].push_back(p);
            hp.spare_bytes += sz;
            return;
        }
    }
    (void)hipHostFree(p);
}

int roft_host_is_pinned(const void* p)
{
    if (!p) return 0;
    HostPool& hp = host_pool();
    std::lock_guard<std::mutex> lk(hp.mu);
    auto it = hp.live.upper_bound(reinterpret_cast<uintptr_t>(p));
    if (it == hp.live.begin()) return 0;
    --it;
    return reinterpret_cast<uintptr_t>(p) < it->first + it->second ? 1 : 0;
}

int roft_default_config(roft_config* c, int width, int height, int flow_type)
{
    if (!c) return fail(ROFT_ERR_INVALID, "null config");
    std::memset(c, 0, sizeof(*c));
    c->cam.width = width;
    c->cam.height = height;
    if (width == 640) {  // config/config_ho3d.cfg shape (fx..cy are read from cam_K.json there)
        c->cam.fx = c->cam.fy = 614.7142806307731;
        c->cam.cx = 320.0; c->cam.cy = 240.0;
    } else {             // config/config_fast_ycb.cfg:5-10
        c->cam.fx = c->cam.fy = 1229.4285612615463 * width / 1280.0;
        c->cam.cx = width / 2.0; c->cam.cy = height / 2.0;
    }
    c->flow_type = flow_type;
    c->flow_grid = (flow_type == ROFT_FLOW_S16C2) ? 4 : 1;
    c->flow_scale = (flow_type == ROFT_FLOW_S16C2) ? 32.0f : 1.0f;
    c->sample_time = 1.0 / 30.0;
    c->ut.alpha = 1.0; c->ut.beta = 2.0; c->ut.kappa = 0.0;
    c->depth_maximum = 2.0;
    c->subsampling_radius = 35.0;
    c->flow_weighting = 1;
    c->use_pose = c->use_pose_resync = c->use_velocity = 1;
    c->outlier_rejection = 1;
    c->flow_aided_segmentation = 1;
    c->mask_frames_between = 6;
    c->pose_frames_between = 6;
    c->max_objects = 64;
    // (2e-4 / 4e-3 until the end of round 2: 1.6 % of the steps of the 64-object workload then fell back to the
    //  eigen-decomposition -- 35 us instead of 17 -- and since a pose chain launch lasts as long as its slowest object,
    //  they cost 4 % of the throughput; twice the thresholds moves ADD-S against the CPU path from 2.610e-9 to 2.618e-9 mm)
    c->ukf_cholesky_guard = 4e-4;
    c->ukf_cholesky_guard_bilinear = 8e-3;
    c->device = 0;
    c->max_batch_frames = 1;
    return ROFT_OK;
}

int roft_default_object(roft_object_desc* o)
{
    if (!o) return fail(ROFT_ERR_INVALID, "null object");
    std::memset(o, 0, sizeof(*o));
    o->p_mean0[9] = 1.0;
    for (int i = 0; i < 12; ++i) o->p_cov0_diag[i] = 1e-3;
    for (int i = 0; i < 6; ++i) { o->v_cov0_diag[i] = 1e-3; o->v_q_diag[i] = 0.1; }
    for (int i = 0; i < 3; ++i) {
        o->p_sigma_ang_vel[i] = 1.0;  // test/test.sh:70
        o->p_psd_lin_acc[i] = 1.0;
        o->p_meas_cov_v[i] = 0.1;
        o->p_meas_cov_w[i] = 1e-4;
        o->p_meas_cov_x[i] = 1e-3;
        o->p_meas_cov_q[i] = 1e-4;    // test/test.sh:71
    }
    o->v_meas_cov_flow[0] = o->v_meas_cov_flow[1] = 1.0;
    return ROFT_OK;
}

int roft_engine_destroy(roft_engine* e);

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
static std::mutex g_stream_mu;
static std::vector<StreamSet*> g_stream_sets;
constexpr int kMaxStreamSetsPerDevice = 12;   // 60 streams: what a process ever creates per device, however many engines it builds

// Microseconds until a one-workgroup kernel on `b` completes while `a` is placing a grid three times the size of the device
// (one 150 KB-LDS workgroup per CU at a time, 100 us each): the host's launch + wait latency (~15) when the streams have
// hardware queues of their own, >= 100 more when the runtime mapped them onto ONE queue -- b's packet then waits until a's
// dispatch has been placed completely.  a == nullptr: the same without the blocker (the base line of this host, now).
static double probe_pair_us(hipStream_t a, hipStream_t b, int* flag)
{
    if (a) (void)hipStreamSynchronize(a);
    (void)hipStreamSynchronize(b);
    const double t0 = host_now_us();
    if (a) hipLaunchKernelGGL(probe_blocker_kernel, dim3(3 * device_cu_count()), dim3(64), 150 * 1024, a, 10000ll);
    hipLaunchKernelGGL(probe_tiny_kernel, dim3(1), dim3(64), 0, b, flag);
    (void)hipStreamSynchronize(b);
    const double dt = host_now_us() - t0;
    if (a) (void)hipStreamSynchronize(a);
    return dt;
}

// number of pairs among the four busy chains' streams (pose lanes, velocity, mask) that share a hardware queue.  The
// threshold follows the host: the best of three launches of the one-workgroup kernel alone is the base line (a loaded host
// core reads 30 - 40 us where an idle one reads 12), a pair counts as sharing a queue at base line + 60 us (the blocker
// holds a shared queue for >= 100 us) in both of two tries.
static int stream_conflicts(const StreamSet* s, int* flag)
{
    hipStream_t st[4] = {s->pose[0], s->pose[1], s->vel, s->mask};
    double base = 1e30;
    for (int i = 0; i < 3; ++i) base = std::min(base, probe_pair_us(nullptr, st[i], flag));
    const double thr = base + 60.0;
    int n = 0;
    for (int a = 0; a < 4; ++a)
        for (int b = a + 1; b < 4; ++b) {
            double us = probe_pair_us(st[a], st[b], flag);
            if (us >= thr) us = std::min(us, probe_pair_us(st[a], st[b], flag));   // (a slow host call is not a conflict)
            if (us >= thr) ++n;
        }
    return n;
}

static int create_stream_set(int device, bool priorities, StreamSet** out)
{
    StreamSet* s = new StreamSet();
    s->device = device;
    s->priorities = priorities;
    // Priorities (measured): the velocity chain -- short kernels every other chain waits for -- high, the
    // mask chain low, the pose lanes in between.  The lanes spend
    // most of a short run WAITING for events of the velocity chain; with the lanes on the high-priority queues (rounds 1 and
    // 2) the 20-frame run of the driver tracked 10 % slower (8.4e5 against 9.3e5 object-frames/s), longer runs the same.
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (!priorities) greatest = least;
    const int normal = (least + greatest) / 2;
    // (ROFT_PRIO=<lane 0><lane 1><velocity><mask>, each h | n | l: experiments)
    int pr[5] = {normal, normal, greatest, least, normal};
    if (const char* pe = getenv("ROFT_PRIO"))
        for (int i = 0; i < 5 && pe[i]; ++i) pr[i] = pe[i] == 'h' ? greatest : (pe[i] == 'l' ? least : normal);
    hipError_t err = hipSuccess;
    for (int l = 0; l < kNumLin && err == hipSuccess; ++l) err = hipStreamCreateWithPriority(&s->pose[l], hipStreamNonBlocking, pr[l]);
    if (err == hipSuccess) err = hipStreamCreateWithPriority(&s->vel, hipStreamNonBlocking, pr[2]);
    if (err == hipSuccess) err = hipStreamCreateWithPriority(&s->mask, hipStreamNonBlocking, pr[3]);
    if (err == hipSuccess) err = hipStreamCreateWithPriority(&s->up, hipStreamNonBlocking, pr[4]);
    if (err != hipSuccess) { delete s; return fail(ROFT_ERR_DEVICE, std::string("stream creation: ") + hipGetErrorString(err)); }
    *out = s;
    return ROFT_OK;
}

// The runtime maps HIP streams round robin onto a few hardware queues (GPU_MAX_HW_QUEUES, four by default) in the order
// in which the PROCESS creates them, whatever their priority: which of the engine's streams end up sharing a queue depends
// on how many streams the application (or an earlier engine) created before.  Two busy chains on one queue serialise at the
// dispatch level -- a launch that waits for CUs holds up the other chain's packets behind it: measured 6.9e5 instead of 8.5e5
// object-frames/s with three streams created ahead of the engine's, 7.7e5 for the third engine of a process.  So a new set is
// probed (stream_conflicts, ~3 ms), and while two of its busy streams share a queue the set is parked -- its streams stay
// alive and shift the round robin -- and another one is created (at most four times; the least bad one is used then).
// The probe measures time on a device it assumes idle: it is skipped while another engine of this process has work in flight
// on the device (that engine would read as conflicts on every pair, and would see the probe's whole-device blocker grids in
// its pipeline) and with ROFT_NO_STREAM_PROBE=1.  A process never holds more than kMaxStreamSetsPerDevice sets per device: beyond
// that, free sets -- parked ones included, least conflicts first -- are handed out again.
static int acquire_streams(int device, bool priorities, StreamSet** out)
{
    std::lock_guard<std::mutex> lk(g_stream_mu);
    int n_dev = 0;
    bool other_busy = false;   // another engine of this process has work in flight on the device
    for (StreamSet* s : g_stream_sets) {
        if (s->device != device) continue;
        ++n_dev;
        if (!s->in_use) continue;
        for (hipStream_t q : {s->pose[0], s->pose[1], s->vel, s->mask, s->up})
            if (hipStreamQuery(q) != hipSuccess) other_busy = true;
        (void)hipGetLastError();   // (hipErrorNotReady is an answer, not an error)
    }
    for (StreamSet* s : g_stream_sets)
        if (!s->in_use && !s->parked && s->device == device && s->priorities == priorities) { s->in_use = true; *out = s; return ROFT_OK; }
    if (n_dev >= kMaxStreamSetsPerDevice) {
        StreamSet* best = nullptr;
        for (StreamSet* s : g_stream_sets)
            if (!s->in_use && s->device == device && s->priorities == priorities && (!best || s->conflicts < best->conflicts)) best = s;
        if (best) { best->in_use = true; *out = best; return ROFT_OK; }
    }
    const char* np = getenv("ROFT_NO_STREAM_PROBE");
    const bool probe = !(np && np[0] == '1') && !other_busy && n_dev < kMaxStreamSetsPerDevice;
    if (!probe) {
        StreamSet* s = nullptr;
        if (int rc = create_stream_set(device, priorities, &s)) return rc;
        s->conflicts = -1;
        s->in_use = true;
        g_stream_sets.push_back(s);
        *out = s;
        return ROFT_OK;
    }
    DevBuf<int> flag;
    HIP_TRY(flag.ensure(1));
    HIP_TRY(set_max_dynamic_lds(reinterpret_cast<const void*>(probe_blocker_kernel), 150 * 1024));
    StreamSet* best = nullptr;
    int best_conflicts = 1 << 30;
    for (int attempt = 0; attempt < 4 && n_dev < kMaxStreamSetsPerDevice; ++attempt, ++n_dev) {
        StreamSet* s = nullptr;
        if (int rc = create_stream_set(device, priorities, &s)) return rc;
        const int c = stream_conflicts(s, flag.p);
        s->conflicts = c;
        if (getenv("ROFT_STREAM_DEBUG")) std::fprintf(stderr, "[roft streams] device %d attempt %d: %d conflicting pairs\n", device, attempt, c);
        if (c < best_conflicts) { best = s; best_conflicts = c; }
        s->parked = true;               // unless chosen below
        g_stream_sets.push_back(s);
        if (c == 0) break;
    }
    (void)hipGetLastError();
    best->parked = false;
    best->in_use = true;
    *out = best;
    return ROFT_OK;
}

static void release_streams(StreamSet* s)
{
    std::lock_guard<std::mutex> lk(g_stream_mu);
    if (s) s->in_use = false;
}

// true when no other engine of this process holds a stream set on the device of `mine` -- a fact of which engines EXIST at the
// submit (round 6, ADVICE r05: rounds 4 - 5 asked the other engines' streams whether they were busy at that instant, a host and
// device timing that another engine could falsify a microsecond later; the launch graph is a function of the batch index, the
// object count and this count only)
static bool alone_on_device(const StreamSet* mine)
{
    std::lock_guard<std::mutex> lk(g_stream_mu);
    for (const StreamSet* s : g_stream_sets)
        if (s != mine && s->in_use && s->device == mine->device) return false;
    return true;
}

static int engine_setup(roft_engine* e, const roft_config* cfg)
{
    constexpr int R = roft_engine::kBatchRing;
    // The image chains of batch b+1 do not depend on the pose chain of batch b (only the other way round, through
    // the twist ring and the mask planes), so the chains run on separate HIP streams, ordered by one event per batch
    // and edge.  ROFT_ONE_STREAM=1 serialises everything on one stream (debugging).
    const char* one = getenv("ROFT_ONE_STREAM");
    e->multi = !(one && one[0] == '1');
    const char* np = getenv("ROFT_NO_STREAM_PRIORITY");
    if (int rc = acquire_streams(cfg->device, !(np && np[0] == '1'), &e->streams)) return rc;
    if (e->multi) {
        e->stream = e->streams->mask;
        e->vel_stream = e->streams->vel;
        for (int l = 0; l < kNumLin; ++l) e->pose_stream[l] = e->streams->pose[l];
        e->up_stream = e->streams->up;
    } else {
        e->stream = e->pose_stream[0] = e->pose_stream[1] = e->vel_stream = e->up_stream = e->streams->mask;
    }
    for (int i = 0; i < R; ++i) {
        HIP_TRY(e->dctrl[i].ensure((size_t)cfg->max_objects * e->T_max, true));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->stage[i]), sizeof(FrameCtrl) * cfg->max_objects * e->T_max));
        for (hipEvent_t* ev : {&e->ev_up[i], &e->ev_ctrl[i], &e->ev_mask[i], &e->ev_part[i], &e->ev_prep[i], &e->ev_feat[i], &e->ev_vel[i], &e->ev_done[i][0], &e->ev_done[i][1]})
            HIP_TRY(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    }
    DevFlowFmt ff;
    ff.type = cfg->flow_type;
    ff.grid = cfg->flow_grid;
    ff.cols = cfg->cam.width / cfg->flow_grid;
    ff.rows = cfg->cam.height / cfg->flow_grid;
    ff.scale = cfg->flow_scale;
    const int radius = (int)(size_t)cfg->subsampling_radius;
    if (int rc = e->arr.alloc(cfg->max_objects, e->T_max, make_cam(cfg->cam), ff, radius)) return rc;
    e->arr.a.n_obj = 0;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->dev_error), sizeof(int), hipHostMallocMapped));
    *e->dev_error = 0;
    {
        void* dp = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&dp, e->dev_error, 0));
        e->arr.a.dev_error = static_cast<int*>(dp);
    }
    e->arr.a.mask_wgs = cfg->mask_workgroups_per_object;
    e->arr.a.outlier_parts = cfg->outlier_bands_per_alternative;
    e->arr.a.ukf_chol_guard = (cfg->ukf_cholesky_guard > 0.0) ? cfg->ukf_cholesky_guard : 0.0;
    e->arr.a.ukf_chol_guard_bil = (cfg->ukf_cholesky_guard_bilinear > 0.0) ? cfg->ukf_cholesky_guard_bilinear : 0.0;
    e->h_params.resize(cfg->max_objects);
    e->staging.resize(e->retain);
    {
        int can = 0;
        if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, cfg->device) != hipSuccess) { (void)hipGetLastError(); can = 0; }
        e->wait_value_ok = can != 0;   // (without it the lanes run behind the velocity chain's event)
    }
    // Nothing of a batch may happen for the first time inside a caller's timed region: every event of the batch ring has
    // completed one dispatch on the stream that will carry it (the first use of an event as a kernel's stop event takes a signal
    // from the runtime's pool -- a host call of its own kind), and every stream has queued a wait on an event and on a value.
    if (e->multi) {
        int* flag = reinterpret_cast<int*>(e->arr.skf_started.p);   // (a word that stays 0 until the first batch; the probe kernel writes 1 ... reset below)
        for (int i = 0; i < R; ++i) {
            hipExtLaunchKernelGGL(probe_tiny_kernel, dim3(1), dim3(64), 0, e->stream, nullptr, e->ev_ctrl[i], 0, flag);
            hipExtLaunchKernelGGL(probe_tiny_kernel, dim3(1), dim3(64), 0, e->stream, nullptr, e->ev_mask[i], 0, flag);
            hipExtLaunchKernelGGL(probe_tiny_kernel, dim3(1), dim3(64), 0, e->stream, nullptr, e->ev_part[i], 0, flag);
            HIP_TRY(hipStreamWaitEvent(e->vel_stream, e->ev_part[i], 0));
            hipExtLaunchKernelGGL(probe_tiny_kernel, dim3(1), dim3(64), 0, e->stream, nullptr, e->ev_feat[i], 0, flag);
            HIP_TRY(hipStreamWaitEvent(e->vel_stream, e->ev_mask[i], 0));
            hipExtLaunchKernelGGL(probe_tiny_kernel, dim3(1), dim3(64), 0, e->vel_stream, nullptr, e->ev_vel[i], 0, flag);
            for (int l = 0; l < kNumLin; ++l) {
                HIP_TRY(hipStreamWaitEvent(e->pose_stream[l], (i & 1) ? e->ev_vel[i] : e->ev_ctrl[i], 0));
                hipExtLaunchKernelGGL(probe_tiny_kernel, dim3(1), dim3(64), 0, e->pose_stream[l], nullptr, e->ev_done[i][l], 0, flag);
            }
            HIP_TRY(hipEventRecord(e->ev_up[i], e->up_stream));
            HIP_TRY(hipStreamWaitEvent(e->up_stream, e->ev_mask[i], 0));
            hipExtLaunchKernelGGL(probe_tiny_kernel, dim3(1), dim3(64), 0, e->up_stream, nullptr, e->ev_prep[i], 0, flag);
            HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_prep[i], 0));
        }
        HIP_TRY(hipGetLastError());
        for (hipStream_t q : {e->stream, e->vel_stream, e->pose_stream[0], e->pose_stream[1], e->up_stream}) HIP_TRY(hipStreamSynchronize(q));
        HIP_TRY(hipMemset(e->arr.skf_started.p, 0, sizeof(unsigned long long)));
        if (e->wait_value_ok)
            for (int l = 0; l < kNumLin; ++l)
                HIP_TRY(hipStreamWaitValue64(e->pose_stream[l], e->arr.skf_started.p, 0ull, hipStreamWaitValueGte, ~0ull));   // (satisfied at once)
        for (int l = 0; l < kNumLin; ++l) HIP_TRY(hipStreamSynchronize(e->pose_stream[l]));
    }
    if (const char* hm = getenv("ROFT_HANDOFF")) e->handoff_mode = atoi(hm);
    if (const char* pm = getenv("ROFT_PREP_AHEAD")) e->prep_mode = atoi(pm);
    if (const char* pm = getenv("ROFT_MASK_PART_GATE")) e->part_mode = atoi(pm);
    // A tool that lets only ONE kernel run at a time (rocprofv3 --pmc: counter collection serialises the dispatches) cannot run a
    // lane next to the velocity filter it waits for -- the runtime's stream-wait itself is a kernel that spins: off under it.
    else if (getenv("ROCPROF_COUNTER_COLLECTION")) e->handoff_mode = 0;
    e->feat_batch.assign((size_t)cfg->max_objects * kFeatRing, -1);
    const char* hpf = getenv("ROFT_HOST_PROF");
    e->host_prof = hpf && hpf[0] == '1';
    return ROFT_OK;
}

int roft_engine_create(const roft_config* cfg, roft_engine** out)
{
    if (!cfg || !out) return fail(ROFT_ERR_INVALID, "null argument");
    if (roft_device_count() <= cfg->device) return fail(ROFT_ERR_DEVICE, "no such HIP device (the engine has no CPU path)");
    if (int rc = check_geometry(cfg->cam.width, cfg->cam.height)) return rc;
    if (cfg->max_objects <= 0) return fail(ROFT_ERR_INVALID, "max_objects must be positive");
    if (cfg->flow_type != ROFT_FLOW_S16C2 && cfg->flow_type != ROFT_FLOW_F32C2)
        return fail(ROFT_ERR_INVALID, "flow_type must be ROFT_FLOW_S16C2 or ROFT_FLOW_F32C2");
    if (cfg->flow_grid <= 0 || cfg->cam.width % cfg->flow_grid) return fail(ROFT_ERR_INVALID, "bad flow grid");
    if (cfg->mask_frames_between > kMaxFlowHist)
        return fail(ROFT_ERR_INVALID, "mask_frames_between > 30 (ROFT_MAX_FLOW_CHASE) is not supported");
    if (cfg->max_batch_frames < 0 || cfg->max_batch_frames > kMaxBatch)
        return fail(ROFT_ERR_INVALID, "max_batch_frames must be 0 .. ROFT_MAX_BATCH_FRAMES");
    if (cfg->mask_workgroups_per_object < 0 || cfg->mask_workgroups_per_object > 8)
        return fail(ROFT_ERR_INVALID, "mask_workgroups_per_object must be 0 (automatic) .. 8");
    if (cfg->outlier_bands_per_alternative < 0 || cfg->outlier_bands_per_alternative > kMaxOutlierParts)
        return fail(ROFT_ERR_INVALID, "outlier_bands_per_alternative must be 0 (automatic) .. 8");
    if ((int)(size_t)cfg->subsampling_radius <= 0) return fail(ROFT_ERR_INVALID, "subsampling_radius must be >= 1");
    const int T = std::max(cfg->max_batch_frames, 1);
    // batches in flight: enough that the host never runs out of enqueued work while it waits for the oldest one -- a
    // batch takes about three batch periods from its mask chain to the end of its pose chain
    const int lead = (T == 1) ? 6 : 5;
    // the re-sync replays at most pose_frames_between + 1 buffered velocities (all of them when that number is unknown)
    if (cfg->pose_frames_between + 2 > kMaxSteps)
        return fail(ROFT_ERR_INVALID, "pose_frames_between too large (the re-sync replays pose_frames_between + 1 steps, at most 9)");
    static_assert(roft_engine::kBatchRing > 6, "batch ring");
    static_assert(5 * kMaxBatch + kMaxBatch + 1 < kPlaneSlots && 6 + 1 + 1 < kPlaneSlots, "plane ring");
    static_assert(5 * kMaxBatch + kMaxSteps + 2 < kTwistRing, "twist ring");
    static_assert(kFeatRing >= kMaxBatch + 2, "feature ring");
    HIP_TRY(hipSetDevice(cfg->device));
    roft_engine* e = new roft_engine();
    e->cfg = *cfg;
    e->cfg.max_batch_frames = T;
    e->T_max = T;
    e->lead = lead;
    // flows kept per object: what one mask can be chased through
    if (cfg->stamped_masks) e->hist_cap = (cfg->mask_frames_between > 0) ? std::min(cfg->mask_frames_between, 29) : 29;
    else e->hist_cap = (cfg->mask_frames_between > 0) ? cfg->mask_frames_between : kMaxFlowHist;
    e->retain = e->hist_cap + lead * T + 2;
    if (const int rc = engine_setup(e, cfg)) {
        const std::string msg = g_last_error;
        (void)roft_engine_destroy(e);
        return fail(rc, msg);
    }
    *out = e;
    return ROFT_OK;
}

int roft_engine_destroy(roft_engine* e)
{
    if (!e) return ROFT_OK;
    constexpr int R = roft_engine::kBatchRing;
    (void)hipSetDevice(e->cfg.device);
    if (e->host_prof && e->hp_batches > 0) {
        static const char* names[7] = {"submit: wait in-flight bound", "submit: frame programs + uploads", "submit: wait uploads",
                                       "step: FrameCtrl upload", "step: mask chain", "step: velocity chain", "step: pose chain"};
        for (int i = 0; i < 7; ++i) std::fprintf(stderr, "[roft host] %-36s %7.2f us/batch\n", names[i], e->hp_acc[i] / e->hp_batches);
    }
    for (hipStream_t s : {e->stream, e->vel_stream, e->pose_stream[0], e->pose_stream[1], e->up_stream})
        if (s) (void)hipStreamSynchronize(s);
    for (int i = 0; i < R; ++i) {
        for (hipEvent_t ev : {e->ev_up[i], e->ev_ctrl[i], e->ev_mask[i], e->ev_part[i], e->ev_prep[i], e->ev_feat[i], e->ev_vel[i], e->ev_done[i][0], e->ev_done[i][1]})
            if (ev) (void)hipEventDestroy(ev);
        if (e->stage[i]) (void)hipHostFree(e->stage[i]);
    }
    release_streams(e->streams);   // (idle: synchronised above)
    for (auto* o : e->objs) delete o;
    if (e->state_host) (void)hipHostFree(e->state_host);
    if (e->dev_error) (void)hipHostFree(e->dev_error);
    for (auto ev : e->tev) (void)hipEventDestroy(ev);
    delete e;
    return ROFT_OK;
}

int roft_engine_retain_frames(const roft_engine* e) { return e ? e->retain : ROFT_RETAIN_FRAMES; }

int roft_engine_get_stats(roft_engine* e, roft_engine_stats* out)
{
    if (!e || !out) return fail(ROFT_ERR_INVALID, "null argument");
    *out = e->stats;
    return ROFT_OK;
}

int roft_object_add(roft_engine* e, const roft_object_desc* d, int* obj_id)
{
    if (!e || !d) return fail(ROFT_ERR_INVALID, "null argument");
    if ((int)e->objs.size() >= e->cfg.max_objects) return fail(ROFT_ERR_CAPACITY, "max_objects reached");
    if (e->frame_counter > 0 || e->submitted) return fail(ROFT_ERR_STATE, "objects must be added before the first frame");
    HIP_TRY(hipSetDevice(e->cfg.device));
    const int id = (int)e->objs.size();
    HostObject* o = new HostObject();
    ObjParams& p = e->h_params[id];
    std::memset(&p, 0, sizeof(p));
    for (int i = 0; i < 3; ++i) {
        p.sigma_ang_vel[i] = d->p_sigma_ang_vel[i];
        p.psd_lin_acc[i] = d->p_psd_lin_acc[i];
        p.R_v[i] = d->p_meas_cov_v[i]; p.R_w[i] = d->p_meas_cov_w[i];
        p.R_x[i] = d->p_meas_cov_x[i]; p.R_q[i] = d->p_meas_cov_q[i];
    }
    for (int i = 0; i < 6; ++i) p.v_q[i] = d->v_q_diag[i];
    p.r_flow[0] = d->v_meas_cov_flow[0];
    p.r_flow[1] = d->v_meas_cov_flow[1];
    auto bail = [&](int code, const std::string& msg) { delete o; return fail(code, msg); };
    if (d->mesh.n_verts > 0 && d->mesh.n_tris > 0) {
        // closed orientable surface?  Then the outlier test's render leaves the triangles that face away out (the render
        // contract, oracle/ro_render.c) and walks the triangles in an order that keeps alike-facing ones together
        PreparedMesh pm;
        prepare_mesh(d->mesh.verts, d->mesh.n_verts, d->mesh.tris, d->mesh.n_tris, pm);
        hipError_t err = o->verts.ensure((size_t)3 * d->mesh.n_verts);
        if (err == hipSuccess) err = o->tris.ensure((size_t)3 * d->mesh.n_tris);
        if (err == hipSuccess && pm.closed) err = o->tri_flip.ensure((size_t)d->mesh.n_tris);
        if (err == hipSuccess) err = hipMemcpy(o->verts.p, d->mesh.verts, sizeof(float) * 3 * d->mesh.n_verts, hipMemcpyHostToDevice);
        if (err == hipSuccess) err = hipMemcpy(o->tris.p, pm.tris(d->mesh.tris), sizeof(int32_t) * 3 * d->mesh.n_tris, hipMemcpyHostToDevice);
        if (err == hipSuccess && pm.closed) err = hipMemcpy(o->tri_flip.p, pm.flip.data(), (size_t)d->mesh.n_tris, hipMemcpyHostToDevice);
        if (err != hipSuccess) return bail(ROFT_ERR_DEVICE, std::string("mesh upload: ") + hipGetErrorString(err));
        p.verts = o->verts.p; p.tris = o->tris.p;
        p.tri_flip = pm.closed ? o->tri_flip.p : nullptr;
        p.n_verts = d->mesh.n_verts; p.n_tris = d->mesh.n_tris;
    } else if (e->cfg.outlier_rejection && e->cfg.use_pose) {
        return bail(ROFT_ERR_INVALID, "outlier rejection needs a mesh");
    }
    // initialization_step (ROFTFilter.cpp:216-237)
    ObjState* st = new ObjState();
    init_state(*st);
    for (int i = 0; i < 6; ++i) { st->v_mean[i] = d->v_mean0[i]; st->v_cov[i * 6 + i] = d->v_cov0_diag[i]; }
    PoseBelief b{};
    for (int i = 0; i < 13; ++i) b.mean[i] = d->p_mean0[i];
    for (int i = 0; i < 12; ++i) b.cov[i * 12 + i] = d->p_cov0_diag[i];
    st->belief[B_LIN0] = b;       // p_corr_belief_
    st->belief[B_LIN1] = b;       // buffered_belief_ = p_corr_belief_ at initialization (ROFTFilter.cpp:231)
    st->belief[B_PRED] = st->belief[B_PRED + 1] = b;
    hipError_t err = hipMemcpy(e->arr.params.p + id, &p, sizeof(p), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(e->arr.state.p + id, st, sizeof(ObjState), hipMemcpyHostToDevice);
    delete st;
    if (err != hipSuccess) return bail(ROFT_ERR_DEVICE, std::string("state upload: ") + hipGetErrorString(err));
    e->arr.a.max_tris = std::max(e->arr.a.max_tris, d->mesh.n_tris);
    e->arr.a.max_verts = std::max(e->arr.a.max_verts, d->mesh.n_verts);
    e->objs.push_back(o);
    e->arr.a.n_obj = (int)e->objs.size();
    if (obj_id) *obj_id = id;
    return ROFT_OK;
}

// The UKF steps of one frame (ROFTFilter.cpp:327-367 over CartesianQuaternionMeasurement::freeze, cpp:92-348).
// Returns false when the frame needs more than kMaxSteps steps.
static bool build_pose_program(const roft_config& cfg, Sched& o, const roft_frame_input& in, FrameCtrl& c)
{
    const int slot = o.frame_idx % kTwistRing;
    c.twist_slot = slot;
    int n = 0;
    bool overflow = false;
    auto add = [&](StepDesc sd) { if (n < kMaxSteps) c.steps[n++] = sd; else overflow = true; };
    auto vel_pop_front = [&]() { std::memmove(o.vel_buf, o.vel_buf + 1, sizeof(int) * (size_t)(--o.n_vel)); };

    // CartesianQuaternionMeasurement::freeze(Standard)  (cpp:176-347)
    const bool has_vel = cfg.use_velocity != 0;
    const bool is_pose = cfg.use_pose && in.pose_valid;
    int type = ROFT_MEAS_NONE;
    if (has_vel && is_pose) type = ROFT_MEAS_POSE_VELOCITY;
    else if (has_vel) type = ROFT_MEAS_VELOCITY;
    else if (is_pose) type = ROFT_MEAS_POSE;
    if (has_vel) {
        // (only the last pose_frames_between + 1 entries are ever replayed; the ring bounds the rest)
        if (o.n_vel == kTwistRing) vel_pop_front();
        o.vel_buf[o.n_vel++] = slot;
        while (o.n_vel > kMaxSteps + 2) vel_pop_front();
        o.last_meas_slot = slot;
    }
    for (int i = 0; i < 3; ++i) c.pose_x[i] = in.pose_x[i];
    for (int i = 0; i < 4; ++i) c.pose_q[i] = in.pose_q[i];

    if (type == ROFT_MEAS_POSE_VELOCITY && cfg.use_pose_resync) {
        // ROFTFilter.cpp:333-340: buffered_belief_ <- p_corr_belief_, p_corr_belief_ <- the old buffered_belief_.
        // The two Gaussians swap roles; nothing is copied (see BeliefSlot in roft_device.h).
        o.cur_slot ^= 1;
    }
    const int cur = B_LIN0 + o.cur_slot;
    const int lin = o.own[o.cur_slot];
    c.lane = lin;
    c.cur_slot = cur;
    StepDesc sd{};
    sd.op = 1;
    sd.src = cur;
    sd.do_predict = 1;
    sd.twist_slot = slot;
    if (type == ROFT_MEAS_POSE_VELOCITY) {
        if (cfg.use_pose_resync) {
            // ROFTFilter.cpp:331-354: continue from the belief buffered at the previous pose arrival and
            // replay the buffered velocities (PopBufferedMeasurement, cpp:97-154)
            bool pose_pending = true;
            for (;;) {
                if (cfg.pose_frames_between > 0)
                    while (o.n_vel > cfg.pose_frames_between + 1) vel_pop_front();
                if (o.n_vel == 0) { o.vel_buf[o.n_vel++] = o.last_meas_slot; break; }
                const int ts = o.vel_buf[0];
                vel_pop_front();
                o.last_meas_slot = ts;
                StepDesc r{};
                r.op = 1;
                r.do_predict = 1;
                r.twist_slot = ts;
                r.src = cur;
                if (pose_pending) {
                    pose_pending = false;
                    if (cfg.outlier_rejection) {
                        r.n_corr = 2;
                        r.type[0] = ROFT_MEAS_POSE_VELOCITY; r.dst[0] = b_alt(lin, 0);
                        r.type[1] = ROFT_MEAS_VELOCITY;      r.dst[1] = b_alt(lin, 1);
                        c.outlier_step = n;
                    } else {
                        r.n_corr = 1;
                        r.type[0] = ROFT_MEAS_POSE_VELOCITY; r.dst[0] = cur;
                    }
                } else {
                    r.n_corr = 1;
                    r.type[0] = ROFT_MEAS_VELOCITY; r.dst[0] = cur;
                }
                add(r);
            }
            // the test reads the features buffered at the previous pose arrival; this frame's are buffered for
            // the next one (ROFTFilter.cpp:353)
            c.feat_read = o.feat_slot;
            if (c.feat_write < 0) { c.feat_write = o.feat_next; o.feat_next = (o.feat_next + 1) % kFeatRing; }
            o.feat_slot = c.feat_write;
        } else {
            if (cfg.outlier_rejection) {
                sd.n_corr = 2;
                sd.type[0] = ROFT_MEAS_POSE_VELOCITY; sd.dst[0] = b_alt(lin, 0);
                sd.type[1] = ROFT_MEAS_VELOCITY;      sd.dst[1] = b_alt(lin, 1);
                c.outlier_step = n;
                // without re-sync the test uses the current frame's depth and mask
                if (c.feat_write < 0) { c.feat_write = o.feat_next; o.feat_next = (o.feat_next + 1) % kFeatRing; }
                c.feat_read = c.feat_write;
                o.feat_slot = c.feat_write;
            } else {
                sd.n_corr = 1;
                sd.type[0] = ROFT_MEAS_POSE_VELOCITY; sd.dst[0] = cur;
            }
            add(sd);
        }
    } else if (type != ROFT_MEAS_NONE) {
        sd.n_corr = 1;
        sd.type[0] = type; sd.dst[0] = cur;
        add(sd);
    } else {
        sd.n_corr = 0;  // p_corr = p_pred (ROFTFilter.cpp:366-367)
        sd.dst[0] = cur;
        add(sd);
    }
    c.n_steps = n;
    return !overflow;
}

// `bytes` of the staging memory that is recycled with `frame`'s slot (bump allocation in 32 MB chunks)
static int stage_alloc(roft_engine* e, int frame, size_t bytes, unsigned char** out)
{
    StageFrame& sf = e->staging[frame % e->retain];
    const size_t need = (bytes + 255) & ~(size_t)255;
    while (sf.cur < sf.chunks.size() && sf.used + need > sf.chunks[sf.cur]->n) { ++sf.cur; sf.used = 0; }
    if (sf.cur == sf.chunks.size()) {
        auto* c = new DevBuf<unsigned char>();
        const hipError_t err = c->ensure(std::max(need, kStageChunk));
        if (err != hipSuccess) { delete c; return fail(ROFT_ERR_DEVICE, std::string("HOST staging memory: ") + hipGetErrorString(err)); }
        sf.chunks.push_back(c);
        sf.used = 0;
    }
    *out = sf.chunks[sf.cur]->p + sf.used;
    sf.used += need;
    return ROFT_OK;
}

// device copy of one HOST image of `frame` (uploads once per distinct host pointer and frame)
static int stage_host(roft_engine* e, int frame, const void* host, size_t bytes, const void** dev)
{
    StageFrame& sf = e->staging[frame % e->retain];
    for (auto& pr : sf.seen)
        if (pr.first == host) { *dev = pr.second; return ROFT_OK; }
    unsigned char* d = nullptr;
    if (int rc = stage_alloc(e, frame, bytes, &d)) return rc;
    HIP_TRY(hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, e->up_stream));
    e->stats.h2d_bytes += (long long)bytes;
    e->stats.h2d_copies++;
    e->had_uploads = true;
    sf.seen.emplace_back(host, d);
    *dev = d;
    return ROFT_OK;
}

// HOST images of consecutive frames of a batch that are CONSECUTIVE IN HOST MEMORY (a recorded sequence held as one
// [frames, H, W] array: frame t + 1 starts where frame t ends) are uploaded with ONE copy per run instead of one per frame --
// a 1.2 MB copy does not reach the link's rate, a batch's worth does (round 6: the shared-scene leg of bench.py moved 26 GB/s
// in per-frame copies against 43 GB/s in the per-object leg, whose 128 copies per frame keep the link busy by their number).
// The run lives in the staging slot of its LAST frame (recycled after every earlier one); each frame's slot learns where its
// image is, so that stage_host below finds it -- for every object that shows the same host pointer, too.
static int stage_host_runs(roft_engine* e, const roft_frame_input* inputs, int n_obj, int T, size_t depth_bytes, size_t flow_bytes_)
{
    if (T < 2) return ROFT_OK;
    const int frame0 = e->frame_counter;
    for (int kind = 0; kind < 2; ++kind) {
        const size_t bytes = kind == 0 ? depth_bytes : flow_bytes_;
        if (bytes == 0 || (bytes & 255)) continue;   // (the pieces of a run must keep the alignment a single image gets)
        for (int id = 0; id < n_obj; ++id) {
            auto ptr = [&](int t) -> const unsigned char* {
                const roft_frame_input& in = inputs[(size_t)t * n_obj + id];
                if (in.mem_kind != ROFT_MEM_HOST) return nullptr;
                return static_cast<const unsigned char*>(kind == 0 ? static_cast<const void*>(in.depth) : in.flow);
            };
            int t0 = 0;
            while (t0 < T) {
                int t1 = t0;
                const unsigned char* p0 = ptr(t0);
                if (p0)
                    while (t1 + 1 < T && ptr(t1 + 1) == p0 + (size_t)(t1 + 1 - t0) * bytes) ++t1;
                if (p0 && t1 > t0) {
                    bool known = false;   // (a shared scene: an object before this one brought the run)
                    for (auto& pr : e->staging[(frame0 + t0) % e->retain].seen)
                        if (pr.first == p0) { known = true; break; }
                    if (!known) {
                        const int len = t1 - t0 + 1;
                        unsigned char* d = nullptr;
                        if (int rc = stage_alloc(e, frame0 + t1, (size_t)len * bytes, &d)) return rc;
                        HIP_TRY(hipMemcpyAsync(d, p0, (size_t)len * bytes, hipMemcpyHostToDevice, e->up_stream));
                        e->stats.h2d_bytes += (long long)((size_t)len * bytes);
                        e->stats.h2d_copies++;
                        e->had_uploads = true;
                        for (int t = t0; t <= t1; ++t)
                            e->staging[(frame0 + t) % e->retain].seen.emplace_back(p0 + (size_t)(t - t0) * bytes, d + (size_t)(t - t0) * bytes);
                    }
                }
                t0 = t1 + 1;
            }
        }
    }
    return ROFT_OK;
}

static int submit_frames(roft_engine* e, const roft_frame_input* inputs, int n_obj, int T)
{
    const roft_config& cfg = e->cfg;
    const size_t npix = (size_t)cfg.cam.width * cfg.cam.height;
    const size_t fbytes = flow_bytes(e->arr.a.ffmt);
    const int b = e->batch_counter;
    FrameCtrl* blk = e->stage[b % roft_engine::kBatchRing];
    int max_outliers[kNumLin] = {0, 0};
    std::vector<int> n_outliers((size_t)n_obj * kNumLin, 0);
    e->lin_any[0] = e->lin_any[1] = false;
    e->lane_objs[0] = e->lane_objs[1] = e->lane_old_first[0] = e->lane_old_first[1] = 0;
    std::vector<unsigned char> lane_seen((size_t)n_obj * kNumLin, 0);
    {
        // Balance of the two pose chain lanes.  A lane's launch lasts as long as its busiest object, so the lanes only
        // overlap if, in every batch, the re-sync replays of all objects are on ONE lane and the ordinary steps in
        // front of them on the other.  An object that missed a pose (or received an extra one) has its lineages on the
        // opposite lanes from then on: hand its two slots over to the other lanes at the batch boundary.  The new lane
        // of a slot must run behind the last batch in which the old lane touched it (relabel_wait; in the steady state
        // that launch was a short one of the previous batch and has long ended).
        int cnt[kNumLin] = {0, 0};
        for (int id = 0; id < n_obj; ++id) cnt[e->objs[id]->s.own[e->objs[id]->s.cur_slot]]++;
        const int c = cnt[1] > cnt[0] ? 1 : 0;
        e->relabel_wait[0] = e->relabel_wait[1] = -1;
        for (int id = 0; id < n_obj; ++id) {
            Sched& o = e->objs[id]->s;
            if (o.own[o.cur_slot] == c) continue;
            e->relabel_wait[c] = std::max(e->relabel_wait[c], o.last_touch[o.cur_slot]);
            e->relabel_wait[1 - c] = std::max(e->relabel_wait[1 - c], o.last_touch[1 - o.cur_slot]);
            std::swap(o.own[0], o.own[1]);
        }
    }

    for (int t = 0; t < T; ++t) {   // the staging slots of the batch's frames are free again: every frame that could read them has ended (in-flight bound)
        StageFrame& sf = e->staging[(e->frame_counter + t) % e->retain];
        sf.cur = 0;
        sf.used = 0;
        sf.seen.clear();
    }
    if (int rc = stage_host_runs(e, inputs, n_obj, T, npix * sizeof(float), fbytes)) return rc;
    for (int t = 0; t < T; ++t) {
        const int frame = e->frame_counter + t;
        for (int id = 0; id < n_obj; ++id) {
            HostObject& ho = *e->objs[id];
            Sched& o = ho.s;
            const roft_frame_input& in = inputs[(size_t)t * n_obj + id];
            FrameCtrl& c = blk[(size_t)t * n_obj + id];
            clear_ctrl(c);
            if (!in.depth) return fail(ROFT_ERR_INVALID, "cannot continue without a continuous depth stream (ROFTFilter.cpp:261-266)");
            c.dt = (in.dt > 0.0) ? in.dt : cfg.sample_time;

            // ---- inputs to device memory
            const float* d_depth;
            const void* d_flow = nullptr;
            const uint8_t* d_mask = nullptr;
            if (in.mem_kind == ROFT_MEM_DEVICE) {
                d_depth = in.depth;
                d_flow = in.flow;
                d_mask = in.mask;
                // the first call of an engine only: a host pointer declared as device memory is a GPU page fault that takes
                // the process down at the first kernel -- the commonest mistake of a new binding is refused here instead
                if (!e->device_pointers_checked) {
                    const void* ptrs[3] = {in.depth, in.flow, in.mask};
                    static const char* const what[3] = {"depth", "flow", "mask"};
                    for (int q = 0; q < 3; ++q) {
                        if (!ptrs[q]) continue;
                        // device or managed memory, or host memory the GPU can address as it is (hipHostMalloc / hipHostRegister:
                        // pinned and mapped -- zero-copy over the bus); unregistered pageable memory is what is refused
                        hipPointerAttribute_t attr{};
                        const hipError_t pe = hipPointerGetAttributes(&attr, ptrs[q]);
                        if (pe != hipSuccess) (void)hipGetLastError();
                        bool usable = pe == hipSuccess && (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged);
                        if (!usable && pe == hipSuccess && attr.type == hipMemoryTypeHost) {
                            void* dp = nullptr;
                            usable = hipHostGetDevicePointer(&dp, const_cast<void*>(ptrs[q]), 0) == hipSuccess && dp == ptrs[q];
                            if (!usable) (void)hipGetLastError();
                        }
                        if (!usable)
                            return fail(ROFT_ERR_INVALID, std::string("mem_kind is ROFT_MEM_DEVICE but the ") + what[q] + " pointer of object " +
                                                              std::to_string(id) + " is neither device memory nor pinned, mapped host memory "
                                                              "(pass ROFT_MEM_HOST for ordinary host buffers)");
                    }
                }
                if ((reinterpret_cast<uintptr_t>(d_mask) & 15) || (reinterpret_cast<uintptr_t>(d_flow) & 7) ||
                    (reinterpret_cast<uintptr_t>(d_depth) & 3))
                    return fail(ROFT_ERR_INVALID, "device buffers must be aligned: mask 16 B, flow 8 B, depth 4 B");
            } else if (in.mem_kind == ROFT_MEM_HOST) {
                const void* p = nullptr;
                if (int rc = stage_host(e, frame, in.depth, npix * sizeof(float), &p)) return rc;
                d_depth = static_cast<const float*>(p);
                if (in.flow) { if (int rc = stage_host(e, frame, in.flow, fbytes, &d_flow)) return rc; }
                if (in.mask) {
                    if (int rc = stage_host(e, frame, in.mask, npix, &p)) return rc;
                    d_mask = static_cast<const uint8_t*>(p);
                }
            } else {
                return fail(ROFT_ERR_INVALID, "mem_kind must be ROFT_MEM_HOST or ROFT_MEM_DEVICE");
            }

            // ---- ImageSegmentationOFAidedSource::step_frame (hpp:127-231), schedule part
            c.slot_prev = (o.frame_idx + kPlaneSlots - 1) % kPlaneSlots;
            c.slot_cur = o.frame_idx % kPlaneSlots;
            c.has_new_mask = d_mask ? 1 : 0;
            c.new_mask = d_mask;
            if (d_mask) e->new_mask_frames |= 1u << t;
            c.first_mask = 0;
            if (d_mask && !o.seg_available) { o.seg_available = true; c.first_mask = 1; }
            if (!o.seg_available)
                return fail(ROFT_ERR_STATE, "no segmentation mask delivered yet: the first frame must carry one");
            const bool valid_flow = d_flow && !o.of_first_frame;
            o.of_first_frame = false;
            if (valid_flow) {
                const int keep = std::min(o.n_hist, e->hist_cap - 1);
                std::memmove(o.hist + 1, o.hist, sizeof(FlowEntry) * (size_t)keep);
                o.hist[0] = FlowEntry{d_flow, o.frame_idx, -1};
                o.n_hist = keep + 1;
                o.flows_since_mask++;
            }
            // Flows that later flows did not push out of the history in time (dropped flow frames): the caller may
            // recycle the buffer once the retention window closes, the reference keeps a clone -- so does the engine.
            for (int j = 0; j < o.n_hist; ++j) {
                FlowEntry& fe = o.hist[j];
                if (fe.owned >= 0 || o.frame_idx - fe.frame < e->hist_cap) continue;
                int k = -1;
                for (size_t q = 0; q < ho.owned.size(); ++q) {
                    bool referenced = ho.owned[q]->last_ref_frame >= e->completed_frames;
                    for (int j2 = 0; j2 < o.n_hist && !referenced; ++j2) referenced = o.hist[j2].owned == (int)q;
                    if (!referenced) { k = (int)q; break; }
                }
                if (k < 0) { ho.owned.push_back(new OwnedFlow()); k = (int)ho.owned.size() - 1; }
                HIP_TRY(ho.owned[k]->buf.ensure(fbytes));
                HIP_TRY(hipMemcpyAsync(ho.owned[k]->buf.p, fe.ptr, fbytes, hipMemcpyDeviceToDevice, e->up_stream));
                e->had_uploads = true;
                fe.ptr = ho.owned[k]->buf.p;
                fe.owned = k;
            }
            c.flow_valid = valid_flow ? 1 : 0;
            if (cfg.stamped_masks) {
                // OpticalFlowQueueHandler: window of 30 stamped flows; get_buffer_region(mask stamp) = the flows stored
                // after the first entry within 1 ms of it (OpticalFlowQueueHandler.cpp:18-58)
                c.stamped = 1;
                if (valid_flow) {
                    if (o.n_stamps == 30) std::memmove(o.stamps, o.stamps + 1, sizeof(double) * (size_t)(--o.n_stamps));
                    o.stamps[o.n_stamps++] = in.stamp;
                }
                c.n_region = 0;
                if (d_mask)
                    for (int i = 0; i < o.n_stamps; ++i)
                        if (std::fabs(o.stamps[i] - in.mask_stamp) < 1e-3) { c.n_region = o.n_stamps - (i + 1); break; }
            } else if (d_mask && !c.first_mask) {
                // a delivered mask consumes (or, when empty and the number of frames between masks is unknown, drops)
                // the buffered flows; with that number unknown ALL of them are chased (hpp:239-245)
                if (cfg.mask_frames_between <= 0 && o.flows_since_mask > kMaxFlowHist)
                    return fail(ROFT_ERR_CAPACITY, "more than ROFT_MAX_FLOW_CHASE flows buffered since the last mask");
                o.flows_since_mask = valid_flow ? 1 : 0;   // upper bound: 0 after a consumed mask, 1 after an empty one
            }
            c.n_hist = o.n_hist;
            for (int j = 0; j < o.n_hist; ++j) {
                c.flow[j] = o.hist[j].ptr;
                if (o.hist[j].owned >= 0) ho.owned[o.hist[j].owned]->last_ref_frame = frame;
            }

            // ---- ImageOpticalFlowMeasurement::freeze state machine (hpp:217-229)
            bool data_in = true;  // segmentation is available at this point
            if (!d_flow || o.flow_first_frame) {
                o.flow_first_frame = false;
                data_in = false;
            }
            c.vel_stage = data_in ? 1 : 0;
            c.depth_prev = o.depth_prev;
            c.depth_cur = d_depth;
            // (data_in implies valid_flow, so c.flow[0] is this frame's flow whenever the velocity stage runs)
            o.depth_prev = d_depth;

            // ---- outlier-rejection features on the first frame (ROFTFilter.cpp:313-322)
            if (cfg.use_pose_resync && !o.features_initialized) {
                c.feat_write = o.feat_next;
                o.feat_next = (o.feat_next + 1) % kFeatRing;
                o.feat_slot = c.feat_write;
                o.features_initialized = true;
            }
            c.frame_idx = frame;
            if (!build_pose_program(cfg, o, in, c))
                return fail(ROFT_ERR_CAPACITY, "more buffered velocities to replay than one frame's program holds (kMaxSteps)");
            e->lin_any[c.lane] = true;
            if (!lane_seen[(size_t)id * kNumLin + c.lane]) {
                // the object's first frame on this lane in the batch: is its first step's twist older than the batch?
                lane_seen[(size_t)id * kNumLin + c.lane] = 1;
                e->lane_objs[c.lane]++;
                const int age = (c.n_steps > 0 && c.steps[0].op) ? ((o.frame_idx - c.steps[0].twist_slot) & (kTwistRing - 1)) : 0;
                if (age > t && c.outlier_step == 0) e->lane_old_first[c.lane]++;   // (a replay whose first step is the one the outlier test follows)
            }
            o.last_touch[o.cur_slot] = b;
            if (c.outlier_step >= 0)
                max_outliers[c.lane] = std::max(max_outliers[c.lane], ++n_outliers[(size_t)id * kNumLin + c.lane]);
            if (c.outlier_step >= 0 && c.feat_read >= 0 && c.feat_read != c.feat_write &&
                e->feat_batch[(size_t)id * kFeatRing + c.feat_read] == b) e->feat_dep_in_batch = true;
            if (c.feat_write >= 0) {
                e->feat_batch[(size_t)id * kFeatRing + c.feat_write] = b;
                e->any_feat = true;
                // a feature set is re-used only when the batch that read or wrote it last has ended
                const int last = o.feat_use[c.feat_write];
                if (last >= 0 && last < b) { if (int rc = wait_batch(e, last)) return rc; }
                o.feat_use[c.feat_write] = b;
            }
            if (c.feat_read >= 0 && c.outlier_step >= 0) o.feat_use[c.feat_read] = b;
            if (c.outlier_step >= 0 && c.feat_read == c.feat_write) e->any_feat_now = true;
            o.frame_idx++;
        }
    }
    for (int l = 0; l < kNumLin; ++l) e->n_segments[l] = 1 + max_outliers[l];
    return ROFT_OK;
}

int roft_frames_submit(roft_engine* e, const roft_frame_input* inputs, int n_objects, int n_frames)
{
    if (!e || !inputs) return fail(ROFT_ERR_INVALID, "null argument");
    if (n_objects != (int)e->objs.size() || n_objects <= 0) return fail(ROFT_ERR_INVALID, "one input per object and frame required");
    if (n_frames < 1 || n_frames > e->T_max) return fail(ROFT_ERR_INVALID, "n_frames must be 1 .. roft_config::max_batch_frames");
    if (e->submitted) return fail(ROFT_ERR_STATE, "previous batch not stepped yet");
    HIP_TRY(hipSetDevice(e->cfg.device));
    double hp_t = e->host_prof ? host_now_us() : 0.0;
    e->cur_submit_t0 = host_now_us();
    // bound the batches in flight (see roft_engine::lead); this also frees the batch ring slot
    if (int rc = wait_batch(e, e->batch_counter - e->lead, &e->throttled)) return rc;
    e->cur_wait_us = host_now_us() - e->cur_submit_t0;
    HP_MARK(e, 0, hp_t);   // time blocked on the GPU
    e->backup.resize(e->objs.size());
    for (size_t i = 0; i < e->objs.size(); ++i) e->backup[i] = e->objs[i]->s;
    e->any_feat = e->any_feat_now = e->had_uploads = false;
    e->feat_dep_in_batch = false;
    e->new_mask_frames = 0;
    const int rc = submit_frames(e, inputs, n_objects, n_frames);
    HP_MARK(e, 1, hp_t);
    int rc2 = ROFT_OK;
    if (e->had_uploads) {
        // HOST buffers belong to the caller again when this call returns
        const int slot = e->batch_counter % roft_engine::kBatchRing;
        hipError_t err = hipEventRecord(e->ev_up[slot], e->up_stream);
        if (err == hipSuccess) err = hipEventSynchronize(e->ev_up[slot]);
        if (err != hipSuccess) rc2 = fail(ROFT_ERR_DEVICE, std::string("input upload: ") + hipGetErrorString(err));
    }
    HP_MARK(e, 2, hp_t);
    if (rc != ROFT_OK || rc2 != ROFT_OK) {
        const std::string msg = g_last_error;
        for (size_t i = 0; i < e->objs.size(); ++i) e->objs[i]->s = e->backup[i];
        return fail(rc != ROFT_OK ? rc : rc2, msg);
    }
    e->cur_T = n_frames;
    e->submitted = true;
    e->device_pointers_checked = true;
    e->cur_submit_us = host_now_us() - e->cur_submit_t0;
    return ROFT_OK;
}

int roft_frame_submit(roft_engine* e, const roft_frame_input* inputs, int n_inputs)
{
    return roft_frames_submit(e, inputs, n_inputs, 1);
}

// Timing marks accumulate over any number of steps until roft_engine_get_timing() collects them:
// mark i closes the interval (event i-1, event i] and attributes it to kernel id tmark[i]
// (-1 = step start, attributes nothing).
static void tmark(roft_engine* e, const char* name, int which = 0)
{
    if (!e->timing) return;
    if (e->timing_level == 1) return;   // only the roofline kernel is timed (tmark_kernel)
    const size_t idx = e->tmark.size();
    while (e->tev.size() <= idx) {
        hipEvent_t ev;
        (void)hipEventCreate(&ev);
        e->tev.push_back(ev);
    }
    int id = -1;
    if (name) {
        for (size_t i = 0; i < e->tnames_s.size(); ++i)
            if (e->tnames_s[i] == name) id = (int)i;
        if (id < 0) { e->tnames_s.push_back(name); id = (int)e->tnames_s.size() - 1; }
    }
    e->tmark.push_back(id);
    e->tstream.push_back(which);
    (void)hipEventRecord(e->tev[idx], which == 1 ? e->pose_stream[0] : (which == 3 ? e->pose_stream[1] : (which == 2 ? e->vel_stream : (which == 4 ? e->up_stream : e->stream))));
}

// Timing of ONE kernel by a start / stop event pair bound to its dispatch (two consecutive marks: the first opens the
// interval, the second closes it and attributes it to `name`).  Leaves the events null when timing is off.
static void tmark_kernel(roft_engine* e, const char* name, int which, hipEvent_t* start, hipEvent_t* stop)
{
    if (!e->timing) return;
    const size_t idx = e->tmark.size();
    while (e->tev.size() <= idx + 1) {
        hipEvent_t ev;
        (void)hipEventCreate(&ev);
        e->tev.push_back(ev);
    }
    int id = -1;
    for (size_t i = 0; i < e->tnames_s.size(); ++i)
        if (e->tnames_s[i] == name) id = (int)i;
    if (id < 0) { e->tnames_s.push_back(name); id = (int)e->tnames_s.size() - 1; }
    e->tmark.push_back(-1);
    e->tstream.push_back(which);
    e->tmark.push_back(id);
    e->tstream.push_back(which);
    *start = e->tev[idx];
    *stop = e->tev[idx + 1];
}

#define CHECK_LAUNCH(what)                                                                              \
    do {                                                                                                \
        hipError_t _e = hipGetLastError();                                                              \
        if (_e != hipSuccess) return fail(ROFT_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(_e)); \
    } while (0)

static int step_batch(roft_engine* e)
{
    constexpr int R = roft_engine::kBatchRing;
    EngineArrays a = e->arr.a;
    hipStream_t s = e->stream, sv = e->vel_stream;
    const int slot = e->batch_counter % R;
    const int T = e->cur_T;
    const bool multi = e->multi;
    const bool full = e->timing && e->timing_level > 1;   // markers between the launches carry the events' roles as well
    long long& launches = e->stats.launches;
    long long& evops = e->stats.event_ops;
    double hp_t = e->host_prof ? host_now_us() : 0.0;
    (void)hipGetLastError();   // a stale error of another library on this thread is not this step's
    a.T = T;
    a.ctrl = e->dctrl[slot].p;
    {
        // this batch's mask tables (parity) and the row of the other table that carries the state in
        const size_t table = (size_t)(kMaxBatch + 1) * a.n_obj;
        const int par = e->batch_counter & 1;
        MaskRec* base = e->arr.mrec.p;
        a.mrec = base + par * table;
        a.mrec_carry = e->prev_T > 0 ? base + (1 - par) * table + (size_t)e->prev_T * a.n_obj : a.mrec;
        a.slot_new = kSlotNew + par * kMaxBatch;
        a.slot_prev0 = (e->frame_counter + kPlaneSlots - 1) % kPlaneSlots;   // (submit_frames: slot_prev of every object)
    }
    static_assert(sizeof(FrameCtrl) % 16 == 0, "FrameCtrl is copied in 16-byte units");
    // Frame-granular hand-over to the pose lanes (below) -- and, with CUs to spare (at most one object per eight CUs), lanes that
    // do not even wait for the velocity filter to be resident: they start behind the batch's control blocks and take every twist
    // when its tag appears, so the first segment of a re-sync (the pose step, which reads a twist of six frames ago) and its
    // outlier test run next to the batch's mask frames instead of behind them.
    const bool cus_to_spare = 8 * a.n_obj <= device_cu_count();
    // (round 5: keyed on the batch index, not on whether the submit call happened to wait -- see roft_engine::steady)
    const bool steady = e->steady = (e->batch_counter - e->idle_mark) >= e->lead;
    const bool handoff = multi && T > 1 && e->handoff_mode > 0 && e->wait_value_ok && !(e->handoff_mode == 1 && steady && !cus_to_spare) &&
                         !e->feat_dep_in_batch && !e->any_feat_now && e->arr.skf_started.p != nullptr;
    static const int early_env = getenv("ROFT_EARLY_LANES") ? atoi(getenv("ROFT_EARLY_LANES")) : 1;   // (experiments; 0 for several processes on one GPU)
    // Early lanes spin inside their kernel for twists whose producer kernel is not even enqueued yet (it sits behind the mask
    // chain on another stream): progress needs (i) hardware queues of their own for the four chains -- a stream set that was
    // PROBED free of conflicts -- and (ii) CUs the lanes do not occupy: at most one object per eight CUs counted over THIS
    // engine, which only holds when no other engine of the process works on the device (other processes: ROFT_EARLY_LANES=0).
    // Otherwise the lanes fall back to the gate on resident velocity-filter workgroups (`handoff`), where a lane only ever
    // waits for workgroups that run.
    const bool early_ok = handoff && !steady && early_env != 0 && e->streams && e->streams->conflicts == 0 &&
                          (e->alone_on_device = alone_on_device(e->streams));
    const bool early_lanes = early_ok && cus_to_spare;   // (bursts: in the steady state a lane is behind anyway, and at 1280x720 the early tests cost 3 %)
    // ... and, whatever the number of objects (round 5): a lane whose objects START the batch with the first step of a re-sync
    // replay.  That step reads the twist of pose_frames_between frames ago -- published by an earlier batch -- and ends the lane's
    // first segment (the outlier test follows it): segment and test need nothing of this batch but its control blocks, so in a
    // burst they run next to the batch's mask frames instead of behind its velocity filter, and only the SECOND segment (the rest
    // of the replay: this batch's twists) is held at the gate.  The few objects of the lane that are out of phase (a dropped
    // pose: they start with an ordinary step) wait for their twist inside the kernel, on CUs nobody needs -- at most one per
    // eight CUs, else the lane is not released early.
    bool early_lane[kNumLin];
    for (int l = 0; l < kNumLin; ++l)
        early_lane[l] = early_lanes || (early_ok && T > 1 && e->n_segments[l] > 1 && e->lane_old_first[l] > 0 &&
                                        8 * (e->lane_objs[l] - e->lane_old_first[l]) <= device_cu_count() &&
                                        // (the replay-first objects wait too -- for a twist of the batch BEFORE, whose velocity filter is
                                        //  enqueued and may still be publishing: all of the lane's workgroups together leave it half the device)
                                        2 * e->lane_objs[l] <= device_cu_count());
    const bool any_early = early_lane[0] || early_lane[1];
    const long long launches0 = e->stats.launches, evops0 = e->stats.event_ops;

    // ---- control blocks of the batch -> device (+ reset of the mask chain's counters), ingest of the masks delivered
    //      with the batch (tables and ingest slots of this batch's parity: the carry of the chain before stays readable).
    //      Batches: on the UPLOAD stream, so that it happens while the mask chain of the batch before is still walking -- the
    //      mask stream is the longest serial chain of the steady state (round 5 timeline: 14 + 38 + 200 us of a 252 us period),
    //      and the 38 us were this preparation.  What it writes was last read by the mask chain TWO batches back (tables and
    //      ingest slots of its parity; the chain in between reads one row of them as its carry, but none of the counters
    //      that are reset here), which it therefore waits for.  Only in the steady state (a function of the batch index): in a
    //      burst the mask stream is not behind, and the event between the two streams is one more hop on the first batches'
    //      critical path -- measured, one box: 120 steps +1.5 %; 20 steps -5 % and 8 objects -5 % if bursts did the same.
    //      (Rounds 3 - 4 measured the same idea 3 % slower at 240 steps: the pose lanes were the bottleneck then.)
    //      And only when the device is full (more than one object per eight CUs): with fewer objects a batch is a chain of
    //      latencies at every load and the mask stream is never the longest one (60 steps, 16 / 32 objects: 5.2e5 / 9.4e5 with
    //      the preparation ahead in steady batches, 5.8e5 / 1.02e6 without).
    //      ROFT_PREP_AHEAD = 0 never, 2 always, 3 in every steady batch.
    const int prep_env = e->prep_mode;
    const bool prep = multi && T > 1 && (prep_env == 2 || (prep_env == 3 && steady) || (prep_env == 1 && steady && !cus_to_spare)) && e->up_stream != s;
    hipStream_t sp0 = prep ? e->up_stream : s;
    if (multi && e->had_uploads && !prep) { HIP_TRY(hipStreamWaitEvent(s, e->ev_up[slot], 0)); ++evops; }   // (prep: same stream as the uploads)
    if (prep && e->batch_counter >= 2) { HIP_TRY(hipStreamWaitEvent(sp0, e->ev_mask[(slot + R - 2) % R], 0)); ++evops; }
    tmark(e, nullptr, prep ? 4 : 0);
    {
        const size_t n16 = sizeof(FrameCtrl) * (size_t)a.n_obj * T / 16;
        // Events that complete with a kernel (hipExtLaunchKernelGGL stop events) cost neither the barrier packet nor
        // the host call of a hipEventRecord behind it.
        hipExtLaunchKernelGGL(ctrl_upload_kernel, dim3((unsigned)std::min<size_t>((n16 + 255) / 256, 64)), dim3(256), 0, sp0,
                              nullptr, (multi && (T == 1 || any_early)) ? e->ev_ctrl[slot] : nullptr, 0,
                              reinterpret_cast<const uint4*>(e->stage[slot]), a, n16, 1);
        ++launches;
    }
    CHECK_LAUNCH("FrameCtrl upload");
    {
        int last = -1;
        for (int t = 0; t < T; ++t)
            if (e->new_mask_frames & (1u << t)) last = t;
        for (int t = 0; t < T; ++t)
            if (e->new_mask_frames & (1u << t)) { launch_mask_ingest(a, t, sp0, (prep && !full && t == last) ? e->ev_prep[slot] : nullptr); ++launches; }
        CHECK_LAUNCH("mask ingest");
        if (prep) {
            if (full || last < 0) { HIP_TRY(hipEventRecord(e->ev_prep[slot], sp0)); ++evops; }
            tmark(e, "mask_prepare", 4);
            HIP_TRY(hipStreamWaitEvent(s, e->ev_prep[slot], 0));
            ++evops;
        }
    }
    HP_MARK(e, 3, hp_t);
    // ---- mask chain: every object's masks frame after frame
    tmark(e, nullptr, 0);
    // In a burst the velocity chain is released when the masks its flow measurements read are complete -- frames 0 .. T - 2: the
    // measurement of frame t is taken inside the mask of frame t - 1 --, one mask frame (the one that chases a delivered mask
    // through six flows, the longest) before the chain ends; the features kernel behind the velocity filter waits for the
    // chain's end.  Not in the steady state (a function of the batch index): latency buys nothing there, and the event costs the
    // mask stream -- the longest serial chain -- one more small launch.  And only with CUs to spare (at most one object per eight
    // CUs): with 64 objects the flow measurement then runs NEXT to the longest mask frame instead of behind it and takes 48 us
    // instead of 30 for no gain in the window (1.084 / 1.072e6), while 16 objects gain 5 - 9 %.
    const int part_env = e->part_mode;   // (0 never, 2 always, 3 in every burst batch)
    const bool part_gate = multi && T > 1 && (part_env == 2 || (part_env == 3 && !steady) || (part_env == 1 && !steady && cus_to_spare));
    launches += launch_mask_chain(a, e->cfg.mask_frames_between, e->cfg.flow_aided_segmentation, e->new_mask_frames, s,
                                  (multi && !full) ? e->ev_mask[slot] : nullptr, part_gate ? e->ev_part[slot] : nullptr);
    CHECK_LAUNCH("mask chain");
    tmark(e, "mask_chain", 0);
    if (multi && full) { HIP_TRY(hipEventRecord(e->ev_mask[slot], s)); ++evops; }
    // Outlier-rejection features of the batch's pose frames (they read the planes the mask chain just wrote).  Batches:
    // on the velocity stream behind the velocity filter -- that stream has waited for this mask chain, has time to spare,
    // and the pose lanes wait for its end anyway, so the features cost the mask chain (the longest one) nothing and need
    // no event of their own.  One-frame submits: on the mask chain's stream; the pose chain waits for them only when a
    // test reads a set buffered in this very frame (older sets are covered by the velocity chain's wait on that stream).
    const bool feat_on_vel = multi && T > 1;
    const bool want_ev_feat = multi && e->any_feat && !feat_on_vel && (T > 1 || e->any_feat_now);
    if (e->any_feat && !feat_on_vel) {
        launch_features(a, s, (want_ev_feat && !full) ? e->ev_feat[slot] : nullptr);
        ++launches;
        CHECK_LAUNCH("features");
        tmark(e, "features", 0);
        if (want_ev_feat && full) { HIP_TRY(hipEventRecord(e->ev_feat[slot], s)); ++evops; }
    }
    HP_MARK(e, 4, hp_t);

    // ---- velocity chain: the measurement of frame k needs the control blocks and the mask planes of frame k-1 --
    //      the previous batch's for a one-frame batch (ordered by the upload, which follows that batch's mask chain),
    //      this batch's mask chain otherwise
    if (multi) { HIP_TRY(hipStreamWaitEvent(sv, T == 1 ? e->ev_ctrl[slot] : (part_gate ? e->ev_part[slot] : e->ev_mask[slot]), 0)); ++evops; }
    const int radius = (int)(size_t)e->cfg.subsampling_radius;
    {
        // the roofline kernel is timed by a start / stop event pair on its own dispatch: its duration as rocprofv3
        // reports it, with no marker packets around it
        hipEvent_t k1_start = nullptr, k1_stop = nullptr;
        tmark_kernel(e, "flow_measure", 2, &k1_start, &k1_stop);
        // ... and, next to it, on the device's own clock: every workgroup leaves its start and end (first one in to last one
        // out = the launch as the kernel trace of a profiler sees it, without the packets the event pair brings along)
        EngineArrays ak = a;
        if (e->timing && (int)e->span_wgs.size() < roft_engine::kSpanLaunches) {
            const size_t per_launch = (size_t)2 * kMaxBatch * e->cfg.max_objects;
            if (e->k1_span.p) {   // (allocated by roft_engine_enable_timing)
                ak.k1_span = e->k1_span.p + per_launch * e->span_wgs.size();
                e->span_wgs.push_back(a.T * a.n_obj);
            }
        }
        launch_flow_measure(ak, e->cfg.depth_maximum, radius, sv, k1_start, k1_stop);
        ++launches;
        CHECK_LAUNCH("flow measurement");
    }
    const bool feat_last = feat_on_vel && e->any_feat;
    // Frame-granular hand-over to the pose lanes: their kernels are released when every workgroup of this velocity filter is
    // resident and take each twist when its tag appears (k_skf.hip / k_ukf.hip), instead of starting behind the filter's last
    // frame and the features kernel.  Not when an outlier test of the batch reads features buffered by this very batch (they
    // are extracted behind the filter), not on one stream, and -- by default -- only while the host is not throttled by the
    // in-flight bound: a lane that waits inside its kernel holds the CU it waits on, which a full pipeline cannot spare.
    // ... unless the device has CUs to spare anyway (at most one object per eight CUs: 32 on an MI355X -- measured: always handing
    // over is worth +4 - 6 % at 8 and 32 objects in 60-step runs, +1 - 2 % in the steady state at 32, -1 % at 64): `handoff` above.
    a.handoff = handoff ? 1 : 0;
    a.skf_started = e->arr.skf_started.p;
    e->vel_used[slot] = multi;
    launch_skf_chain(a, e->cfg.flow_weighting, sv, (multi && !full && !feat_last) ? e->ev_vel[slot] : nullptr);
    ++launches;
    if (hipError_t le = hipGetLastError()) {
        // the filter's workgroups will never count themselves in: no lane may ever wait for them (a stream-wait on a value has
        // no timeout) -- the hand-over is off for the rest of this engine's life
        e->handoff_mode = 0;
        return fail(ROFT_ERR_DEVICE, std::string("velocity filter chain: ") + hipGetErrorString(le));
    }
    e->skf_total += (unsigned long long)a.n_obj;   // (only once the launch is known to be enqueued: the lanes' gates wait for this count)
    tmark(e, "skf_chain", 2);
    if (feat_last) {
        if (part_gate) { HIP_TRY(hipStreamWaitEvent(sv, e->ev_mask[slot], 0)); ++evops; }   // (the planes of the batch's last frame)
        launch_features(a, sv, !full ? e->ev_vel[slot] : nullptr);
        ++launches;
        CHECK_LAUNCH("features");
        tmark(e, "features", 2);
    }
    if (multi && full) { HIP_TRY(hipEventRecord(e->ev_vel[slot], sv)); ++evops; }
    HP_MARK(e, 5, hp_t);

    // ---- pose chain (needs the twists of the batch; the next batches' image chains do not wait for it), one stream per
    //      lane: the frames before a pose arrival and the frames from it on belong to different belief lineages and
    //      do not depend on each other (BeliefSlot in roft_device.h), so the re-sync replay of this batch runs next to
    //      the ordinary steps of the other lineage -- of this batch and of the neighbouring ones
    for (int lin = 0; lin < kNumLin; ++lin) {
        hipStream_t sp = e->pose_stream[lin];
        e->done_used[slot][lin] = e->lin_any[lin];
        // slots handed over to this lane (submit_frames): behind the other lane's last launch that touched them
        const int wb = e->relabel_wait[lin];
        if (multi && wb >= e->completed_batches && wb < e->batch_counter && e->done_used[wb % R][1 - lin]) {
            HIP_TRY(hipStreamWaitEvent(sp, e->ev_done[wb % R][1 - lin], 0));
            ++evops;
        }
        if (!e->lin_any[lin]) continue;
        const int which = lin == 0 ? 1 : 3;
        // (an early lane's OUTLIER TEST waits for the velocity chain of the batch before -- its features kernel: the sets this
        //  batch's tests read were buffered there or earlier; the pose step in front of the test needs none of that and starts
        //  behind the control blocks alone: the features kernel runs ~35 us behind the velocity filter's last twist)
        int wait_prev_vel = -1;
        if (multi && early_lane[lin]) {
            HIP_TRY(hipStreamWaitEvent(sp, e->ev_ctrl[slot], 0));
            ++evops;
            const int pb = e->batch_counter - 1;
            if (pb >= e->completed_batches && pb >= 0 && e->vel_used[pb % R]) wait_prev_vel = pb % R;
        } else if (multi && handoff) {
            HIP_TRY(hipStreamWaitValue64(sp, e->arr.skf_started.p, e->skf_total, hipStreamWaitValueGte, ~0ull));
            ++evops;
        } else if (multi) {
            HIP_TRY(hipStreamWaitEvent(sp, e->ev_vel[slot], 0));
            ++evops;
            // The pose chain reads mask-chain products only through the feature ring.  With one-frame batches the set an
            // outlier test reads was buffered by an earlier batch -- covered by ev_vel, since the velocity chain waited for
            // the mask chain of the batch before -- unless it is this very frame's.
            if (want_ev_feat && e->n_segments[lin] > 1) { HIP_TRY(hipStreamWaitEvent(sp, e->ev_feat[slot], 0)); ++evops; }
        }
        tmark(e, nullptr, which);
        for (int seg = 0; seg < e->n_segments[lin]; ++seg) {
            const bool last = seg == e->n_segments[lin] - 1;
            if (seg == 1 && multi && early_lane[lin] && !early_lanes) {
                // (a lane released early for its replay's first step: what follows needs this batch's twists -- held until the
                //  velocity filter's workgroups are resident, like a lane of a hand-over batch that was not released early)
                HIP_TRY(hipStreamWaitValue64(sp, e->arr.skf_started.p, e->skf_total, hipStreamWaitValueGte, ~0ull));
                ++evops;
            }
            launch_ukf_chain(a, e->cfg.ut, seg == 0, lin, sp, (last && !full) ? e->ev_done[slot][lin] : nullptr);
            ++launches;
            CHECK_LAUNCH("pose chain segment");
            tmark(e, "ukf_chain", which);
            if (!last) {
                // bands per alternative: the caller's number, else by the CUs to spare -- and half of that while the host runs
                // `lead` batches ahead of the device (a long sequence in its steady state: fewer, longer workgroups leave more
                // CUs to the chains; 64 objects: +5 %, and -2.5 % if a 20-frame burst did the same).  The likelihood sums are
                // exact, so the band count changes no result.
                OutlierLaunchOpts oo;
                static const int steady_parts_env = getenv("ROFT_OUTLIER_STEADY_DIV") ? atoi(getenv("ROFT_OUTLIER_STEADY_DIV")) : 2;   // (experiments)
                if (e->cfg.outlier_bands_per_alternative == 0 && steady && steady_parts_env > 1) oo.parts = -steady_parts_env;   // (-d: the automatic count / d)
                if (wait_prev_vel >= 0) { HIP_TRY(hipStreamWaitEvent(sp, e->ev_vel[wait_prev_vel], 0)); ++evops; wait_prev_vel = -1; }
                launch_outlier(a, lin, sp, nullptr, &oo);
                ++launches;
                CHECK_LAUNCH("outlier rejection");
                tmark(e, "outlier_render_likelihood", which);
            }
        }
        if (full) { HIP_TRY(hipEventRecord(e->ev_done[slot][lin], sp)); ++evops; }
    }
    HP_MARK(e, 6, hp_t);
    if (e->host_prof) e->hp_batches++;
    {
        roft_batch_trace& tr = e->trace[e->batch_counter % roft_engine::kTraceRing];
        tr = roft_batch_trace{};
        tr.batch = e->batch_counter;
        tr.frames = T;
        tr.steady = steady; tr.throttled = e->throttled; tr.handoff = handoff; tr.early_lanes = (early_lanes ? 4 : 0) | (early_lane[0] ? 1 : 0) | (early_lane[1] ? 2 : 0);
        tr.outlier_parts_halved = (e->cfg.outlier_bands_per_alternative == 0 && steady) ? 1 : 0;
        tr.launches = (int)(e->stats.launches - launches0);
        tr.event_ops = (int)(e->stats.event_ops - evops0);
        tr.t_submit_us = e->cur_submit_t0; tr.submit_us = e->cur_submit_us; tr.wait_us = e->cur_wait_us;
    }
    HIP_TRY(hipGetLastError());
    return ROFT_OK;
}

int roft_step(roft_engine* e)
{
    if (!e) return fail(ROFT_ERR_INVALID, "null engine");
    if (!e->submitted) return fail(ROFT_ERR_STATE, "roft_frame_submit must precede roft_step");
    HIP_TRY(hipSetDevice(e->cfg.device));
    const double t_step0 = host_now_us();
    const int rc = step_batch(e);
    if (rc != ROFT_OK && e->arr.mask_general.p) {
        // A step that failed between the mask frames and mask_general_kernel (its only reader, which clears the bits it has
        // served) leaves bits of THIS batch's frames behind; the next batch's general kernel would replay those frame indices
        // against its own tables.  Clear them behind whatever the mask stream still carries (best effort: the device may be gone).
        (void)hipMemsetAsync(e->arr.mask_general.p, 0, sizeof(unsigned) * (size_t)std::max(e->arr.a.n_obj, 1), e->stream);
        (void)hipGetLastError();
    }
    {
        roft_batch_trace& tr = e->trace[e->batch_counter % roft_engine::kTraceRing];
        if (tr.batch == e->batch_counter) tr.step_us = host_now_us() - t_step0;
    }
    for (HostObject* ho : e->objs) { ho->stepped_slot = ho->s.cur_slot; ho->stepped_lane = ho->s.own[ho->s.cur_slot]; }
    // (a failed step leaves the engine consistent as far as the host can tell: the batch counts as enqueued)
    const int slot = e->batch_counter % roft_engine::kBatchRing;
    e->frame_counter += e->cur_T;
    e->prev_T = e->cur_T;
    e->batch_end_frame[slot] = e->frame_counter;
    e->batch_counter++;
    e->stats.frames += e->cur_T;
    e->stats.batches++;
    e->submitted = false;
    return rc;
}

int roft_sync(roft_engine* e)
{
    if (!e) return fail(ROFT_ERR_INVALID, "null engine");
    HIP_TRY(hipSetDevice(e->cfg.device));
    // the batches in flight one by one, in order (their completion times go into the batch trace), then whatever else the
    // streams carry (uploads, timing marks, reads of results)
    // (only with several batches in flight: a tracker used live -- one frame submitted, stepped and read back at a time -- goes
    //  straight to the stream synchronisations, whose wake-up is faster than an event's)
    const int first_open = e->completed_batches, n_open = e->batch_counter - e->completed_batches;
    if (e->multi && n_open > 1)
        for (int b = first_open; b < e->batch_counter; ++b)
            if (int rc = wait_batch(e, b)) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->multi) {
        HIP_TRY(hipStreamSynchronize(e->vel_stream));
        for (int l = 0; l < kNumLin; ++l) HIP_TRY(hipStreamSynchronize(e->pose_stream[l]));
        HIP_TRY(hipStreamSynchronize(e->up_stream));
    }
    for (int b = std::max(first_open, e->batch_counter - roft_engine::kTraceRing); b < e->batch_counter; ++b) {
        roft_batch_trace& tr = e->trace[b % roft_engine::kTraceRing];
        if (tr.batch == b && tr.t_done_us == 0.0) tr.t_done_us = host_now_us();
    }
    e->completed_batches = e->batch_counter;
    e->completed_frames = e->frame_counter;
    e->idle_mark = e->batch_counter;   // the device is idle: the next batches are a burst again (roft_engine::steady)
    return check_dev_error(e);
}

int roft_engine_get_batch_trace(roft_engine* e, roft_batch_trace* out, int capacity, int* n_out)
{
    if (!e || !out || !n_out || capacity < 0) return fail(ROFT_ERR_INVALID, "bad arguments");
    const int n = std::min(std::min(capacity, roft_engine::kTraceRing), e->batch_counter);
    for (int i = 0; i < n; ++i) out[i] = e->trace[(e->batch_counter - n + i) % roft_engine::kTraceRing];
    *n_out = n;
    return ROFT_OK;
}

int roft_get_state(roft_engine* e, int id, double pose13[13], double P12[144], double twist6[6], double Pv[36])
{
    if (!e || id < 0 || id >= (int)e->objs.size()) return fail(ROFT_ERR_INVALID, "bad object id");
    HIP_TRY(hipSetDevice(e->cfg.device));
    // v_mean, v_cov and the beliefs of the two lineages are the leading bytes of ObjState: one small copy into pinned
    // memory, queued behind the pose chain of the lineage that holds p_corr_belief_ after the last stepped frame (the
    // last writer of what is returned), then the other chains are waited for as roft_sync does
    static_assert(B_LIN0 == 0 && B_LIN1 == 1 && offsetof(ObjState, v_mean) == 0, "roft_get_state copies the head of ObjState");
    constexpr size_t kHead = offsetof(ObjState, belief) + kNumLin * sizeof(PoseBelief);
    if (!e->state_host) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->state_host), sizeof(ObjState)));
    const int lin = e->objs[id]->stepped_slot;
    hipStream_t last = e->multi ? e->pose_stream[e->objs[id]->stepped_lane] : e->stream;
    HIP_TRY(hipMemcpyAsync(e->state_host, e->arr.state.p + id, kHead, hipMemcpyDeviceToHost, last));
    if (int rc = roft_sync(e)) return rc;
    const ObjState* st = e->state_host;
    if (pose13) std::memcpy(pose13, st->belief[B_LIN0 + lin].mean, sizeof(double) * 13);
    if (P12) std::memcpy(P12, st->belief[B_LIN0 + lin].cov, sizeof(double) * 144);
    if (twist6) std::memcpy(twist6, st->v_mean, sizeof(double) * 6);
    if (Pv) std::memcpy(Pv, st->v_cov, sizeof(double) * 36);
    return ROFT_OK;
}

int roft_get_outputs(roft_engine* e, roft_object_output* outs, int n_outs)
{
    if (!e || !outs || n_outs != (int)e->objs.size()) return fail(ROFT_ERR_INVALID, "bad arguments");
    if (int rc = roft_sync(e)) return rc;
    std::vector<ObjState> st(n_outs);
    HIP_TRY(hipMemcpy(st.data(), e->arr.state.p, sizeof(ObjState) * n_outs, hipMemcpyDeviceToHost));
    for (int i = 0; i < n_outs; ++i) {
        const int lin = e->objs[i]->stepped_lane;
        std::memcpy(outs[i].pose, st[i].belief[B_LIN0 + e->objs[i]->stepped_slot].mean, sizeof(double) * 13);
        std::memcpy(outs[i].twist, st[i].v_mean, sizeof(double) * 6);
        outs[i].n_flow_points = st[i].n_flow_points;
        outs[i].outlier_selected = st[i].lane[lin].outlier_selected;
        outs[i].outlier_L[0] = st[i].lane[lin].outlier_L[0];
        outs[i].outlier_L[1] = st[i].lane[lin].outlier_L[1];
    }
    return ROFT_OK;
}

int roft_engine_enable_log(roft_engine* e, int n_frames)
{
    if (!e || n_frames <= 0) return fail(ROFT_ERR_INVALID, "bad arguments");
    if (int rc = roft_sync(e)) return rc;
    HIP_TRY(e->arr.log.ensure((size_t)n_frames * e->cfg.max_objects, true));
    e->arr.a.out_log = e->arr.log.p;
    e->arr.a.log_cap = n_frames;
    return ROFT_OK;
}

int roft_engine_get_log(roft_engine* e, int first_frame, int n_frames, roft_object_output* outs)
{
    if (!e || !outs || !e->arr.a.out_log) return fail(ROFT_ERR_INVALID, "log not enabled");
    if (int rc = roft_sync(e)) return rc;
    const int n_obj = e->arr.a.n_obj;
    for (int f = 0; f < n_frames;) {   // one copy per contiguous run of ring rows
        const int slot = (first_frame + f) % e->arr.a.log_cap;
        const int run = std::min(n_frames - f, e->arr.a.log_cap - slot);
        HIP_TRY(hipMemcpy(outs + (size_t)f * n_obj, e->arr.a.out_log + (size_t)slot * n_obj,
                          sizeof(roft_object_output) * n_obj * run, hipMemcpyDeviceToHost));
        f += run;
    }
    return ROFT_OK;
}

int roft_engine_get_log_rows(roft_engine* e, int first_frame, int n_frames, double* rows)
{
    if (!e || !rows || !e->arr.a.out_log || n_frames < 0) return fail(ROFT_ERR_INVALID, "log not enabled");
    const int n_obj = e->arr.a.n_obj;
    std::vector<roft_object_output> outs((size_t)n_frames * n_obj);
    if (n_frames == 0) return ROFT_OK;
    if (int rc = roft_engine_get_log(e, first_frame, n_frames, outs.data())) return rc;
    for (size_t i = 0; i < outs.size(); ++i) {
        std::memcpy(rows + 19 * i, outs[i].pose, sizeof(double) * 13);
        std::memcpy(rows + 19 * i + 13, outs[i].twist, sizeof(double) * 6);
    }
    return ROFT_OK;
}

int roft_get_mask(roft_engine* e, int id, uint8_t* mask_out)
{
    if (!e || !mask_out || id < 0 || id >= (int)e->objs.size()) return fail(ROFT_ERR_INVALID, "bad arguments");
    const Sched& o = e->objs[id]->s;
    if (o.frame_idx == 0 || e->submitted) return fail(ROFT_ERR_STATE, "no stepped frame to read the mask of");
    if (int rc = roft_sync(e)) return rc;
    const EngineArrays& a = e->arr.a;
    const int slot = (o.frame_idx - 1) % kPlaneSlots;
    const size_t npix = (size_t)a.cam.W * a.cam.H;
    DevBuf<uint8_t> tmp;
    HIP_TRY(tmp.ensure(npix));
    launch_planes_to_mask(nullptr, a.planes + plane_offset(a, id, slot, 1), (int)npix, tmp.p, e->stream);
    HIP_TRY(hipMemcpyAsync(mask_out, tmp.p, npix, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return ROFT_OK;
}

void* roft_engine_stream(roft_engine* e) { return e ? (void*)e->stream : nullptr; }

int roft_engine_enable_timing(roft_engine* e, int enable)
{
    if (!e) return fail(ROFT_ERR_INVALID, "null engine");
    // (the engine's device, not whatever device is current on this thread: the span buffer, the events and the priming
    //  dispatch below belong to it -- and nothing of a batch in flight may see the timing state change under it)
    HIP_TRY(hipSetDevice(e->cfg.device));
    if (int rc = roft_sync(e)) return rc;
    e->timing = enable != 0;
    e->timing_level = (enable == 1) ? 1 : 2;
    if (e->timing) {
        HIP_TRY(e->k1_span.ensure((size_t)2 * kMaxBatch * e->cfg.max_objects * roft_engine::kSpanLaunches, true));
        // Nothing of the timing machinery may happen for the first time inside the caller's timed region: the events exist
        // before it, and the velocity stream has carried a dispatch with a start / stop event pair (the first such dispatch
        // switches the queue's profiling on -- a host call of its own kind; one bench run in twenty spent 1.3 ms of a 1.4 ms
        // window on the host side of its launches).
        while (e->tev.size() < 64) {
            hipEvent_t ev;
            HIP_TRY(hipEventCreate(&ev));
            e->tev.push_back(ev);
        }
        e->tmark.reserve(256);
        e->tstream.reserve(256);
        hipExtLaunchKernelGGL(probe_tiny_kernel, dim3(1), dim3(64), 0, e->vel_stream, e->tev[0], e->tev[1], 0,
                              reinterpret_cast<int*>(e->k1_span.p));
        HIP_TRY(hipStreamSynchronize(e->vel_stream));
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e->tev[0], e->tev[1]);
        HIP_TRY(hipMemset(e->k1_span.p, 0, sizeof(unsigned long long)));
    }
    return ROFT_OK;
}

int roft_engine_get_timing(roft_engine* e, int* n_out, const char*** names_out, const float** ms_out,
                           const int** launches_out)
{
    if (!e || !n_out) return fail(ROFT_ERR_INVALID, "null argument");
    if (int rc = roft_sync(e)) return rc;
    const size_t nk = e->tnames_s.size();
    e->tms.assign(nk, 0.f);
    e->tlaunches.assign(nk, 0);
    long prev[5] = {-1, -1, -1, -1, -1};
    // ROFT_DUMP_MARKS=<file>: every mark as "stream name end_us duration_us" relative to the first one -- the timeline
    // of the chains without a profiler's launch overhead on the host (tools/marks_timeline.py)
    FILE* dump = nullptr;
    if (const char* path = getenv("ROFT_DUMP_MARKS")) dump = (e->timing_level > 1 && !e->tmark.empty()) ? fopen(path, "a") : nullptr;
    for (size_t i = 0; i < e->tmark.size(); ++i) {
        const int w = e->tstream[i];
        if (dump) {
            float t_ms = 0.f, d_ms = 0.f;
            (void)hipEventElapsedTime(&t_ms, e->tev[0], e->tev[i]);
            if (prev[w] >= 0) (void)hipEventElapsedTime(&d_ms, e->tev[prev[w]], e->tev[i]);
            fprintf(dump, "%d %s %.1f %.1f\n", w, e->tmark[i] >= 0 ? e->tnames_s[e->tmark[i]].c_str() : "-", 1e3 * t_ms, 1e3 * d_ms);
        }
        if (e->tmark[i] >= 0 && prev[w] >= 0) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, e->tev[prev[w]], e->tev[i]));
            e->tms[e->tmark[i]] += ms;
            e->tlaunches[e->tmark[i]] += 1;
        }
        prev[w] = (long)i;
    }
    if (dump) fclose(dump);
    if (!e->span_wgs.empty()) {
        // pseudo kernel "flow_measure_span": first workgroup in -> last workgroup out of each stamped launch, 10 ns ticks
        const size_t per_launch = (size_t)2 * kMaxBatch * e->cfg.max_objects;
        std::vector<unsigned long long> h(per_launch * e->span_wgs.size());
        HIP_TRY(hipMemcpy(h.data(), e->k1_span.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(e->k1_span.p, 0, h.size() * sizeof(unsigned long long)));
        double total_us = 0.0;
        int counted = 0;
        for (size_t l = 0; l < e->span_wgs.size(); ++l) {
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int w = 0; w < e->span_wgs[l]; ++w) {
                const unsigned long long a0 = h[l * per_launch + 2 * w], a1 = h[l * per_launch + 2 * w + 1];
                if (a0 == 0 || a1 == 0) continue;   // (a kernel variant that does not stamp)
                t0 = std::min(t0, a0);
                t1 = std::max(t1, a1);
            }
            if (t1 > t0) { total_us += (double)(t1 - t0) * 0.01; ++counted; }
        }
        e->span_wgs.clear();
        if (counted) {
            int id = -1;
            for (size_t i = 0; i < e->tnames_s.size(); ++i)
                if (e->tnames_s[i] == "flow_measure_span") id = (int)i;
            if (id < 0) { e->tnames_s.push_back("flow_measure_span"); e->tms.push_back(0.f); e->tlaunches.push_back(0); id = (int)e->tnames_s.size() - 1; }
            e->tms[id] = (float)(total_us * 1e-3);
            e->tlaunches[id] = counted;
        }
    }
    e->tmark.clear();
    e->tstream.clear();
    e->tnames.clear();
    for (auto& s : e->tnames_s) e->tnames.push_back(s.c_str());
    *n_out = (int)e->tnames_s.size();
    if (names_out) *names_out = e->tnames.data();
    if (ms_out) *ms_out = e->tms.data();
    if (launches_out) *launches_out = e->tlaunches.data();
    return ROFT_OK;
}

}  // extern "C"

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

// diagnostics (roft_engine.h section 4): phase counters of one object's last kernels; only filled by builds with a
// -DROFT_*_PROFILE switch
extern "C" int roft_debug_get_dbg(roft_engine* e, int id, long long out[32])
{
    if (!e || id < 0 || id >= (int)e->objs.size()) return ROFT_ERR_INVALID;
    if (roft_sync(e) != ROFT_OK) return ROFT_ERR_DEVICE;
    ObjState* st = new ObjState();
    hipError_t err = hipMemcpy(st, e->arr.state.p + id, sizeof(ObjState), hipMemcpyDeviceToHost);
    if (err == hipSuccess) std::memcpy(out, st->dbg, sizeof(long long) * 32);
    delete st;
    // (read and clear: the stamps of the frame kernels are maxima over their workgroups)
    if (err == hipSuccess) err = hipMemset(reinterpret_cast<char*>(e->arr.state.p + id) + offsetof(ObjState, dbg), 0, sizeof(long long) * 32);
    return err == hipSuccess ? ROFT_OK : ROFT_ERR_DEVICE;
}

// Diagnostics: (100 MHz ticks, workgroups) the workgroups of each kernel spent resident since the last call -- ResidencyKernel
// order, only filled by libraries built with -DROFT_RESIDENCY (tools/residency_budget.py)
extern "C" int roft_debug_outlier_split(int mode)
{
    roft::set_outlier_split(mode);
    return ROFT_OK;
}

extern "C" int roft_debug_get_residency(roft_engine* e, unsigned long long out[32])
{
    if (!e || !out) return ROFT_ERR_INVALID;
    if (roft_sync(e) != ROFT_OK) return ROFT_ERR_DEVICE;
    if (hipMemcpy(out, e->arr.residency.p, sizeof(unsigned long long) * 32, hipMemcpyDeviceToHost) != hipSuccess) return ROFT_ERR_DEVICE;
    return hipMemset(e->arr.residency.p, 0, sizeof(unsigned long long) * 32) == hipSuccess ? ROFT_OK : ROFT_ERR_DEVICE;
}

// Diagnostics (roft_engine.h section 4): which of the engine's HIP streams delay each other at the dispatch level.  out[a * 5
// + b] = microseconds until a one-workgroup kernel on stream b completes while stream a is busy placing a grid of three
// one-per-CU workgroups per CU (100 us each); ~15 us = independent, >= 80 us = b's launches queue behind a's.  Stream order:
// pose lane 0, pose lane 1, velocity chain, mask chain, upload.
// The rate at which this device serves SCATTERED 64-byte sectors (sectors per second): 16 M reads at random sector-aligned
// offsets of a 2 GiB scratch buffer, best of four launches.  The roofline of a gather-bound kernel such as the flow
// measurement (bench.py reports its gathers against this figure).  Allocates and frees 2 GiB; ~30 ms.
extern "C" int roft_debug_sector_rate(int device, double* sectors_per_second)
{
    if (!sectors_per_second) return fail(ROFT_ERR_INVALID, "null output");
    HIP_TRY(hipSetDevice(device));
    const size_t bytes = (size_t)2 << 30;
    unsigned* buf = nullptr;
    unsigned* sink = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&buf), bytes));
    const int grid = 2048;   // x 1024 threads x 8 loads = 16 M sectors
    hipError_t err = hipMalloc(reinterpret_cast<void**>(&sink), (size_t)grid * 1024 * sizeof(unsigned));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (err == hipSuccess) err = hipMemset(buf, 0, bytes);
    if (err == hipSuccess) err = hipEventCreate(&e0);
    if (err == hipSuccess) err = hipEventCreate(&e1);
    float best_ms = 0.f;
    for (int rep = 0; rep < 5 && err == hipSuccess; ++rep) {
        (void)hipEventRecord(e0, nullptr);
        hipLaunchKernelGGL(probe_sectors_kernel, dim3(grid), dim3(1024), 0, nullptr, buf, (unsigned)(bytes / 64 - 1), 0x9e3779b9u * (unsigned)(rep + 1), sink);
        (void)hipEventRecord(e1, nullptr);
        err = hipEventSynchronize(e1);
        float ms = 0.f;
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && (best_ms == 0.f || ms < best_ms)) best_ms = ms;   // (the first launch loads the code object)
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    (void)hipFree(buf);
    if (err != hipSuccess || !(best_ms > 0.f)) return fail(ROFT_ERR_DEVICE, std::string("sector-rate probe: ") + hipGetErrorString(err));
    *sectors_per_second = (double)grid * 1024.0 * 8.0 / ((double)best_ms * 1e-3);
    return ROFT_OK;
}

extern "C" int roft_debug_probe_streams(roft_engine* e, double out[25])
{
    if (!e || !out) return ROFT_ERR_INVALID;
    if (roft_sync(e) != ROFT_OK) return ROFT_ERR_DEVICE;
    hipStream_t st[5] = {e->pose_stream[0], e->pose_stream[1], e->vel_stream, e->stream, e->up_stream};
    (void)set_max_dynamic_lds(reinterpret_cast<const void*>(probe_blocker_kernel), 150 * 1024);
    DevBuf<int> flag;
    if (flag.ensure(1) != hipSuccess) return ROFT_ERR_DEVICE;
    const int cus = device_cu_count();
    for (int a = 0; a < 5; ++a)
        for (int b = 0; b < 5; ++b) {
            out[a * 5 + b] = 0.0;
            if (a == b || st[a] == st[b]) continue;
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipDeviceSynchronize();
                const double t0 = host_now_us();
                hipLaunchKernelGGL(probe_blocker_kernel, dim3(3 * cus), dim3(64), 150 * 1024, st[a], 10000ll);
                hipLaunchKernelGGL(probe_tiny_kernel, dim3(1), dim3(64), 0, st[b], flag.p);
                (void)hipStreamSynchronize(st[b]);
                best = std::min(best, host_now_us() - t0);
            }
            out[a * 5 + b] = best;
        }
    (void)hipDeviceSynchronize();
    return hipGetLastError() == hipSuccess ? ROFT_OK : ROFT_ERR_DEVICE;
}

// Host-logic check without a device (roft_engine.h section 4): runs the per-frame program builder -- the
// mirror of the Standard / PopBufferedMeasurement / RepeatOnlyVelocity state machine of
// CartesianQuaternionMeasurement::freeze and of the re-sync loop of ROFTFilter::filtering_step -- over a
// sequence of pose-validity flags and reports, per frame, the number of UKF launches, the number of
// corrections, whether the outlier test runs, and the twist-ring slots replayed.
extern "C" int roft_debug_plan(const roft_config* cfg, const int* pose_valid, int n_frames, int* n_steps, int* n_corrections,
                               int* outlier, int* slots /* n_frames x kMaxSteps, -1 padded */)
{
    if (!cfg || !pose_valid || n_frames < 0) return ROFT_ERR_INVALID;
    Sched o;
    roft_frame_input in{};
    for (int k = 0; k < n_frames; ++k) {
        FrameCtrl c;
        clear_ctrl(c);
        in.pose_valid = pose_valid[k];
        if (!build_pose_program(*cfg, o, in, c)) return ROFT_ERR_CAPACITY;
        o.frame_idx++;
        if (n_steps) n_steps[k] = c.n_steps;
        int nc = 0;
        for (int s = 0; s < c.n_steps; ++s) {
            nc += c.steps[s].n_corr;
            if (slots) slots[k * kMaxSteps + s] = c.steps[s].twist_slot;
        }
        if (slots) for (int s = c.n_steps; s < kMaxSteps; ++s) slots[k * kMaxSteps + s] = -1;
        if (n_corrections) n_corrections[k] = nc;
        if (outlier) outlier[k] = c.outlier_step;
    }
    return ROFT_OK;
}
