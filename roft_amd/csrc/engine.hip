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
#include "engine_internal.h"

namespace {
thread_local std::string g_last_error;
}  // namespace

namespace roft {
namespace host {
int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}
}  // namespace host
}  // namespace roft
const std::string& last_error() { return g_last_error; }

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

// =================================================================================================
// batched engine
// =================================================================================================

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

// a kernel gave up (EngineArrays::dev_error): sticky -- the filter state of the objects is no longer what the reference
// would hold
int check_dev_error(roft_engine* e)
{
    const int code = e->dev_error ? *reinterpret_cast<volatile int*>(e->dev_error) : 0;
    if (code == 0) return ROFT_OK;
    if (code == ROFT_DEV_ERROR_TWIST_WAIT)
        return fail(ROFT_ERR_DEVICE, "a pose lane waited two seconds for a twist of the velocity filter it runs next to and gave up "
                                     "(frame-granular hand-over; ROFT_HANDOFF=0 disables it): the results of the batch are invalid");
    return fail(ROFT_ERR_DEVICE, "a kernel reported error " + std::to_string(code));
}

// blocks until batch b (and therefore every earlier one) has ended on the GPU
int wait_batch(roft_engine* e, int b, bool* waited)
{
    if (waited) *waited = false;
    if (b < e->completed_batches || b >= e->batch_counter) return ROFT_OK;
    // (with the frame-granular hand-over a pose lane can end before the features kernel behind the velocity filter does)
    if (e->vel_used[b % roft_engine::kBatchRing]) {
        if (waited && hipEventQuery(e->ev_vel[b % roft_engine::kBatchRing]) == hipErrorNotReady) *waited = true;
        (void)hipGetLastError();
        HIP_TRY(hipEventSynchronize(e->ev_vel[b % roft_engine::kBatchRing]));
    }
    // (... or before the features kernel behind the batch's mask frames does, which reads the batch's depth images)
    if (e->feat_used[b % roft_engine::kBatchRing]) HIP_TRY(hipEventSynchronize(e->ev_feat[b % roft_engine::kBatchRing]));
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

const char* roft_last_error_string(void) { return g_last_error.c_str(); }

int roft_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- pinned host memory pool (roft_engine.h section 2b) ----
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
bool alone_on_device(const StreamSet* mine)
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
        for (hipEvent_t* ev : {&e->ev_up[i], &e->ev_ctrl[i], &e->ev_mask[i], &e->ev_part[i], &e->ev_prep[i], &e->ev_feat[i], &e->ev_vel[i], &e->ev_skf[i], &e->ev_done[i][0], &e->ev_done[i][1]})
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
            hipExtLaunchKernelGGL(probe_tiny_kernel, dim3(1), dim3(64), 0, e->vel_stream, nullptr, e->ev_skf[i], 0, flag);
            HIP_TRY(hipStreamWaitEvent(e->pose_stream[i & 1], e->ev_skf[i], 0));
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
    if (const char* pm = getenv("ROFT_FEAT_ON_MASK")) e->feat_mask_mode = atoi(pm);
    if (const char* pm = getenv("ROFT_LANES_WAIT_SKF")) e->lanes_wait_skf = atoi(pm);
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
        for (hipEvent_t ev : {e->ev_up[i], e->ev_ctrl[i], e->ev_mask[i], e->ev_part[i], e->ev_prep[i], e->ev_feat[i], e->ev_vel[i], e->ev_skf[i], e->ev_done[i][0], e->ev_done[i][1]})
            if (ev) (void)hipEventDestroy(ev);
        if (e->stage[i]) (void)hipHostFree(e->stage[i]);
        if (e->gather_tab[i]) (void)hipHostFree(e->gather_tab[i]);
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
        if (!d->mesh.verts || !d->mesh.tris) return bail(ROFT_ERR_INVALID, "mesh: null vertex or triangle array");
        for (size_t i = 0; i < (size_t)3 * d->mesh.n_tris; ++i)   // (the rasteriser indexes the vertex array with these)
            if (d->mesh.tris[i] < 0 || d->mesh.tris[i] >= d->mesh.n_verts)
                return bail(ROFT_ERR_INVALID, "mesh: triangle " + std::to_string(i / 3) + " refers to vertex " + std::to_string(d->mesh.tris[i]) +
                                                  " of " + std::to_string(d->mesh.n_verts));
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



