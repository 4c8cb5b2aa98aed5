// engine_submit.hip -- roft_frames_submit / roft_frame_submit: what depends only on the delivery schedule is resolved here, on the
// host, into one FrameCtrl block per object and frame (the frame program: build_pose_program), HOST inputs are staged, the
// lanes are balanced.  Reference: ROFTFilter::filtering_step (src/roft-lib/src/ROFTFilter.cpp:255-452),
// CartesianQuaternionMeasurement::freeze (src/roft-lib/src/CartesianQuaternionMeasurement.cpp:92-348),
// ImageSegmentationOFAidedSource::step_frame (include/ROFT/ImageSegmentationOFAidedSource.hpp:127-231).
#include "engine_internal.h"

// The UKF steps of one frame (ROFTFilter.cpp:327-367 over CartesianQuaternionMeasurement::freeze, cpp:92-348).
// Returns false when the frame needs more than kMaxSteps steps.
bool build_pose_program(const roft_config& cfg, Sched& o, const roft_frame_input& in, FrameCtrl& c)
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

// Many small HOST images -> their staging copies in ONE launch: workgroup (x, y) copies 16-byte units x, x + gridDim.x, ... of item y.
// The sources are PINNED host buffers the device can address (stage_host checks); a 300 KB hipMemcpyAsync costs ~11 us of
// which 6 are the transfer, and a delivery brings one mask per object: 64 copies = 0.7 ms per six frames in the shared-scene leg
// of bench.py (28 GB/s), against one kernel that keeps the link busy.
__global__ __launch_bounds__(256) void gather_copy_kernel(const GatherItem* __restrict__ tab)
{
    const GatherItem it = tab[blockIdx.y];
    const uint4* src = reinterpret_cast<const uint4*>(it.src);
    uint4* dst = reinterpret_cast<uint4*>(it.dst);
    const size_t n16 = it.bytes / 16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// the items collected by stage_host during this submit -> one launch on the upload stream (before the submit waits for its uploads)
static int flush_gather(roft_engine* e)
{
    if (e->gather.empty()) return ROFT_OK;
    const int slot = e->batch_counter % roft_engine::kBatchRing;
    if (!e->gather_tab[slot]) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->gather_tab[slot]), sizeof(GatherItem) * kGatherCap, hipHostMallocMapped));
    const size_t n = e->gather.size();
    std::memcpy(e->gather_tab[slot], e->gather.data(), sizeof(GatherItem) * n);
    size_t largest = 0;
    for (const GatherItem& g : e->gather) largest = std::max(largest, g.bytes);
    const unsigned gx = (unsigned)std::max<size_t>(1, std::min<size_t>(32, (largest / 16 + 2047) / 2048));   // ~8 units per thread
    hipLaunchKernelGGL(gather_copy_kernel, dim3(gx, (unsigned)n), dim3(256), 0, e->up_stream, e->gather_tab[slot]);
    e->gather.clear();
    if (hipError_t le = hipGetLastError()) return fail(ROFT_ERR_DEVICE, std::string("gather copy of HOST inputs: ") + hipGetErrorString(le));
    e->stats.h2d_copies++;
    return ROFT_OK;
}

// the device address of a PINNED host buffer the GPU can read in place, or null (pageable memory, or not identity-mapped)
static const void* pinned_device_pointer(const void* host)
{
    hipPointerAttribute_t attr{};
    if (hipPointerGetAttributes(&attr, host) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (attr.type != hipMemoryTypeHost) return nullptr;
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, const_cast<void*>(host), 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return dp;
}

// device copy of one HOST image of `frame` (uploads once per distinct host pointer and frame)
static int stage_host(roft_engine* e, int frame, const void* host, size_t bytes, const void** dev)
{
    StageFrame& sf = e->staging[frame % e->retain];
    for (auto& pr : sf.seen)
        if (pr.first == host) { *dev = pr.second; return ROFT_OK; }
    unsigned char* d = nullptr;
    if (int rc = stage_alloc(e, frame, bytes, &d)) return rc;
    static const int gather_env = getenv("ROFT_GATHER_COPY") ? atoi(getenv("ROFT_GATHER_COPY")) : 1;   // (experiments: 0 = one copy per image)
    const void* dp = nullptr;
    if (gather_env && bytes <= kGatherMaxBytes && (bytes & 15) == 0 && (reinterpret_cast<uintptr_t>(host) & 15) == 0 &&
        (int)e->gather.size() < kGatherCap && (dp = pinned_device_pointer(host)) != nullptr) {
        e->gather.push_back(GatherItem{dp, d, bytes});   // fetched by flush_gather's one launch
        e->stats.h2d_bytes += (long long)bytes;
        e->had_uploads = true;
        sf.seen.emplace_back(host, d);
        *dev = d;
        return ROFT_OK;
    }
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
                e->feat_frames |= 1u << t;
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
    e->feat_frames = 0;
    e->feat_dep_in_batch = false;
    e->new_mask_frames = 0;
    e->gather.clear();
    int rc = submit_frames(e, inputs, n_objects, n_frames);
    if (rc == ROFT_OK) rc = flush_gather(e);
    e->gather.clear();
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
        const std::string msg = last_error();
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



