// engine_step.hip -- roft_step / roft_sync: the launch graph of one batch on the engine's four HIP streams (step_batch: mask frames,
// velocity chain, two pose lanes) and the timing marks around its launch groups.
//
// ORDERING AND PROGRESS -- why no wait in this file, or in a kernel it launches, can hang.  (The arguments used to be spread over
// DESIGN.md section 5 and comments 1 500 lines apart; they are here, next to the code that relies on them.)
//
//  (1) Every cross-stream dependency of a batch is an EVENT recorded behind the producer (the stop event of its last kernel) and
//      waited for by the consumer's stream before the consumer's first kernel: ev_up (HOST copies) -> mask / velocity stream;
//      ev_prep (control blocks + ingest prepared on the upload stream) -> mask stream; ev_mask / ev_part (mask frames) -> velocity
//      stream and the features kernel; ev_skf (velocity filter) | ev_vel (+ the features behind it) -> pose lanes; ev_feat (a feature
//      kernel that ran on the MASK stream: one-frame submits, engines with few objects) -> the lanes whose tests read this batch's
//      sets, the preparation ahead two batches on (it rewrites the control blocks that kernel reads), the host; ev_done[lane] -> host
//      (in-flight bound, roft_sync).  Events only point from work enqueued EARLIER to work enqueued later, batch by batch and chain by chain in the
//      fixed order of step_batch: the wait-for graph is acyclic by construction.
//  (2) The only waits INSIDE kernels are the frame-granular hand-over (a pose lane's step waits for the tag of the twist it needs,
//      k_ukf.hip ukf_one_step; the velocity filter publishes value then tag, k_skf.hip).  A lane kernel is released in one of
//      three ways, each of which guarantees that what it waits for RUNS:
//        (a) behind ev_skf / ev_vel: the velocity filter of the batch has ended -- nothing is waited for in the kernel;
//        (b) `handoff`: behind a hipStreamWaitValue64 on skf_started >= (all velocity-filter workgroups of the batch): every
//            producer workgroup is RESIDENT on a CU when the lane starts, so the lane only waits for workgroups that run;
//        (c) `early_lane`: behind the batch's control blocks only, while the producer may not even be enqueued.  Progress then
//            needs (i) a hardware queue of its own for each of the four chains -- the stream set was PROBED free of conflicts
//            (StreamSet::conflicts == 0), else (c) is off --, (ii) CUs the spinning lanes do not hold: at most one waiting
//            object per eight CUs, all of the lane's workgroups together at most half the device, counted over THIS engine, which
//            is only meaningful while it is the only engine of the process on the device (alone_on_device: a count of stream sets
//            in use, taken at the submit -- never a timing); several PROCESSES on one GPU set ROFT_EARLY_LANES=0.
//      Every in-kernel wait is bounded (two seconds on the device clock): it then raises ROFT_DEV_ERROR_TWIST_WAIT, the step is NOT
//      applied, and the next synchronisation returns ROFT_ERR_DEVICE -- a wrong assumption above costs a batch, not a hang.
//  (3) The outlier test's workgroups that share an alternative (k_render.hip) never wait for each other: each writes its slab,
//      counts itself in and EXITS unless it is the last to arrive; the last one merges.  No co-residency is needed.
//  (4) The mask frames hand over through kernel boundaries only (one launch per frame): no barrier among workgroups in memory.
//  (5) The host blocks in exactly two places: roft_frames_submit on ev_done / ev_vel / ev_feat of batch b - lead (the in-flight bound that
//      sizes every ring), and roft_sync.  Both wait for events of work already enqueued.
#include "engine_internal.h"

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

int step_batch(roft_engine* e)
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
    if (prep && e->batch_counter >= 2) {
        HIP_TRY(hipStreamWaitEvent(sp0, e->ev_mask[(slot + R - 2) % R], 0));
        ++evops;
        // (... and for the feature kernel behind that mask chain, where there was one: it reads the control blocks of its batch)
        if (e->feat_used[(slot + R - 2) % R]) { HIP_TRY(hipStreamWaitEvent(sp0, e->ev_feat[(slot + R - 2) % R], 0)); ++evops; }
    }
    tmark(e, nullptr, prep ? 4 : 0);
    // Bursts and engines with CUs to spare (no preparation ahead): control blocks and the ingest of the delivered masks in ONE
    // launch -- on the mask stream they and the first mask frame were three dependent launches (27 - 35 us in front of the frame).
    static const int fuse_env = getenv("ROFT_CTRL_INGEST") ? atoi(getenv("ROFT_CTRL_INGEST")) : 1;   // (experiments: 0 = two launches)
    bool fused = false;
    if (fuse_env && !prep && e->new_mask_frames && !e->timing) {
        const size_t n16 = sizeof(FrameCtrl) * (size_t)a.n_obj * T / 16;
        fused = launch_ctrl_ingest(e->stage[slot], a, n16, e->new_mask_frames, sp0, (multi && (T == 1 || any_early)) ? e->ev_ctrl[slot] : nullptr);
        if (fused) {
            ++launches;
            CHECK_LAUNCH("FrameCtrl upload + mask ingest");
        }
    }
    if (!fused) {
        const size_t n16 = sizeof(FrameCtrl) * (size_t)a.n_obj * T / 16;
        // Events that complete with a kernel (hipExtLaunchKernelGGL stop events) cost neither the barrier packet nor
        // the host call of a hipEventRecord behind it.
        hipExtLaunchKernelGGL(ctrl_upload_kernel, dim3((unsigned)std::min<size_t>((n16 + 255) / 256, 64)), dim3(256), 0, sp0,
                              nullptr, (multi && (T == 1 || any_early)) ? e->ev_ctrl[slot] : nullptr, 0,
                              reinterpret_cast<const uint4*>(e->stage[slot]), a, n16, 1);
        ++launches;
    }
    CHECK_LAUNCH("FrameCtrl upload");
    if (!fused) {
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
    // ... and so do batches of an engine with MANY CUs to spare (at most one object per sixteen CUs): there a batch is a chain of
    // latencies on every stream and the velocity stream's -- flow measurement, T filter steps, features -- is the longest, not the
    // mask stream's (round 6, 120 steps: 8 objects 3.46e5 -> 3.64e5, 16 objects 6.46e5 -> 6.62e5; 32 objects -4 %, 64 objects -7.5 %).
    const bool feat_on_vel = multi && T > 1 && !(e->feat_mask_mode == 2 || (e->feat_mask_mode == 1 && 16 * a.n_obj <= device_cu_count()));
    const bool want_ev_feat = multi && e->any_feat && !feat_on_vel && (T > 1 || e->any_feat_now);
    // (a feature kernel on the mask stream always ends with ev_feat -- a stop event costs nothing --: wait_batch waits for it, and so
    //  does a preparation ahead that rewrites the control blocks it reads; the LANES wait for it only when they need this batch's sets)
    const bool rec_ev_feat = multi && e->any_feat && !feat_on_vel;
    e->feat_used[slot] = rec_ev_feat;
    if (e->any_feat && !feat_on_vel) {
        launch_features(a, s, (rec_ev_feat && !full) ? e->ev_feat[slot] : nullptr, e->feat_frames);
        ++launches;
        CHECK_LAUNCH("features");
        tmark(e, "features", 0);
        if (rec_ev_feat && full) { HIP_TRY(hipEventRecord(e->ev_feat[slot], s)); ++evops; }
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
    // (with the feature kernel behind it the filter's own stop event is ev_skf: a lane that waits for the batch's twists does not
    //  wait for the features as well -- 39 us at 64 objects --, which its tests read from sets buffered by EARLIER batches; round 6)
    const bool lanes_wait_skf = multi && feat_last && e->lanes_wait_skf != 0 && !e->feat_dep_in_batch && !e->any_feat_now;
    launch_skf_chain(a, e->cfg.flow_weighting, sv, (multi && !full) ? (feat_last ? (lanes_wait_skf ? e->ev_skf[slot] : nullptr) : e->ev_vel[slot]) : nullptr);
    ++launches;
    if (hipError_t le = hipGetLastError()) {
        // the filter's workgroups will never count themselves in: no lane may ever wait for them (a stream-wait on a value has
        // no timeout) -- the hand-over is off for the rest of this engine's life
        e->handoff_mode = 0;
        return fail(ROFT_ERR_DEVICE, std::string("velocity filter chain: ") + hipGetErrorString(le));
    }
    e->skf_total += (unsigned long long)a.n_obj;   // (only once the launch is known to be enqueued: the lanes' gates wait for this count)
    tmark(e, "skf_chain", 2);
    if (lanes_wait_skf && full) { HIP_TRY(hipEventRecord(e->ev_skf[slot], sv)); ++evops; }
    if (feat_last) {
        if (part_gate) { HIP_TRY(hipStreamWaitEvent(sv, e->ev_mask[slot], 0)); ++evops; }   // (the planes of the batch's last frame)
        launch_features(a, sv, !full ? e->ev_vel[slot] : nullptr, e->feat_frames);
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
            HIP_TRY(hipStreamWaitEvent(sp, lanes_wait_skf ? e->ev_skf[slot] : e->ev_vel[slot], 0));
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
    if (e->host_prof) {
        e->hp_batches++;
        // ROFT_HOST_PROF=1+: the host's time in this step, batch by batch (a burst's first batches are not its later ones)
        static const bool per_batch = getenv("ROFT_HOST_PROF") && getenv("ROFT_HOST_PROF")[0] == '1' && getenv("ROFT_HOST_PROF")[1] == '+';
        if (per_batch)
            std::fprintf(stderr, "[roft host batch %d T=%d] ctrl %.1f mask %.1f vel %.1f lanes %.1f us (cumulative)\n", e->batch_counter, T,
                         e->hp_acc[3], e->hp_acc[4], e->hp_acc[5], e->hp_acc[6]);
    }
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
        // ... and the ingest counters of both mask tables: the chain's last kernel, which leaves them zeroed for the batch that
        // uses a table next (ctrl_ingest_kernel adds to them without a reset of its own), may not have run
        if (e->arr.mrec.p) {
            EngineArrays a2 = e->arr.a;
            a2.T = kMaxBatch;
            for (int par = 0; par < 2; ++par) {
                a2.mrec = e->arr.mrec.p + (size_t)par * (kMaxBatch + 1) * a2.n_obj;
                launch_mask_reset(a2, e->stream);
            }
        }
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



