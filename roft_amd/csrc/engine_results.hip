// engine_results.hip -- readers: batch trace, state, outputs, the per-frame log, masks, kernel timing.
#include "engine_internal.h"

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


