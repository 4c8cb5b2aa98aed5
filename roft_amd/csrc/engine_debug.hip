// engine_debug.hip -- roft_debug_*: diagnostics and experiments (include/roft_engine.h section 4).
#include "engine_internal.h"


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

