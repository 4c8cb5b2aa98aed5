// flow_producer.hip -- host side of the optical-flow producer entry points of include/roft_engine.h (section 3).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/roft_engine.h"
#include "opticalflow.h"

using namespace roft;

extern "C" const char* roft_last_error_string(void);
namespace roft { int set_last_error(int code, const std::string& msg); }

#define OF_TRY(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess) return roft::set_last_error(ROFT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

struct roft_flow_producer {
    int W = 0, H = 0, max_pairs = 0, out_type = 0, device = 0;
    roft_of_params prm{};
    hipStream_t stream = nullptr;
    OfArgs args{};
    float* pyr = nullptr;
    float* coarse = nullptr;
    float* field = nullptr;             // [max_pairs][H*W*2] level-0 fields when the product is CV_16SC2
    void** d_ptrs = nullptr;            // device copy of the pointer tables: prev | cur | out_f32 | out_s16
    void** h_ptrs = nullptr;            // pinned staging of the same
};

extern "C" {

int roft_default_of_params(roft_of_params* p)
{
    if (!p) return roft::set_last_error(ROFT_ERR_INVALID, "null params");
    p->levels = 3;
    p->radius = 3;
    p->iterations = 3;
    p->det_min = 100.0f;
    return ROFT_OK;
}

int roft_flow_producer_create(int W, int H, int max_pairs, const roft_of_params* p, int out_type, int device,
                              roft_flow_producer** out)
{
    if (!p || !out || max_pairs <= 0) return roft::set_last_error(ROFT_ERR_INVALID, "bad argument");
    if (p->levels < 1 || p->levels > 6 || p->radius < 1 || p->radius > 7 || p->iterations < 0)
        return roft::set_last_error(ROFT_ERR_INVALID, "levels 1..6, radius 1..7");
    if (W <= 0 || H <= 0 || (W % (4 << (p->levels - 1))) || (H % (1 << (p->levels - 1))) || (W % 4) || (H % 4))
        return roft::set_last_error(ROFT_ERR_INVALID, "width must be a multiple of 4 * 2^(levels-1), height of 2^(levels-1) and of 4");
    if (out_type != ROFT_FLOW_F32C2 && out_type != ROFT_FLOW_S16C2) return roft::set_last_error(ROFT_ERR_INVALID, "bad out_type");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device)
        return roft::set_last_error(ROFT_ERR_DEVICE, "no HIP device (the flow producer has no CPU path)");
    OF_TRY(hipSetDevice(device));
    roft_flow_producer* fp = new roft_flow_producer();
    fp->W = W; fp->H = H; fp->max_pairs = max_pairs; fp->out_type = out_type; fp->device = device; fp->prm = *p;
    OF_TRY(hipStreamCreateWithFlags(&fp->stream, hipStreamNonBlocking));
    OfArgs& a = fp->args;
    a.levels = p->levels; a.radius = p->radius; a.iterations = p->iterations; a.det_min = p->det_min;
    size_t off = 0, foff = 0;
    for (int l = 0; l < p->levels; ++l) {
        a.lv[l].w = W >> l; a.lv[l].h = H >> l; a.lv[l].off = off;
        off += (size_t)a.lv[l].w * a.lv[l].h;
        off = (off + 3) & ~(size_t)3;
        a.flow_off[l] = foff;
        if (l >= 1) foff += 2 * (size_t)a.lv[l].w * a.lv[l].h;
    }
    a.pyr_stride = off;
    a.flow_stride = std::max<size_t>(foff, 2);
    OF_TRY(hipMalloc(reinterpret_cast<void**>(&fp->pyr), sizeof(float) * 2 * a.pyr_stride * max_pairs));
    OF_TRY(hipMalloc(reinterpret_cast<void**>(&fp->coarse), sizeof(float) * a.flow_stride * max_pairs));
    if (out_type == ROFT_FLOW_S16C2) OF_TRY(hipMalloc(reinterpret_cast<void**>(&fp->field), sizeof(float) * 2 * W * H * max_pairs));
    OF_TRY(hipMalloc(reinterpret_cast<void**>(&fp->d_ptrs), sizeof(void*) * 4 * max_pairs));
    OF_TRY(hipHostMalloc(reinterpret_cast<void**>(&fp->h_ptrs), sizeof(void*) * 4 * max_pairs));
    a.pyr = fp->pyr; a.coarse = fp->coarse;
    *out = fp;
    return ROFT_OK;
}

int roft_flow_producer_destroy(roft_flow_producer* fp)
{
    if (!fp) return ROFT_OK;
    (void)hipSetDevice(fp->device);
    if (fp->stream) (void)hipStreamSynchronize(fp->stream);
    if (fp->pyr) (void)hipFree(fp->pyr);
    if (fp->coarse) (void)hipFree(fp->coarse);
    if (fp->field) (void)hipFree(fp->field);
    if (fp->d_ptrs) (void)hipFree(fp->d_ptrs);
    if (fp->h_ptrs) (void)hipHostFree(fp->h_ptrs);
    if (fp->stream) (void)hipStreamDestroy(fp->stream);
    delete fp;
    return ROFT_OK;
}

int roft_flow_producer_run(roft_flow_producer* fp, const uint8_t* const* prev, const uint8_t* const* cur, void* const* out,
                           int n)
{
    if (!fp || !prev || !cur || !out || n <= 0 || n > fp->max_pairs) return roft::set_last_error(ROFT_ERR_INVALID, "bad argument");
    OF_TRY(hipSetDevice(fp->device));
    OF_TRY(hipStreamSynchronize(fp->stream));   // the pinned pointer table is reused
    const int m = fp->max_pairs;
    for (int i = 0; i < n; ++i) {
        if (!prev[i] || !cur[i] || !out[i] || (reinterpret_cast<uintptr_t>(prev[i]) & 3) || (reinterpret_cast<uintptr_t>(cur[i]) & 3) ||
            (reinterpret_cast<uintptr_t>(out[i]) & 7))
            return roft::set_last_error(ROFT_ERR_INVALID, "null or misaligned image / flow pointer (images 4 B, flow 8 B)");
        fp->h_ptrs[i] = const_cast<uint8_t*>(prev[i]);
        fp->h_ptrs[m + i] = const_cast<uint8_t*>(cur[i]);
        fp->h_ptrs[2 * m + i] = (fp->out_type == ROFT_FLOW_F32C2) ? out[i] : (void*)(fp->field + (size_t)i * 2 * fp->W * fp->H);
        fp->h_ptrs[3 * m + i] = out[i];
    }
    OF_TRY(hipMemcpyAsync(fp->d_ptrs, fp->h_ptrs, sizeof(void*) * 4 * m, hipMemcpyHostToDevice, fp->stream));
    OfArgs a = fp->args;
    a.n = n;
    a.prev = reinterpret_cast<const uint8_t* const*>(fp->d_ptrs);
    a.cur = reinterpret_cast<const uint8_t* const*>(fp->d_ptrs + m);
    a.out_f32 = reinterpret_cast<float* const*>(fp->d_ptrs + 2 * m);
    launch_optical_flow(a, fp->stream);
    if (fp->out_type == ROFT_FLOW_S16C2)
        launch_flow_quantise(reinterpret_cast<const float* const*>(fp->d_ptrs + 2 * m),
                             reinterpret_cast<int16_t* const*>(fp->d_ptrs + 3 * m), n, fp->W, fp->H, fp->stream);
    OF_TRY(hipGetLastError());
    return ROFT_OK;
}

int roft_flow_producer_sync(roft_flow_producer* fp)
{
    if (!fp) return roft::set_last_error(ROFT_ERR_INVALID, "null producer");
    OF_TRY(hipSetDevice(fp->device));
    OF_TRY(hipStreamSynchronize(fp->stream));
    return ROFT_OK;
}

void* roft_flow_producer_stream(roft_flow_producer* fp) { return fp ? (void*)fp->stream : nullptr; }

int roft_optical_flow(const uint8_t* prev, const uint8_t* cur, int W, int H, const roft_of_params* p, int out_type, void* flow_out)
{
    if (!prev || !cur || !p || !flow_out) return roft::set_last_error(ROFT_ERR_INVALID, "null argument");
    roft_flow_producer* fp = nullptr;
    if (int rc = roft_flow_producer_create(W, H, 1, p, out_type, 0, &fp)) return rc;
    const size_t npix = (size_t)W * H;
    const size_t obytes = (out_type == ROFT_FLOW_F32C2) ? npix * 8 : (npix / 16) * 4;
    uint8_t *d0 = nullptr, *d1 = nullptr;
    void* dout = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d0), npix);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d1), npix);
    if (e == hipSuccess) e = hipMalloc(&dout, obytes);
    if (e == hipSuccess) e = hipMemcpy(d0, prev, npix, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d1, cur, npix, hipMemcpyHostToDevice);
    int rc = ROFT_OK;
    if (e == hipSuccess) {
        const uint8_t* pp[1] = {d0};
        const uint8_t* cc[1] = {d1};
        void* oo[1] = {dout};
        rc = roft_flow_producer_run(fp, pp, cc, oo, 1);
        if (rc == ROFT_OK) rc = roft_flow_producer_sync(fp);
        if (rc == ROFT_OK) e = hipMemcpy(flow_out, dout, obytes, hipMemcpyDeviceToHost);
    }
    if (d0) (void)hipFree(d0);
    if (d1) (void)hipFree(d1);
    if (dout) (void)hipFree(dout);
    roft_flow_producer_destroy(fp);
    if (e != hipSuccess) return roft::set_last_error(ROFT_ERR_DEVICE, hipGetErrorString(e));
    return rc;
}

}  // extern "C"
