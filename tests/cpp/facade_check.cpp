// Exercises the C++ facade (include/ROFT/Filters.h) over the C ABI.  Reads a tiny binary problem written
// by tests/test_facade.py, runs KF predict + SKF correct + UKF predict through the facade classes and
// writes the results back.  Without a HIP device every call must throw std::runtime_error.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "ROFT/Filters.h"

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int n = 0;
    if (std::fread(&n, sizeof(int), 1, f) != 1) return 2;
    std::vector<double> y(2 * n), H(12 * n);
    double x[6], P[36], q[6], pm[13], pP[144];
    bool ok = std::fread(x, 8, 6, f) == 6 && std::fread(P, 8, 36, f) == 36 && std::fread(q, 8, 6, f) == 6 &&
              std::fread(y.data(), 8, 2 * n, f) == (size_t)2 * n && std::fread(H.data(), 8, 12 * n, f) == (size_t)12 * n &&
              std::fread(pm, 8, 13, f) == 13 && std::fread(pP, 8, 144, f) == 144;
    std::fclose(f);
    if (!ok) return 2;
    try {
        ROFT::Gaussian prev(6, 0, false), pred(6, 0, false), corr(6, 0, false);
        std::memcpy(prev.mean().data(), x, sizeof(x));
        std::memcpy(prev.covariance().data(), P, sizeof(P));
        ROFT::KFPrediction kf(q, q + 3);
        kf.predict(prev, pred);
        int status = 0;
        double r[2] = {1.0, 1.0};
        if (roft_skf_correct(pred.mean().data(), pred.covariance().data(), n, y.data(), H.data(), r, 1, corr.mean().data(),
                             corr.covariance().data(), &status) != ROFT_OK)
            throw std::runtime_error(roft_last_error_string());
        ROFT::Gaussian pp(9, 1, true), pq(9, 1, true);
        std::memcpy(pp.mean().data(), pm, sizeof(pm));
        std::memcpy(pp.covariance().data(), pP, sizeof(pP));
        const double one[3] = {1.0, 1.0, 1.0};
        ROFT::UKFPrediction up(one, one, 1.0 / 30.0, 1.0, 2.0, 0.0);
        up.predict(pp, pq);
        // optical-flow source facade: a smooth pattern shifted by (2, 1) pixels
        const int W = 64, Hh = 48;
        std::vector<std::uint8_t> g0(W * Hh), g1(W * Hh);
        for (int yy = 0; yy < Hh; ++yy)
            for (int xx = 0; xx < W; ++xx) {
                g0[yy * W + xx] = (std::uint8_t)(128 + (int)(60.0 * std::sin(0.35 * xx) * std::cos(0.27 * yy)));
                g1[yy * W + xx] = (std::uint8_t)(128 + (int)(60.0 * std::sin(0.35 * (xx - 2)) * std::cos(0.27 * (yy - 1))));
            }
        ROFT::ImageOpticalFlowNVOF of(W, Hh, ROFT::ImageOpticalFlowHIP::Product::NVOF_2_0);
        of.parameters().levels = 2;
        of.parameters().det_min = 1.0f;
        const bool first = of.step_frame(g0.data());
        const bool second = of.step_frame(g1.data());
        if (first || !second || !of.flow().first || of.get_matrix_type() != 13 || of.get_grid_size() != 1) return 4;
        FILE* o = std::fopen(argv[2], "wb");
        std::fwrite(corr.mean().data(), 8, 6, o);
        std::fwrite(corr.covariance().data(), 8, 36, o);
        std::fwrite(pq.mean().data(), 8, 13, o);
        std::fwrite(pq.covariance().data(), 8, 144, o);
        std::fwrite(of.flow().second, 4, (size_t)W * Hh * 2, o);
        std::fclose(o);
    } catch (const std::runtime_error& e) {
        std::printf("runtime_error: %s\n", e.what());
        return 3;
    }
    return 0;
}
