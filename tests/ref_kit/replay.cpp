// replay.cpp -- conformance kit: the committed oracle vectors (tests/golden/oracle_vectors/*.txt) replayed through the filter
// classes of hsp-iit/roft v1.2.1 and robotology/bayes-filters-lib, differences printed.
//
// ONE source, TWO builds:
//
//  (1) against the REAL libraries (somewhere the reference builds, e.g. its own dockerfiles/Dockerfile) -- the build that PINS the
//      oracle.  This repository's CPU oracle (oracle/*.c) restates bfl's unscented transform from a recollection of its published
//      algorithm; every parity claim of the HIP path is a claim against that oracle.  If the numbers below
//      agree to ~1e-12, the oracle -- and with it the HIP path -- is pinned.  NOBODY HAS RUN THIS BUILD: Eigen3, BayesFilters and
//      RobotsIO are absent from this repository's container and there is no network.
//        g++ -std=c++17 -O2 tests/ref_kit/replay.cpp -I<roft>/src/roft-lib/include $(pkg-config --cflags eigen3)
//            -lROFT -lBayesFilters -lRobotsIO -o replay && ./replay tests/golden/oracle_vectors
//
//  (2) -DROFT_KIT_FACADE, against THIS repository's drop-in headers (include/ROFT: the reference's class names and constructor
//      signatures over the C ABI; include/compat: stand-ins for the third-party types) and libroft_hip.so -- built by
//      __graft_entry__.build(), run on the GPU by tests/test_facade.py::test_conformance_kit_replays_the_vectors_through_the_class_api.
//      THIS BUILD PINS NOTHING: it replays the oracle's own vectors through the reference's class API over the HIP engine, i.e. it
//      is a third consumer of the vectors (after the oracle itself and the HIP operators through the C ABI).  What it proves: the
//      kit compiles, its vector I/O works, its use of the class API (constructor argument order, measurement order, freeze /
//      correct protocol) is the one the facade -- written against the reference's headers -- accepts; whoever has real bfl only
//      changes the include and link lines.  The cases that call bfl FREE functions (UTWeight, sigma_point, *quaternion*) have no
//      counterpart in a class API and are compiled in build (1) only.
//
// What is replayed (reference file:line of what each case pins):
//   ut_weights_*           bfl::sigma_point::UTWeight(n, alpha, beta, kappa)                  UKFCorrection.cpp:28-33        (1)
//   sigma_points_*         bfl::sigma_point::sigma_point(GaussianMixture, c)                  UKFCorrection.cpp:70-76        (1)
//   quaternion_*           bfl::utils::sum_quaternion_rotation_vector / diff_quaternion       CartesianQuaternionMeasurement.cpp:377,459  (1)
//   ukf_predict_*          bfl::UKFPrediction over ROFT::CartesianQuaternionModel             CartesianQuaternionModel.cpp:86-141          (1)(2)
//   ukf_correct_*          ROFT::UKFCorrection over ROFT::CartesianQuaternionMeasurement      UKFCorrection.cpp:54-133, ...Measurement.cpp:357-487  (1)(2)
//   skf_correct_*          ROFT::SKFCorrection over a linear model with the recorded H        SKFCorrection.cpp:37-153       (1)(2)
// (flow_measurement_* / mask_propagate_* need cv::Mat sources; their reference statements are integer / index arithmetic and are
//  restated line by line in oracle/ro_velocity.c and oracle/ro_mask.c -- a reader with OpenCV can feed the .txt images to
//  ImageOpticalFlowMeasurement<cv::Vec2f> / ImageSegmentationOFAidedSource the same way.)
// Bars: build (1) 1e-9 everywhere (agreement is expected at ~1e-12).  Build (2): the bars of tests/test_parity_gpu.py for the HIP
// operators -- UKF 1e-9 absolute (1e-6 for the case whose sigma rotations pass pi), SKF 1e-8 relative (information form against the
// sequential recursion).
#include <BayesFilters/Gaussian.h>
#include <BayesFilters/LinearMeasurementModel.h>
#include <BayesFilters/UKFPrediction.h>
#ifndef ROFT_KIT_FACADE
#include <BayesFilters/sigma_point.h>
#include <BayesFilters/utils.h>
#endif
#include <ROFT/CartesianQuaternionMeasurement.h>
#include <ROFT/CartesianQuaternionModel.h>
#include <ROFT/SKFCorrection.h>
#include <ROFT/UKFCorrection.h>
#include <RobotsIO/Utils/SpatialVelocityBuffer.h>
#include <RobotsIO/Utils/Transform.h>

#include <Eigen/Dense>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>

#include <cmath>
#include <memory>
#include <string>

using namespace Eigen;
using Case = std::map<std::string, MatrixXd>;
#ifdef ROFT_KIT_FACADE
static const bool kFacade = true;
#else
static const bool kFacade = false;
#endif

static Case load(const std::string& path)   // `key rows cols` then rows x cols values, row-major
{
    Case c;
    std::ifstream in(path);
    std::string key;
    long r, k;
    while (in >> key >> r >> k) {
        MatrixXd m(r, k);
        for (long i = 0; i < r; ++i)
            for (long j = 0; j < k; ++j) { std::string t; in >> t; m(i, j) = std::strtod(t.c_str(), nullptr); }
        c[key] = m;
    }
    if (c.empty()) { std::cerr << "cannot read " << path << std::endl; std::exit(2); }
    return c;
}
static double worst = 0.0;   // worst difference relative to its bar (<= 1 passes)
// (coefficient access only: the stand-in matrix type of build (2) has no expression templates)
static void report(const std::string& what, const MatrixXd& got, const MatrixXd& want, double atol = 1e-9, double rtol = 0.0)
{
    double d = 0.0, rel = 0.0;
    bool shape = got.rows() == want.rows() && got.cols() == want.cols();
    if (!shape && got.rows() == want.cols() && got.cols() == want.rows() && (got.rows() == 1 || got.cols() == 1)) shape = true;   // a vector either way up
    if (!shape) { std::printf("%-44s shape %ldx%ld, expected %ldx%ld   <-- DIFFERS\n", what.c_str(), (long)got.rows(), (long)got.cols(), (long)want.rows(), (long)want.cols()); worst = 1e300; return; }
    const bool flip = got.rows() != want.rows();
    for (long i = 0; i < (long)want.rows(); ++i)
        for (long j = 0; j < (long)want.cols(); ++j) {
            const double g = flip ? got(j, i) : got(i, j), w = want(i, j), e = std::fabs(g - w);
            d = std::max(d, e);
            rel = std::max(rel, e / (atol + rtol * std::fabs(w)));
        }
    worst = std::max(worst, rel);
    std::printf("%-44s max |replayed - oracle| = %.3e %s\n", what.c_str(), d, rel <= 1.0 ? "" : "   <-- DIFFERS");
}
static MatrixXd col(const MatrixXd& row_or_col)   // the n values of a 1 x n or n x 1 matrix as a column
{
    const long n = (long)(row_or_col.rows() * row_or_col.cols());
    MatrixXd v(n, 1);
    for (long i = 0; i < n; ++i) v(i, 0) = row_or_col.rows() == 1 ? row_or_col(0, i) : row_or_col(i, 0);
    return v;
}
static MatrixXd diag3(const MatrixXd& row, long first)   // diag(row[first .. first + 3))
{
    MatrixXd m(3, 3);
    for (long i = 0; i < 3; ++i)
        for (long j = 0; j < 3; ++j) m(i, j) = (i == j) ? row(0, first + i) : 0.0;
    return m;
}
static MatrixXd ones3()
{
    MatrixXd m(3, 3);
    for (long i = 0; i < 3; ++i)
        for (long j = 0; j < 3; ++j) m(i, j) = (i == j) ? 1.0 : 0.0;
    return m;
}
// belief of the pose filter: Gaussian(9 linear, 1 circular, quaternion) -- mean 13, covariance 12 x 12 (ROFTFilter.cpp:64-67)
static bfl::Gaussian belief(const Case& c)
{
    bfl::Gaussian g(9, 1, true);
    g.mean() = col(c.at("mean"));
    g.covariance() = c.at("P");
    return g;
}
// a pose source that always has the recorded pose, a velocity source fed by hand (what ROFTFilter.cpp:305 does every frame)
struct FixedPose : RobotsIO::Utils::Transform {
    Eigen::Transform<double, 3, Affine> T;
    Eigen::Transform<double, 3, Affine> transform() override { return T; }
    bool freeze(const bool = false) override { return true; }
    void set(const double x[3], const double q_wxyz[4])
    {
#ifdef ROFT_KIT_FACADE
        for (int i = 0; i < 3; ++i) T.translation()[i] = x[i];
        for (int i = 0; i < 4; ++i) T.quaternion()[i] = q_wxyz[i];
#else
        T = Translation3d(x[0], x[1], x[2]) * Quaterniond(q_wxyz[0], q_wxyz[1], q_wxyz[2], q_wxyz[3]);
#endif
    }
};
static void set_twist(RobotsIO::Utils::SpatialVelocityBuffer& b, const double v[3], const double w[3])
{
#ifdef ROFT_KIT_FACADE
    b.set_twist(v, w);
#else
    b.set_twist(Vector3d(v[0], v[1], v[2]), Vector3d(w[0], w[1], w[2]));
#endif
}
// the velocity filter's measurement model with the recorded H and y (ImageOpticalFlowMeasurement plays this role)
struct RecordedLinear : bfl::LinearMeasurementModel {
    MatrixXd H, y, R;
    std::pair<bool, bfl::Data> measure(const bfl::Data& = bfl::Data()) const override { return {true, y}; }
    bool freeze(const bfl::Data& = bfl::Data()) override { return true; }
    std::pair<bool, MatrixXd> getNoiseCovarianceMatrix() const override { return {true, R}; }
    MatrixXd getMeasurementMatrix() const override { return H; }
    bfl::VectorDescription getInputDescription() const override { return bfl::VectorDescription(6); }
    bfl::VectorDescription getMeasurementDescription() const override { return bfl::VectorDescription(y.rows()); }
};

int main(int argc, char** argv)
{
    const std::string dir = argc > 1 ? argv[1] : "tests/golden/oracle_vectors";
#ifndef ROFT_KIT_FACADE   // bfl free functions: build (1) only
    for (const char* a : {"1", "0.5"})
        for (int n : {18, 21, 24}) {
            const Case c = load(dir + "/ut_weights_n" + std::to_string(n) + "_a" + a + ".txt");
            bfl::sigma_point::UTWeight w(n, c.at("ut")(0), c.at("ut")(1), c.at("ut")(2));
            MatrixXd got(1, 4);
            got(0, 0) = w.c; got(0, 1) = w.mean(0); got(0, 2) = w.covariance(0); got(0, 3) = w.mean(1);
            report("ut_weights n=" + std::to_string(n) + " alpha=" + a, got, c.at("c_wm0_wc0_wi"));
        }
    for (const char* name : {"sigma_points_process_noise", "sigma_points_velocity_noise"}) {
        const Case c = load(dir + "/" + name + ".txt");
        bfl::Gaussian g = belief(c);
        g.augmentWithNoise(c.at("noise"));
        const int n = 12 + (int)c.at("noise").rows();
        bfl::sigma_point::UTWeight w(n, c.at("ut")(0), c.at("ut")(1), c.at("ut")(2));
        report(name, bfl::sigma_point::sigma_point(g, w.c), c.at("sigma"));
    }
    {
        const Case c = load(dir + "/quaternion_sum_and_difference.txt");
        const MatrixXd q = c.at("q").transpose(), r = c.at("r").transpose(), qb = c.at("q_b").transpose();
        report("sum_quaternion_rotation_vector", MatrixXd(bfl::utils::sum_quaternion_rotation_vector(q, r).transpose()), c.at("q_boxplus_r"));
        report("diff_quaternion", MatrixXd(bfl::utils::diff_quaternion(q, qb.col(0)).transpose()), c.at("diff_q_qb"));
    }
#endif
    for (int i = 0; i < 3; ++i) {
        const Case c = load(dir + "/ukf_predict_" + std::to_string(i) + ".txt");
        // kinematic model as src/roft/src/main.cpp:311-313 packs it: head<3> = sigma of the angular velocity, tail<3> = PSD of the
        // linear acceleration (consumed ROFTFilter.cpp:89-90)
        auto model = std::unique_ptr<ROFT::CartesianQuaternionModel>(new ROFT::CartesianQuaternionModel(
            diag3(c.at("psd_lin_acc"), 0), diag3(c.at("sigma_ang_vel"), 0), c.at("T")(0, 0)));   // 3 x 3 diagonals (ROFTFilter.cpp:89-90)
        bfl::UKFPrediction pred(std::move(model), c.at("ut")(0, 0), c.at("ut")(0, 1), c.at("ut")(0, 2));
        bfl::Gaussian in = belief(c), out(9, 1, true);
        pred.predict(in, out);
        report("ukf_predict_" + std::to_string(i) + " mean", out.mean(), c.at("mean_out"));
        report("ukf_predict_" + std::to_string(i) + " covariance", out.covariance(), c.at("P_out"));
    }
    for (const char* name : {"ukf_correct_velocity_0", "ukf_correct_velocity_1", "ukf_correct_pose_0", "ukf_correct_pose_1",
                             "ukf_correct_pose_velocity_0", "ukf_correct_pose_velocity_1", "ukf_correct_pose_beyond_pi"}) {
        const Case c = load(dir + std::string("/") + name + ".txt");
        const int type = (int)c.at("type")(0, 0);   // 1 velocity, 2 pose, 3 pose + velocity; measurement order [v w | x q(w x y z)]
        const bool has_vel = type & 1, has_pose = type & 2;
        const MatrixXd meas = c.at("meas"), Rd = c.at("Rdiag");
        auto pose = std::make_shared<FixedPose>();
        auto vel = std::make_shared<RobotsIO::Utils::SpatialVelocityBuffer>();
        int k = 0;
        if (has_vel) {
            const double v[3] = {meas(0, 0), meas(0, 1), meas(0, 2)}, w[3] = {meas(0, 3), meas(0, 4), meas(0, 5)};
            set_twist(*vel, v, w);
            k = 6;
        }
        {
            const double x0[3] = {0.0, 0.0, 0.0}, q0[4] = {1.0, 0.0, 0.0, 0.0};
            pose->set(x0, q0);
        }
        if (has_pose) {
            const double x[3] = {meas(0, k), meas(0, k + 1), meas(0, k + 2)}, q[4] = {meas(0, k + 3), meas(0, k + 4), meas(0, k + 5), meas(0, k + 6)};
            pose->set(x, q);
        }
        // sigmas in the order of ROFTFilter.cpp:96-99 / main.cpp:319-323; Rdiag is [R_v R_w | R_x R_q]
        const int o = has_vel ? 6 : 0;
        MatrixXd s_v = ones3(), s_w = ones3(), s_x = ones3(), s_q = ones3();
        if (has_vel) { s_v = diag3(Rd, 0); s_w = diag3(Rd, 3); }
        if (has_pose) { s_x = diag3(Rd, o); s_q = diag3(Rd, o + 3); }
        auto model = std::unique_ptr<ROFT::CartesianQuaternionMeasurement>(new ROFT::CartesianQuaternionMeasurement(
            pose, vel, /* use_screw_velocity */ false, has_pose, has_vel, s_x, s_q, s_v, s_w, false));
        ROFT::UKFCorrection corr(std::move(model), c.at("ut")(0, 0), c.at("ut")(0, 1), c.at("ut")(0, 2));
        corr.getMeasurementModel().freeze();
        bfl::Gaussian in = belief(c), out(9, 1, true);
        corr.correct(in, out);
        // (the case whose sigma rotations pass pi: 1e-6 in build (2), tests/test_parity_gpu.py::test_ukf_correct_sigma_rotations_beyond_pi)
        const double bar = (kFacade && std::string(name).find("beyond_pi") != std::string::npos) ? 1e-6 : 1e-9;
        report(std::string(name) + " mean", out.mean(), c.at("mean_out"), bar);
        report(std::string(name) + " covariance", out.covariance(), c.at("P_out"), bar);
    }
    for (int rw = 0; rw < 2; ++rw) {
        const Case c = load(dir + "/skf_correct_reweight" + std::to_string(rw) + ".txt");
        auto m = std::unique_ptr<RecordedLinear>(new RecordedLinear());
        m->H = c.at("H");
        m->y = col(c.at("y"));
        m->R = MatrixXd(2, 2);
        m->R(0, 0) = c.at("Rdiag")(0, 0); m->R(0, 1) = 0.0; m->R(1, 0) = 0.0; m->R(1, 1) = c.at("Rdiag")(0, 1);
        ROFT::SKFCorrection corr(std::move(m), 2, rw != 0);
        bfl::Gaussian in(6), out(6);
        in.mean() = col(c.at("x_pred"));
        in.covariance() = c.at("P_pred");
        corr.correct(in, out);
        // build (2): information form against the sequential recursion, 1e-8 relative (tests/test_parity_gpu.py SKF_RTOL)
        report("skf_correct reweight=" + std::to_string(rw) + " mean", out.mean(), c.at("x_out"), kFacade ? 1e-12 : 1e-9, kFacade ? 1e-8 : 0.0);
        report("skf_correct reweight=" + std::to_string(rw) + " covariance", out.covariance(), c.at("P_out"), kFacade ? 1e-14 : 1e-9, kFacade ? 1e-8 : 0.0);
    }
    std::printf("worst difference / bar = %.3e -- %s\n", worst,
                worst > 1.0 ? "see the lines marked above"
                            : (kFacade ? "the class API over the HIP engine reproduces the vectors (this build pins nothing)" : "the oracle restates the reference"));
    return worst <= 1.0 ? 0 : 1;
}
