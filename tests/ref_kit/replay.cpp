// replay.cpp -- conformance kit: the committed oracle vectors (tests/golden/oracle_vectors/*.txt) replayed through the REAL
// classes of hsp-iit/roft v1.2.1 and robotology/bayes-filters-lib, differences printed.
//
// THIS FILE CANNOT BE BUILT IN THIS REPOSITORY'S CONTAINER (Eigen3, BayesFilters, RobotsIO are absent and there is no network):
// it has never been compiled by its author.  It exists so that somebody who has a working build of the reference -- e.g. inside
// the reference's own dockerfiles/Dockerfile -- can close the one gap this repository cannot close by itself: its CPU oracle
// (oracle/*.c) restates bfl's unscented transform from a recollection of its published algorithm, and
// every parity claim of the HIP path is a claim against that oracle.  If the numbers below agree to ~1e-12, the oracle -- and
// with it the HIP path, tests/test_parity_gpu.py::test_hip_operators_against_the_committed_conformance_vectors -- is pinned.
//
// Build (inside an environment where the reference builds), from the repository root:
//   g++ -std=c++17 -O2 tests/ref_kit/replay.cpp -I<roft>/src/roft-lib/include $(pkg-config --cflags eigen3) \
//       -lROFT -lBayesFilters -lRobotsIO -o replay && ./replay tests/golden/oracle_vectors
// What is replayed (reference file:line of what each case pins):
//   ut_weights_*           bfl::sigma_point::UTWeight(n, alpha, beta, kappa)                  UKFCorrection.cpp:28-33
//   sigma_points_*         bfl::sigma_point::sigma_point(GaussianMixture, c)                  UKFCorrection.cpp:70-76
//   quaternion_*           bfl::utils::sum_quaternion_rotation_vector / diff_quaternion       CartesianQuaternionMeasurement.cpp:377,459
//   ukf_predict_*          bfl::UKFPrediction over ROFT::CartesianQuaternionModel             CartesianQuaternionModel.cpp:86-141
//   ukf_correct_*          ROFT::UKFCorrection over ROFT::CartesianQuaternionMeasurement      UKFCorrection.cpp:54-133, ...Measurement.cpp:357-487
//   skf_correct_*          ROFT::SKFCorrection over a linear model with the recorded H        SKFCorrection.cpp:37-153
// (flow_measurement_* / mask_propagate_* need cv::Mat sources; their reference statements are integer / index arithmetic and are
//  restated line by line in oracle/ro_velocity.c and oracle/ro_mask.c -- a reader with OpenCV can feed the .txt images to
//  ImageOpticalFlowMeasurement<cv::Vec2f> / ImageSegmentationOFAidedSource the same way.)
#include <BayesFilters/Gaussian.h>
#include <BayesFilters/LinearMeasurementModel.h>
#include <BayesFilters/UKFPrediction.h>
#include <BayesFilters/sigma_point.h>
#include <BayesFilters/utils.h>
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

using namespace Eigen;
using Case = std::map<std::string, MatrixXd>;

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
static double worst = 0.0;
static void report(const std::string& what, const MatrixXd& got, const MatrixXd& want)
{
    const double d = (got - want).cwiseAbs().maxCoeff();
    worst = std::max(worst, d);
    std::printf("%-44s max |reference - oracle| = %.3e %s\n", what.c_str(), d, d < 1e-9 ? "" : "   <-- DIFFERS");
}
// belief of the pose filter: Gaussian(9 linear, 1 circular, quaternion) -- mean 13, covariance 12 x 12 (ROFTFilter.cpp:64-67)
static bfl::Gaussian belief(const Case& c)
{
    bfl::Gaussian g(9, 1, true);
    g.mean() = c.at("mean").transpose();
    g.covariance() = c.at("P");
    return g;
}
// a pose source that always has the recorded pose, a velocity source fed by hand (what ROFTFilter.cpp:305 does every frame)
struct FixedPose : RobotsIO::Utils::Transform {
    Eigen::Transform<double, 3, Affine> T = Eigen::Transform<double, 3, Affine>::Identity();
    Eigen::Transform<double, 3, Affine> transform() override { return T; }
    bool freeze(const bool = false) override { return true; }
};
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
    for (const char* a : {"1", "0.5"})
        for (int n : {18, 21, 24}) {
            const Case c = load(dir + "/ut_weights_n" + std::to_string(n) + "_a" + a + ".txt");
            bfl::sigma_point::UTWeight w(n, c.at("ut")(0), c.at("ut")(1), c.at("ut")(2));
            MatrixXd got(1, 4);
            got << w.c, w.mean(0), w.covariance(0), w.mean(1);
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
        report("sum_quaternion_rotation_vector", bfl::utils::sum_quaternion_rotation_vector(q, r).transpose(), c.at("q_boxplus_r"));
        report("diff_quaternion", bfl::utils::diff_quaternion(q, qb.col(0)).transpose(), c.at("diff_q_qb"));
    }
    for (int i = 0; i < 3; ++i) {
        const Case c = load(dir + "/ukf_predict_" + std::to_string(i) + ".txt");
        // kinematic model as src/roft/src/main.cpp:311-313 packs it: head<3> = sigma of the angular velocity, tail<3> = PSD of the
        // linear acceleration (consumed ROFTFilter.cpp:89-90)
        auto model = std::unique_ptr<ROFT::CartesianQuaternionModel>(new ROFT::CartesianQuaternionModel(
            c.at("psd_lin_acc").transpose(), c.at("sigma_ang_vel").transpose(), c.at("T")(0)));
        bfl::UKFPrediction pred(std::move(model), c.at("ut")(0), c.at("ut")(1), c.at("ut")(2));
        bfl::Gaussian in = belief(c), out(9, 1, true);
        pred.predict(in, out);
        report("ukf_predict_" + std::to_string(i) + " mean", out.mean().transpose(), c.at("mean_out"));
        report("ukf_predict_" + std::to_string(i) + " covariance", out.covariance(), c.at("P_out"));
    }
    for (const char* name : {"ukf_correct_velocity_0", "ukf_correct_velocity_1", "ukf_correct_pose_0", "ukf_correct_pose_1",
                             "ukf_correct_pose_velocity_0", "ukf_correct_pose_velocity_1", "ukf_correct_pose_beyond_pi"}) {
        const Case c = load(dir + std::string("/") + name + ".txt");
        const int type = (int)c.at("type")(0);   // 1 velocity, 2 pose, 3 pose + velocity; measurement order [v w | x q(w x y z)]
        const bool has_vel = type & 1, has_pose = type & 2;
        const MatrixXd meas = c.at("meas"), Rd = c.at("Rdiag");
        auto pose = std::make_shared<FixedPose>();
        auto vel = std::make_shared<RobotsIO::Utils::SpatialVelocityBuffer>();
        int k = 0;
        if (has_vel) { vel->set_twist(meas.block(0, 0, 1, 3).transpose(), meas.block(0, 3, 1, 3).transpose()); k = 6; }
        if (has_pose) {
            pose->T = Translation3d(meas(0, k), meas(0, k + 1), meas(0, k + 2)) * Quaterniond(meas(0, k + 3), meas(0, k + 4), meas(0, k + 5), meas(0, k + 6));
        }
        // sigmas in the order of ROFTFilter.cpp:96-99 / main.cpp:319-323; Rdiag is [R_v R_w | R_x R_q]
        const int o = has_vel ? 6 : 0;
        Vector3d s_v = Vector3d::Ones(), s_w = Vector3d::Ones(), s_x = Vector3d::Ones(), s_q = Vector3d::Ones();
        if (has_vel) { s_v = Rd.block(0, 0, 1, 3).transpose(); s_w = Rd.block(0, 3, 1, 3).transpose(); }
        if (has_pose) { s_x = Rd.block(0, o, 1, 3).transpose(); s_q = Rd.block(0, o + 3, 1, 3).transpose(); }
        auto model = std::unique_ptr<ROFT::CartesianQuaternionMeasurement>(new ROFT::CartesianQuaternionMeasurement(
            pose, vel, /* use_screw_velocity */ false, has_pose, has_vel, s_x.asDiagonal(), s_q.asDiagonal(), s_v.asDiagonal(), s_w.asDiagonal(), false));
        ROFT::UKFCorrection corr(std::move(model), c.at("ut")(0), c.at("ut")(1), c.at("ut")(2));
        corr.getMeasurementModel().freeze();
        bfl::Gaussian in = belief(c), out(9, 1, true);
        corr.correct(in, out);
        report(std::string(name) + " mean", out.mean().transpose(), c.at("mean_out"));
        report(std::string(name) + " covariance", out.covariance(), c.at("P_out"));
    }
    for (int rw = 0; rw < 2; ++rw) {
        const Case c = load(dir + "/skf_correct_reweight" + std::to_string(rw) + ".txt");
        auto m = std::unique_ptr<RecordedLinear>(new RecordedLinear());
        m->H = c.at("H");
        m->y = c.at("y").transpose();
        m->R = Vector2d(c.at("Rdiag")(0), c.at("Rdiag")(1)).asDiagonal();
        ROFT::SKFCorrection corr(std::move(m), 2, rw != 0);
        bfl::Gaussian in(6), out(6);
        in.mean() = c.at("x_pred").transpose();
        in.covariance() = c.at("P_pred");
        corr.correct(in, out);
        report("skf_correct reweight=" + std::to_string(rw) + " mean", out.mean().transpose(), c.at("x_out"));
        report("skf_correct reweight=" + std::to_string(rw) + " covariance", out.covariance(), c.at("P_out"));
    }
    std::printf("worst difference %.3e -- %s\n", worst, worst < 1e-9 ? "the oracle restates the reference" : "see the lines marked above");
    return worst < 1e-9 ? 0 : 1;
}
