// Instantiates every class of the C++ facade (include/ROFT/) through the reference's constructor signatures and runs each
// through its reference entry point (predict / correct / freeze / step_frame).  Inputs: a small numeric problem and a
// recorded stream written by tests/test_facade.py; the results go back to Python, which computes the same quantities
// through the operator-level C ABI and compares bit for bit.  Without a HIP device every arithmetic call must throw
// std::runtime_error.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "mem_sources.h"

// a linear measurement model serving arrays: SKFCorrection only sees the bfl::LinearMeasurementModel interface
class ArrayMeasurement : public bfl::LinearMeasurementModel {
public:
    ArrayMeasurement(const std::vector<double>& y, const std::vector<double>& H) : y_(y.size(), 1), H_(y.size(), 6)
    {
        std::memcpy(y_.data(), y.data(), 8 * y.size());
        std::memcpy(H_.data(), H.data(), 8 * H.size());
    }
    bool freeze(const bfl::Data& = bfl::Data()) override { return true; }
    std::pair<bool, bfl::Data> measure(const bfl::Data& = bfl::Data()) const override { return {true, bfl::Data(y_)}; }
    std::pair<bool, bfl::Data> predictedMeasure(const Eigen::Ref<const Eigen::MatrixXd>&) const override { return {false, bfl::Data()}; }
    std::pair<bool, bfl::Data> innovation(const bfl::Data&, const bfl::Data&) const override { return {false, bfl::Data()}; }
    Eigen::MatrixXd getMeasurementMatrix() const override { return H_; }
    std::pair<bool, Eigen::MatrixXd> getNoiseCovarianceMatrix() const override { return {true, Eigen::MatrixXd::Identity(2, 2)}; }

private:
    Eigen::MatrixXd y_, H_;
};

class OnePose : public RobotsIO::Utils::Transform {
public:
    explicit OnePose(const double xq[7])
    {
        for (int i = 0; i < 3; ++i) T_.translation()[i] = xq[i];
        for (int i = 0; i < 4; ++i) T_.quaternion()[i] = xq[3 + i];
    }
    bool freeze(const bool = false) override { return true; }
    Eigen::Transform<double, 3, Eigen::Affine> transform() override { return T_; }
    int get_frames_between_iterations() const override { return 6; }

private:
    Eigen::Transform<double, 3, Eigen::Affine> T_;
};

static void put(FILE* o, const Eigen::MatrixXd& m) { std::fwrite(m.data(), 8, m.size(), o); }
static Eigen::MatrixXd diag3(double a) { Eigen::VectorXd v(3, 1); v(0) = v(1) = v(2) = a; return v.asDiagonal(); }

int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int n = 0;
    if (std::fread(&n, sizeof(int), 1, f) != 1) return 2;
    std::vector<double> y(2 * n), H(12 * n);
    double x[6], P[36], q[6], pm[13], pP[144], meas[13];
    bool ok = std::fread(x, 8, 6, f) == 6 && std::fread(P, 8, 36, f) == 36 && std::fread(q, 8, 6, f) == 6 &&
              std::fread(y.data(), 8, 2 * n, f) == (size_t)2 * n && std::fread(H.data(), 8, 12 * n, f) == (size_t)12 * n &&
              std::fread(pm, 8, 13, f) == 13 && std::fread(pP, 8, 144, f) == 144 && std::fread(meas, 8, 13, f) == 13;
    std::fclose(f);
    RecordedStream s;
    if (!ok || !s.load(argv[3])) return 2;
    try {
        FILE* o = std::fopen(argv[2], "wb");
        // ---- velocity filter: bfl::KFPrediction over SpatialVelocityModel, SKFCorrection over a linear measurement model
        bfl::Gaussian prev(6, 0, false), pred(6, 0, false), corr(6, 0, false);
        std::memcpy(prev.mean().data(), x, sizeof(x));
        std::memcpy(prev.covariance().data(), P, sizeof(P));
        Eigen::VectorXd qv(3, 1), qw(3, 1);
        for (int i = 0; i < 3; ++i) { qv(i) = q[i]; qw(i) = q[3 + i]; }
        bfl::KFPrediction v_prediction(std::unique_ptr<bfl::LinearStateModel>(new ROFT::SpatialVelocityModel(qv.asDiagonal(), qw.asDiagonal())));
        v_prediction.predict(prev, pred);
        ROFT::SKFCorrection v_correction(std::unique_ptr<bfl::LinearMeasurementModel>(new ArrayMeasurement(y, H)), 2, true);
        v_correction.correct(pred, corr);
        put(o, corr.mean());
        put(o, corr.covariance());
        // ---- pose filter: bfl::UKFPrediction over CartesianQuaternionModel, UKFCorrection over CartesianQuaternionMeasurement
        bfl::Gaussian pp(9, 1, true), pq(9, 1, true), pc(9, 1, true), pv(9, 1, true);
        std::memcpy(pp.mean().data(), pm, sizeof(pm));
        std::memcpy(pp.covariance().data(), pP, sizeof(pP));
        bfl::UKFPrediction p_prediction(std::unique_ptr<bfl::StateModel>(new ROFT::CartesianQuaternionModel(diag3(1.0), diag3(1.0), 1.0 / 30.0)), 1.0, 2.0, 0.0);
        p_prediction.predict(pp, pq);
        put(o, pq.mean());
        put(o, pq.covariance());
        auto velocity = std::make_shared<RobotsIO::Utils::SpatialVelocityBuffer>();
        velocity->set_twist(meas, meas + 3);
        auto* cqm = new ROFT::CartesianQuaternionMeasurement(std::make_shared<OnePose>(meas + 6), velocity, false, true, true, diag3(1e-3), diag3(1e-4),
                                                             diag3(0.1), diag3(1e-4), false);
        ROFT::UKFCorrection p_correction(std::unique_ptr<bfl::MeasurementModel>(cqm), 1.0, 2.0, 0.0);
        using Mode = ROFT::CartesianQuaternionMeasurement::MeasurementMode;
        if (!p_correction.getMeasurementModel().freeze(Mode::Standard)) return 10;
        if (p_correction.getMeasurementModel().getMeasurementDescription().total_size() != 13) return 10;   // pose + velocity
        p_correction.correct(pq, pc);
        p_correction.getMeasurementModel().freeze(Mode::RepeatOnlyVelocity);                                 // second alternative
        if (p_correction.getMeasurementModel().getMeasurementDescription().total_size() != 6) return 11;
        p_correction.correct(pq, pv);
        put(o, pc.mean()); put(o, pc.covariance());
        put(o, pv.mean()); put(o, pv.covariance());
        // re-sync replay: the buffered twist comes back once, with the pose that is still pending (the replay consumes it),
        // then the buffer is empty: the replay ends and the current twist is kept for the next one
        if (!p_correction.getMeasurementModel().freeze(Mode::PopBufferedMeasurement) || cqm->measurement_type() != ROFT_MEAS_POSE_VELOCITY) return 12;
        if (p_correction.getMeasurementModel().freeze(Mode::PopBufferedMeasurement) || cqm->buffered_velocities() != 1) return 13;
        // h(x) and the innovation on the host: the mean column predicts itself
        {
            p_correction.getMeasurementModel().freeze(Mode::Standard);
            Eigen::MatrixXd col(13 + 12, 1);
            for (int i = 0; i < 13; ++i) col(i) = pq.mean()(i);
            bfl::Data hx = cqm->predictedMeasure(col).second, z = cqm->measure().second;
            const Eigen::MatrixXd innov = bfl::any::any_cast<Eigen::MatrixXd>(cqm->innovation(hx, z).second);
            if (innov.rows() != 12) return 14;
            put(o, innov);
        }
        // ---- flow measurement model on the recorded stream: frame 0 latches, frame 1 measures
        auto camera = std::make_shared<ROFT::CameraMeasurement>(std::make_shared<MemCamera>(s));
        auto flow_src = std::make_shared<MemFlow>(s);
        auto seg_meas = std::make_shared<ROFT::ImageSegmentationMeasurement>(std::make_shared<MemSegmentation>(s, 6), camera);
        ROFT::ImageOpticalFlowMeasurement<cv::Vec2f> flow_meas(flow_src, camera, seg_meas, 35, 2.0, Eigen::MatrixXd::Identity(2, 2), false);
        using Freeze = ROFT::ImageOpticalFlowMeasurementBase::FreezeType;
        bool frozen[2];
        for (int k = 0; k < 2; ++k) {
            if (!camera->freeze(ROFT::CameraMeasurementType::RGBD) || !seg_meas->freeze()) return 15;
            frozen[k] = flow_meas.freeze(std::make_pair(Freeze::Complete, s.frames[k].dt));
        }
        if (frozen[0] || !frozen[1] || !flow_meas.setProperty("check_observability")) return 16;
        const Eigen::MatrixXd fy = bfl::any::any_cast<Eigen::MatrixXd>(flow_meas.measure().second), fH = flow_meas.getMeasurementMatrix();
        const double n_pts = (double)(fy.rows() / 2);
        std::fwrite(&n_pts, 8, 1, o);
        put(o, fy);
        put(o, fH);
        // ---- flow-aided segmentation source over eight frames (a new mask arrives with frame 6)
        auto flow2 = std::make_shared<MemFlow>(s);
        ROFT::ImageSegmentationOFAidedSource<cv::Vec2f> aided(std::make_shared<MemSegmentation>(s, 6), flow2, s.parameters(), false);
        for (int k = 0; k < 8; ++k) {
            flow2->step_frame();
            if (!aided.step_frame()) return 17;
        }
        const cv::Mat m = aided.segmentation(false).second;
        std::fwrite(m.data, 1, m.total(), o);
        std::fclose(o);
    } catch (const std::runtime_error& e) {
        std::printf("runtime_error: %s\n", e.what());
        return 3;
    }
    return 0;
}
