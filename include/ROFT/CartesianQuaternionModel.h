// ROFT::CartesianQuaternionModel -- state model of the pose UKF (reference:
// src/roft-lib/include/ROFT/CartesianQuaternionModel.h:24-82, src/CartesianQuaternionModel.cpp:86-141).
// State column [v(3) w(3) x(3) q(4: w x y z)], process noise [n_v n_w n_x] (9).  bfl::UKFPrediction over this model is
// the prediction half of the pose stage; here it runs through roft_ukf_predict, the noise covariance Q(T) through
// roft_pose_process_noise.  motion() / propagate() are kept for callers that walk sigma points themselves.
#pragma once

#include "Compat.h"

namespace ROFT {

class CartesianQuaternionModel : public bfl::StateModel {
public:
    CartesianQuaternionModel(const Eigen::Ref<const Eigen::MatrixXd> psd_linear_acceleration,
                             const Eigen::Ref<const Eigen::MatrixXd> sigma_angular_velocity, const double sample_time)
        : sample_time_(sample_time)
    {
        compat::diagonal_of(psd_linear_acceleration, psd_, 3);
        compat::diagonal_of(sigma_angular_velocity, sigma_w_, 3);
    }
    virtual ~CartesianQuaternionModel() = default;

    // noise-free motion (cpp:86-124 without the noise rows): x' = x + v T, q' = exp(w T / 2) (x) q
    void propagate(const Eigen::Ref<const Eigen::MatrixXd>& cur_states, Eigen::Ref<Eigen::MatrixXd> mot_states) override
    {
        move(cur_states, mot_states, false);
    }
    // columns carry 9 noise rows below the 13 state rows; v and w enter the kinematics WITHOUT their noise (cpp:97, 103)
    void motion(const Eigen::Ref<const Eigen::MatrixXd>& cur_states, Eigen::Ref<Eigen::MatrixXd> mot_states) override
    {
        move(cur_states, mot_states, cur_states.rows() >= 22);
    }
    bool setSamplingTime(const double& sample_time) override { sample_time_ = sample_time; return true; }
    bool setProperty(const std::string& /*property*/) override { return false; }
    Eigen::MatrixXd getNoiseCovarianceMatrix() override
    {
        Eigen::MatrixXd Q(9, 9);
        compat::throw_if(roft_pose_process_noise(psd_, sigma_w_, sample_time_, Q.data()), "CartesianQuaternionModel::getNoiseCovarianceMatrix");
        return Q;
    }
    bfl::VectorDescription getInputDescription() override { return bfl::VectorDescription(9, 1, 9); }
    bfl::VectorDescription getStateDescription() override { return bfl::VectorDescription(9, 1, 0); }
    double sample_time() const { return sample_time_; }

private:
    void move(const Eigen::MatrixXd& cur, Eigen::MatrixXd& mot, bool with_noise) const
    {
        const double T = sample_time_;
        mot.resize(13, cur.cols());
        for (std::size_t c = 0; c < cur.cols(); ++c) {
            double n[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (with_noise) for (int i = 0; i < 9; ++i) n[i] = cur(13 + i, c);
            const double v[3] = {cur(0, c), cur(1, c), cur(2, c)}, w[3] = {cur(3, c), cur(4, c), cur(5, c)};
            for (int i = 0; i < 3; ++i) {
                mot(i, c) = v[i] + n[i];
                mot(3 + i, c) = w[i] + n[3 + i];
                mot(6 + i, c) = (cur(6 + i, c) + n[6 + i]) + v[i] * T;
            }
            const double nw = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]) + 2.220446049250313e-16;
            const double th = nw * T, cs = std::cos(th / 2.0), sn = std::sin(th / 2.0) / nw;
            const double q[4] = {cur(9, c), cur(10, c), cur(11, c), cur(12, c)};
            mot(9, c) = cs * q[0] + sn * (-w[0] * q[1] - w[1] * q[2] - w[2] * q[3]);
            mot(10, c) = cs * q[1] + sn * (w[0] * q[0] - w[2] * q[2] + w[1] * q[3]);
            mot(11, c) = cs * q[2] + sn * (w[1] * q[0] + w[2] * q[1] - w[0] * q[3]);
            mot(12, c) = cs * q[3] + sn * (w[2] * q[0] - w[1] * q[1] + w[0] * q[2]);
        }
    }
    double psd_[3], sigma_w_[3];
    double sample_time_;
    const std::string log_name_ = "CartesianQuaternionModel";
};

}  // namespace ROFT

namespace bfl {

// bfl::UKFPrediction over a generic state model, accelerated for ROFT::CartesianQuaternionModel: the unscented
// transform through motion() with the augmented noise Q(T) is one launch of roft_ukf_predict.
class UKFPrediction : public GaussianPrediction {
public:
    UKFPrediction(std::unique_ptr<StateModel> state_model, const double alpha, const double beta, const double kappa)
        : model_(std::move(state_model)), ut_{alpha, beta, kappa}
    {
        if (!dynamic_cast<ROFT::CartesianQuaternionModel*>(model_.get()))
            throw std::runtime_error("UKFPrediction::ctor. Error: only ROFT::CartesianQuaternionModel is accelerated.");
    }
    StateModel& getStateModel() override { return *model_; }

protected:
    void predictStep(const GaussianMixture& prev_state, GaussianMixture& pred_state) override
    {
        auto& m = static_cast<ROFT::CartesianQuaternionModel&>(*model_);
        const Eigen::MatrixXd Q = m.getNoiseCovarianceMatrix();
        ROFT::compat::throw_if(roft_ukf_predict(prev_state.mean().data(), prev_state.covariance().data(), Q.data(), m.sample_time(), &ut_,
                                                pred_state.mean().data(), pred_state.covariance().data()), "UKFPrediction::predictStep");
    }

private:
    std::unique_ptr<StateModel> model_;
    roft_ut_params ut_;
};

}  // namespace bfl
