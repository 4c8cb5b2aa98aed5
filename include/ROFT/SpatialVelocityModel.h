// ROFT::SpatialVelocityModel -- constant-twist state model of the velocity filter (reference:
// src/roft-lib/include/ROFT/SpatialVelocityModel.h:20-57, src/SpatialVelocityModel.cpp:15-27): F = I_6,
// Q = blkdiag(sigma_v, sigma_w).  bfl::KFPrediction over it is the prediction half of the velocity stage; here that
// prediction runs through roft_kf_predict (include/roft_engine.h).
#pragma once

#include "Compat.h"

namespace ROFT {

class SpatialVelocityModel : public bfl::LinearStateModel {
public:
    SpatialVelocityModel(const Eigen::Ref<const Eigen::MatrixXd> sigma_v, const Eigen::Ref<const Eigen::MatrixXd> sigma_w)
        : F_(Eigen::MatrixXd::Identity(6, 6)), Q_(6, 6)
    {
        if (sigma_v.rows() != 3 || sigma_v.cols() != 3 || sigma_w.rows() != 3 || sigma_w.cols() != 3)
            throw std::runtime_error("SpatialVelocityModel::ctor. Error: sigma_v and sigma_w must be 3 x 3.");
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) { Q_(i, j) = sigma_v(i, j); Q_(3 + i, 3 + j) = sigma_w(i, j); }
    }
    virtual ~SpatialVelocityModel() = default;
    Eigen::MatrixXd getStateTransitionMatrix() override { return F_; }
    Eigen::MatrixXd getNoiseCovarianceMatrix() override { return Q_; }
    bfl::VectorDescription getInputDescription() override { return bfl::VectorDescription(6, 0, 6); }
    bfl::VectorDescription getStateDescription() override { return bfl::VectorDescription(6, 0, 0); }
    bool setProperty(const std::string& /*property*/) override { return false; }

protected:
    Eigen::MatrixXd F_, Q_;
    const std::string log_name_ = "SpatialVelocityModel";
};

}  // namespace ROFT

namespace bfl {

// bfl::KFPrediction over a linear state model.  Accelerated for the model ROFT uses -- F = I and a diagonal Q
// (ROFT::SpatialVelocityModel with diagonal sigma blocks): x- = x, P- = P + Q on the GPU (roft_kf_predict).
class KFPrediction : public GaussianPrediction {
public:
    explicit KFPrediction(std::unique_ptr<LinearStateModel> state_model) : model_(std::move(state_model))
    {
        if (!model_) throw std::runtime_error("KFPrediction::ctor. Error: null state model.");
    }
    StateModel& getStateModel() override { return *model_; }

protected:
    void predictStep(const GaussianMixture& prev_state, GaussianMixture& pred_state) override
    {
        const Eigen::MatrixXd F = model_->getStateTransitionMatrix(), Q = model_->getNoiseCovarianceMatrix();
        if (F.rows() != 6 || Q.rows() != 6) throw std::runtime_error("KFPrediction: only the 6-state velocity model is accelerated");
        double qd[6];
        for (std::size_t i = 0; i < 6; ++i)
            for (std::size_t j = 0; j < 6; ++j) {
                if (F(i, j) != (i == j ? 1.0 : 0.0) || (i != j && Q(i, j) != 0.0))
                    throw std::runtime_error("KFPrediction: only F = I with a diagonal Q is accelerated (SpatialVelocityModel)");
                if (i == j) qd[i] = Q(i, i);
            }
        ROFT::compat::throw_if(roft_kf_predict(prev_state.mean().data(), prev_state.covariance().data(), qd,
                                               pred_state.mean().data(), pred_state.covariance().data()), "KFPrediction::predictStep");
    }

private:
    std::unique_ptr<LinearStateModel> model_;
};

}  // namespace bfl
