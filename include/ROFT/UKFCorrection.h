// ROFT::UKFCorrection -- unscented correction of the pose belief (reference:
// src/roft-lib/include/ROFT/UKFCorrection.h:24-52, src/UKFCorrection.cpp:54-133).  Accelerated for the measurement model
// ROFT uses: with a ROFT::CartesianQuaternionMeasurement the sigma-point fan-out, h(x), the innovation with its
// quaternion rows, Py / Pxy, the gain and the update are one launch of roft_ukf_correct; no measurement or a singular
// innovation covariance leaves corr = pred (cpp:64-68).
#pragma once

#include "CartesianQuaternionMeasurement.h"

namespace ROFT {

class UKFCorrection : public bfl::GaussianCorrection {
public:
    UKFCorrection(std::unique_ptr<bfl::MeasurementModel> meas_model, const double alpha, const double beta, const double kappa) noexcept
        : measurement_model_(std::move(meas_model)), ut_alpha_(alpha), ut_beta_(beta), ut_kappa_(kappa)
    {}
    UKFCorrection(UKFCorrection&& other) noexcept
        : measurement_model_(std::move(other.measurement_model_)), ut_alpha_(other.ut_alpha_), ut_beta_(other.ut_beta_), ut_kappa_(other.ut_kappa_)
    {}
    virtual ~UKFCorrection() noexcept = default;
    bfl::MeasurementModel& getMeasurementModel() override { return *measurement_model_; }
    std::pair<bool, Eigen::VectorXd> getLikelihood() override
    {
        throw std::runtime_error("Error: ROFT::UKFCorrection::getLikelihood() is not implemented.");
    }
    // 0 corrected, 1 no measurement, 2 singular innovation covariance (corr = pred) -- of the last correct()
    int last_status() const { return status_; }

protected:
    void correctStep(const bfl::GaussianMixture& pred_state, bfl::GaussianMixture& corr_state) override
    {
        auto* model = dynamic_cast<CartesianQuaternionMeasurement*>(measurement_model_.get());
        if (!model) throw std::runtime_error("UKFCorrection::correctStep. Error: only ROFT::CartesianQuaternionMeasurement is accelerated.");
        double R[12];
        model->noise_diagonal(R);
        const roft_ut_params ut{ut_alpha_, ut_beta_, ut_kappa_};
        compat::throw_if(roft_ukf_correct(pred_state.mean().data(), pred_state.covariance().data(), model->measurement_type(),
                                          model->measurement_data(), R, &ut, corr_state.mean().data(), corr_state.covariance().data(), &status_),
                         "UKFCorrection::correctStep");
    }
    std::unique_ptr<bfl::MeasurementModel> measurement_model_;
    const double ut_alpha_, ut_beta_, ut_kappa_;
    int status_ = 0;
};

}  // namespace ROFT
