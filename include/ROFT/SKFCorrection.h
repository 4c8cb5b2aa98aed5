// ROFT::SKFCorrection -- sequential Kalman correction of the velocity filter with Laplacian re-weighting of the flow
// residuals (reference: src/roft-lib/include/ROFT/SKFCorrection.h:23-52, src/SKFCorrection.cpp:37-153).  The measurement
// is consumed in blocks of measurement_sub_size = 2 rows with the model's 2 x 2 noise covariance; the N rank-2 updates
// of the reference are one launch of roft_skf_correct (information form, same result to rounding; the median / weight
// arithmetic of cpp:91-116 included).
#pragma once

#include "Compat.h"

namespace ROFT {

class SKFCorrection : public bfl::GaussianCorrection {
public:
    SKFCorrection(std::unique_ptr<bfl::LinearMeasurementModel> measurement_model, const std::size_t measurement_sub_size,
                  const bool use_laplacian_reweighting = false)
        : measurement_model_(std::move(measurement_model)), measurement_sub_size_(measurement_sub_size),
          use_laplacian_reweighting_(use_laplacian_reweighting)
    {
        if (!measurement_model_) throw std::runtime_error(log_name_ + "::ctor. Error: null measurement model.");
        if (measurement_sub_size_ != 2) throw std::runtime_error(log_name_ + "::ctor. Error: measurement_sub_size must be 2 (flow vectors).");
    }
    virtual ~SKFCorrection() = default;
    bfl::MeasurementModel& getMeasurementModel() override { return *measurement_model_; }

protected:
    void correctStep(const bfl::GaussianMixture& pred_state, bfl::GaussianMixture& corr_state) override
    {
        // no measurement, no noise covariance: the predicted belief passes through (cpp:46-69)
        bool valid = false;
        bfl::Data measurement;
        std::tie(valid, measurement) = measurement_model_->measure();
        Eigen::MatrixXd R;
        bool valid_R = false;
        if (valid) std::tie(valid_R, R) = measurement_model_->getNoiseCovarianceMatrix();
        if (!valid || !valid_R) {
            corr_state = pred_state;
            return;
        }
        const Eigen::MatrixXd& y = *bfl::any::any_cast<Eigen::MatrixXd>(&measurement);
        const Eigen::MatrixXd H = measurement_model_->getMeasurementMatrix();
        if (R.rows() != 2 || R.cols() != 2 || R(0, 1) != 0.0 || R(1, 0) != 0.0)
            throw std::runtime_error(log_name_ + "::correctStep. Error: a diagonal 2 x 2 noise covariance is expected.");
        if (H.cols() != 6 || H.rows() != y.rows() || (y.rows() % 2) != 0)
            throw std::runtime_error(log_name_ + "::correctStep. Error: unexpected measurement sizes.");
        const double rdiag[2] = {R(0, 0), R(1, 1)};
        int status = 0;
        compat::throw_if(roft_skf_correct(pred_state.mean().data(), pred_state.covariance().data(), (int)(y.rows() / 2), y.data(), H.data(), rdiag,
                                          use_laplacian_reweighting_ ? 1 : 0, corr_state.mean().data(), corr_state.covariance().data(), &status),
                         "SKFCorrection::correctStep");
    }

private:
    std::unique_ptr<bfl::LinearMeasurementModel> measurement_model_;
    std::size_t measurement_sub_size_;
    bool use_laplacian_reweighting_;
    const std::string log_name_ = "SKFCorrection";
};

}  // namespace ROFT
