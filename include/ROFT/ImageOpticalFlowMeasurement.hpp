// ROFT::ImageOpticalFlowMeasurement<T> -- the linear measurement model of the velocity filter (reference:
// src/roft-lib/include/ROFT/ImageOpticalFlowMeasurement.hpp:43-128; freeze :167-294).  T = cv::Vec2f (CV_32FC2 flow, grid 1)
// or cv::Vec2s (CV_16SC2 S10.5 flow, grid 4).  freeze() keeps the reference's state machine -- first frame / missing flow
// latch depth and mask and report "no measurement" -- and hands the selection of the sampled mask pixels, the validity
// tests and the assembly of y (2N) and H (2N x 6) to the GPU: roft_flow_measurement (include/roft_engine.h).
#pragma once

#include "Sources.h"

namespace ROFT {

class ImageOpticalFlowMeasurementBase {
public:
    enum class FreezeType { OnlyStepSource, ExceptStepSource, Complete };
};

template <class T>
class ImageOpticalFlowMeasurement : public bfl::LinearMeasurementModel, ImageOpticalFlowMeasurementBase {
public:
    using FreezeType = ImageOpticalFlowMeasurementBase::FreezeType;

    ImageOpticalFlowMeasurement(std::shared_ptr<ImageOpticalFlowSource> flow_source, std::shared_ptr<CameraMeasurement> camera_measurement,
                                std::shared_ptr<ROFT::ImageSegmentationMeasurement> segmentation, const std::size_t& segmentation_radius,
                                const double& maximum_depth, Eigen::Ref<const Eigen::MatrixXd> covariance, const bool use_full_covariance_matrix)
        : flow_(std::move(flow_source)), camera_(std::move(camera_measurement)), segmentation_(std::move(segmentation)),
          covariance_(covariance), use_full_covariance_(use_full_covariance_matrix), segmentation_radius_((float)segmentation_radius),
          maximum_depth_(maximum_depth), flow_grid_size_(flow_->get_grid_size()), flow_scaling_factor_(flow_->get_scaling_factor())
    {
        static_assert(sizeof(T) == 8 || sizeof(T) == 4, "T is cv::Vec2f or cv::Vec2s");
        bool valid = false;
        std::tie(valid, camera_parameters_) = camera_->camera_parameters();
        if (!valid) throw std::runtime_error(log_name_ + "::ctor. Error: cannot get camera parameters.");
        if (covariance_.rows() != 2 || covariance_.cols() != 2) throw std::runtime_error(log_name_ + "::ctor. Error: the flow covariance is 2 x 2.");
        if ((sizeof(T) == 8) != (flow_->get_matrix_type() == CV_32FC2))
            throw std::runtime_error(log_name_ + "::ctor. Error: T does not match the matrix type of the flow source.");
    }
    ~ImageOpticalFlowMeasurement() = default;

    // data = std::pair<FreezeType, double sample_time> (hpp:169)
    bool freeze(const bfl::Data& data = bfl::Data()) override
    {
        std::tie(freeze_type_, sample_time_) = bfl::any::any_cast<std::pair<FreezeType, double>>(data);
        if (freeze_type_ != FreezeType::ExceptStepSource && flow_->is_stepping_required()) flow_->step_frame();
        if (freeze_type_ == FreezeType::OnlyStepSource) return true;

        // segmentation and camera have been frozen by the caller (hpp:180-207)
        bool valid = false;
        bfl::Data seg_data, cam_data;
        std::tie(valid, seg_data) = segmentation_->measure();
        if (!valid) return false;
        const cv::Mat segmentation = bfl::any::any_cast<std::pair<bool, cv::Mat>>(seg_data).second;
        std::tie(valid, cam_data) = camera_->measure();
        if (!valid) return false;
        const Eigen::MatrixXf& depth = std::get<2>(*bfl::any::any_cast<CameraMeasurement::CameraMeasurementTuple>(&cam_data));

        cv::Mat flow;
        flow_available_ = false;
        std::tie(flow_available_, flow) = flow_->flow(false);
        if (!flow_available_ || is_first_frame_) {
            // (hpp:217-229) nothing to measure against yet: latch and report no measurement
            previous_depth_ = depth;
            previous_segmentation_ = segmentation;
            is_first_frame_ = false;
            flow_available_ = false;
            return false;
        }
        // every segmentation_radius-th non-zero mask pixel in row-major order, kept if flow and depth are valid; y and H rows
        const int W = (int)camera_parameters_.width(), H = (int)camera_parameters_.height();
        const roft_camera cam{W, H, camera_parameters_.fx(), camera_parameters_.fy(), camera_parameters_.cx(), camera_parameters_.cy()};
        roft_flow fd;
        fd.data = flow.data;
        fd.type = flow.type() == CV_16SC2 ? ROFT_FLOW_S16C2 : ROFT_FLOW_F32C2;
        fd.cols = flow.cols;
        fd.rows = flow.rows;
        fd.grid = (int)flow_grid_size_;
        fd.scale = flow_scaling_factor_;
        fd.valid = 1;
        const int radius = segmentation_radius_ >= 1.f ? (int)segmentation_radius_ : 1;
        const int cap = W * H / radius + 16;
        uv_.assign((std::size_t)2 * cap, 0);
        Eigen::MatrixXd y((std::size_t)2 * cap, 1), Hm((std::size_t)2 * cap, 6);
        int n = 0;
        compat::throw_if(roft_flow_measurement(&cam, previous_segmentation_.data, previous_depth_.data(), &fd, sample_time_, segmentation_radius_,
                                               maximum_depth_, cap, uv_.data(), y.data(), Hm.data(), &n), "ImageOpticalFlowMeasurement::freeze");
        measurement_.resize((std::size_t)2 * n, 1);
        measurement_matrix_.resize((std::size_t)2 * n, 6);
        std::memcpy(measurement_.data(), y.data(), sizeof(double) * 2 * n);
        std::memcpy(measurement_matrix_.data(), Hm.data(), sizeof(double) * 12 * n);
        previous_depth_ = depth;
        previous_segmentation_ = segmentation;
        data_loading_time_ = flow_->get_data_loading_time() + segmentation_->get_data_loading_time();
        return flow_available_;
    }
    std::pair<bool, bfl::Data> measure(const bfl::Data& = bfl::Data()) const override { return std::make_pair(flow_available_, bfl::Data(measurement_)); }
    std::pair<bool, bfl::Data> predictedMeasure(const Eigen::Ref<const Eigen::MatrixXd>& cur_states) const override
    {
        if (!flow_available_) return std::make_pair(false, bfl::Data());
        Eigen::MatrixXd out(measurement_matrix_.rows(), cur_states.cols());
        for (std::size_t i = 0; i < out.rows(); ++i)
            for (std::size_t c = 0; c < out.cols(); ++c) {
                double s = 0.0;
                for (std::size_t k = 0; k < 6; ++k) s += measurement_matrix_(i, k) * cur_states(k, c);
                out(i, c) = s;
            }
        return std::make_pair(true, bfl::Data(std::move(out)));
    }
    std::pair<bool, bfl::Data> innovation(const bfl::Data& predicted_measurements, const bfl::Data& measurements) const override
    {
        const Eigen::MatrixXd& p = *bfl::any::any_cast<Eigen::MatrixXd>(&predicted_measurements);
        const Eigen::MatrixXd& m = *bfl::any::any_cast<Eigen::MatrixXd>(&measurements);
        Eigen::MatrixXd out(p.rows(), p.cols());
        for (std::size_t i = 0; i < p.rows(); ++i)
            for (std::size_t c = 0; c < p.cols(); ++c) out(i, c) = -(p(i, c) - m(i, 0));
        return std::make_pair(true, bfl::Data(std::move(out)));
    }
    Eigen::MatrixXd getMeasurementMatrix() const override { return measurement_matrix_; }
    std::pair<bool, Eigen::MatrixXd> getNoiseCovarianceMatrix() const override
    {
        if (!use_full_covariance_) return std::make_pair(true, covariance_);
        const std::size_t n = measurement_.rows();
        Eigen::MatrixXd full(n, n);
        for (std::size_t i = 0; i < n / 2; ++i)
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b) full(2 * i + a, 2 * i + b) = covariance_(a, b);
        return std::make_pair(true, full);
    }
    bfl::VectorDescription getInputDescription() const override { return bfl::VectorDescription(6, 0, measurement_.size()); }
    bfl::VectorDescription getMeasurementDescription() const override { return bfl::VectorDescription(measurement_.size(), 0); }
    bool setProperty(const std::string& property) override
    {
        if (property == "check_observability") return (measurement_.rows() / 2) >= 3;   // hpp:361-366
        if (property == "reset") { flow_available_ = false; is_first_frame_ = true; return true; }
        return false;
    }
    void reset_data_loading_time() { segmentation_->reset_data_loading_time(); }
    double get_data_loading_time() { return data_loading_time_; }
    // pixel (u, v) of every kept point, in measurement order (not part of the reference's interface)
    const std::vector<std::int32_t>& selected_pixels() const { return uv_; }

private:
    std::shared_ptr<ImageOpticalFlowSource> flow_;
    std::shared_ptr<CameraMeasurement> camera_;
    std::shared_ptr<ImageSegmentationMeasurement> segmentation_;
    Eigen::MatrixXd measurement_matrix_, measurement_, covariance_;
    Eigen::MatrixXf previous_depth_;
    cv::Mat previous_segmentation_;
    bool use_full_covariance_;
    RobotsIO::Camera::CameraParameters camera_parameters_;
    const float segmentation_radius_;
    const double maximum_depth_;
    const std::size_t flow_grid_size_;
    const float flow_scaling_factor_;
    double sample_time_ = 0.0;
    bool flow_available_ = false, is_first_frame_ = true;
    FreezeType freeze_type_ = FreezeType::Complete;
    double data_loading_time_ = 0.0;
    std::vector<std::int32_t> uv_;
    const std::string log_name_ = "ImageOpticalFlowMeasurement";
};

}  // namespace ROFT
