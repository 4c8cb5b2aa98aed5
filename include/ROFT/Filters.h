// Filters.h -- C++ facade over the C ABI (include/roft_engine.h) with the reference's class names and
// call shapes, so that code written against roft-lib's filter classes reads the same:
//
//   reference class (src/roft-lib/include/ROFT/...)          facade here                     C ABI behind it
//   SpatialVelocityModel + bfl::KFPrediction                 ROFT::KFPrediction              roft_kf_predict
//   SKFCorrection (SKFCorrection.h:23-52)                    ROFT::SKFCorrection             roft_skf_correct
//   ImageOpticalFlowMeasurement<T> (…Measurement.hpp:43-128) ROFT::ImageOpticalFlowMeasurement roft_flow_measurement
//   ImageSegmentationOFAidedSource<T>::map + remap           ROFT::ImageSegmentationOFAidedSource roft_mask_propagate
//   CartesianQuaternionModel + bfl::UKFPrediction            ROFT::UKFPrediction             roft_ukf_predict
//   UKFCorrection + CartesianQuaternionMeasurement           ROFT::UKFCorrection             roft_ukf_correct
//   ROFTFilter (ROFTFilter.h:38-194)                         ROFT::ROFTFilter                roft_engine_*
//   ImageOpticalFlowNVOF (ImageOpticalFlowNVOF.h:24-90)      ROFT::ImageOpticalFlowHIP       roft_optical_flow
//
// Conventions kept from the reference: predict(prev, pred) / correct(pred, corr) on Gaussians; an
// invalid / empty measurement leaves corr = pred (SKFCorrection.cpp:46-69, UKFCorrection.cpp:64-68);
// constructors and unrecoverable errors throw std::runtime_error; freeze() returns a validity bool.
// Header-only; link with libroft_hip.so.
#pragma once

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <utility>
#include <vector>

#include "../roft_engine.h"
#include "Compat.h"

namespace ROFT {

using compat::Gaussian;
using compat::MatrixXd;
using compat::VectorXd;

struct CameraParameters {
    int width = 0, height = 0;
    double fx = 0, fy = 0, cx = 0, cy = 0;
    roft_camera c() const { return roft_camera{width, height, fx, fy, cx, cy}; }
};

// ---- velocity filter -------------------------------------------------------------------------
class KFPrediction {
public:
    // SpatialVelocityModel(sigma_v, sigma_w): F = I, Q = diag(sigma_v, sigma_w)
    KFPrediction(const double sigma_v[3], const double sigma_w[3])
    {
        for (int i = 0; i < 3; ++i) { q_[i] = sigma_v[i]; q_[3 + i] = sigma_w[i]; }
    }
    void predict(const Gaussian& prev, Gaussian& pred) const
    {
        compat::throw_if(roft_kf_predict(prev.mean().data(), prev.covariance().data(), q_, pred.mean().data(),
                                         pred.covariance().data()), "KFPrediction::predict");
    }

private:
    double q_[6];
};

class ImageOpticalFlowMeasurement {
public:
    ImageOpticalFlowMeasurement(const CameraParameters& cam, std::size_t subsampling_radius, double maximum_depth,
                                const double cov_flow[2])
        : cam_(cam), radius_(static_cast<float>(subsampling_radius)), max_depth_(maximum_depth)
    {
        r_[0] = cov_flow[0];
        r_[1] = cov_flow[1];
    }
    // freeze(): previous frame's binarised mask and depth + this frame's flow -> (y, H); returns
    // false (and keeps the previous measurement) when the flow is not available, like hpp:217-229
    bool freeze(const std::uint8_t* previous_segmentation, const float* previous_depth, const roft_flow& flow,
                double sample_time)
    {
        if (!flow.valid || !flow.data) return false;
        const int cap = cam_.width * cam_.height / (radius_ >= 1.f ? static_cast<int>(radius_) : 1) + 16;
        uv_.assign(2 * cap, 0);
        y_.resize(2 * cap, 1);
        H_.resize(2 * cap, 6);
        int n = 0;
        roft_camera c = cam_.c();
        compat::throw_if(roft_flow_measurement(&c, previous_segmentation, previous_depth, &flow, sample_time, radius_,
                                               max_depth_, cap, uv_.data(), y_.data(), H_.data(), &n),
                         "ImageOpticalFlowMeasurement::freeze");
        n_ = n;
        return true;
    }
    std::size_t size() const { return n_; }                    // number of kept points
    const double* measure() const { return y_.data(); }        // 2N
    const double* getMeasurementMatrix() const { return H_.data(); }  // 2N x 6
    const double* getNoiseCovarianceMatrix() const { return r_; }
    // setProperty("check_observability") of the reference (hpp:361-366)
    bool check_observability() const { return n_ >= 3; }

private:
    CameraParameters cam_;
    float radius_;
    double max_depth_;
    double r_[2];
    std::vector<std::int32_t> uv_;
    MatrixXd y_, H_;
    std::size_t n_ = 0;
};

class SKFCorrection {
public:
    SKFCorrection(std::shared_ptr<ImageOpticalFlowMeasurement> measurement_model, std::size_t measurement_sub_size,
                  bool use_laplacian_reweighting = false)
        : model_(std::move(measurement_model)), reweight_(use_laplacian_reweighting)
    {
        if (measurement_sub_size != 2) throw std::runtime_error("SKFCorrection: measurement_sub_size must be 2");
    }
    ImageOpticalFlowMeasurement& getMeasurementModel() { return *model_; }
    void correct(const Gaussian& pred, Gaussian& corr)
    {
        int status = 0;
        compat::throw_if(roft_skf_correct(pred.mean().data(), pred.covariance().data(), static_cast<int>(model_->size()),
                                          model_->measure(), model_->getMeasurementMatrix(),
                                          model_->getNoiseCovarianceMatrix(), reweight_ ? 1 : 0, corr.mean().data(),
                                          corr.covariance().data(), &status), "SKFCorrection::correct");
    }

private:
    std::shared_ptr<ImageOpticalFlowMeasurement> model_;
    bool reweight_;
};

// ---- mask propagation -----------------------------------------------------------------------------
class ImageSegmentationOFAidedSource {
public:
    ImageSegmentationOFAidedSource(int width, int height, int frames_between_iterations)
        : w_(width), h_(height), fb_(frames_between_iterations) {}
    // map() + cv::remap(): propagate `mask` (in place) through the given flow frames (chronological)
    void propagate(std::uint8_t* mask, const std::vector<roft_flow>& flows) const
    {
        compat::throw_if(roft_mask_propagate(mask, w_, h_, flows.data(), static_cast<int>(flows.size()), fb_),
                         "ImageSegmentationOFAidedSource::propagate");
    }

private:
    int w_, h_, fb_;
};

// ---- optical-flow source (the step before the filter) ---------------------------------------------------
// Same interface as ImageOpticalFlowNVOF (step_frame / flow / get_grid_size / get_scaling_factor /
// get_matrix_type, ImageOpticalFlowNVOF.cpp:100-200) over the HIP pyramidal Lucas-Kanade producer; the frame is
// handed to step_frame() as an 8-bit gray image (what cv::cvtColor(frame, COLOR_BGR2GRAY) leaves, cpp:123).
class ImageOpticalFlowHIP {
public:
    enum class Product { NVOF_1_0 = 1, NVOF_2_0 = 2 };   // CV_16SC2 S10.5 at grid 4 | CV_32FC2 at grid 1 (cpp:19-80)
    ImageOpticalFlowHIP(int width, int height, Product product) : w_(width), h_(height), product_(product)
    {
        compat::throw_if(roft_default_of_params(&prm_), "ImageOpticalFlowHIP");
        if (width <= 0 || height <= 0) throw std::runtime_error("ImageOpticalFlowHIP: bad image size");
        const bool v1 = product == Product::NVOF_1_0;
        flow_.resize(v1 ? static_cast<std::size_t>(width / 4) * (height / 4) * 4 : static_cast<std::size_t>(width) * height * 8);
    }
    // returns false on the first frame (no previous image yet), like the reference
    bool step_frame(const std::uint8_t* gray)
    {
        const std::size_t n = static_cast<std::size_t>(w_) * h_;
        if (last_.empty()) {
            last_.assign(gray, gray + n);
            return false;
        }
        compat::throw_if(roft_optical_flow(last_.data(), gray, w_, h_, &prm_, static_cast<int>(get_matrix_type() == 11 ? ROFT_FLOW_S16C2 : ROFT_FLOW_F32C2),
                                           flow_.data()), "ImageOpticalFlowHIP::step_frame");
        last_.assign(gray, gray + n);
        flow_in_ = true;
        return true;
    }
    std::pair<bool, const void*> flow(bool /*blocking*/ = false) const { return {flow_in_, flow_.data()}; }
    bool is_stepping_required() const { return true; }
    std::size_t get_grid_size() const { return product_ == Product::NVOF_1_0 ? 4 : 1; }
    float get_scaling_factor() const { return product_ == Product::NVOF_1_0 ? 32.0f : 1.0f; }
    int get_matrix_type() const { return product_ == Product::NVOF_1_0 ? 11 /* CV_16SC2 */ : 13 /* CV_32FC2 */; }
    int flow_cols() const { return w_ / static_cast<int>(get_grid_size()); }
    int flow_rows() const { return h_ / static_cast<int>(get_grid_size()); }
    roft_of_params& parameters() { return prm_; }

private:
    int w_, h_;
    Product product_;
    roft_of_params prm_{};
    std::vector<std::uint8_t> last_;
    std::vector<unsigned char> flow_;
    bool flow_in_ = false;
};
using ImageOpticalFlowNVOF = ImageOpticalFlowHIP;   // drop-in name

// ---- pose filter ------------------------------------------------------------------------------------
class UKFPrediction {
public:
    // CartesianQuaternionModel(psd_linear_acceleration, sigma_angular_velocity, sample_time) + UT parameters
    UKFPrediction(const double psd_lin_acc[3], const double sigma_ang_vel[3], double sample_time, double alpha,
                  double beta, double kappa)
        : T_(sample_time), ut_{alpha, beta, kappa}
    {
        for (int i = 0; i < 3; ++i) { psd_[i] = psd_lin_acc[i]; sw_[i] = sigma_ang_vel[i]; }
    }
    bool setSamplingTime(double sample_time) { T_ = sample_time; return true; }
    void predict(const Gaussian& prev, Gaussian& pred) const
    {
        double Q[81];
        compat::throw_if(roft_pose_process_noise(psd_, sw_, T_, Q), "CartesianQuaternionModel::Q");
        compat::throw_if(roft_ukf_predict(prev.mean().data(), prev.covariance().data(), Q, T_, &ut_, pred.mean().data(),
                                          pred.covariance().data()), "UKFPrediction::predict");
    }

private:
    double psd_[3], sw_[3], T_;
    roft_ut_params ut_;
};

class UKFCorrection {
public:
    UKFCorrection(const double cov_v[3], const double cov_w[3], const double cov_x[3], const double cov_q[3], double alpha,
                  double beta, double kappa)
        : ut_{alpha, beta, kappa}
    {
        for (int i = 0; i < 3; ++i) { rv_[i] = cov_v[i]; rv_[3 + i] = cov_w[i]; rp_[i] = cov_x[i]; rp_[3 + i] = cov_q[i]; }
    }
    // type: ROFT_MEAS_*; measurement laid out as CartesianQuaternionMeasurement does ([v w] | [x q] | [v w x q])
    void correct(const Gaussian& pred, Gaussian& corr, int type, const double* measurement) const
    {
        double R[12];
        int k = 0;
        if (type == ROFT_MEAS_VELOCITY || type == ROFT_MEAS_POSE_VELOCITY) for (int i = 0; i < 6; ++i) R[k++] = rv_[i];
        if (type == ROFT_MEAS_POSE || type == ROFT_MEAS_POSE_VELOCITY) for (int i = 0; i < 6; ++i) R[k++] = rp_[i];
        int status = 0;
        compat::throw_if(roft_ukf_correct(pred.mean().data(), pred.covariance().data(), type, measurement, R, &ut_,
                                          corr.mean().data(), corr.covariance().data(), &status), "UKFCorrection::correct");
    }

private:
    double rv_[6], rp_[6];
    roft_ut_params ut_;
};

// ---- the whole tracker, batched ------------------------------------------------------------------------
// One instance tracks n objects; filtering_step() = ROFTFilter::filtering_step for all of them.
class ROFTFilter {
public:
    explicit ROFTFilter(const roft_config& cfg) : cfg_(cfg)
    {
        compat::throw_if(roft_engine_create(&cfg_, &e_), "ROFTFilter::ctor");
    }
    ~ROFTFilter() { roft_engine_destroy(e_); }
    ROFTFilter(const ROFTFilter&) = delete;
    ROFTFilter& operator=(const ROFTFilter&) = delete;
    int add_object(const roft_object_desc& d)
    {
        int id = -1;
        compat::throw_if(roft_object_add(e_, &d, &id), "ROFTFilter::add_object");
        return id;
    }
    void filtering_step(const std::vector<roft_frame_input>& inputs)
    {
        compat::throw_if(roft_frame_submit(e_, inputs.data(), static_cast<int>(inputs.size())), "ROFTFilter::filtering_step");
        compat::throw_if(roft_step(e_), "ROFTFilter::filtering_step");
    }
    void wait() { compat::throw_if(roft_sync(e_), "ROFTFilter::wait"); }
    void state(int obj, double pose13[13], double twist6[6])
    {
        compat::throw_if(roft_get_state(e_, obj, pose13, nullptr, twist6, nullptr), "ROFTFilter::state");
    }

private:
    roft_config cfg_;
    roft_engine* e_ = nullptr;
};

}  // namespace ROFT
