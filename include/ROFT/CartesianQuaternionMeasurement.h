// ROFT::CartesianQuaternionMeasurement -- measurement model of the pose UKF: the twist delivered by the velocity filter
// every frame and, at a lower rate and with a delay, a pose (reference:
// src/roft-lib/include/ROFT/CartesianQuaternionMeasurement.h:27-123, src/CartesianQuaternionMeasurement.cpp:92-487).
// freeze() keeps the three modes of the reference -- Standard, RepeatOnlyVelocity (second alternative of the outlier test)
// and PopBufferedMeasurement (re-sync replay of the buffered twists) -- as host-side bookkeeping; the correction itself
// (ROFT::UKFCorrection) takes the frozen measurement, its type and noise covariance to the GPU in one roft_ukf_correct.
// predictedMeasure() / innovation() are kept for callers that walk sigma points themselves.
#pragma once

#include "Sources.h"

namespace ROFT {

class CartesianQuaternionMeasurement : public bfl::MeasurementModel {
public:
    enum class MeasurementMode { Standard, RepeatOnlyVelocity, PopBufferedMeasurement };
    enum class TransformFeedback { None, RGB, DepthSegmentation };

    CartesianQuaternionMeasurement(std::shared_ptr<RobotsIO::Utils::Transform> pose_measurement,
                                   std::shared_ptr<RobotsIO::Utils::SpatialVelocity> velocity_measurement, const bool use_screw_velocity,
                                   const bool use_pose_measurement, const bool use_velocity_measurement,
                                   const Eigen::Ref<const Eigen::MatrixXd> sigma_position, const Eigen::Ref<const Eigen::MatrixXd> sigma_quaternion,
                                   const Eigen::Ref<const Eigen::MatrixXd> sigma_linear_velocity,
                                   const Eigen::Ref<const Eigen::MatrixXd> sigma_angular_velocity, const bool enable_log)
        : CartesianQuaternionMeasurement(std::move(pose_measurement), std::move(velocity_measurement), nullptr, nullptr, use_screw_velocity,
                                         use_pose_measurement, use_velocity_measurement, sigma_position, sigma_quaternion,
                                         sigma_linear_velocity, sigma_angular_velocity, false, enable_log)
    {}

    CartesianQuaternionMeasurement(std::shared_ptr<RobotsIO::Utils::Transform> pose_measurement,
                                   std::shared_ptr<RobotsIO::Utils::SpatialVelocity> velocity_measurement,
                                   std::shared_ptr<ROFT::CameraMeasurement> camera_measurement,
                                   std::shared_ptr<ROFT::ImageSegmentationMeasurement> segmentation_measurement, const bool use_screw_velocity,
                                   const bool use_pose_measurement, const bool use_velocity_measurement,
                                   const Eigen::Ref<const Eigen::MatrixXd> sigma_position, const Eigen::Ref<const Eigen::MatrixXd> sigma_quaternion,
                                   const Eigen::Ref<const Eigen::MatrixXd> sigma_linear_velocity,
                                   const Eigen::Ref<const Eigen::MatrixXd> sigma_angular_velocity, const bool wait_source_initialization,
                                   const bool enable_log)
        : pose_measurement_(std::move(pose_measurement)), velocity_measurement_(std::move(velocity_measurement)),
          camera_measurement_(std::move(camera_measurement)), segmentation_measurement_(std::move(segmentation_measurement)),
          use_pose_measurement_(use_pose_measurement), use_velocity_measurement_(use_velocity_measurement), enable_log_(enable_log)
    {
        (void)wait_source_initialization;
        if (use_screw_velocity) throw std::runtime_error(log_name_ + "::ctor. Error: the screw form of the velocity is not supported (ROFTFilter.cpp:157 uses the origin form).");
        if (use_pose_measurement_ && !pose_measurement_) throw std::runtime_error(log_name_ + "::ctor. Error: null pose source.");
        if (use_velocity_measurement_ && !velocity_measurement_) throw std::runtime_error(log_name_ + "::ctor. Error: null velocity source.");
        pose_frames_between_iterations_ = pose_measurement_ ? pose_measurement_->get_frames_between_iterations() : -1;
        // noise in measurement order: [v w] | [x q]  (cpp:49-61)
        compat::diagonal_of(sigma_linear_velocity, r_velocity_, 3);
        compat::diagonal_of(sigma_angular_velocity, r_velocity_ + 3, 3);
        compat::diagonal_of(sigma_position, r_pose_, 3);
        compat::diagonal_of(sigma_quaternion, r_pose_ + 3, 3);
    }
    virtual ~CartesianQuaternionMeasurement() = default;

    // data = MeasurementMode
    bool freeze(const bfl::Data& data = bfl::Data()) override
    {
        const MeasurementMode mode = data.has_value() ? bfl::any::any_cast<MeasurementMode>(data) : MeasurementMode::Standard;
        if (mode == MeasurementMode::PopBufferedMeasurement) {
            // replay of the twists buffered since the frame the delayed pose belongs to (cpp:97-154): at most
            // frames_between + 1 of them; an empty buffer ends the replay and keeps the current twist for the next one
            if (pose_frames_between_iterations_ > 0)
                while ((int)buffer_velocities_.size() > pose_frames_between_iterations_ + 1) buffer_velocities_.pop_front();
            if (buffer_velocities_.empty()) {
                buffer_velocities_.push_back(current_twist());
                return false;
            }
            const Twist tw = buffer_velocities_.front();
            buffer_velocities_.pop_front();
            set_last_twist(tw);
            if (is_pose_) {
                set_type(ROFT_MEAS_POSE_VELOCITY);
                is_pose_ = false;   // consumed by the first replayed step
            } else {
                set_type(ROFT_MEAS_VELOCITY);
            }
            return true;
        }
        if (mode == MeasurementMode::RepeatOnlyVelocity) {
            if (is_first_velocity_in_) set_type(ROFT_MEAS_VELOCITY);   // the current measurement without its pose part (cpp:156-174)
            return true;
        }
        // Standard
        if (use_velocity_measurement_ && velocity_measurement_->freeze(true)) {
            is_first_velocity_in_ = true;
            const double* v = velocity_measurement_->linear_velocity_origin();
            const double* w = velocity_measurement_->angular_velocity();
            for (int i = 0; i < 3; ++i) { last_twist_.v[i] = v[i]; last_twist_.v[3 + i] = w[i]; }
        }
        is_pose_ = false;
        if (use_pose_measurement_) {
            is_pose_ = pose_measurement_->freeze(false);
            if (is_pose_) last_pose_ = pose_measurement_->transform();
            else if (pose_frames_between_iterations_ < 0 && pose_measurement_->transform_received())
                while (buffer_velocities_.size() > 1) buffer_velocities_.pop_front();   // invalid pose, unknown rate (cpp:251-258)
        }
        bool valid_freeze = true;
        if (is_first_velocity_in_ && is_pose_) {
            set_type(ROFT_MEAS_POSE_VELOCITY);
            buffer_velocities_.push_back(last_twist_);
        } else if (is_first_velocity_in_) {
            set_type(ROFT_MEAS_VELOCITY);
            buffer_velocities_.push_back(last_twist_);
        } else if (is_pose_) {
            set_type(ROFT_MEAS_POSE);
        } else {
            set_type(ROFT_MEAS_NONE);
            valid_freeze = false;
        }
        if (enable_log_) {
            // `pose_measurements` = last received pose as x, axis, angle; `velocity_measurements` = v_O, w (cpp:332-345)
            Eigen::VectorXd pose_vector(7), velocity_vector(6);
            const double* q = last_pose_.quaternion();
            const double n = std::sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), sgn = q[0] < 0.0 ? -1.0 : 1.0;
            for (int i = 0; i < 3; ++i) pose_vector(i) = last_pose_.translation()[i];
            for (int i = 0; i < 3; ++i) pose_vector(3 + i) = n > 0.0 ? sgn * q[1 + i] / n : (i == 0 ? 1.0 : 0.0);   // Eigen::AngleAxisd(Quaterniond)
            pose_vector(6) = 2.0 * std::atan2(n, std::fabs(q[0]));
            for (int i = 0; i < 6; ++i) velocity_vector(i) = last_twist_.v[i];
            logger(pose_vector.transpose(), velocity_vector.transpose());
        }
        return valid_freeze;
    }
    std::pair<bool, bfl::Data> measure(const bfl::Data& = bfl::Data()) const override
    {
        return std::make_pair(type_ != ROFT_MEAS_NONE, bfl::Data(measurement_));
    }
    // h(x) per sigma column: state rows [v w x q] followed by the noise rows of getInputDescription() (cpp:357-433)
    std::pair<bool, bfl::Data> predictedMeasure(const Eigen::Ref<const Eigen::MatrixXd>& cur) const override
    {
        if (type_ == ROFT_MEAS_NONE) return std::make_pair(false, bfl::Data());
        const bool has_vel = type_ != ROFT_MEAS_POSE, has_pose = type_ != ROFT_MEAS_VELOCITY;
        Eigen::MatrixXd out(measurement_.rows(), cur.cols());
        for (std::size_t c = 0; c < cur.cols(); ++c) {
            const double v[3] = {cur(0, c), cur(1, c), cur(2, c)}, w[3] = {cur(3, c), cur(4, c), cur(5, c)};
            const double x[3] = {cur(6, c), cur(7, c), cur(8, c)}, q[4] = {cur(9, c), cur(10, c), cur(11, c), cur(12, c)};
            std::size_t row = 0, nrow = 13;
            if (has_vel) {
                // twist at the camera origin: v + w x (-x)
                const double p[3] = {-x[0], -x[1], -x[2]};
                const double cr[3] = {w[1] * p[2] - w[2] * p[1], w[2] * p[0] - w[0] * p[2], w[0] * p[1] - w[1] * p[0]};
                for (int i = 0; i < 3; ++i) { out(row + i, c) = (v[i] + cr[i]) + cur(nrow + i, c); out(row + 3 + i, c) = w[i] + cur(nrow + 3 + i, c); }
                row += 6;
                nrow += 6;
            }
            if (has_pose) {
                for (int i = 0; i < 3; ++i) out(row + i, c) = x[i] + cur(nrow + i, c);
                const double r[3] = {cur(nrow + 3, c), cur(nrow + 4, c), cur(nrow + 5, c)};
                double qo[4];
                boxplus(q, r, qo);
                for (int i = 0; i < 4; ++i) out(row + 3 + i, c) = qo[i];
            }
        }
        return std::make_pair(true, bfl::Data(std::move(out)));
    }
    // linear rows: measurement - prediction; quaternion rows: rotation vector of q_meas (x) q_pred^-1 (cpp:436-487)
    std::pair<bool, bfl::Data> innovation(const bfl::Data& predicted_measurements, const bfl::Data& measurements) const override
    {
        const Eigen::MatrixXd& p = *bfl::any::any_cast<Eigen::MatrixXd>(&predicted_measurements);
        const Eigen::MatrixXd& m = *bfl::any::any_cast<Eigen::MatrixXd>(&measurements);
        const bool has_pose = type_ == ROFT_MEAS_POSE || type_ == ROFT_MEAS_POSE_VELOCITY;
        const std::size_t nlin = p.rows() - (has_pose ? 4 : 0);
        Eigen::MatrixXd out(nlin + (has_pose ? 3 : 0), p.cols());
        for (std::size_t c = 0; c < p.cols(); ++c) {
            for (std::size_t i = 0; i < nlin; ++i) out(i, c) = m(i, 0) - p(i, c);
            if (has_pose) {
                const double a[4] = {m(nlin, 0), m(nlin + 1, 0), m(nlin + 2, 0), m(nlin + 3, 0)};
                const double b[4] = {p(nlin, c), p(nlin + 1, c), p(nlin + 2, c), p(nlin + 3, c)};
                double d[3];
                boxminus(a, b, d);
                for (int i = 0; i < 3; ++i) out(nlin + i, c) = d[i];
            }
        }
        return std::make_pair(true, bfl::Data(std::move(out)));
    }
    std::pair<bool, Eigen::MatrixXd> getNoiseCovarianceMatrix() const override
    {
        double d[12];
        const int n = noise_diagonal(d);
        Eigen::VectorXd v((std::size_t)n, 1);
        for (int i = 0; i < n; ++i) v(i) = d[i];
        return std::make_pair(type_ != ROFT_MEAS_NONE, v.asDiagonal());
    }
    bfl::VectorDescription getInputDescription() const override
    {
        return type_ == ROFT_MEAS_NONE ? bfl::VectorDescription(0, 0, 0) : bfl::VectorDescription(9, 1, type_ == ROFT_MEAS_POSE_VELOCITY ? 12 : 6);
    }
    bfl::VectorDescription getMeasurementDescription() const override
    {
        switch (type_) {
        case ROFT_MEAS_POSE_VELOCITY: return bfl::VectorDescription(9, 1, 0);
        case ROFT_MEAS_VELOCITY: return bfl::VectorDescription(6);
        case ROFT_MEAS_POSE: return bfl::VectorDescription(3, 1, 0);
        default: return bfl::VectorDescription(0, 0, 0);
        }
    }
    bool setProperty(const std::string& property) override
    {
        if (property == "reset") { is_first_velocity_in_ = false; is_pose_ = false; return true; }
        if (property == "transform_feedback_rgb") { transform_feedback_ = TransformFeedback::RGB; return true; }
        if (property == "transform_feedback_depth_segmentation") transform_feedback_ = TransformFeedback::DepthSegmentation;
        return false;
    }

    // what ROFT::UKFCorrection hands to roft_ukf_correct: ROFT_MEAS_* of the frozen measurement, its values laid out
    // [v w] | [x q] | [v w x q] and the diagonal of its noise covariance in the same order
    int measurement_type() const { return type_; }
    const double* measurement_data() const { return measurement_.data(); }
    int noise_diagonal(double out[12]) const
    {
        int k = 0;
        if (type_ == ROFT_MEAS_VELOCITY || type_ == ROFT_MEAS_POSE_VELOCITY) for (int i = 0; i < 6; ++i) out[k++] = r_velocity_[i];
        if (type_ == ROFT_MEAS_POSE || type_ == ROFT_MEAS_POSE_VELOCITY) for (int i = 0; i < 6; ++i) out[k++] = r_pose_[i];
        return k;
    }
    std::size_t buffered_velocities() const { return buffer_velocities_.size(); }

protected:
    // bfl::Logger: enable_log(path, prefix) opens these two files (cpp:535-539)
    std::vector<std::string> log_file_names(const std::string& prefix_path, const std::string& prefix_name) override
    {
        return {prefix_path + "/" + prefix_name + "pose_measurements", prefix_path + "/" + prefix_name + "velocity_measurements"};
    }

private:
    struct Twist { double v[6]; };
    Twist current_twist() const
    {
        Twist t{};
        const bool has_vel = type_ == ROFT_MEAS_VELOCITY || type_ == ROFT_MEAS_POSE_VELOCITY;
        for (int i = 0; i < 6; ++i) t.v[i] = has_vel ? measurement_(i) : last_twist_.v[i];
        return t;
    }
    void set_last_twist(const Twist& t) { last_twist_ = t; }
    void set_type(int type)
    {
        type_ = type;
        const bool has_vel = type == ROFT_MEAS_VELOCITY || type == ROFT_MEAS_POSE_VELOCITY;
        const bool has_pose = type == ROFT_MEAS_POSE || type == ROFT_MEAS_POSE_VELOCITY;
        measurement_.resize((has_vel ? 6 : 0) + (has_pose ? 7 : 0), 1);
        std::size_t k = 0;
        if (has_vel) for (int i = 0; i < 6; ++i) measurement_(k++) = last_twist_.v[i];
        if (has_pose) {
            for (int i = 0; i < 3; ++i) measurement_(k++) = last_pose_.translation()[i];
            for (int i = 0; i < 4; ++i) measurement_(k++) = last_pose_.quaternion()[i];
        }
    }
    static void quat_mul(const double a[4], const double b[4], double o[4])
    {
        o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
        o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
        o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
        o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    }
    static void boxplus(const double q[4], const double r[3], double o[4])   // exp(r) (x) q
    {
        const double n = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        if (n == 0.0) { for (int i = 0; i < 4; ++i) o[i] = q[i]; return; }
        const double s = std::sin(n / 2.0) / n, e[4] = {std::cos(n / 2.0), s * r[0], s * r[1], s * r[2]};
        quat_mul(e, q, o);
    }
    static void boxminus(const double a[4], const double b[4], double o[3])  // rotation vector of a (x) b^-1, shortest arc
    {
        const double bc[4] = {b[0], -b[1], -b[2], -b[3]};
        double p[4];
        quat_mul(a, bc, p);
        const double n = std::sqrt(p[1] * p[1] + p[2] * p[2] + p[3] * p[3]);
        if (n == 0.0) { o[0] = o[1] = o[2] = 0.0; return; }
        const double sgn = p[0] < 0.0 ? -1.0 : 1.0, k = sgn * 2.0 * std::atan2(n, std::fabs(p[0])) / n;
        o[0] = k * p[1]; o[1] = k * p[2]; o[2] = k * p[3];
    }

    std::shared_ptr<RobotsIO::Utils::Transform> pose_measurement_;
    std::shared_ptr<RobotsIO::Utils::SpatialVelocity> velocity_measurement_;
    std::shared_ptr<ROFT::CameraMeasurement> camera_measurement_;
    std::shared_ptr<ROFT::ImageSegmentationMeasurement> segmentation_measurement_;
    const bool use_pose_measurement_, use_velocity_measurement_, enable_log_;
    int pose_frames_between_iterations_ = -1;
    double r_velocity_[6], r_pose_[6];
    Eigen::MatrixXd measurement_;
    Twist last_twist_{};
    Eigen::Transform<double, 3, Eigen::Affine> last_pose_;
    std::deque<Twist> buffer_velocities_;
    int type_ = ROFT_MEAS_NONE;
    bool is_first_velocity_in_ = false, is_pose_ = false;
    TransformFeedback transform_feedback_ = TransformFeedback::None;
    const std::string log_name_ = "CartesianQuaternionMeasurement";
};

}  // namespace ROFT
