// ROFT::ROFTFilter -- the tracker of one object with the constructor and the filtering loop of the reference (reference:
// src/roft-lib/include/ROFT/ROFTFilter.h:38-194; ctor src/ROFTFilter.cpp:32-201; initialization_step :216-237;
// filtering_step :255-452).  The constructor takes the same sources and parameter vectors and composes the same objects;
// filtering_step() polls the sources as the reference does and hands the frame to a one-object engine of the C ABI
// (roft_frame_submit / roft_step, include/roft_engine.h), which runs the velocity stage, the flow-aided segmentation,
// the pose stage with re-sync and the outlier test on the GPU.  Many objects at once: ROFT::ROFTFilterBatch (Filters.h).
#pragma once

#include <chrono>
#include <cstdio>
#include <fstream>
#include <sstream>

#include "CartesianQuaternionModel.h"
#include "ImageOpticalFlowMeasurement.hpp"
#include "ImageSegmentationOFAidedSource.hpp"
#include "SKFCorrection.h"
#include "SpatialVelocityModel.h"
#include "UKFCorrection.h"

namespace ROFT {

// Wavefront OBJ as the reference's meshes are written (`v x y z [r g b]`, `f a//n b//n c//n`, also a/t/n and plain
// indices; polygons are fanned): src/roft-lib/meshes/DOPE/*.obj
inline void load_obj_mesh(const std::string& path, std::vector<float>& verts, std::vector<std::int32_t>& tris)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("load_obj_mesh: cannot open " + path);
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string tag;
        ls >> tag;
        if (tag == "v") {
            float x, y, z;
            if (ls >> x >> y >> z) { verts.push_back(x); verts.push_back(y); verts.push_back(z); }
        } else if (tag == "f") {
            std::vector<std::int32_t> idx;
            std::string tok;
            while (ls >> tok) {
                const long i = std::strtol(tok.c_str(), nullptr, 10);
                idx.push_back((std::int32_t)(i > 0 ? i - 1 : (long)verts.size() / 3 + i));
            }
            for (std::size_t k = 1; k + 1 < idx.size(); ++k) { tris.push_back(idx[0]); tris.push_back(idx[k]); tris.push_back(idx[k + 1]); }
        }
    }
    if (verts.empty() || tris.empty()) throw std::runtime_error("load_obj_mesh: no geometry in " + path);
}

class ROFTFilter : public bfl::FilteringAlgorithm {
public:
    ROFTFilter(std::shared_ptr<ROFT::CameraMeasurement> camera_measurement, std::shared_ptr<RobotsIO::Utils::Segmentation> segmentation_source,
               std::shared_ptr<ROFT::ImageOpticalFlowSource> flow_source, std::shared_ptr<RobotsIO::Utils::Transform> pose_measurement,
               const ModelParameters& model_parameters, const Eigen::Ref<const Eigen::VectorXd>& initial_condition_p,
               const Eigen::Ref<const Eigen::VectorXd>& initial_covariance_p, const Eigen::Ref<const Eigen::VectorXd>& model_covariance_p,
               const Eigen::Ref<const Eigen::VectorXd>& measurement_covariance_p, const Eigen::Ref<const Eigen::VectorXd> initial_condition_v,
               const Eigen::Ref<const Eigen::VectorXd> initial_covariance_v, const Eigen::Ref<const Eigen::VectorXd> model_covariance_v,
               const Eigen::Ref<const Eigen::VectorXd> measurement_covariance_v, const double& ut_alpha, const double& ut_beta,
               const double& ut_kappa, const double& sample_time, const bool pose_meas, const bool pose_resync,
               const bool pose_outlier_rejection, const bool pose_outlier_rejection_gain, const bool velocity_meas, const bool flow_weighting,
               const bool flow_aided_segmentation, const double& maximum_depth, const double& subsampling_radius, const bool enable_log,
               const std::string& log_path, const std::string& log_prefix)
        : p_corr_belief_(9, 1, true), v_corr_belief_(6, 0, false), camera_(std::move(camera_measurement)),
          segmentation_source_(std::move(segmentation_source)), flow_source_(std::move(flow_source)), pose_measurement_(std::move(pose_measurement)),
          sample_time_(sample_time), enable_log_(enable_log), log_path_(log_path), log_prefix_(log_prefix)
    {
        (void)pose_outlier_rejection_gain;   // a bool in the reference as well: the gain is 1 (ROFTFilter.h:64)
        if (!camera_ || !segmentation_source_ || !flow_source_) throw std::runtime_error(log_name_ + "::ctor. Error: null source.");
        if (initial_condition_p.size() != 13 || initial_covariance_p.size() != 12 || model_covariance_p.size() != 6 ||
            measurement_covariance_p.size() != 12 || initial_condition_v.size() != 6 || initial_covariance_v.size() != 6 ||
            model_covariance_v.size() != 6 || measurement_covariance_v.size() != 2)
            throw std::runtime_error(log_name_ + "::ctor. Error: parameter vectors of unexpected size.");
        bool valid = false;
        std::tie(valid, camera_parameters_) = camera_->camera_parameters();
        if (!valid) throw std::runtime_error(log_name_ + "::ctor. Error: cannot get camera parameters.");

        const int type = flow_source_->get_matrix_type() == CV_16SC2 ? ROFT_FLOW_S16C2 : ROFT_FLOW_F32C2;
        compat::throw_if(roft_default_config(&cfg_, (int)camera_parameters_.width(), (int)camera_parameters_.height(), type), "ROFTFilter::ctor");
        cfg_.cam.fx = camera_parameters_.fx(); cfg_.cam.fy = camera_parameters_.fy();
        cfg_.cam.cx = camera_parameters_.cx(); cfg_.cam.cy = camera_parameters_.cy();
        cfg_.flow_grid = (int)flow_source_->get_grid_size();
        cfg_.flow_scale = flow_source_->get_scaling_factor();
        cfg_.sample_time = sample_time;
        cfg_.ut = roft_ut_params{ut_alpha, ut_beta, ut_kappa};
        cfg_.depth_maximum = maximum_depth;
        cfg_.subsampling_radius = subsampling_radius;
        cfg_.flow_weighting = flow_weighting;
        cfg_.use_pose = pose_meas && pose_measurement_;
        cfg_.use_pose_resync = pose_resync;
        cfg_.use_velocity = velocity_meas;
        cfg_.outlier_rejection = pose_outlier_rejection;
        cfg_.flow_aided_segmentation = flow_aided_segmentation;
        cfg_.mask_frames_between = segmentation_source_->get_frames_between_iterations();
        cfg_.pose_frames_between = pose_measurement_ ? pose_measurement_->get_frames_between_iterations() : 0;
        if (cfg_.pose_frames_between < 0) cfg_.pose_frames_between = 0;
        cfg_.max_objects = 1;

        compat::throw_if(roft_default_object(&obj_), "ROFTFilter::ctor");
        for (int i = 0; i < 13; ++i) obj_.p_mean0[i] = initial_condition_p(i);          // v w x q(w x y z)   (cpp:76-79)
        for (int i = 0; i < 12; ++i) obj_.p_cov0_diag[i] = initial_covariance_p(i);
        for (int i = 0; i < 6; ++i) { obj_.v_mean0[i] = initial_condition_v(i); obj_.v_cov0_diag[i] = initial_covariance_v(i); obj_.v_q_diag[i] = model_covariance_v(i); }
        for (int i = 0; i < 3; ++i) {
            obj_.p_sigma_ang_vel[i] = model_covariance_p(i);         // head<3>: sigma of the angular velocity (cpp:89)
            obj_.p_psd_lin_acc[i] = model_covariance_p(3 + i);       // tail<3>: PSD of the linear acceleration (cpp:90)
            obj_.p_meas_cov_v[i] = measurement_covariance_p(i);      // (cpp:96-99)
            obj_.p_meas_cov_w[i] = measurement_covariance_p(3 + i);
            obj_.p_meas_cov_x[i] = measurement_covariance_p(6 + i);
            obj_.p_meas_cov_q[i] = measurement_covariance_p(9 + i);
        }
        obj_.v_meas_cov_flow[0] = measurement_covariance_v(0);
        obj_.v_meas_cov_flow[1] = measurement_covariance_v(1);
        if (!model_parameters.mesh_external_path().empty()) load_obj_mesh(model_parameters.mesh_external_path(), verts_, tris_);
        else if (cfg_.outlier_rejection && cfg_.use_pose)
            throw std::runtime_error(log_name_ + "::ctor. Error: outlier rejection renders the object: ModelParameters::mesh_external_path is empty.");
        obj_.mesh = roft_mesh{verts_.data(), (int)(verts_.size() / 3), tris_.data(), (int)(tris_.size() / 3)};
    }

    virtual ~ROFTFilter() { if (engine_) roft_engine_destroy(engine_); }
    ROFTFilter(const ROFTFilter&) = delete;
    ROFTFilter& operator=(const ROFTFilter&) = delete;

    bool run_condition() override { return true; }

    // beliefs, sources and engine back to their initial state (cpp:216-237)
    bool initialization_step() override
    {
        if (engine_) { roft_engine_destroy(engine_); engine_ = nullptr; }
        compat::throw_if(roft_engine_create(&cfg_, &engine_), "ROFTFilter::initialization_step");
        int id = -1;
        compat::throw_if(roft_object_add(engine_, &obj_, &id), "ROFTFilter::initialization_step");
        for (int i = 0; i < 13; ++i) p_corr_belief_.mean(i) = obj_.p_mean0[i];
        for (int i = 0; i < 6; ++i) v_corr_belief_.mean(i) = obj_.v_mean0[i];
        segmentation_source_->reset();
        flow_source_->reset();
        camera_->reset();
        last_camera_stamp_ = -1;
        mask_received_ = false;
        frames_ = 0;
        return true;
    }
    bool skip(const std::string&, const bool) override { return false; }

    const bfl::Gaussian& pose_belief() const { return p_corr_belief_; }         // mean: v w x q(w x y z)
    const bfl::Gaussian& velocity_belief() const { return v_corr_belief_; }     // mean: v_O w
    const roft_object_output& last_output() const { return out_; }
    long frames() const { return frames_; }
    // one iteration of the loop run() drives
    void step() { filtering_step(); }

protected:
    void filtering_step() override
    {
        if (!camera_->freeze(CameraMeasurementType::RGBD)) {   // cannot continue without a continuous depth stream (cpp:261-266)
            teardown();
            return;
        }
        bool valid = false;
        bfl::Data cam_data;
        std::tie(valid, cam_data) = camera_->measure();
        const Eigen::MatrixXf& depth = std::get<2>(*bfl::any::any_cast<CameraMeasurement::CameraMeasurementTuple>(&cam_data));
        // elapsed time from the RGB stamps once two are known (cpp:273-279)
        double elapsed = sample_time_, stamp = 0.0;
        bool has_stamp = false;
        std::tie(has_stamp, stamp) = camera_->camera_time_stamp_rgb();
        if (has_stamp && last_camera_stamp_ != -1) elapsed = stamp - last_camera_stamp_;
        if (has_stamp) last_camera_stamp_ = stamp;

        if (flow_source_->is_stepping_required()) flow_source_->step_frame();
        bool valid_flow = false;
        cv::Mat flow;
        std::tie(valid_flow, flow) = flow_source_->flow(false);
        if (segmentation_source_->is_stepping_required()) segmentation_source_->step_frame();
        bool new_mask = false;
        cv::Mat mask;
        std::tie(new_mask, mask) = segmentation_source_->segmentation(false);
        mask_received_ = mask_received_ || new_mask;
        if (!mask_received_) return;   // the segmentation is not available yet: nothing to filter (cpp:291)

        roft_frame_input in{};
        in.dt = elapsed;
        in.depth = depth.data();
        in.flow = valid_flow ? flow.data : nullptr;
        in.mask = new_mask ? mask.data : nullptr;
        in.mem_kind = ROFT_MEM_HOST;
        if (cfg_.use_pose && pose_measurement_->freeze(false)) {
            const auto T = pose_measurement_->transform();
            in.pose_valid = 1;
            for (int i = 0; i < 3; ++i) in.pose_x[i] = T.translation()[i];
            for (int i = 0; i < 4; ++i) in.pose_q[i] = T.quaternion()[i];
        }
        compat::throw_if(roft_frame_submit(engine_, &in, 1), "ROFTFilter::filtering_step");
        compat::throw_if(roft_step(engine_), "ROFTFilter::filtering_step");
        compat::throw_if(roft_get_state(engine_, 0, p_corr_belief_.mean().data(), p_corr_belief_.covariance().data(), v_corr_belief_.mean().data(),
                                        v_corr_belief_.covariance().data()), "ROFTFilter::filtering_step");
        compat::throw_if(roft_get_outputs(engine_, &out_, 1), "ROFTFilter::filtering_step");
        ++frames_;
        if (enable_log_) log_row();
    }

private:
    // `pose_estimate` = v w x axis angle, `velocity_estimate` = v_O w, one row per frame (cpp:386-394, 448-451)
    void log_row()
    {
        const double* m = p_corr_belief_.mean().data();
        const double w = m[9], n = std::sqrt(m[10] * m[10] + m[11] * m[11] + m[12] * m[12]);
        double axis[3] = {1.0, 0.0, 0.0};
        const double sgn = w < 0.0 ? -1.0 : 1.0;
        if (n > 0.0) for (int i = 0; i < 3; ++i) axis[i] = sgn * m[10 + i] / n;
        const double angle = 2.0 * std::atan2(n, std::fabs(w));
        std::ofstream fp(log_path_ + "/" + log_prefix_ + "pose_estimate.txt", std::ios::app), fv(log_path_ + "/" + log_prefix_ + "velocity_estimate.txt", std::ios::app);
        for (int i = 0; i < 9; ++i) fp << m[i] << " ";
        fp << axis[0] << " " << axis[1] << " " << axis[2] << " " << angle << "\n";
        for (int i = 0; i < 6; ++i) fv << v_corr_belief_.mean(i) << (i < 5 ? " " : "\n");
    }

    bfl::Gaussian p_corr_belief_, v_corr_belief_;
    std::shared_ptr<ROFT::CameraMeasurement> camera_;
    std::shared_ptr<RobotsIO::Utils::Segmentation> segmentation_source_;
    std::shared_ptr<ROFT::ImageOpticalFlowSource> flow_source_;
    std::shared_ptr<RobotsIO::Utils::Transform> pose_measurement_;
    RobotsIO::Camera::CameraParameters camera_parameters_;
    roft_config cfg_{};
    roft_object_desc obj_{};
    std::vector<float> verts_;
    std::vector<std::int32_t> tris_;
    roft_engine* engine_ = nullptr;
    roft_object_output out_{};
    const double sample_time_;
    double last_camera_stamp_ = -1;
    bool mask_received_ = false;
    long frames_ = 0;
    const bool enable_log_;
    const std::string log_path_, log_prefix_;
    const std::string log_name_ = "ROFTFilter";
};

}  // namespace ROFT
