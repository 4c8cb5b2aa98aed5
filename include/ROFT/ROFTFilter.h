// ROFT::ROFTFilter -- the tracker of one object with the constructor and the filtering loop of the reference (reference:
// src/roft-lib/include/ROFT/ROFTFilter.h:38-194; ctor src/ROFTFilter.cpp:32-201; initialization_step :216-237;
// filtering_step :255-452).  The constructor takes the same sources and parameter vectors and composes the same objects;
// filtering_step() polls the sources as the reference does and hands the frame to a one-object engine of the C ABI
// (roft_frame_submit / roft_step, include/roft_engine.h), which runs the velocity stage, the flow-aided segmentation,
// the pose stage with re-sync and the outlier test on the GPU.  Many objects at once: ROFT::ROFTFilterBatch (Filters.h).
// Logging and probes as the reference has them: enable_log(path, prefix) (bfl::Logger) writes `pose_estimate`,
// `velocity_estimate` and `execution_times` (cpp:247-252, 386-394, 448-451) and -- the measurement model's own log,
// CartesianQuaternionMeasurement.cpp:332-345, 535-539 -- `pose_measurements` and `velocity_measurements`; the probes
// output_pose, output_velocity, output_segmentation and output_segmentation_refined (RobotsIO::Utils::ProbeContainer,
// cpp:396-446) are served when set, so the tail of src/roft/src/main.cpp:393-424 compiles against this class as it is.
#pragma once

#include <deque>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

#include "CartesianQuaternionModel.h"
#include "ImageOpticalFlowMeasurement.hpp"
#include "ImageSegmentationOFAidedSource.hpp"
#include "MeshResource.h"
#include "SKFCorrection.h"
#include "SpatialVelocityModel.h"
#include "UKFCorrection.h"

namespace ROFT {

// Wavefront OBJ as the reference's meshes are written (`v x y z [r g b]`, `f a//n b//n c//n`, also a/t/n and plain
// indices; polygons are fanned): src/roft-lib/meshes/DOPE/*.obj
inline void parse_obj_mesh(std::istream& in, const std::string& what, std::vector<float>& verts, std::vector<std::int32_t>& tris)
{
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
    if (verts.empty() || tris.empty()) throw std::runtime_error("load_obj_mesh: no geometry in " + what);
}
inline void load_obj_mesh(const std::string& path, std::vector<float>& verts, std::vector<std::int32_t>& tris)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("load_obj_mesh: cannot open " + path);
    parse_obj_mesh(in, path, verts, tris);
}

class ROFTFilter : public bfl::FilteringAlgorithm, public RobotsIO::Utils::ProbeContainer {
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
          sample_time_(sample_time)
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
        // the mesh the outlier test renders (ROFTFilter.cpp:186: MeshResource mesh_resource(model_parameters))
        if (model_parameters.use_internal_db() || !model_parameters.mesh_external_path().empty()) {
            std::istringstream mesh_text(MeshResource(model_parameters).as_string());
            parse_obj_mesh(mesh_text, model_parameters.use_internal_db() ? model_parameters.name() : model_parameters.mesh_external_path(), verts_, tris_);
        }
        else if (cfg_.outlier_rejection && cfg_.use_pose)
            throw std::runtime_error(log_name_ + "::ctor. Error: outlier rejection renders the object: ModelParameters::mesh_external_path is empty.");
        obj_.mesh = roft_mesh{verts_.data(), (int)(verts_.size() / 3), tris_.data(), (int)(tris_.size() / 3)};
        // the measurement model of the pose filter logs what it was fed (cpp:157-160: constructed with enable_log and
        // enabled right away; the filter's own log waits for enable_log(), src/roft/src/main.cpp:418-419)
        if (enable_log) measurement_log_.enable_log(log_path, log_prefix);
    }

    virtual ~ROFTFilter()
    {
        // ROFT_FILTER_TIMING=1: what the integer milliseconds of `execution_times` (the reference's resolution, cpp:463) hide
        if (std::getenv("ROFT_FILTER_TIMING") && timed_frames_ > 0)
            std::printf("ROFTFilter: %ld frames, %.1f us per frame outside data loading (submit + step + state read-back: %.1f us; polling the "
                        "sources: %.1f us), %s inputs\n", timed_frames_, exec_us_ / timed_frames_, engine_us_ / timed_frames_,
                        sources_us_ / timed_frames_, in_place_frames_ == timed_frames_ ? "in-place (pinned)" : "staged (HOST)");
        if (engine_) roft_engine_destroy(engine_);
        held_.clear();
    }
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

    // pose (x, axis, angle) of a unit quaternion (w, x, y, z) as Eigen::AngleAxisd(Quaterniond) returns it (cpp:389-394):
    // angle = 2 atan2(|vec|, |w|) in [0, pi], axis = vec / |vec| with the sign of w, (1, 0, 0) for the identity
    static void axis_angle(const double q[4], double axis[3], double& angle)
    {
        const double n = std::sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        axis[0] = 1.0; axis[1] = axis[2] = 0.0;
        const double sgn = q[0] < 0.0 ? -1.0 : 1.0;
        if (n > 0.0) for (int i = 0; i < 3; ++i) axis[i] = sgn * q[1 + i] / n;
        angle = 2.0 * std::atan2(n, std::fabs(q[0]));
    }

    const bfl::Gaussian& pose_belief() const { return p_corr_belief_; }         // mean: v w x q(w x y z)
    const bfl::Gaussian& velocity_belief() const { return v_corr_belief_; }     // mean: v_O w
    const roft_object_output& last_output() const { return out_; }
    long frames() const { return frames_; }
    // one iteration of the loop run() drives
    void step() { filtering_step(); }

protected:
    std::vector<std::string> log_file_names(const std::string& prefix_path, const std::string& prefix_name) override
    {
        return {prefix_path + "/" + prefix_name + "pose_estimate", prefix_path + "/" + prefix_name + "velocity_estimate",
                prefix_path + "/" + prefix_name + "execution_times"};
    }

    void filtering_step() override
    {
        using clock = std::chrono::steady_clock;
        auto ms_since = [](clock::time_point t0) { return (double)std::chrono::duration_cast<std::chrono::milliseconds>(clock::now() - t0).count(); };
        const auto time0 = clock::now();
        if (!camera_->freeze(CameraMeasurementType::RGBD)) {   // cannot continue without a continuous depth stream (cpp:261-266)
            teardown();
            return;
        }
        const double rgbd_load_time = ms_since(time0);
        const auto std_time_0 = clock::now();   // start_time_count(): the RGB-D loading time is excluded (cpp:267-270)
        (void)compat::take_loading_us();
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
        if (new_mask) last_mask_ = mask.clone();
        if (!mask_received_) return;   // the segmentation is not available yet: nothing to filter (cpp:291)

        // The engine keeps REFERRING to a flow handed over in place -- a delayed mask is chased through the flows of the frames
        // since it was taken -- so the buffer must hold this frame's flow for as long as the filter holds it (held_).  A source
        // that does not promise that (one flow matrix rewritten every frame: cv::Mat copies share the buffer), or whose buffer is
        // one the filter already holds, is copied here, once, as the reference's OF-aided source clones the flows it buffers
        // (ImageSegmentationOFAidedSource.hpp:200-209).  Depth needs no such care: the matrix type of the camera tuple is
        // copy-on-write (Compat.h); a mask is read once, by the step below, before the source can touch it again.
        if (valid_flow && (!flow_source_->flow_buffers_are_immutable() || holds_buffer(flow.data))) flow = flow.clone();
        roft_frame_input in{};
        in.dt = elapsed;
        in.depth = depth.data();
        in.flow = valid_flow ? flow.data : nullptr;
        in.mask = new_mask ? mask.data : nullptr;
        // Images that live in the library's pinned, device-mapped pool (the stand-in matrix types allocate image-sized buffers
        // there: Compat.h) are handed over as they are and read in place over the bus -- a frame costs the few hundred KB the
        // kernels touch instead of a 5.5 MB upload; this filter keeps them alive for the engine's retention window (below).
        static const bool staged_only = std::getenv("ROFT_FACADE_STAGED") != nullptr;   // (A/B: force the HOST upload path)
        const bool in_place = !staged_only && roft_host_is_pinned(in.depth) && (!in.flow || roft_host_is_pinned(in.flow)) && (!in.mask || roft_host_is_pinned(in.mask));
        in.mem_kind = in_place ? ROFT_MEM_DEVICE : ROFT_MEM_HOST;
        if (cfg_.use_pose && pose_measurement_->freeze(false)) {
            last_pose_ = pose_measurement_->transform();
            in.pose_valid = 1;
            for (int i = 0; i < 3; ++i) in.pose_x[i] = last_pose_.translation()[i];
            for (int i = 0; i < 4; ++i) in.pose_q[i] = last_pose_.quaternion()[i];
        }
        const auto eng_t0 = clock::now();
        sources_us_ += std::chrono::duration<double, std::micro>(eng_t0 - std_time_0).count() - compat::loading_us_counter();
        compat::throw_if(roft_frame_submit(engine_, &in, 1), "ROFTFilter::filtering_step");
        compat::throw_if(roft_step(engine_), "ROFTFilter::filtering_step");
        compat::throw_if(roft_get_state(engine_, 0, p_corr_belief_.mean().data(), p_corr_belief_.covariance().data(), v_corr_belief_.mean().data(),
                                        v_corr_belief_.covariance().data()), "ROFTFilter::filtering_step");
        compat::throw_if(roft_get_outputs(engine_, &out_, 1), "ROFTFilter::filtering_step");
        ++frames_;
        engine_us_ += std::chrono::duration<double, std::micro>(clock::now() - eng_t0).count();
        if (in_place) {
            ++in_place_frames_;
            // (depth of frame k is read again by frame k + 1, a flow by the masks chased through it for up to
            //  mask_frames_between frames: roft_engine_retain_frames)
            held_.push_back(Held{cam_data, valid_flow ? flow : cv::Mat(), new_mask ? mask : cv::Mat()});
            while ((int)held_.size() > roft_engine_retain_frames(engine_) + 1) held_.pop_front();
        }
        // stop_time_count(): what follows is "for debugging purposes only" (cpp:369-384); the time the sources spent
        // loading data from disk is taken out of the execution time and reported next to the RGB-D loading time
        double exec_time = ms_since(std_time_0);
        const double exec_us_now = std::chrono::duration<double, std::micro>(clock::now() - std_time_0).count();
        const double loading_us = compat::take_loading_us();
        const double load_time = flow_source_->get_data_loading_time() + segmentation_source_->get_data_loading_time();
        segmentation_source_->reset_data_loading_time();
        exec_time -= load_time;
        exec_us_ += exec_us_now - loading_us;   // (the sources report their loading time in whole milliseconds, like the reference's)
        ++timed_frames_;

        // `pose_estimate` = v w x axis angle, `velocity_estimate` = v_O w, `execution_times` (cpp:386-394, 448-451)
        Eigen::VectorXd p_mean(13), v_mean(6), execution_time(2);
        for (int i = 0; i < 9; ++i) p_mean(i) = p_corr_belief_.mean(i);
        double axis[3], angle;
        axis_angle(p_corr_belief_.mean().data() + 9, axis, angle);
        for (int i = 0; i < 3; ++i) p_mean(9 + i) = axis[i];
        p_mean(12) = angle;
        for (int i = 0; i < 6; ++i) v_mean(i) = v_corr_belief_.mean(i);
        if (is_probe("output_pose")) {
            Eigen::VectorXd pose(7);
            for (int i = 0; i < 7; ++i) pose(i) = p_mean(6 + i);
            get_probe("output_pose").set_data(pose);
        }
        if (is_probe("output_velocity")) get_probe("output_velocity").set_data(v_mean);
        if (is_probe("output_segmentation") && is_probe("output_segmentation_refined")) probe_segmentation(cam_data);
        execution_time(0) = exec_time;
        execution_time(1) = load_time + rgbd_load_time;
        logger(p_mean.transpose(), v_mean.transpose(), execution_time.transpose());
        // CartesianQuaternionMeasurement::freeze(Standard), cpp:332-345: the last received pose as x, axis, angle and the
        // twist handed over by the velocity filter this frame
        {
            Eigen::VectorXd pose_vector(7), velocity_vector(6);
            for (int i = 0; i < 3; ++i) pose_vector(i) = last_pose_.translation()[i];
            axis_angle(last_pose_.quaternion(), axis, angle);
            for (int i = 0; i < 3; ++i) pose_vector(3 + i) = axis[i];
            pose_vector(6) = angle;
            for (int i = 0; i < 6; ++i) velocity_vector(i) = out_.twist[i];
            measurement_log_.logger(pose_vector.transpose(), velocity_vector.transpose());
        }
    }

private:
    // bfl::Logger of the pose filter's measurement model (the engine walks that model's state machine itself)
    class MeasurementLog : public bfl::Logger {
    protected:
        std::vector<std::string> log_file_names(const std::string& prefix_path, const std::string& prefix_name) override
        {
            return {prefix_path + "/" + prefix_name + "pose_measurements", prefix_path + "/" + prefix_name + "velocity_measurements"};
        }
    };

    // Debug images of the masks (cpp:405-446).  output_segmentation_refined: the camera image with the mask the filter
    // works on -- propagated through the optical flow to this frame (roft_get_mask) -- blended in green (alpha 0.8);
    // output_segmentation: the camera image with the outline of the last mask the source delivered in red (the reference
    // draws cv::findContours polygons 4 pixels thick; here every mask pixel within 2 pixels of the outside).  A camera
    // without a colour image gets a black one.
    void probe_segmentation(const bfl::Data& cam_data)
    {
        const int W = (int)camera_parameters_.width(), H = (int)camera_parameters_.height();
        const cv::Mat& rgb_in = std::get<1>(*bfl::any::any_cast<CameraMeasurement::CameraMeasurementTuple>(&cam_data));
        cv::Mat rgb = (rgb_in.type() == CV_8UC3 && rgb_in.rows == H && rgb_in.cols == W) ? rgb_in.clone() : cv::Mat(H, W, CV_8UC3);
        cv::Mat refined = rgb.clone();
        std::vector<std::uint8_t> m((std::size_t)W * H);
        compat::throw_if(roft_get_mask(engine_, 0, m.data()), "ROFTFilter::filtering_step");
        for (std::size_t p = 0; p < m.size(); ++p)
            if (m[p]) {
                const unsigned char g[3] = {0, 255, 0};
                for (int k = 0; k < 3; ++k) refined.data[3 * p + k] = (unsigned char)std::lround(0.8 * g[k] + 0.2 * refined.data[3 * p + k]);
            } else {
                for (int k = 0; k < 3; ++k) refined.data[3 * p + k] = (unsigned char)std::lround(0.8 * refined.data[3 * p + k] + 0.2 * refined.data[3 * p + k]);
            }
        if (!last_mask_.empty())
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    if (!last_mask_.data[(std::size_t)y * W + x]) continue;
                    bool edge = false;
                    for (int dy = -2; dy <= 2 && !edge; ++dy)
                        for (int dx = -2; dx <= 2 && !edge; ++dx) {
                            const int yy = y + dy, xx = x + dx;
                            edge = yy < 0 || yy >= H || xx < 0 || xx >= W || !last_mask_.data[(std::size_t)yy * W + xx];
                        }
                    if (edge) { unsigned char* px = rgb.data + 3 * ((std::size_t)y * W + x); px[0] = 0; px[1] = 0; px[2] = 255; }
                }
        get_probe("output_segmentation").set_data(rgb);
        get_probe("output_segmentation_refined").set_data(refined);
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
    // frames handed to the engine in place (pinned pool): camera tuple (depth), flow and mask of the retention window
    struct Held { bfl::Data cam; cv::Mat flow, mask; };
    std::deque<Held> held_;
    bool holds_buffer(const void* p) const
    {
        for (const Held& h : held_)
            if (p && (h.flow.data == p || h.mask.data == p)) return true;
        return false;
    }
    double exec_us_ = 0.0, engine_us_ = 0.0, sources_us_ = 0.0;
    long timed_frames_ = 0, in_place_frames_ = 0;
    roft_object_output out_{};
    const double sample_time_;
    double last_camera_stamp_ = -1;
    bool mask_received_ = false;
    long frames_ = 0;
    cv::Mat last_mask_;                                      // last mask delivered by the source (cpp:438-442)
    Eigen::Transform<double, 3, Eigen::Affine> last_pose_;   // last pose delivered by the pose source
    MeasurementLog measurement_log_;
    const std::string log_name_ = "ROFTFilter";
};

}  // namespace ROFT
