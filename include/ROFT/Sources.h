// Sources.h -- the input side of the filter as the reference's classes see it:
//   ROFT::CameraMeasurement            src/roft-lib/include/ROFT/CameraMeasurement.h:27-92, src/CameraMeasurement.cpp:28-90
//   ROFT::ImageSegmentationMeasurement src/roft-lib/include/ROFT/ImageSegmentationMeasurement.h:27-66, src/...cpp:30-75
//   ROFT::ImageOpticalFlowSource       src/roft-lib/include/ROFT/ImageOpticalFlowSource.h:19-48
//   ROFT::ImageOpticalFlowNVOF         src/roft-lib/include/ROFT/ImageOpticalFlowNVOF.h:24-90 (here: the HIP Lucas-Kanade producer)
//   ROFT::ModelParameters              src/roft-lib/include/ROFT/ModelParameters.h:19-38
// These are host-side adapters (no arithmetic of the hot path except the `> 1 -> 255` threshold of the segmentation
// measurement, ImageSegmentationMeasurement.cpp:65, which the engine applies itself when it ingests a mask).
#pragma once

#include "Compat.h"

namespace ROFT {

enum class CameraMeasurementType { RGB, D, RGBD, PC, RGBPC };

class CameraMeasurement : public bfl::MeasurementModel {
public:
    using CameraMeasurementTuple = std::tuple<Eigen::Transform<double, 3, Eigen::Affine>, cv::Mat, Eigen::MatrixXf>;

    explicit CameraMeasurement(std::shared_ptr<RobotsIO::Camera::Camera> camera) : camera_(std::move(camera))
    {
        if (!camera_) throw std::runtime_error("CameraMeasurement::ctor. Error: null camera.");
    }
    // steps an offline camera and latches the images of the requested type
    bool freeze(const bfl::Data& type = bfl::Data()) override
    {
        if (!camera_->step_frame()) return false;
        measure_type_ = type.has_value() ? bfl::any::any_cast<CameraMeasurementType>(type) : CameraMeasurementType::RGBD;
        const bool use_rgb = measure_type_ == CameraMeasurementType::RGB || measure_type_ == CameraMeasurementType::RGBD;
        const bool use_d = measure_type_ == CameraMeasurementType::D || measure_type_ == CameraMeasurementType::RGBD;
        bool valid = false;
        if (use_rgb) {
            // (the filter's arithmetic never reads the colour image; a camera without one still delivers depth)
            std::tie(valid, rgb_) = camera_->rgb(true);
            std::tie(is_time_stamp_rgb_, time_stamp_rgb_) = camera_->time_stamp_rgb();
        }
        if (use_d) {
            std::tie(valid, depth_) = camera_->depth(true);
            if (!valid) return false;
            std::tie(is_time_stamp_depth_, time_stamp_depth_) = camera_->time_stamp_depth();
        }
        std::tie(valid, pose_) = camera_->pose(true);
        if (!valid) return false;
        measurement_available_ = true;
        return true;
    }
    std::pair<bool, bfl::Data> measure(const bfl::Data& = bfl::Data()) const override
    {
        return std::make_pair(measurement_available_, bfl::Data(std::make_tuple(pose_, rgb_, depth_)));
    }
    std::pair<bool, bfl::Data> predictedMeasure(const Eigen::Ref<const Eigen::MatrixXd>&) const override
    {
        throw std::runtime_error("CameraMeasurement::predictedMeasure. Not implemented.");
    }
    std::pair<bool, bfl::Data> innovation(const bfl::Data&, const bfl::Data&) const override
    {
        throw std::runtime_error("CameraMeasurement::innovation. Not implemented.");
    }
    std::pair<bool, RobotsIO::Camera::CameraParameters> camera_parameters() const { return camera_->parameters(); }
    std::pair<bool, double> camera_time_stamp_rgb() const { return {is_time_stamp_rgb_, time_stamp_rgb_}; }
    std::pair<bool, double> camera_time_stamp_depth() const { return {is_time_stamp_depth_, time_stamp_depth_}; }
    std::int32_t camera_frame_index() const { return camera_->frame_index(); }
    void reset() const { camera_->reset(); }

private:
    std::shared_ptr<RobotsIO::Camera::Camera> camera_;
    cv::Mat rgb_;
    Eigen::MatrixXf depth_;
    Eigen::Transform<double, 3, Eigen::Affine> pose_;
    double time_stamp_rgb_ = 0.0, time_stamp_depth_ = 0.0;
    bool is_time_stamp_rgb_ = false, is_time_stamp_depth_ = false;
    CameraMeasurementType measure_type_ = CameraMeasurementType::RGBD;
    bool measurement_available_ = false;
};

class ImageSegmentationMeasurement : public bfl::MeasurementModel {
public:
    ImageSegmentationMeasurement(std::shared_ptr<RobotsIO::Utils::Segmentation> segmentation_source,
                                 std::shared_ptr<ROFT::CameraMeasurement> camera_measurement, const std::size_t& width = 0,
                                 const std::size_t& height = 0)
        : segmentation_source_(std::move(segmentation_source)), camera_(std::move(camera_measurement)), width_(width), height_(height)
    {
        if (!segmentation_source_) throw std::runtime_error("ImageSegmentationMeasurement::ctor. Error: null segmentation source.");
        if (width_ != 0 || height_ != 0) throw std::runtime_error("ImageSegmentationMeasurement::ctor. Error: resizing the mask is not supported.");
    }
    // polls the source; a new mask is latched and binarised `> 1 -> 255` (cpp:57-65)
    bool freeze(const bfl::Data& = bfl::Data()) override
    {
        if (camera_) {
            bool valid = false;
            bfl::Data camera_data;
            std::tie(valid, camera_data) = camera_->measure();
            if (!valid) return false;
            double stamp = 0.0;
            std::tie(std::ignore, stamp) = camera_->camera_time_stamp_rgb();
            segmentation_source_->set_rgb_image(std::get<1>(bfl::any::any_cast<CameraMeasurement::CameraMeasurementTuple>(camera_data)), stamp);
        }
        if (segmentation_source_->is_stepping_required()) segmentation_source_->step_frame();
        cv::Mat segmentation;
        std::tie(new_segmentation_, segmentation) = segmentation_source_->segmentation(false);
        if (new_segmentation_) {
            segmentation_available_ = true;
            segmentation_ = segmentation.clone();
            for (std::size_t i = 0; i < segmentation_.total(); ++i) segmentation_.data[i] = segmentation_.data[i] > 1 ? 255 : 0;
        }
        return segmentation_available_;
    }
    std::pair<bool, bfl::Data> measure(const bfl::Data& = bfl::Data()) const override
    {
        return std::make_pair(segmentation_available_, bfl::Data(std::make_pair(new_segmentation_, segmentation_)));
    }
    std::pair<bool, bfl::Data> predictedMeasure(const Eigen::Ref<const Eigen::MatrixXd>&) const override
    {
        throw std::runtime_error("ImageSegmentationMeasurement::predictedMeasure. Not implemented.");
    }
    std::pair<bool, bfl::Data> innovation(const bfl::Data&, const bfl::Data&) const override
    {
        throw std::runtime_error("ImageSegmentationMeasurement::innovation. Not implemented.");
    }
    void reset() { segmentation_available_ = false; segmentation_source_->reset(); }
    void reset_data_loading_time() { segmentation_source_->reset_data_loading_time(); }
    double get_data_loading_time() const { return segmentation_source_->get_data_loading_time(); }

protected:
    std::shared_ptr<RobotsIO::Utils::Segmentation> segmentation_source_;
    std::shared_ptr<ROFT::CameraMeasurement> camera_;
    std::size_t width_, height_;
    cv::Mat segmentation_;
    bool segmentation_available_ = false;   // a mask was received at least once
    bool new_segmentation_ = false;         // the latched mask arrived with this frame
};

class ImageOpticalFlowSource {
public:
    virtual ~ImageOpticalFlowSource() = default;
    virtual bool reset() { return true; }
    virtual bool step_frame() { return true; }
    virtual bool is_stepping_required() const = 0;
    virtual double get_data_loading_time() const { return 0.0; }
    virtual std::tuple<bool, cv::Mat> flow(const bool& blocking) = 0;
    virtual std::size_t get_grid_size() const = 0;      // a flow frame is per pixel or per grid x grid block
    virtual float get_scaling_factor() const = 0;       // values of a flow frame are pixels times this factor
    virtual int get_matrix_type() const = 0;            // CV_32FC2 or CV_16SC2
    // (not in the reference's interface) true: every flow() frame lives in a buffer of its own that this source never writes
    // again -- ROFT::ROFTFilter may then hand the buffer to the engine as it is and keep referring to it for the frames the flow
    // stays in use (cv::Mat shares its buffer between copies and is NOT copy-on-write).  The default is the safe answer for a
    // source the filter knows nothing about: the filter then works on a copy of its own, as the reference's OF-aided source
    // does (`flow.clone()`, ImageSegmentationOFAidedSource.hpp:200-209).
    virtual bool flow_buffers_are_immutable() const { return false; }
};

// Same interface as the reference's ImageOpticalFlowNVOF (set_rgb / step_frame / flow / get_grid_size /
// get_scaling_factor / get_matrix_type, ImageOpticalFlowNVOF.cpp:100-200) over the HIP pyramidal Lucas-Kanade producer
// (roft_optical_flow): MI355X has no fixed-function flow unit.  The frame is handed over as the 8-bit gray image
// cv::cvtColor(frame, COLOR_BGR2GRAY) leaves (cpp:123).
class ImageOpticalFlowHIP : public ImageOpticalFlowSource {
public:
    enum class NVOFPerformance_1_0 { Slow, Medium, Fast };
    enum class NVOFPerformance_2_0 { Slow, Medium, Fast };
    enum class Product { NVOF_1_0 = 1, NVOF_2_0 = 2 };   // CV_16SC2 S10.5 at grid 4 | CV_32FC2 at grid 1 (cpp:19-80)
    ImageOpticalFlowHIP(int width, int height, Product product) : w_(width), h_(height), product_(product) { init(); }
    // the reference's constructors (ImageOpticalFlowNVOF.h:43-46): frames come from the camera measurement, which the FILTER
    // steps; the performance setting and the temporal hints of the NVIDIA engine have no counterpart here
    ImageOpticalFlowHIP(std::shared_ptr<ROFT::CameraMeasurement> camera_measurement, const NVOFPerformance_1_0&, const bool = false)
        : camera_(std::move(camera_measurement)), product_(Product::NVOF_1_0) { init_from_camera(); }
    ImageOpticalFlowHIP(std::shared_ptr<ROFT::CameraMeasurement> camera_measurement, const NVOFPerformance_2_0&, const bool = false)
        : camera_(std::move(camera_measurement)), product_(Product::NVOF_2_0) { init_from_camera(); }
    // the next camera frame (gray, width x height); step_frame() then computes the flow from the previous one to it
    void set_gray_image(const std::uint8_t* gray) { pending_.assign(gray, gray + (std::size_t)w_ * h_); }
    bool step_frame() override
    {
        if (camera_) {   // cv::cvtColor(frame, COLOR_BGR2GRAY) of the latched camera image (ImageOpticalFlowNVOF.cpp:112-123)
            bool valid = false;
            bfl::Data data;
            std::tie(valid, data) = camera_->measure();
            if (!valid) return false;
            const cv::Mat gray = compat::bgr_to_gray(std::get<1>(bfl::any::any_cast<CameraMeasurement::CameraMeasurementTuple>(data)));
            if (gray.empty() || gray.cols != w_ || gray.rows != h_) return false;
            return step_frame(gray.data);
        }
        if (pending_.empty()) return false;
        return step_frame(pending_.data());
    }
    // returns false on the first frame (no previous image yet), like the reference
    bool step_frame(const std::uint8_t* gray)
    {
        const std::size_t n = (std::size_t)w_ * h_;
        if (last_.empty()) {
            last_.assign(gray, gray + n);
            flow_in_ = false;
            return false;
        }
        // a buffer of its own for every frame (the pool recycles blocks by size): a consumer that still holds the last frame's
        // cv::Mat -- the filter keeps flows referenced for as long as a mask can be chased through them -- keeps its content
        flow_ = cv::Mat(h_ / (int)get_grid_size(), w_ / (int)get_grid_size(), get_matrix_type());
        compat::throw_if(roft_optical_flow(last_.data(), gray, w_, h_, &prm_, get_matrix_type() == CV_16SC2 ? ROFT_FLOW_S16C2 : ROFT_FLOW_F32C2,
                                           flow_.data), "ImageOpticalFlowHIP::step_frame");
        last_.assign(gray, gray + n);
        flow_in_ = true;
        return true;
    }
    bool reset() override { last_.clear(); flow_in_ = false; return true; }
    bool is_stepping_required() const override { return true; }
    std::tuple<bool, cv::Mat> flow(const bool& /*blocking*/) override { return std::make_tuple(flow_in_, flow_); }
    std::size_t get_grid_size() const override { return product_ == Product::NVOF_1_0 ? 4 : 1; }
    float get_scaling_factor() const override { return product_ == Product::NVOF_1_0 ? 32.0f : 1.0f; }
    int get_matrix_type() const override { return product_ == Product::NVOF_1_0 ? CV_16SC2 : CV_32FC2; }
    bool flow_buffers_are_immutable() const override { return true; }   // (step_frame allocates)
    roft_of_params& parameters() { return prm_; }

private:
    void init()
    {
        compat::throw_if(roft_default_of_params(&prm_), "ImageOpticalFlowHIP");
        if (w_ <= 0 || h_ <= 0) throw std::runtime_error("ImageOpticalFlowHIP: bad image size");
        flow_ = cv::Mat(h_ / (int)get_grid_size(), w_ / (int)get_grid_size(), get_matrix_type());
    }
    void init_from_camera()
    {
        if (!camera_) throw std::runtime_error("ImageOpticalFlowHIP::ctor. Error: null camera measurement.");
        bool valid = false;
        RobotsIO::Camera::CameraParameters p;
        std::tie(valid, p) = camera_->camera_parameters();
        if (!valid) throw std::runtime_error("ImageOpticalFlowHIP::ctor. Error: cannot get camera parameters.");
        w_ = (int)p.width(); h_ = (int)p.height();
        init();
    }
    std::shared_ptr<ROFT::CameraMeasurement> camera_;
    int w_ = 0, h_ = 0;
    Product product_;
    roft_of_params prm_{};
    std::vector<std::uint8_t> last_, pending_;
    cv::Mat flow_;
    bool flow_in_ = false;
};
using ImageOpticalFlowNVOF = ImageOpticalFlowHIP;   // drop-in name

// object model: where the mesh rendered by the outlier test comes from
class ModelParameters : public RobotsIO::Utils::Parameters {
public:
    const std::string& name() const { return name_; }
    void name(const std::string& v) { name_ = v; }
    bool use_internal_db() const { return use_internal_db_; }
    void use_internal_db(bool v) { use_internal_db_ = v; }
    const std::string& internal_db_name() const { return internal_db_name_; }
    void internal_db_name(const std::string& v) { internal_db_name_ = v; }
    const std::string& mesh_external_path() const { return mesh_external_path_; }
    void mesh_external_path(const std::string& v) { mesh_external_path_ = v; }
    const std::string& textured_mesh_external_path() const { return textured_mesh_external_path_; }
    void textured_mesh_external_path(const std::string& v) { textured_mesh_external_path_ = v; }
    const std::string& cloud_external_path() const { return cloud_external_path_; }
    void cloud_external_path(const std::string& v) { cloud_external_path_ = v; }

private:
    std::string name_, internal_db_name_, mesh_external_path_, textured_mesh_external_path_, cloud_external_path_;
    bool use_internal_db_ = false;
};

}  // namespace ROFT
