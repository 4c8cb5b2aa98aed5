// ROFT::ImageSegmentationOFAidedSource<T> -- a segmentation source that keeps a mask alive between the (slow, late)
// deliveries of a segmentation network by pushing it along the optical flow (reference:
// src/roft-lib/include/ROFT/ImageSegmentationOFAidedSource.hpp:34-92; step_frame :127-231; map :234-281).  T = cv::Vec2f
// or cv::Vec2s as for the flow measurement.  The delivery schedule (first mask = initialisation, empty masks ignored, flow
// frames buffered until the next mask) is host-side bookkeeping as in the reference; map() + cv::remap() -- the forward
// chase of every mask pixel through the buffered flows and the gather -- is one call of roft_mask_propagate.
#pragma once

#include "Sources.h"

namespace ROFT {

template <class T>
class ImageSegmentationOFAidedSource : public RobotsIO::Utils::Segmentation {
public:
    ImageSegmentationOFAidedSource(std::shared_ptr<RobotsIO::Utils::Segmentation> segmentation_source,
                                   std::shared_ptr<ROFT::ImageOpticalFlowSource> flow_source,
                                   const RobotsIO::Camera::CameraParameters& camera_parameters, const bool& wait_source_initialization)
        : segmentation_(std::move(segmentation_source)), flow_(std::move(flow_source)), width_((int)camera_parameters.width()),
          height_((int)camera_parameters.height()), flow_grid_size_(flow_->get_grid_size()), flow_scaling_factor_(flow_->get_scaling_factor()),
          segm_frames_between_iterations_(segmentation_->get_frames_between_iterations())
    {
        static_assert(sizeof(T) == 8 || sizeof(T) == 4, "T is cv::Vec2f or cv::Vec2s");
        (void)wait_source_initialization;   // (sources here never block: a missing first mask is simply "not available yet")
    }
    virtual ~ImageSegmentationOFAidedSource() = default;

    void set_rgb_image(const cv::Mat& image, const double& timestamp) override { segmentation_->set_rgb_image(image, timestamp); }

    bool step_frame() override
    {
        if (segmentation_->is_stepping_required()) segmentation_->step_frame();
        bool valid_segmentation = false;
        cv::Mat mask;
        std::tie(valid_segmentation, mask) = segmentation_->segmentation(false);
        if (!segmentation_available_ && valid_segmentation) {
            // the first mask is an initialisation, not a "new" mask (hpp:169-178)
            segmentation_available_ = true;
            mask_ = mask.clone();
            valid_segmentation = false;
        }
        if (valid_segmentation && count_non_zero(mask) == 0) {
            // uninformative mask: skipped; with an unknown delivery rate the buffered flows are dropped (hpp:186-198)
            valid_segmentation = false;
            if (segm_frames_between_iterations_ <= 0) flow_buffer_.clear();
        }
        bool valid_flow = false;
        cv::Mat flow;
        std::tie(valid_flow, flow) = flow_->flow(false);
        valid_flow = valid_flow && !is_first_frame_;
        if (valid_flow) flow_buffer_.push_back(flow.clone());
        if (valid_segmentation) {
            mask_ = mask.clone();
            propagate(flow_buffer_);          // through the buffered flows (the last frames_between of them, hpp:239-245)
            flow_buffer_.clear();
        } else if (valid_flow && segmentation_available_) {
            mask_.data[0] = 0;                // mask_.at<uchar>(0, 0) = 0 (hpp:224)
            propagate({flow});
        }
        is_first_frame_ = false;
        return true;
    }
    bool is_stepping_required() const override { return true; }
    bool reset() override
    {
        segmentation_available_ = false;
        is_first_frame_ = true;
        flow_buffer_.clear();
        return segmentation_->reset();
    }
    void reset_data_loading_time() override { segmentation_->reset_data_loading_time(); }
    double get_data_loading_time() const override { return segmentation_->get_data_loading_time(); }
    int get_frames_between_iterations() const override { return 1; }
    // the propagated mask of the current frame: "new" on every frame once a mask has been received
    std::pair<bool, cv::Mat> segmentation(const bool& = false) override { return std::make_pair(segmentation_available_, mask_); }
    std::pair<bool, cv::Mat> latest_segmentation() override { return std::make_pair(segmentation_available_, mask_); }

private:
    static std::size_t count_non_zero(const cv::Mat& m)
    {
        std::size_t n = 0;
        for (std::size_t i = 0; i < m.total(); ++i) n += m.data[i] != 0;
        return n;
    }
    void propagate(const std::vector<cv::Mat>& flows)
    {
        std::vector<roft_flow> fd(flows.size());
        for (std::size_t i = 0; i < flows.size(); ++i) {
            fd[i].data = flows[i].data;
            fd[i].type = flows[i].type() == CV_16SC2 ? ROFT_FLOW_S16C2 : ROFT_FLOW_F32C2;
            fd[i].cols = flows[i].cols;
            fd[i].rows = flows[i].rows;
            fd[i].grid = (int)flow_grid_size_;
            fd[i].scale = flow_scaling_factor_;
            fd[i].valid = 1;
        }
        compat::throw_if(roft_mask_propagate(mask_.data, width_, height_, fd.data(), (int)fd.size(), segm_frames_between_iterations_),
                         "ImageSegmentationOFAidedSource::step_frame");
    }
    std::shared_ptr<RobotsIO::Utils::Segmentation> segmentation_;
    std::shared_ptr<ROFT::ImageOpticalFlowSource> flow_;
    const int width_, height_;
    const std::size_t flow_grid_size_;
    const float flow_scaling_factor_;
    const int segm_frames_between_iterations_;
    bool segmentation_available_ = false, is_first_frame_ = true;
    cv::Mat mask_;
    std::vector<cv::Mat> flow_buffer_;
    const std::string log_name_ = "ImageSegmentationOFAidedSource";
};

}  // namespace ROFT
