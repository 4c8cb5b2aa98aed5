// ROFT::ImageSegmentationOFAidedSourceStamped<T> -- the flow-aided segmentation source for LIVE mask sources: a mask carries
// the time stamp of the image it was computed on and is carried through the optical-flow frames stored since that image (a
// queue of the last 30, keyed by stamp) instead of "the last n" (reference:
// src/roft-lib/include/ROFT/ImageSegmentationOFAidedSourceStamped.hpp:35-105, step_frame :153-268, map :271-317).  The camera
// image and its stamp arrive through set_rgb_image(); the filter steps camera and flow itself.  map() + cv::remap() are one
// call of roft_mask_propagate, as in ImageSegmentationOFAidedSource.hpp; inside the batched engine the same source is
// roft_config::stamped_masks (roft_frame_input::stamp / mask_stamp).
#pragma once

#include <chrono>
#include <thread>

#include "OpticalFlowQueueHandler.h"
#include "Sources.h"

namespace ROFT {

template <class T>
class ImageSegmentationOFAidedSourceStamped : public RobotsIO::Utils::Segmentation {
public:
    ImageSegmentationOFAidedSourceStamped(std::shared_ptr<RobotsIO::Utils::Segmentation> segmentation_source,
                                          std::shared_ptr<ROFT::ImageOpticalFlowSource> flow_source,
                                          const RobotsIO::Camera::CameraParameters& camera_parameters, const bool& wait_source_initialization,
                                          const std::size_t& source_feed_rate = -1)
        : segmentation_(std::move(segmentation_source)), flow_(std::move(flow_source)), wait_source_initialization_default_(wait_source_initialization),
          wait_source_initialization_(wait_source_initialization), segm_frames_between_iterations_(segmentation_->get_frames_between_iterations()),
          source_feed_rate_(source_feed_rate), flow_grid_size_(flow_->get_grid_size()), flow_scaling_factor_(flow_->get_scaling_factor()),
          width_((int)camera_parameters.width()), height_((int)camera_parameters.height()), flow_handler_(flow_queue_max_size_)
    {
        static_assert(sizeof(T) == 8 || sizeof(T) == 4, "T is cv::Vec2f or cv::Vec2s");
    }
    virtual ~ImageSegmentationOFAidedSourceStamped() = default;

    void set_rgb_image(const cv::Mat& image, const double& timestamp) override
    {
        rgb_image_ = image.clone();
        rgb_image_time_stamp_ = timestamp;
    }

    bool step_frame() override
    {
        // (no stepping of camera and flow on purpose: only the underlying segmentation source is stepped, hpp:156-159)
        if (segmentation_->is_stepping_required()) segmentation_->step_frame();
        bool valid_segmentation = false;
        cv::Mat mask;
        std::tie(valid_segmentation, mask) = segmentation_->segmentation(false);
        double mask_time_stamp = segmentation_->get_time_stamp();

        if (!segmentation_available_ && wait_source_initialization_) {
            // a live source may need the image first and a few hundred milliseconds (hpp:167-193)
            for (std::size_t i = 0; i < 5 && !valid_segmentation; ++i) {
                if (!rgb_image_.empty()) segmentation_->set_rgb_image(rgb_image_, rgb_image_time_stamp_);
                if (segmentation_->is_stepping_required()) segmentation_->step_frame();
                std::tie(valid_segmentation, mask) = segmentation_->segmentation(false);
                mask_time_stamp = segmentation_->get_time_stamp();
                if (!valid_segmentation) std::this_thread::sleep_for(std::chrono::milliseconds(200));
            }
            if (!valid_segmentation || count_non_zero(mask) == 0) return false;
        }
        if (source_feed_rate_ > 0) {   // every source_feed_rate-th frame the source is handed the current image (hpp:196-208)
            if (feed_rate_counter_ == source_feed_rate_) {
                if (!rgb_image_.empty()) {
                    feed_rate_counter_ = 0;
                    segmentation_->set_rgb_image(rgb_image_, rgb_image_time_stamp_);
                }
            } else feed_rate_counter_++;
        }
        if (!segmentation_available_ && valid_segmentation) {   // the first mask is an initialisation (hpp:210-219)
            segmentation_available_ = true;
            mask_ = mask.clone();
            valid_segmentation = false;
        }
        if (valid_segmentation && count_non_zero(mask) == 0) valid_segmentation = false;   // uninformative: skipped (hpp:221-228)

        bool valid_flow = false;
        cv::Mat flow;
        std::tie(valid_flow, flow) = flow_->flow(false);
        valid_flow = valid_flow && !is_first_frame_;
        if (valid_flow) flow_handler_.add_flow(flow, rgb_image_time_stamp_);

        if (valid_segmentation) {
            mask_ = mask.clone();
            const std::vector<cv::Mat> buffer = flow_handler_.get_buffer_region(mask_time_stamp);
            if (!buffer.empty()) propagate(buffer);
            else if (!flow.empty()) {          // stamp not in the queue: through the current flow only (hpp:249-254)
                mask_.data[0] = 0;
                propagate({flow});
            }
        } else if (valid_flow && segmentation_available_) {
            mask_.data[0] = 0;
            propagate({flow});
        }
        is_first_frame_ = false;
        return true;
    }
    bool is_stepping_required() const override { return true; }
    bool reset() override
    {
        wait_source_initialization_ = wait_source_initialization_default_;
        segmentation_available_ = false;
        is_first_frame_ = true;
        feed_rate_counter_ = 0;
        flow_handler_.clear();
        return segmentation_->reset();
    }
    void reset_data_loading_time() override { segmentation_->reset_data_loading_time(); }
    double get_data_loading_time() const override { return segmentation_->get_data_loading_time(); }
    std::pair<bool, cv::Mat> segmentation(const bool& = false) override { return std::make_pair(segmentation_available_, mask_); }
    double get_time_stamp() override { return rgb_image_time_stamp_; }

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
                         "ImageSegmentationOFAidedSourceStamped::step_frame");
    }
    std::shared_ptr<RobotsIO::Utils::Segmentation> segmentation_;
    std::shared_ptr<ROFT::ImageOpticalFlowSource> flow_;
    const bool wait_source_initialization_default_;
    bool wait_source_initialization_;
    bool segmentation_available_ = false, is_first_frame_ = true;
    const int segm_frames_between_iterations_;
    const std::size_t source_feed_rate_;
    std::size_t feed_rate_counter_ = 0;
    const std::size_t flow_grid_size_;
    const float flow_scaling_factor_;
    const int width_, height_;
    cv::Mat rgb_image_, mask_;
    double rgb_image_time_stamp_ = 0.0;
    const std::size_t flow_queue_max_size_ = 30;
    OpticalFlowQueueHandler flow_handler_;
    const std::string log_name_ = "ImageSegmentationOFAidedSourceStamped";
};

}  // namespace ROFT
