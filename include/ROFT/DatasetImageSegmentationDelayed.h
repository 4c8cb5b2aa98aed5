// ROFT::DatasetImageSegmentationDelayed -- the stored masks as a slow segmentation network would deliver them: every
// (fps / simulated_fps)-th frame only and, with `simulate_inference_time`, one such period late -- the mask of frame k
// arrives with frame k + delay, the first one also at the very first frame
// (src/roft-lib/include/ROFT/DatasetImageSegmentationDelayed.h:21-53, src/DatasetImageSegmentationDelayed.cpp:15-81).
#pragma once

#include "DatasetImageSegmentation.h"

namespace ROFT {

class DatasetImageSegmentationDelayed : public DatasetImageSegmentation {
public:
    DatasetImageSegmentationDelayed(const float& fps, const float& simulated_fps, const bool simulate_inference_time, const std::string& dataset_path,
                                    const std::string& format, const std::size_t width, const std::size_t height, const std::string& segmentation_set,
                                    const ModelParameters& model_parameters, const std::size_t heading_zeros = 0, const std::size_t index_offset = 0)
        : DatasetImageSegmentation(dataset_path, format, width, height, segmentation_set, model_parameters, heading_zeros, index_offset),
          first_(head_ + 1), period_(static_cast<int>(fps / simulated_fps)), late_(simulate_inference_time)
    {
        if (period_ < 1) throw std::runtime_error(log_name_ + "Delayed::ctor. Error: the simulated rate exceeds the rate of the data.");
    }
    // (only the frames the schedule delivers are read from disk)
    bool step_frame() override { ++head_; return true; }
    std::pair<bool, cv::Mat> segmentation(const bool&) override
    {
        const long item = compat::delayed_item(head_, first_, period_, late_);
        if (item < 0) return {false, cv::Mat()};
        std::pair<bool, cv::Mat> delivered;
        loading_ms_ = compat::milliseconds_of([&] { delivered = read_file((std::size_t)item); });
        return delivered;
    }
    void reset_data_loading_time() override { loading_ms_ = 0.0; }
    double get_data_loading_time() const override { return loading_ms_; }
    int get_frames_between_iterations() const override { return period_; }

private:
    const int first_, period_;
    const bool late_;
    double loading_ms_ = 0.0;
};

}  // namespace ROFT
