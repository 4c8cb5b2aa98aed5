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
          fps_(fps), simulated_fps_(simulated_fps), simulate_inference_time_(simulate_inference_time), head_0_(head_ + 1),
          delay_(static_cast<int>(fps_ / simulated_fps_))
    {
        if (delay_ < 1) throw std::runtime_error(log_name_ + "Delayed::ctor. Error: the simulated rate exceeds the rate of the data.");
    }
    // (only the frames the schedule delivers are read from disk)
    bool step_frame() override { head_++; return true; }
    std::pair<bool, cv::Mat> segmentation(const bool&) override
    {
        int index = head_;
        if (simulate_inference_time_) index -= delay_;
        if (((index - head_0_) % delay_) != 0) return std::make_pair(false, cv::Mat());
        if (index < 0) index = head_0_;
        const auto t0 = std::chrono::steady_clock::now();
        auto output = read_file((std::size_t)index);
        data_loading_time_ = (double)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        return output;
    }
    void reset_data_loading_time() override { data_loading_time_ = 0.0; }
    double get_data_loading_time() const override { return data_loading_time_; }
    int get_frames_between_iterations() const override { return int(fps_ / simulated_fps_); }

private:
    const float fps_, simulated_fps_;
    const bool simulate_inference_time_;
    const int head_0_, delay_;
    double data_loading_time_ = 0.0;
};

}  // namespace ROFT
