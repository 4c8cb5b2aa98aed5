// ROFT::DatasetImageSegmentation -- the masks stored next to a sequence, <dataset>/masks/<set>/<object>_<index>.<format>,
// one per frame (src/roft-lib/include/ROFT/DatasetImageSegmentation.h:25-77, src/DatasetImageSegmentation.cpp:20-165).
// A colour mask is delivered as the gray image ImageSegmentationMeasurement.cpp:62-63 turns it into before the filter's
// `> 1 -> 255` threshold (the engine ingests one byte per pixel).
#pragma once

#include <chrono>

#include "CompatIO.h"
#include "Sources.h"

namespace ROFT {

class DatasetImageSegmentation : public RobotsIO::Utils::Segmentation {
public:
    DatasetImageSegmentation(const std::string& dataset_path, const std::string& format, const std::size_t width, const std::size_t height,
                             const std::string& segmentation_set, const ModelParameters& model_parameters, const std::size_t heading_zeros = 0,
                             const std::size_t index_offset = 0, const bool simulate_missing_detections = false)
        : format_(format), width_(width), height_(height), object_name_(model_parameters.name()), head_(-1 + (int)index_offset),
          index_offset_(index_offset), heading_zeros_(heading_zeros), simulate_missing_detections_(simulate_missing_detections)
    {
        std::string root = dataset_path;
        if (!root.empty() && root.back() != '/') root += '/';
        dataset_path_ = root + "masks/" + segmentation_set + "/";
    }
    bool reset() override { head_ = -1 + (int)index_offset_; return true; }
    bool step_frame() override
    {
        head_++;
        const auto t0 = std::chrono::steady_clock::now();
        output_ = read_file(head_);
        data_loading_time_ = (double)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        return true;
    }
    bool is_stepping_required() const override { return true; }
    void reset_data_loading_time() override { data_loading_time_ = 0.0; }
    double get_data_loading_time() const override { return data_loading_time_; }
    std::pair<bool, cv::Mat> segmentation(const bool&) override { return output_; }

protected:
    std::pair<bool, cv::Mat> read_file(const std::size_t& frame_index)
    {
        const std::string file_name = dataset_path_ + object_name_ + "_" + compat::padded_index((long)frame_index, heading_zeros_) + "." + format_;
        cv::Mat segmentation = compat::bgr_to_gray(compat::read_png(file_name));
        if (segmentation.empty()) {
            if (simulate_missing_detections_) segmentation = cv::Mat((int)height_, (int)width_, CV_8UC1);
            else {
                std::cout << log_name_ << "::segmentation. Error: cannot load segmentation data for frame" + file_name << std::endl;
                return std::make_pair(false, cv::Mat());
            }
        }
        if ((std::size_t)segmentation.cols != width_ || (std::size_t)segmentation.rows != height_) {
            std::cout << log_name_ << "::segmentation. Error: unexpected size of the mask " + file_name << std::endl;
            return std::make_pair(false, cv::Mat());
        }
        return std::make_pair(true, segmentation);
    }
    std::string dataset_path_;
    const std::string format_;
    const std::size_t width_, height_;
    std::string object_name_;
    int head_;
    const std::size_t index_offset_, heading_zeros_;
    const bool simulate_missing_detections_;
    std::pair<bool, cv::Mat> output_;
    double data_loading_time_ = 0.0;
    const std::string log_name_ = "DatasetImageSegmentation";
};

}  // namespace ROFT
