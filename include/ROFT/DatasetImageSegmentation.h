// ROFT::DatasetImageSegmentation -- the masks stored next to a sequence, <dataset>/masks/<set>/<object>_<index>.<format>,
// one per frame (src/roft-lib/include/ROFT/DatasetImageSegmentation.h:25-77, src/DatasetImageSegmentation.cpp:20-165).
// A colour mask is delivered as the gray image ImageSegmentationMeasurement.cpp:62-63 turns it into before the filter's
// `> 1 -> 255` threshold (the engine ingests one byte per pixel).
#pragma once

#include "CompatIO.h"
#include "Sources.h"

namespace ROFT {

class DatasetImageSegmentation : public RobotsIO::Utils::Segmentation {
public:
    DatasetImageSegmentation(const std::string& dataset_path, const std::string& format, const std::size_t width, const std::size_t height,
                             const std::string& segmentation_set, const ModelParameters& model_parameters, const std::size_t heading_zeros = 0,
                             const std::size_t index_offset = 0, const bool simulate_missing_detections = false)
        : files_(with_slash(dataset_path) + "masks/" + segmentation_set, model_parameters.name() + "_", heading_zeros, "." + format, index_offset),
          head_(-1 + (int)index_offset), mask_width_(width), mask_height_(height), empty_when_missing_(simulate_missing_detections)
    {}
    bool reset() override { files_.rewind(); head_ = (int)files_.cursor(); return true; }
    bool step_frame() override
    {
        head_ = (int)files_.advance();
        loading_ms_ = compat::milliseconds_of([&] { latest_ = read_file((std::size_t)head_); });
        return true;
    }
    bool is_stepping_required() const override { return true; }
    void reset_data_loading_time() override { loading_ms_ = 0.0; }
    double get_data_loading_time() const override { return loading_ms_; }
    std::pair<bool, cv::Mat> segmentation(const bool&) override { return latest_; }

protected:
    // the mask of frame `frame_index` as one byte per pixel; (false, empty) when the file is missing or of another size
    std::pair<bool, cv::Mat> read_file(const std::size_t& frame_index)
    {
        const std::string file_name = files_.path((long)frame_index);
        cv::Mat mask = compat::bgr_to_gray(compat::read_png(file_name));
        if (mask.empty() && empty_when_missing_) mask = cv::Mat((int)mask_height_, (int)mask_width_, CV_8UC1);
        if (mask.empty()) {
            std::cout << log_name_ << "::segmentation. Error: cannot load segmentation data for frame" + file_name << std::endl;
            return {false, cv::Mat()};
        }
        if ((std::size_t)mask.cols != mask_width_ || (std::size_t)mask.rows != mask_height_) {
            std::cout << log_name_ << "::segmentation. Error: unexpected size of the mask " + file_name << std::endl;
            return {false, cv::Mat()};
        }
        return {true, mask};
    }
    static std::string with_slash(std::string p) { if (!p.empty() && p.back() != '/') p += '/'; return p; }
    compat::IndexedFiles files_;
    int head_;                               // index of the current frame (file indices; -1 + index_offset before the first step)
    const std::size_t mask_width_, mask_height_;
    const bool empty_when_missing_;
    std::pair<bool, cv::Mat> latest_{false, cv::Mat()};
    double loading_ms_ = 0.0;
    const std::string log_name_ = "DatasetImageSegmentation";
};

}  // namespace ROFT
