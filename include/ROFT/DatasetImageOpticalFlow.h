// ROFT::DatasetImageOpticalFlow -- optical-flow frames stored next to a sequence, <dataset>/optical_flow/<set>/<index>.float
// (src/roft-lib/include/ROFT/DatasetImageOpticalFlow.h:24-77, src/DatasetImageOpticalFlow.cpp:24-140).  The grid of a set is
// the image width over the width of its frames, CV_16SC2 frames hold pixels x 32 (cpp:46-50); the first frame of a set
// may be missing (there is no flow INTO frame 0): the probe of the constructor takes the first file that reads.
#pragma once

#include "CompatIO.h"
#include "OpticalFlowUtilities.h"
#include "Sources.h"

namespace ROFT {

class DatasetImageOpticalFlow : public ImageOpticalFlowSource {
public:
    DatasetImageOpticalFlow(const std::string& dataset_path, const std::string& set, const std::size_t width, const std::size_t height,
                            const std::size_t& heading_zeros = 0, const std::size_t& index_offset = 0)
        : files_(with_slash(dataset_path) + "optical_flow/" + set, "", heading_zeros, ".float", index_offset), image_width_(width), image_height_(height)
    {
        // (the reference probes without a bound; a set without a single frame among its first 64 is an error here)
        std::pair<bool, cv::Mat> probe{false, cv::Mat()};
        for (long i = 0; i < 64 && !probe.first; ++i) probe = OpticalFlowUtils::read_flow(files_.path(i));
        if (!probe.first) throw std::runtime_error(name_ + "::ctor. Error: no optical flow frames in " + files_.directory());
        type_ = probe.second.type();
        grid_ = image_width_ / (std::size_t)probe.second.cols;
        scale_ = type_ == CV_16SC2 ? 32.0f : 1.0f;
        std::cout << name_ << "::ctor." << std::endl
                  << name_ << "   - grid size: " << grid_ << std::endl
                  << name_ << "   - scaling factor: " << scale_ << std::endl
                  << name_ << "   - matrix type: " << (type_ == CV_32FC2 ? "CV_32FC2" : "CV_16SC2") << std::endl;
    }
    bool reset() override { files_.rewind(); return true; }
    bool step_frame() override
    {
        files_.advance();
        loading_ms_ = compat::milliseconds_of([&] { frame_ = OpticalFlowUtils::read_flow(files_.current()); });
        return true;
    }
    bool is_stepping_required() const override { return true; }
    std::size_t get_grid_size() const override { return grid_; }
    float get_scaling_factor() const override { return scale_; }
    double get_data_loading_time() const override { return loading_ms_; }
    int get_matrix_type() const override { return type_; }
    bool flow_buffers_are_immutable() const override { return true; }   // (read_flow returns a new matrix per file)
    std::tuple<bool, cv::Mat> flow(const bool&) override { return std::make_tuple(frame_.first, frame_.second); }

private:
    static std::string with_slash(std::string p) { if (!p.empty() && p.back() != '/') p += '/'; return p; }
    compat::IndexedFiles files_;
    const std::size_t image_width_, image_height_;
    std::size_t grid_ = 1;
    float scale_ = 1.0f;
    int type_ = CV_32FC2;
    std::pair<bool, cv::Mat> frame_{false, cv::Mat()};
    double loading_ms_ = 0.0;
    const std::string name_ = "DatasetImageOpticalFlow";
};

}  // namespace ROFT
