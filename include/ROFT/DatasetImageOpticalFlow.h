// ROFT::DatasetImageOpticalFlow -- optical-flow frames stored next to a sequence, <dataset>/optical_flow/<set>/<index>.float
// (src/roft-lib/include/ROFT/DatasetImageOpticalFlow.h:24-77, src/DatasetImageOpticalFlow.cpp:24-140).  The grid of a set is
// the image width over the width of its frames, CV_16SC2 frames hold pixels x 32 (cpp:46-50); the first frame of a set
// may be missing (there is no flow INTO frame 0): the probe of the constructor takes the first file that reads.
#pragma once

#include <chrono>

#include "CompatIO.h"
#include "OpticalFlowUtilities.h"
#include "Sources.h"

namespace ROFT {

class DatasetImageOpticalFlow : public ImageOpticalFlowSource {
public:
    DatasetImageOpticalFlow(const std::string& dataset_path, const std::string& set, const std::size_t width, const std::size_t height,
                            const std::size_t& heading_zeros = 0, const std::size_t& index_offset = 0)
        : width_(width), height_(height), head_(-1 + (int)index_offset), index_offset_(index_offset), heading_zeros_(heading_zeros)
    {
        std::string root = dataset_path;
        if (!root.empty() && root.back() != '/') root += '/';
        dataset_path_ = root + "optical_flow/" + set + "/";
        bool valid = false;
        cv::Mat tmp;
        // (the reference probes without a bound; a set without a single frame is an error here)
        for (int counter = 0; !valid && counter < 64; ++counter)
            std::tie(valid, tmp) = OpticalFlowUtils::read_flow(dataset_path_ + compat::padded_index(counter, heading_zeros_) + ".float");
        if (!valid) throw std::runtime_error(log_name_ + "::ctor. Error: no optical flow frames in " + dataset_path_);
        grid_size_ = width_ / (std::size_t)tmp.cols;
        matrix_type_ = tmp.type();
        scaling_factor_ = matrix_type_ == CV_16SC2 ? float(1 << 5) : 1.0f;
        std::cout << log_name_ + "::ctor." << std::endl;
        std::cout << log_name_ + "   - grid size: " << grid_size_ << std::endl;
        std::cout << log_name_ + "   - scaling factor: " << scaling_factor_ << std::endl;
        std::cout << log_name_ + "   - matrix type: " << (matrix_type_ == CV_32FC2 ? "CV_32FC2" : "CV_16SC2") << std::endl;
    }
    bool reset() override { head_ = -1 + (int)index_offset_; return true; }
    bool step_frame() override
    {
        head_++;
        const auto t0 = std::chrono::steady_clock::now();
        std::tie(valid_, output_) = OpticalFlowUtils::read_flow(dataset_path_ + compat::padded_index(head_, heading_zeros_) + ".float");
        data_loading_time_ = (double)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        return true;
    }
    bool is_stepping_required() const override { return true; }
    std::size_t get_grid_size() const override { return grid_size_; }
    float get_scaling_factor() const override { return scaling_factor_; }
    double get_data_loading_time() const override { return data_loading_time_; }
    int get_matrix_type() const override { return matrix_type_; }
    std::tuple<bool, cv::Mat> flow(const bool&) override { return std::make_tuple(valid_, output_); }

private:
    std::string dataset_path_;
    const std::size_t width_, height_;
    int head_;
    const std::size_t index_offset_, heading_zeros_;
    std::size_t grid_size_ = 1;
    float scaling_factor_ = 1.0f;
    int matrix_type_ = CV_32FC2;
    bool valid_ = false;
    cv::Mat output_;
    double data_loading_time_ = 0.0;
    const std::string log_name_ = "DatasetImageOpticalFlow";
};

}  // namespace ROFT
