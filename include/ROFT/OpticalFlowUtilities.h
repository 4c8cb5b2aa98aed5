// ROFT::OpticalFlowUtils -- src/roft-lib/include/ROFT/OpticalFlowUtilities.h:19-31, src/OpticalFlowUtilities.cpp:26-110:
// the validity rule of a flow vector and the `.float` frame format (int OpenCV type -- CV_32FC2 or CV_16SC2 --, two size_t
// width and height, then width x height x 2 elements).
#pragma once

#include <cstdio>
#include <iostream>

#include "Compat.h"

namespace ROFT {
namespace OpticalFlowUtils {

// finite and not one of the huge values optical-flow estimators mark unknown pixels with (OpticalFlowUtilities.h:19-22)
inline bool is_flow_valid(const float& f_x, const float& f_y)
{
    return !std::isnan(f_x) && !std::isnan(f_y) && std::fabs(f_x) < 1e9 && std::fabs(f_y) < 1e9;
}

inline std::pair<bool, cv::Mat> read_flow(const std::string& file_name)
{
    const std::string log_name = "ROFT::OpticalFlowUtils::read_flow";
    std::FILE* in = std::fopen(file_name.c_str(), "rb");
    if (!in) {
        std::cout << log_name << " Error: cannot load flow frame " + file_name << std::endl;
        return std::make_pair(false, cv::Mat());
    }
    int frame_type = 0;
    std::size_t frame_size[2] = {0, 0};
    bool ok = std::fread(&frame_type, sizeof(frame_type), 1, in) == 1 && std::fread(frame_size, sizeof(frame_size), 1, in) == 1 &&
              (frame_type == CV_32FC2 || frame_type == CV_16SC2) && frame_size[0] > 0 && frame_size[1] > 0 && frame_size[0] < 65536 && frame_size[1] < 65536;
    cv::Mat flow;
    if (ok) {
        flow = cv::Mat((int)frame_size[1], (int)frame_size[0], frame_type);
        const std::size_t n = 2 * frame_size[0] * frame_size[1];
        ok = std::fread(flow.data, flow.elemSize() / 2, n, in) == n;
    }
    std::fclose(in);
    if (!ok) {
        std::cout << log_name << " Error: cannot load flow data for frame" + file_name << std::endl;
        return std::make_pair(false, cv::Mat());
    }
    return std::make_pair(true, flow);
}

inline bool save_flow(const cv::Mat& flow, const std::string& output_path)
{
    std::FILE* out = std::fopen(output_path.c_str(), "wb");
    if (!out) return false;
    const int type = flow.type();
    const std::size_t size[2] = {(std::size_t)flow.cols, (std::size_t)flow.rows};
    const std::size_t n = 2 * size[0] * size[1];
    const bool ok = std::fwrite(&type, sizeof(type), 1, out) == 1 && std::fwrite(size, sizeof(size), 1, out) == 1 &&
                    std::fwrite(flow.data, flow.elemSize() / 2, n, out) == n;
    std::fclose(out);
    return ok;
}

}  // namespace OpticalFlowUtils
}  // namespace ROFT
