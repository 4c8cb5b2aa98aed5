// ROFT::OpticalFlowQueueHandler -- the last `window_size` optical-flow frames with the time stamps of the images they lead to
// (src/roft-lib/include/ROFT/OpticalFlowQueueHandler.h:21-50, src/OpticalFlowQueueHandler.cpp:18-64); get_buffer_region(t)
// returns the flows stored AFTER the entry stamped t (within a millisecond): the ones a mask computed on that image has to be
// carried through.  Host-side bookkeeping of the time-stamped mask source.
#pragma once

#include <deque>

#include "Compat.h"

namespace ROFT {

class OpticalFlowQueueHandler {
public:
    struct Entry {
        cv::Mat frame;
        double timestamp;
    };
    explicit OpticalFlowQueueHandler(const std::size_t& window_size) : window_size_(window_size) {}
    void add_flow(const cv::Mat& frame, const double& time_stamp)
    {
        buffer_.push_back(Entry{frame.clone(), time_stamp});
        if (buffer_.size() > window_size_) buffer_.pop_front();
    }
    std::vector<cv::Mat> get_buffer_region(const double& initial_time_stamp)
    {
        std::vector<cv::Mat> region;
        std::size_t index = 0;
        bool found = false;
        for (; index < buffer_.size(); ++index)
            if (std::fabs(buffer_[index].timestamp - initial_time_stamp) < 1e-3) { found = true; break; }
        if (!found) return region;
        // (a flow refers to the image before it: the region starts with the next entry)
        for (++index; index < buffer_.size(); ++index) region.push_back(buffer_[index].frame.clone());
        return region;
    }
    void clear() { buffer_.clear(); }

private:
    std::size_t window_size_;
    std::deque<Entry> buffer_;
};

}  // namespace ROFT
