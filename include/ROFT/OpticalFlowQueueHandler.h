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
    explicit OpticalFlowQueueHandler(const std::size_t& window_size) : capacity_(window_size) {}
    void add_flow(const cv::Mat& frame, const double& time_stamp)
    {
        if (capacity_ == 0) return;
        if (queue_.size() == capacity_) queue_.pop_front();
        queue_.push_back(Entry{frame.clone(), time_stamp});
    }
    std::vector<cv::Mat> get_buffer_region(const double& initial_time_stamp)
    {
        // the entry stamped `initial_time_stamp` (to a millisecond) is the flow INTO that image: what follows it is the region
        auto at = std::find_if(queue_.begin(), queue_.end(), [&](const Entry& e) { return std::fabs(e.timestamp - initial_time_stamp) < 1e-3; });
        std::vector<cv::Mat> region;
        if (at != queue_.end())
            for (++at; at != queue_.end(); ++at) region.push_back(at->frame.clone());
        return region;
    }
    void clear() { queue_.clear(); }

private:
    std::size_t capacity_;
    std::deque<Entry> queue_;
};

}  // namespace ROFT
