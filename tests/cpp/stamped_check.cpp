// ROFT::ImageSegmentationOFAidedSourceStamped<cv::Vec2f> over a recorded stream with an irregular, time-stamped mask delivery:
// dumps the mask the source holds after every frame.
//   stamped_check <stream.bin> <stamps.txt: per frame `frame_stamp mask_stamp`> <frames_between> <out.bin>
#include <cstdio>

#include "ROFT/ImageSegmentationOFAidedSourceStamped.hpp"
#include "mem_sources.h"

class StampedMemSegmentation : public MemSegmentation {
public:
    StampedMemSegmentation(const RecordedStream& s, int fb, const std::vector<double>& mask_stamps) : MemSegmentation(s, fb), stamps_(mask_stamps) {}
    bool step_frame() override { ++k_; return MemSegmentation::step_frame(); }
    double get_time_stamp() override { return k_ >= 0 && k_ < (int)stamps_.size() ? stamps_[k_] : -1.0; }

private:
    const std::vector<double>& stamps_;
    int k_ = -1;
};

int main(int argc, char** argv)
{
    if (argc < 5) return 2;
    try {
        RecordedStream s;
        if (!s.load(argv[1])) return 2;
        std::vector<double> frame_stamp(s.n), mask_stamp(s.n);
        FILE* f = std::fopen(argv[2], "r");
        for (int k = 0; k < s.n; ++k)
            if (std::fscanf(f, "%lf %lf", &frame_stamp[k], &mask_stamp[k]) != 2) return 2;
        std::fclose(f);
        auto seg = std::make_shared<StampedMemSegmentation>(s, std::atoi(argv[3]), mask_stamp);
        auto flow = std::make_shared<MemFlow>(s);
        ROFT::ImageSegmentationOFAidedSourceStamped<cv::Vec2f> src(seg, flow, s.parameters(), false);
        FILE* out = std::fopen(argv[4], "wb");
        for (int k = 0; k < s.n; ++k) {
            flow->step_frame();                              // (the filter steps camera and flow, the source only the masks)
            src.set_rgb_image(cv::Mat(), frame_stamp[k]);
            if (!src.step_frame()) return 4;
            const auto m = src.segmentation(false);
            const unsigned char have = m.first;
            std::fwrite(&have, 1, 1, out);
            if (m.first) std::fwrite(m.second.data, 1, (std::size_t)s.W * s.H, out);
            if (src.get_time_stamp() != frame_stamp[k]) return 5;
        }
        std::fclose(out);
    } catch (const std::exception& e) {
        std::printf("runtime_error %s\n", e.what());
        return 3;
    }
    return 0;
}
