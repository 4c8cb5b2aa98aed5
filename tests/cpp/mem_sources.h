// In-memory sources for the C++ facade tests: a recorded stream (written by tests/test_facade.py::dump_stream) served
// through the source interfaces the reference's classes poll (RobotsIO::Camera::Camera, RobotsIO::Utils::Segmentation,
// ROFT::ImageOpticalFlowSource, RobotsIO::Utils::Transform).
#pragma once

#include <cstdint>
#include <cstdio>
#include <memory>
#include <vector>

#include "ROFT/Filters.h"

struct RecordedStream {
    int W = 0, H = 0, n = 0;
    double cam[4] = {0, 0, 0, 0}, init[13] = {0};
    std::vector<float> verts;
    std::vector<std::int32_t> tris;
    struct Frame {
        double dt = 0;
        bool has_flow = false, has_mask = false, has_pose = false;
        std::vector<float> depth, flow;
        std::vector<std::uint8_t> mask;
        double pose[7] = {0};
    };
    std::vector<Frame> frames;

    template <class T>
    static bool rd(FILE* f, T* p, std::size_t n) { return std::fread(p, sizeof(T), n, f) == n; }

    bool load(const char* path)
    {
        FILE* f = std::fopen(path, "rb");
        if (!f) return false;
        std::int32_t hdr[5];   // W, H, n_frames, n_verts, n_tris
        if (!rd(f, hdr, 5) || !rd(f, cam, 4) || !rd(f, init, 13)) return false;
        W = hdr[0]; H = hdr[1]; n = hdr[2];
        verts.resize(3 * (std::size_t)hdr[3]);
        tris.resize(3 * (std::size_t)hdr[4]);
        if (!rd(f, verts.data(), verts.size()) || !rd(f, tris.data(), tris.size())) return false;
        frames.resize(n);
        for (Frame& fr : frames) {
            std::int32_t flags[3];
            fr.depth.resize((std::size_t)W * H);
            if (!rd(f, &fr.dt, 1) || !rd(f, flags, 3) || !rd(f, fr.depth.data(), fr.depth.size())) return false;
            fr.has_flow = flags[0]; fr.has_mask = flags[1]; fr.has_pose = flags[2];
            if (fr.has_flow) { fr.flow.resize(2 * (std::size_t)W * H); if (!rd(f, fr.flow.data(), fr.flow.size())) return false; }
            if (fr.has_mask) { fr.mask.resize((std::size_t)W * H); if (!rd(f, fr.mask.data(), fr.mask.size())) return false; }
            if (fr.has_pose && !rd(f, fr.pose, 7)) return false;
        }
        std::fclose(f);
        return true;
    }
    RobotsIO::Camera::CameraParameters parameters() const
    {
        RobotsIO::Camera::CameraParameters p;
        p.width(W); p.height(H); p.fx(cam[0]); p.fy(cam[1]); p.cx(cam[2]); p.cy(cam[3]);
        return p;
    }
};

// every source keeps its own cursor: step_frame() moves to the next frame of the recording
class MemCamera : public RobotsIO::Camera::Camera {
public:
    explicit MemCamera(const RecordedStream& s) : s_(s) {}
    bool step_frame() override { return ++k_ < s_.n; }
    bool reset() override { k_ = -1; return true; }
    std::pair<bool, RobotsIO::Camera::CameraParameters> parameters() const override { return {true, s_.parameters()}; }
    std::pair<bool, Eigen::MatrixXf> depth(const bool&) override
    {
        Eigen::MatrixXf d(s_.H, s_.W);
        std::memcpy(d.data(), s_.frames[k_].depth.data(), sizeof(float) * d.size());
        return {true, d};
    }
    std::int32_t frame_index() const override { return k_; }

private:
    const RecordedStream& s_;
    int k_ = -1;
};

class MemSegmentation : public RobotsIO::Utils::Segmentation {
public:
    MemSegmentation(const RecordedStream& s, int frames_between) : s_(s), fb_(frames_between) {}
    bool step_frame() override { ++k_; return true; }
    bool reset() override { k_ = -1; return true; }
    bool is_stepping_required() const override { return true; }
    int get_frames_between_iterations() const override { return fb_; }
    std::pair<bool, cv::Mat> segmentation(const bool&) override
    {
        if (k_ < 0 || k_ >= s_.n || !s_.frames[k_].has_mask) return {false, cv::Mat()};
        return {true, cv::Mat(s_.H, s_.W, CV_8UC1, const_cast<std::uint8_t*>(s_.frames[k_].mask.data()))};
    }

private:
    const RecordedStream& s_;
    int fb_, k_ = -1;
};

class MemFlow : public ROFT::ImageOpticalFlowSource {
public:
    explicit MemFlow(const RecordedStream& s) : s_(s) {}
    bool step_frame() override { ++k_; return true; }
    bool reset() override { k_ = -1; return true; }
    bool is_stepping_required() const override { return true; }
    std::tuple<bool, cv::Mat> flow(const bool&) override
    {
        if (k_ < 0 || k_ >= s_.n || !s_.frames[k_].has_flow) return std::make_tuple(false, cv::Mat());
        return std::make_tuple(true, cv::Mat(s_.H, s_.W, CV_32FC2, const_cast<float*>(s_.frames[k_].flow.data())));
    }
    std::size_t get_grid_size() const override { return 1; }
    float get_scaling_factor() const override { return 1.0f; }
    int get_matrix_type() const override { return CV_32FC2; }

private:
    const RecordedStream& s_;
    int k_ = -1;
};

// The same recording from buffers of the library's pinned pool (cv::Mat allocates image-sized buffers there): what makes
// ROFT::ROFTFilter hand its inputs to the engine in place.  `rewrite` = ONE flow matrix rewritten every frame, the way a live
// source with a single output buffer behaves (every cv::Mat copy shares it): the filter must not keep referring to it.
class MemSegmentationPooled : public MemSegmentation {
public:
    using MemSegmentation::MemSegmentation;
    std::pair<bool, cv::Mat> segmentation(const bool& b) override
    {
        auto r = MemSegmentation::segmentation(b);
        if (r.first) r.second = r.second.clone();
        return r;
    }
};

class MemFlowPooled : public MemFlow {
public:
    MemFlowPooled(const RecordedStream& s, bool rewrite) : MemFlow(s), rewrite_(rewrite), one_(s.H, s.W, CV_32FC2) {}
    std::tuple<bool, cv::Mat> flow(const bool& b) override
    {
        auto r = MemFlow::flow(b);
        if (!std::get<0>(r)) return r;
        if (!rewrite_) return std::make_tuple(true, std::get<1>(r).clone());
        std::memcpy(one_.data, std::get<1>(r).data, one_.total() * one_.elemSize());
        return std::make_tuple(true, one_);
    }
    bool flow_buffers_are_immutable() const override { return !rewrite_; }

private:
    bool rewrite_;
    cv::Mat one_;
};

// poses are polled once per frame: freeze() advances the cursor and says whether this frame delivers one
class MemPose : public RobotsIO::Utils::Transform {
public:
    MemPose(const RecordedStream& s, int frames_between) : s_(s), fb_(frames_between) {}
    bool freeze(const bool = false) override
    {
        ++k_;
        if (k_ < 0 || k_ >= s_.n || !s_.frames[k_].has_pose) return false;
        for (int i = 0; i < 3; ++i) T_.translation()[i] = s_.frames[k_].pose[i];
        for (int i = 0; i < 4; ++i) T_.quaternion()[i] = s_.frames[k_].pose[3 + i];
        return true;
    }
    Eigen::Transform<double, 3, Eigen::Affine> transform() override { return T_; }
    int get_frames_between_iterations() const override { return fb_; }
    void rewind() { k_ = -1; }

private:
    const RecordedStream& s_;
    int fb_, k_ = -1;
    Eigen::Transform<double, 3, Eigen::Affine> T_;
};
