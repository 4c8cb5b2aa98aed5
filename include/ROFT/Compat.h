// Compat.h -- stand-ins for the third-party types the reference's filter classes are written against: Eigen (dense
// matrices, Ref, Transform), BayesFilters (bfl::Data, VectorDescription, Gaussian, the model / prediction / correction
// base classes, KFPrediction, UKFPrediction, Logger, FilteringAlgorithm), OpenCV (cv::Mat as an image buffer) and RobotsIO
// (camera parameters, the Segmentation / Transform / SpatialVelocity source interfaces, Probe / ProbeContainer / ImageFileProbe).  None of them is installed where this
// repository is built; the facade classes of include/ROFT/ keep the reference's class names, constructor signatures and
// virtuals over these types, and a maintainer integrating into the real ROFT tree drops this file (define
// ROFT_HAVE_REAL_DEPENDENCIES and include the real headers first): the facades only use the members declared here.
//
// Only the surface the reference's headers mention is provided -- these are not general-purpose replacements.
// Matrices are dense, ROW-major double / float storage.
#pragma once

#include <any>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <deque>
#include <fstream>
#include <memory>
#include <ostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../roft_engine.h"

// Image-sized buffers of the stand-in matrix types live in pinned, device-mapped host memory when the library finds a device
// (roft_host_alloc, roft_engine.h section 2b): ROFT::ROFTFilter then hands them to the engine as they are -- the kernels read the
// few hundred KB of a frame they need in place, nothing is staged or uploaded -- and smaller buffers in ordinary memory.
// (weak references: a program that uses only the stand-in types and does not link libroft_hip.so -- a log writer, a file
//  converter -- still links; its buffers are ordinary memory then)
#pragma weak roft_host_alloc
#pragma weak roft_host_free
#pragma weak roft_host_is_pinned
namespace ROFT {
namespace compat {
template <class T>
struct ImageAllocator {
    using value_type = T;
    static constexpr std::size_t kPinFrom = (std::size_t)64 << 10;   // bytes
    ImageAllocator() = default;
    template <class U> ImageAllocator(const ImageAllocator<U>&) {}
    T* allocate(std::size_t n)
    {
        const std::size_t bytes = n * sizeof(T);
        if (bytes >= kPinFrom && roft_host_alloc)
            if (void* p = roft_host_alloc(bytes)) return static_cast<T*>(p);
        void* p = std::malloc(bytes ? bytes : 1);
        if (!p) throw std::bad_alloc();
        return static_cast<T*>(p);
    }
    void deallocate(T* p, std::size_t n)
    {
        if (n * sizeof(T) >= kPinFrom && roft_host_is_pinned && roft_host_is_pinned(p)) roft_host_free(p);
        else std::free(p);
    }
    template <class U> bool operator==(const ImageAllocator<U>&) const { return true; }
    template <class U> bool operator!=(const ImageAllocator<U>&) const { return false; }
};

// microseconds the file-backed sources of this process spent loading since the counter was last taken (ROFTFilter's
// ROFT_FILTER_TIMING report; the sources themselves report whole milliseconds, the reference's resolution: CompatIO.h)
inline double& loading_us_counter() { static double us = 0.0; return us; }
inline double take_loading_us() { const double v = loading_us_counter(); loading_us_counter() = 0.0; return v; }
}  // namespace compat
}  // namespace ROFT

#ifndef ROFT_HAVE_REAL_DEPENDENCIES

// ---- Eigen ---------------------------------------------------------------------------------------------------
namespace Eigen {

template <class S>
class DenseMatrix;

// `v.head<3>()`, `v.segment<3>(i)`, `v.tail<4>()` of a vector: a window that can be assigned to and indexed, as
// src/roft/src/main.cpp:286-325 composes the filter's parameter vectors
template <class S>
class VectorBlock {
public:
    VectorBlock(S* p, std::size_t n) : p_(p), n_(n) {}
    std::size_t size() const { return n_; }
    S& operator()(std::size_t i) { return p_[i]; }
    const S& operator()(std::size_t i) const { return p_[i]; }
    VectorBlock& operator=(const DenseMatrix<S>& v);
    VectorBlock& operator=(const VectorBlock& v)
    {
        if (v.n_ != n_) throw std::runtime_error("Eigen stand-in: block assignment of mismatching size");
        for (std::size_t i = 0; i < n_; ++i) p_[i] = v.p_[i];
        return *this;
    }

private:
    S* p_;
    std::size_t n_;
};

// A dense row-major matrix with Eigen's value semantics.  The storage is shared between copies and detached by the first
// mutating access (copy on write): a depth image travels from the camera through CameraMeasurement::measure()'s tuple to the
// filter without being copied three times per frame, and the pointer the engine reads stays the one the file was read into.
template <class S>
class DenseMatrix {
    using Store = std::vector<S, ROFT::compat::ImageAllocator<S>>;

public:
    DenseMatrix() = default;
    DenseMatrix(const VectorBlock<S>& b) : r_(b.size()), c_(1), d_(std::make_shared<Store>(b.size()))
    {
        for (std::size_t i = 0; i < d_->size(); ++i) (*d_)[i] = b(i);
    }
    VectorBlock<S> head(std::size_t n) { return block(0, n); }
    VectorBlock<S> tail(std::size_t n) { return block(size() - n, n); }
    VectorBlock<S> segment(std::size_t i, std::size_t n) { return block(i, n); }
    template <int N> VectorBlock<S> head() { return block(0, N); }
    template <int N> VectorBlock<S> tail() { return block(size() - N, N); }
    template <int N> VectorBlock<S> segment(std::size_t i) { return block(i, N); }
    DenseMatrix(std::size_t r, std::size_t c) : r_(r), c_(c), d_(std::make_shared<Store>(r * c, S(0))) {}
    explicit DenseMatrix(std::size_t n) : r_(n), c_(1), d_(std::make_shared<Store>(n, S(0))) {}
    static DenseMatrix Zero(std::size_t r, std::size_t c = 1) { return DenseMatrix(r, c); }
    static DenseMatrix Identity(std::size_t r, std::size_t c)
    {
        DenseMatrix m(r, c);
        for (std::size_t i = 0; i < r && i < c; ++i) m(i, i) = S(1);
        return m;
    }
    void resize(std::size_t r, std::size_t c = 1) { r_ = r; c_ = c; d_ = std::make_shared<Store>(r * c, S(0)); }
    std::size_t rows() const { return r_; }
    std::size_t cols() const { return c_; }
    std::size_t size() const { return d_ ? d_->size() : 0; }
    S& operator()(std::size_t i, std::size_t j = 0) { return own()[i * c_ + j]; }
    const S& operator()(std::size_t i, std::size_t j = 0) const { return (*d_)[i * c_ + j]; }
    S* data() { return size() ? own().data() : nullptr; }
    const S* data() const { return d_ ? d_->data() : nullptr; }
    DenseMatrix transpose() const
    {
        DenseMatrix m(c_, r_);
        for (std::size_t i = 0; i < r_; ++i)
            for (std::size_t j = 0; j < c_; ++j) m(j, i) = (*this)(i, j);
        return m;
    }
    // diag(v) of a vector, as `v.asDiagonal()` is used in the reference's constructors
    DenseMatrix asDiagonal() const
    {
        DenseMatrix m(size(), size());
        for (std::size_t i = 0; i < size(); ++i) m(i, i) = (*d_)[i];
        return m;
    }

private:
    // the storage for writing: detached from the copies that share it
    Store& own()
    {
        if (!d_) d_ = std::make_shared<Store>();
        else if (d_.use_count() > 1) d_ = std::make_shared<Store>(*d_);
        return *d_;
    }
    VectorBlock<S> block(std::size_t i, std::size_t n)
    {
        if (i + n > size() || (r_ != 1 && c_ != 1)) throw std::runtime_error("Eigen stand-in: block outside of the vector");
        return VectorBlock<S>(own().data() + i, n);
    }
    std::size_t r_ = 0, c_ = 0;
    std::shared_ptr<Store> d_;
};
template <class S>
VectorBlock<S>& VectorBlock<S>::operator=(const DenseMatrix<S>& v)
{
    if (v.size() != n_) throw std::runtime_error("Eigen stand-in: block assignment of mismatching size");
    for (std::size_t i = 0; i < n_; ++i) p_[i] = v.data()[i];
    return *this;
}
// `stream << matrix` with Eigen's default IOFormat: the stream's precision, coefficients separated by one space and padded
// to the width of the widest one, rows by a newline -- what bfl::Logger writes into the reference's log files
template <class S>
std::ostream& operator<<(std::ostream& os, const DenseMatrix<S>& m)
{
    std::vector<std::string> cell(m.size());
    std::size_t width = 0;
    for (std::size_t i = 0; i < m.rows(); ++i)
        for (std::size_t j = 0; j < m.cols(); ++j) {
            std::ostringstream ss;
            ss.copyfmt(os);
            ss.width(0);
            ss << m(i, j);
            cell[i * m.cols() + j] = ss.str();
            width = std::max(width, cell[i * m.cols() + j].size());
        }
    for (std::size_t i = 0; i < m.rows(); ++i) {
        if (i) os << "\n";
        for (std::size_t j = 0; j < m.cols(); ++j) {
            if (j) os << " ";
            const std::string& c = cell[i * m.cols() + j];
            os << std::string(width - c.size(), ' ') << c;
        }
    }
    return os;
}
using MatrixXd = DenseMatrix<double>;
using VectorXd = DenseMatrix<double>;   // n x 1
using MatrixXf = DenseMatrix<float>;    // depth images: (v, u)

// `Eigen::Ref<const Eigen::MatrixXd>` in a signature accepts a matrix by reference
template <class T>
using Ref = T&;

// rotation of `angle` about a UNIT `axis` and the quaternion of it (main.cpp:290: Quaterniond(AngleAxisd(angle, axis)))
template <class S>
class AngleAxis {
public:
    template <class Vec>
    AngleAxis(const S& angle, const Vec& axis) : angle_(angle) { for (int i = 0; i < 3; ++i) axis_[i] = axis(i); }
    const S& angle() const { return angle_; }
    const S* axis() const { return axis_; }

private:
    S angle_, axis_[3];
};
template <class S>
class Quaternion {
public:
    Quaternion(const S& w, const S& x, const S& y, const S& z) : q_{w, x, y, z} {}
    Quaternion(const AngleAxis<S>& aa)
    {
        const S h = aa.angle() / S(2), s = std::sin(h);
        q_[0] = std::cos(h);
        for (int i = 0; i < 3; ++i) q_[1 + i] = s * aa.axis()[i];
    }
    const S& w() const { return q_[0]; }
    const S& x() const { return q_[1]; }
    const S& y() const { return q_[2]; }
    const S& z() const { return q_[3]; }

private:
    S q_[4];
};
using AngleAxisd = AngleAxis<double>;
using Quaterniond = Quaternion<double>;

enum TransformTraits { Affine = 1 };
// rigid transform as the pose sources deliver it: translation + unit quaternion (w, x, y, z)
template <class S, int Dim, int Mode>
class Transform {
public:
    Transform() { t_[0] = t_[1] = t_[2] = S(0); q_[0] = S(1); q_[1] = q_[2] = q_[3] = S(0); }
    S* translation() { return t_; }
    const S* translation() const { return t_; }
    S* quaternion() { return q_; }
    const S* quaternion() const { return q_; }

private:
    S t_[3], q_[4];
};

}  // namespace Eigen

// ---- OpenCV ---------------------------------------------------------------------------------------------------
#ifndef CV_8UC1
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_16SC2 11
#define CV_32FC2 13
#endif

namespace cv {

struct Vec2f { float v[2]; float operator()(int i) const { return v[i]; } };
struct Vec2s { short v[2]; short operator()(int i) const { return v[i]; } };

// an image buffer: rows x cols elements of `type`, row-major, reference counted like cv::Mat
class Mat {
public:
    Mat() = default;
    Mat(int rows, int cols, int type) : rows(rows), cols(cols), type_(type), buf_(std::make_shared<Store>((std::size_t)rows * cols * elem(type), 0)) { data = buf_->data(); }
    // wraps caller memory (no ownership), like cv::Mat(rows, cols, type, void*)
    Mat(int rows, int cols, int type, void* external) : rows(rows), cols(cols), data(static_cast<unsigned char*>(external)), type_(type) {}
    int type() const { return type_; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    std::size_t total() const { return (std::size_t)rows * cols; }
    std::size_t elemSize() const { return elem(type_); }
    Mat clone() const
    {
        Mat m(rows, cols, type_);
        if (!empty()) std::memcpy(m.data, data, total() * elemSize());
        return m;
    }
    template <class T> T& at(int r, int c) { return reinterpret_cast<T*>(data)[(std::size_t)r * cols + c]; }
    template <class T> const T& at(int r, int c) const { return reinterpret_cast<const T*>(data)[(std::size_t)r * cols + c]; }
    int rows = 0, cols = 0;
    unsigned char* data = nullptr;

private:
    static std::size_t elem(int type) { return type == CV_32FC2 ? 8 : (type == CV_16SC2 ? 4 : (type == CV_8UC3 ? 3 : 1)); }
    int type_ = CV_8UC1;
    using Store = std::vector<unsigned char, ROFT::compat::ImageAllocator<unsigned char>>;
    std::shared_ptr<Store> buf_;
};

}  // namespace cv

// ---- RobotsIO ---------------------------------------------------------------------------------------------------
namespace RobotsIO {
namespace Camera {

class CameraParameters {
public:
    std::size_t width() const { return width_; }
    std::size_t height() const { return height_; }
    double fx() const { return fx_; }
    double fy() const { return fy_; }
    double cx() const { return cx_; }
    double cy() const { return cy_; }
    void width(std::size_t v) { width_ = v; }
    void height(std::size_t v) { height_ = v; }
    void fx(double v) { fx_ = v; }
    void fy(double v) { fy_ = v; }
    void cx(double v) { cx_ = v; }
    void cy(double v) { cy_ = v; }
    bool initialized() const { return width_ > 0 && height_ > 0; }

private:
    std::size_t width_ = 0, height_ = 0;
    double fx_ = 0, fy_ = 0, cx_ = 0, cy_ = 0;
};

// RGB-D source: what ROFT::CameraMeasurement polls (RobotsIO::Camera::Camera)
class Camera {
public:
    virtual ~Camera() = default;
    virtual bool status() const { return true; }
    virtual bool step_frame() { return true; }
    virtual bool reset() { return true; }
    virtual std::pair<bool, CameraParameters> parameters() const = 0;
    virtual std::pair<bool, Eigen::MatrixXf> depth(const bool& blocking) = 0;      // metres, (v, u)
    virtual std::pair<bool, cv::Mat> rgb(const bool& /*blocking*/) { return {false, cv::Mat()}; }
    virtual std::pair<bool, Eigen::Transform<double, 3, Eigen::Affine>> pose(const bool&) { return {true, {}}; }
    virtual std::pair<bool, double> time_stamp_rgb() const { return {false, 0.0}; }
    virtual std::pair<bool, double> time_stamp_depth() const { return {false, 0.0}; }
    virtual std::int32_t frame_index() const { return -1; }
};

}  // namespace Camera

namespace Utils {

// key/value parameter bag with the accessor shape of RobotsIO's `robots_io_declare_*_field` macros: p.name() / p.name(v)
class Parameters {
public:
    virtual ~Parameters() = default;
};

class Segmentation {
public:
    virtual ~Segmentation() = default;
    virtual bool reset() { return true; }
    virtual bool step_frame() { return true; }
    virtual bool is_stepping_required() const = 0;
    virtual void reset_data_loading_time() {}
    virtual double get_data_loading_time() const { return 0.0; }
    virtual int get_frames_between_iterations() const { return -1; }
    // (valid, mask): valid == a NEW mask arrived with this frame
    virtual std::pair<bool, cv::Mat> segmentation(const bool& blocking) = 0;
    virtual std::pair<bool, cv::Mat> latest_segmentation() { return {false, cv::Mat()}; }
    virtual double get_time_stamp() { return -1.0; }   // stamp of the image the delivered mask was computed on (s)
    virtual void set_rgb_image(const cv::Mat& /*image*/, const double& /*timestamp*/) {}
};

class Transform {
public:
    virtual ~Transform() = default;
    virtual Eigen::Transform<double, 3, Eigen::Affine> transform() = 0;
    virtual bool freeze(const bool blocking = false) = 0;     // true: a new pose was received
    virtual int get_frames_between_iterations() const { return -1; }
    virtual bool transform_received() { return false; }
};

class SpatialVelocity {
public:
    virtual ~SpatialVelocity() = default;
    virtual bool freeze(const bool blocking = false) = 0;
    // twist expressed at the camera origin: v_O (3), omega (3)
    virtual const double* linear_velocity_origin() = 0;
    virtual const double* angular_velocity() = 0;
    virtual bool is_screw_degenerate() { return false; }
};

class SpatialVelocityBuffer : public SpatialVelocity {
public:
    void set_twist(const double linear[3], const double angular[3], const double elapsed_time = 0.0)
    {
        (void)elapsed_time;
        for (int i = 0; i < 3; ++i) { v_[i] = linear[i]; w_[i] = angular[i]; }
    }
    bool freeze(const bool = false) override { return true; }
    const double* linear_velocity_origin() override { return v_; }
    const double* angular_velocity() override { return w_; }

private:
    double v_[3] = {0, 0, 0}, w_[3] = {0, 0, 0};
};

// RobotsIO::Utils::Probe / ProbeContainer (robots-io, src/RobotsIO/include/RobotsIO/Utils/{Probe,ProbeContainer}.h): named
// sinks a filter hands its outputs to -- ROFTFilter offers output_pose, output_velocity, output_segmentation and
// output_segmentation_refined (ROFTFilter.cpp:396-451), ROFT-tracker attaches image file probes to the last two
// (src/roft/src/main.cpp:403-417)
using Data = std::any;

class Probe {
public:
    virtual ~Probe() = default;
    void set_data(const Data& data)
    {
        data_ = data;
        on_new_data();
    }

protected:
    virtual void on_new_data() = 0;
    Data get_data() { return data_; }

private:
    Data data_;
};

class ProbeContainer {
public:
    virtual ~ProbeContainer() = default;
    bool set_probe(const std::string& name, std::unique_ptr<Probe> probe)
    {
        probes_[name] = std::move(probe);
        return true;
    }
    Probe& get_probe(const std::string& name) const { return *(probes_.at(name)); }
    bool is_probe(const std::string& name) const { return probes_.find(name) != probes_.end(); }

protected:
    std::unordered_map<std::string, std::unique_ptr<Probe>> probes_;
};

// 8-bit gray or BGR image -> PNG with stored (uncompressed) deflate blocks: enough for the debug images of the probes,
// no zlib needed
inline bool write_png(const std::string& path, const cv::Mat& img)
{
    const int ch = img.type() == CV_8UC3 ? 3 : (img.type() == CV_8UC1 ? 1 : 0);
    if (!ch || img.empty()) return false;
    static std::uint32_t crc_table[256];
    static bool have_table = false;
    if (!have_table) {
        for (std::uint32_t n = 0; n < 256; ++n) {
            std::uint32_t c = n;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            crc_table[n] = c;
        }
        have_table = true;
    }
    auto be32 = [](std::vector<unsigned char>& v, std::uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((unsigned char)(x >> s)); };
    std::vector<unsigned char> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    auto chunk = [&](const char* tag, const std::vector<unsigned char>& body) {
        be32(out, (std::uint32_t)body.size());
        std::uint32_t c = 0xFFFFFFFFu;
        auto upd = [&](unsigned char b) { c = crc_table[(c ^ b) & 0xFF] ^ (c >> 8); };
        for (int i = 0; i < 4; ++i) { out.push_back((unsigned char)tag[i]); upd((unsigned char)tag[i]); }
        for (unsigned char b : body) { out.push_back(b); upd(b); }
        be32(out, c ^ 0xFFFFFFFFu);
    };
    std::vector<unsigned char> hdr;
    be32(hdr, (std::uint32_t)img.cols);
    be32(hdr, (std::uint32_t)img.rows);
    hdr.insert(hdr.end(), {8, (unsigned char)(ch == 3 ? 2 : 0), 0, 0, 0});
    chunk("IHDR", hdr);
    // raw scanlines: filter byte 0 + the row (BGR -> RGB)
    std::vector<unsigned char> raw;
    raw.reserve((std::size_t)img.rows * (img.cols * ch + 1));
    for (int r = 0; r < img.rows; ++r) {
        raw.push_back(0);
        const unsigned char* row = img.data + (std::size_t)r * img.cols * ch;
        for (int c = 0; c < img.cols; ++c)
            for (int k = 0; k < ch; ++k) raw.push_back(row[c * ch + (ch == 3 ? 2 - k : k)]);
    }
    std::vector<unsigned char> z = {0x78, 0x01};
    std::uint32_t a = 1, b = 0;
    for (std::size_t off = 0; off < raw.size();) {
        const std::size_t n = std::min<std::size_t>(65535, raw.size() - off);
        z.push_back(off + n >= raw.size() ? 1 : 0);
        z.push_back((unsigned char)(n & 0xFF)); z.push_back((unsigned char)(n >> 8));
        z.push_back((unsigned char)(~n & 0xFF)); z.push_back((unsigned char)((~n >> 8) & 0xFF));
        for (std::size_t i = 0; i < n; ++i) { z.push_back(raw[off + i]); a = (a + raw[off + i]) % 65521u; b = (b + a) % 65521u; }
        off += n;
    }
    be32(z, (b << 16) | a);
    chunk("IDAT", z);
    chunk("IEND", {});
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    f.write(reinterpret_cast<const char*>(out.data()), (std::streamsize)out.size());
    return (bool)f;
}

// RobotsIO::Utils::ImageFileProbe: one image file per set_data(cv::Mat), <path>/<prefix_><counter>.<format>
// (only "png" is written here: 8-bit gray or BGR)
class ImageFileProbe : public Probe {
public:
    ImageFileProbe(const std::string& output_path, const std::string& prefix, const std::string& output_format)
        : output_prefix_(output_path), output_format_(output_format)
    {
        if (!output_prefix_.empty() && output_prefix_.back() != '/') output_prefix_ += '/';
        if (!prefix.empty()) output_prefix_ += prefix + "_";
    }

protected:
    void on_new_data() override
    {
        const cv::Mat img = std::any_cast<cv::Mat>(get_data());
        if (!write_png(output_prefix_ + std::to_string(frame_counter_) + "." + output_format_, img))
            throw std::runtime_error("ImageFileProbe::on_new_data. Error: cannot write " + output_prefix_ + std::to_string(frame_counter_));
        frame_counter_++;
    }

private:
    std::string output_prefix_, output_format_;
    std::size_t frame_counter_ = 0;
};

}  // namespace Utils
}  // namespace RobotsIO

// ---- BayesFilters ---------------------------------------------------------------------------------------------------
namespace bfl {

namespace any {
using std::any_cast;
using any = std::any;
}  // namespace any
using Data = std::any;

// sizes of a state / measurement / noise vector: `circular` components are unit quaternions (4 numbers, 3 degrees of
// freedom), as ROFT uses them (VectorDescription(lin, circ, noise, CircularType::Quaternion))
class VectorDescription {
public:
    enum class CircularType { Euler, Quaternion };
    VectorDescription(std::size_t linear = 0, std::size_t circular = 0, std::size_t noise = 0,
                      CircularType type = CircularType::Quaternion)
        : lin_(linear), circ_(circular), noise_(noise), type_(type) {}
    std::size_t linear_size() const { return lin_; }
    std::size_t circular_size() const { return circ_; }
    std::size_t noise_size() const { return noise_; }
    std::size_t total_size() const { return lin_ + (type_ == CircularType::Quaternion ? 4 : 1) * circ_ + noise_; }
    std::size_t dof_size() const { return lin_ + (type_ == CircularType::Quaternion ? 3 : 1) * circ_ + noise_; }

private:
    std::size_t lin_, circ_, noise_;
    CircularType type_;
};

// Gaussian(dim_linear, dim_circular, use_quaternion): mean size lin + 4 circ, covariance lin + 3 circ -- the sizes
// ROFTFilter relies on: Gaussian(9, 1, true) -> 13 / 12 x 12, Gaussian(6) -> 6 / 6 x 6 (ROFTFilter.cpp:64-67)
class Gaussian {
public:
    Gaussian() : Gaussian(1, 0, false) {}
    explicit Gaussian(std::size_t dim_linear, std::size_t dim_circular = 0, bool use_quaternion = false)
        : dim_linear(dim_linear), dim_circular(dim_circular), use_quaternion(use_quaternion),
          mean_(dim_linear + (use_quaternion ? 4 : 1) * dim_circular, 1),
          cov_(dim_linear + (use_quaternion ? 3 : 1) * dim_circular, dim_linear + (use_quaternion ? 3 : 1) * dim_circular)
    {}
    Eigen::VectorXd& mean() { return mean_; }
    const Eigen::VectorXd& mean() const { return mean_; }
    double& mean(std::size_t i) { return mean_(i); }
    Eigen::MatrixXd& covariance() { return cov_; }
    const Eigen::MatrixXd& covariance() const { return cov_; }
    std::size_t dim_linear, dim_circular;
    bool use_quaternion;

private:
    Eigen::VectorXd mean_;
    Eigen::MatrixXd cov_;
};
using GaussianMixture = Gaussian;  // the reference only ever uses one component (SKFCorrection.cpp:39)

// bfl::Logger (bayes-filters-lib, src/BayesFilters/include/BayesFilters/Logger.h): enable_log(folder, prefix) opens one
// text file per name of log_file_names(), `<name>.txt`, appending; logger(a, b, ...) writes its i-th argument and a
// newline to the i-th file.  Base class of the measurement models and of FilteringAlgorithm.
class Logger {
public:
    virtual ~Logger() = default;
    bool enable_log(const std::string& folder_path, const std::string& file_name_prefix)
    {
        if (log_enabled_) return false;
        folder_path_ = folder_path;
        if (!folder_path_.empty() && folder_path_.back() == '/') folder_path_.pop_back();
        file_name_prefix_ = file_name_prefix;
        file_names_ = log_file_names(folder_path_, file_name_prefix_);
        log_files_.clear();
        for (const std::string& name : file_names_) {
            log_files_.emplace_back(name + ".txt", std::ofstream::out | std::ofstream::app);
            if (!log_files_.back().is_open()) { log_files_.clear(); return false; }
        }
        log_enabled_ = true;
        return true;
    }
    bool disable_log()
    {
        if (!log_enabled_) return false;
        log_files_.clear();
        log_enabled_ = false;
        return true;
    }
    std::string get_folder_path() const { return folder_path_; }
    std::string get_file_name_prefix() const { return file_name_prefix_; }
    template <class... Data>
    void logger(const Data&... data)
    {
        if (!log_enabled_) return;
        std::size_t i = 0;
        ((i < log_files_.size() ? void(log_files_[i] << data << std::endl) : void(0), ++i), ...);
    }

protected:
    virtual std::vector<std::string> log_file_names(const std::string& /*folder_path*/, const std::string& /*file_name_prefix*/) { return {}; }
    virtual void log() {}

private:
    std::string folder_path_, file_name_prefix_;
    std::vector<std::string> file_names_;
    std::vector<std::ofstream> log_files_;
    bool log_enabled_ = false;
};

class MeasurementModel : public Logger {
public:
    virtual ~MeasurementModel() = default;
    virtual bool freeze(const Data& data = Data()) = 0;
    virtual std::pair<bool, Data> measure(const Data& data = Data()) const = 0;
    virtual std::pair<bool, Data> predictedMeasure(const Eigen::Ref<const Eigen::MatrixXd>& cur_states) const = 0;
    virtual std::pair<bool, Data> innovation(const Data& predicted_measurements, const Data& measurements) const = 0;
    virtual std::pair<bool, Eigen::MatrixXd> getNoiseCovarianceMatrix() const { return {false, Eigen::MatrixXd()}; }
    virtual VectorDescription getInputDescription() const { return VectorDescription(); }
    virtual VectorDescription getMeasurementDescription() const { return VectorDescription(); }
    virtual bool setProperty(const std::string& /*property*/) { return false; }
};

// bfl::LinearMeasurementModel: y_hat = H x per state column, innovation = y - y_hat (what the library's class implements, so
// that a model only states H, y and R -- the velocity filter's measurement model, tests/ref_kit/replay.cpp's RecordedLinear)
class LinearMeasurementModel : public MeasurementModel {
public:
    virtual Eigen::MatrixXd getMeasurementMatrix() const = 0;
    std::pair<bool, Data> predictedMeasure(const Eigen::Ref<const Eigen::MatrixXd>& cur_states) const override
    {
        const Eigen::MatrixXd H = getMeasurementMatrix();
        if (H.cols() != cur_states.rows()) return {false, Data()};
        Eigen::MatrixXd out(H.rows(), cur_states.cols());
        for (std::size_t i = 0; i < H.rows(); ++i)
            for (std::size_t c = 0; c < cur_states.cols(); ++c) {
                double s = 0.0;
                for (std::size_t k = 0; k < H.cols(); ++k) s += H(i, k) * cur_states(k, c);
                out(i, c) = s;
            }
        return {true, Data(out)};
    }
    std::pair<bool, Data> innovation(const Data& predicted_measurements, const Data& measurements) const override
    {
        const Eigen::MatrixXd* p = any::any_cast<Eigen::MatrixXd>(&predicted_measurements);
        const Eigen::MatrixXd* m = any::any_cast<Eigen::MatrixXd>(&measurements);
        if (!p || !m || p->rows() != m->rows()) return {false, Data()};
        Eigen::MatrixXd out(p->rows(), p->cols());
        for (std::size_t i = 0; i < p->rows(); ++i)
            for (std::size_t c = 0; c < p->cols(); ++c) out(i, c) = (*m)(i, m->cols() == p->cols() ? c : 0) - (*p)(i, c);
        return {true, Data(out)};
    }
};

class StateModel {
public:
    virtual ~StateModel() = default;
    virtual void propagate(const Eigen::Ref<const Eigen::MatrixXd>& cur_states, Eigen::Ref<Eigen::MatrixXd> mot_states) = 0;
    virtual void motion(const Eigen::Ref<const Eigen::MatrixXd>& cur_states, Eigen::Ref<Eigen::MatrixXd> mot_states) = 0;
    virtual Eigen::MatrixXd getNoiseCovarianceMatrix() = 0;
    virtual bool setSamplingTime(const double& /*sample_time*/) { return false; }
    virtual bool setProperty(const std::string& /*property*/) { return false; }
    virtual VectorDescription getInputDescription() = 0;
    virtual VectorDescription getStateDescription() = 0;
};

class LinearStateModel : public StateModel {
public:
    virtual Eigen::MatrixXd getStateTransitionMatrix() = 0;
    void propagate(const Eigen::Ref<const Eigen::MatrixXd>& cur, Eigen::Ref<Eigen::MatrixXd> mot) override
    {
        const Eigen::MatrixXd F = getStateTransitionMatrix();
        mot.resize(F.rows(), cur.cols());
        for (std::size_t i = 0; i < F.rows(); ++i)
            for (std::size_t c = 0; c < cur.cols(); ++c) {
                double s = 0.0;
                for (std::size_t k = 0; k < F.cols(); ++k) s += F(i, k) * cur(k, c);
                mot(i, c) = s;
            }
    }
    void motion(const Eigen::Ref<const Eigen::MatrixXd>& cur, Eigen::Ref<Eigen::MatrixXd> mot) override { propagate(cur, mot); }
};

class GaussianPrediction {
public:
    virtual ~GaussianPrediction() = default;
    void predict(const GaussianMixture& prev_state, GaussianMixture& pred_state) { predictStep(prev_state, pred_state); }
    virtual StateModel& getStateModel() = 0;

protected:
    virtual void predictStep(const GaussianMixture& prev_state, GaussianMixture& pred_state) = 0;
};

class GaussianCorrection {
public:
    virtual ~GaussianCorrection() = default;
    void correct(const GaussianMixture& pred_state, GaussianMixture& corr_state) { correctStep(pred_state, corr_state); }
    virtual MeasurementModel& getMeasurementModel() = 0;
    virtual std::pair<bool, Eigen::VectorXd> getLikelihood() { return {false, Eigen::VectorXd()}; }

protected:
    virtual void correctStep(const GaussianMixture& pred_state, GaussianMixture& corr_state) = 0;
};

// the filter loop of bfl::FilteringAlgorithm, run on the calling thread: boot() arms it, run() executes
// initialization_step() and then filtering_step() while run_condition() holds, wait() returns when it is over
class FilteringAlgorithm : public Logger {
public:
    virtual ~FilteringAlgorithm() = default;
    bool boot() { booted_ = true; return true; }
    void run()
    {
        if (!booted_) boot();
        teardown_ = false;
        if (!initialization_step()) return;
        step_ = 0;
        while (run_condition() && !teardown_) {
            filtering_step();
            ++step_;
        }
    }
    bool wait() { return true; }
    bool teardown() { teardown_ = true; return true; }
    unsigned int step_number() const { return step_; }
    virtual bool skip(const std::string& /*what_step*/, const bool /*status*/) { return false; }

protected:
    virtual bool initialization_step() = 0;
    virtual void filtering_step() = 0;
    virtual bool run_condition() = 0;

private:
    bool booted_ = false, teardown_ = false;
    unsigned int step_ = 0;
};

}  // namespace bfl

#endif  // ROFT_HAVE_REAL_DEPENDENCIES

namespace ROFT {
namespace compat {

// thrown where the reference throws std::runtime_error; carries roft_last_error_string()
inline void throw_if(int rc, const char* what)
{
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + roft_last_error_string());
}

// diagonal of a covariance given either as an n x n matrix or as an n-vector (the reference passes `v.asDiagonal()`)
inline void diagonal_of(const Eigen::MatrixXd& m, double* out, std::size_t n)
{
    if (m.rows() == n && m.cols() == n) for (std::size_t i = 0; i < n; ++i) out[i] = m(i, i);
    else if (m.size() == n) for (std::size_t i = 0; i < n; ++i) out[i] = m.data()[i];
    else throw std::runtime_error("covariance of unexpected size");
}

// cv::cvtColor(COLOR_BGR2GRAY) on 8-bit data (fixed point, 14 fractional bits)
inline cv::Mat bgr_to_gray(const cv::Mat& m)
{
    if (m.type() != CV_8UC3) return m;
    cv::Mat g(m.rows, m.cols, CV_8UC1);
    for (std::size_t p = 0; p < m.total(); ++p)
        g.data[p] = (unsigned char)((m.data[3 * p + 2] * 4899 + m.data[3 * p + 1] * 9617 + m.data[3 * p] * 1868 + 8192) >> 14);
    return g;
}

// names kept from the first version of the facade
using Gaussian = bfl::Gaussian;
using MatrixXd = Eigen::MatrixXd;
using VectorXd = Eigen::VectorXd;

}  // namespace compat
}  // namespace ROFT
