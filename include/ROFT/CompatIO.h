// CompatIO.h -- the file-backed sources ROFT-tracker builds around the filter (src/roft/src/main.cpp:327-381), over the
// stand-in types of Compat.h:
//   RobotsIO::Camera::DatasetCamera            RGB-D frames of a Fast-YCB / HO-3D style directory (data.txt, rgb/, depth/)
//   RobotsIO::Utils::DatasetTransform          one pose per row of a text file (x y z axis angle), all-zero row = no pose
//   RobotsIO::Utils::DatasetTransformDelayed   the same at a reduced rate, delivered one period late
// RobotsIO is a third-party library that is not part of the reference checkout: the classes here follow the constructor
// calls of main.cpp:327-352, the file formats of SURVEY.md App. B (the `.float` depth frame: two size_t -- width, height --
// then width x height floats; data.txt: stamp_rgb stamp_depth x y z axis angle per frame) and, for the delayed delivery,
// the schedule the reference's own DatasetImageSegmentationDelayed implements next to it
// (src/roft-lib/src/DatasetImageSegmentationDelayed.cpp:42-63).  Host-side plumbing: no arithmetic of the hot path.
// Also here: the PNG decoder the mask and colour images go through (8-bit, non-interlaced; its own inflate, so that the
// facade stays header-only and needs no zlib).
#pragma once

#include <chrono>
#include <cstdio>
#include <iomanip>
#include <iostream>

#include "Compat.h"

namespace ROFT {
namespace compat {

// ---- inflate (RFC 1950 / 1951) --------------------------------------------------------------------------------
class Inflate {
public:
    Inflate(const unsigned char* in, std::size_t n) : in_(in), n_(n) {}
    // zlib stream -> bytes; throws std::runtime_error on malformed input
    std::vector<unsigned char> run()
    {
        if (n_ < 2 || (in_[0] & 0x0F) != 8 || ((in_[0] << 8 | in_[1]) % 31) != 0 || (in_[1] & 0x20)) fail("not a zlib stream");
        pos_ = 2;
        bool last = false;
        while (!last) {
            last = bits(1);
            const unsigned type = bits(2);
            if (type == 0) stored();
            else if (type == 1) { fixed_tables(); codes(); }
            else if (type == 2) { dynamic_tables(); codes(); }
            else fail("bad block type");
        }
        return std::move(out_);
    }

private:
    struct Huffman { std::uint16_t count[16]; std::uint16_t symbol[320]; };
    [[noreturn]] static void fail(const char* what) { throw std::runtime_error(std::string("inflate: ") + what); }
    unsigned bits(int need)
    {
        while (cnt_ < need) {
            if (pos_ >= n_) fail("truncated stream");
            buf_ |= (std::uint32_t)in_[pos_++] << cnt_;
            cnt_ += 8;
        }
        const unsigned v = buf_ & ((1u << need) - 1u);
        buf_ >>= need;
        cnt_ -= need;
        return v;
    }
    void stored()
    {
        buf_ = 0; cnt_ = 0;
        if (pos_ + 4 > n_) fail("truncated stored block");
        const unsigned len = in_[pos_] | in_[pos_ + 1] << 8, nlen = in_[pos_ + 2] | in_[pos_ + 3] << 8;
        pos_ += 4;
        if ((len ^ 0xFFFFu) != nlen || pos_ + len > n_) fail("bad stored block");
        out_.insert(out_.end(), in_ + pos_, in_ + pos_ + len);
        pos_ += len;
    }
    static void build(Huffman& h, const std::uint8_t* length, int n)
    {
        for (int i = 0; i < 16; ++i) h.count[i] = 0;
        for (int i = 0; i < n; ++i) h.count[length[i]]++;
        std::uint16_t offs[16];
        offs[1] = 0;
        for (int i = 1; i < 15; ++i) offs[i + 1] = offs[i] + h.count[i];
        for (int i = 0; i < n; ++i)
            if (length[i]) h.symbol[offs[length[i]]++] = (std::uint16_t)i;
        h.count[0] = 0;
    }
    int decode(const Huffman& h)
    {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; ++len) {
            code |= (int)bits(1);
            const int count = h.count[len];
            if (code - count < first) return h.symbol[index + (code - first)];
            index += count;
            first += count;
            first <<= 1;
            code <<= 1;
        }
        fail("bad Huffman code");
    }
    void fixed_tables()
    {
        std::uint8_t l[320];
        int i = 0;
        for (; i < 144; ++i) l[i] = 8;
        for (; i < 256; ++i) l[i] = 9;
        for (; i < 280; ++i) l[i] = 7;
        for (; i < 288; ++i) l[i] = 8;
        build(lit_, l, 288);
        for (i = 0; i < 30; ++i) l[i] = 5;
        build(dist_, l, 30);
    }
    void dynamic_tables()
    {
        static const std::uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        const int nlen = (int)bits(5) + 257, ndist = (int)bits(5) + 1, ncode = (int)bits(4) + 4;
        if (nlen > 286 || ndist > 30) fail("bad table sizes");
        std::uint8_t l[320] = {};
        for (int i = 0; i < ncode; ++i) l[order[i]] = (std::uint8_t)bits(3);
        Huffman cl;
        build(cl, l, 19);
        int i = 0;
        std::uint8_t len[320] = {};
        while (i < nlen + ndist) {
            int sym = decode(cl);
            if (sym < 16) len[i++] = (std::uint8_t)sym;
            else {
                int prev = 0, rep;
                if (sym == 16) { if (i == 0) fail("repeat without a length"); prev = len[i - 1]; rep = 3 + (int)bits(2); }
                else if (sym == 17) rep = 3 + (int)bits(3);
                else rep = 11 + (int)bits(7);
                if (i + rep > nlen + ndist) fail("too many lengths");
                while (rep--) len[i++] = (std::uint8_t)prev;
            }
        }
        build(lit_, len, nlen);
        build(dist_, len + nlen, ndist);
    }
    void codes()
    {
        static const std::uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const std::uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const std::uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const std::uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        for (;;) {
            int sym = decode(lit_);
            if (sym < 256) out_.push_back((unsigned char)sym);
            else if (sym == 256) return;
            else {
                sym -= 257;
                if (sym >= 29) fail("bad length symbol");
                const int len = lbase[sym] + (int)bits(lext[sym]);
                const int ds = decode(dist_);
                if (ds >= 30) fail("bad distance symbol");
                const std::size_t dist = dbase[ds] + bits(dext[ds]);
                if (dist > out_.size()) fail("distance too far back");
                for (int k = 0; k < len; ++k) out_.push_back(out_[out_.size() - dist]);
            }
        }
    }
    const unsigned char* in_;
    std::size_t n_, pos_ = 0;
    std::uint32_t buf_ = 0;
    int cnt_ = 0;
    std::vector<unsigned char> out_;
    Huffman lit_{}, dist_{};
};

// 8-bit non-interlaced PNG -> cv::Mat as cv::imread(IMREAD_UNCHANGED) orders it: gray CV_8UC1, colour CV_8UC3 in B, G, R
// (an alpha channel is dropped, a palette is expanded).  Returns an empty Mat when the file is missing or not such a PNG.
inline cv::Mat read_png(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return cv::Mat();
    std::vector<unsigned char> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const unsigned char magic[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (d.size() < 8 || std::memcmp(d.data(), magic, 8) != 0) return cv::Mat();
    auto be32 = [&](std::size_t p) { return (std::uint32_t)d[p] << 24 | (std::uint32_t)d[p + 1] << 16 | (std::uint32_t)d[p + 2] << 8 | d[p + 3]; };
    std::uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<unsigned char> idat, plte;
    for (std::size_t pos = 8; pos + 12 <= d.size();) {
        const std::uint32_t len = be32(pos);
        if (pos + 12 + len > d.size()) return cv::Mat();
        const std::string tag(reinterpret_cast<const char*>(&d[pos + 4]), 4);
        const unsigned char* body = &d[pos + 8];
        if (tag == "IHDR" && len == 13) { w = be32(pos + 8); h = be32(pos + 12); depth = body[8]; ctype = body[9]; interlace = body[12]; }
        else if (tag == "PLTE") plte.assign(body, body + len);
        else if (tag == "IDAT") idat.insert(idat.end(), body, body + len);
        else if (tag == "IEND") break;
        pos += 12 + len;
    }
    int ch = 0;
    switch (ctype) { case 0: ch = 1; break; case 2: ch = 3; break; case 3: ch = 1; break; case 4: ch = 2; break; case 6: ch = 4; break; default: return cv::Mat(); }
    if (depth != 8 || interlace != 0 || w == 0 || h == 0) return cv::Mat();
    std::vector<unsigned char> raw;
    try { raw = Inflate(idat.data(), idat.size()).run(); } catch (const std::runtime_error&) { return cv::Mat(); }
    const std::size_t stride = (std::size_t)w * ch;
    if (raw.size() < (stride + 1) * h) return cv::Mat();
    std::vector<unsigned char> img(stride * h), zero(stride, 0);
    for (std::uint32_t y = 0; y < h; ++y) {
        const unsigned char* line = &raw[(stride + 1) * y + 1];
        const int ft = raw[(stride + 1) * y];
        unsigned char* cur = &img[stride * y];
        const unsigned char* prev = y ? &img[stride * (y - 1)] : zero.data();
        for (std::size_t x = 0; x < stride; ++x) {
            const int a = x >= (std::size_t)ch ? cur[x - ch] : 0, b = prev[x], c = x >= (std::size_t)ch ? prev[x - ch] : 0;
            int pr = 0;
            if (ft == 1) pr = a;
            else if (ft == 2) pr = b;
            else if (ft == 3) pr = (a + b) >> 1;
            else if (ft == 4) {
                const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            } else if (ft != 0) return cv::Mat();
            cur[x] = (unsigned char)(line[x] + pr);
        }
    }
    const bool colour = ctype == 2 || ctype == 6 || ctype == 3;
    cv::Mat out((int)h, (int)w, colour ? CV_8UC3 : CV_8UC1);
    for (std::size_t p = 0; p < (std::size_t)w * h; ++p) {
        const unsigned char* s = &img[p * ch];
        if (!colour) out.data[p] = s[0];
        else if (ctype == 3) {
            const std::size_t e = (std::size_t)s[0] * 3;
            for (int k = 0; k < 3; ++k) out.data[3 * p + k] = e + 2 < plte.size() ? plte[e + 2 - k] : 0;
        } else for (int k = 0; k < 3; ++k) out.data[3 * p + k] = s[2 - k];
    }
    return out;
}

// `<index>` left-padded with zeros to `digits` characters (compose_file_name of the reference's data-set sources)
inline std::string padded_index(long index, std::size_t digits)
{
    std::ostringstream ss;
    ss << std::setw((int)digits) << std::setfill('0') << index;
    return ss.str();
}

// the files of one channel of a sequence directory: <directory><stem><index, zero padded><suffix>, walked by a cursor that
// starts one before `first` (the sources step before they read)
class IndexedFiles {
public:
    IndexedFiles() = default;
    IndexedFiles(std::string directory, std::string stem, std::size_t digits, std::string suffix, std::size_t first)
        : directory_(std::move(directory)), stem_(std::move(stem)), suffix_(std::move(suffix)), digits_(digits), first_((long)first), cursor_((long)first - 1)
    {
        if (!directory_.empty() && directory_.back() != '/') directory_ += '/';
    }
    std::string path(long index) const { return directory_ + stem_ + padded_index(index, digits_) + suffix_; }
    std::string current() const { return path(cursor_); }
    long advance() { return ++cursor_; }
    void rewind() { cursor_ = first_ - 1; }
    long cursor() const { return cursor_; }
    long first() const { return first_; }
    const std::string& directory() const { return directory_; }

private:
    std::string directory_, stem_, suffix_;
    std::size_t digits_ = 0;
    long first_ = 0, cursor_ = -1;
};

// wall-clock milliseconds a callable took (the data-loading times the filter subtracts from its execution time)
template <class F>
double milliseconds_of(F&& f)
{
    const auto started = std::chrono::steady_clock::now();
    f();
    const auto d = std::chrono::steady_clock::now() - started;
    loading_us_counter() += std::chrono::duration<double, std::micro>(d).count();
    return (double)std::chrono::duration_cast<std::chrono::milliseconds>(d).count();
}

// rows of doubles of a text file: `skip_rows` leading rows and `skip_cols` leading columns dropped, rows with fewer than
// `cols` values left ignored
inline std::vector<std::vector<double>> read_rows(const std::string& path, std::size_t skip_rows, std::size_t skip_cols, std::size_t cols)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open " + path);
    std::vector<std::vector<double>> rows;
    std::string line;
    for (std::size_t r = 0; std::getline(in, line); ++r) {
        if (r < skip_rows) continue;
        std::istringstream ls(line);
        std::vector<double> v;
        std::string tok;
        while (ls >> tok) v.push_back(std::strtod(tok.c_str(), nullptr));
        if (v.size() >= skip_cols + cols) rows.emplace_back(v.begin() + (long)skip_cols, v.begin() + (long)(skip_cols + cols));
    }
    return rows;
}

// Which stored item a rate-reduced, late source hands out at frame `head` (counted like the file indices, `first` = index of
// the first frame): only every `period`-th frame carries one; with `late` it is the item of the frame one period back, except
// before the first period has passed, when it is the first item.  -1: nothing at this frame.
// (src/roft-lib/src/DatasetImageSegmentationDelayed.cpp:42-63; the pose source of main.cpp:340-346 follows the same rule.)
inline long delayed_item(long head, long first, long period, bool late)
{
    const long item = late ? head - period : head;
    if ((item - first) % period != 0) return -1;
    return item < 0 ? first : item;
}

// x y z axis angle -> rigid transform (translation + unit quaternion w x y z)
inline Eigen::Transform<double, 3, Eigen::Affine> transform_of(const double* r)
{
    Eigen::Transform<double, 3, Eigen::Affine> t;
    for (int i = 0; i < 3; ++i) t.translation()[i] = r[i];
    const double n = std::sqrt(r[3] * r[3] + r[4] * r[4] + r[5] * r[5]);
    if (n > 0.0) {
        const double s = std::sin(r[6] / 2.0);
        t.quaternion()[0] = std::cos(r[6] / 2.0);
        for (int i = 0; i < 3; ++i) t.quaternion()[1 + i] = s * r[3 + i] / n;
    }
    return t;
}

}  // namespace compat
}  // namespace ROFT

#ifndef ROFT_HAVE_REAL_DEPENDENCIES

namespace RobotsIO {
namespace Camera {

class DatasetCamera : public Camera {
public:
    DatasetCamera(const std::string& data_path, const std::string& data_prefix, const std::string& rgb_prefix, const std::string& depth_prefix,
                  const std::string& data_format, const std::string& rgb_format, const std::string& depth_format, const std::size_t& heading_zeros,
                  const std::size_t& index_offset, const std::size_t& width, const std::size_t& height, const double& fx, const double& cx,
                  const double& fy, const double& cy)
        : root_(data_path), rgb_prefix_(rgb_prefix), depth_prefix_(depth_prefix), rgb_format_(rgb_format), depth_format_(depth_format),
          heading_zeros_(heading_zeros), index_offset_(index_offset)
    {
        if (!root_.empty() && root_.back() != '/') root_ += '/';
        parameters_.width(width); parameters_.height(height);
        parameters_.fx(fx); parameters_.fy(fy); parameters_.cx(cx); parameters_.cy(cy);
        data_ = ROFT::compat::read_rows(root_ + data_prefix + "data." + data_format, 0, 0, 9);
        if (data_.empty()) throw std::runtime_error("DatasetCamera::ctor. Error: no frames in " + root_ + data_prefix + "data." + data_format);
        reset();
    }
    // (frame indices are absolute: with index_offset = k the first frame is rgb/<k>, depth/<k> and row k of the data file,
    //  as test/test_ho3d.sh:142 starts the tracker in the middle of a sequence)
    bool status() const override { return frame_ < (long)data_.size(); }
    bool step_frame() override { ++frame_; return status(); }
    bool reset() override { frame_ = -1 + (long)index_offset_; return true; }
    std::pair<bool, CameraParameters> parameters() const override { return {true, parameters_}; }
    std::pair<bool, Eigen::MatrixXf> depth(const bool&) override
    {
        const std::string path = root_ + depth_prefix_ + ROFT::compat::padded_index(frame_, heading_zeros_) + "." + depth_format_;
        std::FILE* in = std::fopen(path.c_str(), "rb");
        if (!in) { std::cout << "DatasetCamera::depth. Error: cannot load depth frame " << path << std::endl; return {false, Eigen::MatrixXf()}; }
        std::size_t dims[2] = {0, 0};
        Eigen::MatrixXf d;
        bool ok = std::fread(dims, sizeof(dims), 1, in) == 1 && dims[0] == parameters_.width() && dims[1] == parameters_.height();
        if (ok) {
            d.resize(dims[1], dims[0]);
            ok = std::fread(d.data(), sizeof(float), dims[0] * dims[1], in) == dims[0] * dims[1];
        }
        std::fclose(in);
        if (!ok) std::cout << "DatasetCamera::depth. Error: cannot load depth data of frame " << path << std::endl;
        return {ok, std::move(d)};
    }
    std::pair<bool, cv::Mat> rgb(const bool&) override
    {
        cv::Mat m = ROFT::compat::read_png(root_ + rgb_prefix_ + ROFT::compat::padded_index(frame_, heading_zeros_) + "." + rgb_format_);
        return {!m.empty(), m};
    }
    std::pair<bool, Eigen::Transform<double, 3, Eigen::Affine>> pose(const bool&) override
    {
        if (!in_range()) return {false, {}};
        return {true, ROFT::compat::transform_of(row().data() + 2)};
    }
    std::pair<bool, double> time_stamp_rgb() const override { return {in_range(), in_range() ? row()[0] : 0.0}; }
    std::pair<bool, double> time_stamp_depth() const override { return {in_range(), in_range() ? row()[1] : 0.0}; }
    std::int32_t frame_index() const override { return (std::int32_t)frame_; }

private:
    bool in_range() const { return frame_ >= 0 && status(); }
    const std::vector<double>& row() const { return data_[(std::size_t)frame_]; }
    std::string root_, rgb_prefix_, depth_prefix_, rgb_format_, depth_format_;
    std::size_t heading_zeros_, index_offset_;
    CameraParameters parameters_;
    std::vector<std::vector<double>> data_;
    long frame_ = -1;
};

}  // namespace Camera

namespace Utils {

class DatasetTransform : public Transform {
public:
    DatasetTransform(const std::string& file_path, const std::size_t& skip_rows, const std::size_t& skip_cols, const std::size_t& expected_cols)
        : rows_(ROFT::compat::read_rows(file_path, skip_rows, skip_cols, expected_cols))
    {
        if (expected_cols != 7) throw std::runtime_error("DatasetTransform::ctor. Error: a pose row holds x y z axis angle (7 values).");
    }
    Eigen::Transform<double, 3, Eigen::Affine> transform() override { return transform_; }
    // steps to the next row; true when it holds a pose (an all-zero row is a missing detection)
    bool freeze(const bool = false) override
    {
        ++head_;
        return latch(head_);
    }
    bool transform_received() override { return received_; }

protected:
    bool latch(long index)
    {
        received_ = false;
        if (index < 0 || index >= (long)rows_.size()) return false;
        const std::vector<double>& r = rows_[(std::size_t)index];
        bool all_zero = true;
        for (double v : r) all_zero = all_zero && v == 0.0;
        if (all_zero) return false;
        transform_ = ROFT::compat::transform_of(r.data());
        received_ = true;
        return true;
    }
    std::vector<std::vector<double>> rows_;
    long head_ = -1;
    Eigen::Transform<double, 3, Eigen::Affine> transform_;
    bool received_ = false;
};

class DatasetTransformDelayed : public DatasetTransform {
public:
    DatasetTransformDelayed(const double& fps, const double& simulated_fps, const bool simulate_inference_time, const std::string& file_path,
                            const std::size_t& skip_rows, const std::size_t& skip_cols, const std::size_t& expected_cols)
        : DatasetTransform(file_path, skip_rows, skip_cols, expected_cols), delay_((int)(fps / simulated_fps)), simulate_inference_time_(simulate_inference_time)
    {
        if (delay_ < 1) throw std::runtime_error("DatasetTransformDelayed::ctor. Error: the simulated rate exceeds the rate of the data.");
    }
    // the pose computed on frame k is delivered at frame k + delay, and only every delay-th frame carries one
    bool freeze(const bool = false) override
    {
        const long item = ROFT::compat::delayed_item(++head_, 0, delay_, simulate_inference_time_);
        if (item < 0) { received_ = false; return false; }
        return latch(item);
    }
    int get_frames_between_iterations() const override { return delay_; }

private:
    const int delay_;
    const bool simulate_inference_time_;
};

}  // namespace Utils
}  // namespace RobotsIO

#endif  // ROFT_HAVE_REAL_DEPENDENCIES
