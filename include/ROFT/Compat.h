// Compat.h -- minimal stand-ins for the bfl / Eigen types the reference's filter classes mention
// (bfl::Gaussian, Eigen::MatrixXd / VectorXd), so that the facade classes in this directory can keep
// the reference's class names and method signatures without Eigen or BayesFilters being installed.
// A maintainer integrating into the real ROFT tree would drop this file and use the real types: the
// facades only need `.data()`, `.rows()`, `.cols()` of row-major double storage.
#pragma once

#include <cstddef>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace ROFT {
namespace compat {

// dense row-major double matrix
class MatrixXd {
public:
    MatrixXd() = default;
    MatrixXd(std::size_t r, std::size_t c) : r_(r), c_(c), d_(r * c, 0.0) {}
    static MatrixXd Zero(std::size_t r, std::size_t c) { return MatrixXd(r, c); }
    static MatrixXd Identity(std::size_t n)
    {
        MatrixXd m(n, n);
        for (std::size_t i = 0; i < n; ++i) m(i, i) = 1.0;
        return m;
    }
    void resize(std::size_t r, std::size_t c) { r_ = r; c_ = c; d_.assign(r * c, 0.0); }
    std::size_t rows() const { return r_; }
    std::size_t cols() const { return c_; }
    std::size_t size() const { return d_.size(); }
    double& operator()(std::size_t i, std::size_t j = 0) { return d_[i * c_ + j]; }
    double operator()(std::size_t i, std::size_t j = 0) const { return d_[i * c_ + j]; }
    double* data() { return d_.data(); }
    const double* data() const { return d_.data(); }

private:
    std::size_t r_ = 0, c_ = 0;
    std::vector<double> d_;
};
using VectorXd = MatrixXd;  // n x 1

// bfl::Gaussian(dim_linear, dim_circular, use_quaternion): mean size lin + 4 circ, covariance lin + 3 circ
// (the sizes ROFTFilter relies on: Gaussian(9, 1, true) -> 13 / 12x12, Gaussian(6, 0, false) -> 6 / 6x6,
// src/roft-lib/src/ROFTFilter.cpp:64-67)
class Gaussian {
public:
    Gaussian() : Gaussian(1, 0, false) {}
    Gaussian(std::size_t dim_linear, std::size_t dim_circular = 0, bool use_quaternion = false)
        : dim_linear(dim_linear), dim_circular(dim_circular), use_quaternion(use_quaternion),
          mean_(dim_linear + (use_quaternion ? 4 : 1) * dim_circular, 1),
          cov_(dim_linear + (use_quaternion ? 3 : 1) * dim_circular, dim_linear + (use_quaternion ? 3 : 1) * dim_circular)
    {}
    VectorXd& mean() { return mean_; }
    const VectorXd& mean() const { return mean_; }
    double& mean(std::size_t i) { return mean_(i); }
    MatrixXd& covariance() { return cov_; }
    const MatrixXd& covariance() const { return cov_; }
    std::size_t dim_linear, dim_circular;
    bool use_quaternion;

private:
    VectorXd mean_;
    MatrixXd cov_;
};
using GaussianMixture = Gaussian;  // the reference only ever uses one component (SKFCorrection.cpp:39)

// thrown where the reference throws std::runtime_error; carries roft_last_error_string()
inline void throw_if(int rc, const char* what);

}  // namespace compat
}  // namespace ROFT

extern "C" const char* roft_last_error_string(void);

inline void ROFT::compat::throw_if(int rc, const char* what)
{
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + roft_last_error_string());
}
