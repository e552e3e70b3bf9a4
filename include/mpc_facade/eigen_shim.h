// Minimal stand-in for the handful of Eigen types the reference's public MPC interface is typed with
// (/root/reference/mpc/include/mpc.h:28-29, trajectory.h:16-18).  Used ONLY where <Eigen/Core> does not exist (this
// container has no Eigen): mpc_facade/mpc.h includes the real Eigen when it is installed, and the facade
// then carries the reference's exact signatures.  Column-major dense storage, double only, just the operations the
// caller's code uses on these objects (controllers/mpc_controller.cpp:57-67: `1*Q`, `-1*Q*des_alg`, element access).
#pragma once
#include <cassert>
#include <cstddef>
#include <initializer_list>
#include <ostream>
#include <vector>

namespace Eigen {

class VectorXd {
public:
    VectorXd() = default;
    explicit VectorXd(std::ptrdiff_t n) : v_(n, 0.0) {}
    static VectorXd Zero(std::ptrdiff_t n) { return VectorXd(n); }
    static VectorXd Constant(std::ptrdiff_t n, double c) { VectorXd r(n); for (auto& x : r.v_) x = c; return r; }
    std::ptrdiff_t size() const { return (std::ptrdiff_t)v_.size(); }
    std::ptrdiff_t rows() const { return size(); }
    void resize(std::ptrdiff_t n) { v_.assign(n, 0.0); }
    double& operator()(std::ptrdiff_t i) { assert(i >= 0 && i < size()); return v_[i]; }
    double operator()(std::ptrdiff_t i) const { assert(i >= 0 && i < size()); return v_[i]; }
    double& operator[](std::ptrdiff_t i) { return (*this)(i); }
    double operator[](std::ptrdiff_t i) const { return (*this)(i); }
    double* data() { return v_.data(); }
    const double* data() const { return v_.data(); }
    VectorXd head(std::ptrdiff_t n) const { VectorXd r(n); for (std::ptrdiff_t i = 0; i < n; i++) r(i) = v_[i]; return r; }
    VectorXd segment(std::ptrdiff_t o, std::ptrdiff_t n) const { VectorXd r(n); for (std::ptrdiff_t i = 0; i < n; i++) r(i) = v_[o + i]; return r; }
    double dot(const VectorXd& o) const { double a = 0; for (std::ptrdiff_t i = 0; i < size(); i++) a += v_[i] * o(i); return a; }
    const VectorXd& transpose() const { return *this; }        // (printing only)
private:
    std::vector<double> v_;
};
inline VectorXd operator*(double a, const VectorXd& x) { VectorXd r(x.size()); for (std::ptrdiff_t i = 0; i < x.size(); i++) r(i) = a * x(i); return r; }
inline VectorXd operator*(const VectorXd& x, double a) { return a * x; }
inline VectorXd operator+(const VectorXd& a, const VectorXd& b) { VectorXd r(a.size()); for (std::ptrdiff_t i = 0; i < a.size(); i++) r(i) = a(i) + b(i); return r; }
inline VectorXd operator-(const VectorXd& a, const VectorXd& b) { VectorXd r(a.size()); for (std::ptrdiff_t i = 0; i < a.size(); i++) r(i) = a(i) - b(i); return r; }
inline std::ostream& operator<<(std::ostream& os, const VectorXd& x) { for (std::ptrdiff_t i = 0; i < x.size(); i++) os << x(i) << (i + 1 < x.size() ? " " : ""); return os; }

class MatrixXd {
public:
    MatrixXd() = default;
    MatrixXd(std::ptrdiff_t r, std::ptrdiff_t c) : r_(r), c_(c), v_(r * c, 0.0) {}
    static MatrixXd Zero(std::ptrdiff_t r, std::ptrdiff_t c) { return MatrixXd(r, c); }
    static MatrixXd Identity(std::ptrdiff_t r, std::ptrdiff_t c) { MatrixXd m(r, c); for (std::ptrdiff_t i = 0; i < r && i < c; i++) m(i, i) = 1; return m; }
    std::ptrdiff_t rows() const { return r_; }
    std::ptrdiff_t cols() const { return c_; }
    double& operator()(std::ptrdiff_t i, std::ptrdiff_t j) { assert(i >= 0 && i < r_ && j >= 0 && j < c_); return v_[j * r_ + i]; }
    double operator()(std::ptrdiff_t i, std::ptrdiff_t j) const { assert(i >= 0 && i < r_ && j >= 0 && j < c_); return v_[j * r_ + i]; }
    const double* data() const { return v_.data(); }
private:
    std::ptrdiff_t r_ = 0, c_ = 0;
    std::vector<double> v_;
};
inline MatrixXd operator*(double a, const MatrixXd& m) { MatrixXd r(m.rows(), m.cols()); for (std::ptrdiff_t j = 0; j < m.cols(); j++) for (std::ptrdiff_t i = 0; i < m.rows(); i++) r(i, j) = a * m(i, j); return r; }
inline VectorXd operator*(const MatrixXd& m, const VectorXd& x) { VectorXd r(m.rows()); for (std::ptrdiff_t i = 0; i < m.rows(); i++) { double a = 0; for (std::ptrdiff_t j = 0; j < m.cols(); j++) a += m(i, j) * x(j); r(i) = a; } return r; }

template <int N>
class FixedVector {
public:
    FixedVector() { for (double& x : v_) x = 0; }
    FixedVector(std::initializer_list<double> l) { int i = 0; for (double x : l) if (i < N) v_[i++] = x; for (; i < N; i++) v_[i] = 0; }
    static FixedVector Zero() { return FixedVector(); }
    static constexpr std::ptrdiff_t size() { return N; }
    double& operator()(std::ptrdiff_t i) { assert(i >= 0 && i < N); return v_[i]; }
    double operator()(std::ptrdiff_t i) const { assert(i >= 0 && i < N); return v_[i]; }
    double& operator[](std::ptrdiff_t i) { return (*this)(i); }
    double operator[](std::ptrdiff_t i) const { return (*this)(i); }
    double* data() { return v_; }
    const double* data() const { return v_; }
    const FixedVector& transpose() const { return *this; }
private:
    double v_[N];
};
template <int N>
inline std::ostream& operator<<(std::ostream& os, const FixedVector<N>& x) { for (int i = 0; i < N; i++) os << x(i) << (i + 1 < N ? " " : ""); return os; }
using Vector2d = FixedVector<2>;
using Vector3d = FixedVector<3>;

}  // namespace Eigen
