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
    VectorXd tail(std::ptrdiff_t n) const { return segment(size() - n, n); }
    void setZero() { for (auto& x : v_) x = 0.0; }
    VectorXd& operator+=(const VectorXd& o) { assert(o.size() == size()); for (std::ptrdiff_t i = 0; i < size(); i++) v_[i] += o(i); return *this; }
    VectorXd cwiseAbs() const { VectorXd r(size()); for (std::ptrdiff_t i = 0; i < size(); i++) r(i) = v_[i] < 0 ? -v_[i] : v_[i]; return r; }
    double maxCoeff() const { assert(size() > 0); double m = v_[0]; for (double x : v_) if (x > m) m = x; return m; }
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
    double* data() { return v_.data(); }
    void resize(std::ptrdiff_t r, std::ptrdiff_t c) { r_ = r; c_ = c; v_.assign(r * c, 0.0); }
    void setZero() { for (auto& x : v_) x = 0.0; }
    MatrixXd& operator+=(const MatrixXd& o) { assert(o.r_ == r_ && o.c_ == c_); for (size_t i = 0; i < v_.size(); i++) v_[i] += o.v_[i]; return *this; }
    MatrixXd middleRows(std::ptrdiff_t o, std::ptrdiff_t n) const { assert(o >= 0 && o + n <= r_); MatrixXd m(n, c_); for (std::ptrdiff_t j = 0; j < c_; j++) for (std::ptrdiff_t i = 0; i < n; i++) m(i, j) = (*this)(o + i, j); return m; }
    MatrixXd topRows(std::ptrdiff_t n) const { return middleRows(0, n); }
    MatrixXd bottomRows(std::ptrdiff_t n) const { return middleRows(r_ - n, n); }
    MatrixXd cwiseAbs() const { MatrixXd m(r_, c_); for (size_t i = 0; i < v_.size(); i++) m.v_[i] = v_[i] < 0 ? -v_[i] : v_[i]; return m; }
    double maxCoeff() const { assert(!v_.empty()); double m = v_[0]; for (double x : v_) if (x > m) m = x; return m; }
private:
    std::ptrdiff_t r_ = 0, c_ = 0;
    std::vector<double> v_;
};
inline MatrixXd operator*(double a, const MatrixXd& m) { MatrixXd r(m.rows(), m.cols()); for (std::ptrdiff_t j = 0; j < m.cols(); j++) for (std::ptrdiff_t i = 0; i < m.rows(); i++) r(i, j) = a * m(i, j); return r; }
inline MatrixXd operator-(const MatrixXd& a, const MatrixXd& b) { assert(a.rows() == b.rows() && a.cols() == b.cols()); MatrixXd r(a.rows(), a.cols()); for (std::ptrdiff_t j = 0; j < a.cols(); j++) for (std::ptrdiff_t i = 0; i < a.rows(); i++) r(i, j) = a(i, j) - b(i, j); return r; }
inline MatrixXd operator/(const MatrixXd& m, double a) { return (1.0 / a) * m; }
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
using Vector4d = FixedVector<4>;
// 3x3, row access only: what `model_.GetIr() * v.segment<3>(3)` needs (controllers/mpc_controller.cpp:258)
class Matrix3d {
public:
    Matrix3d() { for (double& x : v_) x = 0; }
    static Matrix3d Zero() { return Matrix3d(); }
    static Matrix3d Identity() { Matrix3d m; m(0, 0) = m(1, 1) = m(2, 2) = 1; return m; }
    static constexpr std::ptrdiff_t rows() { return 3; }
    static constexpr std::ptrdiff_t cols() { return 3; }
    double& operator()(std::ptrdiff_t i, std::ptrdiff_t j) { assert(i >= 0 && i < 3 && j >= 0 && j < 3); return v_[3 * j + i]; }
    double operator()(std::ptrdiff_t i, std::ptrdiff_t j) const { assert(i >= 0 && i < 3 && j >= 0 && j < 3); return v_[3 * j + i]; }
private:
    double v_[9];
};
inline Vector3d operator*(const Matrix3d& m, const Vector3d& x) { Vector3d r; for (int i = 0; i < 3; i++) r(i) = m(i, 0) * x(0) + m(i, 1) * x(1) + m(i, 2) * x(2); return r; }

}  // namespace Eigen
