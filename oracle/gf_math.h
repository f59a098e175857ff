// TEST INFRASTRUCTURE — tiny dependency-free linear algebra for the CPU oracle (Eigen/Sophus are not installed).
// Restates only what the reference's factors use: 3-vectors, 3x3, quaternions (Eigen conventions: Hamilton,
// q = (w, x,y,z), toRotationMatrix), Utility::{deltaQ,skewSymmetric,Qleft,Qright} (utility/utility.h:23-89),
// Sophus SO3 exp/log and the right Jacobians of utility/sophus_utils.hpp:154-236, small dense helpers.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>
#include <cassert>

namespace gfm {

template <int R, int C> struct Mat {
    double a[R * C];
    Mat() { for (int i = 0; i < R * C; i++) a[i] = 0; }
    double& operator()(int r, int c) { return a[r * C + c]; }
    double operator()(int r, int c) const { return a[r * C + c]; }
    double& operator[](int i) { return a[i]; }
    double operator[](int i) const { return a[i]; }
    static Mat Identity() { Mat m; for (int i = 0; i < (R < C ? R : C); i++) m(i, i) = 1; return m; }
    Mat<C, R> T() const { Mat<C, R> t; for (int r = 0; r < R; r++) for (int c = 0; c < C; c++) t(c, r) = (*this)(r, c); return t; }
    Mat operator+(const Mat& o) const { Mat m; for (int i = 0; i < R * C; i++) m.a[i] = a[i] + o.a[i]; return m; }
    Mat operator-(const Mat& o) const { Mat m; for (int i = 0; i < R * C; i++) m.a[i] = a[i] - o.a[i]; return m; }
    Mat operator-() const { Mat m; for (int i = 0; i < R * C; i++) m.a[i] = -a[i]; return m; }
    Mat operator*(double s) const { Mat m; for (int i = 0; i < R * C; i++) m.a[i] = a[i] * s; return m; }
    Mat operator/(double s) const { Mat m; for (int i = 0; i < R * C; i++) m.a[i] = a[i] / s; return m; }
    Mat& operator+=(const Mat& o) { for (int i = 0; i < R * C; i++) a[i] += o.a[i]; return *this; }
    Mat& operator-=(const Mat& o) { for (int i = 0; i < R * C; i++) a[i] -= o.a[i]; return *this; }
    double squaredNorm() const { double s = 0; for (int i = 0; i < R * C; i++) s += a[i] * a[i]; return s; }
    double norm() const { return std::sqrt(squaredNorm()); }
    double dot(const Mat& o) const { double s = 0; for (int i = 0; i < R * C; i++) s += a[i] * o.a[i]; return s; }
    template <int R2, int C2> void setBlock(int r0, int c0, const Mat<R2, C2>& b) { for (int r = 0; r < R2; r++) for (int c = 0; c < C2; c++) (*this)(r0 + r, c0 + c) = b(r, c); }
    template <int R2, int C2> Mat<R2, C2> block(int r0, int c0) const { Mat<R2, C2> b; for (int r = 0; r < R2; r++) for (int c = 0; c < C2; c++) b(r, c) = (*this)(r0 + r, c0 + c); return b; }
};
template <int R, int K, int C> Mat<R, C> operator*(const Mat<R, K>& x, const Mat<K, C>& y) {
    Mat<R, C> m;
    for (int r = 0; r < R; r++) for (int c = 0; c < C; c++) { double s = 0; for (int k = 0; k < K; k++) s += x(r, k) * y(k, c); m(r, c) = s; }
    return m;
}
template <int R, int C> Mat<R, C> operator*(double s, const Mat<R, C>& m) { return m * s; }
typedef Mat<3, 1> V3;
typedef Mat<3, 3> M3;
inline V3 v3(double x, double y, double z) { V3 v; v[0] = x; v[1] = y; v[2] = z; return v; }
inline V3 cross(const V3& a, const V3& b) { return v3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]); }
inline M3 skew(const V3& q) { M3 m; m(0, 1) = -q[2]; m(0, 2) = q[1]; m(1, 0) = q[2]; m(1, 2) = -q[0]; m(2, 0) = -q[1]; m(2, 1) = q[0]; return m; }  // utility.h:38-46
inline M3 diag3(double a, double b, double c) { M3 m; m(0, 0) = a; m(1, 1) = b; m(2, 2) = c; return m; }

struct Quat {  // Eigen::Quaterniond semantics
    double w, x, y, z;
    Quat() : w(1), x(0), y(0), z(0) {}
    Quat(double w_, double x_, double y_, double z_) : w(w_), x(x_), y(y_), z(z_) {}
    V3 vec() const { return v3(x, y, z); }
    Quat operator*(const Quat& b) const {
        return Quat(w * b.w - x * b.x - y * b.y - z * b.z, w * b.x + x * b.w + y * b.z - z * b.y, w * b.y + y * b.w + z * b.x - x * b.z,
                    w * b.z + z * b.w + x * b.y - y * b.x);
    }
    double squaredNorm() const { return w * w + x * x + y * y + z * z; }
    Quat normalized() const { double n = std::sqrt(squaredNorm()); return Quat(w / n, x / n, y / n, z / n); }
    void normalize() { *this = normalized(); }
    Quat conjugate() const { return Quat(w, -x, -y, -z); }
    Quat inverse() const {  // Eigen: conjugate / squaredNorm
        double n2 = squaredNorm();
        return Quat(w / n2, -x / n2, -y / n2, -z / n2);
    }
    M3 toRotationMatrix() const {  // Eigen/src/Geometry/Quaternion.h
        M3 r;
        const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x,
                     tyy = ty * y, tyz = tz * y, tzz = tz * z;
        r(0, 0) = 1 - (tyy + tzz); r(0, 1) = txy - twz; r(0, 2) = txz + twy;
        r(1, 0) = txy + twz; r(1, 1) = 1 - (txx + tzz); r(1, 2) = tyz - twx;
        r(2, 0) = txz - twy; r(2, 1) = tyz + twx; r(2, 2) = 1 - (txx + tyy);
        return r;
    }
    V3 operator*(const V3& v) const {  // Eigen _transformVector: v + w*uv + u x uv, uv = 2 u x v
        V3 u = vec();
        V3 uv = cross(u, v);
        uv = uv + uv;
        return v + uv * w + cross(u, uv);
    }
};
inline Quat quatFromMatrix(const M3& m) {  // Eigen quaternionbase_assign_impl<Matrix3>
    Quat q;
    double t = m(0, 0) + m(1, 1) + m(2, 2);
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q.w = 0.5 * t; t = 0.5 / t;
        q.x = (m(2, 1) - m(1, 2)) * t; q.y = (m(0, 2) - m(2, 0)) * t; q.z = (m(1, 0) - m(0, 1)) * t;
    } else {
        int i = 0;
        if (m(1, 1) > m(0, 0)) i = 1;
        if (m(2, 2) > m(i, i)) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
        double v[3];
        v[i] = 0.5 * t; t = 0.5 / t;
        q.w = (m(k, j) - m(j, k)) * t;
        v[j] = (m(j, i) + m(i, j)) * t; v[k] = (m(k, i) + m(i, k)) * t;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
    }
    return q;
}
inline Quat deltaQ(const V3& theta) {  // utility.h:23-36
    Quat dq(1.0, theta[0] / 2.0, theta[1] / 2.0, theta[2] / 2.0);
    dq.normalize();
    return dq;
}
inline Mat<4, 4> Qleft(const Quat& q) {  // utility.h:58-66 (positify is the identity, :49-56)
    Mat<4, 4> a;
    a(0, 0) = q.w; a(0, 1) = -q.x; a(0, 2) = -q.y; a(0, 3) = -q.z;
    a(1, 0) = q.x; a(2, 0) = q.y; a(3, 0) = q.z;
    M3 b = M3::Identity() * q.w + skew(q.vec());
    a.setBlock(1, 1, b);
    return a;
}
inline Mat<4, 4> Qright(const Quat& p) {  // utility.h:68-76
    Mat<4, 4> a;
    a(0, 0) = p.w; a(0, 1) = -p.x; a(0, 2) = -p.y; a(0, 3) = -p.z;
    a(1, 0) = p.x; a(2, 0) = p.y; a(3, 0) = p.z;
    M3 b = M3::Identity() * p.w - skew(p.vec());
    a.setBlock(1, 1, b);
    return a;
}
// Sophus::SO3d::exp / log (sophus/so3.hpp), Constants<double>::epsilon() = 1e-10
inline Quat so3_exp(const V3& omega) {
    const double theta_sq = omega.squaredNorm();
    double imag, real;
    if (theta_sq < 1e-10 * 1e-10) {
        const double theta_po4 = theta_sq * theta_sq;
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
        real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_po4;
    } else {
        const double theta = std::sqrt(theta_sq), half = 0.5 * theta;
        imag = std::sin(half) / theta;
        real = std::cos(half);
    }
    return Quat(real, imag * omega[0], imag * omega[1], imag * omega[2]);
}
inline V3 so3_log(const Quat& qin) {
    Quat q = qin.normalized();  // SO3d(quaternion) normalises
    const double squared_n = q.x * q.x + q.y * q.y + q.z * q.z, w = q.w;
    double two_atan_nbyw_by_n;
    if (squared_n < 1e-10 * 1e-10) {
        const double squared_w = w * w;
        two_atan_nbyw_by_n = 2.0 / w - (2.0 / 3.0) * squared_n / (w * squared_w);
    } else {
        const double n = std::sqrt(squared_n);
        if (std::abs(w) < 1e-10) two_atan_nbyw_by_n = (w > 0 ? M_PI : -M_PI) / n;
        else two_atan_nbyw_by_n = 2.0 * std::atan(n / w) / n;
    }
    return v3(two_atan_nbyw_by_n * q.x, two_atan_nbyw_by_n * q.y, two_atan_nbyw_by_n * q.z);
}
inline M3 rightJacobianSO3(const V3& phi) {  // sophus_utils.hpp:154-184
    const double n2 = phi.squaredNorm();
    M3 h = skew(phi), h2 = h * h, J = M3::Identity();
    if (n2 > 1e-10) {
        const double n = std::sqrt(n2), n3 = n2 * n;
        J -= h * (1 - std::cos(n)) / n2;
        J += h2 * (n - std::sin(n)) / n3;
    } else { J -= h / 2; J += h2 / 6; }
    return J;
}
inline M3 rightJacobianInvSO3(const V3& phi) {  // sophus_utils.hpp:194-236
    const double n2 = phi.squaredNorm();
    M3 h = skew(phi), h2 = h * h, J = M3::Identity();
    J += h / 2;
    if (n2 > 1e-10) {
        const double n = std::sqrt(n2);
        if (n < M_PI - 1e-5) J += h2 * (1 / n2 - (1 + std::cos(n)) / (2 * n * std::sin(n)));
        else J += h2 / (M_PI * M_PI);
    } else J += h2 / 12;
    return J;
}

// ---- dynamic dense (row-major) helpers
struct DMat {
    int r = 0, c = 0;
    std::vector<double> a;
    DMat() {}
    DMat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
    double& operator()(int i, int j) { return a[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
};
// in-place lower Cholesky of the leading n x n of A (row-major, stride lda); returns false if not positive definite
inline bool cholesky_lower(double* A, int n, int lda) {
    for (int j = 0; j < n; j++) {
        double d = A[(size_t)j * lda + j];
        for (int k = 0; k < j; k++) d -= A[(size_t)j * lda + k] * A[(size_t)j * lda + k];
        if (!(d > 0.0)) return false;
        d = std::sqrt(d);
        A[(size_t)j * lda + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[(size_t)i * lda + j];
            for (int k = 0; k < j; k++) s -= A[(size_t)i * lda + k] * A[(size_t)j * lda + k];
            A[(size_t)i * lda + j] = s / d;
        }
    }
    return true;
}
inline void cholesky_solve(const double* L, int n, int lda, double* b) {
    for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[(size_t)i * lda + k] * b[k]; b[i] = s / L[(size_t)i * lda + i]; }
    for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < n; k++) s -= L[(size_t)k * lda + i] * b[k]; b[i] = s / L[(size_t)i * lda + i]; }
}
// general inverse by partial-pivot LU (Eigen's inverse() for fixed sizes > 4 uses PartialPivLU)
template <int N> Mat<N, N> inverse(const Mat<N, N>& M) {
    double a[N][2 * N];
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) { a[i][j] = M(i, j); a[i][N + j] = i == j ? 1.0 : 0.0; }
    for (int col = 0; col < N; col++) {
        int piv = col;
        for (int i = col + 1; i < N; i++) if (std::abs(a[i][col]) > std::abs(a[piv][col])) piv = i;
        if (piv != col) for (int j = 0; j < 2 * N; j++) std::swap(a[col][j], a[piv][j]);
        const double d = a[col][col];
        for (int i = col + 1; i < N; i++) {
            const double f = a[i][col] / d;
            if (f != 0) for (int j = col; j < 2 * N; j++) a[i][j] -= f * a[col][j];
        }
    }
    for (int col = N - 1; col >= 0; col--) {
        const double d = a[col][col];
        for (int j = 0; j < 2 * N; j++) a[col][j] /= d;
        for (int i = 0; i < col; i++) { const double f = a[i][col]; if (f != 0) for (int j = 0; j < 2 * N; j++) a[i][j] -= f * a[col][j]; }
    }
    Mat<N, N> R;
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) R(i, j) = a[i][N + j];
    return R;
}
// sqrt_info = LLT(M).matrixL().transpose()  (imu_factor.h:73, wheel_factor.h:85)
template <int N> Mat<N, N> llt_upper(const Mat<N, N>& M) {
    Mat<N, N> L = M;
    cholesky_lower(L.a, N, N);
    Mat<N, N> U;
    for (int i = 0; i < N; i++) for (int j = i; j < N; j++) U(i, j) = L(j, i);
    return U;
}

// symmetric eigen-decomposition (Householder tridiagonalisation + implicit QL; the classic tred2/tql2 pair, the same
// family of algorithm as Eigen::SelfAdjointEigenSolver).  V columns = eigenvectors, d ascending.
inline void sym_eig(int n, const double* Ain, double* d, double* V) {
    std::vector<double> e(n);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) V[(size_t)i * n + j] = Ain[(size_t)i * n + j];
    auto v = [&](int i, int j) -> double& { return V[(size_t)i * n + j]; };
    for (int j = 0; j < n; j++) d[j] = v(n - 1, j);
    for (int i = n - 1; i > 0; i--) {
        double scale = 0.0, h = 0.0;
        for (int k = 0; k < i; k++) scale += std::abs(d[k]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (int j = 0; j < i; j++) { d[j] = v(i - 1, j); v(i, j) = 0.0; v(j, i) = 0.0; }
        } else {
            for (int k = 0; k < i; k++) { d[k] /= scale; h += d[k] * d[k]; }
            double f = d[i - 1], g = std::sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g; h -= f * g; d[i - 1] = f - g;
            for (int j = 0; j < i; j++) e[j] = 0.0;
            for (int j = 0; j < i; j++) {
                f = d[j]; v(j, i) = f; g = e[j] + v(j, j) * f;
                for (int k = j + 1; k <= i - 1; k++) { g += v(k, j) * d[k]; e[k] += v(k, j) * f; }
                e[j] = g;
            }
            f = 0.0;
            for (int j = 0; j < i; j++) { e[j] /= h; f += e[j] * d[j]; }
            const double hh = f / (h + h);
            for (int j = 0; j < i; j++) e[j] -= hh * d[j];
            for (int j = 0; j < i; j++) {
                f = d[j]; g = e[j];
                for (int k = j; k <= i - 1; k++) v(k, j) -= (f * e[k] + g * d[k]);
                d[j] = v(i - 1, j); v(i, j) = 0.0;
            }
        }
        d[i] = h;
    }
    for (int i = 0; i < n - 1; i++) {
        v(n - 1, i) = v(i, i); v(i, i) = 1.0;
        const double h = d[i + 1];
        if (h != 0.0) {
            for (int k = 0; k <= i; k++) d[k] = v(k, i + 1) / h;
            for (int j = 0; j <= i; j++) {
                double g = 0.0;
                for (int k = 0; k <= i; k++) g += v(k, i + 1) * v(k, j);
                for (int k = 0; k <= i; k++) v(k, j) -= g * d[k];
            }
        }
        for (int k = 0; k <= i; k++) v(k, i + 1) = 0.0;
    }
    for (int j = 0; j < n; j++) { d[j] = v(n - 1, j); v(n - 1, j) = 0.0; }
    v(n - 1, n - 1) = 1.0; e[0] = 0.0;
    // tql2
    for (int i = 1; i < n; i++) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    double f = 0.0, tst1 = 0.0;
    const double eps = std::pow(2.0, -52.0);
    for (int l = 0; l < n; l++) {
        tst1 = std::max(tst1, std::abs(d[l]) + std::abs(e[l]));
        int m = l;
        while (m < n) { if (std::abs(e[m]) <= eps * tst1) break; m++; }
        if (m > l) {
            int iter = 0;
            do {
                iter++;
                double g = d[l], p = (d[l + 1] - g) / (2.0 * e[l]), r = std::hypot(p, 1.0);
                if (p < 0) r = -r;
                d[l] = e[l] / (p + r); d[l + 1] = e[l] * (p + r);
                const double dl1 = d[l + 1];
                double h = g - d[l];
                for (int i = l + 2; i < n; i++) d[i] -= h;
                f += h;
                p = d[m];
                double c = 1.0, c2 = c, c3 = c, el1 = e[l + 1], s = 0.0, s2 = 0.0;
                for (int i = m - 1; i >= l; i--) {
                    c3 = c2; c2 = c; s2 = s;
                    g = c * e[i]; h = c * p; r = std::hypot(p, e[i]);
                    e[i + 1] = s * r; s = e[i] / r; c = p / r; p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    for (int k = 0; k < n; k++) { h = v(k, i + 1); v(k, i + 1) = s * v(k, i) + c * h; v(k, i) = c * v(k, i) - s * h; }
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p; d[l] = c * p;
            } while (std::abs(e[l]) > eps * tst1 && iter < 200);
        }
        d[l] = d[l] + f; e[l] = 0.0;
    }
    for (int i = 0; i < n - 1; i++) {  // ascending order
        int k = i; double p = d[i];
        for (int j = i + 1; j < n; j++) if (d[j] < p) { k = j; p = d[j]; }
        if (k != i) { d[k] = d[i]; d[i] = p; for (int j = 0; j < n; j++) std::swap(v(j, i), v(j, k)); }
    }
}

}  // namespace gfm
