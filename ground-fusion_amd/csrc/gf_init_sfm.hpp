// Initialisation while moving: the host-side numerics Estimator::initialStructure needs once the stationary / wheel-activated
// shortcuts have not fired (estimator.cpp:1684-1847).  Runs once per sequence, on the host, like the reference's.
//   solve_relative_rt_pnp   MotionEstimator::solveRelativeRT_PNP, initial/solve_5pts.cpp:244-277 (cv::solvePnPRansac)
//   construct_with_depth    GlobalSFM::constructWithDepth, initial/initial_sfm.cpp:379-594 (cv::solvePnP with a guess, ceres::Solve)
//   solve_pnp_iterative     cv::solvePnP(..., SOLVEPNP_ITERATIVE) of estimator.cpp:1796
//   linear_alignment        LinearAlignmentWithWD / LinearAlignmentWithDepth + RefineGravityWithWD / RefineGravityWithDepth, initial/initial_aligment.cpp:427-638
// The OpenCV 4.2 and Ceres 1.14 pieces are not in the reference tree; they are written from the published algorithms: RANSAC as
// RANSACPointSetRegistrator runs it (cv::RNG multiply-with-carry, 5-point subsets, iteration count from the inlier ratio), EPnP (Lepetit, Moreno-Noguer,
// Fua 2009) as its kernel, the DLT start and the Levenberg-Marquardt refinement of cvFindExtrinsicCameraParams2 with CvLevMarq's lambda schedule, and
// Ceres' trust-region loop with the Levenberg-Marquardt strategy and a Schur complement over the points.  Choices OpenCV leaves to rounding (basis inside a
// null space, eigenvector signs) are made canonically; for planar point sets the homography start is the normalised DLT without OpenCV's refinement of H.  DESIGN.md 7.
#pragma once
#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <utility>
#include <vector>

namespace gfinit {

struct Mat {
    int r = 0, c = 0;
    std::vector<double> a;
    Mat() {}
    Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
    double& operator()(int i, int j) { return a[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
};
inline Mat mul(const Mat& A, const Mat& B) {
    Mat C(A.r, B.c);
    for (int i = 0; i < A.r; i++)
        for (int k = 0; k < A.c; k++) { const double v = A(i, k); if (v == 0.0) continue; for (int j = 0; j < B.c; j++) C(i, j) += v * B(k, j); }
    return C;
}
inline Mat AtA(const Mat& A) {
    Mat C(A.c, A.c);
    for (int k = 0; k < A.r; k++)
        for (int i = 0; i < A.c; i++) { const double v = A(k, i); if (v == 0.0) continue; for (int j = 0; j < A.c; j++) C(i, j) += v * A(k, j); }
    return C;
}
inline std::vector<double> Atb(const Mat& A, const std::vector<double>& b) {
    std::vector<double> o(A.c, 0.0);
    for (int k = 0; k < A.r; k++) for (int i = 0; i < A.c; i++) o[i] += A(k, i) * b[k];
    return o;
}

// symmetric eigen-decomposition, cyclic Jacobi; eigenvalues ascending, eigenvectors as columns of V, each with its largest entry positive
inline void sym_eig(Mat A, std::vector<double>& w, Mat& V) {
    const int n = A.r;
    V = Mat(n, n);
    for (int i = 0; i < n; i++) V(i, i) = 1.0;
    for (int sweep = 0; sweep < 100; sweep++) {
        double off = 0, dia = 0;
        for (int i = 0; i < n; i++) { dia += A(i, i) * A(i, i); for (int j = i + 1; j < n; j++) off += A(i, j) * A(i, j); }
        if (off <= 1e-60 * dia || off == 0.0) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                const double apq = A(p, q);
                if (apq == 0.0) continue;
                const double theta = (A(q, q) - A(p, p)) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(1.0 + theta * theta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int k = 0; k < n; k++) { const double x = A(k, p), y = A(k, q); A(k, p) = c * x - s * y; A(k, q) = s * x + c * y; }
                for (int k = 0; k < n; k++) { const double x = A(p, k), y = A(q, k); A(p, k) = c * x - s * y; A(q, k) = s * x + c * y; }
                for (int k = 0; k < n; k++) { const double x = V(k, p), y = V(k, q); V(k, p) = c * x - s * y; V(k, q) = s * x + c * y; }
            }
    }
    std::vector<int> ord(n);
    for (int i = 0; i < n; i++) ord[i] = i;
    std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return A(x, x) < A(y, y); });
    Mat Vs(n, n);
    w.assign(n, 0.0);
    for (int j = 0; j < n; j++) {
        w[j] = A(ord[j], ord[j]);
        int kb = 0;
        for (int k = 1; k < n; k++) if (fabs(V(k, ord[j])) > fabs(V(kb, ord[j]))) kb = k;
        const double sg = V(kb, ord[j]) < 0 ? -1.0 : 1.0;
        for (int k = 0; k < n; k++) Vs(k, j) = sg * V(k, ord[j]);
    }
    V = Vs;
}

// orthonormal basis of span(first k columns of B) that depends on the subspace only: Gram-Schmidt of the projections of e_0, e_1, ...
inline bool canonical_subspace_basis(const Mat& B, int k, Mat& out) {
    const int n = B.r;
    out = Mat(n, k);
    int got = 0;
    for (int j = 0; j < n && got < k; j++) {
        std::vector<double> v(n, 0.0);
        for (int i = 0; i < n; i++) for (int m = 0; m < k; m++) v[i] += B(i, m) * B(j, m);
        for (int u = 0; u < got; u++) {
            double d = 0;
            for (int i = 0; i < n; i++) d += out(i, u) * v[i];
            for (int i = 0; i < n; i++) v[i] -= out(i, u) * d;
        }
        double nv = 0;
        for (int i = 0; i < n; i++) nv += v[i] * v[i];
        nv = sqrt(nv);
        if (nv > 0.1) { for (int i = 0; i < n; i++) out(i, got) = v[i] / nv; got++; }
    }
    return got == k;
}

// Gaussian elimination with partial pivoting: A x = b (A n x n, row-major, destroyed); false when singular
inline bool lu_solve(std::vector<double> A, std::vector<double> b, int n, std::vector<double>& x) {
    for (int k = 0; k < n; k++) {
        int p = k;
        for (int i = k + 1; i < n; i++) if (fabs(A[(size_t)i * n + k]) > fabs(A[(size_t)p * n + k])) p = i;
        if (A[(size_t)p * n + k] == 0.0 || !std::isfinite(A[(size_t)p * n + k])) return false;
        if (p != k) { for (int j = 0; j < n; j++) std::swap(A[(size_t)k * n + j], A[(size_t)p * n + j]); std::swap(b[k], b[p]); }
        const double inv = 1.0 / A[(size_t)k * n + k];
        for (int i = k + 1; i < n; i++) {
            const double f = A[(size_t)i * n + k] * inv;
            if (f == 0.0) continue;
            for (int j = k; j < n; j++) A[(size_t)i * n + j] -= f * A[(size_t)k * n + j];
            b[i] -= f * b[k];
        }
    }
    x.assign(n, 0.0);
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int j = i + 1; j < n; j++) s -= A[(size_t)i * n + j] * x[j];
        x[i] = s / A[(size_t)i * n + i];
    }
    for (int i = 0; i < n; i++) if (!std::isfinite(x[i])) return false;
    return true;
}

// least squares by Householder QR without pivoting (epnp.cpp qr_solve); false when a diagonal entry of R vanishes
inline bool qr_lsq(Mat A, std::vector<double> b, std::vector<double>& x) {
    const int m = A.r, n = A.c;
    for (int k = 0; k < n; k++) {
        double nrm = 0;
        for (int i = k; i < m; i++) nrm += A(i, k) * A(i, k);
        nrm = sqrt(nrm);
        if (!(nrm > 1e-300)) return false;
        const double alpha = A(k, k) > 0 ? -nrm : nrm;
        std::vector<double> v(m, 0.0);
        for (int i = k; i < m; i++) v[i] = A(i, k);
        v[k] -= alpha;
        double vn = 0;
        for (int i = k; i < m; i++) vn += v[i] * v[i];
        if (vn > 0) {
            for (int j = k; j < n; j++) {
                double d = 0;
                for (int i = k; i < m; i++) d += v[i] * A(i, j);
                d = 2.0 * d / vn;
                for (int i = k; i < m; i++) A(i, j) -= d * v[i];
            }
            double d = 0;
            for (int i = k; i < m; i++) d += v[i] * b[i];
            d = 2.0 * d / vn;
            for (int i = k; i < m; i++) b[i] -= d * v[i];
        }
    }
    x.assign(n, 0.0);
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int j = i + 1; j < n; j++) s -= A(i, j) * x[j];
        if (fabs(A(i, i)) < 1e-300) return false;
        x[i] = s / A(i, i);
    }
    return true;
}

// ---------------------------------------------------------------- 3-vectors / 3 x 3 matrices on plain arrays (row-major)
inline void m3mul(const double* A, const double* B, double* C) {
    double t[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    for (int i = 0; i < 9; i++) C[i] = t[i];
}
inline void m3T(const double* A, double* B) { double t[9]; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * j + i]; for (int i = 0; i < 9; i++) B[i] = t[i]; }
inline void m3v(const double* A, const double* v, double* o) { double t[3]; for (int i = 0; i < 3; i++) t[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2]; for (int i = 0; i < 3; i++) o[i] = t[i]; }
inline void m3Tv(const double* A, const double* v, double* o) { double t[3]; for (int i = 0; i < 3; i++) t[i] = A[i] * v[0] + A[3 + i] * v[1] + A[6 + i] * v[2]; for (int i = 0; i < 3; i++) o[i] = t[i]; }
inline double m3det(const double* A) { return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]); }
inline void skew(const double* v, double* S) { S[0] = 0; S[1] = -v[2]; S[2] = v[1]; S[3] = v[2]; S[4] = 0; S[5] = -v[0]; S[6] = -v[1]; S[7] = v[0]; S[8] = 0; }
inline void cross3(const double* a, const double* b, double* o) { const double t[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}; o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; }

// U V^T of the SVD of M (orthogonal polar factor) through the eigen-decomposition of M^T M
inline bool polar_rotation(const double* M, double* R) {
    Mat A(3, 3);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) A(i, j) += M[3 * k + i] * M[3 * k + j];
    std::vector<double> w; Mat V;
    sym_eig(A, w, V);
    if (!(w[0] > 0)) return false;
    double S[9] = {0};
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) S[3 * i + j] += V(i, k) * (1.0 / sqrt(w[k])) * V(j, k);
    m3mul(M, S, R);
    return true;
}

// U V^T of the SVD M = U S V^T by one-sided (Hestenes) Jacobi rotations of the columns: small singular values come out with their full relative accuracy, which
// the route through the eigen-decomposition of M^T M above does not give (its error grows with the SQUARE of the condition number -- EPnP's 3 x 3 correlation
// matrix of five nearly coplanar points has one: scripts/epnp_mpmath_check.py measured 1e-8 ... 4e-2 against a 60-digit evaluation before this was used there).
inline bool svd_rotation(const double* M, double* R) {
    double U[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 9; i++) U[i] = M[i];
    for (int sweep = 0; sweep < 60; sweep++) {
        bool rotated = false;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double al = 0, be = 0, ga = 0;
                for (int k = 0; k < 3; k++) { al += U[3 * k + p] * U[3 * k + p]; be += U[3 * k + q] * U[3 * k + q]; ga += U[3 * k + p] * U[3 * k + q]; }
                if (ga == 0.0 || fabs(ga) <= 1e-17 * sqrt(al * be)) continue;
                rotated = true;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int k = 0; k < 3; k++) {
                    const double x = U[3 * k + p], y = U[3 * k + q]; U[3 * k + p] = c * x - sn * y; U[3 * k + q] = sn * x + c * y;
                    const double vx = V[3 * k + p], vy = V[3 * k + q]; V[3 * k + p] = c * vx - sn * vy; V[3 * k + q] = sn * vx + c * vy;
                }
            }
        if (!rotated) break;
    }
    double sg[3], smax = 0;
    for (int j = 0; j < 3; j++) { sg[j] = sqrt(U[j] * U[j] + U[3 + j] * U[3 + j] + U[6 + j] * U[6 + j]); smax = std::max(smax, sg[j]); }
    if (!(smax > 0) || !std::isfinite(smax)) return false;
    int dead = -1, ndead = 0;
    for (int j = 0; j < 3; j++) { if (sg[j] > 1e-290 && sg[j] > 1e-200 * smax) for (int k = 0; k < 3; k++) U[3 * k + j] /= sg[j]; else { dead = j; ndead++; } }
    if (ndead > 1) return false;
    if (ndead == 1) {   // an exactly singular matrix: the missing left vector completes the other two (its sign is settled by the caller's determinant test)
        const int a = (dead + 1) % 3, b = (dead + 2) % 3;
        const double ua[3] = {U[a], U[3 + a], U[6 + a]}, ub[3] = {U[b], U[3 + b], U[6 + b]};
        U[dead] = ua[1] * ub[2] - ua[2] * ub[1]; U[3 + dead] = ua[2] * ub[0] - ua[0] * ub[2]; U[6 + dead] = ua[0] * ub[1] - ua[1] * ub[0];
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double v = 0; for (int k = 0; k < 3; k++) v += U[3 * i + k] * V[3 * j + k]; R[3 * i + j] = v; }
    return true;
}

// ---------------------------------------------------------------- cv::Rodrigues
inline void rodrigues(const double* r, double* R) {
    const double th = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (th < DBL_EPSILON) { for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0); return; }
    const double k[3] = {r[0] / th, r[1] / th, r[2] / th}, c = cos(th), s = sin(th);
    double K[9]; skew(k, K);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[3 * i + j] = c * (i == j) + (1 - c) * k[i] * k[j] + s * K[3 * i + j];
}
inline void rodrigues_inv(const double* R, double* r) {   // calibration.cpp cvRodrigues2, matrix branch (the input is a rotation)
    r[0] = R[7] - R[5]; r[1] = R[2] - R[6]; r[2] = R[3] - R[1];
    const double s = sqrt((r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * 0.25);
    const double c = std::min(1.0, std::max(-1.0, (R[0] + R[4] + R[8] - 1) * 0.5)), th = acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t[3] = {sqrt(std::max((R[0] + 1) * 0.5, 0.0)), sqrt(std::max((R[4] + 1) * 0.5, 0.0)), sqrt(std::max((R[8] + 1) * 0.5, 0.0))};
        if (R[1] < 0) t[1] = -t[1];
        if (R[2] < 0) t[2] = -t[2];
        if (fabs(t[0]) < fabs(t[1]) && fabs(t[0]) < fabs(t[2]) && (R[5] > 0) != (t[1] * t[2] > 0)) t[2] = -t[2];
        const double f = th / sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
        for (int i = 0; i < 3; i++) r[i] = t[i] * f;
        return;
    }
    const double f = th / (2 * s);
    for (int i = 0; i < 3; i++) r[i] *= f;
}
inline void d_rodrigues(const double* r, double dR[3][9]) {   // exact derivative of the exponential map (Gallego & Yezzi's compact form)
    const double th2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    if (th2 < 1e-20) { for (int k = 0; k < 3; k++) { const double e[3] = {k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, k == 2 ? 1.0 : 0.0}; skew(e, dR[k]); } return; }
    double R[9], S[9];
    rodrigues(r, R); skew(r, S);
    for (int k = 0; k < 3; k++) {
        const double col[3] = {(k == 0) - R[k], (k == 1) - R[3 + k], (k == 2) - R[6 + k]};   // column k of I - R
        double cr[3], Sc[9], M[9];
        cross3(r, col, cr); skew(cr, Sc);
        for (int i = 0; i < 9; i++) M[i] = r[k] * S[i] + Sc[i];
        m3mul(M, R, dR[k]);
        for (int i = 0; i < 9; i++) dR[k][i] /= th2;
    }
}

// ---------------------------------------------------------------- cv::RNG (core/operations.hpp: multiply-with-carry)
struct CvRNG {
    uint64_t state = 0xFFFFFFFFFFFFFFFFull;
    unsigned next() { state = (uint64_t)(unsigned)state * 4164903690u + (unsigned)(state >> 32); return (unsigned)state; }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

typedef std::array<double, 3> P3;
typedef std::array<double, 2> P2;
inline double f32(double v) { return (double)(float)v; }

// projection with an identity camera matrix; J (2n x 6, [rvec | tvec]) when asked
inline void project(const double* rvec, const double* tvec, const std::vector<P3>& X, std::vector<P2>& uv, Mat* J) {
    double R[9]; rodrigues(rvec, R);
    double dR[3][9];
    if (J) { d_rodrigues(rvec, dR); *J = Mat(2 * (int)X.size(), 6); }
    uv.resize(X.size());
    for (size_t i = 0; i < X.size(); i++) {
        double P[3]; m3v(R, X[i].data(), P);
        for (int k = 0; k < 3; k++) P[k] += tvec[k];
        const double iz = 1.0 / P[2];
        uv[i] = {P[0] * iz, P[1] * iz};
        if (J) {
            const double du[3] = {iz, 0, -P[0] * iz * iz}, dv[3] = {0, iz, -P[1] * iz * iz};
            for (int k = 0; k < 3; k++) {
                double dP[3]; m3v(dR[k], X[i].data(), dP);
                (*J)(2 * (int)i, k) = du[0] * dP[0] + du[1] * dP[1] + du[2] * dP[2];
                (*J)(2 * (int)i + 1, k) = dv[0] * dP[0] + dv[1] * dP[1] + dv[2] * dP[2];
            }
            for (int k = 0; k < 3; k++) { (*J)(2 * (int)i, 3 + k) = du[k]; (*J)(2 * (int)i + 1, 3 + k) = dv[k]; }
        }
    }
}

// CvLevMarq driven the way cvFindExtrinsicCameraParams2 drives it: lambda 10^-3, x 10 on a worse step, / 10 on a better one, 20 iterations,
// relative parameter change < FLT_EPSILON
inline bool pnp_refine(const std::vector<P3>& X, const std::vector<P2>& m, double* rvec, double* tvec) {
    double param[6] = {rvec[0], rvec[1], rvec[2], tvec[0], tvec[1], tvec[2]}, prev[6];
    int lam = -3, iters = 0;
    double prev_err = 0;
    const int n2 = 2 * (int)X.size();
    std::vector<P2> p;
    for (;;) {
        Mat J;
        project(param, param + 3, X, p, &J);
        std::vector<double> err(n2);
        for (size_t i = 0; i < X.size(); i++) { err[2 * i] = p[i][0] - m[i][0]; err[2 * i + 1] = p[i][1] - m[i][1]; }
        const Mat JtJ = AtA(J);
        const std::vector<double> JtE = Atb(J, err);
        for (int i = 0; i < 6; i++) prev[i] = param[i];
        auto step = [&]() {
            std::vector<double> A(JtJ.a), x;
            const double f = 1.0 + exp(lam * log(10.0));
            for (int i = 0; i < 6; i++) A[7 * i] *= f;
            if (!lu_solve(A, JtE, 6, x)) return false;
            for (int i = 0; i < 6; i++) param[i] = prev[i] - x[i];
            return true;
        };
        if (!step()) return false;
        if (iters == 0) { double s = 0; for (double e : err) s += e * e; prev_err = sqrt(s); }
        double en = 0;
        for (;;) {
            project(param, param + 3, X, p, nullptr);
            double s = 0;
            for (size_t i = 0; i < X.size(); i++) { const double a = p[i][0] - m[i][0], b = p[i][1] - m[i][1]; s += a * a + b * b; }
            en = sqrt(s);
            if (en > prev_err) { if (++lam <= 16) { if (!step()) return false; continue; } }   // a NaN error compares false: accepted, as in OpenCV
            break;
        }
        lam = std::max(lam - 1, -16);
        iters++;
        double dn = 0, pn = 0;
        for (int i = 0; i < 6; i++) { dn += (param[i] - prev[i]) * (param[i] - prev[i]); pn += prev[i] * prev[i]; }
        if (iters >= 20 || sqrt(dn) < (double)FLT_EPSILON * sqrt(pn)) break;
        prev_err = en;
    }
    for (int i = 0; i < 3; i++) { rvec[i] = param[i]; tvec[i] = param[3 + i]; }
    return std::isfinite(param[0] + param[1] + param[2] + param[3] + param[4] + param[5]);
}

// cv::findHomography(src, dst, 0): the normalised DLT of HomographyEstimatorCallback::runKernel (fundam.cpp); its Levenberg-Marquardt passes over H are not
// restated (H only starts the pose refinement that follows)
inline bool homography_dlt(const std::vector<P2>& src, const std::vector<P2>& dst, double* H) {
    const int n = (int)src.size();
    double cM[2] = {0, 0}, cm[2] = {0, 0}, sM[2] = {0, 0}, sm[2] = {0, 0};
    for (int i = 0; i < n; i++) for (int k = 0; k < 2; k++) { cM[k] += src[i][k]; cm[k] += dst[i][k]; }
    for (int k = 0; k < 2; k++) { cM[k] /= n; cm[k] /= n; }
    for (int i = 0; i < n; i++) for (int k = 0; k < 2; k++) { sM[k] += fabs(src[i][k] - cM[k]); sm[k] += fabs(dst[i][k] - cm[k]); }
    if (std::min(std::min(sM[0], sM[1]), std::min(sm[0], sm[1])) < DBL_EPSILON) return false;
    for (int k = 0; k < 2; k++) { sM[k] = n / sM[k]; sm[k] = n / sm[k]; }
    Mat A(2 * n, 9);
    for (int i = 0; i < n; i++) {
        const double x = (dst[i][0] - cm[0]) * sm[0], y = (dst[i][1] - cm[1]) * sm[1], Xn = (src[i][0] - cM[0]) * sM[0], Yn = (src[i][1] - cM[1]) * sM[1];
        const double r0[9] = {Xn, Yn, 1, 0, 0, 0, -x * Xn, -x * Yn, -x}, r1[9] = {0, 0, 0, Xn, Yn, 1, -y * Xn, -y * Yn, -y};
        for (int k = 0; k < 9; k++) { A(2 * i, k) = r0[k]; A(2 * i + 1, k) = r1[k]; }
    }
    std::vector<double> w; Mat V;
    sym_eig(AtA(A), w, V);
    double H0[9];
    for (int k = 0; k < 9; k++) H0[k] = V(k, 0);
    const double inv_norm[9] = {1.0 / sm[0], 0, cm[0], 0, 1.0 / sm[1], cm[1], 0, 0, 1}, norm2[9] = {sM[0], 0, -cM[0] * sM[0], 0, sM[1], -cM[1] * sM[1], 0, 0, 1};
    m3mul(inv_norm, H0, H); m3mul(H, norm2, H);
    const double h22 = H[8];
    for (int k = 0; k < 9; k++) H[k] /= h22;
    return true;
}
// cvFindExtrinsicCameraParams2 without a guess, planar branch; w, V: eigen-decomposition (ascending) of the scatter of X
inline void pnp_planar(const std::vector<P3>& X, const std::vector<P2>& uv, const double* mean, const Mat& V, double* rvec, double* tvec) {
    double Rt[9];
    for (int j = 0; j < 3; j++) { Rt[j] = V(j, 2); Rt[3 + j] = V(j, 1); Rt[6 + j] = V(j, 0); }
    if (Rt[2] * Rt[2] + Rt[5] * Rt[5] < 1e-10) for (int i = 0; i < 9; i++) Rt[i] = (i % 4 == 0);
    if (m3det(Rt) < 0) for (int i = 0; i < 9; i++) Rt[i] = -Rt[i];
    double tt[3];
    m3v(Rt, mean, tt);
    for (int k = 0; k < 3; k++) tt[k] = -tt[k];
    std::vector<P2> Mxy(X.size());
    for (size_t i = 0; i < X.size(); i++) { double p[3]; m3v(Rt, X[i].data(), p); Mxy[i] = {p[0] + tt[0], p[1] + tt[1]}; }
    double H[9];
    bool fin = homography_dlt(Mxy, uv, H);
    for (int k = 0; k < 9 && fin; k++) fin = std::isfinite(H[k]);
    for (int k = 0; k < 3; k++) rvec[k] = tvec[k] = 0;
    if (!fin) return;
    const double n1 = sqrt(H[0] * H[0] + H[3] * H[3] + H[6] * H[6]), n2 = sqrt(H[1] * H[1] + H[4] * H[4] + H[7] * H[7]);
    const double h1[3] = {H[0] / std::max(n1, DBL_EPSILON), H[3] / std::max(n1, DBL_EPSILON), H[6] / std::max(n1, DBL_EPSILON)};
    const double h2[3] = {H[1] / std::max(n2, DBL_EPSILON), H[4] / std::max(n2, DBL_EPSILON), H[7] / std::max(n2, DBL_EPSILON)};
    const double f = 2.0 / std::max(n1 + n2, DBL_EPSILON), t[3] = {H[2] * f, H[5] * f, H[8] * f};
    double h3[3];
    cross3(h1, h2, h3);
    const double Mh[9] = {h1[0], h2[0], h3[0], h1[1], h2[1], h3[1], h1[2], h2[2], h3[2]};
    double Rh[9], R[9], Rtt[3];
    if (!polar_rotation(Mh, Rh)) return;
    m3mul(Rh, Rt, R);
    rodrigues_inv(R, rvec);
    m3v(Rh, tt, Rtt);
    for (int k = 0; k < 3; k++) tvec[k] = Rtt[k] + t[k];
}
// cvFindExtrinsicCameraParams2 without a guess: DLT for a non-planar point set, homography for a planar one
inline bool pnp_dlt(const std::vector<P3>& X, const std::vector<P2>& uv, double* rvec, double* tvec) {
    const int n = (int)X.size();
    double mean[3] = {0, 0, 0};
    for (auto& p : X) for (int k = 0; k < 3; k++) mean[k] += p[k];
    for (int k = 0; k < 3; k++) mean[k] /= n;
    Mat MM(3, 3);
    for (auto& p : X) for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) MM(i, j) += (p[i] - mean[i]) * (p[j] - mean[j]);
    std::vector<double> w; Mat V;
    sym_eig(MM, w, V);
    if (w[0] / w[1] < 1e-3) { pnp_planar(X, uv, mean, V, rvec, tvec); return true; }
    Mat L(2 * n, 12);
    for (int i = 0; i < n; i++) {
        const double x = -uv[i][0], y = -uv[i][1];
        for (int k = 0; k < 3; k++) { L(2 * i, k) = X[i][k]; L(2 * i + 1, 4 + k) = X[i][k]; L(2 * i, 8 + k) = x * X[i][k]; L(2 * i + 1, 8 + k) = y * X[i][k]; }
        L(2 * i, 3) = 1; L(2 * i + 1, 7) = 1; L(2 * i, 11) = x; L(2 * i + 1, 11) = y;
    }
    sym_eig(AtA(L), w, V);
    double RR[9], tt[3];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) RR[3 * i + j] = V(4 * i + j, 0); tt[i] = V(4 * i + 3, 0); }
    if (m3det(RR) < 0) { for (int i = 0; i < 9; i++) RR[i] = -RR[i]; for (int i = 0; i < 3; i++) tt[i] = -tt[i]; }
    double sc = 0;
    for (int i = 0; i < 9; i++) sc += RR[i] * RR[i];
    sc = sqrt(sc);
    double R[9];
    if (!(sc > DBL_EPSILON) || !polar_rotation(RR, R)) return false;
    double rn = 0;
    for (int i = 0; i < 9; i++) rn += R[i] * R[i];
    rn = sqrt(rn);
    rodrigues_inv(R, rvec);
    for (int i = 0; i < 3; i++) tvec[i] = tt[i] * (rn / sc);
    return true;
}

// cv::solvePnP(obj, img, I, noArray, rvec, tvec, use_guess, SOLVEPNP_ITERATIVE): points pass through float as cv::Point3f / cv::Point2f do
inline bool solve_pnp_iterative(const std::vector<P3>& X_in, const std::vector<P2>& uv_in, double* rvec, double* tvec, bool use_guess) {
    std::vector<P3> X(X_in.size()); std::vector<P2> uv(uv_in.size());
    for (size_t i = 0; i < X.size(); i++) { X[i] = {f32(X_in[i][0]), f32(X_in[i][1]), f32(X_in[i][2])}; uv[i] = {f32(uv_in[i][0]), f32(uv_in[i][1])}; }
    if (!use_guess && !pnp_dlt(X, uv, rvec, tvec)) return false;
    return pnp_refine(X, uv, rvec, tvec);
}

// ---------------------------------------------------------------- EPnP (epnp.cpp)
inline bool epnp(const std::vector<P3>& X, const std::vector<P2>& uv, double* rvec, double* tvec) {
    static const int PA[6] = {0, 0, 0, 1, 1, 2}, PB[6] = {1, 2, 3, 2, 3, 3};
    const int n = (int)X.size();
    double c0[3] = {0, 0, 0};
    for (auto& p : X) for (int k = 0; k < 3; k++) c0[k] += p[k];
    for (int k = 0; k < 3; k++) c0[k] /= n;
    Mat S(3, 3);
    for (auto& p : X) for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) S(i, j) += (p[i] - c0[i]) * (p[j] - c0[j]);
    std::vector<double> w; Mat V;
    sym_eig(S, w, V);
    double cws[4][3];
    for (int k = 0; k < 3; k++) cws[0][k] = c0[k];
    for (int i = 0; i < 3; i++) { const double kk = sqrt(std::max(w[2 - i], 0.0) / n); for (int k = 0; k < 3; k++) cws[i + 1][k] = c0[k] + kk * V(k, 2 - i); }
    std::vector<double> CC(9);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) CC[3 * i + j] = cws[j + 1][i] - cws[0][i];
    Mat al(n, 4);
    for (int i = 0; i < n; i++) {
        std::vector<double> rhs = {X[i][0] - cws[0][0], X[i][1] - cws[0][1], X[i][2] - cws[0][2]}, a;
        if (!lu_solve(CC, rhs, 3, a)) return false;
        al(i, 0) = 1.0 - (a[0] + a[1] + a[2]); al(i, 1) = a[0]; al(i, 2) = a[1]; al(i, 3) = a[2];
    }
    Mat M(2 * n, 12);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 4; j++) {
            M(2 * i, 3 * j) = al(i, j); M(2 * i, 3 * j + 2) = al(i, j) * (0.0 - uv[i][0]);
            M(2 * i + 1, 3 * j + 1) = al(i, j); M(2 * i + 1, 3 * j + 2) = al(i, j) * (0.0 - uv[i][1]);
        }
    std::vector<double> w12; Mat V12;
    sym_eig(AtA(M), w12, V12);
    int k0 = 0;
    for (int i = 0; i < 12; i++) if (w12[i] < 1e-9 * w12[11]) k0++;
    k0 = std::max(1, k0);
    double vs[4][12];
    for (int i = 0; i < 4; i++) for (int k = 0; k < 12; k++) vs[i][k] = V12(k, i);
    if (k0 >= 2) {   // the basis inside a null space is rounding's choice: take the canonical one
        k0 = std::min(k0, 4);
        Mat Bc;
        if (!canonical_subspace_basis(V12, k0, Bc)) return false;
        for (int i = 0; i < k0; i++) for (int k = 0; k < 12; k++) vs[i][k] = Bc(k, i);
    }
    double dv[4][6][3];
    for (int i = 0; i < 4; i++) for (int p = 0; p < 6; p++) for (int k = 0; k < 3; k++) dv[i][p][k] = vs[i][3 * PA[p] + k] - vs[i][3 * PB[p] + k];
    Mat L(6, 10);
    std::vector<double> rho(6);
    auto dt = [](const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    for (int p = 0; p < 6; p++) {
        const double *d0 = dv[0][p], *d1 = dv[1][p], *d2 = dv[2][p], *d3 = dv[3][p];
        const double row[10] = {dt(d0, d0), 2 * dt(d0, d1), dt(d1, d1), 2 * dt(d0, d2), 2 * dt(d1, d2), dt(d2, d2), 2 * dt(d0, d3), 2 * dt(d1, d3), 2 * dt(d2, d3), dt(d3, d3)};
        for (int k = 0; k < 10; k++) L(p, k) = row[k];
        double d[3];
        for (int k = 0; k < 3; k++) d[k] = cws[PA[p]][k] - cws[PB[p]][k];
        rho[p] = dt(d, d);
    }
    auto sub = [&](std::initializer_list<int> cols) { Mat A(6, (int)cols.size()); int j = 0; for (int c : cols) { for (int i = 0; i < 6; i++) A(i, j) = L(i, c); j++; } return A; };
    double best_err = -1, best_R[9], best_t[3];
    for (int which = 1; which <= 3; which++) {
        double be[4] = {0, 0, 0, 0};
        std::vector<double> bb;
        bool ok = true;
        if (which == 1) {
            ok = qr_lsq(sub({0, 1, 3, 6}), rho, bb);
            if (ok) {
                if (bb[0] < 0) { be[0] = sqrt(-bb[0]); for (int k = 1; k < 4; k++) be[k] = -bb[k] / be[0]; }
                else { be[0] = sqrt(bb[0]); for (int k = 1; k < 4; k++) be[k] = bb[k] / be[0]; }
            }
        } else {
            ok = which == 2 ? qr_lsq(sub({0, 1, 2}), rho, bb) : qr_lsq(sub({0, 1, 2, 3, 4}), rho, bb);
            if (ok) {
                if (bb[0] < 0) { be[0] = sqrt(-bb[0]); be[1] = bb[2] < 0 ? sqrt(-bb[2]) : 0.0; }
                else { be[0] = sqrt(bb[0]); be[1] = bb[2] > 0 ? sqrt(bb[2]) : 0.0; }
                if (bb[1] < 0) be[0] = -be[0];
                if (which == 3) be[2] = bb[3] / be[0];
            }
        }
        for (int it = 0; it < 5 && ok; it++) {   // gauss_newton
            Mat A(6, 4);
            std::vector<double> b(6), dx;
            for (int i = 0; i < 6; i++) {
                const double* l = &L.a[(size_t)i * 10];
                A(i, 0) = 2 * l[0] * be[0] + l[1] * be[1] + l[3] * be[2] + l[6] * be[3];
                A(i, 1) = l[1] * be[0] + 2 * l[2] * be[1] + l[4] * be[2] + l[7] * be[3];
                A(i, 2) = l[3] * be[0] + l[4] * be[1] + 2 * l[5] * be[2] + l[8] * be[3];
                A(i, 3) = l[6] * be[0] + l[7] * be[1] + l[8] * be[2] + 2 * l[9] * be[3];
                b[i] = rho[i] - (l[0] * be[0] * be[0] + l[1] * be[0] * be[1] + l[2] * be[1] * be[1] + l[3] * be[0] * be[2] + l[4] * be[1] * be[2] + l[5] * be[2] * be[2]
                                 + l[6] * be[0] * be[3] + l[7] * be[1] * be[3] + l[8] * be[2] * be[3] + l[9] * be[3] * be[3]);
            }
            bool fin = true;
            for (double v : A.a) fin = fin && std::isfinite(v);
            ok = fin && qr_lsq(A, b, dx);
            if (ok) for (int k = 0; k < 4; k++) be[k] += dx[k];
        }
        if (!ok) continue;
        double ccs[4][3] = {{0}};
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) for (int k = 0; k < 3; k++) ccs[j][k] += be[i] * vs[i][3 * j + k];
        std::vector<P3> pcs(n);
        for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) pcs[i][k] = al(i, 0) * ccs[0][k] + al(i, 1) * ccs[1][k] + al(i, 2) * ccs[2][k] + al(i, 3) * ccs[3][k];
        if (pcs[0][2] < 0) for (auto& p : pcs) for (int k = 0; k < 3; k++) p[k] = -p[k];
        double pc0[3] = {0, 0, 0};
        for (auto& p : pcs) for (int k = 0; k < 3; k++) pc0[k] += p[k];
        for (int k = 0; k < 3; k++) pc0[k] /= n;
        double ABt[9] = {0};
        for (int i = 0; i < n; i++) for (int a = 0; a < 3; a++) for (int b2 = 0; b2 < 3; b2++) ABt[3 * a + b2] += (pcs[i][a] - pc0[a]) * (X[i][b2] - c0[b2]);
        bool fin = true;
        for (double v : ABt) fin = fin && std::isfinite(v);
        double R[9];
        if (!fin || !svd_rotation(ABt, R)) continue;   // R = U V^T of ABt (epnp.cpp estimate_R_and_t: cvSVD)
        if (m3det(R) < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
        double t[3], Rp[3];
        m3v(R, c0, Rp);
        for (int k = 0; k < 3; k++) t[k] = pc0[k] - Rp[k];
        double err = 0;
        for (int i = 0; i < n; i++) {
            double P[3]; m3v(R, X[i].data(), P);
            for (int k = 0; k < 3; k++) P[k] += t[k];
            const double a = P[0] / P[2] - uv[i][0], b2 = P[1] / P[2] - uv[i][1];
            err += sqrt(a * a + b2 * b2);
        }
        err /= n;
        if (std::isfinite(err) && (best_err < 0 || err < best_err)) { best_err = err; for (int i = 0; i < 9; i++) best_R[i] = R[i]; for (int i = 0; i < 3; i++) best_t[i] = t[i]; }
    }
    if (best_err < 0) return false;
    rodrigues_inv(best_R, rvec);
    for (int i = 0; i < 3; i++) tvec[i] = best_t[i];
    return true;
}

// ---------------------------------------------------------------- cv::solvePnPRansac(obj, img, I, noArray, rvec, tvec, false, 100, 1/460, 0.99, inliers, SOLVEPNP_ITERATIVE)
inline int ransac_update_num_iters(double p, double ep, int model_points, int max_iters) {
    p = std::min(std::max(p, 0.0), 1.0); ep = std::min(std::max(ep, 0.0), 1.0);
    double num = std::max(1.0 - p, DBL_MIN), denom = 1.0 - std::pow(1.0 - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = std::log(num); denom = std::log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)std::nearbyint(num / denom);
}
inline bool solve_pnp_ransac(const std::vector<P3>& X_in, const std::vector<P2>& uv_in, double* rvec, double* tvec, std::vector<int>* inliers = nullptr) {
    const int n = (int)X_in.size(), mp = 5;
    if (n < mp) return false;
    std::vector<P3> X(n); std::vector<P2> uv(n);
    for (int i = 0; i < n; i++) { X[i] = {f32(X_in[i][0]), f32(X_in[i][1]), f32(X_in[i][2])}; uv[i] = {f32(uv_in[i][0]), f32(uv_in[i][1])}; }
    const float thr = (float)((1.0 / 460) * (1.0 / 460));
    CvRNG rng;
    int niters = 100, best = 0;
    std::vector<char> best_mask;
    std::vector<P2> pr;
    for (int it = 0; it < niters; it++) {
        int idx[5];
        if (n > mp) {
            for (int i = 0; i < mp;) {   // getSubset: redraw on a repeated index
                const int v = rng.uniform(0, n);
                bool dup = false;
                for (int j = 0; j < i; j++) dup = dup || idx[j] == v;
                if (dup) continue;
                idx[i++] = v;
            }
        } else for (int i = 0; i < mp; i++) idx[i] = i;
        std::vector<P3> Xs(mp); std::vector<P2> us(mp);
        for (int i = 0; i < mp; i++) { Xs[i] = X[idx[i]]; us[i] = uv[idx[i]]; }
        double rv[3], tv[3];
        if (!epnp(Xs, us, rv, tv)) continue;
        project(rv, tv, X, pr, nullptr);
        std::vector<char> mask(n);
        int good = 0;
        for (int i = 0; i < n; i++) {
            const float dx = (float)uv[i][0] - (float)pr[i][0], dy = (float)uv[i][1] - (float)pr[i][1];
            const float e = dx * dx + dy * dy;
            mask[i] = e <= thr;
            good += mask[i];
        }
        if (getenv("GF_INIT_DEBUG")) fprintf(stderr, "ransac it %d subset %d %d %d %d %d good %d rv %.12g %.12g %.12g tv %.12g %.12g %.12g\n", it, idx[0], idx[1], idx[2], idx[3], idx[4], good, rv[0], rv[1], rv[2], tv[0], tv[1], tv[2]);
        if (good > std::max(best, mp - 1)) {
            best = good; best_mask = mask;
            niters = ransac_update_num_iters(0.99, (double)(n - good) / n, mp, niters);
        }
    }
    if (best_mask.empty()) return false;
    std::vector<P3> Xi; std::vector<P2> ui;
    for (int i = 0; i < n; i++) if (best_mask[i]) { Xi.push_back(X[i]); ui.push_back(uv[i]); if (inliers) inliers->push_back(i); }
    return solve_pnp_iterative(Xi, ui, rvec, tvec, false);
}

// MotionEstimator::solveRelativeRT_PNP, solve_5pts.cpp:244-277.  corres: (x, y, z) in frame l and in the newest frame, both scaled by their depth.
// The rotation is built as Rx(r0) Ry(r1) Rz(r2) from the Rodrigues vector, as the SO3(double, double, double) constructor of the non-templated Sophus does.
inline bool solve_relative_rt_pnp(const std::vector<std::array<double, 6>>& corres, double* Rot, double* Tr, std::vector<int>* inliers = nullptr) {
    std::vector<P3> X; std::vector<P2> uv;
    for (auto& c : corres)
        if (c[2] > 0 && c[5] > 0) { X.push_back({c[0], c[1], c[2]}); uv.push_back({c[3] / c[5], c[4] / c[5]}); }
    double rv[3], tv[3];
    if (!solve_pnp_ransac(X, uv, rv, tv, inliers)) return false;
    const double cx = cos(rv[0]), sx = sin(rv[0]), cy = cos(rv[1]), sy = sin(rv[1]), cz = cos(rv[2]), sz = sin(rv[2]);
    const double Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx}, Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy}, Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
    double rota[9];
    m3mul(Rx, Ry, rota); m3mul(rota, Rz, rota);
    m3T(rota, Rot);
    m3v(Rot, tv, Tr);
    for (int i = 0; i < 3; i++) Tr[i] = -Tr[i];
    return true;
}

// ---------------------------------------------------------------- quaternions (w, x, y, z)
inline void quat_rot(const double* q, double* R) {   // rotation of q / |q|
    const double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double w = q[0] / nq, x = q[1] / nq, y = q[2] / nq, z = q[3] / nq;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}
inline void quat_plus(const double* q, const double* d, double* o) {   // ceres::QuaternionParameterization::Plus
    const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (nd == 0.0) { for (int i = 0; i < 4; i++) o[i] = q[i]; return; }
    const double s = sin(nd) / nd, a[4] = {cos(nd), s * d[0], s * d[1], s * d[2]};
    const double t[4] = {a[0] * q[0] - a[1] * q[1] - a[2] * q[2] - a[3] * q[3], a[0] * q[1] + a[1] * q[0] + a[2] * q[3] - a[3] * q[2],
                         a[0] * q[2] - a[1] * q[3] + a[2] * q[0] + a[3] * q[1], a[0] * q[3] + a[1] * q[2] - a[2] * q[1] + a[3] * q[0]};
    for (int i = 0; i < 4; i++) o[i] = t[i];
}
inline void quat_from_R(const double* R, double* q) {   // Eigen::Quaterniond(Matrix3d)
    const double t = R[0] + R[4] + R[8];
    if (t > 0) {
        double s = sqrt(t + 1.0);
        q[0] = 0.5 * s; s = 0.5 / s;
        q[1] = (R[7] - R[5]) * s; q[2] = (R[2] - R[6]) * s; q[3] = (R[3] - R[1]) * s;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        q[1 + i] = 0.5 * s; s = 0.5 / s;
        q[0] = (R[3 * k + j] - R[3 * j + k]) * s;
        q[1 + j] = (R[3 * j + i] + R[3 * i + j]) * s;
        q[1 + k] = (R[3 * k + i] + R[3 * i + k]) * s;
    }
}

// ---------------------------------------------------------------- the structure-from-motion bundle adjustment (ceres::Solve, LEVENBERG_MARQUARDT, DENSE_SCHUR)
struct SfmObs { int frame, point; double u, v; };
// qs / ts / pts are updated in place; returns CONVERGENCE || final_cost < 5e-3 (initial_sfm.cpp:547)
inline bool sfm_bundle_adjust(std::vector<std::array<double, 4>>& qs, std::vector<P3>& ts, std::vector<P3>& pts, const std::vector<SfmObs>& obs, int const_rot, int const_t0, int const_t1,
                              double* final_cost = nullptr) {
    const int nf = (int)qs.size(), np = (int)pts.size(), no = (int)obs.size();
    std::vector<int> rcol(nf, -1), tcol(nf, -1);
    int nc = 0;
    for (int i = 0; i < nf; i++) {
        if (i != const_rot) { rcol[i] = nc; nc += 3; }
        if (i != const_t0 && i != const_t1) { tcol[i] = nc; nc += 3; }
    }
    const int ncol = nc + 3 * np;
    struct ObsJ { double Jr[6], Jt[6], Jp[6], r[2]; };   // 2 x 3 blocks, row-major
    std::vector<ObsJ> oj(no);
    auto evaluate = [&](const std::vector<std::array<double, 4>>& Q, const std::vector<P3>& T, const std::vector<P3>& P, bool jac, std::vector<ObsJ>& out) {
        double cost = 0;
        for (int k = 0; k < no; k++) {
            const SfmObs& o = obs[k];
            double R[9], Rx[3], p[3];
            quat_rot(Q[o.frame].data(), R);
            m3v(R, P[o.point].data(), Rx);
            for (int a = 0; a < 3; a++) p[a] = Rx[a] + T[o.frame][a];
            const double iz = 1.0 / p[2];
            ObsJ& e = out[k];
            e.r[0] = p[0] * iz - o.u; e.r[1] = p[1] * iz - o.v;
            cost += e.r[0] * e.r[0] + e.r[1] * e.r[1];
            if (jac) {
                const double D[6] = {iz, 0, -p[0] * iz * iz, 0, iz, -p[1] * iz * iz};
                double S[9]; skew(Rx, S);
                for (int a = 0; a < 2; a++)
                    for (int b = 0; b < 3; b++) {
                        e.Jt[3 * a + b] = D[3 * a + b];
                        e.Jr[3 * a + b] = -2.0 * (D[3 * a] * S[b] + D[3 * a + 1] * S[3 + b] + D[3 * a + 2] * S[6 + b]);
                        e.Jp[3 * a + b] = D[3 * a] * R[b] + D[3 * a + 1] * R[3 + b] + D[3 * a + 2] * R[6 + b];
                    }
            }
        }
        return 0.5 * cost;
    };
    double cost = evaluate(qs, ts, pts, true, oj);
    if (final_cost) *final_cost = cost;
    if (ncol == 0 || no == 0) return true;
    // column norms -> Jacobi scaling, fixed for the whole solve
    auto col_norms = [&](const std::vector<ObsJ>& J, const std::vector<double>* scale) {
        std::vector<double> s(ncol, 0.0);
        for (int k = 0; k < no; k++) {
            const SfmObs& o = obs[k];
            for (int b = 0; b < 3; b++) {
                const double sr = rcol[o.frame] >= 0 ? (scale ? (*scale)[rcol[o.frame] + b] : 1.0) : 0.0, st = tcol[o.frame] >= 0 ? (scale ? (*scale)[tcol[o.frame] + b] : 1.0) : 0.0,
                             sp = scale ? (*scale)[nc + 3 * o.point + b] : 1.0;
                for (int a = 0; a < 2; a++) {
                    if (rcol[o.frame] >= 0) s[rcol[o.frame] + b] += J[k].Jr[3 * a + b] * sr * J[k].Jr[3 * a + b] * sr;
                    if (tcol[o.frame] >= 0) s[tcol[o.frame] + b] += J[k].Jt[3 * a + b] * st * J[k].Jt[3 * a + b] * st;
                    s[nc + 3 * o.point + b] += J[k].Jp[3 * a + b] * sp * J[k].Jp[3 * a + b] * sp;
                }
            }
        }
        return s;
    };
    std::vector<double> scale = col_norms(oj, nullptr);
    for (double& v : scale) v = 1.0 / (1.0 + sqrt(v));
    double radius = 1e4, decrease = 2.0;
    bool reuse = false, last_ok = true, converged = false;
    int invalid_run = 0;
    std::vector<double> diag;
    for (int it = 0;;) {
        if (it >= 50) break;
        // scaled gradient
        std::vector<double> g(ncol, 0.0);
        for (int k = 0; k < no; k++) {
            const SfmObs& o = obs[k];
            for (int b = 0; b < 3; b++)
                for (int a = 0; a < 2; a++) {
                    if (rcol[o.frame] >= 0) g[rcol[o.frame] + b] += oj[k].Jr[3 * a + b] * oj[k].r[a];
                    if (tcol[o.frame] >= 0) g[tcol[o.frame] + b] += oj[k].Jt[3 * a + b] * oj[k].r[a];
                    g[nc + 3 * o.point + b] += oj[k].Jp[3 * a + b] * oj[k].r[a];
                }
        }
        double gmax = 0;
        for (double v : g) gmax = std::max(gmax, fabs(v));   // unscaled gradient: (J_s^T r) / scale = J^T r
        for (int i = 0; i < ncol; i++) g[i] *= scale[i];
        if (last_ok && gmax <= 1e-10) { converged = true; break; }
        if (radius <= 1e-32) break;
        it++;
        if (!reuse) { diag = col_norms(oj, &scale); for (double& v : diag) v = std::min(std::max(v, 1e-6), 1e32); }
        // normal equations in the scaled variables with the Levenberg-Marquardt diagonal, Schur complement over the points
        Mat Hcc(nc, nc);
        std::vector<std::array<double, 9>> Hpp(np);
        for (auto& h : Hpp) h.fill(0.0);
        struct W { int cols[6], ncols; double w[18]; };   // (camera columns of this observation) x 3
        std::vector<W> Wk(no);
        for (int k = 0; k < no; k++) {
            const SfmObs& o = obs[k];
            W& wk = Wk[k];
            wk.ncols = 0;
            double Jc[12];   // 2 x ncols
            double tmp[2][6];
            for (int b = 0; b < 3; b++) if (rcol[o.frame] >= 0) { wk.cols[wk.ncols] = rcol[o.frame] + b; for (int a = 0; a < 2; a++) tmp[a][wk.ncols] = oj[k].Jr[3 * a + b] * scale[rcol[o.frame] + b]; wk.ncols++; }
            for (int b = 0; b < 3; b++) if (tcol[o.frame] >= 0) { wk.cols[wk.ncols] = tcol[o.frame] + b; for (int a = 0; a < 2; a++) tmp[a][wk.ncols] = oj[k].Jt[3 * a + b] * scale[tcol[o.frame] + b]; wk.ncols++; }
            (void)Jc;
            double Jp[2][3];
            for (int b = 0; b < 3; b++) for (int a = 0; a < 2; a++) Jp[a][b] = oj[k].Jp[3 * a + b] * scale[nc + 3 * o.point + b];
            for (int x = 0; x < wk.ncols; x++) {
                for (int y = 0; y < wk.ncols; y++) Hcc(wk.cols[x], wk.cols[y]) += tmp[0][x] * tmp[0][y] + tmp[1][x] * tmp[1][y];
                for (int b = 0; b < 3; b++) wk.w[3 * x + b] = tmp[0][x] * Jp[0][b] + tmp[1][x] * Jp[1][b];
            }
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Hpp[o.point][3 * a + b] += Jp[0][a] * Jp[0][b] + Jp[1][a] * Jp[1][b];
        }
        for (int i = 0; i < nc; i++) Hcc(i, i) += diag[i] / radius;
        bool valid = true;
        std::vector<std::array<double, 9>> Hinv(np);
        for (int j = 0; j < np && valid; j++) {
            for (int a = 0; a < 3; a++) Hpp[j][4 * a] += diag[nc + 3 * j + a] / radius;
            const double* h = Hpp[j].data();
            const double det = m3det(h);
            if (!(fabs(det) > 0) || !std::isfinite(det)) { valid = false; break; }
            double* o = Hinv[j].data();
            o[0] = (h[4] * h[8] - h[5] * h[7]) / det; o[1] = (h[2] * h[7] - h[1] * h[8]) / det; o[2] = (h[1] * h[5] - h[2] * h[4]) / det;
            o[3] = (h[5] * h[6] - h[3] * h[8]) / det; o[4] = (h[0] * h[8] - h[2] * h[6]) / det; o[5] = (h[2] * h[3] - h[0] * h[5]) / det;
            o[6] = (h[3] * h[7] - h[4] * h[6]) / det; o[7] = (h[1] * h[6] - h[0] * h[7]) / det; o[8] = (h[0] * h[4] - h[1] * h[3]) / det;
        }
        std::vector<double> step(ncol, 0.0);
        if (valid) {
            std::vector<std::vector<int>> of_point(np);
            for (int k = 0; k < no; k++) of_point[obs[k].point].push_back(k);
            std::vector<double> rhs(nc);
            for (int i = 0; i < nc; i++) rhs[i] = -g[i];
            for (int j = 0; j < np; j++) {
                double hg[3];
                m3v(Hinv[j].data(), &g[nc + 3 * j], hg);
                for (int k : of_point[j]) {
                    const W& a = Wk[k];
                    double aw[6][3];   // W_k Hpp^-1
                    for (int x = 0; x < a.ncols; x++) for (int b = 0; b < 3; b++) aw[x][b] = a.w[3 * x] * Hinv[j][b] + a.w[3 * x + 1] * Hinv[j][3 + b] + a.w[3 * x + 2] * Hinv[j][6 + b];
                    for (int x = 0; x < a.ncols; x++) rhs[a.cols[x]] += a.w[3 * x] * hg[0] + a.w[3 * x + 1] * hg[1] + a.w[3 * x + 2] * hg[2];
                    for (int k2 : of_point[j]) {
                        const W& c = Wk[k2];
                        for (int x = 0; x < a.ncols; x++) for (int y = 0; y < c.ncols; y++) Hcc(a.cols[x], c.cols[y]) -= aw[x][0] * c.w[3 * y] + aw[x][1] * c.w[3 * y + 1] + aw[x][2] * c.w[3 * y + 2];
                    }
                }
            }
            std::vector<double> dc;
            if (nc > 0) valid = lu_solve(Hcc.a, rhs, nc, dc);
            if (valid) {
                for (int i = 0; i < nc; i++) step[i] = dc[i];
                for (int j = 0; j < np; j++) {
                    double v[3] = {-g[nc + 3 * j], -g[nc + 3 * j + 1], -g[nc + 3 * j + 2]};
                    for (int k : of_point[j]) { const W& a = Wk[k]; for (int x = 0; x < a.ncols; x++) for (int b = 0; b < 3; b++) v[b] -= a.w[3 * x + b] * dc[a.cols[x]]; }
                    m3v(Hinv[j].data(), v, &step[nc + 3 * j]);
                }
                for (double v : step) valid = valid && std::isfinite(v);
            }
        }
        double mcc = 0;
        if (valid) {
            for (int k = 0; k < no; k++) {
                const SfmObs& o = obs[k];
                for (int a = 0; a < 2; a++) {
                    double mr = 0;
                    for (int b = 0; b < 3; b++) {
                        if (rcol[o.frame] >= 0) mr += oj[k].Jr[3 * a + b] * scale[rcol[o.frame] + b] * step[rcol[o.frame] + b];
                        if (tcol[o.frame] >= 0) mr += oj[k].Jt[3 * a + b] * scale[tcol[o.frame] + b] * step[tcol[o.frame] + b];
                        mr += oj[k].Jp[3 * a + b] * scale[nc + 3 * o.point + b] * step[nc + 3 * o.point + b];
                    }
                    mcc -= mr * (oj[k].r[a] + mr / 2.0);
                }
            }
            valid = mcc > 0.0;
        }
        if (!valid) {
            last_ok = false;
            if (++invalid_run >= 5) break;
            radius /= decrease; decrease *= 2.0; reuse = true;
            continue;
        }
        invalid_run = 0;
        std::vector<std::array<double, 4>> cq = qs;
        std::vector<P3> ct = ts, cp = pts;
        for (int i = 0; i < nf; i++) {
            if (rcol[i] >= 0) { const double d[3] = {step[rcol[i]] * scale[rcol[i]], step[rcol[i] + 1] * scale[rcol[i] + 1], step[rcol[i] + 2] * scale[rcol[i] + 2]}; quat_plus(qs[i].data(), d, cq[i].data()); }
            if (tcol[i] >= 0) for (int b = 0; b < 3; b++) ct[i][b] = ts[i][b] + step[tcol[i] + b] * scale[tcol[i] + b];
        }
        for (int j = 0; j < np; j++) for (int b = 0; b < 3; b++) cp[j][b] = pts[j][b] + step[nc + 3 * j + b] * scale[nc + 3 * j + b];
        std::vector<ObsJ> cj(no);
        const double ccost = evaluate(cq, ct, cp, false, cj);
        double sn = 0, xn = 0;
        for (int i = 0; i < nf; i++) {   // the reduced program: constant blocks are not part of the state vector
            if (rcol[i] >= 0) for (int b = 0; b < 4; b++) { sn += (qs[i][b] - cq[i][b]) * (qs[i][b] - cq[i][b]); xn += qs[i][b] * qs[i][b]; }
            if (tcol[i] >= 0) for (int b = 0; b < 3; b++) { sn += (ts[i][b] - ct[i][b]) * (ts[i][b] - ct[i][b]); xn += ts[i][b] * ts[i][b]; }
        }
        for (int j = 0; j < np; j++) for (int b = 0; b < 3; b++) { sn += (pts[j][b] - cp[j][b]) * (pts[j][b] - cp[j][b]); xn += pts[j][b] * pts[j][b]; }
        if (sqrt(sn) <= 1e-8 * (sqrt(xn) + 1e-8)) { converged = true; break; }
        if (fabs(cost - ccost) <= 1e-6 * cost) { converged = true; break; }
        const double rel = (cost - ccost) / mcc;
        if (rel > 1e-3) {
            qs = cq; ts = ct; pts = cp;
            cost = evaluate(qs, ts, pts, true, oj);
            radius = std::min(1e16, radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3.0))); decrease = 2.0; reuse = false;
            last_ok = true;
        } else { radius /= decrease; decrease *= 2.0; reuse = true; last_ok = false; }
    }
    if (final_cost) *final_cost = cost;
    return converged || cost < 5e-3;
}

// ---------------------------------------------------------------- GlobalSFM::constructWithDepth (initial_sfm.cpp:379-594)
struct SfmFeature {
    bool state = false;
    int id = 0;
    std::vector<std::pair<int, P2>> observation;
    std::vector<double> depth;   // observation_depth[k].second
    double position[3] = {0, 0, 0};
};
// q[i] (w, x, y, z), T[i]: camera i in the frame of camera l; tracked: feature id -> position
inline bool construct_with_depth(int frame_num, int l, const double* relative_R, const double* relative_T, std::vector<SfmFeature>& sfm_f, std::vector<std::array<double, 4>>& q_out,
                                 std::vector<P3>& T_out, std::map<int, P3>& tracked) {
    const int last = frame_num - 1;
    std::vector<std::array<double, 9>> cR(frame_num);
    std::vector<P3> cT(frame_num);
    std::vector<std::array<double, 4>> cQ(frame_num);
    for (auto& r : cR) r = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (auto& t : cT) t = {0, 0, 0};
    for (auto& qq : cQ) qq = {1, 0, 0, 0};
    {
        double ql[4];
        quat_from_R(relative_R, ql);
        const double n2 = ql[0] * ql[0] + ql[1] * ql[1] + ql[2] * ql[2] + ql[3] * ql[3];
        cQ[last] = {ql[0] / n2, -ql[1] / n2, -ql[2] / n2, -ql[3] / n2};
        quat_rot(cQ[last].data(), cR[last].data());
        double t[3];
        m3v(cR[last].data(), relative_T, t);
        cT[last] = {-t[0], -t[1], -t[2]};
    }
    auto pnp = [&](int i, const double* R0, const double* P0) {   // solveFrameByPnP :33-84
        std::vector<P3> X; std::vector<P2> uv;
        for (auto& f : sfm_f) {
            if (!f.state) continue;
            for (auto& o : f.observation) if (o.first == i) { uv.push_back(o.second); X.push_back({f.position[0], f.position[1], f.position[2]}); break; }
        }
        if ((int)uv.size() < 10) return false;
        double rv[3], tv[3] = {P0[0], P0[1], P0[2]};
        rodrigues_inv(R0, rv);
        if (!solve_pnp_iterative(X, uv, rv, tv, true)) return false;
        rodrigues(rv, cR[i].data());
        cT[i] = {tv[0], tv[1], tv[2]};
        quat_from_R(cR[i].data(), cQ[i].data());
        return true;
    };
    auto check = [&](SfmFeature& f, const double* p0, const P2& p1, int f0, int f1) {
        double a[3], b[3], X[3], pr[3];
        m3Tv(cR[f0].data(), p0, a); m3Tv(cR[f0].data(), cT[f0].data(), b);
        for (int k = 0; k < 3; k++) X[k] = a[k] - b[k];
        m3v(cR[f1].data(), X, pr);
        for (int k = 0; k < 3; k++) pr[k] += cT[f1][k];
        const double rx = p1[0] - pr[0] / pr[2], ry = p1[1] - pr[1] / pr[2];
        if (sqrt(rx * rx + ry * ry) < 1.0 / 460) { f.state = true; for (int k = 0; k < 3; k++) f.position[k] = X[k]; }
    };
    auto tri = [&](int f0, int f1) {   // triangulateTwoFramesWithDepth :133-182
        for (auto& f : sfm_f) {
            if (f.state) continue;
            bool h0 = false, h1 = false;
            double p0[3]; P2 p1{};
            for (size_t k = 0; k < f.observation.size(); k++) {
                const double d = f.depth[k];
                if (d < 0.1 || d > 10) continue;
                if (f.observation[k].first == f0) { p0[0] = f.observation[k].second[0] * d; p0[1] = f.observation[k].second[1] * d; p0[2] = d; h0 = true; }
                if (f.observation[k].first == f1) { p1 = f.observation[k].second; h1 = true; }
            }
            if (h0 && h1) check(f, p0, p1, f0, f1);
        }
    };
    for (int i = l; i < last; i++) {
        if (i > l && !pnp(i, cR[i - 1].data(), cT[i - 1].data())) return false;
        tri(i, last);
    }
    for (int i = l + 1; i < last; i++) tri(l, i);
    for (int i = l - 1; i >= 0; i--) {
        if (!pnp(i, cR[i + 1].data(), cT[i + 1].data())) return false;
        tri(i, l);
    }
    for (auto& f : sfm_f) {   // :461-497
        if (f.state || f.observation.size() < 2) continue;
        const double d = f.depth[0];
        if (d < 0.1 || d > 10) continue;
        const double p0[3] = {f.observation[0].second[0] * d, f.observation[0].second[1] * d, d};
        check(f, p0, f.observation.back().second, f.observation[0].first, f.observation.back().first);
    }
    std::vector<P3> pts;
    std::vector<int> live;
    std::vector<SfmObs> obs;
    for (size_t j = 0; j < sfm_f.size(); j++) {
        if (!sfm_f[j].state) continue;
        for (auto& o : sfm_f[j].observation) obs.push_back({o.first, (int)pts.size(), o.second[0], o.second[1]});
        live.push_back((int)j);
        pts.push_back({sfm_f[j].position[0], sfm_f[j].position[1], sfm_f[j].position[2]});
    }
    if (!sfm_bundle_adjust(cQ, cT, pts, obs, l, l, last)) return false;
    for (size_t k = 0; k < live.size(); k++) for (int a = 0; a < 3; a++) sfm_f[live[k]].position[a] = pts[k][a];
    q_out.resize(frame_num); T_out.resize(frame_num);
    for (int i = 0; i < frame_num; i++) {
        const double n2 = cQ[i][0] * cQ[i][0] + cQ[i][1] * cQ[i][1] + cQ[i][2] * cQ[i][2] + cQ[i][3] * cQ[i][3];
        q_out[i] = {cQ[i][0] / n2, -cQ[i][1] / n2, -cQ[i][2] / n2, -cQ[i][3] / n2};
        double R[9], t[3];
        quat_rot(q_out[i].data(), R);
        m3v(R, cT[i].data(), t);
        T_out[i] = {-t[0], -t[1], -t[2]};
    }
    tracked.clear();
    for (auto& f : sfm_f) if (f.state) tracked[f.id] = {f.position[0], f.position[1], f.position[2]};
    return true;
}

// ---------------------------------------------------------------- VisualIMUAlignment after solveGyroscopeBias: linear alignment + gravity refinement
struct AlignFrame { double R[9], T[3], sum_dt, delta_p[3], delta_v[3], wheel_delta_p[3]; };   // pre-integration of the interval that ENDS at this frame
inline void tangent_basis(const double* g0, double* lxly /* 3 x 2 */) {   // initial_aligment.cpp:49-62
    const double n = sqrt(g0[0] * g0[0] + g0[1] * g0[1] + g0[2] * g0[2]);
    const double a[3] = {g0[0] / n, g0[1] / n, g0[2] / n};
    double tmp[3] = {0, 0, 1};
    if (a[0] == tmp[0] && a[1] == tmp[1] && a[2] == tmp[2]) { tmp[0] = 1; tmp[2] = 0; }
    const double d = a[0] * tmp[0] + a[1] * tmp[1] + a[2] * tmp[2];
    double b[3] = {tmp[0] - a[0] * d, tmp[1] - a[1] * d, tmp[2] - a[2] * d};
    const double bn = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    for (int k = 0; k < 3; k++) b[k] /= bn;
    double c[3];
    cross3(a, b, c);
    for (int k = 0; k < 3; k++) { lxly[2 * k] = b[k]; lxly[2 * k + 1] = c[k]; }
}
// Literal: the wheel rows' `scale' column lies on the last gravity column (8 of 9, then 7 of 8); the refinement's normal matrix is not cleared between its four
// passes and is multiplied by 1000 in each.  g: gravity in the frame of camera l; x: the last solution vector (3 n + 2 entries).
inline bool linear_alignment(const std::vector<AlignFrame>& fr, const double* TIC, double g_norm, bool use_wheel, const double* RIO, const double* TIO, double* g, std::vector<double>& x) {
    const int n = (int)fr.size(), rows = use_wheel ? 9 : 6;
    auto blocks = [&](int kg, const double* lxly, const double* g0, std::vector<double>& A, std::vector<double>& b) {
        const int ns = 3 * n + kg;
        A.assign((size_t)ns * ns, 0.0); b.assign(ns, 0.0);
        for (int i = 0; i + 1 < n; i++) {
            const AlignFrame &fi = fr[i], &fj = fr[i + 1];
            const double dt = fj.sum_dt;
            const int nc = 6 + kg;
            double tA[9][9] = {{0}}, tb[9] = {0};
            double RiT[9], RiTRj[9], v[3], u[3];
            m3T(fi.R, RiT); m3mul(RiT, fj.R, RiTRj);
            for (int a = 0; a < 3; a++) { tA[a][a] = -dt; tA[3 + a][a] = -1.0; for (int c = 0; c < 3; c++) tA[3 + a][3 + c] = RiTRj[3 * a + c]; }
            m3v(RiTRj, TIC, v);
            if (kg == 3) {
                for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++) { tA[a][6 + c] = RiT[3 * a + c] * (dt * dt / 2); tA[3 + a][6 + c] = RiT[3 * a + c] * dt; }
                for (int a = 0; a < 3; a++) { tb[a] = fj.delta_p[a] + v[a] - TIC[a]; tb[3 + a] = fj.delta_v[a]; }
            } else {
                double M1[9], M2[9];
                for (int k = 0; k < 9; k++) { M1[k] = RiT[k] * (dt * dt / 2); M2[k] = RiT[k] * dt; }
                for (int a = 0; a < 3; a++) for (int c = 0; c < 2; c++) {
                    tA[a][6 + c] = M1[3 * a] * lxly[c] + M1[3 * a + 1] * lxly[2 + c] + M1[3 * a + 2] * lxly[4 + c];
                    tA[3 + a][6 + c] = M2[3 * a] * lxly[c] + M2[3 * a + 1] * lxly[2 + c] + M2[3 * a + 2] * lxly[4 + c];
                }
                double m1g[3], m2g[3];
                m3v(M1, g0, m1g); m3v(M2, g0, m2g);
                for (int a = 0; a < 3; a++) { tb[a] = fj.delta_p[a] + v[a] - TIC[a] - m1g[a]; tb[3 + a] = fj.delta_v[a] - m2g[a]; }
            }
            const double dT[3] = {fj.T[0] - fi.T[0], fj.T[1] - fi.T[1], fj.T[2] - fi.T[2]};
            if (use_wheel) {
                double RiRio[9], RiRioT[9], RioT[9], w[3], t1[3], t2[3], t3[3], d[3];
                m3mul(fi.R, RIO, RiRio); m3T(RiRio, RiRioT); m3T(RIO, RioT);
                m3v(RiRioT, dT, w);
                for (int a = 0; a < 3; a++) tA[6 + a][nc - 1] = w[a] / 100;
                m3v(RiTRj, TIO, u); m3v(RioT, u, t1);           // RIO^T Ri^T Rj TIO
                m3v(fj.R, TIC, u); m3v(RiRioT, u, t2);          // (Ri RIO)^T Rj TIC
                for (int a = 0; a < 3; a++) d[a] = TIC[a] - TIO[a];
                m3v(RioT, d, t3);
                for (int a = 0; a < 3; a++) tb[6 + a] = fj.wheel_delta_p[a] - t1[a] + t2[a] - t3[a];
            } else {
                m3v(RiT, dT, u);
                for (int a = 0; a < 3; a++) tb[a] -= u[a];
            }
            double rA[9][9] = {{0}}, rb[9] = {0};
            for (int a = 0; a < nc; a++) {
                for (int c = 0; c < nc; c++) { double s = 0; for (int k = 0; k < rows; k++) s += tA[k][a] * tA[k][c]; rA[a][c] = s; }
                double s = 0;
                for (int k = 0; k < rows; k++) s += tA[k][a] * tb[k];
                rb[a] = s;
            }
            auto idx = [&](int a) { return a < 6 ? 3 * i + a : ns - kg + (a - 6); };
            for (int a = 0; a < nc; a++) { for (int c = 0; c < nc; c++) A[(size_t)idx(a) * ns + idx(c)] += rA[a][c]; b[idx(a)] += rb[a]; }
        }
    };
    std::vector<double> A, b;
    blocks(3, nullptr, nullptr, A, b);
    for (double& v : A) v *= 1000.0;
    for (double& v : b) v *= 1000.0;
    if (!lu_solve(A, b, 3 * n + 3, x)) return false;
    double g0[3] = {x[3 * n], x[3 * n + 1], x[3 * n + 2]};
    double gn = sqrt(g0[0] * g0[0] + g0[1] * g0[1] + g0[2] * g0[2]);
    if (fabs(gn - g_norm) > (use_wheel ? 0.5 : 1.0)) return false;
    for (int k = 0; k < 3; k++) g0[k] = g0[k] / gn * g_norm;
    const int ns = 3 * n + 2;
    std::vector<double> Aacc((size_t)ns * ns, 0.0), bacc(ns, 0.0), dA, db;
    for (int pass = 0; pass < 4; pass++) {
        double lxly[6];
        tangent_basis(g0, lxly);
        blocks(2, lxly, g0, dA, db);
        for (size_t k = 0; k < Aacc.size(); k++) Aacc[k] = (Aacc[k] + dA[k]) * 1000.0;
        for (int k = 0; k < ns; k++) bacc[k] = (bacc[k] + db[k]) * 1000.0;
        if (!lu_solve(Aacc, bacc, ns, x)) return false;
        for (int k = 0; k < 3; k++) g0[k] += lxly[2 * k] * x[ns - 2] + lxly[2 * k + 1] * x[ns - 1];
        gn = sqrt(g0[0] * g0[0] + g0[1] * g0[1] + g0[2] * g0[2]);
        for (int k = 0; k < 3; k++) g0[k] = g0[k] / gn * g_norm;
    }
    for (int k = 0; k < 3; k++) g[k] = g0[k];
    return true;
}

}  // namespace gfinit
