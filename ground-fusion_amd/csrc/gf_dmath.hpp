// gf_dmath.hpp — small fixed-size FP64 math shared by the back end's HIP kernels and its host code
// (quaternions in Eigen's Hamilton convention, the helpers of utility/utility.h:23-89, the SO(3) exp/log and
// right Jacobians the wheel factor takes from Sophus / utility/sophus_utils.hpp:154-236).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace gfd {

#define GFD __host__ __device__ __forceinline__

struct V3 { double x, y, z; };
struct M3 { double m[9]; };  // row-major
struct Q4 { double w, x, y, z; };

GFD V3 v3(double x, double y, double z) { return V3{x, y, z}; }
GFD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
GFD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
GFD V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
GFD V3 operator*(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
GFD V3 operator/(V3 a, double s) { return V3{a.x / s, a.y / s, a.z / s}; }
GFD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
GFD V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
GFD double sqn(V3 a) { return dot(a, a); }

GFD M3 m3_identity() { M3 r; for (int i = 0; i < 9; i++) r.m[i] = (i % 4 == 0) ? 1.0 : 0.0; return r; }
GFD M3 m3_zero() { M3 r; for (int i = 0; i < 9; i++) r.m[i] = 0.0; return r; }
GFD M3 m3_diag(double a, double b, double c) { M3 r = m3_zero(); r.m[0] = a; r.m[4] = b; r.m[8] = c; return r; }
GFD M3 operator*(const M3& a, const M3& b) {
    M3 r;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
    return r;
}
GFD V3 operator*(const M3& a, V3 v) { return V3{a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z, a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z}; }
GFD M3 operator*(const M3& a, double s) { M3 r; for (int i = 0; i < 9; i++) r.m[i] = a.m[i] * s; return r; }
GFD M3 operator+(const M3& a, const M3& b) { M3 r; for (int i = 0; i < 9; i++) r.m[i] = a.m[i] + b.m[i]; return r; }
GFD M3 operator-(const M3& a, const M3& b) { M3 r; for (int i = 0; i < 9; i++) r.m[i] = a.m[i] - b.m[i]; return r; }
GFD M3 operator-(const M3& a) { M3 r; for (int i = 0; i < 9; i++) r.m[i] = -a.m[i]; return r; }
GFD M3 transpose(const M3& a) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[3 * i + j] = a.m[3 * j + i]; return r; }
GFD M3 skew(V3 q) { M3 r = m3_zero(); r.m[1] = -q.z; r.m[2] = q.y; r.m[3] = q.z; r.m[5] = -q.x; r.m[6] = -q.y; r.m[7] = q.x; return r; }  // utility.h:38-46

GFD Q4 qmul(Q4 a, Q4 b) {
    return Q4{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
              a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
GFD double qn2(Q4 q) { return q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z; }
GFD Q4 qnormalized(Q4 q) { const double n = sqrt(qn2(q)); return Q4{q.w / n, q.x / n, q.y / n, q.z / n}; }
GFD Q4 qinverse(Q4 q) { const double n2 = qn2(q); return Q4{q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2}; }  // Eigen: conjugate / squaredNorm
GFD V3 qvec(Q4 q) { return V3{q.x, q.y, q.z}; }
GFD M3 qmat(Q4 q) {  // Eigen toRotationMatrix
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z, twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x,
                 tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    M3 r;
    r.m[0] = 1 - (tyy + tzz); r.m[1] = txy - twz; r.m[2] = txz + twy;
    r.m[3] = txy + twz; r.m[4] = 1 - (txx + tzz); r.m[5] = tyz - twx;
    r.m[6] = txz - twy; r.m[7] = tyz + twx; r.m[8] = 1 - (txx + tyy);
    return r;
}
// Eigen::Quaterniond(Matrix3d)
GFD Q4 rot_to_quat(const M3& Rm) {
    const double* R = Rm.m;
    double t = R[0] + R[4] + R[8], w, v[3];
    if (t > 0) { t = sqrt(t + 1.0); w = 0.5 * t; t = 0.5 / t; v[0] = (R[7] - R[5]) * t; v[1] = (R[2] - R[6]) * t; v[2] = (R[3] - R[1]) * t; }
    else {
        int i = 0; if (R[4] > R[0]) i = 1; if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0); v[i] = 0.5 * t; t = 0.5 / t;
        w = (R[3 * k + j] - R[3 * j + k]) * t; v[j] = (R[3 * j + i] + R[3 * i + j]) * t; v[k] = (R[3 * k + i] + R[3 * i + k]) * t;
    }
    return Q4{w, v[0], v[1], v[2]};
}
GFD V3 qrot(Q4 q, V3 v) {  // Eigen _transformVector
    V3 u = qvec(q);
    V3 uv = cross(u, v);
    uv = uv + uv;
    return v + uv * q.w + cross(u, uv);
}
GFD Q4 deltaQ(V3 th) { return qnormalized(Q4{1.0, th.x / 2.0, th.y / 2.0, th.z / 2.0}); }  // utility.h:23-36
// bottom-right 3x3 of Qleft(q) / Qright(q) (utility.h:58-76)
GFD M3 qleft33(Q4 q) { return m3_identity() * q.w + skew(qvec(q)); }
GFD M3 qright33(Q4 q) { return m3_identity() * q.w - skew(qvec(q)); }
// bottom-right 3x3 of Qleft(a) * Qright(b)
GFD M3 qleft_qright33(Q4 a, Q4 b) {
    // (Ql(a) Qr(b))_{1..3,1..3} = a_v (-b_v)^T + (a_w I + [a_v]x)(b_w I - [b_v]x)
    M3 L = qleft33(a), R = qright33(b), P = L * R;
    const V3 av = qvec(a), bv = qvec(b);
    P.m[0] -= av.x * bv.x; P.m[1] -= av.x * bv.y; P.m[2] -= av.x * bv.z;
    P.m[3] -= av.y * bv.x; P.m[4] -= av.y * bv.y; P.m[5] -= av.y * bv.z;
    P.m[6] -= av.z * bv.x; P.m[7] -= av.z * bv.y; P.m[8] -= av.z * bv.z;
    return P;
}

// Sophus::SO3d exp / log, Constants<double>::epsilon() = 1e-10
GFD Q4 so3_exp(V3 om) {
    const double t2 = sqn(om);
    double imag, real;
    if (t2 < 1e-20) { const double t4 = t2 * t2; imag = 0.5 - (1.0 / 48.0) * t2 + (1.0 / 3840.0) * t4; real = 1.0 - (1.0 / 8.0) * t2 + (1.0 / 384.0) * t4; }
    else { const double t = sqrt(t2), h = 0.5 * t; imag = sin(h) / t; real = cos(h); }
    return Q4{real, imag * om.x, imag * om.y, imag * om.z};
}
GFD V3 so3_log(Q4 qin) {
    const Q4 q = qnormalized(qin);
    const double n2 = q.x * q.x + q.y * q.y + q.z * q.z, w = q.w;
    double f;
    if (n2 < 1e-20) f = 2.0 / w - (2.0 / 3.0) * n2 / (w * w * w);
    else { const double n = sqrt(n2); f = fabs(w) < 1e-10 ? (w > 0 ? M_PI : -M_PI) / n : 2.0 * atan(n / w) / n; }
    return V3{f * q.x, f * q.y, f * q.z};
}
GFD M3 rightJacobianSO3(V3 phi) {  // sophus_utils.hpp:154-184
    const double n2 = sqn(phi);
    const M3 h = skew(phi), h2 = h * h;
    M3 J = m3_identity();
    if (n2 > 1e-10) { const double n = sqrt(n2), n3 = n2 * n; J = J - h * ((1 - cos(n)) / n2); J = J + h2 * ((n - sin(n)) / n3); }
    else { J = J - h * 0.5; J = J + h2 * (1.0 / 6.0); }
    return J;
}
GFD M3 rightJacobianInvSO3(V3 phi) {  // sophus_utils.hpp:194-236
    const double n2 = sqn(phi);
    const M3 h = skew(phi), h2 = h * h;
    M3 J = m3_identity() + h * 0.5;
    if (n2 > 1e-10) {
        const double n = sqrt(n2);
        if (n < M_PI - 1e-5) J = J + h2 * (1 / n2 - (1 + cos(n)) / (2 * n * sin(n)));
        else J = J + h2 * (1.0 / (M_PI * M_PI));
    } else J = J + h2 * (1.0 / 12.0);
    return J;
}

}  // namespace gfd
