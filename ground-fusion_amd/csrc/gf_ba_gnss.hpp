// gf_ba_gnss.hpp — GNSS residual blocks of the sliding window on the device:
//   GnssPsrDoppFactor  factor/gnss_psr_dopp_factor.cpp:49-208   pseudorange + Doppler of one satellite, interpolated between two frames
//   DtDdtFactor        factor/gnss_dt_ddt_factor.cpp            receiver clock bias vs drift between consecutive frames (x4 constellations)
//   DdtSmoothFactor    factor/gnss_ddt_smooth_factor.cpp        drift smoothness
//   PoseAnchorFactor   factor/pose_anchor_factor.cpp            gauge anchor of Pose[0] at the first optimisation
// The blocks are tiny (2 x 18, 1 x 4, 1 x 2, 6 x 6): one thread evaluates a GnssPsrDoppFactor, and J^T J / J^T r of the factors that share
// parameter blocks are summed per entry in a fixed order (ba_linearize_gnss below).  ECEF magnitudes are ~6.4e6 m, so everything stays FP64.
// gnss_comm (not vendored by the reference) supplies ecef2geo / ecef2rotation / sat_azel / Saastamoinen / Klobuchar; restated from the published
// algorithms (RTKLIB lineage), the same formulas as the CPU oracle -- both documented as "parity unpinned" against gnss_comm itself.
#pragma once
#include "gf_ba_kernels.hpp"

namespace gfb {

constexpr double GN_C = 2.99792458e8, GN_OMG = 7.2921151467e-5, GN_A = 6378137.0, GN_E2 = 6.69437999014e-3, GN_PI = 3.14159265358979323846;

__device__ inline V3 gn_ecef2geo(V3 p) {   // latitude [deg], longitude [deg], height [m]
    if (p.x == 0 && p.y == 0) return v3(0, 0, 0);
    const double a = GN_A, a2 = a * a, b2 = a2 * (1 - GN_E2), b = sqrt(b2), ep2 = (a2 - b2) / b2, rho = sqrt(p.x * p.x + p.y * p.y);
    double s1 = p.z * a, s2 = rho * b, h = sqrt(s1 * s1 + s2 * s2);
    const double st = s1 / h, ct = s2 / h;
    s1 = p.z + ep2 * b * st * st * st;
    s2 = rho - a * GN_E2 * ct * ct * ct;
    h = sqrt(s1 * s1 + s2 * s2);
    const double sin_lat = s1 / h, cos_lat = s2 / h;
    const double N = a2 / sqrt(a2 * cos_lat * cos_lat + b2 * sin_lat * sin_lat);
    return v3(atan(s1 / s2) * 180.0 / GN_PI, atan2(p.y, p.x) * 180.0 / GN_PI, rho / cos_lat - N);
}
__device__ inline M3 gn_geo2rotation(V3 lla) {   // R_ecef_enu
    const double lat = lla.x * GN_PI / 180.0, lon = lla.y * GN_PI / 180.0, sl = sin(lat), cl = cos(lat), so = sin(lon), co = cos(lon);
    M3 R;
    R.m[0] = -so; R.m[1] = -sl * co; R.m[2] = cl * co;
    R.m[3] = co;  R.m[4] = -sl * so; R.m[5] = cl * so;
    R.m[6] = 0;   R.m[7] = cl;       R.m[8] = sl;
    return R;
}
__device__ inline void gn_sat_azel(V3 rcv, V3 sat, double& az, double& el) {
    V3 dl = sat - rcv; dl = dl / sqrt(sqn(dl));
    const V3 enu = transpose(gn_geo2rotation(gn_ecef2geo(rcv))) * dl;
    az = (sqrt(dl.x * dl.x + dl.y * dl.y) < 1e-12) ? 0.0 : atan2(enu.x, enu.y);
    if (az < 0) az += 2 * GN_PI;
    el = asin(enu.z);
}
__device__ inline double gn_trop_delay(V3 lla, double el) {   // Saastamoinen, standard atmosphere, humidity 0.7
    if (lla.z < -100.0 || 1e4 < lla.z || el <= 0) return 0.0;
    const double hgt = lla.z < 0.0 ? 0.0 : lla.z;
    const double pres = 1013.25 * pow(1.0 - 2.2557e-5 * hgt, 5.2568), temp = 15.0 - 6.5e-3 * hgt + 273.16;
    const double e = 6.108 * 0.7 * exp((17.15 * temp - 4684.0) / (temp - 38.45)), z = GN_PI / 2.0 - el;
    const double trph = 0.0022768 * pres / (1.0 - 0.00266 * cos(2.0 * lla.x * GN_PI / 180.0) - 0.00028 * hgt / 1e3) / cos(z);
    const double trpw = 0.002277 * (1255.0 / temp + 0.05) * e / cos(z);
    return trph + trpw;
}
__device__ inline double gn_ion_delay(double tow, const double* ion_in, V3 lla, double az, double el) {   // Klobuchar
    const double ion_default[8] = {0.1118e-07, -0.7451e-08, -0.5961e-07, 0.1192e-06, 0.1167e+06, -0.2294e+06, -0.1311e+06, 0.1049e+07};
    if (lla.z < -1e3 || el <= 0) return 0.0;
    double nrm = 0;
    for (int i = 0; i < 8; i++) nrm += ion_in[i] * ion_in[i];
    double ion[8];
    for (int i = 0; i < 8; i++) ion[i] = nrm <= 0.0 ? ion_default[i] : ion_in[i];
    const double psi = 0.0137 / (el / GN_PI + 0.11) - 0.022;
    double phi = lla.x / 180.0 + psi * cos(az);
    if (phi > 0.416) phi = 0.416; else if (phi < -0.416) phi = -0.416;
    const double lam = lla.y / 180.0 + psi * sin(az) / cos(phi * GN_PI);
    phi += 0.064 * cos((lam - 1.617) * GN_PI);
    double tt = 43200.0 * lam + tow;
    tt -= floor(tt / 86400.0) * 86400.0;
    const double f = 1.0 + 16.0 * pow(0.53 - el / GN_PI, 3.0);
    double amp = ion[0] + phi * (ion[1] + phi * (ion[2] + phi * ion[3])), per = ion[4] + phi * (ion[5] + phi * (ion[6] + phi * ion[7]));
    amp = amp < 0.0 ? 0.0 : amp; per = per < 72000.0 ? 72000.0 : per;
    const double x = 2.0 * GN_PI * (tt - 50400.0) / per;
    return GN_C * f * (fabs(x) < 1.57 ? 5e-9 + amp * (1.0 + x * x * (-0.5 + x * x / 24.0)) : 5e-9);
}

// GnssPsrDoppFactor::Evaluate (gnss_psr_dopp_factor.cpp:49-208) of one satellite observation: residual r (2) and Jacobian J (2 x 18, row-major).
// local columns: 0-2 P_lo, 3-5 V_lo, 6-8 P_hi, 9-11 V_hi, 12 rcv_dt, 13 rcv_ddt, 14 yaw_enu_local, 15-17 anc_ecef
__device__ inline void gn_psr_dopp(const Dims& d, const double* xs, const double* misc, const double* dat, int fi, int lo, int sys, double* r, double* J) {
    const int NP = d.NP;
    const V3 sv_pos = v3(dat[0], dat[1], dat[2]), sv_vel = v3(dat[3], dat[4], dat[5]);
    const double svdt = dat[6], svddt = dat[7], tgd = dat[8], pr_uura = dat[9], dp_uura = dat[10], psr = dat[11], dopp = dat[12], wavelength = dat[13], tow = dat[14];
    const double ratio = dat[16];
    const double* Pi_ = xs + off_pose(lo); const double* Pj_ = xs + off_pose(lo + 1);
    const double* SBi = xs + off_sb(lo); const double* SBj = xs + off_sb(lo + 1);
    const V3 Pi = p_of(Pi_), Pj = p_of(Pj_), Vi = v3(SBi[0], SBi[1], SBi[2]), Vj = v3(SBj[0], SBj[1], SBj[2]);
    const double rcv_dt = xs[d.GO + 4 * fi + sys], rcv_ddt = xs[d.GO + 4 * NP + fi], yaw_diff = xs[d.GO + 5 * NP];
    const V3 ref_ecef = v3(xs[d.GO + 5 * NP + 1], xs[d.GO + 5 * NP + 2], xs[d.GO + 5 * NP + 3]);
    const V3 local_pos = Pi * ratio + Pj * (1.0 - ratio), local_vel = Vi * ratio + Vj * (1.0 - ratio);
    const double sy = sin(yaw_diff), cy = cos(yaw_diff);
    M3 R_enu_local = m3_zero();
    R_enu_local.m[0] = cy; R_enu_local.m[1] = -sy; R_enu_local.m[3] = sy; R_enu_local.m[4] = cy; R_enu_local.m[8] = 1;
    const M3 R_ecef_enu = gn_geo2rotation(gn_ecef2geo(ref_ecef)), R_ecef_local = R_ecef_enu * R_enu_local;
    const V3 P_ecef = R_ecef_local * local_pos + ref_ecef, V_ecef = R_ecef_local * local_vel;
    double ion_delay = 0, tro_delay = 0, az = 0, el = GN_PI / 2.0;
    if (sqn(P_ecef) > 0) {
        gn_sat_azel(P_ecef, sv_pos, az, el);
        const V3 lla = gn_ecef2geo(P_ecef);
        tro_delay = gn_trop_delay(lla, el);
        ion_delay = gn_ion_delay(tow, misc, lla, az, el);
    }
    const double sin_el = sin(el), sin_el_2 = sin_el * sin_el;
    const double pr_weight = sin_el_2 / pr_uura * 10.0, dp_weight = sin_el_2 / dp_uura * 10.0 * 5.0;   // relative_sqrt_info 10, PSR_TO_DOPP_RATIO 5
    const V3 rcv2sat = sv_pos - P_ecef;
    const double norm2 = sqn(rcv2sat), rng = sqrt(norm2);
    const V3 unit = rcv2sat / rng;
    const double psr_sagnac = GN_OMG * (sv_pos.x * P_ecef.y - sv_pos.y * P_ecef.x) / GN_C;
    const double psr_est = rng + psr_sagnac + rcv_dt - svdt * GN_C + ion_delay + tro_delay + tgd * GN_C;
    const double dopp_sagnac = GN_OMG / GN_C * (sv_vel.x * P_ecef.y + sv_pos.x * V_ecef.y - sv_vel.y * P_ecef.x - sv_pos.y * V_ecef.x);
    const V3 dv = sv_vel - V_ecef;
    const double dopp_est = dot(dv, unit) + dopp_sagnac + rcv_ddt - svddt * GN_C;
    r[0] = (psr_est - psr) * pr_weight; r[1] = (dopp_est + dopp * wavelength) * dp_weight;
    for (int i = 0; i < 36; i++) J[i] = 0.0;
    const double norm3 = rng * rng * rng;
    const double rs[3] = {rcv2sat.x, rcv2sat.y, rcv2sat.z}, un[3] = {unit.x, unit.y, unit.z}, dvv[3] = {dv.x, dv.y, dv.z};
    double t[3];   // (sv_vel - V)^T unit2rcv_pos, unit2rcv_pos = -(d unit / d rcv2sat)
    for (int c = 0; c < 3; c++) { double a = 0; for (int q = 0; q < 3; q++) a += dvv[q] * -((q == c) ? (norm2 - rs[q] * rs[q]) / norm3 : (-rs[q] * rs[c]) / norm3); t[c] = a; }
    for (int c = 0; c < 3; c++) {
        double a = 0, bsum = 0;
        for (int q = 0; q < 3; q++) { a += un[q] * R_ecef_local.m[3 * q + c]; bsum += t[q] * R_ecef_local.m[3 * q + c]; }
        J[c] = -a * pr_weight * ratio;              J[18 + c] = bsum * dp_weight * ratio;
        J[18 + 3 + c] = -a * dp_weight * ratio;
        J[6 + c] = -a * pr_weight * (1.0 - ratio);  J[18 + 6 + c] = bsum * dp_weight * (1.0 - ratio);
        J[18 + 9 + c] = -a * dp_weight * (1.0 - ratio);
        J[15 + c] = -un[c] * pr_weight;
    }
    J[12] = pr_weight; J[18 + 13] = dp_weight;
    M3 d_yaw = m3_zero();
    d_yaw.m[0] = -sy; d_yaw.m[1] = -cy; d_yaw.m[3] = cy; d_yaw.m[4] = -sy;
    J[14] = -dot(unit, R_ecef_enu * (d_yaw * local_pos)) * pr_weight;
    J[18 + 14] = -dot(unit, R_ecef_enu * (d_yaw * local_vel)) * dp_weight;
}

// One 256-thread block per window, launched behind ba_linearize_misc_win on the same stream: ADDS the GNSS blocks to H / g (plain
// read-modify-writes, one owner per entry and phase) and writes their cost.  No atomics: every sum runs in a fixed order.
//  A  one thread per GnssPsrDoppFactor: residual and Jacobian rows to the scratch w.gn_rows ([B][NG][38]).
//  B  the factors are grouped by (frame, lower_idx) on the host (gf_ba.hip: gn_gptr / gn_gitem; all factors of a group share their
//     parameter blocks up to the constellation's clock): per group, thread (a >= b) sums J^T J of the group's 21 local columns
//     [P_lo 3, V_lo 3, P_hi 3, V_hi 3, rcv_dt x 4, rcv_ddt, yaw, anc 3] over the group's factors in list order and adds it; groups one
//     after the other (consecutive groups share pose blocks).
//  C  DtDdtFactor x 4 and DdtSmoothFactor of frame pair (i, i + 1): 10 local columns [dt_i x 4, dt_i+1 x 4, ddt_i, ddt_i+1], even i, then odd i.
//  D  PoseAnchorFactor.
// frame_filter as in ba_linearize_misc_win: 1 = blocks of frame 0 only (MARGIN_OLD, taken whenever GNSS is enabled, estimator.cpp:3390),
// 2 = none; 0 = the solve (only when in_solve, i.e. gnss_ready && !lowspeed, estimator.cpp:3178).
constexpr int GN_ROW = 38;
__global__ void __launch_bounds__(256) ba_linearize_gnss(Win w, int which, int which_state, int only_cand_valid, int frame_filter) {
    __shared__ double s_cst[256];
    __shared__ int s_gcol[24];
    const Dims d = w.d;
    const int b = blockIdx.x, tid = threadIdx.x;
    const SolverState& st = w.st[b];
    if (st.done && only_cand_valid != 2) return;
    if (only_cand_valid == 1 && !st.cand_valid) return;
    if (which < 0) which = 1 - st.cur;
    if (which_state == -2) which_state = st.cur; else if (which_state < 0) which_state = 1 - st.cur;
    const double* misc = w.gn_misc + (size_t)b * (GN_MISC + d.NP);
    const bool enabled = misc[16] != 0.0, in_solve = misc[17] != 0.0, has_anchor = misc[18] != 0.0;
    const double* hdr = misc + GN_MISC;
    const double* xs = w.xs + ((size_t)which_state * d.B + b) * d.XS;
    const int* colf = w.colf + (size_t)b * d.NFB;
    const int RP = d.RP, NP = d.NP, W = d.W;
    double* H = w.H + ((size_t)which * d.B + b) * RP * RP;
    double* g = w.g + ((size_t)which * d.B + b) * RP;
    auto cof = [&](int fb, int o) { const int c0 = colf[fb]; return c0 >= 0 ? c0 + o : -1; };
    const bool on = frame_filter != 2 && enabled && (frame_filter != 0 || in_solve);
    const int ng = on ? w.ngnss[b] : 0;
    double* rows = w.gn_rows + (size_t)b * d.NG * GN_ROW;
    double cst = 0.0;   // cost terms owned by this thread, added in a fixed order below
    // ---- A
    for (int item = tid; item < ng; item += 256) {
        const int* ix = w.gn_idx + ((size_t)b * d.NG + item) * 4;
        if (frame_filter == 1 && ix[0] != 0) continue;
        double r[2], J[36];
        gn_psr_dopp(d, xs, misc, w.gn_data + ((size_t)b * d.NG + item) * GN_STRIDE, ix[0], ix[1], ix[2], r, J);
        double* dst = rows + (size_t)item * GN_ROW;
        dst[0] = r[0]; dst[1] = r[1];
        for (int i = 0; i < 36; i++) dst[2 + i] = J[i];
    }
    __syncthreads();
    // ---- B
    {
        const int* gptr = w.gn_gptr + (size_t)b * (d.NGRP + 2);
        const int* gitem = w.gn_gitem + (size_t)b * d.NG;
        const int ngrp = on ? gptr[d.NGRP + 1] : 0;   // the last slot holds the number of groups
        int la = 0, lb = 0;                       // this thread's entry of the 21 x 21 group tile (tid < 231: J^T J lower triangle; 231..251: J^T r)
        if (tid < 231) { la = tri_row(tid); lb = tid - la * (la + 1) / 2; } else if (tid < 252) la = lb = tid - 231;
        auto item_local = [&](int gl, int sys) -> int { return gl < 12 ? gl : gl < 16 ? (gl - 12 == sys ? 12 : -1) : gl == 16 ? 13 : gl == 17 ? 14 : gl - 3; };
        for (int gi = 0; gi < ngrp; gi++) {
            const int p0 = gptr[gi], p1 = gptr[gi + 1];
            const int* ix0 = w.gn_idx + ((size_t)b * d.NG + gitem[p0]) * 4;
            const int fi = ix0[0], lo = ix0[1];
            if (frame_filter == 1 && fi != 0) continue;   // uniform over the block
            if (tid < 21) {
                const int gl = tid;
                s_gcol[gl] = gl < 3 ? cof(fb_pose(lo), gl) : gl < 6 ? cof(fb_sb(lo), gl - 3) : gl < 9 ? cof(fb_pose(lo + 1), gl - 6) : gl < 12 ? cof(fb_sb(lo + 1), gl - 9)
                           : gl < 16 ? cof(fb_rcvdt(NP, 4 * fi + gl - 12), 0) : gl == 16 ? cof(fb_rcvddt(NP, fi), 0) : gl == 17 ? cof(fb_yaw(NP), 0) : cof(fb_anc(NP), gl - 18);
            }
            __syncthreads();
            if (tid < 252) {
                const int ca = s_gcol[la], cb = s_gcol[lb];
                if (ca >= 0 && cb >= 0) {
                    double sum = 0.0;
                    for (int p = p0; p < p1; p++) {
                        const int item = gitem[p];
                        const int sys = w.gn_idx[((size_t)b * d.NG + item) * 4 + 2];
                        const double* rw = rows + (size_t)item * GN_ROW;
                        const int ia = item_local(la, sys);
                        if (ia < 0) continue;
                        if (tid < 231) { const int ib = item_local(lb, sys); if (ib >= 0) sum += rw[2 + ia] * rw[2 + ib] + rw[20 + ia] * rw[20 + ib]; }
                        else sum += rw[2 + ia] * rw[0] + rw[20 + ia] * rw[1];
                    }
                    double* dst = tid < 231 ? H + (size_t)max(ca, cb) * RP + min(ca, cb) : g + ca;
                    *dst += sum;
                }
            } else if (tid == 255) {
                for (int p = p0; p < p1; p++) { const double* rw = rows + (size_t)gitem[p] * GN_ROW; cst += 0.5 * (rw[0] * rw[0] + rw[1] * rw[1]); }
            }
            __syncthreads();
        }
    }
    // ---- C: clock factors of frame pair (i, i + 1); local columns 0-3 dt_i, 4-7 dt_i+1, 8 ddt_i, 9 ddt_i+1; rows q = 0..3 DtDdt, 4 DdtSmooth
    if (on) {
        const double wt = misc[8];
        for (int par = 0; par < 2; par++) {
            for (int t = tid; t < 65 * W; t += 256) {
                const int i = t / 65, e = t - 65 * i;
                if ((i & 1) != par || (frame_filter == 1 && i != 0)) continue;
                int la, lb;
                if (e < 55) { la = tri_row(e); lb = e - la * (la + 1) / 2; } else la = lb = e - 55;
                auto colc = [&](int l) -> int { return l < 4 ? cof(fb_rcvdt(NP, 4 * i + l), 0) : l < 8 ? cof(fb_rcvdt(NP, 4 * (i + 1) + l - 4), 0) : cof(fb_rcvddt(NP, i + l - 8), 0); };
                const int ca = colc(la), cb = colc(lb);
                const double delta_t = hdr[i + 1] - hdr[i];
                const double ddi = xs[d.GO + 4 * NP + i], ddj = xs[d.GO + 4 * NP + i + 1];
                auto jac = [&](int row, int l) -> double {   // d residual(row) / d local column l
                    if (row < 4) return l == row ? -50.0 : l == 4 + row ? 50.0 : l >= 8 ? -0.5 * delta_t * 50.0 : 0.0;
                    return l == 8 ? wt : l == 9 ? -wt : 0.0;
                };
                auto res = [&](int row) -> double {
                    if (row < 4) return (xs[d.GO + 4 * (i + 1) + row] - xs[d.GO + 4 * i + row] - 0.5 * (ddi + ddj) * delta_t) * 50.0;
                    return (ddi - ddj) * wt;
                };
                if (ca >= 0 && cb >= 0) {
                    double sum = 0.0;
                    for (int row = 0; row < 5; row++) sum += e < 55 ? jac(row, la) * jac(row, lb) : jac(row, la) * res(row);
                    double* dst = e < 55 ? H + (size_t)max(ca, cb) * RP + min(ca, cb) : g + ca;
                    *dst += sum;
                }
                if (e == 64) { double c = 0; for (int row = 0; row < 5; row++) c += 0.5 * res(row) * res(row); cst += c; }
            }
            __syncthreads();
        }
    }
    // ---- D: PoseAnchorFactor, sqrt_info 120 (pose_anchor_factor.h:19); only in the solve
    if (frame_filter == 0 && has_anchor && tid < 27) {
        const double* an = misc + 9;
        const double* P = xs + off_pose(0);
        const double si = 120.0;
        double r[6], J[36];
        for (int i = 0; i < 36; i++) J[i] = 0.0;
        for (int i = 0; i < 3; i++) { r[i] = (P[i] - an[i]) * si; J[i * 6 + i] = 2.0 * si; }
        const Q4 ai = qinverse(Q4{an[6], an[3], an[4], an[5]}), e = qmul(q_of(P), ai);
        r[3] = 2.0 * e.x * si; r[4] = 2.0 * e.y * si; r[5] = 2.0 * e.z * si;
        const double Jq[9] = {ai.w, ai.z, -ai.y, -ai.z, ai.w, ai.x, ai.y, -ai.x, ai.w};
        for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++) J[(3 + a) * 6 + 3 + c] = Jq[3 * a + c] * 2.0 * si;
        int la, lb;
        if (tid < 21) { la = tri_row(tid); lb = tid - la * (la + 1) / 2; } else la = lb = tid - 21;
        const int ca = cof(fb_pose(0), la), cb = cof(fb_pose(0), lb);
        if (ca >= 0 && cb >= 0) {
            double sum = 0.0;
            for (int i = 0; i < 6; i++) sum += tid < 21 ? J[i * 6 + la] * J[i * 6 + lb] : J[i * 6 + la] * r[i];
            double* dst = tid < 21 ? H + (size_t)max(ca, cb) * RP + min(ca, cb) : g + ca;
            *dst += sum;
        }
        if (tid == 26) { double c = 0; for (int i = 0; i < 6; i++) c += 0.5 * r[i] * r[i]; cst += c; }
    }
    // ---- cost: the threads' terms in thread order
    s_cst[tid] = cst;
    __syncthreads();
    if (tid == 0) { double c = 0; for (int i = 0; i < 256; i++) c += s_cst[i]; *cost_part(w, 2, which, b) = c; }
}

}  // namespace gfb
