// gf_ba_kernels.hpp — gfx950 device code of the sliding-window back end (Estimator::optimization()).
//
// Batched over independent windows (one block per window).  Per LM iteration:
//   ba_linearize_visual_win : one lane per ProjectionTwoFrameOneCamFactor (residual, analytic Jacobian, Huber corrector), block rows staged
//                             in LDS, J^T J / J^T r of each frame pair contracted with v_mfma_f64_16x16x4_f64 into per-pair tiles, tiles
//                             reduced in a fixed order into the window's compact visual system Vc; E^T F rows of the free inverse depths
//   ba_linearize_misc_win   : the marginalisation prior, IMU and wheel factors: the only writer of H / g (prior gathered, factor tiles
//                             added in parity phases)
//   ba_step                 : S = s (H + Vc) s, Jacobi scaling, dogleg (Cauchy point, Schur complement via MFMA, blocked Cholesky), candidate
// Every sum has a fixed order (no floating-point atomics anywhere): two runs of the same input give bit-identical results.
// Semantics: reference factors (file:line cited per function) + Ceres 1.14 trust_region_minimizer.cc / dogleg_strategy.cc.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gf_dmath.hpp"

namespace gfb {
using namespace gfd;

typedef double d4 __attribute__((ext_vector_type(4)));

struct Dims {
    int B, W, NP, F, NV, NVP, RP, XS, NFB, FP, NPRI, ECW, GO, NG, NC, NVC, NGRP;
    // NP = W+1 poses; NVP = padded length of the pair-sorted factor order; RP = padded reduced dimension (multiple of 16);
    // XS = state vector stride; NFB = 2*NP + 7 non-feature parameter blocks; FP = F rounded up to 4; NPRI = prior capacity (= RP)
    // ECW = width of the compact rows of the eliminated columns: a feature only touches pose blocks, the camera extrinsic and td
    //       (compact column 6i+q = pose i, 6NP+q = ex_pose, 6NP+6 = td, 6NP+7 = right-hand-side slot), rounded up to 16
    // GO  = offset of the GNSS states in the state vector (rcv_dt 4NP, rcv_ddt NP, yaw 1, anc_ecef 3), 0: handle built without GNSS
    // NG  = capacity of GnssPsrDoppFactor per window; NGRP = capacity of (frame, lower_idx) groups of them
    // NC  = 6 NP + 8 compact columns of the visual system Vc (pose blocks, camera extrinsic, td, right-hand side: the layout of the compact
    //       rows above); NVC = stride of one packed lower triangle NC (NC + 1) / 2 (entry (RHS, RHS) is unused: the cost travels separately)
};
__host__ __device__ inline int off_pose(int i) { return 16 * i; }
__host__ __device__ inline int off_sb(int i) { return 16 * i + 7; }
__host__ __device__ inline int off_ex(int NP) { return 16 * NP; }
__host__ __device__ inline int off_exw(int NP) { return 16 * NP + 7; }
__host__ __device__ inline int off_ix(int NP) { return 16 * NP + 14; }
__host__ __device__ inline int off_td(int NP) { return 16 * NP + 17; }
__host__ __device__ inline int off_tdw(int NP) { return 16 * NP + 18; }
__host__ __device__ inline int off_feat(int NP) { return 16 * NP + 20; }
// f-block indices
__host__ __device__ inline int fb_pose(int i) { return 2 * i; }
__host__ __device__ inline int fb_sb(int i) { return 2 * i + 1; }
__host__ __device__ inline int fb_ex(int NP) { return 2 * NP; }
__host__ __device__ inline int fb_exw(int NP) { return 2 * NP + 1; }
__host__ __device__ inline int fb_sx(int NP) { return 2 * NP + 2; }
__host__ __device__ inline int fb_td(int NP) { return 2 * NP + 5; }
__host__ __device__ inline int fb_tdw(int NP) { return 2 * NP + 6; }
// GNSS f-blocks (only when Dims::GO > 0): rcv_dt[4 NP], rcv_ddt[NP], yaw (always constant in the solver), anc_ecef
__host__ __device__ inline int fb_rcvdt(int NP, int idx) { return 2 * NP + 7 + idx; }
__host__ __device__ inline int fb_rcvddt(int NP, int i) { return 6 * NP + 7 + i; }
__host__ __device__ inline int fb_yaw(int NP) { return 7 * NP + 7; }
__host__ __device__ inline int fb_anc(int NP) { return 7 * NP + 8; }

struct SolverState {  // per window, lives in device memory
    double radius, mu, alpha, x_cost, cand_cost, model_cost_change, dogleg_step_norm, gmax, initial_cost, x_norm, step_norm;
    int cur;            // which of the two state / normal-equation buffers is current
    int reuse, done, termination, iterations, successful, invalid_run, last_successful, have_scale, cand_valid;
    int R, NE;          // reduced dimension and number of eliminated (free inverse-depth) columns
    double mu_solved;   // the mu of the Gauss-Newton solve that gn / yv currently hold
    int chain;          // this window's ba_step runs in the chain form (ba_step_chain; set by the host when the window is packed: standard column layout, no GNSS blocks)
    int h_prior[2];     // buffer set q holds the prior's part of H for this solve's column map (written by its first linearisation: the part outside the band
                        // the IMU / wheel factors touch does not depend on the state, and rewriting 150 KB per window and iteration made the sweep write-bound)
};

struct Win {  // device view of the whole batch
    Dims d;
    double* xs;               // [2][B][XS] states: current / candidate
    const int* colf;          // [B][NFB] column of each non-feature block or -1 (constant)
    const int* cole;          // [B][F]   index of each free feature among the eliminated columns or -1
    const int* nvis; const int* nimu; const int* nwh; const int* nfeat;   // [B]
    const int* vis_idx;       // [B][NV] per factor: feature << 10 | frame i << 5 | frame j (one word instead of three tables: 8 bytes per factor less to upload every frame)
    const double* vis_data;   // [B][NV][5] per factor: pts_j.x, pts_j.y, vel_j(2), td_j -- the observation in frame j (its z never enters the residual: estimator.cpp:3276-3290 passes normalised points)
    const double* feat_obs;   // [B][F][6] per feature: pts_i(3) vel_i(2) td_i -- the observation in the start frame, shared by all factors of the feature
                              // (ProjectionTwoFrameOneCamFactor is built from feature_per_frame[0], estimator.cpp:3276-3290): 96 -> 48 bytes per factor to
                              // upload and to stream in every sweep
    const int* order;         // [B][NVP] factor index sorted by (i,j) pair, pairs padded to even length with -1
    const int* norder;        // [B]
    const int* feat_ptr;      // [B][F+1] CSR: factors of each feature
    const int* vis_pos;       // [B][NV] position of every factor in its feature's list: row of efac
    int pos_ident;            // every resident window lists its factors feature by feature (vis_pos[k] == k: what Estimator::optimization() builds): vis_pos is neither uploaded nor read
    const int* imu_i; const double* imu_data;   // [B][W], [B][W][IMU_STRIDE]
    const int* wh_i; const double* wh_data;     // [B][W], [B][W][WH_STRIDE]
    double* imu_sqrt; double* wh_sqrt;          // [B][W][225], [B][W][36]
    const int* pri_n; const int* pri_nb; const int* pri_bid;  // [B], [B], [B][64]
    const double* pri_J; const double* pri_r; const double* pri_x0;  // [B][NPRI*NPRI], [B][NPRI], [B][NPRI*2]
    double* pri_A; double* pri_b; double* pri_c;  // J0^T J0, J0^T r0, r0^T r0
    double* H;                // [2][B][RP*RP] prior + IMU + wheel (+ GNSS) part of the normal equations, lower triangle of the first R rows (ba_linearize_misc_win)
    double* g;                // [2][B][RP]    the same part of J^T r
    double* Vc;               // [2][B][NVC]   visual part, compact columns, packed lower triangle, row RHS = J^T r (ba_linearize_visual_win)
    double* cost;             // [3][2][B]     cost parts: prior + IMU + wheel, visual, GNSS; added in this order
    double* efac;             // [2][B][NV][EF] per-factor products of the eliminated column, rows in feature-list order (vis_pos)
    SolverState* st;          // [B]
    const double* wpar;       // [B][WPAR] per window: gravity G (3), visual sqrt_info, PoseSubsetParameterization masks of the camera / wheel extrinsic (as doubles)
    long long* stamps;        // optional phase timestamps of window 0 (profiling builds, -DGF_PROFILE_STEP)
    double* vpair;            // [B][NP (NP - 1) / 2][VPG] per frame pair (i < j): what every factor of the pair shares (vis_pair_geo), written by the sweep's own block before it evaluates
    double* vrows;            // [B][NVP][2][16] block rows of the visual factors in HBM (split formulation only: ba_linearize_visual_win MODE 1 / 2); else null
    double* vtile;            // global home of the visual sweep's pair tiles when they do not fit LDS ([B][vtile_stride]); else null
    size_t vtile_stride;
    // GNSS (Dims::GO > 0)
    const int* ngnss;         // [B]
    const int* gn_idx;        // [B][NG][4]: frame i, lower_idx, sys_idx, 0
    const double* gn_data;    // [B][NG][GN_STRIDE]: 16 values of gf_ba_window::gnss_data, ratio
    const double* gn_misc;    // [B][GN_MISC]: iono 8, ddt_weight, anchor 7, enabled, in_solve (= !lowspeed), has_anchor, Headers[NP]
    const int* gn_gptr;       // [B][NGRP + 2]: GnssPsrDoppFactors grouped by (frame, lower_idx): start of each group in gn_gitem; last slot = number of groups
    const int* gn_gitem;      // [B][NG] factor indices in group order
    double* gn_rows;          // [B][NG][38] scratch: residual (2) and Jacobian rows (2 x 18) of every GnssPsrDoppFactor
};
constexpr int WPAR = 6;
constexpr int GN_STRIDE = 18, GN_MISC = 20;   // gn_misc is GN_MISC + NP doubles per window
constexpr int IMU_STRIDE = 16 + 225 + 225;  // sum_dt, dp3, dq4, dv3, lba3, lbg3(=17 used incl. sum_dt -> 0..16) jac, cov
constexpr int IMU_JAC = 17, IMU_COV = 17 + 225;
constexpr int IMU_STRIDE2 = 17 + 450;
constexpr int WH_STRIDE = 1 + 3 + 4 + 18 + 36 + 4 + 12;  // sum_dt, dp, dq, jac(6x3), cov(6x6), lin(4), lin_vel, lin_gyr, vel_1, gyr_1
// per-factor eliminated-column products, one scratch buffer per batch (written and consumed inside one ba_linearize_visual_win launch).
//   EX (camera extrinsic carries columns): 24 doubles = Jd^T[Ji(6) Jj(6) Jtd(1) Jex(6)], Jd^T Jd, Jd^T r, then the factor's frames i, j (as doubles)
//   else (round 5): 10 doubles = 80 bytes: Jd^T[Ji(6) Jtd(1)], Jd^T Jd, Jd^T r, frames (i, j) as two ints in the last slot.  The six products Jd^T Jj are unique to their factor
//         (its second frame): the evaluating lane stores them straight into the feature's E^T F row instead of parking them here for et_rows8 to copy (rounds 3-4: 16 doubles,
//         one 128-byte line; these products were 98 of the ~130 MB a launch of 256 windows moves).
// The sweeps are bound by memory traffic as much as by arithmetic (all 256 windows of a launch move their tables at once: ~3.7 TB/s over the launch); the
// products were 2 x 73 MB (double-buffered, 192 B per factor) against a 256 MB Infinity Cache; now 49 MB.
constexpr int EF = 24;
template <bool EX> __host__ __device__ constexpr int ef_stride() { return EX ? 16 : 10; }
#ifdef GF_PROFILE_STEP
#define GF_WSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && w.stamps) w.stamps[i] = clock64(); } while (0)
#define GF_WSTAMP_T(t, i) do { if (blockIdx.x == 0 && threadIdx.x == (t) && w.stamps) w.stamps[i] = clock64(); } while (0)
#else
#define GF_WSTAMP(i) do { } while (0)
#define GF_WSTAMP_T(t, i) do { } while (0)
#endif
__device__ __forceinline__ double* cost_part(const Win& w, int part, int which, int b) { return w.cost + ((size_t)(part * 2 + which) * w.d.B + b); }
__device__ __forceinline__ double cost_total(const Win& w, int which, int b) { return (*cost_part(w, 0, which, b) + *cost_part(w, 1, which, b)) + *cost_part(w, 2, which, b); }

__device__ __forceinline__ Q4 q_of(const double* p) { return Q4{p[6], p[3], p[4], p[5]}; }
__device__ __forceinline__ V3 p_of(const double* p) { return V3{p[0], p[1], p[2]}; }

// A value every lane of the wavefront holds anyway (loaded from per-window state): tell the compiler, so that it lives in scalar registers, loop bounds and
// branches on it are scalar, and pointers derived from it stay out of the vector register file (ba_step sits at the 256-VGPR limit).
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni(double v) { return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v))); }
// Huber(1.0) + ceres Corrector (loss_function.h / corrector.cc; same arithmetic at marginalization_factor.cpp:27-57)
__device__ __forceinline__ void huber_corrector(double sq, double& rho0, double& sqrt_rho1, double& residual_scaling, double& alpha_sq_norm) {
    double rho1, rho2;
    if (sq > 1.0) { const double r = sqrt(sq); rho0 = 2.0 * r - 1.0; rho1 = fmax(2.2250738585072014e-308, 1.0 / r); rho2 = -rho1 / (2.0 * sq); }
    else { rho0 = sq; rho1 = 1.0; rho2 = 0.0; }
    sqrt_rho1 = sqrt(rho1);
    if (sq == 0.0 || rho2 <= 0.0) { residual_scaling = sqrt_rho1; alpha_sq_norm = 0.0; }
    else { const double D = 1.0 + 2.0 * sq * rho2 / rho1, alpha = 1.0 - sqrt(D); residual_scaling = sqrt_rho1 / (1 - alpha); alpha_sq_norm = alpha / sq; }
}

// ProjectionTwoFrameOneCamFactor::Evaluate (projectionTwoFrameOneCamFactor.cpp:43-151).  row[r][c]: c 0-5 pose_i, 6-11 pose_j, 12 td,
// 13 residual, 16-21 ex_pose; jd[r] = d r / d inv_depth.  want_jac=false: residual only.
struct VisEval { double row[2][22]; double jd[2]; };
__device__ __forceinline__ void visual_eval(const double* Pi_, const double* Pj_, const double* Ex_, double inv_dep_i, double td, const double* vd,
                                            double si, bool want_jac, VisEval& o) {
    const V3 Pi = p_of(Pi_), Pj = p_of(Pj_), tic = p_of(Ex_);
    const Q4 Qi = q_of(Pi_), Qj = q_of(Pj_), qic = q_of(Ex_);
    const V3 pts_i = v3(vd[0], vd[1], vd[2]), pts_j = v3(vd[3], vd[4], vd[5]), vel_i = v3(vd[6], vd[7], 0), vel_j = v3(vd[8], vd[9], 0);
    const double td_i = vd[10], td_j = vd[11];
    const V3 pts_i_td = pts_i - vel_i * (td - td_i), pts_j_td = pts_j - vel_j * (td - td_j);
    // One reciprocal per denominator (inverse depth, depth in frame j) and multiplications from there on: an IEEE double division costs ~25 instructions and this
    // function divided twelve times by one of these two numbers (a quarter of the sweep's arithmetic).  The results move by an ulp; the parity bars are tolerances.
    const double dep_i = 1.0 / inv_dep_i;
    const V3 pts_camera_i = pts_i_td * dep_i;
    const V3 pts_imu_i = qrot(qic, pts_camera_i) + tic;
    const V3 pts_w = qrot(Qi, pts_imu_i) + Pi;
    const V3 pts_imu_j = qrot(qinverse(Qj), pts_w - Pj);
    const V3 pts_camera_j = qrot(qinverse(qic), pts_imu_j - tic);
    const double idj = 1.0 / pts_camera_j.z;
    o.row[0][13] = si * (pts_camera_j.x * idj - pts_j_td.x);
    o.row[1][13] = si * (pts_camera_j.y * idj - pts_j_td.y);
    if (!want_jac) return;
    const M3 Ri = qmat(Qi), Rj = qmat(Qj), ric = qmat(qic);
    const double r00 = si * idj, r02 = -si * pts_camera_j.x * (idj * idj), r11 = r00, r12 = -si * pts_camera_j.y * (idj * idj);
    auto red = [&](const M3& m, int c, double& o0, double& o1) { o0 = r00 * m.m[c] + r02 * m.m[6 + c]; o1 = r11 * m.m[3 + c] + r12 * m.m[6 + c]; };
    auto redv = [&](V3 v, double& o0, double& o1) { o0 = r00 * v.x + r02 * v.z; o1 = r11 * v.y + r12 * v.z; };
    const M3 A = transpose(ric) * transpose(Rj);
    const M3 ARi = A * Ri;
    const M3 Jir = ARi * (-skew(pts_imu_i));
    const M3 Jjr = transpose(ric) * skew(pts_imu_j);
    const M3 nA = -A;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        red(A, c, o.row[0][c], o.row[1][c]);
        red(Jir, c, o.row[0][3 + c], o.row[1][3 + c]);
        red(nA, c, o.row[0][6 + c], o.row[1][6 + c]);
        red(Jjr, c, o.row[0][9 + c], o.row[1][9 + c]);
    }
    const M3 tmp_r = ARi * ric;
    {
        const M3 Jep = transpose(ric) * (transpose(Rj) * Ri - m3_identity());
        const M3 Jer = (-tmp_r) * skew(pts_camera_i) + skew(tmp_r * pts_camera_i) + skew(transpose(ric) * (transpose(Rj) * (Ri * tic + Pi - Pj) - tic));
#pragma unroll
        for (int c = 0; c < 3; c++) { red(Jep, c, o.row[0][16 + c], o.row[1][16 + c]); red(Jer, c, o.row[0][19 + c], o.row[1][19 + c]); }
    }
    {
        double a0, a1;
        redv(tmp_r * pts_i_td, a0, a1);
        const double f = -(dep_i * dep_i);
        o.jd[0] = a0 * f; o.jd[1] = a1 * f;
        redv(tmp_r * vel_i, a0, a1);
        o.row[0][12] = a0 * dep_i * -1.0 + si * vel_j.x;
        o.row[1][12] = a1 * dep_i * -1.0 + si * vel_j.y;
    }
}

// ---- the part of ProjectionTwoFrameOneCamFactor::Evaluate that depends on the frame pair only (round 4).  With X = pts_camera_i the chain of
// projectionTwoFrameOneCamFactor.cpp:60-66 is  pts_camera_j = M X + t,  M = ric^T Rj^T Ri ric (`tmp_r`, :119),  t = ric^T (Rj^T (Ri tic + Pi - Pj) - tic),  and the Jacobian
// blocks are built from  A = ric^T Rj^T (:88-91),  A Ri (:92),  M,  ric, tic  and -- with a free camera extrinsic -- ric^T (Rj^T Ri - I) (:111) and t (:115-118).  The factor
// list is pair-sorted, so the sweep's block computes these once per pair (55 pairs at W = 10 against ~1500 factors) into w.vpair and a lane evaluates its factor from the table:
// four quaternion rotations, three quaternion-to-matrix conversions and five 3 x 3 products per factor become one matrix-vector product, and the 2 x 3 projection rows are
// folded into the pair's matrices before the products with the point (q (-[v]x) = v x q), not after them.
constexpr int VPG = 40;   // A 9, A Ri 9, M 9, t 3, ric^T (Rj^T Ri - I) 9, pad 1
__device__ __forceinline__ void vis_pair_geo(const double* Pi_, const double* Pj_, const double* Ex_, double* out) {
    const V3 Pi = p_of(Pi_), Pj = p_of(Pj_), tic = p_of(Ex_);
    const M3 Ri = qmat(q_of(Pi_)), Rj = qmat(q_of(Pj_)), ric = qmat(q_of(Ex_));
    const M3 RjT = transpose(Rj), ricT = transpose(ric);
    const M3 A = ricT * RjT, ARi = A * Ri, M = ARi * ric;
    const V3 t = ricT * (RjT * (Ri * tic + Pi - Pj) - tic);
    const M3 Jep = ricT * (RjT * Ri - m3_identity());
#pragma unroll
    for (int k = 0; k < 9; k++) { out[k] = A.m[k]; out[9 + k] = ARi.m[k]; out[18 + k] = M.m[k]; out[30 + k] = Jep.m[k]; }
    out[27] = t.x; out[28] = t.y; out[29] = t.z; out[39] = 0.0;
}
__device__ __forceinline__ V3 cross3(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// the same outputs as visual_eval (want_jac), from the pair table `pg`, the window's ric / tic and the factor's own data
template <bool EX>
__device__ __forceinline__ void visual_eval_pg(const double* pg, const M3& ric, V3 tic, double inv_dep_i, double td, const double* vd, double si, VisEval& o) {
    const V3 pts_i = v3(vd[0], vd[1], vd[2]), vel_i = v3(vd[6], vd[7], 0);
    const double td_i = vd[10], td_j = vd[11];
    const V3 pts_i_td = pts_i - vel_i * (td - td_i);
    const double pjx = vd[3] - vd[8] * (td - td_j), pjy = vd[4] - vd[9] * (td - td_j);
    const double dep_i = 1.0 / inv_dep_i;
    const V3 X = pts_i_td * dep_i;                                   // pts_camera_i
    auto row = [&](int base, int r) { return V3{pg[base + 3 * r], pg[base + 3 * r + 1], pg[base + 3 * r + 2]}; };
    auto dot = [](V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; };
    const V3 M0 = row(18, 0), M1 = row(18, 1), M2 = row(18, 2);
    const V3 pc = V3{dot(M0, X) + pg[27], dot(M1, X) + pg[28], dot(M2, X) + pg[29]};   // pts_camera_j
    const double idj = 1.0 / pc.z;
    o.row[0][13] = si * (pc.x * idj - pjx);
    o.row[1][13] = si * (pc.y * idj - pjy);
    const double r00 = si * idj, r02 = -si * pc.x * (idj * idj), r12 = -si * pc.y * (idj * idj);
    // q0 / q1 = the two rows of `reduce` times a 3 x 3 matrix B: r00 B[0] + r02 B[2], r00 B[1] + r12 B[2]
    auto fold = [&](V3 b0, V3 b1, V3 b2, V3& q0, V3& q1) { q0 = b0 * r00 + b2 * r02; q1 = b1 * r00 + b2 * r12; };
    V3 a0, a1, b0, b1, m0, m1, c0, c1;
    fold(row(0, 0), row(0, 1), row(0, 2), a0, a1);                  // reduce * A           -> pose_i position, pose_j position (negated)
    fold(row(9, 0), row(9, 1), row(9, 2), b0, b1);                  // reduce * A Ri        -> pose_i rotation via -[pts_imu_i]x
    fold(M0, M1, M2, m0, m1);                                       // reduce * M           -> inverse depth, td
    fold(V3{ric.m[0], ric.m[3], ric.m[6]}, V3{ric.m[1], ric.m[4], ric.m[7]}, V3{ric.m[2], ric.m[5], ric.m[8]}, c0, c1);   // reduce * ric^T -> pose_j rotation via [pts_imu_j]x
    const V3 pts_imu_i = ric * X + tic, pts_imu_j = ric * pc + tic;
    const V3 ji0 = cross3(pts_imu_i, b0), ji1 = cross3(pts_imu_i, b1);   // q (-[v]x) = v x q
    const V3 jj0 = cross3(c0, pts_imu_j), jj1 = cross3(c1, pts_imu_j);   // q [v]x = q x v
    o.row[0][0] = a0.x; o.row[0][1] = a0.y; o.row[0][2] = a0.z; o.row[1][0] = a1.x; o.row[1][1] = a1.y; o.row[1][2] = a1.z;
    o.row[0][3] = ji0.x; o.row[0][4] = ji0.y; o.row[0][5] = ji0.z; o.row[1][3] = ji1.x; o.row[1][4] = ji1.y; o.row[1][5] = ji1.z;
    o.row[0][6] = -a0.x; o.row[0][7] = -a0.y; o.row[0][8] = -a0.z; o.row[1][6] = -a1.x; o.row[1][7] = -a1.y; o.row[1][8] = -a1.z;
    o.row[0][9] = jj0.x; o.row[0][10] = jj0.y; o.row[0][11] = jj0.z; o.row[1][9] = jj1.x; o.row[1][10] = jj1.y; o.row[1][11] = jj1.z;
    {
        const double f = -(dep_i * dep_i);
        o.jd[0] = dot(m0, pts_i_td) * f; o.jd[1] = dot(m1, pts_i_td) * f;
        o.row[0][12] = dot(m0, vel_i) * dep_i * -1.0 + si * vd[8];
        o.row[1][12] = dot(m1, vel_i) * dep_i * -1.0 + si * vd[9];
    }
    if (EX) {
        V3 e0, e1;
        fold(row(30, 0), row(30, 1), row(30, 2), e0, e1);            // reduce * ric^T (Rj^T Ri - I)
        // Jer = -M [X]x + [M X]x + [t]x with M X + t = pc:  reduce Jer = X x m  +  rows of reduce times [pc]x  ( = (r00, 0, r02) x pc, (0, r00, r12) x pc )
        const V3 g0 = cross3(X, m0) + cross3(V3{r00, 0.0, r02}, pc), g1 = cross3(X, m1) + cross3(V3{0.0, r00, r12}, pc);
        o.row[0][16] = e0.x; o.row[0][17] = e0.y; o.row[0][18] = e0.z; o.row[1][16] = e1.x; o.row[1][17] = e1.y; o.row[1][18] = e1.z;
        o.row[0][19] = g0.x; o.row[0][20] = g0.y; o.row[0][21] = g0.z; o.row[1][19] = g1.x; o.row[1][20] = g1.y; o.row[1][21] = g1.z;
    }
}

// Sums over lanes without LDS round trips (__shfl_xor on a double is two ds_bpermute per step, six dependent steps per sum): four DPP steps inside each
// 16-lane row (xor 1, xor 2, half mirror, mirror -- every lane of the row ends with the row's sum), then the four row sums through v_readlane.  The order
// of the additions is fixed, so the result does not depend on timing; every lane returns the same value.
template <int CTRL> __device__ __forceinline__ double dpp_add_f64(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true), hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_sum16_f64(double v) {
    v = dpp_add_f64<0xB1>(v);    // quad_perm [1, 0, 3, 2]
    v = dpp_add_f64<0x4E>(v);    // quad_perm [2, 3, 0, 1]
    v = dpp_add_f64<0x141>(v);   // row_half_mirror
    v = dpp_add_f64<0x140>(v);   // row_mirror
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v = row_sum16_f64(v);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0)), r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32)), r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return (r0 + r1) + (r2 + r3);
}

// One visual factor per lane: residual, Jacobians, Huber correction, zeroed columns of constant blocks, and the products the Schur
// complement needs for a free inverse depth (stored to efac).  k < 0: padding lane (all zero).  Returns the factor's cost.
template <bool EX>
__device__ __forceinline__ double vis_lane_eval(const Win& w, const Dims& d, int b, int which, int k, const double* xs, const int* colf, VisEval& ev, int& fi, int& fj, double* etw = nullptr) {   // etw (!EX): this window's E^T F rows (every other column of a row is written by et_rows8)
    double cost = 0.0;
    int feat = 0;
    fi = 0; fj = 0;
    // a padding lane (k < 0) evaluates factor 0 like everybody else and is masked where it would leave a trace (cost, efac; the caller zeroes its staged rows):
    // cheaper than zero-filling 46 doubles in every lane and running the evaluation under an exec mask
    const bool live = k >= 0;
    k = max(k, 0);
#pragma unroll
    for (int c = 14; c < 16; c++) { ev.row[0][c] = 0.0; ev.row[1][c] = 0.0; }
    if (!EX) {
#pragma unroll
        for (int c = 16; c < 22; c++) { ev.row[0][c] = 0.0; ev.row[1][c] = 0.0; }
    }
    {
        const size_t kk = (size_t)b * d.NV + k;
        { const int pk_ = w.vis_idx[kk]; feat = pk_ >> 10; fi = (pk_ >> 5) & 31; fj = pk_ & 31; }
        double vd[12];
        {
            const double* fj5 = w.vis_data + kk * 5; const double* fi6 = w.feat_obs + ((size_t)b * d.F + feat) * 6;
            vd[0] = fi6[0]; vd[1] = fi6[1]; vd[2] = fi6[2]; vd[3] = fj5[0]; vd[4] = fj5[1]; vd[5] = 1.0; vd[6] = fi6[3]; vd[7] = fi6[4]; vd[8] = fj5[2]; vd[9] = fj5[3]; vd[10] = fi6[5]; vd[11] = fj5[4];
        }
#ifdef GF_VIS_PAIRGEO
        {
            const double* Ex_ = xs + off_ex(d.NP);
            visual_eval_pg<EX>(w.vpair + ((size_t)b * (d.NP * (d.NP - 1) / 2) + (size_t)(fj * (fj - 1) / 2 + fi)) * VPG, qmat(q_of(Ex_)), p_of(Ex_), xs[off_feat(d.NP) + feat], xs[off_td(d.NP)], vd,
                               w.wpar[WPAR * b + 3], ev);
        }
#else
        visual_eval(xs + off_pose(fi), xs + off_pose(fj), xs + off_ex(d.NP), xs[off_feat(d.NP) + feat], xs[off_td(d.NP)], vd,
                    w.wpar[WPAR * b + 3], true, ev);
#endif
        const double r0 = ev.row[0][13], r1 = ev.row[1][13];
        const double sq = r0 * r0 + r1 * r1;
        cost = live ? 0.5 * sq : 0.0;   // inlier (s <= 1): rho = s, rho' = 1, rho'' = 0 -- the corrector is the identity (sqrt_rho1 = residual_scaling = 1, alpha = 0)
        if (sq > 1.0) {    // outliers only: square roots, divisions and the rank-one correction of 22 columns (a converged window has next to none)
            double rho0, sqrt_rho1, rs, asn;
            huber_corrector(sq, rho0, sqrt_rho1, rs, asn);
            cost = live ? 0.5 * rho0 : 0.0;
            // J = sqrt_rho1 * (J - alpha_sq_norm * r * (r^T J)), r *= residual_scaling
#pragma unroll
            for (int c = 0; c < 22; c++) {
                if (c == 13 || c == 14 || c == 15) continue;
                const double rtj = r0 * ev.row[0][c] + r1 * ev.row[1][c];
                ev.row[0][c] = sqrt_rho1 * (ev.row[0][c] - asn * r0 * rtj);
                ev.row[1][c] = sqrt_rho1 * (ev.row[1][c] - asn * r1 * rtj);
            }
            const double rtj = r0 * ev.jd[0] + r1 * ev.jd[1];
            ev.jd[0] = sqrt_rho1 * (ev.jd[0] - asn * r0 * rtj);
            ev.jd[1] = sqrt_rho1 * (ev.jd[1] - asn * r1 * rtj);
            ev.row[0][13] = r0 * rs; ev.row[1][13] = r1 * rs;
        }
        // constant blocks contribute no columns
        if (colf[fb_pose(fi)] < 0) for (int c = 0; c < 6; c++) ev.row[0][c] = ev.row[1][c] = 0.0;
        if (colf[fb_pose(fj)] < 0) for (int c = 6; c < 12; c++) ev.row[0][c] = ev.row[1][c] = 0.0;
        if (colf[fb_td(d.NP)] < 0) ev.row[0][12] = ev.row[1][12] = 0.0;
        if (EX && colf[fb_ex(d.NP)] < 0) for (int c = 16; c < 22; c++) ev.row[0][c] = ev.row[1][c] = 0.0;
        // eliminated (free inverse depth) column: products needed by the Schur complement
        if (live && w.cole[(size_t)b * d.F + feat] >= 0) {
            double* ef = w.efac + ((size_t)b * d.NV * EF + (size_t)(w.pos_ident ? k : w.vis_pos[kk]) * ef_stride<EX>());   // the factors of a feature are contiguous
            const double ete_f = ev.jd[0] * ev.jd[0] + ev.jd[1] * ev.jd[1], etb_f = ev.jd[0] * ev.row[0][13] + ev.jd[1] * ev.row[1][13];
            if (EX) {   // round 6: one 128-byte line per factor as well -- pose i (summed), td, the six extrinsic products (summed), ete, etb, the frames; pose j goes straight into the row
#pragma unroll
                for (int c = 0; c < 6; c++) ef[c] = ev.jd[0] * ev.row[0][c] + ev.jd[1] * ev.row[1][c];
                ef[6] = ev.jd[0] * ev.row[0][12] + ev.jd[1] * ev.row[1][12];
#pragma unroll
                for (int c = 0; c < 6; c++) ef[7 + c] = ev.jd[0] * ev.row[0][16 + c] + ev.jd[1] * ev.row[1][16 + c];
                ef[13] = ete_f; ef[14] = etb_f; ef[15] = __hiloint2double(fj, fi);
                double* er = etw + (size_t)w.cole[(size_t)b * d.F + feat] * d.ECW + 6 * fj;
#pragma unroll
                for (int c = 0; c < 6; c++) er[c] = ev.jd[0] * ev.row[0][6 + c] + ev.jd[1] * ev.row[1][6 + c];
            } else {
#pragma unroll
                for (int c = 0; c < 6; c++) ef[c] = ev.jd[0] * ev.row[0][c] + ev.jd[1] * ev.row[1][c];                       // pose i: summed over the feature's factors (et_rows8)
                ef[6] = ev.jd[0] * ev.row[0][12] + ev.jd[1] * ev.row[1][12];                                                    // td: summed
                ef[7] = ete_f; ef[8] = etb_f; ef[9] = __hiloint2double(fj, fi);
                double* er = etw + (size_t)w.cole[(size_t)b * d.F + feat] * d.ECW + 6 * fj;                                      // pose j: this factor's alone, straight into its place
#pragma unroll
                for (int c = 0; c < 6; c++) er[c] = ev.jd[0] * ev.row[0][6 + c] + ev.jd[1] * ev.row[1][6 + c];
            }
        }
    }
    return cost;
}

// the cost of one visual factor and nothing else (round 6: the candidate of a solve's last iteration is only ever judged by its cost -- the step behind it accepts or
// rejects and the solve ends -- so its sweep evaluates residuals, not Jacobians: Ceres linearises that point in full as well and throws the result away)
template <bool EX>
__device__ __forceinline__ double vis_lane_cost(const Win& w, const Dims& d, int b, int k, const double* xs) {
    if (k < 0) return 0.0;
    const size_t kk = (size_t)b * d.NV + k;
    const int pk_ = w.vis_idx[kk], feat = pk_ >> 10, fi = (pk_ >> 5) & 31, fj = pk_ & 31;
    double vd[12];
    {
        const double* fj5 = w.vis_data + kk * 5; const double* fi6 = w.feat_obs + ((size_t)b * d.F + feat) * 6;
        vd[0] = fi6[0]; vd[1] = fi6[1]; vd[2] = fi6[2]; vd[3] = fj5[0]; vd[4] = fj5[1]; vd[5] = 1.0; vd[6] = fi6[3]; vd[7] = fi6[4]; vd[8] = fj5[2]; vd[9] = fj5[3]; vd[10] = fi6[5]; vd[11] = fj5[4];
    }
    VisEval ev;
    visual_eval(xs + off_pose(fi), xs + off_pose(fj), xs + off_ex(d.NP), xs[off_feat(d.NP) + feat], xs[off_td(d.NP)], vd, w.wpar[WPAR * b + 3], false, ev);
    const double r0 = ev.row[0][13], r1 = ev.row[1][13];
    const double sq = r0 * r0 + r1 * r1;
    return sq > 1.0 ? 0.5 * (2.0 * sqrt(sq) - 1.0) : 0.5 * sq;   // Huber(1.0): rho(s) = 2 sqrt(s) - 1 beyond s = 1 (huber_corrector's rho0)
}

// ---------------------------------------------------------------------------------------------------------------
// small dense helpers on LDS matrices, executed by one wavefront
__device__ inline void wave_inverse_spd_sqrt(double* M, double* T, int n, int lane) {
    // in: M (n x n row-major, symmetric positive definite covariance).  out: M = upper factor U with U^T U = M^-1, i.e.
    // LLT(M^-1).matrixL().transpose() (imu_factor.h:73, wheel_factor.h:85).  T: n x 2n scratch.  Gauss-Jordan with partial pivoting.
    for (int i = lane; i < n * 2 * n; i += 64) { const int r = i / (2 * n), c = i % (2 * n); T[i] = c < n ? M[r * n + c] : (c - n == r ? 1.0 : 0.0); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    for (int col = 0; col < n; col++) {
        int piv = col; double best = fabs(T[col * 2 * n + col]);
        for (int r = col + 1; r < n; r++) { const double v = fabs(T[r * 2 * n + col]); if (v > best) { best = v; piv = r; } }
        if (piv != col) for (int c = lane; c < 2 * n; c += 64) { const double t = T[col * 2 * n + c]; T[col * 2 * n + c] = T[piv * 2 * n + c]; T[piv * 2 * n + c] = t; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        const double dinv = 1.0 / T[col * 2 * n + col];
        // factors first (they are overwritten by the update)
        double fr[4];
        for (int q = 0; q < 4; q++) { const int r = lane / 16 * 4 + q; fr[q] = (r < n && r != col) ? T[r * 2 * n + col] * dinv : 0.0; }
        __builtin_amdgcn_wave_barrier();
        for (int c = lane & 15; c < 2 * n; c += 16) {
            const double pv = T[col * 2 * n + c];
            for (int q = 0; q < 4; q++) { const int r = lane / 16 * 4 + q; if (r < n && r != col) T[r * 2 * n + c] -= fr[q] * pv; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        for (int c = lane; c < 2 * n; c += 64) T[col * 2 * n + c] *= dinv;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
    // M <- inverse (right half), then Cholesky (lower, column by column), store transposed
    for (int i = lane; i < n * n; i += 64) M[i] = T[(i / n) * 2 * n + n + (i % n)];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    for (int j = 0; j < n; j++) {
        double dd = M[j * n + j];
        for (int k2 = 0; k2 < j; k2++) dd -= M[j * n + k2] * M[j * n + k2];
        dd = sqrt(dd);
        for (int i = j + 1 + lane; i < n; i += 64) {
            double s = M[i * n + j];
            for (int k2 = 0; k2 < j; k2++) s -= M[i * n + k2] * M[j * n + k2];
            M[i * n + j] = s / dd;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) M[j * n + j] = dd;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
    // U = L^T: write into T then back
    for (int i = lane; i < n * n; i += 64) { const int r = i / n, c = i % n; T[i] = c >= r ? M[c * n + r] : 0.0; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < n * n; i += 64) M[i] = T[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int fblock_of(int id, const Dims& d);
__host__ __device__ inline int lsize_kind(int kind);
__host__ __device__ inline int gsize_kind(int kind);

// One-time per solve: sqrt information of every IMU / wheel factor and the prior's normal-equation form.
// grid (B), 256 threads (4 wavefronts).
__global__ void __launch_bounds__(256) ba_setup(Win w) {
    __shared__ double sM[4][225];
    __shared__ double sT[4][450];
    const Dims d = w.d;
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int t = wave; t < w.nimu[b] + w.nwh[b]; t += 4) {
        if (t < w.nimu[b]) {
            const double* src = w.imu_data + ((size_t)b * d.W + t) * IMU_STRIDE2 + IMU_COV;
            for (int i = lane; i < 225; i += 64) sM[wave][i] = src[i];
            wave_inverse_spd_sqrt(sM[wave], sT[wave], 15, lane);
            double* dst = w.imu_sqrt + ((size_t)b * d.W + t) * 225;
            for (int i = lane; i < 225; i += 64) dst[i] = sM[wave][i];
        } else {
            const int k = t - w.nimu[b];
            const double* src = w.wh_data + ((size_t)b * d.W + k) * WH_STRIDE + 26;
            for (int i = lane; i < 36; i += 64) sM[wave][i] = src[i];
            wave_inverse_spd_sqrt(sM[wave], sT[wave], 6, lane);
            double* dst = w.wh_sqrt + ((size_t)b * d.W + k) * 36;
            for (int i = lane; i < 36; i += 64) dst[i] = sM[wave][i];
        }
    }
    // prior: A = J0^T J0, b0 = J0^T r0, c0 = r0^T r0
    const int n = w.pri_n[b];
    if (n > 0) {
        const double* J = w.pri_J + (size_t)b * d.NPRI * d.NPRI;
        const double* r = w.pri_r + (size_t)b * d.NPRI;
        double* A = w.pri_A + (size_t)b * d.NPRI * d.NPRI;
        for (int i = threadIdx.x; i < n * n; i += 256) {
            const int a = i / n, c = i % n;
            double s = 0;
            for (int k2 = 0; k2 < n; k2++) s += J[(size_t)k2 * n + a] * J[(size_t)k2 * n + c];
            A[i] = s;
        }
        for (int a = threadIdx.x; a < n; a += 256) { double s = 0; for (int k2 = 0; k2 < n; k2++) s += J[(size_t)k2 * n + a] * r[k2]; w.pri_b[(size_t)b * d.NPRI + a] = s; }
        if (threadIdx.x == 0) { double s = 0; for (int k2 = 0; k2 < n; k2++) s += r[k2] * r[k2]; w.pri_c[b] = s; }
    }
#ifndef GF_TEST_REINTRODUCE_PRI_C_READ
    else if (threadIdx.x == 0) w.pri_c[b] = 0.0;   // no prior: a defined value all the same (its one reader does not look at it then)
#endif
}

// IMUFactor::Evaluate (imu_factor.h:28-191) + IntegrationBase::evaluate (integration_base.h:169-195): raw (un-whitened) residual and Jacobian.
// Jraw: 15 x 30 row-major in LDS, columns [pose_i 6 | speedbias_i 9 | pose_j 6 | speedbias_j 9].  Executed redundantly by every lane; lane 0 stores.
// `writer`: this lane stores the raw residual (15) and the Jacobian blocks (15 x 30, row-major, into a zeroed Jraw); the arithmetic runs on
// every lane.  The caller zeroes Jraw before and fences after.
// ld: row stride of Jraw (>= 30), rs: stride of rraw.
__device__ inline void imu_raw(const double* Pi_, const double* SBi, const double* Pj_, const double* SBj, const double* dat, const double* G_, double* rraw,
                               double* Jraw, bool want_jac, bool writer, int ld = 30, int rs = 1) {
    const V3 Pi = p_of(Pi_), Pj = p_of(Pj_), Vi = v3(SBi[0], SBi[1], SBi[2]), Bai = v3(SBi[3], SBi[4], SBi[5]), Bgi = v3(SBi[6], SBi[7], SBi[8]);
    const V3 Vj = v3(SBj[0], SBj[1], SBj[2]), Baj = v3(SBj[3], SBj[4], SBj[5]), Bgj = v3(SBj[6], SBj[7], SBj[8]);
    const Q4 Qi = q_of(Pi_), Qj = q_of(Pj_);
    const double sum_dt = dat[0];
    const V3 delta_p = v3(dat[1], dat[2], dat[3]), delta_v = v3(dat[8], dat[9], dat[10]), lba = v3(dat[11], dat[12], dat[13]), lbg = v3(dat[14], dat[15], dat[16]);
    const Q4 delta_q = Q4{dat[4], dat[5], dat[6], dat[7]};
    const double* jac = dat + IMU_JAC;
    auto blk = [&](int r0, int c0) { M3 m; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) m.m[3 * r + c] = jac[(r0 + r) * 15 + c0 + c]; return m; };
    const M3 dp_dba = blk(0, 9), dp_dbg = blk(0, 12), dq_dbg = blk(3, 12), dv_dba = blk(6, 9), dv_dbg = blk(6, 12);
    const V3 G = v3(G_[0], G_[1], G_[2]);
    const V3 dba = Bai - lba, dbg = Bgi - lbg;
    const Q4 cdq = qmul(delta_q, deltaQ(dq_dbg * dbg));
    const V3 cdv = delta_v + dv_dba * dba + dv_dbg * dbg;
    const V3 cdp = delta_p + dp_dba * dba + dp_dbg * dbg;
    const Q4 Qi_inv = qinverse(Qi);
    const V3 t_p = qrot(Qi_inv, G * (0.5 * sum_dt * sum_dt) + Pj - Pi - Vi * sum_dt);
    const V3 t_v = qrot(Qi_inv, G * sum_dt + Vj - Vi);
    const V3 rp = t_p - cdp, rq = qvec(qmul(qinverse(cdq), qmul(Qi_inv, Qj))) * 2.0, rv = t_v - cdv, rba = Baj - Bai, rbg = Bgj - Bgi;
    if (writer) {
        const double rr_[15] = {rp.x, rp.y, rp.z, rq.x, rq.y, rq.z, rv.x, rv.y, rv.z, rba.x, rba.y, rba.z, rbg.x, rbg.y, rbg.z};
#pragma unroll
        for (int q = 0; q < 15; q++) rraw[q * rs] = rr_[q];
    }
    if (!want_jac) return;
    if (writer) {
        auto put = [&](int r0, int c0, const M3& m) { for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Jraw[(r0 + r) * ld + c0 + c] = m.m[3 * r + c]; };
        const M3 Rit = qmat(Qi_inv);
        // pose_i (cols 0..5)
        put(0, 0, -Rit);
        put(0, 3, skew(t_p));
        put(3, 3, -qleft_qright33(qmul(qinverse(Qj), Qi), cdq));
        put(6, 3, skew(t_v));
        // speedbias_i (cols 6..14)
        put(0, 6, (-Rit) * sum_dt);
        put(0, 9, -dp_dba); put(0, 12, -dp_dbg);
        put(3, 12, (-qleft33(qmul(qmul(qinverse(Qj), Qi), delta_q))) * dq_dbg);
        put(6, 6, -Rit);
        put(6, 9, -dv_dba); put(6, 12, -dv_dbg);
        put(9, 9, -m3_identity()); put(12, 12, -m3_identity());
        // pose_j (cols 15..20)
        put(0, 15, Rit);
        put(3, 18, qleft33(qmul(qmul(qinverse(cdq), Qi_inv), Qj)));
        // speedbias_j (cols 21..29)
        put(6, 21, Rit);
        put(9, 24, m3_identity()); put(12, 27, m3_identity());
    }
}

// WheelFactor::Evaluate (wheel_factor.h:28-247) + WheelIntegrationBase::evaluate (wheel_integration_base.h:180-219).
// Jraw: 6 x 22, columns [pose_i 6 | pose_j 6 | T_io 6 | sx | sy | sw | td_wheel].
// same conventions as imu_raw: residual (6), Jacobian 6 x 22 into a zeroed Jraw
// want_ix / want_td: the intrinsic (sx, sy, sw) and time-offset columns are asked for.  They are constant blocks in the shipped configurations
// (estimate_wheel_intrinsic: 0, estimate_td_wheel: 0), and their Jacobians are half of the factor's arithmetic (four right Jacobians, five exponentials):
// a pass whose column map drops them skips that half (the marginalisation passes, where no block is constant, evaluate everything).
__device__ inline void wheel_raw(const double* Pi_, const double* Pj_, const double* Ex, double sx, double sy, double sw, double td, const double* dat, double* rraw,
                                 double* Jraw, bool want_jac, bool writer, int ld = 22, int rs = 1, bool want_ix = true, bool want_td = true) {
    const V3 Pi = p_of(Pi_), Pj = p_of(Pj_), tio = p_of(Ex);
    const Q4 Qi = q_of(Pi_), Qj = q_of(Pj_), qio = q_of(Ex);
    const M3 sv = m3_diag(sx, sy, 1);
    const V3 delta_p = v3(dat[1], dat[2], dat[3]);
    const Q4 delta_q = Q4{dat[4], dat[5], dat[6], dat[7]};
    const double* jac = dat + 8;  // 6 x 3
    const double lsx = dat[62], lsy = dat[63], lsw = dat[64], ltd = dat[65];
    const V3 lin_vel = v3(dat[66], dat[67], dat[68]), lin_gyr = v3(dat[69], dat[70], dat[71]), vel_1 = v3(dat[72], dat[73], dat[74]), gyr_1 = v3(dat[75], dat[76], dat[77]);
    const V3 dp_dsx = v3(jac[0], jac[3], jac[6]), dp_dsy = v3(jac[1], jac[4], jac[7]), dp_dsw = v3(jac[2], jac[5], jac[8]), dq_dsw = v3(jac[11], jac[14], jac[17]);
    const double dsx = sx - lsx, dsy = sy - lsy, dsw = sw - lsw;
    const M3 Ri = qmat(Qi), Rj = qmat(Qj), rio = qmat(qio);
    const V3 cdp = delta_p + dp_dsx * dsx + dp_dsy * dsy + dp_dsw * dsw;
    const Q4 cdq = qnormalized(qmul(qnormalized(delta_q), so3_exp(dq_dsw * dsw)));
    const double dtd = td - ltd;
    const Q4 e_fw = so3_exp(lin_gyr * (sw * dtd));
    const Q4 dq_time = qnormalized(qmul(qnormalized(qmul(e_fw, cdq)), so3_exp(gyr_1 * (-sw * dtd))));
    const V3 dp_time = qmat(e_fw) * (sv * lin_vel * dtd + cdp - qrot(cdq, sv * vel_1 * dtd));
    const M3 Rio_t = transpose(Ri * rio);
    const V3 rp = Rio_t * (Rj * tio + Pj - Ri * tio - Pi) - dp_time;
    const Q4 Qio = qmul(Qi, qio);
    const V3 rr = so3_log(qmul(qmul(qmul(qinverse(dq_time), qinverse(Qio)), Qj), qio));
    if (writer) { rraw[0] = rp.x; rraw[rs] = rp.y; rraw[2 * rs] = rp.z; rraw[3 * rs] = rr.x; rraw[4 * rs] = rr.y; rraw[5 * rs] = rr.z; }
    if (!want_jac) return;
    if (writer) {
        auto put = [&](int r0, int c0, const M3& m) { for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Jraw[(r0 + r) * ld + c0 + c] = m.m[3 * r + c]; };
        auto putv = [&](int r0, int c0, V3 v) { Jraw[r0 * ld + c0] = v.x; Jraw[(r0 + 1) * ld + c0] = v.y; Jraw[(r0 + 2) * ld + c0] = v.z; };
        const M3 Jr_inv = rightJacobianInvSO3(rr);
        const M3 Jr_drdsw = rightJacobianSO3(dq_dsw * (sw - lsw));
        const M3 Rcq = qmat(cdq);
        const M3 Rqio_inv = qmat(qinverse(Qio));
        put(0, 0, -Rqio_inv);
        put(0, 3, Rio_t * (Ri * skew(tio)) + transpose(rio) * skew(transpose(Ri) * (Rj * tio + Pj - Ri * tio - Pi)));
        put(3, 3, -(Jr_inv * qmat(qmul(qinverse(qmul(Qj, qio)), Qi))));
        put(0, 6, Rqio_inv);
        put(0, 9, -(qmat(qmul(qinverse(Qio), Qj)) * skew(tio)));
        put(3, 9, Jr_inv * qmat(qinverse(qio)));
        put(0, 12, Rqio_inv * (Rj - Ri));
        put(0, 15, skew(qrot(qinverse(Qio), qrot(Qj, tio) + Pj - qrot(Qi, tio) - Pi)));
        put(3, 15, Jr_inv * (m3_identity() - qmat(qmul(qmul(qinverse(qmul(Qj, qio)), Qi), qio))));
        if (want_ix || want_td) {
            const V3 fcw = lin_gyr * (sw * dtd), fcv = sv * lin_vel * dtd, bcv = sv * vel_1 * dtd, bcw = gyr_1 * (sw * dtd);
            const M3 Jrtd = rightJacobianSO3(fcw), Jr_mtd = rightJacobianSO3(-fcw);
            const M3 I1 = m3_diag(1, 0, 0), I2 = m3_diag(0, 1, 0);
            const M3 Efw = qmat(so3_exp(fcw));
            const M3 Em = qmat(so3_exp(-rr)), Ebw = qmat(so3_exp(bcw)), Rcq_inv = qmat(qinverse(cdq));
            if (want_ix) {
                const M3 Efv = qmat(so3_exp(fcv));   // wheel_factor.h:199,211 use exp(forward_compensate_v): kept as is
                putv(0, 18, -(Efv * (I1 * lin_vel * dtd + dp_dsx - Rcq * (I1 * vel_1) * dtd)));
                putv(0, 19, -(Efv * (I2 * lin_vel * dtd + dp_dsy - Rcq * (I2 * vel_1) * dtd)));
                putv(0, 20, -(Efw * (dp_dsw - Rcq * skew(Jr_drdsw * dq_dsw) * (sv * vel_1) * dtd + skew(Jrtd * lin_gyr * dtd) * (fcv + cdp - qrot(cdq, bcv)))));
                putv(3, 20, -(Jr_inv * Em * Ebw * (Rcq_inv * (Jrtd * lin_gyr) * dtd + Jr_drdsw * dq_dsw)));
            }
            if (want_td) {
                putv(0, 21, -(Efw * (sv * lin_vel - Rcq * (sv * vel_1) + skew(Jrtd * lin_gyr * sw) * (fcv + cdp - Rcq * bcv))));
                putv(3, 21, -(Jr_inv * Em * (Ebw * Rcq_inv * (Jrtd * lin_gyr) * sw - Jr_mtd * gyr_1 * sw)));
            }
        }
    }
}

// dx of the marginalisation prior for one kept block (marginalization_factor.cpp:348-372)
__device__ __forceinline__ void prior_block_dx(int kind, const double* x, const double* x0, double* dx) {
    const int gs = gsize_kind(kind);
    if (gs != 7) { for (int i = 0; i < gs; i++) dx[i] = x[i] - x0[i]; return; }
    for (int i = 0; i < 3; i++) dx[i] = x[i] - x0[i];
    const Q4 dq = qmul(qinverse(Q4{x0[6], x0[3], x0[4], x0[5]}), Q4{x[6], x[3], x[4], x[5]});
    const double sgn = (dq.w >= 0) ? 2.0 : -2.0;
    dx[3] = sgn * dq.x; dx[4] = sgn * dq.y; dx[5] = sgn * dq.z;
}
__device__ __forceinline__ int state_off_of(int id, const Dims& d) {
    const int kind = id / 4096, i = id % 4096, NP = d.NP;
    switch (kind) {
        case 0: return off_pose(i); case 1: return off_sb(i); case 2: return off_ex(NP); case 3: return off_exw(NP);
        case 4: return off_ix(NP); case 5: return off_ix(NP) + 1; case 6: return off_ix(NP) + 2; case 7: return off_td(NP); case 8: return off_tdw(NP);
        case 10: return d.GO + i; case 11: return d.GO + 4 * NP + i; case 12: return d.GO + 5 * NP; case 13: return d.GO + 5 * NP + 1;
        default: return off_feat(NP) + i;
    }
}
__device__ __forceinline__ int fblock_of(int id, const Dims& d) {
    const int kind = id / 4096, i = id % 4096, NP = d.NP;
    switch (kind) {
        case 0: return fb_pose(i); case 1: return fb_sb(i); case 2: return fb_ex(NP); case 3: return fb_exw(NP);
        case 4: return fb_sx(NP); case 5: return fb_sx(NP) + 1; case 6: return fb_sx(NP) + 2; case 7: return fb_td(NP); case 8: return fb_tdw(NP);
        case 10: return d.GO ? fb_rcvdt(NP, i) : -1; case 11: return d.GO ? fb_rcvddt(NP, i) : -1; case 12: return d.GO ? fb_yaw(NP) : -1; case 13: return d.GO ? fb_anc(NP) : -1;
        default: return -1;
    }
}
__host__ __device__ inline int lsize_kind(int kind) { return (kind == 0 || kind == 2 || kind == 3) ? 6 : kind == 1 ? 9 : kind == 13 ? 3 : 1; }
__host__ __device__ inline int gsize_kind(int kind) { return (kind == 0 || kind == 2 || kind == 3) ? 7 : kind == 1 ? 9 : kind == 13 ? 3 : 1; }

// grid (2W + 1, B), 256 threads: block t < W evaluates IMU factor t (wavefront 0), W <= t < 2W wheel factor t - W, block 2W adds the prior.
// reduced-system column of every local column of an IMU factor (pose_i 6, speed-bias_i 9, pose_j 6, speed-bias_j 9) or a wheel factor
// (pose_i 6, pose_j 6, wheel extrinsic 6, sx, sy, sw, td_wheel); -1: constant block
__device__ __forceinline__ void misc_cols(bool is_imu, int i, const int* colf, int NP, int* scol, int lane) {
    const int j = i + 1;
    if (is_imu) {
        if (lane < 30) {
            const int blk = lane < 6 ? fb_pose(i) : lane < 15 ? fb_sb(i) : lane < 21 ? fb_pose(j) : fb_sb(j);
            const int o = lane < 6 ? lane : lane < 15 ? lane - 6 : lane < 21 ? lane - 15 : lane - 21;
            scol[lane] = colf[blk] >= 0 ? colf[blk] + o : -1;
        }
    } else if (lane < 22) {
        const int blk = lane < 6 ? fb_pose(i) : lane < 12 ? fb_pose(j) : lane < 18 ? fb_exw(NP) : lane < 21 ? fb_sx(NP) + (lane - 18) : fb_tdw(NP);
        const int o = lane < 6 ? lane : lane < 12 ? lane - 6 : lane < 18 ? lane - 12 : 0;
        scol[lane] = colf[blk] >= 0 ? colf[blk] + o : -1;
    }
}

// One wavefront: whiten a factor's block row with its upper-triangular square-root information S (imu_factor.h:73, wheel_factor.h:85:
// residual = S r, J = S J) and form [J | r]^T [J | r] on the matrix cores.  sJp: the factor's padded block row [J | r] in LDS, 4*KS x 33
// doubles, rows >= NRES and columns > NCOL zero.  W' = S [J | r] (KS MFMA steps x 2 column tiles), then W'^T W' (3 tiles): 20 MFMAs for an
// IMU factor.  sW: 4*KS x 33 scratch of this wavefront.  The result -- the packed lower triangle of the (NCOL + 1) x (NCOL + 1) tile
// [[J^T J, .], [r^T J, r^T r]], local column NCOL being the residual -- overwrites the block row in sJp.  Returns the factor's cost.
template <int NRES, int NCOL>
__device__ __forceinline__ double misc_mfma_tile(const double* Sg, double* sJp, double* sW, int lane) {
    constexpr int KS = (NRES + 3) / 4, LD = 33;
    const int lr = lane & 15, lk = lane >> 4;
    double sa[KS];
#pragma unroll
    for (int kk = 0; kk < KS; kk++) { const int k = 4 * kk + lk; sa[kk] = (lr < NRES && k < NRES) ? Sg[lr * NRES + k] : 0.0; }
    d4 w0 = {0, 0, 0, 0}, w1 = {0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < KS; kk++) {
        const double* row = sJp + (4 * kk + lk) * LD + lr;
        w0 = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[kk], row[0], w0, 0, 0, 0);
        w1 = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[kk], row[16], w1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = lk + 4 * r;
        if (row < 4 * KS) { sW[row * LD + lr] = w0[r]; sW[row * LD + 16 + lr] = w1[r]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();   // W' complete: the block row in sJp is dead
    d4 g00 = {0, 0, 0, 0}, g10 = {0, 0, 0, 0}, g11 = {0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < KS; kk++) {
        const double* row = sW + (4 * kk + lk) * LD + lr;
        const double a0 = row[0], a1 = row[16];
        g00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, g00, 0, 0, 0);
        g10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a0, g10, 0, 0, 0);
        g11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, g11, 0, 0, 0);
    }
    // element (a, b) of tile (ta, tb): local columns 16 ta + a, 16 tb + b; a = lk + 4 r, b = lr
    double cost2 = 0.0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int a = lk + 4 * r, la = 16 + a;
        if (lr <= a) sJp[a * (a + 1) / 2 + lr] = g00[r];
        if (la <= NCOL) {
            sJp[la * (la + 1) / 2 + lr] = g10[r];
            if (lr <= a) sJp[la * (la + 1) / 2 + 16 + lr] = g11[r];
            if (la == NCOL && 16 + lr == NCOL) cost2 = g11[r];
        }
    }
    return 0.5 * wave_sum_f64(cost2);
}

}  // namespace gfb

namespace gfb {

#ifdef GF_PROFILE_STEP
#define GF_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && sb.stamps) sb.stamps[i] = clock64(); } while (0)
#else
#define GF_STAMP(i) do { } while (0)
#endif

struct StepBufs {  // per-window global scratch of ba_step
    double* scale;  // [B][VS] Jacobi column scaling 1/(1+|J_c|)        (VS = RP + FP: reduced columns, then eliminated columns)
    double* diag;   // [B][VS] dogleg diagonal
    double* grad;   // [B][VS] scaled gradient / diag
    double* gn;     // [B][VS] Gauss-Newton step (dogleg space)
    double* step;   // [B][VS] last step (scaled space, after /diag)
    double* u;      // [B][VS] scratch
    double* Et;     // [B][FP][RP] E^T F rows of the eliminated columns (unscaled)
    double* Es;     // [B][FP][RP] scaled by s_e s_c / sqrt(ete~)
    double* ete;    // [B][FP]
    double* etb;    // [B][FP]
    double* rhs;    // [B][RP]
    double* yv;     // [B][VS]
    double* Sg;     // [B][SgStride] global home of the reduced system when it does not fit LDS (large windows); else null
    size_t SgStride;
    double* Mg;     // [B][MgStride] same for the kept system of the marginalisation
    size_t MgStride;
    double* Yg;     // [B][NP][YgStride] chain form of ba_step: the finished panel, factor inverse and coupling block of every speed-bias block (ch_y_stride); else null
    size_t YgStride;
    long long* stamps;  // optional phase timestamps of window 0 (builds with -DGF_PROFILE_STEP)
    int VS;
};

__device__ __forceinline__ int pk(int i, int j) { return i * (i + 1) / 2 + j; }  // packed lower, i >= j

// 512-thread block reductions through wavefront shuffles + 8 LDS partials (sred >= 64 doubles)
template <int NV_, int NWV = 8>
__device__ __forceinline__ void block_sum_n(double (&v)[NV_], double* sred, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int q = 0; q < NV_; q++) v[q] = wave_sum_f64(v[q]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NV_; q++) sred[wave * NV_ + q] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV_; q++) { double t = 0; for (int w8 = 0; w8 < NWV; w8++) t += sred[w8 * NV_ + q]; v[q] = t; }
}
template <int NWV = 8>
__device__ __forceinline__ double block_sum(double v, double* sred, int tid, int nthreads) { double a[1] = {v}; block_sum_n<1, NWV>(a, sred, tid); return a[0]; }
template <int NWV = 8>
__device__ __forceinline__ double block_max(double v, double* sred, int tid, int nthreads) {
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    __syncthreads();
    if (lane == 0) sred[wave] = v;
    __syncthreads();
    double t = sred[0];
    for (int w8 = 1; w8 < NWV; w8++) t = fmax(t, sred[w8]);
    return t;
}

__device__ __forceinline__ int tri_row(int t) {  // largest i with i(i+1)/2 <= t
    int i = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while (i * (i + 1) / 2 > t) i--;
    while ((i + 1) * (i + 2) / 2 <= t) i++;
    return i;
}
// Compact rows of E^T F, ete, etb of one eliminated (free inverse-depth) column, by 16 lanes (four features per wavefront): every factor of
// a feature shares its pose_i / ex / td entries (summed in the fixed order of the feature's factor list), its pose_j entries are unique.
// The factor products sit in efac in feature-list order, frames included: no index chasing, one contiguous read per feature.
// eight features per wavefront, eight lanes per feature (sixteen lanes per feature made 3.1 rounds of ~15 k cycles each at 150 features and 12 wavefronts:
// the rounds are load-latency bound, so fewer and fuller ones win)
template <bool EX>
__device__ __forceinline__ void et_rows8(const Win& w, const StepBufs& sb, const Dims& d, int b, int f0, int nfeat, const int* cole, const int* fptr, int which, int lane) {
    constexpr int EFS = ef_stride<EX>();
    const int sub = lane & 7, f = f0 + (lane >> 3);
    const int e = f < nfeat ? cole[f] : -1;
    const int p0 = e >= 0 ? fptr[f] : 0, p1 = e >= 0 ? fptr[f + 1] : 0;
    const double* efac = w.efac + (size_t)b * d.NV * EF;
    double* Et = sb.Et + (((size_t)which * d.B + b) * d.FP + max(e, 0)) * d.ECW;
    if (!EX) {   // the row already holds every factor's pose-j products (round 5: stored by the evaluating lanes); what is left are the sums -- and (round 6) the zeros:
        // the row is no longer cleared at the head of the kernel (with the tracks of a window alive 66 of its 80 columns were written twice, 16 MB of the launch's
        // 78 MB of writes), the eight lanes of the feature clear the columns of the frames that hold no observation of it, the extrinsic columns and the padding
        double acc = 0.0, accb = 0.0;
        int fi = 0;
        unsigned seen = 0;   // frames j with a factor of this feature: their six columns are written by the evaluating lanes
        constexpr int NF = 10;
        for (int pb = p0; pb < p1; pb += NF) {
            double v[NF], vb[NF], fr[NF];
#pragma unroll
            for (int q = 0; q < NF; q++) {
                const bool on = pb + q < p1;
                const double* row = efac + (size_t)(on ? pb + q : p0) * EFS;
                v[q] = on ? row[sub] : 0.0; vb[q] = (on && sub == 0) ? row[8] : 0.0;
                fr[q] = row[9];
            }
            fi = __double2loint(fr[0]);
#pragma unroll
            for (int q = 0; q < NF; q++) { if (pb + q >= p1) continue; acc += v[q]; accb += vb[q]; seen |= 1u << __double2hiint(fr[q]); }
        }
        if (e < 0) return;
        if (p1 > p0) seen |= 1u << fi;   // pose i: the six sums below
        for (int c = sub; c < d.ECW; c += 8) {
            const bool written = c < 6 * d.NP ? ((seen >> (c / 6)) & 1u) != 0 : c == 6 * d.NP + 6;
            if (!written) Et[c] = 0.0;
        }
        if (p1 > p0) {
            if (sub < 6) Et[6 * fi + sub] = acc;                               // pose i
        }
        if (sub == 6) Et[6 * d.NP + 6] = acc;                                  // td
        else if (sub == 7) sb.ete[((size_t)which * d.B + b) * d.FP + e] = acc;
        if (sub == 0) sb.etb[((size_t)which * d.B + b) * d.FP + e] = accb;
        return;
    }
    // EX (round 6: the layout of the other variant plus the extrinsic products -- before, a factor parked all 19 products here and this pass cleared the row and copied
    // the pose-j ones over): lane `sub` owns the products sub (0-5 pose_i, 6 td, 7 ex 0) and 8 + sub (8-12 ex 1-5, 13 ete, 14 etb; 15 holds the frames)
    double acc0 = 0.0, acc1 = 0.0;
    int fi = 0;
    unsigned seen = 0;
    constexpr int NF = 10;   // factors in flight (a track spans <= W frames)
    for (int pb = p0; pb < p1; pb += NF) {
        double v0[NF], v1[NF], fr[NF];
#pragma unroll
        for (int q = 0; q < NF; q++) {
            const bool on = pb + q < p1;
            const double* row = efac + (size_t)(on ? pb + q : p0) * EFS;
            v0[q] = on ? row[sub] : 0.0; v1[q] = (on && sub < 7) ? row[8 + sub] : 0.0; fr[q] = row[15];
        }
        fi = __double2loint(fr[0]);
#pragma unroll
        for (int q = 0; q < NF; q++) { if (pb + q >= p1) continue; acc0 += v0[q]; acc1 += v1[q]; seen |= 1u << __double2hiint(fr[q]); }
    }
    if (e < 0) return;
    if (p1 > p0) seen |= 1u << fi;
    for (int c = sub; c < d.ECW; c += 8) {
        const bool written = c < 6 * d.NP ? ((seen >> (c / 6)) & 1u) != 0 : c <= 6 * d.NP + 6;
        if (!written) Et[c] = 0.0;
    }
    if (sub < 6) { if (p1 > p0) Et[6 * fi + sub] = acc0; }                 // pose i
    else if (sub == 6) Et[6 * d.NP + 6] = acc0;                            // td
    else Et[6 * d.NP] = acc0;                                              // ex 0
    if (sub < 5) Et[6 * d.NP + 1 + sub] = acc1;                            // ex 1-5
    else if (sub == 5) sb.ete[((size_t)which * d.B + b) * d.FP + e] = acc1;
    else if (sub == 6) sb.etb[((size_t)which * d.B + b) * d.FP + e] = acc1;
}

// ---------------------------------------------------------------------------------------------------------------
// Window-level visual sweep: one block of NW wavefronts per window.  The pair-sorted factor list (gf_ba.hip packs it: factors of one
// frame pair (i, j) are contiguous, pairs padded to even length) is cut into NW contiguous ranges of 64-factor chunks, one per wavefront.
// A wavefront evaluates 64 factors at a time (one per lane), stages their block rows in LDS and contracts J^T J / J^T r of a frame pair on
// the matrix cores; the tile of a pair stays in the accumulators across chunk boundaries and is stored -- not added -- once, when the
// pair ends: into the pair's own slot, or, when the pair began in the previous wavefront's range, into this wavefront's continuation slot.
// Afterwards the continuation slots are folded into the pair slots in wavefront order, and every entry of the window's compact visual
// system Vc (columns 6 p + q of pose p, camera extrinsic, td, right-hand side; packed lower triangle) is summed by one thread over the
// pair tiles that touch it, in frame order.  No floating-point atomics: the result does not depend on timing.
// The kernel then builds the compact E^T F rows of the free inverse depths (et_row), which ba_step / ba_marg_finish eliminate.
// Slots: NP (NP - 1) / 2 + NW tiles of 14 x 14 (EX: 20 x 20) packed lower triangles, in LDS when they fit (W = 10), else in w.vtile.
constexpr int kVW = 12;               // wavefronts per block, fixed extrinsic (three per SIMD at the kernel's ~150 VGPRs)
constexpr int kVWX = 6;               // wavefronts per block when the camera extrinsic carries columns
constexpr int kVWM = 8;               // wavefronts per block of the MARGIN_OLD pass (extrinsic columns, pairs (0, j) only: see ba_linearize_visual_win)
__host__ __device__ inline size_t vwin_marg_slot_doubles(int NP) { return ((size_t)(NP - 1) + kVWM) * 210; }
constexpr int kVFP = 511;             // feature lists up to this many features are staged in LDS
__host__ __device__ constexpr int vwin_sg(bool ex) { return ex ? 16 : 32; }       // factors staged per pass
__host__ __device__ constexpr int vwin_lstr(bool ex) { return ex ? 65 : 33; }     // staging row stride: 2 rows x COLS + 1
__host__ __device__ constexpr int vwin_tn(bool ex) { return ex ? 210 : 105; }     // packed tile: 20 x 21 / 2, 14 x 15 / 2
__host__ __device__ inline size_t vwin_slot_doubles(int NP, bool ex) { return ((size_t)NP * (NP - 1) / 2 + (ex ? kVWX : kVW)) * vwin_tn(ex); }
// EX: the camera extrinsic block has columns (free in the solve, or a kept block of the marginalisation): second 16-column tile
// MODE (north_star's formulation as a measured alternative, GF_BA_SPLIT_JTJ / gf_ba_set_split_jtj): 0 = the fused kernel; 1 = the sweep alone -- every factor's block
// rows [J | r] (2 x 16 doubles, 256 B per factor, in the pair-sorted order so that the four rows of one MFMA are 512 consecutive bytes) go to w.vrows in HBM, with the
// per-factor products and the cost, and the kernel ends; 2 = the contraction alone -- reads the block rows back (one coalesced 8-byte load per lane and MFMA), J^T J /
// J^T r per frame pair on the matrix cores, then the tile reduction, Vc and the E^T F rows as in the fused kernel.  Same products in the same order: same bits.
template <bool EX, int NW, int MODE = 0>
__global__ void __launch_bounds__(64 * NW) ba_linearize_visual_win(Win w, StepBufs sb, int which, int which_state, int only_cand_valid) {
    static_assert(MODE == 0 || !EX, "the split formulation is built for the fixed-extrinsic tiles");
    constexpr int COLS = EX ? 32 : 16, LSTR = vwin_lstr(EX), SG = vwin_sg(EX), TN = vwin_tn(EX), NT = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) double v_dyn[];   // pair-tile slots (when they fit)
    __shared__ double Jst[NW * SG * LSTR];
    __shared__ int s_pr[NW][64];
    __shared__ double s_cost[NW];
    __shared__ int s_first[NW], s_last[NW], s_cont[NW];
    __shared__ int s_fptr[kVFP + 1];     // the features' factor lists (feat_ptr) when they fit
    const Dims d = w.d;
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, NP = d.NP;
    const SolverState& st = w.st[b];
    const int st_cur = uni(st.cur);   // per-window scalars: uniform by construction, kept in scalar registers
    if (uni(st.done) && only_cand_valid != 2) return;   // only_cand_valid == 2: marginalisation pass (runs on finished windows)
    if ((only_cand_valid == 1 || only_cand_valid == 3) && !uni(st.cand_valid)) return;
    const int n_order = uni(w.norder[b]);
    if (which < 0) which = 1 - st_cur;
    if (which_state == -2) which_state = st_cur; else if (which_state < 0) which_state = 1 - st_cur;
    const double* xs = w.xs + ((size_t)which_state * d.B + b) * d.XS;
    const int* colf = w.colf + (size_t)b * d.NFB;
    if (only_cand_valid == 3) {   // the last iteration's candidate: its cost only (same chunks, same order of the additions as the full sweep's cost)
        const int nchunks = (n_order + 63) / 64, cpw = (nchunks + NW - 1) / NW;
        const int c_lo = min(nchunks, wave * cpw), c_hi = min(nchunks, c_lo + cpw);
        const int* ord = w.order + (size_t)b * d.NVP;
        double cost = 0.0;
        for (int ch = c_lo; ch < c_hi; ch++) {
            const int entry = ch * 64 + lane;
            const int oe = entry < n_order ? ord[entry] : -1;
            cost += vis_lane_cost<EX>(w, d, b, oe < 0 ? -1 : (oe & 0xffff), xs);
        }
        cost = wave_sum_f64(cost);
        if (lane == 0) s_cost[wave] = cost;
        __syncthreads();
        if (tid == 0) { double c = 0; for (int q = 0; q < NW; q++) c += s_cost[q]; *cost_part(w, 1, which, b) = c; }
        return;
    }
    // MARGIN_OLD pass (only_cand_valid == 2: its factor order holds the factors that start in frame 0 and nothing else): the pairs are (0, j), j = 1 .. NP - 1 -- NP - 1 tile
    // slots instead of NP (NP - 1) / 2, which is what lets this pass run on kVWM = 8 wavefronts with its tiles in LDS at any window size (round 5; it ran on the six wavefronts
    // the 55 + 6 slots of the solve's variant leave room for: 105 us against 50 us for the twelve-wavefront sweep of a whole window)
    const bool pairs0 = EX && only_cand_valid == 2;
    const int NPAIR = pairs0 ? NP - 1 : NP * (NP - 1) / 2;
    auto slot_of = [&](int pi, int pj) -> int { return pairs0 ? pj - 1 : pj * (pj - 1) / 2 + pi; };
    double* slots = w.vtile ? w.vtile + (size_t)b * w.vtile_stride : v_dyn;
    double* bnd = slots + (size_t)NPAIR * TN;
    GF_WSTAMP(80);
    for (int i = tid; i < (NPAIR + NW) * TN; i += NT) slots[i] = 0.0;
    double* etw = sb.Et + ((size_t)which * d.B + b) * d.FP * d.ECW;   // this window's E^T F rows: filled by the evaluating lanes (pose j) and by et_rows8 (sums, and zeros everywhere else)
    if (d.F <= kVFP) for (int i = tid; i <= d.F; i += NT) s_fptr[i] = w.feat_ptr[(size_t)b * (d.F + 1) + i];
#ifdef GF_VIS_PAIRGEO
    if (MODE != 2) {   // what the factors of a frame pair share, once per pair (visible to the whole block behind the barrier below: same CU, same L1)
        for (int p = tid; p < NPAIR; p += NT) {
            int pj = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
            while (pj * (pj - 1) / 2 > p) pj--;
            while ((pj + 1) * pj / 2 <= p) pj++;
            const int pi = p - pj * (pj - 1) / 2;
            vis_pair_geo(xs + off_pose(pi), xs + off_pose(pj), xs + off_ex(NP), w.vpair + ((size_t)b * NPAIR + p) * VPG);
        }
        __threadfence_block();
    }
#endif
    // this wavefront's range of chunks, and the pair keys at its ends (a pair that straddles two ranges has a main and a continuation slot)
    const int nchunks = (n_order + 63) / 64, cpw = (nchunks + NW - 1) / NW;
    const int c_lo = min(nchunks, wave * cpw), c_hi = min(nchunks, c_lo + cpw);
    const int* ord = w.order + (size_t)b * d.NVP;
    {
        int kf = -1, kl = -1;
        if (c_lo < c_hi) {
            const int e0 = 64 * c_lo + lane, e1 = 64 * (c_hi - 1) + lane;
            const int o0 = e0 < n_order ? ord[e0] : -1, o1 = e1 < n_order ? ord[e1] : -1;
            const unsigned long long m0 = __ballot(o0 >= 0), m1 = __ballot(o1 >= 0);
            if (m0) kf = __shfl(o0, __ffsll((long long)m0) - 1) >> 16;
            if (m1) kl = __shfl(o1, 63 - __clzll((long long)m1)) >> 16;
        }
        if (lane == 0) { s_first[wave] = kf; s_last[wave] = kl; }
    }
    __syncthreads();
    const int first_key = s_first[wave];
    const bool continuing = wave > 0 && first_key >= 0 && s_last[wave - 1] == first_key;
    if (lane == 0) s_cont[wave] = continuing ? slot_of(first_key >> 6, first_key & 63) : -1;
    GF_WSTAMP(81);
    double* Jbuf = Jst + wave * SG * LSTR;
    int* s_pair = s_pr[wave];
    double cost = 0.0;
    d4 acc00 = {0, 0, 0, 0}, acc01 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
    int cur_pair = -1;
    auto flush = [&](int pair) {   // store the finished tile of `pair` (local columns: 0-5 pose_i, 6-11 pose_j, 12 td, 13 residual, 14-19 extrinsic)
        if (pair < 0) return;
        const int pi = pair >> 6, pj = pair & 63;
        double* dst = (continuing && pair == first_key) ? bnd + (size_t)wave * TN : slots + (size_t)slot_of(pi, pj) * TN;
        const int tb = lane & 15;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int ta = (lane >> 4) + 4 * r;
            if (ta < 14 && tb <= ta) dst[ta * (ta + 1) / 2 + tb] = acc00[r];
            if (EX) {
                if (ta < 14 && tb < 6) dst[(14 + tb) * (15 + tb) / 2 + ta] = acc01[r];          // extrinsic column tb x tile-0 column ta
                if (ta < 6 && tb <= ta) dst[(14 + ta) * (15 + ta) / 2 + 14 + tb] = acc11[r];
            }
        }
        acc00 = d4{0, 0, 0, 0}; acc01 = d4{0, 0, 0, 0}; acc11 = d4{0, 0, 0, 0};
    };
    for (int ch = c_lo; ch < c_hi; ch++) {
        const int entry = ch * 64 + lane;
        const int oe = entry < n_order ? ord[entry] : -1;
        const int k = oe < 0 ? -1 : (oe & 0xffff);
        int fi, fj;
        VisEval ev;
        if (ch == c_lo) GF_WSTAMP(92);
        if (MODE != 2) cost += vis_lane_eval<EX>(w, d, b, which, k, xs, colf, ev, fi, fj, etw);
        if (ch == c_lo) GF_WSTAMP(93);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();   // the previous chunk's reads of s_pair are done
        const int mykey = oe < 0 ? -1 : (oe >> 16);
        if (MODE == 2) s_pair[lane] = mykey;
        if (MODE == 1) {   // block rows of this lane's factor to HBM: rows[entry][r][16], columns 14 / 15 and padding lanes zero
            if (entry < ((n_order + 63) & ~63)) {
                double2* dst = reinterpret_cast<double2*>(w.vrows + ((size_t)b * ((d.NVP + 63) & ~63) + entry) * 32);   // windows 64 entries apart: a last, partly filled chunk stays inside its window's rows
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int c = 0; c < 16; c += 2) {
                        double2 v;
                        v.x = (c < 14 && k >= 0) ? ev.row[r][c] : 0.0; v.y = (c + 1 < 14 && k >= 0) ? ev.row[r][c + 1] : 0.0;
                        dst[(r * 16 + c) >> 1] = v;
                    }
            }
            continue;
        }
        if (MODE == 2) {   // the contraction alone: 32 MFMAs per chunk, operands straight from HBM (all loads of the chunk in flight before the first product)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
            const double* src = w.vrows + ((size_t)b * ((d.NVP + 63) & ~63) + (size_t)ch * 64) * 32 + lane;
            double av[32];
#pragma unroll
            for (int m = 0; m < 32; m++) av[m] = src[64 * m];
#pragma unroll
            for (int m = 0; m < 32; m++) {
                const int pair = uni(s_pair[2 * m]);
                if (pair < 0) continue;
                if (pair != cur_pair) { flush(cur_pair); cur_pair = pair; }
                acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m], av[m], acc00, 0, 0, 0);
            }
            continue;
        }
        // the 64 block rows go through the staging area SG factors at a time
#ifdef GF_PROFILE_STEP
        long long vq_t = clock64(), vq_stage = 0, vq_mfma = 0;
#endif
#pragma unroll
        for (int part = 0; part < 64 / SG; part++) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();   // the previous part's reads are done
            if (lane / SG == part) {
                const int l = lane % SG;
#pragma unroll
                for (int r = 0; r < 2; r++) {
#pragma unroll
                    for (int c = 0; c < 16; c++) Jbuf[l * LSTR + r * COLS + c] = (c < 14 && k >= 0) ? ev.row[r][c] : 0.0;
                    if (EX) {
#pragma unroll
                        for (int c = 0; c < 16; c++) Jbuf[l * LSTR + r * COLS + 16 + c] = (c < 6 && k >= 0) ? ev.row[r][16 + c] : 0.0;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
#ifdef GF_PROFILE_STEP
            { const long long t_ = clock64(); vq_stage += t_ - vq_t; vq_t = t_; }
#endif
            // The pair of entries 2m, 2m + 1 comes out of the lanes' own keys (a readlane; through s_pair it was an LDS round trip in front of every product), and the
            // operands of step m + 1 are loaded before the products of step m: what is left per step is the matrix core's own time (224 -> ~90 cycles per step).
            const int jb = ((lane >> 5) * LSTR) + ((lane >> 4) & 1) * COLS + (lane & 15);
            double a0n = Jbuf[jb], a1n = EX ? Jbuf[jb + 16] : 0.0;
#pragma unroll
            for (int m = 0; m < SG / 2; m++) {
                const double a0 = a0n, a1 = a1n;
                if (m + 1 < SG / 2) { a0n = Jbuf[2 * (m + 1) * LSTR + jb]; if (EX) a1n = Jbuf[2 * (m + 1) * LSTR + jb + 16]; }
                const int pair = __builtin_amdgcn_readlane(mykey, SG * part + 2 * m);   // pairs are padded to even length: entries 2m and 2m+1 share the pair (or are padding)
                if (pair < 0) continue;
                if (pair != cur_pair) { flush(cur_pair); cur_pair = pair; }
                acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, acc00, 0, 0, 0);
                if (EX) {
                    acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a1, acc01, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, acc11, 0, 0, 0);
                }
            }
#ifdef GF_PROFILE_STEP
            { const long long t_ = clock64(); vq_mfma += t_ - vq_t; vq_t = t_; }
#endif
        }
#ifdef GF_PROFILE_STEP
        if (ch == c_lo && blockIdx.x == 0 && threadIdx.x == 0 && w.stamps) { w.stamps[95] = vq_stage; w.stamps[96] = vq_mfma; }
#endif
    }
    GF_WSTAMP(94);
    flush(cur_pair);
    GF_WSTAMP(82);
    cost = wave_sum_f64(cost);
    if (lane == 0) s_cost[wave] = cost;
    __syncthreads();
    GF_WSTAMP(83);
    if (MODE != 2 && tid == 0) { double c = 0; for (int q = 0; q < NW; q++) c += s_cost[q]; *cost_part(w, 1, which, b) = c; }
    if (MODE == 1) return;
    constexpr int NWV = NW / 2;   // wavefronts that build the compact system; the other NW - NWV build the E^T F rows at the same time
    // ---- continuation slots -> pair slots, in wavefront order
    for (int t = tid; t < TN; t += NT)
        for (int ww = 1; ww < NW; ww++) { const int p = s_cont[ww]; if (p >= 0) slots[(size_t)p * TN + t] += bnd[(size_t)ww * TN + t]; }
    __syncthreads();
    GF_WSTAMP(84);
    // ---- the window's compact visual system: entry (ka >= kb) = sum over the pair tiles that hold both columns, in frame order
    {
        const int EXC = 6 * NP, TD = EXC + 6, RHS = EXC + 7, NCc = EXC + 8;
        double* Vc = w.Vc + ((size_t)which * d.B + b) * d.NVC;
        auto T = [&](int i, int j, int la, int lb) -> double {   // tile of pair (i < j), local entry (la, lb)
            const int hi = max(la, lb), lo = min(la, lb);
            if (pairs0 && i > 0) return 0.0;   // no such pair in a MARGIN_OLD pass
            return slots[(size_t)slot_of(i, j) * TN + hi * (hi + 1) / 2 + lo];
        };
        auto loc_other = [&](int k) -> int { return k < TD ? (EX ? 14 + (k - EXC) : -1) : k == TD ? 12 : 13; };   // local index of a non-pose column
        // Round 6: the compact system (LDS reads with computed addresses) and the E^T F rows below (global loads of the per-factor products) are both chains of latencies that
        // share nothing: the first NWV wavefronts build Vc while the others build the rows -- one phase under the other instead of one behind the other (same sums, same order).
        if (wave < NWV) for (int idx = tid; idx < NCc * (NCc + 1) / 2; idx += 64 * NWV) {
            const int ka = tri_row(idx), kb = idx - ka * (ka + 1) / 2;
            double s = 0.0;
            if (ka < EXC) {                                   // pose x pose (kb <= ka: also a pose column)
                const int fa = ka / 6, qa = ka - 6 * fa, fb = kb / 6, qb = kb - 6 * fb;
                if (fa == fb) {
                    for (int t = 0; t < NP; t++) {
                        if (t == fa) continue;
                        const int o = t > fa ? 0 : 6;          // fa is the pair's first frame (local 0-5) or its second (6-11)
                        s += T(min(t, fa), max(t, fa), o + qa, o + qb);
                    }
                } else s = T(fb, fa, 6 + qa, qb);             // fa > fb: only the pair (fb, fa)
            } else {
                const int la = loc_other(ka);
                if (la >= 0) {
                    if (kb < EXC) {                           // extrinsic / td / rhs x pose
                        const int fb = kb / 6, qb = kb - 6 * fb;
                        for (int t = 0; t < NP; t++) {
                            if (t == fb) continue;
                            s += T(min(t, fb), max(t, fb), la, (t > fb ? 0 : 6) + qb);
                        }
                    } else {
                        const int lb = loc_other(kb);
                        if (lb >= 0 && !(ka == RHS && kb == RHS)) for (int j = 1; j < NP; j++) for (int i = 0; i < j; i++) s += T(i, j, la, lb);
                    }
                }
            }
            Vc[idx] = s;
        }
    }
    GF_WSTAMP(85);
    // ---- compact E^T F rows of the free inverse depths (the factor products efac were written above by this block)
    {
        const int nfeat = uni(w.nfeat[b]);
        const int* cole = w.cole + (size_t)b * d.F;
        const int* fptr = d.F <= kVFP ? s_fptr : w.feat_ptr + (size_t)b * (d.F + 1);
        if (wave >= NWV) for (int f0 = 8 * (wave - NWV); f0 < nfeat; f0 += 8 * (NW - NWV)) et_rows8<EX>(w, sb, d, b, f0, nfeat, cole, fptr, which, lane);
    }
    GF_WSTAMP(86);
}

// ---------------------------------------------------------------------------------------------------------------
// Window-level prior / IMU / wheel sweep: one block of kMW wavefronts per window, the only writer of H and g.
//  phase 1  wavefront 0 evaluates the IMU factors (lane k = factor k: a factor evaluation is a ~10^4-instruction scalar program, so all
//           factors of the window cost the latency of one), wavefront 1 the wheel factors, into padded block rows [J | r] in LDS;
//           the other wavefronts meanwhile write the prior's part: H <- A gathered to this pass's columns (lower triangle of the
//           first R rows), g <- b0 + A dx, prior cost (marginalization_factor.cpp:344-392: r = r0 + J0 dx).
//  phase 2  whitening + [J | r]^T [J | r] of each factor on the matrix cores (misc_mfma_tile): one packed tile per factor, in LDS.
//  phase 3  every entry of H / g inside the band the factors touch (per frame: the 15 x 15 pose / speed-bias block and its coupling
//           to the next frame; the block all wheel factors share: wheel extrinsic, sx, sy, sw, td_wheel, and its rows against the poses)
//           is written by one thread as prior + the tiles that hold it, in factor order.  Costs are added in factor order.
// No floating-point atomics, no read-modify-write: every entry has one final writer and a fixed order of additions.
// frame_filter: 0 all factors; 1 only factors starting at frame 0 (MARGIN_OLD); 2 no IMU / wheel factor (MARGIN_SECOND_NEW).
// Dynamic LDS (doubles): W x 16 x 33 IMU block rows / tiles, W x 9 x 33 wheel block rows / tiles, nscr x 16 x 33 scratch.
constexpr int kMW = 8;
constexpr int kMJi = 16 * 33, kMJw = 9 * 33, kMScr = 16 * 33;
__host__ __device__ inline int misc_win_nscr(int W) { return W <= 12 ? kMW : 4; }   // whitening wavefronts (scratch areas)
__host__ __device__ inline size_t misc_win_rows_doubles(int W) { return (size_t)W * (kMJi + kMJw) + (size_t)misc_win_nscr(W) * kMScr; }
constexpr int kMTab = 64 + kMW + 2 * 512 + (2 * 512 + 2 * 32 + 2) / 2;   // the small tables behind the rows, in doubles
__host__ __device__ inline size_t misc_win_lds_doubles(int W) { return misc_win_rows_doubles(W) + kMTab; }
__device__ __forceinline__ void ba_linearize_misc_body(Win w, int which, int which_state, int only_cand_valid, int frame_filter) {
    extern __shared__ __attribute__((aligned(16))) double m_lds[];
    const Dims d = w.d;
    // the kernel's small tables sit behind the block rows in the dynamic area (no static LDS of its own: next to ba_step's 26 KB in ba_misc_step there is no room for it)
    double* s_fcost = m_lds + misc_win_rows_doubles(d.W);   // [64] per-factor costs: IMU k at k, wheel k at 32 + k
    double* s_wc = s_fcost + 64;                            // [kMW] prior cost partials of the wavefronts
    double* s_dx = s_wc + kMW; double* s_pg = s_dx + 512;   // [512] each, prior: dx, b0 + A dx by local prior index
    int* s_pcol = reinterpret_cast<int*>(s_pg + 512); int* s_pidx = s_pcol + 512;   // [512] each
    int* s_imu_at = s_pidx + 512; int* s_wh_at = s_imu_at + 32;                     // [32] each: factor starting at frame f (or -1)
    int& s_R = s_wh_at[32];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, RP = d.RP, NP = d.NP;
    const SolverState& st = w.st[b];
    const int st_cur = uni(st.cur);
    if (uni(st.done) && only_cand_valid != 2) return;
    const bool cost_only = only_cand_valid == 3;   // the last iteration's candidate: H and g of this point are never read (see vis_lane_cost)
    if ((only_cand_valid == 1 || cost_only) && !uni(st.cand_valid)) return;
    if (which < 0) which = 1 - st_cur;
    if (which_state == -2) which_state = st_cur; else if (which_state < 0) which_state = 1 - st_cur;
    const double* xs = w.xs + ((size_t)which_state * d.B + b) * d.XS;
    const int* colf = w.colf + (size_t)b * d.NFB;
    double* H = w.H + ((size_t)which * d.B + b) * RP * RP;
    double* g = w.g + ((size_t)which * d.B + b) * RP;
    const int nimu = frame_filter == 2 ? 0 : uni(w.nimu[b]), nwh = frame_filter == 2 ? 0 : uni(w.nwh[b]);
    // the prior's part of H outside the band is already in this buffer set (see SolverState::h_prior); not with GNSS blocks, whose kernel adds into H
    const bool h_keeps = only_cand_valid != 2 && frame_filter == 0 && d.GO == 0;
    const bool h_have = h_keeps && uni(w.st[b].h_prior[which]) != 0;
    const int nscr = misc_win_nscr(d.W);
    double* sJi = m_lds;                          // [W][16][33]
    double* sJw = sJi + (size_t)d.W * kMJi;       // [W][9][33]
    double* sW = sJw + (size_t)d.W * kMJw + (size_t)min(wave, nscr - 1) * kMScr;
    const int* imu_i = w.imu_i + (size_t)b * d.W; const int* wh_i = w.wh_i + (size_t)b * d.W;
    auto imu_on = [&](int k) -> bool { return frame_filter != 1 || imu_i[k] == 0; };
    auto wh_on = [&](int k) -> bool { return frame_filter != 1 || wh_i[k] == 0; };
    GF_WSTAMP(64);
    for (int q = tid; q < nimu * kMJi; q += 64 * kMW) sJi[q] = 0.0;
    for (int q = tid; q < nwh * kMJw; q += 64 * kMW) sJw[q] = 0.0;
    if (tid < 64) s_fcost[tid] = 0.0;
    if (tid < kMW) s_wc[tid] = 0.0;
    if (tid < 32) { s_imu_at[tid] = -1; s_wh_at[tid] = -1; }
    for (int q = tid; q < 512; q += 64 * kMW) { s_pidx[q] = -1; s_pcol[q] = -1; }
    if (tid == 0) {   // number of columns of this pass: blocks with a column are laid out contiguously from 0
        int R = 0;
        for (int q = 0; q < d.NFB; q++) {
            if (colf[q] < 0) continue;
            const int ls = q < 2 * NP ? ((q & 1) ? 9 : 6) : q < 2 * NP + 2 ? 6 : (d.GO && q == fb_anc(NP)) ? 3 : 1;
            R = max(R, colf[q] + ls);
        }
        s_R = R;
    }
    __syncthreads();
    const int R = uni(s_R), n = uni(w.pri_n[b]);
    {   // prior: dx of every kept block (marginalization_factor.cpp:348-372), column of every local prior index, and the inverse map
        const int nb = n > 0 ? w.pri_nb[b] : 0;
        const int* bid = w.pri_bid + (size_t)b * 64;
        const double* x0 = w.pri_x0 + (size_t)b * d.NPRI * 2;
        if (tid < nb) {
            int idx = 0, o0 = 0;
            for (int q = 0; q < tid; q++) { idx += lsize_kind(bid[q] / 4096); o0 += gsize_kind(bid[q] / 4096); }
            const int id = bid[tid], kind = id / 4096;
            double dx[9];
            prior_block_dx(kind, xs + state_off_of(id, d), x0 + o0, dx);
            const int fb = fblock_of(id, d);
            const int c0 = fb >= 0 ? colf[fb] : -1;
            for (int q = 0; q < lsize_kind(kind); q++) { s_dx[idx + q] = dx[q]; s_pcol[idx + q] = c0 >= 0 ? c0 + q : -1; if (c0 >= 0) s_pidx[c0 + q] = idx + q; }
        }
        if (tid >= 64 && tid < 64 + nimu && imu_on(tid - 64)) s_imu_at[imu_i[tid - 64]] = tid - 64;
        if (tid >= 128 && tid < 128 + nwh && wh_on(tid - 128)) s_wh_at[wh_i[tid - 128]] = tid - 128;
    }
    __syncthreads();
    GF_WSTAMP(65);
    const double* A = w.pri_A + (size_t)b * d.NPRI * d.NPRI;
    // ---- phase 1: factor evaluation (wavefronts 0, 1) next to the prior's part of H, g, cost (wavefronts 2..)
    if (wave == 0) {
        if (lane < nimu && imu_on(lane)) {
            const int k = lane, i = imu_i[k], j = i + 1;
            imu_raw(xs + off_pose(i), xs + off_sb(i), xs + off_pose(j), xs + off_sb(j), w.imu_data + ((size_t)b * d.W + k) * IMU_STRIDE2, w.wpar + WPAR * b, sJi + kMJi * k + 30,
                    sJi + kMJi * k, true, true, 33, 33);
        }
        GF_WSTAMP_T(0, 66);
    } else if (wave == 1) {
        if (lane < nwh && wh_on(lane)) {
            const int k = lane, i = wh_i[k], j = i + 1;
            wheel_raw(xs + off_pose(i), xs + off_pose(j), xs + off_exw(NP), xs[off_ix(NP)], xs[off_ix(NP) + 1], xs[off_ix(NP) + 2], xs[off_tdw(NP)],
                      w.wh_data + ((size_t)b * d.W + k) * WH_STRIDE, sJw + kMJw * k + 22, sJw + kMJw * k, true, true, 33, 33,
                      colf[fb_sx(NP)] >= 0 || colf[fb_sx(NP) + 1] >= 0 || colf[fb_sx(NP) + 2] >= 0, colf[fb_tdw(NP)] >= 0);
        }
        GF_WSTAMP_T(64, 67);
    } else {
        const int t6 = tid - 128, NT6 = 64 * (kMW - 2);
        const double* b0 = w.pri_b + (size_t)b * d.NPRI;
        // g <- b0 + A dx at the prior's columns, 0 elsewhere; cost = 1/2 (c0 + 2 b0.dx + dx^T A dx)
        double pc = 0.0;
        for (int a = t6; a < n; a += NT6) {
            double v = 0;
            for (int c2 = 0; c2 < n; c2++) v += A[(size_t)a * n + c2] * s_dx[c2];
            pc += s_dx[a] * (b0[a] + 0.5 * v);
            s_pg[a] = b0[a] + v;
            if (s_pcol[a] >= 0 && !cost_only) g[s_pcol[a]] = b0[a] + v;
        }
        if (!cost_only) for (int c = t6; c < R; c += NT6) if (s_pidx[c] < 0) g[c] = 0.0;
        pc = wave_sum_f64(pc);
        if (lane == 0) s_wc[wave] = pc;
        GF_WSTAMP_T(128, 68);
        // H <- A gathered to this pass's columns: lower triangle of the first R rows, four rows per wavefront in flight.  Outside the band of phase 3 the
        // entries are the prior's alone and constant over the solve: each of the two buffer sets receives them once (marginalisation passes always: their
        // column map is another one).
        if (!h_have && !cost_only) for (int r0 = wave - 2; r0 < R; r0 += 4 * (kMW - 2)) {
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int r = r0 + (kMW - 2) * m;
                if (r >= R) continue;
                const int pr = s_pidx[r];
                double* dst = H + (size_t)r * RP;
                for (int c = lane; c <= r; c += 64) {
                    const int pc2 = s_pidx[c];
                    dst[c] = (pr >= 0 && pc2 >= 0) ? A[(size_t)pr * n + pc2] : 0.0;
                }
            }
        }
        GF_WSTAMP_T(128, 69);
    }
    __syncthreads();
    GF_WSTAMP(70);
    // ---- phase 2: one tile per factor on the matrix cores (the tile replaces the factor's block row)
    if (wave < nscr) {
        int slot = 0;
        for (int k = 0; k < nimu; k++) {
            if (!imu_on(k) || (slot++ % nscr) != wave) continue;
            const double c = misc_mfma_tile<15, 30>(w.imu_sqrt + ((size_t)b * d.W + k) * 225, sJi + kMJi * k, sW, lane);
            if (lane == 0) s_fcost[k] = c;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();   // scratch is reused by the next factor
        }
        for (int k = 0; k < nwh; k++) {
            if (!wh_on(k) || (slot++ % nscr) != wave) continue;
            const double c = misc_mfma_tile<6, 22>(w.wh_sqrt + ((size_t)b * d.W + k) * 36, sJw + kMJw * k, sW, lane);
            if (lane == 0) s_fcost[32 + k] = c;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    GF_WSTAMP(71);
    // ---- phase 3: the band.  Local tile indices: IMU factor at frame i: comp c (0-5 pose, 6-14 speed-bias) of frame i -> c, of frame i + 1 -> 15 + c,
    // residual 30; wheel factor at frame i: pose comp q of frame i -> q, of frame i + 1 -> 6 + q, shared column s (wheel extrinsic 0-5, sx, sy, sw,
    // td_wheel) -> 12 + s, residual 22.
    if (!cost_only) {
        auto TI = [&](int k, int la, int lb) -> double { const int hi = max(la, lb), lo = min(la, lb); return sJi[kMJi * k + hi * (hi + 1) / 2 + lo]; };
        auto TW = [&](int k, int la, int lb) -> double { const int hi = max(la, lb), lo = min(la, lb); return sJw[kMJw * k + hi * (hi + 1) / 2 + lo]; };
        auto fcol = [&](int f, int c) -> int { const int c0 = colf[c < 6 ? fb_pose(f) : fb_sb(f)]; return c0 >= 0 ? c0 + (c < 6 ? c : c - 6) : -1; };
        auto scol = [&](int s2) -> int { const int blk = s2 < 6 ? fb_exw(NP) : s2 < 9 ? fb_sx(NP) + (s2 - 6) : fb_tdw(NP); const int c0 = colf[blk]; return c0 >= 0 ? c0 + (s2 < 6 ? s2 : 0) : -1; };
        auto prior = [&](int r, int c) -> double { const int pr = s_pidx[r], pc2 = s_pidx[c]; return (pr >= 0 && pc2 >= 0) ? A[(size_t)pr * n + pc2] : 0.0; };
        auto put = [&](int r, int c, double v) { H[(size_t)max(r, c) * RP + min(r, c)] = v; };
        const int n1 = NP * 120, n2 = (NP - 1) * 225, n3 = NP * 60, n4 = 55, n5 = NP * 15 + 10;
        for (int t = tid; t < n1 + n2 + n3 + n4 + n5; t += 64 * kMW) {
            if (t < n1) {                                             // diagonal block of frame f, entry (a >= c2)
                const int f = t / 120, e = t - 120 * f, a = tri_row(e), c2 = e - a * (a + 1) / 2;
                const int r = fcol(f, a), c = fcol(f, c2);
                if (r < 0 || c < 0) continue;
                double v = prior(r, c);
                const int kp = f > 0 ? s_imu_at[f - 1] : -1, kc = s_imu_at[f];
                if (kp >= 0) v += TI(kp, 15 + a, 15 + c2);
                if (kc >= 0) v += TI(kc, a, c2);
                if (a < 6) {
                    const int wp = f > 0 ? s_wh_at[f - 1] : -1, wc = s_wh_at[f];
                    if (wp >= 0) v += TW(wp, 6 + a, 6 + c2);
                    if (wc >= 0) v += TW(wc, a, c2);
                }
                put(r, c, v);
            } else if (t < n1 + n2) {                                 // coupling of frame f + 1 (row comp a) with frame f (column comp c2)
                const int u = t - n1, f = u / 225, e = u - 225 * f, a = e / 15, c2 = e - 15 * a;
                const int r = fcol(f + 1, a), c = fcol(f, c2);
                if (r < 0 || c < 0) continue;
                double v = prior(r, c);
                const int kc = s_imu_at[f], wc = s_wh_at[f];
                if (kc >= 0) v += TI(kc, 15 + a, c2);
                if (a < 6 && c2 < 6 && wc >= 0) v += TW(wc, 6 + a, c2);
                put(r, c, v);
            } else if (t < n1 + n2 + n3) {                            // shared column s2 against pose comp q of frame f
                const int u = t - n1 - n2, f = u / 60, e = u - 60 * f, s2 = e / 6, q = e - 6 * s2;
                const int r = scol(s2), c = fcol(f, q);
                if (r < 0 || c < 0) continue;
                double v = prior(r, c);
                const int wp = f > 0 ? s_wh_at[f - 1] : -1, wc = s_wh_at[f];
                if (wp >= 0) v += TW(wp, 12 + s2, 6 + q);
                if (wc >= 0) v += TW(wc, 12 + s2, q);
                put(r, c, v);
            } else if (t < n1 + n2 + n3 + n4) {                       // shared x shared: all wheel factors, in factor order
                const int e = t - n1 - n2 - n3, a = tri_row(e), c2 = e - a * (a + 1) / 2;
                const int r = scol(a), c = scol(c2);
                if (r < 0 || c < 0) continue;
                double v = prior(r, c);
                for (int k = 0; k < nwh; k++) if (wh_on(k)) v += TW(k, 12 + a, 12 + c2);
                put(r, c, v);
            } else {                                                  // right-hand side
                const int e = t - n1 - n2 - n3 - n4;
                if (e < NP * 15) {
                    const int f = e / 15, a = e - 15 * f, c = fcol(f, a);
                    if (c < 0) continue;
                    double v = s_pidx[c] >= 0 ? s_pg[s_pidx[c]] : 0.0;
                    const int kp = f > 0 ? s_imu_at[f - 1] : -1, kc = s_imu_at[f];
                    if (kp >= 0) v += TI(kp, 30, 15 + a);
                    if (kc >= 0) v += TI(kc, 30, a);
                    if (a < 6) {
                        const int wp = f > 0 ? s_wh_at[f - 1] : -1, wc = s_wh_at[f];
                        if (wp >= 0) v += TW(wp, 22, 6 + a);
                        if (wc >= 0) v += TW(wc, 22, a);
                    }
                    g[c] = v;
                } else {
                    const int s2 = e - NP * 15, c = scol(s2);
                    if (c < 0) continue;
                    double v = s_pidx[c] >= 0 ? s_pg[s_pidx[c]] : 0.0;
                    for (int k = 0; k < nwh; k++) if (wh_on(k)) v += TW(k, 22, 12 + s2);
                    g[c] = v;
                }
            }
        }
    }
    if (tid == 0) {
        // A window without a prior has no r0^T r0: ba_setup writes pri_c only for n > 0, so the value must not be READ here (rounds 1-4 multiplied it by zero --
        // 0.5 * pri_c[b] * (n > 0 ? 1 : 0) -- which is NaN when the allocation still holds a NaN bit pattern, e.g. the -1 markers of a freed handle's int table:
        // the cost went NaN and the trust-region logic took another path.  Located in round 5: DESIGN.md section 2.)
#ifdef GF_TEST_REINTRODUCE_PRI_C_READ   // self-test of the stale-memory tooling only (scripts/stale_bisect.py must name pri_c on a library built with this)
        double c = 0.5 * w.pri_c[b] * (n > 0 ? 1.0 : 0.0);
#else
        double c = n > 0 ? 0.5 * w.pri_c[b] : 0.0;
#endif
        for (int q = 2; q < kMW; q++) c += s_wc[q];
        for (int k = 0; k < nimu; k++) c += s_fcost[k];
        for (int k = 0; k < nwh; k++) c += s_fcost[32 + k];
        *cost_part(w, 0, which, b) = c;
        *cost_part(w, 2, which, b) = 0.0;   // the GNSS kernel (same stream, later) adds its part
        if (!cost_only) w.st[b].h_prior[which] = h_keeps ? 1 : 0;   // (a cost-only pass leaves the buffer set's H as it found it)
    }
    GF_WSTAMP(75);
}
__global__ void __launch_bounds__(64 * kMW) ba_linearize_misc_win(Win w, int which, int which_state, int only_cand_valid, int frame_filter) {
    ba_linearize_misc_body(w, which, which_state, only_cand_valid, frame_filter);
}

// PoseLocalParameterization::Plus (pose_local_parameterization.cpp:12-28)
__device__ __forceinline__ void pose_plus(const double* x, const double* dl, double* out) {
    out[0] = x[0] + dl[0]; out[1] = x[1] + dl[1]; out[2] = x[2] + dl[2];
    const Q4 q = qnormalized(qmul(Q4{x[6], x[3], x[4], x[5]}, deltaQ(v3(dl[3], dl[4], dl[5]))));
    out[3] = q.x; out[4] = q.y; out[5] = q.z; out[6] = q.w;
}

// reduced column of compact column k (or -1); the right-hand-side slot maps to `rhs_col`
__device__ __forceinline__ int compact_to_col(int k, const int* colf, int NP, int rhs_col) {
    if (k < 6 * NP) { const int c0 = colf[fb_pose(k / 6)]; return c0 >= 0 ? c0 + k % 6 : -1; }
    if (k < 6 * NP + 6) { const int c0 = colf[fb_ex(NP)]; return c0 >= 0 ? c0 + k - 6 * NP : -1; }
    if (k == 6 * NP + 6) return colf[fb_td(NP)];
    if (k == 6 * NP + 7) return rhs_col;
    return -1;
}

// In-register Cholesky of a 16x16 SPD block by 16 lanes (lane r holds row r), followed by the explicit inverse of the factor.
// s_L / s_inv: 16 x 17 LDS tiles.  Returns false if a pivot is not positive.  All 64 lanes of the wavefront must call it.
__device__ __forceinline__ double bcast_lane(double v, int srclane) {  // srclane must be wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane), hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}
// value of lane J of the own 16-lane row, in every lane of the row: one v_mov_b32_dpp row_newbcast per half (no SGPR round trip)
template <int J> __device__ __forceinline__ double row_bcast_c(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x150 + J, 0xf, 0xf, true);   // every source lane is valid: no `old` operand to set up
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x150 + J, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_bcast(double v, int j) {   // j must fold to a constant (unrolled loops)
    switch (j) {
        case 0: return row_bcast_c<0>(v); case 1: return row_bcast_c<1>(v); case 2: return row_bcast_c<2>(v); case 3: return row_bcast_c<3>(v);
        case 4: return row_bcast_c<4>(v); case 5: return row_bcast_c<5>(v); case 6: return row_bcast_c<6>(v); case 7: return row_bcast_c<7>(v);
        case 8: return row_bcast_c<8>(v); case 9: return row_bcast_c<9>(v); case 10: return row_bcast_c<10>(v); case 11: return row_bcast_c<11>(v);
        case 12: return row_bcast_c<12>(v); case 13: return row_bcast_c<13>(v); case 14: return row_bcast_c<14>(v); default: return row_bcast_c<15>(v);
    }
}
// acc += x[lane J of the own 16-lane row] * own, in ONE instruction: v_fmac_f64 takes its first source through DPP (64-bit DPP on gfx90a+ knows exactly one
// control, row_newbcast, which is the one needed here).  The compiler does not combine a 64-bit DPP move into its user, so this is inline assembly; the
// wait states a DPP read needs after the VALU write of its source (2) are the caller's business: dpp_fence() on the source before the first use.
template <int J> __device__ __forceinline__ void fmac_bcast_c(double& acc, double x, double own) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x), "v"(own), "n"(J));
}
// two wait states between the instruction that produced x and whatever reads x next (the data dependence through the operand keeps the order)
__device__ __forceinline__ void dpp_fence(double& x) { asm("s_nop 1" : "+v"(x)); }
// 64-bit row broadcast as one v_mov_b64_dpp (the 32-bit pair of row_bcast_c costs two instructions)
template <int J> __device__ __forceinline__ double row_bcast64_c(double v) { return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + J, 0xf, 0xf, true); }
// Factor and explicit inverse of a 16x16 block in ONE sweep of 16 dependent steps (the serial part of ba_step's blocked Cholesky).
// Lane (g, r) = (lane >> 4, lane & 15) holds row r of the block -- the four 16-lane groups factor redundantly, which keeps every broadcast inside a
// DPP row -- and the entries W[r][4m + g], m = 0..3, of W = L^-1: the groups split the columns of the inverse between them.
//   W[r][c] = (delta_rc - sum_{k<r} L[r][k] W[k][c]) / L[r][r]
// Step j knows column j of L and (scaling by 1 / L_jj) row j of W; it removes their product from the accumulators t_r[c] of the rows below.
// Against factor + separate forward substitution (two chains of 16 steps, the second one fed by LDS broadcasts): 10.2 k -> ~5 k cycles per block.
// The sweep is bound by instruction issue (1 300 VALU instructions, 27 % of them the 32-bit halves of DPP broadcasts: disassembly of scripts/bench_chol16.hip),
// not by the latency of its chain; round 3 folds every broadcast into the multiply-add that consumes it (v_fmac_f64_dpp row_newbcast): 176 broadcasts of
// 2 + 1 (+ 1) instructions become 176 single instructions.
// The block is read straight from the packed lower triangle S (rows / columns j0 .. j0 + nb - 1, padded with the identity); the inverse goes to s_inv
// (operand of the panel product) and, packed, back into S in place of the block -- the factor itself is not needed again.
// the 16 steps as template recursion: every lane index of a DPP control and every index into a[] / t[] is a compile-time constant
template <int J, int C, int N = 16> struct Chol16A {   // a_r[c] -= L_cj * L_rj for the columns c > j (N: order of the block, columns from N on are identity padding)
    static __device__ __forceinline__ void run(double (&a)[16], double nl, double l) { fmac_bcast_c<C>(a[C], nl, l); Chol16A<J, C + 1, N>::run(a, nl, l); }
};
template <int J, int N> struct Chol16A<J, N, N> { static __device__ __forceinline__ void run(double (&)[16], double, double) {} };
template <int J, int M> struct Chol16W {   // t_r[4m + g] += t_j[4m + g] * nlw for the accumulators whose column can be <= j
    static __device__ __forceinline__ void fence(double (&t)[4]) { if (4 * M <= J) { dpp_fence(t[M]); Chol16W<J, M + 1>::fence(t); } }
    static __device__ __forceinline__ void run(double (&t)[4], double nlw) { if (4 * M <= J) { fmac_bcast_c<J>(t[M], t[M], nlw); Chol16W<J, M + 1>::run(t, nlw); } }
};
template <int J> struct Chol16W<J, 4> { static __device__ __forceinline__ void fence(double (&)[4]) {} static __device__ __forceinline__ void run(double (&)[4], double) {} };
template <int J, int N = 16> struct Chol16Step {
    static __device__ __forceinline__ void run(double (&a)[16], double (&t)[4], int& r, double& rd_own, bool& good) {
        dpp_fence(a[J]);                                    // a[J] was last written by the previous step's DPP multiply-add
        const double piv = row_bcast64_c<J>(a[J]);
        if (!(piv > 0.0)) good = false;
        // 1 / sqrt(piv): hardware seed (~2^-26) and two Newton steps y += y (1/2 - (piv/2) y^2), written with explicit fused multiply-adds
        // (this file is compiled without contraction; on the serial chain of the factorisation every dependent operation counts)
        const double hp = 0.5 * piv;
        double rs = __builtin_amdgcn_rsq(piv);
        rs = __builtin_fma(rs, __builtin_fma(-(hp * rs), rs, 0.5), rs);
        rs = __builtin_fma(rs, __builtin_fma(-(hp * rs), rs, 0.5), rs);
        const double l = a[J] * rs;                         // L_rj (rows r >= j; on row j itself a[J] IS the pivot, so this is sqrt(piv)); rs = 1 / L_jj
        a[J] = l;
        // the row masks are formed here, one v_cmp each: hoisted out of the sweep they are 32 SGPR pairs, which the kernel does not have (they were being
        // spilled into VGPR lanes and read back with v_readlane in every step)
        asm("" : "+v"(r));
        if (r == J) rd_own = rs;
        // W: t_r[c] -= L_rj * (t_j[c] / L_jj) for the rows below j, as t_r[c] += t_j[c] * (-(L_rj / L_jj)): one DPP multiply-add per accumulator
        const double nlw = (r > J) ? -(l * rs) : 0.0;
        if (J > 0) Chol16W<J, 0>::fence(t);
        Chol16W<J, 0>::run(t, nlw);
        if (J < N - 1) {
            double nl = -l;
            dpp_fence(nl);
            Chol16A<J, J + 1, N>::run(a, nl, l);
        }
        Chol16Step<J + 1, N>::run(a, t, r, rd_own, good);
    }
};
template <int N> struct Chol16Step<N, N> { static __device__ __forceinline__ void run(double (&)[16], double (&)[4], int&, double&, bool&) {} };
template <class SPtr>
__device__ __forceinline__ bool wave_chol16_fused(SPtr S, int j0, int nb, double* s_inv, int lane) {
    int r = lane & 15;
    const int g = lane >> 4;
    double a[16], t[4];
    {   // lane r reads 16 consecutive entries of row j0 + r of the packed triangle from column j0 on: one base address, 16 loads that issue back to back.
        // Entries right of the diagonal (c > r) belong to the rows behind and are never looked at: row r only ever uses a[c] for c <= r (its own
        // part of column c), and what it computes into a[c > r] feeds nothing but itself.
        const int base = pk(j0 + min(r, nb - 1), j0);
#pragma unroll
        for (int c = 0; c < 16; c++) a[c] = S[base + c];
        if (nb < 16) {   // last block of the system: pad with the identity (wave-uniform branch)
#pragma unroll
            for (int c = 0; c < 16; c++) a[c] = (r < nb && c < nb) ? a[c] : (r == c ? 1.0 : 0.0);
        }
    }
#pragma unroll
    for (int m = 0; m < 4; m++) t[m] = (4 * m + g == r) ? 1.0 : 0.0;
    double rd_own = 0.0;
    bool good = true;
    Chol16Step<0>::run(a, t, r, rd_own, good);
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const int c = 4 * m + g;
        const double w = t[m] * rd_own;
        s_inv[r * 17 + c] = w;
        if (c <= r && r < nb) S[pk(j0 + r, j0 + c)] = w;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    return good;
}

// ---- round 6: the reduced system by its structure ("chain" form of ba_step, template parameter CH).
// With every pose and speed-bias block free the reduced columns are laid out pose_0 (6) sb_0 (9) pose_1 sb_1 ... then the trailing blocks (extrinsics, sx sy sw, td, td_wheel).
// No factor couples speed-bias blocks of frames that are not neighbours (IMUFactor: pose_i sb_i pose_j sb_j, imu_factor.h:28; WheelFactor: poses and wheel blocks only; the prior
// holds sb_0 alone, estimator.cpp:3448-3520), so S restricted to the 9 NP speed-bias columns V is block tridiagonal, and eliminating V first, from sb_{NP-1} down to sb_0, needs
// per step a 9 x 9 factor, the 9 x 9 coupling to the next block and a (ND + 1) x 9 panel against the dense rest D (poses + trailing blocks, ND = R - 9 NP, + the right-hand-side row).
// Only D's packed triangle -- 21 KB at ND = 72 instead of 118 KB at R = 171 -- and two panels live in LDS; the finished panels go to global memory (L2) for the backward pass.
// What this buys is footprint, not a shorter chain (171 pivots stay 171 pivots): two windows fit one CU (ba_step_chain: 256 threads, <= 256 registers, < 80 KB LDS).
__device__ __forceinline__ int ch_dm(int c, int NP) { return c < 15 * NP ? 6 * (c / 15) + (c % 15) : c - 9 * NP; }   // reduced column of D (or the rhs slot R) -> its index in D; NOT for speed-bias columns
__device__ __forceinline__ int ch_d2c(int dj, int NP) { return dj < 6 * NP ? 15 * (dj / 6) + dj % 6 : dj + 9 * NP; }
__device__ __forceinline__ bool ch_isv(int c, int NP) { return c < 15 * NP && (c % 15) >= 6; }
constexpr int kChPS = 13;    // row stride of a panel in LDS: 9 columns + 3 of zero padding (K = 12 for three MFMAs) + 1 against bank conflicts
constexpr int kChYS = 9;     // row stride of a finished panel in global memory
__host__ __device__ inline int ch_panel_rows(int ND) { return (ND + 1 + 15) & ~15; }
__host__ __device__ inline size_t ch_lds_doubles(int ND) { return (((size_t)(ND + 1) * (ND + 2) / 2 + 1) & ~(size_t)1) + 2 * (size_t)ch_panel_rows(ND) * kChPS + 3 * 272; }
__host__ __device__ inline size_t ch_y_stride(int ND) { return (size_t)ch_panel_rows(ND) * kChYS + 2 * 81; }   // per (window, frame): panel, W = L_kk^-1, E = L[sb_{k-1}, sb_k]
// factor + explicit inverse of the nb x nb block held in a 17-stride LDS tile (lower part read), identity-padded to 16: the inverse goes to s_inv (17-stride)
template <int NB, int LD>
__device__ __forceinline__ bool wave_chol_tile(const double* sA, double* s_inv, int lane) {
    int r = lane & 15;
    const int g = lane >> 4;
    double a[16], t[4];
    {
        const double* src = sA + min(r, NB - 1) * LD;
#pragma unroll
        for (int c = 0; c < 16; c++) a[c] = c < NB ? src[c] : 0.0;
#pragma unroll
        for (int c = 0; c < 16; c++) a[c] = (r < NB && c < NB) ? a[c] : (r == c ? 1.0 : 0.0);
    }
#pragma unroll
    for (int m = 0; m < 4; m++) t[m] = (4 * m + g == r) ? 1.0 : 0.0;
    double rd_own = 1.0;   // rows from NB on are identity padding: the NB steps of the sweep never reach them
    bool good = true;
    Chol16Step<0, NB>::run(a, t, r, rd_own, good);
#pragma unroll
    for (int m = 0; m < 4; m++) s_inv[r * 17 + 4 * m + g] = t[m] * rd_own;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    return good;
}

// One 512-thread block per window: accept/reject of the previous candidate (trust_region_minimizer.cc), then the next
// dogleg step (dogleg_strategy.cc): Jacobi scaling, Cauchy point, Gauss-Newton step through the Schur complement
// (MFMA GEMM) and an LDS-resident blocked Cholesky, candidate point.  first: the call that follows the initial linearisation.
// GS: the packed reduced system lives in global memory (sb.Sg) instead of LDS -- windows whose (R+1)(R+2)/2 doubles exceed 160 KB
// (WINDOW_SIZE > 10); same code, the triangular solves then run out of L2.
// NW: wavefronts per block (8: one block owns a CU's registers; 4: half of them, so that other kernels' wavefronts -- the tracker's -- can sit next to it)
template <bool GS, int NW = 8, bool CH = false>
__device__ __forceinline__ void ba_step_body(Win w, StepBufs sb, int first, int max_iters, int finalize_only) {
    static_assert(!(CH && GS), "the chain form keeps its dense part in LDS");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NRC = CH ? 256 : 512;   // capacity of the per-reduced-column tables (chain form: R <= 15 NP + 17 <= 195 by the host's choice)
    constexpr int NCC = CH ? 96 : 256;    // capacity of the per-compact-column tables (chain form: ECW = 80)
    __shared__ double sred[512];
    __shared__ double s_inv[16 * 17];
    __shared__ double s_y[32];   // the solution of the current block of the backward substitution, double-buffered
    __shared__ int s_flag[4];
    __shared__ int s_cmap[NCC];      // compact column -> reduced column (or -1), right-hand-side slot -> R
    __shared__ double s_uc[NCC];     // a vector gathered to the compact layout (the Gauss-Newton solution during the back-substitution)
    __shared__ double s_ucc[NCC];    // the Cauchy direction in the compact layout (kept for the quadratic form u^T H u)
    __shared__ double s_rd[NRC];     // scratch: the Cauchy direction during the Schur pass, the solution during the backward substitution
    __shared__ int s_rc[NRC];        // reduced column -> compact column of the visual system Vc (or -1)
    __shared__ double s_hd[NRC];     // diagonal of H + Vc
    __shared__ double s_gt[NRC];     // g + Vc's right-hand-side row
    __shared__ double s_yd[CH ? 96 : 1];   // chain form: the solution of the dense part
    constexpr int NT = 64 * NW;
    const Dims d = w.d;
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    SolverState& st = w.st[b];
    if (uni(st.done)) return;
    if ((uni(st.chain) != 0) != CH) return;   // a batch may hold windows of both kinds: each form of the kernel takes its own (the host launches a form only when the batch has such windows)
    const int R = uni(st.R), NE = uni(st.NE), RP = d.RP, VS = sb.VS, ECW = d.ECW;
    double* S = GS ? sb.Sg + (size_t)blockIdx.x * sb.SgStride : smem;  // packed lower (R+1)(R+2)/2: row R carries the right-hand side
    // chain form: S is the packed triangle of the DENSE part only (poses + trailing blocks, ND columns, row ND = right-hand side); RC = dimension of what the blocked
    // Cholesky below factors; DMAP: reduced column (or the right-hand-side slot R) -> row / column of S
    const int CNP = d.NP, ND = CH ? R - 9 * CNP : R, RC = CH ? ND : R;
#define DMAP(c) (CH ? ch_dm((c), CNP) : (c))
    const int* colf = w.colf + (size_t)b * d.NFB;
    const int* cole = w.cole + (size_t)b * d.F;
    double* scale = sb.scale + (size_t)b * VS; double* diag = sb.diag + (size_t)b * VS; double* grad = sb.grad + (size_t)b * VS;
    double* gn = sb.gn + (size_t)b * VS; double* stepv = sb.step + (size_t)b * VS; double* u = sb.u + (size_t)b * VS; double* yv = sb.yv + (size_t)b * VS;
    double* Es = sb.Es + (size_t)b * d.FP * ECW;
    if (tid < ECW) s_cmap[tid] = compact_to_col(tid, colf, d.NP, R);
    for (int q = tid; q < NRC; q += NT) s_rc[q] = -1;
    __syncthreads();
    if (tid < 6 * d.NP + 7) { const int c = s_cmap[tid]; if (c >= 0 && c < R) s_rc[c] = tid; }
    GF_STAMP(0);
    // ---------------- accept / reject the candidate of the previous iteration
    if (tid == 0) {
        s_flag[0] = 0;
        if (first) {
            st.x_cost = cost_total(w, st.cur, b);
            st.initial_cost = st.x_cost;
            st.last_successful = 1;
        } else if (st.cand_valid) {
            const double cand_cost = cost_total(w, 1 - st.cur, b);
            st.cand_cost = cand_cost;
            if (st.step_norm <= 1e-8 * (st.x_norm + 1e-8)) { st.done = 1; st.termination = 2; }
            else if (fabs(st.x_cost - cand_cost) <= 1e-6 * st.x_cost) { st.done = 1; st.termination = 1; }
            else {
                const double rel = (st.x_cost - cand_cost) / st.model_cost_change;
                if (rel > 1e-3) {
                    st.cur = 1 - st.cur; st.x_cost = cand_cost; st.successful++; st.last_successful = 1; s_flag[0] = 1;
                    if (rel < 0.25) st.radius *= 0.5;
                    if (rel > 0.75) st.radius = fmax(st.radius, 3.0 * st.dogleg_step_norm);
                    st.mu = fmax(1e-8, 2.0 * st.mu / 10.0);
                    st.reuse = 0;
                } else { st.radius *= 0.5; st.reuse = 1; st.last_successful = 0; }
            }
        }
    }
    __syncthreads();
    if (uni(st.done)) return;
    const int cur = uni(st.cur);
    double* H = w.H + ((size_t)cur * d.B + b) * RP * RP;
    double* g = w.g + ((size_t)cur * d.B + b) * RP;
    const double* xs = w.xs + ((size_t)cur * d.B + b) * d.XS;
    const double* Et = sb.Et + ((size_t)cur * d.B + b) * d.FP * ECW;
    const double* ete = sb.ete + ((size_t)cur * d.B + b) * d.FP;
    const double* etb = sb.etb + ((size_t)cur * d.B + b) * d.FP;
    // the visual part of the normal equations lives in its own compact system Vc (ba_linearize_visual_win): diagonal and right-hand side of
    // the sum, once per call
    const double* Vc = w.Vc + ((size_t)cur * d.B + b) * d.NVC;
    const int RHSK = 6 * d.NP + 7;
    for (int c = tid; c < R; c += NT) {
        const int k = s_rc[c];
        s_hd[c] = H[(size_t)c * RP + c] + (k >= 0 ? Vc[pk(k, k)] : 0.0);
        s_gt[c] = g[c] + (k >= 0 ? Vc[pk(RHSK, k)] : 0.0);
    }
    __syncthreads();
    GF_STAMP(1);
    if (!uni(st.reuse)) {
        // ---------------- Jacobi scaling from the initial Jacobian (trust_region_minimizer.cc: jacobian_scaling_)
        if (!st.have_scale) {
            for (int c = tid; c < R; c += NT) scale[c] = 1.0 / (1.0 + sqrt(s_hd[c]));
            for (int e = tid; e < NE; e += NT) scale[RP + e] = 1.0 / (1.0 + sqrt(ete[e]));
            __syncthreads();
            if (tid == 0) st.have_scale = 1;
        }
        // unscaled gradient max norm (gradient tolerance)
        double gm = 0;
        for (int c = tid; c < R; c += NT) gm = fmax(gm, fabs(s_gt[c]));
        for (int e = tid; e < NE; e += NT) gm = fmax(gm, fabs(etb[e]));
        gm = block_max<NW>(gm, sred, tid, NT);
        if (tid == 0) st.gmax = gm;
    }
    GF_STAMP(2);
    // ---------------- FinalizeIterationAndCheckIfMinimizerCanContinue
    if (tid == 0) {
        if (st.iterations >= max_iters) { st.done = 1; st.termination = 0; }
        else if (st.last_successful && st.gmax <= 1e-10) { st.done = 1; st.termination = 3; }
        else if (st.radius <= 1e-32) { st.done = 1; st.termination = 4; }
        else st.iterations++;
    }
    __syncthreads();
    GF_STAMP(3);
    if (uni(st.done) || finalize_only) return;

    if (!uni(st.reuse)) {
        GF_STAMP(4);
        // ---------------- dogleg diagonal, scaled gradient, Cauchy point
        for (int c = tid; c < R; c += NT) {
            const double dd = sqrt(fmin(fmax(scale[c] * scale[c] * s_hd[c], 1e-6), 1e32));
            diag[c] = dd; grad[c] = scale[c] * s_gt[c] / dd; u[c] = scale[c] * (grad[c] / dd);
        }
        for (int e = tid; e < NE; e += NT) {
            const double sc = scale[RP + e];
            const double dd = sqrt(fmin(fmax(sc * sc * ete[e], 1e-6), 1e32));
            diag[RP + e] = dd; grad[RP + e] = sc * etb[e] / dd; u[RP + e] = sc * (grad[RP + e] / dd);
            gn[RP + e] = u[RP + e];   // the Cauchy direction's eliminated part survives in gn's tail until the back-substitution rewrites it
        }
        __syncthreads();
        if (tid < ECW) { const int c = s_cmap[tid]; s_ucc[tid] = (c >= 0 && c < R) ? u[c] : 0.0; }
        // the passes below read the column scaling and the Cauchy direction once per matrix entry: LDS copies (s_hd and s_rd are free until the
        // next call / the Cholesky)
        for (int c = tid; c < R; c += NT) { s_hd[c] = scale[c]; s_rd[c] = u[c]; }
        double gsq = 0;
        for (int c = tid; c < R; c += NT) gsq += grad[c] * grad[c];
        for (int e = tid; e < NE; e += NT) gsq += grad[RP + e] * grad[RP + e];
        gsq = block_sum<NW>(gsq, sred, tid, NT);
        // alpha = |g~|^2 / (u^T H u) of the Cauchy point: the quadratic form is accumulated below, inside the passes that stream Et (Es build) and
        // H (load of the reduced system) anyway, instead of a separate sweep over both
        double uHu_acc = 0.0;
        bool need_alpha = true, asm_alpha_done = false;
        int ch_alpha_k = 1 << 20;   // chain form: the fronts k >= ch_alpha_k have added their entries of H to u^T H u already (a retry with a larger mu must not add them again)
        constexpr int QN = GS ? 8 : 3;
        double uk[QN];
#pragma unroll
        for (int q = 0; q < QN; q++) uk[q] = (lane + 64 * q) < R ? u[lane + 64 * q] : 0.0;
        // ---------------- Gauss-Newton step: (J^T J + mu D^2) y = J^T r through the Schur complement, retry with larger mu on failure
        bool ok = false;
        while (!ok) {
            const double mu = uni(st.mu);
            if (!(mu < 1.0)) break;
            GF_STAMP(5);
            // eliminated columns: ete~ = s_e^2 ete + mu D_e^2 (kept in yv's tail), row factor f_e = s_e / sqrt(ete~) (in u's tail)
            for (int e = tid; e < NE; e += NT) {
                const double sc = scale[RP + e], lm = diag[RP + e] * sqrt(mu);
                const double et = sc * sc * ete[e] + lm * lm;
                yv[RP + e] = et; u[RP + e] = sc / sqrt(et);
            }
            __syncthreads();
            // The scaled compact rows Es[e][k] = f_e s_c Et[e][k] (c = reduced column of k; the right-hand-side slot carries f_e etb_e so that the GEMM also
            // reduces the right-hand side) are not materialised any more: the Schur GEMM and the back-substitution of the eliminated columns scale the
            // rows of Et as they load them (same products in the same order, so the same bits as the stored copy; 78 KB written and read back per call saved).
            const int NE4 = (NE + 3) & ~3;
            GF_STAMP(6);
            // reduced system in LDS (packed lower): S = s (H + Vc) s + mu D^2, row R = s g.  First the H part, row by row ...
            // (chain form: only the rows and columns of the dense part; what a speed-bias column holds reaches its front in the elimination below)
            for (int r0 = wave; r0 <= RC; r0 += 4 * NW) {   // four rows per wavefront in flight: all loads first, then the arithmetic
                double hv[4][QN];
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const int dr = r0 + NW * m;
                    const int r = CH ? (dr < RC ? ch_d2c(dr, CNP) : R) : dr;
                    const double* hr = H + (size_t)min(r, R - 1) * RP;
#pragma unroll
                    for (int q = 0; q < QN; q++) { const int c = lane + 64 * q; hv[m][q] = (r < R && c <= r) ? hr[c] : 0.0; }
                }
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const int dr = r0 + NW * m;
                    if (dr > RC) continue;
                    const int r = CH ? (dr < RC ? ch_d2c(dr, CNP) : R) : dr;
                    const int base = pk(dr, 0);
                    const double sr = r < R ? s_hd[r] : 1.0, ur = r < R ? s_rd[r] : 0.0;
#pragma unroll
                    for (int q = 0; q < QN; q++) {
                        const int c = lane + 64 * q;
                        if (CH && ch_isv(c, CNP)) continue;
                        if (r < R) {
                            if (c <= r) {
                                double v = sr * s_hd[c] * hv[m][q];
                                if (c == r) { const double lm = diag[r] * sqrt(mu); v += lm * lm; }
                                S[base + DMAP(c)] = v;
                                if (!asm_alpha_done) uHu_acc += (c == r ? 1.0 : 2.0) * hv[m][q] * uk[q] * ur;   // H holds its lower triangle
                            }
                        } else if (c < R) S[base + DMAP(c)] = s_hd[c] * s_gt[c];   // row R: the right-hand side s g (visual part included in s_gt)
                    }
                }
            }
            __syncthreads();
            // ... then the visual part, compact row by compact row: entry (ka >= kb) of Vc lands on its own entry (r >= c) of S
            //     (four compact rows per wavefront in flight: every load is issued before the first use -- one row at a time this pass was a chain of
            //     8-9 dependent global loads per wavefront)
            {
                const int nka = 6 * d.NP + 7;
                constexpr int VQ = GS ? 3 : 2;   // 64-column pieces of a compact row (<= 6 NP + 8 columns)
                for (int ka0 = wave; ka0 < nka; ka0 += 4 * NW) {
                    double vv[4][VQ];
#pragma unroll
                    for (int m = 0; m < 4; m++) {
                        const int ka = ka0 + NW * m;
                        const double* vrow = Vc + pk(min(ka, nka - 1), 0);
#pragma unroll
                        for (int q = 0; q < VQ; q++) { const int kb = lane + 64 * q; vv[m][q] = (ka < nka && kb <= ka) ? vrow[kb] : 0.0; }
                    }
#pragma unroll
                    for (int m = 0; m < 4; m++) {
                        const int ka = ka0 + NW * m;
                        if (ka >= nka) continue;
                        const int r = s_cmap[ka];
                        if (r < 0 || r >= R) continue;
                        const double sr = s_hd[r], ur = s_rd[r];
#pragma unroll
                        for (int q = 0; q < VQ; q++) {
                            const int kb = lane + 64 * q;
                            if (kb > ka) continue;
                            const int c = s_cmap[kb];
                            if (c < 0) continue;
                            const double v = vv[m][q];
                            S[pk(DMAP(r), DMAP(c))] += sr * s_hd[c] * v;
                            if (!asm_alpha_done) uHu_acc += (c == r ? 1.0 : 2.0) * v * ur * s_rd[c];
                        }
                    }
                }
            }
            if (need_alpha && !asm_alpha_done) {   // the eliminated columns' share of u^T H u (E^T F u and E^T E) follows in the back-substitution pass below, which streams Et anyway
                for (int e = tid; e < NE; e += NT) { const double ue = gn[RP + e]; uHu_acc += ete[e] * ue * ue; }
                asm_alpha_done = true;
            }
            __syncthreads();
            GF_STAMP(7);
            // S -= Es^T Es on the matrix cores over the COMPACT columns (poses, ex, td, rhs slot; speed-bias / wheel columns are structurally
            // zero): 16x16 tiles of the compact lower triangle, K = eliminated columns (4 per MFMA), scattered into the packed S.
            {
                constexpr int GB = 4;   // k-steps per batch: two batches of GB loads x 2 operands live at a time (the kernel sits at the 256-VGPR limit: eight spilled)
                const int nt = ECW / 16, ntiles = nt * (nt + 1) / 2, nk = NE4 / 4, nbat = (nk + GB - 1) / GB;
                // f_e = s_e / sqrt(ete~) and f_e etb_e per eliminated column: staged in LDS (sred is free between the reductions) when they fit, else read from global
                const bool fe_lds = NE4 <= 256;
                if (fe_lds) {
                    for (int e = tid; e < NE4; e += NT) { const double f = e < NE ? u[RP + e] : 0.0; sred[e] = f; sred[256 + e] = e < NE ? f * etb[e] : 0.0; }
                    __syncthreads();
                }
                for (int t = wave; t < ntiles; t += NW) {
                    const int ti = tri_row(t), tk = t - ti * (ti + 1) / 2;
                    const int ka = 16 * ti + (lane & 15), kb = 16 * tk + (lane & 15), eg = lane >> 4;
                    const int ca = s_cmap[ka], cb = s_cmap[kb];
                    // per-lane column factors: s_c for a reduced column, the right-hand-side slot takes f_e etb_e instead of f_e s_c Et
                    const double sca = (ca >= 0 && ca < R) ? s_hd[ca] : 0.0, scb = (cb >= 0 && cb < R) ? s_hd[cb] : 0.0;
                    const bool rha = ca == R, rhb = cb == R;
                    const double* pa = Et + (size_t)eg * ECW + ka;
                    const double* pb = Et + (size_t)eg * ECW + kb;
                    d4 acc = {0, 0, 0, 0};
                    // batches of GB k-steps (4 GB eliminated columns), double-buffered: the loads of batch n + 1 are issued before the matrix products of batch n
                    auto load = [&](int bat, double (&av)[GB], double (&bv2)[GB]) {
#pragma unroll
                        for (int q = 0; q < GB; q++) {
                            const int kk = GB * bat + q, e = 4 * kk + eg;
                            const bool live = e < NE;
                            av[q] = live ? pa[(size_t)4 * kk * ECW] : 0.0; bv2[q] = live ? pb[(size_t)4 * kk * ECW] : 0.0;
                        }
                    };
                    auto mma = [&](int bat, const double (&av)[GB], const double (&bv2)[GB]) {
#pragma unroll
                        for (int q = 0; q < GB; q++) {
                            const int e = 4 * (GB * bat + q) + eg;
                            double fe, feb;
                            if (fe_lds) { fe = e < NE4 ? sred[e] : 0.0; feb = e < NE4 ? sred[256 + e] : 0.0; }
                            else { fe = e < NE ? u[RP + e] : 0.0; feb = (e < NE && (rha || rhb)) ? fe * etb[e] : 0.0; }
                            const double a0 = rha ? feb : fe * sca * av[q], b0 = rhb ? feb : fe * scb * bv2[q];   // = the former Es entries
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
                        }
                    };
                    double avA[GB], bvA[GB], avB[GB], bvB[GB];
                    if (nbat > 0) load(0, avA, bvA);
                    for (int bat = 0; bat < nbat; bat += 2) {
                        if (bat + 1 < nbat) load(bat + 1, avB, bvB);
                        mma(bat, avA, bvA);
                        if (bat + 1 < nbat) {
                            if (bat + 2 < nbat) load(bat + 2, avA, bvA);
                            mma(bat + 1, avB, bvB);
                        }
                    }
                    const int col = s_cmap[16 * tk + (lane & 15)];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int row = s_cmap[16 * ti + (lane >> 4) + 4 * r];   // compact -> reduced is monotone: lower stays lower
                        if (row >= 0 && col >= 0 && col <= row && col < R) S[pk(DMAP(row), DMAP(col))] -= acc[r];
                    }
                }
            }
            __syncthreads();
            GF_STAMP(8);
            // ---- blocked Cholesky (block 16).  The right-hand side rides along as row R, so the panel multiply performs the forward
            //      substitution.  Diagonal block: in-register factor + explicit inverse (wavefront 0); panel and trailing update: MFMA.
            if (tid == 0) s_flag[1] = 1;
            __syncthreads();
            if (CH) {
                // ---------------- chain form: eliminate the speed-bias blocks sb_{NP-1} ... sb_0 (see the note in front of this function).  Step k:
                //  (1) all wavefronts: the front of sb_k -- its entries of H are in place already -- receives what column block k + 1 left,
                //      P_k -= Y_{k+1} E_{k+1}^T,  A_k -= E_{k+1} E_{k+1}^T  (MFMA), and the dense part takes Y_{k+1} Y_{k+1}^T;
                //  (2) wavefront 0 factors the 9 x 9 block and inverts the factor (W_k).  Meanwhile the other wavefronts set up the NEXT front in the panel that has just
                //      become free: A_{k-1} + mu D^2, C_{k-1} = S(sb_{k-1}, sb_{k-2}), the panel rows of the poses k - 2 .. k (k - 1 = 0: every row, the prior is dense in sb_0)
                //      and the right-hand side, from values of H loaded one step earlier (and their share of the Cauchy point's u^T H u), and issue the loads of the front behind it;
                //  (3) all wavefronts: Y_k = P_k W_k^T in place, E_k = C_k^T W_k^T; Y_k, W_k, E_k go to global memory for the backward pass.
                // Three block barriers per step; the serial part of a step is the 9 x 9 factor.
                const int PR = ch_panel_rows(ND), NTD = PR >> 4, NTT = NTD * (NTD + 1) / 2;
                double* Pb0 = smem + ((((size_t)(ND + 1) * (ND + 2) / 2) + 1) & ~(size_t)1);
                double* sA0 = Pb0 + 2 * (size_t)PR * kChPS; double* sC0 = sA0 + 272; double* sE = sC0 + 272;   // sA, sC: two 9 x 9 tiles each (row stride 15), one per parity of the step; sE: 16 x 17
                double* Yg = sb.Yg + (size_t)b * CNP * sb.YgStride;
                for (int q = tid; q < 2 * PR * kChPS + 3 * 272; q += NT) Pb0[q] = 0.0;   // padding columns 9 .. 12 and the rows behind ND stay zero for good
                for (int c = tid; c < R; c += NT) sred[c] = diag[c];                    // the dogleg diagonal next to the fronts (sred is free between the reductions)
                __syncthreads();
                const int uw = uni(wave);
                const double smu = sqrt(mu);
                const int ti_ = lane & 15, kq_ = lane >> 4;
                // A finished panel Y_k is zero above row 6 (k - 1) (the fill of the elimination order: sb_k meets the poses k - 1 .. NP - 1 and, k = 0, everything): row tiles
                // in front of ftile(k) are skipped by every product that involves Y_k -- a third of the tile work of the loop.
                auto ftile = [&](int k) -> int { return k == 0 ? 0 : (6 * (k - 1)) >> 4; };
                // the dense part takes Y Y^T of a finished panel straight away: tile by tile, three MFMAs and one read-modify-write of the tile's own entries of S (an update
                // held in registers over the whole loop -- 48 VGPRs -- pushed the kernel into scratch memory)
                auto dense_update = [&](const double* Y, int ft) {
                    int cnt = 0;
                    for (int ti = ft; ti < NTD; ti++)
                        for (int tj = ft; tj <= ti; tj++) {
                            if ((cnt++ & (NW - 1)) != uw) continue;
                            const double* pa = Y + (size_t)(16 * ti + ti_) * kChPS + kq_;
                            const double* pb = Y + (size_t)(16 * tj + ti_) * kChPS + kq_;
                            const double a0 = pa[0], a1 = pa[4], a2 = pa[8], b0 = pb[0], b1 = pb[4], b2 = pb[8];
                            d4 acc = {0, 0, 0, 0};
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc, 0, 0, 0);
                            const int col = 16 * tj + ti_, r0 = 16 * ti + kq_;
                            double* sp = S + pk(r0, col);
                            double cur[4];   // the four entries first, then the four stores: written as read-modify-write per entry the compiler kept them in order, one LDS round trip each
#pragma unroll
                            for (int r = 0; r < 4; r++) { const int row = r0 + 4 * r; cur[r] = (row <= ND && col <= row && col < ND) ? sp[4 * r * r0 + 2 * r * (4 * r + 1)] : 0.0; }   // pk(r0 + 4 r, col) - pk(r0, col) = 4 r r0 + 2 r (4 r + 1)
#pragma unroll
                            for (int r = 0; r < 4; r++) { const int row = r0 + 4 * r; if (row <= ND && col <= row && col < ND) sp[4 * r * r0 + 2 * r * (4 * r + 1)] = cur[r] - acc[r]; }
                        }
                };
                // the entries of H a front takes, enumerated per thread of the wavefronts 1 .. NW - 1.  Fronts k >= 1 have one shape (A lower 45, C 81, three pose blocks
                // 162): entry e of thread t3 is the same kind of entry in every front, its address moves by 15 rows and 15 columns per frame, and its value is loaded one step
                // before it is stored.  Front 0 (no C, every dense row: 45 + 9 ND entries) is gathered by the whole block when its turn comes.
                constexpr int NTF = 64 * (NW - 1);
                const int t3 = tid - 64;
                int f_hb[2], f_r0[2], f_c0[2], f_tg[2], f_kind[2];   // k >= 1: H offset at k = 0, reduced row / column at k = 0, LDS target, kind (0 A, 1 C, 2 P of pose k-1 / k, 3 P of pose k+1, -1 none)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int e = t3 + NTF * j;
                    f_kind[j] = -1; f_hb[j] = 0; f_r0[j] = 0; f_c0[j] = 0; f_tg[j] = 0;
                    if (t3 < 0) continue;
                    if (e < 45) { const int a = tri_row(e), bb = e - a * (a + 1) / 2; f_kind[j] = 0; f_r0[j] = 6 + a; f_c0[j] = 6 + bb; f_hb[j] = (6 + a) * RP + 6 + bb; f_tg[j] = a * 15 + bb; }
                    else if (e < 126) { const int e2 = e - 45, a = e2 / 9, bb = e2 - 9 * a; f_kind[j] = 1; f_r0[j] = 6 + a; f_c0[j] = bb - 9; f_hb[j] = (6 + a) * RP + bb - 9; f_tg[j] = a * 15 + bb; }
                    else if (e < 288) {
                        const int e2 = e - 126, dq = e2 / 9, bb = e2 - 9 * dq, dc = 15 * (dq / 6) - 15 + dq % 6, cc = 6 + bb;   // dense column dc + 15 k, speed-bias column cc + 15 k
                        f_kind[j] = dq < 12 ? 2 : 3; f_r0[j] = dc; f_c0[j] = cc;
                        f_hb[j] = dc > cc ? dc * RP + cc : cc * RP + dc; f_tg[j] = (dq - 6) * kChPS + bb;   // panel row 6 (k - 1) + dq = 6 k + dq - 6
                    }
                }
                double pf[2];
                auto front_load = [&](int k) {   // k >= 1: issue the loads of front k's entries of H (stored one step later)
                    const size_t sh = (size_t)k * (15 * RP + 15);
#pragma unroll
                    for (int j = 0; j < 2; j++) pf[j] = (f_kind[j] >= 0 && !(f_kind[j] == 3 && k == CNP - 1)) ? H[sh + f_hb[j]] : 0.0;
                };
                auto front_store = [&](int k, double* Pn, double* sAn, double* sCn) {   // k >= 1: front k into the (zeroed) panel Pn and the tiles sAn / sCn
                    const bool want_alpha = need_alpha && k < ch_alpha_k;
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        if (f_kind[j] < 0 || (f_kind[j] == 3 && k == CNP - 1)) continue;
                        const int r = f_r0[j] + 15 * k, c = f_c0[j] + 15 * k;
                        const double hh = pf[j];
                        double v = s_hd[r] * s_hd[c] * hh;
                        if (f_kind[j] == 0) { if (r == c) { const double lm = sred[r] * smu; v += lm * lm; } sAn[f_tg[j]] = v; }
                        else if (f_kind[j] == 1) sCn[f_tg[j]] = v;
                        else Pn[(size_t)6 * k * kChPS + f_tg[j]] = v;
                        if (want_alpha) uHu_acc += (r == c ? 1.0 : 2.0) * hh * s_rd[r] * s_rd[c];
                    }
                    if (t3 >= 0 && t3 < 9) { const int col = 15 * k + 6 + t3; Pn[(size_t)ND * kChPS + t3] = s_hd[col] * s_gt[col]; }   // right-hand side s g at the speed-bias columns
                    if (want_alpha) ch_alpha_k = k;
                };
                auto front0_gather = [&](double* Pn, double* sAn) {   // every thread of the block; the panel is zero
                    const bool want_alpha = need_alpha && 0 < ch_alpha_k;
                    const int nE = 45 + 9 * ND;
#pragma unroll 4
                    for (int e = tid; e < nE; e += NT) {
                        if (e < 45) {
                            const int a = tri_row(e), bb = e - a * (a + 1) / 2, r = 6 + a, c = 6 + bb;
                            const double hh = H[(size_t)r * RP + c];
                            double v = s_hd[r] * s_hd[c] * hh;
                            if (a == bb) { const double lm = sred[r] * smu; v += lm * lm; }
                            sAn[a * 15 + bb] = v;
                            if (want_alpha) uHu_acc += (a == bb ? 1.0 : 2.0) * hh * s_rd[r] * s_rd[c];
                        } else {
                            const int e2 = e - 45, dj = e2 / 9, bb = e2 - 9 * dj, cD = ch_d2c(dj, CNP), col = 6 + bb;
                            const double hh = cD > col ? H[(size_t)cD * RP + col] : H[(size_t)col * RP + cD];
                            Pn[(size_t)dj * kChPS + bb] = s_hd[cD] * s_hd[col] * hh;
                            if (want_alpha) uHu_acc += 2.0 * hh * s_rd[cD] * s_rd[col];
                        }
                    }
                    if (tid < 9) { const int col = 6 + tid; Pn[(size_t)ND * kChPS + tid] = s_hd[col] * s_gt[col]; }
                    if (want_alpha) ch_alpha_k = 0;
                };
                // the first front; the loads of the second
                if (CNP == 1) front0_gather(Pb0, sA0);
                else {
                    if (t3 >= 0) { front_load(CNP - 1); front_store(CNP - 1, Pb0, sA0, sC0); if (CNP >= 3) front_load(CNP - 2); }
                }
                __syncthreads();
#ifdef GF_PROFILE_STEP
                long long cq[5] = {0, 0, 0, 0, 0}, cq0 = clock64();
#define GF_CQ(i) do { const long long n_ = clock64(); cq[i] += n_ - cq0; cq0 = n_; } while (0)
#else
#define GF_CQ(i) do { } while (0)
#endif
                for (int k = CNP - 1; k >= 0; k--) {
                    const int par = (CNP - 1 - k) & 1;
                    double* Pc = Pb0 + (size_t)par * PR * kChPS;          // this step's panel (front k in place)
                    double* Pp = Pb0 + (size_t)(par ^ 1) * PR * kChPS;    // the finished panel of step k + 1; then the next front
                    double* sA = sA0 + 136 * par; double* sC = sC0 + 136 * par;
                    if (k < CNP - 1) {   // (1)
                        const int ft = ftile(k + 1);
                        for (int t = ft + uw; t <= NTD; t += NW) {   // tile NTD is the 9 x 9 block
                            d4 acc = {0, 0, 0, 0};
                            const double* pa = t < NTD ? Pp + (size_t)(16 * t + ti_) * kChPS + kq_ : sE + ti_ * 17 + kq_;
                            const double* pb = sE + ti_ * 17 + kq_;
                            const double a0 = pa[0], a1 = pa[4], a2 = pa[8], b0 = pb[0], b1 = pb[4], b2 = pb[8];
                            double c0[4];
                            if (ti_ < 9) {
#pragma unroll
                                for (int r = 0; r < 4; r++) { const int row = kq_ + 4 * r; c0[r] = t < NTD ? Pc[(size_t)(16 * t + row) * kChPS + ti_] : (row < 9 ? sA[row * 15 + ti_] : 0.0); }
                            }
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc, 0, 0, 0);
                            if (ti_ < 9) {
#pragma unroll
                                for (int r = 0; r < 4; r++) {
                                    const int row = kq_ + 4 * r;
                                    if (t < NTD) Pc[(size_t)(16 * t + row) * kChPS + ti_] = c0[r] - acc[r];
                                    else if (row < 9) sA[row * 15 + ti_] = c0[r] - acc[r];
                                }
                            }
                        }
                        dense_update(Pp, ft);
                    }
                    __syncthreads();
                    GF_CQ(0);
                    if (uw == 0) {   // (2)
                        __builtin_amdgcn_s_setprio(3);
                        const bool good = wave_chol_tile<9, 15>(sA, s_inv, lane);
                        if (!good && lane == 0) s_flag[1] = 0;
                        __builtin_amdgcn_s_setprio(0);
                    } else if (k >= 1) {   // the panel of step k + 1 is free now: cleared here, filled with the next front behind the next barrier (another thread's entry must not meet this zero late)
                        for (int q = t3; q < PR * 9; q += NTF) { const int rr = q / 9; Pp[(size_t)rr * kChPS + q - 9 * rr] = 0.0; }
                    }
                    __syncthreads();
                    GF_CQ(1);
                    if (!uni(s_flag[1])) break;
                    double* Yk = Yg + (size_t)k * sb.YgStride;
                    for (int t = ftile(k) + uw; t <= NTD; t += NW) {   // (3)
                        d4 acc = {0, 0, 0, 0};
                        double a0, a1, a2;
                        if (t < NTD) { const double* pa = Pc + (size_t)(16 * t + ti_) * kChPS + kq_; a0 = pa[0]; a1 = pa[4]; a2 = pa[8]; }
                        else { const int i9 = min(ti_, 8); a0 = sC[kq_ * 15 + i9]; a1 = sC[(kq_ + 4) * 15 + i9]; a2 = kq_ == 0 ? sC[8 * 15 + i9] : 0.0; if (ti_ >= 9) { a0 = 0.0; a1 = 0.0; a2 = 0.0; } }   // C_k^T: A[i][a] = C[a][i]
                        const double* pb = s_inv + ti_ * 17 + kq_;
                        const double b0 = pb[0], b1 = pb[4], b2 = pb[8];
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc, 0, 0, 0);
                        if (ti_ < 9) {
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const int row = kq_ + 4 * r;
                                if (t < NTD) { Pc[(size_t)(16 * t + row) * kChPS + ti_] = acc[r]; Yk[(size_t)(16 * t + row) * kChYS + ti_] = acc[r]; }
                                else if (row < 9) { sE[row * 17 + ti_] = acc[r]; Yk[(size_t)PR * kChYS + 81 + row * 9 + ti_] = acc[r]; }
                            }
                        }
                    }
                    if (tid < 81) Yk[(size_t)PR * kChYS + tid] = s_inv[(tid / 9) * 17 + tid % 9];
                    // the next front into the cleared panel (values of H loaded one step ago), then the loads of the one behind it
                    if (k >= 2) { if (t3 >= 0) { front_store(k - 1, Pp, sA0 + 136 * (par ^ 1), sC0 + 136 * (par ^ 1)); if (k >= 3) front_load(k - 2); } }
                    else if (k == 1) front0_gather(Pp, sA0 + 136 * (par ^ 1));
                    __syncthreads();
                    GF_CQ(2);
                }
#ifdef GF_PROFILE_STEP
                if (blockIdx.x == 0 && tid == 0 && sb.stamps) { sb.stamps[87] = cq[0]; sb.stamps[88] = cq[1]; sb.stamps[89] = cq[2]; sb.stamps[90] = clock64(); }
#endif
                if (uni(s_flag[1])) dense_update(Pb0 + (size_t)((CNP - 1) & 1) * PR * kChPS, 0);   // the last panel
                __syncthreads();
            }
#ifdef GF_PROFILE_STEP
            long long tA = 0, tB = 0, tC = 0, t0c = clock64();
            if (blockIdx.x == 0 && tid == 0 && sb.stamps) { sb.stamps[24] = sb.stamps[25] = sb.stamps[26] = sb.stamps[27] = sb.stamps[28] = sb.stamps[29] = sb.stamps[30] = sb.stamps[31] = 0; }
#define GF_SUB(acc) do { const long long n_ = clock64(); acc += n_ - t0c; t0c = n_; } while (0)
#else
#define GF_SUB(acc) do { } while (0)
#endif
            // diagonal block j0: in-register factor + explicit inverse (wavefront 0 only)
            auto diag_block = [&](int j0) {
                const int nb = min(16, RC - j0);
#ifdef GF_PROFILE_STEP
                long long d0 = clock64();
#define GF_DSUB(i) do { const long long n_ = clock64(); if (blockIdx.x == 0 && tid == 0 && sb.stamps) sb.stamps[i] += n_ - d0; d0 = n_; } while (0)
#else
#define GF_DSUB(i) do { } while (0)
#endif
                GF_DSUB(24);
                // the diagonal block of S receives the INVERSE of its factor: the panel below is multiplied with it (MFMA), and the backward
                // substitution becomes a 16x16 product per block instead of a chain of 16 dependent steps
                const bool good = wave_chol16_fused(S, j0, nb, s_inv, lane);
                GF_DSUB(25);
                GF_DSUB(26);
                if (!good && lane == 0) s_flag[1] = 0;
                GF_DSUB(27);
            };
            // ---- left-looking blocked Cholesky.  Block column j first receives ALL its updates -- tile (i, j) -= sum_{k < j} L_ik L_jk^T, accumulated on the
            // matrix cores over the 16 k columns to its left and subtracted ONCE --, then its diagonal block is factored (+ inverted) and the tiles below it
            // are multiplied with the inverse.  Against the right-looking form used before (every block column updated every tile of the trailing matrix:
            // one LDS read-modify-write pass per tile and block column, 15 k cycles for the first columns on seven wavefronts next to a serial wavefront that
            // needed 8 k) every tile is written twice in total, the operands of an update are plain loads, and the serial wavefront only waits for its own
            // diagonal tile: it updates that one itself and goes straight on to factor it while the others update the rest of the column.
            const int NTR = (RC + 16) / 16;   // tile rows: rows 0 .. R (row R carries the right-hand side)
            auto upd_tile = [&](int ti, int tj, int k0, int k1) {   // tile (ti, tj), ti >= tj, minus the products of the block rows ti and tj over the block columns k0 .. k1 - 1 (< tj)
                const int ra = 16 * ti + (lane & 15), rb = 16 * tj + (lane & 15), kq = lane >> 4;
                const bool va = ra <= RC, vb = rb < RC;   // row R (rhs) never acts as a column
                const int base_a = pk(min(ra, RC), 0) + kq, base_b = pk(min(rb, RC - 1), 0) + kq;
                d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
                if (k0 >= k1) return;
#pragma unroll 2
                for (int k = k0; k < k1; k++) {
                    double av[4], bv2[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) { av[q] = S[base_a + 16 * k + 4 * q]; bv2[q] = S[base_b + 16 * k + 4 * q]; }
#pragma unroll
                    for (int q = 0; q < 4; q++) { av[q] = va ? av[q] : 0.0; bv2[q] = vb ? bv2[q] : 0.0; }
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv2[0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv2[1], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv2[2], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv2[3], acc1, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = 16 * ti + (lane >> 4) + 4 * r, col = 16 * tj + (lane & 15);
                    if (row <= RC && col <= row && col < RC) S[pk(row, col)] -= acc0[r] + acc1[r];
                }
            };
            const bool chain_ok = !CH || uni(s_flag[1]) != 0;
            for (int j0 = 0; j0 < RC && chain_ok; j0 += 16) {
                const int nb = min(16, RC - j0), jb = j0 >> 4;
#ifdef GF_PROFILE_STEP
                const long long w0c = clock64();
#endif
                // wavefront 0 runs the serial part (its diagonal tile, then the factor + inverse) next to wavefront 4 on the same SIMD: without priority the two
                // alternate and the serial part takes twice its stand-alone time (3.5 k -> 6.4 k cycles)
                // The diagonal tile of the NEXT block column takes everything that is final already (block columns 0 .. jb - 1) from wavefront 7 during this
                // step -- it has the fewest tiles of the column, none in the later steps where this job is longest --, so that the serial wavefront only adds
                // the last block column (the one whose panel is not done yet) before it factors: its share per step no longer grows with the column index.
                if (wave == 0) {
                    __builtin_amdgcn_s_setprio(3);
                    if (jb > 0) { upd_tile(jb, jb, jb - 1, jb); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
                    diag_block(j0);
                    __builtin_amdgcn_s_setprio(0);
                } else {
                    if (jb > 0) for (int t = wave; t < NTR - jb; t += NW - 1) upd_tile(jb + t, jb, 0, jb);
                    if (wave == NW - 1 && jb + 1 < NTR && 16 * (jb + 1) < RC) upd_tile(jb + 1, jb + 1, 0, jb);
                }
#ifdef GF_PROFILE_STEP
                if (blockIdx.x == 0 && sb.stamps && (tid == 0 || tid == 64 || tid == 256) && jb < 12) sb.stamps[(tid == 0 ? 40 : tid == 64 ? 56 : 72) + jb] = clock64() - w0c;
#endif
                __syncthreads();
                GF_SUB(tC);
                if (!uni(s_flag[1])) break;
                // panel: X = A21 L11^-T for the rows below the block and the rhs row; 16-row tiles, X[i][c] = sum_k A[i][k] Linv[c][k]
                const int r0 = j0 + nb;
                auto panel = [&](auto full) {
                    constexpr bool FULL = decltype(full)::value;
                    const int nrows = RC + 1 - r0, nt = (nrows + 15) / 16;
                    for (int t = wave; t < nt; t += NW) {
                        const int ra = r0 + 16 * t + (lane & 15);
                        const int rb_ = pk(min(ra, RC), j0);
                        d4 acc = {0, 0, 0, 0};
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int kc = 4 * k + (lane >> 4);
                            const double av = (ra <= RC && (FULL || kc < nb)) ? S[rb_ + kc] : 0.0;
                            const double bv2 = s_inv[(lane & 15) * 17 + kc];          // B[k][c] = Linv[c][k]
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv2, acc, 0, 0, 0);
                        }
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int row = r0 + 16 * t + (lane >> 4) + 4 * r, c = lane & 15;
                            if (row <= RC && (FULL || c < nb)) S[pk(row, j0 + c)] = acc[r];
                        }
                    }
                };
                if (nb == 16) panel(std::true_type{}); else panel(std::false_type{});
                __syncthreads();
                GF_SUB(tB);
            }
#ifdef GF_PROFILE_STEP
            if (blockIdx.x == 0 && tid == 0 && sb.stamps) { sb.stamps[20] = tA; sb.stamps[21] = tB; sb.stamps[22] = tC; }
#endif
            ok = uni(s_flag[1]) != 0;
            if (ok) {
                GF_STAMP(9);
                // backward substitution L^T y = z (z = row R), 16-column blocks from the bottom.  Wavefront 0 runs the serial chain alone: it completes the
                // block's right-hand side, solves inside the block (the diagonal block holds the inverse of its factor: a 16 x 16 product), and applies the
                // block's solution to the NEXT block's sixteen entries itself; the other wavefronts apply it to everything below that, one block behind
                // and into an accumulator of their own (s_rd), so that no entry is updated from both sides.  One barrier per block instead of two, and the
                // chain no longer waits for the wide update (11 x 3.8 k -> 11 x ~1.7 k cycles).
                for (int r = tid; r < RC; r += NT) s_rd[r] = 0.0;
                __syncthreads();
                for (int j0 = ((RC - 1) / 16) * 16, par = 0; j0 >= 0; j0 -= 16, par ^= 1) {
                    const int nb = min(16, RC - j0);
#ifdef GF_PROFILE_STEP
                    const long long bq0 = clock64();
#endif
                    if (wave == 0) {
                        __builtin_amdgcn_s_setprio(3);
                        // every load below is unconditional, from an index clamped into the block, and masked afterwards: predicated loads compiled into a branch each
                        // (700 instructions per block for ~150 of arithmetic)
                        const int cc = lane & 15, lz = min(lane, nb - 1);
                        double z = S[pk(RC, j0 + lz)] + s_rd[j0 + lz];
                        z = lane < nb ? z : 0.0;
                        double wc[16];   // column cc of the block's inverse factor (stored in place of the factor): y_c = sum_{r >= c} Linv[r][c] z_r
                        double lnext[16];   // rows of the block at the columns of the next block: L[j0 + c][j0 - 16 + lane]; rows beyond the block meet a zero solution entry
                        const int rn = max(j0 - 16, 0) + cc;
                        if (nb == 16) {   // every block but the last one of the system: no condition on a uniform value inside (each one became a branch with its own wait)
#pragma unroll
                            for (int rr = 0; rr < 16; rr++) { const double v = S[pk(j0 + rr, j0) + min(cc, rr)]; wc[rr] = cc <= rr ? v : 0.0; }
#pragma unroll
                            for (int c = 0; c < 16; c++) lnext[c] = S[pk(j0 + c, 0) + rn];
                        } else {
#pragma unroll
                            for (int rr = 0; rr < 16; rr++) {
                                const int rl = min(rr, nb - 1);
                                const double v = S[pk(j0 + rl, j0 + min(cc, rl))];
                                wc[rr] = (rr < nb && cc <= rr) ? v : 0.0;
                            }
#pragma unroll
                            for (int c = 0; c < 16; c++) lnext[c] = S[pk(j0 + min(c, nb - 1), rn)];
                        }
                        const double zn = S[pk(RC, rn)];
                        double y0 = 0.0, y1 = 0.0;
#pragma unroll
                        for (int rr = 0; rr < 16; rr += 2) { y0 += wc[rr] * row_bcast(z, rr); y1 += wc[rr + 1] * row_bcast(z, rr + 1); }
                        z = y0 + y1;
                        if (lane < nb) { s_y[16 * par + lane] = z; if (CH) s_yd[j0 + lane] = z; else yv[j0 + lane] = z; }
                        if (lane >= nb && lane < 16) s_y[16 * par + lane] = 0.0;
                        if (j0 > 0) {
                            double sv = zn;
#pragma unroll
                            for (int c = 0; c < 16; c++) sv -= lnext[c] * row_bcast(z, c);
                            if (lane < 16) S[pk(RC, rn)] = sv;
                        }
                        __builtin_amdgcn_s_setprio(0);
                    }
#ifdef GF_PROFILE_STEP
                    const long long bq1 = clock64();
#endif
                    __syncthreads();
#ifdef GF_PROFILE_STEP
                    const long long bq2 = clock64();
                    if (blockIdx.x == 0 && tid == 0 && sb.stamps) { sb.stamps[29] += bq1 - bq0; sb.stamps[30] += bq2 - bq1; }
                    if (blockIdx.x == 0 && tid == 64 && sb.stamps) { sb.stamps[31] += bq2 - bq1; }
#endif
                    if (wave > 0) {
                        const double* yb = s_y + 16 * par;
                        for (int r = tid - 64; r < j0 - 16; r += NT - 64) {
                            double lv[16];
#pragma unroll
                            for (int c = 0; c < 16; c++) lv[c] = c < nb ? S[pk(j0 + c, r)] : 0.0;
                            double acc = 0.0;
#pragma unroll
                            for (int c = 0; c < 16; c++) acc += lv[c] * yb[c];
                            s_rd[r] -= acc;
                        }
                    }
                }
                __syncthreads();
#ifdef GF_PROFILE_STEP
                if (CH && blockIdx.x == 0 && tid == 0 && sb.stamps) sb.stamps[91] = clock64();
#endif
                if (CH) {
                    // the dense part's solution sits in s_yd.  Backward through the chain, in reverse elimination order:  y_k = W_k^T (z_k - Y_k[D]^T y_D - E_k^T y_{k-1}),
                    // z_k = the right-hand-side row of the finished panel.  Everything that does not depend on the chain first, by all wavefronts: W_k, E_k into LDS (the panels'
                    // space is free now), z_k - Y_k[D]^T y_D, then  a_k = W_k^T (z_k - t_k)  and  B_k = W_k^T E_k^T;  what is left for one wavefront are eleven 9 x 9
                    // matrix-vector products  y_k = a_k - B_k y_{k-1}.
                    const int PR = ch_panel_rows(ND);
                    double* sWE = smem + ((((size_t)(ND + 1) * (ND + 2) / 2) + 1) & ~(size_t)1);   // [NP][162] W_k, E_k; then B [NP][81]
                    double* sB = sWE + CNP * 162;
                    double* szt = sred; double* sa_ = sred + 128;   // [NP][9] each: z_k - t_k, a_k (sred is free here)
                    const double* Yg = sb.Yg + (size_t)b * CNP * sb.YgStride;
                    {   // 162 doubles per frame, <= 8 per thread at NP = 11: all loads before the first store
                        double wv[8];
#pragma unroll
                        for (int j = 0; j < 8; j++) { const int q = min(tid + NT * j, CNP * 162 - 1), k = q / 162, o = q - 162 * k; wv[j] = Yg[(size_t)k * sb.YgStride + (size_t)PR * kChYS + o]; }
#pragma unroll
                        for (int j = 0; j < 8; j++) { const int q = tid + NT * j; if (q < CNP * 162) sWE[q] = wv[j]; }
                        for (int q = tid + NT * 8; q < CNP * 162; q += NT) { const int k = q / 162, o = q - 162 * k; sWE[q] = Yg[(size_t)k * sb.YgStride + (size_t)PR * kChYS + o]; }
                    }
                    for (int k = wave; k < CNP; k += NW) {
                        const int c = lane & 15, part = lane >> 4, cc = min(c, 8);
                        const double* Yk = Yg + (size_t)k * sb.YgStride;
                        const int r0 = k == 0 ? 0 : ((6 * (k - 1)) >> 4) << 4;   // the panel's rows in front of its first tile were never written (they are zero by structure)
                        double acc = 0.0;
                        {   // <= 24 rows per lane (ND <= 96): every load issued before the first use (one at a time this was a chain of L2 round trips)
                            double yv_[24];
#pragma unroll
                            for (int q = 0; q < 24; q++) { const int dj = part + 4 * q; yv_[q] = Yk[(size_t)min(max(dj, r0), ND - 1) * kChYS + cc]; }
#pragma unroll
                            for (int q = 0; q < 24; q++) { const int dj = part + 4 * q; if (dj >= r0 && dj < ND && c < 9) acc += yv_[q] * s_yd[dj]; }
                        }
                        acc += __shfl_xor(acc, 16); acc += __shfl_xor(acc, 32);
                        if (lane < 9) szt[k * 9 + lane] = Yk[(size_t)ND * kChYS + lane] - acc;
                    }
                    __syncthreads();
                    for (int q = tid; q < CNP * 90; q += NT) {   // B_k[c][b] = sum_a W_k[a][c] E_k[b][a] (q % 90 < 81), a_k[c] = sum_a W_k[a][c] (z_k - t_k)[a] (the other nine)
                        const int k = q / 90, o = q - 90 * k;
                        const double* Wk = sWE + k * 162; const double* Ek = Wk + 81;
                        double v = 0.0;
                        if (o < 81) { const int c = o / 9, bb = o - 9 * c; for (int a2 = 0; a2 < 9; a2++) v += Wk[a2 * 9 + c] * Ek[bb * 9 + a2]; sB[k * 81 + o] = v; }
                        else { const int c = o - 81; for (int a2 = 0; a2 < 9; a2++) v += Wk[a2 * 9 + c] * szt[k * 9 + a2]; sa_[k * 9 + c] = v; }
                    }
                    __syncthreads();
                    if (wave == 0) {
                        const int c = lane & 15, cc = min(c, 8);   // the four 16-lane rows compute the same thing: every broadcast stays inside a DPP row
                        double yprev = 0.0;
                        for (int k = 0; k < CNP; k++) {
                            double brow[9];
#pragma unroll
                            for (int a2 = 0; a2 < 9; a2++) brow[a2] = sB[k * 81 + cc * 9 + a2];
                            double y = sa_[k * 9 + cc];
                            if (k > 0) {
#pragma unroll
                                for (int a2 = 0; a2 < 9; a2++) y -= brow[a2] * row_bcast(yprev, a2);
                            }
                            yprev = c < 9 ? y : 0.0;
                            if (lane < 9) yv[15 * k + 6 + lane] = y;
                        }
                    }
                    for (int dj = tid; dj < ND; dj += NT) yv[ch_d2c(dj, CNP)] = s_yd[dj];
                    __syncthreads();
                }
                GF_STAMP(10);
                // back-substitute the eliminated columns (one wavefront per row), check finiteness
                double bad = 0;
                for (int c = tid; c < R; c += NT) if (!isfinite(yv[c])) bad = 1;
                if (tid < ECW) { const int c = s_cmap[tid]; s_uc[tid] = (c >= 0 && c < R) ? yv[c] : 0.0; }
                __syncthreads();
                {   // sixteen lanes per eliminated column (four columns per wavefront instruction, two such groups in flight): acc_e = sum_k Es[e][k] y_k with
                    // Es = f_e s_c Et scaled on the fly (the right-hand-side slot does not take part: its y entry is zero), and -- once per call -- the Cauchy
                    // direction's dot product with the unscaled row; both reduced inside the 16-lane row by DPP
                    constexpr int EM = GS ? 9 : 5;   // compact columns per lane: ECW <= 16 EM (6 (W + 1) + 8 padded to 16; W <= 10 in LDS, <= 20 in global memory)
                    const int sub = lane & 15, grp = lane >> 4;
                    for (int e0 = 4 * wave + grp; e0 < NE; e0 += 8 * NW) {
                        double etv[2][EM], fe[2];
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const int e = e0 + 4 * NW * h;
                            const bool live = e < NE;
                            fe[h] = live ? u[RP + e] : 0.0;
                            const double* src = Et + (size_t)min(e, NE - 1) * ECW;
#pragma unroll
                            for (int q = 0; q < EM; q++) { const int k = sub + 16 * q; etv[h][q] = (live && k < ECW) ? src[k] : 0.0; }
                        }
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const int e = e0 + 4 * NW * h;
                            double accy = 0.0, accu = 0.0;
#pragma unroll
                            for (int q = 0; q < EM; q++) {
                                const int k = sub + 16 * q;
                                if (k < ECW) {
                                    const int c = s_cmap[k];
                                    if (c >= 0 && c < R) { accy += (fe[h] * s_hd[c] * etv[h][q]) * s_uc[k]; accu += etv[h][q] * s_ucc[k]; }
                                }
                            }
                            accy = row_sum16_f64(accy);
                            if (need_alpha) accu = row_sum16_f64(accu);
                            if (sub == 0 && e < NE) {
                                const double et = yv[RP + e];
                                const double ye = (scale[RP + e] * etb[e] - accy * sqrt(et)) / et;
                                if (need_alpha) uHu_acc += 2.0 * gn[RP + e] * accu;   // gn's tail still holds the Cauchy direction's eliminated part here
                                gn[RP + e] = -diag[RP + e] * ye;
                                if (!isfinite(ye)) bad = 1;
                            }
                        }
                    }
                }
                if (need_alpha) {
                    const double uHu = block_sum<NW>(uHu_acc, sred, tid, NT);
                    if (tid == 0) st.alpha = gsq / uHu;
                    need_alpha = false;
                }
                bad = block_max<NW>(bad, sred, tid, NT);
                if (bad > 0) ok = false;
            }
            if (!ok) { __syncthreads(); if (tid == 0) st.mu *= 10.0; __syncthreads(); }
        }
        if (tid == 0) s_flag[2] = ok ? 1 : 0;
        __syncthreads();
        if (ok) for (int c = tid; c < R; c += NT) gn[c] = -diag[c] * yv[c];
        __syncthreads();
        if (tid == 0) { st.reuse = 1; st.mu_solved = st.mu; }
    } else if (tid == 0) s_flag[2] = 1;
    __syncthreads();
    bool valid = s_flag[2] != 0;
    GF_STAMP(11);
    // ---------------- traditional dogleg interpolation (dogleg_strategy.cc ComputeTraditionalDoglegStep)
    if (valid) {
        double a = 0, c2 = 0, dt = 0;
        for (int c = tid; c < R; c += NT) { a += grad[c] * grad[c]; c2 += gn[c] * gn[c]; dt += grad[c] * gn[c]; }
        for (int e = tid; e < NE; e += NT) { a += grad[RP + e] * grad[RP + e]; c2 += gn[RP + e] * gn[RP + e]; dt += grad[RP + e] * gn[RP + e]; }
        double v3[3] = {a, c2, dt};
        block_sum_n<3, NW>(v3, sred, tid);
        const double gnorm = sqrt(v3[0]), gnn = sqrt(v3[1]), gdot = v3[2];
        const double radius = st.radius, alpha = st.alpha;
        double ca, cb, dsn;  // step = ca * grad + cb * gn
        if (gnn <= radius) { ca = 0; cb = 1; dsn = gnn; }
        else if (gnorm * alpha >= radius) { ca = -(radius / gnorm); cb = 0; dsn = radius; }
        else {
            const double b_dot_a = -alpha * gdot, a_sq = pow(alpha * gnorm, 2.0), bma = a_sq - 2 * b_dot_a + pow(gnn, 2.0);
            const double cc = b_dot_a - a_sq, dd = sqrt(cc * cc + bma * (pow(radius, 2.0) - a_sq));
            const double beta = (cc <= 0) ? (dd - cc) / bma : (radius * radius - a_sq) / (dd + cc);
            ca = -alpha * (1.0 - beta); cb = beta; dsn = -1;
        }
        double nn = 0;
        for (int c = tid; c < R; c += NT) { const double sv = ca * grad[c] + cb * gn[c]; nn += sv * sv; stepv[c] = sv / diag[c]; u[c] = scale[c] * stepv[c]; }
        for (int e = tid; e < NE; e += NT) { const double sv = ca * grad[RP + e] + cb * gn[RP + e]; nn += sv * sv; stepv[RP + e] = sv / diag[RP + e]; u[RP + e] = scale[RP + e] * stepv[RP + e]; }
        nn = block_sum<NW>(nn, sred, tid, NT);
        if (dsn < 0) dsn = sqrt(nn);
        if (tid == 0) st.dogleg_step_norm = dsn;
        GF_STAMP(12);
        // model_cost_change = -(J s)^T (r + J s / 2) = -(u^T g + u^T H u / 2), u = scale .* step (trust_region_minimizer.cc), without another
        // sweep over H: u = ca uC + cb uG with uC the Cauchy direction and uG = -s y the Gauss-Newton step, and (s H s + mu D^2) y = s g gives
        //   uC^T g = |g~|^2,  uG^T g = g~.gn,  uC^T H uC = |g~|^2 / alpha,  uC^T H uG = -|g~|^2 - mu g~.gn,  uG^T H uG = -g~.gn - mu |gn|^2
        // (g~ = grad, gn in dogleg space; all three sums are already reduced above)
        __syncthreads();
        const double mu_s = st.mu_solved;
        const double ug = ca * v3[0] + cb * gdot;
        const double uHu = ca * ca * (v3[0] / alpha) + 2.0 * ca * cb * (-v3[0] - mu_s * gdot) + cb * cb * (-gdot - mu_s * v3[1]);
        const double mcc = -(ug + 0.5 * uHu);
        if (tid == 0) st.model_cost_change = mcc;
        valid = mcc > 0.0;
    }
    __syncthreads();
    GF_STAMP(13);
    // ---------------- candidate point x (+) delta, delta = step .* scale = u  (its normal equations are written, not accumulated, by the next sweeps)
    double* xc = w.xs + ((size_t)(1 - cur) * d.B + b) * d.XS;
    for (int i = tid; i < d.XS; i += NT) xc[i] = xs[i];
    __syncthreads();
    double sn = 0, xn = 0;
    if (valid) {
        for (int blk = tid; blk < d.NFB + d.F; blk += NT) {
            int c0, off, kind;
            if (blk < d.NFB) {
                c0 = colf[blk];
                if (blk < 2 * d.NP) { kind = (blk & 1) ? 1 : 0; off = (blk & 1) ? off_sb(blk >> 1) : off_pose(blk >> 1); }
                else if (blk < 2 * d.NP + 7) { const int q = blk - 2 * d.NP; kind = q == 0 ? 2 : q == 1 ? 3 : q <= 4 ? 4 : q == 5 ? 7 : 8; off = q == 0 ? off_ex(d.NP) : q == 1 ? off_exw(d.NP) : q <= 4 ? off_ix(d.NP) + (q - 2) : q == 5 ? off_td(d.NP) : off_tdw(d.NP); }
                else { const int q = blk - (2 * d.NP + 7); kind = q < 5 * d.NP + 1 ? 10 : 13; off = d.GO + q; }   // rcv_dt, rcv_ddt, yaw: scalars in state order; then anc_ecef (3)
            } else {
                const int f = blk - d.NFB;
                if (f >= w.nfeat[b]) continue;
                const int e = cole[f];
                c0 = e >= 0 ? RP + e : -1; kind = 9; off = off_feat(d.NP) + f;
            }
            if (c0 < 0) continue;
            const int gs = gsize_kind(kind);
            if (gs == 7) {
                if (kind == 2 || kind == 3) {   // PoseSubsetParameterization::Plus (pose_subset_parameterization.cpp:27-56): masked components of the increment are dropped -- here only;
                                                // the columns, the step and the model cost change are those of the full block (its ComputeJacobian is the identity whatever the mask)
                    const int mask = (int)w.wpar[WPAR * b + (kind == 2 ? 4 : 5)];
                    double dm[6];
#pragma unroll
                    for (int q = 0; q < 6; q++) dm[q] = ((mask >> q) & 1) ? 0.0 : u[c0 + q];
                    pose_plus(xs + off, dm, xc + off);
                } else pose_plus(xs + off, u + c0, xc + off);
            }
            else for (int q = 0; q < gs; q++) xc[off + q] = xs[off + q] + u[c0 + q];
            for (int q = 0; q < gs; q++) { const double dv = xs[off + q] - xc[off + q]; sn += dv * dv; xn += xs[off + q] * xs[off + q]; }
        }
    }
    { double v2[2] = {sn, xn}; block_sum_n<2, NW>(v2, sred, tid); sn = v2[0]; xn = v2[1]; }
    GF_STAMP(14);
    if (tid == 0) {
        st.step_norm = sqrt(sn); st.x_norm = sqrt(xn);
        st.cand_valid = valid ? 1 : 0;
        if (valid) st.invalid_run = 0;
        else {  // HandleInvalidStep
            st.last_successful = 0;
            if (++st.invalid_run >= 5) { st.done = 1; st.termination = 4; }
            st.mu *= 10.0; st.reuse = 0;
        }
    }
}
template <bool GS, int NW = 8>
__global__ void __launch_bounds__(64 * NW) ba_step(Win w, StepBufs sb, int first, int max_iters, int finalize_only) { ba_step_body<GS, NW>(w, sb, first, max_iters, finalize_only); }
// the chain form: four wavefronts, at most 256 registers, < 80 KB of LDS -- two windows per CU (see the note in front of ba_step_body)
__global__ void __launch_bounds__(256, 2) ba_step_chain(Win w, StepBufs sb, int first, int max_iters, int finalize_only) { ba_step_body<false, 4, true>(w, sb, first, max_iters, finalize_only); }
// The prior / IMU / wheel sweep of the candidate and the step that judges it in one launch (LDS-resident systems without GNSS blocks): the sweep's H, g and costs
// are read by the same block right away, and the launch boundary between the two -- with the write-back of everything the sweep stored -- is gone.
__global__ void __launch_bounds__(512) ba_misc_step(Win w, StepBufs sb, int max_iters, int finalize_only) {
    ba_linearize_misc_body(w, -1, -1, 1, 0);
    __syncthreads();
    ba_step_body<false>(w, sb, 0, max_iters, finalize_only);
}

}  // namespace gfb
