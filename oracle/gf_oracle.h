// TEST INFRASTRUCTURE — C interface of the CPU oracle (see tracker_oracle.cpp / backend_oracle.cpp headers).
// Not part of the product; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gfo_tracker_cfg {
    int max_cnt;      // MAX_CNT   (config/realsense/m2dgrp.yaml:131)
    int min_dist;     // MIN_DIST  (:132)
    int flow_back;    // FLOW_BACK (:136)
    int depth_cam;    // FeatureTracker::depth_cam
    double fx, fy, cx, cy, k1, k2, p1, p2;  // pinhole intrinsics (config/realsense/wt_cam.yaml)
} gfo_tracker_cfg;

void* gfo_tracker_create(const gfo_tracker_cfg* cfg);
void gfo_tracker_destroy(void* h);
// returns number of features; out_obs = n x 8 doubles (x,y,1,u,v,vx,vy,depth), feature_tracker.cpp:344-368
int gfo_tracker_track(void* h, double t, const uint8_t* img, int w, int hh, int stride, const uint16_t* depth, int dstride,
                      int* out_ids, double* out_obs, int cap);
void gfo_tracker_set_prediction(void* h, const int* ids, const double* xyz, int n);
void gfo_tracker_remove_outliers(void* h, const int* ids, int n);
int gfo_tracker_state(void* h, int* ids, int* track_cnt, float* prev_pts, int cap);
long long gfo_tracker_lk_iters(void* h);
/* 0: int64 LK sums (parity mode); 1: float-lane accumulation of an x86 OpenCV build (sensitivity measurement only, see tracker_oracle.cpp) */
void gfo_set_lk_accum(int mode);
/* CPU-baseline variant (b) of BASELINE.md section 2: n > 1 runs the LK point loop on n threads (OpenCV's parallel_for_ over points; results unchanged)
 * and the marginalisation's A / b construction on 4 threads, factors dealt round-robin and the four partial systems added in thread order
 * (marginalization_factor.cpp:150-181, :232-262).  n = 1 (default): one thread, factors summed in list order. */
void gfo_set_threads(int n);
int gfo_get_threads(void);

void gfo_pyr_down(const uint8_t* src, int w, int h, uint8_t* dst);
void gfo_scharr(const uint8_t* src, int w, int h, int16_t* dst);
void gfo_lk(const uint8_t* prev, const uint8_t* next, int w, int h, const float* prevPts, float* nextPts, uint8_t* status, int n,
            int maxLevel, int maxCount, double eps, int useInitialFlow, long long* iters);
void gfo_fill_circle(uint8_t* img, int w, int h, int cx, int cy, int radius, int color);
void gfo_min_eigen_val(const uint8_t* img, int w, int h, float* eig);
int gfo_good_features(const uint8_t* img, int w, int h, float* corners, int maxCorners, double quality, double minDist, const uint8_t* mask);

#ifdef __cplusplus
}
#endif

/* ------------------------------------------------------------------ back end (window solve) */
#ifdef __cplusplus
extern "C" {
#endif
/* Parameter-block ids used by priors: kind * 4096 + index */
enum { GFO_POSE = 0, GFO_SPEEDBIAS = 1, GFO_EX_POSE = 2, GFO_EX_WHEEL = 3, GFO_SX = 4, GFO_SY = 5, GFO_SW = 6, GFO_TD = 7, GFO_TD_WHEEL = 8, GFO_FEATURE = 9,
       GFO_RCV_DT = 10, GFO_RCV_DDT = 11, GFO_YAW = 12, GFO_ANC = 13 };

typedef struct gfo_window {
    int W, n_feature, n_visual, n_imu, n_wheel;
    int fix_ex_pose, fix_ex_wheel, fix_ix, fix_td, fix_td_wheel, fix_poses;
    double G[3];
    double vis_sqrt_info;
    double* para_Pose; double* para_SpeedBias; double* para_Ex_Pose; double* para_Ex_Pose_wheel; double* para_Ix; double* para_Td;
    double* para_Td_wheel; double* para_Feature;
    const unsigned char* feature_fixed;
    const int* vis_feature; const int* vis_i; const int* vis_j;
    const double* vis_pts_i; const double* vis_pts_j; const double* vis_vel_i; const double* vis_vel_j; const double* vis_td_i; const double* vis_td_j;
    const int* imu_i; const double* imu_sum_dt; const double* imu_delta_p; const double* imu_delta_q; const double* imu_delta_v;
    const double* imu_lin_ba; const double* imu_lin_bg; const double* imu_jacobian; const double* imu_covariance;
    const int* wh_i; const double* wh_sum_dt; const double* wh_delta_p; const double* wh_delta_q; const double* wh_jacobian;
    const double* wh_covariance; const double* wh_lin; const double* wh_lin_vel; const double* wh_lin_gyr; const double* wh_vel_1; const double* wh_gyr_1;
    int prior_n, prior_nblocks;
    const int* prior_block_id; const double* prior_J; const double* prior_r; const double* prior_x0;
    /* GNSS (estimator.cpp:2904-2941, :3178-3229, :3390-3431): blocks and factors.  gnss_enabled = gnss_ready; the factors enter the solve
     * unless gnss_lowspeed (estimator.cpp:3178), and enter the MARGIN_OLD marginalisation whenever gnss_enabled (:3390).
     * gnss_data per factor (16 doubles): sv_pos 3, sv_vel 3, svdt, svddt, tgd, pr_uura, dp_uura, psr, dopp, wavelength, time of GPS week [s], 0.
     * What GnssPsrDoppFactor's constructor derives from observation + ephemeris (gnss_psr_dopp_factor.cpp:3-47) is handed over precomputed. */
    int gnss_enabled, gnss_lowspeed, n_gnss, has_anchor;
    double* para_rcv_dt;         /* 4 (W+1), in/out */
    double* para_rcv_ddt;        /* (W+1) */
    double* para_yaw_enu_local;  /* 1 (held constant, estimator.cpp:2932) */
    double* para_anc_ecef;       /* 3 */
    double gnss_ddt_weight;      /* GNSS_DDT_WEIGHT, parameters.cpp:549 */
    double anchor_value[7];      /* PoseAnchorFactor on Pose[0] (estimator.cpp:2943-2951), sqrt_info 120 */
    const double* gnss_iono;     /* 8 Klobuchar parameters */
    const int* gnss_frame;       /* i: the factor uses rcv_dt[4 i + sys] and rcv_ddt[i] */
    const int* gnss_lower;       /* lower_idx: the factor sits on (Pose, SpeedBias)[lower_idx] and [lower_idx + 1] */
    const int* gnss_sys;         /* sys_idx 0..3 */
    const double* gnss_ratio;    /* ts_ratio */
    const double* gnss_data;     /* n_gnss x 16 */
    const double* gnss_headers;  /* Headers[0..W]: DtDdtFactor(Headers[i+1] - Headers[i]), estimator.cpp:3214-3223 */
    /* PoseSubsetParameterization constancy masks of the camera / wheel extrinsic (pose_subset_parameterization.cpp:10-56; estimator.cpp:2969-2985, :3010-3026):
     * bit q set = increment component q is zeroed in Plus */
    int ex_pose_mask, ex_wheel_mask;
} gfo_window;

typedef struct gfo_summary {
    int iterations, successful_steps, termination; /* 0 max iterations, 1 function tol, 2 parameter tol, 3 gradient tol, 4 failure */
    double initial_cost, final_cost;
    double radius;
} gfo_summary;

/* Estimator::optimization() lines estimator.cpp:2890-3327: ceres::Solve with DENSE_SCHUR + DOGLEG, max_iters iterations */
int gfo_ba_solve(gfo_window* w, int max_iters, gfo_summary* s);
/* estimator.cpp:3334-3631: mode 0 = MARGIN_OLD, 1 = MARGIN_SECOND_NEW. Outputs the next prior (block ids already address-shifted). */
int gfo_ba_marginalize(const gfo_window* w, int mode, int cap_n, int* out_n, int* out_nblocks, int* out_block_id, double* out_J, double* out_r,
                       double* out_x0, int* out_m);
/* test helper: the marginalisation's assembled system A (pos x pos, row-major; dropped columns first), b and its Schur complement A_r, b_r */
int gfo_ba_marg_system(const gfo_window* w, int mode, int cap, double* A, double* b, double* Ar, double* br, int* pos, int* m, int* n);
/* factor evaluation for unit tests: kind 0 visual k, 1 imu k, 2 wheel k, 3 GnssPsrDopp k, 4 DtDdt (k = 4 i + sys), 5 DdtSmooth k, 6 PoseAnchor; returns residuals and the dense Jacobian wrt the factor's
 * parameter blocks in GLOBAL size (row-major, blocks concatenated), as ceres::CostFunction::Evaluate fills them */
int gfo_factor_eval(const gfo_window* w, int kind, int k, double* residuals, double* jacobians, int* nres, int* ncols);
/* IntegrationBase::push_back loop (integration_base.h:39-167); noise = ACC_N, GYR_N, ACC_W, GYR_W */
void gfo_imu_preintegrate(int n, const double* dt, const double* acc, const double* gyr, const double* acc0, const double* gyr0, const double* ba,
                          const double* bg, const double* noise, double* delta_p, double* delta_q, double* delta_v, double* jacobian, double* covariance,
                          double* sum_dt);
/* WheelIntegrationBase::push_back loop (wheel_integration_base.h:41-178); noise = VEL_N_wheel, GYR_N_wheel; lin = sx, sy, sw */
void gfo_wheel_preintegrate(int n, const double* dt, const double* vel, const double* gyr, const double* vel0, const double* gyr0, const double* lin,
                            const double* noise, double* delta_p, double* delta_q, double* jacobian, double* covariance, double* sum_dt);
void gfo_sym_eig(int n, const double* A, double* d, double* V);
void gfo_double2vector(int W, const double* R0_before, const double* P0_before, const double* para_Pose, const double* para_SpeedBias, double* Rs, double* Ps,
                       double* Vs, double* Bas, double* Bgs);
int gfo_ba_linearize(const gfo_window* w, int cap, double* H, double* g, double* cost, int* n_f, int* n_e, int* col_block_id);
#ifdef __cplusplus
}
#endif
