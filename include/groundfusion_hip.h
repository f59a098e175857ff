/* groundfusion_hip.h — C-ABI of the MI355X-native Ground-Fusion hot path (libgroundfusion_hip.so).
 *
 * Drop-in boundary.  The reference has no FFI layer: the boundary is the C++ class surface
 *   FeatureTracker::trackImage      vins_estimator/src/featureTracker/feature_tracker.h:47
 *   FeatureTracker::setPrediction   feature_tracker.h:70
 *   FeatureTracker::removeOutliers  feature_tracker.h:72
 *   Estimator::optimization         vins_estimator/src/estimator/estimator.h (called from processImage, estimator.cpp:1112)
 * Each entry point below names the reference interface it replaces.  Plain pointers and sizes only; no
 * exceptions cross this boundary; every function returns GF_OK (0) or a negative gf_status and records a
 * message retrievable with gf_last_error().  One handle drives `batch` independent sequences on one GPU
 * (batch = 1 reproduces the reference's one-FeatureTracker-per-process use); handles are not re-entrant
 * (same rule as the reference objects, estimator.cpp:209-239).
 */
#ifndef GROUNDFUSION_HIP_H
#define GROUNDFUSION_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gf_status {
    GF_OK = 0,
    GF_ERR_INVALID = -1,   /* bad argument */
    GF_ERR_NO_DEVICE = -2, /* no HIP device / HIP runtime failure: the product path never falls back to CPU */
    GF_ERR_HIP = -3,
    GF_ERR_CAPACITY = -4   /* caller buffer too small */
} gf_status;

const char* gf_last_error(void);
int gf_device_count(int* n);
int gf_set_device(int device);

/* ------------------------------------------------------------------ front end: FeatureTracker */
typedef struct gf_tracker gf_tracker;

typedef struct gf_tracker_cfg {
    int width, height;  /* ROW/COL, parameters.cpp:138ff (image_height/image_width) */
    int batch;          /* number of independent sequences driven by this handle (>=1) */
    int max_cnt;        /* MAX_CNT   config/realsense/m2dgrp.yaml:131 */
    int min_dist;       /* MIN_DIST  m2dgrp.yaml:132 */
    int flow_back;      /* FLOW_BACK m2dgrp.yaml:136 */
    int depth_cam;      /* FeatureTracker::depth_cam, feature_tracker.h:95 */
    double fx, fy, cx, cy, k1, k2, p1, p2; /* camodocal pinhole, config/realsense/wt_cam.yaml */
} gf_tracker_cfg;

/* One element of trackImage's return value map<int, vector<pair<int, Matrix<double,8,1>>>>
 * (feature_tracker.cpp:344-368): v = (x_n, y_n, 1, u, v, vx, vy, depth_m). */
typedef struct gf_feature_obs {
    int id;
    int camera_id;
    double v[8];
} gf_feature_obs;

typedef struct gf_tracker_stats {
    /* accumulated since the last gf_tracker_reset_stats(); times from hipEvents on the handle's stream */
    double ms_pyramid, ms_lk, ms_detect, ms_total_gpu;
    /* host wall-clock split of the same frames: enqueue/pack, wait for LK, setMask bookkeeping, wait for detector, pack output */
    double ms_host_pre, ms_wait_lk, ms_host_mid, ms_wait_detect, ms_host_post;
    long long frames;            /* frame-batches processed */
    long long lk_launches;       /* launches of the LK kernel */
    long long lk_points;         /* points submitted to LK */
    long long lk_level_passes;   /* sum over points of pyramid levels actually processed (fwd+reverse) */
    long long lk_iterations;     /* sum over points/levels of Gauss-Newton iterations */
    long long tracked_features;  /* features that survived tracking (status==1 after all checks) */
    long long output_features;   /* features returned to the caller (tracked + newly detected) */
} gf_tracker_stats;

int gf_tracker_create(const gf_tracker_cfg* cfg, gf_tracker** out);
int gf_tracker_destroy(gf_tracker* h);

/* Replaces FeatureTracker::trackImage(t, img, depth) for sequence `seq` (feature_tracker.h:47).
 * gray: height x width u8 (stride bytes); depth: height x width u16 millimetres (dstride elements) or NULL.
 * out/cap: caller-owned; *n_out = number of observations written (order = the tracker's `ids` vector). */
int gf_tracker_track(gf_tracker* h, int seq, double t, const uint8_t* gray, int stride, const uint16_t* depth,
                     int dstride, gf_feature_obs* out, int cap, int* n_out);

/* Same for all `batch` sequences in one pass; gray[b]/depth[b] are host images, out is [batch][cap].
 * On the host-image entry points (gf_tracker_track, _track_batch, _prefetch_batch / _track_prefetched) the depth image never crosses the bus: the reference reads one
 * pixel of it per feature (feature_tracker.cpp:360 `rightImg.at<ushort>(round(y), round(x))`), and those <= max_cnt samples are taken on the host from the caller's
 * image behind the kernels (round 5: a VGA RGB-D frame costs 307 KB of PCIe traffic instead of 921 KB). */
int gf_tracker_track_batch(gf_tracker* h, const double* t, const uint8_t* const* gray, int stride,
                           const uint16_t* const* depth, int dstride, gf_feature_obs* out, int cap, int* n_out);

/* The same boundary without serialising on the bus: gf_tracker_prefetch_batch starts the host -> device copy of the NEXT frame (a second pair of frame buffers,
 * a copy stream) and returns; gf_tracker_track_prefetched runs trackImage on the OLDEST staged frame as soon as its copy has landed.  Up to two frames can
 * be staged; call order: prefetch(0), then per frame k: prefetch(k + 1), track_prefetched(k) -- the copy of k + 1 runs under the kernels of k.  The host images -- the depth images in particular, which are sampled by track_prefetched itself -- must stay valid until the matching track_prefetched returns; only page-locked memory (gf_host_alloc,
 * or the caller's hipHostRegister) makes the copy overlap.  Images that lie back to back in one allocation travel as one copy per plane. */
/* One caller thread per tracker handle: prefetch / track_prefetched keep an unsynchronised two-slot FIFO, like every other gf_tracker_* entry point they must not
 * be called concurrently on the same handle. */
int gf_tracker_prefetch_batch(gf_tracker* h, const uint8_t* const* gray, int stride, const uint16_t* const* depth, int dstride);
int gf_tracker_track_prefetched(gf_tracker* h, const double* t, gf_feature_obs* out, int cap, int* n_out);
int gf_host_alloc(size_t bytes, void** out);
int gf_host_free(void* p);

/* Same with the images already resident in HBM: d_gray = batch contiguous height*width u8 images,
 * d_depth = batch contiguous height*width u16 images (or NULL).  Device pointers of the current device. */
int gf_tracker_track_batch_device(gf_tracker* h, const double* t, const void* d_gray, const void* d_depth,
                                  gf_feature_obs* out, int cap, int* n_out);

/* FeatureTracker::setPrediction (feature_tracker.cpp:1006-1027): ids[n], xyz[3n] camera-frame points. */
int gf_tracker_set_prediction(gf_tracker* h, int seq, const int* ids, const double* xyz, int n);
/* FeatureTracker::removeOutliers (feature_tracker.cpp:1029-1045) */
int gf_tracker_remove_outliers(gf_tracker* h, int seq, const int* ids, int n);
/* public members ids / track_cnt / prev_pts (feature_tracker.h:85-88) */
int gf_tracker_get_state(gf_tracker* h, int seq, int* ids, int* track_cnt, float* prev_pts_xy, int cap, int* n);

int gf_tracker_set_profiling(gf_tracker* h, int enable); /* hipEvent timing of kernels (default off) */
int gf_tracker_get_stats(gf_tracker* h, gf_tracker_stats* out);
int gf_tracker_reset_stats(gf_tracker* h);

/* ---- building blocks exposed for parity tests (device kernels on caller-provided HOST buffers) ---- */
/* cv::calcOpticalFlowPyrLK(prev,next,prevPts,nextPts,status,err,Size(21,21),maxLevel,
 *   TermCriteria(COUNT+EPS,30,0.01), useInitialFlow?OPTFLOW_USE_INITIAL_FLOW:0) as called at
 * feature_tracker.cpp:122,132,135,141.  pts are (x,y) float pairs. */
int gf_lk_track(const uint8_t* prev, const uint8_t* next, int width, int height, const float* prev_pts,
                float* next_pts, uint8_t* status, int n, int max_level, int use_initial_flow,
                long long* iterations);
/* cv::goodFeaturesToTrack(img, corners, max_corners, 0.01, min_dist, mask) as called at feature_tracker.cpp:198 */
int gf_good_features(const uint8_t* img, int width, int height, const uint8_t* mask, int max_corners,
                     int min_dist, float* corners_xy, int* n_out);
/* cornerMinEigenVal(img, eig, 3, 3) (inside goodFeaturesToTrack) */
int gf_min_eigen_val(const uint8_t* img, int width, int height, float* eig);
/* pyramid level l (0..3) of buildOpticalFlowPyramid and its Scharr derivative (interior only) */
int gf_pyramid_level(const uint8_t* img, int width, int height, int level, uint8_t* out, int16_t* deriv_xy);


/* profiling aid (no counterpart in the reference): kernels with a known byte count in the front end's access patterns, to calibrate rocprofv3's
 * FETCH_SIZE (scripts/pmc_collect.sh).  mode 0 streaming 16 B/lane, 1 LK 32x32 tile refill with 64-byte-aligned row segments, 2 the same straddling
 * two 64-byte lines.  Outputs: bytes the lanes requested, distinct 64-byte lines touched, kernel time. */
int gf_calib_fetch(int mode, size_t buffer_bytes, double* requested_bytes, double* lines64, double* ms);


/* ------------------------------------------------------------------ back end: Estimator::optimization() */
/* Replaces the Ceres problem built and solved in Estimator::optimization() (vins_estimator/src/estimator/estimator.cpp:2890-3327:
 * vector2double -> ceres::Solve(DENSE_SCHUR, DOGLEG, max_num_iterations) ) and the construction of the next marginalisation prior
 * (estimator.cpp:3334-3631, factor/marginalization_factor.cpp:119-308).  A window is handed over exactly as the reference hands it to
 * Ceres: the para_* arrays of vector2double (estimator.cpp:2276-2353) plus the constructor arguments of each factor. */
typedef struct gf_ba gf_ba;

/* parameter-block ids used by priors: kind * 4096 + index */
enum { GF_POSE = 0, GF_SPEEDBIAS = 1, GF_EX_POSE = 2, GF_EX_WHEEL = 3, GF_SX = 4, GF_SY = 5, GF_SW = 6, GF_TD = 7, GF_TD_WHEEL = 8, GF_FEATURE = 9,
       GF_RCV_DT = 10, GF_RCV_DDT = 11, GF_YAW = 12, GF_ANC = 13 };

typedef struct gf_ba_cfg {
    int window_size;   /* WINDOW_SIZE (parameters.h:24 fixes 10; here a runtime value) */
    int max_features;  /* capacity of para_Feature (NUM_OF_F, parameters.h:25) */
    int max_visual;    /* capacity of visual factors per window */
    int batch;         /* independent windows solved per call */
    int max_gnss;      /* capacity of GnssPsrDoppFactor per window; 0 builds the handle without the GNSS blocks (gnss_enable: 0) */
} gf_ba_cfg;

typedef struct gf_ba_window {
    int W, n_feature, n_visual, n_imu, n_wheel;
    /* SetParameterBlockConstant decisions of estimator.cpp:2990-3100, :3233-3246 */
    int fix_ex_pose, fix_ex_wheel, fix_ix, fix_td, fix_td_wheel, fix_poses;
    double G[3];              /* gravity, estimator.h `g` */
    double vis_sqrt_info;     /* FOCAL_LENGTH / 1.5 (estimator.cpp:193) */
    /* parameter blocks, in/out (estimator.h:335-341) */
    double* para_Pose;        /* (W+1) x 7: px py pz qx qy qz qw */
    double* para_SpeedBias;   /* (W+1) x 9 */
    double* para_Ex_Pose;     /* 7 */
    double* para_Ex_Pose_wheel; /* 7 */
    double* para_Ix;          /* sx, sy, sw */
    double* para_Td;          /* 1 */
    double* para_Td_wheel;    /* 1 */
    double* para_Feature;     /* n_feature inverse depths */
    const unsigned char* feature_fixed; /* estimate_flag == 1 -> constant (estimator.cpp:3291-3292) */
    /* ProjectionTwoFrameOneCamFactor(pts_i, pts_j, velocity_i, velocity_j, td_i, td_j) on (Pose[i], Pose[j], Ex_Pose, Feature[f], Td) */
    /* preconditions, checked at upload (GF_ERR_INVALID): i < j; all factors of a feature name the same start frame i and bring the same start-frame observation;
       a feature has at most one factor per frame j -- what estimator.cpp:3269-3297 builds (one factor per later observation, all from feature_per_frame[0]) */
    const int* vis_feature; const int* vis_i; const int* vis_j;
    const double* vis_pts_i; const double* vis_pts_j; const double* vis_vel_i; const double* vis_vel_j; const double* vis_td_i; const double* vis_td_j;
    /* IMUFactor(pre_integrations[i+1]) on (Pose[i], SpeedBias[i], Pose[i+1], SpeedBias[i+1]); delta_q as (w,x,y,z); 15x15 row-major */
    const int* imu_i; const double* imu_sum_dt; const double* imu_delta_p; const double* imu_delta_q; const double* imu_delta_v;
    const double* imu_lin_ba; const double* imu_lin_bg; const double* imu_jacobian; const double* imu_covariance;
    /* WheelFactor(pre_integrations_wheel[i+1]); jacobian 6x3, covariance 6x6 row-major; wh_lin = linearized sx, sy, sw, td */
    const int* wh_i; const double* wh_sum_dt; const double* wh_delta_p; const double* wh_delta_q; const double* wh_jacobian;
    const double* wh_covariance; const double* wh_lin; const double* wh_lin_vel; const double* wh_lin_gyr; const double* wh_vel_1; const double* wh_gyr_1;
    /* MarginalizationFactor(last_marginalization_info): linearized_jacobians (n x n row-major), linearized_residuals, keep_block_data */
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
    /* PoseSubsetParameterization (pose_subset_parameterization.cpp:10-56) of the camera / wheel extrinsic, estimator.cpp:2969-2985, :3010-3026:
     * bit q (0-2 translation, 3-5 rotation) set = component q of the block's increment is zeroed in Plus.  The block keeps its six columns and the
     * solver its full step (ComputeJacobian stays the identity: the reference's quirk); only the candidate point is masked.  0 = ADJUST_*_ALL.
     * Use gf_pose_subset_mask(extrinsic_type) for the YAML values. */
    int ex_pose_mask, ex_wheel_mask;
} gf_ba_window;

/* `extrinsic_type` / `extrinsic_type_wheel` of the YAML files (parameters.cpp:280-306, :394-420) -> constancy mask: 0 ALL {}, 1 TRANSLATION {3,4,5},
 * 2 ROTATION {0,1,2}, 3 NO_Z {2}, 4 NO_ROTATION_NO_Z {2,3,4,5}; any other value leaves the reference's zero-initialised enum = *_TRANSLATION. */
int gf_pose_subset_mask(int extrinsic_type);

typedef struct gf_ba_summary {
    int iterations, successful_steps, termination; /* 0 max iterations, 1 function tol, 2 parameter tol, 3 gradient tol, 4 failure */
    double initial_cost, final_cost, radius;
} gf_ba_summary;

/* caller-owned output of gf_ba_marginalize: J is n x n (row-major, capacity cap_n*cap_n), block ids already address-shifted */
typedef struct gf_ba_prior {
    int cap_n, cap_blocks;
    int n, nblocks, m, valid;
    int* block_id; double* J; double* r; double* x0;
} gf_ba_prior;

typedef struct gf_ba_stats {
    double ms_upload, ms_solve, ms_marginalize, ms_download;   /* hipEvent times accumulated over calls */
    double ms_jtj;                                             /* time inside the visual J^T J (MFMA) kernel */
    long long solves, jtj_launches, jtj_flops;                 /* jtj_flops: MFMA flops issued by that kernel */
    double ms_step;                                            /* time inside ba_step (one full dogleg step per solve is timed) */
    long long step_launches, step_flops;                       /* step_flops: Schur SYRK + Cholesky + substitutions of the timed launches */
    long long jtj_alg_flops;                                   /* algorithmic flops of the timed visual J^T J launches: Nv * 2 * 2 * 91 per window */
    double ms_jtj_contract;                                    /* split formulation (gf_ba_set_split_jtj): time inside the contraction-only MFMA kernel */
    long long jtj_contract_launches;
} gf_ba_stats;

int gf_ba_create(const gf_ba_cfg* cfg, gf_ba** out);
/* north_star's formulation of the visual sweep as a measured alternative (fixed camera extrinsic only): "residual/Jacobian sweep ... writing block-rows that an MFMA
 * J^T J contraction ... reduce[s]" -- the sweep stores every factor's 2 x 16 block row in HBM, a second kernel does nothing but contract them on the matrix cores.
 * Same bits as the fused kernel (same products, same order).  Off by default: it costs a 256-byte round trip per factor. */
int gf_ba_set_split_jtj(gf_ba* h, int on);
int gf_ba_destroy(gf_ba* h);
/* ceres::Solve on `count` <= batch windows (estimator.cpp:3303-3318 with max_solver_time disabled); states updated in place */
int gf_ba_solve(gf_ba* h, gf_ba_window* windows, int count, int max_iters, gf_ba_summary* summaries);
/* next prior; mode 0 = MARGIN_OLD (estimator.cpp:3334-3534), 1 = MARGIN_SECOND_NEW (:3536-3631) */
int gf_ba_marginalize(gf_ba* h, const gf_ba_window* windows, int count, int mode, gf_ba_prior* priors);
/* the same on windows that the preceding gf_ba_solve / gf_ba_upload left resident: only the parameter blocks (changed by double2vector's gauge fix,
 * estimator.cpp:3327-3337) are uploaded again, only the priors' n x n blocks come back.  slots[i]: position of windows[i] in the resident batch */
int gf_ba_marginalize_resident(gf_ba* h, const int* slots, const gf_ba_window* windows, int n, int mode, gf_ba_prior* priors);
/* priors == NULL above: the priors stay in the handle's host mirrors and every owner fetches its own (callable concurrently for different slots) */
int gf_ba_unpack_prior_slot(gf_ba* h, int slot, int mode, gf_ba_prior* prior);
/* throughput path: windows stay resident in HBM between calls */
int gf_ba_upload(gf_ba* h, const gf_ba_window* windows, int count);
int gf_ba_solve_resident(gf_ba* h, int max_iters, int marginalize_mode /* -1: none */, int reset_state);
/* same, returning as soon as the work is enqueued (the reference runs processImage on its own thread, estimator.cpp:209); gf_ba_wait joins */
int gf_ba_solve_resident_async(gf_ba* h, int max_iters, int marginalize_mode, int reset_state);
int gf_ba_wait(gf_ba* h);
int gf_ba_download(gf_ba* h, gf_ba_window* windows, int count, gf_ba_summary* summaries, gf_ba_prior* priors);
/* the newest pose (para_Pose[W]: px py pz qx qy qz qw) of the first `count` resident windows into a DEVICE array [count][7] -- the payload of the
 * per-step pose gather across GPUs (north_star; there is no counterpart in the single-process reference).  Enqueued behind a pending
 * asynchronous solve (complete after gf_ba_wait); otherwise complete on return. */
int gf_ba_export_newest_poses(gf_ba* h, void* d_out, int count);
/* The same solve for callers that own many windows on many threads (gf_estimator_group_*; no counterpart in the reference, whose Estimator owns one
 * window): gf_ba_pack_slot packs one window into slot `slot` of the handle's staging tables -- callable concurrently for different slots, one thread per
 * slot --, gf_ba_solve_packed closes the batch (upload, ceres::Solve on the listed slots -- slots not listed sit this batch out --, one download of all
 * states) and gf_ba_unpack_slot copies the solved state and the summary of a slot back into its owner's window (again callable concurrently for
 * different slots).  Results are identical to gf_ba_solve on the same windows.  gf_ba_marginalize_resident then takes the same slot numbers. */
/* Device-resident priors: a prior fetched with gf_ba_unpack_prior_slot(..., prior->J == NULL) leaves its n x n factor on the device; the slot's next window says
 * so with prior_n > 0 and prior_J == NULL (block ids, r and x0 as usual) and the solve copies it device to device -- 60 KB per window and frame that cross the
 * bus in neither direction.  gf_ba_fetch_resident_prior reads it back for whoever wants to look at it. */
int gf_ba_pack_slot(gf_ba* h, int slot, const gf_ba_window* window);
int gf_ba_fetch_resident_prior(gf_ba* h, int slot, int n, double* J);
int gf_ba_solve_packed(gf_ba* h, const int* slots, int n, int max_iters);
int gf_ba_unpack_slot(gf_ba* h, int slot, gf_ba_window* window, gf_ba_summary* summary);
/* ceres::Solver::Options::max_solver_time_in_seconds (estimator.cpp:3312-3315) for the following gf_ba_solve / gf_ba_solve_resident calls; 0 (default) = not
 * honoured: the iterations are enqueued back to back.  With a limit the host synchronises before every iteration, as Ceres checks its clock there. */
int gf_ba_set_max_solver_time(gf_ba* h, double seconds);
int gf_ba_get_stats(gf_ba* h, gf_ba_stats* out);
int gf_ba_debug_stamps(gf_ba* h, long long* out, int n); /* per-phase clock stamps of ba_step (profiling builds, -DGF_PROFILE_STEP) */
int gf_ba_reset_stats(gf_ba* h);
/* inspection for parity tests: H = J^T J, g = J^T r (loss-corrected, unscaled), cost; canonical column order
 * [free blocks: pose0, sb0, pose1, ..., ex, exw, sx, sy, sw, td, tdw | free features by index] */
int gf_ba_linearize(gf_ba* h, const gf_ba_window* w, int cap, double* H, double* g, double* cost, int* n_f, int* n_e, int* col_block_id);

/* Estimator::double2vector, pose part (estimator.cpp:2440-2497, USE_IMU branch; R2ypr/ypr2R in degrees, utility.h:78-118):
 * R0_before = Rs[0] (3x3 row-major) and P0_before = Ps[0] before the solve; outputs Rs (W+1)x9 row-major, Ps/Vs/Bas/Bgs (W+1)x3. */
int gf_ba_double2vector(int W, const double* R0_before, const double* P0_before, const double* para_Pose, const double* para_SpeedBias,
                        double* Rs, double* Ps, double* Vs, double* Bas, double* Bgs);

/* IntegrationBase::push_back loop (factor/integration_base.h:39-167), host side; noise = ACC_N, GYR_N, ACC_W, GYR_W */
int gf_imu_preintegrate(int n, const double* dt, const double* acc, const double* gyr, const double* acc0, const double* gyr0, const double* ba,
                        const double* bg, const double* noise, double* delta_p, double* delta_q, double* delta_v, double* jacobian,
                        double* covariance, double* sum_dt);
/* The 3-vector / quaternion part of the same loop alone (round 6): delta_p, delta_q (w x y z), delta_v, sum_dt -- bit for bit what gf_imu_preintegrate returns for them,
 * without the Jacobian and the covariance (99 % of the arithmetic).  Estimator::checkimu (estimator.cpp:2173-2216) reads delta_v / sum_dt of every frame on every image. */
int gf_imu_preintegrate_state(int n, const double* dt, const double* acc, const double* gyr, const double* acc0, const double* gyr0, const double* ba,
                              const double* bg, double* delta_p, double* delta_q, double* delta_v, double* sum_dt);
/* The same loop for many intervals at once ON THE DEVICE (row B2 batched): interval i owns the samples first[i] .. first[i+1]-1 of dt / acc / gyr
 * and row i of acc0 / gyr0 / ba / bg (n x 3) and of the outputs (delta_p n x 3, delta_q n x 4 as w x y z, delta_v n x 3, jacobian / covariance n x 225, sum_dt n).
 * One wavefront per interval; every result is bit-identical to gf_imu_preintegrate on the same samples.  No CPU fallback: GF_ERR_NO_DEVICE without a GPU. */
typedef struct gf_preint gf_preint;
int gf_preint_create(gf_preint** out);
int gf_preint_destroy(gf_preint* h);
int gf_imu_preintegrate_batch(gf_preint* h, int n, const int* first, const double* dt, const double* acc, const double* gyr, const double* acc0, const double* gyr0,
                              const double* ba, const double* bg, const double* noise, double* delta_p, double* delta_q, double* delta_v, double* jacobian,
                              double* covariance, double* sum_dt);
int gf_preint_stats(gf_preint* h, long long* launches, long long* intervals, double* kernel_ms); /* kernel_ms: hipEvent time of the kernel alone, summed */
/* The per-feature sweeps of the measurement side for many windows at once ON THE DEVICE, one thread per feature, decisions and depths
 * bit-identical to the host loops of the estimator:
 *   gf_triangulate_with_depth_batch  FeatureManager::triangulateWithDepth, feature_manager.cpp:726-799 (estimated_depth / estimate_flag updated in place)
 *   gf_moving_consistency_batch      Estimator::movingConsistencyCheckW, estimator.cpp:3955-3995 (remove[f] = 1 for the ids the reference puts into removeIndex)
 * Window b: Rs / Ps ((W+1) x 9 row-major / (W+1) x 3), tic (3), ric (9), features first_feature[b] .. first_feature[b+1]-1 (feature_manager's list order);
 * feature f: start_frame[f], observations first_obs[f] .. first_obs[f+1]-1, each x, y, z of the normalised point and the depth-camera depth. */
typedef struct gf_featsweep gf_featsweep;
int gf_featsweep_create(gf_featsweep** out);
int gf_featsweep_destroy(gf_featsweep* h);
int gf_triangulate_with_depth_batch(gf_featsweep* h, int B, int W, const double* Rs, const double* Ps, const double* tic, const double* ric, const int* first_feature,
                                    const int* start_frame, const int* first_obs, const double* obs, double depth_threshold, double init_depth,
                                    double* estimated_depth, int* estimate_flag);
int gf_moving_consistency_batch(gf_featsweep* h, int B, int W, const double* Rs, const double* Ps, const double* tic, const double* ric, const int* first_feature,
                                const int* start_frame, const int* first_obs, const double* obs, const double* estimated_depth, double focal_length, int* remove);
int gf_featsweep_stats(gf_featsweep* h, long long* launches, long long* features, double* kernel_ms);
/* WheelIntegrationBase::push_back loop (factor/wheel_integration_base.h:41-178); noise = VEL_N_wheel, GYR_N_wheel; lin = sx, sy, sw */
int gf_wheel_preintegrate(int n, const double* dt, const double* vel, const double* gyr, const double* vel0, const double* gyr0, const double* lin,
                          const double* noise, double* delta_p, double* delta_q, double* jacobian, double* covariance, double* sum_dt);

/* ------------------------------------------------------------------------------------------------------------------------------
 * Estimator: the call surface of Estimator::processImage and its bookkeeping, one handle per
 * sequence.  Replaces, in vins_estimator/src/estimator/estimator.h: inputIMU :104, inputWheel :106, inputFeature :108,
 * inputImage :105, processImage :110 (via processMeasurements :113), and the FeatureManager it owns (feature_manager.h:139-215).
 * The dense work (Estimator::optimization, estimator.cpp:2890-3636) runs on the HIP back end behind gf_ba_*.
 * Built: RGB-D + IMU (+ wheel) (+ GNSS) configuration, stationary / wheel-activated initialisation (estimator.cpp:1557-1682) and the SfM branch behind
 * them for recordings that begin in motion (estimator.cpp:1684-1926; relativePoseWithDepth, GlobalSFM::constructWithDepth, visualInitialAlign),
 * MULTIPLE_THREAD 0/1 data flow (processed synchronously); GNSS: measurement gating, clock / anchor / yaw states, factors in the solve and the
 * marginalisation, GNSSVIAlign with its initialiser, broadcast ephemerides -> satellite states (gf_estimator_input_ephem + gf_estimator_input_gnss_raw;
 * or states handed in, gf_gnss_obs).  Not built: monocular (DEPTH 0) initialisation, line / plane / motion factors.
 * ------------------------------------------------------------------------------------------------------------------------------ */
typedef struct gf_estimator gf_estimator;

typedef struct gf_estimator_cfg {
    int window_size;        /* WINDOW_SIZE, parameters.h:24 */
    int max_features;       /* NUM_OF_F, parameters.h:25 */
    int max_visual;         /* capacity of visual factors per window */
    int use_imu, use_wheel, depth;                              /* imu / wheel / depth, parameters.cpp:160-176 */
    int estimate_extrinsic, estimate_wheel_extrinsic, estimate_wheel_intrinsic, estimate_td, estimate_td_wheel;
    int use_mcc, wdetect, stationary_detect, only_initial_with_wheel, multiple_thread;
    int num_iterations;     /* max_num_iterations */
    int with_tracker;       /* own a FeatureTracker (gf_estimator_input_image); `tracker` below configures it */
    double acc_n, gyr_n, acc_w, gyr_w, g_norm, wheel_vel_n, wheel_gyr_n;
    double min_parallax_px; /* keyframe_parallax (pixels at FOCAL_LENGTH) */
    double depth_threshold, init_depth, focal_length; /* depth_threshold; INIT_DEPTH parameters.cpp:478; FOCAL_LENGTH parameters.h:23 */
    double td, td_wheel, sx, sy, sw;
    double tic[3], ric[9], tio[3], rio[9];  /* body_T_cam0, body_T_wheel (row-major rotations) */
    gf_tracker_cfg tracker;
    /* GNSS (parameters.cpp:519-552): gnss_enable; thresholds of processGNSS (estimator.cpp:1497-1523); GNSS_DDT_WEIGHT = 1 / gnss_ddt_sigma;
     * max_gnss_per_frame: capacity of gnss_meas_buf[i]; gnss_iono[8]: gnss_iono_default_parameters */
    int gnss_enable, gnss_track_num_thres, max_gnss_per_frame;
    double gnss_elevation_thres, gnss_psr_std_thres, gnss_dopp_std_thres, gnss_ddt_sigma, gnss_local_time_diff;
    double gnss_iono[8];
    /* SOLVER_TIME (`max_solver_time`, parameters.cpp:343): the solve gets 4/5 of it when the oldest frame will be marginalised, all of it otherwise
     * (estimator.cpp:3312-3315).  0 = not honoured (parity runs: the oracle counts iterations only).  A wall-clock cut makes the members of one
     * batch end after different iteration counts depending on who shares the batch, so gf_estimator_group_create refuses a value > 0. */
    double max_solver_time;
    /* extrinsic_type / extrinsic_type_wheel (parameters.cpp:394-420, :280-306): which components of the camera / wheel extrinsic the solver may move
     * once the block is free (estimate_extrinsic / estimate_wheel_extrinsic): 0 all, 1 translation, 2 rotation, 3 no z, 4 no rotation and no z */
    int extrinsic_type, extrinsic_type_wheel;
} gf_estimator_cfg;

/* One L1 observation of a GNSS epoch as Estimator::inputGNSS receives it (ObsPtr), together with what GnssPsrDoppFactor's constructor derives from
 * the matching ephemeris (gnss_psr_dopp_factor.cpp:3-47 via gnss_comm eph2pos / geph2pos / eph2svdt: not part of this build):
 * the satellite state at transmission time. */
typedef struct gf_gnss_obs {
    int sat;                 /* satellite number (tracking statistics, estimator.cpp:1497-1511) */
    int sys;                 /* constellation index 0 GPS, 1 GLO, 2 GAL, 3 BDS (gnss_comm::sys2idx); < 0: any other system, dropped (estimator.cpp:1463-1465) */
    double time;             /* time2sec(obs->time) [s] */
    double psr, dopp, psr_std, dopp_std, wavelength;   /* L1 pseudorange [m], Doppler [Hz], their standard deviations, carrier wavelength [m] */
    double sv_pos[3], sv_vel[3], svdt, svddt, tgd, pr_uura, dp_uura, tow;
} gf_gnss_obs;

int gf_estimator_default_cfg(gf_estimator_cfg* cfg);   /* values of config/realsense/m2dgrp.yaml */
int gf_estimator_create(const gf_estimator_cfg* cfg, gf_estimator** out);
int gf_estimator_destroy(gf_estimator* h);
int gf_estimator_input_imu(gf_estimator* h, double t, const double* acc, const double* gyr);
int gf_estimator_input_wheel(gf_estimator* h, double t, const double* vel, const double* gyr);
/* Broadcast ephemerides as Estimator::inputEphem receives them (estimator.h:97, estimator.cpp:1428-1437; gnss_comm::Ephem / GloEphem) and the L1 entry of an
 * observation (gnss_comm::Obs).  gtime_t fields are seconds (time2sec).  With these the library derives the satellite state itself, the way
 * GnssPsrDoppFactor's constructor does (gnss_psr_dopp_factor.cpp:3-47: transmission time, eph2svdt / eph2pos / eph2vel or their GLONASS counterparts);
 * gnss_comm is not vendored by the reference, so eph2pos etc. are restated from the published broadcast-orbit algorithms (IS-GPS-200 Kepler model, BeiDou
 * GEO rotation, GLONASS ICD Runge-Kutta with 60 s steps; velocities and clock drifts by the 1 ms difference quotient RTKLIB uses). */
typedef struct gf_gnss_ephem {
    int sat, sys, prn;       /* satellite number, constellation index (0 GPS, 2 GAL, 3 BDS), PRN within it (BeiDou GEO: prn <= 5 or >= 59) */
    double toe, toc, toe_tow; /* time2sec(toe), time2sec(toc), toe as seconds of the constellation's week */
    double A, e, i0, OMG0, omg, M0, delta_n, OMG_dot, i_dot, cuc, cus, crc, crs, cic, cis, af0, af1, af2, tgd0, ura;
} gf_gnss_ephem;
typedef struct gf_gnss_glo_ephem {
    int sat;
    double toe;              /* time2sec(toe) */
    double pos[3], vel[3], acc[3], tau_n, gamma;   /* PZ-90 state at toe [m, m/s, m/s^2], clock bias [s] and relative frequency bias */
} gf_gnss_glo_ephem;
typedef struct gf_gnss_raw_obs {
    int sat, sys;            /* as in gf_gnss_obs; sys 1 = GLONASS */
    double time, psr, dopp, psr_std, dopp_std, freq, tow;   /* reception time [s], L1 pseudorange [m], Doppler [Hz], their deviations, carrier frequency [Hz], time of week */
} gf_gnss_raw_obs;
/* the satellite state of one observation from its ephemeris (exactly one of eph / geph non-NULL): what gf_estimator_input_gnss_raw does per observation */
int gf_gnss_obs_from_ephem(const gf_gnss_raw_obs* raw, const gf_gnss_ephem* eph, const gf_gnss_glo_ephem* geph, gf_gnss_obs* out);
/* satellite position [m] and clock bias [s] at time t (eph2pos / geph2pos): building block entry for tests */
int gf_gnss_eph2pos(double t, const gf_gnss_ephem* eph, const gf_gnss_glo_ephem* geph, double* pos3, double* svdt);

/* Estimator::inputGNSS (estimator.h:99, estimator.cpp:397): one epoch; inputGNSSTimeDiff (estimator.cpp:1450); inputIonoParams (estimator.h:98) */
int gf_estimator_input_gnss(gf_estimator* h, double t, const gf_gnss_obs* obs, int n);
/* the same with raw observations: processGNSS then picks, per observation, the ephemeris of its satellite nearest in toe within EPH_VALID_SECONDS (7200 s)
 * (estimator.cpp:1467-1495; observations without one are skipped), evaluates the elevation gate at the reception time and derives the satellite state */
int gf_estimator_input_gnss_raw(gf_estimator* h, double t, const gf_gnss_raw_obs* obs, int n);
int gf_estimator_input_ephem(gf_estimator* h, const gf_gnss_ephem* eph);            /* Estimator::inputEphem, estimator.cpp:1428-1437 */
int gf_estimator_input_glo_ephem(gf_estimator* h, const gf_gnss_glo_ephem* geph);
int gf_estimator_input_gnss_time_diff(gf_estimator* h, double diff_t_gnss_local);
int gf_estimator_input_iono_params(gf_estimator* h, const double* params8);
/* GNSSVIAlign (estimator.cpp:1928-2043) runs inside the library: GNSSVIInitializer (initial/gnss_vi_initializer.cpp: coarse SPP localisation of the
 * window's measurements, yaw alignment on the Doppler residuals, anchor refinement) on the satellite states of gf_gnss_obs, under the reference's
 * preconditions (visual-inertial part initialised, mean horizontal speed of the window >= 0.3 m/s).  A caller that has its own fix may hand it in
 * instead: the next alignment attempt then takes refined_xyzt = anc_ecef + rcv_dt[4], yaw and clock drift from here (rcv_dt[k] = 0: system k unobserved). */
int gf_estimator_set_gnss_alignment(gf_estimator* h, const double* anc_ecef, double yaw_enu_local, const double* rcv_dt4, double rcv_ddt);
/* gnss[8]: gnss_ready, lowspeed, #valid measurements of the newest frame, first_optimization, then 4 reserved; any pointer may be NULL.
 * rcv_dt 4 (W+1), rcv_ddt (W+1), anc_ecef 3, ecef_pos 3, enu_pos 3 (updateGNSSStatistics, estimator.cpp:2045-2058) */
int gf_estimator_get_gnss_state(gf_estimator* h, int* gnss, double* rcv_dt, double* rcv_ddt, double* yaw_enu_local, double* anc_ecef, double* ecef_pos, double* enu_pos);
/* inputFeature + processMeasurements: one call = one processImage once IMU / wheel data cover the frame time */
int gf_estimator_input_feature(gf_estimator* h, double t, const gf_feature_obs* obs, int n);
/* Estimator::processImage(image, header) (estimator.h:110, estimator.cpp:843-1163) called directly, as the reference's public method allows: the frame
 * goes through the keyframe vote, initialisation / optimisation and the window shift with whatever processIMU / processWheel have integrated so far
 * (inputFeature does this for the frames it takes off the queue).  Needs at least one processed IMU sample when use_imu is set. */
int gf_estimator_process_image(gf_estimator* h, double header, const gf_feature_obs* obs, int n);
/* inputImage: trackImage on the owned tracker, then inputFeature (every second frame when multiple_thread, estimator.cpp:226) */
int gf_estimator_input_image(gf_estimator* h, double t, const uint8_t* gray, int stride, const uint16_t* depth, int dstride,
                             gf_feature_obs* out, int cap, int* n_out);
/* window state; arrays of (window_size+1) entries, any pointer may be NULL.  info[16]: frame_count, solver_flag, marginalization_flag,
 * #features, prior valid, prior n, systemstationary, last iterations, last successful steps, #optimizations, openExWheelEstimation,
 * last_track_num, long_track_num, new_feature_num, sum_of_back, sum_of_front.  extr[32]: tic 3, ric 9, tio 3, rio 9, sx, sy, sw, td,
 * td_wheel, last initial cost, last final cost, last_average_parallax */
int gf_estimator_get_state(gf_estimator* h, double* Ps, double* Rs, double* Vs, double* Bas, double* Bgs, double* Headers, int* info, double* extr);
int gf_estimator_set_state(gf_estimator* h, int frame_count, int solver_flag, const double* Ps, const double* Rs, const double* Vs,
                           const double* Bas, const double* Bgs);
int gf_estimator_get_features(gf_estimator* h, int cap, int* id, int* start_frame, int* n_obs, double* estimated_depth, int* estimate_flag,
                              int* solve_flag, int* n);
/* latest_time, latest_P, latest_Q, latest_V (estimator.h:354-356) and their wheel counterparts (estimator.h:239-242): the newest window state propagated
 * through every IMU / wheel sample received since (fastPredictIMU estimator.cpp:4014-4028 inside inputIMU :332, fastPredictWheel :4079-4093 inside inputWheel :363,
 * re-anchored by updateLatestStates :4141-4198 after every optimised frame) -- what pubLatestOdometry / pubWheelLatestOdometry publish at sensor rate.
 * imu / wheel: 16 doubles each = time, P[3], rotation matrix [9] (row-major), V[3]; either may be NULL.  Before the first optimised frame the reference
 * publishes uninitialised members; here they start at zero / identity. */
int gf_estimator_get_latest(gf_estimator* h, double* imu, double* wheel);
/* predictPtsInNextFrame / removeOutliers feedback of the last processImage (estimator.cpp:1132-1136) */
int gf_estimator_get_feedback(gf_estimator* h, int cap, int* predict_ids, double* predict_xyz, int* n_predict, int* remove_ids, int* n_remove);
int gf_estimator_get_prior(gf_estimator* h, int cap_n, int cap_blocks, int* n, int* nblocks, int* block_id, double* J, double* r);
/* single host-side steps by name (FeatureManager members, slideWindow, ...) for unit tests; see gf_estimator.hip */
int gf_estimator_debug(gf_estimator* h, const char* op, const double* in, int n_in, double* out, int cap_out, int* n_out);

/* ---- many sequences on one batched solver (not in the reference; the batched counterpart of gf_estimator_*) -------------------------------
 * n Estimators that keep the reference's single-sequence control flow and share one gf_ba handle of batch n: whenever the members that
 * are busy with a frame have all reached ceres::Solve or the marginalisation, their windows go to the device in one call.  Members take
 * feature frames (run one batched gf_tracker next to the group); IMU / wheel samples and state queries go through the member handles. */
typedef struct gf_estimator_group gf_estimator_group;
int gf_estimator_group_create(const gf_estimator_cfg* cfg, int n, gf_estimator_group** out);
int gf_estimator_group_destroy(gf_estimator_group* g);
/* the IMU intervals a camera frame completes (the frame's own IntegrationBase and the window's, estimator.cpp:760-768, :866-869) are integrated
 * by one device launch per group step (gf_imu_preintegrate_batch's kernel) instead of on the members' host threads; results are bit-identical either way.
 * Off by default (environment GF_GROUP_DEVICE_PREINT=1 turns it on at creation): it pays on hosts with few cores, DESIGN.md section 8. */
int gf_estimator_group_set_device_preint(gf_estimator_group* g, int on);
/* FeatureManager::triangulateWithDepth (feature_manager.cpp:726-799) and Estimator::movingConsistencyCheckW (estimator.cpp:3955-3995) of all members
 * as one launch each per group step (the kernels of gf_triangulate_with_depth_batch / gf_moving_consistency_batch) instead of the members' host loops; depths, flags
 * and removed ids are bit-identical either way.  Off by default (environment GF_GROUP_DEVICE_SWEEPS=1 turns it on at creation): each launch is one more rendezvous
 * of the members, and the host loops cost ~20 us per window. */
int gf_estimator_group_set_device_sweeps(gf_estimator_group* g, int on);
int gf_estimator_group_member(gf_estimator_group* g, int i, gf_estimator** out);   /* owned by the group */
/* Estimator::inputFeature on each listed sequence, concurrently; obs = the frames back to back, n_obs[k] entries for seq[k] */
int gf_estimator_group_input_features(gf_estimator_group* g, int count, const int* seq, const double* t, const gf_feature_obs* obs, const int* n_obs);
/* The same step in two halves, as in the reference: Estimator::inputFeature only queues the frame (estimator.cpp:447-459) and processMeasurements works on it in
 * its own thread (estimator.cpp:470-560) while the tracker already handles the next image.  submit publishes the frames to the group's workers and returns;
 * wait blocks until every listed member has finished its frame and reports the first member's error.  stride >= 0: the observations of seq[k] start at
 * obs + k * stride (a tracker's padded output table as it lies: gf_tracker_track_batch*'s `out` with stride = its capacity); < 0: back to back.
 * obs / seq / n_obs stay the caller's until wait returns; one step in flight per group. */
int gf_estimator_group_submit_features(gf_estimator_group* g, int count, const int* seq, const double* t, const gf_feature_obs* obs, const int* n_obs, long long stride);
int gf_estimator_group_wait(gf_estimator_group* g);
int gf_estimator_group_stats(gf_estimator_group* g, long long* batches, long long* windows, long long* largest_batch);

/* ---- ROS-free I/O around the path: config files, trajectory output, raw frames ---------------------------------------- */
/* readParameters(std::string config_file), vins_estimator/src/estimator/parameters.cpp:138-558, plus the cam0_calib file it names
 * (PinholeCamera::Parameters::readFromYamlFile, camera_models/src/camera_models/PinholeCamera.cc:145-183; path relative to the config
 * file's directory, parameters.cpp:436-443).  Same key names and cv::FileNode defaults (a missing numeric key reads as 0).  Fills the
 * whole cfg including cfg->tracker and sets with_tracker = 1 (gnss_enable and its gnss_* keys are read, parameters.cpp:519-552).  Options
 * outside the built path (use_line, use_yolo, plane, equalize, use_motion, num_of_cam 2, estimate_extrinsic 2,
 * gnss_local_online_sync) return GF_ERR_INVALID instead of being ignored. */
int gf_estimator_cfg_from_yaml(const char* config_file, gf_estimator_cfg* cfg);
/* the line pubOdometry appends to VINS_RESULT_PATH (utility/visualization.cpp:346-357): "t x y z qx qy qz qw", fixed, 9 decimals;
 * P = Ps[WINDOW_SIZE], R = Rs[WINDOW_SIZE] (row-major), quaternion as Eigen::Quaterniond(R) */
int gf_tum_append(const char* path, double t, const double* P, const double* R);
/* VINS_RESULT_PATH (output_path + "/vio.txt", parameters.cpp:347-352): creates the file empty; from then on every processed frame appends
 * its line as pubOdometry does (estimator.cpp:679 -> visualization.cpp:287-357: only once solver_flag == NON_LINEAR or the IMU was found
 * excited).  NULL or "" switches the output off. */
int gf_estimator_set_result_path(gf_estimator* h, const char* vio_txt);
/* binary PGM (P5) reader standing in for sensor_msgs::Image + cv_bridge (rosNodeTest.cpp:229-287): maxval <= 255 -> u8 (MONO8),
 * else u16 in host byte order (MONO16, depth in mm).  pixels == NULL only queries the header. */
int gf_pgm_read(const char* path, int* width, int* height, int* maxval, void* pixels, size_t cap_bytes);

/* ---- multi-GPU exchange without torch: one process per GPU, sequences sharded over the ranks, ONE collective -- the all-gather of the newest pose
 * of every window (7 doubles each) over RCCL / xGMI.  RCCL is resolved at run time (symbols already in the process, else librccl.so.1): the library itself
 * does not link against it.  The reference has no counterpart (one Estimator per process, rosNodeTest.cpp:713); this is north_star's partitioning. */
typedef struct gf_comm gf_comm;
/* rank 0 creates the 128-byte id (ncclGetUniqueId) and hands it to the other ranks out of band; every rank then builds its communicator on its device */
int gf_comm_unique_id(unsigned char* id128);
int gf_comm_create(const unsigned char* id128, int world, int rank, int device, gf_comm** out);
int gf_comm_destroy(gf_comm* c);
/* world, rank, device, the ncclComm_t and the communicator's own hipStream_t (any of them may be NULL) */
int gf_comm_info(gf_comm* c, int* world, int* rank, int* device, void** nccl_comm, void** stream);
/* all-gather of n doubles per rank between host buffers (staged through the device): recv_host = [world][n] */
int gf_comm_allgather_f64(gf_comm* c, const double* send_host, int n, double* recv_host);
/* The exchange step of the path: Ps / Rs[WINDOW_SIZE] (px py pz qx qy qz qw) of this rank's first `count` resident windows -> d_out = [world][count][7] on every
 * rank (device memory), ncclAllGather on `nccl_comm` (ncclComm_t of any owner: gf_comm_info, or torch's) enqueued on `stream` (hipStream_t) behind the solver's
 * stream.  Every rank passes the same count (the largest shard); rows behind a rank's own resident windows are zero.  Asynchronous: synchronise `stream`. */
int gf_pose_gather(gf_ba* h, void* nccl_comm, void* stream, int count, double* d_out);
/* NUMA placement: node of the GPU (/sys/bus/pci/devices/<bus id>/numa_node; -1 = none reported) and its cpulist; gf_pin_thread_to_device_node moves the calling
 * thread onto those cores (intersection with the process's affinity) and returns the node or -1; it acts when several ranks share the host (LOCAL_WORLD_SIZE > 1)
 * or GF_NUMA_PIN=1 says so, GF_NUMA_PIN=0 switches it off.  The tracker's bookkeeping pool
 * and the estimator group's workers pin themselves this way: 8 ranks x (pool + workers) on one host stay next to their own GPU. */
int gf_numa_node_of_device(int device, int* node, char* cpulist, int cap);
int gf_pin_thread_to_device_node(int device);

/* ---- ROS bag files (format 2.0) without ROS: `rosbag play <bag>` into the node's subscribers (README.md:146-187 is the only way the reference is run;
 * rosNodeTest.cpp:678-682 subscribes IMU_TOPIC / WHEEL_TOPIC / IMAGE0_TOPIC / IMAGE1_TOPIC).  Host code (ground-fusion_amd/host/rosbag_reader.h): bag records,
 * chunks (none / bz2 via libbz2.so / lz4 decoded here), index data or a scan of un-indexed chunks, connection records.  gnss_comm messages are not decoded. */
typedef struct gf_bag gf_bag;
int gf_bag_open(const char* path, gf_bag** out);
int gf_bag_close(gf_bag* b);
int gf_bag_connection_count(gf_bag* b);
/* topic / datatype of connection record i (0 .. count - 1) and its id; strings are truncated to the caller's capacity */
int gf_bag_connection(gf_bag* b, int i, int* conn_id, char* topic, int topic_cap, char* type, int type_cap);
/* the messages of the listed topics (n_topics == 0: all) in play order -- record time, ties in file order; returns their number in *count */
int gf_bag_select(gf_bag* b, const char* const* topics, int n_topics, long long* count);
/* message i of the selection: connection id, record time [s], serialized payload (valid until the next gf_bag_message / gf_bag_close on this handle) */
int gf_bag_message(gf_bag* b, long long i, int* conn_id, double* t_record, const unsigned char** data, size_t* len);
/* sensor_msgs/Imu as imu_callback reads it (rosNodeTest.cpp:567-585): header.stamp.toSec(), linear_acceleration, angular_velocity */
int gf_ros_decode_imu(const unsigned char* data, size_t len, double* t, double* acc, double* gyr);
/* nav_msgs/Odometry as wheel_callback reads it (rosNodeTest.cpp:81-189): header.stamp.toSec(), twist.twist.linear / angular, pose.pose.position (may be NULL) */
int gf_ros_decode_odometry(const unsigned char* data, size_t len, double* t, double* linear, double* angular, double* position);
/* sensor_msgs/Image -> what the node hands to trackImage.  depth == 0: getImageFromMsg (rosNodeTest.cpp:238-263): 8UC1 / mono8 copied, rgb8 / bgr8 / rgba8 /
 * bgra8 through cv_bridge::toCvCopy(MONO8) = OpenCV 4.2 cvtColor (R 4899 + G 9617 + B 1868 + 2^13) >> 14; one byte per pixel.  depth != 0:
 * getDepthImageFromMsg (:265-286): the payload relabelled MONO16; two bytes per pixel in host order.  pixels == NULL only queries t / width / height. */
int gf_ros_decode_image(const unsigned char* data, size_t len, int depth, double* t, int* width, int* height, void* pixels, size_t cap_bytes);

#ifdef __cplusplus
}
#endif
#endif
