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

/* Same for all `batch` sequences in one pass; gray[b]/depth[b] are host images, out is [batch][cap]. */
int gf_tracker_track_batch(gf_tracker* h, const double* t, const uint8_t* const* gray, int stride,
                           const uint16_t* const* depth, int dstride, gf_feature_obs* out, int cap, int* n_out);

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

#ifdef __cplusplus
}
#endif
#endif
