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
