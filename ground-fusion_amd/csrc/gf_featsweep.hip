// gf_featsweep.hip — the per-feature sweeps of the measurement side for many windows at once on the device:
//   FeatureManager::triangulateWithDepth   feature_manager.cpp:726-799   (depth of a track from its depth-camera observations, cross-checked between frames)
//   Estimator::movingConsistencyCheckW     estimator.cpp:3955-3995 with reprojectionError / reprojectionError3D :3899-3919
// One thread per feature, the loops of the host code (gf_estimator.hip) in the same order without contraction: decisions and depths are bit-identical to the host's.
// Building blocks with their own C-ABI; the estimator keeps these sweeps on its host threads (0.03 ms of one core per frame, and a device launch
// would need one more rendezvous of the group's threads).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#include <hip/hip_runtime.h>
#include "../../include/groundfusion_hip.h"
#include "gf_dmath.hpp"

namespace gf { int set_err(int code, const char* fmt, ...); }
using namespace gfd;

namespace {
struct SweepArgs {
    int B, W, F;
    const double *Rs, *Ps, *tic, *ric;          // per window: (W+1) x 9, (W+1) x 3, 3, 9
    const int *first_feature, *start_frame, *first_obs;
    const double* obs;                          // per observation: x, y, z of the normalised point, depth-camera depth
    double* estimated_depth; int* estimate_flag; int* remove;
    double depth_threshold, init_depth, focal_length;
};
__device__ __forceinline__ V3 ld3(const double* p) { return v3(p[0], p[1], p[2]); }
__device__ __forceinline__ M3 ld9(const double* p) { M3 m; for (int i = 0; i < 9; i++) m.m[i] = p[i]; return m; }
__device__ int window_of(const int* first_feature, int B, int f) {   // largest b with first_feature[b] <= f
    int lo = 0, hi = B - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (first_feature[mid] <= f) lo = mid; else hi = mid - 1; }
    return lo;
}

__global__ void __launch_bounds__(128) triangulate_with_depth_kernel(SweepArgs A) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= A.F) return;
    const int b = window_of(A.first_feature, A.B, f);
    const int o0 = A.first_obs[f], n = A.first_obs[f + 1] - o0, s = A.start_frame[f];
    if (n < 4) return;
    if (A.estimated_depth[f] > 0) return;
    const double* Rsb = A.Rs + (size_t)b * (A.W + 1) * 9; const double* Psb = A.Ps + (size_t)b * (A.W + 1) * 3;
    const V3 tic = ld3(A.tic + 3 * b); const M3 ric = ld9(A.ric + 9 * b);
    double depth_sum = 0.0; unsigned cnt = 0;
    const M3 Rs_s = ld9(Rsb + 9 * s);
    const V3 tr = ld3(Psb + 3 * s) + Rs_s * tic; const M3 Rr = Rs_s * ric;
    for (int i = 0; i < n; i++) {
        const M3 Rsi = ld9(Rsb + 9 * (s + i));
        const V3 t0 = ld3(Psb + 3 * (s + i)) + Rsi * tic; const M3 R0 = Rsi * ric;
        const double* oi = A.obs + 4 * (size_t)(o0 + i);
        const double d = oi[3];
        if (d < 0.1 || d > A.depth_threshold) continue;
        const V3 point0 = ld3(oi) * d;
        const V3 t2r = transpose(Rr) * (t0 - tr); const M3 R2r = transpose(Rr) * R0;
        for (int j = 0; j < n; j++) {
            if (i == j) continue;
            const M3 Rsj = ld9(Rsb + 9 * (s + j));
            const V3 t1 = ld3(Psb + 3 * (s + j)) + Rsj * tic; const M3 R1 = Rsj * ric;
            const V3 t20 = transpose(R0) * (t1 - t0); const M3 R20 = transpose(R0) * R1;
            const V3 pp = transpose(R20) * point0 - transpose(R20) * t20;
            const double* oj = A.obs + 4 * (size_t)(o0 + j);
            const double rx = oj[0] - pp.x / pp.z, ry = oj[1] - pp.y / pp.z;
            if (sqrt(rx * rx + ry * ry) < 10.0 / 460) { const V3 pr = R2r * point0 + t2r; depth_sum += pr.z; cnt++; }
        }
    }
    if (cnt == 0) return;
    double e = depth_sum / cnt;
    int flag = 1;
    if (e < 0.1) { e = A.init_depth; flag = 0; }
    A.estimated_depth[f] = e; A.estimate_flag[f] = flag;
}

__global__ void __launch_bounds__(128) moving_consistency_kernel(SweepArgs A) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= A.F) return;
    A.remove[f] = 0;
    const int b = window_of(A.first_feature, A.B, f);
    const int o0 = A.first_obs[f], n = A.first_obs[f + 1] - o0, wi = A.start_frame[f];
    if (!(n >= 2 && wi < A.W - 2)) return;
    const double depth = A.estimated_depth[f];
    if (depth < 0) return;
    const double* Rsb = A.Rs + (size_t)b * (A.W + 1) * 9; const double* Psb = A.Ps + (size_t)b * (A.W + 1) * 3;
    const V3 tic = ld3(A.tic + 3 * b); const M3 ric = ld9(A.ric + 9 * b);
    const M3 Ri = ld9(Rsb + 9 * wi); const V3 Pi = ld3(Psb + 3 * wi);
    const V3 uvi = ld3(A.obs + 4 * (size_t)o0);
    double err = 0, err3D = 0; int errCnt = 0;
    for (int k = 1; k < n; k++) {
        const int wj = wi + k;
        const M3 Rj = ld9(Rsb + 9 * wj); const V3 Pj = ld3(Psb + 3 * wj);
        const V3 uvj = ld3(A.obs + 4 * (size_t)(o0 + k));
        {   // reprojectionError
            const V3 pts_w = Ri * (ric * (uvi * depth) + tic) + Pi;
            const V3 pts_cj = transpose(ric) * (transpose(Rj) * (pts_w - Pj) - tic);
            const double rx = pts_cj.x / pts_cj.z - uvj.x, ry = pts_cj.y / pts_cj.z - uvj.y;
            err += sqrt(rx * rx + ry * ry);
        }
        {   // reprojectionError3D
            const V3 pts_w = Ri * (ric * (uvi * depth) + tic) + Pi;
            const V3 pts_cj = transpose(ric) * (transpose(Rj) * (pts_w - Pj) - tic);
            err3D += sqrt(sqn(pts_cj - uvj)) / depth;
        }
        errCnt++;
    }
    if (errCnt > 0 && (A.focal_length * err / errCnt > 10 || err3D / errCnt > 2.0)) A.remove[f] = 1;
}

template <class T> struct DevArr {
    T* d = nullptr; size_t cap = 0;
    int fit(size_t n) {
        if (n <= cap) return GF_OK;
        if (d) (void)hipFree(d);
        d = nullptr; cap = 0;
        const size_t c = std::max(n, (size_t)64);
        if (hipMalloc(&d, c * sizeof(T)) != hipSuccess) return gf::set_err(GF_ERR_HIP, "hipMalloc of %zu bytes failed", c * sizeof(T));
        cap = c;
        return GF_OK;
    }
    ~DevArr() { if (d) (void)hipFree(d); }
};
}  // namespace

struct gf_featsweep {
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    DevArr<double> Rs, Ps, tic, ric, obs, depth;
    DevArr<int> first_feature, start_frame, first_obs, flag, remove;
    double kernel_ms = 0; long long launches = 0, features = 0;
};

#define FS_HIP(x) do { const hipError_t e_ = (x); if (e_ != hipSuccess) return gf::set_err(GF_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_)); } while (0)

static int upload_common(gf_featsweep* h, SweepArgs& A, int B, int W, const double* Rs, const double* Ps, const double* tic, const double* ric, const int* first_feature,
                         const int* start_frame, const int* first_obs, const double* obs, const double* estimated_depth) {
    const int F = first_feature[B];
    if (F < 0 || (F > 0 && (!start_frame || !first_obs || !obs))) return gf::set_err(GF_ERR_INVALID, "%d features listed but start_frame / first_obs / obs is null", F);
    const size_t O = F > 0 ? (size_t)first_obs[F] : 0;
    for (int b = 0; b < B; b++) if (first_feature[b + 1] < first_feature[b]) return gf::set_err(GF_ERR_INVALID, "first_feature must not decrease");
    for (int f = 0; f < F; f++) {
        const int n = first_obs[f + 1] - first_obs[f];
        if (n < 0 || start_frame[f] < 0 || start_frame[f] + n > W + 1) return gf::set_err(GF_ERR_INVALID, "feature %d: %d observations from frame %d do not fit a window of %d frames", f, n, start_frame[f], W + 1);
    }
    if (int rc = h->Rs.fit((size_t)B * (W + 1) * 9)) return rc;
    if (int rc = h->Ps.fit((size_t)B * (W + 1) * 3)) return rc;
    if (int rc = h->tic.fit((size_t)B * 3)) return rc;
    if (int rc = h->ric.fit((size_t)B * 9)) return rc;
    if (int rc = h->obs.fit(O * 4)) return rc;
    if (int rc = h->depth.fit(F)) return rc;
    if (int rc = h->first_feature.fit(B + 1)) return rc;
    if (int rc = h->start_frame.fit(F)) return rc;
    if (int rc = h->first_obs.fit(F + 1)) return rc;
    if (int rc = h->flag.fit(F)) return rc;
    if (int rc = h->remove.fit(F)) return rc;
    hipStream_t s = h->stream;
    FS_HIP(hipMemcpyAsync(h->Rs.d, Rs, sizeof(double) * B * (W + 1) * 9, hipMemcpyHostToDevice, s));
    FS_HIP(hipMemcpyAsync(h->Ps.d, Ps, sizeof(double) * B * (W + 1) * 3, hipMemcpyHostToDevice, s));
    FS_HIP(hipMemcpyAsync(h->tic.d, tic, sizeof(double) * B * 3, hipMemcpyHostToDevice, s));
    FS_HIP(hipMemcpyAsync(h->ric.d, ric, sizeof(double) * B * 9, hipMemcpyHostToDevice, s));
    FS_HIP(hipMemcpyAsync(h->first_feature.d, first_feature, sizeof(int) * (B + 1), hipMemcpyHostToDevice, s));
    if (F > 0) {
        FS_HIP(hipMemcpyAsync(h->obs.d, obs, sizeof(double) * O * 4, hipMemcpyHostToDevice, s));
        FS_HIP(hipMemcpyAsync(h->depth.d, estimated_depth, sizeof(double) * F, hipMemcpyHostToDevice, s));
        FS_HIP(hipMemcpyAsync(h->start_frame.d, start_frame, sizeof(int) * F, hipMemcpyHostToDevice, s));
        FS_HIP(hipMemcpyAsync(h->first_obs.d, first_obs, sizeof(int) * (F + 1), hipMemcpyHostToDevice, s));
    }
    A.B = B; A.W = W; A.F = F;
    A.Rs = h->Rs.d; A.Ps = h->Ps.d; A.tic = h->tic.d; A.ric = h->ric.d; A.first_feature = h->first_feature.d; A.start_frame = h->start_frame.d; A.first_obs = h->first_obs.d;
    A.obs = h->obs.d; A.estimated_depth = h->depth.d; A.estimate_flag = h->flag.d; A.remove = h->remove.d;
    return GF_OK;
}

extern "C" {
int gf_featsweep_create(gf_featsweep** out) {
    if (!out) return gf::set_err(GF_ERR_INVALID, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return gf::set_err(GF_ERR_NO_DEVICE, "no HIP device: the batched feature sweeps have no CPU fallback");
    gf_featsweep* h = new gf_featsweep();
    if (hipStreamCreate(&h->stream) != hipSuccess || hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) { delete h; return gf::set_err(GF_ERR_HIP, "stream / event creation failed"); }
    *out = h;
    return GF_OK;
}
int gf_featsweep_destroy(gf_featsweep* h) {
    if (!h) return GF_OK;
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return GF_OK;
}
int gf_triangulate_with_depth_batch(gf_featsweep* h, int B, int W, const double* Rs, const double* Ps, const double* tic, const double* ric, const int* first_feature,
                                    const int* start_frame, const int* first_obs, const double* obs, double depth_threshold, double init_depth, double* estimated_depth,
                                    int* estimate_flag) {
    if (!h || B < 1 || W < 1 || !Rs || !Ps || !tic || !ric || !first_feature || !first_obs || !estimated_depth || !estimate_flag) return gf::set_err(GF_ERR_INVALID, "bad argument");
    SweepArgs A{};
    if (int rc = upload_common(h, A, B, W, Rs, Ps, tic, ric, first_feature, start_frame, first_obs, obs, estimated_depth)) return rc;
    if (A.F == 0) return GF_OK;
    A.depth_threshold = depth_threshold; A.init_depth = init_depth;
    FS_HIP(hipMemcpyAsync(h->flag.d, estimate_flag, sizeof(int) * A.F, hipMemcpyHostToDevice, h->stream));
    FS_HIP(hipEventRecord(h->ev0, h->stream));
    triangulate_with_depth_kernel<<<dim3((A.F + 127) / 128), 128, 0, h->stream>>>(A);
    FS_HIP(hipGetLastError());
    FS_HIP(hipEventRecord(h->ev1, h->stream));
    FS_HIP(hipMemcpyAsync(estimated_depth, h->depth.d, sizeof(double) * A.F, hipMemcpyDeviceToHost, h->stream));
    FS_HIP(hipMemcpyAsync(estimate_flag, h->flag.d, sizeof(int) * A.F, hipMemcpyDeviceToHost, h->stream));
    FS_HIP(hipStreamSynchronize(h->stream));
    float ms = 0;
    if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) h->kernel_ms += ms;
    h->launches++; h->features += A.F;
    return GF_OK;
}
int gf_moving_consistency_batch(gf_featsweep* h, int B, int W, const double* Rs, const double* Ps, const double* tic, const double* ric, const int* first_feature,
                                const int* start_frame, const int* first_obs, const double* obs, const double* estimated_depth, double focal_length, int* remove) {
    if (!h || B < 1 || W < 1 || !Rs || !Ps || !tic || !ric || !first_feature || !first_obs || !estimated_depth || !remove) return gf::set_err(GF_ERR_INVALID, "bad argument");
    SweepArgs A{};
    if (int rc = upload_common(h, A, B, W, Rs, Ps, tic, ric, first_feature, start_frame, first_obs, obs, estimated_depth)) return rc;
    if (A.F == 0) return GF_OK;
    A.focal_length = focal_length;
    FS_HIP(hipEventRecord(h->ev0, h->stream));
    moving_consistency_kernel<<<dim3((A.F + 127) / 128), 128, 0, h->stream>>>(A);
    FS_HIP(hipGetLastError());
    FS_HIP(hipEventRecord(h->ev1, h->stream));
    FS_HIP(hipMemcpyAsync(remove, h->remove.d, sizeof(int) * A.F, hipMemcpyDeviceToHost, h->stream));
    FS_HIP(hipStreamSynchronize(h->stream));
    float ms = 0;
    if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) h->kernel_ms += ms;
    h->launches++; h->features += A.F;
    return GF_OK;
}
int gf_featsweep_stats(gf_featsweep* h, long long* launches, long long* features, double* kernel_ms) {
    if (!h) return gf::set_err(GF_ERR_INVALID, "null handle");
    if (launches) *launches = h->launches;
    if (features) *features = h->features;
    if (kernel_ms) *kernel_ms = h->kernel_ms;
    return GF_OK;
}
}  // extern "C"
