// gf_tracker.hip — C-ABI front end (include/groundfusion_hip.h): host orchestration of the HIP tracker.
//
// Mirrors FeatureTracker (vins_estimator/src/featureTracker/feature_tracker.{h,cpp}) for `batch`
// independent sequences driven in lock-step on one GPU stream.  Host keeps exactly the bookkeeping the
// reference keeps in C++ (ids, track_cnt, n_id, maps for velocity; setMask's std::sort + greedy keep,
// feature_tracker.cpp:56-83); all image work (pyramids, Scharr, LK forward/reverse, mask rasterisation,
// Shi-Tomasi, candidate sort and min-distance selection, depth sampling) runs in the kernels of
// gf_lk_kernels.hpp / gf_detect_kernels.hpp.  There is no CPU fallback: without a HIP device every entry
// point fails with GF_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <sched.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/groundfusion_hip.h"
#include "gf_comm.hpp"
#include "gf_detect_kernels.hpp"
#include "gf_lk_kernels.hpp"
#include "gf_copy_list.hpp"
#include "gf_host_cpus.hpp"

namespace gf {

static thread_local std::string g_err;
int set_err(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return gf::set_err(GF_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

static int require_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return set_err(GF_ERR_NO_DEVICE, "no HIP device available (%s); the HIP path has no CPU fallback", hipGetErrorString(e));
    return GF_OK;
}

struct P2f { float x, y; };

struct SeqState {  // per-sequence FeatureTracker members (feature_tracker.h:76-98)
    std::vector<P2f> prev_pts, cur_pts, predict_pts, prev_un_pts, cur_un_pts, pts_velocity;
    std::vector<int> ids, track_cnt;
    std::vector<uint16_t> cur_depth;
    // cur_un_pts_map / prev_un_pts_map (feature_tracker.h:89): id -> point, kept as id-sorted vectors (ids are unique)
    std::vector<std::pair<int, P2f>> cur_un_pts_map, prev_un_pts_map;
    std::vector<int> grid_head, grid_next;  // scratch of set_mask_host
    double cur_time = 0, prev_time = 0;
    int n_id = 0;
    bool hasPrediction = false;
};

// Small persistent pool for the per-sequence host bookkeeping (sequences are independent).  GF_HOST_THREADS overrides the size.
class HostPool {
  public:
    // device: the pool's threads go onto the cores of that GPU's NUMA node (8 ranks on one host: every rank's bookkeeping next to its own GPU and memory)
    explicit HostPool(int n, int device) {
        for (int i = 0; i < n; i++) workers_.emplace_back([this, device] { gf::pin_thread_to_device_node(device); run(); });
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; gen_++; }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    // calls fn(b) for b in [0, n); the calling thread takes part
    void parallel_for(int n, const std::function<void(int)>& fn) {
        if (workers_.empty() || n < 8) { for (int b = 0; b < n; b++) fn(b); return; }
        { std::lock_guard<std::mutex> l(m_); fn_ = &fn; n_ = n; next_ = 0; active_ = (int)workers_.size(); gen_++; }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [this] { return active_ == 0; });
        fn_ = nullptr;
    }
  private:
    void work() { for (;;) { const int b = next_.fetch_add(1); if (b >= n_) break; (*fn_)(b); } }
    void run() {
        unsigned long long seen = 0;
        for (;;) {
            { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&] { return gen_ != seen; }); seen = gen_; if (stop_) return; }
            work();
            { std::lock_guard<std::mutex> l(m_); if (--active_ == 0) done_.notify_one(); }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* fn_ = nullptr;
    std::atomic<int> next_{0};
    int n_ = 0, active_ = 0;
    unsigned long long gen_ = 0;
    bool stop_ = false;
};

template <class T> struct DevBuf {
    T* p = nullptr; size_t n = 0;
    int alloc(size_t count) {
        n = count;
        hipError_t e = hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e != hipSuccess) return set_err(GF_ERR_HIP, "hipMalloc(%zu B) failed: %s", count * sizeof(T), hipGetErrorString(e));
        // zeros, explicitly and finished before the handle's non-blocking stream can touch the buffer: hipMalloc returns whatever the previous owner left
        if (hipMemset(p, 0, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) return set_err(GF_ERR_HIP, "hipMemset failed");
        return GF_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; }
};
template <class T> struct PinBuf {
    T* p = nullptr; size_t n = 0;
    T* hd = nullptr;   // the same memory as kernels address it (page-locked memory is mapped into the device's address space), or null
    int alloc(size_t count) {
        n = count;
        hipError_t e = hipHostMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess) return set_err(GF_ERR_HIP, "hipHostMalloc failed: %s", hipGetErrorString(e));
        memset(p, 0, std::max<size_t>(count, 1) * sizeof(T));
        void* q = nullptr; hd = hipHostGetDevicePointer(&q, p, 0) == hipSuccess ? static_cast<T*>(q) : nullptr; (void)hipGetLastError();
        return GF_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; }
};

static void make_disk_table(int radius, DiskTable& T) {  // drawing.cpp Circle(): union of the h-lines per row offset
    T.radius = radius;
    for (int i = 0; i <= kMaxRadius; i++) T.hw[i] = -1;
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        T.hw[dy] = std::max<short>(T.hw[dy], (short)dx);
        T.hw[dx] = std::max<short>(T.hw[dx], (short)dy);
        dy++; err += plus; plus += 2;
        int mask = (err <= 0) - 1;
        err -= minus & mask; dx += mask; minus -= mask & 2;
    }
}

static int build_geom(int w, int h, PyrGeom& G) {
    memset(&G, 0, sizeof G);
    int lw = w, lh = h;
    size_t off = 0;
    int level = 0;
    for (; level < kMaxLevels; level++) {
        LevelGeom& g = G.lv[level];
        g.w = lw; g.h = lh; g.stride = lw + 2 * kPad;
        const size_t rows = lh + 2 * kPad;
        g.img_off = (int)(off + (size_t)kPad * g.stride + kPad);
        off += rows * g.stride;
        off = (off + 255) & ~(size_t)255;
        lw = (lw + 1) / 2; lh = (lh + 1) / 2;
        if (lw <= kWin || lh <= kWin) { level++; break; }  // buildOpticalFlowPyramid stop rule
    }
    G.nlevels = level;
    G.img_bytes = off;
    return GF_OK;
}

}  // namespace gf

using namespace gf;

struct gf_tracker {
    gf_tracker_cfg cfg;
    PyrGeom G;
    DiskTable disk;
    int B = 0, cap = 0, cand_cap = 0, frame = 0, cur_slot = 0, sort_cap = 0;
    bool copy_lists = true;   // GF_TRACKER_COPIES=1: one hipMemcpyAsync per table instead of the copy-list kernels
    bool pyr_head = true;     // GF_PYR_HEAD=0: levels 0 and 1 of the pyramid as two kernels (read when the tracker is created, as the other switches)
    bool select_topk = true;  // GF_SELECT_TOPK=0: every frame's corners through the sort
    int lk_points = 1;        // points per wavefront of the LK kernel: 1 (lk_track_kernel), 2 or 4 (lk_track_mp_kernel, round 6); GF_LK_POINTS
    bool profiling = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev[8] = {};
    gf_tracker_stats stats{};
    std::vector<SeqState> seq;
    HostPool* pool = nullptr;
    // device
    DevBuf<uint8_t> d_img, d_raw, d_mask, d_status, d_fwd_status, d_seqmask;
    DevBuf<int> d_npts, d_cand_count, d_want, d_ncenters, d_out_n;
    DevBuf<uint16_t> d_depth, d_depth_out, d_out_depth;
    // gf_tracker_prefetch_batch: the next frame's images on their way to the second pair of frame buffers while the current frame's kernels run
    DevBuf<uint8_t> d_raw2; DevBuf<uint16_t> d_depth2;
    hipStream_t copy_stream = nullptr; hipEvent_t ev_copy[2] = {nullptr, nullptr};
    int pf_head = 0, pf_count = 0;   // FIFO of staged frames over the two pairs (0: d_raw / d_depth, 1: d_raw2 / d_depth2): oldest pair, number staged (0..2)
    bool pf_depth[2] = {false, false};
    std::vector<const uint16_t*> pf_hdepth[2]; int pf_hdstride[2] = {0, 0};   // the staged frames' depth images (host; sampled by track_core, never copied)
    DevBuf<float2> d_prev_pts, d_init_pts, d_cur_pts, d_out_pts;
    DevBuf<unsigned> d_counters, d_maxkey;
    DevBuf<float> d_eig;
    DevBuf<unsigned long long> d_cand;
    DevBuf<int2> d_centers;
    // pinned host mirrors
    PinBuf<int> h_npts, h_want, h_ncenters, h_out_n, h_cand_count;
    PinBuf<float2> h_prev_pts, h_init_pts, h_cur_pts, h_out_pts;
    PinBuf<uint8_t> h_status, h_fwd_status, h_seqmask;
    PinBuf<uint16_t> h_depth_out, h_out_depth;
    PinBuf<unsigned> h_counters;
    PinBuf<int2> h_centers;
    size_t eig_stride = 0, mask_stride = 0;
    size_t select_lds = 0;

    void release() {
        d_raw2.release(); d_depth2.release();
        for (auto& e : ev_copy) if (e) (void)hipEventDestroy(e);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        d_img.release(); d_raw.release(); d_mask.release(); d_status.release(); d_fwd_status.release(); d_seqmask.release(); d_npts.release();
        d_cand_count.release(); d_want.release(); d_ncenters.release(); d_out_n.release(); d_depth.release(); d_depth_out.release();
        d_out_depth.release(); d_prev_pts.release(); d_init_pts.release(); d_cur_pts.release(); d_out_pts.release(); d_counters.release();
        d_maxkey.release(); d_eig.release(); d_cand.release(); d_centers.release();
        h_npts.release(); h_want.release(); h_ncenters.release(); h_out_n.release(); h_cand_count.release(); h_prev_pts.release();
        h_init_pts.release(); h_cur_pts.release(); h_out_pts.release(); h_status.release(); h_fwd_status.release(); h_seqmask.release(); h_depth_out.release();
        h_out_depth.release(); h_counters.release(); h_centers.release();
        for (auto& e : ev) if (e) (void)hipEventDestroy(e);
        if (stream) (void)hipStreamDestroy(stream);
        delete pool; pool = nullptr;
    }
};

namespace gf {

static inline int cvRoundf(float v) { return (int)lrintf(v); }

// camodocal PinholeCamera (camera_models/src/camera_models/PinholeCamera.cc:450-510, :520-542, :646-662)
static void distortion(const gf_tracker_cfg& c, double x, double y, double& dx, double& dy) {
    double mx2 = x * x, my2 = y * y, mxy = x * y, rho2 = mx2 + my2, rad = c.k1 * rho2 + c.k2 * rho2 * rho2;
    dx = x * rad + 2.0 * c.p1 * mxy + c.p2 * (rho2 + 2.0 * mx2);
    dy = y * rad + 2.0 * c.p2 * mxy + c.p1 * (rho2 + 2.0 * my2);
}
static bool no_distortion(const gf_tracker_cfg& c) { return c.k1 == 0.0 && c.k2 == 0.0 && c.p1 == 0.0 && c.p2 == 0.0; }
static void lift_projective(const gf_tracker_cfg& c, double u, double v, double& X, double& Y) {
    const double i11 = 1.0 / c.fx, i13 = -c.cx / c.fx, i22 = 1.0 / c.fy, i23 = -c.cy / c.fy;
    const double mx_d = i11 * u + i13, my_d = i22 * v + i23;
    if (no_distortion(c)) { X = mx_d; Y = my_d; return; }
    double dux, duy;
    distortion(c, mx_d, my_d, dux, duy);
    double mx_u = mx_d - dux, my_u = my_d - duy;
    for (int i = 1; i < 8; ++i) { distortion(c, mx_u, my_u, dux, duy); mx_u = mx_d - dux; my_u = my_d - duy; }
    X = mx_u; Y = my_u;
}
static void space_to_plane(const gf_tracker_cfg& c, const double* P, double& u, double& v) {
    double xu = P[0] / P[2], yu = P[1] / P[2], xd = xu, yd = yu;
    if (!no_distortion(c)) { double dx, dy; distortion(c, xu, yu, dx, dy); xd = xu + dx; yd = yu + dy; }
    u = c.fx * xd + c.cx; v = c.fy * yd + c.cy;
}

template <class V> static void reduce_vector(std::vector<V>& v, const uint8_t* st) {  // feature_tracker.cpp:30-46
    int j = 0;
    for (int i = 0; i < (int)v.size(); i++) if (st[i]) v[j++] = v[i];
    v.resize(j);
}

// buildOpticalFlowPyramid in three launches: level 0 (copy + REFLECT_101 border), level 1 (interior + border in one pass), and one kernel for
// all remaining levels.  There is no derivative pyramid: lk_solve evaluates the Scharr derivative of the template window itself.
static int launch_pyramid(gf_tracker* h, const uint8_t* d_raw_frames) {
    const PyrGeom& G = h->G;
    const size_t seq_img = 2 * G.img_bytes;
    uint8_t* img = h->d_img.p + (size_t)h->cur_slot * G.img_bytes;
    const LevelGeom g0 = G.lv[0];
    const bool v16 = !((g0.w | g0.stride | (int)(((size_t)g0.w * g0.h) & 15) | (int)(seq_img & 15) | (int)((g0.img_off - kPad * g0.stride - kPad) & 15)) & 15) &&
                     !((reinterpret_cast<uintptr_t>(d_raw_frames) | reinterpret_cast<uintptr_t>(img)) & 15);
    bool vec = true;   // four-pixel kernels need level widths (and with them strides, offsets) that are multiples of 4
    for (int l = 0; l < G.nlevels; l++) vec = vec && !(G.lv[l].w & 3) && !(G.lv[l].img_off & 3);
    for (int l = 1; l < G.nlevels; l++) vec = vec && G.lv[l].w > kPad + 1 && G.lv[l].h > kPad + 1;   // one reflection reaches every border pixel
    // levels 0 and 1 from one read of the raw frame (pyr_head_kernel) when the sizes allow it; GF_PYR_HEAD=0 keeps the two kernels (A/B and fallback)
    const bool head = h->pyr_head && v16 && vec && G.nlevels >= 2 && !(g0.h & 1) && G.lv[1].h * 2 == g0.h && G.lv[1].w * 2 == g0.w && g0.h > kPad + 2 && g0.w > kPad + 2 &&
                      pyr_head_lds_bytes(g0.w) <= 64 * 1024;
    if (head) {
        pyr_head_kernel<<<dim3((G.lv[1].h + kHeadRows - 1) / kHeadRows, h->B), 512, pyr_head_lds_bytes(g0.w), h->stream>>>(d_raw_frames, (size_t)g0.w * g0.h, g0.w, img, seq_img, g0, G.lv[1]);
    } else if (v16) {
        const int n = ((g0.w + 2 * kPad) / 16) * (g0.h + 2 * kPad);
        pyr_level0_vec16_kernel<<<dim3((n + 255) / 256, h->B), 256, 0, h->stream>>>(d_raw_frames, (size_t)g0.w * g0.h, g0.w, img, seq_img, g0);
    } else {
        const int n = ((g0.w + 2 * kPad) / 4) * (g0.h + 2 * kPad);
        pyr_level0_kernel<<<dim3((n + 255) / 256, h->B), 256, 0, h->stream>>>(d_raw_frames, (size_t)g0.w * g0.h, g0.w, img, seq_img, g0);
    }
    if (vec) {
        auto down = [&](int l) {
            const LevelGeom d = G.lv[l];
            pyr_down_pad4_kernel<<<dim3(((d.w >> 2) * d.h + 255) / 256, h->B), 256, 0, h->stream>>>(img, seq_img, G.lv[l - 1], d);
        };
        if (G.nlevels > 1 && !head) down(1);
        if (G.nlevels > 2) {
            const int parts = 4, last = std::min(G.nlevels - 1, 3);
            const LevelGeom d = G.lv[2], e = G.lv[last];
            const int band = last > 2 ? (e.h + parts - 1) / parts + 1 : ((d.h + 1) / 2 + parts - 1) / parts + 1;
            const size_t lds = (size_t)(2 * band + 4) * d.w + (last > 2 ? (size_t)band * e.w : 0);
            if (G.nlevels <= 4 && lds <= 64 * 1024) pyr_down_tail_kernel<<<dim3(parts, h->B), 512, lds, h->stream>>>(img, seq_img, G, 2);
            else for (int l = 2; l < G.nlevels; l++) down(l);
        }
    } else {
        for (int l = 1; l < G.nlevels; l++) {
            const LevelGeom d = G.lv[l];
            const int n = (d.w + 2 * kPad) * (d.h + 2 * kPad);
            pyr_down_kernel<<<dim3((n + 255) / 256, h->B), 256, 0, h->stream>>>(img, seq_img, G.lv[l - 1], d);
        }
    }
    HIPCHK(hipGetLastError());
    return GF_OK;
}

static void launch_lk(gf_tracker* h, int batch, const LkBatchArgs& A) {   // one LK launch over `batch` sequences in the form the handle was created for (same results in every form)
    const int cap = h->cap;
    if (h->lk_points == 4) lk_track_mp_kernel<4><<<dim3((cap + 15) / 16, batch), 256, 0, h->stream>>>(h->G, A);
    else if (h->lk_points == 2) lk_track_mp_kernel<2><<<dim3((cap + 7) / 8, batch), 256, 0, h->stream>>>(h->G, A);
    else lk_track_kernel<<<dim3((cap + 3) / 4, batch), 256, 0, h->stream>>>(h->G, A);
}

static LkBatchArgs lk_args(gf_tracker* h, int fwd_max_level, int use_init, int flow_back, int post_checks, const uint8_t* seqmask,
                           const uint16_t* d_depth) {
    LkBatchArgs A{};
    A.img = h->d_img.p; A.prev_slot = 1 - h->cur_slot; A.cap = h->cap;
    A.n_pts = h->d_npts.p; A.prev_pts = h->d_prev_pts.p; A.init_pts = h->d_init_pts.p; A.cur_pts = h->d_cur_pts.p;
    A.status = h->d_status.p; A.fwd_status = h->d_fwd_status.p; A.depth_out = h->d_depth_out.p; A.depth = d_depth;
    A.depth_seq_stride = (size_t)h->cfg.width * h->cfg.height; A.depth_stride = h->cfg.width;
    A.counters = h->d_counters.p; A.fwd_max_level = fwd_max_level; A.fwd_use_init = use_init; A.flow_back = flow_back;
    A.post_checks = post_checks; A.seq_mask = seqmask;
    return A;
}

// setMask (feature_tracker.cpp:56-83): std::sort by track count (same comparator, same libstdc++ algorithm as
// the reference) and greedy keep of points not covered by an earlier kept point's filled circle.
static void set_mask_host(gf_tracker* h, SeqState& s, int2* centers, int& n_centers) {
    struct E { int cnt; P2f pt; int id; uint16_t depth; };
    static thread_local std::vector<E> v;
    v.clear();
    for (size_t i = 0; i < s.cur_pts.size(); i++) v.push_back({s.track_cnt[i], s.cur_pts[i], s.ids[i], s.cur_depth[i]});
    std::sort(v.begin(), v.end(), [](const E& a, const E& b) { return a.cnt > b.cnt; });
    s.cur_pts.clear(); s.ids.clear(); s.track_cnt.clear(); s.cur_depth.clear();
    n_centers = 0;
    const DiskTable& T = h->disk;
    // `mask.at(pt) == 255` <=> pt is inside no earlier kept point's filled circle; circles reach at most `radius`
    // pixels, so only centres in the 3x3 neighbourhood of radius-sized cells can cover pt.
    const int cell = std::max(T.radius, 1);
    const int gw = h->cfg.width / cell + 1, gh = h->cfg.height / cell + 1;
    s.grid_head.assign((size_t)gw * gh, -1);
    s.grid_next.clear();
    for (auto& it : v) {
        const int x = cvRoundf(it.pt.x), y = cvRoundf(it.pt.y);
        const int cx = x / cell, cy = y / cell;
        bool covered = false;
        for (int yy = std::max(cy - 1, 0); yy <= std::min(cy + 1, gh - 1) && !covered; yy++)
            for (int xx = std::max(cx - 1, 0); xx <= std::min(cx + 1, gw - 1) && !covered; xx++)
                for (int k = s.grid_head[(size_t)yy * gw + xx]; k >= 0; k = s.grid_next[k]) {
                    const int dy = std::abs(y - centers[k].y), dx = std::abs(x - centers[k].x);
                    if (dy <= T.radius && dx <= T.hw[dy]) { covered = true; break; }
                }
        if (!covered) {
            s.cur_pts.push_back(it.pt); s.ids.push_back(it.id); s.track_cnt.push_back(it.cnt); s.cur_depth.push_back(it.depth);
            centers[n_centers] = make_int2(x, y);
            s.grid_next.push_back(s.grid_head[(size_t)cy * gw + cx]);
            s.grid_head[(size_t)cy * gw + cx] = n_centers++;
        }
    }
}

static void pts_velocity(SeqState& s) {  // feature_tracker.cpp:810-847
    const size_t n = s.ids.size();
    s.pts_velocity.resize(n);
    s.cur_un_pts_map.resize(n);
    for (size_t i = 0; i < n; i++) s.cur_un_pts_map[i] = {s.ids[i], s.cur_un_pts[i]};
    std::sort(s.cur_un_pts_map.begin(), s.cur_un_pts_map.end(), [](const std::pair<int, P2f>& a, const std::pair<int, P2f>& b) { return a.first < b.first; });
    if (!s.prev_un_pts_map.empty()) {
        const double dt = s.cur_time - s.prev_time;
        for (size_t i = 0; i < n; i++) {
            const int id = s.ids[i];
            auto it = std::lower_bound(s.prev_un_pts_map.begin(), s.prev_un_pts_map.end(), id, [](const std::pair<int, P2f>& a, int v) { return a.first < v; });
            if (it != s.prev_un_pts_map.end() && it->first == id) {
                const double vx = (s.cur_un_pts[i].x - it->second.x) / dt, vy = (s.cur_un_pts[i].y - it->second.y) / dt;
                s.pts_velocity[i] = {(float)vx, (float)vy};
            } else s.pts_velocity[i] = {0.f, 0.f};
        }
    } else for (size_t i = 0; i < n; i++) s.pts_velocity[i] = {0.f, 0.f};
}

// hdep (host entry points): the callers' depth images, one pointer per sequence, rows of hdstride pixels.  The reference reads ONE pixel of the depth image per
// feature (feature_tracker.cpp:360 `rightImg.at<ushort>(round(y), round(x))`), and on the host-image entry points both the image and the feature coordinates are on
// the host anyway: sampling there keeps 614 KB per frame and sequence (two thirds of an RGB-D VGA frame) off the bus.  d_depth (device entry point) keeps the sample
// in the kernels.  Same pixel, same rounding (round half away from zero on the float coordinates), same u16.
static int track_core(gf_tracker* h, const double* t, const uint8_t* d_gray, const uint16_t* d_depth, gf_feature_obs* out, int cap_out,
                      int* n_out, const uint16_t* const* hdep = nullptr, int hdstride = 0) {
    const int B = h->B, cap = h->cap, W = h->cfg.width, H = h->cfg.height;
    const bool have_depth = d_depth != nullptr || hdep != nullptr;
    const bool prof = h->profiling;
    h->cur_slot = h->frame & 1;
    using clk = std::chrono::steady_clock;
    auto tp = clk::now();
    auto lap = [&](double& acc) { auto n = clk::now(); acc += std::chrono::duration<double, std::milli>(n - tp).count(); tp = n; };
    for (int b = 0; b < B; b++) { h->seq[b].cur_time = t[b]; h->seq[b].cur_pts.clear(); h->seq[b].cur_depth.clear(); }
    // The hand-overs between the host's bookkeeping and the kernels -- two to four small tables down before LK, five up behind it, three down before the detection, four
    // up behind it -- as one copy-list kernel each (gf_copy_list.hpp) instead of one hipMemcpyAsync per table: sixteen submissions and copy latencies per frame become four.
    gfcopy::Builder CL;
    const bool lists = h->copy_lists;
    hipError_t cerr = hipSuccess;
    auto down = [&](auto& dev, auto& pin, size_t count) {   // host -> device
        const size_t bytes = count * sizeof(*pin.p);
        if (lists) CL.add(pin.hd, dev.p, 1, bytes, bytes); else if (cerr == hipSuccess) cerr = hipMemcpyAsync(dev.p, pin.p, bytes, hipMemcpyHostToDevice, h->stream);
    };
    auto up = [&](auto& pin, auto& dev, size_t count) {     // device -> host
        const size_t bytes = count * sizeof(*pin.p);
        if (lists) CL.add(dev.p, pin.hd, 1, bytes, bytes); else if (cerr == hipSuccess) cerr = hipMemcpyAsync(pin.p, dev.p, bytes, hipMemcpyDeviceToHost, h->stream);
    };
    auto flush = [&]() -> int {
        HIPCHK(cerr);
        if (lists) { if (!CL.ok) return set_err(GF_ERR_HIP, "copy list: a page-locked buffer is not mapped into the device's address space (GF_TRACKER_COPIES=1 selects plain copies)"); HIPCHK(CL.launch<1>(h->stream)); CL = gfcopy::Builder(); }
        return GF_OK;
    };
    if (prof) HIPCHK(hipEventRecord(h->ev[0], h->stream));
    if (int rc = launch_pyramid(h, d_gray)) return rc;
    if (prof) HIPCHK(hipEventRecord(h->ev[1], h->stream));

    // ---- temporal optical flow (feature_tracker.cpp:113-176)
    bool any_prev = false, any_pred = false, any_plain = false;
    for (int b = 0; b < B; b++) {
        SeqState& s = h->seq[b];
        const int n = (int)s.prev_pts.size();
        h->h_npts.p[b] = n;
        if (n > cap) return set_err(GF_ERR_CAPACITY, "sequence %d holds %d points > capacity %d", b, n, cap);
        for (int i = 0; i < n; i++) h->h_prev_pts.p[(size_t)b * cap + i] = make_float2(s.prev_pts[i].x, s.prev_pts[i].y);
        if (n > 0) {
            any_prev = true;
            if (s.hasPrediction && (int)s.predict_pts.size() != n) return set_err(GF_ERR_INVALID, "sequence %d: prediction holds %d points but %d are tracked (call removeOutliers before setPrediction, estimator.cpp:1134-1135)", b, (int)s.predict_pts.size(), n);
            if (s.hasPrediction) { any_pred = true; for (int i = 0; i < n; i++) h->h_init_pts.p[(size_t)b * cap + i] = make_float2(s.predict_pts[i].x, s.predict_pts[i].y); }
            else any_plain = true;
        }
    }
    bool lk_timed = false;
    if (any_prev) {
        down(h->d_npts, h->h_npts, B);
        down(h->d_prev_pts, h->h_prev_pts, (size_t)B * cap);
        const uint8_t* mask_plain = nullptr; const uint8_t* mask_pred = nullptr;
        if (any_pred) {
            down(h->d_init_pts, h->h_init_pts, (size_t)B * cap);
            for (int b = 0; b < B; b++) { h->h_seqmask.p[b] = h->seq[b].hasPrediction ? 0 : 1; h->h_seqmask.p[B + b] = h->seq[b].hasPrediction ? 1 : 0; }
            down(h->d_seqmask, h->h_seqmask, 2 * (size_t)B);
            mask_plain = h->d_seqmask.p; mask_pred = h->d_seqmask.p + B;
        }
        if (int rc = flush()) return rc;
        if (prof) HIPCHK(hipEventRecord(h->ev[2], h->stream));
        if (any_plain) { launch_lk(h, B, lk_args(h, 3, 0, h->cfg.flow_back, 1, mask_plain, d_depth)); h->stats.lk_launches++; }
        if (any_pred) { launch_lk(h, B, lk_args(h, 1, 1, h->cfg.flow_back, 1, mask_pred, d_depth)); h->stats.lk_launches++; }
        HIPCHK(hipGetLastError());
        if (prof) { HIPCHK(hipEventRecord(h->ev[3], h->stream)); lk_timed = true; }
        up(h->h_cur_pts, h->d_cur_pts, (size_t)B * cap);
        up(h->h_status, h->d_status, (size_t)B * cap);
        up(h->h_fwd_status, h->d_fwd_status, (size_t)B * cap);
        up(h->h_depth_out, h->d_depth_out, (size_t)B * cap);
        up(h->h_counters, h->d_counters, (size_t)B * cap * 2);
        if (int rc = flush()) return rc;
    }
    HIPCHK(hipEventRecord(h->ev[6], h->stream));
    lap(h->stats.ms_host_pre);
    HIPCHK(hipEventSynchronize(h->ev[6]));
    lap(h->stats.ms_wait_lk);

    if (any_pred) {  // feature_tracker.cpp:124-132: fewer than 10 forward successes -> redo with 3 levels from scratch
        bool need = false;
        for (int b = 0; b < B; b++) {
            h->h_seqmask.p[b] = 0;
            if (!h->seq[b].hasPrediction || h->h_npts.p[b] == 0) continue;
            int succ = 0;
            for (int i = 0; i < h->h_npts.p[b]; i++) succ += h->h_fwd_status.p[(size_t)b * cap + i] ? 1 : 0;
            if (succ < 10) { h->h_seqmask.p[b] = 1; need = true; }
        }
        if (need) {
            std::vector<uint8_t> redo(h->h_seqmask.p, h->h_seqmask.p + B);
            std::vector<uint8_t> keep_status(h->h_status.p, h->h_status.p + (size_t)B * cap);
            std::vector<float2> keep_pts(h->h_cur_pts.p, h->h_cur_pts.p + (size_t)B * cap);
            std::vector<uint16_t> keep_depth(h->h_depth_out.p, h->h_depth_out.p + (size_t)B * cap);
            std::vector<unsigned> keep_cnt(h->h_counters.p, h->h_counters.p + (size_t)B * cap * 2);
            HIPCHK(hipMemcpyAsync(h->d_seqmask.p, h->h_seqmask.p, B, hipMemcpyHostToDevice, h->stream));
            launch_lk(h, B, lk_args(h, 3, 0, h->cfg.flow_back, 1, h->d_seqmask.p, d_depth));
            h->stats.lk_launches++;
            HIPCHK(hipGetLastError());
            up(h->h_cur_pts, h->d_cur_pts, (size_t)B * cap);
            up(h->h_status, h->d_status, (size_t)B * cap);
            up(h->h_depth_out, h->d_depth_out, (size_t)B * cap);
            up(h->h_counters, h->d_counters, (size_t)B * cap * 2);
            if (int rc = flush()) return rc;
            HIPCHK(hipStreamSynchronize(h->stream));
            for (int b = 0; b < B; b++) {
                unsigned* cn = h->h_counters.p + (size_t)b * cap * 2;
                const unsigned* kc = keep_cnt.data() + (size_t)b * cap * 2;
                if (redo[b]) { for (int i = 0; i < 2 * cap; i++) cn[i] += kc[i]; continue; }  // both passes did work
                memcpy(h->h_status.p + (size_t)b * cap, keep_status.data() + (size_t)b * cap, cap);
                memcpy(h->h_cur_pts.p + (size_t)b * cap, keep_pts.data() + (size_t)b * cap, cap * sizeof(float2));
                memcpy(h->h_depth_out.p + (size_t)b * cap, keep_depth.data() + (size_t)b * cap, cap * sizeof(uint16_t));
                memcpy(cn, kc, cap * 2 * sizeof(unsigned));
            }
        }
    }

    // ---- host bookkeeping per sequence (feature_tracker.cpp:170-186)
    std::atomic<long long> a_levels{0}, a_iters{0}, a_points{0}, a_tracked{0};
    std::atomic<int> a_want{0};
    h->pool->parallel_for(B, [&](int b) {
        SeqState& s = h->seq[b];
        const int n = (int)s.prev_pts.size();
        if (n > 0) {
            const uint8_t* st = h->h_status.p + (size_t)b * cap;
            s.cur_pts.resize(n); s.cur_depth.resize(n);
            long long lv = 0, it = 0;
            for (int i = 0; i < n; i++) {
                const float2 c = h->h_cur_pts.p[(size_t)b * cap + i];
                s.cur_pts[i] = {c.x, c.y};
                if (hdep) { const int ry = (int)std::round((double)c.y), rx = (int)std::round((double)c.x); s.cur_depth[i] = st[i] ? hdep[b][(size_t)ry * hdstride + rx] : (uint16_t)0; }   // st: inside the image (post checks)
                else s.cur_depth[i] = h->h_depth_out.p[(size_t)b * cap + i];
                lv += h->h_counters.p[2 * ((size_t)b * cap + i)];
                it += h->h_counters.p[2 * ((size_t)b * cap + i) + 1];
            }
            a_levels += lv; a_iters += it; a_points += n;
            reduce_vector(s.prev_pts, st); reduce_vector(s.cur_pts, st); reduce_vector(s.ids, st); reduce_vector(s.track_cnt, st);
            reduce_vector(s.cur_depth, st);
            a_tracked += (long long)s.cur_pts.size();
        }
        for (auto& c : s.track_cnt) c++;
        int nc = 0;
        set_mask_host(h, s, h->h_centers.p + (size_t)b * cap, nc);
        h->h_ncenters.p[b] = nc;
        const int want = h->cfg.max_cnt - (int)s.cur_pts.size();
        h->h_want.p[b] = want;
        if (want > 0) a_want = 1;
        h->h_out_n.p[b] = 0;
    });
    const bool any_want = a_want.load() != 0;
    h->stats.lk_level_passes += a_levels; h->stats.lk_iterations += a_iters; h->stats.lk_points += a_points; h->stats.tracked_features += a_tracked;

    // ---- Shi-Tomasi top-up (feature_tracker.cpp:190-206)
    if (any_want) {
        down(h->d_centers, h->h_centers, (size_t)B * cap);
        down(h->d_ncenters, h->h_ncenters, B);
        down(h->d_want, h->h_want, B);
        if (int rc = flush()) return rc;
        HIPCHK(hipMemsetAsync(h->d_maxkey.p, 0, B * sizeof(unsigned), h->stream));
        HIPCHK(hipMemsetAsync(h->d_cand_count.p, 0, B * sizeof(int), h->stream));
        if (prof) HIPCHK(hipEventRecord(h->ev[4], h->stream));
        {
            DetectArgs D{};
            D.pyr = h->d_img.p + (size_t)h->cur_slot * h->G.img_bytes; D.pyr_seq_stride = 2 * h->G.img_bytes; D.g = h->G.lv[0];
            D.mask = nullptr; D.mask_seq_stride = 0; D.centers = h->d_centers.p; D.n_centers = h->d_ncenters.p; D.cap = cap; D.want = h->d_want.p;
            D.maxkey = h->d_maxkey.p; D.cand = h->d_cand.p; D.cand_seq_stride = (size_t)h->cand_cap; D.cand_cap = h->cand_cap; D.cand_count = h->d_cand_count.p;
            detect_strip_kernel<kDS_R><<<dim3((W + kDS_W - 1) / kDS_W, (H + kDS_R - 1) / kDS_R, B), 64, 0, h->stream>>>(D, h->disk);
        }
        SelectArgs S{};
        S.cand = h->d_cand.p; S.cand_seq_stride = (size_t)h->cand_cap; S.cand_cap = h->cand_cap; S.cand_count = h->d_cand_count.p; S.maxkey = h->d_maxkey.p; S.want = h->d_want.p;
        S.w = W; S.h = H; S.min_dist = h->cfg.min_dist; S.out_cap = cap; S.sort_cap = h->sort_cap; S.out_pts = h->d_out_pts.p; S.out_depth = h->d_out_depth.p; S.out_n = h->d_out_n.p;
        S.depth = d_depth; S.depth_seq_stride = (size_t)W * H; S.depth_stride = W;
        {   // the sequences that want a handful of corners (every frame but the first ones): one maximum per corner instead of a sort (select_topk_kernel; GF_SELECT_TOPK=0: off)
            const bool topk_on = h->select_topk;
            int max_want = 0;
            for (int b = 0; b < B; b++) max_want = std::max(max_want, h->h_want.p[b]);
            S.skip_small = topk_on ? 1 : 0;
            if (topk_on) select_topk_kernel<<<dim3(B), 1024, 0, h->stream>>>(S);
            if (!topk_on || max_want > kTopKMax) select_corners_kernel<<<dim3(B), 1024, h->select_lds, h->stream>>>(S);
        }
        HIPCHK(hipGetLastError());
        up(h->h_out_n, h->d_out_n, B);
        up(h->h_out_pts, h->d_out_pts, (size_t)B * cap);
        up(h->h_out_depth, h->d_out_depth, (size_t)B * cap);
        up(h->h_cand_count, h->d_cand_count, B);
        if (int rc = flush()) return rc;
    }
    if (prof && !any_want) HIPCHK(hipEventRecord(h->ev[4], h->stream));
    if (prof) HIPCHK(hipEventRecord(h->ev[5], h->stream));
    lap(h->stats.ms_host_mid);
    HIPCHK(hipStreamSynchronize(h->stream));
    lap(h->stats.ms_wait_detect);
    if (prof) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, h->ev[0], h->ev[1])); h->stats.ms_pyramid += ms;
        if (lk_timed) { HIPCHK(hipEventElapsedTime(&ms, h->ev[2], h->ev[3])); h->stats.ms_lk += ms; }
        HIPCHK(hipEventElapsedTime(&ms, h->ev[4], h->ev[5])); h->stats.ms_detect += ms;
        HIPCHK(hipEventElapsedTime(&ms, h->ev[0], h->ev[5])); h->stats.ms_total_gpu += ms;
    }
    if (any_want)
        for (int b = 0; b < B; b++) {
            if (h->h_want.p[b] > 0 && h->h_cand_count.p[b] > h->cand_cap)
                return set_err(GF_ERR_CAPACITY, "sequence %d: %d corner candidates exceed capacity %d", b, h->h_cand_count.p[b], h->cand_cap);
        }

    // ---- addPoints, undistortedPts, ptsVelocity, pack (feature_tracker.cpp:85-93, 210-211, 322-368)
    std::atomic<int> a_overflow{-1};
    std::atomic<long long> a_out{0};
    h->pool->parallel_for(B, [&](int b) {
        SeqState& s = h->seq[b];
        const int nn = h->h_want.p[b] > 0 ? h->h_out_n.p[b] : 0;
        for (int i = 0; i < nn; i++) {
            const float2 p = h->h_out_pts.p[(size_t)b * cap + i];
            s.cur_pts.push_back({p.x, p.y}); s.ids.push_back(s.n_id++); s.track_cnt.push_back(1);
            s.cur_depth.push_back(hdep ? hdep[b][(size_t)(int)p.y * hdstride + (int)p.x] : h->h_out_depth.p[(size_t)b * cap + i]);   // corners sit on pixel centres
        }
        s.cur_un_pts.clear();
        for (auto& p : s.cur_pts) { double X, Y; lift_projective(h->cfg, (double)p.x, (double)p.y, X, Y); s.cur_un_pts.push_back({(float)(X / 1.0), (float)(Y / 1.0)}); }
        pts_velocity(s);
        s.prev_pts = s.cur_pts; s.prev_un_pts = s.cur_un_pts; s.prev_un_pts_map.swap(s.cur_un_pts_map); s.prev_time = s.cur_time;
        s.hasPrediction = false;
        const int n = (int)s.ids.size();
        // depth_cam set but no depth image: neither packing loop of the reference runs (feature_tracker.cpp:320 `depth_cam == 0`, :344 `!_img1.empty()`):
        // the returned featureFrame is empty, the tracker state has advanced all the same
        if (h->cfg.depth_cam && !have_depth) { n_out[b] = 0; return; }
        if (n > cap_out) { a_overflow = n; n_out[b] = 0; return; }
        gf_feature_obs* o = out + (size_t)b * cap_out;
        for (int i = 0; i < n; i++) {
            o[i].id = s.ids[i]; o[i].camera_id = 0;
            o[i].v[0] = s.cur_un_pts[i].x; o[i].v[1] = s.cur_un_pts[i].y; o[i].v[2] = 1; o[i].v[3] = s.cur_pts[i].x; o[i].v[4] = s.cur_pts[i].y;
            o[i].v[5] = s.pts_velocity[i].x; o[i].v[6] = s.pts_velocity[i].y;
            o[i].v[7] = h->cfg.depth_cam ? (double)(int)s.cur_depth[i] / 1000 : -2.4;
        }
        n_out[b] = n;
        a_out += n;
    });
    if (a_overflow.load() >= 0) return set_err(GF_ERR_CAPACITY, "output capacity %d < %d features", cap_out, a_overflow.load());
    h->stats.output_features += a_out;
    h->frame++;
    h->stats.frames++;
    lap(h->stats.ms_host_post);
    return GF_OK;
}

}  // namespace gf

// =============================================================================== C-ABI
extern "C" {

const char* gf_last_error(void) { return gf::g_err.c_str(); }

int gf_device_count(int* n) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return gf::set_err(GF_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *n = c;
    return GF_OK;
}
int gf_set_device(int device) { HIPCHK(hipSetDevice(device)); return GF_OK; }

int gf_tracker_create(const gf_tracker_cfg* cfg, gf_tracker** out) {
    if (!cfg || !out) return gf::set_err(GF_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->width < 32 || cfg->height < 32 || cfg->width % 4 || cfg->batch < 1 || cfg->max_cnt < 1 || cfg->min_dist < 0 || cfg->min_dist > gf::kMaxRadius)
        return gf::set_err(GF_ERR_INVALID, "unsupported tracker configuration (width %% 4 == 0, width/height >= 32, 0 <= min_dist <= %d)", gf::kMaxRadius);
    if (int rc = gf::require_device()) return rc;
    gf_tracker* h = new gf_tracker();
    h->cfg = *cfg;
    h->B = cfg->batch;
    h->cap = (cfg->max_cnt + 3) & ~3;
    gf::build_geom(cfg->width, cfg->height, h->G);
    gf::make_disk_table(cfg->min_dist, h->disk);
    h->seq.resize(h->B);
    {
        // default: up to 16 threads out of this rank's share of the node (a frame of 256 sequences spends 1.3 ms in the bookkeeping with 4 threads, 0.5 ms with 16)
        int share = 1;
        int hw_box = 0, hw = 0;   // ... of the hardware threads this process can really use (its affinity mask and its container's CPU quota, not the box: gf_host_cpus.hpp)
        gf::host_cpus(hw_box, hw);
        // divide among the node's ranks only when the mask is the whole machine: a launcher that pins each rank has divided already (round-5 advisor)
        if (const char* e = getenv("LOCAL_WORLD_SIZE")) if (hw_box <= 0 || hw >= hw_box) share = std::max(1, atoi(e));
        int nthr = std::max(1, std::min(16, hw / (2 * share)));
        if (const char* e = getenv("GF_HOST_THREADS")) nthr = atoi(e);
        nthr = std::max(1, std::min(nthr, std::max(hw, 1)));
        int dev = 0;
        (void)hipGetDevice(&dev);
        h->pool = new gf::HostPool(h->B >= 8 ? nthr - 1 : 0, dev);
        h->copy_lists = !(getenv("GF_TRACKER_COPIES") && atoi(getenv("GF_TRACKER_COPIES")) != 0);
        h->pyr_head = !(getenv("GF_PYR_HEAD") && atoi(getenv("GF_PYR_HEAD")) == 0);
        h->select_topk = !(getenv("GF_SELECT_TOPK") && atoi(getenv("GF_SELECT_TOPK")) == 0);
        if (const char* e = getenv("GF_LK_POINTS")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) h->lk_points = v; }
    }
    const int W = cfg->width, H = cfg->height, B = h->B, cap = h->cap;
    int cc = 1024;
    while (cc < (W * H) / 2) cc <<= 1;
    h->cand_cap = cc;
    h->eig_stride = ((size_t)W * H + 3) & ~(size_t)3;
    h->mask_stride = ((size_t)W * H + 15) & ~(size_t)15;
    const int cell = std::max(cfg->min_dist, 1);
    const int gw = (W + cell - 1) / cell, gh = (H + cell - 1) / cell;
    const size_t grid_lds = (size_t)((gw * gh + 3) & ~3) * 2 + (size_t)cap * 3 * 2 + 64;
    h->sort_cap = gf::kSortLds;
    while (h->sort_cap > 64 && (size_t)h->sort_cap * 8 + grid_lds > 160 * 1024) h->sort_cap >>= 1;
    h->select_lds = (size_t)h->sort_cap * 8 + grid_lds;
    if (h->select_lds > 160 * 1024) { delete h; return gf::set_err(GF_ERR_INVALID, "min_dist %d too small for the selection grid at %dx%d", cfg->min_dist, W, H); }
#define A_(x) do { if (int rc_ = (x)) { h->release(); delete h; return rc_; } } while (0)
#define H_(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { gf::set_err(GF_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_)); h->release(); delete h; return GF_ERR_HIP; } } while (0)
    H_(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    for (auto& e : h->ev) H_(hipEventCreate(&e));
    A_(h->d_img.alloc((size_t)B * 2 * h->G.img_bytes));
    A_(h->d_raw.alloc((size_t)B * W * H));
    // (no device copy of the depth images: the host entry points sample them on the host, the device entry point reads the caller's device pointer)
    A_(h->d_mask.alloc(h->mask_stride));  // explicit masks exist only in the gf_good_features building block
    A_(h->d_eig.alloc(h->eig_stride));    // response image materialised only by gf_min_eigen_val
    A_(h->d_cand.alloc((size_t)B * h->cand_cap));
    A_(h->d_status.alloc((size_t)B * cap)); A_(h->d_fwd_status.alloc((size_t)B * cap)); A_(h->d_seqmask.alloc(2 * (size_t)B));
    A_(h->d_npts.alloc(B)); A_(h->d_cand_count.alloc(B)); A_(h->d_want.alloc(B)); A_(h->d_ncenters.alloc(B)); A_(h->d_out_n.alloc(B));
    A_(h->d_depth_out.alloc((size_t)B * cap)); A_(h->d_out_depth.alloc((size_t)B * cap));
    A_(h->d_prev_pts.alloc((size_t)B * cap)); A_(h->d_init_pts.alloc((size_t)B * cap)); A_(h->d_cur_pts.alloc((size_t)B * cap)); A_(h->d_out_pts.alloc((size_t)B * cap));
    A_(h->d_counters.alloc((size_t)B * cap * 2)); A_(h->d_maxkey.alloc(B)); A_(h->d_centers.alloc((size_t)B * cap));
    A_(h->h_npts.alloc(B)); A_(h->h_want.alloc(B)); A_(h->h_ncenters.alloc(B)); A_(h->h_out_n.alloc(B)); A_(h->h_cand_count.alloc(B));
    A_(h->h_prev_pts.alloc((size_t)B * cap)); A_(h->h_init_pts.alloc((size_t)B * cap)); A_(h->h_cur_pts.alloc((size_t)B * cap)); A_(h->h_out_pts.alloc((size_t)B * cap));
    A_(h->h_status.alloc((size_t)B * cap)); A_(h->h_fwd_status.alloc((size_t)B * cap)); A_(h->h_seqmask.alloc(2 * (size_t)B)); A_(h->h_depth_out.alloc((size_t)B * cap)); A_(h->h_out_depth.alloc((size_t)B * cap));
    A_(h->h_counters.alloc((size_t)B * cap * 2)); A_(h->h_centers.alloc((size_t)B * cap));
    H_(hipMemsetAsync(h->d_img.p, 0, h->d_img.n, h->stream));
    H_(hipFuncSetAttribute(reinterpret_cast<const void*>(gf::select_corners_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->select_lds));
    H_(hipStreamSynchronize(h->stream));
#undef A_
#undef H_
    *out = h;
    return GF_OK;
}

int gf_tracker_destroy(gf_tracker* h) {
    if (!h) return GF_OK;
    h->release();
    delete h;
    return GF_OK;
}

int gf_tracker_track_batch_device(gf_tracker* h, const double* t, const void* d_gray, const void* d_depth, gf_feature_obs* out, int cap,
                                  int* n_out) {
    if (!h || !t || !d_gray || !out || !n_out) return gf::set_err(GF_ERR_INVALID, "null argument");
    return gf::track_core(h, t, (const uint8_t*)d_gray, (const uint16_t*)d_depth, out, cap, n_out);
}

int gf_tracker_track_batch(gf_tracker* h, const double* t, const uint8_t* const* gray, int stride, const uint16_t* const* depth, int dstride,
                           gf_feature_obs* out, int cap, int* n_out) {
    if (!h || !t || !gray || !out || !n_out) return gf::set_err(GF_ERR_INVALID, "null argument");
    const int W = h->cfg.width, H = h->cfg.height;
    bool have_depth = depth != nullptr;
    for (int b = 0; b < h->B; b++) {
        if (!gray[b]) return gf::set_err(GF_ERR_INVALID, "null image for sequence %d", b);
        HIPCHK(hipMemcpy2DAsync(h->d_raw.p + (size_t)b * W * H, W, gray[b], stride, W, H, hipMemcpyHostToDevice, h->stream));
        if (have_depth && !depth[b]) have_depth = false;
    }
    // the depth image stays where it is: its <= max_cnt samples are taken on the host (track_core)
    return gf::track_core(h, t, h->d_raw.p, nullptr, out, cap, n_out, have_depth ? depth : nullptr, dstride);
}

// The host-image boundary (trackImage(const cv::Mat&), feature_tracker.h:47) without serialising on the bus: the images of frame k + 1 go to the second pair of
// frame buffers on a copy stream while frame k's kernels run; gf_tracker_track_prefetched then only waits for that copy.  The host images must stay valid until
// the matching gf_tracker_track_prefetched returns, and only page-locked memory (gf_host_alloc / hipHostRegister) makes the copy asynchronous.
int gf_tracker_prefetch_batch(gf_tracker* h, const uint8_t* const* gray, int stride, const uint16_t* const* depth, int dstride) {
    if (!h || !gray) return gf::set_err(GF_ERR_INVALID, "null argument");
    const int W = h->cfg.width, H = h->cfg.height;
    if (!h->copy_stream) {
        HIPCHK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        for (auto& e : h->ev_copy) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        if (int rc = h->d_raw2.alloc((size_t)h->B * W * H)) return rc;
    }
    if (h->pf_count >= 2) return gf::set_err(GF_ERR_CAPACITY, "two frames are staged already: gf_tracker_track_prefetched has to consume one first");
    const int slot = (h->pf_head + h->pf_count) & 1;      // a pair no frame in flight uses: track calls return when their frame is done
    uint8_t* raw = slot ? h->d_raw2.p : h->d_raw.p;
    bool have_depth = depth != nullptr;
    for (int b = 0; b < h->B; b++) {
        if (!gray[b]) return gf::set_err(GF_ERR_INVALID, "null image for sequence %d", b);
        if (have_depth && !depth[b]) have_depth = false;
    }
    // images that sit back to back in one allocation (a pinned ring of frames) go as ONE copy per plane: 256 separate 2-D copies cost more host time than the bus needs
    bool contig = stride == W;
    for (int b = 1; b < h->B && contig; b++) contig = gray[b] == gray[0] + (size_t)b * W * H;
    if (contig) HIPCHK(hipMemcpyAsync(raw, gray[0], (size_t)h->B * W * H, hipMemcpyHostToDevice, h->copy_stream));
    else for (int b = 0; b < h->B; b++) HIPCHK(hipMemcpy2DAsync(raw + (size_t)b * W * H, W, gray[b], stride, W, H, hipMemcpyHostToDevice, h->copy_stream));
    HIPCHK(hipEventRecord(h->ev_copy[slot], h->copy_stream));
    // the depth images do not travel: gf_tracker_track_prefetched samples them on the host (they must stay valid until it returns, like the gray images until the copy is done)
    h->pf_depth[slot] = have_depth; if (have_depth) h->pf_hdepth[slot].assign(depth, depth + h->B); else h->pf_hdepth[slot].clear();
    h->pf_hdstride[slot] = dstride;
    h->pf_count++;
    return GF_OK;
}

int gf_tracker_track_prefetched(gf_tracker* h, const double* t, gf_feature_obs* out, int cap, int* n_out) {
    if (!h || !t || !out || !n_out) return gf::set_err(GF_ERR_INVALID, "null argument");
    if (!h->pf_count) return gf::set_err(GF_ERR_INVALID, "gf_tracker_track_prefetched without a staged frame (call gf_tracker_prefetch_batch first)");
    const int slot = h->pf_head;
    HIPCHK(hipStreamWaitEvent(h->stream, h->ev_copy[slot], 0));
    h->pf_head ^= 1; h->pf_count--;
    return gf::track_core(h, t, slot ? h->d_raw2.p : h->d_raw.p, nullptr, out, cap, n_out, h->pf_depth[slot] ? h->pf_hdepth[slot].data() : nullptr, h->pf_hdstride[slot]);
}

// page-locked host memory for frames that are handed to gf_tracker_prefetch_batch / gf_tracker_track_batch (pageable memory makes hipMemcpyAsync synchronous)
int gf_host_alloc(size_t bytes, void** out) {
    if (!out || !bytes) return gf::set_err(GF_ERR_INVALID, "bad argument");
    if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) return gf::set_err(GF_ERR_HIP, "hipHostMalloc(%zu) failed", bytes);
    return GF_OK;
}
int gf_host_free(void* p) { if (p) (void)hipHostFree(p); return GF_OK; }

int gf_tracker_track(gf_tracker* h, int seq, double t, const uint8_t* gray, int stride, const uint16_t* depth, int dstride, gf_feature_obs* out,
                     int cap, int* n_out) {
    if (!h) return gf::set_err(GF_ERR_INVALID, "null handle");
    if (h->B != 1 || seq != 0) return gf::set_err(GF_ERR_INVALID, "gf_tracker_track drives a batch-1 handle (sequences of a batch advance in lock-step: use gf_tracker_track_batch)");
    const uint8_t* g[1] = {gray};
    const uint16_t* d[1] = {depth};
    return gf_tracker_track_batch(h, &t, g, stride, depth ? d : nullptr, dstride, out, cap, n_out);
}

int gf_tracker_set_prediction(gf_tracker* h, int seq, const int* ids, const double* xyz, int n) {
    if (!h || seq < 0 || seq >= h->B || (n > 0 && (!ids || !xyz))) return gf::set_err(GF_ERR_INVALID, "bad argument");
    gf::SeqState& s = h->seq[seq];
    s.hasPrediction = true;
    s.predict_pts.clear();
    std::map<int, const double*> m;
    for (int i = 0; i < n; i++) m[ids[i]] = xyz + 3 * i;
    for (size_t i = 0; i < s.ids.size(); i++) {
        auto it = m.find(s.ids[i]);
        if (it != m.end()) { double u, v; gf::space_to_plane(h->cfg, it->second, u, v); s.predict_pts.push_back({(float)u, (float)v}); }
        else s.predict_pts.push_back(s.prev_pts[i]);
    }
    return GF_OK;
}

int gf_tracker_remove_outliers(gf_tracker* h, int seq, const int* ids, int n) {
    if (!h || seq < 0 || seq >= h->B || (n > 0 && !ids)) return gf::set_err(GF_ERR_INVALID, "bad argument");
    gf::SeqState& s = h->seq[seq];
    std::set<int> rm(ids, ids + n);
    std::vector<uint8_t> st;
    for (size_t i = 0; i < s.ids.size(); i++) st.push_back(rm.count(s.ids[i]) ? 0 : 1);
    gf::reduce_vector(s.prev_pts, st.data()); gf::reduce_vector(s.ids, st.data()); gf::reduce_vector(s.track_cnt, st.data());
    return GF_OK;
}

int gf_tracker_get_state(gf_tracker* h, int seq, int* ids, int* track_cnt, float* prev_pts_xy, int cap, int* n) {
    if (!h || seq < 0 || seq >= h->B || !n) return gf::set_err(GF_ERR_INVALID, "bad argument");
    gf::SeqState& s = h->seq[seq];
    *n = (int)s.ids.size();
    if (*n > cap) return gf::set_err(GF_ERR_CAPACITY, "capacity %d < %d", cap, *n);
    for (int i = 0; i < *n; i++) {
        if (ids) ids[i] = s.ids[i];
        if (track_cnt) track_cnt[i] = s.track_cnt[i];
        if (prev_pts_xy) { prev_pts_xy[2 * i] = s.prev_pts[i].x; prev_pts_xy[2 * i + 1] = s.prev_pts[i].y; }
    }
    return GF_OK;
}

int gf_tracker_set_profiling(gf_tracker* h, int enable) { if (!h) return gf::set_err(GF_ERR_INVALID, "null handle"); h->profiling = enable != 0; return GF_OK; }
int gf_tracker_get_stats(gf_tracker* h, gf_tracker_stats* out) { if (!h || !out) return gf::set_err(GF_ERR_INVALID, "null argument"); *out = h->stats; return GF_OK; }
int gf_tracker_reset_stats(gf_tracker* h) { if (!h) return gf::set_err(GF_ERR_INVALID, "null handle"); h->stats = gf_tracker_stats{}; return GF_OK; }

// ------------------------------------------------------------------ building blocks for parity tests
static int tmp_handle(int width, int height, int max_cnt, int min_dist, gf_tracker** h) {
    gf_tracker_cfg c{};
    c.width = width; c.height = height; c.batch = 1; c.max_cnt = max_cnt; c.min_dist = min_dist; c.flow_back = 0; c.depth_cam = 0;
    c.fx = c.fy = 1; c.cx = c.cy = 0;
    return gf_tracker_create(&c, h);
}

int gf_lk_track(const uint8_t* prev, const uint8_t* next, int width, int height, const float* prev_pts, float* next_pts, uint8_t* status, int n,
                int max_level, int use_initial_flow, long long* iterations) {
    if (!prev || !next || !prev_pts || !next_pts || !status || n < 0) return gf::set_err(GF_ERR_INVALID, "bad argument");
    if (n == 0) return GF_OK;
    gf_tracker* h = nullptr;
    if (int rc = tmp_handle(width, height, n, 30, &h)) return rc;
    int rc = GF_OK;
    auto body = [&]() -> int {
        const size_t px = (size_t)width * height;
        HIPCHK(hipMemcpyAsync(h->d_raw.p, prev, px, hipMemcpyHostToDevice, h->stream));
        h->cur_slot = 0;
        if (int r = gf::launch_pyramid(h, h->d_raw.p)) return r;
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpyAsync(h->d_raw.p, next, px, hipMemcpyHostToDevice, h->stream));
        h->cur_slot = 1;
        if (int r = gf::launch_pyramid(h, h->d_raw.p)) return r;
        h->h_npts.p[0] = n;
        for (int i = 0; i < n; i++) {
            h->h_prev_pts.p[i] = make_float2(prev_pts[2 * i], prev_pts[2 * i + 1]);
            h->h_init_pts.p[i] = make_float2(next_pts[2 * i], next_pts[2 * i + 1]);
        }
        HIPCHK(hipMemcpyAsync(h->d_npts.p, h->h_npts.p, sizeof(int), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_prev_pts.p, h->h_prev_pts.p, (size_t)h->cap * sizeof(float2), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_init_pts.p, h->h_init_pts.p, (size_t)h->cap * sizeof(float2), hipMemcpyHostToDevice, h->stream));
        gf::launch_lk(h, 1, gf::lk_args(h, max_level, use_initial_flow ? 1 : 0, 0, 0, nullptr, nullptr));
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h->h_cur_pts.p, h->d_cur_pts.p, (size_t)h->cap * sizeof(float2), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(h->h_status.p, h->d_status.p, (size_t)h->cap, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(h->h_counters.p, h->d_counters.p, (size_t)h->cap * 2 * sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        long long it = 0;
        for (int i = 0; i < n; i++) {
            next_pts[2 * i] = h->h_cur_pts.p[i].x; next_pts[2 * i + 1] = h->h_cur_pts.p[i].y;
            status[i] = h->h_status.p[i];
            it += h->h_counters.p[2 * i + 1];
        }
        if (iterations) *iterations = it;
        return GF_OK;
    };
    rc = body();
    gf_tracker_destroy(h);
    return rc;
}

int gf_pyramid_level(const uint8_t* img, int width, int height, int level, uint8_t* out, int16_t* deriv_xy) {
    if (!img || level < 0) return gf::set_err(GF_ERR_INVALID, "bad argument");
    gf_tracker* h = nullptr;
    if (int rc = tmp_handle(width, height, 4, 30, &h)) return rc;
    auto body = [&]() -> int {
        if (level >= h->G.nlevels) return gf::set_err(GF_ERR_INVALID, "level %d not built (pyramid has %d levels)", level, h->G.nlevels);
        HIPCHK(hipMemcpyAsync(h->d_raw.p, img, (size_t)width * height, hipMemcpyHostToDevice, h->stream));
        h->cur_slot = 0;
        if (int r = gf::launch_pyramid(h, h->d_raw.p)) return r;
        HIPCHK(hipStreamSynchronize(h->stream));
        const gf::LevelGeom g = h->G.lv[level];
        if (out) HIPCHK(hipMemcpy2D(out, g.w, h->d_img.p + g.img_off, g.stride, g.w, g.h, hipMemcpyDeviceToHost));
        if (deriv_xy) {   // the derivative as lk_solve evaluates it (deriv_probe_kernel calls the same device functions)
            gf::DevBuf<int> d;
            if (int r = d.alloc((size_t)g.w * g.h)) return r;
            gf::deriv_probe_kernel<<<dim3((((g.w + 7) >> 3) * g.h + 255) / 256), 256, 0, h->stream>>>(h->d_img.p, g, d.p);
            hipError_t e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            if (e == hipSuccess) e = hipMemcpy(deriv_xy, d.p, (size_t)g.w * g.h * 4, hipMemcpyDeviceToHost);
            d.release();
            HIPCHK(e);
        }
        return GF_OK;
    };
    int rc = body();
    gf_tracker_destroy(h);
    return rc;
}

int gf_min_eigen_val(const uint8_t* img, int width, int height, float* eig) {
    if (!img || !eig) return gf::set_err(GF_ERR_INVALID, "bad argument");
    gf_tracker* h = nullptr;
    if (int rc = tmp_handle(width, height, 4, 30, &h)) return rc;
    auto body = [&]() -> int {
        HIPCHK(hipMemcpyAsync(h->d_raw.p, img, (size_t)width * height, hipMemcpyHostToDevice, h->stream));
        h->cur_slot = 0;
        if (int r = gf::launch_pyramid(h, h->d_raw.p)) return r;
        gf::min_eig_kernel<<<dim3((width + 31) / 32, (height + 7) / 8, 1), 256, 0, h->stream>>>(h->d_img.p, 2 * h->G.img_bytes, h->G.lv[0], h->d_eig.p, h->eig_stride);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(eig, h->d_eig.p, (size_t)width * height * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        return GF_OK;
    };
    int rc = body();
    gf_tracker_destroy(h);
    return rc;
}

int gf_good_features(const uint8_t* img, int width, int height, const uint8_t* mask, int max_corners, int min_dist, float* corners_xy, int* n_out) {
    if (!img || !corners_xy || !n_out || max_corners < 1) return gf::set_err(GF_ERR_INVALID, "bad argument");
    gf_tracker* h = nullptr;
    if (int rc = tmp_handle(width, height, max_corners, min_dist, &h)) return rc;
    auto body = [&]() -> int {
        const int W = width, H = height;
        HIPCHK(hipMemcpyAsync(h->d_raw.p, img, (size_t)W * H, hipMemcpyHostToDevice, h->stream));
        h->cur_slot = 0;
        if (int r = gf::launch_pyramid(h, h->d_raw.p)) return r;
        if (mask) HIPCHK(hipMemcpyAsync(h->d_mask.p, mask, (size_t)W * H, hipMemcpyHostToDevice, h->stream));
        else HIPCHK(hipMemsetAsync(h->d_mask.p, 255, (size_t)W * H, h->stream));
        h->h_want.p[0] = max_corners;
        HIPCHK(hipMemcpyAsync(h->d_want.p, h->h_want.p, sizeof(int), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemsetAsync(h->d_maxkey.p, 0, sizeof(unsigned), h->stream));
        HIPCHK(hipMemsetAsync(h->d_cand_count.p, 0, sizeof(int), h->stream));
        {
            gf::DetectArgs D{};
            D.pyr = h->d_img.p; D.pyr_seq_stride = 2 * h->G.img_bytes; D.g = h->G.lv[0];
            D.mask = h->d_mask.p; D.mask_seq_stride = h->mask_stride; D.centers = nullptr; D.n_centers = nullptr; D.cap = h->cap; D.want = h->d_want.p;
            D.maxkey = h->d_maxkey.p; D.cand = h->d_cand.p; D.cand_seq_stride = (size_t)h->cand_cap; D.cand_cap = h->cand_cap; D.cand_count = h->d_cand_count.p;
            gf::detect_strip_kernel<gf::kDS_R><<<dim3((W + gf::kDS_W - 1) / gf::kDS_W, (H + gf::kDS_R - 1) / gf::kDS_R, 1), 64, 0, h->stream>>>(D, h->disk);
        }
        gf::SelectArgs S{};
        S.cand = h->d_cand.p; S.cand_seq_stride = (size_t)h->cand_cap; S.cand_cap = h->cand_cap; S.cand_count = h->d_cand_count.p; S.maxkey = h->d_maxkey.p; S.want = h->d_want.p;
        S.w = W; S.h = H; S.min_dist = min_dist; S.out_cap = h->cap; S.sort_cap = h->sort_cap; S.out_pts = h->d_out_pts.p; S.out_depth = h->d_out_depth.p; S.out_n = h->d_out_n.p;
        gf::select_corners_kernel<<<dim3(1), 1024, h->select_lds, h->stream>>>(S);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h->h_out_n.p, h->d_out_n.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(h->h_out_pts.p, h->d_out_pts.p, (size_t)h->cap * sizeof(float2), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        *n_out = h->h_out_n.p[0];
        for (int i = 0; i < *n_out; i++) { corners_xy[2 * i] = h->h_out_pts.p[i].x; corners_xy[2 * i + 1] = h->h_out_pts.p[i].y; }
        return GF_OK;
    };
    int rc = body();
    gf_tracker_destroy(h);
    return rc;
}


// ---- calibration of the rocprofv3 FETCH_SIZE counter for the front end's access patterns (profiling aid, scripts/pmc_collect.sh).
// Known byte counts over a buffer far larger than the Infinity Cache, three patterns:
//   mode 0  streaming: every lane reads 16 contiguous bytes (the pattern MI355X_MICROARCH.md calibrates: FETCH_SIZE = 1/2 of the bytes)
//   mode 1  LK tile, aligned: a wavefront reads a 32 x 32 u8 tile, two 16-byte lanes per row (the refill of lk_solve), every 32-byte row segment
//           inside its own 64-byte line, tiles disjoint
//   mode 2  LK tile at an odd 4-byte phase: every row segment straddles two 64-byte lines
namespace gf {
__global__ void __launch_bounds__(256) calib_stream_kernel(const uint4* __restrict__ src, size_t n16, unsigned* __restrict__ sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void __launch_bounds__(256) calib_tile_kernel(const uint8_t* __restrict__ src, size_t row_stride, int tiles_per_row, size_t n_tiles, int phase, unsigned* __restrict__ sink) {
    const size_t t = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n_tiles) return;
    const int lane = threadIdx.x & 63, trow = lane >> 1, thalf = lane & 1;
    const size_t ty = t / tiles_per_row, tx = t % tiles_per_row;
    const uint8_t* p = src + (ty * 32 + trow) * row_stride + tx * 128 + phase + thalf * 16;
    const U4a v = *reinterpret_cast<const U4a*>(p);
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) sink[0] = v.x;
}
}  // namespace gf
int gf_calib_fetch(int mode, size_t buffer_bytes, double* requested_bytes, double* lines64, double* ms) {
    if (mode < 0 || mode > 2 || buffer_bytes < (1u << 20)) return gf::set_err(GF_ERR_INVALID, "bad argument");
    uint8_t* buf = nullptr; unsigned* sink = nullptr;
    if (hipMalloc((void**)&buf, buffer_bytes) != hipSuccess) return gf::set_err(GF_ERR_HIP, "hipMalloc failed");
    if (hipMalloc((void**)&sink, 64) != hipSuccess) { (void)hipFree(buf); return gf::set_err(GF_ERR_HIP, "hipMalloc failed"); }
    (void)hipMemset(buf, 1, buffer_bytes); (void)hipMemset(sink, 0, 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    if (mode == 0) {
        const size_t n16 = buffer_bytes / 16;
        gf::calib_stream_kernel<<<dim3(256 * 8), 256>>>(reinterpret_cast<const uint4*>(buf), n16, sink);
        if (requested_bytes) *requested_bytes = (double)n16 * 16; if (lines64) *lines64 = (double)n16 / 4;
    } else {
        const size_t row_stride = 1u << 16;                 // a 64 KiB wide "image": 512 tile columns of 128 bytes
        const int tiles_per_row = 511;
        const size_t tile_rows = buffer_bytes / row_stride / 32, n_tiles = tile_rows * tiles_per_row;
        const int phase = mode == 1 ? 0 : 44;                // 44: bytes 44..75 of the 128-byte slot -> both halves of the segment in different 64-byte lines
        gf::calib_tile_kernel<<<dim3((unsigned)((n_tiles + 3) / 4)), 256>>>(buf, row_stride, tiles_per_row, n_tiles, phase, sink);
        if (requested_bytes) *requested_bytes = (double)n_tiles * 1024; if (lines64) *lines64 = (double)n_tiles * 32 * (mode == 1 ? 1 : 2);
    }
    (void)hipEventRecord(e1, 0);
    const hipError_t err = hipDeviceSynchronize();
    float t = 0; (void)hipEventElapsedTime(&t, e0, e1);
    if (ms) *ms = t;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(buf); (void)hipFree(sink);
    if (err != hipSuccess) return gf::set_err(GF_ERR_HIP, "calibration kernel failed: %s", hipGetErrorString(err));
    return GF_OK;
}

}  // extern "C"
