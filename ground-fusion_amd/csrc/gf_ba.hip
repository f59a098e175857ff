// gf_ba.hip — C-ABI back end (include/groundfusion_hip.h): host orchestration of the HIP sliding-window solver.
//
// Replaces the Ceres problem of Estimator::optimization() (estimator.cpp:2890-3327) for a batch of independent windows:
// packs the para_* arrays and factor tables into device SoA buffers, runs a fixed schedule of dogleg iterations
// (gf_ba_kernels.hpp) without host round trips, and builds the next marginalisation prior (gf_ba_marg.hpp).
// No CPU fallback: every entry point needs a HIP device.
// The back end's parity bar is a tolerance (1e-6 on poses), not bit-exactness: unlike the front end (built with -ffp-contract=off for the exact LK
// arithmetic) this translation unit lets a*b+c contract into v_fma_f64 -- one instruction and one rounding instead of two.
#pragma clang fp contract(fast)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/groundfusion_hip.h"
#include "gf_comm.hpp"
#include "gf_ba_kernels.hpp"
#include "gf_copy_list.hpp"
#include "gf_ba_marg.hpp"
#include "gf_ba_gnss.hpp"

namespace gf { int set_err(int code, const char* fmt, ...); }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return gf::set_err(GF_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

using namespace gfb;

// GF_BA_POISON=4 (debugging aid, round 5): fresh device buffers start as PLAUSIBLE stale data -- what hipMalloc hands back behind another handle of the process:
// doubles of ordinary magnitude (zeros, +-1, values in +-10, a few small ones), ints that look like counts, indices and the -1 markers of the tables -- instead of
// zeros.  Garbage (0x5A...) turns a stray read into NaN or a crash; plausible data turns it into a slightly different result, which is what a deployment would see.
__global__ void ba_fill_plausible(void* p, size_t i0, size_t i1, int is_double, unsigned seed) {   // elements i0 <= i < i1 of the buffer at p; the content of element i does not depend on the range
    for (size_t i = i0 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < i1; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761ull) ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        // Freed memory does not respect element types: a double buffer inherits int tables (two -1 markers are the bit pattern of a NaN, two small counts a denormal)
        // and an int buffer the halves of doubles -- the read this mode was built to find (round 5) only showed behind an int table.
        // The first eight elements walk through all eight kinds, the foreign bit patterns first: a buffer of one or three elements (pri_c is [batch]) must not depend on
        // its hash to meet a NaN.
        if (is_double) {
            const int sel = i < 8 ? (int)((i + 4) & 7) : (int)(h & 7);
            const double mag = (double)((int)((h >> 8) % 20001u) - 10000) * 1e-3;
            double v = sel == 0 ? 0.0 : sel == 1 ? 1.0 : sel == 2 ? -1.0 : sel == 3 ? mag * 1e-4 : mag;
            if (sel == 4) v = __longlong_as_double(-1LL);                                                                  // (-1, -1)
            if (sel == 5) v = __longlong_as_double((long long)((h >> 8) % 200u) << 32 | (long long)((h >> 16) % 200u));     // two small ints
            static_cast<double*>(p)[i] = v;
        } else {
            const int sel = i < 8 ? (int)i : (int)(h & 7);
            static_cast<int*>(p)[i] = sel == 0 ? -1 : sel == 1 ? (int)(0x3FE00000u + ((h >> 8) & 0xFFFFFu)) : sel == 2 ? (int)(h * 2654435761u) : (int)((h >> 8) % 200u);   // markers, high / low words of doubles, counts
        }
    }
}

namespace {
std::atomic<int> g_alloc_idx{0}, g_alloc_tix{0}, g_alloc_seq_ctr{0};
thread_local const char* g_alloc_what = nullptr;   // the allocation statement being executed (GF_BA_ALLOC_TRACE=1 prints it next to the allocation's index: what GF_BA_POISON_RANGE counts)
std::atomic<long long> g_up_bytes{0}, g_up_calls{0};   // host -> device traffic of this process (GF_GROUP_TIMING prints it)
template <class T> struct Buf {  // device buffer + pinned host mirror
    T* d = nullptr; T* h = nullptr; size_t n = 0;
    T* hd = nullptr;   // the host mirror as kernels address it (page-locked memory is mapped into the device's address space), or null
    int alloc(size_t count, bool host) {
        n = count;
        const int g_alloc_seq = g_alloc_seq_ctr++;
        if (hipMalloc((void**)&d, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return gf::set_err(GF_ERR_HIP, "hipMalloc(%zu B) failed", count * sizeof(T));
        // GF_BA_POISON=1 (debugging aid): fresh device buffers start as 0x5A bytes (2.5e130 as a double -- finite, so that garbage x 0 stays 0 --, 1 515 870 810 as an int) instead of whatever the allocator hands back, so that a kernel reading
        // what nobody wrote shows as NaN in the results instead of as a dependence on the process's history
        static const int pmode = getenv("GF_BA_POISON") ? atoi(getenv("GF_BA_POISON")) : 0;   // 1: device buffers and LDS, 2: device buffers only, 3: LDS only, 4: device buffers with plausible stale data, 5: device buffers full of 0xFF (every double a NaN, every int -1: a read "masked" by a multiplication with zero shows)
        static const bool poison = pmode != 0 && pmode != 3;
        // Every device buffer starts as zeros, explicitly: parts of them are read before anything of THIS handle wrote them (the GNSS cost part of a handle without
        // GNSS factors, rows beyond what a batch fills, ...) and hipMalloc hands back whatever the previous owner left -- zeros in a fresh process, another handle's
        // tables later in the same process (round 4: a group member and a stand-alone estimator disagreed once an earlier test had used the memory).
        bool bad = poison;
        if (const char* r = getenv("GF_BA_POISON_RANGE")) {   // "lo:hi": only the allocations lo <= index < hi of the process (to find which buffer a kernel reads unwritten)
            // one counter for every element type (a static inside this template would count doubles and ints separately, as it did in round 4)
            const int i = g_alloc_idx++, lo = atoi(r), hi = strchr(r, ':') ? atoi(strchr(r, ':') + 1) : lo + 1;
            bad = i >= lo && i < hi;
        }
        // (hipMemset runs on the null stream and the handle's stream is non-blocking: finished here, before anybody can enqueue a copy into the buffer)
        if (hipMemset(d, (bad && pmode != 4 && pmode != 5) ? 0x5A : 0, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) return gf::set_err(GF_ERR_HIP, "hipMemset failed");
        if (bad && pmode == 5 && count > 0) {   // 0xFF over the elements GF_BA_POISON_ELEMS names (default: all)
            size_t lo = 0, hi = count;
            if (const char* e = getenv("GF_BA_POISON_ELEMS")) { lo = std::min<size_t>(count, strtoull(e, nullptr, 10)); hi = strchr(e, ':') ? std::min<size_t>(count, strtoull(strchr(e, ':') + 1, nullptr, 10)) : count; }
            if (hi > lo && (hipMemset(reinterpret_cast<char*>(d) + lo * sizeof(T), 0xFF, (hi - lo) * sizeof(T)) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess)) return gf::set_err(GF_ERR_HIP, "hipMemset failed");
        }
        if (bad && pmode == 4 && count > 0 && (std::is_same<T, double>::value || std::is_same<T, int>::value)) {
            const unsigned seed = 12345u + 7919u * (unsigned)g_alloc_seq;   // a function of the allocation's index in the process: the same content whichever other allocations are poisoned (the bisection relies on it)
            size_t lo = 0, hi = count;   // GF_BA_POISON_ELEMS="lo:hi": only these elements of the selected allocation(s) (scripts/stale_bisect.py narrows a dependence down to an element)
            if (const char* e = getenv("GF_BA_POISON_ELEMS")) { lo = std::min<size_t>(count, strtoull(e, nullptr, 10)); hi = strchr(e, ':') ? std::min<size_t>(count, strtoull(strchr(e, ':') + 1, nullptr, 10)) : count; }
            if (hi > lo) ba_fill_plausible<<<dim3(256), 256, 0, nullptr>>>(d, lo, hi, std::is_same<T, double>::value ? 1 : 0, seed);
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) return gf::set_err(GF_ERR_HIP, "plausible fill failed");
        }
        if (getenv("GF_BA_ALLOC_TRACE")) {   // index, statement, size, and the buffer's first and last 8 bytes as they are now (what a poison mode really left there)
            unsigned long long first = 0, last = 0; const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
            if (bytes >= 8) { (void)hipMemcpy(&first, d, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&last, reinterpret_cast<char*>(d) + ((bytes - 8) & ~(size_t)7), 8, hipMemcpyDeviceToHost); }
            fprintf(stderr, "gf_ba alloc %d: %s  %zu x %zu B%s  first %016llx last %016llx\n", g_alloc_tix++, g_alloc_what ? g_alloc_what : "?", count, sizeof(T), bad ? "  [poisoned]" : "", first, last);
        }
        if (host) { if (hipHostMalloc((void**)&h, std::max<size_t>(count, 1) * sizeof(T), hipHostMallocDefault) != hipSuccess) return gf::set_err(GF_ERR_HIP, "hipHostMalloc failed"); memset(h, 0, std::max<size_t>(count, 1) * sizeof(T));
            void* p = nullptr; hd = hipHostGetDevicePointer(&p, h, 0) == hipSuccess ? static_cast<T*>(p) : nullptr; (void)hipGetLastError(); }
        return GF_OK;
    }
    void release() { if (d) (void)hipFree(d); if (h) (void)hipHostFree(h); d = nullptr; h = nullptr; }
    hipError_t up(hipStream_t s) { g_up_bytes += (long long)(n * sizeof(T)); g_up_calls++; return hipMemcpyAsync(d, h, n * sizeof(T), hipMemcpyHostToDevice, s); }
    // `rows` rows of `pitch` elements, only the first `used` of each are live: one strided copy of what the kernels read
    hipError_t up2d(hipStream_t s, size_t rows, size_t pitch, size_t used) {
        if (used == 0 || rows == 0) return hipSuccess;
        g_up_bytes += (long long)(rows * std::min(used, pitch) * sizeof(T)); g_up_calls++;
        if (used >= pitch) return hipMemcpyAsync(d, h, rows * pitch * sizeof(T), hipMemcpyHostToDevice, s);
        return hipMemcpy2DAsync(d, pitch * sizeof(T), h, pitch * sizeof(T), used * sizeof(T), rows, hipMemcpyHostToDevice, s);
    }
    hipError_t down(hipStream_t s) { return hipMemcpyAsync(h, d, n * sizeof(T), hipMemcpyDeviceToHost, s); }
};
}  // namespace

// per slot: rows 1 .. W - 1 move up by one when the window slid (flag), then the staged rows go to their positions
__global__ void ba_imu_patch(double* tab, const double* patch, const int* pinfo, int W) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int* pi = pinfo + (size_t)b * (W + 2);
    double* t = tab + (size_t)b * W * IMU_STRIDE2;
    if (pi[0]) for (int k = 0; k + 1 < W; k++) {
        for (int i = tid; i < IMU_STRIDE2; i += 256) t[(size_t)k * IMU_STRIDE2 + i] = t[(size_t)(k + 1) * IMU_STRIDE2 + i];
        __syncthreads();
    }
    const double* p = patch + (size_t)b * W * IMU_STRIDE2;
    for (int q = 0; q < pi[1]; q++)
        for (int i = tid; i < IMU_STRIDE2; i += 256) t[(size_t)pi[2 + q] * IMU_STRIDE2 + i] = p[(size_t)q * IMU_STRIDE2 + i];
}

// ---- the upload as ONE kernel (GF_BA_UPLOAD=kernel, the default; gf_copy_list.hpp): every table of a batch is a descriptor, the kernel reads the page-locked host
// mirrors over the bus and writes the device tables
struct UpBuilder {
    gfcopy::Builder B;
    template <class T> void add(const Buf<T>& b, size_t rows, size_t pitch, size_t used) {
        if (used == 0 || rows == 0) return;
        const long long before = B.bytes;
        B.add(b.hd, b.d, rows, pitch * sizeof(T), std::min(used, pitch) * sizeof(T));
        g_up_bytes += B.bytes - before;
    }
    template <class T> void all(const Buf<T>& b) { add(b, 1, b.n, b.n); }
};

struct gf_ba {
    gf_ba_cfg cfg;
    Dims d;
    int count = 0;       // windows currently resident
    bool any_ex = false; // some window estimates the camera extrinsic
    hipStream_t stream = nullptr;
    hipEvent_t ev[10] = {};   // 6, 7: around the second ba_step of a solve (the first full dogleg step); 8, 9: around the upload of gf_ba_solve_packed
    bool packed_upload_timed = false;
    bool pending = false;   // an asynchronous solve is in flight
    int pending_iters = 0;
    gf_ba_stats stats{};
    size_t step_lds = 0, sg_stride = 0, mg_stride = 0; bool big_step = false, big_marg = false;
    // inputs (host mirror + device)
    Buf<double> xs0;     // pristine states [B][XS] (for reset)
    Buf<double> xs;      // [2][B][XS]
    Buf<int> colf, cole, nvis, nimu, nwh, nfeat, vis_idx, order, norder, feat_ptr, vis_pos, imu_i, wh_i, pri_n, pri_nb, pri_bid;
    Buf<double> vis_data, feat_obs, imu_data, wh_data, pri_J, pri_r, pri_x0;
    Buf<SolverState> st, st0;
    // work
    Buf<double> imu_sqrt, wh_sqrt, pri_A, pri_b, pri_c, H, g, Vc, vtile, wpar, cost, efac;
    Buf<double> gather_send;   // [B][7] newest poses, send buffer of gf_pose_gather (allocated on first use)
    hipEvent_t ev_gather = nullptr, ev_gather_done = nullptr; bool gather_in_flight = false;   // export -> collective, and collective -> next export (the send buffer is reused)
    Buf<double> scale, diag, grad, gn, step, u, Et, Es, ete, etb, rhs, yv, Sg, Mg, gn_data, gn_misc;
    Buf<int> ngnss, gn_idx, gn_gptr, gn_gitem;
    Buf<double> gn_rows;
    // marginalisation: column maps per mode (0 MARGIN_OLD, 1 MARGIN_SECOND_NEW), outputs
    Buf<int> mcolf[2], mcole[2], morder[2], mnorder[2], minfo[2], minfo_stage;
    Buf<double> outJ, outr;
    Buf<long long> stamps;
    std::vector<std::vector<int>> keep_ids[2];   // per window: kept block ids (before the address shift), in column order
    size_t marg_lds = 0; int marg_ncap = 0, last_marg_mode = -1;
    size_t vwin_lds = 0;   // dynamic LDS of the visual sweep (its pair tiles); 0: the tiles live in global memory (vtile)
    size_t vwinx_lds = 0;  // the same for the variant with camera-extrinsic columns (free extrinsic)
    size_t vwinm_lds = 0;  // dynamic LDS of the MARGIN_OLD sweep (NP - 1 pair tiles + continuation slots: always fits)
    size_t vtile_stride = 0;   // doubles per window in vtile (0: both variants keep their tiles in LDS)
    size_t mwin_lds = 0;   // dynamic LDS of the prior / IMU / wheel sweep (ba_linearize_misc_win)
    bool fuse_misc = false, can_fuse_misc = false;
    int max_vis = 0, max_order = 0, max_prior = 0, max_feat = 0;   // largest n_visual / factor-order length / prior size of the resident batch (what the uploads copy)
    std::vector<const gf_ba_window*> resident;       // the caller's window behind every resident slot (gf_ba_marginalize_resident)
    // what packing a window into slot b found out about it; reduced over the batch when the batch is closed (pack_slot may run on one thread per slot)
    struct SlotMeta { int mno[2] = {0, 0}; int R = 0; bool chain = false; bool any_ex = false, pri_res = false, pos_ident = true; long long mfma = 0, step = 0, jtj = 0; int nvis = 0, norder = 0, npri = 0, nfeat = 0, imu_dirty = 0; };
    std::vector<SlotMeta> meta;
    // device-resident priors (gf_ba_pack_slot with prior_n > 0 and prior_J == NULL): outJ_n[b] = size of the prior the last gf_ba_marginalize_resident left in
    // the output buffer of slot b (0: none); the next solve of that slot copies it device to device into the prior table instead of taking it from the host
    std::vector<int> outJ_n, active;
    // IMU tables that stay on the device: a window's pre-integrations change little from frame to frame (a new interval at the end; after a MARGIN_OLD slide
    // everything moves up by one).  pack_slot compares the rows it is handed with what the slot's table holds (bytes), finds "same position" or "moved up by one",
    // and stages only the rows that are new; the upload sends those (imu_patch) and a small kernel moves / patches the slot's table.  imu_dev[b]: the device table
    // of slot b equals its host mirror (set by an upload that covered the slot).
    Buf<double> imu_patch; Buf<int> imu_pinfo;   // [B][W][IMU_STRIDE2] rows to write, [B][W + 2]: moved-up flag, number of rows, their positions
    std::vector<char> imu_dev;   // 0: device table unknown / behind the mirror, 1: device == mirror, 2: packed for the patch path (mirror ahead by the staged rows until upload() succeeds)
    std::vector<int> imu_rows;   // rows of the slot's mirror that the device table holds as well (the previous pack's n_imu)
    int max_imu_dirty = 0;
    bool all_pri_res = false, any_pri_res = false, outJ_host_stale = false;
    int step_waves = 8;
    // chain form of ba_step (round 6; gf_ba_kernels.hpp, note in front of ba_step_body): GF_BA_CHAIN=1 or gf_ba_set_chain().  One setting per process for the same reason as
    // step_waves: the two forms pivot in different orders, and a window alone must run the form it runs inside a batch.
    bool chain_mode = false; int n_chain = 0, n_dense = 0, chain_nd = 0; size_t yg_stride = 0; Buf<double> Yg;
    bool cost_only = true;   // the last iteration's candidate is linearised cost-only (GF_BA_COST_ONLY=0 when the handle is created: in full, as every other candidate)
    bool upload_kernel = true;
    bool pos_ident = false;
    bool split_jtj = false, split_timed = false;   // gf_ba_set_split_jtj: the visual sweep as two kernels (block rows through HBM, contraction-only MFMA kernel)
    Buf<double> vrows, vpair; hipEvent_t ev_split[2] = {nullptr, nullptr};
    double max_solver_time = 0.0;   // ceres::Solver::Options::max_solver_time_in_seconds; 0 = not honoured (the fixed schedule runs without host round trips)
    long long mfma_per_lin = 0;   // v_mfma_f64_16x16x4 instructions of one visual linearisation of the resident batch
    long long jtj_alg_flops = 0;  // algorithmic flops of the same: Nv * 2 * 2 * (12 * 13 / 2 + 12 + 1) per window
    long long step_flops = 0;     // dense algebra of one ba_step over the resident batch: Schur SYRK NE*n_c^2 + Cholesky R^3/3 + substitutions 2 R^2
    std::vector<Buf<double>*> dbl() { return {&xs0, &xs, &vis_data, &feat_obs, &imu_data, &wh_data, &pri_J, &pri_r, &pri_x0, &imu_sqrt, &wh_sqrt, &pri_A, &pri_b, &pri_c, &H, &g, &Vc, &vtile, &wpar, &cost, &efac,
                                              &scale, &diag, &grad, &gn, &step, &u, &Et, &Es, &ete, &etb, &rhs, &yv, &Sg, &Mg, &Yg, &gn_data, &gn_misc, &gn_rows}; }
    std::vector<Buf<int>*> ints() { return {&colf, &cole, &nvis, &nimu, &nwh, &nfeat, &vis_idx, &order, &norder, &feat_ptr, &vis_pos, &imu_i, &wh_i, &pri_n, &pri_nb, &pri_bid, &ngnss, &gn_idx, &gn_gptr, &gn_gitem}; }
    void release() {
        for (auto* b : dbl()) b->release();
        for (auto* b : ints()) b->release();
        for (int m = 0; m < 2; m++) { mcolf[m].release(); mcole[m].release(); morder[m].release(); mnorder[m].release(); minfo[m].release(); }
        minfo_stage.release(); imu_patch.release(); imu_pinfo.release();
        outJ.release(); outr.release(); stamps.release();
        st.release(); st0.release();
        for (auto& e : ev) if (e) (void)hipEventDestroy(e);
        if (ev_gather) (void)hipEventDestroy(ev_gather);
        if (ev_gather_done) (void)hipEventDestroy(ev_gather_done);
        for (auto& e : ev_split) if (e) (void)hipEventDestroy(e);
        vrows.release(); vpair.release();
        gather_send.release();
        if (stream) (void)hipStreamDestroy(stream);
    }
    Win win() {
        Win w{};
        w.d = d; w.xs = xs.d; w.colf = colf.d; w.cole = cole.d; w.nvis = nvis.d; w.nimu = nimu.d; w.nwh = nwh.d; w.nfeat = nfeat.d;
        w.vis_idx = vis_idx.d; w.vis_data = vis_data.d; w.feat_obs = feat_obs.d; w.order = order.d; w.norder = norder.d;
        w.ngnss = ngnss.d; w.gn_idx = gn_idx.d; w.gn_data = gn_data.d; w.gn_misc = gn_misc.d; w.gn_gptr = gn_gptr.d; w.gn_gitem = gn_gitem.d; w.gn_rows = gn_rows.d;
        w.feat_ptr = feat_ptr.d; w.vis_pos = vis_pos.d; w.pos_ident = pos_ident ? 1 : 0; w.imu_i = imu_i.d; w.imu_data = imu_data.d; w.wh_i = wh_i.d; w.wh_data = wh_data.d;
        w.imu_sqrt = imu_sqrt.d; w.wh_sqrt = wh_sqrt.d; w.pri_n = pri_n.d; w.pri_nb = pri_nb.d; w.pri_bid = pri_bid.d; w.pri_J = pri_J.d; w.pri_r = pri_r.d;
        w.pri_x0 = pri_x0.d; w.pri_A = pri_A.d; w.pri_b = pri_b.d; w.pri_c = pri_c.d; w.H = H.d; w.g = g.d; w.Vc = Vc.d; w.cost = cost.d; w.efac = efac.d; w.st = st.d;
        w.wpar = wpar.d; w.vtile = nullptr; w.vtile_stride = vtile_stride; w.stamps = stamps.d;
        return w;
    }
    StepBufs sbufs() {
        StepBufs s{};
        s.scale = scale.d; s.diag = diag.d; s.grad = grad.d; s.gn = gn.d; s.step = step.d; s.u = u.d; s.Et = Et.d; s.Es = Es.d; s.ete = ete.d; s.etb = etb.d;
        s.rhs = rhs.d; s.yv = yv.d; s.VS = d.RP + d.FP; s.stamps = stamps.d; s.Sg = Sg.d; s.SgStride = sg_stride; s.Mg = Mg.d; s.MgStride = mg_stride;
        s.Yg = Yg.d; s.YgStride = yg_stride;
        return s;
    }
};

namespace {

// One window into slot b of the pinned staging tables (everything the kernels read about it, both marginalisation layouts included).  Touches only
// slot b: any number of slots may be packed concurrently, one thread per slot.
int pack_slot(gf_ba* h, int b, const gf_ba_window& w) {
    const Dims& d = h->d;
    gf_ba::SlotMeta& M = h->meta[b];
    M = gf_ba::SlotMeta{};
    {
        if (w.W != d.W) return gf::set_err(GF_ERR_INVALID, "window %d: W=%d, handle built for %d", b, w.W, d.W);
        if (w.n_feature > d.F || w.n_visual > d.NV || w.n_imu > d.W || w.n_wheel > d.W || w.prior_n > d.NPRI || w.prior_nblocks > 64)
            return gf::set_err(GF_ERR_CAPACITY, "window %d exceeds capacity (features %d/%d, visual %d/%d, prior %d/%d)", b, w.n_feature, d.F, w.n_visual, d.NV, w.prior_n, d.NPRI);
        if (w.gnss_enabled && (!d.GO || w.n_gnss > d.NG)) return gf::set_err(GF_ERR_CAPACITY, "window %d: %d GNSS factors, handle built for %d (gf_ba_cfg.max_gnss)", b, w.n_gnss, d.NG);
        if (w.gnss_enabled && (!w.para_rcv_dt || !w.para_rcv_ddt || !w.para_yaw_enu_local || !w.para_anc_ecef || !w.gnss_headers || !w.gnss_iono || (w.n_gnss > 0 && (!w.gnss_frame || !w.gnss_lower || !w.gnss_sys || !w.gnss_ratio || !w.gnss_data))))
            return gf::set_err(GF_ERR_INVALID, "window %d: GNSS enabled but a GNSS array is null", b);
        { double* wp = h->wpar.h + WPAR * (size_t)b; wp[0] = w.G[0]; wp[1] = w.G[1]; wp[2] = w.G[2]; wp[3] = w.vis_sqrt_info; wp[4] = (double)(w.ex_pose_mask & 63); wp[5] = (double)(w.ex_wheel_mask & 63); }   // gravity and visual sqrt_info are per window (estimator.h `g`)
        double* x = h->xs0.h + (size_t)b * d.XS;
        memset(x, 0, d.XS * sizeof(double));
        for (int i = 0; i < d.NP; i++) { memcpy(x + off_pose(i), w.para_Pose + 7 * i, 56); memcpy(x + off_sb(i), w.para_SpeedBias + 9 * i, 72); }
        memcpy(x + off_ex(d.NP), w.para_Ex_Pose, 56); memcpy(x + off_exw(d.NP), w.para_Ex_Pose_wheel, 56); memcpy(x + off_ix(d.NP), w.para_Ix, 24);
        x[off_td(d.NP)] = w.para_Td[0]; x[off_tdw(d.NP)] = w.para_Td_wheel[0];
        if (d.GO) {
            h->ngnss.h[b] = w.gnss_enabled ? w.n_gnss : 0;
            double* ms = h->gn_misc.h + (size_t)b * (GN_MISC + d.NP);
            memset(ms, 0, (GN_MISC + d.NP) * sizeof(double));
            h->gn_gptr.h[(size_t)b * (d.NGRP + 2) + d.NGRP + 1] = 0;
            if (w.gnss_enabled) {
                memcpy(x + d.GO, w.para_rcv_dt, 4 * d.NP * 8); memcpy(x + d.GO + 4 * d.NP, w.para_rcv_ddt, d.NP * 8); x[d.GO + 5 * d.NP] = w.para_yaw_enu_local[0];
                memcpy(x + d.GO + 5 * d.NP + 1, w.para_anc_ecef, 24);
                memcpy(ms, w.gnss_iono, 64); ms[8] = w.gnss_ddt_weight; ms[16] = 1.0; ms[17] = w.gnss_lowspeed ? 0.0 : 1.0;
                memcpy(ms + GN_MISC, w.gnss_headers, d.NP * 8);
                for (int k = 0; k < w.n_gnss; k++) {
                    if (w.gnss_frame[k] < 0 || w.gnss_frame[k] > d.W || w.gnss_lower[k] < 0 || w.gnss_lower[k] >= d.W || w.gnss_sys[k] < 0 || w.gnss_sys[k] > 3)
                        return gf::set_err(GF_ERR_INVALID, "window %d: GNSS factor %d has bad indices", b, k);
                    int* ix = h->gn_idx.h + ((size_t)b * d.NG + k) * 4;
                    ix[0] = w.gnss_frame[k]; ix[1] = w.gnss_lower[k]; ix[2] = w.gnss_sys[k]; ix[3] = 0;
                    double* gd = h->gn_data.h + ((size_t)b * d.NG + k) * GN_STRIDE;
                    memcpy(gd, w.gnss_data + 16 * (size_t)k, 128); gd[16] = w.gnss_ratio[k]; gd[17] = 0;
                }
                {   // groups of factors that share their parameter blocks: same (frame, lower_idx), in ascending order; factors keep their order inside a group
                    std::vector<int> idx(w.n_gnss);
                    for (int k = 0; k < w.n_gnss; k++) idx[k] = k;
                    auto key = [&](int k) { return w.gnss_frame[k] * 64 + w.gnss_lower[k]; };
                    std::stable_sort(idx.begin(), idx.end(), [&](int a, int c) { return key(a) < key(c); });
                    int* gp = h->gn_gptr.h + (size_t)b * (d.NGRP + 2);
                    int* gi = h->gn_gitem.h + (size_t)b * d.NG;
                    int ngrp = 0;
                    for (int p = 0; p < w.n_gnss; p++) {
                        if (p == 0 || key(idx[p]) != key(idx[p - 1])) { if (ngrp >= d.NGRP) return gf::set_err(GF_ERR_CAPACITY, "window %d: more than %d (frame, lower_idx) groups of GNSS factors", b, d.NGRP); gp[ngrp++] = p; }
                        gi[p] = idx[p];
                    }
                    gp[ngrp] = w.n_gnss; gp[d.NGRP + 1] = ngrp;
                }
            }
            if (w.has_anchor) { memcpy(ms + 9, w.anchor_value, 56); ms[18] = 1.0; }
        } else if (w.has_anchor) return gf::set_err(GF_ERR_INVALID, "window %d: PoseAnchorFactor needs a handle built with max_gnss > 0", b);
        for (int f = 0; f < w.n_feature; f++) x[off_feat(d.NP) + f] = w.para_Feature[f];
        if (w.fix_poses) for (int i = 0; i < d.NP; i++) x[off_sb(i)] = x[off_sb(i) + 1] = x[off_sb(i) + 2] = 0.0;  // estimator.cpp:3233-3246
        // column maps (canonical order: pose0, sb0, pose1, ..., ex, exw, sx, sy, sw, td, tdw)
        int* cf = h->colf.h + (size_t)b * d.NFB;
        int col = 0;
        auto add = [&](int blk, bool constant, int ls) { if (constant) cf[blk] = -1; else { cf[blk] = col; col += ls; } };
        for (int i = 0; i < d.NP; i++) { add(fb_pose(i), w.fix_poses != 0, 6); add(fb_sb(i), w.fix_poses != 0, 9); }
        add(fb_ex(d.NP), w.fix_ex_pose != 0, 6);
        // wheel blocks take part when a wheel factor or the prior mentions them (Ceres drops parameter blocks without residual blocks)
        auto part = [&](int id) { if (w.n_wheel > 0) return true; for (int q = 0; q < w.prior_nblocks; q++) if (w.prior_block_id[q] == id) return true; return false; };
        add(fb_exw(d.NP), !part(GF_EX_WHEEL * 4096) || w.fix_ex_wheel, 6);
        for (int q = 0; q < 3; q++) add(fb_sx(d.NP) + q, !part((GF_SX + q) * 4096) || w.fix_ix, 1);
        add(fb_td(d.NP), w.fix_td != 0, 1);
        add(fb_tdw(d.NP), !part(GF_TD_WHEEL * 4096) || w.fix_td_wheel, 1);
        if (d.GO) {   // receiver clocks / anchor: free when a residual block or the prior mentions them (estimator.cpp:2904-2941); yaw_enu_local is held constant (:2932)
            auto inpri = [&](int id) { for (int q = 0; q < w.prior_nblocks; q++) if (w.prior_block_id[q] == id) return true; return false; };
            const bool fac = w.gnss_enabled && !w.gnss_lowspeed;
            for (int i = 0; i < d.NP; i++) for (int q = 0; q < 4; q++) add(fb_rcvdt(d.NP, 4 * i + q), !(w.gnss_enabled && (fac || inpri(GF_RCV_DT * 4096 + 4 * i + q))), 1);
            for (int i = 0; i < d.NP; i++) add(fb_rcvddt(d.NP, i), !(w.gnss_enabled && (fac || inpri(GF_RCV_DDT * 4096 + i))), 1);
            add(fb_yaw(d.NP), true, 1);
            add(fb_anc(d.NP), !(w.gnss_enabled && ((fac && w.n_gnss > 0) || inpri(GF_ANC * 4096))), 3);
        }
        if (!w.fix_ex_pose) M.any_ex = true;
        SolverState& st = h->st0.h[b];
        memset(&st, 0, sizeof st);
        st.radius = 1e4; st.mu = 1e-8; st.R = col; st.last_successful = 1;
        {   // chain form of ba_step: every pose and speed-bias block free (then pose_i sits at column 15 i, sb_i at 15 i + 6), no GNSS blocks, and no speed-bias block in
            // the prior but frame 0's (what marginalisation leaves: estimator.cpp:3448-3520) -- the structure the elimination order relies on
            bool ok = h->chain_mode && !h->big_step && d.GO == 0 && !w.fix_poses;
            for (int q = 0; q < w.prior_nblocks && ok; q++) if (w.prior_block_id[q] / 4096 == GF_SPEEDBIAS && w.prior_block_id[q] % 4096 != 0) ok = false;
            M.chain = ok; M.R = col; st.chain = ok ? 1 : 0;
        }
        // visual factors
        std::vector<char> used(std::max(w.n_feature, 1), 0);
        std::vector<uint32_t> seen_j(std::max(w.n_feature, 1), 0);   // frames j a feature already has a factor for (j <= W <= 30)
        std::vector<signed char> start_i(std::max(w.n_feature, 1), -1);
        for (int k = 0; k < w.n_visual; k++) {
            const size_t kk = (size_t)b * d.NV + k;
            if (w.vis_feature[k] < 0 || w.vis_feature[k] >= w.n_feature || w.vis_i[k] < 0 || w.vis_i[k] >= w.vis_j[k] || w.vis_j[k] > d.W)   // frame i is the feature's start frame: i < j
                return gf::set_err(GF_ERR_INVALID, "window %d: visual factor %d has bad indices", b, k);
            {   // the fixed-extrinsic sweep STORES a factor's Jd^T Jj block into its feature's E^T F row at frame j (vis_lane_eval) and et_rows8 stores the pose-i sum: a second
                // factor of the same (feature, j), or a feature whose factors name different start frames, would lose a term without a trace (round-5 advisor).  The reference
                // cannot build either (one factor per later observation of a feature, all from feature_per_frame[0]: estimator.cpp:3269-3297); a caller-built window can.
                const int f = w.vis_feature[k];
                if (start_i[f] >= 0 && start_i[f] != w.vis_i[k])
                    return gf::set_err(GF_ERR_INVALID, "window %d: visual factor %d names start frame %d for feature %d, an earlier factor named %d", b, k, w.vis_i[k], f, (int)start_i[f]);
                if (seen_j[f] >> w.vis_j[k] & 1u)
                    return gf::set_err(GF_ERR_INVALID, "window %d: visual factor %d repeats the observation of feature %d in frame %d", b, k, f, w.vis_j[k]);
                start_i[f] = (signed char)w.vis_i[k]; seen_j[f] |= 1u << w.vis_j[k];
            }
            h->vis_idx.h[kk] = (w.vis_feature[k] << 10) | (w.vis_i[k] << 5) | w.vis_j[k];
            double* vd = h->vis_data.h + kk * 5;
            memcpy(vd, w.vis_pts_j + 3 * k, 16); memcpy(vd + 2, w.vis_vel_j + 2 * k, 16); vd[4] = w.vis_td_j[k];   // pts_j.z does not enter the residual
            // the observation in the start frame is stored once per feature: every factor of a feature must bring the same one (they are built from
            // feature_per_frame[0], estimator.cpp:3276-3290)
            double* fo = h->feat_obs.h + ((size_t)b * d.F + w.vis_feature[k]) * 6;
            double me[6]; memcpy(me, w.vis_pts_i + 3 * k, 24); memcpy(me + 3, w.vis_vel_i + 2 * k, 16); me[5] = w.vis_td_i[k];
            if (!used[w.vis_feature[k]]) memcpy(fo, me, 48);
            else if (memcmp(fo, me, 48) != 0) return gf::set_err(GF_ERR_INVALID, "window %d: visual factor %d brings a start-frame observation that differs from the other factors of feature %d", b, k, w.vis_feature[k]);
            used[w.vis_feature[k]] = 1;
        }
        int ne = 0;
        for (int f = 0; f < d.F; f++) {
            const bool free_f = f < w.n_feature && used[f] && !(w.feature_fixed && w.feature_fixed[f]);
            h->cole.h[(size_t)b * d.F + f] = free_f ? ne++ : -1;
        }
        st.NE = ne;
        { const double R = st.R, nc = 6.0 * d.NP + 8.0; M.step = (long long)(ne * nc * nc + R * R * R / 3.0 + 2.0 * R * R); }
        M.nvis = w.n_visual; M.npri = w.prior_n; M.nfeat = w.n_feature;
        h->nvis.h[b] = w.n_visual; h->nimu.h[b] = w.n_imu; h->nwh.h[b] = w.n_wheel; h->nfeat.h[b] = w.n_feature;
        {   // pair-sorted order with even padding (each MFMA consumes two factors of one frame pair)
            // stable counting sort over the pair keys i * 64 + j (i < j <= W <= 30): the factor lists of a window are built and packed on its member's thread every frame
            std::vector<int> idx(w.n_visual);
            {
                int cnt[32 * 64 + 1] = {0};
                for (int k = 0; k < w.n_visual; k++) cnt[w.vis_i[k] * 64 + w.vis_j[k] + 1]++;
                for (int q = 0; q < 32 * 64; q++) cnt[q + 1] += cnt[q];
                for (int k = 0; k < w.n_visual; k++) idx[cnt[w.vis_i[k] * 64 + w.vis_j[k]]++] = k;
            }
            int* ord = h->order.h + (size_t)b * d.NVP;
            int n = 0;
            for (size_t p = 0; p < idx.size();) {
                size_t q = p;
                const int key = w.vis_i[idx[p]] * 64 + w.vis_j[idx[p]];
                while (q < idx.size() && w.vis_i[idx[q]] * 64 + w.vis_j[idx[q]] == key) ord[n++] = (key << 16) | idx[q++];   // pair key (i * 64 + j) above the factor index
                if ((q - p) & 1) ord[n++] = -1;
                p = q;
            }
            if (n > d.NVP) return gf::set_err(GF_ERR_CAPACITY, "factor order overflow");
            h->norder.h[b] = n;
            for (int i = n; i < d.NVP; i++) ord[i] = -1;
            M.mfma = (long long)(n / 2) * (w.fix_ex_pose ? 1 : 3); M.jtj = (long long)w.n_visual * 4 * 91; M.norder = n;
        }
        {   // CSR feature -> factors
            int* fp = h->feat_ptr.h + (size_t)b * (d.F + 1);
            std::vector<int> cnt(d.F + 1, 0);
            for (int k = 0; k < w.n_visual; k++) cnt[w.vis_feature[k] + 1]++;
            fp[0] = 0;
            for (int f = 0; f < d.F; f++) fp[f + 1] = fp[f] + cnt[f + 1];
            std::vector<int> cur(fp, fp + d.F);
            bool ident = true;
            for (int k = 0; k < w.n_visual; k++) { const int pos = cur[w.vis_feature[k]]++; h->vis_pos.h[(size_t)b * d.NV + k] = pos; ident &= pos == k; }
            M.pos_ident = ident;
        }
        {   // IMU rows: build them, compare with the slot's table, stage what is new (see gf_ba::imu_patch)
            std::vector<double> rows((size_t)std::max(w.n_imu, 1) * IMU_STRIDE2);
            for (int k = 0; k < w.n_imu; k++) {
                h->imu_i.h[(size_t)b * d.W + k] = w.imu_i[k];
                double* dd = rows.data() + (size_t)k * IMU_STRIDE2;
                dd[0] = w.imu_sum_dt[k];
                memcpy(dd + 1, w.imu_delta_p + 3 * k, 24); memcpy(dd + 4, w.imu_delta_q + 4 * k, 32); memcpy(dd + 8, w.imu_delta_v + 3 * k, 24);
                memcpy(dd + 11, w.imu_lin_ba + 3 * k, 24); memcpy(dd + 14, w.imu_lin_bg + 3 * k, 24);
                memcpy(dd + IMU_JAC, w.imu_jacobian + 225 * k, 225 * 8); memcpy(dd + IMU_COV, w.imu_covariance + 225 * k, 225 * 8);
            }
            double* tab = h->imu_data.h + (size_t)b * d.W * IMU_STRIDE2;
            int* pi = h->imu_pinfo.h + (size_t)b * (d.W + 2);
            const size_t RB = (size_t)IMU_STRIDE2 * sizeof(double);
            const bool on_dev = h->imu_dev[b] == 1;   // anything else: a pack that never reached the device (failed batch) left the mirror ahead -> everything is staged again
            const int old_rows = on_dev ? h->imu_rows[b] : 0;
            auto same = [&](int k, int old_k) { return old_k < old_rows && memcmp(rows.data() + (size_t)k * IMU_STRIDE2, tab + (size_t)old_k * IMU_STRIDE2, RB) == 0; };
            int shift = 0;
            if (on_dev && w.n_imu > 0) {
                int m0 = 0, m1 = 0;
                for (int k = 0; k < w.n_imu; k++) { m0 += same(k, k); m1 += same(k, k + 1); }
                shift = m1 > m0 ? 1 : 0;
            }
            int nd = 0;
            for (int k = 0; k < w.n_imu; k++)
                if (!(on_dev && same(k, k + shift))) { memcpy(h->imu_patch.h + ((size_t)b * d.W + nd) * IMU_STRIDE2, rows.data() + (size_t)k * IMU_STRIDE2, RB); pi[2 + nd++] = k; }
            pi[0] = shift; pi[1] = nd;
            M.imu_dirty = nd;
            memcpy(tab, rows.data(), (size_t)w.n_imu * RB);   // the mirror of what the device table holds after the patch (rows beyond n_imu are never read, never compared: imu_rows)
            h->imu_dev[b] = on_dev ? 2 : 0; h->imu_rows[b] = w.n_imu;
        }
        for (int k = 0; k < w.n_wheel; k++) {
            h->wh_i.h[(size_t)b * d.W + k] = w.wh_i[k];
            double* dd = h->wh_data.h + ((size_t)b * d.W + k) * WH_STRIDE;
            dd[0] = w.wh_sum_dt[k];
            memcpy(dd + 1, w.wh_delta_p + 3 * k, 24); memcpy(dd + 4, w.wh_delta_q + 4 * k, 32); memcpy(dd + 8, w.wh_jacobian + 18 * k, 144);
            memcpy(dd + 26, w.wh_covariance + 36 * k, 288); memcpy(dd + 62, w.wh_lin + 4 * k, 32); memcpy(dd + 66, w.wh_lin_vel + 3 * k, 24);
            memcpy(dd + 69, w.wh_lin_gyr + 3 * k, 24); memcpy(dd + 72, w.wh_vel_1 + 3 * k, 24); memcpy(dd + 75, w.wh_gyr_1 + 3 * k, 24);
        }
        h->pri_n.h[b] = w.prior_n; h->pri_nb.h[b] = w.prior_nblocks;
        if (w.prior_n > 0) {
            int gs = 0;
            for (int q = 0; q < w.prior_nblocks; q++) { h->pri_bid.h[(size_t)b * 64 + q] = w.prior_block_id[q]; gs += gsize_kind(w.prior_block_id[q] / 4096); }
            if (w.prior_J) memcpy(h->pri_J.h + (size_t)b * d.NPRI * d.NPRI, w.prior_J, (size_t)w.prior_n * w.prior_n * 8);
            else {   // resident prior: what this slot's last marginalisation left on the device
                if (h->outJ_n[b] != w.prior_n) return gf::set_err(GF_ERR_INVALID, "window %d: prior_J is null but slot %d holds no marginalisation output of %d columns (it holds %d)", b, b, w.prior_n, h->outJ_n[b]);
                M.pri_res = true;
            }
            memcpy(h->pri_r.h + (size_t)b * d.NPRI, w.prior_r, (size_t)w.prior_n * 8);
            memcpy(h->pri_x0.h + (size_t)b * d.NPRI * 2, w.prior_x0, (size_t)gs * 8);
        }
    }
    // ---- marginalisation layouts (first-appearance order of the blocks over [prior, IMU0, wheel0, visual factors from frame 0])
    {
        for (int mode = 0; mode < 2; mode++) {
            std::vector<int> dropb, keepb, dropf;
            std::vector<char> fseen(std::max(w.n_feature, 1), 0);   // a linear search of dropf per visual factor was the most expensive line of pack_slot
            auto has = [](const std::vector<int>& v, int id) { return std::find(v.begin(), v.end(), id) != v.end(); };
            auto touch = [&](int id, bool dropped) {
                if (id / 4096 == GF_FEATURE) { const int f = id % 4096; if (f < w.n_feature ? !fseen[f] : !has(dropf, id)) { dropf.push_back(id); if (f < w.n_feature) fseen[f] = 1; } return; }
                if (dropped) { if (!has(dropb, id)) { dropb.push_back(id); auto it = std::find(keepb.begin(), keepb.end(), id); if (it != keepb.end()) keepb.erase(it); } }
                else if (!has(keepb, id) && !has(dropb, id)) keepb.push_back(id);
            };
            bool valid = true;
            if (mode == 0) {
                for (int q = 0; q < w.prior_nblocks; q++) { const int id = w.prior_block_id[q]; touch(id, id == GF_POSE * 4096 || id == GF_SPEEDBIAS * 4096); }
                for (int k = 0; k < w.n_imu; k++) if (w.imu_i[k] == 0 && w.imu_sum_dt[k] < 10.0) { touch(GF_POSE * 4096, true); touch(GF_SPEEDBIAS * 4096, true); touch(GF_POSE * 4096 + 1, false); touch(GF_SPEEDBIAS * 4096 + 1, false); }
                for (int k = 0; k < w.n_wheel; k++) if (w.wh_i[k] == 0 && w.wh_sum_dt[k] < 10.0) {
                    touch(GF_POSE * 4096, true); touch(GF_POSE * 4096 + 1, false); touch(GF_EX_WHEEL * 4096, false); touch(GF_SX * 4096, false); touch(GF_SY * 4096, false);
                    touch(GF_SW * 4096, false); touch(GF_TD_WHEEL * 4096, false);
                }
                if (d.GO && w.gnss_enabled) {   // estimator.cpp:3390-3431
                    for (int k = 0; k < w.n_gnss; k++) if (w.gnss_frame[k] == 0) {
                        touch(GF_POSE * 4096, true); touch(GF_SPEEDBIAS * 4096, true); touch(GF_POSE * 4096 + 1, false); touch(GF_SPEEDBIAS * 4096 + 1, false);
                        touch(GF_RCV_DT * 4096 + w.gnss_sys[k], true); touch(GF_RCV_DDT * 4096, true); touch(GF_YAW * 4096, false); touch(GF_ANC * 4096, false);
                    }
                    for (int q = 0; q < 4; q++) { touch(GF_RCV_DT * 4096 + q, true); touch(GF_RCV_DT * 4096 + 4 + q, false); touch(GF_RCV_DDT * 4096, true); touch(GF_RCV_DDT * 4096 + 1, false); }
                }
                for (int k = 0; k < w.n_visual; k++) if (w.vis_i[k] == 0) {
                    touch(GF_POSE * 4096, true); touch(GF_POSE * 4096 + w.vis_j[k], false); touch(GF_EX_POSE * 4096, false); touch(GF_FEATURE * 4096 + w.vis_feature[k], true);
                    touch(GF_TD * 4096, false);
                }
            } else {
                bool found = false;
                for (int q = 0; q < w.prior_nblocks; q++) { const int id = w.prior_block_id[q]; const bool dr = id == GF_POSE * 4096 + (d.W - 1); found |= dr; touch(id, dr); }
                valid = found && w.prior_n > 0;
            }
            int* mc = h->mcolf[mode].h + (size_t)b * d.NFB;
            int* me = h->mcole[mode].h + (size_t)b * d.F;
            for (int q = 0; q < d.NFB; q++) mc[q] = -1;
            for (int f = 0; f < d.F; f++) me[f] = -1;
            auto fblk = [&](int id) {
                const int kind = id / 4096, i = id % 4096;
                switch (kind) { case 0: return fb_pose(i); case 1: return fb_sb(i); case 2: return fb_ex(d.NP); case 3: return fb_exw(d.NP); case 4: return fb_sx(d.NP);
                                case 5: return fb_sx(d.NP) + 1; case 6: return fb_sx(d.NP) + 2; case 7: return fb_td(d.NP); case 10: return fb_rcvdt(d.NP, i); case 11: return fb_rcvddt(d.NP, i);
                                case 12: return fb_yaw(d.NP); case 13: return fb_anc(d.NP); default: return fb_tdw(d.NP); }
            };
            int mp = 0, n = 0;
            for (int id : dropb) { mc[fblk(id)] = mp; mp += lsize_kind(id / 4096); }
            for (int id : keepb) { mc[fblk(id)] = mp + n; n += lsize_kind(id / 4096); }
            for (size_t q = 0; q < dropf.size(); q++) me[dropf[q] % 4096] = (int)q;
            if (mp + (int)dropf.size() == 0) valid = false;
            if (valid && (n > h->marg_ncap || mp > gfb::MPMAX || mp + n > d.RP)) return gf::set_err(GF_ERR_CAPACITY, "window %d: marginalisation sizes mp=%d n=%d exceed this build (n <= %d)", b, mp, n, h->marg_ncap);
            int* inf = h->minfo[mode].h + (size_t)b * 4;
            inf[0] = mp; inf[1] = (int)dropf.size(); inf[2] = n; inf[3] = valid ? 1 : 0;
            h->keep_ids[mode][b] = keepb;
            // factor order (mode 0: visual factors of features starting at frame 0)
            int* ord = h->morder[mode].h + (size_t)b * d.NVP;
            int no = 0;
            if (mode == 0) {
                std::vector<int> idx;   // factors of frame 0 by their second frame, stable: counting sort over j
                {
                    int cnt[65] = {0};
                    for (int k = 0; k < w.n_visual; k++) if (w.vis_i[k] == 0) cnt[w.vis_j[k] + 1]++;
                    for (int q = 0; q < 64; q++) cnt[q + 1] += cnt[q];
                    idx.resize(cnt[64]);
                    for (int k = 0; k < w.n_visual; k++) if (w.vis_i[k] == 0) idx[cnt[w.vis_j[k]]++] = k;
                }
                for (size_t p = 0; p < idx.size();) {
                    size_t q = p;
                    while (q < idx.size() && w.vis_j[idx[q]] == w.vis_j[idx[p]]) ord[no++] = (w.vis_j[idx[p]] << 16) | idx[q++];   // pair (0, j)
                    if ((q - p) & 1) ord[no++] = -1;
                    p = q;
                }
            }
            h->mnorder[mode].h[b] = no; M.mno[mode] = no;
            for (int i = no; i < d.NVP; i++) ord[i] = -1;
        }
    }
    return GF_OK;
}

// closes a batch of packed slots: what the launches need to know about the batch as a whole (slots listed in `active`, or 0 .. count - 1)
void finalize_pack(gf_ba* h, const int* active, int n_active, int count) {
    const Dims& d = h->d;
    bool any_ex = false; long long mfma = 0, step = 0, jtj = 0; int mv = 0, mo = 0, mp = 0, mf = 0;
    h->active.clear(); h->all_pri_res = n_active > 0; h->any_pri_res = false;
    for (int q = 0; q < n_active; q++) {
        const gf_ba::SlotMeta& M = h->meta[active ? active[q] : q];
        h->active.push_back(active ? active[q] : q);
        if (M.npri > 0) { h->all_pri_res &= M.pri_res; h->any_pri_res |= M.pri_res; }
        any_ex |= M.any_ex; mfma += M.mfma; step += M.step; jtj += M.jtj; mv = std::max(mv, M.nvis); mo = std::max(mo, M.norder); mp = std::max(mp, M.npri); mf = std::max(mf, M.nfeat);
    }
    h->any_ex = any_ex; h->mfma_per_lin = mfma; h->step_flops = step; h->jtj_alg_flops = jtj;
    h->n_chain = h->n_dense = h->chain_nd = 0;
    for (int q = 0; q < n_active; q++) {
        const gf_ba::SlotMeta& M = h->meta[active ? active[q] : q];
        if (M.chain) { h->n_chain++; h->chain_nd = std::max(h->chain_nd, M.R - 9 * d.NP); } else h->n_dense++;
    }
    h->max_vis = mv; h->max_order = mo; h->max_prior = mp; h->max_feat = mf;
    h->max_imu_dirty = 0;
    for (int q = 0; q < n_active; q++) h->max_imu_dirty = std::max(h->max_imu_dirty, h->meta[active ? active[q] : q].imu_dirty);
    h->count = count;
    (void)d;
}

int pack_windows(gf_ba* h, const gf_ba_window* ws, int count) {
    const Dims& d = h->d;
    if (count < 1 || count > d.B) return gf::set_err(GF_ERR_INVALID, "count %d outside 1..%d", count, d.B);
    // the windows are independent: pack them on a few host threads (the estimator group hands over hundreds per call)
    {
        const int nt = std::max(1, std::min({d.B / 4, (int)std::thread::hardware_concurrency(), 16}));
        std::atomic<int> next{0}, failed{GF_OK};
        std::mutex em; std::string emsg;
        auto work = [&]() {
            for (int b = next++; b < d.B; b = next++) {
                const int rc = pack_slot(h, b, ws[std::min(b, count - 1)]);   // unused slots replicate the last window (kernels run on the whole batch)
                if (rc != GF_OK) { std::lock_guard<std::mutex> lk(em); if (failed == GF_OK) { failed = rc; emsg = gf_last_error(); } }
            }
        };
        std::vector<std::thread> ths;
        for (int t = 1; t < nt; t++) ths.emplace_back(work);
        work();
        for (auto& t : ths) t.join();
        if (failed != GF_OK) return gf::set_err(failed, "%s", emsg.c_str());
    }
    finalize_pack(h, nullptr, count, count);
    return GF_OK;
}

int upload(gf_ba* h) {
    hipStream_t s = h->stream;
    const Dims& d = h->d;
    const size_t B = d.B, nv = (size_t)h->max_vis, no = (size_t)std::max(h->max_order, 1), np2 = (size_t)h->max_prior * h->max_prior;
    static const bool dbg = getenv("GF_BA_UPLOAD_DEBUG") != nullptr;
    long long mark = g_up_bytes; int stage = 0;
    auto lapb = [&](const char* what) { if (dbg) { fprintf(stderr, "upload %d %-28s %8.2f MB\n", stage++, what, (g_up_bytes - mark) / 1e6); mark = g_up_bytes; } };
    // every table either as a copy of its own or as a descriptor of the one gather kernel (h->upload_kernel)
    UpBuilder U;
    const bool ker = h->upload_kernel;
    hipError_t err = hipSuccess;
    auto up1 = [&](auto& b) { if (ker) U.all(b); else if (err == hipSuccess) err = b.up(s); };
    auto up2 = [&](auto& b, size_t rows, size_t pitch, size_t used) { if (ker) U.add(b, rows, pitch, used); else if (err == hipSuccess) err = b.up2d(s, rows, pitch, used); };
    up1(h->xs0); lapb("xs0");
    const size_t nf = (size_t)std::max(h->max_feat, 1);
    for (auto* b : {&h->colf, &h->nvis, &h->nimu, &h->nwh, &h->nfeat, &h->norder, &h->imu_i, &h->wh_i, &h->pri_n, &h->pri_nb, &h->pri_bid}) up1(*b);
    up2(h->cole, B, d.F, (size_t)d.F);   // whole rows: entries beyond n_feature are -1 markers the kernels rely on
    up2(h->feat_ptr, B, (size_t)d.F + 1, (size_t)d.F + 1); lapb("small tables, cole, feat_ptr");
    // per-window tables are laid out for the handle's capacity; only what this batch fills is copied
    up2(h->vis_idx, B, d.NV, nv);
    h->pos_ident = true;   // over every slot's last pack (slots that sit a batch out keep their tables)
    for (const gf_ba::SlotMeta& M : h->meta) h->pos_ident &= M.pos_ident;
    if (!h->pos_ident) up2(h->vis_pos, B, d.NV, nv);
    lapb("vis int tables x2");
    up2(h->order, B, d.NVP, no); lapb("order");
    up2(h->vis_data, B, (size_t)d.NV * 5, nv * 5);
    up2(h->feat_obs, B, (size_t)d.F * 6, nf * 6); lapb("vis_data + feat_obs");
    if (!h->any_pri_res) up2(h->pri_J, B, (size_t)d.NPRI * d.NPRI, np2);
    bool imu_patched = false;
    {   // IMU tables: everything when many rows are new (first frames, whole-batch uploads), else the new rows + one small kernel that moves / patches the slots' tables
        bool all_dev = true;
        for (int b : h->active) all_dev &= h->imu_dev[b] == 2;
        if (all_dev && (int)h->active.size() == d.B && h->max_imu_dirty <= d.W / 2) {
            if (h->max_imu_dirty > 0) up2(h->imu_patch, B, (size_t)d.W * IMU_STRIDE2, (size_t)h->max_imu_dirty * IMU_STRIDE2);
            up1(h->imu_pinfo);
            imu_patched = true;
        } else up1(h->imu_data);
    }
    for (auto* b : {&h->wh_data, &h->pri_r, &h->pri_x0, &h->wpar}) up1(*b);
    lapb("imu, wheel, pri_r, pri_x0, wpar");
    if (h->d.GO) { up1(h->ngnss); up1(h->gn_idx); up1(h->gn_data); up1(h->gn_misc); up1(h->gn_gptr); up1(h->gn_gitem); }
    for (int m = 0; m < 2; m++) {
        for (auto* b : {&h->mcolf[m], &h->mcole[m], &h->mnorder[m], &h->minfo[m]}) up1(*b);
        {   // the marginalisation's factor order: only as far as this layout fills it (MARGIN_OLD: the factors of the features that start at frame 0; MARGIN_SECOND_NEW: none)
            size_t mo = 0;
            for (const gf_ba::SlotMeta& M : h->meta) mo = std::max(mo, (size_t)M.mno[m]);
            up2(h->morder[m], B, d.NVP, mo);
        }
    }
    lapb("gnss + marg layouts");
    up1(h->st0);
    HIPCHK(err);
    if (ker) {
        if (!U.B.ok) return gf::set_err(GF_ERR_HIP, "upload kernel: a host mirror is not mapped into the device's address space, or too many / too large tables (GF_BA_UPLOAD=copies selects the copy path)");
        g_up_calls++;
        HIPCHK(U.B.launch<0>(s));
    }
    // priors: host tables of the slots that brought one (mixed batches), device-resident ones straight from the marginalisation's output
    if (h->any_pri_res && !h->all_pri_res)   // mixed batch: only the active slots that brought a host prior; a slot that sits this batch out keeps what the device holds (its host mirror was never written)
        for (int b : h->active) if (!h->meta[b].pri_res && h->meta[b].npri > 0)
            HIPCHK(hipMemcpyAsync(h->pri_J.d + (size_t)b * d.NPRI * d.NPRI, h->pri_J.h + (size_t)b * d.NPRI * d.NPRI, (size_t)h->meta[b].npri * h->meta[b].npri * 8, hipMemcpyHostToDevice, s));
    if (h->any_pri_res) {   // device-resident priors: marginalisation output -> prior table, without the round trip through the host
        const size_t pitch = (size_t)d.NPRI * d.NPRI * 8;
        if (h->all_pri_res) HIPCHK(hipMemcpy2DAsync(h->pri_J.d, pitch, h->outJ.d, pitch, np2 * 8, B, hipMemcpyDeviceToDevice, s));
        else for (int b : h->active) if (h->meta[b].pri_res) HIPCHK(hipMemcpyAsync(h->pri_J.d + (size_t)b * d.NPRI * d.NPRI, h->outJ.d + (size_t)b * d.NPRI * d.NPRI, (size_t)h->meta[b].npri * h->meta[b].npri * 8, hipMemcpyDeviceToDevice, s));
    }
    lapb("pri_J");
    if (imu_patched) {
        ba_imu_patch<<<dim3(d.B), 256, 0, s>>>(h->imu_data.d, h->imu_patch.d, h->imu_pinfo.d, d.W);
        HIPCHK(hipGetLastError());
    }
    for (int b = 0; b < d.B; b++) h->imu_dev[b] = 1;   // set only here, behind a successful copy: the full upload covers every slot with its mirror; the patch path required state 2 of every slot
    ba_setup<<<dim3(h->d.B), 256, 0, s>>>(h->win());
    HIPCHK(hipGetLastError());
    return GF_OK;
}

// the solved state and the solver summary of slot b, from the host mirrors the last download filled, into the caller's window (either may be null)
void unpack_state(gf_ba* h, int b, gf_ba_window* wp, gf_ba_summary* sp) {
    const Dims& d = h->d;
    const SolverState& st = h->st.h[b];
    if (wp) {
        gf_ba_window& w = *wp;
        const double* x = h->xs.h + ((size_t)st.cur * d.B + b) * d.XS;
        for (int i = 0; i < d.NP; i++) { memcpy(w.para_Pose + 7 * i, x + off_pose(i), 56); memcpy(w.para_SpeedBias + 9 * i, x + off_sb(i), 72); }
        memcpy(w.para_Ex_Pose, x + off_ex(d.NP), 56); memcpy(w.para_Ex_Pose_wheel, x + off_exw(d.NP), 56); memcpy(w.para_Ix, x + off_ix(d.NP), 24);
        w.para_Td[0] = x[off_td(d.NP)]; w.para_Td_wheel[0] = x[off_tdw(d.NP)];
        for (int f = 0; f < w.n_feature; f++) w.para_Feature[f] = x[off_feat(d.NP) + f];
        if (d.GO && w.gnss_enabled) { memcpy(w.para_rcv_dt, x + d.GO, 4 * d.NP * 8); memcpy(w.para_rcv_ddt, x + d.GO + 4 * d.NP, d.NP * 8); w.para_yaw_enu_local[0] = x[d.GO + 5 * d.NP]; memcpy(w.para_anc_ecef, x + d.GO + 5 * d.NP + 1, 24); }
    }
    if (sp) {
        gf_ba_summary& s = *sp;
        s.iterations = st.iterations; s.successful_steps = st.successful; s.termination = st.termination; s.initial_cost = st.initial_cost; s.final_cost = st.x_cost;
        s.radius = st.radius;
    }
}

int reset_state(gf_ba* h) {
    const Dims& d = h->d;
    HIPCHK(hipMemcpyAsync(h->xs.d, h->xs0.d, (size_t)d.B * d.XS * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->st.d, h->st0.d, (size_t)d.B * sizeof(SolverState), hipMemcpyDeviceToDevice, h->stream));
    return GF_OK;
}

__global__ void __launch_bounds__(256) ba_poison_lds(int ndoubles) {   // GF_BA_POISON: every CU's LDS full of NaN before a launch (1024 blocks of 160 KB: every CU gets some)
    extern __shared__ double lds_all[];
    for (int i = threadIdx.x; i < ndoubles; i += 256) lds_all[i] = __longlong_as_double(-1LL);
}
static void poison_lds(gf_ba* h) {
    static const int pmode = getenv("GF_BA_POISON") ? atoi(getenv("GF_BA_POISON")) : 0;
    static const bool poison = pmode == 1 || pmode == 3;
    if (!poison) return;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ba_poison_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    ba_poison_lds<<<dim3(1024), 256, 160 * 1024, h->stream>>>(160 * 1024 / 8);
}

// One linearisation of the resident batch at the state buffer `which_state` into the buffers `which` (-1: the candidate's).
// Visual sweep (Vc, E^T F rows), then prior / IMU / wheel (H, g) and the GNSS blocks, on the handle's stream.  Every buffer
// has one writing kernel and every sum a fixed order: no zeroing or reset passes, no atomics.
int launch_visual(gf_ba* h, Win w, bool ex, int which, int which_state, int only_valid, bool timed_split = false) {
    const Dims& d = h->d;
    const size_t lds = ex ? h->vwinx_lds : h->vwin_lds;
    w.vtile = lds ? nullptr : h->vtile.d;
    w.vpair = h->vpair.d;
    if (ex && only_valid == 2) {   // MARGIN_OLD pass: eight wavefronts, NP - 1 pair tiles, always in LDS
        w.vtile = nullptr;
        ba_linearize_visual_win<true, kVWM><<<dim3(d.B), 64 * kVWM, h->vwinm_lds, h->stream>>>(w, h->sbufs(), which, which_state, only_valid);
    }
    else if (ex) ba_linearize_visual_win<true, kVWX><<<dim3(d.B), 64 * kVWX, lds, h->stream>>>(w, h->sbufs(), which, which_state, only_valid);
    else if (h->split_jtj && only_valid != 2) {   // north_star's formulation, measured next to the fused kernel: the sweep writes block rows to HBM, a second kernel only contracts them
        w.vrows = h->vrows.d;
        ba_linearize_visual_win<false, kVW, 1><<<dim3(d.B), 64 * kVW, lds, h->stream>>>(w, h->sbufs(), which, which_state, only_valid);
        HIPCHK(hipGetLastError());
        if (timed_split) HIPCHK(hipEventRecord(h->ev_split[0], h->stream));
        ba_linearize_visual_win<false, kVW, 2><<<dim3(d.B), 64 * kVW, lds, h->stream>>>(w, h->sbufs(), which, which_state, only_valid);
        if (timed_split) { HIPCHK(hipEventRecord(h->ev_split[1], h->stream)); h->split_timed = true; }
    }
    else ba_linearize_visual_win<false, kVW><<<dim3(d.B), 64 * kVW, lds, h->stream>>>(w, h->sbufs(), which, which_state, only_valid);
    HIPCHK(hipGetLastError());
    return GF_OK;
}

int launch_linearize(gf_ba* h, int which, int which_state, int only_valid, bool timed, bool misc_with_step = false) {
    const Dims& d = h->d;
    Win w = h->win();
    poison_lds(h);
    if (timed) HIPCHK(hipEventRecord(h->ev[2], h->stream));
    if (int rc = launch_visual(h, w, h->any_ex, which, which_state, only_valid, timed)) return rc;
    if (timed) HIPCHK(hipEventRecord(h->ev[3], h->stream));
    if (misc_with_step) return GF_OK;   // the candidate's prior / IMU / wheel sweep rides in front of the next step (ba_misc_step)
    // same stream as the visual sweep: the two sweeps fill the CUs' LDS and registers and so exclude each other anyway, and a second stream only added the
    // cross-stream join in front of the next ba_step (12 us instead of 6; solve 2.49 -> 2.40 ms)
    ba_linearize_misc_win<<<dim3(d.B), 64 * kMW, h->mwin_lds, h->stream>>>(w, which, which_state, only_valid, 0);
    if (d.GO) ba_linearize_gnss<<<dim3(d.B), 256, 0, h->stream>>>(w, which, which_state, only_valid, 0);
    HIPCHK(hipGetLastError());
    return GF_OK;
}

// The fixed launch schedule of one batch solve: initial linearisation, then max_iters x (step, linearise candidate), final accept.
int run_solve(gf_ba* h, int max_iters) {
    const Dims& d = h->d;
    const auto t_begin = std::chrono::steady_clock::now();
    if (int rc = launch_linearize(h, 0, 0, 0, false)) return rc;
    Win w = h->win();
    StepBufs sb = h->sbufs();
    const bool fuse_misc = h->fuse_misc && h->can_fuse_misc && d.GO == 0 && h->n_chain == 0;
    for (int it = 0; it <= max_iters; it++) {
        if (h->max_solver_time > 0.0 && it > 0 && it < max_iters) {
            // trust_region_minimizer.cc checks total_solver_time >= max_solver_time_in_seconds at the top of every iteration: with the option on, the host
            // waits for the iterations enqueued so far and, once over the limit, enqueues only the closing step (accept / reject of the last candidate)
            HIPCHK(hipStreamSynchronize(h->stream));
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() >= h->max_solver_time) it = max_iters;
        }
        poison_lds(h);
        const bool time_step = it == 1 && max_iters >= 1;
        if (time_step) HIPCHK(hipEventRecord(h->ev[6], h->stream));
        // windows in the chain form first (256 threads, two per CU), then -- if the batch holds any -- the others in the dense form; each kernel leaves the other's windows alone
        if (h->n_chain > 0) ba_step_chain<<<dim3(d.B), 256, ch_lds_doubles(h->chain_nd) * sizeof(double), h->stream>>>(w, sb, it == 0 ? 1 : 0, max_iters, it == max_iters ? 1 : 0);
        if (h->n_chain > 0 && h->n_dense == 0) { }
        else if (h->step_waves == 4) {
            if (h->big_step) ba_step<true, 4><<<dim3(d.B), 256, 0, h->stream>>>(w, sb, it == 0 ? 1 : 0, max_iters, it == max_iters ? 1 : 0);
            else ba_step<false, 4><<<dim3(d.B), 256, h->step_lds, h->stream>>>(w, sb, it == 0 ? 1 : 0, max_iters, it == max_iters ? 1 : 0);
        } else if (h->big_step) ba_step<true><<<dim3(d.B), 512, 0, h->stream>>>(w, sb, it == 0 ? 1 : 0, max_iters, it == max_iters ? 1 : 0);
        else if (fuse_misc && it > 0) ba_misc_step<<<dim3(d.B), 512, std::max(h->step_lds, h->mwin_lds), h->stream>>>(w, sb, max_iters, it == max_iters ? 1 : 0);
        else ba_step<false><<<dim3(d.B), 512, h->step_lds, h->stream>>>(w, sb, it == 0 ? 1 : 0, max_iters, it == max_iters ? 1 : 0);
        HIPCHK(hipGetLastError());
        if (time_step) { HIPCHK(hipEventRecord(h->ev[7], h->stream)); h->stats.step_launches++; h->stats.step_flops += h->step_flops; }
        if (it < max_iters) {
            // candidate state lives in buffer (1 - cur) of each window: the kernels pick the right one per window.  The candidate of the LAST iteration is only judged by its
            // cost (the step behind it accepts or rejects and the solve ends): its sweeps run cost-only (only_valid = 3; round 6.  Not with GNSS blocks, whose kernel adds
            // into H and the cost in one pass, not in the split formulation, not when the wall-clock cut may end the solve at another iteration; GF_BA_COST_ONLY=0: off)
            const bool cost_only = h->cost_only && it == max_iters - 1 && it > 0 && d.GO == 0 && !h->split_jtj && !fuse_misc && !(h->max_solver_time > 0.0);
            if (int rc = launch_linearize(h, -1, -1, cost_only ? 3 : 1, it == 0, fuse_misc)) return rc;
            if (it == 0) { h->stats.jtj_launches++; h->stats.jtj_flops += h->mfma_per_lin * 2048; h->stats.jtj_alg_flops += h->jtj_alg_flops; }
        }
    }
    h->stats.solves += h->count;
    return GF_OK;
}

int run_marginalize(gf_ba* h, int mode) {
    const Dims& d = h->d;
    Win w = h->win();
    poison_lds(h);
    Win wm = w;   // marginalisation column maps: dropped blocks first, no block constant
    wm.colf = h->mcolf[mode].d; wm.cole = h->mcole[mode].d; wm.order = h->morder[mode].d; wm.norder = h->mnorder[mode].d;
    // the dropped frame's factors are linearised at the current state into the other buffer set
    if (mode == 0) { if (int rc = launch_visual(h, wm, true, -1, -2, 2)) return rc; }
    ba_linearize_misc_win<<<dim3(d.B), 64 * kMW, h->mwin_lds, h->stream>>>(wm, -1, -2, 2, mode == 0 ? 1 : 2);
    if (mode == 0 && d.GO) ba_linearize_gnss<<<dim3(d.B), 256, 0, h->stream>>>(wm, -1, -2, 2, 1);
    MargOut mo{h->outJ.d, h->outr.d};
    // Pivot cut of the rank-revealing factorisation of the kept system, and the least-squares right-hand side (gf_ba_marg.hpp); environment switches for experiments.
    // The reference cuts EIGENVALUES at 1e-8 (marginalization_factor.cpp:294-299).  A pivot is the largest diagonal entry of the trailing Schur complement S_k, and
    // lambda_(k+1)(A) <= lambda_max(S_k), lambda_max(S_k) / (n - k) <= pivot_k <= lambda_max(S_k): pivots and tail eigenvalues agree within a factor of the trailing
    // dimension only, so no pivot cut reproduces the eigenvalue cut direction for direction.  Measured on the closed-loop GNSS replays (the states that hang on exactly
    // these directions; worst receiver clock / anchor deviation from the oracle pipeline over W = 10 handed-in, W = 20 handed-in, own initialiser, raw ephemerides,
    // configs[4]): cut 1e-10: 1e-6 1e-8 3e-8 6e-5 1e-6;  1e-9: 1e-6 1e-8 3e-8 6e-5 3e-4;  1e-8: 1e-6 7e-8 3e-8 2e-7 4e-4;  3e-8: 1e-6 3e-7 3e-8 2e-7 1e-6 [m].
    static const double piv_eps = getenv("GF_MARG_PIVOT_EPS") ? atof(getenv("GF_MARG_PIVOT_EPS")) : 3e-8;
    static const int ls_rhs = getenv("GF_MARG_LS_RHS") ? atoi(getenv("GF_MARG_LS_RHS")) : 1;
    poison_lds(h);
    if (h->big_marg) ba_marg_finish<true><<<dim3(d.B), 512, 0, h->stream>>>(wm, h->sbufs(), reinterpret_cast<const MargInfo*>(h->minfo[mode].d), mo, mode == 0 ? 1 : 0, piv_eps, ls_rhs, 0);
    else ba_marg_finish<false><<<dim3(d.B), 512, h->marg_lds, h->stream>>>(wm, h->sbufs(), reinterpret_cast<const MargInfo*>(h->minfo[mode].d), mo, mode == 0 ? 1 : 0, piv_eps, ls_rhs, (int)(h->marg_lds / sizeof(double)));
    HIPCHK(hipGetLastError());
    return GF_OK;
}

}  // namespace

extern "C" {

int gf_pose_subset_mask(int extrinsic_type) {   // parameters.cpp:394-420 (camera), :280-306 (wheel) -> the index sets of estimator.cpp:2969-2985, :3010-3026 (index 6, qw, is no tangent component)
    switch (extrinsic_type) {
        case 0: return 0x00;          // ADJUST_*_ALL: {}
        case 2: return 0x07;          // ADJUST_*_ROTATION: {0, 1, 2, 6}
        case 3: return 0x04;          // ADJUST_*_NO_Z: {2, 6}
        case 4: return 0x3c;          // ADJUST_*_NO_ROTATION_NO_Z: {2, 3, 4, 5, 6}
        default: return 0x38;         // 1 = ADJUST_*_TRANSLATION: {3, 4, 5, 6}; out of range: the reference warns and keeps its zero-initialised enum, which is *_TRANSLATION
    }
}

int gf_ba_create(const gf_ba_cfg* cfg, gf_ba** out) {
    if (!cfg || !out) return gf::set_err(GF_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->window_size < 2 || cfg->window_size > 30 || cfg->max_features < 1 || cfg->max_visual < 1 || cfg->batch < 1) return gf::set_err(GF_ERR_INVALID, "bad gf_ba_cfg");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return gf::set_err(GF_ERR_NO_DEVICE, "no HIP device available; the HIP path has no CPU fallback");
    gf_ba* h = new gf_ba();
    h->cfg = *cfg;
    Dims& d = h->d;
    d.B = cfg->batch; d.W = cfg->window_size; d.NP = d.W + 1; d.F = cfg->max_features; d.NV = cfg->max_visual;
    d.NVP = ((d.NV + d.NP * d.NP / 2 + 63) / 64) * 64;
    const bool gnss = cfg->max_gnss > 0;
    d.NG = gnss ? ((cfg->max_gnss + 63) & ~63) : 0; d.NGRP = gnss ? 4 * d.NP : 0;
    const int Rmax = 15 * d.NP + 17 + (gnss ? 5 * d.NP + 3 : 0);
    d.RP = (Rmax + 1 + 15) & ~15; /* one spare column: the Schur GEMM carries the right-hand side in column R */ d.GO = gnss ? ((16 * d.NP + 20 + d.F + 3) & ~3) : 0; d.XS = gnss ? ((d.GO + 5 * d.NP + 4 + 3) & ~3) : ((16 * d.NP + 20 + d.F + 3) & ~3); d.NFB = 2 * d.NP + 7 + (gnss ? 5 * d.NP + 2 : 0); d.FP = (d.F + 3) & ~3; d.NPRI = d.RP; d.ECW = (6 * d.NP + 8 + 15) & ~15; d.NC = 6 * d.NP + 8; d.NVC = (d.NC * (d.NC + 1) / 2 + 3) & ~3;
    h->step_lds = (size_t)(Rmax + 1) * (Rmax + 2) / 2 * sizeof(double);  // packed lower S plus the right-hand-side row
    h->big_step = h->step_lds + 27 * 1024 > 160 * 1024 || getenv("GF_BA_FORCE_GLOBAL") != nullptr;   // reduced system too large for LDS: ba_step<true> keeps it in global memory
    if (Rmax + 1 > 512) { delete h; return gf::set_err(GF_ERR_INVALID, "window_size %d: reduced system (%d) exceeds 511 columns", d.W, Rmax); }
    if (d.NV > 65535) { delete h; return gf::set_err(GF_ERR_INVALID, "max_visual %d exceeds 65535", d.NV); }
    if (misc_win_lds_doubles(d.W) * sizeof(double) > 160 * 1024) { /* the sweep has no static LDS */ delete h; return gf::set_err(GF_ERR_INVALID, "window_size %d: the IMU / wheel block rows exceed LDS (window_size <= 20 in this build)", d.W); }
#define A_(x) do { g_alloc_what = #x; if (int rc_ = (x)) { h->release(); delete h; return rc_; } } while (0)
#define H_(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { gf::set_err(GF_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_)); h->release(); delete h; return GF_ERR_HIP; } } while (0)
    {   // the solver is a chain of short launches that each want every CU: its queue goes first when the tracker's kernels of the same process compete for them
        int least = 0, greatest = 0;
        H_(hipDeviceGetStreamPriorityRange(&least, &greatest));
        const char* e = getenv("GF_BA_STREAM_PRIORITY");
        const int prio = (e && !strcmp(e, "default")) ? least + (greatest - least) / 2 : greatest;
        H_(hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, prio));
    }
    for (auto& e : h->ev) H_(hipEventCreate(&e));
    H_(hipEventCreateWithFlags(&h->ev_gather, hipEventDisableTiming));
    H_(hipEventCreateWithFlags(&h->ev_gather_done, hipEventDisableTiming));
    const size_t B = d.B, VS = d.RP + d.FP;
    A_(h->xs0.alloc(B * d.XS, true)); A_(h->xs.alloc(2 * B * d.XS, true));
    A_(h->colf.alloc(B * d.NFB, true)); A_(h->cole.alloc(B * d.F, true)); A_(h->nvis.alloc(B, true)); A_(h->nimu.alloc(B, true)); A_(h->nwh.alloc(B, true)); A_(h->nfeat.alloc(B, true));
    A_(h->vis_idx.alloc(B * d.NV, true)); A_(h->vis_data.alloc(B * d.NV * 5, true)); A_(h->feat_obs.alloc(B * d.F * 6, true));
    A_(h->order.alloc(B * d.NVP, true)); A_(h->norder.alloc(B, true)); A_(h->feat_ptr.alloc(B * (d.F + 1), true)); A_(h->vis_pos.alloc(B * d.NV, true));
    A_(h->imu_i.alloc(B * d.W, true)); A_(h->imu_data.alloc(B * d.W * IMU_STRIDE2, true)); A_(h->wh_i.alloc(B * d.W, true)); A_(h->wh_data.alloc(B * d.W * WH_STRIDE, true));
    A_(h->pri_n.alloc(B, true)); A_(h->pri_nb.alloc(B, true)); A_(h->pri_bid.alloc(B * 64, true)); A_(h->pri_J.alloc(B * d.NPRI * d.NPRI, true));
    A_(h->pri_r.alloc(B * d.NPRI, true)); A_(h->pri_x0.alloc(B * d.NPRI * 2, true));
    if (hipMalloc((void**)&h->st.d, B * sizeof(SolverState)) != hipSuccess || hipHostMalloc((void**)&h->st.h, B * sizeof(SolverState), hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void**)&h->st0.d, B * sizeof(SolverState)) != hipSuccess || hipHostMalloc((void**)&h->st0.h, B * sizeof(SolverState), hipHostMallocDefault) != hipSuccess) {
        h->release(); delete h; return gf::set_err(GF_ERR_HIP, "allocation of solver state failed");
    }
    h->st.n = h->st0.n = B;
    // page-locked memory is recycled inside the process like device memory: a slot that was never packed would hand the kernels the SolverState of an earlier handle
    memset(h->st.h, 0, B * sizeof(SolverState)); memset(h->st0.h, 0, B * sizeof(SolverState));
    H_(hipMemset(h->st.d, 0, B * sizeof(SolverState))); H_(hipMemset(h->st0.d, 0, B * sizeof(SolverState)));
    { void* p = nullptr; h->st0.hd = hipHostGetDevicePointer(&p, h->st0.h, 0) == hipSuccess ? static_cast<SolverState*>(p) : nullptr; (void)hipGetLastError(); }
    // GF_BA_UPLOAD=kernel | copies: a batch's tables through one gather kernel that reads the page-locked mirrors over the bus, or one hipMemcpy(2D)Async per table
    h->upload_kernel = !(getenv("GF_BA_UPLOAD") && !strcmp(getenv("GF_BA_UPLOAD"), "copies"));
    A_(h->imu_sqrt.alloc(B * d.W * 225, false)); A_(h->wh_sqrt.alloc(B * d.W * 36, false)); A_(h->pri_A.alloc(B * d.NPRI * d.NPRI, false)); A_(h->pri_b.alloc(B * d.NPRI, false));
    A_(h->pri_c.alloc(B, false)); A_(h->H.alloc(2 * B * d.RP * d.RP, true)); A_(h->g.alloc(2 * B * d.RP, true)); A_(h->Vc.alloc(2 * B * d.NVC, true)); A_(h->wpar.alloc(B * WPAR, true));
    A_(h->cost.alloc(6 * B, true)); A_(h->efac.alloc(B * d.NV * EF, false));
    H_(hipMemsetAsync(h->cost.d, 0, 6 * B * sizeof(double), h->stream));
    A_(h->scale.alloc(B * VS, false)); A_(h->diag.alloc(B * VS, false)); A_(h->grad.alloc(B * VS, false)); A_(h->gn.alloc(B * VS, false)); A_(h->step.alloc(B * VS, false));
    A_(h->u.alloc(B * VS, false)); A_(h->Et.alloc(2 * B * d.FP * d.ECW, true)); A_(h->Es.alloc(B * d.FP * d.ECW, false)); A_(h->ete.alloc(2 * B * d.FP, true)); A_(h->etb.alloc(2 * B * d.FP, true));
    A_(h->rhs.alloc(B * d.RP, false)); A_(h->yv.alloc(B * VS, false));
    if (gnss) { A_(h->ngnss.alloc(B, true)); A_(h->gn_idx.alloc(B * d.NG * 4, true)); A_(h->gn_data.alloc(B * d.NG * GN_STRIDE, true)); A_(h->gn_misc.alloc(B * (GN_MISC + d.NP), true)); A_(h->gn_gptr.alloc(B * (d.NGRP + 2), true)); A_(h->gn_gitem.alloc(B * d.NG, true)); A_(h->gn_rows.alloc(B * d.NG * GN_ROW, false)); }
    for (int m = 0; m < 2; m++) { A_(h->mcolf[m].alloc(B * d.NFB, true)); A_(h->mcole[m].alloc(B * d.F, true)); A_(h->morder[m].alloc(B * d.NVP, true)); A_(h->mnorder[m].alloc(B, true)); A_(h->minfo[m].alloc(B * 4, true)); }
    A_(h->minfo_stage.alloc(B * 4, true));
    A_(h->vpair.alloc(B * (size_t)(d.NP * (d.NP - 1) / 2) * VPG, false));
    A_(h->outJ.alloc(B * (size_t)d.NPRI * d.NPRI, true)); A_(h->outr.alloc(B * d.NPRI, true)); A_(h->stamps.alloc(128, true));
    // kept system of the marginalisation: 6 W poses + speed-bias + extrinsics ...; A and V live in LDS up to 92 columns, else in global memory
    const int nkeep = 6 * d.W + 9 + 17 + (gnss ? 9 : 0);
    h->big_marg = nkeep > 92 || getenv("GF_BA_FORCE_GLOBAL") != nullptr;
    h->meta.assign(d.B, gf_ba::SlotMeta{}); h->outJ_n.assign(d.B, 0); h->imu_dev.assign(d.B, 0); h->imu_rows.assign(d.B, 0);
    A_(h->imu_patch.alloc(B * d.W * IMU_STRIDE2, true)); A_(h->imu_pinfo.alloc(B * (d.W + 2), true));
    for (int mode = 0; mode < 2; mode++) h->keep_ids[mode].assign(d.B, {});
    h->marg_ncap = h->big_marg ? std::min(d.NPRI, nkeep + 16) : 92;
    h->marg_lds = h->big_marg ? 0 : (size_t)2 * h->marg_ncap * h->marg_ncap * sizeof(double);
    if (h->big_step) { h->sg_stride = (h->step_lds / sizeof(double) + 15) & ~(size_t)15; A_(h->Sg.alloc(B * h->sg_stride, false)); }
    if (h->big_marg) { h->mg_stride = (size_t)2 * h->marg_ncap * h->marg_ncap + 1024; A_(h->Mg.alloc(B * h->mg_stride, false)); }
    if (!h->big_step) H_(hipFuncSetAttribute(reinterpret_cast<const void*>(ba_step<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->step_lds));
    // GF_BA_STEP_WAVES=4: ba_step on four wavefronts (half of a CU's registers) so that other kernels' blocks -- the tracker's -- can sit next to it; one setting per
    // process (the reductions' order depends on it: a window alone and the same window in a batch must run the same variant)
    if (getenv("GF_BA_SPLIT_JTJ") && atoi(getenv("GF_BA_SPLIT_JTJ"))) if (int rc = gf_ba_set_split_jtj(h, 1)) { h->release(); delete h; return rc; }
    h->step_waves = (getenv("GF_BA_STEP_WAVES") && atoi(getenv("GF_BA_STEP_WAVES")) == 4) ? 4 : 8;
    h->chain_mode = getenv("GF_BA_CHAIN") && atoi(getenv("GF_BA_CHAIN")) != 0 && !h->big_step && !gnss;
    h->cost_only = !(getenv("GF_BA_COST_ONLY") && atoi(getenv("GF_BA_COST_ONLY")) == 0);
    if (h->chain_mode) {
        const int nd_max = Rmax - 9 * d.NP;
        if (ch_lds_doubles(nd_max) * sizeof(double) + 20 * 1024 > 160 * 1024) h->chain_mode = false;   // (cannot happen where the dense form fits LDS; kept as the guard it is)
        else {
            h->yg_stride = ch_y_stride(nd_max);
            A_(h->Yg.alloc(B * d.NP * h->yg_stride, false));
            H_(hipFuncSetAttribute(reinterpret_cast<const void*>(ba_step_chain), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(ch_lds_doubles(nd_max) * sizeof(double))));
        }
    }
    if (h->step_waves == 4 && !h->big_step) H_(hipFuncSetAttribute(reinterpret_cast<const void*>(ba_step<false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->step_lds));
    // GF_BA_FUSE_MISC=1: the candidate's prior / IMU / wheel sweep in front of the step that judges it, one launch (ba_misc_step).  Same bits; measured 163 us against
    // 123 + 33 us for the two launches and 1.75-1.77 instead of 1.79 ms per solve (-1 %): the sweep's time is its blocks' own latency, not a launch boundary.  Off.
    h->fuse_misc = getenv("GF_BA_FUSE_MISC") != nullptr;
    {   // window-level sweeps: staging areas of the wavefronts (static LDS) + pair tiles / block rows (dynamic LDS, or global memory for long windows)
        const bool glob = getenv("GF_BA_GLOBAL_TILES") != nullptr;   // test switch: pair tiles in global memory also where they fit LDS
        const size_t dyn = vwin_slot_doubles(d.NP, false) * sizeof(double), stat = (size_t)kVW * vwin_sg(false) * vwin_lstr(false) * sizeof(double) + kVW * 64 * sizeof(int) + 512 + 2048;
        h->vwin_lds = (dyn + stat <= 160 * 1024 && !glob) ? dyn : 0;
        if (h->vwin_lds) H_(hipFuncSetAttribute(reinterpret_cast<const void*>(ba_linearize_visual_win<false, kVW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->vwin_lds));
        if (h->vwin_lds) H_(hipFuncSetAttribute(reinterpret_cast<const void*>(ba_linearize_visual_win<false, kVW, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->vwin_lds));
        if (h->vwin_lds) H_(hipFuncSetAttribute(reinterpret_cast<const void*>(ba_linearize_visual_win<false, kVW, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->vwin_lds));
        const size_t dynx = vwin_slot_doubles(d.NP, true) * sizeof(double), statx = (size_t)kVWX * vwin_sg(true) * vwin_lstr(true) * sizeof(double) + kVWX * 64 * sizeof(int) + 512 + 2048;
        h->vwinx_lds = (dynx + statx <= 160 * 1024 && !glob) ? dynx : 0;
        if (h->vwinx_lds) H_(hipFuncSetAttribute(reinterpret_cast<const void*>(ba_linearize_visual_win<true, kVWX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->vwinx_lds));
        h->vwinm_lds = vwin_marg_slot_doubles(d.NP) * sizeof(double);
        H_(hipFuncSetAttribute(reinterpret_cast<const void*>(ba_linearize_visual_win<true, kVWM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->vwinm_lds));
        if (!h->vwin_lds || !h->vwinx_lds) { h->vtile_stride = (vwin_slot_doubles(d.NP, true) + 3) & ~(size_t)3; A_(h->vtile.alloc(B * h->vtile_stride, false)); }
        h->mwin_lds = misc_win_lds_doubles(d.W) * sizeof(double);
        H_(hipFuncSetAttribute(reinterpret_cast<const void*>(ba_linearize_misc_win), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->mwin_lds));
        h->can_fuse_misc = !h->big_step && std::max(h->step_lds, h->mwin_lds) + 27 * 1024 <= 160 * 1024;
        if (h->can_fuse_misc) H_(hipFuncSetAttribute(reinterpret_cast<const void*>(ba_misc_step), hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(h->step_lds, h->mwin_lds)));
    }
    if (!h->big_marg) H_(hipFuncSetAttribute(reinterpret_cast<const void*>(ba_marg_finish<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->marg_lds));
    H_(hipStreamSynchronize(h->stream));
#undef A_
#undef H_
    *out = h;
    return GF_OK;
}

int gf_ba_destroy(gf_ba* h) {
    if (!h) return GF_OK;
    h->release();
    delete h;
    return GF_OK;
}

int gf_ba_upload(gf_ba* h, const gf_ba_window* windows, int count) {
    if (!h || !windows) return gf::set_err(GF_ERR_INVALID, "null argument");
    if (int rc = gf_ba_wait(h)) return rc;
    std::fill(h->outJ_n.begin(), h->outJ_n.end(), 0);   // whole-batch uploads do not track per-slot marginalisation outputs
    if (int rc = pack_windows(h, windows, count)) return rc;
    h->resident.assign(count, nullptr);
    for (int b = 0; b < count; b++) h->resident[b] = windows + b;
    HIPCHK(hipEventRecord(h->ev[0], h->stream));
    if (int rc = upload(h)) return rc;
    HIPCHK(hipEventRecord(h->ev[1], h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, h->ev[0], h->ev[1])); h->stats.ms_upload += ms;
    return GF_OK;
}

int gf_ba_solve_resident_async(gf_ba* h, int max_iters, int marginalize_mode, int reset) {
    if (!h || h->count < 1) return gf::set_err(GF_ERR_INVALID, "no resident windows");
    if (h->pending) return gf::set_err(GF_ERR_INVALID, "a solve is already in flight: call gf_ba_wait first");
    if (max_iters < 0 || max_iters > 64) return gf::set_err(GF_ERR_INVALID, "max_iters out of range");
    if (marginalize_mode > 1) return gf::set_err(GF_ERR_INVALID, "marginalize_mode must be -1, 0 or 1");
    if (reset) { if (int rc = reset_state(h)) return rc; }
    HIPCHK(hipEventRecord(h->ev[0], h->stream));
    if (int rc = run_solve(h, max_iters)) return rc;
    HIPCHK(hipEventRecord(h->ev[1], h->stream));
    if (marginalize_mode >= 0) { if (int rc = run_marginalize(h, marginalize_mode)) return rc; h->last_marg_mode = marginalize_mode; }
    HIPCHK(hipEventRecord(h->ev[4], h->stream));
    h->pending = true; h->pending_iters = max_iters;
    return GF_OK;
}

int gf_ba_wait(gf_ba* h) {
    if (!h) return gf::set_err(GF_ERR_INVALID, "null handle");
    if (!h->pending) return GF_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    h->pending = false;
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, h->ev[0], h->ev[1])); h->stats.ms_solve += ms;
    HIPCHK(hipEventElapsedTime(&ms, h->ev[1], h->ev[4])); h->stats.ms_marginalize += ms;
    if (h->pending_iters > 0) { HIPCHK(hipEventElapsedTime(&ms, h->ev[2], h->ev[3])); h->stats.ms_jtj += ms; }
    if (h->pending_iters > 0 && h->split_timed) { HIPCHK(hipEventElapsedTime(&ms, h->ev_split[0], h->ev_split[1])); h->stats.ms_jtj_contract += ms; h->stats.jtj_contract_launches++; h->split_timed = false; }
    if (h->pending_iters > 0) { HIPCHK(hipEventElapsedTime(&ms, h->ev[6], h->ev[7])); h->stats.ms_step += ms; }
    return GF_OK;
}

int gf_ba_solve_resident(gf_ba* h, int max_iters, int marginalize_mode, int reset) {
    if (int rc = gf_ba_solve_resident_async(h, max_iters, marginalize_mode, reset)) return rc;
    return gf_ba_wait(h);
}


int gf_ba_download(gf_ba* h, gf_ba_window* windows, int count, gf_ba_summary* summaries, gf_ba_prior* priors) {
    if (!h || count < 1 || count > h->count) return gf::set_err(GF_ERR_INVALID, "bad argument");
    if (int rc = gf_ba_wait(h)) return rc;
    const Dims& d = h->d;
    if (priors) {
        if (h->last_marg_mode < 0) return gf::set_err(GF_ERR_INVALID, "no marginalisation has been run on the resident windows");
        HIPCHK(h->outJ.down(h->stream)); HIPCHK(h->outr.down(h->stream));
    }
    HIPCHK(hipEventRecord(h->ev[0], h->stream));
    HIPCHK(h->xs.down(h->stream));
    HIPCHK(hipMemcpyAsync(h->st.h, h->st.d, (size_t)d.B * sizeof(SolverState), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipEventRecord(h->ev[1], h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, h->ev[0], h->ev[1])); h->stats.ms_download += ms;
    for (int b = 0; b < count; b++) {
        const SolverState& st = h->st.h[b];
        unpack_state(h, b, windows ? windows + b : nullptr, summaries ? summaries + b : nullptr);
        if (priors) {
            gf_ba_prior& p = priors[b];
            const int mode = h->last_marg_mode;
            const int* inf = h->minfo[mode].h + (size_t)b * 4;
            p.valid = inf[3]; p.m = inf[0] + inf[1]; p.n = 0; p.nblocks = 0;
            if (inf[3]) {
                const int n = inf[2];
                const std::vector<int>& keep = h->keep_ids[mode][b];
                if (n > p.cap_n || (int)keep.size() > p.cap_blocks) return gf::set_err(GF_ERR_CAPACITY, "prior capacity too small (n=%d, blocks=%zu)", n, keep.size());
                p.n = n; p.nblocks = (int)keep.size();
                memcpy(p.J, h->outJ.h + (size_t)b * d.NPRI * d.NPRI, (size_t)n * n * sizeof(double));
                memcpy(p.r, h->outr.h + (size_t)b * d.NPRI, (size_t)n * sizeof(double));
                const double* x = h->xs.h + ((size_t)st.cur * d.B + b) * d.XS;
                int xo = 0;
                for (size_t q = 0; q < keep.size(); q++) {
                    const int id = keep[q], kind = id / 4096, i = id % 4096;
                    int nid = id;  // addr_shift (estimator.cpp:3471-3500 / :3583-3626)
                    if (kind == GF_POSE || kind == GF_SPEEDBIAS || kind == GF_RCV_DDT) nid = mode == 0 ? kind * 4096 + i - 1 : (i == d.W ? kind * 4096 + d.W - 1 : id);
                    else if (kind == GF_RCV_DT) nid = mode == 0 ? id - 4 : (i / 4 == d.W ? id - 4 : id);
                    p.block_id[q] = nid;
                    int off;
                    switch (kind) { case 0: off = off_pose(i); break; case 1: off = off_sb(i); break; case 2: off = off_ex(d.NP); break; case 3: off = off_exw(d.NP); break;
                                    case 4: off = off_ix(d.NP); break; case 5: off = off_ix(d.NP) + 1; break; case 6: off = off_ix(d.NP) + 2; break; case 7: off = off_td(d.NP); break;
                                    case 10: off = d.GO + i; break; case 11: off = d.GO + 4 * d.NP + i; break; case 12: off = d.GO + 5 * d.NP; break; case 13: off = d.GO + 5 * d.NP + 1; break;
                                    default: off = off_tdw(d.NP); }
                    for (int k = 0; k < gsize_kind(kind); k++) p.x0[xo++] = x[off + k];
                }
            }
        }
    }
    return GF_OK;
}

// ---- packed slots: callers that own many windows on many threads (gf_estimator_group) pack every window on its owner's thread, one call closes the
// batch (upload, solve, one download of all states), and every owner unpacks its own slot again.  The serial part of a batched solve shrinks to the
// transfers and the kernels (packing 256 windows on one call's 16 threads took longer than solving them).
int gf_ba_pack_slot(gf_ba* h, int slot, const gf_ba_window* window) {
    if (!h || !window || slot < 0 || slot >= h->d.B) return gf::set_err(GF_ERR_INVALID, "bad argument (0 <= slot < batch)");
    if (h->pending) return gf::set_err(GF_ERR_INVALID, "a solve is in flight: call gf_ba_wait first");
    return pack_slot(h, slot, *window);
}
int gf_ba_solve_packed(gf_ba* h, const int* slots, int n, int max_iters) {
    if (!h || !slots || n < 1 || n > h->d.B) return gf::set_err(GF_ERR_INVALID, "bad argument");
    if (int rc = gf_ba_wait(h)) return rc;
    const Dims& d = h->d;
    std::vector<char> on(d.B, 0);
    for (int q = 0; q < n; q++) { if (slots[q] < 0 || slots[q] >= d.B || on[slots[q]]) return gf::set_err(GF_ERR_INVALID, "slot %d out of range or listed twice", slots[q]); on[slots[q]] = 1; }
    for (int b = 0; b < d.B; b++)
        if (!on[b]) h->st0.h[b].done = 1;   // slots that sit this batch out: the solver kernels leave finished windows alone (their tables, the marginalisation
                                            // layouts included, stay what their owner packed last: a marginalisation of such a slot may still follow)
    finalize_pack(h, slots, n, d.B);
    h->resident.assign(d.B, nullptr);
    HIPCHK(hipEventRecord(h->ev[8], h->stream));
    if (int rc = upload(h)) return rc;
    HIPCHK(hipEventRecord(h->ev[9], h->stream));
    if (int rc = gf_ba_solve_resident_async(h, max_iters, -1, 1)) return rc;
    HIPCHK(h->xs.down(h->stream));
    HIPCHK(hipMemcpyAsync(h->st.h, h->st.d, (size_t)d.B * sizeof(SolverState), hipMemcpyDeviceToHost, h->stream));
    if (int rc = gf_ba_wait(h)) return rc;
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, h->ev[8], h->ev[9])); h->stats.ms_upload += ms;
    return GF_OK;
}
int gf_ba_unpack_slot(gf_ba* h, int slot, gf_ba_window* window, gf_ba_summary* summary) {
    if (!h || slot < 0 || slot >= h->d.B) return gf::set_err(GF_ERR_INVALID, "bad argument");
    if (h->pending) return gf::set_err(GF_ERR_INVALID, "a solve is in flight: call gf_ba_wait first");
    unpack_state(h, slot, window, summary);
    return GF_OK;
}

int gf_ba_solve(gf_ba* h, gf_ba_window* windows, int count, int max_iters, gf_ba_summary* summaries) {
    if (int rc = gf_ba_upload(h, windows, count)) return rc;
    if (int rc = gf_ba_solve_resident(h, max_iters, -1, 1)) return rc;
    return gf_ba_download(h, windows, count, summaries, nullptr);
}

int gf_ba_marginalize(gf_ba* h, const gf_ba_window* windows, int count, int mode, gf_ba_prior* priors) {
    if (!h || !windows || !priors || mode < 0 || mode > 1) return gf::set_err(GF_ERR_INVALID, "bad argument");
    if (int rc = gf_ba_upload(h, windows, count)) return rc;
    if (int rc = gf_ba_solve_resident(h, 0, mode, 1)) return rc;
    return gf_ba_download(h, nullptr, count, nullptr, priors);
}

// MARGIN_OLD / MARGIN_SECOND_NEW on windows that are already resident (the preceding gf_ba_solve / gf_ba_upload of the same structures): only
// the parameter blocks are packed and uploaded again -- Estimator::optimization() runs double2vector (the gauge fix) and vector2double between
// ceres::Solve and the marginalisation (estimator.cpp:3327, :3337), so the states differ from what the solve left on the device, the factors do
// not.  slots[i] = position of windows[i] in the resident batch.  Slots not listed are marginalised too (the kernels run on the whole batch);
// their results are not fetched.
int gf_ba_marginalize_resident(gf_ba* h, const int* slots, const gf_ba_window* windows, int n, int mode, gf_ba_prior* priors) {   // priors == NULL: fetch them with gf_ba_unpack_prior_slot
    if (!h || !slots || !windows || n < 1 || mode < 0 || mode > 1) return gf::set_err(GF_ERR_INVALID, "bad argument");
    if (int rc = gf_ba_wait(h)) return rc;
    const Dims& d = h->d;
    for (int i = 0; i < n; i++) {
        const int b = slots[i];
        if (b < 0 || b >= h->count) return gf::set_err(GF_ERR_INVALID, "slot %d outside the %d resident windows", b, h->count);
        const gf_ba_window& w = windows[i];
        // the factor tables, orders and the prior of slot b stay as the last upload / solve left them: the window handed in must be that one
        if (w.W != d.W || w.n_feature != h->nfeat.h[b] || w.n_visual != h->nvis.h[b] || w.n_imu != h->nimu.h[b] || w.n_wheel != h->nwh.h[b] || w.prior_n != h->pri_n.h[b])
            return gf::set_err(GF_ERR_INVALID, "window %d does not match the structure resident in slot %d (features %d/%d, visual %d/%d, imu %d/%d, wheel %d/%d, prior %d/%d)", i, b,
                               w.n_feature, h->nfeat.h[b], w.n_visual, h->nvis.h[b], w.n_imu, h->nimu.h[b], w.n_wheel, h->nwh.h[b], w.prior_n, h->pri_n.h[b]);
        double* x = h->xs0.h + (size_t)b * d.XS;
        for (int k = 0; k < d.NP; k++) { memcpy(x + off_pose(k), w.para_Pose + 7 * k, 56); memcpy(x + off_sb(k), w.para_SpeedBias + 9 * k, 72); }
        memcpy(x + off_ex(d.NP), w.para_Ex_Pose, 56); memcpy(x + off_exw(d.NP), w.para_Ex_Pose_wheel, 56); memcpy(x + off_ix(d.NP), w.para_Ix, 24);
        x[off_td(d.NP)] = w.para_Td[0]; x[off_tdw(d.NP)] = w.para_Td_wheel[0];
        for (int f = 0; f < w.n_feature; f++) x[off_feat(d.NP) + f] = w.para_Feature[f];
        if (d.GO && w.gnss_enabled) { memcpy(x + d.GO, w.para_rcv_dt, 4 * d.NP * 8); memcpy(x + d.GO + 4 * d.NP, w.para_rcv_ddt, d.NP * 8); x[d.GO + 5 * d.NP] = w.para_yaw_enu_local[0]; memcpy(x + d.GO + 5 * d.NP + 1, w.para_anc_ecef, 24); }
        if (w.fix_poses) for (int k = 0; k < d.NP; k++) x[off_sb(k)] = x[off_sb(k) + 1] = x[off_sb(k) + 2] = 0.0;
    }
    HIPCHK(h->xs0.up(h->stream));
    {   // only the listed slots are marginalised: the others keep whatever prior their last marginalisation left in the output buffers (their owners may
        // not have fetched it yet -- two marginalisation kinds of one batch run back to back), so the kernels see them as invalid for this launch
        std::vector<char> on(d.B, 0);
        for (int i = 0; i < n; i++) on[slots[i]] = 1;
        for (int b = 0; b < d.B; b++) { for (int q = 0; q < 4; q++) h->minfo_stage.h[(size_t)b * 4 + q] = h->minfo[mode].h[(size_t)b * 4 + q]; if (!on[b]) h->minfo_stage.h[(size_t)b * 4 + 3] = 0; }
        HIPCHK(hipMemcpyAsync(h->minfo[mode].d, h->minfo_stage.h, (size_t)d.B * 4 * sizeof(int), hipMemcpyHostToDevice, h->stream));
    }
    if (int rc = gf_ba_solve_resident(h, 0, mode, 1)) return rc;
    // priors of the listed slots: n x n of J, n of r (the rest of the capacity-sized slots stays on the device)
    const int* inf0 = h->minfo[mode].h;
    int nmax = 0;
    for (int i = 0; i < n; i++) if (inf0[(size_t)slots[i] * 4 + 3]) nmax = std::max(nmax, inf0[(size_t)slots[i] * 4 + 2]);
    for (int i = 0; i < n; i++) h->outJ_n[slots[i]] = inf0[(size_t)slots[i] * 4 + 3] ? inf0[(size_t)slots[i] * 4 + 2] : 0;
    h->outJ_host_stale = priors == nullptr;   // the owners take J from the device (resident priors) or fetch it one by one (gf_ba_unpack_prior_slot with a J buffer)
    if (nmax > 0) {
        if (priors) HIPCHK(hipMemcpy2DAsync(h->outJ.h, (size_t)d.NPRI * d.NPRI * 8, h->outJ.d, (size_t)d.NPRI * d.NPRI * 8, (size_t)nmax * nmax * 8, h->count, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpy2DAsync(h->outr.h, (size_t)d.NPRI * 8, h->outr.d, (size_t)d.NPRI * 8, (size_t)nmax * 8, h->count, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    h->last_marg_mode = mode;
    if (priors) for (int i = 0; i < n; i++) if (int rc = gf_ba_unpack_prior_slot(h, slots[i], mode, &priors[i])) return rc;
    return GF_OK;
}

// the prior gf_ba_marginalize_resident left for slot `slot` in the host mirrors (kept block ids after the address shift, J, r, linearisation point): callable
// concurrently for different slots -- the owners of the windows copy their own 60 KB each instead of one thread copying 15 MB for all of them
int gf_ba_unpack_prior_slot(gf_ba* h, int slot, int mode, gf_ba_prior* prior) {
    if (!h || !prior || slot < 0 || slot >= h->d.B || mode < 0 || mode > 1) return gf::set_err(GF_ERR_INVALID, "bad argument");
    const Dims& d = h->d;
    const int b = slot;
    gf_ba_prior& p = *prior;
    {
        const int* inf = h->minfo[mode].h + (size_t)b * 4;
        p.valid = inf[3]; p.m = inf[0] + inf[1]; p.n = 0; p.nblocks = 0;
        if (!inf[3]) return GF_OK;
        const int nn = inf[2];
        const std::vector<int>& keep = h->keep_ids[mode][b];
        if (nn > p.cap_n || (int)keep.size() > p.cap_blocks) return gf::set_err(GF_ERR_CAPACITY, "prior capacity too small (n=%d, blocks=%zu)", nn, keep.size());
        p.n = nn; p.nblocks = (int)keep.size();
        if (p.J) {   // J == NULL: the prior's J stays on the device (the slot's next gf_ba_pack_slot passes prior_J = NULL)
            if (h->outJ_host_stale) HIPCHK(hipMemcpy(h->outJ.h + (size_t)b * d.NPRI * d.NPRI, h->outJ.d + (size_t)b * d.NPRI * d.NPRI, (size_t)nn * nn * sizeof(double), hipMemcpyDeviceToHost));
            memcpy(p.J, h->outJ.h + (size_t)b * d.NPRI * d.NPRI, (size_t)nn * nn * sizeof(double));
        }
        memcpy(p.r, h->outr.h + (size_t)b * d.NPRI, (size_t)nn * sizeof(double));
        const double* x = h->xs0.h + (size_t)b * d.XS;   // the linearisation point of the prior = the states just uploaded
        int xo = 0;
        for (size_t q = 0; q < keep.size(); q++) {
            const int id = keep[q], kind = id / 4096, k = id % 4096;
            int nid = id;  // addr_shift (estimator.cpp:3471-3500 / :3583-3626)
            if (kind == GF_POSE || kind == GF_SPEEDBIAS || kind == GF_RCV_DDT) nid = mode == 0 ? kind * 4096 + k - 1 : (k == d.W ? kind * 4096 + d.W - 1 : id);
            else if (kind == GF_RCV_DT) nid = mode == 0 ? id - 4 : (k / 4 == d.W ? id - 4 : id);
            p.block_id[q] = nid;
            int off;
            switch (kind) { case 0: off = off_pose(k); break; case 1: off = off_sb(k); break; case 2: off = off_ex(d.NP); break; case 3: off = off_exw(d.NP); break;
                            case 4: off = off_ix(d.NP); break; case 5: off = off_ix(d.NP) + 1; break; case 6: off = off_ix(d.NP) + 2; break; case 7: off = off_td(d.NP); break;
                            case 10: off = d.GO + k; break; case 11: off = d.GO + 4 * d.NP + k; break; case 12: off = d.GO + 5 * d.NP; break; case 13: off = d.GO + 5 * d.NP + 1; break;
                            default: off = off_tdw(d.NP); }
            for (int c = 0; c < gsize_kind(kind); c++) p.x0[xo++] = x[off + c];
        }
    }
    return GF_OK;
}

int gf_ba_debug_upload_bytes(long long* bytes, long long* calls) { if (bytes) *bytes = g_up_bytes; if (calls) *calls = g_up_calls; return GF_OK; }
int gf_ba_fetch_resident_prior(gf_ba* h, int slot, int n, double* J) {
    if (!h || !J || slot < 0 || slot >= h->d.B) return gf::set_err(GF_ERR_INVALID, "bad argument");
    if (n < 1 || h->outJ_n[slot] != n) return gf::set_err(GF_ERR_INVALID, "slot %d holds a marginalisation output of %d columns, not %d", slot, h->outJ_n[slot], n);
    if (int rc = gf_ba_wait(h)) return rc;
    HIPCHK(hipMemcpy(J, h->outJ.d + (size_t)slot * h->d.NPRI * h->d.NPRI, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost));
    return GF_OK;
}

int gf_ba_linearize(gf_ba* h, const gf_ba_window* w, int cap, double* Hout, double* gout, double* cost, int* n_f, int* n_e, int* col_block_id) {
    if (!h || !w || !Hout || !gout || !cost) return gf::set_err(GF_ERR_INVALID, "null argument");
    if (int rc = gf_ba_upload(h, w, 1)) return rc;
    if (int rc = reset_state(h)) return rc;
    const Dims& d = h->d;
    if (int rc = launch_linearize(h, 0, 0, 0, false)) return rc;
    HIPCHK(h->H.down(h->stream)); HIPCHK(h->g.down(h->stream)); HIPCHK(h->Vc.down(h->stream)); HIPCHK(h->cost.down(h->stream)); HIPCHK(h->Et.down(h->stream)); HIPCHK(h->ete.down(h->stream)); HIPCHK(h->etb.down(h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    {   // H, g <- H + Vc, g + Vc's right-hand-side row: what ba_step assembles
        const int RHSK = 6 * d.NP + 7;
        auto ccol = [&](int k) -> int { const int* cf = h->colf.h; if (k < 6 * d.NP) { const int c0 = cf[fb_pose(k / 6)]; return c0 >= 0 ? c0 + k % 6 : -1; } if (k < 6 * d.NP + 6) { const int c0 = cf[fb_ex(d.NP)]; return c0 >= 0 ? c0 + k - 6 * d.NP : -1; } return cf[fb_td(d.NP)]; };
        for (int ka = 0; ka < RHSK; ka++) for (int kb = 0; kb <= ka; kb++) { const int r = ccol(ka), c = ccol(kb); if (r >= 0 && c >= 0) h->H.h[(size_t)std::max(r, c) * d.RP + std::min(r, c)] += h->Vc.h[(size_t)ka * (ka + 1) / 2 + kb]; }
        for (int kb = 0; kb < RHSK; kb++) { const int c = ccol(kb); if (c >= 0) h->g.h[c] += h->Vc.h[(size_t)RHSK * (RHSK + 1) / 2 + kb]; }
    }
    const SolverState& st = h->st0.h[0];
    const int R = st.R, NE = st.NE, n = R + NE;
    if (n > cap) return gf::set_err(GF_ERR_CAPACITY, "capacity %d < %d columns", cap, n);
    for (int i = 0; i < n * n; i++) Hout[i] = 0;
    for (int r = 0; r < R; r++) { for (int c = 0; c <= r; c++) Hout[(size_t)r * n + c] = Hout[(size_t)c * n + r] = h->H.h[(size_t)r * d.RP + c]; gout[r] = h->g.h[r]; }   // device H holds the lower triangle
    for (int e = 0; e < NE; e++) {
        Hout[(size_t)(R + e) * n + R + e] = h->ete.h[e]; gout[R + e] = h->etb.h[e];
        for (int k = 0; k < 6 * d.NP + 7; k++) {   // compact row -> reduced columns
            const int* cf = h->colf.h;
            int c = -1;
            if (k < 6 * d.NP) { if (cf[fb_pose(k / 6)] >= 0) c = cf[fb_pose(k / 6)] + k % 6; }
            else if (k < 6 * d.NP + 6) { if (cf[fb_ex(d.NP)] >= 0) c = cf[fb_ex(d.NP)] + k - 6 * d.NP; }
            else c = cf[fb_td(d.NP)];
            if (c >= 0) { Hout[(size_t)(R + e) * n + c] = h->Et.h[(size_t)e * d.ECW + k]; Hout[(size_t)c * n + R + e] = h->Et.h[(size_t)e * d.ECW + k]; }
        }
    }
    *cost = (h->cost.h[0] + h->cost.h[2 * (size_t)d.B]) + h->cost.h[4 * (size_t)d.B]; *n_f = R; *n_e = NE;
    if (col_block_id) {
        const int* cf = h->colf.h;
        for (int i = 0; i < d.NP; i++) { if (cf[fb_pose(i)] >= 0) for (int q = 0; q < 6; q++) col_block_id[cf[fb_pose(i)] + q] = GF_POSE * 4096 + i; if (cf[fb_sb(i)] >= 0) for (int q = 0; q < 9; q++) col_block_id[cf[fb_sb(i)] + q] = GF_SPEEDBIAS * 4096 + i; }
        const int kinds[7] = {GF_EX_POSE, GF_EX_WHEEL, GF_SX, GF_SY, GF_SW, GF_TD, GF_TD_WHEEL};
        for (int q = 0; q < 7; q++) { const int c0 = cf[2 * d.NP + q]; if (c0 >= 0) for (int k = 0; k < (q < 2 ? 6 : 1); k++) col_block_id[c0 + k] = kinds[q] * 4096; }
        if (d.GO) {
            for (int q = 0; q < 4 * d.NP; q++) if (cf[fb_rcvdt(d.NP, q)] >= 0) col_block_id[cf[fb_rcvdt(d.NP, q)]] = GF_RCV_DT * 4096 + q;
            for (int i = 0; i < d.NP; i++) if (cf[fb_rcvddt(d.NP, i)] >= 0) col_block_id[cf[fb_rcvddt(d.NP, i)]] = GF_RCV_DDT * 4096 + i;
            if (cf[fb_anc(d.NP)] >= 0) for (int k = 0; k < 3; k++) col_block_id[cf[fb_anc(d.NP)] + k] = GF_ANC * 4096;
        }
        for (int f = 0; f < d.F; f++) if (h->cole.h[f] >= 0) col_block_id[R + h->cole.h[f]] = GF_FEATURE * 4096 + f;
    }
    return GF_OK;
}

// north_star's pose exchange: the newest pose (px py pz qx qy qz qw) of every resident window, straight from the current state buffer into a
// caller-provided DEVICE array [count][7] (what the RCCL all_gather of bench.py / shard.py sends); returns when the copy has completed.
__global__ void ba_export_newest(gfb::Win w, double* out, int count) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= count) return;
    const double* x = w.xs + ((size_t)w.st[b].cur * w.d.B + b) * w.d.XS + gfb::off_pose(w.d.W);
    for (int q = 0; q < 7; q++) out[(size_t)b * 7 + q] = x[q];
}
int gf_ba_export_newest_poses(gf_ba* h, void* d_out, int count) {
    if (!h || !d_out || count < 1 || count > h->count) return gf::set_err(GF_ERR_INVALID, "bad argument");
    ba_export_newest<<<dim3((count + 63) / 64), 64, 0, h->stream>>>(h->win(), static_cast<double*>(d_out), count);
    HIPCHK(hipGetLastError());
    // Always synchronise: the caller hands d_out to a collective on ANOTHER stream right after this returns.  (With a solve in flight this also
    // waits for the solve -- the exported pose is the solved one; gf_ba_wait afterwards only collects the statistics.)
    HIPCHK(hipStreamSynchronize(h->stream));
    return GF_OK;
}

// north_star's exchange step as one C call: newest poses of this rank's `count` resident windows -> ncclAllGather -> [world][count][7] on every rank.
int gf_pose_gather(gf_ba* h, void* nccl_comm, void* stream, int count, double* d_out) {
    if (!h || !nccl_comm || !d_out || count < 1 || count > h->d.B) return gf::set_err(GF_ERR_INVALID, "bad argument");
    if (!h->gather_send.d) { g_alloc_what = "gather_send.alloc (first gf_pose_gather)"; if (int rc = h->gather_send.alloc((size_t)h->d.B * 7, false)) return rc; HIPCHK(hipMemsetAsync(h->gather_send.d, 0, (size_t)h->d.B * 7 * sizeof(double), h->stream)); }
    // the send buffer is persistent: the export of THIS call must not overtake the collective of the previous one, which reads it on the caller's stream
    if (h->gather_in_flight) { HIPCHK(hipStreamWaitEvent(h->stream, h->ev_gather_done, 0)); h->gather_in_flight = false; }
    // ranks must pass the same count (ncclAllGather): a rank with fewer resident windows sends zero rows behind its own
    const int mine = std::min(count, h->count);
    if (mine > 0) { ba_export_newest<<<dim3((mine + 63) / 64), 64, 0, h->stream>>>(h->win(), h->gather_send.d, mine); HIPCHK(hipGetLastError()); }
    if (mine < count) HIPCHK(hipMemsetAsync(h->gather_send.d + (size_t)mine * 7, 0, (size_t)(count - mine) * 7 * sizeof(double), h->stream));
    if (stream != (void*)h->stream) {   // order the collective behind the export without stalling the host: an event on the solver's stream
        HIPCHK(hipEventRecord(h->ev_gather, h->stream));
        HIPCHK(hipStreamWaitEvent(static_cast<hipStream_t>(stream), h->ev_gather, 0));
    }
    if (int rc = gf::rccl_allgather_f64(h->gather_send.d, d_out, (size_t)count * 7, nccl_comm, stream)) return rc;
    if (stream != (void*)h->stream) { HIPCHK(hipEventRecord(h->ev_gather_done, static_cast<hipStream_t>(stream))); h->gather_in_flight = true; }
    return GF_OK;
}

int gf_ba_set_split_jtj(gf_ba* h, int on) {
    if (!h) return gf::set_err(GF_ERR_INVALID, "null handle");
    if (on && !h->vrows.d) {
        if (int rc = h->vrows.alloc((size_t)h->d.B * ((h->d.NVP + 63) & ~63) * 32 + 64 * 32, false)) return rc;
        for (auto& e : h->ev_split) HIPCHK(hipEventCreate(&e));
    }
    h->split_jtj = on != 0;
    return GF_OK;
}

int gf_ba_debug_stamps(gf_ba* h, long long* out, int n) {  // phase timestamps of the last ba_step launch (profiling builds)
    if (!h || !out || n > 128) return gf::set_err(GF_ERR_INVALID, "bad argument");
    HIPCHK(hipMemcpy(out, h->stamps.d, n * sizeof(long long), hipMemcpyDeviceToHost));
    return GF_OK;
}

int gf_ba_set_max_solver_time(gf_ba* h, double seconds) {
    if (!h || !(seconds >= 0.0)) return gf::set_err(GF_ERR_INVALID, "bad argument");
    h->max_solver_time = seconds;
    return GF_OK;
}
int gf_ba_get_stats(gf_ba* h, gf_ba_stats* out) { if (!h || !out) return gf::set_err(GF_ERR_INVALID, "null argument"); *out = h->stats; return GF_OK; }
int gf_ba_reset_stats(gf_ba* h) { if (!h) return gf::set_err(GF_ERR_INVALID, "null handle"); h->stats = gf_ba_stats{}; return GF_OK; }

}  // extern "C"
