// gf_detect_kernels.hpp — gfx950 device code for the Shi-Tomasi top-up of the front end:
//   setMask's disk rasterisation          (feature_tracker.cpp:56-83, cv::circle filled, radius MIN_DIST)
//   cv::goodFeaturesToTrack(img, n_pts, MAX_CNT-N, 0.01, MIN_DIST, mask)   (feature_tracker.cpp:198)
// following OpenCV 4.2 featureselect.cpp / corner.cpp / drawing.cpp (see DESIGN.md "arithmetic choices").
// Compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gf_lk_kernels.hpp"

namespace gf {

constexpr int kMaxRadius = 128;  // largest MIN_DIST supported by the disk table

struct DiskTable { short hw[kMaxRadius + 1]; int radius; };  // half-width of row |dy| of OpenCV's filled midpoint circle

// cornerMinEigenVal(block 3, Sobel 3): 32x8 output tile per 256-thread block, 34x10 covariance halo in LDS
// evaluated at REFLECT_101 coordinates (the box filter's border mode), image read from the padded level 0.
__global__ void __launch_bounds__(256) min_eig_kernel(const uint8_t* __restrict__ pyr, size_t pyr_seq_stride, LevelGeom g,
                                                      float* __restrict__ eig, size_t eig_seq_stride) {
    __shared__ float cxx[10][36], cxy[10][36], cyy[10][36];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 8, b = blockIdx.z;
    const uint8_t* img = pyr + b * pyr_seq_stride + g.img_off;
    const float f1 = (float)(1.0 * (1.0 / (4.0 * 3.0 * 255.0))), f0 = (float)(2.0 * (1.0 / (4.0 * 3.0 * 255.0)));
    for (int t = threadIdx.x; t < 34 * 10; t += 256) {
        const int ty = t / 34, tx = t - ty * 34;
        const int x = reflect101(bx + tx - 1, g.w), y = reflect101(by + ty - 1, g.h);
        const uint8_t* p = img + (size_t)y * g.stride + x;
        const int a0 = p[-g.stride - 1], a1 = p[-g.stride], a2 = p[-g.stride + 1];
        const int m0 = p[-1], m2 = p[1];
        const int c0 = p[g.stride - 1], c1 = p[g.stride], c2 = p[g.stride + 1];
        const float t0 = (float)(a2 - a0), t1 = (float)(m2 - m0), t2 = (float)(c2 - c0);
        const float dx = (t0 + t2) * f1 + t1 * f0;
        float rt = f1 * (float)a0; rt += f0 * (float)a1; rt += f1 * (float)a2;
        float rb = f1 * (float)c0; rb += f0 * (float)c1; rb += f1 * (float)c2;
        const float dy = rb - rt;
        cxx[ty][tx] = dx * dx; cxy[ty][tx] = dx * dy; cyy[ty][tx] = dy * dy;
    }
    __syncthreads();
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int x = bx + lx, y = by + ly;
    if (x >= g.w || y >= g.h) return;
    double sa = 0, sb = 0, sc = 0;
#pragma unroll
    for (int dy = 0; dy < 3; dy++)
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            sa += (double)cxx[ly + dy][lx + dx];
            sb += (double)cxy[ly + dy][lx + dx];
            sc += (double)cyy[ly + dy][lx + dx];
        }
    const float a = (float)sa * 0.5f, bb = (float)sb, c = (float)sc * 0.5f;
    eig[b * eig_seq_stride + (size_t)y * g.w + x] = (a + c) - sqrtf((a - c) * (a - c) + bb * bb);
}

__device__ __forceinline__ unsigned f32_orderable(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float f32_from_orderable(unsigned k) {
    unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; __builtin_memcpy(&f, &u, 4); return f;
#endif
}


// ---------------------------------------------------------------------------------------------
// Fused Shi-Tomasi pass: image tile -> covariance (LDS) -> min-eigenvalue (LDS) -> masked maximum and
// 3x3 local-maximum candidates.  The response image never touches HBM.  The mask is either an explicit
// u8 image (building-block entry point) or, in trackImage, the union of filled circles of radius MIN_DIST
// around the kept points, tested analytically with OpenCV's midpoint-circle row table (no rasterised mask).
struct DetectArgs {
    const uint8_t* pyr; size_t pyr_seq_stride; LevelGeom g;
    const uint8_t* mask; size_t mask_seq_stride;            // optional explicit mask (non-zero = allowed)
    const int2* centers; const int* n_centers; int cap;     // else: disks
    const int* want;                                        // [batch] skip sequences that need no new corners
    unsigned* maxkey;                                       // [batch] orderable max over unmasked pixels (0 = none)
    unsigned long long* cand; size_t cand_seq_stride; int cand_cap; int* cand_count;
};

// One wavefront walks a strip of R image rows top to bottom, lane l on column X0 - 2 + l, and keeps everything
// a 3 x 3 neighbourhood needs in registers -- the horizontal parts of the Sobel rows of the last three raw rows, the horizontal box sums of the last three
// product rows (double, exact), the last three eigenvalue rows and their horizontal maxima.  Horizontal neighbours come over DPP wavefront shifts; lanes 2..61
// (60 columns) and rows Y0 .. Y0 + R - 1 are outputs, the rest is halo recomputed by the neighbouring strips.  Per pixel the arithmetic is the one of
// min_eig_kernel above (same operations in the same order; the box sums are exact -- nine float products spanning < 2^52 -- so their order is free).
constexpr int kDS_W = 60;   // output columns per wavefront
constexpr int kDS_R = 30;   // rows per strip in trackImage: 36 raw rows for 30 output rows (16: 0.31 ms, 30: 0.27 ms, 32: 0.27 ms per 256 VGA frames)
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
__device__ __forceinline__ float dpp_from_left(float v) {   // lane l <- lane l - 1
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, true));   // lane 0 reads 0: halo, never used
}
__device__ __forceinline__ float dpp_from_right(float v) {  // lane l <- lane l + 1
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}

// the strip's staged candidates -> the sequence's list (one atomic per call); rare inside a strip, so kept out of line
__device__ __noinline__ void strip_flush(unsigned long long* cand, int* count, int cap, const unsigned long long* ckeys, int n, int lane) {
    int base = 0;
    if (lane == 0) base = atomicAdd(count, n);
    base = __builtin_amdgcn_readfirstlane(base);
    for (int k = lane; k < n; k += 64)
        if (base + k < cap) cand[base + k] = ckeys[k];
}

template <int R>
__global__ void __launch_bounds__(64) detect_strip_kernel(DetectArgs A, DiskTable T) {
    static_assert(R >= 1 && R <= 32, "one allow bit per strip row");
    constexpr int NS = R + 6;   // raw rows Y0 - 3 .. Y0 + R + 2
    constexpr int kCap = 320;   // candidates staged per strip (3 x 3 maxima exclude their neighbours: <= 60 R / 4 without plateaus); flushed early when a row might not fit
    __shared__ unsigned long long ckeys[kCap];
    __shared__ short s_hw[kMaxRadius + 1];
    const int b = blockIdx.z;
    if (A.want[b] <= 0) return;
    const LevelGeom g = A.g;
    const int lane = threadIdx.x;
    const int X0 = blockIdx.x * kDS_W, Y0 = blockIdx.y * R;
    const int x = X0 - 2 + lane;
    const bool col_out = lane >= 2 && lane < 62 && x < g.w;
    const int nrows = min(R, g.h - Y0);
    const uint32_t rows_all = nrows >= 32 ? ~0u : (1u << nrows) - 1u;
    // ---- setMask for this strip: bit r of `allow` = pixel (x, Y0 + r) lies in no kept point's filled circle.  The circle table is symmetric in dx <-> dy and
    // monotone (midpoint circle), so a centre covers a contiguous run of rows of one column: |y - cy| <= hw[|x - cx|].
    uint32_t allow = 0;
    if (A.mask) {
        if (col_out)
            for (int r = 0; r < nrows; r++)
                if (A.mask[b * A.mask_seq_stride + (size_t)(Y0 + r) * g.w + x]) allow |= 1u << r;
    } else {
#pragma unroll
        for (int q = 0; q < (kMaxRadius + 64) / 64; q++)
            if (lane + 64 * q <= T.radius) s_hw[lane + 64 * q] = T.hw[lane + 64 * q];
        __syncthreads();
        uint32_t covered = 0;
        const int n = A.n_centers[b];
        for (int i0 = 0; i0 < n; i0 += 64) {
            int2 c = make_int2(-(1 << 20), -(1 << 20));
            if (i0 + lane < n) c = A.centers[(size_t)b * A.cap + i0 + lane];
            const bool hit = c.x + T.radius >= X0 && c.x - T.radius < X0 + kDS_W && c.y + T.radius >= Y0 && c.y - T.radius < Y0 + R;
            unsigned long long hb = __ballot(hit);
            while (hb) {
                const int j = __builtin_ctzll(hb);
                hb &= hb - 1;
                const int cx = __builtin_amdgcn_readlane(c.x, j), cy = __builtin_amdgcn_readlane(c.y, j);
                const int dx = abs(x - cx);
                if (dx <= T.radius) {
                    const int k = s_hw[dx];
                    const int lo = max(cy - k - Y0, 0), hi = min(cy + k - Y0, R - 1);
                    if (lo <= hi) covered |= (hi - lo >= 31 ? ~0u : ((1u << (hi - lo + 1)) - 1u)) << lo;
                }
            }
        }
        if (col_out) allow = ~covered & rows_all;
    }
    if (!__ballot(allow != 0)) return;   // nothing of this strip may hold a corner: no contribution to the masked maximum, no candidate
    // Which rows the strip has to work on at all.  With the window's tracks alive the circles cover most of the image (150 points at MIN_DIST 30: ~90 %), and an output
    // row none of whose pixels is allowed needs neither its own eigenvalues nor -- unless a neighbouring row does -- the product and raw rows under them.  Bit s of
    // the four masks = raw-row index s (image row Y0 - 3 + s): outputs wanted, eigenvalue rows (outputs +- 1: the 3 x 3 test), product rows (+- 1: box filter), raw
    // rows (+- 1: Sobel).  All four are wavefront-uniform, the stages below branch on them; a skipped stage leaves stale registers nobody reads.
    unsigned long long need_out = allow;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) need_out |= (unsigned)__shfl_xor((int)(unsigned)need_out, o);
    need_out = (unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)need_out) << 3;
    const unsigned long long need_eig = need_out | (need_out << 1) | (need_out >> 1);
    const unsigned long long need_prod = need_eig | (need_eig << 1) | (need_eig >> 1);
    const unsigned long long need_raw = need_prod | (need_prod << 1) | (need_prod >> 1);

    const uint8_t* img = A.pyr + b * A.pyr_seq_stride + g.img_off;
    const float f1 = (float)(1.0 * (1.0 / (4.0 * 3.0 * 255.0))), f0 = (float)(2.0 * (1.0 / (4.0 * 3.0 * 255.0)));
    // every raw row of the strip in flight at once: bytes x - 1 .. x + 2 of rows Y0 - 3 ..; columns / rows past the ones an output needs are clamped into the border
    uint32_t raw[NS];
    {
        const int xl = min(x, g.w + 1) - 1;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const int y = min(Y0 - 3 + s, g.h + 1);
            raw[s] = 0;
            if ((need_raw >> s) & 1ull) raw[s] = *reinterpret_cast<const u32_unaligned*>(img + (ptrdiff_t)y * g.stride + xl);
        }
    }
    // the box filter's own border (BORDER_REFLECT_101 on the product images): column -1 takes the products of column 1, column w those of column w - 2
    const bool fix_h = X0 == 0 || X0 - 2 + 63 >= g.w;
    int fix_src = lane;
    if (x == -1) fix_src = lane + (g.w > 1 ? 2 : 1);
    if (x == g.w) fix_src = lane - (g.w > 1 ? 2 : 1);
    const bool fix_v = Y0 == 0 || Y0 + R + 1 >= g.h;   // + 1: the halo row Y0 + R (3 x 3 test of the last output row) can be image row h - 1

    float hd[3] = {0.f, 0.f, 0.f}, hm[3] = {0.f, 0.f, 0.f};                      // per raw row: (float)(p[x+1] - p[x-1]) and the [f1 f0 f1] row sum
    double hxx[3] = {0, 0, 0}, hxy[3] = {0, 0, 0}, hyy[3] = {0, 0, 0};           // horizontal 3-sums of the product rows
    float er[3] = {0.f, 0.f, 0.f}, em[3] = {0.f, 0.f, 0.f};                      // eigenvalue rows and their horizontal 3-maxima
    unsigned best = 0;
    int n_cand = 0;   // uniform
    unsigned long long* const cand_b = A.cand + b * A.cand_seq_stride;
    int* const count_b = A.cand_count + b;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        if ((need_raw >> s) & 1ull) {   // raw row s
            const uint32_t v = raw[s];
            const float p0 = (float)(v & 0xffu), p1 = (float)((v >> 8) & 0xffu), p2 = (float)((v >> 16) & 0xffu);
            hd[s % 3] = p2 - p0;
            float t = f1 * p0; t += f0 * p1; t += f1 * p2;
            hm[s % 3] = t;
        }
        if (s >= 2 && ((need_prod >> (s - 1)) & 1ull)) {   // products of row s - 1 and their horizontal sums
            const float t0 = hd[(s + 1) % 3], t1 = hd[(s + 2) % 3], t2 = hd[s % 3];
            const float dx = (t0 + t2) * f1 + t1 * f0;
            const float dy = hm[s % 3] - hm[(s + 1) % 3];
            float pxx = dx * dx, pxy = dx * dy, pyy = dy * dy;
            if (fix_h) { pxx = __shfl(pxx, fix_src); pxy = __shfl(pxy, fix_src); pyy = __shfl(pyy, fix_src); }
            hxx[(s + 2) % 3] = (double)dpp_from_left(pxx) + ((double)pxx + (double)dpp_from_right(pxx));
            hxy[(s + 2) % 3] = (double)dpp_from_left(pxy) + ((double)pxy + (double)dpp_from_right(pxy));
            hyy[(s + 2) % 3] = (double)dpp_from_left(pyy) + ((double)pyy + (double)dpp_from_right(pyy));
        }
        if (s >= 4 && ((need_eig >> (s - 2)) & 1ull)) {   // eigenvalues of row s - 2 from the product rows s - 3, s - 2, s - 1
            int it = s % 3, ib = (s + 2) % 3;
            const int im = (s + 1) % 3;
            double txx = hxx[it], txy = hxy[it], tyy = hyy[it], bxx = hxx[ib], bxy = hxy[ib], byy = hyy[ib];
            const int ey = Y0 - 3 + s - 2;
            if (fix_v && (ey == 0 || ey == g.h - 1)) {   // rows -1 and h of the product images mirror rows 1 and h - 2 (a real branch: two rows of the image take it)
                asm volatile("" ::: "memory");
                if (ey == 0) { if (g.h > 1) { txx = bxx; txy = bxy; tyy = byy; } else { txx = hxx[im]; txy = hxy[im]; tyy = hyy[im]; } }
                if (ey == g.h - 1) { if (g.h > 1) { bxx = txx; bxy = txy; byy = tyy; } else { bxx = hxx[im]; bxy = hxy[im]; byy = hyy[im]; } }
            }
            const double sa = txx + hxx[im] + bxx, sb = txy + hxy[im] + bxy, sc = tyy + hyy[im] + byy;
            const float a = (float)sa * 0.5f, bb = (float)sb, c = (float)sc * 0.5f;
            const float e = (a + c) - sqrtf((a - c) * (a - c) + bb * bb);
            er[im] = e;
            em[im] = fmaxf(e, fmaxf(dpp_from_left(e), dpp_from_right(e)));
        }
        if (s >= 6 && ((need_out >> (s - 3)) & 1ull)) {   // row s - 3 = Y0 + r: masked maximum and 3 x 3 local maxima
            const int r = s - 6;
            if (r < nrows) {
                const int y = Y0 + r;
                const float v = er[s % 3];
                const bool allowed = (allow >> r) & 1u;
                if (allowed) best = max(best, f32_orderable(v));
                const float around = fmaxf(fmaxf(em[(s + 2) % 3], em[(s + 1) % 3]), fmaxf(dpp_from_left(v), dpp_from_right(v)));
                const bool is_cand = allowed && x >= 1 && x < g.w - 1 && y >= 1 && y < g.h - 1 && v != 0.f && v >= around;
                const unsigned long long cb = __ballot(is_cand);
                if (cb) {
                    if (n_cand > kCap - 64) { strip_flush(cand_b, count_b, A.cand_cap, ckeys, n_cand, lane); n_cand = 0; }
                    if (is_cand) ckeys[n_cand + __builtin_amdgcn_mbcnt_hi((unsigned)(cb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cb, 0u))] =
                        ((unsigned long long)f32_orderable(v) << 32) | (unsigned)(y * g.w + x);
                    n_cand += __builtin_popcountll(cb);
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = max(best, (unsigned)__shfl_xor((int)best, o));
    if (lane == 0 && best) atomicMax(A.maxkey + b, best);
    if (n_cand) strip_flush(cand_b, count_b, A.cand_cap, ckeys, n_cand, lane);
}

struct SelectArgs {
    unsigned long long* cand; size_t cand_seq_stride; int cand_cap;
    const int* cand_count;
    const unsigned* maxkey;  // [batch] orderable masked maximum (threshold = 0.01 * max, THRESH_TOZERO)
    const int* want;         // [batch] maxCorners for this frame (<=0: none)
    int w, h, min_dist, out_cap;
    int sort_cap;            // keys that fit the LDS sort area (power of two <= kSortLds)
    float2* out_pts;         // [batch][out_cap]
    uint16_t* out_depth;     // [batch][out_cap]
    int* out_n;              // [batch]
    const uint16_t* depth; size_t depth_seq_stride; int depth_stride;
    int skip_small;          // sequences that want <= kTopKMax corners have been served by select_topk_kernel
};

constexpr int kSortLds = 16384;  // most keys sorted inside LDS (128 KiB); larger candidate sets sort in global memory

template <class P>
__device__ __forceinline__ void bitonic_desc(P keys, int npow2, int tid, int nthreads) {
    for (int k = 2; k <= npow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npow2; i += nthreads) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = keys[i], c = keys[l];
                    const bool desc = (i & k) == 0;
                    if (desc ? a < c : a > c) { keys[i] = c; keys[l] = a; }
                }
            }
            __syncthreads();
        }
}

// The few corners a frame with its tracks alive asks for (round 6).  In steady state a sequence wants 3-4 new corners out of ~1 500 candidates; sorting them all
// (66 barrier-separated stages of a bitonic network in 128 KB of LDS) to read the first handful was 37 us per frame.  The greedy selection in sorted order is the
// same as: take the largest remaining key, drop every candidate closer than min_dist to it, repeat -- a candidate is rejected in the sorted walk exactly when an
// accepted corner with a larger key lies within min_dist of it.  One 1024-thread block per sequence, one block-wide 64-bit maximum and one barrier per corner, the
// candidates streamed from global memory (L2) each round, the accepted corners in registers of every thread; no dynamic LDS, so the block does not keep a CU's
// LDS from the back end's kernels.  Sequences that want more than kTopKMax corners (the first frame, a lost scene) are left to select_corners_kernel.
constexpr int kTopKMax = 16;
constexpr int kTopKQ = 4;   // candidates per thread held in registers (4 096 per sequence; more are streamed from global memory every round)
__global__ void __launch_bounds__(1024) select_topk_kernel(SelectArgs A) {
    __shared__ unsigned long long s_best[3];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int want = A.want[b];
    if (want <= 0) { if (tid == 0) A.out_n[b] = 0; return; }
    if (want > kTopKMax) return;
    int n = A.cand_count[b];
    if (n > A.cand_cap) n = A.cand_cap;
    if (n == 0) { if (tid == 0) A.out_n[b] = 0; return; }
    const unsigned long long* gk = A.cand + b * A.cand_seq_stride;
    const unsigned mk = A.maxkey[b];
    const double maxVal = mk ? (double)f32_from_orderable(mk) : 0.0;
    const float thresh = (float)(maxVal * 0.01);
    const int md2 = A.min_dist * A.min_dist;
    if (tid < 3) s_best[tid] = 0ull;
    // this thread's candidates above the quality threshold, with their coordinates (0: none / decided)
    unsigned long long kq[kTopKQ];
    int kx[kTopKQ], ky[kTopKQ];
#pragma unroll
    for (int q = 0; q < kTopKQ; q++) {
        const int i = tid + 1024 * q;
        unsigned long long k = i < n ? gk[i] : 0ull;
        if (!(f32_from_orderable((unsigned)(k >> 32)) > thresh)) k = 0ull;
        const unsigned off = (unsigned)(k & 0xffffffffu);
        ky[q] = (int)(off / (unsigned)A.w); kx[q] = (int)(off - (unsigned)ky[q] * A.w);
        kq[q] = k;
    }
    const bool streamed = n > 1024 * kTopKQ;   // the rest of a very long list is read again every round and tested against every accepted corner
    int ax[kTopKMax], ay[kTopKMax];
#pragma unroll
    for (int a = 0; a < kTopKMax; a++) { ax[a] = 0; ay[a] = 0; }
    unsigned long long last = ~0ull;
    __syncthreads();
    int nacc = 0;
    for (int r = 0; r < want; r++) {
        unsigned long long best = 0;
#pragma unroll
        for (int q = 0; q < kTopKQ; q++) if (kq[q] > best) best = kq[q];
        if (streamed)
            for (int i = tid + 1024 * kTopKQ; i < n; i += 1024) {
                const unsigned long long k = gk[i];
                if (k >= last || !(f32_from_orderable((unsigned)(k >> 32)) > thresh)) continue;
                const unsigned off = (unsigned)(k & 0xffffffffu);
                const int y = (int)(off / (unsigned)A.w), x = (int)(off - (unsigned)y * A.w);
                bool ok = true;
#pragma unroll
                for (int a = 0; a < kTopKMax; a++) {
                    const int dx = x - ax[a], dy = y - ay[a];
                    if (a < nacc && dx * dx + dy * dy < md2) ok = false;
                }
                if (ok && k > best) best = k;
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(best >> 32), o), lo = (unsigned)__shfl_xor((int)(unsigned)best, o);
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            if (other > best) best = other;
        }
        if (lane == 0 && best) atomicMax(&s_best[r % 3], best);
        if (tid == 0) s_best[(r + 1) % 3] = 0ull;   // the next round's cell: last read two rounds ago, behind a barrier since
        __syncthreads();
        best = s_best[r % 3];
        if (best == 0) break;   // the same word in every thread
        const unsigned off = (unsigned)(best & 0xffffffffu);
        const int y = (int)(off / (unsigned)A.w), x = (int)(off - (unsigned)y * A.w);
#pragma unroll
        for (int a = 0; a < kTopKMax; a++) if (a == nacc) { ax[a] = x; ay[a] = y; }
#pragma unroll
        for (int q = 0; q < kTopKQ; q++) {   // the corner itself and every candidate closer than min_dist to it are decided
            const int dx = kx[q] - x, dy = ky[q] - y;
            if (kq[q] == best || dx * dx + dy * dy < md2) kq[q] = 0ull;
        }
        if (tid == 0) {
            A.out_pts[(size_t)b * A.out_cap + nacc] = make_float2((float)x, (float)y);
            A.out_depth[(size_t)b * A.out_cap + nacc] = A.depth ? A.depth[b * A.depth_seq_stride + (size_t)y * A.depth_stride + x] : (uint16_t)0;
        }
        nacc++;
        last = best;
    }
    if (tid == 0) A.out_n[b] = nacc;
}

// One 1024-thread block per sequence: sort candidates (value desc, address desc), then wavefront 0 runs the
// greedy minimum-distance selection 64 candidates at a time against a cell grid of accepted corners.
__global__ void __launch_bounds__(1024) select_corners_kernel(SelectArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int want = A.want[b];
    if (want <= 0) { if (tid == 0) A.out_n[b] = 0; return; }
    if (A.skip_small && want <= kTopKMax) return;
    int n = A.cand_count[b];
    if (n > A.cand_cap) n = A.cand_cap;
    if (n == 0) { if (tid == 0) A.out_n[b] = 0; return; }
    unsigned long long* gk = A.cand + b * A.cand_seq_stride;
    int npow2 = 64;
    while (npow2 < n) npow2 <<= 1;
    unsigned long long* lk = reinterpret_cast<unsigned long long*>(smem);
    const bool in_lds = npow2 <= A.sort_cap;
    const unsigned mk = A.maxkey[b];
    const double maxVal = mk ? (double)f32_from_orderable(mk) : 0.0;
    const float thresh = (float)(maxVal * 0.01);
    // candidates are all unmasked non-zero local maxima; keep those above the quality threshold (others sort last as 0)
    auto keep = [&](unsigned long long k) -> unsigned long long { return f32_from_orderable((unsigned)(k >> 32)) > thresh ? k : 0ull; };
    if (in_lds) {
        // only candidates above the quality threshold take part in the sort: they are packed to the front of the LDS area (their order there is settled by the
        // sort: keys are unique), one LDS atomic per wavefront; typically half of the local maxima pass, which halves the sort's size and drops a round of stages
        __shared__ int n_kept;
        if (tid == 0) n_kept = 0;
        __syncthreads();
        for (int i0 = 0; i0 < n; i0 += 1024) {
            const int i = i0 + tid;
            const unsigned long long k = i < n ? keep(gk[i]) : 0ull;
            const unsigned long long bal = __ballot(k != 0ull);
            int base = 0;
            if ((tid & 63) == 0 && bal) base = atomicAdd(&n_kept, __builtin_popcountll(bal));
            base = __builtin_amdgcn_readfirstlane(base);
            if (k) lk[base + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] = k;
        }
        __syncthreads();
        n = n_kept;
        if (n == 0) { if (tid == 0) A.out_n[b] = 0; return; }
        npow2 = 64;
        while (npow2 < n) npow2 <<= 1;
        for (int i = n + tid; i < npow2; i += 1024) lk[i] = 0ull;
        __syncthreads();
        bitonic_desc(lk, npow2, tid, 1024);
    } else {
        for (int i = tid; i < npow2; i += 1024) gk[i] = i < n ? keep(gk[i]) : 0ull;  // cand buffers are sized to a power of two
        __threadfence_block();
        __syncthreads();
        bitonic_desc(gk, npow2, tid, 1024);
    }
    if (tid >= 64) return;
    // ---- greedy selection (featureselect.cpp: grid of cell_size = min_dist, 3x3 neighbourhood test)
    const unsigned long long* keys = in_lds ? lk : gk;
    const int lane = tid;
    const int cell = A.min_dist >= 1 ? A.min_dist : 1;
    const int gw = (A.w + cell - 1) / cell, gh = (A.h + cell - 1) / cell;
    // accepted corners live after the key area: head[gw*gh] (int16 index or -1), then (x,y,next) records
    short* head = reinterpret_cast<short*>(smem + (size_t)A.sort_cap * 8);
    short* rec = head + ((gw * gh + 3) & ~3);  // rec[3*i] = x, y, next
    for (int i = lane; i < gw * gh; i += 64) head[i] = -1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int md2 = A.min_dist * A.min_dist;
    int naccept = 0;
    for (int base = 0; base < n && naccept < want; base += 64) {
        const int ci = base + lane;
        const unsigned long long key = ci < n ? keys[ci] : 0ull;
        const bool valid = key != 0ull;
        if (!__any(valid)) break;  // zero keys (below threshold / padding) sort last
        const unsigned off = (unsigned)(key & 0xffffffffu);
        const int y = (int)(off / (unsigned)A.w), x = (int)(off - (unsigned)y * A.w);
        const int xc = x / cell, yc = y / cell;
        bool good = valid;
        if (good && A.min_dist >= 1) {
            const int x1 = max(0, xc - 1), y1 = max(0, yc - 1), x2 = min(gw - 1, xc + 1), y2 = min(gh - 1, yc + 1);
            for (int yy = y1; yy <= y2 && good; yy++)
                for (int xx = x1; xx <= x2 && good; xx++)
                    for (int k = head[yy * gw + xx]; k >= 0; k = rec[3 * k + 2]) {
                        const int dx = x - rec[3 * k], dy = y - rec[3 * k + 1];
                        if (dx * dx + dy * dy < md2) { good = false; break; }
                    }
        }
        unsigned long long m = __ballot(good);
        while (m && naccept < want) {
            const int l = __builtin_ctzll(m);
            const int ax = __builtin_amdgcn_readlane(x, l), ay = __builtin_amdgcn_readlane(y, l);
            if (lane == l) {
                rec[3 * naccept] = (short)ax; rec[3 * naccept + 1] = (short)ay;
                rec[3 * naccept + 2] = head[yc * gw + xc];
                head[yc * gw + xc] = (short)naccept;
                A.out_pts[(size_t)b * A.out_cap + naccept] = make_float2((float)ax, (float)ay);
                A.out_depth[(size_t)b * A.out_cap + naccept] = A.depth ? A.depth[b * A.depth_seq_stride + (size_t)ay * A.depth_stride + ax] : (uint16_t)0;
            }
            naccept++;
            if (A.min_dist >= 1 && lane > l && good) {
                const int dx = x - ax, dy = y - ay;
                if (dx * dx + dy * dy < md2) good = false;
            }
            if (lane <= l) good = false;
            m = __ballot(good);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) A.out_n[b] = naccept;
}

}  // namespace gf
