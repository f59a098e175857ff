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

// grid.x = kept point, grid.y = sequence; one thread per disk row.
__global__ void __launch_bounds__(256) mask_disks_kernel(uint8_t* __restrict__ mask, size_t mask_seq_stride, int w, int h,
                                                         const int2* __restrict__ centers, const int* __restrict__ n_centers, int cap,
                                                         DiskTable T) {
    const int b = blockIdx.y, i = blockIdx.x;
    if (i >= n_centers[b]) return;
    const int dy = (int)threadIdx.x - T.radius;
    if (dy > T.radius) return;
    const int2 c = centers[(size_t)b * cap + i];
    const int y = c.y + dy;
    if ((unsigned)y >= (unsigned)h) return;
    const int hw = T.hw[dy < 0 ? -dy : dy];
    int x1 = max(c.x - hw, 0), x2 = min(c.x + hw, w - 1);
    uint8_t* row = mask + b * mask_seq_stride + (size_t)y * w;
    for (int x = x1; x <= x2; x++) row[x] = 0;
}

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

// masked maximum (cv::minMaxLoc(eig, 0, &maxVal, 0, 0, mask)); result as orderable uint, 0 == "no pixel"
__global__ void __launch_bounds__(256) masked_max_kernel(const float* __restrict__ eig, size_t eig_seq_stride, const uint8_t* __restrict__ mask,
                                                         size_t mask_seq_stride, int n, unsigned* __restrict__ out) {
    const int b = blockIdx.y;
    const float* e = eig + b * eig_seq_stride;
    const uint8_t* m = mask + b * mask_seq_stride;
    unsigned best = 0;
    for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += gridDim.x * 1024) {
        if (i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(e + i);
            const uchar4 k = *reinterpret_cast<const uchar4*>(m + i);
            if (k.x) best = max(best, f32_orderable(v.x));
            if (k.y) best = max(best, f32_orderable(v.y));
            if (k.z) best = max(best, f32_orderable(v.z));
            if (k.w) best = max(best, f32_orderable(v.w));
        } else {
            for (int q = i; q < n; q++) if (m[q]) best = max(best, f32_orderable(e[q]));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = max(best, (unsigned)__shfl_xor((int)best, o));
    if ((threadIdx.x & 63) == 0 && best) atomicMax(out + b, best);
}

// threshold(TOZERO at 0.01*max) + 3x3 dilate non-maximum suppression + mask; survivors are appended as
// 64-bit keys (value bits << 32 | pixel offset) — descending key order == OpenCV's greaterThanPtr order.
__global__ void __launch_bounds__(256) nms_collect_kernel(const float* __restrict__ eig, size_t eig_seq_stride, const uint8_t* __restrict__ mask,
                                                          size_t mask_seq_stride, int w, int h, const unsigned* __restrict__ maxkey,
                                                          unsigned long long* __restrict__ cand, size_t cand_seq_stride, int cand_cap,
                                                          int* __restrict__ cand_count, const int* __restrict__ want) {
    const int b = blockIdx.y;
    if (want[b] <= 0) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int y = 1 + t / (w - 2), x = 1 + t % (w - 2);
    if (y >= h - 1) return;
    const unsigned mk = maxkey[b];
    const double maxVal = mk ? (double)f32_from_orderable(mk) : 0.0;
    const float thresh = (float)(maxVal * 0.01);
    const float* e = eig + b * eig_seq_stride + (size_t)y * w + x;
    const float v = e[0];
    if (!(v > thresh) || v == 0.f) return;
    if (!mask[b * mask_seq_stride + (size_t)y * w + x]) return;
    bool ismax = v >= e[-w - 1] && v >= e[-w] && v >= e[-w + 1] && v >= e[-1] && v >= e[1] && v >= e[w - 1] && v >= e[w] && v >= e[w + 1];
    if (!ismax) return;
    const int slot = atomicAdd(cand_count + b, 1);
    if (slot < cand_cap)
        cand[b * cand_seq_stride + slot] = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(y * w + x);
}


// ---------------------------------------------------------------------------------------------
// Fused Shi-Tomasi pass: image tile -> covariance (LDS) -> min-eigenvalue (LDS) -> masked maximum and
// 3x3 local-maximum candidates.  The response image never touches HBM.  The mask is either an explicit
// u8 image (building-block entry point) or, in trackImage, the union of filled circles of radius MIN_DIST
// around the kept points, tested analytically with OpenCV's midpoint-circle row table (no rasterised mask).
// Tile: 64x16 outputs per 256-thread block (4 pixels per thread).
constexpr int kDT_W = 64, kDT_H = 16;
constexpr int kRawQ = (kDT_W + 8) / 4;   // dwords per row of the raw patch: bytes bx-4 .. bx+67
static_assert(kDT_W % 4 == 0 && (kDT_W + 2) % 2 == 0 && (kDT_H + 2) % 3 == 0, "work-item shapes of detect_fused_kernel");
__device__ __forceinline__ int byte8(uint32_t lo, uint32_t hi, int k) { return k < 4 ? (int)((lo >> (8 * k)) & 0xff) : (int)((hi >> (8 * (k - 4))) & 0xff); }
struct DetectArgs {
    const uint8_t* pyr; size_t pyr_seq_stride; LevelGeom g;
    const uint8_t* mask; size_t mask_seq_stride;            // optional explicit mask (non-zero = allowed)
    const int2* centers; const int* n_centers; int cap;     // else: disks
    const int* want;                                        // [batch] skip sequences that need no new corners
    unsigned* maxkey;                                       // [batch] orderable max over unmasked pixels (0 = none)
    unsigned long long* cand; size_t cand_seq_stride; int cand_cap; int* cand_count;
};

__global__ void __launch_bounds__(256) detect_fused_kernel(DetectArgs A, DiskTable T) {
    __shared__ float cxx[kDT_H + 4][kDT_W + 5], cxy[kDT_H + 4][kDT_W + 5], cyy[kDT_H + 4][kDT_W + 5];
    __shared__ float eg[kDT_H + 2][kDT_W + 3];
    __shared__ uint32_t raw[kDT_H + 6][kRawQ];
    __shared__ unsigned long long rowmask[kDT_H];
    __shared__ short s_hw[kMaxRadius + 1];
    __shared__ int n_cand, cand_base;
    __shared__ unsigned blk_max;
    constexpr int kCK = 320;   // local maxima of a 64 x 16 tile staged in LDS (a 3 x 3 maximum excludes its neighbours: <= 256 + plateaus); the rest goes out one by one
    __shared__ unsigned long long ckeys[kCK];
    const int b = blockIdx.z;
    if (A.want[b] <= 0) return;
    const LevelGeom g = A.g;
    const int bx = blockIdx.x * kDT_W, by = blockIdx.y * kDT_H, tid = threadIdx.x;
    if (tid == 0) { n_cand = 0; blk_max = 0; }
    if (tid < kDT_H) rowmask[tid] = 0ull;
    if (!A.mask && tid <= T.radius) s_hw[tid] = T.hw[tid];
    __syncthreads();
    // setMask first: a tile without a single unmasked pixel contributes neither to the masked maximum nor a candidate, and with MAX_CNT tracks
    // of radius MIN_DIST most of the image is masked -- such tiles stop here, before any arithmetic on the image.
    // The filled circles (OpenCV's midpoint-circle row table) are rasterised into one 64-bit word per tile row: a kept point whose bounding
    // box touches the tile ORs its span of every row it crosses.
    if (!A.mask) {
        const int n = A.n_centers[b];
        for (int i = tid; i < n; i += 256) {
            const int2 c = A.centers[(size_t)b * A.cap + i];
            if (c.x + T.radius < bx || c.x - T.radius >= bx + kDT_W || c.y + T.radius < by || c.y - T.radius >= by + kDT_H) continue;
            const int l0 = max(0, c.y - T.radius - by), l1 = min(kDT_H - 1, c.y + T.radius - by);
            for (int ly = l0; ly <= l1; ly++) {
                const int hw = s_hw[abs(by + ly - c.y)];
                const int x0 = max(c.x - hw, bx) - bx, x1 = min(c.x + hw, bx + kDT_W - 1) - bx;
                if (x0 > x1) continue;
                const int len = x1 - x0 + 1;
                const unsigned long long bits = (len >= 64 ? ~0ull : ((1ull << len) - 1ull)) << x0;
                atomicOr(&rowmask[ly], bits);
            }
        }
        __syncthreads();
    }
    unsigned allow_bits = 0;
    {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int li = tid + q * 256;
            const int ly = li / kDT_W, lx = li - ly * kDT_W;
            const int x = bx + lx, y = by + ly;
            if (x >= g.w || y >= g.h) continue;
            bool allowed;
            if (A.mask) allowed = A.mask[b * A.mask_seq_stride + (size_t)y * g.w + x] != 0;
            else allowed = !((rowmask[ly] >> lx) & 1ull);
            if (allowed) allow_bits |= 1u << q;
        }
        if (!__syncthreads_or(allow_bits != 0)) return;
    }
    const uint8_t* img = A.pyr + b * A.pyr_seq_stride + g.img_off;
    const float f1 = (float)(1.0 * (1.0 / (4.0 * 3.0 * 255.0))), f0 = (float)(2.0 * (1.0 / (4.0 * 3.0 * 255.0)));
    // stage 1: the image patch behind the tile, rows by-3 .. by+18, bytes bx-4 .. bx+67, as dwords.  The pyramid level carries a
    // REFLECT_101 border of kPad pixels (pyr_level0_kernel), so rows / columns just outside the image are plain reads.
    {
        const bool al = !((g.img_off | g.stride | (int)(A.pyr_seq_stride & 3)) & 3);
        for (int t = tid; t < (kDT_H + 6) * kRawQ; t += 256) {
            const int ry = t / kRawQ, rq = t - ry * kRawQ;
            const int y = by - 3 + ry, x = bx - 4 + 4 * rq;
            uint32_t v = 0;
            if (x + 3 < g.w + kPad) {      // y <= h + 17 and x >= -4 always lie inside the border
                const uint8_t* p = img + (ptrdiff_t)y * g.stride + x;
                if (al) v = *reinterpret_cast<const uint32_t*>(p);
                else v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
            }
            raw[ry][rq] = v;
        }
    }
    __syncthreads();
    // stage 2: Sobel products of the in-image pixels of the tile + 2 halo; one item = four consecutive pixels of one row (six dword reads)
    for (int t = tid; t < (kDT_H + 4) * ((kDT_W + 4) / 4); t += 256) {
        const int ty = t / ((kDT_W + 4) / 4), gq = t - ty * ((kDT_W + 4) / 4);
        const int y = by - 2 + ty;
        if (y < 0 || y >= g.h) continue;
        const uint32_t al_ = raw[ty][gq], ah_ = raw[ty][gq + 1], ml_ = raw[ty + 1][gq], mh_ = raw[ty + 1][gq + 1], cl_ = raw[ty + 2][gq], ch_ = raw[ty + 2][gq + 1];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int tx = 4 * gq + k, x = bx - 2 + tx;
            if (x < 0 || x >= g.w) continue;
            // pixel x sits at byte 4 gq + k + 2 of the raw row
            const int a0 = byte8(al_, ah_, k + 1), a1 = byte8(al_, ah_, k + 2), a2 = byte8(al_, ah_, k + 3);
            const int m0 = byte8(ml_, mh_, k + 1), m2 = byte8(ml_, mh_, k + 3);
            const int c0 = byte8(cl_, ch_, k + 1), c1 = byte8(cl_, ch_, k + 2), c2 = byte8(cl_, ch_, k + 3);
            const float t0 = (float)(a2 - a0), t1 = (float)(m2 - m0), t2 = (float)(c2 - c0);
            const float dx = (t0 + t2) * f1 + t1 * f0;
            float rt = f1 * (float)a0; rt += f0 * (float)a1; rt += f1 * (float)a2;
            float rb = f1 * (float)c0; rb += f0 * (float)c1; rb += f1 * (float)c2;
            const float dy = rb - rt;
            cxx[ty][tx] = dx * dx; cxy[ty][tx] = dx * dy; cyy[ty][tx] = dy * dy;
        }
    }
    // the box filter's own border (BORDER_REFLECT_101 on the product images): entries one or two pixels outside the image take the value
    // of their mirror pixel, which lies in this tile.  Only tiles on the image border have such entries.
    if (bx < 2 || by < 2 || bx + kDT_W + 2 > g.w || by + kDT_H + 2 > g.h) {
        __syncthreads();
        for (int t = tid; t < (kDT_W + 4) * (kDT_H + 4); t += 256) {
            const int ty = t / (kDT_W + 4), tx = t - ty * (kDT_W + 4);
            const int x = bx - 2 + tx, y = by - 2 + ty;
            if ((x >= 0 && x < g.w && y >= 0 && y < g.h) || x < -2 || x > g.w + 1 || y < -2 || y > g.h + 1) continue;
            const int sx = reflect101(x, g.w) - (bx - 2), sy = reflect101(y, g.h) - (by - 2);
            if (sx < 0 || sx >= kDT_W + 4 || sy < 0 || sy >= kDT_H + 4) continue;   // images narrower than the reflection reach: entry is never consulted
            cxx[ty][tx] = cxx[sy][sx]; cxy[ty][tx] = cxy[sy][sx]; cyy[ty][tx] = cyy[sy][sx];
        }
    }
    __syncthreads();
    // stage 3: 3x3 box sums and the smaller eigenvalue on the tile + 1 halo.  One item = 2 columns x 3 rows: row sums of three columns are
    // shared between the two columns and between the rows.  Sums run in double as in cv::boxFilter (ColumnSum<double, float>); nine float
    // products spanning < 2^52 add exactly, so the order of additions is free.  Halo entries outside the image are never consulted below.
    for (int t = tid; t < ((kDT_W + 2) / 2) * ((kDT_H + 2) / 3); t += 256) {
        const int sg = t / ((kDT_W + 2) / 2), tx = 2 * (t - sg * ((kDT_W + 2) / 2)), ty0 = 3 * sg;
        double ha[5][2], hb[5][2], hc[5][2];
#pragma unroll
        for (int rr = 0; rr < 5; rr++) {
            const float* pa = &cxx[ty0 + rr][tx]; const float* pb = &cxy[ty0 + rr][tx]; const float* pc = &cyy[ty0 + rr][tx];
            const double a1 = (double)pa[1] + (double)pa[2], b1 = (double)pb[1] + (double)pb[2], c1 = (double)pc[1] + (double)pc[2];
            ha[rr][0] = (double)pa[0] + a1; ha[rr][1] = a1 + (double)pa[3];
            hb[rr][0] = (double)pb[0] + b1; hb[rr][1] = b1 + (double)pb[3];
            hc[rr][0] = (double)pc[0] + c1; hc[rr][1] = c1 + (double)pc[3];
        }
#pragma unroll
        for (int rr = 0; rr < 3; rr++)
#pragma unroll
            for (int cc = 0; cc < 2; cc++) {
                const double sa = ha[rr][cc] + ha[rr + 1][cc] + ha[rr + 2][cc], sb = hb[rr][cc] + hb[rr + 1][cc] + hb[rr + 2][cc],
                             sc = hc[rr][cc] + hc[rr + 1][cc] + hc[rr + 2][cc];
                const float a = (float)sa * 0.5f, bb = (float)sb, c = (float)sc * 0.5f;
                eg[ty0 + rr][tx + cc] = (a + c) - sqrtf((a - c) * (a - c) + bb * bb);
            }
    }
    __syncthreads();
    unsigned best = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int li = tid + q * 256;
        const int ly = li / kDT_W, lx = li - ly * kDT_W;
        const int x = bx + lx, y = by + ly;
        if (x >= g.w || y >= g.h) continue;
        if (!((allow_bits >> q) & 1u)) continue;
        const float v = eg[ly + 1][lx + 1];
        best = max(best, f32_orderable(v));
        if (x < 1 || y < 1 || x >= g.w - 1 || y >= g.h - 1 || v == 0.f) continue;
        const bool ismax = v >= eg[ly][lx] && v >= eg[ly][lx + 1] && v >= eg[ly][lx + 2] && v >= eg[ly + 1][lx] && v >= eg[ly + 1][lx + 2] &&
                           v >= eg[ly + 2][lx] && v >= eg[ly + 2][lx + 1] && v >= eg[ly + 2][lx + 2];
        if (ismax) {
            const unsigned long long key = ((unsigned long long)f32_orderable(v) << 32) | (unsigned)(y * g.w + x);
            const int k = atomicAdd(&n_cand, 1);
            if (k < kCK) ckeys[k] = key;
            else {   // plateau-ridden tile: straight to the sequence's list (the order of candidates is settled by the sort that follows)
                const int slot = atomicAdd(A.cand_count + b, 1);
                if (slot < A.cand_cap) A.cand[b * A.cand_seq_stride + slot] = key;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = max(best, (unsigned)__shfl_xor((int)best, o));
    if ((tid & 63) == 0 && best) atomicMax(&blk_max, best);
    __syncthreads();
    if (tid == 0) {
        if (blk_max) atomicMax(A.maxkey + b, blk_max);
        n_cand = min(n_cand, kCK);
        cand_base = n_cand ? atomicAdd(A.cand_count + b, n_cand) : 0;
    }
    __syncthreads();
    for (int k = tid; k < n_cand; k += 256)
        if (cand_base + k < A.cand_cap) A.cand[b * A.cand_seq_stride + cand_base + k] = ckeys[k];
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

// One 1024-thread block per sequence: sort candidates (value desc, address desc), then wavefront 0 runs the
// greedy minimum-distance selection 64 candidates at a time against a cell grid of accepted corners.
__global__ void __launch_bounds__(1024) select_corners_kernel(SelectArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int want = A.want[b];
    if (want <= 0) { if (tid == 0) A.out_n[b] = 0; return; }
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
        for (int i = tid; i < npow2; i += 1024) lk[i] = i < n ? keep(gk[i]) : 0ull;
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
