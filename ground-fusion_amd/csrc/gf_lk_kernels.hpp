// gf_lk_kernels.hpp — gfx950 device code for the front end's image pyramid, Scharr derivative and
// pyramidal Lucas-Kanade tracker.  Written for CDNA4: 64-lane wavefronts, one wavefront per feature
// patch, moving-image tiles staged in LDS, exact integer accumulation with DPP/readlane reductions.
//
// Semantics follow the calls the reference makes (feature_tracker.cpp:122,132,135,141):
//   cv::calcOpticalFlowPyrLK(..., Size(21,21), maxLevel, TermCriteria(COUNT+EPS,30,0.01), flags)
// i.e. OpenCV 4.2 modules/video/src/lkpyramid.cpp (buildOpticalFlowPyramid, calcSharrDeriv,
// LKTrackerInvoker) with the int64 accumulator configuration (see DESIGN.md "arithmetic choices").
// Compiled with -ffp-contract=off: no fused multiply-add may be formed in the float solve.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gf {

constexpr int kPad = 32;        // border (pixels) around every pyramid level; rows stay 16-byte aligned
constexpr int kWin = 21;        // LK window (feature_tracker.cpp:122)
constexpr int kMaxLevels = 4;   // maxLevel 3 -> 4 levels
constexpr int kTileW = 32;      // LDS tile of the moving image: 32 x 32 pixels
constexpr int kTileH = 32;
constexpr int kTileStride = 40; // bytes per tile row in LDS (8-byte aligned rows, de-phased banks)
constexpr int kTileBytes = kTileH * kTileStride;

struct LevelGeom {
    int w, h;          // interior size
    int stride;        // bytes per padded image row ( = w + 2*kPad )
    int img_off;       // byte offset of interior pixel (0,0) inside one image pyramid
};

struct PyrGeom {
    LevelGeom lv[kMaxLevels];
    int nlevels;
    size_t img_bytes;  // bytes of one image pyramid
};

__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// ---------------------------------------------------------------------------------------------
// Level 0: copy the raw frame into the padded pyramid and synthesise the REFLECT_101 border.
// One thread per 4 destination bytes; grid.y = sequence.
__global__ void __launch_bounds__(256) pyr_level0_kernel(const uint8_t* __restrict__ raw, size_t raw_seq_stride, int raw_stride,
                                                         uint8_t* __restrict__ pyr, size_t pyr_seq_stride, LevelGeom g) {
    const int pw = g.w + 2 * kPad, ph = g.h + 2 * kPad;
    const int qw = pw >> 2;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= qw * ph) return;
    int py = t / qw, px = (t - py * qw) << 2;
    const uint8_t* src = raw + blockIdx.y * raw_seq_stride + (size_t)reflect101(py - kPad, g.h) * raw_stride;
    uint32_t v = 0;
    if (px >= kPad && px + 3 < kPad + g.w && !((raw_stride | g.w) & 3)) v = *reinterpret_cast<const uint32_t*>(src + (px - kPad));   // interior: one aligned dword
    else {
#pragma unroll
        for (int k = 0; k < 4; k++) v |= (uint32_t)src[reflect101(px + k - kPad, g.w)] << (8 * k);
    }
    uint8_t* dst = pyr + blockIdx.y * pyr_seq_stride + g.img_off - kPad * g.stride - kPad;
    *reinterpret_cast<uint32_t*>(dst + (size_t)py * g.stride + px) = v;
}

// The same with 16 destination bytes per thread, for frames whose width, row pitch and base address are multiples of 16 (640 x 480: every interior group
// is one aligned 16-byte load and one aligned 16-byte store; the 2 x 2 border groups of a row gather their mirror bytes).
__global__ void __launch_bounds__(256) pyr_level0_vec16_kernel(const uint8_t* __restrict__ raw, size_t raw_seq_stride, int raw_stride,
                                                               uint8_t* __restrict__ pyr, size_t pyr_seq_stride, LevelGeom g) {
    const int pw = g.w + 2 * kPad, ph = g.h + 2 * kPad;
    const int qw = pw >> 4;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= qw * ph) return;
    const int py = t / qw, px = (t - py * qw) << 4;
    const uint8_t* src = raw + blockIdx.y * raw_seq_stride + (size_t)reflect101(py - kPad, g.h) * raw_stride;
    uint4 v;
    if (px >= kPad && px + 15 < kPad + g.w) v = *reinterpret_cast<const uint4*>(src + (px - kPad));
    else {
        uint32_t w4[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 16; k++) w4[k >> 2] |= (uint32_t)src[reflect101(px + k - kPad, g.w)] << (8 * (k & 3));
        v = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    }
    uint8_t* dst = pyr + blockIdx.y * pyr_seq_stride + g.img_off - kPad * g.stride - kPad;
    *reinterpret_cast<uint4*>(dst + (size_t)py * g.stride + px) = v;
}

// pyrDown (5-tap [1 4 6 4 1] separable, (s+128)>>8) from level l to l+1, written over the whole padded
// domain of level l+1 (border pixels are the REFLECT_101 images of interior ones, recomputed in place).  Any level size.
__global__ void __launch_bounds__(256) pyr_down_kernel(uint8_t* __restrict__ pyr, size_t pyr_seq_stride, LevelGeom s, LevelGeom d) {
    const int pw = d.w + 2 * kPad, ph = d.h + 2 * kPad;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= pw * ph) return;
    int py = t / pw, px = t - py * pw;
    int x = reflect101(px - kPad, d.w), y = reflect101(py - kPad, d.h);
    uint8_t* base = pyr + blockIdx.y * pyr_seq_stride;
    const uint8_t* sp = base + s.img_off + (size_t)(2 * y - 2) * s.stride + (2 * x - 2);
    int acc = 0;
#pragma unroll
    for (int dy = 0; dy < 5; dy++) {
        const int ky = dy == 0 || dy == 4 ? 1 : (dy == 2 ? 6 : 4);
        const uint8_t* r = sp + dy * s.stride;
        int hs = r[0] + 4 * r[1] + 6 * r[2] + 4 * r[3] + r[4];
        acc += ky * hs;
    }
    base[d.img_off + (py - kPad) * d.stride + (px - kPad)] = (uint8_t)((acc + 128) >> 8);
}

// Four-pixel-per-thread variants for level widths that are multiples of 4 (640 / 320 / 160 / 80): aligned dword loads instead of byte
// loads, one dword store per thread.  Same integer arithmetic as the scalar kernel above.
__device__ __forceinline__ int byte_of(uint32_t v, int k) { return (int)((v >> (8 * k)) & 0xffu); }

// four consecutive pixels (y, x .. x+3) of the next level (x a multiple of 4), packed into one dword, from a pointer to source column 2x of source row 2y - 2
// (rows `stride` bytes apart, four readable bytes to the left of it): global memory or LDS
__device__ __forceinline__ uint32_t pyr_down_quad_at(const uint8_t* sp, ptrdiff_t stride) {
    // Output i of the quad is the [1 4 6 4 1] row sum over source bytes 2 + 2i .. 6 + 2i of the sixteen loaded ones: the first four of them as one dword (the
    // loaded dword itself or a v_alignbyte of two), weighted by v_dot4_u32_u8, plus the fifth byte -- 14 instructions per source row instead of ~50 of
    // shifts, masks and multiply-adds.  Integer arithmetic: the same sums.
    unsigned acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int dy = 0; dy < 5; dy++) {
        const unsigned ky = dy == 0 || dy == 4 ? 1 : (dy == 2 ? 6 : 4);
        const uint32_t* r = reinterpret_cast<const uint32_t*>(sp + (ptrdiff_t)dy * stride);
        const uint32_t w0 = r[-1], w1 = r[0], w2 = r[1], w3 = r[2];   // source columns 2x-4 .. 2x+11
        const uint32_t d0 = __builtin_amdgcn_alignbyte(w1, w0, 2), d2 = __builtin_amdgcn_alignbyte(w2, w1, 2);   // bytes 2..5, 6..9 (d1 = w1: 4..7, d3 = w2: 8..11)
        constexpr uint32_t kTaps = 0x04060401u;   // weights of bytes 0..3
        acc[0] += ky * __builtin_amdgcn_udot4(d0, kTaps, (w1 >> 16) & 0xffu, false);
        acc[1] += ky * __builtin_amdgcn_udot4(w1, kTaps, w2 & 0xffu, false);
        acc[2] += ky * __builtin_amdgcn_udot4(d2, kTaps, (w2 >> 16) & 0xffu, false);
        acc[3] += ky * __builtin_amdgcn_udot4(w2, kTaps, w3 & 0xffu, false);
    }
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) out |= ((acc[i] + 128u) >> 8) << (8 * i);
    return out;
}
// ... of level d from the padded level s in global memory
__device__ __forceinline__ uint32_t pyr_down_quad(const uint8_t* __restrict__ base, const LevelGeom& s, int y, int x) {
    return pyr_down_quad_at(base + s.img_off + (ptrdiff_t)(2 * y - 2) * s.stride + 2 * x, s.stride);   // 8-byte aligned
}
// one pixel of level d at interior coordinates (y, x)
__device__ __forceinline__ uint32_t pyr_down_one(const uint8_t* __restrict__ base, const LevelGeom& s, int y, int x) {
    const uint8_t* sp = base + s.img_off + (ptrdiff_t)(2 * y - 2) * s.stride + (2 * x - 2);
    int acc = 0;
#pragma unroll
    for (int dy = 0; dy < 5; dy++) {
        const int ky = dy == 0 || dy == 4 ? 1 : (dy == 2 ? 6 : 4);
        const uint8_t* r = sp + (ptrdiff_t)dy * s.stride;
        acc += ky * (r[0] + 4 * r[1] + 6 * r[2] + 4 * r[3] + r[4]);
    }
    return (uint32_t)((acc + 128) >> 8);
}

// Level l+1 from level l, interior AND its REFLECT_101 border in one pass: thread = four consecutive interior pixels (one dword).  A pixel within
// kPad of an edge is also the source of border pixels (its mirror images across that edge, and across the corner): the thread that computed it
// stores them too -- row mirrors as dwords, column mirrors as bytes (their dword would straddle an alignment boundary).  No pixel is filtered
// twice and nothing is read back.  Needs d.w, d.h > kPad (one reflection reaches every border pixel) and d.w a multiple of 4.
__global__ void __launch_bounds__(256) pyr_down_pad4_kernel(uint8_t* __restrict__ pyr, size_t pyr_seq_stride, LevelGeom s, LevelGeom d) {
    const int qw = d.w >> 2;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= qw * d.h) return;
    const int y = t / qw, x = (t - y * qw) << 2;
    uint8_t* base = pyr + blockIdx.y * pyr_seq_stride;
    const uint32_t v = pyr_down_quad(base, s, y, x);
    uint8_t* img = base + d.img_off;
    // rows this pixel row is mirrored to: y itself, -y (1 <= y <= kPad), 2(h-1)-y (h-1-kPad <= y <= h-2)
    int rows[3], nr = 0;
    rows[nr++] = y;
    if (y >= 1 && y <= kPad) rows[nr++] = -y;
    if (y >= d.h - 1 - kPad && y <= d.h - 2) rows[nr++] = 2 * (d.h - 1) - y;
    const bool left = x <= kPad, right = x + 3 >= d.w - 1 - kPad;   // some pixel of the quad has a column mirror
    for (int q = 0; q < nr; q++) {
        uint8_t* row = img + (ptrdiff_t)rows[q] * d.stride;
        *reinterpret_cast<uint32_t*>(row + x) = v;
        if (left) {
#pragma unroll
            for (int k = 0; k < 4; k++) { const int c = x + k; if (c >= 1 && c <= kPad) row[-c] = (uint8_t)(v >> (8 * k)); }
        }
        if (right) {
#pragma unroll
            for (int k = 0; k < 4; k++) { const int c = x + k; if (c >= d.w - 1 - kPad && c <= d.w - 2) row[2 * (d.w - 1) - c] = (uint8_t)(v >> (8 * k)); }
        }
    }
}

__device__ __forceinline__ int reflect_once(int p, int len) { return p < 0 ? -p : (p >= len ? 2 * len - 2 - p : p); }   // |overshoot| < len

// one pixel of the next level from rows [row0, ...) of level s held in LDS without borders (REFLECT_101 by index)
__device__ __forceinline__ uint32_t pyr_down_one_lds(const uint8_t* __restrict__ src, int sw, int sh, int row0, int y, int x) {
    const int c0 = reflect_once(2 * x - 2, sw), c1 = reflect_once(2 * x - 1, sw), c2 = 2 * x, c3 = reflect_once(2 * x + 1, sw), c4 = reflect_once(2 * x + 2, sw);
    int acc = 0;
#pragma unroll
    for (int dy = 0; dy < 5; dy++) {
        const int ky = dy == 0 || dy == 4 ? 1 : (dy == 2 ? 6 : 4);
        const uint8_t* r = src + (reflect_once(2 * y - 2 + dy, sh) - row0) * sw;
        acc += ky * (r[c0] + 4 * r[c1] + 6 * r[c2] + 4 * r[c3] + r[c4]);
    }
    return (uint32_t)((acc + 128) >> 8);
}

// padded rows [ra, rb) of a level from their interior pixels in LDS (rows stored from row0), each also written to the border rows it mirrors to
__device__ __forceinline__ void write_padded_rows(uint8_t* __restrict__ base, const LevelGeom& d, const uint8_t* __restrict__ lds, int row0, int ra, int rb) {
    const int pw = d.w + 2 * kPad, qw = pw >> 2;
    for (int t = threadIdx.x; t < qw * (rb - ra); t += blockDim.x) {
        const int yy = t / qw, px = (t - yy * qw) << 2, y = ra + yy;
        const uint8_t* src = lds + (y - row0) * d.w;
        uint32_t v;
        if (px >= kPad && px < kPad + d.w) v = *reinterpret_cast<const uint32_t*>(src + (px - kPad));
        else {
            v = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) v |= (uint32_t)src[reflect_once(px + k - kPad, d.w)] << (8 * k);
        }
        uint8_t* col = base + d.img_off + (px - kPad);
        *reinterpret_cast<uint32_t*>(col + (ptrdiff_t)y * d.stride) = v;
        if (y >= 1 && y <= kPad) *reinterpret_cast<uint32_t*>(col - (ptrdiff_t)y * d.stride) = v;
        if (y >= d.h - 1 - kPad && y <= d.h - 2) *reinterpret_cast<uint32_t*>(col + (ptrdiff_t)(2 * (d.h - 1) - y) * d.stride) = v;
    }
}

// The two small levels (160x120 and 80x60 at VGA) in ONE launch.  grid = (parts, sequences): a block owns a band of rows of level first+1 and
// the rows of level `first` under it.  It filters the rows of level `first` it needs (band + 2 halo rows, 1.1x the work) from global memory into
// LDS, writes its own rows of that level with their borders, filters its band of the next level from LDS and writes that with its borders.
// Global memory is only written, never read back: no fence between the phases (a device-scope fence writes the whole L2 back: 250 us per launch).
__global__ void __launch_bounds__(512) pyr_down_tail_kernel(uint8_t* __restrict__ pyr, size_t pyr_seq_stride, PyrGeom G, int first) {
    extern __shared__ __attribute__((aligned(16))) uint8_t pyr_sm[];
    uint8_t* base = pyr + blockIdx.y * pyr_seq_stride;
    const LevelGeom s = G.lv[first - 1], d = G.lv[first];
    const bool two = first + 1 < G.nlevels;
    const LevelGeom e = G.lv[two ? first + 1 : first];
    const int parts = gridDim.x, p = blockIdx.x;
    // band of the coarser level (or, without one, of level `first` in units of two rows)
    const int nb = two ? e.h : (d.h + 1) / 2;
    const int r0 = (int)((long long)nb * p / parts), r1 = (int)((long long)nb * (p + 1) / parts);
    const int oa = min(d.h, 2 * r0), ob = p == parts - 1 ? d.h : min(d.h, 2 * r1);         // rows of level `first` this block writes
    const int a = max(0, min(oa, 2 * r0 - 2)), b = min(d.h, max(ob, 2 * r1 + 1));           // rows it needs in LDS
    uint8_t* ldsA = pyr_sm;
    uint8_t* ldsB = pyr_sm + (size_t)(b - a) * d.w;
    const int qi = d.w >> 2;
    for (int t = threadIdx.x; t < qi * (b - a); t += blockDim.x) {
        const int yy = t / qi, x = (t - yy * qi) << 2;
        *reinterpret_cast<uint32_t*>(ldsA + yy * d.w + x) = pyr_down_quad(base, s, a + yy, x);
    }
    __syncthreads();
    write_padded_rows(base, d, ldsA, a, oa, ob);
    if (!two) return;
    const int qe = e.w >> 2;
    for (int t = threadIdx.x; t < qe * (r1 - r0); t += blockDim.x) {
        const int yy = t / qe, x = (t - yy * qe) << 2;
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) v |= pyr_down_one_lds(ldsA, d.w, d.h, a, r0 + yy, x + k) << (8 * k);
        *reinterpret_cast<uint32_t*>(ldsB + yy * e.w + x) = v;
    }
    __syncthreads();
    write_padded_rows(base, e, ldsB, r0, r0, r1);
}

// Levels 0 and 1 in ONE launch (round 6): the raw frame is read once.  grid = (bands of kHeadRows rows of level 1, sequences).  A block loads the 2 kHeadRows + 3
// raw rows under its band into LDS (rows past the image as their REFLECT_101 images, four mirrored bytes either side of a row), writes its 2 kHeadRows rows of
// level 0 with their borders in 16-byte units, filters its band of level 1 from LDS (the arithmetic of pyr_down_quad, four aligned dwords per row and quad) and
// writes that with its borders.  Before: a copy kernel (78.6 MB read, 98 MB written: 55 us per 256 VGA frames) and pyr_down_pad4_kernel reading level 0 back
// (32 us).  Needs an even height and a width that is a multiple of 16; other sizes keep the two kernels.
constexpr int kHeadRows = 16;
__host__ __device__ inline size_t pyr_head_lds_bytes(int w0) { return (size_t)(2 * kHeadRows + 3) * (w0 + 8) + (size_t)kHeadRows * (w0 / 2); }
__global__ void __launch_bounds__(512) pyr_head_kernel(const uint8_t* __restrict__ raw, size_t raw_seq_stride, int raw_stride,
                                                       uint8_t* __restrict__ pyr, size_t pyr_seq_stride, LevelGeom g0, LevelGeom g1) {
    extern __shared__ __attribute__((aligned(16))) uint8_t head_sm[];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * kHeadRows, r1 = min(g1.h, r0 + kHeadRows);   // this block's rows of level 1
    const int ya = 2 * r0 - 2, nrow = 2 * (r1 - r0) + 3;                    // level-0 rows ya .. ya + nrow - 1 in LDS
    const int SA = g0.w + 8;                                                // LDS row: 4 mirrored bytes, w pixels, 4 mirrored bytes (a multiple of 8)
    uint8_t* A = head_sm;
    uint8_t* Bq = head_sm + (size_t)(2 * kHeadRows + 3) * SA;               // the band of level 1, no borders
    const uint8_t* src = raw + blockIdx.y * raw_seq_stride;
    uint8_t* base = pyr + blockIdx.y * pyr_seq_stride;
    {
        const int qi = g0.w >> 4;
        for (int t = tid; t < qi * nrow; t += 512) {
            const int yy = t / qi, u = t - yy * qi;
            const uint4 v = *reinterpret_cast<const uint4*>(src + (size_t)reflect101(ya + yy, g0.h) * raw_stride + 16 * u);
            uint32_t* dst = reinterpret_cast<uint32_t*>(A + (size_t)yy * SA + 4 + 16 * u);
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
    }
    __syncthreads();
    for (int t = tid; t < nrow * 8; t += 512) {   // columns -1 .. -4 and w .. w + 3
        const int yy = t >> 3, k = t & 7;
        uint8_t* rowp = A + (size_t)yy * SA + 4;
        if (k < 4) rowp[-1 - k] = rowp[1 + k]; else rowp[g0.w + (k - 4)] = rowp[g0.w - 2 - (k - 4)];
    }
    __syncthreads();
    {   // level 0 with its borders: rows 2 r0 .. 2 r1 - 1, each also to the border rows it mirrors to
        const int q0 = (g0.w + 2 * kPad) >> 4, nout = 2 * (r1 - r0);
        uint8_t* img0 = base + g0.img_off;
        for (int t = tid; t < q0 * nout; t += 512) {
            const int yy = t / q0, px = (t - yy * q0) << 4, y = 2 * r0 + yy;
            const uint8_t* rowp = A + (size_t)(y - ya) * SA + 4;
            uint4 v;
            if (px >= kPad && px + 15 < kPad + g0.w) {
                const uint32_t* s4 = reinterpret_cast<const uint32_t*>(rowp + (px - kPad));
                v = make_uint4(s4[0], s4[1], s4[2], s4[3]);
            } else {
                uint32_t w4[4] = {0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < 16; k++) w4[k >> 2] |= (uint32_t)rowp[reflect_once(px + k - kPad, g0.w)] << (8 * (k & 3));
                v = make_uint4(w4[0], w4[1], w4[2], w4[3]);
            }
            uint8_t* col = img0 + (px - kPad);
            *reinterpret_cast<uint4*>(col + (ptrdiff_t)y * g0.stride) = v;
            if (y >= 1 && y <= kPad) *reinterpret_cast<uint4*>(col - (ptrdiff_t)y * g0.stride) = v;
            if (y >= g0.h - 1 - kPad && y <= g0.h - 2) *reinterpret_cast<uint4*>(col + (ptrdiff_t)(2 * (g0.h - 1) - y) * g0.stride) = v;
        }
    }
    {   // the band of level 1 from LDS
        const int qe = g1.w >> 2;
        for (int t = tid; t < qe * (r1 - r0); t += 512) {
            const int yy = t / qe, x = (t - yy * qe) << 2;
            *reinterpret_cast<uint32_t*>(Bq + (size_t)yy * g1.w + x) = pyr_down_quad_at(A + (size_t)(2 * yy) * SA + 4 + 2 * x, SA);
        }
    }
    __syncthreads();
    write_padded_rows(base, g1, Bq, r0, r0, r1);
}

// ---------------------------------------------------------------------------------------------
// Wave-level exact integer sum.  Per-lane |v| < 2^28 so 8-lane partial sums fit int32; the eight group
// sums are combined on the scalar unit in 64 bits.  Result is wave-uniform.
// Exact sum of one int32 per lane over the wavefront (|sum| < 2^53): three DPP adds inside 8-lane groups (no overflow: 8 x 2^28), then eight
// v_readlane + scalar 64-bit adds.  (Two v_mfma_f64_16x16x4_f64 against a matrix of ones give the same exact sum on the matrix pipe; measured
// 15 % slower here -- the dependent MFMA latency sits on the per-iteration critical path.)
__device__ __forceinline__ double wave_sum_exact(int v) {
    v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true);  // row_half_mirror
    long long s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s += (long long)__builtin_amdgcn_readlane(v, k * 8);
    return (double)s;
}

// the same sum as the float the reference makes of it: (float)(int64), one rounding.  Three DPP adds inside the 8-lane groups (no overflow: 8 x 2^28); the group
// sums are then split into a signed upper and an unsigned lower half-word, and each half goes through the remaining three levels (row_mirror, row_bcast:15,
// row_bcast:31: sums of 8 half-words, < 2^19) to lane 63; the scalar unit puts the two together and normalises for the conversion (find the leading bit, shift,
// sticky bit), two vector instructions finish it (v_cvt_f32_i32, v_ldexp_f32).  Per sum 15 vector + ~12 scalar instructions; reading the eight group sums with
// v_readlane and adding them as 64-bit scalars took 13 + 31 -- and the scalar unit is shared by the four SIMDs of a CU.
__device__ __forceinline__ float wave_sum_f32(int v) {
    v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true);  // row_half_mirror: every lane holds the sum of its group of 8
    int hi = v >> 16, lo = v & 0xffff;                        // v == hi * 65536 + lo
    hi += __builtin_amdgcn_mov_dpp(hi, 0x140, 0xf, 0xf, true);                 // row_mirror: every lane holds the sum of its row of 16
    lo += __builtin_amdgcn_mov_dpp(lo, 0x140, 0xf, 0xf, true);
    hi += __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xa, 0xf, false);          // row_bcast:15 into rows 1 and 3: rows 0 + 1, rows 2 + 3
    lo += __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xa, 0xf, false);
    hi += __builtin_amdgcn_update_dpp(0, hi, 0x143, 0xc, 0xf, false);          // row_bcast:31 into rows 2 and 3: lane 63 holds everything
    lo += __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xc, 0xf, false);
    const long long s = (long long)__builtin_amdgcn_readlane(hi, 63) * 65536 + (long long)__builtin_amdgcn_readlane(lo, 63);
    return (float)s;
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float unif(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }

// cvRound(p * (1 << W_BITS)) of a wave-uniform bilinear weight product 0 <= p <= 1 (lkpyramid.cpp: iw = cvRound(w * (1 << 14))), delivered in a scalar register.
// p * 2^14 is exact; adding 2^23 in the same fused operation rounds it to an integer, ties to even, as cvRound / rintf do; the integer then sits in the low mantissa
// bits.  One v_fma + one v_readfirstlane and a scalar subtraction instead of multiply, v_rndne, v_readfirstlane and v_cvt_i32: the weights, their fourth
// (2^14 minus the other three) and the packing into 16-bit pairs stay on the scalar unit.
__device__ __forceinline__ int rn14(float p) {
    return uni(__builtin_bit_cast(int, __builtin_fmaf(p, 16384.f, 8388608.f))) - 0x4B000000;
}
__device__ __forceinline__ float i64_to_f32(double v) { return (float)v; }  // v holds an exact integer: one rounding, as (float)(int64) in the reference build

struct __attribute__((packed, aligned(4))) U4a { uint32_t x, y, z, w; };

// d = a * b + c on the 24-bit integer multiplier (full rate); |a|, |b| < 2^23
__device__ __forceinline__ int mad_i24(int a, int b, int c) {
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// bytes o..o+7 of the 12-byte little-endian string w0 w1 w2
__device__ __forceinline__ void align8(uint32_t w0, uint32_t w1, uint32_t w2, int o, uint32_t& lo, uint32_t& hi) {
    lo = __builtin_amdgcn_alignbyte(w1, w0, o);
    hi = __builtin_amdgcn_alignbyte(w2, w1, o);
}
__device__ __forceinline__ int byte_of(uint32_t lo, uint32_t hi, int k) {
    return k < 4 ? (int)((lo >> (8 * k)) & 0xff) : (int)((hi >> (8 * (k - 4))) & 0xff);
}

struct LkImages {
    const uint8_t* I;   // template image pyramid (base of one pyramid)
    const uint8_t* J;   // moving image pyramid
};

// ---------------------------------------------------------------------------------------------
// Scharr derivative (calcSharrDeriv, lkpyramid.cpp: kernels (3,10,3) x (-1,0,1)) evaluated where LK needs it, in packed 16-bit lanes.
// A lane holds ten consecutive pixels of four image rows as u16 pairs (c0,c1)(c2,c3)...(c8,c9); from three rows it gets dI/dx and dI/dy of
// the eight inner columns as s16 pairs (d1,d2)(d3,d4)(d5,d6)(d7,d8).  |3(a+c)+10b| <= 4080 and |d| <= 4080: nothing leaves 16 bits.
typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2u as_v2u(uint32_t v) { return __builtin_bit_cast(v2u, v); }
__device__ __forceinline__ v2s as_v2s(uint32_t v) { return __builtin_bit_cast(v2s, v); }
__device__ __forceinline__ uint32_t as_u32(v2s v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t as_u32(v2u v) { return __builtin_bit_cast(uint32_t, v); }

// ten pixels starting at byte o (0..3) of the 16 bytes at p (4-byte aligned), widened to five u16 pairs
__device__ __forceinline__ void load_row10(const uint8_t* p, int o, uint32_t (&E)[5]) {
    const U4a w = *reinterpret_cast<const U4a*>(p);
    const uint32_t e0 = __builtin_amdgcn_alignbyte(w.y, w.x, o), e1 = __builtin_amdgcn_alignbyte(w.z, w.y, o), e2 = __builtin_amdgcn_alignbyte(w.w, w.z, o);
    E[0] = __builtin_amdgcn_perm(0u, e0, 0x0c010c00u); E[1] = __builtin_amdgcn_perm(0u, e0, 0x0c030c02u);
    E[2] = __builtin_amdgcn_perm(0u, e1, 0x0c010c00u); E[3] = __builtin_amdgcn_perm(0u, e1, 0x0c030c02u);
    E[4] = __builtin_amdgcn_perm(0u, e2, 0x0c010c00u);
}
// rows A (y-1), B (y), C (y+1) -> derivative pairs of columns 1..8
__device__ __forceinline__ void scharr8(const uint32_t (&A)[5], const uint32_t (&B)[5], const uint32_t (&C)[5], uint32_t (&DX)[4], uint32_t (&DY)[4]) {
    v2u V[5]; v2s H[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
        V[j] = (as_v2u(A[j]) + as_v2u(C[j])) * (unsigned short)3 + as_v2u(B[j]) * (unsigned short)10;   // 3 (a + c) + 10 b per column
        H[j] = as_v2s(C[j]) - as_v2s(A[j]);                                                             // c - a per column
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
        DX[m] = as_u32(__builtin_bit_cast(v2s, V[m + 1]) - __builtin_bit_cast(v2s, V[m]));             // V[x+1] - V[x-1] for x = 2m+1, 2m+2
        const v2s Hodd = as_v2s(__builtin_amdgcn_alignbit(as_u32(H[m + 1]), as_u32(H[m]), 16));        // (H[2m+1], H[2m+2])
        DY[m] = as_u32((H[m] + H[m + 1]) * (short)3 + Hodd * (short)10);                              // 3 (H[x-1] + H[x+1]) + 10 H[x]
    }
}
// acc + a.lo * w.lo + a.hi * w.hi (signed 16-bit halves): v_dot2c_i32_i16
__device__ __forceinline__ int dot2_acc(int acc, uint32_t w, uint32_t a) {
    return __builtin_amdgcn_sdot2(as_v2s(a), as_v2s(w), acc, false);
}
// the same sum through the three-operand form (v_dot2_i32_i16, selected by its clamp bit; the callers' sums stay far from saturation): the accumulator operand
// survives, so a constant or a value that is needed again does not have to be copied first
__device__ __forceinline__ int dot3_acc(int acc, uint32_t w, uint32_t a) {
    return __builtin_amdgcn_sdot2(as_v2s(a), as_v2s(w), acc, true);
}
// the pair (p[k], p[k+1]) out of pairs P[m] = (p[2m], p[2m+1])
__device__ __forceinline__ uint32_t pair_at(const uint32_t* P, int k) {
    return (k & 1) ? __builtin_amdgcn_alignbit(P[(k + 1) >> 1], P[k >> 1], 16) : P[k >> 1];
}

// One pyramidal LK solve for one point by one wavefront.  All control flow is wave-uniform.
// Lane l < 63 owns window row r = l/3 and the 7-pixel run starting at column 7*(l%3).
// Returns status (1 = tracked); out: nx, ny; counters for the roofline's algorithmic byte count.
__device__ __forceinline__ int lk_solve(const PyrGeom& G, const LkImages im, float ppx, float ppy, float& nx, float& ny,
                                        int maxLevel, bool useInitial, uint8_t* tile, int lane, unsigned& n_levels, unsigned& n_iters) {
    const int r = lane < 63 ? lane / 3 : 20;
    const int s = lane < 63 ? lane - 3 * r : 2;
    const bool live = lane < 63;
    const uint8_t* tile_r = tile + r * kTileStride;
    const float half = (kWin - 1) * 0.5f;
    int status = 1;
    float nextx = nx, nexty = ny;  // running nextPts[ptidx]
    for (int level = maxLevel; level >= 0; level--) {
        const LevelGeom g = G.lv[level];
        const float sc = __builtin_bit_cast(float, (127 - level) << 23);   // 2^-level exactly, as (float)(1. / (1 << level)) -- which compiled to a double-precision division per level pass
        float prevx = ppx * sc, prevy = ppy * sc;
        float ntx, nty;
        if (level == maxLevel) {
            if (useInitial) { ntx = nextx * sc; nty = nexty * sc; } else { ntx = prevx; nty = prevy; }
        } else { ntx = nextx * 2.f; nty = nexty * 2.f; }
        nextx = ntx; nexty = nty;
        prevx -= half; prevy -= half;
        const int ipx = uni((int)floorf(prevx)), ipy = uni((int)floorf(prevy));
        if (ipx < -kWin || ipx >= g.w || ipy < -kWin || ipy >= g.h) {
            if (level == 0) status = 0;
            continue;
        }
        n_levels++;
        float a = prevx - ipx, b = prevy - ipy;
        int iw00 = rn14((1.f - a) * (1.f - b));
        int iw01 = rn14(a * (1.f - b));
        int iw10 = rn14((1.f - a) * b);
        int iw11 = 16384 - iw00 - iw01 - iw10;

        // ---- template: bilinear I (5 fractional bits) and bilinear Scharr derivative for this lane's 7 pixels.  The derivative is evaluated here
        // from four image rows (16 B each) instead of being read from a derivative pyramid (32 B per row and pixel run, plus the pass that wrote it).
        int tI[7], tX[7], tY[7];
        int a11 = 0, a12 = 0, a22 = 0;
        {
            // the 16 bytes around columns x0 - 1 .. x0 + 8 (x0 = ipx + 7 s) of rows ipy + r - 1 .. + 2: a wave-uniform base (scalar registers) + a non-negative
            // 32-bit lane offset, so that the loads take the scalar-base form instead of 64-bit vector address arithmetic
            const int x0 = ipx + 7 * s;
            const int e = (ipx & 3) + 7 * s - 1;          // column x0 - 1 relative to ipx & ~3: >= -1
            const int o = e & 3;
            const uint8_t* sbase = im.I + g.img_off + (ptrdiff_t)(ipy - 1) * g.stride + (ipx & ~3) - 4;
            const unsigned loff = (unsigned)(__mul24(r, g.stride) + (e & ~3) + 4);
            const uint8_t* irow = sbase + loff;
            uint32_t R0[5], R1[5], R2[5], R3[5];   // image rows ipy+r-1 .. ipy+r+2, columns x0-1 .. x0+8
            load_row10(irow, o, R0); load_row10(sbase + g.stride + loff, o, R1); load_row10(sbase + 2 * g.stride + loff, o, R2); load_row10(sbase + 3 * g.stride + loff, o, R3);
            uint32_t XT[4], YT[4], XB[4], YB[4];   // derivative pairs of rows ipy+r (top) and ipy+r+1 (bottom), columns x0 .. x0+7
            scharr8(R0, R1, R2, XT, YT);
            scharr8(R1, R2, R3, XB, YB);
            if (!(ipx >= 0 && ipy >= 0 && ipx + kWin < g.w && ipy + kWin < g.h)) {
                // the window leaves the image: the derivative image is zero outside the interior (copyMakeBorder BORDER_CONSTANT in the reference)
                const int yt = ipy + r;
                const uint32_t mt = (yt >= 0 && yt < g.h) ? 0xffffffffu : 0u, mb = (yt + 1 >= 0 && yt + 1 < g.h) ? 0xffffffffu : 0u;
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const int xa = x0 + 2 * m;
                    const uint32_t mx = ((xa >= 0 && xa < g.w) ? 0x0000ffffu : 0u) | ((xa + 1 >= 0 && xa + 1 < g.w) ? 0xffff0000u : 0u);
                    XT[m] &= mx & mt; YT[m] &= mx & mt; XB[m] &= mx & mb; YB[m] &= mx & mb;
                }
            }
            // bilinear weights as s16 pairs (iw11 >= -1): two v_dot2c per interpolation
            const uint32_t w0 = ((uint32_t)iw00 & 0xffffu) | ((uint32_t)iw01 << 16), w1 = ((uint32_t)iw10 & 0xffffu) | ((uint32_t)iw11 << 16);
#pragma unroll
            for (int k = 0; k < 7; k++) {
                int iv = dot3_acc(dot3_acc(1 << 8, w0, pair_at(R1, k + 1)), w1, pair_at(R2, k + 1)) >> 9;
                int ix = dot3_acc(dot3_acc(1 << 13, w0, pair_at(XT, k)), w1, pair_at(XB, k)) >> 14;
                int iy = dot3_acc(dot3_acc(1 << 13, w0, pair_at(YT, k)), w1, pair_at(YB, k)) >> 14;
                if (!live) { ix = 0; iy = 0; }   // lane 63 repeats lane 62's pixels: with zero derivatives it adds nothing to any of the five sums (its I does not matter then)
                tI[k] = mad_i24(iv, -512, 1 << 8);   // the iteration's accumulator start 2^8 - (iv << 9): (raw >> 9) - iv == (raw - (iv << 9)) >> 9
                tX[k] = ix; tY[k] = iy;
                a11 = mad_i24(ix, ix, a11); a12 = mad_i24(ix, iy, a12); a22 = mad_i24(iy, iy, a22);
            }
        }
        const float FLT_SCALE = 1.f / (1 << 20);
        const float A11 = wave_sum_f32(a11) * FLT_SCALE, A12 = wave_sum_f32(a12) * FLT_SCALE, A22 = wave_sum_f32(a22) * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * kWin * kWin);
        if (minEig < 1e-4f || D < 1.1920928955078125e-07f) {
            if (level == 0) status = 0;
            continue;
        }
        D = 1.f / D;
        float npx = ntx - half, npy = nty - half;
        float pdx = 0.f, pdy = 0.f;
        int tx0 = -100000, ty0 = -100000;  // no tile resident
        const uint8_t* Jbase = im.J + g.img_off;
        for (int j = 0; j < 30; j++) {
            const float fnx = floorf(npx), fny = floorf(npy);   // kept as floats for the fractions below ((float)inx == fnx: no conversion back)
            const int inx = uni((int)fnx), iny = uni((int)fny);
            if (inx < -kWin || inx >= g.w || iny < -kWin || iny >= g.h) {
                if (level == 0) status = 0;
                break;
            }
            n_iters++;
            // window rows iny..iny+21, cols inx..inx+21 must lie inside the resident 32x32 tile
            if ((unsigned)(inx - tx0) > (unsigned)(kTileW - 22) || (unsigned)(iny - ty0) > (unsigned)(kTileH - 22)) {
                tx0 = (inx - 4) & ~3;
                ty0 = iny - 5;
                const int trow = lane >> 1, thalf = lane & 1;
                const U4a v = *reinterpret_cast<const U4a*>(Jbase + (ptrdiff_t)(ty0 + trow) * g.stride + tx0 + thalf * 16);
                __builtin_amdgcn_wave_barrier();
                uint2* dst = reinterpret_cast<uint2*>(tile + trow * kTileStride + thalf * 16);
                dst[0] = make_uint2(v.x, v.y);
                dst[1] = make_uint2(v.z, v.w);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            a = npx - fnx; b = npy - fny;
            iw00 = rn14((1.f - a) * (1.f - b));
            iw01 = rn14(a * (1.f - b));
            iw10 = rn14((1.f - a) * b);
            iw11 = 16384 - iw00 - iw01 - iw10;
            int b1 = 0, b2 = 0;
            {
                const int xo = (inx - tx0) + 7 * s;                 // scalar part + lane constant
                const int o = xo & 3;
                const uint32_t* q0 = reinterpret_cast<const uint32_t*>(tile_r + (iny - ty0) * kTileStride + (xo & ~3));   // the row offset of the lane is loop-invariant, the window's is scalar
                const uint32_t* q1 = q0 + kTileStride / 4;
                uint32_t l0, h0, l1, h1;   // bytes 0..7 of the lane's run in window rows r and r + 1
                align8(q0[0], q0[1], q0[2], o, l0, h0);
                align8(q1[0], q1[1], q1[2], o, l1, h1);
                // VERTICAL u16 pairs V[k] = (J[r][k], J[r + 1][k]), k = 0..7: one v_perm each (eight; horizontal pairs (J[k], J[k + 1]) of both rows took fourteen).
                // The bilinear sample is two 16-bit dot products against the s16 weight pairs (iw00, iw10) and (iw01, iw11) (iw11 = 2^14 - iw00 - iw01 - iw10 >= -1),
                // started from 2^8 - (I << 9) of the template pixel: the same four products and the same integer sum.  The first of the two is the three-operand
                // form (v_dot2_i32_i16, selected through its clamp bit -- nothing here comes near saturation): it leaves tI[k] where it is, the in-place
                // v_dot2c needed a copy of it per pixel and iteration; the second uses the same opcode (dependent dot products of one opcode issue back to back).
                const uint32_t wA = ((uint32_t)iw00 & 0xffffu) | ((uint32_t)iw10 << 16), wB = ((uint32_t)iw01 & 0xffffu) | ((uint32_t)iw11 << 16);
                const uint32_t V[8] = {__builtin_amdgcn_perm(l1, l0, 0x0c040c00u), __builtin_amdgcn_perm(l1, l0, 0x0c050c01u), __builtin_amdgcn_perm(l1, l0, 0x0c060c02u),
                                       __builtin_amdgcn_perm(l1, l0, 0x0c070c03u), __builtin_amdgcn_perm(h1, h0, 0x0c040c00u), __builtin_amdgcn_perm(h1, h0, 0x0c050c01u),
                                       __builtin_amdgcn_perm(h1, h0, 0x0c060c02u), __builtin_amdgcn_perm(h1, h0, 0x0c070c03u)};
#pragma unroll
                for (int k = 0; k < 7; k++) {
                    const int diff = __builtin_amdgcn_sdot2(as_v2s(V[k + 1]), as_v2s(wB), __builtin_amdgcn_sdot2(as_v2s(V[k]), as_v2s(wA), tI[k], true), true) >> 9;
                    b1 = mad_i24(diff, tX[k], b1);
                    b2 = mad_i24(diff, tY[k], b2);
                }
            }
            const float fb1 = wave_sum_f32(b1) * FLT_SCALE, fb2 = wave_sum_f32(b2) * FLT_SCALE;
            const float dx = (A12 * fb2 - A22 * fb1) * D;
            const float dy = (A12 * fb1 - A11 * fb2) * D;
            npx += dx; npy += dy;
            nextx = npx + half; nexty = npy + half;
            if ((double)dx * dx + (double)dy * dy <= 0.01 * 0.01) break;
            if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                nextx -= dx * 0.5f; nexty -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (status && level == 0) {  // err block's bounds re-check (lkpyramid.cpp, err != NULL)
            const int inx = (int)floorf(nextx - half), iny = (int)floorf(nexty - half);
            if (inx < -kWin || inx >= g.w || iny < -kWin || iny >= g.h) status = 0;
        }
    }
    nx = nextx; ny = nexty;
    return status;
}

struct LkBatchArgs {
    const uint8_t* img;   // [batch][2 slots] image pyramids
    int prev_slot;        // slot holding the previous frame; cur = 1 - prev_slot
    int cap;              // per-sequence capacity of the point arrays
    const int* n_pts;     // [batch]
    const float2* prev_pts;  // [batch][cap]
    const float2* init_pts;  // [batch][cap] predicted points (mode 1) or unused
    float2* cur_pts;      // [batch][cap] out
    uint8_t* status;      // [batch][cap] out: after fwd, reverse check, inBorder and brightness test
    uint8_t* fwd_status;  // [batch][cap] out: status of the forward pass alone (feature_tracker.cpp:124-130 counts these)
    uint16_t* depth_out;  // [batch][cap] out: depth(round(y),round(x)) for status==1 (0 if no depth)
    const uint16_t* depth;   // [batch] raw depth frames (may be null)
    size_t depth_seq_stride; int depth_stride;
    unsigned* counters;   // [batch][cap][2] out: level passes, iterations
    int fwd_max_level;    // 3 (feature_tracker.cpp:132,135) or 1 (hasPrediction, :121)
    int fwd_use_init;     // OPTFLOW_USE_INITIAL_FLOW from init_pts (hasPrediction)
    int flow_back;        // reverse LK + 0.5 px check (:138-153)
    int post_checks;      // inBorder + brightness test (:155-168)
    const uint8_t* seq_mask; // optional [batch]: process only sequences with mask!=0 (fallback relaunch)
};

// grid.x = ceil(cap/4) blocks of 4 wavefronts, grid.y = sequence.  Forward LK (feature_tracker.cpp:118-135),
// reverse LK and flow-back test (:138-153), inBorder and the brightness test with the reference's swapped
// row/column indexing (:155-168).
__global__ void __launch_bounds__(256, 7) lk_track_kernel(PyrGeom G, LkBatchArgs A) {
    __shared__ __attribute__((aligned(16))) uint8_t tiles[4 * kTileBytes];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.y;
    const int i = blockIdx.x * 4 + wave;
    if (A.seq_mask && !A.seq_mask[b]) return;
    if (i >= A.n_pts[b]) return;
    uint8_t* tile = tiles + wave * kTileBytes;
    const size_t pi = (size_t)b * A.cap + i;
    float2 pp = A.prev_pts[pi];
    pp.x = unif(pp.x); pp.y = unif(pp.y);
    const uint8_t* prevI = A.img + ((size_t)b * 2 + A.prev_slot) * G.img_bytes;
    const uint8_t* curI = A.img + ((size_t)b * 2 + (1 - A.prev_slot)) * G.img_bytes;
    unsigned n_levels = 0, n_iters = 0;
    float cx, cy;
    int st;
    if (A.fwd_use_init) {
        const float2 ip = A.init_pts[pi];
        cx = unif(ip.x); cy = unif(ip.y);
    } else { cx = 0.f; cy = 0.f; }
    st = lk_solve(G, LkImages{prevI, curI}, pp.x, pp.y, cx, cy, min(A.fwd_max_level, G.nlevels - 1), A.fwd_use_init != 0, tile, lane,
                  n_levels, n_iters);
    const int fwd_st = st;
    if (A.flow_back && st) {
        float rx = pp.x, ry = pp.y;
        int rst = lk_solve(G, LkImages{curI, prevI}, cx, cy, rx, ry, min(1, G.nlevels - 1), true, tile, lane, n_levels, n_iters);
        const double ddx = (double)(pp.x - rx), ddy = (double)(pp.y - ry);
        st = (rst && sqrt(ddx * ddx + ddy * ddy) <= 0.5) ? 1 : 0;
    }
    const LevelGeom g0 = G.lv[0];
    if (st && A.post_checks) {
        const int bx = __float2int_rn(cx), by = __float2int_rn(cy);
        if (!(1 <= bx && bx < g0.w - 1 && 1 <= by && by < g0.h - 1)) st = 0;
    }
    if (st && A.post_checks) {
        const int p_u = (int)cx, p_v = (int)cy;  // x used as ROW (feature_tracker.cpp:160-163)
        int grey = 0;
        if (p_u >= 0 && p_u < g0.h && p_v >= 0 && p_v < g0.w) grey = curI[g0.img_off + (size_t)p_u * g0.stride + p_v];
        if (grey > 250) st = 0;
    }
    if (lane == 0) {
        A.cur_pts[pi] = make_float2(cx, cy);
        A.status[pi] = (uint8_t)st;
        A.fwd_status[pi] = (uint8_t)fwd_st;
        uint16_t d = 0;
        if (st && A.depth && A.post_checks) {
            const int ry = (int)round((double)cy), rx = (int)round((double)cx);
            d = A.depth[b * A.depth_seq_stride + (size_t)ry * A.depth_stride + rx];
        }
        A.depth_out[pi] = d;
        A.counters[2 * pi] = n_levels;
        A.counters[2 * pi + 1] = n_iters;
    }
}

// ---------------------------------------------------------------------------------------------
// Round 6: P points per wavefront (P = 2 or 4).  What one point costs per Gauss-Newton iteration in lk_solve is 47 vector instructions that touch pixels, 30 for the two exact
// 64-lane sums and 38 of wave-uniform float work (weights, the 2 x 2 solve, the termination tests): two thirds of an iteration do not scale with the window.  Here every lane
// keeps its seven template pixels of P points; the per-pixel work runs point after point (skipped for a point that has converged: the branch is scalar), and everything that was
// wave-uniform becomes one instruction stream for all P points -- lanes [16 p, 16 p + 16) (P = 4) or [32 p, 32 p + 32) (P = 2) carry the scalars of point p -- and the P sums of a
// kind are reduced together: v_permlane32_swap / v_permlane16_swap fold value p onto the lanes of group p, four DPP steps finish inside the 16-lane rows.  The arithmetic of a
// point is the arithmetic of lk_solve, operation for operation (exact integer sums, the same float expressions in the same order): results are bit-identical.
template <int P> struct LkGroup { static constexpr int SH = P == 4 ? 4 : 5; };   // a point's scalars live in 1 << SH lanes
template <int P> __device__ __forceinline__ int lk_rl(int v, int p) { return __builtin_amdgcn_readlane(v, p << LkGroup<P>::SH); }   // point p's value as a scalar

__device__ __forceinline__ int fold32(int x, int y) {   // lanes 0..31: x[l] + x[l + 32]; lanes 32..63: y[l - 32] + y[l]
#if __has_builtin(__builtin_amdgcn_permlane32_swap)
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    return (int)r[0] + (int)r[1];
#else
    const int xs = __shfl_xor(x, 32), ys = __shfl_xor(y, 32);
    return (threadIdx.x & 32) ? y + ys : x + xs;
#endif
}
__device__ __forceinline__ int fold16(int x, int y) {   // rows 0, 2: x.row r + x.row (r + 1); rows 1, 3: y.row (r - 1) + y.row r
#if __has_builtin(__builtin_amdgcn_permlane16_swap)
    const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    return (int)r[0] + (int)r[1];
#else
    const int xs = __shfl_xor(x, 16), ys = __shfl_xor(y, 16);
    return (threadIdx.x & 16) ? y + ys : x + xs;
#endif
}
__device__ __forceinline__ int row16_sum(int v) {   // every lane of a 16-lane row ends with the row's sum
    v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true);   // row_half_mirror
    v += __builtin_amdgcn_mov_dpp(v, 0x140, 0xf, 0xf, true);   // row_mirror
    return v;
}
// exact sums over the wavefront of P per-lane partials (|v| < 2^28), each delivered to the lanes of its point's group as the float the reference makes of the int64 sum
// ((float)(int64): one rounding).  Signed upper / unsigned lower half-words are summed separately (64 x 2^16 fits an int32) and put together in double, where the sum is
// exact; the conversion to float then rounds once.
template <int P> __device__ __forceinline__ float multi_sum_f32(const int (&v)[P]) {
    int hi[P], lo[P];
#pragma unroll
    for (int p = 0; p < P; p++) { hi[p] = v[p] >> 16; lo[p] = v[p] & 0xffff; }
    int h, l;
    if (P == 4) {
        const int h02 = fold32(hi[0], hi[2]), h13 = fold32(hi[1], hi[P - 1]), l02 = fold32(lo[0], lo[2]), l13 = fold32(lo[1], lo[P - 1]);
        h = row16_sum(fold16(h02, h13)); l = row16_sum(fold16(l02, l13));
    } else {
        h = row16_sum(fold32(hi[0], hi[P - 1])); l = row16_sum(fold32(lo[0], lo[P - 1]));
        h = fold16(h, h); l = fold16(l, l);   // the two rows of a half
    }
    return (float)((double)h * 65536.0 + (double)l);
}
__device__ __forceinline__ int rn14v(float p) { return __builtin_bit_cast(int, __builtin_fmaf(p, 16384.f, 8388608.f)) - 0x4B000000; }   // rn14 per lane

// template of one point at one level (the block of lk_solve, unchanged): ipx, ipy, w0, w1 wave-uniform
__device__ __forceinline__ void lk_template(const LevelGeom& g, const uint8_t* imI, int ipx, int ipy, uint32_t w0, uint32_t w1, int r, int s, bool live,
                                            int (&tI)[7], int (&tX)[7], int (&tY)[7], int& a11, int& a12, int& a22) {
    const int x0 = ipx + 7 * s;
    const int e = (ipx & 3) + 7 * s - 1;
    const int o = e & 3;
    const uint8_t* sbase = imI + g.img_off + (ptrdiff_t)(ipy - 1) * g.stride + (ipx & ~3) - 4;
    const unsigned loff = (unsigned)(__mul24(r, g.stride) + (e & ~3) + 4);
    const uint8_t* irow = sbase + loff;
    uint32_t R0[5], R1[5], R2[5], R3[5];
    load_row10(irow, o, R0); load_row10(sbase + g.stride + loff, o, R1); load_row10(sbase + 2 * g.stride + loff, o, R2); load_row10(sbase + 3 * g.stride + loff, o, R3);
    uint32_t XT[4], YT[4], XB[4], YB[4];
    scharr8(R0, R1, R2, XT, YT);
    scharr8(R1, R2, R3, XB, YB);
    if (!(ipx >= 0 && ipy >= 0 && ipx + kWin < g.w && ipy + kWin < g.h)) {
        const int yt = ipy + r;
        const uint32_t mt = (yt >= 0 && yt < g.h) ? 0xffffffffu : 0u, mb = (yt + 1 >= 0 && yt + 1 < g.h) ? 0xffffffffu : 0u;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int xa = x0 + 2 * m;
            const uint32_t mx = ((xa >= 0 && xa < g.w) ? 0x0000ffffu : 0u) | ((xa + 1 >= 0 && xa + 1 < g.w) ? 0xffff0000u : 0u);
            XT[m] &= mx & mt; YT[m] &= mx & mt; XB[m] &= mx & mb; YB[m] &= mx & mb;
        }
    }
    a11 = 0; a12 = 0; a22 = 0;
#pragma unroll
    for (int k = 0; k < 7; k++) {
        int iv = dot3_acc(dot3_acc(1 << 8, w0, pair_at(R1, k + 1)), w1, pair_at(R2, k + 1)) >> 9;
        int ix = dot3_acc(dot3_acc(1 << 13, w0, pair_at(XT, k)), w1, pair_at(XB, k)) >> 14;
        int iy = dot3_acc(dot3_acc(1 << 13, w0, pair_at(YT, k)), w1, pair_at(YB, k)) >> 14;
        if (!live) { ix = 0; iy = 0; }
        tI[k] = mad_i24(iv, -512, 1 << 8);
        tX[k] = ix; tY[k] = iy;
        a11 = mad_i24(ix, ix, a11); a12 = mad_i24(ix, iy, a12); a22 = mad_i24(iy, iy, a22);
    }
}
// the residual sums of one point at its current position (the block of lk_solve's iteration): inx, iny, tx0, ty0, wA, wB wave-uniform
__device__ __forceinline__ void lk_pixels(const uint8_t* tile_r, int inx, int iny, int tx0, int ty0, int s, uint32_t wA, uint32_t wB,
                                          const int (&tI)[7], const int (&tX)[7], const int (&tY)[7], int& b1, int& b2) {
    const int xo = (inx - tx0) + 7 * s;
    const int o = xo & 3;
    const uint32_t* q0 = reinterpret_cast<const uint32_t*>(tile_r + (iny - ty0) * kTileStride + (xo & ~3));
    const uint32_t* q1 = q0 + kTileStride / 4;
    uint32_t l0, h0, l1, h1;
    align8(q0[0], q0[1], q0[2], o, l0, h0);
    align8(q1[0], q1[1], q1[2], o, l1, h1);
    const uint32_t V[8] = {__builtin_amdgcn_perm(l1, l0, 0x0c040c00u), __builtin_amdgcn_perm(l1, l0, 0x0c050c01u), __builtin_amdgcn_perm(l1, l0, 0x0c060c02u),
                           __builtin_amdgcn_perm(l1, l0, 0x0c070c03u), __builtin_amdgcn_perm(h1, h0, 0x0c040c00u), __builtin_amdgcn_perm(h1, h0, 0x0c050c01u),
                           __builtin_amdgcn_perm(h1, h0, 0x0c060c02u), __builtin_amdgcn_perm(h1, h0, 0x0c070c03u)};
    b1 = 0; b2 = 0;
#pragma unroll
    for (int k = 0; k < 7; k++) {
        const int diff = __builtin_amdgcn_sdot2(as_v2s(V[k + 1]), as_v2s(wB), __builtin_amdgcn_sdot2(as_v2s(V[k]), as_v2s(wA), tI[k], true), true) >> 9;
        b1 = mad_i24(diff, tX[k], b1);
        b2 = mad_i24(diff, tY[k], b2);
    }
}
__device__ __forceinline__ void lk_refill(uint8_t* tile, const uint8_t* Jbase, int stride, int tx0, int ty0, int lane) {
    const int trow = lane >> 1, thalf = lane & 1;
    const U4a v = *reinterpret_cast<const U4a*>(Jbase + (ptrdiff_t)(ty0 + trow) * stride + tx0 + thalf * 16);
    __builtin_amdgcn_wave_barrier();
    uint2* dst = reinterpret_cast<uint2*>(tile + trow * kTileStride + thalf * 16);
    dst[0] = make_uint2(v.x, v.y);
    dst[1] = make_uint2(v.z, v.w);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// P pyramidal LK solves by one wavefront.  Per-lane arguments carry the values of the lane's point (the same in every lane of a group); `on`: the point takes part.
template <int P>
__device__ __forceinline__ int lk_solve_mp(const PyrGeom& G, const uint8_t* imI, const uint8_t* imJ, float ppx, float ppy, float& nx, float& ny, bool on,
                                           int maxLevel, bool useInitial, uint8_t* tiles, int lane, unsigned& n_levels, unsigned& n_iters) {
    constexpr int SH = LkGroup<P>::SH;
    const int r = lane < 63 ? lane / 3 : 20;
    const int s = lane < 63 ? lane - 3 * r : 2;
    const bool live = lane < 63;
    const float half = (kWin - 1) * 0.5f;
    int status = 1;
    float nextx = nx, nexty = ny;
    int tI[P][7], tX[P][7], tY[P][7];
    for (int level = maxLevel; level >= 0; level--) {
        const LevelGeom g = G.lv[level];
        const float sc = __builtin_bit_cast(float, (127 - level) << 23);
        float prevx = ppx * sc, prevy = ppy * sc;
        float ntx, nty;
        if (level == maxLevel) {
            if (useInitial) { ntx = nextx * sc; nty = nexty * sc; } else { ntx = prevx; nty = prevy; }
        } else { ntx = nextx * 2.f; nty = nexty * 2.f; }
        nextx = ntx; nexty = nty;
        prevx -= half; prevy -= half;
        const int ipx = (int)floorf(prevx), ipy = (int)floorf(prevy);
        const bool lvl = on && !(ipx < -kWin || ipx >= g.w || ipy < -kWin || ipy >= g.h);
        if (on && !lvl && level == 0) status = 0;
        n_levels += lvl ? 1u : 0u;
        float a = prevx - ipx, b = prevy - ipy;
        int iw00 = rn14v((1.f - a) * (1.f - b));
        int iw01 = rn14v(a * (1.f - b));
        int iw10 = rn14v((1.f - a) * b);
        int iw11 = 16384 - iw00 - iw01 - iw10;
        const int w0 = (int)(((uint32_t)iw00 & 0xffffu) | ((uint32_t)iw01 << 16)), w1 = (int)(((uint32_t)iw10 & 0xffffu) | ((uint32_t)iw11 << 16));
        const unsigned long long lm = __ballot(lvl);
        if (lm == 0) continue;
        int a11[P], a12[P], a22[P];
#pragma unroll
        for (int p = 0; p < P; p++) {
            a11[p] = 0; a12[p] = 0; a22[p] = 0;
            if ((lm >> (p << SH)) & 1ull)
                lk_template(g, imI, lk_rl<P>(ipx, p), lk_rl<P>(ipy, p), (uint32_t)lk_rl<P>(w0, p), (uint32_t)lk_rl<P>(w1, p), r, s, live, tI[p], tX[p], tY[p], a11[p], a12[p], a22[p]);
        }
        const float FLT_SCALE = 1.f / (1 << 20);
        const float A11 = multi_sum_f32<P>(a11) * FLT_SCALE, A12 = multi_sum_f32<P>(a12) * FLT_SCALE, A22 = multi_sum_f32<P>(a22) * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * kWin * kWin);
        const bool itok = lvl && !(minEig < 1e-4f || D < 1.1920928955078125e-07f);
        if (lvl && !itok && level == 0) status = 0;
        D = 1.f / D;
        float npx = ntx - half, npy = nty - half;
        float pdx = 0.f, pdy = 0.f;
        bool act = itok;
        int tx0[P], ty0[P];
#pragma unroll
        for (int p = 0; p < P; p++) { tx0[p] = -100000; ty0[p] = -100000; }
        const uint8_t* Jbase = imJ + g.img_off;
        for (int j = 0; j < 30; j++) {
            const float fnx = floorf(npx), fny = floorf(npy);
            const int inx = (int)fnx, iny = (int)fny;
            if (act && (inx < -kWin || inx >= g.w || iny < -kWin || iny >= g.h)) {
                if (level == 0) status = 0;
                act = false;
            }
            const unsigned long long am = __ballot(act);
            if (am == 0) break;
            n_iters += act ? 1u : 0u;
            a = npx - fnx; b = npy - fny;
            iw00 = rn14v((1.f - a) * (1.f - b));
            iw01 = rn14v(a * (1.f - b));
            iw10 = rn14v((1.f - a) * b);
            iw11 = 16384 - iw00 - iw01 - iw10;
            const int wA = (int)(((uint32_t)iw00 & 0xffffu) | ((uint32_t)iw10 << 16)), wB = (int)(((uint32_t)iw01 & 0xffffu) | ((uint32_t)iw11 << 16));
            int b1[P], b2[P];
#pragma unroll
            for (int p = 0; p < P; p++) {
                b1[p] = 0; b2[p] = 0;
                if ((am >> (p << SH)) & 1ull) {
                    const int sx = lk_rl<P>(inx, p), sy = lk_rl<P>(iny, p);
                    uint8_t* tile = tiles + p * kTileBytes;
                    if ((unsigned)(sx - tx0[p]) > (unsigned)(kTileW - 22) || (unsigned)(sy - ty0[p]) > (unsigned)(kTileH - 22)) {
                        tx0[p] = (sx - 4) & ~3;
                        ty0[p] = sy - 5;
                        lk_refill(tile, Jbase, g.stride, tx0[p], ty0[p], lane);
                    }
                    lk_pixels(tile + r * kTileStride, sx, sy, tx0[p], ty0[p], s, (uint32_t)lk_rl<P>(wA, p), (uint32_t)lk_rl<P>(wB, p), tI[p], tX[p], tY[p], b1[p], b2[p]);
                }
            }
            const float fb1 = multi_sum_f32<P>(b1) * FLT_SCALE, fb2 = multi_sum_f32<P>(b2) * FLT_SCALE;
            const float dx = (A12 * fb2 - A22 * fb1) * D;
            const float dy = (A12 * fb1 - A11 * fb2) * D;
            if (act) {
                npx += dx; npy += dy;
                nextx = npx + half; nexty = npy + half;
                const bool small = (double)dx * dx + (double)dy * dy <= 0.01 * 0.01;
                const bool osc = !small && j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01;
                if (osc) { nextx -= dx * 0.5f; nexty -= dy * 0.5f; }
                if (small || osc) act = false;
                pdx = dx; pdy = dy;
            }
        }
        if (itok && status && level == 0) {
            const int inx = (int)floorf(nextx - half), iny = (int)floorf(nexty - half);
            if (inx < -kWin || inx >= g.w || iny < -kWin || iny >= g.h) status = 0;
        }
    }
    nx = nextx; ny = nexty;
    return status;
}

// P points per wavefront, four wavefronts per block: grid.x = ceil(cap / (4 P)), grid.y = sequence.  Same outputs as lk_track_kernel, bit for bit.
template <int P>
__global__ void __launch_bounds__(256, P == 4 ? 2 : 3) lk_track_mp_kernel(PyrGeom G, LkBatchArgs A) {
    __shared__ __attribute__((aligned(16))) uint8_t tiles[4 * P * kTileBytes];
    constexpr int SH = LkGroup<P>::SH;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.y;
    const int i0 = (blockIdx.x * 4 + wave) * P;
    if (A.seq_mask && !A.seq_mask[b]) return;
    const int n = A.n_pts[b];
    if (i0 >= n) return;
    const int i = i0 + (lane >> SH);
    const bool valid = i < n;
    uint8_t* tile = tiles + wave * P * kTileBytes;
    const size_t pi = (size_t)b * A.cap + min(i, n - 1);
    const float2 pp = A.prev_pts[pi];
    const uint8_t* prevI = A.img + ((size_t)b * 2 + A.prev_slot) * G.img_bytes;
    const uint8_t* curI = A.img + ((size_t)b * 2 + (1 - A.prev_slot)) * G.img_bytes;
    unsigned n_levels = 0, n_iters = 0;
    float cx, cy;
    if (A.fwd_use_init) { const float2 ip = A.init_pts[pi]; cx = ip.x; cy = ip.y; } else { cx = 0.f; cy = 0.f; }
    int st = lk_solve_mp<P>(G, prevI, curI, pp.x, pp.y, cx, cy, valid, min(A.fwd_max_level, G.nlevels - 1), A.fwd_use_init != 0, tile, lane, n_levels, n_iters);
    const int fwd_st = st;
    if (A.flow_back) {
        const bool back = valid && st != 0;
        if (__ballot(back)) {
            float rx = pp.x, ry = pp.y;
            const int rst = lk_solve_mp<P>(G, curI, prevI, cx, cy, rx, ry, back, min(1, G.nlevels - 1), true, tile, lane, n_levels, n_iters);
            if (back) {
                const double ddx = (double)(pp.x - rx), ddy = (double)(pp.y - ry);
                st = (rst && sqrt(ddx * ddx + ddy * ddy) <= 0.5) ? 1 : 0;
            }
        }
    }
    const LevelGeom g0 = G.lv[0];
    if (st && A.post_checks) {
        const int bx = __float2int_rn(cx), by = __float2int_rn(cy);
        if (!(1 <= bx && bx < g0.w - 1 && 1 <= by && by < g0.h - 1)) st = 0;
    }
    if (valid && st && A.post_checks) {
        const int p_u = (int)cx, p_v = (int)cy;  // x used as ROW (feature_tracker.cpp:160-163)
        int grey = 0;
        if (p_u >= 0 && p_u < g0.h && p_v >= 0 && p_v < g0.w) grey = curI[g0.img_off + (size_t)p_u * g0.stride + p_v];
        if (grey > 250) st = 0;
    }
    if (valid && (lane & ((1 << SH) - 1)) == 0) {
        const size_t po = (size_t)b * A.cap + i;
        A.cur_pts[po] = make_float2(cx, cy);
        A.status[po] = (uint8_t)st;
        A.fwd_status[po] = (uint8_t)fwd_st;
        uint16_t d = 0;
        if (st && A.depth && A.post_checks) {
            const int ry = (int)round((double)cy), rx = (int)round((double)cx);
            d = A.depth[b * A.depth_seq_stride + (size_t)ry * A.depth_stride + rx];
        }
        A.depth_out[po] = d;
        A.counters[2 * po] = n_levels;
        A.counters[2 * po + 1] = n_iters;
    }
}

// The derivative image of one level as lk_solve evaluates it (same device functions), for gf_pyramid_level's parity check against calcSharrDeriv:
// thread = eight consecutive pixels of one row; out[y][x] = (dx, dy) as s16 pairs.
__global__ void __launch_bounds__(256) deriv_probe_kernel(const uint8_t* __restrict__ pyr, LevelGeom g, int* __restrict__ out) {
    const int q = (g.w + 7) >> 3;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= q * g.h) return;
    const int y = t / q, x0 = (t - y * q) << 3, xb = x0 - 1, o = xb & 3;
    const uint8_t* irow = pyr + g.img_off + (ptrdiff_t)(y - 1) * g.stride + (xb - o);
    uint32_t R0[5], R1[5], R2[5], DX[4], DY[4];
    load_row10(irow, o, R0); load_row10(irow + g.stride, o, R1); load_row10(irow + 2 * g.stride, o, R2);
    scharr8(R0, R1, R2, DX, DY);
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (x0 + k < g.w) {
            const uint32_t dx = (DX[k >> 1] >> (16 * (k & 1))) & 0xffffu, dy = (DY[k >> 1] >> (16 * (k & 1))) & 0xffffu;
            out[(size_t)y * g.w + x0 + k] = (int)(dx | (dy << 16));
        }
}

}  // namespace gf
