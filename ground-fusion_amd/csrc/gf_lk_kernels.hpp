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
    int der_off;       // int32 (= short2) element offset of interior pixel (0,0) inside one derivative pyramid
};

struct PyrGeom {
    LevelGeom lv[kMaxLevels];
    int nlevels;
    size_t img_bytes;  // bytes of one image pyramid
    size_t der_elems;  // short2 elements of one derivative pyramid
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

// pyrDown (5-tap [1 4 6 4 1] separable, (s+128)>>8) from level l to l+1, written over the whole padded
// domain of level l+1 (border pixels are the REFLECT_101 images of interior ones, recomputed in place).
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

// Scharr derivative of every level: out = (dI/dx, dI/dy) as short2, kernels (3,10,3) x (-1,0,1).
// grid.y = sequence, grid.z = level.  The derivative border stays zero (BORDER_CONSTANT).
__global__ void __launch_bounds__(256) scharr_kernel(const uint8_t* __restrict__ pyr, size_t pyr_seq_stride, int* __restrict__ der,
                                                     size_t der_seq_stride, PyrGeom G) {
    const LevelGeom g = G.lv[blockIdx.z];
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= g.w * g.h) return;
    int y = t / g.w, x = t - y * g.w;
    const uint8_t* p = pyr + blockIdx.y * pyr_seq_stride + g.img_off + (size_t)y * g.stride + x;
    const uint8_t* r0 = p - g.stride;
    const uint8_t* r2 = p + g.stride;
    int a0 = r0[-1], a1 = r0[0], a2 = r0[1];
    int b0 = p[-1], b2 = p[1];
    int c0 = r2[-1], c1 = r2[0], c2 = r2[1];
    int dx = ((a2 + c2) * 3 + b2 * 10) - ((a0 + c0) * 3 + b0 * 10);
    int dy = ((c2 - a2) + (c0 - a0)) * 3 + (c1 - a1) * 10;
    der[blockIdx.y * der_seq_stride + g.der_off + (size_t)y * g.stride + x] = (dx & 0xffff) | (dy << 16);
}

// Four-pixel-per-thread variants for level widths that are multiples of 4 (640 / 320 / 160 / 80): aligned dword loads instead of byte
// loads, one dword / int4 store per thread.  Same integer arithmetic as the scalar kernels above.
__device__ __forceinline__ int byte_of(uint32_t v, int k) { return (int)((v >> (8 * k)) & 0xffu); }

// pyrDown, interior of level l+1 only (the border is filled by pyr_border_kernel): thread = 4 consecutive output pixels of one row
__global__ void __launch_bounds__(256) pyr_down4_kernel(uint8_t* __restrict__ pyr, size_t pyr_seq_stride, LevelGeom s, LevelGeom d) {
    const int qw = d.w >> 2;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= qw * d.h) return;
    const int y = t / qw, x = (t - y * qw) << 2;
    uint8_t* base = pyr + blockIdx.y * pyr_seq_stride;
    const uint8_t* sp = base + s.img_off + (size_t)(2 * y - 2) * s.stride + 2 * x;   // 8-byte aligned
    int acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int dy = 0; dy < 5; dy++) {
        const int ky = dy == 0 || dy == 4 ? 1 : (dy == 2 ? 6 : 4);
        const uint32_t* r = reinterpret_cast<const uint32_t*>(sp + (size_t)dy * s.stride);
        const uint32_t w0 = r[-1], w1 = r[0], w2 = r[1], w3 = r[2];   // source columns 2x-4 .. 2x+11
        int p[12];
#pragma unroll
        for (int k = 0; k < 4; k++) { p[k] = byte_of(w0, k); p[4 + k] = byte_of(w1, k); p[8 + k] = byte_of(w2, k); }
        const int p12 = byte_of(w3, 0);
#pragma unroll
        for (int i = 0; i < 4; i++) {   // output i: source columns 2(x+i)-2 .. 2(x+i)+2 = p[2 + 2i] .. p[6 + 2i]
            const int e = 6 + 2 * i < 12 ? p[(6 + 2 * i) < 12 ? 6 + 2 * i : 11] : p12;
            acc[i] += ky * (p[2 + 2 * i] + 4 * p[3 + 2 * i] + 6 * p[4 + 2 * i] + 4 * p[5 + 2 * i] + e);
        }
    }
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) out |= (uint32_t)((acc[i] + 128) >> 8) << (8 * i);
    *reinterpret_cast<uint32_t*>(base + d.img_off + (size_t)y * d.stride + x) = out;
}

// REFLECT_101 border of one level from its own interior: thread = 4 destination bytes of the padded domain, interior groups are skipped
__global__ void __launch_bounds__(256) pyr_border_kernel(uint8_t* __restrict__ pyr, size_t pyr_seq_stride, LevelGeom g) {
    const int pw = g.w + 2 * kPad, ph = g.h + 2 * kPad, qw = pw >> 2;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= qw * ph) return;
    const int py = t / qw, px = (t - py * qw) << 2;
    if (py >= kPad && py < kPad + g.h && px >= kPad && px + 3 < kPad + g.w) return;
    uint8_t* base = pyr + blockIdx.y * pyr_seq_stride + g.img_off;
    const uint8_t* src = base + (size_t)reflect101(py - kPad, g.h) * g.stride;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) v |= (uint32_t)src[reflect101(px + k - kPad, g.w)] << (8 * k);
    *reinterpret_cast<uint32_t*>(base + (ptrdiff_t)(py - kPad) * g.stride + (px - kPad)) = v;
}

// Scharr derivative, thread = 4 consecutive pixels of one row; grid.y = sequence, grid.z = level
__global__ void __launch_bounds__(256) scharr4_kernel(const uint8_t* __restrict__ pyr, size_t pyr_seq_stride, int* __restrict__ der, size_t der_seq_stride, PyrGeom G) {
    const LevelGeom g = G.lv[blockIdx.z];
    const int qw = g.w >> 2;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= qw * g.h) return;
    const int y = t / qw, x = (t - y * qw) << 2;
    const uint8_t* p = pyr + blockIdx.y * pyr_seq_stride + g.img_off + (size_t)y * g.stride + x;
    int a[6], b[6], c[6];   // columns x-1 .. x+4 of rows y-1, y, y+1
    {
        const uint32_t* r = reinterpret_cast<const uint32_t*>(p - g.stride);
        const uint32_t l = r[-1], m = r[0], h2 = r[1];
        a[0] = byte_of(l, 3); a[5] = byte_of(h2, 0);
#pragma unroll
        for (int k = 0; k < 4; k++) a[1 + k] = byte_of(m, k);
    }
    {
        const uint32_t* r = reinterpret_cast<const uint32_t*>(p);
        const uint32_t l = r[-1], m = r[0], h2 = r[1];
        b[0] = byte_of(l, 3); b[5] = byte_of(h2, 0);
#pragma unroll
        for (int k = 0; k < 4; k++) b[1 + k] = byte_of(m, k);
    }
    {
        const uint32_t* r = reinterpret_cast<const uint32_t*>(p + g.stride);
        const uint32_t l = r[-1], m = r[0], h2 = r[1];
        c[0] = byte_of(l, 3); c[5] = byte_of(h2, 0);
#pragma unroll
        for (int k = 0; k < 4; k++) c[1 + k] = byte_of(m, k);
    }
    int o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int dx = ((a[i + 2] + c[i + 2]) * 3 + b[i + 2] * 10) - ((a[i] + c[i]) * 3 + b[i] * 10);
        const int dy = ((c[i + 2] - a[i + 2]) + (c[i] - a[i])) * 3 + (c[i + 1] - a[i + 1]) * 10;
        o[i] = (dx & 0xffff) | (dy << 16);
    }
    *reinterpret_cast<int4*>(der + blockIdx.y * der_seq_stride + g.der_off + (size_t)y * g.stride + x) = make_int4(o[0], o[1], o[2], o[3]);
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

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float unif(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }

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
    const int* dI;      // its derivative pyramid (short2 packed in int)
    const uint8_t* J;   // moving image pyramid
};

// One pyramidal LK solve for one point by one wavefront.  All control flow is wave-uniform.
// Lane l < 63 owns window row r = l/3 and the 7-pixel run starting at column 7*(l%3).
// Returns status (1 = tracked); out: nx, ny; counters for the roofline's algorithmic byte count.
__device__ __forceinline__ int lk_solve(const PyrGeom& G, const LkImages im, float ppx, float ppy, float& nx, float& ny,
                                        int maxLevel, bool useInitial, uint8_t* tile, int lane, unsigned& n_levels, unsigned& n_iters) {
    const int r = lane < 63 ? lane / 3 : 20;
    const int s = lane < 63 ? lane - 3 * r : 2;
    const bool live = lane < 63;
    const float half = (kWin - 1) * 0.5f;
    int status = 1;
    float nextx = nx, nexty = ny;  // running nextPts[ptidx]
    for (int level = maxLevel; level >= 0; level--) {
        const LevelGeom g = G.lv[level];
        const float sc = (float)(1. / (1 << level));
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
        int iw00 = uni(__float2int_rn((1.f - a) * (1.f - b) * 16384.f));
        int iw01 = uni(__float2int_rn(a * (1.f - b) * 16384.f));
        int iw10 = uni(__float2int_rn((1.f - a) * b * 16384.f));
        int iw11 = 16384 - iw00 - iw01 - iw10;

        // ---- template: bilinear I (5 fractional bits) and its derivative for this lane's 7 pixels
        int tI[7], tX[7], tY[7];
        int a11 = 0, a12 = 0, a22 = 0;
        {
            const int x0 = ipx + 7 * s;
            const uint8_t* irow = im.I + g.img_off + (ptrdiff_t)(ipy + r) * g.stride;
            const int o = x0 & 3;
            const uint32_t* q0 = reinterpret_cast<const uint32_t*>(irow + (x0 - o));
            const uint32_t* q1 = reinterpret_cast<const uint32_t*>(irow + g.stride + (x0 - o));
            uint32_t l0, h0, l1, h1;
            align8(q0[0], q0[1], q0[2], o, l0, h0);
            align8(q1[0], q1[1], q1[2], o, l1, h1);
            const int* d0 = im.dI + g.der_off + (ptrdiff_t)(ipy + r) * g.stride + x0;
            const int* d1 = d0 + g.stride;
            U4a da0 = *reinterpret_cast<const U4a*>(d0), da1 = *reinterpret_cast<const U4a*>(d0 + 4);
            U4a db0 = *reinterpret_cast<const U4a*>(d1), db1 = *reinterpret_cast<const U4a*>(d1 + 4);
            const int top[8] = {(int)da0.x, (int)da0.y, (int)da0.z, (int)da0.w, (int)da1.x, (int)da1.y, (int)da1.z, (int)da1.w};
            const int bot[8] = {(int)db0.x, (int)db0.y, (int)db0.z, (int)db0.w, (int)db1.x, (int)db1.y, (int)db1.z, (int)db1.w};
#pragma unroll
            for (int k = 0; k < 7; k++) {
                int i00 = byte_of(l0, h0, k), i01 = byte_of(l0, h0, k + 1), i10 = byte_of(l1, h1, k), i11 = byte_of(l1, h1, k + 1);
                int iv = (__mul24(i00, iw00) + __mul24(i01, iw01) + __mul24(i10, iw10) + __mul24(i11, iw11) + (1 << 8)) >> 9;   // all operands < 2^15
                int x00 = (short)(top[k] & 0xffff), x01 = (short)(top[k + 1] & 0xffff), x10 = (short)(bot[k] & 0xffff), x11 = (short)(bot[k + 1] & 0xffff);
                int y00 = top[k] >> 16, y01 = top[k + 1] >> 16, y10 = bot[k] >> 16, y11 = bot[k + 1] >> 16;
                int ix = (__mul24(x00, iw00) + __mul24(x01, iw01) + __mul24(x10, iw10) + __mul24(x11, iw11) + (1 << 13)) >> 14;
                int iy = (__mul24(y00, iw00) + __mul24(y01, iw01) + __mul24(y10, iw10) + __mul24(y11, iw11) + (1 << 13)) >> 14;
                ix = (short)ix; iy = (short)iy; iv = (short)iv;
                if (!live) { ix = 0; iy = 0; iv = 0; }
                tI[k] = iv; tX[k] = ix; tY[k] = iy;
                a11 += __mul24(ix, ix); a12 += __mul24(ix, iy); a22 += __mul24(iy, iy);
            }
        }
        const double iA11 = wave_sum_exact(a11), iA12 = wave_sum_exact(a12), iA22 = wave_sum_exact(a22);
        const float FLT_SCALE = 1.f / (1 << 20);
        const float A11 = i64_to_f32(iA11) * FLT_SCALE, A12 = i64_to_f32(iA12) * FLT_SCALE, A22 = i64_to_f32(iA22) * FLT_SCALE;
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
            const int inx = uni((int)floorf(npx)), iny = uni((int)floorf(npy));
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
            a = npx - inx; b = npy - iny;
            iw00 = uni(__float2int_rn((1.f - a) * (1.f - b) * 16384.f));
            iw01 = uni(__float2int_rn(a * (1.f - b) * 16384.f));
            iw10 = uni(__float2int_rn((1.f - a) * b * 16384.f));
            iw11 = 16384 - iw00 - iw01 - iw10;
            int b1 = 0, b2 = 0;
            {
                const int xo = inx - tx0 + 7 * s;
                const int o = xo & 3;
                const uint32_t* q0 = reinterpret_cast<const uint32_t*>(tile + __mul24(iny - ty0 + r, kTileStride) + (xo - o));
                const uint32_t* q1 = q0 + kTileStride / 4;
                uint32_t l0, h0, l1, h1;
                align8(q0[0], q0[1], q0[2], o, l0, h0);
                align8(q1[0], q1[1], q1[2], o, l1, h1);
                const uint32_t m0 = __builtin_amdgcn_alignbyte(h0, l0, 3), m1 = __builtin_amdgcn_alignbyte(h1, l1, 3);   // bytes 3..6 of each row
                // weights split into 7-bit halves (w = 128 * wh + wl, wh <= 128): the 4-tap sum becomes two v_dot4_u32_u8.
                // iw11 = 2^14 - iw00 - iw01 - iw10 is >= -1 (three roundings of at most 1/2 each): the unsigned dots take iw11 + 1 and the extra
                // J[r+1][k+1] is subtracted again.  The weights are wave-uniform: pinned to SGPRs so that the packing runs on the scalar unit.
                int sw00 = __builtin_amdgcn_readfirstlane(iw00), sw01 = __builtin_amdgcn_readfirstlane(iw01), sw10 = __builtin_amdgcn_readfirstlane(iw10),
                    sw11 = __builtin_amdgcn_readfirstlane(iw11 + 1);
                asm volatile("" : "+s"(sw00), "+s"(sw01), "+s"(sw10), "+s"(sw11));
                const uint32_t wh = (uint32_t)(sw00 >> 7) | ((uint32_t)(sw01 >> 7) << 8) | ((uint32_t)(sw10 >> 7) << 16) | ((uint32_t)(sw11 >> 7) << 24);
                const uint32_t wl = (uint32_t)(sw00 & 127) | ((uint32_t)(sw01 & 127) << 8) | ((uint32_t)(sw10 & 127) << 16) | ((uint32_t)(sw11 & 127) << 24);
#pragma unroll
                for (int k = 0; k < 7; k++) {
                    // p = { J[r][k], J[r][k+1], J[r+1][k], J[r+1][k+1] }
                    const uint32_t s0 = k < 3 ? l0 : k == 3 ? m0 : h0, s1 = k < 3 ? l1 : k == 3 ? m1 : h1;
                    const int kb = k < 3 ? k : k == 3 ? 0 : k - 4;
                    const uint32_t sel = (uint32_t)kb | ((uint32_t)(kb + 1) << 8) | ((uint32_t)(4 + kb) << 16) | ((uint32_t)(5 + kb) << 24);
                    const uint32_t pq = __builtin_amdgcn_perm(s1, s0, sel);
                    const uint32_t hi = __builtin_amdgcn_udot4(pq, wh, 0u, false);
                    const uint32_t lo = __builtin_amdgcn_udot4(pq, wl, 256u, false);
                    const int raw = (int)((hi << 7) + lo - (pq >> 24));
                    const int diff = (raw >> 9) - tI[k];
                    b1 = mad_i24(diff, tX[k], b1);
                    b2 = mad_i24(diff, tY[k], b2);
                }
            }
            const double ib1 = wave_sum_exact(b1), ib2 = wave_sum_exact(b2);
            const float fb1 = i64_to_f32(ib1) * FLT_SCALE, fb2 = i64_to_f32(ib2) * FLT_SCALE;
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
    const int* der;       // [batch][2 slots] derivative pyramids
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
__global__ void __launch_bounds__(256, 8) lk_track_kernel(PyrGeom G, LkBatchArgs A) {
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
    const int* prevD = A.der + ((size_t)b * 2 + A.prev_slot) * G.der_elems;
    const int* curD = A.der + ((size_t)b * 2 + (1 - A.prev_slot)) * G.der_elems;
    unsigned n_levels = 0, n_iters = 0;
    float cx, cy;
    int st;
    if (A.fwd_use_init) {
        const float2 ip = A.init_pts[pi];
        cx = unif(ip.x); cy = unif(ip.y);
    } else { cx = 0.f; cy = 0.f; }
    st = lk_solve(G, LkImages{prevI, prevD, curI}, pp.x, pp.y, cx, cy, min(A.fwd_max_level, G.nlevels - 1), A.fwd_use_init != 0, tile, lane,
                  n_levels, n_iters);
    const int fwd_st = st;
    if (A.flow_back && st) {
        float rx = pp.x, ry = pp.y;
        int rst = lk_solve(G, LkImages{curI, curD, prevI}, cx, cy, rx, ry, min(1, G.nlevels - 1), true, tile, lane, n_levels, n_iters);
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

}  // namespace gf
