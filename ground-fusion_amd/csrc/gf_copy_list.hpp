// gf_copy_list.hpp — many small transfers between page-locked host memory and device tables as ONE kernel.
// hipHostMalloc'ed memory is mapped into the device's address space, so a kernel can read (upload) or write (download) it over the bus.  A batch of tables --
// the ~40 of a back-end upload (gf_ba.hip), the three to five of the tracker's hand-overs around LK and the detection (gf_tracker.hip) -- is a list of
// descriptors (rows x used bytes out of a pitch); the blocks are shared out by bytes.  One launch replaces one hipMemcpy(2D)Async per table: what those cost is
// their submission and per-copy latency, not their bytes (20 MB in forty copies: 0.72 ms of device time and as much again on the submitting thread; one kernel:
// 0.50-0.54 ms).  Direction is whatever the pointers say; source and destination use the same pitch.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

namespace gfcopy {
struct Desc { const char* src; char* dst; unsigned pitch, used, rows, esize, blk0, nblk; };
constexpr int kMax = 56, kBlocks = 1024;
struct List { Desc e[kMax]; int n; };

#ifdef GF_COPY_LIST_V2
// (prepared at the end of round 4, verified on the CPU by tests/test_copy_list_host.py, not yet run on a GPU: build with -DGF_COPY_LIST_V2 to try it)
// The mapping without divisions: a one-row descriptor is a linear run of units; rows of a strided table go to the descriptor's blocks round-robin and a block's
// threads walk the row.  The default mapping below finds (row, column) of a unit by a 64-bit division by the row length -- eight division sequences, 1 280
// vector instructions for a copy kernel.
template <class V> __host__ __device__ __forceinline__ void run_of_units(const char* src, char* dst, size_t n, size_t first, size_t step) {
    size_t u = first;
    for (; u + 3 * step < n; u += 4 * step) {   // four loads in flight per lane before the first store
        V v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = *reinterpret_cast<const V*>(src + (u + q * step) * sizeof(V));
#pragma unroll
        for (int q = 0; q < 4; q++) *reinterpret_cast<V*>(dst + (u + q * step) * sizeof(V)) = v[q];
    }
    for (; u < n; u += step) *reinterpret_cast<V*>(dst + u * sizeof(V)) = *reinterpret_cast<const V*>(src + u * sizeof(V));
}
template <class V> __host__ __device__ __forceinline__ void units(const Desc& D, unsigned blk, unsigned tid) {
    const unsigned upr = D.used / sizeof(V);
    if (D.rows == 1) { run_of_units<V>(D.src, D.dst, upr, (size_t)blk * 256 + tid, (size_t)D.nblk * 256); return; }
    for (unsigned row = blk; row < D.rows; row += D.nblk) run_of_units<V>(D.src + (size_t)row * D.pitch, D.dst + (size_t)row * D.pitch, upr, tid, 256);
}
#else
// the units (V = 16, 8, 4 or 1 bytes) thread `tid` of the descriptor's block `blk` moves; host-callable so that tests/test_copy_list_host.py can walk a list on the CPU
template <class V> __host__ __device__ __forceinline__ void units(const Desc& D, unsigned blk, unsigned tid) {
    const unsigned upr = D.used / sizeof(V);
    const size_t total = (size_t)upr * D.rows, stride = (size_t)D.nblk * 256;
    size_t u = (size_t)blk * 256 + tid;
    for (; u + 3 * stride < total; u += 4 * stride) {   // four loads in flight per lane before the first store
        V v[4]; size_t off[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { const size_t uu = u + q * stride; const unsigned row = (unsigned)(uu / upr); off[q] = (size_t)row * D.pitch + (uu - (size_t)row * upr) * sizeof(V); v[q] = *reinterpret_cast<const V*>(D.src + off[q]); }
#pragma unroll
        for (int q = 0; q < 4; q++) *reinterpret_cast<V*>(D.dst + off[q]) = v[q];
    }
    for (; u < total; u += stride) { const unsigned row = (unsigned)(u / upr); const size_t off = (size_t)row * D.pitch + (u - (size_t)row * upr) * sizeof(V); *reinterpret_cast<V*>(D.dst + off) = *reinterpret_cast<const V*>(D.src + off); }
}
#endif
// what block `block` of the launch does as thread `tid`
__host__ __device__ __forceinline__ void block_work(const List& L, unsigned block, unsigned tid) {
    int k = 0;
    while (k + 1 < L.n && block >= L.e[k + 1].blk0) k++;
    const Desc& D = L.e[k];
    if (D.esize == 16) units<uint4>(D, block - D.blk0, tid);
    else if (D.esize == 8) units<uint2>(D, block - D.blk0, tid);
    else if (D.esize == 4) units<unsigned>(D, block - D.blk0, tid);
    else units<unsigned char>(D, block - D.blk0, tid);
}
// TAG: one instantiation per translation unit that launches it (a template has vague linkage; the objects are linked into one library)
template <int TAG> __global__ void __launch_bounds__(256) copy_list_kernel(List L) { block_work(L, blockIdx.x, threadIdx.x); }

struct Builder {
    List L{}; bool ok = true; unsigned nblocks = 0; long long bytes = 0;
    // rows x `used` bytes out of `pitch` bytes; both pointers as the device sees them
    void add(const void* src, void* dst, size_t rows, size_t pitch, size_t used) {
        if (used == 0 || rows == 0) return;
        if (used >= pitch) { used = pitch * rows; pitch = used; rows = 1; }
        if (!src || !dst || L.n >= kMax || used >= (1ull << 32) || pitch >= (1ull << 32)) { ok = false; return; }
        Desc& D = L.e[L.n++];
        D.src = static_cast<const char*>(src); D.dst = static_cast<char*>(dst); D.pitch = (unsigned)pitch; D.used = (unsigned)used; D.rows = (unsigned)rows;
        const size_t al = (size_t)D.pitch | D.used | (size_t)(uintptr_t)D.src | (size_t)(uintptr_t)D.dst;
        D.esize = al % 16 == 0 ? 16 : al % 8 == 0 ? 8 : al % 4 == 0 ? 4 : 1;
        bytes += (long long)D.used * D.rows;
    }
    void finish() {   // blocks in proportion to the bytes, at least one each; small lists get few blocks (16 KB per block and pass)
        const double total = (double)std::max<long long>(bytes, 1);
        const unsigned want = (unsigned)std::min<long long>(kBlocks, std::max<long long>(L.n, bytes / (256 * 16 * 4)));
        unsigned at = 0;
        for (int k = 0; k < L.n; k++) { Desc& D = L.e[k]; D.blk0 = at; D.nblk = std::max(1u, (unsigned)((double)D.used * D.rows / total * want)); at += D.nblk; }
        nblocks = at;
    }
    // the same walk on the CPU (tests only: both sides of every descriptor must be host memory)
    void run_on_host() { finish(); for (unsigned b = 0; b < nblocks; b++) for (unsigned t = 0; t < 256; t++) block_work(L, b, t); }
    template <int TAG> hipError_t launch(hipStream_t s) {
        if (L.n == 0) return hipSuccess;
        finish();
        copy_list_kernel<TAG><<<dim3(nblocks), 256, 0, s>>>(L);
        return hipGetLastError();
    }
};
}  // namespace gfcopy
