// micro-benchmark of the in-register 16x16 Cholesky + inverse of ba_step (one wavefront): cycles per call
#pragma clang fp contract(fast)
#include <cstdio>
#include <vector>
#include <hip/hip_runtime.h>
#include "../ground-fusion_amd/csrc/gf_ba_kernels.hpp"
using namespace gfb;
template <int MODE> __global__ void k(const double* A, double* out, long long* cyc, int reps) {
    __shared__ double s_P[136], s_inv[16 * 17], s_rd[16];
    const int lane = threadIdx.x;
    long long t = 0;
    double acc = 0;
    for (int it = 0; it < reps; it++) {
        for (int i = lane; i < 256; i += 64) if ((i & 15) <= (i >> 4)) s_P[pk(i >> 4, i & 15)] = A[i] + (i % 17 == 0 ? it * 1e-9 : 0.0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        const long long t0 = clock64();
        bool good;
        good = wave_chol16_fused(s_P, 0, 16, s_inv, lane);
        t += clock64() - t0;
        acc += s_P[lane] + (good ? 1 : 0) + s_inv[lane % 16 * 17 + 3];
    }
    out[lane] = acc;
    if (lane == 0) cyc[blockIdx.x] = t / reps;
}
int main() {
    std::vector<double> A(256);
    for (int r = 0; r < 16; r++) for (int c = 0; c < 16; c++) A[r * 16 + c] = (r == c ? 20.0 : 0.0) + 1.0 / (1 + r + c);
    double *dA, *dout; long long* dc;
    hipMalloc(&dA, 2048); hipMalloc(&dout, 512); hipMalloc(&dc, 64);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice);
    long long c[2];
    for (int rep = 0; rep < 2; rep++) {
        k<0><<<1, 64>>>(dA, dout, dc, 100); hipMemcpy(&c[0], dc, 8, hipMemcpyDeviceToHost);
        k<1><<<1, 64>>>(dA, dout, dc, 100); hipMemcpy(&c[1], dc, 8, hipMemcpyDeviceToHost);
    }
    printf("cycles per 16x16 block (fused factor + inverse): %lld / %lld\n", c[0], c[1]);
    return 0;
}
