// gf_ba_marg.hpp — device side of the marginalisation prior (MarginalizationInfo::preMarginalize / marginalize,
// factor/marginalization_factor.cpp:119-308, as driven by estimator.cpp:3334-3631).
//
// The dropped-frame factors are linearised by the same kernels as the solver (ba_linearize_visual / ba_linearize_misc)
// with a marginalisation column map: columns [0, mp) = dropped pose/speed-bias block(s), [mp, mp+n) = kept blocks,
// eliminated columns = features starting at frame 0.  ba_marg_finish then forms the Schur complement and its
// thresholded symmetric eigen-decomposition:
//   A_mm = [[P, Q],[Q^T, D]] (D diagonal: one inverse depth per feature) is eliminated block-wise — D by thresholded
//   reciprocals through an MFMA GEMM, P (<= 15x15) through a Jacobi eigen pseudo-inverse — which equals the reference's
//   eigen pseudo-inverse of A_mm (eps 1e-8, :278-283) whenever no eigenvalue is thresholded; the kept n x n system is
//   decomposed by a parallel cyclic Jacobi in LDS, giving linearized_jacobians = sqrt(S) V^T and
//   linearized_residuals = sqrt(S^-1) V^T b (:294-302).
#pragma once
#include "gf_ba_kernels.hpp"

namespace gfb {

struct MargInfo { int mp, nfe, n, valid; };  // dropped non-feature dims, dropped features, kept dims

__global__ void __launch_bounds__(256) ba_zero_other(Win w) {
    const Dims d = w.d;
    const int b = blockIdx.x;
    const int o = 1 - w.st[b].cur;
    double* H = w.H + ((size_t)o * d.B + b) * d.RP * d.RP;
    for (int i = threadIdx.x; i < d.RP * d.RP; i += 256) H[i] = 0.0;
    double* g = w.g + ((size_t)o * d.B + b) * d.RP;
    for (int i = threadIdx.x; i < d.RP; i += 256) g[i] = 0.0;
    if (threadIdx.x == 0) w.cost[(size_t)o * d.B + b] = 0.0;
}

// cyclic Jacobi eigen-decomposition of the symmetric n x n matrix A (row-major, leading dim n) by a 512-thread block;
// V (n x n) receives the eigenvectors in its columns, the eigenvalues end on A's diagonal.  Round-robin pair ordering.
__device__ inline void block_jacobi_eig(double* A, double* V, int n, double* s_c, double* s_s, int* s_p, int* s_q, int* s_flag, int tid, int nthreads) {
    for (int i = tid; i < n * n; i += nthreads) V[i] = (i / n == i % n) ? 1.0 : 0.0;
    const int ne = (n + 1) & ~1, half = ne / 2;
    __syncthreads();
    for (int sweep = 0; sweep < 30; sweep++) {
        if (tid == 0) *s_flag = 0;
        __syncthreads();
        for (int round = 0; round < ne - 1; round++) {
            // pairing: positions 0..ne-1 on a ring, position 0 fixed
            for (int k = tid; k < half; k += nthreads) {
                int a = (k == 0) ? 0 : 1 + (k - 1 + round) % (ne - 1);
                int bq = 1 + (ne - 1 - k - 1 + round) % (ne - 1);
                int p = min(a, bq), q = max(a, bq);
                double c = 1.0, s = 0.0;
                if (q < n) {
                    const double apq = A[p * n + q];
                    const double app = A[p * n + p], aqq = A[q * n + q];
                    if (fabs(apq) > 1e-300 && fabs(apq) > 1e-17 * sqrt(fabs(app * aqq))) {
                        const double tau = (aqq - app) / (2.0 * apq);
                        const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        c = 1.0 / sqrt(1.0 + t * t); s = t * c;
                        *s_flag = 1;
                    }
                } else { p = -1; }
                s_p[k] = p; s_q[k] = q; s_c[k] = c; s_s[k] = s;
            }
            __syncthreads();
            // rows: A <- J^T A
            for (int e = tid; e < half * n; e += nthreads) {
                const int k = e / n, j = e - k * n, p = s_p[k], q = s_q[k];
                if (p < 0 || s_s[k] == 0.0) continue;
                const double c = s_c[k], s = s_s[k], x = A[p * n + j], y = A[q * n + j];
                A[p * n + j] = c * x - s * y; A[q * n + j] = s * x + c * y;
            }
            __syncthreads();
            // columns: A <- A J, V <- V J
            for (int e = tid; e < half * n; e += nthreads) {
                const int k = e / n, i = e - k * n, p = s_p[k], q = s_q[k];
                if (p < 0 || s_s[k] == 0.0) continue;
                const double c = s_c[k], s = s_s[k];
                double x = A[i * n + p], y = A[i * n + q];
                A[i * n + p] = c * x - s * y; A[i * n + q] = s * x + c * y;
                x = V[i * n + p]; y = V[i * n + q];
                V[i * n + p] = c * x - s * y; V[i * n + q] = s * x + c * y;
            }
            __syncthreads();
        }
        if (!*s_flag) break;
        __syncthreads();
    }
}

struct MargOut { double* J; double* r; };  // [B][NPRI*NPRI], [B][NPRI]

// One 512-thread block per window.  M = H[1-cur] (ld RP) holds the dropped-block + kept-block normal equations, g[1-cur] the
// right-hand side, efac[1-cur] the per-factor products of the eliminated feature columns.
__global__ void __launch_bounds__(512) ba_marg_finish(Win w, StepBufs sb, const MargInfo* info, MargOut out) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double sred[512];
    __shared__ double sP[15 * 15], sPV[15 * 15], sPinv[15 * 15], sbp[16];
    __shared__ double s_c[64], s_s[64];
    __shared__ int s_p[64], s_q[64], s_flag;
    const Dims d = w.d;
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const MargInfo mi = info[b];
    if (!mi.valid) return;
    const SolverState& st = w.st[b];
    const int o = 1 - st.cur, RP = d.RP;
    const int mp = mi.mp, NE = mi.nfe, n = mi.n, R = mp + n;
    double* M = w.H + ((size_t)o * d.B + b) * RP * RP;
    double* bv = w.g + ((size_t)o * d.B + b) * RP;
    const double* efac = w.efac + ((size_t)o * d.B + b) * d.NV * EF;
    const int* colf = w.colf + (size_t)b * d.NFB;   // marginalisation maps (the caller swapped them in)
    const int* cole = w.cole + (size_t)b * d.F;
    double* Et = sb.Et + (size_t)b * d.FP * RP; double* Es = sb.Es + (size_t)b * d.FP * RP;
    double* ete = sb.ete + (size_t)b * d.FP; double* etb = sb.etb + (size_t)b * d.FP;
    const double eps = 1e-8;
    // ---- rows of the eliminated feature columns
    for (int i = tid; i < NE * RP; i += 512) Et[i] = 0.0;
    __syncthreads();
    const int nf = w.nfeat[b];
    for (int f = wave; f < nf; f += 8) {
        const int e = cole[f];
        if (e < 0) continue;
        const int p0 = w.feat_ptr[(size_t)b * (d.F + 1) + f], p1 = w.feat_ptr[(size_t)b * (d.F + 1) + f + 1];
        double a = 0, c = 0;
        for (int p = p0; p < p1; p++) {
            const int k = w.feat_fac[(size_t)b * d.NV + p];
            const double* ef = efac + (size_t)k * EF;
            if (lane == 0) { a += ef[19]; c += ef[20]; }
            if (lane < 19) {
                const size_t kk = (size_t)b * d.NV + k;
                const int fi = w.vis_i[kk], fj = w.vis_j[kk];
                const int blk = lane < 6 ? fb_pose(fi) : lane < 12 ? fb_pose(fj) : lane == 12 ? fb_td(d.NP) : fb_ex(d.NP);
                const int oo = lane < 6 ? lane : lane < 12 ? lane - 6 : lane == 12 ? 0 : lane - 13;
                const int c0 = colf[blk];
                if (c0 >= 0) Et[(size_t)e * RP + c0 + oo] += ef[lane];
            }
        }
        if (lane == 0) { ete[e] = a; etb[e] = c; }
    }
    __syncthreads();
    // ---- eliminate the features: M -= sum_f w_f^T w_f / d_f, bv -= sum_f w_f b_f / d_f  (d_f <= eps: no information, dropped)
    for (int i = tid; i < NE * RP; i += 512) {
        const int e = i / RP, c = i - e * RP;
        const double df = ete[e];
        Es[i] = (c < R && df > eps) ? Et[i] / sqrt(df) : 0.0;
    }
    for (int i = NE * RP + tid; i < ((NE + 3) & ~3) * RP; i += 512) Es[i] = 0.0;
    __syncthreads();
    for (int c = tid; c < R; c += 512) {
        double v = bv[c];
        for (int e = 0; e < NE; e++) { const double df = ete[e]; if (df > eps) v -= Es[(size_t)e * RP + c] * (etb[e] / sqrt(df)); }
        bv[c] = v;
    }
    {
        const int nt = (R + 15) / 16, ntiles = nt * (nt + 1) / 2, nk = (NE + 3) / 4;
        for (int t = wave; t < ntiles; t += 8) {
            int ti = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
            while (ti * (ti + 1) / 2 > t) ti--;
            while ((ti + 1) * (ti + 2) / 2 <= t) ti++;
            const int tk = t - ti * (ti + 1) / 2;
            d4 acc = {0, 0, 0, 0};
            const double* pa = Es + (size_t)(lane >> 4) * RP + 16 * ti + (lane & 15);
            const double* pb = Es + (size_t)(lane >> 4) * RP + 16 * tk + (lane & 15);
            for (int k = 0; k < nk; k++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[(size_t)4 * k * RP], pb[(size_t)4 * k * RP], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 16 * ti + (lane >> 4) + 4 * r, col = 16 * tk + (lane & 15);
                if (row < R && col < R) {
                    M[(size_t)row * RP + col] -= acc[r];
                    if (ti != tk) M[(size_t)col * RP + row] -= acc[r];
                }
            }
        }
    }
    __syncthreads();
    // ---- pseudo-inverse of the dropped pose / speed-bias block (eigenvalues <= eps are dropped)
    for (int i = tid; i < mp * mp; i += 512) { const int r = i / mp, c = i % mp; sP[i] = 0.5 * (M[(size_t)r * RP + c] + M[(size_t)c * RP + r]); }
    __syncthreads();
    block_jacobi_eig(sP, sPV, mp, s_c, s_s, s_p, s_q, &s_flag, tid, 512);
    __syncthreads();
    for (int i = tid; i < mp * mp; i += 512) {
        const int r = i / mp, c = i % mp;
        double v = 0;
        for (int k = 0; k < mp; k++) { const double ev = sP[k * mp + k]; if (ev > eps) v += sPV[r * mp + k] * sPV[c * mp + k] / ev; }
        sPinv[i] = v;
    }
    __syncthreads();
    if (tid < mp) { double v = 0; for (int k = 0; k < mp; k++) v += sPinv[tid * mp + k] * bv[k]; sbp[tid] = v; }
    __syncthreads();
    // ---- kept system: A_r = M_kk - M_kp Pinv M_pk (LDS), b_r = b_k - M_kp Pinv b_p
    double* A = smem;            // n x n
    double* V = smem + n * n;    // n x n
    double* br = sb.rhs + (size_t)b * RP;
    for (int i = tid; i < n * n; i += 512) {
        const int r = i / n, c = i % n;
        double v = M[(size_t)(mp + r) * RP + mp + c];
        for (int a = 0; a < mp; a++) {
            double t = 0;
            for (int k = 0; k < mp; k++) t += sPinv[a * mp + k] * M[(size_t)k * RP + mp + c];
            v -= M[(size_t)(mp + r) * RP + a] * t;
        }
        A[i] = v;
    }
    for (int r = tid; r < n; r += 512) { double v = bv[mp + r]; for (int a = 0; a < mp; a++) v -= M[(size_t)(mp + r) * RP + a] * sbp[a]; br[r] = v; }
    __syncthreads();
    for (int i = tid; i < n * n; i += 512) { const int r = i / n, c = i % n; if (c > r) { const double v = 0.5 * (A[i] + A[c * n + r]); A[i] = v; A[c * n + r] = v; } }
    __syncthreads();
    block_jacobi_eig(A, V, n, s_c, s_s, s_p, s_q, &s_flag, tid, 512);
    __syncthreads();
    // ---- linearized_jacobians = sqrt(S) V^T, linearized_residuals = sqrt(S^-1) V^T b   (rows ordered by ascending eigenvalue like Eigen)
    int* rank = reinterpret_cast<int*>(sred);
    for (int k = tid; k < n; k += 512) {
        const double ev = A[k * n + k];
        int rk = 0;
        for (int j = 0; j < n; j++) { const double ej = A[j * n + j]; if (ej < ev || (ej == ev && j < k)) rk++; }
        rank[k] = rk;
    }
    __syncthreads();
    double* J = out.J + (size_t)b * d.NPRI * d.NPRI;
    double* rr = out.r + (size_t)b * d.NPRI;
    for (int i = tid; i < n * n; i += 512) {
        const int k = i / n, j = i % n;
        const double ev = A[k * n + k];
        J[(size_t)rank[k] * n + j] = ev > eps ? sqrt(ev) * V[j * n + k] : 0.0;
    }
    for (int k = tid; k < n; k += 512) {
        const double ev = A[k * n + k];
        double vb = 0;
        for (int j = 0; j < n; j++) vb += V[j * n + k] * br[j];
        rr[rank[k]] = ev > eps ? sqrt(1.0 / ev) * vb : 0.0;
    }
}

}  // namespace gfb
