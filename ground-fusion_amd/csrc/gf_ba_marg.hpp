// gf_ba_marg.hpp — device side of the marginalisation prior (MarginalizationInfo::preMarginalize / marginalize,
// factor/marginalization_factor.cpp:119-308, as driven by estimator.cpp:3334-3631).
//
// The dropped-frame factors are linearised by the same kernels as the solver (ba_linearize_visual_win / ba_linearize_misc_win)
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

// Cyclic Jacobi eigen-decomposition of the symmetric n x n matrix A (row-major, leading dim n) by `nthreads` threads of one block
// (nthreads == 64: a single wavefront, barriers degrade to wave barriers).  V (n x n) receives the eigenvectors in its columns, the
// eigenvalues end on A's diagonal.  Round-robin pairing; every round applies A <- J^T A J on 2x2 blocks (pair k x pair k') in one
// pass, so a round costs two barriers.
template <bool WAVE>
__device__ inline void jacobi_sync() {
    if (WAVE) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    else __syncthreads();
}
template <bool WAVE>
__device__ inline void block_jacobi_eig(double* A, double* V, int n, double* s_c, double* s_s, int* s_p, int* s_q, int* s_flag, int tid, int nthreads) {
    for (int i = tid; i < n * n; i += nthreads) V[i] = (i / n == i % n) ? 1.0 : 0.0;
    const int ne = (n + 1) & ~1, half = ne / 2;
    double dmax = 0;   // rotations below machine precision of the largest diagonal are noise and are skipped (norm-relative criterion)
    for (int i = 0; i < n; i++) dmax = fmax(dmax, fabs(A[i * n + i]));
    const double skip = 1e-16 * dmax;
    jacobi_sync<WAVE>();
    for (int sweep = 0; sweep < 30; sweep++) {
        if (tid == 0) *s_flag = 0;
        jacobi_sync<WAVE>();
        for (int round = 0; round < ne - 1; round++) {
            for (int k = tid; k < half; k += nthreads) {  // pairing: positions on a ring, position 0 fixed
                const int a = (k == 0) ? 0 : 1 + (k - 1 + round) % (ne - 1);
                const int bq = 1 + (ne - 1 - k - 1 + round) % (ne - 1);
                int p = min(a, bq), q = max(a, bq);
                double c = 1.0, s = 0.0;
                if (q < n) {
                    const double apq = A[p * n + q], app = A[p * n + p], aqq = A[q * n + q];
                    if (fabs(apq) > skip && fabs(apq) > 1e-15 * sqrt(fabs(app * aqq))) {
                        const double tau = (aqq - app) / (2.0 * apq);
                        const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        c = 1.0 / sqrt(1.0 + t * t); s = t * c;
                        *s_flag = 1;
                    }
                } else { q = -1; }   // odd n: the unpaired index idles (identity rotation)
                s_p[k] = p; s_q[k] = q; s_c[k] = c; s_s[k] = s;
            }
            jacobi_sync<WAVE>();
            // A <- J^T A J: block (k, k2) = rows {p,q} x cols {p2,q2}
            for (int e = tid; e < half * half; e += nthreads) {
                const int k = e / half, k2 = e - k * half;
                const int p = s_p[k], q = s_q[k], p2 = s_p[k2], q2 = s_q[k2];
                const double c = s_c[k], s = s_s[k], c2 = s_c[k2], s2 = s_s[k2];
                if (s == 0.0 && s2 == 0.0) continue;
                double a00 = A[p * n + p2], a01 = q2 >= 0 ? A[p * n + q2] : 0.0, a10 = q >= 0 ? A[q * n + p2] : 0.0, a11 = (q >= 0 && q2 >= 0) ? A[q * n + q2] : 0.0;
                // left rotation (rows p,q): [r0; r1] = [c -s; s c] [a0; a1]
                double b00 = c * a00 - s * a10, b01 = c * a01 - s * a11, b10 = s * a00 + c * a10, b11 = s * a01 + c * a11;
                // right rotation (cols p2,q2): [x y] <- [x y] [c2 s2; -s2 c2]
                a00 = c2 * b00 - s2 * b01; a01 = s2 * b00 + c2 * b01; a10 = c2 * b10 - s2 * b11; a11 = s2 * b10 + c2 * b11;
                A[p * n + p2] = a00;
                if (q2 >= 0) A[p * n + q2] = a01;
                if (q >= 0) A[q * n + p2] = a10;
                if (q >= 0 && q2 >= 0) A[q * n + q2] = a11;
            }
            // V <- V J
            for (int e = tid; e < n * half; e += nthreads) {
                const int i = e / half, k2 = e - i * half;
                const int p2 = s_p[k2], q2 = s_q[k2];
                const double c2 = s_c[k2], s2 = s_s[k2];
                if (s2 == 0.0 || q2 < 0) continue;
                const double x = V[i * n + p2], y = V[i * n + q2];
                V[i * n + p2] = c2 * x - s2 * y; V[i * n + q2] = s2 * x + c2 * y;
            }
            jacobi_sync<WAVE>();
        }
        if (!*s_flag) break;
        jacobi_sync<WAVE>();
    }
}

constexpr int MPMAX = 20;   // dropped pose + speed-bias (+ receiver clock blocks) columns
// Inverse of a symmetric positive definite n x n block (n <= MPMAX) by Gauss-Jordan without pivoting.  T: n x 2n work area.  The elimination runs on the whole
// block (two entries of [A | I] per thread, two block barriers per pivot: as a one-wavefront job with 13 entries per lane and a division by the run-time row length
// per entry and pivot it was 35 k cycles of the kernel at n = 15), the symmetrisation and the norm on wavefront 0.  True and Ainv when every pivot is positive and
// 1 / |A^-1|_F > eps: then lambda_min(A) >= 1 / |A^-1|_2 > eps, no eigenvalue falls under the reference's truncation threshold
// (marginalization_factor.cpp:279-283) and the pseudo-inverse it builds from the eigen-decomposition is this inverse.  Otherwise the caller takes the eigen route.
__device__ inline bool block_spd_eliminate(const double* A, double* T, int n, int tid) {   // every thread of the 512; ends behind a block barrier
    constexpr int Q = (MPMAX * 2 * MPMAX + 511) / 512;
    const int n2 = 2 * n, tot = n * n2;
    int rr[Q], cc[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) { const int i = tid + 512 * q; rr[q] = i / max(n2, 1); cc[q] = i - rr[q] * n2; }
#pragma unroll
    for (int q = 0; q < Q; q++) { const int i = tid + 512 * q; if (i < tot) T[i] = cc[q] < n ? A[rr[q] * n + cc[q]] : (cc[q] - n == rr[q] ? 1.0 : 0.0); }
    __syncthreads();
    for (int k = 0; k < n; k++) {
        const double piv = T[k * n2 + k];   // the same word for every thread: all of them leave together
        if (!(piv > 0.0)) return false;
        const double dinv = 1.0 / piv;
        double fr[Q], pv[Q], cur[Q];
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int i = tid + 512 * q;
            if (i < tot) { fr[q] = T[rr[q] * n2 + k]; pv[q] = T[k * n2 + cc[q]]; cur[q] = T[i]; }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int i = tid + 512 * q;
            if (i < tot) T[i] = rr[q] == k ? cur[q] * dinv : cur[q] - fr[q] * dinv * pv[q];
        }
        __syncthreads();
    }
    return true;
}
__device__ inline bool wave_spd_finish(const double* T, double* Ainv, int n, double eps, int lane) {   // one wavefront
    const int n2 = 2 * n;
    double f2 = 0.0;
    for (int i = lane; i < n * n; i += 64) { const int r = i / n, c = i - r * n; const double v = 0.5 * (T[r * n2 + n + c] + T[c * n2 + n + r]); Ainv[i] = v; f2 += v * v; }
    f2 = wave_sum_f64(f2);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    return f2 > 0.0 && 1.0 / sqrt(f2) > eps;
}

struct MargOut { double* J; double* r; };  // [B][NPRI*NPRI], [B][NPRI]

// One 512-thread block per window.  M = H[1-cur] (ld RP) holds the prior / IMU / wheel part of the dropped-block + kept-block normal
// equations and g[1-cur] of the right-hand side (ba_linearize_misc_win with the marginalisation's column map), Vc[1-cur] the visual part
// in compact columns (use_vc: MARGIN_OLD), Et / ete / etb[1-cur] the compact rows of the eliminated feature columns.
template <bool GS>   // GS: A and V of the kept system live in global memory (sb.Mg) -- priors larger than 96 columns
__global__ void __launch_bounds__(512) ba_marg_finish(Win w, StepBufs sb, const MargInfo* info, MargOut out, int use_vc, double piv_eps, int ls_rhs, int lds_doubles) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double sP[MPMAX * MPMAX], sPV[MPMAX * MPMAX], sPinv[MPMAX * MPMAX], sbp[MPMAX];
    __shared__ double s_c[64], s_s[64];
    __shared__ int s_p[64], s_q[64], s_flag, s_fastinv;
    __shared__ double s_dg[2][256];   // diagonal of the kept system during the pivoted factorisation (n <= 256: checked by the host)
    const Dims d = w.d;
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const MargInfo mi = info[b];
#ifdef GF_PROFILE_STEP
    const long long t_entry = clock64();
#endif
    if (!uni(mi.valid)) return;
    const SolverState& st = w.st[b];
    const int o = 1 - uni(st.cur), RP = d.RP;
    const int mp = uni(mi.mp), NE = uni(mi.nfe), n = uni(mi.n);   // per-window scalars: scalar registers, scalar loop bounds
    double* M = w.H + ((size_t)o * d.B + b) * RP * RP;
    double* bv = w.g + ((size_t)o * d.B + b) * RP;
    const int ECW = d.ECW;
    const double* Et = sb.Et + ((size_t)o * d.B + b) * d.FP * ECW;   // compact rows built by ba_build_et
    double* Es = sb.Es + (size_t)b * d.FP * ECW;
    __shared__ int s_cmap[256];
    if (tid < ECW) s_cmap[tid] = compact_to_col(tid, w.colf + (size_t)b * d.NFB, d.NP, -1);   // marginalisation column map (swapped in by the caller)
    const double* ete = sb.ete + ((size_t)o * d.B + b) * d.FP; const double* etb = sb.etb + ((size_t)o * d.B + b) * d.FP;
    const double eps = 1e-8;
    __syncthreads();
    if (use_vc) {   // M += visual part: every compact entry maps to its own entry of M (one writer each)
        const double* Vc = w.Vc + ((size_t)o * d.B + b) * d.NVC;
        const int RHSK = 6 * d.NP + 7;
        for (int idx = tid; idx < RHSK * (RHSK + 1) / 2; idx += 512) {
            const int ka = tri_row(idx), kb = idx - ka * (ka + 1) / 2;
            const int r = s_cmap[ka], c = s_cmap[kb];
            if (r >= 0 && c >= 0) M[(size_t)max(r, c) * RP + min(r, c)] += Vc[idx];
        }
        for (int kb = tid; kb < RHSK; kb += 512) { const int c = s_cmap[kb]; if (c >= 0) bv[c] += Vc[(size_t)RHSK * (RHSK + 1) / 2 + kb]; }
        __syncthreads();
    }
#ifdef GF_PROFILE_STEP
#define GF_MST(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && sb.stamps) sb.stamps[32 + (i)] = clock64(); } while (0)
#else
#define GF_MST(i) do { } while (0)
#endif
    GF_MST(0);
    // ---- eliminate the features: M -= sum_f w_f^T w_f / d_f, bv -= sum_f w_f b_f / d_f  (d_f <= eps: no information, dropped)
    double* fe = sb.u + (size_t)b * sb.VS;   // per-feature 1/sqrt(d_f) (0: dropped)
    double* feb = sb.yv + (size_t)b * sb.VS;  // per-feature b_f/sqrt(d_f)  (two scratch vectors of VS = RP + FP entries each: F entries fit either, not both in one)
    for (int e = tid; e < NE; e += 512) { const double df = ete[e]; const double f = df > eps ? 1.0 / sqrt(df) : 0.0; fe[e] = f; feb[e] = f * etb[e]; }
    __syncthreads();
    // Es = diag(fe) Et on the kept compact columns (one wavefront per row, eight rows in flight: every load issued before the first store), then bv -= Es^T feb and
    // M -= Es^T Es.  Es lives in the block's LDS when it fits (the kept system that will occupy it does not exist yet): its column sums and the operands of the
    // rank update then come from LDS instead of L2 -- the two stages were serialised global-load latencies (46 k + 31 k cycles at 86 kept columns) -- else in global
    // memory as before (sb.Es).  Same sums in the same order either way.
    const int NE4 = (NE + 3) & ~3;
    auto eliminate = [&](double* EsP) {
        constexpr int EN = GS ? 4 : 2;   // 64-column pieces of a compact row
        for (int e0 = wave; e0 < NE4; e0 += 64) {
            double ev[8][EN], fv[8];
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const int e = e0 + 8 * m;
                fv[m] = e < NE ? fe[e] : 0.0;
#pragma unroll
                for (int q = 0; q < EN; q++) { const int k = lane + 64 * q; ev[m][q] = (e < NE && k < ECW) ? Et[(size_t)e * ECW + k] : 0.0; }
            }
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const int e = e0 + 8 * m;
                if (e >= NE4) continue;
#pragma unroll
                for (int q = 0; q < EN; q++) { const int k = lane + 64 * q; if (k < ECW) EsP[(size_t)e * ECW + k] = (e < NE && s_cmap[k] >= 0) ? ev[m][q] * fv[m] : 0.0; }
            }
        }
        __syncthreads();
        for (int k = tid; k < ECW; k += 512) {   // bv -= Es^T feb: the sum runs in row order as before, the loads of eight rows at a time are issued together
            const int c = s_cmap[k];
            if (c < 0) continue;
            double v = bv[c];
            for (int e0 = 0; e0 < NE; e0 += 8) {
                double a[8], f[8];
#pragma unroll
                for (int m = 0; m < 8; m++) { const int e = min(e0 + m, NE - 1); a[m] = EsP[(size_t)e * ECW + k]; f[m] = feb[e]; }
#pragma unroll
                for (int m = 0; m < 8; m++) if (e0 + m < NE) v -= a[m] * f[m];
            }
            bv[c] = v;
        }
        GF_MST(1);
        const int nt = ECW / 16, ntiles = nt * (nt + 1) / 2, nk = (NE + 3) / 4;
        for (int t = wave; t < ntiles; t += 8) {
            const int ti = tri_row(t), tk = t - ti * (ti + 1) / 2;
            d4 acc = {0, 0, 0, 0};
            const double* pa = EsP + (size_t)(lane >> 4) * ECW + 16 * ti + (lane & 15);
            const double* pb = EsP + (size_t)(lane >> 4) * ECW + 16 * tk + (lane & 15);
            for (int k = 0; k < nk; k++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[(size_t)4 * k * ECW], pb[(size_t)4 * k * ECW], acc, 0, 0, 0);
            const int col = s_cmap[16 * tk + (lane & 15)];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = s_cmap[16 * ti + (lane >> 4) + 4 * r];
                // M holds its lower triangle; the marginalisation column order is not monotone in the compact index, so order per element
                if (row >= 0 && col >= 0 && (ti != tk || (lane >> 4) + 4 * r >= (lane & 15))) M[(size_t)max(row, col) * RP + min(row, col)] -= acc[r];
            }
        }
    };
    if (!GS && NE4 * ECW <= lds_doubles) eliminate(smem);
    else eliminate(Es);
    __syncthreads();
    GF_MST(2);
    // ---- pseudo-inverse of the dropped pose / speed-bias block (eigenvalues <= eps are dropped)
    for (int i = tid; i < mp * mp; i += 512) { const int r = i / mp, c = i % mp; sP[i] = M[(size_t)max(r, c) * RP + min(r, c)]; }
    __syncthreads();
    // well-conditioned block (the usual case): plain inverse; else eigen-decomposition with the truncation of the reference
    {
        double* Tw = GS ? sb.Mg + (size_t)blockIdx.x * sb.MgStride : smem;
        const bool elim = block_spd_eliminate(sP, Tw, mp, tid);
        if (wave == 0) {
            const bool fast = elim && wave_spd_finish(Tw, sPinv, mp, eps, lane);
            if (lane == 0) s_fastinv = fast ? 1 : 0;
        }
    }
    __syncthreads();
    if (!s_fastinv) {
        if (wave == 0) block_jacobi_eig<true>(sP, sPV, mp, s_c, s_s, s_p, s_q, &s_flag, lane, 64);
        __syncthreads();
        for (int i = tid; i < mp * mp; i += 512) {
            const int r = i / mp, c = i % mp;
            double v = 0;
            for (int k = 0; k < mp; k++) { const double ev = sP[k * mp + k]; if (ev > eps) v += sPV[r * mp + k] * sPV[c * mp + k] / ev; }
            sPinv[i] = v;
        }
        __syncthreads();
    }
    if (tid < mp) { double v = 0; for (int k = 0; k < mp; k++) v += sPinv[tid * mp + k] * bv[k]; sbp[tid] = v; }
    __syncthreads();
    GF_MST(3);
    // ---- kept system: A_r = M_kk - M_kp Pinv M_pk (LDS), b_r = b_k - M_kp Pinv b_p
    double* A = GS ? sb.Mg + (size_t)blockIdx.x * sb.MgStride : smem;   // n x n
    double* V = A + n * n;       // n x n
    double* br = sb.rhs + (size_t)b * RP;
    double* T = V;   // T = Pinv * M_pk (mp x n), staged in the not-yet-used V area
    // M_kp (n x mp) goes through LDS once (rows of M are contiguous in a; the product below would read them n times each from global memory), and M_kk is copied
    // row by row with its mirror image written next to it -- the lower triangle is what M holds, reading the upper one as M[c][r] made half of the loads strided by
    // a row (57 k cycles for this stage at n = 86, mp = 15; profiles/README.md round 6).  Sums and their order are unchanged.
    double* Mkp = V + (size_t)mp * n;   // n x mp
    for (int i = tid; i < n * mp; i += 512) { const int r = i / mp, a = i - r * mp; Mkp[i] = M[(size_t)(mp + r) * RP + a]; }
    for (int i = tid; i < n * n; i += 512) {
        const int r = i / n, c = i - r * n;
        if (c <= r) { const double v = M[(size_t)(mp + r) * RP + mp + c]; A[i] = v; A[c * n + r] = v; }
    }
    __syncthreads();
    for (int i = tid; i < mp * n; i += 512) {
        const int a = i / n, c = i % n;
        double t = 0;
        for (int k = 0; k < mp; k++) t += sPinv[a * mp + k] * Mkp[c * mp + k];
        T[i] = t;
    }
    __syncthreads();
    {   // A -= M_kp T on 4 x 4 register tiles (eight LDS loads for sixteen products instead of two per product); a thread's rows / columns are nq apart, so
        // that the lanes of a wavefront read consecutive doubles of a row of T and share their rows of M_kp.  Per entry the same subtractions in the same order.
        const int nq = (n + 3) >> 2;
        for (int t = tid; t < nq * nq; t += 512) {
            const int tr = t / nq, tc = t - tr * nq;
            int ri[4], ci[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { ri[q] = min(tr + q * nq, n - 1); ci[q] = min(tc + q * nq, n - 1); }
            double v[4][4];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int q = 0; q < 4; q++) v[a][q] = A[ri[a] * n + ci[q]];
            for (int k = 0; k < mp; k++) {
                double m[4], tt[4];
#pragma unroll
                for (int q = 0; q < 4; q++) { m[q] = Mkp[ri[q] * mp + k]; tt[q] = T[k * n + ci[q]]; }
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int q = 0; q < 4; q++) v[a][q] -= m[a] * tt[q];
            }
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int q = 0; q < 4; q++) if (tr + a * nq < n && tc + q * nq < n) A[ri[a] * n + ci[q]] = v[a][q];
        }
    }
    for (int r = tid; r < n; r += 512) { double v = bv[mp + r]; for (int a = 0; a < mp; a++) v -= Mkp[r * mp + a] * sbp[a]; br[r] = v; }
    __syncthreads();
    for (int i = tid; i < n * n; i += 512) { const int r = i / n, c = i % n; if (c > r) { const double v = 0.5 * (A[i] + A[c * n + r]); A[i] = v; A[c * n + r] = v; } }
    __syncthreads();
    GF_MST(4);
    // ---- square-root factor of the kept system.  The reference takes sqrt(S) V^T from an eigen-decomposition with eigenvalues <= eps
    // dropped (marginalization_factor.cpp:294-302); any J with J^T J = A_r and J^T r = b_r is the same prior to the solver (only J^T J,
    // J^T r and |r|^2 enter Ceres), so a rank-revealing (diagonally pivoted) Cholesky with the same absolute threshold is used:
    // A_r = P L L^T P^T, J = L^T P^T (rows beyond the detected rank are zero), r = L^-1 P^T b (forward substitution rides along).
    // Rows and columns are swapped PHYSICALLY when a pivot is chosen (one pass over the two rows / columns), so that everything after it addresses
    // A[i][j] directly.  The matrix is kept full and symmetric: the pivot column is read as the pivot ROW (contiguous), the trailing update
    // A[i][j] -= l_i l_j runs on a 2-D thread grid without index indirection, and every wavefront finds the pivot for itself (DPP row maxima + four
    // scalar reads), so a pivot step costs one block barrier, two when rows have to be swapped (the version that addressed A[perm[i]][perm[j]],
    // searched on one wavefront with shuffles and took an IEEE square root cost 4.8 k cycles per pivot).
    int* perm = reinterpret_cast<int*>(V);   // n ints: original index of the row / column now at position i
    double* zb = V + 256;                    // n doubles: permuted right-hand side being forward-substituted
    double* dinvs = V + 1280;                // 1 / L_kk per accepted pivot
    double* zr = V + 1792;                   // finished entries of the forward substitution
    // The pivot search reads the diagonal from a double-buffered copy in LDS: step k searches s_dg[k & 1], the trailing update of step k writes the
    // new diagonal into s_dg[(k + 1) & 1].  A wavefront that is through with its search may therefore start swapping / updating A while a slower one is
    // still searching -- what the slower one reads is never written in the same step -- and all wavefronts agree on the pivot without a barrier
    // between search and swap (reading A[i][i] itself there was a race: round-2 advisor finding).
    for (int i = tid; i < n; i += 512) { perm[i] = i; zb[i] = br[i]; s_dg[0][i] = A[i * n + i]; }
    __syncthreads();
    int rank = n;
    const int tx = tid & 31, ty = tid >> 5;   // 32 x 16 thread grid of the trailing update
    constexpr int kPB = 8;                    // pivots per block
    // Who walks the pivots of a block.  LDS variant (n <= 92): wavefront 0 alone -- search, swap and the left-looking column of a pivot are n-sized jobs, two
    // entries per lane, and with one wavefront the steps of a block are ordered by wavefront barriers (LDS accesses of one wavefront execute in order) instead
    // of two block barriers per pivot; the other seven wavefronts sleep at the block barrier until the block's update of the trailing matrix, which is theirs too
    // (2.9 k -> 1.2 k cycles per pivot at n = 86, profiles/README.md round 6).  Global-memory variant: all 512 threads with block barriers as before -- a store
    // and a later load of ANOTHER lane through the vector L1 are not ordered by a wavefront barrier.  The arithmetic is the same entry by entry in both.
    const int t0 = GS ? tid : lane;
    constexpr int tstr = GS ? 512 : 64;
    __shared__ int s_rank;
#ifdef GF_PROFILE_STEP
    long long pq[5] = {0, 0, 0, 0, 0}, pt = clock64();   // cycles of block 0 / thread 0 in: pivot search, swap, column, hand-over barrier, trailing update
#define GF_PQ(i) do { const long long t_ = clock64(); pq[i] += t_ - pt; pt = t_; } while (0)
#else
#define GF_PQ(i) do { } while (0)
#endif
#define GF_PIV_SYNC() do { if constexpr (GS) __syncthreads(); else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); } } while (0)
    for (int kb = 0; kb < n; kb += kPB) {     // kb: first pivot of the open block
        const int k1 = min(kb + kPB, n);
        if (GS || wave == 0)
        for (int k = kb; k < k1; k++) {
            // pivot: largest remaining diagonal entry, the lowest lane that holds it on ties (every wavefront computes the same answer)
            double best = -1.0; int bi = k;
            const double* dgo = s_dg[k & 1];
            double* dgn = s_dg[(k + 1) & 1];
            for (int i = k + lane; i < n; i += 64) { const double v = dgo[i]; if (v > best) { best = v; bi = i; } }
            double m = best;
#define GF_DPP_MAX(ctrl) do { const int lo_ = __builtin_amdgcn_mov_dpp(__double2loint(m), ctrl, 0xf, 0xf, true), hi_ = __builtin_amdgcn_mov_dpp(__double2hiint(m), ctrl, 0xf, 0xf, true); \
                              m = fmax(m, __hiloint2double(hi_, lo_)); } while (0)
            GF_DPP_MAX(0xB1); GF_DPP_MAX(0x4E); GF_DPP_MAX(0x141); GF_DPP_MAX(0x140);   // quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
#undef GF_DPP_MAX
            double piv = -1.0;
#pragma unroll
            for (int q = 0; q < 4; q++) piv = fmax(piv, __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(m), 16 * q), __builtin_amdgcn_readlane(__double2loint(m), 16 * q)));
            const unsigned long long hit = __ballot(best == piv);
            const int p = __builtin_amdgcn_readlane(bi, __ffsll((long long)hit) - 1);
            if (!(piv > piv_eps)) { rank = k; break; }
            GF_PQ(0);
            if (p != k) {   // symmetric swap k <-> p of the full matrix in one pass: thread t moves the four entries that involve t; the 2 x 2 corner by one thread
                // (every load of a thread before its first store: the compiler cannot tell the LDS arrays apart and would wait for each store before the next load)
                if constexpr (GS) {
                    for (int t = tid; t < n; t += 512) {
                        if (t == k || t == p) continue;
                        const double a0 = A[k * n + t], a1 = A[p * n + t], b0 = A[t * n + k], b1 = A[t * n + p];
                        A[k * n + t] = a1; A[p * n + t] = a0; A[t * n + k] = b1; A[t * n + p] = b0;
                    }
                    if (tid == 64) { const double dk = A[k * n + k], dp = A[p * n + p]; A[k * n + k] = dp; A[p * n + p] = dk; }   // A[k][p] == A[p][k] stay where they are
                    if (tid == 128) { const int tk = perm[k], tp = perm[p]; perm[k] = tp; perm[p] = tk; }
                    if (tid == 192) { const double zk = zb[k], zp = zb[p]; zb[k] = zp; zb[p] = zk; }
                } else {   // n <= 128: two entries per lane, the corner, perm and zb read by every lane (broadcast) and written by lane 0
                    double a0[2], a1[2], b0[2], b1[2];
                    bool on[2];
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const int t = lane + 64 * q;
                        on[q] = t < n && t != k && t != p;
                        const int tt = on[q] ? t : k;
                        a0[q] = A[k * n + tt]; a1[q] = A[p * n + tt]; b0[q] = A[tt * n + k]; b1[q] = A[tt * n + p];
                    }
                    const double dk = A[k * n + k], dp = A[p * n + p], zk = zb[k], zp = zb[p];
                    const int tk = perm[k], tp = perm[p];
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const int t = lane + 64 * q;
                        if (on[q]) { A[k * n + t] = a1[q]; A[p * n + t] = a0[q]; A[t * n + k] = b1[q]; A[t * n + p] = b0[q]; }
                    }
                    if (lane == 0) { A[k * n + k] = dp; A[p * n + p] = dk; perm[k] = tp; perm[p] = tk; zb[k] = zp; zb[p] = zk; }
                }
                GF_PIV_SYNC();
            }
            GF_PQ(1);
#ifdef GF_MARG_IEEE_SQRT   // conditioning experiments (scripts/gnss_replay_sensitivity.py): the arithmetically equivalent, correctly rounded variant
            double dinv = 1.0 / sqrt(piv);
#else
            // 1 / sqrt(piv): hardware seed, one plain Newton step, one with the residual 1 - piv y^2 formed exactly (product split by an fma), which leaves
            // the final rounding as the only error (an IEEE sqrt and a division cost ~500 cycles per pivot).  Accuracy matters here: a relative error d of 1 / L_kk
            // puts 2 d |l_i l_j| ~ 1e-7 (entries of 1e9) into the trailing block, which is the size of the weak GNSS directions the prior has to carry
            double dinv = __builtin_amdgcn_rsq(piv);
            dinv = dinv * (1.5 - 0.5 * piv * dinv * dinv);
            {
                const double h = piv * dinv, hl = __builtin_fma(piv, dinv, -h);
                const double e = __builtin_fma(-h, dinv, 1.0) - hl * dinv;
                dinv = __builtin_fma(0.5 * dinv, e, dinv);
            }
#endif
            // Left-looking inside blocks of kPB pivots: row k still lacks the updates of the pivots kb .. k - 1 of its block (rows kb .. k - 1 hold their finished,
            // scaled columns of L), so one thread per entry applies them now, in pivot order -- the same fma chain, entry by entry, as a trailing update after every
            // pivot -- and the trailing matrix is touched once per block instead of once per pivot.
            const double rk = zb[k] * dinv;
            double* rowk = A + (size_t)k * n;        // = column k (symmetric); becomes column k of L, scaled
            for (int i = k + 1 + t0; i < n; i += tstr) {
                double acc = rowk[i];
                double lc[kPB - 1], lk[kPB - 1];
#pragma unroll
                for (int q = 0; q < kPB - 1; q++) {   // all loads in flight at once (rows past the open block: row k - 1 again, weighted with zero: fma(-x, 0, acc) == acc)
                    const int c = min(kb + q, k - 1 < kb ? kb : k - 1);
                    lc[q] = A[(size_t)c * n + i];
                    const double v = A[(size_t)c * n + k];
                    lk[q] = kb + q < k ? v : 0.0;
                }
                const double dold = dgo[i == p ? k : i], zold = zb[i];   // the diagonal copy is not swapped (slower wavefronts may still search it): position p holds old row k
#pragma unroll
                for (int q = 0; q < kPB - 1; q++) acc = __builtin_fma(-lc[q], lk[q], acc);
                const double li = acc * dinv;
                rowk[i] = li;
                dgn[i] = __builtin_fma(-li, li, dold);
                zb[i] = __builtin_fma(-li, rk, zold);
            }
            if (t0 == 0) { dinvs[k] = dinv; zr[k] = rk; }
            GF_PIV_SYNC();
            GF_PQ(2);
        }
        if constexpr (!GS) {   // the other wavefronts learn from wavefront 0 whether the block went through
            if (tid == 0) s_rank = rank;
            __syncthreads();
            rank = s_rank;
        }
        GF_PQ(3);
        if (rank != n) break;
        {   // close the block: A[i][j] -= sum_c L[i][c] L[j][c], c in pivot order, 2 x 4 entries per thread
            // thread (tx, ty) owns rows i0 + 16 a and columns j0 + 32 q: the lanes of a wavefront read consecutive doubles (4 consecutive columns per lane cost 4-way bank conflicts)
            // One path for every tile: rows / columns past the matrix are clamped for the loads and skipped by the stores, columns of L past a short last block
            // weigh zero (fma(-x, 0, v) == v) -- a branch around the edge tiles would put every wavefront through both versions.
            for (int i0 = k1 + ty; i0 < n; i0 += 64)
                for (int j0 = k1 + tx; j0 < n; j0 += 64) {
                    double v[4][2], li[kPB][4], lj[kPB][2];
                    int ia[4], jq[2];
#pragma unroll
                    for (int a = 0; a < 4; a++) ia[a] = min(i0 + 16 * a, n - 1);
#pragma unroll
                    for (int q = 0; q < 2; q++) jq[q] = min(j0 + 32 * q, n - 1);
#pragma unroll
                    for (int c = 0; c < kPB; c++) {
                        const bool on = kb + c < k1;
                        const double* lc = A + (size_t)min(kb + c, k1 - 1) * n;
#pragma unroll
                        for (int a = 0; a < 4; a++) li[c][a] = lc[ia[a]];
#pragma unroll
                        for (int q = 0; q < 2; q++) { const double t = lc[jq[q]]; lj[c][q] = on ? t : 0.0; }
                    }
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int q = 0; q < 2; q++) v[a][q] = A[(size_t)ia[a] * n + jq[q]];
#pragma unroll
                    for (int c = 0; c < kPB; c++)
#pragma unroll
                        for (int a = 0; a < 4; a++)
#pragma unroll
                            for (int q = 0; q < 2; q++) v[a][q] = __builtin_fma(-li[c][a], lj[c][q], v[a][q]);
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int q = 0; q < 2; q++) if (i0 + 16 * a < n && j0 + 32 * q < n) A[(size_t)ia[a] * n + jq[q]] = v[a][q];
                }
            __syncthreads();
            GF_PQ(4);
        }
    }
#ifdef GF_PROFILE_STEP
    if (blockIdx.x == 0 && threadIdx.x == 0 && sb.stamps) for (int q = 0; q < 5; q++) sb.stamps[100 + q] = pq[q];
#endif
#undef GF_PQ
#undef GF_PIV_SYNC
    GF_MST(5);
    // ---- the right-hand side of the prior in the directions the factor does not span.  Forward substitution gives r = L1^-1 (P^T b)_1: J^T r then reproduces b in the
    // `rank` pivot rows and PREDICTS it in the other m2 = n - rank rows (L2 L1^-1 b_1).  The reference's r = S^-1/2 V^T b (marginalization_factor.cpp:294-302) makes
    // J^T r the ORTHOGONAL projection of b onto the kept eigenvectors instead.  Where the kept system has eigenvalues near the 1e-8 cut (GNSS windows: yaw_enu_local, the
    // ECEF anchor) the two differ by up to 5e-3 of 2e3, and the next windows' GNSS states carry that.  The least-squares r -- min |J^T r - b| -- is the orthogonal
    // projection onto the factor's own range: on the oracle's systems it sits 10-30 x closer to the reference's (scripts/marg_rhs_projection.py).  With K = L2 L1^-1
    // (m2 x rank, m2 ~ 4-15) and e = b_2 - L2 r_fs (the forward substitution has left it in zb[rank..n)):
    //     r = r_fs + L1^-1 T w,   T = L1^-T L2^T,   (I + T^T T) w = e
    // one backward substitution with m2 right-hand sides, an m2 x m2 solve, one forward substitution.  m2 = 0 (full rank) or m2 > 32: nothing is done.
    {
        constexpr int kM2 = 32, TS = kM2 + 1, kLsRows = 256;   // kept systems of up to 256 columns (s_dg above has the same bound)
        const int m2 = n - rank;
        if (ls_rhs && rank > 0 && m2 > 0 && m2 <= kM2 && n <= kLsRows) {
            // scratch in LDS in both variants: the chains below synchronise lanes of ONE wavefront with wavefront barriers, which orders LDS accesses but not a
            // store and a later load of another lane through the vector L1 (the global-memory variant keeps A, zb, dinvs, zr in sb.Mg; T, G and vv get their own LDS)
            __shared__ double s_ls[GS ? kLsRows * TS + kM2 * (kM2 + 1) + kLsRows : 1];
            double* Tm = GS ? s_ls : V + 2048;     // rank x TS
            double* G = Tm + (size_t)(GS ? kLsRows : n) * TS;       // m2 x (m2 + 1): [I + T^T T | e], then w in its last column
            // The two substitutions are chains of `rank` dependent steps.  Each runs out of registers: a lane owns the rows lane + 64 q of its vector, a step takes the
            // pivot entry with a readlane (the step index is uniform) and the entries of L1 it multiplies are read-only here, so the loads of step k - 1 are issued
            // before the arithmetic of step k -- ~60 cycles per step where the LDS-resident version (Tm / vv read, updated and written back per step, four LDS round
            // trips, ~500 cycles) made the chains 79 k of the kernel's cycles.  The m2 columns of T are independent: one wavefront each.  Same fma chain per entry.
            constexpr int QR = GS ? 4 : 2;   // rows per lane: n <= 256 / n <= 128
            auto rl = [](double v, int l) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l)); };
            GF_MST(8);
            {
                double dv[QR];
#pragma unroll
                for (int q = 0; q < QR; q++) { const int i = lane + 64 * q; dv[q] = i < rank ? dinvs[i] : 0.0; }
                for (int j = wave; j < m2; j += 8) {   // column-oriented backward substitution with L1^T: row k is final when step k starts
                    double t[QR], a[4][QR], nx[4][QR];
                    // L1[k][i] = A[i][k] (row i right of its diagonal); rows / columns past the ends are clamped for the load and not used
                    auto col = [&](double (&a)[QR], int kc) {
#pragma unroll
                        for (int q = 0; q < QR; q++) a[q] = A[(size_t)min(lane + 64 * q, rank - 1) * n + max(kc, 0)];
                    };
                    auto step = [&](int k, const double (&ac)[QR]) {
                        const int kq = k >> 6, kl = k & 63;
                        double ts = t[0], ds = dv[0];
#pragma unroll
                        for (int q = 1; q < QR; q++) if (kq == q) { ts = t[q]; ds = dv[q]; }
                        const double tk = rl(ts, kl) * rl(ds, kl);
#pragma unroll
                        for (int q = 0; q < QR; q++) { const int i = lane + 64 * q; if (i < k) t[q] = __builtin_fma(-ac[q], tk, t[q]); else if (i == k) t[q] = tk; }
                    };
#pragma unroll
                    for (int q = 0; q < QR; q++) { const int i = lane + 64 * q; t[q] = i < rank ? A[(size_t)i * n + rank + j] : 0.0; }
                    // four steps per trip: the loads of the next four columns are issued before the four dependent steps of this trip and land while they run
#pragma unroll
                    for (int u = 0; u < 4; u++) col(a[u], rank - 1 - u);
                    int k = rank - 1;
                    for (; k >= 3; k -= 4) {
#pragma unroll
                        for (int u = 0; u < 4; u++) col(nx[u], k - 4 - u);
#pragma unroll
                        for (int u = 0; u < 4; u++) step(k - u, a[u]);
#pragma unroll
                        for (int u = 0; u < 4; u++)
#pragma unroll
                            for (int q = 0; q < QR; q++) a[u][q] = nx[u][q];
                    }
#pragma unroll
                    for (int u = 0; u < 3; u++) if (k - u >= 0) step(k - u, a[u]);
#pragma unroll
                    for (int q = 0; q < QR; q++) { const int i = lane + 64 * q; if (i < rank) Tm[i * TS + j] = t[q]; }
                }
            }
            __syncthreads();
            GF_MST(9);
            for (int i = tid; i < m2 * (m2 + 1); i += 512) {
                const int a = i / (m2 + 1), c = i - a * (m2 + 1);
                double s;
                if (c == m2) s = zb[rank + a];
                else { s = a == c ? 1.0 : 0.0; for (int k = 0; k < rank; k++) s = __builtin_fma(Tm[k * TS + a], Tm[k * TS + c], s); }
                G[a * (kM2 + 1) + c] = s;
            }
            __syncthreads();
            GF_MST(10);
            if (wave == 0) {   // (I + T^T T) w = e: symmetric positive definite with eigenvalues >= 1, Gaussian elimination without pivoting, lane = row
                const int r = lane;
                for (int c = 0; c < m2; c++) {
                    const double f = (r > c && r < m2) ? G[r * (kM2 + 1) + c] / G[c * (kM2 + 1) + c] : 0.0;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                    if (r > c && r < m2) for (int q = c; q <= m2; q++) G[r * (kM2 + 1) + q] = __builtin_fma(-f, G[c * (kM2 + 1) + q], G[r * (kM2 + 1) + q]);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                }
                for (int c = m2 - 1; c >= 0; c--) {
                    if (r == c) G[c * (kM2 + 1) + m2] /= G[c * (kM2 + 1) + c];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                    if (r < c) G[r * (kM2 + 1) + m2] = __builtin_fma(-G[r * (kM2 + 1) + c], G[c * (kM2 + 1) + m2], G[r * (kM2 + 1) + m2]);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                }
            }
            __syncthreads();
            GF_MST(11);
            if (wave == 0) {   // vv = T w, then the forward substitution with L1 (column k of L1 below its diagonal = row k of A right of it); zr += L1^-1 T w
                double v[QR], dv[QR], a[4][QR], nx[4][QR], zadd[QR] = {};
#pragma unroll
                for (int q = 0; q < QR; q++) {
                    const int i = lane + 64 * q;
                    double s = 0.0;
                    if (i < rank) for (int j = 0; j < m2; j++) s = __builtin_fma(Tm[i * TS + j], G[j * (kM2 + 1) + m2], s);
                    v[q] = s; dv[q] = i < rank ? dinvs[i] : 0.0;
                }
                auto row = [&](double (&a)[QR], int kr) {
#pragma unroll
                    for (int q = 0; q < QR; q++) a[q] = A[(size_t)min(kr, rank - 1) * n + min(lane + 64 * q, n - 1)];
                };
                auto step = [&](int k, const double (&ac)[QR]) {
                    const int kq = k >> 6, kl = k & 63;
                    double vs = v[0], ds = dv[0];
#pragma unroll
                    for (int q = 1; q < QR; q++) if (kq == q) { vs = v[q]; ds = dv[q]; }
                    const double sk = rl(vs, kl) * rl(ds, kl);
#pragma unroll
                    for (int q = 0; q < QR; q++) { const int i = lane + 64 * q; if (i > k && i < rank) v[q] = __builtin_fma(-ac[q], sk, v[q]); else if (i == k) zadd[q] = sk; }
                };
#pragma unroll
                for (int u = 0; u < 4; u++) row(a[u], u);
                int k = 0;
                for (; k + 3 < rank; k += 4) {
#pragma unroll
                    for (int u = 0; u < 4; u++) row(nx[u], k + 4 + u);
#pragma unroll
                    for (int u = 0; u < 4; u++) step(k + u, a[u]);
#pragma unroll
                    for (int u = 0; u < 4; u++)
#pragma unroll
                        for (int q = 0; q < QR; q++) a[u][q] = nx[u][q];
                }
#pragma unroll
                for (int u = 0; u < 3; u++) if (k + u < rank) step(k + u, a[u]);
#pragma unroll
                for (int q = 0; q < QR; q++) { const int i = lane + 64 * q; if (i < rank) zr[i] += zadd[q]; }
            }
            __syncthreads();
        }
    }
    GF_MST(6);
#ifdef GF_PROFILE_STEP
    if (blockIdx.x == 0 && threadIdx.x == 0 && sb.stamps) sb.stamps[39] = rank * 1000 + n;
#endif
    double* J = out.J + (size_t)b * d.NPRI * d.NPRI;
    double* rr = out.r + (size_t)b * d.NPRI;
    for (int i = tid; i < n * n; i += 512) {
        const int k = i / n, pos = i % n;   // J[k][perm[pos]] = L[pos][k] for pos >= k, k < rank (column k of A below the diagonal is still unscaled)
        double v = 0.0;
        if (k < rank && pos >= k) v = pos == k ? 1.0 / dinvs[k] : A[k * n + pos];   // row k right of the diagonal = column k of L
        J[(size_t)k * n + perm[pos]] = v;
    }
    for (int k = tid; k < n; k += 512) rr[k] = k < rank ? zr[k] : 0.0;
#ifdef GF_PROFILE_STEP
    __syncthreads();
    if (threadIdx.x == 0 && sb.stamps) {   // entry -> first stamp, last stamp -> exit of block 0; the longest block of the launch
        const long long t_exit = clock64();
        if (blockIdx.x == 0) { sb.stamps[106] = sb.stamps[32] - t_entry; sb.stamps[107] = t_exit - sb.stamps[38]; sb.stamps[108] = t_exit - t_entry; }
        atomicMax(reinterpret_cast<unsigned long long*>(sb.stamps + 109), (unsigned long long)(t_exit - t_entry));
    }
#endif
}

}  // namespace gfb
