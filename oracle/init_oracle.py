"""TEST INFRASTRUCTURE -- CPU restatement (numpy) of Ground-Fusion's initialisation while moving: what
Estimator::initialStructure (estimator.cpp:1684-1847) calls once the stationary / wheel-activated shortcuts have not fired --
MotionEstimator::solveRelativeRT_PNP (initial/solve_5pts.cpp:244-277), GlobalSFM::constructWithDepth (initial/initial_sfm.cpp:379-594),
cv::solvePnP per frame, VisualIMUAlignment (initial/initial_aligment.cpp:427-653).  Only tests/ may import this file.

PARITY UNPINNED.  Three pieces of this path live in third-party code that the reference does not vendor:
  * cv::solvePnPRansac / cv::solvePnP (OpenCV 4.2 calib3d: solvepnp.cpp, ptsetreg.cpp RANSACPointSetRegistrator, epnp.cpp, calibration.cpp
    cvFindExtrinsicCameraParams2, compat_ptsetreg.cpp CvLevMarq) -- restated from the published algorithms: the RANSAC loop with cv::RNG's
    multiply-with-carry generator and its subset / iteration-count rules, EPnP (Lepetit, Moreno-Noguer, Fua 2009) as the 5-point kernel, the DLT
    start and the Levenberg-Marquardt refinement with OpenCV's lambda schedule (10^-3, x10 on a worse step, /10 on a better one, 20 iterations,
    relative parameter change < FLT_EPSILON);
  * ceres::Solve with DENSE_SCHUR and the default LEVENBERG_MARQUARDT strategy (Ceres 1.14 trust_region_minimizer.cc,
    levenberg_marquardt_strategy.cc) for the structure-from-motion bundle adjustment; the wall-clock cap (0.2 s) is not restated.
Where the outcome of the third-party code is decided by rounding (the basis OpenCV's SVD returns inside the null space of EPnP's 12 x 12 system
with 5 points, the sign of eigenvectors), a canonical choice is made here and in the product so that the two can be compared; for planar point sets
the homography start is the normalised DLT without OpenCV's refinement of H (it only starts the pose refinement).  The reference's own arithmetic -- including that the `scale' read from the last entry of
the alignment vector is a gravity-refinement component in the depth variants (initial_aligment.cpp:427-497, estimator.cpp:1871) -- is followed
literally."""
import math

import numpy as np


_DEBUG = False


def f32(a):
    """cv::Point2f / cv::Point3f storage: values pass through float"""
    return np.asarray(a, np.float32).astype(np.float64)


# ---------------------------------------------------------------- small linear algebra with canonical choices
def canon_sign(v):
    """eigenvector sign: the entry of largest magnitude is positive"""
    k = int(np.argmax(np.abs(v)))
    return -v if v[k] < 0 else v


def sym_eig(A):
    """ascending eigenvalues, eigenvectors as columns with the canonical sign"""
    w, V = np.linalg.eigh((A + A.T) * 0.5)
    return w, np.stack([canon_sign(V[:, i]) for i in range(len(w))], axis=1)


def canonical_subspace_basis(B):
    """orthonormal basis of span(B) that depends on the subspace only: Gram-Schmidt of the projections of e_0, e_1, ... (those that keep more than
    0.1 of their length after the earlier ones are removed)"""
    n, k = B.shape
    out = []
    for j in range(n):
        v = B @ B[j, :]                       # projection of e_j
        for u in out:
            v = v - u * (u @ v)
        nv = math.sqrt(float(v @ v))
        if nv > 0.1:
            out.append(v / nv)
            if len(out) == k:
                break
    assert len(out) == k
    return np.stack(out, axis=1)


def polar_rotation(M):
    """U V^T of the SVD M = U S V^T (the orthogonal polar factor), through the eigen-decomposition of M^T M"""
    w, V = sym_eig(M.T @ M)
    return M @ (V @ np.diag(1.0 / np.sqrt(w)) @ V.T)


# ---------------------------------------------------------------- cv::Rodrigues
def rodrigues(r):
    """rotation vector -> matrix"""
    th = math.sqrt(float(r @ r))
    if th < 2.220446049250313e-16:  # DBL_EPSILON
        return np.eye(3)
    k = r / th
    c, s = math.cos(th), math.sin(th)
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return c * np.eye(3) + (1 - c) * np.outer(k, k) + s * K


def rodrigues_inv(R):
    """rotation matrix -> vector (calibration.cpp cvRodrigues2, matrix branch; the input is a rotation, so its re-orthogonalisation is skipped)"""
    r = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = math.sqrt(float(r @ r) * 0.25)
    c = min(1.0, max(-1.0, (R[0, 0] + R[1, 1] + R[2, 2] - 1) * 0.5))
    th = math.acos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        t = np.array([math.sqrt(max((R[i, i] + 1) * 0.5, 0.0)) for i in range(3)])
        if R[0, 1] < 0:
            t[1] = -t[1]
        if R[0, 2] < 0:
            t[2] = -t[2]
        if abs(t[0]) < abs(t[1]) and abs(t[0]) < abs(t[2]) and (R[1, 2] > 0) != (t[1] * t[2] > 0):
            t[2] = -t[2]
        return t * (th / math.sqrt(float(t @ t)))
    return r * (th / (2 * s))


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def d_rodrigues(r):
    """dR/dr_k, k = 0..2 (exact derivative of the exponential map; Gallego & Yezzi's compact form)"""
    th2 = float(r @ r)
    if th2 < 1e-20:
        return [skew(e) for e in np.eye(3)]
    R = rodrigues(r)
    return [(r[k] * skew(r) + skew(np.cross(r, (np.eye(3) - R)[:, k]))) @ R / th2 for k in range(3)]


# ---------------------------------------------------------------- cv::RNG
class CvRNG:
    """core/operations.hpp RNG: multiply-with-carry, coefficient 4164903690"""

    def __init__(self, state=0xFFFFFFFFFFFFFFFF):
        self.state = state

    def next(self):
        self.state = ((self.state & 0xFFFFFFFF) * 4164903690 + (self.state >> 32)) & 0xFFFFFFFFFFFFFFFF
        return self.state & 0xFFFFFFFF

    def uniform(self, a, b):
        return a if a == b else a + self.next() % (b - a)


# ---------------------------------------------------------------- pose from a guess: OpenCV's Levenberg-Marquardt (CvLevMarq, cvFindExtrinsicCameraParams2)
def project(rvec, tvec, X, jac=False):
    R = rodrigues(rvec)
    P = X @ R.T + tvec
    iz = 1.0 / P[:, 2]
    uv = P[:, :2] * iz[:, None]
    if not jac:
        return uv
    n = len(X)
    J = np.zeros((2 * n, 6))
    dR = d_rodrigues(rvec)
    du = np.stack([iz, np.zeros(n), -P[:, 0] * iz * iz], axis=1)   # d u / d P
    dv = np.stack([np.zeros(n), iz, -P[:, 1] * iz * iz], axis=1)
    for k in range(3):
        dP = X @ dR[k].T
        J[0::2, k] = np.sum(du * dP, axis=1)
        J[1::2, k] = np.sum(dv * dP, axis=1)
    J[0::2, 3:6] = du
    J[1::2, 3:6] = dv
    return uv, J


def pnp_refine(X, uv, rvec, tvec, max_iter=20, eps=1.1920928955078125e-07):
    """CvLevMarq::update / step driven the way cvFindExtrinsicCameraParams2 drives it (calibration.cpp: err = projection - measurement,
    param = prevParam - (J^T J with its diagonal times 1 + lambda)^-1 J^T err)"""
    param = np.concatenate([rvec, tvec]).astype(float)
    lam_lg10, iters = -3, 0
    prev_err = 0.0
    while True:
        # CALC_J
        p, J = project(param[:3], param[3:], X, jac=True)
        err = (p - uv).reshape(-1)
        JtJ, JtE = J.T @ J, J.T @ err
        prev = param.copy()

        def step():
            A = JtJ.copy()
            A[np.diag_indices(6)] *= 1.0 + math.exp(lam_lg10 * math.log(10.0))
            return prev - np.linalg.solve(A, JtE)

        param = step()
        if iters == 0:
            prev_err = math.sqrt(float(err @ err))
        # CHECK_ERR
        while True:
            e = (project(param[:3], param[3:], X) - uv).reshape(-1)
            en = math.sqrt(float(e @ e))
            if en > prev_err:
                lam_lg10 += 1
                if lam_lg10 <= 16:
                    param = step()
                    continue
            break
        lam_lg10 = max(lam_lg10 - 1, -16)
        iters += 1
        dn, pn = math.sqrt(float((param - prev) @ (param - prev))), math.sqrt(float(prev @ prev))
        if iters >= max_iter or dn < eps * pn:     # cvNorm(param, prevParam, CV_RELATIVE_L2) < epsilon
            return param[:3].copy(), param[3:].copy()
        prev_err = en


def homography_dlt(src, dst):
    """cv::findHomography(src, dst, 0): the normalised DLT of HomographyEstimatorCallback::runKernel (fundam.cpp: centroids, mean absolute deviation
    scaling, smallest eigenvector of the 9 x 9 normal matrix, H[2][2] = 1).  Its 10 Levenberg-Marquardt passes over H are not restated: H only
    starts the pose refinement that follows."""
    n = len(src)
    cM, cm = src.mean(axis=0), dst.mean(axis=0)
    sM, sm = np.abs(src - cM).sum(axis=0), np.abs(dst - cm).sum(axis=0)
    if min(sM.min(), sm.min()) < 2.220446049250313e-16:
        return None
    sM, sm = n / sM, n / sm
    A = np.zeros((2 * n, 9))
    x, y = (dst[:, 0] - cm[0]) * sm[0], (dst[:, 1] - cm[1]) * sm[1]
    Xn, Yn = (src[:, 0] - cM[0]) * sM[0], (src[:, 1] - cM[1]) * sM[1]
    A[0::2] = np.stack([Xn, Yn, np.ones(n), np.zeros(n), np.zeros(n), np.zeros(n), -x * Xn, -x * Yn, -x], axis=1)
    A[1::2] = np.stack([np.zeros(n), np.zeros(n), np.zeros(n), Xn, Yn, np.ones(n), -y * Xn, -y * Yn, -y], axis=1)
    _, V = sym_eig(A.T @ A)
    H0 = V[:, 0].reshape(3, 3)
    inv_norm = np.array([[1.0 / sm[0], 0, cm[0]], [0, 1.0 / sm[1], cm[1]], [0, 0, 1]])
    norm2 = np.array([[sM[0], 0, -cM[0] * sM[0]], [0, sM[1], -cM[1] * sM[1]], [0, 0, 1]])
    H = inv_norm @ H0 @ norm2
    return H / H[2, 2]


def pnp_planar(X, uv, w, V):
    """cvFindExtrinsicCameraParams2 without a guess, planar branch (calibration.cpp): the points in the coordinates of their plane, a homography to the
    image, its first two columns made a rotation.  w, V: eigen-decomposition (ascending) of the scatter of X."""
    Rt = np.stack([V[:, 2], V[:, 1], V[:, 0]], axis=0)      # rows: principal axes, largest spread first (V^T of cvSVD)
    if Rt[0, 2] ** 2 + Rt[1, 2] ** 2 < 1e-10:
        Rt = np.eye(3)
    if np.linalg.det(Rt) < 0:
        Rt = -Rt
    tt = -Rt @ X.mean(axis=0)
    Mxy = (X @ Rt.T + tt)[:, :2]
    H = homography_dlt(Mxy, uv)
    if H is None or not np.all(np.isfinite(H)):
        return np.zeros(3), np.zeros(3)
    n1, n2 = math.sqrt(float(H[:, 0] @ H[:, 0])), math.sqrt(float(H[:, 1] @ H[:, 1]))
    h1, h2 = H[:, 0] / max(n1, 2.220446049250313e-16), H[:, 1] / max(n2, 2.220446049250313e-16)
    t = H[:, 2] * (2.0 / max(n1 + n2, 2.220446049250313e-16))
    Rh = polar_rotation(np.stack([h1, h2, np.cross(h1, h2)], axis=1))     # cvRodrigues2 there and back: the nearest rotation
    return rodrigues_inv(Rh @ Rt), Rh @ tt + t


def pnp_dlt(X, uv):
    """cvFindExtrinsicCameraParams2 without a guess (calibration.cpp).  Non-planar: 2N x 12 system, smallest right singular vector, rotation
    made orthogonal, translation rescaled.  Planar (third singular value of the scatter below 1e-3 of the second): pnp_planar."""
    Xc = X - X.mean(axis=0)
    w, Vs = sym_eig(Xc.T @ Xc)                 # ascending
    if w[0] / w[1] < 1e-3:
        return pnp_planar(X, uv, w, Vs)
    n = len(X)
    L = np.zeros((2 * n, 12))
    L[0::2, 0:3], L[0::2, 3] = X, 1.0
    L[1::2, 4:7], L[1::2, 7] = X, 1.0
    L[0::2, 8:11], L[0::2, 11] = -uv[:, 0:1] * X, -uv[:, 0]
    L[1::2, 8:11], L[1::2, 11] = -uv[:, 1:2] * X, -uv[:, 1]
    _, V = sym_eig(L.T @ L)
    RRt = V[:, 0].reshape(3, 4)
    if np.linalg.det(RRt[:, :3]) < 0:
        RRt = -RRt
    RR, tt = RRt[:, :3], RRt[:, 3]
    sc = math.sqrt(float(np.sum(RR * RR)))
    R = polar_rotation(RR)
    return rodrigues_inv(R), tt * (math.sqrt(float(np.sum(R * R))) / sc)


def solve_pnp_iterative(X, uv, guess=None):
    """cv::solvePnP(..., useExtrinsicGuess, SOLVEPNP_ITERATIVE) with an identity camera matrix: (rvec, tvec) or None"""
    X, uv = f32(X), f32(uv)
    if guess is None:
        g = pnp_dlt(X, uv)
        if g is None:
            return None
        guess = g
    return pnp_refine(X, uv, np.asarray(guess[0], float), np.asarray(guess[1], float))


# ---------------------------------------------------------------- EPnP (epnp.cpp), the minimal solver of the RANSAC
_PAIRS = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]


def epnp(X, uv):
    """pose of n >= 4 points by EPnP as OpenCV runs it: 4 control points from the PCA of the points, the 4 smallest eigenvectors of M^T M,
    three closed-form guesses of the betas each polished by 5 Gauss-Newton steps, the one with the smallest reprojection error wins"""
    n = len(X)
    c0 = X.mean(axis=0)
    w, V = sym_eig((X - c0).T @ (X - c0))
    cws = [c0] + [c0 + math.sqrt(max(w[2 - i], 0.0) / n) * V[:, 2 - i] for i in range(3)]   # descending singular values, as cvSVD returns them
    CC = np.stack([cws[i] - cws[0] for i in (1, 2, 3)], axis=1)
    try:
        a123 = np.linalg.solve(CC, (X - cws[0]).T).T
    except np.linalg.LinAlgError:
        return None
    alphas = np.concatenate([1.0 - a123.sum(axis=1, keepdims=True), a123], axis=1)
    M = np.zeros((2 * n, 12))
    for j in range(4):
        M[0::2, 3 * j] = alphas[:, j]
        M[0::2, 3 * j + 2] = alphas[:, j] * (0.0 - uv[:, 0])
        M[1::2, 3 * j + 1] = alphas[:, j]
        M[1::2, 3 * j + 2] = alphas[:, j] * (0.0 - uv[:, 1])
    w12, V12 = sym_eig(M.T @ M)
    k = max(1, int(np.sum(w12 < 1e-9 * w12[-1])))
    vs = [V12[:, i] for i in range(4)]
    if k >= 2:                                                  # the basis inside a null space is rounding's choice: take the canonical one
        k = min(k, 4)
        Bc = canonical_subspace_basis(V12[:, :k])
        for i in range(k):
            vs[i] = Bc[:, i]
    dv = np.zeros((4, 6, 3))
    for i in range(4):
        for p, (a, b) in enumerate(_PAIRS):
            dv[i, p] = vs[i][3 * a:3 * a + 3] - vs[i][3 * b:3 * b + 3]
    L = np.zeros((6, 10))
    for p in range(6):
        d = dv[:, p]
        L[p] = [d[0] @ d[0], 2 * d[0] @ d[1], d[1] @ d[1], 2 * d[0] @ d[2], 2 * d[1] @ d[2], d[2] @ d[2], 2 * d[0] @ d[3], 2 * d[1] @ d[3], 2 * d[2] @ d[3],
                d[3] @ d[3]]
    rho = np.array([float((cws[a] - cws[b]) @ (cws[a] - cws[b])) for a, b in _PAIRS])

    def lsq(A, b):           # epnp.cpp qr_solve: Householder QR, no pivoting
        Q, R = np.linalg.qr(A)
        if np.any(np.abs(np.diag(R)) < 1e-300):
            return np.full(A.shape[1], np.nan)
        return np.linalg.solve(R, Q.T @ b)

    def approx(which):
        be = np.zeros(4)
        if which == 1:
            b4 = lsq(L[:, [0, 1, 3, 6]], rho)
            if b4[0] < 0:
                be[0] = math.sqrt(-b4[0]); be[1:] = -b4[1:] / be[0]
            else:
                be[0] = math.sqrt(b4[0]); be[1:] = b4[1:] / be[0]
        else:
            bb = lsq(L[:, [0, 1, 2]] if which == 2 else L[:, [0, 1, 2, 3, 4]], rho)
            if bb[0] < 0:
                be[0] = math.sqrt(-bb[0]); be[1] = math.sqrt(-bb[2]) if bb[2] < 0 else 0.0
            else:
                be[0] = math.sqrt(bb[0]); be[1] = math.sqrt(bb[2]) if bb[2] > 0 else 0.0
            if bb[1] < 0:
                be[0] = -be[0]
            if which == 3:
                be[2] = bb[3] / be[0]
        return be

    def gauss_newton(be):
        be = be.copy()
        for _ in range(5):
            A = np.stack([2 * L[:, 0] * be[0] + L[:, 1] * be[1] + L[:, 3] * be[2] + L[:, 6] * be[3],
                          L[:, 1] * be[0] + 2 * L[:, 2] * be[1] + L[:, 4] * be[2] + L[:, 7] * be[3],
                          L[:, 3] * be[0] + L[:, 4] * be[1] + 2 * L[:, 5] * be[2] + L[:, 8] * be[3],
                          L[:, 6] * be[0] + L[:, 7] * be[1] + L[:, 8] * be[2] + 2 * L[:, 9] * be[3]], axis=1)
            b = rho - (L[:, 0] * be[0] * be[0] + L[:, 1] * be[0] * be[1] + L[:, 2] * be[1] * be[1] + L[:, 3] * be[0] * be[2] + L[:, 4] * be[1] * be[2]
                       + L[:, 5] * be[2] * be[2] + L[:, 6] * be[0] * be[3] + L[:, 7] * be[1] * be[3] + L[:, 8] * be[2] * be[3] + L[:, 9] * be[3] * be[3])
            be = be + lsq(A, b)
        return be

    def pose(be):
        ccs = np.zeros((4, 3))
        for i in range(4):
            for j in range(4):
                ccs[j] += be[i] * vs[i][3 * j:3 * j + 3]
        pcs = alphas @ ccs
        if pcs[0, 2] < 0:
            ccs, pcs = -ccs, -pcs
        pc0, pw0 = pcs.mean(axis=0), X.mean(axis=0)
        ABt = (pcs - pc0).T @ (X - pw0)
        if not np.all(np.isfinite(ABt)):
            return None
        U, _, Vt = np.linalg.svd(ABt)
        R = U @ Vt
        if np.linalg.det(R) < 0:
            R[2] = -R[2]
        t = pc0 - R @ pw0
        P = X @ R.T + t
        err = float(np.mean(np.sqrt((P[:, 0] / P[:, 2] - uv[:, 0]) ** 2 + (P[:, 1] / P[:, 2] - uv[:, 1]) ** 2)))
        return err, R, t

    best = None
    for which in (1, 2, 3):
        with np.errstate(all="ignore"):
            r = pose(gauss_newton(approx(which)))
        if r is not None and np.isfinite(r[0]) and (best is None or r[0] < best[0]):
            best = r
    return None if best is None else (rodrigues_inv(best[1]), best[2])


# ---------------------------------------------------------------- cv::solvePnPRansac
def ransac_update_num_iters(p, ep, model_points, max_iters):
    p, ep = min(max(p, 0.0), 1.0), min(max(ep, 0.0), 1.0)
    num, denom = max(1.0 - p, 2.2250738585072014e-308), 1.0 - (1.0 - ep) ** model_points
    if denom < 2.2250738585072014e-308:
        return 0
    num, denom = math.log(num), math.log(denom)
    return max_iters if denom >= 0 or -num >= max_iters * (-denom) else int(np.rint(num / denom))


def solve_pnp_ransac(X, uv, iterations=100, reproj=1.0 / 460, confidence=0.99):
    """cv::solvePnPRansac(obj, img, I, noArray, rvec, tvec, false, 100, 1/460, 0.99, inliers, SOLVEPNP_ITERATIVE) (solvepnp.cpp: 5-point EPnP kernel
    inside RANSACPointSetRegistrator::run, then solvePnP ITERATIVE without a guess on the inliers).  Returns (rvec, tvec, inlier indices) or None."""
    X, uv = f32(X), f32(uv)
    n, mp = len(X), 5
    if n < mp:
        return None
    thr = np.float32(reproj * reproj)
    rng = CvRNG()
    niters, best_count, best_mask = max(iterations, 1), 0, None
    it = 0
    while it < niters:
        it += 1
        if n > mp:
            idx = []
            while len(idx) < mp:                     # getSubset: redraw on a repeated index (no further subset check for PnP)
                i = rng.uniform(0, n)
                if i not in idx:
                    idx.append(i)
        else:
            idx = list(range(n))
        m = epnp(X[idx], uv[idx])
        if m is None:
            continue
        with np.errstate(all="ignore"):
            pr = project(m[0], m[1], X).astype(np.float32)
            d = uv.astype(np.float32) - pr
            err = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]
        mask = err <= thr
        good = int(mask.sum())
        if _DEBUG:
            print("ransac it %d subset %s good %d rv %.12g %.12g %.12g tv %.12g %.12g %.12g" % (it - 1, " ".join(str(i) for i in idx), good, *m[0], *m[1]))
        if good > max(best_count, mp - 1):
            best_count, best_mask = good, mask
            niters = ransac_update_num_iters(confidence, (n - good) / n, mp, niters)
    if best_mask is None:
        return None
    inl = np.nonzero(best_mask)[0]
    r = solve_pnp_iterative(X[inl], uv[inl])
    return None if r is None else (r[0], r[1], inl)


# ---------------------------------------------------------------- the structure-from-motion bundle adjustment (ceres::Solve, LM, DENSE_SCHUR)
def quat_rot(q):
    w, x, y, z = q / math.sqrt(float(q @ q))
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def quat_plus(q, d):
    """ceres::QuaternionParameterization::Plus: [cos|d|, sin|d| d/|d|] * q"""
    nd = math.sqrt(float(d @ d))
    if nd == 0.0:
        return q.copy()
    s = math.sin(nd) / nd
    a = np.array([math.cos(nd), s * d[0], s * d[1], s * d[2]])
    return np.array([a[0] * q[0] - a[1] * q[1] - a[2] * q[2] - a[3] * q[3], a[0] * q[1] + a[1] * q[0] + a[2] * q[3] - a[3] * q[2],
                     a[0] * q[2] - a[1] * q[3] + a[2] * q[0] + a[3] * q[1], a[0] * q[3] + a[1] * q[2] - a[2] * q[1] + a[3] * q[0]])


def sfm_bundle_adjust(qs, ts, pts, obs, const_rot, const_trans, max_iters=50):
    """ceres::Solve of initial_sfm.cpp:497-560: residuals ReprojectionError3D (initial_sfm.h:36-63) over camera rotation (quaternion, local size 3),
    camera translation and point position; rotations in `const_rot` and translations in `const_trans` are held.  obs: (frame, point index, u, v).
    Trust-region loop of trust_region_minimizer.cc with the Levenberg-Marquardt strategy (radius 1e4, diagonal clamped to [1e-6, 1e32], radius /
    max(1/3, 1 - (2 rho - 1)^3) on success, / 2, 4, 8 ... on failure), Jacobi scaling, tolerances 1e-6 / 1e-10 / 1e-8.
    Returns (qs, ts, pts, converged, final_cost)."""
    qs, ts, pts = [q.copy() for q in qs], [t.copy() for t in ts], [p.copy() for p in pts]
    nf, npnt = len(qs), len(pts)
    col, ncol = {}, 0
    for i in range(nf):
        if i not in const_rot:
            col[("r", i)] = ncol; ncol += 3
        if i not in const_trans:
            col[("t", i)] = ncol; ncol += 3
    for j in range(npnt):
        col[("p", j)] = ncol; ncol += 3
    nres = 2 * len(obs)

    def evaluate(qs, ts, pts, jac):
        r = np.zeros(nres)
        J = np.zeros((nres, ncol)) if jac else None
        for k, (i, j, u, v) in enumerate(obs):
            R = quat_rot(qs[i])
            Rx = R @ pts[j]
            p = Rx + ts[i]
            iz = 1.0 / p[2]
            r[2 * k], r[2 * k + 1] = p[0] * iz - u, p[1] * iz - v
            if jac:
                D = np.array([[iz, 0, -p[0] * iz * iz], [0, iz, -p[1] * iz * iz]])
                if ("r", i) in col:
                    J[2 * k:2 * k + 2, col[("r", i)]:col[("r", i)] + 3] = D @ (-2.0 * skew(Rx))
                if ("t", i) in col:
                    J[2 * k:2 * k + 2, col[("t", i)]:col[("t", i)] + 3] = D
                J[2 * k:2 * k + 2, col[("p", j)]:col[("p", j)] + 3] = D @ R
        return 0.5 * float(r @ r), r, J

    def x_vec(qs, ts, pts):   # the reduced program: constant blocks are not part of the state vector
        parts = [qs[i] for i in range(nf) if i not in const_rot] + [ts[i] for i in range(nf) if i not in const_trans] + list(pts)
        return np.concatenate(parts) if parts else np.zeros(0)

    cost, res, J = evaluate(qs, ts, pts, True)
    if ncol == 0 or nres == 0:
        return qs, ts, pts, True, cost
    scale = 1.0 / (1.0 + np.sqrt(np.sum(J * J, axis=0)))
    J = J * scale
    radius, decrease, reuse = 1e4, 2.0, False
    diag = None
    last_ok, invalid_run, converged = True, 0, False
    it = 0
    while True:
        if it >= max_iters:
            break
        g = J.T @ res
        if last_ok and np.max(np.abs(g / scale)) <= 1e-10:
            converged = True
            break
        if radius <= 1e-32:
            break
        it += 1
        if not reuse:
            diag = np.clip(np.sum(J * J, axis=0), 1e-6, 1e32)
        H = J.T @ J + np.diag(diag / radius)
        try:
            step = -np.linalg.solve(H, g)
            valid = bool(np.all(np.isfinite(step)))
        except np.linalg.LinAlgError:
            valid = False
        mcc = 0.0
        if valid:
            mr = J @ step
            mcc = -float(mr @ (res + mr / 2.0))
            valid = mcc > 0.0
        if not valid:
            last_ok = False
            invalid_run += 1
            if invalid_run >= 5:
                break
            radius /= decrease; decrease *= 2.0; reuse = True
            continue
        invalid_run = 0
        d = step * scale
        cq = [quat_plus(qs[i], d[col[("r", i)]:col[("r", i)] + 3]) if ("r", i) in col else qs[i].copy() for i in range(nf)]
        ct = [ts[i] + d[col[("t", i)]:col[("t", i)] + 3] if ("t", i) in col else ts[i].copy() for i in range(nf)]
        cp = [pts[j] + d[col[("p", j)]:col[("p", j)] + 3] for j in range(npnt)]
        ccost, _, _ = evaluate(cq, ct, cp, False)
        xv, cv = x_vec(qs, ts, pts), x_vec(cq, ct, cp)
        if math.sqrt(float((xv - cv) @ (xv - cv))) <= 1e-8 * (math.sqrt(float(xv @ xv)) + 1e-8):
            converged = True
            break
        if abs(cost - ccost) <= 1e-6 * cost:
            converged = True
            break
        rel = (cost - ccost) / mcc
        if rel > 1e-3:
            qs, ts, pts = cq, ct, cp
            cost, res, J = evaluate(qs, ts, pts, True)
            J = J * scale
            radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2.0 * rel - 1.0) ** 3)); decrease = 2.0; reuse = False
            last_ok = True
        else:
            radius /= decrease; decrease *= 2.0; reuse = True
            last_ok = False
    return qs, ts, pts, converged, cost


# ---------------------------------------------------------------- GlobalSFM::constructWithDepth
class SFMFeature:
    def __init__(self, fid):
        self.state, self.id, self.observation, self.observation_depth, self.position = False, fid, [], [], np.zeros(3)


def _q_from_R(R):
    from estimator_oracle import R_to_quat
    return np.asarray(R_to_quat(R), float)


def construct_with_depth(frame_num, l, relative_R, relative_T, sfm_f):
    """initial_sfm.cpp:379-594.  Returns (q[], T[], {feature id: position}) in the frame of camera l, or None."""
    from estimator_oracle import quat_to_R
    cR, cT = [None] * frame_num, [None] * frame_num
    last = frame_num - 1
    cR[l], cT[l] = np.eye(3), np.zeros(3)
    q_last = _q_from_R(relative_R)                                   # q[l] * Quaterniond(relative_R), q[l] = identity
    q_inv = np.array([q_last[0], -q_last[1], -q_last[2], -q_last[3]]) / float(q_last @ q_last)
    cR[last] = quat_to_R(*q_inv)
    cT[last] = -1 * (cR[last] @ relative_T)
    cQ = {l: np.array([1.0, 0, 0, 0]), last: q_inv}

    def pnp(i, R0, P0):   # solveFrameByPnP :33-84
        X, uv = [], []
        for f in sfm_f:
            if not f.state:
                continue
            for (fr, p) in f.observation:
                if fr == i:
                    uv.append(p); X.append(f.position.copy())
                    break
        if len(uv) < 10:
            return None
        r = solve_pnp_iterative(np.array(X), np.array(uv), (rodrigues_inv(R0), P0))
        return None if r is None else (rodrigues(r[0]), r[1])

    def tri(f0, f1):      # triangulateTwoFramesWithDepth :133-182
        R0, t0, R1, t1 = cR[f0], cT[f0], cR[f1], cT[f1]
        for f in sfm_f:
            if f.state:
                continue
            p0 = p1 = None
            for (fr, p), (_, d) in zip(f.observation, f.observation_depth):
                if d < 0.1 or d > 10:
                    continue
                if fr == f0:
                    p0 = np.array([p[0] * d, p[1] * d, d])
                if fr == f1:
                    p1 = p
            if p0 is not None and p1 is not None:
                X = R0.T @ p0 - R0.T @ t0
                pr = R1 @ X + t1
                res = p1 - np.array([pr[0] / pr[2], pr[1] / pr[2]])
                if math.sqrt(float(res @ res)) < 1.0 / 460:
                    f.state, f.position = True, X

    for i in range(l, last):
        if i > l:
            r = pnp(i, cR[i - 1], cT[i - 1])
            if r is None:
                return None
            cR[i], cT[i] = r
            cQ[i] = _q_from_R(cR[i])
        tri(i, last)
    for i in range(l + 1, last):
        tri(l, i)
    for i in range(l - 1, -1, -1):
        r = pnp(i, cR[i + 1], cT[i + 1])
        if r is None:
            return None
        cR[i], cT[i] = r
        cQ[i] = _q_from_R(cR[i])
        tri(i, l)
    for f in sfm_f:       # :461-497
        if f.state or len(f.observation) < 2:
            continue
        d = f.observation_depth[0][1]
        if d < 0.1 or d > 10:
            continue
        f0, p = f.observation[0]
        f1, p1 = f.observation[-1]
        p0 = np.array([p[0] * d, p[1] * d, d])
        X = cR[f0].T @ p0 - cR[f0].T @ cT[f0]
        pr = cR[f1] @ X + cT[f1]
        res = p1 - np.array([pr[0] / pr[2], pr[1] / pr[2]])
        if math.sqrt(float(res @ res)) < 1.0 / 460:
            f.state, f.position = True, X
    live = [f for f in sfm_f if f.state]
    index = {id(f): k for k, f in enumerate(live)}
    obs = [(fr, index[id(f)], p[0], p[1]) for f in live for (fr, p) in f.observation]
    qs, ts, pts, conv, cost = sfm_bundle_adjust([cQ[i] for i in range(frame_num)], [cT[i] for i in range(frame_num)], [f.position for f in live], obs,
                                                const_rot={l}, const_trans={l, last})
    if not (conv or cost < 5e-3):
        return None
    for f, p in zip(live, pts):
        f.position = p
    q_out, T_out = [], []
    for i in range(frame_num):
        qi = np.array([qs[i][0], -qs[i][1], -qs[i][2], -qs[i][3]]) / float(qs[i] @ qs[i])
        q_out.append(qi)
        T_out.append(-1 * (quat_to_R(*qi) @ ts[i]))          # Eigen's q * v: rotation by the (here unit up to rounding) quaternion
    return q_out, T_out, {f.id: f.position.copy() for f in sfm_f if f.state}


# ---------------------------------------------------------------- VisualIMUAlignment: linear alignment + gravity refinement
def tangent_basis(g0):  # initial_aligment.cpp:49-62
    a = g0 / math.sqrt(float(g0 @ g0))
    tmp = np.array([0.0, 0, 1])
    if np.array_equal(a, tmp):
        tmp = np.array([1.0, 0, 0])
    b = tmp - a * float(a @ tmp)
    b = b / math.sqrt(float(b @ b))
    return np.stack([b, np.cross(a, b)], axis=1)


def linear_alignment(frames, TIC, G_norm, use_wheel, RIO=None, TIO=None):
    """LinearAlignmentWithWD / LinearAlignmentWithDepth followed by RefineGravityWithWD / RefineGravityWithDepth (initial_aligment.cpp:427-638).
    frames: objects with R, T, pre_integration.r() (delta_p, delta_v, sum_dt) and pre_integration_wheel.r() (delta_p), in time order.
    Literal: the wheel row's `scale' column lies on the last gravity column (8 of 9, then 7 of 8); the normal matrix of the refinement is not cleared
    between its four passes and is multiplied by 1000 in each.  Returns (g, x) or None."""
    n = len(frames)
    ns = n * 3 + 3
    rows = 9 if use_wheel else 6

    def blocks(k_g, lxly, g0):
        A, b = np.zeros((n * 3 + k_g, n * 3 + k_g)), np.zeros(n * 3 + k_g)
        for i in range(n - 1):
            fi, fj = frames[i], frames[i + 1]
            pj = fj.pre_integration.r()
            dt = pj["sum_dt"]
            tA, tb = np.zeros((rows, 6 + k_g)), np.zeros(rows)
            RiT = fi.R.T
            tA[0:3, 0:3] = -dt * np.eye(3)
            tA[3:6, 0:3] = -np.eye(3)
            tA[3:6, 3:6] = RiT @ fj.R
            if k_g == 3:
                tA[0:3, 6:9] = RiT * (dt * dt / 2)
                tA[3:6, 6:9] = RiT * dt
                tb[0:3] = pj["delta_p"] + RiT @ fj.R @ TIC - TIC
                tb[3:6] = pj["delta_v"]
            else:
                tA[0:3, 6:8] = (RiT * (dt * dt / 2)) @ lxly
                tA[3:6, 6:8] = (RiT * dt) @ lxly
                tb[0:3] = pj["delta_p"] + RiT @ fj.R @ TIC - TIC - (RiT * (dt * dt / 2)) @ g0
                tb[3:6] = pj["delta_v"] - (RiT * dt) @ g0
            if use_wheel:
                tA[6:9, 6 + k_g - 1] = (fi.R @ RIO).T @ (fj.T - fi.T) / 100
                tb[6:9] = fj.pre_integration_wheel.r()["delta_p"] - RIO.T @ RiT @ fj.R @ TIO + (fi.R @ RIO).T @ fj.R @ TIC - RIO.T @ (TIC - TIO)
            else:
                tb[0:3] = tb[0:3] - RiT @ (fj.T - fi.T)
            rA, rb = tA.T @ tA, tA.T @ tb
            A[3 * i:3 * i + 6, 3 * i:3 * i + 6] += rA[:6, :6]
            b[3 * i:3 * i + 6] += rb[:6]
            A[-k_g:, -k_g:] += rA[-k_g:, -k_g:]
            b[-k_g:] += rb[-k_g:]
            A[3 * i:3 * i + 6, -k_g:] += rA[:6, -k_g:]
            A[-k_g:, 3 * i:3 * i + 6] += rA[-k_g:, :6]
        return A, b

    A, b = blocks(3, None, None)
    x = np.linalg.solve(A * 1000.0, b * 1000.0)
    g = x[ns - 3:ns].copy()
    if abs(math.sqrt(float(g @ g)) - G_norm) > (0.5 if use_wheel else 1.0):
        return None
    g0 = g / math.sqrt(float(g @ g)) * G_norm
    A, b = np.zeros((n * 3 + 2, n * 3 + 2)), np.zeros(n * 3 + 2)
    for _ in range(4):
        lxly = tangent_basis(g0)
        dA, db = blocks(2, lxly, g0)
        A, b = (A + dA) * 1000.0, (b + db) * 1000.0
        x = np.linalg.solve(A, b)
        g0 = g0 + lxly @ x[-2:]
        g0 = g0 / math.sqrt(float(g0 @ g0)) * G_norm
    return g0, x
