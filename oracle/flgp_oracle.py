"""CPU parity oracle for the FLGP heat-kernel covariance path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package (flgp_amd) never does and fails loudly without its HIP
library instead of falling back to anything in here.

PARITY UNPINNED: the reference ships no tests or golden vectors for this path and cannot
be built in this image (no R / Rcpp / RcppEigen / RcppParallel / RSpectra).  What pins
this oracle: hand-derived known answers, algebraic invariants and the two independent
restatements in this file checking each other (C via ctypes, and pure numpy).

Two layers:
  * ``C``  -- ctypes bindings to oracle/libflgp_oracle.so (flgp_oracle.c): the fast,
    multi-threaded restatement whose arithmetic order the HIP kernels share.
  * ``np_*`` -- a second, independent restatement in plain numpy / Python loops written
    from SURVEY.md Appendix A; used on small cases to check the C code, and to produce
    the fixtures under tests/golden (tests/golden/make_golden.py).

The truncated SVD (src/TruncatedSVD.cpp:9-34) is third-party arithmetic: the reference
calls RSpectra::svds (Spectra's implicitly restarted Lanczos on the s x s operator
A^T A, tol 1e-10, ncv = max(2K+1, 20); version unpinned, source not in /root/reference)
or Eigen::BDCSVD when K == s.  Restated here with the same algorithm families from
scipy: ARPACK's implicitly restarted Lanczos (``scipy.sparse.linalg.svds``) and LAPACK's
divide-and-conquer SVD (``numpy.linalg.svd``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libflgp_oracle.so")

GL_CODES = {"rw": 0, "normalized": 1, "cluster-normalized": 2}


def build(force: bool = False) -> str:
    """Compile flgp_oracle.c with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "flgp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "libflgp_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int)
        ci, cd = ctypes.c_int, ctypes.c_double
        L.flgp_oracle_threads.restype = ci
        L.flgp_oracle_knn.argtypes = [dp, ci, ci, dp, ci, ci, ip, dp]
        L.flgp_oracle_v_to_z.argtypes = [dp, ci, dp]
        L.flgp_oracle_lae_point.argtypes = [dp, ci, dp, ci, dp]
        L.flgp_oracle_lae.argtypes = [dp, ci, ci, dp, ci, ci, ip, ip, dp, ip]
        L.flgp_oracle_se_weights.argtypes = [ip, dp, ci, ci, cd, ip, dp]
        L.flgp_oracle_colsum.argtypes = [ip, dp, ci, ci, ci, dp]
        L.flgp_oracle_graph_laplacian.argtypes = [ip, dp, ci, ci, ci, ci, dp]
        L.flgp_oracle_scale_A.argtypes = [ip, dp, ci, ci, ci, dp]
        L.flgp_oracle_gram.argtypes = [ip, dp, ci, ci, ci, dp]
        L.flgp_oracle_ell_matvec.argtypes = [ip, dp, ci, ci, dp, dp]
        L.flgp_oracle_ell_matvec.restype = None
        L.flgp_oracle_u_recover.argtypes = [ip, dp, ci, ci, ci, dp, dp, ci, dp]
        L.flgp_oracle_hk.argtypes = [dp, dp, ci, ci, cd, ip, ci, ip, ci, dp]
        _lib = L
    return _lib


def threads() -> int:
    return int(lib().flgp_oracle_threads())


def _d(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)) if a is not None else None


def _i(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int)) if a is not None else None


def _f64(a):
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _check(rc, what):
    if rc < 0:
        raise RuntimeError(f"oracle {what} failed with code {rc}")


# ----------------------------------------------------------------------------------
# C-backed restatement
# ----------------------------------------------------------------------------------
def knn(X, U, r, output=False):
    """KNN_cpp (src/Utils.cpp:102-192). Returns ind_knn (n x r, 0-based) [, dist (n x r)]."""
    X = _f64(X); U = _f64(U)
    n, d = X.shape; s = U.shape[0]
    assert U.shape[1] == d
    idx = np.zeros((n, r), dtype=np.int32, order="F")
    dist = np.zeros((n, r), dtype=np.float64, order="F") if output else None
    _check(lib().flgp_oracle_knn(_d(X), n, d, _d(U), s, r, _i(idx), _d(dist)), "knn")
    return (idx, dist) if output else idx


def v_to_z(v):
    """v_to_z_cpp (src/lae.cpp:137-153)."""
    v = np.ascontiguousarray(v, dtype=np.float64).ravel()
    z = np.zeros_like(v)
    _check(lib().flgp_oracle_v_to_z(_d(v), v.size, _d(z)), "v_to_z")
    return z


def local_anchor_embedding(x, U):
    """local_anchor_embedding_cpp(x, U) (src/lae.cpp:76-133); U is r x d."""
    x = np.ascontiguousarray(x, dtype=np.float64).ravel()
    U = _f64(U)
    r, d = U.shape
    z = np.zeros(r)
    _check(lib().flgp_oracle_lae_point(_d(x), d, _d(U), r, _d(z)), "lae_point")
    return z


def lae(X, U, r, knn_idx=None, return_iters=False):
    """LAE_cpp (src/lae.cpp:48-70) -> ELL (idx n x r, val n x r; rows sorted by column)."""
    X = _f64(X); U = _f64(U)
    n, d = X.shape; s = U.shape[0]
    if knn_idx is None:
        knn_idx = knn(X, U, r)
    knn_idx = np.asfortranarray(knn_idx, dtype=np.int32)
    eidx = np.zeros((n, r), dtype=np.int32)
    eval_ = np.zeros((n, r), dtype=np.float64)
    iters = np.zeros(n, dtype=np.int32)
    _check(lib().flgp_oracle_lae(_d(X), n, d, _d(U), s, r, _i(knn_idx), _i(eidx), _d(eval_), _i(iters)), "lae")
    return (eidx, eval_, iters) if return_iters else (eidx, eval_)


def se_weights(knn_idx, knn_dist, epsilon):
    """exp(-dist/(4 eps^2)) on the stored entries (src/Spectrum.cpp:126-132) -> ELL."""
    knn_idx = np.asfortranarray(knn_idx, dtype=np.int32)
    knn_dist = _f64(knn_dist)
    n, r = knn_idx.shape
    eidx = np.zeros((n, r), dtype=np.int32)
    eval_ = np.zeros((n, r), dtype=np.float64)
    _check(lib().flgp_oracle_se_weights(_i(knn_idx), _d(knn_dist), n, r, float(epsilon), _i(eidx), _d(eval_)), "se")
    return eidx, eval_


def colsum(eidx, eval_, s):
    n, r = eidx.shape
    c = np.zeros(s)
    _check(lib().flgp_oracle_colsum(_i(eidx), _d(eval_), n, s, r, _d(c)), "colsum")
    return c


def graph_laplacian(eidx, eval_, s, gl, num_class=None):
    """graphLaplacian_cpp (src/Utils.cpp:195-212). Returns a new val array."""
    if gl not in GL_CODES:
        raise ValueError("Error: the type of graph Laplacian is not supported!")
    n, r = eidx.shape
    out = np.array(eval_, dtype=np.float64, order="C", copy=True)
    nc = None if num_class is None else np.ascontiguousarray(num_class, dtype=np.float64)
    _check(lib().flgp_oracle_graph_laplacian(_i(eidx), _d(out), n, s, r, GL_CODES[gl], _d(nc)), "graph_laplacian")
    return out


def scale_A(eidx, eval_, s):
    """A = Z diag(1/sqrt(|colsum|+1e-9)) (src/Spectrum.cpp:149-150). Returns (A_val, colsum)."""
    n, r = eidx.shape
    out = np.array(eval_, dtype=np.float64, order="C", copy=True)
    c = np.zeros(s)
    _check(lib().flgp_oracle_scale_A(_i(eidx), _d(out), n, s, r, _d(c)), "scale_A")
    return out, c


def gram(eidx, aval, s):
    n, r = eidx.shape
    G = np.zeros((s, s))
    _check(lib().flgp_oracle_gram(_i(eidx), _d(aval), n, s, r, _d(G)), "gram")
    return G


def u_recover(eidx, aval, s, V, sigma):
    n, r = eidx.shape
    V = _f64(V); K = V.shape[1]
    sigma = np.ascontiguousarray(sigma, dtype=np.float64)
    out = np.zeros((n, K), order="F")
    _check(lib().flgp_oracle_u_recover(_i(eidx), _d(aval), n, s, r, _d(V), _d(sigma), K, _d(out)), "u_recover")
    return out


def ell_to_csr(eidx, eval_, s):
    import scipy.sparse as sp
    n, r = eidx.shape
    indptr = np.arange(0, n * r + 1, r, dtype=np.int64)
    return sp.csr_matrix((eval_.ravel(), eidx.ravel(), indptr), shape=(n, s))


def truncated_svd(eidx, aval, s, K, method="auto", seed=0):
    """truncated_SVD_cpp (src/TruncatedSVD.cpp:9-34): returns (values = sigma^2 desc, U n x K).

    method: 'dense' -> LAPACK SVD of the densified A (the K == s BDCSVD branch, :17-20);
            'svds'  -> ARPACK implicitly restarted Lanczos on A^T A (the RSpectra::svds
                       branch, :23-30; tol 1e-10, ncv = max(2K+1, 20) as Spectra's defaults);
            'gram'  -> dense symmetric eigendecomposition of the C-built Gram matrix, then
                       u = A v / sigma: the route the HIP path takes, kept here so tests can
                       separate "route" differences from kernel bugs;
            'auto'  -> 'dense' if K == s else 'svds'.
    """
    n, r = eidx.shape
    if K < 0:
        K = s
    if method == "auto":
        method = "dense" if K == s else "svds"
    if method == "dense":
        A = ell_to_csr(eidx, aval, s).toarray()
        Uu, sv, _ = np.linalg.svd(A, full_matrices=False)
        return (sv[:K] ** 2).copy(), np.asfortranarray(Uu[:, :K])
    if method == "svds":
        from scipy.sparse.linalg import svds
        A = ell_to_csr(eidx, aval, s).tocsc()
        ncv = min(min(n, s) - 1, max(2 * K + 1, 20))  # ARPACK needs k < ncv < min(A.shape)
        v0 = np.random.default_rng(seed).standard_normal(min(n, s))
        Uu, sv, _ = svds(A, k=K, ncv=ncv, tol=1e-10, which="LM", v0=v0, maxiter=1000 * s,
                         return_singular_vectors="u")
        order = np.argsort(-sv, kind="stable")
        return (sv[order] ** 2).copy(), np.asfortranarray(Uu[:, order])
    if method == "gram":
        G = gram(eidx, aval, s)
        w, V = np.linalg.eigh(G)
        w = w[::-1][:K].copy(); V = V[:, ::-1][:, :K]
        sig = np.sqrt(np.maximum(w, 0.0))
        Uu = u_recover(eidx, aval, s, V, sig) / np.sqrt(float(n))
        return w, np.asfortranarray(Uu)
    raise ValueError(method)


def spectrum_from_Z(eidx, zval, s, K, root=False, method="auto"):
    """spectrum_from_Z_cpp (src/Spectrum.cpp:146-161) -> (values K, vectors n x K)."""
    n = eidx.shape[0]
    aval, _ = scale_A(eidx, zval, s)
    values, Uu = truncated_svd(eidx, aval, s, K, method=method)
    if root:
        values = np.sqrt(values)
    return values, np.asfortranarray(Uu * np.sqrt(float(n)))


def hk_from_spectrum(values, vectors, K, t, idx0, idx1):
    """HK_from_spectrum_cpp (src/Spectrum.cpp:83-94)."""
    vectors = _f64(vectors)
    n = vectors.shape[0]
    values = np.ascontiguousarray(values[:K], dtype=np.float64)
    idx0 = np.ascontiguousarray(idx0, dtype=np.int32); idx1 = np.ascontiguousarray(idx1, dtype=np.int32)
    H = np.zeros((idx0.size, idx1.size), order="F")
    _check(lib().flgp_oracle_hk(_d(values), _d(vectors), n, K, float(t), _i(idx0), idx0.size,
                                _i(idx1), idx1.size, _d(H)), "hk")
    return H


def cross_similarity(X, U, r, gl="rw", kernel="lae", epsilon=0.1):
    """cross_similarity_lae_cpp / cross_similarity_se_cpp (src/Spectrum.cpp:101-142).

    U is s x d, or s x (d+1) with cluster sizes in the last column.  Returns ELL (idx, val).
    """
    X = _f64(X); U = _f64(U)
    d = X.shape[1]; s = U.shape[0]
    Ud = np.asfortranarray(U[:, :d])
    if gl == "cluster-normalized":
        if U.shape[1] < d + 1:
            raise ValueError("cluster-normalized needs cluster sizes in column d of U")
        num_class = np.ascontiguousarray(U[:, d])
    else:
        num_class = None
    if kernel == "lae":
        eidx, zval = lae(X, Ud, r)
    elif kernel == "se":
        kidx, kdist = knn(X, Ud, r, output=True)
        eidx, zval = se_weights(kidx, kdist, epsilon)
    else:
        raise ValueError("The kernel type is not supported!")
    return eidx, graph_laplacian(eidx, zval, s, gl, num_class)


def se_spectrum_grid(X_all, U, r, K, a2s, gl="cluster-normalized", root=True, method="auto"):
    """Spectrum part of fit_se_*_gp_cpp (src/Fit.cpp:127-178): one k-NN with distances, then per
    bandwidth a2: Z = exp(-dist/(a2*mean(dist))) (:150), graphLaplacian_cpp (:152-156),
    spectrum_from_Z_cpp (:158).  Returns ([(values, vectors)], distances_mean)."""
    X_all = _f64(X_all); U = _f64(U)
    d = X_all.shape[1]; s = U.shape[0]
    if K < 0:
        K = s
    kidx, kdist = knn(X_all, np.asfortranarray(U[:, :d]), r, output=True)
    mean = float(kdist.sum() / kdist.size)                       # distances_sp.coeffs().sum()/(n*r) (:131)
    num_class = np.ascontiguousarray(U[:, d]) if gl == "cluster-normalized" else None
    out = []
    for a2 in a2s:
        order = np.argsort(kidx, axis=1, kind="stable")
        eidx = np.ascontiguousarray(np.take_along_axis(kidx, order, axis=1), dtype=np.int32)
        zval = np.ascontiguousarray(np.exp(-np.take_along_axis(kdist, order, axis=1) / (a2 * mean)))
        zn = graph_laplacian(eidx, zval, s, gl, num_class)
        out.append(spectrum_from_Z(eidx, zn, s, K, root=root, method=method))
    return out, mean


def heat_kernel_spectrum(X_all, U, r, K, kernel="lae", gl="rw", root=False, epsilon=0.1, method="auto"):
    """heat_kernel_spectrum_cpp (src/Spectrum.cpp:48-76) with the anchors U given
    (subsample_cpp, src/Utils.cpp:32-68, is outside the path: it calls R's kmeans)."""
    s = U.shape[0]
    if K < 0:
        K = s
    eidx, zval = cross_similarity(X_all, U, r, gl=gl, kernel=kernel, epsilon=epsilon)
    return spectrum_from_Z(eidx, zval, s, K, root=root, method=method)


def heat_kernel_covariance(X, X_new, U, r, t, K=-1, kernel="lae", gl="cluster-normalized", root=True,
                           epsilon=0.1, method="auto"):
    """heat_kernel_covariance_cpp (src/Spectrum.cpp:28-43) with the anchors given;
    defaults are the R wrapper's (R/Fit.R:760-770)."""
    X = _f64(X); X_new = _f64(X_new)
    m = X.shape[0]
    X_all = np.asfortranarray(np.vstack([X, X_new]))
    n = X_all.shape[0]
    s = U.shape[0]
    if K < 0:
        K = s
    values, vectors = heat_kernel_spectrum(X_all, U, r, K, kernel, gl, root, epsilon, method)
    return hk_from_spectrum(values, vectors, K, t, np.arange(n, dtype=np.int32), np.arange(m, dtype=np.int32))


# ----------------------------------------------------------------------------------
# Independent numpy restatement (small cases only; written from SURVEY.md Appendix A)
# ----------------------------------------------------------------------------------
def np_knn(X, U, r):
    """A.1 in vectorised numpy: rounding may differ from the FMA chain in the last ulp, so
    compare index sets on tie-free data, and distances to ~1e-12."""
    X = np.asarray(X, float); U = np.asarray(U, float)
    D = ((-2.0 * X) @ U.T + (X * X).sum(1)[:, None]) + (U * U).sum(1)[None, :]
    idx = np.argsort(D, axis=1, kind="stable")[:, :r]
    return idx.astype(np.int32), np.take_along_axis(D, idx, axis=1)


def np_v_to_z(v):
    v = np.asarray(v, float)
    r = v.size
    vd = np.sort(v)[::-1]
    cs = np.cumsum(vd)
    vstar = vd - (cs - 1.0) / np.arange(1, r + 1)
    rho = r
    while rho > 0 and not vstar[rho - 1] > 0:
        rho -= 1
    theta = (vd[:rho].sum() - 1.0) / rho
    return np.maximum(v - theta, 0.0)


def np_lae_point(x, Ui, max_backtrack=64):
    """A.2 with numpy reductions (x: d, Ui: r x d)."""
    x = np.asarray(x, float); Ui = np.asarray(Ui, float)
    r = Ui.shape[0]
    Ut = Ui.T
    UUt = Ui @ Ut
    z_prev = np.full(r, 1.0 / r); z_curr = z_prev.copy()
    dp, dc, bc = 0.0, 1.0, 1.0
    for _ in range(100):
        alpha = (dp - 1.0) / dc
        v = z_curr + alpha * (z_curr - z_prev)
        g_v = ((x - v @ Ui) ** 2).sum() / 2.0
        grad = v @ UUt - x @ Ut
        j = 0
        while True:
            beta = (2.0 ** j) * bc
            z = np_v_to_z(v - 1.0 / beta * grad)
            g_z = ((x - z @ Ui) ** 2).sum() / 2.0
            g_t = g_v + grad @ (z - v) + beta * ((z - v) ** 2).sum() / 2.0
            if g_z <= g_t or j >= max_backtrack:
                bc = beta; z_prev = z_curr; z_curr = z
                break
            j += 1
        dp, dc = dc, (1.0 + np.sqrt(1.0 + 4.0 * dc * dc)) / 2.0
        if ((z_curr - z_prev) ** 2).sum() < 1e-5:
            break
    return z_curr


def np_lae_dense(X, U, r):
    """LAE_cpp as a dense n x s matrix."""
    X = np.asarray(X, float); U = np.asarray(U, float)
    idx, _ = np_knn(X, U, r)
    Z = np.zeros((X.shape[0], U.shape[0]))
    for i in range(X.shape[0]):
        Z[i, idx[i]] = np_lae_point(X[i], U[idx[i]])
    return Z, idx


def np_graph_laplacian_dense(Z, gl, num_class=None):
    Z = np.array(Z, float)
    if gl == "rw":
        pass
    elif gl in ("normalized", "cluster-normalized"):
        Z = Z * (1.0 / (Z.sum(0) + 1e-9))[None, :]
        if gl == "cluster-normalized":
            Z = Z * np.asarray(num_class, float)[None, :]
    else:
        raise ValueError("Error: the type of graph Laplacian is not supported!")
    return (1.0 / (Z.sum(1) + 1e-9))[:, None] * Z


def np_spectrum_dense(Z, K, root=False):
    """A.6 by LAPACK SVD of the dense A."""
    n = Z.shape[0]
    A = Z * (1.0 / np.sqrt(np.abs(Z.sum(0)) + 1e-9))[None, :]
    Uu, sv, _ = np.linalg.svd(A, full_matrices=False)
    vals = sv[:K] ** 2
    if root:
        vals = np.sqrt(vals)
    return vals, Uu[:, :K] * np.sqrt(float(n))


def np_hk(values, vectors, K, t, idx0, idx1):
    w = np.exp(-t * (1.0 - np.asarray(values)[:K]))
    return (vectors[idx0, :K] * w[None, :]) @ vectors[idx1, :K].T


def np_predict_regression(values, vectors, Y, idx0, idx1, K, pars, sigma):
    """predict_regression_cpp with noisepar = "same" (reference src/Predict.cpp:40-75), restated line by line in numpy:
    ``pars = (t, noise)``; the m <= K branch is GPML Algorithm 2.1 on the m x m kernel matrix, the m > K branch the
    Woodbury form on the K x K matrix Q.  Returns Y_pred (m_new x q)."""
    import scipy.linalg as sl
    values = np.asarray(values, dtype=np.float64); vectors = np.asarray(vectors, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64).reshape(len(idx0), -1)
    t, noise = float(pars[0]), float(pars[1])
    m = Y.shape[0]
    if m <= K:
        Cvv = np_hk(values, vectors, K, t, idx0, idx0)                              # :48
        C_noisy = Cvv.copy()
        C_noisy[np.diag_indices(m)] += sigma                                       # :50
        C_noisy[np.diag_indices(m)] += noise                                       # :51
        Cnv = np_hk(values, vectors, K, t, idx1, idx0)                              # :52
        alpha = sl.cho_solve(sl.cho_factor(C_noisy, lower=True), Y)                 # :55-56
        return Cnv @ alpha                                                         # :57
    lam = 1.0 - values[:K]                                                         # :60
    V = vectors[idx0, :K]                                                          # :64
    ls = np.exp(-0.5 * t * lam) + 0.0                                              # :65
    Q = (ls[:, None] * (V.T @ V)) * ls[None, :]                                    # :66
    Q[np.diag_indices(K)] += noise + sigma                                         # :67
    x = sl.cho_solve(sl.cho_factor(Q, lower=True), ls[:, None] * (V.T @ Y))        # :68-69
    alpha = 1.0 / (noise + sigma) * (Y - (V * ls[None, :]) @ x)                    # :69
    Vnv = vectors[idx1, :K]                                                        # :71
    return Vnv @ ((np.exp(-t * lam) + 0.0)[:, None] * (V.T @ alpha))               # :72


def np_predict_regression_different(values, vectors, Y, idx0, idx1, K, pars, sigma):
    """predict_regression_cpp with noisepar = "different" (reference src/Predict.cpp:76-110), line by line in numpy:
    ``pars = (t, noise_1, ..., noise_m)``, one noise variance per training row."""
    import scipy.linalg as sl
    values = np.asarray(values, dtype=np.float64); vectors = np.asarray(vectors, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64).reshape(len(idx0), -1)
    t = float(pars[0]); nz = np.asarray(pars[1:], dtype=np.float64)
    m = Y.shape[0]
    if m <= K:
        C_noisy = np_hk(values, vectors, K, t, idx0, idx0)                          # :79
        C_noisy[np.diag_indices(m)] += sigma                                       # :81
        C_noisy[np.diag_indices(m)] += nz                                          # :82-84
        Cnv = np_hk(values, vectors, K, t, idx1, idx0)                              # :85
        alpha = sl.cho_solve(sl.cho_factor(C_noisy, lower=True), Y)                 # :88-89
        return Cnv @ alpha                                                         # :90
    lam = 1.0 - values[:K]                                                         # :93
    V = vectors[idx0, :K]                                                          # :97
    ls = np.exp(-0.5 * t * lam) + 0.0                                              # :98
    zinv = 1.0 / (nz + sigma)                                                      # :99-102
    VtZV = V.T @ (zinv[:, None] * V)                                               # :103
    Q = (ls[:, None] * VtZV) * ls[None, :]                                         # :104
    Q[np.diag_indices(K)] += 1.0                                                   # :105
    ZY = zinv[:, None] * Y
    x = sl.cho_solve(sl.cho_factor(Q, lower=True), ls[:, None] * (V.T @ ZY))       # :106-107
    alpha = ZY - zinv[:, None] * ((V * ls[None, :]) @ x)                           # :107
    Vnv = vectors[idx1, :K]                                                        # :109
    return Vnv @ ((np.exp(-t * lam) + 0.0)[:, None] * (V.T @ alpha))               # :110


def np_posterior_covariance_regression(values, vectors, idx0, idx1, K, pars, sigma):
    """posterior_covariance_regression (reference src/Utils.cpp:214-250) in numpy: ``pars = (t, var)``; returns the
    posterior variance of the rows idx1 (m_new,)."""
    import scipy.linalg as sl
    values = np.asarray(values, dtype=np.float64); vectors = np.asarray(vectors, dtype=np.float64)
    m = len(idx0)
    t, var = float(pars[0]), float(pars[1])
    lam = 1.0 - values[:K]                                                         # :220
    V2 = vectors[idx1, :K]                                                         # :223
    L = np.exp(-t * lam)                                                           # :224
    if m <= K:
        C11 = np_hk(values, vectors, K, t, idx0, idx0)                              # :228
        K11 = C11.copy(); K11[np.diag_indices(m)] += var + sigma                   # :229-230
        C21 = np_hk(values, vectors, K, t, idx1, idx0)                              # :231
        alpha = C21 @ sl.cho_solve(sl.cho_factor(K11, lower=True), np.eye(m))       # :233-234
        beta = (C21 * alpha).sum(1)                                                # :235
    else:
        V1 = vectors[idx0, :K]                                                     # :237
        ls = np.exp(-0.5 * t * lam) + 0.0                                          # :238
        VtV = V1.T @ V1
        Q = (ls[:, None] * VtV) * ls[None, :]                                      # :239
        Q[np.diag_indices(K)] += var + sigma                                       # :240
        inner = V1 - (V1 * ls[None, :]) @ sl.cho_solve(sl.cho_factor(Q, lower=True), ls[:, None] * VtV)   # :242
        alpha = 1.0 / (var + sigma) * ((L[:, None] * (V1.T @ inner)) * L[None, :])  # :242
        beta = (V2 * (V2 @ alpha)).sum(1)                                          # :243
    return ((V2 * L[None, :]) * V2).sum(1) + var + sigma - beta                    # :246


def np_nystrom_eigenpair(X_all, U, a2, K):
    """The Nystrom-extension spectrum the ``fit_nystrom_*`` drivers build per bandwidth ``a2`` (reference
    src/Fit.cpp:244-286; identical in :399-441, :918-960, :1063-1105 ...), restated line by line in numpy.
    ``eigs_sym`` (RSpectra, largest magnitude) is replaced by LAPACK ``eigh`` + the K largest eigenvalues: W_UU is a
    congruence of the positive-definite Gaussian kernel matrix, so largest magnitude = largest.
    Returns (values (K,), vectors (n, K)); eigenvector signs are arbitrary, as in the reference."""
    X_all = np.asarray(X_all, dtype=np.float64); U = np.asarray(U, dtype=np.float64)
    s = U.shape[0]
    uu = (U * U).sum(1)
    D_UU = (-2.0 * U @ U.T + uu[:, None]) + uu[None, :]                       # :244
    D_XU = (-2.0 * X_all @ U.T + (X_all * X_all).sum(1)[:, None]) + uu[None, :]   # :245
    mean = D_UU.sum() / (s * s)                                             # :248
    Z_UU = np.exp(-D_UU / (a2 * mean))                                      # :266
    rs_UU = Z_UU.sum(1) + 1e-9                                              # :267
    A_UU = (Z_UU / rs_UU[:, None]) / rs_UU[None, :]                         # :268
    sd = 1.0 / np.sqrt(A_UU.sum(1) + 1e-9)                                  # :269
    W_UU = (A_UU * sd[:, None]) * sd[None, :]                               # :270
    w, V = np.linalg.eigh(0.5 * (W_UU + W_UU.T))
    order = np.argsort(-w)[:K]
    values = w[order]; V = V[:, order]                                      # :273-276
    V = sd[:, None] * V                                                     # :278
    V = np.sqrt(s) * V / (np.linalg.norm(V, axis=0) + 1e-9)[None, :]        # :279-280
    Z_XU = np.exp(-D_XU / (a2 * mean))                                      # :283
    rs_XU = Z_XU.sum(1) + 1e-9                                              # :284
    A_XU = (Z_XU / rs_XU[:, None]) / rs_UU[None, :]                         # :285
    W_XU = A_XU / (A_XU.sum(1) + 1e-9)[:, None]                             # :286-287
    vectors = (W_XU @ V) / (np.abs(values) + 1e-9)[None, :]                 # :289
    return values, vectors


def np_kmeans_lloyd(X, init_rows, iter_max=100):
    """Lloyd k-means with the operation order of flgp_amd/csrc/kmeans.hip -- NOT the reference's
    stats::kmeans (src/Utils.cpp:36-45: Hartigan-Wong inside R), which cannot be restated outside R.  Assignment is
    ``knn(X, C, 1)`` (A.1 arithmetic, ties to the lower centre, as src/Utils.cpp:59 does for the sizes); a centre is the
    row-ordered sum of its points divided by their count; an empty centre keeps its position with size 0.
    Returns (U (s x (d+1), sizes last), rounds)."""
    X = _f64(X)
    n, d = X.shape
    C = np.asfortranarray(X[np.asarray(init_rows, dtype=np.int64), :])
    s = C.shape[0]
    size = np.zeros(s)
    prev = np.full(n, -1, dtype=np.int32)
    it = 0
    while True:
        lab = knn(X, C, 1)[:, 0]
        if np.array_equal(lab, prev):
            break
        prev = lab.copy()
        sums = np.zeros((s, d))
        np.add.at(sums, lab, X)                       # unbuffered: rows are added in ascending order
        cnt = np.bincount(lab, minlength=s)
        keep = cnt > 0
        C = np.asfortranarray(np.where(keep[:, None], sums / np.maximum(cnt, 1)[:, None], C))
        size = cnt.astype(np.float64)
        it += 1
        if it >= iter_max:
            break
    return np.asfortranarray(np.hstack([C, size[:, None]])), it


# ---------------------------------------------------------------------------------------------------------------------
# Mini-batch k-means: the algorithm behind subsample_cpp's "minibatchkmeans" branch.
# The reference (src/Utils.cpp:49-62) calls ClusterR::MiniBatchKmeans(data = X, clusters = s, batch_size = 10 s,
# init_fraction = 20 s / n, num_init = nstart) -- a third-party R package, version unpinned in DESCRIPTION, not under
# /root/reference; its other arguments stay at their defaults: max_iters = 100, initializer = "kmeans++",
# early_stop_iter = 10 -- and then counts 1-NN assignments (KNN_cpp(X, centres, 1)).  ClusterR draws from R's RNG, so this
# restates the PUBLISHED algorithm (k-means++: Arthur & Vassilvitskii 2007; mini-batch updates: Sculley 2010) with those
# parameters on the repo's seeded counter RNG, operation for operation as flgp_amd/csrc/minibatch.hip runs it.  Parity
# unpinned like the rest of the oracle.
_M64 = (1 << 64) - 1


def _mb_mix(z):
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def _mb_uniform(seed, st, q):
    h = _mb_mix((_mb_mix((seed + 0x632BE59BD9B4E019 * (st + 1)) & _M64) + q) & _M64)
    return float(h >> 11) * (1.0 / 9007199254740992.0)


def np_kmeans_minibatch(X, s, batch_size=-1, num_init=1, max_iters=100, init_fraction=-1.0, early_stop_iter=10, seed=0):
    """Returns (U (s x (d+1): centres, then 1-NN sizes over all rows), info = (iterations, winning start), tot_withinss).
    batch_size <= 0: the reference's 10 s; init_fraction <= 0: the reference's 20 s / n (src/Utils.cpp:53-54)."""
    X = _f64(X)
    n, d = X.shape
    if batch_size <= 0:
        batch_size = min(n, 10 * s)
    if init_fraction <= 0:
        init_fraction = min(1.0, 20.0 * s / n)
    B = batch_size
    nsub = min(n, max(s, int(np.ceil(init_fraction * n))))
    nblk = (nsub + 1023) // 1024

    def two_level(v):                                             # blocks of 1024 left to right, then the block sums left to right
        bs = []
        for b in range((len(v) + 1023) // 1024):
            acc = 0.0
            for x in v[b * 1024:(b + 1) * 1024]:
                acc = acc + x
            bs.append(acc)
        return bs

    best = None
    for q in range(num_init):
        base = 4 * q
        perm = np.arange(n)
        for i in range(nsub):
            j = min(n - 1, i + int(_mb_uniform(seed, base + 0, i) * float(n - i)))
            perm[i], perm[j] = perm[j], perm[i]
        sub = perm[:nsub].copy()
        Xs = X[sub]                                               # (nsub, d)

        def dist2(c):                                             # coordinates ascending, multiply then add
            acc = np.zeros(nsub)
            for k in range(d):
                df = Xs[:, k] - c[k]
                acc = acc + df * df
            return acc

        C = np.zeros((s, d))
        first = min(nsub - 1, int(_mb_uniform(seed, base + 1, 0) * float(nsub)))
        C[0] = Xs[first]
        d2 = None
        for c in range(1, s):
            v = dist2(C[c - 1])
            d2 = v if d2 is None else np.where(v < d2, v, d2)
            bsum = two_level(d2)
            total = 0.0
            for b in range(nblk):
                total = total + bsum[b]
            target = _mb_uniform(seed, base + 1, c) * total
            acc = 0.0; blk = -1
            for b in range(nblk):
                nx = acc + bsum[b]
                if nx > target:
                    blk = b
                    break
                acc = nx
            if blk < 0:
                blk = nblk - 1
                acc = 0.0
                for b in range(blk):
                    acc = acc + bsum[b]
            i0, i1 = blk * 1024, min(blk * 1024 + 1024, nsub)
            pick = i1 - 1
            for qq in range(i0, i1):
                acc = acc + d2[qq]
                if acc > target:
                    pick = qq
                    break
            C[c] = Xs[pick]
        # mini-batch iterations: the batch is the head of the running permutation after B more Fisher-Yates draws
        cnt = np.zeros(s)
        draw = 0
        best_sse = np.inf; stall = 0; iters = 0
        for it in range(max_iters):
            for p in range(B):
                j = min(n - 1, p + int(_mb_uniform(seed, base + 2, draw) * float(n - p)))
                draw += 1
                perm[p], perm[j] = perm[j], perm[p]
            batch = perm[:B].copy()
            Xb = np.asfortranarray(X[batch])
            lab, dist = knn(Xb, np.asfortranarray(C), 1, output=True)       # the centres as they stand at the start of the iteration
            lab = lab[:, 0]; dist = dist[:, 0]
            sse = 0.0
            for x in two_level(dist):
                sse = sse + x
            for c in np.unique(lab):                              # per centre, its batch points in batch order
                vv = cnt[c]; ck = C[c].copy()
                for p in np.nonzero(lab == c)[0]:
                    vv = vv + 1.0
                    eta = 1.0 / vv
                    ck = (1.0 - eta) * ck + eta * Xb[p]
                cnt[c] = vv; C[c] = ck
            iters = it + 1
            if sse < best_sse:
                best_sse = sse; stall = 0
            else:
                stall += 1
            if stall >= early_stop_iter:
                break
        Cf = np.asfortranarray(C)
        lab = knn(X, Cf, 1)[:, 0]
        sizes = np.bincount(lab, minlength=s).astype(np.float64)
        wss = float(((X - Cf[lab]) ** 2).sum())
        if best is None or wss < best[2]:
            best = (np.asfortranarray(np.hstack([Cf, sizes[:, None]])), (iters, q), wss)
    return best
