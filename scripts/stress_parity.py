"""Random-shape stress of the whole path against the oracle (GPU box): python scripts/stress_parity.py [cases] [seed]"""
import sys, time, numpy as np
sys.path.insert(0, ".")
from flgp_amd import api
from oracle import flgp_oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(cases):
    d = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 17, 32, 40, 64]))
    s = int(rng.integers(12, 400))
    # degenerate inputs are left out: r = 1 makes G the identity (every vector an eigenvector), an SE bandwidth far
    # below the neighbour distances underflows Z to zero, and K = s with few rows per anchor has singular values at
    # rounding level whose left vectors are arbitrary in the reference as well
    r = int(rng.integers(2, min(s, 24) + 1))
    n = int(rng.integers(max(4 * s, 50), 8000))
    m = int(rng.integers(1, min(n, 300) + 1))
    K = int(rng.integers(1, min(s, 60) + 1)) if rng.random() < 0.85 else -1
    kernel = str(rng.choice(["lae", "se"])); gl = str(rng.choice(["rw", "normalized", "cluster-normalized"]))
    root = bool(rng.integers(0, 2)); t = float(rng.choice([0.1, 1.0, 10.0]))
    eps = float(rng.choice([0.5, 1.0, 3.0])) * np.sqrt(d)          # 4 eps^2 comparable to the squared neighbour distances
    X = rng.normal(size=(n, d)) + 3.0 * rng.integers(0, 3, size=(n, 1))
    rows = np.sort(rng.choice(n, size=s, replace=False))
    U0 = X[rows] + 1e-3 * rng.normal(size=(s, d))
    lab = O.knn(X, U0, 1)[:, 0]
    U = np.asfortranarray(np.hstack([U0, np.bincount(lab, minlength=s)[:, None].astype(float)]))
    tag = f"n={n} d={d} s={s} r={r} m={m} K={K} {kernel}/{gl} root={root} t={t}"
    try:
        H = api.heat_kernel_covariance_cpp(X[:m], X[m:], s, r, t, K, dict(kernel=kernel, gl=gl, root=root), 1, eps, U=U)
        Ho = O.heat_kernel_covariance(X[:m], X[m:], U, r, t, K=K, kernel=kernel, gl=gl, root=root, epsilon=eps)
        err = np.abs(H - Ho).max() / max(np.abs(Ho).max(), 1e-300)
        ok = err < 1e-7
        if not ok and K > 0 and K < s:      # a truncation through a cluster of eigenvalues is ill-posed: show the gap
            X_all = np.vstack([X[:m], X[m:]])
            v, _ = O.heat_kernel_spectrum(np.asfortranarray(X_all), U, r, min(K + 1, s), kernel, gl, root, eps, "auto")
            tag += f"  [values K-1..K+1: {v[max(K - 2, 0):K + 1]}]"
    except Exception as e:
        err = float("nan"); ok = False; tag += f"  EXC {type(e).__name__}: {e}"
    if not ok and len(sys.argv) > 3:        # third argument: locate the first stage that differs
        try:
            X_all = np.asfortranarray(np.vstack([X[:m], X[m:]]))
            oi = O.knn(X_all, U0, r); gi = api.KNN_cpp(X_all, U0, r)["ind_knn"]
            tag += f"\n     knn idx differ in {np.sum(np.any(oi != gi, axis=1))} rows"
            ei, ev = O.lae(X_all, U0, r)
            Zg = api.LAE_cpp(X_all, U0, r)
            zg = Zg.data.reshape(n, r); jg = Zg.indices.reshape(n, r)
            tag += f"; lae idx equal {np.array_equal(jg, ei)}, max |dz| {np.abs(zg - ev).max():.2e} rows>1e-12: {np.sum(np.abs(zg - ev).max(1) > 1e-12)}"
            if kernel == "lae":
                ci, cv = O.cross_similarity(X_all, U, r, gl=gl, kernel="lae")
                Cg = api.cross_similarity_lae_cpp(X_all, U, r, gl)
                tag += f"; cross-sim max |d| {np.abs(Cg.data.reshape(n, r) - cv).max():.2e}"
                vo, Vo = O.spectrum_from_Z(ci, cv, s, K if K > 0 else s, root)
                epg = api.spectrum_from_Z_cpp(Cg, K, root)
                tag += f"; values max rel {np.max(np.abs(epg.values - vo) / np.abs(vo)):.2e}"
        except Exception as e2:
            tag += f"\n     debug failed: {type(e2).__name__}: {e2}"
    bad += (not ok)
    print(("ok  " if ok else "BAD ") + f"{err:.2e}  {tag}", flush=True)
print(f"{cases - bad}/{cases} within 1e-7")
