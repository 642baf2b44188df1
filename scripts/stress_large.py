"""A few larger random configurations (block-sparse eigensolver range, many GEMM tiles) against the oracle:
python scripts/stress_large.py [seed] [cases] [smin] [smax]"""
import sys, time, numpy as np
sys.path.insert(0, ".")
from flgp_amd import api, synth
from oracle import flgp_oracle as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for c in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    smin = int(sys.argv[3]) if len(sys.argv) > 3 else 3072; smax = int(sys.argv[4]) if len(sys.argv) > 4 else 4600
    d = int(rng.choice([3, 8, 16, 24])); s = int(rng.integers(smin, smax)); n = int(rng.integers(40000, 120000))
    r = int(rng.integers(3, 12)); K = int(rng.integers(30, 160)); m = int(rng.integers(50, 600))
    kernel = str(rng.choice(["lae", "se"])); gl = str(rng.choice(["rw", "normalized", "cluster-normalized"])); root = bool(rng.integers(0, 2))
    X = synth.gaussian_mixture(n, d, components=int(rng.integers(4, 30)), seed=int(rng.integers(1, 1 << 30)))
    rows = np.sort(rng.choice(n, size=s, replace=False)); U0 = np.asfortranarray(X[rows])
    lab = O.knn(X, U0, 1)[:, 0]
    U = np.asfortranarray(np.hstack([U0, np.bincount(lab, minlength=s)[:, None].astype(float)]))
    eps = float(np.sqrt(np.median(O.knn(X[:2000], U0, r, output=True)[1])))
    t0 = time.perf_counter()
    H = api.heat_kernel_covariance_cpp(X[:m], X[m:], s, r, 5.0, K, dict(kernel=kernel, gl=gl, root=root), 1, eps, U=U)
    t1 = time.perf_counter()
    Ho = O.heat_kernel_covariance(X[:m], X[m:], U, r, 5.0, K=K, kernel=kernel, gl=gl, root=root, epsilon=eps)
    t2 = time.perf_counter()
    err = np.abs(H - Ho).max() / np.abs(Ho).max()
    print(("ok  " if err < 1e-7 else "BAD ") + f"{err:.2e} n={n} d={d} s={s} r={r} K={K} m={m} {kernel}/{gl} root={root}  gpu {t1-t0:.2f}s oracle {t2-t1:.1f}s", flush=True)
