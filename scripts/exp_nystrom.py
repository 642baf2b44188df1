"""Nystrom-extension spectrum: GPU entry point vs the numpy restatement (diagnostics + timing)."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from flgp_amd import api
from oracle import flgp_oracle as o

def case(n, d, s, a2, K, seed=0, check=True):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, d)); U = X[rng.permutation(n)[:s]] + 0.01 * rng.normal(size=(s, d))
    t0 = time.perf_counter(); ep = api.nystrom_eigenpair_cpp(X, U, a2, K); t1 = time.perf_counter()
    t2 = time.perf_counter(); ep = api.nystrom_eigenpair_cpp(X, U, a2, K); t3 = time.perf_counter()
    print(f"n={n} d={d} s={s} a2={a2} K={K}: gpu {1e3*(t1-t0):.1f} ms first, {1e3*(t3-t2):.1f} ms second", flush=True)
    if not check:
        return
    t0 = time.perf_counter(); vals, vecs = o.np_nystrom_eigenpair(X, U, a2, K); t1 = time.perf_counter()
    print(f"  numpy {1e3*(t1-t0):.1f} ms; values[0..3]={vals[:4]} last={vals[-1]:.3e}")
    print("  values max rel err", np.max(np.abs(ep.values - vals) / np.abs(vals)))
    sg = np.sign(np.sum(ep.vectors * vecs, axis=0))
    err = np.max(np.abs(ep.vectors * sg - vecs), axis=0) / np.max(np.abs(vecs), axis=0)
    gap = np.minimum(np.abs(np.diff(vals, prepend=np.inf)), np.abs(np.diff(vals, append=-np.inf))) / vals[0]
    print("  vector rel err (first 6)", err[:6], "max", err.max(), "max err*gap", np.max(err * gap))
    for t in (0.1, 1.0, 10.0):
        Hg = (ep.vectors[:200] * np.exp(-t * (1 - ep.values))) @ ep.vectors[:300].T
        Hr = (vecs[:200] * np.exp(-t * (1 - vals))) @ vecs[:300].T
        print(f"  t={t}: H rel err {np.max(np.abs(Hg - Hr)) / np.max(np.abs(Hr)):.3e}")

case(3000, 3, 300, 1.0, 30)
case(5000, 7, 500, 0.5, 60, seed=1)
case(2000, 16, 257, 10.0, 20, seed=2)
case(20000, 33, 1000, 2.0, 100, seed=3)
case(1000000, 16, 5000, 1.0, 200, seed=4, check=False)
