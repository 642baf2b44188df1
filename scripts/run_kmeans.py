"""Device Lloyd k-means at bench size: python scripts/run_kmeans.py [n d s iter_max]"""
import sys, time, numpy as np
sys.path.insert(0, ".")
from flgp_amd import api, synth
n, d, s, itmax = (int(v) for v in (sys.argv[1:5] + ["1000000", "16", "5000", "100"][len(sys.argv) - 1:]))
X = synth.gaussian_mixture(n, d, components=16, seed=20241022)
rows = np.random.default_rng(0).choice(n, size=s, replace=False)
for rep in range(2):
    t0 = time.perf_counter(); U, it, wss = api.kmeans_lloyd(X, s, rows, iter_max=itmax); t1 = time.perf_counter()
    print(f"n={n} d={d} s={s}: {it} rounds, {1e3*(t1-t0):.1f} ms ({1e3*(t1-t0)/max(it,1):.2f} ms/round incl. upload), wss {wss:.6e}, "
          f"sizes min {U[:, d].min():.0f} max {U[:, d].max():.0f}", flush=True)

# mini-batch k-means with the reference's parameters at the same size (second half)
if len(sys.argv) > 1 and sys.argv[-1] == "minibatch":
    import time as _t
    from flgp_amd import api as _api
    t0 = _t.perf_counter()
    U_mb, info_mb, wss_mb = _api.kmeans_minibatch(X, s, seed=1)
    print("minibatch k-means n=%d d=%d s=%d: %.1f ms, %d iterations, within-SS %.6g" % (X.shape[0], X.shape[1], s, (_t.perf_counter() - t0) * 1e3, info_mb[0], wss_mb))
