"""One large Nystrom-extension spectrum (resident): python scripts/run_nystrom.py [n d s K reps]"""
import sys, time, numpy as np
sys.path.insert(0, ".")
from flgp_amd import api
n, d, s, K, reps = (int(v) for v in (sys.argv[1:6] + ["1000000", "16", "5000", "200", "3"][len(sys.argv) - 1:]))
rng = np.random.default_rng(4)
X = np.asfortranarray(rng.normal(size=(n, d))); U = np.asfortranarray(X[rng.permutation(n)[:s]])   # column-major like an R matrix: no layout copy in the binding
for i in range(reps):
    t0 = time.perf_counter(); rp = api.nystrom_eigenpair_cpp(X, U, 1.0, K, resident=True); t1 = time.perf_counter()
    print(f"n={n} d={d} s={s} K={K}: resident {1e3*(t1-t0):.1f} ms", flush=True)
    rp.free()
