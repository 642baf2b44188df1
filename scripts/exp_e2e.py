"""Host-boundary timing: flgp_heat_kernel_covariance with host buffers in and out (what R sees), C3 shape."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flgp_amd import api, synth
n, d, s, r, K, m = 1000000, 16, 5000, 10, 200, int(os.environ.get("M", "1000"))
X = synth.gaussian_mixture(n, d)
sel = np.sort(synth.random_anchor_rows(n, s))
U0 = synth.anchors_from_rows(X, sel)
idx = api.KNN_cpp(U0, U0, 1)["ind_knn"]  # warm the library
sizes = np.bincount(api.KNN_cpp(X[:200000], U0, 1)["ind_knn"].ravel(), minlength=s).astype(np.float64) + 1.0
U = np.column_stack([U0, sizes])
for rep in range(2):
    t0 = time.perf_counter()
    H = api.heat_kernel_covariance_cpp(X[:m], X[m:], s, r, 10.0, K, {"subsample": "random", "kernel": "lae", "gl": "cluster-normalized", "root": True}, 1, 0.5, U=U)
    t1 = time.perf_counter()
    print(f"m={m} host-to-host {t1 - t0:.3f} s  H {H.shape} {H.nbytes/1e9:.1f} GB  -> {n/(t1-t0):.3g} points/s", flush=True)
