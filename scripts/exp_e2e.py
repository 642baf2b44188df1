"""Host-boundary timing: flgp_heat_kernel_covariance with host buffers in and out -- the C call an R `.Call` makes
(X_all and U already exist on the R side; the result matrix is allocated, untouched, just before), C3 shape."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flgp_amd import _lib, api, synth
n, d, s, r, K, m = 1000000, 16, 5000, 10, 200, int(os.environ.get("M", "1000"))
X = synth.gaussian_mixture(n, d)
sel = np.sort(synth.random_anchor_rows(n, s))
U0 = synth.anchors_from_rows(X, sel)
sizes = np.bincount(api.KNN_cpp(X, U0, 1)["ind_knn"].ravel(), minlength=s).astype(np.float64)
U = np.asfortranarray(np.column_stack([U0, sizes]))
L = _lib.lib()
L.flgp_set_tuning(b"e2e_verbose", 1)
for rep in range(int(os.environ.get("REPS", "4"))):
    H = np.empty((n, m), order="F")          # fresh, untouched pages: what Rf_allocMatrix hands over
    t0 = time.perf_counter()
    _lib.check(L.flgp_heat_kernel_covariance(X.ctypes.data, n, m, d, U.ctypes.data, s, d + 1, r, 10.0, K, b"lae",
                                             b"cluster-normalized", 1, 0.5, H.ctypes.data))
    t1 = time.perf_counter()
    print(f"m={m} C call {t1 - t0:.3f} s  H {H.nbytes/1e9:.1f} GB  -> {n/(t1-t0):.3g} points/s", flush=True)
    del H
