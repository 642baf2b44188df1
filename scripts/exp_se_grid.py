"""Experiment: SE bandwidth grid at BASELINE configs[1] size, concurrent vs sequential."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flgp_amd import api, synth
from oracle import flgp_oracle as O
n, d, s, r, K, m = 100000, 3, 2000, 5, 100, 1000
X, y = synth.swiss_roll(n)
sel = np.sort(synth.random_anchor_rows(n, s))
U0 = np.asfortranarray(X[sel])
sizes = np.bincount(api.KNN_cpp(X, U0, 1)["ind_knn"][:, 0], minlength=s).astype(float)
U = np.asfortranarray(np.hstack([U0, sizes[:, None]]))
for par in (1, 2, 5, 10, 10):
    t0 = time.perf_counter()
    pairs, mean = api.se_spectrum_grid(X[:m], X[m:], s, r, K=K, U=U, max_parallel=par)
    print(f"max_parallel={par}: {time.perf_counter()-t0:.3f} s  (values[0][:3]={pairs[0].values[:3]})", flush=True)
t0 = time.perf_counter()
ep = api.heat_kernel_spectrum_cpp(X[:m], X[m:], s, r, K=K, models=dict(kernel="lae", gl="cluster-normalized", root=True), U=U)
print(f"single LAE spectrum (host entry point, incl. transfers): {time.perf_counter()-t0:.3f} s")
