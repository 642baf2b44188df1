"""Run-to-run reproducibility: the same call twice must give the same bits (fixed-order reductions everywhere, no
floating-point atomics across workgroups)."""
import sys, numpy as np
sys.path.insert(0, ".")
from flgp_amd import api, synth
from oracle import flgp_oracle as O
n, d, s, r, K, m = 300000, 16, 5000, 10, 200, 500
X = synth.gaussian_mixture(n, d)
rows = np.sort(synth.random_anchor_rows(n, s)); U0 = np.asfortranarray(X[rows])
lab = O.knn(X, U0, 1)[:, 0]
U = np.asfortranarray(np.hstack([U0, np.bincount(lab, minlength=s)[:, None].astype(float)]))
models = dict(kernel="lae", gl="cluster-normalized", root=True)
H = [api.heat_kernel_covariance_cpp(X[:m], X[m:], s, r, 10.0, K, models, 1, 0.1, U=U) for _ in range(3)]
print("H identical across 3 runs:", all(np.array_equal(H[0], h) for h in H[1:]), " max|H|", np.abs(H[0]).max())
ep = [api.heat_kernel_spectrum_cpp(X[:m], X[m:], s, r, K, models, U=U) for _ in range(2)]
print("spectrum identical:", np.array_equal(ep[0].values, ep[1].values) and np.array_equal(ep[0].vectors, ep[1].vectors))
ny = [api.nystrom_eigenpair_cpp(X[:50000], U0[:2000], 1.0, 50) for _ in range(2)]
print("nystrom identical:", np.array_equal(ny[0].values, ny[1].values) and np.array_equal(ny[0].vectors, ny[1].vectors))
