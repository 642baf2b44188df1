"""Builds the C3 Gram matrix G = A^T A (s x s) with the CPU oracle and saves it (plus its spectrum and the
anchors) for scripts/model_chfsi.py.  usage: python3 scripts/make_c3_gram.py [outdir=/tmp/model] [n=1000000]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flgp_amd import synth
from oracle import flgp_oracle as O
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/model"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
d, s, r = 16, 5000, 10
t0 = time.time()
X = synth.gaussian_mixture(n, d)
sel = np.sort(synth.random_anchor_rows(n, s))
U = np.asfortranarray(X[sel, :])
k1 = O.knn(X, U, 1)
sizes = np.bincount(k1[:, 0], minlength=s).astype(float)
print("inputs", time.time() - t0, flush=True)
kidx = O.knn(X, U, r)
ei, ev = O.lae(X, U, r, knn_idx=kidx)
zn = O.graph_laplacian(ei, ev, s, "cluster-normalized", sizes)
av, _ = O.scale_A(ei, zn, s)
G = O.gram(ei, av, s)
print("gram", time.time() - t0, flush=True)
np.save(os.path.join(out, "G_c3.npy"), G)
np.save(os.path.join(out, "U_c3.npy"), U)
w = np.linalg.eigvalsh(G)[::-1]
np.save("/tmp/w_c3.npy", w)
print("done", time.time() - t0, "top", w[:3], "w200", w[199], "w255", w[255])
