"""CPU: Gram matrices of the SE-kernel similarity at BASELINE configs[2] size for two neighbouring bandwidths of the
reference's grid (a2s = exp(seq(log 0.1, log 10, length 10)), R/Fit.R:128-130), plus the top eigenvectors of the first --
the input of the warm-start question (scripts/model_chfsi2.py start=...).  usage: make_se_gram.py [outdir] [i0]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flgp_amd import synth
from oracle import flgp_oracle as O
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/model"
i0 = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n, d, s, r, K = 1_000_000, 16, 5000, 10, 200
X = synth.gaussian_mixture(n, d)
sel = np.sort(synth.random_anchor_rows(n, s))
U = np.asfortranarray(X[sel, :])
sizes = np.bincount(O.knn(X, U, 1)[:, 0], minlength=s).astype(float)
kidx, kdist = O.knn(X, U, r, output=True)
a2s = np.exp(np.linspace(np.log(0.1), np.log(10.0), 10))
mean = kdist.mean()
for i in (i0, i0 + 1):
    ei, ev = O.se_weights(kidx, kdist, np.sqrt(a2s[i] * mean / 4.0))
    zn = O.graph_laplacian(ei, ev, s, "cluster-normalized", sizes)
    av, _ = O.scale_A(ei, zn, s)
    G = O.gram(ei, av, s)
    np.save(os.path.join(out, f"G_se{i}.npy"), G)
    w, V = np.linalg.eigh(G)
    np.save(os.path.join(out, f"w_se{i}.npy"), w[::-1])
    rng = np.random.default_rng(1)
    start = np.hstack([V[:, ::-1][:, :K], rng.uniform(-1, 1, (s, 56))])
    np.save(os.path.join(out, f"start_se{i}.npy"), start)
    print(i, a2s[i], "lambda", w[::-1][[0, 15, 16, 199, 200, 255, -1]], flush=True)
