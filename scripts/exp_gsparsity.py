"""How sparse is the Gram matrix G = A^T A at C3?  (anchors j1, j2 couple only if some point has both among its r nearest)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flgp_amd import synth
from flgp_amd.pipeline import HipStages
n, d, s, r = int(os.environ.get("N", 1000000)), int(os.environ.get("D", 16)), int(os.environ.get("S", 5000)), int(os.environ.get("R", 10))
S = HipStages("cuda:0")
X = synth.gaussian_mixture(n, d) if d != 3 else synth.swiss_roll(n)[0]
sel = np.sort(synth.random_anchor_rows(n, s))
dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda(); dU = torch.from_numpy(np.ascontiguousarray(X[sel].T)).cuda()
anc = S.anchor_prep(dU)
idx, _d = S.knn(dX, anc, r)
idx = idx.cpu().numpy().reshape(r, -1).T.astype(np.int64)    # n x r
pairs = set()
P = np.zeros((s, s), dtype=bool)
for a in range(r):
    for b in range(r):
        P[idx[:, a], idx[:, b]] = True
nnz = int(P.sum())
rowc = P.sum(1)
print(f"n={n} d={d} s={s} r={r}: nnz(G) = {nnz} = {100.0*nnz/(s*s):.2f} % of s^2; per row mean {rowc.mean():.1f} max {rowc.max()} min {rowc.min()}")
