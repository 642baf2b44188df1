"""Runs k-NN + LAE once on the C3 cloud (for rocprofv3 counter passes)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flgp_amd import synth
from flgp_amd.pipeline import HipStages
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
S = HipStages("cuda:0")
X = synth.gaussian_mixture(n, 16)
sel = np.sort(synth.random_anchor_rows(n, 5000))
dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda(); dU = torch.from_numpy(np.ascontiguousarray(X[sel].T)).cuda()
anc = S.anchor_prep(dU)
for _ in range(2):
    idx, _d = S.knn(dX, anc, 10)
    ei, ev = S.lae(dX, anc, idx)
torch.cuda.synchronize()
print("done", float(ev.sum()))
