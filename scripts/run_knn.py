"""Times the k-NN kernel alone on the C3 cloud."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flgp_amd import synth, _lib
from flgp_amd.pipeline import HipStages
L = _lib.lib()
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    L.flgp_set_tuning(k.encode(), int(v))
n = int(os.environ.get("KNN_N", "1000000")); d = int(os.environ.get("KNN_D", "16")); r = int(os.environ.get("KNN_R", "10"))
S = HipStages("cuda:0")
X = synth.gaussian_mixture(n, d)
sel = np.sort(synth.random_anchor_rows(n, 5000))
dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda(); dU = torch.from_numpy(np.ascontiguousarray(X[sel].T)).cuda()
anc = S.anchor_prep(dU)
idx, _d = S.knn(dX, anc, r); torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    idx, _d = S.knn(dX, anc, r)
e1.record(); torch.cuda.synchronize()
print(f"n={n} d={d} r={r} knn {e0.elapsed_time(e1)/3:.3f} ms")
