"""Time flgp_dev_knn alone at BASELINE configs[2] shape (n=1e6, d=16, s=5000, r=10) under the given tuning knobs.
usage: python scripts/knn_time.py [n=.. d=.. s=.. r=..] [knob=value ...]"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from flgp_amd import _lib, synth
from flgp_amd.pipeline import HipStages

L = _lib.lib()
n, d, s, r = 1000000, 16, 5000, 10
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    if k in ("n", "d", "s", "r"):
        globals()[k] = int(v)
    else:
        L.flgp_set_tuning(k.encode(), int(v))
S = HipStages("cuda:0")
X_np = synth.gaussian_mixture(n, d)
X = torch.from_numpy(np.ascontiguousarray(X_np.T)).cuda()
sel = np.sort(synth.random_anchor_rows(n, s))
U = torch.from_numpy(np.ascontiguousarray(X_np[sel].T)).cuda()
A = S.anchor_prep(U)
for _ in range(2):
    S.knn(X, A, r)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(torch.cuda.current_stream())
for _ in range(10):
    S.knn(X, A, r)
e1.record(torch.cuda.current_stream())
torch.cuda.synchronize()
print(" ".join(sys.argv[1:]) or "default", "knn ms:", e0.elapsed_time(e1) / 10)
