"""Time flgp_dev_lae alone at BASELINE configs[2] (n=1e6, d=16, s=5000, r=10) under the given tuning knobs.
usage: python scripts/lae_time.py [knob=value ...]"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from flgp_amd import _lib, synth
from flgp_amd.pipeline import HipStages
L = _lib.lib()
for kv in sys.argv[1:]:
    k, v = kv.split("="); L.flgp_set_tuning(k.encode(), int(v))
n, d, s, r = 1000000, 16, 5000, 10
S = HipStages("cuda:0")
X_np = synth.gaussian_mixture(n, d)
X = torch.from_numpy(np.ascontiguousarray(X_np.T)).cuda()
sel = np.sort(synth.random_anchor_rows(n, s))
U = torch.from_numpy(np.ascontiguousarray(X_np[sel].T)).cuda()
A = S.anchor_prep(U)
idx, _ = S.knn(X, A, r)
for _ in range(2): S.lae(X, A, idx)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(torch.cuda.current_stream())
for _ in range(10): ei, ev = S.lae(X, A, idx)
e1.record(torch.cuda.current_stream())
torch.cuda.synchronize()
print(" ".join(sys.argv[1:]) or "default", "lae ms: %.3f" % (e0.elapsed_time(e1) / 10), "checksum %.17g" % float(ev.sum()))
