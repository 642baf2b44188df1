"""Time flgp_dev_lae at an image-like shape (d > 64: the kernel that reads the anchors from memory).  usage: lae_wide_time.py [n d s r]"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from flgp_amd import synth
from flgp_amd.pipeline import HipStages
n, d, s, r = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (70000, 784, 1000, 3)))
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
e0.record()
for _ in range(5): ei, ev = S.lae(X, A, idx)
e1.record(); torch.cuda.synchronize()
print("lae ms:", e0.elapsed_time(e1) / 5, "checksum", float(ev.sum()))
