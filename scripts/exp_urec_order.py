"""GPU box: does U-recovery get faster when the points are processed grouped by their nearest anchor (rows of V^T shared
between neighbours stay in L1 / L2)?  Times flgp_dev_u_recover on the ELL rows in original order and in grouped order."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from flgp_amd import _lib, synth
from flgp_amd.pipeline import HeatKernelPath, HipStages
n, d, s, r, K = 1_000_000, 16, 5000, 10, 200
dev = torch.device("cuda", 0)
S = HipStages(dev); P = HeatKernelPath(S); L = _lib.lib()
for kv in sys.argv[1:]:
    k, v = kv.split("="); L.flgp_set_tuning(k.encode(), int(v))
X_np = synth.gaussian_mixture(n, d)
X = torch.from_numpy(np.ascontiguousarray(X_np.T)).to(dev)
sel = np.sort(synth.random_anchor_rows(n, s))
U = torch.from_numpy(np.ascontiguousarray(X_np[sel, :].T)).to(dev)
anchors = S.anchor_prep(U)
knn_idx, _ = S.knn(X, anchors, r)
ei, ev = S.lae(X, anchors, knn_idx)
V = torch.randn((K, s), dtype=torch.float64, device=dev)
eig = torch.linspace(1.0, 0.2, K, dtype=torch.float64, device=dev)
def timeit(ei_, ev_, tag):
    ts = []
    for _ in range(6):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); S.u_recover(ei_, ev_, V, eig, 1000.0, True); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"{tag}: median {np.median(ts):.3f} ms")
timeit(ei, ev, "original order")
order = torch.argsort(knn_idx[0].to(torch.int64), stable=True)
timeit(ei[order].contiguous(), ev[order].contiguous(), "grouped by nearest anchor")
order2 = torch.argsort(ei[:, 0].to(torch.int64), stable=True)
timeit(ei[order2].contiguous(), ev[order2].contiguous(), "grouped by lowest anchor index")
