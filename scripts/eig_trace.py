"""GPU box: build the C3 Gram matrix once, then run the top-K eigensolver a few times (for rocprofv3 --kernel-trace).
usage: python3 scripts/eig_trace.py [reps] [key=value tuning ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from flgp_amd import _lib, synth
from flgp_amd.pipeline import HeatKernelPath, HipStages, PathConfig

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n, d, s, r, K = int(os.environ.get("N", 1_000_000)), 16, int(os.environ.get("S", 5000)), 10, int(os.environ.get("K", 200))
dev = torch.device("cuda", 0)
S = HipStages(dev); P = HeatKernelPath(S); L = _lib.lib()
for kv in sys.argv[2:]:
    k, v = kv.split("="); L.flgp_set_tuning(k.encode(), int(v))
X_np = synth.gaussian_mixture(n, d)
X = torch.from_numpy(np.ascontiguousarray(X_np.T)).to(dev)
sel = np.sort(synth.random_anchor_rows(n, s))
U = torch.from_numpy(np.ascontiguousarray(X_np[sel, :].T)).to(dev)
anchors = S.anchor_prep(U)
num_class = P.cluster_sizes(X, anchors)
knn_idx, _ = S.knn(X, anchors, r)
ei, ev = S.lae(X, anchors, knn_idx)
csc = S.csc(ei, s)
c = S.colsum(ei, ev, s); S.col_scale(ei, ev, c, num_class, 0); S.row_normalize(ev)
c2 = S.colsum(ei, ev, s); S.col_scale(ei, ev, c2, None, 1)
G = S.gram(ei, ev, csc)
torch.cuda.synchronize()
jtrace = None
if os.environ.get("JAC_TRACE"):
    jtrace = torch.zeros((1 + 8 * 4096,), dtype=torch.int64, device=dev)
ts = []
for it in range(reps):
    if jtrace is not None and it == reps - 1:
        jtrace.zero_(); torch.cuda.synchronize(); L.flgp_dev_jac_set_trace(jtrace.data_ptr())
    t0 = time.perf_counter()
    try:
        eig, V, info = S.eig_topk(G, K)
    except Exception as ex:              # (diagnostic knobs that break the arithmetic end here; the trace below is still valid)
        print("eig_topk failed:", str(ex)[:200]); info = {}
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("eig ms:", " ".join("%.2f" % t for t in ts), info, "top", float(eig[0]) if info else None, "K-th", float(eig[K - 1]) if info else None)
if jtrace is not None:
    L.flgp_dev_jac_set_trace(None)
    T = jtrace.cpu().numpy()
    nrec = min(int(T[0]), 4096); R = T[1:1 + 8 * nrec].reshape(nrec, 8)
    print("jacobi visits recorded:", nrec)
    for cross in (0, 1):
        m = (R[:, 6] == cross) & (R[:, 5] > 0)
        if not m.any(): continue
        d = np.diff(R[m][:, :6], axis=1) / 100.0
        print("cross_only=%d: %d launches; us per phase (load, gram, solve, apply+store B, V): %s; total %.1f" %
              (cross, m.sum(), " ".join("%.1f" % x for x in d.mean(0)), d.sum(1).mean()))
    skipped = (R[:, 5] == 0).sum()
    print("visits of workgroup 0 that rotated nothing:", skipped)
