"""GPU box: per-task timeline of one block-sparse product at BASELINE configs[2] (flgp_dev_bsg_set_trace).
usage: python3 scripts/bsg_timeline.py [key=value tuning ...]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from flgp_amd import _lib, synth
from flgp_amd.pipeline import HeatKernelPath, HipStages

n, d, s, r, b = 1_000_000, 16, 5000, 10, 256
dev = torch.device("cuda", 0)
S = HipStages(dev); P = HeatKernelPath(S); L = _lib.lib()
for kv in sys.argv[1:]:
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
rng = np.random.default_rng(0)
dX = torch.from_numpy(np.asfortranarray(rng.normal(size=(s, b))).T.copy()).to(dev)   # s x b column-major
out = torch.empty((b, s), dtype=torch.float64, device=dev)
wb = L.flgp_dev_bsg_workspace(s, b)
work = torch.empty((wb // 8 + 1,), dtype=torch.float64, device=dev)
ntask_max = 8192
trace = torch.zeros((ntask_max, 4), dtype=torch.int64, device=dev)
info = (ctypes.c_int * 6)()
st = torch.cuda.current_stream().cuda_stream
for rep in range(3):
    trace.zero_()
    L.flgp_dev_bsg_set_trace(trace.data_ptr())
    _lib.check(L.flgp_dev_bsg_apply(st, G.data_ptr(), s, s, dX.data_ptr(), b, 1.0, 0.0, None, out.data_ptr(), work.data_ptr(), wb,
                                    ctypes.addressof(info)))
    L.flgp_dev_bsg_set_trace(None)
T = trace.cpu().numpy()
nt = info[4] * ((b + 63) // 64)
T = T[:nt]
ok = T[:, 0] > 0
t0 = T[ok, 0].min()
st_us = (T[:, 0] - t0) / 100.0; en_us = (T[:, 1] - t0) / 100.0
ns = T[:, 3] & 0xffffffff; wg = T[:, 3] >> 32
xcc = (T[:, 2] >> 56) & 0xf; hw = (T[:, 2] >> 40) & 0xffff; cyc = T[:, 2] & 0xffffffffff
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7          # HW_ID: [11:8] CU, [15:13] SE (layout as on gfx9)
print("tasks", nt, "info", list(info), "span %.1f us" % (en_us[ok].max()))
dur = en_us - st_us
order = np.argsort(-dur)
print("task  wg  xcc se cu  entries  start   end    dur   us/pair  MHz")
for q in list(order[:12]) + list(order[-6:]):
    print("%4d %4d  %d  %d %2d   %3d   %6.1f %6.1f %6.1f  %.2f  %.0f" % (q, wg[q], xcc[q], se[q], cu[q], ns[q], st_us[q], en_us[q], dur[q], dur[q] / max(1, (ns[q] + 1) // 2), cyc[q] / max(dur[q], 1e-9)))
# fit: dur = a + c * pairs
pairs = (ns + 1) // 2
A = np.stack([np.ones(nt), pairs], 1)
coef = np.linalg.lstsq(A[ok], dur[ok], rcond=None)[0]
print("fit over tasks: dur = %.2f us + %.3f us x pairs" % tuple(coef))
first = wg == np.arange(nt) if False else None
# how many tasks per workgroup, and the busiest workgroups
import collections
per = collections.defaultdict(list)
for q in range(nt):
    if ok[q]: per[int(wg[q])].append(q)
busy = sorted(per.items(), key=lambda kv: -max(en_us[q] for q in kv[1]))[:6]
for w, qs in busy:
    print("wg %d: " % w + ", ".join("task %d [%0.1f-%0.1f] n=%d" % (q, st_us[q], en_us[q], ns[q]) for q in sorted(qs, key=lambda q: st_us[q])))
print("start times: min %.1f  median %.1f  p90 %.1f max %.1f (first tasks only)" % tuple(np.percentile(np.array([min(st_us[q] for q in qs) for qs in per.values()]), [0, 50, 90, 100])))
# occupancy per XCC: tasks and summed pairs
for x in range(8):
    m = ok & (xcc == x)
    print("xcc %d: %3d tasks, %4d pairs, last end %.1f" % (x, m.sum(), pairs[m].sum(), en_us[m].max() if m.any() else 0))
