"""GPU box: the eigensolver at BASELINE configs[2] over several start blocks (eig_start_stream = 0..N-1) under one or more
knob sets, all in ONE process (the Gram matrix is built once).  Prints per knob set: min-of-reps time per start block,
iterations, products and their means; checks the eigenvalues against torch.linalg.eigvalsh(G) once.
usage: python3 scripts/eig_seeds.py <nseeds> <reps> "k=v k=v" "k=v" ...      ("" = defaults)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from flgp_amd import _lib, synth
from flgp_amd.pipeline import HeatKernelPath, HipStages

nseeds = int(sys.argv[1]); reps = int(sys.argv[2]); sets = sys.argv[3:] or [""]
n, d, s, r, K = int(os.environ.get("N", 1_000_000)), 16, int(os.environ.get("S", 5000)), 10, int(os.environ.get("K", 200))
dev = torch.device("cuda", 0)
S = HipStages(dev); P = HeatKernelPath(S); L = _lib.lib()
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
wref = torch.linalg.eigvalsh(G).flip(0)[:K].cpu().numpy()
seen = set()
for ks in sets:
    kv = dict(x.split("=") for x in ks.split())
    assert not (seen - set(kv)), "a knob set must name every knob an earlier set changed (there is no reset): %s" % (seen - set(kv))
    for k, v in kv.items(): L.flgp_set_tuning(k.encode(), int(v)); seen.add(k)
    ts, its, prs, errs = [], [], [], []
    for sd in range(nseeds):
        L.flgp_set_tuning(b"eig_start_stream", sd)
        best = None
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            try:
                eig, V, info = S.eig_topk(G, K)
            except Exception as ex:
                print("  failed:", str(ex)[:160]); info = None; break
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        if info is None: continue
        ts.append(best); its.append(info["outer_iterations"]); prs.append(info["g_products"])
        errs.append(float(np.abs(eig.cpu().numpy() - wref).max()))
    L.flgp_set_tuning(b"eig_start_stream", 0)
    if ts:
        print("[%s] times %s | it %s | prods %s | mean %.2f ms, %.2f it, %.1f prods | max eigenvalue err %.1e" %
              (ks, " ".join("%.2f" % t for t in ts), its, prs, sum(ts) / len(ts), sum(its) / len(its), sum(prs) / len(prs), max(errs)), flush=True)
