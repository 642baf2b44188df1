"""Experiment: LAE time per shape (r, d) and kernel variant on a 1e6-point cloud."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flgp_amd import synth, _lib
from flgp_amd.pipeline import HipStages
n = 1000000
S = HipStages("cuda:0")
L = _lib.lib()
shapes = [(3, 2), (5, 3), (3, 16), (5, 16), (8, 16), (10, 8), (10, 12), (10, 16), (12, 16), (16, 16), (10, 32), (10, 64), (16, 32)]
variants = [(0, 0, 1), (4, 1, 1), (8, 1, 1), (8, 2, 1), (4, 4, 1), (8, 4, 1), (0, 0, 0)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for r, d in shapes:
    X = synth.gaussian_mixture(n, d)
    sel = np.sort(synth.random_anchor_rows(n, 5000))
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda(); dU = torch.from_numpy(np.ascontiguousarray(X[sel].T)).cuda()
    anc = S.anchor_prep(dU)
    idx, _d = S.knn(dX, anc, r)
    out = []
    for dpl, lp, var in variants:
        if dpl and dpl * lp < d:
            continue
        L.flgp_set_tuning(b"lae_dpl", dpl); L.flgp_set_tuning(b"lae_lp", lp); L.flgp_set_tuning(b"lae_variant", var)
        ei, ev = S.lae(dX, anc, idx); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); ei, ev = S.lae(dX, anc, idx); e1.record(); torch.cuda.synchronize()
        out.append(f"{'lds' if not var else ('auto' if not dpl else f'{dpl}x{lp}')}:{e0.elapsed_time(e1):7.3f}")
    print(f"r={r:2d} d={d:2d}  " + "  ".join(out), flush=True)
