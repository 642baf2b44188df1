"""Experiment: how many anchors does an exact k-NN have to look at if points are grouped by nearest centroid and a wave
skips the anchor clusters that the triangle inequality rules out for all of its 64 points?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flgp_amd import synth
n, d, s, r = int(os.environ.get("N", 1000000)), int(os.environ.get("D", 16)), int(os.environ.get("S", 5000)), int(os.environ.get("R", 10))
X = synth.gaussian_mixture(n, d) if d != 3 else synth.swiss_roll(n)[0]
sel = np.sort(synth.random_anchor_rows(n, s))
U = torch.from_numpy(X[sel]).cuda()
Xt = torch.from_numpy(np.ascontiguousarray(X)).cuda()
for C in (32, 64, 128):
    cen_idx = torch.arange(C, device="cuda") * (s // C)
    cen = U[cen_idx]
    dU = torch.cdist(U, cen)                    # s x C
    alab = dU.argmin(1)
    R = torch.zeros(C, dtype=torch.float64, device="cuda")
    R.scatter_reduce_(0, alab, dU.gather(1, alab[:, None])[:, 0], reduce="amax")
    csize = torch.bincount(alab, minlength=C)
    # points: nearest centroid, sort
    samp = torch.randperm(n, device="cuda")[:200000]
    Xs = Xt[samp]
    dP = torch.cdist(Xs, cen)                   # m x C
    plab = dP.argmin(1)
    order = torch.argsort(plab, stable=True)
    Xs, dP, plab = Xs[order], dP[order], plab[order]
    # waves of 64 consecutive points
    W = Xs.shape[0] // 64
    examined = 0
    D_all = torch.cdist(Xs[:W * 64], U) ** 2    # true distances (for tau after own cluster)
    for w in range(0, W, max(1, W // 300)):
        sl = slice(w * 64, w * 64 + 64)
        own = int(plab[sl][0])
        # tau after scanning the clusters in order: own first, then by index; simulate exactly
        tau = torch.full((64,), float("inf"), dtype=torch.float64, device="cuda")
        best = torch.full((64, r), float("inf"), dtype=torch.float64, device="cuda")
        seq = [own] + [c for c in range(C) if c != own]
        cnt = 0
        for c in seq:
            lb = torch.clamp(dP[sl, c] - R[c], min=0) ** 2
            if bool((lb <= tau).any()):
                mem = (alab == c).nonzero()[:, 0]
                cnt += mem.numel()
                cand = torch.cat([best, D_all[sl][:, mem]], 1)
                best = torch.topk(cand, r, dim=1, largest=False).values
                tau = best[:, -1]
        examined += cnt
    nw = len(range(0, W, max(1, W // 300)))
    print(f"C={C}: anchors examined per wave {examined / nw:.0f} of {s} ({100.0 * examined / nw / s:.1f} %), cluster sizes {int(csize.min())}..{int(csize.max())}")
