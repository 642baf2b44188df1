"""Experiment: can the Gram matrix be permuted so that most 16 x 128 blocks are empty?
Ordering from G alone: p seed columns, h-hop diffusion affinity G^h[:, seeds], nearest-seed clusters, seeds chained greedily."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flgp_amd import synth
from flgp_amd.pipeline import HipStages
n, d, s, r = int(os.environ.get("N", 1000000)), int(os.environ.get("D", 16)), int(os.environ.get("S", 5000)), int(os.environ.get("R", 10))
S = HipStages("cuda:0")
X = synth.gaussian_mixture(n, d) if d != 3 else synth.swiss_roll(n)[0]
sel = np.sort(synth.random_anchor_rows(n, s))
dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda(); dU = torch.from_numpy(np.ascontiguousarray(X[sel].T)).cuda()
anc = S.anchor_prep(dU)
idx, _d = S.knn(dX, anc, r)
idx = idx.reshape(r, -1).T.long()
P = torch.zeros((s, s), dtype=torch.float64, device="cuda")
for a in range(r):
    for b in range(r):
        P[idx[:, a], idx[:, b]] = 1.0

def occupancy(Pm, bk=16, bn=128):
    sp = (s + bn - 1) // bn * bn
    Q = torch.zeros((sp, sp), dtype=torch.bool, device="cuda"); Q[:s, :s] = Pm > 0
    blk = Q.reshape(sp // bk, bk, sp // bn, bn).any(1).any(2)
    return float(blk.float().mean())

print(f"nnz {float((P>0).float().mean())*100:.2f}%  stage occupancy, given order: {occupancy(P)*100:.1f}%")
for p, hops in [(64, 2), (128, 2), (256, 2), (128, 3), (512, 2)]:
    g = torch.Generator(device="cpu"); g.manual_seed(1)
    seeds = torch.randperm(s, generator=g)[:p].cuda()
    Pn = P / P.sum(1, keepdim=True)
    A = torch.zeros((s, p), dtype=torch.float64, device="cuda"); A[seeds, torch.arange(p, device="cuda")] = 1.0
    for _ in range(hops):
        A = Pn @ A
    lab = A.argmax(1)
    lab[A.max(1).values == 0] = p        # unreachable: own bucket at the end
    # chain the seeds greedily by affinity between clusters
    C = torch.zeros((p + 1, p + 1), dtype=torch.float64, device="cuda")
    oh = torch.nn.functional.one_hot(lab, p + 1).double()
    C = oh.T @ P @ oh
    C = C.cpu().numpy(); np.fill_diagonal(C, 0)
    order = [int(np.argmax(C.sum(1)))]; left = set(range(p + 1)) - set(order)
    while left:
        cur = order[-1]
        nxt = max(left, key=lambda q: C[cur, q])
        order.append(nxt); left.remove(nxt)
    rank = np.empty(p + 1, dtype=np.int64); rank[order] = np.arange(p + 1)
    key = torch.from_numpy(rank).cuda()[lab]
    perm = torch.argsort(key, stable=True)
    Pp = P[perm][:, perm]
    print(f"p={p} hops={hops}: stage occupancy after ordering {occupancy(Pp)*100:.1f}%   (128x128 tiles: {occupancy(Pp,128,128)*100:.1f}%)  unreachable {int((lab==p).sum())}")

# ---- reference point: k-means on the anchor coordinates (not available to the eigensolver, which sees only G)
U = torch.from_numpy(X[sel]).cuda()
for kc in (16, 32, 64):
    g = torch.Generator(device="cpu"); g.manual_seed(2)
    cen = U[torch.randperm(s, generator=g)[:kc].cuda()].clone()
    for _ in range(20):
        lab = torch.cdist(U, cen).argmin(1)
        for c in range(kc):
            mk = lab == c
            if mk.any(): cen[c] = U[mk].mean(0)
    perm = torch.argsort(lab, stable=True)
    Pp = P[perm][:, perm]
    print(f"coordinate k-means, {kc} clusters: stage occupancy {occupancy(Pp)*100:.1f}%  (128x128 tiles {occupancy(Pp,128,128)*100:.1f}%)")

# ---- adaptive seeds: farthest-point in diffusion affinity, then label smoothing
def order_from_G(p, hops, smooth):
    Pn = P / P.sum(1, keepdim=True)
    seeds = [0]
    A = torch.zeros((s, 1), dtype=torch.float64, device="cuda"); A[0, 0] = 1.0
    for _ in range(hops): A = Pn @ A
    best = A[:, 0].clone()
    cols = [A[:, 0]]
    for q in range(1, p):
        nxt = int(torch.argmin(best))
        e = torch.zeros((s, 1), dtype=torch.float64, device="cuda"); e[nxt, 0] = 1.0
        for _ in range(hops): e = Pn @ e
        cols.append(e[:, 0]); best = torch.maximum(best, e[:, 0]); seeds.append(nxt)
    A = torch.stack(cols, 1)
    lab = A.argmax(1)
    for _ in range(smooth):      # label propagation: each anchor takes the label with the largest weight among its neighbours
        oh = torch.nn.functional.one_hot(lab, p).double()
        lab = (P @ oh).argmax(1)
    oh = torch.nn.functional.one_hot(lab, p).double()
    C = (oh.T @ P @ oh).cpu().numpy(); np.fill_diagonal(C, 0)
    order = [int(np.argmax(C.sum(1)))]; left = set(range(p)) - set(order)
    while left:
        cur = order[-1]; nxt = max(left, key=lambda q: C[cur, q]); order.append(nxt); left.remove(nxt)
    rank = np.empty(p, dtype=np.int64); rank[order] = np.arange(p)
    key = torch.from_numpy(rank).cuda()[lab]
    return torch.argsort(key, stable=True)
for p, hops, smooth in [(16, 3, 0), (24, 3, 2), (32, 3, 2), (48, 3, 3), (32, 4, 5)]:
    perm = order_from_G(p, hops, smooth)
    Pp = P[perm][:, perm]
    print(f"adaptive seeds p={p} hops={hops} smooth={smooth}: stage occupancy {occupancy(Pp)*100:.1f}%  (128x128 tiles {occupancy(Pp,128,128)*100:.1f}%)")

if d == 16:
    # ---- ground truth: mixture component of every anchor
    cen = (2.0 * synth.normal(20241022, 1, 16 * d)).reshape(16, d)
    Ut = torch.from_numpy(X[sel]).cuda()
    lab = torch.cdist(Ut, torch.from_numpy(cen).cuda()).argmin(1)
    perm = torch.argsort(lab, stable=True)
    Pp = P[perm][:, perm]
    oh = torch.nn.functional.one_hot(lab, 16).double()
    C = oh.T @ (P > 0).double() @ oh
    print("true components: sizes", torch.bincount(lab).tolist())
    print(f"  stage occupancy {occupancy(Pp)*100:.1f}%  (128x128 tiles {occupancy(Pp,128,128)*100:.1f}%); cross-component nonzeros {float(C.sum()-C.diag().sum()):.0f} of {float(C.sum()):.0f}")

def split_stats(Pm, name, thr=32, bk=16, bn=128):
    sp = (s + bn - 1) // bn * bn
    Q = torch.zeros((sp, sp), dtype=torch.float64, device="cuda"); Q[:s, :s] = (Pm > 0).double()
    cnt = Q.reshape(sp // bk, bk, sp // bn, bn).sum(3).sum(1)
    dense = cnt >= thr
    rem = float(cnt[~dense].sum())
    print(f"{name}: blocks with >= {thr} nonzeros {float(dense.float().mean())*100:.1f}% of {cnt.numel()}; remainder {rem:.0f} nonzeros ({100*rem/float(cnt.sum()):.1f}%)")
split_stats(P, "given order")
for p, hops, smooth in [(32, 3, 2), (64, 3, 2)]:
    perm = order_from_G(p, hops, smooth)
    split_stats(P[perm][:, perm], f"adaptive seeds p={p}")
    for thr in (16, 64): split_stats(P[perm][:, perm], f"adaptive seeds p={p}", thr)
if d == 16:
    perm = torch.argsort(lab, stable=True)
    split_stats(P[perm][:, perm], "true components")
