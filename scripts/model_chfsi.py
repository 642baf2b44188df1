"""numpy model of the eigensolver's outer loop (Chebyshev-filtered subspace iteration) on a saved Gram matrix:
counts products / Rayleigh-Ritz steps for algorithm variants before they are written in HIP.
usage: python3 scripts/model_chfsi.py G.npy [variant options k=v ...]"""
import sys, time, numpy as np
G = np.load(sys.argv[1]); s = G.shape[0]
opt = dict(K=200, guard_pct=25, amp=8, amp_early=3, cut_pct=90, lock=0, skip_rr0=0, skip_n=1, m0=5, m1=5, lo=0.0, tol=5e-11, rr_every=3, amp_active=0, maxit=40, lock_q=16, guard_min=24, verbose=1)
for kv in sys.argv[2:]:
    k, v = kv.split("="); opt[k] = float(v) if "." in v or "e" in v else int(v)
K = opt["K"]; tol = opt["tol"]
guard = max(opt["guard_min"], K * opt["guard_pct"] // 100); b = (K + guard + 15) // 16 * 16
rng = np.random.default_rng(0)
Q = np.linalg.qr(rng.uniform(-1, 1, (s, b)))[0]
prods = 0; nrr = 0; rrdims = []
def cheb(A, B, c, e, sigma1, m):
    # p(G) A given B = G A ; returns filtered block (three-term recurrence as in eig.hip)
    global prods
    sigma = sigma1
    prev = A; cur = (sigma1 / e) * (B - c * A)
    for deg in range(2, m + 1):
        sn = 1.0 / (2.0 / sigma1 - sigma)
        nxt = (2 * sn / e) * (G @ cur - c * cur) - sigma * sn * prev; prods += 1
        prev, cur = cur, nxt; sigma = sn
    return cur
def orth(Y):
    # symmetric (Loewdin) orthonormalisation of the column-normalised block, like the Newton-Schulz path
    d = 1.0 / np.linalg.norm(Y, axis=0); Yn = Y * d
    S = Yn.T @ Yn; w, V = np.linalg.eigh(S)
    return Yn @ (V * (1.0 / np.sqrt(np.maximum(w, 1e-300)))) @ V.T, w.max() / max(w.min(), 1e-300)
XL = np.zeros((s, 0)); thL = np.zeros(0)      # locked pairs
theta = None; rmax_prev = 1.0; since_rr = 0; rate = 0.1
trace_mean = np.trace(G) / s
for it in range(opt["maxit"]):
    ba = Q.shape[1]; Kact = K - XL.shape[1]
    Z = G @ Q; prods += 1
    do_rr = not (opt["skip_rr0"] and it < opt["skip_n"])
    near_done = rmax_prev * rate <= 4 * tol
    if it >= 3 and rmax_prev < 1e-3 and since_rr + 1 < opt["rr_every"] and not near_done: do_rr = False
    if do_rr:
        since_rr = 0
        T = Q.T @ Z; T = 0.5 * (T + T.T); th, W = np.linalg.eigh(T); o = np.argsort(-th); th = th[o]; W = W[:, o]
        nrr += 1; rrdims.append(ba)
        A = Q @ W; B = Z @ W; theta = th
        res = np.linalg.norm(B[:, :Kact] - A[:, :Kact] * th[:Kact], axis=0)
        top_all = thL[0] if len(thL) else th[0]
        rmax = res.max(); conv = res <= tol * top_all
        npre = 0
        while npre < Kact and conv[npre]: npre += 1
        if opt["verbose"]: print(f"it={it} prods={prods} ba={ba} locked={XL.shape[1]} th0={th[0]:.6f} thK={th[Kact-1]:.6f} rmax={rmax:.3e} conv={conv.sum()} prefix={npre}")
        if rmax <= tol * top_all: 
            XL = np.hstack([XL, A[:, :Kact]]); thL = np.concatenate([thL, th[:Kact]]); break
        if it >= 3 and rmax / top_all < rmax_prev: rate = min(0.5, max(0.02, (rmax / top_all / rmax_prev) ** (1.0 / max(1, since_rr + 1))))
        rmax_prev = rmax / top_all
        if opt["lock"]:
            nl = npre // opt["lock_q"] * opt["lock_q"]
            if nl and ba - nl >= 32:
                XL = np.hstack([XL, A[:, :nl]]); thL = np.concatenate([thL, th[:nl]])
                A = A[:, nl:]; B = B[:, nl:]; theta = th[nl:]; Kact -= nl; ba -= nl
    else:
        since_rr += 1; A = Q; B = Z; rmax_prev *= rate
    if theta is None:   # no Rayleigh-Ritz yet: a-priori bounds
        top = np.abs(G).sum(0).max(); cut = trace_mean; m = opt['m0'] if it == 0 else opt['m1']; c = e = 0.5 * cut; sigma1 = e / (top - c)
    else:
        top = max(theta[0], 1e-300)
        cut_pos = Kact + (ba - Kact) * opt["cut_pct"] // 100
        cut = theta[min(ba - 1, max(Kact, cut_pos - 1))]
        cut = min(cut, 0.999 * top)
        lo_ = opt['lo'] if it >= 1 else 0.0
        c = 0.5 * (cut + lo_); e = 0.5 * (cut - lo_); g1 = (top - c) / e
        ampexp = opt["amp_early"] if it < 2 else opt["amp"]
        m = int(np.floor(np.arccosh(10.0 ** ampexp) / np.arccosh(max(g1, 1 + 1e-12)))); m = max(2, min(m, opt.get("mmax", 40)))
        sigma1 = e / (top - c)
    Y = cheb(A, B, c, e, sigma1, m)
    if XL.shape[1]: Y -= XL @ (XL.T @ Y)
    if theta is not None:
        Cm = np.triu(A.T @ Y, 1); Y = Y - A @ Cm       # de-contamination against the old (sorted) Ritz vectors
    Q, cond = orth(Y)
    if cond > 1e8: Q, cond = orth(Q)
    if XL.shape[1]: Q -= XL @ (XL.T @ Q); Q, _ = orth(Q)
w = np.load("/tmp/w_c3.npy") if s == 5000 else None
print(f"RESULT iterations={it+1} products={prods} rr={nrr} rrdims={rrdims}")
if w is not None and len(thL) == K: print("max eigenvalue error", np.abs(np.sort(thL)[::-1] - w[:K]).max(), "orth", np.abs(XL.T @ XL - np.eye(K)).max(), "res", np.linalg.norm(G @ XL - XL * thL, axis=0).max())
