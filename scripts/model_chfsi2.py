"""numpy model (round 4) of the eigensolver's outer loop with LOCKING of converged leading pairs and the filter's
amplification cap taken at the ACTIVE top Ritz value.  Mirrors eig.hip's defaults (a-priori first filter of degree m0,
second of degree m1, amp 1e3 early / 1e8 late, cut at 90 % of the guard, lower interval end = lambda_min from Lanczos).
usage: python3 scripts/model_chfsi2.py G.npy [k=v ...]"""
import sys, numpy as np
G = np.load(sys.argv[1]); s = G.shape[0]
opt = dict(K=200, b=256, amp=8.0, amp_early=3.0, cut_pct=90, m0=8, m1=6, lo=0.113, tol=5e-11, maxit=40, mmax=40,
           lock=0, lock_q=16, lock_min_it=2, amp_lock=8.0, start="random", nsm=1, seed=0, verbose=1, proj_every=0, landing=1,
           lock_tol_mult=1.0, soft=0, soft_tol=5e-11, soft_q=1, stale=0, predict=0, safety=10.0)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    try: opt[k] = int(v)
    except ValueError:
        try: opt[k] = float(v)
        except ValueError: opt[k] = v
K = opt["K"]; b = opt["b"]; tol = opt["tol"]
rng = np.random.default_rng(opt["seed"])
prods = 0
def orth(Y):
    d = 1.0 / np.linalg.norm(Y, axis=0); Yn = Y * d
    S = Yn.T @ Yn; w, V = np.linalg.eigh(S)
    return Yn @ (V * (1.0 / np.sqrt(np.maximum(w, 1e-300)))) @ V.T, w.max() / max(w.min(), 1e-300)
def cheb(A, B, c, e, sigma1, m, XL=None, proj_every=0):
    global prods
    sigma = sigma1
    prev = A; cur = (sigma1 / e) * (B - c * A)
    for deg in range(2, m + 1):
        sn = 1.0 / (2.0 / sigma1 - sigma)
        nxt = (2 * sn / e) * (G @ cur - c * cur) - sigma * sn * prev; prods += 1
        if XL is not None and proj_every and deg % proj_every == 0:
            nxt -= XL @ (XL.T @ nxt); cur = cur - XL @ (XL.T @ cur)
        prev, cur = cur, nxt; sigma = sn
    return cur
if opt["start"] == "random":
    Q = orth(rng.uniform(-1, 1, (s, b)))[0]
else:
    Q = np.load(opt["start"])[:, :b]; Q = orth(Q)[0]
XL = np.zeros((s, 0)); thL = np.zeros(0)
theta = None; rmax_prev = 1.0; res_prev = None; res_cur = None; plan_prev = None; theta_prev = None; plan_cur = None
trace_mean = np.trace(G) / s; n1 = np.abs(G).sum(0).max()
for it in range(opt["maxit"]):
    ba = Q.shape[1]; L = XL.shape[1]; Kact = K - L
    Z = G @ Q; prods += 1
    if it >= 1 or opt["start"] != "random":
        T = Q.T @ Z; T = 0.5 * (T + T.T); th, W = np.linalg.eigh(T); o = np.argsort(-th); th = th[o]; W = W[:, o]
        A = Q @ W; B = Z @ W; theta = th
        res = np.linalg.norm(B[:, :Kact] - A[:, :Kact] * th[:Kact], axis=0)
        top_all = thL[0] if L else th[0]
        rmax = res.max(); conv = res <= tol * top_all * opt["lock_tol_mult"]
        npre = 0
        while npre < Kact and conv[npre]: npre += 1
        if opt["verbose"]: print(f"it={it} prods={prods} ba={ba} locked={L} th0={th[0]:.6f} thK={th[Kact-1]:.6f} rmax={rmax:.3e} prefix={npre}")
        if (res <= tol * top_all).all():
            XL = np.hstack([XL, A[:, :Kact]]); thL = np.concatenate([thL, th[:Kact]]); break
        rmax_prev = rmax / top_all
        res_prev = res_cur; res_cur = res.copy(); theta_prev = theta_cur if 'theta_cur' in globals() else None; theta_cur = th.copy(); plan_prev = plan_cur
        if opt["lock"] and it >= opt["lock_min_it"]:
            nl = npre // opt["lock_q"] * opt["lock_q"]
            if nl and ba - nl >= 32:
                XL = np.hstack([XL, A[:, :nl]]); thL = np.concatenate([thL, th[:nl]])
                A = A[:, nl:]; B = B[:, nl:]; theta = th[nl:]; Kact -= nl; ba -= nl; L += nl
    else:
        A = Q; B = Z
    if theta is None:
        top = n1; cut = min(max(trace_mean, 1e-3 * n1), 0.5 * n1); m = opt["m0"]; c = e = 0.5 * cut; sigma1 = e / (top - c)
    else:
        top = max(theta[0], 1e-300)                      # ACTIVE top: the filter is scaled to 1 there
        top_true = top
        if opt["soft"] and it >= opt["lock_min_it"]:
            ns = 0
            use = res
            if opt["stale"] and it >= 3 and res_prev is not None:
                use = res_prev
                if opt["predict"] and plan_prev is not None:
                    cp, ep, mp = plan_prev
                    gj = np.maximum((theta_prev[:Kact] - cp) / ep, 1.0)
                    use = res_prev * np.minimum(1.0, opt["safety"] * 2.0 * np.exp(-mp * np.arccosh(gj)))
            while ns < Kact and use[ns] <= opt["soft_tol"] * top_all: ns += 1
            ns = ns // opt["soft_q"] * opt["soft_q"]
            top = max(theta[ns], 1e-300)
        cut_pos = Kact + (ba - Kact) * opt["cut_pct"] // 100
        cut = min(theta[min(ba - 1, max(Kact, cut_pos - 1))], 0.999 * top)
        lo_ = opt["lo"] if opt["lo"] < 0.5 * cut else 0.0
        c = 0.5 * (cut + lo_); e = 0.5 * (cut - lo_); g1 = (top - c) / e
        if it < 2 and opt["start"] == "random": ampexp = opt["amp_early"]
        else: ampexp = opt["amp_lock"] if L else opt["amp"]
        m = int(np.floor(np.arccosh(10.0 ** ampexp) / np.arccosh(max(g1, 1 + 1e-12)))); m = max(2, min(m, opt["mmax"]))
        if it == 1 and opt["start"] == "random": m = opt["m1"]
        if opt["landing"] and it >= 3 and rmax_prev < 1e-6:
            gK = (theta[Kact - 1] - c) / e
            if gK > 1 + 1e-9:
                a = np.arccosh(gK); Lg = np.log(rmax_prev / (tol * 0.4)); per0 = m * a - np.log(2)
                if Lg > 0 and per0 > 0:
                    n0 = max(1, int(np.ceil(Lg / per0))); dfor = lambda n: int(np.ceil((Lg / n + np.log(2)) / a))
                    mm = dfor(n0)
                    if n0 >= 2 and dfor(n0 - 1) <= m + 3: mm = dfor(n0 - 1)
                    m = max(2, min(mm, opt["mmax"]))
        sigma1 = e / (top - c)
        if opt["soft"]: sigma1 = e / (top_true - c)
        if opt["verbose"]: print(f"     filter: top={top:.4f} cut={cut:.4f} lo={lo_:.3f} m={m}")
    plan_cur = (c, e, m)
    Y = cheb(A, B, c, e, sigma1, m, XL if L else None, opt["proj_every"])
    if L: Y -= XL @ (XL.T @ Y)
    if theta is not None:
        Cm = np.triu(A.T @ Y, 1); Y = Y - A @ Cm
    Q, cond = orth(Y)
    if cond > 1e8: Q, cond = orth(Q)
    if L: Q -= XL @ (XL.T @ Q); Q, _ = orth(Q)
w = np.load("/tmp/w_c3.npy") if s == 5000 else None
print(f"RESULT iterations={it+1} products={prods} locked={XL.shape[1]}")
if w is not None and len(thL) >= K:
    o = np.argsort(-thL)[:K]
    print("max eigenvalue error", np.abs(thL[o] - w[:K]).max(), "orth", np.abs(XL[:, o].T @ XL[:, o] - np.eye(K)).max(), "res", np.linalg.norm(G @ XL[:, o] - XL[:, o] * thL[o], axis=0).max())
