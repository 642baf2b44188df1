"""Random-shape stress of the remaining API surface: SE bandwidth grid, resident EigenPair (HK with arbitrary index
sets, VtV / VtY / VC), lae_eigenmap, spectrum_from_Z with K = -1.  python scripts/stress_api.py [cases] [seed]"""
import sys, numpy as np
sys.path.insert(0, ".")
from flgp_amd import api
from oracle import flgp_oracle as O
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
def report(ok, msg):
    global bad
    bad += (not ok)
    print(("ok  " if ok else "BAD ") + msg, flush=True)
def hk_err(vals_g, vecs_g, vals_o, vecs_o, K, t, i0, i1):
    Hg = O.np_hk(vals_g, vecs_g, K, t, i0, i1); Ho = O.np_hk(vals_o, vecs_o, K, t, i0, i1)
    return np.abs(Hg - Ho).max() / max(np.abs(Ho).max(), 1e-300)
for c in range(cases):
    d = int(rng.choice([2, 3, 6, 16, 20, 40])); s = int(rng.integers(20, 300)); n = int(rng.integers(4 * s, 5000))
    r = int(rng.integers(2, min(s, 16) + 1)); K = int(rng.integers(2, min(s, 40) + 1)); m = int(rng.integers(5, 200))
    gl = str(rng.choice(["rw", "normalized", "cluster-normalized"])); root = bool(rng.integers(0, 2))
    X = rng.normal(size=(n, d)) + 3.0 * rng.integers(0, 3, size=(n, 1))
    U0 = X[np.sort(rng.choice(n, size=s, replace=False))] + 1e-3 * rng.normal(size=(s, d))
    lab = O.knn(X, U0, 1)[:, 0]
    U = np.asfortranarray(np.hstack([U0, np.bincount(lab, minlength=s)[:, None].astype(float)]))
    tag = f"n={n} d={d} s={s} r={r} K={K} {gl} root={root}"
    try:
        a2s = np.exp(rng.uniform(np.log(0.3), np.log(5.0), size=3))
        pairs, mean = api.se_spectrum_grid(X[:m], X[m:], s, r, K=K, a2s=a2s, models=dict(gl=gl, root=root), U=U, max_parallel=3)
        ref, mean_o = O.se_spectrum_grid(X, U, r, K, a2s, gl=gl, root=root)
        i0 = np.arange(n, dtype=np.int32); i1 = np.arange(m, dtype=np.int32)
        e = max(hk_err(p.values, p.vectors, vo, Vo, K, 1.0, i0, i1) for p, (vo, Vo) in zip(pairs, ref))
        ve = max(np.max(np.abs(p.values - vo) / np.abs(vo)) for p, (vo, _) in zip(pairs, ref))
        report(e < 1e-7 and ve < 1e-9 and abs(mean - mean_o) <= 1e-12 * mean_o, f"se-grid  {tag} H {e:.1e} values {ve:.1e}")
        models = dict(kernel="lae", gl=gl, root=root)
        ep = api.heat_kernel_spectrum_cpp(X[:m], X[m:], s, r, K, models, U=U)
        rp = api.heat_kernel_spectrum_resident(X[:m], X[m:], s, r, K, models, U=U)
        idx0 = rng.permutation(n)[: int(rng.integers(1, 400))].astype(np.int32); idx1 = rng.permutation(n)[: int(rng.integers(1, 100))].astype(np.int32)
        Kq = int(rng.integers(1, K + 1)); t = float(rng.choice([0.2, 2.0]))
        Hr = rp.HK_from_spectrum_cpp(Kq, t, idx0, idx1); Hn = O.np_hk(ep.values, ep.vectors, Kq, t, idx0, idx1)
        e1 = np.abs(Hr - Hn).max() / np.abs(Hn).max()
        V = ep.vectors[idx0][:, :Kq]; Y = rng.normal(size=(idx0.size, 2)); C = rng.normal(size=(Kq, 3))
        e2 = max(np.abs(rp.VtV(Kq, idx0) - V.T @ V).max() / max(np.abs(V.T @ V).max(), 1), np.abs(rp.VtY(Kq, idx0, Y) - V.T @ Y).max() / max(np.abs(V.T @ Y).max(), 1),
                 np.abs(rp.VC(Kq, idx0, C) - V @ C).max() / max(np.abs(V @ C).max(), 1))
        rp.free()
        report(e1 < 1e-12 and e2 < 1e-12, f"resident {tag} HK {e1:.1e} VtV/VtY/VC {e2:.1e}")
        em = api.lae_eigenmap(X, s, r, K, norm=gl, U=U)
        vo, Vo = O.heat_kernel_spectrum(np.asfortranarray(X), U, r, K, "lae", gl, True, 0.1, "auto")
        report(np.max(np.abs(em["eigenvalues"] - (1 - vo))) < 1e-9, f"eigenmap {tag} values {np.max(np.abs(em['eigenvalues'] - (1 - vo))):.1e}")
        if s <= 150:
            Z = api.cross_similarity_lae_cpp(X, U, r, gl)
            epf = api.spectrum_from_Z_cpp(Z, -1, root)
            ci, cv = O.cross_similarity(X, U, r, gl=gl, kernel="lae")
            vo, Vo = O.spectrum_from_Z(ci, cv, s, s, root)
            e3 = hk_err(epf.values, epf.vectors, vo, Vo, s, 1.0, i0, i1)
            report(e3 < 1e-7, f"full-K   {tag} H {e3:.1e} smallest value {vo[-1]:.1e}")
    except Exception as ex:
        report(False, f"EXC {tag}: {type(ex).__name__}: {ex}")
print(f"bad: {bad}")
