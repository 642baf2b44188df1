"""Random-shape stress of the other entry points (k-NN, SE similarity, Nystrom, Lloyd) against the oracle."""
import sys, numpy as np
sys.path.insert(0, ".")
from flgp_amd import api
from oracle import flgp_oracle as O
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
def report(ok, msg):
    global bad
    bad += (not ok)
    print(("ok  " if ok else "BAD ") + msg, flush=True)
for c in range(cases):
    d = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 48, 64]))
    s = int(rng.integers(2, 700)); n = int(rng.integers(s, 9000)); r = int(rng.integers(1, min(s, 32) + 1))
    X = rng.normal(size=(n, d)) + 3.0 * rng.integers(0, 3, size=(n, 1))
    if rng.random() < 0.3:
        X = np.round(X)                                   # lattice: exact ties
    U0 = X[np.sort(rng.choice(n, size=s, replace=False))]
    tag = f"n={n} d={d} s={s} r={r}"
    try:
        res = api.KNN_cpp(X, U0, r, output=True); oi, od = O.knn(X, U0, r, output=True)
        order = np.argsort(oi, axis=1, kind="stable")
        ok = np.array_equal(res["ind_knn"], oi) and np.array_equal(res["distances_sp"].data.reshape(n, r), np.take_along_axis(od, order, axis=1))
        report(ok, "knn      " + tag)
        gl = str(rng.choice(["rw", "normalized"])); eps = float(rng.choice([0.5, 2.0])) * np.sqrt(d)
        Zs = api.cross_similarity_se_cpp(X, U0, r, gl, eps); ci, cv = O.cross_similarity(X, U0, r, gl=gl, kernel="se", epsilon=eps)
        ok = np.array_equal(Zs.indices.reshape(n, r), ci) and np.abs(Zs.data.reshape(n, r) - cv).max() <= 1e-14
        report(ok, f"se-sim   {tag} {gl} max|d|={np.abs(Zs.data.reshape(n, r) - cv).max():.1e}")
        Zl = api.cross_similarity_lae_cpp(X, U0, r, gl); ci, cv = O.cross_similarity(X, U0, r, gl=gl, kernel="lae")
        ok = np.array_equal(Zl.indices.reshape(n, r), ci) and np.array_equal(Zl.data.reshape(n, r), cv)
        report(ok, f"lae-sim  {tag} {gl} max|d|={np.abs(Zl.data.reshape(n, r) - cv).max():.1e}")
        if s >= 8 and n <= 4000 and d <= 64:
            K = int(rng.integers(1, min(s, 30) + 1)); a2 = float(rng.choice([0.3, 1.0, 5.0]))
            Xc = X + 0.01 * rng.normal(size=X.shape); Uc = U0 + 0.01 * rng.normal(size=U0.shape)      # no coincident anchors
            vals, vecs = O.np_nystrom_eigenpair(Xc, Uc, a2, K); ep = api.nystrom_eigenpair_cpp(Xc, Uc, a2, K)
            relv = np.max(np.abs(ep.values - vals) / np.abs(vals))
            sg = np.sign(np.sum(ep.vectors * vecs, axis=0))
            e0 = np.abs(ep.vectors[:, 0] * sg[0] - vecs[:, 0]).max() / np.abs(vecs[:, 0]).max()
            report(relv < 1e-9 and e0 < 1e-9, f"nystrom  {tag} K={K} a2={a2} values {relv:.1e} vec0 {e0:.1e}")
        if s <= 300:
            rows = rng.choice(n, size=s, replace=False); itmax = int(rng.choice([1, 3, 100]))
            Uo, ito = O.np_kmeans_lloyd(X, rows, itmax); Ug, itg, _ = api.kmeans_lloyd(X, s, rows, iter_max=itmax)
            report(itg == ito and np.array_equal(Ug, Uo), f"lloyd    {tag} iter_max={itmax} rounds {itg}/{ito}")
    except Exception as e:
        report(False, f"EXC {tag}: {type(e).__name__}: {e}")
print(f"bad: {bad}")
