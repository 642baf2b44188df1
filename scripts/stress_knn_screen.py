"""Random stress of the screened k-NN kernel (d <= 16, 2 <= r <= 16, s >= 512) against the oracle (GPU box): indices and
distances must be the oracle's bits over clustered, scaled, shifted, low-rank, quantised and duplicated data.
usage: python scripts/stress_knn_screen.py [cases] [seed]"""
import sys, numpy as np
sys.path.insert(0, ".")
from flgp_amd import api
from oracle import flgp_oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(cases):
    d = int(rng.integers(1, 17)); s = int(rng.integers(512, 6001)); r = int(rng.integers(2, 17))
    n = int(rng.integers(max(s, 1000), 30000))
    kind = str(rng.choice(["gauss", "clusters", "lowrank", "quantised", "heavy", "dups"]))
    if kind == "gauss":
        X = rng.normal(size=(n, d))
    elif kind == "clusters":
        k = int(rng.integers(2, 40)); X = rng.normal(size=(n, d)) * rng.uniform(0.01, 1.0) + rng.normal(size=(k, d))[rng.integers(0, k, n)] * 5
    elif kind == "lowrank":
        q = max(1, d // 3); X = rng.normal(size=(n, q)) @ rng.normal(size=(q, d)) + 1e-6 * rng.normal(size=(n, d))
    elif kind == "quantised":
        X = np.round(rng.normal(size=(n, d)) * 4) / 4          # many exact ties
    elif kind == "heavy":
        X = rng.standard_cauchy(size=(n, d))
    else:
        X = rng.normal(size=(n, d)); X[rng.integers(0, n, n // 3)] = X[rng.integers(0, n, 1)]   # a third of the rows one point
    X = X * 10.0 ** rng.integers(-8, 9) + rng.normal(size=(1, d)) * 10.0 ** rng.integers(-3, 4) * (rng.random() < 0.5)
    rows = rng.choice(n, size=s, replace=False)
    U = X[rows] + (0.0 if rng.random() < 0.5 else 1e-3 * np.abs(X).mean() * rng.normal(size=(s, d)))
    tag = f"n={n} d={d} s={s} r={r} {kind}"
    try:
        res = api.KNN_cpp(X, U, r, output=True)
        oi, od = O.knn(X, U, r, output=True)
        order = np.argsort(oi, axis=1, kind="stable")
        ok = np.array_equal(res["ind_knn"], oi) and np.array_equal(res["distances_sp"].data.reshape(n, r), np.take_along_axis(od, order, axis=1))
        if not ok: tag += f"  rows differing: {int(np.sum(np.any(res['ind_knn'] != oi, axis=1)))}"
    except Exception as e:
        ok = False; tag += f"  EXC {type(e).__name__}: {e}"
    bad += (not ok)
    print(("ok  " if ok else "BAD ") + tag, flush=True)
print(f"{cases - bad}/{cases} bit-exact")
