"""LAE parity sweep over (r, d): python scripts/sweep_lae.py"""
import sys, numpy as np
sys.path.insert(0, ".")
from flgp_amd import api
from oracle import flgp_oracle as O
rng = np.random.default_rng(0)
n, s = 600, 60
bad = []
for d in (1, 2, 3, 4, 5, 8, 9, 16, 17, 24, 31, 32, 33, 40, 48, 63, 64):
    X = rng.normal(size=(n, d)) + 3.0 * rng.integers(0, 3, size=(n, 1))
    U0 = X[np.sort(rng.choice(n, size=s, replace=False))] + 1e-3 * rng.normal(size=(s, d))
    row = []
    for r in range(1, 21):
        ei, ev = O.lae(X, U0, r)
        Z = api.LAE_cpp(X, U0, r)
        dz = np.abs(Z.data.reshape(n, r) - ev).max()
        okj = np.array_equal(Z.indices.reshape(n, r), ei)
        row.append("." if (dz == 0 and okj) else ("~" if dz < 1e-12 and okj else "X"))
        if row[-1] == "X":
            bad.append((d, r, dz))
    print(f"d={d:2d} r=1..20: {''.join(row)}", flush=True)
print("bad:", bad)
