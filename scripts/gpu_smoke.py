"""Ad-hoc stage-by-stage check of the HIP path against the oracle (development aid)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flgp_amd import api, synth, _lib  # noqa: E402
from oracle import flgp_oracle as O  # noqa: E402


def main():
    n, d, s, r, K, m = 3000, 16, 200, 10, 40, 100
    if len(sys.argv) > 1:
        n, d, s, r, K, m = [int(v) for v in sys.argv[1:7]]
    X = synth.gaussian_mixture(n, d, components=6, seed=11)
    rows = synth.random_anchor_rows(n, s, seed=11)
    U0 = synth.anchors_from_rows(X, rows)
    sizes = np.bincount(O.knn(X, U0, 1)[:, 0], minlength=s).astype(float)
    U = np.asfortranarray(np.hstack([U0, sizes[:, None]]))
    _lib.lib().flgp_set_tuning(b"eig_verbose", 1)

    print("== knn", flush=True)
    res = api.KNN_cpp(X, U0, r, output=True)
    oi, od = O.knn(X, U0, r, output=True)
    print("idx exact", np.array_equal(res["ind_knn"], oi), flush=True)

    print("== v_to_z / lae point", flush=True)
    print(api.v_to_z_cpp([0.9, 0.3, -1.0]), flush=True)
    z = api.local_anchor_embedding_cpp(X[0], U0[oi[0]])
    zo = O.local_anchor_embedding(X[0], U0[oi[0]])
    print("lae point exact", np.array_equal(z.ravel(), zo), z.ravel()[:4], zo[:4], flush=True)

    print("== LAE", flush=True)
    Z = api.LAE_cpp(X, U0, r)
    ei, ev = O.lae(X, U0, r)
    print("LAE idx exact", np.array_equal(Z.indices.reshape(n, r), ei), "val exact", np.array_equal(Z.data.reshape(n, r), ev),
          "maxdiff", np.abs(Z.data.reshape(n, r) - ev).max(), flush=True)

    for gl in ["rw", "normalized", "cluster-normalized"]:
        Zg = api.cross_similarity_lae_cpp(X, U, r, gl)
        eo, vo = O.cross_similarity(X, U, r, gl=gl)
        dv = np.abs(Zg.data.reshape(n, r) - vo)
        print(f"cross_similarity_lae {gl}: exact {np.array_equal(Zg.data.reshape(n, r), vo)} maxdiff {dv.max():.3e}", flush=True)

    Zse = api.cross_similarity_se_cpp(X, U, r, "cluster-normalized", 0.7)
    eo, vo = O.cross_similarity(X, U, r, gl="cluster-normalized", kernel="se", epsilon=0.7)
    print("cross_similarity_se maxrel", (np.abs(Zse.data.reshape(n, r) - vo) / np.maximum(np.abs(vo), 1e-300)).max(), flush=True)

    print("== spectrum", flush=True)
    Zg = api.cross_similarity_lae_cpp(X, U, r, "cluster-normalized")
    eo, vo = O.cross_similarity(X, U, r, gl="cluster-normalized")
    for Kt, root in ([(K, True), (-1, False)] if s <= 1000 else [(K, True)]):
        t0 = time.time()
        ep = api.spectrum_from_Z_cpp(Zg, Kt, root)
        t1 = time.time()
        ovals, ovec = O.spectrum_from_Z(eo, vo, s, Kt, root=root)
        Kk = ep.values.size
        print(f"K={Kt} root={root}: gpu {t1-t0:.3f}s values maxrel {np.abs(ep.values - ovals).max() / ovals[0]:.3e}", flush=True)
        idx0 = np.arange(n, dtype=np.int32); idx1 = np.arange(m, dtype=np.int32)
        H = api.HK_from_spectrum_cpp(ep, Kk, 3.0, idx0, idx1)
        Ho = O.hk_from_spectrum(ovals, ovec, Kk, 3.0, idx0, idx1)
        Hself = O.hk_from_spectrum(ep.values, ep.vectors, Kk, 3.0, idx0, idx1)
        print(f"   H vs oracle-pipeline maxabs {np.abs(H - Ho).max():.3e} (|H|max {np.abs(Ho).max():.3e});"
              f" HK kernel vs oracle on same spectrum {np.abs(H - Hself).max():.3e}", flush=True)
        VtV = ep.vectors.T @ ep.vectors / n
        print(f"   |V^T V / n - I| {np.abs(VtV - np.eye(Kk)).max():.3e}", flush=True)

    print("== end to end", flush=True)
    t0 = time.time()
    H = api.heat_kernel_covariance_rcpp(X[:m], X[m:], s, r, 3.0, K=K, U=U)
    t1 = time.time()
    Ho = O.heat_kernel_covariance(X[:m], X[m:], U, r, 3.0, K=K)
    print(f"heat_kernel_covariance_rcpp {t1-t0:.3f}s maxabs {np.abs(H - Ho).max():.3e} rel {np.abs(H - Ho).max() / np.abs(Ho).max():.3e}", flush=True)


if __name__ == "__main__":
    main()
