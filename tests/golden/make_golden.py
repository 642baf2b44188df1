"""Generates the fixtures under tests/golden/.

RESTATEMENT-DERIVED, not reference-derived: the reference (junhuihe2000/FLGP) has no tests or
golden vectors for this path and cannot be built or run in this image (no R / Rcpp / RcppEigen /
RcppParallel / RSpectra), so these vectors come from the two independent restatements in
oracle/flgp_oracle.py -- the C one (flgp_oracle.c) and the numpy one (np_*) -- and are only
written when the two agree.  Inputs are stored next to the expected outputs so that no test
depends on the generator.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from flgp_amd import synth  # noqa: E402
from oracle import flgp_oracle as O  # noqa: E402


def case(name, n, d, s, r, K, t, seed, gl, root, m):
    X = synth.gaussian_mixture(n, d, components=4, seed=seed)
    rows = synth.random_anchor_rows(n, s, seed=seed)
    U0 = synth.anchors_from_rows(X, rows)
    sizes = np.bincount(O.knn(X, U0, 1)[:, 0], minlength=s).astype(float)
    U = np.asfortranarray(np.hstack([U0, sizes[:, None]]))
    # C restatement
    kidx, kdist = O.knn(X, U0, r, output=True)
    ei, ev = O.lae(X, U0, r, knn_idx=kidx)
    zn = O.graph_laplacian(ei, ev, s, gl, sizes)
    vals, vecs = O.spectrum_from_Z(ei, zn, s, K, root=root, method="dense" if K in (-1, s) else "svds")
    Kk = vals.size
    H = O.hk_from_spectrum(vals, vecs, Kk, t, np.arange(n), np.arange(m))
    # numpy restatement must agree before anything is written
    nidx, nd = O.np_knn(X, U0, r)
    assert np.array_equal(nidx, kidx), name
    assert np.allclose(nd, kdist, rtol=0, atol=1e-11), name
    Zd, _ = O.np_lae_dense(X, U0, r)
    assert np.abs(O.ell_to_csr(ei, ev, s).toarray() - Zd).max() < 1e-12, name
    Zn = O.np_graph_laplacian_dense(Zd, gl, sizes)
    assert np.abs(O.ell_to_csr(ei, zn, s).toarray() - Zn).max() < 1e-11, name
    nv, nvec = O.np_spectrum_dense(Zn, Kk, root=root)
    assert np.abs(nv - vals).max() < 1e-10, name
    Hn = O.np_hk(nv, nvec, Kk, t, np.arange(n), np.arange(m))
    assert np.abs(Hn - H).max() < 1e-8 * np.abs(H).max(), name
    np.savez_compressed(os.path.join(HERE, name + ".npz"), X=X, U=U, r=r, K=K, t=t, m=m, gl=gl, root=root,
                        knn_idx=kidx, knn_dist=kdist, ell_idx=ei, lae_val=ev, z_val=zn, values=vals, H=H)
    print("wrote", name, "n", n, "d", d, "s", s, "r", r, "K", Kk)


def main():
    # man-page shapes (man/heat_kernel_covariance_rcpp.Rd:54-59: n=5,d=2,s=2,r=2,K=-1; man/LAE_cpp.Rd:26-31)
    case("manpage_hk", n=5, d=2, s=2, r=2, K=-1, t=1.0, seed=3, gl="cluster-normalized", root=True, m=2)
    case("small_rw", n=60, d=3, s=12, r=3, K=-1, t=2.0, seed=5, gl="rw", root=False, m=10)
    case("small_normalized", n=120, d=2, s=20, r=3, K=6, t=4.0, seed=8, gl="normalized", root=True, m=16)
    case("c1_like", n=480, d=2, s=60, r=3, K=10, t=10.0, seed=1234, gl="cluster-normalized", root=True, m=24)
    case("d16_r10", n=300, d=16, s=48, r=10, K=12, t=3.0, seed=20241022, gl="cluster-normalized", root=True, m=20)
    # hand-derivable known answers
    np.savez(os.path.join(HERE, "known_answers.npz"),
             v_to_z_in=np.array([[0.5, 0.5, 0, 0], [2, 0, 0, 0], [1, 1, 1, 0], [0.9, 0.3, -1, 0]], float),
             v_to_z_len=np.array([2, 2, 3, 3]),
             v_to_z_out=np.array([[0.5, 0.5, 0, 0], [1, 0, 0, 0], [1 / 3, 1 / 3, 1 / 3, 0], [0.8, 0.2, 0, 0]], float))
    print("wrote known_answers")


if __name__ == "__main__":
    main()
