"""CPU: pins the oracle (oracle/) -- known answers, the two restatements against each other,
algebraic invariants and the committed fixtures.  The reference itself has no tests for this
path, so these are the pins the HIP parity tests stand on."""
import glob
import os

import numpy as np
import pytest

from conftest import make_case

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_v_to_z_known_answers(oracle):
    ka = np.load(os.path.join(GOLDEN, "known_answers.npz"))
    for v, ln, z in zip(ka["v_to_z_in"], ka["v_to_z_len"], ka["v_to_z_out"]):
        np.testing.assert_allclose(oracle.v_to_z(v[:ln]), z[:ln], rtol=0, atol=1e-15)
        np.testing.assert_allclose(oracle.np_v_to_z(v[:ln]), z[:ln], rtol=0, atol=1e-15)


def test_v_to_z_is_simplex_projection(oracle):
    rng = np.random.default_rng(0)
    for r in (1, 2, 3, 7, 10, 32):
        for _ in range(20):
            v = rng.normal(size=r) * rng.choice([0.1, 1.0, 10.0])
            z = oracle.v_to_z(v)
            assert (z >= 0).all() and abs(z.sum() - 1.0) < 1e-13
            np.testing.assert_allclose(z, oracle.np_v_to_z(v), rtol=0, atol=1e-14)
            # optimality: z - v is constant on the support, and <= that constant off it
            sup = z > 0
            lam = (z - v)[sup]
            assert np.ptp(lam) < 1e-12
            assert ((z - v)[~sup] >= lam.mean() - 1e-12).all()


def test_knn_integer_lattice_exact(oracle):
    # distinct integer distances: exact in fp64, answer derivable by hand
    X = np.array([[0.0, 0.0], [10.0, 0.0], [3.0, 4.0]])
    U = np.array([[0.0, 1.0], [0.0, 3.0], [6.0, 0.0], [10.0, 2.0]])
    idx, dist = oracle.knn(X, U, 2, output=True)
    np.testing.assert_array_equal(idx, [[0, 1], [3, 2], [1, 0]])
    np.testing.assert_array_equal(dist, [[1.0, 9.0], [4.0, 16.0], [10.0, 18.0]])


def test_knn_ties_lower_index_wins(oracle):
    X = np.zeros((1, 2))
    U = np.array([[1.0, 0.0], [0.0, 1.0], [-1.0, 0.0], [0.0, -1.0], [2.0, 0.0]])
    np.testing.assert_array_equal(oracle.knn(X, U, 3), [[0, 1, 2]])
    np.testing.assert_array_equal(oracle.knn(X, U, 5), [[0, 1, 2, 3, 4]])


@pytest.mark.parametrize("n,d,s,r", [(257, 2, 33, 3), (100, 16, 40, 10), (64, 3, 64, 5), (50, 1, 9, 1)])
def test_knn_c_vs_numpy(oracle, n, d, s, r):
    X, U0, _ = make_case(n, d, s, r, seed=n + d, with_sizes=False)
    idx, dist = oracle.knn(X, U0, r, output=True)
    nidx, nd = oracle.np_knn(X, U0, r)
    np.testing.assert_array_equal(idx, nidx)
    np.testing.assert_allclose(dist, nd, rtol=0, atol=1e-11)
    assert (np.diff(dist, axis=1) >= 0).all()


def test_lae_point_faces_of_the_simplex(oracle):
    U = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    z = oracle.local_anchor_embedding(U[1], U)           # x equal to an anchor
    np.testing.assert_allclose(z, [0, 1, 0], atol=2e-3)
    z = oracle.local_anchor_embedding(0.5 * (U[1] + U[2]), U)  # midpoint of two anchors
    np.testing.assert_allclose(z, [0, 0.5, 0.5], atol=2e-3)
    assert abs(z.sum() - 1) < 1e-14 and (z >= 0).all()


@pytest.mark.parametrize("n,d,s,r", [(80, 2, 12, 3), (60, 16, 24, 10), (40, 3, 9, 5), (30, 2, 6, 1)])
def test_lae_c_vs_numpy(oracle, n, d, s, r):
    X, U0, _ = make_case(n, d, s, r, seed=7 * n + r, with_sizes=False)
    ei, ev, it = oracle.lae(X, U0, r, return_iters=True)
    Zd, _ = oracle.np_lae_dense(X, U0, r)
    assert np.abs(oracle.ell_to_csr(ei, ev, s).toarray() - Zd).max() < 1e-11
    assert (ev >= 0).all() and np.abs(ev.sum(1) - 1).max() < 1e-13
    assert (np.diff(ei, axis=1) > 0).all()               # CSR inner order
    assert it.min() >= 1 and it.max() <= 100


def test_graph_laplacian_by_hand(oracle):
    # 3 x 2 dense Z with r = 2 (every entry stored): all three kinds by hand
    Z = np.array([[0.2, 0.8], [0.5, 0.5], [1.0, 0.0]])
    ei = np.tile(np.array([0, 1], dtype=np.int32), (3, 1))
    sizes = np.array([3.0, 1.0])
    for gl in ("rw", "normalized", "cluster-normalized"):
        W = Z.copy()
        if gl != "rw":
            W = W / (W.sum(0) + 1e-9)
            if gl == "cluster-normalized":
                W = W * sizes
        W = W / (W.sum(1, keepdims=True) + 1e-9)
        got = oracle.graph_laplacian(ei, Z.copy(), 2, gl, sizes)
        np.testing.assert_allclose(got, W, rtol=1e-15, atol=0)
    with pytest.raises(ValueError):
        oracle.graph_laplacian(ei, Z.copy(), 2, "bogus", sizes)


@pytest.mark.parametrize("gl", ["rw", "normalized", "cluster-normalized"])
def test_spectrum_invariants(oracle, gl):
    n, d, s, r, K = 400, 3, 40, 4, 12
    X, U0, U = make_case(n, d, s, r, seed=42)
    ei, zn = oracle.cross_similarity(X, U, r, gl=gl)
    rho = zn.sum(1)
    assert np.abs(rho - 1).max() < 1e-6                   # rows of the normalised Z sum to rho/(rho+1e-9)
    for method in ("svds", "gram", "dense"):
        Kk = s if method == "dense" else K
        vals, vec = oracle.spectrum_from_Z(ei, zn, s, Kk, root=True, method=method)
        assert abs(vals[0] - 1.0) < 1e-6                  # sigma_1 = 1 up to the 1e-9 guards
        assert (np.diff(vals) <= 1e-12).all()
        np.testing.assert_allclose(vec.T @ vec / n, np.eye(Kk), atol=1e-9)   # V^T V = n I
        v0 = vec[:, 0] * np.sign(vec[0, 0])
        np.testing.assert_allclose(v0, 1.0, atol=1e-6)    # top left vector is constant
        H0 = oracle.hk_from_spectrum(vals, vec, Kk, 0.0, np.arange(50), np.arange(50))
        np.testing.assert_allclose(H0, vec[:50] @ vec[:50].T, atol=1e-9)     # H(t=0) = V V^T
        H1 = oracle.hk_from_spectrum(vals, vec, Kk, 2.0, np.arange(50), np.arange(50))
        np.testing.assert_allclose(H1, H1.T, atol=1e-10)
        assert np.linalg.eigvalsh(H1).min() > -1e-9
        assert np.trace(H1) <= np.trace(H0) + 1e-9        # decay in t


def test_spectrum_routes_agree(oracle):
    n, d, s, r, K = 600, 16, 60, 10, 15
    X, U0, U = make_case(n, d, s, r, seed=99)
    ei, zn = oracle.cross_similarity(X, U, r, gl="cluster-normalized")
    ref_vals, ref_vec = oracle.spectrum_from_Z(ei, zn, s, s, root=True, method="dense")
    idx0, idx1 = np.arange(n), np.arange(40)
    Href = oracle.hk_from_spectrum(ref_vals, ref_vec, K, 5.0, idx0, idx1)
    for method in ("svds", "gram"):
        vals, vec = oracle.spectrum_from_Z(ei, zn, s, K, root=True, method=method)
        np.testing.assert_allclose(vals, ref_vals[:K], rtol=1e-10)
        H = oracle.hk_from_spectrum(vals, vec, K, 5.0, idx0, idx1)
        assert np.abs(H - Href).max() <= 1e-9 * np.abs(Href).max()


def test_nystrom_restatement_properties(oracle):
    """np_nystrom_eigenpair (reference src/Fit.cpp:244-289): the leading pair is the trivial one (value 1, constant
    vector), and extending to the anchors themselves returns the anchor eigenvectors -- D^-1 A and D^-1/2 A D^-1/2
    share eigenvalues, and W_XU V / lambda = V for X = U."""
    rng = np.random.default_rng(0)
    s, d, K = 120, 3, 10
    U = rng.normal(size=(s, d)); X = np.vstack([U, rng.normal(size=(300, d))])
    vals, vecs = oracle.np_nystrom_eigenpair(X, U, 1.0, K)
    assert vecs.shape == (420, K) and np.all(np.diff(vals) <= 0)
    assert abs(vals[0] - 1.0) < 1e-6
    assert np.ptp(vecs[:, 0]) < 1e-5 * abs(vecs[0, 0])                 # constant vector, also off the anchors
    np.testing.assert_allclose(np.linalg.norm(vecs[:s], axis=0), np.sqrt(s), rtol=1e-5)     # sqrt(s) column norm (:280)
    D = ((U[:, None, :] - U[None, :, :]) ** 2).sum(-1)
    Z = np.exp(-D / (1.0 * D.sum() / s ** 2)); rs = Z.sum(1); A = Z / rs[:, None] / rs[None, :]
    P = A / A.sum(1)[:, None]                                           # random-walk matrix of the anchor graph
    np.testing.assert_allclose(P @ vecs[:s], vecs[:s] * vals, atol=1e-6)


def test_kmeans_lloyd_restatement(oracle):
    """np_kmeans_lloyd: a fixed point of assign/update -- every centre is the mean of the points nearest to it, sizes
    count them, and well-separated blobs are recovered whatever rows it starts from."""
    rng = np.random.default_rng(3)
    cen = np.array([[0, 0], [9, 0], [0, 9], [9, 9]], dtype=float)
    X = np.vstack([rng.normal(size=(250, 2)) + c for c in cen])
    U, it = oracle.np_kmeans_lloyd(X, np.array([0, 250, 500, 750]), 100)
    assert 1 <= it < 100 and U[:, 2].sum() == 1000
    lab = oracle.knn(X, U[:, :2], 1)[:, 0]
    for c in range(4):
        np.testing.assert_allclose(U[c, :2], X[lab == c].mean(0), rtol=0, atol=1e-12)
        assert U[c, 2] == (lab == c).sum() == 250
    assert np.abs(U[:, :2] - cen).max() < 0.3
    U1, it1 = oracle.np_kmeans_lloyd(X, np.array([0, 1, 2, 3]), 1)        # iter_max cuts the loop
    assert it1 == 1


def test_hk_c_vs_numpy(oracle):
    rng = np.random.default_rng(5)
    n, K = 37, 6
    vec = np.asfortranarray(rng.normal(size=(n, K)))
    vals = np.sort(rng.uniform(0.2, 1.0, K))[::-1].copy()
    idx0 = np.array([3, 0, 36, 7, 7], dtype=np.int32)
    idx1 = np.array([5, 1, 2], dtype=np.int32)
    H = oracle.hk_from_spectrum(vals, vec, K, 1.7, idx0, idx1)
    np.testing.assert_allclose(H, oracle.np_hk(vals, vec, K, 1.7, idx0, idx1), rtol=1e-13, atol=1e-14)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))))
def test_golden_fixtures(oracle, path):
    if os.path.basename(path) == "known_answers.npz":
        pytest.skip("covered by test_v_to_z_known_answers")
    g = np.load(path)
    X, U = g["X"], g["U"]
    d = X.shape[1]; s = U.shape[0]; r = int(g["r"]); K = int(g["K"]); t = float(g["t"]); m = int(g["m"])
    gl = str(g["gl"]); root = bool(g["root"])
    U0 = np.asfortranarray(U[:, :d])
    kidx, kdist = oracle.knn(X, U0, r, output=True)
    np.testing.assert_array_equal(kidx, g["knn_idx"])
    np.testing.assert_array_equal(kdist, g["knn_dist"])
    ei, ev = oracle.lae(X, U0, r, knn_idx=kidx)
    np.testing.assert_array_equal(ei, g["ell_idx"])
    np.testing.assert_array_equal(ev, g["lae_val"])
    zn = oracle.graph_laplacian(ei, ev, s, gl, U[:, d])
    np.testing.assert_array_equal(zn, g["z_val"])
    H = oracle.heat_kernel_covariance(X[:m], X[m:], U, r, t, K=K, gl=gl, root=root)
    assert np.abs(H - g["H"]).max() <= 1e-9 * np.abs(g["H"]).max()


def test_predict_regression_different_reduces_to_same(oracle):
    O = oracle
    """The numpy restatement of predict_regression_cpp's noisepar = "different" branch (src/Predict.cpp:76-110) with equal
    noise variances is the "same" branch (:46-75) -- both sides of m <= K."""
    rng = np.random.default_rng(3)
    n, Kt = 400, 30
    vec = np.asfortranarray(np.linalg.qr(rng.normal(size=(n, Kt)))[0] * np.sqrt(n))
    vals = np.sort(rng.uniform(0.2, 1.0, Kt))[::-1].copy()
    for m, K in [(20, 30), (120, 25)]:
        idx0 = rng.permutation(n)[:m]; idx1 = rng.permutation(n)[:77]
        Y = rng.normal(size=(m, 2))
        same = O.np_predict_regression(vals, vec, Y, idx0, idx1, K, (3.0, 0.3), 1e-3)
        diff = O.np_predict_regression_different(vals, vec, Y, idx0, idx1, K, np.concatenate([[3.0], np.full(m, 0.3)]), 1e-3)
        np.testing.assert_allclose(diff, same, rtol=0, atol=1e-10 * np.abs(same).max())
        # unequal variances: a row with a huge variance is ignored by the fit -- same prediction as without the row
        nz = np.full(m, 0.3); nz[0] = 1e12
        with_row = O.np_predict_regression_different(vals, vec, Y, idx0, idx1, K, np.concatenate([[3.0], nz]), 1e-3)
        without = O.np_predict_regression_different(vals, vec, Y[1:], idx0[1:], idx1, min(K, m - 1) if m <= K else K,
                                                    np.concatenate([[3.0], nz[1:]]), 1e-3) if m > K else None
        if without is not None:
            np.testing.assert_allclose(with_row, without, rtol=0, atol=1e-6 * np.abs(without).max())


def test_minibatch_kmeans_restatement(oracle):
    """np_kmeans_minibatch (the algorithm behind src/Utils.cpp:49-62, ClusterR's parameters): deterministic in its seed,
    sizes are the 1-NN counts and add up to n, k-means++ centres are rows of X that the updates have moved at most a little,
    and the result is better than random rows."""
    from flgp_amd import synth
    n, d, s = 1500, 3, 25
    X = synth.gaussian_mixture(n, d, components=5, seed=2)
    U, info, wss = oracle.np_kmeans_minibatch(X, s, seed=5)
    U2, _, wss2 = oracle.np_kmeans_minibatch(X, s, seed=5)
    np.testing.assert_array_equal(U, U2)
    assert U.shape == (s, d + 1) and U[:, d].sum() == n and 1 <= info[0] <= 100
    C = np.asfortranarray(U[:, :d])
    lab = oracle.knn(X, C, 1)[:, 0]
    np.testing.assert_array_equal(U[:, d], np.bincount(lab, minlength=s))
    rows = np.sort(synth.random_anchor_rows(n, s, seed=5))
    R = np.asfortranarray(X[rows])
    wss_rand = float(((X - R[oracle.knn(X, R, 1)[:, 0]]) ** 2).sum())
    assert wss < wss_rand
    U3, _, _ = oracle.np_kmeans_minibatch(X, s, seed=6)
    assert not np.array_equal(U3, U)
