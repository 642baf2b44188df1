"""GPU (-m gpu): the parity tests proper.  Everything goes through the C ABI of libflgp_hip.so
(host entry points via flgp_amd.api, device entry points via flgp_amd.pipeline / ctypes) and is
checked against the CPU oracle on the same seeded inputs, against the committed fixtures, and --
at BASELINE.json's full size -- through size-independent properties.

Bars (north_star): neighbour indices bit-exact; LAE weights, Laplacian scalings, column sums and
the Gram matrix bit-exact as well (same operation order as the oracle); exp()-dependent values to
4 ulp; eigenvalues to 1e-10 relative and covariance entries to 1e-8 relative to max|H| (the
stated tolerance: "eigenpairs to 1e-8 rel")."""
import ctypes
import glob
import os

import numpy as np
import pytest
import torch

from conftest import make_case
from flgp_amd import _lib, api, synth
from flgp_amd.pipeline import HeatKernelPath, HipStages, PathConfig

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
H_RTOL = 1e-8          # covariance entries, relative to max |H|
EIG_RTOL = 1e-10       # eigenvalues, relative


@pytest.fixture(scope="module")
def stages():
    assert torch.cuda.is_available(), "the -m gpu tests need an MI355X"
    return HipStages("cuda:0")


def cm(a, dev="cuda:0", dtype=torch.float64):
    """(n x k) array -> column-major device tensor of shape (k, n)."""
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a).T)).to(dtype).to(dev)


def to_np_cm(t):
    return np.asfortranarray(t.cpu().numpy().T)


# ------------------------------------------------------------------------------ k-NN (k1+k2)
@pytest.mark.parametrize("n,d,s,r", [
    (1000, 2, 150, 3), (777, 16, 333, 10), (513, 3, 128, 1), (300, 64, 140, 5), (200, 17, 50, 20),
    (100, 1, 10, 10), (5, 2, 2, 2), (4097, 16, 1025, 16), (256, 8, 200, 32),
    # d > 64: dot products by the GEMM, one lane per point selects (csrc/knn_wide.hip)
    (600, 65, 130, 5), (500, 100, 257, 10), (300, 784, 200, 3), (1000, 130, 300, 1), (257, 200, 64, 32), (3, 70, 2, 2)])
def test_knn_bit_exact(oracle, n, d, s, r):
    X, U0, _ = make_case(n, d, s, r, seed=1000 + n + d, with_sizes=False)
    res = api.KNN_cpp(X, U0, r, output=True)
    oi, od = oracle.knn(X, U0, r, output=True)
    np.testing.assert_array_equal(res["ind_knn"], oi)
    order = np.argsort(oi, axis=1, kind="stable")          # distances_sp is CSR: columns ascending per row
    sp = res["distances_sp"]
    np.testing.assert_array_equal(sp.indices.reshape(n, r), np.take_along_axis(oi, order, axis=1))
    np.testing.assert_array_equal(sp.data.reshape(n, r), np.take_along_axis(od, order, axis=1))
    np.testing.assert_array_equal(api.KNN_cpp(X, U0, r)["ind_knn"], oi)   # output=FALSE path


@pytest.mark.parametrize("mfma", [0, 1])
def test_knn_both_kernels_bit_exact(oracle, mfma):
    """The VALU kernel and the matrix-core kernel (v_mfma_f64_16x16x4 = a k-ascending FMA chain) are both
    the oracle's arithmetic: forced one after the other over every padded dimension, ragged sizes and ties."""
    from flgp_amd import _lib
    L = _lib.lib()
    try:
        L.flgp_set_tuning(b"knn_mfma", mfma)
        for n, d, s, r in [(777, 16, 333, 10), (300, 64, 140, 5), (200, 17, 50, 20), (513, 3, 128, 1), (100, 1, 10, 10),
                           (4097, 7, 1025, 16), (130, 33, 70, 32)]:
            X, U0, _ = make_case(n, d, s, r, seed=7 * n + d, with_sizes=False)
            res = api.KNN_cpp(X, U0, r, output=True)
            oi, od = oracle.knn(X, U0, r, output=True)
            np.testing.assert_array_equal(res["ind_knn"], oi)
            order = np.argsort(oi, axis=1, kind="stable")
            np.testing.assert_array_equal(res["distances_sp"].data.reshape(n, r), np.take_along_axis(od, order, axis=1))
        g = np.linspace(-1.0, 1.0, 5)     # exact ties: lower anchor index first
        U = np.array([[a, b] for a in g for b in g] + [[0.0, 0.0], [1.0, 0.0]])
        X = np.array([[0.0, 0.0], [0.5, 0.5], [3.0, -3.0], [-1.0, 2.0]])
        for r in (1, 4, 9, 16):
            np.testing.assert_array_equal(api.KNN_cpp(X, U, r)["ind_knn"], oracle.knn(X, U, r))
    finally:
        L.flgp_set_tuning(b"knn_mfma", -1)


@pytest.mark.parametrize("variant", [0, 6])
def test_knn_one_neighbour_kernels_bit_exact(oracle, variant):
    """r = 1 has its own kernels (running minima in the MFMA result layout, merged across 16 lanes at the end; or the
    VALU kernel without its queue): indices AND distances equal the oracle's over every padded dimension, ragged sizes,
    and exact ties (lower anchor index)."""
    from flgp_amd import _lib
    L = _lib.lib()
    try:
        L.flgp_set_tuning(b"knn_nn1_variant", variant)
        for n, d, s in [(1000, 1, 130), (777, 3, 64), (513, 7, 129), (2049, 16, 1000), (300, 33, 70), (260, 64, 140), (3, 5, 3)]:
            X, U0, _ = make_case(n, d, s, 1, seed=11 * n + d, with_sizes=False)
            res = api.KNN_cpp(X, U0, 1, output=True)
            oi, od = oracle.knn(X, U0, 1, output=True)
            np.testing.assert_array_equal(res["ind_knn"], oi)
            np.testing.assert_array_equal(res["distances_sp"].data.reshape(n, 1), od)
        g = np.linspace(-1.0, 1.0, 5)
        U = np.array([[a, b] for a in g for b in g] * 3 + [[0.0, 0.0]])         # every lattice point three times
        X = np.array([[0.0, 0.0], [0.5, 0.5], [3.0, -3.0], [-1.0, 2.0], [0.25, -0.75]])
        np.testing.assert_array_equal(api.KNN_cpp(X, U, 1)["ind_knn"], oracle.knn(X, U, 1))
    finally:
        L.flgp_set_tuning(b"knn_nn1_variant", 0)


def _knn_equal(oracle, X, U, r):
    res = api.KNN_cpp(X, U, r, output=True)
    oi, od = oracle.knn(X, U, r, output=True)
    np.testing.assert_array_equal(res["ind_knn"], oi)
    order = np.argsort(oi, axis=1, kind="stable")
    np.testing.assert_array_equal(res["distances_sp"].data.reshape(X.shape[0], r), np.take_along_axis(od, order, axis=1))


@pytest.mark.parametrize("n,d,s,r,seed", [(3000, 16, 700, 10, 0), (2500, 5, 512, 4, 1), (2999, 8, 2000, 16, 2),
                                           (5700, 12, 5000, 7, 3), (1100, 9, 640, 2, 4), (1257, 16, 1024, 13, 5),
                                           (3000, 3, 700, 3, 6), (2000, 2, 1024, 5, 7), (1500, 4, 600, 16, 8), (900, 1, 513, 2, 9)])
def test_knn_screened_kernel_random(oracle, n, d, s, r, seed):
    """d <= 16, 2 <= r <= 16, s >= 512 runs the kernel that screens anchors on the matrix cores in bf16 pieces and
    evaluates the exact chain only for what is left: indices and distances stay the oracle's, bit for bit."""
    X, U0, _ = make_case(n, d, s, r, seed=4242 + seed, with_sizes=False)
    _knn_equal(oracle, X, U0, r)


def test_knn_screened_kernel_where_the_screen_has_no_say(oracle):
    """Everything the screen cannot decide goes to exact arithmetic alone: far-off data (the error budget is relative
    to |x|^2 + |u|^2, so nearly every anchor stays a candidate), tiny and huge scales, points at the origin, huge
    anchors, more duplicates of one anchor than a candidate queue holds, lattices of exact ties."""
    rng = np.random.default_rng(99)
    n, d, s, r = 1500, 16, 900, 10
    X = rng.normal(size=(n, d)); U = X[rng.choice(n, s, replace=False)] + 0.05 * rng.normal(size=(s, d))
    _knn_equal(oracle, X + 1.0e4, U + 1.0e4, r)                 # offset: queues overflow, whole-wave scans
    _knn_equal(oracle, X * 1e-12, U * 1e-12, r)
    _knn_equal(oracle, X * 1e-17, U * 1e-17, r)                 # |x|^2 below 1e-30
    _knn_equal(oracle, X * 1e9, U * 1e9, r)
    _knn_equal(oracle, X * 1e17, U * 1e17, r)                   # |x|^2 above 1e30
    Xs = X.copy(); Us = U.copy()
    Xs[::7] = 0.0; Xs[3::11] *= 1e-20; Xs[5::13] *= 1e25; Xs[11] = 1e140
    Us[::9] *= 1e22; Us[4] = 0.0; Us[17] = 1e140
    _knn_equal(oracle, Xs, Us, r)
    Ud = U.copy(); Ud[100:260] = Ud[100]; Ud[300:340] = Ud[7]    # 160 and 40 copies of one anchor
    _knn_equal(oracle, X, Ud, r)
    _knn_equal(oracle, np.vstack([X, Ud[100:101], Ud[7:8]]), Ud, 16)
    g = np.arange(-3, 4, dtype=float)                            # 7^3 = 343 lattice points twice: exact ties everywhere
    L3 = np.array([[a, b, c, 0, 0] for a in g for b in g for c in g])
    Ul = np.vstack([L3, L3])
    Xl = np.vstack([L3[::5] + 0.5, L3[::7], rng.integers(-3, 4, size=(300, 5)) * 0.5])
    for rr in (3, 8, 16):
        _knn_equal(oracle, Xl, Ul, rr)


def test_knn_screened_kernel_is_the_other_kernels_result_at_size(oracle):
    """n = 2e5 points against s = 5000 anchors (the anchor count of BASELINE configs[2]): screened and unscreened
    kernels return the same bits."""
    from flgp_amd import _lib
    L = _lib.lib()
    n, d, s, r = 200000, 16, 5000, 10
    X, U0, _ = make_case(n, d, s, r, seed=8, with_sizes=False)
    a = api.KNN_cpp(X, U0, r, output=True)
    try:
        L.flgp_set_tuning(b"knn_screen", 0)
        b = api.KNN_cpp(X, U0, r, output=True)
    finally:
        L.flgp_set_tuning(b"knn_screen", 1)
    np.testing.assert_array_equal(a["ind_knn"], b["ind_knn"])
    np.testing.assert_array_equal(a["distances_sp"].data, b["distances_sp"].data)
    np.testing.assert_array_equal(a["distances_sp"].indices, b["distances_sp"].indices)


def test_knn_single_point(oracle):
    rng = np.random.default_rng(4)
    X = rng.normal(size=(1, 16)); U = rng.normal(size=(300, 16))
    np.testing.assert_array_equal(api.KNN_cpp(X, U, 10)["ind_knn"], oracle.knn(X, U, 10))


def test_knn_ties_and_duplicates(oracle):
    # integer lattice with many exactly equal distances and duplicated anchors: lower index wins
    g = np.arange(-3, 4, dtype=float)
    U = np.array([[a, b] for a in g for b in g] + [[0.0, 0.0], [1.0, 0.0]])
    X = np.array([[0.0, 0.0], [0.5, 0.5], [3.0, -3.0], [-1.0, 2.0]])
    for r in (1, 4, 9, 16):
        np.testing.assert_array_equal(api.KNN_cpp(X, U, r)["ind_knn"], oracle.knn(X, U, r))


def test_knn_wide_in_several_row_blocks(oracle):
    """d > 64 walks the points in row blocks (sized for a 1 GB block of dot products): forced small here, with a ragged
    last block."""
    n, d, s, r = 1500, 96, 333, 7
    X, U0, _ = make_case(n, d, s, r, seed=77, with_sizes=False)
    L = _lib.lib()
    L.flgp_set_tuning(b"knn_wide_block", 512)
    try:
        res = api.KNN_cpp(X, U0, r, output=True)
    finally:
        L.flgp_set_tuning(b"knn_wide_block", 0)
    oi, od = oracle.knn(X, U0, r, output=True)
    np.testing.assert_array_equal(res["ind_knn"], oi)
    order = np.argsort(oi, axis=1, kind="stable")
    np.testing.assert_array_equal(res["distances_sp"].data.reshape(n, r), np.take_along_axis(od, order, axis=1))


def test_knn_errors():
    X = np.zeros((4, 2)); U = np.zeros((3, 2))
    with pytest.raises(api.FlgpError) as e:
        api.KNN_cpp(X, U, 4)          # r > s: undefined behaviour in the reference, an error here
    assert e.value.code == -1
    with pytest.raises(api.FlgpError) as e:
        api.KNN_cpp(np.zeros((4, 16385)), np.zeros((3, 16385)), 2)
    assert e.value.code == -1 and "<= 16384" in e.value.message


def test_bad_inputs_are_refused_not_faulted(oracle):
    """Round-1 advisor findings: a NaN / Inf coordinate (R's NA_real_ is a NaN) left the k-NN list's 0x7fffffff
    sentinel in ind_knn, which LAE / CSC / Gram then used as an address; user-supplied CSR column indices reached the
    same kernels unchecked; a zero singular value put Inf / NaN into every entry of H.  All three are errors now."""
    import scipy.sparse as sp
    X, U0, U = make_case(300, 3, 40, 4, seed=11)
    for bad in (np.nan, np.inf, -np.inf):
        Xb = X.copy(); Xb[17, 1] = bad
        with pytest.raises(api.FlgpError) as e:
            api.KNN_cpp(Xb, U0, 4)
        assert e.value.code == -1 and "NaN" in e.value.message
        with pytest.raises(api.FlgpError):
            api.heat_kernel_covariance_rcpp(Xb[:20], Xb[20:], 40, 4, 1.0, K=10, U=U)
    Ub = U0.copy(); Ub[3, 0] = np.nan
    with pytest.raises(api.FlgpError):
        api.LAE_cpp(X, Ub, 4)
    # the device entry point cannot refuse (asynchronous): it must at least hand out valid indices
    st = HipStages("cuda:0")
    Xb = X.copy(); Xb[5, :] = np.nan
    idx, _ = st.knn(cm(Xb), st.anchor_prep(cm(U0)), 4)
    idx = idx.cpu().numpy()
    assert idx.min() >= 0 and idx.max() < 40
    np.testing.assert_array_equal(np.delete(idx, 5, axis=1), np.delete(oracle.knn(X, U0, 4).T, 5, axis=1))
    # CSR column index out of range
    Z = api.LAE_cpp(X, U0, 4)
    Zb = sp.csr_matrix((Z.data.copy(), Z.indices.copy(), Z.indptr.copy()), shape=Z.shape)
    Zb.indices[123] = 40
    with pytest.raises(api.FlgpError) as e:
        api.graphLaplacian_cpp(Zb, "normalized")
    assert "outside" in e.value.message
    Zb.indices[123] = -1
    with pytest.raises(api.FlgpError):
        api.spectrum_from_Z_cpp(Zb, 5)
    # K == s with an anchor column that is exactly zero: sigma_s = 0
    Zz = sp.csr_matrix((Z.data.copy(), Z.indices.copy(), Z.indptr.copy()), shape=Z.shape)
    Zz.data[Zz.indices == 7] = 0.0
    with pytest.raises(api.FlgpError) as e:
        api.spectrum_from_Z_cpp(Zz, -1)
    assert e.value.code == -5 and "null space" in e.value.message
    ep = api.spectrum_from_Z_cpp(Zz, 10)                  # a smaller K is fine
    assert np.isfinite(ep.vectors).all()


# ------------------------------------------------------------------------------ LAE (k3+k4)
def test_v_to_z(oracle):
    ka = np.load(os.path.join(GOLDEN, "known_answers.npz"))
    for v, ln, z in zip(ka["v_to_z_in"], ka["v_to_z_len"], ka["v_to_z_out"]):
        np.testing.assert_allclose(api.v_to_z_cpp(v[:ln]).ravel(), z[:ln], rtol=0, atol=1e-15)
    rng = np.random.default_rng(3)
    for r in (1, 2, 5, 10, 17, 32):
        v = rng.normal(size=r) * 3
        np.testing.assert_array_equal(api.v_to_z_cpp(v).ravel(), oracle.v_to_z(v))


@pytest.mark.parametrize("r,d", [(1, 2), (2, 2), (3, 2), (5, 3), (8, 16), (10, 16), (12, 4), (16, 3), (20, 2), (10, 64),
                                 (10, 12), (10, 17), (10, 32), (10, 40), (7, 16), (9, 33), (13, 8), (16, 16), (14, 30),
                                 # few anchors over several lanes per point: the shapes a missing fence broke (sweep_lae.py)
                                 (2, 40), (3, 17), (4, 40), (5, 32), (5, 64), (6, 24), (8, 48), (2, 64), (11, 20), (16, 32),
                                 # d > 64: the kernel that reads the anchors from the panel in memory
                                 (3, 100), (10, 65), (5, 784), (12, 130), (20, 72)])
def test_lae_bit_exact(oracle, r, d):
    n, s = 500, 64
    X, U0, _ = make_case(n, d, s, r, seed=31 * r + d, with_sizes=False)
    Z = api.LAE_cpp(X, U0, r)
    ei, ev = oracle.lae(X, U0, r)
    np.testing.assert_array_equal(Z.indptr, np.arange(0, n * r + 1, r))
    np.testing.assert_array_equal(Z.indices.reshape(n, r), ei)
    np.testing.assert_array_equal(Z.data.reshape(n, r), ev)      # bit for bit
    assert (ev >= 0).all() and np.abs(ev.sum(1) - 1).max() < 1e-13
    # the same through the two-pass form of the register kernels (iteration budget + compaction of the unfinished points,
    # csrc/lae_reg.h): forced on at this size, with cuts that park almost every point, about half of them, and hardly any
    L = _lib.lib()
    L.flgp_set_tuning(b"lae_cut_min_n", 0)
    try:
        for cut in (1, 7, 40):
            L.flgp_set_tuning(b"lae_cut", cut)
            Z2 = api.LAE_cpp(X, U0, r)
            np.testing.assert_array_equal(Z2.indices.reshape(n, r), ei)
            np.testing.assert_array_equal(Z2.data.reshape(n, r), ev)
    finally:
        L.flgp_set_tuning(b"lae_cut_min_n", 32768)
        L.flgp_set_tuning(b"lae_cut", 13)


def test_lae_huge_coordinates(oracle):
    # coordinates ~1e140: the simplex projection sees entries beyond 2^900 and must take the IEEE-division
    # branch (lae_dev.h) -- still bit for bit
    n, d, s, r = 300, 16, 40, 10
    X, U0, _ = make_case(n, d, s, r, seed=77, with_sizes=False)
    X = X * 1e140; U0 = U0 * 1e140
    Z = api.LAE_cpp(X, U0, r)
    ei, ev = oracle.lae(X, U0, r)
    assert np.isfinite(ev).all()
    np.testing.assert_array_equal(Z.indices.reshape(n, r), ei)
    np.testing.assert_array_equal(Z.data.reshape(n, r), ev)


def test_lae_slow_converging_points(oracle):
    # d = 2, r = 3 is the regime where a fraction of the points needs many iterations
    # (mean 23, p99 81, some hit the T = 100 cap)
    X, _ = synth.torus(4800)
    U0 = synth.anchors_from_rows(X, synth.random_anchor_rows(4800, 600))
    ei, ev, it = oracle.lae(X, U0, 3, return_iters=True)
    assert it.max() >= 50
    Z = api.LAE_cpp(X, U0, 3)
    np.testing.assert_array_equal(Z.data.reshape(-1, 3), ev)


def test_local_anchor_embedding_point(oracle):
    rng = np.random.default_rng(9)
    for r, d in [(3, 2), (10, 16), (5, 7)]:
        U = rng.normal(size=(r, d)); x = U.mean(0) + 0.1 * rng.normal(size=d)
        np.testing.assert_array_equal(api.local_anchor_embedding_cpp(x, U).ravel(), oracle.local_anchor_embedding(x, U))


# ------------------------------------------------------------------------------ Laplacian (k5)
@pytest.mark.parametrize("gl", ["rw", "normalized", "cluster-normalized"])
@pytest.mark.parametrize("n,d,s,r", [(900, 3, 80, 5), (2500, 16, 300, 10)])
def test_cross_similarity_lae_bit_exact(oracle, gl, n, d, s, r):
    X, U0, U = make_case(n, d, s, r, seed=n + s)
    Z = api.cross_similarity_lae_cpp(X, U, r, gl)
    ei, zn = oracle.cross_similarity(X, U, r, gl=gl)
    np.testing.assert_array_equal(Z.indices.reshape(n, r), ei)
    np.testing.assert_array_equal(Z.data.reshape(n, r), zn)
    # graphLaplacian_cpp on its own, fed with the un-normalised LAE matrix
    Zl = api.LAE_cpp(X, U0, r)
    np.testing.assert_array_equal(api.graphLaplacian_cpp(Zl, gl, U[:, d]).data.reshape(n, r), zn)


def test_cross_similarity_se(oracle):
    n, d, s, r = 1200, 3, 90, 6
    X, U0, U = make_case(n, d, s, r, seed=5)
    Z = api.cross_similarity_se_cpp(X, U, r, "cluster-normalized", 0.5)
    ei, zn = oracle.cross_similarity(X, U, r, gl="cluster-normalized", kernel="se", epsilon=0.5)
    np.testing.assert_array_equal(Z.indices.reshape(n, r), ei)
    np.testing.assert_allclose(Z.data.reshape(n, r), zn, rtol=1e-14, atol=0)   # exp() differs by a few ulp


@pytest.mark.parametrize("seed", range(16))
def test_random_shapes_whole_path(oracle, seed):
    """One random configuration per seed (dimension, anchors, neighbours, rows, K, kernel, Laplacian, root, t) through
    heat_kernel_covariance_cpp against the oracle -- the shapes nobody thought of listing (scripts/stress_parity.py is the
    long version; it found the r <= 5, d > 16 LAE bug).  Degenerate inputs are left out: r = 1 (G is the identity), an SE
    bandwidth far below the neighbour distances (Z underflows), K = s with few rows per anchor (singular values at
    rounding level: their left vectors are arbitrary in the reference too)."""
    rng = np.random.default_rng(1000 + seed)
    d = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 17, 32, 40, 64]))
    s = int(rng.integers(12, 200))
    r = int(rng.integers(2, min(s, 24) + 1))
    n = int(rng.integers(max(4 * s, 50), 3000))
    m = int(rng.integers(1, min(n, 200) + 1))
    K = int(rng.integers(1, min(s, 40) + 1)) if rng.random() < 0.85 else -1
    kernel = str(rng.choice(["lae", "se"])); gl = str(rng.choice(["rw", "normalized", "cluster-normalized"]))
    root = bool(rng.integers(0, 2)); t = float(rng.choice([0.1, 1.0, 10.0]))
    eps = float(rng.choice([0.5, 1.0, 3.0])) * np.sqrt(d)
    X = rng.normal(size=(n, d)) + 3.0 * rng.integers(0, 3, size=(n, 1))
    U0 = X[np.sort(rng.choice(n, size=s, replace=False))] + 1e-3 * rng.normal(size=(s, d))
    lab = oracle.knn(X, U0, 1)[:, 0]
    U = np.asfortranarray(np.hstack([U0, np.bincount(lab, minlength=s)[:, None].astype(float)]))
    H = api.heat_kernel_covariance_cpp(X[:m], X[m:], s, r, t, K, dict(kernel=kernel, gl=gl, root=root), 1, eps, U=U)
    Ho = oracle.heat_kernel_covariance(X[:m], X[m:], U, r, t, K=K, kernel=kernel, gl=gl, root=root, epsilon=eps)
    err = np.abs(H - Ho).max() / np.abs(Ho).max()
    if err >= 1e-7 and 0 < K < s:     # a cut through a cluster of eigenvalues is ill-posed: accept only with the gap shown
        v, _ = oracle.heat_kernel_spectrum(np.asfortranarray(X), U, r, K + 1, kernel, gl, root, eps, "auto")
        assert abs(v[K - 1] - v[K]) < 1e-6 * v[0], (err, v[K - 2:K + 1])
    else:
        assert err < 1e-7, (err, dict(n=n, d=d, s=s, r=r, m=m, K=K, kernel=kernel, gl=gl, root=root, t=t))


def test_run_to_run_bits(oracle):
    """the same call twice gives the same bits: fixed-order reductions everywhere, including the solver's second
    stream (scripts/check_determinism.py is the large version)"""
    X, U0, U = make_case(20000, 8, 1600, 6, seed=12)          # s >= 1536: block-sparse products, overlapped refinement
    models = dict(kernel="lae", gl="cluster-normalized", root=True)
    H = [api.heat_kernel_covariance_cpp(X[:100], X[100:], 1600, 6, 2.0, 60, models, 1, 0.1, U=U) for _ in range(2)]
    np.testing.assert_array_equal(H[0], H[1])
    ny = [api.nystrom_eigenpair_cpp(X[:3000], U0[:300], 1.0, 20) for _ in range(2)]
    np.testing.assert_array_equal(ny[0].vectors, ny[1].vectors)


def test_cluster_normalized_needs_sizes():
    X, U0, U = make_case(100, 2, 10, 3, seed=1)
    with pytest.raises(api.FlgpError) as e:
        api.cross_similarity_lae_cpp(X, U0, 3, "cluster-normalized")   # the reference reads out of bounds here
    assert e.value.code == -1


# ------------------------------------------------------------------------------ device stages
@pytest.mark.parametrize("n,d,s,r", [(3000, 3, 257, 7), (700, 2, 40, 3), (5000, 4, 65, 20), (2049, 3, 300, 1)])
def test_csc_colsum_gram_bit_exact(oracle, stages, n, d, s, r):
    """n > 1024 crosses the chunk boundary of the two-level column-sum order (oracle/flgp_oracle.c); few anchors and
    r = 20 make lanes of one step collide on a column (the ballot-ranked passes of colsum_chunk_kernel)."""
    X, U0, U = make_case(n, d, s, r, seed=77)
    ei, zn = oracle.cross_similarity(X, U, r, gl="normalized")
    d_ei = torch.from_numpy(ei).cuda(); d_ev = torch.from_numpy(zn).cuda()
    csc = stages.csc(d_ei, s)
    colptr = csc["colptr"].cpu().numpy(); pos = csc["pos"].cpu().numpy()
    counts = np.bincount(ei.ravel(), minlength=s)
    np.testing.assert_array_equal(np.diff(colptr), counts)
    for j in (0, 1, s // 2, s - 1):
        seg = pos[colptr[j]:colptr[j + 1]]
        assert (np.diff(seg) > 0).all() and (ei.ravel()[seg] == j).all()   # stable: rows ascending
    np.testing.assert_array_equal(stages.colsum(d_ei, d_ev, s).cpu().numpy(), oracle.colsum(ei, zn, s))
    av, _ = oracle.scale_A(ei, zn, s)
    c = stages.colsum(d_ei, d_ev, s)
    stages.col_scale(d_ei, d_ev, c, None, 1)
    np.testing.assert_array_equal(d_ev.cpu().numpy(), av)
    G = stages.gram(d_ei, d_ev, csc).cpu().numpy()
    np.testing.assert_array_equal(G, oracle.gram(ei, av, s))
    np.testing.assert_array_equal(G, G.T)


@pytest.mark.parametrize("n,d,s,r,window", [(3000, 3, 257, 7, 0), (3000, 3, 257, 7, 100), (5000, 4, 65, 20, 16), (2049, 3, 300, 1, 299), (2500, 2, 130, 18, 64)])
def test_colsum_and_gram_by_column_windows(oracle, stages, n, d, s, r, window):
    """Beyond s = 20000 the per-column tables of the column-sum and Gram kernels no longer fit LDS and are filled window by
    window (16384 columns at a time).  Forced on small inputs here: the same bits as the oracle, window boundaries inside
    rows and at a ragged last window."""
    X, U0, U = make_case(n, d, s, r, seed=78)
    ei, zn = oracle.cross_similarity(X, U, r, gl="normalized")
    d_ei = torch.from_numpy(ei).cuda(); d_ev = torch.from_numpy(zn).cuda()
    csc = stages.csc(d_ei, s)
    stages.L.flgp_set_tuning(b"sparse_window", window)
    try:
        np.testing.assert_array_equal(stages.colsum(d_ei, d_ev, s).cpu().numpy(), oracle.colsum(ei, zn, s))
        av, _ = oracle.scale_A(ei, zn, s)
        c = stages.colsum(d_ei, d_ev, s)
        stages.col_scale(d_ei, d_ev, c, None, 1)
        np.testing.assert_array_equal(d_ev.cpu().numpy(), av)
        G = stages.gram(d_ei, d_ev, csc).cpu().numpy()
    finally:
        stages.L.flgp_set_tuning(b"sparse_window", 0)
    np.testing.assert_array_equal(G, oracle.gram(ei, av, s))


def test_more_than_20000_anchors(oracle, stages):
    """s = 20500 (rounds 1-3 refused s > 20000): column sums and the Gram matrix bit for bit the oracle's, and the path's
    eigenpairs satisfy G v = lambda v on the device's own Gram matrix."""
    n, d, s, r, K = 62000, 3, 20500, 3, 24
    X, U0, U = make_case(n, d, s, r, seed=5)
    ei, zn = oracle.cross_similarity(X, U, r, gl="normalized")
    d_ei = torch.from_numpy(ei).cuda(); d_ev = torch.from_numpy(zn).cuda()
    csc = stages.csc(d_ei, s)
    np.testing.assert_array_equal(stages.colsum(d_ei, d_ev, s).cpu().numpy(), oracle.colsum(ei, zn, s))
    av, _ = oracle.scale_A(ei, zn, s)
    c = stages.colsum(d_ei, d_ev, s)
    stages.col_scale(d_ei, d_ev, c, None, 1)
    np.testing.assert_array_equal(d_ev.cpu().numpy(), av)
    G = stages.gram(d_ei, d_ev, csc)
    Go = oracle.gram(ei, av, s)
    assert torch.equal(G.cpu(), torch.from_numpy(Go))
    del Go
    eig, V, info = stages.eig_topk(G, K)
    R = G @ V.T - V.T * eig            # (the (K, s) tensor V is the column-major s x K block; G is symmetric)
    assert float(R.abs().max()) <= 1e-9 * float(eig[0])
    assert float((V @ V.T - torch.eye(K, dtype=torch.float64, device="cuda")).abs().max()) <= 1e-10
    Z = api.cross_similarity_lae_cpp(X, U, r, "normalized")
    ep = api.spectrum_from_Z_cpp(Z, K, True)
    np.testing.assert_allclose(ep.values, np.sqrt(eig.cpu().numpy()), rtol=1e-9)


@pytest.mark.parametrize("s", [1, 31, 32, 257, 1000])
def test_sym_pack_unpack(stages, s):
    """exchange 3 sends the upper triangle of the Gram partials: pack / unpack are exact copies, and the unpacked
    matrix is symmetric bit for bit whatever was below the diagonal before"""
    rng = np.random.default_rng(s)
    A = rng.normal(size=(s, s))                         # not symmetric: only the upper triangle may be read
    G = torch.from_numpy(np.ascontiguousarray(A.T)).cuda()          # (s, s) tensor == column-major A
    p = stages.sym_pack(G).cpu().numpy()
    np.testing.assert_array_equal(p, np.concatenate([A[:j + 1, j] for j in range(s)]))
    out = stages.sym_unpack(torch.from_numpy(p).cuda(), torch.full((s, s), np.nan, dtype=torch.float64, device="cuda")).cpu().numpy().T
    ref = np.triu(A) + np.triu(A, 1).T
    np.testing.assert_array_equal(out, ref)


@pytest.mark.parametrize("M,N,Kd", [(128, 128, 16), (130, 257, 33), (1, 1, 1), (500, 40, 200), (64, 64, 5000), (300, 300, 7)])
def test_gemm_f64(stages, M, N, Kd):
    rng = np.random.default_rng(M + N + Kd)
    A = rng.normal(size=(M, Kd)); B = rng.normal(size=(Kd, N)); E = rng.normal(size=(M, N))
    ref = 0.75 * (A @ B) - 1.25 * E
    L = stages.L
    st = torch.cuda.current_stream().cuda_stream
    work = torch.empty(1 << 22, dtype=torch.float64, device="cuda")
    def dev(a, colmajor):   # bytes of a column-major (or row-major) copy of `a` on the device
        return torch.from_numpy(np.ascontiguousarray(a.T if colmajor else a)).cuda()

    for a_cm in (True, False):
        for b_cm in (True, False):
            for c_cm in (True, False):
                dA, dB, dE = dev(A, a_cm), dev(B, b_cm), dev(E, c_cm)
                dC = torch.empty_like(dE)
                a_s = (1, M) if a_cm else (Kd, 1)      # (i stride, k stride)
                b_s = (1, Kd) if b_cm else (N, 1)      # (k stride, j stride)
                c_s = (1, M) if c_cm else (N, 1)       # (i stride, j stride)
                for use_work in (False, True):
                    _lib.check(L.flgp_dev_gemm(st, M, N, Kd, 0.75, dA.data_ptr(), a_s[0], a_s[1], dB.data_ptr(), b_s[0], b_s[1],
                                               -1.25, dE.data_ptr(), c_s[0], c_s[1], dC.data_ptr(), c_s[0], c_s[1],
                                               work.data_ptr() if use_work else None, work.numel() if use_work else 0))
                    got = dC.cpu().numpy()
                    got = got.T if c_cm else got
                    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-11 * max(1.0, np.sqrt(Kd)))


@pytest.mark.parametrize("M,N,Kd", [(5000, 256, 256), (333, 64, 40), (4100, 192, 192)])
def test_gemm_pair_is_two_single_products(stages, M, N, Kd):
    """Two products in one launch (the eigensolver's pair of rotations) are bit for bit the two single launches."""
    rng = np.random.default_rng(M + Kd)
    L = stages.L
    st = torch.cuda.current_stream().cuda_stream
    A1, A2 = (torch.from_numpy(rng.normal(size=(Kd, M))).cuda() for _ in range(2))       # column-major M x Kd
    B1 = torch.from_numpy(rng.normal(size=(N, Kd))).cuda()                                  # column-major Kd x N
    B2 = torch.from_numpy(rng.normal(size=(N, Kd))).cuda()
    for Bsecond in (B1, B2):
        C1, C2, R1, R2 = (torch.zeros((N, M), dtype=torch.float64, device="cuda") for _ in range(4))
        _lib.check(L.flgp_dev_gemm_pair(st, M, N, Kd, 1.0, A1.data_ptr(), A2.data_ptr(), 1, M, B1.data_ptr(), Bsecond.data_ptr(), 1, Kd,
                                        C1.data_ptr(), C2.data_ptr(), 1, M))
        for A, B, R in ((A1, B1, R1), (A2, Bsecond, R2)):
            _lib.check(L.flgp_dev_gemm(st, M, N, Kd, 1.0, A.data_ptr(), 1, M, B.data_ptr(), 1, Kd, 0.0, None, 0, 0, R.data_ptr(), 1, M, None, 0))
        torch.cuda.synchronize()
        assert torch.equal(C1, R1) and torch.equal(C2, R2)
        ref = (A2.cpu().numpy().T @ Bsecond.cpu().numpy().T)
        np.testing.assert_allclose(C2.cpu().numpy().T, ref, rtol=0, atol=1e-11 * np.sqrt(Kd))


@pytest.mark.parametrize("s,b", [(5000, 256), (4098, 192), (64, 64), (1030, 128), (20000, 256)])
def test_rotation_kernel_is_the_gemm_bit_for_bit(stages, s, b):
    """csrc/rot.hip: the solver's s x b by b x b rotation on its own kernel -- the same k-ascending chain per element as the
    general GEMM: identical bits, alone, as a pair, and in place with the E term (cur <- cur - Qold T)."""
    rng = np.random.default_rng(s + b)
    L = stages.L
    st = torch.cuda.current_stream().cuda_stream
    X1, X2, E = (torch.from_numpy(rng.normal(size=(b, s))).cuda() for _ in range(3))     # column-major s x b
    # column-major W(k, j) lives at W_cm[k + j b]; the kernel wants WT[k b + j] = W(k, j): the row-major copy of W
    W_cm = torch.from_numpy(rng.normal(size=(b, b))).cuda()        # tensor [j][k] == column-major W(k, j)
    W_km = W_cm.t().contiguous()                                   # tensor [k][j] == WT
    def gemm(Xt, alpha, beta, Et, out):
        _lib.check(L.flgp_dev_gemm(st, s, b, b, alpha, Xt.data_ptr(), 1, s, W_cm.data_ptr(), 1, b, beta, Et.data_ptr() if Et is not None else None,
                                   1, s, out.data_ptr(), 1, s, None, 0))
    R1, R2, O1, O2 = (torch.zeros((b, s), dtype=torch.float64, device="cuda") for _ in range(4))
    gemm(X1, 1.0, 0.0, None, R1); gemm(X2, 1.0, 0.0, None, R2)
    _lib.check(L.flgp_dev_rotate(st, s, b, 1.0, X1.data_ptr(), None, W_km.data_ptr(), 0.0, None, O1.data_ptr(), None))
    torch.cuda.synchronize()
    assert torch.equal(O1, R1)
    O1.zero_()
    _lib.check(L.flgp_dev_rotate(st, s, b, 1.0, X1.data_ptr(), X2.data_ptr(), W_km.data_ptr(), 0.0, None, O1.data_ptr(), O2.data_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(O1, R1) and torch.equal(O2, R2)
    Ein, Eref = E.clone(), E.clone()
    gemm(X1, -1.0, 1.0, Eref, Eref)                                # in place, as the de-contamination step does
    _lib.check(L.flgp_dev_rotate(st, s, b, -1.0, X1.data_ptr(), None, W_km.data_ptr(), 1.0, Ein.data_ptr(), Ein.data_ptr(), None))
    torch.cuda.synchronize()
    assert torch.equal(Ein, Eref)
    ref = X1.cpu().numpy().T @ W_cm.cpu().numpy().T
    np.testing.assert_allclose(R1.cpu().numpy().T, ref, rtol=0, atol=1e-11 * np.sqrt(b))


@pytest.mark.parametrize("s,b", [(5000, 256), (4098, 192), (256, 64), (1030, 128), (20000, 256), (777 * 2, 256)])
def test_gram_kernel_of_the_solver(stages, s, b):
    """csrc/rot.hip: the solver's b x b Gram product Xa^T Xb over s rows on its own kernel.  Same planes as the general GEMM's
    split-K where the row ranges coincide (s = 5000, b = 256: bit for bit), rounding-level otherwise; ragged last ranges."""
    rng = np.random.default_rng(s * 3 + b)
    L = stages.L
    st = torch.cuda.current_stream().cuda_stream
    Xa, Xb = (torch.from_numpy(rng.normal(size=(b, s))).cuda() for _ in range(2))        # column-major s x b
    work = torch.empty(64 * b * b, dtype=torch.float64, device="cuda")
    O, R = (torch.zeros((b, b), dtype=torch.float64, device="cuda") for _ in range(2))
    _lib.check(L.flgp_dev_gram_small(st, s, b, Xa.data_ptr(), Xb.data_ptr(), O.data_ptr(), work.data_ptr(), work.numel()))
    _lib.check(L.flgp_dev_gemm(st, b, b, s, 1.0, Xa.data_ptr(), s, 1, Xb.data_ptr(), 1, s, 0.0, None, 0, 0, R.data_ptr(), 1, b,
                               work.data_ptr(), work.numel()))
    torch.cuda.synchronize()
    ref = Xb.cpu().numpy() @ Xa.cpu().numpy().T            # tensor [jc][ic] of the column-major result
    np.testing.assert_allclose(O.cpu().numpy(), ref, rtol=0, atol=1e-11 * np.sqrt(s))
    if (s, b) == (5000, 256):
        assert torch.equal(O, R)
    _lib.check(L.flgp_dev_gram_small(st, s, b, Xa.data_ptr(), Xa.data_ptr(), O.data_ptr(), work.data_ptr(), work.numel()))
    torch.cuda.synchronize()
    assert torch.equal(O, O.t())                           # a Gram matrix of one block is symmetric bit for bit


@pytest.mark.parametrize("s,K", [(60, 60), (200, 30), (300, 300), (900, 100), (2000, 100),
                                 (1537, 90), (2050, 200), (1601, 40)])      # odd s: the general GEMM instead of csrc/rot.hip's kernels
def test_eig_topk(stages, s, K):
    rng = np.random.default_rng(s + K)
    # PSD with a repeated top eigenvalue (three copies of 1) and a decaying tail, as the path produces
    lam = np.concatenate([[1.0, 1.0, 1.0 - 1e-10], np.sort(rng.uniform(0.0, 0.97, s - 3))[::-1]])
    Qr, _ = np.linalg.qr(rng.normal(size=(s, s)))
    G = (Qr * lam) @ Qr.T
    G = 0.5 * (G + G.T)
    eig, V, info = stages.eig_topk(torch.from_numpy(G).cuda(), K)
    eig = eig.cpu().numpy(); V = to_np_cm(V)
    w = np.linalg.eigvalsh(G)[::-1][:K]
    np.testing.assert_allclose(eig, w, rtol=EIG_RTOL, atol=1e-13)
    np.testing.assert_allclose(V.T @ V, np.eye(K), atol=1e-10)
    assert np.abs(G @ V - V * eig).max() < 1e-9                       # eigenpair residuals
    # the invariant subspace agrees with LAPACK's (individual vectors inside clusters are not unique)
    Vr = np.linalg.eigh(G)[1][:, ::-1][:, :K]
    if K < s and w[K - 1] - np.linalg.eigvalsh(G)[::-1][K] > 1e-6:
        assert np.linalg.norm(Vr - V @ (V.T @ Vr)) < 1e-7


def test_eig_degenerate_spectrum_falls_back(oracle, stages):
    """A spectrum the Chebyshev filter cannot split (r = 1 makes G the identity up to the 1e-9 guards) ends in the
    full Jacobi decomposition instead of an error: any orthonormal basis of the degenerate eigenspace is valid."""
    s, K = 600, 30
    rng = np.random.default_rng(4)
    Q, _ = np.linalg.qr(rng.normal(size=(s, s)))
    G = (Q * (1.0 + 1e-10 * rng.normal(size=s))) @ Q.T
    G = 0.5 * (G + G.T)
    eig, V, info = stages.eig_topk(torch.from_numpy(G).cuda(), K)
    V = V.cpu().numpy().T; lam = eig.cpu().numpy()
    assert info["dense"]                                       # the fallback reports itself
    np.testing.assert_allclose(lam, 1.0, rtol=0, atol=1e-8)
    np.testing.assert_allclose(V.T @ V, np.eye(K), rtol=0, atol=1e-12)
    assert np.abs(G @ V - V * lam).max() < 1e-9
    # the whole path with r = 1: runs, and H = U e^{-t(1-1)} U^T restricted to K columns is a projector-like PSD block
    X, U0, U = make_case(2000, 3, 100, 1, seed=5)
    H = api.heat_kernel_covariance_cpp(X[:50], X[50:], 100, 1, 1.0, 10, dict(kernel="lae", gl="normalized", root=True), 1, 0.1, U=U)
    assert H.shape == (2000, 50) and np.isfinite(H).all()
    assert np.linalg.eigvalsh(0.5 * (H[:50] + H[:50].T)).min() > -1e-9


def test_eig_blocksparse_matches_dense(stages):
    """The Gram matrix of a neighbourhood graph is sparse; from s = 3072 on the solver permutes it and multiplies
    only the populated 64 x 16 blocks (+ a CSR remainder).  That must not change the result beyond rounding:
    same eigenpairs as the dense path and as LAPACK, for a manifold (swiss roll) and for clustered data, and for
    a matrix without structure (where the solver must notice and stay dense)."""
    from flgp_amd import _lib
    L = _lib.lib()
    K = 60
    cases = []
    for name, X in (("swiss", synth.swiss_roll(40000)[0]), ("mixture", synth.gaussian_mixture(40000, 8, components=6))):
        n, s, r = X.shape[0], 3200, 6
        sel = np.sort(synth.random_anchor_rows(n, s, seed=5))
        U0 = np.asfortranarray(X[sel])
        Z = api.cross_similarity_lae_cpp(X, U0, r, "normalized").toarray()
        A = Z / np.sqrt(np.abs(Z.sum(0)) + 1e-9)
        cases.append((name, A.T @ A))
    rng = np.random.default_rng(8)
    Qr, _ = np.linalg.qr(rng.normal(size=(3200, 3200)))
    cases.append(("dense", (Qr * np.sort(rng.uniform(0, 1, 3200))[::-1]) @ Qr.T))
    try:
        for name, G in cases:
            G = 0.5 * (G + G.T)
            w = np.linalg.eigvalsh(G)[::-1][:K]
            out = {}
            for flag in (1, 0):
                L.flgp_set_tuning(b"eig_blocksparse", flag)
                eig, V, info = stages.eig_topk(torch.from_numpy(G).cuda(), K)
                eig = eig.cpu().numpy(); V = to_np_cm(V)
                np.testing.assert_allclose(eig, w, rtol=EIG_RTOL, atol=1e-13, err_msg=name)
                np.testing.assert_allclose(V.T @ V, np.eye(K), atol=1e-10, err_msg=name)
                assert np.abs(G @ V - V * eig).max() < 1e-9, name
                out[flag] = (eig, V)
            np.testing.assert_allclose(out[1][0], out[0][0], rtol=1e-10, atol=1e-13)
    finally:
        L.flgp_set_tuning(b"eig_blocksparse", 1)


@pytest.mark.parametrize("s,b,density,part_cap", [(700, 48, 0.03, 24), (1100, 64, 0.02, 24), (2000, 256, 0.01, 24),
                                                  (1536, 80, 0.05, 24), (2000, 256, 0.06, 8), (1536, 80, 0.1, 8),
                                                  (2000, 256, 0.06, 0)])
def test_blocksparse_product_matches_dense(stages, s, b, density, part_cap):
    """csrc/bsg.hip on its own: alpha G X + beta E through the ordering, the kept 64 x 16 blocks (MFMA) and the CSR
    remainder equals the dense product to rounding -- for a banded-plus-scattered symmetric matrix (blocks AND a
    remainder, tile / stage / column-tile edges that do not divide s or b), including rows without any entry.  With
    a small `eig_bs_part_cap` the tiles' lists are cut into parts whose partial sums meet in slabs inside the launch
    (the last part to arrive adds them in part order): same result, launch after launch (the arrival counters reset)."""
    import ctypes
    L = _lib.lib()
    L.flgp_set_tuning(b"eig_bs_part_cap", part_cap)
    rng = np.random.default_rng(s + b)
    G = np.zeros((s, s))
    band = max(8, int(density * s))
    for off in range(band):                                   # a band: concentrates after the ordering
        v = rng.uniform(0.1, 1.0, s - off)
        G[np.arange(s - off), np.arange(off, s)] = v
    scat = rng.integers(0, s, size=(4 * s, 2))                # scattered entries: the CSR remainder
    G[scat[:, 0], scat[:, 1]] = rng.uniform(0.1, 1.0, len(scat))
    hub = rng.integers(0, s, size=300)                        # one hub row with a few hundred entries
    G[7, hub] = rng.uniform(0.1, 1.0, 300)
    G = 0.5 * (G + G.T)
    G[s // 3, :] = 0.0; G[:, s // 3] = 0.0                    # an isolated anchor
    pi = rng.permutation(s)                                   # hide the band
    G = np.ascontiguousarray(G[np.ix_(pi, pi)])
    X = rng.normal(size=(s, b)); E = rng.normal(size=(s, b))
    dG = torch.from_numpy(G).cuda(); dX = cm(X); dE = cm(E)
    out = torch.empty((b, s), dtype=torch.float64, device="cuda")
    wb = L.flgp_dev_bsg_workspace(s, b)
    work = torch.empty((wb // 8 + 1,), dtype=torch.float64, device="cuda")
    info = (ctypes.c_int * 6)()
    st = torch.cuda.current_stream().cuda_stream
    try:
        prev = {}
        for alpha, beta, e in ((1.0, 0.0, None), (0.7, -1.3, dE), (1.0, 0.0, None)):
            _lib.check(L.flgp_dev_bsg_apply(st, dG.data_ptr(), s, s, dX.data_ptr(), b, alpha, beta,
                                            e.data_ptr() if e is not None else None, out.data_ptr(), work.data_ptr(), wb,
                                            ctypes.addressof(info)))
            ref = alpha * (G @ X) + (beta * E if e is not None else 0.0)
            got = to_np_cm(out)
            assert info[1] == np.count_nonzero(G)
            assert info[2] > 0 and info[3] > 0, "the case must exercise both the blocks and the remainder"
            ntile = (s + 63) // 64
            if part_cap == 8:
                assert info[5] > 0 and info[4] > ntile, "the case must cut tiles into parts"
            if part_cap == 0:
                assert info[5] == 0 and info[4] == ntile
            np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12 * np.abs(ref).max())
            if (alpha, beta) in prev:
                assert np.array_equal(prev[(alpha, beta)], got), "the same product twice: bit-identical"
            prev[(alpha, beta)] = got
    finally:
        L.flgp_set_tuning(b"eig_bs_part_cap", 0)


def test_blocksparse_ordering_does_not_depend_on_timing(stages):
    """Regression for the workspace overrun of round 3 (a229a15): with s % 64 != 0 the chunk tables of the ordering's cluster
    weights overran their buffer into rows other workgroups of the same launch were reading, so the ORDERING -- kept blocks,
    scattered entries, hence the solver's iteration count -- depended on who got there first, and it showed only once the
    set-up's Lanczos steps ran truly concurrently (page-locked result slots).  Here: a Gram matrix of the path with
    s = 2000 (s % 64 = 16); the set-up's counts (info[2] kept blocks, info[3] scattered entries of flgp_dev_bsg_apply) are the
    same launch after launch, and the eigensolver -- whose ordering is computed beside the concurrent Lanczos run -- returns
    the same iteration / product counts and bit-identical eigenvalues on every repeat in BOTH eig_host_slots modes."""
    import ctypes
    L = _lib.lib()
    n, d, s, r, K = 60000, 3, 2000, 5, 100
    assert s % 64 != 0
    X, _ = synth.swiss_roll(n, seed=11)
    U0 = synth.anchors_from_rows(X, np.sort(synth.random_anchor_rows(n, s, seed=11)))
    dX = cm(X); dU = cm(U0)
    anchors = stages.anchor_prep(dU)
    kidx, _ = stages.knn(dX, anchors, r)
    ei, ev = stages.lae(dX, anchors, kidx)
    csc = stages.csc(ei, s)
    stages.row_normalize(ev)
    c2 = stages.colsum(ei, ev, s); stages.col_scale(ei, ev, c2, None, 1)
    G = stages.gram(ei, ev, csc)
    torch.cuda.synchronize()
    b = 128
    Xb = torch.randn((b, s), dtype=torch.float64, device="cuda")
    out = torch.empty((b, s), dtype=torch.float64, device="cuda")
    wb = L.flgp_dev_bsg_workspace(s, b)
    work = torch.empty((wb // 8 + 1,), dtype=torch.float64, device="cuda")
    info = (ctypes.c_int * 6)()
    st = torch.cuda.current_stream().cuda_stream
    seen = set(); outs = []
    for _ in range(6):
        _lib.check(L.flgp_dev_bsg_apply(st, G.data_ptr(), s, s, Xb.data_ptr(), b, 1.0, 0.0, None, out.data_ptr(), work.data_ptr(), wb,
                                        ctypes.addressof(info)))
        torch.cuda.synchronize()
        seen.add((info[0], info[1], info[2], info[3], info[4]))
        outs.append(out.cpu().numpy().copy())
    assert len(seen) == 1 and next(iter(seen))[2] > 0, seen
    for o in outs[1:]:
        np.testing.assert_array_equal(o, outs[0])
    runs = []
    try:
        for mode in (1, 0, 1, 0):
            L.flgp_set_tuning(b"eig_host_slots", mode)
            for _ in range(3):
                eig, V, inf = stages.eig_topk(G, K)
                runs.append((mode, inf["outer_iterations"], inf["g_products"], eig.cpu().numpy().copy()))
    finally:
        L.flgp_set_tuning(b"eig_host_slots", 1)
    assert len({(r_[1], r_[2]) for r_ in runs}) == 1, [(r_[0], r_[1], r_[2]) for r_ in runs]
    for r_ in runs[1:]:
        np.testing.assert_array_equal(r_[3], runs[0][3])


# ------------------------------------------------------------------------------ spectrum + heat kernel
@pytest.mark.parametrize("n,d,s,r,K,root,gl", [
    (1500, 3, 120, 4, 20, True, "cluster-normalized"),
    (1500, 3, 120, 4, -1, False, "rw"),                 # K == s: the dense (BDCSVD) branch
    (4000, 16, 700, 10, 50, True, "normalized"),        # the block eigensolver branch
    (2500, 100, 300, 5, 40, True, "cluster-normalized"),   # d > 64 (an image-like input)
])
def test_spectrum_and_heat_kernel(oracle, n, d, s, r, K, root, gl):
    X, U0, U = make_case(n, d, s, r, seed=n + s + r)
    Z = api.cross_similarity_lae_cpp(X, U, r, gl)
    ei, zn = oracle.cross_similarity(X, U, r, gl=gl)
    ep = api.spectrum_from_Z_cpp(Z, K, root)
    ovals, ovec = oracle.spectrum_from_Z(ei, zn, s, K, root=root)
    Kk = ovals.size
    np.testing.assert_allclose(ep.values, ovals, rtol=EIG_RTOL, atol=1e-12)
    np.testing.assert_allclose(ep.vectors.T @ ep.vectors / n, np.eye(Kk), atol=1e-9)
    idx0 = np.arange(n, dtype=np.int32); idx1 = np.arange(64, dtype=np.int32)
    Kh = min(Kk, 100)          # the bottom of a full spectrum is not unique at sigma ~ 0
    H = api.HK_from_spectrum_cpp(ep, Kh, 2.5, idx0, idx1)
    Ho = oracle.hk_from_spectrum(ovals, ovec, Kh, 2.5, idx0, idx1)
    assert np.abs(H - Ho).max() <= H_RTOL * np.abs(Ho).max()
    # the contraction kernel on its own: same spectrum in, oracle contraction as the checker
    Hs = oracle.hk_from_spectrum(ep.values, ep.vectors, Kh, 2.5, idx0, idx1)
    assert np.abs(H - Hs).max() <= 1e-12 * np.abs(Hs).max()


def test_hk_gather_and_properties(oracle):
    rng = np.random.default_rng(2)
    n, K = 700, 37
    vec = np.asfortranarray(rng.normal(size=(n, K)))
    vals = np.sort(rng.uniform(0.1, 1.0, K))[::-1].copy()
    ep = api.EigenPair(vals, vec)
    idx0 = rng.permutation(n)[:300].astype(np.int32); idx1 = np.array([5, 699, 0, 5, 17], dtype=np.int32)
    H = api.HK_from_spectrum_cpp(ep, K, 1.3, idx0, idx1)              # general mat_indexing gather
    np.testing.assert_allclose(H, oracle.hk_from_spectrum(vals, vec, K, 1.3, idx0, idx1), rtol=0, atol=1e-12)
    full = np.arange(n, dtype=np.int32)
    H0 = api.HK_from_spectrum_cpp(ep, K, 0.0, full, full)
    np.testing.assert_allclose(H0, vec @ vec.T, atol=1e-11)           # H(t = 0) = V V^T
    Ht = api.HK_from_spectrum_cpp(ep, 10, 2.0, full, full)            # K smaller than stored
    np.testing.assert_allclose(Ht, Ht.T, atol=1e-12)
    assert np.linalg.eigvalsh(Ht).min() > -1e-10
    with pytest.raises(api.FlgpError):
        api.HK_from_spectrum_cpp(ep, K, 1.0, np.array([n], dtype=np.int32), idx1)


@pytest.mark.parametrize("n0,n1,K,gather", [(5000, 1000, 200, False), (2111, 77, 37, False), (4100, 300, 288, False),
                                            (3000, 130, 100, True), (2048, 64, 16, False), (70001, 333, 210, False),
                                            (6405, 999, 197, False), (9000, 520, 100, False), (4000, 481, 110, True)])
def test_hk_panel_kernel_bit_identical_to_gemm(oracle, n0, n1, K, gather):
    """The LDS-panel contractions (csrc/hk2.hip: unrolled, 13 or 7 stages of k, n1 >= 480; csrc/hk.hip: any K <= 288)
    against the tiled GEMM they replace for the path's shape: the same k-ascending MFMA chain per element, so every bit
    of H agrees between all three -- ragged panels (n0 % 64), ragged m-tiles and tile pairs (n1 % 16, n1 % 32), ragged k
    stages (K % 16), both register variants of hk.hip (K <= 224 / K <= 288), row gathers, offset ranges -- and the oracle's
    contraction within 1e-8 of max|H| (HK_from_spectrum_cpp, src/Spectrum.cpp:83-94)."""
    rng = np.random.default_rng(n0 + n1 + K)
    n = n0 + 50
    vec = np.asfortranarray(rng.normal(size=(n, K)))
    vals = np.sort(rng.uniform(0.1, 1.0, K))[::-1].copy()
    ep = api.EigenPair(vals, vec)
    if gather:
        idx0 = rng.permutation(n)[:n0].astype(np.int32)
        idx1 = rng.permutation(n)[:n1].astype(np.int32)
    else:
        idx0 = np.arange(7, 7 + n0, dtype=np.int32)          # an offset range: V0 starts at an odd row
        idx1 = np.arange(3, 3 + n1, dtype=np.int32)
    L = _lib.lib()
    Hp = api.HK_from_spectrum_cpp(ep, K, 1.7, idx0, idx1)
    L.flgp_set_tuning(b"hk_panel2", 0)
    try:
        H1 = api.HK_from_spectrum_cpp(ep, K, 1.7, idx0, idx1)      # the general panel kernel alone
        L.flgp_set_tuning(b"hk_panel", 0)
        Hg = api.HK_from_spectrum_cpp(ep, K, 1.7, idx0, idx1)      # the tiled GEMM
    finally:
        L.flgp_set_tuning(b"hk_panel", 1)
        L.flgp_set_tuning(b"hk_panel2", 1)
    np.testing.assert_array_equal(Hp, Hg)
    np.testing.assert_array_equal(H1, Hg)
    Ho = oracle.hk_from_spectrum(vals, vec, K, 1.7, idx0, idx1)
    assert np.abs(Hp - Ho).max() <= H_RTOL * np.abs(Ho).max()


@pytest.mark.parametrize("path", sorted(p for p in glob.glob(os.path.join(GOLDEN, "*.npz")) if "known" not in p))
def test_golden_fixtures_through_the_abi(path):
    g = np.load(path)
    X, U = g["X"], g["U"]
    d = X.shape[1]; s = U.shape[0]; r = int(g["r"]); K = int(g["K"]); t = float(g["t"]); m = int(g["m"])
    gl = str(g["gl"]); root = bool(g["root"])
    U0 = np.asfortranarray(U[:, :d])
    res = api.KNN_cpp(X, U0, r)
    np.testing.assert_array_equal(res["ind_knn"], g["knn_idx"])
    Zl = api.LAE_cpp(X, U0, r)
    np.testing.assert_array_equal(Zl.indices.reshape(-1, r), g["ell_idx"])
    np.testing.assert_array_equal(Zl.data.reshape(-1, r), g["lae_val"])
    Z = api.cross_similarity_lae_cpp(X, U, r, gl)
    np.testing.assert_array_equal(Z.data.reshape(-1, r), g["z_val"])
    H = api.heat_kernel_covariance_cpp(X[:m], X[m:], s, r, t, K, dict(kernel="lae", gl=gl, root=root), 1, 0.1, U=U)
    assert H.shape == g["H"].shape
    assert np.abs(H - g["H"]).max() <= H_RTOL * np.abs(g["H"]).max()


def test_end_to_end_c1_like(oracle):
    """BASELINE configs[0] shape (README torus: n=4800 d=2 s=600 r=3 K=100 m=100) end to end."""
    X, _ = synth.torus(4800)
    rows = synth.random_anchor_rows(4800, 600)
    U0 = synth.anchors_from_rows(X, rows)
    sizes = np.bincount(oracle.knn(X, U0, 1)[:, 0], minlength=600).astype(float)
    U = np.asfortranarray(np.hstack([U0, sizes[:, None]]))
    H = api.heat_kernel_covariance_rcpp(X[:100], X[100:], 600, 3, 10.0, K=100, U=U)
    Ho = oracle.heat_kernel_covariance(X[:100], X[100:], U, 3, 10.0, K=100)
    assert H.shape == (4800, 100)
    assert np.abs(H - Ho).max() <= H_RTOL * np.abs(Ho).max()
    em = api.lae_eigenmap(X, 600, r=3, ndim=4, U=U)
    vals, vecs_o = oracle.heat_kernel_spectrum(X, U, 3, 4, gl="cluster-normalized", root=True)
    np.testing.assert_allclose(em["eigenvalues"], 1 - vals, rtol=0, atol=1e-10)
    # the eigenvectors as well (a14): lae_eigenmap's embedding is the n x ndim block of U sqrt(n).  Individual vectors are
    # defined up to sign (and up to rotation inside a cluster of equal eigenvalues), so they are compared as a SUBSPACE --
    # |(I - Vo Vo^T / n) V| / sqrt(n) -- and, column by column after the sign is fixed, wherever the eigenvalue is
    # separated from its neighbours by more than 1e-6
    Vd = em["eigenvectors"]; n_ = Vd.shape[0]
    assert Vd.shape == vecs_o.shape == (n_, 4)
    resid = Vd - vecs_o @ (vecs_o.T @ Vd) / n_
    assert np.abs(resid).max() / np.sqrt(n_) < 1e-9 and np.linalg.norm(resid) / np.sqrt(n_) < 1e-8
    np.testing.assert_allclose(Vd.T @ Vd / n_, np.eye(4), atol=1e-9)
    gaps = np.minimum(np.abs(np.diff(vals, prepend=np.inf)), np.abs(np.diff(vals, append=-np.inf)))
    for k in range(4):
        if gaps[k] > 1e-6:
            sgn = np.sign(Vd[:, k] @ vecs_o[:, k])
            assert np.abs(sgn * Vd[:, k] - vecs_o[:, k]).max() < 1e-8 / gaps[k] * 1e-2 + 1e-9, k


def test_resident_eigenpair_matches_host_path(oracle):
    """the EigenPair kept in HBM gives, for every (t, idx0, idx1), the matrix the by-value entry point
    gives -- same kernels, so bit for bit -- and the resident spectrum equals the copied-back one."""
    X, U0, U = make_case(3000, 3, 200, 5, seed=21)
    m = 400
    models = dict(kernel="lae", gl="cluster-normalized", root=True)
    ep = api.heat_kernel_spectrum_cpp(X[:m], X[m:], 200, 5, 40, models, U=U)
    rp = api.heat_kernel_spectrum_resident(X[:m], X[m:], 200, 5, 40, models, U=U)
    assert (rp.n, rp.K) == (3000, 40)
    back = rp.to_host()
    np.testing.assert_array_equal(back.values, ep.values)
    np.testing.assert_array_equal(back.vectors, ep.vectors)
    up = api.ResidentEigenPair.from_host(ep)
    rng = np.random.default_rng(2)
    for t, idx0, idx1, K in [(10.0, np.arange(m), np.arange(m), 40), (0.5, np.arange(3000), np.arange(m), 40),
                             (3.0, rng.permutation(3000)[:257], rng.permutation(3000)[:33], 25)]:
        ref = api.HK_from_spectrum_cpp(ep, K, t, idx0, idx1)
        np.testing.assert_array_equal(rp.HK_from_spectrum_cpp(K, t, idx0, idx1), ref)
        np.testing.assert_array_equal(up.HK_from_spectrum_cpp(K, t, idx0, idx1), ref)
    with pytest.raises(api.FlgpError):
        rp.HK_from_spectrum_cpp(41, 1.0, np.arange(3), np.arange(3))        # K beyond the stored pairs
    with pytest.raises(api.FlgpError):
        rp.HK_from_spectrum_cpp(10, 1.0, np.array([3000]), np.arange(3))    # row out of range
    # the m > K consumers of V (src/train.cpp:393-433): V^T V, V^T Y, V C against numpy on the copied-back pair
    for K, idx in [(40, np.arange(m)), (25, rng.permutation(3000)[:777])]:
        V = ep.vectors[idx][:, :K]
        Y = rng.normal(size=(idx.size, 3)); C = rng.normal(size=(K, 2))
        np.testing.assert_allclose(rp.VtV(K, idx), V.T @ V, rtol=0, atol=1e-10 * idx.size)
        np.testing.assert_allclose(rp.VtY(K, idx, Y), V.T @ Y, rtol=0, atol=1e-10 * idx.size)
        np.testing.assert_allclose(rp.VC(K, idx, C), V @ C, rtol=0, atol=1e-11 * K)
    rp.free(); up.free()



@pytest.mark.parametrize("m,K,q", [(60, 40, 1), (40, 40, 2), (300, 40, 1), (900, 100, 3)])
def test_resident_regression_prediction_and_posterior_variance(oracle, m, K, q):
    """SURVEY 8f-2, finished in round 2: predict_regression_cpp (src/Predict.cpp:40-75) and
    posterior_covariance_regression (src/Utils.cpp:214-250) on the resident pair -- both branches (m <= K: Cholesky of
    the m x m kernel matrix; m > K: Woodbury on K x K) against the numpy restatements, index gathers and ranges."""
    n = 3000
    X, U0, U = make_case(n, 3, 200, 5, seed=77)
    models = dict(kernel="lae", gl="cluster-normalized", root=True)
    rp = api.heat_kernel_spectrum_resident(X[:m], X[m:], 200, 5, max(K, 100), models, U=U)
    ep = rp.to_host()
    rng = np.random.default_rng(m + K)
    Y = rng.normal(size=(m, q))
    sigma = 1e-3
    for idx0, idx1 in [(np.arange(m), np.arange(m, n)), (rng.permutation(n)[:m], rng.permutation(n)[:777])]:
        for t, noise in [(10.0, 0.1), (2.0, 1e-2)]:
            ref = oracle.np_predict_regression(ep.values, ep.vectors, Y, idx0, idx1, K, (t, noise), sigma)
            got = rp.predict_regression_cpp(Y, idx0, idx1, K, (t, noise), sigma)
            np.testing.assert_allclose(got, ref, rtol=0, atol=1e-9 * np.abs(ref).max())
            refv = oracle.np_posterior_covariance_regression(ep.values, ep.vectors, idx0, idx1, K, (t, noise), sigma)
            gotv = rp.posterior_covariance_regression(idx0, idx1, K, (t, noise), sigma)
            # the reference's formula is a difference of terms of size prior * |V^T V| / (var + sigma) (|V^T V| ~ m): two
            # correct fp64 evaluations differ by that times a few eps, whatever the result's own size
            prior = ((ep.vectors[idx1, :K] ** 2) * np.exp(-t * (1.0 - ep.values[:K]))).sum(1).max()
            np.testing.assert_allclose(gotv, refv, rtol=0, atol=1e-9 * np.abs(refv).max() + 2e-15 * prior * m / (noise + sigma))
            assert (gotv > 0).all()
    # noisepar = "different" (src/Predict.cpp:76-110): one noise variance per training row
    for idx0, idx1 in [(np.arange(m), np.arange(m, n)), (rng.permutation(n)[:m], rng.permutation(n)[:500])]:
        nz = rng.uniform(0.01, 0.5, m)
        pars = np.concatenate([[6.0], nz])
        ref = oracle.np_predict_regression_different(ep.values, ep.vectors, Y, idx0, idx1, K, pars, sigma)
        got = rp.predict_regression_cpp(Y, idx0, idx1, K, pars, sigma, noisepar="different")
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-9 * np.abs(ref).max())
    # ... and with equal variances it is the "same" model
    same = rp.predict_regression_cpp(Y, np.arange(m), np.arange(m, n), K, (6.0, 0.2), sigma)
    diff = rp.predict_regression_cpp(Y, np.arange(m), np.arange(m, n), K, np.concatenate([[6.0], np.full(m, 0.2)]), sigma, noisepar="different")
    np.testing.assert_allclose(diff, same, rtol=0, atol=1e-9 * np.abs(same).max())
    with pytest.raises(api.FlgpError):
        rp.predict_regression_cpp(Y, np.arange(m), np.array([n]), K, (1.0, 0.1), sigma)          # row out of range
    with pytest.raises(api.FlgpError):
        rp.posterior_covariance_regression(np.arange(m), np.arange(5), K, (1.0, 0.0), 0.0)       # var + sigma = 0
    rp.free()

def test_posterior_variance_many_new_rows(oracle):
    """posterior_covariance_regression's m <= K branch solves one m x m system per NEW row (src/Utils.cpp:227-237): with
    m_new ~ n the right-hand sides must be spread over the chip (round 2 ran them all inside the factorisation's single
    workgroup -- minutes at n = 1e6).  120 000 new rows against the numpy restatement."""
    import time
    n, m, K = 120_000, 60, 80
    X, U0, U = make_case(n, 3, 300, 5, seed=12)
    models = dict(kernel="lae", gl="cluster-normalized", root=True)
    rp = api.heat_kernel_spectrum_resident(X[:m], X[m:], 300, 5, K, models, U=U)
    ep = rp.to_host()
    idx0 = np.arange(m); idx1 = np.arange(m, n)
    t, noise, sigma = 5.0, 0.05, 1e-3
    t0 = time.perf_counter()
    gotv = rp.posterior_covariance_regression(idx0, idx1, K, (t, noise), sigma)
    dt = time.perf_counter() - t0
    refv = oracle.np_posterior_covariance_regression(ep.values, ep.vectors, idx0, idx1, K, (t, noise), sigma)
    prior = ((ep.vectors[idx1, :K] ** 2) * np.exp(-t * (1.0 - ep.values[:K]))).sum(1).max()
    np.testing.assert_allclose(gotv, refv, rtol=0, atol=1e-9 * np.abs(refv).max() + 2e-15 * prior * m / (noise + sigma))
    assert dt < 5.0, dt
    rp.free()


@pytest.mark.parametrize("n,d,s,a2,K,seed", [(3000, 3, 300, 1.0, 30, 0), (5000, 7, 500, 0.5, 60, 1),
                                              (2000, 16, 257, 10.0, 20, 2), (700, 2, 64, 0.1, 64 // 4, 3),
                                              (1500, 100, 200, 2.0, 20, 4)])
def test_nystrom_eigenpair(oracle, n, d, s, a2, K, seed):
    """the Nystrom-extension spectrum of fit_nystrom_* (reference src/Fit.cpp:244-289) against the numpy
    restatement.  fp64 tolerance, not bit-exact: the reference's distances come from an Eigen GEMM and its eigenpairs
    from ARPACK.  An eigenvector is determined up to sign and to 1/gap, so the bound is on error x relative gap."""
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, d)); U = X[rng.permutation(n)[:s]] + 0.01 * rng.normal(size=(s, d))
    vals, vecs = oracle.np_nystrom_eigenpair(X, U, a2, K)
    ep = api.nystrom_eigenpair_cpp(X, U, a2, K)
    np.testing.assert_allclose(ep.values, vals, rtol=1e-10, atol=0)
    sign = np.sign(np.sum(ep.vectors * vecs, axis=0))
    err = np.max(np.abs(ep.vectors * sign - vecs), axis=0) / np.max(np.abs(vecs), axis=0)
    gap = np.minimum(np.abs(np.diff(vals, prepend=np.inf)), np.abs(np.diff(vals, append=-np.inf))) / vals[0]
    assert np.max(err[:-1] * gap[:-1]) < 1e-10, (err, gap)      # the last pair's lower neighbour is not computed
    assert err[0] < 1e-12                                       # the trivial pair (value 1) is well separated
    # the resident variant holds the same numbers and feeds HK_from_spectrum_cpp
    rp = api.nystrom_eigenpair_cpp(X, U, a2, K, resident=True)
    assert (rp.n, rp.K) == (n, K)
    back = rp.to_host()
    np.testing.assert_array_equal(back.values, ep.values)
    np.testing.assert_array_equal(back.vectors, ep.vectors)
    idx = np.arange(min(n, 200))
    H = rp.HK_from_spectrum_cpp(K, 1.0, idx, idx)
    np.testing.assert_array_equal(H, api.HK_from_spectrum_cpp(ep, K, 1.0, idx, idx))
    rp.free()


def test_nystrom_dot_products_both_routes_bit_identical(oracle):
    """d > 32 takes its dot products from the MFMA GEMM, d <= 32 from scalar-operand FMA chains: the same chain, and the
    row sums are added in the same order, so forcing either route gives the same bits (and both match the oracle)."""
    from flgp_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(11)
    for n, d, s, K in [(1500, 40, 200, 12), (900, 9, 130, 8), (3000, 64, 257, 5)]:
        X = rng.normal(size=(n, d)) / np.sqrt(d); U = X[rng.permutation(n)[:s]] + 0.01 * rng.normal(size=(s, d))
        got = []
        try:
            for route in (0, 1):
                L.flgp_set_tuning(b"nystrom_dot_gemm", route)
                got.append(api.nystrom_eigenpair_cpp(X, U, 0.7, K))
        finally:
            L.flgp_set_tuning(b"nystrom_dot_gemm", -1)
        np.testing.assert_array_equal(got[0].values, got[1].values)
        np.testing.assert_array_equal(got[0].vectors, got[1].vectors)
        vals, vecs = oracle.np_nystrom_eigenpair(X, U, 0.7, K)
        np.testing.assert_allclose(got[0].values, vals, rtol=1e-10, atol=0)
        sign = np.sign(np.sum(got[0].vectors * vecs, axis=0))
        np.testing.assert_allclose(got[0].vectors[:, 0] * sign[0], vecs[:, 0], rtol=1e-11, atol=0)


def test_nystrom_row_shards_bit_identical(oracle):
    """pipeline.NystromPath: the extension of a row shard is the same bits as those rows of the whole extension
    (every output element is one k-ascending chain wherever its tile sits), so sharding rows over GPUs needs no
    exchange beyond the anchors."""
    import torch
    from flgp_amd.pipeline import HipStages, NystromPath
    st = HipStages("cuda:0")
    path = NystromPath(st)
    rng = np.random.default_rng(17)
    n, d, s, K = 5000, 5, 300, 20
    X = rng.normal(size=(n, d)); U = X[rng.permutation(n)[:s]]
    Xt = torch.from_numpy(np.ascontiguousarray(X.T)).cuda(); Ut = torch.from_numpy(np.ascontiguousarray(U.T)).cuda()
    vals, vecs = path.run_nystrom(Xt, Ut, 1.3, K)
    ep = api.nystrom_eigenpair_cpp(X, U, 1.3, K)
    np.testing.assert_array_equal(vals.cpu().numpy(), ep.values)
    np.testing.assert_array_equal(vecs.cpu().numpy().T, ep.vectors)
    for lo, hi in [(0, 1777), (1777, 5000), (4999, 5000)]:
        v2, w2 = path.run_nystrom(Xt[:, lo:hi].contiguous(), Ut, 1.3, K)
        np.testing.assert_array_equal(v2.cpu().numpy(), ep.values)
        np.testing.assert_array_equal(w2.cpu().numpy().T, ep.vectors[lo:hi])


@pytest.mark.parametrize("n,d,s,iter_max,seed", [(1200, 2, 4, 100, 0), (5000, 3, 60, 100, 1), (3000, 16, 200, 100, 2),
                                                  (4000, 7, 129, 3, 3), (900, 33, 50, 100, 4), (64, 1, 64, 5, 5),
                                                  (1500, 90, 40, 100, 6)])
def test_kmeans_lloyd_bit_exact(oracle, n, d, s, iter_max, seed):
    """Lloyd k-means on the device against its restatement -- labels come from the bit-exact k-NN kernel and
    the centre sums are taken in row order, so centres, sizes and the number of rounds all agree exactly."""
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, d)) + 4.0 * rng.integers(0, 3, size=(n, 1))
    rows = rng.choice(n, size=s, replace=False)
    Uo, ito = oracle.np_kmeans_lloyd(X, rows, iter_max)
    U, it, wss = api.kmeans_lloyd(X, s, rows, iter_max=iter_max)
    assert it == ito
    np.testing.assert_array_equal(U, Uo)
    assert U[:, d].sum() == n
    if it < iter_max:                                # converged: wss is the objective at the returned centres
        lab = oracle.knn(X, U[:, :d], 1)[:, 0]
        ref = ((X - U[lab, :d]) ** 2).sum()
        assert abs(wss - ref) <= 1e-9 * ref


def test_kmeans_lloyd_duplicates_nstart_and_errors(oracle):
    rng = np.random.default_rng(9)
    X = np.vstack([rng.normal(size=(300, 2)) + c for c in ([0, 0], [8, 0], [0, 8], [8, 8])])
    # two starting centres on the same point: the higher index never wins a tie, keeps its position, size 0
    Xd = X.copy(); Xd[7] = Xd[3]
    rows = np.array([3, 7, 400, 700])
    for iter_max in (1, 100):
        Uo, ito = oracle.np_kmeans_lloyd(Xd, rows, iter_max)
        U, it, _ = api.kmeans_lloyd(Xd, 4, rows, iter_max=iter_max)
        np.testing.assert_array_equal(U, Uo)
        assert it == ito
        if iter_max == 1:
            assert U[1, 2] == 0 and np.array_equal(U[1, :2], Xd[3])
    # nstart: the start with the smaller objective wins (a start with all centres in one blob loses)
    bad = np.array([0, 1, 2, 4]); good = np.array([0, 300, 600, 900])
    Ub, _, wb = api.kmeans_lloyd(X, 4, bad); Ug, _, wg = api.kmeans_lloyd(X, 4, good)
    U2, _, w2 = api.kmeans_lloyd(X, 4, np.stack([bad, good]))
    assert w2 == min(wb, wg)
    np.testing.assert_array_equal(U2, Ug if wg <= wb else Ub)
    # the mirror of subsample_cpp and the path on its anchors
    U = api.subsample_cpp(X, 40, method="lloyd", nstart=2, rng=np.random.default_rng(1))
    assert U.shape == (40, 3) and U[:, 2].sum() == 1200
    H = api.heat_kernel_covariance_cpp(X[:100], X[100:], 40, 3, 5.0, 10, dict(kernel="lae", gl="cluster-normalized", root=True), 1, 0.1, U=U)
    assert H.shape == (1200, 100) and np.isfinite(H).all()
    with pytest.raises(api.FlgpError):
        api.kmeans_lloyd(X, 4, np.array([0, 1, 2, 1200]))          # row out of range
    with pytest.raises(api.FlgpError):
        api.kmeans_lloyd(X[:3], 4, np.array([0, 1, 2, 2]))          # s > n


def test_nystrom_eigenpair_blocks_and_errors(oracle):
    """more rows than one row block of the extension holds; argument checks"""
    rng = np.random.default_rng(5)
    n, d, s, K = 70000, 4, 2100, 12            # row block = one round of the GEMM grid = 65536 rows: two blocks
    X = rng.normal(size=(n, d)); U = X[rng.permutation(n)[:s]]
    vals, vecs = oracle.np_nystrom_eigenpair(X, U, 1.0, K)
    ep = api.nystrom_eigenpair_cpp(X, U, 1.0, K)
    np.testing.assert_allclose(ep.values, vals, rtol=1e-10, atol=0)
    sign = np.sign(np.sum(ep.vectors * vecs, axis=0))
    np.testing.assert_allclose((ep.vectors * sign)[:, :4], vecs[:, :4], rtol=0, atol=1e-9 * np.max(np.abs(vecs[:, :4])))
    with pytest.raises(api.FlgpError):
        api.nystrom_eigenpair_cpp(X[:100], U[:10], 1.0, 11)       # K > s
    with pytest.raises(api.FlgpError):
        api.nystrom_eigenpair_cpp(X[:100], U[:10], 0.0, 3)        # a2 must be positive
    with pytest.raises(api.FlgpError):
        api.nystrom_eigenpair_cpp(X[:100], np.tile(U[:1], (10, 1)), 1.0, 3)   # coincident anchors: mean distance 0


def test_se_bandwidth_grid(oracle):
    """the spectrum part of fit_se_*: one k-NN, ten bandwidths, spectra run concurrently."""
    import time
    n, d, s, r, K, m = 3000, 3, 300, 5, 30, 100
    X, U0, U = make_case(n, d, s, r, seed=808)
    a2s = np.exp(np.linspace(np.log(0.1), np.log(10.0), 10))        # R/Fit.R:128-130
    t0 = time.perf_counter()
    pairs, mean = api.se_spectrum_grid(X[:m], X[m:], s, r, K=K, a2s=a2s, U=U, max_parallel=10)
    t_par = time.perf_counter() - t0
    ref, omean = oracle.se_spectrum_grid(X, U, r, K, a2s)
    assert abs(mean - omean) <= 1e-13 * omean
    idx0 = np.arange(n, dtype=np.int32); idx1 = np.arange(m, dtype=np.int32)
    for ep, (ov, ovec) in zip(pairs, ref):
        np.testing.assert_allclose(ep.values, ov, rtol=EIG_RTOL, atol=1e-12)
        H = oracle.hk_from_spectrum(ep.values, ep.vectors, K, 4.0, idx0, idx1)
        Ho = oracle.hk_from_spectrum(ov, ovec, K, 4.0, idx0, idx1)
        assert np.abs(H - Ho).max() <= H_RTOL * np.abs(Ho).max()
    # same numbers one bandwidth at a time (the concurrency must not change results)
    t0 = time.perf_counter()
    seq, _ = api.se_spectrum_grid(X[:m], X[m:], s, r, K=K, a2s=a2s, U=U, max_parallel=1)
    t_seq = time.perf_counter() - t0
    for a, b in zip(pairs, seq):
        np.testing.assert_array_equal(a.values, b.values)
    print(f"SE grid: 10 spectra concurrent {t_par*1e3:.1f} ms, sequential {t_seq*1e3:.1f} ms")


def test_unusable_spectrum_is_refused_on_every_path(stages):
    """sigma_K = 0 (K reaches into the null space: here K == s with an anchor that no point chose, and an SE bandwidth
    far below the neighbour distances) has no left vectors on the Gram route.  Every path that goes from the
    eigensolver to u = A v / sigma must say so -- the host spectrum, the bandwidth grid and the device stage the sharded
    driver uses (the last two returned zero / noise columns with FLGP_OK)."""
    n, d, s, r, m = 600, 3, 40, 3, 50
    X, U0, U = make_case(n, d, s, r, seed=31)
    U = U.copy(); U[7, :d] = 1e6                      # an anchor far away from every point: its column of Z is empty
    U0 = np.asfortranarray(U[:, :d])
    with pytest.raises(api.FlgpError):
        api.heat_kernel_spectrum_cpp(X[:m], X[m:], s, r, s, dict(kernel="lae", gl="rw", root=True), U=U)
    with pytest.raises(api.FlgpError):
        api.se_spectrum_grid(X[:m], X[m:], s, r, K=s, a2s=np.array([1.0]), models=dict(gl="rw"), U=U, max_parallel=1)
    # the device stage: eigenvalues with a zero at the end
    eig = torch.tensor([1.0, 0.5, 0.0], dtype=torch.float64, device="cuda:0")
    V = torch.eye(3, s, dtype=torch.float64, device="cuda:0")
    ei = torch.zeros((8, r), dtype=torch.int32, device="cuda:0"); ev = torch.ones((8, r), dtype=torch.float64, device="cuda:0")
    with pytest.raises(Exception):
        stages.u_recover(ei, ev, V, eig, 1.0, True)
    # The floor follows what the route that computed the eigenvalues resolves: 1e-10 of the top one for the
    # block solver (residuals 5e-11 lambda_1), 1e-13 for the full decomposition.
    import ctypes
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    for low, full, ok in ((1e-9, 0, True), (1e-11, 0, False), (1e-11, 1, True), (1e-14, 1, False), (0.0, 1, False)):
        e3 = torch.tensor([1.0, 0.5, low], dtype=torch.float64, device="cuda:0")
        rc = L.flgp_dev_spectrum_usable_route(st, e3.data_ptr(), 3, full)
        assert (rc == 0) == ok, (low, full, rc)
        if not ok: assert rc == -5 and b"null space" in L.flgp_last_error()
    assert L.flgp_dev_spectrum_usable(st, torch.tensor([1.0, 1e-11], dtype=torch.float64, device="cuda:0").data_ptr(), 2) == -5


def test_pipeline_matches_host_entry_points(oracle, stages):
    n, d, s, r, K, m, t = 3000, 16, 300, 10, 40, 128, 6.0
    X, U0, U = make_case(n, d, s, r, seed=2024)
    path = HeatKernelPath(stages)
    res = path.run(cm(X), cm(U0), PathConfig(s=s, r=r, K=K, t=t, m=m), n, 0, num_class=torch.from_numpy(U[:, d].copy()).cuda(), keep=True)
    H = api.heat_kernel_covariance_rcpp(X[:m], X[m:], s, r, t, K=K, U=U)
    Hp = to_np_cm(res.H)
    assert np.abs(Hp - H).max() <= 1e-12 * np.abs(H).max()
    np.testing.assert_array_equal(to_np_cm(res.knn_idx), oracle.knn(X, U0, r))


def test_pipelined_copy_of_H_is_the_same_matrix(oracle):
    """The host boundary sends H in column blocks (GEMM, PCIe and the host copy overlapped: csrc/capi.hip,
    hk_ranges_to_host).  Forced down to many small blocks -- with a ragged last one -- it must deliver, bit for bit, the
    matrix of the single-copy path, through both entry points that use it."""
    L = _lib.lib()
    X, U0, U = make_case(20000, 3, 300, 5, seed=31)
    m = 333
    models = dict(kernel="lae", gl="cluster-normalized", root=True)
    try:
        L.flgp_set_tuning(b"hk_pipelined_d2h", 0)
        H0 = api.heat_kernel_covariance_cpp(X[:m], X[m:], 300, 5, 3.0, 40, models, 1, 0.1, U=U)
        ep = api.heat_kernel_spectrum_cpp(X[:m], X[m:], 300, 5, 40, models, U=U)
        G0 = api.HK_from_spectrum_cpp(ep, 40, 3.0, np.arange(100, 20000), np.arange(5, 305))
        L.flgp_set_tuning(b"hk_pipelined_d2h", 1)
        L.flgp_set_tuning(b"hk_block_mb", 1)            # 1 MB blocks: 6 columns of 20000 rows -> 56 blocks
        H1 = api.heat_kernel_covariance_cpp(X[:m], X[m:], 300, 5, 3.0, 40, models, 1, 0.1, U=U)
        G1 = api.HK_from_spectrum_cpp(ep, 40, 3.0, np.arange(100, 20000), np.arange(5, 305))
    finally:
        L.flgp_set_tuning(b"hk_pipelined_d2h", 1)
        L.flgp_set_tuning(b"hk_block_mb", 512)
    np.testing.assert_array_equal(H1, H0)
    np.testing.assert_array_equal(G1, G0)
    # Two callers at once take a staging ring each (the lock covers the hand-out, not the transfer); the idle rings go
    # back to the system on request and the next call simply pins again.
    import threading
    out = [None, None]
    def call(q):
        out[q] = api.HK_from_spectrum_cpp(ep, 40, 3.0, np.arange(100, 20000), np.arange(5, 305))
    try:
        L.flgp_set_tuning(b"hk_block_mb", 1)
        th = [threading.Thread(target=call, args=(q,)) for q in range(2)]
        for t_ in th: t_.start()
        for t_ in th: t_.join()
        L.flgp_release_pinned()
        G2 = api.HK_from_spectrum_cpp(ep, 40, 3.0, np.arange(100, 20000), np.arange(5, 305))
    finally:
        L.flgp_set_tuning(b"hk_block_mb", 512)
    for g in (out[0], out[1], G2):
        np.testing.assert_array_equal(g, G0)


# ------------------------------------------------------------------------------ full size (BASELINE configs[2])
def test_full_size_properties(oracle, stages):
    """BASELINE configs[2] at full size, n = 1e6, d = 16, s = 5000, r = 10, K = 200, m = 1000, against the oracle run on
    the same 1e6 rows: neighbour sets, LAE / Laplacian values and the Gram matrix bit-exact, eigenvalues 1e-10 against
    LAPACK on the oracle's Gram matrix, covariance entries 1e-8 of max|H| against the oracle's chain -- plus the
    size-independent properties."""
    n, d, s, r, K, m, t = 1_000_000, 16, 5000, 10, 200, 1000, 10.0
    X = synth.gaussian_mixture(n, d)
    sel = np.sort(synth.random_anchor_rows(n, s))
    U0 = np.asfortranarray(X[sel])
    path = HeatKernelPath(stages)
    dX = cm(X); dU = cm(U0)
    sizes = path.cluster_sizes(dX, stages.anchor_prep(dU))
    assert float(sizes.sum()) == n
    res = path.run(dX, dU, PathConfig(s=s, r=r, K=K, t=t, m=m), n, 0, num_class=sizes, keep=True)
    # ---- the north_star sentence as a test: the ORACLE runs the similarity build on all n = 1e6 rows
    U = np.asfortranarray(np.hstack([U0, sizes.cpu().numpy()[:, None]]))
    ei_o, zn_o = oracle.cross_similarity(X, U, r, gl="cluster-normalized")            # k-NN + LAE + Laplacian
    av_o, _ = oracle.scale_A(ei_o, zn_o, s)                                            # spectrum scaling
    np.testing.assert_array_equal(res.ell_idx.cpu().numpy(), ei_o)                     # neighbour sets: every row
    np.testing.assert_array_equal(res.ell_val.cpu().numpy(), av_o)                     # LAE / Laplacian / A values: bit-exact
    rows = np.random.default_rng(0).choice(n, 3000, replace=False)
    oi = oracle.knn(np.asfortranarray(X[rows]), U0, r)
    np.testing.assert_array_equal(res.knn_idx[:, torch.from_numpy(rows).cuda()].cpu().numpy().T, oi)   # in distance order too
    G_o = oracle.gram(ei_o, av_o, s)
    np.testing.assert_array_equal(res.G.cpu().numpy(), G_o)                            # Gram: bit-exact
    # eigenvalues against LAPACK on the oracle's Gram matrix; H against the oracle's own chain (eigh -> u = A v / sigma
    # -> sqrt(n) -> heat kernel) on a slice of rows and the whole training block
    import scipy.linalg as sl_
    w_o, V_o = sl_.eigh(G_o, subset_by_index=[s - K, s - 1])
    w_o = w_o[::-1]; V_o = V_o[:, ::-1]
    vals = res.values.cpu().numpy()
    np.testing.assert_allclose(vals ** 2, w_o, rtol=EIG_RTOL, atol=0)                  # root=True: values = sigma
    # ... and against the REFERENCE's route at this size (src/TruncatedSVD.cpp:23-30: RSpectra::svds on A itself, restated
    # with ARPACK's implicitly restarted Lanczos, tol 1e-10, ncv = 2K + 1): the singular values of the 1e6 x 5000 matrix
    from scipy.sparse.linalg import svds as _svds
    A_o = oracle.ell_to_csr(ei_o, av_o, s).tocsc()
    sv = _svds(A_o, k=K, ncv=2 * K + 1, tol=1e-10, which="LM", v0=np.random.default_rng(0).standard_normal(s), maxiter=1000 * s,
               return_singular_vectors=False)
    np.testing.assert_allclose(vals, np.sort(sv)[::-1], rtol=1e-9, atol=0)
    pick = np.concatenate([np.arange(m), np.arange(123456, 123456 + 2048), np.arange(n - 1024, n)])
    Uo = oracle.u_recover(ei_o[pick], av_o[pick], s, np.asfortranarray(V_o), np.sqrt(w_o))
    vec_o = np.asfortranarray(Uo * (np.sqrt(float(n)) / np.sqrt(float(pick.size))))    # (the oracle scales by sqrt(rows given))
    H_o = oracle.hk_from_spectrum(np.sqrt(w_o), vec_o, K, t, np.arange(pick.size), np.arange(m))
    H_d = res.H[:, torch.from_numpy(pick).cuda()].cpu().numpy().T                      # (rows, m)
    assert np.abs(H_d - H_o).max() <= H_RTOL * np.abs(H_o).max(), np.abs(H_d - H_o).max() / np.abs(H_o).max()
    assert int(res.ell_idx.min()) >= 0 and int(res.ell_idx.max()) < s
    assert bool((res.ell_idx[:, 1:] > res.ell_idx[:, :-1]).all())            # CSR inner order, no duplicates
    # spectrum: sigma_1 = 1, descending, V^T V = n I
    assert abs(vals[0] - 1.0) < 1e-6 and (np.diff(vals) <= 1e-12).all() and vals[-1] > 0
    VtV = (res.vectors @ res.vectors.T / n).cpu().numpy()
    np.testing.assert_allclose(VtV, np.eye(K), atol=1e-8)
    # H on the training block: symmetric PSD
    Htrain = res.H[:, :m].cpu().numpy()                                       # (m, m)
    np.testing.assert_allclose(Htrain, Htrain.T, atol=1e-8 * np.abs(Htrain).max())
    assert np.linalg.eigvalsh(0.5 * (Htrain + Htrain.T)).min() > -1e-6 * np.abs(Htrain).max()


def test_c2_swiss_roll_regression_config(oracle):
    """BASELINE configs[1]: Swiss roll n = 1e5, d = 3, s = 2000, r = 5, K = 100 -- the covariance the regression driver
    consumes, through the host entry point (the R boundary), against the oracle's svds route: eigenvalues 1e-10, every
    entry of H within 1e-8 of max|H|."""
    n, s, r, K, m, t = 100_000, 2000, 5, 100, 500, 10.0
    X, _ = synth.swiss_roll(n)
    sel = np.sort(synth.random_anchor_rows(n, s))
    U0 = synth.anchors_from_rows(X, sel)
    sizes = np.bincount(oracle.knn(X, U0, 1)[:, 0], minlength=s).astype(float)
    U = np.asfortranarray(np.hstack([U0, sizes[:, None]]))
    models = dict(kernel="lae", gl="cluster-normalized", root=True)
    ep = api.heat_kernel_spectrum_cpp(X[:m], X[m:], s, r, K, models, U=U)
    vals_o, vec_o = oracle.heat_kernel_spectrum(X, U, r, K, gl="cluster-normalized", root=True, method="svds")
    np.testing.assert_allclose(ep.values, vals_o, rtol=EIG_RTOL, atol=0)
    H = api.heat_kernel_covariance_rcpp(X[:m], X[m:], s, r, t, K=K, U=U)
    H_o = oracle.hk_from_spectrum(vals_o, vec_o, K, t, np.arange(n), np.arange(m))
    assert H.shape == (n, m)
    assert np.abs(H - H_o).max() <= H_RTOL * np.abs(H_o).max(), np.abs(H - H_o).max() / np.abs(H_o).max()
    Z = api.cross_similarity_lae_cpp(X, U, r, "cluster-normalized")
    ei_o, zn_o = oracle.cross_similarity(X, U, r, gl="cluster-normalized")
    np.testing.assert_array_equal(Z.indices.reshape(n, r), ei_o)
    np.testing.assert_array_equal(Z.data.reshape(n, r), zn_o)


def test_c5_nystrom_full_size(oracle, stages):
    """BASELINE configs[4]: n = 5e6, d = 64, s = 10000, K = 500, the Nystrom-extension spectrum, at full size on one GPU.
    The extension is row-local given the anchors, so the numpy restatement on a sample of rows (with all 10^4 anchors:
    the dense 10^4 x 10^4 eigenproblem in LAPACK) checks the full-size run; plus the properties the whole result must
    have.  The cloud is drawn on the device (the host generator needs minutes for 3.2e8 normals)."""
    n, d, s, K, a2 = 5_000_000, 64, 10_000, 500, 1.0
    g = torch.Generator(device="cuda"); g.manual_seed(20241022)
    centers = 2.0 * torch.randn((16, d), generator=g, device="cuda", dtype=torch.float64)
    comp = torch.randint(0, 16, (n,), generator=g, device="cuda")
    X = torch.empty((d, n), dtype=torch.float64, device="cuda")                 # column-major n x d
    for k in range(d):                                                          # a column at a time: no n x d temporaries
        X[k] = centers[comp, k] + torch.randn((n,), generator=g, device="cuda", dtype=torch.float64)
    X /= float(np.sqrt(d))
    sel = torch.randperm(n, generator=g, device="cuda")[:s].sort().values
    U = X[:, sel].contiguous()                                                   # (d, s)
    values, vectors = stages.nystrom(X, U, a2, K)
    torch.cuda.synchronize()
    assert values.shape == (K,) and vectors.shape == (K, n)
    vals = values.cpu().numpy()
    assert abs(vals[0] - 1.0) < 1e-4 and (np.diff(vals) <= 1e-12).all() and vals[-1] > 0   # (1e-9 guards on row sums of ~1e-4)
    assert bool(torch.isfinite(vectors).all())
    # the trivial pair: W 1 = 1, so the first extended vector is constant (+-1 after the reference's scaling)
    v0 = vectors[0]
    assert float((v0 - v0[0]).abs().max()) < 1e-4 * float(v0[0].abs())
    # the oracle on a sample of rows + the anchors themselves (X = U rows reproduce the anchor eigenvectors)
    rows = torch.from_numpy(np.random.default_rng(1).choice(n, 1500, replace=False)).cuda()
    Xs = X[:, rows].T.cpu().numpy(); Uh = U.T.cpu().numpy()
    vals_o, vec_o = oracle.np_nystrom_eigenpair(Xs, Uh, a2, K)
    np.testing.assert_allclose(vals, vals_o, rtol=EIG_RTOL, atol=0)
    got = vectors[:, rows].T.cpu().numpy()
    sign = np.sign(np.sum(got * vec_o, axis=0))
    err = np.max(np.abs(got * sign - vec_o), axis=0) / np.max(np.abs(vec_o), axis=0)
    gap = np.minimum(np.abs(np.diff(vals_o, prepend=np.inf)), np.abs(np.diff(vals_o, append=-np.inf))) / vals_o[0]
    assert np.max(err[:-1] * gap[:-1]) < 1e-9, float(np.max(err[:-1] * gap[:-1]))
    assert err[0] < 1e-10


# ------------------------------------------------------------------------------ row sharding on the real stages
def _shard_worker(rank, world, port, out, cfg=(6000, 16, 400, 10, 40, 300, 5.0, 5, 314), sample=None):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)   # RCCL refuses two ranks on one card
    try:
        from flgp_amd.pipeline import shard_bounds
        n, d, s, r, K, m, t, comps, seed = cfg
        st = HipStages("cuda:0")
        path = HeatKernelPath(st)
        lo, hi = shard_bounds(n, world, rank)
        X = synth.gaussian_mixture(hi - lo, d, components=comps, seed=seed, row_offset=lo)
        sel = np.sort(synth.random_anchor_rows(n, s, seed=seed))
        mine = sel[(sel >= lo) & (sel < hi)] - lo
        U = path.gather_anchors(torch.from_numpy(np.ascontiguousarray(X[mine].T)).cuda())
        dX = cm(X)
        sizes = path.cluster_sizes(dX, st.anchor_prep(U))
        res = path.run(dX, U, PathConfig(s=s, r=r, K=K, t=t, m=m), n, lo, num_class=sizes)
        if sample is None:
            np.save(os.path.join(out, f"H_{world}_{rank}.npy"), to_np_cm(res.H))
        else:      # full-size runs: the rows of `sample` (global indices) this rank owns, in order
            loc = sample[(sample >= lo) & (sample < hi)] - lo
            np.save(os.path.join(out, f"H_{world}_{rank}.npy"), res.H[:, torch.from_numpy(loc).cuda()].T.contiguous().cpu().numpy())
        np.save(os.path.join(out, f"v_{world}_{rank}.npy"), res.values.cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_row_sharding_two_ranks_on_one_gpu(tmp_path):
    """The multi-GPU driver (flgp_amd/pipeline.py) with the real HIP stages: two ranks share the one
    card of this box over gloo and must reproduce the single-rank covariance (column sums and the
    Gram matrix are reduced in a different association: rounding-level differences only)."""
    import socket
    import torch.multiprocessing as mp
    out = str(tmp_path)
    for world in (1, 2):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        mp.spawn(_shard_worker, args=(world, port, out), nprocs=world, join=True)
    H1 = np.load(os.path.join(out, "H_1_0.npy"))
    H2 = np.vstack([np.load(os.path.join(out, f"H_2_{r}.npy")) for r in range(2)])
    assert H1.shape == H2.shape == (6000, 300)
    assert np.abs(H1 - H2).max() <= H_RTOL * np.abs(H1).max()
    np.testing.assert_allclose(np.load(os.path.join(out, "v_2_1.npy")), np.load(os.path.join(out, "v_1_0.npy")), rtol=EIG_RTOL)


def test_c4_full_size_row_sharded_two_ranks(tmp_path):
    """BASELINE configs[3] (C4): the n = 1e6, d = 16, s = 5000, r = 10, K = 200 workload of configs[2] row-sharded over
    two ranks (the one card of this box, gloo standing in for RCCL, the real HIP stages on every rank: k-NN, LAE,
    Laplacian, Gram partials, replicated eigensolve, U-recovery, H) against the single-rank run: eigenvalues 1e-10, H on
    the training block and 8192 sampled rows 1e-8 of max|H| (the column sums and the Gram matrix are added in a
    different association across ranks, nothing else differs)."""
    import socket
    import torch.multiprocessing as mp
    out = str(tmp_path)
    cfg = (1_000_000, 16, 5000, 10, 200, 1000, 10.0, 16, 20241022)
    rng = np.random.default_rng(4)
    sample = np.sort(np.concatenate([np.arange(1000), 1000 + rng.choice(999_000, 8192, replace=False)]))
    for world in (1, 2):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        mp.spawn(_shard_worker, args=(world, port, out, cfg, sample), nprocs=world, join=True)
    H1 = np.load(os.path.join(out, "H_1_0.npy"))
    H2 = np.vstack([np.load(os.path.join(out, f"H_2_{r}.npy")) for r in range(2)])
    assert H1.shape == H2.shape == (sample.size, 1000)
    assert np.abs(H1 - H2).max() <= H_RTOL * np.abs(H1).max(), np.abs(H1 - H2).max() / np.abs(H1).max()
    v1 = np.load(os.path.join(out, "v_1_0.npy"))
    for r in range(2):
        np.testing.assert_allclose(np.load(os.path.join(out, f"v_2_{r}.npy")), v1, rtol=EIG_RTOL)
    assert abs(v1[0] - 1.0) < 1e-6


# ------------------------------------------------------------------------------ the sharded path behind the C ABI
def _run_ranks(world, fn):
    """Run fn(rank) on `world` host threads (the ranks of an in-process communicator) and return the results."""
    import threading
    out = [None] * world; err = [None] * world
    def body(q):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device="cuda:0")):
                out[q] = fn(q)
                torch.cuda.current_stream().synchronize()
        except BaseException as e:      # noqa: BLE001
            err[q] = e
    th = [threading.Thread(target=body, args=(q,)) for q in range(world)]
    for t in th: t.start()
    for t in th: t.join()
    for e in err:
        if e is not None:
            raise e
    return out


@pytest.mark.parametrize("world", [2, 3])
def test_c_abi_communicator_collectives(world):
    """flgp_comm, in-process backend (csrc/comm.hip): ranks are host threads sharing the one card; all-reduce adds the
    ranks' buffers in rank order (bit-identical on every rank), all-gather concatenates -- plus the two helpers built on
    them, the ragged anchor gather and the global 1-NN cluster counts."""
    L = _lib.lib()
    comms = (ctypes.c_void_p * world)()
    _lib.check(L.flgp_comm_inproc_create(world, comms))
    rng = np.random.default_rng(world)
    data = [rng.normal(size=100_003) for _ in range(world)]
    n, d, s = 4000, 5, 37
    X = synth.gaussian_mixture(n, d, components=4, seed=9)
    cnts = [s // world + (1 if q < s % world else 0) for q in range(world)]
    offs = np.concatenate([[0], np.cumsum(cnts)])
    Uall = np.asfortranarray(X[np.sort(synth.random_anchor_rows(n, s, seed=9))])
    from flgp_amd.pipeline import shard_bounds

    def body(q):
        st = torch.cuda.current_stream().cuda_stream
        c = comms[q]
        assert L.flgp_comm_rank(c) == q and L.flgp_comm_world(c) == world
        buf = torch.from_numpy(data[q].copy()).cuda()
        _lib.check(L.flgp_comm_all_reduce_sum(c, buf.data_ptr(), buf.numel(), st))
        send = torch.from_numpy(data[q][:1000].copy()).cuda(); recv = torch.empty(1000 * world, dtype=torch.float64, device="cuda:0")
        _lib.check(L.flgp_comm_all_gather(c, send.data_ptr(), recv.data_ptr(), 1000, st))
        # anchors: rank q contributes rows offs[q]:offs[q+1] (column-major s_loc x d)
        Ul = cm(Uall[offs[q]:offs[q + 1]])
        Uo = torch.empty((d, s), dtype=torch.float64, device="cuda:0")
        _lib.check(L.flgp_dev_gather_anchors(st, c, Ul.data_ptr(), cnts[q], d, Uo.data_ptr(), s))
        lo, hi = shard_bounds(n, world, q)
        Xl = cm(X[lo:hi])
        sizes = torch.empty(s, dtype=torch.float64, device="cuda:0")
        _lib.check(L.flgp_dev_cluster_sizes(st, c, Xl.data_ptr(), hi - lo, hi - lo, d, Uo.data_ptr(), s, s, sizes.data_ptr()))
        torch.cuda.current_stream().synchronize()
        return buf.cpu().numpy(), recv.cpu().numpy(), to_np_cm(Uo), sizes.cpu().numpy()

    try:
        res = _run_ranks(world, body)
    finally:
        for q in range(world):
            L.flgp_comm_destroy(comms[q])
    want = data[0].copy()
    for q in range(1, world):
        want = want + data[q]                                          # rank order
    from oracle import flgp_oracle as O
    sizes_o = np.bincount(O.knn(X, Uall, 1)[:, 0], minlength=s).astype(float)
    for q in range(world):
        np.testing.assert_array_equal(res[q][0], want)
        np.testing.assert_array_equal(res[q][1], np.concatenate([data[z][:1000] for z in range(world)]))
        np.testing.assert_array_equal(res[q][2], Uall)
        np.testing.assert_array_equal(res[q][3], sizes_o)


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_sharded_covariance_behind_the_c_abi(oracle, devices):
    """flgp_heat_kernel_covariance_multi: the host boundary of the row-sharded path (what the R shim calls when
    FLGP_DEVICES lists several GPUs).  Ranks = host threads, here sharing the one card through the in-process
    communicator: k-NN / LAE / scalings per row block, column sums, packed Gram partials and the training block exchanged
    inside the C library, eigensolve replicated.  Against the single-GPU entry point (rounding-level differences from the
    association of the sums only) and the oracle."""
    n, d, s, r, K, m, t = 7001, 6, 300, 6, 40, 333, 4.0
    X, U0, U = make_case(n, d, s, r, seed=77)
    models = dict(kernel="lae", gl="cluster-normalized", root=True)
    H1 = api.heat_kernel_covariance_cpp(X[:m], X[m:], s, r, t, K, models, 1, 0.1, U=U)
    Hm = api.heat_kernel_covariance_cpp(X[:m], X[m:], s, r, t, K, models, 1, 0.1, U=U, devices=devices)
    assert Hm.shape == H1.shape == (n, m)
    assert np.abs(Hm - H1).max() <= H_RTOL * np.abs(H1).max(), np.abs(Hm - H1).max() / np.abs(H1).max()
    Ho = oracle.heat_kernel_covariance(X[:m], X[m:], U, r, t, K=K)
    assert np.abs(Hm - Ho).max() <= H_RTOL * np.abs(Ho).max()
    # the SE kernel and the plain random-walk Laplacian go through the same exchanges
    models = dict(kernel="se", gl="rw", root=False)
    H1 = api.heat_kernel_covariance_cpp(X[:m], X[m:], s, r, t, K, models, 1, 0.7, U=U)
    Hm = api.heat_kernel_covariance_cpp(X[:m], X[m:], s, r, t, K, models, 1, 0.7, U=U, devices=devices)
    assert np.abs(Hm - H1).max() <= H_RTOL * np.abs(H1).max()
    with pytest.raises(api.FlgpError):                                  # an error on the ranks comes back, nobody hangs
        api.heat_kernel_covariance_cpp(X[:m], X[m:], s, s + 1, t, K, models, 1, 0.7, U=U, devices=devices)


def test_one_bad_shard_fails_every_rank_together(oracle):
    """A failure on ONE rank of the sharded host entry must come back from EVERY rank, not hang the others in their next
    exchange (the reference, src/Spectrum.cpp:28-43, is one process and cannot hang).  (1) A NaN in rank 1's rows only:
    the ranks agree after the input check (flgp_comm_agree: one 16-byte all-reduce), rank 1 reports the NaN, rank 0 leaves
    with it.  (2) A rank that leaves without a word before the agreement (test hook multi_test_fail_rank: what a failed
    hipSetDevice looks like): the failing thread aborts every communicator and the peer that waits in the agreement
    returns.  (3) The library is usable afterwards and gives the clean answer."""
    import time
    n, d, s, r, K, m, t = 6000, 5, 200, 5, 30, 100, 4.0
    X, U0, U = make_case(n, d, s, r, seed=123)
    models = dict(kernel="lae", gl="cluster-normalized", root=True)
    Xbad = X.copy()
    Xbad[n - 7, 2] = np.nan                                  # in the second rank's block (rows [3000, 6000))
    t0 = time.time()
    with pytest.raises(api.FlgpError, match="NaN"):
        api.heat_kernel_covariance_cpp(Xbad[:m], Xbad[m:], s, r, t, K, models, 1, 0.1, U=U, devices=[0, 0])
    with pytest.raises(api.FlgpError, match="NaN"):
        api.heat_kernel_covariance_cpp(Xbad[:m], Xbad[m:], s, r, t, K, models, 1, 0.1, U=U, devices=[0, 0, 0])
    L = _lib.lib()
    for bad_rank in (0, 1):
        L.flgp_set_tuning(b"multi_test_fail_rank", bad_rank)
        try:
            with pytest.raises(api.FlgpError, match="injected failure"):
                api.heat_kernel_covariance_cpp(X[:m], X[m:], s, r, t, K, models, 1, 0.1, U=U, devices=[0, 0])
        finally:
            L.flgp_set_tuning(b"multi_test_fail_rank", -1)
    assert time.time() - t0 < 60.0
    H1 = api.heat_kernel_covariance_cpp(X[:m], X[m:], s, r, t, K, models, 1, 0.1, U=U)
    Hm = api.heat_kernel_covariance_cpp(X[:m], X[m:], s, r, t, K, models, 1, 0.1, U=U, devices=[0, 0])
    assert np.abs(Hm - H1).max() <= H_RTOL * np.abs(H1).max()


def test_c4_full_size_through_the_c_sharded_entry():
    """BASELINE configs[3] (C4) at n = 1e6 through the entry `bench.py --gpus N` drives -- flgp_dev_heat_kernel_covariance_sharded
    with a flgp_comm (here the in-process transport, two ranks = two host threads on the one card; on the 8-GPU node the same
    call sites run over RCCL) -- against the same entry on one rank: eigenvalues 1e-10, H on the training block and 8192
    sampled rows 1e-8 of max|H|.  (test_c4_full_size_row_sharded_two_ranks covers the Python driver over gloo.)"""
    from flgp_amd.pipeline import shard_bounds
    L = _lib.lib()
    n, d, s, r, K, m, t = 1_000_000, 16, 5000, 10, 200, 1000, 10.0
    rng = np.random.default_rng(4)
    sample = np.sort(np.concatenate([np.arange(1000), 1000 + rng.choice(n - 1000, 8192, replace=False)]))
    sel = np.sort(synth.random_anchor_rows(n, s))

    def run(world):
        comms = (ctypes.c_void_p * world)()
        _lib.check(L.flgp_comm_inproc_create(world, comms))

        def body(q):
            st = torch.cuda.current_stream().cuda_stream
            lo, hi = shard_bounds(n, world, q)
            Xh = synth.gaussian_mixture(hi - lo, d, row_offset=lo)
            dX = cm(Xh)
            mine = sel[(sel >= lo) & (sel < hi)] - lo
            Ul = cm(Xh[mine])
            U = torch.empty((d, s), dtype=torch.float64, device="cuda:0")
            _lib.check(L.flgp_dev_gather_anchors(st, comms[q], Ul.data_ptr(), len(mine), d, U.data_ptr(), s))
            sizes = torch.empty(s, dtype=torch.float64, device="cuda:0")
            _lib.check(L.flgp_dev_cluster_sizes(st, comms[q], dX.data_ptr(), hi - lo, hi - lo, d, U.data_ptr(), s, s, sizes.data_ptr()))
            H = torch.empty((m, hi - lo), dtype=torch.float64, device="cuda:0")
            vals = torch.empty(K, dtype=torch.float64, device="cuda:0")
            _lib.check(L.flgp_dev_heat_kernel_covariance_sharded(st, comms[q], dX.data_ptr(), hi - lo, hi - lo, d, n, lo, U.data_ptr(), s, s,
                                                                 sizes.data_ptr(), m, r, t, K, b"lae", b"cluster-normalized", 1, 0.1,
                                                                 H.data_ptr(), hi - lo, vals.data_ptr(), None, 0, None))
            loc = sample[(sample >= lo) & (sample < hi)] - lo
            return H[:, torch.from_numpy(loc).cuda()].T.contiguous().cpu().numpy(), vals.cpu().numpy()

        try:
            res = _run_ranks(world, body)
        finally:
            for q in range(world):
                L.flgp_comm_destroy(comms[q])
        return np.vstack([x[0] for x in res]), [x[1] for x in res]

    H1, v1 = run(1)
    torch.cuda.empty_cache()
    H2, v2 = run(2)
    assert H1.shape == H2.shape == (sample.size, m)
    assert np.abs(H1 - H2).max() <= H_RTOL * np.abs(H1).max(), np.abs(H1 - H2).max() / np.abs(H1).max()
    for q in range(2):
        np.testing.assert_allclose(v2[q], v1[0], rtol=EIG_RTOL)
    assert abs(v1[0][0] - 1.0) < 1e-6


class _CommTable(ctypes.Structure):
    """struct flgp_comm (include/flgp_hip.h): the table a transport fills in."""
    _fields_ = [("ctx", ctypes.c_void_p), ("rank", ctypes.c_int), ("world", ctypes.c_int),
                ("all_reduce_sum", ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)),
                ("all_gather", ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)),
                ("destroy", ctypes.CFUNCTYPE(None, ctypes.c_void_p)),
                ("abort", ctypes.CFUNCTYPE(None, ctypes.c_void_p))]


def test_rccl_backend_loads_and_runs_on_one_rank(stages):
    """The RCCL backend (librccl.so through dlopen) with a world of one -- all a one-GPU box can hold, RCCL refuses two
    ranks on one device.  The convenience wrappers return before the table when world <= 1, so the TABLE's entries are
    called directly here: ncclAllReduce / ncclAllGather execute through the restated prototypes of csrc/comm.hip (sum over
    one rank = identity, gather of one rank = copy), ncclCommAbort is loaded and installed as the abort hook (and used on a
    second communicator: collectives afterwards are refused, destroy does not touch the freed handle); then the sharded
    device entry point on top of the communicator, against the stage-by-stage driver."""
    L = _lib.lib()
    comms = (ctypes.c_void_p * 1)()
    dev = (ctypes.c_int * 1)(0)
    _lib.check(L.flgp_comm_rccl_init_all(1, dev, comms))
    idbuf = (ctypes.c_char * 128)()
    _lib.check(L.flgp_comm_rccl_unique_id(idbuf))
    try:
        st = torch.cuda.current_stream().cuda_stream
        tab = ctypes.cast(comms[0], ctypes.POINTER(_CommTable)).contents
        assert tab.rank == 0 and tab.world == 1 and tab.all_reduce_sum and tab.all_gather
        assert tab.abort, "ncclCommAbort was not found in librccl.so: a failing rank could not wake its peers"
        x = torch.arange(1000, dtype=torch.float64, device="cuda:0") * 0.25
        assert tab.all_reduce_sum(tab.ctx, x.data_ptr(), 1000, st) == 0, L.flgp_last_error()        # -> ncclAllReduce
        y = torch.full((1000,), -1.0, dtype=torch.float64, device="cuda:0")
        assert tab.all_gather(tab.ctx, x.data_ptr(), y.data_ptr(), 1000, st) == 0, L.flgp_last_error()   # -> ncclAllGather
        torch.cuda.synchronize()
        np.testing.assert_array_equal(x.cpu().numpy(), np.arange(1000.0) * 0.25)
        np.testing.assert_array_equal(y.cpu().numpy(), np.arange(1000.0) * 0.25)
        assert L.flgp_comm_agree(comms[0], 0, st) == 0 and L.flgp_comm_agree(comms[0], -2, st) == -2     # one rank: its own status
        # abort on a second communicator: refused afterwards, destroyed without touching the freed handle
        c2 = (ctypes.c_void_p * 1)()
        _lib.check(L.flgp_comm_rccl_init_all(1, dev, c2))
        t2 = ctypes.cast(c2[0], ctypes.POINTER(_CommTable)).contents
        L.flgp_comm_abort(c2[0])
        assert t2.all_reduce_sum(t2.ctx, x.data_ptr(), 1000, st) != 0
        L.flgp_comm_destroy(c2[0])
        _lib.check(L.flgp_comm_all_reduce_sum(comms[0], x.data_ptr(), 1000, st))
        _lib.check(L.flgp_comm_all_gather(comms[0], x.data_ptr(), y.data_ptr(), 1000, st))
        torch.cuda.synchronize()
        n, d, s, r, K, m, t = 5000, 4, 256, 5, 32, 200, 3.0
        X, U0, U = make_case(n, d, s, r, seed=5)
        dX = cm(X); dU = cm(U0); sizes = torch.from_numpy(U[:, d].copy()).cuda()
        H = torch.empty((m, n), dtype=torch.float64, device="cuda:0")
        vals = torch.empty(K, dtype=torch.float64, device="cuda:0")
        _lib.check(L.flgp_dev_heat_kernel_covariance_sharded(st, comms[0], dX.data_ptr(), n, n, d, n, 0, dU.data_ptr(), s, s,
                                                             sizes.data_ptr(), m, r, t, K, b"lae", b"cluster-normalized", 1, 0.1,
                                                             H.data_ptr(), n, vals.data_ptr(), None, 0, None))
        path = HeatKernelPath(stages)
        res = path.run(dX, dU, PathConfig(s=s, r=r, K=K, t=t, m=m), n, 0, num_class=sizes)
        np.testing.assert_allclose(vals.cpu().numpy(), res.values.cpu().numpy(), rtol=EIG_RTOL)
        Hp = res.H.cpu().numpy()
        assert np.abs(H.cpu().numpy() - Hp).max() <= H_RTOL * np.abs(Hp).max()
    finally:
        L.flgp_comm_destroy(comms[0])


def test_r_default_k_minus_one_past_4096_anchors(oracle):
    """heat_kernel_covariance_rcpp's default K = -1 (R/Fit.R:760) means K = s: truncated_SVD_cpp takes its dense BDCSVD
    branch (src/TruncatedSVD.cpp:17-20).  Here that is the full Jacobi decomposition of the s x s Gram matrix, which
    rounds 1-2 only built for s <= 4096; beyond it the panels are streamed through LDS (jac_stream_kernel).  s = 4500:
    every eigenvalue against LAPACK on the oracle's Gram matrix, V^T V = n I for all 4500 columns, and the covariance
    against the oracle's dense-SVD route."""
    n, d, s, r, m, t = 24000, 4, 4500, 4, 300, 3.0
    X, U0, U = make_case(n, d, s, r, seed=4500)
    models = dict(kernel="lae", gl="cluster-normalized", root=True)
    H = api.heat_kernel_covariance_rcpp(X[:m], X[m:], s, r, t, U=U)          # K = -1
    Ho = oracle.heat_kernel_covariance(X[:m], X[m:], U, r, t, K=-1)
    assert H.shape == Ho.shape == (n, m)
    assert np.abs(H - Ho).max() <= H_RTOL * np.abs(Ho).max(), np.abs(H - Ho).max() / np.abs(Ho).max()
    ep = api.heat_kernel_spectrum_cpp(X[:m], X[m:], s, r, -1, models, U=U)
    ei, zn = oracle.cross_similarity(X, U, r, gl="cluster-normalized")
    av, _ = oracle.scale_A(ei, zn, s)
    w = np.linalg.eigvalsh(oracle.gram(ei, av, s))[::-1]
    assert ep.values.shape == (s,) and w[-1] > 1e-8
    np.testing.assert_allclose(ep.values ** 2, w, rtol=1e-9, atol=0)
    VtV = ep.vectors.T @ ep.vectors / n
    assert np.abs(VtV - np.eye(s)).max() < 1e-7


@pytest.mark.parametrize("n,d,s,num_init,seed", [(3000, 3, 40, 1, 1), (5000, 16, 64, 2, 7), (2100, 2, 300, 1, 3),
                                                  (2000, 80, 50, 1, 5)])
def test_kmeans_minibatch(oracle, n, d, s, num_init, seed):
    """SURVEY 8f-4, second half: the mini-batch k-means of subsample_cpp's "minibatchkmeans" branch (src/Utils.cpp:49-62:
    ClusterR::MiniBatchKmeans with batch_size = 10 s, kmeans++ on 20 s of the rows, at most 100 iterations, early stop 10)
    on the device.  (1) Bit for bit the numpy restatement of the same algorithm on the same counter RNG -- centres, 1-NN sizes,
    iteration count, winning start.  (2) In distribution against the other methods, which is all that can be asked with
    respect to ClusterR itself (R's RNG): the within-SS is well below that of random rows and in the range of Lloyd's
    (k-means++ seeding + near-full batches can end BELOW Lloyd-from-random-rows, which stops in a poorer local minimum)."""
    X = synth.gaussian_mixture(n, d, components=6, seed=seed)
    U, info, wss = api.kmeans_minibatch(X, s, num_init=num_init, seed=seed)
    Uo, info_o, wss_o = oracle.np_kmeans_minibatch(X, s, num_init=num_init, seed=seed)
    assert tuple(info) == tuple(info_o)
    np.testing.assert_array_equal(U, Uo)
    assert abs(wss - wss_o) <= 1e-10 * wss_o
    C = np.asfortranarray(U[:, :d])
    sizes = np.bincount(oracle.knn(X, C, 1)[:, 0], minlength=s).astype(float)
    np.testing.assert_array_equal(U[:, d], sizes)                       # the sizes ARE the 1-NN counts (src/Utils.cpp:59-62)
    assert U[:, d].sum() == n
    rows = np.sort(synth.random_anchor_rows(n, s, seed=seed))
    R = np.asfortranarray(X[rows])
    wss_rand = float(((X - R[oracle.knn(X, R, 1)[:, 0]]) ** 2).sum())
    _, _, wss_lloyd = api.kmeans_lloyd(X, s, rows, iter_max=100)
    assert wss < 0.8 * wss_rand and wss < 2.0 * wss_lloyd, (wss_lloyd, wss, wss_rand)
    # another seed, other centres; the same seed, the same centres
    U2, _, _ = api.kmeans_minibatch(X, s, num_init=num_init, seed=seed + 1)
    assert not np.array_equal(U2, U)
    U3, _, _ = api.kmeans_minibatch(X, s, num_init=num_init, seed=seed)
    np.testing.assert_array_equal(U3, U)
