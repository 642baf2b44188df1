"""The R `.Call` shim (flgp_amd/csrc/rshim/flgp_rcall.c) EXECUTED against a functional test double of R's C API
(tests/r_mock/rmock.c): the boundary the reference exposes (src/RcppExports.cpp:311-499) -- argument unpacking, the
stacking of [X; X_new], subsampling call-backs into "R", dgRMatrix slot filling, PROTECT balance, the RNG-state bracket
on the error paths, FLGP_DEVICES routing -- with the HIP library behind it.  There is no R in the image; the double is
written from "Writing R Extensions" and stands in for libR only in this test."""
import ctypes
import glob
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
H_RTOL = 1e-8
P = ctypes.c_void_p
CB = ctypes.CFUNCTYPE(P, P)


class RMock:
    def __init__(self, so):
        self.L = L = ctypes.CDLL(so)
        for name, res, args in [
            ("rmock_reset", None, []), ("rmock_protect_depth", ctypes.c_int, []), ("rmock_protect_max", ctypes.c_int, []),
            ("rmock_rng_gets", ctypes.c_int, []), ("rmock_rng_puts", ctypes.c_int, []), ("rmock_last_error", ctypes.c_char_p, []),
            ("rmock_register_function", None, [ctypes.c_char_p, ctypes.c_char_p, CB]), ("rmock_arg", P, [P, ctypes.c_char_p, ctypes.c_int]),
            ("rmock_call", P, [P, ctypes.c_int, P]), ("rmock_lookup", P, [ctypes.c_char_p, ctypes.c_int]),
            ("rmock_dynamic_symbols", ctypes.c_int, []), ("rmock_real_matrix", P, [P, ctypes.c_int, ctypes.c_int]),
            ("rmock_real_vector", P, [P, ctypes.c_int]), ("rmock_int_vector", P, [P, ctypes.c_int]), ("rmock_logical", P, [ctypes.c_int]),
            ("rmock_named_list", P, [ctypes.c_int, P, P]), ("rmock_list_get", P, [P, ctypes.c_char_p]), ("rmock_slot", P, [P, ctypes.c_char_p]),
            ("rmock_class", ctypes.c_char_p, [P]), ("rmock_type", ctypes.c_int, [P]), ("rmock_data", P, [P]), ("rmock_len", ctypes.c_long, [P]),
            ("Rf_ScalarInteger", P, [ctypes.c_int]), ("Rf_ScalarReal", P, [ctypes.c_double]), ("Rf_mkString", P, [ctypes.c_char_p]),
            ("Rf_nrows", ctypes.c_int, [P]), ("Rf_ncols", ctypes.c_int, [P]), ("Rf_isMatrix", ctypes.c_int, [P]),
            ("R_init_FLGPhip", None, [P]),
        ]:
            f = getattr(L, name); f.restype = res; f.argtypes = args
        self._keep = []
        L.rmock_reset()
        L.R_init_FLGPhip(None)

    # ---- R objects from numpy / Python
    def mat(self, a):
        a = np.asfortranarray(a, dtype=np.float64)
        return self.L.rmock_real_matrix(a.ctypes.data, a.shape[0], a.shape[1])

    def vec(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        return self.L.rmock_real_vector(a.ctypes.data, a.size)

    def i(self, v): return self.L.Rf_ScalarInteger(int(v))
    def r(self, v): return self.L.Rf_ScalarReal(float(v))
    def s(self, v): return self.L.Rf_mkString(v.encode())
    def lgl(self, v): return self.L.rmock_logical(int(bool(v)))

    def lst(self, **kw):
        names = (ctypes.c_char_p * len(kw))(*[k.encode() for k in kw])
        vals = (P * len(kw))(*list(kw.values()))
        return self.L.rmock_named_list(len(kw), names, vals)

    # ---- back to numpy
    def real(self, sx):
        n = self.L.rmock_len(sx)
        a = np.ctypeslib.as_array(ctypes.cast(self.L.rmock_data(sx), ctypes.POINTER(ctypes.c_double)), shape=(n,)).copy()
        return a.reshape((self.L.Rf_nrows(sx), self.L.Rf_ncols(sx)), order="F") if self.L.Rf_isMatrix(sx) else a

    def ints(self, sx):
        n = self.L.rmock_len(sx)
        a = np.ctypeslib.as_array(ctypes.cast(self.L.rmock_data(sx), ctypes.POINTER(ctypes.c_int)), shape=(n,)).copy()
        return a.reshape((self.L.Rf_nrows(sx), self.L.Rf_ncols(sx)), order="F") if self.L.Rf_isMatrix(sx) else a

    def dgr(self, sx):
        assert self.L.rmock_class(sx) == b"dgRMatrix"
        return {k: (self.real if k == "x" else self.ints)(self.L.rmock_slot(sx, k.encode())) for k in ("p", "j", "x", "Dim")}

    def call(self, name, *args):
        """.Call("_FLGP_<name>", ...): looked up by name and arity in the table R_init registered; None = R error."""
        fn = self.L.rmock_lookup(("_FLGP_" + name).encode(), len(args))
        assert fn, "no routine _FLGP_%s with %d arguments is registered" % (name, len(args))
        arr = (P * len(args))(*args)
        before = self.L.rmock_protect_depth()
        res = self.L.rmock_call(fn, len(args), arr)
        assert self.L.rmock_protect_depth() == before, "PROTECT stack not balanced after _FLGP_%s" % name
        return res

    def error(self): return self.L.rmock_last_error().decode()

    def register(self, pkg, name, pyfn):
        cb = CB(pyfn)
        self._keep.append(cb)
        self.L.rmock_register_function(pkg.encode(), name.encode(), cb)


@pytest.fixture(scope="module")
def rmock(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("rmock") / "libflgp_rmock.so")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-Wall", "-Wextra", "-Werror", "-Wno-cast-function-type", "-o", out,
                    os.path.join(ROOT, "flgp_amd", "csrc", "rshim", "flgp_rcall.c"), os.path.join(ROOT, "tests", "r_mock", "rmock.c"),
                    "-I", os.path.join(ROOT, "tests", "r_mock"), "-I", os.path.join(ROOT, "include"),
                    "-L", os.path.join(ROOT, "flgp_amd"), "-lflgp_hip", "-Wl,-rpath," + os.path.join(ROOT, "flgp_amd")], check=True)
    return RMock(out)


EXPECT = {"lae_eigenmap": 7, "heat_kernel_covariance_cpp": 9, "cross_similarity_lae_cpp": 4, "subsample_cpp": 4, "KNN_cpp": 6,
          "LAE_cpp": 3, "local_anchor_embedding_cpp": 2, "v_to_z_cpp": 1}


def test_registration_by_name_and_arity(rmock):
    """R_init_FLGPhip registers the reference's names with the reference's arities (src/RcppExports.cpp:471-499) and turns
    dynamic lookup off (:502); a call with another arity finds nothing, as in R."""
    for name, n in EXPECT.items():
        assert rmock.L.rmock_lookup(("_FLGP_" + name).encode(), n)
        assert not rmock.L.rmock_lookup(("_FLGP_" + name).encode(), n + 1)
    assert rmock.L.rmock_dynamic_symbols() == 0


def test_error_paths_leave_rng_state_and_protect_stack_balanced(rmock):
    """No GPU needed: the failures happen before the first HIP call.  An unsupported subsample method (the reference:
    Rcpp::stop at src/Utils.cpp:64) leaves through Rf_error INSIDE the RNG bracket -- PutRNGstate must still run (Rcpp's
    RNGScope is unwound; here R_UnwindProtect) -- and a malformed FLGP_DEVICES is an error, not a silent single-GPU run."""
    X = np.random.default_rng(0).normal(size=(50, 3)); Xn = np.random.default_rng(1).normal(size=(20, 3))
    models = lambda sub: rmock.lst(subsample=rmock.s(sub), kernel=rmock.s("lae"), gl=rmock.s("rw"), root=rmock.lgl(False))
    g0, p0 = rmock.L.rmock_rng_gets(), rmock.L.rmock_rng_puts()
    res = rmock.call("heat_kernel_covariance_cpp", rmock.mat(X), rmock.mat(Xn), rmock.i(10), rmock.i(3), rmock.r(1.0), rmock.i(5),
                     models("bogus"), rmock.i(1), rmock.r(0.1))
    assert res is None and "subsample method is not supported" in rmock.error()
    assert rmock.L.rmock_rng_gets() == g0 + 1 and rmock.L.rmock_rng_puts() == p0 + 1
    res = rmock.call("subsample_cpp", rmock.mat(X), rmock.i(10), rmock.s("kmeans"), rmock.i(1))     # the "R function" is not there: an R-level error inside the call-back
    assert res is None and "could not find function" in rmock.error()
    assert rmock.L.rmock_rng_gets() == rmock.L.rmock_rng_puts()
    for bad in ("0,x", "0,,1", "-1", "1;2"):
        os.environ["FLGP_DEVICES"] = bad
        try:
            g1 = rmock.L.rmock_rng_gets()
            res = rmock.call("heat_kernel_covariance_cpp", rmock.mat(X), rmock.mat(Xn), rmock.i(10), rmock.i(3), rmock.r(1.0), rmock.i(5),
                             models("random"), rmock.i(1), rmock.r(0.1))
            assert res is None and "FLGP_DEVICES" in rmock.error(), bad
            assert rmock.L.rmock_rng_gets() == g1, "nothing may be drawn before the device list is accepted"
        finally:
            del os.environ["FLGP_DEVICES"]
    res = rmock.call("heat_kernel_covariance_cpp", rmock.mat(X), rmock.mat(np.zeros((5, 4))), rmock.i(10), rmock.i(3), rmock.r(1.0),
                     rmock.i(5), models("random"), rmock.i(1), rmock.r(0.1))
    assert res is None and "same number of columns" in rmock.error()
    res = rmock.call("KNN_cpp", rmock.vec(np.zeros(4)), rmock.mat(X), rmock.i(2), rmock.s("Euclidean"), rmock.lgl(False), rmock.i(100))
    assert res is None and "numeric matrix" in rmock.error()


def _kmeans_callback(rmock, U, d):
    """stats::kmeans(x, centers, iter.max, nstart) -> list(centers, size) as the fixture has them (src/Utils.cpp:37-44)."""
    def fn(args):
        assert rmock.L.rmock_arg(args, b"x", -1) and rmock.L.rmock_arg(args, b"nstart", -1)
        s = rmock.ints(rmock.L.rmock_arg(args, b"centers", -1))[0]
        assert s == U.shape[0] and rmock.ints(rmock.L.rmock_arg(args, b"iter.max", -1))[0] == 100
        return rmock.lst(centers=rmock.mat(U[:, :d]), size=rmock.vec(U[:, d]))
    return fn


@pytest.mark.gpu
@pytest.mark.parametrize("path", sorted(p for p in glob.glob(os.path.join(GOLDEN, "*.npz")) if "known" not in p))
def test_golden_fixtures_through_the_call_shim(rmock, path):
    """.Call("_FLGP_KNN_cpp", ..., output = TRUE), _FLGP_LAE_cpp, _FLGP_cross_similarity_lae_cpp and
    _FLGP_heat_kernel_covariance_cpp on the golden fixtures: indices and dgRMatrix slots exact, H within 1e-8, the PROTECT
    stack back where it was after every call (RMock.call asserts it), one GetRNGstate / PutRNGstate pair per call that
    subsamples.  The anchors reach heat_kernel_covariance_cpp as in the reference: subsample_cpp calls stats::kmeans back
    (here a registered call-back that returns the fixture's centres and sizes) or base::sample."""
    g = np.load(path)
    X, U = g["X"], g["U"]
    n, d = X.shape; s = U.shape[0]; r = int(g["r"]); K = int(g["K"]); t = float(g["t"]); m = int(g["m"])
    gl = str(g["gl"]); root = bool(g["root"])
    U0 = np.asfortranarray(U[:, :d])
    # ---- KNN_cpp with distances (src/RcppExports.cpp:375-388)
    res = rmock.call("KNN_cpp", rmock.mat(X), rmock.mat(U0), rmock.i(r), rmock.s("Euclidean"), rmock.lgl(True), rmock.i(100))
    assert res, rmock.error()
    ind = rmock.ints(rmock.L.rmock_list_get(res, b"ind_knn"))
    np.testing.assert_array_equal(ind, g["knn_idx"])
    D = rmock.dgr(rmock.L.rmock_list_get(res, b"distances_sp"))
    np.testing.assert_array_equal(D["Dim"], [n, s])
    np.testing.assert_array_equal(D["p"], np.arange(n + 1) * r)
    order = np.argsort(g["knn_idx"], axis=1, kind="stable")
    np.testing.assert_array_equal(D["j"].reshape(n, r), np.take_along_axis(g["knn_idx"], order, 1))
    if "knn_dist" in g.files:
        np.testing.assert_array_equal(D["x"].reshape(n, r), np.take_along_axis(g["knn_dist"], order, 1))
    else:       # the distances of the listed neighbours, association of src/Utils.cpp:121 up to the oracle's chain order
        dd = ((X[:, None, :] - U0[D["j"].reshape(n, r)]) ** 2).sum(-1)
        np.testing.assert_allclose(D["x"].reshape(n, r), dd, rtol=1e-9, atol=1e-12)
    res1 = rmock.call("KNN_cpp", rmock.mat(X), rmock.mat(U0), rmock.i(r), rmock.s("Euclidean"), rmock.lgl(False), rmock.i(100))
    assert rmock.L.rmock_len(res1) == 1 and not rmock.L.rmock_list_get(res1, b"distances_sp")
    # ---- LAE_cpp / cross_similarity_lae_cpp -> dgRMatrix (src/RcppExports.cpp:421-431, :347-358)
    Z = rmock.dgr(rmock.call("LAE_cpp", rmock.mat(X), rmock.mat(U0), rmock.i(r)))
    np.testing.assert_array_equal(Z["j"].reshape(n, r), g["ell_idx"])
    np.testing.assert_array_equal(Z["x"].reshape(n, r), g["lae_val"])
    np.testing.assert_array_equal(Z["p"], np.arange(n + 1) * r)
    Z = rmock.dgr(rmock.call("cross_similarity_lae_cpp", rmock.mat(X), rmock.mat(U), rmock.i(r), rmock.s(gl)))
    np.testing.assert_array_equal(Z["Dim"], [n, s])
    np.testing.assert_array_equal(Z["j"].reshape(n, r), g["ell_idx"])
    np.testing.assert_array_equal(Z["x"].reshape(n, r), g["z_val"])
    # ---- heat_kernel_covariance_cpp (src/RcppExports.cpp:328-344): X and X_new separately, anchors through the call-back
    has_sizes = U.shape[1] == d + 1
    if has_sizes:
        rmock.register("stats", "kmeans", _kmeans_callback(rmock, U, d))
        sub = "kmeans"
    else:
        rows = np.array([int(np.where((X == U0[a]).all(1))[0][0]) for a in range(s)], dtype=np.int32) if "rows" not in g.files else g["rows"]
        rmock.register("base", "sample", lambda args: rmock.L.rmock_int_vector((rows + 1).astype(np.int32).ctypes.data, s))
        sub = "random"
    models = rmock.lst(subsample=rmock.s(sub), kernel=rmock.s("lae"), gl=rmock.s(gl), root=rmock.lgl(root))
    g0, p0 = rmock.L.rmock_rng_gets(), rmock.L.rmock_rng_puts()
    Hs = rmock.call("heat_kernel_covariance_cpp", rmock.mat(X[:m]), rmock.mat(X[m:]), rmock.i(s), rmock.i(r), rmock.r(t), rmock.i(K),
                    models, rmock.i(1), rmock.r(0.1))
    assert Hs, rmock.error()
    assert (rmock.L.rmock_rng_gets(), rmock.L.rmock_rng_puts()) == (g0 + 1, p0 + 1)
    H = rmock.real(Hs)
    assert H.shape == g["H"].shape == (n, m)
    assert np.abs(H - g["H"]).max() <= H_RTOL * np.abs(g["H"]).max()
    # ---- FLGP_DEVICES: "0" = that GPU through the multi entry's one-device road; "0,0" = two ranks (in-process transport)
    for devs in ("0", "0,0"):
        os.environ["FLGP_DEVICES"] = devs
        try:
            Hd = rmock.call("heat_kernel_covariance_cpp", rmock.mat(X[:m]), rmock.mat(X[m:]), rmock.i(s), rmock.i(r), rmock.r(t), rmock.i(K),
                            models, rmock.i(1), rmock.r(0.1))
            assert Hd, rmock.error()
            Hd = rmock.real(Hd)
        finally:
            del os.environ["FLGP_DEVICES"]
        assert np.abs(Hd - g["H"]).max() <= H_RTOL * np.abs(g["H"]).max(), devs
    rmock.L.rmock_reset(); rmock.L.R_init_FLGPhip(None)


@pytest.mark.gpu
def test_small_entry_points_through_the_call_shim(rmock):
    """v_to_z_cpp / local_anchor_embedding_cpp (1 x r matrices, as Rcpp wraps an Eigen::RowVectorXd), subsample_cpp with the
    device methods and lae_eigenmap's named list."""
    z = rmock.real(rmock.call("v_to_z_cpp", rmock.vec([0.9, 0.3, -1.0])))
    assert z.shape == (1, 3)
    np.testing.assert_allclose(z[0], [0.8, 0.2, 0.0], atol=1e-15)
    U = np.array([[0.0, 0.0], [2.0, 0.0], [0.0, 2.0]])
    z = rmock.real(rmock.call("local_anchor_embedding_cpp", rmock.vec([1.0, 0.0]), rmock.mat(U)))
    assert z.shape == (1, 3) and abs(z.sum() - 1.0) < 1e-12 and z.min() >= 0.0
    from oracle import flgp_oracle as O          # (the iteration stops on a SQUARED step of 1e-5, src/lae.cpp:82-86: the midpoint is met to ~3e-3)
    np.testing.assert_array_equal(z[0], np.ravel(O.local_anchor_embedding(np.array([1.0, 0.0]), np.asfortranarray(U))))
    np.testing.assert_allclose(z[0], [0.5, 0.5, 0.0], atol=1e-2)
    from flgp_amd import synth
    X, _ = synth.swiss_roll(2000, seed=3)
    Us = rmock.call("subsample_cpp", rmock.mat(X), rmock.i(40), rmock.s("minibatch"), rmock.i(1))
    assert Us, rmock.error()
    Us = rmock.real(Us)
    assert Us.shape == (40, 4) and Us[:, 3].sum() == 2000
    assert rmock.L.rmock_rng_gets() == rmock.L.rmock_rng_puts()
    rmock.register("base", "sample", lambda args: rmock.L.rmock_int_vector(np.arange(1, 41, dtype=np.int32).ctypes.data, 40))
    em = rmock.call("lae_eigenmap", rmock.mat(X), rmock.i(40), rmock.i(3), rmock.i(4), rmock.s("lloyd"), rmock.s("cluster-normalized"), rmock.i(1))
    assert em, rmock.error()
    ev = rmock.real(rmock.L.rmock_list_get(em, b"eigenvalues")); vec = rmock.real(rmock.L.rmock_list_get(em, b"eigenvectors"))
    assert ev.shape == (4,) and vec.shape == (2000, 4) and abs(ev[0]) < 1e-8 and np.all(np.diff(ev) >= -1e-12)
    rmock.L.rmock_reset(); rmock.L.R_init_FLGPhip(None)
