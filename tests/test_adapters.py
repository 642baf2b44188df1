"""flgp_amd/csrc/rshim/flgp_cpp_adapters.cpp: the reference's INTERNAL C++ entry points (the ones src/Fit.cpp,
src/train.cpp and src/Predict.cpp call, with Eigen / Rcpp types) forwarding to the C ABI -- what makes the fit_* drivers
drop-in without editing them.  Compiled against a functional test double of Eigen / Rcpp (tests/eigen_mock/) and, with a
GPU, RUN through the drivers' own call sequences (tests/c/adapters_check.cpp) on a golden fixture."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "tests", "c", "adapters_check.cpp"), os.path.join(ROOT, "flgp_amd", "csrc", "rshim", "flgp_cpp_adapters.cpp")]
FLAGS = ["-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-DFLGP_ADAPTERS_TEST", "-I", os.path.join(ROOT, "tests", "eigen_mock"),
         "-I", os.path.join(ROOT, "include")]
H_RTOL, EIG_RTOL = 1e-8, 1e-10


def test_adapters_compile_with_the_reference_signatures():
    """Type check against the restated declarations of src/Spectrum.h:45-124, src/lae.h:34-60, src/Utils.h:35-62
    (tests/eigen_mock/ref_decls.h): a definition whose signature drifted from the declaration would be an overload, and the
    check program's calls would fail to link."""
    subprocess.run(["g++", "-fsyntax-only"] + FLAGS + SRC, check=True)
    text = open(SRC[1]).read()
    for name in ["KNN_cpp", "LAE_cpp", "graphLaplacian_cpp", "spectrum_from_Z_cpp", "HK_from_spectrum_cpp", "heat_kernel_spectrum_cpp",
                 "heat_kernel_covariance_cpp", "cross_similarity_lae_cpp", "cross_similarity_se_cpp", "local_anchor_embedding_cpp",
                 "v_to_z_cpp", "lae_eigenmap"]:
        assert name + "(" in text, name


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", ["d16_r10.npz", "c1_like.npz"])
def test_fit_driver_call_sequences_through_the_adapters(tmp_path, fixture):
    from oracle import flgp_oracle as O
    exe = str(tmp_path / "adapters_check")
    subprocess.run(["g++"] + FLAGS + ["-o", exe] + SRC + ["-L", os.path.join(ROOT, "flgp_amd"), "-lflgp_hip",
                                                        "-Wl,-rpath," + os.path.join(ROOT, "flgp_amd")], check=True)
    g = np.load(os.path.join(ROOT, "tests", "golden", fixture))
    X, U = np.asfortranarray(g["X"]), np.asfortranarray(g["U"])
    n, d = X.shape; s, ucols = U.shape
    r, K, m, t = int(g["r"]), int(g["K"]), int(g["m"]), float(g["t"])
    gl, root = str(g["gl"]), bool(g["root"])
    inp, out = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as f:
        f.write(struct.pack("8i", n, d, s, r, K, m, int(root), ucols)); f.write(struct.pack("d", t)); f.write(gl.encode().ljust(32, b"\0"))
        f.write(X.tobytes(order="F")); f.write(U.tobytes(order="F"))
    res = subprocess.run([exe, inp, out], capture_output=True, text=True)
    assert res.returncode == 0 and "adapters_check ok" in res.stdout, res.stderr
    buf = np.fromfile(out)
    pos = 0
    def take(*shape):
        nonlocal pos
        cnt = int(np.prod(shape)); a = buf[pos:pos + cnt].reshape(shape, order="F"); pos += cnt
        return a
    vals, Cvv, Cnv, H, ev = take(K), take(m, m), take(n - m, m), take(n, m), take(3)
    Hg = g["H"]
    np.testing.assert_allclose(vals, g["values"], rtol=EIG_RTOL)
    assert np.abs(H - Hg).max() <= H_RTOL * np.abs(Hg).max()
    assert np.abs(Cvv - Hg[:m]).max() <= H_RTOL * np.abs(Hg).max() and np.abs(Cnv - Hg[m:]).max() <= H_RTOL * np.abs(Hg).max()
    vo, _ = O.heat_kernel_spectrum(X, U, r, 3, gl=gl, root=True)
    np.testing.assert_allclose(ev, 1 - vo, rtol=0, atol=1e-10)
    # the fit_se_* sequence against the oracle's restatement of the same lines (src/Fit.cpp:127-158)
    U0 = np.asfortranarray(U[:, :d])
    kidx, kdist = O.knn(X, U0, r, output=True)
    for a2 in (0.5, 2.0):
        ei, ev_ = O.se_weights(kidx, kdist, np.sqrt(a2 * kdist.mean() / 4.0))          # exp(-d / (4 eps^2)) with 4 eps^2 = a2 mean(d)
        zn = O.graph_laplacian(ei, ev_, s, gl, U[:, d] if ucols > d else None)
        vs, vec = O.spectrum_from_Z(ei, zn, s, K, root=root)
        Co = O.hk_from_spectrum(vs, vec, K, t, np.arange(m, dtype=np.int32), np.arange(m, dtype=np.int32))
        vals_s, Cs = take(K), take(m, m)
        np.testing.assert_allclose(vals_s, vs, rtol=1e-9)
        assert np.abs(Cs - Co).max() <= H_RTOL * np.abs(Co).max()
    assert pos == buf.size
