"""CPU: the C-ABI library loads and exports every symbol include/flgp_hip.h declares; argument
validation that happens before any device work mirrors the reference's error behaviour.
No compute call is made here (no GPU in the build container)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from flgp_amd import _lib, api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "flgp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(flgp_[A-Za-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [name for name in header_symbols() if not hasattr(L, name)]
    assert not missing, f"declared in flgp_hip.h but not exported: {missing}"


def test_binding_table_matches_header():
    assert sorted(_lib.declared_symbols()) == header_symbols()
    _lib.lib()  # binds every symbol with its signature; raises on a missing one


def test_signatures_have_no_torch_or_cpp_types():
    text = open(os.path.join(ROOT, "include", "flgp_hip.h")).read()
    assert 'extern "C"' in text
    for bad in ("torch", "at::", "std::", "Eigen", "Rcpp", "hipStream_t"):
        assert bad not in re.sub(r"/\*.*?\*/", "", text, flags=re.S), bad


def test_parse_gl_and_unsupported_strings():
    L = _lib.lib()
    assert L.flgp_parse_gl(b"rw") == 0 and L.flgp_parse_gl(b"normalized") == 1
    assert L.flgp_parse_gl(b"cluster-normalized") == 2
    assert L.flgp_parse_gl(b"nope") == -3
    # the reference's message (src/Utils.cpp:207)
    assert b"graph Laplacian is not supported" in L.flgp_last_error()


def test_unsupported_distance_is_the_reference_error():
    X = np.zeros((4, 2)); U = np.zeros((3, 2))
    with pytest.raises(api.FlgpError) as e:
        api.KNN_cpp(X, U, 2, distance="geodesic")
    assert e.value.code == -3 and "distance method of KNN is not supported" in e.value.message


def test_unsupported_kernel_and_gl_raise_before_device_work():
    X = np.zeros((4, 2)); U = np.zeros((3, 3))
    with pytest.raises(api.FlgpError) as e:
        api.heat_kernel_covariance_rcpp(X[:2], X[2:], 3, 2, 1.0, models=dict(kernel="rbf"), U=U)
    assert e.value.code == -3 and "kernel type is not supported" in e.value.message
    with pytest.raises(api.FlgpError) as e:
        api.cross_similarity_lae_cpp(X, U, 2, gl="bogus")
    assert e.value.code == -3


def test_subsample_contract():
    X = np.arange(20.0).reshape(10, 2)
    U = api.subsample_cpp(X, 4, "random", rng=np.random.default_rng(1))
    assert U.shape == (4, 2) and len({tuple(u) for u in U}) == 4
    with pytest.raises(NotImplementedError):
        api.subsample_cpp(X, 4, "kmeans")
    with pytest.raises(api.FlgpError):
        api.subsample_cpp(X, 4, "bogus")


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_means_a_loud_error_not_a_fallback():
    X = np.random.default_rng(0).normal(size=(8, 2)); U = X[:3].copy()
    with pytest.raises(api.FlgpError) as e:
        api.KNN_cpp(X, U, 2)
    assert e.value.code == -4   # FLGP_ERR_HIP: there is no CPU path behind the ABI
    from flgp_amd.pipeline import HipStages
    with pytest.raises(RuntimeError):
        HipStages("cpu")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "flgp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
                assert "flgp_oracle" not in src or f.endswith((".hip", ".h")) and "oracle/flgp_oracle.c" in src, f


def test_r_shim_type_checks_and_registers_the_reference_symbols():
    """The .Call shim cannot run here (no R), but it must at least type-check against the C ABI
    header and register the reference's symbol names with the reference's arities
    (src/RcppExports.cpp:471-499)."""
    import subprocess
    shim = os.path.join(ROOT, "flgp_amd", "csrc", "rshim", "flgp_rcall.c")
    subprocess.run(["gcc", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-cast-function-type", "-I", os.path.join(ROOT, "tests", "r_mock"),
                    "-I", os.path.join(ROOT, "include"), shim], check=True)
    text = open(shim).read()
    expect = {"_FLGP_lae_eigenmap": 7, "_FLGP_heat_kernel_covariance_cpp": 9, "_FLGP_cross_similarity_lae_cpp": 4,
              "_FLGP_subsample_cpp": 4, "_FLGP_KNN_cpp": 6, "_FLGP_LAE_cpp": 3,
              "_FLGP_local_anchor_embedding_cpp": 2, "_FLGP_v_to_z_cpp": 1}
    got = {m.group(1): int(m.group(2)) for m in re.finditer(r'\{"(_FLGP_\w+)",\s*\(DL_FUNC\)&\w+,\s*(\d+)\}', text)}
    assert got == expect


def test_header_states_the_eig_info_contract():
    """flgp_dev_eig_topk writes FOUR ints of `info` (round-1 advisor: the header promised two and a C caller with
    int info[2] got its stack overwritten)."""
    text = open(os.path.join(ROOT, "include", "flgp_hip.h")).read()
    doc = text[text.index("size_t flgp_dev_eig_workspace") - 900:text.index("size_t flgp_dev_eig_workspace")]
    assert "FOUR ints" in doc and "[3]" in doc
    src = open(os.path.join(ROOT, "flgp_amd", "csrc", "eig.hip")).read()
    assert "info[4]" not in src and "info[3]" in src
