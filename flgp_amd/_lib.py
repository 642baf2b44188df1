"""ctypes binding of libflgp_hip.so (include/flgp_hip.h).

There is no CPU fallback anywhere in this package: if the HIP library is missing or does
not load, importing the binding raises.  Build it with ``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C flgp_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libflgp_hip.so")

c_int, c_double, c_long, c_void_p, c_char_p, c_size_t = (
    ctypes.c_int, ctypes.c_double, ctypes.c_long, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t)

# name -> (restype, argtypes); pointers are passed as integers/void* so that both numpy
# (host entry points) and torch data_ptr() (device entry points) can feed them.
P = c_void_p
_SIGNATURES = {
    "flgp_last_error": (c_char_p, []),
    "flgp_version": (c_char_p, []),
    "flgp_device_count": (c_int, []),
    "flgp_dev_pool_release": (c_size_t, []),
    "flgp_release_pinned": (None, []),
    "flgp_set_device": (c_int, [c_int]),
    "flgp_parse_gl": (c_int, [c_char_p]),
    "flgp_set_tuning": (c_int, [c_char_p, c_int]),
    "flgp_prof_enable": (None, [c_int]),
    "flgp_prof_reset": (None, []),
    "flgp_prof_query": (c_int, [c_char_p, P, P, P]),
    "flgp_prof_names": (c_int, [P, c_int]),
    # host-pointer entry points
    "flgp_knn": (c_int, [P, c_int, c_int, P, c_int, c_int, c_char_p, P, P]),
    "flgp_v_to_z": (c_int, [P, c_int, P]),
    "flgp_local_anchor_embedding": (c_int, [P, c_int, P, c_int, P]),
    "flgp_lae": (c_int, [P, c_int, c_int, P, c_int, c_int, P, P, P]),
    "flgp_cross_similarity_lae": (c_int, [P, c_int, c_int, P, c_int, c_int, c_int, c_char_p, P, P, P]),
    "flgp_cross_similarity_se": (c_int, [P, c_int, c_int, P, c_int, c_int, c_int, c_char_p, c_double, P, P, P]),
    "flgp_graph_laplacian": (c_int, [P, P, c_int, c_int, c_int, c_char_p, P]),
    "flgp_spectrum_from_Z": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P, P]),
    "flgp_hk_from_spectrum": (c_int, [P, P, c_int, c_int, c_double, P, c_int, P, c_int, P]),
    "flgp_heat_kernel_spectrum": (c_int, [P, c_int, c_int, P, c_int, c_int, c_int, c_int, c_char_p, c_char_p,
                                          c_int, c_double, P, P]),
    "flgp_eigenpair_from_host": (c_int, [P, P, c_int, c_int, P]),
    "flgp_heat_kernel_spectrum_resident": (c_int, [P, c_int, c_int, P, c_int, c_int, c_int, c_int, c_char_p, c_char_p,
                                                   c_int, c_double, P]),
    "flgp_eigenpair_dims": (c_int, [P, P, P]),
    "flgp_eigenpair_to_host": (c_int, [P, P, P]),
    "flgp_hk_from_eigenpair": (c_int, [P, c_int, c_double, P, c_int, P, c_int, P]),
    "flgp_eigenpair_vtv": (c_int, [P, c_int, P, c_int, P]),
    "flgp_eigenpair_vty": (c_int, [P, c_int, P, c_int, P, c_int, P]),
    "flgp_eigenpair_vc": (c_int, [P, c_int, P, c_int, P, c_int, P]),
    "flgp_eigenpair_predict_regression": (c_int, [P, c_int, P, c_int, P, c_int, P, c_int, c_double, c_double, c_double, P]),
    "flgp_eigenpair_predict_regression_different": (c_int, [P, c_int, P, c_int, P, c_int, P, c_int, c_double, P, c_double, P]),
    "flgp_eigenpair_posterior_variance": (c_int, [P, c_int, P, c_int, P, c_int, c_double, c_double, c_double, P]),
    "flgp_eigenpair_free": (None, [P]),
    "flgp_kmeans_minibatch": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, c_double, c_int, ctypes.c_ulonglong, P, P, P]),
    "flgp_kmeans_lloyd": (c_int, [P, c_int, c_int, c_int, P, c_int, c_int, P, P, P]),
    "flgp_nystrom_eigenpair": (c_int, [P, c_int, c_int, P, c_int, c_double, c_int, P, P]),
    "flgp_nystrom_eigenpair_resident": (c_int, [P, c_int, c_int, P, c_int, c_double, c_int, P]),
    "flgp_heat_kernel_covariance": (c_int, [P, c_int, c_int, c_int, P, c_int, c_int, c_int, c_double, c_int,
                                            c_char_p, c_char_p, c_int, c_double, P]),
    "flgp_se_spectrum_grid": (c_int, [P, c_int, c_int, P, c_int, c_int, c_int, c_int, P, c_int, c_char_p, c_int, P, P, P, c_int]),
    "flgp_lae_eigenmap": (c_int, [P, c_int, c_int, P, c_int, c_int, c_int, c_int, c_char_p, P, P]),
    "flgp_dev_se_spectrum_grid": (c_int, [P, P, c_int, c_int, c_int, P, c_int, c_int, P, c_int, c_int, P, c_int, c_char_p, c_int, P, P, P,
                                          c_int, P]),
    # device-pointer stage entry points
    "flgp_dev_anchor_dpad": (c_int, [c_int]),
    "flgp_dev_anchor_rows": (c_int, [c_int]),
    "flgp_dev_anchor_prep": (c_int, [P, P, c_int, c_int, c_int, P, P]),
    "flgp_dev_nystrom_eigenpair": (c_int, [P, P, c_int, c_int, c_int, P, c_int, c_int, c_double, c_int, P, P, c_int]),
    "flgp_dev_kmeans_lloyd": (c_int, [P, P, c_int, c_int, c_int, c_int, P, c_int, P, c_int, P, P]),
    "flgp_dev_kmeans_init": (c_int, [P, P, c_int, c_int, c_int, P, c_int, P, c_int]),
    "flgp_dev_knn": (c_int, [P, P, c_int, c_int, c_int, P, P, c_int, c_int, P, P, c_int]),
    "flgp_dev_lae": (c_int, [P, P, c_int, c_int, c_int, P, c_int, c_int, P, c_int, P, P]),
    "flgp_dev_v_to_z": (c_int, [P, P, c_int, P]),
    "flgp_dev_se_weights_den": (c_int, [P, P, P, c_int, c_int, c_int, c_double, P, P]),
    "flgp_dev_mean": (c_int, [P, P, c_long, P, P]),
    "flgp_dev_se_weights": (c_int, [P, P, P, c_int, c_int, c_int, c_double, P, P]),
    "flgp_dev_csc_workspace": (c_size_t, [c_int, c_int, c_int]),
    "flgp_dev_csc_build": (c_int, [P, P, c_int, c_int, c_int, P, P, P, c_size_t]),
    "flgp_dev_colsum_workspace": (c_size_t, [c_int, c_int]),
    "flgp_dev_colsum": (c_int, [P, P, P, c_int, c_int, c_int, P, P, c_size_t]),
    "flgp_dev_col_scale": (c_int, [P, P, P, c_int, c_int, P, P, c_int]),
    "flgp_dev_row_normalize": (c_int, [P, P, c_int, c_int]),
    "flgp_dev_col_scale_row_normalize": (c_int, [P, P, P, c_int, c_int, P, P]),
    "flgp_dev_gram": (c_int, [P, P, P, c_int, c_int, c_int, P, P, P, c_int]),
    "flgp_dev_sym_pack": (c_int, [P, P, c_int, c_int, P]),
    "flgp_dev_sym_unpack": (c_int, [P, P, c_int, P, c_int]),
    "flgp_dev_eig_workspace": (c_size_t, [c_int, c_int]),
    "flgp_dev_eig_topk": (c_int, [P, P, c_int, c_int, c_int, c_double, P, P, c_int, P, c_size_t, P]),
    "flgp_dev_bsg_workspace": (c_size_t, [c_int, c_int]),
    "flgp_dev_bsg_set_trace": (None, [P]),
    "flgp_dev_jac_set_trace": (None, [P]),
    "flgp_dev_bsg_apply": (c_int, [P, P, c_int, c_int, P, c_int, c_double, c_double, P, P, P, c_size_t, P]),
    "flgp_dev_u_recover_workspace": (c_size_t, [c_int, c_int]),
    "flgp_dev_spectrum_usable": (c_int, [P, P, c_int]),
    "flgp_dev_spectrum_usable_route": (c_int, [P, P, c_int, c_int]),
    "flgp_dev_u_recover": (c_int, [P, P, P, c_int, c_int, P, c_int, c_int, P, c_int, c_double, c_int, P, c_int, P, P]),
    "flgp_dev_hk": (c_int, [P, P, c_int, c_double, P, c_int, P, c_int, c_int, P, c_int, P, c_int, c_int,
                            P, c_int, P]),
    "flgp_dev_gemm": (c_int, [P, c_int, c_int, c_int, c_double, P, c_long, c_long, P, c_long, c_long,
                              c_double, P, c_long, c_long, P, c_long, c_long, P, c_size_t]),
    "flgp_dev_rotate": (c_int, [P, c_int, c_int, c_double, P, P, P, c_double, P, P, P]),
    "flgp_dev_gram_small": (c_int, [P, c_int, c_int, P, P, P, P, c_size_t]),
    "flgp_dev_gemm_pair": (c_int, [P, c_int, c_int, c_int, c_double, P, P, c_long, c_long, P, P, c_long, c_long, P, P, c_long, c_long]),
    "flgp_dev_gather_rows": (c_int, [P, P, c_int, P, c_int, c_int, P]),
    # row-sharded path behind the C ABI: communicators + sharded entry points
    "flgp_comm_inproc_create": (c_int, [c_int, P]),
    "flgp_comm_rccl_unique_id": (c_int, [P]),
    "flgp_comm_rccl_init_rank": (c_int, [c_int, c_int, P, P]),
    "flgp_comm_rccl_init_all": (c_int, [c_int, P, P]),
    "flgp_comm_destroy": (None, [P]),
    "flgp_comm_all_reduce_sum": (c_int, [P, P, c_size_t, P]),
    "flgp_comm_all_gather": (c_int, [P, P, P, c_size_t, P]),
    "flgp_comm_rank": (c_int, [P]),
    "flgp_comm_world": (c_int, [P]),
    "flgp_dev_gather_anchors": (c_int, [P, P, P, c_int, c_int, P, c_int]),
    "flgp_dev_cluster_sizes": (c_int, [P, P, P, c_int, c_int, c_int, P, c_int, c_int, P]),
    "flgp_dev_heat_kernel_covariance_sharded": (c_int, [P, P, P, c_int, c_int, c_int, c_long, c_long, P, c_int, c_int, P, c_int,
                                                        c_int, c_double, c_int, c_char_p, c_char_p, c_int, c_double, P, c_int,
                                                        P, P, c_int, P]),
    "flgp_heat_kernel_covariance_multi": (c_int, [P, c_int, c_int, c_int, P, c_int, c_int, c_int, c_double, c_int, c_char_p,
                                                  c_char_p, c_int, c_double, c_int, P, P]),
    "flgp_heat_kernel_covariance_rank": (c_int, [P, P, c_long, c_int, c_long, c_long, c_int, c_int, P, c_int, c_int, c_int, c_double,
                                                 c_int, c_char_p, c_char_p, c_int, c_double, P, c_long, P]),
    "flgp_comm_abort": (None, [P]),
    "flgp_comm_agree": (c_int, [P, c_int, P]),
    "flgp_dev_hk_workspace": (c_size_t, [c_int, c_int, c_int, c_int]),
}


class FlgpError(RuntimeError):
    """An error reported by libflgp_hip.so (the R shim raises the same text via Rf_error)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"[flgp {code}] {message}")
        self.code = code
        self.message = message


def declared_symbols():
    return sorted(_SIGNATURES)


_lib = None


def lib():
    """Load libflgp_hip.so (once).  Raises OSError if it is not built: no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(
                f"{LIB_PATH} is missing: the HIP library is the only implementation of this path "
                "(build it with `make -C flgp_amd/csrc` or __graft_entry__.build())")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int):
    if rc != 0:
        raise FlgpError(rc, lib().flgp_last_error().decode("utf-8", "replace"))
    return rc
