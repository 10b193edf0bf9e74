"""ctypes binding of libmused_hip.so (the C ABI declared in include/mused_hip.h).

There is no CPU fallback: if the shared library is missing this module raises, and every
op built on it fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

F32, F64, I64, BITS = 0, 1, 2, 3
METRIC_L2, METRIC_COSINE = 0, 1

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmused_hip.so")

_lib = None


class MusedError(RuntimeError):
    pass


_vp, _i, _l, _d = C.c_void_p, C.c_int, C.c_long, C.c_double

# name -> (restype, argtypes); mirrors include/mused_hip.h one to one
_PROTOS = {
    "mused_last_error": (C.c_char_p, []),
    "mused_version": (_i, []),
    "mused_memcpy_d2d": (_i, [_vp, _vp, _l, _vp]),
    "mused_row_sq_norms": (_i, [_vp, _i, _l, _i, _l, _vp, _vp]),
    "mused_pairwise_scores": (_i, [_vp, _i, _l, _i, _l, _i, _vp, _vp, _vp]),
    "mused_select_k_smallest": (_i, [_vp, _l, _i, _i, _vp, _vp, _i, _vp]),
    "mused_knn_topk": (_i, [_vp, _i, _l, _i, _l, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "mused_knn_fused_ws_bytes": (_l, [_l, _i]),
    "mused_knn_fused": (_i, [_vp, _i, _l, _i, _l, _i, _i, _vp, _l, _i, _vp, _vp, _i, _vp, _vp]),
    "mused_knn_fused_hop": (_i, [_vp, _i, _l, _i, _l, _i, _i, _vp, _l, _i, _l, _i, _vp, _vp, _i, _vp, _vp]),
    "mused_record_scores": (_i, [_vp, _i, _i, _vp, _vp]),
    "mused_group_mask": (_i, [_vp, _i, _vp, _i, _vp]),
    "mused_record_knn": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "mused_jaccard_knn": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "mused_jaccard_scores": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "mused_adj_fuse": (_i, [C.POINTER(_vp), _i, _i, _i, _vp, _vp]),
    "mused_adj_degrees": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "mused_adj_csr_fill": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "mused_adj_transpose": (_i, [_vp, _i, _i, _vp, _vp]),
    "mused_adj_to_dense": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "mused_adj_from_dense": (_i, [_vp, _i, _i, _l, _i, _vp, _vp, _vp]),
    "mused_rsvd_create": (_i, [_i, _i, _l, _i, C.POINTER(_vp)]),
    "mused_rsvd_set_mode": (_i, [_vp, _i]),
    "mused_rsvd_destroy": (_i, [_vp]),
    "mused_rsvd_mask_buffer": (_vp, [_vp]),
    "mused_rsvd_set_q0": (_i, [_vp, _vp, _i, _i, _vp]),
    "mused_rsvd_reduce": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "mused_rsvd_flags": (_vp, [_vp]),
    "mused_rsvd_status": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), _vp]),
    "mused_spmm_binary": (_i, [_vp, _vp, _i, _vp, _l, _i, _vp, _l, _vp]),
    "mused_lu_permute_l": (_i, [_vp, _i, _i, _l, _vp, _vp, _vp]),
    "mused_qr_economic": (_i, [_vp, _i, _i, _l, _vp, _l, _vp, _vp]),
    "mused_syevj_batched": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "mused_gemm_f64": (_i, [_i, _i, _vp, _l, _vp, _l, _vp, _l, _i, _i, _i, _d, _vp]),
    "mused_gemm_f64_batched": (_i, [_i, _i, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _i, _i, _i, _i, _d, _vp]),
    "mused_kmeans_ws_bytes": (_l, [_i, _i, _i]),
    "mused_kmeans_lloyd": (_i, [_vp, _l, _i, _i, _i, _vp, _vp, _d, _i, _vp, C.POINTER(_i), _vp, _l, _vp]),
    "mused_swfd_create": (_i, [_l, _d, _i, _i, _i, C.POINTER(_vp)]),
    "mused_swfd_create_lanes": (_i, [_l, _d, _i, _i, _i, _i, C.POINTER(_vp)]),
    "mused_swfd_lanes": (_i, [_vp]),
    "mused_swfd_append_lanes": (_i, [_vp, _vp, _i, _l, _l, _l, _vp]),
    "mused_swfd_destroy": (_i, [_vp]),
    "mused_swfd_levels": (_i, [_vp]),
    "mused_swfd_profile": (_i, [_vp, _i]),
    "mused_swfd_profile_read": (_i, [_vp, C.POINTER(_d), C.POINTER(_l), C.POINTER(_d)]),
    "mused_swfd_profile_read_direct": (_i, [_vp, C.POINTER(_d), C.POINTER(_l), C.POINTER(_d), C.POINTER(_i), C.POINTER(_d)]),
    "mused_swfd_append": (_i, [_vp, _vp, _i, _l, _l, _vp]),
    "mused_swfd_query": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "mused_swfd_counters": (_i, [_vp, C.POINTER(_l), C.POINTER(_i)]),
    "mused_swfd_status": (_i, [_vp, C.POINTER(_i), _vp]),
    "mused_swfd_half_bytes": (_l, [_vp]),
    "mused_swfd_export_half": (_i, [_vp, _i, _vp, _vp]),
    "mused_swfd_import_half": (_i, [_vp, _i, _vp, _vp]),
    "mused_swfd_begin_epoch": (_i, [_vp, _l, _vp, _vp]),
    "mused_fd_rotate": (_i, [_vp, _i, _i, _vp, _i, _vp]),
}

EXPORTED = tuple(_PROTOS)


def lib():
    """The loaded library; raises MusedError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MusedError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C mused_amd/csrc`)"
            )
        # torch first: libmused_hip must bind to the HIP runtime torch ships (same soname), so that
        # streams and device pointers are shared between the two.
        import torch  # noqa: F401

        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().mused_last_error()
        raise MusedError(f"libmused_hip error {rc}: {msg.decode() if msg else '?'}")


def call(name: str, *args):
    """Call an int-returning entry point and raise on a nonzero status."""
    check(getattr(lib(), name)(*args))
