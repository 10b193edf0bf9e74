"""CPU-side checks: the C-ABI library loads and exports every symbol include/mused_hip.h declares
(no compute calls without a GPU), the ctypes table mirrors the header, host-side logic of the
drop-in modules, and the loud failure when no GPU / no extension is present."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "mused_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mused_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_what_ctypes_binds():
    from mused_amd import _lib

    assert header_symbols() == sorted(_lib.EXPORTED)


def test_library_loads_and_exports_every_declared_symbol():
    from mused_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    L = _lib.lib()
    for name in header_symbols():
        assert hasattr(L, name), name
    assert L.mused_version() >= 100
    assert isinstance(L.mused_last_error(), bytes)


def test_product_path_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mused_amd import matrix_operations as mo
    from mused_amd.engine import WindowEngine
    from mused_amd.swfd import SeqBasedSWFD

    with pytest.raises(Exception):
        WindowEngine(64)
    with pytest.raises(Exception):
        SeqBasedSWFD(N=8, R=1.0, d=4, sketch_dim=2)
    with pytest.raises(Exception):
        mo.create_adjacency_matrix(np.zeros((4, 2)), "", 2)
    with pytest.raises(Exception):
        mo.perform_svd_reduction(np.eye(4), 2, 0)


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mused_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "import_module(\"oracle" not in src and "__import__(\"oracle" not in src, f


def test_host_consumers_match_reference_goldens():
    """match_clusters / perform_clustering of the drop-in module (host side, no GPU needed)."""
    from conftest import load_golden
    from mused_amd import matrix_operations as mo

    g = load_golden("edges")
    out = mo.match_clusters(g["match_prev"], g["match_new"], "hungarian", 3)
    assert np.array_equal(out, g["match_out"])
    out = mo.match_clusters(g["match_prev"], g["match_new_inf"], "hungarian", 3)
    assert np.array_equal(out, g["match_out_inf"])
    new = g["match_new"]
    assert mo.match_clusters(None, new) is new
    assert mo.match_clusters([], new) is new
    with pytest.raises(ValueError):
        mo.match_clusters(g["match_prev"], new, "nope", 3)
    A = np.arange(12.0).reshape(3, 4)
    F = mo.fuse_matrices([A])  # single modality: plain float64 copy (matrix_operations.py:135)
    assert F is not A and F.dtype == np.float64 and np.array_equal(F, A)
    with pytest.raises(Exception):  # metadata types run on the device too: no CPU fallback without one
        mo.adjacency_on_device(np.zeros((4, 2)), "location", 2)


def test_modality_types_map_to_their_own_kernels():
    from mused_amd import matrix_operations as mo

    for t in ("location", "time", "username", "tags"):
        with pytest.raises(ValueError):  # never silently treated as dense Euclidean rows
            mo._metric_for(t)
    assert mo._metric_for("") == "l2" and mo._metric_for("anything") == "l2" and mo._metric_for("cosine") == "cosine"
    assert mo._metric_for("text") == "cosine"  # host TF-IDF + the device cosine kernel (matrix_operations.py:91-110)
    # edges a row can hold, per type (matrix_operations.py:25, 35, 74, 93, 113): sizes the CSR of the eigenstep
    assert [mo.edges_per_row(t, 50) for t in ("", "cosine", "text", "location", "time", "tags", "username")] == \
        [50, 51, 51, 51, 151, 50, None]
