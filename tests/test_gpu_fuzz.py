"""Seeded randomised parity sweep (tools/fuzz_parity.py): SWFD device vs oracle over random shapes, batchings, dtypes
and degenerate streams (constant rows, zero rows, rank-deficient, binary int64), kNN adjacency device vs oracle incl.
duplicates, lattices, zero rows, non-finite rows, k from 0 to n, float32 / float64, both metrics."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_cases(seed):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import fuzz_parity as fz

    for i in range(60):
        rng = np.random.default_rng([seed, i])
        (fz.swfd_case if i % 2 == 0 else fz.knn_case)(rng, i)


def test_random_eigenstep_and_pipeline_cases():
    """Eigenstep on random kNN graphs (sigma to 1e-8, embedding columns with separated singular values to 1e-6, device
    k-means == scikit-learn on the same embedding) and the whole window loop against the oracle's restatement of
    main.py:13-130 with random window sizes, hop ratios, approaches and modality types (labels equal; where the oracle
    raises scipy's "cost matrix is infeasible", so must the device pipeline)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import fuzz_parity as fz

    for i in range(40):
        rng = np.random.default_rng([13, i])
        (fz.rsvd_case if i % 2 == 0 else fz.pipeline_case)(rng, i)


def test_random_lanes_and_metadata_cases():
    """Lock-step lanes (incl. twin lanes) == independent sketches bit for bit and == the oracle; metadata modality
    types on random columns (duplicate geotags / time stamps, everything missing, k = 0) == the oracle."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import fuzz_parity as fz

    for i in range(40):
        rng = np.random.default_rng([14, i])
        (fz.lanes_case if i % 2 == 0 else fz.meta_case)(rng, i)
