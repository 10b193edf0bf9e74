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
