"""GPU parity of every building block of libmused_hip against NumPy / SciPy / the CPU oracle.
All calls go through the C ABI (ctypes).  Run on the MI355X box: pytest -m gpu."""
import ctypes as C

import numpy as np
import pytest
import scipy.linalg

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mused_amd import _lib

    return _lib


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def sync():
    torch.cuda.synchronize()


# ---------------------------------------------------------------- GEMM ------------
@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (1, 0), (0, 1), (0, 0)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (130, 67, 45), (256, 300, 1024), (16, 16, 7), (513, 129, 250)])
def test_gemm_f64_layouts(L, a_kc, b_kc, M, N, K):
    rng = np.random.default_rng(M * 7 + N * 3 + K + a_kc * 2 + b_kc)
    A = rng.standard_normal((M, K))
    B = rng.standard_normal((K, N))
    As = A if a_kc else np.ascontiguousarray(A.T)       # [M][K] or [K][M]
    Bs = np.ascontiguousarray(B.T) if b_kc else B       # [N][K] or [K][N]
    dA, dB = dev(As), dev(Bs)
    dC = torch.full((M, N), np.nan, dtype=torch.float64, device="cuda")
    L.call("mused_gemm_f64", a_kc, b_kc, P(dA), As.shape[1], P(dB), Bs.shape[1], P(dC), N, M, N, K, 0.5, S())
    sync()
    ref = 0.5 * (A @ B)
    np.testing.assert_allclose(dC.cpu().numpy(), ref, rtol=0, atol=1e-12 * K)


def test_gemm_f64_asymmetric_identity(L):
    """A = I with an asymmetric B catches a transposed C write (MFMA C/D layout)."""
    n = 128
    A = np.eye(n)
    B = np.arange(n * n, dtype=np.float64).reshape(n, n)
    dA, dB = dev(A), dev(B)
    dC = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    L.call("mused_gemm_f64", 1, 0, P(dA), n, P(dB), n, P(dC), n, n, n, n, 1.0, S())
    sync()
    assert np.array_equal(dC.cpu().numpy(), B)


def test_gemm_f64_batched(L):
    rng = np.random.default_rng(0)
    b, M, N, K = 5, 96, 70, 200
    A = rng.standard_normal((b, M, K))
    B = rng.standard_normal((b, N, K))
    dA, dB = dev(A), dev(B)
    dC = torch.zeros((b, M, N), dtype=torch.float64, device="cuda")
    L.call("mused_gemm_f64_batched", 1, 1, P(dA), K, M * K, P(dB), K, N * K, P(dC), N, M * N, M, N, K, b, 1.0, S())
    sync()
    np.testing.assert_allclose(dC.cpu().numpy(), np.einsum("bmk,bnk->bmn", A, B), atol=1e-11)


# ---------------------------------------------------------------- scores / select ----
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,d", [(500, 64), (333, 50), (1000, 1024)])
def test_pairwise_scores_match_oracle(L, dtype, n, d):
    from oracle import mo_oracle as mo

    rng = np.random.default_rng(n + d)
    X = rng.standard_normal((n, d)).astype(dtype)
    X[5] = 0  # zero-norm row exercises the cosine normaliser
    dX = dev(X)
    dt = L.F32 if dtype == np.float32 else L.F64
    nrm = torch.empty(n, dtype=torch.float64, device="cuda")
    Sm = torch.empty((n, n), dtype=torch.float64, device="cuda")
    L.call("mused_row_sq_norms", P(dX), dt, n, d, d, P(nrm), S())
    sync()
    np.testing.assert_allclose(nrm.cpu().numpy(), (X.astype(np.float64) ** 2).sum(1), rtol=1e-13)
    L.call("mused_pairwise_scores", P(dX), dt, n, d, d, L.METRIC_L2, P(nrm), P(Sm), S())
    sync()
    ref = mo.sq_euclidean_scores(X)
    np.testing.assert_allclose(Sm.cpu().numpy(), ref, rtol=0, atol=1e-10 * ref.max())
    L.call("mused_pairwise_scores", P(dX), dt, n, d, d, L.METRIC_COSINE, P(nrm), P(Sm), S())
    sync()
    np.testing.assert_allclose(Sm.cpu().numpy(), mo.cosine_scores(X), rtol=0, atol=1e-13)


def _select(L, Smat, k, want_idx=True):
    n = Smat.shape[0]
    w = (n + 63) // 64
    dS = dev(Smat)
    idx = torch.full((n, k), -1, dtype=torch.int32, device="cuda")
    mask = torch.full((n, w), -1, dtype=torch.int64, device="cuda")
    L.call("mused_select_k_smallest", P(dS), n, n, k, P(idx), P(mask), w, S())
    sync()
    return idx.cpu().numpy(), mask.cpu().numpy().view(np.uint64)


def _mask_to_bool(mask, n):
    bits = np.unpackbits(mask.view(np.uint8), axis=1, bitorder="little")
    return bits[:, :n].astype(bool)


@pytest.mark.parametrize("n,k", [(64, 1), (100, 7), (500, 50), (777, 64), (2049, 50), (300, 300)])
def test_select_k_random(L, n, k):
    from oracle.mo_oracle import _select_k_smallest_mask

    rng = np.random.default_rng(n * 31 + k)
    Smat = rng.standard_normal((n, n)) * 50 + 2000
    idx, mask = _select(L, Smat, k)
    ref = _select_k_smallest_mask(Smat, k)
    got = np.zeros((n, n), dtype=bool)
    np.put_along_axis(got, idx.astype(np.int64), True, axis=1)
    assert np.array_equal(got, ref)
    assert np.all(np.diff(idx, axis=1) > 0)  # ascending columns
    ref_noself = ref.copy()
    np.fill_diagonal(ref_noself, False)
    assert np.array_equal(_mask_to_bool(mask, n), ref_noself)


def test_select_k_ties_and_degenerate(L):
    from oracle.mo_oracle import _select_k_smallest_mask

    rng = np.random.default_rng(3)
    n = 400
    for Smat, k in [
        (rng.integers(0, 6, size=(n, n)).astype(np.float64), 37),  # massive ties
        (np.zeros((n, n)), 11),  # all equal -> first k columns
        (np.tile(np.arange(n, dtype=np.float64)[::-1], (n, 1)), 5),  # descending -> last k columns
        (-rng.random((n, n)), 50),  # negative keys (cosine scores)
        (rng.standard_normal((n, n)) * 1e-300, 9),  # denormal-ish magnitudes, both signs
    ]:
        idx, mask = _select(L, Smat, k)
        ref = _select_k_smallest_mask(Smat, k)
        got = np.zeros((n, n), dtype=bool)
        np.put_along_axis(got, idx.astype(np.int64), True, axis=1)
        assert np.array_equal(got, ref)


def test_select_k_long_rows_uncached(L):
    """n > 16384: keys are re-read from global memory instead of LDS."""
    from oracle.mo_oracle import _select_k_smallest_mask

    rng = np.random.default_rng(5)
    n, k = 16500, 50
    Smat = (rng.standard_normal((n, n)).astype(np.float32) * 30 + 1000).astype(np.float64)
    idx, _ = _select(L, Smat, k)
    rows = rng.integers(0, n, size=64)
    ref = _select_k_smallest_mask(Smat[rows], k)
    got = np.zeros((64, n), dtype=bool)
    np.put_along_axis(got, idx[rows].astype(np.int64), True, axis=1)
    assert np.array_equal(got, ref)


# ---------------------------------------------------------------- adjacency ops -------
def test_adjacency_bit_ops(L):
    rng = np.random.default_rng(1)
    n = 333
    w = (n + 63) // 64
    A = rng.random((n, n)) < 0.07
    B = rng.random((n, n)) < 0.02
    Ai = A.astype(np.int64)
    dA = dev(Ai)
    mA = torch.empty((n, w), dtype=torch.int64, device="cuda")
    mB = torch.empty((n, w), dtype=torch.int64, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    dBf = dev(B.astype(np.float64))
    L.call("mused_adj_from_dense", P(dA), L.I64, n, n, w, P(mA), P(flag), S())
    L.call("mused_adj_from_dense", P(dBf), L.F64, n, n, w, P(mB), P(flag), S())
    sync()
    assert int(flag.item()) == 0
    assert np.array_equal(_mask_to_bool(mA.cpu().numpy().view(np.uint64), n), A)
    # non-binary entries are flagged
    C2 = Ai.copy()
    C2[3, 4] = 2
    tmp = torch.empty((n, w), dtype=torch.int64, device="cuda")
    dC2 = dev(C2)
    L.call("mused_adj_from_dense", P(dC2), L.I64, n, n, w, P(tmp), P(flag), S())
    sync()
    assert int(flag.item()) == 1
    # fuse
    out = torch.empty((n, w), dtype=torch.int64, device="cuda")
    arr = (C.c_void_p * 2)(mA.data_ptr(), mB.data_ptr())
    L.call("mused_adj_fuse", arr, 2, n, w, P(out), S())
    sync()
    assert np.array_equal(_mask_to_bool(out.cpu().numpy().view(np.uint64), n), A | B)
    # degrees / csr
    deg = torch.empty(n, dtype=torch.int32, device="cuda")
    rowptr = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    stats = torch.empty(2, dtype=torch.int32, device="cuda")
    L.call("mused_adj_degrees", P(mA), n, w, P(deg), P(rowptr), P(stats), S())
    sync()
    assert np.array_equal(deg.cpu().numpy(), A.sum(1))
    assert np.array_equal(rowptr.cpu().numpy(), np.concatenate([[0], np.cumsum(A.sum(1))]))
    assert stats.cpu().tolist() == [int(A.sum(1).max()), int(A.sum())]
    col = torch.empty(int(A.sum()), dtype=torch.int32, device="cuda")
    L.call("mused_adj_csr_fill", P(mA), n, w, P(rowptr), P(col), S())
    sync()
    assert np.array_equal(col.cpu().numpy(), np.nonzero(A)[1])
    # transpose
    mT = torch.empty((n, w), dtype=torch.int64, device="cuda")
    L.call("mused_adj_transpose", P(mA), n, w, P(mT), S())
    sync()
    assert np.array_equal(_mask_to_bool(mT.cpu().numpy().view(np.uint64), n), A.T)
    # dense export, both dtypes
    d64 = torch.empty((n, n), dtype=torch.float64, device="cuda")
    di = torch.empty((n, n), dtype=torch.int64, device="cuda")
    L.call("mused_adj_to_dense", P(mA), n, w, L.F64, P(d64), S())
    L.call("mused_adj_to_dense", P(mA), n, w, L.I64, P(di), S())
    sync()
    assert np.array_equal(d64.cpu().numpy(), A.astype(np.float64)) and np.array_equal(di.cpu().numpy(), Ai)


def test_spmm_binary(L):
    rng = np.random.default_rng(2)
    n, r = 700, 138
    A = rng.random((n, n)) < 0.05
    A[10] = False  # empty row
    Q = rng.standard_normal((n, r))
    rowptr = np.concatenate([[0], np.cumsum(A.sum(1))]).astype(np.int32)
    col = np.nonzero(A)[1].astype(np.int32)
    Y = torch.full((n, r), np.nan, dtype=torch.float64, device="cuda")
    d_rowptr, d_col, d_Q = dev(rowptr), dev(col), dev(Q)  # keep the device copies alive across the call
    L.call("mused_spmm_binary", P(d_rowptr), P(d_col), n, P(d_Q), r, r, P(Y), r, S())
    sync()
    np.testing.assert_allclose(Y.cpu().numpy(), A.astype(np.float64) @ Q, atol=1e-12)


# ---------------------------------------------------------------- panels ---------------
@pytest.mark.parametrize("n,r", [(500, 26), (2000, 74), (8, 12), (40, 40), (1030, 138)])
def test_lu_permute_l(L, n, r):
    rng = np.random.default_rng(n + r)
    Y = rng.standard_normal((n, r))
    ref, _ = scipy.linalg.lu(Y, permute_l=True)
    dY = dev(Y)
    wi = torch.empty(n + (n + 15) // 16, dtype=torch.int32, device="cuda")
    wf = torch.empty(4 * (r + n), dtype=torch.float64, device="cuda")
    L.call("mused_lu_permute_l", P(dY), n, r, r, P(wi), P(wf), S())
    sync()
    k = min(n, r)
    np.testing.assert_allclose(dY.cpu().numpy()[:, :k], ref, atol=1e-11)


def test_lu_rank_deficient(L):
    """Zero columns / duplicated columns must not produce NaN (LAPACK skips zero pivots)."""
    rng = np.random.default_rng(9)
    n, r = 300, 20
    Y = rng.standard_normal((n, r))
    Y[:, 5] = 0.0
    Y[:, 7] = Y[:, 3]
    dY = dev(Y)
    wi = torch.empty(n + (n + 15) // 16, dtype=torch.int32, device="cuda")
    wf = torch.empty(4 * (r + n), dtype=torch.float64, device="cuda")
    L.call("mused_lu_permute_l", P(dY), n, r, r, P(wi), P(wf), S())
    sync()
    out = dY.cpu().numpy()
    assert np.isfinite(out).all()
    # the column space of the well-posed leading block is reproduced
    ref, _ = scipy.linalg.lu(Y[:, :5], permute_l=True)
    np.testing.assert_allclose(out[:, :5], ref, atol=1e-11)


@pytest.mark.parametrize("n,r", [(500, 26), (2000, 74), (40, 40), (1030, 138)])
def test_qr_economic(L, n, r):
    rng = np.random.default_rng(n * 3 + r)
    Y = rng.standard_normal((n, r))
    ref, _ = scipy.linalg.qr(Y, mode="economic")
    dY = dev(Y)
    Q = torch.empty((n, r), dtype=torch.float64, device="cuda")
    ws = torch.empty(r + ((n + 511) // 512) * r + 2 * n, dtype=torch.float64, device="cuda")
    L.call("mused_qr_economic", P(dY), n, r, r, P(Q), r, P(ws), S())
    sync()
    q = Q.cpu().numpy()
    np.testing.assert_allclose(q.T @ q, np.eye(r), atol=1e-12)
    np.testing.assert_allclose(q, ref, atol=1e-11)  # same Householder sign convention as LAPACK


# ---------------------------------------------------------------- eigensolver -----------
@pytest.mark.parametrize("n,batch,sweeps", [(8, 3, 8), (64, 4, 12), (138, 1, 13), (256, 2, 14), (512, 1, 16)])
def test_syevj(L, n, batch, sweeps):
    rng = np.random.default_rng(n + batch)
    G = np.empty((batch, n, n))
    for b in range(batch):
        B = rng.standard_normal((n, n + 20))
        if b == batch - 1:
            B[: n // 4] = 0.0  # zero rows -> exact zero eigenvalues, as in a half-filled FD buffer
        G[b] = B @ B.T
    ev = torch.empty((batch, n), dtype=torch.float64, device="cuda")
    V = torch.empty((batch, n, n), dtype=torch.float64, device="cuda")
    dG = dev(G)
    L.call("mused_syevj_batched", P(dG), n, batch, sweeps, P(ev), P(V), S())
    sync()
    ev, V = ev.cpu().numpy(), V.cpu().numpy()
    for b in range(batch):
        scale = np.abs(G[b]).max()
        ref = np.maximum(np.linalg.eigvalsh(G[b]), 0)
        got = np.sort(ev[b])
        # one-sided Jacobi on G: the upper half of the spectrum (all the path ever uses) to rounding,
        # eigenvalues far below the largest one to ~1e-7 of it
        # one-sided Jacobi on G: absolute error ~ eps * lam_max^2 / lam_i -> the upper half of the
        # spectrum (all the path uses) to rounding, the smallest eigenvalues to ~1e-10 lam_max
        np.testing.assert_allclose(got[n // 2 :], ref[n // 2 :], rtol=0, atol=2e-12 * scale)
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-10 * scale)
        nz = ev[b] > 1e-9 * scale  # eigenvectors of (numerically) zero eigenvalues are returned as 0 by design
        Vn = V[b][:, nz]
        np.testing.assert_allclose(Vn.T @ Vn, np.eye(int(nz.sum())), atol=1e-11)
        np.testing.assert_allclose(G[b] @ V[b], V[b] * ev[b][None, :], atol=2e-10 * scale)
        assert not V[b][:, ~nz].any() or np.abs(ev[b][~nz]).max() <= 1e-9 * scale


@pytest.mark.parametrize("n", [64, 192, 256, 320, 448, 512])
def test_syevj_special_matrices(L, n):
    """Degenerate inputs of the adaptive one-sided solver: zero matrix, diagonal, rank one, a multiple
    eigenvalue, heavy rank deficiency (rank n/3, dense) -- eigenvalues to 1e-12 of the largest one within the
    sweep cap, no NaN."""
    rng = np.random.default_rng(n)
    mats = [np.zeros((n, n)), np.diag(np.linspace(5.0, 0.0, n))]
    u = rng.standard_normal(n)
    mats.append(np.outer(u, u))
    Qm, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.concatenate([np.full(n // 4, 7.0), np.full(n // 4, 2.0), np.linspace(1.0, 0.5, n - 2 * (n // 4))])
    mats.append((Qm * lam) @ Qm.T)
    B = rng.standard_normal((n, n // 3))
    mats.append(B @ B.T)
    G = np.stack([0.5 * (m + m.T) for m in mats])
    ev = torch.empty((len(mats), n), dtype=torch.float64, device="cuda")
    V = torch.empty((len(mats), n, n), dtype=torch.float64, device="cuda")
    dG = dev(G)
    L.call("mused_syevj_batched", P(dG), n, len(mats), 30, P(ev), P(V), S())
    sync()
    ev, V = ev.cpu().numpy(), V.cpu().numpy()
    assert np.isfinite(ev).all() and np.isfinite(V).all()
    for b in range(len(mats)):
        scale = max(np.abs(G[b]).max(), 1e-300)
        ref = np.maximum(np.linalg.eigvalsh(G[b]), 0)
        np.testing.assert_allclose(np.sort(ev[b]), ref, rtol=0, atol=1e-10 * scale, err_msg=f"matrix {b}")
        nz = ev[b] > 1e-9 * scale  # columns of the numerical null space carry no direction (documented)
        # inside an EXACTLY multiple eigenvalue the adaptive stop leaves 1e-10 .. 1e-8 between the vectors of the cluster
        # (a rotation between two columns of equal norm is large however small their cosine, so the last sweep is
        # first order there, not second); which end of that range depends on the rounding of the host BLAS product
        # that builds the test matrix (measured 1.7e-10 and 1.4e-8 for n = 320 on two hosts).  The path's own
        # matrices (sketch buffers) are checked against the oracle at 1e-8 sigma_1 in test_gpu_path.py.
        np.testing.assert_allclose(G[b] @ V[b][:, nz], V[b][:, nz] * ev[b][nz][None, :], atol=1e-7 * scale, err_msg=f"matrix {b}")
        Vn = V[b][:, nz]
        np.testing.assert_allclose(Vn.T @ Vn, np.eye(int(nz.sum())), atol=1e-7, err_msg=f"matrix {b}")


@pytest.mark.parametrize("ell,d", [(8, 40), (16, 100), (128, 1024)])
def test_fd_rotate(L, ell, d):
    rng = np.random.default_rng(ell + d)
    buf = rng.standard_normal((2 * ell, d)) * np.linspace(3, 0.1, 2 * ell)[:, None]
    dB = dev(buf)
    sig = torch.empty(ell, dtype=torch.float64, device="cuda")
    L.call("mused_fd_rotate", P(dB), ell, d, P(sig), 0, S())
    sync()
    out = dB.cpu().numpy()
    _, s, Vt = np.linalg.svd(buf, full_matrices=False)
    s2 = np.maximum(s[:ell] ** 2 - s[ell - 1] ** 2, 0)
    ref = np.sqrt(s2)[:, None] * Vt[:ell]
    assert not out[ell:].any()
    np.testing.assert_allclose(sig.cpu().numpy(), np.sqrt(s2), rtol=0, atol=1e-9 * s[0])
    # rows match up to sign
    for i in range(ell - 1):
        sgn = np.sign(out[i] @ ref[i]) or 1.0
        np.testing.assert_allclose(sgn * out[i], ref[i], atol=1e-7 * s[0])
    np.testing.assert_allclose(out.T @ out, ref.T @ ref, atol=1e-9 * s[0] ** 2)


# ---------------------------------------------------------------- fused similarity + top-k (no n x n matrix) ----
def _classic_topk(L, X, k, metric):
    n, d = X.shape
    dX = dev(X)
    ws = torch.empty(n * n, dtype=torch.float64, device="cuda")
    nr = torch.empty(n, dtype=torch.float64, device="cuda")
    w = (n + 63) // 64
    idx = torch.empty((n, k), dtype=torch.int32, device="cuda")
    mask = torch.empty((n, w), dtype=torch.int64, device="cuda")
    dt = 0 if X.dtype == np.float32 else 1
    L.call("mused_knn_topk", P(dX), dt, n, d, d, k, metric, P(ws), P(nr), P(idx), P(mask), w, S())
    sync()
    return idx.cpu().numpy(), mask.cpu().numpy()


def _fused_topk(L, X, k, metric, cap):
    n, d = X.shape
    dX = dev(X)
    nb = L.lib().mused_knn_fused_ws_bytes(n, cap)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    w = (n + 63) // 64 + 1  # a pitch larger than needed: the extra word must come out zero
    idx = torch.full((n, k), -1, dtype=torch.int32, device="cuda")
    mask = torch.full((n, w), -1, dtype=torch.int64, device="cuda")
    ovf = torch.full((1,), 7, dtype=torch.int32, device="cuda")
    dt = 0 if X.dtype == np.float32 else 1
    L.call("mused_knn_fused", P(dX), dt, n, d, d, k, metric, P(ws), nb, cap, P(idx), P(mask), w, P(ovf), S())
    sync()
    return idx.cpu().numpy(), mask.cpu().numpy(), int(ovf.item())


@pytest.mark.parametrize("n,d,k,metric,dtype", [
    (100, 9, 7, 0, np.float64),       # one tile
    (128, 16, 128, 0, np.float32),    # k = n
    (300, 33, 20, 0, np.float32),     # 3 tiles, ragged last tile, unaligned rows (scalar loads)
    (1024, 64, 50, 0, np.float32),    # 8 tiles: even tile count (antipodal pairs once)
    (1500, 48, 50, 1, np.float32),    # cosine, 12 tiles
    (2000, 24, 200, 0, np.float64),   # large k: the lists hold 4 k + 128 candidates
    (2900, 40, 31, 1, np.float64),    # 23 tiles: four phases
])
def test_knn_fused_equals_classic(L, n, d, k, metric, dtype):
    """Fused similarity + selection (no score matrix) == mused_pairwise_scores + mused_select_k_smallest, bit for bit:
    neighbour lists (ascending columns) and adjacency bitmask rows."""
    rng = np.random.default_rng(n + d + k)
    X = rng.standard_normal((n, d)).astype(dtype)
    X[n // 3] = X[n // 5]          # duplicate rows: exact score ties across tiles
    if metric == 0 or n <= 1500:
        X[n // 2] = 0.0            # a zero row (cosine: norm replaced by 1, every score of the row equal)
    # (a row of all-equal scores admits every column smaller than its current k-th one: in a long window that overflows
    # the lists and the engine falls back to the classic path -- test_engine_cosine_zero_rows_fall_back below)
    cap = 1024
    i_ref, m_ref = _classic_topk(L, X, k, metric)
    i_fu, m_fu, ovf = _fused_topk(L, X, k, metric, cap)
    assert ovf == 0
    assert np.array_equal(i_fu, i_ref)
    w = (n + 63) // 64
    assert np.array_equal(m_fu[:, :w], m_ref) and not m_fu[:, w:].any()


def test_knn_fused_ties_and_overflow_flag(L):
    """Lattice data (thousands of exactly equal scores): with room in the lists the result equals the classic path,
    with a tight cap the overflow flag is raised (the engine then falls back to the classic path)."""
    rng = np.random.default_rng(5)
    X = rng.integers(0, 2, size=(700, 6)).astype(np.float64)   # 64 distinct points, ~11 copies each
    i_ref, m_ref = _classic_topk(L, X, 30, 0)
    i_fu, m_fu, ovf = _fused_topk(L, X, 30, 0, 1024)
    assert ovf == 0 and np.array_equal(i_fu, i_ref) and np.array_equal(m_fu[:, :11], m_ref)
    _, _, ovf = _fused_topk(L, X, 30, 0, 64)
    assert ovf != 0


def test_engine_falls_back_when_candidate_lists_overflow(L, monkeypatch):
    from mused_amd.engine import WindowEngine
    from oracle import mo_oracle as omo

    rng = np.random.default_rng(6)
    X = rng.integers(0, 2, size=(640, 5)).astype(np.float64)
    monkeypatch.setenv("MUSED_KNN_CAP", "64")
    eng = WindowEngine(640)
    adj = eng.knn_adjacency(torch.from_numpy(X).cuda(), 25)
    assert eng.knn_fallbacks == 1 and eng._scores is not None
    bits = np.unpackbits(adj.mask.cpu().numpy().view(np.uint8), axis=1, bitorder="little")[:, :640].astype(bool)
    assert np.array_equal(bits, omo.create_adjacency_matrix(X, "", 25).astype(bool))
    eng.close()


def test_engine_cosine_zero_rows_fall_back(L):
    """Zero-norm rows under cosine have the same similarity (0) to every row: thousands of exact ties.  Whether the
    candidate lists overflow or not, the adjacency equals the oracle's (ties to the smaller column)."""
    from mused_amd.engine import WindowEngine
    from oracle import mo_oracle as omo

    rng = np.random.default_rng(9)
    X = rng.standard_normal((2900, 24))
    X[::97] = 0.0
    eng = WindowEngine(2900)
    adj = eng.knn_adjacency(torch.from_numpy(X).cuda(), 31, "cosine")
    bits = np.unpackbits(adj.mask.cpu().numpy().view(np.uint8), axis=1, bitorder="little")[:, :2900].astype(bool)
    assert np.array_equal(bits, omo.create_adjacency_matrix(X, "cosine", 31).astype(bool))
    eng.close()


def test_engine_uses_the_classic_path_for_very_large_k(L):
    """k > 224 does not fit the candidate lists (4 k + 128 <= 1024): the engine takes the score-matrix path."""
    from mused_amd.engine import WindowEngine
    from oracle import mo_oracle as omo

    X = np.random.default_rng(8).standard_normal((900, 12))
    eng = WindowEngine(900)
    adj = eng.knn_adjacency(torch.from_numpy(X).cuda(), 500)
    assert eng._scores is not None and eng.knn_fallbacks == 0
    bits = np.unpackbits(adj.mask.cpu().numpy().view(np.uint8), axis=1, bitorder="little")[:, :900].astype(bool)
    assert np.array_equal(bits, omo.create_adjacency_matrix(X, "", 500).astype(bool))
    eng.close()


def test_engine_default_path_allocates_no_score_matrix(L):
    """W = 10,000: the product path keeps < 100 MB of similarity workspace (the classic path needs 800 MB)."""
    from mused_amd import synth
    from mused_amd.engine import WindowEngine

    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    before = torch.cuda.memory_allocated()
    eng = WindowEngine(10000)
    X = torch.from_numpy(synth.stream_window("blob", 3, 10000, 256, 0)[0]).cuda()
    base = torch.cuda.memory_allocated()
    adj = eng.knn_adjacency(X, 50)
    torch.cuda.synchronize()
    assert eng._scores is None and eng.knn_fallbacks == 0
    assert eng._knn_ws.numel() < 100e6                      # candidate lists + thresholds
    assert torch.cuda.memory_allocated() - base < 128e6     # ... plus the adjacency bitmask it returns
    assert int(adj.degrees()[2][0].item()) == 49
    eng.close()
    (before,)
