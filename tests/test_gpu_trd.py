"""The direct eigensolver of the FD rotation (csrc/trd.hip: tridiagonalisation + multisection + twisted factorisation +
back-transformation, order 256, top 128 eigenpairs) phase by phase against NumPy / LAPACK, and its certificate."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def sytd2_lower(G):
    """LAPACK dsytd2 (uplo = 'L'), unblocked: d, e of T = Q^T G Q."""
    A = np.array(G, dtype=np.float64)
    n = A.shape[0]
    d, e = np.zeros(n), np.zeros(n)
    for k in range(n - 1):
        x = A[k + 1:, k].copy()
        alpha, xnorm = x[0], np.linalg.norm(x[1:])
        d[k] = A[k, k]
        if xnorm == 0.0:
            e[k] = alpha
            continue
        beta = -np.copysign(np.hypot(alpha, xnorm), alpha)
        tau = (beta - alpha) / beta
        v = x / (alpha - beta)
        v[0] = 1.0
        e[k] = beta
        p = tau * (A[k + 1:, k + 1:] @ v)
        w = p - 0.5 * tau * (p @ v) * v
        A[k + 1:, k + 1:] -= np.outer(v, w) + np.outer(w, v)
    d[n - 1] = A[n - 1, n - 1]
    return d, e


def run_trd(Gs):
    from mused_amd import _lib
    from mused_amd.engine import ptr, stream_ptr

    L = _lib.lib()
    fn = L.mused_debug_trd
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p] * 1 + [C.c_int] + [C.c_void_p] * 6
    B = len(Gs)
    G = torch.from_numpy(np.ascontiguousarray(np.stack(Gs))).cuda()
    d = torch.zeros(B, 256, dtype=torch.float64, device="cuda")
    e = torch.zeros(B, 256, dtype=torch.float64, device="cuda")
    lam = torch.zeros(B, 128, dtype=torch.float64, device="cuda")
    res = torch.zeros(B, 128, dtype=torch.float64, device="cuda")
    done = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    _lib.check(fn(ptr(G), B, ptr(d), ptr(e), ptr(lam), ptr(res), ptr(done), stream_ptr()))
    torch.cuda.synchronize()
    return G.cpu().numpy(), d.cpu().numpy(), e.cpu().numpy(), lam.cpu().numpy(), res.cpu().numpy(), done.cpu().numpy()


def run_trd_n(Gs, n, need, cert_all=0):
    """The solver on matrices of order n <= 256 (embedded at the bottom right of the 256-layout, see csrc/trd.hip)."""
    from mused_amd import _lib
    from mused_amd.engine import ptr, stream_ptr

    L = _lib.lib()
    fn = L.mused_debug_trd_n
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6
    B = len(Gs)
    G = torch.from_numpy(np.ascontiguousarray(np.stack(Gs))).cuda()
    d = torch.zeros(B, 256, dtype=torch.float64, device="cuda")
    e = torch.zeros(B, 256, dtype=torch.float64, device="cuda")
    lam = torch.zeros(B, 128, dtype=torch.float64, device="cuda")
    res = torch.zeros(B, 128, dtype=torch.float64, device="cuda")
    done = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    _lib.check(fn(ptr(G), n, need, cert_all, B, ptr(d), ptr(e), ptr(lam), ptr(res), ptr(done), stream_ptr()))
    torch.cuda.synchronize()
    return G.cpu().numpy(), d.cpu().numpy(), e.cpu().numpy(), lam.cpu().numpy(), res.cpu().numpy(), done.cpu().numpy()


def fd_buffers(kind, nblk, seed=0, ell=128, d=1024):
    """Gram matrices of successive FD rotation buffers [kept rows; new block; zero rows] of a synthetic stream."""
    from mused_amd import synth

    X, _ = synth.stream_window(kind, 0, ell * nblk, d, seed)
    X = X.astype(np.float64)
    K = np.zeros((0, d))
    out = []
    for blk in range(nblk):
        buf = np.vstack([K, X[blk * ell:(blk + 1) * ell]])
        buf = np.vstack([buf, np.zeros((2 * ell - len(buf), d))])
        out.append(buf @ buf.T)
        _, s, Vt = np.linalg.svd(buf, full_matrices=False)
        lam = s ** 2
        s2 = np.maximum(lam[:ell] - lam[ell - 1], 0)
        keep = s2 > 1e-10 * lam[0]
        K = np.sqrt(s2[keep])[:, None] * Vt[:ell][keep]
    return out


@pytest.mark.parametrize("kind", ["blob", "gauss", "fd"])
def test_direct_solver_on_fd_rotation_buffers(kind):
    Gs = fd_buffers(kind, 5)
    out, d, e, lam, res, done = run_trd(Gs)
    for b, G in enumerate(Gs):
        scale = np.abs(np.linalg.eigvalsh(G)).max()
        dr, er = sytd2_lower(G)
        # phase A (entries of T are only determined to the conditioning of the reflectors: a loose check that catches
        # indexing mistakes; the eigenvalues below are the tight one)
        np.testing.assert_allclose(d[b], dr, rtol=0, atol=1e-8 * scale)
        np.testing.assert_allclose(e[b, :255], er[:255], rtol=0, atol=1e-8 * scale)
        w = np.linalg.eigvalsh(G)[::-1]
        np.testing.assert_allclose(lam[b], w[:128], rtol=0, atol=1e-13 * scale)  # phase B
        assert done[b] == 1, (b, res[b].max())
        cols = out[b][:128].T   # out[b][c] = column c (column-major storage) -> (256, 128)
        nrm = np.linalg.norm(cols, axis=0)
        np.testing.assert_allclose(nrm, np.maximum(w[:128], 0), rtol=0, atol=1e-12 * scale)  # |lam_j v_j| = lam_j
        sig = (w[:128] - w[127]) > 1e-10 * w[0]
        V = cols[:, sig] / nrm[sig]
        assert np.abs(V.T @ V - np.eye(V.shape[1])).max() < 1e-9                # orthonormal
        R = G @ V - V * w[:128][sig]
        assert np.abs(R).max() < 1e-11 * scale                                  # eigenvectors of G
        assert not out[b][128:].any()                                           # the lower half of the spectrum: zero columns


def test_direct_solver_special_matrices_and_certificate():
    rng = np.random.default_rng(1)
    Q = np.linalg.qr(rng.standard_normal((256, 256)))[0]
    cases = {
        "zero": np.zeros((256, 256)),
        "identity": np.eye(256) * 3.0,                                           # 256-fold eigenvalue, but nothing survives the shrink
        "hundredfold": (Q * np.r_[np.full(100, 5.0), np.linspace(4, 1, 156)]) @ Q.T,  # 100-fold eigenvalue -> rejected
        "rank40": (lambda B: B @ B.T)(rng.standard_normal((256, 40))),
        "diagonal": np.diag(np.linspace(1, 300, 256)),
        "graded": (Q * np.logspace(0, -12, 256)) @ Q.T,
        "binary": (lambda A: A @ A.T)((rng.random((256, 2000)) < 0.025).astype(float)),
        # the sign counts run on T scaled by a power of two: the scale of the input must not matter
        "scaled_up": (lambda B: B @ B.T)(rng.standard_normal((256, 300))) * 1e100,
        "scaled_down": (lambda B: B @ B.T)(rng.standard_normal((256, 300))) * 1e-100,
        # Wilkinson W_256^+ (pairs of eigenvalues agreeing to rounding at the top of the spectrum) as the Gram input:
        # eigenvalues must be right; the certificate may or may not reject the close pairs
        "wilkinson": np.diag(np.abs(np.arange(256) - 127.5)) + np.diag(np.ones(255), 1) + np.diag(np.ones(255), -1),
        # tridiagonal with tiny off-diagonals next to equal diagonal entries (zero minors in the sign count)
        "tiny_offdiag": np.diag(np.repeat(np.arange(1.0, 129.0), 2)) + np.diag(np.full(255, 1e-160), 1) + np.diag(np.full(255, 1e-160), -1),
    }
    names = list(cases)
    Gs = [0.5 * (cases[k] + cases[k].T) for k in names]
    out, d, e, lam, res, done = run_trd(Gs)
    for b, name in enumerate(names):
        G = Gs[b]
        w = np.linalg.eigvalsh(G)[::-1]
        scale = max(np.abs(w).max(), 1e-300)
        np.testing.assert_allclose(lam[b], w[:128], rtol=0, atol=1e-13 * scale + 1e-300, err_msg=name)
        if name == "hundredfold" or (name in ("wilkinson", "tiny_offdiag") and done[b] == 0):
            assert done[b] == 0, name                      # certificate: clustered eigenvalues go to the Jacobi solver
            assert np.array_equal(out[b], G), name         # ... with their input untouched
            continue
        assert done[b] == 1, (name, res[b].max())
        cols = out[b][:128].T
        nrm = np.linalg.norm(cols, axis=0)
        np.testing.assert_allclose(nrm, np.maximum(w[:128], 0), rtol=0, atol=1e-12 * scale + 1e-300, err_msg=name)
        sig = (w[:128] > 0) & ((w[:128] - w[127]) > 1e-10 * max(w[0], 0))
        if sig.any():
            V = cols[:, sig] / nrm[sig]
            assert np.abs(V.T @ V - np.eye(V.shape[1])).max() < 1e-8, name
            assert np.abs(G @ V - V * w[:128][sig]).max() < 1e-10 * scale, name


def test_fd_rotation_through_the_plan_matches_svd():
    """mused_fd_rotate (one rotation of a 256 x d buffer, l = 128: the plan now runs the direct solver) against the SVD."""
    from mused_amd import _lib
    from mused_amd.engine import ptr, stream_ptr

    rng = np.random.default_rng(3)
    ell, dd = 128, 700
    buf = rng.standard_normal((2 * ell, dd)) * np.logspace(0, -3, 2 * ell)[:, None]
    _, s, Vt = np.linalg.svd(buf, full_matrices=False)
    lam = s ** 2
    ref = np.sqrt(np.maximum(lam[:ell] - lam[ell - 1], 0))
    t = torch.from_numpy(buf).cuda()
    sig = torch.zeros(ell, dtype=torch.float64, device="cuda")
    _lib.call("mused_fd_rotate", ptr(t), ell, dd, ptr(sig), 0, stream_ptr())
    got = np.sort(sig.cpu().numpy())[::-1]
    keep = ref ** 2 > 1e-10 * lam[0]
    np.testing.assert_allclose(got[: keep.sum()], ref[keep], rtol=0, atol=1e-9 * s[0])
    B = t.cpu().numpy()[:ell]
    np.testing.assert_allclose(B.T @ B, (Vt[:ell].T * np.where(keep, ref ** 2, 0)) @ Vt[:ell], rtol=0, atol=1e-9 * lam[0])


def test_rejected_matrices_go_through_the_jacobi_fallback():
    """A buffer whose Gram matrix has a 100-fold eigenvalue above the cut: the direct solver's certificate rejects it, the
    queue Jacobi of the same plan solves it (mused_fd_rotate -> SWFD rotation plan), and the result equals the SVD's."""
    from mused_amd import _lib
    from mused_amd.engine import ptr, stream_ptr

    rng = np.random.default_rng(7)
    ell, dd = 128, 512
    Q = np.linalg.qr(rng.standard_normal((dd, 2 * ell)))[0].T          # 256 orthonormal rows
    s = np.r_[np.full(100, 5.0), np.linspace(4.0, 0.5, 156)]
    buf = (np.linalg.qr(rng.standard_normal((256, 256)))[0] * s) @ Q   # singular values s, 100 of them equal
    lam = np.sort(s)[::-1] ** 2
    ref = np.sqrt(np.maximum(lam[:ell] - lam[ell - 1], 0))
    t = torch.from_numpy(buf).cuda()
    sig = torch.zeros(ell, dtype=torch.float64, device="cuda")
    _lib.call("mused_fd_rotate", ptr(t), ell, dd, ptr(sig), 0, stream_ptr())
    got = np.sort(sig.cpu().numpy())[::-1]
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-9 * s.max())
    B = t.cpu().numpy()[:ell]
    _, _, Vt = np.linalg.svd(buf, full_matrices=False)
    # the kept rows span the top directions with the shrunk energies: compare B^T B with the SVD's
    np.testing.assert_allclose(B.T @ B, (Vt[:ell].T * ref ** 2) @ Vt[:ell], rtol=0, atol=1e-9 * lam[0])


@pytest.mark.parametrize("n", [20, 32, 100, 128, 150, 200, 254, 256])
def test_direct_solver_embedded_orders_match_lapack(n):
    """Orders below 256 (the reference's own operating range: reduced_dim 10 .. 100 -> rotations of order 20 .. 200, queries
    of order 30 .. 400; /root/reference/main.py:270,309) run embedded in the 256-layout: top n / 2 pairs of FD rotation
    buffers against LAPACK, zero columns / rows everywhere else, same certificate."""
    ell = n // 2
    Gs = fd_buffers("blob", 4, ell=ell, d=max(64, 3 * n)) + fd_buffers("fd", 2, ell=ell, d=max(64, 3 * n), seed=1)
    out, d, e, lam, res, done = run_trd_n(Gs, n, ell)
    off = 256 - n
    for b, G in enumerate(Gs):
        w = np.linalg.eigvalsh(G)[::-1]
        scale = np.abs(w).max()
        assert not d[b, :off].any() and not e[b, :off].any()                    # the skipped steps
        dr, er = sytd2_lower(G)
        np.testing.assert_allclose(d[b, off:], dr, rtol=0, atol=1e-8 * scale)
        np.testing.assert_allclose(e[b, off:255], er[: n - 1], rtol=0, atol=1e-8 * scale)
        np.testing.assert_allclose(lam[b, :ell], w[:ell], rtol=0, atol=1e-13 * scale)
        assert done[b] == 1, (b, res[b, :ell].max())
        cols = out[b].T                                                            # (n, n): column c
        nvec = min(n, 32 * ((ell + 31) // 32))
        nrm = np.linalg.norm(cols[:, :nvec], axis=0)
        np.testing.assert_allclose(nrm, np.maximum(w[:nvec], 0), rtol=0, atol=1e-12 * scale)
        assert not cols[:, nvec:].any()
        sig = (w[:ell] - w[ell - 1]) > 1e-10 * w[0]
        V = cols[:, :ell][:, sig] / nrm[:ell][sig]
        assert np.abs(V.T @ V - np.eye(V.shape[1])).max() < 1e-9
        assert np.abs(G @ V - V * w[:ell][sig]).max() < 1e-11 * scale


def test_direct_solver_certifies_every_needed_pair_when_asked():
    """cert_all (the eigenstep's use: all n_components pairs feed the embedding): a multiple eigenvalue AT the cut -- which
    the FD certificate ignores, nothing survives the shrink there -- must send the matrix to the Jacobi."""
    rng = np.random.default_rng(11)
    n, need = 138, 128
    Q = np.linalg.qr(rng.standard_normal((n, n)))[0]
    w = np.r_[np.linspace(9.0, 2.0, 125), np.full(6, 1.5), np.linspace(1.0, 0.1, 7)]   # lam_125 .. lam_130 equal
    G = (Q * w) @ Q.T
    G = 0.5 * (G + G.T)
    good = (Q * np.linspace(9.0, 0.5, n)) @ Q.T
    good = 0.5 * (good + good.T)
    out, d, e, lam, res, done = run_trd_n([G, good], n, need, cert_all=1)
    assert done[0] == 0 and np.array_equal(out[0], G)          # rejected, untouched
    assert done[1] == 1
    ww = np.linalg.eigvalsh(good)[::-1]
    V = out[1].T[:, :need] / np.linalg.norm(out[1].T[:, :need], axis=0)
    assert np.abs(V.T @ V - np.eye(need)).max() < 1e-9
    assert np.abs(good @ V - V * ww[:need]).max() < 1e-11 * ww[0]
    # the FD certificate accepts the first matrix: the repeated eigenvalue sits at the cut, where the shrink leaves nothing
    _, _, _, _, _, done_fd = run_trd_n([G], n, need, cert_all=0)
    assert done_fd[0] == 1


@pytest.mark.parametrize("n", [46, 92, 256])
def test_direct_solver_exactly_rank_deficient_duplicates(n):
    """Every row of the buffer the same (rank one, exact duplicates; the fuzz sweep's "const" streams): the rounding residue of
    the trailing matrix deflates level by level into the denormal range, where a Householder step has no valid scalars
    (round 4: NaN eigenvalues slipped through the certificate).  Such columns are now zero columns, and a non-finite T or
    spectrum is rejected."""
    rng = np.random.default_rng(0)
    Gs = []
    for rows in (n // 2, n // 2 - 3, n):
        buf = np.zeros((n, 30))
        buf[:rows] = rng.standard_normal(30)
        Gs.append(buf @ buf.T)
    out, d, e, lam, res, done = run_trd_n(Gs, n, n // 2 if n < 256 else 128)
    for b, G in enumerate(Gs):
        w = np.linalg.eigvalsh(G)[::-1]
        assert np.isfinite(lam[b, : min(n // 2, 128)]).all()
        np.testing.assert_allclose(lam[b, :4], w[:4], rtol=0, atol=1e-12 * w[0])
        assert done[b] == 1
        col0 = out[b][0]
        np.testing.assert_allclose(np.linalg.norm(col0), w[0], rtol=1e-12)
        v = col0 / np.linalg.norm(col0)
        assert np.abs(G @ v - w[0] * v).max() < 1e-11 * w[0]


def run_trdx(Gs, n, need, cert_all=0):
    """The blocked solver of orders 320 .. 1024 (csrc/trdx.hip)."""
    from mused_amd import _lib
    from mused_amd.engine import ptr, stream_ptr

    L = _lib.lib()
    fn = L.mused_debug_trdx
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6
    B = len(Gs)
    G = torch.from_numpy(np.ascontiguousarray(np.stack(Gs))).cuda()
    d = torch.zeros(B, n, dtype=torch.float64, device="cuda")
    e = torch.zeros(B, n, dtype=torch.float64, device="cuda")
    lam = torch.zeros(B, n // 2, dtype=torch.float64, device="cuda")
    res = torch.zeros(B, n // 2, dtype=torch.float64, device="cuda")
    done = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    _lib.check(fn(ptr(G), n, need, cert_all, B, ptr(d), ptr(e), ptr(lam), ptr(res), ptr(done), stream_ptr()))
    torch.cuda.synchronize()
    return G.cpu().numpy(), d.cpu().numpy(), e.cpu().numpy(), lam.cpu().numpy(), res.cpu().numpy(), done.cpu().numpy()


@pytest.mark.parametrize("n,need", [(320, 160), (384, 128), (448, 224), (512, 256), (512, 128), (640, 200), (768, 256),
                                    (896, 300), (1024, 256), (1024, 512)])
def test_blocked_direct_solver_matches_lapack(n, need):
    """Orders 320 .. 512 (config 3's rotations at l = 256: order 512, top 256; the sketch query at l = 128: order 384 / 512,
    top 128) and 640 .. 1024 (two matrix rows per thread, panels of 8; config 3's sketch query at l = 256: order 768 / 1024,
    top 256): blocked tridiagonalisation (the same T as the unblocked reduction, to rounding), leading eigenpairs against
    LAPACK, zero columns elsewhere, same certificate."""
    ell = n // 2
    # (d > n: a buffer of rank below its row count leaves the tail of T undetermined -- see below)
    Gs = fd_buffers("blob", 3, ell=ell, d=max(1024, n + 256)) + fd_buffers("fd", 2, ell=ell, d=max(700, n + 100), seed=1)
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, 90))
    Gs.append(B @ B.T)                                     # rank 90: most of the spectrum zero
    out, d, e, lam, res, done = run_trdx(Gs, n, need)
    nvec = 32 * ((need + 31) // 32)
    for b, G in enumerate(Gs):
        w = np.linalg.eigvalsh(G)[::-1]
        scale = np.abs(w).max()
        if b < len(Gs) - 1:   # (the entries of T of the rank-90 matrix are not determined once its rank is exhausted: two
            dr, er = sytd2_lower(G)   # backward-stable reductions drift apart there -- its eigenvalues are the check)
            np.testing.assert_allclose(d[b], dr, rtol=0, atol=1e-8 * scale)
            np.testing.assert_allclose(np.abs(e[b, : n - 1]), np.abs(er[: n - 1]), rtol=0, atol=1e-8 * scale)
        np.testing.assert_allclose(lam[b, :nvec], w[:nvec], rtol=0, atol=2e-13 * scale)
        assert done[b] == 1, (b, res[b, :need].max())
        cols = out[b].T
        nrm = np.linalg.norm(cols[:, :nvec], axis=0)
        np.testing.assert_allclose(nrm, np.maximum(w[:nvec], 0), rtol=0, atol=1e-11 * scale)
        assert not cols[:, nvec:].any()
        sig = (w[:need] > 0) & ((w[:need] - w[need - 1]) > 1e-10 * w[0])
        V = cols[:, :need][:, sig] / nrm[:need][sig]
        assert np.abs(V.T @ V - np.eye(V.shape[1])).max() < 1e-9
        assert np.abs(G @ V - V * w[:need][sig]).max() < 1e-10 * scale


def test_blocked_direct_solver_rejects_clusters_and_leaves_the_matrix():
    rng = np.random.default_rng(5)
    n = 384
    Q = np.linalg.qr(rng.standard_normal((n, n)))[0]
    w = np.r_[np.full(40, 7.0), np.linspace(6.0, 0.5, n - 40)]       # 40-fold eigenvalue above the cut
    G = (Q * w) @ Q.T
    G = 0.5 * (G + G.T)
    out, d, e, lam, res, done = run_trdx([G, np.zeros((n, n)), np.eye(n) * 2.0], n, 128)
    assert done[0] == 0 and np.array_equal(out[0], G)
    np.testing.assert_allclose(lam[0, :128], np.sort(w)[::-1][:128], rtol=0, atol=1e-12 * 7.0)
    assert done[1] == 1 and np.abs(out[1]).max() < 1e-290            # zero matrix: nothing significant
    assert done[2] == 1                                               # identity: nothing survives the shrink


@pytest.mark.parametrize("n", [320, 512, 640, 1024])
def test_blocked_direct_solver_spectra_and_scales(n):
    """The hybrid reduction (blocked columns, then the trailing 256 x 256 on the register-resident kernel) on spectra that stress
    the hand-over: geometric decay over 12 decades, a spectrum that is exhausted inside the blocked columns (rank 40: the tail
    block is rounding residue), one exhausted inside the tail (rank n - 100), and overall scales 1e-120 / 1e+120.  Every matrix is
    either certified with accurate leading pairs or rejected and left untouched."""
    rng = np.random.default_rng(7 * n)
    need = min(256, n // 2)
    nvec = 32 * ((need + 31) // 32)
    Q = np.linalg.qr(rng.standard_normal((n, n)))[0]

    def with_spectrum(w):
        G = (Q * w) @ Q.T
        return 0.5 * (G + G.T)

    geo = np.geomspace(1.0, 1e-12, n)
    lowrank = np.r_[np.linspace(5.0, 1.0, 40), np.zeros(n - 40)]
    tailrank = np.r_[np.linspace(9.0, 0.5, n - 100), np.zeros(100)]
    lin = np.linspace(4.0, 0.01, n)
    Gs = [with_spectrum(geo), with_spectrum(lowrank), with_spectrum(tailrank), with_spectrum(lin) * 1e-120,
          with_spectrum(lin) * 1e120]
    out, d, e, lam, res, done = run_trdx(Gs, n, need)
    for b, G in enumerate(Gs):
        if done[b] == 0:
            assert np.array_equal(out[b], G)
            continue
        w = np.linalg.eigvalsh(G)[::-1]
        scale = np.abs(w).max()
        assert np.isfinite(lam[b, :nvec]).all()
        np.testing.assert_allclose(lam[b, :nvec], w[:nvec], rtol=0, atol=5e-13 * scale)
        cols = out[b].T
        nrm = np.linalg.norm(cols[:, :nvec], axis=0)
        sig = (w[:need] > 0) & ((w[:need] - w[need - 1]) > 1e-10 * w[0])
        V = cols[:, :need][:, sig] / nrm[:need][sig]
        assert np.abs(V.T @ V - np.eye(V.shape[1])).max() < 1e-9
        assert np.abs(G @ V - V * w[:need][sig]).max() < 1e-10 * scale
    assert done[0] == 1 and done[3] == 1 and done[4] == 1      # (the rank-deficient ones may be rejected: zero clusters at the cut)
