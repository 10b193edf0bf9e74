"""CPU statement of the sequence-based sliding-window Frequent Directions sketch
(`SeqBasedSWFD`), the class mused imports from its `swfd` submodule.

TEST INFRASTRUCTURE ONLY (see oracle/mo_oracle.py for the import rule).

PARITY UNPINNED.  The reference's `swfd/` directory is an un-vendored git
submodule (/root/reference/.gitmodules:1-3 ->
github.com/kelaendi/sliding-window-sketching, pin unknown, no copy on disk), so
there is no source to restate and no fixture to check against.  What is known
is the call surface -- `SeqBasedSWFD(N=, R=, d=, sketch_dim=)`, `.fit(row)` with
row of shape (1, d), `.get()` -> 4-tuple whose element 0 is the 2-D sketch
(main.py:10,62,65-76).  The algorithm below is this repo's own specification of
that class, written from the published literature (Liberty 2013 Frequent
Directions; Ghashami et al. 2016; Yin et al. VLDB 2024 "DS-FD", sequence-based
model with row norms^2 in [1, R]) as laid out in SURVEY.md Appendix A.  The HIP
implementation is tested against THIS file; this file is tested through
size-independent properties (covariance error bound, expiry, batching
invariance) in tests/test_swfd_oracle.py.

Specification (every free constant of Appendix A fixed here)
-----------------------------------------------------------
* Levels j = 0 .. L-1, L = ceil(log2(max(R, 1))) + 1, dump threshold
  theta_j = 2^j * N / ell.  Every level keeps a MAIN and an AUX sketch; every
  row goes to every sketch.
* A sketch = kept rows K (<= ell - 1 rows, mutually orthogonal, "Sigma V^T"
  form) + a FIFO queue of snapshots (vector, timestamp), capacity 2*ell;
  when full the oldest snapshot is dropped and its timestamp recorded in
  `dropped_t`.
* Rows are numbered i = 1, 2, ...  An EPOCH is N consecutive rows
  (rows e*N+1 .. (e+1)*N).  Rows since the last rotation are PENDING (raw,
  shared by all sketches).  All sketches ROTATE together when the number of
  rows since the epoch start is a multiple of ell, and at the end of every
  epoch.  Rotation of one sketch at time t:
      expire snapshots with ts + N <= t
      [_, s, Vt] = svd([K ; pending]);  lam = s^2 (zero-padded to ell)
      delta = lam[ell-1];  s2_i = max(lam_i - delta, 0),  i < ell
      rows with s2_i <= 1e-10 * lam_0 are discarded
      rows with s2_i >= theta   -> snapshot (sqrt(s2_i) * Vt[i], t)   [dump]
      the others                -> new K  (in order of decreasing s2)
* At the first row of every epoch after the first: MAIN <- AUX, AUX <- empty
  (the residual of MAIN is thereby expired).
* get() at time `now`: lowest level whose MAIN sketch has dropped no snapshot
  that would still be inside the window (dropped_t + N <= now or nothing
  dropped; top level if none), stack its in-window snapshots, its K and the
  pending rows, one more shrink:  B = sqrt(max(lam - lam[ell-1], 0))[:ell] Vt[:ell]
  (rows with shrunk energy <= 1e-10 * lam_0 set to zero, as in a rotation: they are below what a Gram-based
  implementation resolves) zero-padded to (ell, d); every row's largest-magnitude entry made positive.
  Returns (B, sigma = row norms of B, level, delta).
"""
from __future__ import annotations

import math

import numpy as np

REL_TOL = 1e-10


def _shrink(M: np.ndarray, ell: int):
    """SVD of M and the FD shrink.  Returns (s2 (<=ell,), Vt rows (<=ell, d), lam0, delta)."""
    if M.shape[0] == 0:
        return np.zeros(0), np.zeros((0, M.shape[1])), 0.0, 0.0
    _, s, Vt = np.linalg.svd(M, full_matrices=False)
    lam = s * s
    delta = lam[ell - 1] if len(lam) >= ell else 0.0
    m = min(ell, len(lam))
    s2 = np.maximum(lam[:m] - delta, 0.0)
    return s2, Vt[:m], float(lam[0]), float(delta)


class _Sketch:
    __slots__ = ("K", "qv", "qt", "dropped_t", "theta", "cap")

    def __init__(self, d, theta, cap):
        self.K = np.zeros((0, d))
        self.qv = []
        self.qt = []
        self.dropped_t = 0
        self.theta = theta
        self.cap = cap

    def copy_from(self, other):
        self.K = other.K.copy()
        self.qv = [v.copy() for v in other.qv]
        self.qt = list(other.qt)
        self.dropped_t = other.dropped_t

    def clear(self, d):
        self.K = np.zeros((0, d))
        self.qv = []
        self.qt = []
        self.dropped_t = 0

    def expire(self, now, N):
        while self.qt and self.qt[0] + N <= now:
            self.qt.pop(0)
            self.qv.pop(0)

    def rotate(self, pending, t, N, ell):
        self.expire(t, N)
        M = np.vstack([self.K, pending]) if len(pending) else self.K
        s2, Vt, lam0, _ = _shrink(M, ell)
        keep = []
        tol = REL_TOL * lam0
        for i in range(len(s2)):
            if s2[i] <= tol:
                continue
            row = math.sqrt(s2[i]) * Vt[i]
            if s2[i] >= self.theta:
                if len(self.qt) == self.cap:
                    self.dropped_t = max(self.dropped_t, self.qt.pop(0))
                    self.qv.pop(0)
                self.qv.append(row)
                self.qt.append(t)
            else:
                keep.append(row)
        self.K = np.array(keep).reshape(len(keep), M.shape[1])


class SeqBasedSWFD:
    def __init__(self, N, R, d, sketch_dim):
        self.N = int(N)
        self.R = float(R)
        self.d = int(d)
        self.ell = int(sketch_dim)
        if self.N < 1 or self.d < 1 or self.ell < 1:
            raise ValueError("N, d and sketch_dim must be positive")
        self.L = int(math.ceil(math.log2(max(self.R, 1.0)))) + 1
        cap = 2 * self.ell
        self.theta = [(2.0 ** j) * self.N / self.ell for j in range(self.L)]
        self.main = [_Sketch(self.d, th, cap) for th in self.theta]
        self.aux = [_Sketch(self.d, th, cap) for th in self.theta]
        self.i = 0
        self.pending = np.zeros((0, self.d))

    # -- update ------------------------------------------------------------
    def _rotate_all(self):
        for sk in self.main + self.aux:
            sk.rotate(self.pending, self.i, self.N, self.ell)
        self.pending = np.zeros((0, self.d))

    def fit(self, X):
        X = np.asarray(X, dtype=np.float64)
        if X.ndim == 1:
            X = X[None, :]
        if X.shape[1] != self.d:
            raise ValueError(f"expected rows of length {self.d}, got {X.shape[1]}")
        for r in range(X.shape[0]):
            self.i += 1
            if self.i > 1 and (self.i - 1) % self.N == 0:
                for m, a in zip(self.main, self.aux):
                    m.copy_from(a)
                    a.clear(self.d)
            self.pending = np.vstack([self.pending, X[r : r + 1]])
            in_epoch = self.i - ((self.i - 1) // self.N) * self.N
            if in_epoch % self.ell == 0 or self.i % self.N == 0:
                self._rotate_all()
        return self

    # -- query -------------------------------------------------------------
    def select_level(self):
        now = self.i
        for j, sk in enumerate(self.main):
            if sk.dropped_t == 0 or sk.dropped_t + self.N <= now:
                return j
        return self.L - 1

    def stacked(self, j=None):
        """The rows the query compresses: in-window snapshots, K, pending."""
        if j is None:
            j = self.select_level()
        sk = self.main[j]
        now = self.i
        snaps = [v for v, t in zip(sk.qv, sk.qt) if t + self.N > now]
        parts = []
        if snaps:
            parts.append(np.array(snaps))
        parts.append(sk.K)
        parts.append(self.pending)
        return np.vstack(parts)

    def get(self):
        j = self.select_level()
        M = self.stacked(j)
        s2, Vt, lam0, delta = _shrink(M, self.ell)
        B = np.zeros((self.ell, self.d))
        if len(s2):
            s2 = np.where(s2 > REL_TOL * lam0, s2, 0.0)  # the discard rule of the rotation, applied to the query too
            B[: len(s2)] = np.sqrt(s2)[:, None] * Vt
        idx = np.argmax(np.abs(B), axis=1)
        sg = np.sign(B[np.arange(self.ell), idx])
        sg[sg == 0] = 1.0
        B *= sg[:, None]
        sigma = np.sqrt(np.einsum("ij,ij->i", B, B))
        return B, sigma, j, delta
