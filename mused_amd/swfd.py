"""MI355X-native `SeqBasedSWFD`: the class mused imports with `from swfd import SeqBasedSWFD`
(main.py:10) and drives as

    swfd = SeqBasedSWFD(N=window_size, R=max_norm, d=fused_matrix.shape[1], sketch_dim=reduced_dim)   # :62
    swfd.fit(row)            # row of shape (1, d), once per row of the window                             # :65-67
    reduced_matrix, _, _, _ = swfd.get()                                                                   # :70

The reference's implementation is an un-vendored submodule, so the algorithm is
this repo's specification (oracle/swfd_oracle.py documents it); state and arithmetic live on the
GPU in libmused_hip (mused_amd/csrc/swfd.hip).  No CPU fallback.

`fit` accepts (b, d) blocks of any b >= 1, NumPy (float/int, e.g. the int64 fused matrix of
matrix_operations.py:138) or torch tensors (CUDA tensors are appended without a host round trip).
Host rows are staged and shipped in blocks; the sketch does not depend on the blocking.
`get()` returns NumPy arrays like the reference; `get_device()` returns CUDA tensors.
"""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np
import torch

from . import _lib
from ._lib import BITS, F32, F64, I64, MusedError, call
from .engine import _require_gpu, ptr, stream_ptr

_DT = {torch.float32: F32, torch.float64: F64, torch.int64: I64}


class SeqBasedSWFD:
    def __init__(self, N, R, d, sketch_dim, *, stage_rows: int = 1024, sweeps: int = 0, device="cuda", lanes: int = 1):
        _require_gpu()
        self.N, self.R, self.d, self.ell = int(N), float(R), int(d), int(sketch_dim)
        if self.N < 1 or self.d < 1 or self.ell < 1:
            raise ValueError("N, d and sketch_dim must be positive")
        self.device = device
        self._h = C.c_void_p()
        # lanes > 1: that many independent sketch sets (e.g. the windows of several contiguous blocks of
        # the stream) advanced in lockstep by the same kernel launches -- `fit_lanes` / `get_device`
        self.lanes = int(lanes)
        call("mused_swfd_create_lanes", self.N, self.R, self.d, self.ell, int(sweeps), self.lanes, C.byref(self._h))
        self.L = _lib.lib().mused_swfd_levels(self._h)
        self._stage = np.empty((max(int(stage_rows), 1), self.d), dtype=np.float64)
        self._staged = 0

    # -- update ---------------------------------------------------------------------------
    def _flush(self):
        if self._staged:
            t = torch.from_numpy(self._stage[: self._staged]).to(self.device)
            call("mused_swfd_append", self._h, ptr(t), F64, self._staged, self.d, stream_ptr())
            # `t` may be released right away: the kernels reading it were enqueued on torch's current
            # stream, and torch's allocator reuses a block only in stream order.
            self._staged = 0

    def fit_lanes(self, X: torch.Tensor):
        """X: CUDA tensor (lanes, n, d) (any strides with unit stride along d): n rows for every lane."""
        if X.dim() != 3 or X.shape[0] != self.lanes or X.shape[2] != self.d or not X.is_cuda:
            raise ValueError(f"expected a CUDA tensor of shape ({self.lanes}, n, {self.d})")
        if X.dtype not in _DT:
            X = X.to(torch.float64)
        if X.stride(2) != 1:
            X = X.contiguous()
        call("mused_swfd_append_lanes", self._h, ptr(X), _DT[X.dtype], X.shape[1], X.stride(1), X.stride(0), stream_ptr())
        self._keepalive = X
        return self

    def fit_adjacency(self, adj):
        """Append the rows of a device-resident 0/1 adjacency (`engine.Adjacency`, n x words bitmask) -- what the
        reference's SWFDMC approach does row by row with the dense fused matrix (main.py:65-67); the W x W matrix is
        never materialised: the append kernels expand bits straight into the sketch buffers."""
        if self.lanes != 1:
            raise ValueError("multi-lane sketch: use fit_lanes")
        if adj.n != self.d:
            raise ValueError(f"expected rows of length {self.d}, got {adj.n}")
        self._flush()
        call("mused_swfd_append", self._h, ptr(adj.mask), BITS, adj.n, adj.words, stream_ptr())
        self._keepalive = adj.mask
        return self

    def fit_adjacency_lanes(self, masks: torch.Tensor):
        """`fit_adjacency` for every lane at once: masks = (lanes, n, words) int64 CUDA tensor, the bitmask rows of one 0/1
        adjacency per lane (n = d rows each; an all-zero mask is a window of empty rows -- the halo of a lane whose block
        starts at the beginning of the stream).  The lanes advance in lock-step inside the same launches."""
        if masks.dim() != 3 or masks.shape[0] != self.lanes or masks.shape[1] != self.d or not masks.is_cuda or masks.dtype != torch.int64:
            raise ValueError(f"expected an int64 CUDA tensor of shape ({self.lanes}, {self.d}, words)")
        if masks.shape[2] * 64 < self.d:
            raise ValueError("bit rows shorter than d")
        if not masks.is_contiguous():
            masks = masks.contiguous()
        call("mused_swfd_append_lanes", self._h, ptr(masks), BITS, masks.shape[1], masks.stride(1), masks.stride(0), stream_ptr())
        self._keepalive = masks
        return self

    def fit(self, X):
        if self.lanes != 1:
            raise ValueError("multi-lane sketch: use fit_lanes")
        if isinstance(X, torch.Tensor) and X.is_cuda:
            self._flush()
            t = X if X.dim() == 2 else X.reshape(1, -1)
            if t.dtype not in _DT:
                t = t.to(torch.float64)
            if t.shape[1] != self.d:
                raise ValueError(f"expected rows of length {self.d}, got {t.shape[1]}")
            if t.stride(1) != 1:
                t = t.contiguous()
            call("mused_swfd_append", self._h, ptr(t), _DT[t.dtype], t.shape[0], t.stride(0), stream_ptr())
            self._keepalive = t
            return self
        a = np.asarray(X)
        if a.ndim == 1:
            a = a[None, :]
        if a.shape[1] != self.d:
            raise ValueError(f"expected rows of length {self.d}, got {a.shape[1]}")
        r = 0
        while r < a.shape[0]:
            take = min(a.shape[0] - r, self._stage.shape[0] - self._staged)
            self._stage[self._staged : self._staged + take] = a[r : r + take]
            self._staged += take
            r += take
            if self._staged == self._stage.shape[0]:
                self._flush()
        return self

    # -- query ------------------------------------------------------------------------------
    def get_device(self):
        """(sketch (l, d), sigma (l,), info (2,) = [level, delta]) as fp64 CUDA tensors."""
        self._flush()
        lead = () if self.lanes == 1 else (self.lanes,)
        B = torch.empty(lead + (self.ell, self.d), dtype=torch.float64, device=self.device)
        sig = torch.empty(lead + (self.ell,), dtype=torch.float64, device=self.device)
        info = torch.empty(lead + (2,), dtype=torch.float64, device=self.device)
        call("mused_swfd_query", self._h, ptr(B), ptr(sig), ptr(info), stream_ptr())
        return B, sig, info

    def get(self):
        """4-tuple like the reference's get(): (sketch (l, d) float64, singular values of the sketch,
        level used, delta of the final shrink); main.py:70 consumes element 0 only."""
        B, sig, info = self.get_device()
        self.check()
        info = info.cpu().numpy()
        if self.lanes != 1:
            return B.cpu().numpy(), sig.cpu().numpy(), info[:, 0].astype(int), info[:, 1]
        return B.cpu().numpy(), sig.cpu().numpy(), int(info[0]), float(info[1])

    def check(self):
        """BLOCKING: raise if an eigensolve of this sketch gave up (its results since then are invalid).  `get()` calls
        it; callers of the asynchronous `get_device()` call it once the stream has run (the pipeline does in flush())."""
        st = C.c_int()
        call("mused_swfd_status", self._h, C.byref(st), stream_ptr())
        if st.value:
            raise MusedError(f"SeqBasedSWFD: an eigensolve of a rotation / query gave up (status {st.value}): "
                             "the sketch is invalid")

    # -- live timing of the rotation eigensolver ----------------------------------------------------
    def profile(self, on: bool):
        call("mused_swfd_profile", self._h, 1 if on else 0)

    def profile_read(self):
        """(summed ms of the Jacobi sweep graphs, osj_round_kernel launches covered, bytes per launch)."""
        ms, n, b = C.c_double(), C.c_long(), C.c_double()
        call("mused_swfd_profile_read", self._h, C.byref(ms), C.byref(n), C.byref(b))
        return ms.value, n.value, b.value

    def profile_read_direct(self):
        """(direct, summed ms of the direct-eigensolver launches, launches, matrices solved, ms of that inside the
        tridiagonalisation kernel) -- direct False: the rotations
        of this sketch run the Jacobi (order != 256 or MUSED_EIG_TRD=0), use profile_read()."""
        ms, n, m, d, ta = C.c_double(), C.c_long(), C.c_double(), C.c_int(), C.c_double()
        call("mused_swfd_profile_read_direct", self._h, C.byref(ms), C.byref(n), C.byref(m), C.byref(d), C.byref(ta))
        return bool(d.value), ms.value, n.value, m.value, ta.value

    # -- bookkeeping / multi-GPU state exchange ---------------------------------------------------
    @property
    def rows_seen(self) -> int:
        i, p = C.c_long(), C.c_int()
        call("mused_swfd_counters", self._h, C.byref(i), C.byref(p))
        return int(i.value) + self._staged

    def half_bytes(self) -> int:
        return int(_lib.lib().mused_swfd_half_bytes(self._h))

    def export_half(self, kind: int) -> torch.Tensor:
        """Pack the MAIN (0) or AUX (1) half of the state into a uint8 CUDA tensor."""
        self._flush()
        blob = torch.empty(self.half_bytes(), dtype=torch.uint8, device=self.device)
        call("mused_swfd_export_half", self._h, int(kind), ptr(blob), stream_ptr())
        return blob

    def begin_epoch(self, rows_seen: int, main_half: torch.Tensor | None = None):
        """Start the epoch that begins after `rows_seen` rows (a multiple of N) with MAIN taken from
        `main_half` (the AUX half of whoever sketched the previous window) and AUX empty."""
        self._flush()
        if main_half is not None and main_half.numel() != self.half_bytes():
            raise ValueError("state blob has the wrong size")
        call("mused_swfd_begin_epoch", self._h, int(rows_seen), ptr(main_half) if main_half is not None else None,
             stream_ptr())
        self._keepalive = main_half

    def close(self):
        if getattr(self, "_h", None):
            call("mused_swfd_destroy", self._h)
            self._h = None

    def __del__(self):
        if sys is None or sys.is_finalizing():  # interpreter exit: the HIP runtime may be gone already
            return
        try:
            self.close()
        except Exception:
            pass
