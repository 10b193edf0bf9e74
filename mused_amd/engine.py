"""Device-resident window engine: the MI355X hot path behind the reference's Python call surface.

One `WindowEngine` owns the persistent HBM workspaces for windows of up to `n_max` rows:

    knn_ws   candidate lists of the fused similarity / top-k kernels: n x cap (score, column) + thresholds
             (85 MB at n = 10^4, cap = 704; mused_amd/csrc/knn_fused.hip).  No n x n score matrix.
    masks    per-modality and fused adjacency BITMASKS, n x ceil(n/64) uint64 (12.5 MB at n = 10^4)
    rsvd     handle of the randomized-SVD eigenstep (CSR lists, n x (l+10) panels, hipGraph)
    scores   n x n fp64 (800 MB at n = 10^4): allocated ONLY if the classic path is asked for (MUSED_KNN=classic,
             unit tests of the primitives) or a window overflows the candidate lists (pathological inputs)

Nothing W x W and dense ever leaves the device unless the NumPy-compatible wrappers in
`mused_amd.matrix_operations` ask for it.  torch is used for device memory and streams only;
all arithmetic is in libmused_hip (ctypes, raw pointers).
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import sys

import numpy as np
import torch

from . import _lib
from ._lib import F32, F64, I64, METRIC_COSINE, METRIC_L2, MusedError, call

_DT = {torch.float32: F32, torch.float64: F64, torch.int64: I64}
_FUSED_META_ROWS = 15000  # mused_record_knn (<= 16384) / mused_jaccard_knn (<= 15000) keep one row of scores in LDS


def _require_gpu():
    _lib.lib()  # raises if the extension is not built
    if not torch.cuda.is_available():
        raise MusedError("no HIP device visible: the mused_amd hot path has no CPU fallback")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t: torch.Tensor):
    return C.c_void_p(t.data_ptr())


def words_for(n: int) -> int:
    return (n + 63) // 64


def to_device_rows(data, device="cuda") -> torch.Tensor:
    """Rows as a device tensor with unit stride along the row (any row pitch: a column slice of a wider
    resident matrix is used in place), keeping float32 / float64 (anything else -> float64)."""
    if isinstance(data, torch.Tensor):
        t = data
        if t.dtype not in (torch.float32, torch.float64):
            t = t.to(torch.float64)
        t = t.to(device)
        return t if (t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]) else t.contiguous()
    a = np.asarray(data)
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64)
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


class Adjacency:
    """A directed 0/1 adjacency held as an n x words uint64 bitmask on the device.

    bit j of row i set  <=>  A[i, j] = 1   (matrix_operations.py:123-130: j among the selected
    neighbours of i, j != i).  `fused` records which dtype the reference would return
    (float64 copy for one modality, int64 after a logical_or; matrix_operations.py:135,138).
    """

    def __init__(self, mask: torch.Tensor, n: int, fused: bool = False):
        self.mask = mask  # (n, words) int64 tensor viewed as uint64 words
        self.n = n
        self.words = mask.shape[1]
        self.fused = fused

    def to_dense(self, dtype=None) -> torch.Tensor:
        if dtype is None:
            dtype = torch.int64 if self.fused else torch.float64
        out = torch.empty((self.n, self.n), dtype=dtype, device=self.mask.device)
        call("mused_adj_to_dense", ptr(self.mask), self.n, self.words, _DT[dtype], ptr(out), stream_ptr())
        return out

    def to_numpy(self) -> np.ndarray:
        return self.to_dense().cpu().numpy()

    def degrees(self):
        deg = torch.empty(self.n, dtype=torch.int32, device=self.mask.device)
        rowptr = torch.empty(self.n + 1, dtype=torch.int32, device=self.mask.device)
        stats = torch.empty(2, dtype=torch.int32, device=self.mask.device)
        call("mused_adj_degrees", ptr(self.mask), self.n, self.words, ptr(deg), ptr(rowptr), ptr(stats), stream_ptr())
        return deg, rowptr, stats

    def neighbour_lists(self):
        """(rowptr, colidx) int32 device tensors, columns ascending."""
        deg, rowptr, stats = self.degrees()
        nnz = int(stats[1].item())
        colidx = torch.empty(max(nnz, 1), dtype=torch.int32, device=self.mask.device)
        call("mused_adj_csr_fill", ptr(self.mask), self.n, self.words, ptr(rowptr), ptr(colidx), stream_ptr())
        return rowptr, colidx[:nnz]

    @staticmethod
    def from_dense(dense, device="cuda") -> "Adjacency":
        """Import an n x n 0/1 matrix (NumPy or torch).  Raises on entries other than 0/1."""
        if isinstance(dense, torch.Tensor):
            t = dense.to(device)
        else:
            a = np.asarray(dense)
            if a.dtype == np.bool_:
                a = a.astype(np.int64)
            t = torch.from_numpy(np.ascontiguousarray(a)).to(device)
        if t.dtype not in _DT:
            t = t.to(torch.float64)
        t = t.contiguous()
        if t.dim() != 2 or t.shape[0] != t.shape[1]:
            raise ValueError("adjacency must be a square 2-D matrix")
        n = t.shape[0]
        w = words_for(n)
        mask = torch.empty((n, w), dtype=torch.int64, device=device)
        flag = torch.zeros(1, dtype=torch.int32, device=device)
        call("mused_adj_from_dense", ptr(t), _DT[t.dtype], n, n, w, ptr(mask), ptr(flag), stream_ptr())
        if int(flag.item()):
            raise NotImplementedError(
                "only 0/1 adjacency matrices are supported on the device path "
                "(the reference only ever passes the fused kNN adjacency, main.py:56,79)"
            )
        return Adjacency(mask, n, fused=(t.dtype == torch.int64))


class WindowEngine:
    """Workspaces + ops for windows of at most n_max rows (see module docstring)."""

    def __init__(self, n_max: int, device="cuda"):
        _require_gpu()
        self.n_max = int(n_max)
        self.device = device
        self._scores = None  # classic path only, allocated on first use
        self.norms = torch.empty(self.n_max, dtype=torch.float64, device=device)
        self.knn_mode = os.environ.get("MUSED_KNN", "fused")
        self._knn_ws = None
        self._knn_cap = 0
        self._ovf = torch.zeros(1, dtype=torch.int32, device=device)
        # DEFERRED flags (StreamPipeline): between begin_window(defer=True) and the next begin_window the overflow word of
        # the i-th kNN call goes to _ovf8[i] and is NOT read here -- the caller copies it to the host behind the window's
        # event together with the eigenstep's flags and repeats a flagged window on a fallback engine (no host read on
        # the per-window path)
        self._ovf8 = torch.zeros(8, dtype=torch.int32, device=device)
        self.defer = False
        self._defer_slot = 0
        self.knn_fallbacks = 0  # windows redone on the classic path because a candidate list overflowed
        self._hop_lock = threading.Lock()
        self.slot_keys = []
        self._hop = {}          # hopping windows: key -> workspace + stream position of the window it holds (knn_adjacency_hop)
        self.hop_windows = self.hop_reused = self.hop_recomputes = 0
        self._rsvd = None
        self._rsvd_fb = None       # fallback handle (mode 2), created on first use
        self._rsvd_fb_key = None
        self.rsvd_fallbacks = 0    # windows repeated on the LU / Householder chain after a weak Cholesky pivot
        # MUSED_RSVD_MODE: "cholqr" (default: Cholesky-QR, host-side fallback), "graph" (Cholesky-QR with the Householder
        # fallback recorded in the graph), "lu" (the reference's chain)
        self.rsvd_mode = os.environ.get("MUSED_RSVD_MODE", "cholqr")
        self._rsvd_key = None
        self._rsvd_cap = 0
        self._q0_key = None
        # optional live timing: set to a list and every score-GEMM launch is bracketed by HIP events
        # recorded on the launch stream -> [(start_event, end_event), ...]
        self.score_events = None

    @property
    def scores(self) -> torch.Tensor:
        """n_max x n_max fp64 score workspace of the CLASSIC path (mused_pairwise_scores + mused_select_k_smallest)."""
        if self._scores is None:
            self._scores = torch.empty(self.n_max * self.n_max, dtype=torch.float64, device=self.device)
        return self._scores

    def begin_window(self, defer: bool):
        """Start a window; defer=True: overflow / weak-pivot flags are left on the device for the caller (`window_flags`)."""
        self.defer = bool(defer)
        self._defer_slot = 0
        self.slot_keys = []  # deferred flag word i of this window belongs to hop state slot_keys[i] (None: a plain kNN call)
        if self.defer:
            self._ovf8.zero_()

    def window_flags(self, rsvd_flags) -> torch.Tensor:
        """int32[12] device tensor = [the eigenstep's 4 flags | the overflow words of the window's kNN calls], taken on
        the current stream (two small device-to-device copies)."""
        out = torch.empty(12, dtype=torch.int32, device=self.device)
        _hip_memcpy_d2d(out.data_ptr(), rsvd_flags.data_ptr(), 16)
        _hip_memcpy_d2d(out.data_ptr() + 16, self._ovf8.data_ptr(), 32)
        return out

    def _fused_ws(self, kk: int):
        # a phase admits about twice the k candidates already held (it shows twice the columns seen so far): 4 k + 128
        # leaves a margin of two; beyond 1024 per row (k > 224) the classic path is used instead
        cap = max(704, ((4 * kk + 128 + 63) // 64) * 64)  # 704: room for the 5 x 128 scores of the first phase
        if int(os.environ.get("MUSED_KNN_CAP", "0")) > 0:
            cap = int(os.environ["MUSED_KNN_CAP"])
        if self._knn_ws is None or self._knn_cap != cap:
            nbytes = int(_lib.lib().mused_knn_fused_ws_bytes(self.n_max, cap))
            self._knn_ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._knn_cap = cap
        return self._knn_ws, cap

    # ---- a1 / a2 ------------------------------------------------------------------------
    def knn_adjacency(self, rows, k: int, metric: str = "l2", want_idx: bool = False):
        """Adjacency of the k selected rows per row (self removed).  `rows`: (n, d) float32/float64
        device tensor with finite entries.  metric "l2": k = max(1, k_basis) nearest incl. self
        (matrix_operations.py:113-119); "cosine": k_basis + 1 most similar (:93,108)."""
        X = to_device_rows(rows, self.device)
        n, d = X.shape
        if n > self.n_max:
            raise ValueError(f"window of {n} rows exceeds the engine capacity {self.n_max}")
        if metric == "l2":
            kk, m = max(1, int(k)), METRIC_L2
            if kk > n:
                # same error as sklearn's kneighbors (neighbors/_base.py)
                raise ValueError(
                    f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {kk}, "
                    f"n_samples_fit = {n}, n_samples = {n}"
                )
        elif metric == "cosine":
            kk, m = min(int(k) + 1, n), METRIC_COSINE
        else:
            raise ValueError(f"unknown metric {metric!r}")
        w = words_for(n)
        mask = torch.empty((n, w), dtype=torch.int64, device=self.device)
        idx = torch.empty((n, kk), dtype=torch.int32, device=self.device) if want_idx else None
        if self.score_events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        fused = self.knn_mode != "classic" and 4 * kk + 128 <= 1024
        if fused:
            ws, cap = self._fused_ws(kk)
            fused = kk <= cap <= 1024
        if fused:
            # similarity + selection in one pass over the tiles, no n x n score matrix (knn_fused.hip)
            deferred = self.defer and self._defer_slot < 8
            ovf_ptr = C.c_void_p(self._ovf8.data_ptr() + 4 * self._defer_slot) if deferred else ptr(self._ovf)
            call("mused_knn_fused", ptr(X), _DT[X.dtype], n, d, X.stride(0), kk, m, ptr(ws), ws.numel(), cap,
                 ptr(idx) if want_idx else None, ptr(mask), w, ovf_ptr, stream_ptr())
            if self.score_events is not None:
                e1.record()
                self.score_events.append((e0, e1))
            if deferred:
                self._defer_slot += 1  # the caller reads the word behind the window (window_flags)
                self.slot_keys.append(None)
            elif int(self._ovf.item()) != 0:  # one small blocking read: direct callers get a finished result
                self.knn_fallbacks += 1
                fused = False
        if not fused:
            call("mused_pairwise_scores", ptr(X), _DT[X.dtype], n, d, X.stride(0), m, ptr(self.norms), ptr(self.scores),
                 stream_ptr())
            if self.score_events is not None and self.knn_mode == "classic":
                e1.record()
                self.score_events.append((e0, e1))
            call("mused_select_k_smallest", ptr(self.scores), n, n, kk, ptr(idx) if want_idx else None, ptr(mask), w,
                 stream_ptr())
        adj = Adjacency(mask, n)
        return (adj, idx) if want_idx else adj

    def request_hop_reset(self, key) -> bool:
        """The flag word of hop state `key` was raised (read behind the window): its next call starts from scratch.  Called from
        the thread that reads the flags; a counter, so that a request can never be lost to the enqueueing thread's clear."""
        st = self._hop.get(key)
        if st is None:
            return False
        with self._hop_lock:
            st["reset_req"] = st.get("reset_req", 0) + 1
        return True

    # ---- f3: hopping windows with reuse across consecutive windows -----------------------
    def knn_adjacency_hop(self, rows, k: int, metric: str, key, lo: int):
        """`knn_adjacency` for the window whose first row is stream row `lo`, reusing the candidate lists the previous call
        with the same `key` (one per modality of a stream) left behind when that window overlaps this one
        (step_window_ratio > 1, main.py:32): only the tiles that involve an entering row are computed.  Falls back to a
        computation from scratch when there is nothing to reuse, and repeats from scratch when the device reports that a
        kept list no longer proves a row's k smallest (one small blocking read per window in this mode).  Bit-identical to
        `knn_adjacency` on the same rows."""
        X = to_device_rows(rows, self.device)
        n, d = X.shape
        if metric == "l2":
            kk, m = max(1, int(k)), METRIC_L2
            if kk > n:
                raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {kk}, n_samples_fit = {n}, "
                                 f"n_samples = {n}")
        else:
            kk, m = min(int(k) + 1, n), METRIC_COSINE
        cap = 1024  # the kept lists follow the (3 k)-th score: room for 3 k + what a phase admits
        if self.knn_mode == "classic" or 4 * kk + 128 > 1024:
            return self.knn_adjacency(X, k, metric)
        st = self._hop.get(key)
        if st is None or st["n"] != n or st["cap"] != cap or st["d"] != d or st["kk"] != kk:
            nbytes = int(_lib.lib().mused_knn_fused_ws_bytes(n, cap))
            st = self._hop[key] = dict(ws=torch.empty(nbytes, dtype=torch.uint8, device=self.device), n=n, cap=cap, d=d, kk=kk,
                                       lo=None, flag=torch.zeros(1, dtype=torch.int32, device=self.device))
        n_new = 0
        # a reader of the deferred flag words asks for a rebuild by raising st["reset_req"] (request_hop_reset: the label
        # workers' thread); this thread only ever copies it -- a request between the read and the copy is seen next window
        req = st.get("reset_req", 0)
        if st["lo"] is not None and req == st.get("reset_seen", 0) and 0 < lo - st["lo"] < n:
            n_new = lo - st["lo"]
        st["reset_seen"] = req
        w = words_for(n)
        mask = torch.empty((n, w), dtype=torch.int64, device=self.device)
        deferred = self.defer and self._defer_slot < 8
        if deferred:
            # the flag word (sticky on the device until the state is rebuilt) goes to the window's flag words; whoever reads
            # them later repeats a flagged window elsewhere and calls request_hop_reset, which makes the next call start from scratch
            fptr = C.c_void_p(self._ovf8.data_ptr() + 4 * self._defer_slot)
            self._defer_slot += 1
            self.slot_keys.append(key)
            call("mused_knn_fused_hop", ptr(X), _DT[X.dtype], n, d, X.stride(0), kk, m, ptr(st["ws"]), st["ws"].numel(), cap,
                 int(lo), int(n_new), None, ptr(mask), w, fptr, stream_ptr())
            st["lo"] = int(lo)
            self.hop_windows += 1
            self.hop_reused += 1 if n_new > 0 else 0
            return Adjacency(mask, n)
        for attempt in range(2):
            call("mused_knn_fused_hop", ptr(X), _DT[X.dtype], n, d, X.stride(0), kk, m, ptr(st["ws"]), st["ws"].numel(), cap,
                 int(lo), int(n_new), None, ptr(mask), w, ptr(st["flag"]), stream_ptr())
            flag = int(st["flag"].item())
            if flag == 0:
                break
            if n_new == 0:  # from scratch and still flagged: a list overflowed (thousands of equal scores) -> classic path
                st["lo"] = None
                self.knn_fallbacks += 1
                return self.knn_adjacency(X, k, metric)
            self.hop_recomputes += 1
            n_new = 0
        st["lo"] = int(lo)
        self.hop_windows += 1
        self.hop_reused += 1 if (attempt == 0 and n_new > 0) else 0
        return Adjacency(mask, n)

    # ---- a1, metadata modality types (csrc/meta.hip) ------------------------
    def _select(self, n: int, kk: int) -> Adjacency:
        w = words_for(n)
        mask = torch.empty((n, w), dtype=torch.int64, device=self.device)
        call("mused_select_k_smallest", ptr(self.scores), n, n, kk, None, ptr(mask), w, stream_ptr())
        return Adjacency(mask, n)

    def record_adjacency(self, records, kind: str, kk: int) -> Adjacency:
        """The kk closest rows per row (self removed) of n x 2 records: kind "location" = haversine km between
        (latitude, longitude) pairs (matrix_operations.py:22-31, 250-263), "time" = |d datetaken| + |d dateupload|
        (:33-54).  Ties go to the smaller row index."""
        rec = torch.as_tensor(np.ascontiguousarray(records, dtype=np.float64)).to(self.device)
        n = rec.shape[0]
        if n > self.n_max or rec.shape[1] != 2:
            raise ValueError(f"{kind}: need (n <= {self.n_max}) x 2 records, got {tuple(rec.shape)}")
        kcode = {"location": 0, "time": 1}[kind]
        if n <= _FUSED_META_ROWS and self.knn_mode != "classic":
            # scores of a row computed into LDS by the selection kernel itself: no n x n score matrix
            w = words_for(n)
            mask = torch.empty((n, w), dtype=torch.int64, device=self.device)
            call("mused_record_knn", ptr(rec), n, kcode, kk, None, ptr(mask), w, stream_ptr())
            return Adjacency(mask, n)
        call("mused_record_scores", ptr(rec), n, kcode, ptr(self.scores), stream_ptr())
        return self._select(n, kk)

    def jaccard_adjacency(self, rowptr, tags, n_tags: int, kk: int) -> Adjacency:
        """The kk rows with the largest Jaccard similarity of tag sets per row (matrix_operations.py:73-89); sets as CSR
        (int32 rowptr[n + 1], tags[]: ids < n_tags, unique inside a row).  Ties go to the smaller row index."""
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        tags = np.ascontiguousarray(tags, dtype=np.int32)
        n = len(rowptr) - 1
        if n > self.n_max:
            raise ValueError(f"window of {n} rows exceeds the engine capacity {self.n_max}")
        rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(rowptr))
        order = np.argsort(tags, kind="stable")  # posting lists: rows of a tag in ascending order
        postrow = rows[order]
        postptr = np.concatenate([[0], np.cumsum(np.bincount(tags, minlength=max(n_tags, 1)))]).astype(np.int32)
        dev = [torch.from_numpy(a).to(self.device) for a in (rowptr, tags if len(tags) else np.zeros(1, np.int32),
                                                              postptr, postrow if len(postrow) else np.zeros(1, np.int32))]
        if n <= _FUSED_META_ROWS and self.knn_mode != "classic":
            w = words_for(n)
            mask = torch.empty((n, w), dtype=torch.int64, device=self.device)
            call("mused_jaccard_knn", ptr(dev[0]), ptr(dev[1]), ptr(dev[2]), ptr(dev[3]), n, int(n_tags), kk, None, ptr(mask),
                 w, stream_ptr())
            self._keep = dev  # the CSR arrays stay alive until the stream has consumed them (next call replaces them)
            return Adjacency(mask, n)
        call("mused_jaccard_scores", ptr(dev[0]), ptr(dev[1]), ptr(dev[2]), ptr(dev[3]), n, int(n_tags), ptr(self.scores),
             stream_ptr())
        return self._select(n, kk)

    def group_adjacency(self, ids) -> Adjacency:
        """A[i][j] = 1 iff ids[i] == ids[j] >= 0 and i != j (matrix_operations.py:56-71: same user name)."""
        ids_d = torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int32)).to(self.device)
        n = ids_d.numel()
        w = words_for(n)
        mask = torch.empty((n, w), dtype=torch.int64, device=self.device)
        call("mused_group_mask", ptr(ids_d), n, ptr(mask), w, stream_ptr())
        return Adjacency(mask, n)

    # ---- a3 / a4 ------------------------------------------------------------------------
    def fuse(self, adjs) -> Adjacency:
        n, w = adjs[0].n, adjs[0].words
        for a in adjs:
            if a.n != n or a.words != w:
                raise ValueError("adjacency shapes differ")
        out = torch.empty((n, w), dtype=torch.int64, device=self.device)
        arr = (C.c_void_p * len(adjs))(*[a.mask.data_ptr() for a in adjs])
        call("mused_adj_fuse", arr, len(adjs), n, w, ptr(out), stream_ptr())
        return Adjacency(out, n, fused=len(adjs) > 1)

    @staticmethod
    def max_row_sq_norm(adj: Adjacency) -> float:
        """main.py:61 for a 0/1 matrix: max_i ||row_i||^2 = largest out-degree.  Blocking (.item())."""
        _, _, stats = adj.degrees()
        return float(stats[0].item())

    # ---- a8 -------------------------------------------------------------------------------
    def _rsvd_handle(self, n: int, r: int, nnz_cap: int):
        if self._rsvd is None or self._rsvd_key[0] < n or self._rsvd_key[1] < r or self._rsvd_cap < nnz_cap:
            if self._rsvd is not None:
                call("mused_rsvd_destroy", self._rsvd)
            h = C.c_void_p()
            n_alloc = max(n, self.n_max)
            call("mused_rsvd_create", n_alloc, r, nnz_cap, 0, C.byref(h))
            # Cholesky-QR without the recorded Householder fallback (~690 no-op launches per call): svd_reduce reads the
            # weak-pivot flag and repeats the rare rank-deficient window on a handle that runs the reference's chain
            call("mused_rsvd_set_mode", h, 1 if self.rsvd_mode == "cholqr" else 2 if self.rsvd_mode == "lu" else 0)
            self._rsvd, self._rsvd_key, self._rsvd_cap = h, (n_alloc, r), nnz_cap
            self._q0_key = None
        return self._rsvd

    def _rsvd_fallback_handle(self, n: int, r: int, nnz_cap: int):
        """Handle in mode 2 (LU normaliser + Householder QR, no graph capture) for windows the Cholesky-QR path flags."""
        if self._rsvd_fb is None or self._rsvd_fb_key[0] < n or self._rsvd_fb_key[1] < r or self._rsvd_fb_key[2] < nnz_cap:
            if self._rsvd_fb is not None:
                call("mused_rsvd_destroy", self._rsvd_fb)
            h = C.c_void_p()
            call("mused_rsvd_create", max(n, self.n_max), r, nnz_cap, 0, C.byref(h))
            call("mused_rsvd_set_mode", h, 2)
            self._rsvd_fb, self._rsvd_fb_key = h, (max(n, self.n_max), r, nnz_cap)
        return self._rsvd_fb

    def svd_reduce(self, adj: Adjacency, reduced_dim: int, seed: int, n_iter: int = 5, n_oversamples: int = 10,
                   nnz_cap: int | None = None, want_components: bool = False, want_flags: bool = False):
        """perform_svd_reduction on a device adjacency: (embedding (n, n_comp), sigma (n_comp,)) fp64
        device tensors.  Q0 is generated on the host exactly as sklearn does and cached per (n, r, seed).
        `nnz_cap` is a caller-supplied bound on the number of edges (no host sync); if the adjacency has more, the
        neighbour lists are truncated on the device (memory-safe) and flags[0] is raised: with `want_flags` a copy
        of the int32[4] flag word (taken on the same stream) is appended to the result for the caller to check once
        the stream has run -- `check_rsvd_flags`."""
        n = adj.n
        n_comp = min(int(reduced_dim), n - 1)
        if n_comp < 1:
            raise ValueError("need at least 2 columns")  # TruncatedSVD: ensure_min_features=2
        r = n_comp + n_oversamples
        if nnz_cap is None:
            nnz_cap = int(adj.degrees()[2][1].item())  # blocking; callers on the fast path pass a bound
        nnz_cap = max(int(nnz_cap), 1)
        h = self._rsvd_handle(n, r, nnz_cap)
        if self._q0_key != (n, r, seed):
            q0 = np.random.RandomState(seed).normal(size=(n, r))
            q0_dev = torch.from_numpy(q0).to(self.device)
            call("mused_rsvd_set_q0", h, ptr(q0_dev), n, r, stream_ptr())  # stream-ordered copy
            self._q0_key = (n, r, seed)
        mbuf = _lib.lib().mused_rsvd_mask_buffer(h)
        nbytes = n * adj.words * 8
        _hip_memcpy_d2d(mbuf, adj.mask.data_ptr(), nbytes)
        emb = torch.empty((n, n_comp), dtype=torch.float64, device=self.device)
        sig = torch.empty(n_comp, dtype=torch.float64, device=self.device)
        comp = torch.empty((n, n_comp), dtype=torch.float64, device=self.device) if want_components else None
        call("mused_rsvd_reduce", h, n, n_comp, r, n_iter, ptr(emb), ptr(sig), ptr(comp) if want_components else None,
             stream_ptr())
        flags = torch.empty(4, dtype=torch.int32, device=self.device)
        _hip_memcpy_d2d(flags.data_ptr(), _lib.lib().mused_rsvd_flags(h), 16)
        if self.rsvd_mode == "cholqr" and not self.defer and int(flags[2].item()) != 0:
            # (one small blocking read per window, behind the eigenstep.)  A Cholesky pivot of the final basis was weak:
            # the panel is numerically rank deficient (few non-empty rows) -- repeat on the reference's chain
            self.rsvd_fallbacks += 1
            hf = self._rsvd_fallback_handle(n, r, nnz_cap)
            q0_dev = torch.from_numpy(np.random.RandomState(seed).normal(size=(n, r))).to(self.device)
            call("mused_rsvd_set_q0", hf, ptr(q0_dev), n, r, stream_ptr())
            _hip_memcpy_d2d(_lib.lib().mused_rsvd_mask_buffer(hf), adj.mask.data_ptr(), nbytes)
            call("mused_rsvd_reduce", hf, n, n_comp, r, n_iter, ptr(emb), ptr(sig),
                 ptr(comp) if want_components else None, stream_ptr())
            _hip_memcpy_d2d(flags.data_ptr(), _lib.lib().mused_rsvd_flags(hf), 16)
        out = (emb, sig, comp) if want_components else (emb, sig)
        if want_flags:
            out = out + (flags,)
        return out

    @staticmethod
    def check_rsvd_flags(flags) -> None:
        """Raise if the flag word of an eigenstep (host copy, >= 4 ints) reports truncated neighbour lists or an
        eigensolve that gave up."""
        if int(flags[0]) != 0:
            raise MusedError(
                "randomized-SVD eigenstep: the fused adjacency has more edges than the nnz_cap it was given; "
                "its neighbour lists were truncated and the embedding is invalid"
            )
        if int(flags[3]) != 0:
            raise MusedError("randomized-SVD eigenstep: the r x r eigensolve gave up (work-queue timeout); the embedding "
                             "is invalid")

    def rsvd_status(self):
        flags, stats = (C.c_int * 1)(), (C.c_int * 4)()
        call("mused_rsvd_status", self._rsvd, flags, stats, stream_ptr())
        return int(flags[0]), [int(x) for x in stats]

    def close(self):
        if self._rsvd is not None:
            call("mused_rsvd_destroy", self._rsvd)
            self._rsvd = None
        if self._rsvd_fb is not None:
            call("mused_rsvd_destroy", self._rsvd_fb)
            self._rsvd_fb = None

    def __del__(self):
        if sys is None or sys.is_finalizing():  # interpreter exit: the HIP runtime may already be gone -- leave the
            return                               # handles to the process teardown instead of calling into it
        try:
            self.close()
        except Exception:
            pass


def _hip_memcpy_d2d(dst_ptr: int, src_ptr: int, nbytes: int):
    """Async device-to-device copy on torch's current stream between raw pointers."""
    call("mused_memcpy_d2d", C.c_void_p(dst_ptr), C.c_void_p(src_ptr), nbytes, stream_ptr())


_ENGINES: dict[int, WindowEngine] = {}


def default_engine(n: int) -> WindowEngine:
    """Process-wide engine able to hold windows of n rows (grown on demand)."""
    for cap, eng in _ENGINES.items():
        if cap >= n:
            return eng
    for eng in _ENGINES.values():
        eng.close()
    _ENGINES.clear()
    _ENGINES[n] = WindowEngine(n)
    return _ENGINES[n]
