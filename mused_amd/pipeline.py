"""Window loop of the streaming pipeline (the role of `process_streaming_data`, main.py:13-130),
built on the device engine.

Per full window (trigger rule of main.py:32):
    per-modality kNN adjacency (device bitmask)  ->  OR-fusion  ->
        approach "sSVDMC"/"sSVDMC_hung": randomized-SVD embedding            (main.py:79)
        approach "SWFDMC"             : SeqBasedSWFD over the rows of the fused matrix, R from the
                                        first window only, sketch transposed to (W, l)  (main.py:58-76)
    -> k-means with n_clusters = #distinct true labels in the window (main.py:41,97)
    -> Hungarian matching against the previous window, min_overlap = 3 (main.py:110)
    -> labels appended (main.py:118-119).

Device work of window t+1 is enqueued while the host runs k-means / matching of window t
(`async_labels=True`): the embedding is copied to pinned memory behind an event, a worker thread
waits on the event and runs the scikit-learn / SciPy consumers in window order.
"""
from __future__ import annotations

import os
import threading
import time
from collections import deque
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import matrix_operations as mo
from .engine import WindowEngine
from .swfd import SeqBasedSWFD


class StreamPipeline:
    def __init__(self, window_size, reduced_dim, k_basis, seed, approach="sSVDMC", modality_types=None,
                 step_window_ratio=1, engine=None, async_labels=True, feature_sketch=False, stream=None,
                 assume_finite=False, window_slots=1):
        if approach not in ("sSVDMC", "sSVDMC_hung", "SWFDMC"):
            raise ValueError(f"approach {approach!r} is not on the device hot path")
        self.W, self.ell, self.k, self.seed = int(window_size), int(reduced_dim), int(k_basis), int(seed)
        self.approach = approach
        self.types = modality_types
        # rows with a non-finite entry are dropped from the kNN (matrix_operations.py:114-115).  Finding out needs one
        # small host read per window (on a side stream, so it does not wait for the previous window's device work);
        # `assume_finite=True` skips it for callers that generate their rows (the benchmark's synthetic stream)
        self.assume_finite = bool(assume_finite)
        self._chk = None
        self.ratio = step_window_ratio
        self.eng = engine or WindowEngine(self.W)
        # WINDOW SLOTS (sSVDMC only): adjacency -> fusion -> eigenstep of one window is a chain of ~1,600 small dependent
        # launches that leaves most of the GPU idle (25 ms of latency, a few per cent of its throughput); windows are
        # independent until the label chain, so consecutive windows go to `window_slots` engines on their own streams
        # (each driven by its own host thread: launching the 1,600-node graph costs ~20 ms of host time) and overlap.
        # Measured at config 2 (Cholesky-QR eigenstep): 9.9 -> 6.6 ms per window with 4 slots (1.01 M -> 1.5 M rows/s);
        # 2 slots can be slower than one, the best count varies with how HIP maps the streams onto hardware queues.
        # Opt-in (default 1; MUSED_WINDOW_SLOTS for
        # process_streaming_data).  The sketch approaches carry state from window to window and keep one slot.
        self._nslots = max(1, int(window_slots)) if (approach != "SWFDMC" and not feature_sketch) else 1
        # Every engine records its eigenstep graph once, on the caller's thread, before the slot threads exist (see
        # process_window).  A modality without an edge bound ("username") re-creates an engine's handle -- frees, graph
        # destruction, allocations, a new capture -- whenever a window has more edges than the handle was sized for; the
        # library serialises those resource calls with every capture of the process (csrc/internal.h: capture_mutex;
        # round 2 ran them beside another slot's capture and aborted, gpurun_out/r2_gputest23.log).  Such streams still
        # keep one slot: a re-creation stalls every slot behind that lock for tens of milliseconds.
        if any(mo.edges_per_row(t, self.k) is None for t in (modality_types or [])):
            self._nslots = 1
        # DEFERRED FLAGS: the candidate-list overflow of the fused kNN and the weak-pivot / edge-count flags of the
        # eigenstep are not read on the enqueueing thread; they travel to the host behind the window's event and a
        # flagged (rare) window is repeated by the label worker on a fallback engine (classic kNN path, the reference's
        # LU / Householder chain).  Not for the sketch approaches: the adjacency feeds a sketch that cannot be rewound.
        self._defer = approach != "SWFDMC" and os.environ.get("MUSED_DEFER_FLAGS", "1") != "0"
        self._fb_eng = None
        self._fb_lock = threading.Lock()
        self.redone_windows = 0
        self._nnz_hint = 0         # edges of the densest window seen so far ("username": no a-priori bound)
        self._closed = False
        # reuse across hopping windows (one engine, every window in order).  Default: from ratio 4 on -- measured at config 2
        # (W = 10^4, d = 1024, tools/hop_reuse_time.py): similarity + selection per window 2.70 ms from scratch, with reuse
        # 2.90 ms at ratio 2 (the phases that establish the entering rows' thresholds run on too few tiles to fill the GPU),
        # 2.19 ms at ratio 4, 1.96 ms at ratio 8.  MUSED_HOP_REUSE=1 / 0 forces it on (any ratio > 1) / off.
        hr = os.environ.get("MUSED_HOP_REUSE")
        self._hop_reuse = self._nslots == 1 and int(step_window_ratio) > 1 and (
            hr == "1" or (hr != "0" and int(step_window_ratio) >= 4))
        self._slots = None        # [(engine, stream)], built at the first window
        self._nwin = 0
        self.swfd = None          # SWFDMC sketch over fused-adjacency rows (d = W)
        self.feature_sketch = feature_sketch
        self.fswfd = None         # optional SWFD over the raw feature rows (BASELINE config 2 wording)
        self.prev = None
        self.out = []
        self.trace = []
        self.latencies = []
        # host consumers: k-means of different windows is independent (a small pool), the Hungarian matching is a
        # chain over windows (one worker, jobs run in submission order)
        nk = max(1, int(os.environ.get("MUSED_LABEL_WORKERS", "3")))
        self._kpool = ThreadPoolExecutor(max_workers=nk) if async_labels else None
        self._pool = ThreadPoolExecutor(max_workers=1) if async_labels else None
        self._max_inflight = nk + 2
        self.host_ms = {"kmeans": [], "match": []}
        # scikit-learn's k-means takes one OpenMP thread per visible core; on a many-core host (256 on the MI355X
        # boxes) that is ~3x slower for a (10k, 128) problem than a handful of threads
        self._omp_limit = None
        self._blas_limit = None
        try:
            from threadpoolctl import threadpool_limits

            want = int(os.environ.get("MUSED_LABEL_THREADS", "8"))
            if want > 0 and (os.cpu_count() or 1) > want:
                self._omp_limit = threadpool_limits(limits=want, user_api="openmp")
            # k-means calls BLAS from inside its OpenMP loops and caps it at one thread around them -- by flipping
            # the process-wide BLAS pool size, which several k-means workers do concurrently and out of step
            # (OpenBLAS then warns "Detect OpenMP Loop and this application may hang"): pin it at one thread here
            self._blas_limit = threadpool_limits(limits=1, user_api="blas") if nk > 1 and async_labels else None
        except Exception:  # threadpoolctl missing: keep the library default
            self._omp_limit = None
        # k-means of the embedding: Lloyd iterations on the device (host k-means++ seeding) unless
        # MUSED_KMEANS=host; every label worker drives its own high-priority stream
        self._km_device = os.environ.get("MUSED_KMEANS", "device") != "host"
        self._km_local = threading.local()
        self.km_device_windows = 0
        self._pending = deque()
        # the feature-row sketch is independent of the adjacency / eigenstep of the same window: it
        # runs on its own HIP stream and the two meet again before the results are handed to the host
        # (only when this pipeline has the GPU to itself: HIP maps streams onto a handful of hardware
        # queues -- 4 by default -- and streams sharing a queue run in order, so concurrent pipelines
        # use ONE stream each and overlap with each other instead)
        self._side = torch.cuda.Stream(priority=-1) if (feature_sketch and stream is None) else None
        # all device work of this pipeline is enqueued on `stream` (default: the current stream), so
        # several pipelines -- each owning a contiguous block of windows -- can share one GPU
        self._stream = stream
        # pinned staging buffers are recycled: allocating pinned memory synchronises the device, which
        # would serialise concurrent pipelines
        self._pins = []
        self._flag_pins = []
        self._pin_lock = threading.Lock()
        self._dpools = None
        self._device = torch.cuda.current_device()  # worker threads must select it themselves

    # ---- device side of one window --------------------------------------------------------------
    def window_device(self, mods, eng=None, defer=False, lo=None):
        """mods: list of (W, d_m) float32/float64 tensors on the device.  Returns (reduced (W, m) CUDA
        tensor, sigma CUDA tensor, flags): flags = None (sketch approaches), the eigenstep's int32[4] flag word, or with
        `defer` the int32[12] word of `WindowEngine.window_flags` (nothing read on this thread)."""
        eng = eng or self.eng
        eng.begin_window(defer)
        try:
            return self._window_device(mods, eng, defer, lo)
        finally:
            eng.defer = False  # direct callers of the engine get finished results again

    def _window_device(self, mods, eng, defer, lo=None):
        types = self.types or [""] * len(mods)
        # hopping windows (step_window_ratio > 1): consecutive windows share rows -> the dense modalities reuse their
        # candidate lists across windows (engine.knn_adjacency_hop).  `lo` = stream row of the window's row 0.
        reuse = lo is not None and self._hop_reuse and eng is self.eng
        adjs = [self._adjacency(m, t, eng, hop=((i, lo) if reuse else None)) for i, (m, t) in enumerate(zip(mods, types))]
        fused = eng.fuse(adjs) if len(adjs) > 1 else adjs[0]
        if len(adjs) == 1:
            fused.fused = False
        if self.feature_sketch:
            X = mods[0] if len(mods) == 1 else torch.cat(mods, dim=1)
            if self.fswfd is None:
                R = float((X.double() ** 2).sum(dim=1).max().item())
                self.fswfd = SeqBasedSWFD(N=self.W, R=R, d=X.shape[1], sketch_dim=self.ell)
            if self._side is not None:
                main = torch.cuda.current_stream()
                self._side.wait_stream(main)
                with torch.cuda.stream(self._side):
                    X.record_stream(self._side)
                    self.fswfd.fit(X)
                    self.feature_B, self.feature_sigma, _ = self.fswfd.get_device()
            else:
                self.fswfd.fit(X)
                self.feature_B, self.feature_sigma, _ = self.fswfd.get_device()
        if self.approach == "SWFDMC":
            if self.swfd is None:  # main.py:60-62
                R = self.eng.max_row_sq_norm(fused)
                self.swfd = SeqBasedSWFD(N=self.W, R=R, d=fused.n, sketch_dim=self.ell)
            self.swfd.fit_adjacency(fused)  # rows of the fused 0/1 matrix straight from the bitmask (no W x W dense)
            B, sigma, _ = self.swfd.get_device()
            reduced = B.t().contiguous() if B.shape[0] != self.W else B  # main.py:73-76
            return reduced, sigma, None
        # edges per row: k selected (l2; the row itself is normally one of them) or k + 1 (cosine / text)
        per_row = [mo.edges_per_row(t, self.k) for t in types]
        if all(b is not None for b in per_row):
            nnz_cap = fused.n * sum(per_row)
        elif defer:
            # "username": as many edges as a user has rows.  Optimistic bound (the densest window so far, 8 per row to
            # begin with); a window with more raises flags[0] and is repeated with its exact count by the label worker
            nnz_cap = max(self._nnz_hint, fused.n * (8 + sum(b or 0 for b in per_row)))
        else:  # count them (blocking)
            nnz_cap = max(int(fused.degrees()[2][1].item()), 1)
            self._nnz_hint = max(self._nnz_hint, nnz_cap + nnz_cap // 4)
        emb, sigma, flags = eng.svd_reduce(fused, self.ell, self.seed, nnz_cap=nnz_cap, want_flags=True)
        if defer:
            flags = eng.window_flags(flags)
        return emb, sigma, flags

    def _adjacency(self, m, t, eng=None, hop=None):
        """One modality of one window -> device adjacency, with the reference's row filtering."""
        eng = eng or self.eng
        if t == "text" or t in mo._METADATA_TYPES or not isinstance(m, torch.Tensor):
            # (text: thousands of documents tie at similarity 0 and overflow the candidate lists in most windows -- the
            # overflow word is read at once there instead of repeating the whole window, TF-IDF included, afterwards)
            was, eng.defer = eng.defer, eng.defer and t != "text"
            try:
                return mo.adjacency_on_device(m, t, self.k, engine=eng)
            finally:
                eng.defer = was
        metric = mo._metric_for(t)
        if not self.assume_finite and m.is_floating_point():
            if self._chk is None:
                with self._pin_lock:  # slot threads may get here together
                    if self._chk is None:
                        self._chk = torch.cuda.Stream()
            # The check runs on a side stream so that it does not wait for the previous window's device work -- but it
            # must see the rows: the side stream waits for an event recorded on the stream this window is enqueued on
            # (which already waits for whatever produced the rows on the caller's stream, _device_side), nothing more
            ev = torch.cuda.Event()
            ev.record()
            self._chk.wait_event(ev)
            m.record_stream(self._chk)
            with torch.cuda.stream(self._chk):
                ok = bool(torch.isfinite(m).all().item())
            if not ok:
                return mo.adjacency_on_device(m, t, self.k, engine=eng)
        if hop is not None:
            return eng.knn_adjacency_hop(m, self.k, metric, key=hop[0], lo=hop[1])
        return eng.knn_adjacency(m, self.k, metric)

    # ---- host consumers -------------------------------------------------------------------------
    def _cluster(self, job):
        """Independent per window: wait for the embedding, k-means (main.py:97)."""
        ev, red_pin, sig_pin, n_clusters, trigger, t_start = job[:6]
        flag_pin, reduced_dev, mods = job[6], job[7], job[8]
        slot_keys, job_eng = (job[9], job[10]) if len(job) > 10 else ([], self.eng)
        torch.cuda.set_device(self._device)
        ev.synchronize()
        reduced_host, sigma_host = red_pin.numpy().copy(), sig_pin.numpy().copy()
        flags_host = flag_pin.numpy().copy() if flag_pin is not None else None
        with self._pin_lock:
            self._pins.append((red_pin, sig_pin))
            if flag_pin is not None:
                self._flag_pins.append(flag_pin)
        if flags_host is not None:
            redo = len(flags_host) > 4 and (flags_host[2] != 0 or flags_host[4:].any()
                                            or (flags_host[0] != 0 and self._optimistic_nnz()))
            if redo:
                # a candidate list overflowed / the final Cholesky-QR met a weak pivot / more edges than the optimistic
                # bound: this window again, on the paths that have no such limits (rare; blocking on this worker only)
                for i_, f_ in enumerate(flags_host[4:]):   # hopping-window reuse: rebuild that modality's state next window
                    key = slot_keys[i_] if i_ < len(slot_keys) else None   # (flag word i_ <-> the i_-th DEFERRED kNN call)
                    if f_ and key is not None:
                        job_eng.request_hop_reset(key)
                reduced_dev, sigma_dev, flags_host = self._redo_window(mods)
                reduced_host, sigma_host = reduced_dev.cpu().numpy(), sigma_dev.cpu().numpy()
            WindowEngine.check_rsvd_flags(flags_host)  # raised on the label worker, surfaces in flush()
        t0 = time.perf_counter()
        if self._km_device and reduced_dev is not None and reduced_dev.dtype == torch.float64:
            st = getattr(self._km_local, "stream", None)
            if st is None:
                st = self._km_local.stream = torch.cuda.Stream(priority=-1)
            reduced_dev.record_stream(st)  # produced on the pipeline's stream, complete (ev), consumed on the worker's
            clusters = mo.perform_clustering_on_device(reduced_dev, n_clusters, self.seed, emb_host=reduced_host, stream=st)
            self.km_device_windows += 1
        else:
            clusters = mo.perform_clustering(reduced_host, n_clusters, self.seed)
        self.host_ms["kmeans"].append(1e3 * (time.perf_counter() - t0))
        return clusters, sigma_host

    def _optimistic_nnz(self):
        return any(mo.edges_per_row(t, self.k) is None for t in (self.types or []))

    def _redo_window(self, mods):
        """One window on the fallback engine (score-matrix kNN, LU / Householder eigenstep, exact edge count), on the
        calling worker's stream; returns (reduced, sigma, host flags)."""
        with self._fb_lock:
            if self._fb_eng is None:
                self._fb_eng = WindowEngine(self.W)
                self._fb_eng.knn_mode = "classic"
                self._fb_eng.rsvd_mode = "lu"
            st = getattr(self._km_local, "stream", None)
            if st is None:
                st = self._km_local.stream = torch.cuda.Stream(priority=-1)
            with torch.cuda.stream(st):
                for m in mods:
                    if isinstance(m, torch.Tensor):
                        m.record_stream(st)
                reduced, sigma, flags = self.window_device(mods, self._fb_eng, defer=False)
                flags_host = flags.cpu().numpy()
                st.synchronize()
            self.redone_windows += 1
            if self.redone_windows == 8 or (self.redone_windows > 8 and self.redone_windows % 64 == 0):
                import warnings

                warnings.warn(f"mused_amd: {self.redone_windows} windows of this stream were repeated on the fallback engine "
                              "(candidate-list overflow / weak pivot / edge bound): correct, but several times slower", RuntimeWarning)
            return reduced, sigma, flags_host

    def _chain(self, fut, job):
        """Sequential over windows: Hungarian matching against the previous window (main.py:105-119)."""
        clusters, sigma_host = fut.result() if hasattr(fut, "result") else fut
        trigger, t_start = job[4], job[5]
        t0 = time.perf_counter()
        matched = mo.match_clusters(self.prev, clusters, method="hungarian", min_overlap=3)
        if matched is None or len(matched) == 0:  # main.py:114-116
            matched = np.full(self.W, 0)
        self.host_ms["match"].append(1e3 * (time.perf_counter() - t0))
        self.prev = matched
        self.out.extend(matched)
        self.trace.append(dict(trigger=trigger, sigma=sigma_host, raw=np.asarray(clusters), matched=np.asarray(matched)))
        self.latencies.append(time.perf_counter() - t_start)

    def _get_pins(self, reduced, sigma):
        while True:
            with self._pin_lock:
                for i, (rp, sp) in enumerate(self._pins):
                    if rp.shape == reduced.shape and sp.shape == sigma.shape:
                        return self._pins.pop(i)
            if self._nslots == 1 and len(self._pending) >= self._max_inflight:  # back-pressure: reuse a buffer
                self._pending.popleft().result()
                continue
            return (torch.empty(reduced.shape, dtype=reduced.dtype, pin_memory=True),
                    torch.empty(sigma.shape, dtype=sigma.dtype, pin_memory=True))

    def _device_side(self, mods, n_clusters, trigger, t_start, eng, st, wait_ev, lo=None):
        """Enqueue one window on (eng, st); returns the job the label workers consume."""
        torch.cuda.set_device(self._device)
        if wait_ev is not None:
            st.wait_event(wait_ev)  # whatever produced the rows on the caller's stream
        with torch.cuda.stream(st):
            if self._nslots > 1:
                for m in mods:
                    if isinstance(m, torch.Tensor):
                        m.record_stream(st)
            reduced, sigma, flags = self.window_device(mods, eng, defer=self._defer, lo=lo)
            if self._side is not None:
                torch.cuda.current_stream().wait_stream(self._side)  # window latency includes the sketch
            red_pin, sig_pin = self._get_pins(reduced, sigma)
            red_pin.copy_(reduced, non_blocking=True)
            sig_pin.copy_(sigma, non_blocking=True)
            flag_pin = None
            if flags is not None:
                with self._pin_lock:
                    flag_pin = self._flag_pins.pop() if self._flag_pins else None
                if flag_pin is None or flag_pin.numel() != flags.numel():
                    flag_pin = torch.empty(flags.numel(), dtype=torch.int32, pin_memory=True)
                flag_pin.copy_(flags, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        # (slot -> hop-state key of this window's deferred flag words: the engine has moved on by the time a worker reads them)
        return (ev, red_pin, sig_pin, n_clusters, trigger, t_start, flag_pin, reduced, mods, list(eng.slot_keys), eng)

    def _cluster_after(self, fut_job):
        return self._cluster(fut_job.result())

    def _chain_after(self, fut_cluster, fut_job):
        return self._chain(fut_cluster, fut_job.result())

    def process_window(self, mods, true_labels_window, trigger=None, lo=None):
        """One full window.  `lo`: stream row of its first row -- given for hopping windows it lets the dense modalities
        reuse the previous window's similarity work (only with one engine and windows handed in in order)."""
        t_start = time.perf_counter()
        n_clusters = len(np.unique(true_labels_window))  # main.py:41
        caller = torch.cuda.current_stream()
        if self._slots is None:
            first = self._stream if self._stream is not None else caller
            if self._nslots == 1:
                self._slots = [(self.eng, self._stream)]  # stream None: whatever is current at each call
            else:
                # every slot on a stream of its own (none of them the caller's: a slot then waits for the caller's stream --
                # whatever produced the rows -- without waiting for another slot's window) and with a host thread of its
                # own: launching the ~1,600-node eigenstep graph costs ~20 ms of host time per window
                self._slots = [(self.eng if i == 0 else WindowEngine(self.W), torch.cuda.Stream(priority=first.priority))
                               for i in range(self._nslots)]
                self._dpools = [ThreadPoolExecutor(max_workers=1) for _ in range(self._nslots)]
                # Every engine records its eigenstep graph at its first window.  Do that here, one engine after the other
                # and before any worker thread exists.  What a ThreadLocal capture does not survive on this runtime is a
                # DEVICE-WIDE synchronisation made by another host thread while it records (tools/repro_capture_threads.hip:
                # hipDeviceSynchronize beside a capture fails 38 of 40 captures with "operation failed due to a previous
                # error during capture"; allocations, frees, launches, graph / stream creation and destruction, stream and
                # event synchronisation beside it: 0 of 40) -- the library therefore serialises its captures with its own
                # handle creation / destruction (capture_mutex) and nothing here calls torch.cuda.synchronize() off the
                # caller's thread.
                for e_, s_ in self._slots:
                    s_.wait_stream(caller)
                    with torch.cuda.stream(s_):
                        self.window_device(mods, e_, defer=self._defer)
                    s_.synchronize()
        slot = self._nwin % self._nslots
        eng, st = self._slots[slot]
        st = st if st is not None else caller
        self._nwin += 1
        if self._nslots == 1 or self._pool is None:
            if self._nslots > 1:
                st.wait_stream(caller)
            job = self._device_side(mods, n_clusters, trigger, t_start, eng, st, None, lo)
            if self._pool is None:
                self._chain(self._cluster(job), job)
            else:
                self._pending.append(self._pool.submit(self._chain, self._kpool.submit(self._cluster, job), job))
            return
        while len(self._pending) >= self._max_inflight + self._nslots:  # back-pressure (pinned buffers, engines)
            self._pending.popleft().result()
        wait_ev = torch.cuda.Event()
        wait_ev.record(caller)
        fut_job = self._dpools[slot].submit(self._device_side, mods, n_clusters, trigger, t_start, eng, st, wait_ev)
        fut_cluster = self._kpool.submit(self._cluster_after, fut_job)
        self._pending.append(self._pool.submit(self._chain_after, fut_cluster, fut_job))

    def process_reduced(self, reduced: torch.Tensor, sigma: torch.Tensor, true_labels_window, trigger=None):
        """Hand the label workers a window whose reduced matrix (W, m) is already on the device (current stream): k-means
        (main.py:97) and the trace entry {trigger, sigma, raw, matched}.  For drivers that produce the embedding themselves
        (`SwfdmcLanes`); the matching of `out` follows the order of the calls, `trace[i]["raw"]` is order independent."""
        t_start = time.perf_counter()
        n_clusters = len(np.unique(true_labels_window))
        red_pin, sig_pin = self._get_pins(reduced, sigma)
        red_pin.copy_(reduced, non_blocking=True)
        sig_pin.copy_(sigma, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        job = (ev, red_pin, sig_pin, n_clusters, trigger, t_start, None, reduced, None, [], self.eng)
        if self._pool is None:
            self._chain(self._cluster(job), job)
        else:
            self._pending.append(self._pool.submit(self._chain, self._kpool.submit(self._cluster, job), job))

    def flush(self):
        """Wait for every window handed in so far; re-raises the first error a worker met (the remaining windows are
        still drained, so that nothing is in flight afterwards)."""
        first = None
        while self._pending:
            try:
                self._pending.popleft().result()
            except BaseException as e:  # noqa: BLE001 -- re-raised below
                first = first or e
        if first is None:
            for s in (self.swfd, self.fswfd):
                if s is not None:
                    s.check()  # an eigensolve of the sketch that gave up invalidates its results
        if first is not None:
            raise first

    def run(self, data_modalities, true_labels):
        """Stream whole modalities (host or device arrays) through the window loop; returns the
        concatenated event labels (`all_clusters`, main.py:125)."""
        dev = []
        for m in data_modalities:
            if isinstance(m, torch.Tensor):
                dev.append(m.cuda())
                continue
            a = np.asarray(m)
            # numeric modalities live on the device; string records ("text") stay on the host for the TF-IDF step
            dev.append(torch.from_numpy(np.ascontiguousarray(a)).cuda() if a.dtype.kind in "fiub" else a)
        torch.cuda.current_stream().synchronize()  # rows resident before the first window (see _adjacency)
        n = dev[0].shape[0]
        for i in range(n):
            if i + 1 >= self.W and (i + 1) * self.ratio % self.W == 0:  # main.py:32
                lo = i + 1 - self.W
                self.process_window([m[lo : i + 1] for m in dev], true_labels[lo : i + 1], trigger=i,
                                    lo=lo if self.ratio > 1 else None)
        self.flush()
        return np.array(self.out)

    def close(self):
        """Deterministic teardown, also after a failed flush(): workers are drained and joined first, then the device is
        idle, then every handle this pipeline created is destroyed -- nothing is left to `__del__` / interpreter exit."""
        if self._closed:
            return
        self._closed = True
        err = None
        try:
            self.flush()
        except BaseException as e:  # noqa: BLE001 -- re-raised after the teardown
            err = e
        for dp in (self._dpools or []):
            dp.shutdown(wait=True)
        self._dpools = None
        if self._pool is not None:
            self._kpool.shutdown(wait=True)
            self._pool.shutdown(wait=True)
        try:  # the streams this pipeline enqueued on (no device-wide synchronisation: another pipeline may be capturing)
            for _, st in (self._slots or []):
                if st is not None:
                    st.synchronize()
            for st in (self._stream, self._side, self._chk):
                if st is not None:
                    st.synchronize()
            torch.cuda.current_stream().synchronize()
        except Exception as e:  # noqa: BLE001
            err = err or e
        for s in (self.swfd, self.fswfd):
            if s is not None:
                s.close()
        self.swfd = self.fswfd = None
        for e, _ in (self._slots or [])[1:]:
            e.close()
        self._slots = None
        if self._fb_eng is not None:
            self._fb_eng.close()
            self._fb_eng = None
        for lim in (self._blas_limit, self._omp_limit):
            if lim is not None:
                lim.restore_original_limits()
        self._blas_limit = self._omp_limit = None
        if err is not None:
            raise err

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.close()
        else:  # the body failed: tear down, keep the body's exception
            try:
                self.close()
            except BaseException:  # noqa: BLE001
                pass
        return False


def process_streaming_data(results, data_modalities, modality_types, window_size, reduced_dim, k_basis, n_clusters_total,
                           seed, approach, complete_true_labels, step_window_ratio, noise_rate, label_mode, sorting,
                           eps, min_samples):
    """Same positional parameters as main.py:13.  The reference hands its outputs to
    metrics_evaluation (out of scope here); this returns `results` with the label arrays and the
    wall time instead."""
    t0 = time.time_ns()
    # modality types go through unchanged: "" / anything the reference does not special-case = Euclidean kNN
    # (matrix_operations.py:112), "text" and "cosine" = the cosine kernel, the other SED2012 metadata types raise
    with StreamPipeline(window_size, reduced_dim, k_basis, seed, approach, list(modality_types), step_window_ratio,
                        window_slots=int(os.environ.get("MUSED_WINDOW_SLOTS", "1"))) as pipe:
        clusters = pipe.run(data_modalities, np.asarray(complete_true_labels))
    results = dict(results or {})
    results["all_clusters"] = clusters
    results["processing_time"] = (time.time_ns() - t0) / 1e9
    return results


class SwfdmcLanes:
    """The reference's SWFDMC approach (main.py:58-76: one SeqBasedSWFD over the rows of the fused W x W adjacency of every
    window, d = W; get() transposed to (W, l) -> k-means -> matching) with the windows of a stream dealt to `lanes`
    CONTIGUOUS blocks whose sketches advance in lock-step inside the same launches.  A block is preceded by ONE halo window
    it does not own (MAIN(t) continues AUX(t - 1), AUX starts empty: the sequential sketch is reproduced exactly); the block
    that starts at the beginning of the stream gets a window of empty rows instead.  R (main.py:61) is fixed by stream window
    0.  Raw k-means labels are collected per window and the Hungarian chain (main.py:105-119) is replayed in stream order.

        lanes = SwfdmcLanes(W, l, k, seed, n_lanes, R)      # R = max out-degree of window 0's fused adjacency
        out = lanes.run(windows, labels)                     # windows[t] = list of modality tensors of stream window t
    """

    def __init__(self, W, ell, k, seed, lanes, R, modality_types=None, stream=None, assume_finite=False):
        from .swfd import SeqBasedSWFD

        self.W, self.ell, self.k, self.seed, self.lanes = int(W), int(ell), int(k), int(seed), int(lanes)
        self.types = list(modality_types) if modality_types else None
        self.pipe = StreamPipeline(W, ell, k, seed, "sSVDMC", modality_types=self.types, async_labels=True, stream=stream,
                                   assume_finite=assume_finite)
        self.sk = SeqBasedSWFD(N=self.W, R=float(R), d=self.W, sketch_dim=self.ell, lanes=self.lanes)
        self.words = (self.W + 63) // 64
        self._masks = torch.zeros((self.lanes, self.W, self.words), dtype=torch.int64, device="cuda")

    @staticmethod
    def r_of_first_window(mods, W, k, modality_types=None, seed=0):
        """main.py:61 on the fused adjacency of stream window 0."""
        with StreamPipeline(W, 2, k, seed, "sSVDMC", modality_types=modality_types, async_labels=False) as p:
            types = p.types or [""] * len(mods)
            adjs = [p._adjacency(_to_dev(m), t) for m, t in zip(mods, types)]
            fused = p.eng.fuse(adjs) if len(adjs) > 1 else adjs[0]
            return p.eng.max_row_sq_norm(fused)

    def step(self, mods_per_lane, labels_per_lane, triggers, want):
        """One lock-step: lane p appends the fused adjacency of mods_per_lane[p] (None: a window of empty rows); lanes with
        want[p] get their window's sketch, k-means and a trace entry under triggers[p]."""
        eng, types = self.pipe.eng, None
        for p, mods in enumerate(mods_per_lane):
            if mods is None:
                self._masks[p].zero_()
                continue
            types = self.types or [""] * len(mods)
            eng.begin_window(False)
            adjs = [self.pipe._adjacency(_to_dev(m), t) for m, t in zip(mods, types)]
            fused = eng.fuse(adjs) if len(adjs) > 1 else adjs[0]
            self._masks[p].copy_(fused.mask)
        self.sk.fit_adjacency_lanes(self._masks)
        if not any(want):
            return
        B, sigma, _ = self.sk.get_device()                     # (lanes, l, W)
        for p in range(self.lanes):
            if want[p]:
                reduced = B[p].t().contiguous()                # main.py:73-76: (W, l)
                self.pipe.process_reduced(reduced, sigma[p], labels_per_lane[p], trigger=triggers[p])

    def run(self, windows, labels):
        """windows: list over stream windows of lists of modality arrays / tensors (W rows each); labels: true labels per
        window.  Returns all_clusters in stream order (the concatenated matched labels)."""
        from . import distributed as mdist

        K = len(windows)
        if K < self.lanes:
            raise ValueError(f"{self.lanes} lanes need at least as many windows (got {K})")
        for row in mdist.lane_schedule(K, self.lanes):
            mods = [windows[i] if i >= 0 else None for i, _ in row]
            labs = [labels[i] if i >= 0 else None for i, _ in row]
            self.step(mods, labs, [i for i, _ in row], [own for _, own in row])
        self.pipe.flush()
        self.sk.check()
        raw = {tr["trigger"]: tr["raw"] for tr in self.pipe.trace}
        self.sigma = {tr["trigger"]: tr["sigma"] for tr in self.pipe.trace}
        return mdist.replay_label_chain(np.array([raw[t] for t in range(K)], dtype=np.int64), mo.match_clusters)

    def close(self):
        self.pipe.close()
        self.sk.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def _to_dev(m):
    return m if isinstance(m, torch.Tensor) or not isinstance(m, np.ndarray) or m.dtype.kind not in "fiu" else torch.from_numpy(m).cuda()
