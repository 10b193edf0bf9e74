// a1/a2 of SURVEY section 8 WITHOUT the W x W score matrix (section 8 f1): the k smallest pairwise scores per row,
// selected while the scores come out of the fp64 MFMA GEMM.
//
// The reference materialises the full distance / similarity matrix (sklearn's ArgKmin works on 256-row chunks of it;
// cosine_similarity + argsort on all of it, matrix_operations.py:106-108,118-119).  Here a score leaves the
// accumulators of the tile that produced it only if it can still be among the k smallest of its row:
//
//   * the tile grid is enumerated by CYCLIC TILE DISTANCE delta = (J - I) mod T: tile (I, I + delta) serves rows of I
//     against columns of J and, mirrored, rows of J against columns of I -- the symmetric half of the work, every
//     unordered pair of row tiles once (delta = 0 .. T / 2), and after the tiles of delta <= a every row has seen
//     exactly (2 a + 1) * 128 columns, whatever its position;
//   * the distances are visited in a few PHASES of growing delta ([0,1], [2,4], [5,13], [14,40], ...: each phase
//     triples the columns seen).  Per row a threshold tau_i is kept: an upper bound of the k-th smallest score of the
//     row, +inf at first.  The epilogue appends (score, column) to the row's candidate list (one atomic counter per
//     row) iff score <= tau_i -- for both orientations of the tile;
//   * between two phases one wave per row selects the k smallest candidates by (score, column) -- exact radix select
//     on order-preserving 64-bit keys with wave ballots, ties towards the smaller column as in select_k_kernel --,
//     compacts the list to them and lowers tau_i to the k-th score: after a phase that has shown the row m columns,
//     the next phase admits about k / m of what it computes.  The last pass also emits the neighbour list
//     (ascending columns) and the adjacency bitmask row (own column cleared, matrix_operations.py:128).
//
// Scores are bit-identical to the classic path (mused_pairwise_scores): same MFMA tile loop, each orientation
// evaluated by the same epilogue expression.  At W = 10^4, k = 50 a row collects ~700 candidates in all (384 of them
// in the first phase, where nothing is known yet): ~0.1 GB of candidate traffic instead of the 1.6 GB of writing and
// re-reading the score matrix, and a 61 MB workspace instead of 800 MB.  Lists are bounded by `cap` per row; if
// any row overflows (pathological inputs: thousands of exactly equal scores) a device flag is raised and the caller
// falls back to the classic path -- never a wrong result.
#include <vector>

#include "gemm_f64.h"
#include "internal.h"

extern "C" int mused_row_sq_norms(const void* X, int dtype, long n, int d, long ld, double* out, void* stream);

namespace mused {

// LDS traffic of ONE wave is processed in order; this only stops the compiler from moving accesses across it
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

struct CandArgs {
  double* tau;              // [n] admission threshold per row: the k-th smallest score seen so far (+inf at first) ...
  int* taucol;              // [n] ... and the column of that k-th candidate: (score, column) is admitted iff it is
                            //     lexicographically below (tau, taucol) -- rows of equal scores (a zero row under
                            //     cosine, duplicated points) then admit only the columns that can still displace it
  int* count;      // [n] candidates appended so far (may exceed cap: overflow)
  double* cscore;  // [n][cap]
  int* ccol;       // [n][cap]
  int cap;
  int* overflow;   // device flag (bit 0: a list overflowed; bit 1, hop mode: a row's kept list no longer proves its top k)
  const double* nrm;  // squared norms (l2) / inverse norms (cosine), as in knn.hip
  // Hopping windows (mused_knn_fused_hop): the per-row state lives in RING SLOTS keyed by stream position
  // (slot of window row r = (r + slot_base) mod n), list entries carry ABSOLUTE column ids (window column + col_base), and
  // pairs of rows that were both in the previous window (r, c < n_ret) are not recomputed.  Tumbling windows: 0, 0, 0.
  int slot_base, col_base, n_ret, n;
};

__device__ __forceinline__ int cand_slot(const CandArgs& c, int row) {
  const int s = row + c.slot_base;
  return s >= c.n ? s - c.n : s;
}

template <int METRIC>
__device__ __forceinline__ double score_of(const double* __restrict__ nrm, int row, int col, double v) {
  if (METRIC == 0) {  // EpiSqL2::value
    const double dd = nrm[row] - 2.0 * v + nrm[col];
    return dd > 0.0 ? dd : 0.0;
  }
  return 0.0 - (v * nrm[row] * nrm[col]);  // EpiNegCos::value (never -0.0)
}

// slot: ring slot of the row; col: window column (stored as an absolute id)
__device__ __forceinline__ void cand_push(const CandArgs& c, int slot, int col, double v) {
  const int pos = atomicAdd(&c.count[slot], 1);
  if (pos < c.cap) {
    c.cscore[(long)slot * c.cap + pos] = v;
    c.ccol[(long)slot * c.cap + pos] = col + c.col_base;
  } else {
    atomicOr(c.overflow, 1);
  }
}

// tiles (I, (I + delta) mod T), delta in [d_lo, d_lo + n_delta): grid = ceil(T / 8) * 8 * n_delta workgroups, ordered
// so that 64 consecutive ones (one XCD's share) touch 8 A row-panels and 15 B row-panels.
// DIRECT (first phase, d_lo = 0, nothing known about any row yet): EVERY score is kept, so a row's list has a fixed
// layout -- slot (delta_signed + a) * 128 + (column mod 128) for the 2 a + 1 column tiles within cyclic distance
// a = n_delta - 1 -- and the scores are stored without atomics (columns beyond n as +inf); the host presets the
// counts to (2 a + 1) * 128.  Requires tiles >= 2 a + 2 (no column tile reached from both sides).
template <typename T, bool VEC, int METRIC, bool DIRECT>
__global__ __launch_bounds__(GEMM_THREADS, 2) void knn_band_kernel(GemmArgs g, CandArgs c, int tiles, int d_lo,
                                                                  int n_delta) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int e = xcd_remap(blockIdx.x, gridDim.x);
  const int per_group = 8 * n_delta;
  const int ig = e / per_group, rem = e - ig * per_group;
  const int delta = d_lo + rem / 8;
  // hop mode: at distance delta the tiles that involve an entering row are the contiguous range I >= t_new - delta (t_new =
  // first tile that holds one): the grid enumerates only those (from the smallest start of the phase: a few idle
  // workgroups per distance instead of 1 - (1 - 1 / ratio)^2 of the grid exiting at once, bunched on some XCDs)
  const int t_new = c.n_ret / GEMM_BM;
  const int ibase = t_new > delta ? t_new - delta : 0;
  const int I = ibase + ig * 8 + (rem & 7);
  if (I >= tiles || 2 * delta > tiles) return;
  if (2 * delta == tiles && I >= tiles / 2) return;  // even tile count: the antipodal pairs once
  int J = I + delta;
  if (J >= tiles) J -= tiles;
  if ((I + 1) * GEMM_BM <= c.n_ret && (J + 1) * GEMM_BN <= c.n_ret) return;  // hop mode: both row tiles were in the last window
  const int m0 = I * GEMM_BM, n0 = J * GEMM_BN;
  const T* X = reinterpret_cast<const T*>(g.A);
  v4f64 acc[4][4];
  gemm_tile_mainloop<T, T, true, true, VEC>(g, X, X, m0, n0, 0, g.K, smem, acc);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1, kq = lane >> 4, li = lane & 15;
  const int n = g.M;
  const bool both = (delta != 0);  // a diagonal tile holds both orientations of its pairs itself
  if constexpr (DIRECT) {
    const int a = n_delta - 1;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    const bool hopm = c.n_ret > 0;  // hop mode: fixed slots for the ENTERING rows only, threshold pushes for the staying ones
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = wr * 64 + i * 16 + kq + 4 * r, row = m0 + rl;
        const bool row_new = row >= c.n_ret;
        const int srow = row < n ? cand_slot(c, row) : 0;
        double trow = 0.0;
        int crow = 0;
        if (hopm && !row_new && row < n) { trow = c.tau[srow]; crow = c.taucol[srow]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int cl = wc * 64 + j * 16 + li, col = n0 + cl;
          const double av = acc[i][j][r];
          const bool col_new = col >= c.n_ret;
          if (row < n) {  // row side: columns of tile J at signed distance +delta
            if (row_new) {
              const long p = (long)srow * c.cap + (a + delta) * 128 + cl;
              const bool ok = col < n;
              c.cscore[p] = ok ? score_of<METRIC>(c.nrm, row, col, av) : inf;
              c.ccol[p] = ok ? col + c.col_base : 0x7fffffff;
            } else if (col_new && col < n) {
              const double v = score_of<METRIC>(c.nrm, row, col, av);
              if (v < trow || (v == trow && col + c.col_base < crow)) cand_push(c, srow, col, v);
            }
          }
          if (both && col < n) {  // mirrored: row `col` of tile J sees column `row` of tile I at signed distance -delta
            if (col_new) {
              const long p = (long)cand_slot(c, col) * c.cap + (a - delta) * 128 + rl;
              const bool ok = row < n;
              c.cscore[p] = ok ? score_of<METRIC>(c.nrm, col, row, av) : inf;
              c.ccol[p] = ok ? row + c.col_base : 0x7fffffff;
            } else if (row_new && row < n) {
              const int sc_ = cand_slot(c, col);
              const double tc = c.tau[sc_];
              const double v2 = score_of<METRIC>(c.nrm, col, row, av);
              if (v2 < tc || (v2 == tc && row + c.col_base < c.taucol[sc_])) cand_push(c, sc_, row, v2);
            }
          }
        }
      }
    }
    return;
  }
  // scores are never -0.0 (l2: clamped to +0.0, cosine: 0.0 - x) nor NaN: comparing doubles orders them like their keys
  // (thresholds compare ABSOLUTE column ids: taucol is one)
  double tcol[4];
  int tccol[4], scol[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = n0 + wc * 64 + j * 16 + li;
    const bool ok = both && col < n;
    scol[j] = ok ? cand_slot(c, col) : 0;
    tcol[j] = ok ? c.tau[scol[j]] : __longlong_as_double(0xfff0000000000000ll);  // -inf admits nothing
    tccol[j] = ok ? c.taucol[scol[j]] : -1;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + wr * 64 + i * 16 + kq + 4 * r;
      if (row >= n) continue;
      const int srow = cand_slot(c, row);
      const double trow = c.tau[srow];
      const int crow = c.taucol[srow];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = n0 + wc * 64 + j * 16 + li;
        if (col >= n) continue;
        if (row < c.n_ret && col < c.n_ret) continue;  // hop mode: this pair is in both rows' kept lists (or was never admitted)
        const double a = acc[i][j][r];
        const double v = score_of<METRIC>(c.nrm, row, col, a);
        if (v < trow || (v == trow && col + c.col_base < crow)) cand_push(c, srow, col, v);
        if (both) {
          const double v2 = score_of<METRIC>(c.nrm, col, row, a);  // evaluated in ITS orientation, as the classic path does
          if (v2 < tcol[j] || (v2 == tcol[j] && row + c.col_base < tccol[j])) cand_push(c, scol[j], row, v2);
        }
      }
    }
  }
}

// ---- per-row selection among the candidates: one wave per row ------------------------------------------------
__device__ __forceinline__ double key_to_f64(unsigned long long k) {
  const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;  // inverse of f64_key
  return __longlong_as_double((long long)u);
}

// row_lo: a non-final pass touches rows >= row_lo only (hop mode: the rows that entered with this window; the others keep
// their list and threshold as they are).  check != 0 (final pass, hop mode): a row whose kept list no longer proves its
// k smallest -- fewer than k candidates, the k-th above the row's threshold, or a list close to its capacity -- raises
// bit 1 of the overflow word (the caller then recomputes the window from scratch).
template <int PL>  // candidates per lane: cap <= 64 * PL
__global__ __launch_bounds__(256) void cand_select_kernel(CandArgs c, int n, int k, int final, int* __restrict__ out_idx,
                                                         unsigned long long* __restrict__ out_mask, int mask_words,
                                                         int row_lo, int check) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sel_lds[];  // final: [4 waves][mask_words] bit rows + [4][k] ints
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wrow = blockIdx.x * 4 + wave;  // window row
  if (wrow >= n) return;
  if (!final && wrow < row_lo) return;
  const int row = cand_slot(c, wrow);      // its ring slot: where the list, the counter and the threshold live
  int cnt = c.count[row];
  if (final && check && lane == 0 && (cnt < (k < n ? k : n) || cnt > c.cap - 64)) atomicOr(c.overflow, 2);
  cnt = cnt < c.cap ? cnt : c.cap;
  if (!final && cnt <= k) return;  // fewer candidates than wanted: nothing to tighten yet (tau stays)
  const int kk = k < cnt ? k : cnt;
  const int ns = (cnt + 63) >> 6;  // slots per lane that hold anything (wave-uniform): the loops below stop there
  unsigned long long key[PL];
  int col[PL];
#pragma unroll
  for (int s = 0; s < PL; ++s) {
    const int p = s * 64 + lane;
    const bool ok = p < cnt;
    key[s] = ok ? f64_key(c.cscore[(long)row * c.cap + p]) : ~0ull;
    col[s] = ok ? c.ccol[(long)row * c.cap + p] : 0x7fffffff;
  }
  // bits above the highest bit in which the row's smallest and largest valid keys differ are common
  unsigned long long kmin = ~0ull, kmax = 0ull;
#pragma unroll
  for (int s = 0; s < PL; ++s) {
    const bool ok = (s * 64 + lane) < cnt;
    kmin = (ok && key[s] < kmin) ? key[s] : kmin;
    kmax = (ok && key[s] > kmax) ? key[s] : kmax;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long a = __shfl_xor(kmin, o), b = __shfl_xor(kmax, o);
    kmin = a < kmin ? a : kmin;
    kmax = b > kmax ? b : kmax;
  }
  unsigned long long prefix = kmin, known = ~0ull;
  int rem = kk;  // the rem-th smallest (1-based) among the keys that match `prefix` on the `known` bits
  if (kmin != kmax) {
    const int top = 64 - __clzll(kmin ^ kmax);  // bits [top, 64) are common
    known = (top >= 64) ? 0ull : (~0ull << top);
    prefix = kmin & known;
    int nmatch = cnt;  // keys that agree with `prefix` on the `known` bits
    for (int b = top - 1; b >= 0; --b) {
      const unsigned long long bit = 1ull << b;
      int c0 = 0;  // matching keys with bit b clear
#pragma unroll
      for (int s = 0; s < PL; ++s) {
        if (s < ns) {  // wave-uniform
          const bool m = ((s * 64 + lane) < cnt) && ((key[s] & known) == prefix) && !(key[s] & bit);
          c0 += __popcll(__ballot(m));
        }
      }
      if (rem > c0) {
        rem -= c0;
        prefix |= bit;
        nmatch -= c0;
      } else {
        nmatch = c0;
      }
      known |= bit;
      if (nmatch == 1) {
        // a single key is left under the prefix: it is the one looked for (rem == 1); fetch it instead of resolving its
        // remaining bits one by one (with well spread scores this ends the search after ~log2(cnt) bits)
        unsigned long long mine = ~0ull;
#pragma unroll
        for (int s = 0; s < PL; ++s)
          if (s < ns && ((s * 64 + lane) < cnt) && ((key[s] & known) == prefix)) mine = key[s];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const unsigned long long x = __shfl_xor(mine, o);
          mine = x < mine ? x : mine;
        }
        prefix = mine;
        known = ~0ull;
        break;
      }
    }
  }
  const unsigned long long thr = prefix;  // the kk-th smallest key; `rem` of the keys equal to it are selected
  int eq_all = 0;
#pragma unroll
  for (int s = 0; s < PL; ++s) eq_all += __popcll(__ballot(((s * 64 + lane) < cnt) && key[s] == thr));
  // selected: key < thr, plus the `rem` keys == thr of smallest column
  unsigned taken = 0;  // bit s: slot s of this lane is an equal key that is selected
  if (eq_all == rem) {
#pragma unroll
    for (int s = 0; s < PL; ++s) taken |= (((s * 64 + lane) < cnt) && key[s] == thr) ? (1u << s) : 0u;
  } else {
    for (int t = 0; t < rem; ++t) {
      int best = 0x7fffffff;
#pragma unroll
      for (int s = 0; s < PL; ++s) {
        const bool m = ((s * 64 + lane) < cnt) && key[s] == thr && !(taken & (1u << s));
        best = (m && col[s] < best) ? col[s] : best;
      }
      int wbest = best;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const int x = __shfl_xor(wbest, o);
        wbest = x < wbest ? x : wbest;
      }
#pragma unroll
      for (int s = 0; s < PL; ++s)
        if (((s * 64 + lane) < cnt) && key[s] == thr && col[s] == wbest && !(taken & (1u << s))) {
          taken |= (1u << s);  // columns are unique within a row's list: exactly one slot of one lane
        }
    }
  }
  // compact the selected candidates to the front of the list (every load above has completed: the positions depend
  // on all of them), lower the threshold
  int base = 0;
  int* fcols = nullptr;
  unsigned* bits = nullptr;
  if (final) {
    bits = reinterpret_cast<unsigned*>(sel_lds + (long)wave * mask_words);
    fcols = reinterpret_cast<int*>(sel_lds + 4l * mask_words) + wave * k;
    for (int w = lane; w < 2 * mask_words; w += 64) bits[w] = 0u;
    wave_lds_fence();
  }
#pragma unroll
  for (int s = 0; s < PL; ++s) {
    const bool sel = ((s * 64 + lane) < cnt) && (key[s] < thr || (taken & (1u << s)));
    const unsigned long long bal = __ballot(sel);
    if (sel) {
      const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
      if (final) {
        fcols[pos] = col[s];
      } else {
        c.cscore[(long)row * c.cap + pos] = key_to_f64(key[s]);
        c.ccol[(long)row * c.cap + pos] = col[s];
      }
    }
    base += __popcll(bal);
  }
  wave_lds_fence();
  if (!final || check) {
    // the k-th candidate in (score, column) order: the selected one with key == thr of LARGEST column
    int cmax = -1;
#pragma unroll
    for (int s = 0; s < PL; ++s) cmax = ((taken & (1u << s)) && col[s] > cmax) ? col[s] : cmax;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const int x = __shfl_xor(cmax, o);
      cmax = x > cmax ? x : cmax;
    }
    if (!final) {
      if (lane == 0) {
        c.count[row] = kk;
        c.tau[row] = key_to_f64(thr);
        c.taucol[row] = cmax;
      }
      return;
    }
    // hop mode: the kept list is complete below (tau, taucol) only -- the k-th must lie inside that region
    if (lane == 0) {
      const double tk = key_to_f64(thr), tr = c.tau[row];
      if (!(tk < tr || (tk == tr && cmax <= c.taucol[row]))) atomicOr(c.overflow, 2);
    }
  }
  // final pass: neighbour list in ascending column order, adjacency bitmask row with the own column cleared
  // (LDS writes and reads of one wave complete in order: no barrier)
  for (int j = lane; j < kk; j += 64) {
    const int cj = fcols[j] - c.col_base;  // window column
    int rank = 0;
    for (int m = 0; m < kk; ++m) rank += fcols[m] < fcols[j];
    if (out_idx) out_idx[(long)wrow * k + rank] = cj;
    if (cj != wrow && cj >= 0 && cj < n) atomicOr(&bits[cj >> 5], 1u << (cj & 31));
  }
  wave_lds_fence();
  if (out_mask) {
    const unsigned long long* b64 = reinterpret_cast<const unsigned long long*>(bits);
    for (int w = lane; w < mask_words; w += 64) out_mask[(long)wrow * mask_words + w] = b64[w];
  }
}

__global__ void cand_init_kernel(double* __restrict__ tau, int* __restrict__ taucol, int* __restrict__ count,
                                 int* __restrict__ overflow, int n, int count0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    tau[i] = __longlong_as_double(0x7ff0000000000000ll);  // +inf: everything is admitted until k candidates are known
    taucol[i] = 0x7fffffff;
    count[i] = count0;  // DIRECT first phase: its fixed number of slots per row
  }
  if (i == 0) *overflow = 0;
}

// Hop mode, start of a window: rows that were in the previous window drop the candidates whose column has left the window
// (absolute id < col_base) -- what stays is still every window column below the row's threshold --; the slots of the rows
// that enter are reset.  One wave per window row.
__global__ __launch_bounds__(256) void cand_expire_kernel(CandArgs c, int n, int count0) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wrow = blockIdx.x * 4 + wave;
  if (wrow >= n) return;
  const int row = cand_slot(c, wrow);
  if (wrow >= c.n_ret) {
    if (lane == 0) {
      c.count[row] = count0;  // (the fixed slots of a DIRECT first phase, or nothing)
      c.tau[row] = __longlong_as_double(0x7ff0000000000000ll);
      c.taucol[row] = 0x7fffffff;
    }
    return;
  }
  int cnt = c.count[row];
  cnt = cnt < c.cap ? cnt : c.cap;
  double* sc = c.cscore + (long)row * c.cap;
  int* cc = c.ccol + (long)row * c.cap;
  int base = 0;
  for (int p0 = 0; p0 < cnt; p0 += 64) {  // in place: a chunk is read before anything of it or behind it is written
    const int p = p0 + lane;
    const bool ok = p < cnt;
    const double v = ok ? sc[p] : 0.0;
    const int col = ok ? cc[p] : 0;
    const bool keep = ok && col >= c.col_base;
    const unsigned long long bal = __ballot(keep);
    if (keep) {
      const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
      sc[pos] = v;
      cc[pos] = col;
    }
    base += __popcll(bal);
  }
  if (lane == 0) c.count[row] = base;
}

struct KnnFusedWs {
  double *norms, *cscore, *tau;
  int *taucol, *count, *ccol, *overflow;
};

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t knn_fused_layout(long n, int cap, char* base, KnnFusedWs* ws) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align256(bytes);
    return p;
  };
  char* p0 = take(8 * (size_t)n);
  char* p1 = take(8 * (size_t)n);
  char* p2 = take(4 * (size_t)n);
  char* p2b = take(4 * (size_t)n);
  char* p3 = take(256);
  char* p4 = take(8 * (size_t)n * cap);
  char* p5 = take(4 * (size_t)n * cap);
  if (ws) {
    ws->norms = (double*)p0; ws->tau = (double*)p1; ws->count = (int*)p2; ws->taucol = (int*)p2b;
    ws->overflow = (int*)p3;
    ws->cscore = (double*)p4; ws->ccol = (int*)p5;
  }
  return off;
}

int inv_norms_launch(double* norms, long n, hipStream_t st);  // knn.hip: sq norms -> 1 / norm (zero norms -> 1)

template <typename T, int METRIC, bool DIRECT>
static int band_launch(const GemmArgs& g, const CandArgs& c, bool vec, int tiles, int d_lo, int nd, hipStream_t st) {
  static std::once_flag once[2];
  static hipError_t err[2];
  const int v = vec ? 1 : 0;
  std::call_once(once[v], [&] {
    const void* fn = vec ? (const void*)knn_band_kernel<T, true, METRIC, DIRECT> : (const void*)knn_band_kernel<T, false, METRIC, DIRECT>;
    err[v] = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
  });
  MUSED_CHECK_HIP(err[v]);
  const int t_new = c.n_ret / GEMM_BM, d_hi_ = d_lo + nd - 1;
  const int ibase_min = t_new > d_hi_ ? t_new - d_hi_ : 0;
  const dim3 grid(cdiv(tiles - ibase_min, 8) * 8 * nd), blk(GEMM_THREADS);
  if (vec) hipLaunchKernelGGL((knn_band_kernel<T, true, METRIC, DIRECT>), grid, blk, GEMM_LDS_BYTES, st, g, c, tiles, d_lo, nd);
  else hipLaunchKernelGGL((knn_band_kernel<T, false, METRIC, DIRECT>), grid, blk, GEMM_LDS_BYTES, st, g, c, tiles, d_lo, nd);
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

// lo_abs: stream position of window row 0 (0 for tumbling use); n_new: 0 = compute the window from scratch, else the number
// of rows at the END of the window that were not in the previous one (hop mode: the rest reuse their kept lists)
template <typename T>
static int knn_fused_t(const T* X, long n, int d, long ld, int k, int metric, const KnnFusedWs& ws, int cap, int* out_idx,
                       unsigned long long* out_mask, int mask_words, hipStream_t st, long lo_abs = 0, int n_new = 0,
                       bool ring = false) {
  const int tiles = cdiv(n, GEMM_BM);
  const int hmax = tiles / 2;
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = X; g.B = X; g.lda = ld; g.ldb = ld; g.M = (int)n; g.N = (int)n; g.K = d;
  const bool hop = ring && n_new > 0 && n_new < n;
  CandArgs c{ws.tau, ws.taucol, ws.count, ws.cscore, ws.ccol, cap, ws.overflow, ws.norms,
             ring ? (int)(lo_abs % n) : 0, ring ? (int)(lo_abs & 0x3fffffff) : 0, hop ? (int)(n - n_new) : 0, (int)n};
  const bool vec = vec_ok<T>(X, ld, 0);
  const int pl = cdiv(cap, 64);
  const size_t sel_lds = 4 * ((size_t)mask_words * 8 + (size_t)k * 4) + 16;
  const int row_lo = hop ? (int)(n - n_new) : 0, check = hop ? 1 : 0;
  // Ring mode keeps a MARGIN for the next window: thresholds (and the compaction between the phases) follow the
  // (2 k)-th smallest score instead of the k-th, so a kept list holds ~2 k (W / columns seen before the last phase) columns
  // below its threshold, of which a hop removes the share that left the window: the k smallest of the next window are
  // then provable from the list with room to spare (the final pass still outputs the k smallest).
  int kthr = k;
  if (ring) {
    kthr = 2 * k;
    if (kthr > cap / 4) kthr = cap / 4;
    if (kthr < k) kthr = k;
  }
  auto select = [&](int final) {
    int* oi = final ? out_idx : nullptr;
    unsigned long long* om = final ? out_mask : nullptr;
    const size_t lds = final ? sel_lds : 0;
    const int ks = final ? k : kthr;
    const dim3 grid(cdiv(n, 4)), blk(256);
    if (pl <= 4) hipLaunchKernelGGL(cand_select_kernel<4>, grid, blk, lds, st, c, (int)n, ks, final, oi, om, mask_words, row_lo, check);
    else if (pl <= 8) hipLaunchKernelGGL(cand_select_kernel<8>, grid, blk, lds, st, c, (int)n, ks, final, oi, om, mask_words, row_lo, check);
    else hipLaunchKernelGGL(cand_select_kernel<16>, grid, blk, lds, st, c, (int)n, ks, final, oi, om, mask_words, row_lo, check);
  };
  auto band = [&](int d_lo, int nd, bool direct) -> int {
    if (metric == 0) return direct ? band_launch<T, 0, true>(g, c, vec, tiles, d_lo, nd, st) : band_launch<T, 0, false>(g, c, vec, tiles, d_lo, nd, st);
    return direct ? band_launch<T, 1, true>(g, c, vec, tiles, d_lo, nd, st) : band_launch<T, 1, false>(g, c, vec, tiles, d_lo, nd, st);
  };
  // Phase schedule.  First phase: the column tiles within cyclic distance a (DIRECT: every score kept, no atomics, if the
  // lists have room for them and no column tile is reached from both sides), a = 2 -> 640 columns per row.  Afterwards a
  // row that has seen m columns admits about k / m of what it is shown: the remaining distances go in ONE launch as soon
  // as twice that estimate fits the lists, otherwise the next phase triples the columns seen.
  int a = 2;
  while (a > 0 && ((2 * a + 1) * 128 > cap || kthr > (2 * a) * 128)) --a;
  // (hop mode: the rows that stay keep their lists and receive pushes; the rows that enter get the fixed slots)
  const bool direct = tiles >= 2 * a + 2 && (2 * a + 1) * 128 <= cap && a <= hmax;
  int d_hi = direct ? a : (hmax < 1 ? hmax : 1);
  if (hop) {
    // (the flag word is NOT cleared: once a list has overflowed or stopped proving its row's k smallest, every window built
    // on this state reports it until a computation from scratch -- n_new = 0 -- resets state and flag; the caller may
    // therefore read the flag late, behind several windows)
    hipLaunchKernelGGL(cand_expire_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, c, (int)n, direct ? (2 * a + 1) * 128 : 0);
  } else {
    hipLaunchKernelGGL(cand_init_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, ws.tau, ws.taucol, ws.count, ws.overflow,
                       (int)n, direct ? (2 * a + 1) * 128 : 0);
  }
  int rc;
  int d_lo = 0;
  while (true) {
    if (d_hi > hmax) d_hi = hmax;
    if ((rc = band(d_lo, d_hi - d_lo + 1, direct && d_lo == 0))) return rc;
    const bool last = d_hi >= hmax;
    select(last ? 1 : 0);
    if (last) break;
    d_lo = d_hi + 1;
    const long seen = (2l * d_hi + 1) * 128, left = (long)n - seen;
    const long est = left > 0 ? ((long)kthr * left + seen - 1) / seen : 0;
    // (ring mode: the threshold follows the (2 k)-th score and the lists have cap = 1024: a phase that triples the columns
    // seen admits ~4 k on average with a wide spread between rows on clustered data; the last phase is given more room)
    if (ring) d_hi = (kthr + 3 * est <= cap) ? hmax : 3 * d_hi + 2;
    else d_hi = (kthr + 2 * est <= cap) ? hmax : 3 * d_hi + 2;
  }
  MUSED_LAUNCH_CHECK();
  return MUSED_OK;
}

}  // namespace mused

using namespace mused;

extern "C" {

// bytes of workspace mused_knn_fused needs for windows of n rows and `cap` candidates per row
long mused_knn_fused_ws_bytes(long n, int cap) {
  if (n <= 0 || cap <= 0) return -1;
  return (long)knn_fused_layout(n, cap, nullptr, nullptr);
}

// Replaces NearestNeighbors(n_neighbors=k).fit(X).kneighbors(X) + the adjacency write loop
// (matrix_operations.py:118-130) and cosine_similarity + argsort[:, :k] (:106-108) for dense rows WITHOUT the n x n
// score matrix: out_idx (n x k int32, ascending columns; may be NULL), out_mask (n x mask_words uint64 bitmask, own
// column cleared; may be NULL).  ws: mused_knn_fused_ws_bytes(n, cap) bytes; cap <= 1024 candidates per row
// (>= 4 k + 128 recommended: a phase admits about 2 k).  *overflow_out (device int) != 0 afterwards: some row collected more than cap
// candidates and the outputs are INVALID -- use mused_knn_topk for that window.
int mused_knn_fused(const void* X, int dtype, long n, int d, long ld, int k, int metric, void* ws, long ws_bytes, int cap,
                    int* out_idx, unsigned long long* out_mask, int mask_words, int* overflow_out, void* stream) {
  MUSED_REQUIRE(X && ws && n > 0 && d > 0 && ld >= d && k >= 1 && k <= n, "mused_knn_fused: bad arguments (k=%d n=%ld)", k, n);
  MUSED_REQUIRE(metric == 0 || metric == 1, "mused_knn_fused: metric must be 0 (l2) or 1 (cosine)");
  MUSED_REQUIRE(cap >= k && cap <= 1024, "mused_knn_fused: need k <= cap <= 1024");
  MUSED_REQUIRE(n < (1l << 31) && (!out_mask || mask_words >= (n + 63) / 64), "mused_knn_fused: mask_words too small");
  MUSED_REQUIRE(ws_bytes >= (long)knn_fused_layout(n, cap, nullptr, nullptr), "mused_knn_fused: workspace too small");
  MUSED_REQUIRE((size_t)4 * ((size_t)mask_words * 8 + (size_t)k * 4) + 16 <= 64 * 1024, "mused_knn_fused: window too long for the LDS bit rows");
  hipStream_t st = (hipStream_t)stream;
  KnnFusedWs w;
  knn_fused_layout(n, cap, (char*)ws, &w);
  int rc = mused_row_sq_norms(X, dtype, n, d, ld, w.norms, stream);
  if (rc) return rc;
  if (metric == 1 && (rc = inv_norms_launch(w.norms, n, st))) return rc;
  if (dtype == MUSED_F32) rc = knn_fused_t<float>((const float*)X, n, d, ld, k, metric, w, cap, out_idx, out_mask, mask_words, st);
  else if (dtype == MUSED_F64) rc = knn_fused_t<double>((const double*)X, n, d, ld, k, metric, w, cap, out_idx, out_mask, mask_words, st);
  else {
    set_error("mused_knn_fused: unsupported dtype %d", dtype);
    return MUSED_ERR_UNSUPPORTED;
  }
  if (rc) return rc;
  if (overflow_out) MUSED_CHECK_HIP(hipMemcpyAsync(overflow_out, w.overflow, 4, hipMemcpyDeviceToDevice, st));
  return MUSED_OK;
}

// Hopping windows (step_window_ratio > 1, main.py:32): the same result as mused_knn_fused for the window
// whose row 0 is stream row `lo_abs`, reusing what the previous call on the SAME workspace (for the window n_new rows
// earlier) left behind: a row that stays in the window keeps its candidate list -- every window column below its threshold --
// minus the columns that left; only the tiles that involve one of the n_new rows at the end of the window are computed
// (1 - (1 - n_new / n)^2 of them), which serve the entering rows (all their columns) and the staying rows (their scores against
// the entering columns).  n_new = 0 (or >= n): compute from scratch and leave the state for the next call.
// *flag_out (device int): bit 0 a list overflowed, bit 1 a kept list no longer proves its row's k smallest -- in either case
// the outputs are INVALID and the call has to be repeated with n_new = 0.  The flag is STICKY: every later call that reuses
// this state reports it too, until a call with n_new = 0 -- so a caller may enqueue several windows before it reads the flag
// of the first and then repeat the flagged ones.  The workspace must not be used for anything else
// in between; X must hold the n window rows in stream order (row i = stream row lo_abs + i).
int mused_knn_fused_hop(const void* X, int dtype, long n, int d, long ld, int k, int metric, void* ws, long ws_bytes, int cap,
                        long lo_abs, int n_new, int* out_idx, unsigned long long* out_mask, int mask_words, int* flag_out,
                        void* stream) {
  MUSED_REQUIRE(X && ws && n > 0 && d > 0 && ld >= d && k >= 1 && k <= n, "mused_knn_fused_hop: bad arguments (k=%d n=%ld)", k, n);
  MUSED_REQUIRE(metric == 0 || metric == 1, "mused_knn_fused_hop: metric must be 0 (l2) or 1 (cosine)");
  MUSED_REQUIRE(cap >= k && cap <= 1024, "mused_knn_fused_hop: need k <= cap <= 1024");
  MUSED_REQUIRE(lo_abs >= 0 && n_new >= 0, "mused_knn_fused_hop: negative stream position / row count");
  MUSED_REQUIRE(lo_abs + n < (1l << 30), "mused_knn_fused_hop: stream positions beyond 2^30 (column ids are 32-bit)");
  MUSED_REQUIRE(n < (1l << 30) && (!out_mask || mask_words >= (n + 63) / 64), "mused_knn_fused_hop: mask_words too small");
  MUSED_REQUIRE(ws_bytes >= (long)knn_fused_layout(n, cap, nullptr, nullptr), "mused_knn_fused_hop: workspace too small");
  MUSED_REQUIRE((size_t)4 * ((size_t)mask_words * 8 + (size_t)k * 4) + 16 <= 64 * 1024, "mused_knn_fused_hop: window too long for the LDS bit rows");
  hipStream_t st = (hipStream_t)stream;
  KnnFusedWs w;
  knn_fused_layout(n, cap, (char*)ws, &w);
  int rc = mused_row_sq_norms(X, dtype, n, d, ld, w.norms, stream);
  if (rc) return rc;
  if (metric == 1 && (rc = inv_norms_launch(w.norms, n, st))) return rc;
  const int nn = (n_new >= n) ? 0 : n_new;
  if (dtype == MUSED_F32) rc = knn_fused_t<float>((const float*)X, n, d, ld, k, metric, w, cap, out_idx, out_mask, mask_words, st, lo_abs, nn, true);
  else if (dtype == MUSED_F64) rc = knn_fused_t<double>((const double*)X, n, d, ld, k, metric, w, cap, out_idx, out_mask, mask_words, st, lo_abs, nn, true);
  else {
    set_error("mused_knn_fused_hop: unsupported dtype %d", dtype);
    return MUSED_ERR_UNSUPPORTED;
  }
  if (rc) return rc;
  if (flag_out) MUSED_CHECK_HIP(hipMemcpyAsync(flag_out, w.overflow, 4, hipMemcpyDeviceToDevice, st));
  return MUSED_OK;
}

}  // extern "C"
