// Wave-level data movement on fp64 values without the LDS crossbar: DPP moves and v_permlane{16,32}_swap (gfx950).
// Shared by the Jacobi kernels (eig.hip) and the tridiagonalisation solver (trd.hip).
#pragma once
#include "common.h"

namespace mused {

__device__ __forceinline__ double f64_from_parts(unsigned lo, unsigned hi) {
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  // every lane has a valid source under these controls: no "old" value to initialise (mov_dpp, not update_dpp)
  const int lo = __builtin_amdgcn_mov_dpp((int)(unsigned)(u & 0xffffffffull), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_mov_dpp((int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, false);
  return f64_from_parts((unsigned)lo, (unsigned)hi);
}
// a <- a(l) + a(l ^ 32) on lanes 0-31, b(l) + b(l ^ 32) on lanes 32-63 (a = kept-by-low, b = kept-by-high)
__device__ __forceinline__ double swap32_add(double a, double b) {
  const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
  auto rl = __builtin_amdgcn_permlane32_swap((unsigned)(ua & 0xffffffffull), (unsigned)(ub & 0xffffffffull), false, false);
  auto rh = __builtin_amdgcn_permlane32_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
  return f64_from_parts(rl[0], rh[0]) + f64_from_parts(rl[1], rh[1]);
}
__device__ __forceinline__ double swap16_add(double a, double b) {
  const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
  auto rl = __builtin_amdgcn_permlane16_swap((unsigned)(ua & 0xffffffffull), (unsigned)(ub & 0xffffffffull), false, false);
  auto rh = __builtin_amdgcn_permlane16_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
  return f64_from_parts(rl[0], rh[0]) + f64_from_parts(rl[1], rh[1]);
}
constexpr int DPP_ROW_MIRROR = 0x140, DPP_ROW_HALF_MIRROR = 0x141, DPP_QUAD_XOR2 = 0x4E, DPP_QUAD_XOR1 = 0xB1;


// pure exchanges (no add): value of lane l ^ 32 / l ^ 16
__device__ __forceinline__ double xchg32(double a) {
  const unsigned long long ua = (unsigned long long)__double_as_longlong(a);
  auto rl = __builtin_amdgcn_permlane32_swap((unsigned)(ua & 0xffffffffull), (unsigned)(ua & 0xffffffffull), false, false);
  auto rh = __builtin_amdgcn_permlane32_swap((unsigned)(ua >> 32), (unsigned)(ua >> 32), false, false);
  // lanes 0-31 receive in [1] what lanes 32-63 held, lanes 32-63 receive in [0] what lanes 0-31 held
  const bool hi = (threadIdx.x & 32) != 0;
  return hi ? f64_from_parts(rl[0], rh[0]) : f64_from_parts(rl[1], rh[1]);
}
__device__ __forceinline__ double xchg16(double a) {
  const unsigned long long ua = (unsigned long long)__double_as_longlong(a);
  auto rl = __builtin_amdgcn_permlane16_swap((unsigned)(ua & 0xffffffffull), (unsigned)(ua & 0xffffffffull), false, false);
  auto rh = __builtin_amdgcn_permlane16_swap((unsigned)(ua >> 32), (unsigned)(ua >> 32), false, false);
  const bool hi = (threadIdx.x & 16) != 0;
  return hi ? f64_from_parts(rl[0], rh[0]) : f64_from_parts(rl[1], rh[1]);
}
constexpr int DPP_ROW_ROR8 = 0x128;  // lane l <- lane (l + 8) mod 16 of its row: the value of l ^ 8

// sum over the 64 lanes, result in every lane (6 stages, no LDS)
__device__ __forceinline__ double wave_allsum(double v) {
  v = v + dpp_mov_f64<DPP_QUAD_XOR1>(v);
  v = v + dpp_mov_f64<DPP_QUAD_XOR2>(v);
  v = v + dpp_mov_f64<DPP_ROW_HALF_MIRROR>(v);
  v = v + dpp_mov_f64<DPP_ROW_MIRROR>(v);
  v = v + xchg16(v);
  v = v + xchg32(v);
  return v;
}

}  // namespace mused
