// Issue cost and dependent latency of the fp64 vector instructions the Jacobi kernels are made of (gfx950).
// One workgroup of `waves_per_simd * 4` waves on one CU; every wave runs a loop of CHAINS independent dependency
// chains; s_memtime around the loop (shader cycles).  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/fp64_probe ...
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int CHAINS, int OP>
__global__ void probe(double* out, long long* cyc, int iters, double a, double b) {
  double x[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) x[c] = a + c + threadIdx.x;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        if (OP == 0) x[c] = fma(x[c], b, a);                       // v_fma_f64
        if (OP == 1) x[c] = x[c] * b;                              // v_mul_f64
        if (OP == 2) x[c] = x[c] + b;                              // v_add_f64
        if (OP == 3) x[c] = __builtin_amdgcn_rsq(x[c]) + a;        // v_rsq_f64 (+ add)
        if (OP == 4) x[c] = __builtin_amdgcn_rcp(x[c]) + a;        // v_rcp_f64 (+ add)
        if (OP == 5) x[c] = __shfl_xor(x[c], 1) + b;               // DPP quad_perm pair + add
        if (OP == 6) x[c] = __shfl_xor(x[c], 32) + b;              // permlane32 swap / bpermute + add
        if (OP == 7) { float f = (float)x[c]; f = fmaf(f, 1.0001f, 0.5f); x[c] = f; }  // cvt + v_fma_f32 + cvt
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CHAINS, int OP>
static void run(const char* name, int waves_per_simd) {
  const int threads = 64 * 4 * waves_per_simd, iters = 2000;
  double* out; long long* cyc;
  hipMalloc(&out, sizeof(double) * threads);
  hipMalloc(&cyc, sizeof(long long));
  probe<CHAINS, OP><<<1, threads>>>(out, cyc, iters, 1.0000001, 0.9999999);
  probe<CHAINS, OP><<<1, threads>>>(out, cyc, iters, 1.0000001, 0.9999999);
  hipDeviceSynchronize();
  long long c; hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
  const double per = (double)c / ((double)iters * 16 * CHAINS);
  printf("%-28s chains=%2d waves/SIMD=%d : %7.2f cycles per op per wave (s_memtime ticks), %7.2f per op per SIMD\n", name, CHAINS,
         waves_per_simd, per, per / waves_per_simd);
  hipFree(out); hipFree(cyc);
}

int main() {
  const char* names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rsq_f64+add", "v_rcp_f64+add", "shfl_xor1(f64)+add",
                         "shfl_xor32(f64)+add", "cvt+fma_f32+cvt"};
  for (int w = 1; w <= 2; ++w) {
    run<1, 0>(names[0], w); run<2, 0>(names[0], w); run<4, 0>(names[0], w); run<8, 0>(names[0], w); run<16, 0>(names[0], w);
    run<1, 1>(names[1], w); run<8, 1>(names[1], w);
    run<1, 2>(names[2], w); run<8, 2>(names[2], w);
    run<1, 3>(names[3], w); run<8, 3>(names[3], w);
    run<1, 4>(names[4], w); run<8, 4>(names[4], w);
    run<1, 5>(names[5], w); run<8, 5>(names[5], w);
    run<1, 6>(names[6], w); run<8, 6>(names[6], w);
    run<1, 7>(names[7], w); run<8, 7>(names[7], w);
  }
  return 0;
}
