// Sustained rate of v_mfma_f64_16x16x4_f64 on this chip: every wave runs a register-only loop of MFMAs on 16 independent
// accumulators (no memory traffic), 1 / 2 waves per SIMD on all CUs.  Reference point for the GEMM kernels' "fraction of peak".
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/mfma_f64_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma_loop(double* out, int iters, double a0, double b0, long long* clk) {
  const long long c0 = clock64(), w0 = wall_clock64();
  v4f64 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (v4f64){0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#ifdef ACC_VGPR
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));  // accumulators forced into VGPRs
#else
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);  // (the compiler puts them into AGPRs here)
#endif
    }
  }
  double s = 0.0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}
int main() {
  double* out;
  long long* clk;
  (void)hipMalloc(&out, sizeof(double) * 256 * 4096);
  (void)hipMalloc(&clk, 16);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grids[] = {8, 32, 128, 256, 512};
  for (int gi = 0; gi < 5; ++gi) {
    const int iters = 20000, grid = grids[gi];
    hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, out, 100, 1.0, 1.0, (long long*)nullptr);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1.0, clk);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2];
    (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flops = (double)grid * 4 /*waves*/ * iters * 16.0 * (16.0 * 16.0 * 4.0 * 2.0);
    const double wgs_per_cu = grid / 256.0;
    printf("%3d workgroups of 4 waves (%.2f per CU): %7.3f ms, %6.1f TFLOP/s fp64 MFMA = %5.1f GFLOP/s per busy CU; clock64 / wall_clock64 "
           "ticks of workgroup 0: %lld / %lld (x 100 MHz = %.0f MHz if clock64 counts core cycles); cycles per MFMA and SIMD: %.1f\n",
           grid, wgs_per_cu, ms, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e9 / (grid < 256 ? grid : 256), h[0], h[1],
           100.0 * (double)h[0] / (double)h[1], (double)h[0] / ((double)iters * 16.0 * (grid > 256 ? 2 : 1)));
  }
  return 0;
}
