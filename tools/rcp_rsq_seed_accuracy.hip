#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* x, double* r, double* s, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { r[i] = __builtin_amdgcn_rcp(x[i]); s[i] = __builtin_amdgcn_rsq(x[i]); }
}
int main() {
  const int n = 1 << 20;
  double *hx = new double[n], *hr = new double[n], *hs = new double[n];
  for (int i = 0; i < n; ++i) hx[i] = exp((double)i / n * 40.0 - 20.0) * (1.0 + 0.37 * (i % 97) / 97.0);
  double *dx, *dr, *ds;
  hipMalloc(&dx, 8 * n); hipMalloc(&dr, 8 * n); hipMalloc(&ds, 8 * n);
  hipMemcpy(dx, hx, 8 * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dr, ds, n);
  hipMemcpy(hr, dr, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(hs, ds, 8 * n, hipMemcpyDeviceToHost);
  double er = 0, es = 0;
  for (int i = 0; i < n; ++i) {
    er = fmax(er, fabs(hr[i] * hx[i] - 1.0));
    es = fmax(es, fabs(hs[i] * sqrt(hx[i]) - 1.0));
  }
  printf("max rel err rcp %.3e rsq %.3e\n", er, es);
  return 0;
}
