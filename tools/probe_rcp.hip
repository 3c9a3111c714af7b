#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* x, double* r0, double* r1, long long* cyc) {
  int i = threadIdx.x + blockIdx.x * blockDim.x;
  double d = x[i];
  double a = __builtin_amdgcn_rcp(d);
  r0[i] = a;
  double e = fma(-d, a, 1.0); a = fma(a, e, a);
  r1[i] = a;
  if (i == 0) {
    // dependent fma chain latency
    double v = d; long long t0 = clock64();
#pragma unroll
    for (int k = 0; k < 256; ++k) v = fma(v, 1.0000001, 1e-9);
    long long t1 = clock64();
    double w = d;
#pragma unroll
    for (int k = 0; k < 64; ++k) w = __builtin_amdgcn_rcp(w) + 1.0;
    long long t2 = clock64();
    cyc[0] = t1 - t0; cyc[1] = t2 - t1; r1[0] = v + w;
  }
}
int main() {
  const int n = 1 << 16; double *hx = new double[n], *h0 = new double[n], *h1 = new double[n];
  srand(1); for (int i = 0; i < n; ++i) hx[i] = ldexp(1.0 + rand() / (double)RAND_MAX, rand() % 40 - 20);
  double *dx, *d0, *d1; long long *dc, hc[2];
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&dc, 16);
  hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, d0, d1, dc); hipDeviceSynchronize();
  hipMemcpy(h0, d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0; for (int i = 1; i < n; ++i) { e0 = fmax(e0, fabs(h0[i] * hx[i] - 1.0)); e1 = fmax(e1, fabs(h1[i] * hx[i] - 1.0)); }
  printf("max rel err: rcp %.3e, rcp+1 newton %.3e; 256 dependent fma: %lld cycles (%.1f each); 64 x (rcp+add): %lld (%.1f each)\n", e0, e1, hc[0], hc[0] / 256.0, hc[1], hc[1] / 64.0);
}
