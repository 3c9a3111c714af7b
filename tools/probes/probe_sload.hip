// Probe: latency of dependent scalar loads (s_load_dwordx16) of a per-wave 3.5 KB block, as k_solve_nd's
// banded substitution issues them: first pass (cold scalar cache), second pass (re-read), with every CU busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef const __attribute__((address_space(4))) double cdouble_t;
__global__ __launch_bounds__(256, 4) void k(double* buf, long long* out, int write_first, int stride_dbl)
{
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  double* blk = buf + ((size_t)blockIdx.x * 4 + wave) * stride_dbl;
  if (write_first)
    {
      for (int i = lane; i < 448; i += 64) blk[i] = 1.0 + i * 1e-3;
      __syncthreads();
    }
  double acc = lane;
  long long t[3];
  for (int pass = 0; pass < 2; ++pass)
    {
      t[pass] = __builtin_amdgcn_s_memtime();
      for (int r = 0; r < 56; ++r)
        {
          const double* p = blk + r * 8;
          asm volatile("" : "+s"(p) : "v"(acc));
          cdouble_t* q = (cdouble_t*)p;
#pragma unroll
          for (int k = 0; k < 8; ++k) acc = fma(q[k], 1e-9, acc);
        }
    }
  t[2] = __builtin_amdgcn_s_memtime();
  if (lane == 0)
    {
      out[((size_t)blockIdx.x * 4 + wave) * 2 + 0] = t[1] - t[0];
      out[((size_t)blockIdx.x * 4 + wave) * 2 + 1] = t[2] - t[1];
    }
  if (acc == 12345.678) buf[0] = acc;
}
int main()
{
  const int nb = 1024, stride = 448;
  double* buf; long long* out;
  hipMalloc(&buf, (size_t)nb * 4 * stride * 8 * 2);
  hipMalloc(&out, (size_t)nb * 4 * 2 * 8);
  hipMemset(buf, 0, (size_t)nb * 4 * stride * 8 * 2);
  std::vector<long long> h(nb * 4 * 2);
  for (int wf = 0; wf < 2; ++wf)
    for (int rep = 0; rep < 2; ++rep)
      {
        hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, buf, out, wf, stride);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        double a = 0, b = 0;
        for (int i = 0; i < nb * 4; ++i) { a += h[2 * i]; b += h[2 * i + 1]; }
        printf("write_first=%d rep=%d: pass0 %.0f cycles per load, pass1 %.0f cycles per load (56 dependent loads + 8 FMA each)\n", wf, rep,
               a / (nb * 4) / 56, b / (nb * 4) / 56);
      }
  return 0;
}
