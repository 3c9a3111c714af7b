// k_select: stand-alone launch of the selection stage (select_patch, slod_select.hip.h).
#include "slod_select.hip.h"

namespace
{
  template <int S>
  __global__ __launch_bounds__(256, S == 1 ? 1 : 2) void k_select(const SlodKernelArgs A, int nb_max, int nf_max)
  {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    select_patch<S, true>(A, nb_max, nf_max, blockIdx.x, smem);
  }
} // namespace

size_t slod_select_lds_bytes(int /*S*/, int nb_max, int nc_max, int nf_max)
{
  // must mirror the carve-up at the top of k_select
  const size_t bd = (size_t)nb_max * nc_max > (size_t)nf_max ? (size_t)nb_max * nc_max : (size_t)nf_max;
  const size_t n  = (size_t)nc_max * (nc_max + 1) + (size_t)nc_max * nc_max + bd + 5 * (size_t)nc_max + 8 +
                   2 * (size_t)(nb_max > 160 ? nb_max : 160);
  return n * sizeof(double) + (5 * (size_t)nc_max + 4) * sizeof(int);
}

hipError_t slod_launch_select(int S, const SlodKernelArgs &a, int n_patches, int nb_max, int nf_max,
                              hipStream_t st)
{
  const size_t lds = slod_select_lds_bytes(S, nb_max, a.nc_max, nf_max);
  hipError_t   e;
  if (S == 1)
    {
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_select<1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess)
        return e;
      hipLaunchKernelGGL(k_select<1>, dim3(n_patches), dim3(256), lds, st, a, nb_max, nf_max);
    }
  else
    {
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_select<2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess)
        return e;
      hipLaunchKernelGGL(k_select<2>, dim3(n_patches), dim3(256), lds, st, a, nb_max, nf_max);
    }
  return hipGetLastError();
}
