// k_solve_nd<8, T>, T = 2, 3 (see slod_solve_nd.hip)
#include "slod_solve_nd.hip.h"

hipError_t slod_launch_nd8a(int T, const SlodKernelArgs &a, int n_patches, size_t lds, hipStream_t st)
{
  if (T == 2)
    return launch_nd<8, 2>(a, n_patches, lds, st);
  if (T == 3)
    return launch_nd<8, 3>(a, n_patches, lds, st);
  return hipErrorInvalidValue;
}
