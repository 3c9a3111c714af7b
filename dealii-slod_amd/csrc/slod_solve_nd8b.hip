// k_solve_nd<8, T>, T = 4, 5 (see slod_solve_nd.hip)
#include "slod_solve_nd.hip.h"

hipError_t slod_launch_nd8b(int T, const SlodKernelArgs &a, int n_patches, size_t lds, hipStream_t st)
{
  if (T == 4)
    return launch_nd<8, 4>(a, n_patches, lds, st);
  if (T == 5)
    return launch_nd<8, 5>(a, n_patches, lds, st);
  return hipErrorInvalidValue;
}
