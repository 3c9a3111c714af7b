// k_solve_nd: host side (plan-time sizes, launch dispatch) and the NV = 4 instantiations; the kernel is in
// slod_solve_nd.hip.h, the NV = 8 instantiations in slod_solve_nd8a/b.hip (one object each: build time).
#include "slod_solve_nd.hip.h"

#include <algorithm>

// ---- host side -------------------------------------------------------------------------------
int slod_solve_nd_cell(int S, int n_sub, int m_max, int L_max)
{
  // cell size of the dissection, 0 = this kernel does not take the plan
  if (S != 1 || (n_sub != 4 && n_sub != 8))
    return 0;
  const int nv = n_sub, T = slod_solve_ws_tile(m_max);
  if (T <= 0 || T > 5 || ((m_max + 1) / nv - 1) * (nv - 1) > 32)
    return 0;
  (void)L_max;
  return nv;
}

size_t slod_solve_nd_scratch(int nv, int m_max, int L_max, int nc_max)
{
  return nd_layout(nv, slod_solve_ws_tile(m_max), m_max, L_max, nc_max).total;
}

size_t slod_solve_nd_lds_bytes(int nv, int m_max, int nc_max)
{
  // the largest of: SIMT phases (per wave: pivot row, ring couplings, parked rows), factor lines,
  // skeleton front (three m x m blocks, edge block, coupling / right-hand-side block, pivot row)
  const int    T = slod_solve_ws_tile(m_max), MP = 8 * T, NW = ND_WAVES, ncp = (nc_max + 15) & ~15;
  const size_t a = (size_t)NW * (MP + 4 * nv * 4 + ((nv - 1) * (nv - 1) > 32 ? 24 * (4 * nv + 2) : 0));
  const size_t b = (size_t)NW * (64 / (nv + 1) + 1) * 2 * (nv + 1);
  const size_t c = (size_t)3 * MP * (MP + 1) + 32 * 33 + (size_t)32 * (2 * MP + ncp) + MP;
  return std::max(a, std::max(b, c)) * sizeof(double);
}

hipError_t slod_launch_nd8a(int T, const SlodKernelArgs &a, int n_patches, size_t lds, hipStream_t st);
hipError_t slod_launch_nd8b(int T, const SlodKernelArgs &a, int n_patches, size_t lds, hipStream_t st);

hipError_t slod_launch_solve_nd(const SlodKernelArgs &a, int n_patches, size_t lds, hipStream_t st)
{
  const int T = slod_solve_ws_tile(a.m_max);
  if (a.nv == 4)
    switch (T)
      {
        case 2:
          return launch_nd<4, 2>(a, n_patches, lds, st);
        case 3:
          return launch_nd<4, 3>(a, n_patches, lds, st);
        case 4:
          return launch_nd<4, 4>(a, n_patches, lds, st);
        case 5:
          return launch_nd<4, 5>(a, n_patches, lds, st);
      }
  if (a.nv == 8)
    return T <= 3 ? slod_launch_nd8a(T, a, n_patches, lds, st) : slod_launch_nd8b(T, a, n_patches, lds, st);
  return hipErrorInvalidValue;
}
