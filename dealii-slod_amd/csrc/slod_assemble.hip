// HIP kernels (gfx950 / CDNA4) of the SLOD per-patch basis construction.
//
// One workgroup (256 threads = 4 wave64) owns one oversampling patch; 1024 patches of the
// north-star configuration give 4 co-resident workgroups per CU on the 256 CUs of an
// MI355X.  Reference being replaced: the body of the patch loop of
// LOD<dim,spacedim>::compute_basis_function_candidates() (source/LOD.cc:345-767).
//
//   k_assemble : FE_Q_iso_Q1 sub-element stiffness -> 9-point block stencil
//                (Diffusion.h:143-204, Elasticity.h:197-296; LOD.cc:440-444)
//   k_solve    : X_I = A_II^{-1} P^T_I for all N_c right-hand sides (LOD.cc:512-546,
//                LODtools.h:511-595) as a block-tridiagonal (grid-line) elimination:
//                per line an m x m Schur complement is inverted in registers by a
//                symmetric Gauss-Jordan sweep, pivots broadcast through LDS.
//   k_select   : M = P^T X / H^2, D = M^-1 (LOD.cc:548-553); LOD pick (LOD.cc:566-595) or
//                SLOD boundary trace + one-sided Jacobi SVD least squares with the
//                0.5-truncation loop (LOD.cc:598-757); normalise; psi = A_semi phi
//                (LOD.cc:758-765).
#include "slod_assemble.hip.h"

namespace
{
  // stand-alone launch: one thread per patch node (the default solver assembles its own patch)
  template <int S>
  __global__ __launch_bounds__(256) void k_assemble(const SlodKernelArgs A)
  {
    const SlodPatchDesc d    = A.desc[blockIdx.y];
    const int           node = blockIdx.x * 256 + threadIdx.x;
    if (node < (d.nx + 1) * (d.ny + 1))
      assemble_node<S>(A, d, blockIdx.y, node);
  }
} // namespace

hipError_t slod_launch_assemble(int S, const SlodKernelArgs &a, int n_patches, hipStream_t st)
{
  const dim3 grid((a.nn_max + 255) / 256, n_patches);
  if (S == 1)
    hipLaunchKernelGGL(k_assemble<1>, grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(k_assemble<2>, grid, dim3(256), 0, st, a);
  return hipGetLastError();
}
