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
#include "slod_common.hip.h"

namespace
{
  // ---------------------------------------------------------------------------------
  // K1: stencil assembly.  One thread per patch node gathers its <= 4 elements.
  // stencil slot layout: [(dir*S + a)*S + b][nn_max], dir = (dy+1)*3 + (dx+1)
  // ---------------------------------------------------------------------------------
  template <int S>
  __global__ __launch_bounds__(256) void k_assemble(const SlodKernelArgs A)
  {
    const SlodPatchDesc d    = A.desc[blockIdx.y];
    const int           npx  = d.nx + 1;
    const int           node = blockIdx.x * 256 + threadIdx.x;
    if (node >= npx * (d.ny + 1))
      return;
    const int ix = node % npx, iy = node / npx;
    double    acc[9][S][S];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
      for (int a = 0; a < S; ++a)
#pragma unroll
        for (int b = 0; b < S; ++b)
          acc[i][a][b] = 0.0;
    const double *c0 = A.coef0 + (size_t)d.prob * A.coef_stride;
    const double *c1 = (S == 2) ? A.coef1 + (size_t)d.prob * A.coef_stride : nullptr;
#pragma unroll
    for (int ay = 0; ay < 2; ++ay)
#pragma unroll
      for (int ax = 0; ax < 2; ++ax)
        {
          const int ex = ix - ax, ey = iy - ay;
          if (ex < 0 || ex >= d.nx || ey < 0 || ey >= d.ny)
            continue;
          const size_t ge = ((size_t)(d.oy + ey) * A.NE + (size_t)(d.ox + ex)) * 4;
          const int    a  = ax + 2 * ay;
          double       al[4], mu[4];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            {
              al[q] = c0[ge + q];
              mu[q] = (S == 2) ? c1[ge + q] : 0.0;
            }
#pragma unroll
          for (int q = 0; q < 4; ++q)
            {
              double g[2][4];
              hat_gradients(q, g[0], g[1]);
#pragma unroll
              for (int b = 0; b < 4; ++b)
                {
                  const int    bx = b & 1, by = b >> 1;
                  const int    dir = (by - ay + 1) * 3 + (bx - ax + 1);
                  const double gg  = g[0][a] * g[0][b] + g[1][a] * g[1][b];
                  if (S == 1)
                    acc[dir][0][0] += al[q] * (gg * 0.25);
                  else
                    {
#pragma unroll
                      for (int ca = 0; ca < S; ++ca)
#pragma unroll
                        for (int cb = 0; cb < S; ++cb)
                          {
                            const double sym = ((ca == cb) ? gg : 0.0) + g[cb][a] * g[ca][b];
                            const double dv  = g[ca][a] * g[cb][b];
                            acc[dir][ca][cb] += (mu[q] * sym + al[q] * dv) * 0.25;
                          }
                    }
                }
            }
        }
    double *st = A.st + (size_t)blockIdx.y * A.st_stride;
#pragma unroll
    for (int dir = 0; dir < 9; ++dir)
#pragma unroll
      for (int a = 0; a < S; ++a)
#pragma unroll
        for (int b = 0; b < S; ++b)
          st[(size_t)((dir * S + a) * S + b) * A.nn_max + node] = acc[dir][a][b];
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
