// K1 as a device function: FE_Q_iso_Q1 sub-element stiffness -> 9-point block stencil of one patch node.
#ifndef SLOD_ASSEMBLE_HIP_H
#define SLOD_ASSEMBLE_HIP_H
#include "slod_common.hip.h"

namespace
{
  // ---------------------------------------------------------------------------------
  // K1: stencil assembly.  One thread per patch node gathers its <= 4 elements.
  // stencil slot layout: [(dir*S + a)*S + b][nn_max], dir = (dy+1)*3 + (dx+1)
  // ---------------------------------------------------------------------------------
  template <int S>
  __device__ __forceinline__ void assemble_node(const SlodKernelArgs &A, const SlodPatchDesc &d, const int patch,
                                                const int node)
  {
    const int npx = d.nx + 1;
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
    double *st = A.st + (size_t)patch * A.st_stride;
#pragma unroll
    for (int dir = 0; dir < 9; ++dir)
#pragma unroll
      for (int a = 0; a < S; ++a)
#pragma unroll
        for (int b = 0; b < S; ++b)
          st[(size_t)((dir * S + a) * S + b) * A.nn_max + node] = acc[dir][a][b];
  }
} // namespace

#endif
