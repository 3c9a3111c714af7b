// Device helpers shared by the SLOD kernels (gfx950).  See slod_assemble/solve_*/select.hip.
#ifndef SLOD_COMMON_HIP_H
#define SLOD_COMMON_HIP_H
#include "slod_device.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace
{
  constexpr double kG0 = 0.21132486540518711775; // (1 - 1/sqrt(3))/2, QGauss<1>(2)
  constexpr double kG1 = 0.78867513459481288225; // (1 + 1/sqrt(3))/2

  __device__ __forceinline__ void hat_gradients(int q, double gx[4], double gy[4])
  {
    const double xi = (q & 1) ? kG1 : kG0, eta = (q & 2) ? kG1 : kG0;
    gx[0] = -(1.0 - eta);
    gx[1] = (1.0 - eta);
    gx[2] = -eta;
    gx[3] = eta;
    gy[0] = -(1.0 - xi);
    gy[1] = -xi;
    gy[2] = (1.0 - xi);
    gy[3] = xi;
  }

  // column k of P^T <-> coarse cell of the patch, reference order: centre first, then
  // x-offset outer / y-offset inner (LOD.cc:151-178)
  __device__ __forceinline__ void cell_of_col(const SlodPatchDesc &d, int k, int &kx, int &ky)
  {
    const int c0 = d.ccx * d.my + d.ccy;
    const int t  = (k == 0) ? c0 : ((k <= c0) ? k - 1 : k);
    kx           = t / d.my;
    ky           = t - kx * d.my;
  }

  // entry of the un-zeroed P^T / (h^2/4) (LODtools.h:24-67, LOD.cc:478-495)
  template <int S>
  __device__ __forceinline__ double pt_weight(const SlodPatchDesc &d, int n, int quirk, int ix,
                                              int iy, int comp, int col)
  {
    const int cc = col % S, k = col / S;
    int       kx, ky;
    cell_of_col(d, k, kx, ky);
    const int jx = ix - kx * n, jy = iy - ky * n;
    if (jx < 0 || jx > n || jy < 0 || jy > n)
      return 0.0;
    const bool   ex = (jx == 0 || jx == n), ey = (jy == 0 || jy == n);
    const double w = (ex ? 1.0 : 2.0) * (ey ? 1.0 : 2.0);
    if (S == 1)
      return w;
    int par = comp;
    if (quirk && !(ex && ey))
      {
        // row parity inside FESystem(FE_Q_iso_Q1(n),2): line dofs [c0 x (n-1), c1 x (n-1)],
        // quad dofs [c0 x (n-1)^2, c1 x (n-1)^2] (LODtools.h:43-67 assumes interleaving)
        if (ex || ey)
          par = (comp * (n - 1) + (ex ? jy - 1 : jx - 1)) & 1;
        else
          par = (comp * (n - 1) * (n - 1) + (jx - 1) + (jy - 1) * (n - 1)) & 1;
      }
    return (par == cc) ? w : 0.0;
  }

  // coupling between dof (l,i) and dof (l+dl, i+o) of the interior grid-line numbering
  template <int S>
  __device__ __forceinline__ double coupling(const double *st, int nn_max, int npx, bool tr, int m,
                                             int l, int i, int dl, int o)
  {
    const int j = i + o;
    if (j < 0 || j >= m)
      return 0.0;
    const int pi = i / S, ci = i - pi * S, pj = j / S, cj = j - pj * S, dp = pj - pi;
    if (dp < -1 || dp > 1)
      return 0.0;
    const int ix = tr ? l + 1 : pi + 1, iy = tr ? pi + 1 : l + 1;
    const int dx = tr ? dl : dp, dy = tr ? dp : dl;
    const int dir = (dy + 1) * 3 + dx + 1;
    return st[(size_t)((dir * S + ci) * S + cj) * nn_max + ix + iy * npx];
  }

  constexpr int kColGroup = 32; // right-hand sides per GEMM pass (2 per thread column)

  __host__ __device__ constexpr int solve_min_waves(int R) { return R <= 3 ? 4 : (R == 4 ? 2 : 1); }

  // 1/d to ~1 ulp: v_rcp_f64 + two Newton steps.  Only the pivot thread runs it, and it sits
  // on the latency chain of every Gauss-Jordan step, so the ~30-instruction IEEE division
  // sequence is avoided.
  __device__ __forceinline__ double fast_rcp(double d)
  {
    double x = __builtin_amdgcn_rcp(d);
    double e = fma(-d, x, 1.0);
    x        = fma(x, e, x);
    e        = fma(-d, x, 1.0);
    return fma(x, e, x);
  }

  // 1/sqrt(x) to ~1 ulp: v_rsq_f64 + two Newton steps (the Jacobi rotation's dependent chain
  // otherwise carries two IEEE sqrt and three IEEE divisions, ~250 instructions)
  __device__ __forceinline__ double fast_rsqrt(double x)
  {
    double y = __builtin_amdgcn_rsq(x);
    double h = 0.5 * x;
    y        = y * fma(-h * y, y, 1.5);
    return y * fma(-h * y, y, 1.5);
  }


  __host__ __device__ constexpr int ws_min_waves(int T) { return T <= 5 ? 4 : (T <= 7 ? 2 : 1); }

  // Z tile (16 x 16) = Vs[16 ti .., :] * Rb[:, 16 tj ..] on the fp64 matrix pipe
  // (v_mfma_f64_16x16x4_f64: lane l feeds A[l&15][l>>4], B[l>>4][l&15]; D[(l>>4)+4r][l&15], r<4).
  // Two 8-byte LDS reads per 1024 FMAs instead of 80 bytes per 12 FMAs of the VALU tile, and
  // no VALU issue slots: the helper waves stop competing with the Gauss-Jordan waves.
  typedef double double4_t __attribute__((ext_vector_type(4)));
  __device__ __forceinline__ double4_t gemm_mfma_tile(const double *__restrict__ Vs, int ldv,
                                                      const double *__restrict__ Rb, int ncs, int k4,
                                                      int ti, int tj, int lane)
  {
    double4_t     acc = {0.0, 0.0, 0.0, 0.0};
    const double *ap  = Vs + (16 * ti + (lane & 15)) * ldv + (lane >> 4);
    const double *bp  = Rb + (lane >> 4) * ncs + 16 * tj + (lane & 15);
    for (int k = 0; k < k4; k += 4)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[k], bp[k * ncs], acc, 0, 0, 0);
    return acc;
  }

  // all-reduce inside a 16-lane row with DPP rotations (row_ror:8,4,2,1): ~4 VALU steps
  // instead of four LDS-routed shuffles on the dependent chain of every Jacobi rotation
  template <int CTRL>
  __device__ __forceinline__ double dpp_rot(double v)
  {
    union
    {
      double d;
      int    i[2];
    } in, out;
    in.d     = v;
    out.i[0] = __builtin_amdgcn_mov_dpp(in.i[0], CTRL, 0xf, 0xf, false);
    out.i[1] = __builtin_amdgcn_mov_dpp(in.i[1], CTRL, 0xf, 0xf, false);
    return out.d;
  }
  __device__ __forceinline__ double group16_sum(double v)
  {
    v += dpp_rot<0x128>(v); // row_ror:8
    v += dpp_rot<0x124>(v); // row_ror:4
    v += dpp_rot<0x122>(v); // row_ror:2
    v += dpp_rot<0x121>(v); // row_ror:1
    return v;
  }

  // id-99 boundary nodes in ascending node order (LODtools.h:360-371)
  __device__ __forceinline__ void boundary_node(const SlodPatchDesc &d, int bi, int &ix, int &iy)
  {
    const int l99 = !(d.flags & 1), r99 = !(d.flags & 2), b99 = !(d.flags & 4), t99 = !(d.flags & 8);
    const int side = l99 + r99;
    const int cb   = b99 ? d.nx + 1 : side;
    if (bi < cb)
      {
        iy = 0;
        ix = b99 ? bi : ((l99 && bi == 0) ? 0 : d.nx);
        return;
      }
    bi -= cb;
    const int cm = side * (d.ny - 1);
    if (bi < cm)
      {
        iy            = 1 + bi / side;
        const int wch = bi - (iy - 1) * side;
        ix            = (l99 && wch == 0) ? 0 : d.nx;
        return;
      }
    bi -= cm;
    iy = d.ny;
    ix = t99 ? bi : ((l99 && bi == 0) ? 0 : d.nx);
  }
} // namespace

#endif
