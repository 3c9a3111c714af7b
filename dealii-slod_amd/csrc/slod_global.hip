// Consumers and producers either side of the per-patch basis construction:
//   * the global LOD system  A_LOD = C^T (A C),  C^T f,  its solve and the fine-scale
//     reconstruction  (reference assemble_global_matrix LOD.cc:860-973, solve :976-1002, :1251);
//   * patch descriptors and coefficient sampling evaluated on the device
//     (create_patches / create_mesh_for_patch LOD.cc:122-244,770-858; problem_parameter::value
//     Diffusion.h:40-53 at the points of quadrature_fine, Diffusion.h:154).
// Everything is index arithmetic on the patch-lexicographic layout of include/slod.h: the overlap of
// two patches is a rectangle of global fine nodes.  These kernels are HBM/L2-bound gathers and
// reductions (no MFMA shape in them); one wave per patch pair keeps every reduction inside a wave.
#include "slod_host.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace
{
  // ---- index calculus shared by host and device (mirrors patch_geom() of slod_api.cpp) ----
  __host__ __device__ inline void grid_centre(const SlodGrid &G, uint32_t pid, int &cx, int &cy)
  {
    if (G.morton_bits < 0)
      {
        cx = (int)(pid % (uint32_t)G.N);
        cy = (int)(pid / (uint32_t)G.N);
        return;
      }
    cx = cy = 0;
    for (int b = 0; b < G.morton_bits; ++b)
      {
        cx |= (int)((pid >> (2 * b)) & 1u) << b;
        cy |= (int)((pid >> (2 * b + 1)) & 1u) << b;
      }
  }
  __host__ __device__ inline uint32_t grid_pid(const SlodGrid &G, int cx, int cy)
  {
    if (G.morton_bits < 0)
      return (uint32_t)(cx + G.N * cy);
    uint32_t p = 0;
    for (int b = 0; b < G.morton_bits; ++b)
      p |= ((uint32_t)((cx >> b) & 1) << (2 * b)) | ((uint32_t)((cy >> b) & 1) << (2 * b + 1));
    return p;
  }
  struct Extent
  {
    int x0, y0, mx, my; // coarse cells
  };
  __host__ __device__ inline Extent grid_extent(const SlodGrid &G, int cx, int cy)
  {
    Extent    e;
    const int l = G.oversampling;
    e.x0        = cx - l > 0 ? cx - l : 0;
    e.y0        = cy - l > 0 ? cy - l : 0;
    const int x1 = cx + l < G.N - 1 ? cx + l : G.N - 1, y1 = cy + l < G.N - 1 ? cy + l : G.N - 1;
    e.mx         = x1 - e.x0 + 1;
    e.my         = y1 - e.y0 + 1;
    return e;
  }

  // ---------------------------------------------------------------------------------
  // A_LOD block rows.  Block = one row patch p, wave w = the candidate neighbours j = w, w+4, ...
  // (offsets of the centre cell in [-(2l+1), 2l+1]^2: patches further apart share no node).
  // ---------------------------------------------------------------------------------
  __global__ __launch_bounds__(256) void k_lod_matrix(const SlodGrid G, const uint32_t *rows, const double *basis,
                                                     const double *premult, size_t stride, double *values,
                                                     uint32_t *cols)
  {
    const int      lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int      s = G.spacedim, n = G.n_sub, span = 4 * G.oversampling + 3, cap = span * span;
    const uint32_t p = rows[blockIdx.x];
    int            pcx, pcy;
    grid_centre(G, p, pcx, pcy);
    const Extent  pe = grid_extent(G, pcx, pcy);
    const int     pnx = pe.mx * n + 1, pny = pe.my * n + 1, pnf = s * pnx * pny;
    const double *phi = basis + (size_t)p * stride;
    for (int j = wave; j < cap; j += 4)
      {
        const int    qcx = pcx + j % span - (span / 2), qcy = pcy + j / span - (span / 2);
        const size_t out = (size_t)blockIdx.x * cap + j;
        if (qcx < 0 || qcx >= G.N || qcy < 0 || qcy >= G.N)
          {
            if (lane == 0)
              cols[out] = 0xffffffffu;
            if (lane < s * s)
              values[out * s * s + lane] = 0.0;
            continue;
          }
        const uint32_t q  = grid_pid(G, qcx, qcy);
        const Extent   qe = grid_extent(G, qcx, qcy);
        const int      qnx = qe.mx * n + 1, qny = qe.my * n + 1, qnf = s * qnx * qny;
        // overlap in global fine-node coordinates (inclusive)
        const int xa = max(pe.x0, qe.x0) * n, xb = min(pe.x0 + pe.mx, qe.x0 + qe.mx) * n;
        const int ya = max(pe.y0, qe.y0) * n, yb = min(pe.y0 + pe.my, qe.y0 + qe.my) * n;
        const int w = xb - xa + 1, hgt = yb - ya + 1;
        double    acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
        if (w > 0 && hgt > 0)
          {
            const double *psi = premult + (size_t)q * stride;
            for (int idx = lane; idx < w * hgt; idx += 64)
              {
                const int iy = idx / w, ix = idx - iy * w;
                const int np = (xa + ix - pe.x0 * n) + (ya + iy - pe.y0 * n) * pnx;
                const int nq = (xa + ix - qe.x0 * n) + (ya + iy - qe.y0 * n) * qnx;
                for (int c = 0; c < s; ++c)
                  for (int d = 0; d < s; ++d)
                    {
                      const double ph = phi[(size_t)d * pnf + s * np + c];
                      for (int e = 0; e < s; ++e)
                        acc[d][e] = fma(ph, psi[(size_t)e * qnf + s * nq + c], acc[d][e]);
                    }
              }
          }
        for (int d = 0; d < s; ++d)
          for (int e = 0; e < s; ++e)
            {
              double v = acc[d][e];
              for (int off = 32; off > 0; off >>= 1)
                v += __shfl_xor(v, off, 64);
              if (lane == 0)
                values[out * s * s + d * s + e] = v;
            }
        if (lane == 0)
          cols[out] = (w > 0 && hgt > 0) ? q : 0xffffffffu;
      }
  }

  // C^T f for the row patches: block = one patch, all its nodes
  __global__ __launch_bounds__(256) void k_lod_rhs(const SlodGrid G, const uint32_t *rows, const double *basis,
                                                  size_t stride, const double *frhs, double *out)
  {
    __shared__ double red[4][2];
    const int         s = G.spacedim, n = G.n_sub, NEp = G.N * n + 1;
    const uint32_t    p = rows[blockIdx.x];
    int               pcx, pcy;
    grid_centre(G, p, pcx, pcy);
    const Extent  pe = grid_extent(G, pcx, pcy);
    const int     pnx = pe.mx * n + 1, pny = pe.my * n + 1, pnf = s * pnx * pny;
    const double *phi = basis + (size_t)p * stride;
    double        acc[2] = {0.0, 0.0};
    for (int node = threadIdx.x; node < pnx * pny; node += 256)
      {
        const int iy = node / pnx, ix = node - iy * pnx;
        const int gn = (pe.x0 * n + ix) + (pe.y0 * n + iy) * NEp;
        for (int c = 0; c < s; ++c)
          {
            const double f = frhs[(size_t)gn * s + c];
            for (int d = 0; d < s; ++d)
              acc[d] = fma(phi[(size_t)d * pnf + s * node + c], f, acc[d]);
          }
      }
    for (int d = 0; d < s; ++d)
      {
        double v = acc[d];
        for (int off = 32; off > 0; off >>= 1)
          v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0)
          red[threadIdx.x >> 6][d] = v;
      }
    __syncthreads();
    if (threadIdx.x < s)
      out[(size_t)blockIdx.x * s + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] +
                                                  red[3][threadIdx.x];
  }

  // u_fine = C u_H: one thread per global fine node, gather over the patches that contain it
  __global__ __launch_bounds__(256) void k_lod_reconstruct(const SlodGrid G, const double *basis, size_t stride,
                                                          const double *u, double *fine)
  {
    const int s = G.spacedim, n = G.n_sub, NEp = G.N * n + 1, l = G.oversampling;
    const int gn = blockIdx.x * 256 + threadIdx.x;
    if (gn >= NEp * NEp)
      return;
    const int X = gn % NEp, Y = gn / NEp;
    // cells whose closure contains the node, widened by the oversampling
    const int cxl = max((X + n - 1) / n - 1 - l, 0), cxh = min(X / n + l, G.N - 1);
    const int cyl = max((Y + n - 1) / n - 1 - l, 0), cyh = min(Y / n + l, G.N - 1);
    double    acc[2] = {0.0, 0.0};
    for (int cy = cyl; cy <= cyh; ++cy)
      for (int cx = cxl; cx <= cxh; ++cx)
        {
          const Extent e = grid_extent(G, cx, cy);
          const int    ix = X - e.x0 * n, iy = Y - e.y0 * n;
          if (ix < 0 || ix > e.mx * n || iy < 0 || iy > e.my * n)
            continue;
          const uint32_t p   = grid_pid(G, cx, cy);
          const int      pnx = e.mx * n + 1, pnf = s * pnx * (e.my * n + 1);
          const double  *phi = basis + (size_t)p * stride;
          for (int d = 0; d < s; ++d)
            {
              const double ud = u[(size_t)p * s + d];
              for (int c = 0; c < s; ++c)
                acc[c] = fma(phi[(size_t)d * pnf + s * (ix + iy * pnx) + c], ud, acc[c]);
            }
        }
    for (int c = 0; c < s; ++c)
      fine[(size_t)gn * s + c] = acc[c];
  }

  // ---- Jacobi-preconditioned CG on the block rows (device scalars: no host round trip per step)
  struct CgScalars
  {
    double rz, pAp, rz_new, rr, rhs2;
  };
  __global__ void k_cg_spmv_dot(int nrow, int s, int cap, const double *values, const uint32_t *cols, const double *x,
                                double *y, CgScalars *sc)
  {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double    part = 0.0;
    if (i < nrow)
      {
        const int p = i / s, d = i - p * s;
        double    acc = 0.0;
        for (int j = 0; j < cap; ++j)
          {
            const uint32_t q = cols[(size_t)p * cap + j];
            if (q == 0xffffffffu)
              continue;
            for (int e = 0; e < s; ++e)
              acc = fma(values[((size_t)p * cap + j) * s * s + d * s + e], x[(size_t)q * s + e], acc);
          }
        y[i] = acc;
        part = acc * x[i];
      }
    for (int off = 32; off > 0; off >>= 1)
      part += __shfl_xor(part, off, 64);
    if ((threadIdx.x & 63) == 0 && part != 0.0)
      atomicAdd(&sc->pAp, part);
  }
  __global__ void k_cg_init(int nrow, int s, int cap, const double *values, const uint32_t *cols, const double *rhs,
                            double *x, double *r, double *z, double *pv, double *dinv, CgScalars *sc)
  {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double    a = 0.0, b = 0.0;
    if (i < nrow)
      {
        const int p = i / s, d = i - p * s;
        double    diag = 1.0;
        for (int j = 0; j < cap; ++j)
          if (cols[(size_t)p * cap + j] == (uint32_t)p)
            diag = values[((size_t)p * cap + j) * s * s + d * s + d];
        dinv[i] = diag != 0.0 ? 1.0 / diag : 1.0;
        x[i]    = 0.0;
        r[i]    = rhs[i];
        z[i]    = dinv[i] * rhs[i];
        pv[i]   = z[i];
        a       = r[i] * z[i];
        b       = r[i] * r[i];
      }
    for (int off = 32; off > 0; off >>= 1)
      {
        a += __shfl_xor(a, off, 64);
        b += __shfl_xor(b, off, 64);
      }
    if ((threadIdx.x & 63) == 0)
      {
        atomicAdd(&sc->rz, a);
        atomicAdd(&sc->rhs2, b);
        atomicAdd(&sc->rr, b);
      }
  }
  __global__ void k_cg_update_xr(int nrow, const double *pv, const double *Ap, const double *dinv, double *x, double *r,
                                 double *z, CgScalars *sc)
  {
    const int    i = blockIdx.x * 256 + threadIdx.x;
    const double alpha = sc->pAp != 0.0 ? sc->rz / sc->pAp : 0.0;
    double       a = 0.0, b = 0.0;
    if (i < nrow)
      {
        x[i] = fma(alpha, pv[i], x[i]);
        r[i] = fma(-alpha, Ap[i], r[i]);
        z[i] = dinv[i] * r[i];
        a    = r[i] * z[i];
        b    = r[i] * r[i];
      }
    for (int off = 32; off > 0; off >>= 1)
      {
        a += __shfl_xor(a, off, 64);
        b += __shfl_xor(b, off, 64);
      }
    if ((threadIdx.x & 63) == 0)
      {
        atomicAdd(&sc->rz_new, a);
        atomicAdd(&sc->rr, b);
      }
  }
  __global__ void k_cg_update_p(int nrow, const double *z, double *pv, const CgScalars *sc)
  {
    const int    i = blockIdx.x * 256 + threadIdx.x;
    const double beta = sc->rz != 0.0 ? sc->rz_new / sc->rz : 0.0;
    if (i < nrow)
      pv[i] = fma(beta, pv[i], z[i]);
  }
  __global__ void k_cg_rotate(CgScalars *sc)
  {
    sc->rz     = sc->rz_new;
    sc->rz_new = 0.0;
    sc->pAp    = 0.0;
    sc->rr     = 0.0;
  }

  // ---- fine FEM reference problem on the global fine grid (assemble_and_solve_fem_problem,
  // LOD.cc:1004-1094): load vector of assemble_stiffness (Diffusion.h:149-193) and a matrix-free
  // Jacobi-CG on the 9-point block stencil planes of k_assemble (whole domain = one "patch").
  // Dirichlet nodes (every side of the domain, LOD.cc:1021) are identity rows with value 0.
  __global__ void k_fem_rhs(int NE, int s, double h2q, const double *f_qp, double *rhs)
  {
    const int np = NE + 1, node = blockIdx.x * 256 + threadIdx.x;
    if (node >= np * np)
      return;
    const int  ix = node % np, iy = node / np;
    const bool bnd = ix == 0 || iy == 0 || ix == NE || iy == NE;
    for (int c = 0; c < s; ++c)
      {
        double acc = 0.0;
        if (!bnd)
          for (int ay = 0; ay < 2; ++ay)
            for (int ax = 0; ax < 2; ++ax)
              {
                const int    ex = ix - ax, ey = iy - ay; // element that has this node as its corner (ax, ay)
                const size_t ge = ((size_t)ey * NE + ex) * 4;
                for (int q = 0; q < 4; ++q)
                  {
                    constexpr double g0 = 0.21132486540518711775, g1 = 0.78867513459481288225; // (1 -+ 1/sqrt 3)/2
                    const double     xi = (q & 1) ? g1 : g0, eta = (q & 2) ? g1 : g0;
                    const double     N  = (ax ? xi : 1.0 - xi) * (ay ? eta : 1.0 - eta);
                    acc += N * (f_qp ? f_qp[(size_t)c * NE * NE * 4 + ge + q] : 1.0) * h2q;
                  }
              }
        rhs[(size_t)node * s + c] = acc;
      }
  }
  // y = A x on the interior nodes (x, y: [(NE+1)^2][s]); boundary rows: y = x.  Also accumulates
  // x.y into sc->pAp.  st: planes [(dir*s + a)*s + b][nn], dir = (dy+1)*3 + dx+1.
  __global__ void k_fem_spmv_dot(int NE, int s, const double *st, const double *x, double *y, CgScalars *sc)
  {
    const int    np = NE + 1, nn = np * np, node = blockIdx.x * 256 + threadIdx.x;
    double       part = 0.0;
    if (node < nn)
      {
        const int  ix = node % np, iy = node / np;
        const bool bnd = ix == 0 || iy == 0 || ix == NE || iy == NE;
        for (int a = 0; a < s; ++a)
          {
            double acc = 0.0;
            if (bnd)
              acc = x[(size_t)node * s + a];
            else
              for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx)
                  {
                    const int jx = ix + dx, jy = iy + dy;
                    if (jx == 0 || jy == 0 || jx == NE || jy == NE)
                      continue; // constrained neighbour: value 0
                    const int dir = (dy + 1) * 3 + dx + 1, nb = jx + jy * np;
                    for (int b = 0; b < s; ++b)
                      acc = fma(st[(size_t)((dir * s + a) * s + b) * nn + node], x[(size_t)nb * s + b], acc);
                  }
            y[(size_t)node * s + a] = acc;
            part += acc * x[(size_t)node * s + a];
          }
      }
    for (int off = 32; off > 0; off >>= 1)
      part += __shfl_xor(part, off, 64);
    if ((threadIdx.x & 63) == 0 && part != 0.0)
      atomicAdd(&sc->pAp, part);
  }
  __global__ void k_fem_init(int NE, int s, const double *st, const double *rhs, double *x, double *r, double *z, double *pv,
                             double *dinv, CgScalars *sc)
  {
    const int np = NE + 1, nn = np * np, node = blockIdx.x * 256 + threadIdx.x;
    double    a = 0.0, b = 0.0;
    if (node < nn)
      {
        const int  ix = node % np, iy = node / np;
        const bool bnd = ix == 0 || iy == 0 || ix == NE || iy == NE;
        for (int c = 0; c < s; ++c)
          {
            const size_t i    = (size_t)node * s + c;
            const double diag = bnd ? 1.0 : st[(size_t)((4 * s + c) * s + c) * nn + node];
            const double f    = bnd ? 0.0 : rhs[i];
            dinv[i]           = diag != 0.0 ? 1.0 / diag : 1.0;
            x[i]              = 0.0;
            r[i]              = f;
            z[i]              = dinv[i] * f;
            pv[i]             = z[i];
            a += f * z[i];
            b += f * f;
          }
      }
    for (int off = 32; off > 0; off >>= 1)
      {
        a += __shfl_xor(a, off, 64);
        b += __shfl_xor(b, off, 64);
      }
    if ((threadIdx.x & 63) == 0)
      {
        atomicAdd(&sc->rz, a);
        atomicAdd(&sc->rhs2, b);
        atomicAdd(&sc->rr, b);
      }
  }

  // ---- geometric multigrid V-cycle on the 9-point block stencil planes: the preconditioner of the
  // fine FEM reference solve (the reference uses CG + AMG, LOD.cc:1070-1075).  Levels halve the grid
  // while the number of elements per side is even; bilinear interpolation P, restriction P^T, Galerkin
  // coarse operators P^T A P (again 9-point stencils), damped-Jacobi smoothing with the same number
  // of sweeps before and after the coarse correction (a symmetric positive definite preconditioner).
  // Dirichlet nodes (all four sides) carry 0 on every level and are no unknowns.
  __device__ __forceinline__ double mg_w(int f, int c) // 1-D bilinear weight of coarse node c at fine node f
  {
    const int dlt = f - 2 * c;
    return dlt == 0 ? 1.0 : ((dlt == 1 || dlt == -1) ? 0.5 : 0.0);
  }
  // coarse planes from fine planes: one thread per coarse node
  __global__ void k_mg_galerkin(int Nf, int s, const double *stf, double *stc)
  {
    const int Nc = Nf / 2, npc = Nc + 1, nnc = npc * npc, npf = Nf + 1, nnf = npf * npf;
    const int node = blockIdx.x * 256 + threadIdx.x;
    if (node >= nnc)
      return;
    const int X = node % npc, Y = node / npc;
    double    acc[9][2][2];
    for (int q = 0; q < 9; ++q)
      for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
          acc[q][a][b] = 0.0;
    const bool bnd = X == 0 || Y == 0 || X == Nc || Y == Nc;
    if (!bnd)
      for (int ay = -1; ay <= 1; ++ay)
        for (int ax = -1; ax <= 1; ++ax)
          {
            const int ix = 2 * X + ax, iy = 2 * Y + ay; // fine node in the support of the coarse hat (interior)
            if (ix <= 0 || iy <= 0 || ix >= Nf || iy >= Nf)
              continue;
            const double wi = mg_w(ix, X) * mg_w(iy, Y);
            const int    fn = ix + iy * npf;
            for (int dy = -1; dy <= 1; ++dy)
              for (int dx = -1; dx <= 1; ++dx)
                {
                  const int jx = ix + dx, jy = iy + dy;
                  if (jx <= 0 || jy <= 0 || jx >= Nf || jy >= Nf)
                    continue; // constrained fine neighbour
                  const int dir = (dy + 1) * 3 + dx + 1;
                  for (int DY = -1; DY <= 1; ++DY)
                    for (int DX = -1; DX <= 1; ++DX)
                      {
                        const double wj = mg_w(jx, X + DX) * mg_w(jy, Y + DY);
                        if (wj == 0.0)
                          continue;
                        const int q = (DY + 1) * 3 + DX + 1;
                        for (int a = 0; a < s; ++a)
                          for (int b = 0; b < s; ++b)
                            acc[q][a][b] = fma(wi * wj, stf[(size_t)((dir * s + a) * s + b) * nnf + fn], acc[q][a][b]);
                      }
                }
          }
    for (int q = 0; q < 9; ++q)
      for (int a = 0; a < s; ++a)
        for (int b = 0; b < s; ++b)
          stc[(size_t)((q * s + a) * s + b) * nnc + node] = acc[q][a][b];
  }
  // (A x)(node, a) on the interior, constrained neighbours skipped
  __device__ __forceinline__ double mg_apply(int N, int s, const double *st, const double *x, int ix, int iy, int a)
  {
    const int np = N + 1, nn = np * np, node = ix + iy * np;
    double    acc = 0.0;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx)
        {
          const int jx = ix + dx, jy = iy + dy;
          if (jx == 0 || jy == 0 || jx == N || jy == N)
            continue;
          const int dir = (dy + 1) * 3 + dx + 1, nb = jx + jy * np;
          for (int b = 0; b < s; ++b)
            acc = fma(st[(size_t)((dir * s + a) * s + b) * nn + node], x[(size_t)nb * s + b], acc);
        }
    return acc;
  }
  // xo = xi + omega D^-1 (b - A xi)   (zero_in: xi = 0)
  __global__ void k_mg_smooth(int N, int s, const double *st, const double *b, const double *xi, double *xo, double omega,
                              int zero_in)
  {
    const int np = N + 1, nn = np * np, node = blockIdx.x * 256 + threadIdx.x;
    if (node >= nn)
      return;
    const int  ix = node % np, iy = node / np;
    const bool bnd = ix == 0 || iy == 0 || ix == N || iy == N;
    double     r[2] = {0.0, 0.0}, v[2] = {0.0, 0.0};
    if (!bnd)
      {
        for (int a = 0; a < s; ++a)
          r[a] = b[(size_t)node * s + a] - (zero_in ? 0.0 : mg_apply(N, s, st, xi, ix, iy, a));
        // block Jacobi: the s x s diagonal block of the node (vector problems: point Jacobi stalls where lambda >> mu)
        const double d00 = st[(size_t)((4 * s + 0) * s + 0) * nn + node];
        if (s == 1)
          v[0] = r[0] / d00;
        else
          {
            const double d01 = st[(size_t)((4 * s + 0) * s + 1) * nn + node], d10 = st[(size_t)((4 * s + 1) * s + 0) * nn + node],
                         d11 = st[(size_t)((4 * s + 1) * s + 1) * nn + node];
            const double det = d00 * d11 - d01 * d10;
            v[0]             = (d11 * r[0] - d01 * r[1]) / det;
            v[1]             = (d00 * r[1] - d10 * r[0]) / det;
          }
      }
    for (int a = 0; a < s; ++a)
      {
        const size_t i = (size_t)node * s + a;
        xo[i]          = bnd ? 0.0 : (zero_in ? 0.0 : xi[i]) + omega * v[a];
      }
  }
  // bc = P^T (b - A x): one thread per coarse node gathers its 3 x 3 fine residuals
  __global__ void k_mg_restrict(int Nf, int s, const double *st, const double *b, const double *x, double *bc)
  {
    const int Nc = Nf / 2, npc = Nc + 1, nnc = npc * npc, npf = Nf + 1;
    const int node = blockIdx.x * 256 + threadIdx.x;
    if (node >= nnc)
      return;
    const int  X = node % npc, Y = node / npc;
    const bool bnd = X == 0 || Y == 0 || X == Nc || Y == Nc;
    for (int a = 0; a < s; ++a)
      {
        double acc = 0.0;
        if (!bnd)
          for (int ay = -1; ay <= 1; ++ay)
            for (int ax = -1; ax <= 1; ++ax)
              {
                const int ix = 2 * X + ax, iy = 2 * Y + ay;
                if (ix <= 0 || iy <= 0 || ix >= Nf || iy >= Nf)
                  continue;
                const double r = b[(size_t)(ix + iy * npf) * s + a] - mg_apply(Nf, s, st, x, ix, iy, a);
                acc            = fma(mg_w(ix, X) * mg_w(iy, Y), r, acc);
              }
        bc[(size_t)node * s + a] = acc;
      }
  }
  // x += P xc
  __global__ void k_mg_prolong_add(int Nf, int s, const double *xc, double *x)
  {
    const int Nc = Nf / 2, npc = Nc + 1, npf = Nf + 1, nnf = npf * npf;
    const int node = blockIdx.x * 256 + threadIdx.x;
    if (node >= nnf)
      return;
    const int ix = node % npf, iy = node / npf;
    if (ix == 0 || iy == 0 || ix == Nf || iy == Nf)
      return;
    for (int a = 0; a < s; ++a)
      {
        double acc = 0.0;
        for (int Y = iy / 2; Y <= (iy + 1) / 2; ++Y)
          for (int X = ix / 2; X <= (ix + 1) / 2; ++X)
            acc = fma(mg_w(ix, X) * mg_w(iy, Y), xc[(size_t)(X + Y * npc) * s + a], acc);
        x[(size_t)node * s + a] += acc;
      }
  }
  // CG pieces around a general preconditioner: x += alpha p, r -= alpha Ap, rr; then rz_new = r.z
  __global__ void k_pcg_update_xr(int nrow, const double *pv, const double *Ap, double *x, double *r, CgScalars *sc)
  {
    const int    i = blockIdx.x * 256 + threadIdx.x;
    const double alpha = sc->pAp != 0.0 ? sc->rz / sc->pAp : 0.0;
    double       b = 0.0;
    if (i < nrow)
      {
        x[i] = fma(alpha, pv[i], x[i]);
        r[i] = fma(-alpha, Ap[i], r[i]);
        b    = r[i] * r[i];
      }
    for (int off = 32; off > 0; off >>= 1)
      b += __shfl_xor(b, off, 64);
    if ((threadIdx.x & 63) == 0)
      atomicAdd(&sc->rr, b);
  }
  __global__ void k_pcg_dot_rz(int nrow, const double *r, const double *z, CgScalars *sc, int first)
  {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double    a = i < nrow ? r[i] * z[i] : 0.0;
    for (int off = 32; off > 0; off >>= 1)
      a += __shfl_xor(a, off, 64);
    if ((threadIdx.x & 63) == 0)
      atomicAdd(first ? &sc->rz : &sc->rz_new, a);
  }
  __global__ void k_pcg_init(int NE, int s, const double *rhs, double *x, double *r, CgScalars *sc)
  {
    const int np = NE + 1, nn = np * np, node = blockIdx.x * 256 + threadIdx.x;
    double    b = 0.0;
    if (node < nn)
      {
        const int  ix = node % np, iy = node / np;
        const bool bnd = ix == 0 || iy == 0 || ix == NE || iy == NE;
        for (int c = 0; c < s; ++c)
          {
            const size_t i = (size_t)node * s + c;
            const double f = bnd ? 0.0 : rhs[i];
            x[i]           = 0.0;
            r[i]           = f;
            b += f * f;
          }
      }
    for (int off = 32; off > 0; off >>= 1)
      b += __shfl_xor(b, off, 64);
    if ((threadIdx.x & 63) == 0)
      {
        atomicAdd(&sc->rhs2, b);
        atomicAdd(&sc->rr, b);
      }
  }
  __global__ void k_copy(int n, const double *src, double *dst)
  {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n)
      dst[i] = src[i];
  }

  // ---- inputs produced on the device ----
  // ---- the plan's patch descriptors, built on the device (create_patches + create_mesh_for_patch +
  //      the index-set sizes, LOD.cc:122-244,770-858; one thread per patch of the plan).  Also: the plan's
  //      maxima (atomicMax), the set of coefficient realisations it uses, an error flag for ids out of
  //      range, and the cost key of the balanced launch order.
  __global__ void k_make_desc(const SlodGrid G, SlodPlanBuild B)
  {
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= B.n)
      return;
    const uint32_t gid = B.gids[k];
    if ((uint64_t)gid >= (uint64_t)B.NP * (uint64_t)B.n_problems)
      {
        atomicOr(&B.acc->error, 1);
        return;
      }
    const uint32_t prob = gid / (uint32_t)B.NP, pid = gid % (uint32_t)B.NP;
    const int      n = G.n_sub, s = G.spacedim;
    int            cx, cy;
    grid_centre(G, pid, cx, cy);
    Extent e = grid_extent(G, cx, cy);
    const bool full = e.mx == 2 * G.oversampling + 1 && e.my == 2 * G.oversampling + 1;
    SlodPatchDesc d;
    memset(&d, 0, sizeof(d));
    d.ox = e.x0 * n;
    d.oy = e.y0 * n;
    if (B.reuse_full && full && B.first_full >= 0)
      {
        // quirk Q1 (LOD.cc:354-362,446-450): later full patches copy the first one's matrix
        int fx, fy;
        grid_centre(G, (uint32_t)B.first_full, fx, fy);
        const Extent f = grid_extent(G, fx, fy);
        d.ox           = f.x0 * n;
        d.oy           = f.y0 * n;
      }
    d.nx  = n * e.mx;
    d.ny  = n * e.my;
    d.mx  = e.mx;
    d.my  = e.my;
    d.ccx = cx - e.x0;
    d.ccy = cy - e.y0;
    const int sd0 = e.x0 == 0, sd1 = e.x0 + e.mx == G.N, sd2 = e.y0 == 0, sd3 = e.y0 + e.my == G.N; // LOD.cc:830-843
    d.flags       = sd0 | (sd1 << 1) | (sd2 << 2) | (sd3 << 3);
    const bool lod = !G.lod_stabilization || G.oversampling == 0 || e.mx * e.my == G.N * G.N; // LOD.cc:563-564
    if (lod)
      d.flags |= SLOD_F_LOD;
    if (d.nx > d.ny)
      {
        d.flags |= SLOD_F_TRANSPOSED;
        d.m = s * (d.ny - 1);
        d.L = d.nx - 1;
      }
    else
      {
        d.m = s * (d.nx - 1);
        d.L = d.ny - 1;
      }
    d.n_c = s * e.mx * e.my;
    // id-99 nodes, corners shared with an id-0 side included (LODtools.h:367-369)
    const int side = !sd0 + !sd1;
    int       nb   = (d.ny - 1) * side;
    nb += sd2 ? side : d.nx + 1;
    nb += sd3 ? side : d.nx + 1;
    d.n_b        = s * nb;
    d.prob       = (int32_t)prob;
    d.plan_index = (uint32_t)k;
    d.out_off    = B.offsets ? B.offsets[k] : (uint64_t)k * B.stride;
    B.desc[k]    = d;
    const int nn = (d.nx + 1) * (d.ny + 1);
    atomicMax(&B.acc->m_max, d.m);
    atomicMax(&B.acc->L_max, d.L);
    atomicMax(&B.acc->nc_max, d.n_c);
    atomicMax(&B.acc->nb_max, lod ? 0 : d.n_b);
    atomicMax(&B.acc->nn_max, nn);
    atomicMax(&B.acc->out_size, (unsigned long long)d.out_off + (unsigned long long)s * s * nn);
    B.prob_used[prob] = 1;
    // cost key of the launch order: canonical solve flops + selection stage (rim patches are cheaper)
    const double ni = (double)d.m * d.L, b = (double)d.m + s - 1;
    B.cost[k]       = ni * (b * b + 3 * b) + 4.0 * d.n_c * ni * b + 200.0 * d.n_b * d.n_c;
  }

  // Balanced launch order on the device: rank of every patch by (cost descending, plan index ascending)
  // -- a rank sort, O(n^2) coalesced reads, fine up to a few 10^4 patches -- then the snake over rows of
  // n_cu blocks (see slod_plan_create).
  __global__ void k_balance_order(const double *cost, const SlodPatchDesc *in, SlodPatchDesc *out, size_t n, int n_cu)
  {
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n)
      return;
    const double ck = cost[k];
    size_t       r  = 0;
    for (size_t j = 0; j < n; ++j)
      {
        const double cj = cost[j];
        r += (cj > ck || (cj == ck && j < k)) ? 1 : 0;
      }
    const size_t row = r / (size_t)n_cu, col = r % (size_t)n_cu;
    const size_t len = (size_t)n_cu < n - row * (size_t)n_cu ? (size_t)n_cu : n - row * (size_t)n_cu;
    out[row * (size_t)n_cu + ((row & 1) ? len - 1 - col : col)] = in[k];
  }

  __global__ void k_sample_coefficient(int NE, const double *vals, int r, double *coef)
  {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)NE * NE * 4)
      return;
    const int    q = (int)(i & 3);
    const size_t el = i >> 2;
    const int    ex = (int)(el % (size_t)NE), ey = (int)(el / (size_t)NE);
    const double g0 = 0.21132486540518711775, g1 = 0.78867513459481288225; // QGauss<1>(2)
    const double hf = 1.0 / (double)NE;
    const double x = (ex + ((q & 1) ? g1 : g0)) * hf, y = (ey + ((q & 2) ? g1 : g0)) * hf;
    const int    NC = 1 << r;
    const double eta = 1.0 / (double)NC;
    // Diffusion.h:47-51
    const int idx = (int)floor(x / eta) + NC * (int)floor(y / eta);
    coef[i]       = vals[idx];
  }
} // namespace

// slod_plan_create's device pass (declared in slod_host.h): descriptors, maxima, realisations in use,
// balanced order.  Everything stays on the device except the 40-byte summary and the used-problem map.
hipError_t slod_build_descriptors(const slod_handle *h, const uint32_t *gids, size_t n, const uint64_t *offsets, size_t stride,
                                  int n_cu, bool balance, SlodPatchDesc *d_desc, SlodPatchDesc *d_desc_bal, SlodPlanSummary *sum,
                                  std::vector<char> *prob_used)
{
  SlodPlanBuild B;
  memset(&B, 0, sizeof(B));
  uint32_t *d_gids = nullptr;
  uint64_t *d_offs = nullptr;
  hipError_t e = hipMalloc((void **)&d_gids, n * sizeof(uint32_t));
  if (e == hipSuccess && offsets)
    e = hipMalloc((void **)&d_offs, n * sizeof(uint64_t));
  if (e == hipSuccess)
    e = hipMalloc((void **)&B.cost, n * sizeof(double));
  if (e == hipSuccess)
    e = hipMalloc((void **)&B.acc, sizeof(SlodPlanSummary));
  if (e == hipSuccess)
    e = hipMalloc((void **)&B.prob_used, (size_t)h->cfg.n_problems);
  if (e == hipSuccess)
    e = hipMemsetAsync(B.acc, 0, sizeof(SlodPlanSummary), h->stream);
  if (e == hipSuccess)
    e = hipMemsetAsync(B.prob_used, 0, (size_t)h->cfg.n_problems, h->stream);
  if (e == hipSuccess)
    e = hipMemcpyAsync(d_gids, gids, n * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess && offsets)
    e = hipMemcpyAsync(d_offs, offsets, n * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess)
    {
      B.gids       = d_gids;
      B.offsets    = d_offs;
      B.n          = n;
      B.stride     = stride;
      B.NP         = h->NP;
      B.n_problems = h->cfg.n_problems;
      B.reuse_full = h->cfg.constant_coefficients;
      B.first_full = h->first_full;
      B.desc       = d_desc;
      hipLaunchKernelGGL(k_make_desc, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, slod_grid_of(h), B);
      e = hipGetLastError();
    }
  if (e == hipSuccess && balance && d_desc_bal)
    {
      hipLaunchKernelGGL(k_balance_order, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, B.cost, d_desc, d_desc_bal, n,
                         n_cu);
      e = hipGetLastError();
    }
  if (e == hipSuccess)
    e = hipMemcpyAsync(sum, B.acc, sizeof(SlodPlanSummary), hipMemcpyDeviceToHost, h->stream);
  prob_used->assign((size_t)h->cfg.n_problems, 0);
  if (e == hipSuccess)
    e = hipMemcpyAsync(prob_used->data(), B.prob_used, (size_t)h->cfg.n_problems, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess)
    e = hipStreamSynchronize(h->stream);
  for (void *q : {(void *)d_gids, (void *)d_offs, (void *)B.cost, (void *)B.acc, (void *)B.prob_used})
    if (q)
      (void)hipFree(q);
  return e;
}

#pragma GCC visibility push(default)
extern "C" {

int slod_lod_row_capacity(const slod_handle *h)
{
  if (!h)
    return SLOD_ERR_ARGUMENT;
  const int span = 4 * h->cfg.oversampling + 3;
  return span * span;
}

int slod_lod_pattern(const slod_handle *h, uint32_t patch_id, uint32_t *neighbours, size_t capacity)
{
  if (!h || !neighbours)
    return SLOD_ERR_ARGUMENT;
  if (patch_id >= (uint32_t)h->NP)
    return slod_fail(h, SLOD_ERR_ARGUMENT, "slod_lod_pattern: patch id out of range");
  const SlodGrid G = slod_grid_of(h);
  int            cx, cy;
  grid_centre(G, patch_id, cx, cy);
  const Extent          pe = grid_extent(G, cx, cy);
  std::vector<uint32_t> nb;
  const int             half = 2 * h->cfg.oversampling + 1;
  for (int dy = -half; dy <= half; ++dy)
    for (int dx = -half; dx <= half; ++dx)
      {
        const int qx = cx + dx, qy = cy + dy;
        if (qx < 0 || qx >= h->N || qy < 0 || qy >= h->N)
          continue;
        const Extent qe = grid_extent(G, qx, qy);
        if (std::max(pe.x0, qe.x0) > std::min(pe.x0 + pe.mx, qe.x0 + qe.mx) ||
            std::max(pe.y0, qe.y0) > std::min(pe.y0 + pe.my, qe.y0 + qe.my))
          continue;
        nb.push_back(grid_pid(G, qx, qy));
      }
  std::sort(nb.begin(), nb.end());
  if (capacity < nb.size())
    return slod_fail(h, SLOD_ERR_ARGUMENT, "slod_lod_pattern: buffer too small");
  std::copy(nb.begin(), nb.end(), neighbours);
  return (int)nb.size();
}

static int upload_rows(slod_handle *h, const uint32_t *rows, size_t n, uint32_t **d_rows, hipStream_t st)
{
  for (size_t k = 0; k < n; ++k)
    if (rows[k] >= (uint32_t)h->NP)
      return slod_fail(h, SLOD_ERR_ARGUMENT, "row patch id out of range");
  hipError_t e = hipMalloc((void **)d_rows, std::max<size_t>(n, 1) * sizeof(uint32_t));
  if (e == hipSuccess)
    e = hipMemcpyAsync(*d_rows, rows, n * sizeof(uint32_t), hipMemcpyHostToDevice, st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st); // rows is a caller-owned host array
  return e == hipSuccess ? SLOD_OK : slod_hip_fail(h, e, "row upload");
}

int slod_lod_matrix(slod_handle *h, const uint32_t *rows, size_t n_rows, const double *d_basis, const double *d_premult,
                    size_t stride, double *d_values, uint32_t *d_cols, void *hip_stream)
{
  if (!h || (n_rows && (!rows || !d_basis || !d_premult || !d_values || !d_cols)))
    return SLOD_ERR_ARGUMENT;
  if (n_rows == 0)
    return SLOD_OK;
  if (const int rc = slod_ensure_device(h))
    return rc;
  (void)hipSetDevice(h->cfg.device);
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : h->stream;
  uint32_t   *d_rows = nullptr;
  if (const int rc = upload_rows(h, rows, n_rows, &d_rows, st))
    {
      if (d_rows)
        (void)hipFree(d_rows);
      return rc;
    }
  hipLaunchKernelGGL(k_lod_matrix, dim3((unsigned)n_rows), dim3(256), 0, st, slod_grid_of(h), d_rows, d_basis, d_premult,
                     stride, d_values, d_cols);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess)
    e = hipStreamSynchronize(st); // d_rows is freed below
  (void)hipFree(d_rows);
  return e == hipSuccess ? SLOD_OK : slod_hip_fail(h, e, "slod_lod_matrix");
}

int slod_lod_rhs(slod_handle *h, const uint32_t *rows, size_t n_rows, const double *d_basis, size_t stride,
                 const double *d_fine_rhs, double *d_out, void *hip_stream)
{
  if (!h || (n_rows && (!rows || !d_basis || !d_fine_rhs || !d_out)))
    return SLOD_ERR_ARGUMENT;
  if (n_rows == 0)
    return SLOD_OK;
  if (const int rc = slod_ensure_device(h))
    return rc;
  (void)hipSetDevice(h->cfg.device);
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : h->stream;
  uint32_t   *d_rows = nullptr;
  if (const int rc = upload_rows(h, rows, n_rows, &d_rows, st))
    {
      if (d_rows)
        (void)hipFree(d_rows);
      return rc;
    }
  hipLaunchKernelGGL(k_lod_rhs, dim3((unsigned)n_rows), dim3(256), 0, st, slod_grid_of(h), d_rows, d_basis, stride,
                     d_fine_rhs, d_out);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  (void)hipFree(d_rows);
  return e == hipSuccess ? SLOD_OK : slod_hip_fail(h, e, "slod_lod_rhs");
}

int slod_lod_solve(slod_handle *h, const double *d_values, const uint32_t *d_cols, const double *d_rhs, double *d_u,
                   double rel_tol, int max_iterations, double *rel_residual)
{
  if (!h || !d_values || !d_cols || !d_rhs || !d_u || max_iterations < 0)
    return SLOD_ERR_ARGUMENT;
  if (const int rc = slod_ensure_device(h))
    return rc;
  (void)hipSetDevice(h->cfg.device);
  hipStream_t  st = h->stream;
  const int    s = h->cfg.spacedim, cap = slod_lod_row_capacity(h), nrow = h->NP * s;
  const int    nblk = (nrow + 255) / 256;
  double      *work = nullptr;
  CgScalars   *sc = nullptr, hs;
  hipError_t   e = hipMalloc((void **)&work, (size_t)5 * nrow * sizeof(double));
  if (e == hipSuccess)
    e = hipMalloc((void **)&sc, sizeof(CgScalars));
  if (e == hipSuccess)
    e = hipMemsetAsync(sc, 0, sizeof(CgScalars), st);
  int it = 0;
  if (e == hipSuccess)
    {
      double *r = work, *z = work + nrow, *pv = work + 2 * (size_t)nrow, *Ap = work + 3 * (size_t)nrow,
             *dinv = work + 4 * (size_t)nrow;
      hipLaunchKernelGGL(k_cg_init, dim3(nblk), dim3(256), 0, st, nrow, s, cap, d_values, d_cols, d_rhs, d_u, r, z, pv, dinv,
                         sc);
      e = hipMemcpyAsync(&hs, sc, sizeof(hs), hipMemcpyDeviceToHost, st);
      if (e == hipSuccess)
        e = hipStreamSynchronize(st);
      const double rhs2 = hs.rhs2;
      double       rr   = hs.rr;
      // rr of the initial residual was accumulated by k_cg_init; clear the per-iteration sums
      if (e == hipSuccess)
        {
          hs.rr = 0.0;
          hs.pAp = 0.0;
          hs.rz_new = 0.0;
          e = hipMemcpyAsync(sc, &hs, sizeof(hs), hipMemcpyHostToDevice, st);
        }
      while (e == hipSuccess && it < max_iterations && rhs2 > 0.0 && rr > rel_tol * rel_tol * rhs2)
        {
          // a few iterations per convergence check: the scalars stay on the device in between
          const int burst = std::min(8, max_iterations - it);
          for (int b = 0; b < burst; ++b)
            {
              hipLaunchKernelGGL(k_cg_spmv_dot, dim3(nblk), dim3(256), 0, st, nrow, s, cap, d_values, d_cols, pv, Ap, sc);
              hipLaunchKernelGGL(k_cg_update_xr, dim3(nblk), dim3(256), 0, st, nrow, pv, Ap, dinv, d_u, r, z, sc);
              hipLaunchKernelGGL(k_cg_update_p, dim3(nblk), dim3(256), 0, st, nrow, z, pv, sc);
              if (b + 1 < burst)
                hipLaunchKernelGGL(k_cg_rotate, dim3(1), dim3(1), 0, st, sc);
            }
          it += burst;
          e = hipMemcpyAsync(&hs, sc, sizeof(hs), hipMemcpyDeviceToHost, st);
          if (e == hipSuccess)
            e = hipStreamSynchronize(st);
          rr = hs.rr;
          if (e == hipSuccess)
            hipLaunchKernelGGL(k_cg_rotate, dim3(1), dim3(1), 0, st, sc);
        }
      if (e == hipSuccess)
        e = hipStreamSynchronize(st);
      if (rel_residual)
        *rel_residual = rhs2 > 0.0 ? std::sqrt(rr / rhs2) : 0.0;
    }
  if (work)
    (void)hipFree(work);
  if (sc)
    (void)hipFree(sc);
  if (e != hipSuccess)
    return slod_hip_fail(h, e, "slod_lod_solve");
  return it;
}

int slod_lod_reconstruct(slod_handle *h, const double *d_basis, size_t stride, const double *d_u, double *d_fine,
                         void *hip_stream)
{
  if (!h || !d_basis || !d_u || !d_fine)
    return SLOD_ERR_ARGUMENT;
  if (const int rc = slod_ensure_device(h))
    return rc;
  (void)hipSetDevice(h->cfg.device);
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : h->stream;
  const int   NEp = h->NE + 1;
  hipLaunchKernelGGL(k_lod_reconstruct, dim3((unsigned)((NEp * NEp + 255) / 256)), dim3(256), 0, st, slod_grid_of(h),
                     d_basis, stride, d_u, d_fine);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? SLOD_OK : slod_hip_fail(h, e, "slod_lod_reconstruct");
}

int slod_device_patch_layout(slod_handle *h, const uint32_t *patch_ids, size_t n, slod_patch_info *out)
{
  if (!h || (n && (!patch_ids || !out)))
    return SLOD_ERR_ARGUMENT;
  if (n == 0)
    return SLOD_OK;
  if (const int rc = slod_ensure_device(h))
    return rc;
  (void)hipSetDevice(h->cfg.device);
  // the SAME kernel that builds a plan's descriptors (k_make_desc), read back and reported as
  // slod_patch_info; ids of problem 0 (< num_patches): the id range check runs in the kernel
  SlodPatchDesc *d_desc = nullptr;
  hipError_t     e = hipMalloc((void **)&d_desc, n * sizeof(SlodPatchDesc));
  SlodPlanSummary   sum;
  std::vector<char> used;
  memset(&sum, 0, sizeof(sum));
  if (e == hipSuccess)
    e = slod_build_descriptors(h, patch_ids, n, nullptr, 0, 1, false, d_desc, nullptr, &sum, &used);
  std::vector<SlodPatchDesc> desc(n);
  if (e == hipSuccess && !sum.error)
    e = hipMemcpy(desc.data(), d_desc, n * sizeof(SlodPatchDesc), hipMemcpyDeviceToHost);
  if (d_desc)
    (void)hipFree(d_desc);
  if (e != hipSuccess)
    return slod_hip_fail(h, e, "slod_device_patch_layout");
  if (sum.error)
    return slod_fail(h, SLOD_ERR_ARGUMENT, "slod_device_patch_layout: patch id out of range");
  for (size_t k = 0; k < n; ++k)
    {
      if (desc[k].prob != 0)
        return slod_fail(h, SLOD_ERR_ARGUMENT, "slod_device_patch_layout: patch id out of range");
      slod_desc_to_info(h, desc[k], &out[k]);
    }
  return SLOD_OK;
}

int slod_sample_coefficient(slod_handle *h, uint32_t problem, int field, const double *d_vals, int r)
{
  if (!h || !d_vals)
    return SLOD_ERR_ARGUMENT;
  if (problem >= (uint32_t)h->cfg.n_problems || field < 0 || field >= h->cfg.spacedim || r < 0 || r > 14)
    return slod_fail(h, SLOD_ERR_ARGUMENT, "slod_sample_coefficient: problem/field/r out of range");
  if (const int rc = slod_ensure_device(h))
    return rc;
  (void)hipSetDevice(h->cfg.device);
  const size_t cnt = (size_t)h->NE * h->NE * 4;
  hipLaunchKernelGGL(k_sample_coefficient, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, h->NE, d_vals, r,
                     h->d_coef[field] + (size_t)problem * cnt);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess)
    e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess)
    return slod_hip_fail(h, e, "slod_sample_coefficient");
  h->coef_set[(size_t)problem * 2 + field] = 1;
  return SLOD_OK;
}

int slod_fem_rhs(slod_handle *h, const double *d_f_qp, double *d_fine_rhs, void *hip_stream)
{
  if (!h || !d_fine_rhs)
    return SLOD_ERR_ARGUMENT;
  if (const int rc = slod_ensure_device(h))
    return rc;
  (void)hipSetDevice(h->cfg.device);
  hipStream_t  st = hip_stream ? (hipStream_t)hip_stream : h->stream;
  const int    nn = (h->NE + 1) * (h->NE + 1);
  const double hf = 1.0 / h->NE;
  hipLaunchKernelGGL(k_fem_rhs, dim3((nn + 255) / 256), dim3(256), 0, st, h->NE, h->cfg.spacedim, hf * hf * 0.25, d_f_qp,
                     d_fine_rhs);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? SLOD_OK : slod_hip_fail(h, e, "slod_fem_rhs");
}

int slod_fem_solve(slod_handle *h, uint32_t problem, const double *d_fine_rhs, double *d_fine_u, double rel_tol,
                   int max_iterations, double *rel_residual)
{
  if (!h || !d_fine_rhs || !d_fine_u || max_iterations < 0)
    return SLOD_ERR_ARGUMENT;
  if (problem >= (uint32_t)h->cfg.n_problems)
    return slod_fail(h, SLOD_ERR_ARGUMENT, "slod_fem_solve: problem out of range");
  const int s = h->cfg.spacedim;
  for (int f = 0; f < s; ++f)
    if (!h->coef_set[(size_t)problem * 2 + f])
      return slod_fail(h, SLOD_ERR_ARGUMENT, "slod_fem_solve: coefficient not set");
  if (const int rc = slod_ensure_device(h))
    return rc;
  (void)hipSetDevice(h->cfg.device);
  hipStream_t   st = h->stream;
  const int     NE = h->NE, nn = (NE + 1) * (NE + 1), nblk = (nn + 255) / 256;
  const size_t  nrow = (size_t)nn * s;
  double       *planes = nullptr, *work = nullptr;
  SlodPatchDesc *d_desc = nullptr;
  CgScalars    *sc = nullptr, hs;
  hipError_t    e = hipMalloc((void **)&planes, (size_t)9 * s * s * nn * sizeof(double));
  if (e == hipSuccess)
    e = hipMalloc((void **)&work, 5 * nrow * sizeof(double));
  if (e == hipSuccess)
    e = hipMalloc((void **)&d_desc, sizeof(SlodPatchDesc));
  if (e == hipSuccess)
    e = hipMalloc((void **)&sc, sizeof(CgScalars));
  if (e == hipSuccess)
    e = hipMemsetAsync(sc, 0, sizeof(CgScalars), st);
  int it = 0;
  if (e == hipSuccess)
    {
      // the whole domain as one patch of k_assemble: NE x NE fine elements at the origin
      SlodPatchDesc d;
      std::memset(&d, 0, sizeof(d));
      d.nx   = NE;
      d.ny   = NE;
      d.prob = (int32_t)problem;
      e      = hipMemcpyAsync(d_desc, &d, sizeof(d), hipMemcpyHostToDevice, st);
      SlodKernelArgs a;
      std::memset(&a, 0, sizeof(a));
      a.desc        = d_desc;
      a.coef0       = h->d_coef[0];
      a.coef1       = h->d_coef[1];
      a.coef_stride = (size_t)NE * NE * 4;
      a.NE          = NE;
      a.n_sub       = h->cfg.n_subdivisions;
      a.st          = planes;
      a.st_stride   = (size_t)9 * s * s * nn;
      a.nn_max      = nn;
      if (e == hipSuccess)
        e = slod_launch_assemble(s, a, 1, st);
    }
  const char *pc_env = std::getenv("SLOD_FEM_PRECOND");
  // (scalar problems: with two independent high-contrast Lame fields the point/block-Jacobi smoothed
  // V-cycle is a worse preconditioner than plain Jacobi -- 4500 against 1376 iterations on the 65^2
  // grid -- so vector problems keep Jacobi unless SLOD_FEM_PRECOND=mg asks for it)
  const bool  want_mg = pc_env ? !strcmp(pc_env, "mg") : s == 1;
  const bool  use_mg = want_mg && NE >= 4 && NE % 2 == 0;
  // multigrid hierarchy (level 0 = the fine grid): planes, right-hand side, two iterates per level
  struct MgLevel
  {
    int     N;
    double *st, *b, *x, *y;
  };
  std::vector<MgLevel> lev;
  double              *mg_mem = nullptr;
  if (e == hipSuccess && use_mg)
    {
      size_t need = 0;
      for (int N = NE; ; N /= 2)
        {
          const size_t nnl = (size_t)(N + 1) * (N + 1);
          need += (N == NE ? 0 : (size_t)9 * s * s * nnl) + 3 * nnl * s;
          lev.push_back({N, nullptr, nullptr, nullptr, nullptr});
          if (N % 2 || N / 2 < 2)
            break;
        }
      e = hipMalloc((void **)&mg_mem, need * sizeof(double));
      double *q = mg_mem;
      for (size_t l = 0; l < lev.size() && e == hipSuccess; ++l)
        {
          const size_t nnl = (size_t)(lev[l].N + 1) * (lev[l].N + 1);
          lev[l].st = l == 0 ? planes : q;
          q += l == 0 ? 0 : (size_t)9 * s * s * nnl;
          lev[l].b = q;
          lev[l].x = q + nnl * s;
          lev[l].y = q + 2 * nnl * s;
          q += 3 * nnl * s;
          if (l > 0)
            hipLaunchKernelGGL(k_mg_galerkin, dim3((unsigned)((nnl + 255) / 256)), dim3(256), 0, st, lev[l - 1].N, s, lev[l - 1].st,
                               lev[l].st);
        }
      if (e == hipSuccess)
        e = hipGetLastError();
    }
  // z = V-cycle(r): V(2,2), damped Jacobi (omega 0.8); the coarsest level by a fixed, even number of sweeps
  auto vcycle = [&](const double *rin, double *zout) {
    const double omega = 0.8;
    const int    nu = 2;
    for (size_t l = 0; l < lev.size(); ++l)
      {
        const MgLevel &L = lev[l];
        const int      nnl = (L.N + 1) * (L.N + 1), nb = (nnl + 255) / 256;
        const double  *bl = l == 0 ? rin : L.b;
        const bool     last = l + 1 == lev.size();
        const int      sweeps = last ? (L.N <= 2 ? 2 : 40) : nu;
        double        *xi = L.x, *xo = L.y;
        for (int k = 0; k < sweeps; ++k)
          {
            hipLaunchKernelGGL(k_mg_smooth, dim3(nb), dim3(256), 0, st, L.N, s, L.st, bl, xi, xo, L.N <= 2 ? 1.0 : omega, k == 0 ? 1 : 0);
            std::swap(xi, xo);
          }
        // sweeps is even: the current iterate is back in L.x
        if (!last)
          {
            const int nnc = (L.N / 2 + 1) * (L.N / 2 + 1);
            hipLaunchKernelGGL(k_mg_restrict, dim3((nnc + 255) / 256), dim3(256), 0, st, L.N, s, L.st, bl, L.x, lev[l + 1].b);
          }
      }
    for (size_t l = lev.size() - 1; l-- > 0;)
      {
        const MgLevel &L = lev[l];
        const int      nnl = (L.N + 1) * (L.N + 1), nb = (nnl + 255) / 256;
        const double  *bl = l == 0 ? rin : L.b;
        hipLaunchKernelGGL(k_mg_prolong_add, dim3(nb), dim3(256), 0, st, L.N, s, lev[l + 1].x, L.x);
        double *xi = L.x, *xo = L.y;
        for (int k = 0; k < nu; ++k)
          {
            hipLaunchKernelGGL(k_mg_smooth, dim3(nb), dim3(256), 0, st, L.N, s, L.st, bl, xi, xo, omega, 0);
            std::swap(xi, xo);
          }
      }
    hipLaunchKernelGGL(k_copy, dim3((unsigned)((nrow + 255) / 256)), dim3(256), 0, st, (int)nrow, lev[0].x, zout);
  };
  if (e == hipSuccess && use_mg)
    {
      double   *r = work, *z = work + nrow, *pv = work + 2 * nrow, *Ap = work + 3 * nrow;
      const int nb1 = (int)((nrow + 255) / 256);
      hipLaunchKernelGGL(k_pcg_init, dim3(nblk), dim3(256), 0, st, NE, s, d_fine_rhs, d_fine_u, r, sc);
      vcycle(r, z);
      hipLaunchKernelGGL(k_pcg_dot_rz, dim3(nb1), dim3(256), 0, st, (int)nrow, r, z, sc, 1);
      hipLaunchKernelGGL(k_copy, dim3(nb1), dim3(256), 0, st, (int)nrow, z, pv);
      e = hipMemcpyAsync(&hs, sc, sizeof(hs), hipMemcpyDeviceToHost, st);
      if (e == hipSuccess)
        e = hipStreamSynchronize(st);
      const double rhs2 = hs.rhs2;
      double       rr   = hs.rr;
      if (e == hipSuccess)
        {
          hs.rr = 0.0;
          hs.pAp = 0.0;
          hs.rz_new = 0.0;
          e = hipMemcpyAsync(sc, &hs, sizeof(hs), hipMemcpyHostToDevice, st);
        }
      while (e == hipSuccess && it < max_iterations && rhs2 > 0.0 && rr > rel_tol * rel_tol * rhs2)
        {
          const int burst = std::min(4, max_iterations - it); // iterations per convergence check
          for (int b = 0; b < burst; ++b)
            {
              hipLaunchKernelGGL(k_fem_spmv_dot, dim3(nblk), dim3(256), 0, st, NE, s, planes, pv, Ap, sc);
              hipLaunchKernelGGL(k_pcg_update_xr, dim3(nb1), dim3(256), 0, st, (int)nrow, pv, Ap, d_fine_u, r, sc);
              vcycle(r, z);
              hipLaunchKernelGGL(k_pcg_dot_rz, dim3(nb1), dim3(256), 0, st, (int)nrow, r, z, sc, 0);
              hipLaunchKernelGGL(k_cg_update_p, dim3(nb1), dim3(256), 0, st, (int)nrow, z, pv, sc);
              if (b + 1 < burst)
                hipLaunchKernelGGL(k_cg_rotate, dim3(1), dim3(1), 0, st, sc);
            }
          it += burst;
          e = hipMemcpyAsync(&hs, sc, sizeof(hs), hipMemcpyDeviceToHost, st);
          if (e == hipSuccess)
            e = hipStreamSynchronize(st);
          rr = hs.rr;
          if (e == hipSuccess)
            hipLaunchKernelGGL(k_cg_rotate, dim3(1), dim3(1), 0, st, sc);
        }
      if (e == hipSuccess)
        e = hipStreamSynchronize(st);
      if (rel_residual)
        *rel_residual = rhs2 > 0.0 ? std::sqrt(rr / rhs2) : 0.0;
    }
  else
  if (e == hipSuccess)
    {
      double *r = work, *z = work + nrow, *pv = work + 2 * nrow, *Ap = work + 3 * nrow, *dinv = work + 4 * nrow;
      hipLaunchKernelGGL(k_fem_init, dim3(nblk), dim3(256), 0, st, NE, s, planes, d_fine_rhs, d_fine_u, r, z, pv, dinv, sc);
      e = hipMemcpyAsync(&hs, sc, sizeof(hs), hipMemcpyDeviceToHost, st);
      if (e == hipSuccess)
        e = hipStreamSynchronize(st);
      const double rhs2 = hs.rhs2;
      double       rr   = hs.rr;
      if (e == hipSuccess)
        {
          hs.rr = 0.0;
          hs.pAp = 0.0;
          hs.rz_new = 0.0;
          e = hipMemcpyAsync(sc, &hs, sizeof(hs), hipMemcpyHostToDevice, st);
        }
      const int nb1 = (int)((nrow + 255) / 256);
      while (e == hipSuccess && it < max_iterations && rhs2 > 0.0 && rr > rel_tol * rel_tol * rhs2)
        {
          // a burst of iterations per convergence check: the scalars stay on the device in between
          const int burst = std::min(32, max_iterations - it);
          for (int b = 0; b < burst; ++b)
            {
              hipLaunchKernelGGL(k_fem_spmv_dot, dim3(nblk), dim3(256), 0, st, NE, s, planes, pv, Ap, sc);
              hipLaunchKernelGGL(k_cg_update_xr, dim3(nb1), dim3(256), 0, st, (int)nrow, pv, Ap, dinv, d_fine_u, r, z, sc);
              hipLaunchKernelGGL(k_cg_update_p, dim3(nb1), dim3(256), 0, st, (int)nrow, z, pv, sc);
              if (b + 1 < burst)
                hipLaunchKernelGGL(k_cg_rotate, dim3(1), dim3(1), 0, st, sc);
            }
          it += burst;
          e = hipMemcpyAsync(&hs, sc, sizeof(hs), hipMemcpyDeviceToHost, st);
          if (e == hipSuccess)
            e = hipStreamSynchronize(st);
          rr = hs.rr;
          if (e == hipSuccess)
            hipLaunchKernelGGL(k_cg_rotate, dim3(1), dim3(1), 0, st, sc);
        }
      if (e == hipSuccess)
        e = hipStreamSynchronize(st);
      if (rel_residual)
        *rel_residual = rhs2 > 0.0 ? std::sqrt(rr / rhs2) : 0.0;
    }
  if (mg_mem)
    (void)hipFree(mg_mem);
  if (planes)
    (void)hipFree(planes);
  if (work)
    (void)hipFree(work);
  if (d_desc)
    (void)hipFree(d_desc);
  if (sc)
    (void)hipFree(sc);
  if (e != hipSuccess)
    return slod_hip_fail(h, e, "slod_fem_solve");
  return it;
}

} // extern "C"
#pragma GCC visibility pop
