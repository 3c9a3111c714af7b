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
#include "slod_device.h"

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

  // ---------------------------------------------------------------------------------
  // K2: constrained multi-RHS patch solve
  // ---------------------------------------------------------------------------------
  // Interior dofs are grouped by grid line (m dofs per line, L lines, lines along the
  // shorter patch side).  A_II is block tridiagonal: T_l on the diagonal, B_l between
  // line l and l+1 (both banded, half-bandwidth 2S-1).  Forward elimination
  //     S_l = T_l - B_{l-1}^T V_{l-1} B_{l-1},   V_l = S_l^{-1},
  //     Z_l = V_l (F_l - B_{l-1}^T Z_{l-1}),
  // backward substitution  X_l = Z_l - V_l B_l X_{l+1}.
  // V_l (m x m) is computed IN REGISTERS: a 16x16 thread grid holds an R x R strided tile
  // each (entry (ty+16a, tx+16b)); the symmetric Gauss-Jordan sweep needs only pivot row k,
  // which its 16 owner threads publish to a double-buffered LDS row => one barrier per
  // pivot.  V_l and Z_l go to the per-patch global workspace for the backward pass (they
  // do not fit the 160 KB LDS: 39 lines x 12 KB at the north-star size).
  template <int R, int S>
  __global__ __launch_bounds__(256) void k_solve(const SlodKernelArgs A)
  {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const SlodPatchDesc d = A.desc[blockIdx.x];
    constexpr int       W = 2 * S - 1, BW = 2 * W + 1, NB = 16 * R;
    const int           tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int           m = d.m, L = d.L, nc = d.n_c, n = A.n_sub;
    const int           mm = A.m_max, ldv = mm + 1, ncs = A.nc_max;
    const bool          tr  = (d.flags & SLOD_F_TRANSPOSED) != 0;
    const int           npx = d.nx + 1;

    double *Vs     = smem;               // [mm][ldv]  V of the current / previous line
    double *Rb     = Vs + mm * ldv;      // [mm][ncs]  right-hand side block
    double *Zp     = Rb + mm * ncs;      // [mm][ncs]  Z of the previous line / X of the next
    double *rowbuf = Zp + mm * ncs;      // [2][NB]    published pivot rows
    double *Tb     = rowbuf + 2 * NB;    // [mm][BW]   band of T_l
    double *Bp     = Tb + mm * BW;       // [mm][BW]   band of B_{l-1}
    double *Bn     = Bp + mm * BW;       // [mm][BW]   band of B_l

    const double *st    = A.st + (size_t)blockIdx.x * A.st_stride;
    double       *vg    = A.vinv + (size_t)blockIdx.x * A.v_stride;
    double       *xg    = A.xs + (size_t)blockIdx.x * A.x_stride;
    const size_t  vline = (size_t)mm * mm, xline = (size_t)mm * ncs;

    // ------------------------------ forward elimination ---------------------------
    for (int l = 0; l < L; ++l)
      {
        for (int idx = tid; idx < m * BW; idx += 256)
          {
            const int i = idx / BW, o = idx - i * BW - W;
            Tb[idx] = coupling<S>(st, A.nn_max, npx, tr, m, l, i, 0, o);
            Bn[idx] = (l + 1 < L) ? coupling<S>(st, A.nn_max, npx, tr, m, l, i, 1, o) : 0.0;
          }
        __syncthreads();

        // S_l in registers
        double a[R][R];
#pragma unroll
        for (int ra = 0; ra < R; ++ra)
#pragma unroll
          for (int rb = 0; rb < R; ++rb)
            {
              const int i = ty + 16 * ra, j = tx + 16 * rb;
              double    v = 0.0;
              if (i < m && j < m)
                {
                  const int o = j - i;
                  if (o >= -W && o <= W)
                    v = Tb[i * BW + o + W];
                  if (l > 0)
                    {
                      double acc = 0.0;
                      for (int e = -W; e <= W; ++e)
                        {
                          const int p = i + e;
                          if (p < 0 || p >= m)
                            continue;
                          const double bpi = Bp[p * BW + (W - e)];
                          double       inner = 0.0;
                          for (int f = -W; f <= W; ++f)
                            {
                              const int q = j + f;
                              if (q < 0 || q >= m)
                                continue;
                              inner += Vs[p * ldv + q] * Bp[q * BW + (W - f)];
                            }
                          acc += bpi * inner;
                        }
                      v -= acc;
                    }
                }
              a[ra][rb] = v;
            }

        // right-hand side block F_l - B_{l-1}^T Z_{l-1}; F = rows of P^T (LOD.cc:478-495)
        for (int idx = tid; idx < m * nc; idx += 256)
          {
            const int i = idx / nc, r = idx - i * nc;
            const int pos = i / S, comp = i - pos * S;
            const int ix = tr ? l + 1 : pos + 1, iy = tr ? pos + 1 : l + 1;
            double    v  = A.scale * pt_weight<S>(d, n, A.quirk, ix, iy, comp, r);
            if (l > 0)
              for (int e = -W; e <= W; ++e)
                {
                  const int p = i + e;
                  if (p >= 0 && p < m)
                    v -= Bp[p * BW + (W - e)] * Zp[p * ncs + r];
                }
            Rb[i * ncs + r] = v;
          }
        __syncthreads();

        // symmetric Gauss-Jordan sweep: a <- -S_l^{-1}
        for (int k = 0; k < m; ++k)
          {
            const int ka = k >> 4, kt = k & 15;
            double   *rbuf = rowbuf + (k & 1) * NB;
            if (ty == kt)
              {
#pragma unroll
                for (int ra = 0; ra < R; ++ra)
                  if (ra == ka)
                    {
#pragma unroll
                      for (int rb = 0; rb < R; ++rb)
                        rbuf[tx + 16 * rb] = a[ra][rb];
                    }
              }
            __syncthreads();
            const double piv = rbuf[k];
            if (tid == 0 && !(piv > 0.0))
              atomicOr(A.status, 1);
            const double p = 1.0 / piv;
            double       ri[R], rj[R];
#pragma unroll
            for (int ra = 0; ra < R; ++ra)
              ri[ra] = rbuf[ty + 16 * ra];
#pragma unroll
            for (int rb = 0; rb < R; ++rb)
              rj[rb] = rbuf[tx + 16 * rb];
#pragma unroll
            for (int ra = 0; ra < R; ++ra)
#pragma unroll
              for (int rb = 0; rb < R; ++rb)
                {
                  const bool   rowk = (ty == kt) && (ra == ka);
                  const bool   colk = (tx == kt) && (rb == ka);
                  const double t    = ri[ra] * rj[rb];
                  const double upd  = fma(-t, p, a[ra][rb]);
                  a[ra][rb] = rowk ? (colk ? -p : rj[rb] * p) : (colk ? ri[ra] * p : upd);
                }
          }

        // V_l = -a  -> LDS (next sandwich, GEMM) and global workspace (backward pass)
#pragma unroll
        for (int ra = 0; ra < R; ++ra)
#pragma unroll
          for (int rb = 0; rb < R; ++rb)
            {
              const int i = ty + 16 * ra, j = tx + 16 * rb;
              if (i < m && j < m)
                {
                  const double v  = -a[ra][rb];
                  Vs[i * ldv + j] = v;
                  vg[(size_t)l * vline + (size_t)i * mm + j] = v;
                }
            }
        __syncthreads();

        // Z_l = V_l R_l
        for (int r = tx; r < nc; r += 16)
          {
            double acc[R];
#pragma unroll
            for (int ra = 0; ra < R; ++ra)
              acc[ra] = 0.0;
            for (int k = 0; k < m; ++k)
              {
                const double rk = Rb[k * ncs + r];
#pragma unroll
                for (int ra = 0; ra < R; ++ra)
                  {
                    const int i = ty + 16 * ra;
                    acc[ra] += ((i < m) ? Vs[i * ldv + k] : 0.0) * rk;
                  }
              }
#pragma unroll
            for (int ra = 0; ra < R; ++ra)
              {
                const int i = ty + 16 * ra;
                if (i < m)
                  {
                    Zp[i * ncs + r] = acc[ra];
                    xg[(size_t)l * xline + (size_t)i * ncs + r] = acc[ra];
                  }
              }
          }
        double *t = Bp;
        Bp        = Bn;
        Bn        = t;
        // the barrier after the next band load orders Zp/Vs writes before their readers
      }
    __syncthreads();

    // ------------------------------ backward substitution -------------------------
    // Zp holds X_{L-1} = Z_{L-1}
    for (int l = L - 2; l >= 0; --l)
      {
        for (int idx = tid; idx < m * BW; idx += 256)
          {
            const int i = idx / BW, o = idx - i * BW - W;
            Bn[idx]     = coupling<S>(st, A.nn_max, npx, tr, m, l, i, 1, o);
          }
        for (int idx = tid; idx < m * m; idx += 256)
          {
            const int i = idx / m, j = idx - i * m;
            Vs[i * ldv + j] = vg[(size_t)l * vline + (size_t)i * mm + j];
          }
        __syncthreads();
        for (int idx = tid; idx < m * nc; idx += 256)
          {
            const int i = idx / nc, r = idx - i * nc;
            double    v = 0.0;
            for (int o = -W; o <= W; ++o)
              {
                const int p = i + o;
                if (p >= 0 && p < m)
                  v += Bn[i * BW + o + W] * Zp[p * ncs + r];
              }
            Rb[i * ncs + r] = v;
          }
        __syncthreads();
        for (int r = tx; r < nc; r += 16)
          {
            double acc[R];
#pragma unroll
            for (int ra = 0; ra < R; ++ra)
              acc[ra] = 0.0;
            for (int k = 0; k < m; ++k)
              {
                const double rk = Rb[k * ncs + r];
#pragma unroll
                for (int ra = 0; ra < R; ++ra)
                  {
                    const int i = ty + 16 * ra;
                    acc[ra] += ((i < m) ? Vs[i * ldv + k] : 0.0) * rk;
                  }
              }
#pragma unroll
            for (int ra = 0; ra < R; ++ra)
              {
                const int i = ty + 16 * ra;
                if (i < m)
                  {
                    const size_t gi = (size_t)l * xline + (size_t)i * ncs + r;
                    const double x  = xg[gi] - acc[ra];
                    xg[gi]          = x;
                    Zp[i * ncs + r] = x;
                  }
              }
          }
        __syncthreads();
      }
  }

  // ---------------------------------------------------------------------------------
  // K3: coarse Schur block, (S)LOD selection, normalisation, premultiplication
  // ---------------------------------------------------------------------------------
  __device__ __forceinline__ double group16_sum(double v)
  {
    v += __shfl_xor(v, 8, 16);
    v += __shfl_xor(v, 4, 16);
    v += __shfl_xor(v, 2, 16);
    v += __shfl_xor(v, 1, 16);
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

  template <int S>
  __global__ __launch_bounds__(256) void k_select(const SlodKernelArgs A, int nb_max, int nf_max)
  {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const SlodPatchDesc d   = A.desc[blockIdx.x];
    const int           tid = threadIdx.x;
    const int           nc = d.n_c, nb = d.n_b, n = A.n_sub;
    const int           ncm = A.nc_max, ldm = ncm + 1;
    const int           mm = A.m_max, ncs = A.nc_max;
    const bool          tr  = (d.flags & SLOD_F_TRANSPOSED) != 0;
    const bool          lod = (d.flags & SLOD_F_LOD) != 0;
    const int           npx = d.nx + 1, nn = npx * (d.ny + 1), nf = S * nn;

    double *Ms   = smem;                 // [ncm][ldm]  (later: D)
    double *Vj   = Ms + ncm * ldm;       // [ncm][ncm]  Jacobi rotations
    double *BD   = Vj + ncm * ncm;       // [nb_max][ncm]
    double *phis = BD + (size_t)nb_max * ncm; // [nf_max]
    double *sig  = phis + nf_max;        // [ncm]
    double *utg  = sig + ncm;
    double *gam  = utg + ncm;
    double *cvec = gam + ncm;
    double *rowk = cvec + ncm;           // [ncm]
    double *red  = rowk + ncm;           // [8]
    int    *ord  = reinterpret_cast<int *>(red + 8); // [ncm]
    int    *flag = ord + ncm;            // [2]

    const double *st    = A.st + (size_t)blockIdx.x * A.st_stride;
    const double *xg    = A.xs + (size_t)blockIdx.x * A.x_stride;
    const size_t  xline = (size_t)mm * ncs;

    // row of X for dof (ix,iy,comp), nullptr on the patch boundary (X_B = 0, LOD.cc:512-518)
    auto xrow = [&](int ix, int iy, int comp) -> const double * {
      if (ix <= 0 || ix >= d.nx || iy <= 0 || iy >= d.ny)
        return nullptr;
      const int l = tr ? ix - 1 : iy - 1, pos = tr ? iy - 1 : ix - 1;
      return xg + (size_t)l * xline + (size_t)(pos * S + comp) * ncs;
    };

    // ---- M = P^T X / H^dim (LOD.cc:548-551)
    for (int idx = tid; idx < nc * nc; idx += 256)
      {
        const int a = idx / nc, b = idx - a * nc;
        int       kx, ky;
        cell_of_col(d, a / S, kx, ky);
        double acc = 0.0;
        for (int jy = 0; jy <= n; ++jy)
          for (int jx = 0; jx <= n; ++jx)
            {
              const int ix = kx * n + jx, iy = ky * n + jy;
#pragma unroll
              for (int c = 0; c < S; ++c)
                {
                  const double w = pt_weight<S>(d, n, A.quirk, ix, iy, c, a);
                  if (w != 0.0)
                    {
                      const double *xr = xrow(ix, iy, c);
                      if (xr)
                        acc += w * xr[b];
                    }
                }
            }
        Ms[a * ldm + b] = acc * A.scale * A.invH2;
      }
    __syncthreads();

    // ---- D = M^{-1} (LOD.cc:553) by the symmetric sweep; M is SPD
    for (int k = 0; k < nc; ++k)
      {
        for (int j = tid; j < nc; j += 256)
          rowk[j] = Ms[k * ldm + j];
        __syncthreads();
        const double piv = rowk[k];
        if (tid == 0 && !(piv > 0.0))
          atomicOr(A.status, 2);
        const double p = 1.0 / piv;
        for (int idx = tid; idx < nc * nc; idx += 256)
          {
            const int    i = idx / nc, j = idx - i * nc;
            const double ri = rowk[i], rj = rowk[j];
            double       v;
            if (i == k)
              v = (j == k) ? -p : rj * p;
            else if (j == k)
              v = ri * p;
            else
              v = fma(-(ri * rj), p, Ms[i * ldm + j]);
            Ms[i * ldm + j] = v;
          }
        __syncthreads();
      }
    for (int idx = tid; idx < nc * nc; idx += 256)
      {
        const int i = idx / nc, j = idx - i * nc;
        Ms[i * ldm + j] = -Ms[i * ldm + j];
      }
    __syncthreads();
    double *Ds = Ms;

    for (int dsel = 0; dsel < S; ++dsel)
      {
        for (int j = tid; j < nc; j += 256)
          gam[j] = (j == dsel) ? 1.0 : 0.0;
        if (!lod)
          {
            // ---- BD = (S_BI X_I - P^T_B) D (LOD.cc:609-618); stencil rows instead of the
            //      dense S_boundary
            for (int idx = tid; idx < nb * nc; idx += 256)
              {
                const int bi = idx / nc, c = idx - bi * nc;
                const int bn = bi / S, ca = bi - bn * S;
                int       ix, iy;
                boundary_node(d, bn, ix, iy);
                double acc = -A.scale * pt_weight<S>(d, n, A.quirk, ix, iy, ca, c);
                for (int dy = -1; dy <= 1; ++dy)
                  for (int dx = -1; dx <= 1; ++dx)
                    {
                      const int jx = ix + dx, jy = iy + dy;
                      if (jx <= 0 || jx >= d.nx || jy <= 0 || jy >= d.ny)
                        continue;
                      const int dir = (dy + 1) * 3 + dx + 1;
#pragma unroll
                      for (int cb = 0; cb < S; ++cb)
                        acc += st[(size_t)((dir * S + ca) * S + cb) * A.nn_max + ix + iy * npx] *
                               xrow(jx, jy, cb)[c];
                    }
                BD[bi * ncm + c] = acc;
              }
            __syncthreads();
            // BD <- BD * D, one wave per row, row held across lanes (nc <= 64)
            {
              const int wave = tid >> 6, lane = tid & 63;
              for (int bi = wave; bi < nb; bi += 4)
                {
                  const double mine = (lane < nc) ? BD[bi * ncm + lane] : 0.0;
                  double       acc  = 0.0;
                  for (int j = 0; j < nc; ++j)
                    {
                      const double bj = __shfl(mine, j, 64);
                      acc += bj * ((lane < nc) ? Ds[j * ldm + lane] : 0.0);
                    }
                  if (lane < nc)
                    BD[bi * ncm + lane] = acc;
                }
            }
            // ---- one-sided Jacobi SVD of BD' = BD without column dsel (LOD.cc:656-667:
            //      sigma(G) = sigma(BD')^2, same right singular vectors)
            const int nn1 = nc - 1;           // columns of BD'
            const int nev = (nn1 + 1) & ~1;   // rounded up to even
            for (int idx = tid; idx < nn1 * nn1; idx += 256)
              Vj[idx] = ((idx / nn1) == (idx % nn1)) ? 1.0 : 0.0;
            __syncthreads();
            const int grp = tid >> 4, l16 = tid & 15;
            for (int sweep = 0; sweep < 40; ++sweep)
              {
                if (tid == 0)
                  flag[0] = 0;
                __syncthreads();
                for (int round = 0; round < nev - 1; ++round)
                  {
                    for (int pr = grp; pr < nev / 2; pr += 16)
                      {
                        int pa, pb;
                        if (pr == 0)
                          {
                            pa = nev - 1;
                            pb = round;
                          }
                        else
                          {
                            pa = (round + pr) % (nev - 1);
                            pb = (round - pr + (nev - 1)) % (nev - 1);
                          }
                        if (pa >= nn1 || pb >= nn1)
                          continue;
                        const int p = pa < pb ? pa : pb, q = pa < pb ? pb : pa;
                        const int cp = p < dsel ? p : p + 1, cq = q < dsel ? q : q + 1;
                        double    app = 0, aqq = 0, apq = 0;
                        for (int r = l16; r < nb; r += 16)
                          {
                            const double wp = BD[r * ncm + cp], wq = BD[r * ncm + cq];
                            app += wp * wp;
                            aqq += wq * wq;
                            apq += wp * wq;
                          }
                        app = group16_sum(app);
                        aqq = group16_sum(aqq);
                        apq = group16_sum(apq);
                        if (apq == 0.0 || fabs(apq) <= 1e-15 * sqrt(app * aqq))
                          continue;
                        const double zeta = (aqq - app) / (2.0 * apq);
                        const double t =
                          (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                        const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                        for (int r = l16; r < nb; r += 16)
                          {
                            const double wp = BD[r * ncm + cp], wq = BD[r * ncm + cq];
                            BD[r * ncm + cp] = cs * wp - sn * wq;
                            BD[r * ncm + cq] = sn * wp + cs * wq;
                          }
                        for (int r = l16; r < nn1; r += 16)
                          {
                            const double vp = Vj[r * nn1 + p], vq = Vj[r * nn1 + q];
                            Vj[r * nn1 + p] = cs * vp - sn * vq;
                            Vj[r * nn1 + q] = sn * vp + cs * vq;
                          }
                        if (l16 == 0)
                          flag[0] = 1;
                      }
                    __syncthreads();
                  }
                const int any = flag[0];
                __syncthreads();
                if (!any)
                  break;
              }
            // singular values of G and u_j^T g = w_j . b0
            for (int j = tid; j < nn1; j += 256)
              {
                const int cj = j < dsel ? j : j + 1;
                double    ss = 0, wb = 0;
                for (int r = 0; r < nb; ++r)
                  {
                    const double w = BD[r * ncm + cj];
                    ss += w * w;
                    wb += w * BD[r * ncm + dsel];
                  }
                sig[j] = ss;
                utg[j] = wb;
              }
            __syncthreads();
            if (tid == 0)
              {
                // descending sigma, pseudo-inverse cutoff (LOD.cc:667), d = -G^+ g, then the
                // 0.5-loop (LOD.cc:703-725).  gam[] doubles as d_i storage (others order).
                for (int j = 0; j < nn1; ++j)
                  ord[j] = j;
                for (int a2 = 1; a2 < nn1; ++a2)
                  {
                    const int o = ord[a2];
                    int       b = a2 - 1;
                    while (b >= 0 && sig[ord[b]] < sig[o])
                      {
                        ord[b + 1] = ord[b];
                        --b;
                      }
                    ord[b + 1] = o;
                  }
                const double s0 = sig[ord[0]];
                double      *del = rowk; // scratch [nn1]
                for (int a2 = 0; a2 < nn1; ++a2)
                  del[a2] = 0.0;
                for (int r = 0; r < nn1; ++r)
                  {
                    const int j = ord[r];
                    utg[j]      = (sig[j] > 1e-15 * s0) ? utg[j] / sig[j] : 0.0;
                    for (int a2 = 0; a2 < nn1; ++a2)
                      del[a2] -= Vj[a2 * nn1 + j] * utg[j];
                  }
                for (int r = nn1 - 1; r >= 0; --r)
                  {
                    double dinf = 0.0;
                    for (int a2 = 0; a2 < nn1; ++a2)
                      dinf = fmax(dinf, fabs(del[a2]));
                    if (dinf < 0.5)
                      break;
                    const int j = ord[r];
                    for (int a2 = 0; a2 < nn1; ++a2)
                      del[a2] += Vj[a2 * nn1 + j] * utg[j];
                  }
                for (int j = 0, jj = 0; j < nc; ++j)
                  if (j != dsel)
                    gam[j] = del[jj++];
              }
          }
        __syncthreads();
        // ---- c = D gamma (LOD.cc:727-743 / 576-577)
        for (int i = tid; i < nc; i += 256)
          {
            double acc = 0.0;
            for (int j = 0; j < nc; ++j)
              acc += Ds[i * ldm + j] * gam[j];
            cvec[i] = acc;
          }
        __syncthreads();
        // ---- phi = X c, zero on the boundary (LOD.cc:745-750), l2-normalised (LOD.cc:752)
        double ssq = 0.0;
        for (int dof = tid; dof < nf; dof += 256)
          {
            const int     node = dof / S, comp = dof - node * S;
            const int     ix = node % npx, iy = node / npx;
            const double *xr  = xrow(ix, iy, comp);
            double        acc = 0.0;
            if (xr)
              for (int j = 0; j < nc; ++j)
                acc += xr[j] * cvec[j];
            phis[dof] = acc;
            ssq += acc * acc;
          }
        for (int off = 32; off > 0; off >>= 1)
          ssq += __shfl_xor(ssq, off, 64);
        if ((tid & 63) == 0)
          red[tid >> 6] = ssq;
        __syncthreads();
        const double nrm = sqrt(red[0] + red[1] + red[2] + red[3]);
        double      *ob  = A.basis + d.out_off + (size_t)dsel * nf;
        double      *op  = A.premult + d.out_off + (size_t)dsel * nf;
        for (int dof = tid; dof < nf; dof += 256)
          {
            const double v = phis[dof] / nrm;
            phis[dof]      = v;
            ob[dof]        = v;
          }
        __syncthreads();
        // ---- psi = A_semi phi: identity rows on id-0 dofs (LOD.cc:537-541,758-765)
        for (int dof = tid; dof < nf; dof += 256)
          {
            const int  node = dof / S, comp = dof - node * S;
            const int  ix = node % npx, iy = node / npx;
            const bool dom = (ix == 0 && (d.flags & 1)) || (ix == d.nx && (d.flags & 2)) ||
                             (iy == 0 && (d.flags & 4)) || (iy == d.ny && (d.flags & 8));
            double acc;
            if (dom)
              acc = phis[dof];
            else
              {
                acc = 0.0;
                for (int dy = -1; dy <= 1; ++dy)
                  for (int dx = -1; dx <= 1; ++dx)
                    {
                      const int jx = ix + dx, jy = iy + dy;
                      if (jx < 0 || jx > d.nx || jy < 0 || jy > d.ny)
                        continue;
                      const int dir = (dy + 1) * 3 + dx + 1;
#pragma unroll
                      for (int cb = 0; cb < S; ++cb)
                        acc += st[(size_t)((dir * S + comp) * S + cb) * A.nn_max + node] *
                               phis[(jx + jy * npx) * S + cb];
                    }
              }
            op[dof] = acc;
          }
        __syncthreads();
      }
  }
} // namespace

// -------------------------------------------------------------------------------------
// launchers
// -------------------------------------------------------------------------------------
size_t slod_solve_lds_bytes(int S, int m_max, int nc_max)
{
  const int    R = (m_max + 15) / 16, BW = 2 * (2 * S - 1) + 1;
  const size_t n = (size_t)m_max * (m_max + 1) + 2 * (size_t)m_max * nc_max + 2 * 16 * R +
                   3 * (size_t)m_max * BW;
  return n * sizeof(double);
}

size_t slod_select_lds_bytes(int /*S*/, int nb_max, int nc_max, int nf_max)
{
  const size_t n = (size_t)nc_max * (nc_max + 1) + (size_t)nc_max * nc_max +
                   (size_t)nb_max * nc_max + nf_max + 5 * (size_t)nc_max + 8;
  return n * sizeof(double) + ((size_t)nc_max + 2) * sizeof(int);
}

hipError_t slod_launch_assemble(int S, const SlodKernelArgs &a, int n_patches, hipStream_t st)
{
  const dim3 grid((a.nn_max + 255) / 256, n_patches);
  if (S == 1)
    hipLaunchKernelGGL(k_assemble<1>, grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(k_assemble<2>, grid, dim3(256), 0, st, a);
  return hipGetLastError();
}

template <int R>
static hipError_t launch_solve_R(int S, const SlodKernelArgs &a, int n_patches, size_t lds,
                                 hipStream_t st)
{
  hipError_t e;
  if (S == 1)
    {
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_solve<R, 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess)
        return e;
      hipLaunchKernelGGL((k_solve<R, 1>), dim3(n_patches), dim3(256), lds, st, a);
    }
  else
    {
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_solve<R, 2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess)
        return e;
      hipLaunchKernelGGL((k_solve<R, 2>), dim3(n_patches), dim3(256), lds, st, a);
    }
  return hipGetLastError();
}

hipError_t slod_launch_solve(int S, const SlodKernelArgs &a, int n_patches, hipStream_t st)
{
  const size_t lds = slod_solve_lds_bytes(S, a.m_max, a.nc_max);
  switch ((a.m_max + 15) / 16)
    {
      case 1:
        return launch_solve_R<1>(S, a, n_patches, lds, st);
      case 2:
        return launch_solve_R<2>(S, a, n_patches, lds, st);
      case 3:
        return launch_solve_R<3>(S, a, n_patches, lds, st);
      case 4:
        return launch_solve_R<4>(S, a, n_patches, lds, st);
      case 5:
        return launch_solve_R<5>(S, a, n_patches, lds, st);
      case 6:
        return launch_solve_R<6>(S, a, n_patches, lds, st);
      case 7:
        return launch_solve_R<7>(S, a, n_patches, lds, st);
      default:
        return hipErrorInvalidValue;
    }
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
