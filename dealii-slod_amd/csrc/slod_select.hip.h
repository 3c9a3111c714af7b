// K3 as a device function: coarse Schur block, (S)LOD selection, normalisation, premultiplication.
#ifndef SLOD_SELECT_HIP_H
#define SLOD_SELECT_HIP_H
#include "slod_common.hip.h"

namespace
{
  // ---------------------------------------------------------------------------------
  // K3: coarse Schur block, (S)LOD selection, normalisation, premultiplication
  // ---------------------------------------------------------------------------------

  // The SLOD selection needs  d = -(BD')^+ b0  (LOD.cc:656-671) and, only if ||d||_inf >= 0.5
  // or a singular value falls under the 1e-15 cutoff, the singular triplets of BD' for the
  // truncation loop (LOD.cc:703-725).  So: Householder QR of [BD' | b0] in LDS first.  With
  // R (n x n) and c = Q^T b0:  d = -R^{-1} c.  cond(R) <= ||R||_F ||R^{-1}||_F =: kF is a
  // rigorous bound, so kF^2 < 1e14 proves that no singular value of G = R^T R is cut, and
  // ||d||_inf < 0.5 (with a 1e-9 guard band) proves the loop removes nothing: the fast path
  // takes exactly the reference's decisions.  Otherwise a one-sided Jacobi SVD of R (same
  // singular values / right vectors as BD', u_j^T g = (R v_j).c) replays the loop literally.
  // One 256-thread workgroup, patch `patch` of the launch; `smem` = the workgroup's dynamic LDS
  // (slod_select_lds_bytes).  Called by k_select and, fused, at the end of k_solve_tw.
  // OWN: the stage is a kernel of its own (k_select) and may spend registers on batched LDS loads;
  // fused into a solve kernel it shares that kernel's tighter register budget
  template <int S, bool OWN = false>
  __device__ __forceinline__ void select_patch(const SlodKernelArgs &A, const int nb_max, const int nf_max,
                                               const int patch, double *smem)
  {
    const SlodPatchDesc d   = A.desc[patch];
    const int           tid = threadIdx.x;
    const int           nc = d.n_c, nb = d.n_b, n = A.n_sub;
    const int           ncm = A.nc_max, ldm = ncm + 1;
    const int           mm = A.m_max, ncs = A.nc_max;
    const bool          tr  = (d.flags & SLOD_F_TRANSPOSED) != 0;
    const bool          lod = (d.flags & SLOD_F_LOD) != 0;
    const int           npx = d.nx + 1, nn = npx * (d.ny + 1), nf = S * nn;

    double *Ms   = smem;                 // [ncm][ldm]  M, then D = M^-1
    double *Vj   = Ms + ncm * ldm;       // [ncm][ncm]  R^-1 / Jacobi rotations
    double *BD   = Vj + ncm * ncm;       // [nb_max][ncm]  (nb_max = buffer rows, see TSQR below)
    double *phis = BD;                   // [nf_max] aliases BD (dead once gamma is known)
    double *sig  = BD + max(nb_max * ncm, nf_max); // [ncm]
    double *utg  = sig + ncm;
    double *gam  = utg + ncm;
    double *cvec = gam + ncm;
    double *rowk = cvec + ncm;           // [ncm]
    double *red  = rowk + ncm;           // [8]
    const int vst = max(nb_max, 160);
    double *vcol = red + 8;              // [2][vst] pivot column of the register-resident QR
    int    *colk = reinterpret_cast<int *>(vcol + 2 * vst); // [2][ncm] cell of column
    int    *ord  = colk + 2 * ncm;       // [ncm]
    int    *flag = ord + ncm;            // [4]
    int    *pcol = flag + 4;             // [2][ncm] column order of the pivoted second-stage QR

    const double *st    = A.st + (size_t)patch * A.st_stride;
    const double *xg    = A.xs + (size_t)patch * A.x_stride;
    const size_t  xline = (size_t)mm * ncs;
    const int     wave = tid >> 6, lane = tid & 63, grp = tid >> 4, l16 = tid & 15;

    // row of X for dof (ix,iy,comp), nullptr on the patch boundary (X_B = 0, LOD.cc:512-518)
    auto xrow = [&](int ix, int iy, int comp) -> const double * {
      if (ix <= 0 || ix >= d.nx || iy <= 0 || iy >= d.ny)
        return nullptr;
      const int l = tr ? ix - 1 : iy - 1, pos = tr ? iy - 1 : ix - 1;
      return xg + (size_t)l * xline + (size_t)(pos * S + comp) * ncs;
    };
    // entry of the un-zeroed P^T / (h^2/4)
    auto ptw = [&](int ix, int iy, int comp, int col) -> double {
      if (S == 1)
        {
          const int jx = ix - colk[col] * n, jy = iy - colk[ncm + col] * n;
          if (jx < 0 || jx > n || jy < 0 || jy > n)
            return 0.0;
          return ((jx == 0 || jx == n) ? 1.0 : 2.0) * ((jy == 0 || jy == n) ? 1.0 : 2.0);
        }
      return pt_weight<S>(d, n, A.quirk, ix, iy, comp, col);
    };
    auto block_sum = [&](double v) -> double { // all threads get the sum
      for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off, 64);
      __syncthreads();
      if (lane == 0)
        red[wave] = v;
      __syncthreads();
      return red[0] + red[1] + red[2] + red[3];
    };

    for (int c = tid; c < nc; c += 256)
      {
        int kx, ky;
        cell_of_col(d, c / S, kx, ky);
        colk[c]       = kx;
        colk[ncm + c] = ky;
      }
    __syncthreads();

    // timing experiments (SLOD_DIAG bit 20): thread 0 stamps the 100 MHz clock per phase into ms
    const bool stamping = (SLOD_DG(A, (1 << 20))) && tid == 0 && A.nc_max * A.nc_max >= 12;
    double    *msd      = A.ms + (size_t)patch * A.nc_max * A.nc_max;
    double     tph[3]   = {0.0, 0.0, 0.0};
    double     tlast    = stamping ? (double)wall_clock64() : 0.0;
    auto       stamp    = [&](int i) {
      if (stamping)
        msd[i] = tlast = (double)wall_clock64();
    };
    auto tacc = [&](int i) {
      if (stamping)
        {
          const double now = (double)wall_clock64();
          tph[i] += now - tlast;
          tlast = now;
        }
    };

    // ---- M = P^T X / H^dim (LOD.cc:548-551).  X rows come from the global workspace: the
    //      inner loop has no control dependence (clamped address, zero weight on the patch
    //      boundary where X = 0) so its n+1 loads are in flight together.
    if (A.m_fused)
      {
        const double *mg = A.ms + (size_t)patch * A.nc_max * A.nc_max;
        for (int idx = tid; idx < nc * nc; idx += 256)
          Ms[(idx / nc) * ldm + (idx % nc)] = mg[idx];
      }
    // M = P^T A^-1 P is symmetric and the sweep below treats it as such (column k is taken from
    // row k), so only the entries a <= b are computed (half the load batches) and mirrored
    for (int idx = tid; idx < ((SLOD_DG(A, 64)) || A.m_fused ? 0 : nc * (nc + 1) / 2); idx += 256)
      {
        // row a of the upper triangle holds nc - a entries: a = largest a with a (2 nc - a + 1) / 2 <= idx
        int a = (int)((2.0f * nc + 1.0f - sqrtf((2.0f * nc + 1.0f) * (2.0f * nc + 1.0f) - 8.0f * idx)) * 0.5f);
        a     = min(max(a, 0), nc - 1);
        while (a > 0 && a * (2 * nc - a + 1) / 2 > idx)
          --a;
        while (a + 1 < nc && (a + 1) * (2 * nc - a) / 2 <= idx)
          ++a;
        const int b  = a + idx - a * (2 * nc - a + 1) / 2;
        const int kx = colk[a], ky = colk[ncm + a];
        double    acc = 0.0;
        if (S == 1 || !A.quirk)
          {
            // 3 x 9 points per batch, all loads of a batch independent of each other and of any
            // branch (clamped address, zero weight outside the cell / on the patch boundary):
            // the phase is bound by the latency of the batches, so few and wide ones
            const int ca = a % S;
            for (int jy0 = 0; jy0 <= n; jy0 += 3)
              for (int jx0 = 0; jx0 <= n; jx0 += 9)
                {
                  double xv[3][9];
#pragma unroll
                  for (int ry = 0; ry < 3; ++ry)
                    {
                      const int iy  = ky * n + jy0 + ry;
                      const int iyc = min(max(iy, 1), d.ny - 1);
#pragma unroll
                      for (int rx = 0; rx < 9; ++rx)
                        {
                          const int ix  = kx * n + jx0 + rx;
                          const int ixc = min(max(ix, 1), d.nx - 1);
                          const int l = tr ? ixc - 1 : iyc - 1, pos = tr ? iyc - 1 : ixc - 1;
                          xv[ry][rx]  = xg[(size_t)l * xline + (size_t)(pos * S + ca) * ncs + b];
                        }
                    }
#pragma unroll
                  for (int ry = 0; ry < 3; ++ry)
                    {
                      const int    jy = jy0 + ry, iy = ky * n + jy;
                      const double wy = (jy <= n && iy > 0 && iy < d.ny) ? ((jy == 0 || jy == n) ? 1.0 : 2.0) : 0.0;
                      double       part = 0.0;
#pragma unroll
                      for (int rx = 0; rx < 9; ++rx)
                        {
                          const int    jx = jx0 + rx, ix = kx * n + jx;
                          const double w = (jx <= n && ix > 0 && ix < d.nx) ? ((jx == 0 || jx == n) ? 1.0 : 2.0) : 0.0;
                          part           = fma(w, xv[ry][rx], part);
                        }
                      acc = fma(wy, part, acc);
                    }
                }
          }
        else
          {
            for (int jy = 0; jy <= n; ++jy)
              for (int jx = 0; jx <= n; ++jx)
                {
                  const int ix = kx * n + jx, iy = ky * n + jy;
#pragma unroll
                  for (int c = 0; c < S; ++c)
                    {
                      const double *xr = xrow(ix, iy, c);
                      if (xr)
                        acc = fma(pt_weight<S>(d, n, A.quirk, ix, iy, c, a), xr[b], acc);
                    }
                }
          }
        Ms[a * ldm + b] = acc * A.scale * A.invH2;
        Ms[b * ldm + a] = acc * A.scale * A.invH2;
      }
    __syncthreads();
    stamp(3);

    // ---- D = M^{-1} (LOD.cc:553) by the symmetric sweep; M is SPD
    if (nc * nc <= 1024)
      {
        // register-resident: every thread keeps its <= 4 entries and their (i,j); the pivot row is
        // handed on through a double-buffered LDS line by the threads that own row k+1, so a pivot
        // costs one barrier and no index arithmetic
        double mv[4];
        int    mi[4], mj[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          {
            const int idx = tid + 256 * e;
            mi[e]         = idx / nc;
            mj[e]         = idx - mi[e] * nc;
            mv[e]         = (idx < nc * nc) ? Ms[mi[e] * ldm + mj[e]] : 0.0;
            if (idx < nc * nc && mi[e] == 0)
              cvec[mj[e]] = mv[e];
          }
        __syncthreads();
        for (int k = 0; k < ((SLOD_DG(A, 128)) ? 0 : nc); ++k)
          {
            const double *rk  = (k & 1) ? rowk : cvec;
            double       *rn  = (k & 1) ? cvec : rowk;
            const double  piv = rk[k];
            if (tid == 0 && !(piv > 0.0) && !SLOD_DG(A, -1))
              atomicOr(A.status, 2);
            const double p = fast_rcp(piv);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (tid + 256 * e < nc * nc)
                {
                  const int    i = mi[e], j = mj[e];
                  const double ri = rk[i], rj = rk[j];
                  double       v;
                  if (i == k)
                    v = (j == k) ? -p : rj * p;
                  else if (j == k)
                    v = ri * p;
                  else
                    v = fma(-(ri * rj), p, mv[e]);
                  mv[e] = v;
                  if (i == k + 1)
                    rn[j] = v;
                }
            __syncthreads();
          }
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (tid + 256 * e < nc * nc)
            Ms[mi[e] * ldm + mj[e]] = -mv[e];
        __syncthreads();
      }
    else
      {
        for (int k = 0; k < ((SLOD_DG(A, 128)) ? 0 : nc); ++k)
          {
            for (int j = tid; j < nc; j += 256)
              rowk[j] = Ms[k * ldm + j];
            __syncthreads();
            const double piv = rowk[k];
            if (tid == 0 && !(piv > 0.0) && !SLOD_DG(A, -1))
              atomicOr(A.status, 2);
            const double p = fast_rcp(piv);
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
      }
    stamp(4);
    double *Ds = Ms;

    for (int dsel = 0; dsel < S; ++dsel)
      {
        for (int j = tid; j < nc; j += 256)
          gam[j] = (j == dsel) ? 1.0 : 0.0;
        // decisions of this (patch, component), reported through slod_plan_diagnostics (thread 0)
        SlodPatchDiag pdg = {0, 0, 0, 0, 0.0, 0.0, 0.0};
        if (!lod)
          {
            pdg.path = 1;
            // ---- BD = (S_BI X_I - P^T_B) D (LOD.cc:609-618), built in row chunks that fit the
            //      LDS buffer (nbuf rows) and reduced by Householder QR chunk after chunk
            //      (TSQR): after every chunk the top nn1 rows hold the R factor of all rows
            //      seen so far and c = Q^T b0 sits in column dsel.
            const int nn1 = nc - 1; // columns of BD' = BD without column dsel
            auto      cix = [&](int j) { return j < dsel ? j : j + 1; };
            const int nbuf = nb_max;
            int       nr   = 0;     // rows of the matrix the SVD fallback works on
            bool      need_svd = true, singular = false, did_qr = false;
            int       filled = 0;
            // Householder sweep of a register-resident rows x [BD' | b0] block: the 16-lane group g
            // holds logical columns g and g+16 (column j < nn1 is BD' column j, column nn1 is b0), lane
            // l16 of it the rows l16 + 16 i.  Only the pivot column goes through LDS (double buffered,
            // published together with its norm by the group that owns it one step ahead): one barrier
            // and ~8 LDS operations per reflector and thread instead of ~60.
            auto qr_sweep = [&](auto &a, const int rows, const bool last) __attribute__((always_inline)) {
              constexpr int RQ = sizeof(a[0]) / sizeof(double);
            if (grp == 0)
              {
                double nx0 = 0.0;
#pragma unroll
                for (int i = 0; i < RQ; ++i)
                  {
                    nx0 = fma(a[0][i], a[0][i], nx0);
                    if (l16 + 16 * i < rows)
                      vcol[l16 + 16 * i] = a[0][i];
                  }
                nx0 = group16_sum(nx0);
                if (l16 == 0)
                  sig[0] = nx0;
              }
            __syncthreads();
            for (int k = 0; k < nn1; ++k)
              {
                const double *vc    = vcol + (k & 1) * vst;
                double       *vn    = vcol + ((k + 1) & 1) * vst;
                const double  sigma = sig[k & 1];
                const bool    act   = sigma > 0.0;
                if (!act && last)
                  singular = true; // zero column (rank deficient): replayed through the SVD
                const double x0    = vc[k];
                const double sq    = act ? sigma * fast_rsqrt(sigma) : 0.0;
                const double alpha = (x0 >= 0.0) ? -sq : sq;
                const double v0    = x0 - alpha;
                const double beta  = act ? fast_rcp(sigma - alpha * x0) : 0.0; // 2 / v^T v
                double       vr[RQ];
#pragma unroll
                for (int i = 0; i < RQ; ++i)
                  {
                    const int r = l16 + 16 * i;
                    vr[i]       = (r == k) ? v0 : ((r > k && r < rows) ? vc[r] : 0.0);
                  }
#pragma unroll
                for (int sl = 0; sl < 2; ++sl)
                  {
                    const int j = grp + 16 * sl;
                    if (j == k)
                      {
                        // this column is finished: R_kk on the diagonal, zeros below
#pragma unroll
                        for (int i = 0; i < RQ; ++i)
                          {
                            const int r = l16 + 16 * i;
                            a[sl][i]    = (r == k) ? (act ? alpha : a[sl][i]) : (r > k ? 0.0 : a[sl][i]);
                          }
                      }
                    else if (j > k && j <= nn1)
                      {
                        double sd = 0.0;
#pragma unroll
                        for (int i = 0; i < RQ; ++i)
                          sd = fma(vr[i], a[sl][i], sd);
                        sd = group16_sum(sd) * beta;
#pragma unroll
                        for (int i = 0; i < RQ; ++i)
                          a[sl][i] = fma(-sd, vr[i], a[sl][i]);
                        if (j == k + 1 && j < nn1) // the next pivot column: publish it
                          {
                            double nxt = 0.0;
#pragma unroll
                            for (int i = 0; i < RQ; ++i)
                              {
                                const int r = l16 + 16 * i;
                                if (r > k)
                                  nxt = fma(a[sl][i], a[sl][i], nxt);
                                if (r < rows)
                                  vn[r] = a[sl][i];
                              }
                            nxt = group16_sum(nxt);
                            if (l16 == 0)
                              sig[(k + 1) & 1] = nxt;
                          }
                      }
                  }
                __syncthreads();
              }
            };
            // one pass over all boundary rows when they fit the registers (<= 160 rows): the row chunks
            // only stage BD through LDS (fill, * D), a single sweep of nn1 reflectors follows
            const bool onepass = nb <= 160 && nb > 96 && nc <= 32 && (nbuf & 15) == 0 && !(SLOD_DG(A, (512 | 262144)));
            double     aq[2][10];
            for (int r0 = 0; r0 < nb;)
              {
                const int take = min(nb - r0, nbuf - filled);
                // stencil rows instead of the dense S_boundary.  A boundary node has at most three
                // interior neighbours: the ones one step along the inward normal (the other six
                // couplings of its stencil row multiply X_B = 0), so an entry costs 3 stencil + 3 X
                // loads; the phase is bound by the latency of its dependent load batches, so a thread
                // works on U entries at once.  Lanes run along the columns of a row (coalesced X rows).
                {
                  constexpr int U = 2;
                  for (int idx0 = tid; idx0 < ((SLOD_DG(A, 256)) ? 0 : take * nc); idx0 += 256 * U)
                    {
                      double sv[U][3][S], xv[U][3][S], acc[U];
                      int    dst[U];
#pragma unroll
                      for (int u = 0; u < U; ++u)
                        {
                          const int  idx = idx0 + 256 * u;
                          const bool ok  = idx < take * nc;
                          const int  br = ok ? idx / nc : 0, c = ok ? idx - br * nc : 0;
                          const int  bi = r0 + br;
                          const int  bn = bi / S, ca = bi - bn * S;
                          int        ix, iy;
                          boundary_node(d, bn, ix, iy);
                          acc[u] = -A.scale * ptw(ix, iy, ca, c);
                          dst[u] = ok ? (filled + br) * ncm + c : -1;
                          const bool hor = (iy == 0 || iy == d.ny); // bottom / top row: neighbours along x
                          const int  ndx = ix == 0 ? 1 : -1, ndy = iy == 0 ? 1 : -1;
#pragma unroll
                          for (int e = -1; e <= 1; ++e)
                            {
                              const int  dx = hor ? e : ndx, dy = hor ? ndy : e;
                              const int  jx = ix + dx, jy = iy + dy;
                              const bool in = (jx > 0 && jx < d.nx && jy > 0 && jy < d.ny);
                              const int  jxc = min(max(jx, 1), d.nx - 1), jyc = min(max(jy, 1), d.ny - 1);
                              const int  dir = (dy + 1) * 3 + dx + 1;
                              const int  l = tr ? jxc - 1 : jyc - 1, pos = tr ? jyc - 1 : jxc - 1;
#pragma unroll
                              for (int cb = 0; cb < S; ++cb)
                                {
                                  // unconditional loads + select (a predicated load costs a branch)
                                  const double svv = st[(size_t)((dir * S + ca) * S + cb) * A.nn_max + ix + iy * npx];
                                  sv[u][e + 1][cb] = in ? svv : 0.0;
                                  xv[u][e + 1][cb] = xg[(size_t)l * xline + (size_t)(pos * S + cb) * ncs + c];
                                }
                            }
                        }
#pragma unroll
                      for (int u = 0; u < U; ++u)
                        {
#pragma unroll
                          for (int e = 0; e < 3; ++e)
#pragma unroll
                            for (int cb = 0; cb < S; ++cb)
                              acc[u] = fma(sv[u][e][cb], xv[u][e][cb], acc[u]);
                          if (dst[u] >= 0)
                            BD[dst[u]] = acc[u];
                        }
                    }
                }
                __syncthreads();
                tacc(0);
                // rows <- rows * D on the fp64 matrix pipe (v_mfma_f64_16x16x4_f64): a wave owns
                // 16-row tiles, reads a tile completely (all k, all <= 4 column tiles: nc <= 64)
                // and then overwrites it in place
                for (int ti = wave; 16 * ti < take; ti += 4)
                  {
                    const int  arow = filled + 16 * ti + (lane & 15);
                    const bool aok  = 16 * ti + (lane & 15) < take;
                    double4_t  acc[4];
#pragma unroll
                    for (int tj = 0; tj < 4; ++tj)
                      acc[tj] = double4_t{0.0, 0.0, 0.0, 0.0};
                    for (int k = 0; k < nc; k += 4)
                      {
                        const int    kk = k + (lane >> 4);
                        const double av = (aok && kk < nc) ? BD[arow * ncm + kk] : 0.0;
#pragma unroll
                        for (int tj = 0; tj < 4; ++tj)
                          if (16 * tj < nc)
                            {
                              const int    col = 16 * tj + (lane & 15);
                              const double bv  = (kk < nc && col < nc) ? Ds[kk * ldm + col] : 0.0;
                              acc[tj]          = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[tj], 0, 0, 0);
                            }
                      }
#pragma unroll
                    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                      for (int r = 0; r < 4; ++r)
                        {
                          const int rr = 16 * ti + (lane >> 4) + 4 * r, col = 16 * tj + (lane & 15);
                          if (rr < take && col < nc)
                            BD[(filled + rr) * ncm + col] = acc[tj][r];
                        }
                  }
                __syncthreads();
                tacc(1);
                if (onepass)
                  {
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl)
                      {
                        const int j = grp + 16 * sl, cj = j < nn1 ? cix(j) : dsel;
#pragma unroll
                        for (int i = 0; i < 10; ++i)
                          {
                            const int r = l16 + 16 * i;
                            if (r >= r0 && r < r0 + take)
                              aq[sl][i] = (j <= nn1) ? BD[(r - r0) * ncm + cj] : 0.0;
                            else if (r0 == 0)
                              aq[sl][i] = 0.0;
                          }
                      }
                    r0 += take;
                    __syncthreads(); // the next chunk overwrites the staging rows
                    continue;
                  }
                r0 += take;
                const int rows = filled + take;
                nr             = rows;
                if (rows < nn1 || (SLOD_DG(A, 512)))
                  {
                    filled = rows; // fewer rows than columns so far
                    if (filled >= nbuf)
                      break;       // cannot happen: nbuf > nn1
                    continue;
                  }
                // ---- Householder QR of the rows x [BD' | b0] block.
                did_qr = true;
                if (rows <= 96 && nc <= 32)
                  {
                    // Register-resident: the 16-lane group g holds logical columns g and g+16 (column
                    // j < nn1 is BD' column j, column nn1 is b0), lane l16 of it the rows l16 + 16 i.
                    // Only the pivot column goes through LDS (double buffered, published together with
                    // its norm by the group that owns it one step ahead): one barrier and ~8 LDS
                    // operations per reflector and thread instead of ~60.
                    double a[2][6];
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl)
                      {
                        const int j = grp + 16 * sl;
                        const int cj = j < nn1 ? cix(j) : dsel;
#pragma unroll
                        for (int i = 0; i < 6; ++i)
                          {
                            const int r = l16 + 16 * i;
                            a[sl][i]    = (j <= nn1 && r < rows) ? BD[r * ncm + cj] : 0.0;
                          }
                      }
                    qr_sweep(a, rows, r0 >= nb);
                    // back to LDS: R in the top nn1 rows (zero below the diagonal), zero rows below
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl)
                      {
                        const int j = grp + 16 * sl;
                        const int cj = j < nn1 ? cix(j) : dsel;
#pragma unroll
                        for (int i = 0; i < 6; ++i)
                          {
                            const int r = l16 + 16 * i;
                            if (j <= nn1 && r < rows)
                              BD[r * ncm + cj] = a[sl][i];
                          }
                      }
                  }
                else
                  {
                    // generic path, matrix in LDS: one barrier per reflector, the 16-lane group that
                    // updates the NEXT pivot column also accumulates its norm below the diagonal
                    {
                      double part = 0.0;
                      for (int r = tid; r < rows; r += 256)
                        {
                          const double x = BD[r * ncm + cix(0)];
                          part           = fma(x, x, part);
                        }
                      const double s00 = block_sum(part);
                      if (tid == 0)
                        sig[0] = s00;
                      __syncthreads();
                    }
                    for (int k = 0; k < nn1; ++k)
                      {
                        const int    ck    = cix(k);
                        const double sigma = sig[k & 1];
                        if (!(sigma > 0.0))
                          {
                            // zero column (rank deficient): no reflector; the next column's norm
                            if (r0 >= nb)
                              singular = true; // replayed through the SVD
                            if (k + 1 < nn1)
                              {
                                double part = 0.0;
                                for (int r = k + 1 + tid; r < rows; r += 256)
                                  {
                                    const double x = BD[r * ncm + cix(k + 1)];
                                    part           = fma(x, x, part);
                                  }
                                const double sn = block_sum(part);
                                if (tid == 0)
                                  sig[(k + 1) & 1] = sn;
                              }
                            __syncthreads();
                            continue;
                          }
                        const double x0    = BD[k * ncm + ck];
                        const double sq    = sigma * fast_rsqrt(sigma);
                        const double alpha = (x0 >= 0.0) ? -sq : sq;
                        const double v0    = x0 - alpha;
                        const double beta  = fast_rcp(sigma - alpha * x0); // 2 / v^T v
                        if constexpr (OWN)
                          {
                            // apply H = I - beta v v^T to the trailing columns and to b0.  The lane's rows
                            // of the reflector and of a column are loaded in one batch each (the buffer
                            // holds at most 16 NRL rows; unconditional, clamped loads): the dependent
                            // chain is then arithmetic only, not one LDS round trip per row
                            constexpr int NRL = 10;
                            double        vr[NRL];
    #pragma unroll
                            for (int i = 0; i < NRL; ++i)
                              {
                                const int    r = k + l16 + 16 * i;
                                const double x = BD[min(r, rows - 1) * ncm + ck];
                                vr[i]          = (r < rows) ? ((r == k) ? v0 : x) : 0.0;
                              }
                            for (int t = grp; t < nn1 - k; t += 16)
                              {
                                const int cj = (t == nn1 - k - 1) ? dsel : cix(k + 1 + t);
                                double    xj[NRL];
    #pragma unroll
                                for (int i = 0; i < NRL; ++i)
                                  xj[i] = BD[min(k + l16 + 16 * i, rows - 1) * ncm + cj];
                                double sd = 0.0;
    #pragma unroll
                                for (int i = 0; i < NRL; ++i)
                                  sd = (k + l16 + 16 * i < rows) ? fma(vr[i], xj[i], sd) : sd;
                                sd = group16_sum(sd) * beta;
                                double nxt = 0.0;
    #pragma unroll
                                for (int i = 0; i < NRL; ++i)
                                  {
                                    const int r = k + l16 + 16 * i;
                                    if (r < rows)
                                      {
                                        const double nv  = fma(-sd, vr[i], xj[i]);
                                        BD[r * ncm + cj] = nv;
                                        if (r > k)
                                          nxt = fma(nv, nv, nxt);
                                      }
                                  }
                                if (t == 0 && k + 1 < nn1) // cj is the next pivot column
                                  {
                                    nxt = group16_sum(nxt);
                                    if (l16 == 0)
                                      sig[(k + 1) & 1] = nxt;
                                  }
                              }
                          }
                        else
                          {
                            // apply H = I - beta v v^T to the trailing columns and to b0
                            for (int t = grp; t < nn1 - k; t += 16)
                              {
                                const int cj = (t == nn1 - k - 1) ? dsel : cix(k + 1 + t);
                                double    sd = 0.0;
                                for (int r = k + l16; r < rows; r += 16)
                                  {
                                    const double vr = (r == k) ? v0 : BD[r * ncm + ck];
                                    sd              = fma(vr, BD[r * ncm + cj], sd);
                                  }
                                sd = group16_sum(sd) * beta;
                                double nxt = 0.0;
                                for (int r = k + l16; r < rows; r += 16)
                                  {
                                    const double vr = (r == k) ? v0 : BD[r * ncm + ck];
                                    const double nv = fma(-sd, vr, BD[r * ncm + cj]);
                                    BD[r * ncm + cj] = nv;
                                    if (r > k)
                                      nxt = fma(nv, nv, nxt);
                                  }
                                if (t == 0 && k + 1 < nn1) // cj is the next pivot column
                                  {
                                    nxt = group16_sum(nxt);
                                    if (l16 == 0)
                                      sig[(k + 1) & 1] = nxt;
                                  }
                              }
                          }
                        __syncthreads();
                        if (tid == 0)
                          BD[k * ncm + ck] = alpha; // R_kk (after the barrier: x0 was read from here)
                      }
                    // clear the strict lower triangle of the R block (dead reflector storage): the
                    // next chunk's QR and the SVD fallback read it as part of the matrix
                    for (int idx = tid; idx < nn1 * nn1; idx += 256)
                      {
                        const int r = idx / nn1, j = idx - r * nn1;
                        if (r > j)
                          BD[r * ncm + cix(j)] = 0.0;
                      }
                    for (int r = nn1 + tid; r < rows; r += 256)
                      for (int j = 0; j < nn1; ++j)
                        BD[r * ncm + cix(j)] = 0.0;
                  }
                __syncthreads();
                tacc(2);
                filled = nn1;
                nr     = nn1;
              }
            if (onepass)
              {
                did_qr = true;
                qr_sweep(aq, nb, true);
#pragma unroll
                for (int sl = 0; sl < 2; ++sl)
                  {
                    const int j = grp + 16 * sl, cj = j < nn1 ? cix(j) : dsel;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                      {
                        const int r = l16 + 16 * i;
                        if (j <= nn1 && r < nn1)
                          BD[r * ncm + cj] = aq[sl][i];
                      }
                  }
                __syncthreads();
                tacc(2);
                nr = nn1;
              }
            if (did_qr && !(SLOD_DG(A, 512)))
              {
                if (!singular)
                  {
                    // R^{-1} by columns, Frobenius norms, d = -R^{-1} c.  An 8-lane group solves
                    // R x = e_j: the dot product of a back-substitution step is spread over the
                    // lanes (3 DPP steps), x goes through LDS (one wave: in order), so a column
                    // costs j short steps instead of j^2/2 dependent LDS round trips of one thread
                    double    fr = 0.0, fi = 0.0;
                    const int l8 = tid & 7;
                    for (int j = tid >> 3; j < nn1; j += 32)
                      {
                        const int cj = cix(j);
                        if (l8 == 0)
                          Vj[j * nn1 + j] = 1.0 / BD[j * ncm + cj];
                        for (int i = j - 1; i >= 0; --i)
                          {
                            const double rd = fast_rcp(BD[i * ncm + cix(i)]);
                            double       sa = 0.0;
                            for (int k2 = i + 1 + l8; k2 <= j; k2 += 8)
                              sa = fma(BD[i * ncm + cix(k2)], Vj[k2 * nn1 + j], sa);
                            sa += dpp_rot<0xB1>(sa);  // quad_perm [1,0,3,2]
                            sa += dpp_rot<0x4E>(sa);  // quad_perm [2,3,0,1]
                            sa += dpp_rot<0x141>(sa); // row_half_mirror: the other quad of the 8 lanes
                            if (l8 == 0)
                              Vj[i * nn1 + j] = -sa * rd;
                          }
                        for (int i = l8; i <= j; i += 8)
                          {
                            const double r = BD[i * ncm + cj], x = Vj[i * nn1 + j];
                            fr             = fma(r, r, fr);
                            fi             = fma(x, x, fi);
                          }
                      }
                    const double nr2 = block_sum(fr), ni2 = block_sum(fi);
                    double       del = 0.0;
                    if (tid < nn1)
                      {
                        for (int j = tid; j < nn1; ++j)
                          del = fma(-Vj[tid * nn1 + j], BD[j * ncm + dsel], del);
                        rowk[tid] = del;
                      }
                    double dmax = fabs(del);
                    for (int off = 32; off > 0; off >>= 1)
                      dmax = fmax(dmax, __shfl_xor(dmax, off, 64));
                    __syncthreads();
                    if (lane == 0)
                      red[4 + wave] = dmax;
                    __syncthreads();
                    const double dinf = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
                    pdg.dinf          = dinf;
                    if (nr2 * ni2 < 1e14 && dinf < 0.5 - 1e-9)
                      {
                        need_svd = false;
                        if (tid < nn1)
                          gam[cix(tid)] = rowk[tid];
                      }
                  }
              }
            if (stamping)
              {
                msd[5] = tph[0];
                msd[6] = tph[1];
                msd[7] = tph[2];
              }
            stamp(8);
            if (need_svd && !(SLOD_DG(A, (512 | 4096))))
              {
                pdg.path = 2;
                // ---- one-sided Jacobi SVD (the reference's singular triplets are needed).
                // After the QR the sweeps run on L = R^T (Drmac/Veselic: much faster convergence
                // than on R): L J = W with orthogonal columns w_j = sigma_j v_j (v_j = right
                // singular vectors of R = those of BD'), J = left singular vectors of R, so
                // sigma_j(G) = |w_j|^2, u_j^T g = sigma_j (J_j . c), and the reference's term
                // v_j (u_j^T g) / sigma_j(G) = w_j (J_j . c) / |w_j|^2.  Without a QR (fewer rows
                // than columns) the sweeps run on BD' itself: W = BD' V, term = V_j (w_j . b0)/|w_j|^2.
                const bool tposed = did_qr && nbuf >= 2 * nn1 + 1;
                double    *Wm     = BD;              // matrix whose columns are rotated
                int        wr     = nr;              // its rows
                int        pb     = 0;               // current buffer of the column order pcol
                for (int j = tid; j < nn1; j += 256)
                  pcol[j] = j;
                if (tposed && nc <= 32 && !(SLOD_DG(A, 65536)))
                  {
                    // Second-stage QR of R WITH column pivoting, R P = Q' R' (Drmac/Veselic
                    // preconditioning: the sweeps on R'^T converge in ~6 instead of ~10 sweeps).
                    // Register-resident like the first stage (group g: logical columns g, g+16, lane
                    // l16: rows l16, l16+16); the trailing columns are written through to LDS so that
                    // any of them can become the next pivot; norms and the order are double buffered.
                    // c = Q^T b0 (logical column nn1) takes the reflectors too: c' = Q'^T c.
                    double a[2][2];
                    bool   fin[2] = {false, false};
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl)
                      {
                        const int j = grp + 16 * sl, cj = j < nn1 ? cix(j) : dsel;
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                          {
                            const int r = l16 + 16 * i;
                            a[sl][i]    = (j <= nn1 && r < nn1) ? BD[r * ncm + cj] : 0.0;
                          }
                        const double nrm = group16_sum(fma(a[sl][0], a[sl][0], a[sl][1] * a[sl][1]));
                        if (j < nn1 && l16 == 0)
                          sig[j] = nrm;
                      }
                    __syncthreads();
                    for (int k = 0; k < nn1; ++k)
                      {
                        const int    *pc  = pcol + pb * ncm;
                        int          *pn  = pcol + (1 - pb) * ncm;
                        const double *cnc = (k & 1) ? utg : sig; // trailing column norms^2 (rows >= k)
                        double       *cnn = (k & 1) ? sig : utg;
                        int           p    = k;
                        double        best = cnc[pc[k]];
                        for (int j = k + 1; j < nn1; ++j)
                          {
                            const double v = cnc[pc[j]];
                            if (v > best)
                              {
                                best = v;
                                p    = j;
                              }
                          }
                        if (!(best > 0.0))
                          break; // the trailing block is zero (uniform decision)
                        const int cp = pc[p];
                        if (tid < nn1)
                          pn[tid] = (tid == k) ? cp : (tid == p ? pc[k] : pc[tid]);
                        const int    ccp   = cix(cp);
                        const double x0    = BD[k * ncm + ccp];
                        const double sq    = best * fast_rsqrt(best);
                        const double alpha = (x0 >= 0.0) ? -sq : sq;
                        const double v0    = x0 - alpha;
                        const double beta  = fast_rcp(best - alpha * x0);
                        double       vr[2];
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                          {
                            const int r = l16 + 16 * i;
                            vr[i]       = (r == k) ? v0 : ((r > k && r < nn1) ? BD[r * ncm + ccp] : 0.0);
                          }
#pragma unroll
                        for (int sl = 0; sl < 2; ++sl)
                          {
                            const int j = grp + 16 * sl, cj = j < nn1 ? cix(j) : dsel;
                            if (j > nn1 || fin[sl])
                              continue;
                            if (j == cp)
                              {
                                // finished: kept in registers (its LDS copy is still being read as the
                                // pivot column in this step) and written back after the loop
#pragma unroll
                                for (int i = 0; i < 2; ++i)
                                  {
                                    const int r = l16 + 16 * i;
                                    a[sl][i]    = (r == k) ? alpha : (r > k ? 0.0 : a[sl][i]);
                                  }
                                fin[sl] = true;
                              }
                            else
                              {
                                const double sd = group16_sum(fma(vr[0], a[sl][0], vr[1] * a[sl][1])) * beta;
                                double       nrm = 0.0;
#pragma unroll
                                for (int i = 0; i < 2; ++i)
                                  {
                                    const int r = l16 + 16 * i;
                                    a[sl][i]    = fma(-sd, vr[i], a[sl][i]);
                                    if (r > k)
                                      nrm = fma(a[sl][i], a[sl][i], nrm);
                                    if (r >= k && r < nn1)
                                      BD[r * ncm + cj] = a[sl][i];
                                  }
                                nrm = group16_sum(nrm);
                                if (j < nn1 && l16 == 0)
                                  cnn[j] = nrm;
                              }
                          }
                        __syncthreads();
                        pb ^= 1;
                      }
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl)
                      {
                        const int j = grp + 16 * sl, cj = j < nn1 ? cix(j) : dsel;
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                          {
                            const int r = l16 + 16 * i;
                            if (j <= nn1 && r < nn1)
                              BD[r * ncm + cj] = a[sl][i];
                          }
                      }
                  }
                __syncthreads();
                const int *pc = pcol + pb * ncm; // position -> column of BD'
                if (tposed)
                  {
                    Wm = BD + (size_t)nn1 * ncm;     // rows nn1..2nn1-1 of the buffer are free now
                    for (int idx = tid; idx < nn1 * nn1; idx += 256)
                      {
                        const int i = idx / nn1, j = idx - i * nn1;       // L[i][j] = R'[j][i]
                        Wm[i * ncm + j] = (j <= i) ? BD[j * ncm + cix(pc[i])] : 0.0;
                      }
                    // c'^T rides along as an extra row: it takes every rotation (c'^T J, whose j-th
                    // entry is J_j . c') but stays out of the dot products, so the rotations need
                    // not be accumulated in a second matrix
                    for (int j = tid; j < nn1; j += 256)
                      Wm[nn1 * ncm + j] = BD[j * ncm + dsel];
                    wr = nn1;
                  }
                const int wru = tposed ? wr + 1 : wr; // rows that take the rotations
                auto wcol = [&](int j) { return tposed ? j : cix(j); };
                const int nev = (nn1 + 1) & ~1;
                for (int idx = tid; idx < nn1 * nn1; idx += 256)
                  Vj[idx] = ((idx / nn1) == (idx % nn1)) ? 1.0 : 0.0;
                // Frobenius norm^2 (rotation invariant): columns below 1e-22 of it are numerically
                // zero -- seven orders under the reference's 1e-15 cutoff on sigma(G).  A pair of
                // two such noise columns is not rotated (it never passes the relative test and would
                // keep every sweep busy on rank-deficient rim patches); a noise column IS rotated
                // against a live one: skipping that pair leaves the live column's direction off by
                // |noise|^2/|live|^2 (up to 1e-7 for a triplet just above the cutoff), which
                // oversampling-3 corner patches amplified to 1e-6 .. 1e-4 in phi
                double fro = 0.0;
                for (int idx = tid; idx < wr * nn1; idx += 256)
                  {
                    const double w = Wm[(idx / nn1) * ncm + wcol(idx % nn1)];
                    fro            = fma(w, w, fro);
                  }
                const double tiny = 1e-22 * block_sum(fro);
                if (nev / 2 <= 16 && (SLOD_DG(A, 131072)))
                  {
                    // All nev/2 <= 16 column pairs of a round fit ONE wave (4 lanes per pair): wave 0
                    // runs the sweeps alone, LDS accesses of one wave execute in order, so a round
                    // needs no workgroup barrier at all (the round trip through four waves cost more
                    // than the rotation arithmetic).
                    if (wave == 0)
                      {
                        const int l4 = lane & 3, pr = lane >> 2;
                        for (int sweep = 0; sweep < ((SLOD_DG(A, 8192)) ? 3 : 40); ++sweep)
                          {
                            bool any = false;
                            for (int round = 0; round < nev - 1; ++round)
                              {
                                int pa, pb2;
                                if (pr == 0)
                                  {
                                    pa  = nev - 1;
                                    pb2 = round;
                                  }
                                else
                                  {
                                    pa  = round + pr;
                                    pa  = pa >= nev - 1 ? pa - (nev - 1) : pa;
                                    pb2 = round - pr;
                                    pb2 = pb2 < 0 ? pb2 + (nev - 1) : pb2;
                                  }
                                const bool valid = pr < nev / 2 && pa < nn1 && pb2 < nn1;
                                const int  p = valid ? (pa < pb2 ? pa : pb2) : 0, q = valid ? (pa < pb2 ? pb2 : pa) : 0;
                                const int  cp = wcol(p), cq = wcol(q);
                                double     app = 0, aqq = 0, apq = 0;
                                for (int r = l4; r < wr; r += 4)
                                  {
                                    const double wp = Wm[r * ncm + cp], wq = Wm[r * ncm + cq];
                                    app = fma(wp, wp, app);
                                    aqq = fma(wq, wq, aqq);
                                    apq = fma(wp, wq, apq);
                                  }
                                app += dpp_rot<0xB1>(app); // quad_perm [1,0,3,2]
                                aqq += dpp_rot<0xB1>(aqq);
                                apq += dpp_rot<0xB1>(apq);
                                app += dpp_rot<0x4E>(app); // quad_perm [2,3,0,1]
                                aqq += dpp_rot<0x4E>(aqq);
                                apq += dpp_rot<0x4E>(apq);
                                const bool rot = valid && !(apq == 0.0 || apq * apq <= 1e-30 * (app * aqq) ||
                                                            fmax(app, aqq) <= tiny);
                                if (rot)
                                  {
                                    const double dd = aqq - app;
                                    const double hh = fma(dd, dd, 4.0 * apq * apq);
                                    const double hy = hh * fast_rsqrt(hh); // sqrt(dd^2 + 4 apq^2)
                                    const double t  = (dd >= 0.0 ? 2.0 : -2.0) * apq * fast_rcp(fabs(dd) + hy);
                                    const double cs = fast_rsqrt(fma(t, t, 1.0)), sn = cs * t;
                                    for (int r = l4; r < wru; r += 4)
                                      {
                                        const double wp = Wm[r * ncm + cp], wq = Wm[r * ncm + cq];
                                        Wm[r * ncm + cp] = cs * wp - sn * wq;
                                        Wm[r * ncm + cq] = sn * wp + cs * wq;
                                      }
                                    for (int r = l4; r < (tposed ? 0 : nn1); r += 4)
                                      {
                                        const double vp = Vj[r * nn1 + p], vq = Vj[r * nn1 + q];
                                        Vj[r * nn1 + p] = cs * vp - sn * vq;
                                        Vj[r * nn1 + q] = sn * vp + cs * vq;
                                      }
                                    any = true;
                                  }
                              }
                            if (!__any(any))
                              break;
                          }
                      }
                    __syncthreads();
                  }
                else
                  for (int sweep = 0; sweep < ((SLOD_DG(A, 8192)) ? 3 : 40); ++sweep)
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
                                  pa = round + pr;
                                  pa = pa >= nev - 1 ? pa - (nev - 1) : pa;
                                  pb = round - pr;
                                  pb = pb < 0 ? pb + (nev - 1) : pb;
                                }
                              if (pa >= nn1 || pb >= nn1)
                                continue;
                              const int p = pa < pb ? pa : pb, q = pa < pb ? pb : pa;
                              const int cp = wcol(p), cq = wcol(q);
                              double    app = 0, aqq = 0, apq = 0;
                              for (int r = l16; r < wr; r += 16)
                                {
                                  const double wp = Wm[r * ncm + cp], wq = Wm[r * ncm + cq];
                                  app = fma(wp, wp, app);
                                  aqq = fma(wq, wq, aqq);
                                  apq = fma(wp, wq, apq);
                                }
                              app = group16_sum(app);
                              aqq = group16_sum(aqq);
                              apq = group16_sum(apq);
                              if (apq == 0.0 || apq * apq <= 1e-30 * (app * aqq) || fmax(app, aqq) <= tiny)
                                continue;
                              // t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)), zeta = (aqq-app)/(2 apq),
                              // written without the division by apq; c = 1/sqrt(1+t^2), s = c t
                              const double dd = aqq - app;
                              const double hh = fma(dd, dd, 4.0 * apq * apq);
                              const double hy = hh * fast_rsqrt(hh); // sqrt(dd^2 + 4 apq^2)
                              const double t  = (dd >= 0.0 ? 2.0 : -2.0) * apq * fast_rcp(fabs(dd) + hy);
                              const double cs = fast_rsqrt(fma(t, t, 1.0)), sn = cs * t;
                              for (int r = l16; r < wru; r += 16)
                                {
                                  const double wp = Wm[r * ncm + cp], wq = Wm[r * ncm + cq];
                                  Wm[r * ncm + cp] = cs * wp - sn * wq;
                                  Wm[r * ncm + cq] = sn * wp + cs * wq;
                                }
                              for (int r = l16; r < (tposed ? 0 : nn1); r += 16)
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
                      ++pdg.sweeps;
                      if (!any)
                        break;
                    }
                // sig_j = sigma_j(G); utg_j = coefficient of the j-th term's vector
                for (int j = tid; j < nn1; j += 256)
                  {
                    const int cj = wcol(j);
                    double    ss = 0, wb = 0;
                    for (int r = 0; r < wr; ++r)
                      {
                        const double w = Wm[r * ncm + cj];
                        ss             = fma(w, w, ss);
                        if (!tposed)
                          wb = fma(w, BD[r * ncm + dsel], wb);           // w_j . b0
                      }
                    if (tposed)
                      wb = Wm[nn1 * ncm + j]; // (c'^T J)_j = J_j . c'
                    sig[j] = ss;
                    utg[j] = wb;
                  }
                __syncthreads();
                // term vectors: V_j (no QR) or w_j (after the QR); element a2 of term j
                auto tvec = [&](int a2, int j) { return tposed ? Wm[a2 * ncm + j] : Vj[a2 * nn1 + j]; };
                // descending sigma (stable: ties keep the column order), pseudo-inverse cutoff
                // (LOD.cc:667): thread j ranks its own singular value
                for (int j = tid; j < nn1; j += 256)
                  {
                    const double sj = sig[j];
                    int          rank = 0;
                    for (int i = 0; i < nn1; ++i)
                      {
                        const double si = sig[i];
                        rank += (si > sj || (si == sj && i < j)) ? 1 : 0;
                      }
                    ord[rank] = j;
                  }
                __syncthreads();
                {
                  const double s0 = sig[ord[0]];
                  pdg.sigma_max   = s0;
                  pdg.sigma_min   = sig[ord[nn1 - 1]];
                  if (wave == 0) // nn1 <= 63: one ballot counts the cut singular values
                    pdg.n_cut = __popcll(__ballot(tid < nn1 && !(sig[min(tid, nn1 - 1)] > 1e-15 * s0)));
                  for (int j = tid; j < nn1; j += 256)
                    utg[j] = (sig[j] > 1e-15 * s0) ? utg[j] / sig[j] : 0.0;
                }
                __syncthreads();
                // d = -G^+ g (LOD.cc:669-671), one thread per component
                double del = 0.0;
                if (tid < nn1)
                  for (int j = 0; j < nn1; ++j)
                    del = fma(-tvec(tid, j), utg[j], del);
                // the 0.5-loop (LOD.cc:703-725): put the smallest remaining triplet back while
                // ||d||_inf >= 0.5 (the test precedes every removal)
                if (nn1 <= 64)
                  {
                    // all components live in wave 0: no workgroup barrier per removal
                    if (wave == 0)
                      {
                        for (int r = nn1 - 1; r >= 0; --r)
                          {
                            double dmax = (tid < nn1) ? fabs(del) : 0.0;
                            for (int off = 32; off > 0; off >>= 1)
                              dmax = fmax(dmax, __shfl_xor(dmax, off, 64));
                            pdg.dinf = dmax;
                            if (dmax < 0.5)
                              break;
                            const int j = ord[r];
                            if (tid < nn1)
                              del = fma(tvec(tid, j), utg[j], del);
                            ++pdg.n_dropped;
                          }
                        if (pdg.n_dropped == nn1) // every triplet removed: d = 0 (up to rounding)
                          {
                            double dmax = (tid < nn1) ? fabs(del) : 0.0;
                            for (int off = 32; off > 0; off >>= 1)
                              dmax = fmax(dmax, __shfl_xor(dmax, off, 64));
                            pdg.dinf = dmax;
                          }
                      }
                  }
                else
                  for (int r = nn1 - 1; r >= 0; --r)
                    {
                      double dmax = (tid < nn1) ? fabs(del) : 0.0;
                      for (int off = 32; off > 0; off >>= 1)
                        dmax = fmax(dmax, __shfl_xor(dmax, off, 64));
                      __syncthreads();
                      if (lane == 0)
                        red[4 + wave] = dmax;
                      __syncthreads();
                      const double dinf = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
                      pdg.dinf = dinf;
                      if (dinf < 0.5)
                        break;
                      const int j = ord[r];
                      if (tid < nn1)
                        del = fma(tvec(tid, j), utg[j], del);
                      ++pdg.n_dropped;
                    }
                if (tid < nn1)
                  gam[cix(pc[tid])] = del; // component of the pc[tid]-th column of BD'
              }
          }
        if (tid == 0 && A.pdiag)
          A.pdiag[(size_t)d.plan_index * S + dsel] = pdg;
        __syncthreads();
        stamp(9);
        // ---- c = D gamma (LOD.cc:727-743 / 576-577)
        for (int i = tid; i < nc; i += 256)
          {
            double acc = 0.0;
            for (int j = 0; j < nc; ++j)
              acc = fma(Ds[i * ldm + j], gam[j], acc);
            cvec[i] = acc;
          }
        __syncthreads();
        // ---- phi = X c, zero on the boundary (LOD.cc:745-750), l2-normalised (LOD.cc:752)
        double ssq = 0.0;
        for (int dof = tid; dof < ((SLOD_DG(A, 1024)) ? 0 : nf); dof += 256)
          {
            const int     node = dof / S, comp = dof - node * S;
            const int     ix = node % npx, iy = node / npx;
            const double *xr  = xrow(ix, iy, comp);
            double        acc = 0.0;
            if (xr)
              for (int j0 = 0; j0 < nc; j0 += 13) // 13 independent loads per batch
                {
                  double xv[13];
#pragma unroll
                  for (int e = 0; e < 13; ++e)
                    xv[e] = xr[min(j0 + e, nc - 1)];
#pragma unroll
                  for (int e = 0; e < 13; ++e)
                    acc = fma(xv[e], (j0 + e < nc) ? cvec[j0 + e] : 0.0, acc);
                }
            phis[dof] = acc;
            ssq       = fma(acc, acc, ssq);
          }
        const double nrm = sqrt(block_sum(ssq));
        double      *ob  = A.basis + d.out_off + (size_t)dsel * nf;
        double      *op  = A.premult + d.out_off + (size_t)dsel * nf;
        for (int dof = tid; dof < nf; dof += 256)
          {
            const double v = phis[dof] / nrm;
            phis[dof]      = v;
            ob[dof]        = v;
          }
        __syncthreads();
        stamp(10);
        // ---- psi = A_semi phi: identity rows on id-0 dofs (LOD.cc:537-541,758-765)
        for (int dof = tid; dof < ((SLOD_DG(A, 2048)) ? 0 : nf); dof += 256)
          {
            const int  node = dof / S, comp = dof - node * S;
            const int  ix = node % npx, iy = node / npx;
            const bool dom = (ix == 0 && (d.flags & 1)) || (ix == d.nx && (d.flags & 2)) ||
                             (iy == 0 && (d.flags & 4)) || (iy == d.ny && (d.flags & 8));
            // the 9 stencil loads do not depend on any branch: one batch per dof
            double sv[9][S];
#pragma unroll
            for (int dir = 0; dir < 9; ++dir)
#pragma unroll
              for (int cb = 0; cb < S; ++cb)
                sv[dir][cb] = st[(size_t)((dir * S + comp) * S + cb) * A.nn_max + node];
            double acc = 0.0;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
              for (int dx = -1; dx <= 1; ++dx)
                {
                  const int  jx = ix + dx, jy = iy + dy;
                  const bool in = !(jx < 0 || jx > d.nx || jy < 0 || jy > d.ny);
                  const int  jn = min(max(jx, 0), d.nx) + min(max(jy, 0), d.ny) * npx;
                  const int  dir = (dy + 1) * 3 + dx + 1;
#pragma unroll
                  for (int cb = 0; cb < S; ++cb)
                    acc = fma(in ? sv[dir][cb] : 0.0, phis[jn * S + cb], acc);
                }
            if (dom)
              acc = phis[dof];
            op[dof] = acc;
          }
        __syncthreads();
        stamp(11);
      }
  }
} // namespace

#endif
