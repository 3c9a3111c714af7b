// k_solve: cooperative block-tridiagonal patch solve (fallback, SLOD_SOLVE=coop).
#include "slod_common.hip.h"

namespace
{
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
  // each (entry (ty+16a, tx+16b)).  The symmetric Gauss-Jordan sweep needs only pivot row
  // k: the wave that owns it computes 1/pivot once and publishes the row r and the scaled
  // row s = r/pivot to a double-buffered LDS line, so a pivot step costs every thread
  // 2R LDS reads + R*R FMAs and ONE barrier.  V_l and Z_l go to the per-patch global
  // workspace for the backward pass (39 lines x 12 KB do not fit the 160 KB LDS).
  // The two GEMMs per line use R x 2 register tiles fed by 128-bit LDS reads.
  template <int R>
  __device__ __forceinline__ void gemm_tile(const double *__restrict__ Vs, int ldv,
                                            const double *__restrict__ Rb, int ncs, int m_even,
                                            int ty, int col0, double (&acc)[R][2])
  {
#pragma unroll
    for (int ra = 0; ra < R; ++ra)
      acc[ra][0] = acc[ra][1] = 0.0;
    const double *rp = Rb + col0;
    for (int k = 0; k < m_even; k += 2)
      {
        const double2 r0 = *reinterpret_cast<const double2 *>(rp + k * ncs);
        const double2 r1 = *reinterpret_cast<const double2 *>(rp + (k + 1) * ncs);
#pragma unroll
        for (int ra = 0; ra < R; ++ra)
          {
            const double2 v = *reinterpret_cast<const double2 *>(Vs + (ty + 16 * ra) * ldv + k);
            acc[ra][0]      = fma(v.x, r0.x, acc[ra][0]);
            acc[ra][1]      = fma(v.x, r0.y, acc[ra][1]);
            acc[ra][0]      = fma(v.y, r1.x, acc[ra][0]);
            acc[ra][1]      = fma(v.y, r1.y, acc[ra][1]);
          }
      }
  }

  // Twisted (two-sided) elimination: a 512-thread workgroup runs TWO chains in lockstep,
  // threads 0-255 eliminate lines 0,1,..,mid-1 downwards, threads 256-511 lines L-1,L-2,..,
  // mid+1 upwards; they meet at line mid, whose Schur complement takes a contribution from
  // both.  The backward substitution runs from mid outwards in both chains.  Same flops as
  // a one-sided sweep, half the number of dependent Gauss-Jordan steps / barriers per patch.
  // TW = 1: twisted, 512 threads (latency mode: few patches per CU).  TW = 0: one chain, 256
  // threads, lines 0..L-2 downwards and the "meeting line" L-1 without a second contribution
  // (throughput mode: the workgroup is half as big, twice as many patches are co-resident).
  template <int R, int S, int TW>
  __global__ __launch_bounds__(256 * (TW + 1), solve_min_waves(R)) void k_solve(const SlodKernelArgs A)
  {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const SlodPatchDesc d = A.desc[blockIdx.x];
    constexpr int       W = 2 * S - 1, BW = 2 * W + 1, NB = 16 * R, RBS = 2 * NB, NCH = TW + 1;
    const int           chain = TW ? (int)(threadIdx.x >> 8) : 0, tid = threadIdx.x & 255, ty = tid >> 4, tx = tid & 15;
    const int           m = d.m, L = d.L, nc = d.n_c, n = A.n_sub;
    const int           mm = A.m_max, ldv = (mm + 1) & ~1, m_even = (m + 1) & ~1;
    const int           ncs = (A.nc_max + 1) & ~1;
    const int           ngrp = (nc + kColGroup - 1) / kColGroup;
    const bool          tr  = (d.flags & SLOD_F_TRANSPOSED) != 0;
    const int           npx = d.nx + 1;
    const int           bsz = (mm * BW + 1) & ~1;
    const int           chsz = ldv * ldv + 2 * ldv * ncs + 3 * bsz; // doubles per chain

    // LDS carve-up (doubles); slod_solve_lds_bytes() mirrors it.  gemm_tile reads Vs rows up
    // to 16R-1 >= m: those land in the following arrays (results are discarded).
    double *cbase  = smem + chain * chsz;
    double *Vs     = cbase;                     // [ldv][ldv]  V of the line / U = V B, zero padded
    double *Rb     = Vs + ldv * ldv;            // [ldv][ncs]  right-hand side block
    double *Zp     = Rb + ldv * ncs;            // [ldv][ncs]  Z of the previous line of the chain
    double *Tb     = Zp + ldv * ncs;            // [mm][BW] band of T_line
    double *Bp     = Tb + bsz;                  // [mm][BW] coupling previous line -> this line
    double *Bn     = Bp + bsz;                  // [mm][BW] coupling this line -> next line
    double *rowbuf = smem + NCH * chsz + chain * 2 * RBS; // [2][RBS] published pivot rows
    int    *colk   = reinterpret_cast<int *>(smem + NCH * chsz + NCH * 2 * RBS); // [2][nc_max]
    double *oVs    = smem + (TW ? (1 - chain) : 0) * chsz; // the other chain's arrays
    double *oRb    = oVs + ldv * ldv;
    double *oZp    = oRb + ldv * ncs;

    const double *st    = A.st + (size_t)blockIdx.x * A.st_stride;
    double       *vg    = A.vinv + (size_t)blockIdx.x * A.v_stride;
    double       *xg    = A.xs + (size_t)blockIdx.x * A.x_stride;
    const int     ncg   = A.nc_max;
    const size_t  vline = (size_t)mm * mm, xline = (size_t)mm * ncg;

    const int mid  = TW ? L / 2 : L - 1;                // the line where the chains meet
    const int nmy  = chain == 0 ? mid : L - 1 - mid;   // lines of this chain
    const int nstp = mid > L - 1 - mid ? mid : L - 1 - mid;
    const int dl   = chain == 0 ? 1 : -1;              // direction of this chain

    for (int idx = tid; idx < ldv * ldv + 2 * ldv * ncs; idx += 256)
      cbase[idx] = 0.0;
    for (int c = threadIdx.x; c < nc; c += 256 * NCH)
      {
        int kx, ky;
        cell_of_col(d, c / S, kx, ky);
        colk[c]            = kx;
        colk[A.nc_max + c] = ky;
      }

    double a[R][R];

    // S = T - Bp^T U (U of the previous line of this chain is in Vs) -> a
    auto build_S = [&](bool have_prev) __attribute__((always_inline)) {
#pragma unroll
      for (int ra = 0; ra < R; ++ra)
        {
          const int i = ty + 16 * ra;
          double    bi[BW];
#pragma unroll
          for (int e = 0; e < BW; ++e)
            {
              const int p = i + e - W;
              bi[e] = (have_prev && i < m && p >= 0 && p < m && !(SLOD_DG(A, 1))) ? Bp[p * BW + (2 * W - e)] : 0.0;
            }
#pragma unroll
          for (int rb = 0; rb < R; ++rb)
            {
              const int j = tx + 16 * rb;
              double    v = 0.0;
              if (i < m && j < m)
                {
                  const int o = j - i;
                  if (o >= -W && o <= W)
                    v = Tb[i * BW + o + W];
#pragma unroll
                  for (int e = 0; e < BW; ++e)
                    {
                      const int pc = min(max(i + e - W, 0), m - 1);
                      v            = fma(-bi[e], Vs[pc * ldv + j], v);
                    }
                }
              else if (i == j && i == m && (m & 1))
                v = 1.0; // identity padding for the 2x2 block sweep
              a[ra][rb] = v;
            }
        }
    };
    // Rb = F_line - Bp^T Zp ; F = rows of P^T (LOD.cc:478-495)
    auto build_R = [&](int line, bool have_prev, bool with_F) __attribute__((always_inline)) {
      if (SLOD_DG(A, 2))
        return;
#pragma unroll
      for (int ra = 0; ra < R; ++ra)
        {
          const int i = ty + 16 * ra;
          if (i >= m)
            continue;
          const int pos = i / S, comp = i - pos * S;
          const int ix = tr ? line + 1 : pos + 1, iy = tr ? pos + 1 : line + 1;
          for (int r = tx; r < nc; r += 16)
            {
              double v = 0.0;
              if (with_F)
                {
                  const int jx = ix - colk[r] * n, jy = iy - colk[A.nc_max + r] * n;
                  if (jx >= 0 && jx <= n && jy >= 0 && jy <= n)
                    {
                      if (S == 1)
                        v = A.scale * (((jx == 0 || jx == n) ? 1.0 : 2.0) * ((jy == 0 || jy == n) ? 1.0 : 2.0));
                      else
                        v = A.scale * pt_weight<S>(d, n, A.quirk, ix, iy, comp, r);
                    }
                }
              if (have_prev)
                {
#pragma unroll
                  for (int e = 0; e < BW; ++e)
                    {
                      const int p = i + e - W;
                      if (p >= 0 && p < m)
                        v = fma(-Bp[p * BW + (2 * W - e)], Zp[p * ncs + r], v);
                    }
                }
              Rb[i * ncs + r] = v;
            }
        }
    };
    // symmetric Gauss-Jordan sweep with 2x2 block pivots: a <- -S^{-1}.  Rows k, k+1 are
    // published by their owner lanes, every thread inverts the 2x2 pivot block itself (one
    // Newton reciprocal) => ONE barrier per two pivots, shared by both chains.  Odd m is
    // padded with an identity row/column (build_S).
    auto gauss_jordan = [&](bool active) __attribute__((always_inline)) {
      // The pivot index is k = 16*ka + kt: unrolling over ka makes every register index of
      // the tile a compile-time constant (no dynamic selection, the tile stays in VGPRs).
      // Rows k, k+1 are published by their owner lanes; every thread inverts the 2x2 pivot
      // block itself (one Newton reciprocal): ONE barrier per two pivots.  (Having the owner
      // wave also publish the scaled rows saves instructions but lengthens the dependent
      // chain through that wave and measured slower.)
#pragma unroll
      for (int ka = 0; ka < R; ++ka)
        {
          const int kend = (SLOD_DG(A, 4)) ? (ka == 0 ? 2 : 0) : min(16, m_even - 16 * ka);
          for (int kt = 0; kt < kend; kt += 2)
            {
              const int k    = 16 * ka + kt;
              double   *row0 = rowbuf + ((k >> 1) & 1) * RBS, *row1 = row0 + NB;
              if (active && (ty == kt || ty == kt + 1))
                {
                  double *dst = (ty == kt) ? row0 : row1;
#pragma unroll
                  for (int rb = 0; rb < R; ++rb)
                    dst[tx + 16 * rb] = a[ka][rb];
                }
              __syncthreads();
              if (!active)
                continue;
              const double pa = row0[k], pb = row0[k + 1], pc = row1[k + 1];
              const double det = fma(pa, pc, -(pb * pb));
              if (tid == 0 && !(det > 0.0 && pa > 0.0) && !SLOD_DG(A, -1))
                atomicOr(A.status, 1);
              const double idet = fast_rcp(det);
              const double P00 = pc * idet, P01 = -pb * idet, P11 = pa * idet;
              double       ri0[R], ri1[R], s0[R], s1[R];
#pragma unroll
              for (int ra = 0; ra < R; ++ra)
                {
                  ri0[ra] = row0[ty + 16 * ra];
                  ri1[ra] = row1[ty + 16 * ra];
                }
#pragma unroll
              for (int rb = 0; rb < R; ++rb)
                {
                  const double rj0 = row0[tx + 16 * rb], rj1 = row1[tx + 16 * rb];
                  s0[rb]           = fma(P00, rj0, P01 * rj1);
                  s1[rb]           = fma(P01, rj0, P11 * rj1);
                }
#pragma unroll
              for (int ra = 0; ra < R; ++ra)
#pragma unroll
                for (int rb = 0; rb < R; ++rb)
                  a[ra][rb] = fma(-ri1[ra], s1[rb], fma(-ri0[ra], s0[rb], a[ra][rb]));
              if (ty == kt || ty == kt + 1) // rows k, k+1: P r_j
                {
#pragma unroll
                  for (int rb = 0; rb < R; ++rb)
                    a[ka][rb] = (ty == kt) ? s0[rb] : s1[rb];
                }
              if (tx == kt || tx == kt + 1) // columns k, k+1: P r_i; pivot block: -P
                {
#pragma unroll
                  for (int ra = 0; ra < R; ++ra)
                    {
                      const double t0 = fma(P00, ri0[ra], P01 * ri1[ra]);
                      const double t1 = fma(P01, ri0[ra], P11 * ri1[ra]);
                      a[ra][ka]       = (tx == kt) ? t0 : t1;
                    }
                  if (ty == kt)
                    a[ka][ka] = (tx == kt) ? -P00 : -P01;
                  if (ty == kt + 1)
                    a[ka][ka] = (tx == kt) ? -P01 : -P11;
                }
            }
        }
    };
    // V = -a -> Vs (GEMM, next Schur update) and global workspace (backward pass)
    auto store_V = [&](int line) __attribute__((always_inline)) {
      double *vl = vg + (size_t)line * vline; // wave-uniform base, 32-bit per-thread offsets
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
                if (!(SLOD_DG(A, 32)))
                  vl[i * mm + j] = v;
              }
          }
    };
    // Zp, X(line) <- (sub ? X(line) : 0) -/+ Vs * Rb
    auto gemm_store = [&](int line, bool sub) __attribute__((always_inline)) {
      double *xl = xg + (size_t)line * xline; // wave-uniform base, 32-bit per-thread offsets
      for (int g = 0; g < ngrp; ++g)
        {
          const int col0 = g * kColGroup + 2 * tx;
          if (col0 >= ncs)
            continue;
          double zl[R][2];
#pragma unroll
          for (int ra = 0; ra < R; ++ra)
#pragma unroll
            for (int c = 0; c < 2; ++c)
              {
                const int i = ty + 16 * ra;
                zl[ra][c]   = (sub && i < m && col0 + c < nc)
                                ? xl[i * ncg + col0 + c]
                                : 0.0;
              }
          double acc[R][2];
          gemm_tile<R>(Vs, ldv, Rb, ncs, m_even, ty, col0, acc);
#pragma unroll
          for (int ra = 0; ra < R; ++ra)
            {
              const int i = ty + 16 * ra;
              if (i < m)
                {
#pragma unroll
                  for (int c = 0; c < 2; ++c)
                    if (col0 + c < nc)
                      {
                        const double x = sub ? zl[ra][c] - acc[ra][c] : acc[ra][c];
                        xl[i * ncg + col0 + c] = x;
                        Zp[i * ncs + col0 + c]                                 = x;
                      }
                }
            }
        }
    };
    auto load_bands = [&](int line, bool want_T, bool want_next) __attribute__((always_inline)) {
      for (int idx = tid; idx < m * BW; idx += 256)
        {
          const int i = idx / BW, o = idx - i * BW - W;
          if (want_T)
            Tb[idx] = coupling<S>(st, A.nn_max, npx, tr, m, line, i, 0, o);
          Bn[idx] = want_next ? coupling<S>(st, A.nn_max, npx, tr, m, line, i, dl, o) : 0.0;
        }
    };

    // ------------------------------ forward elimination ---------------------------
    for (int t = 0; t < nstp; ++t)
      {
        const bool active = t < nmy;
        const int  line   = chain == 0 ? t : L - 1 - t;
        if (active)
          load_bands(line, true, true);
        __syncthreads();
        if (active)
          {
            build_S(t > 0);
            build_R(line, t > 0, true);
          }
        __syncthreads();
        gauss_jordan(active);
        if (active)
          store_V(line);
        __syncthreads();
        double u[R][R];
        if (active)
          {
            if (!(SLOD_DG(A, 8)))
              gemm_store(line, false);
            // U = V Bn replaces V in Vs (only the next Schur update of this chain reads it)
            if (!(SLOD_DG(A, 1)))
              {
#pragma unroll
                for (int rb = 0; rb < R; ++rb)
                  {
                    const int j = tx + 16 * rb;
                    double    bj[BW];
#pragma unroll
                    for (int f = 0; f < BW; ++f)
                      {
                        const int q = j + f - W;
                        bj[f]       = (j < m && q >= 0 && q < m) ? Bn[q * BW + (2 * W - f)] : 0.0;
                      }
#pragma unroll
                    for (int ra = 0; ra < R; ++ra)
                      {
                        const int i   = ty + 16 * ra;
                        double    acc = 0.0;
                        if (i < m && j < m)
                          {
#pragma unroll
                            for (int f = 0; f < BW; ++f)
                              {
                                const int qc = min(max(j + f - W, 0), m - 1);
                                acc          = fma(Vs[i * ldv + qc], bj[f], acc);
                              }
                          }
                        u[ra][rb] = acc;
                      }
                  }
              }
          }
        __syncthreads(); // every read of V (GEMM, U tiles) is done
        if (active && !(SLOD_DG(A, 1)))
          {
#pragma unroll
            for (int ra = 0; ra < R; ++ra)
#pragma unroll
              for (int rb = 0; rb < R; ++rb)
                {
                  const int i = ty + 16 * ra, j = tx + 16 * rb;
                  if (i < m && j < m)
                    Vs[i * ldv + j] = u[ra][rb];
                }
          }
        if (active)
          {
            double *tmp = Bp;
            Bp          = Bn;
            Bn          = tmp;
          }
      }

    // ------------------------------ the meeting line ------------------------------
    // S_mid = T_mid - B^T U (from above) - B U' (from below); chain 1 hands its two
    // contributions over through its own Vs / Rb.
    {
      const bool have0 = mid > 0, have1 = TW && (L - 1 - mid > 0);
      if (chain == 0)
        load_bands(mid, true, false);
      __syncthreads();
      if (chain == 0)
        {
          build_S(have0);
          build_R(mid, have0, true);
        }
      else if (have1)
        {
          // a <- -(contribution of the lower chain) via build_S with T = 0
          for (int idx = tid; idx < m * BW; idx += 256)
            Tb[idx] = 0.0;
        }
      __syncthreads();
      if (chain == 1 && have1)
        {
          build_S(true);              // a = -Bp^T U'
          build_R(mid, true, false);  // Rb = -Bp^T Z'
        }
      __syncthreads();
      if (chain == 1 && have1)
        {
#pragma unroll
          for (int ra = 0; ra < R; ++ra)
#pragma unroll
            for (int rb = 0; rb < R; ++rb)
              {
                const int i = ty + 16 * ra, j = tx + 16 * rb;
                if (i < m && j < m)
                  Vs[i * ldv + j] = a[ra][rb];
              }
        }
      __syncthreads();
      if (chain == 0 && have1)
        {
#pragma unroll
          for (int ra = 0; ra < R; ++ra)
            {
              const int i = ty + 16 * ra;
#pragma unroll
              for (int rb = 0; rb < R; ++rb)
                {
                  const int j = tx + 16 * rb;
                  if (i < m && j < m)
                    a[ra][rb] += oVs[i * ldv + j];
                }
              if (i < m)
                for (int r = tx; r < nc; r += 16)
                  Rb[i * ncs + r] += oRb[i * ncs + r];
            }
        }
      __syncthreads();
      gauss_jordan(chain == 0);
      if (chain == 0)
        store_V(mid);
      __syncthreads();
      if (chain == 0)
        gemm_store(mid, false); // X_mid = Z_mid -> Zp (chain 0), global
      __syncthreads();
      if (chain == 1)
        for (int idx = tid; idx < ldv * ncs; idx += 256)
          Zp[idx] = oZp[idx];
    }
    __syncthreads();

    // ------------------------------ backward substitution -------------------------
    // Zp holds X of the line processed before (mid at the start) in both chains.  V of the
    // next line to process is prefetched from the workspace one step ahead.
    double vpre[R][R];
    auto   prefetch_V = [&](int line) __attribute__((always_inline)) {
      const double *vl = vg + (size_t)line * vline;
#pragma unroll
      for (int ra = 0; ra < R; ++ra)
#pragma unroll
        for (int rb = 0; rb < R; ++rb)
          {
            const int i = ty + 16 * ra, j = tx + 16 * rb;
            vpre[ra][rb] = (i < m && j < m) ? vl[i * mm + j] : 0.0;
          }
    };
    if (nmy > 0 && !(SLOD_DG(A, 16)))
      prefetch_V(chain == 0 ? nmy - 1 : L - nmy);
    for (int t = (SLOD_DG(A, 16)) ? -1 : nstp - 1; t >= 0; --t)
      {
        const bool active = t < nmy;
        const int  line   = chain == 0 ? t : L - 1 - t;
        if (active)
          {
            // coupling of this line with the line solved just before (line + dl)
            load_bands(line, false, true);
#pragma unroll
            for (int ra = 0; ra < R; ++ra)
#pragma unroll
              for (int rb = 0; rb < R; ++rb)
                {
                  const int i = ty + 16 * ra, j = tx + 16 * rb;
                  if (i < m && j < m)
                    Vs[i * ldv + j] = vpre[ra][rb];
                }
            if (t > 0)
              prefetch_V(chain == 0 ? t - 1 : L - t);
          }
        __syncthreads();
        if (active)
          {
            // Y = Bn X(line + dl)
#pragma unroll
            for (int ra = 0; ra < R; ++ra)
              {
                const int i = ty + 16 * ra;
                if (i >= m)
                  continue;
                for (int r = tx; r < nc; r += 16)
                  {
                    double v = 0.0;
#pragma unroll
                    for (int o = 0; o < BW; ++o)
                      {
                        const int p = i + o - W;
                        if (p >= 0 && p < m)
                          v = fma(Bn[i * BW + o], Zp[p * ncs + r], v);
                      }
                    Rb[i * ncs + r] = v;
                  }
              }
          }
        __syncthreads();
        if (active)
          gemm_store(line, true); // X_line = Z_line - V_line Y
        __syncthreads();
      }
  }
} // namespace

size_t slod_solve_lds_bytes(int S, int m_max, int nc_max, int twisted)
{
  // must mirror the carve-up at the top of k_solve
  const int    R = (m_max + 15) / 16, BW = 2 * (2 * S - 1) + 1, NB = 16 * R, RBS = 2 * NB;
  const int    ldv = (m_max + 1) & ~1, ncs = (nc_max + 1) & ~1, bsz = (m_max * BW + 1) & ~1;
  const int    nch = twisted ? 2 : 1;
  const size_t chsz = (size_t)ldv * ldv + 2 * (size_t)ldv * ncs + 3 * (size_t)bsz;
  size_t       bytes = (nch * chsz + nch * 2 * RBS) * sizeof(double) + 2 * (size_t)nc_max * sizeof(int);
  // gemm_tile over-reads Vs rows up to NB-1 (results discarded): keep them inside the block
  const size_t over = ((nch - 1) * chsz + (size_t)NB * ldv + ldv) * sizeof(double);
  bytes             = bytes > over ? bytes : over;
  return (bytes + 15) & ~(size_t)15;
}

template <int R, int S, int TW>
static hipError_t launch_solve_RST(const SlodKernelArgs &a, int n_patches, size_t lds, hipStream_t st)
{
  const void *fn = reinterpret_cast<const void *>(k_solve<R, S, TW>);
  hipError_t  e  = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess)
    return e;
  if (a.debug)
    {
      int nb = 0;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256 * (TW + 1), lds);
      fprintf(stderr, "[slod] k_solve<%d,%d,%d>: %d patches, lds %zu B, occupancy %d blocks/CU\n", R, S, TW,
              n_patches, lds, nb);
    }
  hipLaunchKernelGGL((k_solve<R, S, TW>), dim3(n_patches), dim3(256 * (TW + 1)), lds, st, a);
  return hipGetLastError();
}

template <int R>
static hipError_t launch_solve_R(int S, int tw, const SlodKernelArgs &a, int n_patches, hipStream_t st)
{
  const size_t lds = slod_solve_lds_bytes(S, a.m_max, a.nc_max, tw);
  if (S == 1)
    return tw ? launch_solve_RST<R, 1, 1>(a, n_patches, lds, st) : launch_solve_RST<R, 1, 0>(a, n_patches, lds, st);
  return tw ? launch_solve_RST<R, 2, 1>(a, n_patches, lds, st) : launch_solve_RST<R, 2, 0>(a, n_patches, lds, st);
}

hipError_t slod_launch_solve_coop(int S, int tw, const SlodKernelArgs &a, int n_patches, hipStream_t st)
{
  switch ((a.m_max + 15) / 16)
    {
      case 1:
        return launch_solve_R<1>(S, tw, a, n_patches, st);
      case 2:
        return launch_solve_R<2>(S, tw, a, n_patches, st);
      case 3:
        return launch_solve_R<3>(S, tw, a, n_patches, st);
      case 4:
        return launch_solve_R<4>(S, tw, a, n_patches, st);
      case 5:
        return launch_solve_R<5>(S, tw, a, n_patches, st);
      case 6:
        return launch_solve_R<6>(S, tw, a, n_patches, st);
      case 7:
        return launch_solve_R<7>(S, tw, a, n_patches, st);
      default:
        return hipErrorInvalidValue;
    }
}
