// k_solve_ws: wave-specialised single-chain patch solve (SLOD_SOLVE=ws).
#include "slod_common.hip.h"

namespace
{
  // ---------------------------------------------------------------------------------
  // K2 (wave-specialised forward sweep).  The Gauss-Jordan inversion of a line's Schur
  // complement is a chain of m dependent pivot steps; spread over 256 threads every step
  // costs a workgroup barrier and ~7 instructions per useful FMA.  Here ONE wave holds the
  // whole m x m matrix in registers (8x8 lane grid x TxT contiguous tile, m <= 8T), publishes
  // pivot row k to a wave-private LDS line (in-order DS queue of one wave: no barrier) and
  // runs all m steps alone, while the other three waves of the workgroup build the
  // right-hand-side block and do the GEMM Z_l = V_l R_l of the line just inverted on the fp64
  // matrix pipe (v_mfma_f64_16x16x4_f64), which leaves the VALU issue slots to the GJ waves.  Three
  // workgroup barriers per line (not per pivot):
  //     A_l : V_l is in LDS (Vs), Z_{l-1} is in Zp, bands of the next stage are loaded
  //     C_l : helpers have built R_l                     (GJ wave: after its first steps)
  //     B_l : helpers are done with Vs (Z_l is in Zp)    (GJ wave: after its last step)
  // then the GJ wave overwrites Vs with V_{l+1}.  The backward substitution uses all 4 waves.
  // ---------------------------------------------------------------------------------
  template <int T, int S>
  __global__ __launch_bounds__(256, ws_min_waves(T)) void k_solve_ws(const SlodKernelArgs A)
  {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const SlodPatchDesc d = A.desc[blockIdx.x];
    constexpr int       W = 2 * S - 1, BW = 2 * W + 1, MP = 8 * T, PF = (MP * MP + 255) / 256;
    constexpr int       BWP = BW + 1, BROWS = MP + 2 * W; // zero-padded bands: no range predicates
    const int           tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int           m = d.m, L = d.L, nc = d.n_c, n = A.n_sub;
    const int           mm = A.m_max, ldv = MP + 2;
    const int           ncs = (A.nc_max + 1) & ~1;
    const bool          tr  = (d.flags & SLOD_F_TRANSPOSED) != 0;
    const int           npx = d.nx + 1;
    constexpr int       bsz = (BROWS * BWP + 1) & ~1;

    // LDS carve-up (doubles); slod_solve_ws_lds_bytes() mirrors it.  Rows m..ldv-1 of Vs/Rb/Zp
    // stay zero (k-loop padding of the GEMMs); the +2 rows of Vs keep the GEMM's 3-row tile
    // reads of the last row group inside the array.
    double *Vs   = smem;                 // [ldv][ldv]  V of the line being consumed
    double *Rb   = Vs + ldv * ldv;       // [ldv][ncs]  right-hand side block
    double *Zp   = Rb + ldv * ncs;       // [ldv][ncs]  Z of the previous line / X of the next
    double *rowb = Zp + ldv * ncs;       // [MP]        pivot row of the GJ wave
    double *Tn   = rowb + 2 * MP;        // padded band of T_{l+1}
    double *Bc0  = Tn + bsz;             // [mm][BW]    coupling bands, alternating
    double *Bc1  = Bc0 + bsz;
    int    *colk = reinterpret_cast<int *>(Bc1 + bsz); // [2][nc_max]

    const double *st    = A.st + (size_t)blockIdx.x * A.st_stride;
    double       *vg    = A.vinv + (size_t)blockIdx.x * A.v_stride;
    double       *xg    = A.xs + (size_t)blockIdx.x * A.x_stride;
    const int     ncg   = A.nc_max;
    const size_t  vline = (size_t)MP * MP, xline = (size_t)mm * ncg;

    for (int idx = tid; idx < ldv * ldv + 2 * ldv * ncs + 2 * MP + 3 * bsz; idx += 256)
      smem[idx] = 0.0;
    __syncthreads();
    for (int c = tid; c < nc; c += 256)
      {
        int kx, ky;
        cell_of_col(d, c / S, kx, ky);
        colk[c]            = kx;
        colk[A.nc_max + c] = ky;
      }
    // bands of line `line`: T (within the line) and B (line -> line+1)
    auto load_bands = [&](int line, double *Tdst, double *Bdst, int t0, int nt) __attribute__((always_inline)) {
      for (int idx = t0; idx < m * BW; idx += nt)
        {
          const int i = idx / BW, oi = idx - i * BW, o = oi - W;
          if (Tdst)
            Tdst[(i + W) * BWP + oi] = coupling<S>(st, A.nn_max, npx, tr, m, line, i, 0, o);
          if (Bdst)
            Bdst[(i + W) * BWP + oi] = (line + 1 < L) ? coupling<S>(st, A.nn_max, npx, tr, m, line, i, 1, o) : 0.0;
        }
      // identity on the padding rows of T (the 2x2 block sweep may pivot on index m, m odd);
      // the coupling buffers are zero there (Bc1 is used for T_0 first)
      for (int i = m + t0; i < MP; i += nt)
        {
          if (Tdst)
            Tdst[(i + W) * BWP + W] = 1.0;
          if (Bdst)
            Bdst[(i + W) * BWP + W] = 0.0;
        }
    };
    // prologue: T_0 goes to Bc1 (free until B_1 is loaded), T_1 to Tn, B_0 to Bc0
    load_bands(0, Bc1, Bc0, tid, 256);
    if (L > 1)
      load_bands(1, Tn, nullptr, tid, 256);
    __syncthreads();

    // ------------------------------ forward elimination ---------------------------
    // Barrier schedule per line l (all four waves):
    //   C_{l-1}: R_{l-1} built by the helpers        (GJ wave: in the middle of sweep(l))
    //   B'_l   : helpers are done reading Vs (GEMM of line l-1), bands T_{l+1}, B_l are loaded
    //   A_l    : V_l has been written to Vs by the GJ wave
    if (wave == 0)
      {
        // ===== the Gauss-Jordan wave =====
        // it is the critical path of the workgroup and shares its SIMD with helper waves of
        // other workgroups: win the issue arbitration
        __builtin_amdgcn_s_setprio(3);
        const int gy = lane >> 3, gx = lane & 7;
        double    a[T][T];
        // a <- S_{l+1} = Tsrc - Bl^T V_l Bl with V_l = -a, entirely in registers: the tile
        // neighbours in j come from lanes +-1, in i from lanes +-8 (needs T >= W)
        auto next_S = [&](const double *Tsrc, const double *Bl) __attribute__((always_inline)) {
          // U = V Bl, row by row in place.  Coefficients are re-read from the (zero padded)
          // LDS bands instead of being kept in registers: the tile alone is 2*T*T VGPRs.
          const double *cb = Bl + (T * gx) * BWP + 2 * W; // B_l[q][j] = cb[(tb + f) * BWP - f]
#pragma unroll
          for (int ta = 0; ta < T; ++ta)
            {
              double ext[T + 2 * W];
#pragma unroll
              for (int w = 0; w < W; ++w)
                {
                  ext[w]         = -__shfl(a[ta][T - W + w], lane - 1, 64);
                  ext[W + T + w] = -__shfl(a[ta][w], lane + 1, 64);
                }
#pragma unroll
              for (int tb = 0; tb < T; ++tb)
                ext[W + tb] = -a[ta][tb];
#pragma unroll
              for (int tb = 0; tb < T; ++tb)
                {
                  double acc = 0.0;
#pragma unroll
                  for (int f = 0; f < BW; ++f)
                    acc = fma(ext[tb + f], cb[(tb + f) * BWP - f], acc);
                  a[ta][tb] = acc;
                }
              __builtin_amdgcn_sched_barrier(0); // keep the shuffles of one row together (VGPR pressure)
            }
          // S = Tsrc - Bl^T U, column by column in place
          const double *db = Bl + (T * gy) * BWP + 2 * W; // B_l[p][i] = db[(ta + e) * BWP - e]
#pragma unroll
          for (int tb = 0; tb < T; ++tb)
            {
              const int j = T * gx + tb;
              double    ext[T + 2 * W];
#pragma unroll
              for (int w = 0; w < W; ++w)
                {
                  ext[w]         = __shfl(a[T - W + w][tb], lane - 8, 64);
                  ext[W + T + w] = __shfl(a[w][tb], lane + 8, 64);
                }
#pragma unroll
              for (int ta = 0; ta < T; ++ta)
                ext[W + ta] = a[ta][tb];
#pragma unroll
              for (int ta = 0; ta < T; ++ta)
                {
                  const int      i  = T * gy + ta;
                  const unsigned oi = (unsigned)(j - i + W); // outside the band -> zero pad column
                  double         acc = Tsrc[(i + W) * BWP + (oi < (unsigned)BW ? oi : (unsigned)BW)];
#pragma unroll
                  for (int e = 0; e < BW; ++e)
                    acc = fma(-ext[ta + e], db[(ta + e) * BWP - e], acc);
                  a[ta][tb] = acc;
                }
              __builtin_amdgcn_sched_barrier(0);
            }
        };
        // pivots [k0, k1) of the symmetric sweep a <- -S^{-1}; rows are published to rowb
        // (wave-private: the DS queue of one wave is in order, no workgroup barrier needed).
        // (A 2x2 block-pivot variant halves the LDS round trips but needs 20 more live
        // doubles; with the 128-VGPR budget of 4 workgroups/CU it measured slower.)
        bool bad = false;
        auto sweep = [&](int k0, int k1) __attribute__((always_inline)) {
          for (int ka = k0 / T; ka * T < k1; ++ka)
            {
#pragma unroll
              for (int a0 = 0; a0 < T; ++a0)
                {
                  const int k = T * ka + a0;
                  if (k < k0 || k >= k1 || ((SLOD_DG(A, 4)) && k > 0)) // wave-uniform
                    continue;
                  if (gy == ka)
                    {
#pragma unroll
                      for (int tb = 0; tb < T; ++tb)
                        rowb[T * gx + tb] = a[a0][tb];
                    }
                  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                  __builtin_amdgcn_wave_barrier();
                  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                  const double piv = rowb[k];
                  bad |= !(piv > 0.0);
                  const double p = fast_rcp(piv);
                  double       ri[T], sj[T];
#pragma unroll
                  for (int ta = 0; ta < T; ++ta)
                    ri[ta] = rowb[T * gy + ta];
#pragma unroll
                  for (int tb = 0; tb < T; ++tb)
                    sj[tb] = rowb[T * gx + tb] * p;
                  __builtin_amdgcn_wave_barrier(); // every lane has read row k before it is overwritten
#pragma unroll
                  for (int ta = 0; ta < T; ++ta)
#pragma unroll
                    for (int tb = 0; tb < T; ++tb)
                      a[ta][tb] = fma(-ri[ta], sj[tb], a[ta][tb]);
                  if (gy == ka) // row k: r_j / pivot
                    {
#pragma unroll
                      for (int tb = 0; tb < T; ++tb)
                        a[a0][tb] = sj[tb];
                    }
                  if (gx == ka) // column k: r_i / pivot; (k,k): -1/pivot
                    {
#pragma unroll
                      for (int ta = 0; ta < T; ++ta)
                        a[ta][a0] = ri[ta] * p;
                      if (gy == ka)
                        a[a0][a0] = -p;
                    }
                  __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        auto store_V = [&](int line) __attribute__((always_inline)) {
          double *vl = vg + (size_t)line * vline;
#pragma unroll
          for (int ta = 0; ta < T; ++ta)
#pragma unroll
            for (int tb = 0; tb < T; ++tb)
              {
                const int    i = T * gy + ta, j = T * gx + tb;
                const double v = -a[ta][tb]; // zero outside m x m
                Vs[i * ldv + j] = v;
                vl[i * MP + j]  = v;
              }
        };
        const int ksplit = m / 3, ksplit2 = (2 * m) / 3; // barriers C, D are taken inside the sweep
        // S_0 = T_0 (held in Bc1 during the prologue)
#pragma unroll
        for (int ta = 0; ta < T; ++ta)
#pragma unroll
          for (int tb = 0; tb < T; ++tb)
            {
              const int      i = T * gy + ta, j = T * gx + tb;
              const unsigned oi = (unsigned)(j - i + W);
              a[ta][tb]         = Bc1[(i + W) * BWP + (oi < (unsigned)BW ? oi : (unsigned)BW)];
            }
        for (int l = 0; l < L; ++l)
          {
            sweep(0, ksplit);
            if (l > 0)
              __syncthreads(); // C_{l-1}
            sweep(ksplit, ksplit2);
            if (l > 0 && A.m_fused)
              __syncthreads(); // D_{l-1}
            sweep(ksplit2, m);
            if (bad && lane == 0 && !SLOD_DG(A, -1))
              atomicOr(A.status, 1);
            __syncthreads(); // B'_l
            if (!(SLOD_DG(A, 32768)))
              store_V(l);
            __syncthreads(); // A_l
            if (l + 1 < L && !(SLOD_DG(A, 16384)))
              next_S(Tn, (l & 1) ? Bc1 : Bc0);
          }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads(); // C_{L-1}
        if (A.m_fused)
          __syncthreads(); // D_{L-1}
        __syncthreads(); // end of the forward sweep
      }
    else
      {
        // ===== the three helper waves =====
        const int hid = tid - 64; // 0..191
        const int hr = hid >> 5, hc = hid & 31; // RHS build: 6 rows x 32 columns per pass
        // R_l = F_l - B_{l-1}^T Z_{l-1}; F = rows of P^T (LOD.cc:478-495)
        auto build_R = [&](int l) __attribute__((always_inline)) {
          const double *Bprev = (l & 1) ? Bc0 : Bc1; // coupling l-1 -> l
          for (int i = hr; i < ((SLOD_DG(A, 2)) ? 0 : m); i += 6)
            {
              const int pos = i / S, comp = i - pos * S;
              const int ix = tr ? l + 1 : pos + 1, iy = tr ? pos + 1 : l + 1;
              for (int r = hc; r < nc; r += 32)
                {
                  const int jx = ix - colk[r] * n, jy = iy - colk[A.nc_max + r] * n;
                  double    v  = 0.0;
                  if (jx >= 0 && jx <= n && jy >= 0 && jy <= n)
                    {
                      if (S == 1)
                        v = A.scale * (((jx == 0 || jx == n) ? 1.0 : 2.0) * ((jy == 0 || jy == n) ? 1.0 : 2.0));
                      else
                        v = A.scale * pt_weight<S>(d, n, A.quirk, ix, iy, comp, r);
                    }
                  if (l > 0)
                    {
#pragma unroll
                      for (int e = 0; e < BW; ++e)
                        {
                          const int p = i + e - W;
                          if (p >= 0 && p < m)
                            v = fma(-Bprev[(p + W) * BWP + (2 * W - e)], Zp[p * ncs + r], v);
                        }
                    }
                  Rb[i * ncs + r] = v;
                }
            }
        };
        // Z_l = V_l R_l -> Zp, workspace; 16x16 output tiles dealt to the three helper waves
        const int k4 = (m + 3) & ~3, tiles_i = (m + 15) >> 4, tiles_j = (nc + 15) >> 4;
        auto gemm_Z = [&](int l) __attribute__((always_inline)) {
          if (SLOD_DG(A, 8))
            return;
          double *xl = xg + (size_t)l * xline;
          for (int t = wave - 1; t < tiles_i * tiles_j; t += 3)
            {
              const int       ti = t / tiles_j, tj = t - ti * tiles_j;
              const double4_t acc = gemm_mfma_tile(Vs, ldv, Rb, ncs, k4, ti, tj, lane);
              const int       col = 16 * tj + (lane & 15);
#pragma unroll
              for (int r = 0; r < 4; ++r)
                {
                  const int row = 16 * ti + (lane >> 4) + 4 * r;
                  if (row < m && col < nc)
                    {
                      Zp[row * ncs + col] = acc[r];
                      xl[row * ncg + col] = acc[r];
                    }
                }
            }
        };
        // M = P^T A^-1 P / H^2 = sum_l R_l^T Z_l (block LDL^T identity, LOD.cc:548-551): every
        // helper thread owns up to MA entries; accumulated while the GJ wave sweeps
        constexpr int MA = 4; // nc^2 <= 768 (nc <= 27); larger patches let k_select compute M from X
        double        macc[MA];
        int           mab[MA]; // a + 64 * b, or -1
#pragma unroll
        for (int q = 0; q < MA; ++q)
          {
            const int idx = hid + 192 * q;
            macc[q]       = 0.0;
            mab[q]        = (A.m_fused && idx < nc * nc) ? (idx / nc) + 64 * (idx % nc) : -1;
          }
        auto accumulate_M = [&]() __attribute__((always_inline)) {
#pragma unroll
          for (int q = 0; q < MA; ++q)
            if (mab[q] >= 0)
              {
                const int ca = mab[q] & 63, cb = mab[q] >> 6;
                double    acc = macc[q];
                for (int i = 0; i < m; ++i)
                  acc = fma(Rb[i * ncs + ca], Zp[i * ncs + cb], acc);
                macc[q] = acc;
              }
        };
        // The stencil entries of the next bands come from global memory (L2/HBM latency):
        // they are fetched into registers at the top of an iteration and written to LDS after
        // the GEMM, so the latency is off the path to barrier B'.
        constexpr int NBV = (MP * BW + 191) / 192;
        double        tband[NBV], bband[NBV];
        auto fetch_bands = [&](int l) __attribute__((always_inline)) {
#pragma unroll
          for (int q = 0; q < NBV; ++q)
            {
              const int idx = hid + 192 * q;
              const int i = idx / BW, o = idx - i * BW - W;
              const bool in = idx < m * BW && !(SLOD_DG(A, 32));
              tband[q] = (in && l + 1 < L) ? coupling<S>(st, A.nn_max, npx, tr, m, l + 1, i, 0, o) : 0.0;
              bband[q] = (in && l + 1 < L) ? coupling<S>(st, A.nn_max, npx, tr, m, l, i, 1, o) : 0.0;
            }
        };
        auto store_bands = [&](int l) __attribute__((always_inline)) {
          double *Bdst = (l & 1) ? Bc1 : Bc0;
#pragma unroll
          for (int q = 0; q < NBV; ++q)
            {
              const int idx = hid + 192 * q;
              const int i = idx / BW, oi = idx - i * BW;
              if (idx < m * BW)
                {
                  if (l + 1 < L)
                    Tn[(i + W) * BWP + oi] = tband[q];
                  Bdst[(i + W) * BWP + oi] = bband[q];
                }
            }
          for (int i = m + hid; i < MP; i += 192) // padding rows: identity in T, zero in B
            {
              Tn[(i + W) * BWP + W]   = 1.0;
              Bdst[(i + W) * BWP + W] = 0.0;
            }
        };
        for (int l = 0; l < L; ++l)
          {
            if (l > 0)
              {
                fetch_bands(l);
                build_R(l - 1);
                __syncthreads(); // C_{l-1}: R_{l-1} complete, every read of Z_{l-2} is done
                gemm_Z(l - 1);
                if (A.m_fused)
                  {
                    __syncthreads(); // D_{l-1}: Z_{l-1} complete
                    accumulate_M();
                  }
                // bands the GJ wave needs after A_l: T_{l+1}, B_l (fetched before the GEMM)
                store_bands(l);
              }
            __syncthreads(); // B'_l
            __syncthreads(); // A_l
          }
        build_R(L - 1);
        __syncthreads(); // C_{L-1}
        gemm_Z(L - 1);
        if (A.m_fused)
          {
            __syncthreads(); // D_{L-1}
            accumulate_M();
            double *mg = A.ms + (size_t)blockIdx.x * A.nc_max * A.nc_max;
#pragma unroll
            for (int q = 0; q < MA; ++q)
              if (mab[q] >= 0)
                mg[(mab[q] & 63) * nc + (mab[q] >> 6)] = macc[q] * A.invH2;
          }
        __syncthreads(); // end of the forward sweep
      }

    // ------------------------------ backward substitution -------------------------
    // Zp holds X_{L-1} = Z_{L-1}.  All four waves; V_l is prefetched one line ahead.
    double vpre[PF];
    auto   prefetch_V = [&](int line) __attribute__((always_inline)) {
      const double *vl = vg + (size_t)line * vline;
#pragma unroll
      for (int q = 0; q < PF; ++q)
        {
          const int idx = tid + 256 * q;
          vpre[q]       = (idx < m * m) ? vl[(idx / m) * MP + (idx % m)] : 0.0;
        }
    };
    if (L >= 2 && !(SLOD_DG(A, 16)))
      prefetch_V(L - 2);
    for (int l = (SLOD_DG(A, 16)) ? -1 : L - 2; l >= 0; --l)
      {
        double *Bn = Bc0;
        load_bands(l, nullptr, Bn, tid, 256);
#pragma unroll
        for (int q = 0; q < PF; ++q)
          {
            const int idx = tid + 256 * q;
            if (idx < m * m)
              Vs[(idx / m) * ldv + (idx % m)] = vpre[q];
          }
        if (l > 0)
          prefetch_V(l - 1);
        __syncthreads();
        // Y = B_l X_{l+1}
        for (int i = tid >> 5; i < m; i += 8)
          for (int r = tid & 31; r < nc; r += 32)
            {
              double v = 0.0;
#pragma unroll
              for (int o = 0; o < BW; ++o)
                {
                  const int p = i + o - W;
                  if (p >= 0 && p < m)
                    v = fma(Bn[(i + W) * BWP + o], Zp[p * ncs + r], v);
                }
              Rb[i * ncs + r] = v;
            }
        __syncthreads();
        // X_l = Z_l - V_l Y; 16x16 output tiles dealt to the four waves
        {
          double   *xl = xg + (size_t)l * xline;
          const int k4b = (m + 3) & ~3, tib = (m + 15) >> 4, tjb = (nc + 15) >> 4;
          for (int t = wave; t < tib * tjb; t += 4)
            {
              const int ti = t / tjb, tj = t - ti * tjb;
              const int col = 16 * tj + (lane & 15);
              double    zl[4];
#pragma unroll
              for (int r = 0; r < 4; ++r)
                {
                  const int row = 16 * ti + (lane >> 4) + 4 * r;
                  zl[r]         = (row < m && col < nc) ? xl[row * ncg + col] : 0.0;
                }
              const double4_t acc = gemm_mfma_tile(Vs, ldv, Rb, ncs, k4b, ti, tj, lane);
#pragma unroll
              for (int r = 0; r < 4; ++r)
                {
                  const int row = 16 * ti + (lane >> 4) + 4 * r;
                  if (row < m && col < nc)
                    {
                      const double x     = zl[r] - acc[r];
                      xl[row * ncg + col] = x;
                      Zp[row * ncs + col] = x;
                    }
                }
            }
        }
        __syncthreads();
      }
  }

} // namespace

size_t slod_solve_ws_lds_bytes(int S, int m_max, int nc_max)
{
  // must mirror the carve-up at the top of k_solve_ws
  const int    T = slod_solve_ws_tile(m_max), W = 2 * S - 1, BW = 2 * W + 1, MP = 8 * T;
  const int    ldv = MP + 2, ncs = (nc_max + 1) & ~1, bsz = ((MP + 2 * W) * (BW + 1) + 1) & ~1;
  (void)m_max;
  const size_t n = (size_t)ldv * ldv + 2 * (size_t)ldv * ncs + 2 * MP + 3 * (size_t)bsz;
  return ((n * sizeof(double) + 2 * (size_t)nc_max * sizeof(int)) + 15) & ~(size_t)15;
}

int slod_solve_ws_tile(int m_max)
{
  static const int tiles[] = {2, 3, 4, 5, 6, 8, 10, 12, 14};
  for (int t : tiles)
    if (8 * t >= m_max)
      return t;
  return 0;
}

template <int T, int S>
static hipError_t launch_ws_TS(const SlodKernelArgs &a, int n_patches, size_t lds, hipStream_t st)
{
  const void *fn = reinterpret_cast<const void *>(k_solve_ws<T, S>);
  hipError_t  e  = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess)
    return e;
  if (a.debug)
    {
      int nb = 0;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256, lds);
      fprintf(stderr, "[slod] k_solve_ws<%d,%d>: %d patches, lds %zu B, occupancy %d blocks/CU\n", T, S, n_patches,
              lds, nb);
    }
  hipLaunchKernelGGL((k_solve_ws<T, S>), dim3(n_patches), dim3(256), lds, st, a);
  return hipGetLastError();
}

template <int S>
static hipError_t launch_ws_S(const SlodKernelArgs &a, int n_patches, size_t lds, hipStream_t st)
{
  switch (slod_solve_ws_tile(a.m_max))
    {
      case 2:
        return launch_ws_TS<2, S>(a, n_patches, lds, st);
      case 3:
        return launch_ws_TS<3, S>(a, n_patches, lds, st);
      case 4:
        return launch_ws_TS<4, S>(a, n_patches, lds, st);
      case 5:
        return launch_ws_TS<5, S>(a, n_patches, lds, st);
      case 6:
        return launch_ws_TS<6, S>(a, n_patches, lds, st);
      case 8:
        return launch_ws_TS<8, S>(a, n_patches, lds, st);
      case 10:
        return launch_ws_TS<10, S>(a, n_patches, lds, st);
      case 12:
        return launch_ws_TS<12, S>(a, n_patches, lds, st);
      case 14:
        return launch_ws_TS<14, S>(a, n_patches, lds, st);
      default:
        return hipErrorInvalidValue;
    }
}

hipError_t slod_launch_solve_ws(int S, const SlodKernelArgs &a, int n_patches, size_t lds, hipStream_t st)
{
  return S == 1 ? launch_ws_S<1>(a, n_patches, lds, st) : launch_ws_S<2>(a, n_patches, lds, st);
}
