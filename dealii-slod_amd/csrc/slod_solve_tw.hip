// k_solve_tw: twisted + wave-specialised patch solve (default).
#include <type_traits>
#include "slod_assemble.hip.h"
#include "slod_select.hip.h"

namespace
{
  // ---------------------------------------------------------------------------------
  // K2 (twisted + wave-specialised).  Two chains per patch: chain 0 eliminates lines
  // 0..mid-1 downwards, chain 1 lines L-1..mid+1 upwards; they meet at line mid.  Wave c
  // (c = 0,1) is the Gauss-Jordan wave of chain c (k_solve_ws's register scheme), wave 2+c
  // its helper (RHS block, Z = V R on the fp64 MFMA pipe, band fetches).  V and Z/X live only
  // in the global workspace (L2): the helpers feed the MFMA A operand straight from there, so
  // LDS holds just the RHS block, the pivot row and the stencil bands of each chain (26 KB at
  // C2) and four workgroups stay resident per CU while the dependent chain per patch is halved.
  // One workgroup barrier per line pair: A_t = "V of step t is in the workspace, the bands of
  // step t+1 are in LDS".
  // ---------------------------------------------------------------------------------
  template <int T, int S>
  __global__ __launch_bounds__(256, ws_min_waves(T)) void k_solve_tw(const SlodKernelArgs A)
  {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const SlodPatchDesc d = A.desc[blockIdx.x];
    constexpr int       W = 2 * S - 1, BW = 2 * W + 1, MP = 8 * T;
    constexpr int       BWP = BW + 1, BROWS = MP + 2 * W, bsz = (BROWS * BWP + 1) & ~1;
    const int           tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // scalar: roles, chain and LDS blocks are wave-uniform
    // (not const: the step loops of the forward sweep pass it through an empty asm each iteration, so
    // that per-lane addresses are recomputed there instead of being hoisted out of the loop and spilled)
    int                 lane = tid & 63;
    const int           chain = wave & 1;
    const bool          is_gj = wave < 2;
    const int           m = d.m, L = d.L, nc = d.n_c, n = A.n_sub;
    const int           mm = A.m_max, ncs = (A.nc_max + 1) & ~1, ncg = A.nc_max;
    const bool          tr  = (d.flags & SLOD_F_TRANSPOSED) != 0;
    const int           npx = d.nx + 1;
    const int           chsz = MP * ncs + MP + 6 * bsz; // doubles per chain

    double *cb   = smem + chain * chsz;
    double *Rb   = cb;               // [MP][ncs]  RHS block / Y of the chain
    double *rowb = Rb + MP * ncs;    // [MP]       pivot row of the chain's GJ wave
    double *Tf   = rowb + MP;        // padded bands: T of the chain's first line,
    double *Tn0  = Tf + bsz;         //   T bands of the steps (by step parity),
    double *Tn1  = Tn0 + bsz;
    double *Bc0  = Tn1 + bsz;        //   coupling bands of the steps (by step mod 3)
    double *Bc1  = Bc0 + bsz;
    // (third coupling buffer: Bc1 + bsz, addressed through Bbuf)
    auto    Bbuf = [&](double *base, int stp) { return base + ((stp + 3) % 3) * bsz; };
    double *ocb  = smem + (1 - chain) * chsz; // the other chain's block
    int    *colk = reinterpret_cast<int *>(smem + 2 * chsz); // [2][nc_max]
    // Columns of P^T are zero until the sweep reaches their coarse cell, and so are the columns of
    // Z: each chain numbers the columns by the step at which they become active (perm: chain order
    // -> column, inv: column -> chain order, act: first active step, by chain order) and computes,
    // stores and reads back only the active prefix.
    int *perm = colk + 2 * A.nc_max + chain * 3 * A.nc_max, *inv = perm + A.nc_max, *act = inv + A.nc_max;
    int *operm = colk + 2 * A.nc_max + (1 - chain) * 3 * A.nc_max, *oinv = operm + A.nc_max, *oact = oinv + A.nc_max;
    (void)operm;

    const double *st    = A.st + (size_t)blockIdx.x * A.st_stride;
    double       *vg    = A.vinv + (size_t)blockIdx.x * A.v_stride;
    double       *xg    = A.xs + (size_t)blockIdx.x * A.x_stride; // X, column order of P^T
    double       *zg    = A.zs + (size_t)blockIdx.x * A.x_stride; // Z, each chain's own column order
    // V of a line: the 8 x 8 lane tiles of the Gauss-Jordan wave, tile (a, b) as a contiguous T x T
    // block (row major) at index 8 a + b.  Both triangles are stored: a row of the MFMA A operand is then
    // T contiguous doubles per lane tile (wide loads off two base addresses); reading the lower triangle
    // transposed out of a packed upper one cost ten scattered loads and their 64-bit addresses per row tile.
    const size_t  vline = (size_t)64 * T * T, xline = (size_t)mm * ncg;
    // MFMA A operand from that storage: lane (r = lane & 15, kq = lane >> 4) of step kk = g T + kr
    // takes V[16 ti + r][k], k = T (kq + 4 g) + kr -- the K index is permuted so that the lane tile
    // column of k is a per-lane constant plus 4 g and kr is a compile-time constant; the B operand
    // uses the same k (row k of the LDS block).
    auto load_A = [&](const double *vl, int ti, double (&dst)[MP / 4]) __attribute__((always_inline)) {
      const int     row = min(16 * ti + (lane & 15), MP - 1);
      const int     rt = row / T, rr = row - rt * T, kq = lane >> 4;
      const double *base = vl + (rt * 8 + kq) * (T * T) + rr * T;
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int kr = 0; kr < T; ++kr)
          dst[g * T + kr] = base[g * 4 * T * T + kr];
    };
    // row of the B operand (LDS block) of step kk for this lane: T (kq + 4 g) + kr
    auto brow = [&](int kk) { return T * ((lane >> 4) + 4 * (kk / T)) + kk % T; };

    const int mid = L / 2;
    // Row tile ti of X_mid = V_mid R_mid (R_mid in chain 0's LDS block), all columns: after the meeting
    // line is inverted every wave of the workgroup takes one row tile (one operand round trip instead
    // of one per tile in series on a single wave)
    auto mid_tile = [&](int ti) __attribute__((always_inline)) {
      const double *vl = vg + (size_t)mid * vline;
      double       *xl = xg + (size_t)mid * xline;
      const int     tiles_j = (nc + 15) >> 4;
      double        av[MP / 4];
      load_A(vl, ti, av);
      for (int tj = 0; tj < tiles_j; tj += 2)
        {
          double4_t     acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
          const double *bp  = smem + 16 * tj + (lane & 15);
          const bool    two = tj + 1 < tiles_j;
#pragma unroll
          for (int kk = 0; kk < MP / 4; ++kk)
            {
              acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bp[brow(kk) * ncs], acc0, 0, 0, 0);
              if (two)
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bp[brow(kk) * ncs + 16], acc1, 0, 0, 0);
            }
#pragma unroll
          for (int r = 0; r < 4; ++r)
            {
              const int row = 16 * ti + (lane >> 4) + 4 * r, col = 16 * tj + (lane & 15);
              if (row < m && col < nc)
                xl[row * ncg + col] = acc0[r];
              if (two && row < m && col + 16 < nc)
                xl[row * ncg + col + 16] = acc1[r];
            }
        }
    };
    const int n0 = mid, n1 = L - 1 - mid, nstp = n0 > n1 ? n0 : n1;
    const int nmy = chain == 0 ? n0 : n1; // lines of this chain
    const int dl  = chain == 0 ? 1 : -1;
    auto      line_of = [&](int c, int t) { return c == 0 ? t : L - 1 - t; }; // t == n_c gives mid

    if ((SLOD_DG(A, (1 << 20))) && tid == 0)
      {
        A.ms[(size_t)blockIdx.x * A.nc_max * A.nc_max] = (double)wall_clock64();
        if (A.nc_max * A.nc_max >= 16) // where the workgroup runs: HW_ID (reg 4) and XCC_ID (reg 20)
          {
            A.ms[(size_t)blockIdx.x * A.nc_max * A.nc_max + 12] = (double)__builtin_amdgcn_s_getreg((31 << 11) | 4);
            A.ms[(size_t)blockIdx.x * A.nc_max * A.nc_max + 13] = (double)__builtin_amdgcn_s_getreg((31 << 11) | 20);
          }
      }
    if ((SLOD_DG(A, (1 << 20))) && lane == 0 && A.nc_max * A.nc_max >= 24) // SIMD of each wave (HW_ID bits 5:4)
      A.ms[(size_t)blockIdx.x * A.nc_max * A.nc_max + 20 + wave] = (double)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    auto tstamp = [&](int k) {
      if ((SLOD_DG(A, (1 << 20))) && tid == 0 && A.nc_max * A.nc_max >= 48)
        A.ms[(size_t)blockIdx.x * A.nc_max * A.nc_max + 32 + k] = (double)wall_clock64();
    };
    tstamp(0);
    // Fused stencil assembly: the workgroup builds the stencil planes of its own patch (k_assemble
    // as a device function), saving a launch and its tail; the planes still go through the
    // workspace, which the band fetches and the selection stage read back
    if (S == 1 && A.fuse_assemble)
      {
        for (int node = tid; node < npx * (d.ny + 1); node += 256)
          assemble_node<S>(A, d, blockIdx.x, node);
        __syncthreads();
      }
    for (int idx = tid; idx < 2 * chsz; idx += 256)
      smem[idx] = 0.0;
    for (int c = tid; c < nc; c += 256)
      {
        int kx, ky;
        cell_of_col(d, c / S, kx, ky);
        colk[c]            = kx;
        colk[A.nc_max + c] = ky;
      }
    __syncthreads();
    {
      // first active step of a column: the sweep (line index + 1 = node row) reaches its coarse cell
      auto first_step = [&](int ch, int c) {
        const int kq = tr ? colk[c] : colk[A.nc_max + c];
        const int a  = ch == 0 ? kq * n - 1 : L - (kq + 1) * n;
        return a > 0 ? a : 0;
      };
      for (int idx = tid; idx < 2 * nc; idx += 256)
        {
          const int ch = idx / nc, c = idx - ch * nc, a = first_step(ch, c);
          int       rank = 0;
          for (int c2 = 0; c2 < nc; ++c2)
            {
              const int a2 = first_step(ch, c2);
              rank += (a2 < a || (a2 == a && c2 < c)) ? 1 : 0;
            }
          int *pp = colk + 2 * A.nc_max + ch * 3 * A.nc_max;
          pp[rank]                = c;
          pp[A.nc_max + c]        = rank;
          pp[2 * A.nc_max + rank] = a;
        }
      // (visible after the barrier that ends the band prologue below)
    }
    // write the (zero padded) bands of `line`: T (within the line; zero band if !with_T) and the
    // coupling line -> line + dl of this chain; t0/nt = caller's thread slice
    auto put_bands = [&](int line, double *Tdst, bool with_T, double *Bdst, int t0, int nt) __attribute__((always_inline)) {
      for (int idx = t0; idx < m * BW; idx += nt)
        {
          const int i = idx / BW, oi = idx - i * BW, o = oi - W;
          if (Tdst)
            Tdst[(i + W) * BWP + oi] = with_T ? coupling<S>(st, A.nn_max, npx, tr, m, line, i, 0, o) : 0.0;
          if (Bdst)
            Bdst[(i + W) * BWP + oi] = coupling<S>(st, A.nn_max, npx, tr, m, line, i, dl, o);
        }
    };
    // Band schedule: right after the sweep of step t the chain's GJ wave needs the bands of
    // "step t": T of line(t+1) (zero for chain 1's meeting line, whose T is added by chain 0) in
    // the T buffer of parity t, and the coupling line(t) -> line(t+1) in the B buffer t mod 3.
    // They are written one step ahead: steps 0 and 1 here, step t+1 by the helper during step t
    // (mod 3: the helper still reads the coupling of step t-2 for its RHS block in step t).
    auto put_step = [&](int stp, int t0, int nt) __attribute__((always_inline)) {
      if (stp >= nmy)
        return;
      put_bands(line_of(chain, stp + 1), (stp & 1) ? Tn1 : Tn0, !(chain == 1 && stp + 1 == nmy), nullptr, t0, nt);
      put_bands(line_of(chain, stp), nullptr, true, Bbuf(Bc0, stp), t0, nt);
    };
    {
      const int t0 = (wave >> 1) * 64 + lane;
      put_bands(line_of(chain, 0), Tf, true, nullptr, t0, 128);
      put_step(0, t0, 128);
      put_step(1, t0, 128);
    }
    __syncthreads();
    tstamp(1);

    // ------------------------------ forward elimination ---------------------------
    if (is_gj)
      {
        __builtin_amdgcn_s_setprio(3);
        int    gy = lane >> 3, gx = lane & 7;
        double a[T][T];
        // neighbour lane in the same lane-grid row through DPP (row_shr:1 / row_shl:1, no LDS trip):
        // the lanes at the ends of an 8-lane grid row receive a foreign (finite) value that only
        // ever meets a zero band entry
        auto dpp_from_prev = [](double x) {
          const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x111, 0xf, 0xf, true);
          const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x111, 0xf, 0xf, true);
          return __hiloint2double(hi, lo);
        };
        auto dpp_from_next = [](double x) {
          const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x101, 0xf, 0xf, true);
          const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x101, 0xf, 0xf, true);
          return __hiloint2double(hi, lo);
        };
        // Scalar problems (W == 1): the band entries a lane needs do not depend on the tile row (column
        // pass) resp. tile column (row pass): 3T values each, fetched from LDS ONCE in one batch.  The
        // generic version below re-reads them per tile row under its per-row scheduling barriers and
        // ends up with one exposed LDS round trip per FMA pair (measured: 9.2 us per step, tools/tw_timeline.py).
        auto next_S_w1 = [&](const double *Tsrc, const double *Bl) __attribute__((always_inline)) {
          const double *cbp = Bl + (T * gx) * BWP + 2;
          const double *dbp = Bl + (T * gy) * BWP + 2;
          double        cb[T][3], db[T][3];
#pragma unroll
          for (int tb = 0; tb < T; ++tb)
#pragma unroll
            for (int f = 0; f < 3; ++f)
              {
                cb[tb][f] = cbp[(tb + f) * BWP - f];
                db[tb][f] = dbp[(tb + f) * BWP - f];
              }
          // column pass: a <- -(a B) over the lane-grid row (neighbour lanes through DPP)
#pragma unroll
          for (int ta = 0; ta < T; ++ta)
            {
              double ext[T + 2];
              ext[0]     = -dpp_from_prev(a[ta][T - 1]);
              ext[T + 1] = -dpp_from_next(a[ta][0]);
#pragma unroll
              for (int tb = 0; tb < T; ++tb)
                ext[1 + tb] = -a[ta][tb];
#pragma unroll
              for (int tb = 0; tb < T; ++tb)
                a[ta][tb] = fma(ext[tb + 2], cb[tb][2], fma(ext[tb + 1], cb[tb][1], ext[tb] * cb[tb][0]));
            }
          // row pass: a <- T - B^T a over the lane-grid column (neighbour lanes through bpermute, one batch)
          double up[T], dn[T];
#pragma unroll
          for (int tb = 0; tb < T; ++tb)
            {
              up[tb] = __shfl(a[T - 1][tb], lane - 8, 64);
              dn[tb] = __shfl(a[0][tb], lane + 8, 64);
            }
#pragma unroll
          for (int tb = 0; tb < T; ++tb)
            {
              const int j = T * gx + tb;
              double    tv[T], ext[T + 2];
#pragma unroll
              for (int ta = 0; ta < T; ++ta)
                {
                  const int      i  = T * gy + ta;
                  const unsigned oi = (unsigned)(j - i + 1);
                  tv[ta]            = Tsrc[(i + 1) * BWP + (oi < 3u ? oi : 3u)];
                }
              ext[0]     = up[tb];
              ext[T + 1] = dn[tb];
#pragma unroll
              for (int ta = 0; ta < T; ++ta)
                ext[1 + ta] = a[ta][tb];
#pragma unroll
              for (int ta = 0; ta < T; ++ta)
                a[ta][tb] = fma(-ext[ta + 2], db[ta][2], fma(-ext[ta + 1], db[ta][1], fma(-ext[ta], db[ta][0], tv[ta])));
            }
        };
        auto next_S = [&](const double *Tsrc, const double *Bl) __attribute__((always_inline)) {
          if (W == 1)
            {
              next_S_w1(Tsrc, Bl);
              return;
            }
          const double *cbp = Bl + (T * gx) * BWP + 2 * W;
#pragma unroll
          for (int ta = 0; ta < T; ++ta)
            {
              double ext[T + 2 * W];
#pragma unroll
              for (int w = 0; w < W; ++w)
                {
                  ext[w]         = -dpp_from_prev(a[ta][T - W + w]);
                  ext[W + T + w] = -dpp_from_next(a[ta][w]);
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
                    acc = fma(ext[tb + f], cbp[(tb + f) * BWP - f], acc);
                  a[ta][tb] = acc;
                }
              __builtin_amdgcn_sched_barrier(0);
            }
          const double *dbp = Bl + (T * gy) * BWP + 2 * W;
          // rows of the lane-grid neighbours above and below, all columns in one batch
          double up[T][W], dn[T][W];
          if (W == 1) // (wider bands: too many registers, the exchange stays inside the column loop)
            {
#pragma unroll
              for (int tb = 0; tb < T; ++tb)
#pragma unroll
                for (int w = 0; w < W; ++w)
                  {
                    up[tb][w] = __shfl(a[T - W + w][tb], lane - 8, 64);
                    dn[tb][w] = __shfl(a[w][tb], lane + 8, 64);
                  }
            }
#pragma unroll
          for (int tb = 0; tb < T; ++tb)
            {
              const int j = T * gx + tb;
              double    ext[T + 2 * W];
#pragma unroll
              for (int w = 0; w < W; ++w)
                {
                  ext[w]         = W == 1 ? up[tb][w] : __shfl(a[T - W + w][tb], lane - 8, 64);
                  ext[W + T + w] = W == 1 ? dn[tb][w] : __shfl(a[w][tb], lane + 8, 64);
                }
#pragma unroll
              for (int ta = 0; ta < T; ++ta)
                ext[W + ta] = a[ta][tb];
#pragma unroll
              for (int ta = 0; ta < T; ++ta)
                {
                  const int      i  = T * gy + ta;
                  const unsigned oi = (unsigned)(j - i + W);
                  double         acc = Tsrc[(i + W) * BWP + (oi < (unsigned)BW ? oi : (unsigned)BW)];
#pragma unroll
                  for (int e = 0; e < BW; ++e)
                    acc = fma(-ext[ta + e], dbp[(ta + e) * BWP - e], acc);
                  a[ta][tb] = acc;
                }
              __builtin_amdgcn_sched_barrier(0);
            }
        };
        double             acc_p[2] = {0.0, 0.0}; // diag: shader cycles of a pivot up to the scaled row / of its update
        unsigned long long bad = 0; // lanes that saw a non-positive pivot, accumulated on the scalar unit
        auto sweep = [&]() __attribute__((always_inline)) {
          for (int ka = 0; ka * T < m; ++ka)
            {
#pragma unroll
              for (int a0 = 0; a0 < T; ++a0)
                {
                  const int k = T * ka + a0;
                  if (k >= m || ((SLOD_DG(A, 4)) && k > 0)) // wave-uniform
                    continue;
                  const long long q0 = (SLOD_DG(A, (1 << 20))) ? clock64() : 0;
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
                  bad |= __builtin_amdgcn_ballot_w64(!(piv > 0.0));
                  const double p = fast_rcp(piv), pn = -p;
                  // the row is scaled by -1/pivot: the rank-1 update is then a plain multiply-add (no
                  // negated copy of the column per pivot)
                  double ri[T], sn[T];
#pragma unroll
                  for (int ta = 0; ta < T; ++ta)
                    ri[ta] = rowb[T * gy + ta];
#pragma unroll
                  for (int tb = 0; tb < T; ++tb)
                    sn[tb] = rowb[T * gx + tb] * pn;
                  __builtin_amdgcn_wave_barrier();
                  long long q1 = 0;
                  if (SLOD_DG(A, (1 << 20)))
                    {
                      // (diag: the clock is read once the scaled row is in registers)
                      double probe = sn[0] + ri[0];
                      asm volatile("" : "+v"(probe));
                      q1 = clock64();
                    }
#pragma unroll
                  for (int ta = 0; ta < T; ++ta)
#pragma unroll
                    for (int tb = 0; tb < T; ++tb)
                      a[ta][tb] = fma(ri[ta], sn[tb], a[ta][tb]);
                  if (gy == ka)
                    {
#pragma unroll
                      for (int tb = 0; tb < T; ++tb)
                        a[a0][tb] = -sn[tb];
                    }
                  if (gx == ka)
                    {
#pragma unroll
                      for (int ta = 0; ta < T; ++ta)
                        a[ta][a0] = ri[ta] * p;
                      if (gy == ka)
                        a[a0][a0] = pn;
                    }
                  if (SLOD_DG(A, (1 << 20)))
                    {
                      double probe = a[0][0] + a[T - 1][T - 1];
                      asm volatile("" : "+v"(probe));
                      const long long q2 = clock64();
                      acc_p[0] += (double)(q1 - q0);
                      acc_p[1] += (double)(q2 - q1);
                    }
                  __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        // tile <-> global [MP][MP] block (V lines, and the meeting-line contribution of chain 1)
        auto store_tile = [&](double *dst, double sign) __attribute__((always_inline)) {
          double *tp = dst + (gy * 8 + gx) * (T * T);
#pragma unroll
          for (int ta = 0; ta < T; ++ta)
#pragma unroll
            for (int tb = 0; tb < T; ++tb)
              tp[ta * T + tb] = sign * a[ta][tb];
        };
        // a = T of the chain's first line (chain 1 without lines contributes nothing)
#pragma unroll
        for (int ta = 0; ta < T; ++ta)
#pragma unroll
          for (int tb = 0; tb < T; ++tb)
            {
              const int      i = T * gy + ta, j = T * gx + tb;
              const unsigned oi = (unsigned)(j - i + W);
              const double   v  = Tf[(i + W) * BWP + (oi < (unsigned)BW ? oi : (unsigned)BW)];
              a[ta][tb]         = (chain == 1 && nmy == 0) ? 0.0 : v;
            }
        double acc_t[4] = {0.0, 0.0, 0.0, 0.0}; // diag: sweep, store + next_S, barrier wait, store alone
        for (int t = 0; t < nstp; ++t)
          {
            const bool active = t < nmy;
            asm volatile("" : "+v"(lane));
            gy = lane >> 3;
            gx = lane & 7;
            const long long c0 = (SLOD_DG(A, (1 << 20))) ? wall_clock64() : 0;
            long long       c1 = c0, c2 = c0;
            if (active)
              {
                sweep();
                c1 = (SLOD_DG(A, (1 << 20))) ? wall_clock64() : 0;
                if (!(SLOD_DG(A, 32768)))
                  store_tile(vg + (size_t)line_of(chain, t) * vline, -1.0);
                if (SLOD_DG(A, (1 << 20)))
                  acc_t[3] += (double)(wall_clock64() - c1);
                // Schur complement of the next line (the meeting line after the last step), in
                // registers: overlaps the drain of the V stores before the barrier
                if (!(SLOD_DG(A, 16384)))
                  next_S((t & 1) ? Tn1 : Tn0, Bbuf(Bc0, t));
                c2 = (SLOD_DG(A, (1 << 20))) ? wall_clock64() : 0;
              }
            __syncthreads(); // A_t: V of step t is in the workspace, bands of step t+1 are in LDS
            if (SLOD_DG(A, (1 << 20)))
              {
                const long long c3 = wall_clock64();
                acc_t[0] += (double)(c1 - c0);
                acc_t[1] += (double)(c2 - c1);
                acc_t[2] += (double)(c3 - c2);
              }
          }
        if ((SLOD_DG(A, (1 << 20))) && tid == 0 && A.nc_max * A.nc_max >= 48)
          {
            for (int k = 0; k < 4; ++k)
              A.ms[(size_t)blockIdx.x * A.nc_max * A.nc_max + 40 + k] = acc_t[k];
            if (A.nc_max * A.nc_max >= 52)
              for (int k = 0; k < 2; ++k)
                A.ms[(size_t)blockIdx.x * A.nc_max * A.nc_max + 48 + k] = acc_p[k];
          }
        tstamp(2);
        // the meeting line: a0 = T_mid - W_0, a1 = -W_1
        if (chain == 1)
          store_tile(vg + (size_t)mid * vline, 1.0);
        __syncthreads(); // M1: chain 1's contribution is in the workspace
        if (chain == 0)
          {
            const double *w1 = vg + (size_t)mid * vline + (gy * 8 + gx) * (T * T);
#pragma unroll
            for (int ta = 0; ta < T; ++ta)
#pragma unroll
              for (int tb = 0; tb < T; ++tb)
                a[ta][tb] += w1[ta * T + tb];
            sweep();
            store_tile(vg + (size_t)mid * vline, -1.0);
          }
        if (bad != 0 && lane == 0 && !SLOD_DG(A, -1))
          atomicOr(A.status, 1);
        __builtin_amdgcn_s_setprio(0);
        __syncthreads(); // M2: V_mid is in the workspace
        if (!(SLOD_DG(A, 8)))
          for (int ti = wave; 16 * ti < m; ti += 4)
            mid_tile(ti);
        __syncthreads(); // M3: X_mid is in the workspace
      }
    else
      {
        // ===== helper wave of the chain: one wave, so its own phases need no barrier =====
        // R = (with_F ? F_line : 0) - Bprev^T Z(prev line), Z from the workspace
        // Each lane owns column r = lane&31 (+32 ...) and one half of the rows; it walks down its
        // rows with a sliding window of Z(prev)[p][r], p = i-W..i+W: one coalesced workspace load
        // per row instead of 2W+1 gathers, and the loads do not depend on the arithmetic.
        // ncols columns of Rb; cmap: column of P^T of Rb column r (nullptr: r); zmap: column of
        // zprev that holds it (nullptr: r), valid only if that column was active at step tlim of the
        // chain that wrote it (zact, by zprev column)
        auto build_R = [&](int line, const double *Bprev, const double *zprev, int zstride, bool with_F, bool add, int ncols,
                           const int *cmap, const int *zmap, const int *zact, int tlim, bool zlds) __attribute__((always_inline)) {
          if (SLOD_DG(A, 2))
            return;
          // Lane (g, cq) of the 8 x 8 lane grid owns rows T g .. T g + T - 1 of the columns cq + 8 k.
          // Everything a lane needs from the workspace (the T + 2W rows of Z(prev) around its rows, CP
          // columns per pass) is requested in one batch, branch-free: one exposed workspace round trip
          // per pass.  (A sliding window down a column, one load and three dependent LDS reads per row,
          // cost 10 us per step at C2 -- more than the Gauss-Jordan sweep it has to hide behind.)
          constexpr int CP = 2;
          const int     g = lane >> 3, cq = lane & 7;
          const double *zp = zprev ? zprev : zg;
          for (int r0 = 0; r0 < ncols; r0 += 8 * CP)
            {
              double win[CP][T + 2 * W];
              int    kxn[CP], kyn[CP], cc[CP];
              bool   valid[CP];
#pragma unroll
              for (int k = 0; k < CP; ++k)
                {
                  const int r  = r0 + 8 * k + cq;
                  valid[k]     = r < ncols;
                  const int rc = valid[k] ? r : 0;
                  const int c = cmap ? cmap[rc] : rc, zc = zmap ? zmap[rc] : rc;
                  const bool zok = valid[k] && zprev && zact[zc] <= tlim;
                  cc[k]  = c;
                  kxn[k] = colk[c] * n;
                  kyn[k] = colk[A.nc_max + c] * n;
                  if (zlds) // Z(prev) is what the GEMM of the previous step left in the LDS block (row stride ncs)
                    {
#pragma unroll
                      for (int e = 0; e < T + 2 * W; ++e)
                        {
                          const int    p   = T * g + e - W;
                          const bool   inb = zok && p >= 0 && p < m;
                          const double z   = Rb[(inb ? p : 0) * ncs + (inb ? zc : 0)];
                          win[k][e]        = inb ? z : 0.0;
                        }
                    }
                  else
                    {
#pragma unroll
                      for (int e = 0; e < T + 2 * W; ++e)
                        {
                          const int    p   = T * g + e - W;
                          const bool   inb = zok && p >= 0 && p < m;
                          const double z   = zp[(inb ? p : 0) * zstride + (inb ? zc : 0)];
                          win[k][e]        = inb ? z : 0.0;
                        }
                    }
                }
              if (zlds) // the block is rewritten in place: every lane has its window before any lane writes
                {
                  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                  __builtin_amdgcn_wave_barrier();
                  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
#pragma unroll
              for (int ta = 0; ta < T; ++ta)
                {
                  const int i = T * g + ta;
                  double    bnd[BW];
#pragma unroll
                  for (int e = 0; e < BW; ++e) // B[p][i], p = i+e-W (zero padded band rows)
                    bnd[e] = Bprev[(i + e) * BWP + (2 * W - e)];
                  const int pos = i / S, comp = i - pos * S;
                  const int ix = tr ? line + 1 : pos + 1, iy = tr ? pos + 1 : line + 1;
#pragma unroll
                  for (int k = 0; k < CP; ++k)
                    {
                      const int r = r0 + 8 * k + cq;
                      const bool wr = valid[k] && i < m;
                      double    v = (add && wr) ? Rb[i * ncs + r] : 0.0;
                      if (with_F)
                        {
                          const int  jx = ix - kxn[k], jy = iy - kyn[k];
                          const bool in = jx >= 0 && jx <= n && jy >= 0 && jy <= n;
                          if (S == 1)
                            {
                              const double w = ((jx == 0 || jx == n) ? 1.0 : 2.0) * ((jy == 0 || jy == n) ? 1.0 : 2.0);
                              v += in ? A.scale * w : 0.0;
                            }
                          else if (in)
                            v += A.scale * pt_weight<S>(d, n, A.quirk, ix, iy, comp, cc[k]);
                        }
#pragma unroll
                      for (int e = 0; e < BW; ++e)
                        v = fma(-bnd[e], win[k][ta + e], v);
                      if (wr)
                        Rb[i * ncs + r] = v;
                    }
                }
            }
        };
        const int tiles_i = (m + 15) >> 4;
        // (first ncols columns of) Z(line) = V(line) Rb -> workspace row xl; A operand straight from
        // the workspace (rows clamped)
        auto gemm_Z = [&](int line, int ncols, double *xl, int xstride) __attribute__((always_inline)) {
          if (SLOD_DG(A, 8))
            return;
          const double *vl = vg + (size_t)line * vline;
          const int     tiles_j = (ncols + 15) >> 4;
          // one row tile of A (16 x MP of V, from the workspace) feeds all column tiles; the
          // next row tile is fetched while the MFMAs of the current one run
          double av[MP / 4], an[MP / 4];
          load_A(vl, 0, av);
          for (int ti = 0; ti < tiles_i; ++ti)
            {
              if (ti + 1 < tiles_i)
                load_A(vl, ti + 1, an);
              for (int tj = 0; tj < tiles_j; tj += 2)
                {
                  // two independent accumulators (column tiles tj, tj+1) back to back
                  double4_t     acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
                  const double *bp  = Rb + 16 * tj + (lane & 15);
                  const bool    two = tj + 1 < tiles_j;
#pragma unroll
                  for (int kk = 0; kk < MP / 4; ++kk)
                    {
                      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bp[brow(kk) * ncs], acc0, 0, 0, 0);
                      if (two)
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bp[brow(kk) * ncs + 16], acc1, 0, 0, 0);
                    }
#pragma unroll
                  for (int r = 0; r < 4; ++r)
                    {
                      const int row = 16 * ti + (lane >> 4) + 4 * r, col = 16 * tj + (lane & 15);
                      if (row < m && col < ncols)
                        xl[row * xstride + col] = acc0[r];
                      if (two && row < m && col + 16 < ncols)
                        xl[row * xstride + col + 16] = acc1[r];
                    }
                }
#pragma unroll
              for (int kk = 0; kk < MP / 4; ++kk)
                av[kk] = an[kk];
            }
        };
        // The same product for a line whose accumulators fit in registers (TI row tiles x 2 column
        // tiles): Z goes to the workspace (the backward sweep reads it) AND replaces the RHS block in
        // LDS, where the next step's build_R takes its windows from -- no workspace round trip between
        // two steps of the chain.  av: row tile 0 of V, requested by the caller before it built the block.
        constexpr int TI = (MP + 15) / 16;
        auto gemm_Z_keep = [&](int line, int ncols, double *xl, double (&av)[MP / 4]) __attribute__((always_inline)) {
          const double *vl = vg + (size_t)line * vline;
          const bool    two = ncols > 16;
          // the B operands of a lane do not depend on the row tile: read once, then the block is free
          double        bq[2][MP / 4];
          const double *bp = Rb + (lane & 15);
#pragma unroll
          for (int kk = 0; kk < MP / 4; ++kk)
            {
              bq[0][kk] = bp[brow(kk) * ncs];
              bq[1][kk] = bp[brow(kk) * ncs + 16];
            }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          double an[MP / 4];
#pragma unroll
          for (int ti = 0; ti < TI; ++ti)
            {
              if (ti < tiles_i)
                {
                  if (ti + 1 < TI && ti + 1 < tiles_i)
                    load_A(vl, ti + 1, an);
                  double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                  for (int kk = 0; kk < MP / 4; ++kk)
                    {
                      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bq[0][kk], acc0, 0, 0, 0);
                      if (two)
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bq[1][kk], acc1, 0, 0, 0);
                    }
#pragma unroll
                  for (int r = 0; r < 4; ++r)
                    {
                      const int row = 16 * ti + (lane >> 4) + 4 * r, col = lane & 15;
                      if (row < m && col < ncols)
                        {
                          xl[row * ncols + col] = acc0[r];
                          Rb[row * ncs + col]   = acc0[r];
                        }
                      if (two && row < m && col + 16 < ncols)
                        {
                          xl[row * ncols + col + 16] = acc1[r];
                          Rb[row * ncs + col + 16]   = acc1[r];
                        }
                    }
                  if (ti + 1 < TI)
                    {
#pragma unroll
                      for (int kk = 0; kk < MP / 4; ++kk)
                        av[kk] = an[kk];
                    }
                }
            }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        };
        // put_step in two halves for this wave: the stencil values of step stp are requested before the
        // RHS block / GEMM work of the step (branch-free, so that the loads really are in flight) and
        // land in the band buffers after it
        constexpr int NBQ = (MP * BW + 63) / 64;
        double        tq[NBQ], bq_[NBQ];
        auto fetch_step = [&](int stp) __attribute__((always_inline)) {
          const bool on = stp < nmy;
          const int  lT = line_of(chain, stp + 1), lB = line_of(chain, stp);
          const bool with_T = !(chain == 1 && stp + 1 == nmy);
#pragma unroll
          for (int q = 0; q < NBQ; ++q)
            {
              const int  idx = lane + 64 * q, i = idx / BW, oi = idx - i * BW, o = oi - W, j = i + o;
              const int  pi = i / S, ci = i - pi * S, pj = j / S, cj = j - pj * S, dp = pj - pi;
              const bool ok = on && idx < m * BW && j >= 0 && j < m && dp >= -1 && dp <= 1;
              // (same indexing as coupling<S>, with the address clamped instead of a branch)
              const int ixT = tr ? lT + 1 : pi + 1, iyT = tr ? pi + 1 : lT + 1;
              const int ixB = tr ? lB + 1 : pi + 1, iyB = tr ? pi + 1 : lB + 1;
              const int dirT = (tr ? dp : 0) * 3 + (tr ? 0 : dp) + 4;
              const int dirB = ((tr ? dp : dl) + 1) * 3 + (tr ? dl : dp) + 1;
              const double vT = st[ok ? (size_t)((dirT * S + ci) * S + cj) * A.nn_max + ixT + iyT * npx : 0];
              const double vB = st[ok ? (size_t)((dirB * S + ci) * S + cj) * A.nn_max + ixB + iyB * npx : 0];
              tq[q]  = (ok && with_T) ? vT : 0.0;
              bq_[q] = ok ? vB : 0.0;
            }
        };
        auto store_step = [&](int stp) __attribute__((always_inline)) {
          if (stp >= nmy)
            return;
          double *Tdst = (stp & 1) ? Tn1 : Tn0, *Bdst = Bbuf(Bc0, stp);
#pragma unroll
          for (int q = 0; q < NBQ; ++q)
            {
              const int idx = lane + 64 * q, i = idx / BW, oi = idx - i * BW;
              if (idx < m * BW)
                {
                  Tdst[(i + W) * BWP + oi] = tq[q];
                  Bdst[(i + W) * BWP + oi] = bq_[q];
                }
            }
        };
        double acc_h[4] = {0.0, 0.0, 0.0, 0.0}; // diag: fwd_step, put_step, barrier wait, build_R alone
        int  na = 0;           // active columns (chain order) at the step being processed
        bool z_in_lds = false; // the LDS block holds Z of the chain's previous line (gemm_Z_keep)
        // RHS block and Z of step tl: active columns only; Z(prev) of a column that became active
        // at this very step was never written (it is zero)
        // A line of Z is stored compactly: [m][na(line)], so only whole cache lines of active data move
        auto fwd_step = [&](int tl) __attribute__((always_inline)) {
          const int na_prev = na;
          while (na < nc && act[na] <= tl)
            ++na;
          const int line = line_of(chain, tl);
          const long long b0 = (SLOD_DG(A, (1 << 20))) ? wall_clock64() : 0;
          const bool keep = TI <= 3 && na <= 32 && !(SLOD_DG(A, 8));
          if (keep)
            {
              double av[MP / 4];
              load_A(vg + (size_t)line * vline, 0, av); // V(line) became visible at the last barrier
              build_R(line, Bbuf(Bc0, tl - 1), tl > 0 ? zg + (size_t)line_of(chain, tl - 1) * xline : nullptr, na_prev,
                      true, false, na, perm, nullptr, act, tl - 1, z_in_lds);
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
              if (SLOD_DG(A, (1 << 20)))
                acc_h[3] += (double)(wall_clock64() - b0);
              gemm_Z_keep(line, na, zg + (size_t)line * xline, av);
              z_in_lds = true;
              return;
            }
          build_R(line, Bbuf(Bc0, tl - 1), tl > 0 ? zg + (size_t)line_of(chain, tl - 1) * xline : nullptr, na_prev, true,
                  false, na, perm, nullptr, act, tl - 1, false);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          if (SLOD_DG(A, (1 << 20)))
            acc_h[3] += (double)(wall_clock64() - b0);
          gemm_Z(line, na, zg + (size_t)line * xline, na);
          z_in_lds = false;
        };
        for (int t = 0; t < nstp; ++t)
          {
            // line(t-1): its V became visible at A_{t-1}; the coupling line(t-2) -> line(t-1) is the
            // B band of step t-2
            asm volatile("" : "+v"(lane));
            const long long c0 = (SLOD_DG(A, (1 << 20))) ? wall_clock64() : 0;
            if (t > 0 && !(SLOD_DG(A, 32)))
              fetch_step(t + 1);
            if (t > 0 && t - 1 < nmy)
              fwd_step(t - 1);
            const long long c1 = (SLOD_DG(A, (1 << 20))) ? wall_clock64() : 0;
            if (t > 0 && !(SLOD_DG(A, 32)))
              store_step(t + 1); // bands the GJ wave needs after its next sweep
            const long long c2 = (SLOD_DG(A, (1 << 20))) ? wall_clock64() : 0;
            __syncthreads(); // A_t
            if (SLOD_DG(A, (1 << 20)))
              {
                const long long c3 = wall_clock64();
                acc_h[0] += (double)(c1 - c0);
                acc_h[1] += (double)(c2 - c1);
                acc_h[2] += (double)(c3 - c2);
              }
          }
        if ((SLOD_DG(A, (1 << 20))) && tid == 128 && A.nc_max * A.nc_max >= 48)
          for (int k = 0; k < 4; ++k)
            A.ms[(size_t)blockIdx.x * A.nc_max * A.nc_max + 44 + k] = acc_h[k];
        // R/Z of the last step (the shorter chain of an even L already did its last line in the loop)
        if (nmy == nstp && nmy > 0)
          fwd_step(nstp - 1);
        __syncthreads(); // M1 (also: both chains' last Z are in the workspace)
        if (chain == 0)
          {
            // R_mid = F_mid - B^T Z(mid-1) - B'^T Z(mid+1); the bands are the last B of each chain.
            // Built while chain 0's Gauss-Jordan wave inverts the meeting line (it needs Z, not V_mid).
            const double *B0 = Bbuf(Bc0, n0 - 1);
            double       *ob = ocb + MP * ncs + MP + 3 * bsz; // other chain's Bc0
            const double *B1 = Bbuf(ob, n1 - 1);
            // all columns, in the order of P^T; Z(mid-1) / Z(mid+1) through each chain's numbering
            int na0 = 0, na1 = 0; // row strides of the two neighbouring lines of Z
            while (na0 < nc && act[na0] <= n0 - 1)
              ++na0;
            while (na1 < nc && oact[na1] <= n1 - 1)
              ++na1;
            build_R(mid, B0, n0 > 0 ? zg + (size_t)(mid - 1) * xline : nullptr, na0, true, false, nc, nullptr, inv, act,
                    n0 - 1, false);
            __builtin_amdgcn_wave_barrier();
            if (n1 > 0)
              build_R(mid, B1, zg + (size_t)(mid + 1) * xline, na1, false, true, nc, nullptr, oinv, oact, n1 - 1, false);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
          }
        __syncthreads(); // M2: V_mid is in the workspace
        if (!(SLOD_DG(A, 8)))
          for (int ti = wave; 16 * ti < m; ti += 4)
            mid_tile(ti);
        __syncthreads(); // M3
      }

    tstamp(3);
    // ------------------------------ backward substitution -------------------------
    // from the meeting line outwards, both chains at once; each chain = 2 waves (128 threads).
    // Band entries of the next line are fetched (stencil planes, workspace latency) before the
    // GEMM of the current line and written to the other band buffer after it.
    {
      const int t128 = (wave >> 1) * 64 + lane, w2 = wave >> 1;
      const int tiles_i = (m + 15) >> 4, tiles_j = (nc + 15) >> 4;
      constexpr int NBV = (MP * BW + 127) / 128;
      double        bv[NBV];
      auto fetch_B = [&](int line) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NBV; ++q)
          {
            const int idx = t128 + 128 * q, i = idx / BW, o = idx - i * BW - W;
            bv[q] = (idx < m * BW) ? coupling<S>(st, A.nn_max, npx, tr, m, line, i, dl, o) : 0.0;
          }
      };
      auto store_B = [&](double *Bdst) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NBV; ++q)
          {
            const int idx = t128 + 128 * q, i = idx / BW, oi = idx - i * BW;
            if (idx < m * BW)
              Bdst[(i + W) * BWP + oi] = bv[q];
          }
      };
      const int tstart = (SLOD_DG(A, 16)) ? -1 : nstp - 1;
      if (tstart >= 0 && tstart < nmy)
        {
          fetch_B(line_of(chain, tstart));
          store_B((tstart & 1) ? Bc1 : Bc0);
        }
      else if (nmy > 0 && tstart >= 0)
        {
          fetch_B(line_of(chain, nmy - 1)); // shorter chain: its first active step is nmy-1
          store_B(((nmy - 1) & 1) ? Bc1 : Bc0);
        }
      int nab = nc; // active columns = row stride of the line of Z being read (steps run downwards)
      // Lines whose accumulators fit in registers keep X of the line just solved in the chain's LDS
      // block: the band product of the next line takes it from there (in place, as build_R does in the
      // forward sweep), so a line costs no workspace round trip beyond the prefetched V and Z.
      constexpr int TIB   = (MP + 15) / 16;
      const bool    keepb = TIB <= 3 && nc <= 32 && !(SLOD_DG(A, 64));
      if (keepb)
        for (int idx = t128; idx < m * nc; idx += 128)
          {
            const int i = idx / nc, c = idx - i * nc;
            Rb[i * ncs + c] = xg[(size_t)mid * xline + i * ncg + c];
          }
      // (two instances of the loop, one per variant: sharing one loop made the register allocator spill
      // the prefetched operands of the LDS-resident variant)
      auto backward = [&](auto KEEP) __attribute__((always_inline)) {
      constexpr bool keep_x = decltype(KEEP)::value;
      for (int t = tstart; t >= 0; --t)
        {
          while (nab > 0 && act[nab - 1] > t)
            --nab;
          const bool    active = t < nmy;
          const int     line = line_of(chain, t), prev = line_of(chain, t + 1); // prev: solved before (mid first)
          const double *Bn = (t & 1) ? Bc1 : Bc0;
          asm volatile("" : "+v"(lane));
          __syncthreads(); // bands of this line are in LDS, X(prev) is in the workspace / the LDS block
          if (active && t > 0)
            fetch_B(line_of(chain, t - 1));
          if constexpr (keep_x)
            {
              const double *vl = vg + (size_t)line * vline;
              double       *xl = xg + (size_t)line * xline;
              const int     g = lane >> 3, cq = lane & 7; // rows T g .. T g + T - 1, columns cq + 8 (2 w2 + k)
              const int     colw = 16 * w2 + (lane & 15);  // GEMM: wave w2 owns column tile w2
              // (band product and GEMM of a wave touch the same 16 columns of the block, its own: the
              // phases of a line are ordered by wave barriers; only the bands need the workgroup barrier)
              const bool    gemm = active && 16 * w2 < nc;
              constexpr int NPRE = 1; // row tiles of V requested before the band product
              double        av[NPRE][MP / 4], zl[NPRE][4];
              const double *zrow = zg + (size_t)line * xline;
              const int     zc   = colw < nc ? inv[colw] : nc;
              const bool    zok  = zc < nab; // the chain order is sorted by the first active step
              auto          load_Z = [&](int ti, double (&dst)[4]) __attribute__((always_inline)) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  {
                    const int    row = 16 * ti + (lane >> 4) + 4 * r;
                    const bool   ok  = row < m && zok;
                    const double z   = zrow[ok ? row * nab + zc : 0];
                    dst[r]           = ok ? z : 0.0;
                  }
              };
              if (gemm)
                {
#pragma unroll
                  for (int ti = 0; ti < NPRE; ++ti)
                    if (ti < tiles_i)
                      {
                        load_A(vl, ti, av[ti]);
                        load_Z(ti, zl[ti]);
                      }
                }
              double y[2][T];
              if (active)
                {
                  // Y = B(line -> prev) X(prev), X(prev) from the block, one column at a time
#pragma unroll
                  for (int k = 0; k < 2; ++k)
                    {
                      const int c = cq + 8 * (2 * w2 + k);
                      double    win[T + 2 * W];
#pragma unroll
                      for (int e = 0; e < T + 2 * W; ++e)
                        {
                          const int    pr  = T * g + e - W;
                          const bool   inb = c < nc && pr >= 0 && pr < m;
                          const double x   = Rb[(inb ? pr : 0) * ncs + (inb ? c : 0)];
                          win[e]           = inb ? x : 0.0;
                        }
#pragma unroll
                      for (int ta = 0; ta < T; ++ta)
                        {
                          double v = 0.0;
#pragma unroll
                          for (int o = 0; o < BW; ++o)
                            v = fma(Bn[(T * g + ta + W) * BWP + o], win[ta + o], v);
                          y[k][ta] = v;
                        }
                    }
                }
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // every window is read: the block may be rewritten
              if (active)
                {
#pragma unroll
                  for (int k = 0; k < 2; ++k)
                    {
                      const int c = cq + 8 * (2 * w2 + k);
#pragma unroll
                      for (int ta = 0; ta < T; ++ta)
                        if (c < nc && T * g + ta < m)
                          Rb[(T * g + ta) * ncs + c] = y[k][ta];
                    }
                }
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // Y is complete
              double bq[MP / 4];
              if (gemm)
                {
#pragma unroll
                  for (int kk = 0; kk < MP / 4; ++kk)
                    bq[kk] = Rb[brow(kk) * ncs + colw];
                }
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // every B operand is in registers: X(line) may replace Y
              if (gemm)
                {
#pragma unroll
                  for (int ti = 0; ti < TIB; ++ti)
                    if (ti < tiles_i)
                      {
                        double4_t acc = {0.0, 0.0, 0.0, 0.0};
                        double    zt[4];
                        if (ti < NPRE)
                          {
#pragma unroll
                            for (int kk = 0; kk < MP / 4; ++kk)
                              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ti < NPRE ? ti : 0][kk], bq[kk], acc, 0, 0, 0);
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                              zt[r] = zl[ti < NPRE ? ti : 0][r];
                          }
                        else
                          {
                            double al[MP / 4];
                            load_A(vl, ti, al);
                            load_Z(ti, zt);
#pragma unroll
                            for (int kk = 0; kk < MP / 4; ++kk)
                              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(al[kk], bq[kk], acc, 0, 0, 0);
                          }
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                          {
                            const int row = 16 * ti + (lane >> 4) + 4 * r;
                            if (row < m && colw < nc)
                              {
                                const double x       = zt[r] - acc[r];
                                xl[row * ncg + colw] = x;
                                Rb[row * ncs + colw] = x;
                              }
                          }
                      }
                }
              if (active && t > 0)
                store_B(((t - 1) & 1) ? Bc1 : Bc0);
              continue;
            }
          if (active)
            {
              // Y = B(line -> prev) X(prev): lane = (column r, quarter of the rows), sliding window
              const double *xp = xg + (size_t)prev * xline;
              const int     qr = t128 >> 5, i_lo = (qr * m) >> 2, i_hi = ((qr + 1) * m) >> 2;
              for (int r = t128 & 31; r < nc; r += 32)
                {
                  double win[BW];
#pragma unroll
                  for (int e = 0; e < BW; ++e)
                    {
                      const int p = i_lo + e - W;
                      win[e]      = (p >= 0 && p < m) ? xp[p * ncg + r] : 0.0;
                    }
                  for (int i = i_lo; i < i_hi; ++i)
                    {
                      const int    pn    = i + 1 + W;
                      const double xnext = (pn < m) ? xp[pn * ncg + r] : 0.0;
                      double       v     = 0.0;
#pragma unroll
                      for (int o = 0; o < BW; ++o)
                        v = fma(Bn[(i + W) * BWP + o], win[o], v);
                      Rb[i * ncs + r] = v;
#pragma unroll
                      for (int e = 0; e + 1 < BW; ++e)
                        win[e] = win[e + 1];
                      win[BW - 1] = xnext;
                    }
                }
            }
          __syncthreads();
          if (active)
            {
              const double *vl = vg + (size_t)line * vline;
              double       *xl = xg + (size_t)line * xline;
              for (int tt = w2; tt < tiles_i * tiles_j; tt += 2)
                {
                  const int ti = tt / tiles_j, tj = tt - ti * tiles_j;
                  const int col = 16 * tj + (lane & 15);
                  // Z(line) in the chain's column numbering; never written where the column was not
                  // yet active in the forward sweep (zero there)
                  const double *zrow = zg + (size_t)line * xline;
                  const int     zc   = col < nc ? inv[col] : nc;
                  const bool    zok  = zc < nab; // the chain order is sorted by the first active step
                  double        zl[4];
#pragma unroll
                  for (int r = 0; r < 4; ++r)
                    {
                      const int row = 16 * ti + (lane >> 4) + 4 * r;
                      zl[r]         = (row < m && zok) ? zrow[row * nab + zc] : 0.0;
                    }
                  double4_t     acc = {0.0, 0.0, 0.0, 0.0};
                  const double *bp  = Rb + col;
                  double        av[MP / 4];
                  load_A(vl, ti, av);
#pragma unroll
                  for (int kk = 0; kk < MP / 4; ++kk)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bp[brow(kk) * ncs], acc, 0, 0, 0);
#pragma unroll
                  for (int r = 0; r < 4; ++r)
                    {
                      const int row = 16 * ti + (lane >> 4) + 4 * r;
                      if (row < m && col < nc)
                        xl[row * ncg + col] = zl[r] - acc[r];
                    }
                }
              if (t > 0)
                store_B(((t - 1) & 1) ? Bc1 : Bc0);
            }
          // the next iteration's first barrier orders the X and band writes before their readers
        }
      };
      if (keepb)
        backward(std::true_type{});
      else
        backward(std::false_type{});
    }
    // Fused selection stage: the same workgroup goes on with M, D, the boundary trace, the
    // least squares, phi and psi of its patch (X is fresh in this CU's L2 slice).  Patches of
    // different shapes then balance inside ONE launch: rim patches are quick to solve and slow
    // to select (SVD fallback), full patches the other way round.
    const bool stamp = (SLOD_DG(A, (1 << 20))) && tid == 0; // per-patch timeline (100 MHz clock) into ms
    if (stamp)
      A.ms[(size_t)blockIdx.x * A.nc_max * A.nc_max + 1] = (double)wall_clock64();
    if (S == 1 && A.fuse_select)
      {
        __syncthreads(); // X of all lines is written; LDS is free
        select_patch<S>(A, A.nb_buf, A.nf_max, blockIdx.x, smem);
        if (stamp)
          A.ms[(size_t)blockIdx.x * A.nc_max * A.nc_max + 2] = (double)wall_clock64();
      }
  }

} // namespace

size_t slod_solve_tw_lds_bytes(int S, int m_max, int nc_max)
{
  // must mirror the carve-up at the top of k_solve_tw
  const int    T = slod_solve_ws_tile(m_max), W = 2 * S - 1, BW = 2 * W + 1, MP = 8 * T;
  const int    ncs = (nc_max + 1) & ~1, bsz = ((MP + 2 * W) * (BW + 1) + 1) & ~1;
  (void)m_max;
  const size_t chsz = (size_t)MP * ncs + MP + 6 * (size_t)bsz;
  return ((2 * chsz * sizeof(double) + 8 * (size_t)nc_max * sizeof(int)) + 15) & ~(size_t)15; // colk + perm/inv/act
}

template <int T, int S>
static hipError_t launch_tw_TS(const SlodKernelArgs &a, int n_patches, size_t lds, hipStream_t st)
{
  const void *fn = reinterpret_cast<const void *>(k_solve_tw<T, S>);
  hipError_t  e  = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess)
    return e;
  if (a.debug)
    {
      int nb = 0;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256, lds);
      fprintf(stderr, "[slod] k_solve_tw<%d,%d>: %d patches, lds %zu B, occupancy %d blocks/CU\n", T, S, n_patches,
              lds, nb);
    }
  hipLaunchKernelGGL((k_solve_tw<T, S>), dim3(n_patches), dim3(256), lds, st, a);
  return hipGetLastError();
}

template <int S>
static hipError_t launch_tw_S(const SlodKernelArgs &a, int n_patches, size_t lds, hipStream_t st)
{
  switch (slod_solve_ws_tile(a.m_max))
    {
      case 2:
        return launch_tw_TS<2, S>(a, n_patches, lds, st);
      case 3:
        return launch_tw_TS<3, S>(a, n_patches, lds, st);
      case 4:
        return launch_tw_TS<4, S>(a, n_patches, lds, st);
      case 5:
        return launch_tw_TS<5, S>(a, n_patches, lds, st);
      case 6:
        return launch_tw_TS<6, S>(a, n_patches, lds, st);
      case 8:
        return launch_tw_TS<8, S>(a, n_patches, lds, st);
      case 10:
        return launch_tw_TS<10, S>(a, n_patches, lds, st);
      case 12:
        return launch_tw_TS<12, S>(a, n_patches, lds, st);
      case 14:
        return launch_tw_TS<14, S>(a, n_patches, lds, st);
      default:
        return hipErrorInvalidValue;
    }
}

hipError_t slod_launch_solve_tw(int S, const SlodKernelArgs &a, int n_patches, size_t lds, hipStream_t st)
{
  return S == 1 ? launch_tw_S<1>(a, n_patches, lds, st) : launch_tw_S<2>(a, n_patches, lds, st);
}
