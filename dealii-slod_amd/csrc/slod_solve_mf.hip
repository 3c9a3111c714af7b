// k_solve_mf: twisted block-tridiagonal patch solve, Gauss-Jordan on the fp64 matrix pipe (default).
#include "slod_assemble.hip.h"
#include "slod_select.hip.h"

#include <type_traits>

namespace
{
  // ---------------------------------------------------------------------------------
  // K2 (MFMA-factorised).  Same elimination as k_solve_tw -- two chains per patch that meet at
  // line mid, V_l = S_l^-1 explicit so that both right-hand-side sweeps are GEMMs -- but the
  // inversion itself runs on the matrix pipe:
  //   * the Gauss-Jordan wave of a chain keeps S_l as NT x NT tiles of 16 x 16 in the
  //     v_mfma_f64_16x16x4_f64 accumulator layout (lane (g,c) = (l>>4, l&15), register r:
  //     entry (16 ti + 4 r + g, 16 tj + c)) and sweeps FOUR pivots per step: the pivot rows
  //     K = {4 kb .. 4 kb + 3} are register r = kb & 3 of tile row kb >> 2 -- already the B operand
  //     C[K,:] of the rank-4 update, and by symmetry also its A operand C[:,K]^T.  Only
  //     W = C_KK^-1 (4 x 4, solved per lane for its own row) has to be folded in, which needs the
  //     panel once through a wave-private LDS line: U = -(C[:,K] W), then
  //     C <- C + U * C[K,:] is NT^2 MFMAs, the pivot rows are overwritten by W C[K,:] and -W.
  //     One LDS round trip and ~100 VALU instructions per 4 pivots instead of ~95 per pivot.
  //   * the next Schur complement T - B^T V B (B banded) is formed in place, one tile row at a
  //     time, through a small wave-private LDS window: rows (B^T V), then columns ((B^T V) B).
  //   * right-hand sides are independent per column: a helper wave (forward) / every wave
  //     (backward) owns 16-column tiles of Z/X, keeps them in the accumulator layout -- which is
  //     the B operand layout of the next product -- and needs no workgroup barrier in the
  //     backward sweep at all.
  // V is stored per line as NT x NT accumulator tiles (512-byte coalesced stores); by symmetry
  // tile (kk >> 2, ti), register kk & 3 IS the A operand V[16 ti + c][4 kk + g] of the GEMMs.
  // The patch matrix is scaled by a power of two so that its entries are <= 1: the sweep mixes
  // C with the identity (C_KK - I in the column update), harmless only at that scale.
  // ---------------------------------------------------------------------------------
  __host__ __device__ constexpr int mf_min_waves(int NT, int S) { return NT <= 5 ? 2 : 1; }

  typedef double double2_t __attribute__((ext_vector_type(2)));

// per-wave phase clocks of the timing-experiment build (tools/mf_timeline.py): cycles per phase,
// summed over the lines, written to the patch's scratch block at the end
#ifdef SLOD_ENABLE_DIAG
#define SLOD_TMR(i)                       \
  do                                      \
    {                                     \
      const long long now_ = clock64();   \
      tq[i] += now_ - tlast;              \
      tlast = now_;                       \
    }                                     \
  while (0)
#else
#define SLOD_TMR(i) \
  do                \
    {               \
    }               \
  while (0)
#endif

#define SLOD_WAVE_SYNC()                                        \
  do                                                            \
    {                                                           \
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
      __builtin_amdgcn_wave_barrier();                          \
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
    }                                                           \
  while (0)

  template <int NT, int S>
  __global__ __launch_bounds__(256, mf_min_waves(NT, S)) void k_solve_mf(const SlodKernelArgs A)
  {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const SlodPatchDesc d = A.desc[blockIdx.x];
    constexpr int       W = 2 * S - 1, BW = 2 * W + 1, MP = 16 * NT, NB = MP / 4;
    constexpr int       BWP = BW + 1, BROWS = MP + 2 * W, bsz = (BROWS * BWP + 1) & ~1;
    constexpr int       PST = MP + 2;          // panel row stride (even: 16-byte aligned pivot block)
    constexpr int       WST = MP + 2 * W + 1, WROWS = 16 + 3 * W; // mixing window (+ W saved halo rows)
    constexpr int       XST = 17;              // backward strips: [MP + 2 W][16 + 1]
    constexpr int       TSC = 16 * 17;          // tile transposition scratch
    constexpr int       chsz = (4 * PST + WROWS * WST + TSC + 6 * bsz + 1) & ~1; // doubles per chain
    const int           tid = threadIdx.x, lane = tid & 63;
    // the wave index is uniform: as a scalar the role branches are scalar branches and every
    // per-chain LDS / workspace base stays in SGPRs
    const int           wave  = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int           chain = wave & 1;
    const bool          is_gj = wave < 2;
    const int           m = d.m, L = d.L, nc = d.n_c, n = A.n_sub;
    const int           mm = A.m_max, ncg = A.nc_max;
    const int           nct = (nc + 15) >> 4;
    const bool          tr  = (d.flags & SLOD_F_TRANSPOSED) != 0;
    const int           npx = d.nx + 1;

    double *cb    = smem + chain * chsz;
    double *panel = cb;                  // [4][PST]      pivot rows of the current block step
    double *ywin  = panel + 4 * PST;     // [16 + 2 W][WST]  one tile row of V (+ halo rows / zero halo columns)
    double *yhs   = ywin + (16 + 2 * W) * WST; // [W][WST]    saved halo rows for the next tile row
    double *tsc   = ywin + WROWS * WST;  // [16][17]      tile transposition scratch
    double *Tf    = tsc + TSC;           // padded bands: T of the chain's first line,
    double *Tn0   = Tf + bsz;            //   T bands of the steps (by step parity),
    double *Tn1   = Tn0 + bsz;
    double *Bc0   = Tn1 + bsz;           //   coupling bands of the steps (by step mod 3)
    auto    Bbuf  = [&](double *base, int stp) { return base + ((stp + 3) % 3) * bsz; };
    double *ocb   = smem + (1 - chain) * chsz;
    int    *colk  = reinterpret_cast<int *>(smem + 2 * chsz); // [2][nc_max]

    const double *st    = A.st + (size_t)blockIdx.x * A.st_stride;
    double       *vg    = A.vinv + (size_t)blockIdx.x * A.v_stride;
    double       *xg    = A.xs + (size_t)blockIdx.x * A.x_stride;
    const size_t  vline = (size_t)MP * MP, xline = (size_t)mm * ncg;

    const int mid = L / 2;
    const int n0 = mid, n1 = L - 1 - mid, nstp = n0 > n1 ? n0 : n1;
    const int nmy = chain == 0 ? n0 : n1; // lines of this chain
    const int dl  = chain == 0 ? 1 : -1;
    auto      line_of = [&](int ch, int t) { return ch == 0 ? t : L - 1 - t; }; // t == n_ch gives mid

    if (SLOD_DG(A, (1 << 20)) && tid == 0)
      A.ms[(size_t)blockIdx.x * A.nc_max * A.nc_max] = (double)wall_clock64();
#ifdef SLOD_ENABLE_DIAG
    long long tq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = clock64();
#endif
    // fused stencil assembly (scalar problems): the workgroup builds the planes of its own patch
    if (S == 1 && A.fuse_assemble)
      {
        for (int node = tid; node < npx * (d.ny + 1); node += 256)
          assemble_node<S>(A, d, blockIdx.x, node);
      }
    for (int idx = tid; idx < 2 * chsz; idx += 256)
      smem[idx] = 0.0;
    for (int cc = tid; cc < nc; cc += 256)
      {
        int kx, ky;
        cell_of_col(d, cc / S, kx, ky);
        colk[cc]            = kx;
        colk[A.nc_max + cc] = ky;
      }
    __syncthreads();
    // power-of-two scale: the largest diagonal stencil entry of the patch becomes < 1
    double sc = 1.0;
    {
      double dmx = 0.0;
      for (int node = tid; node < npx * (d.ny + 1); node += 256)
#pragma unroll
        for (int a = 0; a < S; ++a)
          dmx = fmax(dmx, st[(size_t)((4 * S + a) * S + a) * A.nn_max + node]);
      for (int off = 32; off > 0; off >>= 1)
        dmx = fmax(dmx, __shfl_xor(dmx, off, 64));
      if (lane == 0)
        smem[wave] = dmx; // panel of chain 0: not yet in use
      __syncthreads();
      dmx = fmax(fmax(smem[0], smem[1]), fmax(smem[2], smem[3]));
      __syncthreads();
      if (dmx > 0.0 && dmx < 1e300)
        {
          int e;
          (void)frexp(dmx, &e);
          sc = ldexp(1.0, -e);
        }
      if (tid < 4)
        smem[tid] = 0.0;
    }
    const double scF = A.scale * sc;
    // write the (zero padded) bands of `line`, scaled: T (within the line; zero band if !with_T)
    // and the coupling line -> line + dl of this chain; t0/nt = caller's thread slice
    auto put_bands = [&](int line, double *Tdst, bool with_T, double *Bdst, int t0, int nt) __attribute__((always_inline)) {
      if (S == 1)
        {
          // scalar problems: entry (i, o) of a band is ONE stencil plane value at node(i) = nd0 + i nds,
          // plane (dy+1)*3 + dx+1 with (dx,dy) = (o, dlb) or transposed; a thread takes whole rows:
          // its BW loads are independent and go out together
          const int     dlb = Tdst ? 0 : dl;
          double       *dst = Tdst ? Tdst : Bdst;
          const int     nd0 = tr ? (line + 1) + npx : 1 + (line + 1) * npx, nds = tr ? npx : 1;
          const double *pl[BW];
#pragma unroll
          for (int e = 0; e < BW; ++e)
            {
              const int o = e - W, dx = tr ? dlb : o, dy = tr ? o : dlb;
              pl[e]       = st + (size_t)((dy + 1) * 3 + dx + 1) * A.nn_max + nd0;
            }
          for (int i = t0; i < m; i += nt)
            {
              double v[BW];
#pragma unroll
              for (int e = 0; e < BW; ++e)
                v[e] = pl[e][i * nds];
#pragma unroll
              for (int e = 0; e < BW; ++e)
                dst[(i + W) * BWP + e] = (with_T && (unsigned)(i + e - W) < (unsigned)m) ? sc * v[e] : 0.0;
            }
          return;
        }
      for (int idx = t0; idx < m * BW; idx += nt)
        {
          const int i = idx / BW, oi = idx - i * BW, o = oi - W;
          if (Tdst)
            Tdst[(i + W) * BWP + oi] = with_T ? sc * coupling<S>(st, A.nn_max, npx, tr, m, line, i, 0, o) : 0.0;
          if (Bdst)
            Bdst[(i + W) * BWP + oi] = sc * coupling<S>(st, A.nn_max, npx, tr, m, line, i, dl, o);
        }
    };
    // Band schedule (as in k_solve_tw): right after the sweep of step t the chain's GJ wave needs
    // T of line(t+1) (zero for chain 1's meeting line, whose T is added by chain 0) in the T
    // buffer of parity t, and the coupling line(t) -> line(t+1) in the B buffer t mod 3; they are
    // written one step ahead: steps 0 and 1 here, step t+1 by the helper during step t.
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

    // ------------------------------ forward elimination ---------------------------
    // Per-lane constants (tile coordinates, LDS addresses, masks) are re-derived from an opaque
    // copy of the lane id wherever they are used: written once up front, the compiler hoists every
    // address and mask of the unrolled code out of the line loops and spills them, and a reload
    // from scratch sits on the dependent chain.  Redoing them costs one or two VALU each.
    auto olane = [&]() __attribute__((always_inline)) {
      // v_mbcnt on an opaque zero: three VALU, nothing kept alive between uses (a copy of `lane`
      // would itself be spilled and reloaded on the dependent chain)
      int z = 0;
      asm volatile("" : "+v"(z));
      return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)z));
    };
    if (is_gj)
      {
        __builtin_amdgcn_s_setprio(3);
        // S_l is symmetric: only the tiles ti <= tj are kept (NT (NT + 1) / 2 accumulators);
        // wherever a lower tile would be read, its mirror is read transposed through LDS.
        constexpr int NU = NT * (NT + 1) / 2;
        double4_t     acc[NU];
        bool          bad = false;
#define UT(ti, tj) ((ti) * NT - ((ti) * ((ti) - 1)) / 2 + (tj) - (ti))

        // entry (16 ti + 4 r + g, 16 tj + c) of a padded T band buffer, |ti - tj| <= 1: element
        // 16 ti BWP + toff[tj - ti + 1][r] (column BW of a band row is a zero pad); identity on
        // the padding diagonal
        auto t_offsets = [&](int og, int oc, int (&toff)[3][4]) __attribute__((always_inline)) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            {
              const int rho = 4 * r + og;
#pragma unroll
              for (int dt = 0; dt < 3; ++dt)
                {
                  const unsigned oi = (unsigned)(oc + 16 * (dt - 1) - rho + W);
                  toff[dt][r]       = (rho + W) * BWP + (int)(oi < (unsigned)BW ? oi : (unsigned)BW);
                }
            }
        };
        auto t_entry = [&](const double *Tsrc, const int (&toff)[3][4], int og, int oc, int ti, int tj, int r,
                           bool pad_identity) __attribute__((always_inline)) {
          double v = 0.0;
          if (tj - ti <= 1 && ti - tj <= 1)
            {
              v = Tsrc[16 * ti * BWP + toff[tj - ti + 1][r]];
              if (ti == tj && pad_identity)
                v = (4 * r + og == oc && 16 * ti + oc >= m) ? 1.0 : v;
            }
          return v;
        };
        auto t_init = [&](const double *Tsrc, bool pad_identity) __attribute__((always_inline)) {
          const int ol = olane(), og = ol >> 4, oc = ol & 15;
          int       toff[3][4];
          t_offsets(og, oc, toff);
#pragma unroll
          for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = ti; tj < NT; ++tj)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                acc[UT(ti, tj)][r] = t_entry(Tsrc, toff, og, oc, ti, tj, r, pad_identity);
        };

        // blocked symmetric sweep of all pivots < m: acc <- -S^-1 (padding: -1 / identity, decoupled)
        auto sweep = [&]() __attribute__((always_inline)) {
#pragma unroll
          for (int kb = 0; kb < NB; ++kb)
            {
              if (4 * kb >= m || (SLOD_DG(A, 4) && kb > 0)) // wave-uniform
                continue;
              const int  tk = kb >> 2, q = kb & 3;
              const int  ol = olane(), og = ol >> 4, oc = ol & 15, oc3 = ol & 3;
              const bool inK = (oc >> 2) == q;
              // pivot rows C[K,:] -> panel.  Columns right of the pivot tile: register q of tile row tk
              // (the B operand already); columns left of it: column K of the tiles above, transposed
              double  vt[NT];
              double *pwr = panel + og * PST + oc;
#pragma unroll
              for (int tj = tk; tj < NT; ++tj)
                {
                  vt[tj]       = acc[UT(tk, tj)][q];
                  pwr[16 * tj] = vt[tj];
                }
              if (inK)
                {
                  double *pwt = panel + oc3 * PST + og;
#pragma unroll
                  for (int tj = 0; tj < tk; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                      pwt[16 * tj + 4 * r] = acc[UT(tj < tk ? tj : tk, tk)][r];
                }
              SLOD_WAVE_SYNC();
#pragma unroll
              for (int tj = 0; tj < tk; ++tj)
                vt[tj] = pwr[16 * tj];
              // pivot block (uniform addresses), upper triangle
              const double2_t *pb  = reinterpret_cast<const double2_t *>(panel + 16 * tk + 4 * q);
              const double2_t  r00 = pb[0], r01 = pb[1];
              const double2_t  r10 = pb[PST / 2], r11 = pb[PST / 2 + 1];
              const double2_t  r21 = pb[PST + 1];
              const double2_t  r31 = pb[3 * (PST / 2) + 1];
              const double     a00 = r00.x, a01 = r00.y, a02 = r01.x, a03 = r01.y;
              const double     a11 = r10.y, a12 = r11.x, a13 = r11.y;
              const double     a22 = r21.x, a23 = r21.y, a33 = r31.y;
              // C_KK = L D L^T, then row g of W = C_KK^-1: solve C_KK w = e_g
              bad |= !(a00 > 0.0);
              const double p0  = fast_rcp(a00);
              const double l10 = a01 * p0, l20 = a02 * p0, l30 = a03 * p0;
              const double b11 = fma(-l10, a01, a11), b21 = fma(-l20, a01, a12), b31 = fma(-l30, a01, a13);
              const double b22 = fma(-l20, a02, a22), b32 = fma(-l30, a02, a23), b33 = fma(-l30, a03, a33);
              bad |= !(b11 > 0.0);
              const double p1  = fast_rcp(b11);
              const double l21 = b21 * p1, l31 = b31 * p1;
              const double c22 = fma(-l21, b21, b22), c32 = fma(-l31, b21, b32), c33 = fma(-l31, b31, b33);
              bad |= !(c22 > 0.0);
              const double p2  = fast_rcp(c22);
              const double l32 = c32 * p2;
              const double d33 = fma(-l32, c32, c33);
              bad |= !(d33 > 0.0);
              const double p3 = fast_rcp(d33);
              // forward (L y = e_g), scale (D), backward (L^T w = z)
              const double e0 = og == 0 ? 1.0 : 0.0, e1 = og == 1 ? 1.0 : 0.0, e2 = og == 2 ? 1.0 : 0.0,
                           e3 = og == 3 ? 1.0 : 0.0;
              const double y0 = e0;
              const double y1 = fma(-l10, y0, e1);
              const double y2 = fma(-l21, y1, fma(-l20, y0, e2));
              const double y3 = fma(-l32, y2, fma(-l31, y1, fma(-l30, y0, e3)));
              const double w3 = y3 * p3;
              const double w2 = fma(-l32, w3, y2 * p2);
              const double w1 = fma(-l31, w3, fma(-l21, w2, y1 * p1));
              const double w0 = fma(-l30, w3, fma(-l20, w2, fma(-l10, w1, y0 * p0)));
              // A operand U = -(C[:,K] W): lane (g,c) holds U[16 ti + c][g]; panel column of this lane
              double        u[NT];
              const double *prd = panel + oc;
#pragma unroll
              for (int ti = 0; ti < NT; ++ti)
                u[ti] = -fma(w3, prd[3 * PST + 16 * ti],
                             fma(w2, prd[2 * PST + 16 * ti], fma(w1, prd[PST + 16 * ti], w0 * prd[16 * ti])));
              // B operand: C[K,:], with C_KK - I in the pivot columns (so that C[J,K] <- C[J,K] W)
              vt[tk] -= (inK && og == oc3) ? 1.0 : 0.0;
              // the tile row that holds the NEXT pivot rows goes first: the next block step can
              // publish its panel while the other tile rows are still in the matrix pipe
              constexpr int NBc = NB;
              const int     tkn = (kb + 1 < NBc) ? (kb + 1) >> 2 : tk;
#pragma unroll
              for (int tt = 0; tt < NT; ++tt)
                {
                  const int ti = (tt + tkn) % NT;
#pragma unroll
                  for (int tj = ti; tj < NT; ++tj)
                    acc[UT(ti, tj)] = __builtin_amdgcn_mfma_f64_16x16x4f64(u[ti], vt[tj], acc[UT(ti, tj)], 0, 0, 0);
                }
              // pivot rows: W C[K,:] = -U, and -W in the pivot block
              const double wsel = oc3 == 0 ? w0 : (oc3 == 1 ? w1 : (oc3 == 2 ? w2 : w3));
#pragma unroll
              for (int tj = tk; tj < NT; ++tj)
                acc[UT(tk, tj)][q] = (tj == tk && inK) ? -wsel : -u[tj];
            }
        };

        // acc = -V_l  ->  acc = T_next - B^T V_l B = T_next + B^T acc B   (B = coupling l -> l+1),
        // upper tiles only, in place, one tile row at a time from the LAST to the first (so that the
        // transposed tiles a tile row reads are still un-mixed), through an LDS window with absolute
        // column indices [W + j]: rows first (B^T acc: needs the W rows above and below -- halo rows --
        // and, for the tile left of the diagonal, the mirror of the tile above), then columns.
        auto next_S = [&](const double *Tsrc, const double *Bl, bool pad_identity) __attribute__((always_inline)) {
#pragma unroll
          for (int tt = 0; tt < NT; ++tt)
            {
              const int ti = NT - 1 - tt, tl = ti > 0 ? ti - 1 : 0;
              const int ol = olane(), og = ol >> 4, oc = ol & 15;
              int       toff[3][4];
              t_offsets(og, oc, toff);
              double *wrow = ywin + (og + W) * WST + W + oc; // own row 4 r + g, column c of tile 0
              // ---- fill for the row mixing, columns of the tiles >= tl
              if (ti > 0)
                {
                  // upper halo: the last W rows of tile row ti - 1 (un-mixed: not processed yet)
                  if (og >= 4 - W)
                    {
#pragma unroll
                      for (int tj = tl; tj < NT; ++tj)
                        ywin[(og - (4 - W)) * WST + W + 16 * tj + oc] = acc[UT(tl, tj)][3];
                    }
                  // own rows, tile left of the diagonal: the mirror of tile (ti - 1, ti)
#pragma unroll
                  for (int r = 0; r < 4; ++r)
                    ywin[(oc + W) * WST + W + 16 * tl + 4 * r + og] = acc[UT(tl, ti)][r];
                }
              else
                for (int x = ol; x < W * WST; x += 64)
                  ywin[x] = 0.0;
#pragma unroll
              for (int tj = ti; tj < NT; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  wrow[4 * r * WST + 16 * tj] = acc[UT(ti, tj)][r];
              if (ti + 1 < NT)
                {
                  // lower halo: the first W rows of tile row ti + 1, un-mixed: saved when that row was
                  // processed (columns >= 16 (ti + 1)); mirrors of the tiles (ti, ti + 1) and (ti - 1, ti + 1)
                  // for the columns of the diagonal tile and of the tile left of it
                  if (og < W)
                    {
#pragma unroll
                      for (int tj = ti + 1; tj < NT; ++tj)
                        ywin[(16 + W + og) * WST + W + 16 * tj + oc] = yhs[og * WST + W + 16 * tj + oc];
                    }
                  if (oc < W) // columns of the diagonal tile (and of the tile left of it): mirrors
                    {
#pragma unroll
                      for (int r = 0; r < 4; ++r)
                        {
                          ywin[(16 + W + oc) * WST + W + 16 * ti + 4 * r + og] = acc[UT(ti, ti + 1 < NT ? ti + 1 : ti)][r];
                          if (ti > 0)
                            ywin[(16 + W + oc) * WST + W + 16 * tl + 4 * r + og] = acc[UT(tl, ti + 1 < NT ? ti + 1 : ti)][r];
                        }
                    }
                }
              else
                for (int x = ol; x < W * WST; x += 64)
                  ywin[(16 + W) * WST + x] = 0.0;
              if (og < W) // the first rows of this tile row, for the tile row above
                {
#pragma unroll
                  for (int tj = ti; tj < NT; ++tj)
                    yhs[og * WST + W + 16 * tj + oc] = acc[UT(ti, tj)][0];
                }
              SLOD_WAVE_SYNC();
              // ---- rows: Y[i][j] = sum_e B[i + e - W][i] acc[i + e - W][j]
              double4_t yl = {0.0, 0.0, 0.0, 0.0};
              {
                const double *rbp = Bl + og * BWP + 2 * W;       // B[i + e - W][i] = rbp[(16 ti + 4 r + e) BWP - e]
                const double *yr  = ywin + og * WST + W + oc;     // acc[i + e - W][j] = yr[(4 r + e) WST + 16 tj]
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  {
                    double rb[BW];
#pragma unroll
                    for (int e = 0; e < BW; ++e)
                      rb[e] = rbp[(16 * ti + 4 * r + e) * BWP - e];
                    if (ti > 0)
                      {
                        double v = 0.0;
#pragma unroll
                        for (int e = 0; e < BW; ++e)
                          v = fma(rb[e], yr[(4 * r + e) * WST + 16 * tl], v);
                        yl[r] = v;
                      }
#pragma unroll
                    for (int tj = ti; tj < NT; ++tj)
                      {
                        double v = rb[W] * acc[UT(ti, tj)][r];
#pragma unroll
                        for (int e = 0; e < BW; ++e)
                          if (e != W)
                            v = fma(rb[e], yr[(4 * r + e) * WST + 16 * tj], v);
                        acc[UT(ti, tj)][r] = v;
                      }
                  }
              }
              SLOD_WAVE_SYNC(); // every read of the un-mixed rows precedes the rewrite
              // ---- columns: acc[i][j] = T[i][j] + sum_f Y[i][j + f - W] B[j + f - W][j]
              if (ti > 0)
                {
#pragma unroll
                  for (int r = 0; r < 4; ++r)
                    wrow[4 * r * WST + 16 * tl] = yl[r];
                }
#pragma unroll
              for (int tj = ti; tj < NT; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  wrow[4 * r * WST + 16 * tj] = acc[UT(ti, tj)][r];
              SLOD_WAVE_SYNC();
              {
                const double *cbp = Bl + oc * BWP + 2 * W;          // B[j + f - W][j] = cbp[(16 tj + f) BWP - f]
                const double *yc  = ywin + (og + W) * WST + oc;      // Y[i][j + f - W] = yc[4 r WST + 16 tj + f]
#pragma unroll
                for (int tj = ti; tj < NT; ++tj)
                  {
                    double cbv[BW];
#pragma unroll
                    for (int f = 0; f < BW; ++f)
                      cbv[f] = cbp[(16 * tj + f) * BWP - f];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                      {
                        double v = fma(cbv[W], acc[UT(ti, tj)][r], t_entry(Tsrc, toff, og, oc, ti, tj, r, pad_identity));
#pragma unroll
                        for (int f = 0; f < BW; ++f)
                          if (f != W)
                            v = fma(cbv[f], yc[4 * r * WST + 16 * tj + f], v);
                        acc[UT(ti, tj)][r] = v;
                      }
                  }
              }
              SLOD_WAVE_SYNC();
            }
        };

        // tiles -> workspace in the accumulator layout (512-byte coalesced per register).  Upper tiles
        // from the registers; store_full adds the mirrors below the diagonal (the GEMMs of the
        // helpers read every tile as the A operand), transposed through a 16 x 17 LDS scratch.
        auto store_upper = [&](double *dst, double sign) __attribute__((always_inline)) {
          const unsigned ul = (unsigned)olane();
#pragma unroll
          for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = ti; tj < NT; ++tj)
              {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  (dst + ((ti * NT + tj) * 4 + r) * 64)[ul] = sign * acc[UT(ti, tj)][r]; // uniform base + lane
                __builtin_amdgcn_sched_barrier(0);
              }
        };
        auto store_full = [&](double *dst, double sign) __attribute__((always_inline)) {
          store_upper(dst, sign);
          const int      ol = olane(), og = ol >> 4, oc = ol & 15;
          const unsigned ul = (unsigned)ol;
#pragma unroll
          for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = ti + 1; tj < NT; ++tj)
              {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  tsc[oc * 17 + 4 * r + og] = sign * acc[UT(ti, tj)][r];
                SLOD_WAVE_SYNC();
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  (dst + ((tj * NT + ti) * 4 + r) * 64)[ul] = tsc[(4 * r + og) * 17 + oc];
                SLOD_WAVE_SYNC();
              }
        };

        // acc = T of the chain's first line (chain 1 without lines contributes nothing)
        if (chain == 1 && nmy == 0)
          {
#pragma unroll
            for (int k = 0; k < NU; ++k)
              acc[k] = double4_t{0.0, 0.0, 0.0, 0.0};
          }
        else
          t_init(Tf, true);
        for (int t = 0; t < nstp; ++t)
          {
            if (t < nmy)
              {
                SLOD_TMR(0);
                sweep();
                SLOD_TMR(1);
                if (!SLOD_DG(A, 32768))
                  store_full(vg + (size_t)line_of(chain, t) * vline, -1.0);
                SLOD_TMR(2);
                // Schur complement of the next line (the meeting line after the last step)
                if (!SLOD_DG(A, 16384))
                  next_S((t & 1) ? Tn1 : Tn0, Bbuf(Bc0, t), !(chain == 1 && t + 1 == nmy));
                SLOD_TMR(3);
              }
            __syncthreads(); // A_t: V of step t is in the workspace, bands of step t+1 are in LDS
            SLOD_TMR(4);
          }
        // the meeting line: chain 0 holds T_mid - W_0, chain 1 holds -W_1 (upper tiles)
        if (chain == 1)
          store_upper(vg + (size_t)mid * vline, 1.0);
        __syncthreads(); // M1: chain 1's contribution is in the workspace
        if (chain == 0)
          {
            const double  *w1 = vg + (size_t)mid * vline;
            const unsigned ul = (unsigned)olane();
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
              for (int tj = ti; tj < NT; ++tj)
                {
#pragma unroll
                  for (int r = 0; r < 4; ++r)
                    acc[UT(ti, tj)][r] += (w1 + ((ti * NT + tj) * 4 + r) * 64)[ul];
                  __builtin_amdgcn_sched_barrier(0);
                }
            sweep();
            store_full(vg + (size_t)mid * vline, -1.0);
          }
#undef UT
        if (bad && lane == 0 && !SLOD_DG(A, -1))
          atomicOr(A.status, 1);
        __builtin_amdgcn_s_setprio(0);
        __syncthreads(); // M2: V_mid is in the workspace
        __syncthreads(); // M3: X_mid is in the workspace
      }
    else
      {
        // ===== helper wave of the chain: right-hand sides of the line one step behind =====
        // One 16-column tile of Z at a time (runtime loop: the code of a line exists once).
        // B operand of Z = V R: rop[kk] = R[4 kk + g][16 tj + c],
        // R = (with_F ? F_line : 0) - Bprev^T Z(prev line) [+ rop]; Z from the workspace, no range
        // tests: outside [0,m) the band coefficient is zero and the workspace is guarded (a line
        // without predecessor passes a zero band and any valid line as zprev).
        // Branch-free over the k-steps (all NB of them: rows >= m are masked): a branch per k-step
        // would fence every group of loads behind its own wait.
        auto build_rop = [&](double (&rop)[NB], int tj, int line, const double *Bprev, const double *zprev,
                             bool with_F, bool add) __attribute__((always_inline)) {
          if (SLOD_DG(A, 2))
            return;
          const int      ol = olane(), og = ol >> 4, oc = ol & 15;
          const unsigned zl = (unsigned)(og * ncg + oc);              // lane part of a workspace address
          const double  *bcp = Bprev + og * BWP + 2 * W;              // B[i + e - W][i] = bcp[(4 kk + e) BWP - e]
          const double  *zb  = zprev - W * ncg + 16 * tj;
          const bool     cok = 16 * tj + oc < nc;
          // scalar problems: F = scale * w(across the lines) * w(along the line), w = 1 / 2 / 0
          const int    cc  = min(16 * tj + oc, nc - 1);
          const int    kxn = colk[cc] * n, kyn = colk[A.nc_max + cc] * n;
          const int    jl  = line + 1 - (tr ? kxn : kyn);
          const double wL  = (!with_F || (unsigned)jl > (unsigned)n) ? 0.0 : ((jl == 0 || jl == n) ? scF : 2.0 * scF);
          const int    ja  = og + 1 - (tr ? kyn : kxn);                // + 4 kk = node offset along the line
          // k-steps in groups of KG: first every workspace load of the group (independent, all in
          // flight together), a scheduling fence, then the arithmetic.  Left to itself the compiler
          // keeps each load next to its FMA and waits for them one by one.
          constexpr int KG = (S == 1) ? 4 : 1;
#pragma unroll
          for (int k0 = 0; k0 < NB; k0 += KG)
            {
              double zv[KG][BW];
#pragma unroll
              for (int kq = 0; kq < KG; ++kq)
#pragma unroll
                for (int e = 0; e < BW; ++e)
                  zv[kq][e] = (k0 + kq < NB) ? (zb + (4 * (k0 + kq) + e) * ncg)[zl] : 0.0;
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int kq = 0; kq < KG; ++kq)
                {
                  const int kk = k0 + kq;
                  if (kk >= NB)
                    continue;
                  double v = add ? rop[kk] : 0.0;
                  if (S == 1)
                    {
                      const int j = ja + 4 * kk;
                      v = fma(wL, (unsigned)j > (unsigned)n ? 0.0 : ((j == 0 || j == n) ? 1.0 : 2.0), v);
                    }
                  else if (with_F)
                    {
                      const int i = 4 * kk + og, pos = i / S, comp = i - pos * S;
                      const int ix = tr ? line + 1 : pos + 1, iy = tr ? pos + 1 : line + 1;
                      v += scF * pt_weight<S>(d, n, A.quirk, ix, iy, comp, cc);
                    }
#pragma unroll
                  for (int e = 0; e < BW; ++e)
                    v = fma(-bcp[(4 * kk + e) * BWP - e], zv[kq][e], v);
                  rop[kk] = (4 * kk + og < m && cok) ? v : 0.0;
                }
              __builtin_amdgcn_sched_barrier(0);
            }
        };
        // Z(line) = V(line) R -> workspace; A operand = stored tiles of V (by symmetry)
        auto gemm_Z = [&](const double (&rop)[NB], int tj, int line) __attribute__((always_inline)) {
          if (SLOD_DG(A, 8))
            return;
          const double  *vl = vg + (size_t)line * vline;
          double        *xl = xg + (size_t)line * xline + 16 * tj;
          const int      ol = olane(), og = ol >> 4, oc = ol & 15;
          const unsigned ul = (unsigned)ol, zl = (unsigned)(og * ncg + oc);
          // k-steps in whole tiles, the last tile k-step by k-step (operands of rows >= m are zero);
          // the A tiles of the next tile row are fetched while the MFMAs of this one run
          double av[NB], an[NB];
          auto   load_A = [&](int ti, double (&dst)[NB]) __attribute__((always_inline)) {
#pragma unroll
            for (int tk = 0; tk < NT; ++tk)
              if (16 * tk < m)
                {
#pragma unroll
                  for (int q = 0; q < 4; ++q)
                    dst[4 * tk + q] = (vl + ((tk * NT + ti) * 4 + q) * 64)[ul];
                }
          };
          load_A(0, av);
#pragma unroll
          for (int ti = 0; ti < NT; ++ti)
            {
              if (16 * ti >= m) // wave-uniform
                continue;
              if (ti + 1 < NT && 16 * (ti + 1) < m)
                load_A(ti + 1, an);
              double4_t acc0 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
              for (int tk = 0; tk < NT; ++tk)
                if (16 * tk + 16 <= m)
                  {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[4 * tk + q], rop[4 * tk + q], acc0, 0, 0, 0);
                  }
                else if (16 * tk < m)
                  {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                      if (16 * tk + 4 * q < m)
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[4 * tk + q], rop[4 * tk + q], acc0, 0, 0, 0);
                  }
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (16 * ti + 4 * r + og < m && 16 * tj + oc < nc)
                  (xl + (16 * ti + 4 * r) * ncg)[zl] = acc0[r];
#pragma unroll
              for (int kk = 0; kk < NB; ++kk)
                av[kk] = an[kk];
            }
        };
        double rop[NB];
        // zero band for a line without predecessor: the B buffer of "step -1" is never written
        // before step 1 has been processed (LDS starts zeroed)
        for (int t = 0; t <= nstp; ++t)
          {
            // RHS block and Z of line(t-1): its V became visible at A_{t-1}; the coupling
            // line(t-2) -> line(t-1) is the B band of step t-2.  t == nstp: the last step (the
            // shorter chain of an even L already did its last line inside the loop)
            if (t > 0 && t - 1 < nmy)
              {
                const int line = line_of(chain, t - 1);
                for (int tj = 0; tj < nct; ++tj)
                  {
                    SLOD_TMR(0);
                    build_rop(rop, tj, line, Bbuf(Bc0, t - 2), xg + (size_t)line_of(chain, t > 1 ? t - 2 : t - 1) * xline,
                              true, false);
                    SLOD_TMR(1);
                    gemm_Z(rop, tj, line);
                    SLOD_TMR(2);
                  }
              }
            if (t == nstp)
              break;
            if (t > 0 && !SLOD_DG(A, 32))
              put_step(t + 1, lane, 64); // bands the GJ wave needs after its next sweep
            SLOD_TMR(3);
            __syncthreads(); // A_t
            SLOD_TMR(4);
          }
        __syncthreads(); // M1 (also: both chains' last Z are in the workspace)
        __syncthreads(); // M2: V_mid is in the workspace
        if (chain == 0)
          {
            // R_mid = F_mid - B^T Z(mid-1) - B'^T Z(mid+1); the bands are the last B of each chain
            // (zero bands if a chain has no line)
            const double *B0 = Bbuf(Bc0, n0 - 1);
            double       *ob = ocb + 4 * PST + WROWS * WST + TSC + 3 * bsz; // other chain's Bc0
            const double *B1 = Bbuf(ob, n1 - 1);
            for (int tj = 0; tj < nct; ++tj)
              {
                build_rop(rop, tj, mid, B0, xg + (size_t)(n0 > 0 ? mid - 1 : mid) * xline, true, false);
                build_rop(rop, tj, mid, B1, xg + (size_t)(n1 > 0 ? mid + 1 : mid) * xline, false, true);
                gemm_Z(rop, tj, mid); // X_mid
              }
          }
        __syncthreads(); // M3
      }

    // ------------------------------ backward substitution -------------------------
    // from the meeting line outwards.  Wave (chain, w) owns the column tiles w, w + 2 of its chain:
    // X(line) = Z(line) - V(line) (B X(prev)), X(prev) kept in the accumulator layout and, for the
    // banded product, in a wave-private LDS strip with W zero rows above and below -- no workgroup
    // barrier, no re-read of X.
    if (nmy > 0 && !SLOD_DG(A, 16))
      {
        constexpr int XROWS = MP + 2 * W;
        const int     w2    = wave >> 1;
        double       *xs    = smem + (size_t)wave * (XROWS * XST); // aliases the forward buffers (dead now)
        for (int x = lane; x < W * XST; x += 64)
          {
            xs[x]                    = 0.0;
            xs[(MP + W) * XST + x] = 0.0;
          }
        // scalar problems: the coupling line -> prev of dof i is one stencil plane entry per offset o,
        // node(i) = nd0 + i * nds (coupling<S>() for vector-valued problems)
        const int nds = tr ? npx : 1;
        for (int tj = w2; tj < nct; tj += 2)
          {
            double4_t xa[NT]; // X(prev), then Z(line), then X(line)
            {
              const int      ol = olane(), og = ol >> 4, oc = ol & 15;
              const unsigned zl = (unsigned)(og * ncg + oc);
              const double  *xm = xg + (size_t)mid * xline + 16 * tj;
#pragma unroll
              for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  {
                    const double v = (xm + (16 * ti + 4 * r) * ncg)[zl];
                    xa[ti][r]      = (16 * ti + 4 * r + og < m && 16 * tj + oc < nc) ? v : 0.0;
                  }
            }
            double av[NB], an[NB];
            auto   load_V = [&](const double *vl, int ti, unsigned ul, double (&dst)[NB]) __attribute__((always_inline)) {
#pragma unroll
              for (int tk = 0; tk < NT; ++tk)
                if (16 * tk < m)
                  {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                      dst[4 * tk + q] = (vl + ((tk * NT + ti) * 4 + q) * 64)[ul];
                  }
            };
            load_V(vg + (size_t)line_of(chain, nmy - 1) * vline, 0, (unsigned)olane(), av);
            for (int t = nmy - 1; t >= 0; --t)
              {
                const int      line = line_of(chain, t);
                const double  *vl = vg + (size_t)line * vline;
                double        *xl = xg + (size_t)line * xline + 16 * tj;
                const int      ol = olane(), og = ol >> 4, oc = ol & 15;
                const unsigned ul = (unsigned)ol, zl = (unsigned)(og * ncg + oc), sl = (unsigned)(og * nds);
                const bool     cok = 16 * tj + oc < nc;
                double        *xsl = xs + og * XST + oc; // strip row 4 r + g - W ... of this lane
                SLOD_TMR(5);
                // X(prev) -> strip; then the registers take Z(line), the start of the accumulators
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                  for (int r = 0; r < 4; ++r)
                    xsl[(16 * ti + 4 * r + W) * XST] = xa[ti][r];
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                  for (int r = 0; r < 4; ++r)
                    {
                      const double v = (xl + (16 * ti + 4 * r) * ncg)[zl];
                      xa[ti][r]      = (16 * ti + 4 * r + og < m && cok) ? v : 0.0;
                    }
                SLOD_WAVE_SYNC();
                // B operand: -(B X(prev))[4 kk + g][col], B = coupling line -> prev (stencil planes)
                const double *bp[BW];
                const int     nd0 = tr ? (line + 1) + npx : 1 + (line + 1) * npx;
#pragma unroll
                for (int e = 0; e < BW; ++e)
                  {
                    const int o = e - W, dx = tr ? dl : o, dy = tr ? o : dl;
                    bp[e]       = st + (size_t)((dy + 1) * 3 + dx + 1) * A.nn_max + nd0;
                  }
                double yop[NB];
                // groups of k-steps: all band loads of a group first, a scheduling fence, then the FMAs
                constexpr int KGB = (S == 1) ? 4 : 1;
#pragma unroll
                for (int k0 = 0; k0 < NB; k0 += KGB)
                  {
                    double bev[KGB][BW];
#pragma unroll
                    for (int kq = 0; kq < KGB; ++kq)
#pragma unroll
                      for (int e = 0; e < BW; ++e)
                        {
                          const int kk = k0 + kq;
                          if (kk >= NB)
                            bev[kq][e] = 0.0;
                          else if (S == 1) // rows >= m: some finite entry of the (slack-padded) planes, dropped below
                            bev[kq][e] = (bp[e] + 4 * kk * nds)[sl];
                          else
                            bev[kq][e] = (4 * kk + og < m) ? coupling<S>(st, A.nn_max, npx, tr, m, line, 4 * kk + og, dl, e - W) : 0.0;
                        }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kq = 0; kq < KGB; ++kq)
                      {
                        const int kk = k0 + kq;
                        if (kk >= NB)
                          continue;
                        double v = 0.0;
#pragma unroll
                        for (int e = 0; e < BW; ++e) // rows outside [0,m) of the strip are zero: no range test
                          v = fma(-(sc * bev[kq][e]), xsl[(4 * kk + e) * XST], v);
                        yop[kk] = (4 * kk + og < m) ? v : 0.0;
                      }
                    __builtin_amdgcn_sched_barrier(0);
                  }
                SLOD_WAVE_SYNC(); // the strip is rewritten by the next line
                SLOD_TMR(6);
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
                  {
                    if (16 * ti >= m)
                      continue;
                    // next tile row of V: this line's, or the first one of the next line
                    if (ti + 1 < NT && 16 * (ti + 1) < m)
                      load_V(vl, ti + 1, ul, an);
                    else if (t > 0)
                      load_V(vg + (size_t)line_of(chain, t - 1) * vline, 0, ul, an);
#pragma unroll
                    for (int tk = 0; tk < NT; ++tk)
                      if (16 * tk + 16 <= m)
                        {
#pragma unroll
                          for (int q = 0; q < 4; ++q)
                            xa[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[4 * tk + q], yop[4 * tk + q], xa[ti], 0, 0, 0);
                        }
                      else if (16 * tk < m)
                        {
#pragma unroll
                          for (int q = 0; q < 4; ++q)
                            if (16 * tk + 4 * q < m)
                              xa[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[4 * tk + q], yop[4 * tk + q], xa[ti], 0, 0, 0);
                        }
#pragma unroll
                    for (int kk = 0; kk < NB; ++kk)
                      av[kk] = an[kk];
                  }
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                  for (int r = 0; r < 4; ++r)
                    if (16 * ti + 4 * r + og < m && cok)
                      (xl + (16 * ti + 4 * r) * ncg)[zl] = xa[ti][r];
              }
          }
      }
#ifdef SLOD_ENABLE_DIAG
    SLOD_TMR(7);
    if (SLOD_DG(A, (1 << 21)) && lane == 0 && A.nc_max * A.nc_max >= 48)
      for (int i = 0; i < 8; ++i)
        A.ms[(size_t)blockIdx.x * A.nc_max * A.nc_max + 16 + wave * 8 + i] = (double)tq[i];
#endif
    // Fused selection stage: the same workgroup goes on with M, D, the boundary trace, the
    // least squares, phi and psi of its patch.
    const bool stamp = (SLOD_DG(A, (1 << 20))) && tid == 0 && A.nc_max * A.nc_max >= 12;
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
#undef SLOD_WAVE_SYNC

} // namespace

int slod_solve_mf_tiles(int S, int m_max)
{
  const int nt = (m_max + 15) / 16;
  if (nt < 1 || nt > (S == 1 ? 7 : 5))
    return 0;
  return nt;
}

size_t slod_solve_mf_lds_bytes(int S, int m_max, int nc_max)
{
  // must mirror the carve-up at the top of k_solve_mf
  const int NT = slod_solve_mf_tiles(S, m_max);
  if (NT == 0)
    return ~(size_t)0;
  const int    W = 2 * S - 1, BW = 2 * W + 1, MP = 16 * NT, BWP = BW + 1;
  const int    bsz = ((MP + 2 * W) * BWP + 1) & ~1, PST = MP + 2, WST = MP + 2 * W + 1, WROWS = 16 + 3 * W;
  const size_t chsz = (size_t)((4 * PST + WROWS * WST + 16 * 17 + 6 * bsz + 1) & ~1);
  size_t       n    = 2 * chsz;
  const size_t back = (size_t)4 * (MP + 2 * W) * 17; // backward strips alias the chain blocks
  if (back > n)
    n = back;
  return ((n * sizeof(double) + 2 * (size_t)nc_max * sizeof(int)) + 15) & ~(size_t)15;
}

template <int NT, int S>
static hipError_t launch_mf_TS(const SlodKernelArgs &a, int n_patches, size_t lds, hipStream_t st)
{
  const void *fn = reinterpret_cast<const void *>(k_solve_mf<NT, S>);
  hipError_t  e  = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess)
    return e;
  if (a.debug)
    {
      int nb = 0;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256, lds);
      fprintf(stderr, "[slod] k_solve_mf<%d,%d>: %d patches, lds %zu B, occupancy %d blocks/CU\n", NT, S, n_patches,
              lds, nb);
    }
  hipLaunchKernelGGL((k_solve_mf<NT, S>), dim3(n_patches), dim3(256), lds, st, a);
  return hipGetLastError();
}

hipError_t slod_launch_solve_mf(int S, const SlodKernelArgs &a, int n_patches, size_t lds, hipStream_t st)
{
  const int nt = slod_solve_mf_tiles(S, a.m_max);
  if (S == 1)
    switch (nt)
      {
        case 1:
          return launch_mf_TS<1, 1>(a, n_patches, lds, st);
        case 2:
          return launch_mf_TS<2, 1>(a, n_patches, lds, st);
        case 3:
          return launch_mf_TS<3, 1>(a, n_patches, lds, st);
        case 4:
          return launch_mf_TS<4, 1>(a, n_patches, lds, st);
        case 5:
          return launch_mf_TS<5, 1>(a, n_patches, lds, st);
        case 6:
          return launch_mf_TS<6, 1>(a, n_patches, lds, st);
        case 7:
          return launch_mf_TS<7, 1>(a, n_patches, lds, st);
        default:
          return hipErrorInvalidValue;
      }
  switch (nt)
    {
      case 1:
        return launch_mf_TS<1, 2>(a, n_patches, lds, st);
      case 2:
        return launch_mf_TS<2, 2>(a, n_patches, lds, st);
      case 3:
        return launch_mf_TS<3, 2>(a, n_patches, lds, st);
      case 4:
        return launch_mf_TS<4, 2>(a, n_patches, lds, st);
      case 5:
        return launch_mf_TS<5, 2>(a, n_patches, lds, st);
      default:
        return hipErrorInvalidValue;
    }
}
