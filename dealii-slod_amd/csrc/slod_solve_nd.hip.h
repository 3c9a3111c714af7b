// k_solve_nd: patch solve by nested dissection (static condensation per cell + skeleton solve).
//
// Replaces the line-by-line block elimination of k_solve_tw (reference step: Gauss_elimination,
// LODtools.h:511-595, called at LOD.cc:546; what it computes is unchanged: X_I = A_II^{-1} P^T_I).
// The interior nodes of the patch are split into three levels (line coordinates: l = grid line,
// i = position in the line, lines along the shorter side; NV = cell size in fine elements):
//   level 0  the (NV-1)^2 interior nodes of each NV x NV cell: independent banded SPD blocks
//            (bandwidth NV).  One lane factorises a cell (LDL^T in registers); the condensation
//            is SIMT: wave = cell, lane = one of the 4 NV ring nodes of the cell (+ the cell's own
//            column of P^T), every lane runs the banded substitution on its own column with the
//            whole column in registers and the factor rows arriving in SGPRs (scalar loads: no
//            LDS traffic, no cross-lane traffic, the FMAs take the factor as scalar operand).
//   level 1  per cell row ("strip") the nodes on the vertical cell edges: <= 32 unknowns, dense
//            Gauss-Jordan inverse in one wave, Schur complement onto the horizontal lines by MFMA.
//   level 2  the horizontal skeleton lines: block tridiagonal with dense m x m blocks, Ca - 1
//            blocks instead of L: the dependent chain of the patch drops from L x m pivots
//            (C2: 39 x 39, twisted 20 x 40) to (NV-1)^2 + 28 + (Ca-1) m (C2: 49 + 28 + 156).
// Back substitution runs the levels in reverse; level 0 again SIMT (lane = column of P^T).
// All intermediate matrices live in the per-patch scratch slot (A.vinv); X goes to A.xs in the
// layout the selection stage reads.
#ifndef SLOD_SOLVE_ND_HIP_H
#define SLOD_SOLVE_ND_HIP_H
#include "slod_assemble.hip.h"

#ifndef ND_GB
#define ND_GB 3 // rows of a cell factor per scalar-load batch
#endif
#ifndef ND_WAVES
#define ND_WAVES 8 // 512-thread workgroups, two per CU: 80 KB of LDS hold the skeleton front of a patch
#endif
#ifndef ND_PHASES
#define ND_PHASES 0xffff // development switch (tools/): compile only some phases of k_solve_nd
#endif
namespace
{
  typedef const __attribute__((address_space(4))) double cdouble_t;

  constexpr int ND_BST = (ND_GB * 9 + 7) & ~7; // doubles per stored batch of factor rows (rows, then 1/d)

  // ---- scratch slot layout (doubles); host and device use the same function ---------------
  struct NdLayout
  {
    int    cblk;                                   // doubles per cell factor block
    size_t fac, cmat, gvec, y, yg, w, zs, p, total;
  };
  __host__ __device__ inline NdLayout nd_layout(int nv, int T, int m_max, int L_max, int nc_max)
  {
    NdLayout  l;
    const int nr = (nv - 1) * (nv - 1), ring = 4 * nv, MP = 8 * T, NEP = 32;
    const int Ca = (L_max + 1) / nv, Cb = (m_max + 1) / nv, ncell = Ca * Cb;
    l.cblk       = ((nr + ND_GB - 1) / ND_GB) * ND_BST; // batches of ND_GB rows: L entries, then 1/d, padded
    size_t o     = 0;
    auto   take  = [&](size_t n) { const size_t at = o; o += (n + 7) & ~(size_t)7; return at; };
    l.fac        = take((size_t)ncell * l.cblk);
    l.cmat       = take((size_t)ncell * ring * ring);
    l.gvec       = take((size_t)ncell * ring);
    l.y          = take((size_t)Ca * NEP * 2 * MP);  // Y_H of every strip (edge back substitution)
    l.yg         = take((size_t)Ca * NEP * nc_max);  // Y_G
    l.w          = take((size_t)Ca * MP * MP);       // W_a = V_a B_a of every line (backward sweep)
    l.zs         = take((size_t)Ca * MP * nc_max);   // Z_a
    l.p          = take(8);
    l.total      = o;
    return l;
  }

  // ---- ring of a cell: 0..NV bottom (l = l0-1, i = i0-1+j), NV+1..2NV+1 top (l = l0+NV-1),
  //      2NV+2..3NV left (i = i0-1, l = l0+q), 3NV+1..4NV-1 right (i = i0+NV-1) ---------------
  template <int NV>
  __host__ __device__ __forceinline__ constexpr int ring_side(int j)
  {
    return j <= NV ? 0 : (j <= 2 * NV + 1 ? 1 : (j <= 3 * NV ? 2 : 3));
  }
  template <int NV>
  __host__ __device__ __forceinline__ constexpr int ring_off(int j) // offset along the side (bottom/top: i - (i0-1); left/right: l - l0)
  {
    return j <= NV ? j : (j <= 2 * NV + 1 ? j - NV - 1 : (j <= 3 * NV ? j - 2 * NV - 2 : j - 3 * NV - 1));
  }
  // interior row (li*(NV-1)+ii) of the s-th interior neighbour of ring node j, -1 if outside the cell
  template <int NV>
  __host__ __device__ __forceinline__ constexpr int ring_nb(int j, int s)
  {
    const int N1 = NV - 1, side = ring_side<NV>(j), off = ring_off<NV>(j);
    if (side < 2)
      {
        const int ii = off - 2 + s; // node i = i0-1+off; neighbours i-1, i, i+1 -> ii = off-2+s
        return (ii >= 0 && ii < N1) ? (side == 0 ? 0 : N1 - 1) * N1 + ii : -1;
      }
    const int li = off - 1 + s;
    return (li >= 0 && li < N1) ? li * N1 + (side == 2 ? 0 : N1 - 1) : -1;
  }

  // compile-time loop: the index arrives as a type, so every table look-up that depends on it
  // (ring_nb, hi_slot, nd_ring_row) is a constant expression whatever the optimiser's inlining budget
  template <int I>
  struct NdIdx
  {
    static constexpr int value = I;
  };
  template <int B, int E, class F>
  __device__ __forceinline__ void nd_static_for(F &&f)
  {
    if constexpr (B < E)
      {
        f(NdIdx<B>{});
        nd_static_for<B + 1, E>(f);
      }
  }

  // A batch of the cell factor in scalar registers: rows GB b .. GB b + GB - 1, L[r][r-k] at l[r - GB b][k-1],
  // 1/d_r at d[r - GB b].  Storage is batch-major (ND_BST doubles per batch), so ONE opaque base and
  // immediate offsets fetch it; the asm keeps the (invariant) scalar loads behind this point of the
  // instruction stream.  Every scalar load of a wave retires behind one lgkmcnt(0): one wait per batch.
  struct NdRow
  {
    double v[8];
  };
  // Pull 16 cache lines (1 KB) at p towards L2, no wait: lane k < 16 loads one dword of line k into the
  // caller's dummy register (an in/out operand, so the register stays reserved while the load is in
  // flight; hidden loads only make the compiler's own vmcnt waits stronger, never weaker).  The factor
  // rows are then scalar-loaded from L2 (~270 cycles under load) instead of from HBM (~2 us).
  __device__ __forceinline__ void nd_prefetch_1k(const double *p, int lane, int &dummy)
  {
    const char *q = reinterpret_cast<const char *>(p) + (lane < 16 ? lane : 15) * 64;
    asm volatile("global_load_dword %0, %1, off" : "+v"(dummy) : "v"(q) : "memory");
  }
  __device__ __forceinline__ const double *nd_opaque(const double *p)
  {
    asm volatile("" : "+s"(p));
    return p;
  }
  template <int N>
  __device__ __forceinline__ NdRow nd_load_row(const double *p)
  {
    cdouble_t *q = (cdouble_t *)p;
    NdRow      r;
#pragma unroll
    for (int k = 0; k < N; ++k)
      r.v[k] = q[k];
#pragma unroll
    for (int k = N; k < 8; ++k)
      r.v[k] = 0.0;
    return r;
  }
  __host__ __device__ constexpr int nd_fac_l(int r, int k) { return (r / ND_GB) * ND_BST + (r % ND_GB) * 8 + (k - 1); } // L[r][r-k]
  __host__ __device__ constexpr int nd_fac_d(int r) { return (r / ND_GB) * ND_BST + 8 * ND_GB + (r % ND_GB); }         // 1/d_r

  template <int NV>
  __host__ __device__ constexpr bool nd_ring_row(int r)
  {
    const int N1 = NV - 1, li = r / N1, ii = r - li * N1;
    return li == 0 || li == N1 - 1 || ii == 0 || ii == N1 - 1;
  }
  // rows of a cell column that wait in LDS between the forward and the backward sweep (the others
  // stay in registers): with all (NV-1)^2 = 49 rows in registers the SIMT phases need 100+ VGPRs
  template <int NV>
  struct NdCell
  {
    static constexpr int NR = (NV - 1) * (NV - 1), HR = NR > 32 ? 24 : 0, NHI = NR - HR;
    // ring rows >= HR keep their values in a register array: slot of row r, number of such rows
    static constexpr int hi_slot(int r)
    {
      int n = 0;
      for (int q = HR; q < r; ++q)
        n += nd_ring_row<NV>(q) ? 1 : 0;
      return n;
    }
    static constexpr int NHR = hi_slot(NR);
  };

  // y = A_cc^{-1} b for one cell, one column per lane.  fac: the cell's factor block (wave-uniform):
  // rows L[r][.] (8 doubles each), then 1/d_r.  b_of(r): right-hand side of row r (r a compile-time
  // constant after unrolling); emit(r, y_r): called once per row, last row first; park: this wave's
  // LDS rows [HR][pst], column `lane`.  Scalar loads come in batches of four rows: every scalar load
  // of a wave retires behind one lgkmcnt(0), so one wait per batch, then 4 x 8 FMAs with the factor
  // entries as scalar operands (no LDS or cross-lane traffic for the factor).
  template <int NV, class FB, class FE>
  __device__ __forceinline__ void nd_band_solve(const double *fac, const double *fac_next, double *park, int pst, int lane,
                                                int lane_pf, int &pf, FB b_of, FE emit)
  {
    constexpr int NR = NdCell<NV>::NR, HR = NdCell<NV>::HR, NHI = NdCell<NV>::NHI, BW = NV, GB = ND_GB;
    constexpr int NB = (NR + GB - 1) / GB;
    double        u[NR];   // forward values, each live for BW rows only
    double        vh[NHI > 0 ? NHI : 1]; // D^-1 u, then the backward accumulators / y of rows >= HR
    // ---- forward: L u = b (dot form over the rows of L), v = D^-1 u
    static_assert(GB == 3, "three named rows per batch");
    static_assert(ND_BST == 32, "prefetch granularity: four batches per KB");

    nd_static_for<0, NB>([&](auto B) __attribute__((always_inline)) {
      constexpr int b = decltype(B)::value, r0 = GB * b, rows = NR - r0 < GB ? NR - r0 : GB;
      if constexpr (b % 4 == 0 && b + 8 < NB) // batches b+8 .. b+11: needed ~1.5 us from now
        nd_prefetch_1k(fac + (b + 8) * ND_BST, lane_pf, pf);
      const double *pb = nd_opaque(fac + b * ND_BST);
      const NdRow   ra = nd_load_row<8>(pb), rb = rows > 1 ? nd_load_row<8>(pb + 8) : ra,
                  rc = rows > 2 ? nd_load_row<8>(pb + 16) : ra, rd = nd_load_row<GB>(pb + 8 * GB);
      __builtin_amdgcn_sched_barrier(0); // the loads of the batch go out together: one wait per batch
      nd_static_for<0, rows>([&](auto Q) __attribute__((always_inline)) {
        constexpr int q = decltype(Q)::value, r = r0 + q;
        const NdRow  &lr = q == 0 ? ra : (q == 1 ? rb : rc);
        double        acc = b_of(NdIdx<r>{});
        nd_static_for<1, BW + 1>([&](auto K) __attribute__((always_inline)) {
          constexpr int k = decltype(K)::value;
          if constexpr (r - k >= 0)
            acc = fma(-lr.v[k - 1], u[r - k], acc);
        });
        u[r] = acc;
        const double v = acc * rd.v[q];
        if constexpr (r < HR)
          park[r * pst + lane] = v;
        else
          vh[r - HR] = v;
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    // ---- backward: L^T y = v (axpy form, again over the rows of L), same batches, last first
    static_assert(HR == 0 || HR + BW + 2 * GB <= NR, "parked rows must lie below the first two backward batches");
    double tl[HR > 0 ? HR : 1]; // accumulators of the parked rows, loaded one batch before they are touched
    nd_static_for<0, NB>([&](auto B) __attribute__((always_inline)) {
      constexpr int b = NB - 1 - decltype(B)::value, r0 = GB * b, rows = NR - r0 < GB ? NR - r0 : GB;
      if constexpr (b == NB - 1 || b == NB - 2) // the first eight batches of the wave's next cell
        nd_prefetch_1k(fac_next + (b == NB - 1 ? 0 : 4) * ND_BST, lane_pf, pf);
      const double *pb = nd_opaque(fac + b * ND_BST);
      const NdRow   ra = nd_load_row<8>(pb), rb = rows > 1 ? nd_load_row<8>(pb + 8) : ra,
                  rc = rows > 2 ? nd_load_row<8>(pb + 16) : ra;
      __builtin_amdgcn_sched_barrier(0);
      // parked rows that the rows of the NEXT batch (r0 - GB .. r0 - 1) are the first to update
      nd_static_for<0, GB>([&](auto Q) __attribute__((always_inline)) {
        constexpr int e = r0 - GB + decltype(Q)::value - BW;
        if constexpr (e >= 0 && e < HR)
          tl[e] = park[e * pst + lane];
      });
      nd_static_for<0, rows>([&](auto Q) __attribute__((always_inline)) {
        constexpr int q = rows - 1 - decltype(Q)::value, r = r0 + q;
        const NdRow  &lr = q == 0 ? ra : (q == 1 ? rb : rc);
        double        y;
        if constexpr (r < HR)
          y = tl[r];
        else
          y = vh[r - HR];
        emit(NdIdx<r>{}, y);
        nd_static_for<1, BW + 1>([&](auto K) __attribute__((always_inline)) {
          constexpr int k = decltype(K)::value;
          if constexpr (r - k >= 0)
            {
              if constexpr (r - k < HR)
                tl[r - k] = fma(-lr.v[k - 1], y, tl[r - k]);
              else
                vh[r - k - HR] = fma(-lr.v[k - 1], y, vh[r - k - HR]);
            }
        });
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  }

  // Gauss-Jordan sweep of a symmetric positive definite mp x mp matrix held by one wave: 8 x 8 lane
  // grid, T x T contiguous tile per lane, pivot row through a wave-private LDS line (k_solve_tw's
  // scheme).  On return the tile holds -S^{-1} in its leading mm x mm block.
  template <int T>
  __device__ __forceinline__ void nd_gj_sweep(double (&a)[T][T], double *rowb, int mm, int lane, bool &bad)
  {
    const int gy = lane >> 3, gx = lane & 7;
    for (int ka = 0; ka * T < mm; ++ka)
      {
#pragma unroll
        for (int a0 = 0; a0 < T; ++a0)
          {
            const int k = T * ka + a0;
            if (k >= mm) // wave-uniform
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
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ta = 0; ta < T; ++ta)
#pragma unroll
              for (int tb = 0; tb < T; ++tb)
                a[ta][tb] = fma(-ri[ta], sj[tb], a[ta][tb]);
            if (gy == ka)
              {
#pragma unroll
                for (int tb = 0; tb < T; ++tb)
                  a[a0][tb] = sj[tb];
              }
            if (gx == ka)
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
  }

  // 16 x 16 output tile of a product on the fp64 matrix pipe; fa(row, k), fb(k, col) fetch the
  // operands (0 outside their range); K = 4 KT.  D[(lane>>4) + 4 r][lane & 15], r < 4.
  template <int KT, class FA, class FB>
  __device__ __forceinline__ double4_t nd_mfma_tile(FA fa, FB fb, int ti, int tj, int lane)
  {
    const int row = 16 * ti + (lane & 15), col = 16 * tj + (lane & 15), kq = lane >> 4;
    double    av[KT], bv[KT];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk)
      {
        av[kk] = fa(row, 4 * kk + kq);
        bv[kk] = fb(4 * kk + kq, col);
      }
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < KT; ++kk)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv[kk], acc, 0, 0, 0);
    return acc;
  }

  // The kernel arguments, re-read from the kernarg segment through an opaque pointer: every phase of
  // k_solve_nd starts from this copy, so no scalar derived from the arguments stays live across the
  // SIMT phases (they need the SGPRs for the factor rows; ~80 long-lived scalars made the allocator
  // park every loaded row in VGPR lanes).
  __device__ __forceinline__ SlodKernelArgs nd_args()
  {
    auto p = (const __attribute__((address_space(4))) SlodKernelArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
#if defined(__HIP_DEVICE_COMPILE__)
    return *p;
#else
    return SlodKernelArgs();
#endif
  }

// locals of a phase of k_solve_nd (NV, T, smem, tid, wave, lane_ in scope)
#define ND_LOCALS                                                                                                      \
  const SlodKernelArgs A = nd_args();                                                                                  \
  const SlodPatchDesc  d = A.desc[blockIdx.x];                                                                         \
  const int            m = d.m, L = d.L, nc = d.n_c, n = A.n_sub, ncg = A.nc_max;                                      \
  const bool           tr  = (d.flags & SLOD_F_TRANSPOSED) != 0;                                                       \
  const int            npx = d.nx + 1;                                                                                 \
  const int            Ca = (L + 1) / NV, Cb = (m + 1) / NV, ncell = Ca * Cb, nH = Ca - 1, NE = (Cb - 1) * N1;         \
  const double        *st  = A.st + (size_t)blockIdx.x * A.st_stride;                                                  \
  double              *ws  = A.vinv + (size_t)blockIdx.x * A.v_stride;                                                 \
  double              *xg  = A.xs + (size_t)blockIdx.x * A.x_stride;                                                   \
  const size_t         xline = (size_t)A.m_max * ncg;                                                                  \
  const NdLayout       lay = nd_layout(NV, T, A.m_max, A.L_max, A.nc_max);                                             \
  double *const        facg = ws + lay.fac, *const cmat = ws + lay.cmat, *const gvec = ws + lay.gvec;                  \
  double *const        yh = ws + lay.y, *const yg = ws + lay.yg;                                                       \
  double *const        wm = ws + lay.w, *const zs = ws + lay.zs;                                                       \
  double *const        pm = ws + lay.p; /* one dead word: store target of idle lanes */                               \
  (void)nc; (void)n; (void)ncell; (void)nH; (void)NE; (void)xline; (void)facg; (void)cmat; (void)gvec;                 \
  (void)yh; (void)yg; (void)wm; (void)zs; (void)pm; (void)Ca; (void)Cb; (void)npx;                                     \
  /* A[(l,i),(l+dl,i+o)], both nodes interior to the patch */                                                          \
  auto cpl = [&](int l, int i, int dl, int o) -> double {                                                              \
    return coupling<1>(st, A.nn_max, npx, tr, m, l, i, dl, o);                                                         \
  };                                                                                                                   \
  auto live = [&](int l, int i) { return l >= 0 && l < L && i >= 0 && i < m; };                                        \
  auto xrow = [&](int l, int i) -> double * { return xg + (size_t)l * xline + (size_t)i * ncg; };                      \
  /* column of P^T that owns the interior of cell (a, b) */                                                            \
  auto own_col = [&](int a, int b) {                                                                                   \
    const int ka = (a * NV) / n, kb = (b * NV) / n;                                                                    \
    const int kx = tr ? ka : kb, ky = tr ? kb : ka;                                                                    \
    const int t = kx * d.my + ky, c0 = d.ccx * d.my + d.ccy;                                                           \
    return t == c0 ? 0 : (t < c0 ? t + 1 : t);                                                                         \
  };                                                                                                                   \
  /* ring node j of cell (a,b) */                                                                                      \
  auto ring_node = [&](int a, int b, int j, int &l, int &i) {                                                          \
    const int l0 = a * NV, i0 = b * NV, side = ring_side<NV>(j), off = ring_off<NV>(j);                                \
    l = side == 0 ? l0 - 1 : (side == 1 ? l0 + NV - 1 : l0 + off);                                                     \
    i = side < 2 ? i0 - 1 + off : (side == 2 ? i0 - 1 : i0 + NV - 1);                                                  \
  };                                                                                                                   \
  /* inverse of ring_node: position of skeleton node (l, i) in the ring of cell (a, b) */                              \
  auto ring_index = [&](int a, int b, int l, int i) {                                                                  \
    const int l0 = a * NV, i0 = b * NV;                                                                                \
    return l == l0 - 1 ? i - (i0 - 1)                                                                                  \
                       : (l == l0 + NV - 1 ? NV + 1 + i - (i0 - 1) : (i == i0 - 1 ? 2 * NV + 2 + l - l0 : 3 * NV + 1 + l - l0)); \
  };                                                                                                                   \
  (void)cpl; (void)live; (void)xrow; (void)own_col; (void)ring_node; (void)ring_index;

  template <int NV, int T>
  __global__ __launch_bounds__(64 * ND_WAVES, 4) void k_solve_nd(const SlodKernelArgs Akern)
  {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    (void)Akern; // read through nd_args()
    constexpr int       N1 = NV - 1, NR = N1 * N1, RING = 4 * NV, MP = 8 * T, NEP = 32, MP2 = 2 * MP;
    constexpr int       KTM = MP / 4; // k-steps over a line
    constexpr int       NW = ND_WAVES, NT = 64 * NW; // waves / threads of the workgroup
    const int           tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = tid & 63;
    const int           lane_ = lane;
    // LDS: per wave a pivot row [MP] and the ring couplings of the wave's current cell [RING][4]
    using CL = NdCell<NV>;
    constexpr int PST = RING + 2; // lanes per parked row; the last one takes the writes of the idle lanes
    double *rowb = smem + wave * (MP + RING * 4 + CL::HR * PST);
    double *kt   = rowb + MP;       // [RING][4]
    double *park = kt + RING * 4;   // [HR][PST] rows of the cell columns between the two sweeps

    bool bad = false;
    // per-patch timeline (SLOD_ENABLE_DIAG builds, tools/nd_timeline.py): 100 MHz clock into ms[32 + k]
    auto stamp = [&](int k) {
#ifdef SLOD_ENABLE_DIAG
      const SlodKernelArgs A = nd_args();
      if ((SLOD_DG(A, (1 << 20))) && tid == 0 && A.nc_max * A.nc_max >= 48)
        A.ms[(size_t)blockIdx.x * A.nc_max * A.nc_max + 32 + k] = (double)wall_clock64();
#else
      (void)k;
#endif
    };
    stamp(0);
    {
    ND_LOCALS
    // ------------------------------ stencil planes of the patch --------------------------------
    if ((ND_PHASES & 512) && A.fuse_assemble)
      {
        for (int node = tid; node < npx * (d.ny + 1); node += NT)
          assemble_node<1>(A, d, blockIdx.x, node);
        __syncthreads();
      }

    }
    stamp(1);
    {
    ND_LOCALS
    // ------------------------------ level 0: cell factors ---------------------------------------
    // Banded LDL^T of every cell, NV + 1 lanes per cell: the lane that holds column C of the band
    // (C mod (NV+1) = its position in the group) keeps a[C..C+NV][C] in registers.  Per pivot the
    // owner of column r scales it and publishes (d, l_1..l_NV) through a small LDS line; the other
    // lanes of the group read the l_k they need and update their own column.  No shared window, no
    // unrolling over the (NV-1)^2 pivots: small code, few registers.
    if (ND_PHASES & 1)
      {
        constexpr int GL = NV + 1, CPW = 64 / GL; // lanes per cell, cells per wave
        double       *lb = smem + wave * ((CPW + 1) * 2 * GL); // (one spare line: the lanes past the last group)
        for (int c0 = 0; c0 < ncell; c0 += NW * CPW)
          {
            const int  grp = lane / GL, q0 = lane - grp * GL;
            const int  c = c0 + wave * CPW + grp;
            const bool act = grp < CPW && c < ncell;
            const int  cc = act ? c : 0;
            const int  a = cc / Cb, b = cc - a * Cb, l0 = a * NV, i0 = b * NV;
            double    *fac = facg + (size_t)cc * lay.cblk;
            double    *lg = lb + grp * (2 * GL); // [0] d, [1..NV] l, [NV+1 ..] zeros
            // band entries of column C: rows C (diag), C+1, C+N1-1, C+N1, C+N1+1 -> offsets 0, 1, N1-1, N1, N1+1
            auto load_col = [&](int C, double (&e)[5]) {
              const int  li = C / N1, ii = C - li * N1;
              const bool ok = act && C < NR;
              e[0] = ok ? cpl(l0 + li, i0 + ii, 0, 0) : 0.0;
              e[1] = (ok && ii + 1 < N1) ? cpl(l0 + li, i0 + ii, 0, 1) : 0.0;
              e[2] = (ok && li + 1 < N1 && ii >= 1) ? cpl(l0 + li, i0 + ii, 1, -1) : 0.0;
              e[3] = (ok && li + 1 < N1) ? cpl(l0 + li, i0 + ii, 1, 0) : 0.0;
              e[4] = (ok && li + 1 < N1 && ii + 1 < N1) ? cpl(l0 + li, i0 + ii, 1, 1) : 0.0;
            };
            auto expand = [&](const double (&e)[5], double (&col)[GL]) {
#pragma unroll
              for (int p = 0; p < GL; ++p)
                col[p] = 0.0;
              col[0] = e[0];
              col[1] = e[1];
              // N1 - 1 may coincide with 1 (NV = 3): entries of different nodes never both non-zero
              col[N1 - 1] += e[2];
              col[N1] = e[3];
              col[N1 + 1] = e[4];
            };
            double col[GL], nxt[5];
            {
              double e[5];
              load_col(q0, e);
              expand(e, col);
              load_col(q0 + GL, nxt);
            }
            if (q0 + 1 + NV < 2 * GL) // zero tail of the LDS line (read as l_k, k > NV)
              lg[GL + q0] = 0.0;
            int C = q0; // column this lane holds
            for (int r = 0; r < NR; ++r)
              {
                const bool owner = C == r;
                if (owner)
                  {
                    bad |= act && !(col[0] > 0.0);
                    const double dinv = fast_rcp(col[0]);
                    lg[0]             = col[0];
                    if (act)
                      fac[nd_fac_d(r)] = dinv;
#pragma unroll
                    for (int p = 1; p <= NV; ++p)
                      {
                        const double l = col[p] * dinv;
                        lg[p]          = l;
                        if (act && r + p < NR)
                          fac[nd_fac_l(r + p, p)] = l;
                      }
                  }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (!owner)
                  {
                    const int    q = C - r; // 1..NV
                    const double f = lg[q] * lg[0];
#pragma unroll
                    for (int p = 0; p < NV; ++p) // row C + p <= r + NV
                      col[p] = fma(-lg[q + p], f, col[p]);
                  }
                else
                  {
                    // next column of this lane: r + NV + 1, its entries were fetched NV + 1 pivots ago
                    C += GL;
                    expand(nxt, col);
                    load_col(C + GL, nxt);
                  }
                __builtin_amdgcn_wave_barrier(); // the line is rewritten by the next owner
              }
          }
      }

    }
    stamp(2);
    {
    ND_LOCALS
    __syncthreads(); // the cell factors are in the scratch slot (and still in L2)
    // ------------------------------ level 0: condensation ---------------------------------------
    // wave = cell, lane j < RING = ring node j (unit boundary value), lane RING = the cell's own
    // column of P^T.  y = A_cc^{-1} (own f - A_cs xs); output column: A_sc y.
    int pf = 0; // target register of the prefetch loads
    if ((ND_PHASES & 4) && wave < ncell)
      {
        nd_prefetch_1k(facg + (size_t)wave * lay.cblk, lane_, pf);
        nd_prefetch_1k(facg + (size_t)wave * lay.cblk + 4 * ND_BST, lane_, pf);
      }
    if (ND_PHASES & 4)
    for (int c = wave; c < ncell; c += NW)
      {
        const int     a = c / Cb, b = c - a * Cb, l0 = a * NV, i0 = b * NV;
        const double *fac = facg + (size_t)c * lay.cblk, *fac_next = facg + (size_t)(c + NW < ncell ? c + NW : c) * lay.cblk;
        // opaque copy of the lane id: everything derived from it is recomputed per cell instead of
        // being hoisted out of the loop and spilled (LICM would keep ~100 per-lane values alive)
        int lane = lane_;
        asm volatile("" : "+v"(lane));
        const int pl = lane < PST ? lane : PST - 1;
        __builtin_amdgcn_wave_barrier(); // the previous cell's readers of kt are done
        if (lane < RING)
          {
            int l, i;
            ring_node(a, b, lane, l, i);
            const int  side = ring_side<NV>(lane);
            const bool lv = live(l, i);
#pragma unroll
            for (int s = 0; s < 3; ++s)
              {
                // neighbour s: bottom/top (l -+ 1 towards the cell, i-1+s); left/right (l-1+s, i +- 1)
                const int dl = side == 0 ? 1 : (side == 1 ? -1 : s - 1);
                const int o  = side < 2 ? s - 1 : (side == 2 ? 1 : -1);
                const int ln = l + dl - l0, in = i + o - i0;
                double    v  = 0.0;
                if (lv && ln >= 0 && ln < N1 && in >= 0 && in < N1)
                  v = cpl(l, i, dl, o);
                kt[lane * 4 + s] = v;
              }
          }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const bool own = lane == RING;
        // right-hand side of this lane's column: -A_cs e_lane has the lane's own <= 3 couplings at
        // the interior rows next to its ring node (the own column: f = h^2 everywhere)
        double    kv[3];
        int       kr[3];
#pragma unroll
        for (int s = 0; s < 3; ++s)
          {
            kr[s] = lane < RING ? ring_nb<NV>(lane, s) : -1;
            kv[s] = -kt[(lane < RING ? lane : 0) * 4 + s];
          }
        const double f0 = own ? 4.0 * A.scale : 0.0;
        auto         b_of = [&](auto R) -> double {
          constexpr int r = decltype(R)::value;
          double        v = f0;
          if constexpr (nd_ring_row<NV>(r))
            {
              v = kr[0] == r ? kv[0] : v;
              v = kr[1] == r ? kv[1] : v;
              v = kr[2] == r ? kv[2] : v;
            }
          return v;
        };
        double yhr[CL::NHR]; // y of the ring rows >= HR (the others go back to their LDS row)
        auto   emit = [&](auto R, double y) {
          constexpr int r = decltype(R)::value;
          if constexpr (nd_ring_row<NV>(r))
            {
              if constexpr (r < CL::HR)
                park[r * PST + pl] = y;
              else
                yhr[CL::hi_slot(r)] = y;
            }
        };
        nd_band_solve<NV>(fac, fac_next, park, PST, pl, lane, pf, b_of, emit);
        double *cm = cmat + (size_t)c * RING * RING, *gv = gvec + (size_t)c * RING;
        nd_static_for<0, RING>([&](auto I) {
          constexpr int i = decltype(I)::value;
          double        v = 0.0;
          nd_static_for<0, 3>([&](auto S) {
            constexpr int nb = ring_nb<NV>(i, decltype(S)::value);
            if constexpr (nb >= 0)
              {
                double y;
                if constexpr (nb < CL::HR)
                  y = park[nb * PST + pl];
                else
                  y = yhr[CL::hi_slot(nb)];
                v = fma(kt[i * 4 + decltype(S)::value], y, v);
              }
          });
          // lanes < RING: column of the cell matrix; the own lane: condensed right-hand side
          double *dst = lane < RING ? cm + i * RING + lane : gv + i;
          if (lane <= RING)
            *dst = lane < RING ? v : -v;
          if constexpr ((i & 3) == 3)
            __builtin_amdgcn_sched_barrier(0); // keep the LDS reads of the later rows behind these stores
        });
      }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf) : : "memory"); // the prefetch loads have landed: pf's register is free
    __syncthreads();

    }
    stamp(3);
    {
    ND_LOCALS
    // ------------------------------ right-hand sides of the skeleton nodes -----------------------
    // rows of X of the skeleton nodes <- P^T rows (LODtools.h:24-67: (h^2/4) {1,2,4}) plus the condensed
    // right-hand sides of the (at most four) cells around the node, added in a fixed order
    if (ND_PHASES & 2)
    {
      const int nsk = nH * m + Ca * NE;
      for (int idx = tid; idx < nsk * nc; idx += NT)
        {
          const int s = idx / nc, k = idx - s * nc;
          int       l, i;
          if (s < nH * m)
            {
              const int a = s / m;
              l = a * NV + NV - 1;
              i = s - a * m;
            }
          else
            {
              const int t = s - nH * m, a = t / NE, e = t - a * NE, be = e / N1;
              l = a * NV + (e - be * N1);
              i = be * NV + NV - 1;
            }
          const int ix = tr ? l + 1 : i + 1, iy = tr ? i + 1 : l + 1;
          double    v  = A.scale * pt_weight<1>(d, n, A.quirk, ix, iy, 0, k);
          const int a_lo = l / NV, a_hi = (l + 1) / NV, b_lo = i / NV, b_hi = (i + 1) / NV; // cells whose ring holds the node
          for (int a = a_lo; a <= a_hi && a < Ca; ++a)
            for (int b = b_lo; b <= b_hi && b < Cb; ++b)
              if (own_col(a, b) == k)
                v += gvec[(size_t)(a * Cb + b) * RING + ring_index(a, b, l, i)];
          xrow(l, i)[k] = v;
        }
    }
    __syncthreads();
    }

    stamp(4);
    {
    ND_LOCALS
    // ------------------------------ skeleton: strip by strip, in LDS ----------------------------
    // Frontal sweep over the cell rows.  LDS holds the front only: three m x m blocks (T of the line
    // below the strip, the coupling B between the two lines of the strip, T of the line above), the
    // edge-set matrix of the strip and its coupling / right-hand-side block.  Per strip: gather the
    // cell matrices (global, read once), invert the edge block (one wave, registers), Y = V_EE R and
    // the Schur complement R_H^T Y straight from the accumulators (the D fragment of a 16-row tile IS
    // the B fragment of k-steps 4 ti .. 4 ti + 3), eliminate the finished line below (Gauss-Jordan in
    // one wave, W = V B and P = B^T W again accumulator to operand).  Only Y, W and Z leave the CU.
    if (ND_PHASES & 8)
      {
        constexpr int LDT = MP + 1, LDE = NEP + 1, TH = MP2 / 16, TI = (MP + 15) / 16;
        const int     NCP = (ncg + 15) & ~15, RC = MP2 + NCP, TG = NCP / 16; // right-hand sides in whole column tiles
        static_assert(MP2 % 16 == 0, "line pair in whole column tiles");
        double *TT = smem, *EE = TT + 3 * MP * LDT, *RR = EE + NEP * LDE, *rowg = RR + NEP * RC;
        int     sPrev = 0, sB = 1, sCur = 2;
        const int gy = lane >> 3, gx = lane & 7, r16 = lane & 15, kq = lane >> 4;
        for (int a = 0; a < Ca; ++a)
          {
            const bool has_bot = a > 0, has_top = a < Ca - 1;
            double    *Tp = TT + sPrev * MP * LDT, *Bb = TT + sB * MP * LDT, *Tc = TT + sCur * MP * LDT;
            // (1) direct couplings (stencil entries between skeleton nodes), right-hand sides of the edges
            for (int idx = tid; idx < NEP * NEP; idx += NT)
              {
                const int e1 = idx / NEP, e2 = idx - e1 * NEP;
                double    v = 0.0;
                if (e1 < NE && e2 < NE)
                  {
                    const int b1 = e1 / N1, q1 = e1 - b1 * N1, b2 = e2 / N1, q2 = e2 - b2 * N1;
                    if (b1 == b2 && q2 - q1 <= 1 && q1 - q2 <= 1)
                      v = cpl(a * NV + q1, b1 * NV + NV - 1, q2 - q1, 0);
                  }
                EE[e1 * LDE + e2] = v;
              }
            for (int idx = tid; idx < NEP * RC; idx += NT)
              {
                const int e = idx / RC, h = idx - e * RC;
                double    v = 0.0;
                if (e < NE)
                  {
                    const int be = e / N1, q = e - be * N1, ie = be * NV + NV - 1;
                    if (h < MP2)
                      {
                        const int top = h >= MP, pos = top ? h - MP : h;
                        if (pos < m && pos - ie <= 1 && ie - pos <= 1)
                          {
                            if (!top && q == 0 && has_bot)
                              v = cpl(a * NV, ie, -1, pos - ie);
                            if (top && q == N1 - 1 && has_top)
                              v = cpl(a * NV + q, ie, 1, pos - ie);
                          }
                      }
                    else if (h - MP2 < nc)
                      v = xrow(a * NV + q, ie)[h - MP2];
                  }
                RR[idx] = v;
              }
            if (has_top)
              for (int idx = tid; idx < MP * MP; idx += NT)
                {
                  const int i = idx / MP, j = idx - i * MP;
                  double    v = 0.0;
                  if (i < m && j < m && j - i <= 1 && i - j <= 1)
                    v = cpl(a * NV + NV - 1, i, 0, j - i);
                  Tc[i * LDT + j] = v;
                  Bb[i * LDT + j] = 0.0;
                }
            __syncthreads();
            // (2) the cell matrices of the strip; cells b and b + 1 share ring nodes: even b, then odd b
            for (int pb = 0; pb < 2; ++pb)
              {
                const int nbc = (Cb - pb + 1) / 2;
                for (int idx = tid; idx < nbc * RING * RING; idx += NT)
                  {
                    const int cc = idx / (RING * RING), ent = idx - cc * RING * RING;
                    const int b = 2 * cc + pb, c = a * Cb + b;
                    const int j1 = ent / RING, j2 = ent - j1 * RING;
                    int       l1, i1, l2, i2;
                    ring_node(a, b, j1, l1, i1);
                    ring_node(a, b, j2, l2, i2);
                    if (!live(l1, i1) || !live(l2, i2))
                      continue;
                    const int    s1 = ring_side<NV>(j1), s2 = ring_side<NV>(j2);
                    const double v = cmat[(size_t)c * RING * RING + ent];
                    if (s1 < 2 && s2 < 2)
                      {
                        if (s1 == s2)
                          (s1 == 0 ? Tp : Tc)[i1 * LDT + i2] += v;
                        else if (s1 == 0)
                          Bb[i1 * LDT + i2] += v;
                      }
                    else if (s1 >= 2 && s2 >= 2)
                      {
                        const int e1 = (s1 == 2 ? b - 1 : b) * N1 + (l1 - a * NV), e2 = (s2 == 2 ? b - 1 : b) * N1 + (l2 - a * NV);
                        EE[e1 * LDE + e2] += v;
                      }
                    else if (s1 >= 2)
                      {
                        const int e1 = (s1 == 2 ? b - 1 : b) * N1 + (l1 - a * NV);
                        RR[e1 * RC + (s2 == 1 ? MP : 0) + i2] += v;
                      }
                  }
                __syncthreads();
              }
            // (3) edge set: V_EE, Y = V_EE R, Schur complement onto the two lines
            if (NE > 0)
              {
                if (wave == 0)
                  {
                    double t4[4][4];
#pragma unroll
                    for (int p = 0; p < 4; ++p)
#pragma unroll
                      for (int q = 0; q < 4; ++q)
                        t4[p][q] = EE[(4 * gy + p) * LDE + 4 * gx + q];
                    nd_gj_sweep<4>(t4, rowg, NE, lane, bad);
#pragma unroll
                    for (int p = 0; p < 4; ++p)
#pragma unroll
                      for (int q = 0; q < 4; ++q)
                        EE[(4 * gy + p) * LDE + 4 * gx + q] = -t4[p][q];
                  }
                __syncthreads();
                for (int jt = wave; jt < TH + TG; jt += NW)
                  {
                    const int col = 16 * jt + r16;
                    double    bop[NEP / 4];
#pragma unroll
                    for (int kk = 0; kk < NEP / 4; ++kk)
                      bop[kk] = RR[(4 * kk + kq) * RC + col];
                    double4_t y[2];
#pragma unroll
                    for (int ti = 0; ti < 2; ++ti)
                      {
                        double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int kk = 0; kk < NEP / 4; ++kk)
                          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(EE[(16 * ti + r16) * LDE + 4 * kk + kq], bop[kk], acc, 0, 0, 0);
                        y[ti] = acc;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                          {
                            const int e = 16 * ti + kq + 4 * q;
                            if (jt < TH)
                              yh[(size_t)a * NEP * MP2 + e * MP2 + col] = acc[q];
                            else if (col - MP2 < ncg)
                              yg[(size_t)a * NEP * ncg + e * ncg + (col - MP2)] = acc[q];
                          }
                      }
                    // U[:, jt] = R_H^T Y[:, jt]: the accumulator registers of Y are the B fragments
                    for (int it = 0; it < TH; ++it)
                      {
                        if (jt < TH && 16 * it >= MP && 16 * jt + 15 < MP)
                          continue; // top x bottom only: the transpose of bottom x top
                        double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int kk = 0; kk < NEP / 4; ++kk)
                          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(RR[(4 * kk + kq) * RC + 16 * it + r16], y[kk >> 2][kk & 3], acc, 0, 0, 0);
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                          {
                            const int  r1 = 16 * it + kq + 4 * q;
                            const bool t1 = r1 >= MP;
                            const int  p1 = t1 ? r1 - MP : r1;
                            if (p1 >= m)
                              continue;
                            if (jt < TH)
                              {
                                const bool t2 = col >= MP;
                                const int  p2 = t2 ? col - MP : col;
                                if (p2 >= m)
                                  continue;
                                if (!t1 && !t2 && has_bot)
                                  Tp[p1 * LDT + p2] -= acc[q];
                                else if (t1 && t2 && has_top)
                                  Tc[p1 * LDT + p2] -= acc[q];
                                else if (!t1 && t2 && has_bot && has_top)
                                  Bb[p1 * LDT + p2] -= acc[q];
                              }
                            else if (col - MP2 < nc && (t1 ? has_top : has_bot))
                              xrow((t1 ? a : a - 1) * NV + NV - 1, p1)[col - MP2] -= acc[q];
                          }
                      }
                  }
                __syncthreads();
              }
            // (4) the line below the strip is complete: eliminate it
            if (has_bot)
              {
                const int la = (a - 1) * NV + NV - 1;
                if (wave == 0)
                  {
                    double tt[T][T];
#pragma unroll
                    for (int p = 0; p < T; ++p)
#pragma unroll
                      for (int q = 0; q < T; ++q)
                        tt[p][q] = Tp[(T * gy + p) * LDT + T * gx + q];
                    nd_gj_sweep<T>(tt, rowg, m, lane, bad);
#pragma unroll
                    for (int p = 0; p < T; ++p)
#pragma unroll
                      for (int q = 0; q < T; ++q)
                        Tp[(T * gy + p) * LDT + T * gx + q] = -tt[p][q];
                  }
                __syncthreads();
                // column tiles: [0, TI) of W = V B (and P = B^T W into T of the line above), then the
                // tiles of Z = V G (and G of the line above -= B^T Z)
                const int tjn = (nc + 15) >> 4;
                for (int jt = wave; jt < (has_top ? TI : 0) + tjn; jt += NW)
                  {
                    const bool isw = has_top && jt < TI;
                    const int  jz = jt - (has_top ? TI : 0), col = 16 * (isw ? jt : jz) + r16;
                    double     bop[KTM];
#pragma unroll
                    for (int kk = 0; kk < KTM; ++kk)
                      {
                        const int k = 4 * kk + kq;
                        if (isw)
                          bop[kk] = col < MP ? Bb[k * LDT + col] : 0.0;
                        else
                          bop[kk] = (k < m && col < nc) ? xrow(la, k)[col] : 0.0;
                      }
                    double4_t w[TI];
#pragma unroll
                    for (int ti = 0; ti < TI; ++ti)
                      {
                        double4_t acc = {0.0, 0.0, 0.0, 0.0};
                        const int row = 16 * ti + r16 < MP ? 16 * ti + r16 : MP - 1;
#pragma unroll
                        for (int kk = 0; kk < KTM; ++kk)
                          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Tp[row * LDT + 4 * kk + kq], bop[kk], acc, 0, 0, 0);
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                          {
                            const int rr = 16 * ti + kq + 4 * q;
                            if (rr >= m) // rows past the line are no unknowns: zero operand below
                              acc[q] = 0.0;
                            if (rr < MP)
                              {
                                if (isw && col < MP)
                                  wm[(size_t)(a - 1) * MP * MP + rr * MP + col] = acc[q];
                                else if (!isw && col < ncg)
                                  zs[(size_t)(a - 1) * MP * ncg + rr * ncg + col] = acc[q];
                              }
                          }
                        w[ti] = acc;
                      }
                    if (has_top)
                      for (int it = 0; it < TI; ++it)
                        {
                          double4_t acc = {0.0, 0.0, 0.0, 0.0};
                          const int rowi = 16 * it + r16;
#pragma unroll
                          for (int kk = 0; kk < 4 * TI; ++kk)
                            {
                              const int    k = 4 * kk + kq;
                              const double av = (k < MP && rowi < MP) ? Bb[k * LDT + rowi] : 0.0;
                              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, w[kk >> 2][kk & 3], acc, 0, 0, 0);
                            }
#pragma unroll
                          for (int q = 0; q < 4; ++q)
                            {
                              const int rr = 16 * it + kq + 4 * q;
                              if (rr >= m)
                                continue;
                              if (isw && col < m)
                                Tc[rr * LDT + col] -= acc[q];
                              else if (!isw && col < nc)
                                xrow(la + NV, rr)[col] -= acc[q];
                            }
                        }
                  }
                __syncthreads();
              }
            const int keep = sPrev;
            sPrev = sCur;
            sCur  = sB;
            sB    = keep;
          }
      }
    }

    stamp(6);
    {
    ND_LOCALS
    if (ND_PHASES & 32)
    {
      constexpr int TI = (MP + 15) / 16;
      const int     tjn = (ncg + 15) >> 4;
      stamp(12);
      // backward: X_a = Z_a - W_a X_{a+1}
      for (int a = nH - 1; a >= 0; --a)
        {
          const int     la = a * NV + NV - 1;
          const double *wp = wm + (size_t)a * MP * MP, *zp = zs + (size_t)a * MP * ncg;
          auto          fa = [&](int row, int k) -> double { return row < MP ? wp[row * MP + k] : 0.0; };
          auto fb = [&](int k, int col) -> double { return (k < m && col < nc) ? xrow(la + NV, k)[col] : 0.0; };
          for (int t = wave; t < TI * tjn; t += NW)
            {
              const int ti = t / tjn, tj = t - ti * tjn;
              double4_t acc = {0.0, 0.0, 0.0, 0.0};
              if (a < nH - 1)
                acc = nd_mfma_tile<KTM>(fa, fb, ti, tj, lane);
#pragma unroll
              for (int q = 0; q < 4; ++q)
                {
                  const int row = 16 * ti + (lane >> 4) + 4 * q, col = 16 * tj + (lane & 15);
                  if (row < m && col < nc)
                    xrow(la, row)[col] = zp[row * ncg + col] - acc[q];
                }
            }
          __syncthreads();
        }
    }

    }
    stamp(7);
    {
    ND_LOCALS
    // ------------------------------ back substitution: edges ------------------------------------
    if ((ND_PHASES & 64) && NE > 0)
      {
        const int tjn = (ncg + 15) >> 4;
        for (int t = wave; t < Ca * 2 * tjn; t += NW)
          {
            const int     a = t / (2 * tjn), r = t - a * 2 * tjn, ti = r / tjn, tj = r - ti * tjn;
            const double *ye = yh + (size_t)a * NEP * MP2, *ge = yg + (size_t)a * NEP * ncg;
            auto          fa = [&](int row, int k) -> double { return ye[row * MP2 + k]; };
            auto          fb = [&](int k, int col) -> double {
              const bool top = k >= MP;
              const int  pos = top ? k - MP : k;
              if (pos >= m || col >= nc || (top ? a >= Ca - 1 : a == 0))
                return 0.0;
              return xrow((top ? a : a - 1) * NV + NV - 1, pos)[col];
            };
            const double4_t acc = nd_mfma_tile<MP2 / 4>(fa, fb, ti, tj, lane);
#pragma unroll
            for (int q = 0; q < 4; ++q)
              {
                const int e = 16 * ti + (lane >> 4) + 4 * q, col = 16 * tj + (lane & 15);
                if (e < NE && col < nc)
                  {
                    const int be = e / N1;
                    xrow(a * NV + (e - be * N1), be * NV + NV - 1)[col] = ge[e * ncg + col] - acc[q];
                  }
              }
          }
        __syncthreads();
      }

    }
    stamp(8);
    {
    ND_LOCALS
    // ------------------------------ back substitution: cells ------------------------------------
    // wave = cell, lane = column of P^T: x_c = A_cc^{-1} (f_c - A_cs x_s)
    int pf = 0;
    if ((ND_PHASES & 128) && wave < ncell)
      {
        nd_prefetch_1k(facg + (size_t)wave * lay.cblk, lane_, pf);
        nd_prefetch_1k(facg + (size_t)wave * lay.cblk + 4 * ND_BST, lane_, pf);
      }
    if (ND_PHASES & 128)
    for (int c = wave; c < ncell; c += NW)
      {
        const int     a = c / Cb, b = c - a * Cb, l0 = a * NV, i0 = b * NV;
        const double *fac = facg + (size_t)c * lay.cblk, *fac_next = facg + (size_t)(c + NW < ncell ? c + NW : c) * lay.cblk;
        int           lane = lane_;
        asm volatile("" : "+v"(lane));
        const int pl = lane < PST ? lane : PST - 1;
        __builtin_amdgcn_wave_barrier();
        if (lane < RING)
          {
            int l, i;
            ring_node(a, b, lane, l, i);
            const int  side = ring_side<NV>(lane);
            const bool lv = live(l, i);
#pragma unroll
            for (int s = 0; s < 3; ++s)
              {
                const int dl = side == 0 ? 1 : (side == 1 ? -1 : s - 1);
                const int o  = side < 2 ? s - 1 : (side == 2 ? 1 : -1);
                const int ln = l + dl - l0, in = i + o - i0;
                double    v  = 0.0;
                if (lv && ln >= 0 && ln < N1 && in >= 0 && in < N1)
                  v = cpl(l, i, dl, o);
                kt[lane * 4 + s] = v;
              }
          }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int  col = lane < nc ? lane : nc - 1;
        const bool own = col == own_col(a, b);
        // right-hand sides of the ring rows: f_c - A_cs x_s (x_s straight from the rows of X; the
        // registers are free at this point).  Rows < HR wait in their LDS row, the others in registers.
        const double f0 = own ? 4.0 * A.scale : 0.0;
        double       bh[CL::NHR];
        {
          double br[NR];
          nd_static_for<0, NR>([&](auto R) { br[decltype(R)::value] = f0; });
          nd_static_for<0, 4>([&](auto SD) {
            constexpr int side = decltype(SD)::value;
            constexpr int j0 = side == 0 ? 0 : (side == 1 ? NV + 1 : (side == 2 ? 2 * NV + 2 : 3 * NV + 1));
            constexpr int j1 = side == 0 ? NV + 1 : (side == 1 ? 2 * NV + 2 : (side == 2 ? 3 * NV + 1 : 4 * NV));
            double        xs[NV + 1];
            nd_static_for<j0, j1>([&](auto J) {
              constexpr int j = decltype(J)::value, off = ring_off<NV>(j);
              const int     l = side == 0 ? l0 - 1 : (side == 1 ? l0 + NV - 1 : l0 + off);
              const int     i = side < 2 ? i0 - 1 + off : (side == 2 ? i0 - 1 : i0 + NV - 1);
              // (unconditional load from a clamped row, then a select: no branch per ring node)
              const int    lc = l < 0 ? 0 : (l >= L ? L - 1 : l), ic = i < 0 ? 0 : (i >= m ? m - 1 : i);
              const double raw = xrow(lc, ic)[col];
              xs[j - j0] = live(l, i) ? raw : 0.0;
            });
            nd_static_for<j0, j1>([&](auto J) {
              constexpr int j = decltype(J)::value;
              nd_static_for<0, 3>([&](auto S) {
                constexpr int nb = ring_nb<NV>(j, decltype(S)::value);
                if constexpr (nb >= 0)
                  br[nb] = fma(-kt[j * 4 + decltype(S)::value], xs[j - j0], br[nb]);
              });
            });
            __builtin_amdgcn_sched_barrier(0);
          });
          nd_static_for<0, NR>([&](auto R) {
            constexpr int r = decltype(R)::value;
            if constexpr (nd_ring_row<NV>(r))
              {
                if constexpr (r < CL::HR)
                  park[r * PST + pl] = br[r];
                else
                  bh[CL::hi_slot(r)] = br[r];
              }
          });
        }
        auto b_of = [&](auto R) -> double {
          constexpr int r = decltype(R)::value;
          if constexpr (!nd_ring_row<NV>(r))
            return f0;
          else if constexpr (r < CL::HR)
            return park[r * PST + pl];
          else
            return bh[CL::hi_slot(r)];
        };
        // rows are emitted last first: a running pointer into X instead of (NV-1)^2 hoisted addresses
        // (idle lanes store to a dead scratch word: no branch per row, the factor rows stay in SGPRs)
        double      *px = lane < nc ? xrow(l0 + N1 - 1, i0 + N1 - 1) + lane : pm;
        const size_t back_line = lane < nc ? xline - (size_t)(N1 - 1) * ncg : 0, back_one = lane < nc ? ncg : 0;
        auto         emit = [&](auto R, double y) {
          *px = y;
          px -= (decltype(R)::value % N1 == 0) ? back_line : back_one;
          asm volatile("" : "+v"(px));
        };
        nd_band_solve<NV>(fac, fac_next, park, PST, pl, lane, pf, b_of, emit);
      }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf) : : "memory");
    if (bad && !SLOD_DG(A, -1))
      atomicOr(A.status, 1);
    stamp(9);

    // (the selection stage runs as its own launch, k_select: it is written for 256-thread workgroups)
    }
  }

} // namespace


template <int NV, int T>
static hipError_t launch_nd(const SlodKernelArgs &a, int n_patches, size_t lds, hipStream_t st)
{
  const void *fn = reinterpret_cast<const void *>(k_solve_nd<NV, T>);
  hipError_t  e  = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess)
    return e;
  if (a.debug)
    {
      int nb = 0;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * ND_WAVES, lds);
      fprintf(stderr, "[slod] k_solve_nd<%d,%d>: %d patches, lds %zu B, occupancy %d blocks/CU\n", NV, T, n_patches, lds, nb);
    }
  hipLaunchKernelGGL((k_solve_nd<NV, T>), dim3(n_patches), dim3(64 * ND_WAVES), lds, st, a);
  return hipGetLastError();
}


#endif
