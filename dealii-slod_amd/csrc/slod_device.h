// Shared host/device declarations of libslod_hip (not part of the public ABI).
#ifndef SLOD_DEVICE_H
#define SLOD_DEVICE_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// One oversampling patch as the kernels see it.  Produced on the host from the index
// calculus of create_patches()/create_mesh_for_patch() (reference source/LOD.cc:122-244,
// 770-858); nothing in here depends on the coefficient.
struct SlodPatchDesc
{
  int32_t  ox, oy;   // origin of the coefficient tile in the global fine-element grid
  int32_t  nx, ny;   // fine elements per side
  int32_t  mx, my;   // coarse cells per side
  int32_t  ccx, ccy; // centre cell in patch-local cell coordinates (column 0 of P^T)
  int32_t  flags;    // bit0..3: side (L,R,B,T) is on the domain boundary (id 0);
                     // bit4: LOD branch (LOD.cc:563-564); bit5: lines run along y
  int32_t  m, L;     // dofs per grid line, number of interior lines (m <= L)
  int32_t  n_c, n_b; // coarse dofs, id-99 boundary dofs
  int32_t  prob;     // coefficient realisation
  uint32_t plan_index; // position of the patch in the caller's list (diagnostics slot)
  uint64_t out_off;  // offset (doubles) of this patch in basis / premult
};

enum : int32_t
{
  SLOD_F_LOD        = 1 << 4,
  SLOD_F_TRANSPOSED = 1 << 5
};

// Phase-skip / timeline masks of the timing experiments (tools/): compiled in only with
// -DSLOD_ENABLE_DIAG (lib/libslod_hip_diag.so); the release library has no such branches.
#ifdef SLOD_ENABLE_DIAG
#define SLOD_DG(A, mask) ((A).diag & (mask))
#else
#define SLOD_DG(A, mask) 0
#endif

// Decisions of the SLOD selection stage for one (patch, component): what include/slod.h
// exposes as slod_patch_diag (same layout).
struct SlodPatchDiag
{
  int32_t path;      // 0 LOD branch, 1 SLOD without decisions (QR, proven), 2 SLOD with SVD replay
  int32_t n_cut;     // singular values of G under the 1e-15 cutoff (LOD.cc:667)
  int32_t n_dropped; // triplets removed by the 0.5-loop (LOD.cc:703-725)
  int32_t sweeps;    // Jacobi sweeps of the SVD replay
  double  dinf;      // final ||d||_inf
  double  sigma_max, sigma_min; // extreme singular values of G (path 2), else 0
};

struct SlodKernelArgs
{
  const SlodPatchDesc *desc; // [n_patches of this launch]
  const double        *coef0;
  const double        *coef1;
  size_t               coef_stride; // doubles per problem
  int32_t              NE;          // fine elements per side of the global grid
  int32_t              n_sub;
  int32_t              quirk;       // projection quirk Q2
  int32_t              diag;        // timing experiments (SLOD_ENABLE_DIAG builds only): phase skip mask
  int32_t              debug;       // print the launch configuration (SLOD_DEBUG)
  double               scale;       // h^2/4
  double               invH2;       // 1/H^dim
  // workspace (one slot per patch of the launch)
  double *st;
  size_t  st_stride;
  int32_t nn_max; // node stride inside a stencil slot
  double *vinv;
  size_t  v_stride;
  int32_t m_max;
  int32_t L_max;    // most interior lines of a patch of the plan
  int32_t nv;       // k_solve_nd: cell size of the dissection (fine elements per side)
  double *xs;
  double *zs;       // k_solve_tw: Z of the forward sweep (same strides as xs); other kernels keep Z in xs
  size_t  x_stride;
  int32_t nc_max;
  double *ms;       // per patch M = P^T A^-1 P / H^2 accumulated by k_solve_ws (nc_max^2 doubles)
  int32_t m_fused;  // 1: k_select reads M from ms instead of recomputing it from X
  int32_t nb_buf;   // rows of the selection stage's boundary-trace buffer
  int32_t nf_max;   // largest n_fine of the plan
  int32_t fuse_select; // set by slod_launch_solve: the solve kernel also ran the selection stage
  int32_t fuse_assemble; // the solve kernel assembles the stencil of its own patch first
  // outputs
  double  *basis;
  double  *premult;
  int32_t *status;
  SlodPatchDiag *pdiag; // [patches of the PLAN][S], indexed by SlodPatchDesc::plan_index
};

enum SlodSolverKind : int32_t
{
  SLOD_K_MF   = 1, // slod_solve_mf.hip
  SLOD_K_TW   = 2, // slod_solve_tw.hip
  SLOD_K_WS   = 3, // slod_solve_ws.hip
  SLOD_K_COOP = 4, // slod_solve_coop.hip
  SLOD_K_ND   = 5  // slod_solve_nd.hip
};

// tuning knobs (environment, read once per plan) and the resulting kernel choice: slod_dispatch.cpp
struct SlodTuning
{
  int solver = 0;        // 0 = automatic, else a SlodSolverKind
  int fuse_select = 1, fuse_assemble = 1, fuse_m = 0;
  int twisted = -1;      // coop kernel only: -1 = automatic
  int balance = 1;       // launch order balanced over the CUs (SLOD_BALANCE=0: the caller's order)
  int debug = 0;
};
struct SlodSolveChoice
{
  int    kind = 0;
  size_t lds  = 0;
  int    fuse_select = 0, fuse_assemble = 0, m_fused = 0, twisted = 0, debug = 0;
  int    v_line_pad = 0; // rows = columns of a V line as the kernel sees it
  size_t v_line_elems = 0; // doubles per stored V line (k_solve_tw: the 36 upper lane tiles only)
  int    nv = 0;            // k_solve_nd: cell size
  size_t v_patch_elems = 0; // k_solve_nd: doubles of scratch per patch (replaces L_max * v_line_elems)
};
SlodTuning slod_read_tuning();
bool       slod_choose_solver(int S, int n_sub, int m_max, int L_max, int nc_max, int nb_buf, int nf_max, size_t n_patches,
                              const SlodTuning &t, SlodSolveChoice *out);

// launchers (slod_assemble.hip, slod_solve_{mf,tw,ws,coop}.hip, slod_select.hip)
hipError_t slod_launch_assemble(int S, const SlodKernelArgs &a, int n_patches, hipStream_t st);
hipError_t slod_launch_solve(int S, const SlodSolveChoice &c, SlodKernelArgs &a, int n_patches, hipStream_t st);
hipError_t slod_launch_solve_mf(int S, const SlodKernelArgs &a, int n_patches, size_t lds, hipStream_t st);
hipError_t slod_launch_solve_tw(int S, const SlodKernelArgs &a, int n_patches, size_t lds, hipStream_t st);
hipError_t slod_launch_solve_ws(int S, const SlodKernelArgs &a, int n_patches, size_t lds, hipStream_t st);
hipError_t slod_launch_solve_coop(int S, int twisted, const SlodKernelArgs &a, int n_patches, hipStream_t st);
hipError_t slod_launch_solve_nd(const SlodKernelArgs &a, int n_patches, size_t lds, hipStream_t st);
int        slod_solve_nd_cell(int S, int n_sub, int m_max, int L_max); // cell size, 0 = not applicable
size_t     slod_solve_nd_scratch(int nv, int m_max, int L_max, int nc_max);
size_t     slod_solve_nd_lds_bytes(int nv, int m_max, int nc_max);
hipError_t slod_launch_select(int S, const SlodKernelArgs &a, int n_patches, int nb_max,
                              int nf_max, hipStream_t st);
size_t     slod_solve_lds_bytes(int S, int m_max, int nc_max, int twisted);
size_t     slod_solve_ws_lds_bytes(int S, int m_max, int nc_max);
size_t     slod_solve_tw_lds_bytes(int S, int m_max, int nc_max);
size_t     slod_solve_mf_lds_bytes(int S, int m_max, int nc_max);
int        slod_solve_ws_tile(int m_max);
int        slod_solve_mf_tiles(int S, int m_max); // 16 x 16 tiles per line side, 0 = does not fit
size_t     slod_select_lds_bytes(int S, int nb_max, int nc_max, int nf_max);

#endif
