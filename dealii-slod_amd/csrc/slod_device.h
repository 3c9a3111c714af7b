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
  uint64_t out_off;  // offset (doubles) of this patch in basis / premult
};

enum : int32_t
{
  SLOD_F_LOD        = 1 << 4,
  SLOD_F_TRANSPOSED = 1 << 5
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
  int32_t              diag;        // timing diagnostics only (env SLOD_DIAG): phase skip mask
  double               scale;       // h^2/4
  double               invH2;       // 1/H^dim
  // workspace (one slot per patch of the launch)
  double *st;
  size_t  st_stride;
  int32_t nn_max; // node stride inside a stencil slot
  double *vinv;
  size_t  v_stride;
  int32_t m_max;
  double *xs;
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
};

// launchers (slod_assemble.hip, slod_solve_{tw,ws,coop}.hip, slod_select.hip)
hipError_t slod_launch_assemble(int S, const SlodKernelArgs &a, int n_patches, hipStream_t st);
hipError_t slod_launch_solve(int S, SlodKernelArgs &a, int n_patches, hipStream_t st); // slod_dispatch.cpp
bool       slod_solve_fuses_assemble(int S, const SlodKernelArgs &a);               // slod_dispatch.cpp
hipError_t slod_launch_solve_tw(int S, const SlodKernelArgs &a, int n_patches, size_t lds, hipStream_t st);
hipError_t slod_launch_solve_ws(int S, const SlodKernelArgs &a, int n_patches, size_t lds, hipStream_t st);
hipError_t slod_launch_solve_coop(int S, int twisted, const SlodKernelArgs &a, int n_patches, hipStream_t st);
hipError_t slod_launch_select(int S, const SlodKernelArgs &a, int n_patches, int nb_max,
                              int nf_max, hipStream_t st);
size_t     slod_solve_lds_bytes(int S, int m_max, int nc_max, int twisted);
size_t     slod_solve_ws_lds_bytes(int S, int m_max, int nc_max);
size_t     slod_solve_tw_lds_bytes(int S, int m_max, int nc_max);
int        slod_solve_ws_tile(int m_max);
size_t     slod_select_lds_bytes(int S, int nb_max, int nc_max, int nf_max);

#endif
