// Host-side internals of libslod_hip.so shared by slod_api.cpp and slod_global.hip (not part of
// the public ABI).
#ifndef SLOD_HOST_H
#define SLOD_HOST_H
#pragma GCC visibility push(default)
#include "../../include/slod.h"
#pragma GCC visibility pop
#include "slod_device.h"

#include <string>
#include <vector>

struct slod_handle
{
  slod_config         cfg;
  int                 N  = 0; // coarse cells per side
  int                 NE = 0; // fine elements per side
  int                 NP = 0; // patches per problem
  int                 first_full = -1;
  double             *d_coef[2]  = {nullptr, nullptr};
  std::vector<char>   coef_set;  // [problem*2 + field]
  hipStream_t         stream = nullptr;
  bool                device_ready = false; // stream and coefficient storage exist
  mutable std::string error;
};

// error text of the last failed slod_create on this thread (slod_api.cpp)
std::string &slod_create_error();
inline int   slod_fail(const slod_handle *h, int code, const std::string &msg)
{
  if (h)
    h->error = msg;
  else
    slod_create_error() = msg;
  return code;
}
inline int slod_hip_fail(const slod_handle *h, hipError_t e, const char *what)
{
  return slod_fail(h, SLOD_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}
// stream + coefficient storage; called by every entry point that touches the device (slod_api.cpp)
int slod_ensure_device(slod_handle *h);
// Device copy of the index calculus a kernel needs: everything derives from these scalars.
struct SlodGrid
{
  int32_t N, n_sub, oversampling, spacedim, morton_bits; // morton_bits < 0: row-major patch ids
  int32_t lod_stabilization;
};
inline SlodGrid slod_grid_of(const slod_handle *h)
{
  SlodGrid g;
  g.N                 = h->N;
  g.n_sub             = h->cfg.n_subdivisions;
  g.oversampling      = h->cfg.oversampling;
  g.spacedim          = h->cfg.spacedim;
  g.morton_bits       = h->cfg.n_cells_per_side > 0 ? -1 : h->cfg.n_global_refinements;
  g.lod_stabilization = h->cfg.lod_stabilization;
  return g;
}
// plan construction on the device (slod_global.hip: k_make_desc, k_balance_order)
struct SlodPlanSummary
{
  int32_t            m_max, L_max, nc_max, nb_max, nn_max, error;
  unsigned long long out_size;
};
struct SlodPlanBuild
{
  const uint32_t  *gids;
  const uint64_t  *offsets; // may be null: uniform stride
  size_t           n, stride;
  int32_t          NP, n_problems, reuse_full, first_full;
  SlodPatchDesc   *desc;
  double          *cost;
  SlodPlanSummary *acc;
  char            *prob_used;
};
// descriptor -> public patch layout (slod_plan_patch_layout, slod_device_patch_layout)
inline void slod_desc_to_info(const slod_handle *h, const SlodPatchDesc &d, slod_patch_info *info)
{
  const int n = h->cfg.n_subdivisions, s = h->cfg.spacedim;
  *info       = slod_patch_info();
  info->mx    = d.mx;
  info->my    = d.my;
  info->nx    = d.nx;
  info->ny    = d.ny;
  info->x0    = d.ox / n; // (under the reuse quirk Q1 the coefficient origin is the first full patch's)
  info->y0    = d.oy / n;
  info->cx    = info->x0 + d.ccx;
  info->cy    = info->y0 + d.ccy;
  for (int k = 0; k < 4; ++k)
    info->side_domain[k] = (d.flags >> k) & 1;
  info->n_fine     = s * (d.nx + 1) * (d.ny + 1);
  info->n_internal = s * (d.nx - 1) * (d.ny - 1);
  info->n_boundary = d.n_b;
  info->n_coarse   = d.n_c;
  info->is_lod     = (d.flags & SLOD_F_LOD) ? 1 : 0;
}
hipError_t slod_build_descriptors(const slod_handle *h, const uint32_t *gids, size_t n, const uint64_t *offsets, size_t stride,
                                  int n_cu, bool balance, SlodPatchDesc *d_desc, SlodPatchDesc *d_desc_bal, SlodPlanSummary *sum,
                                  std::vector<char> *prob_used);
#endif
