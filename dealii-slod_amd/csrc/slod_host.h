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
#endif
