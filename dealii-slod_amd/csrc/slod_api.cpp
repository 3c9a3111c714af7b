// Host side of libslod_hip.so: handle / plan management, patch index calculus, launches.
// Implements include/slod.h.  No CPU fallback exists: every compute entry point needs a
// HIP device and fails with SLOD_ERR_DEVICE otherwise.
// only the C-ABI of include/slod.h is exported (the library is built with -fvisibility=hidden)
#include "slod_host.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace
{
  struct PatchGeom
  {
    int cx, cy, x0, y0, mx, my;
    int side_domain[4];
  };
} // namespace

std::string &slod_create_error()
{
  static thread_local std::string e;
  return e;
}

struct slod_plan
{
  slod_handle               *h = nullptr;
  size_t                     n = 0;
  std::vector<SlodPatchDesc> desc;
  SlodPatchDesc             *d_desc = nullptr;     // the caller's order (pieces of slod_plan_execute_allgather)
  SlodPatchDesc             *d_desc_bal = nullptr; // launch order balanced over the CUs (slod_plan_execute)
  int                        m_max = 0, L_max = 0, nc_max = 0, nb_max = 0, nn_max = 0, nf_max = 0;
  int                        nb_buf = 0; // rows of k_select's boundary-trace buffer
  size_t                     stride = 0, out_size = 0;
  size_t                     chunk = 0;
  double                    *ws_st = nullptr, *ws_v = nullptr, *ws_x = nullptr, *ws_m = nullptr;
  double                    *ws_z = nullptr; // k_solve_tw: Z of the forward sweep (active column prefix per line)
  double                    *ws_x_alloc = nullptr; // ws_x sits `guard` doubles inside this allocation
  size_t                     guard = 0;
  size_t                     st_stride = 0, v_stride = 0, x_stride = 0;
  int32_t                   *d_status = nullptr;
  SlodPatchDiag             *d_pdiag  = nullptr; // [n][spacedim] decisions of the selection stage
  SlodSolveChoice            choice;             // kernel chosen at plan creation (slod_dispatch.cpp)
  std::vector<hipEvent_t>    ev; // [depth][n_chunks][4]
  size_t                     n_chunks = 0;
  int                        depth = 1; // event slots (slod_plan_profile)
  size_t                     n_exec = 0;
  bool                       ran = false;
  // slod_plan_set_overlap(2): a second workspace and two internal streams, consecutive executes alternate
  struct AltWs
  {
    double  *ws_st = nullptr, *ws_v = nullptr, *ws_x_alloc = nullptr, *ws_x = nullptr, *ws_z = nullptr, *ws_m = nullptr;
    int32_t *d_status = nullptr;
  } alt;
  int         overlap = 1;
  hipStream_t istream[2] = {nullptr, nullptr};
  hipEvent_t  fork_ev = nullptr, done_ev[2] = {nullptr, nullptr};
  bool        inflight[2] = {false, false};
  bool                       uniform_stride = true; // NULL offsets: patch k at k * stride
  std::vector<char>          prob_used;             // [n_problems] coefficient realisations the plan reads
};

namespace
{
  int fail(const slod_handle *h, int code, const std::string &msg) { return slod_fail(h, code, msg); }
  int hip_fail(const slod_handle *h, hipError_t e, const char *what) { return slod_hip_fail(h, e, what); }

  // Morton order of hyper_cube + refine_global (x in the even bits), or row-major for a
  // non-2^k grid.  Reference: patches are stored in active-cell order (LOD.cc:184-192).
  void patch_centre(const slod_handle *h, uint32_t pid, int &cx, int &cy)
  {
    if (h->cfg.n_cells_per_side > 0)
      {
        cx = (int)(pid % (uint32_t)h->N);
        cy = (int)(pid / (uint32_t)h->N);
        return;
      }
    cx = cy = 0;
    for (int b = 0; b < h->cfg.n_global_refinements; ++b)
      {
        cx |= (int)((pid >> (2 * b)) & 1u) << b;
        cy |= (int)((pid >> (2 * b + 1)) & 1u) << b;
      }
  }

  // LOD.cc:140-181 (extent) and LOD.cc:830-843 (boundary ids of the four sides)
  PatchGeom patch_geom(const slod_handle *h, uint32_t pid)
  {
    PatchGeom g;
    const int l = h->cfg.oversampling, N = h->N;
    patch_centre(h, pid, g.cx, g.cy);
    g.x0         = std::max(g.cx - l, 0);
    g.y0         = std::max(g.cy - l, 0);
    const int x1 = std::min(g.cx + l, N - 1), y1 = std::min(g.cy + l, N - 1);
    g.mx             = x1 - g.x0 + 1;
    g.my             = y1 - g.y0 + 1;
    g.side_domain[0] = (g.x0 == 0);
    g.side_domain[1] = (x1 == N - 1);
    g.side_domain[2] = (g.y0 == 0);
    g.side_domain[3] = (y1 == N - 1);
    return g;
  }

  void fill_info(const slod_handle *h, const PatchGeom &g, slod_patch_info *info)
  {
    const int n = h->cfg.n_subdivisions, s = h->cfg.spacedim;
    info->cx = g.cx;
    info->cy = g.cy;
    info->x0 = g.x0;
    info->y0 = g.y0;
    info->mx = g.mx;
    info->my = g.my;
    info->nx = n * g.mx;
    info->ny = n * g.my;
    for (int i = 0; i < 4; ++i)
      info->side_domain[i] = g.side_domain[i];
    info->n_fine     = s * (info->nx + 1) * (info->ny + 1);
    info->n_internal = s * (info->nx - 1) * (info->ny - 1);
    info->n_coarse   = s * g.mx * g.my;
    // id-99 nodes, corners shared with an id-0 side included (LODtools.h:367-369)
    int nb = 0;
    if (!g.side_domain[2])
      nb += info->nx + 1;
    else
      nb += !g.side_domain[0] + !g.side_domain[1];
    if (!g.side_domain[3])
      nb += info->nx + 1;
    else
      nb += !g.side_domain[0] + !g.side_domain[1];
    nb += (info->ny - 1) * (!g.side_domain[0] + !g.side_domain[1]);
    info->n_boundary = s * nb;
    info->is_lod     = (!h->cfg.lod_stabilization) || h->cfg.oversampling == 0 ||
                   (g.mx * g.my == h->N * h->N);
  }

  bool is_full(const slod_handle *h, const PatchGeom &g)
  {
    const int f = 2 * h->cfg.oversampling + 1;
    return g.mx == f && g.my == f;
  }

  // stream + coefficient storage; called by every entry point that touches the device
  int ensure_device(slod_handle *h) { return slod_ensure_device(h); }
} // namespace

int slod_ensure_device(slod_handle *h)
{
  {
    if (h->device_ready)
      return SLOD_OK;
    hipError_t e = hipSetDevice(h->cfg.device);
    if (e != hipSuccess)
      return hip_fail(h, e, "hipSetDevice (no usable HIP device; this library has no CPU fallback)");
    hipStream_t  stream = nullptr;
    double      *coef[2] = {nullptr, nullptr};
    e = hipStreamCreate(&stream);
    if (e != hipSuccess)
      return hip_fail(h, e, "hipStreamCreate");
    const size_t bytes = (size_t)h->cfg.n_problems * h->NE * h->NE * 4 * sizeof(double);
    for (int f = 0; f < h->cfg.spacedim && e == hipSuccess; ++f)
      e = hipMalloc((void **)&coef[f], bytes);
    if (e != hipSuccess)
      {
        // nothing half-initialised is kept: the next call starts from scratch
        for (int f = 0; f < 2; ++f)
          if (coef[f])
            (void)hipFree(coef[f]);
        (void)hipStreamDestroy(stream);
        return hip_fail(h, e, "hipMalloc(coefficient field)");
      }
    h->stream       = stream;
    h->d_coef[0]    = coef[0];
    h->d_coef[1]    = coef[1];
    h->device_ready = true;
    return SLOD_OK;
  }
}

namespace
{
  SlodKernelArgs make_args(const slod_plan *p, size_t first, double *d_basis, double *d_premult, bool balanced, int slot = 0)
  {
    const slod_handle *h = p->h;
    SlodKernelArgs     a;
    std::memset(&a, 0, sizeof(a));
    a.desc        = (balanced && p->d_desc_bal ? p->d_desc_bal : p->d_desc) + first;
    a.coef0       = h->d_coef[0];
    a.coef1       = h->d_coef[1];
    a.coef_stride = (size_t)h->NE * h->NE * 4;
    a.NE          = h->NE;
    a.n_sub       = h->cfg.n_subdivisions;
    a.quirk       = h->cfg.projection_quirk;
#ifdef SLOD_ENABLE_DIAG
    if (const char *dg = std::getenv("SLOD_DIAG")) // lib/libslod_hip_diag.so only (tools/)
      a.diag = std::atoi(dg); // timing experiments: results are wrong when non-zero
#endif
    const double H = 1.0 / (double)h->N, hh = H / (double)h->cfg.n_subdivisions;
    a.scale     = hh * hh / 4.0; // LOD.cc:341
    a.invH2     = 1.0 / (H * H); // LOD.cc:551
    a.st        = p->ws_st;
    a.st_stride = p->st_stride;
    a.nn_max    = p->nn_max;
    a.vinv      = p->ws_v;
    a.v_stride  = p->v_stride;
    a.m_max     = p->m_max;
    a.L_max     = p->L_max;
    a.xs        = p->ws_x;
    a.zs        = p->ws_z ? p->ws_z : p->ws_x;
    a.x_stride  = p->x_stride;
    a.nc_max    = p->nc_max;
    a.ms        = p->ws_m;
    a.m_fused   = 0; // set by slod_launch_solve when the wave-specialised kernel runs
    a.nb_buf    = p->nb_buf;
    a.nf_max    = p->nf_max;
    a.fuse_select = 0; // set by slod_launch_solve
    a.fuse_assemble = 0;
    a.basis     = d_basis;
    a.premult   = d_premult;
    a.status    = p->d_status;
    a.pdiag     = p->d_pdiag;
    if (slot == 1)
      {
        a.st     = p->alt.ws_st;
        a.vinv   = p->alt.ws_v;
        a.xs     = p->alt.ws_x;
        a.zs     = p->alt.ws_z ? p->alt.ws_z : p->alt.ws_x;
        a.ms     = p->alt.ws_m;
        a.status = p->alt.d_status;
      }
    return a;
  }
} // namespace

namespace
{
  // the launches of the patches [first, first + cnt) of a plan (cnt <= chunk: one workspace slot per
  // workgroup); ev (optional): four events around the three stages
  hipError_t launch_range(slod_plan *p, size_t first, int cnt, double *d_basis, double *d_premult, hipStream_t st,
                          hipEvent_t *ev, bool balanced, int slot = 0)
  {
    const int      s = p->h->cfg.spacedim;
    SlodKernelArgs a = make_args(p, first, d_basis, d_premult, balanced, slot);
    hipError_t     e = ev ? hipEventRecord(ev[0], st) : hipSuccess;
    if (e == hipSuccess && !p->choice.fuse_assemble)
      e = slod_launch_assemble(s, a, cnt, st);
    if (e == hipSuccess && ev)
      e = hipEventRecord(ev[1], st);
    if (e == hipSuccess)
      e = slod_launch_solve(s, p->choice, a, cnt, st); // sets a.m_fused, a.fuse_select, a.fuse_assemble
    if (e == hipSuccess && ev)
      e = hipEventRecord(ev[2], st);
    if (e == hipSuccess && !a.fuse_select)
      e = slod_launch_select(s, a, cnt, p->nb_buf, p->nf_max, st);
    if (e == hipSuccess && ev)
      e = hipEventRecord(ev[3], st);
    return e;
  }
} // namespace

#pragma GCC visibility push(default)
extern "C" {

int slod_abi_version(void) { return SLOD_ABI_VERSION; }

const char *slod_last_error(const slod_handle *h)
{
  return h ? h->error.c_str() : slod_create_error().c_str();
}

int slod_create(const slod_config *cfg, slod_handle **out)
{
  if (!cfg || !out)
    return fail(nullptr, SLOD_ERR_ARGUMENT, "slod_create: null argument");
  *out = nullptr;
  if (cfg->dim != 2)
    return fail(nullptr, SLOD_ERR_UNSUPPORTED, "slod_create: only dim == 2 (reference LOD.cc:1470-1471)");
  if (cfg->spacedim != 1 && cfg->spacedim != 2)
    return fail(nullptr, SLOD_ERR_UNSUPPORTED, "slod_create: spacedim must be 1 or 2");
  if (cfg->n_subdivisions < 1 || cfg->oversampling < 0 || cfg->n_global_refinements < 0 ||
      cfg->n_global_refinements > 14 || cfg->n_cells_per_side < 0 || cfg->n_problems < 1)
    return fail(nullptr, SLOD_ERR_ARGUMENT, "slod_create: parameter out of range");
  slod_handle *h = new slod_handle;
  h->cfg         = *cfg;
  h->N           = cfg->n_cells_per_side > 0 ? cfg->n_cells_per_side : (1 << cfg->n_global_refinements);
  h->NE          = h->N * cfg->n_subdivisions;
  h->NP          = h->N * h->N;
  if ((uint64_t)h->NP * (uint64_t)cfg->n_problems > 0xffffffffull)
    {
      delete h;
      return fail(nullptr, SLOD_ERR_ARGUMENT, "slod_create: too many patches for 32-bit ids");
    }
  for (int pid = 0; pid < h->NP; ++pid)
    if (is_full(h, patch_geom(h, (uint32_t)pid)))
      {
        h->first_full = pid;
        break;
      }
  h->coef_set.assign((size_t)cfg->n_problems * 2, 0);
  // device resources are created lazily (ensure_device): the index calculus below works
  // without a GPU, every compute entry point needs one.
  *out = h;
  return SLOD_OK;
}

void slod_destroy(slod_handle *h)
{
  if (!h)
    return;
  for (int f = 0; f < 2; ++f)
    if (h->d_coef[f])
      (void)hipFree(h->d_coef[f]);
  if (h->stream)
    (void)hipStreamDestroy(h->stream);
  delete h;
}

int slod_num_patches(const slod_handle *h) { return h ? h->NP : SLOD_ERR_ARGUMENT; }

int slod_patch_layout(const slod_handle *h, uint32_t patch_id, slod_patch_info *info)
{
  if (!h || !info)
    return SLOD_ERR_ARGUMENT;
  if (patch_id >= (uint32_t)h->NP)
    return fail(h, SLOD_ERR_ARGUMENT, "slod_patch_layout: patch id out of range");
  fill_info(h, patch_geom(h, patch_id), info);
  return SLOD_OK;
}

int slod_patch_cells(const slod_handle *h, uint32_t patch_id, uint32_t *cells, size_t capacity)
{
  if (!h || !cells)
    return SLOD_ERR_ARGUMENT;
  if (patch_id >= (uint32_t)h->NP)
    return fail(h, SLOD_ERR_ARGUMENT, "slod_patch_cells: patch id out of range");
  const PatchGeom g = patch_geom(h, patch_id);
  if (capacity < (size_t)(g.mx * g.my))
    return fail(h, SLOD_ERR_ARGUMENT, "slod_patch_cells: buffer too small");
  const int l = h->cfg.oversampling, N = h->N;
  size_t    c = 0;
  cells[c++]  = (uint32_t)(g.cx + N * g.cy); // LOD.cc:151-154
  for (int lr = -l; lr <= l; ++lr)           // LOD.cc:156-178
    {
      const int x = g.cx + lr;
      if (x < 0 || x >= N)
        continue;
      for (int lc = -l; lc <= l; ++lc)
        {
          const int y = g.cy + lc;
          if (y < 0 || y >= N || (lr == 0 && lc == 0))
            continue;
          cells[c++] = (uint32_t)(x + N * y);
        }
    }
  return (int)c;
}

int slod_patch_dof_permutation(const slod_handle *h, uint32_t patch_id, uint32_t *perm, size_t capacity)
{
  if (!h || !perm)
    return SLOD_ERR_ARGUMENT;
  if (patch_id >= (uint32_t)h->NP)
    return fail(h, SLOD_ERR_ARGUMENT, "slod_patch_dof_permutation: patch id out of range");
  const PatchGeom g = patch_geom(h, patch_id);
  slod_patch_info info;
  fill_info(h, g, &info);
  if (capacity < (size_t)info.n_fine)
    return fail(h, SLOD_ERR_ARGUMENT, "slod_patch_dof_permutation: buffer too small");
  const int             n = h->cfg.n_subdivisions, s = h->cfg.spacedim, npx = info.nx + 1, N = h->N;
  std::vector<uint32_t> cells((size_t)g.mx * g.my);
  slod_patch_cells(h, patch_id, cells.data(), cells.size());
  std::vector<int32_t> number((size_t)info.n_fine, -1);
  uint32_t             next  = 0;
  auto                 touch = [&](int ix, int iy, int c) {
    const int lex = s * (ix + iy * npx) + c;
    if (number[lex] < 0)
      {
        number[lex]  = (int32_t)next;
        perm[next++] = (uint32_t)lex;
      }
  };
  // DoFHandler::distribute_dofs on the patch sub-triangulation: cells in creation order
  // (= patch->cells, LOD.cc:803-819), per cell vertex, line, quad dofs, first touch;
  // FESystem keeps the component copies of one geometric object consecutive.
  for (uint32_t cell : cells)
    {
      const int bx = ((int)(cell % (uint32_t)N) - g.x0) * n, by = ((int)(cell / (uint32_t)N) - g.y0) * n;
      for (int v = 0; v < 4; ++v)
        for (int c = 0; c < s; ++c)
          touch(bx + (v & 1) * n, by + (v >> 1) * n, c);
      for (int line = 0; line < 4; ++line)
        for (int c = 0; c < s; ++c)
          for (int t = 1; t < n; ++t)
            {
              if (line == 0)
                touch(bx, by + t, c);
              else if (line == 1)
                touch(bx + n, by + t, c);
              else if (line == 2)
                touch(bx + t, by, c);
              else
                touch(bx + t, by + n, c);
            }
      for (int c = 0; c < s; ++c)
        for (int jy = 1; jy < n; ++jy)
          for (int jx = 1; jx < n; ++jx)
            touch(bx + jx, by + jy, c);
    }
  return (int)next;
}

int slod_partition(uint64_t n_total, uint32_t n_ranks, uint32_t rank, uint64_t *begin, uint64_t *end)
{
  if (!begin || !end || n_ranks == 0 || rank >= n_ranks)
    return SLOD_ERR_ARGUMENT;
  // Utilities::MPI::create_evenly_distributed_partitioning: the first (n % p) ranks own
  // one element more.
  const uint64_t q = n_total / n_ranks, r = n_total % n_ranks;
  *begin = (uint64_t)rank * q + std::min<uint64_t>(rank, r);
  *end   = *begin + q + (rank < r ? 1 : 0);
  return SLOD_OK;
}

int slod_set_coefficient(slod_handle *h, uint32_t problem, int field, const double *data, int layout,
                         size_t count, int on_device)
{
  if (!h || !data)
    return SLOD_ERR_ARGUMENT;
  if (problem >= (uint32_t)h->cfg.n_problems || field < 0 || field >= h->cfg.spacedim)
    return fail(h, SLOD_ERR_ARGUMENT, "slod_set_coefficient: problem/field out of range");
  const size_t ne = (size_t)h->NE * h->NE;
  if ((layout == 0 && count != ne) || (layout == 1 && count != 4 * ne) || (layout != 0 && layout != 1))
    return fail(h, SLOD_ERR_ARGUMENT, "slod_set_coefficient: wrong element count for layout");
  if (const int rc = ensure_device(h))
    return rc;
  (void)hipSetDevice(h->cfg.device);
  double    *dst = h->d_coef[field] + (size_t)problem * ne * 4;
  hipError_t e;
  if (layout == 1)
    e = hipMemcpyAsync(dst, data, 4 * ne * sizeof(double),
                       on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream);
  else if (on_device)
    {
      e = hipSuccess;
      for (int q = 0; q < 4 && e == hipSuccess; ++q)
        e = hipMemcpy2DAsync(dst + q, 4 * sizeof(double), data, sizeof(double), sizeof(double), ne,
                             hipMemcpyDeviceToDevice, h->stream);
    }
  else
    {
      std::vector<double> tmp(4 * ne);
      for (size_t i = 0; i < ne; ++i)
        tmp[4 * i] = tmp[4 * i + 1] = tmp[4 * i + 2] = tmp[4 * i + 3] = data[i];
      e = hipMemcpyAsync(dst, tmp.data(), 4 * ne * sizeof(double), hipMemcpyHostToDevice, h->stream);
      if (e == hipSuccess)
        e = hipStreamSynchronize(h->stream);
    }
  if (e == hipSuccess)
    e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess)
    return hip_fail(h, e, "slod_set_coefficient");
  h->coef_set[(size_t)problem * 2 + field] = 1;
  return SLOD_OK;
}

int slod_plan_create(slod_handle *h, const uint32_t *gids, size_t n, const uint64_t *offsets, slod_plan **out)
{
  if (!h || !out || (!gids && n))
    return SLOD_ERR_ARGUMENT;
  *out = nullptr;
  if (const int rc = ensure_device(h))
    return rc;
  (void)hipSetDevice(h->cfg.device);
  const uint64_t total = (uint64_t)h->NP * (uint64_t)h->cfg.n_problems;
  slod_plan     *p     = new slod_plan;
  p->h                 = h;
  p->n                 = n;
  const int s = h->cfg.spacedim, n_sub = h->cfg.n_subdivisions, full = 2 * h->cfg.oversampling + 1;
  // uniform stride = the full patch's s vectors of n_fine (what an all-gather slab uses)
  p->stride         = (size_t)s * s * (size_t)(n_sub * full + 1) * (size_t)(n_sub * full + 1);
  p->uniform_stride = offsets == nullptr;
  if (n == 0)
    {
      *out = p;
      return SLOD_OK;
    }
  // The descriptors are built ON THE DEVICE (k_make_desc: create_patches + create_mesh_for_patch +
  // index-set sizes per patch, LOD.cc:122-244,770-858), together with the plan's maxima, the id range
  // check, the realisations in use and the balanced launch order (k_balance_order): no per-patch work
  // on the host.  The host copy p->desc is one bulk read-back for the entry points that hand out
  // per-patch data (slod_compute_basis, slod_plan_patch_layout).
  (void)total;
  int n_cu = 256;
  (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->cfg.device);
  n_cu               = std::max(n_cu, 1);
  const bool balance = slod_read_tuning().balance && n > 1 && n <= 65536;
  bool       ok      = hipMalloc((void **)&p->d_desc, n * sizeof(SlodPatchDesc)) == hipSuccess;
  if (ok && balance)
    ok = hipMalloc((void **)&p->d_desc_bal, n * sizeof(SlodPatchDesc)) == hipSuccess;
  SlodPlanSummary sum;
  std::memset(&sum, 0, sizeof(sum));
  if (ok)
    {
      const hipError_t be = slod_build_descriptors(h, gids, n, offsets, p->stride, n_cu, balance, p->d_desc, p->d_desc_bal, &sum,
                                                   &p->prob_used);
      if (be != hipSuccess)
        {
          slod_plan_destroy(p);
          return hip_fail(h, be, "slod_plan_create: descriptor kernels");
        }
    }
  else
    {
      const hipError_t le = hipGetLastError();
      slod_plan_destroy(p);
      return hip_fail(h, le, "slod_plan_create: device allocation");
    }
  if (sum.error)
    {
      slod_plan_destroy(p);
      return fail(h, SLOD_ERR_ARGUMENT, "slod_plan_create: patch id out of range");
    }
  p->m_max    = sum.m_max;
  p->L_max    = sum.L_max;
  p->nc_max   = sum.nc_max;
  p->nb_max   = sum.nb_max;
  p->nn_max   = sum.nn_max;
  p->nf_max   = s * sum.nn_max;
  p->out_size = (size_t)sum.out_size;
  p->desc.resize(n);
  if (hipMemcpy(p->desc.data(), p->d_desc, n * sizeof(SlodPatchDesc), hipMemcpyDeviceToHost) != hipSuccess)
    {
      const hipError_t le = hipGetLastError();
      slod_plan_destroy(p);
      return hip_fail(h, le, "slod_plan_create: descriptor read-back");
    }
  if (p->m_max > 16 * 7)
    {
      slod_plan_destroy(p);
      return fail(h, SLOD_ERR_UNSUPPORTED, "slod_plan_create: more than 112 dofs per grid line");
    }
  if (p->nc_max > 64)
    {
      slod_plan_destroy(p);
      return fail(h, SLOD_ERR_UNSUPPORTED, "slod_plan_create: more than 64 coarse dofs per patch");
    }
  // k_select reduces the boundary-trace matrix by QR in row chunks (TSQR): the LDS buffer
  // holds nb_buf rows, at least nc_max + 16 so every chunk brings new rows
  // (vector problems: 80 rows keep k_select<2> under 80 KB of LDS, i.e. two workgroups per CU)
  p->nb_buf = std::min(p->nb_max, std::max(s == 1 ? 96 : 80, p->nc_max + 16));
  // the kernel family, its LDS size and the fused stages are fixed here, once (the same function
  // the launch uses): a plan that no kernel can run is rejected now, not at execute
  if (p->nb_buf > 160) // k_select's register-resident reflector holds 16 lanes x NRL = 160 rows (slod_select.hip.h)
    {
      slod_plan_destroy(p);
      return fail(h, SLOD_ERR_UNSUPPORTED, "slod_plan_create: boundary-trace buffer beyond the selection kernel's 160 rows");
    }
  if (!slod_choose_solver(s, n_sub, p->m_max, p->L_max, p->nc_max, p->nb_buf, p->nf_max, n, slod_read_tuning(), &p->choice))
    {
      slod_plan_destroy(p);
      return fail(h, SLOD_ERR_UNSUPPORTED, "slod_plan_create: patch does not fit the 160 KB LDS of any solver kernel");
    }
  p->nn_max    = (p->nn_max + 31) & ~31; // 256-byte aligned stencil planes
  p->st_stride = (size_t)9 * s * s * p->nn_max;
  p->v_stride  = p->choice.v_patch_elems ? p->choice.v_patch_elems : (size_t)p->L_max * p->choice.v_line_elems;
  p->x_stride  = (size_t)p->L_max * p->m_max * p->nc_max;
  const bool   own_z = p->choice.kind == SLOD_K_TW;
  const size_t per_patch = (p->st_stride + p->v_stride + (own_z ? 2 : 1) * p->x_stride) * sizeof(double);
  // Workspace budget: 24 GB, or 60 % of what is free on the device if that is more (an MI355X has
  // 288 GB: big plans then run in few, long launches -- C3 in 3 chunks instead of 14, less idle
  // tail per chunk), never more than 80 % of what is free right now (the caller's outputs and
  // further plans of the handle need room too: allocate outputs BEFORE the plan, or they compete
  // with it); SLOD_WORKSPACE_MB overrides.  If an allocation still fails the chunk is halved and
  // tried again: the plan degrades to more, shorter launches instead of failing.
  size_t budget_mb = 24 * 1024;
  {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
      {
        const size_t free_mb = free_b / (1024 * 1024);
        budget_mb = std::min(std::max<size_t>(budget_mb, free_mb * 6 / 10), std::max<size_t>(64, free_mb * 8 / 10));
      }
  }
  if (const char *env = std::getenv("SLOD_WORKSPACE_MB"))
    budget_mb = (size_t)std::max(1L, std::atol(env));
  p->chunk = std::max<size_t>(1, std::min<size_t>(n, budget_mb * 1024 * 1024 / per_patch));
  ok = true;
  // Guards: the kernels read the banded neighbours of a row / node without range tests (the
  // matching band coefficient is zero, or the result is dropped), so up to a few rows before the
  // first and after the last slot are touched.  X is zero-filled once: a skipped product must meet
  // a finite number.
  p->guard = (size_t)24 * p->nc_max + 64;
  const size_t st_slack = (size_t)4 * p->nn_max;
  auto free_ws = [&]() {
    for (double **q : {&p->ws_st, &p->ws_v, &p->ws_x_alloc, &p->ws_z, &p->ws_m})
      if (*q)
        {
          (void)hipFree(*q);
          *q = nullptr;
        }
    p->ws_x = nullptr;
  };
  for (;;)
    {
      bool w = ok;
      w      = w && hipMalloc((void **)&p->ws_st, (p->chunk * p->st_stride + st_slack) * sizeof(double)) == hipSuccess;
      w      = w && hipMalloc((void **)&p->ws_v, std::max<size_t>(1, p->chunk * p->v_stride) * sizeof(double)) == hipSuccess;
      w      = w && hipMalloc((void **)&p->ws_x_alloc, (p->chunk * p->x_stride + 2 * p->guard) * sizeof(double)) == hipSuccess;
      if (own_z)
        w = w && hipMalloc((void **)&p->ws_z, (p->chunk * p->x_stride + p->guard) * sizeof(double)) == hipSuccess;
      w = w && hipMalloc((void **)&p->ws_m, p->chunk * (size_t)p->nc_max * p->nc_max * sizeof(double)) == hipSuccess;
      if (w || !ok || p->chunk == 1)
        {
          ok = w;
          break;
        }
      (void)hipGetLastError(); // out of memory: clear it, halve the chunk, try again
      free_ws();
      p->chunk = (p->chunk + 1) / 2;
    }
  ok      = ok && hipMemset(p->ws_st, 0, (p->chunk * p->st_stride + st_slack) * sizeof(double)) == hipSuccess;
  ok      = ok && hipMemset(p->ws_x_alloc, 0, (p->chunk * p->x_stride + 2 * p->guard) * sizeof(double)) == hipSuccess;
  p->ws_x = p->ws_x_alloc ? p->ws_x_alloc + p->guard : nullptr;
  if (own_z)
    ok = ok && hipMemset(p->ws_z, 0, (p->chunk * p->x_stride + p->guard) * sizeof(double)) == hipSuccess;
  ok       = ok && hipMalloc((void **)&p->d_status, sizeof(int32_t)) == hipSuccess;
  ok       = ok && hipMalloc((void **)&p->d_pdiag, n * (size_t)s * sizeof(SlodPatchDiag)) == hipSuccess;
  ok       = ok && hipMemset(p->d_pdiag, 0, n * (size_t)s * sizeof(SlodPatchDiag)) == hipSuccess;
  // Launch order (d_desc_bal, built by k_balance_order above).  All workgroups of a launch are resident
  // at once (a few per CU) and the step ends with the slowest CU; measured on MI355X
  // (tools/patch_timeline.py) the blocks b, b + n_cu, b + 2 n_cu, ... share a CU.  Patches ranked by
  // estimated cost (canonical solve flops, rim patches are cheaper) and dealt in a snake over rows of
  // n_cu give every CU the same mix.
  ok = ok && hipMemset(p->d_status, 0, sizeof(int32_t)) == hipSuccess;
  p->n_chunks = (n + p->chunk - 1) / p->chunk;
  p->ev.assign(4 * p->n_chunks, nullptr);
  for (auto &ev : p->ev)
    ok = ok && hipEventCreate(&ev) == hipSuccess;
  if (!ok)
    {
      const hipError_t le = hipGetLastError();
      slod_plan_destroy(p);
      return hip_fail(h, le, "slod_plan_create: device allocation");
    }
  *out = p;
  return SLOD_OK;
}

void slod_plan_destroy(slod_plan *p)
{
  if (!p)
    return;
  for (auto &ev : p->ev)
    if (ev)
      (void)hipEventDestroy(ev);
  if (p->d_desc)
    (void)hipFree(p->d_desc);
  if (p->d_desc_bal)
    (void)hipFree(p->d_desc_bal);
  if (p->ws_st)
    (void)hipFree(p->ws_st);
  if (p->ws_v)
    (void)hipFree(p->ws_v);
  if (p->ws_x_alloc)
    (void)hipFree(p->ws_x_alloc);
  if (p->ws_z)
    (void)hipFree(p->ws_z);
  if (p->ws_m)
    (void)hipFree(p->ws_m);
  if (p->d_status)
    (void)hipFree(p->d_status);
  for (void *q : {(void *)p->alt.ws_st, (void *)p->alt.ws_v, (void *)p->alt.ws_x_alloc, (void *)p->alt.ws_z, (void *)p->alt.ws_m,
                  (void *)p->alt.d_status})
    if (q)
      (void)hipFree(q);
  for (int k = 0; k < 2; ++k)
    {
      if (p->istream[k])
        (void)hipStreamDestroy(p->istream[k]);
      if (p->done_ev[k])
        (void)hipEventDestroy(p->done_ev[k]);
    }
  if (p->fork_ev)
    (void)hipEventDestroy(p->fork_ev);
  if (p->d_pdiag)
    (void)hipFree(p->d_pdiag);
  delete p;
}

size_t slod_plan_stride(const slod_plan *p) { return p ? p->stride : 0; }
size_t slod_plan_output_size(const slod_plan *p) { return p ? p->out_size : 0; }

int slod_plan_execute(slod_plan *p, double *d_basis, double *d_premult, void *hip_stream)
{
  if (!p)
    return SLOD_ERR_ARGUMENT;
  slod_handle *h = p->h;
  if (p->n == 0)
    return SLOD_OK;
  if (!d_basis || !d_premult)
    return fail(h, SLOD_ERR_ARGUMENT, "slod_plan_execute: null output pointer");
  for (size_t pb = 0; pb < p->prob_used.size(); ++pb) // realisations the plan reads (flagged by k_make_desc)
    for (int f = 0; f < h->cfg.spacedim && p->prob_used[pb]; ++f)
      if (!h->coef_set[pb * 2 + f])
        return fail(h, SLOD_ERR_STATE, "slod_plan_execute: coefficient field not set");
  (void)hipSetDevice(h->cfg.device);
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : h->stream;
  const int   s  = h->cfg.spacedim;
  hipError_t  e  = hipSuccess;
  int         slot = 0;
  if (p->overlap == 2)
    {
      // consecutive executes alternate between two workspaces and two internal streams: the work is
      // ordered after everything submitted to the caller's stream so far, the caller's stream is NOT
      // ordered after it (slod_plan_join / slod_plan_status do that)
      slot            = (int)(p->n_exec & 1);
      hipStream_t own = p->istream[slot];
      e               = hipEventRecord(p->fork_ev, st);
      if (e == hipSuccess)
        e = hipStreamWaitEvent(own, p->fork_ev, 0);
      st = own;
    }
  if (e == hipSuccess)
    e = hipMemsetAsync(slot ? p->alt.d_status : p->d_status, 0, sizeof(int32_t), st);
  size_t ci = 0;
  (void)s;
  for (size_t first = 0; first < p->n && e == hipSuccess; first += p->chunk, ++ci)
    e = launch_range(p, first, (int)std::min(p->chunk, p->n - first), d_basis, d_premult, st,
                     &p->ev[4 * ((p->n_exec % (size_t)p->depth) * p->n_chunks + ci)], true, slot);
  if (e == hipSuccess && p->overlap == 2)
    {
      e                 = hipEventRecord(p->done_ev[slot], st);
      p->inflight[slot] = true;
    }
  if (e != hipSuccess)
    return hip_fail(h, e, "slod_plan_execute");
  p->ran = true;
  ++p->n_exec;
  return SLOD_OK;
}

int slod_plan_set_overlap(slod_plan *p, int depth)
{
  if (!p || (depth != 1 && depth != 2))
    return SLOD_ERR_ARGUMENT;
  slod_handle *h = p->h;
  if (p->n == 0 || depth == p->overlap)
    return SLOD_OK;
  (void)hipSetDevice(h->cfg.device);
  (void)hipDeviceSynchronize();
  if (depth == 1)
    {
      p->overlap = 1; // the second workspace stays allocated until the plan is destroyed
      return SLOD_OK;
    }
  if (p->n_chunks != 1)
    return fail(h, SLOD_ERR_STATE, "slod_plan_set_overlap: the plan runs in several workspace chunks (they de-phase by themselves)");
  const bool own_z = p->ws_z != nullptr;
  const size_t st_slack = (size_t)4 * p->nn_max;
  bool ok = true;
  if (!p->alt.ws_st)
    {
      ok = ok && hipMalloc((void **)&p->alt.ws_st, (p->chunk * p->st_stride + st_slack) * sizeof(double)) == hipSuccess;
      ok = ok && hipMalloc((void **)&p->alt.ws_v, std::max<size_t>(1, p->chunk * p->v_stride) * sizeof(double)) == hipSuccess;
      ok = ok && hipMalloc((void **)&p->alt.ws_x_alloc, (p->chunk * p->x_stride + 2 * p->guard) * sizeof(double)) == hipSuccess;
      if (own_z)
        ok = ok && hipMalloc((void **)&p->alt.ws_z, (p->chunk * p->x_stride + p->guard) * sizeof(double)) == hipSuccess;
      ok = ok && hipMalloc((void **)&p->alt.ws_m, p->chunk * (size_t)p->nc_max * p->nc_max * sizeof(double)) == hipSuccess;
      ok = ok && hipMalloc((void **)&p->alt.d_status, sizeof(int32_t)) == hipSuccess;
      ok = ok && hipMemset(p->alt.ws_st, 0, (p->chunk * p->st_stride + st_slack) * sizeof(double)) == hipSuccess;
      ok = ok && hipMemset(p->alt.ws_x_alloc, 0, (p->chunk * p->x_stride + 2 * p->guard) * sizeof(double)) == hipSuccess;
      if (own_z)
        ok = ok && hipMemset(p->alt.ws_z, 0, (p->chunk * p->x_stride + p->guard) * sizeof(double)) == hipSuccess;
      ok = ok && hipMemset(p->alt.d_status, 0, sizeof(int32_t)) == hipSuccess;
      p->alt.ws_x = p->alt.ws_x_alloc ? p->alt.ws_x_alloc + p->guard : nullptr;
      for (int k = 0; k < 2 && ok; ++k)
        {
          ok = ok && hipStreamCreateWithFlags(&p->istream[k], hipStreamNonBlocking) == hipSuccess;
          ok = ok && hipEventCreateWithFlags(&p->done_ev[k], hipEventDisableTiming) == hipSuccess;
        }
      ok = ok && hipEventCreateWithFlags(&p->fork_ev, hipEventDisableTiming) == hipSuccess;
    }
  if (!ok)
    return hip_fail(h, hipGetLastError(), "slod_plan_set_overlap: device allocation");
  p->overlap = 2;
  p->n_exec  = 0; // the profile slots start over
  return SLOD_OK;
}

int slod_plan_join(slod_plan *p, void *hip_stream)
{
  if (!p)
    return SLOD_ERR_ARGUMENT;
  if (p->overlap != 2 && !p->inflight[0] && !p->inflight[1])
    return SLOD_OK;
  (void)hipSetDevice(p->h->cfg.device);
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : p->h->stream;
  hipError_t  e  = hipSuccess;
  for (int k = 0; k < 2 && e == hipSuccess; ++k)
    if (p->inflight[k])
      e = hipStreamWaitEvent(st, p->done_ev[k], 0);
  return e == hipSuccess ? SLOD_OK : hip_fail(p->h, e, "slod_plan_join");
}

int slod_plan_patch_layout(slod_plan *p, size_t k, int launch_order, slod_patch_info *info, uint32_t *plan_index)
{
  if (!p || !info)
    return SLOD_ERR_ARGUMENT;
  if (k >= p->n)
    return fail(p->h, SLOD_ERR_ARGUMENT, "slod_plan_patch_layout: index out of range");
  (void)hipSetDevice(p->h->cfg.device);
  SlodPatchDesc    d;
  const SlodPatchDesc *src = (launch_order && p->d_desc_bal ? p->d_desc_bal : p->d_desc) + k;
  if (hipMemcpy(&d, src, sizeof(d), hipMemcpyDeviceToHost) != hipSuccess)
    return hip_fail(p->h, hipGetLastError(), "slod_plan_patch_layout");
  slod_desc_to_info(p->h, d, info);
  if (plan_index)
    *plan_index = d.plan_index;
  return SLOD_OK;
}

int slod_plan_diagnostics(slod_plan *p, slod_patch_diag *out, size_t capacity)
{
  static_assert(sizeof(slod_patch_diag) == sizeof(SlodPatchDiag), "slod_patch_diag layout");
  if (!p || (!out && p->n))
    return SLOD_ERR_ARGUMENT;
  const size_t cnt = p->n * (size_t)p->h->cfg.spacedim;
  if (capacity < cnt)
    return fail(p->h, SLOD_ERR_ARGUMENT, "slod_plan_diagnostics: buffer too small");
  if (!p->ran)
    return fail(p->h, SLOD_ERR_STATE, "slod_plan_diagnostics: plan has not been executed");
  (void)hipSetDevice(p->h->cfg.device);
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess)
    e = hipMemcpy(out, p->d_pdiag, cnt * sizeof(SlodPatchDiag), hipMemcpyDeviceToHost);
  if (e != hipSuccess)
    return hip_fail(p->h, e, "slod_plan_diagnostics");
  return (int)cnt;
}

#ifdef SLOD_ENABLE_DIAG
// timing experiments only (lib/libslod_hip_diag.so, not part of include/slod.h): the per-patch
// scratch block `ms` of the first workspace chunk, which the kernels fill with clock stamps under
// SLOD_DIAG bit 20
int slod_debug_read_ms(slod_plan *p, double *out, size_t count)
{
  if (!p || !out)
    return SLOD_ERR_ARGUMENT;
  const size_t have = p->chunk * (size_t)p->nc_max * p->nc_max;
  (void)hipSetDevice(p->h->cfg.device);
  (void)hipDeviceSynchronize();
  return hipMemcpy(out, p->ws_m, std::min(count, have) * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess
             ? SLOD_OK
             : SLOD_ERR_DEVICE;
}
#endif

int slod_plan_profile(slod_plan *p, int depth)
{
  if (!p || depth < 1)
    return SLOD_ERR_ARGUMENT;
  if (p->n == 0)
    return SLOD_OK;
  (void)hipSetDevice(p->h->cfg.device);
  (void)hipDeviceSynchronize();
  for (auto &ev : p->ev)
    if (ev)
      (void)hipEventDestroy(ev);
  p->ev.assign((size_t)depth * p->n_chunks * 4, nullptr);
  for (auto &ev : p->ev)
    if (hipEventCreate(&ev) != hipSuccess)
      return hip_fail(p->h, hipGetLastError(), "slod_plan_profile: hipEventCreate");
  p->depth  = depth;
  p->n_exec = 0;
  p->ran    = false;
  return SLOD_OK;
}

int slod_plan_kernel_ms(slod_plan *p, float ms[3])
{
  if (!p || !ms)
    return SLOD_ERR_ARGUMENT;
  ms[0] = ms[1] = ms[2] = 0.f;
  if (!p->ran)
    return fail(p->h, SLOD_ERR_STATE, "slod_plan_kernel_ms: plan has not been executed");
  const size_t slots = std::min<size_t>(p->n_exec, (size_t)p->depth);
  if (slots == 0)
    return fail(p->h, SLOD_ERR_STATE, "slod_plan_kernel_ms: the last execute recorded no events (slod_plan_execute_allgather)");
  double       acc[3] = {0, 0, 0};
  for (size_t sl = 0; sl < slots; ++sl)
    for (size_t c = 0; c < p->n_chunks; ++c)
      {
        hipEvent_t *ev = &p->ev[4 * (sl * p->n_chunks + c)];
        hipError_t  e  = hipEventSynchronize(ev[3]);
        for (int k = 0; k < 3 && e == hipSuccess; ++k)
          {
            float t = 0.f;
            e       = hipEventElapsedTime(&t, ev[k], ev[k + 1]);
            acc[k] += t;
          }
        if (e != hipSuccess)
          return hip_fail(p->h, e, "slod_plan_kernel_ms");
      }
  for (int k = 0; k < 3; ++k)
    ms[k] = (float)(acc[k] / (double)slots);
  return SLOD_OK;
}

int slod_plan_status(slod_plan *p)
{
  if (!p)
    return SLOD_ERR_ARGUMENT;
  if (!p->ran || p->n == 0)
    return SLOD_OK;
  int32_t    st = 0, st2 = 0;
  hipError_t e  = hipDeviceSynchronize();
  if (e == hipSuccess)
    e = hipMemcpy(&st, p->d_status, sizeof(st), hipMemcpyDeviceToHost);
  if (e == hipSuccess && p->alt.d_status)
    e = hipMemcpy(&st2, p->alt.d_status, sizeof(st2), hipMemcpyDeviceToHost);
  st |= st2;
  p->inflight[0] = p->inflight[1] = false;
  if (e != hipSuccess)
    return hip_fail(p->h, e, "slod_plan_status");
  if (st)
    return fail(p->h, SLOD_ERR_NUMERIC, "non-positive pivot in a patch solve (status bits " +
                                          std::to_string(st) + ")");
  return SLOD_OK;
}

int slod_compute_basis(slod_handle *h, const uint32_t *gids, size_t n, double *basis, double *premult,
                       const uint64_t *offsets)
{
  if (!h || (n && (!gids || !basis || !premult)))
    return SLOD_ERR_ARGUMENT;
  slod_plan *p  = nullptr;
  int        rc = slod_plan_create(h, gids, n, offsets, &p);
  if (rc)
    return rc;
  if (n == 0)
    {
      slod_plan_destroy(p);
      return SLOD_OK;
    }
  const size_t bytes = slod_plan_output_size(p) * sizeof(double);
  double      *d_b = nullptr, *d_p = nullptr;
  hipError_t   e = hipMalloc((void **)&d_b, bytes);
  if (e == hipSuccess)
    e = hipMalloc((void **)&d_p, bytes);
  // gaps between ragged patches stay defined
  if (e == hipSuccess)
    e = hipMemsetAsync(d_b, 0, bytes, h->stream);
  if (e == hipSuccess)
    e = hipMemsetAsync(d_p, 0, bytes, h->stream);
  if (e != hipSuccess)
    rc = hip_fail(h, e, "slod_compute_basis: output allocation");
  if (!rc)
    rc = slod_plan_execute(p, d_b, d_p, nullptr);
  if (!rc)
    rc = slod_plan_status(p);
  if (!rc)
    {
      // copy back only what the plan wrote (caller gaps untouched)
      const int s = h->cfg.spacedim;
      for (size_t k = 0; k < n && e == hipSuccess; ++k)
        {
          const SlodPatchDesc &d   = p->desc[k];
          const size_t         len = (size_t)s * s * (d.nx + 1) * (d.ny + 1) * sizeof(double);
          e = hipMemcpyAsync(basis + d.out_off, d_b + d.out_off, len, hipMemcpyDeviceToHost, h->stream);
          if (e == hipSuccess)
            e = hipMemcpyAsync(premult + d.out_off, d_p + d.out_off, len, hipMemcpyDeviceToHost,
                               h->stream);
        }
      if (e == hipSuccess)
        e = hipStreamSynchronize(h->stream);
      if (e != hipSuccess)
        rc = hip_fail(h, e, "slod_compute_basis: copy back");
    }
  if (d_b)
    (void)hipFree(d_b);
  if (d_p)
    (void)hipFree(d_p);
  slod_plan_destroy(p);
  return rc;
}

// ---- pieces for parity tests: run the pipeline for ONE patch and read the workspace ----
static int run_single(slod_handle *h, uint32_t gid, slod_plan **pp, double **db, double **dp)
{
  int rc = slod_plan_create(h, &gid, 1, nullptr, pp);
  if (rc)
    return rc;
  const size_t bytes = slod_plan_output_size(*pp) * sizeof(double);
  hipError_t   e     = hipMalloc((void **)db, bytes);
  if (e == hipSuccess)
    e = hipMalloc((void **)dp, bytes);
  if (e != hipSuccess)
    return hip_fail(h, e, "slod debug: allocation");
  rc = slod_plan_execute(*pp, *db, *dp, nullptr);
  if (!rc)
    {
      e = hipStreamSynchronize(h->stream);
      if (e != hipSuccess)
        rc = hip_fail(h, e, "slod debug: execute");
    }
  return rc;
}

int slod_assemble_stiffness_for_patch(slod_handle *h, uint32_t gid, double *stencil)
{
  if (!h || !stencil)
    return SLOD_ERR_ARGUMENT;
  slod_plan *p  = nullptr;
  double    *db = nullptr, *dp = nullptr;
  int        rc = run_single(h, gid, &p, &db, &dp);
  if (!rc)
    {
      const SlodPatchDesc &d  = p->desc[0];
      const int            s  = h->cfg.spacedim, nn = (d.nx + 1) * (d.ny + 1);
      std::vector<double>  tmp(p->st_stride);
      hipError_t e = hipMemcpy(tmp.data(), p->ws_st, p->st_stride * sizeof(double), hipMemcpyDeviceToHost);
      if (e != hipSuccess)
        rc = hip_fail(h, e, "slod_assemble_stiffness_for_patch");
      else
        for (int node = 0; node < nn; ++node)
          for (int dir = 0; dir < 9; ++dir)
            for (int a = 0; a < s; ++a)
              for (int b = 0; b < s; ++b)
                stencil[(((size_t)node * 9 + dir) * s + a) * s + b] =
                  tmp[(size_t)((dir * s + a) * s + b) * p->nn_max + node];
    }
  if (db)
    (void)hipFree(db);
  if (dp)
    (void)hipFree(dp);
  slod_plan_destroy(p);
  return rc;
}

int slod_patch_solution(slod_handle *h, uint32_t gid, double *X)
{
  if (!h || !X)
    return SLOD_ERR_ARGUMENT;
  slod_plan *p  = nullptr;
  double    *db = nullptr, *dp = nullptr;
  int        rc = run_single(h, gid, &p, &db, &dp);
  if (!rc)
    {
      const SlodPatchDesc &d = p->desc[0];
      const int            s = h->cfg.spacedim, npx = d.nx + 1, nf = s * npx * (d.ny + 1);
      const bool           tr = (d.flags & SLOD_F_TRANSPOSED) != 0;
      std::vector<double>  tmp(p->x_stride);
      hipError_t e = hipMemcpy(tmp.data(), p->ws_x, p->x_stride * sizeof(double), hipMemcpyDeviceToHost);
      if (e != hipSuccess)
        rc = hip_fail(h, e, "slod_patch_solution");
      else
        {
          std::fill(X, X + (size_t)nf * d.n_c, 0.0);
          const size_t xline = (size_t)p->m_max * p->nc_max;
          for (int iy = 1; iy < d.ny; ++iy)
            for (int ix = 1; ix < d.nx; ++ix)
              for (int c = 0; c < s; ++c)
                {
                  const int l = tr ? ix - 1 : iy - 1, pos = tr ? iy - 1 : ix - 1;
                  const double *row = tmp.data() + (size_t)l * xline + (size_t)(pos * s + c) * p->nc_max;
                  for (int k = 0; k < d.n_c; ++k)
                    X[((size_t)(ix + iy * npx) * s + c) * d.n_c + k] = row[k];
                }
        }
    }
  if (db)
    (void)hipFree(db);
  if (dp)
    (void)hipFree(dp);
  slod_plan_destroy(p);
  return rc;
}

} // extern "C"

// ---- multi-GPU exchange for a C/C++ host: RCCL, resolved at run time -------------------
// librccl is dlopen'ed on first use (RTLD_NOLOAD first: a process that already runs RCCL --
// PyTorch ships its own copy -- must not get a second one); the library has no link-time
// dependency on it and single-GPU users never load it.
#include <dlfcn.h>

namespace
{
  struct Rccl
  {
    void *lib = nullptr;
    int (*GetUniqueId)(void *)                                                             = nullptr;
    int (*CommInitRank)(void **, int, slod_comm_id, int)                                   = nullptr;
    int (*CommDestroy)(void *)                                                             = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t)               = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t)          = nullptr;
    int (*GroupStart)()                                                                    = nullptr;
    int (*GroupEnd)()                                                                      = nullptr;
    const char *(*GetErrorString)(int)                                                     = nullptr;
    std::string error;
  };
  Rccl &rccl()
  {
    static Rccl r;
    if (r.lib || !r.error.empty())
      return r;
    const char *names[] = {"librccl.so", "librccl.so.1"};
    for (const char *n : names)
      if (!r.lib)
        r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char *n : names)
      if (!r.lib)
        r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!r.lib)
      {
        r.error = std::string("librccl not found: ") + dlerror();
        return r;
      }
    auto sym = [&](const char *n) {
      void *f = dlsym(r.lib, n);
      if (!f && r.error.empty())
        r.error = std::string("librccl lacks ") + n;
      return f;
    };
    r.GetUniqueId    = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank   = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy    = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllGather      = (decltype(r.AllGather))sym("ncclAllGather");
    r.Broadcast      = (decltype(r.Broadcast))sym("ncclBroadcast");
    r.GroupStart     = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd       = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    return r;
  }
  constexpr int kNcclFloat64 = 8; // rccl.h: ncclFloat64 = ncclDouble = 8
} // namespace

struct slod_comm
{
  void       *nccl = nullptr;
  int         n_ranks = 1, rank = 0, device = 0;
  std::string error;
};

namespace
{
  thread_local std::string g_comm_error;
  int comm_fail(slod_comm *c, int code, const std::string &msg)
  {
    (c ? c->error : g_comm_error) = msg;
    return code;
  }
  int nccl_fail(slod_comm *c, int rc, const char *what)
  {
    return comm_fail(c, SLOD_ERR_DEVICE, std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(rc) : "?"));
  }
} // namespace

extern "C" {

const char *slod_comm_last_error(const slod_comm *c) { return c ? c->error.c_str() : g_comm_error.c_str(); }

int slod_comm_unique_id(slod_comm_id *id)
{
  if (!id)
    return SLOD_ERR_ARGUMENT;
  Rccl &r = rccl();
  if (!r.error.empty())
    return comm_fail(nullptr, SLOD_ERR_DEVICE, r.error);
  const int rc = r.GetUniqueId(id);
  return rc ? nccl_fail(nullptr, rc, "ncclGetUniqueId") : SLOD_OK;
}

int slod_comm_create(const slod_comm_id *id, int n_ranks, int rank, int device, slod_comm **out)
{
  if (!id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks)
    return SLOD_ERR_ARGUMENT;
  *out    = nullptr;
  Rccl &r = rccl();
  if (!r.error.empty())
    return comm_fail(nullptr, SLOD_ERR_DEVICE, r.error);
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess)
    return comm_fail(nullptr, SLOD_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
  slod_comm *c = new slod_comm;
  c->n_ranks   = n_ranks;
  c->rank      = rank;
  c->device    = device;
  const int rc = r.CommInitRank(&c->nccl, n_ranks, *id, rank);
  if (rc)
    {
      const int code = nccl_fail(nullptr, rc, "ncclCommInitRank");
      delete c;
      return code;
    }
  *out = c;
  return SLOD_OK;
}

void slod_comm_destroy(slod_comm *c)
{
  if (!c)
    return;
  if (c->nccl && rccl().CommDestroy)
    (void)rccl().CommDestroy(c->nccl);
  delete c;
}

int slod_comm_allgather(slod_comm *c, const double *d_send, double *d_recv, size_t count, void *hip_stream)
{
  if (!c || !d_send || !d_recv)
    return SLOD_ERR_ARGUMENT;
  const int rc = rccl().AllGather(d_send, d_recv, count, kNcclFloat64, c->nccl, (hipStream_t)hip_stream);
  return rc ? nccl_fail(c, rc, "ncclAllGather") : SLOD_OK;
}

int slod_gather_piece(uint64_t patches_per_rank, uint32_t n_pieces, uint32_t piece, uint64_t *first, uint64_t *count)
{
  if (!first || !count || n_pieces == 0 || piece >= n_pieces)
    return SLOD_ERR_ARGUMENT;
  const uint64_t pp = (patches_per_rank + n_pieces - 1) / n_pieces;
  *first            = std::min<uint64_t>((uint64_t)piece * pp, patches_per_rank);
  *count            = std::min<uint64_t>(pp, patches_per_rank - *first);
  return SLOD_OK;
}

int slod_plan_execute_allgather(slod_plan *p, slod_comm *c, double *d_basis_all, double *d_premult_all,
                                size_t patches_per_rank, int n_pieces, void *compute_stream, void *comm_stream)
{
  if (!p || !c || n_pieces < 1)
    return SLOD_ERR_ARGUMENT;
  slod_handle *h = p->h;
  if (!d_basis_all || !d_premult_all || !compute_stream || !comm_stream || compute_stream == comm_stream)
    return fail(h, SLOD_ERR_ARGUMENT, "slod_plan_execute_allgather: needs two distinct streams and both slabs");
  if (p->n > patches_per_rank)
    return fail(h, SLOD_ERR_ARGUMENT, "slod_plan_execute_allgather: plan larger than a rank's slab");
  if (!p->uniform_stride)
    return fail(h, SLOD_ERR_STATE, "slod_plan_execute_allgather: the plan must use the uniform stride (NULL offsets)");
  for (size_t pb = 0; pb < p->prob_used.size(); ++pb)
    for (int f = 0; f < h->cfg.spacedim && p->prob_used[pb]; ++f)
      if (!h->coef_set[pb * 2 + f])
        return fail(h, SLOD_ERR_STATE, "slod_plan_execute_allgather: coefficient field not set");
  (void)hipSetDevice(h->cfg.device);
  hipStream_t  cs = (hipStream_t)compute_stream, ns = (hipStream_t)comm_stream;
  const size_t slab = patches_per_rank * p->stride;
  double      *mb = d_basis_all + (size_t)c->rank * slab, *mp = d_premult_all + (size_t)c->rank * slab;
  hipError_t   e  = p->n ? hipMemsetAsync(p->d_status, 0, sizeof(int32_t), cs) : hipSuccess;
  Rccl        &r  = rccl();
  for (int piece = 0; piece < n_pieces && e == hipSuccess; ++piece)
    {
      uint64_t first, count;
      (void)slod_gather_piece(patches_per_rank, (uint32_t)n_pieces, (uint32_t)piece, &first, &count);
      if (count == 0)
        break;
      // this rank's patches of the piece (the padded tail of a rank with fewer patches stays as it is)
      const size_t mine = first < p->n ? std::min<size_t>(count, p->n - first) : 0;
      for (size_t k = 0; k < mine && e == hipSuccess; k += p->chunk)
        e = launch_range(p, first + k, (int)std::min(p->chunk, mine - k), mb, mp, cs, nullptr, false);
      // the exchange of piece i overlaps the computation of piece i + 1: the communication stream
      // waits for the piece only
      hipEvent_t ev = nullptr;
      if (e == hipSuccess)
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
      if (e == hipSuccess)
        e = hipEventRecord(ev, cs);
      if (e == hipSuccess)
        e = hipStreamWaitEvent(ns, ev, 0);
      if (ev)
        (void)hipEventDestroy(ev); // released when the recorded work has completed
      if (e != hipSuccess)
        break;
      // every rank's piece to every rank, in place: one broadcast per root and array, grouped
      // (the pieces of different ranks are not contiguous in the gathered slab, so this is not
      // an ncclAllGather; xGMI is point to point: the group drives all links at once)
      int rc = r.GroupStart();
      for (int root = 0; root < c->n_ranks && !rc; ++root)
        {
          double *pb = d_basis_all + (size_t)root * slab + first * p->stride;
          double *pq = d_premult_all + (size_t)root * slab + first * p->stride;
          rc         = r.Broadcast(pb, pb, count * p->stride, kNcclFloat64, root, c->nccl, ns);
          if (!rc)
            rc = r.Broadcast(pq, pq, count * p->stride, kNcclFloat64, root, c->nccl, ns);
        }
      const int rc2 = r.GroupEnd();
      if (rc || rc2)
        return nccl_fail(c, rc ? rc : rc2, "ncclBroadcast (grouped)"), fail(h, SLOD_ERR_DEVICE, c->error);
    }
  // later work on the compute stream sees the gathered slabs
  hipEvent_t done = nullptr;
  if (e == hipSuccess)
    e = hipEventCreateWithFlags(&done, hipEventDisableTiming);
  if (e == hipSuccess)
    e = hipEventRecord(done, ns);
  if (e == hipSuccess)
    e = hipStreamWaitEvent(cs, done, 0);
  if (done)
    (void)hipEventDestroy(done);
  if (e != hipSuccess)
    return hip_fail(h, e, "slod_plan_execute_allgather");
  if (p->n)
    {
      p->ran    = true;
      p->n_exec = 0; // no events were recorded for this execute: slod_plan_kernel_ms has nothing to report
    }
  return SLOD_OK;
}

} // extern "C"
#pragma GCC visibility pop
