#include "LOD.h"

#include <chrono>
#include <cstdlib>

namespace slod
{
  template <int dim, int spacedim>
  LOD<dim, spacedim>::LOD(const LODParameters<dim, spacedim> &par)
    : par(par)
  {
    if (const char *r = std::getenv("RANK"))
      this_mpi_process = (unsigned int)std::atoi(r);
    if (const char *w = std::getenv("WORLD_SIZE"))
      n_mpi_processes = (unsigned int)std::max(1, std::atoi(w));
  }

  template <int dim, int spacedim>
  LOD<dim, spacedim>::~LOD()
  {
    slod_destroy(handle);
  }

  template <int dim, int spacedim>
  void LOD<dim, spacedim>::check(const int status, const char *what) const
  {
    if (status < 0)
      throw std::runtime_error(std::string(what) + ": " + slod_last_error(handle));
  }

  // GridGenerator::hyper_cube + refine_global + evenly distributed partitioning (LOD.cc:110-119)
  template <int dim, int spacedim>
  void LOD<dim, spacedim>::make_grid()
  {
    slod_config cfg{};
    cfg.dim                   = dim;
    cfg.spacedim              = spacedim;
    cfg.n_global_refinements  = (int32_t)par.n_global_refinements;
    cfg.n_subdivisions        = (int32_t)par.n_subdivisions;
    cfg.oversampling          = (int32_t)par.oversampling;
    cfg.lod_stabilization     = par.LOD_stabilization;
    cfg.constant_coefficients = par.constant_coefficients;
    cfg.projection_quirk      = par.projection_quirk;
    cfg.n_problems            = 1;
    cfg.device                = par.device;
    slod_handle *h            = nullptr;
    if (slod_create(&cfg, &h) != SLOD_OK)
      throw std::runtime_error(std::string("slod_create: ") + slod_last_error(nullptr));
    handle = h;
    uint64_t b = 0, e = 0;
    check(slod_partition((uint64_t)slod_num_patches(handle), n_mpi_processes, this_mpi_process, &b, &e),
          "slod_partition");
    locally_owned_patches = {(unsigned int)b, (unsigned int)e};
  }

  // the FE spaces are fixed by (n_subdivisions, spacedim): nothing to build (LOD.cc:67-106)
  template <int dim, int spacedim>
  void LOD<dim, spacedim>::make_fe()
  {}

  // LOD.cc:122-244
  template <int dim, int spacedim>
  void LOD<dim, spacedim>::create_patches()
  {
    const unsigned int n_patches = (unsigned int)slod_num_patches(handle);
    patches.clear();
    patches.resize(n_patches);
    std::size_t size_biggest_patch = 0, size_tiniest_patch = n_patches;
    for (unsigned int id = 0; id < n_patches; ++id)
      {
        slod_patch_info info;
        check(slod_patch_layout(handle, id, &info), "slod_patch_layout");
        auto &patch = patches[id];
        patch.cells.resize((std::size_t)info.mx * info.my);
        check(slod_patch_cells(handle, id, patch.cells.data(), patch.cells.size()), "slod_patch_cells");
        size_biggest_patch = std::max(size_biggest_patch, patch.cells.size());
        size_tiniest_patch = std::min(size_tiniest_patch, patch.cells.size());
      }
    if (this_mpi_process == 0)
      std::cout << "Number of coarse cell = " << n_patches << ", number of patches = " << patches.size()
                << " (locally owned: " << locally_owned_patches.second - locally_owned_patches.first
                << ") \n"
                << "Patches size in (" << size_tiniest_patch << ", " << size_biggest_patch << ")"
                << std::endl; // LOD.cc:237-242
  }

  // LOD.cc:770-858
  template <int dim, int spacedim>
  void LOD<dim, spacedim>::create_mesh_for_patch(Patch<dim> &current_patch)
  {
    const unsigned int id = (unsigned int)(&current_patch - patches.data());
    slod_patch_info    info;
    check(slod_patch_layout(handle, id, &info), "slod_patch_layout");
    PatchMesh &m = current_patch.sub_tria;
    m.x0         = info.x0;
    m.y0         = info.y0;
    m.mx         = info.mx;
    m.my         = info.my;
    m.nx         = info.nx;
    m.ny         = info.ny;
    for (int s = 0; s < 4; ++s)
      m.boundary_id[s] = info.side_domain[s] ? 0u : 99u;
    current_patch.dealii_to_lexicographic.resize(info.n_fine);
    check(slod_patch_dof_permutation(handle, id, current_patch.dealii_to_lexicographic.data(),
                                     current_patch.dealii_to_lexicographic.size()),
          "slod_patch_dof_permutation");
  }

  // LOD.cc:1380-1393
  template <int dim, int spacedim>
  void LOD<dim, spacedim>::initialize_patches()
  {
    create_patches();
    for (unsigned int id = locally_owned_patches.first; id < locally_owned_patches.second; ++id)
      create_mesh_for_patch(patches[id]);
  }

  template <int dim, int spacedim>
  void LOD<dim, spacedim>::assemble_stiffness(const unsigned int patch_id, std::vector<double> &stencil)
  {
    slod_patch_info info;
    check(slod_patch_layout(handle, patch_id, &info), "slod_patch_layout");
    stencil.assign((std::size_t)(info.n_fine / spacedim) * 9 * spacedim * spacedim, 0.0);
    check(slod_assemble_stiffness_for_patch(handle, patch_id, stencil.data()),
          "slod_assemble_stiffness_for_patch");
  }

  // LOD.cc:296-768: the whole patch loop runs on the GPU
  template <int dim, int spacedim>
  void LOD<dim, spacedim>::compute_basis_function_candidates()
  {
    // coefficient at the quadrature points of QIterated(QGauss<1>(2), n) on every fine
    // element of the global grid (what FEValues::get_quadrature_points feeds to
    // Alpha.value_list in Diffusion.h:154)
    const unsigned int N  = 1u << par.n_global_refinements;
    const unsigned int NE = N * par.n_subdivisions;
    const double       hf = 1.0 / NE, g0 = 0.5 * (1.0 - 1.0 / std::sqrt(3.0)),
                 g1 = 0.5 * (1.0 + 1.0 / std::sqrt(3.0));
    std::vector<Point<dim>> points((std::size_t)NE * NE * 4);
    for (unsigned int ey = 0; ey < NE; ++ey)
      for (unsigned int ex = 0; ex < NE; ++ex)
        for (unsigned int q = 0; q < 4; ++q)
          {
            Point<dim> &p = points[((std::size_t)ey * NE + ex) * 4 + q];
            p(0)          = (ex + ((q & 1) ? g1 : g0)) * hf;
            p(1)          = (ey + ((q & 2) ? g1 : g0)) * hf;
          }
    std::vector<double> values;
    for (unsigned int field = 0; field < (unsigned int)spacedim; ++field)
      {
        coefficients_at_quadrature_points(field, points, values);
        check(slod_set_coefficient(handle, 0, (int)field, values.data(), 1, values.size(), 0),
              "slod_set_coefficient");
      }

    const auto                t0 = std::chrono::steady_clock::now();
    const unsigned int        n  = locally_owned_patches.second - locally_owned_patches.first;
    std::vector<uint32_t>     ids(n);
    std::vector<uint64_t>     offsets(n);
    std::vector<unsigned int> n_fine(n);
    uint64_t                  total = 0;
    for (unsigned int k = 0; k < n; ++k)
      {
        ids[k] = locally_owned_patches.first + k;
        slod_patch_info info;
        check(slod_patch_layout(handle, ids[k], &info), "slod_patch_layout");
        n_fine[k]  = (unsigned int)info.n_fine;
        offsets[k] = total;
        total += (uint64_t)spacedim * info.n_fine;
      }
    std::vector<double> basis(total), premult(total);
    check(slod_compute_basis(handle, ids.data(), n, basis.data(), premult.data(), offsets.data()),
          "slod_compute_basis");
    // scatter into Patch::basis_function(_premultiplied) in the patch-local deal.II numbering
    // (LOD.cc:592,754,764; consumer LOD.cc:931-962)
    for (unsigned int k = 0; k < n; ++k)
      {
        Patch<dim> &patch = patches[ids[k]];
        patch.basis_function.assign(spacedim, std::vector<double>(n_fine[k]));
        patch.basis_function_premultiplied.assign(spacedim, std::vector<double>(n_fine[k]));
        for (int d = 0; d < spacedim; ++d)
          for (unsigned int i = 0; i < n_fine[k]; ++i)
            {
              const unsigned int lex = patch.dealii_to_lexicographic[i];
              patch.basis_function[d][i] = basis[offsets[k] + (uint64_t)d * n_fine[k] + lex];
              patch.basis_function_premultiplied[d][i] = premult[offsets[k] + (uint64_t)d * n_fine[k] + lex];
            }
      }
    last_build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }

  template <int dim, int spacedim>
  void LOD<dim, spacedim>::run()
  {
    make_grid();
    make_fe();
    initialize_patches();
    create_random_problem_coefficients();
    compute_basis_function_candidates();
  }

  template class LOD<2, 1>;
  template class LOD<2, 2>;
} // namespace slod
