// adapter/LOD_hip.cc -- deal.II side of the drop-in: the body of
//     template <int dim, int spacedim> void LOD<dim, spacedim>::compute_basis_function_candidates()
// (reference include/LOD.h:175-176, source/LOD.cc:296-768) re-implemented on top of the C-ABI of
// libslod_hip.so (include/slod.h).  NOT compiled in this repository (deal.II 9.6 + Trilinos are not
// available in the build image); it is written against the reference's headers plus the three
// header edits of adapter/dealii-slod.patch and uses no other symbol:
//   * LOD<dim,spacedim>::coefficient_values(field, points, values)   -- new virtual hook, default
//     throws; overridden in DiffusionProblem (Alpha) and ElasticityProblem (Lambda, Mu), where the
//     coefficient Functions live (reference include/Diffusion.h:68, include/Elasticity.h:111-113);
//   * everything else is existing reference state: par, patches, locally_owned_patches, fe_fine,
//     computing_timer (include/LOD.h:196-257).
// Build: adapter/CMakeLists.snippet.  source/LOD.cc keeps its own definition under
// #ifndef DEALII_SLOD_WITH_HIP (the patch adds the guard), so both variants stay buildable.
#include <deal.II/base/exceptions.h>
#include <deal.II/base/point.h>

#include <deal.II/dofs/dof_handler.h>

#include <LOD.h>
#include <slod.h>

#include <cmath>
#include <memory>
#include <vector>

namespace
{
  // slod_destroy on every exit path, AssertThrow included
  struct SlodHandleDeleter
  {
    void
    operator()(slod_handle *h) const
    {
      slod_destroy(h);
    }
  };
  using SlodHandle = std::unique_ptr<slod_handle, SlodHandleDeleter>;
} // namespace

template <int dim, int spacedim>
void
LOD<dim, spacedim>::compute_basis_function_candidates()
{
  TimerOutput::Scope t(computing_timer, "2: compute basis function (HIP)");
  AssertThrow(dim == 2, ExcNotImplemented()); // as the reference: source/LOD.cc:1470-1471

  // --- configuration = the scalars the reference loop reads (LOD.cc:325-326,357-360,563)
  slod_config cfg{};
  cfg.dim                   = dim;
  cfg.spacedim              = spacedim;
  cfg.n_global_refinements  = par.n_global_refinements;
  cfg.n_subdivisions        = par.n_subdivisions;
  cfg.oversampling          = par.oversampling;
  cfg.lod_stabilization     = par.LOD_stabilization;
  cfg.constant_coefficients = par.constant_coefficients; // reference quirk Q1 (LOD.cc:354-362)
  cfg.projection_quirk      = (spacedim == 2);           // reference quirk Q2 (LODtools.h:43-67)
  cfg.n_problems            = 1;
  cfg.device                = 0;
  slod_handle *raw = nullptr;
  AssertThrow(slod_create(&cfg, &raw) == SLOD_OK, ExcMessage(slod_last_error(nullptr)));
  SlodHandle h(raw);

  // --- coefficient at the points of quadrature_fine = QIterated(QGauss<1>(2), n) on every fine
  //     element (LOD.cc:91-92), exactly what assemble_stiffness evaluates (Diffusion.h:154,
  //     Elasticity.h:208-209); q = q0 + 2 q1
  const unsigned int      N  = 1u << par.n_global_refinements;
  const unsigned int      NE = N * par.n_subdivisions;
  std::vector<Point<dim>> pts(std::size_t(NE) * NE * 4);
  const double            g[2] = {0.5 - 0.5 / std::sqrt(3.), 0.5 + 0.5 / std::sqrt(3.)};
  const double            hf   = 1. / NE;
  for (unsigned int ey = 0; ey < NE; ++ey)
    for (unsigned int ex = 0; ex < NE; ++ex)
      for (unsigned int q = 0; q < 4; ++q)
        pts[(std::size_t(ey) * NE + ex) * 4 + q] =
          Point<dim>((ex + g[q & 1]) * hf, (ey + g[q >> 1]) * hf);
  std::vector<double> values(pts.size());
  for (unsigned int field = 0; field < spacedim; ++field)
    {
      coefficient_values(field, pts, values); // Alpha, or Lambda / Mu (see the patch)
      AssertThrow(slod_set_coefficient(h.get(), 0, field, values.data(), 1, values.size(), 0) ==
                    SLOD_OK,
                  ExcMessage(slod_last_error(h.get())));
    }

  // --- all locally owned patches in one call (LOD.cc:345)
  std::vector<uint32_t> ids;
  std::vector<uint64_t> offsets;
  uint64_t              total = 0;
  for (const auto id : locally_owned_patches)
    {
      slod_patch_info info;
      AssertThrow(slod_patch_layout(h.get(), id, &info) == SLOD_OK,
                  ExcMessage(slod_last_error(h.get())));
      AssertThrow(patches[id].cells.size() == (unsigned int)(info.mx * info.my),
                  ExcMessage("slod_patch_layout disagrees with create_patches()"));
      ids.push_back(id);
      offsets.push_back(total);
      total += uint64_t(spacedim) * info.n_fine;
    }
  std::vector<double> basis(total), premult(total);
  AssertThrow(slod_compute_basis(h.get(), ids.data(), ids.size(), basis.data(), premult.data(),
                                 offsets.data()) == SLOD_OK,
              ExcMessage(slod_last_error(h.get())));

  // --- scatter into Patch::basis_function(_premultiplied) in the numbering of dh_fine_patch
  //     (consumer: assemble_global_matrix, LOD.cc:931-962).  The permutation is NOT taken from
  //     slod_patch_dof_permutation (a re-statement of deal.II's numbering rule that cannot be
  //     checked without deal.II) but from cell->get_dof_indices() on dh_fine_patch, as the
  //     reference itself does at LOD.cc:481-483: dof -> (support point, component) ->
  //     patch-lexicographic index  spacedim * (ix + iy * (nx + 1)) + comp.
  DoFHandler<dim>                      dh_fine_patch;
  std::vector<types::global_dof_index> fine_dofs(fe_fine->n_dofs_per_cell());
  const auto                          &unit_pts = fe_fine->get_unit_support_points();
  const double                         H        = 1. / N;
  for (std::size_t k = 0; k < ids.size(); ++k)
    {
      auto           &patch = patches[ids[k]];
      slod_patch_info info;
      slod_patch_layout(h.get(), ids[k], &info);
      dh_fine_patch.reinit(patch.sub_tria);
      dh_fine_patch.distribute_dofs(*fe_fine); // LOD.cc:365-366
      const unsigned int        n_f = info.n_fine; // dofs of ONE vector = spacedim * nodes
      std::vector<unsigned int> lex(n_f, numbers::invalid_unsigned_int);
      const double              x0 = info.x0 * H, y0 = info.y0 * H, h_fine = H / par.n_subdivisions;
      for (const auto &cell : dh_fine_patch.active_cell_iterators())
        {
          cell->get_dof_indices(fine_dofs);
          const Point<dim> origin = cell->vertex(0); // Cartesian cell of side H
          for (unsigned int i = 0; i < fine_dofs.size(); ++i)
            {
              const unsigned int comp = fe_fine->system_to_component_index(i).first;
              const Point<dim>   p    = origin + H * unit_pts[i];
              const unsigned int ix   = (unsigned int)std::lround((p[0] - x0) / h_fine);
              const unsigned int iy   = (unsigned int)std::lround((p[1] - y0) / h_fine);
              lex[fine_dofs[i]]       = spacedim * (ix + iy * (info.nx + 1)) + comp;
            }
        }
      for (unsigned int d = 0; d < spacedim; ++d)
        {
          Vector<double> phi(n_f), psi(n_f);
          for (unsigned int i = 0; i < n_f; ++i)
            {
              Assert(lex[i] < n_f, ExcInternalError());
              phi[i] = basis[offsets[k] + uint64_t(d) * n_f + lex[i]];
              psi[i] = premult[offsets[k] + uint64_t(d) * n_f + lex[i]];
            }
          patch.basis_function.push_back(phi);               // LOD.cc:592 / 754
          patch.basis_function_premultiplied.push_back(psi); // LOD.cc:764
        }
      dh_fine_patch.clear(); // LOD.cc:766
    }
}

// the instantiations of source/LOD.cc:1470-1471
template void LOD<2, 1>::compute_basis_function_candidates();
template void LOD<2, 2>::compute_basis_function_candidates();
