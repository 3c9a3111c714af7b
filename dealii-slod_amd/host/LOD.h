// Host-side C++ mirror of the reference's class interface for the basis-construction path
// (reference include/LOD.h:68-262, source/LOD.cc), without deal.II: same class and member
// names, same argument meaning, same error behaviour (exceptions, as AssertThrow does in
// LODtools.h:416-438).  compute_basis_function_candidates() is the drop-in body: it calls
// the HIP library through the C-ABI of include/slod.h.  Everything the reference does
// after the basis build (assemble_global_matrix, solve, FEM comparison, VTU output:
// LOD.cc:860-1378) is out of scope and not mirrored.
#ifndef slod_host_lod_h
#define slod_host_lod_h

#include "../../include/slod.h"

#include <array>
#include <cmath>
#include <cstdint>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace slod
{
  template <int dim>
  struct Point
  {
    std::array<double, dim> x{};
    double operator()(const unsigned int i) const { return x[i]; }
    double &operator()(const unsigned int i) { return x[i]; }
  };

  // dealii::Function<dim> as far as the path uses it (Diffusion.h:40-53,154)
  template <int dim>
  class Function
  {
  public:
    virtual ~Function() = default;
    virtual double value(const Point<dim> &p, const unsigned int component = 0) const = 0;
    void value_list(const std::vector<Point<dim>> &points, std::vector<double> &values,
                    const unsigned int component = 0) const
    {
      values.resize(points.size());
      for (std::size_t i = 0; i < points.size(); ++i)
        values[i] = value(points[i], component);
    }
  };

  // What create_mesh_for_patch() leaves in Patch::sub_tria (LOD.cc:770-858), reduced to what
  // the path reads: the cell box and the boundary id of each side (0 = domain boundary,
  // 99 = SPECIAL_NUMBER, LOD.cc:7).
  struct PatchMesh
  {
    unsigned int x0 = 0, y0 = 0, mx = 0, my = 0; // coarse cells
    unsigned int nx = 0, ny = 0;                 // fine elements
    std::array<unsigned int, 4> boundary_id{{99, 99, 99, 99}}; // left, right, bottom, top
    unsigned int n_active_cells() const { return mx * my; }
  };

  // reference include/LOD.h:68-82
  template <int dim>
  class Patch
  {
  public:
    std::vector<unsigned int>        cells; // vector_cell_index, centre first (LOD.cc:151-178)
    PatchMesh                        sub_tria;
    std::vector<std::vector<double>> basis_function;               // [spacedim][n_fine], deal.II dof order
    std::vector<std::vector<double>> basis_function_premultiplied; // [spacedim][n_fine]
    std::vector<unsigned int>        dealii_to_lexicographic;      // dof renumbering of the patch
    unsigned int                     contained_patches = 0;
  };

  // reference include/LOD.h:85-157 (the members the path reads) + the two quirk switches
  template <int dim, int spacedim>
  class LODParameters
  {
  public:
    unsigned int oversampling          = 1;
    unsigned int n_subdivisions        = 2;
    unsigned int n_global_refinements  = 2;
    bool         LOD_stabilization     = false;
    bool         constant_coefficients = true; // quirk Q1 when the coefficient is not constant
    bool         projection_quirk      = false; // quirk Q2 (LODtools.h:43-67), spacedim 2 only
    int          device                = 0;
  };

  template <int dim, int spacedim>
  class LOD
  {
  public:
    explicit LOD(const LODParameters<dim, spacedim> &par);
    virtual ~LOD();
    LOD(const LOD &) = delete;

    // make_grid, make_fe, initialize_patches, create_random_problem_coefficients,
    // compute_basis_function_candidates (LOD.cc:1425-1433); stops there.
    virtual void run();

    const std::vector<Patch<dim>> &get_patches() const { return patches; }
    // wall time of the last compute_basis_function_candidates() [s]
    double basis_build_seconds() const { return last_build_seconds; }

  protected:
    void make_fe();
    void make_grid();
    void create_patches();
    void create_mesh_for_patch(Patch<dim> &current_patch);
    void initialize_patches();
    void compute_basis_function_candidates();
    virtual void create_random_problem_coefficients() {}
    // Replaces the coefficient evaluation inside the virtual assemble_stiffness
    // (Diffusion.h:154, Elasticity.h:208-209): fill `values` with field `field` (0 = alpha or
    // lambda, 1 = mu) at the quadrature points, index ((ey*NE + ex)*4 + q).
    virtual void coefficients_at_quadrature_points(const unsigned int          field,
                                                   const std::vector<Point<dim>> &points,
                                                   std::vector<double> &        values) = 0;
    // assemble_stiffness(patch_stiffness_matrix, dummy, dh_fine_patch, empty_constraints)
    // (LOD.cc:440-444) for one patch: the unconstrained 9-point block stencil
    // stencil[node][dir][a][b] computed on the GPU.
    void assemble_stiffness(const unsigned int patch_id, std::vector<double> &stencil);

    const LODParameters<dim, spacedim> &par;
    std::vector<Patch<dim>>             patches;
    std::pair<unsigned int, unsigned int> locally_owned_patches{0, 0}; // [begin, end)
    unsigned int                          this_mpi_process = 0, n_mpi_processes = 1;
    slod_handle *                         handle = nullptr;
    double                                last_build_seconds = 0.0;

    void check(const int status, const char *what) const;
  };
} // namespace slod

#endif
