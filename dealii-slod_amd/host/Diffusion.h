// Mirror of reference include/Diffusion.h:7-68: the random piecewise-constant coefficient and
// the Poisson problem class.  The sub-element stiffness loop (Diffusion.h:143-204) runs in the
// HIP kernel k_assemble; only the coefficient sampling stays on the host.
#ifndef slod_host_diffusion_h
#define slod_host_diffusion_h

#include "LOD.h"

#include <cstdlib>

namespace slod
{
  // Piecewise-constant field on a uniform 2^r x 2^r grid of the unit square, one draw of the C library
  // generator per grid cell in row order (x fastest).  What must agree with the reference bit for
  // bit is the draw itself (Diffusion.h:30-36): lo + float(rand()) / float(RAND_MAX / (hi - lo));
  // slod_sample_coefficient evaluates the same table on the device.
  template <int dim>
  class problem_parameter : public Function<dim>
  {
    static_assert(dim == 2, "the reference samples x and y only (Diffusion.h:47-51)");

  public:
    problem_parameter(double lo, double hi, unsigned int r)
      : cells_per_side(1u << r)
      , constant(lo == hi)
      , constant_value(lo)
    {
      if (constant)
        return;
      table.resize((size_t)cells_per_side * cells_per_side);
      const float inv_range = static_cast<float>(RAND_MAX / (hi - lo));
      for (double &t : table)
        t = lo + static_cast<float>(rand()) / inv_range;
    }

    double value(const Point<dim> &p, const unsigned int = 0) const override
    {
      if (constant)
        return constant_value;
      const double   cells = (double)cells_per_side; // eta = 1 / cells; the reference divides by eta
      const double   eta = 1.0 / cells;
      const unsigned cx = (unsigned)(int)std::floor(p(0) / eta), cy = (unsigned)(int)std::floor(p(1) / eta);
      return table[cx + (size_t)cells_per_side * cy];
    }

    unsigned int grid_cells_per_side() const { return cells_per_side; }

  private:
    unsigned int        cells_per_side;
    bool                constant;
    double              constant_value;
    std::vector<double> table;
  };

  template <int dim, int spacedim>
  class DiffusionProblem : public LOD<dim, spacedim>
  {
  public:
    DiffusionProblem(const LODParameters<dim, spacedim> &par, double amin = 1, double amax = 100,
                     unsigned int r = 8)
      : LOD<dim, spacedim>(par)
      , Alpha(amin, amax, r)
    {}

  protected:
    problem_parameter<dim> Alpha; // reference: Alpha(1, 100, 8), Diffusion.h:62

    void coefficients_at_quadrature_points(const unsigned int, const std::vector<Point<dim>> &points,
                                           std::vector<double> &values) override
    {
      Alpha.value_list(points, values); // Diffusion.h:154
    }
  };
} // namespace slod
#endif
