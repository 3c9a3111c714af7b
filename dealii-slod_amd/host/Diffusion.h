// Mirror of reference include/Diffusion.h:7-68: the random piecewise-constant coefficient and
// the Poisson problem class.  The sub-element stiffness loop (Diffusion.h:143-204) runs in the
// HIP kernel k_assemble; only the coefficient sampling stays on the host.
#ifndef slod_host_diffusion_h
#define slod_host_diffusion_h

#include "LOD.h"

#include <cstdlib>

namespace slod
{
  template <int dim>
  class problem_parameter : public Function<dim>
  {
  private:
    const double        min_val;
    const double        max_val;
    const unsigned int  refinement;
    std::vector<double> random_values;
    unsigned int        N_cells_per_line;
    double              eta;

  public:
    problem_parameter(double min, double max, unsigned int r)
      : min_val(min)
      , max_val(max)
      , refinement(r)
    {
      N_cells_per_line     = 1u << refinement;
      eta                  = (double)1 / N_cells_per_line;
      unsigned int N_cells = 1;
      for (int d = 0; d < dim; ++d)
        N_cells *= N_cells_per_line;
      if (max_val != min_val)
        for (unsigned int i = 0; i < N_cells; ++i)
          random_values.push_back(min_val + static_cast<float>(rand()) /
                                              (static_cast<float>(RAND_MAX / (max_val - min_val))));
    }

    double value(const Point<dim> &p, const unsigned int = 0) const override
    {
      if (max_val == min_val) // constant coefficients
        return min_val;
      const unsigned int vector_cell_index =
        (int)std::floor(p(0) / eta) + N_cells_per_line * (int)std::floor(p(1) / eta);
      return random_values[vector_cell_index];
    }
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
