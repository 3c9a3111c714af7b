// Mirror of reference include/Elasticity.h:92-113,208-209: (lambda, mu) fields of the
// vector-valued problem; rand() is consumed in the order Lambda, then Mu.
#ifndef slod_host_elasticity_h
#define slod_host_elasticity_h

#include "Diffusion.h"

namespace slod
{
  template <int dim, int spacedim = dim>
  class ElasticityProblem : public LOD<dim, spacedim>
  {
  public:
    ElasticityProblem(const LODParameters<dim, spacedim> &par, double cmin = 1, double cmax = 100,
                      unsigned int r = 6)
      : LOD<dim, spacedim>(par)
      , Lambda(cmin, cmax, r)
      , Mu(cmin, cmax, r)
    {}

  protected:
    problem_parameter<dim> Lambda; // Elasticity.h:104
    problem_parameter<dim> Mu;     // Elasticity.h:105

    void coefficients_at_quadrature_points(const unsigned int field, const std::vector<Point<dim>> &points,
                                           std::vector<double> &values) override
    {
      if (field == 0)
        Lambda.value_list(points, values); // Elasticity.h:208
      else
        Mu.value_list(points, values, 1);  // Elasticity.h:209
    }
  };
} // namespace slod
#endif
