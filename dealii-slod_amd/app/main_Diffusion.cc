// Counterpart of reference app/main_Diffusion.cc for the basis-construction path only:
//   main_Diffusion [n_global_refinements n_subdivisions oversampling stabilize [dump.bin]]
// prints the reference's patch summary (LOD.cc:237-242) and a digest of the basis; with a
// file name it dumps, per patch, phi and psi in patch-lexicographic order (parity tests).
#include "../host/Diffusion.h"

#include <cstdio>
#include <cstdlib>

using namespace slod;

class Problem : public DiffusionProblem<2, 1>
{
public:
  using DiffusionProblem<2, 1>::DiffusionProblem;
  void dump(const char *file) const
  {
    FILE *f = std::fopen(file, "wb");
    if (!f)
      throw std::runtime_error("cannot open dump file");
    for (const auto &p : get_patches())
      {
        const std::size_t   n = p.basis_function[0].size();
        std::vector<double> lex(n);
        for (int which = 0; which < 2; ++which)
          {
            const auto &v = which ? p.basis_function_premultiplied[0] : p.basis_function[0];
            for (std::size_t i = 0; i < n; ++i)
              lex[p.dealii_to_lexicographic[i]] = v[i];
            std::fwrite(lex.data(), sizeof(double), n, f);
          }
      }
    std::fclose(f);
  }
};

int main(int argc, char **argv)
{
  try
    {
      LODParameters<2, 1> par;
      par.n_global_refinements  = argc > 1 ? std::atoi(argv[1]) : 3;
      par.n_subdivisions        = argc > 2 ? std::atoi(argv[2]) : 4;
      par.oversampling          = argc > 3 ? std::atoi(argv[3]) : 1;
      par.LOD_stabilization     = argc > 4 ? std::atoi(argv[4]) != 0 : true;
      par.constant_coefficients = false;
      std::srand(1);
      Problem problem(par, 1, 100, 3);
      problem.run();
      double s1 = 0, s2 = 0;
      for (const auto &p : problem.get_patches())
        for (std::size_t i = 0; i < p.basis_function[0].size(); ++i)
          {
            s1 += p.basis_function[0][i];
            s2 += p.basis_function_premultiplied[0][i] * p.basis_function[0][i];
          }
      std::printf("basis digest: sum phi = %.12e, sum phi.psi = %.12e\n", s1, s2);
      std::printf("basis build time: %.3f ms\n", problem.basis_build_seconds() * 1e3);
      if (argc > 5)
        problem.dump(argv[5]);
    }
  catch (std::exception &exc)
    {
      std::cerr << std::endl
                << "----------------------------------------------------" << std::endl
                << "Exception on processing: " << std::endl
                << exc.what() << std::endl
                << "Aborting!" << std::endl
                << "----------------------------------------------------" << std::endl;
      return 1; // as app/main_Diffusion.cc:23-47
    }
  return 0;
}
