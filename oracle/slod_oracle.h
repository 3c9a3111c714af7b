/*
 * oracle/slod_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, fp64) of the SLOD per-patch basis construction of
 * camillabelponer/dealii-slod, i.e. LOD<dim,spacedim>::compute_basis_function_candidates()
 * (reference source/LOD.cc:296-768) and the helpers it calls.  It exists to CHECK the
 * HIP product path; nothing under dealii-slod_amd/ may include, link or call it.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Parity pinning: checked against the reference's own goldens
 *   tests/fe_q_iso_q1_01.output, tests/solve_poisson_problem_on_patch_01.output,
 *   tests/create_patch_01.output, "fem rhs l2 norm" of tests/Poisson_LOD_Example.output
 * (copied as data under tests/golden/reference/) and against an independent
 * numpy/scipy restatement (oracle/slod_numpy.py).  The SLOD-specific steps
 * (LOD.cc:598-757) and elasticity have NO reference golden: "parity unpinned by the
 * reference" for those; they are pinned oracle-vs-numpy only.
 *
 * Numbering: all per-patch vectors are in PATCH-LEXICOGRAPHIC node order,
 * component-minor: dof = spacedim*(ix + iy*(nx+1)) + comp, ix in [0,nx], iy in [0,ny],
 * nx = n_sub*mx fine elements across the patch.
 */
#ifndef SLOD_ORACLE_H
#define SLOD_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct
{
  int nref;          /* n_global_refinements: N = 2^nref coarse cells per side (LOD.cc:133-134) */
  int n_cells;       /* if > 0: N = n_cells (row-major patch order); used for non-2^k goldens   */
  int n_sub;         /* par.n_subdivisions (FE_Q_iso_Q1(n), LOD.cc:87-89)                       */
  int oversampling;  /* par.oversampling (LOD.cc:156-178)                                       */
  int spacedim;      /* 1 = Poisson (Diffusion.h), 2 = elasticity (Elasticity.h)                */
  int stabilize;     /* par.LOD_stabilization: 1 = SLOD branch (LOD.cc:563-564)                 */
  int reuse_full;    /* quirk Q1: par.constant_coefficients matrix reuse (LOD.cc:354-362)       */
  int proj_quirk;    /* quirk Q2: projection_P1_P0<2,2> row-parity components (LODtools.h:43-67)*/
} so_cfg;

typedef struct
{
  int pid;           /* patch id = active_cell_index (Morton) of the centre cell          */
  int cx, cy;        /* centre cell                                                       */
  int x0, y0;        /* lowest coarse cell of the patch                                   */
  int mx, my;        /* extent in coarse cells                                            */
  int nx, ny;        /* fine elements per side                                            */
  int side_domain[4];/* left,right,bottom,top: 1 = boundary id 0 (domain), 0 = id 99      */
  int n_f, n_i, n_b, n_c;
  int is_lod;        /* 1 if the LOD (non-stabilised) branch is taken (LOD.cc:563-564)    */
} so_patch;

typedef struct
{
  int    n_dropped[2];     /* singular triplets removed by the 0.5-loop (LOD.cc:703-725)  */
  int    n_cut[2];         /* singular values below the 1e-15 cutoff (LOD.cc:667)         */
  double dinf[2];          /* final ||d||_inf                                             */
  double sigma_max[2], sigma_min[2]; /* singular values of G = BD'^T BD'                  */
  double cond_hint;
} so_diag;

/* ---- index calculus ---------------------------------------------------- */
int  so_num_cells_per_side(const so_cfg *cfg);
int  so_num_patches(const so_cfg *cfg);
void so_patch_centre(const so_cfg *cfg, int pid, int *cx, int *cy);
int  so_patch_id_of_cell(const so_cfg *cfg, int cx, int cy);
void so_patch_init(const so_cfg *cfg, int pid, so_patch *p);
/* coarse cells of the patch in the reference's order (centre first, then x-offset outer,
 * y-offset inner: LOD.cc:151-178); entries are cx + N*cy.  Returns the count (= mx*my). */
int  so_patch_cells(const so_cfg *cfg, const so_patch *p, int *cells);

/* ---- element kernels (Diffusion.h:156-187, Elasticity.h:211-258) ------- */
/* alpha[q], q = q0 + 2*q1; K row-major 4x4 over local nodes (0,0),(1,0),(0,1),(1,1). */
void so_local_matrix_poisson(const double alpha[4], double K[16]);
/* K row-major 8x8, local dof = 2*node + comp. */
void so_local_matrix_elasticity(const double lambda[4], const double mu[4], double K[64]);
/* FE_Q_iso_Q1(n) cell matrix in deal.II hierarchic numbering for alpha == 1
 * (tests/fe_q_iso_q1_01.cc); M is (n+1)^dim square, row-major. */
void so_fe_q_iso_q1_cell_matrix(int dim, int n, double *M);
void so_lexicographic_to_hierarchic(int dim, int n, int *map);

/* ---- per-patch pipeline -------------------------------------------------- */
/* coef[f] : global per-quadrature-point field f (0 = alpha or lambda, 1 = mu),
 *           index ((ey*NE + ex)*4 + q), NE = N*n_sub fine elements per side.
 * stencil : [n_f_nodes][9][s][s], dir = (dy+1)*3 + (dx+1): coupling of node (ix,iy)
 *           with node (ix+dx, iy+dy); the UNCONSTRAINED patch matrix (LOD.cc:440-444). */
void so_assemble_patch(const so_cfg *cfg, const so_patch *p, const double *const *coef,
                       double *stencil);
/* X = A_c^{-1} PT (LOD.cc:546): X is [n_f*s][n_c] row-major, zero on every boundary dof. */
int  so_patch_solve(const so_cfg *cfg, const so_patch *p, const double *stencil, double *X);
/* generic constrained solve A_II u = f on the patch (all four sides Dirichlet 0):
 * rhs/out are [n_f*s][nrhs] row-major (boundary rows ignored / zeroed). */
int  so_solve_interior(int nx, int ny, int s, const double *stencil, const double *rhs,
                       int nrhs, double *out);
/* full pipeline for one patch: phi, psi are [s][n_f*s]. Returns 0 on success. */
int  so_patch_basis(const so_cfg *cfg, const double *const *coef, int pid,
                    double *phi, double *psi, so_diag *diag);
/* all patches ids[0..n), outputs ragged by offsets[] (in doubles, per array), OpenMP over
 * patches with nthreads (<=1: serial).  offsets may be NULL -> dense stride s*s*n_f_max. */
int  so_basis_many(const so_cfg *cfg, const double *const *coef, const int *ids, int n,
                   double *phi, double *psi, const long long *offsets, int nthreads);

/* intermediate results for tests: M (n_c x n_c, before inversion), D = M^-1,
 * BD (n_b x n_c), boundary dof list (n_b). Any pointer may be NULL. */
int  so_patch_debug(const so_cfg *cfg, const double *const *coef, int pid,
                    double *M, double *D, double *BD, int *bdofs, double *X);

/* dense un-zeroed P^T [n_f][n_c] (LOD.cc:471-496) */
void so_patch_pt(const so_cfg *cfg, int pid, double *PT);

/* 0 (default): one-sided Jacobi SVD of BD'; 1: literal Gram matrix G = BD'^T BD' +
 * Jacobi eigen-solver (LOD.cc:660-667 with dgesdd replaced). */
void so_set_svd_mode(int mode);
/* conditioning probe for the parity tests: relative noise eps on X before the selection stage
 * (0 = off); see slod_oracle.c */
void so_set_solver_noise(double eps, unsigned long long seed);

/* ---- synthetic coefficient fields ---------------------------------------- */
/* splitmix64 stream, one value per fine ELEMENT (row-major, ex fastest), broadcast to
 * its 4 quadrature points.  dist 0: uniform [lo,hi]; dist 1: log-uniform lo*(hi/lo)^u. */
void so_fill_coefficient(unsigned long long seed, int dist, double lo, double hi,
                         int n_elems_per_side, double *coef_qp);
/* the reference's problem_parameter (Diffusion.h:19-51): 2^r x 2^r piecewise constants
 * drawn with glibc rand() (caller seeds with srand), sampled at the quadrature points. */
void so_fill_coefficient_rand(double lo, double hi, int r, int n_elems_per_side,
                              double *coef_qp);

#ifdef __cplusplus
}
#endif
#endif
