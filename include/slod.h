/*
 * include/slod.h -- flat C-ABI of libslod_hip.so: the MI355X-native replacement for the
 * per-patch SLOD basis construction of camillabelponer/dealii-slod.
 *
 * The reference has no FFI; its boundary for this path is the C++ member function
 *     void LOD<dim,spacedim>::compute_basis_function_candidates()
 *         (reference include/LOD.h:175-176, source/LOD.cc:296-768)
 * reading  patches[id].cells / sub_tria, par.{n_global_refinements, n_subdivisions,
 *          oversampling, LOD_stabilization, constant_coefficients} and the problem's
 *          coefficient Function (include/Diffusion.h:68, include/Elasticity.h:111-113),
 * writing  patches[id].basis_function / basis_function_premultiplied
 *          (include/LOD.h:79-80, source/LOD.cc:592,754,764).
 * Every entry point below names the reference code it replaces.  INTEGRATION.md shows the
 * deal.II-side binding (the body a maintainer puts into source/LOD.cc).
 *
 * Conventions: plain C types only; return 0 on success, a negative slod_status otherwise
 * (never throws across the ABI; slod_last_error() gives the text -- the reference throws
 * deal.II exceptions instead, LODtools.h:416-438).  Caller owns every buffer it passes;
 * the library owns its device workspace.  A handle is thread-compatible (one thread at a
 * time), like the reference's non-re-entrant patch loop (LOD.cc:302-322).  A plan owns ONE
 * workspace and ONE status word: executes of the same plan must not overlap (enqueue them on
 * one stream, or wait for the previous one); different plans of a handle may run concurrently
 * on different streams.
 *
 * Vector layout: per patch, PATCH-LEXICOGRAPHIC node order, component-minor:
 *     dof = spacedim*(ix + iy*(nx+1)) + comp,  ix in [0,nx], iy in [0,ny], nx = n_sub*mx.
 * slod_patch_dof_permutation() maps it to the deal.II patch-local numbering that
 * Patch::basis_function uses (consumer: assemble_global_matrix, LOD.cc:931-962).
 */
#ifndef SLOD_H
#define SLOD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLOD_ABI_VERSION 5

typedef enum
{
  SLOD_OK              = 0,
  SLOD_ERR_ARGUMENT    = -1, /* bad config / null pointer / id out of range        */
  SLOD_ERR_UNSUPPORTED = -2, /* dim != 2, spacedim not in {1,2}, patch too large    */
  SLOD_ERR_DEVICE      = -3, /* HIP runtime error (no GPU, launch failure, OOM)     */
  SLOD_ERR_STATE       = -4, /* coefficient not set, plan/handle mismatch           */
  SLOD_ERR_NUMERIC     = -5  /* non-positive pivot in a patch solve                 */
} slod_status;

/* The five scalars of LODParameters the path reads (include/LOD.h:85-157) + device. */
typedef struct
{
  int32_t dim;                   /* must be 2 (reference: source/LOD.cc:1470-1471)            */
  int32_t spacedim;              /* 1 = DiffusionProblem, 2 = ElasticityProblem               */
  int32_t n_global_refinements;  /* N = 2^n coarse cells per side, patches in Morton order    */
  int32_t n_cells_per_side;      /* 0, or N for a non-2^k grid (then row-major patch order)   */
  int32_t n_subdivisions;        /* FE_Q_iso_Q1(n) (LOD.cc:87-89)                             */
  int32_t oversampling;          /* LOD.cc:156-178                                            */
  int32_t lod_stabilization;     /* 1 = SLOD branch (LOD.cc:563-564)                          */
  int32_t constant_coefficients; /* quirk Q1: re-use first full patch matrix (LOD.cc:354-362) */
  int32_t projection_quirk;      /* quirk Q2: projection_P1_P0<2,2> row parity (LODtools.h:43-67) */
  int32_t n_problems;            /* >= 1 independent coefficient realisations (ensemble)      */
  int32_t device;                /* HIP device ordinal                                        */
  int32_t reserved;
} slod_config;

/* What create_patches()/create_mesh_for_patch() (LOD.cc:122-244,770-858) and
 * fill_dofs_indices_vector() (LODtools.h:334-375) produce for one patch. */
typedef struct
{
  int32_t cx, cy;          /* centre cell                                                   */
  int32_t x0, y0, mx, my;  /* patch extent in coarse cells                                  */
  int32_t nx, ny;          /* fine elements per side                                        */
  int32_t side_domain[4];  /* left,right,bottom,top: 1 = boundary id 0, 0 = id 99           */
  int32_t n_fine;          /* N_f  = spacedim*(nx+1)*(ny+1)                                 */
  int32_t n_internal;      /* N_I                                                           */
  int32_t n_boundary;      /* N_b  (id-99 dofs)                                             */
  int32_t n_coarse;        /* N_c  = spacedim*mx*my                                         */
  int32_t is_lod;          /* branch of LOD.cc:563-564 taken for this patch                 */
} slod_patch_info;

typedef struct slod_handle slod_handle;
typedef struct slod_plan   slod_plan;

int         slod_abi_version(void);
/* text of the last error on this handle (handle may be NULL: last slod_create failure). */
const char *slod_last_error(const slod_handle *h);

/* replaces LOD::LOD + make_grid + make_fe + initialize_patches (LOD.cc:12-31,65-119,1380-1393) */
int  slod_create(const slod_config *cfg, slod_handle **out);
void slod_destroy(slod_handle *h);

/* patches per problem = N*N (LOD.cc:237-242) */
int slod_num_patches(const slod_handle *h);
/* replaces create_patches + create_mesh_for_patch + fill_dofs_indices_vector for one patch */
int slod_patch_layout(const slod_handle *h, uint32_t patch_id, slod_patch_info *info);
/* patch->cells in the reference's order, centre first (LOD.cc:151-178); entries cx + N*cy */
int slod_patch_cells(const slod_handle *h, uint32_t patch_id, uint32_t *cells, size_t capacity);
/* perm[dealii_dof] = lexicographic dof, for DoFHandler::distribute_dofs(FESystem(FE_Q_iso_Q1(n),
 * spacedim)) on the patch sub_tria (LOD.cc:365-366); see DESIGN.md for the numbering rule. */
int slod_patch_dof_permutation(const slod_handle *h, uint32_t patch_id, uint32_t *perm,
                               size_t capacity);
/* Utilities::MPI::create_evenly_distributed_partitioning (LOD.cc:116-118) */
int slod_partition(uint64_t n_total, uint32_t n_ranks, uint32_t rank, uint64_t *begin,
                   uint64_t *end);

/* Coefficient field of problem `problem` (replaces Alpha/Lambda/Mu.value_list at the
 * quadrature points, Diffusion.h:154, Elasticity.h:208-209).  field 0 = alpha or lambda,
 * 1 = mu.  layout 0: one value per fine element, [NE][NE] row-major (ex fastest);
 * layout 1: four values per element, [NE][NE][4], q = q0 + 2*q1 of QIterated(QGauss<1>(2),n)
 * (LOD.cc:91-92).  NE = N*n_subdivisions.  `on_device` != 0: data is a device pointer. */
int slod_set_coefficient(slod_handle *h, uint32_t problem, int field, const double *data,
                         int layout, size_t count, int on_device);

/* ---- the hot path ------------------------------------------------------------------ */
/* A plan fixes the list of (global) patch ids  gid = problem*num_patches + patch_id  and
 * where each patch's result goes (offsets in doubles into basis/premult; NULL = uniform
 * stride slod_plan_stride()).  Replaces the loop header LOD.cc:345-352. */
int    slod_plan_create(slod_handle *h, const uint32_t *gids, size_t n, const uint64_t *offsets,
                        slod_plan **out);
void   slod_plan_destroy(slod_plan *p);
size_t slod_plan_stride(const slod_plan *p);       /* doubles per patch with NULL offsets   */
size_t slod_plan_output_size(const slod_plan *p);  /* doubles needed in basis (and premult) */
/* Runs the loop body LOD.cc:353-767 for every patch of the plan on the GPU.  d_basis /
 * d_premult are DEVICE pointers; per patch: spacedim vectors of n_fine doubles each
 * (= Patch::basis_function[d], Patch::basis_function_premultiplied[d]).  Asynchronous on
 * `hip_stream` (a hipStream_t, NULL = the handle's own stream). */
int slod_plan_execute(slod_plan *p, double *d_basis, double *d_premult, void *hip_stream);
/* Keep HIP-event records of the next `depth` executes (default 1 = the last one only);
 * resets the record.  The events sit on the stream the kernels are launched on. */
int slod_plan_profile(slod_plan *p, int depth);
/* mean per-kernel device time over the recorded executes (HIP events around every launch of
 * slod_plan_execute; slod_plan_execute_allgather records none: SLOD_ERR_STATE after it);
 * synchronises.  ms[0] assemble, ms[1] patch solve, ms[2] selection */
int slod_plan_kernel_ms(slod_plan *p, float ms[3]);
/* Overlap of consecutive executes inside the library (depth 1 = default, 2).  All workgroups of a step
 * start together and run through the same phases in lock-step; two steps half a phase apart fill each
 * other's idle issue slots (+14..16 % patches/s at BASELINE config C2).  With depth 2 the plan owns a
 * second workspace and two internal streams; slod_plan_execute alternates between them.  Each execute
 * is ordered AFTER everything submitted to its hip_stream so far, but hip_stream is NOT ordered after
 * the execute: call slod_plan_join(plan, stream) to make a stream wait for the executes in flight
 * (asynchronous), or slod_plan_status (synchronises).  Plans that run in several workspace chunks are
 * refused (SLOD_ERR_STATE): their launches de-phase by themselves.  Reference: the serial patch loop
 * LOD.cc:345-767 has no counterpart. */
int slod_plan_set_overlap(slod_plan *p, int depth);
int slod_plan_join(slod_plan *p, void *hip_stream);
/* The k-th patch descriptor the plan's kernels LAUNCH with, read back from the device (the descriptors
 * are produced by a device kernel from the grid scalars: create_patches + create_mesh_for_patch,
 * LOD.cc:122-244,770-858).  launch_order = 0: the caller's order; 1: the balanced launch order
 * (plan_index then tells which entry of the caller's list sits at launch position k). */
int slod_plan_patch_layout(slod_plan *p, size_t k, int launch_order, slod_patch_info *info, uint32_t *plan_index);
/* numerical status of the last execute (0 or SLOD_ERR_NUMERIC); synchronises. */
int slod_plan_status(slod_plan *p);

/* Decisions the SLOD selection stage took for one (patch, component): the discontinuous part
 * of LOD.cc:656-725 (pseudo-inverse cutoff :667, "drop the smallest triplet while
 * ||d||_inf >= 0.5" :703-725).  Parity tests assert them equal to the CPU oracle's. */
typedef struct
{
  int32_t path;       /* 0 = LOD branch (LOD.cc:566-595); 1 = SLOD, proven decision-free (QR:
                         no singular value near the cutoff, ||d||_inf < 0.5); 2 = SLOD, loop
                         replayed on the singular triplets                                     */
  int32_t n_cut;      /* singular values of G = BD'^T BD' with sigma <= 1e-15 sigma_0 (:667)   */
  int32_t n_dropped;  /* triplets put back by the 0.5-loop (:703-725)                          */
  int32_t sweeps;     /* Jacobi sweeps of the replay (path 2)                                  */
  double  dinf;       /* final ||d||_inf                                                       */
  double  sigma_max, sigma_min; /* extreme singular values of G (path 2; 0 otherwise)          */
} slod_patch_diag;
/* out[k*spacedim + d] for patch k of the plan, component d, of the LAST execute; HOST buffer of
 * `capacity` entries.  Returns the number of entries written or a negative slod_status;
 * synchronises. */
int slod_plan_diagnostics(slod_plan *p, slod_patch_diag *out, size_t capacity);

/* Host-buffer convenience wrapper: plan + execute + copy back (what the deal.II adapter
 * calls).  basis/premult are HOST pointers. */
int slod_compute_basis(slod_handle *h, const uint32_t *gids, size_t n, double *basis,
                       double *premult, const uint64_t *offsets);

/* ---- consumers of (phi, psi): the global LOD system ----------------------------------
 * Reference: assemble_global_matrix (LOD.cc:860-973), solve (LOD.cc:976-1002), and the
 * fine-scale reconstruction  solution_fine = C u_H  (LOD.cc:1251).  All vectors of ONE problem
 * sit in a slab with uniform stride (slod_plan_stride(): what a plan with NULL offsets
 * writes and what the all-gather of the multi-GPU path produces): patch p at
 * d_basis[p * stride + d * n_fine(p) + dof].  Overlaps of patches are index arithmetic on
 * the patch-lexicographic layout; no deal.II numbering is involved. */
/* Upper bound of patches q whose node set meets that of one patch: (4 l + 3)^2 (the closed patches
 * of two cells share a node as soon as their centres are at most 2 l + 1 cells apart per axis). */
int slod_lod_row_capacity(const slod_handle *h);
/* Patches coupled with `patch_id` in A_LOD (LOD.cc:970-971 pattern of Tmmult), ascending ids;
 * returns their count.  HOST buffer. */
int slod_lod_pattern(const slod_handle *h, uint32_t patch_id, uint32_t *neighbours, size_t capacity);
/* Block rows of  A_LOD = C^T (A C)  (basis_matrix_transposed.Tmmult(global_stiffness_matrix,
 * premultiplied_basis_matrix), LOD.cc:970-971) for the patches rows[0..n_rows):
 *   d_values[(k * cap + j) * s * s + d * s + e] = sum_i phi_{rows[k],d}(i) psi_{q,e}(i),
 *   q = d_cols[k * cap + j]  (0xffffffff = unused slot), cap = slod_lod_row_capacity().
 * Column (q,e) of the reference matrix is spacedim * q + e (LOD.cc:942-944).  rows is a HOST
 * array; d_* are DEVICE pointers.  The call uploads the row list (a device allocation of its own) and
 * returns after hip_stream has finished the kernel: it SYNCHRONISES hip_stream. */
int slod_lod_matrix(slod_handle *h, const uint32_t *rows, size_t n_rows, const double *d_basis,
                    const double *d_premult, size_t stride, double *d_values, uint32_t *d_cols,
                    void *hip_stream);
/* system_rhs = C^T fem_rhs (basis_matrix_transposed.Tvmult, LOD.cc:982): d_out[k * s + d] =
 * sum_i phi_{rows[k],d}(i) f(i); d_fine_rhs is the fine FEM load vector on the GLOBAL fine
 * grid, [(NE+1)^2][s] lexicographic, component-minor. */
int slod_lod_rhs(slod_handle *h, const uint32_t *rows, size_t n_rows, const double *d_basis, size_t stride,
                 const double *d_fine_rhs, double *d_out, void *hip_stream);
/* Solves A_LOD u = rhs for all num_patches * s unknowns (the reference: CG + SSOR(1.2),
 * LOD.cc:990-998; here Jacobi-preconditioned CG, all on the device) from the block rows of
 * slod_lod_matrix for rows = 0 .. num_patches-1.  Returns the iteration count (>= 0) or a
 * negative slod_status; *rel_residual (HOST, may be NULL) receives ||r|| / ||rhs||.  Synchronises. */
int slod_lod_solve(slod_handle *h, const double *d_values, const uint32_t *d_cols, const double *d_rhs,
                   double *d_u, double rel_tol, int max_iterations, double *rel_residual);
/* solution_fine = C u_H (LOD.cc:1251: basis_matrix_transposed.vmult): d_fine[(ix + iy (NE+1)) s + c]
 * = sum over patches covering the node, sum_d phi_{p,d}(node, c) u[p s + d]. */
int slod_lod_reconstruct(slod_handle *h, const double *d_basis, size_t stride, const double *d_u,
                         double *d_fine, void *hip_stream);

/* ---- fine FEM reference problem (assemble_and_solve_fem_problem, LOD.cc:1004-1094) ----
 * What the reference compares the LOD solution with (compare_lod_with_fem, LOD.cc:1240-1378).
 * fem_rhs of assemble_stiffness (Diffusion.h:149-193) on the global fine grid, [(NE+1)^2][s],
 * zero on the Dirichlet nodes (all sides, LOD.cc:1021): d_f_qp = right-hand side function at the
 * quadrature points, DEVICE, [s][NE][NE][4] with the layout of slod_set_coefficient, or NULL for
 * f = (1,..,1) (the example's "fem rhs l2 norm = 0.109375", tests/Poisson_LOD_Example.output).
 * Asynchronous on hip_stream. */
int slod_fem_rhs(slod_handle *h, const double *d_f_qp, double *d_fine_rhs, void *hip_stream);
/* Fine FEM solution for the coefficient of `problem` (the reference: CG + AMG, LOD.cc:1070-1075;
 * here a matrix-free Jacobi-preconditioned CG on the 9-point stencil planes, all on the device).
 * Returns the iteration count (>= 0) or a negative slod_status; *rel_residual (HOST, may be
 * NULL) receives ||r|| / ||rhs||.  d_fine_u: DEVICE, [(NE+1)^2][s], zero on the boundary.
 * Synchronises. */
int slod_fem_solve(slod_handle *h, uint32_t problem, const double *d_fine_rhs, double *d_fine_u, double rel_tol,
                   int max_iterations, double *rel_residual);

/* ---- inputs of the path produced on the device --------------------------------------
 * create_patches + create_mesh_for_patch + fill_dofs_indices_vector (LOD.cc:122-244,
 * 770-858; LODtools.h:334-375) evaluated by a kernel, one thread per patch; out is a HOST
 * array of n entries, equal to slod_patch_layout() entry by entry (tests check both against
 * the reference golden tests/create_patch_01.output). */
int slod_device_patch_layout(slod_handle *h, const uint32_t *patch_ids, size_t n, slod_patch_info *out);
/* The reference's coefficient object problem_parameter(min, max, r) (Diffusion.h:7-54): a
 * piecewise constant on a 2^r x 2^r grid, value(p) = vals[floor(x / eta) + 2^r floor(y / eta)],
 * eta = 2^-r (:47-51), sampled ON THE DEVICE at the points of quadrature_fine of every fine
 * element (what Alpha.value_list does at Diffusion.h:154).  d_vals: DEVICE, 4^r values in
 * the reference's fill order (:30-36). */
int slod_sample_coefficient(slod_handle *h, uint32_t problem, int field, const double *d_vals, int r);

/* ---- multi-GPU exchange (north_star: RCCL all-gather over xGMI of the basis vectors) ----
 * The reference never communicates basis vectors (its MPI path is unfinished, LOD.cc:225-229,
 * 895-897); the split it prescribes is contiguous blocks of patch ids per rank (LOD.cc:116-118,
 * slod_partition).  One process per GPU; the C/C++ host exchanges the 128-byte id out of band
 * (MPI_Bcast in dealii-slod) and then needs nothing but this library: RCCL is resolved at run
 * time (no link-time dependency; a process that already runs RCCL re-uses that copy). */
typedef struct
{
  char internal[128]; /* ncclUniqueId */
} slod_comm_id;
typedef struct slod_comm slod_comm;
const char *slod_comm_last_error(const slod_comm *c); /* c may be NULL: last failed create */
int         slod_comm_unique_id(slod_comm_id *id);    /* on rank 0; ship it to the other ranks */
int         slod_comm_create(const slod_comm_id *id, int n_ranks, int rank, int device, slod_comm **out);
void        slod_comm_destroy(slod_comm *c);
/* plain ncclAllGather of `count` doubles per rank on hip_stream (uniform-stride slabs) */
int slod_comm_allgather(slod_comm *c, const double *d_send, double *d_recv, size_t count, void *hip_stream);
/* Patch range [first, first + count) of piece `piece` out of n_pieces of a rank's padded slab of
 * patches_per_rank patches (host-only index calculus of slod_plan_execute_allgather). */
int slod_gather_piece(uint64_t patches_per_rank, uint32_t n_pieces, uint32_t piece, uint64_t *first,
                      uint64_t *count);
/* Basis construction of this rank's patches and their exchange, overlapped: the plan (uniform
 * stride, at most patches_per_rank patches) is executed in n_pieces pieces on compute_stream
 * into this rank's slab  d_*_all + rank * patches_per_rank * stride;  as soon as a piece is
 * done, comm_stream exchanges that piece of EVERY rank (grouped in-place broadcasts, one per
 * root) while the next piece is computed.  On return (asynchronous) compute_stream is ordered
 * after the exchange.  Every rank must call it with the same patches_per_rank and n_pieces.
 * Error path: a rank that fails before the grouped broadcasts are issued returns non-zero WITHOUT
 * entering the collective; its peers then wait inside RCCL.  Treat a non-zero return on any rank as
 * fatal for the communicator (destroy it, or abort the job): there is no recovery inside the call. */
int slod_plan_execute_allgather(slod_plan *p, slod_comm *c, double *d_basis_all, double *d_premult_all,
                                size_t patches_per_rank, int n_pieces, void *compute_stream,
                                void *comm_stream);

/* ---- pieces exposed for parity tests ----------------------------------------------- */
/* unconstrained patch stiffness (replaces assemble_stiffness with empty constraints,
 * LOD.cc:440-444 -> Diffusion.h:111-207 / Elasticity.h:163-299) as a 9-point block stencil:
 * stencil[node][dir][a][b], dir = (dy+1)*3+(dx+1).  HOST buffer of n_nodes*9*s*s doubles. */
int slod_assemble_stiffness_for_patch(slod_handle *h, uint32_t gid, double *stencil);
/* Ainv_PT of Gauss_elimination (LOD.cc:546): HOST buffer [n_fine][n_coarse] row-major. */
int slod_patch_solution(slod_handle *h, uint32_t gid, double *X);

#ifdef __cplusplus
}
#endif
#endif /* SLOD_H */
