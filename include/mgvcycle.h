/*
 * mgvcycle.h - C ABI of libmgvcycle.so, the MI355X (gfx950) multigrid-cycle library.
 *
 * Drop-in boundary for the hot path of JuliaInv/Multigrid.jl (reference @ v0.8.0):
 *   recursiveCycle  src/Multigrid/MGcycle.jl:1-118      -> mg_cycle_*
 *   relax           src/Multigrid/MGcycle.jl:122-136    -> fused inside mg_cycle_*
 *   solveCoarsest   src/Multigrid/MGcycle.jl:138-181    -> fused inside mg_cycle_* (default branch l.177)
 *   solveMG         src/Multigrid/SolveFuncs.jl:3-39    -> mg_solve_*
 *   SpMatMul        src/Multigrid/SpMatMul.jl:4-26      -> mg_spmv_*
 *   addVectors      src/Multigrid/SpMatMul.jl:29-37     -> fused into the SpMV epilogues
 *   MGparam         src/Multigrid/MGdef.jl:91-116       -> mg_hierarchy (opaque) + mg_set_*
 *   adjustMemoryForNumRHS  src/Multigrid/MGsetup.jl:166-223 -> mg_set_nrhs
 *   replaceMatrixInHierarchy (numeric part) MGsetup.jl:226-270 -> mg_replace_values_FP64
 *   clear!/destroyCoarsestLU  MGdef.jl:179-206          -> mg_destroy
 *
 * Calling convention follows the reference's own ccall idiom (src/Multigrid/parRelax.jl:61-64,
 * deps/src/parRelax.h:7-43): sparse operators are passed exactly as Julia's SparseMatrixCSC holds
 * them - colptr/rowval as 1-based Int64, nzval as Float64 - and the library subtracts 1; scalars are
 * long long / double; symbols carry the _FP64_INT64 suffix.  Unlike the reference's void functions,
 * every entry point returns an int status (MG_OK == 0) and mg_last_error() returns the message.
 *
 * The reference stores every operator TRANSPOSED (MGdef.jl:75-77), so the CSC arrays of AT are the
 * CSR arrays of A: colptr = row pointers, rowval = column indices.  "n_rows" below is therefore
 * length(colptr)-1 = size(AT,2) and "n_cols" is size(AT,1).
 *
 * Ownership: the caller owns every host array; the library copies the hierarchy to HBM inside
 * mg_set_* / mg_finalize and retains no host pointer.  b is read-only, x is in/out in the caller's
 * buffer (the in-place contract test/Multigrid/testGMG.jl:54-55 asserts).  A handle is not
 * thread-safe (neither is MGparam: the F-cycle mutates it, MGcycle.jl:82-84).
 */
#ifndef MGVCYCLE_H
#define MGVCYCLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mg_hierarchy mg_hierarchy;

/* status codes */
#define MG_OK 0
#define MG_ERR_INVALID 1   /* bad argument / inconsistent hierarchy */
#define MG_ERR_HIP 2       /* HIP runtime error (no device, out of memory, launch failure) */
#define MG_ERR_STATE 3     /* call out of order (e.g. cycle before finalize) */
#define MG_ERR_UNSUPPORTED 4

/* operator selector for mg_set_operator / mg_spmv / mg_replace_values */
#define MG_OP_A 0 /* As[level]  : n_l x n_l                                  */
#define MG_OP_P 1 /* Ps[level]  : CSR of P, n_l rows  x n_{l+1} cols         */
#define MG_OP_R 2 /* Rs[level]  : CSR of R, n_{l+1} rows x n_l cols          */

/* kernel selector for mg_time_op_dev_FP64 / mg_profile_get */
#define MG_K_SPMV 0     /* y = alpha*A*x + beta*y          (SpMatMul.jl:4-13)          */
#define MG_K_RESIDUAL 1 /* r = b - A*x                     (MGcycle.jl:58-60 fused)    */
#define MG_K_SMOOTH 2   /* x' = x + d.*(b - A*x)           (MGcycle.jl:128-131 fused)  */
#define MG_K_RESTRICT 3 /* bc = R*r                        (MGcycle.jl:66)             */
#define MG_K_PROLONG 4  /* x += P*xc                       (MGcycle.jl:90)             */
#define MG_K_DSCALE 5   /* x = d.*b (first sweep from x=0) (MGcycle.jl:134)            */
#define MG_K_COARSE 6   /* xc = LU \ bc                    (MGcycle.jl:177)            */
#define MG_K_NORM 7     /* ||r||^2                         (SolveFuncs.jl:30)          */
#define MG_K_SMOOTH_PROLONG 8 /* small levels (round 6): x + P*xc formed in LDS inside the first post-smoothing sweep's launch (grid27_small_prolong_smooth; MGcycle.jl:90-102); profile slot only.  (The fine-level form of rounds 2-5 - the prolongation inside the marching sweep's staging - was slower in every round and is retired.)  A small level's residual + restriction as one launch (grid27_small_resid_restrict) is counted under MG_K_RESTRICT */
#define MG_K_SMOOTH_RESIDUAL 9 /* t = x + d.*(b - A*x) and r = b - A*t in one pass (MGcycle.jl:129-131 + 58-60 / SolveFuncs.jl:26-27) */
#define MG_K_SMOOTH_RESIDUAL_NORM 10 /* the same pass in the solve loop: last post-smoothing sweep + the stopping test's residual: ||r||^2 and t + d.*r out (SolveFuncs.jl:26-30); profile slot only */
#define MG_K_FOUR_STAGE 11 /* solve loop, fine level: the last post-smoothing sweep + stopping-test residual of step k AND the second pre-smoothing sweep + residual of step k+1 in one pass (SolveFuncs.jl:24-37 around MGcycle.jl:26-31,54-60): x, b in; t', r' and ||r||^2 out; profile slot only */
#define MG_K_GHOST 12 /* ghost-layer form of the sharded cycle: pack + (host-staged / dry transport) unpack kernels of one exchange on the compute stream; profile slot only */
#define MG_K_COUNT 13

/* ---- lifecycle ---------------------------------------------------------------------------- */

/* Allocate an empty hierarchy of `nlevels` levels (length(param.As)) on HIP device `device_id`,
 * sized for `nrhs` right-hand sides.  Levels are 1-based in every call below, as in the reference. */
int mg_create(long long nlevels, long long nrhs, long long device_id, mg_hierarchy** out);

/* Upload one operator from Julia's CSC-of-the-transpose arrays (1-based Int64). */
int mg_set_operator_FP64_INT64(mg_hierarchy* h, long long level, long long which,
                               long long n_rows, long long n_cols,
                               const long long* colptr, const long long* rowval, const double* nzval);

/* relaxPrecs[level] (length n_level) and relaxPre(level)/relaxPost(level) evaluated by the host
 * (they are Julia functions of the level, MGdef.jl:98-99). */
int mg_set_relax_FP64(mg_hierarchy* h, long long level, const double* d, long long n,
                      long long relaxPre, long long relaxPost);

/* Launch-bound coarse sub-cycles (from the first level of at most `graph_max_rows` rows*nrhs down, option
 * "graph_max_rows", default 300000; "no_graph" switches it off) are captured once into HIP graphs and replayed: the
 * reference has no counterpart (its recursion is host code, MGcycle.jl:1-118).  Counters for tests and tuning:
 * *launches = graph replays so far, *graphs = graphs currently cached.  Graphs are dropped by every call that
 * changes the hierarchy. */
int mg_graph_launches(mg_hierarchy* h, long long* launches, long long* graphs);

/* Optional performance hint (results are unchanged): the rows of As[level] are the nodes of an x-fastest
 * n1 x n2 x n3 regular grid (param.Meshes[level].n .+ 1 for geometric multigrid, MGsetup.jl:54).  Lets the
 * library walk the row blocks in L2-sized y-tiles when three grid planes of the gathered vector do not fit
 * an XCD's L2 (block right-hand sides, grids beyond ~400^2 nodes per plane).  n3 = 1 for 2-D. */
int mg_set_grid_hint(mg_hierarchy* h, long long level, long long n1, long long n2, long long n3);

/* Override one format-selection switch of THIS handle (DESIGN.md section 3 lists them; key = the environment name
 * without the MG_ prefix, lower case: "no_rowclass", "no_tile", "tile_min_wg", "nt", ...).  The environment itself is
 * read once, inside mg_create; nothing on the launch path reads it.  Applies to operators uploaded after the call
 * (call it right after mg_create), e.g. mg_set_option(h, "no_rowclass", 1) forces the streaming CSR formats. */
int mg_set_option(mg_hierarchy* h, const char* key, double value);

/* relaxType: 0 = pointwise relaxPrecs ("Jac", "SPAI": relax(), MGcycle.jl:122-136);
 * 1 = "Jac-GMRES": FGMRES_relaxation with npre/npost inner directions, preconditioned by relaxPrecs
 * (FGMRES.jl:48-126, MGcycle.jl:35-38,48-50,96-98).  Call before mg_finalize. */
int mg_set_relax_type(mg_hierarchy* h, long long relaxType);

/* cycleType: 'V', 'W', 'F' (MGcycle.jl:78-85) or 'K' (2-step FGMRES recursion, MGcycle.jl:72-76). */
int mg_set_cycle_type(mg_hierarchy* h, long long cycleType);

/* Coarsest solve, default branch `z = param.LU\b` (MGcycle.jl:177): the host factors the coarsest
 * operator (UMFPACK in the reference, MGsetup.jl:350) and hands over the explicit inverse,
 * column-major n x n, applied on device as a dense product. */
int mg_set_coarse_dense_inverse_FP64(mg_hierarchy* h, long long n, const double* Ainv_colmajor);

/* The same solve from SPARSE factors, for coarsest levels too large for an explicit inverse: the layout of the
 * reference's native applier (deps/src/parLU.cpp:120-190 / setupLUFactor, parallelJuliaSolver.jl:113-148): CSR L
 * with the diagonal LAST in every row, CSR U with the diagonal FIRST, 1-based Int64 pointers/indices, p and q with
 * A[p,q] = L*U, so that x[q] = U \ (L \ b[p]).  Applied by one workgroup walking the dependency levels. */
int mg_set_coarse_lu_FP64_INT64(mg_hierarchy* h, long long n, const long long* Lptr, const long long* Lcol,
                                const double* Lval, const long long* Uptr, const long long* Ucol,
                                const double* Uval, const long long* p, const long long* q);

/* coarseSolveType "GMRES" (MGcycle.jl:152-168): the coarsest level is solved by one restart of Jacobi-preconditioned
 * FGMRES(10) with tol 0.01 from x = 0; d = relaxParam ./ diag(A_c) is what defineCoarsestAinv keeps in param.LU
 * (MGsetup.jl:334).  A block of right-hand sides takes the reference's blockFGMRES branch (MGcycle.jl:164-166): block
 * flexible GMRES(10), one restart, Frobenius residual estimate <= 1e-2, same preconditioner per column. */
int mg_set_coarse_gmres_FP64(mg_hierarchy* h, long long n, const double* d);

/* Validate the hierarchy (shapes chain, every level complete), build the row-block partitions,
 * allocate the per-level b/r/x scratch (CYCLEmem, MGdef.jl:56-60). */
int mg_finalize(mg_hierarchy* h);

/* adjustMemoryForNumRHS: re-size the scratch when the number of RHS columns changes. */
int mg_set_nrhs(mg_hierarchy* h, long long nrhs);

/* Replace the numerical values of an operator whose sparsity is unchanged (nnz must match). */
int mg_replace_values_FP64(mg_hierarchy* h, long long level, long long which,
                           const double* nzval, long long nnz);

/* replaceMatrixInHierarchy on the device (MGsetup.jl:226-270): new fine values on the unchanged sparsity, then
 * per level relaxPrecs[l] (relaxKind 0: Jac omega/diag, 1: SPAI omega*diag/colnorm^2; omega[l] per level) and the
 * Galerkin product As[l+1] = Ps[l]*As[l]*Rs[l] on the fixed patterns (rows of As[l+1] of any length: 2048 target columns
 * at a time; deterministic - the same sums in the same order on every run).  The coarsest factorisation stays on the host: fetch the
 * coarsest values (mg_get_values_FP64), factor, mg_set_coarse_dense_inverse_FP64, mg_finalize. */
int mg_rap_FP64(mg_hierarchy* h, const double* fine_nzval, long long nnz, long long relaxKind,
                const double* omega, long long* levels_done);
int mg_get_values_FP64(mg_hierarchy* h, long long level, long long which, double* out, long long nnz);
int mg_get_relax_FP64(mg_hierarchy* h, long long level, double* out, long long n);

int mg_destroy(mg_hierarchy* h);

/* ---- the hot path, host buffers (what the Julia glue ccalls) -------------------------------- */

/* One cycle x <- recursiveCycle(param,b,x,1).  b, x: column-major n x nrhs (Julia Array{Float64}).
 * x_is_zero: 1 = caller guarantees x==0 (preconditioner closure, SolveFuncs.jl:59), 0 = x!=0,
 * -1 = decide like the reference does with norm(x)>0 (MGcycle.jl:29). */
int mg_cycle_FP64(mg_hierarchy* h, const double* b, double* x, long long n, long long nrhs,
                  long long x_is_zero);

/* solveMG: cycles until ||b-Ax||_F/||r0||_F < tol or maxIter (SolveFuncs.jl:14-37).
 * resvec (length maxIter+1, may be NULL) receives r0 and the residual norm after every cycle. */
int mg_solve_FP64(mg_hierarchy* h, const double* b, double* x, long long n, long long nrhs,
                  double tol, long long maxIter, long long* iters, double* resvec);

/* solveCG_MG (SolveFuncs.jl:104-116): KrylovMethods.cg (v0.6.0, external) preconditioned with one cycle
 * from x = 0 (getMultigridPreconditioner, SolveFuncs.jl:59), vectors resident on device across iterations.
 * nrhs = 1 (blockCG is not on the device path).  resvec (length maxIter) receives ||r||/||b|| per iteration;
 * flag: 0 converged, -1 maxIter reached, -2 breakdown (alpha = Inf or < 0), -9 b = 0. */
int mg_pcg_FP64(mg_hierarchy* h, const double* b, double* x, long long n, double tol,
                long long maxIter, long long* iters, long long* flag, double* resvec);
int mg_pcg_dev_FP64(mg_hierarchy* h, const double* b_dev, double* x_dev, long long n, double tol,
                    long long maxIter, long long* iters, long long* flag, double* resvec);

/* solveBiCGSTAB_MG (SolveFuncs.jl:87-101): KrylovMethods.bicgstb (external) with M1 = one cycle from x = 0,
 * M2 = identity.  resvec (length 2*maxIter+1) receives ||r0||/||b|| and two entries per iteration; *nres their
 * number.  flag: 0 converged, -1 maxIter, -2 breakdown, -3 converged on the half step, -9 b = 0.  nrhs = 1. */
int mg_bicgstab_FP64(mg_hierarchy* h, const double* b, double* x, long long n, double tol,
                     long long maxIter, long long* iters, long long* flag, double* resvec,
                     long long* nres);
int mg_bicgstab_dev_FP64(mg_hierarchy* h, const double* b_dev, double* x_dev, long long n, double tol,
                         long long maxIter, long long* iters, long long* flag, double* resvec,
                         long long* nres);

/* solveGMRES_MG (SolveFuncs.jl:119-133): KrylovMethods.fgmres (external), flexible restarted GMRES(inner) with one
 * cycle from x = 0 as preconditioner.  maxIter counts restarts; resvec (length inner*maxIter) receives the residual
 * estimate after every inner step, *nres their number, *iters the total number of inner steps.  nrhs = 1. */
int mg_fgmres_FP64(mg_hierarchy* h, const double* b, double* x, long long n, long long inner, double tol,
                   long long maxIter, long long* iters, long long* flag, double* resvec, long long* nres);
/* the same with b and x in HBM (device pointers), like mg_pcg_dev_FP64 / mg_bicgstab_dev_FP64 */
int mg_fgmres_dev_FP64(mg_hierarchy* h, const double* b_dev, double* x_dev, long long n, long long inner, double tol,
                       long long maxIter, long long* iters, long long* flag, double* resvec, long long* nres);

/* The block branches of the same three drivers (size(b,2) > 1: KrylovMethods.blockCG / blockBiCGSTB / blockFGMRES,
 * SolveFuncs.jl:95,113,130), nrhs <= 16, every n x nrhs block resident in HBM across iterations.  The package is not
 * vendored: the published algorithms are restated in oracle/mg_oracle.py (O'Leary's block CG with a pseudo-inverse of
 * P'AP; El Guennouni-Jbilou-Sadok block BiCGSTAB, M1 = the cycle; block flexible GMRES with block modified Gram-Schmidt,
 * blocks orthonormalised by Cholesky QR applied twice).  Stopping: max over columns of ||r_j||/||b_j|| <= tol (CG,
 * BiCGSTAB; resmat is maxIter x nrhs row-major, resvec as for the single-vector driver), Frobenius norm for FGMRES.
 * Flags as for the single-vector drivers.  b / x: column-major host blocks, or row-major [n][nrhs] device blocks (_dev). */
int mg_block_pcg_FP64(mg_hierarchy* h, const double* b, double* x, long long n, long long nrhs, double tol,
                      long long maxIter, long long* iters, long long* flag, double* resmat);
int mg_block_pcg_dev_FP64(mg_hierarchy* h, const double* b_dev, double* x_dev, long long n, long long nrhs, double tol,
                          long long maxIter, long long* iters, long long* flag, double* resmat);
int mg_block_bicgstab_FP64(mg_hierarchy* h, const double* b, double* x, long long n, long long nrhs, double tol,
                           long long maxIter, long long* iters, long long* flag, double* resvec, long long* nres);
int mg_block_bicgstab_dev_FP64(mg_hierarchy* h, const double* b_dev, double* x_dev, long long n, long long nrhs,
                               double tol, long long maxIter, long long* iters, long long* flag, double* resvec,
                               long long* nres);
int mg_block_fgmres_FP64(mg_hierarchy* h, const double* b, double* x, long long n, long long nrhs, long long inner,
                         double tol, long long maxIter, long long* iters, long long* flag, double* resvec,
                         long long* nres);
int mg_block_fgmres_dev_FP64(mg_hierarchy* h, const double* b_dev, double* x_dev, long long n, long long nrhs,
                             long long inner, double tol, long long maxIter, long long* iters, long long* flag,
                             double* resvec, long long* nres);

/* Mixed-precision preconditioner hook of getMultigridPreconditioner (SolveFuncs.jl:52-58): a Float32 block against the
 * Float64 hierarchy - bl .= b; z .= 0; recursiveCycle(param,bl,z,1); z2 .= z.  Column-major n x nrhs host blocks. */
int mg_cycle_mixed_FP32(mg_hierarchy* h, const float* b32, float* z32, long long n, long long nrhs);

/* Page-lock / release a long-lived host array (the preconditioner's z = param.memCycle[1].x, a solver's b and x) so the
 * host-pointer entry points above copy it at full PCIe rate.  The caller owns the lifetime: unregister before the
 * array is freed or resized (hipHostRegister / hipHostUnregister). */
int mg_host_register(void* ptr, long long bytes);
int mg_host_unregister(void* ptr);

/* target = beta*target + alpha*Op*x on one level (SpMatMul.jl:4-13); column-major host blocks. */
int mg_spmv_FP64(mg_hierarchy* h, long long level, long long which, double alpha, const double* x,
                 double beta, double* y, long long nrhs);

/* ---- the hot path, device-resident buffers -------------------------------------------------- */
/* Same semantics with b/x already in HBM (hipMalloc'd or torch tensors' data_ptr).  Blocks with
 * nrhs>1 use the library's device layout: row-major [n][nrhs].  Work is enqueued on the library's
 * stream; the call returns after the stream has drained (synchronous from the host's view).
 * The library's stream is non-blocking: it does NOT order against the caller's streams.  Whatever the caller still has in
 * flight for these buffers (a fill, a copy, the kernel that produced b) must be complete on entry - synchronise the
 * producing stream first (the Python binding does: device.py::_sync_torch), or hand the library that stream (mg_set_stream). */
int mg_cycle_dev_FP64(mg_hierarchy* h, const double* b_dev, double* x_dev, long long n,
                      long long nrhs, long long x_is_zero);
int mg_solve_dev_FP64(mg_hierarchy* h, const double* b_dev, double* x_dev, long long n,
                      long long nrhs, double tol, long long maxIter, long long* iters,
                      double* resvec);
int mg_spmv_dev_FP64(mg_hierarchy* h, long long level, long long which, double alpha,
                     const double* x_dev, double beta, double* y_dev, long long nrhs);
/* Fused forms used inside the cycle, exposed for kernel-level parity tests:
 *   MG_K_RESIDUAL: out = b - A*x ;  MG_K_SMOOTH: out = x + d.*(b - A*x)   (out must not alias x). */
int mg_fused_dev_FP64(mg_hierarchy* h, long long level, long long kernel, const double* b_dev,
                      const double* x_dev, double* out_dev, long long nrhs);
/* Two stages in one pass over level `level` (1-based, not the coarsest): t = x + d.*(b - A*x)  (relax's sweep,
 * MGcycle.jl:129-131) and r = b - A*t (MGcycle.jl:58-60 / SolveFuncs.jl:26-27); optionally xn = t + d.*r (the next
 * cycle's first update) and *norm_r = ||r|| (SolveFuncs.jl:30).  r_dev, xn_dev, norm_r may be NULL.  x, t, r, xn must be
 * four different 16-byte aligned device buffers.  MG_ERR_UNSUPPORTED when the level is not served by the two-stage
 * marching kernel (the cycle then runs the two launches). */
int mg_sweep_residual_dev_FP64(mg_hierarchy* h, long long level, const double* b_dev, const double* x_dev,
                               double* t_dev, double* r_dev, double* xn_dev, double* norm_r);

/* The solve loop's two fine-level passes across the stopping test as ONE four-stage pass (round 4; replaces the pair
 * SolveFuncs.jl:24-37 runs back to back: the last post-smoothing sweep + `r = b - A x`, `norm(r)` of step k, then
 * MGcycle.jl:26-31,54-60 of step k+1 - `r[:] = b - A x`, two sweeps of relax, `r = b - A x` for the restriction):
 *   t = x + d.*(b - A x) ; r = b - A t ; *norm_r = ||r|| ; xn = t + d.*r ; tp = xn + d.*(b - A xn) ; rp = b - A tp.
 * x, tp, rp: three different device buffers (x 16-byte aligned); t is not stored.  MG_ERR_UNSUPPORTED when the level is not
 * served (needs the 2-D tile form of the two-stage pass, relaxPre = 2, one right-hand side, a pointwise smoother).
 * mg_solve*_FP64 uses it on every step but the last by count; a stopping test that ends the loop earlier re-creates the
 * iterate with one sweep of the pass's input.  Exposed for kernel tests. */
int mg_four_stage_dev_FP64(mg_hierarchy* h, long long level, const double* b_dev, const double* x_dev, double* tp_dev,
                           double* rp_dev, double* norm_r);
/* transposeHierarchy (MGsetup.jl:274-318) on the resident hierarchy: As[l] <- As[l]' (every level), Ps[l] <- Rs[l]' (Rs[l] keeps its
 * values: the reference's second assignment reads the new Ps[l]), relaxPrecs unchanged, dense coarsest inverse transposed in place.
 * The transposes are computed in HBM (counting sort + per-column order: the stored-order CSR of the transpose), the device formats
 * rebuilt from them; mg_finalize is called.  MG_ERR_UNSUPPORTED, nothing changed, when the coarsest solve is held as sparse
 * factors or a column has more than 4096 entries (both checked before the first operator is replaced): hand the transposed
 * operators over with mg_set_operator_* instead.  Any OTHER error (MG_ERR_HIP: an allocation or a launch failed half way) leaves
 * the handle unfinalized - every entry point refuses it - and the hierarchy must be uploaded again. */
int mg_transpose_hierarchy(mg_hierarchy* h);
/* shape[3] = rows, columns, stored entries of operator `which` of `level` as the device holds it (after a transpose: of the transposed one). */
int mg_operator_shape(mg_hierarchy* h, long long level, long long which, long long* shape);
/* *yes = 1 when level `level` has the four-stage form; geometry[12] as mg_sweep_residual_form's tile geometry. */
int mg_four_stage_form(mg_hierarchy* h, long long level, long long* yes, long long* geometry);

/* ---- measurement ---------------------------------------------------------------------------- */
/* Launch kernel `kernel` of `level` `reps` times back to back on the library's stream between two
 * HIP events and return the average duration in ms; *bytes receives the ALGORITHMIC bytes of one
 * launch. */
int mg_time_op_dev_FP64(mg_hierarchy* h, long long level, long long kernel, long long nrhs,
                        long long reps, double* ms_avg, double* bytes);
/* Per-kernel HIP-event accounting of everything the cycle launches (off by default). */
int mg_profile_enable(mg_hierarchy* h, long long on);
int mg_profile_get(mg_hierarchy* h, long long level, long long kernel, double* total_ms,
                   long long* launches, double* bytes_per_launch);
int mg_profile_reset(mg_hierarchy* h);
/* Bytes one launch of the kernel IN USE has to move: the device format's matrix side (row classes: 2 or 6 B/row +
 * dictionary; pattern-coded: 8 B/nnz + descriptors; plain CSR: 12 B/nnz + row pointers) + every vector element once.
 * This - not the CSR-priced figure of mg_profile_get - is what a bandwidth fraction must be computed from. */
int mg_profile_get_moved(mg_hierarchy* h, long long level, long long kernel, double* moved_bytes_per_launch);
/* Device format of one operator: number of distinct row patterns (0 = plain CSR with int32 column indices),
 * dictionary length, and the index-side bytes (row pointers + column information) of one nrhs=1 launch. */
int mg_operator_format(mg_hierarchy* h, long long level, long long which, long long* npatterns,
                       long long* dict_entries, double* index_bytes_per_launch);
/* Row-class form of one operator (csr_rowclass_spmv): number of distinct rows (same column offsets from the row's
 * first column AND bit-identical values; 0 = the operator is not stored that way), dictionary length, and the
 * matrix-side bytes one nrhs=1 launch of the kernel in use actually streams (row classes: 6 B/row + dictionary;
 * pattern-coded: 8 B/nnz + descriptors; plain CSR: 12 B/nnz + row pointers).  Lossless; chosen at upload. */
int mg_operator_rowclasses(mg_hierarchy* h, long long level, long long which, long long* nclasses,
                           long long* dict_entries, double* matrix_bytes_per_launch);
/* Two refinements of the row-class form: implicit_first = 1 when the class also fixes (first column - row index), so
 * no per-row first column is stored (2 B/row instead of 6); class_relax = 1 when the level's relaxPrec is constant per
 * class and the fused sweep reads it from the dictionary instead of streaming 8 B/row; kernel_variant = which kernel
 * serves the operator at nrhs == 1: -1 none (streaming formats), 0 csr_rowclass_spmv, 1 csr_rowclass_window_spmv
 * (x staged in LDS per workgroup of consecutive rows), 2 csr_rowclass_tile_spmv (plane tiles from the grid hint),
 * 3 csr_rowclass_march_spmv (z-marching ring of slabs from the grid hint), 4 csr_rowclass_lane_spmv (every lane
 * walks its own row's class, dictionary in LDS: the default for operators that are not staged), 7 csr_rowclass_marchr_spmv
 * (a restriction of a vertex-centred grid pair walking the fine planes: mg_marchr.hpp), 8 the one-trip kernels of the small grid
 * levels (mg_small.hpp), 9 band-27 (planar values of a variable-coefficient 27-point grid operator), 10 grid_cell_prolong (the
 * trilinear prolongation of a vertex-centred grid pair, a lane per coarse cell), 11 grid_wave_restrict (its restriction, 62 coarse
 * nodes per wavefront) - 10 and 11 for grid pairs of any size whose P / R the upload verified entry by entry; with a block of
 * right-hand sides (current nrhs > 1): 5 csr_rowclass_lane_spmm (one column per lane), 6 csr_rowclass_lane_spmm2 (even
 * nrhs: two columns = 16 bytes per lane; its residual form also writes the ||r||^2 partials and x + d.*r), -1 the
 * CSR-stream SpMM;
 * exception_rows = rows outside the dictionary classes (computed from the CSR arrays: in the last workgroup of the
 * row-class kernel when there are at most 256 of them - second template argument `true` - else by csr_rows_spmv). */
int mg_operator_rowclass_flags(mg_hierarchy* h, long long level, long long which, long long* implicit_first,
                               long long* class_relax, long long* kernel_variant, long long* exception_rows);
/* Algorithmic HBM bytes of one full cycle (x0 = 0) with the current nrhs, DESIGN.md section 5. */
int mg_cycle_bytes(mg_hierarchy* h, double* bytes);
/* HBM bytes held by the hierarchy. */
int mg_device_bytes(mg_hierarchy* h, double* bytes);

/* ---- building blocks of the multi-GPU cycle ---------------------------------------------------- */
/* One operator resident on its own (a rank's local rows of A, P or R with halo columns appended:
 * rectangular), applied asynchronously on the caller's HIP stream (hipStream_t passed as void*).
 * kernel: MG_K_SPMV/RESTRICT/PROLONG -> y = alpha*M*x + beta*y ; MG_K_RESIDUAL -> y = b - M*x ;
 * MG_K_SMOOTH -> y = x + d.*(b - M*x).  Blocks are row-major [n][nrhs]. */
typedef struct mg_operator mg_operator;
int mg_op_create_FP64_INT64(long long device_id, long long n_rows, long long n_cols,
                            const long long* colptr, const long long* rowval, const double* nzval,
                            mg_operator** out);
/* A rank's local A of a sharded level in BOX form: square [owned box in natural x-fastest order | halo] with empty halo
 * rows, n1*n2*n3 == regular_cols = the owned box.  Rows that read a halo column are kept apart as exception rows; all
 * others keep the row-class form of the global grid operator and run the staged kernels of the single-GPU path. */
int mg_op_create_box_FP64_INT64(long long device_id, long long n_rows, long long n_cols, const long long* colptr,
                                const long long* rowval, const double* nzval, long long n1, long long n2, long long n3,
                                long long regular_cols, mg_operator** out);
/* A rank's local P of a sharded level with grid hints: rows = the owned fine box f1 x f2 x f3 (== n_rows, natural
 * x-fastest order), columns = [owned coarse box c1 x c2 x c3 (== regular_cols) | halo].  Rows that read a halo column are
 * kept apart as exception rows; the others take the LDS-staged prolongation kernel of the single-GPU path where the
 * pattern fits it (checked on the data).  mg_op_apply_phase_dev_FP64: phase 1 = the rows that read owned coarse entries
 * only, phase 2 = the exception rows (ParSpMatVec.jl:49-71 semantics unchanged: y = alpha*P*x + beta*y).
 * With f1 = ... = c3 = 0 only the owned | halo split is made (any local operator, e.g. a restriction: rows that read a
 * halo column in phase 2, the others in phase 1, whatever kernel serves them). */
int mg_op_create_grid_FP64_INT64(long long device_id, long long n_rows, long long n_cols, const long long* colptr,
                                 const long long* rowval, const double* nzval, long long regular_cols, long long f1,
                                 long long f2, long long f3, long long c1, long long c2, long long c3, mg_operator** out);
/* Announce the relaxPrec vector (device; one entry per row, for a box operator per owned row) the operator will be
 * swept with.  Where it is bit-identical over every dictionary class of a row-class operator, the fused sweeps and the
 * fused residual called with THIS pointer (nrhs = 1, row_offset = 0) read it from the dictionary instead of streaming
 * 8 B/row - what mg_finalize does for the levels of a hierarchy (class_relax of mg_operator_rowclass_flags).  Call again
 * after the contents change.  mg_dist_finalize binds every level's vector itself. */
int mg_op_bind_relax_dev_FP64(mg_operator* op, const double* d_dev, long long n);

/* Which kernel fuses a damped-Jacobi sweep with the residual that follows it (MGcycle.jl:129-131 + 58-60) on this level's
 * A: *form = 0 none (two launches), 2 the 1-D chunk form, 3 the 2-D in-plane tile form, 4 the same with the coefficients
 * streamed per row (band form of a grid operator without row classes), 5 the 27-point marching form (csr_rowclass_march27_spmv: t and r
 * out only; it also serves single sweeps / residuals of the level; geometry: the pair's, [7] = workgroups of the single-product
 * geometry); geometry (optional, 12 entries, forms 3 and 4): tiles per line, tiles per column, TX, TY, rows per lane, workgroups, LDS bytes, estimated fill bytes per row x
 * 100, threads per workgroup, segments of the lockstep schedule (0: balanced), planes per segment, class-table entries. */
int mg_sweep_residual_form(mg_hierarchy* h, long long level, long long* form, long long* geometry);
/* Band form (form 4 above) of level `level`: info[0] = 1 when it is held; info[1] = 1 when the 7 value planes are in canonical
 * slots (-z, -y, -x, diagonal, +x, +y, +z: every structure class a 7-point star); info[2] = 1 when the values are symmetric
 * entry by entry (checked on the device whenever the values change: G' diag(sigma) G is) - the pass then reads the three lower
 * entries of a row from the upper ones of its neighbours; info[3] = value planes streamed from HBM per pass (7, or 4).
 * info[0] = 2: the level holds the band-27 form (27 planar arrays, kernel_variant 9); info[2] / info[3] likewise (27, or 14). */
int mg_band_form(mg_hierarchy* h, long long level, long long* info);
/* kernel variant serving the operator at nrhs == 1 (as mg_operator_rowclass_flags) and its exception rows */
int mg_op_kernel_variant(mg_operator* op, long long* variant, long long* exception_rows);
int mg_op_destroy(mg_operator* op);
int mg_op_apply_dev_FP64(mg_operator* op, long long kernel, double alpha, const double* x_dev,
                         double beta, double* y_dev, const double* b_dev, const double* d_dev,
                         long long nrhs, void* stream);
/* The operator holds the rows [row_offset, row_offset + n_rows) of the level (interior / boundary split of
 * the overlapped halo exchange): y, b, d and the smoother's own-x term are read/written at that offset of the
 * full-level vectors passed here; x is still the whole gathered vector. */
int mg_op_apply_rows_dev_FP64(mg_operator* op, long long kernel, double alpha, const double* x_dev,
                              double beta, double* y_dev, const double* b_dev, const double* d_dev,
                              long long nrhs, long long row_offset, void* stream);
/* phase 0: the whole product; 1: the rows in dictionary classes only (do not read the halo of a box operator);
 * 2: the exception rows only.  The sharded cycle runs phase 1 while the halo is in flight and phase 2 after it landed. */
int mg_op_apply_phase_dev_FP64(mg_operator* op, long long kernel, double alpha, const double* x_dev, double beta,
                               double* y_dev, const double* b_dev, const double* d_dev, long long nrhs,
                               long long row_offset, long long phase, void* stream);
/* r = b - M x with ||r||^2 fused: one partial sum per workgroup from partials_dev on (*nparts of them); optionally
 * xnext = x + d.*r as a second output (mg_op_can_fuse_next says whether the kernel serving the operator can write it),
 * in which case r_dev may be NULL.  phase as above; a phase-2 call gets the pointer advanced past phase 1's partials. */
int mg_op_residual_fused_dev_FP64(mg_operator* op, const double* x_dev, const double* b_dev, const double* d_dev,
                                  double* r_dev, double* xnext_dev, double* partials_dev, long long phase,
                                  long long* nparts, void* stream);
int mg_op_can_fuse_next(mg_operator* op, const double* x_dev, long long* yes);
/* The two-stage pass on a stand-alone operator (csr_rowclass_march3_spmv; what the sharded sequencer calls on the box-form
 * levels): t = x + d.*(b - M x) on every row that has a row class and r = b - M t [xn = t + d.*r, ||r||^2 partials] on every
 * such row that is not next to a row without one - the last sweep of relax and the residual that follows it
 * (MGcycle.jl:129-131 + 58-60; SolveFuncs.jl:26-30) in one pass.  The rows it leaves out are computed from the CSR arrays
 * by mg_op_apply_list_dev_FP64: list 1 = rows without a class (a box operator's rows that read the halo: t and r), list 2 =
 * rows next to them (r).  d: the vector bound with mg_op_bind_relax_dev_FP64; t, r, xn: each optional, all distinct. */
int mg_op_can_sweep_residual(mg_operator* op, const double* x_dev, const double* d_dev, long long* yes, long long* list1_rows,
                             long long* list2_rows);
int mg_op_sweep_residual_dev_FP64(mg_operator* op, const double* x_dev, const double* b_dev, const double* d_dev, double* t_dev,
                                  double* r_dev, double* xn_dev, double* partials_dev, long long* nparts, void* stream);
/* kernel: MG_K_SMOOTH (y = x + d.*(b - M x)) or MG_K_RESIDUAL (y = b - M x; y2, if given, = x + d.*y; partials_dev, if
 * given, receives *nparts partial sums of y.^2).  y or y2 may be NULL for MG_K_RESIDUAL. */
int mg_op_apply_list_dev_FP64(mg_operator* op, long long list, long long kernel, const double* x_dev, double* y_dev,
                              const double* b_dev, const double* d_dev, double* y2_dev, double* partials_dev, long long* nparts,
                              void* stream);
int mg_op_info(mg_operator* op, long long* n_rows, long long* n_cols, long long* nnz,
               double* device_bytes);
/* x = d.*b ; xout = x + d.*r ; out[0] = sum x^2 (workspace >= 1024 doubles) - asynchronous. */
int mg_vec_dscale_dev_FP64(const double* d_dev, const double* b_dev, double* x_dev, long long n,
                           long long nrhs, void* stream);
int mg_vec_xpdr_dev_FP64(const double* x_dev, const double* d_dev, const double* r_dev,
                         double* xout_dev, long long n, long long nrhs, void* stream);
int mg_vec_sumsq_dev_FP64(const double* x_dev, long long len, double* workspace_dev, double* out_dev,
                          void* stream);
/* Make a hierarchy enqueue on the caller's stream; enqueue one cycle without waiting. */
int mg_set_stream(mg_hierarchy* h, void* stream);
int mg_cycle_async_dev_FP64(mg_hierarchy* h, const double* b_dev, double* x_dev, long long n,
                            long long nrhs, long long x_is_zero);
/* The K-cycle's step INTO the hierarchy's first level (MGcycle.jl:72-76): x = 2 steps of FGMRES_relaxation on A_1 x = b
 * from x = 0, preconditioned by the K-cycle of level 1.  Used by the sharded sequencer when the level above the
 * replicated tail runs a K-cycle; one right-hand side; enqueues on the handle's stream (the FGMRES dots synchronise). */
int mg_kcycle_step_async_dev_FP64(mg_hierarchy* h, const double* b_dev, double* x_dev, long long n);

/* ---- hybrid Kaczmarz relaxation --------------------------------------------------------------------------
 * Replaces the reference's native applyHybridKaczmarz_FP64_INT64 (deps/src/parRelax.h:7-43; ccall at
 * src/Multigrid/parRelax.jl:61-64) with the same arguments: rowptr/colA 1-based Int64 and valA of the transposed CSC
 * (= CSR of A), ArrIdxs the domainLength x numDomains UInt32 array of 1-based row lists (0 = padding,
 * DDService.jl:2-18), invD = omega ./ sum(|A|^2 over each row) (parRelax.jl:44).  create uploads once; apply runs
 * `numit` sweeps on x (in/out) and b, n x nrhs column-major.  sequential = 1: one wavefront walks the sub-domains in
 * order (bit-identical to the reference binary with numCores = 1); 0: one wavefront per sub-domain, unsynchronised
 * between sub-domains like the reference's OpenMP threads. */
typedef struct mg_kaczmarz mg_kaczmarz;
int mg_kaczmarz_create_FP64_INT64(long long device_id, long long n, const long long* rowptr, const double* valA,
                                  const long long* colA, long long numDomains, long long domainLength,
                                  const unsigned int* ArrIdxs, const double* invD, mg_kaczmarz** out);
int mg_kaczmarz_apply_FP64(mg_kaczmarz* k, double* x, const double* b, long long nrhs, long long numit,
                           long long sequential);
int mg_kaczmarz_apply_dev_FP64(mg_kaczmarz* k, double* x_dev, const double* b_dev, long long nrhs, long long numit,
                               long long sequential);
int mg_kaczmarz_destroy(mg_kaczmarz* k);

/* ---- stand-alone sparse-factor applier -----------------------------------------------------------------------------
 * Device counterpart of applyLUsolve_FP64_INT64 (deps/src/parLU.cpp:52-63; ccall parallelJuliaSolver.jl:214-217): the
 * "Julia factors, native applies" back end 3 of src/ParallelJuliaSolver.  Factors exactly as setupLUFactor leaves them
 * (parallelJuliaSolver.jl:113-148): L and U in CSR with 1-based Int64 indices, L's diagonal LAST and U's diagonal FIRST
 * in every row, row scaling folded into L, p / q 1-based with A[p,q] = L*U.  solve: x[q] = U \ (L \ b[p])
 * (parLU.cpp:120-190); doTranspose != 0: the system with the transposed matrix, x[p] = L' \ (U' \ b[q])
 * (parLU.cpp:194-260) - the transposed factors are built and uploaded on first use.  b, x: n x nrhs column-major on
 * the host (b is left untouched: the reference uses it as work space), or row-major [n][nrhs] in HBM for the _dev
 * form.  All right-hand sides travel together (the reference: one OpenMP task per column). */
typedef struct mg_lu mg_lu;
int mg_lu_create_FP64_INT64(long long device_id, long long n, const long long* Lptr, const long long* Lcol,
                            const double* Lval, const long long* Uptr, const long long* Ucol, const double* Uval,
                            const long long* p, const long long* q, mg_lu** out);
int mg_lu_solve_FP64(mg_lu* f, const double* b, double* x, long long n, long long nrhs, long long doTranspose);
int mg_lu_solve_dev_FP64(mg_lu* f, const double* b_dev, double* x_dev, long long n, long long nrhs,
                         long long doTranspose);
int mg_lu_destroy(mg_lu* f);

/* ---- native multi-GPU sequencer (one process per GPU) ---------------------------------------------------------
 * The sharded cycle of src/DomainDecomposition's partition (box rule DDIndices.jl:41-47, numbering DDService.jl:27-48;
 * worker map analogue DDParallel.jl:105,133-139) behind the C ABI: the host cuts every sharded level into local
 * operators [owned | halo] (mg_op_create_FP64_INT64) and halo plans; this handle owns the vectors and the hot loop -
 * per SpMV the boundary values are packed, exchanged with ncclSend/ncclRecv inside ncclGroupStart/End on a SIDE stream
 * while the interior rows run, the boundary rows after the event; one scalar all-reduce per solveMG step; levels below
 * the sharded ones are all-gathered and run replicated by an ordinary mg_hierarchy (the tail).  One right-hand side.
 * Transport: RCCL (pass the 128-byte id of mg_dist_unique_id, broadcast by the host, to mg_dist_create) or a
 * host-staged plug-in (tests; several ranks sharing one GPU).  Levels are 1-based; `which` is MG_OP_A / P / R. */
typedef struct mg_dist mg_dist;
/* op 0: all_to_all (send/recv splits per peer, in doubles) ; 1: all_reduce sum of `count` doubles ; 2: all_gather of `count`
 * doubles per rank.  Host buffers.  Returns 0 on success. */
typedef int (*mg_exchange_fn)(void* user, long long op, const double* send, const long long* send_splits, double* recv,
                              const long long* recv_splits, long long count);
int mg_dist_unique_id(char* id128);
int mg_dist_create(long long device_id, long long rank, long long world, const char* unique_id128, long long nlevels_sharded,
                   long long nlevels_total, long long cycleType, mg_dist** out);
int mg_dist_set_exchange_plugin(mg_dist* h, mg_exchange_fn fn, void* user);
/* This rank's part of sharded level `level`: rows renumbered [interior | boundary] (interior = rows of A without halo
 * columns), A held as two operators, P / R with halo columns appended, relaxPrecs of the owned rows (device). */
int mg_dist_set_level(mg_dist* h, long long level, long long n_own, long long n_int, mg_operator* A_int, mg_operator* A_bnd,
                      mg_operator* P, mg_operator* R, const double* d_dev, long long relaxPre, long long relaxPost);
/* Halo plan of one operator: the source vector is [n_own_src owned | n_halo received]; send_idx (0-based, into the owned
 * part) grouped by destination rank with send_splits[world]; recv_splits[world]; active = 0 if no rank exchanges anything. */
/* The level's A was handed over as ONE box operator (A_int, n_int = n_own, A_bnd = NULL): phase 1 overlaps the exchange. */
/* Smoother of the sharded levels: 0 = pointwise (Jac / SPAI: the relaxPrec vectors), 1 = Jac-GMRES (FGMRES_relaxation
 * preconditioned by the relaxPrec vector, relaxPre / relaxPost = its inner dimensions: MGcycle.jl:48-50,96-98).  The
 * replicated tail carries its own setting (mg_set_relax_type).  cycleType 'K' (mg_dist_create) runs the K-cycle's FGMRES
 * steps on sharded levels with all-reduced dots (MGcycle.jl:72-76). */
int mg_dist_set_relax_type(mg_dist* h, long long relax_type);
/* Right-hand sides per call (default 1; row-major [n][nrhs] blocks; MGdef.jl:163-176, SolveFuncs.jl:30): right after
 * mg_dist_create.  b_loc / x_loc of the cycle and solve entry points are then n_own x nrhs blocks. */
int mg_dist_set_nrhs(mg_dist* h, long long nrhs);
int mg_dist_set_level_box(mg_dist* h, long long level, long long on);
int mg_dist_set_plan_INT64(mg_dist* h, long long level, long long which, long long n_own_src, long long n_halo,
                           long long n_send, const long long* send_idx, const long long* send_splits,
                           const long long* recv_splits, long long active);
/* The replicated tail: an mg_hierarchy of the levels below the sharded ones (its stream is re-pointed to this handle's),
 * this rank's share own_tail of its first level, the padded share max_tail, and gather_index[n_tail] into the
 * all-gathered [world][max_tail] array. */
int mg_dist_set_tail_INT64(mg_dist* h, mg_hierarchy* tail, long long n_tail, long long own_tail, long long max_tail,
                           const long long* gather_index);
int mg_dist_finalize(mg_dist* h);
/* b_loc / x_loc: this rank's fine rows (device, n_own doubles, the [interior | boundary] order). */
int mg_dist_cycle_dev_FP64(mg_dist* h, const double* b_loc, double* x_loc, long long n_own, long long x_is_zero);
int mg_dist_solve_dev_FP64(mg_dist* h, const double* b_loc, double* x_loc, long long n_own, double tol, long long maxIter,
                           long long* iters, double* resvec);
/* Number of ranks of the handle's RCCL communicator as the library itself reports it (ncclCommCount); 0 for the
 * host-staged plug-in transport. */
int mg_dist_comm_count(mg_dist* h, long long* count);
/* Hand the tail's mg_hierarchy back to its owner (stream and graph option as before mg_dist_set_tail_INT64).  Needed
 * only when the tail's handle is to be used or destroyed while this sequencer still exists; mg_dist_destroy does it
 * itself otherwise, so the tail must outlive the sequencer or be released first. */
int mg_dist_release_tail(mg_dist* h);
int mg_dist_destroy(mg_dist* h);

/* ---- sharded cycle with deep ghost layers (one process per GPU; csrc/mg_ghost.inc) -------------------------------------
 * The communication-avoiding form of the sharded cycle - the reference's `overlap` (getBoxWithOverlap,
 * src/DomainDecomposition/DDIndices.jl:61-92; box rule l.41-47; fan-out DDParallel.jl:87-105,133-139).  The host builds this
 * rank's part of the hierarchy on EXTENDED boxes - owned box + ghost layers towards every neighbour, every level with the width
 * its own passes consume (round 6: 5 on the fine level = four-stage pass + restriction, 3 on the coarser ones; a level's box ends on
 * nodes of the next level, whose box holds at least the parents of all its nodes - P and R then pair a box with a SUB-BOX of the next
 * one, which the transfer kernels find in the uploaded operators) - and uploads it as an ORDINARY mg_hierarchy (mg_create ...
 * mg_finalize): levels 1..nlevels_sharded are extended-box grid operators, the levels below them are replicated on every rank; the
 * restriction into the first replicated level has one row per node of that level, non-empty for the nodes this rank owns.  These
 * calls attach geometry, exchange plans and transport; afterwards the device-pointer entry points run SHARDED on vectors of the
 * extended fine box (b: owned rows valid on entry; x: owned rows valid on return), with every single-GPU kernel form (four-stage pass,
 * 27-point marching form, arithmetic transfers, pipelined stopping test): the library tracks on how many ghost layers each level
 * vector is still valid (a product with A costs one) and refreshes all layers at once where the next operation needs more - on the
 * fine level right behind the four-stage pass, overlapped with the whole coarse cycle on a side stream; exchanges awaited at once
 * stay on the compute stream.  A pass that would consume more layers than its input has is MG_ERR_STATE, never a clamp.  Norms, dots
 * and Gram matrices are sums over the owned rows of all ranks.  Served: mg_cycle_dev_FP64 / mg_solve_dev_FP64 (V / W / F / K cycles,
 * Jac / SPAI / Jac-GMRES, direct coarsest solve; a block of 2-24 right-hand sides - mg_create / mg_set_nrhs before the attach, vectors
 * [n_ext][nrhs] row-major - column by column where the column-wise solve serves the fine level: four-stage pass, V(2,*)), the
 * MG-preconditioned Krylov drivers mg_pcg_dev_FP64 / mg_bicgstab_dev_FP64 / mg_fgmres_dev_FP64 (SolveFuncs.jl:74-133) and their block
 * forms mg_block_*_dev_FP64.  Refused (MG_ERR_UNSUPPORTED): host-pointer entry points, coarseSolveType "GMRES", mg_rap_FP64; general
 * CSR (SA-AMG) hierarchies take mg_dist_*.  Transport: RCCL (the 128-byte id of mg_dist_unique_id) or the host-staged plug-in (ops 0 and 1). */
int mg_ghost_attach(mg_hierarchy* h, long long rank, long long world, long long nlevels_sharded, const char* unique_id128);
/* (RCCL transport, optional but recommended for world > 1) a SECOND communicator - another id of mg_dist_unique_id - for the
 * ghost-layer send / recv on the side stream: RCCL serialises the operations of one communicator in issue order whatever
 * their streams; with its own communicator the fine level's exchange overlaps the coarse cycle and its all-reduces. */
int mg_ghost_set_side_comm(mg_hierarchy* h, const char* unique_id128);
int mg_ghost_set_exchange_plugin(mg_hierarchy* h, mg_exchange_fn fn, void* user);
/* Sharded level `level` (1-based): extended box ext[3] (nodes per dimension, x fastest, 1 for unused dimensions); owned box
 * [own_lo, own_hi) in extended-box coordinates; gmin = the smallest ghost width over the cut sides of ANY rank (every rank
 * must take the same exchange decisions); send_idx: extended-box ids (0-based) of owned nodes grouped by destination rank,
 * recv_idx: extended-box ids of ghost nodes grouped by owner - each peer's part in ascending global id on both sides. */
int mg_ghost_set_level_INT64(mg_hierarchy* h, long long level, const long long* ext, const long long* own_lo,
                             const long long* own_hi, long long gmin, long long n_send, const long long* send_idx,
                             const long long* send_splits, long long n_recv, const long long* recv_idx,
                             const long long* recv_splits);
/* COLLECTIVE over the ranks (one all-reduce): they agree on what every rank's kernels can do (four-stage pass, fused sweep + residual, its
 * from-zero form, d.*bc out of the restriction) - the boxes of two ranks differ by a line or two, a format builder may serve one and not
 * the other, and every rank must exchange at the same points.  A block handle whose ranks do not all have the four-stage pass fails here. */
int mg_ghost_finalize(mg_hierarchy* h);
/* Timing aid: rank R of a world of N alone on its GPU - exchanges run their pack / unpack kernels, nothing travels, sums stay
 * local (before mg_ghost_finalize; not with RCCL).  Results are meaningless, the step time is one GPU's compute share. */
int mg_ghost_set_dry(mg_hierarchy* h, long long on);
/* exchanges started / doubles sent by this rank since mg_ghost_attach; ranks of the RCCL communicator (0: plug-in) */
int mg_ghost_stats(mg_hierarchy* h, long long* exchanges, long long* doubles_sent);
int mg_ghost_comm_count(mg_hierarchy* h, long long* count);
/* all-reduces this rank entered since mg_ghost_attach (the scalars of norms / dots, the rows of the first replicated level) */
int mg_ghost_allreduce_count(mg_hierarchy* h, long long* count);

const char* mg_last_error(void);
const char* mg_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MGVCYCLE_H */
