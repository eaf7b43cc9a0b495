/* ddm_hip.h -- C ABI of the MI355X-native two-level Schwarz + GenEO hot path.
 *
 * This is the drop-in boundary: a DUNE-side adaptor (header-only classes with the reference's
 * names, see dune-ddm_amd/dune/ddm/ and INTEGRATION.md) flattens its BCRSMatrix / BlockVector /
 * Interface objects once and then only calls the entry points below.  Plain pointers and sizes,
 * no C++ or torch types.  Citations are file:line in nilsfriess/dune-ddm (the reference).
 *
 * Conventions
 *   - All arithmetic FP64; column indices int32; row pointers / sizes int64.
 *   - "rank-local" vectors are the concatenation of the rank's subdomains (one subdomain per rank
 *     in the reference; several per GPU are allowed here so that the 8-subdomain problem also
 *     runs on 1, 2 or 4 GPUs).  n_o = non-overlapping size, n = overlapping size.
 *   - Vector arguments are DEVICE pointers unless the function name ends in _host.
 *   - Every call returns 0 on success or a negative DDM_E* code; ddm_last_error() gives the text
 *     (the adaptor turns it into DUNE_THROW, cf. dune/ddm/schwarz.hh:83,91,190,193).
 *   - All work is enqueued on the context's HIP stream; calls are asynchronous unless stated.
 *   - There is NO CPU fallback: without a HIP device every compute entry point fails.
 */
#ifndef DDM_HIP_H
#define DDM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDM_OK 0
#define DDM_EINVAL (-1)   /* bad argument / size mismatch (Dune::Exception, InvalidStateException) */
#define DDM_EHIP (-2)     /* HIP runtime error */
#define DDM_ENOTIMPL (-3) /* Dune::NotImplemented */
#define DDM_ENUMERIC (-4) /* zero pivot, singular coarse matrix, eigensolver failure */
#define DDM_ECOMM (-5)    /* exchange callback failed (MPI_Abort in the reference) */

typedef struct ddm_ctx ddm_ctx;
typedef struct ddm_csr ddm_csr;
typedef struct ddm_ilu0 ddm_ilu0;
typedef struct ddm_halo ddm_halo;
typedef struct ddm_op ddm_op;
typedef struct ddm_schwarz ddm_schwarz;
typedef struct ddm_galerkin ddm_galerkin;
typedef struct ddm_combined ddm_combined;

/* ---- context ---------------------------------------------------------------------------- */
/* One context per process/GPU.  stream == NULL: the library creates its own stream. */
int ddm_ctx_create(int device, void *hip_stream, ddm_ctx **out);
void ddm_ctx_destroy(ddm_ctx *ctx);
const char *ddm_last_error(const ddm_ctx *ctx); /* ctx == NULL: the calling thread's last failure of a context-free entry point */
int ddm_ctx_sync(ddm_ctx *ctx); /* hipStreamSynchronize */
/* the host waits for the work enqueued on the context's stream so far (event record + wait; later work is not waited for):
 * what an exchange callback of a host-driven transport (MPI) calls before it touches the packed send buffer */
int ddm_ctx_fence(ddm_ctx *ctx);
void *ddm_ctx_stream(ddm_ctx *ctx);

/* Inter-rank exchange is delegated to the host program (MPI in a DUNE build, torch.distributed
 * over RCCL in bench.py).  Both callbacks are invoked with the data already produced on the
 * context's stream; they must enqueue on (or synchronise with) that stream.
 *   alltoall: send/recv buffers are device pointers with the per-peer layout fixed at
 *             ddm_halo_create time; `tag` is the halo's id.
 *   allreduce_sum: in-place sum over all ranks of n doubles at a device pointer.
 * With no callbacks installed the context is single-rank (all subdomains local). */
typedef int (*ddm_alltoall_fn)(void *user, int tag, const double *sendbuf, double *recvbuf);
typedef int (*ddm_allreduce_fn)(void *user, double *buf, int64_t n);
int ddm_ctx_set_comm(ddm_ctx *ctx, int rank, int nranks, ddm_alltoall_fn a2a, ddm_allreduce_fn allreduce, void *user);

/* The same exchange INSIDE the library over RCCL (xGMI): grouped ncclSend / ncclRecv per halo and ncclAllReduce for the dots and the
 * coarse defect, all enqueued on the context's stream -- no callback round trip through the host program.  librccl is opened with
 * dlopen (a copy the host program already loaded is reused).  ddm_rccl_unique_id fills 128 bytes on ONE rank; the host program
 * distributes them (MPI_Bcast / torch.distributed.broadcast) and every rank calls ddm_ctx_set_rccl (collective: ncclCommInitRank).
 * self_test != 0: also route the rank's own halo segment and all reductions through RCCL (exercises the path on a single GPU). */
int ddm_rccl_unique_id(void *id128);
int ddm_ctx_set_rccl(ddm_ctx *ctx, int rank, int nranks, const void *id128, int self_test);
/* *count = number of ranks RCCL itself reports for the context's communicator (ncclCommCount), 0 without ddm_ctx_set_rccl:
 * lets a launcher check that the exchange really spans the ranks it started (bench.py prints it next to n_gpus). */
int ddm_ctx_rccl_size(ddm_ctx *ctx, int *count);
/* counts[3] = {all-reduces, doubles they carried, grouped halo exchanges} issued through this context so far, counted as a run over
 * several ranks launches them (one all-reduce / one send-receive group = one RCCL launch), also on a single rank where nothing is
 * sent: lets a benchmark report collectives per iteration.  In ddm_cg_steps the squared defect norm of an iteration rides on the
 * coarse-defect all-reduce of the next one (K + 1 doubles, galerkin_preconditioner.hh:170-183): per CG iteration 3 all-reduces
 * (<p, q>, <r, z>, coarse defect + norm) and 3 halo groups instead of 4 + 3. */
int ddm_ctx_comm_counts(ddm_ctx *ctx, int64_t *counts);

/* raw device memory helpers for callers that do not bring their own allocator */
int ddm_malloc(ddm_ctx *ctx, int64_t bytes, void **dptr);
int ddm_free(ddm_ctx *ctx, void *dptr);
int ddm_memset_zero(ddm_ctx *ctx, void *dptr, int64_t bytes);                 /* enqueued on the context's stream */
int ddm_memcpy_h2d(ddm_ctx *ctx, void *dst, const void *src, int64_t bytes); /* synchronous */
int ddm_memcpy_d2h(ddm_ctx *ctx, void *dst, const void *src, int64_t bytes); /* synchronous */

/* ---- CSR matrix (flattened Dune::BCRSMatrix<FieldMatrix<double,1,1>>; cf. the in-tree
 *      precedent dune/ddm/strumpack.hh:36-62) ---------------------------------------------- */
int ddm_csr_create(ddm_ctx *ctx, int64_t nrows, int64_t ncols, const int64_t *rowptr, const int32_t *col,
                   const double *val, ddm_csr **out);
/* The same object without device arrays, for matrices the library only reads on the host: A_neu / B_neu of ddm_geneo_basis (the
 * pencil A_neu + sigma D B D is assembled from the host arrays; at 216^3 the two device copies were 7 GB and 0.6 s for nothing).
 * Entry points that need the device arrays (products, factorisations) return DDM_EINVAL for it. */
int ddm_csr_create_host(ddm_ctx *ctx, int64_t nrows, int64_t ncols, const int64_t *rowptr, const int32_t *col, const double *val, ddm_csr **out); /* host arrays are copied */
void ddm_csr_destroy(ddm_csr *A);
int64_t ddm_csr_rows(const ddm_csr *A);
int64_t ddm_csr_nnz(const ddm_csr *A);
/* y = A x  (BCRSMatrix::mv, dune/ddm/nonoverlapping_operator.hh:37) */
int ddm_csr_mv(ddm_ctx *ctx, const ddm_csr *A, const double *x, double *y);
/* y += alpha A x  (BCRSMatrix::usmv, nonoverlapping_operator.hh:47) */
int ddm_csr_usmv(ddm_ctx *ctx, const ddm_csr *A, double alpha, const double *x, double *y);

/* Y = A X for row-major n x nrhs block vectors (the B*x / A*x products of the GenEO eigensolver,
 * eigensolvers/spectra.hh:100-105, on a block of vectors) */
int ddm_csr_mm(ddm_ctx *ctx, const ddm_csr *A, int nrhs, const double *X, double *Y);

/* Host-only (no device): the cache-blocked processing order the library uses for the block products A~ X, C~ X of the GenEO
 * eigensolver when the diagonal blocks of the matrix come from a structured grid in lexicographic numbering (strides read off the
 * column offsets most rows share; 16 x 4 x 4 bricks).  order_out[n] is always a permutation of the rows; returns 1 when a grid
 * structure was found, 0 when not (identity), < 0 on bad arguments.  A performance hint only: every row is computed as before. */
int ddm_csr_row_order_tiled_host(int64_t nblocks, const int64_t *block_ptr, const int64_t *rowptr, const int32_t *col, int32_t *order_out);

/* ---- local subdomain solver: ILU(0), natural row order ------------------------------------
 * The InverseOperator behind schwarz.hh:57,92,133 for [subdomain_solver] type=loopsolver maxit=1,
 * preconditioner type=ilu n=0.  block_ptr[0..nblocks] = row ranges of the independent diagonal
 * blocks (subdomains) of A; pass nblocks=1, block_ptr={0,n} for one subdomain. */
int ddm_ilu0_create(ddm_ctx *ctx, const ddm_csr *A, int64_t nblocks, const int64_t *block_ptr, ddm_ilu0 **out);
void ddm_ilu0_destroy(ddm_ilu0 *F);
/* x = (LU)^-1 d ; d and x must not alias */
int ddm_ilu0_solve(ddm_ctx *ctx, ddm_ilu0 *F, const double *d, double *x);
/* X = (LU)^-1 D for row-major n x nrhs block vectors (cf. the reference's multi-RHS triangular
 * solve eigensolvers/umfpack.hh:131-197); D and X must not alias */
int ddm_ilu0_solve_multi(ddm_ctx *ctx, ddm_ilu0 *F, int nrhs, const double *D, double *X);
/* the same with SINGLE-PRECISION sweeps (factor entries and the work block in float; D is read and X written in double): preconditioner
 * grade, relative error ~1e-6 x the growth of the triangular solves -- what the GenEO block iteration applies as W = T r (its eigenpairs
 * and residuals are computed in double).  nrhs % 4 != 0 or a sparse direct factor: the double sweeps. */
int ddm_ilu0_solve_multi_f32(ddm_ctx *ctx, ddm_ilu0 *F, int nrhs, const double *D, double *X);
int64_t ddm_ilu0_num_levels(const ddm_ilu0 *F, int upper);
/* status of the persistent (single-launch) triangular solve: 0 ok, 1 = a wave timed out waiting for a
 * dependency level (results invalid).  Synchronous.  DDM_TRSV_MODE=levels selects one launch per level. */
int ddm_ilu0_status(ddm_ctx *ctx, const ddm_ilu0 *F, int *status);
/* The same status word WITHOUT synchronising (it lives in pinned host memory the kernel writes to): non-zero as soon as a solve
 * that has finished gave up.  ddm_schwarz_apply / ddm_combined_apply (and with them the Krylov drivers) look at it on entry and
 * return DDM_ENUMERIC -- a time-out surfaces at the NEXT apply, not only in post().
 * Co-residency: the single-launch engines (`pipe`, `xcd2`) spin-wait on other workgroups of the same launch, so all their
 * workgroups (2 per CU) must be resident at once: ONE process per GPU, no other kernel holding CUs for the duration of a solve.
 * Every spin is bounded (it ends with the status word set, never in a hang); DDM_TRSV_MODE=levels (one launch per dependency
 * level, no spinning) is the engine for a GPU that is shared with other processes. */
int ddm_ilu0_peek_status(const ddm_ilu0 *F);
/* engine the next ddm_ilu0_solve uses: 8 = pipe, 4 = xcd2 (also when pipe declined the matrix), 0 = one launch per level */
int ddm_ilu0_engine(const ddm_ilu0 *F);
/* ddm_ilu0_create returns while the schedule of the single-launch engine is still being built on host threads (2.6 s at 216^3; the
 * first ddm_ilu0_solve waits for it, so does ddm_ilu0_engine).  ddm_ilu0_wait joins that work now and returns its error, so that a
 * caller can account the whole setup before it starts its clock.  DDM_PIPE_ASYNC=0: everything inside ddm_ilu0_create.  The matrix
 * handed to ddm_ilu0_create must stay alive as long as the factor (it always had to: the factor borrows its pattern). */
int ddm_ilu0_wait(ddm_ctx *ctx, ddm_ilu0 *F);
/* Diagnostic of the box engine (structured blocks; csrc/trsv_box.hpp): with DDM_BOX_CHECK=1 in the environment at creation its sweep
 * kernels stamp every plane of block 0; out1024 = [2 sweeps][128 planes][4] = {start, end (100 MHz clock), polls of the previous plane's
 * progress word, XCC id} of the last solve (zeros without the switch). */
int ddm_ilu0_box_check(const ddm_ilu0 *F, unsigned long long *out1024);
/* diagnostic: one solve with in-kernel cycle stamps of one compute wave */
int ddm_ilu0_debug_stamps(ddm_ctx *ctx, ddm_ilu0 *F, const double *d, double *x, unsigned long long *out6_host);
/* diagnostic: one solve with the stamped build of the pipe engine's kernel (DDM_TRSV_MODE=pipe, the default); per task 272
 * words, the first 16: start / first step / end (100 MHz ticks), cycles waiting for tiles / for producer tasks / in the signalled section,
 * steps, XCC id, four segment sums of the compute wave's step (csrc/trsv_pipe.hpp), then the time of each of the first 256
 * steps' result stores; meta_host (optional) receives group and
 * sweep per task.  out_host == NULL queries *ntasks. */
int ddm_ilu0_pipe_trace(ddm_ctx *ctx, ddm_ilu0 *F, const double *d, double *x, unsigned long long *out_host, int32_t *meta_host,
                        int64_t capacity_tasks, int64_t *ntasks);
/* factor values in the pattern of A (inverse pivots on the diagonal), for parity tests */
int ddm_ilu0_get_factors_host(ddm_ctx *ctx, const ddm_ilu0 *F, double *lu_host);

/* ---- sparse direct local solver -------------------------------------------------------------
 * What the reference gets from SuiteSparse through the solver factory: [subdomain_solver] type = cholmod / umfpack
 * (schwarz.hh:85-92, examples/poisson.ini:23,26) and the factorisation of A - sigma B inside the GenEO eigensolver
 * (eigensolvers/spectra.hh:28-89).  Sparse Cholesky of a symmetric positive definite block-diagonal matrix: nested-dissection
 * ordering, symbolic analysis and numeric factorisation on host threads (setup work, one thread per block, like the ILU(0)
 * factorisation); the factor is handed to the same device triangular-solve engines as an ILU(0) factor (the returned object IS a
 * ddm_ilu0: ddm_ilu0_solve / _solve_multi / _destroy apply), solves run in the fill-reducing order with a permutation on either side.
 * max_flops > 0: analyse first and return DDM_ENOTIMPL without factorising if the factorisation needs more floating-point
 * operations (sum of squared column counts) -- the caller then stays with ILU(0).  DDM_ENUMERIC: not positive definite. */
int ddm_chol_create(ddm_ctx *ctx, const ddm_csr *A, int64_t nblocks, const int64_t *block_ptr, double max_flops, ddm_ilu0 **out);
/* general != 0: L U without pivoting on the pattern of A + A^T, for non-symmetric matrices whose symmetric part is positive
 * definite (the DG convection-diffusion operator; `type = umfpack`); DDM_ENUMERIC on a vanishing pivot.  general == 0 = ddm_chol_create. */
int ddm_direct_create(ddm_ctx *ctx, const ddm_csr *A, int64_t nblocks, const int64_t *block_ptr, int general, double max_flops, ddm_ilu0 **out);
/* Engines of ddm_chol_create / ddm_direct_create, environment DDM_DIRECT_ENGINE = device | host.  Default: the device engine when the
 * factorisation needs at least DDM_DIRECT_DEVICE_MIN_FLOPS multiply-adds -- 2e10 for ddm_direct_create / the Schwarz local solver
 * (the host engine's CSR level solves are the faster single-vector solves, 1.31 against 1.75 ms on configs[4], but its factorisation
 * takes ~1 s per 1e10 multiply-adds: above 2e10 the device engine wins the time to solution), 1e10 for the factors the library
 * builds for a handful of block solves (GenEO preconditioner, harmonic extensions):
 *   device  SUPERNODAL Cholesky (general = 0) or L U (general = 1) with numeric factorisation AND solves on the GPU (csrc/sn_chol.hpp):
 *           nested-dissection supernodes of at most 128 columns, dense panels, FP64-MFMA updates, level by level of the supernodal
 *           elimination tree; the host only orders and analyses.  L U: threshold partial pivoting (0.1, UMFPACK's default) INSIDE the
 *           diagonal block of a supernode; rows are never exchanged between supernodes; a pivot column that vanishes inside its block
 *           is replaced by sqrt(eps) max|a_ij| (static perturbation).  ITERATIVE REFINEMENT as dune/ddm/eigensolvers/umfpack.hh:42-129
 *           (and as UMFPACK's own solve): backward error omega = ||b - A x|| / (||A||_inf ||x|| + ||b||), stop below 1e-14 or when a
 *           step does not halve it, at most 3 steps -- decided ONCE per factor on a probe right-hand side (the solves stay captured HIP
 *           graphs), then applied in every single- and multi-vector solve (ddm_ilu0_refinement reports it; DDM_DIRECT_REFINE = off |
 *           <steps> overrides).  A factor whose probe stays above 1e-9 is refused (DDM_ENUMERIC when the engine was forced, else the
 *           host engine takes over).  Results are bitwise REPRODUCIBLE since round 4: supernodes of one tree level whose updates would
 *           meet in an ancestor entry are coloured apart and the colours run one after the other; the forward sweeps write into one
 *           slot per (supernode, row) and the row's owner adds its slots in a fixed order -- no atomics anywhere.  DDM_ENOTIMPL if the
 *           panels do not fit into the free device memory.
 *   host    up-looking factorisation on host threads (L U without pivoting for general = 1), CSR level solves on the device
 *           (bitwise reproducible).
 * The host half of the device engine alone (no device needed; CPU tests): */
typedef struct ddm_sn_host ddm_sn_host;
int ddm_sn_host_create(int64_t n, const int64_t *rowptr, const int32_t *col, int64_t nblocks, const int64_t *block_ptr, ddm_sn_host **out);
void ddm_sn_host_destroy(ddm_sn_host *H);
/* sizes[4] = {supernodes, length of `rows`, panel entries (doubles), tree levels} of one block; *flops = multiply-adds */
int ddm_sn_host_sizes(const ddm_sn_host *H, int64_t block, int64_t *sizes, double *flops);
/* perm[n_b] (perm[new] = old, block-local), first[nsn + 1], rptr[nsn + 1], rows[], parent[nsn], level[nsn]; any pointer may be NULL */
int ddm_sn_host_get(const ddm_sn_host *H, int64_t block, int32_t *perm, int32_t *first, int64_t *rptr, int32_t *rows, int32_t *parent, int32_t *level);
int ddm_ilu0_is_direct(const ddm_ilu0 *F); /* 1 for a ddm_chol_create / ddm_direct_create factor */
int64_t ddm_ilu0_nnz(const ddm_ilu0 *F);   /* stored factor entries (L + D + U) */
/* iterative refinement of a device factor: returns the steps every solve performs; omega[5] (may be NULL) = backward error of the probe
 * right-hand side after 0, 1, .. steps (0 where not run) */
int ddm_ilu0_refinement(const ddm_ilu0 *F, double *omega);
/* The host part alone (no device needed; used by the CPU tests): va == NULL stops after the symbolic analysis.
 * get: perm[n] (perm[new] = old), and the factor in the storage convention of ddm_ilu0_get_factors_host -- CSR over the
 * PERMUTED indices with pattern L + D + L^T: unit lower factor, inverse pivots on the diagonal, D L^T above. */
typedef struct ddm_chol_host ddm_chol_host;
int ddm_chol_host_create(int64_t n, const int64_t *rowptr, const int32_t *col, const double *val, int64_t nblocks,
                         const int64_t *block_ptr, ddm_chol_host **out);
int ddm_direct_host_create(int64_t n, const int64_t *rowptr, const int32_t *col, const double *val, int64_t nblocks,
                           const int64_t *block_ptr, int general, ddm_chol_host **out);
void ddm_chol_host_destroy(ddm_chol_host *H);
int64_t ddm_chol_host_nnz(const ddm_chol_host *H);        /* entries of the factor CSR (0 after a symbolic-only run) */
int64_t ddm_chol_host_nnz_factor(const ddm_chol_host *H); /* nnz(L) from the symbolic analysis */
double ddm_chol_host_flops(const ddm_chol_host *H);       /* sum of squared column counts */
int ddm_chol_host_get(const ddm_chol_host *H, int32_t *perm, int64_t *rowptr, int32_t *col, double *lu);

/* ---- halo exchange plan: one per DUNE interface -------------------------------------------
 * (copyOwnerToAll: schwarz.hh:125, galerkin_preconditioner.hh:162;
 *  addOwnerCopyToOwnerCopy: nonoverlapping_operator.hh:38,48, schwarz.hh:138,142;
 *  addOwnerCopyToAll: galerkin_preconditioner.hh:190.)
 * send side : sendbuf[k] = v[send_idx[k]], k < nsend, grouped by destination rank
 *             (send_counts[nranks]); the segment for this rank itself is delivered locally.
 * recv side : recvbuf is the concatenation of the peers' segments (recv_counts[nranks]).
 *             For entry t < ndst:  v[dst_idx[t]] (=|+=) sum_{k=dst_ptr[t]}^{dst_ptr[t+1]-1} recvbuf[src_pos[k]]
 *             contributions are added in list order (the adaptor lists them by ascending source
 *             rank, which is the order DUNE's BufferedCommunicator scatters them).
 * mode      : 0 = copy (overwrite), 1 = add. */
int ddm_halo_create(ddm_ctx *ctx, int tag, int mode, int64_t nsend, const int64_t *send_idx,
                    const int64_t *send_counts, const int64_t *recv_counts, int64_t ndst, const int64_t *dst_idx,
                    const int64_t *dst_ptr, const int64_t *src_pos, ddm_halo **out);
void ddm_halo_destroy(ddm_halo *H);
int ddm_halo_exchange(ddm_ctx *ctx, ddm_halo *H, double *v); /* pack -> (callback) -> unpack, in place */
/* pack from src, unpack into dst (entries of dst outside dst_idx are left alone).  With a copy-mode halo that carries ONE
 * neighbour's index lists and dst zeroed beforehand this is CopyGatherScatterWithRank (galerkin_preconditioner.hh:66-103): the
 * neighbour's vector restricted to the shared indices, zero elsewhere -- no mask, no host round trip. */
int ddm_halo_exchange_to(ddm_ctx *ctx, ddm_halo *H, const double *src, double *dst);
/* Buffers the alltoall callback is handed (device pointers; stable for the halo's lifetime). */
double *ddm_halo_sendbuf(ddm_halo *H);
double *ddm_halo_recvbuf(ddm_halo *H);

/* ---- NonOverlappingOperator + NonOverlappingScalarProduct (nonoverlapping_operator.hh) ---- */
int ddm_op_create(ddm_ctx *ctx, const ddm_csr *A, ddm_halo *novlp_add, const uint8_t *owner_mask_host, ddm_op **out);
void ddm_op_destroy(ddm_op *op);
int ddm_op_apply(ddm_ctx *ctx, ddm_op *op, const double *x, double *y);                       /* :34-39 */
int ddm_op_applyscaleadd(ddm_ctx *ctx, ddm_op *op, double alpha, const double *x, double *y); /* :41-50 */
int ddm_dot(ddm_ctx *ctx, ddm_op *op, const double *x, const double *y, double *result_host); /* :76-81, synchronous */
int ddm_norm(ddm_ctx *ctx, ddm_op *op, const double *x, double *result_host);                 /* :83, synchronous */

/* ---- SchwarzPreconditioner (schwarz.hh:54-220) --------------------------------------------
 * type: 0 = standard, 1 = restricted (schwarz.hh:80-83).  pou may be NULL (schwarz.hh:140).
 * ext_map[n]  : index into the non-overlapping vector or -1  ("extend": schwarz.hh:121-122)
 * A_dir is only read during creation (factorisation). */
int ddm_schwarz_create(ddm_ctx *ctx, const ddm_csr *A_dir, int64_t nblocks, const int64_t *block_ptr, int64_t n_novlp,
                       const int32_t *ext_map_host, const double *pou_host, int type, ddm_halo *ovlp_copy,
                       ddm_halo *ovlp_add, ddm_schwarz **out);
/* the same with the `type` key of [schwarz.subdomain_solver] (schwarz.hh:85-92): "ilu0" (default of ddm_schwarz_create) or
 * "cholmod" / "ldl" = this library's sparse Cholesky (SPD input; DDM_ENUMERIC otherwise), "umfpack" = its L U without pivoting,
 * "direct" = Cholesky if the values are symmetric, else L U */
int ddm_schwarz_create_ex(ddm_ctx *ctx, const ddm_csr *A_dir, int64_t nblocks, const int64_t *block_ptr, int64_t n_novlp,
                          const int32_t *ext_map_host, const double *pou_host, int type, const char *subdomain_solver,
                          ddm_halo *ovlp_copy, ddm_halo *ovlp_add, ddm_schwarz **out);
void ddm_schwarz_destroy(ddm_schwarz *S);
int ddm_schwarz_apply(ddm_ctx *ctx, ddm_schwarz *S, double *x, const double *d); /* :115-149 */
int64_t ddm_schwarz_num_levels(const ddm_schwarz *S, int upper); /* dependency levels of the local L / U solve */
int64_t ddm_schwarz_factor_nnz(const ddm_schwarz *S); /* stored entries of the local solver's factor (ILU(0): nnz(A_dir); direct: nnz(L + U)) */
int ddm_schwarz_engine(const ddm_schwarz *S);                    /* ddm_ilu0_engine of the local solver */
/* Synchronous: DDM_OK, or DDM_ENUMERIC if a local solve since creation gave up waiting (results invalid).  apply has no
 * error return in the reference (schwarz.hh:131 discards the InverseOperatorResult); adaptors call this in post(). */
int ddm_schwarz_status(ddm_ctx *ctx, const ddm_schwarz *S);
/* the local solver object behind `solver->apply` (schwarz.hh:57,133; returned by SchwarzPreconditioner::getSolver(), :155):
 * borrowed, owned by S; ddm_ilu0_solve / _solve_multi / _status apply */
ddm_ilu0 *ddm_schwarz_local_solver(ddm_schwarz *S);

/* ---- GalerkinPreconditioner (galerkin_preconditioner.hh:40-363) ---------------------------
 * basis_host: kmax x n row-major (vector j contiguous), zero rows where a subdomain has fewer
 * vectors; sub_ptr[nsub+1]: overlapping row ranges of the rank's subdomains; coarse_index[nsub*kmax]:
 * global coarse row of (subdomain, vector) or -1; K = total coarse dimension;
 * a0inv_host: K x K row-major inverse of the coarse matrix R A R^T (every rank solves the
 * replicated coarse problem instead of the rank-0 gather/solve/scatter of :170-183).
 * Shares the ext_map / halos of the Schwarz object. */
int ddm_galerkin_create(ddm_ctx *ctx, int64_t n, int64_t n_novlp, const int32_t *ext_map_host, int64_t nsub,
                        const int64_t *sub_ptr, int64_t kmax, const double *basis_host, const int64_t *coarse_index,
                        int64_t K, const double *a0inv_host, ddm_halo *ovlp_copy, ddm_halo *ovlp_add,
                        ddm_galerkin **out);
void ddm_galerkin_destroy(ddm_galerkin *G);
int ddm_galerkin_apply(ddm_ctx *ctx, ddm_galerkin *G, double *x, const double *d); /* :151-194 */
/* Local slab of the Galerkin product (build_solver, :219-349; layout helpers.hh:252):
 * Y = A_dir * V for nvec device vectors V (nvec x n row-major) then out[i*nvec_r + j] = <R_i, Y_j>
 * restricted to the subdomain row range [row0,row1).  Used by the host to assemble R A R^T. */
int ddm_galerkin_products(ddm_ctx *ctx, const ddm_csr *A_dir, int64_t nleft, const double *left, int64_t nright,
                          const double *right, int64_t row0, int64_t row1, double *out_host);

/* ---- GenEO coarse-basis builder ---------------------------------------------------------------
 * GenEOCoarseSpace(A, B, pou, ptree, taskflow, prefix) -> get_basis() (dune/ddm/coarsespaces/coarse_spaces.hh:219-256, 286-331):
 * C = D B_neu D, the lowest nev eigenpairs of A_neu x = lambda C x per subdomain, v <- D v / ||D v||_2, entries of Dirichlet
 * DoFs zeroed (the caller's zero_at_dirichlet, examples/poisson.cc:235-238).  The fields of ddm_geneo_params are the keys of the
 * `<prefix>.eigensolver` sub-tree (dune/ddm/eigensolvers/eigensolver_params.hh:8-62); the block method behind it is described
 * in csrc/geneo.hpp (ncv / blocksize / maxit of the reference's single-vector Lanczos have no meaning for it).
 *   A_neu, B_neu : block-diagonal over the rank's subdomains (sub_ptr[nsub+1] row ranges), B_neu may be the same object as A_neu
 *   pou_host[n], dirichlet_host[n] (may be NULL)
 *   basis_host   : kmax x n row-major (vector j of every subdomain in row j, zero outside the subdomain's rows is NOT implied:
 *                  row j holds vector j of ALL local subdomains side by side); info->nev rows are written
 *   nconv[nsub]  : vectors to use per subdomain (= nev, or the count below `threshold` in threshold mode, spectra.hh:157-163)
 *   eigenvalues_host : nsub x kmax, ascending
 * Synchronous.  DDM_ENUMERIC if a projected eigenproblem breaks down; not converged within maxit is reported in info only. */
typedef struct {
  int32_t nev;             /* eigensolver.nev (default 16) */
  int32_t nev_max;         /* upper bound of the threshold mode (default 2 nev) */
  double tolerance;        /* eigensolver.tolerance (1e-5) */
  double shift;            /* eigensolver.shift (1e-3) */
  double threshold;        /* eigensolver.threshold (-0.5 = off) */
  int32_t maxit;           /* block iterations (400) */
  int32_t extra;           /* guard vectors iterated beyond nev (4); nev + extra <= 132 */
  int32_t seed;            /* start block */
  int32_t preconditioner;  /* 0 = sparse Cholesky of A + shift C if its flop count <= max_direct_flops, else ILU(0); 1 = ILU(0); 2 = Cholesky */
  double max_direct_flops; /* default: a per-rank TIME budget (DDM_GENEO_DIRECT_SECONDS, 6 s) x the measured factorisation rate (1.1e13 multiply-adds / s);
                            * the panels must also fit into 85 % of the free device memory */
  int32_t verbose;
  int32_t raw;             /* 1: return the eigenvectors normalised to ||v||_2 = 1 without the "v <- D v" of finalize_eigenvectors
                            * (the ring coarse spaces extend the ring eigenvectors first, coarse_spaces.hh:612-627) */
} ddm_geneo_params;
typedef struct {
  int32_t iterations, converged, used_direct, nev;
  double worst_residual, setup_s, iterate_s, direct_flops;
} ddm_geneo_info;
int ddm_geneo_params_default(ddm_geneo_params *p);
int ddm_geneo_basis(ddm_ctx *ctx, const ddm_csr *A_neu, const ddm_csr *B_neu, int64_t nsub, const int64_t *sub_ptr, const double *pou_host,
                    const uint8_t *dirichlet_host, const ddm_geneo_params *params, int64_t kmax, double *basis_host, int32_t *nconv,
                    double *eigenvalues_host, ddm_geneo_info *info);
/* MsGFEMCoarseSpace(A_neu, A_dir, pou, dirichlet_mask, subdomain_boundary_mask, ptree, taskflow, prefix = "msgfem")
 * (coarse_spaces.hh:663-831; the default coarse space of examples/poisson.ini:36): GenEO's eigenproblem with right-hand side
 * D A_neu D on the interior DoFs, restricted to the a-harmonic functions (A_dir u = 0 in interior rows); Dirichlet DoFs are left
 * out and get zero entries.  Arguments as ddm_geneo_basis plus boundary_host[n] (the subdomain boundary mask); A_dir must be
 * symmetric in its interior block (DDM_ENOTIMPL otherwise).  csrc/geneo.hpp describes how the saddle-point pencil of the
 * reference is replaced by an iteration inside the constrained subspace. */
int ddm_msgfem_basis(ddm_ctx *ctx, const ddm_csr *A_neu, const ddm_csr *A_dir, int64_t nsub, const int64_t *sub_ptr, const double *pou_host,
                     const uint8_t *dirichlet_host, const uint8_t *boundary_host, const ddm_geneo_params *params, int64_t kmax, double *basis_host,
                     int32_t *nconv, double *eigenvalues_host, ddm_geneo_info *info);
/* SVDCoarseSpace(A_ovlp, pou, subdomain_boundary_mask, dirichlet_boundary_mask, ptree, taskflow, prefix = "svd_coarse_space")
 * (coarse_spaces.hh:1268-1407): the n_vectors leading left singular vectors of T = D A_ii^-1 A_{i,Gamma}, zero outside the interior
 * DoFs; with mult_pou != 0 followed by finalize_eigenvectors (:1403).  basis_host: n_vectors x n row-major;
 * singular_values_host: nsub x n_vectors, descending.  T is never formed (csrc/geneo.hpp). */
int ddm_svd_basis(ddm_ctx *ctx, const ddm_csr *A_dir, int64_t nsub, const int64_t *sub_ptr, const double *pou_host, const uint8_t *dirichlet_host,
                  const uint8_t *boundary_host, int n_vectors, int mult_pou, double tolerance, int maxit, double *basis_host, double *singular_values_host,
                  ddm_geneo_info *info);
/* EnergyMinimalExtension(A, interior_indices, boundary_indices) (coarsespaces/energy_minimal_extension.hh:36-229), the building
 * block of the ring and harmonic-extension coarse spaces (coarse_spaces.hh:598, 1097, 1250): u_i = -A_ii^-1 (A [0; u_b])_i.
 * A may be block diagonal (block_ptr[nblocks+1]: one sparse direct factor per block; Cholesky if the interior block is symmetric,
 * LU without pivoting otherwise).  extend works IN PLACE on a row-major DEVICE block X (n x nrhs, leading dimension ldx) that
 * holds the boundary values: interior rows are overwritten, every other row is left alone; values in rows that are neither
 * interior nor boundary do not enter (they count as zero, :109-118). */
typedef struct ddm_harmonic ddm_harmonic;
int ddm_harmonic_create(ddm_ctx *ctx, const ddm_csr *A, int64_t nblocks, const int64_t *block_ptr, int64_t n_interior, const int64_t *interior_host,
                        int64_t n_boundary, const int64_t *boundary_host, ddm_harmonic **out);
void ddm_harmonic_destroy(ddm_harmonic *H);
int ddm_harmonic_extend(ddm_ctx *ctx, ddm_harmonic *H, int nrhs, double *X, int64_t ldx);
/* The two dense contractions of the block eigensolver on their own (FP64 MFMA, csrc/geneo_kernels.hpp), for row-major DEVICE
 * blocks split into subdomain row ranges sub_ptr[nsub+1]:
 *   gram  : G_host[s] = U[rows of s]^T V[rows of s]   (nsub matrices pu x pv, row-major; split-K over 2048-row chunks, summed in
 *           chunk order)                              -- Spectra's V^T B f / Gram products
 *   rotate: Out[rows of s, 0:q) = (Base[rows of s, 0:q) -) U[rows of s, 0:p) Y_host[s]   (Y: nsub matrices p x q; more than 48 output
 *           or 80 inner columns run as panels)
 *                                                     -- Spectra's compress_V (Arnoldi.h:310-329)
 * Synchronous. */
int ddm_blockvec_gram(ddm_ctx *ctx, int64_t nsub, const int64_t *sub_ptr, const double *U, int64_t ldu, int pu, const double *V, int64_t ldv,
                      int pv, double *G_host);
/* gram2_sym: G1_host[s] = U^T V1, G2_host[s] = U^T V2 (p x p) for products that are SYMMETRIC by construction (V1 = A~ U, V2 = C~ U:
 * the two projected matrices of the Rayleigh-Ritz step) -- one pass over U, upper tiles only, the lower triangle is the mirrored
 * upper one. */
int ddm_blockvec_gram2_sym(ddm_ctx *ctx, int64_t nsub, const int64_t *sub_ptr, const double *U, int64_t ldu, const double *V1, const double *V2, int64_t ldv, int p,
                           double *G1_host, double *G2_host);
int ddm_blockvec_rotate(ddm_ctx *ctx, int64_t nsub, const int64_t *sub_ptr, const double *U, int64_t ldu, int p, const double *Y_host, int q,
                        const double *Base, int64_t ldb, double *Out, int64_t ldo);
/* host logic of the Rayleigh-Ritz step, exposed for the CPU tests: symmetric eigen-decomposition (V: matrix in, eigenvectors as
 * columns out; w ascending) and the rank-revealing Rayleigh-Ritz of gC y = mu gA y (returns the rank used, < 0 on failure) */
int ddm_dense_sym_eig_host(int n, double *V, double *w);
int ddm_dense_rayleigh_ritz_host(int p, const double *gA, const double *gC, int keep, double tau, double *mu, double *Y);

/* ---- CombinedPreconditioner (combined_preconditioner.hh:39-180) --------------------------- */
/* mode 0 = additive, 1 = multiplicative (:56-69).  galerkin may be NULL (one-level). */
int ddm_combined_create(ddm_ctx *ctx, int mode, ddm_op *op, ddm_schwarz *schwarz, ddm_galerkin *galerkin,
                        ddm_combined **out);
void ddm_combined_destroy(ddm_combined *C);
int ddm_combined_status(ddm_ctx *ctx, const ddm_combined *C); /* ddm_schwarz_status of the fine level */
int ddm_combined_apply(ddm_ctx *ctx, ddm_combined *C, double *x, const double *d); /* :127-163 */

/* ---- outer Krylov loop: dune-istl CGSolver::apply as driven by examples/poisson.cc:311-319 -- */
typedef struct {
  int32_t iterations;
  int32_t converged;
  double def0;        /* initial defect norm */
  double reduction;   /* achieved ||r_k|| / ||r_0|| */
  double elapsed_s;   /* wall time of the loop (host clock, stream synchronised) */
} ddm_solve_result;
/* x: initial guess / solution; b: right-hand side, overwritten by the defect (as in dune-istl).
 * hist_host (may be NULL): maxit+1 doubles receiving ||r_0||, ||r_1||, ...
 * fixed_iterations > 0: run exactly that many iterations without testing convergence (bench). */
int ddm_cg_solve(ddm_ctx *ctx, ddm_op *op, ddm_combined *prec, double *x, double *b, double reduction, int maxit,
                 int fixed_iterations, double *hist_host, ddm_solve_result *res);

/* dune-istl RestartedGMResSolver::apply (left-preconditioned, modified Gram-Schmidt, Givens rotations; the
 * monitored norm is that of the preconditioned defect): [solver] type = restartedgmressolver,
 * examples/poisson.ini:12-17; default of dune/ddm/twolevel_schwarz.hh:121-130.  Needed for the
 * non-symmetric preconditioners (restricted Schwarz, multiplicative combination) and operators.
 * hist_host (may be NULL): maxit+1 doubles. */
int ddm_gmres_solve(ddm_ctx *ctx, ddm_op *op, ddm_combined *prec, double *x, double *b, double reduction, int maxit,
                    int restart, double *hist_host, ddm_solve_result *res);

/* dune-istl BiCGSTABSolver::apply ([solver] type = bicgstabsolver): right-preconditioned, two half steps per iteration, the defect norm
 * is tested after each half step.  hist_host (may be NULL): up to 2 maxit + 1 doubles (one per half step), *nhist receives the count;
 * res->iterations = ceil of the half-step counter (what dune-istl reports).  DDM_ENUMERIC on the breakdowns dune-istl aborts on. */
int ddm_bicgstab_solve(ddm_ctx *ctx, ddm_op *op, ddm_combined *prec, double *x, double *b, double reduction, int maxit, double *hist_host,
                       int32_t *nhist, ddm_solve_result *res);

/* The same loop in pieces, so that a caller can bracket an exact number of iterations
 * (bench.py): begin = "b -= A x; def0 = ||b||" (synchronous); steps = k iterations enqueued
 * without host synchronisation; defect = ||b|| of the last enqueued iteration (synchronous). */
typedef struct ddm_cg ddm_cg;
int ddm_cg_begin(ddm_ctx *ctx, ddm_op *op, ddm_combined *prec, double *x, double *b, ddm_cg **out);
int ddm_cg_steps(ddm_ctx *ctx, ddm_cg *cg, int k);
int ddm_cg_defect(ddm_ctx *ctx, ddm_cg *cg, double *def_host);
double ddm_cg_def0(const ddm_cg *cg);
void ddm_cg_end(ddm_ctx *ctx, ddm_cg *cg);

/* ---- instrumentation ---------------------------------------------------------------------
 * Named event timers mirroring the reference's Logger events ("Schwarz/local solve", ...,
 * dune/ddm/schwarz.hh:178-181).  Times are HIP-event milliseconds accumulated on the stream. */
int ddm_timing_enable(ddm_ctx *ctx, int on);
int ddm_timing_get(ddm_ctx *ctx, const char *name, double *total_ms, int64_t *count);
int ddm_timing_reset(ddm_ctx *ctx);

/* ---- input synthesis (host only; no ddm_ctx, no device) ------------------------------------
 * The Q1 diffusion matrix of a structured node box restricted to a node subset, in the caller's numbering, Dirichlet rows and columns
 * eliminated symmetrically: the matrices the reference receives from PDELab's assembler (A of make_communication,
 * A_dir / A_neu / B_neu of examples/pdelab_helper.hh:113-436) for the synthetic benchmark problem.  Row-by-row on host threads, bit for
 * bit what dune_ddm_amd/synth.py builds with numpy.  All shapes x first.  Two calls: indices == NULL fills indptr[n + 1] (row pointers),
 * the second call (same arguments, indptr kept) fills indices / data.  inset, loc_of_box, box_index, diag may be NULL (all nodes,
 * identity numbering, unit Dirichlet diagonal). */
int ddm_synth_q1_matrix(int dim, const int64_t *bshape, const double *ke, const int64_t *eshape, const int64_t *eoff, const double *K,
                        const uint8_t *inset, const int64_t *loc_of_box, int64_t n, const int64_t *box_index, const uint8_t *dmask,
                        const double *diag, int64_t *indptr, int32_t *indices, double *data, int nthreads);

#ifdef __cplusplus
}
#endif
#endif /* DDM_HIP_H */
