/*
 * spk.h -- C ABI of libspk.so: the MI355X-native replacement for what
 * KSPSolve() executes at /root/reference/src/SaddlePointProblem.c:70.
 *
 * The reference reaches its solver through six PETSc calls
 * (SaddlePointProblem.c:65-72):
 *     KSPCreate, KSPSetOperators(ksp,A,A), KSPSetFromOptions, KSPSetUp,
 *     KSPSolve(ksp,f,u), KSPDestroy.
 * Everything below KSPSolve (MatMult_*AIJ, PCApply_Jacobi,
 * PCApply_FieldSplit_Schur, KSPSolve_FGMRES, VecMDot/VecMAXPY/VecNorm,
 * VecScatter halo, MPI_Allreduce) runs inside PETSc.  This header is what a
 * PETSc plugin (PCSHELL / MATSHELL / KSPRegister'd type, see
 * plugin/spk_petsc.c and INTEGRATION.md) binds instead.  Plain C: opaque
 * context, raw CSR / vector pointers, sizes, int status.  No PETSc, no torch.
 *
 * Conventions
 *   - every function returns 0 on success, a negative SPK_ERR_* otherwise;
 *     spk_last_error() then holds a message.  Non-convergence is NOT an error:
 *     it is reported through spk_result.reason (PETSc KSPConvergedReason
 *     values), as KSPSolve does.
 *   - the caller owns every array it passes; the library copies at
 *     spk_set_block() and never keeps host pointers.  All device memory
 *     belongs to the context and is released by spk_destroy().
 *   - one context per KSP; no global state; a context is driven by one host
 *     thread at a time; every call is synchronous at return.
 *   - vectors are FP64.  A system vector is [u ; lambda]: the rank's n_local
 *     rows of the (0,0) block followed by ALL m constraint multipliers
 *     (replicated on every rank).  m = 0 when no constraint block is set.
 *   - `mem` arguments: SPK_MEM_HOST or SPK_MEM_DEVICE for the x/y/b pointers.  Device
 *     vectors must come from spk_vec_create (zero-padded to a whole 16-byte pair: the
 *     kernels read and write vectors two doubles at a time).
 */
#ifndef SPK_H
#define SPK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPK_VERSION 100

typedef struct spk_ctx spk_ctx;

enum { SPK_OK = 0, SPK_ERR_ARG = -1, SPK_ERR_HIP = -2, SPK_ERR_STATE = -3,
       SPK_ERR_COMM = -4, SPK_ERR_NOMEM = -5, SPK_ERR_UNSUPPORTED = -6 };

enum { SPK_MEM_HOST = 0, SPK_MEM_DEVICE = 1 };

/* Blocks of the nest K = [A00 A01; A10 0] sketched at
 * SaddlePointProblem.c:45-60 (MatGetSize/MatSetSizes(B, 4, nCols)).
 * A01 is always A10^T and is derived by the library. */
enum { SPK_BLOCK_A00 = 0, SPK_BLOCK_A10 = 1 };

/* -pc_type {none,jacobi,fieldsplit(schur)} as read by KSPSetFromOptions
 * (SaddlePointProblem.c:67). */
enum { SPK_PC_NONE = 0, SPK_PC_JACOBI = 1, SPK_PC_SCHUR = 2 };
/* -pc_fieldsplit_schur_fact_type */
enum { SPK_SCHUR_DIAG = 0, SPK_SCHUR_LOWER = 1, SPK_SCHUR_UPPER = 2, SPK_SCHUR_FULL = 3 };
/* -ksp_gmres_{classical,modified}gramschmidt */
enum { SPK_ORTHOG_CGS = 0, SPK_ORTHOG_MGS = 1 };
/* -ksp_gmres_cgs_refinement_type {never,ifneeded,always} */
enum { SPK_REFINE_NEVER = 0, SPK_REFINE_IFNEEDED = 1, SPK_REFINE_ALWAYS = 2 };
/* SpMV storage of the (0,0) block */
enum { SPK_SPMV_CSR = 0 };

/* KSPConvergedReason values (PETSc numbering). */
enum { SPK_CONVERGED_RTOL = 2, SPK_CONVERGED_ATOL = 3, SPK_CONVERGED_ITS = 4,
       SPK_CONVERGED_HAPPY_BREAKDOWN = 7, SPK_DIVERGED_NULL = -2,
       SPK_DIVERGED_ITS = -3, SPK_DIVERGED_DTOL = -4, SPK_DIVERGED_BREAKDOWN = -5,
       SPK_DIVERGED_NANORINF = -9, SPK_ITERATING = 0 };

/* Solver options = the slice of the PETSc options database the reference
 * exposes through KSPSetFromOptions (SaddlePointProblem.c:67).  Fill with
 * spk_default_opts() first (PETSc defaults). */
typedef struct spk_opts {
    int32_t restart;        /* -ksp_gmres_restart            (30); 1..1022 -- restart + m <= 62: the fused iteration forms;
                               beyond: the head kernel stays, Gram-Schmidt runs in chunks of 40 vectors and the
                               Givens step is a launch of its own */
    int32_t max_it;         /* -ksp_max_it                   (10000) */
    double rtol;            /* -ksp_rtol                     (1e-5)  */
    double abstol;          /* -ksp_atol                     (1e-50) */
    double dtol;            /* -ksp_divtol                   (1e4)   */
    int32_t guess_nonzero;  /* -ksp_initial_guess_nonzero    (0)     */
    int32_t orthog;         /* SPK_ORTHOG_*                  (CGS)   */
    int32_t check_every;    /* host looks at the device convergence word every
                               this many iterations (a stream synchronisation each
                               time); 0 = once per restart cycle, without draining the
                               stream: the cycle's first kernel reports the state into
                               pinned memory and the host reads it while the cycle runs.
                               The iterate never depends on it.       */
    int32_t fused;          /* 1: fused PC+operator kernels where the PC allows,
                               0: PCApply and MatMult as separate steps */
    int32_t cgs_refine;     /* -ksp_gmres_cgs_refinement_type: SPK_REFINE_* (never) */
    int32_t single_reduce;  /* head-kernel paths with CGS only, OFF by default: 1 = h = V^T w, B D w and
                               w.w from ONE pass and ONE all-reduce per iteration; ||w'||^2 = w.w - |h|^2
                               and B D w' by recurrence.  Every scalar the next iteration's head needs
                               is then known before the update starts, so MAXPY, VecScale, PCApply
                               and the Givens step run as ONE launch: three launches per iteration
                               instead of four (1024^2: 227 -> 219 us, a 1/8 slab: 51 -> 45 us) and one
                               collective per iteration instead of two on many GPUs.  The price: the
                               subtraction cancels (||w'|| << ||w|| behind a good preconditioner):
                               measured 5e-6 relative drift of the residual history inside the
                               first cycle at 1024^2 (two-reduction path: 3e-11).  Safeguards: below
                               64 eps w.w the difference is kept at that floor (an over-estimated
                               ||w'|| over-estimates the residual), and a convergence seen by the
                               recurrence only ENDS THE CYCLE -- the solve ends when the true
                               residual computed at the restart confirms it, never on the
                               recurrence alone.  Needs restart + m <= 63 (falls back to two
                               reductions otherwise). */
    int32_t iteration_form; /* how the head-kernel paths launch one classical Gram-Schmidt iteration (same
                               algorithm, two reductions, norms taken from w' itself):
                               SPK_ITER_AUTO (0): SPK_ITER_UNNORM wherever it applies (classical Gram-Schmidt
                               without refinement, two reductions), else four launches;
                               SPK_ITER_UNNORM (5): three launches on an UN-NORMALISED basis -- VecMDot (raw inner
                               products and B D w~), VecMAXPY + norm + the next PCApply (+ B^T part), plain MatMult
                               carrying the Givens step in one extra workgroup.  V~_j = h_{j,j-1} v_j is stored with
                               a scale factor beside it and every consumer scales the scalars, never the vectors:
                               no VecScale traffic, either matrix format, any number of ranks;
                               SPK_ITER_FOUR_LAUNCH (1): head (VecScale + PCApply), SpMV, MDot, MAXPY;
                               SPK_ITER_TWO_LAUNCH (2): SpMV with MDot in its tile epilogues (VecScale of v and z
                               folded in), MAXPY with the norm and the next iteration's preconditioner + B^T
                               product on the un-normalised vector (B D w' by linearity from B D w).  Needs
                               the 2x2-blocked matrix layout and restart + m <= 62; otherwise four launches;
                               SPK_ITER_THREE_LAUNCH (3): as 2 with VecMDot (h and B D w) as a launch of its own;
                               SPK_ITER_BA (4, single rank, small systems): VecMAXPY + norm + next PCApply AND the next
                               MatMult in one launch behind neighbour flags, un-normalised basis (two launches per
                               iteration; measured no faster than 5: bandwidth-bound).
                               SPK_ITER_RESIDENT (6; what AUTO takes on small single-rank systems: <= 512 block rows per
                               compute unit, restart <= 30, the row-type matrix layout): ONE launch per restart cycle, one
                               workgroup per CU, every thread keeps its entries of the un-normalised basis in registers --
                               VecMDot is a register dot product + an all-to-all of the partial sums (every workgroup adds
                               all of them in one order and runs the Hessenberg / Givens / convergence scalars itself),
                               VecMAXPY touches no memory, the product gathers z~ from the neighbours' write-through
                               stores.  Same algorithm and basis as 5.  SPK_RESIDENT=0 keeps AUTO on 5;
                               Measured us per iteration, forms 1 / 3 / 5: 1/8 slab of 1024^2 47.9 / 44.7 / 43.3,
                               512^2 71.8 / 67.6 / 66.0, 1024^2 219.7 / 218 / 209.2 (profiles/r02*). */
    int32_t reserved;
} spk_opts;
enum { SPK_ITER_AUTO = 0, SPK_ITER_FOUR_LAUNCH = 1, SPK_ITER_TWO_LAUNCH = 2, SPK_ITER_THREE_LAUNCH = 3, SPK_ITER_BA = 4,
       SPK_ITER_UNNORM = 5, SPK_ITER_RESIDENT = 6, SPK_ITER_LAST = 6 };

typedef struct spk_result {
    int32_t its;            /* KSPGetIterationNumber   */
    int32_t reason;         /* KSPGetConvergedReason   */
    double rnorm;           /* KSPGetResidualNorm (unpreconditioned estimate) */
    double rnorm0;          /* residual norm at iteration 0 */
    int32_t hist_len;       /* entries written to history[] */
    int32_t cycles;         /* restart cycles executed */
    double solve_seconds;   /* wall time inside spk_fgmres, upload excluded */
} spk_result;

/* ---- lifetime (KSPCreate / KSPDestroy, SaddlePointProblem.c:65,72) ------- */
int spk_create(spk_ctx **ctx, int device);
int spk_destroy(spk_ctx *ctx);
/* Message of the last failure on ctx (ctx may be NULL: last spk_create error). */
const char *spk_last_error(const spk_ctx *ctx);
int spk_version(void);
void spk_default_opts(spk_opts *opts);

/* ---- multi-GPU wiring (replaces PETSC_COMM_WORLD, SaddlePointProblem.c:65) */
/* One process per GPU: rank 0 calls spk_comm_unique_id, the host side
 * broadcasts the 128 bytes (MPI_Bcast in a PETSc plugin, torch.distributed in
 * bench.py), every rank calls spk_comm_init_rccl before spk_set_block. */
int spk_comm_unique_id(void *id128);
int spk_comm_init_rccl(spk_ctx *ctx, int rank, int nranks, const void *id128);
/* Host-callback transport (rehearsal only: lets N processes share ONE GPU, which RCCL
 * refuses, so that the multi-process flow of bench.py can be run on a 1-GPU box over
 * gloo/MPI).  Every collective is staged through the host and synchronises the stream.
 *   allreduce(user, buf, count): in-place sum over ranks of `count` doubles
 *   exchange(user, peer, send, nsend, recv, nrecv): one matched send/recv pair of doubles
 *   allgather(user, in, out, bytes_each): bytes from every rank, rank order */
typedef struct spk_host_comm {
    void *user;
    int (*allreduce)(void *user, double *buf, int count);
    int (*exchange)(void *user, int peer, const double *send, int64_t nsend, double *recv, int64_t nrecv);
    int (*allgather)(void *user, const void *in, void *out, int64_t bytes_each);
} spk_host_comm;
int spk_comm_init_host(spk_ctx *ctx, int rank, int nranks, const spk_host_comm *cb);
/* Peer-store collectives (call after spk_comm_init_*, before spk_set_block; collective over the
 * ranks).  The Krylov all-reduces (<= 64 doubles) and the halo rows are then written by the solver's
 * own kernels straight into windows of the peers' HBM over xGMI -- 8-byte {sequence, payload}
 * granules that are their own arrival flags: one network traversal, no RCCL launch, and where the
 * kernel allows it no launch at all (the all-reduce rides in the finish of the reducing kernel).
 * Sums are formed in rank order on every rank: all ranks hold the same bits.  The windows are
 * uncached device memory shared through HIP IPC; the communicator set before stays in place for
 * set-up traffic and as the fallback.  *enabled = 1 when every rank mapped every window and a
 * self-test all-reduce gave the right sums everywhere, else 0 (the previous backend keeps
 * working; spk_comm_backend() tells which one is active).  2..8 ranks.  Device-side waits are
 * bounded (SPK_PEER_TIMEOUT_MS, default 30000): a rank that never arrives turns into SPK_ERR_COMM. */
int spk_comm_enable_peer(spk_ctx *ctx, int32_t *enabled);
/* "self" | "rccl" | "host-callback" | "local" | "peer-store" */
const char *spk_comm_backend(const spk_ctx *ctx);
/* Diagnostics of the communicator, per rank (bench.py gathers them to rank 0 so that an N-GPU run
 * explains itself): which backend is active and why the peer-store backend stayed off if it did,
 * which kind of window memory passed the self-test, how many collectives went which way, and how
 * long the device waited inside them (100 MHz ticks, accumulated by one lane per collective). */
typedef struct spk_comm_info {
    int32_t rank, nranks;
    int32_t peer_enabled;        /* 1: peer-store collectives active */
    int32_t window_tier;         /* 0 uncached, 1 fine-grained, 2 plain device memory, -1 none */
    int32_t self_test_ok;        /* all-reduce self-test passed on every rank */
    int32_t halo_mode;           /* 0 none, 1 granules, 2 bulk chunks, 3 inner backend (fallback) */
    int32_t halo_fused;          /* 1: the exchange rides inside the head kernels */
    int32_t device;
    int64_t n_allreduce_fused;   /* all-reduces executed in the finish of a reducing kernel */
    int64_t n_allreduce_kernel;  /* as stand-alone granule launches */
    int64_t n_allreduce_inner;   /* handed to the inner backend (RCCL / host) */
    int64_t n_halo_fused, n_halo_kernel, n_halo_inner;
    uint64_t wait_ticks[4];      /* [0] all-reduce after MDot, [1] after MAXPY, [2] stand-alone, [3] halo */
    uint64_t wait_count[4];
    char backend[32];
    char inner_backend[32];
    char why[256];               /* reason the peer-store backend is off / last set-up message */
} spk_comm_info;
int spk_comm_get_info(spk_ctx *ctx, spk_comm_info *info);
/* Test hook: a `nranks`-rank peer-store all-reduce played by `nranks` workgroups of one launch through
 * `nranks` windows that all live in this process (no IPC): exercises every lane of the window layout
 * (up to 8) on one device.  vals: nranks x count inputs; out: nranks x count results (identical rows). */
int spk_debug_peer_allreduce_loopback(spk_ctx *ctx, int nranks, int count, int rounds, const double *vals, double *out);
/* In-process logical ranks on one device (parity tests of the partitioned
 * algorithm on a 1-GPU box): a group is shared by `nranks` contexts, each
 * driven by its own host thread. */
typedef struct spk_local_group spk_local_group;
int spk_local_group_create(spk_local_group **grp, int nranks);
int spk_local_group_destroy(spk_local_group *grp);
int spk_comm_init_local(spk_ctx *ctx, spk_local_group *grp, int rank);

/* ---- operators (KSPSetOperators, SaddlePointProblem.c:66) ---------------- */
/* CSR rows [row_begin, row_begin+nrows_local) of a block with ncols_global
 * columns; colidx are GLOBAL column numbers, int32 (PetscInt), ascending or
 * not.  The library splits diagonal / off-rank columns and builds the halo
 * plan (what MatMPIAIJ + VecScatter do in the reference's PETSc).
 *   A00: the rank's row slab of A; rows must tile [0,n) in rank order.
 *   A10: ALL m rows of B restricted to the rank's owned columns
 *        (column-partitioned like the rows of A00); pass row_begin = 0,
 *        nrows_local = m. */
int spk_set_block(spk_ctx *ctx, int which, int64_t row_begin, int32_t nrows_local,
                  int64_t ncols_global, const int32_t *rowptr, const int32_t *colidx,
                  const double *val);

/* ---- preconditioner (KSPSetUp, SaddlePointProblem.c:68) ------------------ */
/* Builds diag(A)^-1 and, for SPK_PC_SCHUR, S^ = diag(B diag(A)^-1 B^T). */
int spk_pc_setup(spk_ctx *ctx, int pc_type, int schur_fact);
/* Inner solve standing for A^-1 inside the preconditioner (BASELINE config 5: "mixed FP32
 * inner solve"; PETSc: -fieldsplit_0_ksp_type richardson -fieldsplit_0_ksp_max_it k
 * -fieldsplit_0_ksp_richardson_scale omega -fieldsplit_0_pc_type jacobi): `sweeps` damped-Jacobi
 * Richardson sweeps on A in SINGLE precision,
 *     y_1 = omega D^-1 x ;  y_{s+1} = y_s + omega D^-1 (x - A y_s),
 * input and result converted from / to FP64 (the outer FGMRES is flexible, so an inexact,
 * lower-precision A^-1 is legitimate).  sweeps = 0 (default) keeps the plain diag(A)^-1.
 * Call before spk_pc_setup. */
int spk_pc_set_inner(spk_ctx *ctx, int sweeps, double omega);
/* Copies S^ (m doubles) to the host, for inspection. */
int spk_get_schur_diag(spk_ctx *ctx, double *shat);
int spk_get_jacobi_diag(spk_ctx *ctx, double *dinv /* n_local */);
/* Dense planes of B diag(A)^-1 the fused Schur kernels stream per pass: m, or m/2 when rows 2q / 2q+1
 * live on even / odd vector entries (x / y degrees of freedom of a dof-2 grid) and share a plane;
 * 0 when the fused path is not set up.  For byte models. */
int spk_get_bd_planes(const spk_ctx *ctx, int32_t *planes);

/* ---- the three plug points ------------------------------------------------ */
/* MATSHELL:  y = K x.   PCSHELL: y = M^-1 x.   Lengths n_local + m. */
int spk_mult(spk_ctx *ctx, const double *x, double *y, int mem);
int spk_pc_apply(spk_ctx *ctx, const double *x, double *y, int mem);
/* KSP type: the whole KSPSolve (SaddlePointProblem.c:70) on the device.
 * b, x: n_local + m values.  history (may be NULL) receives the residual norm
 * per iteration, starting with iteration 0. */
int spk_fgmres(spk_ctx *ctx, const double *b, double *x, int mem, const spk_opts *opts,
               spk_result *result, double *history, int32_t history_cap);

/* How the LAST spk_fgmres on ctx launched its iterations: *form = the SPK_ITER_* actually run (AUTO resolved; options
 * the chosen form does not cover fall back, see spk_opts.iteration_form), or -1 for the step-by-step path (PCApply and
 * MatMult as launches of their own: unfused preconditioners, FP32 inner sweeps, general constraint blocks; a restart
 * beyond 62, MGS or CGS refinement on a fusable preconditioner report SPK_ITER_FOUR_LAUNCH: the head kernel runs);
 * *single_reduce = 1 when the single-reduction mode ran.  For byte models (bench.py). */
int spk_get_iteration_form(const spk_ctx *ctx, int32_t *form, int32_t *single_reduce);

/* ---- device vectors for callers that keep b/x resident in HBM --------------- */
/* (a PCSHELL/MATSHELL glue over device Vecs, bench.py).  Zero-filled, length
 * rounded up so that every kernel may read whole 16-byte pairs. */
int spk_vec_create(spk_ctx *ctx, int64_t n, double **dev);
int spk_vec_destroy(spk_ctx *ctx, double *dev);
int spk_vec_set(spk_ctx *ctx, double *dev, const double *host, int64_t n);
int spk_vec_get(spk_ctx *ctx, const double *dev, double *host, int64_t n);

/* ---- sizes ---------------------------------------------------------------- */
int spk_get_sizes(const spk_ctx *ctx, int64_t *n_global, int32_t *n_local, int32_t *m,
                  int64_t *nnz_local, int32_t *n_ghost);

/* Storage the A-block SpMV actually streams: format 0 = CSR (12 B per stored non-zero),
 * 1 = 2x2-blocked CSR (36 B per 4 non-zeros; chosen automatically when rows 2k, 2k+1 share
 * their pattern and columns pair up, as for a dof-2 DMDA; SPK_SPMV_FORMAT=csr forces CSR).
 * layout_bytes = bytes one SpMV reads and writes in that layout (matrix + x + y). */
int spk_get_spmv_info(const spk_ctx *ctx, int32_t *format, int64_t *layout_bytes);
/* Formats 3 / 4: ROW TYPES + DEVIATION CODES over the 2x2 / 3x3 blocks (the default wherever the layout exists).  On the
 * reference's uniform grid (Discretization.c:25; one element matrix for all elements, :293-332) the assembled entries
 * scatter by rounding noise around a handful of ideal values (:96-128: a Jacobian formed from coordinates): A holds a few
 * dozen block CLASSES (blocks equal up to that noise) in a few dozen ROW TYPES (sequences of (column offset, class)).
 * They are found in the caller's CSR at spk_set_block (hashing, then EVERY value decoded and compared bit for bit;
 * nothing is assumed about the grid).  Stored: two bytes of type per block row and per value an integer k with
 * value = base + k 2^g exactly, as a bit field of the width its class entry needs (one 64-bit word per 2x2 block, two
 * per 3x3 block) -- same products, same order, same bits as the CSR loop.  Matrices that do not fit keep formats 0..2;
 * SPK_SPMV_FORMAT=bcsr (or csr) switches the layout off, SPK_DICT_VERBOSE=1 reports what was found or why not.
 * Byte models of one product y = A x on this rank's diagonal block in the three layouts (0 where a layout does not
 * exist), and the layout's size (patterns = row types, blocks = block classes). */
int spk_get_spmv_models(const spk_ctx *ctx, int64_t *csr_bytes, int64_t *blocked_bytes, int64_t *dict_bytes,
                        int32_t *patterns, int32_t *blocks);

/* ---- single kernels through the ABI (parity tests, bench.py) -------------- */
/* h[i] = V_i . w  (i < nv), V given as nv vectors of length n with stride ldv
 * (VecMDot).  Host pointers. */
int spk_kernel_mdot(spk_ctx *ctx, int64_t n, int32_t nv, const double *V, int64_t ldv,
                    const double *w, double *h);
/* w += sum_i a[i] V_i   (VecMAXPY); returns ||w_new||^2 in *nrm2 if non-NULL. */
int spk_kernel_maxpy(spk_ctx *ctx, int64_t n, int32_t nv, const double *a, const double *V,
                     int64_t ldv, double *w, double *nrm2);
/* Times `reps` launches of the A-block SpMV kernel with HIP events on the
 * context's stream after `warmup` untimed launches; *ms_per_launch = average.
 * x is a deterministic fill sin(0.37 i). */
int spk_time_spmv(spk_ctx *ctx, int warmup, int reps, double *ms_per_launch);
/* Generic form for the other kernels of an iteration (tuning / profiles):
 * which = "spmv" | "spmv_bcsr" | "spmv_bcsr3" | "spmv_dict" | "spmv_acc" | "spmv_ride" | "spmv_gated" | "mult" | "pc" | "mdot" | "maxpy" | "maxpy_nonorm" | "scale" | "wide_dot" |
 * "bt_update"; nv = vectors for mdot/maxpy.  Needs operators (and pc_setup for
 * "pc"); allocates its own scratch vectors. */
int spk_time_kernel(spk_ctx *ctx, const char *which, int nv, int warmup, int reps,
                    double *ms_per_launch);

/* Test hook: launches a small cross-workgroup reduction in which one partial sum is never
 * published.  The reducer gives up after timeout_ms and raises the context's sticky execution-error
 * word; the call (like spk_fgmres / spk_mult when it happens inside them) returns SPK_ERR_HIP -- an
 * execution failure is never reported as a numerical reason (KSP_DIVERGED_NANORINF).  The context
 * re-arms its reduction buffer and stays usable. */
int spk_debug_finish_timeout(spk_ctx *ctx, int timeout_ms);

/* Test hook: the bound of every device-side wait for another workgroup's data (cross-workgroup reductions, the resident
 * cycle kernel's exchanges) in 100 MHz ticks; 0 restores the default (4 s).  A bound of one tick makes the next solve fail
 * in the MIDDLE of a cycle with SPK_ERR_HIP -- what a lost workgroup would cause -- so that tests can check that the context
 * stays usable afterwards. */
int spk_debug_set_wait_bound(spk_ctx *ctx, uint32_t ticks);

/* Measurement hook (bench.py's roofline): with max_launches > 0 the next solves take the kernel's OWN start and stop time
 * stamps (hipExtLaunchKernelGGL) of each product launch of their ITERATIONS (y (+)= A x with the Givens rider in the
 * launch -- the kernel the roofline names) into a pair of HIP events on the solver's stream, up to max_launches of them; 0
 * switches it off.  spk_get_product_timing waits for the stream and returns their count and the mean / median / shortest /
 * longest duration in ms: the kernel as it runs inside a solve -- behind the MAXPY pass, with the caches in the state that
 * pass leaves -- not a batch of back-to-back repetitions; what `rocprofv3 --kernel-trace --stats` averages for it.
 * Launches shorter than half the median (gated off by the device: the iterations enqueued ahead of a solve's end) are left
 * out and reported apart (gated, gated_mean_ms).  (The resident form launches no product per iteration: count 0.) */
int spk_debug_time_products(spk_ctx *ctx, int32_t max_launches);
int spk_get_product_timing(spk_ctx *ctx, int32_t *launches, double *mean_ms, double *median_ms, double *min_ms, double *max_ms,
                           int32_t *gated, double *gated_mean_ms);

/* ---- host-only helpers: row-slab partition and halo plan ------------------ */
/* (callable without a GPU; used by the multi-rank CPU tests) */
/* Rows owned by `rank` of `nranks` when `nlines` grid lines of `line_rows`
 * rows each are dealt in contiguous slabs (PETSc's default DMDA split in y). */
int spk_partition_slab(int64_t nlines, int64_t line_rows, int rank, int nranks,
                       int64_t *row_begin, int64_t *row_end);
/* Splits a local CSR slab with global columns into the diagonal block (local
 * column numbers) and the off-diagonal block (ghost numbers 0..n_ghost-1,
 * ghosts sorted by global column) -- MatMPIAIJ's (Ad, Ao, garray).
 * Call once with the output arrays NULL to get the sizes. */
int spk_partition_split(int64_t row_begin, int32_t nrows_local, const int32_t *rowptr,
                        const int32_t *colidx, const double *val,
                        int32_t *d_rowptr, int32_t *d_colidx, double *d_val,
                        int32_t *o_rowptr, int32_t *o_colidx, double *o_val,
                        int32_t *garray, int64_t *nnz_d, int64_t *nnz_o, int32_t *n_ghost);

#ifdef __cplusplus
}
#endif
#endif /* SPK_H */
