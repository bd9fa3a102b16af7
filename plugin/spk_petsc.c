/*
 * spk_petsc.c -- PETSc glue for libspk.so: the reference-side binding.
 *
 * STATUS: written against the documented public PETSc API (>= 3.7, the version
 * floor of /root/reference/CMakeLists.txt:13); NOT COMPILED AND NOT TESTED --
 * PETSc is absent from the build image and from the GPU box (no network).  It
 * is compiled only where PETSc exists:
 *
 *     mpicc -c plugin/spk_petsc.c $(pkg-config --cflags PETSc) -Iinclude
 *
 * It gives a maintainer of the reference three plug points at the call site
 * /root/reference/src/SaddlePointProblem.c:65-72 (see INTEGRATION.md for the
 * patch), in order of how much of KSPSolve they move onto the GPU:
 *
 *   SpkPCShellAttach(pc, A, B)   -pc_type shell: PETSc keeps KSPSolve_FGMRES and
 *                                MatMult on the host and calls our PCApply.
 *   SpkMatShellCreate(A, B, &K)  MATSHELL whose MatMult is spk_mult.
 *   SpkKSPSolveNative(...)       the whole KSPSolve (FGMRES + PC + MatMult +
 *                                Gram-Schmidt) on the device: spk_fgmres.
 *   SpkKSPRegister()             the same as a registered KSP type: after one call at
 *                                start-up, `-ksp_type spk_fgmres` makes the reference's
 *                                unmodified KSPSolve(ksp, f, u) (SaddlePointProblem.c:70)
 *                                run on the device.  Operators are taken from
 *                                KSPGetOperators: a MATNEST {A, B^T; B, 0} or a plain AIJ A.
 *
 * Data extraction follows SURVEY.md section 8(b): MatGetOwnershipRange +
 * MatGetRow on the rank's rows (works for SeqAIJ and MPIAIJ, global columns --
 * exactly what spk_set_block takes), VecGetArrayRead / VecGetArray for vectors.
 * PetscInt must be 32-bit and PetscScalar real double (asserted).
 */
#include <petscksp.h>
#include <petsc/private/kspimpl.h> /* KSPRegister'd type: ksp->ops, ksp->data */

#include "spk.h"

#define SPK_CHK(ctx, call)                                                                     \
    do {                                                                                       \
        int rc_ = (call);                                                                      \
        if (rc_) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_LIB, "libspk: %s", spk_last_error(ctx));   \
    } while (0)

typedef struct {
    spk_ctx *ctx;
    PetscInt n_local, m;
    double *xbuf, *ybuf; /* [u_local ; lambda] staging when the nest vector is not contiguous */
} SpkGlue;

static PetscErrorCode SpkCheckTypes(void)
{
    PetscFunctionBegin;
    if (sizeof(PetscInt) != 4) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_SUP, "libspk needs 32-bit PetscInt");
    if (sizeof(PetscScalar) != 8) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_SUP, "libspk needs real double PetscScalar");
#if defined(PETSC_USE_COMPLEX)
    SETERRQ(PETSC_COMM_SELF, PETSC_ERR_SUP, "libspk needs a real PETSc build");
#endif
    PetscFunctionReturn(0);
}

/* Copies the rank's rows of an AIJ matrix (global column numbers) into CSR arrays. */
static PetscErrorCode SpkExtractRows(Mat A, PetscInt rstart, PetscInt rend, PetscInt **rowptr, PetscInt **colidx,
                                     PetscScalar **val)
{
    PetscErrorCode ierr;
    PetscInt r, nnz = 0, ncols, k;
    const PetscInt *cols;
    const PetscScalar *vals;

    PetscFunctionBegin;
    ierr = PetscMalloc1(rend - rstart + 1, rowptr); CHKERRQ(ierr);
    (*rowptr)[0] = 0;
    for (r = rstart; r < rend; ++r) {
        ierr = MatGetRow(A, r, &ncols, NULL, NULL); CHKERRQ(ierr);
        nnz += ncols;
        (*rowptr)[r - rstart + 1] = nnz;
        ierr = MatRestoreRow(A, r, &ncols, NULL, NULL); CHKERRQ(ierr);
    }
    ierr = PetscMalloc1(nnz ? nnz : 1, colidx); CHKERRQ(ierr);
    ierr = PetscMalloc1(nnz ? nnz : 1, val); CHKERRQ(ierr);
    for (r = rstart; r < rend; ++r) {
        ierr = MatGetRow(A, r, &ncols, &cols, &vals); CHKERRQ(ierr);
        for (k = 0; k < ncols; ++k) {
            (*colidx)[(*rowptr)[r - rstart] + k] = cols[k];
            (*val)[(*rowptr)[r - rstart] + k] = vals[k];
        }
        ierr = MatRestoreRow(A, r, &ncols, &cols, &vals); CHKERRQ(ierr);
    }
    PetscFunctionReturn(0);
}

/* Uploads A (row slab of this rank) and, when B != NULL, the rank's COLUMN slab of
 * B (all m rows restricted to the columns this rank owns in A).  B is tiny in rows
 * (4 in the reference, SaddlePointProblem.c:49), so every rank extracts it through
 * a redundant sequential copy. */
static PetscErrorCode SpkGlueCreate(MPI_Comm comm, Mat A, Mat B, SpkGlue **out)
{
    PetscErrorCode ierr;
    SpkGlue *g;
    PetscInt rstart, rend, N, *rp, *ci;
    PetscScalar *va;
    PetscMPIInt rank, size;
    unsigned char id[128];

    PetscFunctionBegin;
    ierr = SpkCheckTypes(); CHKERRQ(ierr);
    ierr = PetscNew(&g); CHKERRQ(ierr);
    ierr = MPI_Comm_rank(comm, &rank); CHKERRQ(ierr);
    ierr = MPI_Comm_size(comm, &size); CHKERRQ(ierr);
    {
        /* Which GPU this rank drives: -spk_device <d> when given; otherwise the rank's position among the
         * ranks of its node (MPI_COMM_TYPE_SHARED) -- one process per GPU, as the bench launches them.  A
         * launcher that already restricts visibility (ROCR_VISIBLE_DEVICES=<local rank>) passes -spk_device 0. */
        MPI_Comm node;
        PetscMPIInt local = 0;
        PetscInt dev = -1;
        PetscBool given = PETSC_FALSE;
        ierr = MPI_Comm_split_type(comm, MPI_COMM_TYPE_SHARED, rank, MPI_INFO_NULL, &node); CHKERRQ(ierr);
        ierr = MPI_Comm_rank(node, &local); CHKERRQ(ierr);
        ierr = MPI_Comm_free(&node); CHKERRQ(ierr);
        ierr = PetscOptionsGetInt(NULL, NULL, "-spk_device", &dev, &given); CHKERRQ(ierr);
        if (!given) dev = (PetscInt)local;
        if (spk_create(&g->ctx, (int)dev)) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_LIB, "libspk (device %d): %s", (int)dev, spk_last_error(NULL));
    }
    if (size > 1) {
        if (!rank && spk_comm_unique_id(id)) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_LIB, "libspk: %s", spk_last_error(NULL));
        ierr = MPI_Bcast(id, 128, MPI_BYTE, 0, comm); CHKERRQ(ierr);
        SPK_CHK(g->ctx, spk_comm_init_rccl(g->ctx, rank, size, id));
        /* Krylov all-reduces and halo rows written by the solver's kernels into the peers' HBM
         * over xGMI; stays on RCCL (collectively) when a rank cannot map a peer's window */
        SPK_CHK(g->ctx, spk_comm_enable_peer(g->ctx, NULL));
    }
    ierr = MatGetOwnershipRange(A, &rstart, &rend); CHKERRQ(ierr);
    ierr = MatGetSize(A, &N, NULL); CHKERRQ(ierr);
    ierr = SpkExtractRows(A, rstart, rend, &rp, &ci, &va); CHKERRQ(ierr);
    SPK_CHK(g->ctx, spk_set_block(g->ctx, SPK_BLOCK_A00, rstart, rend - rstart, N, rp, ci, va));
    ierr = PetscFree(rp); CHKERRQ(ierr);
    ierr = PetscFree(ci); CHKERRQ(ierr);
    ierr = PetscFree(va); CHKERRQ(ierr);
    g->n_local = rend - rstart;
    g->m = 0;
    if (B) {
        Mat Bseq, *sub;
        IS allrows, mycols;
        PetscInt m, k, r, keep = 0, *rp2;
        ierr = MatGetSize(B, &m, NULL); CHKERRQ(ierr);
        ierr = ISCreateStride(PETSC_COMM_SELF, m, 0, 1, &allrows); CHKERRQ(ierr);
        ierr = ISCreateStride(PETSC_COMM_SELF, rend - rstart, rstart, 1, &mycols); CHKERRQ(ierr);
        ierr = MatCreateSubMatrices(B, 1, &allrows, &mycols, MAT_INITIAL_MATRIX, &sub); CHKERRQ(ierr);
        Bseq = sub[0];
        ierr = SpkExtractRows(Bseq, 0, m, &rp, &ci, &va); CHKERRQ(ierr);
        for (k = 0; k < rp[m]; ++k) ci[k] += rstart; /* back to global column numbers */
        (void)r; (void)keep; (void)rp2;
        SPK_CHK(g->ctx, spk_set_block(g->ctx, SPK_BLOCK_A10, 0, m, N, rp, ci, va));
        ierr = PetscFree(rp); CHKERRQ(ierr);
        ierr = PetscFree(ci); CHKERRQ(ierr);
        ierr = PetscFree(va); CHKERRQ(ierr);
        ierr = MatDestroySubMatrices(1, &sub); CHKERRQ(ierr);
        ierr = ISDestroy(&allrows); CHKERRQ(ierr);
        ierr = ISDestroy(&mycols); CHKERRQ(ierr);
        g->m = m;
    }
    ierr = PetscMalloc2(g->n_local + g->m, &g->xbuf, g->n_local + g->m, &g->ybuf); CHKERRQ(ierr);
    *out = g;
    PetscFunctionReturn(0);
}

static PetscErrorCode SpkGlueDestroy(SpkGlue *g)
{
    PetscErrorCode ierr;
    PetscFunctionBegin;
    if (!g) PetscFunctionReturn(0);
    spk_destroy(g->ctx);
    ierr = PetscFree2(g->xbuf, g->ybuf); CHKERRQ(ierr);
    ierr = PetscFree(g); CHKERRQ(ierr);
    PetscFunctionReturn(0);
}

/* The nest vector [u ; lambda]: u is distributed like A's rows, lambda (m values)
 * lives on rank 0 in PETSc's layout; libspk wants lambda replicated. */
static PetscErrorCode SpkGather(SpkGlue *g, Vec x, double *buf)
{
    PetscErrorCode ierr;
    const PetscScalar *a;
    PetscInt nloc;
    MPI_Comm comm;
    PetscMPIInt rank;

    PetscFunctionBegin;
    ierr = PetscObjectGetComm((PetscObject)x, &comm); CHKERRQ(ierr);
    ierr = MPI_Comm_rank(comm, &rank); CHKERRQ(ierr);
    ierr = VecGetLocalSize(x, &nloc); CHKERRQ(ierr);
    ierr = VecGetArrayRead(x, &a); CHKERRQ(ierr);
    ierr = PetscMemcpy(buf, a, sizeof(double) * g->n_local); CHKERRQ(ierr);
    if (g->m) {
        if (!rank) { ierr = PetscMemcpy(buf + g->n_local, a + g->n_local, sizeof(double) * g->m); CHKERRQ(ierr); }
        ierr = MPI_Bcast(buf + g->n_local, g->m, MPI_DOUBLE, 0, comm); CHKERRQ(ierr);
    }
    ierr = VecRestoreArrayRead(x, &a); CHKERRQ(ierr);
    PetscFunctionReturn(0);
}

static PetscErrorCode SpkScatter(SpkGlue *g, const double *buf, Vec y)
{
    PetscErrorCode ierr;
    PetscScalar *a;
    PetscInt nloc;

    PetscFunctionBegin;
    ierr = VecGetLocalSize(y, &nloc); CHKERRQ(ierr);
    ierr = VecGetArray(y, &a); CHKERRQ(ierr);
    ierr = PetscMemcpy(a, buf, sizeof(double) * nloc); CHKERRQ(ierr); /* rank 0 also takes lambda */
    ierr = VecRestoreArray(y, &a); CHKERRQ(ierr);
    PetscFunctionReturn(0);
}

/* ------------------------------------------------------------------ PCSHELL */
static PetscErrorCode SpkPCApply(PC pc, Vec x, Vec y)
{
    PetscErrorCode ierr;
    SpkGlue *g;
    PetscFunctionBegin;
    ierr = PCShellGetContext(pc, (void **)&g); CHKERRQ(ierr);
    ierr = SpkGather(g, x, g->xbuf); CHKERRQ(ierr);
    SPK_CHK(g->ctx, spk_pc_apply(g->ctx, g->xbuf, g->ybuf, SPK_MEM_HOST));
    ierr = SpkScatter(g, g->ybuf, y); CHKERRQ(ierr);
    PetscFunctionReturn(0);
}

static PetscErrorCode SpkPCDestroy(PC pc)
{
    PetscErrorCode ierr;
    SpkGlue *g;
    PetscFunctionBegin;
    ierr = PCShellGetContext(pc, (void **)&g); CHKERRQ(ierr);
    ierr = SpkGlueDestroy(g); CHKERRQ(ierr);
    PetscFunctionReturn(0);
}

/* pc_type: SPK_PC_JACOBI or SPK_PC_SCHUR; schur_fact: SPK_SCHUR_* */
PetscErrorCode SpkPCShellAttach(PC pc, Mat A, Mat B, int pc_type, int schur_fact)
{
    PetscErrorCode ierr;
    SpkGlue *g;
    MPI_Comm comm;
    PetscFunctionBegin;
    ierr = PetscObjectGetComm((PetscObject)pc, &comm); CHKERRQ(ierr);
    ierr = SpkGlueCreate(comm, A, B, &g); CHKERRQ(ierr);
    SPK_CHK(g->ctx, spk_pc_setup(g->ctx, pc_type, schur_fact));
    ierr = PCSetType(pc, PCSHELL); CHKERRQ(ierr);
    ierr = PCShellSetContext(pc, g); CHKERRQ(ierr);
    ierr = PCShellSetApply(pc, SpkPCApply); CHKERRQ(ierr);
    ierr = PCShellSetDestroy(pc, SpkPCDestroy); CHKERRQ(ierr);
    ierr = PCShellSetName(pc, "libspk MI355X block preconditioner"); CHKERRQ(ierr);
    PetscFunctionReturn(0);
}

/* ----------------------------------------------------------------- MATSHELL */
static PetscErrorCode SpkMatMult(Mat K, Vec x, Vec y)
{
    PetscErrorCode ierr;
    SpkGlue *g;
    PetscFunctionBegin;
    ierr = MatShellGetContext(K, (void **)&g); CHKERRQ(ierr);
    ierr = SpkGather(g, x, g->xbuf); CHKERRQ(ierr);
    SPK_CHK(g->ctx, spk_mult(g->ctx, g->xbuf, g->ybuf, SPK_MEM_HOST));
    ierr = SpkScatter(g, g->ybuf, y); CHKERRQ(ierr);
    PetscFunctionReturn(0);
}

static PetscErrorCode SpkMatDestroy(Mat K)
{
    PetscErrorCode ierr;
    SpkGlue *g;
    PetscFunctionBegin;
    ierr = MatShellGetContext(K, (void **)&g); CHKERRQ(ierr);
    ierr = SpkGlueDestroy(g); CHKERRQ(ierr);
    PetscFunctionReturn(0);
}

PetscErrorCode SpkMatShellCreate(Mat A, Mat B, Mat *K)
{
    PetscErrorCode ierr;
    SpkGlue *g;
    MPI_Comm comm;
    PetscMPIInt rank;
    PetscInt nloc, N, m = 0;
    PetscFunctionBegin;
    ierr = PetscObjectGetComm((PetscObject)A, &comm); CHKERRQ(ierr);
    ierr = MPI_Comm_rank(comm, &rank); CHKERRQ(ierr);
    ierr = SpkGlueCreate(comm, A, B, &g); CHKERRQ(ierr);
    ierr = MatGetLocalSize(A, &nloc, NULL); CHKERRQ(ierr);
    ierr = MatGetSize(A, &N, NULL); CHKERRQ(ierr);
    if (B) { ierr = MatGetSize(B, &m, NULL); CHKERRQ(ierr); }
    ierr = MatCreateShell(comm, nloc + (rank ? 0 : m), nloc + (rank ? 0 : m), N + m, N + m, g, K); CHKERRQ(ierr);
    ierr = MatShellSetOperation(*K, MATOP_MULT, (void (*)(void))SpkMatMult); CHKERRQ(ierr);
    ierr = MatShellSetOperation(*K, MATOP_DESTROY, (void (*)(void))SpkMatDestroy); CHKERRQ(ierr);
    PetscFunctionReturn(0);
}

/* --------------------------------------------------- whole KSPSolve on device */
/* Reads the solver options the reference exposes through KSPSetFromOptions
 * (SaddlePointProblem.c:67) from the PETSc options database and runs spk_fgmres.
 * b, x: nest vectors [f ; g], [u ; lambda]. */
PetscErrorCode SpkKSPSolveNative(MPI_Comm comm, Mat A, Mat B, Vec b, Vec x, PetscInt *its, KSPConvergedReason *reason)
{
    PetscErrorCode ierr;
    SpkGlue *g;
    spk_opts o;
    spk_result res;
    PetscReal rtol = 1e-5, atol = 1e-50, dtol = 1e4;
    PetscInt maxit = 10000, restart = 30;
    PetscBool nz = PETSC_FALSE;
    char fact[32] = "full";
    int sf = SPK_SCHUR_FULL;

    PetscFunctionBegin;
    ierr = PetscOptionsGetReal(NULL, NULL, "-ksp_rtol", &rtol, NULL); CHKERRQ(ierr);
    ierr = PetscOptionsGetReal(NULL, NULL, "-ksp_atol", &atol, NULL); CHKERRQ(ierr);
    ierr = PetscOptionsGetReal(NULL, NULL, "-ksp_divtol", &dtol, NULL); CHKERRQ(ierr);
    ierr = PetscOptionsGetInt(NULL, NULL, "-ksp_max_it", &maxit, NULL); CHKERRQ(ierr);
    ierr = PetscOptionsGetInt(NULL, NULL, "-ksp_gmres_restart", &restart, NULL); CHKERRQ(ierr);
    ierr = PetscOptionsGetBool(NULL, NULL, "-ksp_initial_guess_nonzero", &nz, NULL); CHKERRQ(ierr);
    ierr = PetscOptionsGetString(NULL, NULL, "-pc_fieldsplit_schur_fact_type", fact, sizeof fact, NULL); CHKERRQ(ierr);
    if (!strcmp(fact, "diag")) sf = SPK_SCHUR_DIAG;
    else if (!strcmp(fact, "lower")) sf = SPK_SCHUR_LOWER;
    else if (!strcmp(fact, "upper")) sf = SPK_SCHUR_UPPER;

    ierr = SpkGlueCreate(comm, A, B, &g); CHKERRQ(ierr);
    SPK_CHK(g->ctx, spk_pc_setup(g->ctx, B ? SPK_PC_SCHUR : SPK_PC_JACOBI, sf));
    spk_default_opts(&o);
    o.rtol = rtol; o.abstol = atol; o.dtol = dtol; o.max_it = maxit; o.restart = restart; o.guess_nonzero = nz;
    ierr = SpkGather(g, b, g->xbuf); CHKERRQ(ierr);
    if (nz) { ierr = SpkGather(g, x, g->ybuf); CHKERRQ(ierr); }
    SPK_CHK(g->ctx, spk_fgmres(g->ctx, g->xbuf, g->ybuf, SPK_MEM_HOST, &o, &res, NULL, 0));
    ierr = SpkScatter(g, g->ybuf, x); CHKERRQ(ierr);
    if (its) *its = res.its;
    if (reason) *reason = (KSPConvergedReason)res.reason; /* same numbering */
    ierr = SpkGlueDestroy(g); CHKERRQ(ierr);
    PetscFunctionReturn(0);
}

/* ------------------------------------------------- registered KSP type "spk_fgmres" */
typedef struct {
    SpkGlue *glue;
    Mat A, B; /* borrowed from the operator */
    int schur_fact;
} KSP_SPK;

static PetscErrorCode KSPSetUp_SPK(KSP ksp)
{
    PetscErrorCode ierr;
    KSP_SPK *d = (KSP_SPK *)ksp->data;
    Mat Amat, Pmat;
    PetscBool isnest;
    MPI_Comm comm;

    PetscFunctionBegin;
    ierr = PetscObjectGetComm((PetscObject)ksp, &comm); CHKERRQ(ierr);
    ierr = KSPGetOperators(ksp, &Amat, &Pmat); CHKERRQ(ierr);
    ierr = PetscObjectTypeCompare((PetscObject)Amat, MATNEST, &isnest); CHKERRQ(ierr);
    if (isnest) { /* {A, B^T; B, 0}: blocks (0,0) and (1,0) */
        ierr = MatNestGetSubMat(Amat, 0, 0, &d->A); CHKERRQ(ierr);
        ierr = MatNestGetSubMat(Amat, 1, 0, &d->B); CHKERRQ(ierr);
    } else {      /* as written in the reference: KSPSetOperators(ksp, A, A), :66 */
        d->A = Amat;
        d->B = NULL;
    }
    if (d->glue) { ierr = SpkGlueDestroy(d->glue); CHKERRQ(ierr); d->glue = NULL; }
    ierr = SpkGlueCreate(comm, d->A, d->B, &d->glue); CHKERRQ(ierr);
    SPK_CHK(d->glue->ctx, spk_pc_setup(d->glue->ctx, d->B ? SPK_PC_SCHUR : SPK_PC_JACOBI, d->schur_fact));
    PetscFunctionReturn(0);
}

static PetscErrorCode KSPSolve_SPK(KSP ksp)
{
    PetscErrorCode ierr;
    KSP_SPK *d = (KSP_SPK *)ksp->data;
    spk_opts o;
    spk_result res;

    PetscFunctionBegin;
    spk_default_opts(&o);
    o.rtol = ksp->rtol;          /* -ksp_rtol   */
    o.abstol = ksp->abstol;      /* -ksp_atol   */
    o.dtol = ksp->divtol;        /* -ksp_divtol */
    o.max_it = ksp->max_it;      /* -ksp_max_it */
    o.guess_nonzero = ksp->guess_zero ? 0 : 1;
    ierr = PetscOptionsGetInt(((PetscObject)ksp)->options, ((PetscObject)ksp)->prefix, "-ksp_gmres_restart", &o.restart, NULL); CHKERRQ(ierr);
    {   /* private option: one reduction and three launches per iteration (see spk_opts.single_reduce) */
        PetscInt sr = 0;
        ierr = PetscOptionsGetInt(((PetscObject)ksp)->options, ((PetscObject)ksp)->prefix, "-spk_single_reduce", &sr, NULL); CHKERRQ(ierr);
        o.single_reduce = (int32_t)sr;
    }
    {   /* private option: how an iteration is launched (spk_opts.iteration_form; 0 = automatic) */
        PetscInt form = 0;
        ierr = PetscOptionsGetInt(((PetscObject)ksp)->options, ((PetscObject)ksp)->prefix, "-spk_iteration_form", &form, NULL); CHKERRQ(ierr);
        o.iteration_form = (int32_t)form;
    }
    ierr = SpkGather(d->glue, ksp->vec_rhs, d->glue->xbuf); CHKERRQ(ierr);
    if (o.guess_nonzero) { ierr = SpkGather(d->glue, ksp->vec_sol, d->glue->ybuf); CHKERRQ(ierr); }
    SPK_CHK(d->glue->ctx, spk_fgmres(d->glue->ctx, d->glue->xbuf, d->glue->ybuf, SPK_MEM_HOST, &o, &res, NULL, 0));
    ierr = SpkScatter(d->glue, d->glue->ybuf, ksp->vec_sol); CHKERRQ(ierr);
    ksp->its = res.its;
    ksp->rnorm = res.rnorm;
    ksp->reason = (KSPConvergedReason)res.reason; /* same numbering as KSPConvergedReason */
    PetscFunctionReturn(0);
}

static PetscErrorCode KSPSetFromOptions_SPK(PetscOptionItems *PetscOptionsObject, KSP ksp)
{
    PetscErrorCode ierr;
    KSP_SPK *d = (KSP_SPK *)ksp->data;
    char fact[32] = "full";
    PetscBool set;

    PetscFunctionBegin;
    (void)PetscOptionsObject;
    ierr = PetscOptionsGetString(((PetscObject)ksp)->options, ((PetscObject)ksp)->prefix, "-pc_fieldsplit_schur_fact_type", fact, sizeof fact, &set); CHKERRQ(ierr);
    if (set) {
        if (!strcmp(fact, "diag")) d->schur_fact = SPK_SCHUR_DIAG;
        else if (!strcmp(fact, "lower")) d->schur_fact = SPK_SCHUR_LOWER;
        else if (!strcmp(fact, "upper")) d->schur_fact = SPK_SCHUR_UPPER;
        else d->schur_fact = SPK_SCHUR_FULL;
    }
    PetscFunctionReturn(0);
}

static PetscErrorCode KSPDestroy_SPK(KSP ksp)
{
    PetscErrorCode ierr;
    KSP_SPK *d = (KSP_SPK *)ksp->data;
    PetscFunctionBegin;
    if (d->glue) { ierr = SpkGlueDestroy(d->glue); CHKERRQ(ierr); }
    ierr = PetscFree(ksp->data); CHKERRQ(ierr);
    PetscFunctionReturn(0);
}

static PetscErrorCode KSPCreate_SPK(KSP ksp)
{
    PetscErrorCode ierr;
    KSP_SPK *d;
    PetscFunctionBegin;
    ierr = PetscNew(&d); CHKERRQ(ierr);
    d->schur_fact = SPK_SCHUR_FULL; /* PETSc's default -pc_fieldsplit_schur_fact_type */
    ksp->data = (void *)d;
    ierr = KSPSetSupportedNorm(ksp, KSP_NORM_UNPRECONDITIONED, PC_RIGHT, 3); CHKERRQ(ierr); /* FGMRES: right PC */
    ksp->ops->setup = KSPSetUp_SPK;
    ksp->ops->solve = KSPSolve_SPK;
    ksp->ops->destroy = KSPDestroy_SPK;
    ksp->ops->setfromoptions = KSPSetFromOptions_SPK;
    ksp->ops->view = NULL;
    ksp->ops->buildsolution = KSPBuildSolutionDefault;
    ksp->ops->buildresidual = KSPBuildResidualDefault;
    PetscFunctionReturn(0);
}

/* Call once after PetscInitialize (main.c:12); then the reference's call site works
 * unchanged with  -ksp_type spk_fgmres [-pc_fieldsplit_schur_fact_type full] ... */
PetscErrorCode SpkKSPRegister(void)
{
    PetscErrorCode ierr;
    PetscFunctionBegin;
    ierr = KSPRegister("spk_fgmres", KSPCreate_SPK); CHKERRQ(ierr);
    PetscFunctionReturn(0);
}
