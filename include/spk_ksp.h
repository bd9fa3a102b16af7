/*
 * spk_ksp.h -- host-side mirror of the reference's solver call site.
 *
 * /root/reference/src/SaddlePointProblem.c:65-72 is
 *     KSPCreate(PETSC_COMM_WORLD,&ksp); KSPSetOperators(ksp,A,A);
 *     KSPSetFromOptions(ksp); KSPSetUp(ksp); KSPSolve(ksp,f,*u); KSPDestroy(&ksp);
 * The functions below keep those names (Spk prefix), argument meaning and
 * error behaviour (int error code, 0 = success, non-convergence is a reason,
 * not an error) on plain CSR arrays instead of Mat/Vec, and read the same
 * option names KSPSetFromOptions reads from the PETSc options database (:67).
 * They sit on top of the C ABI in spk.h; nothing here touches the GPU directly.
 */
#ifndef SPK_KSP_H
#define SPK_KSP_H
#include <stdint.h>
#include "spk.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct SpkKSP_s *SpkKSP;

/* A CSR row slab with GLOBAL column indices (what MatMPIAIJ rows look like);
 * single rank: row_begin = 0, nrows_local = ncols_global. */
typedef struct SpkMatCSR {
    int64_t row_begin;
    int32_t nrows_local;
    int32_t pad;
    int64_t ncols_global;
    const int32_t *rowptr, *colidx;
    const double *val;
} SpkMatCSR;

int SpkKSPCreate(int device, SpkKSP *ksp);                                 /* KSPCreate      :65 */
/* optional, before SetOperators: one process per GPU over RCCL */
int SpkKSPSetCommRCCL(SpkKSP ksp, int rank, int nranks, const void *id128);
/* A = (0,0) block (operator and preconditioning matrix, as KSPSetOperators(ksp,A,A));
 * B = (1,0) block of the nest sketched at :45-60, or NULL for the as-written A-only solve */
int SpkKSPSetOperators(SpkKSP ksp, const SpkMatCSR *A, const SpkMatCSR *B); /* KSPSetOperators :66 */
/* argv-style option list, e.g. {"-ksp_type","fgmres","-ksp_rtol","1e-8",
 * "-pc_type","fieldsplit","-pc_fieldsplit_type","schur",
 * "-pc_fieldsplit_schur_fact_type","full"}.  Unknown -ksp_/-pc_/-fieldsplit_
 * options are an error; other options are ignored (as PETSc leaves them unused).
 * DEVIATION from PETSc, on purpose: -ksp_type and -pc_type have NO default here.  PETSc would fall
 * back to gmres (left preconditioning) and ilu (bjacobi+ilu in parallel), neither of which this
 * library implements; SpkKSPSetUp / SpkKSPSolve return SPK_ERR_UNSUPPORTED with a message unless
 * "-ksp_type fgmres" and "-pc_type jacobi|fieldsplit|none" were given. */
int SpkKSPSetFromOptions(SpkKSP ksp, int argc, const char *const *argv);    /* KSPSetFromOptions :67 */
int SpkKSPSetUp(SpkKSP ksp);                                                /* KSPSetUp       :68 */
/* b, x: host vectors of n_local + m values ([u ; lambda]) */
int SpkKSPSolve(SpkKSP ksp, const double *b, double *x);                    /* KSPSolve       :70 */
int SpkKSPDestroy(SpkKSP *ksp);                                             /* KSPDestroy     :72 */

int SpkKSPGetIterationNumber(SpkKSP ksp, int32_t *its);
int SpkKSPGetConvergedReason(SpkKSP ksp, int32_t *reason);
int SpkKSPGetResidualNorm(SpkKSP ksp, double *rnorm);
int SpkKSPGetResidualHistory(SpkKSP ksp, const double **hist, int32_t *n);
int SpkKSPGetSolveTime(SpkKSP ksp, double *seconds);
int SpkKSPGetOptions(SpkKSP ksp, spk_opts *opts, int32_t *pc_type, int32_t *schur_fact);
int SpkKSPGetContext(SpkKSP ksp, spk_ctx **ctx);
const char *SpkKSPGetError(SpkKSP ksp);
const char *SpkKSPConvergedReasonName(int32_t reason);

#ifdef __cplusplus
}
#endif
#endif
