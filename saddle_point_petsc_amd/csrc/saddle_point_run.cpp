// saddle_point_run.cpp -- stand-alone driver with the shape of the reference's
// program (/root/reference/src/main.c:7-19 -> SolveSaddlePointProblem,
// SaddlePointProblem.c:8-25 -> SolveConstraintLaplaceProblem, :34-76):
// set up the grid, assemble A, f (and the build-defined B, g), apply the
// boundary conditions, then the six solver calls of :65-72 through the
// KSP-shaped facade, options PETSc-style on the command line:
//
//   saddle_point_run -da_grid_x 257 -da_grid_y 257 -ksp_type fgmres -ksp_rtol 1e-8 \
//       -pc_type fieldsplit -pc_fieldsplit_type schur -pc_fieldsplit_schur_fact_type full \
//       -ksp_converged_reason [-saddle 0] [-solution_view] [-no_vtk]
//
// The reference hard-codes Nx = Ny = 3 elements (main.c:14), i.e. a 4 x 4 node
// grid; that is the default here too.  -saddle 0 solves A u = f alone, as the
// reference does at HEAD (KSPSetOperators(ksp, A, A), :66).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/spk_assembly.h"
#include "../../include/spk_ksp.h"

static int opt_int(int argc, char **argv, const char *name, int dflt)
{
    for (int i = 1; i + 1 < argc; ++i)
        if (!std::strcmp(argv[i], name)) return std::atoi(argv[i + 1]);
    return dflt;
}
static bool opt_flag(int argc, char **argv, const char *name)
{
    for (int i = 1; i < argc; ++i)
        if (!std::strcmp(argv[i], name)) return true;
    return false;
}

#define CHK(call)                                                             \
    do {                                                                      \
        int rc_ = (call);                                                     \
        if (rc_) {                                                            \
            std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ksp ? SpkKSPGetError(ksp) : ""); \
            return 1;                                                         \
        }                                                                     \
    } while (0)

int main(int argc, char **argv)
{
    SpkKSP ksp = nullptr;
    const int mx = opt_int(argc, argv, "-da_grid_x", 4), my = opt_int(argc, argv, "-da_grid_y", 4);
    const bool saddle = opt_int(argc, argv, "-saddle", 1) != 0;
    int64_t n = 0, nnz = 0;
    CHK(SpkAssemblySizes(mx, my, &n, &nnz));

    // SetupDMDA + AssembleOperator_Laplace + AssembleRHS_Laplace + ApplyBC_Laplace
    std::vector<int32_t> rowptr((size_t)n + 1), colidx((size_t)nnz);
    std::vector<double> val((size_t)nnz), rhs((size_t)n + 4, 0.0), sol((size_t)n + 4, 0.0);
    CHK(SpkAssembleOperator_Laplace(mx, my, 0, n, rowptr.data(), colidx.data(), val.data(), rhs.data(), 1, 0));
    SpkMatCSR A = {0, (int32_t)n, 0, n, rowptr.data(), colidx.data(), val.data()};

    // AssembleOperator_Constraints + AssembleRHS_Constraints (stubs in the reference)
    std::vector<int32_t> brp(5), bci;
    std::vector<double> bv;
    SpkMatCSR B = {0, 4, 0, n, nullptr, nullptr, nullptr};
    if (saddle) {
        const int64_t bn = SpkConstraintsSlabNnz(mx, my, 0, n);
        if (bn < 0) { std::fprintf(stderr, "grid too small for the constraint block\n"); return 1; }
        bci.resize((size_t)bn);
        bv.resize((size_t)bn);
        CHK(SpkAssembleOperator_Constraints(mx, my, 0, n, brp.data(), bci.data(), bv.data()));
        CHK(SpkAssembleRHS_Constraints(rhs.data() + n));
        B.rowptr = brp.data(); B.colidx = bci.data(); B.val = bv.data();
    }

    // the call site SaddlePointProblem.c:65-72
    CHK(SpkKSPCreate(opt_int(argc, argv, "-spk_device", 0), &ksp));
    CHK(SpkKSPSetOperators(ksp, &A, saddle ? &B : nullptr));
    CHK(SpkKSPSetFromOptions(ksp, argc - 1, argv + 1));
    CHK(SpkKSPSetUp(ksp));
    CHK(SpkKSPSolve(ksp, rhs.data(), sol.data()));

    int32_t its = 0, reason = 0;
    double rnorm = 0.0, secs = 0.0;
    SpkKSPGetIterationNumber(ksp, &its);
    SpkKSPGetConvergedReason(ksp, &reason);
    SpkKSPGetResidualNorm(ksp, &rnorm);
    SpkKSPGetSolveTime(ksp, &secs);
    std::printf("grid %d x %d nodes, %lld rows%s: %s after %d iterations, residual %.6e, solve %.3f ms\n", mx, my,
                (long long)n, saddle ? " + 4 multipliers" : "", SpkKSPConvergedReasonName(reason), its, rnorm, secs * 1e3);
    if (opt_flag(argc, argv, "-solution_view")) {  // VecViewFromOptions(u, NULL, "-solution_view"), :20
        for (int64_t i = 0; i < n + (saddle ? 4 : 0); ++i) std::printf("%.15e\n", sol[(size_t)i]);
    }
    // WriteVTK(da_u, u, "test.vtk"), SaddlePointProblem.c:22 -- with the field this time
    if (!opt_flag(argc, argv, "-no_vtk")) CHK(SpkWriteVTK(mx, my, sol.data(), "test.vtk"));
    CHK(SpkKSPDestroy(&ksp));
    return reason > 0 ? 0 : 2;
}
