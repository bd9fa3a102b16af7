// spk_ksp.cpp -- KSP-shaped host facade over the C ABI (see include/spk_ksp.h):
// the call sequence of /root/reference/src/SaddlePointProblem.c:65-72 with the
// option names KSPSetFromOptions (:67) would read.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/spk_ksp.h"

struct SpkKSP_s {
    spk_ctx *ctx = nullptr;  // created on first use so that option handling needs no GPU
    int device = 0;
    spk_opts opts;
    int32_t pc_type = SPK_PC_NONE, schur_fact = SPK_SCHUR_FULL;
    int32_t inner_sweeps = 0;  // -fieldsplit_0_ksp_max_it with -fieldsplit_0_ksp_type richardson
    double inner_omega = 1.0;  // -fieldsplit_0_ksp_richardson_scale
    bool inner_richardson = false;
    bool have_ops = false, is_setup = false, has_B = false;
    // PETSc's own defaults (-ksp_type gmres with left preconditioning, -pc_type ilu / bjacobi+ilu) are
    // not implemented here: a run that leaves them unset must be refused, not silently changed
    bool ksp_type_given = false, pc_type_given = false;
    bool monitor = false, print_reason = false, view = false;
    spk_result result;
    std::vector<double> history;
    std::string err;
};

namespace {
int set_err(SpkKSP k, int code, const std::string &m)
{
    k->err = m;
    return code;
}
int from_ctx(SpkKSP k, int code)
{
    if (code != SPK_OK) k->err = spk_last_error(k->ctx);
    return code;
}
bool parse_double(const char *s, double *v)
{
    char *e = nullptr;
    *v = std::strtod(s, &e);
    return e && e != s && *e == 0;
}
bool parse_int(const char *s, int32_t *v)
{
    char *e = nullptr;
    const long l = std::strtol(s, &e, 10);
    *v = (int32_t)l;
    return e && e != s && *e == 0;
}
bool parse_bool(const char *s, bool *v)
{
    const std::string t(s);
    if (t == "1" || t == "true" || t == "yes" || t == "on") { *v = true; return true; }
    if (t == "0" || t == "false" || t == "no" || t == "off") { *v = false; return true; }
    return false;
}
}  // namespace

extern "C" {

const char *SpkKSPConvergedReasonName(int32_t r)
{
    switch (r) {
    case SPK_CONVERGED_RTOL: return "CONVERGED_RTOL";
    case SPK_CONVERGED_ATOL: return "CONVERGED_ATOL";
    case SPK_CONVERGED_ITS: return "CONVERGED_ITS";
    case SPK_CONVERGED_HAPPY_BREAKDOWN: return "CONVERGED_HAPPY_BREAKDOWN";
    case SPK_DIVERGED_NULL: return "DIVERGED_NULL";
    case SPK_DIVERGED_ITS: return "DIVERGED_ITS";
    case SPK_DIVERGED_DTOL: return "DIVERGED_DTOL";
    case SPK_DIVERGED_BREAKDOWN: return "DIVERGED_BREAKDOWN";
    case SPK_DIVERGED_NANORINF: return "DIVERGED_NANORINF";
    case SPK_ITERATING: return "CONVERGED_ITERATING";
    default: return "UNKNOWN";
    }
}

int SpkKSPCreate(int device, SpkKSP *out)
{
    if (!out) return SPK_ERR_ARG;
    *out = nullptr;
    SpkKSP k = new SpkKSP_s();
    spk_default_opts(&k->opts);
    std::memset(&k->result, 0, sizeof k->result);
    k->device = device;
    *out = k;
    return SPK_OK;
}

static int ensure_ctx(SpkKSP k)
{
    if (k->ctx) return SPK_OK;
    const int rc = spk_create(&k->ctx, k->device);
    if (rc != SPK_OK) k->err = spk_last_error(nullptr);
    return rc;
}

int SpkKSPDestroy(SpkKSP *k)
{
    if (!k || !*k) return SPK_OK;
    if ((*k)->ctx) spk_destroy((*k)->ctx);
    delete *k;
    *k = nullptr;
    return SPK_OK;
}

const char *SpkKSPGetError(SpkKSP k) { return k ? k->err.c_str() : "null KSP"; }

int SpkKSPSetCommRCCL(SpkKSP k, int rank, int nranks, const void *id128)
{
    if (!k) return SPK_ERR_ARG;
    if (const int rc = ensure_ctx(k)) return rc;
    if (const int rc = spk_comm_init_rccl(k->ctx, rank, nranks, id128)) return from_ctx(k, rc);
    // peer-store collectives on top (falls back to RCCL collectively; see include/spk.h)
    return from_ctx(k, spk_comm_enable_peer(k->ctx, nullptr));
}

int SpkKSPSetOperators(SpkKSP k, const SpkMatCSR *A, const SpkMatCSR *B)
{
    if (!k) return SPK_ERR_ARG;
    if (!A) return set_err(k, SPK_ERR_ARG, "KSPSetOperators: null operator");
    int rc = ensure_ctx(k);
    if (rc != SPK_OK) return rc;
    rc = spk_set_block(k->ctx, SPK_BLOCK_A00, A->row_begin, A->nrows_local, A->ncols_global, A->rowptr, A->colidx, A->val);
    if (rc != SPK_OK) return from_ctx(k, rc);
    k->has_B = false;
    if (B) {
        rc = spk_set_block(k->ctx, SPK_BLOCK_A10, 0, B->nrows_local, B->ncols_global, B->rowptr, B->colidx, B->val);
        if (rc != SPK_OK) return from_ctx(k, rc);
        k->has_B = B->nrows_local > 0;
    }
    k->have_ops = true;
    k->is_setup = false;
    return SPK_OK;
}

int SpkKSPSetFromOptions(SpkKSP k, int argc, const char *const *argv)
{
    if (!k) return SPK_ERR_ARG;
    if (argc > 0 && !argv) return set_err(k, SPK_ERR_ARG, "KSPSetFromOptions: null argv");
    for (int i = 0; i < argc; ++i) {
        const std::string key = argv[i] ? argv[i] : "";
        if (key.empty() || key[0] != '-') continue;
        const bool has_val = (i + 1 < argc) && argv[i + 1] && !(argv[i + 1][0] == '-' && !(argv[i + 1][1] >= '0' && argv[i + 1][1] <= '9') && argv[i + 1][1] != '.');
        const char *val = has_val ? argv[i + 1] : nullptr;
        auto need = [&](const char *what) -> int {
            return set_err(k, SPK_ERR_ARG, "option " + key + " needs " + what);
        };
        auto bad = [&]() -> int { return set_err(k, SPK_ERR_UNSUPPORTED, "option " + key + " " + (val ? val : "") + " is not supported"); };
        bool flag = true;
        if (key == "-ksp_type") {
            if (!val) return need("a type");
            if (std::string(val) != "fgmres") return bad();
            k->ksp_type_given = true;
        } else if (key == "-ksp_rtol") {
            if (!val || !parse_double(val, &k->opts.rtol)) return need("a real");
        } else if (key == "-ksp_atol") {
            if (!val || !parse_double(val, &k->opts.abstol)) return need("a real");
        } else if (key == "-ksp_divtol") {
            if (!val || !parse_double(val, &k->opts.dtol)) return need("a real");
        } else if (key == "-ksp_max_it") {
            if (!val || !parse_int(val, &k->opts.max_it)) return need("an integer");
        } else if (key == "-ksp_gmres_restart") {
            if (!val || !parse_int(val, &k->opts.restart)) return need("an integer");
        } else if (key == "-ksp_initial_guess_nonzero") {
            if (val && !parse_bool(val, &flag)) return need("a boolean");
            k->opts.guess_nonzero = flag;
        } else if (key == "-ksp_gmres_classicalgramschmidt") {
            k->opts.orthog = SPK_ORTHOG_CGS;
        } else if (key == "-ksp_gmres_modifiedgramschmidt") {
            k->opts.orthog = SPK_ORTHOG_MGS;
        } else if (key == "-ksp_gmres_cgs_refinement_type") {
            if (!val) return need("a type");
            const std::string v(val);
            if (v == "never" || v == "refine_never") k->opts.cgs_refine = SPK_REFINE_NEVER;
            else if (v == "ifneeded" || v == "refine_ifneeded") k->opts.cgs_refine = SPK_REFINE_IFNEEDED;
            else if (v == "always" || v == "refine_always") k->opts.cgs_refine = SPK_REFINE_ALWAYS;
            else return bad();
        } else if (key == "-ksp_pc_side") {
            if (!val) return need("a side");
            if (std::string(val) != "right") return bad();
        } else if (key == "-ksp_norm_type") {
            if (!val) return need("a type");
            if (std::string(val) != "unpreconditioned") return bad();
        } else if (key == "-ksp_monitor" || key == "-ksp_monitor_true_residual") {
            k->monitor = true;
        } else if (key == "-ksp_converged_reason") {
            k->print_reason = true;
        } else if (key == "-ksp_view") {
            k->view = true;
        } else if (key == "-pc_type") {
            if (!val) return need("a type");
            const std::string v(val);
            if (v == "none") k->pc_type = SPK_PC_NONE;
            else if (v == "jacobi") k->pc_type = SPK_PC_JACOBI;
            else if (v == "fieldsplit") k->pc_type = SPK_PC_SCHUR;
            else return bad();
            k->pc_type_given = true;
        } else if (key == "-pc_fieldsplit_type") {
            if (!val) return need("a type");
            if (std::string(val) != "schur") return bad();
        } else if (key == "-pc_fieldsplit_schur_fact_type") {
            if (!val) return need("a type");
            const std::string v(val);
            if (v == "diag") k->schur_fact = SPK_SCHUR_DIAG;
            else if (v == "lower") k->schur_fact = SPK_SCHUR_LOWER;
            else if (v == "upper") k->schur_fact = SPK_SCHUR_UPPER;
            else if (v == "full") k->schur_fact = SPK_SCHUR_FULL;
            else return bad();
        } else if (key == "-pc_fieldsplit_schur_precondition") {
            if (!val) return need("a type");
            if (std::string(val) != "selfp") return bad();
        } else if (key == "-pc_fieldsplit_detect_saddle_point") {
            /* implied by the nest */
        } else if (key == "-fieldsplit_0_ksp_type") {
            if (!val) return need("a type");
            const std::string v(val);
            if (v == "richardson") k->inner_richardson = true;   // FP32 Jacobi-Richardson inner solve
            else if (v == "preonly") k->inner_richardson = false;
            else return bad();
        } else if (key == "-fieldsplit_0_ksp_max_it" || key == "-spk_inner_sweeps") {
            if (!val || !parse_int(val, &k->inner_sweeps)) return need("an integer");
            if (key == "-spk_inner_sweeps") k->inner_richardson = k->inner_sweeps > 0;
        } else if (key == "-fieldsplit_0_ksp_richardson_scale" || key == "-spk_inner_omega") {
            if (!val || !parse_double(val, &k->inner_omega)) return need("a real");
        } else if (key == "-fieldsplit_1_ksp_type") {
            if (!val) return need("a type");
            if (std::string(val) != "preonly") return bad();
        } else if (key == "-fieldsplit_0_pc_type" || key == "-fieldsplit_1_pc_type") {
            if (!val) return need("a type");
            if (std::string(val) != "jacobi") return bad();
        } else if (key == "-spk_single_reduce") {
            if (!val || !parse_int(val, &k->opts.single_reduce)) return need("an integer (0 off, 1 on)");
        } else if (key == "-spk_iteration_form") {
            if (!val || !parse_int(val, &k->opts.iteration_form) || k->opts.iteration_form < 0 || k->opts.iteration_form > SPK_ITER_LAST)
                return need("an integer 0..5 (0 automatic, 1 four / 2 two / 3 three launches per iteration, 4 BA, 5 three launches on an un-normalised basis)");
        } else if (key == "-spk_check_every") {
            if (!val || !parse_int(val, &k->opts.check_every)) return need("an integer");
        } else if (key.rfind("-ksp_", 0) == 0 || key.rfind("-pc_", 0) == 0 || key.rfind("-fieldsplit_", 0) == 0) {
            return set_err(k, SPK_ERR_UNSUPPORTED, "unknown solver option " + key);
        }
    }
    k->is_setup = false;
    return SPK_OK;
}

int SpkKSPSetUp(SpkKSP k)
{
    if (!k) return SPK_ERR_ARG;
    if (!k->ksp_type_given)
        return set_err(k, SPK_ERR_UNSUPPORTED, "KSPSetUp: no -ksp_type given; PETSc's default (gmres, left preconditioning) is "
                                               "not implemented -- pass -ksp_type fgmres");
    if (!k->pc_type_given)
        return set_err(k, SPK_ERR_UNSUPPORTED, "KSPSetUp: no -pc_type given; PETSc's default (ilu, bjacobi+ilu in parallel) is "
                                               "not implemented -- pass -pc_type jacobi | fieldsplit | none");
    if (!k->have_ops) return set_err(k, SPK_ERR_STATE, "KSPSetUp: KSPSetOperators has not been called");
    if (k->pc_type == SPK_PC_SCHUR && !k->has_B)
        return set_err(k, SPK_ERR_STATE, "KSPSetUp: -pc_type fieldsplit (schur) needs the constraint block B");
    int rc = spk_pc_set_inner(k->ctx, k->inner_richardson ? k->inner_sweeps : 0, k->inner_omega);
    if (rc != SPK_OK) return from_ctx(k, rc);
    rc = spk_pc_setup(k->ctx, k->pc_type, k->schur_fact);
    if (rc != SPK_OK) return from_ctx(k, rc);
    k->is_setup = true;
    return SPK_OK;
}

int SpkKSPSolve(SpkKSP k, const double *b, double *x)
{
    if (!k) return SPK_ERR_ARG;
    if (!k->have_ops) return set_err(k, SPK_ERR_STATE, "KSPSolve: KSPSetOperators has not been called");
    if (!b || !x) return set_err(k, SPK_ERR_ARG, "KSPSolve: null vector");
    if (!k->is_setup) {  // KSPSolve calls KSPSetUp itself when needed
        const int rc = SpkKSPSetUp(k);
        if (rc != SPK_OK) return rc;
    }
    const int64_t cap = (int64_t)k->opts.max_it + 2;
    k->history.assign((size_t)(cap > (1 << 22) ? (1 << 22) : cap), 0.0);
    const int rc = spk_fgmres(k->ctx, b, x, SPK_MEM_HOST, &k->opts, &k->result, k->history.data(), (int32_t)k->history.size());
    if (rc != SPK_OK) return from_ctx(k, rc);
    k->history.resize((size_t)k->result.hist_len);
    if (k->monitor)
        for (size_t i = 0; i < k->history.size(); ++i) std::printf("%3zu KSP Residual norm %.12e\n", i, k->history[i]);
    if (k->print_reason)
        std::printf("Linear solve %s due to %s iterations %d\n", k->result.reason > 0 ? "converged" : "did not converge",
                    SpkKSPConvergedReasonName(k->result.reason), k->result.its);
    if (k->view)
        std::printf("KSP Object: type fgmres (MI355X device-resident), restart=%d, classical Gram-Schmidt, rtol=%g atol=%g divtol=%g max_it=%d, right preconditioning, pc=%d schur_fact=%d\n",
                    k->opts.restart, k->opts.rtol, k->opts.abstol, k->opts.dtol, k->opts.max_it, k->pc_type, k->schur_fact);
    return SPK_OK;
}

int SpkKSPGetIterationNumber(SpkKSP k, int32_t *its) { if (!k || !its) return SPK_ERR_ARG; *its = k->result.its; return SPK_OK; }
int SpkKSPGetConvergedReason(SpkKSP k, int32_t *r) { if (!k || !r) return SPK_ERR_ARG; *r = k->result.reason; return SPK_OK; }
int SpkKSPGetResidualNorm(SpkKSP k, double *v) { if (!k || !v) return SPK_ERR_ARG; *v = k->result.rnorm; return SPK_OK; }
int SpkKSPGetSolveTime(SpkKSP k, double *v) { if (!k || !v) return SPK_ERR_ARG; *v = k->result.solve_seconds; return SPK_OK; }
int SpkKSPGetResidualHistory(SpkKSP k, const double **h, int32_t *n)
{
    if (!k || !h || !n) return SPK_ERR_ARG;
    *h = k->history.data();
    *n = (int32_t)k->history.size();
    return SPK_OK;
}
int SpkKSPGetOptions(SpkKSP k, spk_opts *o, int32_t *pc, int32_t *sf)
{
    if (!k) return SPK_ERR_ARG;
    if (o) *o = k->opts;
    if (pc) *pc = k->pc_type;
    if (sf) *sf = k->schur_fact;
    return SPK_OK;
}
int SpkKSPGetContext(SpkKSP k, spk_ctx **c) { if (!k || !c) return SPK_ERR_ARG; *c = k->ctx; return SPK_OK; }

}  // extern "C"
