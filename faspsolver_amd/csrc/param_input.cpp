// param_input.cpp -- the ini front-end and the Fortran-style wrapper of the path.
//
//   fasp_hip_param_input            = fasp_param_input (AuxInput.c:86) + fasp_param_init (AuxParam.c:34)
//                                     for the two parameter structs of this path (ITS_param, AMG_param)
//   fasp_fwrapper_dcsr_krylov_amg_  = SolWrapper.c:261 (reads "ini/amg.dat" like the reference)
//
// The reference parses with one hand-written strcmp block per keyword; here the same keywords,
// value formats (fscanf %d / %lf / %s tokens, " = " with spaces, '%' '[' '|' comment lines, the
// rest of a line ignored after the value) and the same range check are table-driven.  Pure host code.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "fasp_internal.h"

namespace {

constexpr int STRLEN = 256;  // fasp.h:77

// the members of the reference's input_param (fasp.h:1123-1190) that feed ITS_param / AMG_param
struct InParam {
    // defaults: fasp_param_input_init, AuxParam.c:100-166
    char   workdir[STRLEN] = "../data/";
    int    print_level = PRINT_SOME, output_type = 0, problem_num = 10;
    int    solver_type = SOLVER_CG, decoup_type = 1, precond_type = PREC_AMG, stop_type = STOP_REL_RES;
    double itsolver_tol = 1e-6, itsolver_abstol = 1e-18;
    int    itsolver_maxit = 500, restart = 25;
    int    ILU_type = 1 /* ILUk */, ILU_lfil = 0;
    double ILU_droptol = 0.001, ILU_relax = 0, ILU_permtol = 0.0;
    int    SWZ_mmsize = 200, SWZ_maxlvl = 2, SWZ_type = 1, SWZ_blksolver = SOLVER_DEFAULT;
    int    AMG_type = CLASSIC_AMG, AMG_levels = 20, AMG_cycle_type = V_CYCLE, AMG_smoother = SMOOTHER_GS;
    int    AMG_smooth_order = CF_ORDER, AMG_presmooth_iter = 1, AMG_postsmooth_iter = 1;
    double AMG_relaxation = 1.0;
    // The reference leaves AMG_polynomial_degree uninitialised when the file does not set it
    // (AuxParam.c:100 has no line for it); fasp_param_amg_init's default is used here.
    int    AMG_polynomial_degree = 3;
    int    AMG_coarse_dof = 500, AMG_coarse_solver = 0;
    double AMG_tol = 1e-6;
    int    AMG_maxit = 1, AMG_ILU_levels = 0, AMG_SWZ_levels = 0, AMG_coarse_scaling = 0;
    int    AMG_amli_degree = 1, AMG_nl_amli_krylov_type = 2;
    int    AMG_coarsening_type = 1, AMG_interpolation_type = 1;
    double AMG_max_row_sum = 0.9, AMG_strong_threshold = 0.3, AMG_truncation_threshold = 0.2;
    int    AMG_aggressive_level = 0, AMG_aggressive_path = 1;
    int    AMG_aggregation_type = PAIRWISE;
    double AMG_quality_bound = 8.0;
    int    AMG_pair_number = 2;
    double AMG_strong_coupled = 0.25;
    int    AMG_max_aggregation = 9;
    double AMG_tentative_smooth = 0.67;
    int    AMG_smooth_filter = 1, AMG_smooth_restriction = 1, AMG_aggregation_norm_type = -1;
};

struct Name { const char* upper; const char* lower; int value; };

enum Kind { K_INT, K_REAL, K_ONOFF, K_NAME, K_STR };
struct Key {
    const char* name;
    Kind        kind;
    size_t      off;          // offset of the int / double / char[] member in InParam
    const Name* names;        // K_NAME: accepted spellings
    int         nnames;
};

const Name AMG_TYPES[] = {{"C", "c", CLASSIC_AMG}, {"SA", "sa", SA_AMG}, {"UA", "ua", UA_AMG}};
const Name CYCLES[] = {{"V", "v", V_CYCLE}, {"W", "w", W_CYCLE}, {"A", "a", AMLI_CYCLE}, {"NA", "na", NL_AMLI_CYCLE},
                       {"VW", "vw", VW_CYCLE}, {"WV", "wv", WV_CYCLE}};
const Name SMOOTHERS[] = {{"JACOBI", "jacobi", SMOOTHER_JACOBI}, {"JACOBIF", "jacobif", 11}, {"GS", "gs", SMOOTHER_GS},
                          {"GSF", "gsf", 12}, {"SGS", "sgs", SMOOTHER_SGS}, {"CG", "cg", 4}, {"SOR", "sor", SMOOTHER_SOR},
                          {"SSOR", "ssor", SMOOTHER_SSOR}, {"GSOR", "gsor", SMOOTHER_GSOR}, {"SGSOR", "sgsor", SMOOTHER_SGSOR},
                          {"POLY", "poly", SMOOTHER_POLY}, {"L1DIAG", "l1diag", SMOOTHER_L1DIAG}, {"BLKOIL", "blkoil", 21},
                          {"SPETEN", "speten", 22}, {"CPRGS", "cprgs", 23}, {"CPTRGS", "cptrgs", 24}};
const Name ORDERS[] = {{"NO", "no", NO_ORDER}, {"CF", "cf", CF_ORDER}};

#define OFF_(m) offsetof(InParam, m)
const Key KEYS[] = {
    {"workdir", K_STR, OFF_(workdir), nullptr, 0},
    {"problem_num", K_INT, OFF_(problem_num), nullptr, 0},
    {"print_level", K_INT, OFF_(print_level), nullptr, 0},
    {"output_type", K_INT, OFF_(output_type), nullptr, 0},
    {"solver_type", K_INT, OFF_(solver_type), nullptr, 0},
    {"stop_type", K_INT, OFF_(stop_type), nullptr, 0},
    {"decoup_type", K_INT, OFF_(decoup_type), nullptr, 0},
    {"precond_type", K_INT, OFF_(precond_type), nullptr, 0},
    {"itsolver_tol", K_REAL, OFF_(itsolver_tol), nullptr, 0},
    {"itsolver_abstol", K_REAL, OFF_(itsolver_abstol), nullptr, 0},
    {"itsolver_maxit", K_INT, OFF_(itsolver_maxit), nullptr, 0},
    {"itsolver_restart", K_INT, OFF_(restart), nullptr, 0},
    {"AMG_ILU_levels", K_INT, OFF_(AMG_ILU_levels), nullptr, 0},
    {"AMG_SWZ_levels", K_INT, OFF_(AMG_SWZ_levels), nullptr, 0},
    {"AMG_type", K_NAME, OFF_(AMG_type), AMG_TYPES, 3},
    {"AMG_strong_coupled", K_REAL, OFF_(AMG_strong_coupled), nullptr, 0},
    {"AMG_max_aggregation", K_INT, OFF_(AMG_max_aggregation), nullptr, 0},
    {"AMG_tentative_smooth", K_REAL, OFF_(AMG_tentative_smooth), nullptr, 0},
    {"AMG_smooth_filter", K_ONOFF, OFF_(AMG_smooth_filter), nullptr, 0},
    {"AMG_smooth_restriction", K_ONOFF, OFF_(AMG_smooth_restriction), nullptr, 0},
    {"AMG_coarse_solver", K_INT, OFF_(AMG_coarse_solver), nullptr, 0},
    {"AMG_coarse_scaling", K_ONOFF, OFF_(AMG_coarse_scaling), nullptr, 0},
    {"AMG_levels", K_INT, OFF_(AMG_levels), nullptr, 0},
    {"AMG_tol", K_REAL, OFF_(AMG_tol), nullptr, 0},
    {"AMG_maxit", K_INT, OFF_(AMG_maxit), nullptr, 0},
    {"AMG_coarse_dof", K_INT, OFF_(AMG_coarse_dof), nullptr, 0},
    {"AMG_cycle_type", K_NAME, OFF_(AMG_cycle_type), CYCLES, 6},
    {"AMG_smoother", K_NAME, OFF_(AMG_smoother), SMOOTHERS, 16},
    {"AMG_smooth_order", K_NAME, OFF_(AMG_smooth_order), ORDERS, 2},
    {"AMG_coarsening_type", K_INT, OFF_(AMG_coarsening_type), nullptr, 0},
    {"AMG_interpolation_type", K_INT, OFF_(AMG_interpolation_type), nullptr, 0},
    {"AMG_aggregation_type", K_INT, OFF_(AMG_aggregation_type), nullptr, 0},
    {"AMG_aggregation_norm_type", K_INT, OFF_(AMG_aggregation_norm_type), nullptr, 0},
    {"AMG_pair_number", K_INT, OFF_(AMG_pair_number), nullptr, 0},
    {"AMG_quality_bound", K_REAL, OFF_(AMG_quality_bound), nullptr, 0},
    {"AMG_aggressive_level", K_INT, OFF_(AMG_aggressive_level), nullptr, 0},
    {"AMG_aggressive_path", K_INT, OFF_(AMG_aggressive_path), nullptr, 0},
    {"AMG_presmooth_iter", K_INT, OFF_(AMG_presmooth_iter), nullptr, 0},
    {"AMG_postsmooth_iter", K_INT, OFF_(AMG_postsmooth_iter), nullptr, 0},
    {"AMG_relaxation", K_REAL, OFF_(AMG_relaxation), nullptr, 0},
    {"AMG_polynomial_degree", K_INT, OFF_(AMG_polynomial_degree), nullptr, 0},
    {"AMG_strong_threshold", K_REAL, OFF_(AMG_strong_threshold), nullptr, 0},
    {"AMG_truncation_threshold", K_REAL, OFF_(AMG_truncation_threshold), nullptr, 0},
    {"AMG_max_row_sum", K_REAL, OFF_(AMG_max_row_sum), nullptr, 0},
    {"AMG_amli_degree", K_INT, OFF_(AMG_amli_degree), nullptr, 0},
    {"AMG_nl_amli_krylov_type", K_INT, OFF_(AMG_nl_amli_krylov_type), nullptr, 0},
    {"ILU_type", K_INT, OFF_(ILU_type), nullptr, 0},
    {"ILU_lfil", K_INT, OFF_(ILU_lfil), nullptr, 0},
    {"ILU_droptol", K_REAL, OFF_(ILU_droptol), nullptr, 0},
    {"ILU_relax", K_REAL, OFF_(ILU_relax), nullptr, 0},
    {"ILU_permtol", K_REAL, OFF_(ILU_permtol), nullptr, 0},
    {"SWZ_mmsize", K_INT, OFF_(SWZ_mmsize), nullptr, 0},
    {"SWZ_maxlvl", K_INT, OFF_(SWZ_maxlvl), nullptr, 0},
    {"SWZ_type", K_INT, OFF_(SWZ_type), nullptr, 0},
    {"SWZ_blksolver", K_INT, OFF_(SWZ_blksolver), nullptr, 0},
};
#undef OFF_

void skip_line(FILE* fp)
{
    if (fscanf(fp, "%*[^\n]")) { /* rest of the line is ignored */ }
}

// ON / OFF in the spellings the reference accepts (AuxInput.c:437-452)
bool parse_onoff(const char* s, int& v)
{
    static const char* on[] = {"ON", "on", "On", "oN"};
    static const char* off[] = {"OFF", "off", "ofF", "oFf", "Off", "oFF", "OfF", "OFf"};
    for (const char* q : on) if (!std::strcmp(s, q)) { v = 1; return true; }
    for (const char* q : off) if (!std::strcmp(s, q)) { v = 0; return true; }
    return false;
}

int parse_file(const char* fname, InParam& in)
{
    FILE* fp = std::fopen(fname, "r");
    if (!fp) return ERROR_OPEN_FILE;
    char buffer[512];
    int  status = FASP_SUCCESS;
    while (status == FASP_SUCCESS) {
        int val = fscanf(fp, "%500s", buffer);
        if (val == EOF) break;
        if (val != 1) { status = ERROR_INPUT_PAR; break; }
        if (buffer[0] == '[' || buffer[0] == '%' || buffer[0] == '|') { skip_line(fp); continue; }
        const Key* key = nullptr;
        for (const Key& k : KEYS) if (!std::strcmp(buffer, k.name)) { key = &k; break; }
        if (!key) {
            std::printf("### WARNING: Unknown input keyword %s!\n", buffer);
            skip_line(fp);
            continue;
        }
        val = fscanf(fp, "%500s", buffer);
        if (val != 1 || std::strcmp(buffer, "=") != 0) { status = ERROR_INPUT_PAR; break; }
        char* base = reinterpret_cast<char*>(&in) + key->off;
        switch (key->kind) {
            case K_INT: {
                int v;
                if (fscanf(fp, "%d", &v) != 1) { status = ERROR_INPUT_PAR; break; }
                *reinterpret_cast<int*>(base) = v;
            } break;
            case K_REAL: {
                double v;
                if (fscanf(fp, "%lf", &v) != 1) { status = ERROR_INPUT_PAR; break; }
                *reinterpret_cast<double*>(base) = v;
            } break;
            case K_STR: {
                if (fscanf(fp, "%500s", buffer) != 1) { status = ERROR_INPUT_PAR; break; }
                std::strncpy(base, buffer, STRLEN - 1);
                base[STRLEN - 1] = '\0';
            } break;
            case K_ONOFF: {
                int v;
                if (fscanf(fp, "%500s", buffer) != 1 || !parse_onoff(buffer, v)) { status = ERROR_INPUT_PAR; break; }
                *reinterpret_cast<int*>(base) = v;
            } break;
            case K_NAME: {
                if (fscanf(fp, "%500s", buffer) != 1) { status = ERROR_INPUT_PAR; break; }
                bool hit = false;
                for (int i = 0; i < key->nnames && !hit; ++i)
                    if (!std::strcmp(buffer, key->names[i].upper) || !std::strcmp(buffer, key->names[i].lower)) {
                        *reinterpret_cast<int*>(base) = key->names[i].value;
                        hit = true;
                    }
                if (!hit) status = ERROR_INPUT_PAR;
            } break;
        }
        if (status == FASP_SUCCESS) skip_line(fp);
    }
    std::fclose(fp);
    return status;
}

// fasp_param_check, AuxInput.c:33-72
int check(const InParam& p)
{
    if (p.problem_num < 0 || p.solver_type < 0 || p.solver_type > 50 || p.precond_type < 0 || p.decoup_type < 0 ||
        p.itsolver_tol < 0 || p.itsolver_abstol < 0 || p.itsolver_maxit < 0 || p.stop_type <= 0 || p.stop_type > 3 ||
        p.restart < 0 || p.ILU_type <= 0 || p.ILU_type > 3 || p.ILU_lfil < 0 || p.ILU_droptol <= 0 || p.ILU_relax < 0 ||
        p.ILU_permtol < 0 || p.SWZ_mmsize < 0 || p.SWZ_maxlvl < 0 || p.SWZ_type < 0 || p.SWZ_blksolver < 0 ||
        p.AMG_type <= 0 || p.AMG_type > 3 || p.AMG_cycle_type <= 0 || p.AMG_levels < 0 || p.AMG_ILU_levels < 0 ||
        p.AMG_coarse_dof <= 0 || p.AMG_tol < 0 || p.AMG_maxit < 0 || p.AMG_coarsening_type <= 0 ||
        p.AMG_coarsening_type > 4 || p.AMG_coarse_solver < 0 || p.AMG_interpolation_type < 0 ||
        p.AMG_interpolation_type > 5 || p.AMG_smoother < 0 || p.AMG_smoother > 30 || p.AMG_strong_threshold < 0.0 ||
        p.AMG_strong_threshold > 0.9999 || p.AMG_truncation_threshold < 0.0 || p.AMG_truncation_threshold > 0.9999 ||
        p.AMG_max_row_sum < 0.0 || p.AMG_presmooth_iter < 0 || p.AMG_postsmooth_iter < 0 || p.AMG_amli_degree < 0 ||
        p.AMG_aggressive_level < 0 || p.AMG_aggressive_path < 0 || p.AMG_aggregation_type < 0 || p.AMG_pair_number < 0 ||
        p.AMG_strong_coupled < 0 || p.AMG_max_aggregation <= 0 || p.AMG_tentative_smooth < 0 ||
        p.AMG_smooth_filter < 0 || p.AMG_smooth_restriction < 0 || p.AMG_smooth_restriction > 1)
        return ERROR_INPUT_PAR;
    return FASP_SUCCESS;
}

constexpr int SOLVER_AMG = 21, SOLVER_FMG = 22;  // fasp_const.h:120-121

}  // namespace

extern "C" {

// fasp_param_input(fname, &in) + fasp_param_init(&in, itsparam, amgparam, NULL, NULL).
// fname == NULL: the defaults of fasp_param_input_init.  Returns FASP_SUCCESS or ERROR_OPEN_FILE /
// ERROR_INPUT_PAR (the reference prints and exits there; a library reports instead).
int fasp_hip_param_input(const char* fname, ITS_param* itsparam, AMG_param* amgparam)
{
    InParam in;
    if (fname) {
        int st = parse_file(fname, in);
        if (st == FASP_SUCCESS) st = check(in);
        if (st < 0) return st;
    }
    if (itsparam) {  // fasp_param_solver_init + fasp_param_solver_set, AuxParam.c:572 / :798
        fasp_param_solver_init(itsparam);
        itsparam->print_level = (short)in.print_level;
        itsparam->itsolver_type = (short)in.solver_type;
        itsparam->decoup_type = (short)in.decoup_type;
        itsparam->precond_type = (short)in.precond_type;
        itsparam->stop_type = (short)in.stop_type;
        itsparam->restart = in.restart;
        if (itsparam->itsolver_type == SOLVER_AMG) {
            itsparam->tol = in.AMG_tol;
            itsparam->maxit = in.AMG_maxit;
        } else {
            itsparam->tol = in.itsolver_tol;
            itsparam->abstol = in.itsolver_abstol;
            itsparam->maxit = in.itsolver_maxit;
        }
    }
    if (amgparam) {  // fasp_param_amg_init + fasp_param_amg_set, AuxParam.c:431 / :653
        AMG_param* p = amgparam;
        fasp_param_amg_init(p);
        p->AMG_type = (short)in.AMG_type;
        p->print_level = (short)in.print_level;
        if (in.solver_type == SOLVER_AMG || in.solver_type == SOLVER_FMG) { p->maxit = in.itsolver_maxit; p->tol = in.itsolver_tol; }
        else { p->maxit = in.AMG_maxit; p->tol = in.AMG_tol; }
        p->max_levels = (short)in.AMG_levels;
        p->cycle_type = (short)in.AMG_cycle_type;
        p->smoother = (short)in.AMG_smoother;
        p->smooth_order = (short)in.AMG_smooth_order;
        p->relaxation = in.AMG_relaxation;
        p->coarse_solver = (short)in.AMG_coarse_solver;
        p->polynomial_degree = (short)in.AMG_polynomial_degree;
        p->presmooth_iter = (short)in.AMG_presmooth_iter;
        p->postsmooth_iter = (short)in.AMG_postsmooth_iter;
        p->coarse_dof = in.AMG_coarse_dof;
        p->coarse_scaling = (short)in.AMG_coarse_scaling;
        p->amli_degree = (short)in.AMG_amli_degree;
        p->amli_coef = nullptr;
        p->nl_amli_krylov_type = (short)in.AMG_nl_amli_krylov_type;
        p->coarsening_type = (short)in.AMG_coarsening_type;
        p->interpolation_type = (short)in.AMG_interpolation_type;
        p->strong_threshold = in.AMG_strong_threshold;
        p->truncation_threshold = in.AMG_truncation_threshold;
        p->max_row_sum = in.AMG_max_row_sum;
        p->aggressive_level = in.AMG_aggressive_level;
        p->aggressive_path = in.AMG_aggressive_path;
        p->aggregation_type = (short)in.AMG_aggregation_type;
        p->pair_number = in.AMG_pair_number;
        p->quality_bound = in.AMG_quality_bound;
        p->strong_coupled = in.AMG_strong_coupled;
        p->max_aggregation = in.AMG_max_aggregation;
        p->tentative_smooth = in.AMG_tentative_smooth;
        p->smooth_filter = (short)in.AMG_smooth_filter;
        p->smooth_restriction = (short)in.AMG_smooth_restriction;
        p->aggregation_norm_type = (short)in.AMG_aggregation_norm_type;
        p->ILU_levels = (short)in.AMG_ILU_levels;
        p->ILU_type = (short)in.ILU_type;
        p->ILU_lfil = in.ILU_lfil;
        p->ILU_droptol = in.ILU_droptol;
        p->ILU_relax = in.ILU_relax;
        p->ILU_permtol = in.ILU_permtol;
        p->SWZ_levels = in.AMG_SWZ_levels;
        p->SWZ_mmsize = in.SWZ_mmsize;
        p->SWZ_maxlvl = in.SWZ_maxlvl;
        p->SWZ_type = in.SWZ_type;
        if (!itsparam) p->maxit = p->maxit > 50 ? p->maxit : 50;  // AuxParam.c:63-65
    }
    return FASP_SUCCESS;
}

// ---------------------------------------------------------------------------
// Readers of the reference's ASCII formats (BlaIO.c:164, :807, :938; data/ files of the reference).
// They return an error code where the reference prints and exits (fasp_chkerr).  Arrays are
// calloc'ed; release them with fasp_hip_free_system / free().
// ---------------------------------------------------------------------------
namespace {
struct File {
    FILE* fp;
    explicit File(const char* name) : fp(std::fopen(name, "r")) {}
    ~File() { if (fp) std::fclose(fp); }
};
// comment lines (% ! /) at the head of a data file, BlaIOUtil.inl:27
int skip_comments(FILE* fp)
{
    for (;;) {
        char buffer[500];
        const long loc = std::ftell(fp);
        if (fscanf(fp, "%499s", buffer) != 1) return ERROR_WRONG_FILE;
        if (buffer[0] == '%' || buffer[0] == '!' || buffer[0] == '/') { skip_line(fp); continue; }
        std::fseek(fp, loc, SEEK_SET);
        return FASP_SUCCESS;
    }
}
}  // namespace

// BlaIO.c:164: matrix file "n / IA(n+1) / JA(nnz) / val(nnz)" with 1-based indices, rhs file "n / values"
int fasp_dcsrvec_read2(const char* filemat, const char* filerhs, dCSRmat* A, dvector* b)
{
    if (!filemat || !filerhs || !A || !b) return ERROR_INPUT_PAR;
    File fm(filemat);
    if (!fm.fp) return ERROR_OPEN_FILE;
    std::printf("%s: reading file %s ...\n", __func__, filemat);
    int n, tmp;
    if (skip_comments(fm.fp) < 0 || fscanf(fm.fp, "%d", &n) != 1 || n <= 0) return ERROR_WRONG_FILE;
    A->row = A->col = n;
    A->IA = static_cast<int*>(std::calloc((size_t)n + 1, sizeof(int)));
    A->JA = nullptr; A->val = nullptr; b->val = nullptr;
    auto fail = [&](int code) {
        std::free(A->IA); std::free(A->JA); std::free(A->val); std::free(b->val);
        A->IA = A->JA = nullptr; A->val = nullptr; b->val = nullptr;
        return code;
    };
    for (int i = 0; i <= n; ++i) {
        if (fscanf(fm.fp, "%d", &tmp) != 1) return fail(ERROR_WRONG_FILE);
        A->IA[i] = tmp - 1;
    }
    const int nz = A->IA[n];
    if (nz < 0) return fail(ERROR_WRONG_FILE);
    A->nnz = nz;
    A->JA = static_cast<int*>(std::calloc((size_t)std::max(nz, 1), sizeof(int)));
    A->val = static_cast<double*>(std::calloc((size_t)std::max(nz, 1), sizeof(double)));
    for (int i = 0; i < nz; ++i) {
        if (fscanf(fm.fp, "%d", &tmp) != 1) return fail(ERROR_WRONG_FILE);
        A->JA[i] = tmp - 1;
    }
    for (int i = 0; i < nz; ++i)
        if (fscanf(fm.fp, "%le", &A->val[i]) != 1) return fail(ERROR_WRONG_FILE);
    File fr(filerhs);
    if (!fr.fp) return fail(ERROR_OPEN_FILE);
    std::printf("%s: reading file %s ...\n", __func__, filerhs);
    int nr;
    if (fscanf(fr.fp, "%d", &nr) != 1) return fail(ERROR_WRONG_FILE);
    if (nr != n) {
        std::printf("### WARNING: rhs size = %d, matrix size = %d!\n", nr, n);
        return fail(ERROR_MAT_SIZE);
    }
    b->row = n;
    b->val = static_cast<double*>(std::calloc((size_t)n, sizeof(double)));
    for (int i = 0; i < n; ++i)
        if (fscanf(fr.fp, "%le", &b->val[i]) != 1) return fail(ERROR_WRONG_FILE);
    return FASP_SUCCESS;
}

// BlaIO.c:938: "n / values"
int fasp_dvec_read(const char* filename, dvector* b)
{
    if (!filename || !b) return ERROR_INPUT_PAR;
    File f(filename);
    if (!f.fp) return ERROR_OPEN_FILE;
    std::printf("%s: reading file %s ...\n", __func__, filename);
    int n;
    if (skip_comments(f.fp) < 0 || fscanf(f.fp, "%d", &n) != 1 || n < 0) return ERROR_WRONG_FILE;
    b->row = n;
    b->val = static_cast<double*>(std::calloc((size_t)std::max(n, 1), sizeof(double)));
    for (int i = 0; i < n; ++i) {
        if (fscanf(f.fp, "%le", &b->val[i]) != 1) { std::free(b->val); b->val = nullptr; return ERROR_WRONG_FILE; }
        if (b->val[i] > fasp::BIGREAL) {
            std::printf("### ERROR: Wrong value = %lf!\n", b->val[i]);
            std::free(b->val); b->val = nullptr;
            return ERROR_INPUT_PAR;
        }
    }
    return FASP_SUCCESS;
}

// BlaIO.c:807: "ROW COL NNZ / nb / storage_manner / n IA.. / n JA.. / n val.." with 0-based indices
int fasp_dbsr_read(const char* filename, dBSRmat* A)
{
    if (!filename || !A) return ERROR_INPUT_PAR;
    File f(filename);
    if (!f.fp) return ERROR_OPEN_FILE;
    std::printf("%s: reading file %s ...\n", __func__, filename);
    int ROW, COL, NNZ, nb, sm, n;
    if (skip_comments(f.fp) < 0 || fscanf(f.fp, "%d %d %d", &ROW, &COL, &NNZ) != 3 || fscanf(f.fp, "%d", &nb) != 1 ||
        fscanf(f.fp, "%d", &sm) != 1 || ROW <= 0 || COL <= 0 || NNZ < 0 || nb <= 0)
        return ERROR_WRONG_FILE;
    A->ROW = ROW; A->COL = COL; A->NNZ = NNZ; A->nb = nb; A->storage_manner = sm;
    A->IA = static_cast<int*>(std::calloc((size_t)ROW + 1, sizeof(int)));
    A->JA = static_cast<int*>(std::calloc((size_t)std::max(NNZ, 1), sizeof(int)));
    A->val = static_cast<double*>(std::calloc((size_t)std::max(NNZ, 1) * nb * nb, sizeof(double)));
    auto fail = [&]() {
        std::free(A->IA); std::free(A->JA); std::free(A->val);
        A->IA = A->JA = nullptr; A->val = nullptr;
        return ERROR_WRONG_FILE;
    };
    if (fscanf(f.fp, "%d", &n) != 1 || n != ROW + 1) return fail();
    for (int i = 0; i < n; ++i) if (fscanf(f.fp, "%d", &A->IA[i]) != 1) return fail();
    if (fscanf(f.fp, "%d", &n) != 1 || n != NNZ) return fail();
    for (int i = 0; i < n; ++i) if (fscanf(f.fp, "%d", &A->JA[i]) != 1) return fail();
    if (fscanf(f.fp, "%d", &n) != 1 || (long long)n != (long long)NNZ * nb * nb) return fail();
    for (int i = 0; i < n; ++i) if (fscanf(f.fp, "%le", &A->val[i]) != 1) return fail();
    return FASP_SUCCESS;
}

namespace {
// The coordinate-format readers of BlaIO.c share one body: "m n nnz" then nnz triples "i j value";
// they differ in the index base and in whether the file holds one triangle of a symmetric matrix.
// Conversion = fasp_format_dcoo_dcsr (BlaFormat.c:36): stable counting sort by row, so every row
// keeps the file's order of its entries.
int read_coo(const char* fn, const char* filename, dCSRmat* A, int base, bool sym)
{
    if (!filename || !A) return ERROR_INPUT_PAR;
    File f(filename);
    if (!f.fp) return ERROR_OPEN_FILE;
    std::printf("%s: reading file %s ...\n", fn, filename);
    int m, n, nnz;
    if (skip_comments(f.fp) < 0 || fscanf(f.fp, "%d %d %d", &m, &n, &nnz) != 3 || m <= 0 || n <= 0 || nnz < 0)
        return ERROR_WRONG_FILE;
    if (sym) nnz = 2 * (nnz - m) + m;  // BlaIO.c:645: a full diagonal is assumed
    if (nnz < 0) return ERROR_WRONG_FILE;
    std::vector<int> ri((size_t)nnz + 1), ci((size_t)nnz + 1);
    std::vector<double> v((size_t)nnz + 1);
    int k = 0;
    while (k < nnz) {
        int i, j; double value;
        if (fscanf(f.fp, "%d %d %le", &i, &j, &value) != 3) return ERROR_WRONG_FILE;
        i -= base; j -= base;
        if (i < 0 || i >= m || j < 0 || j >= n) return ERROR_WRONG_FILE;
        ri[k] = i; ci[k] = j; v[k] = value; ++k;
        if (sym && i != j) {
            if (k >= nnz) return ERROR_WRONG_FILE;  // more off-diagonal entries than the header promises
            ri[k] = j; ci[k] = i; v[k] = value; ++k;
        }
    }
    A->row = m; A->col = n; A->nnz = nnz;
    A->IA = static_cast<int*>(std::calloc((size_t)m + 1, sizeof(int)));
    A->JA = static_cast<int*>(std::calloc((size_t)std::max(nnz, 1), sizeof(int)));
    A->val = static_cast<double*>(std::calloc((size_t)std::max(nnz, 1), sizeof(double)));
    std::vector<int> ind((size_t)m + 1, 0);
    for (int q = 0; q < nnz; ++q) ind[ri[q] + 1]++;
    for (int i = 1; i <= m; ++i) { A->IA[i] = A->IA[i - 1] + ind[i]; ind[i] = A->IA[i]; }
    for (int q = 0; q < nnz; ++q) {
        const int pos = ind[ri[q]]++;
        A->JA[pos] = ci[q]; A->val[pos] = v[q];
    }
    return FASP_SUCCESS;
}
}  // namespace

int fasp_dcoo_read(const char* filename, dCSRmat* A) { return read_coo(__func__, filename, A, 0, false); }        // BlaIO.c:332
int fasp_dcoo_read1(const char* filename, dCSRmat* A) { return read_coo(__func__, filename, A, 1, false); }       // BlaIO.c:384
int fasp_dcoo_shift_read(const char* filename, dCSRmat* A) { return read_coo(__func__, filename, A, 1, false); }  // BlaIO.c:514
int fasp_dmtx_read(const char* filename, dCSRmat* A) { return read_coo(__func__, filename, A, 1, false); }        // BlaIO.c:567
int fasp_dmtxsym_read(const char* filename, dCSRmat* A) { return read_coo(__func__, filename, A, 1, true); }      // BlaIO.c:624

// BlaIO.c:1388: "n / values" with 15 digits
int fasp_dvec_write(const char* filename, dvector* vec)
{
    if (!filename || !vec) return ERROR_INPUT_PAR;
    FILE* fp = std::fopen(filename, "w");
    if (!fp) return ERROR_OPEN_FILE;
    std::printf("%s: writing to file %s ...\n", __func__, filename);
    std::fprintf(fp, "%d\n", vec->row);
    for (int i = 0; i < vec->row; ++i) std::fprintf(fp, "%0.15e\n", vec->val[i]);
    std::fclose(fp);
    return FASP_SUCCESS;
}

// BlaIO.c:1623: comment line with the sizes, then 1-based triples (what fasp_dcoo_read1 reads back)
int fasp_dcsr_write_coo(const char* filename, const dCSRmat* A)
{
    if (!filename || !A) return ERROR_INPUT_PAR;
    FILE* fp = std::fopen(filename, "w");
    if (!fp) return ERROR_OPEN_FILE;
    std::printf("%s: writing to file %s ...\n", __func__, filename);
    std::fprintf(fp, "%% dimension of the matrix and nonzeros %d  %d  %d\n", A->row, A->col, A->nnz);
    for (int i = 0; i < A->row; i++)
        for (int j = A->IA[i]; j < A->IA[i + 1]; j++)
            std::fprintf(fp, "%d %d %+.15E\n", i + 1, A->JA[j] + 1, A->val[j]);
    std::fclose(fp);
    return FASP_SUCCESS;
}

// BlaIO.c:1145: the two-file format fasp_dcsrvec_read2 reads
int fasp_dcsrvec_write2(const char* filemat, const char* filerhs, dCSRmat* A, dvector* b)
{
    if (!filemat || !filerhs || !A || !b) return ERROR_INPUT_PAR;
    const int m = A->row, nnz = A->nnz;
    FILE* fp = std::fopen(filemat, "w");
    if (!fp) return ERROR_OPEN_FILE;
    std::printf("%s: writing to file %s ...\n", __func__, filemat);
    std::fprintf(fp, "%d\n", m);
    for (int i = 0; i < m + 1; ++i) std::fprintf(fp, "%d\n", A->IA[i] + 1);
    for (int i = 0; i < nnz; ++i) std::fprintf(fp, "%d\n", A->JA[i] + 1);
    for (int i = 0; i < nnz; ++i) std::fprintf(fp, "%le\n", A->val[i]);
    std::fclose(fp);
    fp = std::fopen(filerhs, "w");
    if (!fp) return ERROR_OPEN_FILE;
    std::printf("%s: writing to file %s ...\n", __func__, filerhs);
    std::fprintf(fp, "%d\n", b->row);
    for (int i = 0; i < b->row; ++i) std::fprintf(fp, "%le\n", b->val[i]);
    std::fclose(fp);
    return FASP_SUCCESS;
}

void fasp_hip_free_bsr(dBSRmat* A)
{
    if (!A) return;
    std::free(A->IA); std::free(A->JA); std::free(A->val);
    A->IA = A->JA = nullptr; A->val = nullptr;
}

// SolWrapper.c:261 -- Fortran callers: CALL FASP_FWRAPPER_DCSR_KRYLOV_AMG(n, nnz, ia, ja, a, b, u, tol, maxit, prtlvl).
// Parameters come from "ini/amg.dat" in the working directory, exactly as in the reference.
void fasp_fwrapper_dcsr_krylov_amg_(int* n, int* nnz, int* ia, int* ja, double* a, double* b, double* u, double* tol,
                                    int* maxit, int* ptrlvl)
{
    ITS_param itsparam;
    AMG_param amgparam;
    const int st = fasp_hip_param_input("ini/amg.dat", &itsparam, &amgparam);
    if (st < 0) {  // fasp_chkerr: the reference prints and exits
        std::printf("### ERROR: %s [fasp_param_input]\n", st == ERROR_OPEN_FILE ? "Cannot open ini/amg.dat!" : "Wrong input parameters!");
        std::exit(st);
    }
    itsparam.tol = *tol;
    itsparam.maxit = *maxit;
    itsparam.print_level = (short)*ptrlvl;
    dCSRmat mat;
    mat.row = *n; mat.col = *n; mat.nnz = *nnz; mat.IA = ia; mat.JA = ja; mat.val = a;
    dvector rhs{*n, b}, sol{*n, u};
    const int ret = fasp_solver_dcsr_krylov_amg(&mat, &rhs, &sol, &itsparam, &amgparam);
    if (ret < 0) std::printf("### WARNING: fasp_solver_dcsr_krylov_amg returned %d\n", ret);
}

// SolWrapper.c:136 -- AMG as the solver, parameters from fasp_param_amg_init
void fasp_fwrapper_dcsr_amg_(int* n, int* nnz, int* ia, int* ja, double* a, double* b, double* u, double* tol, int* maxit,
                             int* ptrlvl)
{
    AMG_param amgparam;
    fasp_param_amg_init(&amgparam);
    amgparam.tol = *tol;
    amgparam.print_level = (short)*ptrlvl;
    amgparam.maxit = *maxit;
    dCSRmat mat;
    mat.row = *n; mat.col = *n; mat.nnz = *nnz; mat.IA = ia; mat.JA = ja; mat.val = a;
    dvector rhs{*n, b}, sol{*n, u};
    const int ret = fasp_solver_amg(&mat, &rhs, &sol, &amgparam);
    if (ret < 0) std::printf("### WARNING: fasp_solver_amg returned %d\n", ret);
}

// SolWrapper.c:397 -- block matrices: UA-AMG + VFGMRES.  The reference leaves storage_manner
// uninitialised here; its kernels only know the row-major layout 0, which is what is set.
void fasp_fwrapper_dbsr_krylov_amg_(int* n, int* nnz, int* nb, int* ia, int* ja, double* a, double* b, double* u,
                                    double* tol, int* maxit, int* ptrlvl)
{
    AMG_param amgparam;
    ITS_param itsparam;
    fasp_param_amg_init(&amgparam);
    amgparam.AMG_type = UA_AMG;
    amgparam.print_level = (short)*ptrlvl;
    fasp_param_solver_init(&itsparam);
    itsparam.tol = *tol;
    itsparam.print_level = (short)*ptrlvl;
    itsparam.maxit = *maxit;
    itsparam.itsolver_type = SOLVER_VFGMRES;
    dBSRmat mat;
    mat.ROW = *n; mat.COL = *n; mat.NNZ = *nnz; mat.nb = *nb; mat.storage_manner = 0;
    mat.IA = ia; mat.JA = ja; mat.val = a;
    dvector rhs{*n * *nb, b}, sol{*n * *nb, u};
    const int ret = fasp_solver_dbsr_krylov_amg(&mat, &rhs, &sol, &itsparam, &amgparam);
    if (ret < 0) std::printf("### WARNING: fasp_solver_dbsr_krylov_amg returned %d\n", ret);
}

}  // extern "C"
