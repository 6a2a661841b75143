/*
 * ref_shim.c -- flat accessors around the REFERENCE library, for fixture
 * generation and oracle validation.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is our own code.  It is compiled against the reference's headers
 * where they lie (-I/root/reference/base/include ...) and linked with the
 * reference's own objects into oracle/_ref/libfasp_ref.so by oracle/Makefile.
 * It exists because ctypes cannot comfortably walk the reference's AMG_data
 * (1104-byte struct with optional-solver members).
 */
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fasp.h"
#include "fasp_functs.h"
#include "poisson_fdm.h"

/* ---- ABI facts ---------------------------------- */
int ref_sizeof(int which)
{
    switch (which) {
        case 0: return (int)sizeof(dCSRmat);
        case 1: return (int)sizeof(dvector);
        case 2: return (int)sizeof(ITS_param);
        case 3: return (int)sizeof(AMG_param);
        case 4: return (int)sizeof(AMG_data);
        case 5: return (int)sizeof(precond_data);
        case 6: return (int)sizeof(precond);
        case 7: return (int)sizeof(ivector);
        default: return -1;
    }
}
/* every field of AMG_data / precond_data, in declaration order (fasp.h:804-888, :894-981) */
int ref_offsetof_amgdata(int which)
{
    switch (which) {
        case 0: return (int)offsetof(AMG_data, max_levels);
        case 1: return (int)offsetof(AMG_data, num_levels);
        case 2: return (int)offsetof(AMG_data, A);
        case 3: return (int)offsetof(AMG_data, R);
        case 4: return (int)offsetof(AMG_data, P);
        case 5: return (int)offsetof(AMG_data, b);
        case 6: return (int)offsetof(AMG_data, x);
        case 7: return (int)offsetof(AMG_data, Numeric);
        case 8: return (int)offsetof(AMG_data, pdata);
        case 9: return (int)offsetof(AMG_data, cfmark);
        case 10: return (int)offsetof(AMG_data, ILU_levels);
        case 11: return (int)offsetof(AMG_data, LU);
        case 12: return (int)offsetof(AMG_data, near_kernel_dim);
        case 13: return (int)offsetof(AMG_data, near_kernel_basis);
        case 14: return (int)offsetof(AMG_data, SWZ_levels);
        case 15: return (int)offsetof(AMG_data, Schwarz);
        case 16: return (int)offsetof(AMG_data, w);
        case 17: return (int)offsetof(AMG_data, mumps);
        case 18: return (int)offsetof(AMG_data, cycle_type);
        case 19: return (int)offsetof(AMG_data, ic);
        case 20: return (int)offsetof(AMG_data, icmap);
        case 21: return (int)offsetof(AMG_data, colors);
        case 22: return (int)offsetof(AMG_data, weight);
        case 23: return (int)sizeof(AMG_data);
        case 24: return (int)sizeof(ILU_data);
        case 25: return (int)sizeof(SWZ_data);
        case 26: return (int)sizeof(ILU_param);
        default: return -1;
    }
}
int ref_offsetof_precdata(int which)
{
    switch (which) {
        case 0: return (int)offsetof(precond_data, AMG_type);
        case 1: return (int)offsetof(precond_data, print_level);
        case 2: return (int)offsetof(precond_data, maxit);
        case 3: return (int)offsetof(precond_data, max_levels);
        case 4: return (int)offsetof(precond_data, tol);
        case 5: return (int)offsetof(precond_data, cycle_type);
        case 6: return (int)offsetof(precond_data, smoother);
        case 7: return (int)offsetof(precond_data, smooth_order);
        case 8: return (int)offsetof(precond_data, presmooth_iter);
        case 9: return (int)offsetof(precond_data, postsmooth_iter);
        case 10: return (int)offsetof(precond_data, relaxation);
        case 11: return (int)offsetof(precond_data, polynomial_degree);
        case 12: return (int)offsetof(precond_data, coarsening_type);
        case 13: return (int)offsetof(precond_data, coarse_solver);
        case 14: return (int)offsetof(precond_data, coarse_scaling);
        case 15: return (int)offsetof(precond_data, amli_degree);
        case 16: return (int)offsetof(precond_data, nl_amli_krylov_type);
        case 17: return (int)offsetof(precond_data, tentative_smooth);
        case 18: return (int)offsetof(precond_data, amli_coef);
        case 19: return (int)offsetof(precond_data, mgl_data);
        case 20: return (int)offsetof(precond_data, LU);
        case 21: return (int)offsetof(precond_data, A);
        case 22: return (int)offsetof(precond_data, A_nk);
        case 23: return (int)offsetof(precond_data, P_nk);
        case 24: return (int)offsetof(precond_data, R_nk);
        case 25: return (int)offsetof(precond_data, r);
        case 26: return (int)offsetof(precond_data, w);
        case 27: return (int)sizeof(precond_data);
        default: return -1;
    }
}
int ref_offsetof_amgparam(int which)
{
    switch (which) {
        case 0: return (int)offsetof(AMG_param, tol);
        case 1: return (int)offsetof(AMG_param, coarse_dof);
        case 2: return (int)offsetof(AMG_param, relaxation);
        case 3: return (int)offsetof(AMG_param, amli_coef);
        case 4: return (int)offsetof(AMG_param, strong_threshold);
        case 5: return (int)offsetof(AMG_param, theta);
        case 6: return (int)offsetof(AMG_param, smoother);
        case 7: return (int)offsetof(AMG_param, ILU_levels);
        case 8: return (int)offsetof(AMG_param, SWZ_levels);
        default: return -1;
    }
}

/* ---- synthetic input: the reference's own generator --------------------- */
int ref_poisson7pt(int nx, int ny, int nz, dCSRmat* A, dvector* b, dvector* u)
{
    fsls_BandMatrix* B = NULL;
    fsls_CSRMatrix*  C = NULL;
    fsls_XVector *   f = NULL, *ue = NULL;
    fsls_BuildLinearSystem_7pt3d(0, nx, ny, nz, &B, &f, &ue);
    fsls_Band2CSRMatrix(B, &C);
    const int n = fsls_CSRMatrixNumRows(C), nnz = fsls_CSRMatrixI(C)[n];
    *A = fasp_dcsr_create(n, n, nnz);
    memcpy(A->IA, fsls_CSRMatrixI(C), (size_t)(n + 1) * sizeof(int));
    memcpy(A->JA, fsls_CSRMatrixJ(C), (size_t)nnz * sizeof(int));
    memcpy(A->val, fsls_CSRMatrixData(C), (size_t)nnz * sizeof(double));
    *b = fasp_dvec_create(n);
    *u = fasp_dvec_create(n);
    memcpy(b->val, fsls_XVectorData(f), (size_t)n * sizeof(double));
    memcpy(u->val, fsls_XVectorData(ue), (size_t)n * sizeof(double));
    fsls_BandMatrixDestroy(B);
    fsls_CSRMatrixDestroy(C);
    fsls_XVectorDestroy(f);
    fsls_XVectorDestroy(ue);
    return 0;
}

/* ---- classical setup ----------------------------------------------------- */
void* ref_amg_setup_rs(dCSRmat* A, AMG_param* param)
{
    AMG_data* mgl = fasp_amg_data_create(param->max_levels);
    mgl[0].A = fasp_dcsr_create(A->row, A->col, A->nnz);
    fasp_dcsr_cp(A, &mgl[0].A);
    mgl[0].b = fasp_dvec_create(A->col);
    mgl[0].x = fasp_dvec_create(A->col);
    if ((param->AMG_type == SA_AMG ? fasp_amg_setup_sa(mgl, param)
         : param->AMG_type == UA_AMG ? fasp_amg_setup_ua(mgl, param) : fasp_amg_setup_rs(mgl, param)) < 0) return NULL;
    return mgl;
}
int ref_amg_num_levels(void* h) { return ((AMG_data*)h)[0].num_levels; }
int ref_amg_get_matrix(void* h, int l, int which, dCSRmat* view)
{
    AMG_data* mgl = (AMG_data*)h;
    *view = which == 0 ? mgl[l].A : which == 1 ? mgl[l].P : mgl[l].R;
    return 0;
}
int* ref_amg_get_cfmark(void* h, int l) { return ((AMG_data*)h)[l].cfmark.val; }
void ref_amg_free(void* h, AMG_param* param) { fasp_amg_data_free((AMG_data*)h, param); }

/* z = B r through the reference's fasp_precond_amg (PreCSR.c:416) */
void ref_precond_amg(void* h, AMG_param* param, double* r, double* z)
{
    AMG_data*    mgl = (AMG_data*)h;
    precond_data pcdata;
    fasp_param_amg_to_prec(&pcdata, param);
    pcdata.max_levels = mgl[0].num_levels;
    pcdata.mgl_data   = mgl;
    fasp_precond_amg(r, z, &pcdata);
}

/* ---- full solve with a full-precision residual history ------------------ */
typedef struct {
    precond_data* pcdata;
    double*       hist;
    int           cap, n, m;
    int           fmg;
} hist_pc;

static void hist_pc_fct(double* r, double* z, void* data)
{
    hist_pc* h = (hist_pc*)data;
    if (h->hist && h->n < h->cap) h->hist[h->n] = fasp_blas_darray_norm2(h->m, r);
    h->n++;
    if (h->fmg) { fasp_precond_famg(r, z, h->pcdata); return; }   /* SolCSR.c:537 */
    switch (h->pcdata->cycle_type) {   /* SolCSR.c:540-549 */
        case AMLI_CYCLE: fasp_precond_amli(r, z, h->pcdata); break;
        case NL_AMLI_CYCLE: fasp_precond_namli(r, z, h->pcdata); break;
        default: fasp_precond_amg(r, z, h->pcdata);
    }
}

/* Mirrors SolCSR.c:476-569 step by step, only swapping pc.fct for the recording
 * wrapper above.  hist[k] = ||r_k||_2 for every r handed to the preconditioner
 * (k = 0 .. iters-1); hist[iters] = ||b - A x||_2 recomputed from the returned x
 * with the reference's own aAxpy + norm2 (what KryPcg.c:280-287 leaves in absres). */
int ref_krylov_amg_hist(dCSRmat* A, dvector* b, dvector* x, ITS_param* itparam,
                        AMG_param* amgparam, double* hist, int cap, int* nhist)
{
    const int m = A->row, n = A->col, nnz = A->nnz;
    int       status;
    AMG_data* mgl = fasp_amg_data_create(amgparam->max_levels);
    mgl[0].A      = fasp_dcsr_create(m, n, nnz);
    fasp_dcsr_cp(A, &mgl[0].A);
    mgl[0].b = fasp_dvec_create(n);
    mgl[0].x = fasp_dvec_create(n);
    status = (amgparam->AMG_type == SA_AMG) ? fasp_amg_setup_sa(mgl, amgparam)
             : (amgparam->AMG_type == UA_AMG) ? fasp_amg_setup_ua(mgl, amgparam) : fasp_amg_setup_rs(mgl, amgparam);
    if (status < 0) goto FINISHED;

    precond_data pcdata;
    fasp_param_amg_to_prec(&pcdata, amgparam);
    pcdata.max_levels = mgl[0].num_levels;
    pcdata.mgl_data   = mgl;

    hist_pc hp = {&pcdata, hist, cap, 0, m, itparam->precond_type == PREC_FMG};
    precond pc;
    pc.data = &hp;
    pc.fct  = hist_pc_fct;
    status  = fasp_solver_dcsr_itsolver(A, b, x, &pc, itparam);
    {
        double* r = (double*)malloc((size_t)m * sizeof(double));
        memcpy(r, b->val, (size_t)m * sizeof(double));
        fasp_blas_dcsr_aAxpy(-1.0, A, x->val, r);
        if (hist && hp.n < cap) hist[hp.n] = fasp_blas_darray_norm2(m, r);
        hp.n++;
        free(r);
    }
    if (nhist) *nhist = hp.n;
FINISHED:
    fasp_amg_data_free(mgl, amgparam);
    return status;
}

/* coarsest-level solver exactly as PreMGUtil.inl:37 wires it */
int ref_coarse_spcg(dCSRmat* A, dvector* b, dvector* x, double ctol)
{
    const int n     = A->row;
    const int maxit = MAX(250, MIN(n * n, 1000));
    return fasp_solver_dcsr_spcg(A, b, x, NULL, ctol, maxit, 1, 0);
}

/* safety-net GMRES exactly as PreMGUtil.inl:51 wires it (pc == NULL, restart 20) */
int ref_coarse_spvgmres(dCSRmat* A, dvector* b, dvector* x, double ctol, int maxit, int restart)
{
    return fasp_solver_dcsr_spvgmres(A, b, x, NULL, ctol, maxit, (SHORT)restart, 1, 0);
}

/* ---- BSR path ------------------------------------------------------------ */
#include "fasp_block.h"
void* ref_bsr_setup_ua(dBSRmat* A, AMG_param* param)
{
    AMG_data_bsr* mgl = fasp_amg_data_bsr_create(param->max_levels);
    mgl[0].A = fasp_dbsr_create(A->ROW, A->COL, A->NNZ, A->nb, A->storage_manner);
    mgl[0].b = fasp_dvec_create(mgl[0].A.ROW * mgl[0].A.nb);
    mgl[0].x = fasp_dvec_create(mgl[0].A.COL * mgl[0].A.nb);
    fasp_dbsr_cp(A, &(mgl[0].A));
    if (fasp_amg_setup_ua_bsr(mgl, param) < 0) return NULL;
    return mgl;
}
int ref_bsr_num_levels(void* h) { return ((AMG_data_bsr*)h)[0].num_levels; }
int ref_bsr_get_matrix(void* h, int l, int which, dBSRmat* view)
{
    AMG_data_bsr* mgl = (AMG_data_bsr*)h;
    *view = which == 0 ? mgl[l].A : which == 1 ? mgl[l].P : mgl[l].R;
    return 0;
}
double* ref_bsr_get_diaginv(void* h, int l) { return ((AMG_data_bsr*)h)[l].diaginv.val; }
void ref_bsr_free(void* h, AMG_param* param) { fasp_amg_data_bsr_free((AMG_data_bsr*)h, param); }
int ref_sizeof_bsr(void) { return (int)sizeof(dBSRmat); }
/* z = B r through the reference's fasp_precond_dbsr_amg (PreBSR.c:1149), its precond_data_bsr filled field by field as
 * fasp_solver_dbsr_krylov_amg does (SolBSR.c:399-415; the fields that function leaves unset are zero here) */
void ref_bsr_precond(void* h, AMG_param* amgparam, dBSRmat* A, double* r, double* z)
{
    AMG_data_bsr*    mgl = (AMG_data_bsr*)h;
    precond_data_bsr precdata;
    memset(&precdata, 0, sizeof(precdata));
    precdata.print_level      = amgparam->print_level;
    precdata.maxit            = amgparam->maxit;
    precdata.tol              = amgparam->tol;
    precdata.cycle_type       = amgparam->cycle_type;
    precdata.smoother         = amgparam->smoother;
    precdata.presmooth_iter   = amgparam->presmooth_iter;
    precdata.postsmooth_iter  = amgparam->postsmooth_iter;
    precdata.coarsening_type  = amgparam->coarsening_type;
    precdata.relaxation       = amgparam->relaxation;
    precdata.coarse_scaling   = amgparam->coarse_scaling;
    precdata.amli_degree      = amgparam->amli_degree;
    precdata.amli_coef        = amgparam->amli_coef;
    precdata.tentative_smooth = amgparam->tentative_smooth;
    precdata.max_levels       = mgl[0].num_levels;
    precdata.mgl_data         = mgl;
    precdata.A                = A;
    fasp_precond_dbsr_amg(r, z, &precdata);
}


/* fasp_param_input + fasp_param_init on an ini file (AuxInput.c:86, AuxParam.c:34) */
void ref_param_from_file(const char* fname, ITS_param* itsparam, AMG_param* amgparam)
{
    input_param in;
    ILU_param   ilu;
    memset(&in, 0, sizeof(in));  /* (the reference leaves AMG_polynomial_degree unset) */
    fasp_param_input(fname, &in);
    fasp_param_init(&in, itsparam, amgparam, &ilu, NULL);
}
