/* poisson_pcg.c -- the flow of the reference's tutorial caller (tutorial/main/poisson-pcg.c: read the FE system,
 * fasp_precond_setup(PREC_AMG), fasp_solver_dcsr_pcg with that preconditioner, fasp_amg_data_free on
 * ((precond_data*)pc->data)->mgl_data, fasp_mem_free of pc->data and pc) against libfasp_hip.so: every call below is
 * a reference entry point with the reference's signature; only the parameter source differs (defaults of
 * fasp_param_solver_init / fasp_param_amg_init with the tutorial's ini/pcg.dat values set by hand instead of
 * fasp_param_set + fasp_param_init).  Plain C.
 *
 *   gcc -Iinclude examples/poisson_pcg.c -o poisson_pcg -Lfaspsolver_amd -lfasp_hip -Wl,-rpath,$PWD/faspsolver_amd
 *   ./poisson_pcg csrmat_FE.dat rhs_FE.dat
 */
#include <stdio.h>
#include <stdlib.h>

#include "fasp_hip.h"

int main(int argc, char** argv)
{
    if (argc != 3) {
        fprintf(stderr, "usage: %s csr-matrix-file rhs-file\n", argv[0]);
        return 2;
    }
    ITS_param itparam;
    AMG_param amgparam;
    ILU_param iluparam = {0, 1, 2, 0.1, 0.9, 0.001};
    fasp_param_solver_init(&itparam);
    fasp_param_amg_init(&amgparam);
    /* tutorial/ini/pcg.dat as the shipped run used it (tutorial/out/poisson-pcg-c.out) */
    itparam.print_level = 2; itparam.precond_type = PREC_AMG; itparam.stop_type = 1;
    itparam.tol = 1e-6; itparam.maxit = 500;
    amgparam.print_level = 2;

    const short  prtlvl = itparam.print_level, pc_type = itparam.precond_type, stop_type = itparam.stop_type;
    const int    maxit = itparam.maxit;
    const double tol = itparam.tol, abstol = itparam.abstol;

    dCSRmat A;
    dvector b, x;
    if (fasp_dcsrvec_read2(argv[1], argv[2], &A, &b) < 0) { fprintf(stderr, "cannot read the system\n"); return 1; }
    printf("A: m = %d, n = %d, nnz = %d\n", A.row, A.col, A.nnz);
    printf("b: n = %d\n", b.row);

    /* Step 3 of the tutorial: the preconditioner object */
    precond* pc = fasp_precond_setup(pc_type, &amgparam, &iluparam, &A);

    /* the hierarchy is visible through the reference's own structures */
    precond_data* pcdata = (precond_data*)pc->data;
    AMG_data*     mgl = pcdata->mgl_data;
    for (int l = 0; l < mgl[0].num_levels; ++l)
        printf("mgl[%d]: A %d x %d, %d nonzeros%s\n", l, mgl[l].A.row, mgl[l].A.col, mgl[l].A.nnz,
               l + 1 < mgl[0].num_levels ? "" : " (coarsest)");

    /* Step 4: zero initial guess, PCG called directly */
    fasp_dvec_alloc(A.row, &x);
    fasp_dvec_set(A.row, &x, 0.0);
    const int status = fasp_solver_dcsr_pcg(&A, &b, &x, pc, tol, abstol, maxit, stop_type, prtlvl);
    printf("status = %d\n", status);

    /* one more application of the preconditioner through its function pointer, host vectors in and out */
    dvector z = fasp_dvec_create(A.row);
    pc->fct(b.val, z.val, pc->data);
    double zb = 0.0;
    for (int i = 0; i < A.row; ++i) zb += z.val[i] * b.val[i];
    printf("(B b, b) = %.10e\n", zb);
    fasp_dvec_free(&z);

    /* Step 5: clean up exactly as the tutorial does */
    fasp_amg_data_free(((precond_data*)pc->data)->mgl_data, &amgparam);
    if (pc_type != PREC_NULL) fasp_mem_free(pc->data);
    fasp_mem_free(pc);
    fasp_dcsr_free(&A);
    fasp_dvec_free(&b);
    fasp_dvec_free(&x);
    return status >= 0 ? 0 : 1;
}
