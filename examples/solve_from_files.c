/* solve_from_files.c -- the flow of the reference's test driver (test/main/test.c: read ini, read
 * matrix + rhs, fasp_solver_dcsr_krylov_amg, report) against libfasp_hip.so.  Plain C.
 *
 *   gcc -Iinclude examples/solve_from_files.c -o solve -Lfaspsolver_amd -lfasp_hip -Wl,-rpath,$PWD/faspsolver_amd -lm
 *   ./solve ini-file matrix-file rhs-file
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "fasp_hip.h"

int main(int argc, char** argv)
{
    if (argc != 4) {
        fprintf(stderr, "usage: %s ini-file csr-matrix-file rhs-file\n", argv[0]);
        return 2;
    }
    ITS_param itsparam;
    AMG_param amgparam;
    dCSRmat   A;
    dvector   b, x;
    int       status = fasp_hip_param_input(argv[1], &itsparam, &amgparam);
    if (status < 0) { fprintf(stderr, "cannot read parameters from %s (%d)\n", argv[1], status); return 1; }
    status = fasp_dcsrvec_read2(argv[2], argv[3], &A, &b);
    if (status < 0) { fprintf(stderr, "cannot read the system (%d)\n", status); return 1; }
    printf("A: m = %d, n = %d, nnz = %d\n", A.row, A.col, A.nnz);

    x.row = A.row;
    x.val = (double*)calloc((size_t)A.row, sizeof(double));
    status = fasp_solver_dcsr_krylov_amg(&A, &b, &x, &itsparam, &amgparam);

    /* true relative residual, computed on the host */
    double rr = 0.0, bb = 0.0;
    for (int i = 0; i < A.row; ++i) {
        double r = b.val[i];
        for (int k = A.IA[i]; k < A.IA[i + 1]; ++k) r -= A.val[k] * x.val[A.JA[k]];
        rr += r * r;
        bb += b.val[i] * b.val[i];
    }
    printf("status = %d, ||b - A x|| / ||b|| = %.6e\n", status, sqrt(rr / bb));
    fasp_hip_free_system(&A, &b, &x);
    return status >= 0 ? 0 : 1;
}
