/*
 * fasp_oracle.h -- CPU restatement of the reference's AMG-preconditioned
 * Krylov path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (faspsolver_amd/) never links, imports or
 * calls it.  Every function cites the reference file:line it restates
 * (paths relative to the reference tree).  Parity of this restatement is
 * pinned against (a) the reference's own golden logs (test/out/reg.out,
 * tutorial/out/poisson-pcg-c.out) and (b) the reference itself compiled from
 * its own sources into oracle/_ref/libfasp_ref.so (see oracle/Makefile) --
 * tests/test_oracle_vs_ref.py compares hierarchies bit-for-bit.
 *
 * Types come from include/fasp_hip.h (layout == serial reference headers).
 */
#ifndef FASP_ORACLE_H
#define FASP_ORACLE_H

#include "../include/fasp_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LVL 20 /* MAX_AMG_LVL, fasp_const.h:259 */

typedef struct {
    dCSRmat A, P, R;
    ivector cfmark;
    dvector b, x, w;
} orc_level;

typedef struct {
    int       num_levels;
    orc_level L[ORC_MAX_LVL];
    /* counters (not in the reference; instrumentation for tests) */
    long long coarse_iters;
    long long cycles;
    /* AMLI polynomial coefficients (PreAMGSetupRS.c:93-97; the reference hangs them on AMG_param) */
    double amli_coef[32];
    /* AMG_data.cycle_type of every level (PreAMGSetupRS.c:325, PreAMGSetupUA.c:380-401): read by the
     * nonlinear AMLI cycle, which drops to a V-cycle where it is <= 1 */
    int level_cycle_type[ORC_MAX_LVL];
    /* borrowed hierarchies only: the level matrices are this structure's own copies (orc_amg_place) */
    int placed;
} orc_amg;

/* threads used by the row-parallel loops (1 = exact serial order everywhere,
 * including reductions; >1 keeps per-row arithmetic identical but sums
 * reductions in fixed per-thread blocks). */
void orc_set_threads(int n);
int  orc_get_threads(void);

/* parameters: AuxParam.c:431 / :572 */
void orc_param_amg_init(AMG_param* p);
void orc_param_solver_init(ITS_param* p);

/* synthetic input: test/src/FdmPoisson.c:439 (7-pt band system) + :731 (band->CSR) */
int  orc_poisson7pt(int nx, int ny, int nz, dCSRmat* A, dvector* b, dvector* u);
/* config-5 synthetic input: Q1 FE for -div(diag(kx,ky,kz) grad u) = 1 (our own generator) */
int  orc_aniso27pt(int n, double kx, double ky, double kz, dCSRmat* A, dvector* b);
void orc_free_csr(dCSRmat* A);
void orc_free_vec(dvector* v);

/* BLAS-1 / SpMV: BlaArray.c, BlaSpmvCSR.c */
void   orc_mxv(const dCSRmat* A, const double* x, double* y);                  /* BlaSpmvCSR.c:242 */
void   orc_aAxpy(double alpha, const dCSRmat* A, const double* x, double* y);  /* BlaSpmvCSR.c:494 */
double orc_dotprod(int n, const double* x, const double* y);                   /* BlaArray.c:771 */
double orc_norm2(int n, const double* x);                                      /* BlaArray.c:691 */
double orc_norminf(int n, const double* x);                                    /* BlaArray.c:719 */
void   orc_axpy(int n, double a, const double* x, double* y);                  /* BlaArray.c:90 */
void   orc_axpby(int n, double a, const double* x, double b, double* y);       /* BlaArray.c:620 */

/* block (BSR) operators: BlaSpmvBSR.c:1055 / :514, BlaSparseBSR.c:543, ItrSmootherBSR.c:263 */
void    orc_bsr_mxv(const dBSRmat* A, const double* x, double* y);
void    orc_bsr_aAxpy(double alpha, const dBSRmat* A, const double* x, double* y);
double* orc_bsr_getdiaginv(const dBSRmat* A);
void    orc_bsr_jacobi1(const dBSRmat* A, const double* b, double* u, const double* diaginv);
/* ItrSmootherBSR.c:552 / :683 (block Gauss-Seidel) and :1115 / :1234 (block SOR), ascending or descending */
void    orc_bsr_gs_sor(const dBSRmat* A, const double* b, double* u, const double* diaginv, int descend, int sor, double weight);
void    orc_free(void* p);

/* smoothers: ItrSmootherCSR.c */
void orc_smoother_jacobi(double* u, int i_1, int i_n, int s, const dCSRmat* A,
                         const double* b, int L, double w);                    /* :98 */
void orc_smoother_gs(double* u, int i_1, int i_n, int s, const dCSRmat* A,
                     const double* b, int L);                                  /* :251 */
void orc_smoother_gs_cf(double* u, const dCSRmat* A, const double* b, int L,
                        const int* mark, int order);                           /* :432 */
void orc_smoother_sgs(double* u, const dCSRmat* A, const double* b, int L);    /* :808 */
void orc_smoother_sor(double* u, int i_1, int i_n, int s, const dCSRmat* A,
                      const double* b, int L, double w);                       /* :932 */
void orc_smoother_l1diag(double* u, int i_1, int i_n, int s, const dCSRmat* A,
                         const double* b, int L);                              /* :1509 */
void orc_smoother_gs_ff(double* u, const dCSRmat* A, const double* b, int L, const int* mark); /* GS on the non-C rows */
void orc_smoother_jacobi_ff(double* x, const dCSRmat* A, const double* b, int nsweeps,
                            const int* ordering, double relax);                /* :34 */
void orc_smoother_poly(const dCSRmat* A, const double* b, double* u, int n, int ndeg,
                       int L);                                                 /* ItrSmootherCSRpoly.c:67 */

/* sparse utilities used by the setup */
void orc_dcsr_trans(const dCSRmat* A, dCSRmat* AT);                            /* BlaSparseCSR.c:952 */
void orc_dcsr_rap(const dCSRmat* R, const dCSRmat* A, const dCSRmat* P,
                  dCSRmat* RAP);                                               /* BlaSpmvCSR.c:999 */

/* classical AMG setup: PreAMGSetupRS.c:52 (+ PreAMGCoarsenRS.c, PreAMGInterp.c).
 * A is deep-copied into mgl->L[0].A.  Returns FASP_SUCCESS or an ERROR_* code. */
int  orc_amg_setup_rs(orc_amg* mgl, const dCSRmat* A, AMG_param* param);
/* smoothed aggregation: PreAMGSetupSA.c:63 (smoothed P, smoothed R; VMB aggregation) */
int orc_amg_setup_ua(orc_amg* mgl, const dCSRmat* A, AMG_param* param);
int  orc_amg_setup_sa(orc_amg* mgl, const dCSRmat* A, AMG_param* param);
void orc_amg_free(orc_amg* mgl);

/* one multigrid cycle: PreMGCycle.c:48 */
void orc_mgcycle(orc_amg* mgl, const AMG_param* param);
/* one full-multigrid cycle: PreMGCycleFull.c:47 */
void orc_fmgcycle(orc_amg* mgl, const AMG_param* param);
/* z = B r with the parameter hand-over of PreCSR.c:416 (tol is NOT forwarded) */
void orc_precond_amg(orc_amg* mgl, const AMG_param* amgparam, const double* r, double* z);

/* Krylov methods.  pc == NULL means no preconditioner.  hist (may be NULL)
 * receives the recurrence residual norms ||r_k||_2 for k = 0..iters (what fasp_itinfo
 * prints) followed by ONE trailing entry: the value of `absres` at exit, i.e. the
 * recomputed true residual ||b - A u||_2 when the false-convergence check ran
 * (KryPcg.c:277-287).  *nhist = number of entries (iters + 2 on normal exit). */
typedef void (*orc_pc_fct)(double* r, double* z, void* data);
int orc_pcg(const dCSRmat* A, const dvector* b, dvector* u, orc_pc_fct pc, void* pcdata,
            double tol, double abstol, int MaxIt, int StopType, int PrtLvl,
            double* hist, int hist_cap, int* nhist, double* final_relres);      /* KryPcg.c:96 */
int orc_spcg(const dCSRmat* A, const dvector* b, dvector* u, double tol, int MaxIt,
             int StopType, int PrtLvl);                                        /* KrySPcg.c:60 (pc == NULL) */

/* mode 0 pvgmres (KryPvgmres.c:66), 1 pvfgmres (KryPvfgmres.c:67), 2 spvgmres (KrySPvgmres.c:68) */
int orc_gmres(int mode, const dCSRmat* A, const dvector* b, dvector* x, orc_pc_fct pc, void* pcdata,
              double tol, double abstol, int MaxIt, int restart, int StopType, int PrtLvl,
              double* final_relres);

/* the entry point: SolCSR.c:476 (+ SolCSR.c:56 dispatch) */
int orc_solver_dcsr_krylov_amg(dCSRmat* A, dvector* b, dvector* x, ITS_param* itparam,
                               AMG_param* amgparam, double* hist, int hist_cap, int* nhist,
                               double* final_relres);

/* Solve with an already built hierarchy (used by bench.py's cpu_baseline leg so
 * the setup is not paid twice). */
int orc_krylov_dcsr(int which, dCSRmat* A, dvector* b, dvector* x, void (*fct)(double*, double*, void*),
                    void* data, double tol, double abstol, int MaxIt, int restart, int StopType, int PrtLvl,
                    double* final_relres);
/* matrix-free family on a CSR operator (KryPcg.c:1260, KryPvgmres.c:1468, KryPvfgmres.c:1026, KryPbcgs.c:1349,
 * KryPgmres.c:1309, KryPgcg.c:213): which = 0 CG, 1 VGMRES, 2 VFGMRES, 3 BiCGstab, 4 GMRES, 6 GCG */
int orc_krylov_mf(int which, dCSRmat* A, dvector* b, dvector* x, void (*fct)(double*, double*, void*), void* data,
                  double tol, double abstol, int MaxIt, int restart, int StopType, int PrtLvl, double* final_relres);
int orc_solver_amg(dCSRmat* A, dvector* b, dvector* x, AMG_param* param, double* hist, int hist_cap,
                   int* nhist, double* final_relres);
int orc_solve_with_hierarchy(orc_amg* mgl, const dCSRmat* A, const dvector* b, dvector* x,
                             const ITS_param* itparam, const AMG_param* amgparam,
                             double* hist, int hist_cap, int* nhist, double* final_relres);

/* Build an orc_amg that *borrows* externally owned level arrays (no copies):
 * call orc_amg_borrow_begin, then orc_amg_borrow_level for l = 0..nl-1 (P and R
 * may be NULL on the last level), then orc_amg_borrow_end which allocates the
 * b/x/w work vectors.  Free with orc_amg_borrow_free. */
void orc_amg_borrow_begin(orc_amg* mgl, int num_levels);
void orc_amg_borrow_level(orc_amg* mgl, int l, const dCSRmat* A, const dCSRmat* P,
                          const dCSRmat* R, const int* cfmark);
void orc_amg_borrow_end(orc_amg* mgl);
/* cpu_baseline timing only (bench.py): pin the OpenMP team (thread t -> cpus[t mod n]) / undo it; replace the borrowed
 * level matrices by copies whose pages are first touched by the threads that will read them (static row schedule) --
 * what an OpenMP code on a multi-socket host does.  Arithmetic is untouched. */
int  orc_pin_threads(const int* cpus, int n);
void orc_unpin_threads(void);
void orc_amg_place(orc_amg* mgl);
void orc_amg_borrow_free(orc_amg* mgl);

/* BSR path (config 3): UA-AMG on the condensed matrix + block Jacobi + Krylov.
 * PreAMGSetupUABSR.c:55, PreMGCycle.c:287, PreBSR.c:1149, SolBSR.c:349 */
typedef struct {
    dBSRmat A, P, R;
    double* diaginv;
    dvector b, x, w;
} orc_bsr_level;
typedef struct {
    int           num_levels;
    orc_bsr_level L[ORC_MAX_LVL];
} orc_amg_bsr;
int  orc_amg_setup_ua_bsr(orc_amg_bsr* mgl, const dBSRmat* A, AMG_param* param);
void orc_amg_bsr_free(orc_amg_bsr* mgl);
void orc_mgcycle_bsr(orc_amg_bsr* mgl, const AMG_param* param);
int  orc_solver_dbsr_krylov_amg(dBSRmat* A, dvector* b, dvector* x, ITS_param* itparam,
                                AMG_param* amgparam, int* num_levels, double* final_relres);
int  orc_sizeof_amg_bsr(void);

int orc_sizeof_amg(void);

#ifdef __cplusplus
}
#endif
#endif
