/*
 * fasp_hip.h -- C-ABI of the MI355X-native AMG-preconditioned Krylov path.
 *
 * This header is the drop-in boundary of libfasp_hip.so.  Every type below is
 * layout-identical to the SERIAL (non-OpenMP) build of the reference headers,
 * and every `fasp_*` entry point keeps the reference's name, argument order,
 * argument meaning and return convention, so that a program compiled against
 * the reference's own `fasp.h` / `fasp_functs.h` can be linked against this
 * library instead of libfasp for the path listed here.
 *
 *   reference type / function                       reference location
 *   ------------------------------------------------------------------------
 *   INT = int, REAL = double, SHORT = short         base/include/fasp.h:71-75
 *   dCSRmat  {row,col,nnz,IA,JA,val}                base/include/fasp.h:151-180
 *   dvector / ivector                               base/include/fasp.h:354-376
 *   ITS_param                                       base/include/fasp.h:386-398
 *   AMG_param                                       base/include/fasp.h:455-595
 *   precond {data,fct}                              base/include/fasp.h:1095-1103
 *   fasp_param_amg_init                             base/src/AuxParam.c:431
 *   fasp_param_solver_init                          base/src/AuxParam.c:572
 *   fasp_solver_dcsr_krylov_amg                     base/src/SolCSR.c:476
 *   fasp_blas_dcsr_mxv / _aAxpy                     base/src/BlaSpmvCSR.c:242 / :494
 *   fasp_blas_darray_{dotprod,norm2,norminf,axpy,axpby}
 *                                                   base/src/BlaArray.c:771/691/719/90/620
 *   fasp_smoother_dcsr_jacobi                       base/src/ItrSmootherCSR.c:98
 *
 * The `fasp_hip_*` functions are extensions (no reference counterpart): they
 * expose the device-resident objects behind fasp_solver_dcsr_krylov_amg so a
 * caller can set the hierarchy up once and solve many times, bind a GPU, join
 * an RCCL communicator and read timing / roofline counters.
 *
 * No torch types, no C++ types: plain pointers and sizes only.
 */
#ifndef FASP_HIP_H
#define FASP_HIP_H

#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------ */
/* constants (values of base/include/fasp_const.h)                          */
/* ------------------------------------------------------------------------ */
#define FASP_SUCCESS            0
#define ERROR_OPEN_FILE       (-10)
#define ERROR_WRONG_FILE      (-11)
#define ERROR_INPUT_PAR       (-13)
#define ERROR_MAT_SIZE        (-15)
#define ERROR_MISC            (-19)
#define ERROR_ALLOC_MEM       (-20)
#define ERROR_DATA_STRUCTURE  (-21)
#define ERROR_AMG_INTERP_TYPE (-30)
#define ERROR_AMG_SMOOTH_TYPE (-31)
#define ERROR_AMG_COARSE_TYPE (-32)
#define ERROR_AMG_SETUP       (-39)
#define ERROR_SOLVER_TYPE     (-40)
#define ERROR_SOLVER_PRECTYPE (-41)
#define ERROR_SOLVER_STAG     (-42)
#define ERROR_SOLVER_SOLSTAG  (-43)
#define ERROR_SOLVER_TOLSMALL (-44)
#define ERROR_SOLVER_MISC     (-46)
#define ERROR_SOLVER_MAXIT    (-48)
#define ERROR_SOLVER_EXIT     (-49)
#define ERROR_UNKNOWN         (-99)

#define PRINT_NONE 0
#define PRINT_MIN  1
#define PRINT_SOME 2
#define PRINT_MORE 4
#define PRINT_MOST 8
#define PRINT_ALL  10

#define SOLVER_DEFAULT 0
#define SOLVER_CG      1
#define SOLVER_BiCGstab 2
#define SOLVER_MinRes  3
#define SOLVER_GMRES   4
#define SOLVER_VGMRES  5
#define SOLVER_VFGMRES 6
#define SOLVER_GCG     7
#define SOLVER_GCR     8

#define STOP_REL_RES     1
#define STOP_REL_PRECRES 2
#define STOP_MOD_REL_RES 3

#define PREC_NULL 0
#define PREC_DIAG 1
#define PREC_AMG  2
#define PREC_FMG  3

#define CLASSIC_AMG 1
#define SA_AMG      2
#define UA_AMG      3

#define V_CYCLE       1
#define W_CYCLE       2
#define AMLI_CYCLE    3
#define NL_AMLI_CYCLE 4
#define VW_CYCLE      12
#define WV_CYCLE      21

#define SMOOTHER_JACOBI  1
#define SMOOTHER_GS      2
#define SMOOTHER_SGS     3
#define SMOOTHER_CG      4
#define SMOOTHER_SOR     5
#define SMOOTHER_SSOR    6
#define SMOOTHER_GSOR    7
#define SMOOTHER_SGSOR   8
#define SMOOTHER_POLY    9
#define SMOOTHER_L1DIAG  10
#define SMOOTHER_JACOBIF 11
#define SMOOTHER_GSF     12

#define COARSE_RS  1
#define COARSE_RSP 2
#define COARSE_CR  3
#define COARSE_AC  4
#define COARSE_MIS 5

#define INTERP_DIR 1
#define INTERP_STD 2
#define INTERP_ENG 3
#define INTERP_RDC 4
#define INTERP_EXT 6

#define NO_ORDER 0
#define CF_ORDER 1

#define PAIRWISE 1
#define VMB      2

#define FASP_ILUk 1

/* ------------------------------------------------------------------------ */
/* types (serial layout of base/include/fasp.h; sizes checked in tests)     */
/* ------------------------------------------------------------------------ */

/* fasp.h:151  sizeof == 40 */
typedef struct dCSRmat {
    int     row;
    int     col;
    int     nnz;
    int*    IA;  /* row+1 row pointers, 0-based */
    int*    JA;  /* nnz column indices, unsorted within a row */
    double* val; /* nnz values */
} dCSRmat;

/* fasp_block.h:34  sizeof == 48.  Block CSR: ROW x COL blocks of nb x nb, NNZ stored blocks,
 * val[NNZ*nb*nb] with every block row-major (storage_manner 0). */
typedef struct dBSRmat {
    int     ROW;
    int     COL;
    int     NNZ;
    int     nb;
    int     storage_manner;
    double* val;
    int*    IA;
    int*    JA;
} dBSRmat;

/* fasp.h:354  sizeof == 16 */
typedef struct dvector {
    int     row;
    double* val;
} dvector;

/* fasp.h:368  sizeof == 16 */
typedef struct ivector {
    int  row;
    int* val;
} ivector;

/* fasp.h:386  sizeof == 40 */
typedef struct {
    short  print_level;
    short  itsolver_type;
    short  decoup_type;
    short  precond_type;
    short  stop_type;
    int    restart;
    int    maxit;
    double tol;
    double abstol;
} ITS_param;

/* fasp.h:455  sizeof == 224 */
typedef struct {
    short   AMG_type;
    short   print_level;
    int     maxit;
    double  tol;
    short   max_levels;
    int     coarse_dof;
    short   cycle_type;
    double  quality_bound;
    short   smoother;
    short   smooth_order;
    short   presmooth_iter;
    short   postsmooth_iter;
    double  relaxation;
    short   polynomial_degree;
    short   coarse_solver;
    short   coarse_scaling;
    short   amli_degree;
    double* amli_coef;
    short   nl_amli_krylov_type;
    short   coarsening_type;
    short   aggregation_type;
    short   aggregation_norm_type;
    short   interpolation_type;
    double  strong_threshold;
    double  max_row_sum;
    double  truncation_threshold;
    int     aggressive_level;
    int     aggressive_path;
    int     pair_number;
    double  strong_coupled;
    int     max_aggregation;
    double  tentative_smooth;
    short   smooth_filter;
    short   smooth_restriction;
    short   ILU_levels;
    short   ILU_type;
    int     ILU_lfil;
    double  ILU_droptol;
    double  ILU_relax;
    double  ILU_permtol;
    int     SWZ_levels;
    int     SWZ_mmsize;
    int     SWZ_maxlvl;
    int     SWZ_type;
    int     SWZ_blksolver;
    double  theta;
} AMG_param;

/* fasp.h:1095  sizeof == 16 */
typedef struct {
    void* data;
    void (*fct)(double*, double*, void*);
} precond;

/* ---- reference-layout data of the AMG preconditioner (fasp.h:596-981, serial build without UMFPACK / MUMPS /
 * PARDISO / MULTI_COLOR_ORDER).  Field order and types are the reference's, so code compiled against fasp.h and
 * code compiled against this header agree on every offset (sizes pinned in tests/golden/abi.npz:
 * AMG_data 1104 bytes, precond_data 152 bytes).  ILU / Schwarz / direct-solver members are carried for layout
 * only: those smoothers and coarse solvers are out of scope here and stay zero. ---- */
typedef struct {            /* fasp.h:404  ILU_param */
    short  print_level;
    short  ILU_type;
    int    ILU_lfil;
    double ILU_droptol;
    double ILU_relax;
    double ILU_permtol;
} ILU_param;
struct SWZ_param_;          /* fasp.h:430: only ever a pointer here */
typedef struct { int job; } Mumps_data;            /* fasp.h:596 */
typedef struct { void* pt[64]; } Pardiso_data;     /* fasp.h:623 */
typedef struct {            /* fasp.h:651  ILU_data */
    dCSRmat* A;
    int      type, row, col, nzlu;
    int*     ijlu;
    double*  luval;
    int      nb, nwork;
    double*  work;
    int*     iperm;
    int      ncolors;
    int     *ic, *icmap, *uptr;
    int      nlevL, nlevU;
    int     *ilevL, *ilevU, *jlevL, *jlevU;
} ILU_data;
typedef struct {            /* fasp.h:724  SWZ_data */
    dCSRmat  A;
    int      nblk;
    int     *iblock, *jblock;
    double*  rhsloc;
    dvector  rhsloc1, xloc1;
    double  *au, *al;
    int      SWZ_type, blk_solver, memt;
    int*     mask;
    int      maxbs;
    int*     maxa;
    dCSRmat* blk_data;
    Mumps_data*        mumps;
    struct SWZ_param_* swzparam;
} SWZ_data;
typedef struct {            /* fasp.h:804  AMG_data: one per level, mgl[0 .. num_levels) */
    short        max_levels;
    short        num_levels;
    dCSRmat      A, R, P;          /* host copies of this level's operators (this library: views of its own hierarchy) */
    dvector      b, x;             /* host vectors of this level (scratch of the reference's cycle; unused by the device cycle) */
    void*        Numeric;
    Pardiso_data pdata;
    ivector      cfmark;
    int          ILU_levels;
    ILU_data     LU;
    int          near_kernel_dim;
    double**     near_kernel_basis;
    int          SWZ_levels;
    SWZ_data     Schwarz;
    dvector      w;
    Mumps_data   mumps;
    int          cycle_type;
    int         *ic, *icmap;
    int          colors;
    double       weight;
} AMG_data;
typedef struct {            /* fasp.h:894  precond_data */
    short     AMG_type, print_level;
    int       maxit;
    short     max_levels;
    double    tol;
    short     cycle_type, smoother, smooth_order, presmooth_iter, postsmooth_iter;
    double    relaxation;
    short     polynomial_degree, coarsening_type, coarse_solver, coarse_scaling, amli_degree, nl_amli_krylov_type;
    double    tentative_smooth;
    double*   amli_coef;
    AMG_data* mgl_data;
    ILU_data* LU;
    dCSRmat  *A, *A_nk, *P_nk, *R_nk;
    dvector   r;
    double*   w;
} precond_data;

/* fasp_block.h:255  data of the block-diagonal preconditioner */
typedef struct {
    int     nb;
    dvector diag;   /* ROW inverse diagonal blocks of nb*nb doubles */
} precond_diag_bsr;

/* fasp.h:1109  matrix-free operator: y = A x through a function pointer */
typedef struct {
    void* data;
    void (*fct)(const void*, const double*, double*);
} mxv_matfree;
#define MAT_FREE 0
#define MAT_CSR  1
#define MAT_BSR  2

/* ------------------------------------------------------------------------ */
/* reference entry points kept by this library                              */
/* ------------------------------------------------------------------------ */

/* AuxParam.c:431 -- defaults: classical AMG, GS smoother with C/F order, V(1,1),
 * coarse_dof 500, max_levels 20, theta 0.3, max_row_sum 0.9, trunc 0.2 ... */
void fasp_param_amg_init(AMG_param* amgparam);

/* AuxParam.c:572 -- defaults: CG, AMG precond, STOP_REL_RES, maxit 500,
 * restart 25, tol 1e-6, abstol 1e-18 */
void fasp_param_solver_init(ITS_param* itsparam);

/* SolCSR.c:476 -- the drop-in entry point.  Host arrays in, host arrays out.
 * Deep-copies A, builds the AMG hierarchy (host), uploads it, runs the whole
 * Krylov loop on the bound MI355X, copies x back, frees everything.
 * Returns the iteration count (>=0) or a negative ERROR_* code.  Parameter
 * combinations this library has no device path for (ILU/Schwarz smoothers,
 * AMLI cycles, direct coarse solvers, ...) return an error; they never fall
 * back to a CPU solve. */
int fasp_solver_dcsr_krylov_amg(dCSRmat* A, dvector* b, dvector* x,
                                ITS_param* itparam, AMG_param* amgparam);

/* Kernel-level operators of the path, same names and semantics as the
 * reference, host pointers in/out (upload -> HIP kernel -> download).  They
 * exist so the reference's own call sites / tests of these operators can be
 * pointed at the device kernels; hot loops use the handle API below instead. */
void   fasp_blas_dcsr_mxv(const dCSRmat* A, const double* x, double* y);              /* BlaSpmvCSR.c:242 */
void   fasp_blas_dcsr_aAxpy(const double alpha, const dCSRmat* A, const double* x,
                            double* y);                                               /* BlaSpmvCSR.c:494 */
double fasp_blas_darray_dotprod(const int n, const double* x, const double* y);       /* BlaArray.c:771 */
double fasp_blas_darray_norm2(const int n, const double* x);                          /* BlaArray.c:691 */
double fasp_blas_darray_norminf(const int n, const double* x);                        /* BlaArray.c:719 */
void   fasp_blas_darray_axpy(const int n, const double a, const double* x, double* y); /* BlaArray.c:90 */
void   fasp_blas_darray_axpby(const int n, const double a, const double* x,
                              const double b, double* y);                             /* BlaArray.c:620 */
/* the remaining kernel-level names of the path, host arrays in and out */
double fasp_blas_dcsr_vmv(const dCSRmat* A, const double* x, const double* y);           /* BlaSpmvCSR.c:839  y' A x */
void   fasp_blas_dcsr_mxv_agg(const dCSRmat* A, const double* x, double* y);             /* BlaSpmvCSR.c:438  unit entries, val unread */
void   fasp_blas_dcsr_aAxpy_agg(const double alpha, const dCSRmat* A, const double* x,
                                double* y);                                              /* BlaSpmvCSR.c:727 */
void   fasp_blas_darray_ax(const int n, const double a, double* x);                      /* BlaArray.c:43 */
void   fasp_blas_darray_axpyz(const int n, const double a, const double* x,
                              const double* y, double* z);                               /* BlaArray.c:403 */
double fasp_blas_darray_norm1(const int n, const double* x);                             /* BlaArray.c:663 */
void   fasp_darray_cp(const int n, const double* x, double* y);                          /* AuxArray.c:210 */
void   fasp_darray_set(const int n, double* x, const double val);                        /* AuxArray.c:41 */
short  fasp_dvec_isnan(const dvector* u);                                                /* AuxVector.c:39 */
void   fasp_smoother_dcsr_jacobi(dvector* u, const int i_1, const int i_n, const int s,
                                 dCSRmat* A, dvector* b, int L, const double w);      /* ItrSmootherCSR.c:98 */

/* Block (BSR) operators of config 3, same conventions (host pointers, device kernels).
 * nb <= 7: every block contributes y_r += (A_r0 x_0 + A_r1 x_1 + ...), inner sum first, exactly
 * as fasp_blas_smat_ypAx (BlaSmallMat.c:779). */
void   fasp_blas_dbsr_mxv(const dBSRmat* A, const double* x, double* y);              /* BlaSpmvBSR.c:1055 */
void   fasp_blas_dbsr_aAxpy(const double alpha, const dBSRmat* A, const double* x,
                            double* y);                                               /* BlaSpmvBSR.c:514 */
dvector fasp_dbsr_getdiaginv(const dBSRmat* A);                                       /* BlaSparseBSR.c:543 (host; nb <= 3) */
void   fasp_smoother_dbsr_jacobi1(dBSRmat* A, dvector* b, dvector* u, double* diaginv); /* ItrSmootherBSR.c:263 */
/* mean milliseconds per launch of the BSR SpMV kernel on a resident copy of A (HIP events) */
double fasp_hip_time_bsr_mxv(const dBSRmat* A, int reps);

/* ini front-end: fasp_param_input (AuxInput.c:86) + fasp_param_init (AuxParam.c:34) for the two parameter
 * structs of this path -- same keywords, value formats, defaults and range check as the reference's
 * ini files (test/ini/, .dat).  fname == NULL: defaults.  Returns 0, ERROR_OPEN_FILE (-10) or
 * ERROR_INPUT_PAR; pure host code. */
int  fasp_hip_param_input(const char* fname, ITS_param* itsparam, AMG_param* amgparam);
/* Readers of the reference's ASCII data formats (base/src/BlaIO.c:164, :938, :807).  Same formats and
 * messages; an error code is returned where the reference exits.  Arrays are malloc-family
 * allocations: release with fasp_hip_free_system / fasp_hip_free_bsr / free(). */
int  fasp_dcsrvec_read2(const char* filemat, const char* filerhs, dCSRmat* A, dvector* b);
int  fasp_dvec_read(const char* filename, dvector* b);
int  fasp_dbsr_read(const char* filename, dBSRmat* A);
/* coordinate formats (BlaIO.c:332 0-based, :384 / :514 / :567 1-based, :624 one triangle of a
 * symmetric MatrixMarket file) and the writers that pair with the readers (:1388, :1623, :1145) */
int  fasp_dcoo_read(const char* filename, dCSRmat* A);
int  fasp_dcoo_read1(const char* filename, dCSRmat* A);
int  fasp_dcoo_shift_read(const char* filename, dCSRmat* A);
int  fasp_dmtx_read(const char* filename, dCSRmat* A);
int  fasp_dmtxsym_read(const char* filename, dCSRmat* A);
int  fasp_dvec_write(const char* filename, dvector* vec);
int  fasp_dcsr_write_coo(const char* filename, const dCSRmat* A);
int  fasp_dcsrvec_write2(const char* filemat, const char* filerhs, dCSRmat* A, dvector* b);
void fasp_hip_free_bsr(dBSRmat* A);
/* Fortran-style wrapper, SolWrapper.c:261: parameters from "ini/amg.dat" in the working directory */
void fasp_fwrapper_dcsr_krylov_amg_(int* n, int* nnz, int* ia, int* ja, double* a, double* b, double* u,
                                    double* tol, int* maxit, int* ptrlvl);
/* SolWrapper.c:136 (AMG as the solver) and :397 (block matrix: UA-AMG + VFGMRES) */
void fasp_fwrapper_dcsr_amg_(int* n, int* nnz, int* ia, int* ja, double* a, double* b, double* u,
                             double* tol, int* maxit, int* ptrlvl);
void fasp_fwrapper_dbsr_krylov_amg_(int* n, int* nnz, int* nb, int* ia, int* ja, double* a, double* b,
                                    double* u, double* tol, int* maxit, int* ptrlvl);

/* AMG-preconditioned Krylov solve on a block matrix -- replaces base/src/SolBSR.c:349.
 * Unsmoothed aggregation (PreAMGSetupUABSR.c:55, VMB on the condensed matrix, identity-block
 * prolongation, block Galerkin product) on the host, block-Jacobi V/W cycle
 * (PreMGCycle.c:287) with GMRES(25) on the coarsest level and PCG / VGMRES / VFGMRES
 * (KryPcg.c:386, KryPvgmres.c:416, KryPvfgmres.c:386) on the device.  Returns the iteration
 * count or a negative ERROR_* code; unsupported parameters are refused, never run on the CPU. */
int fasp_solver_dbsr_krylov_amg(dBSRmat* A, dvector* b, dvector* x, ITS_param* itparam,
                                AMG_param* amgparam);

/* Resident form of the block path (extension): setup + upload once, solve many times. */
typedef struct fasp_hip_amg_bsr fasp_hip_amg_bsr;
int  fasp_hip_bsr_amg_create(fasp_hip_amg_bsr** h, const dBSRmat* A, AMG_param* amgparam);
int  fasp_hip_bsr_amg_create_host(fasp_hip_amg_bsr** h, const dBSRmat* A, AMG_param* amgparam); /* no GPU needed */
void fasp_hip_bsr_amg_destroy(fasp_hip_amg_bsr* h);
int  fasp_hip_bsr_amg_num_levels(const fasp_hip_amg_bsr* h);
/* which: 0 = A_l, 1 = P_l, 2 = R_l; the view aliases host memory owned by the handle */
int  fasp_hip_bsr_amg_get_matrix(const fasp_hip_amg_bsr* h, int level, int which, dBSRmat* view);
const double* fasp_hip_bsr_amg_get_diaginv(const fasp_hip_amg_bsr* h, int level);

/* ------------------------------------------------------------------------ */
/* extensions: device binding, resident hierarchy, instrumentation          */
/* ------------------------------------------------------------------------ */

/* Bind the calling process to one GPU (default: device 0 on first use). */
int fasp_hip_set_device(int device);
/* Number of visible GPUs, or a negative ERROR_* code. */
int fasp_hip_device_count(void);
/* PCI bus id of the device this process is bound to into out (cap >= 16); returns the device index or a negative ERROR_* code. */
int fasp_hip_device_identity(char* out, int cap);
/* 1 when a gfx950 device is usable, else 0 (never initialises a fallback). */
int fasp_hip_available(void);

typedef struct fasp_hip_amg fasp_hip_amg; /* opaque: host + device hierarchy */

/* Host setup (classical RS, bit-compatible with PreAMGSetupRS.c:52) followed by
 * upload of every level's A/R/P to HBM.  A is deep-copied.  amgparam is
 * mutated exactly as the reference mutates it (tentative_smooth = 1.0 ...). */
int fasp_hip_amg_create(fasp_hip_amg** out, const dCSRmat* A, AMG_param* amgparam);
/* The two halves of fasp_hip_amg_create: host setup only (needs no GPU; the handle
 * can be inspected with the getters below) and the upload to the bound GPU. */
int fasp_hip_amg_create_host(fasp_hip_amg** out, const dCSRmat* A, AMG_param* amgparam);
int fasp_hip_amg_upload(fasp_hip_amg* h);
/* One host setup per node: the process that ran fasp_hip_amg_create_host publishes the host
 * hierarchy (and the parameters as the setup left them) in the POSIX shared-memory segment /<name>; the other ranks
 * of the node attach to it -- a host-only handle whose arrays are read-only views of the segment -- and call
 * fasp_hip_amg_upload, which extracts and uploads the rows they own.  unpublish removes the name once every rank has
 * attached (the mapping of an attached handle lives until fasp_hip_amg_destroy). */
int fasp_hip_amg_publish(const fasp_hip_amg* h, const char* name);
int fasp_hip_amg_attach(fasp_hip_amg** out, const char* name);
int fasp_hip_amg_unpublish(const char* name);
void fasp_hip_amg_destroy(fasp_hip_amg* h);

/* Hierarchy inspection (parity tests compare these with the oracle). */
int fasp_hip_amg_num_levels(const fasp_hip_amg* h);
/* host-only round-trip check of the lossless matrix coding applied at upload:
 * *kind = 5 row-pattern coded, 4 byte-dictionary coded, 0 stays plain CSR; returns 0 if exact */
int  fasp_hip_coding_selftest(const dCSRmat* A, int* kind);
/* kernel family of operator `which` (0 A, 1 P, 2 R) on a level -- 0: sub-wavefront per row, 2: wave-level
 * stream, 4: byte-dictionary coded, 5: row-pattern coded -- and the matrix bytes one pass of it reads */
int  fasp_hip_amg_kernel_info(const fasp_hip_amg* h, int level, int which, int* kind, double* matrix_bytes);
/* which: 0 = A_l, 1 = P_l, 2 = R_l.  Returns host views owned by the handle. */
int fasp_hip_amg_get_matrix(const fasp_hip_amg* h, int level, int which, dCSRmat* view);
int fasp_hip_amg_get_cfmark(const fasp_hip_amg* h, int level, ivector* view);

/* Per-solve statistics filled by fasp_hip_solve. */
typedef struct {
    int    iters;            /* return value of the Krylov method */
    int    nhist;            /* entries written to hist */
    double relres;           /* final relative residual (as printed by ITS_FINAL) */
    double absres;           /* final ||r||_2 */
    double normr0;           /* max(1e-20, ||r0||_2) */
    double solve_seconds;    /* wall time of the Krylov loop, device-synchronised */
    double upload_seconds;   /* H2D of b, x0 and D2H of x */
    double spmv_ms;          /* mean duration of the level-0 A*p kernel (HIP events) */
    long long spmv_launches; /* number of launches averaged in spmv_ms */
    long long coarse_iters;  /* total coarsest-level SPCG iterations */
    long long vcycles;       /* number of multigrid cycles executed */
} fasp_hip_stats;

/* Krylov solve on a resident hierarchy.  hist (may be NULL) receives the
 * absolute residual norms ||r_k||_2, k = 0..iters, up to hist_cap entries. */
int fasp_hip_solve(fasp_hip_amg* h, const dvector* b, dvector* x, const ITS_param* itparam,
                   double* hist, int hist_cap, fasp_hip_stats* stats);

int fasp_hip_bsr_solve(fasp_hip_amg_bsr* h, const dvector* b, dvector* x, const ITS_param* itparam,
                       double* hist, int hist_cap, fasp_hip_stats* stats);
/* one process per GPU (fasp_hip_comm_init before the create call): block rows of level 0 this rank owns --
 * info[0] = 1 if level 0 is whole on every rank, [1] first owned block row, [2] owned block rows, [3] ghost blocks,
 * [4] first level kept whole, [5] block size.  fasp_hip_bsr_solve takes the GLOBAL b and x; a rank fills its rows of x. */
int fasp_hip_bsr_dist_info(const fasp_hip_amg_bsr* h, int* info);

/* The same solve in three steps, for callers that keep b and x resident in HBM:
 * upload the right-hand side / initial guess (x == NULL: zeros), run the Krylov loop on
 * the resident vectors, download the solution. */
int fasp_hip_set_rhs(fasp_hip_amg* h, const dvector* b);
int fasp_hip_set_guess(fasp_hip_amg* h, const dvector* x);
int fasp_hip_solve_resident(fasp_hip_amg* h, const ITS_param* itparam, double* hist, int hist_cap,
                            fasp_hip_stats* stats);
int fasp_hip_get_solution(fasp_hip_amg* h, dvector* x);
int fasp_hip_device_synchronize(void);

/* AMG as a stand-alone solver -- replaces base/src/SolAMG.c:49 (+ PreMGSolve.c:49): setup,
 * then multigrid cycles until ||b - A x|| / ||b|| < param->tol or param->maxit cycles.
 * Returns the cycle count or a negative ERROR_* code (a failed setup returns its code; the
 * reference's CPU GMRES fallback does not exist here). */
int fasp_solver_amg(dCSRmat* A, dvector* b, dvector* x, AMG_param* param);
/* One full-multigrid cycle as the solver -- replaces base/src/SolFAMG.c:41 (void there; the status is an
 * extension).  x: initial guess in, result out.  itparam->precond_type = PREC_FMG selects the same cycle as
 * the preconditioner of fasp_solver_dcsr_krylov_amg (SolCSR.c:537). */
int fasp_solver_famg(const dCSRmat* A, const dvector* b, dvector* x, AMG_param* param);
/* the same iteration on a resident hierarchy; param == NULL: the parameters of the setup */
int fasp_hip_amg_solve(fasp_hip_amg* h, const dvector* b, dvector* x, const AMG_param* param,
                       double* hist, int hist_cap, fasp_hip_stats* stats);

/* ---- plug-in level (fasp.h:1095-1103): Krylov methods with a caller-supplied preconditioner ---- */
/* KryPcg.c:96, KryPvgmres.c:66, KryPvfgmres.c:67 -- same arguments, return values and
 * safeguards; SpMV, BLAS-1 and orthogonalisation run on the device.  pc == NULL: no
 * preconditioner.  pc->fct == fasp_hip_precond_fct: the whole iteration stays in HBM.
 * Any other pc->fct is called as a host function on host copies of r and z (one PCIe round
 * trip per application) -- e.g. a reference preconditioner (ILU, Schwarz) kept on the CPU. */
int fasp_solver_dcsr_pcg(dCSRmat* A, dvector* b, dvector* u, precond* pc, const double tol,
                         const double abstol, const int MaxIt, const short StopType, const short PrtLvl);
int fasp_solver_dcsr_pbcgs(dCSRmat* A, dvector* b, dvector* u, precond* pc, const double tol,
                           const double abstol, const int MaxIt, const short StopType, const short PrtLvl); /* KryPbcgs.c:62 */
/* Solver-level entry points next to the AMG drop-in (SolCSR.c:56/:245/:333, SolBSR.c:64/:145/:186): dispatch on
 * itsolver_type with a caller's preconditioner, without one, or with the (block-)diagonal one.  fasp_precond_diag
 * (PreCSR.c:172) and fasp_precond_dbsr_diag (PreBSR.c:49) are host functions usable anywhere a `precond` is; handed
 * to this library's Krylov methods they are recognised by their address and applied on the device. */
void fasp_precond_diag(double* r, double* z, void* data);        /* data: dvector* of diagonal entries */
void fasp_precond_dbsr_diag(double* r, double* z, void* data);   /* data: precond_diag_bsr* */
int  fasp_solver_dcsr_itsolver(dCSRmat* A, dvector* b, dvector* x, precond* pc, ITS_param* itparam);
int  fasp_solver_dcsr_krylov(dCSRmat* A, dvector* b, dvector* x, ITS_param* itparam);
int  fasp_solver_dcsr_krylov_diag(dCSRmat* A, dvector* b, dvector* x, ITS_param* itparam);
int  fasp_solver_dbsr_itsolver(dBSRmat* A, dvector* b, dvector* x, precond* pc, ITS_param* itparam);
int  fasp_solver_dbsr_krylov(dBSRmat* A, dvector* b, dvector* x, ITS_param* itparam);
int  fasp_solver_dbsr_krylov_diag(dBSRmat* A, dvector* b, dvector* x, ITS_param* itparam);

/* Matrix-free interface (SolMatFree.c:58/:157/:201; KryPcg.c:1260, KryPbcgs.c:1349, KryPgcg.c:213,
 * KryPgmres.c:1309, KryPvgmres.c:1468, KryPvfgmres.c:1026, KryPminres.c:1283 -- the reference keeps older texts of
 * CG, GMRES and MinRes for this interface; they are what runs here).  fasp_solver_matfree_init(MAT_CSR | MAT_BSR)
 * installs fasp_hip_mxv_csr / _bsr: such an operator is uploaded once and the iteration stays in HBM.
 * Any other mf->fct is called as a host function on host copies (one PCIe round trip per product).
 * fasp_solver_pminres: the reference's restart branches skip the preconditioner when one is given and dereference
 * pc == NULL otherwise (KryPminres.c:1524, :1605); here both cases take tz = tp there (krylov.hip.h). */
void fasp_hip_mxv_csr(const void* A, const double* x, double* y);
void fasp_hip_mxv_bsr(const void* A, const double* x, double* y);
void fasp_solver_matfree_init(int matrix_format, mxv_matfree* mf, void* A);
int  fasp_solver_pcg(mxv_matfree* mf, dvector* b, dvector* u, precond* pc, const double tol, const double abstol,
                     const int MaxIt, const short StopType, const short PrtLvl);
int  fasp_solver_pbcgs(mxv_matfree* mf, dvector* b, dvector* u, precond* pc, const double tol, const double abstol,
                       const int MaxIt, const short StopType, const short PrtLvl);
int  fasp_solver_pgcg(mxv_matfree* mf, dvector* b, dvector* u, precond* pc, const double tol, const double abstol,
                      const int MaxIt, const short StopType, const short PrtLvl);
int  fasp_solver_pminres(mxv_matfree* mf, dvector* b, dvector* u, precond* pc, const double tol, const double abstol,
                         const int MaxIt, const short StopType, const short PrtLvl);
int  fasp_solver_pgmres(mxv_matfree* mf, dvector* b, dvector* x, precond* pc, const double tol, const double abstol,
                        const int MaxIt, const short restart, const short StopType, const short PrtLvl);
int  fasp_solver_pvgmres(mxv_matfree* mf, dvector* b, dvector* x, precond* pc, const double tol, const double abstol,
                         const int MaxIt, short restart, const short StopType, const short PrtLvl);
int  fasp_solver_pvfgmres(mxv_matfree* mf, dvector* b, dvector* x, precond* pc, const double tol, const double abstol,
                          const int MaxIt, const short restart, const short StopType, const short PrtLvl);
int  fasp_solver_itsolver(mxv_matfree* mf, dvector* b, dvector* x, precond* pc, ITS_param* itparam);
int  fasp_solver_krylov(mxv_matfree* mf, dvector* b, dvector* x, ITS_param* itparam);

int fasp_solver_dcsr_pminres(dCSRmat* A, dvector* b, dvector* u, precond* pc, const double tol,
                             const double abstol, const int MaxIt, const short StopType, const short PrtLvl); /* KryPminres.c:61 */
int fasp_solver_dcsr_pgcg(dCSRmat* A, dvector* b, dvector* u, precond* pc, const double tol,
                          const double abstol, const int MaxIt, const short StopType, const short PrtLvl);    /* KryPgcg.c:60 */
int fasp_solver_dcsr_pgcr(dCSRmat* A, dvector* b, dvector* x, precond* pc, const double tol,
                          const double abstol, const int MaxIt, const short restart,
                          const short StopType, const short PrtLvl);                      /* KryPgcr.c:55 */
int fasp_solver_dcsr_pgmres(dCSRmat* A, dvector* b, dvector* x, precond* pc, const double tol,
                            const double abstol, const int MaxIt, const short restart,
                            const short StopType, const short PrtLvl);                    /* KryPgmres.c:66 */
int fasp_solver_dcsr_pvgmres(dCSRmat* A, dvector* b, dvector* x, precond* pc, const double tol,
                             const double abstol, const int MaxIt, const short restart,
                             const short StopType, const short PrtLvl);
int fasp_solver_dcsr_pvfgmres(dCSRmat* A, dvector* b, dvector* x, precond* pc, const double tol,
                              const double abstol, const int MaxIt, const short restart,
                              const short StopType, const short PrtLvl);
/* ---- the reference's own preconditioner objects (relink-only callers: tutorial/main/poisson-pcg.c:81,91) ----
 * fasp_precond_setup (PreCSR.c:46): PREC_AMG / PREC_FMG build the hierarchy on the host, upload it, and hand out
 * precond{data = precond_data*, fct = fasp_precond_amg | _amli | _namli | _famg}; mgl_data[l].A / P / R / cfmark show
 * the host hierarchy (read-only views owned by the library), b / x / w are allocated as the reference's setup
 * leaves them.  PREC_DIAG gives fasp_precond_diag on a copy of the diagonal, PREC_NULL returns NULL; PREC_ILU /
 * PREC_SCHWARZ are out of scope: message + exit, as fasp_chkerr does.
 * fasp_precond_amg (PreCSR.c:416): z = B r for host vectors r, z -- pcdata->maxit cycles from a zero guess, cycle /
 * smoother parameters re-read from pcdata at every call (fasp_param_prec_to_amg).  Handed to this library's Krylov
 * methods these functions are recognised by their address and the whole iteration stays in HBM.
 * fasp_amg_data_create / _free (PreDataInit.c:64 / :101): a hierarchy obtained from fasp_precond_setup is released
 * together with its device copy; an AMG_data the caller filled itself is freed field by field like the reference. */
AMG_data* fasp_amg_data_create(short max_levels);
void      fasp_amg_data_free(AMG_data* mgl, AMG_param* param);
precond*  fasp_precond_setup(const short precond_type, AMG_param* amgparam, ILU_param* iluparam, dCSRmat* A);
void      fasp_precond_amg(double* r, double* z, void* data);
void      fasp_precond_famg(double* r, double* z, void* data);    /* PreCSR.c:450 */
void      fasp_precond_amli(double* r, double* z, void* data);    /* PreCSR.c:484 */
void      fasp_precond_namli(double* r, double* z, void* data);   /* PreCSR.c:518 */
void      fasp_param_amg_to_prec(precond_data* pcdata, const AMG_param* amgparam);   /* AuxParam.c:782 */
void      fasp_param_prec_to_amg(AMG_param* amgparam, const precond_data* pcdata);   /* AuxParam.c:816 */
/* the small host utilities such callers use around the solver (AuxMemory.c:152, AuxVector.c:105/:222/:145,
 * BlaSparseCSR.c:184/:34) */
void    fasp_mem_free(void* mem);
void*   fasp_mem_calloc(const unsigned int size, const unsigned int type);
void    fasp_dvec_alloc(const int m, dvector* u);
void    fasp_dvec_set(int n, dvector* x, const double val);
void    fasp_dvec_free(dvector* u);
dvector fasp_dvec_create(const int m);
dCSRmat fasp_dcsr_create(const int m, const int n, const int nnz);
void    fasp_dcsr_free(dCSRmat* A);
/* stand-alone sweeps of the other hot-path smoothers (ItrSmootherCSR.c:251 / :932 / :1509), host vectors in and
 * out, the sweep itself on the device by level scheduling (same result as the sequential sweep) */
void fasp_smoother_dcsr_gs(dvector* u, const int i_1, const int i_n, const int s, dCSRmat* A, dvector* b, int L);
void fasp_smoother_dcsr_sor(dvector* u, const int i_1, const int i_n, const int s, dCSRmat* A, dvector* b, int L,
                            const double w);
void fasp_smoother_dcsr_L1diag(dvector* u, const int i_1, const int i_n, const int s, dCSRmat* A, dvector* b, int L);

/* The AMG preconditioner as a `precond` -- the role of fasp_precond_setup(PREC_AMG, ..)
 * (PreCSR.c:46) + fasp_precond_amg (PreCSR.c:416).  The returned object can be handed to the
 * functions above or to the REFERENCE's own CPU Krylov methods (they only call pc->fct). */
precond* fasp_hip_precond_setup(dCSRmat* A, AMG_param* amgparam);
void     fasp_hip_precond_fct(double* r, double* z, void* data);
void     fasp_hip_precond_free(precond* pc);

/* The same plug-in level for block matrices (KryPcg.c:386, KryPbcgs.c:400, KryPgmres.c:357,
 * KryPvgmres.c:416, KryPvfgmres.c:386; PreBSR.c:1149 for the preconditioner action). */
int fasp_solver_dbsr_pcg(dBSRmat* A, dvector* b, dvector* u, precond* pc, const double tol,
                         const double abstol, const int MaxIt, const short StopType, const short PrtLvl);
int fasp_solver_dbsr_pbcgs(dBSRmat* A, dvector* b, dvector* u, precond* pc, const double tol,
                           const double abstol, const int MaxIt, const short StopType, const short PrtLvl);
int fasp_solver_dbsr_pgmres(dBSRmat* A, dvector* b, dvector* x, precond* pc, const double tol,
                            const double abstol, const int MaxIt, const short restart,
                            const short StopType, const short PrtLvl);
int fasp_solver_dbsr_pvgmres(dBSRmat* A, dvector* b, dvector* x, precond* pc, const double tol,
                             const double abstol, const int MaxIt, const short restart,
                             const short StopType, const short PrtLvl);
int fasp_solver_dbsr_pvfgmres(dBSRmat* A, dvector* b, dvector* x, precond* pc, const double tol,
                              const double abstol, const int MaxIt, const short restart,
                              const short StopType, const short PrtLvl);
precond* fasp_hip_bsr_precond_setup(dBSRmat* A, AMG_param* amgparam);
void     fasp_hip_bsr_precond_fct(double* r, double* z, void* data);
void     fasp_hip_bsr_precond_free(precond* pc);

/* One application of the AMG preconditioner z = B r (PreCSR.c:416) on the
 * resident hierarchy; host vectors in/out. */
int fasp_hip_precond_amg(fasp_hip_amg* h, const double* r, double* z);

/* Synthetic input of the headline benchmark: 3-D 7-point FD Poisson on the
 * unit cube, nx*ny*nz interior points, lexicographic, x fastest; row layout and
 * values as produced by test/src/FdmPoisson.c:439 + :731 of the reference.
 * A, b, u are allocated with malloc and released by fasp_hip_free_system. */
int  fasp_hip_poisson7pt(int nx, int ny, int nz, dCSRmat* A, dvector* b, dvector* u_exact);
/* Synthetic input of config 5: Q1 trilinear FE matrix (27-point stencil) of
 * -div(diag(kx,ky,kz) grad u) = 1 on the unit cube, n^3 interior nodes.  The reference has no
 * 3-D FE generator; this one is ours.  Free with fasp_hip_free_system. */
int  fasp_hip_aniso27pt(int n, double kx, double ky, double kz, dCSRmat* A, dvector* b);
void fasp_hip_free_system(dCSRmat* A, dvector* b, dvector* u);

/* Multi-GPU (1-D row partition, RCCL over xGMI).  The unique id is produced on
 * rank 0 and distributed by the caller (e.g. torch.distributed broadcast). */
#define FASP_HIP_UNIQUE_ID_BYTES 128
int fasp_hip_comm_unique_id(char* id_out);
int fasp_hip_comm_init(int rank, int nranks, const char* id);
int fasp_hip_comm_finalize(void);
/* Validation transport: host-staged through the POSIX shared-memory segment /<name>, so
 * several processes sharing ONE GPU can run the distributed solver (tests only). */
int fasp_hip_comm_init_shm(int rank, int nranks, const char* name);
/* Peer windows (round 4; csrc/comm_ipc.h): every rank maps every peer's uncached device window through hipIpc handles exchanged
 * in the shared-memory segment /<name>; a halo exchange is one kernel that stores this rank's boundary entries straight into the
 * neighbours' mailboxes (xGMI stores between the GPUs of a node) and polls / copies its own; the Krylov scalars are reduced the
 * same way, summed in rank order.  At most 8 ranks (one node); ranks may also share one GPU (validation).  FASP_HIP_IPC_CAP =
 * doubles per mailbox (default 524 288). */
int fasp_hip_comm_init_ipc(int rank, int nranks, const char* name);
int fasp_hip_comm_rank(void);
int fasp_hip_comm_size(void);

/* Measurement and test entries (kernel timers, run-time switches of the A/B identity tests, partition inspection, self-tests):
 * include/fasp_hip_dev.h -- in the library, but not part of the drop-in boundary. */

/* Library/version string. */
const char* fasp_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FASP_HIP_H */
