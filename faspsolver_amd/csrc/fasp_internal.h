// fasp_internal.h -- internal declarations shared by the host setup and the
// device solver of libfasp_hip.so.  Not part of the public ABI (include/fasp_hip.h).
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/fasp_hip.h"
#include "../../include/fasp_hip_dev.h"

namespace fasp {

constexpr double SMALLREAL  = 1e-20;  // fasp_const.h:256
constexpr double SMALLREAL2 = 1e-40;  // fasp_const.h:257
constexpr double BIGREAL    = 1e+20;  // fasp_const.h:255
constexpr int    MIN_CDOF   = 20;     // fasp_const.h:260
constexpr int    MAX_AMG_LVL = 20;    // fasp_const.h:259
constexpr int    MAX_RESTART = 20;    // fasp_const.h:263
constexpr int    MAX_STAG    = 20;    // fasp_const.h:264
constexpr double STAG_RATIO  = 1e-4;  // fasp_const.h:265

constexpr int UNPT = -1, FGPT = 0, CGPT = 1, ISPT = 2;  // fasp_const.h:231-235

// Large host arrays ask for transparent huge pages (the setup touches tens of gigabytes of fresh memory: one fault per 2 MB
// instead of one per 4 KB); FASP_HIP_THP=0 switches the request off.  free() releases either kind.
void* buf_malloc(size_t bytes);

// Uninitialised, malloc-backed array (std::vector would zero-fill gigabytes).
template <class T>
struct Buf {
    T*     p = nullptr;
    size_t n = 0;
    bool   own = true;   // false: a view of memory owned elsewhere (a hierarchy mapped from shared memory)
    Buf() = default;
    explicit Buf(size_t n_) { alloc(n_); }
    Buf(const Buf&)            = delete;
    Buf& operator=(const Buf&) = delete;
    Buf(Buf&& o) noexcept : p(o.p), n(o.n), own(o.own) { o.p = nullptr; o.n = 0; o.own = true; }
    Buf& operator=(Buf&& o) noexcept
    {
        if (this != &o) { release(); p = o.p; n = o.n; own = o.own; o.p = nullptr; o.n = 0; o.own = true; }
        return *this;
    }
    ~Buf() { release(); }
    void release() { if (own) std::free(p); p = nullptr; n = 0; own = true; }
    void alloc(size_t n_)
    {
        release();
        n = n_;
        const size_t bytes = (n_ ? n_ : 1) * sizeof(T);
        p = static_cast<T*>(buf_malloc(bytes));
        if (!p) throw std::bad_alloc();
    }
    void view(T* q, size_t n_) { release(); p = q; n = n_; own = false; }
    void zero() { std::memset(p, 0, n * sizeof(T)); }
    void shrink(size_t n_)
    {
        if (own) {
            T* q = static_cast<T*>(std::realloc(p, (n_ ? n_ : 1) * sizeof(T)));
            if (q) p = q;
        }
        n = n_;
    }
    T&       operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
    T*       data() { return p; }
    const T* data() const { return p; }
};

struct HostCSR {
    int         row = 0, col = 0, nnz = 0;
    bool        row_aligned = false;   // row i's diagonal is column i although col > row: a rank's rows of a partitioned level (ghost columns behind the own ones)
    Buf<int>    ia, ja;
    Buf<double> val;
    dCSRmat     view() const
    {
        dCSRmat v;
        v.row = row; v.col = col; v.nnz = nnz;
        v.IA = const_cast<int*>(ia.data());
        v.JA = const_cast<int*>(ja.data());
        v.val = const_cast<double*>(val.data());
        return v;
    }
};

struct HostLevel {
    HostCSR  A, P, R;
    Buf<int> cfmark;  // C/F marker of this level (size A.row) when a coarser level exists
    bool     has_coarse = false;
};

// reorder.cpp: brick renumbering of a level (breadth-first balls of 64 rows inside chunks of `chunk` consecutive rows; order[k] = old
// index of the row that gets the new index k) and the permutation of an operator's rows / columns (entries keep their storage order)
void cluster_order(const HostCSR& A, int chunk, std::vector<int>& order);
void permute_csr(const HostCSR& A, const int* rperm, const int* cinv, HostCSR& B);

struct HostHierarchy {
    std::vector<HostLevel> L;
    double setup_seconds = 0.0;
};

// ---- 1-D row partition over the GPUs of a node (dist_plan.cpp) ----------------------
constexpr int DIST_WIN_ALIGN = 1024;   // row windows start at multiples of this (every kernel's tile size divides it)
struct DistLevel {
    bool             replicated = true;  // every rank holds (and computes) the whole level
    int              nglobal = 0;        // rows of the level
    int              row0 = 0, nloc = 0; // owned rows [row0, row0 + nloc)  (replicated: all)
    std::vector<int> start;              // nranks+1 ownership offsets of this level
    std::vector<int> ghosts;             // sorted global ids of halo entries of this level's vectors
    std::vector<int> recv_off;           // nranks+1: ghosts[recv_off[q] .. recv_off[q+1]) come from rank q
    std::vector<int> send_off;           // nranks+1 offsets into send_idx
    std::vector<int> send_idx;           // local ids (0..nloc) to send, grouped by destination rank
    HostCSR          A, P, R;            // local rows, local column numbering (distributed levels only)
    // interior windows of the local operators: rows [win[0], win[1]) read no ghost column (win[1] < 0: none worth a
    // split launch) -- they run while the halo exchange is in flight (hierarchy.hip.h, dist_launch)
    int              winA[2] = {0, -1}, winP[2] = {0, -1}, winR[2] = {0, -1};
};
// Rows of M that read no column >= nown, as the largest window [lo, hi) around the middle of the block with lo and hi
// multiples of `align` (or the row count); hi = -1 when fewer than half of the rows qualify.
void find_row_window(const HostCSR& M, int nown, int align, int win[2]);
struct DistPlan {
    int                    rank = 0, nranks = 1;
    int                    first_replicated = 0;  // levels >= this are replicated
    std::vector<DistLevel> L;
};
// Team size of the host-side OpenMP loops.  A process that does not set OMP_NUM_THREADS would start
// one thread per hardware thread (hundreds on a GPU node) in every parallel region, and libgomp rebuilds
// its pool whenever consecutive regions differ in size: measured 0.1 s PER REGION on the MI355X host,
// 28 s of setup instead of 11 s at 256^3.  The memory-bound setup loops saturate well below 32 threads.
// HostThreads pins the calling thread's nthreads ICV for the duration of a host phase and restores it.
int host_threads();  // min(omp_get_max_threads(), FASP_HIP_HOST_THREADS or 32)
struct HostThreads {
    int saved;
    HostThreads();
    ~HostThreads();
};
extern int g_parallel_min_nnz;  // host setup: threshold of the parallel (result-identical) transposes; fasp_hip_tune("host_parallel_min", n)
// Levels with fewer than min_rows rows are replicated.  Pure host code.
int build_dist_plan(const HostHierarchy& H, int rank, int nranks, int min_rows, DistPlan& D);
// The same partition for the block (BSR) hierarchy of config 3: the plan is built on the BLOCK pattern (a block row is a
// row, the aggregation hierarchy is cut into equal contiguous blocks), then the local block operators are gathered.
// local[l] = {A, P, R} of the distributed levels (empty matrices on replicated levels).
struct HostHierarchyBSR;
struct DistLocalBSR;
int build_dist_plan_bsr(const HostHierarchyBSR& H, int rank, int nranks, int min_rows, DistPlan& D, std::vector<DistLocalBSR>& local);

// Classical (Ruge-Stuben) AMG setup, host side.  Restates PreAMGSetupRS.c:52
// (+ PreAMGCoarsenRS.c, PreAMGInterp.c, BlaSparseCSR.c transposes, BlaSpmvCSR.c RAP)
// with the reference's serial arithmetic and ordering, parallelised only where the
// result is order-independent.  Returns FASP_SUCCESS or a negative ERROR_* code.
int host_setup_rs(const dCSRmat* A, AMG_param* param, HostHierarchy& H);
// When set, the classical setup reports every level as soon as its A, P, R and C/F marker are final (level l after
// its Galerkin product; the coarsest one at the end): the device upload of level l then overlaps the setup of l + 1.
extern void (*g_on_level_ready)(int level, void* ctx);
extern void (*g_on_level_matrix)(int level, void* ctx);   // classical setup: the level's A is final (P, R, cfmark are not yet); same ctx
extern void* g_on_level_ready_ctx;
// Smoothed aggregation (PreAMGSetupSA.c:63: VMB aggregation, smoothed P and R).
int host_setup_sa(const dCSRmat* A, AMG_param* param, HostHierarchy& H);
// Unsmoothed aggregation (PreAMGSetupUA.c:55: VMB aggregation, boolean P, rap_agg).
int host_setup_ua(const dCSRmat* A, AMG_param* param, HostHierarchy& H);

// ---- block (BSR) hierarchy of the unsmoothed-aggregation setup (config 3) ------------
struct HostBSR {
    int         ROW = 0, COL = 0, NNZ = 0, nb = 0;
    Buf<int>    ia, ja;
    Buf<double> val;  // NNZ blocks of nb*nb doubles, row-major (storage_manner 0)
    dBSRmat     view() const
    {
        dBSRmat v;
        v.ROW = ROW; v.COL = COL; v.NNZ = NNZ; v.nb = nb; v.storage_manner = 0;
        v.val = const_cast<double*>(val.data());
        v.IA = const_cast<int*>(ia.data());
        v.JA = const_cast<int*>(ja.data());
        return v;
    }
};
struct DistLocalBSR { HostBSR A, P, R; };   // local rows of a distributed block level, local block-column numbering
struct HostLevelBSR {
    HostBSR     A, P, R;
    Buf<double> diaginv;  // inverse diagonal blocks of A (levels that are smoothed)
    bool        has_coarse = false;
};
struct HostHierarchyBSR {
    std::vector<HostLevelBSR> L;
    double setup_seconds = 0.0;
};
// Unsmoothed aggregation on a block matrix (PreAMGSetupUABSR.c:55): VMB aggregation of the
// condensed scalar matrix, identity-block prolongation, block Galerkin product.
int host_setup_ua_bsr(const dBSRmat* A, AMG_param* param, HostHierarchyBSR& H);
// inverse diagonal blocks (BlaSparseBSR.c:543 -> fasp_smat_inv): closed forms for nb = 2, 3, 4, pivoting Gauss-Jordan for 5..7
int bsr_diaginv(const dBSRmat* A, double* out);
int check_supported_bsr(const ITS_param* itparam, const AMG_param* amgparam, int nb);

// Parameter screening: every AMG_param / ITS_param combination without a device
// path returns a negative ERROR_* code here (never a silent CPU fallback).
int check_supported(const ITS_param* itparam, const AMG_param* amgparam);

double wall_seconds();

}  // namespace fasp
