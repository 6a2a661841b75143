// host_setup.cpp -- host side of libfasp_hip.so: classical (Ruge-Stuben) AMG
// setup producing the hierarchy the HIP solve phase runs on.
//
// The setup decides the hierarchy and therefore the Krylov iteration counts, so it
// reproduces the reference's SERIAL algorithm exactly (same C/F splitting order,
// same floating-point expressions, same column order inside every row).  It is
// host code by design: the C/F splitting is an
// inherently sequential greedy graph algorithm.  Everything that is row-independent
// (strength of connection, interpolation weights, truncation, Galerkin product) runs
// under OpenMP; results are bit-identical to the serial reference because each row is
// still evaluated in the reference's order.
//
// Reference (paths relative to the reference tree):
//   fasp_amg_setup_rs          base/src/PreAMGSetupRS.c:52
//   fasp_amg_coarsening_rs     base/src/PreAMGCoarsenRS.c:76
//     strong_couplings   :236   compress_S :403   cfsplitting_cls :507
//     clean_ff_couplings :1709  form_P_pattern_dir :1891
//   fasp_amg_interp / interp_DIR / amg_interp_trunc   base/src/PreAMGInterp.c:64/302/127
//   fasp_icsr_trans / fasp_dcsr_trans                  base/src/BlaSparseCSR.c:875/952
//   fasp_blas_dcsr_rap                                 base/src/BlaSpmvCSR.c:999
//   list-of-lists helpers                              base/src/PreAMGUtil.inl:121,207
#include <omp.h>
#include <sys/mman.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>

#include "fasp_internal.h"

namespace fasp {

int host_threads()
{
    static const int cap = [] {
        const char* e = std::getenv("FASP_HIP_HOST_THREADS");
        const int v = e ? std::atoi(e) : 32;
        return v > 0 ? v : 32;
    }();
    return std::max(1, std::min(omp_get_max_threads(), cap));
}
HostThreads::HostThreads() : saved(omp_get_max_threads()) { omp_set_num_threads(std::min(saved, host_threads())); }
HostThreads::~HostThreads() { omp_set_num_threads(saved); }

int g_parallel_min_nnz = 1 << 20;  // below this many entries the order-preserving host loops stay serial

double wall_seconds()
{
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

namespace {

inline double dabs(double a) { return (a >= 0.0) ? a : -a; }  // ABS, fasp.h:84

struct Pattern {  // integer CSR (iCSRmat without values)
    int      row = 0, col = 0, nnz = 0;
    Buf<int> ia, ja;
};

// ---------------------------------------------------------------------------
// strength of connection + compression  (PreAMGCoarsenRS.c:321-384, :403-430)
// ---------------------------------------------------------------------------
// Per row i: row_sum = sum |a_ij| (diagonal included); row_scl = theta * max_{j!=i}|a_ij|;
// diagonal is never strong; if row_sum < (2 - max_row_sum)*|a_ii| the whole row is weak;
// otherwise j is weak when -a_ij <= row_scl.  a_ii = first diagonal hit, 0 if absent.
// The compressed S keeps the surviving columns in storage order.
// `strong` (one flag per stored entry of A) is the uncompressed strength pattern the reference keeps in
// iCSRmat S before compress_S; the RSP splitting needs it once more after the first pass.
void compress_strength(const HostCSR& A, const Buf<unsigned char>& strong, Pattern& S)
{
    const int row = A.row;
    const int *ia = A.ia.data(), *ja = A.ja.data();
    Buf<int> cnt((size_t)row + 1);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < row; ++i) {
        int c = 0;
        for (int j = ia[i]; j < ia[i + 1]; ++j) c += strong[j] ? 1 : 0;
        cnt[i] = c;
    }
    S.row = row; S.col = A.col;
    S.ia.alloc((size_t)row + 1);
    int acc = 0;
    for (int i = 0; i < row; ++i) { S.ia[i] = acc; acc += cnt[i]; }
    S.ia[row] = acc;
    S.nnz = acc;
    S.ja.alloc((size_t)std::max(acc, 1));
#pragma omp parallel for schedule(static)
    for (int i = 0; i < row; ++i) {
        int o = S.ia[i];
        for (int j = ia[i]; j < ia[i + 1]; ++j)
            if (strong[j]) S.ja[o++] = ja[j];
    }
}

int strength_compressed(const HostCSR& A, const AMG_param& param, Pattern& S, Buf<unsigned char>* keep = nullptr)
{
    const int    row = A.row;
    const double max_row_sum = param.max_row_sum, eps = param.strong_threshold;
    const int *  ia = A.ia.data(), *ja = A.ja.data();
    const double* aj = A.val.data();
    const int nd = std::min(A.row, A.col);

    const bool rsp = param.coarsening_type == COARSE_RSP;
    Buf<unsigned char> strong((size_t)std::max(A.nnz, 1));
    Buf<int>           cnt((size_t)row + 1);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < row; ++i) {
        const int b = ia[i], e = ia[i + 1];
        double row_scl = 0.0, row_sum = 0.0, diag = 0.0;
        bool   have_diag = false;
        for (int j = b; j < e; ++j) {
            row_sum += dabs(aj[j]);
            if (ja[j] != i) row_scl = std::max(row_scl, dabs(aj[j]));
            else if (!have_diag && i < nd) { diag = aj[j]; have_diag = true; }
        }
        row_scl *= eps;
        int  c = 0;
        const bool all_weak = row_sum < (2 - max_row_sum) * dabs(diag);
        bool diag_removed = false;
        for (int j = b; j < e; ++j) {
            bool s = true;
            if (ja[j] == i && !diag_removed) { s = false; diag_removed = true; }  // first diagonal hit only (:350-355)
            if (all_weak) s = false;
            else if (rsp ? (dabs(aj[j]) <= row_scl) : (-aj[j] <= row_scl)) s = false;  // :368-379: RSP keeps positive couplings too
            strong[j] = s;
            c += s;
        }
        cnt[i] = c;
    }
    compress_strength(A, strong, S);
    const int status = (S.nnz <= 0) ? ERROR_UNKNOWN : FASP_SUCCESS;
    if (keep) *keep = std::move(strong);
    return status;
}

// rem_positive_ff (PreAMGCoarsenRS.c:444): strong positive F-F couplings come back into the uncompressed
// pattern and the largest of each F row becomes a C point.  Sequential: vec changes as the rows go by.
void rem_positive_ff(const HostCSR& A, Buf<unsigned char>& strong, int* vec)
{
    const int *ia = A.ia.data(), *ja = A.ja.data();
    const double* av = A.val.data();
    for (int i = 0; i < A.row; ++i) {
        if (vec[i] != FGPT) continue;
        double row_scl = 0.0;
        for (int ji = ia[i]; ji < ia[i + 1]; ++ji)
            if (ja[ji] != i) row_scl = std::max(row_scl, dabs(av[ji]));
        row_scl *= 0.75;
        int max_index = -1;
        double max_entry = 0.0;
        for (int ji = ia[i]; ji < ia[i + 1]; ++ji) {
            const int j = ja[ji];
            if (j == i || vec[j] != FGPT) continue;
            if (av[ji] > row_scl) {
                strong[ji] = 1;
                if (av[ji] > max_entry) { max_entry = av[ji]; max_index = j; }
            }
        }
        if (max_index != -1) vec[max_index] = CGPT;
    }
}

// Stable counting transpose in parallel: the result is the one of the reference's serial loops
// (BlaSparseCSR.c:875 / :952 -- row j of the transpose lists its sources in increasing i), obtained
// as a two-level radix pass.  Threads own contiguous row chunks of A and drop their entries into
// buckets of 2^shift destination rows (chunks ascend in i, so a bucket fills in increasing i);
// then every bucket is counting-sorted by destination row on its own.  ia_t has m + 1 entries.
template <bool HASVAL>
void transpose_stable(int n, int m, int nnz, const int* ia, const int* ja, const double* val, int* ia_t, int* ja_t,
                      double* val_t)
{
    const int T = omp_get_max_threads();
    if (T <= 1 || nnz < g_parallel_min_nnz) {  // small: the serial text
        std::vector<int> cur((size_t)m + 2, 0);
        for (int j = 0; j < nnz; ++j) cur[(size_t)ja[j] + 2]++;
        for (int i = 2; i <= m + 1; ++i) cur[i] += cur[i - 1];
        for (int i = 0; i < n; ++i)
            for (int p = ia[i]; p < ia[i + 1]; ++p) {
                const int k = cur[(size_t)ja[p] + 1]++;
                ja_t[k] = i;
                if (HASVAL) val_t[k] = val[p];
            }
        ia_t[0] = 0;
        for (int j = 0; j < m; ++j) ia_t[j + 1] = cur[(size_t)j + 1];
        return;
    }
    int shift = 0;
    while (((m - 1) >> shift) >= 4096) ++shift;   // at most 4096 buckets
    shift = std::max(shift, 12);                  // at least 4096 rows per bucket
    const int B = ((m - 1) >> shift) + 1;
    std::vector<int> rstart((size_t)T + 1);       // row chunks balanced by entries
    for (int t = 0; t <= T; ++t) {
        const long long target = (long long)nnz * t / T;
        rstart[t] = (int)(std::lower_bound(ia, ia + n + 1, (int)target) - ia);
    }
    rstart[0] = 0; rstart[T] = n;
    for (int t = 1; t <= T; ++t) rstart[t] = std::max(rstart[t], rstart[t - 1]);
    std::vector<long long> cnt((size_t)T * B, 0);
#pragma omp parallel num_threads(T)
    {
        const int t = omp_get_thread_num();
        long long* c = cnt.data() + (size_t)t * B;
        for (int p = ia[rstart[t]]; p < ia[rstart[t + 1]]; ++p) c[ja[p] >> shift]++;
    }
    std::vector<long long> bstart((size_t)B + 1, 0);
    for (int b = 0; b < B; ++b) {
        long long run = bstart[b];
        for (int t = 0; t < T; ++t) { const long long c = cnt[(size_t)t * B + b]; cnt[(size_t)t * B + b] = run; run += c; }
        bstart[b + 1] = run;
    }
    Buf<int> tcol((size_t)nnz), trow((size_t)nnz);
    Buf<double> tval(HASVAL ? (size_t)nnz : 1);
#pragma omp parallel num_threads(T)
    {
        const int t = omp_get_thread_num();
        long long* off = cnt.data() + (size_t)t * B;
        for (int i = rstart[t]; i < rstart[t + 1]; ++i)
            for (int p = ia[i]; p < ia[i + 1]; ++p) {
                const int col = ja[p];
                const long long k = off[col >> shift]++;
                tcol[(size_t)k] = col; trow[(size_t)k] = i;
                if (HASVAL) tval[(size_t)k] = val[p];
            }
    }
#pragma omp parallel num_threads(T)
    {
        std::vector<int> cur((size_t)(1 << shift) + 1);
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; ++b) {
            const int c0 = b << shift, c1 = std::min(m, (b + 1) << shift), w = c1 - c0;
            std::fill(cur.begin(), cur.begin() + w + 1, 0);
            for (long long k = bstart[b]; k < bstart[b + 1]; ++k) cur[(size_t)(tcol[(size_t)k] - c0) + 1]++;
            int run = (int)bstart[b];
            for (int j = 0; j < w; ++j) { const int c = cur[(size_t)j + 1]; ia_t[c0 + j] = run; cur[j] = run; run += c; }
            for (long long k = bstart[b]; k < bstart[b + 1]; ++k) {
                const int pos = cur[(size_t)(tcol[(size_t)k] - c0)]++;
                ja_t[pos] = trow[(size_t)k];
                if (HASVAL) val_t[pos] = tval[(size_t)k];
            }
        }
    }
    ia_t[m] = nnz;
}

// stable counting transpose of a pattern (BlaSparseCSR.c:875-936): row j of the
// transpose lists its sources in increasing i.
void transpose_pattern(const Pattern& A, Pattern& AT)
{
    const int n = A.row, m = A.col, nnz = A.nnz;
    AT.row = m; AT.col = n; AT.nnz = nnz;
    AT.ia.alloc((size_t)m + 2);
    AT.ja.alloc((size_t)std::max(nnz, 1));
    transpose_stable<false>(n, m, nnz, A.ia.data(), A.ja.data(), nullptr, AT.ia.data(), AT.ja.data(), nullptr);
}

// ---------------------------------------------------------------------------
// bucket lists by measure: one FIFO per measure value, append at the tail
// (enter_list, PreAMGUtil.inl:264-271), the splitting takes the head of the highest
// non-empty list (PreAMGCoarsenRS.c:654).  Arrays instead of heap nodes: O(1) updates.
// The splitting is a sequential greedy pass that hops between vertices; list links, measure
// and marker of a vertex share one 16-byte record, so a hop costs one cache line, not four.
// ---------------------------------------------------------------------------
struct CfNode { int next, prev, lambda, vec; };
struct Buckets {
    std::vector<int> head, tail;
    CfNode*          nd;
    int              cur_max = 0;
    Buckets(CfNode* nodes, int cap) : head(cap + 1, -1), tail(cap + 1, -1), nd(nodes) {}
    void enter(int m, int v)
    {
        if (m >= (int)head.size()) { head.resize(2 * m + 2, -1); tail.resize(2 * m + 2, -1); }
        nd[v].next = -1;
        nd[v].prev = tail[m];
        if (tail[m] >= 0) nd[tail[m]].next = v; else head[m] = v;
        tail[m] = v;
        if (m > cur_max) cur_max = m;
    }
    void remove(int m, int v)
    {
        const int p = nd[v].prev, n = nd[v].next;
        if (p >= 0) nd[p].next = n; else head[m] = n;
        if (n >= 0) nd[n].prev = p; else tail[m] = p;
    }
    int top()
    {
        while (cur_max > 0 && head[cur_max] < 0) --cur_max;
        return head[cur_max];
    }
};

// C/F splitting, first pass + C1 second pass (PreAMGCoarsenRS.c:507-785, RS_C1 ON)
// a_rowlen != nullptr selects cfsplitting_clsp (:806): "isolated" = a matrix row of at most one entry, no C1 pass
int cfsplitting_cls(const Pattern& S, int* vec, const int* a_ia = nullptr)
{
    const int row = S.row;
    int       col = 0, num_left = 0;
    Pattern   ST;
    static const bool timing = std::getenv("FASP_HIP_SETUP_TIMING") != nullptr;
    double tp = wall_seconds();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const double now = wall_seconds();
        std::printf("      [C/F] %-12s %8.3f s\n", what, now - tp);
        tp = now;
    };
    transpose_pattern(S, ST);
    lap("S^T");

    Buf<CfNode> nodes((size_t)std::max(row, 1));
    CfNode* nd = nodes.data();
    int maxdeg = 0;
#pragma omp parallel for schedule(static) reduction(max : maxdeg) reduction(+ : num_left)
    for (int i = 0; i < row; ++i) {
        CfNode x;
        x.next = x.prev = -1;
        x.lambda = ST.ia[i + 1] - ST.ia[i];
        maxdeg = std::max(maxdeg, x.lambda);
        const bool isolated = a_ia ? ((a_ia[i + 1] - a_ia[i]) <= 1) : (S.ia[i + 1] == S.ia[i]);
        if (isolated) { x.vec = ISPT; x.lambda = 0; }
        else { x.vec = UNPT; ++num_left; }
        nd[i] = x;
    }
    Buckets B(nd, 2 * maxdeg + 2);

    for (int i = 0; i < row; ++i) {  // :614-648
        if (nd[i].vec == ISPT) continue;
        const int measure = nd[i].lambda;
        if (measure > 0) {
            B.enter(measure, i);
        } else {
            if (measure < 0) std::printf("### WARNING: Negative lambda[%d]!\n", i);
            nd[i].vec = FGPT;
            --num_left;
            for (int k = S.ia[i]; k < S.ia[i + 1]; ++k) {
                const int j = S.ja[k];
                if (nd[j].vec == ISPT) continue;
                if (j < i) {
                    if (nd[j].lambda > 0) B.remove(nd[j].lambda, j);
                    B.enter(++nd[j].lambda, j);
                } else {
                    ++nd[j].lambda;
                }
            }
        }
    }

    lap("fill");
    // (measured at 256^3, same box, on / off: level 1 -- 19 couplings per row -- 1.30 against 1.85 s; level 0 -- 7 per row -- no gain:
    // the requests cost what they save there)
    static const bool prefetch_on = !(std::getenv("FASP_HIP_SETUP_PREFETCH") && std::atoi(std::getenv("FASP_HIP_SETUP_PREFETCH")) == 0);
    const bool prefetch = prefetch_on && (long long)S.nnz > 7ll * row;
    while (num_left > 0) {  // :651-717
        const int maxnode = B.top();
        const int maxmeas = nd[maxnode].lambda;
        if (maxmeas == 0) std::printf("### WARNING: Head of the list has measure 0!\n");
        nd[maxnode].vec    = CGPT;
        nd[maxnode].lambda = 0;
        --num_left;
        B.remove(maxmeas, maxnode);
        ++col;

        // The pass hops between vertices whose records are cache misses (268 MB of them at 256^3).  Which records a step
        // touches is known before it touches them: the neighbours of the new C point, their neighbours, and the list
        // neighbours of every vertex that changes lists -- requested up front, level by level, so the misses of a step
        // overlap instead of queueing behind one another.  (Prefetches only: the pass itself is untouched.)
        const int sb = ST.ia[maxnode], se = ST.ia[maxnode + 1], cb = S.ia[maxnode], ce = S.ia[maxnode + 1];
        if (prefetch) {
        for (int i = sb; i < se; ++i) { __builtin_prefetch(&nd[ST.ja[i]]); __builtin_prefetch(&S.ia[ST.ja[i]]); }
        for (int i = cb; i < ce; ++i) { __builtin_prefetch(&nd[S.ja[i]]); __builtin_prefetch(&S.ia[S.ja[i]]); }
        for (int i = sb; i < se; ++i) {
            const int j = ST.ja[i];
            if (nd[j].vec != UNPT) continue;
            __builtin_prefetch(&S.ja[S.ia[j]]);
            if (nd[j].prev >= 0) __builtin_prefetch(&nd[nd[j].prev]);
            if (nd[j].next >= 0) __builtin_prefetch(&nd[nd[j].next]);
        }
        for (int i = sb; i < se; ++i) {
            const int j = ST.ja[i];
            if (nd[j].vec != UNPT) continue;
            for (int l = S.ia[j]; l < S.ia[j + 1]; ++l) __builtin_prefetch(&nd[S.ja[l]]);
        }
        }
        for (int i = sb; i < se; ++i) {
            const int j = ST.ja[i];
            if (nd[j].vec != UNPT) continue;
            nd[j].vec = FGPT;
            B.remove(nd[j].lambda, j);
            --num_left;
            const int lb = S.ia[j], le = S.ia[j + 1];
            for (int l = lb; prefetch && l < le; ++l) {   // (list neighbours of the vertices about to move, and the tails they move behind)
                const int k = S.ja[l];
                if (nd[k].vec != UNPT) continue;
                if (nd[k].prev >= 0) __builtin_prefetch(&nd[nd[k].prev]);
                if (nd[k].next >= 0) __builtin_prefetch(&nd[nd[k].next]);
                const int tm = nd[k].lambda + 1;
                if (tm < (int)B.tail.size() && B.tail[tm] >= 0) __builtin_prefetch(&nd[B.tail[tm]]);
            }
            for (int l = lb; l < le; ++l) {
                const int k = S.ja[l];
                if (nd[k].vec == UNPT) {
                    B.remove(nd[k].lambda, k);
                    B.enter(++nd[k].lambda, k);
                }
            }
        }
        for (int i = cb; i < ce; ++i) {
            const int j = S.ja[i];
            if (nd[j].vec != UNPT) continue;
            int measure = nd[j].lambda;
            B.remove(measure, j);
            nd[j].lambda = --measure;
            if (measure > 0) {
                B.enter(measure, j);
            } else {
                nd[j].vec = FGPT;
                --num_left;
                for (int l = S.ia[j]; l < S.ia[j + 1]; ++l) {
                    const int k = S.ja[l];
                    if (nd[k].vec == UNPT) {
                        B.remove(nd[k].lambda, k);
                        B.enter(++nd[k].lambda, k);
                    }
                }
            }
        }
    }

    lap("first pass");
    // C1 criterion, :719-763 (graph_array re-uses the measure field)
#pragma omp parallel for schedule(static)
    for (int i = 0; i < row; ++i) { vec[i] = nd[i].vec; nd[i].lambda = -1; }
    int jkeep = 0;
    for (int i = 0; i < (a_ia ? 0 : row); ++i) {
        if (vec[i] != FGPT) continue;
        const int e = S.ia[i + 1];
        for (int ji = S.ia[i]; ji < e; ++ji) {
            const int j = S.ja[ji];
            if (vec[j] == CGPT) nd[j].lambda = i;
        }
        int cnt = 0;
        for (int ji = S.ia[i]; ji < e; ++ji) {
            const int j = S.ja[ji];
            if (vec[j] != FGPT) continue;
            bool set_empty = true;
            for (int jj = S.ia[j]; jj < S.ia[j + 1]; ++jj)
                if (nd[S.ja[jj]].lambda == i) { set_empty = false; break; }
            if (set_empty) {
                if (cnt == 0) {
                    vec[j] = CGPT;
                    ++col;
                    nd[j].lambda = i;
                    jkeep = j;
                    cnt = 1;
                } else {
                    vec[i]     = CGPT;
                    vec[jkeep] = FGPT;
                    break;
                }
            }
        }
    }
    lap("C1 pass");
    return col;
}

// Couplings between the temporary C points for aggressive coarsening (strong_couplings_agg1 / _agg2,
// PreAMGCoarsenRS.c:1065 / :1243): path 1 = a strong path of length <= 2, path 2 = a direct coupling or at
// least two paths of length 2.  The reference's marks are per source point (ci / ci+1 / -ci-1), so rows are
// independent: count and fill run in parallel with a private mark array, column order as in the reference.
static void couplings_between_c(const Pattern& S, const int* vec, bool two_paths, std::vector<int>& cp_index,
                                Pattern& Sh)
{
    const int row = S.row;
    std::vector<int> cp_rindex((size_t)std::max(row, 1));
    cp_index.clear();
    for (int i = 0; i < row; ++i)
        if (vec[i] == CGPT) { cp_rindex[i] = (int)cp_index.size(); cp_index.push_back(i); }
    const int num_c = (int)cp_index.size();
    Sh.row = Sh.col = num_c;
    Sh.ia.alloc((size_t)num_c + 2);
    Sh.ia[0] = 0;

    // visit the couplings of C point ci in the reference's order; emit(c) is called once per coupled C point
    auto walk = [&](int ci, int* visited, auto&& emit) {
        const int i = cp_index[ci], hit = two_paths ? ci + 1 : ci;
        for (int j = S.ia[i]; j < S.ia[i + 1]; ++j) {
            const int fj = S.ja[j];
            if (vec[fj] == CGPT && fj != i) {
                const int cj = cp_rindex[fj];
                if (visited[cj] != hit) { visited[cj] = hit; emit(cj); }
            } else if (vec[fj] == FGPT) {
                for (int k = S.ia[fj]; k < S.ia[fj + 1]; ++k) {
                    const int ck = S.ja[k];
                    if (vec[ck] != CGPT || ck == i) continue;
                    const int cck = cp_rindex[ck];
                    if (!two_paths) {
                        if (visited[cck] != hit) { visited[cck] = hit; emit(cck); }
                    } else if (visited[cck] == hit) {
                    } else if (visited[cck] == -hit) {  // second path
                        visited[cck] = hit;
                        emit(cck);
                    } else {
                        visited[cck] = -hit;
                    }
                }
            }
        }
    };
    const int init = two_paths ? 0 : -1;
#pragma omp parallel
    {
        std::vector<int> visited((size_t)std::max(num_c, 1), init);
#pragma omp for schedule(static)
        for (int ci = 0; ci < num_c; ++ci) {
            int count = 0;
            walk(ci, visited.data(), [&](int) { ++count; });
            Sh.ia[ci + 1] = count;
        }
    }
    for (int ci = 0; ci < num_c; ++ci) Sh.ia[ci + 1] += Sh.ia[ci];
    Sh.nnz = Sh.ia[num_c];
    Sh.ja.alloc((size_t)std::max(Sh.nnz, 1));
#pragma omp parallel
    {
        // the fill pass starts from fresh marks like the reference's (:1168 / :1353); a row's marks from the
        // counting pass would otherwise suppress its own entries
        std::vector<int> visited((size_t)std::max(num_c, 1), init);
#pragma omp for schedule(static)
        for (int ci = 0; ci < num_c; ++ci) {
            int o = Sh.ia[ci];
            walk(ci, visited.data(), [&](int c) { Sh.ja[o++] = c; });
        }
    }
}

// Aggressive coarsening (cfsplitting_agg, PreAMGCoarsenRS.c:1435-1684): the classical splitting, then a
// second greedy pass over its C points with the couplings above (real C points 3, fake ones 4 in the
// reference; :1543/:1557/:1589), then F points without a C point within distance two become C points.
int cfsplitting_agg(const Pattern& S, int* vec, int aggressive_path)
{
    enum { REAL_C = 3, FAKE_C = 4 };
    const int row = S.row;
    int col = 0, num_left = 0;
    cfsplitting_cls(S, vec);

    std::vector<int> cp_index;
    Pattern Sh, ShT;
    couplings_between_c(S, vec, aggressive_path >= 2, cp_index, Sh);
    transpose_pattern(Sh, ShT);
    const int num_c = Sh.row;

    Buf<CfNode> nodes((size_t)std::max(num_c, 1));
    CfNode* nd = nodes.data();
    int maxdeg = 0;
    for (int ci = 0; ci < num_c; ++ci) {
        nd[ci].next = nd[ci].prev = -1;
        nd[ci].lambda = ShT.ia[ci + 1] - ShT.ia[ci];
        nd[ci].vec = CGPT;
        maxdeg = std::max(maxdeg, nd[ci].lambda);
    }
    Buckets B(nd, 2 * maxdeg + 2);

    for (int ci = 0; ci < num_c; ++ci) {  // :1492-1532 (num_left counts the listed points here)
        if (nd[ci].lambda > 0) {
            B.enter(nd[ci].lambda, ci);
            ++num_left;
        } else {
            nd[ci].vec = FGPT;
            for (int k = Sh.ia[ci]; k < Sh.ia[ci + 1]; ++k) {
                const int cj = Sh.ja[k];
                if (cj < ci) {  // (a point already set to F is listed again, as in the reference)
                    if (nd[cj].lambda > 0) { B.remove(nd[cj].lambda, cj); --num_left; }
                    B.enter(++nd[cj].lambda, cj);
                    ++num_left;
                } else {
                    ++nd[cj].lambda;
                }
            }
        }
    }

    auto bump_neighbours = [&](int cj) {
        for (int l = Sh.ia[cj]; l < Sh.ia[cj + 1]; ++l) {
            const int ck = Sh.ja[l];
            if (nd[ck].vec == CGPT) {
                B.remove(nd[ck].lambda, ck);
                B.enter(++nd[ck].lambda, ck);
            }
        }
    };
    while (num_left > 0) {  // :1535-1604
        const int maxnode = B.top();
        const int maxmeas = nd[maxnode].lambda;
        if (maxmeas == 0) std::printf("### WARNING: Head of the list has measure 0!\n");
        nd[maxnode].vec = REAL_C;
        --num_left;
        B.remove(maxmeas, maxnode);
        nd[maxnode].lambda = 0;
        ++col;
        for (int i = ShT.ia[maxnode]; i < ShT.ia[maxnode + 1]; ++i) {
            const int cj = ShT.ja[i];
            if (nd[cj].vec != CGPT) continue;
            nd[cj].vec = FAKE_C;
            B.remove(nd[cj].lambda, cj);
            --num_left;
            bump_neighbours(cj);
        }
        for (int i = Sh.ia[maxnode]; i < Sh.ia[maxnode + 1]; ++i) {
            const int cj = Sh.ja[i];
            if (nd[cj].vec != CGPT) continue;
            int measure = nd[cj].lambda;
            B.remove(measure, cj);
            nd[cj].lambda = --measure;
            if (measure > 0) {
                B.enter(measure, cj);
            } else {
                nd[cj].vec = FAKE_C;
                --num_left;
                bump_neighbours(cj);
            }
        }
    }
    for (int ci = 0; ci < num_c; ++ci) vec[cp_index[ci]] = (nd[ci].vec == REAL_C) ? CGPT : FGPT;  // :1611-1620

    for (int i = 0; i < row; ++i) {  // :1628-1660; sequential: a promoted point serves the rows after it
        if (vec[i] != FGPT) continue;
        bool has_c = false;
        for (int j = S.ia[i]; j < S.ia[i + 1] && !has_c; ++j) {
            const int k = S.ja[j];
            if (vec[k] == CGPT) has_c = true;
            else if (vec[k] == FGPT)
                for (int l = S.ia[k]; l < S.ia[k + 1]; ++l)
                    if (vec[S.ja[l]] == CGPT) { has_c = true; break; }
        }
        if (!has_c) { vec[i] = CGPT; ++col; }
    }
    return col;
}

// COARSE_MIS (PreAMGCoarsenRS.c:120-128): greedy maximal independent set of the strength graph (cfsplitting_mis,
// :2127) in the order of ordering1 (:2179): natural order, the first vertex of highest degree swapped to the front.
int cfsplitting_mis(const Pattern& S, int* vec)
{
    const int n = S.row;
    int col = 0, maxind = 0, maxdeg = 0;
    for (int i = 0; i < n; ++i) {
        const int degree = S.ia[i + 1] - S.ia[i];
        if (degree > maxdeg) { maxind = i; maxdeg = degree; }
    }
    std::fill(vec, vec + n, (int)UNPT);
    for (int i = 0; i < n; ++i) {
        const int ind = (i == 0) ? maxind : (i == maxind ? 0 : i);
        if (vec[ind] != UNPT) continue;
        bool c_neighbour = false;
        for (int j = S.ia[ind]; j < S.ia[ind + 1]; ++j)
            if (vec[S.ja[j]] == CGPT) { c_neighbour = true; break; }
        if (c_neighbour) { vec[ind] = FGPT; continue; }
        vec[ind] = CGPT;
        ++col;
        for (int j = S.ia[ind]; j < S.ia[ind + 1]; ++j) vec[S.ja[j]] = FGPT;
    }
    return col;
}

// F-F couplings without a common C point (PreAMGCoarsenRS.c:1709-1781), with the
// reference's "tentatively promote j, re-check i" roll-back (:1762-1769).
int clean_ff_couplings(const Pattern& S, int* vec, int row, int col)
{
    std::vector<int> cindex(row, -1);
    bool C_i_nonempty = false;
    int  ci_tilde = -1, ci_tilde_mark = -1;
    for (int i = 0; i < row; ++i) {
        if (vec[i] != FGPT) continue;
        for (int ji = S.ia[i]; ji < S.ia[i + 1]; ++ji) {
            const int j = S.ja[ji];
            cindex[j] = (vec[j] == CGPT) ? i : -1;
        }
        if (ci_tilde_mark != i) ci_tilde = -1;
        for (int ji = S.ia[i]; ji < S.ia[i + 1]; ++ji) {
            const int j = S.ja[ji];
            if (vec[j] != FGPT) continue;
            bool set_empty = true;
            for (int jj = S.ia[j]; jj < S.ia[j + 1]; ++jj)
                if (cindex[S.ja[jj]] == i) { set_empty = false; break; }
            if (!set_empty) continue;
            if (C_i_nonempty) {
                vec[i] = CGPT;
                ++col;
                if (ci_tilde > -1) {
                    vec[ci_tilde] = FGPT;
                    --col;
                    ci_tilde = -1;
                }
                C_i_nonempty = false;
                break;
            } else {
                vec[j] = CGPT;
                ++col;
                ci_tilde      = j;
                ci_tilde_mark = i;
                C_i_nonempty  = true;
                --i;  // re-check row i
                break;
            }
        }
    }
    return col;
}

// ---------------------------------------------------------------------------
// interpolation: pattern (PreAMGCoarsenRS.c:1891-1985), direct interpolation weights
// (PreAMGInterp.c:411-487), coarse numbering (:491-517), truncation (:127-228)
// ---------------------------------------------------------------------------
void finish_interp(int row, const int* vec, const Buf<int>& pia, const Buf<int>& pja, const Buf<double>& pval,
                   double eps_tr, HostCSR& P);

// one row of the standard-interpolation pattern (form_P_pattern_std, PreAMGCoarsenRS.c:2006): strong C
// neighbours and the strong C neighbours of strong F neighbours, in discovery order
static inline void std_pattern_row(const Pattern& S, const int* vec, int i, std::vector<int>& cols)
{
    cols.clear();
    if (vec[i] == FGPT) {
        auto add = [&](int h) {
            for (int c : cols) if (c == h) return;
            cols.push_back(h);
        };
        for (int j = S.ia[i]; j < S.ia[i + 1]; ++j) {
            const int k = S.ja[j];
            if (vec[k] == CGPT) add(k);
            else if (vec[k] == FGPT && k != i)
                for (int l = S.ia[k]; l < S.ia[k + 1]; ++l) { const int h = S.ja[l]; if (vec[h] == CGPT) add(h); }
        }
    } else if (vec[i] == CGPT) {
        cols.push_back(i);
    }
}

// std_pattern: direct-interpolation WEIGHTS on the standard pattern -- what the reference computes on the level
// where aggressive coarsening ends (the pattern is formed while coarsening_type is still COARSE_AC,
// PreAMGCoarsenRS.c:96, the weights after it was switched back, PreAMGSetupRS.c:198-199 / PreAMGInterp.c:68-71)
void build_interp_dir(const HostCSR& A, const Pattern& S, const int* vec, const AMG_param& param,
                      HostCSR& P, bool std_pattern = false)
{
    const int row = A.row;
    const int *ia = A.ia.data(), *ja = A.ja.data();
    const double* av = A.val.data();
    const double eps_tr = param.truncation_threshold;

    // pattern of the untruncated P, in fine-column indices
    Buf<int> pia((size_t)row + 1);
    pia[0] = 0;
#pragma omp parallel
    {
        std::vector<int> cols;
#pragma omp for schedule(static)
        for (int i = 0; i < row; ++i) {
            int c = 0;
            if (std_pattern) {
                std_pattern_row(S, vec, i, cols);
                c = (int)cols.size();
            } else if (vec[i] == FGPT) {
                for (int j = S.ia[i]; j < S.ia[i + 1]; ++j) c += (vec[S.ja[j]] == CGPT);
            } else if (vec[i] == CGPT) {
                c = 1;
            }
            pia[i + 1] = c;
        }
    }
    for (int i = 0; i < row; ++i) pia[i + 1] += pia[i];
    const int   pnnz = pia[row];
    Buf<int>    pja((size_t)std::max(pnnz, 1));
    Buf<double> pval((size_t)std::max(pnnz, 1));
#pragma omp parallel
    {
    std::vector<int> cols;
#pragma omp for schedule(static)
    for (int i = 0; i < row; ++i) {
        int o = pia[i];
        if (std_pattern) {
            std_pattern_row(S, vec, i, cols);
            for (int c : cols) pja[o++] = c;
        } else if (vec[i] == FGPT) {
            for (int j = S.ia[i]; j < S.ia[i + 1]; ++j) {
                const int k = S.ja[j];
                if (vec[k] == CGPT) pja[o++] = k;
            }
        } else if (vec[i] == CGPT) {
            pja[o++] = i;
        }
    }
    }

    // The reference keeps `aii` in a function-scope variable (PreAMGInterp.c:314): a row
    // without a diagonal entry inherits the previous row's (possibly modified) value.
    // Rows are independent only when every row has a diagonal; otherwise run serially.
    bool all_diag = true;
#pragma omp parallel for schedule(static) reduction(&& : all_diag)
    for (int i = 0; i < row; ++i) {
        bool f = false;
        for (int k = ia[i]; k < ia[i + 1]; ++k)
            if (ja[k] == i) { f = true; break; }
        all_diag = all_diag && f;
    }

    auto weights_row = [&](int i, double& aii, int* mark) {
        const int b = ia[i], e = ia[i + 1];
        int idiag = b;
        for (; idiag < e; ++idiag)
            if (ja[idiag] == i) { aii = av[idiag]; break; }
        if (vec[i] == FGPT) {
            // mark[c] == i  <=>  c is a pattern (strong C) column of row i; the reference
            // searches P's row linearly (:436-442) -- same predicate.
            if (mark)
                for (int k = pia[i]; k < pia[i + 1]; ++k) mark[pja[k]] = i;
            double amN = 0.0, amP = 0.0, apN = 0.0, apP = 0.0;
            int    num_pcouple = 0;
            for (int j = b; j < e; ++j) {
                if (j == idiag) continue;
                bool is_strong = false;
                if (mark) {
                    is_strong = (mark[ja[j]] == i);
                } else {
                    for (int k = pia[i]; k < pia[i + 1]; ++k)
                        if (pja[k] == ja[j]) { is_strong = true; break; }
                }
                if (av[j] > 0) {
                    apN += av[j];
                    if (is_strong) { apP += av[j]; ++num_pcouple; }
                } else {
                    amN += av[j];
                    if (is_strong) amP += av[j];
                }
            }
            amP = (amP < -SMALLREAL) ? amP : -SMALLREAL;
            apP = (apP > SMALLREAL) ? apP : SMALLREAL;
            const double alpha = amN / amP;
            double       beta;
            if (num_pcouple > 0) {
                beta = apN / apP;
            } else {
                beta = 0.0;
                aii += apN;
            }
            for (int j = pia[i]; j < pia[i + 1]; ++j) {
                const int k = pja[j];
                int       l = b;
                for (; l < e; ++l)
                    if (ja[l] == k) break;
                // (standard pattern only) a column absent from the row: the reference reads the entry just past
                // the row (:481-486) = the next row's first entry; past the array's end that is undefined there, 0 here
                const double ail = (l < A.nnz) ? av[l] : 0.0;
                if (ail > 0) pval[j] = -beta * ail / aii;
                else pval[j] = -alpha * ail / aii;
            }
        } else if (vec[i] == CGPT) {
            pval[pia[i]] = 1.0;
        }
    };

    // marker arrays only pay off for long rows; short rows use the reference's linear search
    const bool use_mark = (double)A.nnz / std::max(row, 1) > 16.0;
    if (all_diag) {
#pragma omp parallel
        {
            std::vector<int> mark;
            if (use_mark) mark.assign(A.col, -1);
#pragma omp for schedule(static)
            for (int i = 0; i < row; ++i) {
                double aii = 0.0;
                weights_row(i, aii, use_mark ? mark.data() : nullptr);
            }
        }
    } else {
        std::vector<int> mark;
        if (use_mark) mark.assign(A.col, -1);
        double aii = 0.0;
        for (int i = 0; i < row; ++i) weights_row(i, aii, use_mark ? mark.data() : nullptr);
    }

    finish_interp(row, vec, pia, pja, pval, eps_tr, P);
}

// coarse numbering (PreAMGInterp.c:491-517) and truncation (:127-228) of an interpolation given in
// fine-column indices
void finish_interp(int row, const int* vec, const Buf<int>& pia, const Buf<int>& pja, const Buf<double>& pval,
                   double eps_tr, HostCSR& P)
{
    // coarse numbering: C points in increasing fine index (:491-493)
    std::vector<int> cindex(row, 0);
    int              ncoarse = 0;
    for (int i = 0; i < row; ++i)
        if (vec[i] == CGPT) cindex[i] = ncoarse++;

    // truncation (:127-228): two passes so rows can run in parallel; values per row are
    // evaluated exactly as the reference's in-place sweep.
    Buf<int> tia((size_t)row + 1);
    tia[0] = 0;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < row; ++i) {
        double Min_neg = 0, Max_pos = 0;
        for (int j = pia[i]; j < pia[i + 1]; ++j) {
            if (pval[j] > 0) Max_pos = std::max(Max_pos, pval[j]);
            else Min_neg = std::min(Min_neg, pval[j]);
        }
        Max_pos *= eps_tr;
        Min_neg *= eps_tr;
        int c = 0;
        for (int j = pia[i]; j < pia[i + 1]; ++j)
            if (pval[j] >= Max_pos || pval[j] <= Min_neg) ++c;
        tia[i + 1] = c;
    }
    for (int i = 0; i < row; ++i) tia[i + 1] += tia[i];
    const int tnnz = tia[row];
    P.row = row; P.col = ncoarse; P.nnz = tnnz;
    P.ia = std::move(tia);
    P.ja.alloc((size_t)tnnz);
    P.val.alloc((size_t)tnnz);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < row; ++i) {
        double Min_neg = 0, Max_pos = 0, Sum_neg = 0, Sum_pos = 0, TSum_neg = 0, TSum_pos = 0;
        for (int j = pia[i]; j < pia[i + 1]; ++j) {
            if (pval[j] > 0) {
                Sum_pos += pval[j];
                Max_pos = std::max(Max_pos, pval[j]);
            } else {
                Sum_neg += pval[j];
                Min_neg = std::min(Min_neg, pval[j]);
            }
        }
        Max_pos *= eps_tr;
        Min_neg *= eps_tr;
        for (int j = pia[i]; j < pia[i + 1]; ++j) {
            if (pval[j] >= Max_pos) TSum_pos += pval[j];
            else if (pval[j] <= Min_neg) TSum_neg += pval[j];
        }
        const double Fac_pos = (TSum_pos > SMALLREAL) ? Sum_pos / TSum_pos : 1.0;
        const double Fac_neg = (TSum_neg < -SMALLREAL) ? Sum_neg / TSum_neg : 1.0;
        int o = P.ia[i];
        for (int j = pia[i]; j < pia[i + 1]; ++j) {
            if (pval[j] >= Max_pos) {
                P.ja[o] = cindex[pja[j]];
                P.val[o++] = pval[j] * Fac_pos;
            } else if (pval[j] <= Min_neg) {
                P.ja[o] = cindex[pja[j]];
                P.val[o++] = pval[j] * Fac_neg;
            }
        }
    }
}

// Standard interpolation: pattern (form_P_pattern_std, PreAMGCoarsenRS.c:2006: strong C neighbours and the
// strong C neighbours of strong F neighbours, in discovery order) and weights (interp_STD,
// PreAMGInterp.c:547-745, RS_C1 ON).  The reference works with row-sized scratch arrays (visited, rindi,
// rindk, Ahat) that carry nothing from one row to the next; here every row uses short local lists and
// linear searches instead, so rows run in parallel and every value is formed by the same operations in
// the same order.
void build_interp_std(const HostCSR& A, const Pattern& S, const int* vec, const AMG_param& param, HostCSR& P)
{
    const int row = A.row;
    const int *ia = A.ia.data(), *ja = A.ja.data();
    const double* av = A.val.data();
    auto pattern_row = [&](int i, std::vector<int>& cols) { std_pattern_row(S, vec, i, cols); };
    Buf<int> pia((size_t)row + 1);
    pia[0] = 0;
#pragma omp parallel
    {
        std::vector<int> cols;
#pragma omp for schedule(static)
        for (int i = 0; i < row; ++i) { pattern_row(i, cols); pia[i + 1] = (int)cols.size(); }
    }
    for (int i = 0; i < row; ++i) pia[i + 1] += pia[i];
    const int   pnnz = pia[row];
    Buf<int>    pja((size_t)std::max(pnnz, 1));
    Buf<double> pval((size_t)std::max(pnnz, 1));

    // Step 0 (:588-613): diagonal (last hit), sums over strong C couplings / all off-diagonals / non-isolated ones
    std::vector<double> csum((size_t)row), psum((size_t)row), nsum((size_t)row), diag((size_t)row);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < row; ++i) {
        double cs = 0.0, ps = 0.0, ns = 0.0, dg = 0.0;
        for (int j = ia[i]; j < ia[i + 1]; ++j) {
            const int k = ja[j];
            bool strongC = false;  // cindex[k] == i: k is a C point in the strength row of i
            if (vec[k] == CGPT)
                for (int q = S.ia[i]; q < S.ia[i + 1]; ++q) if (S.ja[q] == k) { strongC = true; break; }
            if (strongC) cs += av[j];
            if (k == i) dg = av[j];
            else { ns += av[j]; if (vec[k] != ISPT) ps += av[j]; }
        }
        csum[i] = cs; psum[i] = ps; nsum[i] = ns; diag[i] = dg;
    }
    auto entry = [&](int r, int c) -> double {  // A(r, c) as the reference's reverse index finds it: the LAST stored hit
        double v = 0.0;
        for (int m = ia[r]; m < ia[r + 1]; ++m) if (ja[m] == c) v = av[m];
        return v;
    };
#pragma omp parallel
    {
        std::vector<int> cols;
        std::vector<double> ah;
#pragma omp for schedule(dynamic, 512)
        for (int i = 0; i < row; ++i) {
            pattern_row(i, cols);
            const int o = pia[i];
            for (size_t c = 0; c < cols.size(); ++c) pja[o + (int)c] = cols[c];
            if (vec[i] == CGPT) { pval[o] = 1.0; continue; }
            if (vec[i] != FGPT) continue;
            ah.assign(cols.size(), 0.0);
            auto slot = [&](int h) -> double* {
                for (size_t c = 0; c < cols.size(); ++c) if (cols[c] == h) return &ah[c];
                return nullptr;  // not in the pattern: the reference's scratch entry is never read
            };
            double alN = psum[i], alP = csum[i], ahat_i = diag[i];
            for (int j = S.ia[i]; j < S.ia[i + 1]; ++j) {
                const int    k = S.ja[j];
                const double aik = entry(i, k);
                if (vec[k] == CGPT) { if (double* q = slot(k)) *q += aik; }
                else if (vec[k] == FGPT) {
                    const double akk = diag[k], factor = aik / akk;
                    double aki = 0.0;
                    for (int m = ia[k]; m < ia[k + 1]; ++m)
                        if (ja[m] == i) { aki = av[m]; ahat_i -= factor * aki; }
                    for (int m = S.ia[k]; m < S.ia[k + 1]; ++m) {
                        const int l = S.ja[m];
                        if (vec[l] == CGPT) { const double akl = entry(k, l); if (double* q = slot(l)) *q -= factor * akl; }
                    }
                    alN -= factor * (nsum[k] - aki + akk);
                    alP -= factor * csum[k];
                }
            }
            if (!cols.empty()) {
                const double alpha = alN / alP;
                for (size_t c = 0; c < cols.size(); ++c) pval[o + (int)c] = -alpha * ah[c] / ahat_i;
            }
        }
    }
    finish_interp(row, vec, pia, pja, pval, param.truncation_threshold, P);
}

// Reduction-based AMG (INTERP_RDC).  rdc_theta: the diagonal-dominance measure of the F rows, min over rows
// (form_P_pattern_rdc, PreAMGCoarsenRS.c:1796-1850: the diagonal itself counts in the F-row sum), and what the
// reference derives from it (:167-202): theta raised to 0.5 + 1e-5 if needed, the weight of the F-point Jacobi smoother.
void rdc_theta(const HostCSR& A, const int* vec, AMG_param& param)
{
    const int row = A.row;
    double th = 1.0;
#pragma omp parallel for schedule(static) reduction(min : th)
    for (int i = 0; i < row; ++i) {
        if (vec[i] == CGPT) continue;
        double sum = 0.0;
        int    diagptr = -1;
        for (int k = A.ia[i]; k < A.ia[i + 1]; ++k) {
            const int j = A.ja[k];
            if (vec[j] != CGPT) sum += dabs(A.val[k]);
            if (j == i) diagptr = k;
        }
        if (diagptr > -1) th = std::min(th, dabs(A.val[diagptr]) / sum);
    }
    param.theta = th;
    if (param.theta <= 0.5) {
        if (param.print_level > PRINT_MIN)
            std::printf("### WARNING: theta = %e <= 0.5, use %e instead \n", param.theta, 0.5 + 1e-5);
        param.theta = 0.5 + 1e-5;
    }
    if (param.theta >= 0.0) {
        const double t = param.theta, eps = (2 - 2 * t) / (2 * t - 1), sigma = 2 / (2 + eps);
        if (param.smoother == SMOOTHER_JACOBIF) param.relaxation = sigma / (2 - 1 / t);
    }
}

// P = [-(alpha D_FF)^{-1} A_FC; I], alpha = 2 - 1/theta (interp_RDC, PreAMGInterp.c:240-280); no truncation
void build_interp_rdc(const HostCSR& A, const int* vec, const AMG_param& param, HostCSR& P)
{
    const int    row = A.row;
    const double alpha = 2.0 - 1.0 / param.theta;
    std::vector<int> cindex((size_t)row, 0);
    int ncoarse = 0;
    for (int i = 0; i < row; ++i)
        if (vec[i] == CGPT) cindex[i] = ncoarse++;
    P.row = row; P.col = ncoarse;
    P.ia.alloc((size_t)row + 1);
    P.ia[0] = 0;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < row; ++i) {
        int c = 1;
        if (vec[i] != CGPT) {
            c = 0;
            for (int k = A.ia[i]; k < A.ia[i + 1]; ++k) c += (vec[A.ja[k]] == CGPT);
        }
        P.ia[i + 1] = c;
    }
    for (int i = 0; i < row; ++i) P.ia[i + 1] += P.ia[i];
    P.nnz = P.ia[row];
    P.ja.alloc((size_t)std::max(P.nnz, 1));
    P.val.alloc((size_t)std::max(P.nnz, 1));
    // (a row without a diagonal entry inherits the previous row's position in the reference: rows then depend on
    // each other, which a matrix that reaches this point does not have -- theta would be undefined there)
#pragma omp parallel for schedule(static)
    for (int i = 0; i < row; ++i) {
        int o = P.ia[i];
        if (vec[i] == CGPT) { P.ja[o] = cindex[i]; P.val[o] = 1.0; continue; }
        int idiag = A.ia[i];
        for (int k = A.ia[i]; k < A.ia[i + 1]; ++k)
            if (A.ja[k] == i) { idiag = k; break; }
        const double Dii = alpha * A.val[idiag];
        for (int k = A.ia[i]; k < A.ia[i + 1]; ++k)
            if (vec[A.ja[k]] == CGPT) { P.ja[o] = cindex[A.ja[k]]; P.val[o++] = -A.val[k] / Dii; }
    }
}

// stable counting transpose with values (BlaSparseCSR.c:952-1018)
void transpose_csr(const HostCSR& A, HostCSR& AT)
{
    const int n = A.row, m = A.col, nnz = A.nnz;
    AT.row = m; AT.col = n; AT.nnz = nnz;
    AT.ia.alloc((size_t)m + 1);
    AT.ja.alloc((size_t)std::max(nnz, 1));
    AT.val.alloc((size_t)std::max(nnz, 1));
    transpose_stable<true>(n, m, nnz, A.ia.data(), A.ja.data(), A.val.data(), AT.ia.data(), AT.ja.data(), AT.val.data());
}

// Galerkin product RAP (BlaSpmvCSR.c:1114-1142 symbolic, :1204-1244 numeric).  Every
// coarse row starts with its diagonal slot (0.0, then accumulated); the other columns
// are appended in discovery order of the loop R-row -> A-row -> P-row; products are
// formed as (r*a)*p and accumulated with += in that order.  Rows are independent, so
// they run in parallel with per-thread marker arrays (the reference's own OpenMP
// branch does the same); the result is bit-identical to the serial sweep.
// the marker form: per-thread full-length stamp arrays (levels whose markers are small: filling them costs nothing there)
static void galerkin_rap_markers(const HostCSR& R, const HostCSR& A, const HostCSR& P, HostCSR& C)
{
    const int nc = R.row, nf = A.row;
    const int *Ri = R.ia.data(), *Rj = R.ja.data(), *Ai = A.ia.data(), *Aj = A.ja.data(),
              *Pi = P.ia.data(), *Pj = P.ja.data();
    const double *Rv = R.val.data(), *Av = A.val.data(), *Pv = P.val.data();
    int nthreads = omp_get_max_threads();
    if (nthreads > 32) nthreads = 32;  // marker memory: (nc + nf) ints per thread
    if (nc < 2000) nthreads = 1;

    Buf<int> cia((size_t)nc + 1);
    cia[0] = 0;
    int*    Cj = nullptr;
    double* Cv = nullptr;
    bool    overflow = false;
    // One parallel region for the symbolic and the numeric pass: the per-thread marker arrays
    // ((nc + nf) ints each) are touched once; the numeric pass stamps with -2 - ic, values the
    // symbolic pass (stamps ic >= 0, initial -1) never wrote.
#pragma omp parallel num_threads(nthreads)
    {
        std::vector<int> pstamp(nc, -1), ppos(nc, 0), astamp(nf, -1);
#pragma omp for schedule(dynamic, 256)
        for (int ic = 0; ic < nc; ++ic) {
            int cnt = 1;
            pstamp[ic] = ic;
            for (int j1 = Ri[ic]; j1 < Ri[ic + 1]; ++j1) {
                const int i1 = Rj[j1];
                for (int j2 = Ai[i1]; j2 < Ai[i1 + 1]; ++j2) {
                    const int i2 = Aj[j2];
                    if (astamp[i2] != ic) {
                        astamp[i2] = ic;
                        for (int j3 = Pi[i2]; j3 < Pi[i2 + 1]; ++j3) {
                            const int i3 = Pj[j3];
                            if (pstamp[i3] != ic) { pstamp[i3] = ic; ++cnt; }
                        }
                    }
                }
            }
            cia[ic + 1] = cnt;
        }
#pragma omp single
        {
            long long total = 0;
            for (int ic = 0; ic < nc; ++ic) total += cia[ic + 1];
            if (total > 2147483647LL) overflow = true;  // INT is 32-bit in the ABI
            else {
                for (int ic = 0; ic < nc; ++ic) cia[ic + 1] += cia[ic];
                const int cnnz = cia[nc];
                C.row = nc; C.col = nc; C.nnz = cnnz;
                C.ja.alloc((size_t)cnnz);
                C.val.alloc((size_t)cnnz);
                Cj = C.ja.data();
                Cv = C.val.data();
            }
        }  // implicit barrier
        if (!overflow) {
#pragma omp for schedule(dynamic, 256)
            for (int ic = 0; ic < nc; ++ic) {
                const int stamp = -2 - ic;
                int pos = cia[ic];
                pstamp[ic] = stamp;
                ppos[ic]   = pos;
                Cj[pos]    = ic;
                Cv[pos]    = 0.0;
                ++pos;
                for (int j1 = Ri[ic]; j1 < Ri[ic + 1]; ++j1) {
                    const double r_entry = Rv[j1];
                    const int    i1 = Rj[j1];
                    for (int j2 = Ai[i1]; j2 < Ai[i1 + 1]; ++j2) {
                        const double ra = r_entry * Av[j2];
                        const int    i2 = Aj[j2];
                        if (astamp[i2] != stamp) {
                            astamp[i2] = stamp;
                            for (int j3 = Pi[i2]; j3 < Pi[i2 + 1]; ++j3) {
                                const double rap = ra * Pv[j3];
                                const int    i3 = Pj[j3];
                                if (pstamp[i3] != stamp) {
                                    pstamp[i3] = stamp;
                                    ppos[i3]   = pos;
                                    Cv[pos]    = rap;
                                    Cj[pos]    = i3;
                                    ++pos;
                                } else {
                                    Cv[ppos[i3]] += rap;
                                }
                            }
                        } else {
                            for (int j3 = Pi[i2]; j3 < Pi[i2 + 1]; ++j3) Cv[ppos[Pj[j3]]] += ra * Pv[j3];
                        }
                    }
                }
            }
        }
    }
    if (overflow) throw std::bad_alloc();
    C.ia = std::move(cia);
}

// (Round 4: the per-thread markers -- "seen this fine row / this coarse column in the current row, and where" -- were three full-length
// int arrays per thread, 134 MB each at level 0 of P7(256) and 4 GB over the team, filled before the first row.  They are small
// open-addressing tables now, sized per row from the lengths of the rows it multiplies and validated by a per-row stamp: nothing
// to fill, everything a row touches stays in the L1 / L2.  Discovery order and accumulation order are untouched.)
namespace {
struct RapTable {
    std::vector<int> key, stamp, val;
    void ensure(size_t cap) { if (key.size() < cap) { key.resize(cap); val.resize(cap); stamp.assign(cap, 0); } }   // (stamps of a grown table start over: the row that grows it has not used it yet)
    // slot of k (inserted if absent: fresh = true); mask + 1 = power of two >= twice the number of keys of the row
    inline int slot(int k, int st, unsigned mask, bool& fresh)
    {
        for (unsigned s = (((unsigned)k * 2654435761u) >> 11) & mask;; s = (s + 1) & mask) {
            if (stamp[s] != st) { stamp[s] = st; key[s] = k; fresh = true; return (int)s; }
            if (key[s] == k) { fresh = false; return (int)s; }
        }
    }
};
inline unsigned pow2_mask(long long need) { unsigned c = 16; while ((long long)c < need) c <<= 1; return c - 1; }
}  // namespace

void galerkin_rap(const HostCSR& R, const HostCSR& A, const HostCSR& P, HostCSR& C)
{
    const int nc = R.row, nf = A.row;
    const int *Ri = R.ia.data(), *Rj = R.ja.data(), *Ai = A.ia.data(), *Aj = A.ja.data(),
              *Pi = P.ia.data(), *Pj = P.ja.data();
    const double *Rv = R.val.data(), *Av = A.val.data(), *Pv = P.val.data();
    static const int table_min = std::getenv("FASP_HIP_RAP_TABLE_MIN") ? std::atoi(std::getenv("FASP_HIP_RAP_TABLE_MIN")) : 12000000;
    if (nf < table_min) { galerkin_rap_markers(R, A, P, C); return; }   // (measured on P7(256): level 0, 16.8 M fine rows, 0.93 -> 0.46 s with the tables; level 1, 8.4 M, 0.45 -> 0.52 s; levels 2+ 0.11 -> 0.22 s)
    int nthreads = omp_get_max_threads();
    if (nc < 2000) nthreads = 1;

    Buf<int> cia((size_t)nc + 1);
    cia[0] = 0;
    int*    Cj = nullptr;
    double* Cv = nullptr;
    bool    overflow = false;
    // One parallel region for the symbolic and the numeric pass.  Stamps: symbolic 2 ic + 1, numeric 2 ic + 2 (never 0, never reused).
    if (nc > 1000000000) throw std::bad_alloc();
#pragma omp parallel num_threads(nthreads)
    {
        RapTable TA, TP;   // fine rows seen in this coarse row; coarse columns of this row -> position
        std::vector<int> seen2;   // the fine rows of the current coarse row, in discovery order (symbolic pass)
        auto mask_a = [&](int ic) {
            long long ub2 = 0;
            for (int j1 = Ri[ic]; j1 < Ri[ic + 1]; ++j1) ub2 += Ai[Rj[j1] + 1] - Ai[Rj[j1]];
            const unsigned ma = pow2_mask(2 * std::min<long long>(ub2, nf) + 2);
            TA.ensure((size_t)ma + 1);
            return ma;
        };
#pragma omp for schedule(dynamic, 256)
        for (int ic = 0; ic < nc; ++ic) {
            const unsigned ma = mask_a(ic);
            const int st = 2 * ic + 1;
            bool fresh;
            seen2.clear();
            long long ub3 = 1;   // entries of P behind the distinct fine rows (+ the diagonal): bounds the distinct coarse columns
            for (int j1 = Ri[ic]; j1 < Ri[ic + 1]; ++j1) {
                const int i1 = Rj[j1];
                for (int j2 = Ai[i1]; j2 < Ai[i1 + 1]; ++j2) {
                    const int i2 = Aj[j2];
                    (void)TA.slot(i2, st, ma, fresh);
                    if (fresh) { seen2.push_back(i2); ub3 += Pi[i2 + 1] - Pi[i2]; }
                }
            }
            const unsigned mp = pow2_mask(2 * std::min<long long>(ub3, nc) + 2);
            TP.ensure((size_t)mp + 1);
            int cnt = 1;
            (void)TP.slot(ic, st, mp, fresh);
            for (int i2 : seen2)
                for (int j3 = Pi[i2]; j3 < Pi[i2 + 1]; ++j3) {
                    bool f3;
                    (void)TP.slot(Pj[j3], st, mp, f3);
                    cnt += f3 ? 1 : 0;
                }
            cia[ic + 1] = cnt;
        }
#pragma omp single
        {
            long long total = 0;
            for (int ic = 0; ic < nc; ++ic) total += cia[ic + 1];
            if (total > 2147483647LL) overflow = true;  // INT is 32-bit in the ABI
            else {
                for (int ic = 0; ic < nc; ++ic) cia[ic + 1] += cia[ic];
                const int cnnz = cia[nc];
                C.row = nc; C.col = nc; C.nnz = cnnz;
                C.ja.alloc((size_t)cnnz);
                C.val.alloc((size_t)cnnz);
                Cj = C.ja.data();
                Cv = C.val.data();
            }
        }  // implicit barrier
        if (!overflow) {
#pragma omp for schedule(dynamic, 256)
            for (int ic = 0; ic < nc; ++ic) {
                const unsigned ma = mask_a(ic), mp = pow2_mask(2ll * (cia[ic + 1] - cia[ic]) + 2);   // (the symbolic pass counted the row's columns)
                TP.ensure((size_t)mp + 1);
                const int st = 2 * ic + 2;
                bool fresh;
                int pos = cia[ic];
                TP.val[(size_t)TP.slot(ic, st, mp, fresh)] = pos;
                Cj[pos]    = ic;
                Cv[pos]    = 0.0;
                ++pos;
                for (int j1 = Ri[ic]; j1 < Ri[ic + 1]; ++j1) {
                    const double r_entry = Rv[j1];
                    const int    i1 = Rj[j1];
                    for (int j2 = Ai[i1]; j2 < Ai[i1 + 1]; ++j2) {
                        const double ra = r_entry * Av[j2];
                        const int    i2 = Aj[j2];
                        (void)TA.slot(i2, st, ma, fresh);
                        if (fresh) {
                            for (int j3 = Pi[i2]; j3 < Pi[i2 + 1]; ++j3) {
                                const double rap = ra * Pv[j3];
                                const int    i3 = Pj[j3];
                                bool f3;
                                const int s3 = TP.slot(i3, st, mp, f3);
                                if (f3) {
                                    TP.val[(size_t)s3] = pos;
                                    Cv[pos]    = rap;
                                    Cj[pos]    = i3;
                                    ++pos;
                                } else {
                                    Cv[TP.val[(size_t)s3]] += rap;
                                }
                            }
                        } else {
                            for (int j3 = Pi[i2]; j3 < Pi[i2 + 1]; ++j3) {
                                bool f3;
                                Cv[TP.val[(size_t)TP.slot(Pj[j3], st, mp, f3)]] += ra * Pv[j3];
                            }
                        }
                    }
                }
            }
        }
    }
    if (overflow) throw std::bad_alloc();
    C.ia = std::move(cia);
}

void copy_csr(const dCSRmat* A, HostCSR& B)
{
    B.row = A->row; B.col = A->col; B.nnz = A->nnz;
    B.ia.alloc((size_t)A->row + 1);
    B.ja.alloc((size_t)A->nnz);
    B.val.alloc((size_t)A->nnz);
    std::memcpy(B.ia.data(), A->IA, ((size_t)A->row + 1) * sizeof(int));
    std::memcpy(B.ja.data(), A->JA, (size_t)A->nnz * sizeof(int));
    std::memcpy(B.val.data(), A->val, (size_t)A->nnz * sizeof(double));
}

// ---------------------------------------------------------------------------
// smoothed aggregation (PreAMGSetupSA.c:63 -> amg_setup_smoothP_smoothR :254)
// ---------------------------------------------------------------------------
void first_diag(const HostCSR& A, std::vector<double>& d)  // fasp_dcsr_getdiag(0, ..): first hit, 0 if absent
{
    const int n = std::min(A.row, A.col);
    d.assign(std::max(A.row, 1), 0.0);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        for (int k = A.ia[i]; k < A.ia[i + 1]; ++k)
            if (A.ja[k] == i) { d[i] = A.val[k]; break; }
}

// VMB aggregation, PreAMGAggregation.inl:368-640.  The strongly-coupled neighbourhood
// (a_ij^2 >= eps^2 |a_ii a_jj|, eps halved per level) is filtered in parallel; the three
// aggregation sweeps are the reference's sequential greedy passes.
int aggregation_vmb(const HostCSR& A, std::vector<int>& vv, const AMG_param& param, int NumLevels, HostCSR& N,
                    int& NumAggregates)
{
    const int row = A.row;
    const int max_aggregation = param.max_aggregation;
    std::vector<double> diag;
    first_diag(A, diag);
    double strongly_coupled = param.strong_coupled;
    if (param.tentative_smooth >= SMALLREAL) strongly_coupled = param.strong_coupled * ::pow(0.5, NumLevels - 1);
    const double sc2 = ::pow(strongly_coupled, 2);

    N.row = row; N.col = A.col;
    N.ia.alloc((size_t)row + 1);
    N.ia[0] = 0;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < row; ++i) {
        int c = 0;
        for (int j = A.ia[i]; j < A.ia[i + 1]; ++j)
            c += (A.ja[j] == i) || (::pow(A.val[j], 2) >= sc2 * dabs(diag[i] * diag[A.ja[j]]));
        N.ia[i + 1] = c;
    }
    for (int i = 0; i < row; ++i) N.ia[i + 1] += N.ia[i];
    N.nnz = N.ia[row];
    N.ja.alloc((size_t)N.nnz);
    N.val.alloc((size_t)N.nnz);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < row; ++i) {
        int o = N.ia[i];
        for (int j = A.ia[i]; j < A.ia[i + 1]; ++j)
            if ((A.ja[j] == i) || (::pow(A.val[j], 2) >= sc2 * dabs(diag[i] * diag[A.ja[j]]))) {
                N.ja[o] = A.ja[j];
                N.val[o++] = A.val[j];
            }
    }
    const int *NIA = N.ia.data(), *NJA = N.ja.data();

    vv.assign(row, -2);
    NumAggregates = 0;
    int num_left = row;
    for (int i = 0; i < row; ++i) {  // Step 1
        if (A.ia[i + 1] - A.ia[i] == 1) {
            vv[i] = UNPT;
            --num_left;
            continue;
        }
        bool subset = true;
        for (int j = NIA[i]; j < NIA[i + 1]; ++j)
            if (vv[NJA[j]] >= UNPT) { subset = false; break; }
        if (!subset) continue;
        int count = 1;
        vv[i] = NumAggregates;
        --num_left;
        for (int j = NIA[i]; j < NIA[i + 1]; ++j)
            if (NJA[j] != i && count < max_aggregation) { vv[NJA[j]] = NumAggregates; --num_left; ++count; }
        ++NumAggregates;
    }
    if (NumAggregates < MIN_CDOF) return -33;  // ERROR_AMG_COARSEING
    std::vector<int> temp_C(vv), num_each_agg(NumAggregates, 0);  // Step 2
    for (int i = row; i--;)
        if (vv[i] >= 0) num_each_agg[vv[i]]++;
    for (int i = 0; i < row; ++i) {
        if (vv[i] >= UNPT) continue;
        for (int j = NIA[i]; j < NIA[i + 1]; ++j) {
            const int tc = temp_C[NJA[j]];
            if (tc > UNPT && num_each_agg[tc] < max_aggregation) {
                vv[i] = tc;
                --num_left;
                num_each_agg[tc]++;
                break;
            }
        }
    }
    while (num_left > 0) {  // Step 3
        for (int i = 0; i < row; ++i) {
            if (vv[i] >= UNPT) continue;
            int count = 1;
            vv[i] = NumAggregates;
            --num_left;
            for (int j = NIA[i]; j < NIA[i + 1]; ++j)
                if (NJA[j] != i && vv[NJA[j]] < UNPT && count < max_aggregation) {
                    vv[NJA[j]] = NumAggregates;
                    --num_left;
                    ++count;
                }
            ++NumAggregates;
        }
    }
    return FASP_SUCCESS;
}

// P = S * tentp with S = I - w D^-1 M (M = A, or A filtered onto the neighbourhood N);
// smooth_agg :115 + form_tentative_p (PreAMGAggregationCSR.inl:40) + fasp_blas_dcsr_mxm
// (BlaSpmvCSR.c:893): tentp has one unit entry per aggregated row, so row i of P collects,
// in storage order of M's row, the aggregates of its neighbours (columns in discovery order,
// values accumulated in that same order).
void smoothed_prolongator(const HostCSR& A, HostCSR& N, const std::vector<int>& vv, int nagg,
                          const AMG_param& param, HostCSR& P)
{
    const int row = A.row;
    const double w = param.tentative_smooth;
    const HostCSR* M = &A;
    if (param.smooth_filter == 1) {
#pragma omp parallel for schedule(static)
        for (int i = 0; i < row; ++i) {
            double sA = 0.0, sN = 0.0;
            for (int j = A.ia[i]; j < A.ia[i + 1]; ++j) if (A.ja[j] != i) sA += A.val[j];
            for (int j = N.ia[i]; j < N.ia[i + 1]; ++j) if (N.ja[j] != i) sN += N.val[j];
            for (int j = N.ia[i]; j < N.ia[i + 1]; ++j) if (N.ja[j] == i) N.val[j] += sA - sN;
        }
        M = &N;
    }
    std::vector<double> diag;
    first_diag(*M, diag);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < row; ++i) if (dabs(diag[i]) < 1e-6) diag[i] = 1.0;

    auto sval = [&](int i, int j) {
        return (M->ja[j] == i) ? 1 - w * M->val[j] / diag[i] : -w * M->val[j] / diag[i];
    };
    P.row = row; P.col = nagg;
    P.ia.alloc((size_t)row + 1);
    P.ia[0] = 0;
#pragma omp parallel
    {
        std::vector<int> stamp(nagg, -1);
#pragma omp for schedule(static)
        for (int i = 0; i < row; ++i) {
            int c = 0;
            for (int j = M->ia[i]; j < M->ia[i + 1]; ++j) {
                const int a = vv[M->ja[j]];
                if (a > UNPT && stamp[a] != i) { stamp[a] = i; ++c; }
            }
            P.ia[i + 1] = c;
        }
    }
    for (int i = 0; i < row; ++i) P.ia[i + 1] += P.ia[i];
    P.nnz = P.ia[row];
    P.ja.alloc((size_t)P.nnz);
    P.val.alloc((size_t)P.nnz);
#pragma omp parallel
    {
        std::vector<int> stamp(nagg, -1), pos(nagg, 0);
#pragma omp for schedule(static)
        for (int i = 0; i < row; ++i) {
            int o = P.ia[i];
            for (int j = M->ia[i]; j < M->ia[i + 1]; ++j) {
                const int a = vv[M->ja[j]];
                if (a <= UNPT) continue;
                const double v = sval(i, j) * 1.0;  // S_ik * tentp_k,a
                if (stamp[a] != i) { stamp[a] = i; pos[a] = o; P.ja[o] = a; P.val[o] = 0; P.val[o] += v; ++o; }
                else P.val[pos[a]] += v;
            }
        }
    }
}

// AuxMessage.c:84-123
void print_complexity(const HostHierarchy& H, int prtlvl)
{
    if (prtlvl < PRINT_SOME) return;
    double gridcom = 0.0, opcom = 0.0;
    std::printf("-----------------------------------------------------------\n");
    std::printf("  Level   Num of rows   Num of nonzeros   Avg. NNZ / row   \n");
    std::printf("-----------------------------------------------------------\n");
    for (size_t l = 0; l < H.L.size(); ++l) {
        const HostCSR& A = H.L[l].A;
        std::printf("%5d %13d %17d %14.2f\n", (int)l, A.row, A.nnz, (double)A.nnz / A.row);
        gridcom += A.row;
        opcom += A.nnz;
    }
    std::printf("-----------------------------------------------------------\n");
    gridcom /= H.L[0].A.row;
    opcom /= H.L[0].A.nnz;
    std::printf("  Grid complexity = %.3f  |", gridcom);
    std::printf("  Operator complexity = %.3f\n", opcom);
    std::printf("-----------------------------------------------------------\n");
}

}  // namespace

// Parameter combinations with a device path.  Anything else is refused loudly.
int check_supported(const ITS_param* it, const AMG_param* amg)
{
    if (amg) {
        if (amg->AMG_type != CLASSIC_AMG && amg->AMG_type != SA_AMG && amg->AMG_type != UA_AMG) {
            std::printf("### ERROR: fasp_hip: unknown AMG_type %d\n", amg->AMG_type);
            return ERROR_INPUT_PAR;
        }
        if (amg->AMG_type == UA_AMG && amg->aggregation_type != VMB && amg->aggregation_type != PAIRWISE) {  // (the SA setup always aggregates by VMB, PreAMGSetupSA.c:330)
            std::printf("### ERROR: fasp_hip: aggregation_type %d has no host setup here (VMB and symmetric pairwise "
                        "matching only)\n", amg->aggregation_type);
            return ERROR_INPUT_PAR;
        }
        if (amg->AMG_type == SA_AMG && amg->smooth_restriction != 1) {
            std::printf("### ERROR: fasp_hip: SA with unsmoothed restriction has no device path\n");
            return ERROR_INPUT_PAR;
        }
        if (amg->AMG_type == CLASSIC_AMG && amg->coarsening_type != COARSE_RS && amg->coarsening_type != COARSE_RSP &&
            amg->coarsening_type != COARSE_AC && amg->coarsening_type != COARSE_MIS) {
            std::printf("### ERROR: fasp_hip: coarsening_type %d not supported (COARSE_RS, COARSE_RSP, COARSE_AC and COARSE_MIS only)\n",
                        amg->coarsening_type);
            return ERROR_AMG_COARSE_TYPE;
        }
        if (amg->AMG_type == CLASSIC_AMG && amg->interpolation_type != INTERP_DIR && amg->interpolation_type != INTERP_STD &&
            amg->interpolation_type != INTERP_EXT && !(amg->interpolation_type == INTERP_RDC && amg->coarsening_type != COARSE_AC)) {
            // (INTERP_RDC with aggressive coarsening: on the level where COARSE_AC ends the reference fills the standard
            // pattern with interp_RDC's entry stream of a different length -- undefined there, refused here)
            std::printf("### ERROR: fasp_hip: interpolation_type %d not supported (INTERP_DIR, INTERP_STD, INTERP_EXT, INTERP_RDC w/o COARSE_AC)\n",
                        amg->interpolation_type);
            return ERROR_AMG_INTERP_TYPE;
        }
        if (amg->ILU_levels > 0 || amg->SWZ_levels > 0) {
            std::printf("### ERROR: fasp_hip: ILU / Schwarz smoothers have no device path\n");
            return ERROR_INPUT_PAR;
        }
        switch (amg->cycle_type) {
            case V_CYCLE: case W_CYCLE: case VW_CYCLE: case WV_CYCLE: break;
            case AMLI_CYCLE:
                if (amg->amli_degree >= 0 && amg->amli_degree <= 30) break;
                return ERROR_INPUT_PAR;
            case NL_AMLI_CYCLE: break;
            default:
                std::printf("### ERROR: fasp_hip: cycle_type %d has no device path\n",
                            amg->cycle_type);
                return ERROR_INPUT_PAR;
        }
        switch (amg->smoother) {
            case SMOOTHER_JACOBI: case SMOOTHER_L1DIAG:  // order independent: bandwidth-bound kernels
            case SMOOTHER_POLY:                           // SpMVs + elementwise steps (ItrSmootherCSRpoly.c:67)
            case SMOOTHER_GS: case SMOOTHER_SGS: case SMOOTHER_SOR: case SMOOTHER_SSOR:
            case SMOOTHER_GSOR: case SMOOTHER_SGSOR:     // sequential sweeps: level-scheduled
                break;
            case SMOOTHER_CG:                             // `nsweeps` steps of CG on the level's system
                break;
            case SMOOTHER_GSF:                            // Gauss-Seidel on the F points: needs the C/F marker
            case SMOOTHER_JACOBIF:                        // Jacobi on the F points: needs the C/F marker
                if (amg->AMG_type == CLASSIC_AMG) break;
                std::printf("### ERROR: fasp_hip: the F-point smoothers need a classical (C/F) hierarchy\n");
                return ERROR_AMG_SMOOTH_TYPE;
            default:
                std::printf("### ERROR: fasp_hip: smoother %d has no device path\n", amg->smoother);
                return ERROR_AMG_SMOOTH_TYPE;
        }
        if (amg->smoother == SMOOTHER_GS && amg->smooth_order != NO_ORDER && amg->smooth_order != CF_ORDER)
            return ERROR_INPUT_PAR;
        if (amg->coarse_solver != SOLVER_DEFAULT) {
            std::printf("### ERROR: fasp_hip: direct coarse solvers are not available\n");
            return ERROR_INPUT_PAR;
        }
        if (amg->max_levels < 1 || amg->max_levels > MAX_AMG_LVL) return ERROR_INPUT_PAR;
    }
    if (it) {
        if (it->itsolver_type != SOLVER_CG && it->itsolver_type != SOLVER_BiCGstab && it->itsolver_type != SOLVER_GMRES &&
            it->itsolver_type != SOLVER_VGMRES && it->itsolver_type != SOLVER_VFGMRES &&
            it->itsolver_type != SOLVER_MinRes && it->itsolver_type != SOLVER_GCG && it->itsolver_type != SOLVER_GCR) {
            std::printf("### ERROR: Unknown iterative solver type %d! [%s]\n", it->itsolver_type,
                        "fasp_solver_dcsr_itsolver");
            return ERROR_SOLVER_TYPE;
        }
        if (it->stop_type < STOP_REL_RES || it->stop_type > STOP_MOD_REL_RES) return ERROR_INPUT_PAR;
        if ((it->itsolver_type == SOLVER_GMRES || it->itsolver_type == SOLVER_VGMRES || it->itsolver_type == SOLVER_VFGMRES ||
             it->itsolver_type == SOLVER_GCR) &&
            (it->restart < 1 || it->restart > 1000)) return ERROR_INPUT_PAR;
    }
    return FASP_SUCCESS;
}

int host_setup_rs(const dCSRmat* A, AMG_param* param, HostHierarchy& H)
{
    HostThreads team;  // bounded, constant team size for every parallel loop below
    const int    prtlvl   = param->print_level;
    const int    min_cdof = std::max(param->coarse_dof, MIN_CDOF);
    const double t0       = wall_seconds();
    int          status   = FASP_SUCCESS;
    int          max_lvls = param->max_levels;

    if (!A || !A->IA || !A->JA || !A->val || A->row <= 0 || A->row != A->col) return ERROR_DATA_STRUCTURE;

    H.L.clear();
    H.L.reserve(MAX_AMG_LVL + 1);
    H.L.emplace_back();
    copy_csr(A, H.L[0].A);  // SolCSR.c:503-504: the callee works on a deep copy

    if (prtlvl > PRINT_NONE) std::printf("\nSetting up Classical AMG ...\n");
    param->tentative_smooth = 1.0;  // PreAMGSetupRS.c:83
    if (param->coarsening_type == COARSE_AC) param->aggressive_level = std::max<int>(param->aggressive_level, 1);  // :88-89

    std::vector<int> vertices(A->row);
    int lvl = 0;
    try {
        while (H.L[lvl].A.row > min_cdof && lvl < max_lvls - 1) {
            HostLevel& Lv = H.L[lvl];
            if (g_on_level_matrix) g_on_level_matrix(lvl, g_on_level_ready_ctx);   // A of this level is final (its P, R, cfmark are not yet)
            Pattern    S;
            static const bool timing = std::getenv("FASP_HIP_SETUP_TIMING") != nullptr;
            double tp = wall_seconds();
            auto lap = [&](const char* what) {
                if (!timing) return;
                const double now = wall_seconds();
                std::printf("  [setup level %d] %-12s %8.3f s\n", lvl, what, now - tp);
                tp = now;
            };
            const bool rsp = param->coarsening_type == COARSE_RSP;
            Buf<unsigned char> strong;
            status = strength_compressed(Lv.A, *param, S, rsp ? &strong : nullptr);
            lap("strength");
            int col = -1;
            const bool agg = param->coarsening_type == COARSE_AC;
            if (param->coarsening_type == COARSE_MIS) {  // (the reference ignores an empty strength pattern here: every vertex is a C point then)
                col = cfsplitting_mis(S, vertices.data());
                status = FASP_SUCCESS;
            } else if (status >= 0)
                col = agg ? cfsplitting_agg(S, vertices.data(), param->aggressive_path)
                          : cfsplitting_cls(S, vertices.data(), rsp ? Lv.A.ia.data() : nullptr);
            if (status >= 0 && rsp) {  // :1020-1036: positive F-F couplings, then the pattern is compressed again
                rem_positive_ff(Lv.A, strong, vertices.data());
                Pattern S2;
                compress_strength(Lv.A, strong, S2);
                if (S2.nnz > 0) S = std::move(S2);
            }
            lap("C/F split");
            if (status < 0 || col <= 0) {  // Check 1, PreAMGSetupRS.c:162-173
                if (prtlvl > PRINT_MIN) {
                    std::printf("### WARNING: Could not find any C-variables!\n");
                    std::printf("### WARNING: Stop coarsening on level=%d!\n", lvl);
                }
                status = FASP_SUCCESS;
                break;
            }
            // the standard pattern (no F-F clean-up, PreAMGCoarsenRS.c:152) is forced for aggressive coarsening (:96)
            // (INTERP_EXT is the same pattern and, in the reference, the same text as interp_STD: PreAMGInterp.c:760 vs :547)
            const bool std_like = param->interpolation_type == INTERP_STD || param->interpolation_type == INTERP_EXT;
            const bool std_pattern = agg || std_like;
            const bool rdc = !agg && param->interpolation_type == INTERP_RDC;
            if (rdc) rdc_theta(Lv.A, vertices.data(), *param);  // :157-202, before the checks below as in the reference
            else if (!std_pattern) col = clean_ff_couplings(S, vertices.data(), Lv.A.row, col);
            lap("FF clean-up");
            if (col < MIN_CDOF) break;  // Check 2, :176-181
            if (Lv.A.row > col * 10.0) {  // Check 3, :184-195
                if (prtlvl > PRINT_MIN) {
                    std::printf("### WARNING: Coarsening might be too aggressive!\n");
                    std::printf("### WARNING: Fine level = %d, coarse level = %d. Discard!\n",
                                Lv.A.row, col);
                }
                break;
            }
            // :198-199: the reference falls back to the classical splitting once coarsening slows down
            // (and after the aggressive levels; aggressive_level is 0 unless COARSE_AC is requested)
            if (col * 1.5 > Lv.A.row) param->coarsening_type = COARSE_RS;   // (P.col there = the splitting's C count)
            if (lvl == param->aggressive_level) param->coarsening_type = COARSE_RS;
            Lv.cfmark.alloc((size_t)Lv.A.row);  // :201-206
            std::memcpy(Lv.cfmark.data(), vertices.data(), (size_t)Lv.A.row * sizeof(int));

            // PreAMGInterp.c:68-71 looks at coarsening_type AFTER the switch above: on the level where aggressive
            // coarsening ends, the user's interpolation fills the standard pattern
            if (rdc) build_interp_rdc(Lv.A, vertices.data(), *param, Lv.P);
            else if (std_like || param->coarsening_type == COARSE_AC)
                build_interp_std(Lv.A, S, vertices.data(), *param, Lv.P);
            else build_interp_dir(Lv.A, S, vertices.data(), *param, Lv.P, std_pattern);  // :209
            lap("interpolation");
            transpose_csr(Lv.P, Lv.R);                                 // :212
            lap("transpose");
            H.L.emplace_back();
            galerkin_rap(H.L[lvl].R, H.L[lvl].A, H.L[lvl].P, H.L[lvl + 1].A);  // :213
            lap("RAP");
            H.L[lvl].has_coarse = true;
            if (g_on_level_ready) g_on_level_ready(lvl, g_on_level_ready_ctx);   // A, P, R, cfmark of this level are final
            ++lvl;
            const HostCSR& Ac = H.L[lvl].A;
            if (Ac.nnz / Ac.row > Ac.col * 0.2) {  // Check 4, :261 (integer division)
                if (prtlvl > PRINT_MIN) {
                    std::printf("### WARNING: Coarse matrix is too dense!\n");
                    std::printf("### WARNING: m = n = %d, nnz = %d!\n", Ac.col, Ac.nnz);
                }
                break;
            }
        }
    } catch (const std::bad_alloc&) {
        std::printf("### ERROR: fasp_hip: host allocation failed during AMG setup\n");
        return ERROR_ALLOC_MEM;
    }
    H.setup_seconds = wall_seconds() - t0;
    if (prtlvl > PRINT_NONE) {
        print_complexity(H, prtlvl);
        std::printf("Classical AMG setup costs %.4f seconds.\n", H.setup_seconds);
    }
    return status;
}

void* buf_malloc(size_t bytes)
{
    static const bool thp = !(std::getenv("FASP_HIP_THP") && std::atoi(std::getenv("FASP_HIP_THP")) == 0);
    constexpr size_t HUGE = (size_t)2 << 20;
    if (thp && bytes >= 4 * HUGE) {
        void* q = nullptr;
        if (posix_memalign(&q, HUGE, bytes) == 0 && q) {
            (void)madvise(q, bytes, MADV_HUGEPAGE);
            return q;
        }
    }
    return std::malloc(bytes);
}
void (*g_on_level_ready)(int level, void* ctx) = nullptr;
void (*g_on_level_matrix)(int level, void* ctx) = nullptr;
void* g_on_level_ready_ctx = nullptr;

int host_setup_sa(const dCSRmat* A, AMG_param* param, HostHierarchy& H)
{
    HostThreads team;  // bounded, constant team size for every parallel loop below
    const int    prtlvl   = param->print_level;
    const short  min_cdof = (short)std::max(param->coarse_dof, 50);  // SHORT in the reference (:260)
    const double t0       = wall_seconds();
    int          status   = FASP_SUCCESS;
    const int    max_levels = param->max_levels;

    if (!A || !A->IA || !A->JA || !A->val || A->row <= 0 || A->row != A->col) return ERROR_DATA_STRUCTURE;
    H.L.clear();
    H.L.reserve(MAX_AMG_LVL + 1);
    H.L.emplace_back();
    copy_csr(A, H.L[0].A);
    if (prtlvl > PRINT_NONE) std::printf("\nSetting up SA AMG ...\n");
    if (param->aggregation_type == PAIRWISE) param->pair_number = std::min<int>(param->pair_number, max_levels);

    int lvl = 0;
    try {
        while (H.L[lvl].A.row > min_cdof && lvl < max_levels - 1) {
            HostLevel& Lv = H.L[lvl];
            HostCSR N;
            std::vector<int> vv;
            int nagg = 0;
            status = aggregation_vmb(Lv.A, vv, *param, lvl + 1, N, nagg);
            if (status < 0) {  // Check 1
                if (prtlvl > PRINT_MIN) std::printf("### WARNING: Forming aggregates on level-%d failed!\n", lvl);
                status = FASP_SUCCESS;
                break;
            }
            smoothed_prolongator(Lv.A, N, vv, nagg, *param, Lv.P);
            if (Lv.P.col < MIN_CDOF) { Lv.P = HostCSR(); break; }  // Check 2
            if (Lv.P.row > Lv.P.col * 20.0) {                      // Check 3 (MAX_CRATE)
                if (prtlvl > PRINT_MIN) {
                    std::printf("### WARNING: Coarsening might be too aggressive!\n");
                    std::printf("### WARNING: Fine level = %d, coarse level = %d. Discard!\n", Lv.P.row, Lv.P.col);
                }
                Lv.P = HostCSR();
                break;
            }
            transpose_csr(Lv.P, Lv.R);
            H.L.emplace_back();
            galerkin_rap(H.L[lvl].R, H.L[lvl].A, H.L[lvl].P, H.L[lvl + 1].A);
            H.L[lvl].has_coarse = true;
            ++lvl;
        }
    } catch (const std::bad_alloc&) {
        std::printf("### ERROR: fasp_hip: host allocation failed during AMG setup\n");
        return ERROR_ALLOC_MEM;
    }
    H.setup_seconds = wall_seconds() - t0;
    if (prtlvl > PRINT_NONE) {
        print_complexity(H, prtlvl);
        std::printf("Smoothed aggregation setup costs %.4f seconds.\n", H.setup_seconds);
    }
    return status;
}

// Unsmoothed aggregation on a scalar matrix (PreAMGSetupUA.c:55, VMB aggregation): tentative
// (boolean) prolongation, R = P^T and the Galerkin product with unit entries -- for which the
// general product equals fasp_blas_dcsr_rap_agg (BlaSpmvCSR.c:1276) bit for bit.
namespace {
constexpr int G0PT = -5;                 // fasp_const.h:231
constexpr int ERROR_AMG_COARSEING = -33; // fasp_const.h:38

// boolean prolongation of an aggregation (form_tentative_p / form_boolean_p with unit values)
void boolean_p(const int* vv, int row, int nagg, HostCSR& P)
{
    P.row = row; P.col = nagg;
    P.ia.alloc((size_t)row + 1);
    int j = 0;
    for (int i = 0; i < row; ++i) { P.ia[i] = j; if (vv[i] > UNPT) ++j; }
    P.ia[row] = j;
    P.nnz = j;
    P.ja.alloc((size_t)std::max(j, 1)); P.val.alloc((size_t)std::max(j, 1));
    j = 0;
    for (int i = 0; i < row; ++i)
        if (vv[i] > UNPT) { P.ja[j] = vv[i]; P.val[j] = 1.0; ++j; }
}

// fasp_dcsr_diagpref (BlaSparseCSR.c:680): the diagonal entry changes places with the row's first entry
int diagpref(HostCSR& A)
{
    for (int i = 0; i < A.row; ++i) {
        const int b = A.ia[i], e = A.ia[i + 1];
        if (b < e && A.ja[b] == i) continue;
        int j = b + 1;
        for (; j < e; ++j)
            if (A.ja[j] == i) { std::swap(A.ja[b], A.ja[j]); std::swap(A.val[b], A.val[j]); break; }
        if (j >= e) { std::printf("### ERROR: Diagonal entry %d is zero!\n", i); return ERROR_MISC; }
    }
    return FASP_SUCCESS;
}

// One pass of pairwise matching (form_pairwise, PreAMGAggregationUA.inl:170) on a matrix whose rows
// start with the diagonal.  Sequential by definition: a vertex can only take an UNmatched neighbour.
void form_pairwise(const HostCSR& A, int pair, double k_tg, int* vertices, int& num_agg)
{
    const int row = A.row;
    const int *AIA = A.ia.data(), *AJA = A.ja.data();
    const double* Aval = A.val.data();
    if (pair == 1) {  // Step 1: strongly diagonally dominant rows stay out (G0)
        for (int i = 0; i < row; ++i) {
            double sum = 0.0;
            for (int j = AIA[i] + 1; j < AIA[i + 1]; ++j) sum += std::fabs(Aval[j]);
            vertices[i] = (Aval[AIA[i]] >= ((k_tg + 1.) / (k_tg - 1.)) * sum) ? G0PT : UNPT;
        }
    } else {
        std::fill(vertices, vertices + row, (int)UNPT);
    }
    std::vector<double> s((size_t)std::max(row, 1), 0.0);  // Step 2: minus the off-diagonal row sums
    for (int i = 0; i < row; ++i) {
        if (vertices[i] == G0PT) continue;
        double si = 0.0;
        for (int j = AIA[i] + 1; j < AIA[i + 1]; ++j) si -= Aval[j];
        s[i] = si;
    }
    num_agg = 0;
    int index = 0;
    for (int i = 0; i < row; ++i) {  // Step 3
        if (vertices[i] != UNPT) continue;
        double min_mu = BIGREAL;
        const int row_start = AIA[i], row_end = AIA[i + 1];
        const double aii = Aval[row_start];
        for (int j = row_start + 1; j < row_end; ++j) {
            const int col = AJA[j];
            if (vertices[col] != UNPT) continue;
            const double aij = Aval[j], ajj = Aval[AIA[col]];
            double temp1 = aii + s[i] + 2 * aij;
            double temp2 = ajj + s[col] + 2 * aij;
            temp2 = 1.0 / temp1 + 1.0 / temp2;
            const double temp3 = std::max(std::fabs(aii - s[i]), SMALLREAL);
            double temp4 = std::max(std::fabs(ajj - s[col]), SMALLREAL);
            temp4 = -aij + 1. / (1.0 / temp3 + 1.0 / temp4);
            if (std::fabs(temp4) < SMALLREAL) temp4 = (temp4 > 0) ? SMALLREAL : -SMALLREAL;
            const double mu = (-aij + 1.0 / temp2) / temp4;
            if (min_mu > mu) { min_mu = mu; index = col; }
        }
        vertices[i] = num_agg;
        if (min_mu <= k_tg) vertices[index] = num_agg;
        num_agg += 1;
    }
}

// aggregation_symmpair (PreAMGAggregationUA.inl:363): pair_number matching passes, each on the
// Galerkin matrix of the previous one, then the aggregate indices composed.  Literal, including the
// exit that returns the count of a pass it does not compose.
int aggregation_symmpair(const HostCSR& A0, AMG_param& param, std::vector<int>& vv, int& nagg_out)
{
    const int pair_number = param.pair_number;
    double quality_bound = param.quality_bound;
    int num_agg = 0, dopass = 0, lvl = 0, bandwidth = 0;
    for (int i = 0; i < A0.row; ++i) bandwidth = std::max(bandwidth, A0.ia[i + 1] - A0.ia[i]);
    if (bandwidth > 5.0) param.quality_bound = quality_bound = 1.0 * bandwidth;
    std::vector<HostCSR> Amid((size_t)std::max(pair_number, 1) + 1);
    std::vector<std::vector<int>> vert((size_t)std::max(pair_number, 1) + 1);
    const HostCSR* ptrA = &A0;
    for (int i = 1; i <= pair_number; ++i) {
        vert[lvl].assign((size_t)std::max(ptrA->row, 1), 0);
        form_pairwise(*ptrA, i, quality_bound, vert[lvl].data(), num_agg);
        if (i == 1 && num_agg < MIN_CDOF) {
            int domin = 0;
            for (int k = 0; k < ptrA->row; ++k) if (vert[lvl][k] == G0PT) ++domin;
            const double isorate = (double)num_agg / domin;
            if (isorate < 0.1) return ERROR_AMG_COARSEING;
        }
        if (i < pair_number) {
            HostCSR P, R;
            boolean_p(vert[lvl].data(), ptrA->row, num_agg, P);
            if (P.col < MIN_CDOF) break;
            transpose_csr(P, R);
            galerkin_rap(R, *ptrA, P, Amid[lvl + 1]);
            ptrA = &Amid[lvl + 1];
        }
        ++lvl; ++dopass;
    }
    vv = vert[0];
    if (dopass > 1) {
        for (int i = 0; i < A0.row; ++i) {
            int aggindex = vv[i];
            if (aggindex < 0) continue;
            for (int j = 1; j < dopass; ++j) aggindex = vert[j][aggindex];
            vv[i] = aggindex;
        }
    }
    nagg_out = num_agg;
    return FASP_SUCCESS;
}
}  // namespace

int host_setup_ua(const dCSRmat* A, AMG_param* param, HostHierarchy& H)
{
    HostThreads team;  // bounded, constant team size for every parallel loop below
    const int    prtlvl   = param->print_level;
    const short  min_cdof = (short)std::max(param->coarse_dof, 50);
    const double t0       = wall_seconds();
    int          status   = FASP_SUCCESS;
    const int    max_levels = param->max_levels;

    if (!A || !A->IA || !A->JA || !A->val || A->row <= 0 || A->row != A->col) return ERROR_DATA_STRUCTURE;
    H.L.clear();
    H.L.reserve(MAX_AMG_LVL + 1);
    H.L.emplace_back();
    copy_csr(A, H.L[0].A);
    if (prtlvl > PRINT_NONE) std::printf("\nSetting up UA AMG ...\n");
    if (param->aggregation_type == PAIRWISE) {  // PreAMGSetupUA.c:177-180: matching needs the diagonal first
        param->pair_number = std::min<int>(param->pair_number, max_levels);
        if (diagpref(H.L[0].A) < 0) return ERROR_MISC;
    }

    int lvl = 0;
    try {
        while (H.L[lvl].A.row > min_cdof && lvl < max_levels - 1) {
            HostLevel& Lv = H.L[lvl];
            HostCSR N;
            std::vector<int> vv;
            int nagg = 0;
            if (param->aggregation_type == VMB) {
                status = aggregation_vmb(Lv.A, vv, *param, lvl + 1, N, nagg);
                if (nagg * 4.0 > Lv.A.row) param->strong_coupled /= 2.0;  // :235-238
                else if (nagg * 1.25 < Lv.A.row) param->strong_coupled *= 2.0;
            } else {
                status = aggregation_symmpair(Lv.A, *param, vv, nagg);    // :262
            }
            if (status < 0) {  // Check 1
                if (prtlvl > PRINT_MIN) std::printf("### WARNING: Stop coarsening on level %d!\n", lvl);
                status = FASP_SUCCESS;
                break;
            }
            {   // form_tentative_p (PreAMGAggregationCSR.inl:40) with the constant near-kernel vector
                HostCSR& P = Lv.P;
                const int row = Lv.A.row;
                P.row = row; P.col = nagg;
                P.ia.alloc((size_t)row + 1);
                int j = 0;
                for (int i = 0; i < row; ++i) { P.ia[i] = j; if (vv[i] > UNPT) ++j; }
                P.ia[row] = j;
                P.nnz = j;
                P.ja.alloc((size_t)std::max(j, 1)); P.val.alloc((size_t)std::max(j, 1));
                j = 0;
                for (int i = 0; i < row; ++i)
                    if (vv[i] > UNPT) { P.ja[j] = vv[i]; P.val[j] = 1.0; ++j; }
            }
            if (Lv.P.col < MIN_CDOF) { Lv.P = HostCSR(); break; }  // Check 2
            if (Lv.P.row > Lv.P.col * 20.0) {                      // Check 3 (MAX_CRATE)
                if (prtlvl > PRINT_MIN) {
                    std::printf("### WARNING: Coarsening might be too aggressive!\n");
                    std::printf("### WARNING: Fine level = %d, coarse level = %d. Discard!\n", Lv.P.row, Lv.P.col);
                }
                Lv.P = HostCSR();
                break;
            }
            transpose_csr(Lv.P, Lv.R);
            H.L.emplace_back();
            galerkin_rap(H.L[lvl].R, H.L[lvl].A, H.L[lvl].P, H.L[lvl + 1].A);
            H.L[lvl].has_coarse = true;
            ++lvl;
        }
    } catch (const std::bad_alloc&) {
        std::printf("### ERROR: fasp_hip: host allocation failed during AMG setup\n");
        return ERROR_ALLOC_MEM;
    }
    H.setup_seconds = wall_seconds() - t0;
    if (prtlvl > PRINT_NONE) {
        print_complexity(H, prtlvl);
        std::printf("Unsmoothed aggregation setup costs %.4f seconds.\n", H.setup_seconds);
    }
    return status;
}

// ===========================================================================
// Block (BSR) unsmoothed aggregation, config 3 of BASELINE.json
// ===========================================================================
namespace {

// fasp_blas_smat_mul (BlaSmallMat.c): c_ij = a_i0 b_0j + a_i1 b_1j + ..., left to right
inline void smat_mul(const double* a, const double* b, double* c, int n)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = a[i * n] * b[j];
            for (int k = 1; k < n; ++k) s = s + a[i * n + k] * b[k * n + j];
            c[i * n + j] = s;
        }
}

void copy_bsr(const dBSRmat* A, HostBSR& B)
{
    const size_t nb2 = (size_t)A->nb * A->nb;
    B.ROW = A->ROW; B.COL = A->COL; B.NNZ = A->NNZ; B.nb = A->nb;
    B.ia.alloc((size_t)A->ROW + 1); B.ja.alloc((size_t)A->NNZ); B.val.alloc((size_t)A->NNZ * nb2);
    std::memcpy(B.ia.data(), A->IA, sizeof(int) * ((size_t)A->ROW + 1));
    std::memcpy(B.ja.data(), A->JA, sizeof(int) * (size_t)A->NNZ);
    std::memcpy(B.val.data(), A->val, sizeof(double) * (size_t)A->NNZ * nb2);
}

// condenseBSR (PreAMGAggregationBSR.inl:30-70): the (0,0) entry of every block, then
// fasp_dcsr_compress_inplace(.., 1e-8) (BlaSparseCSR.c:1166) keeps |v| > 1e-8 and diagonals
void condense_bsr(const HostBSR& A, HostCSR& S)
{
    const size_t nb2 = (size_t)A.nb * A.nb;
    S.row = A.ROW; S.col = A.COL;
    S.ia.alloc((size_t)A.ROW + 1);
    S.ja.alloc((size_t)A.NNZ); S.val.alloc((size_t)A.NNZ);
    int k = 0;
    S.ia[0] = 0;
    for (int i = 0; i < A.ROW; ++i) {
        for (int j = A.ia[i]; j < A.ia[i + 1]; ++j) {
            const double v = A.val[(size_t)j * nb2];
            if (std::fabs(v) > 1e-8 || A.ja[j] == i) { S.ja[k] = A.ja[j]; S.val[k] = v; ++k; }
        }
        S.ia[i + 1] = k;
    }
    S.nnz = k;
}

// fasp_dbsr_trans (BlaSparseBSR.c:246): stable counting transpose, every block transposed
void transpose_bsr(const HostBSR& A, HostBSR& AT)
{
    const int n = A.ROW, m = A.COL, nnz = A.NNZ, nb = A.nb, nb2 = nb * nb;
    AT.ROW = m; AT.COL = n; AT.NNZ = nnz; AT.nb = nb;
    AT.ia.alloc((size_t)m + 2); AT.ia.zero();
    AT.ja.alloc((size_t)nnz); AT.val.alloc((size_t)nnz * nb2);
    for (int j = 0; j < nnz; ++j) { const int c = A.ja[j]; if (c < m - 1) AT.ia[c + 2]++; }
    for (int i = 2; i <= m; ++i) AT.ia[i] += AT.ia[i - 1];
    for (int i = 0; i < n; ++i)
        for (int p = A.ia[i]; p < A.ia[i + 1]; ++p) {
            const int j = A.ja[p] + 1;
            const int k = AT.ia[j];
            AT.ja[k] = i;
            for (int a = 0; a < nb; ++a)
                for (int b = 0; b < nb; ++b)
                    AT.val[(size_t)nb2 * k + a * nb + b] = A.val[(size_t)nb2 * p + b * nb + a];
            AT.ia[j] = k + 1;
        }
}

// fasp_blas_dbsr_rap (BlaSpmvBSR.c:5466, serial branch): the scalar product's two passes
// (diagonal first, then columns in discovery order) with block products (R_ik A_kl) P_lj
// formed by smat_mul and accumulated with +=.  Rows are independent: the symbolic pass
// counts per row in parallel, the numeric pass fills each row with a private marker.
void galerkin_rap_bsr(const HostBSR& R, const HostBSR& A, const HostBSR& P, HostBSR& B)
{
    const int row = R.ROW, col = P.COL, nb = A.nb, nb2 = nb * nb;
    B.ROW = row; B.COL = col; B.nb = nb;
    B.ia.alloc((size_t)row + 1);
    std::vector<int> cnt((size_t)row);
    const int ncmark = std::max(row, col);
#pragma omp parallel
    {
        std::vector<int> Pm((size_t)ncmark, -1), Am((size_t)A.ROW, -1);
#pragma omp for schedule(dynamic, 256)
        for (int i = 0; i < row; ++i) {
            int c = 1;
            Pm[i] = i;  // stamp = row index
            for (int j1 = R.ia[i]; j1 < R.ia[i + 1]; ++j1) {
                const int i1 = R.ja[j1];
                for (int j2 = A.ia[i1]; j2 < A.ia[i1 + 1]; ++j2) {
                    const int i2 = A.ja[j2];
                    if (Am[i2] != i) {
                        Am[i2] = i;
                        for (int j3 = P.ia[i2]; j3 < P.ia[i2 + 1]; ++j3) {
                            const int i3 = P.ja[j3];
                            if (Pm[i3] != i) { Pm[i3] = i; ++c; }
                        }
                    }
                }
            }
            cnt[i] = c;
        }
    }
    B.ia[0] = 0;
    for (int i = 0; i < row; ++i) B.ia[i + 1] = B.ia[i] + cnt[i];
    B.NNZ = B.ia[row];
    B.ja.alloc((size_t)B.NNZ); B.val.alloc((size_t)B.NNZ * nb2);
#pragma omp parallel
    {
        std::vector<int> Pm((size_t)ncmark, -1), Am((size_t)A.ROW, -1);
        std::vector<double> tmp((size_t)2 * nb2);
        double *t1 = tmp.data(), *t2 = tmp.data() + nb2;
#pragma omp for schedule(dynamic, 256)
        for (int i = 0; i < row; ++i) {
            const int begin = B.ia[i];
            int counter = begin;
            Pm[i] = counter;
            B.ja[counter] = i;
            for (int e = 0; e < nb2; ++e) B.val[(size_t)counter * nb2 + e] = 0.0;
            ++counter;
            for (int j1 = R.ia[i]; j1 < R.ia[i + 1]; ++j1) {
                const int i1 = R.ja[j1];
                for (int j2 = A.ia[i1]; j2 < A.ia[i1 + 1]; ++j2) {
                    smat_mul(R.val.data() + (size_t)j1 * nb2, A.val.data() + (size_t)j2 * nb2, t1, nb);
                    const int i2 = A.ja[j2];
                    const bool first = Am[i2] != i;
                    Am[i2] = i;
                    for (int j3 = P.ia[i2]; j3 < P.ia[i2 + 1]; ++j3) {
                        const int i3 = P.ja[j3];
                        smat_mul(t1, P.val.data() + (size_t)j3 * nb2, t2, nb);
                        if (first && Pm[i3] < begin) {
                            Pm[i3] = counter;
                            std::memcpy(B.val.data() + (size_t)counter * nb2, t2, sizeof(double) * nb2);
                            B.ja[counter] = i3;
                            ++counter;
                        } else {
                            double* dst = B.val.data() + (size_t)Pm[i3] * nb2;
                            for (int e = 0; e < nb2; ++e) dst[e] += t2[e];
                        }
                    }
                }
            }
        }
    }
}

}  // namespace

namespace {
// fasp_smat_inv_nc4 (BlaSmallMatInv.c:111): adjugate over determinant.  Cofactor m is the sum of three products of the
// listed entries minus three more, accumulated left to right -- the reference's grouping, so the inverse is bit-identical.
const unsigned char kCof4[16][6][3] = {
    {{5,10,15},{6,11,13},{7,9,14},{5,11,14},{6,9,15},{7,10,13}}, {{1,11,14},{2,9,15},{3,10,13},{1,10,15},{2,11,13},{3,9,14}},
    {{1,6,15},{2,7,13},{3,5,14},{1,7,14},{2,5,15},{3,6,13}},     {{1,7,10},{2,5,11},{3,6,9},{1,6,11},{2,7,9},{3,5,10}},
    {{4,11,14},{6,8,15},{7,10,12},{4,10,15},{6,11,12},{7,8,14}}, {{0,10,15},{2,11,12},{3,8,14},{0,11,14},{2,8,15},{3,10,12}},
    {{0,7,14},{2,4,15},{3,6,12},{0,6,15},{2,7,12},{3,4,14}},     {{0,6,11},{2,7,8},{3,4,10},{0,7,10},{2,4,11},{3,6,8}},
    {{4,9,15},{5,11,12},{7,8,13},{4,11,13},{5,8,15},{7,9,12}},   {{0,11,13},{1,8,15},{3,9,12},{0,9,15},{1,11,12},{3,8,13}},
    {{0,5,15},{1,7,12},{3,4,13},{0,7,13},{1,4,15},{3,5,12}},     {{0,7,9},{1,4,11},{3,5,8},{0,5,11},{1,7,8},{3,4,9}},
    {{4,10,13},{5,8,14},{6,9,12},{4,9,14},{5,10,12},{6,8,13}},   {{0,9,14},{1,10,12},{2,8,13},{0,10,13},{1,8,14},{2,9,12}},
    {{0,6,13},{1,4,14},{2,5,12},{0,5,14},{1,6,12},{2,4,13}},     {{0,5,10},{1,6,8},{2,4,9},{0,6,9},{1,4,10},{2,5,8}}};
void block_inv4(double* a)
{
    double c[16], cof[16];
    std::memcpy(c, a, sizeof(c));
    for (int m = 0; m < 16; ++m) {
        double v = 0.0;
        for (int q = 0; q < 6; ++q) {
            const double pr = c[kCof4[m][q][0]] * c[kCof4[m][q][1]] * c[kCof4[m][q][2]];
            v = q == 0 ? pr : (q < 3 ? v + pr : v - pr);
        }
        cof[m] = v;
    }
    const double det = c[0] * cof[0] + c[1] * cof[4] + c[2] * cof[8] + c[3] * cof[12];
    if (std::fabs(det) < SMALLREAL) { for (int m = 0; m < 16; ++m) a[m] = (m % 5 == 0) ? 1.0 : 0.0; return; }
    const double det_inv = 1.0 / det;
    for (int m = 0; m < 16; ++m) a[m] = cof[m] * det_inv;
}
// fasp_smat_invp_nc (BlaSmallMatInv.c:508; what fasp_smat_inv runs for n >= 5): in-place Gauss-Jordan elimination with
// full pivoting -- pivot = the LAST entry of largest magnitude among the rows and columns not yet used, row swap onto the
// diagonal, scaling, elimination of the column, columns unscrambled at the end.  A pivot below SMALLREAL stops it.
int block_inv_pivot(double* a, int n)
{
    int prow_of[8], pcol_of[8], taken[8];
    for (int j = 0; j < n; ++j) taken[j] = 0;
    for (int step = 0; step < n; ++step) {
        double big = 0.0;
        int pr = 0, pc = 0;
        for (int j = 0; j < n; ++j)
            if (taken[j] != 1)
                for (int k = 0; k < n; ++k)
                    if (taken[k] == 0 && std::fabs(a[j * n + k]) >= big) { big = std::fabs(a[j * n + k]); pr = j; pc = k; }
        ++taken[pc];
        if (pr != pc)
            for (int l = 0; l < n; ++l) std::swap(a[pr * n + l], a[pc * n + l]);
        prow_of[step] = pr; pcol_of[step] = pc;
        double& piv = a[pc * n + pc];
        if (std::fabs(piv) < SMALLREAL) return ERROR_SOLVER_EXIT;
        const double pinv = 1.0 / piv;
        piv = 1.0;
        for (int l = 0; l < n; ++l) a[pc * n + l] *= pinv;
        for (int r = 0; r < n; ++r) {
            if (r == pc) continue;
            const double f = a[r * n + pc];
            a[r * n + pc] = 0.0;
            for (int l = 0; l < n; ++l) a[r * n + l] -= a[pc * n + l] * f;
        }
    }
    for (int step = n - 1; step >= 0; --step)
        if (prow_of[step] != pcol_of[step])
            for (int k = 0; k < n; ++k) std::swap(a[k * n + prow_of[step]], a[k * n + pcol_of[step]]);
    return FASP_SUCCESS;
}
}  // namespace

int bsr_diaginv(const dBSRmat* A, double* out)
{
    if (!A || A->nb < 1 || A->nb > 7) return ERROR_INPUT_PAR;
    const int nb = A->nb, nb2 = nb * nb;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < A->ROW; ++i) {
        double* a = out + (size_t)i * nb2;
        for (int e = 0; e < nb2; ++e) a[e] = 0.0;
        for (int k = A->IA[i]; k < A->IA[i + 1]; ++k)
            if (A->JA[k] == i) std::memcpy(a, A->val + (size_t)k * nb2, sizeof(double) * nb2);
        if (nb == 1) {
            a[0] = 1.0 / a[0];
        } else if (nb == 4) {
            block_inv4(a);
        } else if (nb > 4) {   // fasp_smat_inv's default branch (the nc5 closed form is switched off there: `case -5`)
            (void)block_inv_pivot(a, nb);
        } else if (nb == 2) {  // fasp_smat_inv_nc2, BlaSmallMatInv.c:33
            const double a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
            const double det = a0 * a3 - a1 * a2;
            if (std::fabs(det) < SMALLREAL) { a[0] = 1.0; a[1] = 0.0; a[2] = 0.0; a[3] = 1.0; }
            else {
                const double det_inv = 1.0 / det;
                a[0] = a3 * det_inv; a[1] = -a1 * det_inv; a[2] = -a2 * det_inv; a[3] = a0 * det_inv;
            }
        } else {  // fasp_smat_inv_nc3, BlaSmallMatInv.c:67
            const double a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5], a6 = a[6], a7 = a[7], a8 = a[8];
            const double M0 = a4 * a8 - a5 * a7, M3 = a2 * a7 - a1 * a8, M6 = a1 * a5 - a2 * a4;
            const double M1 = a5 * a6 - a3 * a8, M4 = a0 * a8 - a2 * a6, M7 = a2 * a3 - a0 * a5;
            const double M2 = a3 * a7 - a4 * a6, M5 = a1 * a6 - a0 * a7, M8 = a0 * a4 - a1 * a3;
            const double det = a0 * M0 + a3 * M3 + a6 * M6;
            if (std::fabs(det) < SMALLREAL) {
                a[0] = 1.0; a[1] = 0.0; a[2] = 0.0; a[3] = 0.0; a[4] = 1.0; a[5] = 0.0; a[6] = 0.0; a[7] = 0.0; a[8] = 1.0;
            } else {
                const double det_inv = 1.0 / det;
                a[0] = M0 * det_inv; a[1] = M3 * det_inv; a[2] = M6 * det_inv;
                a[3] = M1 * det_inv; a[4] = M4 * det_inv; a[5] = M7 * det_inv;
                a[6] = M2 * det_inv; a[7] = M5 * det_inv; a[8] = M8 * det_inv;
            }
        }
    }
    return FASP_SUCCESS;
}

int host_setup_ua_bsr(const dBSRmat* A, AMG_param* param, HostHierarchyBSR& H)
{
    HostThreads team;  // bounded, constant team size for every parallel loop below
    const int    prtlvl   = param->print_level;
    const short  min_cdof = (short)std::max(param->coarse_dof, 50);  // SHORT, PreAMGSetupUABSR.c:77
    const double t0       = wall_seconds();
    int          status   = FASP_SUCCESS;
    const int    max_levels = std::min<int>(param->max_levels, MAX_AMG_LVL);

    if (!A || !A->IA || !A->JA || !A->val || A->ROW <= 0 || A->ROW != A->COL || A->storage_manner != 0)
        return ERROR_DATA_STRUCTURE;
    H.L.clear();
    H.L.reserve(MAX_AMG_LVL + 1);
    H.L.emplace_back();
    copy_bsr(A, H.L[0].A);
    if (prtlvl > PRINT_NONE) std::printf("\nSetting up UA AMG (BSR) ...\n");
    if (param->aggregation_type == PAIRWISE) param->pair_number = std::min<int>(param->pair_number, max_levels);  // :163

    int lvl = 0;
    try {
        while (H.L[lvl].A.ROW > min_cdof && lvl < max_levels - 1) {
            HostLevelBSR& Lv = H.L[lvl];
            const int nb = Lv.A.nb;
            const size_t nb2 = (size_t)nb * nb;
            Lv.diaginv.alloc((size_t)Lv.A.ROW * nb2);
            { dBSRmat v = Lv.A.view(); if ((status = bsr_diaginv(&v, Lv.diaginv.data())) < 0) return status; }
            HostCSR S, N;
            condense_bsr(Lv.A, S);
            std::vector<int> vv;
            int nagg = 0;
            if (param->aggregation_type == VMB) {
                status = aggregation_vmb(S, vv, *param, lvl + 1, N, nagg);
                // the reference adapts the coupling threshold to the coarsening rate (:201-205)
                if (nagg * 4 > S.row) param->strong_coupled /= 8.0;
                else if (nagg * 1.25 < S.row) param->strong_coupled *= 1.5;
            } else {  // :218-222: symmetric pairwise matching on the condensed matrix, taken as it is
                status = aggregation_symmpair(S, *param, vv, nagg);
            }
            if (status < 0) {
                if (prtlvl > PRINT_MIN) std::printf("### WARNING: Forming aggregates on level-%d failed!\n", lvl);
                status = FASP_SUCCESS;
                break;
            }
            {  // form_boolean_p_bsr, PreAMGAggregationBSR.inl:141: one identity block per aggregated row
                HostBSR& P = Lv.P;
                P.ROW = S.row; P.COL = nagg; P.nb = nb;
                P.ia.alloc((size_t)P.ROW + 1);
                int j = 0;
                for (int i = 0; i < P.ROW; ++i) { P.ia[i] = j; if (vv[i] > -1) ++j; }
                P.ia[P.ROW] = j;
                P.NNZ = j;
                P.ja.alloc((size_t)j); P.val.alloc((size_t)j * nb2); P.val.zero();
                j = 0;
                for (int i = 0; i < P.ROW; ++i)
                    if (vv[i] > -1) {
                        P.ja[j] = vv[i];
                        for (int d = 0; d < nb; ++d) P.val[(size_t)j * nb2 + d * nb + d] = 1.0;
                        ++j;
                    }
            }
            transpose_bsr(Lv.P, Lv.R);
            H.L.emplace_back();
            galerkin_rap_bsr(H.L[lvl].R, H.L[lvl].A, H.L[lvl].P, H.L[lvl + 1].A);
            H.L[lvl].has_coarse = true;
            ++lvl;
        }
    } catch (const std::bad_alloc&) {
        std::printf("### ERROR: fasp_hip: host allocation failed during AMG setup\n");
        return ERROR_ALLOC_MEM;
    }
    H.setup_seconds = wall_seconds() - t0;
    if (prtlvl > PRINT_NONE)
        std::printf("Unsmoothed aggregation (BSR) setup costs %.4f seconds, %d levels.\n", H.setup_seconds, lvl + 1);
    return status;
}

// Parameter combinations of the block path with a device implementation
int check_supported_bsr(const ITS_param* it, const AMG_param* amg, int nb)
{
    if (nb < 1 || nb > 7) {
        std::printf("### ERROR: fasp_hip: BSR AMG needs 1 <= nb <= 7 (block kernels), got %d\n", nb);
        return ERROR_INPUT_PAR;
    }
    if (amg) {
        if (amg->AMG_type == SA_AMG) {
            std::printf("### ERROR: fasp_hip: smoothed aggregation on BSR matrices has no device path\n");
            return ERROR_INPUT_PAR;
        }
        if (amg->aggregation_type != VMB && amg->aggregation_type != PAIRWISE) {
            std::printf("### ERROR: fasp_hip: BSR aggregation_type %d not supported (VMB and symmetric pairwise matching only)\n", amg->aggregation_type);
            return ERROR_INPUT_PAR;
        }
        if (amg->ILU_levels > 0 || amg->SWZ_levels > 0) {
            std::printf("### ERROR: fasp_hip: ILU / Schwarz smoothers have no device path\n");
            return ERROR_INPUT_PAR;
        }
        switch (amg->smoother) {  // the five smoothers of fasp_solver_mgcycle_bsr (PreMGCycle.c:327-365)
            case SMOOTHER_JACOBI:                                                      // bandwidth-bound kernel
            case SMOOTHER_GS: case SMOOTHER_SGS: case SMOOTHER_SOR: case SMOOTHER_SSOR:  // level-scheduled sweeps
                break;
            default:
                std::printf("### ERROR: fasp_hip: BSR smoother %d has no device path\n", amg->smoother);
                return ERROR_AMG_SMOOTH_TYPE;
        }
        if (amg->coarse_scaling == 1 || amg->coarse_solver != SOLVER_DEFAULT) return ERROR_INPUT_PAR;
        switch (amg->cycle_type) {
            case V_CYCLE: case W_CYCLE: case VW_CYCLE: case WV_CYCLE: break;  // nu_l[l] < cycle_type rule
            default: return ERROR_INPUT_PAR;
        }
        if (amg->max_levels < 1 || amg->max_levels > MAX_AMG_LVL) return ERROR_INPUT_PAR;
    }
    if (it) {
        if (it->itsolver_type != SOLVER_CG && it->itsolver_type != SOLVER_BiCGstab && it->itsolver_type != SOLVER_GMRES &&
            it->itsolver_type != SOLVER_VGMRES && it->itsolver_type != SOLVER_VFGMRES) {
            std::printf("### ERROR: Unknown iterative solver type %d! [%s]\n", it->itsolver_type,
                        "fasp_solver_dbsr_itsolver");
            return ERROR_SOLVER_TYPE;
        }
        if (it->stop_type < STOP_REL_RES || it->stop_type > STOP_MOD_REL_RES) return ERROR_INPUT_PAR;
        if ((it->itsolver_type == SOLVER_GMRES || it->itsolver_type == SOLVER_VGMRES || it->itsolver_type == SOLVER_VFGMRES) &&
            (it->restart < 1 || it->restart > 1000)) return ERROR_INPUT_PAR;
    }
    return FASP_SUCCESS;
}

}  // namespace fasp

// ---------------------------------------------------------------------------
// public host-only entry points
// ---------------------------------------------------------------------------
extern "C" {

// AuxParam.c:431-489
void fasp_param_amg_init(AMG_param* p)
{
    std::memset(p, 0, sizeof(*p));
    p->AMG_type = CLASSIC_AMG;
    p->print_level = PRINT_NONE;
    p->maxit = 1;
    p->tol = 1e-6;
    p->max_levels = 20;
    p->coarse_dof = 500;
    p->cycle_type = V_CYCLE;
    p->smoother = SMOOTHER_GS;
    p->smooth_order = CF_ORDER;
    p->presmooth_iter = 1;
    p->postsmooth_iter = 1;
    p->coarse_solver = SOLVER_DEFAULT;
    p->relaxation = 1.0;
    p->polynomial_degree = 3;
    p->coarse_scaling = 0;
    p->amli_degree = 2;
    p->amli_coef = nullptr;
    p->nl_amli_krylov_type = 7;  // SOLVER_GCG
    p->coarsening_type = COARSE_RS;
    p->interpolation_type = INTERP_DIR;
    p->max_row_sum = 0.9;
    p->strong_threshold = 0.3;
    p->truncation_threshold = 0.2;
    p->aggressive_level = 0;
    p->aggressive_path = 1;
    p->aggregation_type = PAIRWISE;
    p->quality_bound = 10.0;
    p->pair_number = 2;
    p->strong_coupled = 0.08;
    p->max_aggregation = 20;
    p->tentative_smooth = 0.67;
    p->smooth_filter = 1;
    p->smooth_restriction = 1;
    p->aggregation_norm_type = -1;
    p->ILU_type = FASP_ILUk;
    p->ILU_levels = 0;
    p->ILU_lfil = 0;
    p->ILU_droptol = 0.001;
    p->ILU_relax = 0;
    p->SWZ_levels = 0;
    p->SWZ_mmsize = 200;
    p->SWZ_maxlvl = 3;
    p->SWZ_type = 1;
    p->SWZ_blksolver = SOLVER_DEFAULT;
    p->theta = -1.0;
}

// AuxParam.c:572-583
void fasp_param_solver_init(ITS_param* p)
{
    std::memset(p, 0, sizeof(*p));
    p->print_level = PRINT_NONE;
    p->itsolver_type = SOLVER_CG;
    p->decoup_type = 1;
    p->precond_type = PREC_AMG;
    p->stop_type = STOP_REL_RES;
    p->maxit = 500;
    p->restart = 25;
    p->tol = 1e-6;
    p->abstol = 1e-18;
}

// Synthetic input P7: test/src/FdmPoisson.c:439 (7-point band system on the unit cube,
// couplings across the boundary zeroed) + :731 (band -> CSR: diagonal first, then the
// offsets -1,+1,-nx,+nx,-nx*ny,+nx*ny; exact zeros deleted, :930).
int fasp_hip_poisson7pt(int nx, int ny, int nz, dCSRmat* A, dvector* b, dvector* u)
{
    const double    PI = 3.1415926535897932;  // test/include/poisson_fdm.h:16
    const long long ngrid = (long long)nx * ny * nz;
    if (nx <= 0 || ny <= 0 || nz <= 0 || 7 * ngrid > 2147483647LL) return ERROR_INPUT_PAR;
    const int    n = (int)ngrid, nplane = nx * ny;
    const double hx = 1.0 / (double)(nx + 1), hy = 1.0 / (double)(ny + 1), hz = 1.0 / (double)(nz + 1);
    const double hx2 = hx * hx, hy2 = hy * hy, hz2 = hz * hz;
    const double fx = 1.0 / hx2, fy = 1.0 / hy2, fz = 1.0 / hz2;
    const double dd = 2.0 * (fx + fy + fz);
    const double constant = 3.0 * PI * PI;

    auto rownnz = [&](int i, int j, int k) {
        return 1 + (i > 0) + (i < nx - 1) + (j > 0) + (j < ny - 1) + (k > 0) + (k < nz - 1);
    };
    int* ia = (int*)std::malloc(((size_t)n + 1) * sizeof(int));
    if (!ia) return ERROR_ALLOC_MEM;
    // row offsets: closed form per (j,k) line keeps this O(n) and parallel
    long long nnz = 0;
    {
        int row = 0;
        for (int k = 0; k < nz; ++k)
            for (int j = 0; j < ny; ++j)
                for (int i = 0; i < nx; ++i) { ia[row++] = (int)nnz; nnz += rownnz(i, j, k); }
        ia[n] = (int)nnz;
    }
    int*    ja = (int*)std::malloc((size_t)nnz * sizeof(int));
    double* a  = (double*)std::malloc((size_t)nnz * sizeof(double));
    double* f  = (double*)std::malloc((size_t)n * sizeof(double));
    double* ue = (double*)std::malloc((size_t)n * sizeof(double));
    if (!ja || !a || !f || !ue) { std::free(ia); std::free(ja); std::free(a); std::free(f); std::free(ue); return ERROR_ALLOC_MEM; }
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nz; ++k) {
        const double z = hz * (k + 1), ss = std::sin(PI * z);
        for (int j = 0; j < ny; ++j) {
            const double y = hy * (j + 1), s = ss * std::sin(PI * y);
            for (int i = 0; i < nx; ++i) {
                const int row = (k * ny + j) * nx + i;
                int       c = ia[row];
                ja[c] = row; a[c++] = dd;
                if (i > 0)      { ja[c] = row - 1;      a[c++] = -fx; }
                if (i < nx - 1) { ja[c] = row + 1;      a[c++] = -fx; }
                if (j > 0)      { ja[c] = row - nx;     a[c++] = -fy; }
                if (j < ny - 1) { ja[c] = row + nx;     a[c++] = -fy; }
                if (k > 0)      { ja[c] = row - nplane; a[c++] = -fz; }
                if (k < nz - 1) { ja[c] = row + nplane; a[c++] = -fz; }
                const double x = hx * (i + 1);
                const double tmp = s * std::sin(PI * x);
                ue[row] = tmp * 1.0;
                f[row]  = tmp * 0 + constant * ue[row];
            }
        }
    }
    A->row = A->col = n; A->nnz = (int)nnz; A->IA = ia; A->JA = ja; A->val = a;
    b->row = n; b->val = f;
    u->row = n; u->val = ue;
    return FASP_SUCCESS;
}


/* Synthetic input of BASELINE config 5 (the reference ships no 3-D FE generator, SURVEY.md
 * section 8d): Q1 trilinear finite elements for -div(K grad u) = 1 on the unit cube, K =
 * diag(kx, ky, kz), homogeneous Dirichlet boundary, n^3 interior nodes, lexicographic, x
 * fastest.  Tensor-product stencil: a(dx,dy,dz) = kx s(dx) m(dy) m(dz) + ky m(dx) s(dy) m(dz)
 * + kz m(dx) m(dy) s(dz) with the 1-D stiffness s = (2,-1)/h and mass m = (4,1) h/6.
 * Row entries in increasing column order; rhs b_i = h^3 (load vector of f = 1). */
int fasp_hip_aniso27pt(int n, double kx, double ky, double kz, dCSRmat* A, dvector* b)
{
    const long long N = (long long)n * n * n;
    if (n <= 0 || 27 * N > 2147483647LL) return ERROR_INPUT_PAR;
    const double h = 1.0 / (double)(n + 1);
    const double s[3] = {-1.0 / h, 2.0 / h, -1.0 / h};
    const double m[3] = {h / 6.0, 4.0 * h / 6.0, h / 6.0};
    int* ia = (int*)std::malloc(((size_t)N + 1) * sizeof(int));
    long long nnz = 0;
    for (int k = 0; k < n; ++k)
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) {
                const int cx = 3 - (i == 0) - (i == n - 1), cy = 3 - (j == 0) - (j == n - 1),
                          cz = 3 - (k == 0) - (k == n - 1);
                ia[(size_t)(k * n + j) * n + i] = (int)nnz;
                nnz += (long long)cx * cy * cz;
            }
    ia[N] = (int)nnz;
    int* ja = (int*)std::malloc((size_t)nnz * sizeof(int));
    double* a = (double*)std::malloc((size_t)nnz * sizeof(double));
    double* f = (double*)std::malloc((size_t)N * sizeof(double));
    if (!ia || !ja || !a || !f) { std::free(ia); std::free(ja); std::free(a); std::free(f); return ERROR_ALLOC_MEM; }
    for (int k = 0; k < n; ++k)
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) {
                const int row = (k * n + j) * n + i;
                int c = ia[row];
                for (int dz = -1; dz <= 1; ++dz) {
                    if (k + dz < 0 || k + dz >= n) continue;
                    for (int dy = -1; dy <= 1; ++dy) {
                        if (j + dy < 0 || j + dy >= n) continue;
                        for (int dx = -1; dx <= 1; ++dx) {
                            if (i + dx < 0 || i + dx >= n) continue;
                            ja[c] = row + (dz * n + dy) * n + dx;
                            a[c] = kx * s[dx + 1] * m[dy + 1] * m[dz + 1] + ky * m[dx + 1] * s[dy + 1] * m[dz + 1] +
                                   kz * m[dx + 1] * m[dy + 1] * s[dz + 1];
                            ++c;
                        }
                    }
                }
                f[row] = h * h * h;
            }
    A->row = A->col = (int)N; A->nnz = (int)nnz; A->IA = ia; A->JA = ja; A->val = a;
    b->row = (int)N; b->val = f;
    return FASP_SUCCESS;
}

void fasp_hip_free_system(dCSRmat* A, dvector* b, dvector* u)
{
    if (A) { std::free(A->IA); std::free(A->JA); std::free(A->val); A->IA = A->JA = nullptr; A->val = nullptr; }
    if (b) { std::free(b->val); b->val = nullptr; }
    if (u) { std::free(u->val); u->val = nullptr; }
}

const char* fasp_hip_version(void) { return "fasp_hip 0.1 (gfx950, HIP + RCCL)"; }

}  // extern "C"
