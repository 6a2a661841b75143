// reorder.cpp -- host side of the brick renumbering of the mid levels (round 5; hierarchy.hip.h, upload_level): breadth-first balls of
// 64 rows over a level's matrix graph, and the symmetric permutation of a level's operators.
//
// Why: the plain-CSR kernels take 64 consecutive rows per wave (k_csr_wstream2, k_csr_xtile) or a sub-wavefront per row (k_csr_rows);
// what a CU has to pull for their x operands is set by how many DISTINCT cache lines those rows touch.  A Ruge-Stuben coarse level
// inherits the lexicographic order of its C points: 64 consecutive rows are a LINE of the coarse grid and share 1.5 entries per distinct
// column; numbered in compact clusters (balls of the matrix graph) they share 4-7 (profiles/r03_cluster_order_xtile.txt: 10-17 % per
// operator on level 2, 12 % on level 3).  The renumbering is internal to the device copy of the hierarchy: rows keep their storage
// order, so every row sum is the same sum -- a cycle is BIT-IDENTICAL with and without it for smoothers that do not depend on the
// order of the rows (Jacobi, L1); the sequential smoothers sweep in index order and keep the natural numbering.
#include <algorithm>
#include <cstdio>
#include <vector>

#include <omp.h>

#include "fasp_internal.h"

namespace fasp {

// B = A with rows and columns renumbered: row k of B is row rperm[k] of A (rperm == nullptr: rows keep their numbers), column j of A
// becomes cinv[j] (cinv == nullptr: columns keep theirs).  The entries of a row keep their storage order.
void permute_csr(const HostCSR& A, const int* rperm, const int* cinv, HostCSR& B)
{
    const int n = A.row;
    B.row = A.row; B.col = A.col; B.nnz = A.nnz; B.row_aligned = A.row_aligned;
    B.ia.alloc((size_t)n + 1); B.ja.alloc((size_t)std::max(A.nnz, 1)); B.val.alloc((size_t)std::max(A.nnz, 1));
    B.ia[0] = 0;
    for (int k = 0; k < n; ++k) { const int i = rperm ? rperm[k] : k; B.ia[(size_t)k + 1] = B.ia[k] + (A.ia[i + 1] - A.ia[i]); }
    HostThreads team;
#pragma omp parallel for schedule(static)
    for (int k = 0; k < n; ++k) {
        const int i = rperm ? rperm[k] : k;
        int o = B.ia[k];
        for (int e = A.ia[i]; e < A.ia[i + 1]; ++e, ++o) { B.ja[o] = cinv ? cinv[A.ja[e]] : A.ja[e]; B.val[o] = A.val[e]; }
    }
}

// order[k] = old index of the row that gets the new index k.  Rows are clustered inside chunks of `chunk` consecutive
// old indices (independent -> parallel, and the new order stays a coarse copy of the old one: a row moves by less than
// one chunk, which keeps the transfer operators' locality): breadth-first balls of 64 rows grown from the lowest
// unassigned index, over the symmetric closure of the pattern restricted to the chunk.
void cluster_order(const HostCSR& A, int chunk, std::vector<int>& order)
{
    const int n = A.row;
    order.resize((size_t)n);
    if (chunk < 64) chunk = 64;
    const int nchunk = (n + chunk - 1) / chunk;
    HostThreads team;
#pragma omp parallel
    {
        std::vector<unsigned char> state;   // 0 free, 1 queued, 2 placed
        std::vector<int> queue;
#pragma omp for schedule(dynamic, 1)
        for (int c = 0; c < nchunk; ++c) {
            const int c0 = c * chunk, c1 = std::min(n, c0 + chunk);
            state.assign((size_t)(c1 - c0), 0);
            int pos = c0, seed = c0;
            while (pos < c1) {
                while (state[(size_t)(seed - c0)] != 0) ++seed;
                queue.clear();
                queue.push_back(seed);
                state[(size_t)(seed - c0)] = 1;
                size_t head = 0;
                int count = 0;
                while (head < queue.size() && count < 64) {
                    const int v = queue[head++];
                    order[(size_t)pos++] = v;
                    state[(size_t)(v - c0)] = 2;
                    ++count;
                    for (int k = A.ia[v]; k < A.ia[v + 1]; ++k) {
                        const int j = A.ja[k];
                        if (j >= c0 && j < c1 && state[(size_t)(j - c0)] == 0) { state[(size_t)(j - c0)] = 1; queue.push_back(j); }
                    }
                }
                for (size_t q = head; q < queue.size(); ++q) state[(size_t)(queue[q] - c0)] = 0;   // reached, not taken: free again
            }
        }
    }
}

}  // namespace fasp


extern "C" {
// test entries (host only; include/fasp_hip_dev.h): the brick order of a square matrix (order[k] = old index at new position k), and an
// operator with rows / columns renumbered (rperm / cinv may be NULL; B's arrays are allocated with fasp_mem_calloc semantics: malloc)
int fasp_hip_cluster_order(const dCSRmat* A, int chunk, int* order)
{
    if (!A || !order || A->row != A->col) return ERROR_INPUT_PAR;
    fasp::HostCSR M;
    M.row = A->row; M.col = A->col; M.nnz = A->nnz;
    M.ia.view(A->IA, (size_t)A->row + 1); M.ja.view(A->JA, (size_t)std::max(A->nnz, 1)); M.val.view(A->val, (size_t)std::max(A->nnz, 1));
    std::vector<int> o;
    fasp::cluster_order(M, chunk, o);
    std::copy(o.begin(), o.end(), order);
    return FASP_SUCCESS;
}
int fasp_hip_permute_csr(const dCSRmat* A, const int* rperm, const int* cinv, int* ia, int* ja, double* val)
{
    if (!A || !ia || !ja || !val) return ERROR_INPUT_PAR;
    fasp::HostCSR M, B;
    M.row = A->row; M.col = A->col; M.nnz = A->nnz;
    M.ia.view(A->IA, (size_t)A->row + 1); M.ja.view(A->JA, (size_t)std::max(A->nnz, 1)); M.val.view(A->val, (size_t)std::max(A->nnz, 1));
    fasp::permute_csr(M, rperm, cinv, B);
    std::copy(B.ia.data(), B.ia.data() + A->row + 1, ia);
    std::copy(B.ja.data(), B.ja.data() + A->nnz, ja);
    std::copy(B.val.data(), B.val.data() + A->nnz, val);
    return FASP_SUCCESS;
}
}
