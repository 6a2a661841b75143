// tools/lab/reorder.cpp -- EXPERIMENT, not part of libfasp_hip.so (moved out of csrc/ in round 4).  Breadth-first balls of `chunk` rows
// over A's row pattern (A^T is not traversed: for the structurally symmetric levels it was tried on the two coincide); nothing in the
// product renumbers a level.  Results: profiles/r03_cluster_order_xtile.txt; DESIGN.md section 8.  To use it again, add it to the library's
// Makefile and declare fasp_hip_cluster_order in include/fasp_hip_dev.h (tools/lab/expt_cluster_order.py and expt_renumber.py call it).
// reorder.cpp -- host side: brick-like renumbering of a coarse level from its matrix graph alone.
//
// Why: the wave-stream kernels (kernels2.hip.h) take 64 consecutive rows per wave; what a CU has to pull from its L2
// for their x operands is set by how many DISTINCT columns those rows touch (profiles/r03_coded_kernel_experiments.txt:
// the kernels of the fine levels are bound by the L2 -> CU rate, not by HBM).  A Ruge-Stuben coarse level inherits the
// lexicographic order of its C points: 64 consecutive rows are a LINE of the coarse grid and share 1.5 entries per
// distinct column.  Numbered in compact clusters of 64 (balls of the matrix graph) they share 4-7, and k_csr_xtile --
// the tile's distinct x entries staged once in LDS -- applies.  The renumbering is internal to the device copy of the
// hierarchy: rows keep their storage order, so every row sum is the same sum; only vectors of the renumbered levels
// are stored in another order (smoothers that sweep in index order keep the natural numbering: hierarchy.hip.h).
#include <algorithm>
#include <cstdio>
#include <vector>

#include <omp.h>

#include "fasp_internal.h"

namespace fasp {

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
// development / test entry: the cluster order of a square matrix (order[k] = old index at new position k)
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
}
