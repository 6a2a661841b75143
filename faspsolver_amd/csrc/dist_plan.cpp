// dist_plan.cpp -- 1-D row partition of the AMG hierarchy over the GPUs of one node.
//
// The reference has no distributed path; this file is new design.
//
//  * Level 0 is split into contiguous row blocks, rank r owns [start_0[r], start_0[r+1]).
//  * A coarse row inherits the owner of its C point.  The classical setup numbers C points
//    in increasing fine index (PreAMGInterp.c:491-493), so every level stays a contiguous
//    row range per rank: start_{l+1}[r] = #C-points of level l below start_l[r].
//  * For a distributed level the rank keeps the rows it owns of A_l, R_l (coarse rows) and
//    P_l (fine rows), with columns renumbered to [0, nloc) for owned entries and
//    [nloc, nloc + nghost) for the halo ("ghost") entries, ghosts sorted by global index and
//    therefore grouped by owner.  One ghost set per level's vector space (the union over
//    the operators that read it: A_l, R_l, P_{l-1}) -> one halo plan per level.
//  * Aggregation hierarchies (SA / UA) carry no C/F marker: every level is cut into equal contiguous row blocks
//    of its own.  The VMB aggregates are numbered in discovery order, i.e. essentially by their first member
//    (PreAMGAggregation.inl:455-486), so a rank's coarse block still sits under its fine block and the halos of
//    P and R stay near the block ends -- but nothing below depends on that: ghost sets and send lists come from
//    the actual columns of the rows a rank owns.
//  * Levels with fewer than `min_rows` rows are REPLICATED: every rank holds the whole
//    level and computes it redundantly; the rhs of the first replicated level is assembled
//    with one all-gather.  Small dense coarse levels would otherwise need all-to-all halos
//    that cost more than the work they spread.
//  * Column order inside every local row is the global row's order, so local row sums
//    are bit-identical to the single-GPU ones.
#include <omp.h>

#include <algorithm>
#include <cstdio>

#include "fasp_internal.h"

namespace fasp {

namespace {

inline int owner_of(const std::vector<int>& start, int g)
{
    // start has P+1 entries, ascending; returns r with start[r] <= g < start[r+1]
    return (int)(std::upper_bound(start.begin(), start.end(), g) - start.begin()) - 1;
}

// columns of rows [r0, r1) of M that fall outside [c0, c1) -> appended to out
void collect_ghost_cols(const HostCSR& M, int r0, int r1, int c0, int c1, std::vector<int>& out)
{
    for (int i = r0; i < r1; ++i)
        for (int k = M.ia[i]; k < M.ia[i + 1]; ++k) {
            const int c = M.ja[k];
            if (c < c0 || c >= c1) out.push_back(c);
        }
}

void sort_unique(std::vector<int>& v)
{
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
}

// rows [r0, r1) of M with columns renumbered: owned [c0,c1) -> c - c0, others -> nloc + rank in ghosts;
// ghosts == nullptr: keep global column indices (operand vector is replicated)
void extract_rows(const HostCSR& M, int r0, int r1, int c0, int c1, const std::vector<int>* ghosts,
                  int ncol_local, HostCSR& out)
{
    const int n = r1 - r0;
    out.row = n;
    out.col = ncol_local;
    out.ia.alloc((size_t)n + 1);
    const int base = M.ia[r0];
    out.nnz = M.ia[r1] - base;
    out.ja.alloc((size_t)out.nnz);
    out.val.alloc((size_t)out.nnz);
    const int nloc = c1 - c0;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        out.ia[i] = M.ia[r0 + i] - base;
        for (int k = M.ia[r0 + i]; k < M.ia[r0 + i + 1]; ++k) {
            const int c = M.ja[k];
            int lc;
            if (!ghosts) lc = c;
            else if (c >= c0 && c < c1) lc = c - c0;
            else lc = nloc + (int)(std::lower_bound(ghosts->begin(), ghosts->end(), c) - ghosts->begin());
            out.ja[k - base]  = lc;
            out.val[k - base] = M.val[k];
        }
    }
    out.ia[n] = out.nnz;
}

}  // namespace

void find_row_window(const HostCSR& M, int nown, int align, int win[2])
{
    win[0] = 0; win[1] = -1;
    const int n = M.row;
    if (n < 4 * align) return;
    const int half = n / 2;
    int last_lo = -1, first_hi = n;   // last ghost-reading row of the first half, first one of the second half
#pragma omp parallel for schedule(static) reduction(max : last_lo) reduction(min : first_hi)
    for (int i = 0; i < n; ++i) {
        bool g = false;
        for (int k = M.ia[i]; k < M.ia[i + 1] && !g; ++k) g = M.ja[k] >= nown;
        if (!g) continue;
        if (i < half) last_lo = std::max(last_lo, i);
        else first_hi = std::min(first_hi, i);
    }
    const int lo = (last_lo + 1 + align - 1) / align * align;
    const int hi = first_hi == n ? n : first_hi / align * align;
    if (hi - lo < n / 2) return;   // ghost readers all over the block: nothing to overlap
    win[0] = lo; win[1] = hi;
}

int build_dist_plan(const HostHierarchy& H, int rank, int nranks, int min_rows, DistPlan& D)
{
    const int nl = (int)H.L.size();
    if (nranks < 1 || rank < 0 || rank >= nranks) return ERROR_INPUT_PAR;
    D.rank = rank;
    D.nranks = nranks;
    D.L.clear();
    D.L.resize(nl);

    // ownership ranges of every level (classical AMG: coarse rows follow their C points;
    // aggregation hierarchies: equal blocks per level)
    const bool can_partition = true;
    bool has_cf = true;
    for (int l = 0; l + 1 < nl; ++l)
        if (H.L[l].cfmark.n != (size_t)H.L[l].A.row) has_cf = false;
    for (int l = 0; l < nl; ++l) D.L[l].start.assign(nranks + 1, 0);
    if (!has_cf) {
        for (int l = 0; l < nl; ++l)
            for (int r = 0; r <= nranks; ++r) D.L[l].start[r] = (int)((long long)H.L[l].A.row * r / nranks);
    } else {
        const int n0 = H.L[0].A.row;
        for (int r = 0; r <= nranks; ++r) D.L[0].start[r] = (int)((long long)n0 * r / nranks);
        for (int l = 0; l + 1 < nl; ++l) {
            const int  n = H.L[l].A.row;
            const int* cf = H.L[l].cfmark.data();
            std::vector<int>& s = D.L[l].start;
            std::vector<int>& sc = D.L[l + 1].start;
            int cnt = 0, r = 0;
            for (int i = 0; i <= n; ++i) {
                while (r <= nranks && s[r] == i) sc[r++] = cnt;
                if (i < n && cf[i] == CGPT) ++cnt;
            }
            if (cnt != H.L[l + 1].A.row) return ERROR_DATA_STRUCTURE;
        }
    }
    // which levels are distributed: a prefix (once replicated, all coarser levels are too)
    int first_rep = nl;
    for (int l = 0; l < nl; ++l)
        if (nranks == 1 || !can_partition || H.L[l].A.row < min_rows) { first_rep = l; break; }
    // the coarsest level is always replicated: its safe-CG solve runs without communication
    if (nranks > 1 && first_rep > nl - 1) first_rep = nl - 1;
    D.first_replicated = first_rep;

    for (int l = 0; l < nl; ++l) {
        DistLevel&       DL = D.L[l];
        const HostLevel& HL = H.L[l];
        DL.replicated = (l >= first_rep);
        DL.nglobal = HL.A.row;
        DL.row0 = DL.replicated ? 0 : DL.start[rank];
        DL.nloc = DL.replicated ? HL.A.row : DL.start[rank + 1] - DL.start[rank];
        DL.ghosts.clear();
    }

    // ghost sets of distributed levels: union over A_l, R_l (reads V_l), P_{l-1} (reads V_l)
    for (int l = 0; l < first_rep; ++l) {
        DistLevel& DL = D.L[l];
        const int r0 = DL.start[rank], r1 = DL.start[rank + 1];
        collect_ghost_cols(H.L[l].A, r0, r1, r0, r1, DL.ghosts);
        if (H.L[l].has_coarse) {
            const int cr0 = D.L[l + 1].start[rank], cr1 = D.L[l + 1].start[rank + 1];
            collect_ghost_cols(H.L[l].R, cr0, cr1, r0, r1, DL.ghosts);
        }
        if (l > 0) {
            const int f0 = D.L[l - 1].start[rank], f1 = D.L[l - 1].start[rank + 1];
            collect_ghost_cols(H.L[l - 1].P, f0, f1, r0, r1, DL.ghosts);
        }
        sort_unique(DL.ghosts);
    }

    // halo plans: recv = my ghosts grouped by owner; send = what each peer's ghost set takes from me
    for (int l = 0; l < first_rep; ++l) {
        DistLevel& DL = D.L[l];
        DL.recv_off.assign(nranks + 1, 0);
        for (int g : DL.ghosts) DL.recv_off[owner_of(DL.start, g) + 1]++;
        for (int r = 0; r < nranks; ++r) DL.recv_off[r + 1] += DL.recv_off[r];
        // send lists: recompute every peer's ghost set restricted to my range
        DL.send_off.assign(nranks + 1, 0);
        DL.send_idx.clear();
        const int m0 = DL.start[rank], m1 = DL.start[rank + 1];
        for (int q = 0; q < nranks; ++q) {
            std::vector<int> need;
            if (q != rank) {
                const int q0 = DL.start[q], q1 = DL.start[q + 1];
                auto collect_mine = [&](const HostCSR& M, int a, int b) {
                    for (int i = a; i < b; ++i)
                        for (int k = M.ia[i]; k < M.ia[i + 1]; ++k) {
                            const int c = M.ja[k];
                            if (c >= m0 && c < m1) need.push_back(c);
                        }
                };
                collect_mine(H.L[l].A, q0, q1);
                if (H.L[l].has_coarse) collect_mine(H.L[l].R, D.L[l + 1].start[q], D.L[l + 1].start[q + 1]);
                if (l > 0) collect_mine(H.L[l - 1].P, D.L[l - 1].start[q], D.L[l - 1].start[q + 1]);
                sort_unique(need);
            }
            for (int c : need) DL.send_idx.push_back(c - m0);
            DL.send_off[q + 1] = (int)DL.send_idx.size();
        }
    }

    // local operators
    for (int l = 0; l < nl; ++l) {
        DistLevel&       DL = D.L[l];
        const HostLevel& HL = H.L[l];
        if (DL.replicated) continue;  // replicated levels use the global matrices directly
        const int r0 = DL.start[rank], r1 = DL.start[rank + 1];
        const int ncol = DL.nloc + (int)DL.ghosts.size();
        extract_rows(HL.A, r0, r1, r0, r1, &DL.ghosts, ncol, DL.A);
        if (HL.has_coarse) {
            DistLevel& DC = D.L[l + 1];
            const int cr0 = DC.start[rank], cr1 = DC.start[rank + 1];
            // R_l: coarse rows I own (also when level l+1 is replicated: my slice, all-gathered later)
            extract_rows(HL.R, cr0, cr1, r0, r1, &DL.ghosts, ncol, DL.R);
            // P_l: fine rows I own; operand lives on level l+1
            if (DC.replicated) extract_rows(HL.P, r0, r1, 0, DC.nglobal, nullptr, DC.nglobal, DL.P);
            else extract_rows(HL.P, r0, r1, cr0, cr1, &DC.ghosts, (cr1 - cr0) + (int)DC.ghosts.size(), DL.P);
            find_row_window(DL.R, DL.nloc, DIST_WIN_ALIGN, DL.winR);          // R reads this level's vectors
            if (!DC.replicated) find_row_window(DL.P, cr1 - cr0, DIST_WIN_ALIGN, DL.winP);   // P reads the next level's
        }
        find_row_window(DL.A, DL.nloc, DIST_WIN_ALIGN, DL.winA);
    }
    return FASP_SUCCESS;
}

// ---- block (BSR) hierarchy: the CSR planner on the block pattern ---------------------------------------------------
int build_dist_plan_bsr(const HostHierarchyBSR& H, int rank, int nranks, int min_rows, DistPlan& D, std::vector<DistLocalBSR>& local)
{
    const int nl = (int)H.L.size();
    // pattern hierarchy: a block row is a row; the "value" of an entry is its block index in the global matrix
    HostHierarchy Hp;
    Hp.L.resize((size_t)nl);
    auto pattern = [](const HostBSR& B, HostCSR& C) {
        C.row = B.ROW; C.col = B.COL; C.nnz = B.NNZ;
        C.ia.view(const_cast<int*>(B.ia.data()), (size_t)B.ROW + 1);
        C.ja.view(const_cast<int*>(B.ja.data()), (size_t)std::max(B.NNZ, 1));
        C.val.alloc((size_t)std::max(B.NNZ, 1));
#pragma omp parallel for schedule(static)
        for (int k = 0; k < B.NNZ; ++k) C.val[k] = (double)k;
    };
    for (int l = 0; l < nl; ++l) {
        pattern(H.L[(size_t)l].A, Hp.L[(size_t)l].A);
        Hp.L[(size_t)l].has_coarse = H.L[(size_t)l].has_coarse;
        if (H.L[(size_t)l].has_coarse) { pattern(H.L[(size_t)l].P, Hp.L[(size_t)l].P); pattern(H.L[(size_t)l].R, Hp.L[(size_t)l].R); }
    }
    const int st = build_dist_plan(Hp, rank, nranks, min_rows, D);
    if (st < 0) return st;
    local.clear();
    local.resize((size_t)nl);
    auto gather = [](const HostCSR& Lc, const HostBSR& G, HostBSR& out) {
        const int nb = G.nb, nb2 = nb * nb;
        out.ROW = Lc.row; out.COL = Lc.col; out.NNZ = Lc.nnz; out.nb = nb;
        out.ia.alloc((size_t)Lc.row + 1); out.ja.alloc((size_t)std::max(Lc.nnz, 1)); out.val.alloc((size_t)std::max(Lc.nnz, 1) * nb2);
        std::memcpy(out.ia.data(), Lc.ia.data(), sizeof(int) * ((size_t)Lc.row + 1));
#pragma omp parallel for schedule(static)
        for (int k = 0; k < Lc.nnz; ++k) {
            out.ja[k] = Lc.ja[k];
            std::memcpy(out.val.data() + (size_t)k * nb2, G.val.data() + (size_t)Lc.val[k] * nb2, sizeof(double) * nb2);
        }
    };
    for (int l = 0; l < nl; ++l) {
        DistLevel& DL = D.L[(size_t)l];
        if (DL.replicated) continue;
        gather(DL.A, H.L[(size_t)l].A, local[(size_t)l].A);
        if (H.L[(size_t)l].has_coarse) { gather(DL.P, H.L[(size_t)l].P, local[(size_t)l].P); gather(DL.R, H.L[(size_t)l].R, local[(size_t)l].R); }
        DL.A = HostCSR(); DL.P = HostCSR(); DL.R = HostCSR();   // the index-carrying patterns are no longer needed
    }
    return FASP_SUCCESS;
}

}  // namespace fasp
