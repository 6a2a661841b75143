// seq_sched.cpp -- host side of the sequential sweeps' split form (seq_split.hip.h): strips, chunks, slots, ghost lists, virtual rows,
// the rest CSR and the per-row records of one (level, sweep kind), built once on first use.  Pure host code in a translation
// unit of its own: the OpenMP loops below are compiled by g++ (the HIP translation unit is compiled by clang with
// -fopenmp=libgomp, which parses the directives and generates NO parallel code -- measured: the fill ran on one thread).
#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <utility>
#include <vector>

#include "fasp_internal.h"
#include "seq_sched.h"

namespace fasp {


// ---------------------------------------------------------------------------
// The chain form (seq_sched.h, seq_chain.hip.h).  Returns FASP_SUCCESS, 1 when the form does not apply (a row that is left alone,
// more positions than 16-bit columns address), or a negative error code.  Everything is row-parallel: the sequential part of a
// schedule -- the dependency classes, which decide WHETHER this form is taken -- has been done by the caller.
// ---------------------------------------------------------------------------
static int build_chain_host(const HostCSR& A, const int* seq, int ns, const Buf<int>& pos, int n1_blocks, bool timing, SplitHost& H)
{
    const int n = A.row;
    if (ns <= 0 || ns > 65535 - 128) return 1;
    const int nb = (ns + 63) / 64, npad = nb * 64;
    double tl = wall_seconds();
    auto lap = [&](const char* what) { if (timing) { const double t = wall_seconds(); std::printf("    [chain schedule] %-28s %.3f s\n", what, t - tl); tl = t; } };
    // ---- rows the reference leaves alone (no usable diagonal) have no chain form
    int alone_rows = 0;
#pragma omp parallel for schedule(static) reduction(+ : alone_rows)
    for (int q = 0; q < ns; ++q) {
        const int i = seq[q];
        double dg = 0.0;
        for (int k = A.ia[i]; k < A.ia[i + 1]; ++k)
            if (A.ja[k] == i) dg = A.val[k];
        if (!(std::fabs(dg) > SMALLREAL)) ++alone_rows;
    }
    if (alone_rows) return 1;   // (rows the reference leaves alone: the dataflow form handles them)
    int n1b = n1_blocks;
    // Tier 1 is bound by what ONE compute unit streams (10 bytes per entry, padded to the block's longest row, against a block of 64
    // rows per 2 us) and by its helpers' time; tier 2 needs the window as slack (exported -> gate -> last gathers -> G2 -> importer:
    // 8-10 us = four to five blocks).  Measured on levels 5-8 of P7(256), all sweep kinds (profiles/r05_gs_chain.txt): six blocks beat
    // four, eight, twelve and "as many as stay below 64 entries per row" (9-32 blocks on the wider levels) everywhere but on the
    // F rows of the last level.  That was measured on sweeps repeated back to back, tier 1's entries in the Infinity Cache; INSIDE a solve
    // they stream from memory, through the chain's own compute unit, and the smaller window wins (whole solves, tools/lab/gs_build_ab.py:
    // GS-CF at 256^3 291.2 / 288.3 / 285.2 / 284.0 / 283.7 ms for 6 / 5 / 4 / 3 / 2 blocks; SOR 258.2 -> 253.9; GS-CF at 128^3 48.5 -> 47.5).
    if (n1b <= 0) n1b = 2;
    n1b = std::max(1, std::min(48, n1b));
    ChainHost& C = H.C;
    C.nb = nb; C.npad = npad; C.n1b = n1b;
    C.rx = 64 * (n1b + CHAIN_HA + 3);
    C.rg = 64 * (CHAIN_HA + 2);
    // ---- per block: steps of the two tiers (longest row of the block), rest offsets by position
    C.blk.alloc((size_t)nb);
    Buf<int> nrest((size_t)npad);
    auto tier_of = [&](int K, int pj) { const int db = K - (pj >> 6); return db <= 1 ? 0 : db <= n1b + 1 ? 1 : 2; };
#pragma omp parallel for schedule(dynamic, 4)
    for (int K = 0; K < nb; ++K) {
        int m1 = 0, m2 = 0;
        for (int jl = 0; jl < 64; ++jl) {
            const int q = K * 64 + jl;
            if (q >= ns) { nrest[(size_t)q] = 0; continue; }
            const int i = seq[q];
            int c1 = 0, c2 = 0, cr = 0;
            for (int k = A.ia[i]; k < A.ia[i + 1]; ++k) {
                const int j = A.ja[k];
                if (j == i) continue;
                const int pj = j < n ? pos[j] : -1;
                if ((unsigned)pj < (unsigned)q) { const int t = tier_of(K, pj); c1 += t == 1; c2 += t == 2; }
                else ++cr;
            }
            nrest[(size_t)q] = cr;
            m1 = std::max(m1, c1); m2 = std::max(m2, c2);
        }
        // (steps come in groups of eight: the kernels fetch whole groups)
        C.blk[(size_t)K].t1_n = (m1 + 7) & ~7; C.blk[(size_t)K].t2_n = (m2 + 7) & ~7;
    }
    long long o1 = 0, o2 = 0;
    for (int K = 0; K < nb; ++K) {
        C.blk[(size_t)K].t1_off = (int)o1; C.blk[(size_t)K].t2_off = (int)o2;
        o1 += C.blk[(size_t)K].t1_n; o2 += C.blk[(size_t)K].t2_n;
        if (o1 > 0x1ffffffll || o2 > 0x1ffffffll) return 1;
    }
    C.t1_steps = o1; C.t2_steps = o2;
    H.ria.alloc((size_t)npad + 1);
    H.ria[0] = 0;
    long long nr = 0;
    for (int q = 0; q < npad; ++q) { nr += nrest[(size_t)q]; if (nr > 0x7fffffffll) return ERROR_INPUT_PAR; H.ria[(size_t)q + 1] = (int)nr; }
    lap("tiers and offsets");
    // ---- fill
    const size_t band_steps = (size_t)nb * 64 + CHAIN_PF;
    C.band.alloc(band_steps * 128);
    C.drd.alloc(2 * (size_t)npad);
    C.t1v.alloc(std::max<size_t>((size_t)o1 * 64, 1)); C.t1c.alloc(std::max<size_t>((size_t)o1 * 64, 1));
    C.t2v.alloc(std::max<size_t>((size_t)o2 * 64, 1)); C.t2c.alloc(std::max<size_t>((size_t)o2 * 64, 1));
    C.t1need.alloc(std::max<size_t>((size_t)o1 / 8, 1)); C.t2need.alloc(std::max<size_t>((size_t)o2 / 8, 1));
    H.rja.alloc((size_t)std::max<long long>(nr, 1)); H.rval.alloc((size_t)std::max<long long>(nr, 1));
    H.tr.alloc(2 * (size_t)npad);
    H.dr.alloc(2);   // (the split form's per-position pairs: here C.drd)
    long long nband = 0, nt1 = 0, nt2 = 0;
#pragma omp parallel reduction(+ : nband, nt1, nt2)
    {
        std::vector<std::pair<int, double>> e1, e2;
#pragma omp for schedule(dynamic, 4)
        for (int K = 0; K < nb + 1; ++K) {
            if (K == nb) { std::memset(C.band.data() + (size_t)nb * 64 * 128, 0, sizeof(double) * CHAIN_PF * 128); continue; }
            double* bd = C.band.data() + (size_t)K * 64 * 128;   // [step c][lane j]{TA, TB}: TA = l(64 K + j, 64 K + c), TB = l(64 (K + 1) + j, 64 K + c)
            // TA of this block and TB of the PREVIOUS block's plane are written by the rows of this block: zero first (each plane half once)
            for (int c = 0; c < 64; ++c) for (int j = 0; j < 64; ++j) bd[((size_t)c * 64 + j) * 2] = 0.0;
            if (K > 0) { double* bp = bd - 64 * 128; for (int c = 0; c < 64; ++c) for (int j = 0; j < 64; ++j) bp[((size_t)c * 64 + j) * 2 + 1] = 0.0; }
            if (K == nb - 1) for (int c = 0; c < 64; ++c) for (int j = 0; j < 64; ++j) bd[((size_t)c * 64 + j) * 2 + 1] = 0.0;   // (no block behind the last)
            const ChainBlk B = C.blk[(size_t)K];
            double* v1 = C.t1v.data() + (size_t)B.t1_off * 64; unsigned short* c1 = C.t1c.data() + (size_t)B.t1_off * 64;
            double* v2 = C.t2v.data() + (size_t)B.t2_off * 64; unsigned short* c2 = C.t2c.data() + (size_t)B.t2_off * 64;
            for (size_t t = 0; t < (size_t)B.t1_n * 64; ++t) { v1[t] = 0.0; c1[t] = (unsigned short)C.rx; }
            for (size_t t = 0; t < (size_t)B.t2_n * 64; ++t) { v2[t] = 0.0; c2[t] = (unsigned short)npad; }
            auto cidx = [](size_t st, int lane) { return ((st >> 3) * 64 + (size_t)lane) * 8 + (st & 7); };   // columns: [group of 8 steps][lane][8] (one 16-byte load per lane and group)
            int* nd1 = C.t1need.data() + B.t1_off / 8; int* nd2 = C.t2need.data() + B.t2_off / 8;
            for (int g = 0; g < B.t1_n / 8; ++g) nd1[g] = -1;
            for (int g = 0; g < B.t2_n / 8; ++g) nd2[g] = -1;
            for (int jl = 0; jl < 64; ++jl) {
                const int q = K * 64 + jl;
                if (q >= ns) {   // padding row: x = 0 / 1, no row of u
                    C.drd[2 * (size_t)q] = 1.0; C.drd[2 * (size_t)q + 1] = 1.0;
                    H.tr[2 * (size_t)q] = 0; H.tr[2 * (size_t)q + 1] = -1;
                    continue;
                }
                const int i = seq[q];
                e1.clear(); e2.clear();
                double dg = 0.0;
                size_t kr = (size_t)H.ria[q];
                for (int k = A.ia[i]; k < A.ia[i + 1]; ++k) {
                    const int j = A.ja[k];
                    if (j == i) { dg = A.val[k]; continue; }   // the last diagonal hit, as the reference's loop leaves it
                    const int pj = j < n ? pos[j] : -1;
                    if ((unsigned)pj < (unsigned)q) {
                        const int t = tier_of(K, pj);
                        if (t == 0) {
                            // (a column that occurs twice in a row -- no setup of this library produces one -- is merged: one coefficient per (row, column))
                            const int c = pj & 63;
                            double* slot = (pj >> 6) == K ? &bd[((size_t)c * 64 + jl) * 2] : &(bd - 64 * 128)[((size_t)c * 64 + jl) * 2 + 1];
                            *slot += A.val[k];
                            ++nband;
                        } else if (t == 1) e1.emplace_back(pj, A.val[k]);
                        else e2.emplace_back(pj, A.val[k]);
                    } else { H.rja[kr] = j; H.rval[kr] = A.val[k]; ++kr; }
                }
                auto by_pos = [](const std::pair<int, double>& x, const std::pair<int, double>& y) { return x.first < y.first; };
                std::stable_sort(e1.begin(), e1.end(), by_pos); std::stable_sort(e2.begin(), e2.end(), by_pos);
                nt1 += (long long)e1.size(); nt2 += (long long)e2.size();
                // right-aligned: the newest columns are the last steps of the block
                for (size_t t = 0; t < e1.size(); ++t) { const size_t st = (size_t)B.t1_n - e1.size() + t; v1[st * 64 + jl] = e1[t].second; c1[cidx(st, jl)] = (unsigned short)(e1[t].first % C.rx); nd1[st / 8] = std::max(nd1[st / 8], e1[t].first >> 6); }
                for (size_t t = 0; t < e2.size(); ++t) { const size_t st = (size_t)B.t2_n - e2.size() + t; v2[st * 64 + jl] = e2[t].second; c2[cidx(st, jl)] = (unsigned short)e2[t].first; nd2[st / 8] = std::max(nd2[st / 8], e2[t].first >> 6); }
                C.drd[2 * (size_t)q] = dg; C.drd[2 * (size_t)q + 1] = 1.0 / dg;
                H.tr[2 * (size_t)q] = 0; H.tr[2 * (size_t)q + 1] = i;
            }
        }
    }
    C.nband = nband; C.nt1 = nt1; C.nt2 = nt2;
    lap("fill");
    H.chain = true;
    H.ns = npad; H.nrows = ns; H.nvirt = 0; H.L = 64; H.nolower = false; H.pfs = 8; H.kt = 0; H.nstrips = 0; H.nchunk = 0; H.maxent = C.rx + C.rg;
    H.nghost = 0; H.slot_bytes = 0; H.nrest = nr; H.flow_ok = true; H.par = 1;
    H.cptr.assign(2, 0);
    H.chunks.alloc(4); H.cstrip.alloc(1); H.lchunks.alloc(1); H.gpos.alloc(1); H.slots.alloc(16);
    const double avg_rest = ns > 0 ? (double)nr / ns : 0.0;
    H.LR = 1;
    while (H.LR < 64 && 4 * H.LR < avg_rest) H.LR *= 2;
    return FASP_SUCCESS;
}

int build_split_host(const HostCSR& A, const int* seq, int ns, int strip_kb, int seq_lanes, bool timing, SplitHost& H, int team, int spine, int chain, int chain_n1)
{
    HostThreads host_team;   // (bounded OpenMP team for the row-parallel loops below; the dependency pass itself is sequential)
    if (team > 0) omp_set_num_threads(std::min(team, omp_get_max_threads()));   // (bounded OpenMP team for the row-parallel loops below; the dependency pass itself is sequential)
    const int n = A.row;
    double tl = wall_seconds();
    auto lap = [&](const char* what) { if (timing) { const double t = wall_seconds(); std::printf("    [sweep schedule] %-28s %.3f s\n", what, t - tl); tl = t; } };
    Buf<int> pos((size_t)std::max(n, 1));
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) pos[i] = -1;
#pragma omp parallel for schedule(static)
    for (int q = 0; q < ns; ++q) pos[seq[q]] = q;
    // lanes per row from a sample-free pass over the row lengths would need the lower counts; they come out of the dependency pass
    // below, which therefore runs first with nothing but the matrix: class (dependency level) and number of lower entries per row
    Buf<int> lev((size_t)std::max(ns, 1)), nlow((size_t)std::max(ns, 1)), nrest((size_t)std::max(ns, 1));
    int nlev = ns > 0 ? 1 : 0;
    long long lower_total = 0, upper_total = 0;   // entries that read rows the sweep has visited / has yet to visit
    // The ONE sequential pass over the matrix: dependency class and number of lower entries of every row, and the strips --
    // contiguous ranges of the sweep sequence, closed when the lower part reaches the target size (12 bytes per entry + 40 per
    // row) or the LDS is full (own rows + distinct earlier rows read + the constant).
    // Strip size: the caller's, or -- when the caller leaves it to the schedule (<= 0, the default) -- what the levels of P7(256) measured best with once the deep
    // levels had left for the chain form (profiles/r05_gs_chain.txt, tools/perf_gs_levels.py 256 seq_strip_kb=...): sweeps over ALL rows
    // of a wide level (natural order: twice the lower entries per row of a C / F sweep) 1 MB (level 0 2277 -> 1930 us, level 1 2029 ->
    // 1745), the long-row levels that stay in the dataflow form 256 KB (level 4: C rows 885 -> 756, F rows 620 -> 565), 512 KB otherwise.
    if (strip_kb <= 0) {
        strip_kb = 512;
        if (ns == n && n >= 1000000) strip_kb = 1024;
        else if (n > 0 && (double)A.nnz / n >= 128.0) strip_kb = 256;
    }
    const long long target = std::max(16, strip_kb) * 1024ll;
    constexpr int VCAP_MAX = (TRI_PFMAX - TRI_SPINE) * 64 + TRI_SPINE;   // slots of a work item with 64 lanes and a spine
    std::vector<int> sq0(1, 0), sng;   // first sequence index of every strip (+ end), ghosts per strip
    bool flow_ok = true;
    {
        Buf<int> gmark((size_t)std::max(ns, 1));
#pragma omp parallel for schedule(static)
        for (int q = 0; q < ns; ++q) gmark[q] = -1;
        int sid = 0, rows = 0, ng = 0;
        long long bytes = 0;
        std::vector<int> fresh;
        for (int q = 0; q < ns; ++q) {
            const int i = seq[q], q0 = sq0.back();
            int l = 0, c = 0, dg = 0;
            fresh.clear();
            for (int k = A.ia[i]; k < A.ia[i + 1]; ++k) {
                const int j = A.ja[k];
                if (j == i) { ++dg; continue; }
                if (j >= n) continue;
                const int pj = pos[j];
                if ((unsigned)pj < (unsigned)q) {
                    l = std::max(l, lev[pj]); ++c;
                    if (pj < q0 && gmark[pj] != sid) { gmark[pj] = sid; fresh.push_back(pj); }
                } else if (pj > q) ++upper_total;
            }
            lev[q] = l + 1; nlow[q] = c; nrest[q] = A.ia[i + 1] - A.ia[i] - c - dg;
            nlev = std::max(nlev, l + 1);
            lower_total += c;
            // (a row of more entries than a work item holds comes with virtual rows, below: counted here as if the item were the largest
            // possible -- the lanes per row are not known yet; verified once they are)
            const int vx = c > VCAP_MAX ? (c - VCAP_MAX) / (VCAP_MAX - 1) + 1 : 0;
            const long long rb = 40ll * (1 + vx) + 12ll * (c + vx);
            if (rows > 0 && (bytes + rb > target || rows + 1 + vx + ng + (int)fresh.size() + 1 > FLOW_LDS_ENT || rows + 1 + vx >= 0xffff)) {
                // close the strip in front of this row; the row opens the next one: every earlier row it reads is a ghost now
                sng.push_back(ng);
                sq0.push_back(q);
                ++sid; rows = 0; ng = 0; bytes = 0;
                for (int k = A.ia[i]; k < A.ia[i + 1]; ++k) {
                    const int j = A.ja[k];
                    if (j == i || j >= n) continue;
                    const int pj = pos[j];
                    if ((unsigned)pj < (unsigned)q && gmark[pj] != sid) { gmark[pj] = sid; ++ng; }
                }
            } else ng += (int)fresh.size();
            if (1 + vx + ng + 1 > FLOW_LDS_ENT) flow_ok = false;   // one row that reads more than the LDS holds: no dataflow form for this sweep
            rows += 1 + vx; bytes += rb;
        }
        if (ns > 0) { sng.push_back(ng); sq0.push_back(ns); }
    }
    lap("classes and strips");
    if (lower_total == 0) {
        // No row of the sweep reads another one's new value (the C rows / the F rows of a 7-point level 0): the sweep is pass (1)
        // and an elementwise update.  Positions = sequence order; nothing but the rest CSR and the per-row records is needed.
        H.ria.alloc((size_t)ns + 1);
        H.ria[0] = 0;
        long long nr = 0;
        for (int q = 0; q < ns; ++q) { nr += nrest[q]; if (nr > 0x7fffffffll) return ERROR_INPUT_PAR; H.ria[(size_t)q + 1] = (int)nr; }
        H.rja.alloc((size_t)std::max<long long>(nr, 1)); H.rval.alloc((size_t)std::max<long long>(nr, 1));
        H.tr.alloc(2 * (size_t)std::max(ns, 1)); H.dr.alloc(2 * (size_t)std::max(ns, 1));
        H.chunks.alloc(4); H.cstrip.alloc(1); H.lchunks.alloc(1); H.gpos.alloc(1); H.slots.alloc(16);
#pragma omp parallel for schedule(static)
        for (int q = 0; q < ns; ++q) {
            const int i = seq[q];
            size_t kr = (size_t)H.ria[q];
            double dg = 0.0;
            for (int k = A.ia[i]; k < A.ia[i + 1]; ++k) {
                if (A.ja[k] == i) { dg = A.val[k]; continue; }
                H.rja[kr] = A.ja[k]; H.rval[kr] = A.val[k]; ++kr;
            }
            const bool alone = !(std::fabs(dg) > SMALLREAL);
            H.dr[2 * (size_t)q] = dg; H.dr[2 * (size_t)q + 1] = alone ? 0.0 : 1.0 / dg;
            H.tr[2 * (size_t)q] = alone ? (int)0x80000000 : 0; H.tr[2 * (size_t)q + 1] = i;
        }
        lap("rest (no lower entries)");
        H.cptr.assign(2, 0);
        H.independent = upper_total == 0;   // no row of the sweep reads another row of the sweep at all: one pass that updates in place
        H.ns = ns; H.nrows = ns; H.nvirt = 0; H.nclasses = nlev; H.L = 1; H.nolower = true; H.pfs = 4; H.nstrips = 0; H.nchunk = 0; H.maxent = 0; H.nghost = 0; H.slot_bytes = 0; H.nrest = nr; H.flow_ok = true;
        const double avg = ns > 0 ? (double)nr / ns : 0.0;
        H.LR = 1;
        while (H.LR < 64 && 4 * H.LR < avg) H.LR *= 2;
        return FASP_SUCCESS;
    }
    // ---- chain-bound sweeps (classes of a few rows: the deep levels) take the chain form where it applies (seq_sched.h): one
    // wavefront walks the rows in sweep order at ~30 ns per row, whatever the dependency graph looks like -- against 0.4-0.8 us per
    // dependency CLASS in the dataflow form below
    H.nclasses = nlev;
    // (break-even, measured on P7(256): the dataflow form pays 0.49-0.57 us per class at 29 rows per class, the chain form 31-36 ns per row)
    if (chain == 2 || (chain == 1 && ns >= 256 && (long long)ns < 16ll * nlev)) {
        const int st = build_chain_host(A, seq, ns, pos, chain_n1, timing, H);
        if (st == FASP_SUCCESS) { H.nclasses = nlev; return FASP_SUCCESS; }
        if (st < 0) return st;
    }
    // lanes per row of the triangular part: TRI_PF * L slots cover the lower entries of 90 % of the rows
    int len90 = 0;
    {
        std::vector<long long> hist(258, 0);
        for (int q = 0; q < ns; ++q) hist[std::min(nlow[q], 257)]++;
        long long acc = 0;
        for (int v = 0; v < 258; ++v) { acc += hist[v]; if (acc * 10 >= (long long)ns * 9) { len90 = v; break; } }
    }
    // wide classes (levels 1-2 of a 3-D problem: hundreds of rows each): what a chunk costs there is instructions, per LANE
    // mostly: half the lanes with twice the rounds is less work per row.  Narrow classes (a few rows: the deep levels) are a
    // latency chain: more lanes, shorter chains.
    const bool wide = nlev > 0 && ns / nlev >= 128;
    int L = 1;
    while (L < 64 && (wide ? 6 : TRI_PF) * L < len90) L *= 2;   // (six rounds on the wide levels: measured at 128^3 against eight, level 1 F rows 472 -> 424 us, level 2 381 -> 360)
    if (seq_lanes > 0) { L = 1; while (L < 64 && L < seq_lanes) L *= 2; }
    // rounds per chunk of this schedule: four where no row needs more (then the kernels with room for four run it)
    int nlowmax = 0;
#pragma omp parallel for schedule(static) reduction(max : nlowmax)
    for (int q = 0; q < ns; ++q) nlowmax = std::max(nlowmax, nlow[q]);
    // spine rounds (seq_sched.h): on chain-bound levels whose rows take eight rounds anyway (measured at 128^3 on the levels of
    // P7: per sweep 13-22 % less there; schedules of four rounds gain nothing from two more for a spine)
    int KT = (spine == 2 ? L >= 2 : (spine == 1 && !wide && L >= 8 && (nlowmax + L - 1) / L > 4)) ? TRI_SPINE : 0;
    // A work item holds CAP lower entries.  A row with more is split (seq_sched.h, "virtual rows"); its virtual rows must fit its own slots.
    auto cap_of = [&](int lanes) { return (TRI_PFMAX - KT) * lanes + KT; };
    while (L < 64 && (long long)cap_of(L) * (cap_of(L) - 1) < nlowmax) L *= 2;
    if ((long long)cap_of(L) * (cap_of(L) - 1) < nlowmax) { H.flow_ok = false; return 1; }
    const int rpw = 64 / L;   // rows per chunk (one wavefront)
    const int CAP = cap_of(L);
    auto rounds_of = [&](int mx) { return std::max(1, std::min(TRI_PFMAX, KT + (std::max(0, mx - KT) + L - 1) / L)); };   // rounds a chunk stores, counted from the last
    if (!flow_ok) { H.flow_ok = false; return 1; }   // (the caller falls back to whole-row level scheduling: build_schedule + k_seq_level)
    const int nstrips = (int)sq0.size() - 1;
    const int PFS = (KT || (nlowmax + L - 1) / L > 4) ? TRI_PFMAX : 4;   // (a spine comes with eight rounds: the kernels with room for four have none)
    // ---- virtual rows: V(q) per sequence index, their classes.  Classes are DOUBLED from here on: a row of class l is 2 l, a
    // virtual row whose newest entry has class l' is 2 l' + 1 -- above everything it reads, below the row that reads it
    // (l' <= l - 1), never in one chunk or one launch-per-class with it.
    auto nvirt_of = [&](int c) { return c > CAP ? (c - CAP + CAP - 2) / (CAP - 1) : 0; };
    Buf<int> voff((size_t)ns + 1);
    voff[0] = 0;
    for (int q = 0; q < ns; ++q) voff[(size_t)q + 1] = voff[q] + nvirt_of(nlow[q]);
    const int nvirt = voff[ns];
    if ((long long)ns + nvirt > 0x7fffffffll) return ERROR_INPUT_PAR;
    const int npos = ns + nvirt;
    Buf<int> vcls((size_t)std::max(nvirt, 1)), vcnt((size_t)std::max(nvirt, 1));   // class (doubled) and entry count of every virtual row
    // the lower entries of a row in the order of their dependency classes (= the order in which they become available), by
    // (class, sequence): the same order however the strips are cut
    auto sorted_lower = [&](int q, std::vector<std::pair<int, double>>& low) {
        const int i = seq[q];
        low.clear();
        for (int k = A.ia[i]; k < A.ia[i + 1]; ++k) {
            const int j = A.ja[k];
            if (j == i || j >= n) continue;
            const int pj = pos[j];
            if ((unsigned)pj < (unsigned)q) low.emplace_back(pj, A.val[k]);
        }
        std::stable_sort(low.begin(), low.end(), [&](const std::pair<int, double>& x, const std::pair<int, double>& y) {
            return lev[x.first] != lev[y.first] ? lev[x.first] < lev[y.first] : x.first < y.first; });
    };
    if (nvirt > 0) {
#pragma omp parallel
        {
            std::vector<std::pair<int, double>> low;
#pragma omp for schedule(dynamic, 64)
            for (int q = 0; q < ns; ++q) {
                const int V = voff[(size_t)q + 1] - voff[q];
                if (!V) continue;
                sorted_lower(q, low);
                const int nlo = (int)low.size(), NV = nlo - (CAP - V);   // entries that go to the virtual rows: the oldest
                for (int k = 0; k < V; ++k) {
                    const int c = std::min(CAP, NV - k * CAP);
                    vcnt[(size_t)voff[q] + k] = c;
                    vcls[(size_t)voff[q] + k] = 2 * lev[low[(size_t)(k * CAP + c - 1)].first] + 1;
                }
            }
        }
    }
    lap("virtual rows");
    // ---- per strip: items (rows and virtual rows) by (class, sequence, part), chunks; positions = strip base + local index
    Buf<int> newpos((size_t)std::max(ns, 1)), vpos((size_t)std::max(nvirt, 1));   // position of row q / of virtual row voff[q] + k
    Buf<int> itq((size_t)std::max(npos, 1)), itk((size_t)std::max(npos, 1));       // item at position p: its row, its part (V(q): the row itself)
    std::vector<int> schunks((size_t)nstrips + 1, 0);
    std::vector<long long> sbytes((size_t)nstrips + 1, 0);
    auto item_class = [&](int q, int k) { const int V = voff[(size_t)q + 1] - voff[q]; return k < V ? vcls[(size_t)voff[q] + k] : 2 * lev[q]; };
    auto item_count = [&](int q, int k) { const int V = voff[(size_t)q + 1] - voff[q]; return k < V ? vcnt[(size_t)voff[q] + k] : (V ? CAP : nlow[q]); };
    int bad_strip = 0;
#pragma omp parallel reduction(+ : bad_strip)
    {
        std::vector<int> cnt;
#pragma omp for schedule(dynamic, 1)
        for (int s = 0; s < nstrips; ++s) {
            const int q0 = sq0[s], q1 = sq0[s + 1], p0 = q0 + voff[q0], p1 = q1 + voff[q1];
            if (p1 - p0 + sng[(size_t)s] + 1 > FLOW_LDS_ENT || p1 - p0 > 0xffff) ++bad_strip;
            int lmin = 2 * lev[q0], lmax = 2 * lev[q0];
            for (int q = q0; q < q1; ++q)
                for (int k = 0, V = voff[(size_t)q + 1] - voff[q]; k <= V; ++k) { const int c = item_class(q, k); lmin = std::min(lmin, c); lmax = std::max(lmax, c); }
            cnt.assign((size_t)(lmax - lmin + 2), 0);
            for (int q = q0; q < q1; ++q)
                for (int k = 0, V = voff[(size_t)q + 1] - voff[q]; k <= V; ++k) cnt[(size_t)(item_class(q, k) - lmin + 1)]++;
            for (size_t l = 1; l < cnt.size(); ++l) cnt[l] += cnt[l - 1];
            for (int q = q0; q < q1; ++q)
                for (int k = 0, V = voff[(size_t)q + 1] - voff[q]; k <= V; ++k) {
                    const int p = p0 + cnt[(size_t)(item_class(q, k) - lmin)]++;
                    itq[p] = q; itk[p] = k;
                    if (k < V) vpos[(size_t)voff[q] + k] = p; else newpos[q] = p;
                }
            // chunks: runs of one class, rpw items at most
            int nch = 0;
            long long by = 0;
            for (int p = p0; p < p1;) {
                const int l = item_class(itq[p], itk[p]);
                int e = p, mx = 0;
                while (e < p1 && e - p < rpw && item_class(itq[e], itk[e]) == l) { mx = std::max(mx, item_count(itq[e], itk[e])); ++e; }
                const int pf = rounds_of(mx);
                by += 16ll * (e - p) * L * (1 + PFS / 2 - (PFS - pf) / 2);
                ++nch; p = e;
            }
            schunks[(size_t)s + 1] = nch; sbytes[(size_t)s + 1] = by;
        }
    }
    if (bad_strip) { H.flow_ok = false; return 1; }   // (the estimate of the virtual rows in the strip cutting was too low: whole-row level scheduling)
    lap("order and chunks per strip");
    // How the strips overlap in dependency depth: a strip works on class c only if c lies in the range of classes of its rows.  On
    // a chain-bound level (classes of a few rows) the sweep is a front that moves through the strips; more workgroups than a
    // few times the number of strips that share a class only wait -- and their importer waves poll memory while they do
    // (measured on the deep levels of P7(256): 3-8 % per sweep with the launch capped at 2-16 workgroups).  Wide levels are
    // not a front (every grid plane is at work at once): no cap there.
    int par = nstrips;
    if (nstrips > 1 && !wide) {
        std::vector<int> diff((size_t)nlev + 2, 0);
        for (int s = 0; s < nstrips; ++s) {
            int lmin = lev[sq0[s]], lmax = lmin;
            for (int q = sq0[s]; q < sq0[s + 1]; ++q) { lmin = std::min(lmin, lev[q]); lmax = std::max(lmax, lev[q]); }
            diff[(size_t)lmin]++; diff[(size_t)lmax + 1]--;
        }
        int cur = 0, mx = 0;
        for (int l = 0; l <= nlev; ++l) { cur += diff[(size_t)l]; mx = std::max(mx, cur); }
        par = std::max(1, mx);
        if (timing) std::printf("    [sweep schedule] at most %d of %d strips share a dependency class\n", par, nstrips);
    }
    H.par = std::max(1, par);
    lap("strip overlap");
    std::vector<int> sghost((size_t)nstrips + 1, 0);
    for (int s = 0; s < nstrips; ++s) { schunks[(size_t)s + 1] += schunks[(size_t)s]; sbytes[(size_t)s + 1] += sbytes[(size_t)s]; sghost[(size_t)s + 1] = sghost[(size_t)s] + sng[(size_t)s]; }
    const int nchunk = nstrips ? schunks[(size_t)nstrips] : 0;
    const long long slot_bytes = nstrips ? sbytes[(size_t)nstrips] : 0, nghost = nstrips ? sghost[(size_t)nstrips] : 0;
    // ---- rest offsets by position (virtual rows have no rest)
    H.ria.alloc((size_t)npos + 1);
    Buf<int>& ria = H.ria;
    ria[0] = 0;
    long long nrest_total = 0;
    for (int p = 0; p < npos; ++p) {
        const int q = itq[p];
        if (itk[p] == voff[(size_t)q + 1] - voff[q]) nrest_total += nrest[q];
        if (nrest_total > 0x7fffffffll) return ERROR_INPUT_PAR;
        ria[(size_t)p + 1] = (int)nrest_total;
    }
    lap("offsets");
    // ---- fill: chunk descriptors, slots, ghost lists, the rest, the per-item records
    H.strips.assign((size_t)nstrips, FlowStrip{});
    std::vector<FlowStrip>& strips = H.strips;
    H.chunks.alloc(4 * (size_t)std::max(nchunk, 1));
    Buf<int>& chunks = H.chunks;
    H.cstrip.alloc((size_t)std::max(nchunk, 1));
    Buf<int>& cstrip = H.cstrip;
    Buf<int> clev((size_t)std::max(nchunk, 1));
    H.slots.alloc((size_t)std::max<long long>(slot_bytes, 16));
    Buf<unsigned char>& slots = H.slots;
    H.gpos.alloc((size_t)std::max<long long>(nghost, 1));
    Buf<int>& gpos = H.gpos;
    H.rja.alloc((size_t)std::max<long long>(nrest_total, 1)); H.tr.alloc(2 * (size_t)std::max(npos, 1));
    Buf<int>&rja = H.rja, &tr = H.tr;
    H.rval.alloc((size_t)std::max<long long>(nrest_total, 1)); H.dr.alloc(2 * (size_t)std::max(npos, 1));
    Buf<double>&rval = H.rval, &dr = H.dr;
    int pfmax = 1, maxent = 0, bad = 0;
#pragma omp parallel reduction(max : pfmax, maxent) reduction(+ : bad)
    {
        // ghost index of an earlier position: open addressing, emptied per strip by a stamp
        constexpr int HB = 1 << 16;   // (2 x FLOW_LDS_ENT rounded up: at most FLOW_LDS_ENT distinct keys)
        std::vector<int> hkey((size_t)HB, -1), hval((size_t)HB, 0), hstamp((size_t)HB, -1);
        std::vector<std::pair<int, double>> low, ent;
#pragma omp for schedule(dynamic, 1)
        for (int s = 0; s < nstrips; ++s) {
            const int q0 = sq0[s], q1 = sq0[s + 1], p0 = q0 + voff[q0], p1 = q1 + voff[q1];
            FlowStrip& F = strips[(size_t)s];
            F.slot0 = sbytes[(size_t)s]; F.row0 = p0; F.nrows = p1 - p0; F.chunk0 = schunks[(size_t)s]; F.nchunk = schunks[(size_t)s + 1] - schunks[(size_t)s];
            F.ghost0 = sghost[(size_t)s]; F.nghost = sng[(size_t)s];
            maxent = std::max(maxent, F.nrows + F.nghost);
            const int zero_idx = F.nrows + F.nghost;
            int ng = 0;
            auto lds_index = [&](int p) -> int {   // position of an operand -> LDS index of this strip
                if (p >= p0) return p - p0;
                unsigned h = ((unsigned)p * 2654435761u) >> 16;
                for (;; h = (h + 1) & (HB - 1)) {
                    if (hstamp[h] != s) { hstamp[h] = s; hkey[h] = p; hval[h] = ng; gpos[(size_t)F.ghost0 + ng] = p; return F.nrows + ng++; }
                    if (hkey[h] == p) return F.nrows + hval[h];
                }
            };
            unsigned char* sb = slots.data() + F.slot0;
            long long off = 0;
            int ck = F.chunk0;
            for (int p = p0; p < p1; ++ck) {
                const int l = item_class(itq[p], itk[p]);
                int e = p, mx = 0;
                while (e < p1 && e - p < rpw && item_class(itq[e], itk[e]) == l) { mx = std::max(mx, item_count(itq[e], itk[e])); ++e; }
                const int pf = rounds_of(mx), nr = e - p, nl = nr * L;
                pfmax = std::max(pfmax, pf);
                int wown = -1, wghost = -1;   // the operand expected last: the chunk's highest own row, else its latest ghost
                cstrip[(size_t)ck] = s; clev[(size_t)ck] = l;
                unsigned short* cols = reinterpret_cast<unsigned short*>(sb + off);
                double* vals = reinterpret_cast<double*>(sb + off + 16ll * nl);
                for (int t = 0; t < nl * 8; ++t) cols[t] = (unsigned short)zero_idx;
                const int g0 = (PFS - pf) / 2;   // value planes in front of g0 are not stored
                for (long long t = 0; t < 2ll * nl * (PFS / 2 - g0); ++t) vals[t] = 0.0;
                for (int pp = p; pp < e; ++pp) {
                    const int q = itq[pp], k = itk[pp], i = seq[q], V = voff[(size_t)q + 1] - voff[q];
                    // the item's entries as (position, value), oldest first.  A virtual row: its slice of the row's oldest entries.  The row
                    // itself: its virtual rows (coefficient 1: their values are sums of products), then its newest entries.
                    ent.clear();
                    double dg = 0.0;
                    if (k == V) {   // the row: also its rest (everything that reads old values) and its diagonal
                        size_t kr = (size_t)ria[pp];
                        low.clear();
                        for (int kk = A.ia[i]; kk < A.ia[i + 1]; ++kk) {
                            const int j = A.ja[kk];
                            if (j == i) { dg = A.val[kk]; continue; }   // the last diagonal hit, as the reference's loop leaves it
                            const int pj = j < n ? pos[j] : -1;
                            if ((unsigned)pj < (unsigned)q) low.emplace_back(pj, A.val[kk]);
                            else { rja[kr] = j; rval[kr] = A.val[kk]; ++kr; }
                        }
                        std::stable_sort(low.begin(), low.end(), [&](const std::pair<int, double>& x, const std::pair<int, double>& y) {   // by (class, sequence): the same order however the strips are cut
                            return lev[x.first] != lev[y.first] ? lev[x.first] < lev[y.first] : x.first < y.first; });
                        for (int v = 0; v < V; ++v) ent.emplace_back(vpos[(size_t)voff[q] + v], 1.0);
                        const int first = V ? (int)low.size() - (CAP - V) : 0;
                        for (int en = first; en < (int)low.size(); ++en) ent.emplace_back(newpos[low[(size_t)en].first], low[(size_t)en].second);
                    } else {
                        sorted_lower(q, low);
                        for (int en = k * CAP; en < k * CAP + vcnt[(size_t)voff[q] + k]; ++en) ent.emplace_back(newpos[low[(size_t)en].first], low[(size_t)en].second);
                    }
                    // the last KT entries: the spine (last lane, last rounds).  The others right-aligned in the rounds in front of the
                    // spine: the LAST L of them fill the last of those rounds
                    const int nlo = (int)ent.size(), ksp = std::min(KT, nlo), nb = nlo - ksp, shift = (PFS - KT) * L - nb;
                    if (nb > (PFS - KT) * L) { ++bad; continue; }
                    for (int en = 0; en < nlo; ++en) {
                        const int c = lds_index(ent[(size_t)en].first);
                        if (c < F.nrows) wown = std::max(wown, c); else wghost = std::max(wghost, c);
                        int qe, lane;   // round, lane of the chunk
                        if (en >= nb) { qe = PFS - (nlo - en); lane = (pp - p) * L + L - 1; }
                        else { const int e2 = en + shift; qe = e2 / L; lane = (pp - p) * L + e2 % L; }
                        cols[lane * 8 + qe] = (unsigned short)c;
                        vals[(size_t)(qe / 2 - g0) * 2 * nl + (size_t)lane * 2 + (qe & 1)] = ent[(size_t)en].second;
                    }
                    if (k == V) {
                        const bool alone = !(std::fabs(dg) > SMALLREAL);
                        dr[2 * (size_t)pp] = dg; dr[2 * (size_t)pp + 1] = alone ? 0.0 : 1.0 / dg;
                        tr[2 * (size_t)pp] = alone ? (int)0x80000000 : 0; tr[2 * (size_t)pp + 1] = i;
                    } else {   // a virtual row: value = the sum of its products; no diagonal, no right-hand side, no row of u
                        dr[2 * (size_t)pp] = 1.0; dr[2 * (size_t)pp + 1] = 1.0;
                        tr[2 * (size_t)pp] = FLOW_VIRTUAL; tr[2 * (size_t)pp + 1] = -1;
                    }
                }
                { int* cd = &chunks[4 * (size_t)ck]; cd[0] = (p - p0) | (nr << 16) | (pf << 24); cd[1] = (int)(off / 16); cd[2] = wown >= 0 ? wown : wghost >= 0 ? wghost : zero_idx; cd[3] = 0; }
                off += 16ll * nl * (1 + PFS / 2 - g0);
                p = e;
            }
            if (ng != F.nghost || off != sbytes[(size_t)s + 1] - sbytes[(size_t)s] || off > 0x7fffffffll) ++bad;
        }
    }
    lap("fill");
    if (bad) { std::fprintf(stderr, "### ERROR: fasp_hip: inconsistent strip bookkeeping in the sweep schedule\n"); return ERROR_MISC; }
    // chunks by dependency class (k_tri_level): a counting sort over the doubled classes
    const int ncls = 2 * nlev + 2;
    H.cptr.assign((size_t)ncls + 1, 0);
    for (int c = 0; c < nchunk; ++c) H.cptr[(size_t)clev[(size_t)c] + 1]++;
    for (int l = 0; l < ncls; ++l) H.cptr[(size_t)l + 1] += H.cptr[(size_t)l];
    H.lchunks.alloc((size_t)std::max(nchunk, 1));
    Buf<int>& lchunks = H.lchunks;
    {
        std::vector<int> cur(H.cptr.begin(), H.cptr.end() - 1);
        for (int c = 0; c < nchunk; ++c) lchunks[(size_t)cur[(size_t)clev[(size_t)c]]++] = c;
    }
    lap("chunks by class");
    H.ns = npos; H.nrows = ns; H.nvirt = nvirt; H.L = L; H.nolower = lower_total == 0; H.pfs = PFS; H.kt = KT; (void)pfmax; H.nstrips = nstrips; H.nchunk = nchunk; H.maxent = maxent;
    H.nghost = nghost; H.slot_bytes = slot_bytes; H.nrest = nrest_total; H.flow_ok = flow_ok; H.nclasses = nlev;
    const double avg_rest = ns > 0 ? (double)nrest_total / ns : 0.0;
    H.LR = 1;
    while (H.LR < 64 && 4 * H.LR < avg_rest) H.LR *= 2;
    return FASP_SUCCESS;
}

}  // namespace fasp

// Host-side check of a schedule (tests/test_seq_schedule.py, no GPU): build it, then walk it the way the kernels do -- strips in
// ticket order, chunks in order, every operand through its LDS index (own row, ghost list or the constant) -- for one
// Gauss-Seidel sweep, and compare with the plain sequential sweep over the same rows.  Returns the largest difference relative
// to the largest entry (< 0: error).  Also checks what the kernels rely on: every operand of a chunk is produced by an earlier
// chunk of the strip or by an earlier strip; slots of unused rounds point at the constant.
extern "C" double fasp_hip_seq_schedule_selftest(const dCSRmat* Av, const int* seq, int ns, int strip_kb, int lanes, int spine, int* info)
{
    using namespace fasp;
    if (!Av || !seq || ns < 0) return -1.0;
    HostCSR A;
    A.row = Av->row; A.col = Av->col; A.nnz = Av->nnz;
    A.ia.view(Av->IA, (size_t)Av->row + 1); A.ja.view(Av->JA, (size_t)std::max(Av->nnz, 1)); A.val.view(Av->val, (size_t)std::max(Av->nnz, 1));
    SplitHost H;
    const int st = build_split_host(A, seq, ns, strip_kb, lanes, false, H, 0, spine < 0 ? 1 : spine);
    if (st != FASP_SUCCESS) return st == 1 ? -2.0 : -3.0;
    if (info) { info[0] = H.L; info[1] = H.pfs; info[2] = H.kt; info[3] = H.nvirt; info[4] = H.nstrips; info[5] = H.nchunk; }
    const int n = std::max(A.row, A.col), L = H.L, PF = H.pfs, KT = H.kt;   // (a rank's local rows of a partitioned level: columns beyond the rows are ghosts, never swept)
    std::vector<double> u((size_t)n), b((size_t)n), uref;
    for (int i = 0; i < n; ++i) { u[(size_t)i] = std::sin(0.37 * i) + 0.1; b[(size_t)i] = std::cos(0.11 * i); }
    uref = u;
    for (int q = 0; q < ns; ++q) {   // the reference's sweep: t = b_i - sum_{j != i} a_ij u_j, u_i = t / a_ii
        const int i = seq[q];
        double t = b[(size_t)i], d = 0.0;
        for (int k = A.ia[i]; k < A.ia[i + 1]; ++k) { if (A.ja[k] == i) d = A.val[k]; else t -= A.val[k] * uref[(size_t)A.ja[k]]; }
        if (std::fabs(d) > SMALLREAL) uref[(size_t)i] = t / d;
    }
    const int np = H.ns;   // positions: the rows of the sweep and the virtual rows
    if (H.nrows != ns || np != ns + H.nvirt) return -6.0;
    std::vector<double> W((size_t)std::max(np, 1)), rec((size_t)std::max(np, 1));
    std::vector<char> done((size_t)std::max(np, 1), 0);
    int nreal = 0;
    for (int p = 0; p < np; ++p) {   // pass (1)
        const int row = H.tr[2 * (size_t)p + 1];
        if ((H.tr[2 * (size_t)p] & FLOW_VIRTUAL) != 0 && H.tr[2 * (size_t)p] >= 0) { if (row != -1 || H.ria[p] != H.ria[p + 1]) return -6.0; rec[(size_t)p] = 0.0; continue; }
        ++nreal;
        double s = 0.0;
        for (int k = H.ria[p]; k < H.ria[p + 1]; ++k) s += H.rval[k] * u[(size_t)H.rja[k]];
        rec[(size_t)p] = b[(size_t)row] - s;
    }
    if (nreal != ns) return -6.0;
    if (H.nolower) { for (int p = 0; p < np; ++p) { const double d = H.dr[2 * (size_t)p]; if (H.tr[2 * (size_t)p] >= 0) u[(size_t)H.tr[2 * (size_t)p + 1]] = rec[(size_t)p] / d; } }
    else
    for (int s = 0; s < H.nstrips; ++s) {
        const FlowStrip& F = H.strips[(size_t)s];
        if (F.nrows + F.nghost + 1 > FLOW_LDS_ENT + 1) return -4.0;
        auto operand = [&](int c, bool& ok) -> double {
            if (c < F.nrows) { if (!done[(size_t)(F.row0 + c)]) ok = false; return W[(size_t)(F.row0 + c)]; }
            if (c < F.nrows + F.nghost) { const int gp = H.gpos[(size_t)F.ghost0 + (size_t)(c - F.nrows)]; if (gp >= F.row0 || !done[(size_t)gp]) ok = false; return W[(size_t)gp]; }
            if (c != F.nrows + F.nghost) ok = false;
            return 0.0;
        };
        for (int ck = F.chunk0; ck < F.chunk0 + F.nchunk; ++ck) {
            const int* cd = &H.chunks[4 * (size_t)ck];
            const int lo = cd[0] & 0xffff, nr = (cd[0] >> 16) & 0xff, pf = (cd[0] >> 24) & 0xf, nl = nr * L, g0 = (PF - pf) / 2;
            const unsigned char* sb = H.slots.data() + F.slot0 + 16ll * cd[1];
            const unsigned short* cols = reinterpret_cast<const unsigned short*>(sb);
            const double* vals = reinterpret_cast<const double*>(sb + 16ll * nl);
            std::vector<double> un((size_t)nr);
            for (int r = 0; r < nr; ++r) {
                const int p = F.row0 + lo + r;
                bool ok = true;
                double tot = 0.0;
                for (int sl = 0; sl < L; ++sl) {
                    const int lane = r * L + sl;
                    double sacc = 0.0;
                    for (int q = 0; q < PF - KT; ++q) {
                        const int c = cols[lane * 8 + q];
                        const double v = q / 2 >= g0 ? vals[(size_t)(q / 2 - g0) * 2 * nl + (size_t)lane * 2 + (q & 1)] : 0.0;
                        if (q < PF - pf && c != F.nrows + F.nghost) ok = false;
                        sacc += v * operand(c, ok);
                    }
                    for (int q = PF - KT; q < PF; ++q)   // spine rounds: the last lane's only
                        if (sl != L - 1 && cols[lane * 8 + q] != F.nrows + F.nghost) ok = false;
                    tot += sacc;
                }
                double T = rec[(size_t)p] - tot;
                for (int q = PF - KT; q < PF; ++q) {   // the spine, in order
                    const int lane = r * L + L - 1, c = cols[lane * 8 + q];
                    const double v = q / 2 >= g0 ? vals[(size_t)(q / 2 - g0) * 2 * nl + (size_t)lane * 2 + (q & 1)] : 0.0;
                    T -= v * operand(c, ok);
                }
                if (!ok) return -5.0;
                const double d = H.dr[2 * (size_t)p];
                const int flags = H.tr[2 * (size_t)p];
                un[(size_t)r] = flags < 0 ? u[(size_t)H.tr[2 * (size_t)p + 1]] : (flags & FLOW_VIRTUAL) ? -T : T / d;
            }
            for (int r = 0; r < nr; ++r) {
                const int p = F.row0 + lo + r;
                W[(size_t)p] = un[(size_t)r]; done[(size_t)p] = 1;
                if (H.tr[2 * (size_t)p + 1] >= 0) u[(size_t)H.tr[2 * (size_t)p + 1]] = un[(size_t)r];
            }
        }
    }
    double diff = 0.0, big = 0.0;
    for (int i = 0; i < n; ++i) { diff = std::max(diff, std::fabs(u[(size_t)i] - uref[(size_t)i])); big = std::max(big, std::fabs(uref[(size_t)i])); }
    return big > 0.0 ? diff / big : diff;
}

// Host-side check of a CHAIN schedule (tests/test_seq_schedule.py, no GPU): build it (chain form wherever it applies), then walk it
// the way k_tri_chain_ref does -- block after block, tier 2 and tier 1 as chains of fused multiply-adds in step order, the band's
// 64 steps with two accumulators -- for one sweep with update formula `form` (0: t * (1 / a_ii), 1: t / a_ii, 2: SOR with w), and
// compare with the plain sequential sweep over the same rows.  Returns the largest difference relative to the largest entry
// (< 0: error; -2: the form does not apply).  Also checks what the kernels rely on: tier-1 columns lie in the n1b blocks in front
// of the band (and their ring indices are unambiguous), tier-2 columns in front of those, padding entries point at the constants,
// the band planes hold zeros where the kernels expect no-ops.  out_u (optional, length max(row, col)): the swept vector.
extern "C" double fasp_hip_seq_chain_selftest(const dCSRmat* Av, const int* seq, int ns, int n1_blocks, int form, double w, int* info, double* out_u)
{
    using namespace fasp;
    if (!Av || !seq || ns < 0) return -1.0;
    HostCSR A;
    A.row = Av->row; A.col = Av->col; A.nnz = Av->nnz;
    A.ia.view(Av->IA, (size_t)Av->row + 1); A.ja.view(Av->JA, (size_t)std::max(Av->nnz, 1)); A.val.view(Av->val, (size_t)std::max(Av->nnz, 1));
    SplitHost H;
    const int st = build_split_host(A, seq, ns, 512, 0, false, H, 0, 1, 2, n1_blocks);
    if (st != FASP_SUCCESS) return st == 1 ? -2.0 : -3.0;
    if (!H.chain) return -2.0;
    const ChainHost& C = H.C;
    if (info) { info[0] = C.nb; info[1] = C.n1b; info[2] = C.rx; info[3] = C.rg; info[4] = (int)C.t1_steps; info[5] = (int)C.t2_steps; info[6] = (int)C.nband; info[7] = (int)C.nt1; info[8] = (int)C.nt2; info[9] = H.nclasses; }
    const int n = std::max(A.row, A.col), nb = C.nb, npad = C.npad;
    std::vector<double> u((size_t)n), b((size_t)n), uref;
    for (int i = 0; i < n; ++i) { u[(size_t)i] = std::sin(0.37 * i) + 0.1; b[(size_t)i] = std::cos(0.11 * i); }
    uref = u;
    for (int q = 0; q < ns; ++q) {   // the reference's sweep
        const int i = seq[q];
        double t = b[(size_t)i], d = 0.0;
        for (int k = A.ia[i]; k < A.ia[i + 1]; ++k) { if (A.ja[k] == i) d = A.val[k]; else t -= A.val[k] * uref[(size_t)A.ja[k]]; }
        if (std::fabs(d) > SMALLREAL) uref[(size_t)i] = form == 0 ? t * (1.0 / d) : form == 1 ? t / d : w * (t / d) + (1 - w) * uref[(size_t)i];
    }
    if (H.nrows != ns || H.ns != npad || npad != nb * 64) return -6.0;
    std::vector<double> W((size_t)npad + 1, 0.0), T((size_t)npad, 0.0), uo((size_t)npad, 0.0);
    for (int p = 0; p < npad; ++p) {   // pass (1)
        const int row = H.tr[2 * (size_t)p + 1];
        if (p >= ns) { if (row != -1 || H.ria[p] != H.ria[p + 1]) return -6.0; continue; }
        if (row != seq[p]) return -6.0;
        double s = 0.0;
        for (int k = H.ria[p]; k < H.ria[p + 1]; ++k) s += H.rval[k] * u[(size_t)H.rja[k]];
        T[(size_t)p] = b[(size_t)row] - s; uo[(size_t)p] = u[(size_t)row];
    }
    auto update = [&](double t, double d, double rd, double ku) {
        auto tdiv = [&]() { const double q = t * rd; const double r = std::fma(-d, q, t); return std::fma(r, rd, q); };
        return form == 0 ? t * rd : form == 1 ? tdiv() : w * tdiv() + ku;
    };
    std::vector<double> accB(64, 0.0), accA(64), x(64);
    for (int K = 0; K < nb; ++K) {
        const ChainBlk B = C.blk[(size_t)K];
        if ((B.t1_n & 7) || (B.t2_n & 7)) return -7.0;
        for (int j = 0; j < 64; ++j) {
            const int p = K * 64 + j;
            double g2 = T[(size_t)p];
            for (int s = 0; s < B.t2_n; ++s) {
                const size_t e = ((size_t)B.t2_off + s) * 64 + j;
                const int qp = C.t2c[(((size_t)(B.t2_off + s) >> 3) * 64 + (size_t)j) * 8 + (s & 7)];
                if (qp == npad) { if (C.t2v[e] != 0.0) return -7.0; }
                else if (qp >= (K - 1 - C.n1b) * 64) return -7.0;   // tier 2 lies in front of tier 1's window
                if (qp != npad && (qp >> 6) > C.t2need[(size_t)(B.t2_off + s) / 8]) return -9.0;   // the gate of the group covers every entry
                g2 = std::fma(-C.t2v[e], W[(size_t)qp], g2);
            }
            double s1 = 0.0;
            const int base = std::max(0, (K - 1 - C.n1b) * 64);
            for (int s = 0; s < B.t1_n; ++s) {
                const size_t e = ((size_t)B.t1_off + s) * 64 + j;
                const int r = C.t1c[(((size_t)(B.t1_off + s) >> 3) * 64 + (size_t)j) * 8 + (s & 7)];
                int qp;
                if (r >= C.rx) { if (r != C.rx || C.t1v[e] != 0.0) return -7.0; qp = npad; }
                else { qp = base + ((r - base % C.rx) + C.rx) % C.rx; if (qp >= (K - 1) * 64 || qp < base) return -7.0; }
                if (qp != npad && (qp >> 6) > C.t1need[(size_t)(B.t1_off + s) / 8]) return -9.0;
                s1 = std::fma(-C.t1v[e], W[(size_t)qp], s1);
            }
            accA[(size_t)j] = (g2 + s1) + accB[(size_t)j];
            accB[(size_t)j] = 0.0;
        }
        const double* bd = C.band.data() + (size_t)K * 64 * 128;
        for (int c = 0; c < 64; ++c) {
            const int pc = K * 64 + c;
            const double ku = form == 2 ? (1 - w) * uo[(size_t)pc] : 0.0;
            const double xc = update(accA[(size_t)c], C.drd[2 * (size_t)pc], C.drd[2 * (size_t)pc + 1], ku);
            x[(size_t)c] = xc;
            for (int j = 0; j < 64; ++j) {
                const double ta = bd[((size_t)c * 64 + j) * 2], tb = bd[((size_t)c * 64 + j) * 2 + 1];
                if (j <= c && ta != 0.0) return -8.0;                        // accA_j keeps t_j once it is final
                if ((K == nb - 1 || (K + 1) * 64 + j >= ns) && tb != 0.0) return -8.0;
                if (j > c) accA[(size_t)j] = std::fma(-ta, xc, accA[(size_t)j]);
                accB[(size_t)j] = std::fma(-tb, xc, accB[(size_t)j]);
            }
        }
        for (int c = 0; c < 64; ++c) {
            const int pc = K * 64 + c;
            W[(size_t)pc] = x[(size_t)c];
            if (H.tr[2 * (size_t)pc + 1] >= 0) u[(size_t)H.tr[2 * (size_t)pc + 1]] = x[(size_t)c];
        }
    }
    for (size_t t = (size_t)nb * 64 * 128; t < C.band.n; ++t) if (C.band[t] != 0.0) return -8.0;   // the steps the chain wave fetches behind the last block
    double diff = 0.0, big = 0.0;
    for (int i = 0; i < n; ++i) { diff = std::max(diff, std::fabs(u[(size_t)i] - uref[(size_t)i])); big = std::max(big, std::fabs(uref[(size_t)i])); }
    if (out_u) for (int i = 0; i < n; ++i) out_u[i] = u[(size_t)i];
    return big > 0.0 ? diff / big : diff;
}
