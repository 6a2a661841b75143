// smoothers.hip.h -- smoothers on a resident level: Jacobi / L1 / polynomial kernels, level-scheduled sequential sweeps.
// Part of the single translation unit solver.hip (included there, in this order; not a stand-alone header).

// ---------------------------------------------------------------------------
// smoothers on the resident level (PreMGSmoother.inl:49 / :155).  Jacobi and L1-diag
// are order independent, so pre (ascending) and post (descending) sweeps coincide.
// ---------------------------------------------------------------------------
static void materialise_zero(DevLevel& D)
{
    if (D.x_zero) {
        (void)hipMemsetAsync(D.x, 0, sizeof(double) * D.nvec, g_ctx.stream);
        D.x_zero = false;
    }
}

// Level schedule of one sequential sweep over the rows `seq` (in sweep order) of the host
// matrix A: level(i) = 1 + max level of the rows coupled to i (pattern of A and of A^T) that
// come earlier in the sweep.  Rows outside the sweep are not updated and impose nothing.
//
// multicolor == true (fasp_hip_tune("gs_multicolor", 1); NOT the reference's iteration, see seq_sweep): the "levels"
// are the colour classes of a greedy colouring of the swept rows (ascending row order, smallest colour no coupled row
// has), visited in ascending colour order by an ascending sweep and in descending colour order by a descending one.
// Rows of one colour are not coupled, so a class is one launch whatever its size: 2 launches per sweep on the 7-point
// level 0 instead of 3n - 2 dependency levels.
static int build_schedule(const HostCSR& A, const std::vector<int>& seq, DevLevel::Sched& S, bool multicolor = false)
{
    const int n = A.row;
    std::vector<int> pos(n, -1), lev(n, 0);
    for (int q = 0; q < (int)seq.size(); ++q) pos[seq[q]] = q;
    // transpose pattern for the anti-dependencies of structurally unsymmetric matrices
    std::vector<int> tia(n + 2, 0), tja(A.nnz);
    for (int k = 0; k < A.nnz; ++k) if (A.ja[k] < n) tia[A.ja[k] + 2]++;
    for (int i = 2; i <= n + 1; ++i) tia[i] += tia[i - 1];
    for (int i = 0; i < n; ++i)
        for (int k = A.ia[i]; k < A.ia[i + 1]; ++k)
            if (A.ja[k] < n) tja[tia[A.ja[k] + 1]++] = i;
    int nlev = 0;
    if (multicolor) {
        std::vector<int> rows(seq);
        std::sort(rows.begin(), rows.end());
        std::vector<int> color(n, -1), mark(rows.size() + 2, -1);
        for (int i : rows) {
            auto see = [&](int j) { if (j != i && j < n && color[j] >= 0) mark[color[j]] = i; };
            for (int k = A.ia[i]; k < A.ia[i + 1]; ++k) see(A.ja[k]);
            for (int k = tia[i]; k < tia[i + 1]; ++k) see(tja[k]);
            int c = 0;
            while (mark[c] == i) ++c;
            color[i] = c;
            nlev = std::max(nlev, c + 1);
        }
        const bool descending = seq.size() > 1 && seq.front() > seq.back();
        for (int i : rows) lev[i] = descending ? nlev - color[i] : color[i] + 1;
    }
    for (int q = 0; !multicolor && q < (int)seq.size(); ++q) {
        const int i = seq[q];
        int l = 0;
        for (int k = A.ia[i]; k < A.ia[i + 1]; ++k) {
            const int j = A.ja[k];
            if (j != i && j < n && pos[j] >= 0 && pos[j] < q) l = std::max(l, lev[j]);
        }
        for (int k = tia[i]; k < tia[i + 1]; ++k) {
            const int j = tja[k];
            if (j != i && pos[j] >= 0 && pos[j] < q) l = std::max(l, lev[j]);
        }
        lev[i] = l + 1;
        nlev = std::max(nlev, l + 1);
    }
    S.ptr.assign(nlev + 1, 0);
    for (int i : seq) S.ptr[lev[i]]++;
    for (int l = 0; l < nlev; ++l) S.ptr[l + 1] += S.ptr[l];
    std::vector<int> cur(S.ptr.begin(), S.ptr.end() - 1), order(seq.size());
    for (int i : seq) order[cur[lev[i] - 1]++] = i;
    S.release();
    HIPCK(hipMalloc(&S.d_order, sizeof(int) * std::max<size_t>(order.size(), 1)));
    if (!order.empty()) HIPCK(hipMemcpy(S.d_order, order.data(), sizeof(int) * order.size(), hipMemcpyHostToDevice));
    S.built = true;
    S.multicolor = multicolor;
    return FASP_SUCCESS;
}

// Error word of the dataflow form of the triangular solve (a waiter that spun for two seconds: seq_split.hip.h).  It is copied
// to pinned host memory behind every launch and looked at -- without waiting -- before the next sweep of that form and where
// a solve synchronises anyway: the solve in which it happened fails loudly, the process goes on with one launch per
// dependency class (k_tri_level).
static unsigned* g_seq_herr = nullptr;
static bool      g_flow_disabled = false;
static bool seq_err_pending() { return g_seq_herr && *g_seq_herr != 0u; }
static void seq_err_watch(const unsigned* d_err)
{
    if (!g_seq_herr) {
        if (hipHostMalloc((void**)&g_seq_herr, 64, hipHostMallocDefault) != hipSuccess) { g_seq_herr = nullptr; return; }
        std::memset(g_seq_herr, 0, 64);
    }
    if (*g_seq_herr == 0u) (void)hipMemcpyAsync(g_seq_herr, d_err, sizeof(unsigned), hipMemcpyDeviceToHost, g_ctx.stream);
}
static int seq_err_check()   // after a stream synchronisation (or between sweeps: whatever has arrived)
{
    if (!seq_err_pending()) return FASP_SUCCESS;
    std::fprintf(stderr, "### ERROR: fasp_hip: the dataflow triangular solve of the sequential smoothers timed out; this solve fails, "
                         "later sweeps run as one launch per dependency class (fasp_hip_tune(\"seq_flow\", 1) re-enables the dataflow form)\n");
    g_flow_disabled = true;
    *g_seq_herr = 0u;   // (the device word is reset by pass (1) of the next sweep)
    return ERROR_MISC;
}

// ---------------------------------------------------------------------------
// The split form of a sequential sweep (seq_split.hip.h), built once per (level, sweep kind) on first use: the sweep
// sequence cut into strips, every strip's rows ordered by dependency class (TRUE dependencies only: row i after the
// coupled rows the sweep visits before it) and cut into chunks of 64 / L rows, the lower entries in slot storage with
// LDS indices as columns, the strip's ghost list, the rest as a CSR in position order.
// ---------------------------------------------------------------------------
constexpr int TRI_PF = 4;   // slot rounds the lanes-per-row choice aims at (TRI_PFMAX = 8 is what a chunk can store)
template <class T>
static int split_upload(DevLevel::Sched& S, T** dst, const T* v, size_t n)
{
    *dst = nullptr;
    HIPCK(hipMalloc((void**)dst, sizeof(T) * std::max<size_t>(n, 1)));
    S.owned.push_back(*dst);
    if (n) HIPCK(hipMemcpy(*dst, v, sizeof(T) * n, hipMemcpyHostToDevice));
    return FASP_SUCCESS;
}
template <class T>
static int split_upload(DevLevel::Sched& S, T** dst, const std::vector<T>& v) { return split_upload(S, dst, v.data(), v.size()); }

static int build_split(const HostCSR& A, const std::vector<int>& seq, DevLevel::Sched& S)
{
    HostThreads host_team;   // (bounded OpenMP team for the row-parallel loops below; the dependency pass itself is sequential)
    const int n = A.row, ns = (int)seq.size();
    Buf<int> pos((size_t)std::max(n, 1));
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) pos[i] = -1;
#pragma omp parallel for schedule(static)
    for (int q = 0; q < ns; ++q) pos[seq[q]] = q;
    // lanes per row from a sample-free pass over the row lengths would need the lower counts; they come out of the dependency pass
    // below, which therefore runs first with nothing but the matrix: class (dependency level) and number of lower entries per row
    Buf<int> lev((size_t)std::max(ns, 1)), nlow((size_t)std::max(ns, 1)), nrest((size_t)std::max(ns, 1));
    int nlev = ns > 0 ? 1 : 0;
    long long lower_total = 0;
    for (int q = 0; q < ns; ++q) {
        const int i = seq[q];
        int l = 0, c = 0, dg = 0;
        for (int k = A.ia[i]; k < A.ia[i + 1]; ++k) {
            const int j = A.ja[k];
            if (j == i) { ++dg; continue; }
            if (j < n) { const int pj = pos[j]; if ((unsigned)pj < (unsigned)q) { l = std::max(l, lev[pj]); ++c; } }
        }
        lev[q] = l + 1; nlow[q] = c; nrest[q] = A.ia[i + 1] - A.ia[i] - c - dg;
        nlev = std::max(nlev, l + 1);
        lower_total += c;
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
    while (L < 64 && (wide ? TRI_PFMAX : TRI_PF) * L < len90) L *= 2;
    if (g_tune.seq_lanes > 0) { L = 1; while (L < 64 && L < g_tune.seq_lanes) L *= 2; }
    const int rpw = 64 / L;   // rows per chunk (one wavefront)
    // ---- strips: contiguous ranges of the sweep sequence, closed when the slot bytes reach the target or the LDS is full
    // (own rows + distinct earlier rows read + the constant).  Sequential: one more pass over the lower entries.
    const long long target = std::max(16, g_tune.seq_strip_kb) * 1024ll;
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
            const int i = seq[q];
            const int q0 = sq0.back();
            const long long rb = 40 + 16ll * L * (1 + (std::min(TRI_PFMAX, (nlow[q] + L - 1) / L) + 1) / 2) + 12ll * std::max(0, nlow[q] - TRI_PFMAX * L);
            fresh.clear();
            for (int k = A.ia[i]; k < A.ia[i + 1]; ++k) {
                const int j = A.ja[k];
                if (j == i || j >= n) continue;
                const int pj = pos[j];
                if ((unsigned)pj < (unsigned)q0 && gmark[pj] != sid) { gmark[pj] = sid; fresh.push_back(pj); }
            }
            if (rows > 0 && (bytes + rb > target || rows + 1 + ng + (int)fresh.size() + 1 > FLOW_LDS_ENT || rows >= 0xffff)) {
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
            if (1 + ng + 1 > FLOW_LDS_ENT) flow_ok = false;   // one row that reads more than the LDS holds: no dataflow form for this sweep
            ++rows; bytes += rb;
        }
        if (ns > 0) { sng.push_back(ng); sq0.push_back(ns); }
    }
    if (!flow_ok) { S.flow_ok = false; return 1; }   // (the caller falls back to whole-row level scheduling: build_schedule + k_seq_level)
    const int nstrips = (int)sq0.size() - 1;
    // rounds per chunk of this schedule: four where no row needs more (then the kernels with room for four run it)
    int nlowmax = 0;
#pragma omp parallel for schedule(static) reduction(max : nlowmax)
    for (int q = 0; q < ns; ++q) nlowmax = std::max(nlowmax, nlow[q]);
    const int PFS = (nlowmax + L - 1) / L > 4 ? TRI_PFMAX : 4;
    // ---- per strip: rows by (class, sequence), chunks; positions = strip base + local index
    Buf<int> newpos((size_t)std::max(ns, 1)), seqof((size_t)std::max(ns, 1));   // position of sequence index q; sequence index at position p
    std::vector<int> schunks((size_t)nstrips + 1, 0);
    std::vector<long long> sbytes((size_t)nstrips + 1, 0);
#pragma omp parallel
    {
        std::vector<int> cnt;
#pragma omp for schedule(dynamic, 1)
        for (int s = 0; s < nstrips; ++s) {
            const int q0 = sq0[s], q1 = sq0[s + 1];
            int lmin = lev[q0], lmax = lev[q0];
            for (int q = q0; q < q1; ++q) { lmin = std::min(lmin, lev[q]); lmax = std::max(lmax, lev[q]); }
            cnt.assign((size_t)(lmax - lmin + 2), 0);
            for (int q = q0; q < q1; ++q) cnt[(size_t)(lev[q] - lmin + 1)]++;
            for (size_t l = 1; l < cnt.size(); ++l) cnt[l] += cnt[l - 1];
            for (int q = q0; q < q1; ++q) { const int p = q0 + cnt[(size_t)(lev[q] - lmin)]++; newpos[q] = p; seqof[p] = q; }
            // chunks: runs of one class, rpw rows at most
            int nch = 0;
            long long by = 0;
            for (int p = q0; p < q1;) {
                const int l = lev[seqof[p]];
                int e = p, mx = 0;
                while (e < q1 && e - p < rpw && lev[seqof[e]] == l) { mx = std::max(mx, nlow[seqof[e]]); ++e; }
                const int pf = std::max(1, std::min(TRI_PFMAX, (mx + L - 1) / L));
                by += 16ll * (e - p) * L * (1 + PFS / 2 - (PFS - pf) / 2);
                ++nch; p = e;
            }
            schunks[(size_t)s + 1] = nch; sbytes[(size_t)s + 1] = by;
        }
    }
    std::vector<int> sghost((size_t)nstrips + 1, 0);
    for (int s = 0; s < nstrips; ++s) { schunks[(size_t)s + 1] += schunks[(size_t)s]; sbytes[(size_t)s + 1] += sbytes[(size_t)s]; sghost[(size_t)s + 1] = sghost[(size_t)s] + sng[(size_t)s]; }
    const int nchunk = nstrips ? schunks[(size_t)nstrips] : 0;
    const long long slot_bytes = nstrips ? sbytes[(size_t)nstrips] : 0, nghost = nstrips ? sghost[(size_t)nstrips] : 0;
    // ---- tail and rest offsets by position
    Buf<int> tia((size_t)ns + 1), ria((size_t)ns + 1);
    tia[0] = 0; ria[0] = 0;
    long long ntail = 0, nrest_total = 0;
    for (int p = 0; p < ns; ++p) {
        const int q = seqof[p];
        ntail += std::max(0, nlow[q] - TRI_PFMAX * L); nrest_total += nrest[q];
        if (ntail > 0x7fffffffll || nrest_total > 0x7fffffffll) return ERROR_INPUT_PAR;
        tia[(size_t)p + 1] = (int)ntail; ria[(size_t)p + 1] = (int)nrest_total;
    }
    // ---- fill: chunk descriptors, slots, ghost lists, tails, the rest, the per-row records
    std::vector<FlowStrip> strips((size_t)nstrips);
    Buf<int4> chunks((size_t)std::max(nchunk, 1));
    Buf<int> cstrip((size_t)std::max(nchunk, 1)), clev((size_t)std::max(nchunk, 1));
    Buf<unsigned char> slots((size_t)std::max<long long>(slot_bytes, 16));
    Buf<int> gpos((size_t)std::max<long long>(nghost, 1));
    Buf<int> tja((size_t)std::max<long long>(ntail, 1)), rja((size_t)std::max<long long>(nrest_total, 1)), tr(2 * (size_t)std::max(ns, 1));
    Buf<double> tval((size_t)std::max<long long>(ntail, 1)), rval((size_t)std::max<long long>(nrest_total, 1)), dr(2 * (size_t)std::max(ns, 1));
    int pfmax = 1, maxent = 0, bad = 0;
#pragma omp parallel reduction(max : pfmax, maxent) reduction(+ : bad)
    {
        // ghost index of an earlier position: open addressing, emptied per strip by a stamp
        constexpr int HB = 1 << 16;   // (2 x FLOW_LDS_ENT rounded up: at most FLOW_LDS_ENT distinct keys)
        std::vector<int> hkey((size_t)HB, -1), hval((size_t)HB, 0), hstamp((size_t)HB, -1);
        std::vector<std::pair<int, double>> low;
#pragma omp for schedule(dynamic, 1)
        for (int s = 0; s < nstrips; ++s) {
            const int q0 = sq0[s], q1 = sq0[s + 1];
            FlowStrip& F = strips[(size_t)s];
            F.slot0 = sbytes[(size_t)s]; F.row0 = q0; F.nrows = q1 - q0; F.chunk0 = schunks[(size_t)s]; F.nchunk = schunks[(size_t)s + 1] - schunks[(size_t)s];
            F.ghost0 = sghost[(size_t)s]; F.nghost = sng[(size_t)s];
            maxent = std::max(maxent, F.nrows + F.nghost);
            const int zero_idx = F.nrows + F.nghost;
            int ng = 0;
            auto lds_index = [&](int p) -> int {   // position of a lower entry -> LDS index of this strip
                if (p >= q0) return p - q0;
                unsigned h = ((unsigned)p * 2654435761u) >> 16;
                for (;; h = (h + 1) & (HB - 1)) {
                    if (hstamp[h] != s) { hstamp[h] = s; hkey[h] = p; hval[h] = ng; gpos[(size_t)F.ghost0 + ng] = p; return F.nrows + ng++; }
                    if (hkey[h] == p) return F.nrows + hval[h];
                }
            };
            unsigned char* sb = slots.data() + F.slot0;
            long long off = 0;
            int ck = F.chunk0;
            for (int p = q0; p < q1; ++ck) {
                const int l = lev[seqof[p]];
                int e = p, mx = 0;
                while (e < q1 && e - p < rpw && lev[seqof[e]] == l) { mx = std::max(mx, nlow[seqof[e]]); ++e; }
                const int pf = std::max(1, std::min(TRI_PFMAX, (mx + L - 1) / L)), nr = e - p, nl = nr * L;
                pfmax = std::max(pfmax, pf);
                int wown = -1, wghost = -1;   // the operand expected last: the chunk's highest own row, else its latest ghost
                cstrip[(size_t)ck] = s; clev[(size_t)ck] = l;
                unsigned short* cols = reinterpret_cast<unsigned short*>(sb + off);
                double* vals = reinterpret_cast<double*>(sb + off + 16ll * nl);
                for (int t = 0; t < nl * 8; ++t) cols[t] = (unsigned short)zero_idx;
                const int g0 = (PFS - pf) / 2;   // value planes in front of g0 are not stored
                for (long long t = 0; t < 2ll * nl * (PFS / 2 - g0); ++t) vals[t] = 0.0;
                for (int pp = p; pp < e; ++pp) {
                    const int q = seqof[pp], i = seq[q];
                    size_t kt = (size_t)tia[pp], kr = (size_t)ria[pp];
                    double dg = 0.0;
                    low.clear();
                    for (int k = A.ia[i]; k < A.ia[i + 1]; ++k) {
                        const int j = A.ja[k];
                        if (j == i) { dg = A.val[k]; continue; }   // the last diagonal hit, as the reference's loop leaves it
                        const int pj = j < n ? pos[j] : -1;
                        if ((unsigned)pj < (unsigned)q) low.emplace_back(pj, A.val[k]);
                        else { rja[kr] = j; rval[kr] = A.val[k]; ++kr; }
                    }
                    // lower entries in the order of their dependency classes (= the order in which they become available): what a row
                    // still waits for sits in its last slots (k_tri_flow sums the complete rounds while it waits)
                    std::stable_sort(low.begin(), low.end(), [&](const std::pair<int, double>& x, const std::pair<int, double>& y) {   // by (class, sequence): the same order however the strips are cut
                        return lev[x.first] != lev[y.first] ? lev[x.first] < lev[y.first] : x.first < y.first; });
                    // right-aligned in the PFS rounds: the row's LAST L entries fill the last round; what does not fit (the oldest) is the tail
                    const int nlo = (int)low.size(), ntl = std::max(0, nlo - TRI_PFMAX * L), shift = PFS * L - (nlo - ntl);
                    for (int en = 0; en < nlo; ++en) {
                        const int c = lds_index(newpos[low[(size_t)en].first]);
                        if (c < F.nrows) wown = std::max(wown, c); else wghost = std::max(wghost, c);
                        if (en >= ntl) {
                            const int e2 = en - ntl + shift, qe = e2 / L, lane = (pp - p) * L + e2 % L;   // round, lane of the chunk
                            cols[lane * 8 + qe] = (unsigned short)c;
                            vals[(size_t)(qe / 2 - g0) * 2 * nl + (size_t)lane * 2 + (qe & 1)] = low[(size_t)en].second;
                        } else { tja[kt] = c; tval[kt] = low[(size_t)en].second; ++kt; }
                    }
                    const bool alone = !(std::fabs(dg) > SMALLREAL);
                    dr[2 * (size_t)pp] = dg; dr[2 * (size_t)pp + 1] = alone ? 0.0 : 1.0 / dg;
                    tr[2 * (size_t)pp] = (tia[(size_t)pp + 1] - tia[pp]) | (alone ? (int)0x80000000 : 0); tr[2 * (size_t)pp + 1] = i;
                }
                chunks[(size_t)ck] = make_int4((p - q0) | (nr << 16) | (pf << 24), (int)(off / 16), wown >= 0 ? wown : wghost >= 0 ? wghost : zero_idx, 0);
                off += 16ll * nl * (1 + PFS / 2 - g0);
                p = e;
            }
            if (ng != F.nghost || off != sbytes[(size_t)s + 1] - sbytes[(size_t)s] || off > 0x7fffffffll) ++bad;
        }
    }
    if (bad) { std::fprintf(stderr, "### ERROR: fasp_hip: inconsistent strip bookkeeping in the sweep schedule\n"); return ERROR_MISC; }
    // chunks by dependency class (k_tri_level): a counting sort
    S.cptr.assign((size_t)nlev + 1, 0);
    for (int c = 0; c < nchunk; ++c) S.cptr[(size_t)clev[(size_t)c]]++;
    for (int l = 0; l < nlev; ++l) S.cptr[(size_t)l + 1] += S.cptr[(size_t)l];
    Buf<int> lchunks((size_t)std::max(nchunk, 1));
    {
        std::vector<int> cur(S.cptr.begin(), S.cptr.end() - 1);
        for (int c = 0; c < nchunk; ++c) lchunks[(size_t)cur[(size_t)clev[(size_t)c] - 1]++] = c;
    }
    S.release();
    int st = FASP_SUCCESS;
    FlowStrip* d_strips = nullptr; int4* d_chunks = nullptr;
    if ((st = split_upload(S, &d_strips, strips)) < 0) return st;
    if ((st = split_upload(S, &d_chunks, chunks.data(), (size_t)nchunk)) < 0) return st;
    S.d_strips = d_strips; S.d_chunks = d_chunks;
    if ((st = split_upload(S, &S.d_slots, slots.data(), (size_t)slot_bytes)) < 0) return st;
    if ((st = split_upload(S, &S.d_gpos, gpos.data(), (size_t)nghost)) < 0) return st;
    if ((st = split_upload(S, &S.d_cstrip, cstrip.data(), (size_t)nchunk)) < 0) return st;
    if ((st = split_upload(S, &S.d_lchunks, lchunks.data(), (size_t)nchunk)) < 0) return st;
    if ((st = split_upload(S, &S.d_tia, tia.data(), (size_t)ns + 1)) < 0) return st;
    if ((st = split_upload(S, &S.d_tja, tja.data(), (size_t)ntail)) < 0) return st;
    if ((st = split_upload(S, &S.d_tval, tval.data(), (size_t)ntail)) < 0) return st;
    if ((st = split_upload(S, &S.d_ria, ria.data(), (size_t)ns + 1)) < 0) return st;
    if ((st = split_upload(S, &S.d_rja, rja.data(), (size_t)nrest_total)) < 0) return st;
    if ((st = split_upload(S, &S.d_rval, rval.data(), (size_t)nrest_total)) < 0) return st;
    if ((st = split_upload(S, &S.d_dr, dr.data(), 2 * (size_t)ns)) < 0) return st;
    if ((st = split_upload(S, &S.d_tr, tr.data(), 2 * (size_t)ns)) < 0) return st;
    HIPCK(hipMalloc((void**)&S.d_rec, sizeof(double) * 2 * (size_t)std::max(ns, 1))); S.owned.push_back(S.d_rec);
    HIPCK(hipMalloc((void**)&S.d_W, sizeof(double) * (size_t)std::max(ns, 1))); S.owned.push_back(S.d_W);
    HIPCK(hipMemset(S.d_W, 0, sizeof(double) * (size_t)std::max(ns, 1)));
    if ((st = split_upload(S, &S.d_prog, std::vector<unsigned>(64, 0u))) < 0) return st;   // [0] ticket counter, [1] error word
    S.ns = ns; S.L = L; S.nolower = lower_total == 0; S.ntail = ntail; S.pfmax = PFS; (void)pfmax; S.nstrips = nstrips; S.nchunk = nchunk; S.maxent = maxent;
    S.nghost = nghost; S.slot_bytes = slot_bytes; S.flow_ok = flow_ok;
    const double avg_rest = ns > 0 ? (double)nrest_total / ns : 0.0;
    S.LR = 1;
    while (S.LR < 64 && 4 * S.LR < avg_rest) S.LR *= 2;
    S.built = true;
    S.multicolor = false;
    return FASP_SUCCESS;
}

// one sequential sweep of schedule `kind` with update formula `form` (see tri_update)
static int seq_sweep(fasp_hip_amg* h, int level, int kind, int form, double w)
{
    DevLevel& D = h->L[level];
    DevLevel::Sched& S = D.sched[kind];
    // Multicolour mode -- a FLAGGED NON-PARITY mode for speed: the sweep visits the rows colour by colour instead of
    // in the reference's index order, which is a different (equally convergent, deterministic) Gauss-Seidel / SOR
    // iteration; iteration counts and residuals then differ from the reference's.  The default is the split sweep
    // (seq_split.hip.h), which reproduces the reference's sequential sweep.
    const bool multicolor = g_tune.gs_multicolor != 0;
    if (!S.built || S.multicolor != multicolor) {
        const HostCSR& A = h->H.L[level].A;
        const int n = A.row;
        std::vector<int> seq;
        seq.reserve(n);
        const int* cf = h->H.L[level].cfmark.n ? h->H.L[level].cfmark.data() : nullptr;
        switch (kind) {
            case 0: for (int i = 0; i < n; ++i) seq.push_back(i); break;
            case 1: for (int i = n - 1; i >= 0; --i) seq.push_back(i); break;
            case 2: for (int i = 0; i < n; ++i) if (cf && cf[i] == 1) seq.push_back(i); break;
            case 3: for (int i = 0; i < n; ++i) if (!cf || cf[i] != 1) seq.push_back(i); break;
            default: for (int i = n - 2; i >= 0; --i) seq.push_back(i); break;
        }
        const double t0 = wall_seconds();
        int st = multicolor ? build_schedule(A, seq, S, true) : build_split(A, seq, S);
        if (st < 0) return st;
        S.rowlevels = false;
        if (st == 1) {   // a row of this sweep reads more earlier rows than a strip's LDS holds: whole rows, one launch per dependency level
            if ((st = build_schedule(A, seq, S, false)) < 0) return st;
            S.rowlevels = true;
        }
        if (std::getenv("FASP_HIP_SETUP_TIMING")) {
            if (multicolor || S.rowlevels) std::printf("  [sweep schedule] level %d, sweep kind %d, %s: %d rows in %d classes\n", level, kind, multicolor ? "colours" : "whole-row dependency levels", (int)seq.size(), (int)S.ptr.size() - 1);
            else std::printf("  [sweep schedule] level %d, sweep kind %d: %d rows in %d dependency classes, %d strips (%lld ghosts, at most %d values in LDS), %d chunks, %d lanes per row, "
                             "%.1f slot bytes per row (%lld tail entries), rest pass %d lanes per row%s, built in %.3f s\n",
                             level, kind, S.ns, (int)S.cptr.size() - 1, S.nstrips, S.nghost, S.maxent + 1, S.nchunk, S.L, S.ns ? (double)S.slot_bytes / S.ns : 0.0, S.ntail, S.LR,
                             "", wall_seconds() - t0);
        }
    }
    materialise_zero(D);
    if (multicolor || S.rowlevels) {
        const int nlev = (int)S.ptr.size() - 1;
        // one launch per colour (or per dependency level of whole rows); lanes per row as the level's SpMV kernel
        const double avg_len = D.A.row > 0 ? (double)D.A.nnz / D.A.row : 0.0;
        const int L = g_tune.seq_lanes > 0 ? g_tune.seq_lanes : (avg_len >= 96.0 ? 64 : D.A.lanes);
        for (int l = 0; l < nlev; ++l) {
            const int lo = S.ptr[l], hi = S.ptr[l + 1];
            const int rpb = BLOCK / L;
            const int grid = std::max(1, std::min(MAXGRID, (hi - lo + rpb - 1) / rpb));
#define SEQ_LAUNCH(LL) hipLaunchKernelGGL((k_seq_level<LL>), dim3(grid), dim3(BLOCK), 0, g_ctx.stream, \
            (const int*)S.d_order, lo, hi, (const int*)D.A.ia, (const int*)D.A.ja, (const double*)D.A.val,    \
            (const double*)D.b, (const double*)D.diag, D.x, form, w)
            switch (L) {
                case 2: SEQ_LAUNCH(2); break;
                case 4: SEQ_LAUNCH(4); break;
                case 8: SEQ_LAUNCH(8); break;
                case 16: SEQ_LAUNCH(16); break;
                case 32: SEQ_LAUNCH(32); break;
                default: SEQ_LAUNCH(64); break;
            }
#undef SEQ_LAUNCH
        }
        return FASP_SUCCESS;
    }
    const int ns = S.ns;
    if (ns == 0) return FASP_SUCCESS;
    FlowArgs fa{};
    fa.strips = (const FlowStrip*)S.d_strips; fa.chunks = (const int4*)S.d_chunks; fa.slots = S.d_slots; fa.gpos = S.d_gpos; fa.cstrip = S.d_cstrip;
    fa.tia = S.d_tia; fa.tja = S.d_tja; fa.tval = S.d_tval; fa.rec = S.d_rec; fa.dr = S.d_dr; fa.tr = S.d_tr; fa.W = S.d_W; fa.u = D.x;
    fa.sync = S.d_prog; fa.nstrips = S.nstrips; fa.form = form; fa.w = w;
    // pass (1): everything that reads old values, all rows at once
    {
        const int rpb = BLOCK / S.LR;
        const int grid = std::max(1, std::min(MAXGRID, (ns + rpb - 1) / rpb));
#define REST_LAUNCH(LL) hipLaunchKernelGGL((k_split_rest<LL>), dim3(grid), dim3(BLOCK), 0, g_ctx.stream, ns, (const int*)S.d_tr, \
        (const int*)S.d_ria, (const int*)S.d_rja, (const double*)S.d_rval, (const double*)D.b, (const double*)D.x, S.d_rec, S.d_W, S.d_prog)
        switch (S.LR) {
            case 1: REST_LAUNCH(1); break;
            case 2: REST_LAUNCH(2); break;
            case 4: REST_LAUNCH(4); break;
            case 8: REST_LAUNCH(8); break;
            case 16: REST_LAUNCH(16); break;
            case 32: REST_LAUNCH(32); break;
            default: REST_LAUNCH(64); break;
        }
#undef REST_LAUNCH
    }
    const int sgrid = std::max(1, std::min(MAXGRID, (ns + BLOCK - 1) / BLOCK));
    if (S.nolower) {   // no row of the sweep reads another one's new value (the C rows / F rows of the 7-point level 0)
        hipLaunchKernelGGL(k_split_scatter, dim3(sgrid), dim3(BLOCK), 0, g_ctx.stream, ns, fa, 1);
        return FASP_SUCCESS;
    }
    // pass (2), the dataflow form: one launch, workgroups draw strips from the ticket counter (seq_split.hip.h)
    const int L = S.L;
    if (g_tune.seq_flow && S.flow_ok && !g_flow_disabled) {
        if (seq_err_check() < 0) return ERROR_MISC;   // an earlier sweep's time-out that has arrived meanwhile
        const size_t dyn = sizeof(double) * ((size_t)S.maxent + 1);
        const int by_lds = (int)((160 * 1024 - 64) / (dyn + 16));
#define FLOW_ONE(LL, PP, TT)                                                                                                \
        {                                                                                                                   \
            static bool attr_set = false;                                                                                   \
            if (!attr_set) {                                                                                                \
                HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tri_flow<LL, PP, TT>),                            \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * (FLOW_LDS_ENT + 1)))); \
                attr_set = true;                                                                                            \
            }                                                                                                               \
            const int nt = FlowGeom<PP>::NT, per_cu = std::max(1, std::min(2048 / nt, by_lds));                             \
            const int grid = std::max(1, std::min(S.nstrips, per_cu * g_ctx.num_cu));                                       \
            hipLaunchKernelGGL((k_tri_flow<LL, PP, TT>), dim3(grid), dim3(nt), dyn, g_ctx.stream, fa);                      \
        }
#define FLOW_LAUNCH(LL)                                                                                                     \
        if (S.ntail) FLOW_ONE(LL, TRI_PFMAX, true)                                                                          \
        else if (S.pfmax > 4) FLOW_ONE(LL, TRI_PFMAX, false)                                                                \
        else FLOW_ONE(LL, 4, false)
        switch (L) {
            case 1: FLOW_LAUNCH(1); break;
            case 2: FLOW_LAUNCH(2); break;
            case 4: FLOW_LAUNCH(4); break;
            case 8: FLOW_LAUNCH(8); break;
            case 16: FLOW_LAUNCH(16); break;
            case 32: FLOW_LAUNCH(32); break;
            default: FLOW_LAUNCH(64); break;
        }
#undef FLOW_LAUNCH
#undef FLOW_ONE
        seq_err_watch(S.d_prog + 1);
        return FASP_SUCCESS;
    }
    // the plain form: one launch per dependency class, one wavefront per chunk
    const int nlev = (int)S.cptr.size() - 1;
    for (int l = 0; l < nlev; ++l) {
        const int c0 = S.cptr[l], grid = S.cptr[l + 1] - c0;
        if (grid <= 0) continue;
#define TRIL_LAUNCH(LL) if (S.pfmax > 4) hipLaunchKernelGGL((k_tri_level<LL, TRI_PFMAX>), dim3(grid), dim3(64), 0, g_ctx.stream, fa, (const int*)S.d_lchunks, c0); \
                        else hipLaunchKernelGGL((k_tri_level<LL, 4>), dim3(grid), dim3(64), 0, g_ctx.stream, fa, (const int*)S.d_lchunks, c0)
        switch (L) {
            case 1: TRIL_LAUNCH(1); break;
            case 2: TRIL_LAUNCH(2); break;
            case 4: TRIL_LAUNCH(4); break;
            case 8: TRIL_LAUNCH(8); break;
            case 16: TRIL_LAUNCH(16); break;
            case 32: TRIL_LAUNCH(32); break;
            default: TRIL_LAUNCH(64); break;
        }
#undef TRIL_LAUNCH
    }
    hipLaunchKernelGGL(k_split_scatter, dim3(sgrid), dim3(BLOCK), 0, g_ctx.stream, ns, fa, 0);
    return FASP_SUCCESS;
}

// Smoother dispatch of PreMGSmoother.inl:49 (pre) / :155 (post).  Jacobi and L1-diag are
// order independent, so their pre (ascending) and post (descending) sweeps coincide; the
// Gauss-Seidel / SOR family runs as level-scheduled sequential sweeps.
// fasp_smoother_dcsr_poly (ItrSmootherCSRpoly.c:67): per sweep r = b - A u, then the recurrence of
// Rr (:551) -- ndeg SpMVs and elementwise steps -- and u += correction.  Order independent, so the
// level may be row-partitioned (every SpMV input gets its halo).  Dinv and the coefficients depend
// on the matrix only: formed once per level on the host exactly as the reference does per call.
static int cg_smooth(fasp_hip_amg* h, int level, int nsweeps);  // defined after the Krylov drivers

static int poly_smooth(fasp_hip_amg* h, int level, int ndeg, int nsweeps)
{
    DevLevel& D = h->L[level];
    const int n = D.A.row;
    DevLevel::Poly& Q = D.poly;
    hipStream_t s = g_ctx.stream;
    if (!Q.built) {
        // every rank holds the whole host hierarchy: local row i is global row r0 + i, and the
        // norm (a maximum over ALL rows of the level) needs no exchange
        const HostCSR& A = h->H.L[level].A;
        const int r0 = D.replicated ? 0 : D.row0;
        std::vector<double> dinv((size_t)std::max(n, 1));
        double norm = 0.0;
        for (int gi = 0; gi < A.row; ++gi) {  // Diaginv :392 (first hit) and DinvAnorminf :428
            int j = A.ia[gi];
            for (; j < A.ia[gi + 1]; ++j) if (A.ja[j] == gi) break;
            const double di = 1.0 / A.val[j];
            if (gi >= r0 && gi < r0 + n) dinv[(size_t)(gi - r0)] = di;
            double temp = 0.0;
            for (int q = A.ia[gi]; q < A.ia[gi + 1]; ++q) temp += std::fabs(A.val[q]);
            temp *= di;
            norm = std::max(norm, temp);
        }
        double mu0 = norm;
        mu0 = 1.0 / mu0;
        const double mu1 = 4.0 * mu0, smu0 = std::sqrt(mu0), smu1 = std::sqrt(mu1);
        Q.k[1] = (mu0 + mu1) / 2.0;
        Q.k[2] = (smu0 + smu1) * (smu0 + smu1) / 2.0;
        Q.k[3] = mu0 * mu1;
        Q.k[4] = 2.0 * Q.k[3] / Q.k[2];
        Q.k[5] = (mu1 - 2.0 * smu0 * smu1 + mu0) / (mu1 + 2.0 * smu0 * smu1 + mu0);
        HIPCK(hipMalloc(&Q.dinv, sizeof(double) * std::max(n, 1)));
        HIPCK(hipMemcpy(Q.dinv, dinv.data(), sizeof(double) * n, hipMemcpyHostToDevice));
        for (double*& q : Q.w) {
            if (alloc_vec(&q, (size_t)D.nvec) < 0) return ERROR_ALLOC_MEM;
        }
        Q.built = true;
    }
    double *r = Q.w[0], *rbar = Q.w[1], *v0 = Q.w[2], *v1 = Q.w[3], *vnew = Q.w[4];
    const int G = vec_grid(n);
    for (int it = 0; it < nsweeps; ++it) {
        if (D.x_zero) {  // u == 0: r = b exactly, no matrix pass
            HIPCK(hipMemcpyAsync(r, D.b, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
            materialise_zero(D);
        } else {
            if (halo_exchange(D, D.x) < 0) return ERROR_MISC;
            d_resid(D.A, D.x, D.b, r);
        }
        hipLaunchKernelGGL(k_poly_scale, dim3(G), dim3(BLOCK), 0, s, n, Q.dinv, r, rbar);
        if (halo_exchange(D, rbar) < 0) return ERROR_MISC;
        d_mxv(D.A, rbar, v1);
        hipLaunchKernelGGL(k_poly_start, dim3(G), dim3(BLOCK), 0, s, n, Q.k[1], Q.k[2], Q.k[3], Q.dinv, rbar, v0, v1);
        if (ndeg <= 1) HIPCK(hipMemsetAsync(vnew, 0, sizeof(double) * n, s));  // the reference's correction stays zero
        for (int j = 1; j < ndeg; ++j) {
            if (halo_exchange(D, v1) < 0) return ERROR_MISC;
            d_mxv(D.A, v1, rbar);
            hipLaunchKernelGGL(k_poly_step, dim3(G), dim3(BLOCK), 0, s, n, Q.k[4], Q.k[5], Q.dinv, r, rbar, v0, v1, vnew);
        }
        d_axpy(n, 1.0, vnew, D.x);
    }
    return FASP_SUCCESS;
}

static int smooth(fasp_hip_amg* h, int level, bool post, int smoother, int order, int nsweeps, double relax, int ndeg)
{
    DevLevel& D = h->L[level];
    const int n = D.A.row;
    if (smoother == SMOOTHER_POLY) return poly_smooth(h, level, ndeg, nsweeps);
    if (smoother == SMOOTHER_JACOBIF) {  // fasp_smoother_dcsr_jacobi_ff, ItrSmootherCSR.c:34
        const Buf<int>& cf = h->H.L[level].cfmark;
        if (!D.replicated || cf.n != (size_t)n) {
            std::printf("### ERROR: fasp_hip: Jacobi-F needs the C/F marker of a classical hierarchy (one GPU)\n");
            return ERROR_AMG_SMOOTH_TYPE;
        }
        if (!D.d_mark) {
            HIPCK(hipMalloc(&D.d_mark, sizeof(int) * std::max(n, 1)));
            HIPCK(hipMemcpy(D.d_mark, cf.data(), sizeof(int) * n, hipMemcpyHostToDevice));
        }
        materialise_zero(D);
        for (int s = 0; s < nsweeps; ++s) {
            CsrArgs a{};
            a.x = D.x; a.y = D.xo; a.b = D.b; a.omega = relax; a.diag = D.diag; a.mark = D.d_mark;
            launch_csr<OP_L1DIAG>(D.A, a);
            std::swap(D.x, D.xo);
        }
        return FASP_SUCCESS;
    }
    if (smoother == SMOOTHER_JACOBI || smoother == SMOOTHER_L1DIAG) {
        for (int s = 0; s < nsweeps; ++s) {
            if (D.x_zero && D.presmoothed && smoother == SMOOTHER_JACOBI) {   // this sweep was written with the rhs
                D.presmoothed = false; D.x_zero = false;
                continue;
            }
            if (D.x_zero) {
                // zero initial guess: t_i = b_i exactly, no matrix pass
                if (smoother == SMOOTHER_JACOBI)
                    hipLaunchKernelGGL(k_jacobi_zero, dim3(vec_grid(n)), dim3(BLOCK), 0, g_ctx.stream, n, relax,
                                       D.b, D.diag, D.x);
                else
                    hipLaunchKernelGGL(k_l1_zero, dim3(vec_grid(n)), dim3(BLOCK), 0, g_ctx.stream, n, D.b, D.l1, D.x);
                D.x_zero = false;
                continue;
            }
            CsrArgs a{};
            a.x = D.x; a.y = D.xo; a.b = D.b; a.omega = relax;
            int st;   // (row-partitioned level: the halo of x travels beside the interior rows)
            const bool ask_zr = smoother == SMOOTHER_JACOBI && post && level == 0 && s == nsweeps - 1 && h->want_zr && g_tune.fuse_zr;
            if (ask_zr) { a.partials = g_ctx.d_partials; g_jacobi_dot_done = false; }
            if (smoother == SMOOTHER_JACOBI) {
                a.diag = D.diag; st = dist_launch<OP_JACOBI>(D, D.A, a);
                if (ask_zr && g_jacobi_dot_done && st > 0) h->zr_G = st;
            }
            else { a.diag = D.l1; st = dist_launch<OP_L1DIAG>(D, D.A, a); }
            if (st < 0) return ERROR_MISC;
            std::swap(D.x, D.xo);
        }
        return FASP_SUCCESS;
    }
    if (!D.replicated) return ERROR_AMG_SMOOTH_TYPE;  // sequential sweeps are not distributed
    if (smoother == SMOOTHER_CG) return cg_smooth(h, level, nsweeps);
    const bool has_cf = h->H.L[level].cfmark.n == (size_t)n;
    if (smoother == SMOOTHER_GSF) {  // fasp_smoother_dcsr_gs_ff (ItrSmootherCSR.c:700): GS over the non-C rows, ascending, before and after
        if (!has_cf) {
            std::printf("### ERROR: fasp_hip: the F-point Gauss-Seidel smoother needs the C/F marker of a classical hierarchy\n");
            return ERROR_AMG_SMOOTH_TYPE;
        }
        for (int sw = 0; sw < nsweeps; ++sw) { const int st = seq_sweep(h, level, 3, 1, 0.0); if (st < 0) return st; }
        return FASP_SUCCESS;
    }
    auto rep = [&](int kind, int form, double w) -> int {  // nsweeps repetitions, as the `while (L--)` loops
        for (int s = 0; s < nsweeps; ++s) { const int st = seq_sweep(h, level, kind, form, w); if (st < 0) return st; }
        return FASP_SUCCESS;
    };
    int st = FASP_SUCCESS;
    switch (smoother) {
        case SMOOTHER_GS:
            if (order == NO_ORDER || !has_cf) st = rep(post ? 1 : 0, 0, 0.0);
            else if (order == CF_ORDER) {  // fasp_smoother_dcsr_gs_cf: pre C then F, post F then C
                for (int s = 0; s < nsweeps && st >= 0; ++s) {
                    st = seq_sweep(h, level, post ? 3 : 2, 1, 0.0);
                    if (st >= 0) st = seq_sweep(h, level, post ? 2 : 3, 1, 0.0);
                }
            }
            break;
        case SMOOTHER_SGS:
            for (int s = 0; s < nsweeps && st >= 0; ++s) {
                st = seq_sweep(h, level, 0, 1, 0.0);
                if (st >= 0) st = seq_sweep(h, level, 4, 1, 0.0);
            }
            break;
        case SMOOTHER_SOR: st = rep(post ? 1 : 0, 2, relax); break;
        case SMOOTHER_SSOR:
            st = rep(0, 2, relax);
            if (st >= 0) st = rep(1, 2, relax);
            break;
        case SMOOTHER_GSOR:
            if (!post) { st = rep(0, 0, 0.0); if (st >= 0) st = rep(1, 2, relax); }
            else       { st = rep(0, 2, relax); if (st >= 0) st = rep(1, 0, 0.0); }
            break;
        case SMOOTHER_SGSOR:
            if (!post) {
                st = rep(0, 0, 0.0); if (st >= 0) st = rep(1, 0, 0.0);
                if (st >= 0) st = rep(0, 2, relax); if (st >= 0) st = rep(1, 2, relax);
            } else {
                st = rep(0, 2, relax); if (st >= 0) st = rep(1, 2, relax);
                if (st >= 0) st = rep(0, 0, 0.0); if (st >= 0) st = rep(1, 0, 0.0);
            }
            break;
        default: return ERROR_AMG_SMOOTH_TYPE;
    }
    return st;
}

