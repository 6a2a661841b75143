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

// Error word of the cluster form of the triangular solve (a solver workgroup that was not resident: bounded spins, then this
// word).  It is copied to pinned host memory behind every launch and looked at -- without waiting -- before the next sweep and
// where a solve synchronises anyway: a broken cluster fails the solve loudly, and the process goes on without that form.
static unsigned* g_seq_herr = nullptr;
static bool seq_err_pending() { return g_seq_herr && *g_seq_herr != 0u; }
static void seq_err_watch(const unsigned* d_err)
{
    if (!g_seq_herr && hipHostMalloc((void**)&g_seq_herr, 64, hipHostMallocDefault) != hipSuccess) { g_seq_herr = nullptr; return; }
    if (*g_seq_herr == 0u) (void)hipMemcpyAsync(g_seq_herr, d_err, sizeof(unsigned), hipMemcpyDeviceToHost, g_ctx.stream);
}
static int seq_err_check()   // after a stream synchronisation
{
    if (!seq_err_pending()) return FASP_SUCCESS;
    std::fprintf(stderr, "### ERROR: fasp_hip: a workgroup of the clustered triangular solve (sequential smoothers) was not resident; "
                         "fasp_hip_tune(\"seq_cluster\", 0) selects one launch per dependency class\n");
    return ERROR_MISC;
}

// ---------------------------------------------------------------------------
// The split form of a sequential sweep (seq_split.hip.h): class-major numbering of the swept rows by TRUE dependencies
// only (row i after the coupled rows the sweep visits before it), the lower part in slot storage, the rest as a CSR
// in sweep numbering.  Built once per (level, sweep kind) on first use.
// ---------------------------------------------------------------------------
constexpr int TRI_PF = 4;   // slot rounds the lanes-per-row choice aims at (TRI_PFMAX = 8 is what a chunk can store)
template <class T>
static int split_upload(DevLevel::Sched& S, T** dst, const std::vector<T>& v)
{
    *dst = nullptr;
    HIPCK(hipMalloc((void**)dst, sizeof(T) * std::max<size_t>(v.size(), 1)));
    S.owned.push_back(*dst);
    if (!v.empty()) HIPCK(hipMemcpy(*dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    return FASP_SUCCESS;
}
static int build_split(const HostCSR& A, const std::vector<int>& seq, DevLevel::Sched& S)
{
    HostThreads host_team;   // (bounded OpenMP team for the row-parallel loops below; the dependency levels themselves are a sequential pass)
    const int n = A.row, ns = (int)seq.size();
    std::vector<int> pos(n, -1), lev(ns, 0);
    for (int q = 0; q < ns; ++q) pos[seq[q]] = q;
    int nlev = ns > 0 ? 1 : 0;
    std::vector<int> nlow(ns, 0);
    auto is_lower = [&](int i, int q, int j) { return j != i && j < n && pos[j] >= 0 && pos[j] < q; };
    for (int q = 0; q < ns; ++q) {
        const int i = seq[q];
        int l = 0, c = 0;
        for (int k = A.ia[i]; k < A.ia[i + 1]; ++k)
            if (is_lower(i, q, A.ja[k])) { l = std::max(l, lev[pos[A.ja[k]]]); ++c; }
        lev[q] = l + 1; nlow[q] = c;
        nlev = std::max(nlev, l + 1);
    }
    std::vector<int> cls(nlev + 1, 0);
    for (int q = 0; q < ns; ++q) cls[lev[q]]++;
    for (int l = 0; l < nlev; ++l) cls[l + 1] += cls[l];
    std::vector<int> cur(cls.begin(), cls.end() - 1), newpos(ns), order(ns), len(ns);
    for (int q = 0; q < ns; ++q) { const int p = cur[lev[q] - 1]++; newpos[q] = p; order[p] = seq[q]; len[p] = nlow[q]; }
    // lanes per row of the triangular part: TRI_PF * L slots cover the lower entries of 90 % of the rows
    long long lower_total = 0;
    int len90 = 0;
    {
        std::vector<long long> hist(258, 0);
        for (int q = 0; q < ns; ++q) { hist[std::min(nlow[q], 257)]++; lower_total += nlow[q]; }
        long long acc = 0;
        for (int v = 0; v < 258; ++v) { acc += hist[v]; if (acc * 10 >= (long long)ns * 9) { len90 = v; break; } }
    }
    // wide classes (levels 1-2 of a 3-D problem: hundreds of rows each) keep every wavefront of the solving workgroup busy, and
    // what a chunk costs there is instructions, per LANE mostly: half the lanes with twice the rounds is less work per row.
    // Narrow classes (a few rows: the deep levels) are a latency chain: more lanes, shorter chains.
    const bool wide = nlev > 0 && ns / nlev >= 128;   // (measured at 128^3: classes of 190 and 1 700 rows gain 28-33 %, of 33 and 109 rows lose)
    int L = 1;
    while (L < 64 && (wide ? TRI_PFMAX : TRI_PF) * L < len90) L *= 2;
    if (g_tune.seq_lanes > 0) { L = 1; while (L < 64 && L < g_tune.seq_lanes) L *= 2; }
    // chunks: the classes cut into rounds of a TRI_BLOCK-thread workgroup; slots per lane = what the chunk's longest row needs
    const int rpb = TRI_BLOCK / L;
    std::vector<int> lo_of(1, 0);
    S.cptr.assign(1, 0);
    for (int l = 0; l < nlev; ++l) {
        for (int lo = cls[l]; lo < cls[l + 1]; lo += rpb) lo_of.push_back(std::min(lo + rpb, cls[l + 1]));
        S.cptr.push_back((int)lo_of.size() - 1);
    }
    const int nchunk = (int)lo_of.size() - 1;
    if (ns >= TRI_FAR_BIT) return ERROR_INPUT_PAR;
    std::vector<int> chunk_of(ns), pf_of(nchunk, 0), sbase(nchunk + 1, 0);
    for (int c = 0; c < nchunk; ++c)
        for (int p = lo_of[c]; p < lo_of[c + 1]; ++p) chunk_of[p] = c;
    // The LDS ring the one-workgroup solve keeps the new values in covers `ringcap` positions behind the end of the chunk
    // at work.  A lower entry further back is FAR: it goes to the row's tail, flagged, and is read from W in memory
    // (written there dozens of chunks -- several drained memory counters -- earlier).  The ring is the largest power of
    // two that fits beside the chunk descriptors; no far entries where it would be shorter than four groups of chunks.
    int ringcap = 0;
    for (int c = 16384; c >= 1024 && !ringcap; c >>= 1)
        if (c <= g_tune.seq_ring && (size_t)c * 8 + 2 * sizeof(int) * (size_t)(nchunk + 1) <= TRI_LDS_CAP) ringcap = c;   // (seq_ring: a smaller ring, so that tests meet far entries on small grids)
    if (ringcap < 16 * rpb) ringcap = 0;
    auto is_far = [&](int p, int cpos) { return ringcap > 0 && lo_of[chunk_of[p] + 1] - cpos > ringcap; };
    std::vector<int> nfar_of(ns, 0);
    long long nfar = 0;
#pragma omp parallel for schedule(static) reduction(+ : nfar)
    for (int q = 0; q < ns; ++q) {
        const int i = seq[q], p = newpos[q];
        int f = 0;
        for (int k = A.ia[i]; k < A.ia[i + 1]; ++k)
            if (is_lower(i, q, A.ja[k]) && is_far(p, newpos[pos[A.ja[k]]])) ++f;
        nfar_of[p] = f; nfar += f;
    }
    if (nfar * 50 > lower_total) {   // more than a few stragglers: not a schedule for the ring (seq_sweep then weighs the L2 form against launches)
        ringcap = 0; nfar = 0;
        std::fill(nfar_of.begin(), nfar_of.end(), 0);
    }
    for (int p = 0; p < ns; ++p) len[p] -= nfar_of[p];   // len: NEAR lower entries from here on
    S.ptr.assign(nchunk + 1, 0);
    long long nslot = 0;
    int pfmax = 1;
    double bytes_us = 0.0;   // the cost model of seq_sweep: microseconds of one workgroup's memory traffic
    for (int c = 0; c < nchunk; ++c) {
        int mx = 0;
        for (int p = lo_of[c]; p < lo_of[c + 1]; ++p) mx = std::max(mx, len[p]);
        pf_of[c] = lower_total == 0 ? 0 : std::max(1, std::min(TRI_PFMAX, (mx + L - 1) / L));   // (at least one round: tri_fetch is branch-free; none at all for a sweep without lower entries: it never reaches the triangular kernels)
        pfmax = std::max(pfmax, pf_of[c]);
        sbase[c] = (int)nslot;
        nslot += (long long)((pf_of[c] + 3) & ~3) * L * (lo_of[c + 1] - lo_of[c]);   // (packs of four rounds)
        if (nslot > 0x0fffffffll) return ERROR_INPUT_PAR;   // (byte offsets of the slot values stay below 2^31)
        S.ptr[c] = lo_of[c] | (pf_of[c] << 28);
        const double bytes = (double)(lo_of[c + 1] - lo_of[c]) * L * (12.0 * ((pf_of[c] + 3) & ~3) + 32.0);
        bytes_us += std::max(0.7, bytes / 25e3);   // (measured: one compute unit sustains ~25 GB/s of such fetches)
    }
    sbase[nchunk] = (int)nslot; S.ptr[nchunk] = ns;
    std::vector<int>    sc((size_t)nslot), tia(ns + 1, 0), ria(ns + 1, 0);
    std::vector<double> sv((size_t)nslot, 0.0), dr(2 * (size_t)ns, 0.0);
    std::vector<int>    tr(2 * (size_t)ns, 0);
#pragma omp parallel for schedule(dynamic, 64)
    for (int c = 0; c < nchunk; ++c) {
        const int lo = lo_of[c], hi = lo_of[c + 1];
        for (int q = 0; q < ((pf_of[c] + 3) & ~3); ++q)
            for (int p = lo; p < hi; ++p)
                for (int sl = 0; sl < L; ++sl) sc[(size_t)sbase[c] + ((size_t)(q / 4) * L * (hi - lo) + (size_t)(p - lo) * L + sl) * 4 + (q % 4)] = p;
    }
    long long ntail = 0, nrest = 0;
#pragma omp parallel for schedule(static) reduction(+ : ntail, nrest)
    for (int q = 0; q < ns; ++q) {
        const int i = seq[q], p = newpos[q];
        const int t = std::max(0, len[p] - TRI_PFMAX * L) + nfar_of[p];   // near entries beyond the slots + far entries
        tia[p + 1] = t; ntail += t;
        int r = 0;
        double dg = 0.0;
        for (int k = A.ia[i]; k < A.ia[i + 1]; ++k) {
            const int j = A.ja[k];
            if (j == i) { dg = A.val[k]; continue; }   // the last diagonal hit, as the reference's loop leaves it
            if (!is_lower(i, q, j)) ++r;
        }
        ria[p + 1] = r; nrest += r;
        const bool alone = !(std::fabs(dg) > SMALLREAL);
        dr[2 * (size_t)p] = dg; dr[2 * (size_t)p + 1] = alone ? 0.0 : 1.0 / dg;
        tr[2 * (size_t)p] = t | (alone ? (int)0x80000000 : 0); tr[2 * (size_t)p + 1] = i;
    }
    for (int p = 0; p < ns; ++p) { tia[p + 1] += tia[p]; ria[p + 1] += ria[p]; }
    std::vector<int>    tja((size_t)ntail), rja((size_t)nrest);
    std::vector<double> tval((size_t)ntail), rval((size_t)nrest);
    int reach = 0;   // how far back (in positions, from the end of its chunk) a row reads: the LDS ring must cover it
#pragma omp parallel for schedule(static) reduction(max : reach)
    for (int q = 0; q < ns; ++q) {
        const int i = seq[q], p = newpos[q];
        const int ck = chunk_of[p], lo = lo_of[ck], hi = lo_of[ck + 1];
        int e = 0;
        size_t kt = (size_t)tia[p], kr = (size_t)ria[p];
        for (int k = A.ia[i]; k < A.ia[i + 1]; ++k) {
            const int j = A.ja[k];
            if (j == i) continue;
            if (is_lower(i, q, j)) {
                const int cpos = newpos[pos[j]];
                if (is_far(p, cpos)) { tja[kt] = cpos | TRI_FAR_BIT; tval[kt] = A.val[k]; ++kt; continue; }
                reach = std::max(reach, hi - cpos);
                if (e < TRI_PFMAX * L) {
                    const int    qe = e / L;   // round of this entry
                    const size_t at = (size_t)sbase[ck] + ((size_t)(qe / 4) * L * (hi - lo) + (size_t)(p - lo) * L + (e % L)) * 4 + (qe % 4);
                    sc[at] = cpos; sv[at] = A.val[k];
                } else { tja[kt] = cpos; tval[kt] = A.val[k]; ++kt; }
                ++e;
            } else { rja[kr] = j; rval[kr] = A.val[k]; ++kr; }
        }
    }
    S.release();
    int st = FASP_SUCCESS;
    if ((st = split_upload(S, &S.d_order, order)) < 0) return st;
    if ((st = split_upload(S, &S.d_ptr, S.ptr)) < 0) return st;
    if ((st = split_upload(S, &S.d_sbase, sbase)) < 0) return st;
    if ((st = split_upload(S, &S.d_sc, sc)) < 0) return st;
    if ((st = split_upload(S, &S.d_sv, sv)) < 0) return st;
    if ((st = split_upload(S, &S.d_tia, tia)) < 0) return st;
    if ((st = split_upload(S, &S.d_tja, tja)) < 0) return st;
    if ((st = split_upload(S, &S.d_tval, tval)) < 0) return st;
    if ((st = split_upload(S, &S.d_ria, ria)) < 0) return st;
    if ((st = split_upload(S, &S.d_rja, rja)) < 0) return st;
    if ((st = split_upload(S, &S.d_rval, rval)) < 0) return st;
    if ((st = split_upload(S, &S.d_rec, std::vector<double>(2 * (size_t)ns, 0.0))) < 0) return st;
    if ((st = split_upload(S, &S.d_dr, dr)) < 0) return st;
    if ((st = split_upload(S, &S.d_tr, tr)) < 0) return st;
    if ((st = split_upload(S, &S.d_W, std::vector<double>((size_t)ns, 0.0))) < 0) return st;
    if ((st = split_upload(S, &S.d_prog, std::vector<unsigned>(64, 0u))) < 0) return st;   // [0] progress (one-workgroup form); cluster form: [4] arrivals, [5] error, [6] progress, [20 + b] XCC ids
    if ((st = split_upload(S, &S.d_cptr, S.cptr)) < 0) return st;
    {   // cluster form: the first chunk of every (class, solver)
        int maxw = 0;
        for (size_t l = 0; l + 1 < S.cptr.size(); ++l) maxw = std::max(maxw, S.cptr[l + 1] - S.cptr[l]);
        const int cnb = std::min(16, std::max(1, maxw));
        std::vector<int> cdesc(4 * (size_t)cnb * std::max<size_t>(S.cptr.size() - 1, 1), 0);
        for (size_t l = 0; l + 1 < S.cptr.size(); ++l)
            for (int b = 0; b < cnb; ++b) {
                const int c = S.cptr[l] + b;
                int* d = &cdesc[4 * (l * cnb + b)];
                if (c < S.cptr[l + 1]) { d[0] = S.ptr[c]; d[1] = lo_of[c + 1]; d[2] = sbase[c]; }
            }
        if ((st = split_upload(S, &S.d_cdesc, cdesc)) < 0) return st;
    }
    S.maxw = 0;
    for (size_t l = 0; l + 1 < S.cptr.size(); ++l) S.maxw = std::max(S.maxw, S.cptr[l + 1] - S.cptr[l]);
    S.nfar_chunks = 0;
    for (int c = 0; c < nchunk; ++c) {
        bool any = false;
        for (int p = lo_of[c]; p < lo_of[c + 1] && !any; ++p) any = nfar_of[p] > 0;
        S.nfar_chunks += any;
    }
    S.ns = ns; S.L = L; S.nolower = lower_total == 0; S.ntail = ntail; S.reach = reach; S.nfar = nfar; S.ringcap = ringcap; S.block_us = bytes_us; S.nslot = nslot; S.pfmax = pfmax;
    const double avg_rest = ns > 0 ? (double)nrest / ns : 0.0;
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
        const int st = multicolor ? build_schedule(A, seq, S, true) : build_split(A, seq, S);
        if (st < 0) return st;
        if (std::getenv("FASP_HIP_SETUP_TIMING")) {
            if (multicolor) std::printf("  [sweep schedule] level %d, sweep kind %d, colours: %d rows in %d classes\n", level, kind, (int)seq.size(), (int)S.ptr.size() - 1);
            else std::printf("  [sweep schedule] level %d, sweep kind %d: %d rows in %d dependency classes (%d chunks, reach %d, %lld far entries in %d chunks), %d lanes per row, %.1f slots per row (%lld tail entries), rest pass %d lanes per row, one-workgroup estimate %.0f us, built in %.3f s\n",
                             level, kind, S.ns, (int)S.cptr.size() - 1, (int)S.ptr.size() - 1, S.reach, S.nfar, S.nfar_chunks, S.L, S.ns ? (double)S.nslot / S.ns : 0.0, S.ntail, S.LR, S.block_us, wall_seconds() - t0);
        }
    }
    materialise_zero(D);
    if (multicolor) {
        const int nlev = (int)S.ptr.size() - 1;
        // one launch per colour; lanes per row as the level's SpMV kernel
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
    const int nchunk = (int)S.ptr.size() - 1, nlev = (int)S.cptr.size() - 1;
    TriArgs ta{};
    ta.lptr = S.d_ptr; ta.sbase = S.d_sbase; ta.nchunk = nchunk; ta.sc = S.d_sc; ta.sv = S.d_sv; ta.tia = S.d_tia; ta.tja = S.d_tja; ta.tval = S.d_tval;
    ta.rec = S.d_rec; ta.dr = S.d_dr; ta.tr = S.d_tr; ta.nrow = D.A.row; ta.order = S.d_order; ta.W = S.d_W; ta.u = D.x; ta.form = form; ta.w = w; ta.far = S.nfar > 0;
    // pass (1): everything that reads old values, all rows at once
    {
        const int rpb = BLOCK / S.LR;
        const int grid = std::max(1, std::min(MAXGRID, (ns + rpb - 1) / rpb));
#define REST_LAUNCH(LL) hipLaunchKernelGGL((k_split_rest<LL>), dim3(grid), dim3(BLOCK), 0, g_ctx.stream, ns, (const int*)S.d_order, \
        (const int*)S.d_ria, (const int*)S.d_rja, (const double*)S.d_rval, (const double*)D.b, (const double*)D.x, S.d_rec, S.d_prog)
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
        hipLaunchKernelGGL(k_split_scatter, dim3(sgrid), dim3(BLOCK), 0, g_ctx.stream, ns, ta, 1);
        return FASP_SUCCESS;
    }
    // pass (2) in ONE workgroup (k_tri_block) when its chain of chunks beats one launch per class.  MEASURED on P7(128)
    // (tools/perf_gs_levels.py): ~0.7 us per chunk with the new values in an LDS ring (more where chunks are wide: one
    // compute unit sustains ~25 GB/s of these fetches), ~2.2 us per chunk when the schedule reaches further back than the ring and
    // W goes through the L2; a launch per class costs ~3.2 us (classes of one chunk) to ~3.9 us.
    // fasp_hip_tune("seq_block", 0) / ("seq_ulds", 0) switch the workgroup form / the ring off: same slots, same row
    // arithmetic, same bits.
    const int L = S.L;
    const size_t lds_ptr = 2 * sizeof(int) * (size_t)(nchunk + 1);
    constexpr size_t LDS_CAP = TRI_LDS_CAP;
    int cap = 0;
    if (g_tune.seq_ulds) {
        if (S.nfar) cap = S.ringcap;   // (the far entries were chosen against exactly this ring)
        else {
            for (int c = 16384; c >= 1024 && !cap; c >>= 1)
                if (c <= g_tune.seq_ring && S.reach <= c && (size_t)c * 8 + lds_ptr <= LDS_CAP) cap = c;
            while (cap > 1024 && (cap >> 1) >= S.reach) cap >>= 1;   // (no larger than needed: the ring is zeroed per launch)
        }
    }
    const double cost_block = cap ? S.block_us + 1.5 * S.nfar_chunks : std::max(S.block_us, 2.2 * nchunk), cost_launch = (double)nlev * 3.2 + 0.15 * (nchunk - nlev);
    // the cluster form (k_tri_cluster): a few workgroups on one XCD, a barrier per class instead of a launch: 2.4 us per class
    // measured (0.7 the barrier round itself, the rest the chain through the L2 and the drained stores), 0.4 us per further
    // chunk of a solver
    const int  cl_nb = std::min(16, std::max(1, S.maxw));
    const double cost_cluster = (double)nlev * 2.4 + 0.4 * std::max(0, (nchunk - nlev * cl_nb + cl_nb - 1) / cl_nb);   // (measured on P7(128): 2.4-2.5 us per class)
    static bool cluster_disabled = false;
    if (seq_err_pending()) { cluster_disabled = true; return ERROR_MISC; }
    if (g_tune.seq_cluster && !cluster_disabled && !comm_shares_devices() && nlev >= 8 && cl_nb >= 2 &&
        cost_cluster < cost_launch && (!g_tune.seq_block || cost_cluster < cost_block || lds_ptr > LDS_CAP)) {
        const int nhelp = std::max(0, std::max(g_tune.seq_help, 8));   // (several compute units to stream what a cluster consumes)
        const double chunk_bytes = nchunk ? (12.0 * S.nslot + 40.0 * ns) / nchunk : 1.0;
        const int ahead = (int)std::min(4096.0, std::max(2.0 * cl_nb, 1.5e6 / chunk_bytes));
        unsigned* sync = S.d_prog + 4;
#define TRIC_LAUNCH(LL) hipLaunchKernelGGL((k_tri_cluster<LL>), dim3(8 * (cl_nb + nhelp)), dim3(TRI_BLOCK), 0, g_ctx.stream, ta, (const int*)S.d_cptr, (const int*)S.d_cdesc, nlev, ns, cl_nb, nhelp, ahead, sync)
        switch (L) {
            case 1: TRIC_LAUNCH(1); break;
            case 2: TRIC_LAUNCH(2); break;
            case 4: TRIC_LAUNCH(4); break;
            case 8: TRIC_LAUNCH(8); break;
            case 16: TRIC_LAUNCH(16); break;
            case 32: TRIC_LAUNCH(32); break;
            default: TRIC_LAUNCH(64); break;
        }
#undef TRIC_LAUNCH
        seq_err_watch(sync + 1);
        return FASP_SUCCESS;
    }
    if (g_tune.seq_block && nlev >= 8 && lds_ptr <= LDS_CAP && cost_block < cost_launch) {
        const size_t dyn = (cap ? (size_t)cap * 8 : 0) + lds_ptr;
        // helper workgroups that read ahead of the solver into the XCD's L2 (tri_prefetch): ~1.5 MB ahead, at least three groups of chunks
        const int nhelp = comm_shares_devices() ? 0 : std::max(0, g_tune.seq_help);
        const double chunk_bytes = nchunk ? (12.0 * S.nslot + 40.0 * ns) / nchunk : 1.0;
        const int ahead = (int)std::min(4096.0, std::max(12.0, 1.5e6 / chunk_bytes));
#define TRIB_ONE(LL, PP, WW, TT)                                                                                           \
        {                                                                                                                   \
            static bool attr_set = false;                                                                                   \
            if (!attr_set) {                                                                                                \
                HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tri_block<LL, PP, WW, TT>),                       \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_CAP));                       \
                attr_set = true;                                                                                            \
            }                                                                                                               \
            hipLaunchKernelGGL((k_tri_block<LL, PP, WW, TT>), dim3(1 + 8 * nhelp), dim3(TRI_BLOCK), dyn, g_ctx.stream, ta, ns, cap ? cap : 2, nhelp, ahead, S.d_prog); \
        }
#define TRIB_LAUNCH(LL)                                                                                                     \
        if (!cap) TRIB_ONE(LL, TRI_PFMAX, false, true)                                                                      \
        else if (S.ntail) TRIB_ONE(LL, TRI_PFMAX, true, true)                                                               \
        else if (S.pfmax > 4) TRIB_ONE(LL, TRI_PFMAX, true, false)                                                          \
        else TRIB_ONE(LL, 4, true, false)
        switch (L) {
            case 1: TRIB_LAUNCH(1); break;
            case 2: TRIB_LAUNCH(2); break;
            case 4: TRIB_LAUNCH(4); break;
            case 8: TRIB_LAUNCH(8); break;
            case 16: TRIB_LAUNCH(16); break;
            case 32: TRIB_LAUNCH(32); break;
            default: TRIB_LAUNCH(64); break;
        }
#undef TRIB_LAUNCH
#undef TRIB_ONE
        return FASP_SUCCESS;
    }
    // wide classes: one launch per class, one workgroup per chunk
    for (int l = 0; l < nlev; ++l) {
        const int c0 = S.cptr[l], grid = S.cptr[l + 1] - c0;
#define TRIL_LAUNCH(LL) hipLaunchKernelGGL((k_tri_level<LL>), dim3(grid), dim3(TRI_BLOCK), 0, g_ctx.stream, ta, c0)
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
    hipLaunchKernelGGL(k_split_scatter, dim3(sgrid), dim3(BLOCK), 0, g_ctx.stream, ns, ta, 0);
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

