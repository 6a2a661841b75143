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
static unsigned* g_seq_herr = nullptr;   // host-mapped: the kernels write it themselves (flow_give_up)
static unsigned* g_seq_derr = nullptr;   // its device address
static bool      g_flow_disabled = false;
static bool seq_err_pending() { return g_seq_herr && *reinterpret_cast<volatile unsigned*>(g_seq_herr) != 0u; }
static unsigned* seq_err_device_word()   // (allocated on first use, by whichever thread uploads a schedule first; nullptr: no host-mapped memory -- then a time-out shows as a failed solve only)
{
    static std::once_flag once;
    std::call_once(once, [] {
        unsigned* hp = nullptr;
        if (hipHostMalloc((void**)&hp, 64, hipHostMallocMapped) != hipSuccess) return;
        std::memset(hp, 0, 64);
        if (hipHostGetDevicePointer((void**)&g_seq_derr, hp, 0) != hipSuccess) g_seq_derr = nullptr;
        g_seq_herr = hp;
    });
    return g_seq_derr;
}
static void seq_err_watch(const unsigned*) {}   // (round 4: a device-to-host copy of the word behind every launch; the kernels write host memory now)
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

static int upload_split(SplitHost& H, DevLevel::Sched& S, hipStream_t stream = nullptr);
static int build_split(const HostCSR& A, const std::vector<int>& seq, DevLevel::Sched& S)
{
    const bool timing = std::getenv("FASP_HIP_SETUP_TIMING") != nullptr;
    SplitHost H;
    const int st = build_split_host(A, seq.data(), (int)seq.size(), g_tune.seq_strip_kb, g_tune.seq_lanes, timing, H, 0, g_tune.seq_spine, g_tune.seq_chain, g_tune.seq_chain_n1);   // (seq_sched.cpp)
    if (st != FASP_SUCCESS) { S.flow_ok = false; return st; }
    return upload_split(H, S);
}
static int upload_split(SplitHost& H, DevLevel::Sched& S, hipStream_t stream)
{
    if (!stream) stream = g_ctx.stream;
    const bool timing = std::getenv("FASP_HIP_SETUP_TIMING") != nullptr;
    const double t0 = wall_seconds();
    const int ns = H.ns;
    S.release();
    S.cptr = H.cptr;
    // ONE device allocation per schedule (a large hipMalloc costs about a millisecond; there were eighteen), the arrays at
    // 256-byte-aligned offsets inside it
    struct Piece { void** dst; const void* src; size_t bytes, copy; };
    std::vector<Piece> pieces;
    auto piece = [&](auto** dst, const auto* src, size_t n, bool has_src = true) {
        using TT = std::remove_pointer_t<std::remove_pointer_t<decltype(dst)>>;
        pieces.push_back({reinterpret_cast<void**>(dst), src, sizeof(TT) * std::max<size_t>(n, 1), has_src ? sizeof(TT) * n : 0});
    };
    FlowStrip* d_strips = nullptr; int* d_chunks = nullptr;
    static const unsigned zeros[64] = {};
    if (H.chain) {
        // the chain form (seq_chain.hip.h): the rest pass's CSR and records as in the split form, positions padded to blocks of 64
        const ChainHost& Cc = H.C;
        piece(&S.d_ria, H.ria.data(), (size_t)ns + 1);
        piece(&S.d_rja, H.rja.data(), (size_t)H.nrest);
        piece(&S.d_rval, H.rval.data(), (size_t)H.nrest);
        piece(&S.d_tr, H.tr.data(), 2 * (size_t)ns);
        piece(&S.d_rec, (const double*)nullptr, 2 * (size_t)ns, false);
        piece(&S.d_W, (const double*)nullptr, (size_t)ns + 128, false);     // W[npad] = 0.0: the operand of tier 2's padding entries; W[npad + 64 ..): scratch of the exporter
        piece(&S.d_G2, (const double*)nullptr, (size_t)ns, false);
        piece(&S.d_prog, zeros, 64);   // [0] role ticket, [1] error word, [2] tier-2 ticket
        piece(&S.d_band, Cc.band.data(), Cc.band.n);
        piece(&S.d_drd, Cc.drd.data(), Cc.drd.n);
        ChainBlk* d_blk = nullptr;
        piece(&d_blk, Cc.blk.data(), Cc.blk.n);
        piece(&S.d_t1v, Cc.t1v.data(), (size_t)Cc.t1_steps * 64);
        piece(&S.d_t1c, Cc.t1c.data(), (size_t)Cc.t1_steps * 64);
        piece(&S.d_t2v, Cc.t2v.data(), (size_t)Cc.t2_steps * 64);
        piece(&S.d_t2c, Cc.t2c.data(), (size_t)Cc.t2_steps * 64);
        piece(&S.d_t1need, Cc.t1need.data(), (size_t)Cc.t1_steps / 8);
        piece(&S.d_t2need, Cc.t2need.data(), (size_t)Cc.t2_steps / 8);
        size_t total = 0;
        for (const Piece& q : pieces) total += (q.bytes + 255) & ~(size_t)255;
        char* base = nullptr;
        HIPCK(hipMalloc((void**)&base, total));
        S.owned.push_back(base);
        size_t off = 0;
        for (const Piece& q : pieces) {
            *q.dst = base + off;
            if (q.copy) HIPCK(hipMemcpyAsync(base + off, q.src, q.copy, hipMemcpyHostToDevice, stream));
            else HIPCK(hipMemsetAsync(base + off, 0, q.bytes, stream));
            off += (q.bytes + 255) & ~(size_t)255;
        }
        const unsigned long long herr_addr = (unsigned long long)seq_err_device_word();   // (alive until the synchronisation below)
        HIPCK(hipMemcpyAsync(S.d_prog + 6, &herr_addr, 8, hipMemcpyHostToDevice, stream));
        HIPCK(hipStreamSynchronize(stream));
        S.d_blk = d_blk;
        if (timing) std::printf("    [chain schedule] %-28s %.3f s\n", "upload", wall_seconds() - t0);
        S.chain = true; S.nb = Cc.nb; S.npad = Cc.npad; S.n1b = Cc.n1b; S.rx = Cc.rx; S.rg = Cc.rg; S.t1_steps = Cc.t1_steps; S.t2_steps = Cc.t2_steps;
        S.nband = Cc.nband; S.nt1 = Cc.nt1; S.nt2 = Cc.nt2;
        S.ns = ns; S.L = 64; S.LR = H.LR; S.nolower = false; S.nvirt = 0; S.nrows = H.nrows; S.nclasses = H.nclasses; S.pfmax = 8; S.kt = 0; S.par = 1; S.nstrips = 0; S.nchunk = 0; S.maxent = H.maxent;
        S.nghost = 0; S.slot_bytes = 0; S.flow_ok = true;
        S.built = true;
        S.multicolor = false;
        return FASP_SUCCESS;
    }
    piece(&d_strips, H.strips.data(), H.strips.size());
    piece(&d_chunks, H.chunks.data(), 4 * (size_t)H.nchunk);
    piece(&S.d_slots, H.slots.data(), (size_t)H.slot_bytes);
    piece(&S.d_gpos, H.gpos.data(), (size_t)H.nghost);
    piece(&S.d_cstrip, H.cstrip.data(), (size_t)H.nchunk);
    piece(&S.d_lchunks, H.lchunks.data(), (size_t)H.nchunk);
    piece(&S.d_ria, H.ria.data(), (size_t)ns + 1);
    piece(&S.d_rja, H.rja.data(), (size_t)H.nrest);
    piece(&S.d_rval, H.rval.data(), (size_t)H.nrest);
    piece(&S.d_dr, H.dr.data(), 2 * (size_t)ns);
    piece(&S.d_tr, H.tr.data(), 2 * (size_t)ns);
    piece(&S.d_rec, (const double*)nullptr, 2 * (size_t)ns, false);
    piece(&S.d_W, (const double*)nullptr, (size_t)ns, false);
    piece(&S.d_prog, zeros, 64);   // [0] ticket counter, [1] error word
    size_t total = 0;
    for (const Piece& q : pieces) total += (q.bytes + 255) & ~(size_t)255;
    char* base = nullptr;
    HIPCK(hipMalloc((void**)&base, total));
    S.owned.push_back(base);
    size_t off = 0;
    for (const Piece& q : pieces) {
        *q.dst = base + off;
        if (q.copy) HIPCK(hipMemcpyAsync(base + off, q.src, q.copy, hipMemcpyHostToDevice, stream));
        off += (q.bytes + 255) & ~(size_t)255;
    }
    HIPCK(hipMemsetAsync(S.d_W, 0, sizeof(double) * (size_t)std::max(ns, 1), stream));
    const unsigned long long herr_addr = (unsigned long long)seq_err_device_word();   // (alive until the synchronisation below)
    HIPCK(hipMemcpyAsync(S.d_prog + 6, &herr_addr, 8, hipMemcpyHostToDevice, stream));
    HIPCK(hipStreamSynchronize(stream));   // (the host arrays go away with H)
    S.d_strips = d_strips; S.d_chunks = d_chunks;
    if (timing) std::printf("    [sweep schedule] %-28s %.3f s\n", "upload", wall_seconds() - t0);
    S.ns = ns; S.L = H.L; S.LR = H.LR; S.nolower = H.nolower; S.independent = H.nolower && H.independent; S.nvirt = H.nvirt; S.nrows = H.nrows; S.nclasses = H.nclasses; S.pfmax = H.pfs; S.kt = H.kt; S.par = H.par; S.nstrips = H.nstrips; S.nchunk = H.nchunk; S.maxent = H.maxent;
    S.nghost = H.nghost; S.slot_bytes = H.slot_bytes; S.flow_ok = H.flow_ok;
    S.built = true;
    S.multicolor = false;
    return FASP_SUCCESS;
}

// rows of sweep `kind` in sweep order: 0 all ascending, 1 all descending, 2 C rows, 3 the others, 4 descending from n - 2 (SGS).
// row0 / nloc / nglobal: the rows this rank owns of a row-partitioned level (local numbering in `seq`); whole level: 0 / n / n.
static void sweep_sequence(const HostLevel& HL, int kind, std::vector<int>& seq, int row0 = 0, int nloc = -1)
{
    const int nglobal = HL.A.row;
    if (nloc < 0) nloc = nglobal;
    seq.clear();
    seq.reserve((size_t)nloc);
    const int* cf = HL.cfmark.n ? HL.cfmark.data() + row0 : nullptr;
    switch (kind) {
        case 0: for (int i = 0; i < nloc; ++i) seq.push_back(i); break;
        case 1: for (int i = nloc - 1; i >= 0; --i) seq.push_back(i); break;
        case 2: for (int i = 0; i < nloc; ++i) if (cf && cf[i] == 1) seq.push_back(i); break;
        case 3: for (int i = 0; i < nloc; ++i) if (!cf || cf[i] != 1) seq.push_back(i); break;
        default: for (int i = nloc - 1; i >= 0; --i) if (row0 + i <= nglobal - 2) seq.push_back(i); break;
    }
}
// The schedules a hierarchy's smoother is going to need -- two sweep kinds on every level but the coarsest -- are built SIDE BY
// SIDE at the first sweep, one host thread each (the dependency pass of a schedule is sequential; fourteen of them are not): the
// first solve of P7(128) with the reference's defaults pays for the longest one instead of for the sum.
// the two sweep kinds the hierarchy's smoother runs on level l (-1: none / not a sequential smoother)
static bool sched_kinds(const fasp_hip_amg* h, int l, int& k0, int& k1)
{
    k0 = k1 = -1;
    const bool has_cf = h->H.L[(size_t)l].cfmark.n == (size_t)h->H.L[(size_t)l].A.row;
    switch (h->param.smoother) {
        case SMOOTHER_GS: if (h->param.smooth_order == CF_ORDER && has_cf) { k0 = 2; k1 = 3; } else { k0 = 0; k1 = 1; } break;
        case SMOOTHER_SGS: k0 = 0; k1 = 4; break;
        case SMOOTHER_SOR: case SMOOTHER_SSOR: case SMOOTHER_GSOR: case SMOOTHER_SGSOR: k0 = 0; k1 = 1; break;
        case SMOOTHER_GSF: k0 = 3; k1 = -1; break;
        default: return false;
    }
    return true;
}
static void sched_job_launch(fasp_hip_amg* h, int level, int kind, int team)
{
    const int strip_kb = g_tune.seq_strip_kb, lanes = g_tune.seq_lanes, spine = g_tune.seq_spine, chain = g_tune.seq_chain, chain_n1 = g_tune.seq_chain_n1;
    h->sched_jobs.emplace_back(new SchedJob);
    SchedJob* J = h->sched_jobs.back().get();
    J->level = level; J->kind = kind;
    const HostLevel* HL = &h->H.L[(size_t)level];
    const int dev = g_ctx.device;
    J->th = std::thread([J, HL, team, strip_kb, lanes, spine, chain, chain_n1, dev]() {
        std::vector<int> seq;
        sweep_sequence(*HL, J->kind, seq);
        J->st = build_split_host(HL->A, seq.data(), (int)seq.size(), strip_kb, lanes, false, J->H, team, spine, chain, chain_n1);
        // the schedule goes to the device from here, over a stream of its own: the first sweep finds it there
        hipStream_t st = nullptr;
        if (J->st == FASP_SUCCESS && hipSetDevice(dev) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess) {
            J->st = upload_split(J->H, J->S, st);
            J->uploaded = J->st == FASP_SUCCESS;   // (a failed upload has released what it had allocated: nothing to hand over)
            if (!J->uploaded) J->S.release();
            (void)hipStreamDestroy(st);
            J->H = SplitHost();
        }
    });
}
static void sched_jobs_start(fasp_hip_amg* h)
{
    if (h->sched_jobs_started) return;
    h->sched_jobs_started = true;
    if (g_tune.gs_multicolor || !g_tune.seq_jobs) return;
    const int nl = (int)h->L.size();
    std::vector<std::pair<int, int>> jobs;
    for (int l = 0; l + 1 < nl; ++l) {
        if (!h->L[l].replicated) return;   // (sequential sweeps run on whole levels only)
        int k0, k1;
        if (!sched_kinds(h, l, k0, k1)) return;
        if (h->L[l].sched[k0].built == false) jobs.push_back({l, k0});
        if (k1 >= 0 && !h->L[l].sched[k1].built) jobs.push_back({l, k1});
    }
    const int team = std::max(2, host_threads() * 2 / std::max<int>(1, (int)jobs.size()));
    for (const auto& lk : jobs) sched_job_launch(h, lk.first, lk.second, team);
}
// The same for ONE level, while the host setup is still at work on the coarser ones (fasp_hip_amg_create: the thread that uploads
// level l ahead of the setup calls this behind the upload): the greedy C/F passes of the setup are sequential, the cores are
// there, and the first solve of P7(256) with the reference's defaults no longer waits 1.7 s for its schedules.
// early = true: called when only the level's matrix is final -- sweeps in natural order need nothing else and start then
// (the C/F-ordered ones wait for the marker: early = false, behind the level's upload)
static void sched_jobs_start_level(fasp_hip_amg* h, int l, bool early = false)
{
    if (g_tune.gs_multicolor || !g_tune.seq_jobs || !h->L[(size_t)l].replicated) return;
    const int sm = h->param.smoother;
    const bool natural = sm == SMOOTHER_SGS || sm == SMOOTHER_SOR || sm == SMOOTHER_SSOR || sm == SMOOTHER_GSOR || sm == SMOOTHER_SGSOR ||
                         (sm == SMOOTHER_GS && h->param.smooth_order != CF_ORDER);
    if (early != natural) return;
    int k0, k1;
    if (early) { k0 = 0; k1 = sm == SMOOTHER_SGS ? 4 : 1; }
    else if (!sched_kinds(h, l, k0, k1)) return;
    const int team = std::max(2, host_threads() / 8);
    std::lock_guard<std::mutex> lk(h->sched_mu);
    sched_job_launch(h, l, k0, team);
    if (k1 >= 0) sched_job_launch(h, l, k1, team);
    h->sched_jobs_started = true;   // (what the first sweep would start is under way)
}
static void sched_jobs_join(fasp_hip_amg* h)   // (hierarchy teardown)
{
    for (auto& J : h->sched_jobs) {
        if (!J) continue;
        if (J->th.joinable()) J->th.join();
        if (J->uploaded) J->S.release();   // (a schedule nobody came for)
    }
    h->sched_jobs.clear();
}

// one sequential sweep of schedule `kind` with update formula `form` (see tri_update) over the rows THIS rank holds of the level
static int seq_sweep_local(fasp_hip_amg* h, int level, int kind, int form, double w)
{
    DevLevel& D = h->L[level];
    DevLevel::Sched& S = D.sched[kind];
    // Multicolour mode -- a FLAGGED NON-PARITY mode for speed: the sweep visits the rows colour by colour instead of
    // in the reference's index order, which is a different (equally convergent, deterministic) Gauss-Seidel / SOR
    // iteration; iteration counts and residuals then differ from the reference's.  The default is the split sweep
    // (seq_split.hip.h), which reproduces the reference's sequential sweep.
    const bool multicolor = g_tune.gs_multicolor != 0;
    if (!S.built || S.multicolor != multicolor) {
        // a row-partitioned level: the rank's own rows in local numbering -- ghost columns (>= the row count) are never "lower":
        // they hold what the ranks swept before this one have sent (new values) or what the later ones still have (old values)
        const HostCSR& A = D.replicated ? h->H.L[level].A : h->dist.L[level].A;
        if (A.row != D.nloc || !A.ia.data()) {   // (the local operator is kept on the host only when the hierarchy was uploaded with seq_partition on)
            std::printf("### ERROR: fasp_hip: sequential sweep on a partitioned level without its local host matrix (fasp_hip_tune(\"seq_partition\", 1) before the upload)\n");
            return ERROR_AMG_SMOOTH_TYPE;
        }
        std::vector<int> seq;
        const double t0 = wall_seconds();
        int st = 1000;
        if (multicolor && !D.replicated) {
            std::printf("### ERROR: fasp_hip: the multicolour sweep mode runs on whole levels only\n");
            return ERROR_AMG_SMOOTH_TYPE;
        }
        if (!multicolor && D.replicated) {
            sched_jobs_start(h);
            for (auto& J : h->sched_jobs)
                if (J && J->level == level && J->kind == kind) {
                    if (J->th.joinable()) J->th.join();
                    st = J->st;
                    if (st == FASP_SUCCESS && J->uploaded) { S.release(); S = std::move(J->S); J->S = DevLevel::Sched(); }
                    else if (st == FASP_SUCCESS) st = upload_split(J->H, S);
                    else { if (J->uploaded) J->S.release(); S.flow_ok = false; }
                    J.reset();
                    break;
                }
        }
        if (st == 1000 || st == 1) sweep_sequence(h->H.L[level], kind, seq, D.replicated ? 0 : D.row0, D.replicated ? -1 : D.nloc);
        if (st == 1000) st = multicolor ? build_schedule(A, seq, S, true) : build_split(A, seq, S);
        if (st < 0) return st;
        S.rowlevels = false;
        if (st == 1) {   // a row of this sweep reads more earlier rows than a strip's LDS holds: whole rows, one launch per dependency level
            if ((st = build_schedule(A, seq, S, false)) < 0) return st;
            S.rowlevels = true;
        }
        if (std::getenv("FASP_HIP_SETUP_TIMING")) {
            if (S.chain) std::printf("  [sweep schedule] level %d, sweep kind %d: CHAIN form, %d rows in %d dependency classes, %d blocks of 64, band %lld entries, tier 1 (%d blocks, ring of %d) %lld entries in %lld steps, "
                                     "tier 2 %lld entries in %lld steps, rest pass %d lanes per row, built in %.3f s\n", level, kind, S.nrows, S.nclasses, S.nb, S.nband, S.n1b, S.rx, S.nt1, S.t1_steps, S.nt2, S.t2_steps, S.LR, wall_seconds() - t0);
            else if (multicolor || S.rowlevels) std::printf("  [sweep schedule] level %d, sweep kind %d, %s: %d rows in %d classes\n", level, kind, multicolor ? "colours" : "whole-row dependency levels", (int)(seq.empty() ? S.ns : (int)seq.size()), (int)S.ptr.size() - 1);
            else std::printf("  [sweep schedule] level %d, sweep kind %d: %d rows in %d dependency classes, %d strips (%lld ghosts, at most %d values in LDS), %d chunks, %d lanes per row, "
                             "%d rounds (%d of them spine), %.1f slot bytes per row (%d virtual rows), rest pass %d lanes per row%s, built in %.3f s\n",
                             level, kind, S.nrows, S.nclasses, S.nstrips, S.nghost, S.maxent + 1, S.nchunk, S.L, S.pfmax, S.kt, S.nrows ? (double)S.slot_bytes / S.nrows : 0.0, S.nvirt, S.LR,
                             "", wall_seconds() - t0);
        }
    }
    // the level's vector is all zeros when this sweep starts (the first sweep of a pre-smoothing step; one rank -- in a sweep by turns the
    // halo has been exchanged by then): pass (1) has nothing to read but b
    const int zero_old = (D.x_zero && g_tune.seq_zero_skip && (D.replicated || comm_size() <= 1)) ? 1 : 0;
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
    const int ns = S.chain ? S.nrows : S.ns;   // (chain form: S.ns = positions padded to blocks of 64; the rest pass runs over the rows)
    if (ns == 0) return FASP_SUCCESS;
    FlowArgs fa{};
    fa.strips = (const FlowStrip*)S.d_strips; fa.chunks = (const int4*)S.d_chunks; fa.slots = S.d_slots; fa.gpos = S.d_gpos; fa.cstrip = S.d_cstrip;
    fa.rec = S.d_rec; fa.dr = S.d_dr; fa.tr = S.d_tr; fa.W = S.d_W; fa.u = D.x;
    fa.sync = S.d_prog; fa.nstrips = S.nstrips; fa.form = form; fa.w = w; fa.kt = S.kt;
    if (g_tune.seq_rest_lanes > 0) { int lr = 1; while (lr < 64 && lr < g_tune.seq_rest_lanes) lr *= 2; S.LR = lr; }   // (A/B)
    if (S.nolower && S.independent && !S.chain) {   // the rows of the sweep do not couple: one pass, in place
        const int rpb = BLOCK / S.LR;
        const int grid = std::max(1, std::min(MAXGRID, (ns + rpb - 1) / rpb));
#define DIRECT_LAUNCH(LL) hipLaunchKernelGGL((k_split_direct<LL>), dim3(grid), dim3(BLOCK), 0, g_ctx.stream, ns, (const int*)S.d_tr, \
        (const int*)S.d_ria, (const int*)S.d_rja, (const double*)S.d_rval, (const double*)S.d_dr, (const double*)D.b, D.x, form, w, zero_old)
        switch (S.LR) {
            case 1: DIRECT_LAUNCH(1); break;
            case 2: DIRECT_LAUNCH(2); break;
            case 4: DIRECT_LAUNCH(4); break;
            case 8: DIRECT_LAUNCH(8); break;
            case 16: DIRECT_LAUNCH(16); break;
            case 32: DIRECT_LAUNCH(32); break;
            default: DIRECT_LAUNCH(64); break;
        }
#undef DIRECT_LAUNCH
        return FASP_SUCCESS;
    }
    // pass (1): everything that reads old values, all rows at once
    {
        const int rpb = BLOCK / S.LR;
        const int grid = std::max(1, std::min(MAXGRID, (ns + rpb - 1) / rpb));
#define REST_LAUNCH(LL) hipLaunchKernelGGL((k_split_rest<LL>), dim3(grid), dim3(BLOCK), 0, g_ctx.stream, ns, (const int*)S.d_tr, \
        (const int*)S.d_ria, (const int*)S.d_rja, (const double*)S.d_rval, (const double*)D.b, (const double*)D.x, S.d_rec, S.d_W, S.d_prog, \
        S.chain ? S.d_G2 : (double*)nullptr, S.chain ? S.npad : 0, zero_old)
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
    if (S.chain) {
        // pass (2), the chain form (seq_chain.hip.h): one workgroup walks the rows, the others sum what lies far behind it
        ChainArgs ca{};
        ca.band = (const f64x2_t*)S.d_band; ca.drd = (const f64x2_t*)S.d_drd; ca.blk = (const ChainBlk*)S.d_blk;
        ca.t1v = S.d_t1v; ca.t1c = S.d_t1c; ca.t1need = S.d_t1need; ca.t2v = S.d_t2v; ca.t2c = S.d_t2c; ca.t2need = S.d_t2need; ca.rec = S.d_rec; ca.tr = S.d_tr;
        ca.W = S.d_W; ca.G2 = S.d_G2; ca.u = D.x; ca.sync = S.d_prog; ca.nb = S.nb; ca.npad = S.npad; ca.rx = S.rx; ca.rg = S.rg; ca.has_t2 = S.t2_steps > 0; ca.form = form; ca.w = w;
        const bool plain = g_tune.seq_chain_ref || !g_tune.seq_flow || g_flow_disabled;
        if (!plain && seq_err_check() < 0) return ERROR_MISC;   // an earlier sweep's time-out that has arrived meanwhile
        const size_t dyn = sizeof(double) * ((size_t)S.rx + 1 + 2 * (size_t)S.rg);
        // tier-2 workgroups: measured on levels 5-8 of P7(256) (profiles/r05_gs_chain.txt): 95 against 47 -- natural-order sweeps of the last two
        // levels (560-680 tier-2 entries per row) 773 -> 651 and 652 -> 515 us, everything else unchanged; 191: no further gain; 15: 2 x slower there
        const int far_wg = S.t2_steps > 0 && !g_tune.seq_test_hang ? (g_tune.seq_chain_grid > 0 ? g_tune.seq_chain_grid : 95) : 0;   // (seq_test_hang: nobody sums tier 2 -- the time-out path, tests)
        ca.touch_lead = (far_wg > 8 && g_tune.seq_chain_touch > 0) ? g_tune.seq_chain_touch : 0;
        ca.touch_t1 = g_tune.seq_chain_touch_t1;   // (tier 1's entries too: 298.7 -> 295.9 ms per GS-default solve of P7(256))   // (a workgroup of the launch lands on the chain's XCD only when there are more than eight others)
#define CHAIN_LAUNCH(FF)                                                                                              \
        if (plain) hipLaunchKernelGGL((k_tri_chain_ref<FF>), dim3(1), dim3(64), 0, g_ctx.stream, ca, S.n1b);             \
        else hipLaunchKernelGGL((k_tri_chain<FF>), dim3(1 + far_wg), dim3(CHAIN_NT), dyn, g_ctx.stream, ca)
        switch (form) {
            case 0: CHAIN_LAUNCH(0); break;
            case 1: CHAIN_LAUNCH(1); break;
            default: CHAIN_LAUNCH(2); break;
        }
#undef CHAIN_LAUNCH
        if (!plain) seq_err_watch(S.d_prog + 1);
        return FASP_SUCCESS;
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
#define FLOW_ONE(LL, PP)                                                                                                    \
        {                                                                                                                   \
            static bool attr_set = false;                                                                                   \
            if (!attr_set) {                                                                                                \
                HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tri_flow<LL, PP>),                            \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * (FLOW_LDS_ENT + 1)))); \
                attr_set = true;                                                                                            \
            }                                                                                                               \
            const int nt = FlowGeom<PP>::NT, per_cu = std::max(1, std::min(2048 / nt, by_lds));                             \
            const int grid = std::max(1, std::min(std::min(S.nstrips, per_cu * g_ctx.num_cu), g_tune.seq_grid > 0 ? g_tune.seq_grid : g_tune.seq_grid < 0 || S.par >= S.nstrips ? 1 << 30 : 2 * S.par + 2)); \
            hipLaunchKernelGGL((k_tri_flow<LL, PP>), dim3(grid), dim3(nt), dyn, g_ctx.stream, fa);                      \
        }
#define FLOW_LAUNCH(LL)                                                                                                     \
        if (S.pfmax > 4) FLOW_ONE(LL, TRI_PFMAX)                                                                            \
        else FLOW_ONE(LL, 4)
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

// A sequential sweep of a ROW-PARTITIONED level (round 4; fasp_hip_tune("seq_partition", 1) or FASP_HIP_SEQ_PARTITION=1 before the
// upload -- by default hierarchies with sequential smoothers keep every level whole on every rank).  With contiguous row blocks in
// rank order the sweep visits rank 0's rows, then rank 1's, ...: the ranks take turns -- halo exchange (a rank's ghosts then hold
// the NEW values of the ranks before it and the OLD ones of the ranks after it, exactly what the sequential sweep reads), the rank
// whose turn it is sweeps its rows (ghost columns belong to the parallel pass), the next exchange passes its boundary on.  The
// same iteration as on one GPU, no parallelism across ranks in a sweep (there is none in the reference's order), the level's
// matrix and vectors stay distributed.  Descending sweeps take the ranks in descending order.  Every rank enters every exchange.
static int seq_sweep(fasp_hip_amg* h, int level, int kind, int form, double w)
{
    DevLevel& D = h->L[level];
    if (D.replicated || comm_size() <= 1) return seq_sweep_local(h, level, kind, form, w);
    const int P = comm_size(), me = comm_rank();
    const bool descending = kind == 1 || kind == 4;
    materialise_zero(D);
    int st = FASP_SUCCESS;
    for (int s = 0; s < P; ++s) {
        const int turn = descending ? P - 1 - s : s;
        if (halo_exchange(D, D.x) < 0) return ERROR_MISC;
        if (turn == me && st >= 0) st = seq_sweep_local(h, level, kind, form, w);
    }
    return st;
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
    if (!D.replicated && (smoother == SMOOTHER_CG || !g_seq_partition)) return ERROR_AMG_SMOOTH_TYPE;  // (CG smoothing, and sequential sweeps unless asked for, run on whole levels)
    if (smoother == SMOOTHER_CG) return cg_smooth(h, level, nsweeps);
    const bool has_cf = h->H.L[level].cfmark.n == (size_t)D.nglobal;
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

