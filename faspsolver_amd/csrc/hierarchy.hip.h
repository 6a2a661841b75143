// hierarchy.hip.h -- the resident hierarchy: per-level device data, halo exchange, upload.
// Part of the single translation unit solver.hip (included there, in this order; not a stand-alone header).

// ---------------------------------------------------------------------------
// resident hierarchy
// ---------------------------------------------------------------------------
struct DevLevel {
    DevCSR  A, P, R;
    double* diag = nullptr;  // last diagonal hit per row (Jacobi: ItrSmootherCSR.c:160)
    double* l1   = nullptr;  // sum_j |a_ij| (L1-diag: ItrSmootherCSR.c:1566)
    double *b = nullptr, *xa = nullptr, *xb = nullptr, *w = nullptr;
    double* x  = nullptr;    // current iterate: xa or xb
    double* xo = nullptr;    // the other buffer
    bool    x_zero = true;   // x is (conceptually) all zeros and not materialised
    bool    presmoothed = false;  // x already holds the first Jacobi sweep from zero (written by the kernel that produced b)
    double* halo_pending = nullptr;  // Krylov operator bundle: vector whose halo the next operator exchanges (cycles.hip.h, csr_ops)
    bool    owns_b = true;
    // distribution (single GPU: nloc == nvec == rows, no halo)
    bool    replicated = true;   // whole level on every rank, computed redundantly
    int     nloc = 0;            // owned entries of this level's vectors
    int     nvec = 0;            // vector length incl. ghost entries [nloc, nvec)
    int     row0 = 0;            // global index of the first owned row
    int     nglobal = 0;
    std::vector<int> send_off, recv_off;  // nranks+1 each
    int*    d_send_idx = nullptr;
    double* d_sendbuf  = nullptr;
    bool    has_halo() const { return !replicated && nvec > nloc; }
    // level schedules of the sequential sweeps (built on first use): kind 0 ascending,
    // 1 descending, 2 ascending C rows, 3 ascending F rows, 4 descending from row n-2
    struct Sched {
        bool built = false, multicolor = false; int* d_order = nullptr; int* d_ptr = nullptr; std::vector<int> ptr;
        // split form (seq_split.hip.h): strips / chunks / slots of the lower part (rows and virtual rows), the rest as a CSR, per-position records and W
        int ns = 0, L = 1, LR = 1, pfmax = 1, kt = 0, par = 1, nstrips = 0, nchunk = 0, maxent = 0; bool nolower = false, flow_ok = false, rowlevels = false, independent = false; int nrows = 0, nvirt = 0, nclasses = 0; long long nghost = 0, slot_bytes = 0;
        std::vector<int> cptr;   // split form: dependency class -> first entry of d_lchunks (k_tri_level)
        void* d_strips = nullptr; void* d_chunks = nullptr; unsigned char* d_slots = nullptr; int* d_gpos = nullptr; int* d_cstrip = nullptr; int* d_lchunks = nullptr;
        int* d_ria = nullptr; int* d_rja = nullptr; double* d_rval = nullptr; double* d_rec = nullptr; double* d_dr = nullptr; int* d_tr = nullptr; double* d_W = nullptr; unsigned* d_prog = nullptr;
        // chain form (seq_chain.hip.h): band planes, per-position records, the two tiers, tier 2's sums
        bool chain = false; int nb = 0, npad = 0, n1b = 0, rx = 0, rg = 0; long long t1_steps = 0, t2_steps = 0, nband = 0, nt1 = 0, nt2 = 0;
        double* d_band = nullptr; double* d_drd = nullptr; void* d_blk = nullptr; double* d_t1v = nullptr; double* d_t2v = nullptr;
        unsigned short* d_t1c = nullptr; unsigned short* d_t2c = nullptr; double* d_G2 = nullptr; int* d_t1need = nullptr; int* d_t2need = nullptr;
        std::vector<void*> owned;   // device arrays of the split form (d_order and d_ptr among them)
        void release()
        {
            if (!owned.empty()) { for (void* q : owned) (void)hipFree(q); owned.clear(); }
            else { if (d_order) (void)hipFree(d_order); if (d_ptr) (void)hipFree(d_ptr); }
            d_order = nullptr; d_ptr = nullptr; built = false;
        }
    };
    Sched   sched[5];
    // polynomial smoother (built on first use): 1 / first diagonal hit, the coefficients k[1..5] of
    // ItrSmootherCSRpoly.c:101-109, work vectors r, rbar, v0, v1, vnew
    struct Poly { bool built = false; double* dinv = nullptr; double k[6] = {0, 0, 0, 0, 0, 0}; double* w[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; };
    Poly    poly;
    int*    d_perm = nullptr;  // brick renumbering behind a coded level: this level's order (new -> natural) on the device, for the numbering bridges of the transfer operators above
    bool    diag_uniform = false;  // every row has the same diagonal entry (bitwise): diag_value
    double  diag_value = 0.0;
    int*    d_mark = nullptr;  // C/F marker on the device (Jacobi-F smoother), built on first use
    double* w2 = nullptr;      // AMLI cycle: the coarse residual r1 of the level above, built on first use
    double* kw[4] = {nullptr, nullptr, nullptr, nullptr};  // K-cycle work vectors r, x1, v1, v2 of this level
};

struct EventPair { hipEvent_t a, b; };

}  // namespace fasp

using namespace fasp;

// a sweep schedule under construction on a host thread of its own (smoothers.hip.h, sched_jobs_start)
struct SchedJob { int level = 0, kind = 0, st = 0; bool uploaded = false; SplitHost H; DevLevel::Sched S; std::thread th; };   // S: the schedule on the device, uploaded by the job's own thread

struct fasp_hip_amg {
    // brick renumbering (upload_level): perm[l][k] = natural index of the row that level l's device copy numbers k (empty: natural order)
    std::vector<std::vector<int>> perm;
    std::vector<std::unique_ptr<SchedJob>> sched_jobs;   // sequential smoothers: schedules being built side by side at the first sweep
    bool                  sched_jobs_started = false;
    std::mutex            sched_mu;    // jobs are started from the setup thread (natural-order sweeps) or the upload thread (C/F-ordered ones)
    HostHierarchy         H;
    DistPlan              dist;        // row partition (nranks == 1: trivial)
    bool                  distributed = false;  // level 0 is row-partitioned over the ranks
    std::vector<DevLevel> L;
    AMG_param             param;  // copy of the user's parameters after setup
    // PCG asks the preconditioner for the partials of (z, r) (fused into the last Jacobi sweep of level 0): want_zr is
    // set around the last cycle of an apply, zr_G = number of partials waiting in g_ctx.d_partials (0: not produced)
    bool                  want_zr = false;
    bool                  pre_marked = false;   // the next precond_amg finds level 0's first Jacobi sweep written (K.mark_presmoothed)
    int                   zr_G = 0;
    // Krylov work vectors on level 0
    double *b = nullptr, *u = nullptr, *p = nullptr, *t = nullptr, *r = nullptr;
    // coarse-level SPCG work vectors
    double *cp = nullptr, *cr = nullptr, *ct = nullptr, *cbest = nullptr;
    // GMRES basis vectors (allocated on first use): level-0 set and coarse-level set
    std::vector<double*> gm[2];
    size_t               gm_len[2] = {0, 0};
    double*              gm_hh = nullptr;  // device Hessenberg column
    double*              spcg_fused_buf = nullptr;  // second parity of r, p, t and the broadcast record (k_spcg_fused)
    int                  spcg_last_iters = 0;   // iterations of the previous coarse solve: sizes the first batch of the next one
    SpcgState*           spcg_state = nullptr;  // device-resident state of the batched coarse CG
    // k_spcg_persist: the coarsest matrix dealt to the waves of the chip (built on first use; tried == true afterwards)
    struct Persist { bool tried = false, ok = false; int NE = 0, nblocks = 0; double* vals = nullptr; unsigned short* cols = nullptr;
                     int *wrow = nullptr, *wend = nullptr; double* t2 = nullptr; unsigned* sync = nullptr; unsigned launches = 0; } persist;
    std::vector<double>  amli_coef;             // AMLI polynomial coefficients (amli_degree + 1), formed on first use
    std::vector<int>     level_cycle_type;      // AMG_data.cycle_type per level as the setup leaves it (K-cycle)
    bool                 use_fmg = false;       // the preconditioner is one full-multigrid cycle (precond_type == PREC_FMG)
    // instrumentation
    std::vector<EventPair> ev;
    int                    ev_used = 0;
    long long              coarse_iters = 0, vcycles = 0;
    // lazy coarse verdicts (precond_amg): device words {min status, iteration sum}, their pinned host copy; coarse_sync: a coarse
    // solve gave up once on this hierarchy -- verdicts are read per solve from then on
    int*                   d_lazy = nullptr; int* h_lazy = nullptr; bool lazy_active = false, coarse_sync = false;
    double*                reg_img = nullptr; int reg_mc = 0;   // k_spcg_reg: the coarsest matrix as its threads hold it
    double                 upload_seconds = 0.0;
};

namespace fasp {

static void free_level(DevLevel& D)
{
    D.A.release(); D.P.release(); D.R.release();
    if (D.diag) (void)hipFree(D.diag);
    if (D.l1) (void)hipFree(D.l1);
    if (D.b && D.owns_b) (void)hipFree(D.b);
    if (D.xa) (void)hipFree(D.xa);
    if (D.xb) (void)hipFree(D.xb);
    if (D.w) (void)hipFree(D.w);
    if (D.d_send_idx) (void)hipFree(D.d_send_idx);
    if (D.d_sendbuf) (void)hipFree(D.d_sendbuf);
    for (auto& sc : D.sched) sc.release();
    if (D.poly.dinv) (void)hipFree(D.poly.dinv);
    for (double* q : D.poly.w) if (q) (void)hipFree(q);
    if (D.d_mark) (void)hipFree(D.d_mark);
    if (D.d_perm) (void)hipFree(D.d_perm);
    D.d_perm = nullptr;
    if (D.w2) (void)hipFree(D.w2);
    for (double* q : D.kw) if (q) (void)hipFree(q);
    D = DevLevel();
}

static int alloc_vec(double** p, size_t n)
{
    HIPCK(hipMalloc(p, sizeof(double) * std::max<size_t>(n, 1)));
    return FASP_SUCCESS;
}

// diag / l1 are derived on the host in the reference's order (one pass, setup time)
static int upload_diag(const HostCSR& A, DevLevel& D)
{
    const int n = A.row;
    std::vector<double> d(n, 0.0), s(n, 0.0);
    std::vector<int>    dp(n, -1);
    int                 ndup = 0;
#pragma omp parallel for schedule(static) reduction(+ : ndup)
    for (int i = 0; i < n; ++i) {
        double di = 0.0, si = 0.0;
        int    hits = 0;
        for (int k = A.ia[i]; k < A.ia[i + 1]; ++k) {
            if (A.ja[k] == i) { di = A.val[k]; dp[i] = k; ++hits; }
            si += (A.val[k] >= 0.0) ? A.val[k] : -A.val[k];
        }
        d[i] = di; s[i] = si;
        if (hits > 1) ++ndup;
    }
    D.A.dup_diag = ndup > 0;
    {   // one diagonal value for all rows (constant-coefficient operators): whoever only needs d_i can take it as a scalar and skip an
        // 8-byte-per-row stream (k_cg_update's first Jacobi sweep of the new residual: 134 MB of 1.07 GB on P7(256))
        bool same = n > 0;
        for (int i = 1; i < n && same; ++i) same = std::memcmp(&d[(size_t)i], &d[0], sizeof(double)) == 0;
        D.diag_uniform = same;
        D.diag_value = same ? d[0] : 0.0;
    }
    if (!D.A.sorted) {  // (storage indices of the host order: meaningless for a re-sorted device copy)
        HIPCK(hipMalloc(&D.A.dpos, sizeof(int) * std::max(n, 1)));
        HIPCK(hipMemcpy(D.A.dpos, dp.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    }
    if (alloc_vec(&D.diag, n) < 0 || alloc_vec(&D.l1, n) < 0) return ERROR_ALLOC_MEM;
    HIPCK(hipMemcpy(D.diag, d.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(D.l1, s.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    return FASP_SUCCESS;
}

// gather v[idx[i]] into a contiguous send buffer
__global__ __launch_bounds__(BLOCK) void k_pack(int n, const int* __restrict__ idx,
                                                 const double* __restrict__ v, double* __restrict__ out)
{
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) out[i] = v[idx[i]];
}

// Refresh the ghost entries [nloc, nvec) of a level-l vector from their owners: pack the
// entries the peers need, one grouped RCCL send/recv, receive straight into the ghost
// slots (ghosts are sorted by owner, so every peer's block is contiguous).
static int halo_exchange(DevLevel& D, double* v, hipStream_t stream = nullptr)
{
    if (!stream) stream = g_ctx.stream;
    // Every rank of a distributed level enters the exchange, also with nothing to send or receive: whether a level
    // is distributed is known to all ranks alike, whether THIS rank has halo traffic is not (a rank that
    // skipped the collective left its peers in the transport's barriers).
    if (D.replicated || comm_size() <= 1 || D.send_off.empty()) return FASP_SUCCESS;
    const int P = comm_size(), me = comm_rank();
    const int nsend = D.send_off.back();
    if (nsend > 0)
        hipLaunchKernelGGL(k_pack, dim3(vec_grid(nsend)), dim3(BLOCK), 0, stream, nsend, D.d_send_idx, v,
                           D.d_sendbuf);
    std::vector<CommXfer> sends, recvs;
    for (int q = 0; q < P; ++q) {
        if (q == me) continue;
        const int ns = D.send_off[q + 1] - D.send_off[q], nr = D.recv_off[q + 1] - D.recv_off[q];
        if (ns > 0) sends.push_back({q, D.d_sendbuf + D.send_off[q], (size_t)ns});
        if (nr > 0) recvs.push_back({q, v + D.nloc + D.recv_off[q], (size_t)nr});
    }
    const int st = comm_exchange(sends.data(), (int)sends.size(), recvs.data(), (int)recvs.size(), stream);
    if (st < 0) comm_mark_failed();
    return st;
}

// ---- halo exchange overlapped with the interior rows ------------------------------------------------------------
// y = OP(M, x) on a row-partitioned level V (the level x lives on):  the exchange of x's ghost entries runs on its own
// stream while the compute stream works on the rows [win_lo, win_hi) of M that read no ghost column; the boundary
// rows (the two ends of the block: with a 1-D partition of a grid in natural ordering, one plane each) follow when
// the ghosts have arrived.  Three launches of the same kernel through its row window: the arithmetic of every row is
// what the single launch does, only the partition of the fused dot product into per-block partials changes.
static int g_seq_partition = std::getenv("FASP_HIP_SEQ_PARTITION") && std::atoi(std::getenv("FASP_HIP_SEQ_PARTITION")) != 0;   // sequential smoothers on row-partitioned levels (ranks take turns)
static int g_halo_overlap = 1;   // fasp_hip_tune("halo_overlap", 0): exchange, then one launch (the round-1 sequence)

// ---- replicated levels, "split" mode (fasp_hip_tune("coarse_mode", 1)) -----------------------------------------------
// Levels below FASP_HIP_DIST_MIN_ROWS are replicated: every rank holds the whole level and (mode 0) computes all of it,
// which costs no time but caps the speed-up (Amdahl).  Mode 1 keeps the vectors replicated and splits the WORK: a rank
// applies an operator to its share of the rows only (row windows, multiples of WIN_ALIGN) and one all-gather completes
// the result on every rank.  Per operator: 1 / P of the kernel + an all-gather of 8 m bytes -- worth it where the kernel
// is long against the collective's latency (DESIGN.md section 4 has the per-level model); the arithmetic of a row does
// not depend on who computes it, so the iteration is the same one.  Operators with fewer rows than
// `g_coarse_split_min` stay redundant.
// Round 5: mode -1 (the default) = split where the transport makes the all-gather cheap -- peer windows: one small kernel, no
// library call -- and redundant otherwise (RCCL: a collective launch per operator; shared memory: a host round trip).
static int g_coarse_mode = -1, g_coarse_split_min = 16384;
static bool coarse_split_active(const DevCSR& M)
{
    return (g_coarse_mode == 1 || (g_coarse_mode < 0 && comm_is_peer_window())) && comm_size() > 1 && M.row >= g_coarse_split_min;
}
template <int OP>
static int rep_launch(const DevCSR& M, CsrArgs a)
{
    if (!coarse_split_active(M) || a.partials || a.zx) return launch_csr<OP>(M, a);
    const int P = comm_size(), me = comm_rank();
    const int nwin = (M.row + WIN_ALIGN - 1) / WIN_ALIGN, per = (nwin + P - 1) / P;
    std::vector<int> counts((size_t)P), displs((size_t)P);
    for (int q = 0; q < P; ++q) {
        const int lo = std::min(M.row, q * per * WIN_ALIGN), hi = std::min(M.row, (q + 1) * per * WIN_ALIGN);
        displs[(size_t)q] = lo; counts[(size_t)q] = hi - lo;
    }
    RowWin w; w.lo = displs[(size_t)me]; w.hi = w.lo + counts[(size_t)me]; w.goff = 0;
    int G = 0;
    if (w.hi > w.lo) G = launch_csr<OP>(M, a, w);
    if (comm_allgatherv(a.y + w.lo, counts[(size_t)me], a.y, counts.data(), displs.data(), g_ctx.stream) < 0) { comm_mark_failed(); return -1; }
    return G;
}

// (Out: the level the result lives on when it is not V -- the prolongation from a replicated level onto a distributed one
// is that rank's own rows and must not be split)
template <int OP>
static int dist_launch(DevLevel& V, const DevCSR& M, CsrArgs a, const DevLevel* Out = nullptr)
{
    double* v = const_cast<double*>(a.x);
    if (V.replicated && (!Out || Out->replicated) && comm_size() > 1) return rep_launch<OP>(M, a);
    if (V.replicated || comm_size() <= 1 || V.send_off.empty()) return launch_csr<OP>(M, a);
    if (!g_halo_overlap || M.win_hi < 0) {
        if (halo_exchange(V, v) < 0) return -1;
        return launch_csr<OP>(M, a);
    }
    // the exchange waits for whatever produced x, then runs beside the interior rows.  A local HIP failure never makes this
    // rank skip the exchange its peers enter (they would wait for it: the shared-memory transport until its timeout, RCCL
    // for ever): the failure is recorded -- the next fetch_red turns it into ERROR_MISC on every rank -- and the sequence
    // goes on in exchange-then-launch order.
    bool local_ok = hipEventRecord(g_ctx.ev_ready, g_ctx.stream) == hipSuccess &&
                    hipStreamWaitEvent(g_ctx.comm_stream, g_ctx.ev_ready, 0) == hipSuccess;
    if (!local_ok) {
        comm_mark_failed();
        (void)hipStreamSynchronize(g_ctx.stream);
        (void)halo_exchange(V, v);
        (void)launch_csr<OP>(M, a);
        return -1;
    }
    RowWin w;
    w.lo = M.win_lo; w.hi = M.win_hi; w.goff = 0;
    int G = launch_csr<OP>(M, a, w);
    // (the transport may block the host -- the shared-memory one does: the interior launch is queued before it)
    const int hst = halo_exchange(V, v, g_ctx.comm_stream);
    if (hipEventRecord(g_ctx.ev_halo, g_ctx.comm_stream) != hipSuccess ||
        hipStreamWaitEvent(g_ctx.stream, g_ctx.ev_halo, 0) != hipSuccess) {
        comm_mark_failed();
        (void)hipStreamSynchronize(g_ctx.comm_stream);   // the compute stream could not be ordered behind the halo: wait here
        local_ok = false;
    }
    w.lo = 0; w.hi = M.win_lo; w.goff = G;
    G += launch_csr<OP>(M, a, w);
    w.lo = M.win_hi; w.hi = M.row; w.goff = G;
    G += launch_csr<OP>(M, a, w);
    return (hst < 0 || !local_ok) ? -1 : G;
}

// ---- brick renumbering of the uncoded mid levels (round 5; reorder.cpp) ---------------------------------------------------
// Levels 1 .. n-2 whose operators are not row-pattern / dictionary coded are renumbered in breadth-first balls of 64 rows when the
// smoother does not depend on the order of the rows (Jacobi, L1): A_l with rows and columns permuted, R_{l-1} / P_{l-1} and
// R_l / P_l with the side that lives on level l permuted; the vectors of such a level simply live in the new order (level 0 and the
// coarsest level -- whose safe CG sums dot products -- keep theirs).  Row sums keep their storage order: a cycle is bit-identical
// with and without it (tests/test_gpu_scale.py::test_renumbered_levels_are_bit_transparent).  Not for the sequential smoothers (they
// sweep in index order), the recursive cycles and coarse scaling (dot products on the mid levels), row-partitioned hierarchies.
static bool renumber_param_ok(const AMG_param& p)
{
    return (p.smoother == SMOOTHER_JACOBI || p.smoother == SMOOTHER_L1DIAG) &&
           (p.cycle_type == V_CYCLE || p.cycle_type == W_CYCLE || p.cycle_type == VW_CYCLE || p.cycle_type == WV_CYCLE) &&
           p.coarse_scaling != 1 && p.AMG_type == CLASSIC_AMG;
}
static bool renumber_allowed(const fasp_hip_amg* h)
{
    if (g_oneshot_upload) return false;   // (one solve per setup -- fasp_solver_dcsr_krylov_amg: like the matrix coding, it would cost more than it saves)
    return g_tune.renumber != 0 && comm_size() == 1 && renumber_param_ok(h->param);
}
// The numbering is decided ONCE, at upload, from the parameters of the setup; a later call may bring other ones (fasp_hip_amg_solve's
// per-call AMG_param, a precond_data the caller edited between setup and apply).  Sweep schedules, C/F markers and polynomial
// diagonals are built from the host hierarchy in natural order: applied to brick-ordered vectors they would be silently wrong.
// Such a call is refused; fasp_hip_tune("renumber", 0) before the setup keeps every level natural.
static int renumber_conflict(const fasp_hip_amg* h, const AMG_param& p)
{
    bool any = false;
    for (const auto& pm : h->perm) any = any || !pm.empty();
    if (!any || renumber_param_ok(p)) return FASP_SUCCESS;
    std::printf("### ERROR: fasp_hip: this hierarchy was uploaded for an order-independent smoother (its mid levels are renumbered); "
                "smoother %d / cycle %d / coarse_scaling %d need the natural order -- set them before the setup, or "
                "fasp_hip_tune(\"renumber\", 0)\n", (int)p.smoother, (int)p.cycle_type, (int)p.coarse_scaling);
    return ERROR_INPUT_PAR;
}
static bool dev_coded(const DevCSR& M) { return M.pat != nullptr || M.code != nullptr; }

// the transfer operators between levels l and l + 1 (P_l: rows of level l; R_l: rows of level l + 1), in the numberings of both
static int upload_transfer(fasp_hip_amg* h, int l, const DistLevel* DLp)
{
    const HostLevel& HL = h->H.L[l];
    DevLevel& D = h->L[l];
    const bool rep = !DLp || DLp->replicated;
    const std::vector<int>* pf = (size_t)l < h->perm.size() && !h->perm[(size_t)l].empty() ? &h->perm[(size_t)l] : nullptr;
    const std::vector<int>* pc = (size_t)l + 1 < h->perm.size() && !h->perm[(size_t)l + 1].empty() ? &h->perm[(size_t)l + 1] : nullptr;
    // Level l coded, level l + 1 renumbered: the transfer operators keep the natural numbering on BOTH sides -- permuted, their row
    // patterns would be gone (measured: R_1 47 -> 126 us, P_1 102 -> 120 us on P7(256)) -- and a numbering bridge permutes the vector of
    // level l + 1 on the fly (DevCSR::bridge: 1.4 M entries, a few microseconds).
    const bool bridged = pc && !pf && dev_coded(D.A);
    if (bridged) {
        if (upload_csr(HL.P, D.P) < 0) return ERROR_ALLOC_MEM;
        if (upload_csr(HL.R, D.R) < 0) return ERROR_ALLOC_MEM;
        DevLevel& C = h->L[(size_t)l + 1];
        if (!C.d_perm) {
            HIPCK(hipMalloc(&C.d_perm, sizeof(int) * pc->size()));
            HIPCK(hipMemcpy(C.d_perm, pc->data(), sizeof(int) * pc->size(), hipMemcpyHostToDevice));
        }
        HIPCK(hipMalloc(&D.R.bscratch, sizeof(double) * pc->size()));
        HIPCK(hipMalloc(&D.P.bscratch, sizeof(double) * pc->size()));
        D.R.bridge = C.d_perm; D.R.bridge_dir = 1;
        D.P.bridge = C.d_perm; D.P.bridge_dir = 2;
    } else if (pf || pc) {
        std::vector<int> invf, invc;
        if (pf) { invf.resize(pf->size()); for (size_t k = 0; k < pf->size(); ++k) invf[(size_t)(*pf)[k]] = (int)k; }
        if (pc) { invc.resize(pc->size()); for (size_t k = 0; k < pc->size(); ++k) invc[(size_t)(*pc)[k]] = (int)k; }
        HostCSR T;
        permute_csr(HL.P, pf ? pf->data() : nullptr, pc ? invc.data() : nullptr, T);
        if (upload_csr(T, D.P) < 0) return ERROR_ALLOC_MEM;
        HIPCK(hipStreamSynchronize(g_ctx.stream));   // (T goes away)
        permute_csr(HL.R, pc ? pc->data() : nullptr, pf ? invf.data() : nullptr, T);
        if (upload_csr(T, D.R) < 0) return ERROR_ALLOC_MEM;
        HIPCK(hipStreamSynchronize(g_ctx.stream));
    } else {
        if (upload_csr(rep ? HL.P : DLp->P, D.P) < 0) return ERROR_ALLOC_MEM;
        if (upload_csr(rep ? HL.R : DLp->R, D.R) < 0) return ERROR_ALLOC_MEM;
    }
    // the grid plane of the transfer operators (XCD strips of the coded kernels, device_csr.hip.h): P's rows are this level's, R's
    // the next one's -- where the coarse rows are an exact fraction of the fine plane
    if (D.A.plane > 0 && D.R.row > 0 && D.R.col > 0) {
        D.P.plane = D.A.plane;
        const long long pr = (long long)D.A.plane * D.R.row;
        if (pr % D.R.col == 0) D.R.plane = (int)(pr / D.R.col);
    }
    if (!rep) {
        D.R.win_lo = DLp->winR[0]; D.R.win_hi = DLp->winR[1];
        D.P.win_lo = DLp->winP[0]; D.P.win_hi = DLp->winP[1];
    }
    return FASP_SUCCESS;
}

// One level to the device.  DL == nullptr: single rank, the level is whole (the overlapped upload of
// fasp_hip_amg_create runs this from a second thread while the host setup builds the next levels).
// With the renumbering allowed, the transfer operators of level l - 1 go up in THIS level's turn: only now is it known whether
// level l is smoothed (has a coarser level) and how it is numbered.
static int upload_level(fasp_hip_amg* h, int l, const DistLevel* DLp)
{
    const HostLevel& HL = h->H.L[l];
    DevLevel& D = h->L[l];
    const bool rep = !DLp || DLp->replicated;
    D.replicated = rep;
    D.nloc = rep ? HL.A.row : DLp->nloc; D.row0 = rep ? 0 : DLp->row0; D.nglobal = HL.A.row;
    D.nvec = rep ? HL.A.row : DLp->nloc + (int)DLp->ghosts.size();
    const HostCSR& A = rep ? HL.A : DLp->A;
    static const bool timing = std::getenv("FASP_HIP_SETUP_TIMING") != nullptr;
    double tp = wall_seconds();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const double now = wall_seconds();
        std::printf("    [upload level %d] %-10s %8.3f s\n", l, what, now - tp);
        tp = now;
    };
    const bool renum = renumber_allowed(h);
    if (h->perm.size() < h->L.size()) h->perm.resize(h->L.size());
    HostCSR Aperm;                 // the level's matrix in its device numbering (when renumbered)
    const HostCSR* Adev = &A;
    if (!rep && g_tune.local_square) {   // (the flag only steers the lossless coding of the device copy)
        HostCSR view;
        view.row = A.row; view.col = A.col; view.nnz = A.nnz; view.row_aligned = true;
        view.ia.view(const_cast<int*>(A.ia.data()), A.ia.n); view.ja.view(const_cast<int*>(A.ja.data()), A.ja.n); view.val.view(const_cast<double*>(A.val.data()), A.val.n);
        if (upload_csr(view, D.A) < 0) return ERROR_ALLOC_MEM;
    } else {
        // renumber this level?  Not level 0, not the coarsest, not behind a coded level (its coded transfer operators carry this level's
        // numbering in their patterns), not a level that is coded itself (short rows: tried in natural order first).
        // Levels of more than two million rows stay as they are: measured on the variable-coefficient twin of P7(256), level 1 (8.4 M rows,
        // 19 entries per row) gains 3 % per operator and costs the upload thread six seconds (profiles/r05_renumber.txt).
        bool want = renum && rep && l > 0 && HL.has_coarse && A.row >= 4096 && A.row <= 2000000 && (g_tune.renumber >= 2 || !dev_coded(h->L[(size_t)l - 1].A));   // (renumber = 2: also behind a coded level, whose transfer operators then take a numbering bridge, upload_transfer -- measured on P7(256): level 2 159 -> 132 us per product, the bridges 13 us each, 0.4 ms of 40 per solve for 1.5 s more setup: off by default, profiles/r05_renumber.txt)
        bool uploaded = false;
        if (want && compress_enabled() && (double)A.nnz <= 48.0 * A.row) {
            if (upload_csr(A, D.A) < 0) return ERROR_ALLOC_MEM;
            if (dev_coded(D.A)) { want = false; uploaded = true; }
            else { HIPCK(hipStreamSynchronize(g_ctx.stream)); D.A.release(); D.A = DevCSR(); }
        }
        if (want) {
            std::vector<int>& pm = h->perm[(size_t)l];
            cluster_order(A, std::max(4096, g_tune.renumber_chunk), pm);
            std::vector<int> inv(pm.size());
            for (size_t k = 0; k < pm.size(); ++k) inv[(size_t)pm[k]] = (int)k;
            permute_csr(A, pm.data(), inv.data(), Aperm);
            Adev = &Aperm;
            lap("renumber");
        }
        if (!uploaded && upload_csr(*Adev, D.A) < 0) return ERROR_ALLOC_MEM;
    }
    lap("A");
    if (renum) {   // the transfer operators of the level above, now that this level's numbering is known
        if (l > 0 && h->H.L[(size_t)l - 1].has_coarse && !h->L[(size_t)l - 1].P.ia) {
            const int st = upload_transfer(h, l - 1, nullptr);
            if (st < 0) return st;
            lap("P, R of the level above");
        }
    } else if (HL.has_coarse) {
        const int st = upload_transfer(h, l, DLp);
        if (st < 0) return st;
        lap("P, R");
    }
    if (!rep) { D.A.win_lo = DLp->winA[0]; D.A.win_hi = DLp->winA[1]; }   // interior windows (dist_launch), found with the partition (dist_plan.cpp)
    HIPCK(hipStreamSynchronize(g_ctx.stream));
    lap("sync");
    if (upload_diag(*Adev, D) < 0) return ERROR_ALLOC_MEM;
    lap("diag");
    const size_t n = D.nvec;
    if (l > 0) { if (alloc_vec(&D.b, n) < 0) return ERROR_ALLOC_MEM; }
    else D.owns_b = false;  // level-0 rhs aliases the Krylov residual (PreCSR.c:429 copy elided)
    if (alloc_vec(&D.xa, n) < 0 || alloc_vec(&D.xb, n) < 0 || alloc_vec(&D.w, n) < 0) return ERROR_ALLOC_MEM;
    D.x = D.xa; D.xo = D.xb; D.x_zero = true;
    if (!rep) {
        D.send_off = DLp->send_off; D.recv_off = DLp->recv_off;
        const size_t ns = DLp->send_idx.size();
        HIPCK(hipMalloc(&D.d_send_idx, sizeof(int) * std::max<size_t>(ns, 1)));
        HIPCK(hipMalloc(&D.d_sendbuf, sizeof(double) * std::max<size_t>(ns, 1)));
        if (ns) HIPCK(hipMemcpy(D.d_send_idx, DLp->send_idx.data(), sizeof(int) * ns, hipMemcpyHostToDevice));
    }
    return FASP_SUCCESS;
}

static int upload_hierarchy(fasp_hip_amg* h)
{
    HostThreads team;  // matrix coding / re-sorting / partition loops
    const double t0 = wall_seconds();
    const int nl = (int)h->H.L.size();
    h->L.resize(nl);   // (never grows here when levels were uploaded ahead: the vector was reserved for MAX_AMG_LVL + 1)
    int min_rows = 200000;
    if (const char* e = std::getenv("FASP_HIP_DIST_MIN_ROWS")) min_rows = std::atoi(e);
    // sequential (Gauss-Seidel / SOR) sweeps couple all rows of a level: such hierarchies
    // are not row-partitioned, every rank keeps (and computes) all levels
    const bool seq_ok = g_seq_partition && h->param.smoother != SMOOTHER_CG && h->param.smoother != SMOOTHER_JACOBIF;   // (sweeps by turns: smoothers.hip.h, seq_sweep)
    if (h->param.smoother != SMOOTHER_JACOBI && h->param.smoother != SMOOTHER_L1DIAG && h->param.smoother != SMOOTHER_POLY && !seq_ok) min_rows = 2147483647;
    if (h->param.cycle_type == AMLI_CYCLE || h->param.cycle_type == NL_AMLI_CYCLE) min_rows = 2147483647;  // the recursive cycles run on whole levels
    {   // AMG_data.cycle_type of every level: the setup's cycle type for levels >= 1 (PreAMGSetupRS.c:325,
        // PreAMGSetupSA.c:495); the UA setup derives it from the operator complexity (PreAMGSetupUA.c:390-401)
        h->level_cycle_type.assign((size_t)nl, h->param.cycle_type);
        h->level_cycle_type[0] = 0;
        if (h->param.AMG_type == UA_AMG) {
            const double cplxmax = 3.0, xsi = 0.6, eta = xsi / ((1 - xsi) * (cplxmax - 1));
            int icum = 1;
            h->level_cycle_type[0] = 1;
            h->level_cycle_type[(size_t)nl - 1] = 0;
            for (int lvl = 1; lvl < nl - 1; ++lvl) {
                const double fracratio = (double)h->H.L[lvl].A.nnz / h->H.L[0].A.nnz;
                int ct = (int)(std::pow(xsi, (double)lvl) / (eta * fracratio * icum));
                ct = std::max(1, std::min(2, ct));
                h->level_cycle_type[(size_t)lvl] = ct;
                icum = icum * ct;
            }
        }
    }
    {
        const int st = build_dist_plan(h->H, comm_rank(), comm_size(), min_rows, h->dist);
        if (st < 0) return st;
    }
    h->distributed = !h->dist.L[0].replicated;
    for (int l = 0; l < nl; ++l) {
        if (h->L[l].A.ia) continue;   // uploaded already, while the host setup was still running (single rank: replicated)
        const int st = upload_level(h, l, &h->dist.L[l]);
        if (st < 0) return st;
    }
    const size_t m = h->L[0].nvec;
    if (alloc_vec(&h->b, m) < 0 || alloc_vec(&h->u, m) < 0 || alloc_vec(&h->p, m) < 0 ||
        alloc_vec(&h->t, m) < 0 || alloc_vec(&h->r, m) < 0) return ERROR_ALLOC_MEM;
    HIPCK(hipMemsetAsync(h->u, 0, sizeof(double) * m, g_ctx.stream));
    HIPCK(hipMemsetAsync(h->p, 0, sizeof(double) * m, g_ctx.stream));
    const size_t mc = h->L[nl - 1].nvec;
    if (alloc_vec(&h->cp, mc) < 0 || alloc_vec(&h->cr, mc) < 0 || alloc_vec(&h->ct, mc) < 0 ||
        alloc_vec(&h->cbest, mc) < 0) return ERROR_ALLOC_MEM;
    h->ev.resize(64);
    for (auto& e : h->ev) { HIPCK(hipEventCreate(&e.a)); HIPCK(hipEventCreate(&e.b)); }
    HIPCK(hipStreamSynchronize(g_ctx.stream));
    // the local copies of the partitioned operators are only needed for the upload -- and, for sequential smoothers on partitioned
    // levels, the local A for the sweep schedules (smoothers.hip.h, seq_sweep_local)
    for (auto& DL : h->dist.L) { if (!seq_ok) DL.A = HostCSR(); DL.P = HostCSR(); DL.R = HostCSR(); }
    h->upload_seconds = wall_seconds() - t0;
    return FASP_SUCCESS;
}

