// bsr.hip.h -- block (BSR) operators, hierarchy and cycle.
// Part of the single translation unit solver.hip (included there, in this order; not a stand-alone header).

// ---------------------------------------------------------------------------
// BSR operators (config 3): host pointers in/out, device kernels
// ---------------------------------------------------------------------------
namespace fasp_bsr {
struct TmpBSR {
    int ROW = 0, nb = 0, NNZ = 0;
    int *ia = nullptr, *ja = nullptr;
    double* val = nullptr;
    bool ok = false;
    explicit TmpBSR(const dBSRmat* A)
    {
        if (ctx_init() < 0 || !A || A->nb < 1 || A->nb > 7 || A->storage_manner != 0) return;
        ROW = A->ROW; nb = A->nb; NNZ = A->NNZ;
        const size_t nv = (size_t)NNZ * nb * nb;
        if (hipMalloc(&ia, sizeof(int) * ((size_t)ROW + 1)) != hipSuccess) return;
        if (hipMalloc(&ja, sizeof(int) * std::max(NNZ, 1)) != hipSuccess) return;
        if (hipMalloc(&val, sizeof(double) * std::max<size_t>(nv, 1)) != hipSuccess) return;
        (void)hipMemcpy(ia, A->IA, sizeof(int) * ((size_t)ROW + 1), hipMemcpyHostToDevice);
        (void)hipMemcpy(ja, A->JA, sizeof(int) * (size_t)NNZ, hipMemcpyHostToDevice);
        (void)hipMemcpy(val, A->val, sizeof(double) * nv, hipMemcpyHostToDevice);
        ok = true;
    }
    ~TmpBSR() { if (ia) (void)hipFree(ia); if (ja) (void)hipFree(ja); if (val) (void)hipFree(val); }
};

template <int OP>
void launch_bsr(const TmpBSR& M, BsrArgs a)
{
    a.ROW = M.ROW; a.ia = M.ia; a.ja = M.ja; a.val = M.val;
    const int rw = 64 / M.nb;
    a.ntiles = (M.ROW + 4 * rw - 1) / (4 * rw);
#define BSR_CASE(NBV)                                                                              \
    case NBV: {                                                                                    \
        int cap = resident_blocks_per_cu(k_bsr_wstream<NBV, OP>) * g_ctx.num_cu;                   \
        const int grid = std::max(1, std::min(std::min(cap, MAXGRID), a.ntiles));                  \
        hipLaunchKernelGGL((k_bsr_wstream<NBV, OP>), dim3(grid), dim3(BLOCK), 0, g_ctx.stream, a); \
    } break;
    switch (M.nb) {
        BSR_CASE(1) BSR_CASE(2) BSR_CASE(3) BSR_CASE(4) BSR_CASE(5) BSR_CASE(6) BSR_CASE(7)
        default: break;
    }
#undef BSR_CASE
}
[[noreturn]] void die_bsr(const char* fn)
{
    std::fprintf(stderr, "### ERROR: %s: needs a HIP device, storage_manner 0 and 1 <= nb <= 7 "
                         "(libfasp_hip has no CPU fallback)\n", fn);
    std::exit(ERROR_MISC);
}
}  // namespace fasp_bsr
using namespace fasp_bsr;

// ---------------------------------------------------------------------------
// BSR AMG hierarchy resident in HBM (config 3): unsmoothed aggregation, block-Jacobi
// V/W cycle (PreMGCycle.c:287), GMRES on the coarsest level, Krylov drivers shared with CSR
// ---------------------------------------------------------------------------
struct BsrLevel {
    std::unique_ptr<TmpBSR> A, P, R;
    double *dinv = nullptr, *b = nullptr, *x = nullptr, *x2 = nullptr, *w = nullptr;
    int  n = 0;  // scalar rows (owned)
    bool x_zero = false;
    DevLevel::Sched sched[2];  // level schedules of the sequential block sweeps: 0 ascending, 1 descending
    // row partition (one process per GPU; single GPU: replicated, nv == n): owned block rows [row0, row0 + nloc) of
    // nglobal, vectors of nv = (nloc + ghost blocks) * nb scalars, halo lists in SCALAR indices (a block = nb entries)
    bool replicated = true;
    int  nv = 0, row0 = 0, nloc = 0, nglobal = 0, nghost = 0;
    std::vector<int> send_off, recv_off;   // nranks + 1 each, in scalars
    int*    d_send_idx = nullptr;
    double* d_sendbuf = nullptr;
};
struct fasp_hip_amg_bsr {
    HostHierarchyBSR      H;
    std::vector<BsrLevel> L;
    AMG_param             param;
    DistPlan              dist;
    bool                  distributed = false;   // level 0 is row-partitioned over the ranks
    double *b = nullptr, *u = nullptr, *p = nullptr, *t = nullptr, *r = nullptr, *z = nullptr;
    std::vector<double*> gm[2];
    size_t               gm_len[2] = {0, 0};
    double*              gm_hh = nullptr;
    double*              small_ws = nullptr;  // workspace of the single-workgroup coarse GMRES
    long long            coarse_iters = 0, vcycles = 0;
};

namespace fasp_bsr {

static int dalloc(double** p, size_t n)
{
    HIPCK(hipMalloc(p, sizeof(double) * std::max<size_t>(n, 1)));
    HIPCK(hipMemsetAsync(*p, 0, sizeof(double) * n, g_ctx.stream));
    return 0;
}

static void bsr_mxv(const TmpBSR& M, const double* x, double* y)
{
    BsrArgs a{}; a.x = x; a.y = y;
    launch_bsr<0>(M, a);
}
// r = b - A x with the reference's rounding: y = b; y *= -1; y += A x; y *= -1 (BlaSpmvBSR.c:548)
static void bsr_resid(const TmpBSR& M, const double* x, const double* b, double* r)
{
    BsrArgs a{}; a.x = x; a.y = r; a.b = b; a.alpha = -1.0;
    launch_bsr<1>(M, a);
}
static int bsr_halo_fwd(BsrLevel& Lv, double* v);
static void bsr_jacobi(BsrLevel& Lv)
{
    const TmpBSR& M = *Lv.A;
    if (!Lv.x_zero && !Lv.replicated && bsr_halo_fwd(Lv, Lv.x) < 0) comm_mark_failed();   // the sweep reads its neighbours' old values
    if (Lv.x_zero) {
        hipLaunchKernelGGL(k_bsr_dinv_apply, dim3(vec_grid(Lv.n)), dim3(BLOCK), 0, g_ctx.stream, Lv.n, M.nb,
                           (const double*)Lv.dinv, (const double*)Lv.b, Lv.x);
        Lv.x_zero = false;
        return;
    }
    BsrArgs a{}; a.x = Lv.x; a.y = Lv.x2; a.b = Lv.b; a.dinv = Lv.dinv;
    launch_bsr<2>(M, a);
    std::swap(Lv.x, Lv.x2);
}

// One sequential block sweep (Gauss-Seidel or SOR, ascending or descending) as level-scheduled launches:
// the rows of a dependency level are mutually uncoupled, so the result is the sequential sweep.
static int bsr_seq_sweep(fasp_hip_amg_bsr* h, int level, bool descend, bool sor, double w)
{
    BsrLevel& Lv = h->L[level];
    DevLevel::Sched& S = Lv.sched[descend ? 1 : 0];
    const TmpBSR& M = *Lv.A;
    if (!S.built) {
        const HostBSR& A = h->H.L[level].A;
        HostCSR pat;  // block pattern only (build_schedule does not read values)
        pat.row = A.ROW; pat.col = A.COL; pat.nnz = A.NNZ;
        pat.ia.alloc((size_t)A.ROW + 1); pat.ja.alloc((size_t)std::max(A.NNZ, 1));
        std::memcpy(pat.ia.data(), A.ia.data(), sizeof(int) * ((size_t)A.ROW + 1));
        std::memcpy(pat.ja.data(), A.ja.data(), sizeof(int) * (size_t)A.NNZ);
        std::vector<int> seq((size_t)A.ROW);
        for (int i = 0; i < A.ROW; ++i) seq[(size_t)i] = descend ? A.ROW - 1 - i : i;
        const int st = build_schedule(pat, seq, S);
        if (st < 0) return st;
    }
    if (Lv.x_zero) { HIPCK(hipMemsetAsync(Lv.x, 0, sizeof(double) * Lv.n, g_ctx.stream)); Lv.x_zero = false; }
    const int nlev = (int)S.ptr.size() - 1;
    for (int l = 0; l < nlev; ++l) {
        const int lo = S.ptr[l], hi = S.ptr[l + 1];
        const int grid = std::max(1, std::min(MAXGRID, (hi - lo + BLOCK - 1) / BLOCK));
#define BSEQ_LAUNCH(NBV) hipLaunchKernelGGL((k_bsr_seq_level<NBV>), dim3(grid), dim3(BLOCK), 0, g_ctx.stream, \
        (const int*)S.d_order, lo, hi, (const int*)M.ia, (const int*)M.ja, (const double*)M.val, (const double*)Lv.b, \
        (const double*)Lv.dinv, Lv.x, sor ? 1 : 0, w)
        switch (M.nb) {
            case 1: BSEQ_LAUNCH(1); break;
            case 2: BSEQ_LAUNCH(2); break;
            case 3: BSEQ_LAUNCH(3); break;
            case 4: BSEQ_LAUNCH(4); break;
            case 5: BSEQ_LAUNCH(5); break;
            case 6: BSEQ_LAUNCH(6); break;
            case 7: BSEQ_LAUNCH(7); break;
            default: return ERROR_INPUT_PAR;
        }
#undef BSEQ_LAUNCH
    }
    return FASP_SUCCESS;
}

// smoother dispatch of fasp_solver_mgcycle_bsr, PreMGCycle.c:327-365 (pre) and :513-549 (post)
static int bsr_smooth(fasp_hip_amg_bsr* h, int level, bool post, int smoother, int steps, double relax)
{
    BsrLevel& Lv = h->L[level];
    int st = FASP_SUCCESS;
    if (steps <= 0) return st;
    if (!Lv.replicated && smoother != SMOOTHER_JACOBI) return ERROR_AMG_SMOOTH_TYPE;   // block Gauss-Seidel / SOR sweeps couple all rows: whole levels only
    switch (smoother) {
        case SMOOTHER_JACOBI: for (int i = 0; i < steps; ++i) bsr_jacobi(Lv); break;
        case SMOOTHER_GS: for (int i = 0; i < steps && st >= 0; ++i) st = bsr_seq_sweep(h, level, post, false, 0.0); break;
        case SMOOTHER_SGS:
            for (int i = 0; i < steps && st >= 0; ++i) {
                st = bsr_seq_sweep(h, level, false, false, 0.0);
                if (st >= 0) st = bsr_seq_sweep(h, level, true, false, 0.0);
            }
            break;
        case SMOOTHER_SOR: for (int i = 0; i < steps && st >= 0; ++i) st = bsr_seq_sweep(h, level, post, true, relax); break;
        case SMOOTHER_SSOR:  // `steps` ascending sweeps, then ONE descending sweep -- before and after the coarse correction
            for (int i = 0; i < steps && st >= 0; ++i) st = bsr_seq_sweep(h, level, false, true, relax);
            if (st >= 0) st = bsr_seq_sweep(h, level, true, true, relax);
            break;
        default: return ERROR_AMG_SMOOTH_TYPE;
    }
    return st;
}

static KOps bsr_ops(fasp_hip_amg_bsr* h, int level, int set);

// ghost entries of a vector of a row-partitioned block level from their owners (hierarchy.hip.h, halo_exchange: the same
// pack + grouped send / receive, lists expanded to scalars; every rank of a distributed level enters it)
static int bsr_halo(BsrLevel& Lv, double* v);
static int bsr_halo_fwd(BsrLevel& Lv, double* v) { return bsr_halo(Lv, v); }
static int bsr_halo(BsrLevel& Lv, double* v)
{
    if (Lv.replicated || comm_size() <= 1 || Lv.send_off.empty()) return FASP_SUCCESS;
    const int P = comm_size(), me = comm_rank();
    const int nsend = Lv.send_off.back();
    if (nsend > 0)
        hipLaunchKernelGGL(k_pack, dim3(vec_grid(nsend)), dim3(BLOCK), 0, g_ctx.stream, nsend, Lv.d_send_idx, v, Lv.d_sendbuf);
    std::vector<CommXfer> sends, recvs;
    for (int q = 0; q < P; ++q) {
        if (q == me) continue;
        const int ns = Lv.send_off[q + 1] - Lv.send_off[q], nr = Lv.recv_off[q + 1] - Lv.recv_off[q];
        if (ns > 0) sends.push_back({q, Lv.d_sendbuf + Lv.send_off[q], (size_t)ns});
        if (nr > 0) recvs.push_back({q, v + Lv.n + Lv.recv_off[q], (size_t)nr});
    }
    const int st = comm_exchange(sends.data(), (int)sends.size(), recvs.data(), (int)recvs.size(), g_ctx.stream);
    if (st < 0) comm_mark_failed();
    return st;
}

// fasp_solver_mgcycle_bsr, PreMGCycle.c:287-566
static int mgcycle_bsr(fasp_hip_amg_bsr* h, const AMG_param& param)
{
    const int nl = (int)h->L.size(), cycle_type = param.cycle_type, steps = param.presmooth_iter;
    int nu_l[MAX_AMG_LVL + 1] = {0}, l = 0;
    hipStream_t s = g_ctx.stream;
    ++h->vcycles;
ForwardSweep:
    while (l < nl - 1) {
        BsrLevel& Lv = h->L[l];
        ++nu_l[l];
        { const int st = bsr_smooth(h, l, false, param.smoother, steps, param.relaxation); if (st < 0) return st; }
        if (Lv.x_zero) { HIPCK(hipMemsetAsync(Lv.x, 0, sizeof(double) * Lv.nv, s)); Lv.x_zero = false; }
        if (bsr_halo(Lv, Lv.x) < 0) return ERROR_MISC;
        bsr_resid(*Lv.A, Lv.x, Lv.b, Lv.w);
        if (bsr_halo(Lv, Lv.w) < 0) return ERROR_MISC;       // R reads this level's residual, ghosts included
        {
            BsrLevel& C = h->L[l + 1];
            if (!Lv.replicated && C.replicated) {
                // first replicated level: every rank restricts onto the coarse block rows it owns, one all-gather
                // assembles the whole right-hand side on every rank
                const std::vector<int>& cs = h->dist.L[(size_t)l + 1].start;
                const int nb = Lv.A->nb, P = comm_size(), me = comm_rank();
                std::vector<int> counts((size_t)P), displs((size_t)P);
                for (int q = 0; q < P; ++q) { counts[(size_t)q] = (cs[(size_t)q + 1] - cs[(size_t)q]) * nb; displs[(size_t)q] = cs[(size_t)q] * nb; }
                bsr_mxv(*Lv.R, Lv.w, C.b + displs[(size_t)me]);
                if (comm_allgatherv(C.b + displs[(size_t)me], counts[(size_t)me], C.b, counts.data(), displs.data(), s) < 0) return ERROR_MISC;
            } else {
                bsr_mxv(*Lv.R, Lv.w, C.b);
            }
        }
        ++l;
        h->L[l].x_zero = true;  // fasp_dvec_set(.., 0.0), materialised lazily
    }
    {   // coarsest level: fasp_solver_dbsr_pvgmres(A, b, x, NULL, tol, tol*1e-8, min(n^2,200), 25, 1, 0), :443-459
        BsrLevel& Lc = h->L[nl - 1];
        if (!Lc.replicated) return ERROR_INPUT_PAR;   // (the plan keeps the coarsest level whole on every rank)
        if (Lc.x_zero) { HIPCK(hipMemsetAsync(Lc.x, 0, sizeof(double) * Lc.n, s)); Lc.x_zero = false; }
        const int csize = Lc.n;
        const int cmaxit = (int)std::min<unsigned>((unsigned)csize * (unsigned)csize, 200u);
        const double ctol = param.tol, atol = ctol * 1e-8;
        int st;
        const TmpBSR& Ac = *Lc.A;
        if (small_coarse_ok(csize, (long long)Ac.NNZ * Ac.nb * Ac.nb)) {
            if (!h->small_ws) HIPCK(hipMalloc(&h->small_ws, sizeof(double) * (size_t)(25 + 2) * std::max(csize, 1)));
            GmresArgs<SmallBSR> a{};
            a.A = SmallBSR{Ac.ROW, Ac.nb, Ac.ia, Ac.ja, Ac.val};
            a.b = Lc.b; a.x = Lc.x; a.ws = h->small_ws; a.tol = ctol; a.abstol = atol;
            a.MaxIt = cmaxit; a.restart = 25; a.out = small_out_dev();
            size_t lds = sizeof(double) * (size_t)(25 + 2) * (size_t)csize;
            // rows beyond the first 512 of an nb = 3 system keep their blocks in LDS behind the basis (k_gmres_small)
            const size_t lds2 = (size_t)std::max(csize - SMALL_BLOCK, 0) * GM_CB * 28;
            if (Ac.nb == 3 && csize <= 2 * SMALL_BLOCK && lds + lds2 <= 140 * 1024) { a.cache2 = 1; lds += lds2; }
            static bool attr = false;
            if (!attr) {
                (void)hipFuncSetAttribute((const void*)k_gmres_small<SmallBSR, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
                attr = true;
            }
            if (g_tune.small_lds && lds <= 140 * 1024)
                hipLaunchKernelGGL((k_gmres_small<SmallBSR, true>), dim3(1), dim3(SMALL_BLOCK), lds, s, a);
            else
                hipLaunchKernelGGL((k_gmres_small<SmallBSR, false>), dim3(1), dim3(SMALL_BLOCK), 0, s, a);
            SmallOut o;
            if (small_out_fetch(o) < 0) return ERROR_MISC;
            st = o.status;
            h->coarse_iters += o.iters;
        } else {
            KOps K = bsr_ops(h, nl - 1, 1);
            PcgOut po{BIGREAL, BIGREAL, BIGREAL};
            st = gmres_device(K, Lc.b, Lc.x, 0, ctol, atol, cmaxit, 25, STOP_REL_RES, 0, nullptr, &po);
            if (st >= 0) h->coarse_iters += st;
        }
        if (st < 0 && st != ERROR_SOLVER_MAXIT && st != ERROR_SOLVER_STAG && st != ERROR_SOLVER_SOLSTAG &&
            st != ERROR_SOLVER_TOLSMALL) return st;  // device failure, not a convergence verdict
        if (st < 0 && param.print_level > PRINT_MIN) {
            std::printf("### WARNING: Coarse level solver did not converge!\n");
            std::printf("### WARNING: Consider to increase maxit to %d!\n", 2 * cmaxit);
        }
    }
    while (l > 0) {
        --l;
        BsrLevel& Lv = h->L[l];
        {   // x_l += P x_{l+1}  (fasp_blas_dbsr_aAxpy with alpha = 1); a distributed coarse level: its ghosts first
            if (bsr_halo(h->L[l + 1], h->L[l + 1].x) < 0) return ERROR_MISC;
            BsrArgs a{}; a.x = h->L[l + 1].x; a.y = Lv.x; a.alpha = 1.0;
            launch_bsr<1>(*Lv.P, a);
        }
        // the reference post-smooths `steps` = presmooth_iter times (:543)
        { const int st = bsr_smooth(h, l, true, param.smoother, steps, param.relaxation); if (st < 0) return st; }
        if (nu_l[l] < cycle_type) break;
        nu_l[l] = 0;
    }
    if (l > 0) goto ForwardSweep;
    return FASP_SUCCESS;
}

// fasp_precond_dbsr_amg, PreBSR.c:1149: z = (maxit cycles from a zero guess)(r); the AMG_param
// handed to the cycle is re-initialised (tol stays 1e-6) apart from the copied fields
static int precond_amg_bsr(fasp_hip_amg_bsr* h, double* r, double** z)
{
    AMG_param p;
    fasp_param_amg_init(&p);
    const AMG_param& u = h->param;
    p.cycle_type = u.cycle_type; p.smoother = u.smoother; p.presmooth_iter = u.presmooth_iter;
    p.postsmooth_iter = u.postsmooth_iter; p.relaxation = u.relaxation;
    p.coarse_scaling = u.coarse_scaling; p.tentative_smooth = u.tentative_smooth;
    BsrLevel& L0 = h->L[0];
    double* saved_b = L0.b;
    L0.b = r;  // level-0 rhs aliases the Krylov residual (the cycle never writes b_0)
    L0.x_zero = true;
    int st = FASP_SUCCESS;
    for (int i = u.maxit; i--;)
        if ((st = mgcycle_bsr(h, p)) < 0) break;
    L0.b = saved_b;
    if (L0.x_zero) { HIPCK(hipMemsetAsync(L0.x, 0, sizeof(double) * L0.nv, g_ctx.stream)); L0.x_zero = false; }
    *z = L0.x;
    return st;
}

static KOps bsr_ops(fasp_hip_amg_bsr* h, int level, int set)
{
    KOps K;
    BsrLevel* Lv = &h->L[level];
    K.n = Lv->n; K.nvec = (size_t)Lv->nv; K.fmt = "BSR"; K.dist = (level == 0) && h->distributed;
    K.halo = [Lv](double* v) { return bsr_halo(*Lv, v); };
    K.mxv = [Lv](const double* x, double* y) { bsr_mxv(*Lv->A, x, y); };
    K.resid = [Lv](const double* x, const double* b, double* r) { bsr_resid(*Lv->A, x, b, r); };
    if (set == 0) K.pc = [h](double* in, double** out) { return precond_amg_bsr(h, in, out); };
    K.ws = &h->gm[set]; K.ws_len = &h->gm_len[set]; K.hh = &h->gm_hh;
    K.stats = nullptr;
    return K;
}

}  // namespace fasp_bsr


