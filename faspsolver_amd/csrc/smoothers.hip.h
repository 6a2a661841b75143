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
    if (S.graph_exec) { (void)hipGraphExecDestroy(S.graph_exec); S.graph_exec = nullptr; }
    if (S.d_order) { (void)hipFree(S.d_order); S.d_order = nullptr; }
    if (S.d_ptr) { (void)hipFree(S.d_ptr); S.d_ptr = nullptr; }
    HIPCK(hipMalloc(&S.d_order, sizeof(int) * std::max<size_t>(order.size(), 1)));
    if (!order.empty()) HIPCK(hipMemcpy(S.d_order, order.data(), sizeof(int) * order.size(), hipMemcpyHostToDevice));
    S.built = true;
    S.multicolor = multicolor;
    return FASP_SUCCESS;
}

// error word of the persistent sweep kernel: read back where the solve synchronises anyway
static unsigned* g_seq_sync = nullptr;
static unsigned* g_seq_herr = nullptr;
static int seq_persist_check()
{
    if (!g_seq_sync) return FASP_SUCCESS;
    HIPCK(hipMemcpyAsync(g_seq_herr, g_seq_sync + 3, sizeof(unsigned), hipMemcpyDeviceToHost, g_ctx.stream));
    HIPCK(hipStreamSynchronize(g_ctx.stream));
    if (*g_seq_herr) {
        std::fprintf(stderr, "### ERROR: fasp_hip: the persistent sweep kernel timed out at a grid meeting (a block was not resident); "
                             "fasp_hip_tune(\"seq_persist\", 0) selects one launch per dependency level\n");
        return ERROR_MISC;
    }
    return FASP_SUCCESS;
}

// one sequential sweep of schedule `kind` with update formula `form` (see k_seq_level)
static int seq_sweep(fasp_hip_amg* h, int level, int kind, int form, double w)
{
    DevLevel& D = h->L[level];
    DevLevel::Sched& S = D.sched[kind];
    // Multicolour mode -- a FLAGGED NON-PARITY mode for speed: the sweep visits the rows colour by colour instead of
    // in the reference's index order, which is a different (equally convergent, deterministic) Gauss-Seidel / SOR
    // iteration; iteration counts and residuals then differ from the reference's.  The default is the level-scheduled
    // sweep, which reproduces the reference's sequential sweep exactly.
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
        const int st = build_schedule(A, seq, S, multicolor);
        if (st < 0) return st;
        if (std::getenv("FASP_HIP_SETUP_TIMING"))
            std::printf("  [sweep schedule] level %d, sweep kind %d, %s: %d rows in %d classes\n", level, kind,
                        multicolor ? "colours" : "dependency levels", (int)seq.size(), (int)S.ptr.size() - 1);
    }
    materialise_zero(D);
    // lanes per row: a class of a sequential sweep holds few rows (on the deep levels one to a few dozen), so a long row
    // gets a whole wavefront whatever the SpMV kernel of the level uses (fasp_hip_tune("seq_lanes", L) overrides)
    const double avg_len = D.A.row > 0 ? (double)D.A.nnz / D.A.row : 0.0;
    const int L = g_tune.seq_lanes > 0 ? g_tune.seq_lanes : (avg_len >= 96.0 ? 64 : D.A.lanes);
    const int nlev = (int)S.ptr.size() - 1;
    // Optional (fasp_hip_tune("seq_persist", 1)): one launch per sweep (k_seq_sweep), the dependency levels separated by
    // grid meetings instead of kernel boundaries.  MEASURED SLOWER than one launch per level (P7(128), GS-CF defaults:
    // 922 ms against 852 ms per solve; profiles/r02_gs_persistent_sweep.txt): a meeting costs what a launch boundary
    // costs (~3 us: drained write-through stores + arrival + poll), and the time of a sweep is the DEPTH of the
    // dependency DAG on the dense coarse levels (thousands of levels of one to three rows), not the launch count.
    // Needs every block resident at once, so never when validation ranks share the device.
    // Schedules of many small classes (the deep, dense levels -- in either mode): the whole sweep in one workgroup
    // (k_seq_block, kernels2.hip.h), a barrier and one memory round trip per class instead of a launch.
    if (g_tune.seq_block && !multicolor && nlev >= 8 && (long long)S.ptr[nlev] <= (long long)nlev * (2 * SEQ_BLOCK / L)) {
        if (!S.d_ptr) {
            HIPCK(hipMalloc(&S.d_ptr, sizeof(int) * (size_t)(nlev + 1)));
            HIPCK(hipMemcpy(S.d_ptr, S.ptr.data(), sizeof(int) * (size_t)(nlev + 1), hipMemcpyHostToDevice));
        }
        SeqSweepArgs sa{};
        sa.order = S.d_order; sa.lptr = S.d_ptr; sa.nlev = nlev; sa.ia = D.A.ia; sa.ja = D.A.ja; sa.val = D.A.val;
        sa.b = D.b; sa.diag = D.diag; sa.u = D.x; sa.form = form; sa.w = w; sa.sync = nullptr;
        const int   nrow = D.A.row;
        const bool  ulds = g_tune.seq_ulds && (size_t)nrow * 8 <= 150 * 1024;   // u of the level in the workgroup's LDS
        const size_t dyn = ulds ? (size_t)nrow * 8 : 0;
#define SEQB_LAUNCH(LL)                                                                                                     \
        if (ulds) {                                                                                                         \
            static bool attr_set = false;                                                                                   \
            if (!attr_set) {                                                                                                \
                HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_seq_block<LL, true>),                             \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));                         \
                attr_set = true;                                                                                            \
            }                                                                                                               \
            hipLaunchKernelGGL((k_seq_block<LL, true>), dim3(1), dim3(SEQ_BLOCK), dyn, g_ctx.stream, sa, nrow);             \
        } else hipLaunchKernelGGL((k_seq_block<LL, false>), dim3(1), dim3(SEQ_BLOCK), 0, g_ctx.stream, sa, nrow)
        switch (L) {
            case 2: SEQB_LAUNCH(2); break;
            case 4: SEQB_LAUNCH(4); break;
            case 8: SEQB_LAUNCH(8); break;
            case 16: SEQB_LAUNCH(16); break;
            case 32: SEQB_LAUNCH(32); break;
            default: SEQB_LAUNCH(64); break;
        }
#undef SEQB_LAUNCH
        return FASP_SUCCESS;
    }
    static bool seq_persist_disabled = false;
    if (g_tune.seq_persist && !seq_persist_disabled && !comm_shares_devices() && nlev >= 4) {
        if (!S.d_ptr) {
            HIPCK(hipMalloc(&S.d_ptr, sizeof(int) * (size_t)(nlev + 1)));
            HIPCK(hipMemcpy(S.d_ptr, S.ptr.data(), sizeof(int) * (size_t)(nlev + 1), hipMemcpyHostToDevice));
        }
        static unsigned* d_sync = nullptr;
        static unsigned* h_err = nullptr;
        if (!d_sync) { HIPCK(hipMalloc(&d_sync, 1024)); HIPCK(hipHostMalloc((void**)&h_err, 64, hipHostMallocDefault)); }
        HIPCK(hipMemsetAsync(d_sync, 0, 1024, g_ctx.stream));
        SeqSweepArgs sa{};
        sa.order = S.d_order; sa.lptr = S.d_ptr; sa.nlev = nlev; sa.ia = D.A.ia; sa.ja = D.A.ja; sa.val = D.A.val;
        sa.b = D.b; sa.diag = D.diag; sa.u = D.x; sa.form = form; sa.w = w; sa.sync = d_sync;
        // two 256-thread blocks per CU: far below any residency limit of this small kernel, enough waves to hide the gathers
        int widest = 0;
        for (int l = 0; l < nlev; ++l) widest = std::max(widest, S.ptr[l + 1] - S.ptr[l]);
        const int rpb = BLOCK / L;
        const int grid = std::max(8, std::min(2 * g_ctx.num_cu, (widest + rpb - 1) / rpb));
#define SEQP_LAUNCH(LL) hipLaunchKernelGGL((k_seq_sweep<LL>), dim3(grid), dim3(BLOCK), 0, g_ctx.stream, sa)
        switch (L) {
            case 2: SEQP_LAUNCH(2); break;
            case 4: SEQP_LAUNCH(4); break;
            case 8: SEQP_LAUNCH(8); break;
            case 16: SEQP_LAUNCH(16); break;
            case 32: SEQP_LAUNCH(32); break;
            default: SEQP_LAUNCH(64); break;
        }
#undef SEQP_LAUNCH
        // the error word is checked lazily (no synchronisation per sweep): once per solve by seq_persist_check()
        g_seq_sync = d_sync; g_seq_herr = h_err;
        return FASP_SUCCESS;
    }
    // One launch per class.  Optional (fasp_hip_tune("seq_graph", 1)): the sweep's launches captured once as a HIP graph per
    // (schedule, vectors, update form) and replayed.  MEASURED: no gain (P7(128) GS, 348 ms with and without) -- the
    // ~4.4 us per class are the GPU's dispatch of a dependent kernel, not host launch overhead -- so it is off by default.
    const bool use_graph = g_tune.seq_graph && nlev >= 16 && !comm_shares_devices();
    if (use_graph && S.graph_exec && S.g_b == D.b && S.g_x == D.x && S.g_form == form && S.g_w == w && S.g_L == L)
        return hipGraphLaunch(S.graph_exec, g_ctx.stream) == hipSuccess ? FASP_SUCCESS : ERROR_MISC;
    bool capturing = false;
    if (use_graph) {
        if (S.graph_exec) { (void)hipGraphExecDestroy(S.graph_exec); S.graph_exec = nullptr; }
        capturing = hipStreamBeginCapture(g_ctx.stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
    }
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
    if (capturing) {
        hipGraph_t g = nullptr;
        if (hipStreamEndCapture(g_ctx.stream, &g) != hipSuccess || !g) return ERROR_MISC;
        const hipError_t e = hipGraphInstantiate(&S.graph_exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) { S.graph_exec = nullptr; return ERROR_MISC; }
        S.g_b = D.b; S.g_x = D.x; S.g_form = form; S.g_w = w; S.g_L = L;
        return hipGraphLaunch(S.graph_exec, g_ctx.stream) == hipSuccess ? FASP_SUCCESS : ERROR_MISC;
    }
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

