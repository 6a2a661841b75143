// pcg.hip.h -- preconditioned CG on device vectors; CG as a smoother.
// Part of the single translation unit solver.hip (included there, in this order; not a stand-alone header).

// ---------------------------------------------------------------------------
// preconditioned CG (KryPcg.c:96-362) on device vectors
// ---------------------------------------------------------------------------
static void itinfo(int ptrlvl, int stop_type, int iter, double relres, double absres, double factor)
{  // AuxMessage.c:41-71
    if (ptrlvl < PRINT_SOME) return;
    if (iter > 0) {
        std::printf("%6d | %13.6e   | %13.6e  | %10.4f\n", iter, relres, absres, factor);
    } else {
        std::printf("-----------------------------------------------------------\n");
        switch (stop_type) {
            case STOP_REL_RES: std::printf("It Num |   ||r||/||b||   |     ||r||      |  Conv. Factor\n"); break;
            case STOP_REL_PRECRES: std::printf("It Num | ||r||_B/||b||_B |    ||r||_B     |  Conv. Factor\n"); break;
            case STOP_MOD_REL_RES: std::printf("It Num |   ||r||/||x||   |     ||r||      |  Conv. Factor\n"); break;
        }
        std::printf("-----------------------------------------------------------\n");
        std::printf("%6d | %13.6e   | %13.6e  |     -.-- \n", iter, relres, absres);
    }
}

static bool g_pcg_nested = false;   // a top-level pcg_device is running with its (z, r) on the device (slots 12, 13 of d_red)
struct PcgVecs { const double* b; double *u, *p, *t, *r; };
static int pcg_device(KOps& K, const PcgVecs& V, double tol, double abstol, int MaxIt, int StopType,
                      int PrtLvl, Hist& hist, PcgOut& out)
{
    const int m = K.n;         // owned rows
    const bool dist = K.dist;  // reductions are all-reduced over the ranks
    fasp_hip_amg* h = K.stats;
    const double maxdiff = tol * STAG_RATIO, sol_inf_tol = SMALLREAL;
    int iter = 0, stag = 1, more_step = 1;
    double absres0 = BIGREAL, absres = BIGREAL, relres = BIGREAL, normu = BIGREAL, normr0 = BIGREAL;
    double reldiff, factor, alpha = 0.0, beta, temp1 = 0.0, temp2, red[8];
    double *p = V.p, *r = V.r, *t = V.t, *u = V.u, *z = nullptr;
    const double* b = V.b;
    hipStream_t s = g_ctx.stream;
    const int G = vec_grid(m);
    int st;
    // (z, r) on the device -- see below; declared here: the early exits jump over the place where it is decided
    constexpr int ZR_SLOT = 12;   // slots 12, 13 of d_red
    bool zr_dev = false;
    int  zr_cur = 0;              // slot ZR_SLOT + zr_cur holds temp1
    bool zr_pending = false;      // the host's temp1 is one iteration old: the newest value comes with the next round trip
    int  tp_G = 0;                // > 0: (t,p) is still this many per-block partials when k_cg_update starts
    struct NestGuard { bool on = false; void arm() { on = true; g_pcg_nested = true; } ~NestGuard() { if (on) g_pcg_nested = false; } } nest_guard;

    auto apply_pc = [&]() -> int {
        if (K.pc) return K.pc(r, &z);
        z = r;
        return FASP_SUCCESS;
    };
    // residual norm per stop type (KryPcg.c:186-203 and the three copies below it)
    auto resnorm = [&](double rr_known, bool have_rr) -> int {
        switch (StopType) {
            case STOP_REL_RES:
                if (!have_rr) { if (d_dot(m, r, r, red, dist) < 0) return ERROR_MISC; rr_known = red[0]; }
                absres = std::sqrt(rr_known);
                relres = absres / normr0;
                break;
            case STOP_REL_PRECRES:
                if ((st = apply_pc()) < 0) return st;
                if (d_dot(m, z, r, red, dist) < 0) return ERROR_MISC;
                absres = std::sqrt(std::fabs(red[0]));
                relres = absres / normr0;
                break;
            case STOP_MOD_REL_RES:
                if (!have_rr) { if (d_dot(m, r, r, red, dist) < 0) return ERROR_MISC; rr_known = red[0]; }
                absres = std::sqrt(rr_known);
                relres = absres / normu;
                break;
        }
        return FASP_SUCCESS;
    };

    if (PrtLvl > PRINT_NONE) std::printf("\nCalling CG solver (%s) ...\n", K.fmt);

    { if (K.halo(u) < 0) return ERROR_MISC; K.resid(u, b, r); }  // r = b - A u
    if ((st = apply_pc()) < 0) return st;
    switch (StopType) {
        case STOP_REL_RES:
            if (d_dot(m, r, r, red, dist) < 0) return ERROR_MISC;
            absres0 = std::sqrt(red[0]);
            normr0  = std::max(SMALLREAL, absres0);
            relres  = absres0 / normr0;
            break;
        case STOP_REL_PRECRES:
            if (d_dot(m, r, z, red, dist) < 0) return ERROR_MISC;
            absres0 = std::sqrt(red[0]);
            normr0  = std::max(SMALLREAL, absres0);
            relres  = absres0 / normr0;
            break;
        case STOP_MOD_REL_RES:
            if (d_dot(m, r, r, red, dist) < 0) return ERROR_MISC;
            absres0 = std::sqrt(red[0]);
            if (d_dot(m, u, u, red, dist) < 0) return ERROR_MISC;
            normu  = std::max(SMALLREAL, std::sqrt(red[0]));
            relres = absres0 / normu;
            break;
        default:
            std::printf("### ERROR: Unknown stopping type! [%s]\n", "fasp_solver_dcsr_pcg");
            goto FINISHED;
    }
    hist.push(absres0);
    if (relres < tol || absres0 < abstol) goto FINISHED;

    itinfo(PrtLvl, StopType, iter, relres, absres0, 0.0);
    HIPCK(hipMemcpyAsync(p, z, sizeof(double) * m, hipMemcpyDeviceToDevice, s));
    if (d_dot(m, z, r, red, dist) < 0) return ERROR_MISC;
    temp1 = red[0];
    // (z, r) stays on the device (round 5): the top-level solve keeps the value of this and of the previous iteration in two slots of
    // d_red in turn; beta = temp2 / temp1 and alpha = temp1 / (t, p) are formed there (one IEEE division each, as on the host), and the
    // host reads (z, r) with the scalars of the NEXT iteration's single round trip -- one wait per iteration instead of two.  Not for a
    // nested solve (CG as a smoother: the outer solve's slots would be overwritten) nor when (z, r) is the stopping quantity.
    zr_dev = g_tune.pcg_dev_beta && h != nullptr && StopType != STOP_REL_PRECRES && !g_pcg_nested;
    if (zr_dev) {
        HIPCK(hipMemcpyAsync(g_ctx.d_red + ZR_SLOT, g_ctx.d_red, sizeof(double), hipMemcpyDeviceToDevice, s));
        nest_guard.arm();
    }

    while (iter++ < MaxIt) {
        // t = A p with the partial sums of (t,p); timed for the roofline report
        {
            if (K.halo(p) < 0) return ERROR_MISC;
            // (every fourth launch: an event pair costs the stream ~12 us of idle time around the launch it brackets)
            EventPair* ep = (h && (g_tune.ev_every <= 1 || (iter % g_tune.ev_every) == 2 % g_tune.ev_every) && h->ev_used < (int)h->ev.size()) ? &h->ev[h->ev_used++] : nullptr;
            if (ep) (void)hipEventRecord(ep->a, s);
            int gdot = K.mxv_dot ? K.mxv_dot(p, t) : -1;
            if (ep) (void)hipEventRecord(ep->b, s);
            tp_G = 0;
            if (gdot >= 0 && !(dist && comm_size() > 1) && g_tune.pcg_fold) tp_G = gdot;   // one rank: k_cg_update sums the partials itself (k_finalize's order) -- one launch less
            else if (gdot >= 0) d_finalize(gdot, 1, 0u, 8, dist);
            else {  // format without a fused kernel: t = A p, then (t,p) into slot 8
                K.mxv(p, t);
                if (d_dot_to(m, t, p, 8, dist) < 0) return ERROR_MISC;
            }
        }
        // alpha = temp1/(t,p) on device; u += alpha p; r -= alpha t; partial ||r||^2
        // (+ the preconditioner's first Jacobi sweep of the new r, when it has one: K.pre_x)
        double* const pre_x = (K.pre_x && StopType != STOP_REL_PRECRES) ? K.pre_x() : nullptr;
        // (tp_G > 0: the (t,p) partials are read from d_partials by every block, so this kernel's own partials go to the second set)
        double* const upd_partials = tp_G > 0 ? g_ctx.d_partials2 : g_ctx.d_partials;
        hipLaunchKernelGGL(k_cg_update, dim3(G), dim3(BLOCK), 0, s, m, temp1, (const double*)(g_ctx.d_red + 8),
                           (const double*)(tp_G > 0 ? g_ctx.d_partials : nullptr), tp_G, p, t, u, r, upd_partials, 0,
                           tp_G > 0 ? g_ctx.d_red + 8 : (double*)nullptr,
                           pre_x, K.pre_diag_uniform ? (const double*)nullptr : K.pre_diag, K.pre_omega,
                           zr_dev ? (const double*)(g_ctx.d_red + ZR_SLOT + zr_cur) : (const double*)nullptr, K.pre_diag_value);
        bool r_is_updates = pre_x != nullptr;   // false again as soon as r is recomputed from u
        if (tp_G > 0) hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLOCK), 0, s, (const double*)g_ctx.d_partials2, G, 1, 0u, g_ctx.d_red);
        else d_finalize(G, 1, 0u, 0, dist);
        HIPCK(hipMemcpyAsync(g_ctx.h_red, g_ctx.d_red, sizeof(double) * (zr_dev ? ZR_SLOT + 2 : 9), hipMemcpyDeviceToHost, s));
        HIPCK(hipStreamSynchronize(s));
        if (zr_pending) { temp1 = g_ctx.h_red[ZR_SLOT + zr_cur]; zr_pending = false; }   // (the value cg_update has just divided)
        temp2 = g_ctx.h_red[8];
        if (std::fabs(temp2) > SMALLREAL2) {
            alpha = temp1 / temp2;
        } else {
            std::printf("### WARNING: Divided by zero! [%s:%d]\n", "fasp_solver_dcsr_pcg", 175);
            goto FINISHED;
        }
        if ((st = resnorm(g_ctx.h_red[0], true)) < 0) return st;
        factor = absres / absres0;
        itinfo(PrtLvl, StopType, iter, relres, absres, factor);
        hist.push(absres);

        if (factor > 0.9) {  // Check I / II, only when converging slowly
            if (d_norms(m, u, red, dist) < 0) return ERROR_MISC;
            if (red[1] <= sol_inf_tol) {
                if (PrtLvl > PRINT_MIN)
                    std::printf("### WARNING: Iteration stopped -- solution almost zero! [%s:%d]\n",
                                "fasp_solver_dcsr_pcg", 218);
                iter = ERROR_SOLVER_SOLSTAG;
                break;
            }
            normu = std::sqrt(red[0]);
            if (d_dot(m, p, p, red, dist) < 0) return ERROR_MISC;
            reldiff = std::fabs(alpha) * std::sqrt(red[0]) / normu;
            if ((stag <= MAX_STAG) & (reldiff < maxdiff)) {
                if (PrtLvl >= PRINT_MORE) {
                    std::printf("||u-u'|| = %.10e and the comp. rel. res. = %.10e.\n", reldiff, relres);
                    std::printf("### WARNING: Iteration restarted -- stagnation! [%s:%d]\n",
                                "fasp_solver_dcsr_pcg", 232);
                }
                { if (K.halo(u) < 0) return ERROR_MISC; K.resid(u, b, r); }
                r_is_updates = false;
                if ((st = resnorm(0.0, false)) < 0) return st;
                if (PrtLvl >= PRINT_MORE)
                    std::printf("### WARNING: The actual relative residual = %.10e!\n", relres);
                if (relres < tol) break;
                if (stag >= MAX_STAG) {
                    if (PrtLvl > PRINT_MIN)
                        std::printf("### WARNING: Iteration stopped -- staggnation! [%s:%d]\n",
                                    "fasp_solver_dcsr_pcg", 266);
                    iter = ERROR_SOLVER_STAG;
                    break;
                }
                HIPCK(hipMemsetAsync(p, 0, sizeof(double) * m, s));
                ++stag;
            }
        }

        if (relres < tol) {  // Check III: prevent false convergence
            const double updated_relres = relres;
            { if (K.halo(u) < 0) return ERROR_MISC; K.resid(u, b, r); }
            r_is_updates = false;
            if ((st = resnorm(0.0, false)) < 0) return st;
            if (relres < tol) break;
            if (PrtLvl >= PRINT_MORE) {
                std::printf("### WARNING: The computed relative residual = %.10e!\n", updated_relres);
                std::printf("### WARNING: The actual relative residual = %.10e!\n", relres);
            }
            if (more_step >= MAX_RESTART) {
                if (PrtLvl > PRINT_MIN)
                    std::printf("### WARNING: The tolerence might be too small! [%s:%d]\n",
                                "fasp_solver_dcsr_pcg", 315);
                iter = ERROR_SOLVER_TOLSMALL;
                break;
            }
            HIPCK(hipMemsetAsync(p, 0, sizeof(double) * m, s));
            ++more_step;
        }

        absres0 = absres;
        int zr_G = 0;   // partials of (z, r) left behind by the preconditioner's last sweep (0: none)
        if (StopType != STOP_REL_PRECRES) {
            if (r_is_updates && K.mark_presmoothed) K.mark_presmoothed();
            if (K.pc_zr) { if ((st = K.pc_zr(r, &z, &zr_G)) < 0) return st; }
            else if ((st = apply_pc()) < 0) return st;
        }
        if (zr_dev) {
            // (z, r) into the other slot, beta on the device, no wait: the host learns the value with the next iteration's scalars
            const int nxt = zr_cur ^ 1;
            int fold = 0;   // > 0: k_axpby_beta sums the (z, r) partials itself (k_finalize's order) and block 0 leaves the sum in the slot
            if (zr_G > 0 && !(dist && comm_size() > 1) && g_tune.pcg_fold) fold = zr_G;
            else if (zr_G > 0) d_finalize(zr_G, 1, 0u, ZR_SLOT + nxt, dist);
            else if (d_dot_to(m, z, r, ZR_SLOT + nxt, dist) < 0) return ERROR_MISC;
            hipLaunchKernelGGL(k_axpby_beta, dim3(vec_grid(m / 2 + 1)), dim3(BLOCK), 0, s, m, (const double*)z,
                               (const double*)(g_ctx.d_red + ZR_SLOT + nxt), (const double*)(g_ctx.d_red + ZR_SLOT + zr_cur), p,
                               (const double*)g_ctx.d_partials, fold, g_ctx.d_red + ZR_SLOT + nxt);  // p = z + beta p
            zr_cur = nxt;
            zr_pending = true;
            continue;
        }
        if (zr_G > 0) {
            d_finalize(zr_G, 1, 0u, 0, dist);
            if (fetch_red(0, 1, red) < 0) return ERROR_MISC;
        } else if (d_dot(m, z, r, red, dist) < 0) return ERROR_MISC;
        temp2 = red[0];
        beta  = temp2 / temp1;
        temp1 = temp2;
        d_axpby(m, 1.0, z, beta, p);  // p = z + beta p
    }

FINISHED:
    if (PrtLvl > PRINT_NONE) {  // ITS_FINAL, KryUtil.inl:95-105
        if (iter > MaxIt)
            std::printf("### WARNING: MaxIt = %d reached with relative residual %.10e.\n", MaxIt, relres);
        else if (iter >= 0)
            std::printf("Number of iterations = %d with relative residual %.10e.\n", iter, relres);
    }
    hist.push(absres);  // trailing entry: absres at exit (true residual after Check III)
    out.relres = relres; out.absres = absres; out.normr0 = normr0;
    HIPCK(hipStreamSynchronize(s));
    if (iter > MaxIt) return ERROR_SOLVER_MAXIT;
    return iter;
}

// CG as a smoother: fasp_solver_dcsr_pcg(A, b, x, NULL, 1e-3, 1e-15, nsweeps, STOP_REL_RES, PRINT_NONE),
// PreMGSmoother.inl:116 / :222 -- `nsweeps` CG steps on the level's system; its return code (normally
// "MaxIt reached") is ignored there too
static int cg_smooth(fasp_hip_amg* h, int level, int nsweeps)
{
    DevLevel& D = h->L[level];
    for (int q = 0; q < 3; ++q) if (!D.kw[q]) { if (alloc_vec(&D.kw[q], (size_t)D.nvec) < 0) return ERROR_ALLOC_MEM; }
    materialise_zero(D);
    KOps K = csr_ops(h, level, false);
    K.stats = nullptr;
    PcgVecs V{D.b, D.x, D.kw[0], D.kw[1], D.kw[2]};
    Hist H{nullptr, 0, 0};
    PcgOut po{BIGREAL, BIGREAL, BIGREAL};
    const int st = pcg_device(K, V, 1e-3, 1e-15, nsweeps, STOP_REL_RES, PRINT_NONE, H, po);
    return (st == ERROR_MISC || st == ERROR_ALLOC_MEM) ? st : FASP_SUCCESS;
}

