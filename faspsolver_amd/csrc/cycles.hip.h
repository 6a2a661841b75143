// cycles.hip.h -- multigrid cycles on the resident hierarchy: V / W / hybrid, AMLI, K-cycle, full multigrid; the AMG preconditioner.
// Part of the single translation unit solver.hip (included there, in this order; not a stand-alone header).

// ---------------------------------------------------------------------------
// one multigrid cycle on the resident hierarchy (PreMGCycle.c:48-274)
// ---------------------------------------------------------------------------
// fasp_coarse_itsolver (PreMGUtil.inl:37): safe CG on the coarsest level, SPVGMRES as the safety net
static int coarse_solve(fasp_hip_amg* h, const AMG_param& param, double tol)
{
    const int nl = (int)h->L.size();
    DevLevel& Lc = h->L[nl - 1];
    int st = coarse_spcg(h, Lc, tol, param.print_level);
    if (st == ERROR_MISC) return st;  // device failure, not a solver verdict
    if (st < 0) {
        // safety net of PreMGUtil.inl:50-52: fasp_solver_dcsr_spvgmres(A, b, x, NULL, ctol, maxit, 20, 1, ..)
        const int m = Lc.A.row;
        const int nn = (int)((unsigned)m * (unsigned)m);
        const int maxit = std::max(250, std::min(nn, 1000));
        KOps Kc = csr_ops(h, nl - 1, false);
        st = gmres_device(Kc, Lc.b, Lc.x, 2, tol, 0.0, maxit, 20, STOP_REL_RES, param.print_level - 4,
                          nullptr, nullptr);
        if (st == ERROR_MISC) return st;
        if (st < 0 && param.print_level >= PRINT_MORE) {
            std::printf("### WARNING: Coarse level solver did not converge!\n");
            std::printf("### WARNING: Consider to increase maxit to %d!\n", 2 * maxit);
        }
    }
    return FASP_SUCCESS;
}

// fasp_amg_amli_coef (PreMGRecurAMLI.c:791): coefficients of the degree-`degree` polynomial that
// approximates 1/t on [lambda_min, lambda_max]
static void amli_coef(double lambda_max, double lambda_min, int degree, double* coef)
{
    const double mu0 = 1.0 / lambda_max, mu1 = 1.0 / lambda_min;
    const double c = (std::sqrt(mu0) + std::sqrt(mu1)) * (std::sqrt(mu0) + std::sqrt(mu1));
    const double a = (4 * mu0 * mu1) / (c);
    const double kappa = lambda_max / lambda_min;
    const double delta = (std::sqrt(kappa) - 1.0) / (std::sqrt(kappa) + 1.0);
    const double b = delta * delta;
    if (degree == 0) coef[0] = 0.5 * (mu0 + mu1);
    else if (degree == 1) { coef[0] = 0.5 * c; coef[1] = -1.0 * mu0 * mu1; }
    else if (degree > 1) {
        std::vector<double> work((size_t)2 * degree - 1, 0.0);
        double *coef_k = work.data(), *coef_km1 = work.data() + degree;
        amli_coef(lambda_max, lambda_min, degree - 1, coef_k);
        amli_coef(lambda_max, lambda_min, degree - 2, coef_km1);
        coef[0] = a - b * coef_km1[0] + (1 + b) * coef_k[0];
        for (int i = 1; i < degree - 1; i++) coef[i] = -b * coef_km1[i] + (1 + b) * coef_k[i] - a * coef_k[i - 1];
        coef[degree - 1] = (1 + b) * coef_k[degree - 1] - a * coef_k[degree - 2];
        coef[degree] = -a * coef_k[degree - 1];
    }
}

// fasp_solver_amli (PreMGRecurAMLI.c:58): the coarse-grid correction of every level is a polynomial of
// degree amli_degree in the recursively preconditioned coarse operator (coefficients for the interval
// [0.5, 2], PreAMGSetupRS.c:93-97).  One GPU (AMLI hierarchies are not row-partitioned).
static int amli_cycle(fasp_hip_amg* h, const AMG_param& param, int l)
{
    const int nl = (int)h->L.size(), degree = param.amli_degree;
    hipStream_t s = g_ctx.stream;
    DevLevel& D = h->L[l];
    int st;
    if (l >= nl - 1) return coarse_solve(h, param, param.tol * 1e-4);
    DevLevel& C = h->L[l + 1];
    const int m0 = D.A.row, m1 = C.A.row;
    const double* coef = h->amli_coef.data();
    if (!C.w2) { if (alloc_vec(&C.w2, (size_t)C.nvec) < 0) return ERROR_ALLOC_MEM; }
    double* r1 = C.w2;
    if ((st = smooth(h, l, false, param.smoother, param.smooth_order, param.presmooth_iter, param.relaxation, param.polynomial_degree)) < 0) return st;
    if (D.x_zero) HIPCK(hipMemcpyAsync(D.w, D.b, sizeof(double) * m0, hipMemcpyDeviceToDevice, s));
    else d_resid(D.A, D.x, D.b, D.w);
    d_mxv(D.R, D.w, C.b);
    HIPCK(hipMemcpyAsync(r1, C.b, sizeof(double) * m1, hipMemcpyDeviceToDevice, s));
    for (int i = 1; i <= degree; i++) {
        C.x_zero = true;
        if ((st = amli_cycle(h, param, l + 1)) < 0) return st;
        materialise_zero(C);
        d_mxv(C.A, C.x, C.b);                                           // b1 = A1 e1
        d_axpy(m1, coef[degree - i] / coef[degree], r1, C.b);           // b1 += (q_{degree-i} / q_degree) r1
    }
    C.x_zero = true;
    if ((st = amli_cycle(h, param, l + 1)) < 0) return st;
    materialise_zero(C);
    d_scale(m1, coef[degree], C.x);
    double alpha = 1.0;
    if (param.coarse_scaling == 1) {  // alpha = (e1, r1) / (A1 e1, e1), capped at 1; C.w is free scratch here
        double red[2];
        CsrArgs a{}; a.x = C.x; a.y = C.w; a.dotv = C.x; a.partials = g_ctx.d_partials;
        const int gdot = launch_csr<OP_MXV_DOT>(C.A, a);
        d_finalize(gdot, 1, 0u, 1, false);
        if (fetch_red(1, 1, red + 1) < 0) return ERROR_MISC;
        if (d_dot(m1, C.x, r1, red, false) < 0) return ERROR_MISC;
        alpha = std::min(red[0] / red[1], 1.0);
    }
    materialise_zero(D);
    d_aAxpy(alpha, D.P, C.x, D.x);
    return smooth(h, l, true, param.smoother, param.smooth_order, param.postsmooth_iter, param.relaxation, param.polynomial_degree);
}

// Nonlinear AMLI / K-cycle (fasp_solver_namli, PreMGRecurAMLI.c:291; Kcycle_dcsr_pgcg / _pgcr,
// PreMGRecurAMLI.inl:36 / :139; fasp_precond_namli, PreCSR.c:524): the coarse problem of a level whose
// AMG_data.cycle_type is > 1 is solved by at most two steps of a Krylov method preconditioned by the
// same cycle one level down.  `base` = the level the reference's shifted pointer &mgl[l+1] points at.
static int namli_cycle(fasp_hip_amg* h, const AMG_param& param, int base, int num_levels);

static int namli_precond(fasp_hip_amg* h, const AMG_param& user, int base, int num_levels, const double* r, double* z)
{
    AMG_param p;  // fasp_param_amg_init + fasp_param_prec_to_amg (AuxParam.c:816): tol is not carried over
    fasp_param_amg_init(&p);
    p.AMG_type = user.AMG_type; p.print_level = user.print_level; p.cycle_type = user.cycle_type;
    p.smoother = user.smoother; p.smooth_order = user.smooth_order; p.presmooth_iter = user.presmooth_iter;
    p.postsmooth_iter = user.postsmooth_iter; p.relaxation = user.relaxation;
    p.polynomial_degree = user.polynomial_degree; p.coarse_solver = user.coarse_solver;
    p.coarse_scaling = user.coarse_scaling; p.amli_degree = user.amli_degree;
    p.nl_amli_krylov_type = user.nl_amli_krylov_type; p.tentative_smooth = user.tentative_smooth;
    DevLevel& L = h->L[base];
    const int m = L.A.row;
    HIPCK(hipMemcpyAsync(L.b, r, sizeof(double) * m, hipMemcpyDeviceToDevice, g_ctx.stream));
    L.x_zero = true;
    const int st = namli_cycle(h, p, base, num_levels);
    if (st < 0) return st;
    materialise_zero(L);
    HIPCK(hipMemcpyAsync(z, L.x, sizeof(double) * m, hipMemcpyDeviceToDevice, g_ctx.stream));
    return FASP_SUCCESS;
}

// at most two steps of GCG (gcr == false) or GCR on level `base` (matrix L.A, right-hand side L.b), result in x
static int kcycle(fasp_hip_amg* h, const AMG_param& param, bool gcr, int base, int num_levels, double* x)
{
    DevLevel& L = h->L[base];
    const int m = L.A.row;
    hipStream_t s = g_ctx.stream;
    for (double*& q : L.kw) if (!q) { if (alloc_vec(&q, (size_t)L.nvec) < 0) return ERROR_ALLOC_MEM; }
    double *r = L.kw[0], *x1 = L.kw[1], *v1 = L.kw[2], *v2 = L.kw[3];
    double red[2], normb, absres, relres, alpha1, alpha2, gamma, rho1, rho2;
    auto dot = [&](const double* a, const double* b, double& v) -> int {
        if (d_dot(m, a, b, red, false) < 0) return ERROR_MISC;
        v = red[0];
        return 0;
    };
    int st;
    if ((st = dot(L.b, L.b, normb)) < 0) return st;
    normb = std::sqrt(normb);
    HIPCK(hipMemcpyAsync(r, L.b, sizeof(double) * m, hipMemcpyDeviceToDevice, s));
    if ((st = namli_precond(h, param, base, num_levels, r, x)) < 0) return st;
    d_mxv(L.A, x, v1);
    if (!gcr) {
        if ((st = dot(x, v1, rho1)) < 0 || (st = dot(x, r, alpha1)) < 0) return st;
        const double beta1 = alpha1 / rho1;
        d_axpy(m, -beta1, v1, r);
        if ((st = dot(r, r, absres)) < 0) return st;
        relres = std::sqrt(absres) / normb;
        if (relres < 0.2) { d_scale(m, beta1, x); return FASP_SUCCESS; }
        if ((st = namli_precond(h, param, base, num_levels, r, x1)) < 0) return st;
        d_mxv(L.A, x1, v2);
        if ((st = dot(x1, v1, gamma)) < 0 || (st = dot(x1, r, alpha2)) < 0 || (st = dot(x1, v2, rho2)) < 0) return st;
        const double beta2 = rho2 - gamma * gamma / rho1;
        if (std::fabs(beta2) < SMALLREAL) return FASP_SUCCESS;
        const double beta3 = (alpha1 - gamma * alpha2 / beta2) / rho1, beta4 = alpha2 / beta2;
        d_scale(m, beta3, x);
        d_axpy(m, beta4, x1, x);
    } else {
        double beta;
        if ((st = dot(v1, v1, rho1)) < 0 || (st = dot(v1, r, alpha1)) < 0) return st;
        const double alpha = alpha1 / rho1;
        d_axpy(m, -alpha, v1, r);
        if ((st = dot(r, r, absres)) < 0) return st;
        relres = std::sqrt(absres) / normb;
        if (relres < 0.2) { d_scale(m, alpha, x); return FASP_SUCCESS; }
        if ((st = namli_precond(h, param, base, num_levels, r, x1)) < 0) return st;
        d_mxv(L.A, x1, v2);
        if ((st = dot(v1, v2, gamma)) < 0 || (st = dot(v2, v2, beta)) < 0 || (st = dot(r, v2, alpha2)) < 0) return st;
        rho2 = beta - gamma * gamma / rho1;
        const double alpha3 = alpha1 / rho1 - gamma * alpha2 / (rho1 * rho2), alpha4 = alpha2 / rho2;
        d_scale(m, alpha3, x);
        d_axpy(m, alpha4, x1, x);
    }
    return FASP_SUCCESS;
}

static int namli_cycle(fasp_hip_amg* h, const AMG_param& param, int base, int num_levels)
{
    hipStream_t s = g_ctx.stream;
    DevLevel& D = h->L[base];
    int st;
    if (num_levels <= 1) {  // coarsest level of this sub-hierarchy == coarsest level of the hierarchy
        if (base != (int)h->L.size() - 1) return ERROR_INPUT_PAR;
        return coarse_solve(h, param, param.tol * 1e-4);
    }
    DevLevel& C = h->L[base + 1];
    const int m0 = D.A.row, m1 = C.A.row;
    if ((st = smooth(h, base, false, param.smoother, param.smooth_order, param.presmooth_iter, param.relaxation, param.polynomial_degree)) < 0) return st;
    if (D.x_zero) HIPCK(hipMemcpyAsync(D.w, D.b, sizeof(double) * m0, hipMemcpyDeviceToDevice, s));
    else d_resid(D.A, D.x, D.b, D.w);
    d_mxv(D.R, D.w, C.b);
    const int ct = (base + 1 < (int)h->level_cycle_type.size()) ? h->level_cycle_type[(size_t)base + 1] : 1;
    if (ct <= 1) {  // a V-cycle is enforced on this level
        C.x_zero = true;
        if ((st = namli_cycle(h, param, base + 1, num_levels - 1)) < 0) return st;
        materialise_zero(C);
    } else {
        if (!C.w2) { if (alloc_vec(&C.w2, (size_t)C.nvec) < 0) return ERROR_ALLOC_MEM; }
        double* uH = C.w2;
        HIPCK(hipMemsetAsync(uH, 0, sizeof(double) * m1, s));
        if ((st = kcycle(h, param, param.nl_amli_krylov_type != SOLVER_GCG, base + 1, num_levels - 1, uH)) < 0) return st;
        HIPCK(hipMemcpyAsync(C.x, uH, sizeof(double) * m1, hipMemcpyDeviceToDevice, s));
        C.x_zero = false;
    }
    materialise_zero(D);
    d_aAxpy(1.0, D.P, C.x, D.x);
    return smooth(h, base, true, param.smoother, param.smooth_order, param.postsmooth_iter, param.relaxation, param.polynomial_degree);
}

// fasp_solver_fmgcycle (PreMGCycleFull.c:47): the right-hand side is restricted to every level, the
// coarsest system solved, then level by level the solution is interpolated and improved by up to 3
// V-cycles from that level.  As in the reference the iterate of an intermediate level is not reset
// before the interpolated correction is added: it keeps what the previous call left there.  One GPU.
static int fmg_cycle(fasp_hip_amg* h, const AMG_param& param)
{
    const int nl = (int)h->L.size(), maxit = 3;
    const double tol = param.tol * 1e-4;
    hipStream_t s = g_ctx.stream;
    int st, l;
    if (h->distributed) return ERROR_INPUT_PAR;
    h->vcycles++;
    for (l = 0; l < nl - 1; ++l) d_mxv(h->L[l].R, h->L[l].b, h->L[l + 1].b);
    h->L[l].x_zero = true;
    if (nl == 1) return coarse_solve(h, param, tol);
    auto scaled_prolongation = [&](int lf) -> int {  // x_lf += alpha P x_{lf+1}
        DevLevel& D = h->L[lf];
        DevLevel& C = h->L[lf + 1];
        double alpha = 1.0;
        materialise_zero(C);
        if (param.coarse_scaling == 1) {
            double red[2];
            CsrArgs a{}; a.x = C.x; a.y = C.w; a.dotv = C.x; a.partials = g_ctx.d_partials;
            const int gdot = launch_csr<OP_MXV_DOT>(C.A, a);
            d_finalize(gdot, 1, 0u, 1, false);
            if (fetch_red(1, 1, red + 1) < 0) return ERROR_MISC;
            if (d_dot(C.A.row, C.x, C.b, red, false) < 0) return ERROR_MISC;
            alpha = std::min(red[0] / red[1], 1.0);
        }
        materialise_zero(D);
        d_aAxpy(alpha, D.P, C.x, D.x);
        return FASP_SUCCESS;
    };
    for (int i = 1; i < nl; ++i) {
        if ((st = coarse_solve(h, param, tol)) < 0) return st;
        --l;
        if ((st = scaled_prolongation(l)) < 0) return st;
        int num_cycle = 0;
        double relerr = BIGREAL, red[2];
        while (relerr > param.tol && num_cycle < maxit) {
            ++num_cycle;
            {
                DevLevel& D = h->L[l];
                d_resid(D.A, D.x, D.b, D.w);
                double nw, nb;
                if (d_dot(D.A.row, D.w, D.w, red, false) < 0) return ERROR_MISC;
                nw = std::sqrt(red[0]);
                if (d_dot(D.A.row, D.b, D.b, red, false) < 0) return ERROR_MISC;
                nb = std::sqrt(red[0]);
                relerr = nw / nb;
            }
            for (int lvl = 0; lvl < i; ++lvl) {
                DevLevel& D = h->L[l];
                if ((st = smooth(h, l, false, param.smoother, param.smooth_order, param.presmooth_iter, param.relaxation, param.polynomial_degree)) < 0) return st;
                if (D.x_zero) HIPCK(hipMemcpyAsync(D.w, D.b, sizeof(double) * D.A.row, hipMemcpyDeviceToDevice, s));
                else d_resid(D.A, D.x, D.b, D.w);
                d_mxv(D.R, D.w, h->L[l + 1].b);
                ++l;
                h->L[l].x_zero = true;
            }
            if ((st = coarse_solve(h, param, tol)) < 0) return st;
            for (int lvl = 0; lvl < i; ++lvl) {
                --l;
                if ((st = scaled_prolongation(l)) < 0) return st;
                if ((st = smooth(h, l, true, param.smoother, param.smooth_order, param.postsmooth_iter, param.relaxation, param.polynomial_degree)) < 0) return st;
            }
        }
    }
    return FASP_SUCCESS;
}

static int mgcycle(fasp_hip_amg* h, const AMG_param& param)
{
    const int nl = (int)h->L.size();
    if (param.cycle_type == NL_AMLI_CYCLE) {  // fasp_precond_namli (PreCSR.c:524) / fasp_amg_solve_namli (PreMGSolve.c:230)
        if (h->distributed) return ERROR_INPUT_PAR;
        h->vcycles++;
        return namli_cycle(h, param, 0, nl);
    }
    if (param.cycle_type == AMLI_CYCLE) {  // fasp_precond_amli (PreCSR.c:482) / fasp_amg_solve_amli (PreMGSolve.c:142)
        if (h->distributed || param.amli_degree < 0 || param.amli_degree > 30) return ERROR_INPUT_PAR;
        if ((int)h->amli_coef.size() != param.amli_degree + 1) {
            h->amli_coef.assign((size_t)param.amli_degree + 1, 0.0);
            amli_coef(2.0, 0.5, param.amli_degree, h->amli_coef.data());
        }
        h->vcycles++;
        return amli_cycle(h, param, 0);
    }
    const int smoother = param.smoother, cycle_type = param.cycle_type;
    const double relax = param.relaxation;
    const double tol = param.tol * 1e-4;
    int num_lvl[MAX_AMG_LVL] = {0}, ncycles[MAX_AMG_LVL], l = 0;
    for (int i = 0; i < MAX_AMG_LVL; ++i) ncycles[i] = 1;
    switch (cycle_type) {
        case 12: for (int i = MAX_AMG_LVL - 2; i > 0; i -= 2) ncycles[i] = 2; break;
        case 21: for (int i = MAX_AMG_LVL - 1; i > 0; i -= 2) ncycles[i] = 2; break;
        default: for (int i = 0; i < MAX_AMG_LVL; ++i) ncycles[i] = cycle_type;
    }
    h->vcycles++;
    int st0 = FASP_SUCCESS;

ForwardSweep:
    while (l < nl - 1) {
        DevLevel& D = h->L[l];
        num_lvl[l]++;
        if ((st0 = smooth(h, l, false, smoother, param.smooth_order, param.presmooth_iter, relax, param.polynomial_degree)) < 0) return st0;
        // w = b - A x ; b_{l+1} = R w
        if (D.x_zero) {
            HIPCK(hipMemcpyAsync(D.w, D.b, sizeof(double) * D.A.row, hipMemcpyDeviceToDevice, g_ctx.stream));
        } else {
            CsrArgs a{}; a.x = D.x; a.y = D.w; a.b = D.b;
            if (dist_launch<OP_RESID>(D, D.A, a) < 0) return ERROR_MISC;   // halo of x beside the interior rows
        }
        bool fused_presmooth = false;
        {
            DevLevel& C = h->L[l + 1];
            CsrArgs ra{}; ra.x = D.w;
            // the restriction also writes the first pre-smoothing sweep of the next level (Jacobi from a zero guess:
            // x = (w b) / d, k_jacobi_zero's expression) when that level is smoothed and both levels are laid out alike
            if (g_tune.fuse_presmooth && smoother == SMOOTHER_JACOBI && param.presmooth_iter >= 1 && l + 1 < nl - 1 &&
                D.replicated == C.replicated && !(D.replicated && coarse_split_active(D.R))) {   // (a split restriction gathers b only)
                ra.zx = C.x; ra.zdiag = C.diag; ra.zomega = relax;
                fused_presmooth = true;
            }
            if (!D.replicated && C.replicated) {
                // first replicated level: every rank restricts onto the coarse rows it owns,
                // one all-gather assembles the whole right-hand side on every rank
                const std::vector<int>& cs = h->dist.L[l + 1].start;
                std::vector<int> counts(comm_size());
                for (int q = 0; q < comm_size(); ++q) counts[q] = cs[q + 1] - cs[q];
                ra.y = C.b + cs[comm_rank()];
                if (dist_launch<OP_MXV>(D, D.R, ra) < 0) return ERROR_MISC;
                if (comm_allgatherv(C.b + cs[comm_rank()], counts[comm_rank()], C.b, counts.data(), cs.data(),
                                    g_ctx.stream) < 0) return ERROR_MISC;
            } else {
                ra.y = C.b;
                if (dist_launch<OP_MXV>(D, D.R, ra) < 0) return ERROR_MISC;
            }
        }
        ++l;
        h->L[l].x_zero = true;  // fasp_dvec_set(x_{l}, 0): materialised lazily
        h->L[l].presmoothed = fused_presmooth;
    }

    if ((st0 = coarse_solve(h, param, tol)) < 0) return st0;

    while (l > 0) {
        --l;
        DevLevel& D = h->L[l];
        materialise_zero(D);
        DevLevel& C = h->L[l + 1];
        double alpha = 1.0;
        if (param.coarse_scaling == 1) {
            if (halo_exchange(C, C.x) < 0) return ERROR_MISC;
            // PreMGCycle.c:210-216: alpha = (x_c, b_c) / (A_c x_c, x_c), capped at 1
            // (fasp_blas_dcsr_vmv, BlaSpmvCSR.c:839); C.w is free scratch on the way up
            const bool cdist = !C.replicated && comm_size() > 1;
            double red[2];
            CsrArgs a{}; a.x = C.x; a.y = C.w; a.dotv = C.x; a.partials = g_ctx.d_partials;
            const int gdot = launch_csr<OP_MXV_DOT>(C.A, a);
            d_finalize(gdot, 1, 0u, 1, cdist);
            if (fetch_red(1, 1, red + 1) < 0) return ERROR_MISC;
            if (d_dot(C.A.row, C.x, C.b, red, cdist) < 0) return ERROR_MISC;
            alpha = std::min(red[0] / red[1], 1.0);
        }
        if (param.coarse_scaling == 1) d_aAxpy(alpha, D.P, C.x, D.x);  // x_l += alpha P x_{l+1}  (ghosts of x_{l+1} are there)
        else {   // alpha == 1: the halo of x_{l+1} travels beside the interior rows of P
            CsrArgs pa{}; pa.x = C.x; pa.y = D.x; pa.alpha = 1.0;
            if (dist_launch<OP_ADD>(C, D.P, pa, &D) < 0) return ERROR_MISC;
        }
        if ((st0 = smooth(h, l, true, smoother, param.smooth_order, param.postsmooth_iter, relax, param.polynomial_degree)) < 0) return st0;
        if (num_lvl[l] < ncycles[l]) break;
        else num_lvl[l] = 0;
    }
    if (l > 0) goto ForwardSweep;
    return FASP_SUCCESS;
}

// z = B r  (PreCSR.c:416-435).  The AMG_param used by the cycle is re-initialised and
// only the fields of fasp_param_prec_to_amg (AuxParam.c:816-834) are carried over: tol
// stays 1e-6.  r is used in place as the level-0 rhs and the result is left in the
// level-0 iterate; *z receives that pointer (both copies of the reference are elided).
static int precond_amg(fasp_hip_amg* h, double* r, double** z)
{
    AMG_param p;
    fasp_param_amg_init(&p);
    const AMG_param& u = h->param;
    p.AMG_type = u.AMG_type; p.print_level = u.print_level; p.cycle_type = u.cycle_type;
    p.smoother = u.smoother; p.smooth_order = u.smooth_order; p.presmooth_iter = u.presmooth_iter;
    p.postsmooth_iter = u.postsmooth_iter; p.relaxation = u.relaxation;
    p.polynomial_degree = u.polynomial_degree; p.coarse_solver = u.coarse_solver;
    p.coarse_scaling = u.coarse_scaling; p.tentative_smooth = u.tentative_smooth;
    p.amli_degree = u.amli_degree; p.nl_amli_krylov_type = u.nl_amli_krylov_type;
    { const int rc = renumber_conflict(h, p); if (rc < 0) return rc; }
    DevLevel& D0 = h->L[0];
    const bool pre_marked = h->pre_marked;
    h->pre_marked = false;
    const bool ask_zr = h->want_zr && u.cycle_type != AMLI_CYCLE && u.cycle_type != NL_AMLI_CYCLE;   // (set by the PCG operator bundle for this apply only; V / W cycles end with the level-0 sweep)
    // Lazy coarse verdicts.  A coarsest level small enough for a one-launch solver is visited up to 2^(levels - 1) times per
    // W-cycle, and reading every solve's verdict (has the safe CG given up? then the reference's SPVGMRES net takes over) was a
    // host synchronisation each: 712 per solve of config 5.  The solvers now leave the minimum status and the iteration sum
    // in two device words, read ONCE per application; if a solve did give up, the application is replayed from r -- which a
    // cycle only reads -- with the verdicts read as they come, and the hierarchy stays in that mode.
    const DevLevel& Dc = h->L.back();
    // (not for full multigrid: its cycle carries the levels' iterates over from one application to the next, a replay would not start where the first attempt did)
    const bool lazy = g_tune.lazy_coarse && !h->coarse_sync && !h->use_fmg && h->L.size() > 1 && small_coarse_ok(Dc.A.row, Dc.A.nnz);
    if (lazy && !h->d_lazy) {
        HIPCK(hipMalloc((void**)&h->d_lazy, 2 * sizeof(int)));
        HIPCK(hipHostMalloc((void**)&h->h_lazy, 4 * sizeof(int), hipHostMallocDefault));
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        const bool la = lazy && attempt == 0;
        D0.b = r;
        D0.x_zero = true;
        D0.presmoothed = attempt == 0 && pre_marked && !h->use_fmg;   // (the CG update wrote x = (w r) / d of this r: csr_ops, pcg.hip.h; a replay redoes that sweep itself)
        h->zr_G = 0;
        if (la) {
            h->h_lazy[2] = 0x7fffffff; h->h_lazy[3] = 0;
            HIPCK(hipMemcpyAsync(h->d_lazy, h->h_lazy + 2, 2 * sizeof(int), hipMemcpyHostToDevice, g_ctx.stream));
        }
        h->lazy_active = la;
        int st = FASP_SUCCESS;
        for (int i = u.maxit; i-- && st >= 0;) {
            h->want_zr = ask_zr && i == 0 && !h->use_fmg;   // the last sweep of the last cycle leaves the (z, r) partials
            st = h->use_fmg ? fmg_cycle(h, p) : mgcycle(h, p);  // fasp_precond_famg (PreCSR.c:560) / fasp_precond_amg
        }
        h->lazy_active = false;
        h->want_zr = false;
        if (st < 0) return st;
        if (!la) break;
        HIPCK(hipMemcpyAsync(h->h_lazy, h->d_lazy, 2 * sizeof(int), hipMemcpyDeviceToHost, g_ctx.stream));
        HIPCK(hipStreamSynchronize(g_ctx.stream));
        if (h->h_lazy[0] >= 0 && g_tune.lazy_coarse != 2) { h->coarse_iters += h->h_lazy[1]; break; }
        h->coarse_sync = true;   // (fasp_hip_tune("lazy_coarse", 2): every first application is replayed -- tests)
    }
    materialise_zero(D0);
    *z = D0.x;
    return FASP_SUCCESS;
}

// fasp_amg_solve (PreMGSolve.c:49): multigrid cycles as a stand-alone iteration on the resident
// vectors h->b, h->u.  The cycle receives the caller's AMG_param (coarse tolerance tol * 1e-4).
static int amg_solve_device(fasp_hip_amg* h, const AMG_param& param, Hist& hist, PcgOut& out)
{
    DevLevel& D0 = h->L[0];
    const int m = D0.A.row, MaxIt = param.maxit, prtlvl = param.print_level;
    const bool dist = h->distributed;
    const double tol = param.tol;
    hipStream_t s = g_ctx.stream;
    double red[2], relres1 = 1.0, absres0, absres = 0.0;
    int iter = 0, st;
    if ((st = renumber_conflict(h, param)) < 0) return st;
    if (d_dot(m, h->b, h->b, red, dist) < 0) return ERROR_MISC;
    const double sumb = std::sqrt(red[0]);
    absres0 = sumb;
    D0.b = h->b;
    HIPCK(hipMemcpyAsync(D0.x, h->u, sizeof(double) * m, hipMemcpyDeviceToDevice, s));
    D0.x_zero = false;
    itinfo(prtlvl, STOP_REL_RES, iter, relres1, sumb, 0.0);
    hist.push(sumb);
    if (sumb <= SMALLREAL) HIPCK(hipMemsetAsync(D0.x, 0, sizeof(double) * m, s));
    while ((iter++ < MaxIt) & (sumb > SMALLREAL)) {
        if ((st = mgcycle(h, param)) < 0) return st;
        materialise_zero(D0);
        { CsrArgs a{}; a.x = D0.x; a.y = D0.w; a.b = D0.b; if (dist_launch<OP_RESID>(D0, D0.A, a) < 0) return ERROR_MISC; }
        if (d_dot(m, D0.w, D0.w, red, dist) < 0) return ERROR_MISC;
        absres = std::sqrt(red[0]);
        relres1 = absres / std::max(SMALLREAL, sumb);
        const double factor = absres / absres0;
        absres0 = absres;
        itinfo(prtlvl, STOP_REL_RES, iter, relres1, absres, factor);
        hist.push(absres);
        if (relres1 < tol) break;
    }
    HIPCK(hipMemcpyAsync(h->u, D0.x, sizeof(double) * m, hipMemcpyDeviceToDevice, s));
    HIPCK(hipStreamSynchronize(s));
    if (prtlvl > PRINT_NONE) {
        if (iter > MaxIt) std::printf("### WARNING: MaxIt = %d reached with relative residual %.10e.\n", MaxIt, relres1);
        else if (iter >= 0) std::printf("Number of iterations = %d with relative residual %.10e.\n", iter, relres1);
    }
    out.relres = relres1; out.absres = absres; out.normr0 = sumb;
    return iter > MaxIt ? ERROR_SOLVER_MAXIT : iter;
}

// Krylov operator bundles of the CSR hierarchy: level 0 (with the AMG preconditioner) and the
// coarsest level (no preconditioner: the SPVGMRES safety net)
static KOps csr_ops(fasp_hip_amg* h, int level, bool with_pc)
{
    KOps K;
    DevLevel* Lv = &h->L[level];
    K.n = Lv->A.row; K.nvec = (size_t)Lv->nvec; K.fmt = "CSR";
    K.dist = (level == 0) && h->distributed;
    // The Krylov texts call halo(v) right before the operator that reads v.  Here halo() only notes the vector; the
    // operator that follows does the exchange beside its interior rows (dist_launch).  An operator that finds a
    // different vector pending exchanges that one first, the plain way.
    K.halo = [Lv](double* v) { Lv->halo_pending = v; return 0; };
    auto run = [Lv](auto op_tag, CsrArgs a) {
        constexpr int OP = decltype(op_tag)::value;
        double* pend = Lv->halo_pending;
        Lv->halo_pending = nullptr;
        if (pend && pend != a.x) { if (halo_exchange(*Lv, pend) < 0) return -1; pend = nullptr; }
        if (pend) return dist_launch<OP>(*Lv, Lv->A, a);
        return launch_csr<OP>(Lv->A, a);
    };
    K.mxv = [run](const double* x, double* y) {
        CsrArgs a{}; a.x = x; a.y = y;
        if (run(std::integral_constant<int, OP_MXV>(), a) < 0) comm_mark_failed();
    };
    K.resid = [run](const double* x, const double* b, double* r) {
        CsrArgs a{}; a.x = x; a.y = r; a.b = b;
        if (run(std::integral_constant<int, OP_RESID>(), a) < 0) comm_mark_failed();
    };
    K.mxv_dot = [run](const double* x, double* y) {
        CsrArgs a{}; a.x = x; a.y = y; a.dotv = x; a.partials = g_ctx.d_partials;
        const int G = run(std::integral_constant<int, OP_MXV_DOT>(), a);
        if (G < 0) comm_mark_failed();
        return std::max(G, 8);
    };
    if (with_pc) {
        K.pc = [h](double* in, double** out) { return precond_amg(h, in, out); };
        // PCG: the same apply, asking for the partials of (z, r); returns their number (0: take the dot product yourself)
        const AMG_param& ap = h->param;
        if (g_tune.fuse_presmooth && ap.smoother == SMOOTHER_JACOBI && ap.presmooth_iter >= 1 && ap.maxit >= 1 && !h->use_fmg &&
            ap.cycle_type != AMLI_CYCLE && ap.cycle_type != NL_AMLI_CYCLE && h->L.size() > 1) {
            K.pre_x = [Lv]() { return Lv->x; };
            K.pre_diag = Lv->diag; K.pre_omega = ap.relaxation;
            K.pre_diag_uniform = Lv->diag_uniform; K.pre_diag_value = Lv->diag_value;
            K.mark_presmoothed = [h]() { h->pre_marked = true; };
        }
        K.pc_zr = [h](double* in, double** out, int* G) {
            h->want_zr = true;
            const int st = precond_amg(h, in, out);
            h->want_zr = false;
            *G = st < 0 ? 0 : h->zr_G;
            h->zr_G = 0;
            return st;
        };
    }
    const int set = level == 0 ? 0 : 1;
    K.ws = &h->gm[set]; K.ws_len = &h->gm_len[set]; K.hh = &h->gm_hh;
    K.stats = h;
    return K;
}

