// krylov.hip.h -- Krylov drivers on device vectors against the operator bundle KOps: GMRES family, BiCGstab, MinRes, GCG, GCR, matrix-free texts.
// Part of the single translation unit solver.hip (included there, in this order; not a stand-alone header).

struct KOps;
static KOps csr_ops(fasp_hip_amg* h, int level, bool with_pc);
// forward declarations (the coarse fallback and the preconditioner call each other's owners)
static int precond_amg(fasp_hip_amg* h, double* r, double** z);
static void itinfo(int ptrlvl, int stop_type, int iter, double relres, double absres, double factor);
struct PcgOut { double relres, absres, normr0; };
struct Hist {
    double* h; int cap; int n;
    void push(double v) { if (h && n < cap) h[n] = v; ++n; }
};

// Operator bundle of the Krylov drivers: the reference has one textual copy of every Krylov
// method per matrix format (KryPcg.c:96 / :386, KryPvgmres.c:66 / :416, ...); here the
// drivers are written once against these callbacks.
struct KOps {
    int    n = 0;       // owned entries
    size_t nvec = 0;    // vector length incl. ghosts
    bool   dist = false;
    const char* fmt = "CSR";
    std::function<int(double*)> halo;                                      // refresh ghost entries of v
    std::function<void(const double*, double*)> mxv;                       // y = A x
    std::function<void(const double*, const double*, double*)> resid;      // r = b - A x
    std::function<int(const double*, double*)> mxv_dot;                    // y = A x + partials of (y,x); returns #partials, < 0: unavailable
    std::function<int(double*, double**)> pc;                              // *out = B in (empty: identity)
    // PCG: the update kernel may write the preconditioner's first Jacobi sweep of r (pre_x() = where, pre_diag, pre_omega);
    // mark_presmoothed() tells the preconditioner that its next apply finds that sweep done
    std::function<double*()> pre_x;
    const double* pre_diag = nullptr;
    bool   pre_diag_uniform = false;   // every entry of pre_diag is pre_diag_value: the kernel need not read the vector
    double pre_diag_value = 0.0;
    double pre_omega = 0.0;
    std::function<void()> mark_presmoothed;
    std::function<int(double*, double**, int*)> pc_zr;                     // pc + partials of (out, in) in g_ctx.d_partials (count in *G, 0: none)
    std::vector<double*>* ws = nullptr;                                    // GMRES workspace
    size_t* ws_len = nullptr;
    double** hh = nullptr;
    fasp_hip_amg* stats = nullptr;                                         // event pool for the SpMV timer
};

static void d_scale(int n, double a, double* x)
{
    if (a == 1.0) return;  // BlaArray.c:46
    hipLaunchKernelGGL(k_scale, dim3(vec_grid(n)), dim3(BLOCK), 0, g_ctx.stream, n, a, x);
}

// ---------------------------------------------------------------------------
// Variable-restart right-preconditioned GMRES family on device vectors:
//   mode 0  fasp_solver_dcsr_pvgmres    KryPvgmres.c:66-412
//   mode 1  fasp_solver_dcsr_pvfgmres   KryPvfgmres.c:67-384   (flexible)
//   mode 2  fasp_solver_dcsr_spvgmres   KrySPvgmres.c:68-441   (safe net; coarse-level fallback)
// The scalar control flow (restart adaptation, Givens rotations, back substitution, false-
// convergence check, best-iterate safety net) runs on the host exactly as in the reference;
// the modified Gram-Schmidt chain runs on the device without host round trips (k_mgs_step).
// `set` selects the workspace (0: level 0, 1: coarsest level); Lv is the level the operator
// acts on (halo plan); use_pc applies the AMG preconditioner (level 0 only).
// ---------------------------------------------------------------------------
static int gmres_device(KOps& K, const double* b, double* x, int mode_in, double tol, double abstol, int MaxIt,
                        int restart, int StopType, int PrtLvl, Hist* hist, PcgOut* out)
{
    const bool fixed = mode_in == 3;       // mode 3: fixed restart, fasp_solver_d*_pgmres (KryPgmres.c:66) ...
    const int  mode = fixed ? 0 : mode_in; // ... the text of mode 0 with four differences
    const int n = K.n;
    const size_t nv = K.nvec;
    const bool dist = K.dist;
    const int MIN_ITER = 0;
    const double epsmac = SMALLREAL, cr_max = 0.99, cr_min = 0.174, maxdiff = tol * STAG_RATIO;
    int iter = 0, i = 0, j, k, st;
    double r_norm, r_normb, gamma, t, red[8];
    double absres0 = BIGREAL, absres = BIGREAL, relres = BIGREAL, normu = BIGREAL;
    double b_norm = 0.0, den_norm = 0.0, epsilon = 0.0, cr = 1.0, r_norm_old = 0.0;
    const int d = 3, restart_max = restart, restart_min = 3;
    int Restart = fixed ? std::min(restart, MaxIt) : restart;
    const int Restart1 = restart + 1;
    int iter_best = 0;
    double absres_best = BIGREAL;
    hipStream_t s = g_ctx.stream;
    const int G = vec_grid(n);

    // workspace: p[0..Restart], w, x_best (mode 2), z[0..Restart) (mode 1)
    const size_t need = (size_t)Restart1 + 2 + (mode == 1 ? (size_t)Restart1 : 0);
    if (*K.ws_len != nv) {
        for (double* q : *K.ws) if (q) (void)hipFree(q);
        K.ws->clear();
        *K.ws_len = nv;
    }
    while (K.ws->size() < need) {
        double* q = nullptr;
        HIPCK(hipMalloc(&q, sizeof(double) * std::max<size_t>(nv, 1)));
        HIPCK(hipMemsetAsync(q, 0, sizeof(double) * nv, s));
        K.ws->push_back(q);
    }
    if (!*K.hh) HIPCK(hipMalloc(K.hh, sizeof(double) * 1024));
    if (Restart1 + 2 > 1024) return ERROR_INPUT_PAR;
    std::vector<double*>& W = *K.ws;
    double* const gm_hh = *K.hh;
    double** p = W.data();
    double*  w = W[Restart1];
    double*  x_best = W[Restart1 + 1];
    double** z = mode == 1 ? W.data() + Restart1 + 2 : nullptr;
    double*  r = nullptr;  // preconditioned / work vector (may alias an internal buffer)
    std::vector<double> rs(Restart1 + 1, 0.0), c(Restart + 1, 0.0), sn(Restart + 1, 0.0);
    std::vector<std::vector<double>> hh(Restart1, std::vector<double>(Restart + 1, 0.0));
    std::vector<double> norms((size_t)MaxIt + 2, 0.0);

    auto apply_pc = [&](double* in, double** outp) -> int {  // *outp = B in (pointer to the result)
        if (K.pc) return K.pc(in, outp);
        *outp = in;
        return FASP_SUCCESS;
    };
    auto true_residual = [&](const double* xx, double* rr) -> int {  // rr = b - A xx
        if (K.halo(const_cast<double*>(xx)) < 0) return ERROR_MISC;
        K.resid(xx, b, rr);
        return FASP_SUCCESS;
    };

    if (PrtLvl > PRINT_NONE)
        std::printf(fixed ? "\nCalling GMRes solver (%s) ...\n" : mode == 0 ? "\nCalling VGMRes solver (%s) ...\n"
                    : mode == 1 ? "\nCalling VFGMRes solver (%s) ...\n" : "\nCalling Safe VGMRes solver (%s) ...\n", K.fmt);

    if ((st = true_residual(x, p[0])) < 0) return st;
    if (mode == 1) { if (d_dot(n, b, b, red, dist) < 0) return ERROR_MISC; b_norm = std::sqrt(red[0]); }
    if (d_dot(n, p[0], p[0], red, dist) < 0) return ERROR_MISC;
    r_norm = std::sqrt(red[0]);

    if (mode == 1) {
        norms[0] = r_norm;
        if (PrtLvl >= PRINT_SOME) {
            std::printf("L2 norm of %s = %.10e.\n", "right-hand side", b_norm);
            std::printf("L2 norm of %s = %.10e.\n", "residual", r_norm);
        }
        den_norm = (b_norm > 0.0) ? b_norm : r_norm;
        epsilon = tol * den_norm;
        if (hist) hist->push(r_norm);
        if (r_norm < epsilon || r_norm < abstol) goto FINISHED;
        if (b_norm > 0.0) itinfo(PrtLvl, StopType, iter, norms[iter] / b_norm, norms[iter], 0);
        else itinfo(PrtLvl, StopType, iter, norms[iter], norms[iter], 0);
    } else {
        switch (StopType) {
            case STOP_REL_RES:
                absres0 = std::max(SMALLREAL, r_norm);
                relres = r_norm / absres0;
                break;
            case STOP_REL_PRECRES:
                if ((st = apply_pc(p[0], &r)) < 0) return st;
                if (d_dot(n, p[0], r, red, dist) < 0) return ERROR_MISC;
                r_normb = std::sqrt(red[0]);
                absres0 = std::max(SMALLREAL, r_normb);
                relres = r_normb / absres0;
                break;
            case STOP_MOD_REL_RES:
                if (d_dot(n, x, x, red, dist) < 0) return ERROR_MISC;
                normu = std::max(SMALLREAL, std::sqrt(red[0]));
                absres0 = r_norm;
                relres = absres0 / normu;
                break;
            default:
                std::printf("### ERROR: Unknown stopping type! [%s]\n", "fasp_solver_dcsr_pvgmres");
                goto FINISHED;
        }
        if (hist) hist->push(r_norm);
        if (mode == 0) { if (relres < tol || absres0 < abstol) goto FINISHED; }
        else           { if (relres < tol) goto FINISHED; }
        itinfo(PrtLvl, StopType, 0, relres, absres0, 0);
        norms[0] = relres;
    }

    while (iter < MaxIt && (!fixed || relres > tol)) {
        rs[0] = r_norm_old = r_norm;
        if (mode == 1 && r_norm == 0.0) { if (out) { out->relres = 0.0; out->absres = 0.0; out->normr0 = den_norm; } return iter; }
        if (mode != 1) d_scale(n, 1.0 / r_norm, p[0]);

        if (!fixed) {
            if (cr > cr_max || iter == 0) Restart = restart_max;
            else if (cr < cr_min) { /* keep */ }
            else { if (Restart - d > restart_min) Restart -= d; else Restart = restart_max; }
        }

        if (mode == 1) d_scale(n, 1.0 / r_norm, p[0]);

        i = 0;
        while (i < Restart && iter < MaxIt) {
            i++; iter++;
            if ((st = apply_pc(p[i - 1], &r)) < 0) return st;
            if (mode == 1 && r != z[i - 1])
                HIPCK(hipMemcpyAsync(z[i - 1], r, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
            if (K.halo(r) < 0) return ERROR_MISC;
            K.mxv(r, p[i]);
            // modified Gram-Schmidt on the device: hh_0 = (p_0, p_i); then i fused steps
            // One rank: a step sums the previous step's partials itself (two partial buffers in turn) -- i + 2 launches per column
            // instead of 2 i + 2.  Several ranks: every coefficient is all-reduced, one k_finalize + collective each.
            if (!(dist && comm_size() > 1) && g_tune.pcg_fold) {
                double* const pb[2] = {g_ctx.d_partials, g_ctx.d_partials2 + 4 * MAXGRID};
                hipLaunchKernelGGL(k_dot, dim3(G), dim3(BLOCK), 0, s, n, p[0], p[i], pb[0]);
                for (j = 0; j < i; j++)
                    hipLaunchKernelGGL(k_mgs_step, dim3(G), dim3(BLOCK), 0, s, n, (const double*)nullptr,
                                       (const double*)p[j], p[i], (const double*)(j + 1 < i ? p[j + 1] : nullptr),
                                       pb[(j + 1) & 1], (const double*)pb[j & 1], G, gm_hh + j);
                hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLOCK), 0, s, (const double*)pb[i & 1], G, 1, 0u, gm_hh + i);
            } else {
                hipLaunchKernelGGL(k_dot, dim3(G), dim3(BLOCK), 0, s, n, p[0], p[i], g_ctx.d_partials);
                d_finalize_to(G, 1, 0u, gm_hh, dist);
                for (j = 0; j < i; j++) {
                    hipLaunchKernelGGL(k_mgs_step, dim3(G), dim3(BLOCK), 0, s, n, (const double*)(gm_hh + j),
                                       (const double*)p[j], p[i], (const double*)(j + 1 < i ? p[j + 1] : nullptr),
                                       g_ctx.d_partials);
                    d_finalize_to(G, 1, 0u, gm_hh + j + 1, dist);
                }
            }
            HIPCK(hipMemcpyAsync(g_ctx.h_part, gm_hh, sizeof(double) * (i + 1), hipMemcpyDeviceToHost, s));
            HIPCK(hipStreamSynchronize(s));
            for (j = 0; j < i; j++) hh[j][i - 1] = g_ctx.h_part[j];
            t = std::sqrt(g_ctx.h_part[i]);
            hh[i][i - 1] = t;
            if (fixed ? (std::fabs(t) > SMALLREAL) : (t != 0.0)) d_scale(n, 1.0 / t, p[i]);
            for (j = 1; j < i; ++j) {
                t = hh[j - 1][i - 1];
                hh[j - 1][i - 1] = sn[j - 1] * hh[j][i - 1] + c[j - 1] * t;
                hh[j][i - 1] = -sn[j - 1] * t + c[j - 1] * hh[j][i - 1];
            }
            t = hh[i][i - 1] * hh[i][i - 1];
            t += hh[i - 1][i - 1] * hh[i - 1][i - 1];
            gamma = std::sqrt(t);
            if (fixed) gamma = std::max(gamma, SMALLREAL);
            else if (gamma == 0.0) gamma = epsmac;
            c[i - 1] = hh[i - 1][i - 1] / gamma;
            sn[i - 1] = hh[i][i - 1] / gamma;
            rs[i] = -sn[i - 1] * rs[i - 1];
            rs[i - 1] = c[i - 1] * rs[i - 1];
            hh[i - 1][i - 1] = sn[i - 1] * hh[i][i - 1] + c[i - 1] * hh[i - 1][i - 1];
            if (mode == 1) {
                r_norm = std::fabs(rs[i]);
                norms[iter] = r_norm;
                if (b_norm > 0) itinfo(PrtLvl, StopType, iter, norms[iter] / b_norm, norms[iter], norms[iter] / norms[iter - 1]);
                else itinfo(PrtLvl, StopType, iter, norms[iter], norms[iter], norms[iter] / norms[iter - 1]);
                if (hist) hist->push(r_norm);
                if (r_norm <= epsilon && iter >= MIN_ITER) break;
            } else {
                absres = r_norm = std::fabs(rs[i]);
                relres = absres / absres0;
                norms[iter] = relres;
                itinfo(PrtLvl, StopType, iter, relres, absres, norms[iter] / norms[iter - 1]);
                if (hist) hist->push(absres);
                if (mode == 0) { if (relres < tol && iter >= MIN_ITER) break; }
                else           { if (relres <= tol && iter >= MIN_ITER) break; }
            }
        }

        // back substitution (host) and solution update
        rs[i - 1] = rs[i - 1] / hh[i - 1][i - 1];
        for (k = i - 2; k >= 0; k--) {
            t = 0.0;
            for (j = k + 1; j < i; j++) t -= hh[k][j] * rs[j];
            t += rs[k];
            rs[k] = t / hh[k][k];
        }
        if (mode == 1) {
            HIPCK(hipMemcpyAsync(w, z[i - 1], sizeof(double) * n, hipMemcpyDeviceToDevice, s));
            d_scale(n, rs[i - 1], w);
            for (j = i - 2; j >= 0; j--) d_axpy(n, rs[j], z[j], w);
            r = w;
        } else {
            HIPCK(hipMemcpyAsync(w, p[i - 1], sizeof(double) * n, hipMemcpyDeviceToDevice, s));
            d_scale(n, rs[i - 1], w);
            for (j = i - 2; j >= 0; j--) d_axpy(n, rs[j], p[j], w);
            if ((st = apply_pc(w, &r)) < 0) return st;
        }
        d_axpy(n, 1.0, r, x);

        if (mode == 2) {  // safety net, KrySPvgmres.c:287-299
            if (d_norms(n, x, red, dist) < 0) return ERROR_MISC;
            if (std::isnan(red[0])) { absres = BIGREAL; goto RESTORE_BESTSOL; }
            if (absres < absres_best - maxdiff) {
                absres_best = absres;
                iter_best = iter;
                HIPCK(hipMemcpyAsync(x_best, x, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
            }
        }

        if ((mode == 0 && relres < tol && iter >= MIN_ITER) || (mode == 2 && relres <= tol && iter >= MIN_ITER) ||
            (mode == 1 && r_norm <= epsilon && iter >= MIN_ITER)) {
            const double computed_relres = relres;
            if ((st = true_residual(x, w)) < 0) return st;
            if (d_dot(n, w, w, red, dist) < 0) return ERROR_MISC;
            r_norm = std::sqrt(red[0]);
            switch (StopType) {
                case STOP_REL_RES:
                    if (mode == 1) relres = r_norm / den_norm;
                    else { absres = r_norm; relres = absres / absres0; }
                    break;
                case STOP_REL_PRECRES: {
                    double* zz = nullptr;
                    if ((st = apply_pc(w, &zz)) < 0) return st;
                    if (d_dot(n, zz, w, red, dist) < 0) return ERROR_MISC;
                    if (mode == 1) { r_normb = std::sqrt(red[0]); relres = r_normb / den_norm; }
                    else { absres = std::sqrt(red[0]); relres = absres / absres0; }
                } break;
                case STOP_MOD_REL_RES:
                    if (d_dot(n, x, x, red, dist) < 0) return ERROR_MISC;
                    normu = std::max(SMALLREAL, std::sqrt(red[0]));
                    if (mode == 1) relres = r_norm / normu;
                    else { absres = r_norm; relres = absres / normu; }
                    break;
            }
            if (mode != 1) norms[iter] = relres;
            if ((mode == 0 && relres < tol) || (mode != 0 && relres <= tol)) break;
            if (mode == 1 && PrtLvl >= PRINT_SOME)
                std::printf("### WARNING: False convergence! [%s:%d]\n", "fasp_solver_dcsr_pvfgmres", 328);
            HIPCK(hipMemcpyAsync(p[0], w, sizeof(double) * n, hipMemcpyDeviceToDevice, s));  // restart from the true residual
            i = 0;
            if (mode == 0 && PrtLvl >= PRINT_MORE) {
                std::printf("### WARNING: The computed relative residual = %.10e!\n", computed_relres);
                std::printf("### WARNING: The actual relative residual = %.10e!\n", relres);
            }
        }

        // residual vector of the restart (KryPvgmres.c:390-401)
        for (j = i; j > 0; j--) {
            rs[j - 1] = -sn[j - 1] * rs[j];
            rs[j] = c[j - 1] * rs[j];
        }
        if (i) hipLaunchKernelGGL(k_axpy_self, dim3(G), dim3(BLOCK), 0, s, n, rs[i] - 1.0, p[i]);
        for (j = i - 1; j > 0; j--) d_axpy(n, rs[j], p[j], p[i]);
        if (i) {
            hipLaunchKernelGGL(k_axpy_self, dim3(G), dim3(BLOCK), 0, s, n, rs[0] - 1.0, p[0]);
            d_axpy(n, 1.0, p[i], p[0]);
        }
        cr = r_norm / r_norm_old;
    }

RESTORE_BESTSOL:
    if (mode == 2 && iter != iter_best) {  // KrySPvgmres.c:357-389
        if ((st = true_residual(x_best, w)) < 0) return st;
        if (d_dot(n, w, w, red, dist) < 0) return ERROR_MISC;
        absres_best = std::sqrt(red[0]);
        if (absres > absres_best + maxdiff || std::isnan(absres)) {
            if (PrtLvl > PRINT_NONE)
                std::printf("### WARNING: Discard current iteration. Restore iteration %d!\n", iter_best);
            HIPCK(hipMemcpyAsync(x, x_best, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
            relres = absres_best / absres0;
        }
    }

FINISHED:
    {
        const double fr = (mode == 1) ? r_norm / den_norm : relres;
        if (PrtLvl > PRINT_NONE) {
            if (iter > MaxIt) std::printf("### WARNING: MaxIt = %d reached with relative residual %.10e.\n", MaxIt, fr);
            else if (iter >= 0) std::printf("Number of iterations = %d with relative residual %.10e.\n", iter, fr);
        }
        if (out) { out->relres = fr; out->absres = r_norm; out->normr0 = (mode == 1) ? den_norm : absres0; }
    }
    HIPCK(hipStreamSynchronize(s));
    if (iter >= MaxIt) return ERROR_SOLVER_MAXIT;
    return iter;
}

// ---------------------------------------------------------------------------
// BiCGstab (KryPbcgs.c:62 / :400: the formulation of MATLAB's bicgstab with half steps,
// stagnation counters and the minimal-residual iterate) on device vectors
// ---------------------------------------------------------------------------
static int bicgstab_device(KOps& K, const double* b, double* x, double tol, int MaxIt, int PrtLvl, Hist* hist,
                           PcgOut* out)
{
    const int m = K.n;
    const size_t nv = K.nvec;
    const bool dist = K.dist;
    hipStream_t s = g_ctx.stream;
    if (*K.ws_len != nv) {
        for (double* q : *K.ws) if (q) (void)hipFree(q);
        K.ws->clear();
        *K.ws_len = nv;
    }
    while (K.ws->size() < 9) {
        double* q = nullptr;
        HIPCK(hipMalloc(&q, sizeof(double) * std::max<size_t>(nv, 1)));
        HIPCK(hipMemsetAsync(q, 0, sizeof(double) * nv, s));
        K.ws->push_back(q);
    }
    std::vector<double*>& W = *K.ws;
    double *r = W[0], *rt = W[1], *p = W[2], *v = W[3], *xhalf = W[4], *sv = W[5], *t = W[6], *xmin = W[7], *tmp = W[8];
    double *ph = nullptr, *sh = nullptr;  // preconditioned vectors (may alias the preconditioner's output)
    double red[8];
    auto cp = [&](double* dst, const double* src) -> int {
        HIPCK(hipMemcpyAsync(dst, src, sizeof(double) * m, hipMemcpyDeviceToDevice, s));
        return 0;
    };
    auto nrm2 = [&](const double* y, double& val) -> int {
        if (d_dot(m, y, y, red, dist) < 0) return ERROR_MISC;
        val = std::sqrt(red[0]);
        return 0;
    };
    auto dot = [&](const double* y, const double* z, double& val) -> int {
        if (d_dot(m, y, z, red, dist) < 0) return ERROR_MISC;
        val = red[0];
        return 0;
    };
    auto resid = [&](double* xx, double* rr) -> int {  // rr = b - A xx
        if (K.halo(xx) < 0) return ERROR_MISC;
        K.resid(xx, b, rr);
        return 0;
    };
    auto apply_pc = [&](double* in, double** outp) -> int {
        if (K.pc) return K.pc(in, outp);
        *outp = in;
        return FASP_SUCCESS;
    };
    double n2b, tolb, relres = BIGREAL, absres0 = BIGREAL, absres = BIGREAL;
    double alpha, beta, omega, rho, rho1, rtv, tt, st_, normr, normr_act, normph, normx, imin, norm_sh, norm_xhalf, normrmin = 0.0;
    int iter = 0, stag = 1, moresteps = 1, maxmsteps = 1, flag = 1, maxstagsteps = 3, st;
    (void)stag; (void)moresteps; (void)maxmsteps;

    if (PrtLvl > PRINT_NONE) std::printf("\nCalling BiCGstab solver (%s) ...\n", K.fmt);
    if (nrm2(b, n2b) < 0) return ERROR_MISC;
    if (cp(xmin, x) < 0) return ERROR_MISC;
    imin = 0;
    tolb = n2b * tol;
    if (resid(x, r) < 0) return ERROR_MISC;
    if (nrm2(r, normr) < 0) return ERROR_MISC;
    normr_act = normr;
    relres = normr / n2b;
    if (hist) hist->push(normr);
    if (normr <= tolb) { flag = 0; iter = 0; goto FINISHED; }
    itinfo(PrtLvl, STOP_REL_RES, iter, relres, n2b, 0.0);
    if (cp(rt, r) < 0) return ERROR_MISC;
    normrmin = normr;
    rho = 1.0; omega = 1.0; stag = 0; alpha = 0.0;
    moresteps = 0; maxmsteps = 10;

    for (iter = 1; iter <= MaxIt; iter++) {
        rho1 = rho;
        if (dot(rt, r, rho) < 0) return ERROR_MISC;
        if ((rho == 0.0) || (std::fabs(rho) >= DBL_MAX)) { flag = 4; goto FINISHED; }
        if (iter == 1) { if (cp(p, r) < 0) return ERROR_MISC; }
        else {
            beta = (rho / rho1) * (alpha / omega);
            if ((beta == 0) || (std::fabs(beta) > DBL_MAX)) { flag = 4; goto FINISHED; }
            d_axpy(m, -omega, v, p);
            d_axpby(m, 1.0, r, beta, p);
        }
        if ((st = apply_pc(p, &ph)) < 0) return st;
        if (K.halo(ph) < 0) return ERROR_MISC;
        K.mxv(ph, v);
        if (dot(rt, v, rtv) < 0) return ERROR_MISC;
        if ((rtv == 0.0) || (std::fabs(rtv) > DBL_MAX)) { flag = 4; goto FINISHED; }
        alpha = rho / rtv;
        if (std::fabs(alpha) > DBL_MAX) {
            flag = 4;
            std::printf("### WARNING: Divided by zero! [%s:%d]\n", "fasp_solver_dcsr_pbcgs", 178);
            goto FINISHED;
        }
        if (nrm2(x, normx) < 0 || nrm2(ph, normph) < 0) return ERROR_MISC;
        if (std::fabs(alpha) * normph < DBL_EPSILON * normx) stag = stag + 1; else stag = 0;
        if (cp(xhalf, x) < 0) return ERROR_MISC;
        d_axpy(m, alpha, ph, xhalf);   // xhalf = alpha ph + x
        if (cp(sv, r) < 0) return ERROR_MISC;
        d_axpy(m, -alpha, v, sv);      // s = -alpha v + r
        if (nrm2(sv, normr) < 0) return ERROR_MISC;
        normr_act = normr;
        absres = normr_act;
        itinfo(PrtLvl, STOP_REL_RES, iter, normr_act / n2b, absres, absres / absres0);
        if (hist) hist->push(absres);
        if ((normr <= tolb) || (stag >= maxstagsteps) || moresteps) {
            if (resid(xhalf, sv) < 0) return ERROR_MISC;
            if (nrm2(sv, normr_act) < 0) return ERROR_MISC;
            if (normr_act <= tolb) {
                if (cp(x, xhalf) < 0) return ERROR_MISC;
                flag = 0; imin = iter - 0.5;
                goto FINISHED;
            } else {
                if ((stag >= maxstagsteps) && (moresteps == 0)) stag = 0;
                moresteps = moresteps + 1;
                if (moresteps >= maxmsteps) { flag = 3; if (cp(x, xhalf) < 0) return ERROR_MISC; goto FINISHED; }
            }
        }
        if (stag >= maxstagsteps) { flag = 3; goto FINISHED; }
        if (normr_act < normrmin) {
            normrmin = normr_act;
            if (cp(xmin, xhalf) < 0) return ERROR_MISC;
            imin = iter - 0.5;
        }
        if ((st = apply_pc(sv, &sh)) < 0) return st;
        if (K.halo(sh) < 0) return ERROR_MISC;
        K.mxv(sh, t);
        if (dot(t, t, tt) < 0) return ERROR_MISC;
        if ((tt == 0) || (tt >= DBL_MAX)) { flag = 4; goto FINISHED; }
        if (dot(sv, t, st_) < 0) return ERROR_MISC;
        omega = st_ / tt;
        if (std::fabs(omega) > DBL_MAX) { flag = 4; goto FINISHED; }
        if (nrm2(sh, norm_sh) < 0 || nrm2(xhalf, norm_xhalf) < 0) return ERROR_MISC;
        if (std::fabs(omega) * norm_sh < DBL_EPSILON * norm_xhalf) stag = stag + 1; else stag = 0;
        if (cp(x, xhalf) < 0) return ERROR_MISC;
        d_axpy(m, omega, sh, x);       // x = omega sh + xhalf
        if (cp(r, sv) < 0) return ERROR_MISC;
        d_axpy(m, -omega, t, r);       // r = -omega t + s
        if (nrm2(r, normr) < 0) return ERROR_MISC;
        normr_act = normr;
        if ((normr <= tolb) || (stag >= maxstagsteps) || moresteps) {
            if (resid(x, r) < 0) return ERROR_MISC;
            if (nrm2(r, normr_act) < 0) return ERROR_MISC;
            if (normr_act <= tolb) { flag = 0; goto FINISHED; }
            else {
                if ((stag >= maxstagsteps) && (moresteps == 0)) stag = 0;
                moresteps = moresteps + 1;
                if (moresteps >= maxmsteps) { flag = 3; goto FINISHED; }
            }
        }
        if (normr_act < normrmin) { normrmin = normr_act; if (cp(xmin, x) < 0) return ERROR_MISC; imin = iter; }
        if (stag >= maxstagsteps) { flag = 3; goto FINISHED; }
        if (PrtLvl >= PRINT_MORE) std::printf("### WARNING: The actual relative residual = %.10e!\n", relres);
        absres0 = absres;
    }
FINISHED:
    if (flag == 0) relres = normr_act / n2b;
    else {
        if (resid(xmin, tmp) < 0) return ERROR_MISC;
        if (nrm2(tmp, normr) < 0) return ERROR_MISC;
        if (normr <= normr_act) { if (cp(x, xmin) < 0) return ERROR_MISC; iter = (int)imin; relres = normr / n2b; }
        else relres = normr_act / n2b;
    }
    if (PrtLvl > PRINT_NONE) {
        if (iter > MaxIt) std::printf("### WARNING: MaxIt = %d reached with relative residual %.10e.\n", MaxIt, relres);
        else if (iter >= 0) std::printf("Number of iterations = %d with relative residual %.10e.\n", iter, relres);
    }
    if (out) { out->relres = relres; out->absres = relres * n2b; out->normr0 = n2b; }
    HIPCK(hipStreamSynchronize(s));
    return iter > MaxIt ? ERROR_SOLVER_MAXIT : iter;
}

// ---------------------------------------------------------------------------
// The remaining `itsolver_type`s of fasp_solver_dcsr_itsolver (SolCSR.c:56): MinRes, GCG, GCR.
// Host control flow as in the reference, vectors and every operation on the device; these
// are completeness modes (one host round trip per scalar), the tuned drivers are CG and GMRES.
// ---------------------------------------------------------------------------
struct KVecOps {
    KOps& K;
    const int m;
    const size_t nv;
    hipStream_t s;
    double red[8];
    explicit KVecOps(KOps& K_) : K(K_), m(K_.n), nv(K_.nvec), s(g_ctx.stream) {}
    int ensure(size_t count)
    {
        if (*K.ws_len != nv) {
            for (double* q : *K.ws) if (q) (void)hipFree(q);
            K.ws->clear();
            *K.ws_len = nv;
        }
        while (K.ws->size() < count) {
            double* q = nullptr;
            HIPCK(hipMalloc(&q, sizeof(double) * std::max<size_t>(nv, 1)));
            HIPCK(hipMemsetAsync(q, 0, sizeof(double) * nv, s));
            K.ws->push_back(q);
        }
        return 0;
    }
    double* vec(size_t i) { return (*K.ws)[i]; }
    int cp(double* dst, const double* src)
    {
        if (dst != src) HIPCK(hipMemcpyAsync(dst, src, sizeof(double) * m, hipMemcpyDeviceToDevice, s));
        return 0;
    }
    int zero(double* dst) { HIPCK(hipMemsetAsync(dst, 0, sizeof(double) * m, s)); return 0; }
    int dot(const double* y, const double* z, double& val)
    {
        if (d_dot(m, y, z, red, K.dist) < 0) return ERROR_MISC;
        val = red[0];
        return 0;
    }
    int nrm2(const double* y, double& val)
    {
        if (d_dot(m, y, y, red, K.dist) < 0) return ERROR_MISC;
        val = std::sqrt(red[0]);
        return 0;
    }
    int mxv(double* x, double* y)  // y = A x (x's ghost entries refreshed first)
    {
        if (K.halo(x) < 0) return ERROR_MISC;
        K.mxv(x, y);
        return 0;
    }
    int resid(double* x, const double* b, double* r)  // r = b - A x
    {
        if (K.halo(x) < 0) return ERROR_MISC;
        K.resid(x, b, r);
        return 0;
    }
    int pc(double* in, double* dst)  // dst = B in (the preconditioner's own output buffer is copied out)
    {
        double* o = in;
        if (K.pc) { const int st = K.pc(in, &o); if (st < 0) return st; }
        return cp(dst, o);
    }
};
#define KCK(expr) do { const int st__ = (expr); if (st__ < 0) return st__; } while (0)

// fasp_solver_dcsr_pminres, KryPminres.c:61-448
static int minres_device(KOps& K, const double* b, double* u, double tol, double abstol, int MaxIt, int StopType,
                         int PrtLvl, Hist* hist, PcgOut* out)
{
    KVecOps V(K);
    const int m = V.m;
    const double maxdiff = tol * STAG_RATIO, sol_inf_tol = SMALLREAL;
    int iter = 0, stag = 1, more_step = 1;
    double absres0 = BIGREAL, absres = BIGREAL, normr0 = BIGREAL, relres = BIGREAL;
    double normu2 = BIGREAL, normuu, normp, factor, alpha, alpha0, alpha1, temp2, red[8];
    KCK(V.ensure(11));
    double *p0 = V.vec(0), *p1 = V.vec(1), *p2 = V.vec(2), *z0 = V.vec(3), *z1 = V.vec(4), *t0 = V.vec(5),
           *t1 = V.vec(6), *t = V.vec(7), *tp = V.vec(8), *tz = V.vec(9), *r = V.vec(10);
    auto resnorm = [&]() -> int {  // :228-247 and its two copies
        switch (StopType) {
            case STOP_REL_RES:
                KCK(V.dot(r, r, temp2)); absres = std::sqrt(temp2); relres = absres / normr0; break;
            case STOP_REL_PRECRES:
                KCK(V.pc(r, t)); KCK(V.dot(r, t, temp2)); temp2 = std::fabs(temp2);
                absres = std::sqrt(temp2); relres = absres / normr0; break;
            case STOP_MOD_REL_RES:
                KCK(V.dot(r, r, temp2)); absres = std::sqrt(temp2); relres = absres / normu2; break;
        }
        return 0;
    };
    auto restart = [&]() -> int {  // :331-368 == :409-446
        KCK(V.zero(p0));
        KCK(V.pc(r, p1));
        KCK(V.mxv(p1, tp));
        KCK(V.pc(tp, tz));
        KCK(V.dot(tz, tp, normp));
        normp = std::sqrt(normp);
        KCK(V.cp(t, p1));
        KCK(V.zero(t0)); KCK(V.zero(z0)); KCK(V.zero(t1)); KCK(V.zero(z1)); KCK(V.zero(p1));
        d_axpy(m, 1 / normp, t, p1);
        d_axpy(m, 1 / normp, tp, t1);
        d_axpy(m, 1 / normp, tz, z1);
        return 0;
    };
    if (PrtLvl > PRINT_NONE) std::printf("\nCalling MinRes solver (%s) ...\n", K.fmt);
    KCK(V.zero(p0));
    KCK(V.resid(u, b, r));
    KCK(V.pc(r, p1));
    switch (StopType) {
        case STOP_REL_RES:
            KCK(V.nrm2(r, absres0)); normr0 = std::max(SMALLREAL, absres0); relres = absres0 / normr0; break;
        case STOP_REL_PRECRES:
            KCK(V.dot(r, p1, temp2)); absres0 = std::sqrt(temp2);
            normr0 = std::max(SMALLREAL, absres0); relres = absres0 / normr0; break;
        case STOP_MOD_REL_RES:
            KCK(V.nrm2(r, absres0)); KCK(V.nrm2(u, normu2)); normu2 = std::max(SMALLREAL, normu2);
            relres = absres0 / normu2; break;
        default:
            std::printf("### ERROR: Unknown stopping type! [%s]\n", "fasp_solver_dcsr_pminres");
            goto FINISHED;
    }
    if (hist) hist->push(absres0);
    if (relres < tol || absres0 < abstol) goto FINISHED;
    itinfo(PrtLvl, StopType, iter, relres, absres0, 0.0);
    KCK(V.mxv(p1, tp));
    KCK(V.pc(tp, tz));
    KCK(V.dot(tz, tp, normp));
    normp = std::sqrt(std::fabs(normp));
    KCK(V.cp(t, p1));
    KCK(V.zero(p1));
    d_axpy(m, 1 / normp, t, p1);
    KCK(V.zero(t0)); KCK(V.zero(z0)); KCK(V.zero(t1)); KCK(V.zero(z1));
    d_axpy(m, 1.0 / normp, tp, t1);
    d_axpy(m, 1.0 / normp, tz, z1);

    while (iter++ < MaxIt) {
        KCK(V.dot(r, z1, alpha));
        d_axpy(m, alpha, p1, u);
        d_axpy(m, -alpha, t1, r);
        KCK(V.mxv(z1, t));
        KCK(V.dot(z1, t, alpha1));
        KCK(V.mxv(z0, t));
        KCK(V.dot(z1, t, alpha0));
        KCK(V.cp(p2, z1));
        d_axpy(m, -alpha1, p1, p2);
        d_axpy(m, -alpha0, p0, p2);
        KCK(V.mxv(p2, tp));
        KCK(V.pc(tp, tz));
        KCK(V.dot(tz, tp, normp));
        normp = std::sqrt(std::fabs(normp));
        KCK(V.cp(t, p2));
        KCK(V.zero(p2));
        d_axpy(m, 1 / normp, t, p2);
        KCK(V.cp(p0, p1)); KCK(V.cp(p1, p2)); KCK(V.cp(t0, t1)); KCK(V.cp(z0, z1));
        KCK(V.zero(t1)); KCK(V.zero(z1));
        d_axpy(m, 1 / normp, tp, t1);
        d_axpy(m, 1 / normp, tz, z1);
        if (d_norms(m, u, red, K.dist) < 0) return ERROR_MISC;  // ||u||^2, max|u|
        normu2 = std::sqrt(red[0]);
        KCK(resnorm());
        factor = absres / absres0;
        itinfo(PrtLvl, StopType, iter, relres, absres, factor);
        if (hist) hist->push(absres);

        if (factor > 0.9) {  // Check I, II (:256-373)
            if (red[1] <= sol_inf_tol) {
                if (PrtLvl > PRINT_MIN)
                    std::printf("### WARNING: Iteration stopped -- solution almost zero! [%s:%d]\n", "fasp_solver_dcsr_pminres", 262);
                iter = ERROR_SOLVER_SOLSTAG;
                break;
            }
            KCK(V.nrm2(p1, normuu));
            normuu = std::fabs(alpha) * (normuu / normu2);
            if (normuu < maxdiff) {
                if (stag < MAX_STAG && PrtLvl >= PRINT_MORE) {
                    std::printf("||u-u'|| = %.10e and the comp. rel. res. = %.10e.\n", normuu, relres);
                    std::printf("### WARNING: Iteration restarted -- stagnation! [%s:%d]\n", "fasp_solver_dcsr_pminres", 276);
                }
                KCK(V.resid(u, b, r));
                KCK(resnorm());
                if (PrtLvl >= PRINT_MORE) std::printf("### WARNING: The actual relative residual = %.10e!\n", relres);
                if (relres < tol) break;
                if (stag >= MAX_STAG) {
                    if (PrtLvl > PRINT_MIN)
                        std::printf("### WARNING: Iteration stopped -- staggnation! [%s:%d]\n", "fasp_solver_dcsr_pminres", 318);
                    iter = ERROR_SOLVER_STAG;
                    break;
                }
                ++stag;
                KCK(restart());
            }
        }

        if (relres < tol) {  // Check III (:376-447)
            if (PrtLvl >= PRINT_MORE) std::printf("### WARNING: The computed relative residual = %.10e!\n", relres);
            KCK(V.resid(u, b, r));
            KCK(resnorm());
            if (PrtLvl >= PRINT_MORE) std::printf("### WARNING: The actual relative residual = %.10e!\n", relres);
            if (relres < tol) break;
            if (more_step >= MAX_RESTART) {
                if (PrtLvl > PRINT_MIN)
                    std::printf("### WARNING: The tolerence might be too small! [%s:%d]\n", "fasp_solver_dcsr_pminres", 412);
                iter = ERROR_SOLVER_TOLSMALL;
                break;
            }
            ++more_step;
            KCK(restart());
        }
        absres0 = absres;
    }
FINISHED:
    if (PrtLvl > PRINT_NONE) {
        if (iter > MaxIt) std::printf("### WARNING: MaxIt = %d reached with relative residual %.10e.\n", MaxIt, relres);
        else if (iter >= 0) std::printf("Number of iterations = %d with relative residual %.10e.\n", iter, relres);
    }
    if (out) { out->relres = relres; out->absres = absres; out->normr0 = normr0; }
    HIPCK(hipStreamSynchronize(V.s));
    return iter > MaxIt ? ERROR_SOLVER_MAXIT : iter;
}

// fasp_solver_pminres, KryPminres.c:1283-1650: the OLDER text of MinRes the reference keeps for mxv_matfree (oracle:
// minres_mf_core).  Against minres_device: no iteration-0 line, |<r, B r>| at the start, absres is always ||r||_2, the
// solution / stagnation checks run in every iteration, and the two restart branches take tz = tp WITHOUT the
// preconditioner (:1524-1527, :1605-1608 read `if (pc == NULL) pc->fct(...) else cp`; without a preconditioner the
// reference dereferences the null pointer there -- here that case copies as well).
static int minres_mf_device(KOps& K, const double* b, double* u, double tol, double abstol, int MaxIt, int StopType,
                            int PrtLvl, PcgOut* out)
{
    KVecOps V(K);
    const int m = V.m;
    const double maxdiff = tol * STAG_RATIO, sol_inf_tol = SMALLREAL;
    int iter = 0, stag = 1, more_step = 1;
    double absres0 = BIGREAL, absres = BIGREAL, normr0 = BIGREAL, relres = BIGREAL;
    double normu2 = BIGREAL, normuu, normp, factor, alpha, alpha0, alpha1, temp2, red[8];
    KCK(V.ensure(11));
    double *p0 = V.vec(0), *p1 = V.vec(1), *p2 = V.vec(2), *z0 = V.vec(3), *z1 = V.vec(4), *t0 = V.vec(5),
           *t1 = V.vec(6), *t = V.vec(7), *tp = V.vec(8), *tz = V.vec(9), *r = V.vec(10);
    auto recheck = [&]() -> int {  // :1476-1497 == :1553-1573 (no default case)
        KCK(V.resid(u, b, r));
        KCK(V.dot(r, r, temp2));
        absres = std::sqrt(temp2);
        switch (StopType) {
            case STOP_REL_RES: relres = std::sqrt(temp2) / normr0; break;
            case STOP_REL_PRECRES:
                KCK(V.pc(r, t)); KCK(V.dot(r, t, temp2)); temp2 = std::fabs(temp2);
                relres = std::sqrt(temp2) / normr0; break;
            case STOP_MOD_REL_RES: relres = std::sqrt(temp2) / normu2; break;
        }
        return 0;
    };
    auto restart = [&]() -> int {  // :1512-1543 == :1593-1624
        KCK(V.zero(p0));
        KCK(V.pc(r, p1));
        KCK(V.mxv(p1, tp));
        KCK(V.cp(tz, tp));
        KCK(V.dot(tz, tp, normp));
        normp = std::sqrt(normp);
        KCK(V.cp(t, p1));
        KCK(V.zero(t0)); KCK(V.zero(z0)); KCK(V.zero(t1)); KCK(V.zero(z1)); KCK(V.zero(p1));
        d_axpy(m, 1 / normp, t, p1);
        d_axpy(m, 1 / normp, tp, t1);
        d_axpy(m, 1 / normp, tz, z1);
        return 0;
    };
    if (PrtLvl > PRINT_NONE) std::printf("\nCalling MinRes solver (MatFree) ...\n");
    KCK(V.zero(p0));
    KCK(V.resid(u, b, r));
    KCK(V.pc(r, p1));
    switch (StopType) {
        case STOP_REL_PRECRES:
            KCK(V.dot(r, p1, temp2)); absres0 = std::sqrt(std::fabs(temp2));
            normr0 = std::max(SMALLREAL, absres0); relres = absres0 / normr0; break;
        case STOP_MOD_REL_RES:
            KCK(V.nrm2(r, absres0)); KCK(V.nrm2(u, normu2)); normu2 = std::max(SMALLREAL, normu2);
            relres = absres0 / normu2; break;
        default:
            KCK(V.nrm2(r, absres0)); normr0 = std::max(SMALLREAL, absres0); relres = absres0 / normr0; break;
    }
    if (relres < tol || absres0 < abstol) goto FINISHED;
    KCK(V.mxv(p1, tp));
    KCK(V.pc(tp, tz));
    KCK(V.dot(tz, tp, normp));
    normp = std::sqrt(std::fabs(normp));
    KCK(V.cp(t, p1));
    KCK(V.zero(p1));
    d_axpy(m, 1 / normp, t, p1);
    KCK(V.zero(t0)); KCK(V.zero(z0)); KCK(V.zero(t1)); KCK(V.zero(z1));
    d_axpy(m, 1.0 / normp, tp, t1);
    d_axpy(m, 1.0 / normp, tz, z1);

    while (iter++ < MaxIt) {
        KCK(V.dot(r, z1, alpha));
        d_axpy(m, alpha, p1, u);
        d_axpy(m, -alpha, t1, r);
        KCK(V.mxv(z1, t));
        KCK(V.dot(z1, t, alpha1));
        KCK(V.mxv(z0, t));
        KCK(V.dot(z1, t, alpha0));
        KCK(V.cp(p2, z1));
        d_axpy(m, -alpha1, p1, p2);
        d_axpy(m, -alpha0, p0, p2);
        KCK(V.mxv(p2, tp));
        KCK(V.pc(tp, tz));
        KCK(V.dot(tz, tp, normp));
        normp = std::sqrt(std::fabs(normp));
        KCK(V.cp(t, p2));
        KCK(V.zero(p2));
        d_axpy(m, 1 / normp, t, p2);
        KCK(V.cp(p0, p1)); KCK(V.cp(p1, p2)); KCK(V.cp(t0, t1)); KCK(V.cp(z0, z1));
        KCK(V.zero(t1)); KCK(V.zero(z1));
        d_axpy(m, 1 / normp, tp, t1);
        d_axpy(m, 1 / normp, tz, z1);
        KCK(V.dot(r, r, temp2));
        absres = std::sqrt(temp2);
        if (d_norms(m, u, red, K.dist) < 0) return ERROR_MISC;  // ||u||^2, max|u|
        normu2 = std::sqrt(red[0]);
        switch (StopType) {
            case STOP_REL_PRECRES:
                KCK(V.pc(r, t)); KCK(V.dot(r, t, temp2)); temp2 = std::fabs(temp2);
                relres = std::sqrt(temp2) / normr0; break;
            case STOP_MOD_REL_RES: relres = std::sqrt(temp2) / normu2; break;
            default: relres = std::sqrt(temp2) / normr0; break;
        }
        factor = absres / absres0;
        itinfo(PrtLvl, StopType, iter, relres, absres, factor);
        if (red[1] <= sol_inf_tol) {
            if (PrtLvl > PRINT_MIN)
                std::printf("### WARNING: Iteration stopped -- solution almost zero! [%s:%d]\n", "fasp_solver_pminres", 1460);
            iter = ERROR_SOLVER_SOLSTAG;
            break;
        }
        KCK(V.nrm2(p1, normuu));
        normuu = std::fabs(alpha) * (normuu / normu2);
        if (normuu < maxdiff) {
            if (stag < MAX_STAG && PrtLvl >= PRINT_MORE) {
                std::printf("||u-u'|| = %.10e and the comp. rel. res. = %.10e.\n", normuu, relres);
                std::printf("### WARNING: Iteration restarted -- stagnation! [%s:%d]\n", "fasp_solver_pminres", 1473);
            }
            KCK(recheck());
            if (PrtLvl >= PRINT_MORE) std::printf("### WARNING: The actual relative residual = %.10e!\n", relres);
            if (relres < tol) break;
            if (stag >= MAX_STAG) {
                if (PrtLvl > PRINT_MIN)
                    std::printf("### WARNING: Iteration stopped -- staggnation! [%s:%d]\n", "fasp_solver_pminres", 1505);
                iter = ERROR_SOLVER_STAG;
                break;
            }
            ++stag;
            KCK(restart());
        }
        if (relres < tol) {
            if (PrtLvl >= PRINT_MORE) std::printf("### WARNING: The computed relative residual = %.10e!\n", relres);
            KCK(recheck());
            if (PrtLvl >= PRINT_MORE) std::printf("### WARNING: The actual relative residual = %.10e!\n", relres);
            if (relres < tol) break;
            if (more_step >= MAX_RESTART) {
                if (PrtLvl > PRINT_MIN)
                    std::printf("### WARNING: The tolerence might be too small! [%s:%d]\n", "fasp_solver_pminres", 1581);
                iter = ERROR_SOLVER_TOLSMALL;
                break;
            }
            if (PrtLvl > PRINT_NONE)
                std::printf("### WARNING: Iteration restarted -- stagnation! [%s:%d]\n", "fasp_solver_pminres", 1587);
            ++more_step;
            KCK(restart());
        }
        absres0 = absres;
    }
FINISHED:
    if (PrtLvl > PRINT_NONE) {
        if (iter > MaxIt) std::printf("### WARNING: MaxIt = %d reached with relative residual %.10e.\n", MaxIt, relres);
        else if (iter >= 0) std::printf("Number of iterations = %d with relative residual %.10e.\n", iter, relres);
    }
    if (out) { out->relres = relres; out->absres = absres; out->normr0 = normr0; }
    HIPCK(hipStreamSynchronize(V.s));
    return iter > MaxIt ? ERROR_SOLVER_MAXIT : iter;
}

// fasp_solver_dcsr_pgcg, KryPgcg.c:60-195.  The reference allocates all MaxIt search directions
// up front; here a direction is allocated when its iteration is reached.
static int gcg_device(KOps& K, const double* b, double* u, double tol, double abstol, int MaxIt, int StopType,
                      int PrtLvl, Hist* hist, PcgOut* out)
{
    KVecOps V(K);
    const int m = V.m;
    int iter = 0, i;
    double absres0 = BIGREAL, absres = BIGREAL, relres = BIGREAL, normb = BIGREAL, alpha, factor, num, den, beta;
    KCK(V.ensure(4));
    double *r = V.vec(0), *Br = V.vec(1), *Ap = V.vec(2);
    auto P = [&](int k) { return V.vec(3 + (size_t)k); };
    auto vmv = [&](double* x, const double* y, double& val) -> int {  // y^T A x, BlaSpmvCSR.c:839
        KCK(V.mxv(x, Ap));
        return V.dot(y, Ap, val);
    };
    auto step = [&](double* p) -> int {  // alpha = (r,p)/(p,Ap); u += alpha p; r -= alpha A p
        KCK(V.dot(r, p, num));
        KCK(vmv(p, p, den));
        alpha = num / den;
        d_axpy(m, alpha, p, u);
        d_axpy(m, -1.0 * alpha, Ap, r);  // Ap still holds A p
        KCK(V.nrm2(r, absres));
        factor = absres / absres0;
        relres = absres / normb;
        itinfo(PrtLvl, StopType, iter, relres, absres, factor);
        if (hist) hist->push(absres);
        return 0;
    };
    if (PrtLvl > PRINT_NONE) std::printf("\nCalling GCG solver (%s) ...\n", K.fmt);
    KCK(V.nrm2(b, normb));
    KCK(V.resid(u, b, r));
    KCK(V.pc(r, P(0)));
    KCK(step(P(0)));
    absres0 = absres;
    for (iter = 1; iter < MaxIt; iter++) {
        KCK(V.ensure(4 + (size_t)iter));
        r = V.vec(0); Br = V.vec(1); Ap = V.vec(2);
        double* pi = P(iter);
        KCK(V.pc(r, Br));
        KCK(V.cp(pi, Br));
        for (i = 0; i < iter; i++) {
            KCK(vmv(Br, P(i), num));
            KCK(vmv(P(i), P(i), den));
            beta = (-1.0) * (num / den);
            d_axpy(m, beta, P(i), pi);
        }
        KCK(step(pi));
        if (relres < tol || absres < abstol) break;
        absres0 = absres;
    }
    if (PrtLvl > PRINT_NONE) {
        if (iter > MaxIt) std::printf("### WARNING: MaxIt = %d reached with relative residual %.10e.\n", MaxIt, relres);
        else if (iter >= 0) std::printf("Number of iterations = %d with relative residual %.10e.\n", iter, relres);
    }
    if (out) { out->relres = relres; out->absres = absres; out->normr0 = normb; }
    HIPCK(hipStreamSynchronize(V.s));
    return iter > MaxIt ? ERROR_SOLVER_MAXIT : iter;
}

// fasp_solver_dcsr_pgcr, KryPgcr.c:55-425 (+ dense_aAtxpby :450)
static int gcr_device(KOps& K, const double* b, double* x, double tol, double abstol, int MaxIt, int restart_in,
                      int StopType, int PrtLvl, Hist* hist, PcgOut* out)
{
    KVecOps V(K);
    const int n = V.m;
    int iter = 0, i, j, k, rst = -1;
    double gamma, alpha, beta, checktol, absres0 = BIGREAL, absres = BIGREAL, relres = BIGREAL, prev;
    const int Restart = std::min(restart_in, MaxIt);
    (void)abstol;
    KCK(V.ensure(1 + 2 * (size_t)std::max(Restart, 0)));
    double* r = V.vec(0);
    auto Z = [&](int q) { return V.vec(1 + (size_t)q); };
    auto Cv = [&](int q) { return V.vec(1 + (size_t)Restart + (size_t)q); };
    std::vector<double> alp((size_t)std::max(Restart, 1)), tmpx((size_t)std::max(Restart, 1));
    std::vector<std::vector<double>> h((size_t)std::max(Restart, 1), std::vector<double>((size_t)std::max(Restart, 1), 0.0));
    if (PrtLvl > PRINT_NONE) std::printf("\nCalling GCR solver (%s) ...\n", K.fmt);
    KCK(V.resid(x, b, r));
    KCK(V.dot(r, r, absres));
    absres0 = std::max(SMALLREAL, absres);
    relres = absres / absres0;
    itinfo(PrtLvl, StopType, 0, relres, std::sqrt(absres0), 0.0);
    if (hist) hist->push(std::sqrt(absres));
    prev = relres;
    checktol = std::max(tol * tol * absres0, absres * 1.0e-4);
    while (iter < MaxIt && std::sqrt(relres) > tol) {
        i = -1;
        rst++;
        while (i < Restart - 1 && iter < MaxIt) {
            i++;
            iter++;
            KCK(V.pc(r, Z(i)));
            KCK(V.mxv(Z(i), Cv(i)));
            for (j = 0; j < i; j++) {  // modified Gram-Schmidt
                KCK(V.dot(Cv(j), Cv(i), gamma));
                h[i][j] = gamma / h[j][j];
                d_axpy(n, -h[i][j], Cv(j), Cv(i));
            }
            KCK(V.dot(Cv(i), Cv(i), gamma));
            h[i][i] = gamma;
            KCK(V.dot(Cv(i), r, alpha));
            beta = alpha / gamma;
            alp[i] = beta;
            d_axpy(n, -beta, Cv(i), r);
            absres = absres - alpha * alpha / gamma;
            if (absres < checktol) {
                KCK(V.dot(r, r, absres));
                checktol = std::max(tol * tol * absres0, absres * 1.0e-4);
            }
            relres = absres / absres0;
            itinfo(PrtLvl, StopType, iter, std::sqrt(relres), std::sqrt(absres), std::sqrt(relres / prev));
            if (hist) hist->push(std::sqrt(absres));
            prev = relres;
            if (std::sqrt(relres) < tol) break;
        }
        for (k = i; k >= 0; k--) {
            tmpx[k] = alp[k];
            for (j = 0; j < k; ++j) alp[j] -= h[k][j] * tmpx[k];
        }
        // dense_aAtxpby(n, i+1, z, 1.0, tmpx, rst == 0 ? 0.0 : 1.0, x): columns scaled in place,
        // summed into column 0 one after the other, x = 1.0 z_0 + beta x (the first cycle overwrites x)
        for (k = 0; k < i + 1; k++) d_scale(n, tmpx[k], Z(k));
        for (j = 1; j < i + 1; j++) d_axpy(n, 1.0, Z(j), Z(0));
        d_axpby(n, 1.0, Z(0), rst == 0 ? 0.0 : 1.0, x);
    }
    if (PrtLvl > PRINT_NONE) {
        if (iter > MaxIt) std::printf("### WARNING: MaxIt = %d reached with relative residual %.10e.\n", MaxIt, std::sqrt(relres));
        else if (iter >= 0) std::printf("Number of iterations = %d with relative residual %.10e.\n", iter, std::sqrt(relres));
    }
    if (out) { out->relres = std::sqrt(relres); out->absres = std::sqrt(absres); out->normr0 = std::sqrt(absres0); }
    HIPCK(hipStreamSynchronize(V.s));
    return iter >= MaxIt ? ERROR_SOLVER_MAXIT : iter;
}

// ---------------------------------------------------------------------------
// Matrix-free family (SolMatFree.c): the reference keeps older texts of CG and of the GMRES
// variants for the mxv_matfree interface; they are restated separately (oracle: pcg_mf_core,
// gmres_mf_core).  BiCGstab and GCG perform the arithmetic of their CSR texts.
// ---------------------------------------------------------------------------
// fasp_solver_pcg, KryPcg.c:1260-1540
static int pcg_mf_device(KOps& K, const double* b, double* u, double tol, double abstol, int MaxIt, int StopType,
                         int PrtLvl, PcgOut* out)
{
    KVecOps V(K);
    const int m = V.m;
    const double maxdiff = tol * STAG_RATIO, sol_inf_tol = SMALLREAL;
    int iter = 0, stag = 1, more_step = 1;
    double absres0 = BIGREAL, absres = BIGREAL, relres = BIGREAL, normu = BIGREAL, normr0 = BIGREAL;
    double reldiff, factor, alpha, beta, temp1 = 0.0, temp2, red[8], pp;
    KCK(V.ensure(4));
    double *p = V.vec(0), *z = V.vec(1), *r = V.vec(2), *t = V.vec(3);
    auto rel_from = [&]() -> int {  // relres per stop type from the current r (absres is ||r||_2 throughout)
        switch (StopType) {
            case STOP_REL_PRECRES:
                KCK(V.pc(r, z)); KCK(V.dot(z, r, temp2));
                relres = std::sqrt(std::fabs(temp2)) / normr0; break;
            case STOP_MOD_REL_RES: relres = absres / normu; break;
            default: relres = absres / normr0; break;
        }
        return 0;
    };
    if (PrtLvl > PRINT_NONE) std::printf("\nCalling CG solver (MatFree) ...\n");
    KCK(V.resid(u, b, r));
    KCK(V.pc(r, z));
    switch (StopType) {
        case STOP_REL_PRECRES:
            KCK(V.dot(r, z, temp2)); absres0 = std::sqrt(temp2); normr0 = std::max(SMALLREAL, absres0); relres = absres0 / normr0; break;
        case STOP_MOD_REL_RES:
            KCK(V.nrm2(r, absres0)); KCK(V.nrm2(u, normu)); normu = std::max(SMALLREAL, normu); relres = absres0 / normu; break;
        default:
            KCK(V.nrm2(r, absres0)); normr0 = std::max(SMALLREAL, absres0); relres = absres0 / normr0; break;
    }
    if (relres < tol || absres0 < abstol) goto FINISHED;
    KCK(V.cp(p, z));
    KCK(V.dot(z, r, temp1));
    while (iter++ < MaxIt) {
        KCK(V.mxv(p, t));
        KCK(V.dot(t, p, temp2));
        alpha = temp1 / temp2;
        d_axpy(m, alpha, p, u);
        d_axpy(m, -alpha, t, r);
        KCK(V.nrm2(r, absres));
        factor = absres / absres0;
        KCK(rel_from());
        itinfo(PrtLvl, StopType, iter, relres, absres, factor);
        if (d_norms(m, u, red, K.dist) < 0) return ERROR_MISC;
        if (red[1] <= sol_inf_tol) {
            if (PrtLvl > PRINT_MIN) std::printf("### WARNING: Iteration stopped -- solution almost zero! [%s:%d]\n", "fasp_solver_pcg", 1390);
            iter = ERROR_SOLVER_SOLSTAG;
            break;
        }
        normu = std::sqrt(red[0]);
        KCK(V.nrm2(p, pp));
        reldiff = std::fabs(alpha) * pp / normu;
        if ((stag <= MAX_STAG) & (reldiff < maxdiff)) {
            if (PrtLvl >= PRINT_MORE) {
                std::printf("||u-u'|| = %.10e and the comp. rel. res. = %.10e.\n", reldiff, relres);
                std::printf("### WARNING: Iteration restarted -- stagnation! [%s:%d]\n", "fasp_solver_pcg", 1404);
            }
            KCK(V.resid(u, b, r));
            KCK(V.nrm2(r, absres));
            KCK(rel_from());
            if (PrtLvl >= PRINT_MORE) std::printf("### WARNING: The actual relative residual = %.10e!\n", relres);
            if (relres < tol) break;
            if (stag >= MAX_STAG) {
                if (PrtLvl > PRINT_MIN) std::printf("### WARNING: Iteration stopped -- staggnation! [%s:%d]\n", "fasp_solver_pcg", 1437);
                iter = ERROR_SOLVER_STAG;
                break;
            }
            KCK(V.zero(p));
            ++stag;
        }
        if (relres < tol) {
            if (PrtLvl >= PRINT_MORE) std::printf("### WARNING: The computed relative residual = %.10e!\n", relres);
            KCK(V.resid(u, b, r));
            if (StopType != STOP_REL_PRECRES) KCK(V.nrm2(r, absres));
            KCK(rel_from());
            if (PrtLvl >= PRINT_MORE) std::printf("### WARNING: The actual relative residual = %.10e!\n", relres);
            if (relres < tol) break;
            if (more_step >= MAX_RESTART) {
                if (PrtLvl > PRINT_MIN) std::printf("### WARNING: The tolerence might be too small! [%s:%d]\n", "fasp_solver_pcg", 1487);
                iter = ERROR_SOLVER_TOLSMALL;
                break;
            }
            KCK(V.zero(p));
            ++more_step;
        }
        absres0 = absres;
        if (StopType != STOP_REL_PRECRES) KCK(V.pc(r, z));
        KCK(V.dot(z, r, temp2));
        beta = temp2 / temp1;
        temp1 = temp2;
        d_axpby(m, 1.0, z, beta, p);
    }
FINISHED:
    if (PrtLvl > PRINT_NONE) {
        if (iter > MaxIt) std::printf("### WARNING: MaxIt = %d reached with relative residual %.10e.\n", MaxIt, relres);
        else if (iter >= 0) std::printf("Number of iterations = %d with relative residual %.10e.\n", iter, relres);
    }
    if (out) { out->relres = relres; out->absres = absres; out->normr0 = normr0; }
    HIPCK(hipStreamSynchronize(V.s));
    return iter > MaxIt ? ERROR_SOLVER_MAXIT : iter;
}

// fasp_solver_pgmres / _pvgmres / _pvfgmres for mxv_matfree (KryPgmres.c:1309, KryPvgmres.c:1468,
// KryPvfgmres.c:1026): one text with two switches; stops on ||r|| <= tol ||b||, StopType ignored.
static int gmres_mf_device(KOps& K, bool variable, bool flexible, const double* b, double* x, double tol, int MaxIt,
                           int restart, int StopType, int PrtLvl, PcgOut* out)
{
    KVecOps V(K);
    const int n = V.m, min_iter = 0;
    const double cr_max = 0.99, cr_min = 0.174, epsmac = SMALLREAL;
    int iter = 0, i, j, k;
    double r_norm, b_norm, den_norm, epsilon, gamma, t, cr = 1.0, r_norm_old = 0.0, prev;
    const int d = 3, restart_max = restart, restart_min = 3;
    int Restart = restart;
    const int Restart1 = restart + 1;
    if (restart < 1) return ERROR_INPUT_PAR;
    KCK(V.ensure(2 + (size_t)Restart1 * (flexible ? 2 : 1)));
    double *r = V.vec(0), *w = V.vec(1);
    auto P = [&](int q) { return V.vec(2 + (size_t)q); };
    auto Z = [&](int q) { return V.vec(2 + (size_t)Restart1 + (size_t)q); };
    std::vector<double> rs((size_t)Restart1 + 1), c((size_t)Restart1), sn((size_t)Restart1);
    std::vector<std::vector<double>> hh((size_t)Restart1, std::vector<double>((size_t)restart + 1, 0.0));
    if (PrtLvl > PRINT_NONE)
        std::printf(flexible ? "\nCalling VFGMRes solver (MatFree) ...\n" : variable ? "\nCalling VGMRes solver (MatFree) ...\n"
                                                                                    : "\nCalling GMRes solver (MatFree) ...\n");
    KCK(V.resid(x, b, P(0)));
    KCK(V.nrm2(b, b_norm));
    KCK(V.nrm2(P(0), r_norm));
    prev = r_norm;
    if (PrtLvl >= PRINT_SOME) {
        std::printf("L2 norm of %s = %.10e.\n", "right-hand side", b_norm);
        std::printf("L2 norm of %s = %.10e.\n", "residual", r_norm);
    }
    den_norm = (b_norm > 0.0) ? b_norm : r_norm;
    epsilon = tol * den_norm;
    while (iter < MaxIt) {
        rs[0] = r_norm;
        r_norm_old = r_norm;
        if (r_norm == 0.0) {
            if (out) { out->relres = 0.0; out->absres = 0.0; out->normr0 = den_norm; }
            HIPCK(hipStreamSynchronize(V.s));
            return iter;
        }
        if (variable) {
            if (cr > cr_max || iter == 0) Restart = restart_max;
            else if (cr < cr_min) { /* keep */ }
            else { if (Restart - d > restart_min) Restart -= d; else Restart = restart_max; }
        }
        if (r_norm <= epsilon && iter >= min_iter) {
            KCK(V.resid(x, b, r));
            KCK(V.nrm2(r, r_norm));
            if (r_norm <= epsilon) break;
            if (PrtLvl >= PRINT_SOME) std::printf("### WARNING: False convergence! [%s:%d]\n", "fasp_solver_pvgmres", 1620);
        }
        d_scale(n, 1.0 / r_norm, P(0));
        i = 0;
        while (i < Restart && iter < MaxIt) {
            i++; iter++;
            if (flexible) { KCK(V.pc(P(i - 1), Z(i - 1))); KCK(V.mxv(Z(i - 1), P(i))); }
            else          { KCK(V.pc(P(i - 1), r));        KCK(V.mxv(r, P(i))); }
            for (j = 0; j < i; j++) {  // modified Gram-Schmidt
                KCK(V.dot(P(j), P(i), hh[j][i - 1]));
                d_axpy(n, -hh[j][i - 1], P(j), P(i));
            }
            KCK(V.nrm2(P(i), t));
            hh[i][i - 1] = t;
            if (t != 0.0) d_scale(n, 1.0 / t, P(i));
            for (j = 1; j < i; ++j) {
                t = hh[j - 1][i - 1];
                hh[j - 1][i - 1] = sn[j - 1] * hh[j][i - 1] + c[j - 1] * t;
                hh[j][i - 1] = -sn[j - 1] * t + c[j - 1] * hh[j][i - 1];
            }
            t = hh[i][i - 1] * hh[i][i - 1];
            t += hh[i - 1][i - 1] * hh[i - 1][i - 1];
            gamma = std::sqrt(t);
            if (gamma == 0.0) gamma = epsmac;
            c[i - 1] = hh[i - 1][i - 1] / gamma;
            sn[i - 1] = hh[i][i - 1] / gamma;
            rs[i] = -sn[i - 1] * rs[i - 1];
            rs[i - 1] = c[i - 1] * rs[i - 1];
            hh[i - 1][i - 1] = sn[i - 1] * hh[i][i - 1] + c[i - 1] * hh[i - 1][i - 1];
            r_norm = std::fabs(rs[i]);
            if (b_norm > 0) itinfo(PrtLvl, StopType, iter, r_norm / b_norm, r_norm, r_norm / prev);
            else itinfo(PrtLvl, StopType, iter, r_norm, r_norm, r_norm / prev);
            prev = r_norm;
            if (r_norm <= epsilon && iter >= min_iter) break;
        }
        rs[i - 1] = rs[i - 1] / hh[i - 1][i - 1];
        for (k = i - 2; k >= 0; k--) {
            t = 0.0;
            for (j = k + 1; j < i; j++) t -= hh[k][j] * rs[j];
            t += rs[k];
            rs[k] = t / hh[k][k];
        }
        if (flexible) {
            KCK(V.cp(r, Z(i - 1)));
            d_scale(n, rs[i - 1], r);
            for (j = i - 2; j >= 0; j--) d_axpy(n, rs[j], Z(j), r);
        } else {
            KCK(V.cp(w, P(i - 1)));
            d_scale(n, rs[i - 1], w);
            for (j = i - 2; j >= 0; j--) d_axpy(n, rs[j], P(j), w);
            KCK(V.pc(w, r));
        }
        d_axpy(n, 1.0, r, x);
        if (r_norm <= epsilon && iter >= min_iter) {
            KCK(V.resid(x, b, r));
            KCK(V.nrm2(r, r_norm));
            if (r_norm <= epsilon) break;
            if (PrtLvl >= PRINT_SOME) std::printf("### WARNING: False convergence! [%s:%d]\n", "fasp_solver_pvgmres", 1757);
            KCK(V.cp(P(0), r));
            i = 0;
        }
        for (j = i; j > 0; j--) {
            rs[j - 1] = -sn[j - 1] * rs[j];
            rs[j] = c[j - 1] * rs[j];
        }
        // p[i] += (rs[i] - 1) p[i] is evaluated elementwise as y + a y (BlaArray.c:90), not as a scaling
        if (i) hipLaunchKernelGGL(k_axpy_self, dim3(vec_grid(n)), dim3(BLOCK), 0, V.s, n, rs[i] - 1.0, P(i));
        for (j = i - 1; j > 0; j--) d_axpy(n, rs[j], P(j), P(i));
        if (i) {
            hipLaunchKernelGGL(k_axpy_self, dim3(vec_grid(n)), dim3(BLOCK), 0, V.s, n, rs[0] - 1.0, P(0));
            d_axpy(n, 1.0, P(i), P(0));
        }
        if (variable) cr = r_norm / r_norm_old;
    }
    if (PrtLvl > PRINT_NONE) {
        if (iter > MaxIt) std::printf("### WARNING: MaxIt = %d reached with relative residual %.10e.\n", MaxIt, r_norm);
        else if (iter >= 0) std::printf("Number of iterations = %d with relative residual %.10e.\n", iter, r_norm);
    }
    if (out) { out->relres = r_norm / den_norm; out->absres = r_norm; out->normr0 = den_norm; }
    HIPCK(hipStreamSynchronize(V.s));
    return iter >= MaxIt ? ERROR_SOLVER_MAXIT : iter;
}

