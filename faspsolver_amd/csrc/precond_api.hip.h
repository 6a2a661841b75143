// precond_api.hip.h -- the reference's own preconditioner objects on top of the resident hierarchy:
// AMG_data / precond_data in the reference's layout, fasp_precond_setup (PreCSR.c:46), fasp_precond_amg
// (PreCSR.c:416) and its cycle variants, fasp_amg_data_create / _free (PreDataInit.c:64 / :101), the parameter
// converters (AuxParam.c:782 / :816), the small host utilities a tutorial-style caller uses around them, and
// stand-alone sweeps of the sequential smoothers (ItrSmootherCSR.c:251 / :932 / :1509).
// Part of the single translation unit solver.hip (included inside its extern "C" block).

namespace {
// AMG_data arrays handed out by fasp_precond_setup -> the handle that owns the host arrays they show and the
// device copy.  Looked up by fasp_precond_amg, fasp_amg_data_free and the Krylov plug-in level.
struct MglEntry { AMG_data* mgl; fasp_hip_amg* h; };
std::vector<MglEntry> g_mgl_registry;

fasp_hip_amg* handle_of_mgl(const AMG_data* mgl)
{
    for (const MglEntry& e : g_mgl_registry)
        if (e.mgl == mgl) return e.h;
    return nullptr;
}

void prec_to_handle_param(fasp_hip_amg* h, const precond_data* pc)
{
    AMG_param& p = h->param;  // fasp_param_prec_to_amg (AuxParam.c:816) + maxit, which fasp_precond_amg reads from pcdata
    p.AMG_type = pc->AMG_type; p.print_level = pc->print_level; p.cycle_type = pc->cycle_type;
    p.smoother = pc->smoother; p.smooth_order = pc->smooth_order; p.presmooth_iter = pc->presmooth_iter;
    p.postsmooth_iter = pc->postsmooth_iter; p.relaxation = pc->relaxation;
    p.polynomial_degree = pc->polynomial_degree; p.coarse_solver = pc->coarse_solver;
    p.coarse_scaling = pc->coarse_scaling; p.amli_degree = pc->amli_degree;
    p.nl_amli_krylov_type = pc->nl_amli_krylov_type; p.tentative_smooth = pc->tentative_smooth;
    p.maxit = pc->maxit;
}
}  // namespace

void fasp_precond_amg(double* r, double* z, void* data);
void fasp_precond_famg(double* r, double* z, void* data);
void fasp_precond_amli(double* r, double* z, void* data);
void fasp_precond_namli(double* r, double* z, void* data);

// The device hierarchy behind a reference-style `precond` (nullptr: not one of ours).  Also re-reads the cycle
// parameters from the caller's precond_data, as every fasp_precond_amg call does.
static fasp_hip_amg* amg_handle_of_precond(precond* pc)
{
    if (!pc || !pc->data) return nullptr;
    if (pc->fct != fasp_precond_amg && pc->fct != fasp_precond_famg && pc->fct != fasp_precond_amli &&
        pc->fct != fasp_precond_namli)
        return nullptr;
    precond_data* pd = static_cast<precond_data*>(pc->data);
    fasp_hip_amg* h = handle_of_mgl(pd->mgl_data);
    if (!h) return nullptr;
    prec_to_handle_param(h, pd);
    h->use_fmg = pc->fct == fasp_precond_famg;
    return h;
}

// ---- AuxMemory.c / AuxVector.c / BlaSparseCSR.c: what a caller of the objects above needs ----
void* fasp_mem_calloc(const unsigned int size, const unsigned int type)
{
    const size_t tsize = (size_t)size * type;
    if (tsize == 0) return nullptr;
    void* mem = std::calloc(size, type);
    if (!mem) { std::printf("### WARNING: Trying to allocate %.3lfMB RAM...\n", (double)tsize / 1048576.0);
                std::printf("### ERROR: Failed to allocate %u Bytes!\n", size * type); std::exit(ERROR_ALLOC_MEM); }
    return mem;
}
void    fasp_mem_free(void* mem) { if (mem) std::free(mem); }
void    fasp_dvec_alloc(const int m, dvector* u) { u->row = m; u->val = (double*)fasp_mem_calloc((unsigned)m, sizeof(double)); }
void    fasp_dvec_set(int n, dvector* x, const double val)
{
    if (n > 0) x->row = n; else n = x->row;
    for (int i = 0; i < n; ++i) x->val[i] = val;
}
void    fasp_dvec_free(dvector* u) { if (!u) return; std::free(u->val); u->row = 0; u->val = nullptr; }
dvector fasp_dvec_create(const int m) { dvector u; u.row = m; u.val = (double*)fasp_mem_calloc((unsigned)m, sizeof(double)); return u; }
dCSRmat fasp_dcsr_create(const int m, const int n, const int nnz)
{
    dCSRmat A;
    std::memset(&A, 0, sizeof(A));
    if (m <= 0 || n <= 0) { std::printf("### ERROR: Matrix dim %d, %d must be positive! [%s]\n", m, n, __func__); return A; }
    A.IA = (int*)fasp_mem_calloc((unsigned)m + 1, sizeof(int));
    if (nnz > 0) { A.JA = (int*)fasp_mem_calloc((unsigned)nnz, sizeof(int)); A.val = (double*)fasp_mem_calloc((unsigned)nnz, sizeof(double)); }
    A.row = m; A.col = n; A.nnz = nnz;
    return A;
}
void fasp_dcsr_free(dCSRmat* A)
{
    FASP_ENTRY();
    if (!A) return;
    std::free(A->IA); std::free(A->JA); std::free(A->val);
    A->row = A->col = A->nnz = 0; A->IA = A->JA = nullptr; A->val = nullptr;
}

void fasp_param_amg_to_prec(precond_data* pcdata, const AMG_param* amgparam)
{
    FASP_ENTRY();
    pcdata->AMG_type = amgparam->AMG_type; pcdata->print_level = amgparam->print_level;
    pcdata->maxit = amgparam->maxit; pcdata->max_levels = amgparam->max_levels; pcdata->tol = amgparam->tol;
    pcdata->cycle_type = amgparam->cycle_type; pcdata->smoother = amgparam->smoother;
    pcdata->smooth_order = amgparam->smooth_order; pcdata->presmooth_iter = amgparam->presmooth_iter;
    pcdata->postsmooth_iter = amgparam->postsmooth_iter; pcdata->coarsening_type = amgparam->coarsening_type;
    pcdata->coarse_solver = amgparam->coarse_solver; pcdata->relaxation = amgparam->relaxation;
    pcdata->polynomial_degree = amgparam->polynomial_degree; pcdata->coarse_scaling = amgparam->coarse_scaling;
    pcdata->amli_degree = amgparam->amli_degree; pcdata->amli_coef = amgparam->amli_coef;
    pcdata->nl_amli_krylov_type = amgparam->nl_amli_krylov_type; pcdata->tentative_smooth = amgparam->tentative_smooth;
}
void fasp_param_prec_to_amg(AMG_param* amgparam, const precond_data* pcdata)
{
    FASP_ENTRY();
    amgparam->AMG_type = pcdata->AMG_type; amgparam->print_level = pcdata->print_level;
    amgparam->cycle_type = pcdata->cycle_type; amgparam->smoother = pcdata->smoother;
    amgparam->smooth_order = pcdata->smooth_order; amgparam->presmooth_iter = pcdata->presmooth_iter;
    amgparam->postsmooth_iter = pcdata->postsmooth_iter; amgparam->relaxation = pcdata->relaxation;
    amgparam->polynomial_degree = pcdata->polynomial_degree; amgparam->coarse_solver = pcdata->coarse_solver;
    amgparam->coarse_scaling = pcdata->coarse_scaling; amgparam->amli_degree = pcdata->amli_degree;
    amgparam->amli_coef = pcdata->amli_coef; amgparam->nl_amli_krylov_type = pcdata->nl_amli_krylov_type;
    amgparam->tentative_smooth = pcdata->tentative_smooth;
    amgparam->ILU_levels = pcdata->mgl_data ? pcdata->mgl_data->ILU_levels : 0;
}

// PreDataInit.c:64
AMG_data* fasp_amg_data_create(short max_levels)
{
    max_levels = std::max<short>(1, max_levels);
    AMG_data* mgl = (AMG_data*)fasp_mem_calloc((unsigned)max_levels, sizeof(AMG_data));
    for (int i = 0; i < max_levels; ++i) {
        mgl[i].max_levels = max_levels;
        mgl[i].num_levels = 0;
        mgl[i].near_kernel_dim = 0;
        mgl[i].near_kernel_basis = nullptr;
        mgl[i].cycle_type = 0;
    }
    return mgl;
}

// PreDataInit.c:101.  The operators of a hierarchy built by fasp_precond_setup are views of the handle's host
// arrays: they go with the handle (and its device copy); b / x / w were allocated here and are freed here.
void fasp_amg_data_free(AMG_data* mgl, AMG_param* param)
{
    FASP_ENTRY();
    if (!mgl) return;
    const int nl = std::max<int>(1, mgl[0].num_levels);
    fasp_hip_amg* h = handle_of_mgl(mgl);
    for (int i = 0; i < nl; ++i) {
        if (!h) {
            fasp_dcsr_free(&mgl[i].A);
            if (nl > 1) { fasp_dcsr_free(&mgl[i].P); fasp_dcsr_free(&mgl[i].R); }
            std::free(mgl[i].cfmark.val);
        }
        fasp_dvec_free(&mgl[i].b); fasp_dvec_free(&mgl[i].x); fasp_dvec_free(&mgl[i].w);
    }
    if (h) {
        for (size_t q = 0; q < g_mgl_registry.size(); ++q)
            if (g_mgl_registry[q].mgl == mgl) { g_mgl_registry.erase(g_mgl_registry.begin() + (long)q); break; }
        fasp_hip_amg_destroy(h);
    }
    std::free(mgl);
    if (param && param->cycle_type == AMLI_CYCLE) { std::free(param->amli_coef); param->amli_coef = nullptr; }
}

// PreCSR.c:46
precond* fasp_precond_setup(const short precond_type, AMG_param* amgparam, ILU_param* iluparam, dCSRmat* A)
{
    FASP_ENTRY();
    (void)iluparam;
    if (precond_type == PREC_NULL) return nullptr;
    if (!A) { std::printf("### ERROR: fasp_precond_setup: A == NULL\n"); std::exit(ERROR_INPUT_PAR); }
    precond* pc = (precond*)fasp_mem_calloc(1, sizeof(precond));
    if (precond_type == PREC_DIAG) {  // PreCSR.c:132: a dvector of the diagonal, fasp_precond_diag
        dvector* diag = (dvector*)fasp_mem_calloc(1, sizeof(dvector));
        const int n = std::min(A->row, A->col);
        fasp_dvec_alloc(n, diag);
        for (int i = 0; i < n; ++i)  // fasp_dcsr_getdiag (BlaSparseCSR.c:537): first diagonal hit of the row, 0 if none
            for (int k = A->IA[i]; k < A->IA[i + 1]; ++k)
                if (A->JA[k] == i) { diag->val[i] = A->val[k]; break; }
        pc->data = diag;
        pc->fct = fasp_precond_diag;
        return pc;
    }
    if (precond_type != PREC_AMG && precond_type != PREC_FMG) {
        std::printf("### ERROR: fasp_precond_setup: preconditioner type %d (ILU / Schwarz) has no device path in libfasp_hip\n",
                    (int)precond_type);
        std::exit(ERROR_SOLVER_PRECTYPE);
    }
    if (!amgparam) { std::printf("### ERROR: fasp_precond_setup: amgparam == NULL\n"); std::exit(ERROR_INPUT_PAR); }
    fasp_hip_amg* h = nullptr;
    const int st = fasp_hip_amg_create(&h, A, amgparam);
    if (st < 0) {
        std::printf("### ERROR: fasp_precond_setup: AMG setup failed with status %d\n", st);
        std::exit(st);
    }
    const int nl = (int)h->H.L.size();
    AMG_data* mgl = fasp_amg_data_create((short)std::max<int>(amgparam->max_levels, nl));
    for (int l = 0; l < nl; ++l) {
        const HostLevel& L = h->H.L[l];
        mgl[l].A = L.A.view();
        if (L.has_coarse) { mgl[l].P = L.P.view(); mgl[l].R = L.R.view(); }
        if (L.cfmark.n) { mgl[l].cfmark.row = (int)L.cfmark.n; mgl[l].cfmark.val = const_cast<int*>(L.cfmark.data()); }
        const int m = L.A.row;
        mgl[l].num_levels = (short)nl;
        mgl[l].b = fasp_dvec_create(m);
        mgl[l].x = fasp_dvec_create(m);
        // work space as the setups leave it (PreAMGSetupRS.c:316-334): m on level 0, 2 m below (3 m for the K-cycle)
        const int wmul = l == 0 ? 1 : (amgparam->cycle_type == NL_AMLI_CYCLE ? 3 : 2);
        mgl[l].w = fasp_dvec_create(wmul * m);
        mgl[l].cycle_type = l < (int)h->level_cycle_type.size() ? h->level_cycle_type[l] : 0;
    }
    g_mgl_registry.push_back(MglEntry{mgl, h});
    precond_data* pcdata = (precond_data*)fasp_mem_calloc(1, sizeof(precond_data));
    fasp_param_amg_to_prec(pcdata, amgparam);
    pcdata->max_levels = mgl[0].num_levels;
    pcdata->mgl_data = mgl;
    pc->data = pcdata;
    if (precond_type == PREC_FMG) pc->fct = fasp_precond_famg;
    else if (amgparam->cycle_type == AMLI_CYCLE) pc->fct = fasp_precond_amli;
    else if (amgparam->cycle_type == NL_AMLI_CYCLE) pc->fct = fasp_precond_namli;
    else pc->fct = fasp_precond_amg;
    return pc;
}

namespace {
void precond_apply_common(const char* fn, double* r, double* z, void* data, bool fmg)
{
    precond_data* pd = static_cast<precond_data*>(data);
    fasp_hip_amg* h = pd ? handle_of_mgl(pd->mgl_data) : nullptr;
    if (!h) {
        std::fprintf(stderr, "### ERROR: %s: this precond_data carries no hierarchy built by fasp_precond_setup of libfasp_hip "
                             "(there is no CPU cycle in this library)\n", fn);
        std::exit(ERROR_MISC);
    }
    prec_to_handle_param(h, pd);
    h->use_fmg = fmg;
    if (fasp_hip_precond_amg(h, r, z) < 0) {
        std::fprintf(stderr, "### ERROR: %s: device preconditioner failed\n", fn);
        std::exit(ERROR_MISC);
    }
}
}  // namespace
void fasp_precond_amg(double* r, double* z, void* data) { precond_apply_common(__func__, r, z, data, false); }
void fasp_precond_famg(double* r, double* z, void* data) { precond_apply_common(__func__, r, z, data, true); }
void fasp_precond_amli(double* r, double* z, void* data) { precond_apply_common(__func__, r, z, data, false); }
void fasp_precond_namli(double* r, double* z, void* data) { precond_apply_common(__func__, r, z, data, false); }

// ---- stand-alone sweeps of the sequential smoothers and of L1-diag: host vectors in and out ----
namespace {
void smoother_standalone(const char* fn, dvector* u, int i_1, int i_n, int s, dCSRmat* A, dvector* b, int L, int smoother,
                         double w)
{
    if (!u || !A || !b || A->row != A->col || u->row < A->row || b->row < A->row) {
        std::fprintf(stderr, "### ERROR: %s: inconsistent arguments\n", fn);
        std::exit(ERROR_INPUT_PAR);
    }
    const int n = A->row;
    if (std::min(i_1, i_n) != 0 || std::max(i_1, i_n) != n - 1 || (s != 1 && s != -1)) {
        std::fprintf(stderr, "### ERROR: %s (device): only full sweeps 0..n-1 with step +-1 are supported\n", fn);
        std::exit(ERROR_INPUT_PAR);
    }
    AMG_param p;
    fasp_param_amg_init(&p);
    p.max_levels = 1; p.print_level = 0;  // one level: the operator itself, resident
    p.smoother = (short)smoother; p.smooth_order = NO_ORDER;
    fasp_hip_amg* h = nullptr;
    if (fasp_hip_amg_create(&h, A, &p) < 0) die_no_device(fn);
    DevLevel& D = h->L[0];
    D.b = h->b;  // the level-0 right-hand side aliases the handle's Krylov vector
    (void)hipMemcpyAsync(D.b, b->val, sizeof(double) * n, hipMemcpyHostToDevice, g_ctx.stream);
    (void)hipMemcpyAsync(D.x, u->val, sizeof(double) * n, hipMemcpyHostToDevice, g_ctx.stream);
    D.x_zero = false;
    const int st = smooth(h, 0, s < 0, smoother, NO_ORDER, L, w, 0);
    (void)hipStreamSynchronize(g_ctx.stream);
    (void)hipMemcpy(u->val, D.x, sizeof(double) * n, hipMemcpyDeviceToHost);
    fasp_hip_amg_destroy(h);
    if (st < 0) { std::fprintf(stderr, "### ERROR: %s: sweep failed (%d)\n", fn, st); std::exit(st); }
}
}  // namespace
// ItrSmootherCSR.c:251: s = +1 ascending, -1 descending, u_i = t * (1 / a_ii)
void fasp_smoother_dcsr_gs(dvector* u, const int i_1, const int i_n, const int s, dCSRmat* A, dvector* b, int L)
{
    FASP_ENTRY();
    smoother_standalone(__func__, u, i_1, i_n, s, A, b, L, SMOOTHER_GS, 1.0);
}
// ItrSmootherCSR.c:932
void fasp_smoother_dcsr_sor(dvector* u, const int i_1, const int i_n, const int s, dCSRmat* A, dvector* b, int L, const double w)
{
    FASP_ENTRY();
    smoother_standalone(__func__, u, i_1, i_n, s, A, b, L, SMOOTHER_SOR, w);
}
// ItrSmootherCSR.c:1509 (order independent)
void fasp_smoother_dcsr_L1diag(dvector* u, const int i_1, const int i_n, const int s, dCSRmat* A, dvector* b, int L)
{
    FASP_ENTRY();
    smoother_standalone(__func__, u, i_1, i_n, s, A, b, L, SMOOTHER_L1DIAG, 1.0);
}
