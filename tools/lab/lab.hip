// lab.hip -- kernel laboratory (development tool, not part of the product): the whole solver
// translation unit plus candidate kernels, timed against each other in ONE process on the same
// operator, each result compared with a host evaluation of the reference's row sums.
//
//   build:  make -C tools/lab            (hipcc, gfx950; links the host objects of faspsolver_amd/csrc)
//   run:    tools/lab/lab <n> <what> [reps]
//           what = l0      level-0 operator of P7(n): plain CSR + coded kernels, old and new
//                  var     D*A*D with a smooth non-constant D (no coding applies)
//                  copy    device copy / read / triad ceilings
#define FASP_LAB_DEBUG 1
#include "../../faspsolver_amd/csrc/solver.hip"
#include "../../faspsolver_amd/csrc/kernels2.hip.h"

#include <string>

using namespace fasp;

static double time_launch(const std::function<void()>& f, int reps)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); f();
    (void)hipEventRecord(e0, g_ctx.stream);
    for (int i = 0; i < reps; ++i) f();
    (void)hipEventRecord(e1, g_ctx.stream);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return (double)ms / reps * 1e3;  // us
}

struct Case {
    HostCSR H;
    DevCSR  D;
    std::vector<double> x, b, diag, yref_mxv, yref_jac;
    double *dx = nullptr, *dy = nullptr, *db = nullptr, *ddiag = nullptr;
    double  bytes = 0;
};

static void host_copy(const dCSRmat& A, HostCSR& H, bool scale)
{
    H.row = A.row; H.col = A.col; H.nnz = A.nnz;
    H.ia.alloc((size_t)A.row + 1); H.ja.alloc((size_t)A.nnz); H.val.alloc((size_t)A.nnz);
    std::memcpy(H.ia.data(), A.IA, sizeof(int) * ((size_t)A.row + 1));
    std::memcpy(H.ja.data(), A.JA, sizeof(int) * (size_t)A.nnz);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < A.row; ++i) {
        const double di = 1.0 + 0.5 * std::sin(1e-3 * i);
        for (int k = A.IA[i]; k < A.IA[i + 1]; ++k) {
            const double dj = 1.0 + 0.5 * std::sin(1e-3 * A.JA[k]);
            H.val[k] = scale ? di * A.val[k] * dj : A.val[k];
        }
    }
}

static void prepare(Case& C, double omega)
{
    const int n = C.H.row;
    C.x.resize(n); C.b.resize(n); C.diag.resize(n); C.yref_mxv.resize(n); C.yref_jac.resize(n);
    unsigned s = 12345u;
    for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; C.x[i] = (double)(s >> 8) / (1 << 24) - 0.5; }
    for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; C.b[i] = (double)(s >> 8) / (1 << 24) - 0.5; }
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        double t = 0.0, tj = C.b[i], d = 0.0;
        for (int k = C.H.ia[i]; k < C.H.ia[i + 1]; ++k) {
            const int j = C.H.ja[k];
            t += C.H.val[k] * C.x[j];
            if (j != i) tj -= C.H.val[k] * C.x[j]; else d = C.H.val[k];
        }
        C.diag[i] = d;
        C.yref_mxv[i] = t;
        C.yref_jac[i] = (std::fabs(d) > 1e-20) ? (1 - omega) * C.x[i] + omega * tj / d : C.x[i];
    }
    (void)hipMalloc(&C.dx, 8 * (size_t)n); (void)hipMalloc(&C.dy, 8 * (size_t)n);
    (void)hipMalloc(&C.db, 8 * (size_t)n); (void)hipMalloc(&C.ddiag, 8 * (size_t)n);
    (void)hipMemcpy(C.dx, C.x.data(), 8 * (size_t)n, hipMemcpyHostToDevice);
    (void)hipMemcpy(C.db, C.b.data(), 8 * (size_t)n, hipMemcpyHostToDevice);
    (void)hipMemcpy(C.ddiag, C.diag.data(), 8 * (size_t)n, hipMemcpyHostToDevice);
    C.bytes = 12.0 * C.H.nnz + 4.0 * (n + 1) + 8.0 * C.H.col + 8.0 * n;
}

static void check(Case& C, const std::vector<double>& ref, const char* tag)
{
    const int n = C.H.row;
    std::vector<double> y(n);
    (void)hipStreamSynchronize(g_ctx.stream);
    (void)hipMemcpy(y.data(), C.dy, 8 * (size_t)n, hipMemcpyDeviceToHost);
    long long bad = 0; double md = 0; int first = -1;
    for (int i = 0; i < n; ++i) {
        if (std::memcmp(&y[i], &ref[i], 8) != 0) { ++bad; md = std::max(md, std::fabs(y[i] - ref[i])); if (first < 0) first = i; }
    }
    std::printf("    check %-28s: %lld rows differ bitwise, max |diff| %.3e (first row %d)\n", tag, bad, md, first);
}

static int g_nt = 1;
static int* g_xrows = nullptr;
static int g_nxrows = 0;
static std::string g_filter;
template <class K>
static void run_variant(Case& C, const char* name, K kernel, int op, int rows_per_tile, int xcd, int reps, bool coded,
                        int grid_override = 0)
{
    if (!g_filter.empty() && std::string(name).find(g_filter) == std::string::npos) return;
    CsrArgs a{};
    const DevCSR& M = C.D;
    a.nrow = M.row; a.ia = M.ia; a.ja = M.ja; a.val = M.val; a.dpos = M.dpos; a.ncol = M.col;
    a.x = C.dx; a.y = C.dy; a.b = C.db; a.diag = C.ddiag; a.omega = 0.6667; a.dotv = C.dx; a.partials = g_ctx.d_partials;
    a.nt = g_nt; a.xcd_map = xcd;
    a.pat = M.pat; a.pstart = M.pstart; a.plen = M.plen; a.poff = M.poff; a.pval = M.pval; a.npat = M.npat; a.npent = M.npent;
    a.rowbase = M.rowbase;

    a.ntiles = (M.row + rows_per_tile - 1) / rows_per_tile;
    a.tiles_per_xcd = (a.ntiles + 7) / 8;
    const int save = g_tune.maxgrid;
    if (grid_override) g_tune.maxgrid = grid_override;
    int grid = 0;
    (void)hipMemsetAsync(C.dy, 0, 8 * (size_t)M.row, g_ctx.stream);
    const double us = time_launch([&]() { grid = launch_persistent(kernel, a.ntiles, a); }, reps);
    g_tune.maxgrid = save;
    hipError_t e = hipStreamSynchronize(g_ctx.stream);
    const double extra = (op == OP_JACOBI) ? 8.0 * M.row : 0.0;
    std::printf("  %-36s nt %d xcd %3d grid %4d : %8.1f us  %7.0f GB/s plain-CSR-bytes (%.3f of 8 TB/s)%s\n", name, g_nt, xcd, grid, us,
                (C.bytes + extra) / us * 1e-3, (C.bytes + extra) / us * 1e-3 / 8000.0,
                e == hipSuccess ? "" : "  ### HIP ERROR");
    if (e != hipSuccess) { std::printf("  error: %s\n", hipGetErrorString(e)); std::exit(1); }
    (void)coded;
    // one clean launch for the check (ADD-type ops are not used here)
    (void)hipMemsetAsync(C.dy, 0, 8 * (size_t)M.row, g_ctx.stream);
    launch_persistent(kernel, a.ntiles, a);
    check(C, op == OP_JACOBI ? C.yref_jac : C.yref_mxv, name);
}

static void ceilings(int reps)
{
    const size_t bytes = (size_t)1 << 30;  // 1 GiB per buffer: beyond the 256 MiB Infinity Cache
    f64x2_t *p = nullptr, *q = nullptr, *r = nullptr;
    (void)hipMalloc(&p, bytes); (void)hipMalloc(&q, bytes); (void)hipMalloc(&r, bytes);
    (void)hipMemset(p, 0, bytes); (void)hipMemset(q, 0, bytes); (void)hipMemset(r, 0, bytes);
    const size_t n16 = bytes / 16;
    for (int grid : {1024, 2048, 4096, 8192}) {
        double us = time_launch([&]() { hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(BLOCK), 0, g_ctx.stream, n16, p, q); }, reps);
        std::printf("  copy16  grid %5d: %8.1f us  %7.0f GB/s (read+write)\n", grid, us, 2.0 * bytes / us * 1e-3);
        us = time_launch([&]() { hipLaunchKernelGGL(k_triad16, dim3(grid), dim3(BLOCK), 0, g_ctx.stream, n16, 0.5, p, q, r); }, reps);
        std::printf("  triad16 grid %5d: %8.1f us  %7.0f GB/s\n", grid, us, 3.0 * bytes / us * 1e-3);
        us = time_launch([&]() { hipLaunchKernelGGL(k_read16, dim3(grid), dim3(BLOCK), 0, g_ctx.stream, n16, p, (double*)q); }, reps);
        std::printf("  read16  grid %5d: %8.1f us  %7.0f GB/s\n", grid, us, 1.0 * bytes / us * 1e-3);
    }
    (void)hipFree(p); (void)hipFree(q); (void)hipFree(r);
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? std::atoi(argv[1]) : 128;
    const std::string what = argc > 2 ? argv[2] : "l0";
    const int reps = argc > 3 ? std::atoi(argv[3]) : 20;
    if (ctx_init() != FASP_SUCCESS) return 2;
    if (what == "copy") { ceilings(reps); return 0; }

    dCSRmat A; dvector b, u;
    fasp_hip_poisson7pt(n, n, n, &A, &b, &u);
    Case C;
    host_copy(A, C.H, what == "var");
    std::printf("P7(%d)%s: rows %d nnz %d\n", n, what == "var" ? " scaled D*A*D" : "", C.H.row, C.H.nnz);
    if (upload_csr(C.H, C.D) != FASP_SUCCESS) return 3;
    (void)hipStreamSynchronize(g_ctx.stream);
    {   // dpos for the old Jacobi stream kernel
        std::vector<int> dp(C.H.row, -1);
        for (int i = 0; i < C.H.row; ++i)
            for (int k = C.H.ia[i]; k < C.H.ia[i + 1]; ++k) if (C.H.ja[k] == i) dp[i] = k;
        (void)hipMalloc(&C.D.dpos, 4 * (size_t)C.H.row);
        (void)hipMemcpy(C.D.dpos, dp.data(), 4 * (size_t)C.H.row, hipMemcpyHostToDevice);
    }
    prepare(C, 0.6667);
    std::printf("coded: pat %s (npat %d npent %d)\n", C.D.pat ? "yes" : "no", C.D.npat, C.D.npent);
    if (C.D.pat) {  // exception list of k_csr_rowpat4: row pairs that are not (dom, dom), dom = pattern of row 128 w + 64
        std::vector<unsigned short> hp(C.H.row);
        (void)hipMemcpy(hp.data(), C.D.pat, 2 * (size_t)C.H.row, hipMemcpyDeviceToHost);
        std::vector<int> xr;
        const int nr = C.H.row;
        for (int w0 = 0; w0 < nr; w0 += 128) {
            const unsigned dom = hp[2 * std::min((w0 + 64) / 2, (nr + 1) / 2 - 1)];  // the kernel's lane 32, clamped like its pair load
            for (int r = w0; r < std::min(w0 + 128, nr); r += 2) {
                const bool vb = r + 1 < nr;
                const bool mine = vb && hp[r] == dom && hp[r + 1] == dom;
                if (!mine) { xr.push_back(r); if (vb) xr.push_back(r + 1); }
            }
        }
        g_nxrows = (int)xr.size();
        (void)hipMalloc(&g_xrows, 4 * std::max<size_t>(xr.size(), 1));
        (void)hipMemcpy(g_xrows, xr.data(), 4 * xr.size(), hipMemcpyHostToDevice);
        std::printf("rowpat4 exception list: %d rows (%.2f %%)\n", g_nxrows, 100.0 * g_nxrows / nr);
    }

    std::vector<int> xcds = {16, 64, -1};
    if (argc > 4) xcds = {std::atoi(argv[4])};
    if (argc > 5) g_filter = argv[5];
    std::vector<int> nts = {1, 3, 7};
    if (argc > 6) nts = {std::atoi(argv[6])};
    const int rounds = argc > 7 ? std::atoi(argv[7]) : 2;
    for (int rep = 0; rep < rounds; ++rep) {
        std::printf("--- round %d: plain CSR kernels\n", rep);
        g_nt = 1;
        run_variant(C, "wstream<MXV,64,512> (r1)", k_csr_wstream<OP_MXV, 64, 512>, OP_MXV, 256, 16, reps, false);
        run_variant(C, "wstream<MXV_DOT,64,512> (r1)", k_csr_wstream<OP_MXV_DOT, 64, 512>, OP_MXV_DOT, 256, 16, reps, false);
        run_variant(C, "wstream<JACOBI,64,512> (r1)", k_csr_wstream<OP_JACOBI, 64, 512>, OP_JACOBI, 256, 16, reps, false);
        for (int nt : nts) for (int xcd : xcds) for (int g : {1024, 1280}) {
            g_nt = nt;
            run_variant(C, "lstream<MXV,512>", k_csr_lstream<OP_MXV, 512>, OP_MXV, 256, xcd, reps, false, g);
            if (nt == 1) run_variant(C, "lstream<MXV_DOT,512>", k_csr_lstream<OP_MXV_DOT, 512>, OP_MXV_DOT, 256, xcd, reps, false, g);
            if (nt == 1) run_variant(C, "lstream<JACOBI,512>", k_csr_lstream<OP_JACOBI, 512>, OP_JACOBI, 256, xcd, reps, false, g);
        }
        if (C.D.pat) {
            std::printf("--- round %d: row-pattern kernels\n", rep);
            const bool small = C.D.npat <= 64 && C.D.npent <= 512;
            for (int nt : nts) for (int xcd : xcds) {
                g_nt = nt;
                if (small) {
                    if (nt == 1) {
                    run_variant(C, "rowpat<MXV,2,1> (r1)", k_csr_rowpat<OP_MXV, 2, 1>, OP_MXV, 256, xcd, reps, true);
                    run_variant(C, "rowpat<MXV_DOT,2,1> (r1)", k_csr_rowpat<OP_MXV_DOT, 2, 1>, OP_MXV_DOT, 256, xcd, reps, true);
                    run_variant(C, "rowpat<JACOBI,2,1> (r1)", k_csr_rowpat<OP_JACOBI, 2, 1>, OP_JACOBI, 256, xcd, reps, true);
                    }
                    for (int g : {1024, 1280, 1536, 1792, 2048}) run_variant(C, "rowpat4g<MXV>", k_csr_rowpat4<OP_MXV>, OP_MXV, 512, xcd, reps, true, g);
                    for (int g : {1024, 1280, 1536, 1792, 2048}) run_variant(C, "rowpat4g<JACOBI>", k_csr_rowpat4<OP_JACOBI>, OP_JACOBI, 512, xcd, reps, true, g);
                    run_variant(C, "rowpat4<MXV>", k_csr_rowpat4<OP_MXV>, OP_MXV, 512, xcd, reps, true);
                    run_variant(C, "rowpat4<MXV_DOT>", k_csr_rowpat4<OP_MXV_DOT>, OP_MXV_DOT, 512, xcd, reps, true);
                    run_variant(C, "rowpat4<JACOBI>", k_csr_rowpat4<OP_JACOBI>, OP_JACOBI, 512, xcd, reps, true);
                } else {
                    run_variant(C, "rowpat<MXV,1,1> (r1)", k_csr_rowpat<OP_MXV, 1, 1>, OP_MXV, 256, xcd, reps, true);
                }
            }
        }
    }
    if (argc > 5) return 0;
    ceilings(5);
    return 0;
}
