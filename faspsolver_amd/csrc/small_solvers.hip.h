// small_solvers.hip.h -- coarsest-level solvers as ONE workgroup, ONE launch.
//
// The coarsest level of an aggregation hierarchy is tiny (config 5: 89 rows / 5.6 k nonzeros,
// config 3: 185 block rows), and the reference solves it iteratively to 1e-10 / 1e-6: hundreds
// of Krylov iterations per cycle, thousands per solve.  Driven from the host, every iteration
// costs launches plus one synchronisation (30-100 us) for a few microseconds of arithmetic.
// Here the whole solver -- control flow included -- runs inside one 512-thread workgroup: every
// thread evaluates the scalar recurrences redundantly from block reductions that all threads
// sum in the same fixed order, so branches are uniform without any flag traffic; vectors live
// in global memory (L1/L2 resident at these sizes) and are made visible between phases by
// the workgroup barrier.
//
//   k_spcg_small   fasp_solver_dcsr_spcg   KrySPcg.c:60   (unpreconditioned safe CG, CSR)
//   k_gmres_small  fasp_solver_d{csr,bsr}_pvgmres  KryPvgmres.c:66/:416  (no preconditioner,
//                  STOP_REL_RES, variable restart)
//
// Both restate the same branches as the host-driven versions in solver.hip (coarse_spcg,
// gmres_device); those remain for coarsest levels too large for one CU (e.g. P7(256): 4 971
// rows, 6.4 M nonzeros).
#pragma once

#include <hip/hip_runtime.h>

namespace fasp {

constexpr int SMALL_BLOCK = 512;
constexpr int SMALL_WAVES = SMALL_BLOCK / 64;
constexpr int SMALL_MAX_RESTART = 32;
constexpr int GM_NE = 16;   // k_gmres_small: systems of at most 64 * GM_NE rows orthogonalise in one wavefront

// result record written by thread 0
struct SmallOut {
    int    status;  // iterations (>= 0) or ERROR_* (< 0)
    int    iters;   // iterations performed (also when status < 0)
    double relres, absres;
};

// Sum over the 64 lanes of a wavefront by data-parallel-primitive moves (no LDS crossbar round trips): inclusive
// row_shr 1, 2, 4, 8 inside the rows of 16, then row_bcast 15 and 31 across them; the total ends in lane 63.  Fixed order.
__device__ __forceinline__ double dpp_mov_f64(double x, const int ctrl, const int row_mask)
{
    const long long b = __double_as_longlong(x);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    switch (ctrl) {   // the control word must be a compile-time constant
        case 0x111: lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, true); break;
        case 0x112: lo = __builtin_amdgcn_update_dpp(0, lo, 0x112, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x112, 0xf, 0xf, true); break;
        case 0x114: lo = __builtin_amdgcn_update_dpp(0, lo, 0x114, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x114, 0xf, 0xf, true); break;
        case 0x118: lo = __builtin_amdgcn_update_dpp(0, lo, 0x118, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x118, 0xf, 0xf, true); break;
        case 0x142: lo = __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xa, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xa, 0xf, true); break;
        default:    lo = __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xc, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x143, 0xc, 0xf, true); break;
    }
    (void)row_mask;
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double wave_sum_to_lane63(double x)
{
    x += dpp_mov_f64(x, 0x111, 0xf);
    x += dpp_mov_f64(x, 0x112, 0xf);
    x += dpp_mov_f64(x, 0x114, 0xf);
    x += dpp_mov_f64(x, 0x118, 0xf);
    x += dpp_mov_f64(x, 0x142, 0xa);   // lanes of rows 1 and 3 take lane 15 / 47 (rows not in the mask add 0)
    x += dpp_mov_f64(x, 0x143, 0xc);   // lanes of rows 2 and 3 take lane 31
    return x;
}

__device__ __forceinline__ double wave_max_to_lane63(double x)
{
    x = fmax(x, dpp_mov_f64(x, 0x111, 0xf));   // (rows shifted in are 0: the quantities reduced this way are >= 0)
    x = fmax(x, dpp_mov_f64(x, 0x112, 0xf));
    x = fmax(x, dpp_mov_f64(x, 0x114, 0xf));
    x = fmax(x, dpp_mov_f64(x, 0x118, 0xf));
    x = fmax(x, dpp_mov_f64(x, 0x142, 0xa));
    x = fmax(x, dpp_mov_f64(x, 0x143, 0xc));
    return x;
}
// Sum (or max, per bit of maxmask) of NQ per-thread values over the workgroup.  Every thread
// ends with the same result, summed in the same order: the wave's fixed DPP pattern, then the wave
// partials left to right.  (max: for quantities >= 0 only -- lanes shifted in contribute 0.)
template <int NQ, int NW = SMALL_WAVES>
__device__ __forceinline__ void blk_reduce(double (&v)[NQ], double* sh, unsigned maxmask = 0u)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        // wave stage on data-parallel-primitive moves: a dozen dependent ds_bpermute round trips per quantity (the
        // shuffle tree this replaces) were most of an iteration of the single-workgroup solvers -- config 5's
        // 124 000 coarse CG iterations ran at 4.8 us each
        const bool mx = (maxmask >> q) & 1u;
        const double x = mx ? wave_max_to_lane63(v[q]) : wave_sum_to_lane63(v[q]);
        if (lane == 63) sh[w * NQ + q] = x;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const bool mx = (maxmask >> q) & 1u;
        double x = sh[q];
        for (int k = 1; k < NW; ++k) {
            const double y = sh[k * NQ + q];
            x = mx ? fmax(x, y) : x + y;
        }
        v[q] = x;
    }
    __syncthreads();
}

// blk_reduce with ONE barrier: the wave partials go to one of two alternating buffers (sh2: 2 * NW * NQ doubles, par
// toggles per call).  A wave that runs ahead writes the other buffer; it cannot get two reductions ahead, because the
// barrier of the next one needs every wave to have read this one's.  (The single-workgroup Krylov solvers are chains of
// such reductions: the modified Gram-Schmidt of the coarse GMRES does a dozen per iteration.)
template <int NQ, int NW = SMALL_WAVES>
__device__ __forceinline__ void blk_reduce1(double (&v)[NQ], double* sh2, int& par, unsigned maxmask = 0u)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double* sh = sh2 + par * (NW * NQ);
    par ^= 1;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const bool mx = (maxmask >> q) & 1u;
        const double x = mx ? wave_max_to_lane63(v[q]) : wave_sum_to_lane63(v[q]);
        if (lane == 63) sh[w * NQ + q] = x;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const bool mx = (maxmask >> q) & 1u;
        double x = sh[q];
        for (int k = 1; k < NW; ++k) {
            const double y = sh[k * NQ + q];
            x = mx ? fmax(x, y) : x + y;
        }
        v[q] = x;
    }
}

template <int NT = SMALL_BLOCK>
__device__ __forceinline__ double blk_dot(int n, const double* x, const double* y, double* sh)
{
    double v[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += NT) v[0] += x[i] * y[i];
    blk_reduce<1, NT / 64>(v, sh);
    return v[0];
}

// ---- operators ---------------------------------------------------------------------------
// CSR: 16 lanes per row, strided partial sums, xor butterfly (every lane of the group ends with
// the same sum).  f(row, rowsum) runs on the group's first lane.
struct SmallCSR {
    int           m;
    const int*    ia;
    const int*    ja;
    const double* val;
    __device__ __forceinline__ int rows() const { return m; }
    template <int NT = SMALL_BLOCK, class F>
    __device__ __forceinline__ void for_rows(const double* x, F&& f) const
    {
        const int sl = threadIdx.x & 15;
        for (int row = threadIdx.x >> 4; row < m; row += NT / 16) {
            double s = 0.0;
            int k = ia[row] + sl;
            const int ke = ia[row + 1];
            for (; k + 48 < ke; k += 64) {   // four entries per round trip (LDS or L2), added in the old order
                const int    c0 = ja[k], c1 = ja[k + 16], c2 = ja[k + 32], c3 = ja[k + 48];
                const double v0 = val[k], v1 = val[k + 16], v2 = val[k + 32], v3 = val[k + 48];
                const double x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
                s += v0 * x0; s += v1 * x1; s += v2 * x2; s += v3 * x3;
            }
            for (; k < ke; k += 16) s += val[k] * x[ja[k]];
            s += __shfl_xor(s, 8);
            s += __shfl_xor(s, 4);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 1);
            if (sl == 0) f(row, s, 0.0);
        }
    }
};

// BSR: one thread per scalar row (block row br, component r); every block contributes
// (A_r0 x_0 + A_r1 x_1 + ...), inner sum first, blocks in storage order -- the arithmetic of
// k_bsr_wstream / fasp_blas_smat_ypAx.  `start` seeds the accumulator (0, or -b for residuals).
struct SmallBSR {
    int           ROW, nb;
    const int*    ia;
    const int*    ja;
    const double* val;
    __device__ __forceinline__ int rows() const { return ROW * nb; }
    template <class F>
    __device__ __forceinline__ void for_rows(const double* x, F&& f, const double* seed = nullptr) const
    {
        const int nb2 = nb * nb;
        for (int row = threadIdx.x; row < ROW * nb; row += SMALL_BLOCK) {
            const int br = row / nb, r = row - br * nb;
            double acc = seed ? -seed[row] : 0.0;
            int k = ia[br];
            const int ke = ia[br + 1];
            if (nb == 3) {  // four blocks' loads in flight (the common block size; same arithmetic order)
                for (; k + 3 < ke; k += 4) {
                    int    j[4];
                    double av[4][3], xv[4][3];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        j[q] = ja[k + q];
                        const double* A = val + (size_t)(k + q) * 9 + r * 3;
                        av[q][0] = A[0]; av[q][1] = A[1]; av[q][2] = A[2];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double* xb = x + (size_t)j[q] * 3;
                        xv[q][0] = xb[0]; xv[q][1] = xb[1]; xv[q][2] = xb[2];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        double s = av[q][0] * xv[q][0];
                        s = s + av[q][1] * xv[q][1];
                        s = s + av[q][2] * xv[q][2];
                        acc += s;
                    }
                }
            }
            for (; k < ke; ++k) {
                const double* A = val + (size_t)k * nb2 + r * nb;
                const double* xb = x + (size_t)ja[k] * nb;
                double s = A[0] * xb[0];
                for (int c = 1; c < nb; ++c) s = s + A[c] * xb[c];
                acc += s;
            }
            f(row, acc, 1.0);
        }
    }
};

// y = A x
template <class OP>
__device__ __forceinline__ void small_mxv(const OP& A, const double* x, double* y)
{
    A.for_rows(x, [&](int row, double s, double) { y[row] = s; });
    __syncthreads();
}
// r = b - A x with the reference's rounding (CSR: b_i - t_i, BlaSpmvCSR.c:557; BSR: -((-b_i) + t), :548)
template <int NT = SMALL_BLOCK>
__device__ __forceinline__ void small_resid(const SmallCSR& A, const double* x, const double* b, double* r)
{
    A.template for_rows<NT>(x, [&](int row, double s, double) { r[row] = b[row] - s; });
    __syncthreads();
}
__device__ __forceinline__ void small_resid(const SmallBSR& A, const double* x, const double* b, double* r)
{
    A.for_rows(x, [&](int row, double acc, double) { r[row] = -acc; }, b);
    __syncthreads();
}

// ---------------------------------------------------------------------------
// Safe-net CG, no preconditioner (KrySPcg.c:60-365; z = r throughout)
// ---------------------------------------------------------------------------
struct SpcgArgs {
    SmallCSR      A;
    const double* b;
    double *u, *p, *r, *t, *u_best;
    double   tol;
    int      MaxIt;
    int      x_zero;  // the iterate is zero on entry (skips the first matrix pass)
    int      nnz;
    SmallOut* out;
    const double* img;   // k_spcg_reg: the matrix as its threads hold it, [MC][256] (spcg_reg_image)
    int*     lazy;    // != nullptr: the verdict is not read per solve -- [0] takes the minimum status, [1] the sum of the iteration counts (precond_amg)
};
__device__ __forceinline__ void spcg_report(const SpcgArgs& a, int iter, int MaxIt, double relres, double absres)
{
    const int iters = iter > 0 ? (iter > MaxIt ? MaxIt : iter) : 0;
    const int status = iter > MaxIt ? -48 : iter;  // ERROR_SOLVER_MAXIT
    if (a.lazy) { atomicMin(a.lazy, status); atomicAdd(a.lazy + 1, iters); return; }
    a.out->iters = iters;
    a.out->status = status;
    a.out->relres = relres;
    a.out->absres = absres;
}

// LV: the five work vectors live in dynamic LDS (5 m doubles); LM: so does the matrix (copied once):
// every iteration then runs out of LDS, global memory is touched at entry (b, u) and exit (u) only.
// NT: 512 threads, or ONE wavefront for coarsest levels of a few dozen rows (config 5 of BASELINE.json: 89 rows, 124 000
// coarse CG iterations per solve): no second wave to wait for at any barrier, lane = row products.
template <bool LV, bool LM, int NT = SMALL_BLOCK>
__global__ __launch_bounds__(NT) void k_spcg_small(SpcgArgs a)
{
    __shared__ double sh[(NT / 64) * 5];
    extern __shared__ double dyn[];
    SmallCSR A = a.A;
    const int m = A.m, tid = threadIdx.x;
    if (LM) {  // layout: [5 m doubles | val (nnz doubles) | ja (nnz ints) | ia (m + 1 ints)]
        const int nnz = a.nnz;
        double* lval = dyn + 5 * (size_t)m;
        int*    lja  = reinterpret_cast<int*>(lval + nnz);
        int*    lia  = lja + nnz;
        for (int i = tid; i < nnz; i += NT) { lval[i] = a.A.val[i]; lja[i] = a.A.ja[i]; }
        for (int i = tid; i <= m; i += NT) lia[i] = a.A.ia[i];
        A.val = lval; A.ja = lja; A.ia = lia;
    }
    const double tol = a.tol, maxdiff = tol * 1e-4 /* STAG_RATIO */, sol_inf_tol = 1e-20;
    const double BIG = 1e+20, SMALL = 1e-20, SMALL2 = 1e-40;
    const int MaxIt = a.MaxIt, MAX_STAG = 20, MAX_RESTART = 20;
    int iter = 0, stag = 1, more_step = 1, iter_best = 0;
    double absres0 = BIG, absres = BIG, relres = BIG, normu, normr0 = BIG;
    double reldiff, alpha = 0.0, beta, temp1, temp2, absres_best = BIG;
    double *u = LV ? dyn : a.u, *p = LV ? dyn + m : a.p, *r = LV ? dyn + 2 * (size_t)m : a.r,
           *t = LV ? dyn + 3 * (size_t)m : a.t, *u_best = LV ? dyn + 4 * (size_t)m : a.u_best;
    const double* b = a.b;
    double red[5];

    for (int i = tid; i < m; i += NT) u_best[i] = 0.0;
    if (a.x_zero) {
        for (int i = tid; i < m; i += NT) { r[i] = b[i]; u[i] = 0.0; }
        __syncthreads();
    } else {
        if (LV) for (int i = tid; i < m; i += NT) u[i] = a.u[i];
        __syncthreads();
        small_resid<NT>(A, u, b, r);
    }
    temp1 = blk_dot<NT>(m, r, r, sh);
    absres0 = sqrt(temp1);
    normr0 = fmax(SMALL, absres0);
    relres = absres0 / normr0;
    if (relres < tol) goto FINISHED;
    for (int i = tid; i < m; i += NT) p[i] = r[i];
    __syncthreads();

    while (iter++ < MaxIt) {
        // t = A p and (t, p)
        {
            double v[1] = {0.0};
            A.template for_rows<NT>(p, [&](int row, double s, double) { t[row] = s; v[0] += s * p[row]; });
            blk_reduce<1, NT / 64>(v, sh);
            temp2 = v[0];
        }
        if (fabs(temp2) > SMALL2) alpha = temp1 / temp2;
        else goto RESTORE_BESTSOL;
        // u += alpha p; r -= alpha t; ||r||^2, ||u||^2, ||p||^2, max|u|, #NaN(u)
        red[0] = red[1] = red[2] = red[3] = red[4] = 0.0;
        for (int i = tid; i < m; i += NT) {
            const double pi = p[i];
            const double ui = u[i] + alpha * pi;
            const double ri = r[i] - alpha * t[i];
            u[i] = ui; r[i] = ri;
            red[0] += ri * ri; red[1] += ui * ui; red[2] += pi * pi;
            red[3] = fmax(red[3], fabs(ui));
            red[4] += (ui != ui) ? 1.0 : 0.0;
        }
        blk_reduce<5, NT / 64>(red, sh, 1u << 3);
        absres = sqrt(red[0]);
        relres = absres / normr0;
        if (red[4] > 0.0) {  // fasp_dvec_isnan(u), :185
            absres = BIG;
            goto RESTORE_BESTSOL;
        }
        if (absres < absres_best - maxdiff) {
            absres_best = absres;
            iter_best = iter;
            for (int i = tid; i < m; i += NT) u_best[i] = u[i];
        }
        if (red[3] <= sol_inf_tol) {  // Check I
            iter = -43;               // ERROR_SOLVER_SOLSTAG
            break;
        }
        normu = sqrt(red[1]);
        reldiff = fabs(alpha) * sqrt(red[2]) / normu;
        if ((stag <= MAX_STAG) & (reldiff < maxdiff)) {  // Check II
            __syncthreads();
            small_resid<NT>(A, u, b, r);
            red[0] = blk_dot<NT>(m, r, r, sh);
            absres = sqrt(red[0]);
            relres = absres / normr0;
            if (relres < tol) break;
            if (stag >= MAX_STAG) { iter = -42; break; }  // ERROR_SOLVER_STAG
            for (int i = tid; i < m; i += NT) p[i] = 0.0;
            ++stag;
        }
        if (relres < tol) {  // Check III: true residual
            __syncthreads();
            small_resid<NT>(A, u, b, r);
            red[0] = blk_dot<NT>(m, r, r, sh);
            absres = sqrt(red[0]);
            relres = absres / normr0;
            if (relres < tol) break;
            if (more_step >= MAX_RESTART) { iter = -44; break; }  // ERROR_SOLVER_TOLSMALL
            for (int i = tid; i < m; i += NT) p[i] = 0.0;
            ++more_step;
        }
        absres0 = absres;
        temp2 = red[0];  // (z, r) with z = r
        beta = temp2 / temp1;
        temp1 = temp2;
        for (int i = tid; i < m; i += NT) p[i] = 1.0 * r[i] + beta * p[i];  // fasp_blas_darray_axpby
        __syncthreads();
    }

RESTORE_BESTSOL:
    __syncthreads();
    if (iter != iter_best) {
        small_resid<NT>(A, u_best, b, r);
        absres_best = sqrt(blk_dot<NT>(m, r, r, sh));
        if (absres > absres_best + maxdiff || absres != absres) {
            for (int i = tid; i < m; i += NT) u[i] = u_best[i];
            relres = absres_best / normr0;
        }
    }
FINISHED:
    if (LV) {
        __syncthreads();
        for (int i = tid; i < m; i += NT) a.u[i] = u[i];
    }
    if (tid == 0) {
        spcg_report(a, iter, MaxIt, relres, absres);
    }
    (void)absres0;
}

// ---------------------------------------------------------------------------
// k_spcg_wave: the same safe CG (KrySPcg.c:60-365) for coarsest levels of at most 128 rows, in ONE wavefront.
// The 512-thread kernel spends 3.7 us per iteration on a level of 89 rows (config 5 of BASELINE.json: a W-cycle over
// five levels = 124 000 coarse iterations per solve, 84 % of it): every phase of an iteration is a chain of LDS round
// trips and barriers that eight waves wait through together.  Here a lane owns rows i and i + 64; the iterate, the
// residual, the direction and the product live in its registers, the matrix sits DENSE in LDS (row stride = 1 mod 32
// doubles: conflict-free for lane = row reads), p is broadcast from LDS; the row sums of t = A p are 2 m independent
// LDS reads per lane with no index to wait for, the reductions are data-parallel-primitive moves + readlane.  Row sums
// are four interleaved partial sums over ascending columns (the reference: storage order): O(1e-16) apart, like every
// reduction here.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_bcast63(double x)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_allsum(double x) { return wave_bcast63(wave_sum_to_lane63(x)); }
__device__ __forceinline__ double wave_allmax(double x) { return wave_bcast63(wave_max_to_lane63(x)); }

__global__ __launch_bounds__(64) void k_spcg_wave(SpcgArgs a, int LD)
{
    extern __shared__ double dyn[];   // [128 * LD] dense matrix, [128] p
    const SmallCSR A = a.A;
    const int m = A.m, lane = threadIdx.x;
    double* Ad = dyn;
    double* pb = dyn + 128 * (size_t)LD;
    const int r0 = lane, r1 = lane + 64;
    const bool h0 = r0 < m, h1 = r1 < m;
    // dense copy (rows beyond m and absent entries are zeros; a column stored twice is added up)
    for (int i = lane; i < 128 * LD; i += 64) Ad[i] = 0.0;
    __syncthreads();
    for (int e = 0; e < 2; ++e) {
        const int row = lane + 64 * e;
        if (row < m)
            for (int k = A.ia[row]; k < A.ia[row + 1]; ++k) Ad[row * LD + A.ja[k]] += A.val[k];
    }
    __syncthreads();
    const double* A0 = Ad + (size_t)r0 * LD;
    const double* A1 = Ad + (size_t)r1 * LD;
    auto mxv = [&](double x0, double x1, double& y0, double& y1) {   // y = A x, x through the broadcast buffer
        __syncthreads();
        pb[r0] = x0; pb[r1] = x1;
        __syncthreads();
        // four partial sums per row (columns j = c mod 4): the additions of a row are not one chain of m dependent
        // operations -- a lone wavefront has nobody to hide that latency behind -- and eight columns' loads are in flight
        double sa[4] = {0.0, 0.0, 0.0, 0.0}, sb[4] = {0.0, 0.0, 0.0, 0.0};
        int j = 0;
        for (; j + 7 < m; j += 8) {
            double q[8], x0[8], x1[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) { q[c] = pb[j + c]; x0[c] = A0[j + c]; x1[c] = A1[j + c]; }
#pragma unroll
            for (int c = 0; c < 8; ++c) { sa[c & 3] += x0[c] * q[c]; sb[c & 3] += x1[c] * q[c]; }
        }
        for (; j < m; ++j) { const double q = pb[j]; sa[j & 3] += A0[j] * q; sb[j & 3] += A1[j] * q; }
        const double s0 = (sa[0] + sa[1]) + (sa[2] + sa[3]), s1 = (sb[0] + sb[1]) + (sb[2] + sb[3]);
        y0 = s0; y1 = s1;
    };
    const double tol = a.tol, maxdiff = tol * 1e-4 /* STAG_RATIO */, sol_inf_tol = 1e-20;
    const double BIG = 1e+20, SMALL = 1e-20, SMALL2 = 1e-40;
    const int MaxIt = a.MaxIt, MAX_STAG = 20, MAX_RESTART = 20;
    int iter = 0, stag = 1, more_step = 1, iter_best = 0;
    double absres0 = BIG, absres = BIG, relres = BIG, normu, normr0 = BIG;
    double reldiff, alpha = 0.0, beta, temp1, temp2, absres_best = BIG;
    const double b0 = h0 ? a.b[r0] : 0.0, b1 = h1 ? a.b[r1] : 0.0;
    double u0 = 0.0, u1 = 0.0, p0 = 0.0, p1 = 0.0, rr0, rr1, t0 = 0.0, t1 = 0.0, ub0 = 0.0, ub1 = 0.0;
    auto resid = [&](double x0, double x1, double& y0, double& y1) {   // b - A x (rows beyond m: 0 - 0)
        double s0, s1;
        mxv(x0, x1, s0, s1);
        y0 = b0 - s0; y1 = b1 - s1;
    };
    if (a.x_zero) { rr0 = b0; rr1 = b1; }
    else {
        u0 = h0 ? a.u[r0] : 0.0; u1 = h1 ? a.u[r1] : 0.0;
        resid(u0, u1, rr0, rr1);
    }
    temp1 = wave_allsum(rr0 * rr0 + rr1 * rr1);
    absres0 = sqrt(temp1);
    normr0 = fmax(SMALL, absres0);
    relres = absres0 / normr0;
    if (relres < tol) goto FINISHED;
    p0 = rr0; p1 = rr1;

    while (iter++ < MaxIt) {
        mxv(p0, p1, t0, t1);
        temp2 = wave_allsum(t0 * p0 + t1 * p1);
        if (fabs(temp2) > SMALL2) alpha = temp1 / temp2;
        else goto RESTORE_BESTSOL;
        u0 = u0 + alpha * p0; u1 = u1 + alpha * p1;
        rr0 = rr0 - alpha * t0; rr1 = rr1 - alpha * t1;
        const double q_rr = wave_allsum(rr0 * rr0 + rr1 * rr1);
        const double q_uu = wave_allsum(u0 * u0 + u1 * u1);
        const double q_pp = wave_allsum(p0 * p0 + p1 * p1);
        const double q_mx = wave_allmax(fmax(fabs(u0), fabs(u1)));
        const double q_nan = wave_allsum(((u0 != u0) ? 1.0 : 0.0) + ((u1 != u1) ? 1.0 : 0.0));
        absres = sqrt(q_rr);
        relres = absres / normr0;
        double red0 = q_rr;
        if (q_nan > 0.0) {  // fasp_dvec_isnan(u), :185
            absres = BIG;
            goto RESTORE_BESTSOL;
        }
        if (absres < absres_best - maxdiff) {
            absres_best = absres;
            iter_best = iter;
            ub0 = u0; ub1 = u1;
        }
        if (q_mx <= sol_inf_tol) {  // Check I
            iter = -43;             // ERROR_SOLVER_SOLSTAG
            break;
        }
        normu = sqrt(q_uu);
        reldiff = fabs(alpha) * sqrt(q_pp) / normu;
        if ((stag <= MAX_STAG) & (reldiff < maxdiff)) {  // Check II
            resid(u0, u1, rr0, rr1);
            red0 = wave_allsum(rr0 * rr0 + rr1 * rr1);
            absres = sqrt(red0);
            relres = absres / normr0;
            if (relres < tol) break;
            if (stag >= MAX_STAG) { iter = -42; break; }  // ERROR_SOLVER_STAG
            p0 = 0.0; p1 = 0.0;
            ++stag;
        }
        if (relres < tol) {  // Check III: true residual
            resid(u0, u1, rr0, rr1);
            red0 = wave_allsum(rr0 * rr0 + rr1 * rr1);
            absres = sqrt(red0);
            relres = absres / normr0;
            if (relres < tol) break;
            if (more_step >= MAX_RESTART) { iter = -44; break; }  // ERROR_SOLVER_TOLSMALL
            p0 = 0.0; p1 = 0.0;
            ++more_step;
        }
        absres0 = absres;
        temp2 = red0;  // (z, r) with z = r
        beta = temp2 / temp1;
        temp1 = temp2;
        p0 = 1.0 * rr0 + beta * p0; p1 = 1.0 * rr1 + beta * p1;  // fasp_blas_darray_axpby
    }

RESTORE_BESTSOL:
    if (iter != iter_best) {
        double s0, s1;
        resid(ub0, ub1, s0, s1);
        absres_best = sqrt(wave_allsum(s0 * s0 + s1 * s1));
        if (absres > absres_best + maxdiff || absres != absres) {
            u0 = ub0; u1 = ub1;
            relres = absres_best / normr0;
        }
    }
FINISHED:
    if (h0) a.u[r0] = u0;
    if (h1) a.u[r1] = u1;
    if (lane == 0) {
        spcg_report(a, iter, MaxIt, relres, absres);
    }
    (void)absres0;
}

// ---------------------------------------------------------------------------
// k_spcg_reg<MC>: the same safe CG for coarsest levels of at most 128 rows with the MATRIX IN REGISTERS: four
// wavefronts, thread (row, h) = (tid / 2, tid % 2) keeps the entries of its row in the columns 4 (k / 2) + 2 h + k % 2 (k < MC,
// dense, zeros where the level stores nothing) in MC register pairs for the whole solve.  An iteration of k_spcg_wave reads
// 3 m doubles per lane from LDS (its two dense rows and the broadcast of p) -- 3 us at 89 rows, and config 5 of
// BASELINE.json runs 124 000 of them per solve; here the product is MC multiply-adds on registers against p read as
// 16-byte broadcasts (all lanes one address), the two halves of a row meet by one lane swap, and the price is a
// workgroup barrier where the single wavefront had none: two around the broadcast of p, one per reduction.
// Row sums: four interleaved partial sums per half over ascending k, then half 0 + half 1 (the reference: storage
// order) -- O(1e-16) apart, like every reduction here.  Same exit and restart logic as k_spcg_small, line for line.
// ---------------------------------------------------------------------------
constexpr int SPCG_REG_NT = 256;
#ifndef SPCG_REG_G
#define SPCG_REG_G 12   // 16-byte reads of p in flight per thread (k_spcg_reg's product)
#endif
__device__ __forceinline__ double pair_swap(double x)   // the value of the other lane of the pair (2 i, 2 i + 1)
{
    const long long b = __double_as_longlong(x);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_mov_dpp(lo, 0xb1, 0xf, 0xf, true);   // quad_perm [1, 0, 3, 2]
    hi = __builtin_amdgcn_mov_dpp(hi, 0xb1, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
#ifdef SR_TIMING
#define SRT(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); srt[k] += now_ - srt_last; srt_last = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SRT(k)
#endif
template <int MC>
__global__ __launch_bounds__(SPCG_REG_NT) void k_spcg_reg(SpcgArgs a, int LD)
{
#ifdef SR_TIMING
    unsigned long long srt[6] = {0, 0, 0, 0, 0, 0}, srt_last = __builtin_amdgcn_s_memrealtime();
#endif
    extern __shared__ __attribute__((aligned(16))) double dyn[];   // [2 MC + 4] p
    __shared__ double sh2[2 * 4 * 5];
    const SmallCSR A = a.A;
    const int m = A.m, tid = threadIdx.x;
    const int row = tid >> 1, h = tid & 1;
    const bool has = row < m, mine = has && h == 0;   // `mine`: the lane that counts the row in reductions
    int par = 0;
    (void)LD;
    // the thread's share of the matrix: entry k = column 4 (k / 2) + 2 h + k % 2 of its row, dense, from the image the host
    // lays out once per hierarchy (spcg_reg_image: [MC][256]; a column stored twice is added up there in storage order) --
    // building it here took 15-35 us of every one of config 5's 712 coarse solves per solve
    double Ar[MC];
#pragma unroll
    for (int k = 0; k < MC; ++k) Ar[k] = a.img[(size_t)k * SPCG_REG_NT + tid];
    double* pb = dyn;
    for (int i = tid; i < 2 * MC + 4; i += SPCG_REG_NT) pb[i] = 0.0;
    __syncthreads();
    // The scalar tail of an iteration is three square roots and three divisions that every lane would work through one
    // after the other (each a sequence of ~30 dependent instructions, ~0.1 us): lanes 0, 1, 2 take one each, side by
    // side, and the results come back by readlane -- the same operations on the same operands, a third of the time.
    const int lane = tid & 63;
    auto bcast = [&](double x, int from) -> double {
        const unsigned long long b = (unsigned long long)__double_as_longlong(x);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, from);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), from);
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    auto sqrt3 = [&](double x0, double x1, double x2, double& y0, double& y1, double& y2) {
        const double y = sqrt(lane == 0 ? x0 : lane == 1 ? x1 : x2);
        y0 = bcast(y, 0); y1 = bcast(y, 1); y2 = bcast(y, 2);
    };
    auto div3 = [&](double n0, double d0, double n1, double d1, double n2, double d2, double& y0, double& y1, double& y2) {
        const double y = (lane == 0 ? n0 : lane == 1 ? n1 : n2) / (lane == 0 ? d0 : lane == 1 ? d1 : d2);
        y0 = bcast(y, 0); y1 = bcast(y, 1); y2 = bcast(y, 2);
    };
    // (every product is followed by a reduction before the next one: its barrier also says that everybody is done
    // reading the previous broadcast, so the broadcast needs one barrier, not two)
    const f64x2_t* pmine = reinterpret_cast<const f64x2_t*>(pb + 2 * h);
    auto mxv = [&](double x) -> double {   // y_row = (A x)_row, x_row given in every lane of the pair
        if (mine) pb[row] = x;
        __syncthreads();
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        // two columns per 16-byte read (even lanes one address, odd lanes the next: a two-address broadcast).  The reads go out in
        // groups of SPCG_REG_G, the next group before the multiply-adds of the current one: left to itself the scheduler keeps two
        // reads in flight (register pressure is what it minimises) and the product pays the LDS latency MC / 4 times over
        constexpr int NRD = MC / 2, G0 = MC > 48 ? 8 : SPCG_REG_G /* (128 register pairs of matrix: shorter groups) */, G = NRD < G0 ? NRD : G0, NG = (NRD + G - 1) / G;
        f64x2_t qa[G], qb[G];
#pragma unroll
        for (int j = 0; j < G; ++j) qa[j] = pmine[2 * j];     // columns 4 j + 2 h, 4 j + 2 h + 1
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            f64x2_t (&cur)[G] = (g & 1) ? qb : qa;
            f64x2_t (&nxt)[G] = (g & 1) ? qa : qb;
            if (g + 1 < NG) {
#pragma unroll
                for (int j = 0; j < G; ++j)
                    if ((g + 1) * G + j < NRD) nxt[j] = pmine[2 * ((g + 1) * G + j)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int k = 2 * (g * G + j);
                if (k < MC) {
                    s[k & 3] += Ar[k] * cur[j][0];
                    s[(k + 1) & 3] += Ar[k + 1] * cur[j][1];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const double sh_ = (s[0] + s[1]) + (s[2] + s[3]);
        const double so = pair_swap(sh_);
        return h ? so + sh_ : sh_ + so;    // half 0 + half 1 in both lanes
    };
    auto allsum1 = [&](double x) -> double { double v[1] = {mine ? x : 0.0}; blk_reduce1<1, 4>(v, sh2, par); return v[0]; };
    const double tol = a.tol, maxdiff = tol * 1e-4 /* STAG_RATIO */, sol_inf_tol = 1e-20;
    const double BIG = 1e+20, SMALL = 1e-20, SMALL2 = 1e-40;
    const int MaxIt = a.MaxIt, MAX_STAG = 20, MAX_RESTART = 20;
    int iter = 0, stag = 1, more_step = 1, iter_best = 0;
    double absres0 = BIG, absres = BIG, relres = BIG, normu, normr0 = BIG;
    double reldiff, alpha = 0.0, beta, temp1, temp2, absres_best = BIG;
    const double b = has ? a.b[row] : 0.0;
    double u = 0.0, p = 0.0, r, t = 0.0, ub = 0.0;
    if (a.x_zero) r = b;
    else {
        u = has ? a.u[row] : 0.0;
        r = b - mxv(u);
    }
    temp1 = allsum1(r * r);
    absres0 = sqrt(temp1);
    normr0 = fmax(SMALL, absres0);
    relres = absres0 / normr0;
    if (relres < tol) goto FINISHED;
    p = r;

    SRT(0);
    while (iter++ < MaxIt) {
        SRT(5);
        t = mxv(p);
        SRT(1);
        temp2 = allsum1(t * p);
        SRT(2);
        if (fabs(temp2) > SMALL2) alpha = temp1 / temp2;
        else goto RESTORE_BESTSOL;
        u = u + alpha * p;
        r = r - alpha * t;
        // (u has a NaN exactly when (u, u) is one: squares are >= 0, so no sum of them cancels to NaN -- the fifth quantity of
        // fasp_dvec_isnan's count is read off the second)
        double q[5] = {mine ? r * r : 0.0, mine ? u * u : 0.0, mine ? p * p : 0.0, 1.0, 0.0};
        {
            double q3[3] = {q[0], q[1], q[2]};
            blk_reduce1<3, 4>(q3, sh2, par);
            q[0] = q3[0]; q[1] = q3[1]; q[2] = q3[2];
            q[4] = (q3[1] != q3[1]) ? 1.0 : 0.0;
            // Check I asks whether max |u_i| <= 1e-20.  (u, u) > 2 m 1e-40 settles it (some u_i^2 exceeds 1e-40 then, whatever the
            // rounding of the sum); only below that -- an iterate that is zero to forty digits -- is the maximum itself formed.
            if (!(q3[1] > 2.0 * m * 1e-40)) {
                double qm[1] = {mine ? fabs(u) : 0.0};
                blk_reduce1<1, 4>(qm, sh2, par, 1u);
                q[3] = qm[0];
            }
        }
        SRT(3);
        double sq_pp, fac;
        sqrt3(q[0], q[1], q[2], absres, normu, sq_pp);      // absres = sqrt(rr), normu = sqrt(uu), sqrt(pp)
        fac = fabs(alpha) * sq_pp;
        div3(absres, normr0, fac, normu, q[0], temp1, relres, reldiff, beta);   // relres, reldiff, and beta = rr / temp1 for the usual path
        SRT(4);
        double red0 = q[0];
        if (q[4] > 0.0) {  // fasp_dvec_isnan(u), :185
            absres = BIG;
            goto RESTORE_BESTSOL;
        }
        if (absres < absres_best - maxdiff) {
            absres_best = absres;
            iter_best = iter;
            ub = u;
        }
        if (q[3] <= sol_inf_tol) {  // Check I
            iter = -43;             // ERROR_SOLVER_SOLSTAG
            break;
        }
        if ((stag <= MAX_STAG) & (reldiff < maxdiff)) {  // Check II
            r = b - mxv(u);
            red0 = allsum1(r * r);
            absres = sqrt(red0);
            relres = absres / normr0;
            if (relres < tol) break;
            if (stag >= MAX_STAG) { iter = -42; break; }  // ERROR_SOLVER_STAG
            p = 0.0;
            ++stag;
        }
        if (relres < tol) {  // Check III: true residual
            r = b - mxv(u);
            red0 = allsum1(r * r);
            absres = sqrt(red0);
            relres = absres / normr0;
            if (relres < tol) break;
            if (more_step >= MAX_RESTART) { iter = -44; break; }  // ERROR_SOLVER_TOLSMALL
            p = 0.0;
            ++more_step;
        }
        absres0 = absres;
        temp2 = red0;  // (z, r) with z = r
        if (red0 != q[0]) beta = temp2 / temp1;   // (a check recomputed the residual: not the quotient formed above)
        temp1 = temp2;
        p = 1.0 * r + beta * p;  // fasp_blas_darray_axpby
    }

RESTORE_BESTSOL:
    if (iter != iter_best) {
        const double s = b - mxv(ub);
        absres_best = sqrt(allsum1(s * s));
        if (absres > absres_best + maxdiff || absres != absres) {
            u = ub;
            relres = absres_best / normr0;
        }
    }
FINISHED:
#ifdef SR_TIMING
    if (tid == 0 && iter > 50) printf("[spcg_reg] iters %d: setup %.2f us | per iteration: mxv %.3f dot %.3f update+reduce3 %.3f sqrt/div %.3f rest %.3f us\n", iter, srt[0] * 0.01, srt[1] * 0.01 / iter, srt[2] * 0.01 / iter, srt[3] * 0.01 / iter, srt[4] * 0.01 / iter, srt[5] * 0.01 / iter);
#endif
    if (mine) a.u[row] = u;
    if (tid == 0) {
        spcg_report(a, iter, MaxIt, relres, absres);
    }
    (void)absres0;
}

// ---------------------------------------------------------------------------
// k_spcg_dpp<NBK, DUP>: the same safe CG for coarsest levels of at most 16 NBK <= 128 rows, the matrix in registers as in
// k_spcg_reg, with the BROADCAST of the direction inside the multiply-add and the iteration itself in ONE wavefront.
// k_spcg_reg's product reads p from LDS as 16-byte broadcasts: 24 reads per wavefront and product at 89 rows, and a broadcast read
// returns 1 KB to the register file like any other -- four wavefronts' reads queue behind one another on the one LDS pipeline for
// 0.32 of the product's 0.43 us; and each of its three reductions per iteration crosses four wavefronts through LDS and a barrier
// (stamps: profiles/r06_configs_3_5.txt).  gfx950's double-precision unit takes a data-parallel-primitive operand of its own:
// v_fmac_f64_dpp ... row_newbcast:j multiplies by lane j's value of each row of 16 lanes (tools/micro/dppfma.hip: exact, 1.2 x the
// time of a plain multiply-add).  So the matrix is cut into 16 x 16 blocks; a row of 16 lanes ("group") owns one COLUMN block cb and
// multiplies RB = NBK / DUP row blocks of it by its lanes' 16 values of the vector -- no broadcast read at all.  What a product
// leaves are partial row sums per column block, added in ascending column-block order.
// Wavefront 0 carries the vectors (lane i: elements i and i + 64, k_spcg_wave's layout) and runs the reference's iteration with
// reductions that never leave it (data-parallel-primitive moves + readlane); the other wavefronts only multiply: per product the
// vector goes out through LDS (barrier A), everybody multiplies its blocks, the partials come back (barrier B).  Two barriers per
// iteration and no cross-wavefront reduction, against k_spcg_reg's three barriers each with one.  Same exit and restart logic
// as k_spcg_small.
// ---------------------------------------------------------------------------
template <int J>
__device__ __forceinline__ void dpp_fmac(double& acc, double x, double a)
{
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(a), "n"(J));
}
// RB 16 x 16 blocks against the same 16 values of x: acc[k][j & 3] += a[16 k + j] * x(lane j of the row), j ascending, the blocks
// interleaved (RB independent chains per j: no multiply-add waits for the previous one of its sum).  (The wait states a
// data-parallel-primitive read needs behind a vector write of its operand or of EXEC are the compiler's to insert between ordinary
// instructions; behind inline assembly it sees nothing, so they stand in front -- volatile statements keep their order.)
template <int J, int RB>
__device__ __forceinline__ void dpp_col(double (&acc)[RB][4], double x, const double* a)
{
#pragma unroll
    for (int k = 0; k < RB; ++k) dpp_fmac<J>(acc[k][J & 3], x, a[16 * k + J]);
}
template <int RB>
__device__ __forceinline__ void dpp_blocks16(double (&acc)[RB][4], double x, const double* a)
{
    asm volatile("s_nop 4");
    dpp_col<0, RB>(acc, x, a);  dpp_col<1, RB>(acc, x, a);  dpp_col<2, RB>(acc, x, a);  dpp_col<3, RB>(acc, x, a);
    dpp_col<4, RB>(acc, x, a);  dpp_col<5, RB>(acc, x, a);  dpp_col<6, RB>(acc, x, a);  dpp_col<7, RB>(acc, x, a);
    dpp_col<8, RB>(acc, x, a);  dpp_col<9, RB>(acc, x, a);  dpp_col<10, RB>(acc, x, a); dpp_col<11, RB>(acc, x, a);
    dpp_col<12, RB>(acc, x, a); dpp_col<13, RB>(acc, x, a); dpp_col<14, RB>(acc, x, a); dpp_col<15, RB>(acc, x, a);
}
// three sums over the wavefront at once, every lane ends with all three: the steps of the three interleaved (written one after the
// other the compiler runs them one after the other, each step waiting for the previous one of its own sum)
__device__ __forceinline__ void wave_allsum3(double& x, double& y, double& z)
{
#define FASP_SUM3_STEP(ctrl, mask) { const double tx = dpp_mov_f64(x, ctrl, mask), ty = dpp_mov_f64(y, ctrl, mask), tz = dpp_mov_f64(z, ctrl, mask); \
                                     x += tx; y += ty; z += tz; __builtin_amdgcn_sched_barrier(0); }
    __builtin_amdgcn_sched_barrier(0);
    FASP_SUM3_STEP(0x111, 0xf) FASP_SUM3_STEP(0x112, 0xf) FASP_SUM3_STEP(0x114, 0xf) FASP_SUM3_STEP(0x118, 0xf)
    FASP_SUM3_STEP(0x142, 0xa) FASP_SUM3_STEP(0x143, 0xc)
#undef FASP_SUM3_STEP
    x = wave_bcast63(x); y = wave_bcast63(y); z = wave_bcast63(z);
}
__device__ __forceinline__ void wave_allsum2(double& x, double& y)
{
#define FASP_SUM2_STEP(ctrl, mask) { const double tx = dpp_mov_f64(x, ctrl, mask), ty = dpp_mov_f64(y, ctrl, mask); x += tx; y += ty; __builtin_amdgcn_sched_barrier(0); }
    __builtin_amdgcn_sched_barrier(0);
    FASP_SUM2_STEP(0x111, 0xf) FASP_SUM2_STEP(0x112, 0xf) FASP_SUM2_STEP(0x114, 0xf) FASP_SUM2_STEP(0x118, 0xf)
    FASP_SUM2_STEP(0x142, 0xa) FASP_SUM2_STEP(0x143, 0xc)
#undef FASP_SUM2_STEP
    x = wave_bcast63(x); y = wave_bcast63(y);
}
template <int NBK, int DUP, bool AHEAD>
__global__ __launch_bounds__(16 * NBK * DUP + (AHEAD ? 64 : 0)) void k_spcg_dpp(SpcgArgs a, int LD)
{
#ifdef SR_TIMING
    unsigned long long srt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, srt_last = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int RB = NBK / DUP, MP = 16 * NBK;   // row blocks per group, padded order
    static_assert(RB * DUP == NBK && NBK * DUP <= 16 && (NBK * DUP) % 4 == 0 && NBK >= 4, "k_spcg_dpp: whole wavefronts of four groups, the first of them = column blocks 0 .. 3");
    __shared__ double part[NBK * MP];   // part[cb][row]: the row sums over column block cb
    __shared__ double xbuf[MP];         // the vector of the coming product, from wavefront 0
    __shared__ int    cmd;              // 1: a product follows, 0: the solve is over
    const int m = a.A.m, tid = threadIdx.x, lane = tid & 63;
    const int ts = AHEAD ? (tid >= 64 ? tid - 64 : 0) : tid;   // AHEAD: wavefront 0 multiplies nothing, the groups start behind it
    const int g = ts >> 4, l = ts & 15;
    const int cb = g % NBK, half = g / NBK;
    (void)LD;
    double Ar[RB * 16];   // entry 16 k + j: A(16 (half RB + k) + l, 16 cb + j), from the image the host lays out once per hierarchy
#pragma unroll
    for (int k = 0; k < RB * 16; ++k) Ar[k] = a.img[(size_t)k * (16 * NBK * DUP) + ts];
    double* const pw = part + cb * MP + 16 * (half * RB) + l;   // this lane's partials: row blocks half RB .. half RB + RB - 1
    auto blocks = [&](double x) {   // x: element 16 cb + l of the vector
        double acc[RB][4];
#pragma unroll
        for (int k = 0; k < RB; ++k) acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.0;
        dpp_blocks16<RB>(acc, x, Ar);
#pragma unroll
        for (int k = 0; k < RB; ++k) pw[16 * k] = (acc[k][0] + acc[k][1]) + (acc[k][2] + acc[k][3]);
    };
    if (tid >= 64) {   // the wavefronts that only multiply (AHEAD: all that multiply)
        for (;;) {
            __syncthreads();   // (A) the vector is out
            if (cmd == 0) break;
            blocks(xbuf[16 * cb + l]);
            __syncthreads();   // (B) the partials are out
        }
        return;
    }
    // ---- wavefront 0: the iteration.  Lane i carries elements i and i + 64 (k_spcg_wave's layout); as a multiplying wavefront it is
    // groups 0 .. 3 = column blocks 0 .. 3, whose elements are its lanes' first ones.
    // The reference's loop (KrySPcg.c:160-330) asks for a product in five places: the initial residual, the iteration, the true
    // residuals of Checks II and III, the residual of the best iterate.  Here they are ONE site that the control flow comes back to
    // with `st` saying what the product is for: inlined five times the multiply-adds made 40 KB of code with the iteration jumping
    // between its pieces, and a taken branch of this lone wavefront is an instruction fetch that nobody hides.
    const int r0 = lane, r1 = lane + 64;
    const bool h0 = r0 < m, h1 = r1 < m;
    auto bcast = [&](double x, int from) -> double {
        const unsigned long long b = (unsigned long long)__double_as_longlong(x);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, from);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), from);
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    auto sqrt3 = [&](double x0, double x1, double x2, double& y0, double& y1, double& y2) {   // (k_spcg_reg: three lanes side by side)
        const double y = sqrt(lane == 0 ? x0 : lane == 1 ? x1 : x2);
        y0 = bcast(y, 0); y1 = bcast(y, 1); y2 = bcast(y, 2);
    };
    auto div3 = [&](double n0, double d0, double n1, double d1, double n2, double d2, double& y0, double& y1, double& y2) {
        const double y = (lane == 0 ? n0 : lane == 1 ? n1 : n2) / (lane == 0 ? d0 : lane == 1 ? d1 : d2);
        y0 = bcast(y, 0); y1 = bcast(y, 1); y2 = bcast(y, 2);
    };
    enum { ST_INIT, ST_ITER, ST_CHK2, ST_CHK3, ST_BEST };
    const double tol = a.tol, maxdiff = tol * 1e-4 /* STAG_RATIO */, sol_inf_tol = 1e-20;
    const double BIG = 1e+20, SMALL = 1e-20, SMALL2 = 1e-40;
    const int MaxIt = a.MaxIt, MAX_STAG = 20, MAX_RESTART = 20;
    int iter = 0, stag = 1, more_step = 1, iter_best = 0, st = ST_INIT;
    double absres0 = BIG, absres = BIG, relres = BIG, normu = 0.0, normr0 = BIG;
    double reldiff = 0.0, alpha = 0.0, beta = 0.0, temp1 = 0.0, temp2, absres_best = BIG;
    const double b0 = h0 ? a.b[r0] : 0.0, b1 = h1 ? a.b[r1] : 0.0;
    double u0 = 0.0, u1 = 0.0, p0 = 0.0, p1 = 0.0, rr0 = b0, rr1 = b1, ub0 = 0.0, ub1 = 0.0;
    double x0 = 0.0, x1 = 0.0, y0, y1, red0 = 0.0, q_rr = 0.0, q_uu, q_pp, q_mx, sq_pp, fac;
    bool slow, leave = false, hot = false;
    auto send = [&](double v0, double v1) {   // the vector of a product goes out
        xbuf[r0] = v0;
        if (MP > 64) xbuf[r1 < MP ? r1 : r0] = r1 < MP ? v1 : v0;
        if (lane == 0) cmd = 1;
        __syncthreads();   // (A)
        SRT(6);
        if (!AHEAD) blocks(v0);
        SRT(7);
    };
    auto receive = [&](double& w0, double& w1) {   // its row sums come back
        __syncthreads();   // (B)
        SRT(8);
        double q0[NBK], q1[NBK];
#pragma unroll
        for (int c = 0; c < NBK; ++c) { q0[c] = part[c * MP + r0]; q1[c] = part[c * MP + (r1 < MP ? r1 : r0)]; }   // (no second element: read the first again, dropped below)
        w0 = q0[0]; w1 = q1[0];
#pragma unroll
        for (int c = 1; c < NBK; ++c) { w0 += q0[c]; w1 += q1[c]; }   // ascending column blocks
        if (!(r1 < MP)) w1 = 0.0;
    };
    auto mxv = [&](double v0, double v1, double& w0, double& w1) { send(v0, v1); receive(w0, w1); };   // w = A v
    // the start of the iteration once the initial residual is in rr; true: converged at once (KrySPcg.c:140-150)
    auto start = [&]() -> bool {
        temp1 = wave_allsum(rr0 * rr0 + rr1 * rr1);
        absres0 = sqrt(temp1);
        normr0 = fmax(SMALL, absres0);
        relres = absres0 / normr0;
        p0 = rr0; p1 = rr1;
        return relres < tol;
    };
    if (a.x_zero) {
        if (start()) goto FINISHED;
        hot = true;
    } else {
        u0 = h0 ? a.u[r0] : 0.0; u1 = h1 ? a.u[r1] : 0.0;
        x0 = u0; x1 = u1; st = ST_INIT;
    }
    SRT(0);
    // Outer loop: the products outside the iteration proper (initial residual, true residuals of Checks II and III, residual of the
    // best iterate), each followed by the iteration loop below -- a loop of its own with one way round, so that what the compiler
    // makes of an ordinary iteration is one straight piece of code.
    for (;;) {
        if (!hot) {
            mxv(x0, x1, y0, y1);
            if (st == ST_INIT) {
                rr0 = b0 - y0; rr1 = b1 - y1;
                if (start()) break;
            } else if (st == ST_BEST) {
                const double s0 = b0 - y0, s1 = b1 - y1;
                absres_best = sqrt(wave_allsum(s0 * s0 + s1 * s1));
                if (absres > absres_best + maxdiff || absres != absres) {
                    u0 = ub0; u1 = ub1;
                    relres = absres_best / normr0;
                }
                break;
            } else {   // Checks II and III: r = b - A u
                rr0 = b0 - y0; rr1 = b1 - y1;
                red0 = wave_allsum(rr0 * rr0 + rr1 * rr1);
                absres = sqrt(red0);
                relres = absres / normr0;
                leave = relres < tol;
                if (!leave) {
                    if (st == ST_CHK2) {
                        if (stag >= MAX_STAG) { iter = -42; leave = true; }  // ERROR_SOLVER_STAG
                        ++stag;
                        // (Check III after Check II asks `relres < tol` of the residual just recomputed: it was not, a few lines up)
                    } else {
                        if (more_step >= MAX_RESTART) { iter = -44; leave = true; }  // ERROR_SOLVER_TOLSMALL
                        ++more_step;
                    }
                }
                if (!leave) {
                    absres0 = absres;
                    beta = red0 / temp1;   // (a check recomputed the residual: not the quotient the iteration formed)
                    temp1 = red0;          // (z, r) with z = r
                    p0 = 1.0 * rr0 + beta * 0.0; p1 = 1.0 * rr1 + beta * 0.0;  // fasp_blas_darray_axpby on the direction the check zeroed
                }
            }
        }
        hot = false;
        // ---- the iteration (KrySPcg.c:160-330)
        if constexpr (AHEAD) {
            // What the NEXT direction needs of an iteration is (t, p), alpha, the new residual, its square and beta; everything else
            // -- (u, u) and (p, p), the three square roots, relres and reldiff, the NaN test, the best iterate, Checks I-III --
            // decides only whether there IS a next iteration.  So the direction goes out first, and that rest is worked through while
            // the other wavefronts multiply; in the rare iteration in which a test fires the product sent ahead is dropped (it has no
            // side effect) and the iteration count and (z, r) step back to where the reference's loop stands at that test.
            bool pend = false;
            double uu_l = 0.0, pp_l = 0.0, temp1_old = temp1;
            while (!leave) {
                SRT(5);
                const bool more = iter++ < MaxIt;
                if (__builtin_expect(more, 1)) send(p0, p1);   // t = A p under way
                if (pend) {   // the tests of iteration iter - 1
                    pend = false;
                    q_uu = uu_l; q_pp = pp_l;
                    wave_allsum2(q_uu, q_pp);
                    q_mx = 1.0;
                    if (__builtin_expect(!(q_uu > 2.0 * m * 1e-40), 0)) q_mx = wave_allmax(fmax(fabs(u0), fabs(u1)));
                    SRT(3);
                    sqrt3(q_rr, q_uu, q_pp, absres, normu, sq_pp);
                    fac = fabs(alpha) * sq_pp;
                    div3(absres, normr0, fac, normu, 1.0, 1.0, relres, reldiff, y0);
                    SRT(4);
                    red0 = q_rr;
                    const bool isnan = q_uu != q_uu;   // fasp_dvec_isnan(u), :185 (leaves before the best iterate is touched)
                    {
                        const bool better = !isnan & (absres < absres_best - maxdiff);
                        absres_best = better ? absres : absres_best;
                        iter_best = better ? iter - 1 : iter_best;
                        ub0 = better ? u0 : ub0; ub1 = better ? u1 : ub1;
                    }
                    slow = isnan | (q_mx <= sol_inf_tol) | ((stag <= MAX_STAG) & (reldiff < maxdiff)) | (relres < tol);
                    if (__builtin_expect(slow, 0)) {
                        if (more) __syncthreads();   // (B) of the product sent ahead: dropped
                        iter -= 1;
                        temp1 = temp1_old;
                        if (isnan) { absres = BIG; leave = true; break; }
                        if (q_mx <= sol_inf_tol) {  // Check I
                            iter = -43;             // ERROR_SOLVER_SOLSTAG
                            leave = true; break;
                        }
                        x0 = u0; x1 = u1;           // the true residual: Check II if it asks for one, else Check III
                        st = ((stag <= MAX_STAG) & (reldiff < maxdiff)) ? ST_CHK2 : ST_CHK3;
                        break;
                    }
                    absres0 = absres;
                }
                if (__builtin_expect(!more, 0)) { leave = true; break; }
                receive(y0, y1);
                SRT(1);
                temp2 = wave_allsum(y0 * p0 + y1 * p1);
                SRT(2);
                if (__builtin_expect(!(fabs(temp2) > SMALL2), 0)) { leave = true; break; }
                alpha = temp1 / temp2;
                u0 = u0 + alpha * p0; u1 = u1 + alpha * p1;
                rr0 = rr0 - alpha * y0; rr1 = rr1 - alpha * y1;
                uu_l = u0 * u0 + u1 * u1; pp_l = p0 * p0 + p1 * p1;
                SRT(9);
                // (Measured and dropped, config 5 on one box, 187.2 ms as it stands -- three groups of lanes per column block at 65-96 rows (18
                // groups of two blocks, a wavefront more) instead of two: 191.5 against 190.6 on another box; (u, u) and (p, p) as passengers of this sum, three at
                // once: 190.2; u += alpha p moved behind the send with the tests, the partial row sums added as a tree: 189.7.)
                q_rr = wave_allsum(rr0 * rr0 + rr1 * rr1);
                beta = q_rr / temp1;
                temp1_old = temp1;
                temp1 = q_rr;  // (z, r) with z = r
                p0 = 1.0 * rr0 + beta * p0; p1 = 1.0 * rr1 + beta * p1;  // fasp_blas_darray_axpby
                pend = true;
            }
        } else
        while (!leave) {
            SRT(5);
            if (__builtin_expect(!(iter++ < MaxIt), 0)) { leave = true; break; }
            mxv(p0, p1, y0, y1);   // t = A p
            SRT(1);
            temp2 = wave_allsum(y0 * p0 + y1 * p1);
            SRT(2);
            if (__builtin_expect(!(fabs(temp2) > SMALL2), 0)) { leave = true; break; }
            alpha = temp1 / temp2;
            u0 = u0 + alpha * p0; u1 = u1 + alpha * p1;
            rr0 = rr0 - alpha * y0; rr1 = rr1 - alpha * y1;
            SRT(9);
            q_rr = rr0 * rr0 + rr1 * rr1; q_uu = u0 * u0 + u1 * u1; q_pp = p0 * p0 + p1 * p1;
            wave_allsum3(q_rr, q_uu, q_pp);
            // (u has a NaN exactly when (u, u) is one; Check I's maximum is formed only when (u, u) does not settle it: k_spcg_reg)
            q_mx = 1.0;
            if (__builtin_expect(!(q_uu > 2.0 * m * 1e-40), 0)) q_mx = wave_allmax(fmax(fabs(u0), fabs(u1)));
            SRT(3);
            sqrt3(q_rr, q_uu, q_pp, absres, normu, sq_pp);
            fac = fabs(alpha) * sq_pp;
            div3(absres, normr0, fac, normu, q_rr, temp1, relres, reldiff, beta);   // relres, reldiff, and beta = rr / temp1 for the usual path
            SRT(4);
            red0 = q_rr;
            if (__builtin_expect(q_uu != q_uu, 0)) {  // fasp_dvec_isnan(u), :185
                absres = BIG;
                leave = true; break;
            }
            {   // (selects, not a branch: most iterations of a converging solve improve on the best residual)
                const bool better = absres < absres_best - maxdiff;
                absres_best = better ? absres : absres_best;
                iter_best = better ? iter : iter_best;
                ub0 = better ? u0 : ub0; ub1 = better ? u1 : ub1;
            }
            // Checks I, II, III behind ONE test: none of them fires in an ordinary iteration
            slow = (q_mx <= sol_inf_tol) | ((stag <= MAX_STAG) & (reldiff < maxdiff)) | (relres < tol);
            if (__builtin_expect(slow, 0)) {
                if (q_mx <= sol_inf_tol) {  // Check I
                    iter = -43;             // ERROR_SOLVER_SOLSTAG
                    leave = true; break;
                }
                x0 = u0; x1 = u1;           // the true residual: Check II if it asks for one, else Check III
                st = ((stag <= MAX_STAG) & (reldiff < maxdiff)) ? ST_CHK2 : ST_CHK3;
                break;
            }
            absres0 = absres;
            temp1 = red0;  // (z, r) with z = r
            p0 = 1.0 * rr0 + beta * p0; p1 = 1.0 * rr1 + beta * p1;  // fasp_blas_darray_axpby
        }
        if (leave) {   // the residual of the best iterate is one more product, unless the last iterate is the best one
            if (iter == iter_best) break;
            leave = false;
            x0 = ub0; x1 = ub1; st = ST_BEST;
        }
    }
FINISHED:
    if (lane == 0) cmd = 0;
    __syncthreads();   // (A) with nothing behind it: the multiplying wavefronts leave
#ifdef SR_TIMING
    if (tid == 0 && iter > 50) printf("[spcg_dpp] iters %d: setup %.2f us | per iteration: vector out + barrier A %.3f blocks %.3f barrier B %.3f partials in %.3f dot %.3f alpha + update %.3f reduce3 %.3f sqrt/div %.3f rest %.3f loop top %.3f us\n", iter, srt[0] * 0.01, srt[6] * 0.01 / iter, srt[7] * 0.01 / iter, srt[8] * 0.01 / iter, srt[1] * 0.01 / iter, srt[2] * 0.01 / iter, srt[9] * 0.01 / iter, srt[3] * 0.01 / iter, srt[4] * 0.01 / iter, srt[5] * 0.01 / iter, 0.0);
#endif
    if (h0) a.u[r0] = u0;
    if (h1) a.u[r1] = u1;
    if (lane == 0) {
        spcg_report(a, iter, MaxIt, relres, absres);
    }
    (void)absres0;
}

// ---------------------------------------------------------------------------
// Safe CG on a coarsest level too large for k_spcg_small (P7(256): 4 971 rows, 6.4 M nonzeros):
// the SpMV stays a full-chip kernel, everything between two SpMVs is ONE one-block launch that
// keeps the iteration state on the device -- (t,p) from the SpMV's per-block partials, alpha,
// u += alpha p, r -= alpha t, the five norms, the best-iterate copy, the reference's exit and
// restart TESTS, beta and p = r + beta p (KrySPcg.c:160-330).  The host queues a batch of
// iterations without waiting; when a test fires the step kernel raises `stop`, the queued
// launches behind it return at once, and the host runs that branch of the reference (true
// residual, restart, ...) from the recorded scalars and resumes.  One synchronisation per batch
// instead of one per iteration.
// ---------------------------------------------------------------------------
enum SpcgStop : int { SPCG_RUN = 0, SPCG_CONV = 1, SPCG_STAG = 2, SPCG_NAN = 3, SPCG_DIV0 = 4, SPCG_SOLSTAG = 5, SPCG_MAXIT = 6, SPCG_ZERO_RHS = 7 };
struct SpcgState {
    double temp1, temp1_prev, absres_best, normr0, tol, maxdiff;  // inputs carried from step to step
    double tp, rr, uu, pp, maxu, nan, alpha, absres, relres;  // scalars of the latest step
    int    iter, iter_best, stag, MaxIt, stop, pad;
};
struct SpcgStepArgs {
    int           m;
    SpcgState*    st;
    const double* tp_partials;  // per-block partials of (t,p) from the fused SpMV
    int           ntp;
    const double* t;
    double *p, *u, *r, *u_best;
};

__global__ __launch_bounds__(SMALL_BLOCK) void k_spcg_step(SpcgStepArgs a)
{
    __shared__ double sh[SMALL_WAVES * 5];
    SpcgState& S = *a.st;
    if (S.stop != SPCG_RUN) return;
    const int tid = threadIdx.x, m = a.m;
    const double temp1 = S.temp1, absres_best = S.absres_best, normr0 = S.normr0, tol = S.tol, maxdiff = S.maxdiff;
    const int it = S.iter + 1, stag = S.stag, MaxIt = S.MaxIt;
    __syncthreads();  // every thread has read the state before thread 0 rewrites it
    double v1[1] = {0.0};
    for (int i = tid; i < a.ntp; i += SMALL_BLOCK) v1[0] += a.tp_partials[i];
    blk_reduce<1>(v1, sh);
    const double tp = v1[0];
    if (!(fabs(tp) > 1e-40)) {  // KrySPcg.c:172-177: breakdown, nothing is updated
        if (tid == 0) { S.tp = tp; S.iter = it; S.stop = SPCG_DIV0; }
        return;
    }
    const double alpha = temp1 / tp;
    double red[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int i = tid; i < m; i += SMALL_BLOCK) {
        const double pi = a.p[i];
        const double ui = a.u[i] + alpha * pi;
        const double ri = a.r[i] - alpha * a.t[i];
        a.u[i] = ui; a.r[i] = ri;
        red[0] += ri * ri; red[1] += ui * ui; red[2] += pi * pi;
        red[3] = fmax(red[3], fabs(ui));
        red[4] += (ui != ui) ? 1.0 : 0.0;
    }
    blk_reduce<5>(red, sh, 1u << 3);
    const double absres = sqrt(red[0]), relres = absres / normr0;
    int    stop = SPCG_RUN, iter_best = S.iter_best;
    double best = absres_best;
    if (red[4] > 0.0) stop = SPCG_NAN;
    else {
        if (absres < absres_best - maxdiff) {
            best = absres; iter_best = it;
            for (int i = tid; i < m; i += SMALL_BLOCK) a.u_best[i] = a.u[i];
        }
        const double reldiff = fabs(alpha) * sqrt(red[2]) / sqrt(red[1]);
        if (red[3] <= 1e-20) stop = SPCG_SOLSTAG;                        // Check I
        else if ((stag <= 20) & (reldiff < maxdiff)) stop = SPCG_STAG;   // Check II fires: host recomputes r
        else if (relres < tol) stop = SPCG_CONV;                         // Check III fires: host checks the true residual
        const double beta = red[0] / temp1;
        for (int i = tid; i < m; i += SMALL_BLOCK) a.p[i] = 1.0 * a.r[i] + beta * a.p[i];
        if (stop == SPCG_RUN && it >= MaxIt) stop = SPCG_MAXIT;
    }
    if (tid == 0) {
        S.tp = tp; S.rr = red[0]; S.uu = red[1]; S.pp = red[2]; S.maxu = red[3]; S.nan = red[4];
        S.alpha = alpha; S.absres = absres; S.relres = relres;
        S.absres_best = best; S.iter_best = iter_best; S.iter = it;
        S.temp1_prev = temp1;
        S.temp1 = red[0];  // (z,r) of the next step, z = r
        S.stop = stop;
    }
}

// The same step with the vectors held in registers (m <= 512 * E): all loads of a step are issued
// together, so the kernel is one memory round trip plus two block reductions long.
template <int E>
__global__ __launch_bounds__(512) void k_spcg_step_reg(SpcgStepArgs a)
{
    constexpr int NT = 512, NW = NT / 64;
    __shared__ double sh[NW * 5];
    SpcgState& S = *a.st;
    if (S.stop != SPCG_RUN) return;
    const int tid = threadIdx.x, m = a.m;
    const double temp1 = S.temp1, absres_best = S.absres_best, normr0 = S.normr0, tol = S.tol, maxdiff = S.maxdiff;
    const int it = S.iter + 1, stag = S.stag, MaxIt = S.MaxIt;
    double pi[E], ti[E], ui[E], ri[E];
    double v1[1] = {0.0};
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = min(tid + e * NT, m - 1);
        pi[e] = a.p[i]; ti[e] = a.t[i]; ui[e] = a.u[i]; ri[e] = a.r[i];
    }
    for (int i = tid; i < a.ntp; i += NT) v1[0] += a.tp_partials[i];
    __syncthreads();  // every thread has read the state before thread 0 rewrites it
    blk_reduce<1, NW>(v1, sh);
    const double tp = v1[0];
    if (!(fabs(tp) > 1e-40)) {
        if (tid == 0) { S.tp = tp; S.iter = it; S.stop = SPCG_DIV0; }
        return;
    }
    const double alpha = temp1 / tp;
    double red[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (tid + e * NT < m) {
            ui[e] = ui[e] + alpha * pi[e];
            ri[e] = ri[e] - alpha * ti[e];
            red[0] += ri[e] * ri[e]; red[1] += ui[e] * ui[e]; red[2] += pi[e] * pi[e];
            red[3] = fmax(red[3], fabs(ui[e]));
            red[4] += (ui[e] != ui[e]) ? 1.0 : 0.0;
        }
    }
    blk_reduce<5, NW>(red, sh, 1u << 3);
    const double absres = sqrt(red[0]), relres = absres / normr0;
    int    stop = SPCG_RUN, iter_best = S.iter_best;
    double best = absres_best;
    const bool nan = red[4] > 0.0;
    bool better = false;
    double beta = 0.0;
    if (nan) stop = SPCG_NAN;
    else {
        better = absres < absres_best - maxdiff;
        if (better) { best = absres; iter_best = it; }
        const double reldiff = fabs(alpha) * sqrt(red[2]) / sqrt(red[1]);
        if (red[3] <= 1e-20) stop = SPCG_SOLSTAG;
        else if ((stag <= 20) & (reldiff < maxdiff)) stop = SPCG_STAG;
        else if (relres < tol) stop = SPCG_CONV;
        beta = red[0] / temp1;
        if (stop == SPCG_RUN && it >= MaxIt) stop = SPCG_MAXIT;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = tid + e * NT;
        if (i < m) {
            a.u[i] = ui[e]; a.r[i] = ri[e];
            if (!nan) {
                if (better) a.u_best[i] = ui[e];
                a.p[i] = 1.0 * ri[e] + beta * pi[e];
            }
        }
    }
    if (tid == 0) {
        S.tp = tp; S.rr = red[0]; S.uu = red[1]; S.pp = red[2]; S.maxu = red[3]; S.nan = red[4];
        S.alpha = alpha; S.absres = absres; S.relres = relres;
        S.absres_best = best; S.iter_best = iter_best; S.iter = it;
        S.temp1_prev = temp1;
        S.temp1 = red[0];
        S.stop = stop;
    }
}

// ---------------------------------------------------------------------------
// The batched safe CG with ONE launch per iteration (coarsest levels whose vectors fit the LDS of a CU).
// Every block repeats the step of k_spcg_step_reg on its own (same loads, same reduction order, hence the
// same alpha, residual and beta in every block, bit for bit), leaves p_new in its LDS and multiplies ITS rows
// of A with it, gathering from LDS instead of through the vector cache; block 0 does no rows: it carries
// the part of the step nobody else needs (u, the norms of u and p, the best iterate, the exit tests) and
// publishes r_new, p_new and the two scalars the next launch needs.  r, p, t and that broadcast record
// are double-buffered by launch parity, so within a launch nothing that is read is also written.
// (t,p) is formed from the finished vector t in the next launch's prologue: no per-block partials.
// ---------------------------------------------------------------------------
struct SpcgBc { double temp1; int stop, pad; };
struct SpcgFusedArgs {
    int        m, first, in;  // in: parity of the buffers holding r, p, t and the broadcast record on entry
    SpcgState* st;            // touched by block 0 only
    SpcgBc*    bc;            // [2]
    const int* ia; const int* ja; const unsigned short* ja16; const double* val;
    double *r[2], *p[2], *t[2];
    double *u, *u_best;
};

// One lane's share of a row sum, x gathered from LDS.  The row is consumed in chunks of 8 wave-strides whose
// 16 loads are all issued before the first use; the last chunk is padded (index clamped into the row, value
// replaced by 0), so a row of n entries costs ceil(n / 512) memory round trips, not one per leftover stride.
template <class IDX>
__device__ __forceinline__ double fused_row_sum(const IDX* __restrict__ ja, const double* __restrict__ val,
                                                const double* sp, int k, int ke)
{
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (; k - (int)(threadIdx.x & 63) < ke; k += 8 * 64) {   // (uniform trip count over the wave)
        int c[8]; double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int kk = k + q * 64, kc = min(kk, ke - 1);
            c[q] = (int)__builtin_nontemporal_load(ja + kc);
            const double x = __builtin_nontemporal_load(val + kc);
            v[q] = kk < ke ? x : 0.0;
        }
        s0 += v[0] * sp[c[0]]; s1 += v[1] * sp[c[1]]; s2 += v[2] * sp[c[2]]; s3 += v[3] * sp[c[3]];
        s0 += v[4] * sp[c[4]]; s1 += v[5] * sp[c[5]]; s2 += v[6] * sp[c[6]]; s3 += v[7] * sp[c[7]];
    }
    return (s0 + s1) + (s2 + s3);
}

// Start of a coarse solve without a host round trip (KrySPcg.c:88-135): r = b (u = 0 on entry from the cycle; otherwise
// the caller has put b - A u into r), u_best = 0, p = r, and the iteration state from (r,r).
struct SpcgInitArgs {
    int m, x_zero, MaxIt;
    double tol, maxdiff;
    const double* b;
    double *u, *r, *p, *u_best;
    SpcgState* st;
};
__global__ __launch_bounds__(512) void k_spcg_init(SpcgInitArgs a)
{
    constexpr int NT = 512, NW = NT / 64;
    __shared__ double sh[NW];
    double v[1] = {0.0};
    for (int i = threadIdx.x; i < a.m; i += NT) {
        double ri;
        if (a.x_zero) { ri = a.b[i]; a.r[i] = ri; a.u[i] = 0.0; }
        else ri = a.r[i];
        a.p[i] = ri;
        a.u_best[i] = 0.0;
        v[0] += ri * ri;
    }
    blk_reduce<1, NW>(v, sh);
    if (threadIdx.x == 0) {
        const double absres0 = sqrt(v[0]), normr0 = fmax(1e-20, absres0);  // SMALLREAL
        SpcgState S{};
        S.temp1 = v[0]; S.temp1_prev = v[0]; S.absres_best = 1e+20; S.normr0 = normr0; S.tol = a.tol; S.maxdiff = a.maxdiff;
        S.absres = 1e+20; S.relres = absres0 / normr0; S.rr = v[0];
        S.iter = 0; S.iter_best = 0; S.stag = 1; S.MaxIt = a.MaxIt; S.pad = 0;
        S.stop = (S.relres < a.tol) ? SPCG_ZERO_RHS : SPCG_RUN;
        *a.st = S;
    }
}

template <int E>
__global__ __launch_bounds__(512) void k_spcg_fused(SpcgFusedArgs a)
{
    constexpr int NT = 512, NW = NT / 64;
    extern __shared__ double sp[];  // p_new, m doubles
    __shared__ double sh[NW * 4];
    const int tid = threadIdx.x, m = a.m, in = a.in, out = in ^ 1;
    const bool lead = blockIdx.x == 0;
    if (a.first) {
        // first launch of a batch: no step, p = the host's p; block 0 moves r, p and the scalars to the other parity
        for (int i = tid; i < m; i += NT) {
            const double pi = a.p[in][i];
            sp[i] = pi;
            if (lead) { a.p[out][i] = pi; a.r[out][i] = a.r[in][i]; }
        }
        // (a zero right-hand side: k_spcg_init has raised the stop already; this product is then idle work)
        if (lead && tid == 0) { a.bc[out].temp1 = a.st->temp1; a.bc[out].stop = a.st->stop; a.st->pad = out; }
    } else {
        const SpcgBc bc = a.bc[in];
        if (bc.stop != SPCG_RUN) {
            if (lead && tid == 0) a.bc[out] = bc;  // hand the verdict on to the launches queued behind
            return;
        }
        const double temp1 = bc.temp1;
        const double *rp = a.r[in], *pp = a.p[in], *tp_ = a.t[in];
        double pi[E], ti[E], ri[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = min(tid + e * NT, m - 1);
            pi[e] = pp[i]; ti[e] = tp_[i]; ri[e] = rp[i];
        }
        double v1[1] = {0.0};
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (tid + e * NT < m) v1[0] += ti[e] * pi[e];
        blk_reduce<1, NW>(v1, sh);
        const double tp = v1[0];
        SpcgState& S = *a.st;
        if (!(fabs(tp) > 1e-40)) {  // KrySPcg.c:172-177: breakdown, nothing is updated
            if (lead && tid == 0) {
                S.tp = tp; S.iter = S.iter + 1; S.stop = SPCG_DIV0;
                a.bc[out].temp1 = temp1; a.bc[out].stop = SPCG_DIV0;
            }
            return;
        }
        const double alpha = temp1 / tp;
        v1[0] = 0.0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (tid + e * NT < m) {
                ri[e] = ri[e] - alpha * ti[e];
                v1[0] += ri[e] * ri[e];
            }
        }
        blk_reduce<1, NW>(v1, sh);
        const double rr = v1[0], beta = rr / temp1;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = tid + e * NT;
            if (i < m) sp[i] = 1.0 * ri[e] + beta * pi[e];
        }
        if (lead) {
            const double absres_best = S.absres_best, normr0 = S.normr0, tol = S.tol, maxdiff = S.maxdiff;
            const int it = S.iter + 1, stag = S.stag, MaxIt = S.MaxIt;
            int iter_best = S.iter_best;
            double red[4] = {0.0, 0.0, 0.0, 0.0};  // (u,u), (p,p), max |u|, NaN count
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = tid + e * NT;
                if (i < m) {
                    const double ui = a.u[i] + alpha * pi[e];
                    a.u[i] = ui;
                    a.r[out][i] = ri[e];
                    a.p[out][i] = sp[i];
                    red[0] += ui * ui; red[1] += pi[e] * pi[e];
                    red[2] = fmax(red[2], fabs(ui));
                    red[3] += (ui != ui) ? 1.0 : 0.0;
                }
            }
            blk_reduce<4, NW>(red, sh, 1u << 2);  // (also orders the reads of S above before thread 0 rewrites it)
            const double absres = sqrt(rr), relres = absres / normr0;
            int    stop = SPCG_RUN;
            double best = absres_best;
            if (red[3] > 0.0) stop = SPCG_NAN;
            else {
                if (absres < absres_best - maxdiff) {
                    best = absres; iter_best = it;
                    for (int i = tid; i < m; i += NT) a.u_best[i] = a.u[i];  // own stores of this thread
                }
                const double reldiff = fabs(alpha) * sqrt(red[1]) / sqrt(red[0]);
                if (red[2] <= 1e-20) stop = SPCG_SOLSTAG;                        // Check I
                else if ((stag <= 20) & (reldiff < maxdiff)) stop = SPCG_STAG;   // Check II: host recomputes r
                else if (relres < tol) stop = SPCG_CONV;                         // Check III: host checks the true residual
                if (stop == SPCG_RUN && it >= MaxIt) stop = SPCG_MAXIT;
            }
            if (tid == 0) {
                S.tp = tp; S.rr = rr; S.uu = red[0]; S.pp = red[1]; S.maxu = red[2]; S.nan = red[3];
                S.alpha = alpha; S.absres = absres; S.relres = relres;
                S.absres_best = best; S.iter_best = iter_best; S.iter = it;
                S.temp1_prev = temp1;
                S.temp1 = rr;
                S.stop = stop;
                S.pad = out;  // parity of the buffers that now hold r and p
                a.bc[out].temp1 = rr; a.bc[out].stop = stop;
            }
            return;
        }
    }
    if (lead) return;
    __syncthreads();
    // t_out = A p_new on the rows of this block: one wave per row, rows dealt round-robin over all waves
    const int lane = tid & 63;
    const int nwaves = ((int)gridDim.x - 1) * NW;
    double* t_out = a.t[out];
    for (int row = ((int)blockIdx.x - 1) * NW + (tid >> 6); row < m; row += nwaves) {
        const int kb = a.ia[row], ke = a.ia[row + 1];
        double s = 0.0;
        if (ke > kb) s = a.ja16 ? fused_row_sum(a.ja16, a.val, sp, kb + lane, ke) : fused_row_sum(a.ja, a.val, sp, kb + lane, ke);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
        if (lane == 0) t_out[row] = s;
    }
}

// ---------------------------------------------------------------------------
// Variable-restart GMRES without preconditioner, STOP_REL_RES (KryPvgmres.c:66 / :416)
// ---------------------------------------------------------------------------
template <class OP>
struct GmresArgs {
    OP            A;
    const double* b;
    double*       x;
    double*       ws;  // (restart + 2) vectors of length n: p[0..restart], w
    double        tol, abstol;
    int           MaxIt, restart;
    SmallOut*     out;
    int           cache2;   // LV only: rows beyond the first 512 have GM_CB * 28 bytes each of LDS behind the basis (k_gmres_small: the matrix held on chip)
};
constexpr int GM_CB = 14;   // blocks of a row k_gmres_small holds on chip (nb = 3)
#ifndef GM_OV
#define GM_OV 3             // blocks beyond them in flight per trip
#endif

// LV: the Krylov basis p[0..restart] and w live in dynamic LDS ((restart + 2) n doubles) instead of a.ws
// Modified Gram-Schmidt of the new Krylov vector `pi` against basis vectors 0 .. i-1 (n doubles apart) in ONE wavefront:
// NE elements per lane in registers, a dot product = NE multiply-adds per lane (four interleaved partial sums) + the
// wavefront's DPP sum, no LDS partials, no barrier.  Coefficient j comes back in lane j (hlane); returns the norm of what is
// left and leaves pi normalised (KryPvgmres.c:207-233).
template <int NE>
__device__ __forceinline__ double gm_mgs_wave(int n, int i, int lane, const double* basis, double* pi, double& hlane)
{
    // NE = ceil(n / 64) exactly: elements k < NE - 1 exist in every lane, element NE - 1 in the lanes below n - 64 (NE - 1).  Every read
    // is unconditional (the last element's index clamped into the vector, its value zeroed by a select): a read under a lane predicate is
    // a block of its own for the compiler, and at the join behind it the wait-count pass no longer knows that the NEXT vector's reads were
    // issued after this one's -- it waited for all of them (lgkmcnt(0)) in front of every dot product, which is what the two register
    // sets are there to avoid.
    const bool vl = lane + 64 * (NE - 1) < n;
    const int  il = min(lane + 64 * (NE - 1), n - 1);
    double wv[NE];
#pragma unroll
    for (int k = 0; k < NE - 1; ++k) wv[k] = pi[lane + 64 * k];
    { const double x = pi[il]; wv[NE - 1] = vl ? x : 0.0; }
    // basis vector j + 1 is read while the dot product with vector j is summed over the wavefront (the coefficient of one vector is
    // needed before the next dot product, the vector itself is not): two register sets, the read of the next one unconditional
    // (past the last vector: the last one again) so that it stays in flight across the use of the current one
    auto load = [&](double (&pv)[NE], int j) {
        const double* pj = basis + (size_t)j * n;
#pragma unroll
        for (int k = 0; k < NE - 1; ++k) pv[k] = pj[lane + 64 * k];
        pv[NE - 1] = pj[il];
    };
    double hmine = 0.0;   // lane j keeps coefficient j; they leave together behind the loop (at most 30 of them)
    auto step = [&](double (&pv)[NE], int j) {
        pv[NE - 1] = vl ? pv[NE - 1] : 0.0;
        double hs[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < NE; ++k) hs[k & 3] += pv[k] * wv[k];
        const double h = wave_allsum((hs[0] + hs[1]) + (hs[2] + hs[3]));
        hmine = (lane == j) ? h : hmine;
#pragma unroll
        for (int k = 0; k < NE; ++k) wv[k] += -h * pv[k];
    };
    if constexpr (NE <= 12) {
        double pa[NE], pb[NE];
        int j = 0;
        if (i > 0) load(pa, 0);
        for (; j + 1 < i; j += 2) {
            load(pb, j + 1);
            __builtin_amdgcn_sched_barrier(0);
            step(pa, j);
            load(pa, min(j + 2, i - 1));
            __builtin_amdgcn_sched_barrier(0);
            step(pb, j + 1);
        }
        if (j < i) step(pa, j);   // (an odd count: the last vector is the one read ahead)
    } else {   // (more elements per lane: no registers for a second set beside the matrix the kernel keeps)
        double pa[NE];
        for (int j = 0; j < i; ++j) { load(pa, j); step(pa, j); }
    }
    hlane = hmine;   // (lane j < i: the coefficient against basis vector j -- the caller rotates the column where it sits)
    double ts[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < NE; ++k) ts[k & 3] += wv[k] * wv[k];
    const double t = sqrt(wave_allsum((ts[0] + ts[1]) + (ts[2] + ts[3])));
    if (t != 0.0) {
        const double s = 1.0 / t;
#pragma unroll
        for (int k = 0; k < NE; ++k) wv[k] *= s;
    }
#pragma unroll
    for (int k = 0; k < NE - 1; ++k) pi[lane + 64 * k] = wv[k];
    if (vl) pi[il] = wv[NE - 1];
    return t;
}

#ifdef GM_TIMING
#define GMT(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); gmt[k] += now_ - gmt_last; gmt_last = now_; } while (0)
#else
#define GMT(k)
#endif
template <class OP, bool LV>
__global__ __launch_bounds__(SMALL_BLOCK) void k_gmres_small(GmresArgs<OP> a)
{
#ifdef GM_TIMING
    unsigned long long gmt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gmt_last = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int R = SMALL_MAX_RESTART;
    extern __shared__ double dyn[];
    __shared__ double sh[SMALL_WAVES];
    __shared__ double sh2[2 * SMALL_WAVES];
    int par = 0;
    __shared__ double hh[(R + 1) * R];  // hh[j][k] -> hh[j * R + k]
    __shared__ double rs[R + 2], c[R + 1], sn[R + 1];
    __shared__ double sc_rnorm;         // |rs[i]| published by thread 0
    const OP A = a.A;
    const int n = A.rows(), tid = threadIdx.x;
    const double tol = a.tol, abstol = a.abstol, epsmac = 1e-20, cr_max = 0.99, cr_min = 0.174;
    const int MaxIt = a.MaxIt, restart_max = a.restart, restart_min = 3, d = 3;
    int iter = 0, i = 0, Restart = a.restart;
    double r_norm, r_norm_old = 0.0, absres0, absres = 1e+20, relres, cr = 1.0, t;
    double g_c = 0.0, g_s = 0.0;   // wavefront 0: lane j keeps Givens rotation j - 1 of the current restart cycle
    const double* b = a.b;
    double* x = a.x;
    double* const basis = LV ? dyn : a.ws;
    auto P = [&](int k) { return basis + (size_t)k * n; };
    double* w = basis + (size_t)(a.restart + 1) * n;

    // nb = 3 systems of at most 1 024 rows keep the MATRIX on chip for the whole solve: a thread holds the first GM_CB
    // blocks of its row t (its three values per block + the column) in registers, the rows t + 512 sit in LDS behind
    // the basis.  A product then reads x (the basis vector, in LDS) and nothing else -- from memory it was two passes of
    // four dependent round trips through the L2, 12 us of an iteration's 12.5 on config 3's coarsest level (555 rows,
    // 5 297 iterations per solve).  Same blocks in the same order: identical bits.
    const bool oc = LV && A.nb == 3 && n <= 2 * SMALL_BLOCK && (n <= SMALL_BLOCK || a.cache2);
    double ca[GM_CB][3];
    unsigned cj[GM_CB];   // low half: the block columns of row tid; high half: those of row tid + 512, whose values sit in LDS (at most 342 block rows)
    int    cnt0 = 0, kb0 = 0, r0 = 0, cnt1 = 0, kb1 = 0, r1 = 0;
    static_assert(GM_CB % 7 == 0, "k_gmres_small: the on-chip product works in groups of seven blocks");
    const int n1 = max(n - SMALL_BLOCK, 0);
    double* c1a = dyn + (size_t)(a.restart + 2) * n;                     // [n1][GM_CB][3]
    int*    c1j = reinterpret_cast<int*>(c1a + (size_t)n1 * GM_CB * 3);  // [n1][GM_CB]
    if (oc) {
#pragma unroll
        for (int q = 0; q < GM_CB; ++q) cj[q] = 0u;
        if (tid < n) {
            const int br = tid / 3;
            r0 = tid - br * 3; kb0 = A.ia[br]; cnt0 = A.ia[br + 1] - kb0;
#pragma unroll
            for (int q = 0; q < GM_CB; ++q) {
                const int k = kb0 + min(q, max(cnt0 - 1, 0));
                cj[q] = (unsigned)A.ja[k];
                const double* B = A.val + (size_t)k * 9 + r0 * 3;
                ca[q][0] = B[0]; ca[q][1] = B[1]; ca[q][2] = B[2];
            }
        }
        if (tid < n1) {
            const int row = tid + SMALL_BLOCK, br = row / 3;
            r1 = row - br * 3; kb1 = A.ia[br]; cnt1 = A.ia[br + 1] - kb1;
#pragma unroll
            for (int q = 0; q < GM_CB; ++q) cj[q] |= (unsigned)A.ja[kb1 + min(q, max(cnt1 - 1, 0))] << 16;
            for (int q = 0; q < min(cnt1, GM_CB); ++q) {
                c1j[tid * GM_CB + q] = A.ja[kb1 + q];
                const double* B = A.val + (size_t)(kb1 + q) * 9 + r1 * 3;
                double* C = c1a + ((size_t)tid * GM_CB + q) * 3;
                C[0] = B[0]; C[1] = B[1]; C[2] = B[2];
            }
        }
    }
    // y-callback f(row, acc): acc = (seed ? -seed_row : 0) + sum of the row's block products in storage order
    auto rows_onchip = [&](const double* xin, const double* seed, auto&& f) {
        if (tid < n) {
            double acc = seed ? -seed[tid] : 0.0;
            // the operands of seven blocks at a time, all of them (the column of a block the row does not have is that of its last
            // one: a valid address), THEN the products, added in storage order by selects: a branch per block made fourteen LDS round
            // trips one after the other of what is two
#pragma unroll
            for (int q0 = 0; q0 < GM_CB; q0 += 7) {
                double xv[7][3];
#pragma unroll
                for (int e = 0; e < 7; ++e) {
                    const double* xb = xin + (size_t)(cj[q0 + e] & 0xffffu) * 3;
                    xv[e][0] = xb[0]; xv[e][1] = xb[1]; xv[e][2] = xb[2];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 7; ++e) {
                    double sq = ca[q0 + e][0] * xv[e][0];
                    sq = sq + ca[q0 + e][1] * xv[e][1];
                    sq = sq + ca[q0 + e][2] * xv[e][2];
                    acc = (q0 + e < cnt0) ? acc + sq : acc;
                }
            }
            // a row of more than GM_CB blocks: the rest from memory (the L1 after the first product: nothing else of this kernel goes through
            // it) -- a third of config 3's coarsest rows have up to nine such blocks, and one block per trip made their wavefronts the last
            // at the barrier by 1.6 us.  GM_OV blocks' columns and values in flight per trip, added in storage order by selects.
            for (int k0 = kb0 + GM_CB; k0 < kb0 + cnt0; k0 += GM_OV) {
                int    jj[GM_OV];
                double bv[GM_OV][3], xv[GM_OV][3];
#pragma unroll
                for (int e = 0; e < GM_OV; ++e) {
                    const int kk = min(k0 + e, kb0 + cnt0 - 1);
                    const double* B = A.val + (size_t)kk * 9 + r0 * 3;
                    jj[e] = A.ja[kk];
                    bv[e][0] = B[0]; bv[e][1] = B[1]; bv[e][2] = B[2];
                }
#pragma unroll
                for (int e = 0; e < GM_OV; ++e) {
                    const double* xb = xin + (size_t)jj[e] * 3;
                    xv[e][0] = xb[0]; xv[e][1] = xb[1]; xv[e][2] = xb[2];
                }
#pragma unroll
                for (int e = 0; e < GM_OV; ++e) {
                    double sq = bv[e][0] * xv[e][0];
                    sq = sq + bv[e][1] * xv[e][1];
                    sq = sq + bv[e][2] * xv[e][2];
                    acc = (k0 + e < kb0 + cnt0) ? acc + sq : acc;
                }
            }
            f(tid, acc);
        }
        GMT(5);   // (lab: the first 512 rows are done)
        if (tid < n1) {
            const int row = tid + SMALL_BLOCK;
            double acc = seed ? -seed[row] : 0.0;
            const int nq = min(cnt1, GM_CB);
            // four blocks' operands in flight (values from LDS, columns from registers: one round trip, not two), added in storage order
            // by selects.  (Seven, as above, do not fit beside the first rows' matrix registers.)
#pragma unroll
            for (int q0 = 0; q0 < GM_CB; q0 += 4) {
                double cv[4][3], xv[4][3];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int q = q0 + e < GM_CB ? q0 + e : GM_CB - 1;
                    const double* C = c1a + ((size_t)tid * GM_CB + min(q, max(nq - 1, 0))) * 3;
                    const double* xb = xin + (size_t)(cj[q] >> 16) * 3;
                    cv[e][0] = C[0]; cv[e][1] = C[1]; cv[e][2] = C[2];
                    xv[e][0] = xb[0]; xv[e][1] = xb[1]; xv[e][2] = xb[2];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    double v = cv[e][0] * xv[e][0];
                    v = v + cv[e][1] * xv[e][1];
                    v = v + cv[e][2] * xv[e][2];
                    acc = (q0 + e < nq) ? acc + v : acc;
                }
            }
            for (int k = kb1 + GM_CB; k < kb1 + cnt1; ++k) {
                const double* B = A.val + (size_t)k * 9 + r1 * 3;
                const double* xb = xin + (size_t)A.ja[k] * 3;
                double sq = B[0] * xb[0];
                sq = sq + B[1] * xb[1];
                sq = sq + B[2] * xb[2];
                acc += sq;
            }
            f(row, acc);
        }
        GMT(6);   // (lab: the rows beyond 512)
        __syncthreads();
    };
    auto g_mxv = [&](const double* xin, double* y) {
        if (oc) rows_onchip(xin, nullptr, [&](int row, double acc) { y[row] = acc; });
        else small_mxv(A, xin, y);
    };
    auto g_resid = [&](const double* xin, const double* bb, double* rr) {   // -((-b_i) + t) (BlaSpmvBSR: the reference's rounding)
        if (oc) rows_onchip(xin, bb, [&](int row, double acc) { rr[row] = -acc; });
        else small_resid(A, xin, bb, rr);
    };
    if (oc) __syncthreads();

    g_resid(x, b, P(0));
    r_norm = sqrt(blk_dot(n, P(0), P(0), sh));
    absres0 = fmax(1e-20, r_norm);
    relres = r_norm / absres0;
    if (relres < tol || absres0 < abstol) goto FINISHED;

    while (iter < MaxIt) {
        r_norm_old = r_norm;
        if (tid == 0) rs[0] = r_norm;
        {
            const double s = 1.0 / r_norm;
            double* p0 = P(0);
            for (int e = tid; e < n; e += SMALL_BLOCK) p0[e] *= s;
        }
        if (cr > cr_max || iter == 0) Restart = restart_max;
        else if (cr < cr_min) { /* keep */ }
        else { if (Restart - d > restart_min) Restart -= d; else Restart = restart_max; }
        __syncthreads();

        i = 0;
        while (i < Restart && iter < MaxIt) {
            i++; iter++;
            double* pi = P(i);
            GMT(0);
            g_mxv(P(i - 1), pi);
            GMT(1);
            if (n <= 64 * GM_NE) {
                // Modified Gram-Schmidt in ONE wavefront: the new vector in its registers (GM_NE elements per lane), a dot product
                // = GM_NE multiply-adds per lane + a DPP sum, no LDS partials, no barrier -- the chain of i dependent
                // (dot, update) pairs was half of an iteration when eight wavefronts met at a barrier for every one of them
                // (config 3: 555 rows, 5 297 coarse iterations per solve).  Sums: per lane over ascending elements, then the
                // wavefront's fixed DPP order.
                if (tid < 64) {
                    const int ne = (n + 63) >> 6;   // elements per lane: the instantiation that has exactly as many
                    double hl = 0.0;
                    switch (ne) {
#define FASP_GM_CASE(q) case q: t = gm_mgs_wave<q>(n, i, tid, basis, pi, hl); break;
                        FASP_GM_CASE(1) FASP_GM_CASE(2) FASP_GM_CASE(3) FASP_GM_CASE(4) FASP_GM_CASE(5) FASP_GM_CASE(6) FASP_GM_CASE(7) FASP_GM_CASE(8)
                        FASP_GM_CASE(9) FASP_GM_CASE(10) FASP_GM_CASE(11) FASP_GM_CASE(12) FASP_GM_CASE(13) FASP_GM_CASE(14) FASP_GM_CASE(15)
#undef FASP_GM_CASE
                        default: t = gm_mgs_wave<GM_NE>(n, i, tid, basis, pi, hl); break;
                    }
                    GMT(2);
                    // Givens rotations on the new Hessenberg column (KryPvgmres.c:236-259) where it sits.  Rotation j - 1 = (c, s) lives in
                    // lane j since the iteration that formed it, entry h_j in lane j.  What the reference's loop carries from rotation to
                    // rotation is ONE number -- tt_j = -s tt_{j-1} + c h_j, the entry the next rotation pairs with -- so that is all that
                    // has to go from lane to lane (one wavefront shift, one product, one sum per rotation; c h_j is formed beforehand in
                    // every lane at once); the other halves, s h_j + c tt_{j-1}, are one parallel step at the end.  Same products, same
                    // sums.  One thread walking the column through LDS took 1.0 of an iteration's 7.4 us (config 3, 15 rotations on
                    // average): a chain of LDS round trips and four dependent operations per rotation.
                    auto rdl = [&](double x, int from) -> double {
                        const unsigned long long bb = (unsigned long long)__double_as_longlong(x);
                        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bb, from);
                        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bb >> 32), from);
                        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
                    };
                    auto shr1 = [&](double x) -> double {   // lane j takes lane j - 1's value (wave_shr:1; lane 0: 0)
                        const long long bb = __double_as_longlong(x);
                        int lo = (int)(bb & 0xffffffffll), hi = (int)(bb >> 32);
                        lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
                        hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
                        return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
                    };
                    const double rprev = rs[i - 1];
                    const double bj = g_c * hl;          // c_{j-1} h_j
                    const double ns = -g_s;
                    double cur = hl;                     // tt_0 = h_0 in lane 0; lane j is right after step j
                    for (int st = 1; st < i; ++st) {
                        const double nc = ns * shr1(cur) + bj;
                        cur = (tid >= 1) ? nc : cur;
                    }
                    const double np = g_s * hl + g_c * shr1(cur);   // the final entry j - 1, formed in lane j (1 <= j < i)
                    const double curL = rdl(cur, i - 1);            // what the new rotation pairs with t
                    double g = t * t;
                    g += curL * curL;
                    double gamma = sqrt(g);
                    if (gamma == 0.0) gamma = epsmac;
                    const double ci = curL / gamma, si = t / gamma;
                    const double rsi = -si * rprev, rsp = ci * rprev;
                    const double hfin = si * t + ci * curL;
                    g_c = (tid == i) ? ci : g_c;
                    g_s = (tid == i) ? si : g_s;
                    if (tid >= 1 && tid < i) hh[(tid - 1) * R + (i - 1)] = np;
                    if (tid == 0) {
                        hh[(i - 1) * R + (i - 1)] = hfin; hh[i * R + (i - 1)] = t;
                        c[i - 1] = ci; sn[i - 1] = si; rs[i] = rsi; rs[i - 1] = rsp; sc_rnorm = fabs(rsi);
                    }
                }
            } else {
            // modified Gram-Schmidt: a thread updates only its own elements between the dots
            for (int j = 0; j < i; ++j) {
                const double* pj = P(j);
                double hv[1] = {0.0};
                for (int e = tid; e < n; e += SMALL_BLOCK) hv[0] += pj[e] * pi[e];
                blk_reduce1<1>(hv, sh2, par);
                const double h = hv[0];
                if (tid == 0) hh[j * R + (i - 1)] = h;
                for (int e = tid; e < n; e += SMALL_BLOCK) pi[e] += -h * pj[e];
            }
            {
                double tv[1] = {0.0};
                for (int e = tid; e < n; e += SMALL_BLOCK) tv[0] += pi[e] * pi[e];
                blk_reduce1<1>(tv, sh2, par);
                t = sqrt(tv[0]);
            }
            if (t != 0.0) {
                const double s = 1.0 / t;
                for (int e = tid; e < n; e += SMALL_BLOCK) pi[e] *= s;
            }
            }
            if (!(n <= 64 * GM_NE)) GMT(2);
            if (!(n <= 64 * GM_NE) && tid == 0) {  // Givens rotations on the new Hessenberg column
                hh[i * R + (i - 1)] = t;
                for (int j = 1; j < i; ++j) {
                    const double tt = hh[(j - 1) * R + (i - 1)];
                    hh[(j - 1) * R + (i - 1)] = sn[j - 1] * hh[j * R + (i - 1)] + c[j - 1] * tt;
                    hh[j * R + (i - 1)] = -sn[j - 1] * tt + c[j - 1] * hh[j * R + (i - 1)];
                }
                double g = hh[i * R + (i - 1)] * hh[i * R + (i - 1)];
                g += hh[(i - 1) * R + (i - 1)] * hh[(i - 1) * R + (i - 1)];
                double gamma = sqrt(g);
                if (gamma == 0.0) gamma = epsmac;
                c[i - 1] = hh[(i - 1) * R + (i - 1)] / gamma;
                sn[i - 1] = hh[i * R + (i - 1)] / gamma;
                rs[i] = -sn[i - 1] * rs[i - 1];
                rs[i - 1] = c[i - 1] * rs[i - 1];
                hh[(i - 1) * R + (i - 1)] = sn[i - 1] * hh[i * R + (i - 1)] + c[i - 1] * hh[(i - 1) * R + (i - 1)];
                sc_rnorm = fabs(rs[i]);
            }
            GMT(3);
            __syncthreads();
            GMT(4);
            absres = r_norm = sc_rnorm;
            relres = absres / absres0;
            if (relres < tol) break;
        }

        if (tid == 0) {  // back substitution
            rs[i - 1] = rs[i - 1] / hh[(i - 1) * R + (i - 1)];
            for (int k = i - 2; k >= 0; k--) {
                double tt = 0.0;
                int j = k + 1;
                for (; j + 5 < i; j += 6) {   // six terms' operands in flight, subtracted in the reference's order (one term per LDS round trip
                                              // made the 435 terms of a full restart 18 us: 0.6 us of every iteration on config 3)
                    double hv[6], rv[6];
#pragma unroll
                    for (int e = 0; e < 6; ++e) { hv[e] = hh[k * R + j + e]; rv[e] = rs[j + e]; }
#pragma unroll
                    for (int e = 0; e < 6; ++e) tt -= hv[e] * rv[e];
                }
                for (; j < i; j++) tt -= hh[k * R + j] * rs[j];
                tt += rs[k];
                rs[k] = tt / hh[k * R + k];
            }
        }
        __syncthreads();
        // w = rs[i-1] p[i-1] + rs[i-2] p[i-2] + ... + rs[0] p[0];  x += w
        for (int e = tid; e < n; e += SMALL_BLOCK) {
            double we = P(i - 1)[e] * rs[i - 1];
            for (int j = i - 2; j >= 0; j--) we += rs[j] * P(j)[e];
            w[e] = we;
            x[e] += we;
        }
        __syncthreads();

        if (relres < tol) {  // check the true residual
            g_resid(x, b, w);
            r_norm = sqrt(blk_dot(n, w, w, sh));
            absres = r_norm;
            relres = absres / absres0;
            if (relres < tol) break;
            for (int e = tid; e < n; e += SMALL_BLOCK) P(0)[e] = w[e];
            i = 0;
        }

        // residual vector of the restart (KryPvgmres.c:390-401)
        if (tid == 0)
            for (int j = i; j > 0; j--) {
                rs[j - 1] = -sn[j - 1] * rs[j];
                rs[j] = c[j - 1] * rs[j];
            }
        __syncthreads();
        if (i) {
            double* pi = P(i);
            double* p0 = P(0);
            for (int e = tid; e < n; e += SMALL_BLOCK) {
                double v = pi[e];
                v = v + (rs[i] - 1.0) * v;
                for (int j = i - 1; j > 0; j--) v += rs[j] * P(j)[e];
                pi[e] = v;
                double q = p0[e];
                q = q + (rs[0] - 1.0) * q;
                p0[e] = q + v;
            }
        }
        __syncthreads();
        cr = r_norm / r_norm_old;
    }

FINISHED:
#ifdef GM_TIMING
    if (tid == 0 && iter > 20) printf("[gmres_small] iters %d: other %.2f spmv %.2f (wavefront 0: rows < 512 %.2f, rows >= 512 %.2f, barrier %.2f) mgs %.2f givens %.2f barrier %.2f us per iteration\n", iter, gmt[0] * 0.01 / iter, (gmt[1] + gmt[5] + gmt[6]) * 0.01 / iter, gmt[5] * 0.01 / iter, gmt[6] * 0.01 / iter, gmt[1] * 0.01 / iter, gmt[2] * 0.01 / iter, gmt[3] * 0.01 / iter, gmt[4] * 0.01 / iter);
#endif
    if (tid == 0) {
        a.out->iters = iter;
        a.out->status = iter >= MaxIt ? -48 : iter;
        a.out->relres = relres;
        a.out->absres = absres;
    }
}

// ---------------------------------------------------------------------------
// k_spcg_persist<NE>: the safe CG of a coarsest level too large for one CU (P7(256): 4 971 rows, 6.4 M
// entries = 64 MB) as ONE launch per coarse solve, the matrix resident ON CHIP for all its iterations.
//
//   * The chip's register files hold the matrix: wave w of block b >= 1 owns up to 4 whole rows (assigned on the
//     host, longest-processing-time first: 56 slots of 64 entries against an ideal of 50 on P7(256)) and keeps
//     their values (NE doubles per lane) and LDS byte offsets of their columns (NE 16-bit values per lane) in
//     VGPRs from the first iteration to the last.  64 MB are read once per solve instead of once per iteration.
//   * Every block keeps r and p in its LDS and repeats the scalar recurrence on its own (same loads, same
//     reduction order -> bit-identical alpha, beta, ||r|| in every block: k_spcg_fused's scheme); p is gathered
//     from LDS by the row sums.
//   * Per iteration ONE exchange and no meeting (round 3): a finished row leaves as a self-validating 16-byte record
//     {low word, epoch, high word, epoch} -- two 8-byte units, each atomic on its own, written with ONE write-through
//     (sc1) store and nothing behind it: no drain of the stores, no arrival counter, no block barrier before the
//     read.  Every block polls the m records with 16-byte sc1 loads until all carry this iteration's epoch; the
//     epoch is unique over launches and iterations, the records are double-buffered by iteration parity (a block can
//     run at most one iteration ahead of the slowest: it needs everybody's products to get further), so a stale
//     record is never taken for a fresh one.  Round 2's meeting (stores drained, shard counter, top counter, poll:
//     four L2 round trips, 3.9 of 12 us per iteration) is gone; what remains is store -> L2 -> load.
//   * Block 0 owns no rows; it carries what nobody else needs (u, its norms, the best iterate, the reference's
//     exit tests, KrySPcg.c:147-199) and publishes its verdict as one more record, which every block polls with the
//     products of the next iteration -- that also keeps the grid within one iteration of block 0.
//   The host queues nothing: it reads the state once per solve and replays a fired test exactly as for
//   k_spcg_fused (coarse_cg.hip.h).  Every spin is bounded (SPCG_HANG + error word).
// ---------------------------------------------------------------------------
constexpr int SPCG_HANG = 8;
#ifndef SPCG_POLL_GROUP
#define SPCG_POLL_GROUP 6   // records a thread polls at once (k_spcg_persist; 12 = all of them: lab builds)
#endif

struct SpcgPersistArgs {
    int        m, max_steps, nblocks;
    int        u_lds;                  // the launch carries a fourth LDS vector: block 0 keeps u on chip
    SpcgState* st;
    double    *r, *p, *u, *u_best;     // r, p: in / out (block 0 writes them back at the end)
    double*    t2;                     // [2][m] published products as 16-byte records {lo, epoch, hi, epoch}
    unsigned*  sync;                   // [3] error word, [8..11] block 0's verdict records {stop, epoch} by parity; zeroed before the launch
    unsigned   epoch0;                 // epoch of iteration s of this launch: epoch0 + s + 1 (unique over launches)
    const double*         vals;        // [(nblocks-1)*8][NE][64]
    const unsigned short* cols;        // same shape: column * 8 (byte offset into p's LDS image)
    const int*            wrow;        // [(nblocks-1)*8][8] rows of the wave (-1: none)
    const int*            wend;        // [(nblocks-1)*8][8] end slot (exclusive) of each of them
};

// workgroup barrier that orders LDS traffic only: global stores in flight (the published products, the best iterate)
// stay in flight across it -- __syncthreads() would wait for every one of them to be acknowledged
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// blk_reduce with LDS-only barriers (global stores in flight stay in flight)
template <int NQ, int NW>
__device__ __forceinline__ void blk_reduce_dpp(double (&v)[NQ], double* sh, unsigned maxmask = 0u)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const bool mx = (maxmask >> q) & 1u;
        const double x = mx ? wave_max_to_lane63(v[q]) : wave_sum_to_lane63(v[q]);
        if (lane == 63) sh[w * NQ + q] = x;
    }
    lds_barrier();
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const bool mx = (maxmask >> q) & 1u;
        double x = sh[q];
        for (int k = 1; k < NW; ++k) {
            const double y = sh[k * NQ + q];
            x = mx ? fmax(x, y) : x + y;
        }
        v[q] = x;
    }
    lds_barrier();
}

__device__ __forceinline__ bool absres_improves(double rr, double best, double maxdiff) { return sqrt(rr) < best - maxdiff; }

// (LEAD: block 0's instantiation.  The same text, but its registers never hold a matrix: what the early verdict needs
// extra does not add to the product loop's register pressure -- as one body it spilled there, 2.66 -> 3.5 us per product)
template <int NE, bool LEAD>
__device__ __forceinline__ void spcg_persist_body(const SpcgPersistArgs& a)
{
    constexpr int NT = 512, NW = 8, E = 12, CH = 6;  // m <= E * NT = 6144; two waves per SIMD: 256 registers per lane
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    __shared__ double sh[NW * 4];
    __shared__ int    s_flag;
    typedef __attribute__((address_space(1))) unsigned           gu32;
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    const int tid = threadIdx.x, lane = tid & 63, m = a.m;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr bool lead = LEAD;
    double* sp = dyn;        // p
    double* sr = dyn + m;    // r
    double* st = dyn + 2 * m;  // t of this iteration, read back from the published copy
    double* su = dyn + 3 * m;  // block 0, a.u_lds: the iterate itself (written back once, at the end)
    gu32* g_err = (gu32*)(a.sync + 3);
    gu64* g_verdict = (gu64*)(a.sync + 8);   // [parity]: {stop, epoch} of block 0
    __shared__ int s_hang;
    typedef unsigned int pu32x4 __attribute__((ext_vector_type(4)));
    SpcgState& S = *a.st;
    if (S.stop != SPCG_RUN) return;   // (written before this launch: every block reads the same value)

    // ---- the wave's slice of the matrix -> registers ----
    double   v[NE];
    unsigned cpk[NE / 2];
    int      my_rows = 0, endsv = 0, rowsv = -1;   // lane j < 8 holds end slot / row id of the wave's j-th row
    if (!lead) {
        const int gw = ((int)blockIdx.x - 1) * NW + wave;
        const double* pv = a.vals + ((size_t)gw * NE) * 64 + lane;
        const unsigned short* pc = a.cols + ((size_t)gw * NE) * 64 + lane;
#pragma unroll
        for (int k = 0; k < NE; ++k) v[k] = __builtin_nontemporal_load(pv + (size_t)k * 64);
#pragma unroll
        for (int k = 0; k < NE / 2; ++k)
            cpk[k] = (unsigned)__builtin_nontemporal_load(pc + (size_t)(2 * k) * 64) |
                     ((unsigned)__builtin_nontemporal_load(pc + (size_t)(2 * k + 1) * 64) << 16);
        if (lane < 8) { endsv = a.wend[gw * 8 + lane]; rowsv = a.wrow[gw * 8 + lane]; }
        my_rows = __popcll(__ballot(lane < 8 && rowsv >= 0));
    }
    for (int i = tid; i < m; i += NT) { sp[i] = a.p[i]; sr[i] = a.r[i]; }
    const bool u_lds = a.u_lds != 0;
    if (lead && u_lds) for (int i = tid; i < m; i += NT) su[i] = a.u[i];
    double temp1 = S.temp1;
    const double absres_best0 = S.absres_best, normr0 = S.normr0, tol = S.tol, maxdiff = S.maxdiff;
    const int    it0 = S.iter, stag = S.stag, MaxIt = S.MaxIt;
    double best = absres_best0;
    int    iter_best = S.iter_best;
    // scalars of the latest finished step (block 0, thread 0 writes them out at the end)
    double o_tp = S.tp, o_rr = S.rr, o_uu = S.uu, o_pp = S.pp, o_maxu = S.maxu, o_nan = S.nan, o_alpha = S.alpha,
           o_absres = S.absres, o_relres = S.relres, o_temp1_prev = S.temp1_prev;
    int steps_done = 0, my_stop = SPCG_RUN;
    if (tid == 0) {
        s_hang = 0;
        if (lead)   // nothing to stop before the first iteration
            __hip_atomic_store(g_verdict, ((unsigned long long)(a.epoch0 + 1u) << 32) | (unsigned)SPCG_RUN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
#ifdef SPCG_PERSIST_STAMPS
    unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = __builtin_amdgcn_s_memrealtime();
#define STAMP(q) do { const unsigned long long tn_ = __builtin_amdgcn_s_memrealtime(); tk[q] += tn_ - tl; tl = tn_; } while (0)
#else
#define STAMP(q) do { } while (0)
#endif

    for (int step = 0; step < a.max_steps; ++step) {
        const int par = step & 1;
        const unsigned ep = a.epoch0 + (unsigned)step + 1u;
        const __amdgpu_buffer_rsrc_t tr =
            __builtin_amdgcn_make_buffer_rsrc(a.t2 + (size_t)par * 2 * m, 0, (int)((unsigned)m * 16u), 0x00020000);
        // ---- t = A p on the rows of this wave, p from LDS, matrix from registers ----
        // (2.4 us of the iteration's 9.  Round 6, stamps of block 1 on one box each: the entries of a row dealt to the lanes so that the
        // 32 addresses of a half-wavefront sit on 32 different bank pairs: 2.40 -> 2.35; two or four interleaved lane sums instead of
        // the one chain of NE multiply-adds: 2.36 / 2.33 -- neither the LDS banks nor the dependent chain is what the product waits for;
        // both removed again.)
        if (!lead && my_rows > 0 && my_stop == SPCG_RUN) {
            double acc = 0.0;
            int    j = 0;
            int    next_end = __builtin_amdgcn_readlane(endsv, 0);
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const unsigned off = (k & 1) ? (cpk[k >> 1] >> 16) : (cpk[k >> 1] & 0xffffu);
                const double x = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(sp) + off);
                acc += v[k] * x;
                if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // eight gathers in flight, not NE of them: the registers hold the matrix
                if (k + 1 == next_end) {   // wave-uniform: the row is complete
                    const double ssum = wave_sum_to_lane63(acc);
                    const int row = __builtin_amdgcn_readlane(rowsv, j);
                    if (lane == 63) {   // one write-through 16-byte store: {lo, epoch, hi, epoch}
                        const unsigned long long bits = (unsigned long long)__double_as_longlong(ssum);
                        const pu32x4 rec = {(unsigned)bits, ep, (unsigned)(bits >> 32), ep};
                        __builtin_amdgcn_raw_buffer_store_b128(rec, tr, row * 16, 0, 16 /* sc1 */);
                    }
                    acc = 0.0;
                    ++j;
                    next_end = (j < 8) ? __builtin_amdgcn_readlane(endsv, j & 7) : -1;
                    if (j >= my_rows) next_end = -1;
                }
            }
        }
        // ---- the exchange: poll the m records of this iteration (sc1 loads: never the L1) and block 0's verdict ----
        // (t, p and r live in LDS and are re-read pass by pass: the registers belong to the matrix)
        STAMP(0);
        {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            bool hung = false;
            // (block 0's verdict is asked for FIRST and looked at last: its round trip runs beside those of the records instead of
            // behind them -- 0.5 of the exchange's 4 us were this one dependent load)
            unsigned long long vr0 = 0ull;
            if (tid == 0) vr0 = __hip_atomic_load(g_verdict + par, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll 1
            for (int e0 = 0; e0 < E && !hung; e0 += SPCG_POLL_GROUP) {   // six records in flight per thread (more spill: the registers hold the matrix)
#ifdef SPCG_PERSIST_STAMPS
                if (e0 == 6 && !LEAD) STAMP(6);   // (lab: slot 6 = the first group of records, slot 0 continues with the second)
#endif
                unsigned pend = 0u;
#pragma unroll
                for (int e = 0; e < SPCG_POLL_GROUP; ++e) if (tid + (e0 + e) * NT < m) pend |= 1u << e;
                while (pend) {
#ifdef SPCG_PERSIST_STAMPS
                    tk[7] += 1;   // poll rounds (two per step if every record is there at the first look)
#endif
                    pu32x4 q[SPCG_POLL_GROUP];
#pragma unroll
                    for (int e = 0; e < SPCG_POLL_GROUP; ++e)
                        q[e] = __builtin_amdgcn_raw_buffer_load_b128(tr, ((pend >> e) & 1u) ? (tid + (e0 + e) * NT) * 16 : (int)0xfffffff0u, 0, 16 /* sc1 */);
#pragma unroll
                    for (int e = 0; e < SPCG_POLL_GROUP; ++e)
                        if (((pend >> e) & 1u) && q[e].y == ep && q[e].w == ep) {
                            st[tid + (e0 + e) * NT] = __longlong_as_double((long long)(((unsigned long long)q[e].z << 32) | q[e].x));
                            pend &= ~(1u << e);
                        }
                    if (pend && __builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { hung = true; break; }   // 2 s at 100 MHz: a block is not resident
                }
            }
            if (tid == 0 && !hung) {
                int flag = -1;
                for (bool first = true;; first = false) {
                    const unsigned long long vr = first ? vr0 : __hip_atomic_load(g_verdict + par, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(vr >> 32) == ep) { flag = (int)(unsigned)vr; break; }
                    __builtin_amdgcn_s_sleep(1);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { hung = true; break; }
                }
                s_flag = flag;
            }
            if (hung) {
                s_hang = 1;
                __hip_atomic_store(g_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        lds_barrier();
        STAMP(1);
        if (s_hang) { my_stop = SPCG_HANG; break; }
        const int flag = s_flag;
        if (flag != SPCG_RUN) break;                  // the previous iteration raised a stop: uniform over the grid
        if (my_stop != SPCG_RUN) break;
        STAMP(2);
        // the vector passes run in chunks of 8 entries per thread: 8 LDS reads in flight, few registers (they hold the matrix)
        double v1[1] = {0.0};
#pragma unroll 1
        for (int e0 = 0; e0 < E; e0 += CH) {
            double tv[CH], pv[CH];
#pragma unroll
            for (int e = 0; e < CH; ++e) { const int i = min(tid + (e0 + e) * NT, m - 1); tv[e] = st[i]; pv[e] = sp[i]; }
#pragma unroll
            for (int e = 0; e < CH; ++e) if (tid + (e0 + e) * NT < m) v1[0] += tv[e] * pv[e];
        }
        blk_reduce_dpp<1, NW>(v1, sh);
        const double tp = v1[0];
        if (!(fabs(tp) > 1e-40)) {   // KrySPcg.c:172-177: breakdown, nothing is updated; every block takes this exit
            o_tp = tp; steps_done = step + 1; my_stop = SPCG_DIV0;
            break;
        }
        const double alpha = temp1 / tp;
        // Block 0's verdict is what every other block waits for at the NEXT exchange, so block 0 forms it as early as it can:
        // the iterate and its norms ride with the residual pass (u += alpha p needs nothing the p update produces), one
        // three-quantity reduction, the scalar tail, the verdict out -- and only then the new direction.  With the norms in
        // the third pass (round 2) the verdict left 8.1 us after the exchange and the others, done with their products after
        // 5.9, waited for it: 10.4 us per iteration (stamps in coarse_cg.hip.h).
        double red[3] = {0.0, 0.0, 0.0};   // (r,r) everywhere; block 0: + (u,u), (p,p)
#pragma unroll 1
        for (int e0 = 0; e0 < E; e0 += CH) {
            double tv[CH], rv[CH], pv[CH], uv[CH];
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                const int i = min(tid + (e0 + e) * NT, m - 1);
                tv[e] = st[i]; rv[e] = sr[i];
                pv[e] = lead ? sp[i] : 0.0;
                uv[e] = lead ? (u_lds ? su[i] : a.u[i]) : 0.0;
            }
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                const int i = tid + (e0 + e) * NT;
                if (i < m) {
                    const double rn = rv[e] - alpha * tv[e];
                    sr[i] = rn;   // own entries of this thread in every pass
                    red[0] += rn * rn;
                    if (lead) {
                        const double ui = uv[e] + alpha * pv[e];
                        if (u_lds) su[i] = ui; else a.u[i] = ui;
                        red[1] += ui * ui; red[2] += pv[e] * pv[e];
                    }
                }
            }
        }
        double maxu = 1.0, nan_u = 0.0;
        if (lead) {
            blk_reduce_dpp<3, NW>(red, sh);
            // (u has a NaN exactly when (u,u) is one -- squares do not cancel; max |u_i| <= 1e-20 is impossible when
            // (u,u) > 2 m 1e-40, and only below that is the maximum itself formed: Check I, KrySPcg.c:195)
            nan_u = (red[1] != red[1]) ? 1.0 : 0.0;
            if (!(red[1] > 2.0 * m * 1e-40)) {
                double mx[1] = {0.0};
                for (int i = tid; i < m; i += NT) mx[0] = fmax(mx[0], fabs(u_lds ? su[i] : a.u[i]));
                blk_reduce_dpp<1, NW>(mx, sh, 1u);
                maxu = mx[0];
            }
        } else {
            double r1[1] = {red[0]};
            blk_reduce_dpp<1, NW>(r1, sh);
            red[0] = r1[0];
        }
        const double rr = red[0], beta = rr / temp1;
        STAMP(4);
        const bool keep_best = lead && absres_improves(rr, best, maxdiff);   // KrySPcg.c:189-193, decided from rr alone
        if (lead) {
            const int it = it0 + step + 1;
            STAMP(6);
            // the scalar tail: three square roots side by side in lanes 0-2, then the two quotients (the same operations on the
            // same operands as one after the other)
            const double sq = sqrt(lane == 0 ? rr : lane == 1 ? red[1] : red[2]);
            auto bc = [&](double x, int from) -> double {
                const unsigned long long bb = (unsigned long long)__double_as_longlong(x);
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bb, from);
                const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bb >> 32), from);
                return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
            };
            const double absres = bc(sq, 0), normu = bc(sq, 1), sq_pp = bc(sq, 2);
            const double fac = fabs(alpha) * sq_pp;
            const double qd = (lane == 0 ? absres : fac) / (lane == 0 ? normr0 : normu);
            const double relres = bc(qd, 0), reldiff = bc(qd, 1);
            int stop = SPCG_RUN;
            if (nan_u > 0.0) stop = SPCG_NAN;
            else {
                if (absres < best - maxdiff) { best = absres; iter_best = it; }   // (the copy into u_best goes out with the next pass)
                if (maxu <= 1e-20) stop = SPCG_SOLSTAG;                          // Check I
                else if ((stag <= 20) & (reldiff < maxdiff)) stop = SPCG_STAG;   // Check II: host recomputes r
                else if (relres < tol) stop = SPCG_CONV;                         // Check III: host checks the true residual
                if (stop == SPCG_RUN && it >= MaxIt) stop = SPCG_MAXIT;
            }
            o_tp = tp; o_rr = rr; o_uu = red[1]; o_pp = red[2]; o_maxu = maxu; o_nan = nan_u;
            o_alpha = alpha; o_absres = absres; o_relres = relres; o_temp1_prev = temp1;
            my_stop = stop;
            if (tid == 0)   // the verdict every block polls with the products of the next iteration
                __hip_atomic_store(g_verdict + (par ^ 1), ((unsigned long long)(ep + 1u) << 32) | (unsigned)stop, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        // the new direction (block 0: and the best iterate, if this one is it)
#pragma unroll 1
        for (int e0 = 0; e0 < E; e0 += CH) {
            double rv[CH], pv[CH];
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                const int i = min(tid + (e0 + e) * NT, m - 1);
                rv[e] = sr[i]; pv[e] = sp[i];
            }
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                const int i = tid + (e0 + e) * NT;
                if (i < m) {
                    sp[i] = 1.0 * rv[e] + beta * pv[e];
                    if (keep_best) a.u_best[i] = u_lds ? su[i] : a.u[i];
                }
            }
        }
        STAMP(5);
        temp1 = rr;
        steps_done = step + 1;
        lds_barrier();   // p, r complete in LDS before the next product gathers from them
        STAMP(3);
    }
#ifdef SPCG_PERSIST_STAMPS
    if (tid == 0 && blockIdx.x < 2)
        for (int q = 0; q < 8; ++q) a.sync[32 + blockIdx.x * 8 + q] = (unsigned)tk[q];
#endif
#undef STAMP

    if (lead) {
        // r and p as the last finished step left them (a breakdown updates nothing: the LDS images are the old ones)
        for (int i = tid; i < m; i += NT) { a.r[i] = sr[i]; a.p[i] = sp[i]; if (u_lds) a.u[i] = su[i]; }
        if (tid == 0) {
            S.tp = o_tp; S.rr = o_rr; S.uu = o_uu; S.pp = o_pp; S.maxu = o_maxu; S.nan = o_nan;
            S.alpha = o_alpha; S.absres = o_absres; S.relres = o_relres;
            S.absres_best = best; S.iter_best = iter_best; S.iter = it0 + steps_done;
            S.temp1_prev = o_temp1_prev;
            S.temp1 = (my_stop == SPCG_DIV0) ? S.temp1 : temp1;
            S.stop = my_stop;
        }
    }
}

template <int NE>
__global__ __launch_bounds__(512) void k_spcg_persist(SpcgPersistArgs a)
{
    if (blockIdx.x == 0) spcg_persist_body<NE, true>(a);
    else spcg_persist_body<NE, false>(a);
}

}  // namespace fasp
