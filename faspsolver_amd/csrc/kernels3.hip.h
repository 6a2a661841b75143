// kernels3.hip.h -- entry-parallel stream kernel of the long-row levels (gfx950 / CDNA4, wave64), round 6.
//
//   y = A x family (fasp_blas_dcsr_mxv, BlaSpmvCSR.c:242; fasp_blas_dcsr_aAxpy, :494; the residual of PreMGCycle.c:136) and the
//   weighted-Jacobi / L1 sweeps (ItrSmootherCSR.c:98, :1509) on operators with rows of 48 ... thousands of entries -- levels 3-8 of
//   P7(256): 15-20 M entries each in 10 K - 310 K rows.
//
// Why not k_csr_rows (kernels.hip.h) there: a wave of that kernel walks ITS rows -- 8-byte value loads and 2-byte index loads per lane
// straight from memory, in a chain IA -> (JA, val) -> x, and the unit of work is a row: with 8 192 waves and 10-16 K rows of
// 500-2 500 entries the slowest wave carries twice the mean (profiles/r06_pmc_bound.txt).  Here the MATRIX STREAM is decoupled from the
// row structure:
//   * the entries [0, nnz) are cut into W equal wave ranges (W = 4 x blocks; an XCD's waves take one contiguous eighth), a range into
//     chunks of at most CAP entries; tables built once at upload say where a chunk starts and which row that entry lies in;
//   * a chunk travels as 16-byte loads (values: 16 B x CAP/128 per lane, 16-bit columns: 16 B x CAP/512) one chunk AHEAD of the
//     arithmetic -- registers are the second buffer, the wave's LDS slab the first -- together with the row pointers (and column bases)
//     of the chunk's rows; nothing in the loop waits for a store or drains the vector-memory counter;
//   * the rows of a chunk are then worked through from LDS by sub-wavefronts of L lanes (64 / L rows at a time, row r by group r mod
//     64/L: a row that continues in the next chunk finds its partial sums in the same lanes), lane = entry, columns sorted: the
//     gathers of neighbouring lanes fall into neighbouring lines;
//   * a row cut by a wave boundary is summed from its parts IN WAVE ORDER by whichever of its waves arrives last (a write-through
//     part per wave, one returning atomic per cut -- no wave ever waits for another, no second launch): the association of every row
//     sum is fixed by the tables, i.e. results are deterministic run to run.
// The row sums are lane-strided partial sums + a shuffle tree, like k_csr_rows' (these operators' device copies are sorted by column:
// neither kernel follows the reference's storage order; agreement with the reference 1e-13 per cycle, tests/test_gpu_estream.py).
#pragma once

#include <type_traits>

#include "kernels2.hip.h"

namespace fasp {

constexpr int ES_CAP = 512;    // entries per chunk (LDS per wave: 4 KB of values / products + 1 KB of columns + 512 B of row pointers)
constexpr int ES_IAW = 127;    // rows of a chunk whose pointers are staged (row 128 onwards -- rows of < 4 entries -- reads them from memory)

// table reads on the scalar unit (constant address space: s_load, counted in lgkmcnt -- a vector load of a uniform word would sit in the
// in-order vector-memory counter between the gathers and the stream)
typedef const __attribute__((address_space(4))) int* es_int_cp;
__device__ __forceinline__ int es_tab(const int* p, int i) { return ((es_int_cp)(unsigned long long)p)[i]; }

template <int OP>
__device__ __forceinline__ void es_epilogue(const CsrArgs& a, int r, double s)
{
    if (OP == OP_MXV) { a.y[r] = s; zx_store(a, r, s); }
    else if (OP == OP_RESID) a.y[r] = a.b[r] - s;
    else if (OP == OP_ADD) a.y[r] += s;
    else if (OP == OP_SUB) a.y[r] -= s;
    else if (OP == OP_AXPY) a.y[r] += s * a.alpha;
    else if (OP == OP_JACOBI) {   // s holds EVERY entry of the row: the diagonal's product goes out here (see k_csr_estream)
        const double d = a.diag[r], xi = a.x[r];
        const double tt = a.b[r] - (s - d * xi);
        a.y[r] = (fabs(d) > 1e-20) ? (1 - a.omega) * xi + a.omega * tt / d : xi;
    } else if (OP == OP_L1DIAG) {
        const double d = a.diag[r], xi = a.x[r];
        const double tt = a.b[r] - s;
        a.y[r] = l1_or_jacobi_f(a, r, tt, d, xi);
    }
}

// One part of a row that a wave boundary cuts: the part goes to memory (write-through), then the wave counts itself in at the row's
// FIRST wave w0; the wave whose count completes the row (es_np[w0] parts: the tail of w0, then the heads of w0 + 1, ...) sums them in
// that order, whoever it is, applies the epilogue and puts the counter back to zero for the next launch.
template <int OP>
__device__ __forceinline__ void es_arrive(const CsrArgs& a, int slot, int w0, int r, double part)
{
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    typedef __attribute__((address_space(1))) unsigned           gu32;
    __hip_atomic_store((gu64*)(a.es_part + slot), (unsigned long long)__double_as_longlong(part), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the part has left this CU before the count says so
    const int      np  = es_tab(a.es_np, w0);
    const unsigned old = __hip_atomic_fetch_add((gu32*)(a.es_cnt + w0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == (unsigned)np) {
        double s = __longlong_as_double((long long)__hip_atomic_load((gu64*)(a.es_part + 2 * w0 + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        for (int q = 1; q < np; ++q)
            s += __longlong_as_double((long long)__hip_atomic_load((gu64*)(a.es_part + 2 * (w0 + q)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        __hip_atomic_store((gu32*)(a.es_cnt + w0), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        es_epilogue<OP>(a, r, s);
    }
}

// sum over the L lanes of every group of a wavefront on data-parallel-primitive moves (row_shr 1, 2, 4, 8 inside rows of 16, row_bcast 15 / 31
// across them); the group's total ends in its LAST lane
template <int L>
__device__ __forceinline__ double es_group_sum(double x)
{
    auto mv = [](double v, auto ctrl_tag) -> double {
        constexpr int ctrl = decltype(ctrl_tag)::value;
        constexpr int rmask = ctrl == 0x142 ? 0xa : ctrl == 0x143 ? 0xc : 0xf;
        const long long b = __double_as_longlong(v);
        int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
        lo = __builtin_amdgcn_update_dpp(0, lo, ctrl, rmask, 0xf, true);
        hi = __builtin_amdgcn_update_dpp(0, hi, ctrl, rmask, 0xf, true);
        return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
    };
    if (L >= 2) x += mv(x, std::integral_constant<int, 0x111>{});
    if (L >= 4) x += mv(x, std::integral_constant<int, 0x112>{});
    if (L >= 8) x += mv(x, std::integral_constant<int, 0x114>{});
    if (L >= 16) x += mv(x, std::integral_constant<int, 0x118>{});
    if (L >= 32) x += mv(x, std::integral_constant<int, 0x142>{});
    if (L >= 64) x += mv(x, std::integral_constant<int, 0x143>{});
    return x;
}
__device__ __forceinline__ double es_shfl(double v, int src)   // lane src's value (per-lane src: ds_bpermute)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_bpermute(src << 2, (int)(b & 0xffffffffll)), hi = __builtin_amdgcn_ds_bpermute(src << 2, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}

// Two phases per chunk, both fed from the wave's LDS slab (the k_csr_wstream2 scheme, kernels2.hip.h, with sub-wavefront row sums):
//   (1) lane = ENTRY over the whole chunk, whatever the rows: columns and values from the slab, eight gathers of x in flight, products
//       back into the slab in place.  The operands a finished row's epilogue needs (b, a_ii, x_r / the old y_r of the chunk's rows, lane =
//       row) and then the NEXT chunk's 16-byte stream loads and row pointers are issued right behind the gathers: waiting for the
//       gathers leaves the stream in flight (the vector-memory counter retires in order -- a stream load issued in FRONT of a gather
//       would be waited for with it), and nothing younger than the stream is waited for until the products are summed.  Straight-line
//       code between the gathers and the products: at a control-flow join the compiler's wait-count pass assumes the worse of both
//       paths, which for a prefetch under an `if` means waiting for it (profiles/r06_estream.txt);
//   (2) the chunk's rows, 64 / L at a time, L lanes per row (row r by group r mod 64/L), sum their products from the slab -- no memory
//       access -- lane-strided partial sums that a row cut by the chunk's end carries into the next chunk in the same lanes; a row that
//       ends here: DPP sum over the group, epilogue.  L is chosen so that the rows of a chunk normally fit ONE pass (device_csr.hip.h):
//       the kernel is bound by instruction issue, not by memory (in-kernel stamps: profiles/r06_estream.txt), and a pass has a fixed cost.
//       A lane reads its j-th product at a constant offset from a base address; products it does not have come from a slot that holds 0.0.
// Weighted Jacobi: the reference leaves the diagonal entry out of the row sum (ItrSmootherCSR.c:130); lane = entry does not know the
// entry's row, so the sum takes every entry and the epilogue subtracts a_ii x_i -- the last stored diagonal, as the reference's d
// (operators that store a diagonal twice keep the row kernel: launch_csr) -- before it forms b_i - sum.
#ifdef ES_TIMING
#define EST(k) do { const long long n_ = (long long)__builtin_readcyclecounter(); est[k] += n_ - est_last; est_last = n_; } while (0)
#else
#define EST(k)
#endif
template <int L, int OP, int DBG = 0>   // DBG (lab builds of the y = A x form only): 1 no gathers, 2 no stream loads, 4 no row sums
__global__ __launch_bounds__(BLOCK) void k_csr_estream(CsrArgs a)
{
    if (a.stop && *a.stop) return;
    constexpr int CAP = ES_CAP, G = 64 / L, NV = CAP / 128, NJ = CAP / 512, NU = CAP / 64;
    static_assert(CAP % 512 == 0 && (L & (L - 1)) == 0 && L >= 2 && L <= 64, "chunk = whole 16-byte pieces of 16-bit columns per lane");
    constexpr bool NEED_B = OP == OP_RESID || OP == OP_JACOBI || OP == OP_L1DIAG;
    constexpr bool NEED_D = OP == OP_JACOBI || OP == OP_L1DIAG;
    constexpr bool NEED_X = OP == OP_JACOBI || OP == OP_L1DIAG || OP == OP_ADD || OP == OP_SUB || OP == OP_AXPY;   // x_r, or the old y_r
    // per wave: CAP + 2 doubles (products; slot CAP holds 0.0), CAP 16-bit columns, 128 row pointers.  Dynamic: the launch asks for
    // ES_LDS_BYTES so that exactly ES_BPC workgroups fit a CU -- the grid is ES_BPC x CUs workgroups, every CU gets the same number
    extern __shared__ __attribute__((aligned(16))) unsigned char es_lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double*         sv  = reinterpret_cast<double*>(es_lds) + wave * (CAP + 2);
    unsigned short* sj  = reinterpret_cast<unsigned short*>(es_lds + 4 * (CAP + 2) * 8) + wave * CAP;
    int*            sia = reinterpret_cast<int*>(es_lds + 4 * (CAP + 2) * 8 + 4 * CAP * 2) + wave * 128;
    const int g = lane / L, sl = lane & (L - 1);
    // this wave's range: blocks b and b + 8 share an XCD -- an XCD's waves take one contiguous eighth of the entries
    const int W = 4 * (int)gridDim.x;
    const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x & 7) * (W >> 3) + (int)(blockIdx.x >> 3) * 4 + wave);
    const int c0 = es_tab(a.es_wc, w), cend = es_tab(a.es_wc, w + 1);
    if (c0 >= cend) return;   // (no workgroup barrier anywhere below)
    const int e0w = es_tab(a.es_centry, c0), e1w = es_tab(a.es_centry, cend);
    const unsigned short* const cols = a.es_ja16 ? a.es_ja16 : a.ja16;   // 16-bit columns relative to the CHUNK's smallest (operators with > 65536 columns) or absolute

    f64x2_t qv[NV];
    u32x4_t qj[NJ];
    int     qi0 = 0, qi1 = 0;
    // the stream loads of a chunk [lo, lo + n) whose first row is rf; n = 0 (no next chunk): the buffer range is empty, nothing is fetched --
    // issued UNCONDITIONALLY (see above)
    auto stage_load = [&](int lo, int n, int rf) {
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.val + lo), 0, ((n + 1) & ~1) * 8, 0x00020000);
        const __amdgpu_buffer_rsrc_t rj = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(cols + lo), 0, ((n + 7) & ~7) * 2, 0x00020000);
#pragma unroll
        for (int q = 0; q < NV; ++q)
            qv[q] = __builtin_bit_cast(f64x2_t, __builtin_amdgcn_raw_buffer_load_b128(rv, (lane + 64 * q) * 16, 0, 0));
#pragma unroll
        for (int q = 0; q < NJ; ++q)
            qj[q] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rj, (lane + 64 * q) * 16, 0, 0));
        const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(a.ia + rf), 0, n > 0 ? (a.nrow + 1 - rf) * 4 : 0, 0x00020000);
        qi0 = __builtin_amdgcn_raw_buffer_load_b32(ri, lane * 4, 0, 0);          // (beyond the last row pointer: 0, never used)
        qi1 = __builtin_amdgcn_raw_buffer_load_b32(ri, (64 + lane) * 4, 0, 0);
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int q = 0; q < NV; ++q) reinterpret_cast<f64x2_t*>(sv)[lane + 64 * q] = qv[q];
#pragma unroll
        for (int q = 0; q < NJ; ++q) reinterpret_cast<u32x4_t*>(sj)[lane + 64 * q] = qj[q];
        sia[lane] = qi0; sia[64 + lane] = qi1;
    };

#ifdef ES_TIMING
    long long est[8] = {0, 0, 0, 0, 0, 0, 0, 0}, est_last = (long long)__builtin_readcyclecounter();
    const long long est_t0 = est_last;
#endif
    double acc = 0.0;
    int    pend_r = -1, pend_kb = 0;   // the row this lane's group leaves unfinished at the end of the wave's range
    int lo = e0w, hi = es_tab(a.es_centry, c0 + 1), rf = es_tab(a.es_crow, c0), rl = es_tab(a.es_crow, c0 + 1);
    if (lane == 0) sv[CAP] = 0.0;
    stage_load(lo, hi - lo, rf);
    stage_store();
    wave_order();
    EST(0);
    for (int c = c0; c < cend; ++c) {
        // everything the scalar unit has to fetch for this chunk and for the stream loads of the next, up front
        const bool more = c + 1 < cend;
        const int  hi2 = es_tab(a.es_centry, more ? c + 2 : c + 1), rl2 = es_tab(a.es_crow, more ? c + 2 : c + 1);
        const int  cb = a.es_cbase ? es_tab(a.es_cbase, c) : 0;
        // ---- phase 1: lane = entry
        int    cc[NU];
        double wv[NU], xv[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) cc[u] = cb + (int)sj[lane + 64 * u];   // (beyond the chunk: 0 -> column cb, value 0.0: never summed)
#pragma unroll
        for (int u = 0; u < NU; ++u) wv[u] = sv[lane + 64 * u];
#pragma unroll
        for (int u = 0; u < NU; ++u) xv[u] = (DBG & 1) ? 1.0 : a.x[cc[u]];
        // the epilogue's operands of the chunk's rows rf + lane, rf + 64 + lane (clamped: the clamped copies are never used)
        double eb0 = 0.0, eb1 = 0.0, ed0 = 0.0, ed1 = 0.0, ex0 = 0.0, ex1 = 0.0;
        const int ra = min(rf + lane, a.nrow - 1), rc = min(rf + 64 + lane, a.nrow - 1);
        if (NEED_B) { eb0 = a.b[ra]; eb1 = a.b[rc]; }
        if (NEED_D) { ed0 = a.diag[ra]; ed1 = a.diag[rc]; }
        if (NEED_X) { const double* src = (OP == OP_JACOBI || OP == OP_L1DIAG) ? a.x : a.y; ex0 = src[ra]; ex1 = src[rc]; }
        EST(1);
        stage_load(hi, (more && !(DBG & 2)) ? hi2 - hi : 0, rl);
        EST(2);
#pragma unroll
        for (int u = 0; u < NU; ++u) sv[lane + 64 * u] = wv[u] * xv[u];
        wave_order();
        EST(3);
        // ---- phase 2: the chunk's rows from the slab.  Two copies of the pass: the normal one takes row pointers and epilogue operands from
        // what phase 1 staged and contains NO vector-memory load whose result it uses -- a load on any path into a use makes the
        // compiler wait for the whole in-order counter there, i.e. for the stream (that cost 4 000-6 000 cycles per chunk before the
        // split: profiles/r06_estream.txt); chunks that touch more rows than were staged (rows of < 4 entries) take the other copy
        auto rows_pass = [&](auto slow_tag) {
            constexpr bool SLOW = decltype(slow_tag)::value;
            for (int rb = rf & ~(G - 1); rb <= rl; rb += G) {
                const int  r   = rb + g;
                const bool act = r >= rf && r <= rl;
                const int  wi  = r - rf;
                int kb = 0, ke = 0;
                if (SLOW) { if (act) { kb = a.ia[r]; ke = a.ia[r + 1]; } }
                else { const int wc_ = act ? wi : 0; kb = sia[wc_]; ke = sia[wc_ + 1]; if (!act) { kb = 0; ke = 0; } }
                // this lane's products: entries k0, k0 + L, ... below kq: n of them, the j-th at slab index i0 + j L (none: n <= 0)
                const int k0 = max(kb, lo) + sl, kq = min(ke, hi);
                const int n  = (kq - k0 + L - 1) >> __builtin_ctz(L);   // (arithmetic shift: <= 0 when the lane has none)
                const double* const pp = sv + (k0 - lo);
                const double* const pz = sv + CAP;                       // 0.0
                EST(4);
                for (int u = 0; !(DBG & 4) && __any(u < n); u += 4) {
                    const double* q = pp + u * L;
                    const double p0 = *(u + 0 < n ? q : pz), p1 = *(u + 1 < n ? q + L : pz), p2 = *(u + 2 < n ? q + 2 * L : pz), p3 = *(u + 3 < n ? q + 3 * L : pz);
                    acc += p0; acc += p1; acc += p2; acc += p3;
                }
                EST(6);
                const bool fin = act && ke <= hi;
                if (__any(fin)) {
                    const double tot = es_group_sum<L>(acc);
                    double eb = 0.0, ed = 0.0, ex = 0.0;
                    if (!SLOW) {   // operands of this lane's row from the lane that loaded them
                        const int wl = wi & 63;
                        if (NEED_B) { const double t0 = es_shfl(eb0, wl), t1 = es_shfl(eb1, wl); eb = wi < 64 ? t0 : t1; }
                        if (NEED_D) { const double t0 = es_shfl(ed0, wl), t1 = es_shfl(ed1, wl); ed = wi < 64 ? t0 : t1; }
                        if (NEED_X) { const double t0 = es_shfl(ex0, wl), t1 = es_shfl(ex1, wl); ex = wi < 64 ? t0 : t1; }
                    }
                    if (fin) {
                        if (sl == L - 1) {
                            if (kb < e0w) es_arrive<OP>(a, 2 * w, es_tab(a.es_hw0, w), r, tot);   // it began in an earlier wave: this is its last part
                            else if (SLOW) es_epilogue<OP>(a, r, tot);
                            else {   // the whole row lies in this wave's range: es_epilogue's expressions on the operands loaded in phase 1
                                if (OP == OP_MXV) { a.y[r] = tot; zx_store(a, r, tot); }
                                else if (OP == OP_RESID) a.y[r] = eb - tot;
                                else if (OP == OP_ADD) a.y[r] = ex + tot;
                                else if (OP == OP_SUB) a.y[r] = ex - tot;
                                else if (OP == OP_AXPY) a.y[r] = ex + tot * a.alpha;
                                else if (OP == OP_JACOBI) { const double tt = eb - (tot - ed * ex); a.y[r] = (fabs(ed) > 1e-20) ? (1 - a.omega) * ex + a.omega * tt / ed : ex; }
                                else if (OP == OP_L1DIAG) { const double tt = eb - tot; a.y[r] = l1_or_jacobi_f(a, r, tt, ed, ex); }
                            }
                        }
                        acc = 0.0;
                    }
                }
                if (act && !fin && !more) { pend_r = r; pend_kb = kb; }
                EST(7);
            }
        };
        if (rl - rf < ES_IAW) rows_pass(std::false_type{});
        else rows_pass(std::true_type{});
        wave_order();
        stage_store();   // (after the last chunk: zeros from the empty range)
        wave_order();
        EST(5);
        lo = hi; hi = hi2; rf = rl; rl = rl2;
    }
#ifdef ES_TIMING
    if (lane == 0 && (w % 509) == 0) printf("[es] wave %d: %d chunks, total %lld cycles; per chunk: issue gathers %lld, issue stream %lld, wait gathers + products %lld, row setup %lld, product loop %lld, sum + epilogue %lld, stream -> LDS %lld; prologue %lld\n", w, cend - c0, (long long)__builtin_readcyclecounter() - est_t0, est[1] / (cend - c0), est[2] / (cend - c0), est[3] / (cend - c0), est[4] / (cend - c0), est[6] / (cend - c0), est[7] / (cend - c0), est[5] / (cend - c0), est[0]);
#endif
    // the row the range ends in the middle of: its first part (tail of this wave) or a middle one (the whole range lies inside one row)
    const double tot = es_group_sum<L>(acc);
    if (sl == L - 1 && pend_r >= 0 && pend_kb < e1w) {
        if (pend_kb >= e0w) es_arrive<OP>(a, 2 * w + 1, w, pend_r, tot);
        else es_arrive<OP>(a, 2 * w, es_tab(a.es_hw0, w), pend_r, tot);
    }
}
constexpr int ES_BPC = 5;                                                   // workgroups per CU
constexpr int ES_LDS_MIN = 4 * ((ES_CAP + 2) * 8 + ES_CAP * 2 + 128 * 4);   // what the kernel uses
constexpr int ES_LDS_BYTES = 160 * 1024 / ES_BPC - 4096;                    // what a launch asks for: ES_BPC fit a CU, ES_BPC + 1 do not
static_assert(ES_LDS_BYTES >= ES_LDS_MIN && (ES_BPC + 1) * ES_LDS_BYTES > 160 * 1024, "exactly ES_BPC workgroups per CU");

// ---- upload helpers: 16-bit columns relative to the smallest column of the CHUNK (operators with more than 65536 columns) -----------
// one wavefront per chunk: its smallest column; flag |= 1 when a chunk spans 65536 columns or more (then the operator keeps the row kernel)
__global__ __launch_bounds__(BLOCK) void k_es_chunk_base(int nc, const int* __restrict__ centry, const int* __restrict__ ja, int* __restrict__ cbase, int* __restrict__ flag)
{
    const int lane = threadIdx.x & 63;
    for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < nc; c += gridDim.x * 4) {
        const int lo = centry[c], hi = centry[c + 1];
        int mn = 2147483647, mx = -1;
        for (int k = lo + lane; k < hi; k += 64) { const int j = ja[k]; mn = min(mn, j); mx = max(mx, j); }
        for (int off = 32; off > 0; off >>= 1) { mn = min(mn, __shfl_xor(mn, off, 64)); mx = max(mx, __shfl_xor(mx, off, 64)); }
        if (lane == 0) { cbase[c] = hi > lo ? mn : 0; if (hi > lo && mx - mn >= 65536) atomicOr(flag, 1); }
    }
}
__global__ __launch_bounds__(BLOCK) void k_es_chunk_cols(int nc, const int* __restrict__ centry, const int* __restrict__ ja, const int* __restrict__ cbase, unsigned short* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < nc; c += gridDim.x * 4) {
        const int lo = centry[c], hi = centry[c + 1], cb = cbase[c];
        for (int k = lo + lane; k < hi; k += 64) out[k] = (unsigned short)(ja[k] - cb);
    }
}

}  // namespace fasp
