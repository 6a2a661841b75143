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

#include "kernels2.hip.h"

namespace fasp {

constexpr int ES_CAP = 512;    // entries per chunk (LDS per wave: 4 KB of values + 1 KB of columns + 1 KB of row pointers / bases)
constexpr int ES_IAW = 127;    // rows of a chunk whose pointers are staged (row 128 onwards -- rows of < 4 entries -- reads them from memory)

template <int OP>
__device__ __forceinline__ void es_epilogue(const CsrArgs& a, int r, double s)
{
    if (OP == OP_MXV) { a.y[r] = s; zx_store(a, r, s); }
    else if (OP == OP_RESID) a.y[r] = a.b[r] - s;
    else if (OP == OP_ADD) a.y[r] += s;
    else if (OP == OP_SUB) a.y[r] -= s;
    else if (OP == OP_AXPY) a.y[r] += s * a.alpha;
    else if (OP == OP_JACOBI) {
        const double d = a.diag[r], xi = a.x[r];
        const double tt = a.b[r] - s;
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
    const int      np  = a.es_np[w0];
    const unsigned old = __hip_atomic_fetch_add((gu32*)(a.es_cnt + w0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == (unsigned)np) {
        double s = __longlong_as_double((long long)__hip_atomic_load((gu64*)(a.es_part + 2 * w0 + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        for (int q = 1; q < np; ++q)
            s += __longlong_as_double((long long)__hip_atomic_load((gu64*)(a.es_part + 2 * (w0 + q)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        __hip_atomic_store((gu32*)(a.es_cnt + w0), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        es_epilogue<OP>(a, r, s);
    }
}

template <int L, int OP, int NT>
__global__ __launch_bounds__(BLOCK) void k_csr_estream(CsrArgs a)
{
    if (a.stop && *a.stop) return;
    constexpr int CAP = ES_CAP, G = 64 / L, NV = CAP / 128, NJ = CAP / 512;
    static_assert(CAP % 512 == 0 && (L & (L - 1)) == 0 && L >= 2 && L <= 64, "chunk = whole 16-byte pieces of 16-bit columns per lane");
    __shared__ __attribute__((aligned(16))) double         sv_all[4 * CAP];
    __shared__ __attribute__((aligned(16))) unsigned short sj_all[4 * CAP];
    __shared__ int sia_all[4 * 128];
    __shared__ int sjb_all[4 * 128];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double*         sv  = sv_all + wave * CAP;
    unsigned short* sj  = sj_all + wave * CAP;
    int*            sia = sia_all + wave * 128;
    int*            sjb = sjb_all + wave * 128;
    const int g = lane / L, sl = lane & (L - 1);
    // this wave's range: blocks b and b + 8 share an XCD -- an XCD's waves take one contiguous eighth of the entries
    const int W = 4 * (int)gridDim.x;
    const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x & 7) * (W >> 3) + (int)(blockIdx.x >> 3) * 4 + wave);
    const int c0 = a.es_wc[w], cend = a.es_wc[w + 1];
    if (c0 >= cend) return;   // (no workgroup barrier anywhere below)
    const int e0w = a.es_centry[c0], e1w = a.es_centry[cend];
    const bool rel = a.jbase != nullptr;

    f64x2_t qv[NV];
    u32x4_t qj[NJ];
    int     qi0 = 0, qi1 = 0, qb0 = 0, qb1 = 0;
    auto stage_load = [&](int cc) {
        const int lo = a.es_centry[cc], hi = a.es_centry[cc + 1], rf = a.es_crow[cc];
        const int n = hi - lo;
        // buffer loads, range = the chunk rounded up to whole 16-byte pieces: lanes beyond it are not fetched (the arrays carry 16 bytes of slack)
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.val + lo), 0, ((n + 1) & ~1) * 8, 0x00020000);
        const __amdgpu_buffer_rsrc_t rj = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.ja16 + lo), 0, ((n + 7) & ~7) * 2, 0x00020000);
#pragma unroll
        for (int q = 0; q < NV; ++q)
            qv[q] = __builtin_bit_cast(f64x2_t, __builtin_amdgcn_raw_buffer_load_b128(rv, (lane + 64 * q) * 16, 0, NT ? 2 : 0));
#pragma unroll
        for (int q = 0; q < NJ; ++q)
            qj[q] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rj, (lane + 64 * q) * 16, 0, NT ? 2 : 0));
        qi0 = a.ia[min(rf + lane, a.nrow)];
        qi1 = a.ia[min(rf + 64 + lane, a.nrow)];
        if (rel) { qb0 = a.jbase[min(rf + lane, a.nrow - 1)]; qb1 = a.jbase[min(rf + 64 + lane, a.nrow - 1)]; }
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int q = 0; q < NV; ++q) reinterpret_cast<f64x2_t*>(sv)[lane + 64 * q] = qv[q];
#pragma unroll
        for (int q = 0; q < NJ; ++q) reinterpret_cast<u32x4_t*>(sj)[lane + 64 * q] = qj[q];
        sia[lane] = qi0; sia[64 + lane] = qi1;
        if (rel) { sjb[lane] = qb0; sjb[64 + lane] = qb1; }
    };

    double acc = 0.0;
    int    pend_r = -1, pend_kb = 0;   // the row this lane's group leaves unfinished at the end of the wave's range
    stage_load(c0);
    stage_store();
    wave_order();
    for (int c = c0; c < cend; ++c) {
        if (c + 1 < cend) stage_load(c + 1);
        const int lo = a.es_centry[c], hi = a.es_centry[c + 1], rf = a.es_crow[c], rl = a.es_crow[c + 1];
        for (int rb = rf & ~(G - 1); rb <= rl; rb += G) {
            const int  r   = rb + g;
            const bool act = r >= rf && r <= rl;
            const int  wi  = r - rf;
            int kb = 0, ke = 0, jb = 0;
            if (act) {
                if (wi < ES_IAW) { kb = sia[wi]; ke = sia[wi + 1]; if (rel) jb = sjb[wi]; }
                else { kb = a.ia[r]; ke = a.ia[r + 1]; if (rel) jb = a.jbase[r]; }
            }
            const int kq = min(ke, hi);
            int k = max(kb, lo) + sl;
            while (__any(k < kq)) {
                int    cu[4];
                double vu[4], xu[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int  kk = k + u * L;
                    const bool in = kk < kq;
                    const int  ix = in ? kk - lo : 0;
                    const int  cj = (int)sj[ix];
                    const double vv = sv[ix];
                    cu[u] = in ? jb + cj : 0;
                    vu[u] = in ? vv : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) xu[u] = a.x[cu[u]];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (OP == OP_JACOBI) { if (cu[u] != r) acc += vu[u] * xu[u]; }
                    else acc += vu[u] * xu[u];
                }
                k += 4 * L;
            }
            const bool   fin = act && ke <= hi;
            const double tot = subwave_sum<L>(acc);
            if (fin) {
                if (sl == 0) {
                    if (kb >= e0w) es_epilogue<OP>(a, r, tot);          // the whole row lies in this wave's range
                    else es_arrive<OP>(a, 2 * w, a.es_hw0[w], r, tot);   // it began in an earlier wave: this is its last part
                }
                acc = 0.0;
            } else if (act && c + 1 == cend) { pend_r = r; pend_kb = kb; }
        }
        wave_order();
        if (c + 1 < cend) stage_store();
        wave_order();
    }
    // the row the range ends in the middle of: its first part (tail of this wave) or a middle one (the whole range lies inside one row)
    const double tot = subwave_sum<L>(acc);
    if (sl == 0 && pend_r >= 0 && pend_kb < e1w) {
        if (pend_kb >= e0w) es_arrive<OP>(a, 2 * w + 1, w, pend_r, tot);
        else es_arrive<OP>(a, 2 * w, a.es_hw0[w], pend_r, tot);
    }
}

}  // namespace fasp
