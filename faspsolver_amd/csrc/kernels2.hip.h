// kernels2.hip.h -- second-generation row kernels of the fine levels (gfx950 / CDNA4, wave64).
//
// What round 1's kernels were bound by (rocprofv3, profiles/r01_*): not HBM bytes but the number of
// vector-memory wave-instructions.  The texture-address unit of a CU takes a 64-lane instruction in
// ~16 cycles whatever the bytes per lane, and every distinct cache line a gather touches costs one
// more tag lookup; a kernel that moves 4 or 8 bytes per lane and instruction, or whose gathers hit
// 10-20 lines, runs at a fraction of what the same bytes cost as 16-byte, line-aligned accesses.
//
//   k_csr_lstream   plain CSR, short rows (y = A x family, BlaSpmvCSR.c:242 / :494, Jacobi
//                   ItrSmootherCSR.c:98): JA / val are streamed with 16-byte loads one tile AHEAD
//                   (registers are the second buffer), parked in the wave's LDS slab, and consumed
//                   lane = ROW: the k-th gathers of 64 consecutive rows of a banded matrix fall into
//                   4-8 cache lines instead of 10-20, and every row sum is the reference's
//                   left-to-right sum (bit-identical).  No workgroup barrier, no fence that drains vmcnt.
//   k_csr_rowpat2   row-pattern-coded square matrices (kernels.hip.h, k_csr_rowpat): a lane owns the
//                   TWO consecutive rows 2i, 2i+1; when they share a pattern (all interior rows do) one
//                   16-byte load fetches x for both, and b / y / the dotted vector move as 16-byte
//                   accesses too: half the memory instructions per row.  Rows whose partner has a
//                   different pattern are finished by a short divergent tail.
#pragma once

#include "kernels.hip.h"

namespace fasp {

typedef int    i32x4_t __attribute__((ext_vector_type(4)));
typedef double f64x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// compiler-only ordering of a wavefront's own LDS traffic (the hardware executes a wave's LDS
// instructions in order; a workgroup-scope fence would also wait for every outstanding global load)
__device__ __forceinline__ void wave_order()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int OP>
__device__ __forceinline__ void row_epilogue(const CsrArgs& a, int r, double s, double& dotacc)
{
    if (OP == OP_MXV) { if (a.nt & 2) __builtin_nontemporal_store(s, a.y + r); else a.y[r] = s; zx_store(a, r, s); }
    else if (OP == OP_RESID) a.y[r] = a.b[r] - s;
    else if (OP == OP_ADD) a.y[r] += s;
    else if (OP == OP_SUB) a.y[r] -= s;
    else if (OP == OP_AXPY) a.y[r] += s * a.alpha;
    else if (OP == OP_JACOBI) {
        const double d = a.diag[r], xi = a.x[r];
        const double xn = (fabs(d) > 1e-20) ? (1 - a.omega) * xi + a.omega * s / d : xi;
        a.y[r] = xn;
        if (a.partials) dotacc += xn * a.b[r];   // last sweep of level 0 under PCG: (z, r) on the way out
    } else if (OP == OP_L1DIAG) {
        const double d = a.diag[r], xi = a.x[r];
        a.y[r] = l1_or_jacobi_f(a, r, s, d, xi);
    } else if (OP == OP_MXV_DOT) {
        a.y[r] = s;
        dotacc += s * a.dotv[r];
    }
}

// ---------------------------------------------------------------------------
// k_csr_lstream<OP, CAP>: a wavefront owns 64 consecutive rows per step; CAP = entries its LDS slab holds.
// Software pipeline per wave (A = tile in the slab, B = next tile, C = the one after):
//     gathers of A issued | 16-byte JA / val loads of B issued | row pointers of C issued
//     | A's sums, epilogue | B's registers -> slab
// A tile whose span exceeds CAP (never on the matrices this kernel is selected for, but legal) is read
// straight from global memory by its rows.
// ---------------------------------------------------------------------------
template <int OP, int CAP>
__global__ __launch_bounds__(BLOCK) void k_csr_lstream(CsrArgs a)
{
    if (a.stop && *a.stop) return;
    static_assert(CAP % 256 == 0 && CAP <= 1024, "slab = 1..4 rounds of 64 x 16-byte JA pieces");
    constexpr int NJ = CAP / 256;  // 16-byte JA pieces per lane
    constexpr int NV = CAP / 128;  // 16-byte val pieces per lane
    __shared__ __attribute__((aligned(16))) double sv_all[4 * CAP];
    __shared__ __attribute__((aligned(16))) int    sj_all[4 * CAP];
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* sv = sv_all + wave * CAP;
    int*    sj = sj_all + wave * CAP;
    const int vmax = tile_vmax(a);
    const int G = gridDim.x;
    double dotacc = 0.0;

    // next tile of this wave: first row (-1: none) and row count, wave-uniform
    auto advance = [&](int& v, int& r0, int& nr) {
        for (;;) {
            r0 = -1; nr = 0;
            if (v >= vmax) return;
            const int t = tile_of(a, v);
            v += G;
            if (t >= a.ntiles) continue;
            const int rr = (t + a.tile0) * BLOCK + wave * 64;
            if (rr >= a.nrow) continue;
            r0 = rr; nr = min(64, a.nrow - rr);
            return;
        }
    };
    auto load_ia = [&](int r0, int nr, int& kb, int& ke) {
        kb = ke = 0;
        if (r0 >= 0 && lane < nr) { kb = a.ia[r0 + lane]; ke = a.ia[r0 + lane + 1]; }
    };
    // span of a tile: [s0, k1), s0 = first entry rounded down to a multiple of 4 (16-byte aligned JA piece)
    auto span = [&](int r0, int nr, int kb, int ke, int& s0, int& k1) {
        s0 = k1 = 0;
        if (r0 >= 0) {
            s0 = __builtin_amdgcn_readlane(kb, 0) & ~3;
            k1 = __builtin_amdgcn_readlane(ke, nr - 1);
        }
    };
    i32x4_t qj[NJ];
    f64x2_t qv[NV];
    auto stage_load = [&](int s0, int k1) {
        const int n = k1 - s0;
        if (n > CAP) return;
        const i32x4_t* pj = reinterpret_cast<const i32x4_t*>(a.ja + s0);
        const f64x2_t* pv = reinterpret_cast<const f64x2_t*>(a.val + s0);
#pragma unroll
        for (int q = 0; q < NJ; ++q)
            if ((lane + 64 * q) * 4 < n) qj[q] = __builtin_nontemporal_load(pj + lane + 64 * q);
#pragma unroll
        for (int q = 0; q < NV; ++q)
            if ((lane + 64 * q) * 2 < n) qv[q] = __builtin_nontemporal_load(pv + lane + 64 * q);
    };
    auto stage_store = [&](int s0, int k1) {
        const int n = k1 - s0;
        if (n > CAP) return;
#pragma unroll
        for (int q = 0; q < NJ; ++q)
            if ((lane + 64 * q) * 4 < n) reinterpret_cast<i32x4_t*>(sj)[lane + 64 * q] = qj[q];
#pragma unroll
        for (int q = 0; q < NV; ++q)
            if ((lane + 64 * q) * 2 < n) reinterpret_cast<f64x2_t*>(sv)[lane + 64 * q] = qv[q];
    };

    int v = blockIdx.x;
    int r0A, nrA, kbA, keA, s0A, k1A;
    int r0B, nrB, kbB, keB, s0B, k1B;
    advance(v, r0A, nrA);
    load_ia(r0A, nrA, kbA, keA);
    advance(v, r0B, nrB);
    load_ia(r0B, nrB, kbB, keB);
    span(r0A, nrA, kbA, keA, s0A, k1A);
    if (r0A >= 0) { stage_load(s0A, k1A); stage_store(s0A, k1A); }
    wave_order();

    while (r0A >= 0) {
        const bool fit = (k1A - s0A) <= CAP;  // wave-uniform
        const int  r = r0A + lane;
        const int  len = keA - kbA;            // 0 for lanes beyond the tile
        const int  o = kbA - s0A;
        double acc = 0.0;
        if ((OP == OP_JACOBI || OP == OP_L1DIAG) && lane < nrA) acc = a.b[r];

        // first round of gathers (rows of up to 8 entries need no second one); branch-free: lanes past
        // the end of their row read slab entry 0 and gather x[0], their products are dropped by a select
        int    c[8];
        double xv[8];
        if (fit) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int cc = sj[(u < len) ? o + u : 0];
                c[u] = (u < len) ? cc : 0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[u] = __any(u < len) ? a.x[c[u]] : 0.0;  // wave-uniform skip
        }

        // look ahead: JA / val of B, row pointers of C
        span(r0B, nrB, kbB, keB, s0B, k1B);
        if (r0B >= 0) stage_load(s0B, k1B);
        int r0C, nrC, kbC, keC;
        advance(v, r0C, nrC);
        load_ia(r0C, nrC, kbC, keC);

        if (fit) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double pr = sv[(u < len) ? o + u : 0] * xv[u];
                bool use = u < len;
                if (OP == OP_JACOBI) use = use && c[u] != r;
                const double nxt = (OP == OP_JACOBI || OP == OP_L1DIAG) ? acc - pr : acc + pr;
                acc = use ? nxt : acc;
            }
            // longer rows: further rounds of 8
            for (int j = 8; __any(j < len); j += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int cc = sj[(j + u < len) ? o + j + u : 0];
                    c[u] = (j + u < len) ? cc : 0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) xv[u] = a.x[c[u]];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double pr = sv[(j + u < len) ? o + j + u : 0] * xv[u];
                    bool use = j + u < len;
                    if (OP == OP_JACOBI) use = use && c[u] != r;
                    const double nxt = (OP == OP_JACOBI || OP == OP_L1DIAG) ? acc - pr : acc + pr;
                    acc = use ? nxt : acc;
                }
            }
        } else {  // oversized tile: every row reads its own entries from global memory
            for (int k = kbA; k < keA; ++k) {
                const int    cc = a.ja[k];
                const double pr = a.val[k] * a.x[cc];
                if (OP == OP_JACOBI) { if (cc != r) acc -= pr; }
                else if (OP == OP_L1DIAG) acc -= pr;
                else acc += pr;
            }
        }
        if (lane < nrA) row_epilogue<OP>(a, r, acc, dotacc);

        wave_order();
        if (r0B >= 0) stage_store(s0B, k1B);
        wave_order();
        r0A = r0B; nrA = nrB; kbA = kbB; keA = keB; s0A = s0B; k1A = k1B;
        r0B = r0C; nrB = nrC; kbB = kbC; keB = keC;
    }
    if (OP == OP_MXV_DOT || (OP == OP_JACOBI && a.partials)) {
        const double tot = block_sum(dotacc, red);
        if (threadIdx.x == 0) a.partials[blockIdx.x] = tot;
    }
}

// ---------------------------------------------------------------------------
// k_csr_wstream2<OP>: plain CSR with rows of any length up to ~48 on average (levels 1-2, R, P of an unstructured
// hierarchy) -- k_csr_wstream's two phases (lane = entry products into LDS, then lane = row sums in storage order: the
// reference's left-to-right row sums, bit-identical) fed like k_csr_lstream: JA / val of the NEXT 512-entry chunk are
// in flight as 16-byte loads while the current one is multiplied and summed, and nothing in the loop waits for a
// store or drains the vector-memory counter (k_csr_wstream's workgroup-scope fences did both).
// ---------------------------------------------------------------------------
template <int OP>
__global__ __launch_bounds__(BLOCK) void k_csr_wstream2(CsrArgs a)
{
    if (a.stop && *a.stop) return;
    constexpr int CAP = 512, NJ = 2, NV = 4;
    __shared__ __attribute__((aligned(16))) double sv_all[4 * CAP];   // values, then (in place) products
    __shared__ __attribute__((aligned(16))) int    sj_all[4 * CAP];
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* sv = sv_all + wave * CAP;
    int*    sj = sj_all + wave * CAP;
    const int vmax = tile_vmax(a);
    const int G = gridDim.x;
    double dotacc = 0.0;

    auto advance = [&](int& v, int& r0, int& nr) {
        for (;;) {
            r0 = -1; nr = 0;
            if (v >= vmax) return;
            const int t = tile_of(a, v);
            v += G;
            if (t >= a.ntiles) continue;
            const int rr = (t + a.tile0) * BLOCK + wave * 64;
            if (rr >= a.nrow) continue;
            r0 = rr; nr = min(64, a.nrow - rr);
            return;
        }
    };
    auto load_ia = [&](int r0, int nr, int& kb, int& ke) {
        kb = ke = 0;
        if (r0 >= 0 && lane < nr) { kb = a.ia[r0 + lane]; ke = a.ia[r0 + lane + 1]; }
    };
    i32x4_t qj[NJ];
    f64x2_t qv[NV];
    // chunk [lo, hi) staged from the 4-aligned entry s = lo & ~3: slab index of entry k is k - s.  Buffer loads with
    // the range set to the chunk (a scalar descriptor per chunk): no lane is branched around, entries at or beyond hi
    // come back as 0 (column 0, value 0.0) without being fetched, and the whole slab is written every time.
    auto stage_load = [&](int s, int hi) {
        const int n4 = (hi - s + 3) & ~3;
        const __amdgpu_buffer_rsrc_t rj = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(a.ja + s), 0, n4 * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.val + s), 0, n4 * 8, 0x00020000);
#pragma unroll
        for (int q = 0; q < NJ; ++q)
            qj[q] = __builtin_bit_cast(i32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rj, (lane + 64 * q) * 16, 0, 2));
#pragma unroll
        for (int q = 0; q < NV; ++q)
            qv[q] = __builtin_bit_cast(f64x2_t, __builtin_amdgcn_raw_buffer_load_b128(rv, (lane + 64 * q) * 16, 0, 2));
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int q = 0; q < NJ; ++q) reinterpret_cast<i32x4_t*>(sj)[lane + 64 * q] = qj[q];
#pragma unroll
        for (int q = 0; q < NV; ++q) reinterpret_cast<f64x2_t*>(sv)[lane + 64 * q] = qv[q];
    };
    // chunks of a tile: [k0, k1) cut at lo_c = k0 + c * (CAP - 4): with s = lo & ~3 every chunk fits the slab
    constexpr int STEP = CAP - 4;

    int v = blockIdx.x;
    int r0A, nrA, kbA, keA, r0B, nrB, kbB, keB;
    advance(v, r0A, nrA);
    load_ia(r0A, nrA, kbA, keA);
    advance(v, r0B, nrB);
    load_ia(r0B, nrB, kbB, keB);
    int k0 = 0, k1 = 0, lo = 0, hi = 0;
    if (r0A >= 0) {
        k0 = __builtin_amdgcn_readlane(kbA, 0); k1 = __builtin_amdgcn_readlane(keA, nrA - 1);
        lo = k0; hi = min(lo + STEP, k1);
        stage_load(lo & ~3, hi);
        stage_store();
    }
    wave_order();
    while (r0A >= 0) {
        const int r = r0A + lane;
        double acc = ((OP == OP_JACOBI || OP == OP_L1DIAG) && lane < nrA) ? a.b[r] : 0.0;
        const int dk = (OP == OP_JACOBI && lane < nrA) ? a.dpos[r] : -1;
        for (;;) {   // chunks of tile A; the slab holds [lo & ~3, hi)
            const int s = lo & ~3;
            // phase 1: lane = entry -- columns and values from the slab, x gathered, products back in place
            int    c[8];
            double w[8], xv[8];
            // (every slab entry is readable; the up to three entries between hi and the end of the last 16-byte group
            // belong to the next rows -- or, at the very end of the arrays, to nobody: their columns are not used)
            int cs[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) cs[u] = sj[lane + 64 * u];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = sv[lane + 64 * u];
#pragma unroll
            for (int u = 0; u < 8; ++u) c[u] = (s + lane + 64 * u < hi) ? cs[u] : 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[u] = a.x[c[u]];
            // look ahead: the next chunk of this tile, or the first chunk of tile B (and the row pointers of tile C)
            const bool last = hi >= k1;
            int nlo, nhi, nk0 = 0, nk1 = 0;
            int r0C = -1, nrC = 0, kbC = 0, keC = 0;
            if (!last) { nlo = hi; nhi = min(nlo + STEP, k1); }
            else {
                if (r0B >= 0) { nk0 = __builtin_amdgcn_readlane(kbB, 0); nk1 = __builtin_amdgcn_readlane(keB, nrB - 1); }
                nlo = nk0; nhi = min(nlo + STEP, nk1);
                advance(v, r0C, nrC);
                load_ia(r0C, nrC, kbC, keC);
            }
            if (!last || r0B >= 0) stage_load(nlo & ~3, nhi);
#pragma unroll
            for (int u = 0; u < 8; ++u) sv[lane + 64 * u] = w[u] * xv[u];   // (beyond the chunk: 0.0 * x[0], never summed)
            wave_order();
            // phase 2: lane = row, storage order
            // Eight products per LDS round trip (the adds stay one after the other, in storage order); the last,
            // partial batch reads clamped slots and keeps the sum where the slot is past the row's end.
            if (lane < nrA) {
                const int pb = max(kbA, lo), pe = min(keA, hi);
                for (int k = pb; k < pe; k += 8) {
                    double p[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) p[u] = sv[min(k + u, pe - 1) - s];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const double t = (OP == OP_JACOBI || OP == OP_L1DIAG) ? acc - p[u] : acc + p[u];
                        const bool take = (k + u < pe) && !(OP == OP_JACOBI && k + u == dk);
                        acc = take ? t : acc;
                    }
                }
            }
            wave_order();
            if (last) {
                if (lane < nrA) row_epilogue<OP>(a, r, acc, dotacc);
                if (r0B >= 0) stage_store();
                wave_order();
                r0A = r0B; nrA = nrB; kbA = kbB; keA = keB;
                r0B = r0C; nrB = nrC; kbB = kbC; keB = keC;
                k0 = nk0; k1 = nk1; lo = nlo; hi = nhi;
                break;
            }
            stage_store();
            wave_order();
            lo = nlo; hi = nhi;
        }
    }
    if (OP == OP_MXV_DOT || (OP == OP_JACOBI && a.partials)) {
        const double tot = block_sum(dotacc, red);
        if (threadIdx.x == 0) a.partials[blockIdx.x] = tot;
    }
}

// ---------------------------------------------------------------------------
// k_csr_xtile<OP>: k_csr_wstream2 with the operand staged per tile.  The levels this serves (20-60 nonzeros per row,
// x of tens of MB) were bound by the x gathers: ~30 distinct cache lines per 64-lane gather, every one an L1 miss,
// 2-3 GB of L2 -> L1 line traffic per pass beside 0.6 GB of matrix stream -- whatever the occupancy, the grid or the
// order of the entries.  But the 64 rows of a wave tile share their columns: ~2 240 entries
// touch only 300-600 DISTINCT columns.  At upload every tile gets the sorted list of its distinct columns and every
// entry the 16-bit position of its column in that list (device_csr.hip.h, build_xtile); the kernel gathers each
// distinct x entry ONCE per tile into the wave's LDS (a handful of well-clustered gathers) and the chunk loop --
// 16-byte staged loads of the 16-bit indices and the values, products, left-to-right row sums -- never leaves the
// LDS.  Bytes per entry 8 + 2 (+ ~0.7 for the lists) instead of 12; the arithmetic and its order are k_csr_wstream2's
// (bit-identical results).  A tile with more than XCAP distinct columns has no list: its operands are gathered from
// global memory (the builder accepts a few such tiles -- the seams of a clustered numbering).
// ---------------------------------------------------------------------------
constexpr int XT_XCAP = 1024;
template <int OP>
__global__ __launch_bounds__(BLOCK) void k_csr_xtile(CsrArgs a)
{
    if (a.stop && *a.stop) return;
    constexpr int CAP = 512, NV = 4, XR = XT_XCAP / 64;
    __shared__ __attribute__((aligned(16))) double         sv_all[4 * CAP];      // values, then (in place) products
    __shared__ __attribute__((aligned(16))) unsigned short sj_all[4 * CAP];      // local column positions
    __shared__ __attribute__((aligned(16))) double         xl_all[4 * XT_XCAP];  // the tile's distinct x entries
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double*         sv = sv_all + wave * CAP;
    unsigned short* sj = sj_all + wave * CAP;
    double*         xl = xl_all + wave * XT_XCAP;
    const int vmax = tile_vmax(a);
    const int G = gridDim.x;
    double dotacc = 0.0;

    auto advance = [&](int& v, int& r0, int& nr) {
        for (;;) {
            r0 = -1; nr = 0;
            if (v >= vmax) return;
            const int t = tile_of(a, v);
            v += G;
            if (t >= a.ntiles) continue;
            const int rr = (t + a.tile0) * BLOCK + wave * 64;
            if (rr >= a.nrow) continue;
            r0 = rr; nr = min(64, a.nrow - rr);
            return;
        }
    };
    auto load_ia = [&](int r0, int nr, int& kb, int& ke) {
        kb = ke = 0;
        if (r0 >= 0 && lane < nr) { kb = a.ia[r0 + lane]; ke = a.ia[r0 + lane + 1]; }
    };
    u32x4_t qj;
    f64x2_t qv[NV];
    // chunk [lo, hi) staged from the 8-aligned entry s = lo & ~7 (16 bytes of 16-bit positions): slab index of entry k is k - s
    auto stage_load = [&](int s, int hi) {
        const int n8 = (hi - s + 7) & ~7;
        const __amdgpu_buffer_rsrc_t rj = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.lja16 + s), 0, n8 * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.val + s), 0, n8 * 8, 0x00020000);
        qj = __builtin_amdgcn_raw_buffer_load_b128(rj, lane * 16, 0, 2);
#pragma unroll
        for (int q = 0; q < NV; ++q)
            qv[q] = __builtin_bit_cast(f64x2_t, __builtin_amdgcn_raw_buffer_load_b128(rv, (lane + 64 * q) * 16, 0, 2));
    };
    auto stage_store = [&]() {
        reinterpret_cast<u32x4_t*>(sj)[lane] = qj;
#pragma unroll
        for (int q = 0; q < NV; ++q) reinterpret_cast<f64x2_t*>(sv)[lane + 64 * q] = qv[q];
    };
    // the distinct x entries of wave tile r0 / 64 into the wave's LDS (positions at or beyond the list read column 0)
    auto stage_x = [&](int r0) -> int {
        const int t = r0 >> 6;
        const int p0 = a.tptr[t], n = a.tptr[t + 1] - p0;
        if (n == 0) return 0;   // a fat tile (more than XT_XCAP distinct columns: no list) -- its operands come from global memory
        int    col[XR];
        double xv[XR];
#pragma unroll
        for (int q = 0; q < XR; ++q) col[q] = (lane + 64 * q < n) ? a.tcols[p0 + lane + 64 * q] : -1;
#pragma unroll
        for (int q = 0; q < XR; ++q) xv[q] = (col[q] >= 0) ? a.x[col[q]] : 0.0;
#pragma unroll
        for (int q = 0; q < XR; ++q)
            if (64 * q < n) xl[lane + 64 * q] = xv[q];     // wave-uniform skip of the rounds beyond the list
        return n;
    };
    constexpr int STEP = CAP - 8;

    int v = blockIdx.x;
    int r0A, nrA, kbA, keA, r0B, nrB, kbB, keB;
    advance(v, r0A, nrA);
    load_ia(r0A, nrA, kbA, keA);
    advance(v, r0B, nrB);
    load_ia(r0B, nrB, kbB, keB);
    int k0 = 0, k1 = 0, lo = 0, hi = 0;
    int nxA = 0;   // distinct columns of tile A in the wave's LDS image (0: fat tile, gather from global memory)
    if (r0A >= 0) {
        k0 = __builtin_amdgcn_readlane(kbA, 0); k1 = __builtin_amdgcn_readlane(keA, nrA - 1);
        lo = k0; hi = min(lo + STEP, k1);
        stage_load(lo & ~7, hi);
        nxA = stage_x(r0A);
        stage_store();
    }
    wave_order();
    while (r0A >= 0) {
        const int r = r0A + lane;
        double acc = ((OP == OP_JACOBI || OP == OP_L1DIAG) && lane < nrA) ? a.b[r] : 0.0;
        const int dk = (OP == OP_JACOBI && lane < nrA) ? a.dpos[r] : -1;
        for (;;) {   // chunks of tile A; the slab holds [lo & ~7, hi)
            const int s = lo & ~7;
            // look ahead first: the next chunk of this tile, or the first chunk of tile B (and the row pointers of tile C)
            const bool last = hi >= k1;
            int nlo, nhi, nk0 = 0, nk1 = 0;
            int r0C = -1, nrC = 0, kbC = 0, keC = 0;
            if (!last) { nlo = hi; nhi = min(nlo + STEP, k1); }
            else {
                if (r0B >= 0) { nk0 = __builtin_amdgcn_readlane(kbB, 0); nk1 = __builtin_amdgcn_readlane(keB, nrB - 1); }
                nlo = nk0; nhi = min(nlo + STEP, nk1);
                advance(v, r0C, nrC);
                load_ia(r0C, nrC, kbC, keC);
            }
            if (!last || r0B >= 0) stage_load(nlo & ~7, nhi);
            // phase 1: lane = entry -- positions and values from the slab, x from the tile's LDS image, products in place
            // (slab entries beyond the chunk hold position 0 / value 0.0, or the next rows' entries: never summed)
            int    c[8];
            double w[8], xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) c[u] = sj[lane + 64 * u];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = sv[lane + 64 * u];
            if (nxA > 0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) xv[u] = xl[(s + lane + 64 * u < hi) ? c[u] : 0];
            } else {   // wave-uniform, rare: a tile without an LDS image
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int k = s + lane + 64 * u; xv[u] = (k >= lo && k < hi) ? a.x[a.ja[k]] : 0.0; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) sv[lane + 64 * u] = w[u] * xv[u];
            wave_order();
            // phase 2: lane = row, storage order, eight products per LDS round trip
            if (lane < nrA) {
                const int pb = max(kbA, lo), pe = min(keA, hi);
                for (int k = pb; k < pe; k += 8) {
                    double p[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) p[u] = sv[min(k + u, pe - 1) - s];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const double t = (OP == OP_JACOBI || OP == OP_L1DIAG) ? acc - p[u] : acc + p[u];
                        const bool take = (k + u < pe) && !(OP == OP_JACOBI && k + u == dk);
                        acc = take ? t : acc;
                    }
                }
            }
            wave_order();
            if (last) {
                if (lane < nrA) row_epilogue<OP>(a, r, acc, dotacc);
                if (r0B >= 0) { nxA = stage_x(r0B); stage_store(); }
                wave_order();
                r0A = r0B; nrA = nrB; kbA = kbB; keA = keB;
                r0B = r0C; nrB = nrC; kbB = kbC; keB = keC;
                k0 = nk0; k1 = nk1; lo = nlo; hi = nhi;
                break;
            }
            stage_store();
            wave_order();
            lo = nlo; hi = nhi;
        }
    }
    if (OP == OP_MXV_DOT || (OP == OP_JACOBI && a.partials)) {
        const double tot = block_sum(dotacc, red);
        if (threadIdx.x == 0) a.partials[blockIdx.x] = tot;
    }
}

__device__ __forceinline__ f64x2_t buf_load_f64x2(__amdgpu_buffer_rsrc_t rs, unsigned byte_off)
{
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 0);
    return __builtin_bit_cast(f64x2_t, v);
}

// ---------------------------------------------------------------------------
// k_csr_rowpat4<OP>: k_csr_rowpat3 without its divergent tail.  The sweep computes every row pair
// whose two rows have the pattern of their wave's middle row (row 128 w + 64 of wave tile w) and stores
// nothing for the others; those rows (domain boundaries: 2-3 % of the rows of a 3-D stencil) are parked in
// the wave's LDS queue and computed lane = row, 8 gathers in flight, 64 at a time by the wave that met
// them -- while the x planes they read are still in this XCD's L2.  (Round 2 computed them from a list built
// at upload, in a pass behind the sweep: its scattered 64-byte fetches were 0.2 GB of the 0.56 GB a level-0
// pass of P7(256) moved.  Same time, less traffic: profiles/r03_coded_kernel_experiments.txt.)  Pattern table
// read through the scalar cache in the sweep, through the vector cache for the queued rows.
// ---------------------------------------------------------------------------
constexpr int RP_QCAP = 192;   // per-wave queue of deferred rows: < 64 left over + at most 128 pushed by one step
template <int OP>
__global__ __launch_bounds__(BLOCK) void k_csr_rowpat4(CsrArgs a)
{
    if (a.stop && *a.stop) return;
    __shared__ double red[4];
    __shared__ int xq[4 * RP_QCAP];
    int* const myq = xq + (threadIdx.x >> 6) * RP_QCAP;
    int        qn = 0;   // wave-uniform
    const int  lane = threadIdx.x & 63;
    const bool stream = (a.nt & 4) != 0;   // wave-uniform: y, pattern ids and b with streaming (nt) hints
    constexpr bool NEG = (OP == OP_JACOBI || OP == OP_L1DIAG);
    const __amdgpu_buffer_rsrc_t xr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.x), 0, (int)((unsigned)a.ncol * 8u), 0x00020000);
    const __amdgpu_buffer_rsrc_t yr =
        __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)((unsigned)(a.nrow & ~1) * 8u), 0x00020000);
    typedef const __attribute__((address_space(4))) int    cint_t;
    typedef const __attribute__((address_space(4))) double cdbl_t;
    cint_t* cpstart = (cint_t*)(a.pstart);
    cint_t* cplen   = (cint_t*)(a.plen);
    cint_t* cpoff   = (cint_t*)(a.poff);
    cdbl_t* cpval   = (cdbl_t*)(a.pval);
    const int vmax = tile_vmax(a);  // tiles of 2 * BLOCK rows
    const int G = gridDim.x;
    const int last = a.nrow - 1;
    const unsigned* pat2 = reinterpret_cast<const unsigned*>(a.pat);
    const int npair = (a.nrow + 1) >> 1;
    double dotacc = 0.0;

    auto advance = [&](int& v) -> int {
        for (;;) {
            if (v >= vmax) return -1;
            const int t = tile_of(a, v);
            v += G;
            if (t < a.ntiles) return (t + a.tile0) * (2 * BLOCK);
        }
    };
    auto ld_pp = [&](int r0) -> unsigned { const unsigned* q = pat2 + min((max(r0, 0) >> 1) + (int)threadIdx.x, npair - 1); return stream ? __builtin_nontemporal_load(q) : *q; };

    // a row outside its wave's pattern: lane = row, its own list through the vector cache, 8 gathers in flight
    auto one_row = [&](int r) {
        const unsigned pid = a.pat[r];
        const int      ps = a.pstart[pid], len = a.plen[pid];
        const unsigned base = (unsigned)r * 8u;
        double acc = NEG ? a.b[r] : 0.0, xi = 0.0, dgr = 0.0;
        bool   hasd = false;
        for (int k = 0; k < len; k += 8) {
            int    of[8];
            double xk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) of[u] = a.poff[ps + k + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) xk[u] = buf_load_f64(xr, base + (unsigned)(of[u] * 8));
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double w = a.pval[ps + k + u];
                const double pr = w * xk[u];
                bool use = k + u < len;
                if (OP == OP_JACOBI) {
                    if (use && of[u] == 0) { xi = xk[u]; dgr = w; hasd = true; }
                    use = use && of[u] != 0;
                }
                const double nxt = NEG ? acc - pr : acc + pr;
                acc = use ? nxt : acc;
            }
        }
        if (OP == OP_JACOBI) {
            if (!hasd) xi = a.x[r];
            const double xn = (fabs(dgr) > 1e-20) ? (1 - a.omega) * xi + a.omega * acc / dgr : xi;
            a.y[r] = xn;
            if (a.partials) dotacc += xn * a.b[r];
        } else if (OP == OP_L1DIAG) {
            a.y[r] = l1_or_jacobi_f(a, r, acc, a.diag[r], a.x[r]);
        } else {
            row_epilogue<OP>(a, r, acc, dotacc);
        }
    };
    auto drain = [&](int n) {
        wave_order();
        const int r = lane < n ? myq[qn - n + lane] : -1;
        if (r >= 0) one_row(r);
        qn -= n;
    };
    int      v = blockIdx.x;
    int      r0A = advance(v);
    unsigned pp = ld_pp(r0A);
    f64x2_t  pend_out = {0.0, 0.0};
    unsigned pend_off = 0xfffffff0u;
    auto flush = [&]() {  // one unconditional buffer store per step; lanes with nothing to store are out of range
        // (streaming: y must not push the x planes out of the L2 -- only where the vectors exceed the Infinity Cache anyway)
        if (stream) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, pend_out), yr, (int)pend_off, 0, 2);
        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, pend_out), yr, (int)pend_off, 0, 0);
        pend_off = 0xfffffff0u;
    };
    while (r0A >= 0) {
        const int      r0B = advance(v);
        const unsigned ppB = ld_pp(r0B);
        const int  ra = r0A + 2 * (int)threadIdx.x;
        const bool vb = ra + 1 <= last;
        const unsigned pidA = pp & 0xffffu, pidB = pp >> 16;
        const unsigned dom = (unsigned)__builtin_amdgcn_readlane((int)pidA, 32);
        const bool     mine = vb && pidA == dom && pidB == dom;  // everything else is on the list
        const int      dps = cpstart[dom], dlen = cplen[dom];
        const unsigned baseA = (unsigned)min(ra, last) * 8u;
        const int      rs = mine ? ra : 0;  // lanes that store nothing load from the first rows: always valid
        f64x2_t bb = {0.0, 0.0}, aux = {0.0, 0.0};
        if (OP == OP_JACOBI || OP == OP_L1DIAG || OP == OP_RESID) bb = stream ? __builtin_nontemporal_load(reinterpret_cast<const f64x2_t*>(a.b + rs)) : *reinterpret_cast<const f64x2_t*>(a.b + rs);
        if (OP == OP_MXV_DOT) aux = *reinterpret_cast<const f64x2_t*>(a.dotv + rs);
        else if (OP == OP_ADD || OP == OP_SUB || OP == OP_AXPY) aux = *reinterpret_cast<const f64x2_t*>(a.y + rs);
        else if (OP == OP_L1DIAG) aux = *reinterpret_cast<const f64x2_t*>(a.diag + rs);
        double accA = NEG ? bb.x : 0.0, accB = NEG ? bb.y : 0.0;
        double xiA = 0.0, xiB = 0.0, dg = 0.0;
        for (int k = 0; k < dlen; k += 8) {  // wave-uniform trip count; lists are padded to multiples of 8 (offset 0)
            int     of[8];
            f64x2_t xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) of[u] = cpoff[dps + k + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[u] = buf_load_f64x2(xr, baseA + (unsigned)(of[u] * 8));
            if (k == 0) flush();  // the previous step's result leaves behind this step's loads
            double w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = cpval[dps + k + u];
            auto entry = [&](int u) {
                const double pa = w[u] * xv[u].x, pb = w[u] * xv[u].y;
                if (OP == OP_JACOBI) {
                    if (of[u] != 0) { accA -= pa; accB -= pb; }
                    else { xiA = xv[u].x; xiB = xv[u].y; dg = w[u]; }
                } else if (OP == OP_L1DIAG) { accA -= pa; accB -= pb; }
                else { accA += pa; accB += pb; }
            };
            switch (min(dlen - k, 8)) {  // padding entries contribute nothing, not even a signed zero
                case 8: entry(0); entry(1); entry(2); entry(3); entry(4); entry(5); entry(6); entry(7); break;
                case 7: entry(0); entry(1); entry(2); entry(3); entry(4); entry(5); entry(6); break;
                case 6: entry(0); entry(1); entry(2); entry(3); entry(4); entry(5); break;
                case 5: entry(0); entry(1); entry(2); entry(3); entry(4); break;
                case 4: entry(0); entry(1); entry(2); entry(3); break;
                case 3: entry(0); entry(1); entry(2); break;
                case 2: entry(0); entry(1); break;
                default: entry(0); break;
            }
        }
        if (dlen == 0) flush();
        f64x2_t out = {0.0, 0.0};
        if (OP == OP_MXV) { out.x = accA; out.y = accB; if (mine) { zx_store(a, ra, accA); zx_store(a, ra + 1, accB); } }
        else if (OP == OP_RESID) { out.x = bb.x - accA; out.y = bb.y - accB; }
        else if (OP == OP_ADD) { out.x = aux.x + accA; out.y = aux.y + accB; }
        else if (OP == OP_SUB) { out.x = aux.x - accA; out.y = aux.y - accB; }
        else if (OP == OP_AXPY) { out.x = aux.x + accA * a.alpha; out.y = aux.y + accB * a.alpha; }
        else if (OP == OP_JACOBI) {
            if (dg == 0.0) {  // wave-uniform: the pattern stores no diagonal
                const f64x2_t xi = *reinterpret_cast<const f64x2_t*>(a.x + rs);
                xiA = xi.x; xiB = xi.y;
            }
            out.x = (fabs(dg) > 1e-20) ? (1 - a.omega) * xiA + a.omega * accA / dg : xiA;
            out.y = (fabs(dg) > 1e-20) ? (1 - a.omega) * xiB + a.omega * accB / dg : xiB;
            if (a.partials && mine) { dotacc += out.x * bb.x; dotacc += out.y * bb.y; }   // (z, r) of the last level-0 sweep
        } else if (OP == OP_L1DIAG) {
            const f64x2_t xi = *reinterpret_cast<const f64x2_t*>(a.x + rs);
            out.x = l1_or_jacobi_f(a, rs, accA, aux.x, xi.x);
            out.y = l1_or_jacobi_f(a, rs + 1, accB, aux.y, xi.y);
        } else {  // OP_MXV_DOT
            out.x = accA; out.y = accB;
            if (mine) { dotacc += accA * aux.x; dotacc += accB * aux.y; }
        }
        if (mine) { pend_out = out; pend_off = (unsigned)ra * 8u; }
        {   // pairs that are not the wave's pattern: queued, computed 64 rows at a time by this wave
            const bool ex = ra <= last && !mine;
            const unsigned long long exm = __ballot(ex);
            if (exm) {
                const int pos = qn + 2 * __popcll(exm & ((1ull << lane) - 1ull));
                if (ex) { myq[pos] = ra; myq[pos + 1] = (ra + 1 <= last) ? ra + 1 : -1; }
                qn += 2 * __popcll(exm);
                while (qn >= 64) drain(64);
            }
        }
        r0A = r0B;
        pp = ppB;
    }
    flush();

    if (qn > 0) drain(qn);
    if (OP == OP_MXV_DOT || (OP == OP_JACOBI && a.partials)) {
        const double tot = block_sum(dotacc, red);
        if (threadIdx.x == 0) a.partials[blockIdx.x] = tot;
    }
}

// ---------------------------------------------------------------------------
// k_csr_rowpat5<OP>: k_csr_rowpat4 for row-pattern-coded RECTANGULAR operators (the restriction and prolongation of
// the coded levels: every row has its own column base, a.rowbase).  A lane again owns rows 2i and 2i+1; the PAIR of
// patterns of the wave's middle lane is treated as wave-uniform -- the rows of a transfer operator alternate between
// two patterns (C point: one entry; F point: its C neighbours), in one order along a grid line and the other along the
// next -- both lists through the scalar cache.  x is gathered with 8-byte buffer loads (the two rows have unrelated
// bases), y / the dotted vector move as 16-byte accesses, the store is deferred behind the next step's loads, pairs with
// other patterns go through the wave's LDS queue (k_csr_rowpat4).  OPs: MXV, ADD, SUB, AXPY,
// RESID, MXV_DOT (no smoother runs on a rectangular operator).
// ---------------------------------------------------------------------------
template <int OP>
__global__ __launch_bounds__(BLOCK) void k_csr_rowpat5(CsrArgs a)
{
    if (a.stop && *a.stop) return;
    static_assert(OP != OP_JACOBI && OP != OP_L1DIAG, "transfer operators are not smoothed");
    __shared__ double red[4];
    __shared__ int xq[4 * RP_QCAP];
    int* const myq = xq + (threadIdx.x >> 6) * RP_QCAP;
    int        qn = 0;   // wave-uniform
    const int  lane = threadIdx.x & 63;
    const bool stream = (a.nt & 4) != 0;   // wave-uniform: y, pattern ids and b with streaming (nt) hints
    const __amdgpu_buffer_rsrc_t xr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.x), 0, (int)((unsigned)a.ncol * 8u), 0x00020000);
    const __amdgpu_buffer_rsrc_t yr =
        __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)((unsigned)(a.nrow & ~1) * 8u), 0x00020000);
    typedef const __attribute__((address_space(4))) int    cint_t;
    typedef const __attribute__((address_space(4))) double cdbl_t;
    cint_t* cpstart = (cint_t*)(a.pstart);
    cint_t* cplen   = (cint_t*)(a.plen);
    cint_t* cpoff   = (cint_t*)(a.poff);
    cdbl_t* cpval   = (cdbl_t*)(a.pval);
    const int vmax = tile_vmax(a);  // tiles of 2 * BLOCK rows
    const int G = gridDim.x;
    const int last = a.nrow - 1;
    const unsigned* pat2 = reinterpret_cast<const unsigned*>(a.pat);
    const int npair = (a.nrow + 1) >> 1;
    double dotacc = 0.0;

    auto advance = [&](int& v) -> int {
        for (;;) {
            if (v >= vmax) return -1;
            const int t = tile_of(a, v);
            v += G;
            if (t < a.ntiles) return (t + a.tile0) * (2 * BLOCK);
        }
    };
    auto pair_of = [&](int r0) -> int { return min((max(r0, 0) >> 1) + (int)threadIdx.x, npair - 1); };
    // a row outside its wave's pattern pair: lane = row, its own list through the vector cache, 8 gathers in flight
    auto one_row = [&](int r) {
        const unsigned pid = a.pat[r];
        const int      ps = a.pstart[pid], len = a.plen[pid];
        const unsigned base = (unsigned)a.rowbase[r] * 8u;
        double acc = 0.0;
        for (int k = 0; k < len; k += 8) {
            int    of[8];
            double xk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) of[u] = a.poff[ps + k + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) xk[u] = buf_load_f64(xr, base + (unsigned)(of[u] * 8));
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double pr = a.pval[ps + k + u] * xk[u];
                acc = (k + u < len) ? acc + pr : acc;
            }
        }
        row_epilogue<OP>(a, r, acc, dotacc);
    };
    auto drain = [&](int n) {   // (k_csr_rowpat4: the deferred rows of the sweep, 64 at a time, by the wave that met them)
        wave_order();
        const int r = lane < n ? myq[qn - n + lane] : -1;
        if (r >= 0) one_row(r);
        qn -= n;
    };
    int      v = blockIdx.x;
    int      r0A = advance(v);
    unsigned pp = stream ? __builtin_nontemporal_load(pat2 + pair_of(r0A)) : pat2[pair_of(r0A)];
    f64x2_t  pend_out = {0.0, 0.0};
    unsigned pend_off = 0xfffffff0u;
    auto flush = [&]() {
        if (stream) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, pend_out), yr, (int)pend_off, 0, 2);
        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, pend_out), yr, (int)pend_off, 0, 0);
        pend_off = 0xfffffff0u;
    };
    // the two wave-uniform lists over per-lane bases, chunk by chunk in lockstep: 16 gathers in flight per lane
    auto sweep_lists = [&](int psA, int lenA, unsigned baseA8, double& accA, int psB, int lenB, unsigned baseB8, double& accB) {
        const int lmax = max(lenA, lenB);
        for (int k = 0; k < lmax; k += 8) {
            const bool doA = k < lenA, doB = k < lenB;   // wave-uniform
            int    ofA[8], ofB[8];
            double xa[8], xb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { ofA[u] = doA ? cpoff[psA + k + u] : 0; ofB[u] = doB ? cpoff[psB + k + u] : 0; }
            // only the gathers the lists have (wave-uniform counts: scalar branches) -- a prolongation's rows hold 1 and 2-6 entries, padded
            // to 8: all 16 gathers of a chunk pair were issued for 3-7 operands (round 5: P of level 0 128 -> 94 us; the same in the square sweep, whose lists fill 7 of 8 and 19 of 24 slots, gains nothing)
            const int nA = doA ? min(lenA - k, 8) : 0, nB = doB ? min(lenB - k, 8) : 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                xa[u] = 0.0;
                if (u < nA) xa[u] = buf_load_f64(xr, baseA8 + (unsigned)(ofA[u] * 8));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                xb[u] = 0.0;
                if (u < nB) xb[u] = buf_load_f64(xr, baseB8 + (unsigned)(ofB[u] * 8));
            }
            if (k == 0) flush();  // the previous step's result leaves behind this step's first loads
            auto chunk = [&](int ps, int len, const double (&xv)[8], double& acc) {
                double w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) w[u] = cpval[ps + k + u];
                auto entry = [&](int u) { acc += w[u] * xv[u]; };
                switch (min(len - k, 8)) {  // padding entries contribute nothing, not even a signed zero
                    case 8: entry(0); entry(1); entry(2); entry(3); entry(4); entry(5); entry(6); entry(7); break;
                    case 7: entry(0); entry(1); entry(2); entry(3); entry(4); entry(5); entry(6); break;
                    case 6: entry(0); entry(1); entry(2); entry(3); entry(4); entry(5); break;
                    case 5: entry(0); entry(1); entry(2); entry(3); entry(4); break;
                    case 4: entry(0); entry(1); entry(2); entry(3); break;
                    case 3: entry(0); entry(1); entry(2); break;
                    case 2: entry(0); entry(1); break;
                    default: entry(0); break;
                }
            };
            if (doA) chunk(psA, lenA, xa, accA);
            if (doB) chunk(psB, lenB, xb, accB);
        }
        if (lmax == 0) flush();
    };
    while (r0A >= 0) {
        const int      r0B = advance(v);
        const unsigned ppB = stream ? __builtin_nontemporal_load(pat2 + pair_of(r0B)) : pat2[pair_of(r0B)];
        const int  ra = r0A + 2 * (int)threadIdx.x;
        const bool vb = ra + 1 <= last;
        const unsigned pidA = pp & 0xffffu, pidB = pp >> 16;
        const unsigned domA = (unsigned)__builtin_amdgcn_readlane((int)pidA, 32);
        unsigned       domB = (unsigned)__builtin_amdgcn_readlane((int)pidB, 32);
        // (odd row count: the partner of the last row is the 0xffff pad, not a pattern -- no pair of that wave tile is swept)
        const bool     domok = domB < (unsigned)a.npat;
        if (!domok) domB = 0;
        const bool     mine = vb && domok && pidA == domA && pidB == domB;  // everything else is on the list
        const int      rs = mine ? ra : 0;  // lanes that store nothing work on the first rows: always valid
        const int      psA = cpstart[domA], lenA = cplen[domA], psB = cpstart[domB], lenB = cplen[domB];
        // the two column bases in one 8-byte load
        const int2 cb = *reinterpret_cast<const int2*>(a.rowbase + rs);
        f64x2_t aux = {0.0, 0.0};
        if (OP == OP_RESID) aux = *reinterpret_cast<const f64x2_t*>(a.b + rs);
        else if (OP == OP_MXV_DOT) aux = *reinterpret_cast<const f64x2_t*>(a.dotv + rs);
        else if (OP == OP_ADD || OP == OP_SUB || OP == OP_AXPY) aux = *reinterpret_cast<const f64x2_t*>(a.y + rs);
        double accA = 0.0, accB = 0.0;
        sweep_lists(psA, lenA, (unsigned)cb.x * 8u, accA, psB, lenB, (unsigned)cb.y * 8u, accB);
        f64x2_t out = {0.0, 0.0};
        if (OP == OP_MXV) { out.x = accA; out.y = accB; if (mine) { zx_store(a, ra, accA); zx_store(a, ra + 1, accB); } }
        else if (OP == OP_RESID) { out.x = aux.x - accA; out.y = aux.y - accB; }
        else if (OP == OP_ADD) { out.x = aux.x + accA; out.y = aux.y + accB; }
        else if (OP == OP_SUB) { out.x = aux.x - accA; out.y = aux.y - accB; }
        else if (OP == OP_AXPY) { out.x = aux.x + accA * a.alpha; out.y = aux.y + accB * a.alpha; }
        else {  // OP_MXV_DOT
            out.x = accA; out.y = accB;
            if (mine) { dotacc += accA * aux.x; dotacc += accB * aux.y; }
        }
        if (mine) { pend_out = out; pend_off = (unsigned)ra * 8u; }
        {   // pairs that are not the wave's pattern: queued, computed 64 rows at a time by this wave
            const bool ex = ra <= last && !mine;
            const unsigned long long exm = __ballot(ex);
            if (exm) {
                const int pos = qn + 2 * __popcll(exm & ((1ull << lane) - 1ull));
                if (ex) { myq[pos] = ra; myq[pos + 1] = (ra + 1 <= last) ? ra + 1 : -1; }
                qn += 2 * __popcll(exm);
                while (qn >= 64) drain(64);
            }
        }
        r0A = r0B;
        pp = ppB;
    }
    flush();

    if (qn > 0) drain(qn);
    if (OP == OP_MXV_DOT) {
        const double tot = block_sum(dotacc, red);
        if (threadIdx.x == 0) a.partials[blockIdx.x] = tot;
    }
}

// ---------------------------------------------------------------------------
// Device copy / triad ceilings (bench.py reports them beside the roofline): 16 bytes per lane, grid-stride
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_copy16(size_t n16, const f64x2_t* __restrict__ src, f64x2_t* __restrict__ dst)
{
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n16; i += (size_t)gridDim.x * BLOCK) dst[i] = src[i];
}
__global__ __launch_bounds__(BLOCK) void k_triad16(size_t n16, double s, const f64x2_t* __restrict__ b,
                                                   const f64x2_t* __restrict__ c, f64x2_t* __restrict__ out)
{
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n16; i += (size_t)gridDim.x * BLOCK) {
        const f64x2_t bv = b[i], cv = c[i];
        f64x2_t o;
        o.x = bv.x + s * cv.x;
        o.y = bv.y + s * cv.y;
        out[i] = o;
    }
}
__global__ __launch_bounds__(BLOCK) void k_read16(size_t n16, const f64x2_t* __restrict__ src, double* __restrict__ out)
{
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n16; i += (size_t)gridDim.x * BLOCK) {
        const f64x2_t v = src[i];
        acc += v.x + v.y;
    }
    if (acc == 1.2345e-300) out[0] = acc;  // keeps the loads alive, never true on real data
}

}  // namespace fasp
