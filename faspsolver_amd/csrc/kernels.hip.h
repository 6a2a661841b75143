// kernels.hip.h -- hand-written HIP kernels (gfx950 / CDNA4, wave64) of the
// AMG-preconditioned Krylov hot path.  Everything here is HBM-bandwidth bound
// (<= 2 flop per 12 bytes): no MFMA, the levers are coalesced row-block loads of
// IA/JA/val, sub-wavefront shuffle reductions, LDS-staged block reductions,
// non-temporal streaming of the matrix so the x vector stays in the XCD's L2, and an
// XCD-aware blockIdx -> row-tile mapping (blocks b and b+8 share an XCD).
//
// Reference semantics (file:line relative to the reference tree):
//   y = A x                 fasp_blas_dcsr_mxv      base/src/BlaSpmvCSR.c:242
//   y += alpha A x          fasp_blas_dcsr_aAxpy    base/src/BlaSpmvCSR.c:494
//   weighted Jacobi sweep   fasp_smoother_dcsr_jacobi  base/src/ItrSmootherCSR.c:98
//   L1-diagonal sweep       fasp_smoother_dcsr_L1diag  base/src/ItrSmootherCSR.c:1509
//   dot / norm2 / norminf   base/src/BlaArray.c:771 / :691 / :719
//   axpy / axpby            base/src/BlaArray.c:90 / :620
//
// Compiled with -ffp-contract=off: the reference (gcc -O3, x86-64) has no FMA, so
// every elementwise expression below rounds exactly like the reference's.  Row sums
// and reductions are evaluated in a fixed tree order (deterministic run to run) that
// differs from the reference's left-to-right order by O(1e-16) relative.
#pragma once

#include <hip/hip_runtime.h>

namespace fasp {

constexpr int BLOCK   = 256;   // 4 wavefronts
constexpr int MAXGRID = 2048;  // 256 CUs x 8 blocks: persistent grid, grid-stride inside

enum RowOp : int {
    OP_MXV = 0,     // y = t
    OP_RESID,       // y = b - t            (w = b; w -= A x  of PreMGCycle.c:136-137, KryPcg.c:125-126)
    OP_ADD,         // y += t               (aAxpy, alpha == 1: prolongation)
    OP_SUB,         // y -= t               (aAxpy, alpha == -1)
    OP_AXPY,        // y += t * alpha       (aAxpy, general alpha)
    OP_JACOBI,      // y = (1-w) x_i + w (b_i - sum_{j!=i}) / d_i
    OP_L1DIAG,      // y = x_i + (b_i - t) / l1_i
    OP_MXV_DOT      // y = t and partial sums of t_i * dotv_i   (t = A p fused with (t,p))
};

struct CsrArgs {
    int           nrow;
    const int*    ia;
    const int*    ja;
    const double* val;
    const double* x;     // gathered vector
    double*       y;     // output
    const double* b;     // rhs (RESID / JACOBI / L1DIAG)
    const double* diag;  // a_ii (JACOBI) or sum_j |a_ij| (L1DIAG)
    const int*    dpos;  // JACOBI stream kernel: storage index of the (last) diagonal entry of each row, -1 if none
    const double* dotv;  // OP_MXV_DOT: vector dotted with the result
    double*       partials;  // OP_MXV_DOT: one partial per block
    double        alpha;     // OP_AXPY
    double        omega;     // OP_JACOBI
    int           ntiles;
    int           tiles_per_xcd;
    int           tpp;      // xcd_map == -2: tiles per grid plane (tiles_per_xcd = tpp / 8: the strip of a plane one XCD sweeps)
    int           xcd_map;  // G > 0: an XCD works on runs of G consecutive tiles; 0: plain grid-stride
    int           nt;       // 1: non-temporal loads of JA / val
    // dictionary-coded matrices (k_csr_dict8): one byte per entry selects (column offset, value)
    const unsigned char* code;
    const int*    rowbase;  // column base of each row (nullptr: the row index itself)
    const int*    doff;     // 256 column offsets
    const double* dval;     // 256 values
    // row-pattern-coded matrices (k_csr_rowpat): two bytes per ROW select its whole entry list
    const unsigned short* pat;  // pattern id of every row
    const int*    pstart;   // start of every pattern in poff / pval (lists padded to multiples of 8 entries)
    const int*    plen;     // true length of every pattern
    const int*    poff;     // column offsets (relative to the row base) of every pattern, in storage order
    const double* pval;     // values (padding: offset 0, value 0)
    int           npat, npent;
    int           ncol;     // length of x (buffer-load range check)
    const int*    stop;     // != nullptr: the launch returns at once when *stop != 0 (queued-ahead iterations)
    const int*    mark;     // OP_L1DIAG only, != nullptr: C/F marker, the sweep is Jacobi on the F points (0) with weight omega
    const unsigned short* ja16;  // != nullptr: the column indices once more as 16-bit values (operators with <= 65536 columns)
    const int*            jbase; // != nullptr: ja16 is relative to this per-row base (operators with more columns whose rows span < 65536)
    // row window of a launch (distributed levels: interior rows while the halo is in flight, boundary rows after it):
    // the launch covers the tiles tile0 .. tile0 + ntiles of the kernel's own tile size, i.e. the rows [row_lo, nrow)
    int           tile0, row_lo;
    // OP_MXV only, zx != nullptr: the result is the right-hand side of a level whose first smoothing step is a Jacobi
    // sweep from a zero guess -- the kernel writes that sweep's result too, zx_i = (w y_i) / zdiag_i (k_jacobi_zero's
    // expression), saving the pass that would read y again
    double*       zx;
    const double* zdiag;
    double        zomega;
    // k_csr_xtile: per 64-row wave tile the sorted list of its distinct columns (tcols[tptr[t] .. tptr[t+1])) and, per
    // entry, the 16-bit position of its column in that list
    const unsigned short* lja16;
    const int*    tptr;
    const int*    tcols;
    // k_csr_estream (kernels3.hip.h): tables of the entry-parallel decomposition, built at upload (device_csr.hip.h, build_estream)
    const int*    es_wc;      // first chunk of every wave range (W + 1)
    const int*    es_centry;  // first entry of every chunk (chunks + 1; multiples of 8 but the last)
    const int*    es_crow;    // the row that entry lies in (chunks + 1)
    const int*    es_hw0;     // per wave: the wave in which the row its range begins inside of starts
    const int*    es_np;      // per wave: parts of the row its range ends inside of, when that row starts in this wave (else 0)
    const int*    es_cbase;   // per chunk: its smallest column (nullptr: the 16-bit columns are absolute)
    const unsigned short* es_ja16;   // 16-bit columns relative to es_cbase (nullptr: ja16 as it is)
    double*       es_part;    // 2 per wave: head part, tail part
    unsigned*     es_cnt;     // per wave: parts arrived (zero between launches)
};

__device__ __forceinline__ void zx_store(const CsrArgs& a, int r, double s)
{
    if (a.zx) {
        const double di = a.zdiag[r];
        a.zx[r] = (fabs(di) > 1e-20) ? (1 - a.zomega) * 0.0 + a.zomega * s / di : 0.0;
    }
}

// Epilogue of OP_L1DIAG, t = b_i - sum_j a_ij x_j accumulated from b_i entry by entry: the L1
// smoother x_i + t / sum_j |a_ij| (ItrSmootherCSR.c:1509), or -- with a C/F marker -- Jacobi on the
// F points only, x_i + w t / a_ii there and x_i elsewhere (fasp_smoother_dcsr_jacobi_ff, :34-74;
// no guard on the diagonal in the reference).
__device__ __forceinline__ double l1_or_jacobi_f(const CsrArgs& a, int r, double t, double d, double xi)
{
    if (a.mark) return a.mark[r] == 0 ? xi + a.omega * t / d : xi;
    return (fabs(d) > 1e-20) ? xi + t / d : xi;
}

// Column index of entry k.  Long-row operators with at most 65536 columns carry a 16-bit copy of JA:
// 10 instead of 12 bytes per entry on the levels where the matrix stream is all there is.
__device__ __forceinline__ int ld_ja(const CsrArgs& a, int k)
{
    if (a.ja16) return a.nt ? (int)__builtin_nontemporal_load(a.ja16 + k) : (int)a.ja16[k];
    return a.nt ? __builtin_nontemporal_load(a.ja + k) : a.ja[k];
}
__device__ __forceinline__ double ld_val(const CsrArgs& a, int k)
{
    return a.nt ? __builtin_nontemporal_load(a.val + k) : a.val[k];
}
// Persistent-grid tile schedule.  Block b visits the virtual indices v = b, b + grid, ...;
// v is mapped to a row tile so that the tiles an XCD works on (blocks b and b+8 share an
// XCD) come in runs of G consecutive tiles: neighbouring rows, which gather the same x
// entries, then hit the same XCD-private L2.  xcd_map = G (0: identity, plain grid-stride).
__device__ __forceinline__ int tile_vmax(const CsrArgs& a)
{
    if (a.xcd_map == -2) return 8 * a.tiles_per_xcd * ((a.ntiles + a.tpp - 1) / a.tpp);  // strip mode: planes x strip
    if (a.xcd_map < 0) return 8 * a.tiles_per_xcd;  // slab mode
    if (a.xcd_map <= 0) return a.ntiles;
    const int span = 8 * a.xcd_map;
    return (a.ntiles + span - 1) / span * span;
}
__device__ __forceinline__ int tile_of(const CsrArgs& a, int v)
{
    // xcd_map < 0: every XCD sweeps ONE contiguous eighth of the rows in order (its blocks take
    // consecutive tiles), so an x entry gathered by rows far apart in index but close in the
    // sweep (the +-nx*ny neighbours of a 3-D stencil) is fetched once per XCD and then hit in
    // that XCD's L2.  Pays when x dominates the traffic (compressed matrices).
    // xcd_map == -2 (operators on a 3-D grid: rows x-fastest, tpp tiles per z-plane): every XCD sweeps ONE STRIP of every plane --
    // tpp / 8 consecutive tiles, i.e. an eighth of the y range -- plane after plane.  What an XCD keeps of x between the sweep of
    // plane z and of plane z + 1 is three strips (a few hundred KB at 256^3), not three planes next to the streams of y and the
    // pattern ids in a 4 MB L2; an entry of x is fetched by the XCD whose strip holds it and by its two y-neighbours' edges.
    if (a.xcd_map == -2) {
        const int q = v >> 3, z = q / a.tiles_per_xcd;
        return z * a.tpp + (v & 7) * a.tiles_per_xcd + (q - z * a.tiles_per_xcd);
    }
    if (a.xcd_map < 0) return (v & 7) * a.tiles_per_xcd + (v >> 3);
    if (a.xcd_map <= 0) return v;
    const int G = a.xcd_map;
    const int xcd = v & 7, q = v >> 3;
    int chunk, within;
    if ((G & (G - 1)) == 0) { const int sh = __ffs(G) - 1; chunk = q >> sh; within = q & (G - 1); }  // no integer division
    else { chunk = q / G; within = q - chunk * G; }
    return (chunk * 8 + xcd) * G + within;
}

template <int W>
__device__ __forceinline__ double subwave_sum(double v)
{
#pragma unroll
    for (int off = W / 2; off > 0; off >>= 1) v += __shfl_down(v, off, W);
    return v;
}

// orders a wavefront's own LDS writes before its later LDS reads (no workgroup barrier)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Deterministic block reduction: wave shuffle tree, then the 4 wave results are added
// in wave order by thread 0.
__device__ __forceinline__ double block_sum(double v, double* lds /* >= 4 doubles */)
{
    v = subwave_sum<64>(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = ((lds[0] + lds[1]) + lds[2]) + lds[3];
    return r;
}
__device__ __forceinline__ double block_max(double v, double* lds)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = fmax(fmax(lds[0], lds[1]), fmax(lds[2], lds[3]));
    return r;
}

// ---------------------------------------------------------------------------
// CSR row kernel: L lanes cooperate on one row (L = 2..64, chosen per matrix from its
// nnz/row), a 256-thread block covers 256/L consecutive rows per tile, so a wavefront's
// val/JA loads cover one contiguous span of the CSR arrays.  The grid is persistent
// (<= 2048 blocks); block b works on tiles of XCD slab (b & 7) so that each XCD
// streams one contiguous eighth of the matrix and re-uses its part of x in its own L2.
// val/JA are read exactly once per launch -> non-temporal loads.
// ---------------------------------------------------------------------------
template <int L, int OP>
__global__ __launch_bounds__(BLOCK) void k_csr_rows(CsrArgs a)
{
    if (a.stop && *a.stop) return;
    constexpr int RPB = BLOCK / L;
    const int sl   = threadIdx.x & (L - 1);
    const int rloc = threadIdx.x / L;
    const int vmax = tile_vmax(a);
    double acc = 0.0;

    for (int v = blockIdx.x; v < vmax; v += gridDim.x) {
        const int t = tile_of(a, v);
        if (t >= a.ntiles) continue;
        const int r = (t + a.tile0) * RPB + rloc;
        if (r < a.nrow) {
            const int kb = a.ia[r], ke = a.ia[r + 1];
            const int jb = a.jbase ? a.jbase[r] : 0;   // (16-bit indices relative to the row's smallest column)
            double s = 0.0;
            // 4 independent (JA, val) loads and x gathers in flight per lane; the adds stay
            // in k order
            int k = kb + sl;
            for (; k + 3 * L < ke; k += 4 * L) {
                const int    c0 = jb + ld_ja(a, k);
                const int    c1 = jb + ld_ja(a, k + L);
                const int    c2 = jb + ld_ja(a, k + 2 * L);
                const int    c3 = jb + ld_ja(a, k + 3 * L);
                const double v0 = ld_val(a, k);
                const double v1 = ld_val(a, k + L);
                const double v2 = ld_val(a, k + 2 * L);
                const double v3 = ld_val(a, k + 3 * L);
                const double x0 = a.x[c0], x1 = a.x[c1], x2 = a.x[c2], x3 = a.x[c3];
                if (OP == OP_JACOBI) {
                    if (c0 != r) s += v0 * x0;
                    if (c1 != r) s += v1 * x1;
                    if (c2 != r) s += v2 * x2;
                    if (c3 != r) s += v3 * x3;
                } else {
                    s += v0 * x0;
                    s += v1 * x1;
                    s += v2 * x2;
                    s += v3 * x3;
                }
            }
            if (k < ke) {
                // the last one to three strides in ONE round trip instead of one each: indices clamped into
                // the row, padded values replaced by 0 (same adds in the same order, then + 0)
                const int    kl = ke - 1;
                const int    k1 = min(k + L, kl), k2 = min(k + 2 * L, kl);
                const int    c0 = jb + ld_ja(a, k), c1 = jb + ld_ja(a, k1), c2 = jb + ld_ja(a, k2);
                const double v0 = ld_val(a, k);
                const double w1 = ld_val(a, k1), w2 = ld_val(a, k2);
                const double v1 = (k + L < ke) ? w1 : 0.0, v2 = (k + 2 * L < ke) ? w2 : 0.0;
                const double x0 = a.x[c0], x1 = a.x[c1], x2 = a.x[c2];
                if (OP == OP_JACOBI) {
                    if (c0 != r) s += v0 * x0;
                    if (c1 != r) s += v1 * x1;
                    if (c2 != r) s += v2 * x2;
                } else {
                    s += v0 * x0;
                    s += v1 * x1;
                    s += v2 * x2;
                }
            }
            s = subwave_sum<L>(s);
            if (sl == 0) {
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
                } else if (OP == OP_MXV_DOT) {
                    a.y[r] = s;
                    acc += s * a.dotv[r];
                }
            }
        }
    }
    if (OP == OP_MXV_DOT) {
        __shared__ double lds[4];
        const double tot = block_sum(acc, lds);
        if (threadIdx.x == 0) a.partials[blockIdx.x] = tot;
    }
}

constexpr int STREAM_MAXR = 1024;  // (tile bound of the retired block-stream kernel; pick_kernel still sizes tile_rows with it)
// ---------------------------------------------------------------------------
// Dictionary-coded CSR ("value indexing" + "delta units", Kourtis / Goumas / Koziris 2008):
// when a matrix holds at most 256 distinct (column - row base, value) pairs -- every level of
// a stencil problem that the coarsening keeps regular: P7(n) level 0 has 14 pairs, level 1 has
// 138 -- an entry is stored as ONE byte instead of 12 (int32 column + f64 value).  The
// dictionary holds the exact doubles, so this is lossless: row sums are formed left to right
// from the same values as the plain kernels and are bit-identical to them.  One lane per row:
// the k-th entries of 64 consecutive rows of a stencil matrix gather 64 consecutive x entries
// (coalesced), the dictionary sits in LDS (mostly broadcast reads).  Traffic per row drops from
// 12 nnz/row + 4 + 16 to nnz/row + 4 + 16 bytes.
// ---------------------------------------------------------------------------
//
// Row-pattern coding goes one step further (k_csr_rowpat below): when whole ROWS repeat -- the
// (offset, value) list of a row is one of at most 65 536 distinct lists, as for every interior
// / face / edge / corner row of a stencil (P7(n): 27 lists on level 0, 102 on level 1) -- a
// row is stored as ONE 16-bit pattern id; no row pointer, no per-entry code at all: 2 bytes of
// matrix per row instead of 12 nnz/row + 4.
//
// Schedule: a wavefront owns 64 consecutive rows per step.  Its code bytes form one contiguous
// span, fetched with 16-byte loads into registers ONE STEP AHEAD (together with the row
// pointers of the step after that), parked in the wave's LDS slab, and read back per entry;
// so the only memory latency exposed per step is that of the x gathers (U in flight per lane).
// No workgroup barrier inside the loop.
template <int OP, int U>
__global__ __launch_bounds__(BLOCK) void k_csr_dict8(CsrArgs a)
{
    if (a.stop && *a.stop) return;
    constexpr int CAP = 2048;  // code bytes staged per wave and step
    __shared__ double s_val[256];
    __shared__ int    s_off[256];
    __shared__ uint4  s_code[4][CAP / 16];
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    s_val[threadIdx.x] = a.dval[threadIdx.x];
    s_off[threadIdx.x] = a.doff[threadIdx.x];
    __syncthreads();
    uint4* slab = s_code[wave];
    const unsigned char* lds = reinterpret_cast<const unsigned char*>(slab);
    const int vmax = tile_vmax(a);
    const int G = gridDim.x;
    double dotacc = 0.0;

    // next step of this wave: first row r0 (-1: none) and row count, wave-uniform
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
    auto stage = [&](int k0, int k1, int& s0, uint4& q0, uint4& q1) {
        s0 = k0 & ~15;
        if (k1 - s0 > CAP) return;  // oversized span: entries are read from global memory instead
        const uint4* src = reinterpret_cast<const uint4*>(a.code + s0);
        const int nseg = (k1 - s0 + 15) >> 4;
        if (lane < nseg) q0 = src[lane];
        if (lane + 64 < nseg) q1 = src[lane + 64];
    };

    int v = blockIdx.x;
    int r0A, nrA, r0B, nrB, kbA = 0, keA = 0, kbB = 0, keB = 0;
    advance(v, r0A, nrA);
    if (r0A >= 0 && lane < nrA) { kbA = a.ia[r0A + lane]; keA = a.ia[r0A + lane + 1]; }
    advance(v, r0B, nrB);
    if (r0B >= 0 && lane < nrB) { kbB = a.ia[r0B + lane]; keB = a.ia[r0B + lane + 1]; }
    int k0A = 0, k1A = 0, s0A = 0;
    uint4 qA0 = make_uint4(0, 0, 0, 0), qA1 = make_uint4(0, 0, 0, 0);
    if (r0A >= 0) {
        k0A = __shfl(kbA, 0); k1A = __shfl(keA, nrA - 1);
        stage(k0A, k1A, s0A, qA0, qA1);
    }

    while (r0A >= 0) {
        const bool staged = (k1A - s0A) <= CAP;
        if (staged) { slab[lane] = qA0; slab[lane + 64] = qA1; }
        wave_lds_sync();
        // look ahead: row pointers of step C, code span of step B
        int r0C, nrC, kbC = 0, keC = 0;
        advance(v, r0C, nrC);
        if (r0C >= 0 && lane < nrC) { kbC = a.ia[r0C + lane]; keC = a.ia[r0C + lane + 1]; }
        int k0B = 0, k1B = 0, s0B = 0;
        uint4 qB0 = make_uint4(0, 0, 0, 0), qB1 = make_uint4(0, 0, 0, 0);
        if (r0B >= 0) {
            k0B = __shfl(kbB, 0); k1B = __shfl(keB, nrB - 1);
            stage(k0B, k1B, s0B, qB0, qB1);
        }

        if (lane < nrA) {
            const int r = r0A + lane;
            const int base = a.rowbase ? a.rowbase[r] : r;
            double acc = (OP == OP_JACOBI || OP == OP_L1DIAG) ? a.b[r] : 0.0;
            for (int k = kbA; k < keA; k += U) {
                unsigned c[U];
                double   xv[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    c[u] = (k + u < keA) ? (staged ? (unsigned)lds[k + u - s0A] : (unsigned)a.code[k + u]) : 0u;
#pragma unroll
                for (int u = 0; u < U; ++u) xv[u] = (k + u < keA) ? a.x[base + s_off[c[u]]] : 0.0;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (k + u < keA) {
                        const double pr = s_val[c[u]] * xv[u];
                        if (OP == OP_JACOBI) { if (base + s_off[c[u]] != r) acc -= pr; }
                        else if (OP == OP_L1DIAG) acc -= pr;
                        else acc += pr;
                    }
                }
            }
            const double s = acc;
            if (OP == OP_MXV) { a.y[r] = s; zx_store(a, r, s); }
            else if (OP == OP_RESID) a.y[r] = a.b[r] - s;
            else if (OP == OP_ADD) a.y[r] += s;
            else if (OP == OP_SUB) a.y[r] -= s;
            else if (OP == OP_AXPY) a.y[r] += s * a.alpha;
            else if (OP == OP_JACOBI) {
                const double d = a.diag[r], xi = a.x[r];
                a.y[r] = (fabs(d) > 1e-20) ? (1 - a.omega) * xi + a.omega * s / d : xi;
            } else if (OP == OP_L1DIAG) {
                const double d = a.diag[r], xi = a.x[r];
                a.y[r] = l1_or_jacobi_f(a, r, s, d, xi);
            } else if (OP == OP_MXV_DOT) {
                a.y[r] = s;
                dotacc += s * a.dotv[r];
            }
        }
        wave_lds_sync();
        r0A = r0B; nrA = nrB; kbA = kbB; keA = keB; k0A = k0B; k1A = k1B; s0A = s0B; qA0 = qB0; qA1 = qB1;
        r0B = r0C; nrB = nrC; kbB = kbC; keB = keC;
    }
    if (OP == OP_MXV_DOT) {
        const double tot = block_sum(dotacc, red);
        if (threadIdx.x == 0) a.partials[blockIdx.x] = tot;
    }
}

// Row-pattern-coded CSR: lane = row; the pattern table (entry lists of all distinct rows) sits
// in LDS when it is small (LDS_TAB), otherwise in global memory (L1/L2 resident); interior
// rows share one pattern, so table reads are broadcasts and the k-th gathers of 64 consecutive
// rows are coalesced.  The pattern id of the next step is fetched one step ahead.  Row sums
// are formed left to right from the exact stored values: bit-identical to the plain kernels.
// The loop body is branch-free: pattern lists are padded to multiples of 8 entries (offset 0,
// value 0), x is gathered with buffer loads (32-bit offsets, hardware range check), padding
// entries are dropped by a select on the accumulate, rows beyond the matrix are clamped and
// only their store is predicated.  RPL rows per lane and step (rows r, r + 256, ...).
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double buf_load_f64(__amdgpu_buffer_rsrc_t rs, unsigned byte_off)
{
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)byte_off, 0, 0);
    return __builtin_bit_cast(double, v);
}

// LDS_TAB: 0 table in global memory; 1 in LDS, up to 512 patterns / 2048 entries (28 KB: five blocks per CU);
// 2 in LDS, up to 64 patterns / 512 entries (6.5 KB: the register file bounds the occupancy, not the LDS)
template <int OP, int LDS_TAB, int RPL>
__global__ __launch_bounds__(BLOCK) void k_csr_rowpat(CsrArgs a)
{
    if (a.stop && *a.stop) return;
    constexpr int U = 8;
    constexpr int MAXP = LDS_TAB == 2 ? 64 : 512, MAXE = LDS_TAB == 2 ? 512 : 2048;
    __shared__ int    s_start[LDS_TAB ? MAXP : 1];
    __shared__ int    s_len[LDS_TAB ? MAXP : 1];
    __shared__ int    s_off[LDS_TAB ? MAXE : 1];
    __shared__ double s_val[LDS_TAB ? MAXE : 1];
    __shared__ double red[4];
    if (LDS_TAB) {
        for (int i = threadIdx.x; i < a.npat; i += BLOCK) { s_start[i] = a.pstart[i]; s_len[i] = a.plen[i]; }
        for (int i = threadIdx.x; i < a.npent; i += BLOCK) { s_off[i] = a.poff[i]; s_val[i] = a.pval[i]; }
        __syncthreads();
    }
    const int*    pstart = LDS_TAB ? s_start : a.pstart;
    const int*    plen   = LDS_TAB ? s_len : a.plen;
    const int*    poff   = LDS_TAB ? s_off : a.poff;
    const double* pval   = LDS_TAB ? s_val : a.pval;
    const __amdgpu_buffer_rsrc_t xr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.x), 0, (int)((unsigned)a.ncol * 8u), 0x00020000);
    const int vmax = tile_vmax(a);  // tiles of BLOCK * RPL rows
    const int G = gridDim.x;
    const int last = a.nrow - 1;
    double dotacc = 0.0;

    auto advance = [&](int& v) -> int {  // first row of the next tile of this block, -1: none
        for (;;) {
            if (v >= vmax) return -1;
            const int t = tile_of(a, v);
            v += G;
            if (t < a.ntiles) return (t + a.tile0) * (BLOCK * RPL);
        }
    };
    int      v = blockIdx.x;
    int      rA[RPL], cbA[RPL];
    unsigned pidA[RPL];
    int      r0A = advance(v);
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
        rA[q] = max(r0A, 0) + q * BLOCK + (int)threadIdx.x;
        const int rc = min(rA[q], last);
        pidA[q] = a.pat[rc];
        cbA[q] = a.rowbase ? a.rowbase[rc] : rc;
    }
    while (r0A >= 0) {
        const int r0B = advance(v);
        int      rB[RPL], cbB[RPL];
        unsigned pidB[RPL];
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            rB[q] = max(r0B, 0) + q * BLOCK + (int)threadIdx.x;
            const int rc = min(rB[q], last);
            pidB[q] = a.pat[rc];
            cbB[q] = a.rowbase ? a.rowbase[rc] : rc;
        }
        int    ps[RPL], len[RPL], self[RPL];
        double acc[RPL], dg[RPL], xi[RPL];
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            const int rc = min(rA[q], last);
            ps[q] = pstart[pidA[q]]; len[q] = plen[pidA[q]];
            self[q] = rc - cbA[q];
            acc[q] = dg[q] = xi[q] = 0.0;
            if (OP == OP_JACOBI || OP == OP_L1DIAG) { acc[q] = a.b[rc]; dg[q] = a.diag[rc]; xi[q] = a.x[rc]; }
        }
        int kmax = len[0];
#pragma unroll
        for (int q = 1; q < RPL; ++q) kmax = max(kmax, len[q]);
        for (int k = 0; k < kmax; k += U) {
            int    off[RPL][U];
            double xv[RPL][U];
#pragma unroll
            for (int q = 0; q < RPL; ++q)
#pragma unroll
                for (int u = 0; u < U; ++u) off[q][u] = (k < len[q]) ? poff[ps[q] + k + u] : 0;
#pragma unroll
            for (int q = 0; q < RPL; ++q)
#pragma unroll
                for (int u = 0; u < U; ++u) xv[q][u] = buf_load_f64(xr, (unsigned)(cbA[q] + off[q][u]) * 8u);
#pragma unroll
            for (int q = 0; q < RPL; ++q)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const double pr = ((k < len[q]) ? pval[ps[q] + k + u] : 0.0) * xv[q][u];
                    bool use = k + u < len[q];
                    if (OP == OP_JACOBI) use = use && off[q][u] != self[q];
                    const double nxt = (OP == OP_JACOBI || OP == OP_L1DIAG) ? acc[q] - pr : acc[q] + pr;
                    acc[q] = use ? nxt : acc[q];
                }
        }
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            if (rA[q] > last) continue;
            const int    r = rA[q];
            const double s = acc[q];
            if (OP == OP_MXV) { a.y[r] = s; zx_store(a, r, s); }
            else if (OP == OP_RESID) a.y[r] = a.b[r] - s;
            else if (OP == OP_ADD) a.y[r] += s;
            else if (OP == OP_SUB) a.y[r] -= s;
            else if (OP == OP_AXPY) a.y[r] += s * a.alpha;
            else if (OP == OP_JACOBI) a.y[r] = (fabs(dg[q]) > 1e-20) ? (1 - a.omega) * xi[q] + a.omega * s / dg[q] : xi[q];
            else if (OP == OP_L1DIAG) a.y[r] = l1_or_jacobi_f(a, r, s, dg[q], xi[q]);
            else if (OP == OP_MXV_DOT) {
                a.y[r] = s;
                dotacc += s * a.dotv[r];
            }
        }
        r0A = r0B;
#pragma unroll
        for (int q = 0; q < RPL; ++q) { rA[q] = rB[q]; pidA[q] = pidB[q]; cbA[q] = cbB[q]; }
    }
    if (OP == OP_MXV_DOT) {
        const double tot = block_sum(dotacc, red);
        if (threadIdx.x == 0) a.partials[blockIdx.x] = tot;
    }
}

// ---------------------------------------------------------------------------
// Wavefront-level stream kernel: the same two phases as k_csr_stream, but every
// wavefront owns its own tile of RW consecutive rows and its own LDS slab, so there is
// no workgroup barrier anywhere: a wave's LDS traffic is ordered by the hardware, the
// four waves of a block run fully decoupled and each keeps 8 (JA, val) pairs + 8 x
// gathers in flight per lane.  Row sums are again the reference's left-to-right sums.
// ---------------------------------------------------------------------------

template <int OP, int RW, int CAPW>
__global__ __launch_bounds__(BLOCK) void k_csr_wstream(CsrArgs a)
{
    if (a.stop && *a.stop) return;
    // Jacobi skips the diagonal entry by its storage index (a.dpos), so no column staging
    __shared__ double prod_all[4 * CAPW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* prod = prod_all + wave * CAPW;

    const int vmax = tile_vmax(a);
    double dotacc = 0.0;

    for (int v = blockIdx.x; v < vmax; v += gridDim.x) {
        const int t = tile_of(a, v);
        if (t >= a.ntiles) continue;
        const int r0 = ((t + a.tile0) * 4 + wave) * RW;
        if (r0 >= a.nrow) continue;  // wave-uniform
        const int nr = min(RW, a.nrow - r0);
        int kb = 0, ke = 0;
        if (lane < nr) {
            kb = a.ia[r0 + lane];
            ke = a.ia[r0 + lane + 1];
        }
        const int k0 = __shfl(kb, 0), k1 = __shfl(ke, nr - 1);
        const int r = r0 + lane;
        double acc = ((OP == OP_JACOBI || OP == OP_L1DIAG) && lane < nr) ? a.b[r] : 0.0;
        const int dk = (OP == OP_JACOBI && lane < nr) ? a.dpos[r] : -1;

        for (int lo = k0; lo < k1; lo += CAPW) {
            const int hi = min(lo + CAPW, k1);
            for (int base = lo; base < hi; base += 512) {
                int    c[8];
                double v[8], xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = base + lane + 64 * u;
                    const bool ok = k < hi;
                    c[u] = ok ? ld_ja(a, k) : 0;
                    v[u] = ok ? ld_val(a, k) : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) xv[u] = a.x[c[u]];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = base + lane + 64 * u;
                    if (k < hi) {
                        prod[k - lo] = v[u] * xv[u];

                    }
                }
            }
            wave_lds_sync();
            if (lane < nr) {
                const int pb = max(kb, lo), pe = min(ke, hi);
                if (OP == OP_JACOBI) {
                    for (int k = pb; k < pe; ++k)
                        if (k != dk) acc -= prod[k - lo];
                } else if (OP == OP_L1DIAG) {
                    for (int k = pb; k < pe; ++k) acc -= prod[k - lo];
                } else {
                    for (int k = pb; k < pe; ++k) acc += prod[k - lo];
                }
            }
            wave_lds_sync();
        }

        if (lane < nr) {
            const double s = acc;
            if (OP == OP_MXV) { a.y[r] = s; zx_store(a, r, s); }
            else if (OP == OP_RESID) a.y[r] = a.b[r] - s;
            else if (OP == OP_ADD) a.y[r] += s;
            else if (OP == OP_SUB) a.y[r] -= s;
            else if (OP == OP_AXPY) a.y[r] += s * a.alpha;
            else if (OP == OP_JACOBI) {
                const double d = a.diag[r], xi = a.x[r];
                a.y[r] = (fabs(d) > 1e-20) ? (1 - a.omega) * xi + a.omega * s / d : xi;
            } else if (OP == OP_L1DIAG) {
                const double d = a.diag[r], xi = a.x[r];
                a.y[r] = l1_or_jacobi_f(a, r, s, d, xi);
            } else if (OP == OP_MXV_DOT) {
                a.y[r] = s;
                dotacc += s * a.dotv[r];
            }
        }
    }
    if (OP == OP_MXV_DOT) {
        const double tot = block_sum(dotacc, prod_all);  // block_sum barriers before writing
        if (threadIdx.x == 0) a.partials[blockIdx.x] = tot;
    }
}

// ---------------------------------------------------------------------------
// BSR (block CSR, nb x nb row-major blocks) row kernel, wave-level stream design: a wavefront
// owns 64/NB consecutive block rows; phase 1 sweeps the tile's contiguous span of block
// values into the wave's LDS slab with unit-stride loads; phase 2 lane (block row, r) walks
// the blocks of its row in storage order and adds (A_r0 x_0 + A_r1 x_1 + ...) -- inner sum
// first, as fasp_blas_smat_ypAx (BlaSmallMat.c:779) -- so results are bit-identical to
// fasp_blas_dbsr_mxv / _aAxpy / fasp_smoother_dbsr_jacobi1 for nb <= 7.
//   OP 0: y = A x     OP 1: y = alpha ((1/alpha) y + A x)     OP 2: u' = Dinv (b - sum_{j != i} A_ij u_j)
// ---------------------------------------------------------------------------
struct BsrArgs {
    int           ROW;
    const int*    ia;
    const int*    ja;
    const double* val;
    const double* x;     // gathered vector (OP 2: the old iterate u)
    double*       y;     // output
    const double* b;     // OP 2: right-hand side
    const double* dinv;  // OP 2: inverse diagonal blocks
    double        alpha; // OP 1
    int           ntiles;
};

typedef double bsr_f64x2_u __attribute__((ext_vector_type(2), aligned(8)));    // a 16-byte global load at the alignment of its elements
typedef double bsr_f64x2_t __attribute__((ext_vector_type(2), aligned(16)));   // a 16-byte LDS store
template <int NB, int OP>
__global__ __launch_bounds__(BLOCK) void k_bsr_wstream(BsrArgs a)
{
    constexpr int NB2 = NB * NB;
    constexpr int RW = 64 / NB;                 // block rows per wave tile
    constexpr int CAPB = 1536 / NB2;            // blocks per LDS chunk: 12 KiB per wave, three workgroups per CU (2048: two, measured 0.49 of peak against 0.58 on P7(128) x B3; 1024 splits the tiles of 7-block rows in two: 0.40)
    constexpr int SLAB = (CAPB * NB2 + 3) & ~1;   // (even, and one spare pair: 16-byte stores of an odd-length chunk)
    __shared__ __attribute__((aligned(16))) double lds_all[4 * SLAB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* lds = lds_all + wave * SLAB;
    const int lb = lane / NB, r = lane - lb * NB;

    for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
        const int br0 = (t * 4 + wave) * RW;
        if (br0 >= a.ROW) continue;
        const int nbr = min(RW, a.ROW - br0);
        const bool act = lb < nbr;
        const int br = br0 + lb;
        int kb = 0, ke = 0;
        if (act) { kb = a.ia[br]; ke = a.ia[br + 1]; }
        const int k0 = __shfl(kb, 0), k1 = __shfl(ke, (nbr - 1) * NB);
        const size_t row = (size_t)br * NB + r;
        double acc = 0.0;
        if (act) {
            if (OP == 1) {  // y0 taken from b when given (residual w = b - A x without a copy of b)
                const double y0 = a.b ? a.b[row] : a.y[row];
                acc = (a.alpha != 1.0) ? y0 * (1.0 / a.alpha) : y0;
            }
            if (OP == 2) acc = a.b[row];
        }
        // Round 4: the value stream -- 8 nb^2 of the 8 nb^2 + 4 bytes of a block -- travels as 16-byte loads (a 64-lane load costs the
        // address unit the same whatever the bytes per lane: kernels2.hip.h), and the NEXT chunk of the tile is in flight, in
        // registers, while this one is consumed from LDS.  (Block values are 8-byte aligned: global memory takes 16-byte loads
        // there; chunk sizes are even numbers of doubles or end the tile.)
        constexpr int NV = (CAPB * NB2 + 127) / 128;   // 16-byte loads per lane and chunk
        bsr_f64x2_u pv[NV];
        auto fetch = [&](int lo) {
            const int hi = min(lo + CAPB, k1);
            const int ne = (hi - lo) * NB2;
            const double* src = a.val + (size_t)lo * NB2;
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int e = 2 * (lane + 64 * u);
                bsr_f64x2_u v;
                v[0] = 0.0; v[1] = 0.0;
                if (e + 1 < ne) v = __builtin_nontemporal_load(reinterpret_cast<const bsr_f64x2_u*>(src + e));
                else if (e < ne) v[0] = __builtin_nontemporal_load(src + e);
                pv[u] = v;
            }
        };
        if (k0 < k1) fetch(k0);
        for (int lo = k0; lo < k1; lo += CAPB) {
            const int hi = min(lo + CAPB, k1);
            const int ne = (hi - lo) * NB2;
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int e = 2 * (lane + 64 * u);
                if (e < ne) { bsr_f64x2_t w; w[0] = pv[u][0]; w[1] = pv[u][1]; *reinterpret_cast<bsr_f64x2_t*>(lds + e) = w; }   // (the slab has room for the odd last double's neighbour)
            }
            if (lo + CAPB < k1) fetch(lo + CAPB);
            wave_lds_sync();
            if (act) {
                // four (nb <= 3: eight) blocks per round trip: their column indices first, then the entries of x they name, then the
                // products in storage order (one block at a time -- index, then x, then the next index -- made a row of
                // seven blocks a chain of fourteen dependent memory latencies)
                const int pb = max(kb, lo), pe = min(ke, hi);
                constexpr int U = NB <= 3 ? 8 : 4;
                for (int k = pb; k < pe; k += U) {
                    int    j[U];
                    double xv[U][NB];
#pragma unroll
                    for (int u = 0; u < U; ++u) j[u] = a.ja[min(k + u, pe - 1)];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const double* xb = a.x + (size_t)j[u] * NB;
#pragma unroll
                        for (int c = 0; c < NB; ++c) xv[u][c] = xb[c];
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (k + u >= pe || (OP == 2 && j[u] == br)) continue;
                        const double* A = lds + (k + u - lo) * NB2 + r * NB;
                        double s = A[0] * xv[u][0];
#pragma unroll
                        for (int c = 1; c < NB; ++c) s = s + A[c] * xv[u][c];
                        if (OP == 2) acc -= s; else acc += s;
                    }
                }
            }
            wave_lds_sync();
        }
        if (OP == 2) {
            // u_i = Dinv_i * bt_i (fasp_blas_smat_mxv): bt lives in the NB lanes of the block row
            double out = 0.0;
            const double* D = a.dinv + (size_t)(act ? br : 0) * NB2 + r * NB;
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                const double btc = __shfl(acc, lb * NB + c);
                if (act) out = (c == 0) ? D[0] * btc : out + D[c] * btc;
            }
            if (act) a.y[row] = out;
        } else if (act) {
            a.y[row] = (OP == 1 && a.alpha != 1.0) ? acc * a.alpha : acc;
        }
    }
}

// fasp_precond_diag (PreCSR.c:172): z = r, then z_i /= d_i where |d_i| > SMALLREAL
__global__ __launch_bounds__(BLOCK) void k_diag_precond(int n, const double* __restrict__ d, const double* __restrict__ r,
                                                         double* __restrict__ z)
{
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const double di = d[i], ri = r[i];
        z[i] = (fabs(di) > 1e-20) ? ri / di : ri;
    }
}

// Block-Jacobi sweep from a zero iterate: u = Dinv b (the off-diagonal sum vanishes exactly)
__global__ __launch_bounds__(BLOCK) void k_bsr_dinv_apply(int n, int nb, const double* __restrict__ dinv,
                                                           const double* __restrict__ b, double* __restrict__ u)
{
    for (int row = blockIdx.x * BLOCK + threadIdx.x; row < n; row += gridDim.x * BLOCK) {
        const int br = row / nb, r = row - br * nb;
        const double* D = dinv + (size_t)br * nb * nb + r * nb;
        const double* bb = b + (size_t)br * nb;
        double s = D[0] * bb[0];
        for (int c = 1; c < nb; ++c) s = s + D[c] * bb[c];
        u[row] = s;
    }
}

// ---------------------------------------------------------------------------
// Sequential sweeps (Gauss-Seidel / SOR family, ItrSmootherCSR.c:251-1040) by level
// scheduling: the host groups the rows of a sweep into the levels of its dependency DAG
// (a row depends on every coupled row that comes earlier in the sweep order), one launch
// per DAG level updates all rows of that level in place; rows of a level are mutually
// uncoupled, so the result equals the sequential sweep (row sums in a fixed tree order).
//   form 0  u_i = t * (1/a_ii)                 fasp_smoother_dcsr_gs      :327-334
//   form 1  u_i = t / a_ii                     _gs_cf :572-693, _sgs :845-880
//   form 2  u_i = w (t / a_ii) + (1-w) u_i     _sor :981-993
// with t = b_i - sum_{j != i} a_ij u_j.
// ---------------------------------------------------------------------------
// Off-diagonal part of one row of a sequential sweep, entries k, k + L, k + 2L, ... of the calling lane, added in that
// order: four (JA, val) pairs and four gathers of u in flight per round trip (rows of the deep levels hold hundreds
// to thousands of entries; one entry per round trip made a dependency level of three such rows cost 15-20 us), the
// last one to three strides in one more round trip with clamped indices.  ldu(c) reads u_c.
template <int L, class LDU>
__device__ __forceinline__ double seq_row_sum(const int* __restrict__ ja, const double* __restrict__ val, int k, int ke,
                                              int r, LDU ldu, double s = 0.0)
{
    for (; k + 3 * L < ke; k += 4 * L) {
        const int    c0 = ja[k], c1 = ja[k + L], c2 = ja[k + 2 * L], c3 = ja[k + 3 * L];
        const double v0 = val[k], v1 = val[k + L], v2 = val[k + 2 * L], v3 = val[k + 3 * L];
        const double u0 = ldu(c0), u1 = ldu(c1), u2 = ldu(c2), u3 = ldu(c3);
        if (c0 != r) s += v0 * u0;
        if (c1 != r) s += v1 * u1;
        if (c2 != r) s += v2 * u2;
        if (c3 != r) s += v3 * u3;
    }
    if (k < ke) {
        const int    kl = ke - 1;
        const int    k1 = min(k + L, kl), k2 = min(k + 2 * L, kl);
        const int    c0 = ja[k], c1 = ja[k1], c2 = ja[k2];
        const double v0 = val[k], v1 = val[k1], v2 = val[k2];
        const double u0 = ldu(c0), u1 = ldu(c1), u2 = ldu(c2);
        if (c0 != r) s += v0 * u0;
        if (k + L < ke && c1 != r) s += v1 * u1;
        if (k + 2 * L < ke && c2 != r) s += v2 * u2;
    }
    return s;
}

template <int L>
__global__ __launch_bounds__(BLOCK) void k_seq_level(const int* __restrict__ order, int lo, int hi,
                                                      const int* __restrict__ ia, const int* __restrict__ ja,
                                                      const double* __restrict__ val,
                                                      const double* __restrict__ b,
                                                      const double* __restrict__ diag, double* u, int form,
                                                      double w)
{
    constexpr int RPB = BLOCK / L;
    const int sl = threadIdx.x & (L - 1);
    const int rloc = threadIdx.x / L;
    for (int idx = lo + blockIdx.x * RPB + rloc; idx < hi; idx += gridDim.x * RPB) {
        const int r = order[idx];
        const int kb = ia[r], ke = ia[r + 1];
        double s = seq_row_sum<L>(ja, val, kb + sl, ke, r, [&](int c) { return u[c]; });
        s = subwave_sum<L>(s);
        if (sl == 0) {
            const double d = diag[r];
            const double t = b[r] - s;
            if (fabs(d) > 1e-20) {
                if (form == 0) u[r] = t * (1.0 / d);
                else if (form == 1) u[r] = t / d;
                else u[r] = w * (t / d) + (1 - w) * u[r];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Block rows of ONE dependency level of a sequential block sweep (fasp_smoother_dbsr_gs_ascend /
// _descend, ItrSmootherBSR.c:552 / :683; _sor_ascend / _descend, :1115 / :1234): one thread per block
// row, blocks in storage order, b_tmp -= (A_r0 u_0 + A_r1 u_1 + ...) per block (fasp_blas_smat_ymAx),
// then u_i = Dinv_i b_tmp (fasp_blas_smat_mxv) or the SOR update in the operation order of
// fasp_blas_smat_aAxpby (BlaSmallMat.c:1140).  A parity mode, not a bandwidth kernel.
// ---------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(BLOCK) void k_bsr_seq_level(const int* __restrict__ order, int lo, int hi,
                                                          const int* __restrict__ ia, const int* __restrict__ ja,
                                                          const double* __restrict__ val, const double* __restrict__ b,
                                                          const double* __restrict__ dinv, double* u, int sor, double w)
{
    constexpr int NB2 = NB * NB;
    for (int idx = lo + blockIdx.x * BLOCK + threadIdx.x; idx < hi; idx += gridDim.x * BLOCK) {
        const int i = order[idx];
        double bt[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) bt[r] = b[(size_t)i * NB + r];
        for (int k = ia[i]; k < ia[i + 1]; ++k) {
            const int j = ja[k];
            if (j == i) continue;
            const double* A = val + (size_t)k * NB2;
            double x[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) x[c] = u[(size_t)j * NB + c];
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                double s = A[r * NB] * x[0];
#pragma unroll
                for (int c = 1; c < NB; ++c) s = s + A[r * NB + c] * x[c];
                bt[r] -= s;
            }
        }
        const double* D = dinv + (size_t)i * NB2;
        double* y = u + (size_t)i * NB;
        if (!sor) {
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                double s = D[r * NB] * bt[0];
#pragma unroll
                for (int c = 1; c < NB; ++c) s = s + D[r * NB + c] * bt[c];
                y[r] = s;
            }
        } else if (NB == 1) {
            y[0] = (1.0 - w) * y[0] + w * (bt[0] * D[0]);   // ItrSmootherBSR.c:1166
        } else {
            const double omw = 1.0 - w;
            if (w == 0) {
#pragma unroll
                for (int r = 0; r < NB; ++r) y[r] *= omw;
            } else {
                const double tmp = omw / w;
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    double yr = y[r];
                    if (tmp != 1.0) yr *= tmp;
#pragma unroll
                    for (int c = 0; c < NB; ++c) yr += D[r * NB + c] * bt[c];
                    if (w != 1.0) yr *= w;
                    y[r] = yr;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// BLAS-1.  n is a double count; vectors come from hipMalloc (256-B aligned) so the
// double2 path is always aligned; the odd tail element is handled by thread 0 of the
// last block.  Partials layout: partials[q * gridDim.x + blockIdx.x].
// ---------------------------------------------------------------------------
struct Vec2 { double x, y; };

// y += a*x   (a == +-1 round identically to the reference's y += x / y -= x branches)
// ---------------------------------------------------------------------------
// polynomial smoother (ItrSmootherCSRpoly.c:551 Rr): the elementwise steps between its SpMVs
// ---------------------------------------------------------------------------
// rbar = Dinv r
__global__ __launch_bounds__(BLOCK) void k_poly_scale(int n, const double* __restrict__ dinv, const double* __restrict__ r,
                                                       double* __restrict__ rbar)
{
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) rbar[i] = dinv[i] * r[i];
}
// v1 = Dinv v1; v0 = k1 rbar; v1 = k2 rbar - k3 v1
__global__ __launch_bounds__(BLOCK) void k_poly_start(int n, double k1, double k2, double k3, const double* __restrict__ dinv,
                                                       const double* __restrict__ rbar, double* __restrict__ v0,
                                                       double* __restrict__ v1)
{
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const double w = dinv[i] * v1[i];
        v0[i] = k1 * rbar[i];
        v1[i] = k2 * rbar[i] - k3 * w;
    }
}
// rbar = (r - A v1) Dinv (rbar holds A v1 on entry); vnew = v1 + k5 (v1 - v0) + k4 rbar; v0 = v1; v1 = vnew
__global__ __launch_bounds__(BLOCK) void k_poly_step(int n, double k4, double k5, const double* __restrict__ dinv,
                                                      const double* __restrict__ r, double* __restrict__ rbar,
                                                      double* __restrict__ v0, double* __restrict__ v1,
                                                      double* __restrict__ vnew)
{
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const double rb = (r[i] - rbar[i]) * dinv[i];
        const double a1 = v1[i];
        const double nw = a1 + k5 * (a1 - v0[i]) + k4 * rb;
        rbar[i] = rb; vnew[i] = nw; v0[i] = a1; v1[i] = nw;
    }
}

__global__ __launch_bounds__(BLOCK) void k_axpy(int n, double a, const double* __restrict__ x,
                                                 double* __restrict__ y)
{
    const int n2 = n >> 1;
    const double2* x2 = reinterpret_cast<const double2*>(x);
    double2*       y2 = reinterpret_cast<double2*>(y);
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n2; i += gridDim.x * BLOCK) {
        double2 xv = x2[i], yv = y2[i];
        yv.x += a * xv.x;
        yv.y += a * xv.y;
        y2[i] = yv;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] += a * x[n - 1];
}

// y = a*x + b*y   (BlaArray.c:644)
__global__ __launch_bounds__(BLOCK) void k_axpby(int n, double a, const double* __restrict__ x,
                                                  double b, double* __restrict__ y)
{
    const int n2 = n >> 1;
    const double2* x2 = reinterpret_cast<const double2*>(x);
    double2*       y2 = reinterpret_cast<double2*>(y);
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n2; i += gridDim.x * BLOCK) {
        double2 xv = x2[i], yv = y2[i];
        yv.x = a * xv.x + b * yv.x;
        yv.y = a * xv.y + b * yv.y;
        y2[i] = yv;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = a * x[n - 1] + b * y[n - 1];
}

// p = z + beta p with beta = (*num) / (*den) formed on the device (the host's `beta = temp2 / temp1`, KryPcg.c:340-343: one IEEE division
// either way): the (z, r) of this and of the previous iteration never travel to the host in between (pcg.hip.h)
// npart > 0: the numerator is still npart per-block partials -- every block sums them in k_finalize's order (thread i takes i, i + 256, ...;
// then the block tree: the same bits in every block), block 0 leaves the sum in *num_out for the next k_cg_update and for the host
__global__ __launch_bounds__(BLOCK) void k_axpby_beta(int n, const double* __restrict__ x, const double* __restrict__ num,
                                                       const double* __restrict__ den, double* __restrict__ y,
                                                       const double* __restrict__ partials = nullptr, int npart = 0,
                                                       double* __restrict__ num_out = nullptr)
{
    double nu;
    if (npart > 0) {
        __shared__ double lds[4];
        __shared__ double bcast;
        double sacc = 0.0;
        for (int i = threadIdx.x; i < npart; i += BLOCK) sacc += partials[i];
        sacc = block_sum(sacc, lds);
        if (threadIdx.x == 0) { bcast = sacc; if (blockIdx.x == 0) *num_out = sacc; }
        __syncthreads();
        nu = bcast;
    } else {
        nu = *num;
    }
    const double b = nu / (*den);
    const int n2 = n >> 1;
    const double2* x2 = reinterpret_cast<const double2*>(x);
    double2*       y2 = reinterpret_cast<double2*>(y);
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n2; i += gridDim.x * BLOCK) {
        const double2 xv = x2[i];
        double2       yv = y2[i];
        yv.x = 1.0 * xv.x + b * yv.x;
        yv.y = 1.0 * xv.y + b * yv.y;
        y2[i] = yv;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = 1.0 * x[n - 1] + b * y[n - 1];
}

// first Jacobi sweep from a zero initial guess: (1-w)*0 + w*b_i/d_i  ==  (w*b_i)/d_i exactly
// (ItrSmootherCSR.c:148-170 with u == 0: t_i = b_i, no matrix pass needed)
__global__ __launch_bounds__(BLOCK) void k_jacobi_zero(int n, double w, const double* __restrict__ b,
                                                        const double* __restrict__ d,
                                                        double* __restrict__ x)
{
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const double di = d[i];
        x[i] = (fabs(di) > 1e-20) ? (1 - w) * 0.0 + w * b[i] / di : 0.0;
    }
}
// first L1-diag sweep from zero: x_i = 0 + b_i / l1_i
__global__ __launch_bounds__(BLOCK) void k_l1_zero(int n, const double* __restrict__ b,
                                                    const double* __restrict__ d,
                                                    double* __restrict__ x)
{
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const double di = d[i];
        x[i] = (fabs(di) > 1e-20) ? 0.0 + b[i] / di : 0.0;
    }
}

// partial sums of x_i*y_i
__global__ __launch_bounds__(BLOCK) void k_dot(int n, const double* __restrict__ x,
                                                const double* __restrict__ y,
                                                double* __restrict__ partials)
{
    __shared__ double lds[4];
    double acc = 0.0;
    const int n2 = n >> 1;
    const double2* x2 = reinterpret_cast<const double2*>(x);
    const double2* y2 = reinterpret_cast<const double2*>(y);
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n2; i += gridDim.x * BLOCK) {
        const double2 xv = x2[i], yv = y2[i];
        acc += xv.x * yv.x;
        acc += xv.y * yv.y;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) acc += x[n - 1] * y[n - 1];
    const double tot = block_sum(acc, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// Krylov update fused with the norms the control flow needs (KryPcg.c:180-189,
// KrySPcg.c:147-199):  alpha = temp1 / (t,p);  u += alpha p;  r -= alpha t;
// partial sums q0 = r.r, q1 = u.u, q2 = p.p, q3 = max|u| (max), q4 = #NaN in u.
// (t,p) comes either reduced from *red_tp (distributed levels: finalize + all-reduce), or
// -- ntp > 0 -- as the ntp per-block partials of the producing SpMV, which every block sums
// in the same fixed order (no separate finalize launch on the replicated coarse level).
// If |(t,p)| <= 1e-40 (CG breakdown) nothing is updated; the host sees (t,p) itself.
__global__ __launch_bounds__(BLOCK) void k_cg_update(int n, double temp1, const double* __restrict__ red_tp,
                                                      const double* __restrict__ tp_partials, int ntp,
                                                      const double* __restrict__ p,
                                                      const double* __restrict__ t,
                                                      double* __restrict__ u, double* __restrict__ r,
                                                      double* __restrict__ partials, int full_norms,
                                                      double* __restrict__ temp2_out,
                                                      double* __restrict__ zx = nullptr,
                                                      const double* __restrict__ zdiag = nullptr, double zomega = 0.0,
                                                      const double* __restrict__ temp1_dev = nullptr, double zdiag_value = 0.0)
{
    // zx != nullptr: r is about to be the right-hand side of a preconditioner whose first step is a Jacobi sweep
    // from zero -- written here, zx_i = (w r_i) / zdiag_i (k_jacobi_zero's expression), instead of re-reading r
    // temp1_dev != nullptr: (z, r) stayed on the device (pcg.hip.h: no host round trip between the cycle and the next t = A p)
    if (temp1_dev) temp1 = *temp1_dev;
    __shared__ double lds[5][4];
    __shared__ double bcast;
    double temp2;
    if (ntp > 0) {
        double sacc = 0.0;
        for (int i = threadIdx.x; i < ntp; i += BLOCK) sacc += tp_partials[i];
        sacc = block_sum(sacc, lds[0]);
        if (threadIdx.x == 0) bcast = sacc;
        __syncthreads();
        temp2 = bcast;
        if (temp2_out && blockIdx.x == 0 && threadIdx.x == 0) *temp2_out = temp2;
    } else {
        temp2 = *red_tp;
    }
    const bool   ok = fabs(temp2) > 1e-40;
    const double alpha = ok ? temp1 / temp2 : 0.0;
    double q[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (ok) {
        for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
            const double pi = p[i];
            const double ui = u[i] + alpha * pi;
            const double ri = r[i] + (-alpha) * t[i];
            u[i] = ui;
            r[i] = ri;
            if (zx) {
                const double di = zdiag ? zdiag[i] : zdiag_value;   // (zdiag == nullptr: the same diagonal entry in every row)
                zx[i] = (fabs(di) > 1e-20) ? (1 - zomega) * 0.0 + zomega * ri / di : 0.0;
            }
            q[0] += ri * ri;
            if (full_norms) {
                q[1] += ui * ui;
                q[2] += pi * pi;
                q[3] = fmax(q[3], fabs(ui));
                q[4] += (ui != ui) ? 1.0 : 0.0;
            }
        }
    }
    const int nq = full_norms ? 5 : 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    for (int k = 0; k < nq; ++k) {
        double v = q[k];
        if (k == 3) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
        } else {
            v = subwave_sum<64>(v);
        }
        if (lane == 0) lds[k][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < nq) {
        const int k = threadIdx.x;
        const double v = (k == 3) ? fmax(fmax(lds[k][0], lds[k][1]), fmax(lds[k][2], lds[k][3]))
                                  : ((lds[k][0] + lds[k][1]) + lds[k][2]) + lds[k][3];
        partials[k * gridDim.x + blockIdx.x] = v;
    }
}

// One modified-Gram-Schmidt step of GMRES (KryPvgmres.c:228-231) fused with the NEXT
// reduction:  p_i += (-h) p_j  with h = *h_ptr (the dot product produced on the device by
// the previous step), then the partial sums of (p_next, p_i) -- or of (p_i, p_i) when
// p_next == nullptr (the norm that ends the orthogonalisation, :232).  The whole chain of
// i dependent dot -> axpy pairs runs without a host round trip.
// npart > 0: h is still npart per-block partials of the previous step (in another buffer than `partials`) -- every block sums them in
// k_finalize's order (thread i takes i, i + 256, ...; then the block tree: the same bits in every block), block 0 leaves the sum in
// *h_out for the host's Hessenberg column: one launch per Gram-Schmidt step instead of two (k_axpby_beta's scheme)
__global__ __launch_bounds__(BLOCK) void k_mgs_step(int n, const double* __restrict__ h_ptr,
                                                     const double* __restrict__ pj, double* __restrict__ pi,
                                                     const double* __restrict__ pnext,
                                                     double* __restrict__ partials,
                                                     const double* __restrict__ prev_partials = nullptr, int npart = 0,
                                                     double* __restrict__ h_out = nullptr)
{
    __shared__ double lds[4];
    double h;
    if (npart > 0) {
        __shared__ double bcast;
        double sacc = 0.0;
        for (int i = threadIdx.x; i < npart; i += BLOCK) sacc += prev_partials[i];
        sacc = block_sum(sacc, lds);
        if (threadIdx.x == 0) { bcast = sacc; if (blockIdx.x == 0) *h_out = sacc; }
        __syncthreads();
        h = bcast;
    } else {
        h = *h_ptr;
    }
    const double a = -h;
    double acc = 0.0;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const double v = pi[i] + a * pj[i];
        pi[i] = v;
        acc += (pnext ? pnext[i] : v) * v;
    }
    const double tot = block_sum(acc, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// x *= a  (fasp_blas_darray_ax, BlaArray.c:43; the caller skips a == 1)
__global__ __launch_bounds__(BLOCK) void k_scale(int n, double a, double* __restrict__ x)
{
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) x[i] *= a;
}
// x = v  (fasp_darray_set, AuxArray.c:41)
__global__ __launch_bounds__(BLOCK) void k_set(int n, double v, double* __restrict__ x)
{
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) x[i] = v;
}
// z = a*x + y  (fasp_blas_darray_axpyz, BlaArray.c:403; z may alias x or y)
__global__ __launch_bounds__(BLOCK) void k_axpyz(int n, double a, const double* x, const double* y, double* z)
{
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) z[i] = a * x[i] + y[i];
}
// q0 = sum |x_i| (fasp_blas_darray_norm1, BlaArray.c:663), q1 = number of NaN entries (fasp_dvec_isnan, AuxVector.c:39)
__global__ __launch_bounds__(BLOCK) void k_norm1_nan(int n, const double* __restrict__ x, double* __restrict__ partials)
{
    __shared__ double lds[4];
    double s = 0.0, nn = 0.0;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const double v = x[i];
        s += fabs(v);
        nn += (v != v) ? 1.0 : 0.0;
    }
    const int G = gridDim.x;
    double v = block_sum(s, lds);
    if (threadIdx.x == 0) partials[0 * G + blockIdx.x] = v;
    v = block_sum(nn, lds);
    if (threadIdx.x == 0) partials[1 * G + blockIdx.x] = v;
}
// y += a*y  (fasp_blas_darray_axpy called with x == y, KryPvgmres.c:395,398)
__global__ __launch_bounds__(BLOCK) void k_axpy_self(int n, double a, double* __restrict__ y)
{
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const double v = y[i];
        y[i] = v + a * v;
    }
}

// norms of one vector: q0 = x.x, q1 = max|x|
__global__ __launch_bounds__(BLOCK) void k_norms(int n, const double* __restrict__ x,
                                                  double* __restrict__ partials)
{
    __shared__ double lds[4];
    double ss = 0.0, mm = 0.0;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const double v = x[i];
        ss += v * v;
        mm = fmax(mm, fabs(v));
    }
    const int G = gridDim.x;
    double v = block_sum(ss, lds);
    if (threadIdx.x == 0) partials[0 * G + blockIdx.x] = v;
    v = block_max(mm, lds);
    if (threadIdx.x == 0) partials[1 * G + blockIdx.x] = v;
}

// Final stage of every reduction: ONE block sums G partials per quantity in a fixed
// order (thread i takes i, i+256, ...; then the block tree) and writes out[q].
// Quantities with bit q set in maxmask are max-reduced instead of summed.
__global__ __launch_bounds__(BLOCK) void k_finalize(const double* __restrict__ partials, int G,
                                                     int nq, unsigned maxmask,
                                                     double* __restrict__ out)
{
    __shared__ double lds[4];
    for (int q = 0; q < nq; ++q) {
        const double* p = partials + (size_t)q * G;
        if ((maxmask >> q) & 1u) {
            double m = 0.0;
            for (int i = threadIdx.x; i < G; i += BLOCK) m = fmax(m, p[i]);
            m = block_max(m, lds);
            if (threadIdx.x == 0) out[q] = m;
        } else {
            double s = 0.0;
            for (int i = threadIdx.x; i < G; i += BLOCK) s += p[i];
            s = block_sum(s, lds);
            if (threadIdx.x == 0) out[q] = s;
        }
    }
}

}  // namespace fasp
