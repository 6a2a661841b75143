// kernels4.hip.h -- panel stream: the long-row operators with the gathered vector in LDS (gfx950 / CDNA4, wave64), round 6.
//
//   y = A x family (fasp_blas_dcsr_mxv, BlaSpmvCSR.c:242; fasp_blas_dcsr_aAxpy, :494; the residual of PreMGCycle.c:136) and the
//   weighted-Jacobi / L1 sweeps (ItrSmootherCSR.c:98, :1509) on operators with rows of 48 ... thousands of entries.
//
// What bounds every kernel that gathers x from memory on these levels (in-kernel stamps and PMC counters, profiles/r06_estream.txt,
// r06_pmc_bound.txt): not HBM and not instruction issue but the CU's texture-address / tag pipeline.  A gather of 64 lanes whose columns
// lie in ~45 different cache lines -- what a row of a coarse AMG level looks like, sorted or not -- takes ~45+ cycles there whether
// the lines hit or miss, eight of them per 512 entries use up two thirds of the ~520 cycles a CU has per 512 entries at 6 TB/s, and the
// stream loads, row pointers and epilogue operands share the same pipe.  k_csr_rows and k_csr_estream, as different as they are, end up
// at the same 3.3 TB/s.
//
// So the gathers leave that pipe:
//   * the columns are cut into PANELS of 8192 (64 KB of x); the entries are stored panel-major -- all entries of panel 0 row by row, then
//     panel 1, ... (values 8 B + 13-bit column in 16 bits: 10 bytes per entry for ANY number of columns, no per-row base);
//   * one workgroup of 16 wavefronts per CU takes an equal share of that entry sequence; for every panel its share touches (one,
//     sometimes two) it copies the panel of x into LDS once -- coalesced 16-byte loads, 64 instructions per CU -- and its waves then work
//     through the share's chunks exactly like k_csr_estream (stream one chunk ahead through the wave's LDS slab, lane = entry products,
//     sub-wavefront sums), except that x comes from LDS: ds_read_b64 at per-lane addresses, a few cycles per 64 lanes;
//   * what a wave sums is a SUB-ROW: the entries of one row inside one panel and one wave's share (cuts are made when the tables are
//     built, so a sub-row never leaves its wave: no cross-wave traffic in the kernel).  Sub-row sums go to S;
//   * a second, small kernel (k_pcombine) adds each row's sub-row sums in panel order and applies the epilogue -- y = A x, the
//     residual, y += alpha A x, Jacobi, L1, and the fused dot products (its partials are per block of rows: deterministic).
// Vector-memory instructions per 512 entries: 5 stream loads + 2 row-pointer loads + a handful of 8-byte stores, against 8 scattered
// gathers + the same.  Row sums associate by (panel, wave share, lane): fixed by the tables, deterministic; agreement with the
// reference 1e-13 per cycle like the other long-row kernels (tests/test_gpu_estream.py).
#pragma once

#include <type_traits>

#include "kernels3.hip.h"

namespace fasp {

constexpr int PS_P     = 8192;    // columns per panel (64 KB of x in LDS)
constexpr int PS_CAP   = 512;     // entries per chunk
constexpr int PS_WAVES = 16;      // wavefronts per workgroup (one workgroup per CU)
constexpr int PS_NT    = 64 * PS_WAVES;
constexpr int PS_IAW   = 127;     // sub-rows of a chunk whose pointers are staged
constexpr int PS_WAVE_LDS = (PS_CAP + 2) * 8 + PS_CAP * 2 + 128 * 4;   // products (+ the 0.0 slot), columns, sub-row pointers
constexpr int PS_LDS_BYTES = PS_P * 8 + PS_WAVES * PS_WAVE_LDS;
static_assert(PS_LDS_BYTES <= 160 * 1024, "x panel + 16 wave slabs fit one CU's LDS");

struct PsArgs {
    const double*         val;      // panel-major values
    const unsigned short* col;      // column - panel * PS_P
    const int*            wg_seg;   // per workgroup: its segments (NWG + 1)
    const int*            seg_panel;
    const int*            task_chunk;  // per (segment, wave): chunks (nseg * PS_WAVES + 1)
    const int*            centry;   // first entry of every chunk (+ end)
    const int*            csub;     // the sub-row that entry lies in
    const int*            sptr;     // first entry of every sub-row (ns + 1, in panel-major order)
    const double*         x;
    double*               S;        // sub-row sums
    int                   ncol, ns;
    int                   pw;       // columns per panel of THIS operator: ncol cut into equal panels of at most PS_P (a multiple of 8)
    const int*            stop;
};

template <int L>
__global__ __launch_bounds__(PS_NT) void k_csr_pstream(PsArgs a)
{
    if (a.stop && *a.stop) return;
    constexpr int CAP = PS_CAP, G = 64 / L, NV = CAP / 128, NJ = CAP / 512, NU = CAP / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char ps_lds[];
    double* const xs = reinterpret_cast<double*>(ps_lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* const slab = ps_lds + PS_P * 8 + wave * PS_WAVE_LDS;
    double*         sv  = reinterpret_cast<double*>(slab);
    unsigned short* sj  = reinterpret_cast<unsigned short*>(slab + (CAP + 2) * 8);
    int*            ssp = reinterpret_cast<int*>(slab + (CAP + 2) * 8 + CAP * 2);
    const int g = lane / L, sl = lane & (L - 1);
#ifdef ES_TIMING
    long long est[8] = {0, 0, 0, 0, 0, 0, 0, 0}, est_last = (long long)__builtin_readcyclecounter();
    const long long est_t0 = est_last;
    int est_chunks = 0;
    const unsigned long long est_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (lane == 0) sv[CAP] = 0.0;
    const int b = blockIdx.x;
    const int seg0 = es_tab(a.wg_seg, b), seg1 = es_tab(a.wg_seg, b + 1);
    for (int seg = seg0; seg < seg1; ++seg) {
        const int task = seg * PS_WAVES + wave;
        const int c0 = es_tab(a.task_chunk, task), cend = es_tab(a.task_chunk, task + 1);
        const bool work = c0 < cend;   // (uniform per wave; the barriers below are reached by every wave of the workgroup)

        f64x2_t qv[NV];
        u32x4_t qj[NJ];
        int     qi0 = 0, qi1 = 0;
        auto stage_load = [&](int lo, int n, int rf) {
            const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.val + lo), 0, ((n + 1) & ~1) * 8, 0x00020000);
            const __amdgpu_buffer_rsrc_t rj = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.col + lo), 0, ((n + 7) & ~7) * 2, 0x00020000);
#pragma unroll
            for (int q = 0; q < NV; ++q)
                qv[q] = __builtin_bit_cast(f64x2_t, __builtin_amdgcn_raw_buffer_load_b128(rv, (lane + 64 * q) * 16, 0, 0));
#pragma unroll
            for (int q = 0; q < NJ; ++q)
                qj[q] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rj, (lane + 64 * q) * 16, 0, 0));
            const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(a.sptr + rf), 0, n > 0 ? (a.ns + 1 - rf) * 4 : 0, 0x00020000);
            qi0 = __builtin_amdgcn_raw_buffer_load_b32(ri, lane * 4, 0, 0);
            qi1 = __builtin_amdgcn_raw_buffer_load_b32(ri, (64 + lane) * 4, 0, 0);
        };
        auto stage_store = [&]() {
#pragma unroll
            for (int q = 0; q < NV; ++q) reinterpret_cast<f64x2_t*>(sv)[lane + 64 * q] = qv[q];
#pragma unroll
            for (int q = 0; q < NJ; ++q) reinterpret_cast<u32x4_t*>(sj)[lane + 64 * q] = qj[q];
            ssp[lane] = qi0; ssp[64 + lane] = qi1;
        };
        // the task's chunk table in two registers: lane j holds the first entry / first sub-row of its j-th chunk (lane nch: the end) -- one
        // coalesced load per table and task instead of scalar loads per chunk, every one of which missed the scalar cache on cold tables
        // (2-3 us each, serialised: profiles/r06_estream.txt); a chunk's bounds are then read off with v_readlane
        const int nch = work ? cend - c0 : 0;   // (<= 62: build_pstream_host)
        int tb_e = 0, tb_s = 0;
        if (work) { const int j = c0 + min(lane, nch); tb_e = a.centry[j]; tb_s = a.csub[j]; }
        int lo = __builtin_amdgcn_readlane(tb_e, 0), hi = __builtin_amdgcn_readlane(tb_e, 1), rf = __builtin_amdgcn_readlane(tb_s, 0), rl = __builtin_amdgcn_readlane(tb_s, 1);
        stage_load(lo, hi - lo, rf);   // the task's first chunk travels while the panel is copied (no work: empty ranges, nothing fetched)
        // ---- this segment's panel of x into LDS (columns beyond the operator's last: 0.0, never addressed)
        __syncthreads();
        {
            const int panel = es_tab(a.seg_panel, seg);
            const int x0 = panel * a.pw, nc = min(a.pw, a.ncol - x0);
            // (16-byte loads: x0 is a multiple of 8 columns; the buffer range ends at the panel's last column -- a load across it returns 0.0 there)
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.x + x0), 0, nc * 8, 0x00020000);
#pragma unroll
            for (int q = 0; q < PS_P / 2 / PS_NT; ++q) {
                const int i = tid + q * PS_NT;   // pair index
                reinterpret_cast<f64x2_t*>(xs)[i] = __builtin_bit_cast(f64x2_t, __builtin_amdgcn_raw_buffer_load_b128(rx, i * 16, 0, 0));
            }
        }
        __syncthreads();
        EST(0);
        if (!work) continue;
        double acc = 0.0;
        stage_store();
        wave_order();
        EST(1);
        for (int c = 0; c < nch; ++c) {
#ifdef ES_TIMING
            ++est_chunks;
#endif
            const bool more = c + 1 < nch;
            const int  j2 = more ? c + 2 : c + 1;
            const int  hi2 = __builtin_amdgcn_readlane(tb_e, j2), rl2 = __builtin_amdgcn_readlane(tb_s, j2);
            stage_load(hi, more ? hi2 - hi : 0, rl);   // the next chunk travels while this one is worked on (nothing below loads from memory)
            // ---- phase 1: lane = entry, x from the panel in LDS
            double pr[NU];
            {
                int    cc[NU];
                double wv[NU], xv[NU];
#pragma unroll
                for (int u = 0; u < NU; ++u) cc[u] = (int)sj[lane + 64 * u];
#pragma unroll
                for (int u = 0; u < NU; ++u) wv[u] = sv[lane + 64 * u];
#pragma unroll
                for (int u = 0; u < NU; ++u) xv[u] = xs[cc[u]];
#pragma unroll
                for (int u = 0; u < NU; ++u) pr[u] = wv[u] * xv[u];   // (beyond the chunk: value 0.0 x x[first column of the panel])
            }
            EST(2);
            if (L == 64) {
                // ---- phase 2, long sub-rows (a chunk touches two or three): the sums are formed from the products IN REGISTERS -- entry
                // lo + lane + 64 u sits in pr[u] of this lane; a sub-row is a contiguous range of entries; its bounds come through the scalar
                // cache; every branch is wave-uniform.  No LDS round trip (a pass over the slab costs ~3 600 cycles with sixteen waves on the
                // CU's LDS: profiles/r06_estream.txt).  A sub-row that the chunk's end cuts leaves its per-lane partial sums in acc.
                const int e0 = lane;   // (entry index relative to lo)
                for (int r = rf; r <= rl && r < a.ns; ++r) {
                    // (bounds from the window of sub-row pointers staged with the chunk; a chunk of such an operator touches a handful)
                    const int wi = r - rf;
                    int kb, ke;
                    if (wi < PS_IAW) { kb = __builtin_amdgcn_readfirstlane(ssp[wi]); ke = __builtin_amdgcn_readfirstlane(ssp[wi + 1]); }
                    else { kb = es_tab(a.sptr, r); ke = es_tab(a.sptr, r + 1); }
                    const int ks = max(kb, lo) - lo, kn = min(ke, hi) - lo - ks;   // first entry (relative to lo) and count inside the chunk
                    double t = (r == rf) ? acc : 0.0;
                    if (ks == 0 && kn >= hi - lo) {
#pragma unroll
                        for (int u = 0; u < NU; ++u) t += pr[u];
                    } else if (kn > 0) {
#pragma unroll
                        for (int u = 0; u < NU; ++u) t += ((unsigned)(e0 + 64 * u - ks) < (unsigned)kn) ? pr[u] : 0.0;
                    }
                    if (ke <= hi) {
                        const double tot = es_group_sum<64>(t);
                        if (lane == 63) a.S[r] = tot;
                        acc = 0.0;
                    } else acc = t;
                }
            } else {
#pragma unroll
                for (int u = 0; u < NU; ++u) sv[lane + 64 * u] = pr[u];
                wave_order();
                // ---- phase 2, short sub-rows: 64 / L at a time from the slab, L lanes each
                auto rows_pass = [&](auto slow_tag) {
                    constexpr bool SLOW = decltype(slow_tag)::value;
                    for (int rb = rf & ~(G - 1); rb <= rl; rb += G) {
                        const int  r   = rb + g;
                        const bool act = r >= rf && r <= rl && r < a.ns;
                        const int  wi  = r - rf;
                        int kb = 0, ke = 0;
                        if (SLOW) { if (act) { kb = a.sptr[r]; ke = a.sptr[r + 1]; } }
                        else { const int wc_ = act ? wi : 0; kb = ssp[wc_]; ke = ssp[wc_ + 1]; if (!act) { kb = 0; ke = 0; } }
                        const int k0 = max(kb, lo) + sl, kq = min(ke, hi);
                        const int n  = (kq - k0 + L - 1) >> __builtin_ctz(L);
                        const double* const pp = sv + (k0 - lo);
                        const double* const pz = sv + CAP;
                        for (int u = 0; __any(u < n); u += 4) {
                            const double* q = pp + u * L;
                            const double p0 = *(u + 0 < n ? q : pz), p1 = *(u + 1 < n ? q + L : pz), p2 = *(u + 2 < n ? q + 2 * L : pz), p3 = *(u + 3 < n ? q + 3 * L : pz);
                            acc += p0; acc += p1; acc += p2; acc += p3;
                        }
                        const bool fin = act && ke <= hi;
                        if (__any(fin)) {
                            const double tot = es_group_sum<L>(acc);
                            if (fin) {
                                if (sl == L - 1) a.S[r] = tot;
                                acc = 0.0;
                            }
                        }
                    }
                };
                if (rl - rf < PS_IAW) rows_pass(std::false_type{});
                else rows_pass(std::true_type{});
            }
            EST(3);
            wave_order();
            stage_store();
            wave_order();
            EST(4);
            lo = hi; hi = hi2; rf = rl; rl = rl2;
        }
    }
#ifdef ES_TIMING
    if (lane == 0 && (((b * PS_WAVES + wave) % 797) == 0 || __builtin_amdgcn_s_memrealtime() - est_r0 > 6000ull) && est_chunks) printf("[ps] wg %d wave %d: real time %llu .. %llu (10 ns ticks), %d chunks, total %lld cycles; fill + barriers %lld, first chunk %lld; per chunk: products %lld, row sums %lld, stream -> LDS %lld\n", b, wave, est_r0 % 10000000ull, __builtin_amdgcn_s_memrealtime() % 10000000ull, est_chunks, (long long)__builtin_readcyclecounter() - est_t0, est[0], est[1], est[2] / est_chunks, est[3] / est_chunks, est[4] / est_chunks);
#endif
}

// ---- a row's sub-row sums in panel order, then the epilogue (one thread per row; rows in blocks: the dot-product partials are per block)
struct PcArgs {
    int           nrow;
    const int*    rp;      // per row: its sub-rows in rsub (nrow + 1)
    const int*    rsub;    // sub-row ids, a row's in panel (= summation) order
    const double* S;
};
template <int OP>
__global__ __launch_bounds__(BLOCK) void k_pcombine(PcArgs p, CsrArgs a)
{
    if (a.stop && *a.stop) return;
    __shared__ double red[4];
    double dotacc = 0.0;
    for (int r = blockIdx.x * BLOCK + threadIdx.x; r < p.nrow; r += gridDim.x * BLOCK) {
        double s = 0.0;
        const int q0 = p.rp[r], q1 = p.rp[r + 1];
        for (int q = q0; q < q1; ++q) s += p.S[p.rsub[q]];
        if (OP == OP_MXV) { a.y[r] = s; zx_store(a, r, s); }
        else if (OP == OP_RESID) a.y[r] = a.b[r] - s;
        else if (OP == OP_ADD) a.y[r] += s;
        else if (OP == OP_SUB) a.y[r] -= s;
        else if (OP == OP_AXPY) a.y[r] += s * a.alpha;
        else if (OP == OP_JACOBI) {   // s holds every entry of the row: the diagonal's product goes out here (kernels3.hip.h)
            const double d = a.diag[r], xi = a.x[r];
            const double tt = a.b[r] - (s - d * xi);
            const double xn = (fabs(d) > 1e-20) ? (1 - a.omega) * xi + a.omega * tt / d : xi;
            a.y[r] = xn;
            if (a.partials) dotacc += xn * a.b[r];
        } else if (OP == OP_L1DIAG) {
            const double d = a.diag[r], xi = a.x[r];
            a.y[r] = l1_or_jacobi_f(a, r, a.b[r] - s, d, xi);
        } else if (OP == OP_MXV_DOT) {
            a.y[r] = s;
            dotacc += s * a.dotv[r];
        }
    }
    if (OP == OP_MXV_DOT || (OP == OP_JACOBI && a.partials)) {
        const double tot = block_sum(dotacc, red);
        if (threadIdx.x == 0) a.partials[blockIdx.x] = tot;
    }
}

}  // namespace fasp
