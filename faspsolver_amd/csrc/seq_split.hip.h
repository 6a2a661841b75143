// seq_split.hip.h -- the sequential sweeps (Gauss-Seidel / SOR family, ItrSmootherCSR.c:251-334, :932-1040) as
//   (1) ONE fully parallel pass over everything a row reads that the sweep has not touched yet, and
//   (2) a sparse lower-triangular solve over what remains.
// Part of the single translation unit solver.hip (after small_solvers.hip.h; not a stand-alone header).
//
// The reference's sweep over the rows i_0, i_1, ... computes, row after row,
//     t = b_i - sum_{j != i} a_ij u_j,   u_i = t / a_ii          (SOR: u_i = w t / a_ii + (1 - w) u_i)
// where u_j is the NEW value for the rows j the sweep visited before i and the OLD one otherwise.  The old values do
// not change while the sweep runs, so their part of every row sum can be formed up front, for all rows at once:
//     W_p = b_i - sum_{j not swept before i} a_ij u_j^old        (k_split_rest: a plain gather SpMV, any order)
// and what remains is the triangular recurrence over the sweep's own rows, numbered p = 0, 1, ... in sweep order:
//     W_p <- (W_p - sum_{q < p coupled} l_pq W_q) / a_ii          (k_tri_block / k_tri_level)
// The dependency chain carries only the "lower" entries (about half of a row; a quarter in a C-row or F-row sweep),
// the iterate the chain reads is the COMPACT vector W of the swept rows (13 000 C rows of a 35 000-row level fit the
// LDS of one workgroup where the level's whole iterate does not), and the anti-dependencies (row j reads the old u_i
// of a later row) vanish because pass (1) has read every old value before anything is written.
// Arithmetic: t is formed as (b_i - rest) - lower instead of b_i - (all entries in storage order): a regrouping of
// the same products (differences of a few ulp per row; the tests pin iteration counts and |relres - ref| <= 1e-10
// against the reference's own runs, tests/golden/p7_sweeps.npz).
//
// Storage of the lower part ("slots"): the rows of a dependency class are numbered consecutively (class-major) and cut
// into CHUNKS of at most 512 / L rows (L lanes per row: one chunk is one round of a 512-thread workgroup).  A chunk
// [lo, hi) stores pf * L slots per row (pf <= 8 rounds: what the longest row of the chunk needs; entry e of a row is round
// e / L of lane e % L), in packs of four rounds, a lane's four slots of a pack side by side:   slot (q, lane sl) of row p at
//     sbase + ((q / 4) * L * (hi - lo) + (p - lo) * L + sl) * 4 + q % 4
// so that the address of everything a chunk needs follows from its descriptor (lo | pf << 28, sbase) alone: ONE memory
// round trip per chunk (no row pointer -> entries chain), perfectly coalesced, issued several chunks ahead of the one
// being computed.  Unused slots hold (column p, value 0).  Rows with more than 4 L lower entries keep the excess in a
// CSR tail.  The per-row scalars travel as three small records: (b - rest, old u_i) from pass (1), (a_ii, 1 / a_ii) and
// (tail count, row index) from the schedule.
#pragma once

namespace fasp {

struct TriArgs {
    const int*    lptr;    // nchunk + 1 chunk descriptors: first position | slots per lane << 28
    const int*    sbase;   // nchunk + 1 slot offsets
    int           nchunk;
    const int*    sc;      // slot columns (positions) ...
    const double* sv;      // ... and values
    const int*    tia;     // tail CSR (positions + 1 offsets; entries beyond the slots)
    const int*    tja;
    const double* tval;
    const double* rec;     // 2 doubles per position, written by pass (1): b - rest, old u_i
    const double* dr;      // 2 doubles per position: a_ii, 1 / a_ii rounded to nearest (0 for a row that is left alone)
    const int*    tr;      // 2 ints per position: tail count | (row left alone) << 31, row index
    const int*    order;   // row of position p
    double*       W;       // the new iterate of the swept rows (variants that do not keep it in LDS)
    double*       u;       // the level's iterate (scatter target)
    int           nrow;    // rows of the level
    int           far;     // the schedule has far entries: the one-workgroup solve also writes W
    int           form;    // 0  u_i = t * (1/a_ii)   1  u_i = t / a_ii   2  u_i = w (t / a_ii) + (1 - w) u_i
    double        w;
};
constexpr int TRI_POS_MASK = 0x0fffffff;
constexpr int TRI_FAR_BIT = 0x10000000;          // tail columns: an entry the LDS ring does not reach back to (read from W in memory)
constexpr size_t TRI_LDS_CAP = 158 * 1024;      // dynamic LDS of the one-workgroup solve

// t / d from the stored reciprocal rd = RN(1 / d): q = RN(t rd), then one correction step with the exact remainder
// (Markstein): q' = RN(q + (t - d q) rd) -- the correctly rounded quotient (the IEEE division the reference performs)
// up to rare last-place cases, at three operations instead of the two dozen of the division sequence.  The division
// sits in the middle of a chain of thousands of dependent rows.
__device__ __forceinline__ double tri_div(double t, double d, double rd)
{
    const double q = t * rd;
    const double r = __builtin_fma(-d, q, t);
    return __builtin_fma(r, rd, q);
}
__device__ __forceinline__ double tri_update(double t, double d, double rd, bool alone, int form, double w, double uold)
{
    if (alone) return uold;                  // ItrSmootherCSR.c: rows with |a_ii| <= SMALLREAL are left alone
    if (form == 0) return t * rd;            // t * (1.0 / a_ii)
    if (form == 1) return tri_div(t, d, rd);
    return w * tri_div(t, d, rd) + (1 - w) * uold;
}

// Sum over the L lanes of a row group by data-parallel-primitive moves (no LDS crossbar round trips; small_solvers.hip.h):
// inclusive row_shr 1, 2, 4, 8 inside the rows of 16, row_bcast 15 / 31 across them.  The total ends in the group's
// LAST lane (sl == L - 1).  Fixed order.
template <int L>
__device__ __forceinline__ double group_sum_last(double x)
{
    if (L >= 2) x += dpp_mov_f64(x, 0x111, 0xf);
    if (L >= 4) x += dpp_mov_f64(x, 0x112, 0xf);
    if (L >= 8) x += dpp_mov_f64(x, 0x114, 0xf);
    if (L >= 16) x += dpp_mov_f64(x, 0x118, 0xf);
    if (L >= 32) x += dpp_mov_f64(x, 0x142, 0xa);
    if (L >= 64) x += dpp_mov_f64(x, 0x143, 0xc);
    return x;
}

// pass (1): rec_p = (b_i - (entries of row i that read old values), u_i); L lanes per row, grid-stride
template <int L>
__global__ __launch_bounds__(BLOCK) void k_split_rest(int nseq, const int* __restrict__ order, const int* __restrict__ ria,
                                                       const int* __restrict__ rja, const double* __restrict__ rval,
                                                       const double* __restrict__ b, const double* __restrict__ u,
                                                       double* __restrict__ rec, unsigned* __restrict__ prog)
{
    constexpr int RPB = BLOCK / L;
    if (blockIdx.x == 0 && threadIdx.x == 0) { prog[0] = 0u; prog[4] = 0u; prog[6] = 0u; }   // progress word of the one-workgroup solve that follows (tri_prefetch); arrivals and progress of the cluster form
    const int sl = threadIdx.x & (L - 1);
    const int rloc = threadIdx.x / L;
    for (int p0 = blockIdx.x * RPB; p0 < nseq; p0 += gridDim.x * RPB) {   // (whole wavefronts walk the loop: the DPP moves read neighbours)
        const int p = p0 + rloc;
        const bool on = p < nseq;
        const int kb = on ? ria[p] : 0, ke = on ? ria[p + 1] : 0;
        double s = seq_row_sum<L>(rja, rval, kb + sl, ke, -1, [&](int c) { return u[c]; });
        s = group_sum_last<L>(s);
        if (on && sl == L - 1) {
            const int r = order[p];
            f64x2_t o;
            o[0] = b[r] - s; o[1] = u[r];
            *reinterpret_cast<f64x2_t*>(rec + 2 * (size_t)p) = o;
        }
    }
}

// u_i <- W_p (final == 0), or the update of a sweep without any lower entry straight from pass (1) (final == 1)
__global__ __launch_bounds__(BLOCK) void k_split_scatter(int nseq, TriArgs a, int final)
{
    for (int p = blockIdx.x * BLOCK + threadIdx.x; p < nseq; p += gridDim.x * BLOCK) {
        double v;
        if (final) v = tri_update(a.rec[2 * (size_t)p], a.dr[2 * (size_t)p], a.dr[2 * (size_t)p + 1], a.tr[2 * (size_t)p] < 0, a.form, a.w, a.rec[2 * (size_t)p + 1]);
        else v = a.W[p];
        a.u[a.order[p]] = v;
    }
}

// what a lane holds of a row before the chain reaches it: PF slots and the row's records
template <int PF>
struct TriPre { int c[PF]; double v[PF]; double t, uo, d, rd; int tn, row; };

// buffer resources of the arrays a chunk is fetched from: one 32-bit offset register per lane serves every slot round
// (the round's displacement is wave-uniform and travels in a scalar register), instead of 64-bit address arithmetic per load
struct TriBufs { __amdgpu_buffer_rsrc_t sc, sv, rec, dr, tr, u; };
__device__ __forceinline__ TriBufs tri_bufs(const TriArgs& a)
{
    TriBufs B;
    B.sc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(a.sc), 0, 0x7fffffff, 0x00020000);
    B.sv = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.sv), 0, 0x7fffffff, 0x00020000);
    B.rec = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.rec), 0, 0x7fffffff, 0x00020000);
    B.dr = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.dr), 0, 0x7fffffff, 0x00020000);
    B.tr = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(a.tr), 0, 0x7fffffff, 0x00020000);
    B.u = __builtin_amdgcn_make_buffer_rsrc(a.u, 0, (int)((unsigned)a.nrow * 8u), 0x00020000);
    return B;
}
// Branch-free, and the same number of loads whatever the chunk looks like: the compiler's wait-count bookkeeping can
// then tell how many younger loads are in flight behind the ones a chunk is about to use (s_waitcnt vmcnt(N), N > 0);
// with loads under conditions it has to assume the fewest and waits for ALL of them.  Idle lanes (p >= hi) and unused
// slot rounds (q >= pf) put their offset beyond the buffers' range instead: such a lane gets 0 back -- column 0, value
// 0: a product that adds nothing -- and causes no memory access.
constexpr int TRI_OOR = (int)0x80000000u;   // (the resources declare 2^31 - 1 bytes)
template <int L, int PF>
__device__ __forceinline__ void tri_fetch(const TriBufs& B, TriPre<PF>& r, int lo, int hi, int pf, int sb, int rloc, int sl)
{
    static_assert(PF % 4 == 0, "slots come in packs of four");
    const bool on = rloc < hi - lo;
    const int p = lo + rloc;
    const int e = sb + (rloc * L + sl) * 4;   // first slot of this lane's pack 0 (host: nslot < 2^28)
    const int gs = 4 * L * (hi - lo);         // slots per pack of the chunk: wave-uniform
#pragma unroll
    for (int g = 0; g < PF / 4; ++g) {        // four slots of a lane are contiguous: one 16-byte load of columns, two of values
        const bool use = on && 4 * g < pf;
        const u32x4_t cq = __builtin_amdgcn_raw_buffer_load_b128(B.sc, use ? e * 4 : TRI_OOR, g * gs * 4, 0);
        const f64x2_t v0 = __builtin_bit_cast(f64x2_t, __builtin_amdgcn_raw_buffer_load_b128(B.sv, use ? e * 8 : TRI_OOR, g * gs * 8, 0));
        const f64x2_t v1 = __builtin_bit_cast(f64x2_t, __builtin_amdgcn_raw_buffer_load_b128(B.sv, use ? e * 8 + 16 : TRI_OOR, g * gs * 8, 0));
        r.c[4 * g] = (int)cq[0]; r.c[4 * g + 1] = (int)cq[1]; r.c[4 * g + 2] = (int)cq[2]; r.c[4 * g + 3] = (int)cq[3];
        r.v[4 * g] = v0[0]; r.v[4 * g + 1] = v0[1]; r.v[4 * g + 2] = v1[0]; r.v[4 * g + 3] = v1[1];
    }
    const f64x2_t r0 = buf_load_f64x2(B.rec, on ? (unsigned)p * 16u : (unsigned)TRI_OOR), r1 = buf_load_f64x2(B.dr, on ? (unsigned)p * 16u : (unsigned)TRI_OOR);
    const u32x2_t r2 = __builtin_amdgcn_raw_buffer_load_b64(B.tr, on ? p * 8 : TRI_OOR, 0, 0);
    r.t = r0[0]; r.uo = r0[1]; r.d = r1[0]; r.rd = r1[1];
    r.tn = (int)r2[0]; r.row = (int)r2[1];
}

// the row arithmetic shared by every variant (identical bits): lane-strided products in slot order, then the tail,
// the DPP tree, the update in the group's last lane.  ldw(c) reads W_c; returns the new value (valid in lane L - 1).
template <int L, int PF, bool TAIL, class LDW>
__device__ __forceinline__ double tri_row(const TriArgs& a, const TriPre<PF>& r, int p, int sl, LDW ldw)
{
    double x[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) x[q] = ldw(r.c[q]);
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < PF; ++q) s += r.v[q] * x[q];
    if (TAIL && (r.tn & 0x7fffffff)) {   // (memory loads inside the chain: only schedules that have such rows compile this in)
        const int kb = a.tia[p], ke = a.tia[p + 1];
        s = seq_row_sum<L>(a.tja, a.tval, kb + sl, ke, -1, ldw, s);
    }
    s = group_sum_last<L>(s);
    return tri_update(r.t - s, r.d, r.rd, r.tn < 0, a.form, a.w, r.uo);
}

// ONE dependency class per launch, one workgroup per chunk (classes of thousands of rows: the upper levels)
#ifndef TRI_BLOCK_THREADS
#define TRI_BLOCK_THREADS 512
#endif
constexpr int TRI_BLOCK = TRI_BLOCK_THREADS;
constexpr int TRI_PFMAX = 8;   // slot rounds a chunk can store; the kernels come with room for 4 (schedules that never need more) or 8
template <int L>
__global__ __launch_bounds__(TRI_BLOCK) void k_tri_level(TriArgs a, int chunk0)
{
    const int sl = threadIdx.x & (L - 1);
    const int rloc = threadIdx.x / L;
    const int ck = chunk0 + blockIdx.x;
    const int d0 = a.lptr[ck], lo = d0 & TRI_POS_MASK, pf = (int)((unsigned)d0 >> 28), hi = a.lptr[ck + 1] & TRI_POS_MASK;
    const int p = lo + rloc;
    if (p < hi) {
        const TriBufs B = tri_bufs(a);
        TriPre<TRI_PFMAX> r;
        tri_fetch<L, TRI_PFMAX>(B, r, lo, hi, pf, a.sbase[ck], rloc, sl);
        const double un = tri_row<L, TRI_PFMAX, true>(a, r, p, sl, [&](int c) -> double { return a.W[c & TRI_POS_MASK]; });
        if (sl == L - 1) a.W[p] = un;
    }
}

// A whole triangular solve in ONE workgroup: a barrier per chunk.  Everything a chunk needs from memory is fetched a
// GROUP of G chunks ahead (the addresses follow from the chunk descriptors, which sit in LDS), so the chain is
//   LDS gather of W -> PF multiply-adds -> DPP tree -> update -> LDS store -> LDS-only barrier
// (global loads and stores in flight stay in flight; what a chunk costs is the wavefront's serial instruction stream --
// the chain plus the issue of the next fetch -- hence buffer loads with scalar displacements, the stored reciprocal,
// the DPP tree).
// Two register sets of G chunks each: drain the memory counter (the set about to be used was requested a whole group
// ago), request the next group into the other set, run the G chains.  A ring of single chunks with per-chunk waits
// would be finer-grained, but the compiler's wait-count bookkeeping does not survive a loop back-edge with loads in
// flight: it drained the counter -- including the loads issued one chunk earlier -- once per trip (measured 0.4 us of
// stall per chunk).  A full drain at group granularity costs nothing, because nothing younger is in flight yet.
// WIN: the new values live in an LDS ring of `cap` doubles (cap a power of two): position c sits at c & (cap - 1).  The
// host checks that no row reads further back than the ring reaches (hi - c <= cap for every lower entry c of a row of
// chunk [lo, hi)), so W never travels through memory: results go to the ring and straight to u_i.
// !WIN (a schedule that reaches further back): W goes through the L2 with agent-scope atomics (a wave must see what a
// wave of another SIMD stored in the chunk before), a chunk costs two L2 round trips; ends with the scatter u_i <- W_p.
// Helper workgroups of k_tri_block (blocks 8, 16, ... of its grid: under the round-robin placement of workgroups they share
// the solving workgroup's XCD, hence its L2): they READ what the solver is going to need -- slots and row records of the
// chunks h, h + nhelp, ... -- a bounded distance ahead of its published progress and throw it away.  One compute unit
// sustains ~25 GB/s of fetches from HBM (a few hundred cache lines in flight against ~2 us); from the L2 the same lines
// in flight come back three to four times faster.  Purely a hint: if the placement is another one, or the progress word
// is never seen, the solver reads from memory as before and the result is the same.
__device__ __forceinline__ void tri_prefetch(const TriArgs& a, int h, int nhelp, int ahead, const unsigned* prog)
{
    typedef __attribute__((address_space(1))) unsigned gu32;
    unsigned long long sink = 0;
    for (int c = h; c < a.nchunk; c += nhelp) {
        if (threadIdx.x == 0) {
            for (int spin = 0; spin < 4096; ++spin) {   // bounded: a helper that cannot see the progress just runs on
                if ((int)__hip_atomic_load((gu32*)prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + ahead >= c) break;
                __builtin_amdgcn_s_sleep(8);
            }
        }
        __syncthreads();
        const int lo = a.lptr[c] & TRI_POS_MASK, hi = a.lptr[c + 1] & TRI_POS_MASK;
        const size_t s0 = (size_t)a.sbase[c], s1 = (size_t)a.sbase[c + 1];
        const u32x4_t* v16 = reinterpret_cast<const u32x4_t*>(a.sv + s0);
        const size_t   nv = (s1 - s0) / 2;   // 16-byte units of the values (slot counts are multiples of four)
        for (size_t i = threadIdx.x; i < nv; i += TRI_BLOCK) { const u32x4_t q = v16[i]; sink += q[0] ^ q[3]; }
        const u32x4_t* c16 = reinterpret_cast<const u32x4_t*>(a.sc + s0);
        for (size_t i = threadIdx.x; i < nv / 2; i += TRI_BLOCK) { const u32x4_t q = c16[i]; sink += q[1]; }
        for (int p = lo + (int)threadIdx.x; p < hi; p += TRI_BLOCK) {
            sink += (unsigned long long)__double_as_longlong(a.dr[2 * (size_t)p]) ^ (unsigned long long)a.tr[2 * (size_t)p]
                    ^ (unsigned long long)__double_as_longlong(a.rec[2 * (size_t)p]);
        }
    }
    asm volatile("" ::"v"((unsigned)sink), "v"((unsigned)(sink >> 32)));   // (the loads are the point)
}

template <int L, int PF, bool WIN, bool TAIL>
__global__ __launch_bounds__(TRI_BLOCK) void k_tri_block(TriArgs a, int nseq, int cap, int nhelp, int ahead, unsigned* prog)
{
    if (blockIdx.x != 0) {
        if ((blockIdx.x & 7) == 0) tri_prefetch(a, (int)(blockIdx.x >> 3) - 1, nhelp, ahead, prog);
        return;
    }
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    extern __shared__ __attribute__((aligned(16))) double tri_lds[];
    constexpr int G = !WIN ? 1 : PF <= 4 ? 4 : 3;   // (!WIN drains the counter per chunk anyway)
    // dynamic LDS: [ring (WIN only)] [chunk descriptors] [slot offsets]
    double* ring = tri_lds;
    int*    lptr = reinterpret_cast<int*>(tri_lds + (WIN ? cap : 0));
    int*    sbase = lptr + a.nchunk + 1;
    const int mask = cap - 1;
    const int sl = threadIdx.x & (L - 1);
    const int rloc = threadIdx.x / L;
    const int wave_rloc0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u) / L);   // first row slot of this wavefront
    const TriBufs B = tri_bufs(a);
    for (int i = threadIdx.x; i <= a.nchunk; i += TRI_BLOCK) { lptr[i] = a.lptr[i]; sbase[i] = a.sbase[i]; }
    if (WIN)
        for (int i = threadIdx.x; i < cap; i += TRI_BLOCK) ring[i] = 0.0;   // (unused slots read their own, not yet written, position: times 0)
    __syncthreads();
    auto ldw = [&](int c) -> double {
        if (WIN && !(TAIL && (c & TRI_FAR_BIT))) return ring[c & mask];
        return __longlong_as_double((long long)__hip_atomic_load((gu64*)(a.W + (c & TRI_POS_MASK)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    };
    struct Set { TriPre<PF> r[G]; int lo[G], hi[G]; };
    Set S0, S1;
    auto fetch_group = [&](Set& S, int l0) {   // chunks l0 .. l0 + G - 1 (past the end: the last chunk again)
#pragma unroll
        for (int d = 0; d < G; ++d) {
            const int l = min(l0 + d, a.nchunk - 1);
            const int d0 = __builtin_amdgcn_readfirstlane(lptr[l]), d1 = __builtin_amdgcn_readfirstlane(lptr[l + 1]);
            const int sb = __builtin_amdgcn_readfirstlane(sbase[l]);
            S.lo[d] = d0 & TRI_POS_MASK; S.hi[d] = d1 & TRI_POS_MASK;
            // a wavefront none of whose rows exists in this chunk skips the loads: the address unit of the compute unit
            // is shared, and eleven loads from each of eight wavefronts -- even all out of range -- took longer than the chain
            if (wave_rloc0 < S.hi[d] - S.lo[d]) tri_fetch<L, PF>(B, S.r[d], S.lo[d], S.hi[d], (int)((unsigned)d0 >> 28), sb, rloc, sl);
        }
    };
    auto run_chunk = [&](const Set& S, int d, int l0) {
        const int l = l0 + d;
        const int p = S.lo[d] + rloc;
        if (l < a.nchunk && p < S.hi[d]) {   // (whole row groups: the DPP moves stay inside a group; idle wavefronts go straight to the barrier)
            const double un = tri_row<L, PF, TAIL>(a, S.r[d], p, sl, ldw);
            if (sl == L - 1) {
                if (WIN) {
                    ring[p & mask] = un;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, un), B.u, S.r[d].row * 8, 0, 0);
                    if (a.far) a.W[p] = un;   // (read back, L1 bypassed, at least four drained groups of chunks later)
                } else __hip_atomic_store((gu64*)(a.W + p), (unsigned long long)__double_as_longlong(un), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (WIN) lds_barrier();
        else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    };
    // one group: its first chain (the compiler's own wait for the set lands here: everything requested a group ago has
    // arrived, nothing younger is in flight), THEN the request for the next group, then the remaining chains
    auto run_group = [&](const Set& S, Set& N, int l0) {
        if (nhelp && threadIdx.x == 0) __hip_atomic_store((__attribute__((address_space(1))) unsigned*)prog, (unsigned)l0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (also what makes a far read safe: every store of W older than a group has landed)
        run_chunk(S, 0, l0);
        fetch_group(N, l0 + G);
#pragma unroll
        for (int d = 1; d < G; ++d) run_chunk(S, d, l0);
    };
    fetch_group(S0, 0);
    for (int l0 = 0; l0 < a.nchunk; l0 += 2 * G) {
        run_group(S0, S1, l0);
        run_group(S1, S0, l0 + G);
    }
    if (!WIN)
        for (int p = threadIdx.x; p < nseq; p += TRI_BLOCK) a.u[a.order[p]] = ldw(p);
}

// ---------------------------------------------------------------------------
// k_tri_cluster<L>: the triangular solve of a schedule with WIDE dependency classes (hundreds to thousands of rows: the upper
// levels) in one launch of a few cooperating workgroups.  One launch per class costs 3.2-3.9 us of dispatch; a barrier
// among 2-16 workgroups on ONE XCD -- arrival by an atomic add, release by polling, with the class's values written
// before and read (L1 bypassed) after it -- costs 0.7 us (tools/micro/xcdbar.hip: 0.68-0.70 us for 2-8 workgroups,
// 0.72-1.05 us when they sit on several XCDs).
//   * grid = 8 (nb + nhelp) workgroups; those whose index is a multiple of 8 take part (one XCD under the round-robin
//     placement of workgroups): nb solvers, then nhelp helpers that read ahead into the XCD's L2 (tri_prefetch).
//   * class l: solver b takes chunks cptr[l] + b, + nb, ...; W lives in memory: read with L1-bypassing loads, written
//     with plain stores (they stay in the XCD's L2) when every solver reported the same HW_REG_XCC_ID at the start,
//     with write-through (agent-scope) stores otherwise -- correct on any placement, fastest on the usual one.
//   * the fetch for the next class is issued after the drain of this class's stores, so it travels during the wait.
//   * every spin is bounded; a solver that is not resident raises the error word (the host checks it, smoothers.hip.h).
// Same slots, same row arithmetic (tri_row) as the other two forms: identical bits.
// ---------------------------------------------------------------------------
template <int L>
__global__ __launch_bounds__(TRI_BLOCK) void k_tri_cluster(TriArgs a, const int* __restrict__ cptr, const int* __restrict__ cdesc, int nlev, int nseq, int nb, int nhelp,
                                                            int ahead, unsigned* sync /* [0] arrivals, [1] error, [2] progress, [16 + b] XCC ids */)
{
    typedef __attribute__((address_space(1))) unsigned           gu32;
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    if (blockIdx.x & 7) return;
    const int b = (int)(blockIdx.x >> 3);
    if (b >= nb) { tri_prefetch(a, b - nb, nhelp, ahead, sync + 2); return; }
    __shared__ int s_ok, s_same;
    const int tid = threadIdx.x;
    const int sl = tid & (L - 1);
    const int rloc = tid / L;
    const TriBufs B = tri_bufs(a);
    gu32* g_cnt = (gu32*)sync;
    gu32* g_err = (gu32*)(sync + 1);
    unsigned round = 0;
    auto barrier = [&]() -> bool {   // (the caller has issued everything it wants visible; returns false when the cluster is broken)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ++round;
        if (tid == 0) {
            __hip_atomic_fetch_add(g_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)nb * round;
            int ok = 1;
            unsigned spins = 0;
            unsigned long long t0 = 0;
            while (__hip_atomic_load(g_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if ((++spins & 1023u) == 0u) {   // (the clock is a memory operation of its own: not in every turn of the poll)
                    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                    if (!t0) t0 = now;
                    else if (now - t0 > 200000000ull) {   // 2 s at 100 MHz: a solver is not resident
                        __hip_atomic_store(g_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = 0;
                        break;
                    }
                }
            }
            if (__hip_atomic_load(g_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ok = 0;
            s_ok = ok;
        }
        __syncthreads();
        return s_ok != 0;
    };
    // placement: do the solvers share an XCD?
    if (tid == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        __hip_atomic_store((gu32*)(sync + 16 + b), id & 0xfu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!barrier()) return;
    if (tid == 0) {
        int same = 1;
        const unsigned id0 = __hip_atomic_load((gu32*)(sync + 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int q = 1; q < nb; ++q)
            if (__hip_atomic_load((gu32*)(sync + 16 + q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != id0) same = 0;
        s_same = same;
    }
    __syncthreads();
    const bool same_xcd = s_same != 0;
    auto ldw = [&](int c) -> double {
        return __longlong_as_double((long long)__hip_atomic_load((gu64*)(a.W + (c & TRI_POS_MASK)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    };
    // cdesc[(l * nb + b) * 4 ..]: {first position | rounds << 28, end position, slot offset, -} of solver b's first chunk of class l
    // (an empty range when the class has fewer chunks): ONE load at an address that follows from (l, b), taken a class ahead
    // of the slots it describes -- cptr -> chunk offsets -> slots was a chain of three round trips in front of every class
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const i32x4* cd = reinterpret_cast<const i32x4*>(cdesc);
    TriPre<TRI_PFMAX> nxt;
    int nlo = 0, nhi = 0;
    i32x4 dn = cd[(size_t)0 * nb + b];                       // descriptor of class 0 ...
    auto fetch_first = [&](int l) {   // this solver's first chunk of class l, from the descriptor at hand; then the descriptor of l + 1
        nlo = nhi = 0;
        if (l < nlev) {
            nlo = dn[0] & TRI_POS_MASK; nhi = dn[1];
            if (nhi > nlo) tri_fetch<L, TRI_PFMAX>(B, nxt, nlo, nhi, (int)((unsigned)dn[0] >> 28), dn[2], rloc, sl);
            if (l + 1 < nlev) dn = cd[(size_t)(l + 1) * nb + b];
        }
    };
    auto run = [&](const TriPre<TRI_PFMAX>& r, int lo, int hi) {
        const int p = lo + rloc;
        if (p < hi) {
            const double un = tri_row<L, TRI_PFMAX, true>(a, r, p, sl, ldw);
            if (sl == L - 1) {
                if (same_xcd) a.W[p] = un;
                else __hip_atomic_store((gu64*)(a.W + p), (unsigned long long)__double_as_longlong(un), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
#ifdef TC_TIMING
    unsigned long long tct[5] = {0, 0, 0, 0, 0}, tcl = __builtin_amdgcn_s_memrealtime();
#define TCT(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); tct[k] += n_ - tcl; tcl = n_; } while (0)
#else
#define TCT(k)
#endif
    fetch_first(0);
    for (int l = 0; l < nlev; ++l) {
        TCT(4);
        if (nhelp && b == 0 && tid == 0) __hip_atomic_store((gu32*)(sync + 2), (unsigned)cptr[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const TriPre<TRI_PFMAX> cur = nxt;
        if (nhi > nlo && rloc < nhi - nlo) { asm volatile("" :: "v"(cur.c[0]), "v"(cur.row)); }
        TCT(0);
        run(cur, nlo, nhi);
        TCT(1);
        for (int c = cptr[l] + b + nb; c < cptr[l + 1]; c += nb) {   // classes of more chunks than solvers: fetched on the spot
            TriPre<TRI_PFMAX> r;
            const int d0 = a.lptr[c], lo = d0 & TRI_POS_MASK, hi = a.lptr[c + 1] & TRI_POS_MASK;
            tri_fetch<L, TRI_PFMAX>(B, r, lo, hi, (int)((unsigned)d0 >> 28), a.sbase[c], rloc, sl);
            run(r, lo, hi);
        }
        // drain the stores, arrive; the next class's fetch travels during the wait
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TCT(2);
        fetch_first(l + 1);
        __syncthreads();
        ++round;
        if (tid == 0) {
            __hip_atomic_fetch_add(g_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)nb * round;
            int ok = 1;
            unsigned spins = 0;
            unsigned long long t0 = 0;
            while (__hip_atomic_load(g_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if ((++spins & 1023u) == 0u) {
                    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                    if (!t0) t0 = now;
                    else if (now - t0 > 200000000ull) {
                        __hip_atomic_store(g_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = 0;
                        break;
                    }
                }
            }
            if (__hip_atomic_load(g_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ok = 0;
            s_ok = ok;
        }
        __syncthreads();
        TCT(3);
        if (!s_ok) return;
    }
#ifdef TC_TIMING
    if (tid == 0 && b == 0 && nlev > 100) printf("[tri_cluster] %d classes, %d solvers: wait for slots %.2f, row chain %.2f, drain %.2f, barrier %.2f, rest %.2f us per class\n", nlev, nb, tct[0] * 0.01 / nlev, tct[1] * 0.01 / nlev, tct[2] * 0.01 / nlev, tct[3] * 0.01 / nlev, tct[4] * 0.01 / nlev);
#endif
    // u_i <- W_p, a slice per solver (every W is final and visible: the last barrier)
    for (int p = b * TRI_BLOCK + tid; p < nseq; p += nb * TRI_BLOCK) a.u[a.order[p]] = ldw(p);
}

}  // namespace fasp
