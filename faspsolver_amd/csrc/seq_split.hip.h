// seq_split.hip.h -- the sequential sweeps (Gauss-Seidel / SOR family, ItrSmootherCSR.c:251-334, :932-1040) as
//   (1) ONE fully parallel pass over everything a row reads that the sweep has not touched yet, and
//   (2) a sparse lower-triangular solve over what remains, run as a DATAFLOW over the whole chip (round 4).
// Part of the single translation unit solver.hip (after small_solvers.hip.h; not a stand-alone header).
//
// The reference's sweep over the rows i_0, i_1, ... computes, row after row,
//     t = b_i - sum_{j != i} a_ij u_j,   u_i = t / a_ii          (SOR: u_i = w t / a_ii + (1 - w) u_i)
// where u_j is the NEW value for the rows j the sweep visited before i and the OLD one otherwise.  The old values do
// not change while the sweep runs, so their part of every row sum can be formed up front, for all rows at once:
//     W_p = b_i - sum_{j not swept before i} a_ij u_j^old        (k_split_rest: a plain gather SpMV, any order)
// and what remains is the triangular recurrence over the sweep's own rows, numbered p = 0, 1, ... :
//     W_p <- (W_p - sum_{q coupled, swept before p} l_pq W_q) / a_ii          (k_tri_flow / k_tri_level)
// The dependency chain carries only the "lower" entries (about half of a row; a quarter in a C-row or F-row sweep) and
// the anti-dependencies (row j reads the old u_i of a later row) vanish because pass (1) has read every old value
// before anything is written.
// Arithmetic: t is formed as (b_i - rest) - lower instead of b_i - (all entries in storage order): a regrouping of
// the same products (differences of a few ulp per row; the tests pin iteration counts and |relres - ref| <= 1e-10
// against the reference's own runs, tests/golden/p7_sweeps*.npz).
//
// Round 4: the triangular solve is a point-to-point dataflow instead of barriers and launches (rounds 2-3: one launch per
// dependency class, one workgroup with a barrier per chunk, a cluster of workgroups with a barrier per class).
//   * STRIPS.  The sweep sequence is cut into contiguous ranges ("strips": for a grid in natural order, slabs of grid
//     planes) of a few hundred KB of lower entries each, bounded by what the strip's own values plus the values it reads
//     from earlier strips ("ghosts") take in LDS.  A strip is solved by ONE workgroup; workgroups draw strips from a
//     ticket counter in order, so a strip only ever waits for strips that are already running or done: no residency
//     assumption, no deadlock, any number of strips.
//   * A VALUE IS ITS OWN FLAG.  W is set to a signalling-NaN sentinel by pass (1); a finished row overwrites it with one
//     naturally aligned 8-byte store (arithmetic cannot produce the sentinel's bits).  Inside a strip the values live in
//     LDS, between strips they travel through W in memory (write-through stores / L1-bypassing loads): a consumer polls
//     the value itself.  No flags, no fences, no barriers, no drained counters.
//   * ROLES.  Waves 0 and 1 of a workgroup import the strip's ghosts (they poll W in memory, in the order the strip needs
//     them, into LDS), the other waves compute: wave w takes the strip's chunks w, w + nw, ... (a chunk = 64 / L rows of
//     one dependency class, L lanes per row, ordered by class), its operands are LDS indices, the slots of its NEXT
//     chunk(s) are in flight while it waits for the operands of this one; a finished row goes to LDS, to W (one
//     write-through store, nothing waits for it) and to u_i.  A chunk first polls ONE word -- the operand the host expects
//     last -- and only then looks at all of them: waiting waves cost the LDS one broadcast read per turn.
//     A chunk waits only for the rows it actually reads, so a wide class is as many chunks in flight as there are waves
//     on the chip, and a chain of narrow classes advances at an LDS round trip per row.
// k_tri_level is the plain form of the same arithmetic (one launch per dependency class, one wavefront per chunk, W in
// memory): the fallback when the dataflow is switched off (fasp_hip_tune("seq_flow", 0)) or has reported a timeout,
// and the A/B partner of the bit-identity test.
//
// Storage of the lower part.  Per chunk of nl = rows * L lanes, at a 16-byte-aligned offset from the strip's base:
//     nl * 16 bytes   columns: eight 16-bit LDS indices per lane
//     PF / 2 - (PF - pf) / 2 planes of nl * 16 bytes   values: rounds 2g, 2g + 1 of a lane side by side, g >= (PF - pf) / 2
// (PF = 4 or 8 rounds per schedule, pf <= PF = what the longest row of the chunk needs).  A row's lower entries are sorted by
// dependency class and RIGHT-ALIGNED in the PF rounds of L lanes: its last L entries -- the ones it waits for -- are round
// PF - 1, the L before them round PF - 2, ...; the leading rounds of a short row are empty.  Everything a chunk needs sits at addresses that follow
// from its 8-byte descriptor: one memory round trip, perfectly coalesced, one chunk ahead.  Unused slots hold (index of
// a constant 0.0, value 0).  A row with more lower entries than its slots hold hands the oldest to VIRTUAL ROWS (seq_sched.h):
// work items of their own in the same strip whose value is the sum of their products, read by the row as one operand each; on
// chain-bound schedules a row's last two entries sit in its last lane (the SPINE, seq_sched.h).
// The per-row scalars travel as three small records: (b - rest, old u_i) from pass (1), (a_ii, 1 / a_ii) and
// (flags, row index) from the schedule.
#pragma once
#include "seq_sched.h"

namespace fasp {

struct FlowArgs {
    const FlowStrip*     strips;
    const int4*          chunks;   // {first local row | rows << 16 | rounds << 24, offset of the slots in 16-byte units, LDS index of the operand expected last, -}
    const unsigned char* slots;
    const int*           gpos;
    const int*           cstrip;   // strip of a chunk (k_tri_level)
    const double*        rec;      // 2 doubles per position, written by pass (1): b - rest, old u_i
    const double*        dr;       // 2 doubles per position: a_ii, 1 / a_ii rounded to nearest (0 for a row that is left alone)
    const int*           tr;       // 2 ints per position: (row left alone) << 31 | (virtual row: FLOW_VIRTUAL), row index (-1: virtual row)
    double*              W;        // the new iterate of the swept rows, by position
    double*              u;        // the level's iterate (scatter target)
    unsigned*            sync;     // [0] ticket counter, [1] error word
    int                  nstrips;
    int                  form;     // 0  u_i = t * (1/a_ii)   1  u_i = t / a_ii   2  u_i = w (t / a_ii) + (1 - w) u_i
    int                  kt;       // spine rounds of the schedule (seq_sched.h): 0 or TRI_SPINE
    double               w;
};
constexpr unsigned long long FLOW_SENT = 0x7FF4DEADBEEF0001ull;   // a signalling NaN: no arithmetic result carries these bits

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

// pass (1): rec_p = (b_i - (entries of row i that read old values), u_i), W_p = not yet; L lanes per row, grid-stride
template <int L>
__global__ __launch_bounds__(BLOCK) void k_split_rest(int nseq, const int* __restrict__ tr, const int* __restrict__ ria,
                                                       const int* __restrict__ rja, const double* __restrict__ rval,
                                                       const double* __restrict__ b, const double* __restrict__ u,
                                                       double* __restrict__ rec, double* __restrict__ W, unsigned* __restrict__ sync,
                                                       double* __restrict__ G2 = nullptr, int npad = 0, int zero_old = 0)
{
    // zero_old (round 5): the level's vector is all zeros -- the first sweep of a pre-smoothing step on every level of a cycle.  Every
    // product of this pass is a zero then and the record is b_i itself: the rows' entries are not read (the reference subtracts the
    // same zeros one by one).
    constexpr int RPB = BLOCK / L;
    if (blockIdx.x == 0 && threadIdx.x == 0) { sync[0] = 0u; sync[1] = 0u; sync[2] = 0u; sync[4] = 0u; sync[5] = 0u; sync[8] = 0u; sync[9] = 0u; }   // ticket counter and error word of the dataflow solve that follows (chain form: + the tier-2 ticket)
    if (G2)   // chain form (seq_chain.hip.h): tier 2's sums are "not yet" too, padding rows included
        for (int p = blockIdx.x * BLOCK + threadIdx.x; p < npad; p += gridDim.x * BLOCK) G2[p] = __longlong_as_double((long long)FLOW_SENT);
    const int sl = threadIdx.x & (L - 1);
    const int rloc = threadIdx.x / L;
    for (int p0 = blockIdx.x * RPB; p0 < nseq; p0 += gridDim.x * RPB) {   // (whole wavefronts walk the loop: the DPP moves read neighbours)
        const int p = p0 + rloc;
        const bool on = p < nseq;
        const int kb = (on && !zero_old) ? ria[p] : 0, ke = (on && !zero_old) ? ria[p + 1] : 0;
        double s = seq_row_sum<L>(rja, rval, kb + sl, ke, -1, [&](int c) { return u[c]; });
        s = group_sum_last<L>(s);
        if (on && sl == L - 1) {
            const int r = tr[2 * (size_t)p + 1];
            f64x2_t o;
            if (r >= 0) { o[0] = b[r] - s; o[1] = u[r]; }
            else { o[0] = 0.0; o[1] = 0.0; }   // a virtual row (seq_sched.h): no right-hand side, no old value
            if ((unsigned long long)__double_as_longlong(o[1]) == FLOW_SENT) o[1] = __longlong_as_double((long long)(FLOW_SENT | 0x0008000000000000ull));   // (an untouched u_i must not look like "not there yet")
            *reinterpret_cast<f64x2_t*>(rec + 2 * (size_t)p) = o;
            W[p] = __longlong_as_double((long long)FLOW_SENT);
        }
    }
}

// A sweep whose rows do not couple at all (no row reads another row of the sweep, earlier or later: the C rows / the F rows of a
// 7-point level 0): ONE pass, the update in place -- no records, no W, no scatter pass (round 5: 350 -> 250 us per sweep of 8.4 M rows)
template <int L>
__global__ __launch_bounds__(BLOCK) void k_split_direct(int nseq, const int* __restrict__ tr, const int* __restrict__ ria, const int* __restrict__ rja,
                                                         const double* __restrict__ rval, const double* __restrict__ dr, const double* __restrict__ b,
                                                         double* u, int form, double w, int zero_old = 0)
{
    constexpr int RPB = BLOCK / L;
    const int sl = threadIdx.x & (L - 1);
    const int rloc = threadIdx.x / L;
    for (int p0 = blockIdx.x * RPB; p0 < nseq; p0 += gridDim.x * RPB) {
        const int p = p0 + rloc;
        const bool on = p < nseq;
        const int kb = (on && !zero_old) ? ria[p] : 0, ke = (on && !zero_old) ? ria[p + 1] : 0;   // (zero_old: as k_split_rest)
        double s = seq_row_sum<L>(rja, rval, kb + sl, ke, -1, [&](int c) { return u[c]; });   // (columns outside the sweep: nobody writes them in this launch)
        s = group_sum_last<L>(s);
        if (on && sl == L - 1) {
            const int r = tr[2 * (size_t)p + 1];
            if (r >= 0) {
                const f64x2_t d = *reinterpret_cast<const f64x2_t*>(dr + 2 * (size_t)p);
                u[r] = tri_update(b[r] - s, d[0], d[1], tr[2 * (size_t)p] < 0, form, w, u[r]);
            }
        }
    }
}

// u_i <- W_p (final == 0), or the update of a sweep without any lower entry straight from pass (1) (final == 1)
__global__ __launch_bounds__(BLOCK) void k_split_scatter(int nseq, FlowArgs a, int final)
{
    for (int p = blockIdx.x * BLOCK + threadIdx.x; p < nseq; p += gridDim.x * BLOCK) {
        double v;
        if (final) v = tri_update(a.rec[2 * (size_t)p], a.dr[2 * (size_t)p], a.dr[2 * (size_t)p + 1], a.tr[2 * (size_t)p] < 0, a.form, a.w, a.rec[2 * (size_t)p + 1]);
        else v = a.W[p];
        const int r = a.tr[2 * (size_t)p + 1];
        if (r >= 0) a.u[r] = v;
    }
}

// what a lane holds of a chunk before the chain reaches it: PF slots and the row's records
template <int PF>
struct FlowSet { unsigned cw[PF / 2]; double v[PF]; double t, uo, d, rd; int tn, row; int lo, n, pf, wait; };

struct FlowBufs { __amdgpu_buffer_rsrc_t slots, rec, dr, tr; };
__device__ __forceinline__ FlowBufs flow_bufs(const FlowArgs& a, const FlowStrip& S)
{
    FlowBufs B;   // per strip: 32-bit offsets stay small whatever the size of the level
    B.slots = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.slots + S.slot0), 0, 0x7fffffff, 0x00020000);
    B.rec = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.rec + 2 * (size_t)S.row0), 0, 0x7fffffff, 0x00020000);
    B.dr = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.dr + 2 * (size_t)S.row0), 0, 0x7fffffff, 0x00020000);
    B.tr = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(a.tr + 2 * (size_t)S.row0), 0, 0x7fffffff, 0x00020000);
    return B;
}
// Branch-free, and the same number of loads whatever the chunk looks like: idle lanes, and value planes the chunk does not
// have, put their offset beyond the buffers' range -- such a lane gets 0 back and causes no memory access.
constexpr int FLOW_OOR = (int)0x80000000u;   // (the resources declare 2^31 - 1 bytes)
template <int L, int PF>
__device__ __forceinline__ void flow_fetch(const FlowBufs& B, FlowSet<PF>& r, int4 d, int lane)
{
    static_assert(PF == 4 || PF == 8, "slots come in rounds of four or eight");
    r.lo = d.x & 0xffff; r.n = (d.x >> 16) & 0xff; r.pf = (d.x >> 24) & 0xf; r.wait = d.z;
    const int nl = r.n * L;
    const bool on = lane < nl;
    const int base = d.y * 16;
    const int off = on ? base + lane * 16 : FLOW_OOR;
    if constexpr (PF == 8) {
        const u32x4_t cq = __builtin_amdgcn_raw_buffer_load_b128(B.slots, off, 0, 0);
        r.cw[0] = cq[0]; r.cw[1] = cq[1]; r.cw[2] = cq[2]; r.cw[3] = cq[3];
    } else {
        const u32x2_t cq = __builtin_amdgcn_raw_buffer_load_b64(B.slots, off, 0, 0);
        r.cw[0] = cq[0]; r.cw[1] = cq[1];
    }
    const int g0 = (PF - r.pf) >> 1;   // planes in front of g0 are not stored: rounds that no row of the chunk uses
#pragma unroll
    for (int g = 0; g < PF / 2; ++g) {
        const f64x2_t v = __builtin_bit_cast(f64x2_t, __builtin_amdgcn_raw_buffer_load_b128(B.slots, g >= g0 ? off : FLOW_OOR, (1 + g - g0) * nl * 16, 0));
        r.v[2 * g] = v[0]; r.v[2 * g + 1] = v[1];
    }
    const int rloc = lane / L;
    const unsigned po = on ? (unsigned)(r.lo + rloc) * 16u : (unsigned)FLOW_OOR;
    const f64x2_t r0 = buf_load_f64x2(B.rec, po), r1 = buf_load_f64x2(B.dr, po);
    const u32x2_t r2 = __builtin_amdgcn_raw_buffer_load_b64(B.tr, on ? (r.lo + rloc) * 8 : FLOW_OOR, 0, 0);
    r.t = r0[0]; r.uo = r0[1]; r.d = r1[0]; r.rd = r1[1];
    r.tn = (int)r2[0]; r.row = (int)r2[1];
}
template <int PF>
__device__ __forceinline__ int flow_col(const FlowSet<PF>& r, int q) { return (int)((q & 1) ? (r.cw[q >> 1] >> 16) : (r.cw[q >> 1] & 0xffffu)); }

// the row arithmetic shared by both forms (identical bits): s = the lane's slot products in slot order
// (summed left to right from 0.0) over the rounds in front of the spine; the DPP tree; in the group's last lane t - s, minus the
// spine products one after the other (KT = 0: none), and the update.  Returns the new value (valid in lane L - 1).
template <int L, int KT>
__device__ __forceinline__ double flow_row(const FlowArgs& a, double t, double uo, double d, double rd, int tn, double s, double v0 = 0.0, double x0 = 0.0, double v1 = 0.0, double x1 = 0.0)
{
    static_assert(KT == 0 || KT == 2, "no spine, or two rounds of it");
    s = group_sum_last<L>(s);
    double T = t - s;
    if (KT) { T = __builtin_fma(-v0, x0, T); T = __builtin_fma(-v1, x1, T); }
    const double un = tri_update(T, d, rd, false, a.form, a.w, uo);
    // a row that is left alone keeps its value (pass (1) made sure it does not look like "not there yet"); a virtual row is the sum itself (t = 0)
    return tn < 0 ? uo : (tn & FLOW_VIRTUAL) ? -T : un;
}

// descriptors are read through the scalar cache (constant address space: nothing in a launch writes them)
typedef const __attribute__((address_space(4))) int* flow_int_cp;
__device__ __forceinline__ FlowStrip flow_strip(const FlowArgs& a, int s)
{
    const flow_int_cp q = (flow_int_cp)(unsigned long long)(a.strips + s);
    FlowStrip S;
    S.slot0 = (long long)(((unsigned long long)(unsigned)q[1] << 32) | (unsigned)q[0]);
    S.row0 = q[2]; S.nrows = q[3]; S.chunk0 = q[4]; S.nchunk = q[5]; S.ghost0 = q[6]; S.nghost = q[7];
    return S;
}
__device__ __forceinline__ int4 flow_chunk(const FlowArgs& a, int c)
{
    const flow_int_cp q = (flow_int_cp)(unsigned long long)(a.chunks + c);
    return make_int4(q[0], q[1], q[2], q[3]);
}

// ONE dependency class per launch, one wavefront per chunk, W in memory (plain loads and stores: launches order them)
template <int L, int PF>
__global__ __launch_bounds__(64) void k_tri_level(FlowArgs a, const int* __restrict__ lchunks, int c0)
{
    const int lane = threadIdx.x;
    const int ck = lchunks[c0 + blockIdx.x];
    const FlowStrip S = flow_strip(a, __builtin_amdgcn_readfirstlane(a.cstrip[ck]));
    const FlowBufs B = flow_bufs(a, S);
    FlowSet<PF> r;
    flow_fetch<L, PF>(B, r, flow_chunk(a, __builtin_amdgcn_readfirstlane(ck)), lane);
    const int rloc = lane / L, sl = lane & (L - 1);
    auto ldw = [&](int c) -> double {   // LDS index -> position
        if (c < S.nrows) return a.W[S.row0 + c];
        if (c < S.nrows + S.nghost) return a.W[a.gpos[S.ghost0 + c - S.nrows]];
        return 0.0;
    };
    if (rloc < r.n) {
        const int p = S.row0 + r.lo + rloc;
        double s = 0.0;
        double un;
        if (a.kt) {
#pragma unroll
            for (int q = 0; q < PF - TRI_SPINE; ++q) s += r.v[q] * ldw(flow_col(r, q));
            un = flow_row<L, TRI_SPINE>(a, r.t, r.uo, r.d, r.rd, r.tn, s, r.v[PF - 2], ldw(flow_col(r, PF - 2)), r.v[PF - 1], ldw(flow_col(r, PF - 1)));
        } else {
#pragma unroll
            for (int q = 0; q < PF; ++q) s += r.v[q] * ldw(flow_col(r, q));
            un = flow_row<L, 0>(a, r.t, r.uo, r.d, r.rd, r.tn, s);
        }
        if (sl == L - 1) a.W[p] = un;
    }
}

// ---------------------------------------------------------------------------
// k_tri_flow<L, PF>: the dataflow solve (header of this file).
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool flow_ready(double v) { return (unsigned long long)__double_as_longlong(v) != FLOW_SENT; }
// a waiter that has spun for two seconds (or sees that somebody else has) raises the error word and goes on with what it
// has: the launch ends, the host fails the solve loudly and stops using this form (smoothers.hip.h)
__device__ __forceinline__ bool flow_give_up(unsigned* sync, unsigned& spins, unsigned long long& t0)
{
    typedef __attribute__((address_space(1))) unsigned gu32;
    if ((++spins & 4095u) != 0u) return false;
    if (__hip_atomic_load((gu32*)(sync + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true;
    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
    if (!t0) { t0 = now; return false; }
    if (now - t0 > 200000000ull) {   // 2 s at 100 MHz
        __hip_atomic_store((gu32*)(sync + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ... and the process-wide word in host-mapped memory (its device address sits in sync[6..7], written at upload): the host reads
        // it without a copy -- round 4 queued a 4-byte device-to-host copy behind every sweep (466 per GS-default solve of P7(256))
        unsigned* herr = reinterpret_cast<unsigned*>(((unsigned long long)sync[7] << 32) | sync[6]);
        if (herr) __hip_atomic_store(herr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return true;
    }
    return false;
}

// Workgroup size and register sets per compute wave: 1024-thread workgroups (14 compute waves, 128 registers each) with two sets
// -- the slots of a wave's NEXT chunk are in flight -- for chunks of four and of eight rounds alike (measured at 128^3 against
// 512-thread workgroups with three sets for eight rounds: 6 compute waves do not keep up with classes of two or three rows of
// 450 entries; sweep pair over all levels 3.96 -> 3.36 ms).
// importer waves per workgroup / ghosts per lane and batch: levels with wide classes (chunks of four rounds) import as many values
// as they compute -- four waves with one load in flight per lane pick a ghost up sooner than two with four (measured at 128^3:
// level 1 556 -> 497 us, level 2 428 -> 388 us per ascending sweep); the deep levels keep their waves for the rows
#ifndef FLOW_NIMP4
#define FLOW_NIMP4 4
#define FLOW_GB4 1
#define FLOW_NIMP8 3
#define FLOW_GB8 2
#endif
#ifndef FLOW_NT8
#define FLOW_NT8 1024
#define FLOW_NSET8 2
#endif
template <int PF> struct FlowGeom { static constexpr int NT = PF <= 4 ? 1024 : FLOW_NT8, NSET = PF <= 4 ? 2 : FLOW_NSET8, NIMP = PF <= 4 ? FLOW_NIMP4 : FLOW_NIMP8, GB = PF <= 4 ? FLOW_GB4 : FLOW_GB8; };
#ifdef FLOW_TIMING
__device__ unsigned long long g_flow_times[8192];   // start / end of every strip of the last launch (100 MHz clock)
#endif
template <int L, int PF>
__global__ __launch_bounds__(FlowGeom<PF>::NT) void k_tri_flow(FlowArgs a)
{
    constexpr int FLOW_THREADS = FlowGeom<PF>::NT, NSET = FlowGeom<PF>::NSET;
    typedef __attribute__((address_space(1))) unsigned           gu32;
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    extern __shared__ __attribute__((aligned(16))) double flow_lds[];
    __shared__ int s_strip;
    constexpr int NW = FLOW_THREADS / 64, NIMP = FlowGeom<PF>::NIMP, GB = FlowGeom<PF>::GB, NWC = NW - NIMP;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const double sent = __longlong_as_double((long long)FLOW_SENT);
    auto lds_get = [&](int c) -> double { return __hip_atomic_load(&flow_lds[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto lds_put = [&](int c, double v) { __hip_atomic_store(&flow_lds[c], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    for (;;) {
        if (tid == 0) s_strip = (int)__hip_atomic_fetch_add((gu32*)a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int s = __builtin_amdgcn_readfirstlane(s_strip);
        if (s >= a.nstrips) break;
        const FlowStrip S = flow_strip(a, s);
        const int nent = S.nrows + S.nghost;
#ifdef FLOW_TIMING
        const unsigned long long strip_t0 = __builtin_amdgcn_s_memrealtime();
#endif
        for (int i = tid; i < nent; i += FLOW_THREADS) flow_lds[i] = sent;
        if (tid == 0) flow_lds[nent] = 0.0;
        __syncthreads();
        if (wave < NIMP) {
            // ---- importers: the strip's ghosts, in the order the strip needs them, 64 GB at a time, batches dealt to the importer waves in turn;
            // a ghost goes to LDS the moment it is seen
            for (int g0 = wave * 64 * GB; g0 < S.nghost; g0 += NIMP * 64 * GB) {
                int gp[GB];
                unsigned pend = 0u;
#pragma unroll
                for (int k = 0; k < GB; ++k) {
                    const int g = g0 + k * 64 + lane;
                    gp[k] = g < S.nghost ? a.gpos[S.ghost0 + g] : 0;
                    if (g < S.nghost) pend |= 1u << k;
                }
                unsigned spins = 0;
                unsigned long long t0 = 0;
                for (;;) {
                    unsigned long long v[GB];   // (only what is still missing is asked for: every chip-wide poll is a fabric round trip)
#pragma unroll
                    for (int k = 0; k < GB; ++k) v[k] = ((pend >> k) & 1u) ? __hip_atomic_load((gu64*)(a.W + gp[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : FLOW_SENT;
#pragma unroll
                    for (int k = 0; k < GB; ++k)
                        if (((pend >> k) & 1u) && v[k] != FLOW_SENT) { lds_put(S.nrows + g0 + k * 64 + lane, __longlong_as_double((long long)v[k])); pend &= ~(1u << k); }
                    if (!__builtin_amdgcn_ballot_w64(pend != 0u)) break;
                    if (flow_give_up(a.sync, spins, t0)) {
#pragma unroll
                        for (int k = 0; k < GB; ++k) if ((pend >> k) & 1u) lds_put(S.nrows + g0 + k * 64 + lane, 0.0);
                        break;
                    }
                }
            }
        } else {
            // ---- compute: chunks w, w + NWC, ... of the strip; the next chunk's slots travel while this one waits for its operands
            const int w = wave - NIMP;
            const int rloc = lane / L, sl = lane & (L - 1);
            const FlowBufs B = flow_bufs(a, S);
            const int zero_idx = nent;
#ifdef FLOW_TIMING
            long long ft[5] = {0, 0, 0, 0, 0}, fl = clock64();
#define FT(k) do { const long long n_ = clock64(); ft[k] += n_ - fl; fl = n_; } while (0)
#else
#define FT(k)
#endif
            auto fetch = [&](FlowSet<PF>& X, int ci) {   // (past the end: an empty chunk, every load out of range)
                flow_fetch<L, PF>(B, X, ci < S.nchunk ? flow_chunk(a, S.chunk0 + ci) : make_int4(0, 0, zero_idx, 0), lane);
            };
            auto run = [&](const FlowSet<PF>& X, auto ktc) {
                constexpr int KT = decltype(ktc)::value, PB = PF - KT;   // spine rounds (seq_sched.h), rounds in front of them
                const bool on = rloc < X.n;
                const int pl = X.lo + rloc;
                double x[PF];
                unsigned pend = on ? (((1u << X.pf) - 1u) << (PF - X.pf)) : 0u;   // the chunk's rounds are the last pf of PF
                unsigned spins = 0;
                unsigned long long t0 = 0;
                FT(4);
                // every round of every lane once (unused ones point at the constant), back to back: most operands are old
#pragma unroll
                for (int q = 0; q < PF; ++q) {
                    x[q] = lds_get(flow_col(X, q));
                    if (flow_ready(x[q])) pend &= ~(1u << q);
                }
                double s = 0.0;
                FT(2);
                // The entries of a row sit in the order of their dependency classes, right-aligned in the PB rounds: what is still
                // missing is in the LAST round(s).  The rounds that are complete are summed while the wave waits.
                int qr = PB;   // first round somebody still waits for (wave-uniform)
#pragma unroll
                for (int q = PB - 1; q >= 0; --q)
                    if (__builtin_amdgcn_ballot_w64(((pend >> q) & 1u) != 0u)) qr = q;
#pragma unroll
                for (int q = 0; q < PB; ++q)
                    if (q < qr) s += X.v[q] * x[q];
                // a tight loop per round somebody misses (read, compare, branch; wave-uniform control flow: every lane reads, the
                // lanes that miss decide)
                auto await = [&](int q) {
                    const bool miss = ((pend >> q) & 1u) != 0u;
                    if (__builtin_amdgcn_ballot_w64(miss)) {
                        const int c = flow_col(X, q);
                        double y;
                        for (;;) {
                            y = lds_get(c);
                            if (!__builtin_amdgcn_ballot_w64(miss && !flow_ready(y))) break;
                            if (flow_give_up(a.sync, spins, t0)) break;
                        }
                        if (miss) x[q] = y;
                    }
                };
                if (qr < PB) {
                    // round after round, oldest first.  The waits for the older rounds end while the chain is still on its way;
                    // only the last one is part of it.
#pragma unroll
                    for (int q = 0; q < PB; ++q)
                        if (q >= qr) await(q);
                    if constexpr (KT == 0) {
                        __builtin_amdgcn_s_setprio(3);   // from here to the store of the row's value the wave is the chain
                        switch (qr) {   // the remaining rounds, in order
                            case 0: s += X.v[0] * x[0]; [[fallthrough]];
                            case 1: s += X.v[1] * x[1]; [[fallthrough]];
                            case 2: s += X.v[2] * x[2]; [[fallthrough]];
                            case 3: s += X.v[3] * x[3]; if (PF == 4) break; [[fallthrough]];
                            case 4: s += X.v[PF - 4] * x[PF - 4]; [[fallthrough]];
                            case 5: s += X.v[PF - 3] * x[PF - 3]; [[fallthrough]];
                            case 6: s += X.v[PF - 2] * x[PF - 2]; [[fallthrough]];
                            default: s += X.v[PF - 1] * x[PF - 1]; break;
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < PB; ++q)
                            if (q >= qr) s += X.v[q] * x[q];
                    }
                }
                FT(1);
                double un;
                if (KT) {
                    // the spine: everything else has met in the row's last lane before the operands the chain brings are there
                    const double S0 = group_sum_last<L>(s);
                    double Tt = X.t - S0;
                    await(PF - 2);
                    Tt = __builtin_fma(-X.v[PF - 2], x[PF - 2], Tt);
                    await(PF - 1);
                    __builtin_amdgcn_s_setprio(3);   // from here to the store of the row's value the wave is the chain
                    Tt = __builtin_fma(-X.v[PF - 1], x[PF - 1], Tt);
                    const double uu = tri_update(Tt, X.d, X.rd, false, a.form, a.w, X.uo);
                    un = X.tn < 0 ? X.uo : (X.tn & FLOW_VIRTUAL) ? -Tt : uu;
                } else un = flow_row<L, 0>(a, X.t, X.uo, X.d, X.rd, X.tn, s);
                if (on && sl == L - 1) {
                    lds_put(pl, un);
                    if (X.row >= 0) {   // (a virtual row's value is read inside its strip only and is no entry of u)
                        __hip_atomic_store((gu64*)(a.W + S.row0 + pl), (unsigned long long)__double_as_longlong(un), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (write-through: other strips poll it)
                        a.u[X.row] = un;
                    }
                }
                __builtin_amdgcn_s_setprio(0);
                FT(3);
            };
            const bool spine = PF == 8 && a.kt != 0;   // (schedules of four rounds have none: seq_sched.cpp)
            auto run_any = [&](const FlowSet<PF>& X) {
                if constexpr (PF == 4) run(X, std::integral_constant<int, 0>{});
                else if (spine) run(X, std::integral_constant<int, TRI_SPINE>{});
                else run(X, std::integral_constant<int, 0>{});
            };
            // A set is TOUCHED (its last load's result named in an empty asm) before the next fetch is issued: whatever wait
            // the compiler derives for the set lands in front of the younger loads, not behind them.
            FlowSet<PF> X0, X1, X2;
            fetch(X0, w);
            if (NSET == 3) fetch(X1, w + NWC);
            for (int ci = w; ci < S.nchunk; ci += NSET * NWC) {
                FT(4); asm volatile("" ::"v"(X0.tn)); FT(0);
                if (NSET == 3) fetch(X2, ci + 2 * NWC); else fetch(X1, ci + NWC);
                run_any(X0);
                if (ci + NWC >= S.nchunk) break;
                FT(4); asm volatile("" ::"v"(X1.tn)); FT(0);
                if (NSET == 3) fetch(X0, ci + 3 * NWC); else fetch(X0, ci + 2 * NWC);
                run_any(X1);
                if (NSET == 3) {
                    if (ci + 2 * NWC >= S.nchunk) break;
                    FT(4); asm volatile("" ::"v"(X2.tn)); FT(0);
                    fetch(X1, ci + 4 * NWC);
                    run_any(X2);
                }
            }
#ifdef FLOW_TIMING
            FT(4);
            if (lane == 0 && a.nstrips <= 64 && S.nchunk >= 6 * NWC) printf("[flow] strip %3d wave %2d: %4d chunks; per chunk of this wave: slots %.0f, last operand %.0f, all operands %.0f, row %.0f, rest %.0f cycles\n", s, w, (S.nchunk - w + NWC - 1) / NWC,
                                             (double)ft[0] * NWC / (S.nchunk), (double)ft[1] * NWC / S.nchunk, (double)ft[2] * NWC / S.nchunk, (double)ft[3] * NWC / S.nchunk, (double)ft[4] * NWC / S.nchunk);
#endif
        }
        __syncthreads();   // every role is done with this strip's LDS
#ifdef FLOW_TIMING
        if (tid == 0 && s < 4096) { g_flow_times[2 * s] = strip_t0; g_flow_times[2 * s + 1] = __builtin_amdgcn_s_memrealtime(); }
#endif
    }
}

}  // namespace fasp
