// seq_chain.hip.h -- the triangular solve of a CHAIN-BOUND sequential sweep (Gauss-Seidel / SOR family, ItrSmootherCSR.c:251-334,
// :432-693, :932-1040) as a blocked substitution whose dependency chain never leaves one wavefront (round 5).
// Part of the single translation unit solver.hip (after seq_split.hip.h; not a stand-alone header).  Layout and tiers: seq_sched.h.
//
// Why: on the deep levels of a 3-D hierarchy (rows of 200-700 lower entries, dependency classes of 2-11 rows) the dataflow form
// (k_tri_flow) pays 0.4-0.8 us per dependency CLASS -- every link of the chain is an LDS poll between wavefronts (216-346 ns
// measured against an 80 ns hand-off floor).  Here the chain is a register-to-register affair:
//     lane j of the CHAIN WAVE owns row 64 K + j of block K.  Step c = 0 .. 63: every lane forms its candidate update from its
//     accumulator (three dependent operations for the division; only lane c's is final), x_c is broadcast from lane c by
//     v_readlane, and every lane does   accA -= TA[j][c] x_c   (the block's own triangle; zero for j <= c, so accA_j keeps t_j
//     once it is final)   and   accB -= TB[j][c] x_c   (the NEXT block's rows against this block).  One link = one multiply-add,
//     the update, one readlane: no poll, no LDS, no cross-lane sum.  After 64 steps the block's 64 values are the update of accA
//     over all lanes at once (the same operations on the same inputs as the per-step candidates: identical bits).
// What is not in the band -- columns older than the previous block -- reaches the chain as TWO numbers per row (S1, G2), formed ahead
// of time by other wavefronts and taken from rings in LDS at the block boundary:
//     tier 1 (the n1b blocks in front of the band): helper waves of the chain's workgroup, x through a ring in LDS that the chain
//            wave fills block by block; a helper may run CHAIN_HA blocks ahead; sums to the ring S1;
//     tier 2 (everything older): the other workgroups of the launch (those on the chain's XCD step aside: they would stream through
//            the L2 the band planes are waiting in), x through memory (W, written by the EXPORTER wave of the chain's workgroup),
//            sums through memory (G2), brought into the ring G2r by the IMPORTER wave.  Tier 2 has n1b blocks of slack, tier 1 one.
// Nobody polls operands: the schedule tells every group of eight steps the newest block it reads, and a wave waits on ONE word -- blocks
// in the ring (LDS) / blocks exported (memory) -- before it gathers; entries are in column order, so only a block's last groups wait.
// S1 and G2 -- one number per row each -- are their own flags (signalling-NaN sentinel), as the values of the dataflow form are.
// Roles are dealt by tickets (the workgroup that draws ticket 0 is the chain's: it is running by definition; blocks of both tiers
// are drawn in order, so every block the chain waits for is in the hands of a running wave): no residency assumption.
// k_tri_chain_ref is the plain form of the same arithmetic -- ONE wavefront, block after block, no polling -- the fallback after a
// reported time-out (tests/test_gpu_scale.py::test_chain_form_time_out_fails_loudly_and_falls_back) and the A/B partner of the
// bit-identity test (test_chain_form_kernels_agree_bit_for_bit).  Measurements and what was tried: profiles/r05_gs_chain.txt.
#pragma once
#include "seq_sched.h"

namespace fasp {

struct ChainArgs {
    const f64x2_t*        band;   // (nb * 64 + CHAIN_PF) steps x 64 lanes x (TA, TB)
    const f64x2_t*        drd;    // npad x (a_ii, 1 / a_ii)
    const ChainBlk*       blk;
    const double*         t1v; const unsigned short* t1c; const int* t1need;   // values [step][lane]; columns [group of 8 steps][lane][8]; need: per group of eight steps, the newest BLOCK its entries read (-1: none)
    const double*         t2v; const unsigned short* t2c; const int* t2need;
    const double*         rec;    // 2 doubles per position, written by pass (1): b - rest, old u_i
    const int*            tr;     // 2 ints per position: flags, row index (-1: padding)
    double*               W;      // the new values by position (sentinel until written); W[npad] = 0.0 for the padding entries of tier 2; W[npad + 64 ..+ 128): scratch
    double*               G2;     // tier 2's sums by position (sentinel until written)
    double*               u;
    unsigned*             sync;   // [0] role ticket, [1] error word, [2] tier-2 block ticket, [4] blocks exported to W, [5] 1 + XCD of the chain's workgroup, [8] toucher ticket, [9] workgroups that stepped aside
    int                   nb, npad, rx, rg;
    int                   touch_lead; // > 0: a workgroup of its own on the chain's XCD pulls the band planes into the L2, this many blocks ahead of the exported count (0: the importer wave does, four blocks ahead)
    int                   touch_t1;   // the toucher also pulls tier 1's entries
    int                   has_t2; // 0: no row has a tier-2 entry (no tier-2 workgroups in the launch: tier 1 starts from pass (1)'s record itself)
    int                   form;   // as tri_update
    double                w;
};

#ifndef FASP_CHAIN_NT
#define FASP_CHAIN_NT 512
#endif
static_assert(FASP_CHAIN_NT >= 256 && FASP_CHAIN_NT <= 1024 && FASP_CHAIN_NT % 64 == 0, "the chain workgroup deals four roles to whole wavefronts (chain, exporter, importer, >= 1 helper)");
static_assert(64 % CHAIN_PF == 0 && CHAIN_PF <= 64 && CHAIN_PF >= 1, "chain_block's register ring walks a block's 64 steps in whole turns of CHAIN_PF");
static_assert(CHAIN_HA >= 1 && 64 * (48 + CHAIN_HA + 3) <= 65535, "tier 1's ring index is a 16-bit value (rx = 64 (n1b + CHAIN_HA + 3), n1b <= 48)");
// compiler-only ordering between a payload and its flag (both relaxed atomics to DIFFERENT addresses: the hardware keeps a wave's LDS
// operations, and its stores, in program order; nothing in the language keeps the compiler from swapping them -- ADVICE r5).  No instruction.
#define CHAIN_CFENCE() __atomic_signal_fence(__ATOMIC_SEQ_CST)
constexpr int CHAIN_NT = FASP_CHAIN_NT;   // 8 waves (256 registers each: the chain wave keeps CHAIN_PF coefficient pairs in flight): chain, exporter, importer, 5 tier-1 helpers (chain workgroup) / 8 tier-2 workers (the others)

template <int FORM>
__device__ __forceinline__ double chain_update(double t, double d, double rd, double w, double ku)
{
    if (FORM == 0) return t * rd;                  // t * (1.0 / a_ii)
    if (FORM == 1) return tri_div(t, d, rd);       // t / a_ii
    return w * tri_div(t, d, rd) + ku;             // w (t / a_ii) + (1 - w) u_i   (ku = (1 - w) u_i)
}
__device__ __forceinline__ double chain_bcast(double x, int c)   // lane c's value to every lane (c: a constant)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, c), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), c);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// the 64 steps of one block: coefficients through the ring cf (CHAIN_PF steps in flight; refilled from `next`)
template <int FORM>
__device__ __forceinline__ void chain_block(double& accA, double& accB, double d, double rd, double w, double ku, f64x2_t (&cf)[CHAIN_PF], const f64x2_t* next)
{
#pragma unroll
    for (int c = 0; c < 64; ++c) {
        const double x = chain_update<FORM>(accA, d, rd, w, ku);
        const double sx = chain_bcast(x, c);
        accA = __builtin_fma(-cf[c % CHAIN_PF][0], sx, accA);
        accB = __builtin_fma(-cf[c % CHAIN_PF][1], sx, accB);
        asm volatile("" : "+v"(accA), "+v"(accB));   // (both accumulators exist here: code sinking would otherwise park accB's operands in scratch)
        cf[c % CHAIN_PF] = next[(size_t)c * 64];     // (into the registers this step has just read)
        __builtin_amdgcn_sched_barrier(0);   // a step is a step: left alone, the scheduler gathers the 64 loads of a block up front (256 registers and
                                             // spills) and defers accB's multiply-adds behind a table of spilled broadcasts
    }
}


// One lane's share of a tier: n steps (a multiple of 8) of (value, column), acc <- fma(-value, x[column], acc) in step order, as a
// pipeline over groups of eight steps: the (value, column) pairs of the next CHAIN_NS - 1 groups travel (streams from HBM: one to two
// microseconds; 25 KB in flight per wave) while the operands of the next group are gathered (LDS, or memory) and this one is summed.
// GATING instead of polling the operands: `need` = the newest block a group reads (one int per group, from the schedule; -1: none);
// the wave waits -- on ONE word -- until that block has been published, then gathers without looking at what it gets.  Entries are
// in column order and right-aligned: only a block's last groups ever wait.
constexpr int CHAIN_NS = 6;   // register sets of (value, column) groups in flight
constexpr int CHAIN_NX = 3;   // operand sets: the gathers of the next two groups are in flight while this one is summed
// columns of a group: eight 16-bit indices per lane, side by side (one 16-byte load per lane and group)
template <bool STREAM, class Gate, class Get>
__device__ __forceinline__ void chain_tier_sum(const double* v, const u32x4_t* c, const int* need, int n, double& acc, Gate gate, Get get)
{
    constexpr int U = 8, NS = CHAIN_NS, NX = CHAIN_NX;
    static_assert(NS % NX == 0, "the operand sets rotate with the register sets");
    double vv[NS][U], xx[NX][U];
    u32x4_t cc[NS];
    int nn[NS];
    auto fetch = [&](double (&vs)[U], u32x4_t& cs, int& ns, int s) {
        if (s < n) {
#pragma unroll
            for (int i = 0; i < U; ++i) vs[i] = STREAM ? __builtin_nontemporal_load(v + (size_t)(s + i) * 64) : v[(size_t)(s + i) * 64];   // (tier 2 streams past the caches)
            cs = STREAM ? __builtin_nontemporal_load(c + (size_t)(s >> 3) * 64) : c[(size_t)(s >> 3) * 64];
            ns = need[s >> 3];
        }
    };
    auto gather = [&](double (&xs)[U], const u32x4_t& cs, int ns, int s) {
        if (s < n) {
            gate(__builtin_amdgcn_readfirstlane(ns));
#pragma unroll
            for (int i = 0; i < U; ++i) xs[i] = get((int)((i & 1) ? (cs[i >> 1] >> 16) : (cs[i >> 1] & 0xffffu)));
        }
    };
#pragma unroll
    for (int k = 0; k < NS - 1; ++k) fetch(vv[k], cc[k], nn[k], k * U);
    gather(xx[0], cc[0], nn[0], 0);
    gather(xx[1], cc[1], nn[1], U);
    for (int s0 = 0; s0 < n; s0 += NS * U) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int s = s0 + k * U;
            if (s < n) {
                fetch(vv[(k + NS - 1) % NS], cc[(k + NS - 1) % NS], nn[(k + NS - 1) % NS], s + (NS - 1) * U);
                gather(xx[(k + 2) % NX], cc[(k + 2) % NS], nn[(k + 2) % NS], s + 2 * U);
#pragma unroll
                for (int i = 0; i < U; ++i) acc = __builtin_fma(-vv[k][i], xx[k % NX][i], acc);
            }
        }
    }
}

__device__ __forceinline__ bool chain_spin(unsigned* sync, unsigned& spins, unsigned long long& t0) { return flow_give_up(sync, spins, t0); }

// ---- the chain wave.  LDS words it publishes: s_k = the block it is in (helpers run at most CHAIN_HA blocks ahead), s_done = blocks
// whose values are in the ring X (what tier 1 and the exporter wait for).  At a block boundary it takes the coming block's S1 (tier 1's
// sum, from a helper) and G2 (tier 2's, brought in by the porter wave) from their rings in LDS: values that are their own flags.
template <int FORM>
__device__ __forceinline__ void chain_wave(const ChainArgs& a, double* X, double* S1, double* G2r, int* s_k, int* s_done, int* s_exp, int lane)
{
    const double sent = __longlong_as_double((long long)FLOW_SENT);
    auto lds_get = [&](double* p) -> double { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto lds_put = [&](double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    f64x2_t cf[CHAIN_PF];
    const f64x2_t* bp = a.band + lane;
#pragma unroll
    for (int i = 0; i < CHAIN_PF; ++i) cf[i] = bp[(size_t)i * 64];
    unsigned spins = 0;
    unsigned long long t0 = 0;
#ifdef CHAIN_TIMING
    long long ct[4] = {0, 0, 0, 0}, cl = clock64();
#define CT(k) do { const long long n_ = clock64(); ct[k] += n_ - cl; cl = n_; } while (0)
#else
#define CT(k)
#endif
    auto take = [&](double* ring, int gi) -> double {   // this lane's number of the coming block; the slot is handed back for the block CHAIN_HA + 2 later
        double g;
        for (;;) {
            g = lds_get(ring + gi);
            if (!__builtin_amdgcn_ballot_w64(!flow_ready(g))) break;
            if (chain_spin(a.sync, spins, t0)) break;
        }
        lds_put(ring + gi, sent);
        return g;
    };
    auto take_G = [&](int gi, double g2, double s1) -> double {   // (G2 + S1: the association of k_tri_chain_ref); g2, s1: what an early look at the slots found
        if (__builtin_amdgcn_ballot_w64(!flow_ready(g2))) g2 = take(G2r, gi); else lds_put(G2r + gi, sent);
        CT(3);
        if (__builtin_amdgcn_ballot_w64(!flow_ready(s1))) s1 = take(S1, gi); else lds_put(S1 + gi, sent);
        CT(1);
        return g2 + s1;
    };
    int xi = lane, gi = lane;   // ring slots of this block's x, of the next block's S1 / G2
    const int ringb = a.rx / 64;
    f64x2_t dr = a.drd[lane];
    double uo = FORM == 2 ? a.rec[2 * (size_t)lane + 1] : 0.0;
    double accB = 0.0, accA = take_G(gi, sent, sent) + accB;
    CT(2);
#ifdef CHAIN_TIMING
    ct[0] = ct[1] = ct[2] = ct[3] = 0;
    const long long c_start = clock64();
    const unsigned long long r_start = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) printf("[t] chain loop starts %llu\n", r_start % 100000000ull);
#endif
    __builtin_amdgcn_s_setprio(3);   // the chain wave IS the critical path: its instructions go first on its SIMD
    for (int K = 0; K < a.nb; ++K) {
        if (lane == 0) __hip_atomic_store(s_k, K, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const double d = dr[0], rd = dr[1], ku = FORM == 2 ? (1 - a.w) * uo : 0.0;
        const int pn = (K + 1 < a.nb ? K + 1 : K) * 64 + lane;
        dr = a.drd[pn];                                             // the next block's records travel behind the steps
        if (FORM == 2) uo = a.rec[2 * (size_t)pn + 1];
        // an early look at the coming block's S1 and G2 (the helpers run ahead: normally there): their LDS round trip hides behind the steps
        const int gn = gi + 64 >= a.rg ? gi + 64 - a.rg : gi + 64;
        const double g2e = lds_get(G2r + gn), s1e = lds_get(S1 + gn);
        CT(2);
        chain_block<FORM>(accA, accB, d, rd, a.w, ku, cf, bp + ((size_t)K * 64 + CHAIN_PF) * 64);
        CT(0);
        // the ring slots of block K held block K - rx / 64: every helper that read it is done (rx = 64 (n1b + CHAIN_HA + 3)), and the
        // porter has exported it (it trails the chain by a block or two; checked, not assumed)
        while (K >= ringb && __hip_atomic_load(s_exp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= K - ringb) { if (chain_spin(a.sync, spins, t0)) break; }
        CHAIN_CFENCE();
        lds_put(X + xi, chain_update<FORM>(accA, d, rd, a.w, ku));
        CHAIN_CFENCE();
        if (lane == 0) __hip_atomic_store(s_done, K + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (behind the values: LDS operations of a wave complete in order)
        xi += 64; if (xi >= a.rx) xi -= a.rx;
        gi += 64; if (gi >= a.rg) gi -= a.rg;
        CT(2);
        if (K + 1 < a.nb) { accA = take_G(gi, g2e, s1e) + accB; accB = 0.0; }
    }
#ifdef CHAIN_TIMING
    if (lane == 0) printf("[chain] %d blocks: per block %.0f cycles in the 64 steps, %.0f waiting for tier 1, %.0f for tier 2, %.0f other; %.3f ticks of clock64 per ns\n", a.nb, (double)ct[0] / a.nb, (double)ct[1] / a.nb, (double)ct[3] / a.nb, (double)ct[2] / a.nb,
                          (double)(clock64() - c_start) / (10.0 * (double)(__builtin_amdgcn_s_memrealtime() - r_start)));
#endif
}

// ---- exporter: the finished blocks from the ring to W and to u, then the count of exported blocks (sync[4]: what tier 2 waits for).
// The count follows the stores at a distance: a block's stores are acknowledged a microsecond or two after they are issued -- waiting
// for each block's before the next one is touched made the exporter slower than the chain (and with it tier 2's gates, and the chain).
// While blocks keep coming the count trails by two blocks (stores of one wave complete in order: "all but the last four" are done);
// the moment the exporter has nothing to do it drains and catches up.
__device__ __forceinline__ void chain_export(const ChainArgs& a, double* X, int* s_done, int* s_exp, int lane)
{
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    typedef __attribute__((address_space(1))) unsigned gu32;
    int xi = lane, published = 0;
    unsigned spins = 0;
    unsigned long long t0 = 0;
    auto publish = [&](int n) {
        CHAIN_CFENCE();
        if (a.has_t2 && n > published) { published = n; if (lane == 0) __hip_atomic_store((gu32*)(a.sync + 4), (unsigned)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    };
    for (int ex = 0; ex < a.nb; ++ex) {
        while (__hip_atomic_load(s_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= ex) {
            if (published < ex) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); publish(ex); }   // idle: everything issued so far has arrived
            if (chain_spin(a.sync, spins, t0)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        CHAIN_CFENCE();
        const double x = __hip_atomic_load(X + xi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        CHAIN_CFENCE();
        const int p = ex * 64 + lane;
        const int row = a.tr[2 * (size_t)p + 1];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the row index above: a load; from here on this wave has stores in flight only)
        if (a.has_t2) __hip_atomic_store((gu64*)(a.W + p), (unsigned long long)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (write-through)
        else a.W[p] = x;
        *(row >= 0 ? a.u + row : a.W + a.npad + 64 + lane) = x;   // (padding rows write to scratch behind W: every block issues exactly two stores per lane)
        xi += 64; if (xi >= a.rx) xi -= a.rx;
        CHAIN_CFENCE();
        if (lane == 0) __hip_atomic_store(s_exp, ex + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (behind this wave's LDS read of the block)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    publish(a.nb);
}

// ---- importer: tier 2's sums (G2 in memory: values that are their own flags) into their ring in LDS, in block order, at most
// CHAIN_HA + 1 blocks ahead of the chain; four blocks are asked for at a time (a poll is a round trip to memory: one block per poll
// is slower than the chain).  Also pulls the band planes of the blocks up to four ahead of the chain into the L2, one dword per
// 128-byte line: the chain wave's loads then cost an L2 hit.  Eight loads per block, paced by the chain itself -- a wave that streams
// touches as fast as it can (tried: 32 in flight, tier 1's entries too) slows the chain wave's own loads down by more than it saves.
__device__ __forceinline__ void chain_import(const ChainArgs& a, double* G2r, int* s_k, int lane)
{
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    constexpr int NB = 4;
    int gi = lane, im = 0, tb = 0, touched = 0;
    unsigned spins = 0;
    unsigned long long t0 = 0;
    const char* bb = reinterpret_cast<const char*>(a.band);
    for (;;) {
        const int k = __hip_atomic_load(s_k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (im >= a.nb && k >= a.nb - 1) break;
        const int lim = k + CHAIN_HA + 1;
        bool moved = false;
        if (a.touch_lead <= 0 && tb < a.nb && tb <= k + 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) touched += *reinterpret_cast<const volatile int*>(bb + (size_t)tb * 65536 + ((size_t)i * 64 + lane) * 128);
            ++tb;
            moved = true;
        }
        unsigned long long g2[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int b = im + j, p = b * 64 + lane;
            g2[j] = FLOW_SENT;
            if (b < a.nb && b <= lim) g2[j] = a.has_t2 ? __hip_atomic_load((gu64*)(a.G2 + p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned long long)__double_as_longlong(a.rec[2 * (size_t)p]);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (__builtin_amdgcn_ballot_w64(g2[j] == FLOW_SENT)) break;   // (in block order: the first one that is not there ends the turn)
            CHAIN_CFENCE();
            __hip_atomic_store(G2r + gi, __longlong_as_double((long long)g2[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            gi += 64; if (gi >= a.rg) gi -= a.rg;
            ++im;
            moved = true;
        }
        if (!moved) { if (chain_spin(a.sync, spins, t0)) break; __builtin_amdgcn_s_sleep(4); }
    }
    if (touched == 0x7fffffff) a.sync[3] = 1u;   // (keeps the touches)
}

// ---- toucher: one workgroup on the chain's XCD (a tier-2 workgroup that would have stepped aside).  A block's band planes are 64 KB =
// 512 lines of 128 bytes: one dword per thread.  It follows the exported count (sync[4], a block or two behind the chain) at a distance of
// touch_lead blocks and sleeps in between; it ends with the chain (exported count = nb), on the error word, or after two seconds.
__device__ __forceinline__ void chain_touch(const ChainArgs& a, int tid)
{
    typedef __attribute__((address_space(1))) unsigned gu32;
    const char* bb = reinterpret_cast<const char*>(a.band);
    int tb = 0, touched = 0;
    unsigned spins = 0;
    unsigned long long t0 = 0;
    for (;;) {
        const int k = (int)__hip_atomic_load((gu32*)(a.sync + 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k >= a.nb || tb >= a.nb) break;
        if (__hip_atomic_load((gu32*)(a.sync + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        const int lim = min(a.nb, k + a.touch_lead);
        if (tb < lim) {
            for (; tb < lim; ++tb) {
                for (int l = tid; l < 512; l += CHAIN_NT) touched += *reinterpret_cast<const volatile int*>(bb + (size_t)tb * 65536 + (size_t)l * 128);
                if (a.touch_t1) {   // tier 1's values and columns of the block (the helper waves sit on the chain's CU: the same L2)
                    const ChainBlk B = a.blk[tb];
                    const char* v1 = reinterpret_cast<const char*>(a.t1v + (size_t)B.t1_off * 64);
                    const char* c1 = reinterpret_cast<const char*>(a.t1c + (size_t)B.t1_off * 64);
                    for (int l = tid; l < B.t1_n * 4; l += CHAIN_NT) touched += *reinterpret_cast<const volatile int*>(v1 + (size_t)l * 128);
                    for (int l = tid; l < B.t1_n; l += CHAIN_NT) touched += *reinterpret_cast<const volatile int*>(c1 + (size_t)l * 128);
                }
            }
            spins = 0;
        } else {
            if (spins == 0) t0 = __builtin_amdgcn_s_memrealtime();
            if ((++spins & 1023u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) break;   // (100 MHz: two seconds without a new block -- the chain's own time-out reports it)
            __builtin_amdgcn_s_sleep(8);
        }
    }
    if (touched == 0x7fffffff) a.sync[3] = 1u;   // (keeps the touches)
}

// ---- tier 1: helper waves of the chain's workgroup; blocks drawn in order from a counter in LDS; a block's sums go to the ring S1
__device__ __forceinline__ void chain_tier1(const ChainArgs& a, double* X, double* S1, int* s_k, int* s_done, int* s_ticket, int lane)
{
    unsigned spins = 0;
    unsigned long long t0 = 0;
    int have = 0;   // blocks in the ring, as last seen
#ifdef CHAIN_TIMING
    long long ct[3] = {0, 0, 0}, cl = clock64(); int nblk = 0;
#endif
    for (;;) {
        int K = 0;
        if (lane == 0) K = __hip_atomic_fetch_add(s_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        K = __builtin_amdgcn_readfirstlane(K);
        if (K >= a.nb) break;
        while (K > __hip_atomic_load(s_k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + CHAIN_HA) {
            if (chain_spin(a.sync, spins, t0)) break;
            __builtin_amdgcn_s_sleep(2);
        }
        CT(0);
        const flow_int_cp q = (flow_int_cp)(unsigned long long)(a.blk + K);
        const int off = q[0], n = q[1];
        double acc = 0.0;
        chain_tier_sum<false>(a.t1v + (size_t)off * 64 + lane, reinterpret_cast<const u32x4_t*>(a.t1c) + (size_t)(off >> 3) * 64 + lane, a.t1need + (off >> 3), n, acc,
                       [&](int need) {
                           while (need >= have) {
                               have = __hip_atomic_load(s_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                               if (need < have || chain_spin(a.sync, spins, t0)) break;
                           }
                           CHAIN_CFENCE();
                       },
                       [&](int ci) -> double { return __hip_atomic_load(X + ci, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); });
        CHAIN_CFENCE();
        __hip_atomic_store(S1 + (K * 64 + lane) % a.rg, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        CT(1);
#ifdef CHAIN_TIMING
        ++nblk;
#endif
    }
#ifdef CHAIN_TIMING
    if (lane == 0 && nblk) printf("[tier 1] %d blocks: per block %.0f cycles throttled / idle, %.0f in the sum (incl. waits for x)\n", nblk, (double)ct[0] / nblk, (double)ct[1] / nblk);
#endif
}

// ---- tier 2: every wave of the other workgroups; blocks drawn in order from a counter in memory
__device__ __forceinline__ void chain_tier2(const ChainArgs& a, int lane, int wg)
{
    typedef __attribute__((address_space(1))) unsigned      gu32;
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    unsigned spins = 0;
    unsigned long long t0 = 0;
    int have = 0;   // blocks exported, as last seen
#ifdef CHAIN_TIMING
    long long ct[3] = {0, 0, 0}, cl = clock64(), tw = 0; int nblk = 0;
#endif
    for (;;) {
        unsigned Ku = 0;
        if (lane == 0) Ku = __hip_atomic_fetch_add((gu32*)(a.sync + 2), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int K = (int)__builtin_amdgcn_readfirstlane(Ku);
        if (K >= a.nb) break;
        CT(0);
        const flow_int_cp q = (flow_int_cp)(unsigned long long)(a.blk + K);
        const int off = q[2], n = q[3];
        const int p = K * 64 + lane;
        double acc = a.rec[2 * (size_t)p];   // b_i - (what reads old values): pass (1)
        chain_tier_sum<true>(a.t2v + (size_t)off * 64 + lane, reinterpret_cast<const u32x4_t*>(a.t2c) + (size_t)(off >> 3) * 64 + lane, a.t2need + (off >> 3), n, acc,
                       [&](int need) {
#ifdef CHAIN_TIMING
                           const long long w0 = clock64();
#endif
                           while (need >= have) {
                               have = (int)__hip_atomic_load((gu32*)(a.sync + 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                               if (need < have || chain_spin(a.sync, spins, t0)) break;
                               __builtin_amdgcn_s_sleep(8);
                           }
                           CHAIN_CFENCE();
#ifdef CHAIN_TIMING
                           tw += clock64() - w0;
#endif
                       },
                       [&](int ci) -> double { return __longlong_as_double((long long)__hip_atomic_load((gu64*)(a.W + ci), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); });
        __hip_atomic_store((gu64*)(a.G2 + p), (unsigned long long)__double_as_longlong(acc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        CT(1);
#ifdef CHAIN_TIMING
        ++nblk;
#endif
    }
#ifdef CHAIN_TIMING
    if (lane == 0 && nblk && wg <= 2 && (threadIdx.x >> 6) == 0) printf("[tier 2, workgroup %d] %d blocks: per block %.0f cycles in the sum, of which %.0f at the gate\n", wg, nblk, (double)ct[1] / nblk, (double)tw / nblk);
#endif
}

template <int FORM>
__global__ __launch_bounds__(CHAIN_NT) void k_tri_chain(ChainArgs a)
{
    typedef __attribute__((address_space(1))) unsigned gu32;
    extern __shared__ __attribute__((aligned(16))) double chain_lds[];   // X[rx + 1], S1[rg], G2r[rg]
    __shared__ int s_role, s_k, s_ticket, s_exp, s_done;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef CHAIN_TIMING
    const unsigned long long t_enter = __builtin_amdgcn_s_memrealtime();
#define CTEND(what) do { if (lane == 0) printf("[t] %s wave %d: entered %llu, done %llu (10 ns ticks)\n", what, wave, t_enter % 100000000ull, __builtin_amdgcn_s_memrealtime() % 100000000ull); } while (0)
#else
#define CTEND(what)
#endif
    if (tid == 0) s_role = (int)__hip_atomic_fetch_add((gu32*)a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int role = __builtin_amdgcn_readfirstlane(s_role);
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xfu;
    if (role != 0) {
        // Tier 2 streams hundreds of KB per block: on the chain's XCD it would push the band planes (pulled in ahead of the chain wave) out
        // of the L2 they share.  The chain's workgroup -- running by the time anybody draws a later ticket -- says where it is; workgroups
        // that find themselves on the same XCD leave (an eighth of them).
        unsigned where = 0, spins = 0;
        unsigned long long t0 = 0;
        while (!(where = __hip_atomic_load((gu32*)(a.sync + 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { if (flow_give_up(a.sync, spins, t0)) break; __builtin_amdgcn_s_sleep(2); }
        // (At most a quarter of the launch steps aside -- sync[9] counts them: with eight XCDs that is everybody who shares the chain's, an
        // eighth; on a device or partition with ONE XCD, where every workgroup "shares" it, the other three quarters stay and sum tier 2
        // instead of leaving the chain to its time-out: ADVICE r5.)
        bool aside = false;
        if (where == xcc + 1u && gridDim.x > 9) {
            if (tid == 0) s_role = (int)__hip_atomic_fetch_add((gu32*)(a.sync + 9), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            aside = (unsigned)__builtin_amdgcn_readfirstlane(s_role) < (gridDim.x - 1u) / 4u;
            __syncthreads();
        }
        if (aside) {
            // ... all but the first of them (round 5): it pulls the band planes into this XCD's L2 well ahead of the chain.  Inside a cycle the
            // planes come from memory, not from the Infinity Cache (where repeated sweeps of one level find them): the importer wave's touches,
            // four blocks ahead and paced by its polls, arrive late then -- 64 steps in 6 500 cycles instead of 4 700 (profiles/r05_gs_chain.txt).
            if (a.touch_lead <= 0) return;
            if (tid == 0) s_role = (int)__hip_atomic_fetch_add((gu32*)(a.sync + 8), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (__builtin_amdgcn_readfirstlane(s_role) != 0) return;
            chain_touch(a, tid);
            return;
        }
        chain_tier2(a, lane, role);
        if (role <= 2 || role >= (int)gridDim.x - 2) CTEND("tier-2");
        return;
    }
    if (tid == 0) __hip_atomic_store((gu32*)(a.sync + 5), xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double* X = chain_lds;
    double* S1 = chain_lds + a.rx + 1;
    double* G2r = S1 + a.rg;
    const double sent = __longlong_as_double((long long)FLOW_SENT);
    for (int i = tid; i < a.rx + 1 + 2 * a.rg; i += CHAIN_NT) chain_lds[i] = i == a.rx ? 0.0 : sent;
    if (tid == 0) { s_k = 0; s_ticket = 0; s_exp = 0; s_done = 0; }
    __syncthreads();
    if (wave == 0) { chain_wave<FORM>(a, X, S1, G2r, &s_k, &s_done, &s_exp, lane); CTEND("chain"); }
    else if (wave == 1) { chain_export(a, X, &s_done, &s_exp, lane); CTEND("exporter"); }
    else if (wave == 2) { chain_import(a, G2r, &s_k, lane); CTEND("importer"); }
    else { chain_tier1(a, X, S1, &s_k, &s_done, &s_ticket, lane); CTEND("tier-1"); }
}

// The plain form: ONE wavefront, block after block -- tier 2 and tier 1 of the block (x from W in memory, written by this wave in
// earlier turns), then the block's 64 steps with the same chain_block.  No polling, no roles.
template <int FORM>
__global__ __launch_bounds__(64) void k_tri_chain_ref(ChainArgs a, int n1b)
{
    const int lane = threadIdx.x;
    double accB = 0.0;
    for (int K = 0; K < a.nb; ++K) {
        const ChainBlk B = a.blk[K];
        const int p = K * 64 + lane;
        double g2 = a.rec[2 * (size_t)p];
        for (int s = 0; s < B.t2_n; ++s) {
            const size_t e = ((size_t)B.t2_off + s) * 64 + lane, ec = (((size_t)(B.t2_off + s) >> 3) * 64 + lane) * 8 + (s & 7);   // (columns: eight per lane and group, side by side)
            g2 = __builtin_fma(-a.t2v[e], a.W[a.t2c[ec]], g2);
        }
        double s1 = 0.0;
        const int base = (K - 1 - n1b) * 64;   // first position of tier 1's window (may be negative: then the ring index is the position)
        for (int s = 0; s < B.t1_n; ++s) {
            const size_t e = ((size_t)B.t1_off + s) * 64 + lane, ec = (((size_t)(B.t1_off + s) >> 3) * 64 + lane) * 8 + (s & 7);
            const int r = a.t1c[ec];
            int qp;
            if (r >= a.rx) qp = a.npad;   // padding: the constant 0.0
            else { const int b0 = base > 0 ? base : 0; qp = b0 + ((r - b0 % a.rx) + a.rx) % a.rx; }
            s1 = __builtin_fma(-a.t1v[e], a.W[qp], s1);
        }
        double accA = (g2 + s1) + accB;
        accB = 0.0;
        const f64x2_t dr = a.drd[p];
        const double ku = FORM == 2 ? (1 - a.w) * a.rec[2 * (size_t)p + 1] : 0.0;
        const f64x2_t* bp = a.band + (size_t)K * 64 * 64 + lane;
#pragma unroll 4
        for (int c = 0; c < 64; ++c) {
            const f64x2_t m = bp[(size_t)c * 64];
            const double x = chain_update<FORM>(accA, dr[0], dr[1], a.w, ku);
            const double sx = __shfl(x, c, 64);
            accA = __builtin_fma(-m[0], sx, accA);
            accB = __builtin_fma(-m[1], sx, accB);
        }
        const double x = chain_update<FORM>(accA, dr[0], dr[1], a.w, ku);
        a.W[p] = x;
        const int row = a.tr[2 * (size_t)p + 1];
        if (row >= 0) a.u[row] = x;
        __threadfence();   // (the next blocks of this wave read W)
    }
}

}  // namespace fasp
