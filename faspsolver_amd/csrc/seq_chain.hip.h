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
// What is not in the band -- columns older than the previous block -- reaches the chain as ONE number per row (G), formed ahead of
// time by other wavefronts and read from LDS at the block boundary:
//     tier 1 (the n1b blocks in front of the band): helper waves of the chain's workgroup, x through a ring in LDS that the chain
//            wave fills block by block; a helper may run CHAIN_HA blocks ahead; a value is its own flag (signalling-NaN sentinel);
//     tier 2 (everything older): the other workgroups of the launch, x through memory (W, written by the exporter wave of the
//            chain's workgroup), results through memory (G2).  Tier 2 has n1b blocks of slack, tier 1 one block.
// Roles are dealt by tickets (the workgroup that draws ticket 0 is the chain's: it is running by definition; blocks of both tiers
// are drawn in order, so every block the chain waits for is in the hands of a running wave): no residency assumption.
// k_tri_chain_ref is the plain form of the same arithmetic -- ONE wavefront, block after block, no polling -- the fallback after a
// reported time-out and the A/B partner of the bit-identity test (tests/test_gpu_parity.py).
#pragma once
#include "seq_sched.h"

namespace fasp {

struct ChainArgs {
    const f64x2_t*        band;   // (nb * 64 + CHAIN_PF) steps x 64 lanes x (TA, TB)
    const f64x2_t*        drd;    // npad x (a_ii, 1 / a_ii)
    const ChainBlk*       blk;
    const double*         t1v; const unsigned short* t1c;
    const double*         t2v; const unsigned short* t2c;
    const double*         rec;    // 2 doubles per position, written by pass (1): b - rest, old u_i
    const int*            tr;     // 2 ints per position: flags, row index (-1: padding)
    double*               W;      // the new values by position (sentinel until written); W[npad] = 0.0 for the padding entries of tier 2
    double*               G2;     // tier 2's sums by position (sentinel until written)
    double*               u;
    unsigned*             sync;   // [0] role ticket, [1] error word, [2] tier-2 block ticket
    int                   nb, npad, rx, rg;
    int                   form;   // as tri_update
    double                w;
};

constexpr int CHAIN_NT = 512;   // 8 waves (256 registers each: the chain wave keeps CHAIN_PF coefficient pairs in flight): chain, exporter, 6 tier-1 helpers (chain workgroup) / 8 tier-2 workers (the others)
constexpr int CHAIN_U1 = 8, CHAIN_U2 = 8;   // steps per group of tier 1 / tier 2 (the schedules pad a block's steps to multiples of 8)

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


// One lane's share of a tier: n steps (a multiple of U) of (value, column), acc <- fma(-value, x[column], acc) in step order.
// Three register sets: the steps of the next TWO groups travel while this one gathers (the streams come from HBM: a microsecond).
// A group's operands are read until none of them is the sentinel (wave-uniform retry; `get` = the gather, LDS or memory).
template <int U, int NAP, class Get>
__device__ __forceinline__ void chain_tier_sum(const double* v, const unsigned short* c, int n, double& acc, unsigned* sync, unsigned& spins, unsigned long long& t0, Get get)
{
    double v0[U], v1[U], v2[U];
    int c0[U], c1[U], c2[U];
    auto fetch = [&](double (&vv)[U], int (&cc)[U], int s) {
        if (s < n) {
#pragma unroll
            for (int i = 0; i < U; ++i) { vv[i] = __builtin_nontemporal_load(v + (size_t)(s + i) * 64); cc[i] = __builtin_nontemporal_load(c + (size_t)(s + i) * 64); }
        }
    };
    auto use = [&](const double (&vv)[U], const int (&cc)[U]) {
        unsigned long long x[U];
        for (;;) {
            bool miss = false;
#pragma unroll
            for (int i = 0; i < U; ++i) { x[i] = get(cc[i]); miss |= x[i] == FLOW_SENT; }
            if (!__builtin_amdgcn_ballot_w64(miss)) break;
            if (flow_give_up(sync, spins, t0)) break;
            if (NAP) __builtin_amdgcn_s_sleep(NAP);
        }
#pragma unroll
        for (int i = 0; i < U; ++i) acc = __builtin_fma(-vv[i], __longlong_as_double((long long)x[i]), acc);
    };
    fetch(v0, c0, 0); fetch(v1, c1, U);
    for (int s = 0; s < n; s += 3 * U) {
        fetch(v2, c2, s + 2 * U); use(v0, c0);
        if (s + U >= n) break;
        fetch(v0, c0, s + 3 * U); use(v1, c1);
        if (s + 2 * U >= n) break;
        fetch(v1, c1, s + 4 * U); use(v2, c2);
    }
}

__device__ __forceinline__ bool chain_spin(unsigned* sync, unsigned& spins, unsigned long long& t0) { return flow_give_up(sync, spins, t0); }

// ---- the chain wave
template <int FORM>
__device__ __forceinline__ void chain_wave(const ChainArgs& a, double* X, double* G, int* s_k, int* s_exp, int lane)
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
    auto take_G = [&](int gi) -> double {   // this lane's G of the coming block; the slot is handed back (sentinel) for the block CHAIN_HA + 2 later
        double g;
        for (;;) {
            g = lds_get(G + gi);
            if (!__builtin_amdgcn_ballot_w64(!flow_ready(g))) break;
            if (chain_spin(a.sync, spins, t0)) break;
        }
        lds_put(G + gi, sent);
        return g;
    };
    int xi = lane, gi = lane, ri = lane + CHAIN_HA * 64;   // ring slots of this block's x, of the next block's G, of the block whose x slots are reset
    while (ri >= a.rx) ri -= a.rx;
    f64x2_t dr = a.drd[lane];
    double uo = FORM == 2 ? a.rec[2 * (size_t)lane + 1] : 0.0;
    double accB = 0.0, accA = take_G(gi) + accB;
    for (int K = 0; K < a.nb; ++K) {
        // the x slots of block K + CHAIN_HA: nobody looks for that block's values yet (helpers run at most CHAIN_HA blocks ahead and
        // read columns two blocks behind their rows), everybody who read the slots' previous owner is done (rx = 64 (n1b + CHAIN_HA + 3))
        // -- and the exporter has taken that owner's values (it trails the chain by a block or two; checked, not assumed)
        if (K + CHAIN_HA < a.nb && K > 0) {
            const int owner = K + CHAIN_HA - a.rx / 64;
            while (owner >= 0 && __hip_atomic_load(s_exp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= owner) { if (chain_spin(a.sync, spins, t0)) break; }
            lds_put(X + ri, sent);   // (blocks 0 .. CHAIN_HA are sentinel from the start)
        }
        if (lane == 0) __hip_atomic_store(s_k, K, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const double d = dr[0], rd = dr[1], ku = FORM == 2 ? (1 - a.w) * uo : 0.0;
        const int pn = (K + 1 < a.nb ? K + 1 : K) * 64 + lane;
        dr = a.drd[pn];                                             // the next block's records travel behind the steps
        if (FORM == 2) uo = a.rec[2 * (size_t)pn + 1];
        chain_block<FORM>(accA, accB, d, rd, a.w, ku, cf, bp + ((size_t)K * 64 + CHAIN_PF) * 64);
        lds_put(X + xi, chain_update<FORM>(accA, d, rd, a.w, ku));
        xi += 64; if (xi >= a.rx) xi -= a.rx;
        ri += 64; if (ri >= a.rx) ri -= a.rx;
        gi += 64; if (gi >= a.rg) gi -= a.rg;
        if (K + 1 < a.nb) { accA = take_G(gi) + accB; accB = 0.0; }
    }
}

// ---- exporter: the block's values from the ring to W (tier 2 polls it) and to u; touches the band a few blocks ahead into the L2
__device__ __forceinline__ void chain_export(const ChainArgs& a, double* X, int* s_exp, int lane)
{
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    constexpr int AHEAD = 4;
    int xi = lane, touched = 0;
    unsigned spins = 0;
    unsigned long long t0 = 0;
    const char* bb = reinterpret_cast<const char*>(a.band);
    const size_t band_bytes = ((size_t)a.nb * 64 + CHAIN_PF) * 1024;
    for (int K = 0; K < a.nb; ++K) {
        const size_t tb = (size_t)(K + AHEAD) * 65536;
        if (tb + 65536 <= band_bytes) {
#pragma unroll
            for (int i = 0; i < 8; ++i) touched += *reinterpret_cast<const volatile int*>(bb + tb + ((size_t)i * 64 + lane) * 128);
        }
        double x;
        for (;;) {
            x = __hip_atomic_load(X + xi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (!__builtin_amdgcn_ballot_w64(!flow_ready(x))) break;
            if (chain_spin(a.sync, spins, t0)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        const int p = K * 64 + lane;
        __hip_atomic_store((gu64*)(a.W + p), (unsigned long long)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int row = a.tr[2 * (size_t)p + 1];
        if (row >= 0) a.u[row] = x;
        xi += 64; if (xi >= a.rx) xi -= a.rx;
        if (lane == 0) __hip_atomic_store(s_exp, K + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (behind this wave's LDS reads of the block: LDS operations of a wave complete in order)
    }
    if (touched == 0x7fffffff) a.sync[3] = 1u;   // (keeps the touches)
}

// ---- tier 1: helper waves of the chain's workgroup; blocks drawn in order from a counter in LDS
__device__ __forceinline__ void chain_tier1(const ChainArgs& a, double* X, double* G, int* s_k, int* s_ticket, int lane)
{
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    unsigned spins = 0;
    unsigned long long t0 = 0;
    for (;;) {
        int K = 0;
        if (lane == 0) K = __hip_atomic_fetch_add(s_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        K = __builtin_amdgcn_readfirstlane(K);
        if (K >= a.nb) break;
        while (K > __hip_atomic_load(s_k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + CHAIN_HA) {
            if (chain_spin(a.sync, spins, t0)) break;
            __builtin_amdgcn_s_sleep(2);
        }
        const flow_int_cp q = (flow_int_cp)(unsigned long long)(a.blk + K);
        const int off = q[0], n = q[1];
        const double* v = a.t1v + (size_t)off * 64 + lane;
        const unsigned short* c = a.t1c + (size_t)off * 64 + lane;
        double acc = 0.0;
        chain_tier_sum<CHAIN_U1, 0>(v, c, n, acc, a.sync, spins, t0, [&](int ci) -> unsigned long long {
            return (unsigned long long)__double_as_longlong(__hip_atomic_load(X + ci, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); });
        // tier 2's sum of this row (memory; L1-bypassing polls), then G
        const int p = K * 64 + lane;
        unsigned long long g2;
        for (;;) {
            g2 = __hip_atomic_load((gu64*)(a.G2 + p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!__builtin_amdgcn_ballot_w64(g2 == FLOW_SENT)) break;
            if (chain_spin(a.sync, spins, t0)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        int gi = p % a.rg;
        __hip_atomic_store(G + gi, __longlong_as_double((long long)g2) + acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// ---- tier 2: every wave of the other workgroups; blocks drawn in order from a counter in memory
__device__ __forceinline__ void chain_tier2(const ChainArgs& a, int lane)
{
    typedef __attribute__((address_space(1))) unsigned      gu32;
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    unsigned spins = 0;
    unsigned long long t0 = 0;
    for (;;) {
        unsigned Ku = 0;
        if (lane == 0) Ku = __hip_atomic_fetch_add((gu32*)(a.sync + 2), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int K = (int)__builtin_amdgcn_readfirstlane(Ku);
        if (K >= a.nb) break;
        const flow_int_cp q = (flow_int_cp)(unsigned long long)(a.blk + K);
        const int off = q[2], n = q[3];
        const double* v = a.t2v + (size_t)off * 64 + lane;
        const unsigned short* c = a.t2c + (size_t)off * 64 + lane;
        const int p = K * 64 + lane;
        double acc = a.rec[2 * (size_t)p];   // b_i - (what reads old values): pass (1)
        chain_tier_sum<CHAIN_U2, 4>(v, c, n, acc, a.sync, spins, t0, [&](int ci) -> unsigned long long {
            return __hip_atomic_load((gu64*)(a.W + ci), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); });
        __hip_atomic_store((gu64*)(a.G2 + p), (unsigned long long)__double_as_longlong(acc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int FORM>
__global__ __launch_bounds__(CHAIN_NT) void k_tri_chain(ChainArgs a)
{
    typedef __attribute__((address_space(1))) unsigned gu32;
    extern __shared__ __attribute__((aligned(16))) double chain_lds[];   // X[rx + 1], G[rg]
    __shared__ int s_role, s_k, s_ticket, s_exp;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) s_role = (int)__hip_atomic_fetch_add((gu32*)a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int role = __builtin_amdgcn_readfirstlane(s_role);
    if (role != 0) { chain_tier2(a, lane); return; }
    double* X = chain_lds;
    double* G = chain_lds + a.rx + 1;
    const double sent = __longlong_as_double((long long)FLOW_SENT);
    for (int i = tid; i < a.rx + 1 + a.rg; i += CHAIN_NT) chain_lds[i] = i == a.rx ? 0.0 : sent;
    if (tid == 0) { s_k = 0; s_ticket = 0; s_exp = 0; }
    __syncthreads();
    if (wave == 0) chain_wave<FORM>(a, X, G, &s_k, &s_exp, lane);
    else if (wave == 1) chain_export(a, X, &s_exp, lane);
    else chain_tier1(a, X, G, &s_k, &s_ticket, lane);
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
            const size_t e = ((size_t)B.t2_off + s) * 64 + lane;
            g2 = __builtin_fma(-a.t2v[e], a.W[a.t2c[e]], g2);
        }
        double s1 = 0.0;
        const int base = (K - 1 - n1b) * 64;   // first position of tier 1's window (may be negative: then the ring index is the position)
        for (int s = 0; s < B.t1_n; ++s) {
            const size_t e = ((size_t)B.t1_off + s) * 64 + lane;
            const int r = a.t1c[e];
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
