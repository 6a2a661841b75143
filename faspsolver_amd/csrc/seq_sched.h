// seq_sched.h -- what the host side of the sequential sweeps' split form (seq_sched.cpp) hands to the device side (seq_split.hip.h)
#pragma once
#include <vector>

#include "fasp_internal.h"

namespace fasp {

struct FlowStrip {     // 32 bytes
    long long slot0;   // byte offset of the strip's slot storage
    int row0, nrows;   // positions [row0, row0 + nrows): LDS index = position - row0
    int chunk0, nchunk;
    int ghost0, nghost;   // gpos[ghost0 ..]: positions of the values read from earlier strips; LDS index = nrows + k; the 0.0 sits at nrows + nghost
};
constexpr int FLOW_LDS_ENT = 19 * 1024;        // doubles of LDS per strip: rows + ghosts + the constant 0.0
constexpr int TRI_PFMAX = 8;                   // slot rounds a chunk can store; the kernels come with room for 4 (schedules that never need more) or 8
constexpr int TRI_PF = 4;                      // slot rounds the lanes-per-row choice aims at
// The SPINE of a row (chain-bound levels -- classes of a few rows -- with eight or more lanes per row and eight rounds): its last TRI_SPINE lower
// entries in (class, sequence) order, i.e. the operands that become available last, sit in the row's LAST lane in the last
// TRI_SPINE rounds (the other lanes' slots of those rounds point at the constant).  The lanes' partial sums over everything else
// meet in that lane BEFORE the chain arrives; what is left between the arrival of the last operand and the row's value is two
// multiply-adds and the update -- not a six-step cross-lane sum.
constexpr int TRI_SPINE = 2;
// VIRTUAL ROWS: a work item (a row's share of a chunk) holds CAP = (TRI_PFMAX - spine) L + spine lower entries.  A row with more
// hands its OLDEST entries to virtual rows -- work items of their own, whose value is the plain sum of their products, kept in the
// strip's LDS next to the rows' values -- and reads each of them as one more operand with coefficient 1.  The oldest entries are the
// ones that are available long before the chain reaches the row: their sums are formed by other waves, ahead of time, out of
// prefetched slots like everything else (round 4 until then: a "tail" CSR read entry by entry by the row's own wave).  A virtual
// row sits in the class between its newest entry's and its row's (classes are doubled: rows even, virtual rows odd).
constexpr int FLOW_VIRTUAL = 0x40000000;       // flag in tr[2 p]: position p is a virtual row (tr[2 p + 1] = -1: no row of u)

// ---------------------------------------------------------------------------
// CHAIN form (round 5, seq_chain.hip.h): the triangular solve of a chain-bound sweep -- dependency classes of a few rows, rows of
// hundreds of lower entries: the deep levels -- as a BLOCKED substitution that keeps the dependency chain inside ONE wavefront.
// Positions = sweep order, padded to blocks of 64 (lane j of the chain wave <-> row 64 K + j).  The lower entries of a row of
// block K fall into three tiers by the block of their column:
//   band    blocks K (the 64 x 64 triangle) and K - 1 (the full square): dense coefficient planes, one (TA, TB) pair per lane and
//           step; the chain wave broadcasts x_c from lane c (v_readlane) and every lane does two multiply-adds -- no poll, no LDS
//           round trip on the critical path;
//   tier 1  blocks K - 1 - n1b .. K - 2: summed by helper waves of the chain's workgroup, x through a ring in LDS;
//   tier 2  everything older: summed by the other workgroups of the launch, x through memory (W), results through memory (G2).
// Both tiers are one lane per row, entries in column order, right-aligned in padded steps of 64 lanes (ELL per block).
// Row arithmetic (every tier a chain of fused multiply-adds in column order): G2 = T - sum(tier 2), S1 = -sum(tier 1) from +0.0,
// SB = -sum(block K - 1) from +0.0, t = ((G2 + S1) + SB) - sum(block K); then the update of tri_update.
// ---------------------------------------------------------------------------
#ifndef FASP_CHAIN_HA
#define FASP_CHAIN_HA 4   // (whole GS-default solves of P7(256), tools/build_variant.sh ha<N> -DFASP_CHAIN_HA=<N>: 2-5 blocks 277 ms, 6 / 8 / 12 / 20 blocks 282-283 ms: helpers far ahead of the chain only wait -- and poll the LDS the chain wave lives on)
#endif
constexpr int CHAIN_HA = FASP_CHAIN_HA;   // blocks the tier-1 helpers may run ahead of the chain
#ifndef FASP_CHAIN_PF
#define FASP_CHAIN_PF 32
#endif
constexpr int CHAIN_PF = FASP_CHAIN_PF;          // steps of band coefficients the chain wave keeps in flight (and zero steps behind the last block)
struct ChainBlk { int t1_off, t1_n, t2_off, t2_n; };   // offsets / counts in steps of 64 lanes
struct ChainHost {
    int nb = 0, npad = 0, n1b = 0, rx = 0, rg = 0;
    long long t1_steps = 0, t2_steps = 0, nband = 0, nt1 = 0, nt2 = 0;   // (entry counts: statistics)
    Buf<double>         band;   // (nb * 64 + CHAIN_PF) steps x 64 lanes x (TA, TB)
    Buf<double>         drd;    // npad x (a_ii, RN(1 / a_ii)); padding rows (1, 1)
    Buf<ChainBlk>       blk;
    Buf<double>         t1v, t2v;
    Buf<unsigned short> t1c, t2c;   // tier 1: ring index of the column's position (padding: rx); tier 2: the position (padding: npad)
    Buf<int>            t1need, t2need;   // per group of eight steps: the newest block its entries read (-1: padding only)
};

struct SplitHost {
    bool chain = false;   // the chain form below instead of strips / chunks / slots (ria, rja, rval, tr of the rest pass are shared)
    ChainHost C;
    int ns = 0, nrows = 0, nvirt = 0, nclasses = 0, L = 1, LR = 1, pfs = 4, kt = 0, nstrips = 0, nchunk = 0, maxent = 0, par = 1;   // ns: POSITIONS = rows of the sweep + virtual rows;   // kt: spine rounds (below); par: strips that share a dependency class at most (chain-bound levels; else nstrips)
    bool nolower = false, flow_ok = true, independent = false;   // independent (with nolower): the rows of the sweep do not couple at all (C rows / F rows of a 7-point level 0)
    long long nghost = 0, slot_bytes = 0, nrest = 0;
    std::vector<FlowStrip> strips;
    std::vector<int>       cptr;       // (doubled) dependency class -> first entry of lchunks
    Buf<int>               chunks;     // 4 ints per chunk: first local row | rows << 16 | rounds << 24, slot offset / 16, LDS index of the operand expected last, 0
    Buf<int>               cstrip, lchunks, gpos, ria, rja, tr;
    Buf<unsigned char>     slots;
    Buf<double>            rval, dr;
};
// Returns FASP_SUCCESS, 1 when a row of the sweep reads more earlier rows than a strip's LDS holds (no split form: the caller
// falls back to whole-row level scheduling), or a negative error code.
// team > 0: OpenMP team of this call (several schedules are built side by side, one host thread each: smoothers.hip.h)
// spine: 1 where it pays (above), 0 never, 2 wherever a row has two lanes or more (tests)
// chain: 0 never the chain form, 1 where the sweep is chain-bound (few rows per dependency class) and the form applies, 2 wherever it applies;
// chain_n1: blocks of tier 1 (0: two, the value whole solves measured best with -- seq_sched.cpp); strip_kb <= 0: chosen by level shape
int build_split_host(const HostCSR& A, const int* seq, int ns, int strip_kb, int seq_lanes, bool timing, SplitHost& H, int team = 0, int spine = 1,
                     int chain = 0, int chain_n1 = 0);

}  // namespace fasp
