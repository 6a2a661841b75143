// solver.hip -- device-resident AMG hierarchy, multigrid cycle, coarse-level safe CG,
// preconditioned CG driver and the C-ABI of libfasp_hip.so.
//
// Boundary (include/fasp_hip.h): fasp_solver_dcsr_krylov_amg() keeps the reference's
// signature (base/src/SolCSR.c:476).  Inside it the host builds the hierarchy
// (host_setup.cpp), uploads it once, and the whole Krylov loop runs on the GPU; only
// reduction scalars cross back, through a pinned buffer.
//
// Reference control flow restated here (paths relative to the reference tree):
//   fasp_solver_dcsr_pcg     base/src/KryPcg.c:96-362   (incl. stagnation / false-convergence)
//   fasp_precond_amg         base/src/PreCSR.c:416-435  (tol NOT forwarded: coarse tol 1e-10)
//   fasp_solver_mgcycle      base/src/PreMGCycle.c:48-274
//   fasp_coarse_itsolver     base/src/PreMGUtil.inl:37-58
//   fasp_solver_dcsr_spcg    base/src/KrySPcg.c:60-367  (pc == NULL)
//
// There is NO CPU fallback: without a usable gfx950 device every entry point that
// computes returns ERROR_MISC after printing why.
#include <hip/hip_runtime.h>

#include <omp.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <vector>

#include "fasp_comm.h"
#include "fasp_internal.h"
#include "kernels.hip.h"
#include "small_solvers.hip.h"

namespace fasp {

#define HIPCK(expr)                                                                        \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            std::fprintf(stderr, "### ERROR: fasp_hip: %s failed: %s [%s:%d]\n", #expr,    \
                         hipGetErrorString(e_), __FILE__, __LINE__);                       \
            return ERROR_MISC;                                                             \
        }                                                                                  \
    } while (0)

// ---------------------------------------------------------------------------
// device context (one per process: one process per GPU)
// ---------------------------------------------------------------------------
struct Ctx {
    bool        ready  = false;
    int         device = -1;
    hipStream_t stream = nullptr;
    double*     d_partials = nullptr;  // 8 quantities x MAXGRID
    double*     d_partials2 = nullptr; // second set (consumer kernels that read the first)
    double*     h_part     = nullptr;  // pinned mirror of d_partials2
    double*     d_red      = nullptr;  // reduced scalars (device)
    double*     h_red      = nullptr;  // pinned host mirror
    int         num_cu     = 256;
};
static Ctx g_ctx;
static int g_requested_device = -1;

constexpr int RED_SLOTS = 16;

static int ctx_init()
{
    if (g_ctx.ready) return FASP_SUCCESS;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        std::fprintf(stderr, "### ERROR: fasp_hip: no HIP device available (%s). This library has "
                             "no CPU fallback.\n", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return ERROR_MISC;
    }
    int dev = g_requested_device >= 0 ? g_requested_device : 0;
    if (const char* lr = std::getenv("FASP_HIP_DEVICE")) dev = std::atoi(lr);
    if (dev >= ndev) dev = dev % ndev;
    HIPCK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    HIPCK(hipGetDeviceProperties(&prop, dev));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        std::fprintf(stderr, "### WARNING: fasp_hip: device %d is %s; kernels are built for gfx950\n",
                     dev, prop.gcnArchName);
    g_ctx.num_cu = prop.multiProcessorCount;
    g_ctx.device = dev;
    HIPCK(hipStreamCreateWithFlags(&g_ctx.stream, hipStreamNonBlocking));
    HIPCK(hipMalloc(&g_ctx.d_partials, sizeof(double) * 8 * MAXGRID));
    HIPCK(hipMalloc(&g_ctx.d_partials2, sizeof(double) * (8 * MAXGRID + 8)));
    HIPCK(hipHostMalloc(&g_ctx.h_part, sizeof(double) * (8 * MAXGRID + 8), hipHostMallocDefault));
    HIPCK(hipMalloc(&g_ctx.d_red, sizeof(double) * RED_SLOTS));
    HIPCK(hipHostMalloc(&g_ctx.h_red, sizeof(double) * RED_SLOTS, hipHostMallocDefault));
    g_ctx.ready = true;
    return FASP_SUCCESS;
}

static inline int vec_grid(int n)
{
    long long g = ((long long)n + BLOCK - 1) / BLOCK;
    if (g > MAXGRID) g = MAXGRID;
    if (g < 1) g = 1;
    return (int)g;
}

// ---------------------------------------------------------------------------
// device matrices
// ---------------------------------------------------------------------------
struct DevCSR {
    int     row = 0, col = 0, nnz = 0;
    int*    ia  = nullptr;
    int*    ja  = nullptr;
    double* val = nullptr;
    int*    dpos = nullptr;  // storage index of the last diagonal entry per row (-1: none); A matrices only
    bool    dup_diag = false;  // some row stores its diagonal more than once
    bool    sorted = false;    // device copy has every row sorted by column (long-row operators)
    int     lanes = 8;       // vector kernel: lanes cooperating on one row
    int     kind = 0;        // 0 vector, 1 block-level stream, 2 wave-level stream
    int     tile_rows = 256; // block stream kernel: rows per block tile
    int     wrows = 64, wcap = 512;  // wave stream kernel: rows per wave tile, LDS products per wave
    // dictionary coding (k_csr_dict8): present when the matrix has <= 256 distinct (offset, value) pairs
    unsigned char* code = nullptr;
    int*    rowbase = nullptr;  // nullptr: column offsets are relative to the row index
    int*    doff = nullptr;
    double* dval = nullptr;
    // row-pattern coding (k_csr_rowpat): present when the matrix has <= 65536 distinct rows
    unsigned short* pat = nullptr;
    int*    pstart = nullptr;
    int*    plen = nullptr;
    int*    poff = nullptr;
    double* pval = nullptr;
    int     npat = 0, npent = 0;
    unsigned short* ja16 = nullptr;  // JA as 16-bit values (long-row operators with <= 65536 columns)
    void    release()
    {
        if (ja16) (void)hipFree(ja16);
        ja16 = nullptr;
        if (ia) (void)hipFree(ia);
        if (ja) (void)hipFree(ja);
        if (val) (void)hipFree(val);
        if (dpos) (void)hipFree(dpos);
        if (code) (void)hipFree(code);
        if (rowbase) (void)hipFree(rowbase);
        if (doff) (void)hipFree(doff);
        if (dval) (void)hipFree(dval);
        if (pat) (void)hipFree(pat);
        if (pstart) (void)hipFree(pstart);
        if (plen) (void)hipFree(plen);
        if (poff) (void)hipFree(poff);
        if (pval) (void)hipFree(pval);
        pat = nullptr; pstart = plen = poff = nullptr; pval = nullptr;
        ia = ja = dpos = rowbase = doff = nullptr; val = dval = nullptr; code = nullptr;
    }
};

// Kernel family per matrix, from its mean row length (measured on MI355X, P7(256)
// hierarchy, profiles/r01_kernel_sweep.md):
//   <= 48 nnz/row : wave-level stream kernel, 64 rows / 512 products per wave
//   longer rows   : sub-wavefront-per-row vector kernel with ~avg/4 lanes per row
static void pick_kernel(DevCSR& M)
{
    const double avg = M.row > 0 ? (double)M.nnz / M.row : 1.0;
    static const double stream_max = std::getenv("FASP_HIP_STREAM_MAX") ? std::atof(std::getenv("FASP_HIP_STREAM_MAX")) : 48.0;
    M.kind = avg <= stream_max ? 2 : 0;  // (kind 3, one workgroup per row, measured slower than L = 64: profiles/)
    M.lanes = avg < 128.0 ? 16 : avg < 300.0 ? 32 : 64;
    if (avg < 48.0) M.lanes = avg < 3.0 ? 2 : avg < 6.0 ? 4 : avg < 24.0 ? 8 : 16;  // short rows on the sub-wavefront kernel
    // Small transfer operators (fewer 256-row tiles than the chip has block slots): the stream kernel's
    // long per-tile chain is pure latency there; the sub-wavefront kernel spreads the rows over 8-32x
    // more blocks (level-3 restriction of P7(256): 25 -> 10 us).  Square operators keep the stream
    // kernel (its row sums follow the reference's order).
    static const int small_rows = std::getenv("FASP_HIP_SMALL_XFER_ROWS") ? std::atoi(std::getenv("FASP_HIP_SMALL_XFER_ROWS")) : 256 * 1024;
    if (M.kind == 2 && M.row != M.col && M.row < small_rows) M.kind = 0;
    int R = 256;
    while (R < STREAM_MAXR && R * 2 * avg <= 3072.0) R <<= 1;
    M.tile_rows = R;
    M.wrows = 64;
    M.wcap  = 512;
}

// Lossless dictionary coding of a matrix with <= 256 distinct (column - base, value) pairs
// (kernels.hip.h, k_csr_dict8).  base = row index for square matrices, first stored column of
// the row otherwise.  Returns false (nothing allocated) when the matrix has more pairs.
// One-shot callers (fasp_solver_dcsr_krylov_amg / fasp_solver_amg: one solve per setup) skip the
// hashing / re-sorting of the device copies: at 256^3 it costs 3.7 s of host time and saves 0.04 s per solve.
// Resident handles (fasp_hip_amg_create + many fasp_hip_solve) keep it.
static bool g_oneshot_upload = false;
static bool compress_enabled()
{
    if (g_oneshot_upload) return false;
    static int en = -1;
    if (en < 0) { const char* e = std::getenv("FASP_HIP_COMPRESS"); en = (e && std::atoi(e) == 0) ? 0 : 1; }
    return en != 0;
}
struct PairKey { int off; unsigned long long bits; };
static inline unsigned pair_hash(int off, unsigned long long bits)
{
    unsigned long long h = bits * 0x9E3779B97F4A7C15ull + (unsigned long long)(unsigned)off * 0xC2B2AE3D27D4EB4Full;
    return (unsigned)(h >> 40);
}
static bool build_dict8(const HostCSR& M, std::vector<int>& doff, std::vector<double>& dval,
                        Buf<unsigned char>& code, Buf<int>& rowbase)
{
    const bool square = M.row == M.col;
    const int n = M.row;
    if (n <= 0 || M.nnz <= 0) return false;
    constexpr int SLOTS = 1024;
    struct Table {
        int      cnt = 0;
        int      slot_id[SLOTS];
        PairKey  keys[257];
        Table() { for (int& s : slot_id) s = -1; }
        // returns the id of the pair, inserting it; -1 when the table is full
        int find_or_add(int off, unsigned long long bits, bool add)
        {
            unsigned h = pair_hash(off, bits) & (SLOTS - 1);
            for (;;) {
                const int id = slot_id[h];
                if (id < 0) {
                    if (!add || cnt >= 257) return -1;
                    keys[cnt] = PairKey{off, bits};
                    slot_id[h] = cnt;
                    return cnt++;
                }
                if (keys[id].off == off && keys[id].bits == bits) return id;
                h = (h + 1) & (SLOTS - 1);
            }
        }
    };
    auto base_of = [&](int r) { return square ? r : (M.ia[r] < M.ia[r + 1] ? M.ja[M.ia[r]] : 0); };
    auto bits_of = [](double v) { unsigned long long b; std::memcpy(&b, &v, 8); return b; };
    // pass 1: distinct pairs (every thread scans its share, bails out beyond 256)
    const int nt = omp_get_max_threads();
    std::vector<Table> local((size_t)nt);
    bool fail = false;
#pragma omp parallel num_threads(nt)
    {
        Table& T = local[(size_t)omp_get_thread_num()];
#pragma omp for schedule(static)
        for (int r = 0; r < n; ++r) {
            if (fail || T.cnt > 256) continue;
            const int base = base_of(r);
            for (int k = M.ia[r]; k < M.ia[r + 1]; ++k)
                if (T.find_or_add(M.ja[k] - base, bits_of(M.val[k]), true) < 0 || T.cnt > 256) { fail = true; break; }
        }
    }
    if (fail) return false;
    std::vector<PairKey> all;
    for (const Table& T : local) all.insert(all.end(), T.keys, T.keys + T.cnt);
    std::sort(all.begin(), all.end(), [](const PairKey& x, const PairKey& y) {
        return x.off != y.off ? x.off < y.off : x.bits < y.bits; });
    all.erase(std::unique(all.begin(), all.end(), [](const PairKey& x, const PairKey& y) {
        return x.off == y.off && x.bits == y.bits; }), all.end());
    if (all.size() > 256) return false;
    doff.assign(256, 0); dval.assign(256, 0.0);
    Table G;
    for (size_t i = 0; i < all.size(); ++i) {
        G.find_or_add(all[i].off, all[i].bits, true);  // ids in sorted order: deterministic
        doff[i] = all[i].off;
        std::memcpy(&dval[i], &all[i].bits, 8);
    }
    // pass 2: codes
    code.alloc((size_t)M.nnz);
    if (!square) rowbase.alloc((size_t)n);
#pragma omp parallel
    {
        Table T = G;
#pragma omp for schedule(static)
        for (int r = 0; r < n; ++r) {
            const int base = base_of(r);
            if (!square) rowbase[r] = base;
            for (int k = M.ia[r]; k < M.ia[r + 1]; ++k)
                code[k] = (unsigned char)T.find_or_add(M.ja[k] - base, bits_of(M.val[k]), false);
        }
    }
    return true;
}

// Row-pattern coding (kernels.hip.h, k_csr_rowpat): every row is replaced by the id of its
// (column - base, value) list when the matrix has at most 65 536 distinct lists with at most
// 1 M entries in total.  Lossless; ids are numbered by first occurrence, so the coding is
// deterministic.  Returns false when the matrix does not qualify (or on a hash collision).
static bool build_rowpat(const HostCSR& M, Buf<unsigned short>& pat, std::vector<int>& pstart, std::vector<int>& plen,
                         std::vector<int>& poff, std::vector<double>& pval, Buf<int>& rowbase)
{
    const bool square = M.row == M.col;
    const int n = M.row;
    if (n <= 0 || M.nnz <= 0) return false;
    constexpr int MAXPAT = 65536, MAXENT = 1 << 20;
    auto base_of = [&](int r) { return square ? r : (M.ia[r] < M.ia[r + 1] ? M.ja[M.ia[r]] : 0); };
    auto bits_of = [](double v) { unsigned long long b; std::memcpy(&b, &v, 8); return b; };
    auto row_hash = [&](int r) {
        unsigned long long h = 0x9FB21C651E98DF25ull * (unsigned long long)(M.ia[r + 1] - M.ia[r] + 1);
        const int base = base_of(r);
        for (int k = M.ia[r]; k < M.ia[r + 1]; ++k) {
            h ^= (unsigned long long)(unsigned)(M.ja[k] - base) * 0x9E3779B97F4A7C15ull + bits_of(M.val[k]) * 0xC2B2AE3D27D4EB4Full;
            h = (h << 23 | h >> 41) * 0xD6E8FEB86659FD93ull;
        }
        return h ? h : 1ull;
    };
    auto same_row = [&](int r, int q) {  // identical (offset, value) lists
        const int len = M.ia[r + 1] - M.ia[r];
        if (len != M.ia[q + 1] - M.ia[q]) return false;
        const int br = base_of(r), bq = base_of(q);
        for (int j = 0; j < len; ++j) {
            if (M.ja[M.ia[r] + j] - br != M.ja[M.ia[q] + j] - bq) return false;
            if (bits_of(M.val[M.ia[r] + j]) != bits_of(M.val[M.ia[q] + j])) return false;
        }
        return true;
    };
    // pass 1: hash of every row; per-thread sets of (hash -> first row), bounded
    Buf<unsigned long long> rh((size_t)n);
    const int nt = omp_get_max_threads();
    constexpr int SLOTS = 1 << 18;  // open addressing, <= 25 % load
    struct Set {
        std::vector<unsigned long long> key;
        std::vector<int> first;
        int cnt = 0;
        Set() : key(SLOTS, 0ull), first(SLOTS, -1) {}
        bool add(unsigned long long h, int r)
        {
            unsigned s = (unsigned)(h >> 20) & (SLOTS - 1);
            for (;;) {
                if (key[s] == 0ull) { key[s] = h; first[s] = r; ++cnt; return true; }
                if (key[s] == h) { if (r < first[s]) first[s] = r; return true; }
                s = (s + 1) & (SLOTS - 1);
            }
        }
        int find(unsigned long long h) const
        {
            unsigned s = (unsigned)(h >> 20) & (SLOTS - 1);
            for (;;) {
                if (key[s] == 0ull) return -1;
                if (key[s] == h) return first[s];
                s = (s + 1) & (SLOTS - 1);
            }
        }
    };
    std::vector<Set*> local((size_t)nt, nullptr);
    bool fail = false;
#pragma omp parallel num_threads(nt)
    {
        Set* S = new Set();
        local[(size_t)omp_get_thread_num()] = S;
#pragma omp for schedule(static)
        for (int r = 0; r < n; ++r) {
            const unsigned long long h = row_hash(r);
            rh[r] = h;
            if (fail) continue;
            S->add(h, r);
            if (S->cnt > MAXPAT) fail = true;
        }
    }
    Set* Gs = nullptr;
    std::vector<std::pair<int, unsigned long long>> reps;  // (first row, hash)
    if (!fail) {
        Gs = new Set();
        for (Set* S : local)
            if (S)
                for (int s = 0; s < SLOTS && !fail; ++s)
                    if (S->key[s]) { Gs->add(S->key[s], S->first[s]); if (Gs->cnt > MAXPAT) fail = true; }
    }
    for (Set* S : local) delete S;
    if (fail) { delete Gs; return false; }
    for (int s = 0; s < SLOTS; ++s)
        if (Gs->key[s]) reps.emplace_back(Gs->first[s], Gs->key[s]);
    std::sort(reps.begin(), reps.end());
    long long tot = 0;
    for (auto& q : reps) tot += (M.ia[q.first + 1] - M.ia[q.first] + 7) / 8 * 8;
    // worth it only when rows really repeat: >= 8 rows per pattern and a table << the matrix
    if (tot > MAXENT || (long long)reps.size() * 8 > n || tot * 4 > M.nnz) { delete Gs; return false; }
    // pattern table; the set now maps hash -> pattern id
    pstart.assign(reps.size(), 0); plen.assign(reps.size(), 0);
    poff.clear(); pval.clear();
    for (size_t i = 0; i < reps.size(); ++i) {  // lists padded to multiples of 8 entries with (offset 0, value 0)
        const int r = reps[i].first, base = base_of(r);
        pstart[i] = (int)poff.size();
        plen[i] = M.ia[r + 1] - M.ia[r];
        for (int k = M.ia[r]; k < M.ia[r + 1]; ++k) { poff.push_back(M.ja[k] - base); pval.push_back(M.val[k]); }
        while (poff.size() % 8) { poff.push_back(0); pval.push_back(0.0); }
    }
    for (int s = 0; s < SLOTS; ++s) Gs->first[s] = -1;
    {
        Set& S = *Gs;
        for (size_t i = 0; i < reps.size(); ++i) {
            unsigned s = (unsigned)(reps[i].second >> 20) & (SLOTS - 1);
            while (S.key[s] != reps[i].second) s = (s + 1) & (SLOTS - 1);
            S.first[s] = (int)i;
        }
    }
    pat.alloc((size_t)n);
    if (!square) rowbase.alloc((size_t)n);
    bool collision = false;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < n; ++r) {
        const int id = Gs->find(rh[r]);
        if (id < 0 || !same_row(r, reps[(size_t)id].first)) { collision = true; continue; }
        pat[r] = (unsigned short)id;
        if (!square) rowbase[r] = base_of(r);
    }
    delete Gs;
    return !collision;
}

// 16-bit copy of the column indices (in the order of the device copy) for the operators the
// sub-wavefront kernel serves: their time is the (JA, val) stream, 12 -> 10 bytes per entry.
static int upload_ja16(DevCSR& D, const int* ja_dev_order)
{
    static const bool on = !(std::getenv("FASP_HIP_JA16") && std::atoi(std::getenv("FASP_HIP_JA16")) == 0);
    if (!on || D.kind != 0 || D.col > 65536 || D.nnz < 4096) return FASP_SUCCESS;
    Buf<unsigned short> j16((size_t)D.nnz);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < D.nnz; ++k) j16[k] = (unsigned short)ja_dev_order[k];
    HIPCK(hipMalloc(&D.ja16, sizeof(unsigned short) * (size_t)D.nnz));
    HIPCK(hipMemcpy(D.ja16, j16.data(), sizeof(unsigned short) * (size_t)D.nnz, hipMemcpyHostToDevice));
    return FASP_SUCCESS;
}

static int upload_csr(const HostCSR& H, DevCSR& D)
{
    D.row = H.row; D.col = H.col; D.nnz = H.nnz;
    HIPCK(hipMalloc(&D.ia, sizeof(int) * ((size_t)H.row + 1)));
    HIPCK(hipMalloc(&D.ja, sizeof(int) * std::max<size_t>(H.nnz, 1)));
    HIPCK(hipMalloc(&D.val, sizeof(double) * std::max<size_t>(H.nnz, 1)));
    HIPCK(hipMemcpyAsync(D.ia, H.ia.data(), sizeof(int) * ((size_t)H.row + 1), hipMemcpyHostToDevice, g_ctx.stream));
    pick_kernel(D);
    auto upload_plain = [&]() -> int {
        HIPCK(hipMemcpyAsync(D.ja, H.ja.data(), sizeof(int) * (size_t)H.nnz, hipMemcpyHostToDevice, g_ctx.stream));
        HIPCK(hipMemcpyAsync(D.val, H.val.data(), sizeof(double) * (size_t)H.nnz, hipMemcpyHostToDevice, g_ctx.stream));
        return 0;
    };
    // (long rows keep the sub-wavefront kernel: the coded kernels are one-lane-per-row designs)
    if (compress_enabled() && H.nnz >= 4096 && (double)H.nnz <= 48.0 * H.row) {
        Buf<unsigned short> pat; Buf<int> prb;
        std::vector<int> pstart, plen, poff; std::vector<double> pval;
        static const bool rowpat_on = !(std::getenv("FASP_HIP_ROWPAT") && std::atoi(std::getenv("FASP_HIP_ROWPAT")) == 0);
        if (rowpat_on && build_rowpat(H, pat, pstart, plen, poff, pval, prb)) {
            D.npat = (int)pstart.size(); D.npent = (int)poff.size();
            HIPCK(hipMalloc(&D.pat, sizeof(unsigned short) * (size_t)H.row));
            HIPCK(hipMalloc(&D.pstart, sizeof(int) * pstart.size()));
            HIPCK(hipMalloc(&D.plen, sizeof(int) * plen.size()));
            HIPCK(hipMemcpy(D.plen, plen.data(), sizeof(int) * plen.size(), hipMemcpyHostToDevice));
            HIPCK(hipMalloc(&D.poff, sizeof(int) * std::max<size_t>(poff.size(), 1)));
            HIPCK(hipMalloc(&D.pval, sizeof(double) * std::max<size_t>(pval.size(), 1)));
            HIPCK(hipMemcpy(D.pat, pat.data(), sizeof(unsigned short) * (size_t)H.row, hipMemcpyHostToDevice));
            HIPCK(hipMemcpy(D.pstart, pstart.data(), sizeof(int) * pstart.size(), hipMemcpyHostToDevice));
            HIPCK(hipMemcpy(D.poff, poff.data(), sizeof(int) * poff.size(), hipMemcpyHostToDevice));
            HIPCK(hipMemcpy(D.pval, pval.data(), sizeof(double) * pval.size(), hipMemcpyHostToDevice));
            if (prb.n) {
                HIPCK(hipMalloc(&D.rowbase, sizeof(int) * (size_t)H.row));
                HIPCK(hipMemcpy(D.rowbase, prb.data(), sizeof(int) * (size_t)H.row, hipMemcpyHostToDevice));
            }
            D.kind = 2;  // plain-CSR twin of a coded operator: the stream kernel (same row-sum order; used by the A/B tests)
            return upload_plain() < 0 ? ERROR_ALLOC_MEM : FASP_SUCCESS;
        }
        std::vector<int> doff; std::vector<double> dval;
        Buf<unsigned char> code; Buf<int> rowbase;
        if (build_dict8(H, doff, dval, code, rowbase)) {
            HIPCK(hipMalloc(&D.code, (size_t)H.nnz + 256));  // + slack: spans are fetched in 16-byte units
            HIPCK(hipMalloc(&D.doff, sizeof(int) * 256));
            HIPCK(hipMalloc(&D.dval, sizeof(double) * 256));
            HIPCK(hipMemcpy(D.code, code.data(), (size_t)H.nnz, hipMemcpyHostToDevice));
            HIPCK(hipMemcpy(D.doff, doff.data(), sizeof(int) * 256, hipMemcpyHostToDevice));
            HIPCK(hipMemcpy(D.dval, dval.data(), sizeof(double) * 256, hipMemcpyHostToDevice));
            if (rowbase.n) {
                HIPCK(hipMalloc(&D.rowbase, sizeof(int) * (size_t)H.row));
                HIPCK(hipMemcpy(D.rowbase, rowbase.data(), sizeof(int) * (size_t)H.row, hipMemcpyHostToDevice));
            }
            D.kind = 2;
            return upload_plain() < 0 ? ERROR_ALLOC_MEM : FASP_SUCCESS;
        }
    }
    // not coded: plain CSR, rows re-sorted by column where the gathers dominate
    static const bool sort_long = !(std::getenv("FASP_HIP_SORT_LONG_ROWS") && std::atoi(std::getenv("FASP_HIP_SORT_LONG_ROWS")) == 0);
    static const int  sort_stream = std::getenv("FASP_HIP_SORT_STREAM") ? std::atoi(std::getenv("FASP_HIP_SORT_STREAM")) : 0;
    const double avg_len = H.row > 0 ? (double)H.nnz / H.row : 0.0;
    const bool do_sort = sort_long && !g_oneshot_upload && H.nnz > 0 && (D.kind == 0 || (D.kind == 2 && sort_stream > 0 && avg_len >= sort_stream));
    if (do_sort) {
        // Operators whose time goes into the x gathers -- one L1 tag lookup per distinct cache line, up to 64
        // per wavefront load when a row's columns come in discovery order: the DEVICE copy keeps every row's
        // entries sorted by column, so neighbouring lanes gather neighbouring entries.  (The sub-wavefront
        // kernel sums lane-strided partials + a shuffle tree, i.e. it never followed the storage order.)
        Buf<int> sj((size_t)H.nnz);
        Buf<double> sv((size_t)H.nnz);
        std::vector<int> dp;
        const bool square = H.row == H.col;
        if (square) dp.assign((size_t)H.row, -1);
#pragma omp parallel
        {
            std::vector<std::pair<int, double>> tmp;
#pragma omp for schedule(dynamic, 64)
            for (int i = 0; i < H.row; ++i) {
                const int kb = H.ia[i], ke = H.ia[i + 1];
                tmp.resize((size_t)(ke - kb));
                for (int k = kb; k < ke; ++k) tmp[(size_t)(k - kb)] = {H.ja[k], H.val[k]};
                std::stable_sort(tmp.begin(), tmp.end(),
                                 [](const std::pair<int, double>& x, const std::pair<int, double>& y) { return x.first < y.first; });
                for (int k = kb; k < ke; ++k) {
                    sj[k] = tmp[(size_t)(k - kb)].first; sv[k] = tmp[(size_t)(k - kb)].second;
                    if (square && sj[k] == i) dp[(size_t)i] = k;  // last diagonal hit (stable sort keeps their order)
                }
            }
        }
        HIPCK(hipMemcpy(D.ja, sj.data(), sizeof(int) * (size_t)H.nnz, hipMemcpyHostToDevice));
        HIPCK(hipMemcpy(D.val, sv.data(), sizeof(double) * (size_t)H.nnz, hipMemcpyHostToDevice));
        if (square && D.kind == 2) {
            HIPCK(hipMalloc(&D.dpos, sizeof(int) * (size_t)std::max(H.row, 1)));
            HIPCK(hipMemcpy(D.dpos, dp.data(), sizeof(int) * (size_t)H.row, hipMemcpyHostToDevice));
        }
        D.sorted = true;
        return upload_ja16(D, sj.data());
    }
    if (upload_plain() < 0) return ERROR_ALLOC_MEM;
    return upload_ja16(D, H.ja.data());
}

// development knobs (fasp_hip_tune): -1 = automatic
struct Tuning { int maxgrid = -1, xcd = 16, nt = 1, kind = -1, lanes = -1, wrows = -1, wcap = -1, compress = 1, rpl = -1, lds_tab = 1, xcd_pat = 64, spcg_batch = 8, small_lds = 1, ja16 = 1; };
static Tuning g_tune;

// Blocks of one kernel instantiation that are co-resident on a CU (VGPR / LDS / wave
// limits), from the occupancy API, cached per instantiation.  The persistent grids
// below are sized to exactly this residency: a grid larger than what is resident
// serialises into two rounds (measured: +35 % time), a smaller one leaves CUs idle.
template <class K>
static int resident_blocks_per_cu(K kernel)
{
    static int cached = -1;
    if (cached < 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, BLOCK, 0) != hipSuccess || nb < 1) nb = 4;
        cached = std::min(nb, 8);
    }
    return cached;
}

template <class K>
static int launch_persistent(K kernel, int ntiles, CsrArgs& a)
{
    int cap = resident_blocks_per_cu(kernel) * g_ctx.num_cu;
    if (g_tune.maxgrid > 0) cap = g_tune.maxgrid;
    cap = std::min(cap, MAXGRID);
    int grid = std::min(cap, ntiles);
    grid = std::max(8, (grid + 7) / 8 * 8);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK), 0, g_ctx.stream, a);
    return grid;
}

// Launches the row kernel of family M.kind for operation OP; returns the grid size
// (= number of per-block partials written by OP_MXV_DOT).
template <int OP>
static int launch_csr(const DevCSR& M0, CsrArgs a)
{
    DevCSR M = M0;  // shallow copy: tuning overrides
    if (M.code && g_tune.compress) M.kind = 4;  // dictionary-coded copy present: one byte per entry
    if (M.pat && g_tune.compress) M.kind = 5;   // row-pattern-coded copy present: two bytes per row
    if (g_tune.kind >= 0 && !(g_tune.kind == 4 && !M.code) && !(g_tune.kind == 5 && !M.pat)) M.kind = g_tune.kind;
    if (g_tune.lanes > 0) M.lanes = g_tune.lanes;
    if (g_tune.wrows > 0) M.wrows = g_tune.wrows;
    if (g_tune.wcap > 0) M.wcap = g_tune.wcap;
    if (OP == OP_JACOBI && M.kind != 0 && M.kind < 4 && (M.dup_diag || !M.dpos)) M.kind = 0;  // needs the c != r test
    a.xcd_map = g_tune.xcd;
    a.nt = g_tune.nt;
    a.nrow = M.row; a.ia = M.ia; a.ja = M.ja; a.val = M.val; a.dpos = M.dpos;
    a.ja16 = g_tune.ja16 ? M.ja16 : nullptr;
    const int rpb = M.kind >= 4 ? BLOCK : M.kind == 2 ? 4 * M.wrows : M.kind == 1 ? M.tile_rows : M.kind == 3 ? 1 : BLOCK / M.lanes;
    a.ntiles = (M.row + rpb - 1) / rpb;
    a.tiles_per_xcd = (a.ntiles + 7) / 8;
    if (M.kind == 5) {
        a.pat = M.pat; a.pstart = M.pstart; a.poff = M.poff; a.pval = M.pval; a.rowbase = M.rowbase;
        a.npat = M.npat; a.npent = M.npent;
        const double avg = M.row > 0 ? (double)M.nnz / M.row : 1.0;
        const bool lds = M.npat <= 512 && M.npent <= 2048 && g_tune.lds_tab != 0;
        const int rpl = g_tune.rpl > 0 ? g_tune.rpl : 1;
        a.plen = M.plen; a.ncol = M.col;
        if (g_tune.xcd_pat != 0) a.xcd_map = g_tune.xcd_pat;
        a.ntiles = (M.row + BLOCK * rpl - 1) / (BLOCK * rpl);
        a.tiles_per_xcd = (a.ntiles + 7) / 8;
        (void)avg;
        if (lds) {
            if (rpl == 1) return launch_persistent(k_csr_rowpat<OP, true, 1>, a.ntiles, a);
            return launch_persistent(k_csr_rowpat<OP, true, 2>, a.ntiles, a);
        }
        if (rpl == 1) return launch_persistent(k_csr_rowpat<OP, false, 1>, a.ntiles, a);
        return launch_persistent(k_csr_rowpat<OP, false, 2>, a.ntiles, a);
    }
    if (M.kind == 4) {
        a.code = M.code; a.rowbase = M.rowbase; a.doff = M.doff; a.dval = M.dval;
        const double avg = M.row > 0 ? (double)M.nnz / M.row : 1.0;
        if (avg <= 8.5) return launch_persistent(k_csr_dict8<OP, 8>, a.ntiles, a);
        if (avg <= 20.0) return launch_persistent(k_csr_dict8<OP, 16>, a.ntiles, a);
        return launch_persistent(k_csr_dict8<OP, 24>, a.ntiles, a);
    }
    if (M.kind == 2) {
        if (M.wrows == 64 && M.wcap == 512) return launch_persistent(k_csr_wstream<OP, 64, 512>, a.ntiles, a);
        if (M.wrows == 64) return launch_persistent(k_csr_wstream<OP, 64, 1024>, a.ntiles, a);
        if (M.wrows == 32 && M.wcap == 512) return launch_persistent(k_csr_wstream<OP, 32, 512>, a.ntiles, a);
        return launch_persistent(k_csr_wstream<OP, 32, 1024>, a.ntiles, a);
    }
    if (M.kind == 3) return launch_persistent(k_csr_blockrow<OP>, M.row, a);
    if (M.kind == 1) {
        int cap = g_tune.maxgrid > 0 ? g_tune.maxgrid : 4 * g_ctx.num_cu;
        int grid = std::max(8, (std::min(std::min(cap, MAXGRID), a.ntiles) + 7) / 8 * 8);
        hipLaunchKernelGGL((k_csr_stream<OP>), dim3(grid), dim3(BLOCK), 0, g_ctx.stream, a, M.tile_rows);
        return grid;
    }
    switch (M.lanes) {
        case 2:  return launch_persistent(k_csr_rows<2, OP>, a.ntiles, a);
        case 4:  return launch_persistent(k_csr_rows<4, OP>, a.ntiles, a);
        case 8:  return launch_persistent(k_csr_rows<8, OP>, a.ntiles, a);
        case 16: return launch_persistent(k_csr_rows<16, OP>, a.ntiles, a);
        case 32: return launch_persistent(k_csr_rows<32, OP>, a.ntiles, a);
        default: return launch_persistent(k_csr_rows<64, OP>, a.ntiles, a);
    }
}

// y = A x
static void d_mxv(const DevCSR& A, const double* x, double* y)
{
    CsrArgs a{}; a.x = x; a.y = y;
    launch_csr<OP_MXV>(A, a);
}
// y = b - A x
static void d_resid(const DevCSR& A, const double* x, const double* b, double* y)
{
    CsrArgs a{}; a.x = x; a.y = y; a.b = b;
    launch_csr<OP_RESID>(A, a);
}
// y += alpha A x  (three rounding-distinct paths of BlaSpmvCSR.c:494)
static void d_aAxpy(double alpha, const DevCSR& A, const double* x, double* y)
{
    CsrArgs a{}; a.x = x; a.y = y; a.alpha = alpha;
    if (alpha == 1.0) launch_csr<OP_ADD>(A, a);
    else if (alpha == -1.0) launch_csr<OP_SUB>(A, a);
    else launch_csr<OP_AXPY>(A, a);
}

// --- reductions ----------------------------------------------------------------
// local partials -> d_red[slot .. slot+nq) -> (all-reduce over ranks) .  Host copy on demand.
static void d_finalize_to(int G, int nq, unsigned maxmask, double* out, bool dist)
{
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLOCK), 0, g_ctx.stream, g_ctx.d_partials, G, nq,
                       maxmask, out);
    if (dist && comm_size() > 1) comm_allreduce(out, nq, maxmask, g_ctx.stream);
}
static void d_finalize(int G, int nq, unsigned maxmask, int slot, bool dist)
{
    d_finalize_to(G, nq, maxmask, g_ctx.d_red + slot, dist);
}
static int fetch_red(int slot, int nq, double* out)
{
    HIPCK(hipMemcpyAsync(g_ctx.h_red + slot, g_ctx.d_red + slot, sizeof(double) * nq,
                         hipMemcpyDeviceToHost, g_ctx.stream));
    HIPCK(hipStreamSynchronize(g_ctx.stream));
    for (int q = 0; q < nq; ++q) out[q] = g_ctx.h_red[slot + q];
    return FASP_SUCCESS;
}
// (x,y) left on the device in reduction slot `slot` (no host round trip)
static int d_dot_to(int n, const double* x, const double* y, int slot, bool dist)
{
    const int G = vec_grid(n);
    hipLaunchKernelGGL(k_dot, dim3(G), dim3(BLOCK), 0, g_ctx.stream, n, x, y, g_ctx.d_partials);
    d_finalize(G, 1, 0u, slot, dist);
    return 0;
}
static int d_dot(int n, const double* x, const double* y, double* out, bool dist = false)
{
    const int G = vec_grid(n);
    hipLaunchKernelGGL(k_dot, dim3(G), dim3(BLOCK), 0, g_ctx.stream, n, x, y, g_ctx.d_partials);
    d_finalize(G, 1, 0u, 0, dist);
    return fetch_red(0, 1, out);
}
// out[0] = sum x^2, out[1] = max |x|
static int d_norms(int n, const double* x, double* out, bool dist = false)
{
    const int G = vec_grid(n);
    hipLaunchKernelGGL(k_norms, dim3(G), dim3(BLOCK), 0, g_ctx.stream, n, x, g_ctx.d_partials);
    d_finalize(G, 2, 0x2u, 0, dist);
    return fetch_red(0, 2, out);
}
static void d_axpy(int n, double a, const double* x, double* y)
{
    hipLaunchKernelGGL(k_axpy, dim3(vec_grid(n / 2 + 1)), dim3(BLOCK), 0, g_ctx.stream, n, a, x, y);
}
static void d_axpby(int n, double a, const double* x, double b, double* y)
{
    hipLaunchKernelGGL(k_axpby, dim3(vec_grid(n / 2 + 1)), dim3(BLOCK), 0, g_ctx.stream, n, a, x, b, y);
}

// ---------------------------------------------------------------------------
// resident hierarchy
// ---------------------------------------------------------------------------
struct DevLevel {
    DevCSR  A, P, R;
    double* diag = nullptr;  // last diagonal hit per row (Jacobi: ItrSmootherCSR.c:160)
    double* l1   = nullptr;  // sum_j |a_ij| (L1-diag: ItrSmootherCSR.c:1566)
    double *b = nullptr, *xa = nullptr, *xb = nullptr, *w = nullptr;
    double* x  = nullptr;    // current iterate: xa or xb
    double* xo = nullptr;    // the other buffer
    bool    x_zero = true;   // x is (conceptually) all zeros and not materialised
    bool    owns_b = true;
    // distribution (single GPU: nloc == nvec == rows, no halo)
    bool    replicated = true;   // whole level on every rank, computed redundantly
    int     nloc = 0;            // owned entries of this level's vectors
    int     nvec = 0;            // vector length incl. ghost entries [nloc, nvec)
    int     row0 = 0;            // global index of the first owned row
    int     nglobal = 0;
    std::vector<int> send_off, recv_off;  // nranks+1 each
    int*    d_send_idx = nullptr;
    double* d_sendbuf  = nullptr;
    bool    has_halo() const { return !replicated && nvec > nloc; }
    // level schedules of the sequential sweeps (built on first use): kind 0 ascending,
    // 1 descending, 2 ascending C rows, 3 ascending F rows, 4 descending from row n-2
    struct Sched { bool built = false; int* d_order = nullptr; std::vector<int> ptr; };
    Sched   sched[5];
    // polynomial smoother (built on first use): 1 / first diagonal hit, the coefficients k[1..5] of
    // ItrSmootherCSRpoly.c:101-109, work vectors r, rbar, v0, v1, vnew
    struct Poly { bool built = false; double* dinv = nullptr; double k[6] = {0, 0, 0, 0, 0, 0}; double* w[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; };
    Poly    poly;
    int*    d_mark = nullptr;  // C/F marker on the device (Jacobi-F smoother), built on first use
    double* w2 = nullptr;      // AMLI cycle: the coarse residual r1 of the level above, built on first use
    double* kw[4] = {nullptr, nullptr, nullptr, nullptr};  // K-cycle work vectors r, x1, v1, v2 of this level
};

struct EventPair { hipEvent_t a, b; };

}  // namespace fasp

using namespace fasp;

struct fasp_hip_amg {
    HostHierarchy         H;
    DistPlan              dist;        // row partition (nranks == 1: trivial)
    bool                  distributed = false;  // level 0 is row-partitioned over the ranks
    std::vector<DevLevel> L;
    AMG_param             param;  // copy of the user's parameters after setup
    // Krylov work vectors on level 0
    double *b = nullptr, *u = nullptr, *p = nullptr, *t = nullptr, *r = nullptr;
    // coarse-level SPCG work vectors
    double *cp = nullptr, *cr = nullptr, *ct = nullptr, *cbest = nullptr;
    // GMRES basis vectors (allocated on first use): level-0 set and coarse-level set
    std::vector<double*> gm[2];
    size_t               gm_len[2] = {0, 0};
    double*              gm_hh = nullptr;  // device Hessenberg column
    SpcgState*           spcg_state = nullptr;  // device-resident state of the batched coarse CG
    std::vector<double>  amli_coef;             // AMLI polynomial coefficients (amli_degree + 1), formed on first use
    std::vector<int>     level_cycle_type;      // AMG_data.cycle_type per level as the setup leaves it (K-cycle)
    bool                 use_fmg = false;       // the preconditioner is one full-multigrid cycle (precond_type == PREC_FMG)
    // instrumentation
    std::vector<EventPair> ev;
    int                    ev_used = 0;
    long long              coarse_iters = 0, vcycles = 0;
    double                 upload_seconds = 0.0;
};

namespace fasp {

static void free_level(DevLevel& D)
{
    D.A.release(); D.P.release(); D.R.release();
    if (D.diag) (void)hipFree(D.diag);
    if (D.l1) (void)hipFree(D.l1);
    if (D.b && D.owns_b) (void)hipFree(D.b);
    if (D.xa) (void)hipFree(D.xa);
    if (D.xb) (void)hipFree(D.xb);
    if (D.w) (void)hipFree(D.w);
    if (D.d_send_idx) (void)hipFree(D.d_send_idx);
    if (D.d_sendbuf) (void)hipFree(D.d_sendbuf);
    for (auto& sc : D.sched) if (sc.d_order) (void)hipFree(sc.d_order);
    if (D.poly.dinv) (void)hipFree(D.poly.dinv);
    for (double* q : D.poly.w) if (q) (void)hipFree(q);
    if (D.d_mark) (void)hipFree(D.d_mark);
    if (D.w2) (void)hipFree(D.w2);
    for (double* q : D.kw) if (q) (void)hipFree(q);
    D = DevLevel();
}

static int alloc_vec(double** p, size_t n)
{
    HIPCK(hipMalloc(p, sizeof(double) * std::max<size_t>(n, 1)));
    return FASP_SUCCESS;
}

// diag / l1 are derived on the host in the reference's order (one pass, setup time)
static int upload_diag(const HostCSR& A, DevLevel& D)
{
    const int n = A.row;
    std::vector<double> d(n, 0.0), s(n, 0.0);
    std::vector<int>    dp(n, -1);
    int                 ndup = 0;
#pragma omp parallel for schedule(static) reduction(+ : ndup)
    for (int i = 0; i < n; ++i) {
        double di = 0.0, si = 0.0;
        int    hits = 0;
        for (int k = A.ia[i]; k < A.ia[i + 1]; ++k) {
            if (A.ja[k] == i) { di = A.val[k]; dp[i] = k; ++hits; }
            si += (A.val[k] >= 0.0) ? A.val[k] : -A.val[k];
        }
        d[i] = di; s[i] = si;
        if (hits > 1) ++ndup;
    }
    D.A.dup_diag = ndup > 0;
    if (!D.A.sorted) {  // (storage indices of the host order: meaningless for a re-sorted device copy)
        HIPCK(hipMalloc(&D.A.dpos, sizeof(int) * std::max(n, 1)));
        HIPCK(hipMemcpy(D.A.dpos, dp.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    }
    if (alloc_vec(&D.diag, n) < 0 || alloc_vec(&D.l1, n) < 0) return ERROR_ALLOC_MEM;
    HIPCK(hipMemcpy(D.diag, d.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(D.l1, s.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    return FASP_SUCCESS;
}

// gather v[idx[i]] into a contiguous send buffer
__global__ __launch_bounds__(BLOCK) void k_pack(int n, const int* __restrict__ idx,
                                                 const double* __restrict__ v, double* __restrict__ out)
{
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) out[i] = v[idx[i]];
}

// Refresh the ghost entries [nloc, nvec) of a level-l vector from their owners: pack the
// entries the peers need, one grouped RCCL send/recv, receive straight into the ghost
// slots (ghosts are sorted by owner, so every peer's block is contiguous).
static int halo_exchange(DevLevel& D, double* v)
{
    if (!D.has_halo() && (D.send_off.empty() || D.send_off.back() == 0)) return FASP_SUCCESS;
    const int P = comm_size(), me = comm_rank();
    const int nsend = D.send_off.back();
    if (nsend > 0)
        hipLaunchKernelGGL(k_pack, dim3(vec_grid(nsend)), dim3(BLOCK), 0, g_ctx.stream, nsend, D.d_send_idx, v,
                           D.d_sendbuf);
    std::vector<CommXfer> sends, recvs;
    for (int q = 0; q < P; ++q) {
        if (q == me) continue;
        const int ns = D.send_off[q + 1] - D.send_off[q], nr = D.recv_off[q + 1] - D.recv_off[q];
        if (ns > 0) sends.push_back({q, D.d_sendbuf + D.send_off[q], (size_t)ns});
        if (nr > 0) recvs.push_back({q, v + D.nloc + D.recv_off[q], (size_t)nr});
    }
    return comm_exchange(sends.data(), (int)sends.size(), recvs.data(), (int)recvs.size(), g_ctx.stream);
}

static int upload_hierarchy(fasp_hip_amg* h)
{
    HostThreads team;  // matrix coding / re-sorting / partition loops
    const double t0 = wall_seconds();
    const int nl = (int)h->H.L.size();
    h->L.resize(nl);
    int min_rows = 200000;
    if (const char* e = std::getenv("FASP_HIP_DIST_MIN_ROWS")) min_rows = std::atoi(e);
    // sequential (Gauss-Seidel / SOR) sweeps couple all rows of a level: such hierarchies
    // are not row-partitioned, every rank keeps (and computes) all levels
    if (h->param.smoother != SMOOTHER_JACOBI && h->param.smoother != SMOOTHER_L1DIAG && h->param.smoother != SMOOTHER_POLY) min_rows = 2147483647;
    if (h->param.cycle_type == AMLI_CYCLE || h->param.cycle_type == NL_AMLI_CYCLE) min_rows = 2147483647;  // the recursive cycles run on whole levels
    {   // AMG_data.cycle_type of every level: the setup's cycle type for levels >= 1 (PreAMGSetupRS.c:325,
        // PreAMGSetupSA.c:495); the UA setup derives it from the operator complexity (PreAMGSetupUA.c:390-401)
        h->level_cycle_type.assign((size_t)nl, h->param.cycle_type);
        h->level_cycle_type[0] = 0;
        if (h->param.AMG_type == UA_AMG) {
            const double cplxmax = 3.0, xsi = 0.6, eta = xsi / ((1 - xsi) * (cplxmax - 1));
            int icum = 1;
            h->level_cycle_type[0] = 1;
            h->level_cycle_type[(size_t)nl - 1] = 0;
            for (int lvl = 1; lvl < nl - 1; ++lvl) {
                const double fracratio = (double)h->H.L[lvl].A.nnz / h->H.L[0].A.nnz;
                int ct = (int)(std::pow(xsi, (double)lvl) / (eta * fracratio * icum));
                ct = std::max(1, std::min(2, ct));
                h->level_cycle_type[(size_t)lvl] = ct;
                icum = icum * ct;
            }
        }
    }
    {
        const int st = build_dist_plan(h->H, comm_rank(), comm_size(), min_rows, h->dist);
        if (st < 0) return st;
    }
    h->distributed = !h->dist.L[0].replicated;
    for (int l = 0; l < nl; ++l) {
        const HostLevel& HL = h->H.L[l];
        const DistLevel& DL = h->dist.L[l];
        DevLevel& D = h->L[l];
        D.replicated = DL.replicated;
        D.nloc = DL.nloc; D.row0 = DL.row0; D.nglobal = DL.nglobal;
        D.nvec = DL.replicated ? DL.nglobal : DL.nloc + (int)DL.ghosts.size();
        const HostCSR& A = DL.replicated ? HL.A : DL.A;
        if (upload_csr(A, D.A) < 0) return ERROR_ALLOC_MEM;
        if (HL.has_coarse) {
            if (upload_csr(DL.replicated ? HL.P : DL.P, D.P) < 0) return ERROR_ALLOC_MEM;
            if (upload_csr(DL.replicated ? HL.R : DL.R, D.R) < 0) return ERROR_ALLOC_MEM;
        }
        HIPCK(hipStreamSynchronize(g_ctx.stream));
        if (upload_diag(A, D) < 0) return ERROR_ALLOC_MEM;
        const size_t n = D.nvec;
        if (l > 0) { if (alloc_vec(&D.b, n) < 0) return ERROR_ALLOC_MEM; }
        else D.owns_b = false;  // level-0 rhs aliases the Krylov residual (PreCSR.c:429 copy elided)
        if (alloc_vec(&D.xa, n) < 0 || alloc_vec(&D.xb, n) < 0 || alloc_vec(&D.w, n) < 0) return ERROR_ALLOC_MEM;
        D.x = D.xa; D.xo = D.xb; D.x_zero = true;
        if (!DL.replicated) {
            D.send_off = DL.send_off; D.recv_off = DL.recv_off;
            const size_t ns = DL.send_idx.size();
            HIPCK(hipMalloc(&D.d_send_idx, sizeof(int) * std::max<size_t>(ns, 1)));
            HIPCK(hipMalloc(&D.d_sendbuf, sizeof(double) * std::max<size_t>(ns, 1)));
            if (ns) HIPCK(hipMemcpy(D.d_send_idx, DL.send_idx.data(), sizeof(int) * ns, hipMemcpyHostToDevice));
        }
    }
    const size_t m = h->L[0].nvec;
    if (alloc_vec(&h->b, m) < 0 || alloc_vec(&h->u, m) < 0 || alloc_vec(&h->p, m) < 0 ||
        alloc_vec(&h->t, m) < 0 || alloc_vec(&h->r, m) < 0) return ERROR_ALLOC_MEM;
    HIPCK(hipMemsetAsync(h->u, 0, sizeof(double) * m, g_ctx.stream));
    HIPCK(hipMemsetAsync(h->p, 0, sizeof(double) * m, g_ctx.stream));
    const size_t mc = h->L[nl - 1].nvec;
    if (alloc_vec(&h->cp, mc) < 0 || alloc_vec(&h->cr, mc) < 0 || alloc_vec(&h->ct, mc) < 0 ||
        alloc_vec(&h->cbest, mc) < 0) return ERROR_ALLOC_MEM;
    h->ev.resize(64);
    for (auto& e : h->ev) { HIPCK(hipEventCreate(&e.a)); HIPCK(hipEventCreate(&e.b)); }
    HIPCK(hipStreamSynchronize(g_ctx.stream));
    // the local copies of the partitioned operators are only needed for the upload
    for (auto& DL : h->dist.L) { DL.A = HostCSR(); DL.P = HostCSR(); DL.R = HostCSR(); }
    h->upload_seconds = wall_seconds() - t0;
    return FASP_SUCCESS;
}

// ---------------------------------------------------------------------------
// smoothers on the resident level (PreMGSmoother.inl:49 / :155).  Jacobi and L1-diag
// are order independent, so pre (ascending) and post (descending) sweeps coincide.
// ---------------------------------------------------------------------------
static void materialise_zero(DevLevel& D)
{
    if (D.x_zero) {
        (void)hipMemsetAsync(D.x, 0, sizeof(double) * D.nvec, g_ctx.stream);
        D.x_zero = false;
    }
}

// Level schedule of one sequential sweep over the rows `seq` (in sweep order) of the host
// matrix A: level(i) = 1 + max level of the rows coupled to i (pattern of A and of A^T) that
// come earlier in the sweep.  Rows outside the sweep are not updated and impose nothing.
static int build_schedule(const HostCSR& A, const std::vector<int>& seq, DevLevel::Sched& S)
{
    const int n = A.row;
    std::vector<int> pos(n, -1), lev(n, 0);
    for (int q = 0; q < (int)seq.size(); ++q) pos[seq[q]] = q;
    // transpose pattern for the anti-dependencies of structurally unsymmetric matrices
    std::vector<int> tia(n + 2, 0), tja(A.nnz);
    for (int k = 0; k < A.nnz; ++k) if (A.ja[k] < n) tia[A.ja[k] + 2]++;
    for (int i = 2; i <= n + 1; ++i) tia[i] += tia[i - 1];
    for (int i = 0; i < n; ++i)
        for (int k = A.ia[i]; k < A.ia[i + 1]; ++k)
            if (A.ja[k] < n) tja[tia[A.ja[k] + 1]++] = i;
    int nlev = 0;
    for (int q = 0; q < (int)seq.size(); ++q) {
        const int i = seq[q];
        int l = 0;
        for (int k = A.ia[i]; k < A.ia[i + 1]; ++k) {
            const int j = A.ja[k];
            if (j != i && j < n && pos[j] >= 0 && pos[j] < q) l = std::max(l, lev[j]);
        }
        for (int k = tia[i]; k < tia[i + 1]; ++k) {
            const int j = tja[k];
            if (j != i && pos[j] >= 0 && pos[j] < q) l = std::max(l, lev[j]);
        }
        lev[i] = l + 1;
        nlev = std::max(nlev, l + 1);
    }
    S.ptr.assign(nlev + 1, 0);
    for (int i : seq) S.ptr[lev[i]]++;
    for (int l = 0; l < nlev; ++l) S.ptr[l + 1] += S.ptr[l];
    std::vector<int> cur(S.ptr.begin(), S.ptr.end() - 1), order(seq.size());
    for (int i : seq) order[cur[lev[i] - 1]++] = i;
    HIPCK(hipMalloc(&S.d_order, sizeof(int) * std::max<size_t>(order.size(), 1)));
    if (!order.empty()) HIPCK(hipMemcpy(S.d_order, order.data(), sizeof(int) * order.size(), hipMemcpyHostToDevice));
    S.built = true;
    return FASP_SUCCESS;
}

// one sequential sweep of schedule `kind` with update formula `form` (see k_seq_level)
static int seq_sweep(fasp_hip_amg* h, int level, int kind, int form, double w)
{
    DevLevel& D = h->L[level];
    DevLevel::Sched& S = D.sched[kind];
    if (!S.built) {
        const HostCSR& A = h->H.L[level].A;
        const int n = A.row;
        std::vector<int> seq;
        seq.reserve(n);
        const int* cf = h->H.L[level].cfmark.n ? h->H.L[level].cfmark.data() : nullptr;
        switch (kind) {
            case 0: for (int i = 0; i < n; ++i) seq.push_back(i); break;
            case 1: for (int i = n - 1; i >= 0; --i) seq.push_back(i); break;
            case 2: for (int i = 0; i < n; ++i) if (cf && cf[i] == 1) seq.push_back(i); break;
            case 3: for (int i = 0; i < n; ++i) if (!cf || cf[i] != 1) seq.push_back(i); break;
            default: for (int i = n - 2; i >= 0; --i) seq.push_back(i); break;
        }
        const int st = build_schedule(A, seq, S);
        if (st < 0) return st;
    }
    materialise_zero(D);
    const int L = D.A.lanes;
    const int nlev = (int)S.ptr.size() - 1;
    for (int l = 0; l < nlev; ++l) {
        const int lo = S.ptr[l], hi = S.ptr[l + 1];
        const int rpb = BLOCK / L;
        const int grid = std::max(1, std::min(MAXGRID, (hi - lo + rpb - 1) / rpb));
#define SEQ_LAUNCH(LL) hipLaunchKernelGGL((k_seq_level<LL>), dim3(grid), dim3(BLOCK), 0, g_ctx.stream, \
        (const int*)S.d_order, lo, hi, (const int*)D.A.ia, (const int*)D.A.ja, (const double*)D.A.val,    \
        (const double*)D.b, (const double*)D.diag, D.x, form, w)
        switch (L) {
            case 2: SEQ_LAUNCH(2); break;
            case 4: SEQ_LAUNCH(4); break;
            case 8: SEQ_LAUNCH(8); break;
            case 16: SEQ_LAUNCH(16); break;
            case 32: SEQ_LAUNCH(32); break;
            default: SEQ_LAUNCH(64); break;
        }
#undef SEQ_LAUNCH
    }
    return FASP_SUCCESS;
}

// Smoother dispatch of PreMGSmoother.inl:49 (pre) / :155 (post).  Jacobi and L1-diag are
// order independent, so their pre (ascending) and post (descending) sweeps coincide; the
// Gauss-Seidel / SOR family runs as level-scheduled sequential sweeps.
// fasp_smoother_dcsr_poly (ItrSmootherCSRpoly.c:67): per sweep r = b - A u, then the recurrence of
// Rr (:551) -- ndeg SpMVs and elementwise steps -- and u += correction.  Order independent, so the
// level may be row-partitioned (every SpMV input gets its halo).  Dinv and the coefficients depend
// on the matrix only: formed once per level on the host exactly as the reference does per call.
static int cg_smooth(fasp_hip_amg* h, int level, int nsweeps);  // defined after the Krylov drivers

static int poly_smooth(fasp_hip_amg* h, int level, int ndeg, int nsweeps)
{
    DevLevel& D = h->L[level];
    const int n = D.A.row;
    DevLevel::Poly& Q = D.poly;
    hipStream_t s = g_ctx.stream;
    if (!Q.built) {
        // every rank holds the whole host hierarchy: local row i is global row r0 + i, and the
        // norm (a maximum over ALL rows of the level) needs no exchange
        const HostCSR& A = h->H.L[level].A;
        const int r0 = D.replicated ? 0 : D.row0;
        std::vector<double> dinv((size_t)std::max(n, 1));
        double norm = 0.0;
        for (int gi = 0; gi < A.row; ++gi) {  // Diaginv :392 (first hit) and DinvAnorminf :428
            int j = A.ia[gi];
            for (; j < A.ia[gi + 1]; ++j) if (A.ja[j] == gi) break;
            const double di = 1.0 / A.val[j];
            if (gi >= r0 && gi < r0 + n) dinv[(size_t)(gi - r0)] = di;
            double temp = 0.0;
            for (int q = A.ia[gi]; q < A.ia[gi + 1]; ++q) temp += std::fabs(A.val[q]);
            temp *= di;
            norm = std::max(norm, temp);
        }
        double mu0 = norm;
        mu0 = 1.0 / mu0;
        const double mu1 = 4.0 * mu0, smu0 = std::sqrt(mu0), smu1 = std::sqrt(mu1);
        Q.k[1] = (mu0 + mu1) / 2.0;
        Q.k[2] = (smu0 + smu1) * (smu0 + smu1) / 2.0;
        Q.k[3] = mu0 * mu1;
        Q.k[4] = 2.0 * Q.k[3] / Q.k[2];
        Q.k[5] = (mu1 - 2.0 * smu0 * smu1 + mu0) / (mu1 + 2.0 * smu0 * smu1 + mu0);
        HIPCK(hipMalloc(&Q.dinv, sizeof(double) * std::max(n, 1)));
        HIPCK(hipMemcpy(Q.dinv, dinv.data(), sizeof(double) * n, hipMemcpyHostToDevice));
        for (double*& q : Q.w) {
            if (alloc_vec(&q, (size_t)D.nvec) < 0) return ERROR_ALLOC_MEM;
        }
        Q.built = true;
    }
    double *r = Q.w[0], *rbar = Q.w[1], *v0 = Q.w[2], *v1 = Q.w[3], *vnew = Q.w[4];
    const int G = vec_grid(n);
    for (int it = 0; it < nsweeps; ++it) {
        if (D.x_zero) {  // u == 0: r = b exactly, no matrix pass
            HIPCK(hipMemcpyAsync(r, D.b, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
            materialise_zero(D);
        } else {
            if (halo_exchange(D, D.x) < 0) return ERROR_MISC;
            d_resid(D.A, D.x, D.b, r);
        }
        hipLaunchKernelGGL(k_poly_scale, dim3(G), dim3(BLOCK), 0, s, n, Q.dinv, r, rbar);
        if (halo_exchange(D, rbar) < 0) return ERROR_MISC;
        d_mxv(D.A, rbar, v1);
        hipLaunchKernelGGL(k_poly_start, dim3(G), dim3(BLOCK), 0, s, n, Q.k[1], Q.k[2], Q.k[3], Q.dinv, rbar, v0, v1);
        if (ndeg <= 1) HIPCK(hipMemsetAsync(vnew, 0, sizeof(double) * n, s));  // the reference's correction stays zero
        for (int j = 1; j < ndeg; ++j) {
            if (halo_exchange(D, v1) < 0) return ERROR_MISC;
            d_mxv(D.A, v1, rbar);
            hipLaunchKernelGGL(k_poly_step, dim3(G), dim3(BLOCK), 0, s, n, Q.k[4], Q.k[5], Q.dinv, r, rbar, v0, v1, vnew);
        }
        d_axpy(n, 1.0, vnew, D.x);
    }
    return FASP_SUCCESS;
}

static int smooth(fasp_hip_amg* h, int level, bool post, int smoother, int order, int nsweeps, double relax, int ndeg)
{
    DevLevel& D = h->L[level];
    const int n = D.A.row;
    if (smoother == SMOOTHER_POLY) return poly_smooth(h, level, ndeg, nsweeps);
    if (smoother == SMOOTHER_JACOBIF) {  // fasp_smoother_dcsr_jacobi_ff, ItrSmootherCSR.c:34
        const Buf<int>& cf = h->H.L[level].cfmark;
        if (!D.replicated || cf.n != (size_t)n) {
            std::printf("### ERROR: fasp_hip: Jacobi-F needs the C/F marker of a classical hierarchy (one GPU)\n");
            return ERROR_AMG_SMOOTH_TYPE;
        }
        if (!D.d_mark) {
            HIPCK(hipMalloc(&D.d_mark, sizeof(int) * std::max(n, 1)));
            HIPCK(hipMemcpy(D.d_mark, cf.data(), sizeof(int) * n, hipMemcpyHostToDevice));
        }
        materialise_zero(D);
        for (int s = 0; s < nsweeps; ++s) {
            CsrArgs a{};
            a.x = D.x; a.y = D.xo; a.b = D.b; a.omega = relax; a.diag = D.diag; a.mark = D.d_mark;
            launch_csr<OP_L1DIAG>(D.A, a);
            std::swap(D.x, D.xo);
        }
        return FASP_SUCCESS;
    }
    if (smoother == SMOOTHER_JACOBI || smoother == SMOOTHER_L1DIAG) {
        for (int s = 0; s < nsweeps; ++s) {
            if (D.x_zero) {
                // zero initial guess: t_i = b_i exactly, no matrix pass
                if (smoother == SMOOTHER_JACOBI)
                    hipLaunchKernelGGL(k_jacobi_zero, dim3(vec_grid(n)), dim3(BLOCK), 0, g_ctx.stream, n, relax,
                                       D.b, D.diag, D.x);
                else
                    hipLaunchKernelGGL(k_l1_zero, dim3(vec_grid(n)), dim3(BLOCK), 0, g_ctx.stream, n, D.b, D.l1, D.x);
                D.x_zero = false;
                continue;
            }
            if (halo_exchange(D, D.x) < 0) return ERROR_MISC;
            CsrArgs a{};
            a.x = D.x; a.y = D.xo; a.b = D.b; a.omega = relax;
            if (smoother == SMOOTHER_JACOBI) { a.diag = D.diag; launch_csr<OP_JACOBI>(D.A, a); }
            else { a.diag = D.l1; launch_csr<OP_L1DIAG>(D.A, a); }
            std::swap(D.x, D.xo);
        }
        return FASP_SUCCESS;
    }
    if (!D.replicated) return ERROR_AMG_SMOOTH_TYPE;  // sequential sweeps are not distributed
    if (smoother == SMOOTHER_CG) return cg_smooth(h, level, nsweeps);
    const bool has_cf = h->H.L[level].cfmark.n == (size_t)n;
    if (smoother == SMOOTHER_GSF) {  // fasp_smoother_dcsr_gs_ff (ItrSmootherCSR.c:700): GS over the non-C rows, ascending, before and after
        if (!has_cf) {
            std::printf("### ERROR: fasp_hip: the F-point Gauss-Seidel smoother needs the C/F marker of a classical hierarchy\n");
            return ERROR_AMG_SMOOTH_TYPE;
        }
        for (int sw = 0; sw < nsweeps; ++sw) { const int st = seq_sweep(h, level, 3, 1, 0.0); if (st < 0) return st; }
        return FASP_SUCCESS;
    }
    auto rep = [&](int kind, int form, double w) -> int {  // nsweeps repetitions, as the `while (L--)` loops
        for (int s = 0; s < nsweeps; ++s) { const int st = seq_sweep(h, level, kind, form, w); if (st < 0) return st; }
        return FASP_SUCCESS;
    };
    int st = FASP_SUCCESS;
    switch (smoother) {
        case SMOOTHER_GS:
            if (order == NO_ORDER || !has_cf) st = rep(post ? 1 : 0, 0, 0.0);
            else if (order == CF_ORDER) {  // fasp_smoother_dcsr_gs_cf: pre C then F, post F then C
                for (int s = 0; s < nsweeps && st >= 0; ++s) {
                    st = seq_sweep(h, level, post ? 3 : 2, 1, 0.0);
                    if (st >= 0) st = seq_sweep(h, level, post ? 2 : 3, 1, 0.0);
                }
            }
            break;
        case SMOOTHER_SGS:
            for (int s = 0; s < nsweeps && st >= 0; ++s) {
                st = seq_sweep(h, level, 0, 1, 0.0);
                if (st >= 0) st = seq_sweep(h, level, 4, 1, 0.0);
            }
            break;
        case SMOOTHER_SOR: st = rep(post ? 1 : 0, 2, relax); break;
        case SMOOTHER_SSOR:
            st = rep(0, 2, relax);
            if (st >= 0) st = rep(1, 2, relax);
            break;
        case SMOOTHER_GSOR:
            if (!post) { st = rep(0, 0, 0.0); if (st >= 0) st = rep(1, 2, relax); }
            else       { st = rep(0, 2, relax); if (st >= 0) st = rep(1, 0, 0.0); }
            break;
        case SMOOTHER_SGSOR:
            if (!post) {
                st = rep(0, 0, 0.0); if (st >= 0) st = rep(1, 0, 0.0);
                if (st >= 0) st = rep(0, 2, relax); if (st >= 0) st = rep(1, 2, relax);
            } else {
                st = rep(0, 2, relax); if (st >= 0) st = rep(1, 2, relax);
                if (st >= 0) st = rep(0, 0, 0.0); if (st >= 0) st = rep(1, 0, 0.0);
            }
            break;
        default: return ERROR_AMG_SMOOTH_TYPE;
    }
    return st;
}

// ---------------------------------------------------------------------------
// coarsest level: safe-net CG without preconditioner (KrySPcg.c:60, called from
// PreMGUtil.inl:47 with StopType = STOP_REL_RES, maxit = MAX(250, MIN(n*n, 1000))).
// One host synchronisation per iteration: alpha is formed on the device.
// ---------------------------------------------------------------------------
// Coarsest levels that fit one CU's caches are solved by the single-workgroup kernels of
// small_solvers.hip.h (one launch, one synchronisation per solve instead of per iteration).
static bool small_coarse_ok(long long rows, long long stored_values)
{
    static int enabled = -1;
    if (enabled < 0) {
        const char* e = std::getenv("FASP_HIP_SMALL_COARSE");
        enabled = (e && std::atoi(e) == 0) ? 0 : 1;
    }
    return enabled && rows <= 4096 && stored_values <= 131072;
}
static SmallOut* small_out_dev() { return reinterpret_cast<SmallOut*>(g_ctx.d_partials2); }
static int small_out_fetch(SmallOut& o)
{
    HIPCK(hipMemcpyAsync(g_ctx.h_part, g_ctx.d_partials2, sizeof(SmallOut), hipMemcpyDeviceToHost, g_ctx.stream));
    HIPCK(hipStreamSynchronize(g_ctx.stream));
    std::memcpy(&o, g_ctx.h_part, sizeof(SmallOut));
    return 0;
}

static int coarse_spcg(fasp_hip_amg* h, DevLevel& D, double tol, int prtlvl)
{
    const DevCSR& A = D.A;
    const int m = A.row;
    const int nn = (int)((unsigned)m * (unsigned)m);
    const int MaxIt = std::max(250, std::min(nn, 1000));
    if (small_coarse_ok(m, A.nnz)) {
        SpcgArgs a{};
        a.A = SmallCSR{m, A.ia, A.ja, A.val};
        a.b = D.b; a.u = D.x; a.p = h->cp; a.r = h->cr; a.t = h->ct; a.u_best = h->cbest;
        a.tol = tol; a.MaxIt = MaxIt; a.x_zero = D.x_zero ? 1 : 0; a.out = small_out_dev(); a.nnz = A.nnz;
        // everything in LDS when it fits: vectors 5 m doubles, matrix 12 nnz + 4 (m + 1) bytes
        const size_t lds_v = sizeof(double) * 5 * (size_t)m;
        const size_t lds_m = 12 * (size_t)A.nnz + 4 * ((size_t)m + 1);
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)k_spcg_small<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            (void)hipFuncSetAttribute((const void*)k_spcg_small<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            attr = true;
        }
        if (g_tune.small_lds && lds_v + lds_m <= 148 * 1024)
            hipLaunchKernelGGL((k_spcg_small<true, true>), dim3(1), dim3(SMALL_BLOCK), lds_v + lds_m, g_ctx.stream, a);
        else if (g_tune.small_lds && lds_v <= 148 * 1024)
            hipLaunchKernelGGL((k_spcg_small<true, false>), dim3(1), dim3(SMALL_BLOCK), lds_v, g_ctx.stream, a);
        else
            hipLaunchKernelGGL((k_spcg_small<false, false>), dim3(1), dim3(SMALL_BLOCK), 0, g_ctx.stream, a);
        D.x_zero = false;
        SmallOut o;
        if (small_out_fetch(o) < 0) return ERROR_MISC;
        h->coarse_iters += o.iters;
        if (std::getenv("FASP_HIP_DEBUG_COARSE")) std::printf("[coarse small] status %d iters %d relres %.6e\n", o.status, o.iters, o.relres);
        return o.status;
    }
    const double maxdiff = tol * STAG_RATIO;
    int iter = 0, stag = 1, more_step = 1, iter_best = 0;
    double absres0 = BIGREAL, absres = BIGREAL, relres = BIGREAL, normu = BIGREAL, normr0 = BIGREAL;
    double reldiff, factor, alpha = 0.0, beta, temp1, temp2, absres_best = BIGREAL;
    double *p = h->cp, *r = h->cr, *t = h->ct, *u_best = h->cbest, *u = D.x;
    const double* b = D.b;
    double red[8];
    hipStream_t s = g_ctx.stream;
    (void)prtlvl;

    // u_best starts as zeros (calloc'ed work array, KrySPcg.c:88)
    HIPCK(hipMemsetAsync(u_best, 0, sizeof(double) * m, s));

    // r = b - A u  (u == 0 on entry from the cycle: r = b, no matrix pass)
    if (D.x_zero) {
        HIPCK(hipMemcpyAsync(r, b, sizeof(double) * m, hipMemcpyDeviceToDevice, s));
        HIPCK(hipMemsetAsync(u, 0, sizeof(double) * m, s));
        D.x_zero = false;
    } else {
        d_resid(A, u, b, r);
    }
    if (d_dot(m, r, r, red) < 0) return ERROR_MISC;  // z = r: (r,r) serves both ||r|| and (z,r)
    absres0 = std::sqrt(red[0]);
    normr0  = std::max(SMALLREAL, absres0);
    relres  = absres0 / normr0;
    if (relres < tol) goto FINISHED;
    HIPCK(hipMemcpyAsync(p, r, sizeof(double) * m, hipMemcpyDeviceToDevice, s));
    temp1 = red[0];

    {
        // Device-resident iteration state; the host queues `batch` iterations (SpMV + step kernel
        // each) without waiting and synchronises once per batch.  When one of the reference's
        // tests fires, k_spcg_step raises `stop`, the launches queued behind it return at once,
        // and the branch is replayed here from the recorded scalars (KrySPcg.c:172-330).
        const int batch = std::max(1, std::min(g_tune.spcg_batch, 64));
        if (!h->spcg_state) HIPCK(hipMalloc(&h->spcg_state, sizeof(SpcgState)));
        SpcgState S{};
        S.temp1 = temp1; S.temp1_prev = temp1; S.absres_best = absres_best; S.normr0 = normr0; S.tol = tol;
        S.maxdiff = maxdiff; S.iter = 0; S.iter_best = 0; S.stag = stag; S.MaxIt = MaxIt; S.stop = SPCG_RUN;
        S.absres = absres; S.relres = relres; S.alpha = 0.0;  // values before the first iteration
        HIPCK(hipMemcpyAsync(h->spcg_state, &S, sizeof(S), hipMemcpyHostToDevice, s));
        for (;;) {
            for (int q = 0; q < batch; ++q) {
                CsrArgs a{}; a.x = p; a.y = t; a.dotv = p; a.partials = g_ctx.d_partials; a.stop = &h->spcg_state->stop;
                SpcgStepArgs sa{};
                sa.m = m; sa.st = h->spcg_state; sa.t = t; sa.p = p; sa.u = u; sa.r = r; sa.u_best = u_best;
                sa.ntp = launch_csr<OP_MXV_DOT>(A, a);
                sa.tp_partials = g_ctx.d_partials;
                if (m <= 512 * 4) hipLaunchKernelGGL(k_spcg_step_reg<4>, dim3(1), dim3(512), 0, s, sa);
                else if (m <= 512 * 10) hipLaunchKernelGGL(k_spcg_step_reg<10>, dim3(1), dim3(512), 0, s, sa);
                else hipLaunchKernelGGL(k_spcg_step, dim3(1), dim3(SMALL_BLOCK), 0, s, sa);
            }
            HIPCK(hipMemcpyAsync(g_ctx.h_part, h->spcg_state, sizeof(SpcgState), hipMemcpyDeviceToHost, s));
            HIPCK(hipStreamSynchronize(s));
            std::memcpy(&S, g_ctx.h_part, sizeof(S));
            iter = S.iter; absres_best = S.absres_best; iter_best = S.iter_best;
            if (S.stop == SPCG_RUN) continue;
            // a test fired in iteration S.iter: finish that iteration as the reference does
            temp2 = S.tp; temp1 = S.temp1_prev;
            red[0] = S.rr; red[1] = S.uu; red[2] = S.pp; red[3] = S.maxu; red[4] = S.nan;
            // (on a breakdown the step kernel leaves absres / relres of the PREVIOUS iteration in the state,
            // which is what the reference's variables hold when it jumps to RESTORE_BESTSOL, KrySPcg.c:176)
            alpha = S.alpha; absres = S.absres; relres = S.relres;
            if (S.stop == SPCG_DIV0) goto RESTORE_BESTSOL;
            factor = absres / absres0; (void)factor; (void)alpha;
            if (S.stop == SPCG_NAN) { absres = BIGREAL; goto RESTORE_BESTSOL; }
            if (S.stop == SPCG_SOLSTAG) { iter = ERROR_SOLVER_SOLSTAG; break; }  // Check I
            if (S.stop == SPCG_MAXIT) { iter = MaxIt + 1; break; }
            normu = std::sqrt(red[1]);
            reldiff = std::fabs(S.alpha) * std::sqrt(red[2]) / normu;
            if ((stag <= MAX_STAG) & (reldiff < maxdiff)) {  // Check II
                d_resid(A, u, b, r);
                if (d_dot(m, r, r, red) < 0) return ERROR_MISC;
                absres = std::sqrt(red[0]);
                relres = absres / normr0;
                if (relres < tol) break;
                if (stag >= MAX_STAG) { iter = ERROR_SOLVER_STAG; break; }
                HIPCK(hipMemsetAsync(p, 0, sizeof(double) * m, s));
                ++stag;
            }
            if (relres < tol) {  // Check III: true residual
                d_resid(A, u, b, r);
                if (d_dot(m, r, r, red) < 0) return ERROR_MISC;
                absres = std::sqrt(red[0]);
                relres = absres / normr0;
                if (relres < tol) break;
                if (more_step >= MAX_RESTART) { iter = ERROR_SOLVER_TOLSMALL; break; }
                HIPCK(hipMemsetAsync(p, 0, sizeof(double) * m, s));
                ++more_step;
            }
            // every branch that gets here restarted: p was zeroed, so p = z + beta p = r
            absres0 = absres;
            temp2 = red[0];
            beta = temp2 / temp1;
            temp1 = temp2;
            d_axpby(m, 1.0, r, beta, p);
            S.temp1 = temp1; S.temp1_prev = temp1; S.stag = stag; S.stop = SPCG_RUN;
            HIPCK(hipMemcpyAsync(h->spcg_state, &S, sizeof(S), hipMemcpyHostToDevice, s));
            HIPCK(hipStreamSynchronize(s));  // S lives on this stack frame
        }
    }

RESTORE_BESTSOL:
    if (iter != iter_best) {
        d_resid(A, u_best, b, r);
        if (d_dot(m, r, r, red) < 0) return ERROR_MISC;
        absres_best = std::sqrt(red[0]);
        if (absres > absres_best + maxdiff || std::isnan(absres)) {
            HIPCK(hipMemcpyAsync(u, u_best, sizeof(double) * m, hipMemcpyDeviceToDevice, s));
            relres = absres_best / normr0;
        }
    }
FINISHED:
    if (std::getenv("FASP_HIP_DEBUG_COARSE")) std::printf("[coarse batched] iter %d relres %.6e absres %.6e best %d\n", iter, relres, absres, iter_best);
    if (iter > 0) h->coarse_iters += iter;
    if (iter > MaxIt) return ERROR_SOLVER_MAXIT;
    return iter;
}

struct KOps;
static KOps csr_ops(fasp_hip_amg* h, int level, bool with_pc);
// forward declarations (the coarse fallback and the preconditioner call each other's owners)
static int precond_amg(fasp_hip_amg* h, double* r, double** z);
static void itinfo(int ptrlvl, int stop_type, int iter, double relres, double absres, double factor);
struct PcgOut { double relres, absres, normr0; };
struct Hist {
    double* h; int cap; int n;
    void push(double v) { if (h && n < cap) h[n] = v; ++n; }
};

// Operator bundle of the Krylov drivers: the reference has one textual copy of every Krylov
// method per matrix format (KryPcg.c:96 / :386, KryPvgmres.c:66 / :416, ...); here the
// drivers are written once against these callbacks.
struct KOps {
    int    n = 0;       // owned entries
    size_t nvec = 0;    // vector length incl. ghosts
    bool   dist = false;
    const char* fmt = "CSR";
    std::function<int(double*)> halo;                                      // refresh ghost entries of v
    std::function<void(const double*, double*)> mxv;                       // y = A x
    std::function<void(const double*, const double*, double*)> resid;      // r = b - A x
    std::function<int(const double*, double*)> mxv_dot;                    // y = A x + partials of (y,x); returns #partials, < 0: unavailable
    std::function<int(double*, double**)> pc;                              // *out = B in (empty: identity)
    std::vector<double*>* ws = nullptr;                                    // GMRES workspace
    size_t* ws_len = nullptr;
    double** hh = nullptr;
    fasp_hip_amg* stats = nullptr;                                         // event pool for the SpMV timer
};

static void d_scale(int n, double a, double* x)
{
    if (a == 1.0) return;  // BlaArray.c:46
    hipLaunchKernelGGL(k_scale, dim3(vec_grid(n)), dim3(BLOCK), 0, g_ctx.stream, n, a, x);
}

// ---------------------------------------------------------------------------
// Variable-restart right-preconditioned GMRES family on device vectors:
//   mode 0  fasp_solver_dcsr_pvgmres    KryPvgmres.c:66-412
//   mode 1  fasp_solver_dcsr_pvfgmres   KryPvfgmres.c:67-384   (flexible)
//   mode 2  fasp_solver_dcsr_spvgmres   KrySPvgmres.c:68-441   (safe net; coarse-level fallback)
// The scalar control flow (restart adaptation, Givens rotations, back substitution, false-
// convergence check, best-iterate safety net) runs on the host exactly as in the reference;
// the modified Gram-Schmidt chain runs on the device without host round trips (k_mgs_step).
// `set` selects the workspace (0: level 0, 1: coarsest level); Lv is the level the operator
// acts on (halo plan); use_pc applies the AMG preconditioner (level 0 only).
// ---------------------------------------------------------------------------
static int gmres_device(KOps& K, const double* b, double* x, int mode_in, double tol, double abstol, int MaxIt,
                        int restart, int StopType, int PrtLvl, Hist* hist, PcgOut* out)
{
    const bool fixed = mode_in == 3;       // mode 3: fixed restart, fasp_solver_d*_pgmres (KryPgmres.c:66) ...
    const int  mode = fixed ? 0 : mode_in; // ... the text of mode 0 with four differences
    const int n = K.n;
    const size_t nv = K.nvec;
    const bool dist = K.dist;
    const int MIN_ITER = 0;
    const double epsmac = SMALLREAL, cr_max = 0.99, cr_min = 0.174, maxdiff = tol * STAG_RATIO;
    int iter = 0, i = 0, j, k, st;
    double r_norm, r_normb, gamma, t, red[8];
    double absres0 = BIGREAL, absres = BIGREAL, relres = BIGREAL, normu = BIGREAL;
    double b_norm = 0.0, den_norm = 0.0, epsilon = 0.0, cr = 1.0, r_norm_old = 0.0;
    const int d = 3, restart_max = restart, restart_min = 3;
    int Restart = fixed ? std::min(restart, MaxIt) : restart;
    const int Restart1 = restart + 1;
    int iter_best = 0;
    double absres_best = BIGREAL;
    hipStream_t s = g_ctx.stream;
    const int G = vec_grid(n);

    // workspace: p[0..Restart], w, x_best (mode 2), z[0..Restart) (mode 1)
    const size_t need = (size_t)Restart1 + 2 + (mode == 1 ? (size_t)Restart1 : 0);
    if (*K.ws_len != nv) {
        for (double* q : *K.ws) if (q) (void)hipFree(q);
        K.ws->clear();
        *K.ws_len = nv;
    }
    while (K.ws->size() < need) {
        double* q = nullptr;
        HIPCK(hipMalloc(&q, sizeof(double) * std::max<size_t>(nv, 1)));
        HIPCK(hipMemsetAsync(q, 0, sizeof(double) * nv, s));
        K.ws->push_back(q);
    }
    if (!*K.hh) HIPCK(hipMalloc(K.hh, sizeof(double) * 1024));
    if (Restart1 + 2 > 1024) return ERROR_INPUT_PAR;
    std::vector<double*>& W = *K.ws;
    double* const gm_hh = *K.hh;
    double** p = W.data();
    double*  w = W[Restart1];
    double*  x_best = W[Restart1 + 1];
    double** z = mode == 1 ? W.data() + Restart1 + 2 : nullptr;
    double*  r = nullptr;  // preconditioned / work vector (may alias an internal buffer)
    std::vector<double> rs(Restart1 + 1, 0.0), c(Restart + 1, 0.0), sn(Restart + 1, 0.0);
    std::vector<std::vector<double>> hh(Restart1, std::vector<double>(Restart + 1, 0.0));
    std::vector<double> norms((size_t)MaxIt + 2, 0.0);

    auto apply_pc = [&](double* in, double** outp) -> int {  // *outp = B in (pointer to the result)
        if (K.pc) return K.pc(in, outp);
        *outp = in;
        return FASP_SUCCESS;
    };
    auto true_residual = [&](const double* xx, double* rr) -> int {  // rr = b - A xx
        if (K.halo(const_cast<double*>(xx)) < 0) return ERROR_MISC;
        K.resid(xx, b, rr);
        return FASP_SUCCESS;
    };

    if (PrtLvl > PRINT_NONE)
        std::printf(fixed ? "\nCalling GMRes solver (%s) ...\n" : mode == 0 ? "\nCalling VGMRes solver (%s) ...\n"
                    : mode == 1 ? "\nCalling VFGMRes solver (%s) ...\n" : "\nCalling Safe VGMRes solver (%s) ...\n", K.fmt);

    if ((st = true_residual(x, p[0])) < 0) return st;
    if (mode == 1) { if (d_dot(n, b, b, red, dist) < 0) return ERROR_MISC; b_norm = std::sqrt(red[0]); }
    if (d_dot(n, p[0], p[0], red, dist) < 0) return ERROR_MISC;
    r_norm = std::sqrt(red[0]);

    if (mode == 1) {
        norms[0] = r_norm;
        if (PrtLvl >= PRINT_SOME) {
            std::printf("L2 norm of %s = %.10e.\n", "right-hand side", b_norm);
            std::printf("L2 norm of %s = %.10e.\n", "residual", r_norm);
        }
        den_norm = (b_norm > 0.0) ? b_norm : r_norm;
        epsilon = tol * den_norm;
        if (hist) hist->push(r_norm);
        if (r_norm < epsilon || r_norm < abstol) goto FINISHED;
        if (b_norm > 0.0) itinfo(PrtLvl, StopType, iter, norms[iter] / b_norm, norms[iter], 0);
        else itinfo(PrtLvl, StopType, iter, norms[iter], norms[iter], 0);
    } else {
        switch (StopType) {
            case STOP_REL_RES:
                absres0 = std::max(SMALLREAL, r_norm);
                relres = r_norm / absres0;
                break;
            case STOP_REL_PRECRES:
                if ((st = apply_pc(p[0], &r)) < 0) return st;
                if (d_dot(n, p[0], r, red, dist) < 0) return ERROR_MISC;
                r_normb = std::sqrt(red[0]);
                absres0 = std::max(SMALLREAL, r_normb);
                relres = r_normb / absres0;
                break;
            case STOP_MOD_REL_RES:
                if (d_dot(n, x, x, red, dist) < 0) return ERROR_MISC;
                normu = std::max(SMALLREAL, std::sqrt(red[0]));
                absres0 = r_norm;
                relres = absres0 / normu;
                break;
            default:
                std::printf("### ERROR: Unknown stopping type! [%s]\n", "fasp_solver_dcsr_pvgmres");
                goto FINISHED;
        }
        if (hist) hist->push(r_norm);
        if (mode == 0) { if (relres < tol || absres0 < abstol) goto FINISHED; }
        else           { if (relres < tol) goto FINISHED; }
        itinfo(PrtLvl, StopType, 0, relres, absres0, 0);
        norms[0] = relres;
    }

    while (iter < MaxIt && (!fixed || relres > tol)) {
        rs[0] = r_norm_old = r_norm;
        if (mode == 1 && r_norm == 0.0) { if (out) { out->relres = 0.0; out->absres = 0.0; out->normr0 = den_norm; } return iter; }
        if (mode != 1) d_scale(n, 1.0 / r_norm, p[0]);

        if (!fixed) {
            if (cr > cr_max || iter == 0) Restart = restart_max;
            else if (cr < cr_min) { /* keep */ }
            else { if (Restart - d > restart_min) Restart -= d; else Restart = restart_max; }
        }

        if (mode == 1) d_scale(n, 1.0 / r_norm, p[0]);

        i = 0;
        while (i < Restart && iter < MaxIt) {
            i++; iter++;
            if ((st = apply_pc(p[i - 1], &r)) < 0) return st;
            if (mode == 1 && r != z[i - 1])
                HIPCK(hipMemcpyAsync(z[i - 1], r, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
            if (K.halo(r) < 0) return ERROR_MISC;
            K.mxv(r, p[i]);
            // modified Gram-Schmidt on the device: hh_0 = (p_0, p_i); then i fused steps
            hipLaunchKernelGGL(k_dot, dim3(G), dim3(BLOCK), 0, s, n, p[0], p[i], g_ctx.d_partials);
            d_finalize_to(G, 1, 0u, gm_hh, dist);
            for (j = 0; j < i; j++) {
                hipLaunchKernelGGL(k_mgs_step, dim3(G), dim3(BLOCK), 0, s, n, (const double*)(gm_hh + j),
                                   (const double*)p[j], p[i], (const double*)(j + 1 < i ? p[j + 1] : nullptr),
                                   g_ctx.d_partials);
                d_finalize_to(G, 1, 0u, gm_hh + j + 1, dist);
            }
            HIPCK(hipMemcpyAsync(g_ctx.h_part, gm_hh, sizeof(double) * (i + 1), hipMemcpyDeviceToHost, s));
            HIPCK(hipStreamSynchronize(s));
            for (j = 0; j < i; j++) hh[j][i - 1] = g_ctx.h_part[j];
            t = std::sqrt(g_ctx.h_part[i]);
            hh[i][i - 1] = t;
            if (fixed ? (std::fabs(t) > SMALLREAL) : (t != 0.0)) d_scale(n, 1.0 / t, p[i]);
            for (j = 1; j < i; ++j) {
                t = hh[j - 1][i - 1];
                hh[j - 1][i - 1] = sn[j - 1] * hh[j][i - 1] + c[j - 1] * t;
                hh[j][i - 1] = -sn[j - 1] * t + c[j - 1] * hh[j][i - 1];
            }
            t = hh[i][i - 1] * hh[i][i - 1];
            t += hh[i - 1][i - 1] * hh[i - 1][i - 1];
            gamma = std::sqrt(t);
            if (fixed) gamma = std::max(gamma, SMALLREAL);
            else if (gamma == 0.0) gamma = epsmac;
            c[i - 1] = hh[i - 1][i - 1] / gamma;
            sn[i - 1] = hh[i][i - 1] / gamma;
            rs[i] = -sn[i - 1] * rs[i - 1];
            rs[i - 1] = c[i - 1] * rs[i - 1];
            hh[i - 1][i - 1] = sn[i - 1] * hh[i][i - 1] + c[i - 1] * hh[i - 1][i - 1];
            if (mode == 1) {
                r_norm = std::fabs(rs[i]);
                norms[iter] = r_norm;
                if (b_norm > 0) itinfo(PrtLvl, StopType, iter, norms[iter] / b_norm, norms[iter], norms[iter] / norms[iter - 1]);
                else itinfo(PrtLvl, StopType, iter, norms[iter], norms[iter], norms[iter] / norms[iter - 1]);
                if (hist) hist->push(r_norm);
                if (r_norm <= epsilon && iter >= MIN_ITER) break;
            } else {
                absres = r_norm = std::fabs(rs[i]);
                relres = absres / absres0;
                norms[iter] = relres;
                itinfo(PrtLvl, StopType, iter, relres, absres, norms[iter] / norms[iter - 1]);
                if (hist) hist->push(absres);
                if (mode == 0) { if (relres < tol && iter >= MIN_ITER) break; }
                else           { if (relres <= tol && iter >= MIN_ITER) break; }
            }
        }

        // back substitution (host) and solution update
        rs[i - 1] = rs[i - 1] / hh[i - 1][i - 1];
        for (k = i - 2; k >= 0; k--) {
            t = 0.0;
            for (j = k + 1; j < i; j++) t -= hh[k][j] * rs[j];
            t += rs[k];
            rs[k] = t / hh[k][k];
        }
        if (mode == 1) {
            HIPCK(hipMemcpyAsync(w, z[i - 1], sizeof(double) * n, hipMemcpyDeviceToDevice, s));
            d_scale(n, rs[i - 1], w);
            for (j = i - 2; j >= 0; j--) d_axpy(n, rs[j], z[j], w);
            r = w;
        } else {
            HIPCK(hipMemcpyAsync(w, p[i - 1], sizeof(double) * n, hipMemcpyDeviceToDevice, s));
            d_scale(n, rs[i - 1], w);
            for (j = i - 2; j >= 0; j--) d_axpy(n, rs[j], p[j], w);
            if ((st = apply_pc(w, &r)) < 0) return st;
        }
        d_axpy(n, 1.0, r, x);

        if (mode == 2) {  // safety net, KrySPvgmres.c:287-299
            if (d_norms(n, x, red, dist) < 0) return ERROR_MISC;
            if (std::isnan(red[0])) { absres = BIGREAL; goto RESTORE_BESTSOL; }
            if (absres < absres_best - maxdiff) {
                absres_best = absres;
                iter_best = iter;
                HIPCK(hipMemcpyAsync(x_best, x, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
            }
        }

        if ((mode == 0 && relres < tol && iter >= MIN_ITER) || (mode == 2 && relres <= tol && iter >= MIN_ITER) ||
            (mode == 1 && r_norm <= epsilon && iter >= MIN_ITER)) {
            const double computed_relres = relres;
            if ((st = true_residual(x, w)) < 0) return st;
            if (d_dot(n, w, w, red, dist) < 0) return ERROR_MISC;
            r_norm = std::sqrt(red[0]);
            switch (StopType) {
                case STOP_REL_RES:
                    if (mode == 1) relres = r_norm / den_norm;
                    else { absres = r_norm; relres = absres / absres0; }
                    break;
                case STOP_REL_PRECRES: {
                    double* zz = nullptr;
                    if ((st = apply_pc(w, &zz)) < 0) return st;
                    if (d_dot(n, zz, w, red, dist) < 0) return ERROR_MISC;
                    if (mode == 1) { r_normb = std::sqrt(red[0]); relres = r_normb / den_norm; }
                    else { absres = std::sqrt(red[0]); relres = absres / absres0; }
                } break;
                case STOP_MOD_REL_RES:
                    if (d_dot(n, x, x, red, dist) < 0) return ERROR_MISC;
                    normu = std::max(SMALLREAL, std::sqrt(red[0]));
                    if (mode == 1) relres = r_norm / normu;
                    else { absres = r_norm; relres = absres / normu; }
                    break;
            }
            if (mode != 1) norms[iter] = relres;
            if ((mode == 0 && relres < tol) || (mode != 0 && relres <= tol)) break;
            if (mode == 1 && PrtLvl >= PRINT_SOME)
                std::printf("### WARNING: False convergence! [%s:%d]\n", "fasp_solver_dcsr_pvfgmres", 328);
            HIPCK(hipMemcpyAsync(p[0], w, sizeof(double) * n, hipMemcpyDeviceToDevice, s));  // restart from the true residual
            i = 0;
            if (mode == 0 && PrtLvl >= PRINT_MORE) {
                std::printf("### WARNING: The computed relative residual = %.10e!\n", computed_relres);
                std::printf("### WARNING: The actual relative residual = %.10e!\n", relres);
            }
        }

        // residual vector of the restart (KryPvgmres.c:390-401)
        for (j = i; j > 0; j--) {
            rs[j - 1] = -sn[j - 1] * rs[j];
            rs[j] = c[j - 1] * rs[j];
        }
        if (i) hipLaunchKernelGGL(k_axpy_self, dim3(G), dim3(BLOCK), 0, s, n, rs[i] - 1.0, p[i]);
        for (j = i - 1; j > 0; j--) d_axpy(n, rs[j], p[j], p[i]);
        if (i) {
            hipLaunchKernelGGL(k_axpy_self, dim3(G), dim3(BLOCK), 0, s, n, rs[0] - 1.0, p[0]);
            d_axpy(n, 1.0, p[i], p[0]);
        }
        cr = r_norm / r_norm_old;
    }

RESTORE_BESTSOL:
    if (mode == 2 && iter != iter_best) {  // KrySPvgmres.c:357-389
        if ((st = true_residual(x_best, w)) < 0) return st;
        if (d_dot(n, w, w, red, dist) < 0) return ERROR_MISC;
        absres_best = std::sqrt(red[0]);
        if (absres > absres_best + maxdiff || std::isnan(absres)) {
            if (PrtLvl > PRINT_NONE)
                std::printf("### WARNING: Discard current iteration. Restore iteration %d!\n", iter_best);
            HIPCK(hipMemcpyAsync(x, x_best, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
            relres = absres_best / absres0;
        }
    }

FINISHED:
    {
        const double fr = (mode == 1) ? r_norm / den_norm : relres;
        if (PrtLvl > PRINT_NONE) {
            if (iter > MaxIt) std::printf("### WARNING: MaxIt = %d reached with relative residual %.10e.\n", MaxIt, fr);
            else if (iter >= 0) std::printf("Number of iterations = %d with relative residual %.10e.\n", iter, fr);
        }
        if (out) { out->relres = fr; out->absres = r_norm; out->normr0 = (mode == 1) ? den_norm : absres0; }
    }
    HIPCK(hipStreamSynchronize(s));
    if (iter >= MaxIt) return ERROR_SOLVER_MAXIT;
    return iter;
}

// ---------------------------------------------------------------------------
// BiCGstab (KryPbcgs.c:62 / :400: the formulation of MATLAB's bicgstab with half steps,
// stagnation counters and the minimal-residual iterate) on device vectors
// ---------------------------------------------------------------------------
static int bicgstab_device(KOps& K, const double* b, double* x, double tol, int MaxIt, int PrtLvl, Hist* hist,
                           PcgOut* out)
{
    const int m = K.n;
    const size_t nv = K.nvec;
    const bool dist = K.dist;
    hipStream_t s = g_ctx.stream;
    if (*K.ws_len != nv) {
        for (double* q : *K.ws) if (q) (void)hipFree(q);
        K.ws->clear();
        *K.ws_len = nv;
    }
    while (K.ws->size() < 9) {
        double* q = nullptr;
        HIPCK(hipMalloc(&q, sizeof(double) * std::max<size_t>(nv, 1)));
        HIPCK(hipMemsetAsync(q, 0, sizeof(double) * nv, s));
        K.ws->push_back(q);
    }
    std::vector<double*>& W = *K.ws;
    double *r = W[0], *rt = W[1], *p = W[2], *v = W[3], *xhalf = W[4], *sv = W[5], *t = W[6], *xmin = W[7], *tmp = W[8];
    double *ph = nullptr, *sh = nullptr;  // preconditioned vectors (may alias the preconditioner's output)
    double red[8];
    auto cp = [&](double* dst, const double* src) -> int {
        HIPCK(hipMemcpyAsync(dst, src, sizeof(double) * m, hipMemcpyDeviceToDevice, s));
        return 0;
    };
    auto nrm2 = [&](const double* y, double& val) -> int {
        if (d_dot(m, y, y, red, dist) < 0) return ERROR_MISC;
        val = std::sqrt(red[0]);
        return 0;
    };
    auto dot = [&](const double* y, const double* z, double& val) -> int {
        if (d_dot(m, y, z, red, dist) < 0) return ERROR_MISC;
        val = red[0];
        return 0;
    };
    auto resid = [&](double* xx, double* rr) -> int {  // rr = b - A xx
        if (K.halo(xx) < 0) return ERROR_MISC;
        K.resid(xx, b, rr);
        return 0;
    };
    auto apply_pc = [&](double* in, double** outp) -> int {
        if (K.pc) return K.pc(in, outp);
        *outp = in;
        return FASP_SUCCESS;
    };
    double n2b, tolb, relres = BIGREAL, absres0 = BIGREAL, absres = BIGREAL;
    double alpha, beta, omega, rho, rho1, rtv, tt, st_, normr, normr_act, normph, normx, imin, norm_sh, norm_xhalf, normrmin = 0.0;
    int iter = 0, stag = 1, moresteps = 1, maxmsteps = 1, flag = 1, maxstagsteps = 3, st;
    (void)stag; (void)moresteps; (void)maxmsteps;

    if (PrtLvl > PRINT_NONE) std::printf("\nCalling BiCGstab solver (%s) ...\n", K.fmt);
    if (nrm2(b, n2b) < 0) return ERROR_MISC;
    if (cp(xmin, x) < 0) return ERROR_MISC;
    imin = 0;
    tolb = n2b * tol;
    if (resid(x, r) < 0) return ERROR_MISC;
    if (nrm2(r, normr) < 0) return ERROR_MISC;
    normr_act = normr;
    relres = normr / n2b;
    if (hist) hist->push(normr);
    if (normr <= tolb) { flag = 0; iter = 0; goto FINISHED; }
    itinfo(PrtLvl, STOP_REL_RES, iter, relres, n2b, 0.0);
    if (cp(rt, r) < 0) return ERROR_MISC;
    normrmin = normr;
    rho = 1.0; omega = 1.0; stag = 0; alpha = 0.0;
    moresteps = 0; maxmsteps = 10;

    for (iter = 1; iter <= MaxIt; iter++) {
        rho1 = rho;
        if (dot(rt, r, rho) < 0) return ERROR_MISC;
        if ((rho == 0.0) || (std::fabs(rho) >= DBL_MAX)) { flag = 4; goto FINISHED; }
        if (iter == 1) { if (cp(p, r) < 0) return ERROR_MISC; }
        else {
            beta = (rho / rho1) * (alpha / omega);
            if ((beta == 0) || (std::fabs(beta) > DBL_MAX)) { flag = 4; goto FINISHED; }
            d_axpy(m, -omega, v, p);
            d_axpby(m, 1.0, r, beta, p);
        }
        if ((st = apply_pc(p, &ph)) < 0) return st;
        if (K.halo(ph) < 0) return ERROR_MISC;
        K.mxv(ph, v);
        if (dot(rt, v, rtv) < 0) return ERROR_MISC;
        if ((rtv == 0.0) || (std::fabs(rtv) > DBL_MAX)) { flag = 4; goto FINISHED; }
        alpha = rho / rtv;
        if (std::fabs(alpha) > DBL_MAX) {
            flag = 4;
            std::printf("### WARNING: Divided by zero! [%s:%d]\n", "fasp_solver_dcsr_pbcgs", 178);
            goto FINISHED;
        }
        if (nrm2(x, normx) < 0 || nrm2(ph, normph) < 0) return ERROR_MISC;
        if (std::fabs(alpha) * normph < DBL_EPSILON * normx) stag = stag + 1; else stag = 0;
        if (cp(xhalf, x) < 0) return ERROR_MISC;
        d_axpy(m, alpha, ph, xhalf);   // xhalf = alpha ph + x
        if (cp(sv, r) < 0) return ERROR_MISC;
        d_axpy(m, -alpha, v, sv);      // s = -alpha v + r
        if (nrm2(sv, normr) < 0) return ERROR_MISC;
        normr_act = normr;
        absres = normr_act;
        itinfo(PrtLvl, STOP_REL_RES, iter, normr_act / n2b, absres, absres / absres0);
        if (hist) hist->push(absres);
        if ((normr <= tolb) || (stag >= maxstagsteps) || moresteps) {
            if (resid(xhalf, sv) < 0) return ERROR_MISC;
            if (nrm2(sv, normr_act) < 0) return ERROR_MISC;
            if (normr_act <= tolb) {
                if (cp(x, xhalf) < 0) return ERROR_MISC;
                flag = 0; imin = iter - 0.5;
                goto FINISHED;
            } else {
                if ((stag >= maxstagsteps) && (moresteps == 0)) stag = 0;
                moresteps = moresteps + 1;
                if (moresteps >= maxmsteps) { flag = 3; if (cp(x, xhalf) < 0) return ERROR_MISC; goto FINISHED; }
            }
        }
        if (stag >= maxstagsteps) { flag = 3; goto FINISHED; }
        if (normr_act < normrmin) {
            normrmin = normr_act;
            if (cp(xmin, xhalf) < 0) return ERROR_MISC;
            imin = iter - 0.5;
        }
        if ((st = apply_pc(sv, &sh)) < 0) return st;
        if (K.halo(sh) < 0) return ERROR_MISC;
        K.mxv(sh, t);
        if (dot(t, t, tt) < 0) return ERROR_MISC;
        if ((tt == 0) || (tt >= DBL_MAX)) { flag = 4; goto FINISHED; }
        if (dot(sv, t, st_) < 0) return ERROR_MISC;
        omega = st_ / tt;
        if (std::fabs(omega) > DBL_MAX) { flag = 4; goto FINISHED; }
        if (nrm2(sh, norm_sh) < 0 || nrm2(xhalf, norm_xhalf) < 0) return ERROR_MISC;
        if (std::fabs(omega) * norm_sh < DBL_EPSILON * norm_xhalf) stag = stag + 1; else stag = 0;
        if (cp(x, xhalf) < 0) return ERROR_MISC;
        d_axpy(m, omega, sh, x);       // x = omega sh + xhalf
        if (cp(r, sv) < 0) return ERROR_MISC;
        d_axpy(m, -omega, t, r);       // r = -omega t + s
        if (nrm2(r, normr) < 0) return ERROR_MISC;
        normr_act = normr;
        if ((normr <= tolb) || (stag >= maxstagsteps) || moresteps) {
            if (resid(x, r) < 0) return ERROR_MISC;
            if (nrm2(r, normr_act) < 0) return ERROR_MISC;
            if (normr_act <= tolb) { flag = 0; goto FINISHED; }
            else {
                if ((stag >= maxstagsteps) && (moresteps == 0)) stag = 0;
                moresteps = moresteps + 1;
                if (moresteps >= maxmsteps) { flag = 3; goto FINISHED; }
            }
        }
        if (normr_act < normrmin) { normrmin = normr_act; if (cp(xmin, x) < 0) return ERROR_MISC; imin = iter; }
        if (stag >= maxstagsteps) { flag = 3; goto FINISHED; }
        if (PrtLvl >= PRINT_MORE) std::printf("### WARNING: The actual relative residual = %.10e!\n", relres);
        absres0 = absres;
    }
FINISHED:
    if (flag == 0) relres = normr_act / n2b;
    else {
        if (resid(xmin, tmp) < 0) return ERROR_MISC;
        if (nrm2(tmp, normr) < 0) return ERROR_MISC;
        if (normr <= normr_act) { if (cp(x, xmin) < 0) return ERROR_MISC; iter = (int)imin; relres = normr / n2b; }
        else relres = normr_act / n2b;
    }
    if (PrtLvl > PRINT_NONE) {
        if (iter > MaxIt) std::printf("### WARNING: MaxIt = %d reached with relative residual %.10e.\n", MaxIt, relres);
        else if (iter >= 0) std::printf("Number of iterations = %d with relative residual %.10e.\n", iter, relres);
    }
    if (out) { out->relres = relres; out->absres = relres * n2b; out->normr0 = n2b; }
    HIPCK(hipStreamSynchronize(s));
    return iter > MaxIt ? ERROR_SOLVER_MAXIT : iter;
}

// ---------------------------------------------------------------------------
// The remaining `itsolver_type`s of fasp_solver_dcsr_itsolver (SolCSR.c:56): MinRes, GCG, GCR.
// Host control flow as in the reference, vectors and every operation on the device; these
// are completeness modes (one host round trip per scalar), the tuned drivers are CG and GMRES.
// ---------------------------------------------------------------------------
struct KVecOps {
    KOps& K;
    const int m;
    const size_t nv;
    hipStream_t s;
    double red[8];
    explicit KVecOps(KOps& K_) : K(K_), m(K_.n), nv(K_.nvec), s(g_ctx.stream) {}
    int ensure(size_t count)
    {
        if (*K.ws_len != nv) {
            for (double* q : *K.ws) if (q) (void)hipFree(q);
            K.ws->clear();
            *K.ws_len = nv;
        }
        while (K.ws->size() < count) {
            double* q = nullptr;
            HIPCK(hipMalloc(&q, sizeof(double) * std::max<size_t>(nv, 1)));
            HIPCK(hipMemsetAsync(q, 0, sizeof(double) * nv, s));
            K.ws->push_back(q);
        }
        return 0;
    }
    double* vec(size_t i) { return (*K.ws)[i]; }
    int cp(double* dst, const double* src)
    {
        if (dst != src) HIPCK(hipMemcpyAsync(dst, src, sizeof(double) * m, hipMemcpyDeviceToDevice, s));
        return 0;
    }
    int zero(double* dst) { HIPCK(hipMemsetAsync(dst, 0, sizeof(double) * m, s)); return 0; }
    int dot(const double* y, const double* z, double& val)
    {
        if (d_dot(m, y, z, red, K.dist) < 0) return ERROR_MISC;
        val = red[0];
        return 0;
    }
    int nrm2(const double* y, double& val)
    {
        if (d_dot(m, y, y, red, K.dist) < 0) return ERROR_MISC;
        val = std::sqrt(red[0]);
        return 0;
    }
    int mxv(double* x, double* y)  // y = A x (x's ghost entries refreshed first)
    {
        if (K.halo(x) < 0) return ERROR_MISC;
        K.mxv(x, y);
        return 0;
    }
    int resid(double* x, const double* b, double* r)  // r = b - A x
    {
        if (K.halo(x) < 0) return ERROR_MISC;
        K.resid(x, b, r);
        return 0;
    }
    int pc(double* in, double* dst)  // dst = B in (the preconditioner's own output buffer is copied out)
    {
        double* o = in;
        if (K.pc) { const int st = K.pc(in, &o); if (st < 0) return st; }
        return cp(dst, o);
    }
};
#define KCK(expr) do { const int st__ = (expr); if (st__ < 0) return st__; } while (0)

// fasp_solver_dcsr_pminres, KryPminres.c:61-448
static int minres_device(KOps& K, const double* b, double* u, double tol, double abstol, int MaxIt, int StopType,
                         int PrtLvl, Hist* hist, PcgOut* out)
{
    KVecOps V(K);
    const int m = V.m;
    const double maxdiff = tol * STAG_RATIO, sol_inf_tol = SMALLREAL;
    int iter = 0, stag = 1, more_step = 1;
    double absres0 = BIGREAL, absres = BIGREAL, normr0 = BIGREAL, relres = BIGREAL;
    double normu2 = BIGREAL, normuu, normp, factor, alpha, alpha0, alpha1, temp2, red[8];
    KCK(V.ensure(11));
    double *p0 = V.vec(0), *p1 = V.vec(1), *p2 = V.vec(2), *z0 = V.vec(3), *z1 = V.vec(4), *t0 = V.vec(5),
           *t1 = V.vec(6), *t = V.vec(7), *tp = V.vec(8), *tz = V.vec(9), *r = V.vec(10);
    auto resnorm = [&]() -> int {  // :228-247 and its two copies
        switch (StopType) {
            case STOP_REL_RES:
                KCK(V.dot(r, r, temp2)); absres = std::sqrt(temp2); relres = absres / normr0; break;
            case STOP_REL_PRECRES:
                KCK(V.pc(r, t)); KCK(V.dot(r, t, temp2)); temp2 = std::fabs(temp2);
                absres = std::sqrt(temp2); relres = absres / normr0; break;
            case STOP_MOD_REL_RES:
                KCK(V.dot(r, r, temp2)); absres = std::sqrt(temp2); relres = absres / normu2; break;
        }
        return 0;
    };
    auto restart = [&]() -> int {  // :331-368 == :409-446
        KCK(V.zero(p0));
        KCK(V.pc(r, p1));
        KCK(V.mxv(p1, tp));
        KCK(V.pc(tp, tz));
        KCK(V.dot(tz, tp, normp));
        normp = std::sqrt(normp);
        KCK(V.cp(t, p1));
        KCK(V.zero(t0)); KCK(V.zero(z0)); KCK(V.zero(t1)); KCK(V.zero(z1)); KCK(V.zero(p1));
        d_axpy(m, 1 / normp, t, p1);
        d_axpy(m, 1 / normp, tp, t1);
        d_axpy(m, 1 / normp, tz, z1);
        return 0;
    };
    if (PrtLvl > PRINT_NONE) std::printf("\nCalling MinRes solver (%s) ...\n", K.fmt);
    KCK(V.zero(p0));
    KCK(V.resid(u, b, r));
    KCK(V.pc(r, p1));
    switch (StopType) {
        case STOP_REL_RES:
            KCK(V.nrm2(r, absres0)); normr0 = std::max(SMALLREAL, absres0); relres = absres0 / normr0; break;
        case STOP_REL_PRECRES:
            KCK(V.dot(r, p1, temp2)); absres0 = std::sqrt(temp2);
            normr0 = std::max(SMALLREAL, absres0); relres = absres0 / normr0; break;
        case STOP_MOD_REL_RES:
            KCK(V.nrm2(r, absres0)); KCK(V.nrm2(u, normu2)); normu2 = std::max(SMALLREAL, normu2);
            relres = absres0 / normu2; break;
        default:
            std::printf("### ERROR: Unknown stopping type! [%s]\n", "fasp_solver_dcsr_pminres");
            goto FINISHED;
    }
    if (hist) hist->push(absres0);
    if (relres < tol || absres0 < abstol) goto FINISHED;
    itinfo(PrtLvl, StopType, iter, relres, absres0, 0.0);
    KCK(V.mxv(p1, tp));
    KCK(V.pc(tp, tz));
    KCK(V.dot(tz, tp, normp));
    normp = std::sqrt(std::fabs(normp));
    KCK(V.cp(t, p1));
    KCK(V.zero(p1));
    d_axpy(m, 1 / normp, t, p1);
    KCK(V.zero(t0)); KCK(V.zero(z0)); KCK(V.zero(t1)); KCK(V.zero(z1));
    d_axpy(m, 1.0 / normp, tp, t1);
    d_axpy(m, 1.0 / normp, tz, z1);

    while (iter++ < MaxIt) {
        KCK(V.dot(r, z1, alpha));
        d_axpy(m, alpha, p1, u);
        d_axpy(m, -alpha, t1, r);
        KCK(V.mxv(z1, t));
        KCK(V.dot(z1, t, alpha1));
        KCK(V.mxv(z0, t));
        KCK(V.dot(z1, t, alpha0));
        KCK(V.cp(p2, z1));
        d_axpy(m, -alpha1, p1, p2);
        d_axpy(m, -alpha0, p0, p2);
        KCK(V.mxv(p2, tp));
        KCK(V.pc(tp, tz));
        KCK(V.dot(tz, tp, normp));
        normp = std::sqrt(std::fabs(normp));
        KCK(V.cp(t, p2));
        KCK(V.zero(p2));
        d_axpy(m, 1 / normp, t, p2);
        KCK(V.cp(p0, p1)); KCK(V.cp(p1, p2)); KCK(V.cp(t0, t1)); KCK(V.cp(z0, z1));
        KCK(V.zero(t1)); KCK(V.zero(z1));
        d_axpy(m, 1 / normp, tp, t1);
        d_axpy(m, 1 / normp, tz, z1);
        if (d_norms(m, u, red, K.dist) < 0) return ERROR_MISC;  // ||u||^2, max|u|
        normu2 = std::sqrt(red[0]);
        KCK(resnorm());
        factor = absres / absres0;
        itinfo(PrtLvl, StopType, iter, relres, absres, factor);
        if (hist) hist->push(absres);

        if (factor > 0.9) {  // Check I, II (:256-373)
            if (red[1] <= sol_inf_tol) {
                if (PrtLvl > PRINT_MIN)
                    std::printf("### WARNING: Iteration stopped -- solution almost zero! [%s:%d]\n", "fasp_solver_dcsr_pminres", 262);
                iter = ERROR_SOLVER_SOLSTAG;
                break;
            }
            KCK(V.nrm2(p1, normuu));
            normuu = std::fabs(alpha) * (normuu / normu2);
            if (normuu < maxdiff) {
                if (stag < MAX_STAG && PrtLvl >= PRINT_MORE) {
                    std::printf("||u-u'|| = %.10e and the comp. rel. res. = %.10e.\n", normuu, relres);
                    std::printf("### WARNING: Iteration restarted -- stagnation! [%s:%d]\n", "fasp_solver_dcsr_pminres", 276);
                }
                KCK(V.resid(u, b, r));
                KCK(resnorm());
                if (PrtLvl >= PRINT_MORE) std::printf("### WARNING: The actual relative residual = %.10e!\n", relres);
                if (relres < tol) break;
                if (stag >= MAX_STAG) {
                    if (PrtLvl > PRINT_MIN)
                        std::printf("### WARNING: Iteration stopped -- staggnation! [%s:%d]\n", "fasp_solver_dcsr_pminres", 318);
                    iter = ERROR_SOLVER_STAG;
                    break;
                }
                ++stag;
                KCK(restart());
            }
        }

        if (relres < tol) {  // Check III (:376-447)
            if (PrtLvl >= PRINT_MORE) std::printf("### WARNING: The computed relative residual = %.10e!\n", relres);
            KCK(V.resid(u, b, r));
            KCK(resnorm());
            if (PrtLvl >= PRINT_MORE) std::printf("### WARNING: The actual relative residual = %.10e!\n", relres);
            if (relres < tol) break;
            if (more_step >= MAX_RESTART) {
                if (PrtLvl > PRINT_MIN)
                    std::printf("### WARNING: The tolerence might be too small! [%s:%d]\n", "fasp_solver_dcsr_pminres", 412);
                iter = ERROR_SOLVER_TOLSMALL;
                break;
            }
            ++more_step;
            KCK(restart());
        }
        absres0 = absres;
    }
FINISHED:
    if (PrtLvl > PRINT_NONE) {
        if (iter > MaxIt) std::printf("### WARNING: MaxIt = %d reached with relative residual %.10e.\n", MaxIt, relres);
        else if (iter >= 0) std::printf("Number of iterations = %d with relative residual %.10e.\n", iter, relres);
    }
    if (out) { out->relres = relres; out->absres = absres; out->normr0 = normr0; }
    HIPCK(hipStreamSynchronize(V.s));
    return iter > MaxIt ? ERROR_SOLVER_MAXIT : iter;
}

// fasp_solver_dcsr_pgcg, KryPgcg.c:60-195.  The reference allocates all MaxIt search directions
// up front; here a direction is allocated when its iteration is reached.
static int gcg_device(KOps& K, const double* b, double* u, double tol, double abstol, int MaxIt, int StopType,
                      int PrtLvl, Hist* hist, PcgOut* out)
{
    KVecOps V(K);
    const int m = V.m;
    int iter = 0, i;
    double absres0 = BIGREAL, absres = BIGREAL, relres = BIGREAL, normb = BIGREAL, alpha, factor, num, den, beta;
    KCK(V.ensure(4));
    double *r = V.vec(0), *Br = V.vec(1), *Ap = V.vec(2);
    auto P = [&](int k) { return V.vec(3 + (size_t)k); };
    auto vmv = [&](double* x, const double* y, double& val) -> int {  // y^T A x, BlaSpmvCSR.c:839
        KCK(V.mxv(x, Ap));
        return V.dot(y, Ap, val);
    };
    auto step = [&](double* p) -> int {  // alpha = (r,p)/(p,Ap); u += alpha p; r -= alpha A p
        KCK(V.dot(r, p, num));
        KCK(vmv(p, p, den));
        alpha = num / den;
        d_axpy(m, alpha, p, u);
        d_axpy(m, -1.0 * alpha, Ap, r);  // Ap still holds A p
        KCK(V.nrm2(r, absres));
        factor = absres / absres0;
        relres = absres / normb;
        itinfo(PrtLvl, StopType, iter, relres, absres, factor);
        if (hist) hist->push(absres);
        return 0;
    };
    if (PrtLvl > PRINT_NONE) std::printf("\nCalling GCG solver (%s) ...\n", K.fmt);
    KCK(V.nrm2(b, normb));
    KCK(V.resid(u, b, r));
    KCK(V.pc(r, P(0)));
    KCK(step(P(0)));
    absres0 = absres;
    for (iter = 1; iter < MaxIt; iter++) {
        KCK(V.ensure(4 + (size_t)iter));
        r = V.vec(0); Br = V.vec(1); Ap = V.vec(2);
        double* pi = P(iter);
        KCK(V.pc(r, Br));
        KCK(V.cp(pi, Br));
        for (i = 0; i < iter; i++) {
            KCK(vmv(Br, P(i), num));
            KCK(vmv(P(i), P(i), den));
            beta = (-1.0) * (num / den);
            d_axpy(m, beta, P(i), pi);
        }
        KCK(step(pi));
        if (relres < tol || absres < abstol) break;
        absres0 = absres;
    }
    if (PrtLvl > PRINT_NONE) {
        if (iter > MaxIt) std::printf("### WARNING: MaxIt = %d reached with relative residual %.10e.\n", MaxIt, relres);
        else if (iter >= 0) std::printf("Number of iterations = %d with relative residual %.10e.\n", iter, relres);
    }
    if (out) { out->relres = relres; out->absres = absres; out->normr0 = normb; }
    HIPCK(hipStreamSynchronize(V.s));
    return iter > MaxIt ? ERROR_SOLVER_MAXIT : iter;
}

// fasp_solver_dcsr_pgcr, KryPgcr.c:55-425 (+ dense_aAtxpby :450)
static int gcr_device(KOps& K, const double* b, double* x, double tol, double abstol, int MaxIt, int restart_in,
                      int StopType, int PrtLvl, Hist* hist, PcgOut* out)
{
    KVecOps V(K);
    const int n = V.m;
    int iter = 0, i, j, k, rst = -1;
    double gamma, alpha, beta, checktol, absres0 = BIGREAL, absres = BIGREAL, relres = BIGREAL, prev;
    const int Restart = std::min(restart_in, MaxIt);
    (void)abstol;
    KCK(V.ensure(1 + 2 * (size_t)std::max(Restart, 0)));
    double* r = V.vec(0);
    auto Z = [&](int q) { return V.vec(1 + (size_t)q); };
    auto Cv = [&](int q) { return V.vec(1 + (size_t)Restart + (size_t)q); };
    std::vector<double> alp((size_t)std::max(Restart, 1)), tmpx((size_t)std::max(Restart, 1));
    std::vector<std::vector<double>> h((size_t)std::max(Restart, 1), std::vector<double>((size_t)std::max(Restart, 1), 0.0));
    if (PrtLvl > PRINT_NONE) std::printf("\nCalling GCR solver (%s) ...\n", K.fmt);
    KCK(V.resid(x, b, r));
    KCK(V.dot(r, r, absres));
    absres0 = std::max(SMALLREAL, absres);
    relres = absres / absres0;
    itinfo(PrtLvl, StopType, 0, relres, std::sqrt(absres0), 0.0);
    if (hist) hist->push(std::sqrt(absres));
    prev = relres;
    checktol = std::max(tol * tol * absres0, absres * 1.0e-4);
    while (iter < MaxIt && std::sqrt(relres) > tol) {
        i = -1;
        rst++;
        while (i < Restart - 1 && iter < MaxIt) {
            i++;
            iter++;
            KCK(V.pc(r, Z(i)));
            KCK(V.mxv(Z(i), Cv(i)));
            for (j = 0; j < i; j++) {  // modified Gram-Schmidt
                KCK(V.dot(Cv(j), Cv(i), gamma));
                h[i][j] = gamma / h[j][j];
                d_axpy(n, -h[i][j], Cv(j), Cv(i));
            }
            KCK(V.dot(Cv(i), Cv(i), gamma));
            h[i][i] = gamma;
            KCK(V.dot(Cv(i), r, alpha));
            beta = alpha / gamma;
            alp[i] = beta;
            d_axpy(n, -beta, Cv(i), r);
            absres = absres - alpha * alpha / gamma;
            if (absres < checktol) {
                KCK(V.dot(r, r, absres));
                checktol = std::max(tol * tol * absres0, absres * 1.0e-4);
            }
            relres = absres / absres0;
            itinfo(PrtLvl, StopType, iter, std::sqrt(relres), std::sqrt(absres), std::sqrt(relres / prev));
            if (hist) hist->push(std::sqrt(absres));
            prev = relres;
            if (std::sqrt(relres) < tol) break;
        }
        for (k = i; k >= 0; k--) {
            tmpx[k] = alp[k];
            for (j = 0; j < k; ++j) alp[j] -= h[k][j] * tmpx[k];
        }
        // dense_aAtxpby(n, i+1, z, 1.0, tmpx, rst == 0 ? 0.0 : 1.0, x): columns scaled in place,
        // summed into column 0 one after the other, x = 1.0 z_0 + beta x (the first cycle overwrites x)
        for (k = 0; k < i + 1; k++) d_scale(n, tmpx[k], Z(k));
        for (j = 1; j < i + 1; j++) d_axpy(n, 1.0, Z(j), Z(0));
        d_axpby(n, 1.0, Z(0), rst == 0 ? 0.0 : 1.0, x);
    }
    if (PrtLvl > PRINT_NONE) {
        if (iter > MaxIt) std::printf("### WARNING: MaxIt = %d reached with relative residual %.10e.\n", MaxIt, std::sqrt(relres));
        else if (iter >= 0) std::printf("Number of iterations = %d with relative residual %.10e.\n", iter, std::sqrt(relres));
    }
    if (out) { out->relres = std::sqrt(relres); out->absres = std::sqrt(absres); out->normr0 = std::sqrt(absres0); }
    HIPCK(hipStreamSynchronize(V.s));
    return iter >= MaxIt ? ERROR_SOLVER_MAXIT : iter;
}

// ---------------------------------------------------------------------------
// Matrix-free family (SolMatFree.c): the reference keeps older texts of CG and of the GMRES
// variants for the mxv_matfree interface; they are restated separately (oracle: pcg_mf_core,
// gmres_mf_core).  BiCGstab and GCG perform the arithmetic of their CSR texts.
// ---------------------------------------------------------------------------
// fasp_solver_pcg, KryPcg.c:1260-1540
static int pcg_mf_device(KOps& K, const double* b, double* u, double tol, double abstol, int MaxIt, int StopType,
                         int PrtLvl, PcgOut* out)
{
    KVecOps V(K);
    const int m = V.m;
    const double maxdiff = tol * STAG_RATIO, sol_inf_tol = SMALLREAL;
    int iter = 0, stag = 1, more_step = 1;
    double absres0 = BIGREAL, absres = BIGREAL, relres = BIGREAL, normu = BIGREAL, normr0 = BIGREAL;
    double reldiff, factor, alpha, beta, temp1 = 0.0, temp2, red[8], pp;
    KCK(V.ensure(4));
    double *p = V.vec(0), *z = V.vec(1), *r = V.vec(2), *t = V.vec(3);
    auto rel_from = [&]() -> int {  // relres per stop type from the current r (absres is ||r||_2 throughout)
        switch (StopType) {
            case STOP_REL_PRECRES:
                KCK(V.pc(r, z)); KCK(V.dot(z, r, temp2));
                relres = std::sqrt(std::fabs(temp2)) / normr0; break;
            case STOP_MOD_REL_RES: relres = absres / normu; break;
            default: relres = absres / normr0; break;
        }
        return 0;
    };
    if (PrtLvl > PRINT_NONE) std::printf("\nCalling CG solver (MatFree) ...\n");
    KCK(V.resid(u, b, r));
    KCK(V.pc(r, z));
    switch (StopType) {
        case STOP_REL_PRECRES:
            KCK(V.dot(r, z, temp2)); absres0 = std::sqrt(temp2); normr0 = std::max(SMALLREAL, absres0); relres = absres0 / normr0; break;
        case STOP_MOD_REL_RES:
            KCK(V.nrm2(r, absres0)); KCK(V.nrm2(u, normu)); normu = std::max(SMALLREAL, normu); relres = absres0 / normu; break;
        default:
            KCK(V.nrm2(r, absres0)); normr0 = std::max(SMALLREAL, absres0); relres = absres0 / normr0; break;
    }
    if (relres < tol || absres0 < abstol) goto FINISHED;
    KCK(V.cp(p, z));
    KCK(V.dot(z, r, temp1));
    while (iter++ < MaxIt) {
        KCK(V.mxv(p, t));
        KCK(V.dot(t, p, temp2));
        alpha = temp1 / temp2;
        d_axpy(m, alpha, p, u);
        d_axpy(m, -alpha, t, r);
        KCK(V.nrm2(r, absres));
        factor = absres / absres0;
        KCK(rel_from());
        itinfo(PrtLvl, StopType, iter, relres, absres, factor);
        if (d_norms(m, u, red, K.dist) < 0) return ERROR_MISC;
        if (red[1] <= sol_inf_tol) {
            if (PrtLvl > PRINT_MIN) std::printf("### WARNING: Iteration stopped -- solution almost zero! [%s:%d]\n", "fasp_solver_pcg", 1390);
            iter = ERROR_SOLVER_SOLSTAG;
            break;
        }
        normu = std::sqrt(red[0]);
        KCK(V.nrm2(p, pp));
        reldiff = std::fabs(alpha) * pp / normu;
        if ((stag <= MAX_STAG) & (reldiff < maxdiff)) {
            if (PrtLvl >= PRINT_MORE) {
                std::printf("||u-u'|| = %.10e and the comp. rel. res. = %.10e.\n", reldiff, relres);
                std::printf("### WARNING: Iteration restarted -- stagnation! [%s:%d]\n", "fasp_solver_pcg", 1404);
            }
            KCK(V.resid(u, b, r));
            KCK(V.nrm2(r, absres));
            KCK(rel_from());
            if (PrtLvl >= PRINT_MORE) std::printf("### WARNING: The actual relative residual = %.10e!\n", relres);
            if (relres < tol) break;
            if (stag >= MAX_STAG) {
                if (PrtLvl > PRINT_MIN) std::printf("### WARNING: Iteration stopped -- staggnation! [%s:%d]\n", "fasp_solver_pcg", 1437);
                iter = ERROR_SOLVER_STAG;
                break;
            }
            KCK(V.zero(p));
            ++stag;
        }
        if (relres < tol) {
            if (PrtLvl >= PRINT_MORE) std::printf("### WARNING: The computed relative residual = %.10e!\n", relres);
            KCK(V.resid(u, b, r));
            if (StopType != STOP_REL_PRECRES) KCK(V.nrm2(r, absres));
            KCK(rel_from());
            if (PrtLvl >= PRINT_MORE) std::printf("### WARNING: The actual relative residual = %.10e!\n", relres);
            if (relres < tol) break;
            if (more_step >= MAX_RESTART) {
                if (PrtLvl > PRINT_MIN) std::printf("### WARNING: The tolerence might be too small! [%s:%d]\n", "fasp_solver_pcg", 1487);
                iter = ERROR_SOLVER_TOLSMALL;
                break;
            }
            KCK(V.zero(p));
            ++more_step;
        }
        absres0 = absres;
        if (StopType != STOP_REL_PRECRES) KCK(V.pc(r, z));
        KCK(V.dot(z, r, temp2));
        beta = temp2 / temp1;
        temp1 = temp2;
        d_axpby(m, 1.0, z, beta, p);
    }
FINISHED:
    if (PrtLvl > PRINT_NONE) {
        if (iter > MaxIt) std::printf("### WARNING: MaxIt = %d reached with relative residual %.10e.\n", MaxIt, relres);
        else if (iter >= 0) std::printf("Number of iterations = %d with relative residual %.10e.\n", iter, relres);
    }
    if (out) { out->relres = relres; out->absres = absres; out->normr0 = normr0; }
    HIPCK(hipStreamSynchronize(V.s));
    return iter > MaxIt ? ERROR_SOLVER_MAXIT : iter;
}

// fasp_solver_pgmres / _pvgmres / _pvfgmres for mxv_matfree (KryPgmres.c:1309, KryPvgmres.c:1468,
// KryPvfgmres.c:1026): one text with two switches; stops on ||r|| <= tol ||b||, StopType ignored.
static int gmres_mf_device(KOps& K, bool variable, bool flexible, const double* b, double* x, double tol, int MaxIt,
                           int restart, int StopType, int PrtLvl, PcgOut* out)
{
    KVecOps V(K);
    const int n = V.m, min_iter = 0;
    const double cr_max = 0.99, cr_min = 0.174, epsmac = SMALLREAL;
    int iter = 0, i, j, k;
    double r_norm, b_norm, den_norm, epsilon, gamma, t, cr = 1.0, r_norm_old = 0.0, prev;
    const int d = 3, restart_max = restart, restart_min = 3;
    int Restart = restart;
    const int Restart1 = restart + 1;
    if (restart < 1) return ERROR_INPUT_PAR;
    KCK(V.ensure(2 + (size_t)Restart1 * (flexible ? 2 : 1)));
    double *r = V.vec(0), *w = V.vec(1);
    auto P = [&](int q) { return V.vec(2 + (size_t)q); };
    auto Z = [&](int q) { return V.vec(2 + (size_t)Restart1 + (size_t)q); };
    std::vector<double> rs((size_t)Restart1 + 1), c((size_t)Restart1), sn((size_t)Restart1);
    std::vector<std::vector<double>> hh((size_t)Restart1, std::vector<double>((size_t)restart + 1, 0.0));
    if (PrtLvl > PRINT_NONE)
        std::printf(flexible ? "\nCalling VFGMRes solver (MatFree) ...\n" : variable ? "\nCalling VGMRes solver (MatFree) ...\n"
                                                                                    : "\nCalling GMRes solver (MatFree) ...\n");
    KCK(V.resid(x, b, P(0)));
    KCK(V.nrm2(b, b_norm));
    KCK(V.nrm2(P(0), r_norm));
    prev = r_norm;
    if (PrtLvl >= PRINT_SOME) {
        std::printf("L2 norm of %s = %.10e.\n", "right-hand side", b_norm);
        std::printf("L2 norm of %s = %.10e.\n", "residual", r_norm);
    }
    den_norm = (b_norm > 0.0) ? b_norm : r_norm;
    epsilon = tol * den_norm;
    while (iter < MaxIt) {
        rs[0] = r_norm;
        r_norm_old = r_norm;
        if (r_norm == 0.0) {
            if (out) { out->relres = 0.0; out->absres = 0.0; out->normr0 = den_norm; }
            HIPCK(hipStreamSynchronize(V.s));
            return iter;
        }
        if (variable) {
            if (cr > cr_max || iter == 0) Restart = restart_max;
            else if (cr < cr_min) { /* keep */ }
            else { if (Restart - d > restart_min) Restart -= d; else Restart = restart_max; }
        }
        if (r_norm <= epsilon && iter >= min_iter) {
            KCK(V.resid(x, b, r));
            KCK(V.nrm2(r, r_norm));
            if (r_norm <= epsilon) break;
            if (PrtLvl >= PRINT_SOME) std::printf("### WARNING: False convergence! [%s:%d]\n", "fasp_solver_pvgmres", 1620);
        }
        d_scale(n, 1.0 / r_norm, P(0));
        i = 0;
        while (i < Restart && iter < MaxIt) {
            i++; iter++;
            if (flexible) { KCK(V.pc(P(i - 1), Z(i - 1))); KCK(V.mxv(Z(i - 1), P(i))); }
            else          { KCK(V.pc(P(i - 1), r));        KCK(V.mxv(r, P(i))); }
            for (j = 0; j < i; j++) {  // modified Gram-Schmidt
                KCK(V.dot(P(j), P(i), hh[j][i - 1]));
                d_axpy(n, -hh[j][i - 1], P(j), P(i));
            }
            KCK(V.nrm2(P(i), t));
            hh[i][i - 1] = t;
            if (t != 0.0) d_scale(n, 1.0 / t, P(i));
            for (j = 1; j < i; ++j) {
                t = hh[j - 1][i - 1];
                hh[j - 1][i - 1] = sn[j - 1] * hh[j][i - 1] + c[j - 1] * t;
                hh[j][i - 1] = -sn[j - 1] * t + c[j - 1] * hh[j][i - 1];
            }
            t = hh[i][i - 1] * hh[i][i - 1];
            t += hh[i - 1][i - 1] * hh[i - 1][i - 1];
            gamma = std::sqrt(t);
            if (gamma == 0.0) gamma = epsmac;
            c[i - 1] = hh[i - 1][i - 1] / gamma;
            sn[i - 1] = hh[i][i - 1] / gamma;
            rs[i] = -sn[i - 1] * rs[i - 1];
            rs[i - 1] = c[i - 1] * rs[i - 1];
            hh[i - 1][i - 1] = sn[i - 1] * hh[i][i - 1] + c[i - 1] * hh[i - 1][i - 1];
            r_norm = std::fabs(rs[i]);
            if (b_norm > 0) itinfo(PrtLvl, StopType, iter, r_norm / b_norm, r_norm, r_norm / prev);
            else itinfo(PrtLvl, StopType, iter, r_norm, r_norm, r_norm / prev);
            prev = r_norm;
            if (r_norm <= epsilon && iter >= min_iter) break;
        }
        rs[i - 1] = rs[i - 1] / hh[i - 1][i - 1];
        for (k = i - 2; k >= 0; k--) {
            t = 0.0;
            for (j = k + 1; j < i; j++) t -= hh[k][j] * rs[j];
            t += rs[k];
            rs[k] = t / hh[k][k];
        }
        if (flexible) {
            KCK(V.cp(r, Z(i - 1)));
            d_scale(n, rs[i - 1], r);
            for (j = i - 2; j >= 0; j--) d_axpy(n, rs[j], Z(j), r);
        } else {
            KCK(V.cp(w, P(i - 1)));
            d_scale(n, rs[i - 1], w);
            for (j = i - 2; j >= 0; j--) d_axpy(n, rs[j], P(j), w);
            KCK(V.pc(w, r));
        }
        d_axpy(n, 1.0, r, x);
        if (r_norm <= epsilon && iter >= min_iter) {
            KCK(V.resid(x, b, r));
            KCK(V.nrm2(r, r_norm));
            if (r_norm <= epsilon) break;
            if (PrtLvl >= PRINT_SOME) std::printf("### WARNING: False convergence! [%s:%d]\n", "fasp_solver_pvgmres", 1757);
            KCK(V.cp(P(0), r));
            i = 0;
        }
        for (j = i; j > 0; j--) {
            rs[j - 1] = -sn[j - 1] * rs[j];
            rs[j] = c[j - 1] * rs[j];
        }
        // p[i] += (rs[i] - 1) p[i] is evaluated elementwise as y + a y (BlaArray.c:90), not as a scaling
        if (i) hipLaunchKernelGGL(k_axpy_self, dim3(vec_grid(n)), dim3(BLOCK), 0, V.s, n, rs[i] - 1.0, P(i));
        for (j = i - 1; j > 0; j--) d_axpy(n, rs[j], P(j), P(i));
        if (i) {
            hipLaunchKernelGGL(k_axpy_self, dim3(vec_grid(n)), dim3(BLOCK), 0, V.s, n, rs[0] - 1.0, P(0));
            d_axpy(n, 1.0, P(i), P(0));
        }
        if (variable) cr = r_norm / r_norm_old;
    }
    if (PrtLvl > PRINT_NONE) {
        if (iter > MaxIt) std::printf("### WARNING: MaxIt = %d reached with relative residual %.10e.\n", MaxIt, r_norm);
        else if (iter >= 0) std::printf("Number of iterations = %d with relative residual %.10e.\n", iter, r_norm);
    }
    if (out) { out->relres = r_norm / den_norm; out->absres = r_norm; out->normr0 = den_norm; }
    HIPCK(hipStreamSynchronize(V.s));
    return iter >= MaxIt ? ERROR_SOLVER_MAXIT : iter;
}

// ---------------------------------------------------------------------------
// one multigrid cycle on the resident hierarchy (PreMGCycle.c:48-274)
// ---------------------------------------------------------------------------
// fasp_coarse_itsolver (PreMGUtil.inl:37): safe CG on the coarsest level, SPVGMRES as the safety net
static int coarse_solve(fasp_hip_amg* h, const AMG_param& param, double tol)
{
    const int nl = (int)h->L.size();
    DevLevel& Lc = h->L[nl - 1];
    int st = coarse_spcg(h, Lc, tol, param.print_level);
    if (st == ERROR_MISC) return st;  // device failure, not a solver verdict
    if (st < 0) {
        // safety net of PreMGUtil.inl:50-52: fasp_solver_dcsr_spvgmres(A, b, x, NULL, ctol, maxit, 20, 1, ..)
        const int m = Lc.A.row;
        const int nn = (int)((unsigned)m * (unsigned)m);
        const int maxit = std::max(250, std::min(nn, 1000));
        KOps Kc = csr_ops(h, nl - 1, false);
        st = gmres_device(Kc, Lc.b, Lc.x, 2, tol, 0.0, maxit, 20, STOP_REL_RES, param.print_level - 4,
                          nullptr, nullptr);
        if (st == ERROR_MISC) return st;
        if (st < 0 && param.print_level >= PRINT_MORE) {
            std::printf("### WARNING: Coarse level solver did not converge!\n");
            std::printf("### WARNING: Consider to increase maxit to %d!\n", 2 * maxit);
        }
    }
    return FASP_SUCCESS;
}

// fasp_amg_amli_coef (PreMGRecurAMLI.c:791): coefficients of the degree-`degree` polynomial that
// approximates 1/t on [lambda_min, lambda_max]
static void amli_coef(double lambda_max, double lambda_min, int degree, double* coef)
{
    const double mu0 = 1.0 / lambda_max, mu1 = 1.0 / lambda_min;
    const double c = (std::sqrt(mu0) + std::sqrt(mu1)) * (std::sqrt(mu0) + std::sqrt(mu1));
    const double a = (4 * mu0 * mu1) / (c);
    const double kappa = lambda_max / lambda_min;
    const double delta = (std::sqrt(kappa) - 1.0) / (std::sqrt(kappa) + 1.0);
    const double b = delta * delta;
    if (degree == 0) coef[0] = 0.5 * (mu0 + mu1);
    else if (degree == 1) { coef[0] = 0.5 * c; coef[1] = -1.0 * mu0 * mu1; }
    else if (degree > 1) {
        std::vector<double> work((size_t)2 * degree - 1, 0.0);
        double *coef_k = work.data(), *coef_km1 = work.data() + degree;
        amli_coef(lambda_max, lambda_min, degree - 1, coef_k);
        amli_coef(lambda_max, lambda_min, degree - 2, coef_km1);
        coef[0] = a - b * coef_km1[0] + (1 + b) * coef_k[0];
        for (int i = 1; i < degree - 1; i++) coef[i] = -b * coef_km1[i] + (1 + b) * coef_k[i] - a * coef_k[i - 1];
        coef[degree - 1] = (1 + b) * coef_k[degree - 1] - a * coef_k[degree - 2];
        coef[degree] = -a * coef_k[degree - 1];
    }
}

// fasp_solver_amli (PreMGRecurAMLI.c:58): the coarse-grid correction of every level is a polynomial of
// degree amli_degree in the recursively preconditioned coarse operator (coefficients for the interval
// [0.5, 2], PreAMGSetupRS.c:93-97).  One GPU (AMLI hierarchies are not row-partitioned).
static int amli_cycle(fasp_hip_amg* h, const AMG_param& param, int l)
{
    const int nl = (int)h->L.size(), degree = param.amli_degree;
    hipStream_t s = g_ctx.stream;
    DevLevel& D = h->L[l];
    int st;
    if (l >= nl - 1) return coarse_solve(h, param, param.tol * 1e-4);
    DevLevel& C = h->L[l + 1];
    const int m0 = D.A.row, m1 = C.A.row;
    const double* coef = h->amli_coef.data();
    if (!C.w2) { if (alloc_vec(&C.w2, (size_t)C.nvec) < 0) return ERROR_ALLOC_MEM; }
    double* r1 = C.w2;
    if ((st = smooth(h, l, false, param.smoother, param.smooth_order, param.presmooth_iter, param.relaxation, param.polynomial_degree)) < 0) return st;
    if (D.x_zero) HIPCK(hipMemcpyAsync(D.w, D.b, sizeof(double) * m0, hipMemcpyDeviceToDevice, s));
    else d_resid(D.A, D.x, D.b, D.w);
    d_mxv(D.R, D.w, C.b);
    HIPCK(hipMemcpyAsync(r1, C.b, sizeof(double) * m1, hipMemcpyDeviceToDevice, s));
    for (int i = 1; i <= degree; i++) {
        C.x_zero = true;
        if ((st = amli_cycle(h, param, l + 1)) < 0) return st;
        materialise_zero(C);
        d_mxv(C.A, C.x, C.b);                                           // b1 = A1 e1
        d_axpy(m1, coef[degree - i] / coef[degree], r1, C.b);           // b1 += (q_{degree-i} / q_degree) r1
    }
    C.x_zero = true;
    if ((st = amli_cycle(h, param, l + 1)) < 0) return st;
    materialise_zero(C);
    d_scale(m1, coef[degree], C.x);
    double alpha = 1.0;
    if (param.coarse_scaling == 1) {  // alpha = (e1, r1) / (A1 e1, e1), capped at 1; C.w is free scratch here
        double red[2];
        CsrArgs a{}; a.x = C.x; a.y = C.w; a.dotv = C.x; a.partials = g_ctx.d_partials;
        const int gdot = launch_csr<OP_MXV_DOT>(C.A, a);
        d_finalize(gdot, 1, 0u, 1, false);
        if (fetch_red(1, 1, red + 1) < 0) return ERROR_MISC;
        if (d_dot(m1, C.x, r1, red, false) < 0) return ERROR_MISC;
        alpha = std::min(red[0] / red[1], 1.0);
    }
    materialise_zero(D);
    d_aAxpy(alpha, D.P, C.x, D.x);
    return smooth(h, l, true, param.smoother, param.smooth_order, param.postsmooth_iter, param.relaxation, param.polynomial_degree);
}

// Nonlinear AMLI / K-cycle (fasp_solver_namli, PreMGRecurAMLI.c:291; Kcycle_dcsr_pgcg / _pgcr,
// PreMGRecurAMLI.inl:36 / :139; fasp_precond_namli, PreCSR.c:524): the coarse problem of a level whose
// AMG_data.cycle_type is > 1 is solved by at most two steps of a Krylov method preconditioned by the
// same cycle one level down.  `base` = the level the reference's shifted pointer &mgl[l+1] points at.
static int namli_cycle(fasp_hip_amg* h, const AMG_param& param, int base, int num_levels);

static int namli_precond(fasp_hip_amg* h, const AMG_param& user, int base, int num_levels, const double* r, double* z)
{
    AMG_param p;  // fasp_param_amg_init + fasp_param_prec_to_amg (AuxParam.c:816): tol is not carried over
    fasp_param_amg_init(&p);
    p.AMG_type = user.AMG_type; p.print_level = user.print_level; p.cycle_type = user.cycle_type;
    p.smoother = user.smoother; p.smooth_order = user.smooth_order; p.presmooth_iter = user.presmooth_iter;
    p.postsmooth_iter = user.postsmooth_iter; p.relaxation = user.relaxation;
    p.polynomial_degree = user.polynomial_degree; p.coarse_solver = user.coarse_solver;
    p.coarse_scaling = user.coarse_scaling; p.amli_degree = user.amli_degree;
    p.nl_amli_krylov_type = user.nl_amli_krylov_type; p.tentative_smooth = user.tentative_smooth;
    DevLevel& L = h->L[base];
    const int m = L.A.row;
    HIPCK(hipMemcpyAsync(L.b, r, sizeof(double) * m, hipMemcpyDeviceToDevice, g_ctx.stream));
    L.x_zero = true;
    const int st = namli_cycle(h, p, base, num_levels);
    if (st < 0) return st;
    materialise_zero(L);
    HIPCK(hipMemcpyAsync(z, L.x, sizeof(double) * m, hipMemcpyDeviceToDevice, g_ctx.stream));
    return FASP_SUCCESS;
}

// at most two steps of GCG (gcr == false) or GCR on level `base` (matrix L.A, right-hand side L.b), result in x
static int kcycle(fasp_hip_amg* h, const AMG_param& param, bool gcr, int base, int num_levels, double* x)
{
    DevLevel& L = h->L[base];
    const int m = L.A.row;
    hipStream_t s = g_ctx.stream;
    for (double*& q : L.kw) if (!q) { if (alloc_vec(&q, (size_t)L.nvec) < 0) return ERROR_ALLOC_MEM; }
    double *r = L.kw[0], *x1 = L.kw[1], *v1 = L.kw[2], *v2 = L.kw[3];
    double red[2], normb, absres, relres, alpha1, alpha2, gamma, rho1, rho2;
    auto dot = [&](const double* a, const double* b, double& v) -> int {
        if (d_dot(m, a, b, red, false) < 0) return ERROR_MISC;
        v = red[0];
        return 0;
    };
    int st;
    if ((st = dot(L.b, L.b, normb)) < 0) return st;
    normb = std::sqrt(normb);
    HIPCK(hipMemcpyAsync(r, L.b, sizeof(double) * m, hipMemcpyDeviceToDevice, s));
    if ((st = namli_precond(h, param, base, num_levels, r, x)) < 0) return st;
    d_mxv(L.A, x, v1);
    if (!gcr) {
        if ((st = dot(x, v1, rho1)) < 0 || (st = dot(x, r, alpha1)) < 0) return st;
        const double beta1 = alpha1 / rho1;
        d_axpy(m, -beta1, v1, r);
        if ((st = dot(r, r, absres)) < 0) return st;
        relres = std::sqrt(absres) / normb;
        if (relres < 0.2) { d_scale(m, beta1, x); return FASP_SUCCESS; }
        if ((st = namli_precond(h, param, base, num_levels, r, x1)) < 0) return st;
        d_mxv(L.A, x1, v2);
        if ((st = dot(x1, v1, gamma)) < 0 || (st = dot(x1, r, alpha2)) < 0 || (st = dot(x1, v2, rho2)) < 0) return st;
        const double beta2 = rho2 - gamma * gamma / rho1;
        if (std::fabs(beta2) < SMALLREAL) return FASP_SUCCESS;
        const double beta3 = (alpha1 - gamma * alpha2 / beta2) / rho1, beta4 = alpha2 / beta2;
        d_scale(m, beta3, x);
        d_axpy(m, beta4, x1, x);
    } else {
        double beta;
        if ((st = dot(v1, v1, rho1)) < 0 || (st = dot(v1, r, alpha1)) < 0) return st;
        const double alpha = alpha1 / rho1;
        d_axpy(m, -alpha, v1, r);
        if ((st = dot(r, r, absres)) < 0) return st;
        relres = std::sqrt(absres) / normb;
        if (relres < 0.2) { d_scale(m, alpha, x); return FASP_SUCCESS; }
        if ((st = namli_precond(h, param, base, num_levels, r, x1)) < 0) return st;
        d_mxv(L.A, x1, v2);
        if ((st = dot(v1, v2, gamma)) < 0 || (st = dot(v2, v2, beta)) < 0 || (st = dot(r, v2, alpha2)) < 0) return st;
        rho2 = beta - gamma * gamma / rho1;
        const double alpha3 = alpha1 / rho1 - gamma * alpha2 / (rho1 * rho2), alpha4 = alpha2 / rho2;
        d_scale(m, alpha3, x);
        d_axpy(m, alpha4, x1, x);
    }
    return FASP_SUCCESS;
}

static int namli_cycle(fasp_hip_amg* h, const AMG_param& param, int base, int num_levels)
{
    hipStream_t s = g_ctx.stream;
    DevLevel& D = h->L[base];
    int st;
    if (num_levels <= 1) {  // coarsest level of this sub-hierarchy == coarsest level of the hierarchy
        if (base != (int)h->L.size() - 1) return ERROR_INPUT_PAR;
        return coarse_solve(h, param, param.tol * 1e-4);
    }
    DevLevel& C = h->L[base + 1];
    const int m0 = D.A.row, m1 = C.A.row;
    if ((st = smooth(h, base, false, param.smoother, param.smooth_order, param.presmooth_iter, param.relaxation, param.polynomial_degree)) < 0) return st;
    if (D.x_zero) HIPCK(hipMemcpyAsync(D.w, D.b, sizeof(double) * m0, hipMemcpyDeviceToDevice, s));
    else d_resid(D.A, D.x, D.b, D.w);
    d_mxv(D.R, D.w, C.b);
    const int ct = (base + 1 < (int)h->level_cycle_type.size()) ? h->level_cycle_type[(size_t)base + 1] : 1;
    if (ct <= 1) {  // a V-cycle is enforced on this level
        C.x_zero = true;
        if ((st = namli_cycle(h, param, base + 1, num_levels - 1)) < 0) return st;
        materialise_zero(C);
    } else {
        if (!C.w2) { if (alloc_vec(&C.w2, (size_t)C.nvec) < 0) return ERROR_ALLOC_MEM; }
        double* uH = C.w2;
        HIPCK(hipMemsetAsync(uH, 0, sizeof(double) * m1, s));
        if ((st = kcycle(h, param, param.nl_amli_krylov_type != SOLVER_GCG, base + 1, num_levels - 1, uH)) < 0) return st;
        HIPCK(hipMemcpyAsync(C.x, uH, sizeof(double) * m1, hipMemcpyDeviceToDevice, s));
        C.x_zero = false;
    }
    materialise_zero(D);
    d_aAxpy(1.0, D.P, C.x, D.x);
    return smooth(h, base, true, param.smoother, param.smooth_order, param.postsmooth_iter, param.relaxation, param.polynomial_degree);
}

// fasp_solver_fmgcycle (PreMGCycleFull.c:47): the right-hand side is restricted to every level, the
// coarsest system solved, then level by level the solution is interpolated and improved by up to 3
// V-cycles from that level.  As in the reference the iterate of an intermediate level is not reset
// before the interpolated correction is added: it keeps what the previous call left there.  One GPU.
static int fmg_cycle(fasp_hip_amg* h, const AMG_param& param)
{
    const int nl = (int)h->L.size(), maxit = 3;
    const double tol = param.tol * 1e-4;
    hipStream_t s = g_ctx.stream;
    int st, l;
    if (h->distributed) return ERROR_INPUT_PAR;
    h->vcycles++;
    for (l = 0; l < nl - 1; ++l) d_mxv(h->L[l].R, h->L[l].b, h->L[l + 1].b);
    h->L[l].x_zero = true;
    if (nl == 1) return coarse_solve(h, param, tol);
    auto scaled_prolongation = [&](int lf) -> int {  // x_lf += alpha P x_{lf+1}
        DevLevel& D = h->L[lf];
        DevLevel& C = h->L[lf + 1];
        double alpha = 1.0;
        materialise_zero(C);
        if (param.coarse_scaling == 1) {
            double red[2];
            CsrArgs a{}; a.x = C.x; a.y = C.w; a.dotv = C.x; a.partials = g_ctx.d_partials;
            const int gdot = launch_csr<OP_MXV_DOT>(C.A, a);
            d_finalize(gdot, 1, 0u, 1, false);
            if (fetch_red(1, 1, red + 1) < 0) return ERROR_MISC;
            if (d_dot(C.A.row, C.x, C.b, red, false) < 0) return ERROR_MISC;
            alpha = std::min(red[0] / red[1], 1.0);
        }
        materialise_zero(D);
        d_aAxpy(alpha, D.P, C.x, D.x);
        return FASP_SUCCESS;
    };
    for (int i = 1; i < nl; ++i) {
        if ((st = coarse_solve(h, param, tol)) < 0) return st;
        --l;
        if ((st = scaled_prolongation(l)) < 0) return st;
        int num_cycle = 0;
        double relerr = BIGREAL, red[2];
        while (relerr > param.tol && num_cycle < maxit) {
            ++num_cycle;
            {
                DevLevel& D = h->L[l];
                d_resid(D.A, D.x, D.b, D.w);
                double nw, nb;
                if (d_dot(D.A.row, D.w, D.w, red, false) < 0) return ERROR_MISC;
                nw = std::sqrt(red[0]);
                if (d_dot(D.A.row, D.b, D.b, red, false) < 0) return ERROR_MISC;
                nb = std::sqrt(red[0]);
                relerr = nw / nb;
            }
            for (int lvl = 0; lvl < i; ++lvl) {
                DevLevel& D = h->L[l];
                if ((st = smooth(h, l, false, param.smoother, param.smooth_order, param.presmooth_iter, param.relaxation, param.polynomial_degree)) < 0) return st;
                if (D.x_zero) HIPCK(hipMemcpyAsync(D.w, D.b, sizeof(double) * D.A.row, hipMemcpyDeviceToDevice, s));
                else d_resid(D.A, D.x, D.b, D.w);
                d_mxv(D.R, D.w, h->L[l + 1].b);
                ++l;
                h->L[l].x_zero = true;
            }
            if ((st = coarse_solve(h, param, tol)) < 0) return st;
            for (int lvl = 0; lvl < i; ++lvl) {
                --l;
                if ((st = scaled_prolongation(l)) < 0) return st;
                if ((st = smooth(h, l, true, param.smoother, param.smooth_order, param.postsmooth_iter, param.relaxation, param.polynomial_degree)) < 0) return st;
            }
        }
    }
    return FASP_SUCCESS;
}

static int mgcycle(fasp_hip_amg* h, const AMG_param& param)
{
    const int nl = (int)h->L.size();
    if (param.cycle_type == NL_AMLI_CYCLE) {  // fasp_precond_namli (PreCSR.c:524) / fasp_amg_solve_namli (PreMGSolve.c:230)
        if (h->distributed) return ERROR_INPUT_PAR;
        h->vcycles++;
        return namli_cycle(h, param, 0, nl);
    }
    if (param.cycle_type == AMLI_CYCLE) {  // fasp_precond_amli (PreCSR.c:482) / fasp_amg_solve_amli (PreMGSolve.c:142)
        if (h->distributed || param.amli_degree < 0 || param.amli_degree > 30) return ERROR_INPUT_PAR;
        if ((int)h->amli_coef.size() != param.amli_degree + 1) {
            h->amli_coef.assign((size_t)param.amli_degree + 1, 0.0);
            amli_coef(2.0, 0.5, param.amli_degree, h->amli_coef.data());
        }
        h->vcycles++;
        return amli_cycle(h, param, 0);
    }
    const int smoother = param.smoother, cycle_type = param.cycle_type;
    const double relax = param.relaxation;
    const double tol = param.tol * 1e-4;
    int num_lvl[MAX_AMG_LVL] = {0}, ncycles[MAX_AMG_LVL], l = 0;
    for (int i = 0; i < MAX_AMG_LVL; ++i) ncycles[i] = 1;
    switch (cycle_type) {
        case 12: for (int i = MAX_AMG_LVL - 2; i > 0; i -= 2) ncycles[i] = 2; break;
        case 21: for (int i = MAX_AMG_LVL - 1; i > 0; i -= 2) ncycles[i] = 2; break;
        default: for (int i = 0; i < MAX_AMG_LVL; ++i) ncycles[i] = cycle_type;
    }
    h->vcycles++;
    int st0 = FASP_SUCCESS;

ForwardSweep:
    while (l < nl - 1) {
        DevLevel& D = h->L[l];
        num_lvl[l]++;
        if ((st0 = smooth(h, l, false, smoother, param.smooth_order, param.presmooth_iter, relax, param.polynomial_degree)) < 0) return st0;
        // w = b - A x ; b_{l+1} = R w
        if (D.x_zero) {
            HIPCK(hipMemcpyAsync(D.w, D.b, sizeof(double) * D.A.row, hipMemcpyDeviceToDevice, g_ctx.stream));
        } else {
            if (halo_exchange(D, D.x) < 0) return ERROR_MISC;
            d_resid(D.A, D.x, D.b, D.w);
        }
        if (halo_exchange(D, D.w) < 0) return ERROR_MISC;
        {
            DevLevel& C = h->L[l + 1];
            if (!D.replicated && C.replicated) {
                // first replicated level: every rank restricts onto the coarse rows it owns,
                // one all-gather assembles the whole right-hand side on every rank
                const std::vector<int>& cs = h->dist.L[l + 1].start;
                std::vector<int> counts(comm_size());
                for (int q = 0; q < comm_size(); ++q) counts[q] = cs[q + 1] - cs[q];
                d_mxv(D.R, D.w, C.b + cs[comm_rank()]);
                if (comm_allgatherv(C.b + cs[comm_rank()], counts[comm_rank()], C.b, counts.data(), cs.data(),
                                    g_ctx.stream) < 0) return ERROR_MISC;
            } else {
                d_mxv(D.R, D.w, C.b);
            }
        }
        ++l;
        h->L[l].x_zero = true;  // fasp_dvec_set(x_{l}, 0): materialised lazily
    }

    if ((st0 = coarse_solve(h, param, tol)) < 0) return st0;

    while (l > 0) {
        --l;
        DevLevel& D = h->L[l];
        materialise_zero(D);
        DevLevel& C = h->L[l + 1];
        if (halo_exchange(C, C.x) < 0) return ERROR_MISC;
        double alpha = 1.0;
        if (param.coarse_scaling == 1) {
            // PreMGCycle.c:210-216: alpha = (x_c, b_c) / (A_c x_c, x_c), capped at 1
            // (fasp_blas_dcsr_vmv, BlaSpmvCSR.c:839); C.w is free scratch on the way up
            const bool cdist = !C.replicated && comm_size() > 1;
            double red[2];
            CsrArgs a{}; a.x = C.x; a.y = C.w; a.dotv = C.x; a.partials = g_ctx.d_partials;
            const int gdot = launch_csr<OP_MXV_DOT>(C.A, a);
            d_finalize(gdot, 1, 0u, 1, cdist);
            if (fetch_red(1, 1, red + 1) < 0) return ERROR_MISC;
            if (d_dot(C.A.row, C.x, C.b, red, cdist) < 0) return ERROR_MISC;
            alpha = std::min(red[0] / red[1], 1.0);
        }
        d_aAxpy(alpha, D.P, C.x, D.x);  // x_l += alpha P x_{l+1}
        if ((st0 = smooth(h, l, true, smoother, param.smooth_order, param.postsmooth_iter, relax, param.polynomial_degree)) < 0) return st0;
        if (num_lvl[l] < ncycles[l]) break;
        else num_lvl[l] = 0;
    }
    if (l > 0) goto ForwardSweep;
    return FASP_SUCCESS;
}

// z = B r  (PreCSR.c:416-435).  The AMG_param used by the cycle is re-initialised and
// only the fields of fasp_param_prec_to_amg (AuxParam.c:816-834) are carried over: tol
// stays 1e-6.  r is used in place as the level-0 rhs and the result is left in the
// level-0 iterate; *z receives that pointer (both copies of the reference are elided).
static int precond_amg(fasp_hip_amg* h, double* r, double** z)
{
    AMG_param p;
    fasp_param_amg_init(&p);
    const AMG_param& u = h->param;
    p.AMG_type = u.AMG_type; p.print_level = u.print_level; p.cycle_type = u.cycle_type;
    p.smoother = u.smoother; p.smooth_order = u.smooth_order; p.presmooth_iter = u.presmooth_iter;
    p.postsmooth_iter = u.postsmooth_iter; p.relaxation = u.relaxation;
    p.polynomial_degree = u.polynomial_degree; p.coarse_solver = u.coarse_solver;
    p.coarse_scaling = u.coarse_scaling; p.tentative_smooth = u.tentative_smooth;
    p.amli_degree = u.amli_degree; p.nl_amli_krylov_type = u.nl_amli_krylov_type;
    DevLevel& D0 = h->L[0];
    D0.b = r;
    D0.x_zero = true;
    for (int i = u.maxit; i--;) {
        const int st = h->use_fmg ? fmg_cycle(h, p) : mgcycle(h, p);  // fasp_precond_famg (PreCSR.c:560) / fasp_precond_amg
        if (st < 0) return st;
    }
    materialise_zero(D0);
    *z = D0.x;
    return FASP_SUCCESS;
}

// fasp_amg_solve (PreMGSolve.c:49): multigrid cycles as a stand-alone iteration on the resident
// vectors h->b, h->u.  The cycle receives the caller's AMG_param (coarse tolerance tol * 1e-4).
static int amg_solve_device(fasp_hip_amg* h, const AMG_param& param, Hist& hist, PcgOut& out)
{
    DevLevel& D0 = h->L[0];
    const int m = D0.A.row, MaxIt = param.maxit, prtlvl = param.print_level;
    const bool dist = h->distributed;
    const double tol = param.tol;
    hipStream_t s = g_ctx.stream;
    double red[2], relres1 = 1.0, absres0, absres = 0.0;
    int iter = 0, st;
    if (d_dot(m, h->b, h->b, red, dist) < 0) return ERROR_MISC;
    const double sumb = std::sqrt(red[0]);
    absres0 = sumb;
    D0.b = h->b;
    HIPCK(hipMemcpyAsync(D0.x, h->u, sizeof(double) * m, hipMemcpyDeviceToDevice, s));
    D0.x_zero = false;
    itinfo(prtlvl, STOP_REL_RES, iter, relres1, sumb, 0.0);
    hist.push(sumb);
    if (sumb <= SMALLREAL) HIPCK(hipMemsetAsync(D0.x, 0, sizeof(double) * m, s));
    while ((iter++ < MaxIt) & (sumb > SMALLREAL)) {
        if ((st = mgcycle(h, param)) < 0) return st;
        materialise_zero(D0);
        if (halo_exchange(D0, D0.x) < 0) return ERROR_MISC;
        d_resid(D0.A, D0.x, D0.b, D0.w);
        if (d_dot(m, D0.w, D0.w, red, dist) < 0) return ERROR_MISC;
        absres = std::sqrt(red[0]);
        relres1 = absres / std::max(SMALLREAL, sumb);
        const double factor = absres / absres0;
        absres0 = absres;
        itinfo(prtlvl, STOP_REL_RES, iter, relres1, absres, factor);
        hist.push(absres);
        if (relres1 < tol) break;
    }
    HIPCK(hipMemcpyAsync(h->u, D0.x, sizeof(double) * m, hipMemcpyDeviceToDevice, s));
    HIPCK(hipStreamSynchronize(s));
    if (prtlvl > PRINT_NONE) {
        if (iter > MaxIt) std::printf("### WARNING: MaxIt = %d reached with relative residual %.10e.\n", MaxIt, relres1);
        else if (iter >= 0) std::printf("Number of iterations = %d with relative residual %.10e.\n", iter, relres1);
    }
    out.relres = relres1; out.absres = absres; out.normr0 = sumb;
    return iter > MaxIt ? ERROR_SOLVER_MAXIT : iter;
}

// Krylov operator bundles of the CSR hierarchy: level 0 (with the AMG preconditioner) and the
// coarsest level (no preconditioner: the SPVGMRES safety net)
static KOps csr_ops(fasp_hip_amg* h, int level, bool with_pc)
{
    KOps K;
    DevLevel* Lv = &h->L[level];
    K.n = Lv->A.row; K.nvec = (size_t)Lv->nvec; K.fmt = "CSR";
    K.dist = (level == 0) && h->distributed;
    K.halo = [Lv](double* v) { return halo_exchange(*Lv, v); };
    K.mxv = [Lv](const double* x, double* y) { d_mxv(Lv->A, x, y); };
    K.resid = [Lv](const double* x, const double* b, double* r) { d_resid(Lv->A, x, b, r); };
    K.mxv_dot = [Lv](const double* x, double* y) {
        CsrArgs a{}; a.x = x; a.y = y; a.dotv = x; a.partials = g_ctx.d_partials;
        return launch_csr<OP_MXV_DOT>(Lv->A, a);
    };
    if (with_pc) K.pc = [h](double* in, double** out) { return precond_amg(h, in, out); };
    const int set = level == 0 ? 0 : 1;
    K.ws = &h->gm[set]; K.ws_len = &h->gm_len[set]; K.hh = &h->gm_hh;
    K.stats = h;
    return K;
}

// ---------------------------------------------------------------------------
// preconditioned CG (KryPcg.c:96-362) on device vectors
// ---------------------------------------------------------------------------
static void itinfo(int ptrlvl, int stop_type, int iter, double relres, double absres, double factor)
{  // AuxMessage.c:41-71
    if (ptrlvl < PRINT_SOME) return;
    if (iter > 0) {
        std::printf("%6d | %13.6e   | %13.6e  | %10.4f\n", iter, relres, absres, factor);
    } else {
        std::printf("-----------------------------------------------------------\n");
        switch (stop_type) {
            case STOP_REL_RES: std::printf("It Num |   ||r||/||b||   |     ||r||      |  Conv. Factor\n"); break;
            case STOP_REL_PRECRES: std::printf("It Num | ||r||_B/||b||_B |    ||r||_B     |  Conv. Factor\n"); break;
            case STOP_MOD_REL_RES: std::printf("It Num |   ||r||/||x||   |     ||r||      |  Conv. Factor\n"); break;
        }
        std::printf("-----------------------------------------------------------\n");
        std::printf("%6d | %13.6e   | %13.6e  |     -.-- \n", iter, relres, absres);
    }
}

struct PcgVecs { const double* b; double *u, *p, *t, *r; };
static int pcg_device(KOps& K, const PcgVecs& V, double tol, double abstol, int MaxIt, int StopType,
                      int PrtLvl, Hist& hist, PcgOut& out)
{
    const int m = K.n;         // owned rows
    const bool dist = K.dist;  // reductions are all-reduced over the ranks
    fasp_hip_amg* h = K.stats;
    const double maxdiff = tol * STAG_RATIO, sol_inf_tol = SMALLREAL;
    int iter = 0, stag = 1, more_step = 1;
    double absres0 = BIGREAL, absres = BIGREAL, relres = BIGREAL, normu = BIGREAL, normr0 = BIGREAL;
    double reldiff, factor, alpha = 0.0, beta, temp1 = 0.0, temp2, red[8];
    double *p = V.p, *r = V.r, *t = V.t, *u = V.u, *z = nullptr;
    const double* b = V.b;
    hipStream_t s = g_ctx.stream;
    const int G = vec_grid(m);
    int st;

    auto apply_pc = [&]() -> int {
        if (K.pc) return K.pc(r, &z);
        z = r;
        return FASP_SUCCESS;
    };
    // residual norm per stop type (KryPcg.c:186-203 and the three copies below it)
    auto resnorm = [&](double rr_known, bool have_rr) -> int {
        switch (StopType) {
            case STOP_REL_RES:
                if (!have_rr) { if (d_dot(m, r, r, red, dist) < 0) return ERROR_MISC; rr_known = red[0]; }
                absres = std::sqrt(rr_known);
                relres = absres / normr0;
                break;
            case STOP_REL_PRECRES:
                if ((st = apply_pc()) < 0) return st;
                if (d_dot(m, z, r, red, dist) < 0) return ERROR_MISC;
                absres = std::sqrt(std::fabs(red[0]));
                relres = absres / normr0;
                break;
            case STOP_MOD_REL_RES:
                if (!have_rr) { if (d_dot(m, r, r, red, dist) < 0) return ERROR_MISC; rr_known = red[0]; }
                absres = std::sqrt(rr_known);
                relres = absres / normu;
                break;
        }
        return FASP_SUCCESS;
    };

    if (PrtLvl > PRINT_NONE) std::printf("\nCalling CG solver (%s) ...\n", K.fmt);

    { if (K.halo(u) < 0) return ERROR_MISC; K.resid(u, b, r); }  // r = b - A u
    if ((st = apply_pc()) < 0) return st;
    switch (StopType) {
        case STOP_REL_RES:
            if (d_dot(m, r, r, red, dist) < 0) return ERROR_MISC;
            absres0 = std::sqrt(red[0]);
            normr0  = std::max(SMALLREAL, absres0);
            relres  = absres0 / normr0;
            break;
        case STOP_REL_PRECRES:
            if (d_dot(m, r, z, red, dist) < 0) return ERROR_MISC;
            absres0 = std::sqrt(red[0]);
            normr0  = std::max(SMALLREAL, absres0);
            relres  = absres0 / normr0;
            break;
        case STOP_MOD_REL_RES:
            if (d_dot(m, r, r, red, dist) < 0) return ERROR_MISC;
            absres0 = std::sqrt(red[0]);
            if (d_dot(m, u, u, red, dist) < 0) return ERROR_MISC;
            normu  = std::max(SMALLREAL, std::sqrt(red[0]));
            relres = absres0 / normu;
            break;
        default:
            std::printf("### ERROR: Unknown stopping type! [%s]\n", "fasp_solver_dcsr_pcg");
            goto FINISHED;
    }
    hist.push(absres0);
    if (relres < tol || absres0 < abstol) goto FINISHED;

    itinfo(PrtLvl, StopType, iter, relres, absres0, 0.0);
    HIPCK(hipMemcpyAsync(p, z, sizeof(double) * m, hipMemcpyDeviceToDevice, s));
    if (d_dot(m, z, r, red, dist) < 0) return ERROR_MISC;
    temp1 = red[0];

    while (iter++ < MaxIt) {
        // t = A p with the partial sums of (t,p); timed for the roofline report
        {
            if (K.halo(p) < 0) return ERROR_MISC;
            EventPair* ep = (h && h->ev_used < (int)h->ev.size()) ? &h->ev[h->ev_used++] : nullptr;
            if (ep) (void)hipEventRecord(ep->a, s);
            int gdot = K.mxv_dot ? K.mxv_dot(p, t) : -1;
            if (ep) (void)hipEventRecord(ep->b, s);
            if (gdot >= 0) d_finalize(gdot, 1, 0u, 8, dist);
            else {  // format without a fused kernel: t = A p, then (t,p) into slot 8
                K.mxv(p, t);
                if (d_dot_to(m, t, p, 8, dist) < 0) return ERROR_MISC;
            }
        }
        // alpha = temp1/(t,p) on device; u += alpha p; r -= alpha t; partial ||r||^2
        hipLaunchKernelGGL(k_cg_update, dim3(G), dim3(BLOCK), 0, s, m, temp1, (const double*)(g_ctx.d_red + 8),
                           (const double*)nullptr, 0, p, t, u, r, g_ctx.d_partials, 0, (double*)nullptr);
        d_finalize(G, 1, 0u, 0, dist);
        HIPCK(hipMemcpyAsync(g_ctx.h_red, g_ctx.d_red, sizeof(double) * 9, hipMemcpyDeviceToHost, s));
        HIPCK(hipStreamSynchronize(s));
        temp2 = g_ctx.h_red[8];
        if (std::fabs(temp2) > SMALLREAL2) {
            alpha = temp1 / temp2;
        } else {
            std::printf("### WARNING: Divided by zero! [%s:%d]\n", "fasp_solver_dcsr_pcg", 175);
            goto FINISHED;
        }
        if ((st = resnorm(g_ctx.h_red[0], true)) < 0) return st;
        factor = absres / absres0;
        itinfo(PrtLvl, StopType, iter, relres, absres, factor);
        hist.push(absres);

        if (factor > 0.9) {  // Check I / II, only when converging slowly
            if (d_norms(m, u, red, dist) < 0) return ERROR_MISC;
            if (red[1] <= sol_inf_tol) {
                if (PrtLvl > PRINT_MIN)
                    std::printf("### WARNING: Iteration stopped -- solution almost zero! [%s:%d]\n",
                                "fasp_solver_dcsr_pcg", 218);
                iter = ERROR_SOLVER_SOLSTAG;
                break;
            }
            normu = std::sqrt(red[0]);
            if (d_dot(m, p, p, red, dist) < 0) return ERROR_MISC;
            reldiff = std::fabs(alpha) * std::sqrt(red[0]) / normu;
            if ((stag <= MAX_STAG) & (reldiff < maxdiff)) {
                if (PrtLvl >= PRINT_MORE) {
                    std::printf("||u-u'|| = %.10e and the comp. rel. res. = %.10e.\n", reldiff, relres);
                    std::printf("### WARNING: Iteration restarted -- stagnation! [%s:%d]\n",
                                "fasp_solver_dcsr_pcg", 232);
                }
                { if (K.halo(u) < 0) return ERROR_MISC; K.resid(u, b, r); }
                if ((st = resnorm(0.0, false)) < 0) return st;
                if (PrtLvl >= PRINT_MORE)
                    std::printf("### WARNING: The actual relative residual = %.10e!\n", relres);
                if (relres < tol) break;
                if (stag >= MAX_STAG) {
                    if (PrtLvl > PRINT_MIN)
                        std::printf("### WARNING: Iteration stopped -- staggnation! [%s:%d]\n",
                                    "fasp_solver_dcsr_pcg", 266);
                    iter = ERROR_SOLVER_STAG;
                    break;
                }
                HIPCK(hipMemsetAsync(p, 0, sizeof(double) * m, s));
                ++stag;
            }
        }

        if (relres < tol) {  // Check III: prevent false convergence
            const double updated_relres = relres;
            { if (K.halo(u) < 0) return ERROR_MISC; K.resid(u, b, r); }
            if ((st = resnorm(0.0, false)) < 0) return st;
            if (relres < tol) break;
            if (PrtLvl >= PRINT_MORE) {
                std::printf("### WARNING: The computed relative residual = %.10e!\n", updated_relres);
                std::printf("### WARNING: The actual relative residual = %.10e!\n", relres);
            }
            if (more_step >= MAX_RESTART) {
                if (PrtLvl > PRINT_MIN)
                    std::printf("### WARNING: The tolerence might be too small! [%s:%d]\n",
                                "fasp_solver_dcsr_pcg", 315);
                iter = ERROR_SOLVER_TOLSMALL;
                break;
            }
            HIPCK(hipMemsetAsync(p, 0, sizeof(double) * m, s));
            ++more_step;
        }

        absres0 = absres;
        if (StopType != STOP_REL_PRECRES)
            if ((st = apply_pc()) < 0) return st;
        if (d_dot(m, z, r, red, dist) < 0) return ERROR_MISC;
        temp2 = red[0];
        beta  = temp2 / temp1;
        temp1 = temp2;
        d_axpby(m, 1.0, z, beta, p);  // p = z + beta p
    }

FINISHED:
    if (PrtLvl > PRINT_NONE) {  // ITS_FINAL, KryUtil.inl:95-105
        if (iter > MaxIt)
            std::printf("### WARNING: MaxIt = %d reached with relative residual %.10e.\n", MaxIt, relres);
        else if (iter >= 0)
            std::printf("Number of iterations = %d with relative residual %.10e.\n", iter, relres);
    }
    hist.push(absres);  // trailing entry: absres at exit (true residual after Check III)
    out.relres = relres; out.absres = absres; out.normr0 = normr0;
    HIPCK(hipStreamSynchronize(s));
    if (iter > MaxIt) return ERROR_SOLVER_MAXIT;
    return iter;
}

// CG as a smoother: fasp_solver_dcsr_pcg(A, b, x, NULL, 1e-3, 1e-15, nsweeps, STOP_REL_RES, PRINT_NONE),
// PreMGSmoother.inl:116 / :222 -- `nsweeps` CG steps on the level's system; its return code (normally
// "MaxIt reached") is ignored there too
static int cg_smooth(fasp_hip_amg* h, int level, int nsweeps)
{
    DevLevel& D = h->L[level];
    for (int q = 0; q < 3; ++q) if (!D.kw[q]) { if (alloc_vec(&D.kw[q], (size_t)D.nvec) < 0) return ERROR_ALLOC_MEM; }
    materialise_zero(D);
    KOps K = csr_ops(h, level, false);
    K.stats = nullptr;
    PcgVecs V{D.b, D.x, D.kw[0], D.kw[1], D.kw[2]};
    Hist H{nullptr, 0, 0};
    PcgOut po{BIGREAL, BIGREAL, BIGREAL};
    const int st = pcg_device(K, V, 1e-3, 1e-15, nsweeps, STOP_REL_RES, PRINT_NONE, H, po);
    return (st == ERROR_MISC || st == ERROR_ALLOC_MEM) ? st : FASP_SUCCESS;
}

}  // namespace fasp

// ---------------------------------------------------------------------------
// BSR operators (config 3): host pointers in/out, device kernels
// ---------------------------------------------------------------------------
namespace fasp_bsr {
struct TmpBSR {
    int ROW = 0, nb = 0, NNZ = 0;
    int *ia = nullptr, *ja = nullptr;
    double* val = nullptr;
    bool ok = false;
    explicit TmpBSR(const dBSRmat* A)
    {
        if (ctx_init() < 0 || !A || A->nb < 1 || A->nb > 7 || A->storage_manner != 0) return;
        ROW = A->ROW; nb = A->nb; NNZ = A->NNZ;
        const size_t nv = (size_t)NNZ * nb * nb;
        if (hipMalloc(&ia, sizeof(int) * ((size_t)ROW + 1)) != hipSuccess) return;
        if (hipMalloc(&ja, sizeof(int) * std::max(NNZ, 1)) != hipSuccess) return;
        if (hipMalloc(&val, sizeof(double) * std::max<size_t>(nv, 1)) != hipSuccess) return;
        (void)hipMemcpy(ia, A->IA, sizeof(int) * ((size_t)ROW + 1), hipMemcpyHostToDevice);
        (void)hipMemcpy(ja, A->JA, sizeof(int) * (size_t)NNZ, hipMemcpyHostToDevice);
        (void)hipMemcpy(val, A->val, sizeof(double) * nv, hipMemcpyHostToDevice);
        ok = true;
    }
    ~TmpBSR() { if (ia) (void)hipFree(ia); if (ja) (void)hipFree(ja); if (val) (void)hipFree(val); }
};

template <int OP>
void launch_bsr(const TmpBSR& M, BsrArgs a)
{
    a.ROW = M.ROW; a.ia = M.ia; a.ja = M.ja; a.val = M.val;
    const int rw = 64 / M.nb;
    a.ntiles = (M.ROW + 4 * rw - 1) / (4 * rw);
#define BSR_CASE(NBV)                                                                              \
    case NBV: {                                                                                    \
        int cap = resident_blocks_per_cu(k_bsr_wstream<NBV, OP>) * g_ctx.num_cu;                   \
        const int grid = std::max(1, std::min(std::min(cap, MAXGRID), a.ntiles));                  \
        hipLaunchKernelGGL((k_bsr_wstream<NBV, OP>), dim3(grid), dim3(BLOCK), 0, g_ctx.stream, a); \
    } break;
    switch (M.nb) {
        BSR_CASE(1) BSR_CASE(2) BSR_CASE(3) BSR_CASE(4) BSR_CASE(5) BSR_CASE(6) BSR_CASE(7)
        default: break;
    }
#undef BSR_CASE
}
[[noreturn]] void die_bsr(const char* fn)
{
    std::fprintf(stderr, "### ERROR: %s: needs a HIP device, storage_manner 0 and 1 <= nb <= 7 "
                         "(libfasp_hip has no CPU fallback)\n", fn);
    std::exit(ERROR_MISC);
}
}  // namespace fasp_bsr
using namespace fasp_bsr;

// ---------------------------------------------------------------------------
// BSR AMG hierarchy resident in HBM (config 3): unsmoothed aggregation, block-Jacobi
// V/W cycle (PreMGCycle.c:287), GMRES on the coarsest level, Krylov drivers shared with CSR
// ---------------------------------------------------------------------------
struct BsrLevel {
    std::unique_ptr<TmpBSR> A, P, R;
    double *dinv = nullptr, *b = nullptr, *x = nullptr, *x2 = nullptr, *w = nullptr;
    int  n = 0;  // scalar rows
    bool x_zero = false;
    DevLevel::Sched sched[2];  // level schedules of the sequential block sweeps: 0 ascending, 1 descending
};
struct fasp_hip_amg_bsr {
    HostHierarchyBSR      H;
    std::vector<BsrLevel> L;
    AMG_param             param;
    double *b = nullptr, *u = nullptr, *p = nullptr, *t = nullptr, *r = nullptr, *z = nullptr;
    std::vector<double*> gm[2];
    size_t               gm_len[2] = {0, 0};
    double*              gm_hh = nullptr;
    double*              small_ws = nullptr;  // workspace of the single-workgroup coarse GMRES
    long long            coarse_iters = 0, vcycles = 0;
};

namespace fasp_bsr {

static int dalloc(double** p, size_t n)
{
    HIPCK(hipMalloc(p, sizeof(double) * std::max<size_t>(n, 1)));
    HIPCK(hipMemsetAsync(*p, 0, sizeof(double) * n, g_ctx.stream));
    return 0;
}

static void bsr_mxv(const TmpBSR& M, const double* x, double* y)
{
    BsrArgs a{}; a.x = x; a.y = y;
    launch_bsr<0>(M, a);
}
// r = b - A x with the reference's rounding: y = b; y *= -1; y += A x; y *= -1 (BlaSpmvBSR.c:548)
static void bsr_resid(const TmpBSR& M, const double* x, const double* b, double* r)
{
    BsrArgs a{}; a.x = x; a.y = r; a.b = b; a.alpha = -1.0;
    launch_bsr<1>(M, a);
}
static void bsr_jacobi(BsrLevel& Lv)
{
    const TmpBSR& M = *Lv.A;
    if (Lv.x_zero) {
        hipLaunchKernelGGL(k_bsr_dinv_apply, dim3(vec_grid(Lv.n)), dim3(BLOCK), 0, g_ctx.stream, Lv.n, M.nb,
                           (const double*)Lv.dinv, (const double*)Lv.b, Lv.x);
        Lv.x_zero = false;
        return;
    }
    BsrArgs a{}; a.x = Lv.x; a.y = Lv.x2; a.b = Lv.b; a.dinv = Lv.dinv;
    launch_bsr<2>(M, a);
    std::swap(Lv.x, Lv.x2);
}

// One sequential block sweep (Gauss-Seidel or SOR, ascending or descending) as level-scheduled launches:
// the rows of a dependency level are mutually uncoupled, so the result is the sequential sweep.
static int bsr_seq_sweep(fasp_hip_amg_bsr* h, int level, bool descend, bool sor, double w)
{
    BsrLevel& Lv = h->L[level];
    DevLevel::Sched& S = Lv.sched[descend ? 1 : 0];
    const TmpBSR& M = *Lv.A;
    if (!S.built) {
        const HostBSR& A = h->H.L[level].A;
        HostCSR pat;  // block pattern only (build_schedule does not read values)
        pat.row = A.ROW; pat.col = A.COL; pat.nnz = A.NNZ;
        pat.ia.alloc((size_t)A.ROW + 1); pat.ja.alloc((size_t)std::max(A.NNZ, 1));
        std::memcpy(pat.ia.data(), A.ia.data(), sizeof(int) * ((size_t)A.ROW + 1));
        std::memcpy(pat.ja.data(), A.ja.data(), sizeof(int) * (size_t)A.NNZ);
        std::vector<int> seq((size_t)A.ROW);
        for (int i = 0; i < A.ROW; ++i) seq[(size_t)i] = descend ? A.ROW - 1 - i : i;
        const int st = build_schedule(pat, seq, S);
        if (st < 0) return st;
    }
    if (Lv.x_zero) { HIPCK(hipMemsetAsync(Lv.x, 0, sizeof(double) * Lv.n, g_ctx.stream)); Lv.x_zero = false; }
    const int nlev = (int)S.ptr.size() - 1;
    for (int l = 0; l < nlev; ++l) {
        const int lo = S.ptr[l], hi = S.ptr[l + 1];
        const int grid = std::max(1, std::min(MAXGRID, (hi - lo + BLOCK - 1) / BLOCK));
#define BSEQ_LAUNCH(NBV) hipLaunchKernelGGL((k_bsr_seq_level<NBV>), dim3(grid), dim3(BLOCK), 0, g_ctx.stream, \
        (const int*)S.d_order, lo, hi, (const int*)M.ia, (const int*)M.ja, (const double*)M.val, (const double*)Lv.b, \
        (const double*)Lv.dinv, Lv.x, sor ? 1 : 0, w)
        switch (M.nb) {
            case 1: BSEQ_LAUNCH(1); break;
            case 2: BSEQ_LAUNCH(2); break;
            case 3: BSEQ_LAUNCH(3); break;
            default: return ERROR_INPUT_PAR;  // inverse diagonal blocks exist for nb <= 3 only
        }
#undef BSEQ_LAUNCH
    }
    return FASP_SUCCESS;
}

// smoother dispatch of fasp_solver_mgcycle_bsr, PreMGCycle.c:327-365 (pre) and :513-549 (post)
static int bsr_smooth(fasp_hip_amg_bsr* h, int level, bool post, int smoother, int steps, double relax)
{
    BsrLevel& Lv = h->L[level];
    int st = FASP_SUCCESS;
    if (steps <= 0) return st;
    switch (smoother) {
        case SMOOTHER_JACOBI: for (int i = 0; i < steps; ++i) bsr_jacobi(Lv); break;
        case SMOOTHER_GS: for (int i = 0; i < steps && st >= 0; ++i) st = bsr_seq_sweep(h, level, post, false, 0.0); break;
        case SMOOTHER_SGS:
            for (int i = 0; i < steps && st >= 0; ++i) {
                st = bsr_seq_sweep(h, level, false, false, 0.0);
                if (st >= 0) st = bsr_seq_sweep(h, level, true, false, 0.0);
            }
            break;
        case SMOOTHER_SOR: for (int i = 0; i < steps && st >= 0; ++i) st = bsr_seq_sweep(h, level, post, true, relax); break;
        case SMOOTHER_SSOR:  // `steps` ascending sweeps, then ONE descending sweep -- before and after the coarse correction
            for (int i = 0; i < steps && st >= 0; ++i) st = bsr_seq_sweep(h, level, false, true, relax);
            if (st >= 0) st = bsr_seq_sweep(h, level, true, true, relax);
            break;
        default: return ERROR_AMG_SMOOTH_TYPE;
    }
    return st;
}

static KOps bsr_ops(fasp_hip_amg_bsr* h, int level, int set);

// fasp_solver_mgcycle_bsr, PreMGCycle.c:287-566
static int mgcycle_bsr(fasp_hip_amg_bsr* h, const AMG_param& param)
{
    const int nl = (int)h->L.size(), cycle_type = param.cycle_type, steps = param.presmooth_iter;
    int nu_l[MAX_AMG_LVL + 1] = {0}, l = 0;
    hipStream_t s = g_ctx.stream;
    ++h->vcycles;
ForwardSweep:
    while (l < nl - 1) {
        BsrLevel& Lv = h->L[l];
        ++nu_l[l];
        { const int st = bsr_smooth(h, l, false, param.smoother, steps, param.relaxation); if (st < 0) return st; }
        if (Lv.x_zero) { HIPCK(hipMemsetAsync(Lv.x, 0, sizeof(double) * Lv.n, s)); Lv.x_zero = false; }
        bsr_resid(*Lv.A, Lv.x, Lv.b, Lv.w);
        bsr_mxv(*Lv.R, Lv.w, h->L[l + 1].b);
        ++l;
        h->L[l].x_zero = true;  // fasp_dvec_set(.., 0.0), materialised lazily
    }
    {   // coarsest level: fasp_solver_dbsr_pvgmres(A, b, x, NULL, tol, tol*1e-8, min(n^2,200), 25, 1, 0), :443-459
        BsrLevel& Lc = h->L[nl - 1];
        if (Lc.x_zero) { HIPCK(hipMemsetAsync(Lc.x, 0, sizeof(double) * Lc.n, s)); Lc.x_zero = false; }
        const int csize = Lc.n;
        const int cmaxit = (int)std::min<unsigned>((unsigned)csize * (unsigned)csize, 200u);
        const double ctol = param.tol, atol = ctol * 1e-8;
        int st;
        const TmpBSR& Ac = *Lc.A;
        if (small_coarse_ok(csize, (long long)Ac.NNZ * Ac.nb * Ac.nb)) {
            if (!h->small_ws) HIPCK(hipMalloc(&h->small_ws, sizeof(double) * (size_t)(25 + 2) * std::max(csize, 1)));
            GmresArgs<SmallBSR> a{};
            a.A = SmallBSR{Ac.ROW, Ac.nb, Ac.ia, Ac.ja, Ac.val};
            a.b = Lc.b; a.x = Lc.x; a.ws = h->small_ws; a.tol = ctol; a.abstol = atol;
            a.MaxIt = cmaxit; a.restart = 25; a.out = small_out_dev();
            const size_t lds = sizeof(double) * (size_t)(25 + 2) * (size_t)csize;
            static bool attr = false;
            if (!attr) {
                (void)hipFuncSetAttribute((const void*)k_gmres_small<SmallBSR, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
                attr = true;
            }
            if (g_tune.small_lds && lds <= 140 * 1024)
                hipLaunchKernelGGL((k_gmres_small<SmallBSR, true>), dim3(1), dim3(SMALL_BLOCK), lds, s, a);
            else
                hipLaunchKernelGGL((k_gmres_small<SmallBSR, false>), dim3(1), dim3(SMALL_BLOCK), 0, s, a);
            SmallOut o;
            if (small_out_fetch(o) < 0) return ERROR_MISC;
            st = o.status;
            h->coarse_iters += o.iters;
        } else {
            KOps K = bsr_ops(h, nl - 1, 1);
            PcgOut po{BIGREAL, BIGREAL, BIGREAL};
            st = gmres_device(K, Lc.b, Lc.x, 0, ctol, atol, cmaxit, 25, STOP_REL_RES, 0, nullptr, &po);
            if (st >= 0) h->coarse_iters += st;
        }
        if (st < 0 && st != ERROR_SOLVER_MAXIT && st != ERROR_SOLVER_STAG && st != ERROR_SOLVER_SOLSTAG &&
            st != ERROR_SOLVER_TOLSMALL) return st;  // device failure, not a convergence verdict
        if (st < 0 && param.print_level > PRINT_MIN) {
            std::printf("### WARNING: Coarse level solver did not converge!\n");
            std::printf("### WARNING: Consider to increase maxit to %d!\n", 2 * cmaxit);
        }
    }
    while (l > 0) {
        --l;
        BsrLevel& Lv = h->L[l];
        {   // x_l += P x_{l+1}  (fasp_blas_dbsr_aAxpy with alpha = 1)
            BsrArgs a{}; a.x = h->L[l + 1].x; a.y = Lv.x; a.alpha = 1.0;
            launch_bsr<1>(*Lv.P, a);
        }
        // the reference post-smooths `steps` = presmooth_iter times (:543)
        { const int st = bsr_smooth(h, l, true, param.smoother, steps, param.relaxation); if (st < 0) return st; }
        if (nu_l[l] < cycle_type) break;
        nu_l[l] = 0;
    }
    if (l > 0) goto ForwardSweep;
    return FASP_SUCCESS;
}

// fasp_precond_dbsr_amg, PreBSR.c:1149: z = (maxit cycles from a zero guess)(r); the AMG_param
// handed to the cycle is re-initialised (tol stays 1e-6) apart from the copied fields
static int precond_amg_bsr(fasp_hip_amg_bsr* h, double* r, double** z)
{
    AMG_param p;
    fasp_param_amg_init(&p);
    const AMG_param& u = h->param;
    p.cycle_type = u.cycle_type; p.smoother = u.smoother; p.presmooth_iter = u.presmooth_iter;
    p.postsmooth_iter = u.postsmooth_iter; p.relaxation = u.relaxation;
    p.coarse_scaling = u.coarse_scaling; p.tentative_smooth = u.tentative_smooth;
    BsrLevel& L0 = h->L[0];
    double* saved_b = L0.b;
    L0.b = r;  // level-0 rhs aliases the Krylov residual (the cycle never writes b_0)
    L0.x_zero = true;
    int st = FASP_SUCCESS;
    for (int i = u.maxit; i--;)
        if ((st = mgcycle_bsr(h, p)) < 0) break;
    L0.b = saved_b;
    if (L0.x_zero) { HIPCK(hipMemsetAsync(L0.x, 0, sizeof(double) * L0.n, g_ctx.stream)); L0.x_zero = false; }
    *z = L0.x;
    return st;
}

static KOps bsr_ops(fasp_hip_amg_bsr* h, int level, int set)
{
    KOps K;
    BsrLevel* Lv = &h->L[level];
    K.n = Lv->n; K.nvec = (size_t)Lv->n; K.fmt = "BSR"; K.dist = false;
    K.halo = [](double*) { return 0; };
    K.mxv = [Lv](const double* x, double* y) { bsr_mxv(*Lv->A, x, y); };
    K.resid = [Lv](const double* x, const double* b, double* r) { bsr_resid(*Lv->A, x, b, r); };
    if (set == 0) K.pc = [h](double* in, double** out) { return precond_amg_bsr(h, in, out); };
    K.ws = &h->gm[set]; K.ws_len = &h->gm_len[set]; K.hh = &h->gm_hh;
    K.stats = nullptr;
    return K;
}

}  // namespace fasp_bsr


// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

int fasp_hip_set_device(int device)
{
    if (g_ctx.ready && device != g_ctx.device) {
        std::fprintf(stderr, "### ERROR: fasp_hip: device already bound to %d\n", g_ctx.device);
        return ERROR_INPUT_PAR;
    }
    g_requested_device = device;
    return ctx_init();
}

int fasp_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return ERROR_MISC;
    return n;
}

int fasp_hip_available(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    return 1;
}

int fasp_hip_amg_create_host(fasp_hip_amg** out, const dCSRmat* A, AMG_param* amgparam)
{
    if (!out || !A || !amgparam) return ERROR_INPUT_PAR;
    *out = nullptr;
    int st = check_supported(nullptr, amgparam);
    if (st < 0) return st;
    fasp_hip_amg* h = new fasp_hip_amg();
    st = (amgparam->AMG_type == SA_AMG)   ? host_setup_sa(A, amgparam, h->H)  // SolCSR.c:509-521
         : (amgparam->AMG_type == UA_AMG) ? host_setup_ua(A, amgparam, h->H)
                                          : host_setup_rs(A, amgparam, h->H);
    if (st < 0) { delete h; return st; }
    h->param = *amgparam;
    *out = h;
    return FASP_SUCCESS;
}

int fasp_hip_amg_upload(fasp_hip_amg* h)
{
    if (!h) return ERROR_INPUT_PAR;
    if (!h->L.empty()) return FASP_SUCCESS;
    int st = ctx_init();
    if (st < 0) return st;
    return upload_hierarchy(h);
}

int fasp_hip_amg_create(fasp_hip_amg** out, const dCSRmat* A, AMG_param* amgparam)
{
    if (!out || !A || !amgparam) return ERROR_INPUT_PAR;
    *out = nullptr;
    int st = check_supported(nullptr, amgparam);
    if (st < 0) return st;
    if ((st = ctx_init()) < 0) return st;  // fail before the (long) host setup when there is no GPU
    fasp_hip_amg* h = nullptr;
    st = fasp_hip_amg_create_host(&h, A, amgparam);
    if (st < 0) return st;
    st = upload_hierarchy(h);
    if (st < 0) { fasp_hip_amg_destroy(h); return st; }
    *out = h;
    return FASP_SUCCESS;
}

void fasp_hip_amg_destroy(fasp_hip_amg* h)
{
    if (!h) return;
    if (g_ctx.ready) (void)hipStreamSynchronize(g_ctx.stream);
    for (auto& D : h->L) free_level(D);
    double* v[] = {h->b, h->u, h->p, h->t, h->r, h->cp, h->cr, h->ct, h->cbest};
    for (double* q : v)
        if (q) (void)hipFree(q);
    for (auto& e : h->ev) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (int s = 0; s < 2; ++s)
        for (double* q : h->gm[s])
            if (q) (void)hipFree(q);
    if (h->gm_hh) (void)hipFree(h->gm_hh);
    if (h->spcg_state) (void)hipFree(h->spcg_state);
    delete h;
}

// Host-only check of the lossless matrix coding (no GPU needed): codes A the way upload_csr would,
// decodes it again and compares with A bit for bit.  *kind_out = 5 (row patterns), 4 (byte
// dictionary) or 0 (stays plain CSR).  Returns 0 when the round trip is exact.
int fasp_hip_coding_selftest(const dCSRmat* A, int* kind_out)
{
    if (!A || !kind_out) return ERROR_INPUT_PAR;
    HostCSR M;
    M.row = A->row; M.col = A->col; M.nnz = A->nnz;
    M.ia.alloc((size_t)A->row + 1); M.ja.alloc((size_t)std::max(A->nnz, 1)); M.val.alloc((size_t)std::max(A->nnz, 1));
    std::memcpy(M.ia.data(), A->IA, sizeof(int) * ((size_t)A->row + 1));
    std::memcpy(M.ja.data(), A->JA, sizeof(int) * (size_t)A->nnz);
    std::memcpy(M.val.data(), A->val, sizeof(double) * (size_t)A->nnz);
    const bool square = M.row == M.col;
    *kind_out = 0;
    if (!(M.nnz >= 4096 && (double)M.nnz <= 48.0 * M.row)) return FASP_SUCCESS;
    auto bits = [](double v) { unsigned long long b; std::memcpy(&b, &v, 8); return b; };
    {
        Buf<unsigned short> pat; Buf<int> rb;
        std::vector<int> pstart, plen, poff; std::vector<double> pval;
        if (build_rowpat(M, pat, pstart, plen, poff, pval, rb)) {
            *kind_out = 5;
            for (int r = 0; r < M.row; ++r) {
                const int id = pat[r], base = square ? r : rb[r];
                if (plen[id] != M.ia[r + 1] - M.ia[r]) return ERROR_MISC;
                for (int j = 0; j < plen[id]; ++j) {
                    const int k = M.ia[r] + j;
                    if (base + poff[pstart[id] + j] != M.ja[k] || bits(pval[pstart[id] + j]) != bits(M.val[k])) return ERROR_MISC;
                }
                for (int j = plen[id]; j < (plen[id] + 7) / 8 * 8; ++j)  // padding: offset 0, value +0.0
                    if (poff[pstart[id] + j] != 0 || bits(pval[pstart[id] + j]) != 0ull) return ERROR_MISC;
            }
            return FASP_SUCCESS;
        }
    }
    {
        std::vector<int> doff; std::vector<double> dval;
        Buf<unsigned char> code; Buf<int> rb;
        if (build_dict8(M, doff, dval, code, rb)) {
            *kind_out = 4;
            for (int r = 0; r < M.row; ++r) {
                const int base = square ? r : rb[r];
                for (int k = M.ia[r]; k < M.ia[r + 1]; ++k)
                    if (base + doff[code[k]] != M.ja[k] || bits(dval[code[k]]) != bits(M.val[k])) return ERROR_MISC;
            }
        }
    }
    return FASP_SUCCESS;
}

int fasp_hip_amg_num_levels(const fasp_hip_amg* h) { return h ? (int)h->H.L.size() : ERROR_INPUT_PAR; }

// which kernel family serves operator `which` (0 A, 1 P, 2 R) of a level, and how many bytes of
// matrix data one pass of it reads (row pointers / indices / values, or their coded form)
int fasp_hip_amg_kernel_info(const fasp_hip_amg* h, int level, int which, int* kind, double* matrix_bytes)
{
    if (!h || level < 0 || level >= (int)h->L.size() || which < 0 || which > 2) return ERROR_INPUT_PAR;
    const DevLevel& D = h->L[level];
    const DevCSR& M = which == 0 ? D.A : which == 1 ? D.P : D.R;
    if (!M.ia) return ERROR_INPUT_PAR;
    int k = M.kind;
    double bytes = 12.0 * M.nnz + 4.0 * (M.row + 1.0);
    if (M.code && g_tune.compress) { k = 4; bytes = 1.0 * M.nnz + 4.0 * (M.row + 1.0) + (M.rowbase ? 4.0 * M.row : 0.0); }
    if (M.pat && g_tune.compress) { k = 5; bytes = 2.0 * M.row + (M.rowbase ? 4.0 * M.row : 0.0) + 12.0 * M.npent; }
    if (kind) *kind = k;
    if (matrix_bytes) *matrix_bytes = bytes;
    return FASP_SUCCESS;
}

int fasp_hip_amg_get_matrix(const fasp_hip_amg* h, int level, int which, dCSRmat* view)
{
    if (!h || !view || level < 0 || level >= (int)h->H.L.size()) return ERROR_INPUT_PAR;
    const HostLevel& L = h->H.L[level];
    if (which != 0 && !L.has_coarse) return ERROR_INPUT_PAR;
    *view = which == 0 ? L.A.view() : which == 1 ? L.P.view() : L.R.view();
    return FASP_SUCCESS;
}

int fasp_hip_amg_get_cfmark(const fasp_hip_amg* h, int level, ivector* view)
{
    if (!h || !view || level < 0 || level >= (int)h->H.L.size() || !h->H.L[level].has_coarse)
        return ERROR_INPUT_PAR;
    view->row = h->H.L[level].A.row;
    view->val = const_cast<int*>(h->H.L[level].cfmark.data());
    return FASP_SUCCESS;
}

int fasp_hip_set_rhs(fasp_hip_amg* h, const dvector* b)
{
    if (!h || !b || h->L.empty()) return ERROR_INPUT_PAR;
    const DevLevel& D0 = h->L[0];
    if (b->row != D0.nglobal) return ERROR_MAT_SIZE;  // host vectors are global; a rank uploads its rows
    HIPCK(hipMemcpyAsync(h->b, b->val + D0.row0, sizeof(double) * D0.nloc, hipMemcpyHostToDevice, g_ctx.stream));
    HIPCK(hipStreamSynchronize(g_ctx.stream));
    return FASP_SUCCESS;
}

int fasp_hip_set_guess(fasp_hip_amg* h, const dvector* x)
{
    if (!h || h->L.empty()) return ERROR_INPUT_PAR;
    const DevLevel& D0 = h->L[0];
    if (x) {
        if (x->row != D0.nglobal) return ERROR_MAT_SIZE;
        HIPCK(hipMemcpyAsync(h->u, x->val + D0.row0, sizeof(double) * D0.nloc, hipMemcpyHostToDevice, g_ctx.stream));
    } else {
        HIPCK(hipMemsetAsync(h->u, 0, sizeof(double) * D0.nvec, g_ctx.stream));
    }
    HIPCK(hipStreamSynchronize(g_ctx.stream));
    return FASP_SUCCESS;
}

int fasp_hip_get_solution(fasp_hip_amg* h, dvector* x)
{
    if (!h || !x || h->L.empty()) return ERROR_INPUT_PAR;
    const DevLevel& D0 = h->L[0];
    if (x->row != D0.nglobal) return ERROR_MAT_SIZE;  // a rank fills the rows it owns
    HIPCK(hipMemcpyAsync(x->val + D0.row0, h->u, sizeof(double) * D0.nloc, hipMemcpyDeviceToHost, g_ctx.stream));
    HIPCK(hipStreamSynchronize(g_ctx.stream));
    return FASP_SUCCESS;
}

int fasp_hip_device_synchronize(void)
{
    if (!g_ctx.ready) return FASP_SUCCESS;
    HIPCK(hipStreamSynchronize(g_ctx.stream));
    return FASP_SUCCESS;
}

int fasp_hip_solve_resident(fasp_hip_amg* h, const ITS_param* itparam, double* hist, int hist_cap,
                            fasp_hip_stats* stats)
{
    if (!h || !itparam) return ERROR_INPUT_PAR;
    if (h->L.empty()) return ERROR_INPUT_PAR;  // hierarchy not uploaded
    int st = check_supported(itparam, &h->param);
    if (st < 0) return st;
    // ITS_CHECK, KryUtil.inl:71-83
    if (itparam->tol < SMALLREAL)
        std::printf("### WARNING: Convergence tolerance is too small! [%s:%d]\n", "ITS_CHECK", 74);
    if (itparam->maxit <= 0)
        std::printf("### WARNING: Max number of iterations must be POSITIVE! [%s:%d]\n", "ITS_CHECK", 78);

    h->ev_used = 0;
    h->use_fmg = itparam->precond_type == PREC_FMG;   // SolCSR.c:537-538
    const long long ci0 = h->coarse_iters, vc0 = h->vcycles;
    Hist   H{hist, hist_cap, 0};
    PcgOut po{BIGREAL, BIGREAL, BIGREAL};
    const double t0 = wall_seconds();
    // SolCSR.c:530-551: the AMG preconditioner is always installed on this path;
    // SolCSR.c:84-130: dispatch on itsolver_type (restart is narrowed to SHORT, :62)
    switch (itparam->itsolver_type) {
        case SOLVER_BiCGstab:
        {
            KOps K = csr_ops(h, 0, true);
            st = bicgstab_device(K, h->b, h->u, itparam->tol, itparam->maxit, itparam->print_level, &H, &po);
        } break;
        case SOLVER_GMRES:
        case SOLVER_VGMRES:
        case SOLVER_VFGMRES:
        {
            KOps K = csr_ops(h, 0, true);
            st = gmres_device(K, h->b, h->u, itparam->itsolver_type == SOLVER_VFGMRES ? 1 : itparam->itsolver_type == SOLVER_GMRES ? 3 : 0, itparam->tol,
                              itparam->abstol, itparam->maxit, (short)itparam->restart, itparam->stop_type,
                              itparam->print_level, &H, &po);
        } break;
        case SOLVER_MinRes:
        {
            KOps K = csr_ops(h, 0, true);
            st = minres_device(K, h->b, h->u, itparam->tol, itparam->abstol, itparam->maxit, itparam->stop_type,
                               itparam->print_level, &H, &po);
        } break;
        case SOLVER_GCG:
        {
            KOps K = csr_ops(h, 0, true);
            st = gcg_device(K, h->b, h->u, itparam->tol, itparam->abstol, itparam->maxit, itparam->stop_type,
                            itparam->print_level, &H, &po);
        } break;
        case SOLVER_GCR:
        {
            KOps K = csr_ops(h, 0, true);
            st = gcr_device(K, h->b, h->u, itparam->tol, itparam->abstol, itparam->maxit, (short)itparam->restart,
                            itparam->stop_type, itparam->print_level, &H, &po);
        } break;
        default:
        {
            KOps K = csr_ops(h, 0, true);
            PcgVecs V{h->b, h->u, h->p, h->t, h->r};
            st = pcg_device(K, V, itparam->tol, itparam->abstol, itparam->maxit, itparam->stop_type,
                            itparam->print_level, H, po);
        }
    }
    HIPCK(hipStreamSynchronize(g_ctx.stream));
    const double t_solve = wall_seconds() - t0;

    if (stats) {
        stats->iters = st; stats->nhist = H.n; stats->relres = po.relres; stats->absres = po.absres;
        stats->normr0 = po.normr0; stats->solve_seconds = t_solve; stats->upload_seconds = 0.0;
        double ms = 0.0;
        for (int i = 0; i < h->ev_used; ++i) {
            float e = 0.f;
            if (hipEventElapsedTime(&e, h->ev[i].a, h->ev[i].b) == hipSuccess) ms += e;
        }
        stats->spmv_launches = h->ev_used;
        stats->spmv_ms = h->ev_used ? ms / h->ev_used : 0.0;
        stats->coarse_iters = h->coarse_iters - ci0;
        stats->vcycles = h->vcycles - vc0;
    }
    if (itparam->print_level >= PRINT_SOME && st >= 0)
        std::printf("Iterative method costs %.4f seconds.\n", t_solve);
    return st;
}

int fasp_hip_solve(fasp_hip_amg* h, const dvector* b, dvector* x, const ITS_param* itparam, double* hist,
                   int hist_cap, fasp_hip_stats* stats)
{
    if (!h || !b || !x || !itparam) return ERROR_INPUT_PAR;
    double t0 = wall_seconds();
    int st = fasp_hip_set_rhs(h, b);
    if (st < 0) return st;
    if ((st = fasp_hip_set_guess(h, x)) < 0) return st;
    double t_up = wall_seconds() - t0;
    st = fasp_hip_solve_resident(h, itparam, hist, hist_cap, stats);
    t0 = wall_seconds();
    const int st2 = fasp_hip_get_solution(h, x);
    if (st2 < 0) return st2;
    t_up += wall_seconds() - t0;
    if (stats) stats->upload_seconds = t_up;
    return st;
}

// AMG as a stand-alone solver on a resident hierarchy (PreMGSolve.c:49); param == NULL: the
// parameters the hierarchy was built with
int fasp_hip_amg_solve(fasp_hip_amg* h, const dvector* b, dvector* x, const AMG_param* param, double* hist,
                       int hist_cap, fasp_hip_stats* stats)
{
    if (!h || !b || !x || h->L.empty()) return ERROR_INPUT_PAR;
    const AMG_param& p = param ? *param : h->param;
    int st = check_supported(nullptr, &p);
    if (st < 0) return st;
    double t0 = wall_seconds();
    if ((st = fasp_hip_set_rhs(h, b)) < 0) return st;
    if ((st = fasp_hip_set_guess(h, x)) < 0) return st;
    double t_up = wall_seconds() - t0;
    h->ev_used = 0;
    const long long ci0 = h->coarse_iters, vc0 = h->vcycles;
    Hist   H{hist, hist_cap, 0};
    PcgOut po{BIGREAL, BIGREAL, BIGREAL};
    t0 = wall_seconds();
    st = amg_solve_device(h, p, H, po);
    const double t_solve = wall_seconds() - t0;
    t0 = wall_seconds();
    const int st2 = fasp_hip_get_solution(h, x);
    if (st2 < 0) return st2;
    t_up += wall_seconds() - t0;
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->iters = st; stats->nhist = H.n; stats->relres = po.relres; stats->absres = po.absres;
        stats->normr0 = po.normr0; stats->solve_seconds = t_solve; stats->upload_seconds = t_up;
        stats->coarse_iters = h->coarse_iters - ci0;
        stats->vcycles = h->vcycles - vc0;
    }
    if (p.print_level > PRINT_NONE) std::printf("AMG solve costs %.4f seconds.\n", t_solve);
    return st;
}

// SolAMG.c:49.  A failed setup returns its error code (the reference would fall back to an
// unpreconditioned CPU GMRES there; this library has no CPU solve path).
int fasp_solver_amg(dCSRmat* A, dvector* b, dvector* x, AMG_param* param)
{
    if (!A || !b || !x || !param) return ERROR_INPUT_PAR;
    const double t0 = wall_seconds();
    int st = check_supported(nullptr, param);
    if (st < 0) return st;
    fasp_hip_amg* h = nullptr;
    g_oneshot_upload = std::getenv("FASP_HIP_ONESHOT_CODING") == nullptr;
    st = fasp_hip_amg_create(&h, A, param);
    g_oneshot_upload = false;
    if (st < 0) return st;
    st = fasp_hip_amg_solve(h, b, x, param, nullptr, 0, nullptr);
    if (param->print_level > PRINT_NONE) std::printf("AMG totally costs %.4f seconds.\n", wall_seconds() - t0);
    fasp_hip_amg_destroy(h);
    return st;
}

// SolFAMG.c:41 -> fasp_famg_solve (PreMGSolve.c:300): ONE full-multigrid cycle as the solver; x is the
// initial guess of the finest level and receives the result.  void in the reference; the status is an extension.
int fasp_solver_famg(const dCSRmat* A, const dvector* b, dvector* x, AMG_param* param)
{
    if (!A || !b || !x || !param) return ERROR_INPUT_PAR;
    const double t0 = wall_seconds();
    int st = check_supported(nullptr, param);
    if (st < 0) return st;
    fasp_hip_amg* h = nullptr;
    g_oneshot_upload = std::getenv("FASP_HIP_ONESHOT_CODING") == nullptr;
    st = fasp_hip_amg_create(&h, A, param);
    g_oneshot_upload = false;
    if (st < 0) return st;
    if ((st = fasp_hip_set_rhs(h, b)) >= 0 && (st = fasp_hip_set_guess(h, x)) >= 0) {
        DevLevel& D0 = h->L[0];
        const int m = D0.A.row;
        double red[2];
        D0.b = h->b;
        (void)hipMemcpyAsync(D0.x, h->u, sizeof(double) * m, hipMemcpyDeviceToDevice, g_ctx.stream);
        D0.x_zero = false;
        st = d_dot(m, h->b, h->b, red, false) < 0 ? ERROR_MISC : FASP_SUCCESS;
        const double sumb = std::sqrt(red[0]);
        if (st >= 0 && sumb <= SMALLREAL) (void)hipMemsetAsync(D0.x, 0, sizeof(double) * m, g_ctx.stream);
        if (st >= 0) st = fmg_cycle(h, *param);
        if (st >= 0) {
            d_resid(D0.A, D0.x, D0.b, D0.w);
            if (d_dot(m, D0.w, D0.w, red, false) < 0) st = ERROR_MISC;
            else if (param->print_level > PRINT_NONE)
                std::printf("FMG finishes with relative residual %e.\n", std::sqrt(red[0]) / std::max(SMALLREAL, sumb));
            (void)hipMemcpyAsync(h->u, D0.x, sizeof(double) * m, hipMemcpyDeviceToDevice, g_ctx.stream);
            const int st2 = fasp_hip_get_solution(h, x);
            if (st2 < 0) st = st2;
        }
    }
    if (param->print_level > PRINT_NONE) std::printf("FAMG totally costs %.4f seconds.\n", wall_seconds() - t0);
    fasp_hip_amg_destroy(h);
    return st;
}

int fasp_hip_precond_amg(fasp_hip_amg* h, const double* r, double* z)
{
    if (!h || !r || !z || h->L.empty()) return ERROR_INPUT_PAR;
    const int m = h->L[0].nloc;
    r += h->L[0].row0; z += h->L[0].row0;  // global host vectors, own rows
    HIPCK(hipMemcpyAsync(h->r, r, sizeof(double) * m, hipMemcpyHostToDevice, g_ctx.stream));
    double* dz = nullptr;
    const int st = precond_amg(h, h->r, &dz);
    if (st < 0) return st;
    HIPCK(hipMemcpyAsync(z, dz, sizeof(double) * m, hipMemcpyDeviceToHost, g_ctx.stream));
    HIPCK(hipStreamSynchronize(g_ctx.stream));
    return FASP_SUCCESS;
}

// ---- row-partition inspection (host only; used by the CPU-side distributed tests) ----
int fasp_hip_dist_plan(fasp_hip_amg* h, int rank, int nranks, int min_rows)
{
    if (!h) return ERROR_INPUT_PAR;
    return build_dist_plan(h->H, rank, nranks, min_rows, h->dist);
}

int fasp_hip_dist_level_info(const fasp_hip_amg* h, int level, int* info)
{
    if (!h || !info || level < 0 || level >= (int)h->dist.L.size()) return ERROR_INPUT_PAR;
    const DistLevel& D = h->dist.L[level];
    info[0] = D.replicated; info[1] = D.nglobal; info[2] = D.row0; info[3] = D.nloc;
    info[4] = (int)D.ghosts.size(); info[5] = (int)D.send_idx.size();
    info[6] = h->dist.first_replicated; info[7] = h->dist.nranks;
    return FASP_SUCCESS;
}

int fasp_hip_dist_get_matrix(const fasp_hip_amg* h, int level, int which, dCSRmat* view)
{
    if (!h || !view || level < 0 || level >= (int)h->dist.L.size()) return ERROR_INPUT_PAR;
    const DistLevel& D = h->dist.L[level];
    if (D.replicated) return fasp_hip_amg_get_matrix(h, level, which, view);
    const HostCSR& M = which == 0 ? D.A : which == 1 ? D.P : D.R;
    if (!M.ia.data()) return ERROR_INPUT_PAR;
    *view = M.view();
    return FASP_SUCCESS;
}

int fasp_hip_dist_get_list(const fasp_hip_amg* h, int level, int which, ivector* view)
{
    if (!h || !view || level < 0 || level >= (int)h->dist.L.size()) return ERROR_INPUT_PAR;
    const DistLevel& D = h->dist.L[level];
    const std::vector<int>* v = which == 0 ? &D.ghosts : which == 1 ? &D.recv_off : which == 2 ? &D.send_off
                              : which == 3 ? &D.send_idx : which == 4 ? &D.start : nullptr;
    if (!v) return ERROR_INPUT_PAR;
    view->row = (int)v->size();
    view->val = const_cast<int*>(v->data());
    return FASP_SUCCESS;
}

// SolCSR.c:476 -- the drop-in entry point
int fasp_solver_dcsr_krylov_amg(dCSRmat* A, dvector* b, dvector* x, ITS_param* itparam, AMG_param* amgparam)
{
    if (!A || !b || !x || !itparam || !amgparam) return ERROR_INPUT_PAR;
    const double t0 = wall_seconds();
    int st = check_supported(itparam, amgparam);
    if (st < 0) return st;
    fasp_hip_amg* h = nullptr;
    g_oneshot_upload = std::getenv("FASP_HIP_ONESHOT_CODING") == nullptr;  // (set the variable to keep the coding)
    st = fasp_hip_amg_create(&h, A, amgparam);
    g_oneshot_upload = false;
    if (st < 0) return st;
    st = fasp_hip_solve(h, b, x, itparam, nullptr, 0, nullptr);
    if (itparam->print_level >= PRINT_MIN)
        std::printf("AMG_Krylov method totally costs %.4f seconds.\n", wall_seconds() - t0);
    fasp_hip_amg_destroy(h);
    return st;
}

// ---------------------------------------------------------------------------
// BSR path: resident hierarchy + the drop-in of SolBSR.c:349
// ---------------------------------------------------------------------------
void fasp_hip_bsr_amg_destroy(fasp_hip_amg_bsr* h)
{
    if (!h) return;
    if (g_ctx.ready) (void)hipStreamSynchronize(g_ctx.stream);
    for (auto& Lv : h->L) {
        double* v[] = {Lv.dinv, Lv.b, Lv.x, Lv.x2, Lv.w};
        for (double* q : v) if (q) (void)hipFree(q);
        for (auto& sc : Lv.sched) if (sc.d_order) (void)hipFree(sc.d_order);
    }
    double* v[] = {h->b, h->u, h->p, h->t, h->r};
    for (double* q : v) if (q) (void)hipFree(q);
    for (int s = 0; s < 2; ++s)
        for (double* q : h->gm[s]) if (q) (void)hipFree(q);
    if (h->gm_hh) (void)hipFree(h->gm_hh);
    if (h->small_ws) (void)hipFree(h->small_ws);
    delete h;
}

// host part only (no GPU needed): the hierarchy can be inspected, not solved with
int fasp_hip_bsr_amg_create_host(fasp_hip_amg_bsr** out, const dBSRmat* A, AMG_param* amgparam)
{
    if (!out || !A || !amgparam) return ERROR_INPUT_PAR;
    *out = nullptr;
    int st = check_supported_bsr(nullptr, amgparam, A->nb);
    if (st < 0) return st;
    fasp_hip_amg_bsr* h = new fasp_hip_amg_bsr();
    st = host_setup_ua_bsr(A, amgparam, h->H);
    if (st < 0) { delete h; return st; }
    h->param = *amgparam;
    *out = h;
    return FASP_SUCCESS;
}

int fasp_hip_bsr_amg_create(fasp_hip_amg_bsr** out, const dBSRmat* A, AMG_param* amgparam)
{
    if (!out || !A || !amgparam) return ERROR_INPUT_PAR;
    *out = nullptr;
    int st = check_supported_bsr(nullptr, amgparam, A->nb);
    if (st < 0) return st;
    if ((st = ctx_init()) < 0) return st;
    fasp_hip_amg_bsr* h = nullptr;
    if ((st = fasp_hip_bsr_amg_create_host(&h, A, amgparam)) < 0) return st;
    const int nl = (int)h->H.L.size();
    h->L.resize(nl);
    for (int l = 0; l < nl; ++l) {
        const HostLevelBSR& HL = h->H.L[l];
        BsrLevel& Lv = h->L[l];
        const dBSRmat vA = HL.A.view();
        Lv.A.reset(new TmpBSR(&vA));
        bool ok = Lv.A->ok;
        Lv.n = HL.A.ROW * HL.A.nb;
        if (HL.has_coarse) {
            const dBSRmat vP = HL.P.view(), vR = HL.R.view();
            Lv.P.reset(new TmpBSR(&vP));
            Lv.R.reset(new TmpBSR(&vR));
            ok = ok && Lv.P->ok && Lv.R->ok;
            const size_t nd = (size_t)HL.A.ROW * HL.A.nb * HL.A.nb;
            if (hipMalloc(&Lv.dinv, sizeof(double) * std::max<size_t>(nd, 1)) != hipSuccess) ok = false;
            else (void)hipMemcpy(Lv.dinv, HL.diaginv.data(), sizeof(double) * nd, hipMemcpyHostToDevice);
        }
        if (!ok || dalloc(&Lv.b, Lv.n) < 0 || dalloc(&Lv.x, Lv.n) < 0 || dalloc(&Lv.x2, Lv.n) < 0 ||
            dalloc(&Lv.w, Lv.n) < 0) {
            fasp_hip_bsr_amg_destroy(h);
            return ERROR_ALLOC_MEM;
        }
    }
    const size_t n0 = (size_t)h->L[0].n;
    if (dalloc(&h->b, n0) < 0 || dalloc(&h->u, n0) < 0 || dalloc(&h->p, n0) < 0 || dalloc(&h->t, n0) < 0 ||
        dalloc(&h->r, n0) < 0) {
        fasp_hip_bsr_amg_destroy(h);
        return ERROR_ALLOC_MEM;
    }
    HIPCK(hipStreamSynchronize(g_ctx.stream));
    *out = h;
    return FASP_SUCCESS;
}

int fasp_hip_bsr_amg_num_levels(const fasp_hip_amg_bsr* h) { return h ? (int)h->H.L.size() : ERROR_INPUT_PAR; }

int fasp_hip_bsr_amg_get_matrix(const fasp_hip_amg_bsr* h, int level, int which, dBSRmat* view)
{
    if (!h || !view || level < 0 || level >= (int)h->H.L.size()) return ERROR_INPUT_PAR;
    const HostLevelBSR& L = h->H.L[level];
    if (which != 0 && !L.has_coarse) return ERROR_INPUT_PAR;
    *view = which == 0 ? L.A.view() : which == 1 ? L.P.view() : L.R.view();
    return FASP_SUCCESS;
}

const double* fasp_hip_bsr_amg_get_diaginv(const fasp_hip_amg_bsr* h, int level)
{
    if (!h || level < 0 || level >= (int)h->H.L.size() || !h->H.L[level].has_coarse) return nullptr;
    return h->H.L[level].diaginv.data();
}

int fasp_hip_bsr_solve(fasp_hip_amg_bsr* h, const dvector* b, dvector* x, const ITS_param* itparam, double* hist,
                       int hist_cap, fasp_hip_stats* stats)
{
    if (!h || !b || !x || !itparam || h->L.empty()) return ERROR_INPUT_PAR;
    const int n = h->L[0].n;
    if (b->row != n || x->row != n) return ERROR_MAT_SIZE;
    int st = check_supported_bsr(itparam, &h->param, h->H.L[0].A.nb);
    if (st < 0) return st;
    if (itparam->tol < SMALLREAL)
        std::printf("### WARNING: Convergence tolerance is too small! [%s:%d]\n", "ITS_CHECK", 74);
    if (itparam->maxit <= 0)
        std::printf("### WARNING: Max number of iterations must be POSITIVE! [%s:%d]\n", "ITS_CHECK", 78);
    hipStream_t s = g_ctx.stream;
    double t0 = wall_seconds();
    HIPCK(hipMemcpyAsync(h->b, b->val, sizeof(double) * n, hipMemcpyHostToDevice, s));
    HIPCK(hipMemcpyAsync(h->u, x->val, sizeof(double) * n, hipMemcpyHostToDevice, s));
    HIPCK(hipStreamSynchronize(s));
    double t_up = wall_seconds() - t0;

    const long long ci0 = h->coarse_iters, vc0 = h->vcycles;
    Hist   H{hist, hist_cap, 0};
    PcgOut po{BIGREAL, BIGREAL, BIGREAL};
    t0 = wall_seconds();
    KOps K = bsr_ops(h, 0, 0);
    switch (itparam->itsolver_type) {  // fasp_solver_dbsr_itsolver, SolBSR.c:55-150
        case SOLVER_BiCGstab:
            st = bicgstab_device(K, h->b, h->u, itparam->tol, itparam->maxit, itparam->print_level, &H, &po);
            break;
        case SOLVER_GMRES:
        case SOLVER_VGMRES:
        case SOLVER_VFGMRES:
            st = gmres_device(K, h->b, h->u, itparam->itsolver_type == SOLVER_VFGMRES ? 1 : itparam->itsolver_type == SOLVER_GMRES ? 3 : 0, itparam->tol,
                              itparam->abstol, itparam->maxit, (short)itparam->restart, itparam->stop_type,
                              itparam->print_level, &H, &po);
            break;
        default:
        {
            PcgVecs V{h->b, h->u, h->p, h->t, h->r};
            st = pcg_device(K, V, itparam->tol, itparam->abstol, itparam->maxit, itparam->stop_type,
                            itparam->print_level, H, po);
        }
    }
    HIPCK(hipStreamSynchronize(s));
    const double t_solve = wall_seconds() - t0;
    t0 = wall_seconds();
    HIPCK(hipMemcpy(x->val, h->u, sizeof(double) * n, hipMemcpyDeviceToHost));
    t_up += wall_seconds() - t0;
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->iters = st; stats->nhist = H.n; stats->relres = po.relres; stats->absres = po.absres;
        stats->normr0 = po.normr0; stats->solve_seconds = t_solve; stats->upload_seconds = t_up;
        stats->coarse_iters = h->coarse_iters - ci0;
        stats->vcycles = h->vcycles - vc0;
    }
    if (itparam->print_level >= PRINT_SOME && st >= 0)
        std::printf("Iterative method costs %.4f seconds.\n", t_solve);
    return st;
}

// SolBSR.c:349: UA-AMG setup on the host, hierarchy uploaded, Krylov loop on the device
int fasp_solver_dbsr_krylov_amg(dBSRmat* A, dvector* b, dvector* x, ITS_param* itparam, AMG_param* amgparam)
{
    if (!A || !b || !x || !itparam || !amgparam) return ERROR_INPUT_PAR;
    const double t0 = wall_seconds();
    int st = check_supported_bsr(itparam, amgparam, A->nb);
    if (st < 0) return st;
    fasp_hip_amg_bsr* h = nullptr;
    st = fasp_hip_bsr_amg_create(&h, A, amgparam);
    if (st < 0) return st;
    st = fasp_hip_bsr_solve(h, b, x, itparam, nullptr, 0, nullptr);
    if (itparam->print_level >= PRINT_MIN)
        std::printf("AMG_Krylov method totally costs %.4f seconds.\n", wall_seconds() - t0);
    fasp_hip_bsr_amg_destroy(h);
    return st;
}

// ---------------------------------------------------------------------------
// kernel-level operators with host pointers (reference names, device kernels)
// ---------------------------------------------------------------------------
namespace {
struct TmpCSR {
    DevCSR D;
    bool   ok = false;
    explicit TmpCSR(const dCSRmat* A)
    {
        if (ctx_init() < 0) return;
        D.row = A->row; D.col = A->col; D.nnz = A->nnz;
        if (hipMalloc(&D.ia, sizeof(int) * ((size_t)A->row + 1)) != hipSuccess) return;
        if (hipMalloc(&D.ja, sizeof(int) * std::max(A->nnz, 1)) != hipSuccess) return;
        if (hipMalloc(&D.val, sizeof(double) * std::max(A->nnz, 1)) != hipSuccess) return;
        (void)hipMemcpy(D.ia, A->IA, sizeof(int) * ((size_t)A->row + 1), hipMemcpyHostToDevice);
        (void)hipMemcpy(D.ja, A->JA, sizeof(int) * (size_t)A->nnz, hipMemcpyHostToDevice);
        (void)hipMemcpy(D.val, A->val, sizeof(double) * (size_t)A->nnz, hipMemcpyHostToDevice);
        pick_kernel(D);
        ok = true;
    }
    ~TmpCSR() { D.release(); }
};
struct TmpVec {
    double* d = nullptr;
    size_t  n;
    TmpVec(const double* h, size_t n_) : n(n_)
    {
        if (hipMalloc(&d, sizeof(double) * std::max<size_t>(n, 1)) != hipSuccess) { d = nullptr; return; }
        if (h) (void)hipMemcpy(d, h, sizeof(double) * n, hipMemcpyHostToDevice);
    }
    void get(double* h) { (void)hipStreamSynchronize(g_ctx.stream); (void)hipMemcpy(h, d, sizeof(double) * n, hipMemcpyDeviceToHost); }
    ~TmpVec() { if (d) (void)hipFree(d); }
};
[[noreturn]] void die_no_device(const char* fn)
{
    std::fprintf(stderr, "### ERROR: %s: no usable HIP device and no CPU fallback in libfasp_hip\n", fn);
    std::exit(ERROR_MISC);
}
}  // namespace

// ---------------------------------------------------------------------------
// plug-in level of the reference: the Krylov methods with a caller
// supplied `precond` (fasp.h:1095), and the AMG preconditioner as such a plug-in
// ---------------------------------------------------------------------------
// PreCSR.c:416 signature: z = B r, host vectors; data is the fasp_hip_amg* of fasp_hip_precond_setup
void fasp_hip_precond_fct(double* r, double* z, void* data)
{
    fasp_hip_amg* h = static_cast<fasp_hip_amg*>(data);
    if (fasp_hip_precond_amg(h, r, z) < 0) {
        std::fprintf(stderr, "### ERROR: fasp_hip_precond_fct: device preconditioner failed\n");
        std::exit(ERROR_MISC);
    }
}

// PreCSR.c:46 for PREC_AMG: hierarchy built and uploaded once, handed out as a `precond`
precond* fasp_hip_precond_setup(dCSRmat* A, AMG_param* amgparam)
{
    fasp_hip_amg* h = nullptr;
    if (fasp_hip_amg_create(&h, A, amgparam) < 0) return nullptr;
    precond* pc = static_cast<precond*>(std::calloc(1, sizeof(precond)));
    pc->data = h;
    pc->fct = fasp_hip_precond_fct;
    return pc;
}

void fasp_hip_precond_free(precond* pc)
{
    if (!pc) return;
    if (pc->fct == fasp_hip_precond_fct) fasp_hip_amg_destroy(static_cast<fasp_hip_amg*>(pc->data));
    std::free(pc);
}

namespace {
bool same_host_matrix(const HostCSR& M, const dCSRmat* A)
{
    return M.row == A->row && M.col == A->col && M.nnz == A->nnz &&
           std::memcmp(M.ia.data(), A->IA, sizeof(int) * ((size_t)A->row + 1)) == 0 &&
           std::memcmp(M.ja.data(), A->JA, sizeof(int) * (size_t)A->nnz) == 0 &&
           std::memcmp(M.val.data(), A->val, sizeof(double) * (size_t)A->nnz) == 0;
}

// which: 0 PCG, 1 VGMRES, 2 VFGMRES, 3 BiCGstab, 4 GMRES (fixed restart), 5 MinRes, 6 GCG, 7 GCR
int krylov_plugin(const char* fn, int which, dCSRmat* A, dvector* b, dvector* u, precond* pc, double tol,
                  double abstol, int MaxIt, short restart, short StopType, short PrtLvl)
{
    if (ctx_init() < 0) die_no_device(fn);
    if (!A || !b || !u || A->row != A->col || b->row != A->row || u->row != A->row) return ERROR_INPUT_PAR;
    if (comm_size() > 1) return ERROR_INPUT_PAR;  // plug-in level: one GPU
    const int n = b->row;
    fasp_hip_amg* h = (pc && pc->fct == fasp_hip_precond_fct) ? static_cast<fasp_hip_amg*>(pc->data) : nullptr;
    if (h && (h->L.empty() || h->L[0].A.row != n)) return ERROR_INPUT_PAR;
    std::unique_ptr<TmpCSR> own;
    const DevCSR* dA = nullptr;
    if (h && same_host_matrix(h->H.L[0].A, A)) dA = &h->L[0].A;  // the resident level-0 operator is A itself
    else {
        own.reset(new TmpCSR(A));
        if (!own->ok) return ERROR_ALLOC_MEM;
        dA = &own->D;
    }
    TmpVec db(b->val, n), du(u->val, n), dp(nullptr, n), dt(nullptr, n), dr(nullptr, n), dz(nullptr, n);
    if (!db.d || !du.d || !dp.d || !dt.d || !dr.d || !dz.d) return ERROR_ALLOC_MEM;
    std::vector<double> hr, hz;
    std::vector<double*> ws;
    size_t ws_len = 0;
    double* hh = nullptr;
    KOps K;
    K.n = n; K.nvec = (size_t)n; K.fmt = "CSR"; K.dist = false;
    K.halo = [](double*) { return 0; };
    K.mxv = [dA](const double* x, double* y) { d_mxv(*dA, x, y); };
    K.resid = [dA](const double* x, const double* bb, double* r) { d_resid(*dA, x, bb, r); };
    K.mxv_dot = [dA](const double* x, double* y) {
        CsrArgs a{}; a.x = x; a.y = y; a.dotv = x; a.partials = g_ctx.d_partials;
        return launch_csr<OP_MXV_DOT>(*dA, a);
    };
    std::unique_ptr<TmpVec> ddiag;
    if (h) {
        K.pc = [h](double* in, double** out) { return precond_amg(h, in, out); };  // stays in HBM
    } else if (pc && pc->fct == fasp_precond_diag && pc->data && static_cast<dvector*>(pc->data)->row == n) {
        // the reference's diagonal preconditioner: recognised by its function pointer, applied on the device
        ddiag.reset(new TmpVec(static_cast<dvector*>(pc->data)->val, n));
        if (!ddiag->d) return ERROR_ALLOC_MEM;
        const double* dd = ddiag->d;
        double* zz = dz.d;
        K.pc = [dd, zz, n](double* in, double** out) {
            hipLaunchKernelGGL(k_diag_precond, dim3(vec_grid(n)), dim3(BLOCK), 0, g_ctx.stream, n, dd, (const double*)in, zz);
            *out = zz;
            return 0;
        };
    } else if (pc && pc->fct) {
        // foreign preconditioner: a host function; the residual is staged through host memory
        hr.resize((size_t)n); hz.resize((size_t)n);
        K.pc = [&, pc](double* in, double** out) {
            HIPCK(hipMemcpyAsync(hr.data(), in, sizeof(double) * n, hipMemcpyDeviceToHost, g_ctx.stream));
            HIPCK(hipStreamSynchronize(g_ctx.stream));
            pc->fct(hr.data(), hz.data(), pc->data);
            HIPCK(hipMemcpyAsync(dz.d, hz.data(), sizeof(double) * n, hipMemcpyHostToDevice, g_ctx.stream));
            *out = dz.d;
            return 0;
        };
    }
    K.ws = &ws; K.ws_len = &ws_len; K.hh = &hh;
    K.stats = nullptr;
    Hist   H{nullptr, 0, 0};
    PcgOut po{BIGREAL, BIGREAL, BIGREAL};
    int st;
    if (which == 0) {
        PcgVecs V{db.d, du.d, dp.d, dt.d, dr.d};
        st = pcg_device(K, V, tol, abstol, MaxIt, StopType, PrtLvl, H, po);
    } else if (which == 3) {
        st = bicgstab_device(K, db.d, du.d, tol, MaxIt, PrtLvl, &H, &po);
    } else if (which == 5) {
        st = minres_device(K, db.d, du.d, tol, abstol, MaxIt, StopType, PrtLvl, &H, &po);
    } else if (which == 6) {
        st = gcg_device(K, db.d, du.d, tol, abstol, MaxIt, StopType, PrtLvl, &H, &po);
    } else if (which == 7) {
        st = gcr_device(K, db.d, du.d, tol, abstol, MaxIt, restart, StopType, PrtLvl, &H, &po);
    } else {
        st = gmres_device(K, db.d, du.d, which == 2 ? 1 : which == 4 ? 3 : 0, tol, abstol, MaxIt, restart, StopType, PrtLvl, &H, &po);
    }
    du.get(u->val);
    for (double* q : ws) if (q) (void)hipFree(q);
    if (hh) (void)hipFree(hh);
    return st;
}
}  // namespace

// KryPcg.c:96
int fasp_solver_dcsr_pcg(dCSRmat* A, dvector* b, dvector* u, precond* pc, const double tol, const double abstol,
                         const int MaxIt, const short StopType, const short PrtLvl)
{
    return krylov_plugin(__func__, 0, A, b, u, pc, tol, abstol, MaxIt, 0, StopType, PrtLvl);
}
// KryPgmres.c:66
int fasp_solver_dcsr_pgmres(dCSRmat* A, dvector* b, dvector* x, precond* pc, const double tol, const double abstol,
                            const int MaxIt, const short restart, const short StopType, const short PrtLvl)
{
    return krylov_plugin(__func__, 4, A, b, x, pc, tol, abstol, MaxIt, restart, StopType, PrtLvl);
}
// KryPvgmres.c:66
int fasp_solver_dcsr_pvgmres(dCSRmat* A, dvector* b, dvector* x, precond* pc, const double tol, const double abstol,
                             const int MaxIt, const short restart, const short StopType, const short PrtLvl)
{
    return krylov_plugin(__func__, 1, A, b, x, pc, tol, abstol, MaxIt, restart, StopType, PrtLvl);
}
// KryPbcgs.c:62
int fasp_solver_dcsr_pbcgs(dCSRmat* A, dvector* b, dvector* u, precond* pc, const double tol, const double abstol,
                           const int MaxIt, const short StopType, const short PrtLvl)
{
    return krylov_plugin(__func__, 3, A, b, u, pc, tol, abstol, MaxIt, 0, StopType, PrtLvl);
}
// KryPminres.c:61
int fasp_solver_dcsr_pminres(dCSRmat* A, dvector* b, dvector* u, precond* pc, const double tol, const double abstol,
                             const int MaxIt, const short StopType, const short PrtLvl)
{
    return krylov_plugin(__func__, 5, A, b, u, pc, tol, abstol, MaxIt, 0, StopType, PrtLvl);
}
// KryPgcg.c:60
int fasp_solver_dcsr_pgcg(dCSRmat* A, dvector* b, dvector* u, precond* pc, const double tol, const double abstol,
                          const int MaxIt, const short StopType, const short PrtLvl)
{
    return krylov_plugin(__func__, 6, A, b, u, pc, tol, abstol, MaxIt, 0, StopType, PrtLvl);
}
// KryPgcr.c:55
int fasp_solver_dcsr_pgcr(dCSRmat* A, dvector* b, dvector* x, precond* pc, const double tol, const double abstol,
                          const int MaxIt, const short restart, const short StopType, const short PrtLvl)
{
    return krylov_plugin(__func__, 7, A, b, x, pc, tol, abstol, MaxIt, restart, StopType, PrtLvl);
}
// KryPvfgmres.c:67
int fasp_solver_dcsr_pvfgmres(dCSRmat* A, dvector* b, dvector* x, precond* pc, const double tol, const double abstol,
                              const int MaxIt, const short restart, const short StopType, const short PrtLvl)
{
    return krylov_plugin(__func__, 2, A, b, x, pc, tol, abstol, MaxIt, restart, StopType, PrtLvl);
}

// ---- the same plug-in level for block matrices ---------------------------------
// z = B r with the resident block hierarchy (PreBSR.c:1149 signature); data = fasp_hip_amg_bsr*
void fasp_hip_bsr_precond_fct(double* r, double* z, void* data)
{
    fasp_hip_amg_bsr* h = static_cast<fasp_hip_amg_bsr*>(data);
    if (!h || h->L.empty() || ctx_init() < 0) die_bsr(__func__);
    const int n = h->L[0].n;
    TmpVec dr(r, (size_t)n);
    double* dz = nullptr;
    if (!dr.d || precond_amg_bsr(h, dr.d, &dz) < 0) die_bsr(__func__);
    (void)hipStreamSynchronize(g_ctx.stream);
    (void)hipMemcpy(z, dz, sizeof(double) * n, hipMemcpyDeviceToHost);
}

precond* fasp_hip_bsr_precond_setup(dBSRmat* A, AMG_param* amgparam)
{
    fasp_hip_amg_bsr* h = nullptr;
    if (fasp_hip_bsr_amg_create(&h, A, amgparam) < 0) return nullptr;
    precond* pc = static_cast<precond*>(std::calloc(1, sizeof(precond)));
    pc->data = h;
    pc->fct = fasp_hip_bsr_precond_fct;
    return pc;
}

void fasp_hip_bsr_precond_free(precond* pc)
{
    if (!pc) return;
    if (pc->fct == fasp_hip_bsr_precond_fct) fasp_hip_bsr_amg_destroy(static_cast<fasp_hip_amg_bsr*>(pc->data));
    std::free(pc);
}

namespace {
int krylov_plugin_bsr(const char* fn, int which, dBSRmat* A, dvector* b, dvector* u, precond* pc, double tol,
                      double abstol, int MaxIt, short restart, short StopType, short PrtLvl)
{
    if (!A || !b || !u || A->ROW != A->COL || b->row != A->ROW * A->nb || u->row != b->row) return ERROR_INPUT_PAR;
    TmpBSR M(A);
    if (!M.ok) die_bsr(fn);
    const int n = b->row;
    fasp_hip_amg_bsr* h = (pc && pc->fct == fasp_hip_bsr_precond_fct) ? static_cast<fasp_hip_amg_bsr*>(pc->data) : nullptr;
    if (h && (h->L.empty() || h->L[0].n != n)) return ERROR_INPUT_PAR;
    TmpVec db(b->val, n), du(u->val, n), dp(nullptr, n), dt(nullptr, n), dr(nullptr, n), dz(nullptr, n);
    if (!db.d || !du.d || !dp.d || !dt.d || !dr.d || !dz.d) return ERROR_ALLOC_MEM;
    std::vector<double> hr, hz;
    std::vector<double*> ws;
    size_t ws_len = 0;
    double* hh = nullptr;
    KOps K;
    K.n = n; K.nvec = (size_t)n; K.fmt = "BSR"; K.dist = false;
    K.halo = [](double*) { return 0; };
    const TmpBSR* Mp = &M;
    K.mxv = [Mp](const double* x, double* y) { bsr_mxv(*Mp, x, y); };
    K.resid = [Mp](const double* x, const double* bb, double* r) { bsr_resid(*Mp, x, bb, r); };
    std::unique_ptr<TmpVec> ddiag;
    if (h) {
        K.pc = [h](double* in, double** out) { return precond_amg_bsr(h, in, out); };
    } else if (pc && pc->fct == fasp_precond_dbsr_diag && pc->data &&
               static_cast<precond_diag_bsr*>(pc->data)->diag.row == A->ROW * A->nb * A->nb) {
        // block-diagonal preconditioner of the reference (PreBSR.c:49): z_i = Dinv_i r_i on the device
        ddiag.reset(new TmpVec(static_cast<precond_diag_bsr*>(pc->data)->diag.val, (size_t)A->ROW * A->nb * A->nb));
        if (!ddiag->d) return ERROR_ALLOC_MEM;
        const double* dd = ddiag->d;
        double* zz = dz.d;
        const int nb = A->nb;
        K.pc = [dd, zz, n, nb](double* in, double** out) {
            hipLaunchKernelGGL(k_bsr_dinv_apply, dim3(vec_grid(n)), dim3(BLOCK), 0, g_ctx.stream, n, nb, dd, (const double*)in, zz);
            *out = zz;
            return 0;
        };
    } else if (pc && pc->fct) {
        hr.resize((size_t)n); hz.resize((size_t)n);
        K.pc = [&, pc](double* in, double** out) {
            HIPCK(hipMemcpyAsync(hr.data(), in, sizeof(double) * n, hipMemcpyDeviceToHost, g_ctx.stream));
            HIPCK(hipStreamSynchronize(g_ctx.stream));
            pc->fct(hr.data(), hz.data(), pc->data);
            HIPCK(hipMemcpyAsync(dz.d, hz.data(), sizeof(double) * n, hipMemcpyHostToDevice, g_ctx.stream));
            *out = dz.d;
            return 0;
        };
    }
    K.ws = &ws; K.ws_len = &ws_len; K.hh = &hh;
    Hist   H{nullptr, 0, 0};
    PcgOut po{BIGREAL, BIGREAL, BIGREAL};
    int st;
    if (which == 0) {
        PcgVecs V{db.d, du.d, dp.d, dt.d, dr.d};
        st = pcg_device(K, V, tol, abstol, MaxIt, StopType, PrtLvl, H, po);
    } else if (which == 3) {
        st = bicgstab_device(K, db.d, du.d, tol, MaxIt, PrtLvl, &H, &po);
    } else {
        st = gmres_device(K, db.d, du.d, which == 2 ? 1 : which == 4 ? 3 : 0, tol, abstol, MaxIt, restart, StopType, PrtLvl, &H, &po);
    }
    du.get(u->val);
    for (double* q : ws) if (q) (void)hipFree(q);
    if (hh) (void)hipFree(hh);
    return st;
}
}  // namespace

// KryPcg.c:386, KryPbcgs.c:400, KryPgmres.c:357, KryPvgmres.c:416, KryPvfgmres.c:386
int fasp_solver_dbsr_pcg(dBSRmat* A, dvector* b, dvector* u, precond* pc, const double tol, const double abstol,
                         const int MaxIt, const short StopType, const short PrtLvl)
{
    return krylov_plugin_bsr(__func__, 0, A, b, u, pc, tol, abstol, MaxIt, 0, StopType, PrtLvl);
}
int fasp_solver_dbsr_pbcgs(dBSRmat* A, dvector* b, dvector* u, precond* pc, const double tol, const double abstol,
                           const int MaxIt, const short StopType, const short PrtLvl)
{
    return krylov_plugin_bsr(__func__, 3, A, b, u, pc, tol, abstol, MaxIt, 0, StopType, PrtLvl);
}
int fasp_solver_dbsr_pgmres(dBSRmat* A, dvector* b, dvector* x, precond* pc, const double tol, const double abstol,
                            const int MaxIt, const short restart, const short StopType, const short PrtLvl)
{
    return krylov_plugin_bsr(__func__, 4, A, b, x, pc, tol, abstol, MaxIt, restart, StopType, PrtLvl);
}
int fasp_solver_dbsr_pvgmres(dBSRmat* A, dvector* b, dvector* x, precond* pc, const double tol, const double abstol,
                             const int MaxIt, const short restart, const short StopType, const short PrtLvl)
{
    return krylov_plugin_bsr(__func__, 1, A, b, x, pc, tol, abstol, MaxIt, restart, StopType, PrtLvl);
}
int fasp_solver_dbsr_pvfgmres(dBSRmat* A, dvector* b, dvector* x, precond* pc, const double tol, const double abstol,
                              const int MaxIt, const short restart, const short StopType, const short PrtLvl)
{
    return krylov_plugin_bsr(__func__, 2, A, b, x, pc, tol, abstol, MaxIt, restart, StopType, PrtLvl);
}

// ---------------------------------------------------------------------------
// The other solver-level entry points of SolCSR.c / SolBSR.c: dispatch on itsolver_type with a caller's
// preconditioner, no preconditioner, or the (block-)diagonal one.
// ---------------------------------------------------------------------------
void fasp_precond_diag(double* r, double* z, void* data)  // PreCSR.c:172 (host arrays)
{
    const dvector* diag = static_cast<const dvector*>(data);
    std::memcpy(z, r, sizeof(double) * (size_t)diag->row);
    for (int i = 0; i < diag->row; ++i)
        if (std::fabs(diag->val[i]) > SMALLREAL) z[i] /= diag->val[i];
}
void fasp_precond_dbsr_diag(double* r, double* z, void* data)  // PreBSR.c:49 (host arrays): z_i = Dinv_i r_i
{
    const precond_diag_bsr* d = static_cast<const precond_diag_bsr*>(data);
    const int nb = d->nb, nb2 = nb * nb, m = d->diag.row / nb2;
    for (int i = 0; i < m; ++i)
        for (int rr = 0; rr < nb; ++rr) {
            const double* D = d->diag.val + (size_t)i * nb2 + rr * nb;
            double s = D[0] * r[(size_t)i * nb];
            for (int c = 1; c < nb; ++c) s = s + D[c] * r[(size_t)i * nb + c];
            z[(size_t)i * nb + rr] = s;
        }
}

// SolCSR.c:56
int fasp_solver_dcsr_itsolver(dCSRmat* A, dvector* b, dvector* x, precond* pc, ITS_param* itparam)
{
    if (!itparam) return ERROR_INPUT_PAR;
    const short prtlvl = itparam->print_level, stop_type = itparam->stop_type, restart = (short)itparam->restart;
    const int MaxIt = itparam->maxit;
    const double tol = itparam->tol, abstol = itparam->abstol, t0 = wall_seconds();
    int iter;
    if (tol < SMALLREAL) std::printf("### WARNING: Convergence tolerance is too small! [%s:%d]\n", "ITS_CHECK", 74);
    if (MaxIt <= 0) std::printf("### WARNING: Max number of iterations must be POSITIVE! [%s:%d]\n", "ITS_CHECK", 78);
    switch (itparam->itsolver_type) {
        case SOLVER_CG: iter = fasp_solver_dcsr_pcg(A, b, x, pc, tol, abstol, MaxIt, stop_type, prtlvl); break;
        case SOLVER_BiCGstab: iter = fasp_solver_dcsr_pbcgs(A, b, x, pc, tol, abstol, MaxIt, stop_type, prtlvl); break;
        case SOLVER_MinRes: iter = fasp_solver_dcsr_pminres(A, b, x, pc, tol, abstol, MaxIt, stop_type, prtlvl); break;
        case SOLVER_GMRES: iter = fasp_solver_dcsr_pgmres(A, b, x, pc, tol, abstol, MaxIt, restart, stop_type, prtlvl); break;
        case SOLVER_VGMRES: iter = fasp_solver_dcsr_pvgmres(A, b, x, pc, tol, abstol, MaxIt, restart, stop_type, prtlvl); break;
        case SOLVER_VFGMRES: iter = fasp_solver_dcsr_pvfgmres(A, b, x, pc, tol, abstol, MaxIt, restart, stop_type, prtlvl); break;
        case SOLVER_GCG: iter = fasp_solver_dcsr_pgcg(A, b, x, pc, tol, abstol, MaxIt, stop_type, prtlvl); break;
        case SOLVER_GCR: iter = fasp_solver_dcsr_pgcr(A, b, x, pc, tol, abstol, MaxIt, restart, stop_type, prtlvl); break;
        default:
            std::printf("### ERROR: Unknown iterative solver type %d! [%s]\n", itparam->itsolver_type, __func__);
            return ERROR_SOLVER_TYPE;
    }
    if ((prtlvl >= PRINT_SOME) && (iter >= 0)) std::printf("Iterative method costs %.4f seconds.\n", wall_seconds() - t0);
    return iter;
}
// SolCSR.c:245
int fasp_solver_dcsr_krylov(dCSRmat* A, dvector* b, dvector* x, ITS_param* itparam)
{
    if (!itparam) return ERROR_INPUT_PAR;
    const double t0 = wall_seconds();
    const int status = fasp_solver_dcsr_itsolver(A, b, x, nullptr, itparam);
    if (itparam->print_level >= PRINT_MIN) std::printf("Krylov method totally costs %.4f seconds.\n", wall_seconds() - t0);
    return status;
}
// SolCSR.c:333: diagonal preconditioner from fasp_dcsr_getdiag(0, A, ..) -- the FIRST diagonal hit of each row
int fasp_solver_dcsr_krylov_diag(dCSRmat* A, dvector* b, dvector* x, ITS_param* itparam)
{
    if (!A || !itparam || !A->IA || !A->JA || !A->val) return ERROR_INPUT_PAR;
    const double t0 = wall_seconds();
    const int n = std::min(A->row, A->col);
    std::vector<double> dv((size_t)std::max(n, 1), 0.0);
    for (int i = 0; i < n; ++i)
        for (int k = A->IA[i]; k < A->IA[i + 1]; ++k)
            if (A->JA[k] == i) { dv[(size_t)i] = A->val[k]; break; }
    dvector diag{n, dv.data()};
    precond pc{&diag, fasp_precond_diag};
    const int status = fasp_solver_dcsr_itsolver(A, b, x, &pc, itparam);
    if (itparam->print_level >= PRINT_MIN) std::printf("Diag_Krylov method totally costs %.4f seconds.\n", wall_seconds() - t0);
    return status;
}
// SolBSR.c:64
int fasp_solver_dbsr_itsolver(dBSRmat* A, dvector* b, dvector* x, precond* pc, ITS_param* itparam)
{
    if (!itparam) return ERROR_INPUT_PAR;
    const short prtlvl = itparam->print_level, stop_type = itparam->stop_type, restart = (short)itparam->restart;
    const int MaxIt = itparam->maxit;
    const double tol = itparam->tol, abstol = itparam->abstol, t0 = wall_seconds();
    int iter;
    if (tol < SMALLREAL) std::printf("### WARNING: Convergence tolerance is too small! [%s:%d]\n", "ITS_CHECK", 74);
    if (MaxIt <= 0) std::printf("### WARNING: Max number of iterations must be POSITIVE! [%s:%d]\n", "ITS_CHECK", 78);
    switch (itparam->itsolver_type) {
        case SOLVER_CG: iter = fasp_solver_dbsr_pcg(A, b, x, pc, tol, abstol, MaxIt, stop_type, prtlvl); break;
        case SOLVER_BiCGstab: iter = fasp_solver_dbsr_pbcgs(A, b, x, pc, tol, abstol, MaxIt, stop_type, prtlvl); break;
        case SOLVER_GMRES: iter = fasp_solver_dbsr_pgmres(A, b, x, pc, tol, abstol, MaxIt, restart, stop_type, prtlvl); break;
        case SOLVER_VGMRES: iter = fasp_solver_dbsr_pvgmres(A, b, x, pc, tol, abstol, MaxIt, restart, stop_type, prtlvl); break;
        case SOLVER_VFGMRES: iter = fasp_solver_dbsr_pvfgmres(A, b, x, pc, tol, abstol, MaxIt, restart, stop_type, prtlvl); break;
        default:
            std::printf("### ERROR: Unknown iterative solver type %d! [%s]\n", itparam->itsolver_type, __func__);
            return ERROR_SOLVER_TYPE;
    }
    if ((prtlvl >= PRINT_SOME) && (iter >= 0)) std::printf("Iterative method costs %.4f seconds.\n", wall_seconds() - t0);
    return iter;
}
// SolBSR.c:145
int fasp_solver_dbsr_krylov(dBSRmat* A, dvector* b, dvector* x, ITS_param* itparam)
{
    if (!itparam) return ERROR_INPUT_PAR;
    const double t0 = wall_seconds();
    const int status = fasp_solver_dbsr_itsolver(A, b, x, nullptr, itparam);
    if (itparam->print_level >= PRINT_MIN) std::printf("Krylov method totally costs %.4f seconds.\n", wall_seconds() - t0);
    return status;
}
// SolBSR.c:186: block-diagonal preconditioner, inverse blocks by fasp_smat_inv (nb <= 3 here)
int fasp_solver_dbsr_krylov_diag(dBSRmat* A, dvector* b, dvector* x, ITS_param* itparam)
{
    if (!A || !itparam || !A->IA || !A->JA || !A->val) return ERROR_INPUT_PAR;
    if (A->nb < 1 || A->nb > 3) {
        std::printf("### ERROR: fasp_hip: block-diagonal preconditioner needs 1 <= nb <= 3, got %d\n", A->nb);
        return ERROR_INPUT_PAR;
    }
    const double t0 = wall_seconds();
    const int nb2 = A->nb * A->nb;
    std::vector<double> dv((size_t)std::max(A->ROW, 1) * nb2, 0.0);
    const int st = bsr_diaginv(A, dv.data());
    if (st < 0) return st;
    precond_diag_bsr diag;
    diag.nb = A->nb; diag.diag.row = A->ROW * nb2; diag.diag.val = dv.data();
    precond pc{&diag, fasp_precond_dbsr_diag};
    const int status = fasp_solver_dbsr_itsolver(A, b, x, &pc, itparam);
    if (itparam->print_level > PRINT_NONE) std::printf("Diag_Krylov method totally costs %.4f seconds.\n", wall_seconds() - t0);
    return status;
}

// ---------------------------------------------------------------------------
// Matrix-free interface of the reference (fasp.h:1109 mxv_matfree, SolMatFree.c): Krylov methods
// that see the operator only as y = A x.  An operator installed by fasp_solver_matfree_init
// (MAT_CSR / MAT_BSR) is recognised by its function pointer: the matrix is uploaded once and the
// whole iteration stays in HBM.  Any other mf->fct is a host callback: its vectors are staged
// through host memory once per application (as for foreign preconditioners).
// ---------------------------------------------------------------------------
void fasp_hip_mxv_csr(const void* A, const double* x, double* y)  // SolMatFree.c: fasp_blas_mxv_csr
{
    fasp_blas_dcsr_mxv(static_cast<const dCSRmat*>(A), x, y);
}
void fasp_hip_mxv_bsr(const void* A, const double* x, double* y)  // SolMatFree.c: fasp_blas_mxv_bsr
{
    fasp_blas_dbsr_mxv(static_cast<const dBSRmat*>(A), x, y);
}

// SolMatFree.c:201
void fasp_solver_matfree_init(int matrix_format, mxv_matfree* mf, void* A)
{
    switch (matrix_format) {
        case MAT_CSR: mf->fct = fasp_hip_mxv_csr; break;
        case MAT_BSR: mf->fct = fasp_hip_mxv_bsr; break;
        default:  // the reference also knows STR / BLC / CSRL: formats this library does not have
            std::printf("### ERROR: Unknown matrix format %d!\n", matrix_format);
            std::exit(ERROR_DATA_STRUCTURE);
    }
    mf->data = A;
}

namespace {
// which: 0 CG, 1 VGMRES, 2 VFGMRES, 3 BiCGstab, 4 GMRES, 6 GCG
int krylov_matfree(const char* fn, int which, mxv_matfree* mf, dvector* b, dvector* u, precond* pc, double tol,
                   double abstol, int MaxIt, short restart, short StopType, short PrtLvl)
{
    if (ctx_init() < 0) die_no_device(fn);
    if (!mf || !mf->fct || !b || !u || b->row != u->row || b->row <= 0) return ERROR_INPUT_PAR;
    if (comm_size() > 1) return ERROR_INPUT_PAR;
    const int n = b->row;
    std::unique_ptr<TmpCSR> csr;
    std::unique_ptr<fasp_bsr::TmpBSR> bsr;
    KOps K;
    K.n = n; K.nvec = (size_t)n; K.fmt = "MatFree"; K.dist = false;
    K.halo = [](double*) { return 0; };
    TmpVec db(b->val, n), du(u->val, n), dz(nullptr, n), dy(nullptr, n);
    if (!db.d || !du.d || !dz.d || !dy.d) return ERROR_ALLOC_MEM;
    std::vector<double> hx, hy, hr, hz;
    if (mf->fct == fasp_hip_mxv_csr) {
        const dCSRmat* A = static_cast<const dCSRmat*>(mf->data);
        if (!A || A->row != n || A->col != n) return ERROR_INPUT_PAR;
        csr.reset(new TmpCSR(A));
        if (!csr->ok) return ERROR_ALLOC_MEM;
        const DevCSR* dA = &csr->D;
        K.mxv = [dA](const double* x, double* y) { d_mxv(*dA, x, y); };
        K.resid = [dA](const double* x, const double* bb, double* r) { d_resid(*dA, x, bb, r); };
    } else if (mf->fct == fasp_hip_mxv_bsr) {
        const dBSRmat* A = static_cast<const dBSRmat*>(mf->data);
        if (!A || A->ROW * A->nb != n || A->ROW != A->COL) return ERROR_INPUT_PAR;
        bsr.reset(new fasp_bsr::TmpBSR(A));
        if (!bsr->ok) return ERROR_ALLOC_MEM;
        const fasp_bsr::TmpBSR* Mp = bsr.get();
        K.mxv = [Mp](const double* x, double* y) { fasp_bsr::bsr_mxv(*Mp, x, y); };
        K.resid = [Mp](const double* x, const double* bb, double* r) { fasp_bsr::bsr_resid(*Mp, x, bb, r); };
    } else {
        hx.resize((size_t)n); hy.resize((size_t)n);
        auto host_mxv = [&, mf](const double* x, double* y) {
            (void)hipMemcpyAsync(hx.data(), x, sizeof(double) * n, hipMemcpyDeviceToHost, g_ctx.stream);
            (void)hipStreamSynchronize(g_ctx.stream);
            mf->fct(mf->data, hx.data(), hy.data());
            (void)hipMemcpyAsync(y, hy.data(), sizeof(double) * n, hipMemcpyHostToDevice, g_ctx.stream);
            (void)hipStreamSynchronize(g_ctx.stream);  // hy is reused by the next application
        };
        K.mxv = host_mxv;
        K.resid = [&, host_mxv](const double* x, const double* bb, double* r) {  // r = 1.0 b + (-1.0) A x
            host_mxv(x, dy.d);
            (void)hipMemcpyAsync(r, dy.d, sizeof(double) * n, hipMemcpyDeviceToDevice, g_ctx.stream);
            d_axpby(n, 1.0, bb, -1.0, r);
        };
    }
    fasp_hip_amg* h = (pc && pc->fct == fasp_hip_precond_fct) ? static_cast<fasp_hip_amg*>(pc->data) : nullptr;
    if (h && (h->L.empty() || h->L[0].A.row != n)) return ERROR_INPUT_PAR;
    if (h) {
        K.pc = [h](double* in, double** out) { return precond_amg(h, in, out); };
    } else if (pc && pc->fct) {
        hr.resize((size_t)n); hz.resize((size_t)n);
        K.pc = [&, pc](double* in, double** out) {
            HIPCK(hipMemcpyAsync(hr.data(), in, sizeof(double) * n, hipMemcpyDeviceToHost, g_ctx.stream));
            HIPCK(hipStreamSynchronize(g_ctx.stream));
            pc->fct(hr.data(), hz.data(), pc->data);
            HIPCK(hipMemcpyAsync(dz.d, hz.data(), sizeof(double) * n, hipMemcpyHostToDevice, g_ctx.stream));
            *out = dz.d;
            return 0;
        };
    }
    std::vector<double*> ws;
    size_t ws_len = 0;
    double* hh = nullptr;
    K.ws = &ws; K.ws_len = &ws_len; K.hh = &hh;
    K.stats = nullptr;
    Hist   H{nullptr, 0, 0};
    PcgOut po{BIGREAL, BIGREAL, BIGREAL};
    int st;
    switch (which) {
        case 0: st = pcg_mf_device(K, db.d, du.d, tol, abstol, MaxIt, StopType, PrtLvl, &po); break;
        case 1: st = gmres_mf_device(K, true, false, db.d, du.d, tol, MaxIt, restart, StopType, PrtLvl, &po); break;
        case 2: st = gmres_mf_device(K, true, true, db.d, du.d, tol, MaxIt, restart, StopType, PrtLvl, &po); break;
        case 3: st = bicgstab_device(K, db.d, du.d, tol, MaxIt, PrtLvl, &H, &po); break;
        case 4: st = gmres_mf_device(K, false, false, db.d, du.d, tol, MaxIt, restart, StopType, PrtLvl, &po); break;
        case 6: st = gcg_device(K, db.d, du.d, tol, abstol, MaxIt, StopType, PrtLvl, &H, &po); break;
        default: st = ERROR_SOLVER_TYPE;
    }
    du.get(u->val);
    for (double* q : ws) if (q) (void)hipFree(q);
    if (hh) (void)hipFree(hh);
    return st;
}
}  // namespace

// KryPcg.c:1260, KryPbcgs.c:1349, KryPgcg.c:213, KryPgmres.c:1309, KryPvgmres.c:1468, KryPvfgmres.c:1026
int fasp_solver_pcg(mxv_matfree* mf, dvector* b, dvector* u, precond* pc, const double tol, const double abstol,
                    const int MaxIt, const short StopType, const short PrtLvl)
{
    return krylov_matfree(__func__, 0, mf, b, u, pc, tol, abstol, MaxIt, 0, StopType, PrtLvl);
}
int fasp_solver_pbcgs(mxv_matfree* mf, dvector* b, dvector* u, precond* pc, const double tol, const double abstol,
                      const int MaxIt, const short StopType, const short PrtLvl)
{
    return krylov_matfree(__func__, 3, mf, b, u, pc, tol, abstol, MaxIt, 0, StopType, PrtLvl);
}
int fasp_solver_pgcg(mxv_matfree* mf, dvector* b, dvector* u, precond* pc, const double tol, const double abstol,
                     const int MaxIt, const short StopType, const short PrtLvl)
{
    return krylov_matfree(__func__, 6, mf, b, u, pc, tol, abstol, MaxIt, 0, StopType, PrtLvl);
}
int fasp_solver_pgmres(mxv_matfree* mf, dvector* b, dvector* x, precond* pc, const double tol, const double abstol,
                       const int MaxIt, const short restart, const short StopType, const short PrtLvl)
{
    return krylov_matfree(__func__, 4, mf, b, x, pc, tol, abstol, MaxIt, restart, StopType, PrtLvl);
}
int fasp_solver_pvgmres(mxv_matfree* mf, dvector* b, dvector* x, precond* pc, const double tol, const double abstol,
                        const int MaxIt, short restart, const short StopType, const short PrtLvl)
{
    return krylov_matfree(__func__, 1, mf, b, x, pc, tol, abstol, MaxIt, restart, StopType, PrtLvl);
}
int fasp_solver_pvfgmres(mxv_matfree* mf, dvector* b, dvector* x, precond* pc, const double tol, const double abstol,
                         const int MaxIt, const short restart, const short StopType, const short PrtLvl)
{
    return krylov_matfree(__func__, 2, mf, b, x, pc, tol, abstol, MaxIt, restart, StopType, PrtLvl);
}
// KryPminres.c:1283.  Refused: the restart branches of the reference's matrix-free MinRes call
// pc->fct exactly when pc == NULL (:1485, :1563) and skip the preconditioner otherwise.
int fasp_solver_pminres(mxv_matfree*, dvector*, dvector*, precond*, const double, const double, const int, const short,
                        const short)
{
    std::printf("### ERROR: fasp_hip: fasp_solver_pminres (matrix-free MinRes) is not provided; "
                "fasp_solver_dcsr_pminres is\n");
    return ERROR_SOLVER_TYPE;
}

// SolMatFree.c:58: dispatch on itsolver_type
int fasp_solver_itsolver(mxv_matfree* mf, dvector* b, dvector* x, precond* pc, ITS_param* itparam)
{
    if (!itparam) return ERROR_INPUT_PAR;
    const short prtlvl = itparam->print_level, stop_type = itparam->stop_type;
    const int restart = itparam->restart, MaxIt = itparam->maxit;
    const double tol = itparam->tol, abstol = itparam->abstol;
    const double t0 = wall_seconds();
    int iter = ERROR_SOLVER_TYPE;
    if (tol < SMALLREAL) std::printf("### WARNING: Convergence tolerance is too small! [%s:%d]\n", "ITS_CHECK", 74);
    if (MaxIt <= 0) std::printf("### WARNING: Max number of iterations must be POSITIVE! [%s:%d]\n", "ITS_CHECK", 78);
    switch (itparam->itsolver_type) {
        case SOLVER_CG: iter = fasp_solver_pcg(mf, b, x, pc, tol, abstol, MaxIt, stop_type, prtlvl); break;
        case SOLVER_BiCGstab: iter = fasp_solver_pbcgs(mf, b, x, pc, tol, abstol, MaxIt, stop_type, prtlvl); break;
        case SOLVER_MinRes: iter = fasp_solver_pminres(mf, b, x, pc, tol, abstol, MaxIt, stop_type, prtlvl); break;
        case SOLVER_GMRES: iter = fasp_solver_pgmres(mf, b, x, pc, tol, abstol, MaxIt, (short)restart, stop_type, prtlvl); break;
        case SOLVER_VGMRES: iter = fasp_solver_pvgmres(mf, b, x, pc, tol, abstol, MaxIt, (short)restart, stop_type, prtlvl); break;
        case SOLVER_VFGMRES: iter = fasp_solver_pvfgmres(mf, b, x, pc, tol, abstol, MaxIt, (short)restart, stop_type, prtlvl); break;
        case SOLVER_GCG: iter = fasp_solver_pgcg(mf, b, x, pc, tol, abstol, MaxIt, stop_type, prtlvl); break;
        default:
            std::printf("### ERROR: Unknown iterative solver type %d! [%s]\n", itparam->itsolver_type, __func__);
            return ERROR_SOLVER_TYPE;
    }
    if ((prtlvl >= PRINT_SOME) && (iter >= 0)) std::printf("Iterative method costs %.4f seconds.\n", wall_seconds() - t0);
    return iter;
}

// SolMatFree.c:157: Krylov method without preconditioner
int fasp_solver_krylov(mxv_matfree* mf, dvector* b, dvector* x, ITS_param* itparam)
{
    if (!itparam) return ERROR_INPUT_PAR;
    const double t0 = wall_seconds();
    const int status = fasp_solver_itsolver(mf, b, x, nullptr, itparam);
    if (itparam->print_level >= PRINT_MIN) std::printf("Krylov method totally costs %.4f seconds.\n", wall_seconds() - t0);
    return status;
}

void fasp_blas_dcsr_mxv(const dCSRmat* A, const double* x, double* y)
{
    TmpCSR M(A);
    if (!M.ok) die_no_device(__func__);
    TmpVec dx(x, A->col), dy(nullptr, A->row);
    d_mxv(M.D, dx.d, dy.d);
    dy.get(y);
}

void fasp_blas_dcsr_aAxpy(const double alpha, const dCSRmat* A, const double* x, double* y)
{
    TmpCSR M(A);
    if (!M.ok) die_no_device(__func__);
    TmpVec dx(x, A->col), dy(y, A->row);
    d_aAxpy(alpha, M.D, dx.d, dy.d);
    dy.get(y);
}

double fasp_blas_darray_dotprod(const int n, const double* x, const double* y)
{
    if (ctx_init() < 0) die_no_device(__func__);
    TmpVec dx(x, n), dy(y, n);
    double out = 0.0;
    (void)d_dot(n, dx.d, dy.d, &out);
    return out;
}

double fasp_blas_darray_norm2(const int n, const double* x)
{
    if (ctx_init() < 0) die_no_device(__func__);
    TmpVec dx(x, n);
    double out[2] = {0, 0};
    (void)d_norms(n, dx.d, out);
    return std::sqrt(out[0]);
}

double fasp_blas_darray_norminf(const int n, const double* x)
{
    if (ctx_init() < 0) die_no_device(__func__);
    TmpVec dx(x, n);
    double out[2] = {0, 0};
    (void)d_norms(n, dx.d, out);
    return out[1];
}

void fasp_blas_darray_axpy(const int n, const double a, const double* x, double* y)
{
    if (ctx_init() < 0) die_no_device(__func__);
    TmpVec dx(x, n), dy(y, n);
    d_axpy(n, a, dx.d, dy.d);
    dy.get(y);
}

void fasp_blas_darray_axpby(const int n, const double a, const double* x, const double b, double* y)
{
    if (ctx_init() < 0) die_no_device(__func__);
    TmpVec dx(x, n), dy(y, n);
    d_axpby(n, a, dx.d, b, dy.d);
    dy.get(y);
}

void fasp_blas_dbsr_mxv(const dBSRmat* A, const double* x, double* y)
{
    TmpBSR M(A);
    if (!M.ok) die_bsr(__func__);
    TmpVec dx(x, (size_t)A->COL * A->nb), dy(nullptr, (size_t)A->ROW * A->nb);
    BsrArgs a{}; a.x = dx.d; a.y = dy.d;
    launch_bsr<0>(M, a);
    dy.get(y);
}

void fasp_blas_dbsr_aAxpy(const double alpha, const dBSRmat* A, const double* x, double* y)
{
    if (alpha == 0.0) return;  // BlaSpmvBSR.c:548
    TmpBSR M(A);
    if (!M.ok) die_bsr(__func__);
    TmpVec dx(x, (size_t)A->COL * A->nb), dy(y, (size_t)A->ROW * A->nb);
    BsrArgs a{}; a.x = dx.d; a.y = dy.d; a.alpha = alpha;
    launch_bsr<1>(M, a);
    dy.get(y);
}

// BlaSparseBSR.c:543 (host): diagonal blocks inverted with the reference's closed forms
// (fasp_smat_inv_nc2 / _nc3, BlaSmallMatInv.c:33 / :67); nb == 1: reciprocals
dvector fasp_dbsr_getdiaginv(const dBSRmat* A)
{
    dvector out{0, nullptr};
    if (!A || A->nb < 1 || A->nb > 3) {
        std::fprintf(stderr, "### ERROR: fasp_dbsr_getdiaginv: block size %d not supported (1..3)\n", A ? A->nb : -1);
        return out;
    }
    out.row = A->ROW * A->nb * A->nb;
    out.val = (double*)std::calloc((size_t)std::max(out.row, 1), sizeof(double));
    (void)bsr_diaginv(A, out.val);
    return out;
}

void fasp_smoother_dbsr_jacobi1(dBSRmat* A, dvector* b, dvector* u, double* diaginv)
{
    TmpBSR M(A);
    if (!M.ok) die_bsr(__func__);
    const size_t n = (size_t)A->ROW * A->nb;
    TmpVec du(u->val, n), dun(nullptr, n), db(b->val, n), dd(diaginv, (size_t)A->ROW * A->nb * A->nb);
    BsrArgs a{}; a.x = du.d; a.y = dun.d; a.b = db.d; a.dinv = dd.d;
    launch_bsr<2>(M, a);
    dun.get(u->val);
}

double fasp_hip_time_bsr_mxv(const dBSRmat* A, int reps)
{
    TmpBSR M(A);
    if (!M.ok || reps <= 0) return -1.0;
    TmpVec dx(nullptr, (size_t)A->COL * A->nb), dy(nullptr, (size_t)A->ROW * A->nb);
    (void)hipMemsetAsync(dx.d, 0, sizeof(double) * dx.n, g_ctx.stream);
    BsrArgs a{}; a.x = dx.d; a.y = dy.d;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch_bsr<0>(M, a); launch_bsr<0>(M, a);
    (void)hipEventRecord(e0, g_ctx.stream);
    for (int i = 0; i < reps; ++i) launch_bsr<0>(M, a);
    (void)hipEventRecord(e1, g_ctx.stream);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return (double)ms / reps;
}

// ItrSmootherCSR.c:98 -- rows i_1..i_n (either direction) of the square system
void fasp_smoother_dcsr_jacobi(dvector* u, const int i_1, const int i_n, const int s, dCSRmat* A, dvector* b,
                               int L, const double w)
{
    (void)s;
    TmpCSR M(A);
    if (!M.ok) die_no_device(__func__);
    const int n = A->row;
    const int lo = std::min(i_1, i_n), hi = std::max(i_1, i_n);
    if (lo != 0 || hi != n - 1) {
        std::fprintf(stderr, "### ERROR: fasp_smoother_dcsr_jacobi (device): only full sweeps 0..n-1 supported\n");
        std::exit(ERROR_INPUT_PAR);
    }
    std::vector<double> d(n, 0.0);
    for (int i = 0; i < n; ++i)
        for (int k = A->IA[i]; k < A->IA[i + 1]; ++k)
            if (A->JA[k] == i) d[i] = A->val[k];
    TmpVec du(u->val, n), du2(nullptr, n), db(b->val, n), dd(d.data(), n);
    double *x = du.d, *xo = du2.d;
    while (L--) {
        CsrArgs a{};
        a.x = x; a.y = xo; a.b = db.d; a.diag = dd.d; a.omega = w;
        launch_csr<OP_JACOBI>(M.D, a);
        std::swap(x, xo);
    }
    (void)hipStreamSynchronize(g_ctx.stream);
    (void)hipMemcpy(u->val, x, sizeof(double) * n, hipMemcpyDeviceToHost);
}

// development knob: override kernel selection / launch geometry at run time
int fasp_hip_tune(const char* key, int value)
{
    if (!key) return ERROR_INPUT_PAR;
    if (!std::strcmp(key, "maxgrid")) g_tune.maxgrid = value;
    else if (!std::strcmp(key, "xcd")) g_tune.xcd = value;
    else if (!std::strcmp(key, "nt")) g_tune.nt = value;
    else if (!std::strcmp(key, "kind")) g_tune.kind = value;
    else if (!std::strcmp(key, "compress")) g_tune.compress = value;
    else if (!std::strcmp(key, "rpl")) g_tune.rpl = value;
    else if (!std::strcmp(key, "lds_tab")) g_tune.lds_tab = value;
    else if (!std::strcmp(key, "xcd_pat")) g_tune.xcd_pat = value;
    else if (!std::strcmp(key, "spcg_batch")) g_tune.spcg_batch = value;
    else if (!std::strcmp(key, "small_lds")) g_tune.small_lds = value;
    else if (!std::strcmp(key, "ja16")) g_tune.ja16 = value;
    else if (!std::strcmp(key, "host_parallel_min")) g_parallel_min_nnz = value;
    else if (!std::strcmp(key, "lanes")) g_tune.lanes = value;
    else if (!std::strcmp(key, "wrows")) g_tune.wrows = value;
    else if (!std::strcmp(key, "wcap")) g_tune.wcap = value;
    else return ERROR_INPUT_PAR;
    return FASP_SUCCESS;
}

// timed micro-benchmark of one kernel class on a resident level
double fasp_hip_time_kernel(fasp_hip_amg* h, int kind, int level, int reps)
{
    if (!h || level < 0 || level >= (int)h->L.size() || reps <= 0) return -1.0;
    DevLevel& D = h->L[level];
    const int n = D.A.row;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.0;
    double* x = D.xa; double* y = D.xb; double* w = D.w;
    auto run = [&]() {
        switch (kind) {
            case 0: d_mxv(D.A, x, y); break;
            case 1: d_aAxpy(-1.0, D.A, x, y); break;
            case 2: { CsrArgs a{}; a.x = x; a.y = y; a.b = w; a.diag = D.diag; a.omega = 0.6667; launch_csr<OP_JACOBI>(D.A, a); } break;
            case 3: hipLaunchKernelGGL(k_dot, dim3(vec_grid(n)), dim3(BLOCK), 0, g_ctx.stream, n, x, y, g_ctx.d_partials); break;
            case 4: d_axpy(n, 0.5, x, y); break;
            case 5: { CsrArgs a{}; a.x = x; a.y = y; a.dotv = x; a.partials = g_ctx.d_partials; launch_csr<OP_MXV_DOT>(D.A, a); } break;
            case 6: if (D.R.ia) d_mxv(D.R, w, h->L[level + 1].xa); break;
            case 7: if (D.P.ia) d_aAxpy(1.0, D.P, h->L[level + 1].xa, y); break;
            default: break;
        }
    };
    (void)hipMemsetAsync(x, 0, sizeof(double) * n, g_ctx.stream);
    (void)hipMemsetAsync(y, 0, sizeof(double) * n, g_ctx.stream);
    (void)hipMemsetAsync(w, 0, sizeof(double) * n, g_ctx.stream);
    run(); run();
    (void)hipEventRecord(e0, g_ctx.stream);
    for (int i = 0; i < reps; ++i) run();
    (void)hipEventRecord(e1, g_ctx.stream);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return (double)ms / reps;
}

}  // extern "C"
