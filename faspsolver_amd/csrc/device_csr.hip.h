// device_csr.hip.h -- device matrices: CSR in HBM, lossless coding at upload, kernel selection and launch, BLAS-1 wrappers.
// Part of the single translation unit solver.hip (included there, in this order; not a stand-alone header).

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
    unsigned short* ja16 = nullptr;  // JA as 16-bit values (long-row operators with <= 65536 columns, or whose rows each span less than 65536 columns: then relative to jbase)
    int*            jbase = nullptr; // ja16 is relative to the row's smallest column (operators with more than 65536 columns); nullptr: absolute
    int     nxrows = -1;             // k_csr_rowpat4 / 5: rows outside their wave's uniform pattern(s); -1: the pair sweep does not apply
    int     plane = 0;               // square row-pattern-coded operators: the largest column offset of a pattern (the rows of a z-plane of a 3-D grid)
    // local operator of a row-partitioned level: the rows [win_lo, win_hi) read no ghost column (multiples of WIN_ALIGN
    // or the row count; win_hi < 0: no such window worth a split launch) -- they run while the halo is in flight
    int     win_lo = 0, win_hi = -1;
    // k_csr_xtile: distinct columns per 64-row wave tile + 16-bit positions per entry (nullptr: not built)
    unsigned short* lja16 = nullptr;
    int*    tptr = nullptr;
    int*    tcols = nullptr;
    long long ntcols = 0;            // entries of tcols
    // numbering bridge (hierarchy.hip.h, brick renumbering behind a CODED level): the operator itself keeps the natural numbering on both
    // sides -- its row patterns stay what they are -- and the side that lives on the renumbered level is permuted on the fly:
    // bridge_dir 1 (restriction): the product goes to bscratch, y[k] = bscratch[bridge[k]]; 2 (prolongation): bscratch[bridge[k]] = x[k], then the
    // product reads bscratch.  bridge = the level's order (new -> natural), owned by the level; bscratch owned here.
    const int* bridge = nullptr;
    double*    bscratch = nullptr;
    int        bridge_dir = 0;
    // k_csr_estream (kernels3.hip.h): the entry-parallel decomposition of a long-row operator with 16-bit columns (build_estream)
    int*       es_tab = nullptr;    // one allocation: wc[W + 1], centry[nc + 1], crow[nc + 1], hw0[W], np[W], cbase[nc] (relative columns only)
    unsigned short* es_ja16 = nullptr;   // 16-bit columns relative to the chunk's smallest column (operators with > 65536 columns)
    double*    es_part = nullptr;   // 2 W doubles + W counters behind them
    int        es_W = 0, es_nc = 0;
    void    release()
    {
        if (es_tab) (void)hipFree(es_tab);
        if (es_part) (void)hipFree(es_part);
        if (es_ja16) (void)hipFree(es_ja16);
        es_tab = nullptr; es_part = nullptr; es_ja16 = nullptr; es_W = es_nc = 0;
        if (bscratch) (void)hipFree(bscratch);
        bscratch = nullptr; bridge = nullptr; bridge_dir = 0;
        if (lja16) (void)hipFree(lja16);
        if (tptr) (void)hipFree(tptr);
        if (tcols) (void)hipFree(tcols);
        lja16 = nullptr; tptr = nullptr; tcols = nullptr;
        nxrows = -1;
        if (ja16) (void)hipFree(ja16);
        ja16 = nullptr;
        if (jbase) (void)hipFree(jbase);
        jbase = nullptr;
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
    // lanes per row of the sub-wavefront kernel, measured on the levels of P7(256) (profiles/r02_lanes_per_row.txt):
    // 16 lanes up to ~300 entries per row, 32 up to ~1000, 64 beyond -- and never so few that rows x lanes leaves
    // the 2048 x 256 threads of the persistent grid unfilled (the last levels: few, very long rows)
    M.lanes = avg < 300.0 ? 16 : avg < 1000.0 ? 32 : 64;
    while (M.lanes < 64 && (double)M.row * M.lanes < 0.75 * MAXGRID * BLOCK) M.lanes *= 2;
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
    // column offsets relative to the row index: square operators, and a rank's rows of a partitioned level (the ghost columns of
    // a boundary plane are numbered like the plane: the same offset in every row of it)
    const bool square = M.row == M.col || M.row_aligned;
    const int n = M.row;
    if (n <= 0 || M.nnz <= 0) return false;
    if ((long long)M.col >= (1ll << 28)) return false;   // the coded kernels form byte offsets of x in 32 bits
    constexpr int MAXPAT = 65535, MAXENT = 1 << 20;   // (id 0xffff is the pad of the pair kernels)
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
    // worth it only when rows really repeat (>= 8 rows per pattern, below): the sets are sized for that bound, and a
    // small operator is hashed by a small team (a 3 MB set per thread of a 32-thread team costs more than the hashing)
    const int maxpat = std::min<long long>(MAXPAT, n / 8);
    if (maxpat < 1) return false;
    const int nt = std::max(1, std::min(omp_get_max_threads(), n / 32768));
    int slots_pow = 1024;
    while (slots_pow < 4 * (maxpat + 1)) slots_pow *= 2;   // open addressing, <= 25 % load
    const int SLOTS = slots_pow;
    struct Set {
        std::vector<unsigned long long> key;
        std::vector<int> first;
        int cnt = 0, SLOTS;
        explicit Set(int slots) : key((size_t)slots, 0ull), first((size_t)slots, -1), SLOTS(slots) {}
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
        Set* S = new Set(SLOTS);
        local[(size_t)omp_get_thread_num()] = S;
#pragma omp for schedule(static)
        for (int r = 0; r < n; ++r) {
            const unsigned long long h = row_hash(r);
            rh[r] = h;
            if (fail) continue;
            S->add(h, r);
            if (S->cnt > maxpat) fail = true;
        }
    }
    Set* Gs = nullptr;
    std::vector<std::pair<int, unsigned long long>> reps;  // (first row, hash)
    if (!fail) {
        Gs = new Set(SLOTS);
        for (Set* S : local)
            if (S)
                for (int s = 0; s < SLOTS && !fail; ++s)
                    if (S->key[s]) { Gs->add(S->key[s], S->first[s]); if (Gs->cnt > maxpat) fail = true; }
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
#pragma omp parallel for schedule(static) num_threads(nt)
    for (int r = 0; r < n; ++r) {
        const int id = Gs->find(rh[r]);
        if (id < 0 || !same_row(r, reps[(size_t)id].first)) { collision = true; continue; }
        pat[r] = (unsigned short)id;
        if (!square) rowbase[r] = base_of(r);
    }
    delete Gs;
    return !collision;
}

// The "plane" of a square pattern-coded operator: the row distance P (a multiple of 8 tiles of the pair sweep) for which the
// column offsets of the patterns are closest to multiples of P -- the rows of a z-plane of a 3-D grid numbered x-fastest, or of
// whatever the coarse numbering has made of it.  With it an XCD can sweep ONE STRIP (P / 8 rows) of every plane: a row's z-neighbours
// are in the same strip of the neighbouring planes, its x / y neighbours in the same strip except at the strip's edges
// (kernels.hip.h, tile_of, xcd_map == -2).  Cost of a candidate = mean over the pattern entries of min(|o mod+- P|, P / 8) / (P / 8):
// the share of rows whose neighbour lies in another XCD's strip.  0: no candidate below 0.2.
static int grid_plane_of(const std::vector<int>& poff, int nrow)
{
    constexpr int UNIT = 8 * 2 * BLOCK;   // eight tiles of 2 * BLOCK rows
    std::vector<int> cand;
    for (int o : poff) {
        const long long m = std::llabs((long long)o);
        const long long c = (m + UNIT / 2) / UNIT * UNIT;
        if (c >= UNIT && c <= nrow / 4 && std::find(cand.begin(), cand.end(), (int)c) == cand.end() && cand.size() < 64) cand.push_back((int)c);
    }
    int best = 0;
    double best_cost = 0.2;
    for (int P : cand) {
        double cost = 0.0;
        for (int o : poff) {
            long long r = (long long)o % P;
            if (r > P / 2) r -= P;
            if (r < -P / 2) r += P;
            cost += (double)std::min<long long>(std::llabs(r), P / 8) / (P / 8);
        }
        cost /= std::max<size_t>(poff.size(), 1);
        if (cost < best_cost || (cost == best_cost && P > best)) { best_cost = cost; best = P; }
    }
    return best;
}

// 16-bit copy of the column indices (in the order of the device copy) for the operators the
// sub-wavefront kernel serves: their time is the (JA, val) stream, 12 -> 10 bytes per entry.
// (Round 5: operators with MORE than 65536 columns whose every row spans less than 65536 of them -- levels 3 and 4 of P7(256): 257 139 and
// 80 619 rows, spans of a few grid planes -- store their indices relative to the row's smallest column, 4 bytes per row more.)
static bool rows_span_16bit(const int* ia, const int* ja, int row)
{
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad)
    for (int i = 0; i < row; ++i) {
        if (ia[i + 1] <= ia[i]) continue;
        int lo = ja[ia[i]], hi = lo;
        for (int k = ia[i] + 1; k < ia[i + 1]; ++k) { lo = std::min(lo, ja[k]); hi = std::max(hi, ja[k]); }
        if (hi - lo >= 65536) ++bad;
    }
    return bad == 0;
}
static int upload_ja16(DevCSR& D, const int* ia_host, const int* ja_dev_order)
{
    static const bool on = !(std::getenv("FASP_HIP_JA16") && std::atoi(std::getenv("FASP_HIP_JA16")) == 0);
    if (!on || D.kind != 0 || D.nnz < 4096) return FASP_SUCCESS;
    const bool relative = D.col > 65536;
    if (relative && !rows_span_16bit(ia_host, ja_dev_order, D.row)) return FASP_SUCCESS;
    Buf<unsigned short> j16((size_t)D.nnz);
    Buf<int> base(relative ? (size_t)D.row : 0);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < D.row; ++i) {
        int lo = 0;
        if (relative) {
            lo = ia_host[i + 1] > ia_host[i] ? ja_dev_order[ia_host[i]] : 0;
            for (int k = ia_host[i]; k < ia_host[i + 1]; ++k) lo = std::min(lo, ja_dev_order[k]);
            base[(size_t)i] = lo;
        }
        for (int k = ia_host[i]; k < ia_host[i + 1]; ++k) j16[k] = (unsigned short)(ja_dev_order[k] - lo);
    }
    HIPCK(hipMalloc(&D.ja16, sizeof(unsigned short) * ((size_t)D.nnz + 8)));
    HIPCK(hipMemset(D.ja16 + D.nnz, 0, sizeof(unsigned short) * 8));   // (k_csr_estream reads whole 16-byte pieces: the slack must hold valid columns)
    HIPCK(hipMemcpy(D.ja16, j16.data(), sizeof(unsigned short) * (size_t)D.nnz, hipMemcpyHostToDevice));
    if (relative) {
        HIPCK(hipMalloc(&D.jbase, sizeof(int) * (size_t)D.row));
        HIPCK(hipMemcpy(D.jbase, base.data(), sizeof(int) * (size_t)D.row, hipMemcpyHostToDevice));
    }
    return FASP_SUCCESS;
}

static int g_device_sort = 1;   // fasp_hip_tune("device_sort", 0): per-row sort of the device copies on the host (A/B tests)
// Stable sort of every row's entries by column ON THE DEVICE: one workgroup per row, bitonic network in LDS over the
// keys (column << 32 | position in the row) -- keys are distinct, so the result is THE stable order, the same the host
// std::stable_sort of the fallback below produces.  P7(256): levels 2-9 carry 15-19 M entries each in rows of 64-3100;
// sorting them on the host cost 0.2-0.5 s per level (2.9 s of the 4.4 s upload), here < 5 ms.
constexpr int SORT_BLK = 256, SORT_MAXLEN = 16384;   // 16384 keys = 128 KB of LDS
__global__ __launch_bounds__(SORT_BLK) void k_sort_rows(int nrow, const int* __restrict__ ia, const int* __restrict__ ja_in,
                                                        const double* __restrict__ val_in, int* __restrict__ ja_out,
                                                        double* __restrict__ val_out, unsigned short* __restrict__ ja16, int* __restrict__ jbase)
{
    extern __shared__ unsigned long long sort_keys[];
    const int tid = threadIdx.x;
    for (int r = blockIdx.x; r < nrow; r += gridDim.x) {
        const int kb = ia[r], len = ia[r + 1] - kb;
        int P = 1;
        while (P < len) P <<= 1;
        for (int i = tid; i < P; i += SORT_BLK)
            sort_keys[i] = i < len ? ((unsigned long long)(unsigned)ja_in[kb + i] << 32) | (unsigned)i : ~0ull;
        __syncthreads();
        for (int k = 2; k <= P; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < P / 2; t += SORT_BLK) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                    const unsigned long long a = sort_keys[i], b = sort_keys[l];
                    if ((a > b) == ((i & k) == 0)) { sort_keys[i] = b; sort_keys[l] = a; }
                }
                __syncthreads();
            }
        const int base = (jbase && len > 0) ? (int)(sort_keys[0] >> 32) : 0;   // (sorted: the row's smallest column)
        if (jbase && tid == 0) jbase[r] = base;
        for (int i = tid; i < len; i += SORT_BLK) {
            const unsigned long long key = sort_keys[i];
            const int col = (int)(key >> 32);
            ja_out[kb + i] = col;
            val_out[kb + i] = val_in[kb + (int)(unsigned)key];
            if (ja16) ja16[kb + i] = (unsigned short)(col - base);
        }
        __syncthreads();
    }
}

static int upload_sorted_on_device(const HostCSR& H, DevCSR& D, int maxlen)
{
    static const bool ja16_on = !(std::getenv("FASP_HIP_JA16") && std::atoi(std::getenv("FASP_HIP_JA16")) == 0);
    int* tj = nullptr; double* tv = nullptr;
    HIPCK(hipMalloc(&tj, sizeof(int) * (size_t)H.nnz));
    if (hipMalloc(&tv, sizeof(double) * (size_t)H.nnz) != hipSuccess) { (void)hipFree(tj); return ERROR_ALLOC_MEM; }
    int st = FASP_SUCCESS;
    if (hipMemcpyAsync(tj, H.ja.data(), sizeof(int) * (size_t)H.nnz, hipMemcpyHostToDevice, g_ctx.stream) != hipSuccess ||
        hipMemcpyAsync(tv, H.val.data(), sizeof(double) * (size_t)H.nnz, hipMemcpyHostToDevice, g_ctx.stream) != hipSuccess) st = ERROR_ALLOC_MEM;
    const bool relative = D.col > 65536;
    const bool want16 = ja16_on && D.kind == 0 && D.nnz >= 4096 && (!relative || rows_span_16bit(H.ia.data(), H.ja.data(), H.row));   // as upload_ja16
    if (st >= 0 && want16 && hipMalloc(&D.ja16, sizeof(unsigned short) * ((size_t)D.nnz + 8)) != hipSuccess) st = ERROR_ALLOC_MEM;
    if (st >= 0 && want16 && hipMemsetAsync(D.ja16 + D.nnz, 0, sizeof(unsigned short) * 8, g_ctx.stream) != hipSuccess) st = ERROR_MISC;   // (the slack k_csr_estream's last 16-byte piece reads)
    if (st >= 0 && want16 && relative && hipMalloc(&D.jbase, sizeof(int) * (size_t)std::max(D.row, 1)) != hipSuccess) st = ERROR_ALLOC_MEM;
    if (st >= 0) {
        int P = 64;
        while (P < maxlen) P <<= 1;
        const size_t lds = sizeof(unsigned long long) * (size_t)P;
        if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(k_sort_rows), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) st = ERROR_MISC;
        if (st >= 0) {
            const int grid = std::min(H.row, 256 * 64);
            hipLaunchKernelGGL(k_sort_rows, dim3(grid), dim3(SORT_BLK), lds, g_ctx.stream, H.row, D.ia, tj, tv, D.ja, D.val, D.ja16, D.jbase);
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(g_ctx.stream) != hipSuccess) st = ERROR_MISC;
        }
    }
    (void)hipFree(tj); (void)hipFree(tv);
    if (st >= 0) D.sorted = true;
    return st;
}

// k_csr_xtile's structures (kernels2.hip.h): per wave tile of 64 rows the sorted list of the distinct columns of its
// entries, per entry the position of its column in that list.  Refused (nothing built) when a tile has more than
// XT_XCAP distinct columns or the lists would not pay for themselves.
static int build_xtile(const HostCSR& H, DevCSR& D)
{
    static const bool on = !(std::getenv("FASP_HIP_XTILE") && std::atoi(std::getenv("FASP_HIP_XTILE")) == 0);
    if (!on || H.row < 4096 || H.nnz < 16 * H.row) return FASP_SUCCESS;   // short rows: k_csr_lstream's territory
    const int ntile = (H.row + 63) / 64;
    double min_share = 2.5;
    if (const char* e = std::getenv("FASP_HIP_XTILE_MIN_SHARE")) min_share = std::atof(e);
    {   // a sample of the tiles first (every 61st): most operators are decided here, for a percent of the work
        long long ent = 0, dist = 0, nfat = 0, nsamp = 0;
        std::vector<int> cols;
        for (int t = 0; t < ntile; t += 61) {
            const int k0 = H.ia[t * 64], k1 = H.ia[std::min(H.row, t * 64 + 64)];
            cols.assign(H.ja.data() + k0, H.ja.data() + k1);
            std::sort(cols.begin(), cols.end());
            cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
            ent += k1 - k0; dist += (long long)cols.size();
            ++nsamp;
            if ((int)cols.size() > XT_XCAP) ++nfat;
        }
        const bool fat = nfat * 20 > nsamp;   // more than 5 % of the tiles beyond the LDS list
        if (fat || (double)dist * min_share > 1.1 * (double)ent) {
            if (std::getenv("FASP_HIP_SETUP_TIMING"))
                std::printf("        [xtile %d x %d, %d nnz] sample: %.1f entries per distinct column%s: not built\n", H.row, H.col, H.nnz,
                            (double)ent / std::max<long long>(dist, 1), fat ? ", a tile beyond the LDS list" : "");
            return FASP_SUCCESS;
        }
    }
    std::vector<int> cnt((size_t)ntile + 1, 0);
    Buf<unsigned short> l16((size_t)H.nnz);
    std::vector<std::vector<int>> lists((size_t)ntile);
    long long nfat_all = 0;
#pragma omp parallel
    {
        std::vector<int> cols;
#pragma omp for schedule(dynamic, 64) reduction(+ : nfat_all)
        for (int t = 0; t < ntile; ++t) {
            const int k0 = H.ia[t * 64], k1 = H.ia[std::min(H.row, t * 64 + 64)];
            cols.assign(H.ja.data() + k0, H.ja.data() + k1);
            std::sort(cols.begin(), cols.end());
            cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
            if ((int)cols.size() > XT_XCAP) {   // no list: the kernel gathers this tile's operands from global memory
                for (int k = k0; k < k1; ++k) l16[k] = 0;
                ++nfat_all;
                continue;
            }
            for (int k = k0; k < k1; ++k)
                l16[k] = (unsigned short)(std::lower_bound(cols.begin(), cols.end(), H.ja[k]) - cols.begin());
            cnt[(size_t)t + 1] = (int)cols.size();
            lists[(size_t)t] = cols;
        }
    }
    static const bool timing = std::getenv("FASP_HIP_SETUP_TIMING") != nullptr;
    if (nfat_all * 20 > ntile) {
        if (timing) std::printf("        [xtile %d x %d, %d nnz] %lld of %d tiles have more than %d distinct columns: not built\n", H.row, H.col, H.nnz, nfat_all, ntile, XT_XCAP);
        return FASP_SUCCESS;
    }
    int fattest = 0;
    for (int t = 0; t < ntile; ++t) { fattest = std::max(fattest, cnt[(size_t)t + 1]); cnt[(size_t)t + 1] += cnt[(size_t)t]; }
    const long long total = cnt[(size_t)ntile];
    if (timing) std::printf("        [xtile %d x %d, %d nnz] %lld distinct columns in %d tiles (%.1f entries per distinct column, fattest listed tile %d, %lld tiles without a list)\n",
                            H.row, H.col, H.nnz, total, ntile, (double)H.nnz / std::max<long long>(total, 1), fattest, nfat_all);
    // worth it when the rows of a tile share their columns: >= 2.5 entries per distinct column (then 10 + 4 / 2.5 = 11.6
    // bytes per entry against 12, and 2.5 x fewer gathers).  AMG coarse levels in C-point order share little (P7(256):
    // 1.5 on level 1, > 1024 distinct columns per tile on level 2): they keep k_csr_wstream2 until their rows are
    // re-ordered in bricks.  FASP_HIP_XTILE_MIN_SHARE overrides (tests).
    if ((double)total * min_share > (double)H.nnz) return FASP_SUCCESS;
    std::vector<int> flat((size_t)std::max<long long>(total, 1));
#pragma omp parallel for schedule(static)
    for (int t = 0; t < ntile; ++t) std::copy(lists[(size_t)t].begin(), lists[(size_t)t].end(), flat.begin() + cnt[(size_t)t]);
    HIPCK(hipMalloc(&D.lja16, sizeof(unsigned short) * ((size_t)H.nnz + 8)));
    HIPCK(hipMalloc(&D.tptr, sizeof(int) * ((size_t)ntile + 1)));
    HIPCK(hipMalloc(&D.tcols, sizeof(int) * flat.size()));
    HIPCK(hipMemcpy(D.lja16, l16.data(), sizeof(unsigned short) * (size_t)H.nnz, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(D.tptr, cnt.data(), sizeof(int) * ((size_t)ntile + 1), hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(D.tcols, flat.data(), sizeof(int) * flat.size(), hipMemcpyHostToDevice));
    D.ntcols = total;
    return FASP_SUCCESS;
}

// k_csr_estream's tables (kernels3.hip.h): the entries of a long-row operator cut into W equal wave ranges and those into chunks of at
// most ES_CAP entries, with the row every chunk starts in, and for the rows that a wave boundary cuts who counts their parts.  Built
// from the row pointers alone (the device copy's rows are sorted, its pointers are the host's).  W depends on the operator only --
// not on the device -- so the association of every row sum is a property of the matrix.
struct EsTables { int W = 0, nc = 0; std::vector<int> wc, centry, crow, hw0, np; };
static void build_estream_host(const int* ia, int nrow, int nnz, int per_wave, int wmax, EsTables& T)
{
    int W = (int)std::min<long long>(wmax, ((long long)nnz + per_wave - 1) / per_wave);
    W = std::max(32, (W + 31) / 32 * 32);   // 8 XCDs x 4 waves per workgroup
    std::vector<int> ebeg((size_t)W + 1);
    for (int w = 0; w <= W; ++w) ebeg[(size_t)w] = (int)(((long long)nnz * w / W) & ~7ll);
    ebeg[(size_t)W] = nnz;
    T.W = W;
    T.wc.assign((size_t)W + 1, 0); T.hw0.assign((size_t)W, 0); T.np.assign((size_t)W, 0);
    T.centry.clear();
    T.centry.reserve((size_t)nnz / ES_CAP + (size_t)W + 2);
    for (int w = 0; w < W; ++w) {
        T.wc[(size_t)w] = (int)T.centry.size();
        const int len = ebeg[(size_t)w + 1] - ebeg[(size_t)w];
        if (len <= 0) continue;
        const int nch = (len + ES_CAP - 1) / ES_CAP;
        const int size = ((len + nch - 1) / nch + 7) & ~7;
        for (int j = 0; j < nch && ebeg[(size_t)w] + j * size < ebeg[(size_t)w + 1]; ++j) T.centry.push_back(ebeg[(size_t)w] + j * size);
    }
    T.wc[(size_t)W] = (int)T.centry.size();
    T.centry.push_back(nnz);
    const int nc = T.nc = (int)T.centry.size() - 1;
    T.crow.resize((size_t)nc + 1);
    auto row_of = [&](int e) {   // the row entry e lies in (e < nnz): the last r with ia[r] <= e
        return (int)(std::upper_bound(ia, ia + nrow + 1, e) - ia) - 1;
    };
#pragma omp parallel for schedule(static)
    for (int c = 0; c <= nc; ++c) T.crow[(size_t)c] = c == 0 ? 0 : c == nc ? nrow - 1 : std::min(nrow - 1, row_of(T.centry[(size_t)c]));
    auto wave_of = [&](int e) { return (int)(std::upper_bound(ebeg.begin(), ebeg.end(), e) - ebeg.begin()) - 1; };
    for (int w = 0; w < W; ++w) {
        const int e0 = ebeg[(size_t)w], e1 = ebeg[(size_t)w + 1];
        if (e1 <= e0) continue;
        const int rh = e0 == 0 ? 0 : row_of(e0);
        if (ia[rh] < e0) T.hw0[(size_t)w] = wave_of(ia[rh]);
        if (e1 < nnz) {
            const int rt = row_of(e1);
            if (ia[rt] < e1 && ia[rt] >= e0) T.np[(size_t)w] = wave_of(ia[rt + 1] - 1) - w + 1;
        }
    }
}

static int build_estream(const int* ia, int nrow, int nnz, DevCSR& D)
{
    static const bool on = !(std::getenv("FASP_HIP_ESTREAM") && std::atoi(std::getenv("FASP_HIP_ESTREAM")) == 0);
    if (!on || !D.ja16 || D.kind != 0 || nnz < 65536 || nrow < 1 || (double)nnz < 48.0 * nrow) return FASP_SUCCESS;   // (long rows: the small transfer operators that run on the row kernel have rows of 3-10 entries)
    static const int per_wave = std::getenv("FASP_HIP_ESTREAM_PER_WAVE") ? std::atoi(std::getenv("FASP_HIP_ESTREAM_PER_WAVE")) : 3072;
    static const int wmax = std::getenv("FASP_HIP_ESTREAM_WMAX") ? std::atoi(std::getenv("FASP_HIP_ESTREAM_WMAX")) : 5120;   // (five workgroups of four waves per CU: what every instantiation keeps resident at once)
    EsTables T;
    build_estream_host(ia, nrow, nnz, per_wave, wmax, T);
    const int W = T.W, nc = T.nc;
    const bool relative = D.jbase != nullptr;
    const size_t nt = (size_t)W + 1 + 2 * ((size_t)nc + 1) + 2 * (size_t)W + (relative ? (size_t)nc + 1 : 0);
    std::vector<int> tab(nt);
    size_t o = 0;
    std::copy(T.wc.begin(), T.wc.end(), tab.begin() + o); o += T.wc.size();
    std::copy(T.centry.begin(), T.centry.end(), tab.begin() + o); o += T.centry.size();
    std::copy(T.crow.begin(), T.crow.end(), tab.begin() + o); o += T.crow.size();
    std::copy(T.hw0.begin(), T.hw0.end(), tab.begin() + o); o += T.hw0.size();
    std::copy(T.np.begin(), T.np.end(), tab.begin() + o);
    HIPCK(hipMalloc(&D.es_tab, sizeof(int) * nt));
    HIPCK(hipMemcpy(D.es_tab, tab.data(), sizeof(int) * nt, hipMemcpyHostToDevice));
    HIPCK(hipMalloc(&D.es_part, sizeof(double) * 2 * (size_t)W + sizeof(unsigned) * (size_t)W));
    HIPCK(hipMemset(D.es_part, 0, sizeof(double) * 2 * (size_t)W + sizeof(unsigned) * (size_t)W));
    D.es_W = W; D.es_nc = nc;
    if (relative) {
        // more than 65536 columns: the row-relative 16-bit columns of the row kernel need every entry's ROW; the stream works on chunks,
        // so it gets columns relative to the smallest column of the CHUNK (built on the device from the sorted 32-bit copy) -- when
        // every chunk spans less than 65536 columns; otherwise the operator keeps the row kernel
        int* cbase = D.es_tab + (size_t)W + 1 + 2 * ((size_t)nc + 1) + 2 * (size_t)W;
        int* flag = cbase + nc;
        const int* centry = D.es_tab + (size_t)W + 1;
        HIPCK(hipMemsetAsync(flag, 0, sizeof(int), g_ctx.stream));
        const int grid = std::max(1, std::min(MAXGRID, (nc + 3) / 4));
        hipLaunchKernelGGL(k_es_chunk_base, dim3(grid), dim3(BLOCK), 0, g_ctx.stream, nc, centry, (const int*)D.ja, cbase, flag);
        int hflag = 0;
        HIPCK(hipMemcpyAsync(&hflag, flag, sizeof(int), hipMemcpyDeviceToHost, g_ctx.stream));
        HIPCK(hipStreamSynchronize(g_ctx.stream));
        if (hflag) {   // a chunk too wide for 16 bits
            (void)hipFree(D.es_tab); (void)hipFree(D.es_part);
            D.es_tab = nullptr; D.es_part = nullptr; D.es_W = D.es_nc = 0;
            return FASP_SUCCESS;
        }
        HIPCK(hipMalloc(&D.es_ja16, sizeof(unsigned short) * ((size_t)nnz + 8)));
        HIPCK(hipMemsetAsync(D.es_ja16 + nnz, 0, sizeof(unsigned short) * 8, g_ctx.stream));
        hipLaunchKernelGGL(k_es_chunk_cols, dim3(grid), dim3(BLOCK), 0, g_ctx.stream, nc, centry, (const int*)D.ja, (const int*)cbase, D.es_ja16);
        HIPCK(hipStreamSynchronize(g_ctx.stream));
    }
    return FASP_SUCCESS;
}

// Host walk of k_csr_estream's control flow over the tables (CPU tests; no device): every entry is covered by exactly one chunk, every
// row is finished exactly once -- inside one wave, or by parts whose count is what its first wave's table says -- and the sum of the
// parts in wave order is the row's sum.  Returns 0, or the negative number of the first check that failed.
static int estream_selftest_host(const int* ia, int nrow, int nnz, int per_wave, int wmax, int* info)
{
    EsTables T;
    build_estream_host(ia, nrow, nnz, per_wave, wmax, T);
    if (info) { info[0] = T.W; info[1] = T.nc; }
    std::vector<int> done((size_t)nrow, 0), parts((size_t)T.W, 0);
    std::vector<long long> covered((size_t)nrow, 0);   // entries of the row seen by chunks
    long long cut_rows = 0;
    for (int c = 0; c < T.nc; ++c) {
        if (T.centry[(size_t)c + 1] <= T.centry[(size_t)c] || T.centry[(size_t)c + 1] - T.centry[(size_t)c] > ES_CAP) return -1;
        if ((T.centry[(size_t)c] & 7) != 0) return -2;
        if (T.crow[(size_t)c] > T.crow[(size_t)c + 1]) return -3;
    }
    if (T.centry[0] != 0 || T.centry[(size_t)T.nc] != nnz) return -4;
    for (int w = 0; w < T.W; ++w) {
        const int c0 = T.wc[(size_t)w], cend = T.wc[(size_t)w + 1];
        if (c0 >= cend) continue;
        const int e0w = T.centry[(size_t)c0], e1w = T.centry[(size_t)cend];
        int pend_r = -1, pend_kb = 0;
        for (int c = c0; c < cend; ++c) {
            const int lo = T.centry[(size_t)c], hi = T.centry[(size_t)c + 1], rf = T.crow[(size_t)c], rl = T.crow[(size_t)c + 1];
            for (int r = rf; r <= rl; ++r) {
                const int kb = ia[r], ke = ia[r + 1];
                covered[(size_t)r] += std::max(0, std::min(ke, hi) - std::max(kb, lo));
                if (ke <= hi) {
                    if (kb >= e0w) { if (done[(size_t)r]++) return -5; }
                    else {
                        const int w0 = T.hw0[(size_t)w];
                        if (w0 < 0 || w0 >= w || T.np[(size_t)w0] < 2 || w0 + T.np[(size_t)w0] - 1 != w) return -6;   // this is the row's LAST part
                        if (++parts[(size_t)w0] == T.np[(size_t)w0]) { if (done[(size_t)r]++) return -7; ++cut_rows; }
                    }
                } else if (c + 1 == cend) { pend_r = r; pend_kb = kb; }
            }
        }
        if (pend_r >= 0 && pend_kb < e1w) {
            const int w0 = pend_kb >= e0w ? w : T.hw0[(size_t)w];
            if (w0 < 0 || w0 > w || T.np[(size_t)w0] < 2 || w0 + T.np[(size_t)w0] - 1 <= w) return -8;   // a first or a middle part: the last one comes later
            if (++parts[(size_t)w0] == T.np[(size_t)w0]) return -9;   // (walking the waves in order, the last part is a finished row above)
        }
    }
    for (int r = 0; r < nrow; ++r) {
        if (done[(size_t)r] != 1) return -10;
        if (covered[(size_t)r] != (long long)ia[r + 1] - ia[r]) return -11;
    }
    for (int w = 0; w < T.W; ++w)
        if (parts[(size_t)w] != T.np[(size_t)w]) return -12;
    if (info) info[2] = (int)cut_rows;
    return 0;
}

static int upload_csr(const HostCSR& H, DevCSR& D)
{
    D.row = H.row; D.col = H.col; D.nnz = H.nnz;
    HIPCK(hipMalloc(&D.ia, sizeof(int) * ((size_t)H.row + 1)));
    HIPCK(hipMalloc(&D.ja, sizeof(int) * std::max<size_t>(H.nnz, 1)));
    HIPCK(hipMalloc(&D.val, sizeof(double) * (std::max<size_t>(H.nnz, 1) + 2)));   // (+ 16 bytes: k_csr_estream's last 16-byte piece)
    HIPCK(hipMemsetAsync(D.val + std::max<size_t>(H.nnz, 1), 0, sizeof(double) * 2, g_ctx.stream));
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
            HIPCK(hipMalloc(&D.pat, sizeof(unsigned short) * ((size_t)H.row + 2)));  // + pad: the pair kernels load two ids at once
            HIPCK(hipMalloc(&D.pstart, sizeof(int) * pstart.size()));
            HIPCK(hipMalloc(&D.plen, sizeof(int) * plen.size()));
            HIPCK(hipMemcpy(D.plen, plen.data(), sizeof(int) * plen.size(), hipMemcpyHostToDevice));
            HIPCK(hipMalloc(&D.poff, sizeof(int) * std::max<size_t>(poff.size(), 1)));
            HIPCK(hipMalloc(&D.pval, sizeof(double) * std::max<size_t>(pval.size(), 1)));
            HIPCK(hipMemset(D.pat + H.row, 0xff, sizeof(unsigned short) * 2));   // pad = 0xffff: never a pattern id (MAXPAT 65535)
            HIPCK(hipMemcpy(D.pat, pat.data(), sizeof(unsigned short) * (size_t)H.row, hipMemcpyHostToDevice));
            HIPCK(hipMemcpy(D.pstart, pstart.data(), sizeof(int) * pstart.size(), hipMemcpyHostToDevice));
            HIPCK(hipMemcpy(D.poff, poff.data(), sizeof(int) * poff.size(), hipMemcpyHostToDevice));
            HIPCK(hipMemcpy(D.pval, pval.data(), sizeof(double) * pval.size(), hipMemcpyHostToDevice));
            if (prb.n) {
                HIPCK(hipMalloc(&D.rowbase, sizeof(int) * ((size_t)H.row + 2)));   // + pad: the pair kernel loads two bases at once
                HIPCK(hipMemcpy(D.rowbase, prb.data(), sizeof(int) * (size_t)H.row, hipMemcpyHostToDevice));
                // k_csr_rowpat5 (kernels2.hip.h): the sweep computes the row pairs whose two patterns are those of the pair
                // (128 w + 64, 128 w + 65) of their wave tile w; every other row goes on this list.
                const int nr = H.row, npair = (nr + 1) / 2;
                long long nx = 0;
#pragma omp parallel for schedule(static) reduction(+ : nx)
                for (int w0 = 0; w0 < nr; w0 += 128) {
                    const int    pm = std::min((w0 + 64) / 2, npair - 1);
                    const unsigned domA = pat[(size_t)2 * pm], domB = (2 * pm + 1 < nr) ? pat[(size_t)2 * pm + 1] : 0xffffu;   // (the device copy is padded with 0xffff)
                    for (int r = w0; r < std::min(w0 + 128, nr); r += 2) {
                        const bool vb = r + 1 < nr;
                        if (!(vb && pat[r] == domA && pat[r + 1] == domB)) nx += vb ? 2 : 1;
                    }
                }
                if (nx * 4 <= nr) D.nxrows = (int)nx;   // worth it when most rows are swept (the others go through the wave's queue)
            } else {
                D.plane = grid_plane_of(poff, H.row);
                if (std::getenv("FASP_HIP_SETUP_TIMING")) std::printf("        [upload_csr %d x %d, %d nnz] %d row patterns; grid plane for the XCD strips: %d rows\n", H.row, H.col, H.nnz, (int)pstart.size(), D.plane);
                // k_csr_rowpat4 (kernels2.hip.h): the sweep computes the row pairs (2i, 2i+1) whose two rows have the
                // pattern of row 128 w + 64 of their wave tile w; every other row goes on this list.
                const int nr = H.row, npair = (nr + 1) / 2;
                long long nx = 0;
#pragma omp parallel for schedule(static) reduction(+ : nx)
                for (int w0 = 0; w0 < nr; w0 += 128) {
                    const unsigned dom = pat[(size_t)2 * std::min((w0 + 64) / 2, npair - 1)];
                    for (int r = w0; r < std::min(w0 + 128, nr); r += 2) {
                        const bool vb = r + 1 < nr;
                        if (!(vb && pat[r] == dom && pat[r + 1] == dom)) nx += vb ? 2 : 1;
                    }
                }
                if (nx * 4 <= nr) D.nxrows = (int)nx;  // worth it when most rows are swept
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
    static const bool timing = std::getenv("FASP_HIP_SETUP_TIMING") != nullptr;
    double tp = wall_seconds();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const double now = wall_seconds();
        std::printf("        [upload_csr %d x %d, %d nnz] %-10s %8.3f s\n", H.row, H.col, H.nnz, what, now - tp);
        tp = now;
    };
    if (do_sort && D.kind == 0) {   // (kind 2 wants the diagonal positions of the sorted copy: host path below)
        int maxlen = 0;
#pragma omp parallel for schedule(static) reduction(max : maxlen)
        for (int i = 0; i < H.row; ++i) maxlen = std::max(maxlen, H.ia[i + 1] - H.ia[i]);
        if (g_device_sort && maxlen <= SORT_MAXLEN) {
            const int st = upload_sorted_on_device(H, D, maxlen);
            lap("device sort");
            return st < 0 ? st : build_estream(H.ia.data(), H.row, H.nnz, D);
        }
    }
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
        lap("row sort");
        HIPCK(hipMemcpy(D.ja, sj.data(), sizeof(int) * (size_t)H.nnz, hipMemcpyHostToDevice));
        HIPCK(hipMemcpy(D.val, sv.data(), sizeof(double) * (size_t)H.nnz, hipMemcpyHostToDevice));
        lap("memcpy");
        if (square && D.kind == 2) {
            HIPCK(hipMalloc(&D.dpos, sizeof(int) * (size_t)std::max(H.row, 1)));
            HIPCK(hipMemcpy(D.dpos, dp.data(), sizeof(int) * (size_t)H.row, hipMemcpyHostToDevice));
        }
        D.sorted = true;
        if (upload_ja16(D, H.ia.data(), sj.data()) < 0) return ERROR_ALLOC_MEM;
        return build_estream(H.ia.data(), H.row, H.nnz, D);
    }
    if (upload_plain() < 0) return ERROR_ALLOC_MEM;
    if (D.kind == 2 && !g_oneshot_upload && build_xtile(H, D) < 0) return ERROR_ALLOC_MEM;
    if (upload_ja16(D, H.ia.data(), H.ja.data()) < 0) return ERROR_ALLOC_MEM;
    return build_estream(H.ia.data(), H.row, H.nnz, D);
}

// development knobs (fasp_hip_tune): -1 = automatic
struct Tuning { int gen2 = 2, ws2_bpc = 3, maxgrid = -1, xcd = 16, nt = 1, kind = -1, lanes = -1, wrows = -1, wcap = -1, compress = 1, rpl = -1, lds_tab = 1, xcd_pat = 64, spcg_batch = 16, small_lds = 1, ja16 = 1, spcg_fused = 1, spcg_grid = 0, spcg_persist = 1, split_rows = 0, gs_multicolor = 0, seq_flow = 1, seq_strip_kb = 0, seq_jobs = 1, seq_spine = 1, seq_grid = 0, seq_chain = 1, seq_chain_n1 = 0, seq_chain_grid = 0, seq_chain_ref = 0, seq_test_hang = 0, seq_rest_lanes = 0, local_square = 1, fuse_zr = 1, fuse_presmooth = 1, seq_lanes = 0, xtile = 1, rp5_max = 45, rp_bpc = 5, rp_xcd = -1, rp_strip = 2, spcg_test_hang = 0, small_onewave = 4, lazy_coarse = 1, rp_stream = -1, renumber = 1, renumber_chunk = 262144, pcg_dev_beta = 1, spcg_spec = 1, ev_every = 4, pcg_fold = 1, seq_chain_touch = 8, seq_chain_touch_t1 = 1, seq_zero_skip = 1, estream = 1, es_dbg = 0; };
static Tuning g_tune;

// Blocks of one kernel instantiation that are co-resident on a CU (VGPR / LDS / wave
// limits), from the occupancy API, cached per instantiation.  The persistent grids
// below are sized to exactly this residency: a grid larger than what is resident
// serialises into two rounds (measured: +35 % time), a smaller one leaves CUs idle.
template <class K>
static int resident_blocks_per_cu(K kernel)
{
    // keyed by the kernel's ADDRESS: K is only the signature, which all CsrArgs kernels share (round 1 cached one
    // value per signature -- whatever kernel ran first decided the grid of all the others)
    static std::vector<std::pair<const void*, int>> cache;
    const void* key = reinterpret_cast<const void*>(kernel);
    for (const auto& e : cache)
        if (e.first == key) return e.second;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, BLOCK, 0) != hipSuccess || nb < 1) nb = 4;
    nb = std::min(nb, 8);
    cache.emplace_back(key, nb);
    return nb;
}

template <class K>
static int launch_persistent(K kernel, int ntiles, CsrArgs& a, int blocks_per_cu = 0)
{
    int cap = (blocks_per_cu > 0 ? std::min(blocks_per_cu, resident_blocks_per_cu(kernel)) : resident_blocks_per_cu(kernel)) * g_ctx.num_cu;
    if (g_tune.maxgrid > 0) cap = g_tune.maxgrid;
    cap = std::min(cap, MAXGRID);
    int grid = std::min(cap, ntiles);
    grid = std::max(8, (grid + 7) / 8 * 8);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK), 0, g_ctx.stream, a);
    return grid;
}

// (the transfer-operator kernel has no smoother instantiations)
template <int OP>
static int launch_rowpat5(CsrArgs& a)
{
    if constexpr (OP == OP_JACOBI || OP == OP_L1DIAG) { (void)a; return 0; }
    else return launch_persistent(k_csr_rowpat5<OP>, a.ntiles, a, 5);
}

// Launches the row kernel of family M.kind for operation OP; returns the grid size
// (= number of per-block partials written by OP_MXV_DOT).
// Row window of a launch: rows [lo, hi) (lo a multiple of WIN_ALIGN, which every kernel's tile size divides),
// per-block partials written from slot goff on.  hi < 0: the whole operator.
constexpr int WIN_ALIGN = DIST_WIN_ALIGN;
struct RowWin { int lo = 0, hi = -1, goff = 0; };
// OP_JACOBI with a.partials set asks for the partials of (x_new, b) on the way out (the (z, r) of PCG from the last
// sweep of level 0).  Only the kernels of the fast paths do it; a launch that did sets this flag.
static bool g_jacobi_dot_done = false;

// numbering bridge of a transfer operator (DevCSR::bridge): gather behind a restriction -- with the next level's first Jacobi sweep from
// the zero guess written along, zx_store's expression -- and scatter in front of a prolongation
__global__ __launch_bounds__(BLOCK) void k_bridge_gather(int n, const int* __restrict__ perm, const double* __restrict__ s, double* __restrict__ y,
                                                         double* __restrict__ zx, const double* __restrict__ zdiag, double zomega)
{
    for (int k = blockIdx.x * BLOCK + threadIdx.x; k < n; k += gridDim.x * BLOCK) {
        const double v = s[perm[k]];
        y[k] = v;
        if (zx) { const double di = zdiag[k]; zx[k] = (fabs(di) > 1e-20) ? (1 - zomega) * 0.0 + zomega * v / di : 0.0; }
    }
}
__global__ __launch_bounds__(BLOCK) void k_bridge_scatter(int n, const int* __restrict__ perm, const double* __restrict__ x, double* __restrict__ s)
{
    for (int k = blockIdx.x * BLOCK + threadIdx.x; k < n; k += gridDim.x * BLOCK) s[perm[k]] = x[k];
}

template <int OP>
static int launch_csr(const DevCSR& M0, CsrArgs a, RowWin win = RowWin())
{
    if (M0.bridge && win.hi < 0) {
        DevCSR Mn = M0;   // (shallow: the operator without its bridge)
        Mn.bridge = nullptr; Mn.bscratch = nullptr; Mn.bridge_dir = 0;
        if (M0.bridge_dir == 1 && OP == OP_MXV) {
            CsrArgs b = a;
            b.y = M0.bscratch; b.zx = nullptr; b.zdiag = nullptr;
            const int G = launch_csr<OP>(Mn, b);
            const int grid = std::max(1, std::min(MAXGRID, (M0.row + BLOCK - 1) / BLOCK));
            hipLaunchKernelGGL(k_bridge_gather, dim3(grid), dim3(BLOCK), 0, g_ctx.stream, M0.row, M0.bridge, (const double*)M0.bscratch, a.y, a.zx, a.zdiag, a.zomega);
            return G;
        }
        if (M0.bridge_dir == 2 && (OP == OP_ADD || OP == OP_SUB || OP == OP_AXPY || OP == OP_MXV)) {
            const int grid = std::max(1, std::min(MAXGRID, (M0.col + BLOCK - 1) / BLOCK));
            hipLaunchKernelGGL(k_bridge_scatter, dim3(grid), dim3(BLOCK), 0, g_ctx.stream, M0.col, M0.bridge, a.x, M0.bscratch);
            CsrArgs b = a;
            b.x = M0.bscratch;
            return launch_csr<OP>(Mn, b);
        }
        std::fprintf(stderr, "### ERROR: fasp_hip: operation %d on a transfer operator with a numbering bridge (direction %d)\n", (int)OP, M0.bridge_dir);
        std::abort();
    }
    if (win.hi < 0 && g_tune.split_rows > 0 && M0.row >= 4 * WIN_ALIGN) {
        // test mode (fasp_hip_tune("split_rows", k)): every operator in three row windows, as dist_launch issues them
        const int lo = std::min(g_tune.split_rows, M0.row / 4) / WIN_ALIGN * WIN_ALIGN, hi = (M0.row - lo) / WIN_ALIGN * WIN_ALIGN;
        RowWin w; w.lo = lo; w.hi = hi; w.goff = 0;
        int G = launch_csr<OP>(M0, a, w);
        w.lo = 0; w.hi = lo; w.goff = G;
        G += launch_csr<OP>(M0, a, w);
        w.lo = hi; w.hi = M0.row; w.goff = G;
        return G + launch_csr<OP>(M0, a, w);
    }
    DevCSR M = M0;  // shallow copy: tuning overrides
    const int w_lo = win.hi >= 0 ? win.lo : 0, w_hi = win.hi >= 0 ? std::min(win.hi, M0.row) : M0.row;
    if (w_hi <= w_lo && win.hi >= 0) return 0;
    if (a.partials) a.partials += win.goff;
    auto set_tiles = [&](int rpb) {
        // a row window starts at a multiple of WIN_ALIGN; a kernel whose tile size does not divide it (only reachable through
        // fasp_hip_tune: block-stream tiles beyond 1024 rows, k_csr_rowpat with 8 rows per lane) would compute rows outside
        // the window a second time -- refuse loudly instead
        if (win.hi >= 0 && rpb > 0 && (w_lo % rpb) != 0) {
            std::fprintf(stderr, "### ERROR: fasp_hip: row window at %d with a kernel tile of %d rows (a tuning knob broke the window alignment)\n", w_lo, rpb);
            std::abort();
        }
        a.nrow = w_hi; a.row_lo = w_lo; a.tile0 = w_lo / rpb;
        a.ntiles = (w_hi - w_lo + rpb - 1) / rpb;
        a.tiles_per_xcd = (a.ntiles + 7) / 8;
    };
    // XCD strips (kernels.hip.h, tile_of, xcd_map == -2) for a kernel with tiles of `rpb` rows, where the operator knows its grid plane
    auto strips = [&](int rpb) {
        if (!g_tune.rp_strip || M.plane < 8 * rpb || M.plane % (8 * rpb) != 0 || a.ntiles < 4 * (M.plane / rpb)) return false;
        a.xcd_map = -2; a.tpp = M.plane / rpb; a.tiles_per_xcd = a.tpp / 8;
        return true;
    };
    if (M.code && g_tune.compress) M.kind = 4;  // dictionary-coded copy present: one byte per entry
    if (M.pat && g_tune.compress) M.kind = 5;   // row-pattern-coded copy present: two bytes per row
    if (g_tune.kind >= 0 && !(g_tune.kind == 4 && !M.code) && !(g_tune.kind == 5 && !M.pat)) M.kind = g_tune.kind;
    if (g_tune.lanes > 0) M.lanes = g_tune.lanes;
    if (g_tune.wrows > 0) M.wrows = g_tune.wrows;
    if (g_tune.wcap > 0) M.wcap = g_tune.wcap;
    const bool lstream_ok = g_tune.gen2 && M.kind == 2 && M.wrows == 64 && M.wcap == 512 && (double)M.nnz <= 7.6 * M.row;  // tests c != r itself
    if (OP == OP_JACOBI && M.kind != 0 && M.kind < 4 && (M.dup_diag || !M.dpos) && !lstream_ok) M.kind = 0;  // needs the c != r test
    a.xcd_map = g_tune.xcd;
    a.nt = g_tune.nt;
    // coded pair kernels: streaming hints on y / pattern ids / b where a vector does not fit the Infinity Cache beside the
    // others anyway (P7(256): level 0 yes -- 0.546 -> 0.519 GB, 93 -> 88 us; level 1, 67 MB vectors, no: 63 -> 73 us with hints)
    if (g_tune.rp_stream > 0 || (g_tune.rp_stream < 0 && (size_t)M.row * 8 > (size_t)96 << 20)) a.nt |= 4;
    a.nrow = M.row; a.ia = M.ia; a.ja = M.ja; a.val = M.val; a.dpos = M.dpos;
    a.ja16 = g_tune.ja16 ? M.ja16 : nullptr;
    a.jbase = g_tune.ja16 ? M.jbase : nullptr;
    if (M.jbase && M.kind != 0) { a.ja16 = nullptr; a.jbase = nullptr; }   // (a tuning knob sent a sub-wavefront operator elsewhere: only k_csr_rows adds the row base)
    if (M.kind == 1 || M.kind == 3) M.kind = 0;   // (block-level stream, one workgroup per row: measured slower, retired to tools/lab/)
    const int rpb = M.kind >= 4 ? BLOCK : M.kind == 2 ? 4 * M.wrows : BLOCK / M.lanes;
    set_tiles(rpb);
    if (M.kind == 5 && g_tune.gen2 && M.nxrows >= 0 && !M.rowbase && g_tune.rpl <= 0) {
        // square row-pattern-coded operator: scalar-pattern sweep, other rows through the wave's LDS queue (kernels2.hip.h)
        a.pat = M.pat; a.pstart = M.pstart; a.plen = M.plen; a.poff = M.poff; a.pval = M.pval;
        a.npat = M.npat; a.npent = M.npent; a.ncol = M.col; a.rowbase = nullptr;
        set_tiles(2 * BLOCK);
        a.xcd_map = a.ntiles >= 8 * 64 ? g_tune.rp_xcd : 16;  // slabs: x is fetched once per XCD (PMC: 0.18 GB instead of 0.45 GB per level-0 pass)
        // strips instead of slabs where the operator says how long a grid plane is (kernels.hip.h, tile_of)
        if (a.xcd_map == -1) strips(2 * BLOCK);
        if (OP == OP_JACOBI && a.partials) g_jacobi_dot_done = true;
        // (strips: three blocks per CU -- six planes of an XCD's strip in flight -- measured against two, four, five: level-0 t = A p
        // 87-98 us against 95-105 with five; with four, 108)
        return launch_persistent(k_csr_rowpat4<OP>, a.ntiles, a, a.xcd_map == -2 && g_tune.rp_bpc == 5 ? 3 : g_tune.rp_bpc);
    }
    // (measured on P7(256) level 0: prolongation, 1-6 entries per row, 140 -> 131 us; restriction, 7-13 entries per row,
    // 68 -> 80 us: the sweep pays for short lists only)
    if (M.kind == 5 && g_tune.gen2 >= 2 && M.nxrows >= 0 && M.rowbase && g_tune.rpl <= 0 && OP != OP_JACOBI && OP != OP_L1DIAG &&
        (double)M.nnz <= 0.1 * g_tune.rp5_max * M.row) {
        // rectangular row-pattern-coded operator (R, P of the coded levels): pair-of-patterns sweep + LDS queue
        a.pat = M.pat; a.pstart = M.pstart; a.plen = M.plen; a.poff = M.poff; a.pval = M.pval;
        a.npat = M.npat; a.npent = M.npent; a.ncol = M.col; a.rowbase = M.rowbase;
        set_tiles(2 * BLOCK);
        a.xcd_map = a.ntiles >= 8 * 64 ? -1 : 16;
        if (g_tune.rp_strip >= 2) strips(2 * BLOCK);
        return launch_rowpat5<OP>(a);
    }
    if (M.kind == 5) {
        a.pat = M.pat; a.pstart = M.pstart; a.poff = M.poff; a.pval = M.pval; a.rowbase = M.rowbase;
        a.npat = M.npat; a.npent = M.npent;
        const double avg = M.row > 0 ? (double)M.nnz / M.row : 1.0;
        const bool lds = M.npat <= 512 && M.npent <= 2048 && g_tune.lds_tab != 0;
        const int rpl = g_tune.rpl > 0 ? g_tune.rpl : 1;
        a.plen = M.plen; a.ncol = M.col;
        if (g_tune.xcd_pat != 0) a.xcd_map = g_tune.xcd_pat;
        set_tiles(BLOCK * rpl);
        if (g_tune.xcd_pat == 64 && g_tune.rp_strip >= 2) strips(BLOCK * rpl);
        (void)avg;
        if (lds && M.npat <= 64 && M.npent <= 512 && g_tune.lds_tab != 3) {  // small table: more resident blocks
            if (rpl == 1) return launch_persistent(k_csr_rowpat<OP, 2, 1>, a.ntiles, a);
            return launch_persistent(k_csr_rowpat<OP, 2, 2>, a.ntiles, a);
        }
        if (lds) {
            if (rpl == 1) return launch_persistent(k_csr_rowpat<OP, 1, 1>, a.ntiles, a);
            return launch_persistent(k_csr_rowpat<OP, 1, 2>, a.ntiles, a);
        }
        if (rpl == 1) return launch_persistent(k_csr_rowpat<OP, 0, 1>, a.ntiles, a);
        return launch_persistent(k_csr_rowpat<OP, 0, 2>, a.ntiles, a);
    }
    if (M.kind == 4) {
        a.code = M.code; a.rowbase = M.rowbase; a.doff = M.doff; a.dval = M.dval;
        const double avg = M.row > 0 ? (double)M.nnz / M.row : 1.0;
        if (avg <= 8.5) return launch_persistent(k_csr_dict8<OP, 8>, a.ntiles, a);
        if (avg <= 20.0) return launch_persistent(k_csr_dict8<OP, 16>, a.ntiles, a);
        return launch_persistent(k_csr_dict8<OP, 24>, a.ntiles, a);
    }
    if (M.kind == 2 && g_tune.gen2 && M.wrows == 64 && M.wcap == 512 && (double)M.nnz <= 7.6 * M.row) {
        // short rows (64 rows fit the 512-entry slab with room for ragged tiles): 16-byte staged stream, lane = row
        if (OP == OP_JACOBI && a.partials) g_jacobi_dot_done = true;
        return launch_persistent(k_csr_lstream<OP, 512>, a.ntiles, a, 4);
    }
    if (M.kind == 2 && g_tune.gen2 >= 2 && g_tune.xtile && M.lja16 && M.wrows == 64 && M.wcap == 512 && (OP != OP_JACOBI || (M.dpos && !M.dup_diag)))
    {
        // mid levels (20-60 nonzeros per row): the tile's distinct x entries staged in LDS, 16-bit column positions
        a.lja16 = M.lja16; a.tptr = M.tptr; a.tcols = M.tcols;
        if (OP == OP_JACOBI && a.partials) g_jacobi_dot_done = true;
        return launch_persistent(k_csr_xtile<OP>, a.ntiles, a, 3);
    }
    if (M.kind == 2 && g_tune.gen2 >= 2 && M.wrows == 64 && M.wcap == 512 && (OP != OP_JACOBI || (M.dpos && !M.dup_diag)))
    {
        if (OP == OP_JACOBI && a.partials) g_jacobi_dot_done = true;
        return launch_persistent(k_csr_wstream2<OP>, a.ntiles, a, g_tune.ws2_bpc);   // rows of any length: staged, prefetched stream
    }
    if (M.kind == 2) {
        if (M.wrows == 64 && M.wcap == 512) return launch_persistent(k_csr_wstream<OP, 64, 512>, a.ntiles, a);
        if (M.wrows == 64) return launch_persistent(k_csr_wstream<OP, 64, 1024>, a.ntiles, a);
        if (M.wrows == 32 && M.wcap == 512) return launch_persistent(k_csr_wstream<OP, 32, 512>, a.ntiles, a);
        return launch_persistent(k_csr_wstream<OP, 32, 1024>, a.ntiles, a);
    }
    // long rows with 16-bit columns, whole-operator launches: the entry-parallel stream (kernels3.hip.h).  Not for the fused dot
    // products (whichever wave completes a cut row would own its term of the sum) and not for row windows (they keep the row kernel).
    // Where it is the faster one (cold, P7(256): profiles/r06_estream.txt): mean rows below 256 entries -- levels 3 and 4 there, 66 -> 50 and
    // 51 -> 48 us per product; on the longer rows the two tie (both sit on the gather rate of the texture-address pipe) and the row
    // kernel's epilogue is lighter.  fasp_hip_tune("estream", 2): wherever the tables exist (tests, A/B runs); 0: never.
    const bool es_rule = g_tune.estream >= 2 || (g_tune.estream == 1 && (double)M.nnz < 256.0 * M.row);
    if (M.es_tab && es_rule && a.ja16 && win.hi < 0 && g_tune.split_rows <= 0 && OP != OP_MXV_DOT && !(OP == OP_JACOBI && (a.partials || M.dup_diag))) {
        const int W = M.es_W, nc = M.es_nc;
        a.es_wc = M.es_tab; a.es_centry = a.es_wc + W + 1; a.es_crow = a.es_centry + nc + 1; a.es_hw0 = a.es_crow + nc + 1; a.es_np = a.es_hw0 + W;
        a.es_part = M.es_part; a.es_cnt = reinterpret_cast<unsigned*>(M.es_part + 2 * (size_t)W);
        a.es_cbase = M.es_ja16 ? a.es_np + W : nullptr; a.es_ja16 = M.es_ja16;
        // lanes per row: so that the rows a 512-entry chunk touches normally fit one pass of 64 / L rows (kernels3.hip.h)
        const double avg = M.row > 0 ? (double)M.nnz / M.row : 1.0;
        const int lanes = avg >= 512.0 ? 32 : avg >= 256.0 ? 16 : avg >= 128.0 ? 8 : 4;
        static bool attr_set = false;
#define ES_ATTR(LL) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_csr_estream<LL, OP>), hipFuncAttributeMaxDynamicSharedMemorySize, ES_LDS_BYTES)
        if (!attr_set) { ES_ATTR(32); ES_ATTR(16); ES_ATTR(8); ES_ATTR(4); attr_set = true; }
#undef ES_ATTR
#define ES_LAUNCH(LL) hipLaunchKernelGGL((k_csr_estream<LL, OP>), dim3(W / 4), dim3(BLOCK), ES_LDS_BYTES, g_ctx.stream, a)
#ifdef FASP_LAB_DEBUG
        if (OP == OP_MXV && g_tune.es_dbg) {   // (tools/lab: which part of the kernel costs what)
#define ES_DBG(LL, DD) do { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_csr_estream<LL, OP_MXV, DD>), hipFuncAttributeMaxDynamicSharedMemorySize, ES_LDS_BYTES); hipLaunchKernelGGL((k_csr_estream<LL, OP_MXV, DD>), dim3(W / 4), dim3(BLOCK), ES_LDS_BYTES, g_ctx.stream, a); } while (0)
            const int dd = g_tune.es_dbg;
            if (lanes == 32) { if (dd == 1) ES_DBG(32, 1); else if (dd == 2) ES_DBG(32, 2); else if (dd == 3) ES_DBG(32, 3); else if (dd == 4) ES_DBG(32, 4); else ES_DBG(32, 7); }
            else if (lanes == 16) { if (dd == 1) ES_DBG(16, 1); else if (dd == 2) ES_DBG(16, 2); else if (dd == 3) ES_DBG(16, 3); else if (dd == 4) ES_DBG(16, 4); else ES_DBG(16, 7); }
            else if (lanes == 8) { if (dd == 1) ES_DBG(8, 1); else if (dd == 2) ES_DBG(8, 2); else if (dd == 3) ES_DBG(8, 3); else if (dd == 4) ES_DBG(8, 4); else ES_DBG(8, 7); }
            else { if (dd == 1) ES_DBG(4, 1); else if (dd == 2) ES_DBG(4, 2); else if (dd == 3) ES_DBG(4, 3); else if (dd == 4) ES_DBG(4, 4); else ES_DBG(4, 7); }
#undef ES_DBG
            return W / 4;
        }
#endif
        static const bool es_log = std::getenv("FASP_HIP_ES_LOG") != nullptr;   // (debugging: every launch announced and waited for)
        if (es_log) { std::fprintf(stderr, "[es rank %d] op %d rows %d cols %d nnz %d W %d chunks %d lanes %d rel %d x %p y %p b %p\n", comm_rank(), (int)OP, M.row, M.col, M.nnz, W, nc, lanes, M.es_ja16 != nullptr, (const void*)a.x, (void*)a.y, (const void*)a.b); std::fflush(stderr); }
        if (lanes == 32) ES_LAUNCH(32); else if (lanes == 16) ES_LAUNCH(16); else if (lanes == 8) ES_LAUNCH(8); else ES_LAUNCH(4);
#undef ES_LAUNCH
        if (es_log) { const hipError_t e = hipStreamSynchronize(g_ctx.stream); std::fprintf(stderr, "[es rank %d] done: %s\n", comm_rank(), hipGetErrorString(e)); std::fflush(stderr); }
        return W / 4;
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
    if (dist && comm_size() > 1 && comm_allreduce(out, nq, maxmask, g_ctx.stream) < 0) comm_mark_failed();  // seen by fetch_red
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
    if (comm_failed()) {  // an all-reduce or halo exchange failed: the scalars are local, unreduced values
        std::fprintf(stderr, "### ERROR: fasp_hip: rank %d: a collective failed; the iteration is abandoned\n", comm_rank());
        return ERROR_MISC;
    }
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

