// retired_kernels.hip.h -- round-1 / round-2 kernels that no product path selects any more (kept for the kernel
// laboratory's A/B tables only; included by lab.hip behind the solver translation unit).
#pragma once
namespace fasp {

// ---------------------------------------------------------------------------
// One workgroup per row: for the few, very long rows of the coarsest levels (1000+ nnz
// per row, < 64 K rows) a wavefront per row leaves the chip under-filled and serialises 20
// dependent load rounds; with 256 lanes per row every lane issues its whole share of the
// row (<= 8 loads) at once and the kernel is one memory round trip deep.
// ---------------------------------------------------------------------------
template <int OP>
__global__ __launch_bounds__(BLOCK) void k_csr_blockrow(CsrArgs a)
{
    __shared__ double lds[4];
    double acc = 0.0;
    for (int r = a.row_lo + blockIdx.x; r < a.nrow; r += gridDim.x) {
        const int kb = a.ia[r], ke = a.ia[r + 1];
        double s = 0.0;
        for (int base = kb + threadIdx.x; base < ke; base += 8 * BLOCK) {
            int    c[8];
            double v[8], xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int  k = base + u * BLOCK;
                const bool ok = k < ke;
                c[u] = ok ? ld_ja(a, k) : 0;
                v[u] = ok ? ld_val(a, k) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[u] = a.x[c[u]];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = base + u * BLOCK;
                if (k < ke && (OP != OP_JACOBI || c[u] != r)) s += v[u] * xv[u];
            }
        }
        s = block_sum(s, lds);
        if (threadIdx.x == 0) {
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
    if (OP == OP_MXV_DOT && threadIdx.x == 0) a.partials[blockIdx.x] = acc;
}

// ---------------------------------------------------------------------------
// CSR "stream" kernel for short rows (<= ~48 nnz/row: the fine levels, R and P, which
// hold 3/4 of all nonzeros).  A block owns a tile of R consecutive rows (R = 256..1024):
//   phase 1  all 256 threads sweep the tile's contiguous span of val/JA with unit-stride
//            loads (fully coalesced whatever the row lengths), gather x through L2 and
//            park the products val[k]*x[JA[k]] in LDS, <= CAP entries per chunk;
//   phase 2  thread i sums the products of rows i, i+256, ... from LDS in storage order,
//            starting from 0.0 (or from b_i for Jacobi): the same left-to-right sum as the
//            reference's scalar loops (BlaSpmvCSR.c:414-416, ItrSmootherCSR.c:151-160), so
//            row results are bit-identical to the reference's;
//   epilogue consecutive threads own consecutive rows: b, diag, x, y are all coalesced.
// ---------------------------------------------------------------------------
constexpr int STREAM_CAP  = 4096;  // products per chunk (32 KiB of LDS)

template <int OP>
__global__ __launch_bounds__(BLOCK) void k_csr_stream(CsrArgs a, int R)
{
    __shared__ double prod[STREAM_CAP];
    __shared__ int    rowptr[STREAM_MAXR + 1];
    __shared__ int    colidx[OP == OP_JACOBI ? STREAM_CAP : 1];
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const int vmax = tile_vmax(a);
    double dotacc = 0.0;

    for (int v = blockIdx.x; v < vmax; v += gridDim.x) {
        const int t = tile_of(a, v);
        if (t >= a.ntiles) continue;
        const int r0 = (t + a.tile0) * R;
        const int nr = min(R, a.nrow - r0);
        for (int i = tid; i <= nr; i += BLOCK) rowptr[i] = a.ia[r0 + i];
        __syncthreads();
        const int k0 = rowptr[0], k1 = rowptr[nr];

        double acc[STREAM_MAXR / BLOCK];
#pragma unroll
        for (int q = 0; q < STREAM_MAXR / BLOCK; ++q) {
            const int i = tid + q * BLOCK;
            acc[q] = ((OP == OP_JACOBI || OP == OP_L1DIAG) && i < nr) ? a.b[r0 + i] : 0.0;
        }

        for (int lo = k0; lo < k1; lo += STREAM_CAP) {
            const int hi = min(lo + STREAM_CAP, k1);
            // phase 1: coalesced sweep of the chunk
            for (int k = lo + tid; k < hi; k += BLOCK) {
                const int    c = ld_ja(a, k);
                const double v = ld_val(a, k);
                prod[k - lo] = v * a.x[c];
                if (OP == OP_JACOBI) colidx[k - lo] = c;
            }
            __syncthreads();
            // phase 2: sequential per-row sums over the part of each row inside the chunk
#pragma unroll
            for (int q = 0; q < STREAM_MAXR / BLOCK; ++q) {
                const int i = tid + q * BLOCK;
                if (i < nr) {
                    const int kb = max(rowptr[i], lo), ke = min(rowptr[i + 1], hi);
                    double s = acc[q];
                    if (OP == OP_JACOBI) {
                        const int r = r0 + i;
                        for (int k = kb; k < ke; ++k)
                            if (colidx[k - lo] != r) s -= prod[k - lo];
                    } else if (OP == OP_L1DIAG) {
                        for (int k = kb; k < ke; ++k) s -= prod[k - lo];
                    } else {
                        for (int k = kb; k < ke; ++k) s += prod[k - lo];
                    }
                    acc[q] = s;
                }
            }
            __syncthreads();
        }
        if (k0 >= k1) __syncthreads();  // rowptr is re-staged by the next tile

        // epilogue: coalesced
#pragma unroll
        for (int q = 0; q < STREAM_MAXR / BLOCK; ++q) {
            const int i = tid + q * BLOCK;
            if (i < nr) {
                const int    r = r0 + i;
                const double s = acc[q];
                if (OP == OP_MXV) { a.y[r] = s; zx_store(a, r, s); }
                else if (OP == OP_RESID) a.y[r] = a.b[r] - s;
                else if (OP == OP_ADD) a.y[r] += s;
                else if (OP == OP_SUB) a.y[r] -= s;
                else if (OP == OP_AXPY) a.y[r] += s * a.alpha;
                else if (OP == OP_JACOBI) {
                    const double d = a.diag[r], xi = a.x[r];  // s == t_i of the reference
                    a.y[r] = (fabs(d) > 1e-20) ? (1 - a.omega) * xi + a.omega * s / d : xi;
                } else if (OP == OP_L1DIAG) {
                    const double d = a.diag[r], xi = a.x[r];  // s == t_i of the reference
                    a.y[r] = l1_or_jacobi_f(a, r, s, d, xi);
                } else if (OP == OP_MXV_DOT) {
                    a.y[r] = s;
                    dotacc += s * a.dotv[r];
                }
            }
        }
    }
    if (OP == OP_MXV_DOT) {
        const double tot = block_sum(dotacc, red);
        if (tid == 0) a.partials[blockIdx.x] = tot;
    }
}


// ---------------------------------------------------------------------------
// k_csr_rowpat2<OP, LDS_TAB>: row-pattern-coded SQUARE matrix (column base = row index), a lane owns rows
// 2i and 2i+1 of its wave's 128-row tile.  Pattern lists are padded to multiples of 8 entries (offset 0,
// value 0); x goes through buffer loads (hardware range check: the second half of a 16-byte load at
// the last column returns 0 and is never used).
//   LDS_TAB 2: table of <= 64 patterns / 512 entries in LDS;  1: <= 512 / 2048;  0: table in global memory
// ---------------------------------------------------------------------------

template <int OP, int LDS_TAB>
__global__ __launch_bounds__(BLOCK) void k_csr_rowpat2(CsrArgs a)
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
    const int vmax = tile_vmax(a);  // tiles of 2 * BLOCK rows
    const int G = gridDim.x;
    const int last = a.nrow - 1;
    const unsigned* pat2 = reinterpret_cast<const unsigned*>(a.pat);  // two 16-bit ids per load
    const int npair = (a.nrow + 1) >> 1;                              // the id array is padded to an even count
    double dotacc = 0.0;

    auto advance = [&](int& v) -> int {  // first row of the next tile of this block, -1: none
        for (;;) {
            if (v >= vmax) return -1;
            const int t = tile_of(a, v);
            v += G;
            if (t < a.ntiles) return (t + a.tile0) * (2 * BLOCK);
        }
    };
    int v = blockIdx.x;
    int r0A = advance(v);
    auto ld_pp = [&](int r0) -> unsigned {
        const unsigned* q = pat2 + min((max(r0, 0) >> 1) + (int)threadIdx.x, npair - 1);
        return (a.nt & 4) ? __builtin_nontemporal_load(q) : *q;
    };
    unsigned pp = ld_pp(r0A);
    while (r0A >= 0) {
        const int r0B = advance(v);
        const unsigned ppB = ld_pp(r0B);
        const int  ra = r0A + 2 * (int)threadIdx.x, rb = ra + 1;
        const bool va = ra <= last, vb = rb <= last;
        const unsigned pidA = pp & 0xffffu, pidB = vb ? (pp >> 16) : pidA;
        const bool same = pidA == pidB;
        const int  psA = pstart[pidA], lenA = va ? plen[pidA] : 0;
        double accA = 0.0, accB = 0.0;
        f64x2_t bb = {0.0, 0.0};
        if (OP == OP_JACOBI || OP == OP_L1DIAG || OP == OP_RESID) {
            if (vb) bb = *reinterpret_cast<const f64x2_t*>(a.b + ra);
            else if (va) bb.x = a.b[ra];
            if (OP != OP_RESID) { accA = bb.x; accB = bb.y; }
        }
        const unsigned baseA = (unsigned)min(ra, last) * 8u;
        for (int k = 0; __any(k < lenA); k += U) {
            int     off[U];
            f64x2_t xv[U];
            const bool live = k < lenA;
#pragma unroll
            for (int u = 0; u < U; ++u) off[u] = live ? poff[psA + k + u] : 0;
#ifdef FASP_LAB_DEBUG
            if (a.nt & 8) { for (int u = 0; u < U; ++u) off[u] = 0; }
            if (a.nt & 16) { for (int u = 0; u < U; ++u) off[u] = abs(off[u]) > 4096 ? 0 : off[u]; }
            if (a.nt & 32) { for (int u = 0; u < U; ++u) off[u] = abs(off[u]) > 1 ? 0 : off[u]; }
#endif
#pragma unroll
            for (int u = 0; u < U; ++u) xv[u] = buf_load_f64x2(xr, baseA + (unsigned)off[u] * 8u);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double w = live ? pval[psA + k + u] : 0.0;
                bool use = k + u < lenA;
                if (OP == OP_JACOBI) use = use && off[u] != 0;
                const double pa = w * xv[u].x, pb = w * xv[u].y;
                const bool neg = (OP == OP_JACOBI || OP == OP_L1DIAG);
                accA = use ? (neg ? accA - pa : accA + pa) : accA;
                accB = (use && same) ? (neg ? accB - pb : accB + pb) : accB;
            }
        }
        if (__any(!same && vb)) {  // rows whose partner has another pattern: the partner's own list, 8-byte loads
            if (!same && vb) {
                const int psB = pstart[pidB], lenB = plen[pidB];
                const unsigned baseB = (unsigned)rb * 8u;
                for (int k = 0; k < lenB; ++k) {
                    const int    of = poff[psB + k];
                    const double w = pval[psB + k];
                    const double pr = w * buf_load_f64(xr, baseB + (unsigned)of * 8u);
                    if (OP == OP_JACOBI) { if (of != 0) accB -= pr; }
                    else if (OP == OP_L1DIAG) accB -= pr;
                    else accB += pr;
                }
            }
        }
        // epilogue: 16-byte accesses when both rows exist
        if (vb) {
            f64x2_t out;
            if (OP == OP_MXV) { out.x = accA; out.y = accB; }
            else if (OP == OP_RESID) { out.x = bb.x - accA; out.y = bb.y - accB; }
            else if (OP == OP_ADD || OP == OP_SUB || OP == OP_AXPY) {
                const f64x2_t y0 = *reinterpret_cast<const f64x2_t*>(a.y + ra);
                if (OP == OP_ADD) { out.x = y0.x + accA; out.y = y0.y + accB; }
                else if (OP == OP_SUB) { out.x = y0.x - accA; out.y = y0.y - accB; }
                else { out.x = y0.x + accA * a.alpha; out.y = y0.y + accB * a.alpha; }
            } else if (OP == OP_JACOBI || OP == OP_L1DIAG) {
                const f64x2_t dg = *reinterpret_cast<const f64x2_t*>(a.diag + ra);
                const f64x2_t xi = *reinterpret_cast<const f64x2_t*>(a.x + ra);
                if (OP == OP_JACOBI) {
                    out.x = (fabs(dg.x) > 1e-20) ? (1 - a.omega) * xi.x + a.omega * accA / dg.x : xi.x;
                    out.y = (fabs(dg.y) > 1e-20) ? (1 - a.omega) * xi.y + a.omega * accB / dg.y : xi.y;
                } else {
                    out.x = l1_or_jacobi_f(a, ra, accA, dg.x, xi.x);
                    out.y = l1_or_jacobi_f(a, rb, accB, dg.y, xi.y);
                }
            } else {  // OP_MXV_DOT
                const f64x2_t dv = *reinterpret_cast<const f64x2_t*>(a.dotv + ra);
                out.x = accA; out.y = accB;
                dotacc += accA * dv.x;
                dotacc += accB * dv.y;
            }
#ifdef FASP_LAB_DEBUG
            if ((a.nt & 64) && out.x != 1.2345e-300) { /* ablation: no store */ } else
#endif
            if (a.nt & 2) __builtin_nontemporal_store(out, reinterpret_cast<f64x2_t*>(a.y + ra));
            else *reinterpret_cast<f64x2_t*>(a.y + ra) = out;
        } else if (va) {
            row_epilogue<OP>(a, ra, accA, dotacc);
        }
        r0A = r0B;
        pp = ppB;
    }
    if (OP == OP_MXV_DOT) {
        const double tot = block_sum(dotacc, red);
        if (threadIdx.x == 0) a.partials[blockIdx.x] = tot;
    }
}

// ---------------------------------------------------------------------------
// k_csr_rowpat3<OP, LDS_TAB>: what the PMC counters say about k_csr_rowpat / k_csr_rowpat2 on level 0 of
// P7(256) -- SQ_ACTIVE_INST_ANY x 8 waves = 92 % of every SIMD's issue slots, 365 instructions per 128
// rows, the same 79 us with every gather redirected to x[r] -- is that they are bound by instruction
// issue, not by memory; and that a store in front of the next step's loads serialises the wave on the
// store's acknowledgement (vmcnt retires in order: 106 -> 79 us without the store).  So:
//   * the pattern of the wave's MIDDLE row is treated as wave-uniform: its offsets and values come from
//     scalar loads, the loop has no selects, no LDS reads, no per-lane address arithmetic beyond one add;
//     rows with another pattern (domain boundaries) are recomputed afterwards by a short divergent tail
//     that walks their own list (table in LDS);
//   * the result of a step is stored AFTER the next step's loads have been issued;
//   * Jacobi takes x_i and a_ii from the entry with offset 0 (no separate loads).
// Two consecutive rows per lane, 16-byte accesses, sums left to right from the exact stored values:
// bit-identical to the plain kernels.
// ---------------------------------------------------------------------------
template <int OP, int LDS_TAB>
__global__ __launch_bounds__(BLOCK) void k_csr_rowpat3(CsrArgs a)
{
    if (a.stop && *a.stop) return;
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
    constexpr bool NEG = (OP == OP_JACOBI || OP == OP_L1DIAG);
    const __amdgpu_buffer_rsrc_t xr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.x), 0, (int)((unsigned)a.ncol * 8u), 0x00020000);
    // the pattern table through the scalar cache (constant address space: read-only for the whole launch)
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
    auto ld_pp = [&](int r0) -> unsigned { return pat2[min((max(r0, 0) >> 1) + (int)threadIdx.x, npair - 1)]; };
    // one row by its own list (divergent tail)
    auto own_row = [&](unsigned pid, int r, double acc0, double& xi, double& dg) -> double {
        const int ps = pstart[pid], len = plen[pid];
        const unsigned base = (unsigned)r * 8u;
        double acc = acc0;
        for (int k = 0; k < len; ++k) {
            const int    of = poff[ps + k];
            const double w = pval[ps + k];
            const double xk = buf_load_f64(xr, base + (unsigned)of * 8u);
            const double pr = w * xk;
            if (OP == OP_JACOBI) { if (of != 0) acc -= pr; else { xi = xk; dg = w; } }
            else if (OP == OP_L1DIAG) acc -= pr;
            else acc += pr;
        }
        return acc;
    };

    int      v = blockIdx.x;
    int      r0A = advance(v);
    unsigned pp = ld_pp(r0A);
    // Deferred result of the previous step.  The store is ONE unconditional buffer store per step (lanes
    // with nothing to store carry an out-of-range offset, which the hardware drops): a store under a
    // branch makes the compiler wait vmcnt(0) for the loads issued in front of it.
    const __amdgpu_buffer_rsrc_t yr =
        __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)((unsigned)(a.nrow & ~1) * 8u), 0x00020000);
    f64x2_t  pend_out = {0.0, 0.0};
    unsigned pend_off = 0xfffffff0u;
    auto flush = [&]() {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, pend_out), yr, (int)pend_off, 0, 0);
        pend_off = 0xfffffff0u;
    };
    while (r0A >= 0) {
        const int      r0B = advance(v);
        const unsigned ppB = ld_pp(r0B);
        const int  ra = r0A + 2 * (int)threadIdx.x, rb = ra + 1;
        const bool va = ra <= last, vb = rb <= last;
        const unsigned pidA = pp & 0xffffu, pidB = vb ? (pp >> 16) : pidA;
        // the wave's middle row decides the pattern every lane runs in the scalar pass
        const unsigned dom = (unsigned)__builtin_amdgcn_readlane((int)pidA, 32);
        const int      dps = cpstart[dom], dlen = cplen[dom];
        const unsigned baseA = (unsigned)min(ra, last) * 8u;
        f64x2_t bb = {0.0, 0.0}, aux = {0.0, 0.0};
        if (OP == OP_JACOBI || OP == OP_L1DIAG || OP == OP_RESID) {
            if (vb) bb = *reinterpret_cast<const f64x2_t*>(a.b + ra);
            else if (va) bb.x = a.b[ra];
        }
        if (OP == OP_MXV_DOT) {
            if (vb) aux = *reinterpret_cast<const f64x2_t*>(a.dotv + ra);
            else if (va) aux.x = a.dotv[ra];
        } else if (OP == OP_ADD || OP == OP_SUB || OP == OP_AXPY) {
            if (vb) aux = *reinterpret_cast<const f64x2_t*>(a.y + ra);
            else if (va) aux.x = a.y[ra];
        } else if (OP == OP_L1DIAG) {
            if (vb) aux = *reinterpret_cast<const f64x2_t*>(a.diag + ra);
            else if (va) aux.x = a.diag[ra];
        }
        double accA = NEG ? bb.x : 0.0, accB = NEG ? bb.y : 0.0;
        double xiA = 0.0, xiB = 0.0, dgA = 0.0, dgB = 0.0;
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
                    else { xiA = xv[u].x; xiB = xv[u].y; dgA = dgB = w[u]; }
                } else if (OP == OP_L1DIAG) { accA -= pa; accB -= pb; }
                else { accA += pa; accB += pb; }
            };
            // padding entries contribute nothing, not even a signed zero: straight-line code per length
            switch (min(dlen - k, 8)) {
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
        // a 16-byte load that starts below x[0] is dropped as a whole: whenever row A is not of the wave's
        // pattern, row B is recomputed too
        const bool exA = va && pidA != dom, exB = vb && (pidB != dom || pidA != dom);
        if (__any(exA || exB)) {
            if (exA) accA = own_row(pidA, ra, NEG ? bb.x : 0.0, xiA, dgA);
            if (exB) accB = own_row(pidB, rb, NEG ? bb.y : 0.0, xiB, dgB);
        }
        f64x2_t out = {0.0, 0.0};
        if (OP == OP_MXV) { out.x = accA; out.y = accB; }
        else if (OP == OP_RESID) { out.x = bb.x - accA; out.y = bb.y - accB; }
        else if (OP == OP_ADD) { out.x = aux.x + accA; out.y = aux.y + accB; }
        else if (OP == OP_SUB) { out.x = aux.x - accA; out.y = aux.y - accB; }
        else if (OP == OP_AXPY) { out.x = aux.x + accA * a.alpha; out.y = aux.y + accB * a.alpha; }
        else if (OP == OP_JACOBI) {
            if (__any((va && dgA == 0.0) || (vb && dgB == 0.0))) {  // a row without a stored diagonal: x_i from memory
                if (va && dgA == 0.0) xiA = a.x[ra];
                if (vb && dgB == 0.0) xiB = a.x[rb];
            }
            out.x = (fabs(dgA) > 1e-20) ? (1 - a.omega) * xiA + a.omega * accA / dgA : xiA;
            out.y = (fabs(dgB) > 1e-20) ? (1 - a.omega) * xiB + a.omega * accB / dgB : xiB;
        } else if (OP == OP_L1DIAG) {
            f64x2_t xi = {0.0, 0.0};
            if (vb) xi = *reinterpret_cast<const f64x2_t*>(a.x + ra);
            else if (va) xi.x = a.x[ra];
            out.x = va ? l1_or_jacobi_f(a, ra, accA, aux.x, xi.x) : 0.0;
            out.y = vb ? l1_or_jacobi_f(a, rb, accB, aux.y, xi.y) : 0.0;
        } else {  // OP_MXV_DOT
            out.x = accA; out.y = accB;
            if (va) dotacc += accA * aux.x;
            if (vb) dotacc += accB * aux.y;
        }
        if (vb) { pend_out = out; pend_off = (unsigned)ra * 8u; }
        else if (va) a.y[ra] = out.x;  // the last row of a matrix with an odd row count
        r0A = r0B;
        pp = ppB;
    }
    flush();
    if (OP == OP_MXV_DOT) {
        const double tot = block_sum(dotacc, red);
        if (threadIdx.x == 0) a.partials[blockIdx.x] = tot;
    }
}


}  // namespace fasp
