/* fasp_hip_dev.h -- measurement and test entries of libfasp_hip.so.
 *
 * NOT part of the drop-in boundary (include/fasp_hip.h): nothing a FASP application calls is declared here.  These are the
 * entries bench.py, tools/ and tests/ use to time single kernels on a resident hierarchy, to switch kernel families for the
 * bit-identity A/B tests, to inspect the row partition and to run the host-side self-tests.  They live in the shipped library
 * because the driver-run tests and the benchmark load exactly that library.
 */
#ifndef FASP_HIP_DEV_H
#define FASP_HIP_DEV_H
#include "fasp_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Timed micro-benchmark of one device kernel class on the resident level-0
 * matrix: returns mean milliseconds per launch over `reps` launches measured
 * with HIP events on the launch stream.  kind: 0 = SpMV, 1 = aAxpy(-1),
 * 2 = Jacobi sweep, 3 = dot, 4 = axpy. */
double fasp_hip_time_kernel(fasp_hip_amg* h, int kind, int level, int reps);
/* development / test entry: one operator through the resident upload path (coding + kernel selection), ms per launch */
double fasp_hip_time_matrix(const dCSRmat* A, int op, int reps, int* kind_out);
/* test entry (host only, no GPU): build the sweep schedule of the rows seq[0..ns) of A (csrc/seq_sched.cpp) and walk it on the host as the
 * device kernels do, against the plain sequential Gauss-Seidel sweep: largest deviation relative to the largest entry; < 0: error
 * (-2: a row reads more earlier rows than a strip holds: no split form).  spine: -1 / 1 where the schedule chooses it, 0 never, 2 wherever
 * a row has two lanes; info (may be NULL): {lanes per row, rounds, spine rounds, virtual rows, strips, chunks} of the schedule */
double fasp_hip_seq_schedule_selftest(const dCSRmat* A, const int* seq, int ns, int strip_kb, int lanes, int spine, int* info);
/* test entry (host only, no GPU): the CHAIN form of the same sweep (csrc/seq_chain.hip.h, round 5: blocked substitution, the dependency
 * chain inside one wavefront) built wherever it applies and walked on the host as k_tri_chain_ref does, with the update formula `form`
 * (0: t * (1 / a_ii) -- fasp_smoother_dcsr_gs; 1: t / a_ii -- the C/F-ordered sweep; 2: SOR with weight w), against the plain sequential
 * sweep; < 0: error (-2: the form does not apply).  n1_blocks: blocks of 64 rows in tier 1 (0: chosen).  info (may be NULL, 10 ints):
 * {blocks, tier-1 blocks, x ring, G ring, tier-1 steps, tier-2 steps, band entries, tier-1 entries, tier-2 entries, dependency classes};
 * out_u (may be NULL, max(row, col) doubles): the swept vector (input: u_i = sin(0.37 i) + 0.1, b_i = cos(0.11 i)) */
/* test entries (host only, no GPU): the brick renumbering of the uncoded mid levels (csrc/reorder.cpp, round 5).  fasp_hip_cluster_order:
 * order[k] = old index of the row that gets the new index k -- breadth-first balls of 64 rows grown inside chunks of `chunk` consecutive rows.
 * fasp_hip_permute_csr: row k of the result is row rperm[k] of A (NULL: rows keep their numbers), column j becomes cinv[j] (NULL: kept); the
 * entries of a row keep their storage order.  ia / ja / val: caller's arrays of row + 1 / nnz / nnz entries. */
int fasp_hip_cluster_order(const dCSRmat* A, int chunk, int* order);
int fasp_hip_permute_csr(const dCSRmat* A, const int* rperm, const int* cinv, int* ia, int* ja, double* val);
double fasp_hip_seq_chain_selftest(const dCSRmat* A, const int* seq, int ns, int n1_blocks, int form, double w, int* info, double* out_u);
/* measured device ceilings reported beside the roofline: out[0..2] = GB/s of a 16-byte-per-lane read, copy and
 * triad over buffers of `bytes` each (>= 512 MiB: beyond the Infinity Cache) */
int fasp_hip_measure_ceilings(double* out, size_t bytes, int reps);

/* Run-time switches (A/B tests, profiling, and ONE behavioural mode):
 *   kernel selection / launch geometry: maxgrid, xcd, nt, kind, lanes, wrows, wcap (-1 = automatic), gen2 (0 round-1
 *     kernels, 1, 2 = default), compress (lossless matrix coding on/off), ja16, ws2_bpc, rpl, lds_tab, xcd_pat, rp_strip (coded operators
 *     of a 3-D grid: an XCD sweeps a strip of every grid plane -- 1: the square operators, 2 (default): the transfer operators too -- or,
 *     0, a slab of planes), estream (k_csr_estream, the entry-parallel kernel of long-row operators: 1 (default) = where it measured
 *     faster -- mean rows of fewer than 256 entries --, 2 = wherever its tables exist, 0 = the row kernels);
 *   coarse solve: spcg_persist, spcg_fused, spcg_batch, spcg_grid, small_lds, small_onewave (coarsest levels of <= 128
 *     rows: 4 (default) = matrix in registers as 16 x 16 blocks, the direction broadcast inside the multiply-adds, the next
 *     direction sent before the tests of the iteration (k_spcg_dpp<.., true>); 3 = the same without sending ahead; 2 = matrix
 *     in registers, four wavefronts, p broadcast from LDS; 1 = dense in LDS, one wavefront; 0 = the general kernel),
 *     lazy_coarse (the one-launch solvers' verdicts read once per application of the preconditioner, default 1;
 *     2 = replay every first application as if a coarse solve had given up: tests), coarse_mode / coarse_split_min
 *     (multi-GPU: replicated levels computed in row windows + all-gather);
 *   upload: device_sort (per-row sorts of the long-row levels on the device, default 1);
 *   fusions: fuse_zr ((z, r) of PCG from the last level-0 Jacobi sweep), fuse_presmooth (first Jacobi sweep written with
 *     its right-hand side) -- both default 1, results identical (fuse_presmooth: bit for bit; fuse_zr: to rounding);
 *   sequential sweeps (a parallel pass + a sparse triangular solve, csrc/seq_split.hip.h): seq_flow (the triangular solve as a
 *     dataflow over strips of the sweep sequence, default 1; 0 = one launch per dependency class), seq_strip_kb (slot bytes per
 *     strip when a schedule is built, default 512), seq_lanes (lanes per row, 0 = from the row lengths) -- same slots, same
 *     arithmetic, same bits; seq_spine (csrc/seq_sched.h: a row's last two operands in its last lane behind the cross-lane sum; 1 =
 *     on chain-bound eight-round schedules (default), 0 never, 2 wherever a row has two lanes -- part of a schedule's arithmetic: the
 *     modes agree to rounding, not bit for bit), seq_grid (workgroups of the dataflow solve at most; 0 = the schedule's own cap,
 *     < 0 = every resident one), seq_jobs (the schedules built side by side on host threads -- behind the host setup for large
 *     operators, else at the first sweep; default 1);
 *     gs_multicolor = 1 selects the MULTICOLOUR Gauss-Seidel / SOR sweep -- NOT the reference's iteration (rows are
 *     relaxed colour by colour instead of in index order; faster, converges alike, other iteration counts). Default 0:
 *     the reference's sequential sweep, reproduced exactly;
 *   multi-GPU: halo_overlap (exchange beside the interior rows, default 1), split_rows (test mode: every operator in
 *     three row windows), seq_partition (set before the upload, or FASP_HIP_SEQ_PARTITION=1: hierarchies with Gauss-Seidel / SOR
 *     smoothers are row-partitioned too and the ranks sweep by turns; default 0: such hierarchies keep every level whole),
 *     local_square (a rank's rows of a partitioned level coded with row-relative column offsets like the square operator, default 1;
 *     read at upload).
 * Unknown keys return ERROR_INPUT_PAR. */
int fasp_hip_tune(const char* key, int value);

/* Counters of the communicator since the last reset: out[0] halo exchanges, [1] all-reduces, [2] all-gathers, [3] doubles sent in
 * exchanges, [4] doubles contributed to all-gathers, [5..7] seconds spent in the three -- filled only in the diagnostic mode
 * fasp_hip_comm_timing(1), which drains the stream around every call (a breakdown of a serialised solve, not a benchmark). */
int fasp_hip_comm_stats(double* out8, int reset);
int fasp_hip_comm_timing(int on);

/* Row partition of a hierarchy over `nranks` GPUs as rank `rank` sees it (host only; levels
 * with fewer than min_rows rows are replicated).  fasp_hip_amg_upload() builds the same
 * plan from the communicator; these entry points expose it to tests.
 * info = {replicated, nglobal, row0, nloc, nghost, nsend, first_replicated_level, nranks};
 * get_matrix: the rank's local rows of A (0) / P (1) / R (2) in local column numbering;
 * get_list:   0 ghost global ids, 1 recv offsets, 2 send offsets, 3 send local ids,
 *             4 ownership offsets of the level. */
int fasp_hip_dist_plan(fasp_hip_amg* h, int rank, int nranks, int min_rows);
int fasp_hip_dist_level_info(const fasp_hip_amg* h, int level, int* info);
int fasp_hip_dist_get_matrix(const fasp_hip_amg* h, int level, int which, dCSRmat* view);
int fasp_hip_dist_get_list(const fasp_hip_amg* h, int level, int which, ivector* view);
/* k_csr_estream's decomposition tables (csrc/kernels3.hip.h) for a matrix with these row pointers, built and walked on the HOST the way the
 * kernel walks them: 0 when every entry is covered once and every row is finished exactly once, else the negative number of the check
 * that failed.  info (may be NULL) = {wave ranges, chunks, rows cut by a wave boundary}.  No GPU needed. */
int  fasp_hip_estream_selftest(const int* ia, int nrow, int nnz, int per_wave, int wmax, int* info);
/* one-rank exercise of every RCCL call the transport makes (0 = all results correct) */
int  fasp_hip_comm_selftest(void);

#ifdef __cplusplus
}
#endif
#endif /* FASP_HIP_DEV_H */
