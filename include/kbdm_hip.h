/*
 * kbdm_hip.h - C ABI of libkbdm_hip.so: the MI355X (gfx950) KBDM ensemble solver.
 *
 * The reference (danilomendesdias/llckbdm) is pure Python and has no FFI layer; its hot
 * path is the loop llckbdm/sampling.py:52-70 calling llckbdm/kbdm.py:19-92 once per
 * ensemble member.  This library replaces exactly that loop with one batched HIP
 * pipeline.  Everything here is plain C: pointers, sizes, error codes.  No exceptions
 * cross the boundary; every function returns 0 on success or a negative KBDM_E_* code
 * (kbdm_last_error() gives the text).  The Python side (llckbdm_amd/_lib.py) binds these
 * symbols with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Data conventions
 *   complex128  = two consecutive doubles (re, im), as numpy complex128.
 *   signals     : S x N complex128, row-major (one time signal per row).
 *   items       : ensemble members; item i solves signal sig_idx[i] with Krylov size m[i]
 *                 and retained rank l[i] (1 <= l[i] <= m[i], 2*m[i] + p - 1 <= N).
 *   lines       : concatenated per item, row-major (l_i x 4) doubles in the reference's
 *                 column order (amplitude, T2, frequency, phase)   [kbdm.py:88-90]
 *   sv          : concatenated per item, m_i doubles, descending   [KbdmInfo.singular_values]
 *   mu          : concatenated per item, l_i complex128 eigenvalues [kbdm.py:192]
 *   keep        : concatenated per item, l_i bytes: 1 iff A > 1e-6 and T2 > 0
 *                 (filter_samples, sampling.py:75-97)
 *   status      : per item bit mask, 0 = ok (KBDM_STAT_*).
 *   Row order inside one item's line list is the order in which this library's
 *   eigen-solver delivers eigenvalues; like LAPACK zgeev's order in the reference it
 *   carries no meaning.
 */
#ifndef KBDM_HIP_H
#define KBDM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KBDM_ABI_VERSION 3

#define KBDM_OK 0
#define KBDM_E_INVALID (-1)   /* bad argument (sizes, null pointers, m/l/p constraint) */
#define KBDM_E_HIP (-2)       /* a HIP runtime call failed                            */
#define KBDM_E_NOMEM (-3)     /* workspace does not fit                               */
#define KBDM_E_NODEVICE (-4)  /* no gfx950 device visible                             */

#define KBDM_STAT_SVD_NOCONV 1
#define KBDM_STAT_EIG_NOCONV 2
#define KBDM_STAT_INVIT_WEAK 4

/* number of per-stage timers reported by kbdm_plan_stage_ms */
#define KBDM_NSTAGES 16

typedef struct kbdm_ctx kbdm_ctx;
typedef struct kbdm_plan kbdm_plan;

int kbdm_abi_version(void);
int kbdm_device_count(void);
const char* kbdm_last_error(void);

/* One context per process and GPU: device selection, stream, module attributes. */
int kbdm_ctx_create(int device, kbdm_ctx** out);
/* the same with the number of lanes (concurrent sub-batches by member size) given instead of read from KBDM_LANES; lanes <= 0:
 * as kbdm_ctx_create.  Results do not depend on it. */
int kbdm_ctx_create_lanes(int device, int lanes, kbdm_ctx** out);
int kbdm_ctx_destroy(kbdm_ctx* ctx);
/* Cooperative panels: the two panel kernels of the blocked reductions (the serial chains of a member: bidiagonalisation,
 * Hessenberg reduction) run with teams of up to T workgroups per member of lane 0 (all_lanes != 0: of every lane) whenever
 * count x T workgroups fit `budget`.  The workgroups of a team wait for each other, so all teams of a launch must be
 * resident: keep the SUM of the budgets of the contexts that may run at the same time on one GPU at or below its number of
 * compute units (256).  lane0_frac > 0 also sets lane 0's share of a batch's cost (teams shorten lane 0's chain, so it can
 * take more).  Results do not depend on any of this (a member gets the same bits from a team of any size).  Default: T = 1. */
int kbdm_ctx_set_panel_teams(kbdm_ctx* ctx, int T, int budget, int all_lanes, double lane0_frac);

/* A plan fixes the batch geometry (replaces the arguments of sample_kbdm,
 * sampling.py:8: data/m_range/p/l/q, for many signals at once) and owns the device
 * workspace.  Items may have different m and l. */
int kbdm_plan_create(kbdm_ctx* ctx, int S, int N, int B, const int32_t* sig_idx, const int32_t* m,
                     const int32_t* l, int p, double q, double dwell, kbdm_plan** out);
int kbdm_plan_destroy(kbdm_plan* plan);
int64_t kbdm_plan_total_lines(const kbdm_plan* plan); /* sum of l_i */
int64_t kbdm_plan_total_sv(const kbdm_plan* plan);    /* sum of m_i */
/* per-item offsets (B+1 entries each) into lines/keep/mu (units: lines) and sv (units: values) */
int kbdm_plan_offsets(const kbdm_plan* plan, int64_t* line_off, int64_t* sv_off);

int kbdm_plan_upload(kbdm_plan* plan, const double* signals_host); /* S*N complex128, H2D   */
int kbdm_plan_execute(kbdm_plan* plan);                            /* enqueue every kernel  */
int kbdm_plan_sync(kbdm_plan* plan);                               /* wait for the stream   */
int kbdm_plan_download(kbdm_plan* plan, double* lines, double* sv, double* mu, uint8_t* keep,
                       int32_t* status);                           /* D2H (null = skip)     */
/* Host -> host without a blocking copy (the unit the metric counts: llckbdm/sampling.py:52-70 takes host data and
 * returns host line lists): kbdm_plan_submit copies the signals (S*N complex128; null = keep the uploaded ones) into
 * the plan's pinned staging buffer and enqueues H2D, every kernel and the D2H of all outputs; it returns at once, so
 * that several plans (on different contexts) can be in flight.  kbdm_plan_collect waits for the plan and copies the
 * results out (null = skip).  One submit per collect. */
int kbdm_plan_submit(kbdm_plan* plan, const double* signals_host);
int kbdm_plan_collect(kbdm_plan* plan, double* lines, double* sv, double* mu, uint8_t* keep, int32_t* status);

/* Conservative execution mode of a plan, used by the host for the ONE retry of members whose status word reports
 * non-convergence before it raises (scipy.linalg.svd / eig raise LinAlgError there: llckbdm/kbdm.py:166,192):
 *   KBDM_MODE_SOLO_QR      every member's QR iteration runs in one workgroup (no chase + helper teams: the one place
 *                          where workgroups of a launch wait for each other) */
#define KBDM_MODE_SOLO_QR 2
/*   KBDM_MODE_KERNEL_TIMERS a measuring mode (bench.py's roofline): pairs of HIP events around every launch of the kernel
 *                          classes below, on the stream the kernel runs on; read with kbdm_plan_kernel_ms */
#define KBDM_MODE_KERNEL_TIMERS 4
int kbdm_plan_set_mode(kbdm_plan* plan, int mode);
/* Kernel classes of the per-kernel timers; lane 0's launches (the largest members) are the ones timed. */
#define KBDM_K_HANKEL 0
#define KBDM_K_BIDIAG_PANEL 1
#define KBDM_K_TRAIL_UPDATE 2
#define KBDM_K_HESS_PANEL 3
#define KBDM_K_HESS_Z 4
#define KBDM_K_HESS_UPDATE 5
#define KBDM_K_AB_ITER 6
#define KBDM_K_WY_APPLY 7
#define KBDM_NKCLASSES 8
const char* kbdm_kernel_class_name(int klass);
/* Sum of the HIP-event durations of the launches of `klass` in the last KBDM_MODE_KERNEL_TIMERS run and their number
 * (k_ab_iter: one bracket around its back-to-back launches); waits for the plan. */
int kbdm_plan_kernel_ms(kbdm_plan* plan, int klass, float* total_ms, int32_t* launches);
/* Test hook of the status contract (tests/test_gpu_api_r3.py), process-wide: kbdm_plan_download / kbdm_plan_collect OR
 * `always` into every member's status word, and `once` into the words of the NEXT collected run only.  (0, 0) disarms it.
 * Nothing on the result path reads the environment. */
int kbdm_debug_force_status(int always, int once);
/* device bytes a plan of this geometry will allocate (before chunking) / a plan holds */
int64_t kbdm_workspace_estimate(int B, const int32_t* m, const int32_t* l);
int64_t kbdm_plan_workspace_bytes(const kbdm_plan* plan);

/* device-resident outputs (for a collective over xGMI without a host bounce) */
void* kbdm_plan_lines_device(kbdm_plan* plan);
void* kbdm_plan_sv_device(kbdm_plan* plan);
/* device-to-device copy of the packed lines (total_lines x 4 doubles) into a caller-owned
 * device buffer (e.g. a torch tensor that RCCL gathers); synchronous on the plan's stream */
int kbdm_plan_copy_lines_device(kbdm_plan* plan, void* dst_device, int64_t dst_bytes);
/* elapsed ms of each pipeline stage in the last execute (HIP events on the plan's stream);
 * also returns the kernel names through kbdm_stage_name */
/* Block the calling thread until the critical lane of the run in flight has finished `stage` (index as in
 * kbdm_stage_name; first group of the plan).  Lets a host that keeps two plans in flight start the second one
 * half a cycle after the first (bench.py), so that the throughput-bound stages of one ensemble meet the QR
 * iteration of the other.  No reference counterpart. */
int kbdm_plan_wait_stage(kbdm_plan* plan, int stage);
int kbdm_plan_stage_ms(kbdm_plan* plan, float* ms, int n);
const char* kbdm_stage_name(int stage);
/* A plan runs its members in concurrent "lanes" (sub-batches by size, largest members in lane 0,
 * each an in-order pipeline on its own HIP stream).  The stage timers above are lane 0's: the
 * critical path.  Returns the number of members (the largest ones) that lane 0 holds. */
int kbdm_plan_lane0_members(const kbdm_plan* plan);
/* Members of the last run whose eigenvalues came from the QR iteration because the divide-and-conquer Ehrlich-Aberth
 * path declined them (negligible subdiagonal, a root that did not settle, power-sum check); waits for the plan. */
int kbdm_plan_eig_fallbacks(kbdm_plan* plan);
/* the same count for the last kbdm_eig_batch call on this context (the stage entry point owns its plan); -1: no context */
int kbdm_ctx_last_eig_fallbacks(kbdm_ctx* ctx);
/* Diagnostics of that path: out[step * 24 + iteration] = root tiles (64 roots each) that were still iterating in that
 * launch, summed over the runs since the last call (n <= 16 * 24 entries); waits for the plan. */
int kbdm_plan_ab_stats(kbdm_plan* plan, int32_t* out, int n);

/* ---- multi-GPU: the sharded ensemble (reference loop sampling.py:52-70, members dealt over the ranks) ends in ONE
 * variable-length gather of every rank's packed results over xGMI.  The library binds librccl.so itself (dlopen); the
 * context owns the communicator.  Launcher protocol: rank 0 calls kbdm_comm_unique_id and hands the 128 bytes to the
 * other ranks by whatever means the launcher has (before any collective); every rank then calls kbdm_comm_init.
 * A rank's packed block is  [lines: L x 4 f64][sv: SV f64][status: B i32][keep: L u8][pad to 16 B][trailer: 16 B]  for its
 * L lines, SV singular values and B members; kbdm_packed_bytes gives its size, so every rank can size every other rank's
 * block from the (deterministic) shard table alone: no size exchange, no host bounce.  The trailer is
 * {u32 magic "KBDM", u32 sender rank, u64 sequence number of the gather}: the receiver checks on the device that every
 * block it got belongs to THIS gather (ABI 3), so ranks that issue their gathers in different orders get an error from
 * kbdm_gather_wait instead of each other's steps. */
#define KBDM_UNIQUE_ID_BYTES 128
int kbdm_comm_unique_id(unsigned char* id_out);
int kbdm_comm_init(kbdm_ctx* ctx, int world, int rank, const unsigned char* id);
int kbdm_comm_destroy(kbdm_ctx* ctx);
/* A process that keeps several contexts on one GPU (ensembles in flight) needs ONE communicator: `ctx` borrows
 * `owner`'s (same device).  The gathers of all these contexts must then be issued in the same order on every rank, and
 * `owner` must outlive the borrowers' last gather; kbdm_comm_destroy on a borrower only detaches. */
int kbdm_comm_attach(kbdm_ctx* ctx, kbdm_ctx* owner);
int64_t kbdm_packed_bytes(int64_t lines, int64_t sv, int64_t members);
/* Packs this plan's results on the device and gathers the blocks of all `world` ranks (bytes[r] = block size of rank
 * r, blocks concatenated in rank order) with one grouped ncclSend/ncclRecv on the plan's stream: to every rank
 * (root < 0) or to `root` only.  host_out (sum of bytes, receiving ranks; may be null) gets a copy; the device copy
 * stays available through kbdm_gathered_device (after kbdm_gather_wait).  world == 1 needs no communicator. */
int kbdm_plan_gather(kbdm_plan* plan, int world, int rank, const int64_t* bytes, int root, void* host_out);
/* The transfer runs on a communication stream of the context behind an event of the plan's stream: with host_out = null
 * kbdm_plan_gather only ENQUEUES it and returns (the plan's context can take its next run at once; its pack buffer is
 * reused only after the transfer); kbdm_gather_wait blocks until the context's last gather has landed - for at most
 * KBDM_GATHER_TIMEOUT_S seconds (environment, default 300): a peer that died leaves the transfer hanging, the wait then
 * returns KBDM_E_HIP so that the process can exit and its launcher tear the job down - and reports a trailer mismatch as
 * KBDM_E_HIP too.  With host_out the call itself waits. */
int kbdm_gather_wait(kbdm_ctx* ctx);
void* kbdm_gathered_device(kbdm_ctx* ctx);

/* One-shot: plan + upload + execute + download. */
int kbdm_solve_batch(kbdm_ctx* ctx, const double* signals, int S, int N, int B,
                     const int32_t* sig_idx, const int32_t* m, const int32_t* l, int p, double q,
                     double dwell, double* lines, double* sv, double* mu, uint8_t* keep,
                     int32_t* status);

/* ---- stage entry points (parity tests of SURVEY.md 8a rows a3, a4, a7) ---------------- */
/* Hankel assembly, reference kbdm.py:95-130: for item i writes U0, U^{p-1}, U^p, each
 * m_i x m_i ROW-major (numpy C order), items concatenated.  Any output may be null. */
int kbdm_hankel_batch(kbdm_ctx* ctx, const double* signals, int S, int N, int B,
                      const int32_t* sig_idx, const int32_t* m, int p, double* U0, double* Up1,
                      double* Up);
/* Full SVD, reference kbdm.py:166: A_i (m_i x m_i, row-major, concatenated) ->
 * L_i (row-major), s_i, R_i (row-major) with A = L diag(s) R^H. */
int kbdm_svd_batch(kbdm_ctx* ctx, const double* A, int B, const int32_t* m, double* L, double* s,
                   double* R, int32_t* status);
/* Eigen-decomposition, reference kbdm.py:192: W_i (n_i x n_i row-major) -> mu_i, P_i
 * (row-major, column k = right eigenvector of mu_k, arbitrary scale). */
int kbdm_eig_batch(kbdm_ctx* ctx, const double* W, int B, const int32_t* n, double* mu, double* P,
                   int32_t* status);

/* ---- rows next to the hot path (consumers of the line lists) -------------------------------- */

/* Frequency-domain RMSE of `ncand` candidate line lists against one signal: replaces
 * llckbdm/metrics.py:7-17 (calculate_freq_domain_rmse) as called per candidate by
 * llckbdm/min_rmse_kbdm.py:33-38.  data: N complex128; lines: packed rows (A, T2, F, PH) of all
 * candidates; cand_off[ncand+1]: row ranges (an empty range gives +inf, as min_rmse_kbdm.py:36-37);
 * rmse_out[ncand]. */
int kbdm_rmse_batch(kbdm_ctx* ctx, const double* data, int N, double dwell, const double* lines,
                    const int64_t* cand_off, int ncand, double* rmse_out);

/* Silhouette coefficient of every sample, Euclidean metric: replaces
 * sklearn.metrics.silhouette_samples as called by llckbdm/llckbdm.py:291.  X: n x dim float64
 * row-major (dim <= 8); labels[n] (any integers; -1 is a class of its own, as in the reference);
 * out[n].  Needs at least 2 classes and at most n - 1 (sklearn's precondition). */
int kbdm_silhouette_samples(kbdm_ctx* ctx, const double* X, int n, int dim, const int32_t* labels, double* out);

/* The same for nfits labelings of the same samples in one call (the sweep scores every fit: llckbdm.py:291 inside the loop
 * at :104-110): labels[nfits * n], out[nfits * n]; the same bits as one kbdm_silhouette_samples call per labeling.  A
 * labeling outside sklearn's precondition (fewer than 2 or more than n - 1 label values) gets valid_out[f] = 0 and zeros
 * instead of an error; valid_out may be null. */
int kbdm_silhouette_sweep(kbdm_ctx* ctx, const double* X, int n, int dim, const int32_t* labels, int nfits, double* out,
                          int32_t* valid_out);

/* HDBSCAN* for every value of `min_samples` of the clustering sweep at once: replaces the loop of
 * `hdbscan.HDBSCAN(min_samples=k).fit(X).labels_` at llckbdm/llckbdm.py:104-110,280-283 (Euclidean, alpha 1,
 * excess of mass, no single-cluster result; semantics of scikit-learn's HDBSCAN: the point itself counts towards
 * min_samples).  X: n x dim float64 row-major (dim <= 8); min_samples[nfits] (1 <= k <= n);
 * labels_out[nfits * n] (-1 = noise), nclusters_out[nfits] (may be null).  The O(n^2) parts (k-nearest-neighbour
 * distances once for all fits, one Prim MST per fit, all fits concurrently) run on the GPU, the trees on the host.
 * No limit on min_samples other than n. */
int kbdm_hdbscan_sweep(kbdm_ctx* ctx, const double* X, int n, int dim, const int32_t* min_samples, int nfits,
                       int min_cluster_size, int32_t* labels_out, int32_t* nclusters_out);

/* Core distances (the distance to the k-th nearest sample, the sample itself counted: what the sweep above feeds its
 * mutual-reachability weights with; hdbscan's `core_distances_`) for every k of min_samples[nfits]: out[nfits * n].
 * Same kernel, same limits as the sweep: none on k (lists that do not fit the LDS are produced in passes). */
int kbdm_core_distances(kbdm_ctx* ctx, const double* X, int n, int dim, const int32_t* min_samples, int nfits, double* out);

/* The host half on its own (no GPU): labels from the n-1 edges (a[i], b[i], w[i]) of a minimum spanning tree of the
 * mutual-reachability graph.  Returns the number of clusters. */
int kbdm_hdbscan_labels_from_mst(int n, const int32_t* a, const int32_t* b, const double* w, int min_cluster_size,
                                 int32_t* labels_out);

#ifdef __cplusplus
}
#endif
#endif /* KBDM_HIP_H */
