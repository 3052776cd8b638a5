/*
 * safe_hip.h -- C ABI of libsafe_hip.so: the MI355X (gfx950) implementation of the
 * SAFE hot path (neighborhood definition + enrichment p-values).
 *
 * The reference (baryshnikova-lab/safepy) is pure Python and has no FFI; this header
 * is the boundary a maintainer binds with ctypes.  Every entry
 * point cites the reference code it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every function returns 0 on success, a negative SAFE_E_* code on failure;
 *     safe_last_error() returns a thread-local message for the last failure.
 *   - "host" pointers are ordinary process memory owned by the caller; "dev" pointers
 *     are HIP device memory (from safe_dev_alloc or any other HIP allocator, e.g. a
 *     torch tensor's data_ptr()).  Outputs are always pre-allocated by the caller.
 *   - the library owns device memory only behind its opaque handles.
 *   - work is enqueued on the context's stream (safe_ctx_set_stream; default: a
 *     stream the context owns).  Functions with host outputs synchronise before
 *     returning; functions with only dev outputs are asynchronous.
 *   - one context per device; a context is not thread-safe.
 *   - there is NO CPU fallback: without a HIP device every compute call fails.
 */
#ifndef SAFE_HIP_H
#define SAFE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAFE_HIP_ABI_VERSION 3

#define SAFE_OK 0
#define SAFE_E_INVALID (-1)   /* bad argument */
#define SAFE_E_HIP (-2)       /* HIP runtime error (message has the hipError string) */
#define SAFE_E_NOMEM (-3)
#define SAFE_E_UNSUPPORTED (-4)
#define SAFE_E_VALUE (-5)     /* input data violates a contract (e.g. non-0/1 membership) */

#define SAFE_DTYPE_F32 0
#define SAFE_DTYPE_F64 1
#define SAFE_DTYPE_U8 2       /* safe_attr_create_host only: 0/1 (or small integer) values as bytes, no missing values */

#define SAFE_SCORE_SUM 0      /* neighborhood_score_type == 'sum'     */
#define SAFE_SCORE_ZSCORE 1   /* neighborhood_score_type == 'z-score' */

#define SAFE_SIGN_HIGHEST 0   /* attribute_sign == 'highest' */
#define SAFE_SIGN_LOWEST 1    /* attribute_sign == 'lowest'  */
#define SAFE_SIGN_BOTH 2      /* attribute_sign == 'both'    */

typedef struct safe_ctx safe_ctx;
typedef struct safe_nbr safe_nbr;       /* neighborhood membership, device resident     */
typedef struct safe_attr safe_attr;     /* node x attribute matrix, device resident     */
typedef struct safe_perms safe_perms;   /* composed row-permutation tables, device res. */
typedef struct safe_comm safe_comm;     /* RCCL communicator of the attribute-sharded path */
typedef struct PermRing safe_ring;      /* node-shared permutation stream (shared memory)    */

/* ------------------------------------------------------------------ context ---- */
int safe_abi_version(void);
const char *safe_last_error(void);
int safe_device_count(int *count);
/* PCI address of a device ("0000:c1:00.0"): lets the host side keep a rank's threads on the NUMA node of its GPU
 * (safepy_amd.backend.pin_threads_to_device_numa; the reference has no counterpart -- its multiprocessing
 * workers, safe.py:503-524, run wherever the OS puts them). */
int safe_device_pci_bus_id(int device, char *buf, size_t buf_len);
int safe_ctx_create(int device, safe_ctx **out);
int safe_ctx_destroy(safe_ctx *ctx);
/* Host waits of every context in this process: 0 (default) = hipStreamSynchronize, the runtime spins (lowest latency, one
 * busy core per waiting thread); 1 = sleep on interrupt-backed events (several ranks sharing few host cores: the
 * reference's worker pool, safepy/safe.py:503-524, simply oversubscribes).  Set before the first context is created;
 * SAFE_HIP_BLOCKING_SYNC=1 in the environment has the same effect. */
int safe_set_blocking_sync(int on);
/* Use an existing hipStream_t (passed as void*) for all subsequent work; NULL restores
 * the context's own stream. */
int safe_ctx_set_stream(safe_ctx *ctx, void *hip_stream);
int safe_ctx_sync(safe_ctx *ctx);
int safe_ctx_info(safe_ctx *ctx, int *num_cu, int64_t *hbm_bytes, char *arch, size_t arch_len);
/* How this library was built: the compiler version and the verdict of the build's disassembly check of the bit-sliced
 * permutation kernel's hidden registers (the kernel behind run_permutations, safepy/safe_extras.py:36-70, keeps two id quads in
 * registers the compiler does not allocate; a library built where the check could not run uses the form without them). */
int safe_build_info(char *out, size_t out_len);
int safe_dev_alloc(safe_ctx *ctx, size_t bytes, void **out_dev);
int safe_dev_free(safe_ctx *ctx, void *dev);
int safe_dev_memset(safe_ctx *ctx, void *dev, int value, size_t bytes);
int safe_memcpy_h2d(safe_ctx *ctx, void *dev, const void *host, size_t bytes);   /* synchronous */
int safe_memcpy_d2h(safe_ctx *ctx, void *host, const void *dev, size_t bytes);   /* synchronous */
/* The same copy for a destination whose pages are RESIDENT (an array that has been written before, e.g. the recycled host
 * array of a result matrix -- self.nes and its siblings, safepy/safe.py:530-554): one plain copy at the link's rate (56 GB/s
 * measured), no helper threads.  (safe_memcpy_d2h spreads the page faults of a FRESH destination over copy threads.) */
int safe_memcpy_d2h_resident(safe_ctx *ctx, void *host, const void *dev, size_t bytes);
/* Event pair on the context stream, for timing a region that runs on that stream. */
int safe_timer_start(safe_ctx *ctx);
int safe_timer_stop_ms(safe_ctx *ctx, double *elapsed_ms);   /* synchronises */

/* ------------------------------------------------------------ neighborhoods ---- */
/* Replaces SAFE.define_neighborhoods, 'euclidean' branch (safepy/safe.py:389-399):
 * A[i,j] = (sqrt(dx*dx + dy*dy) < nr), each op rounded to f64 (scipy pdist), diagonal
 * kept.  xy_host is [n,2] row-major; nr = radius * (max x - min x) is computed by the
 * caller exactly as safe.py:390-391 does. */
int safe_nbr_euclidean(safe_ctx *ctx, const double *xy_host, int64_t n, double nr, safe_nbr **out);

/* Replaces the two shortest-path branches (safepy/safe.py:401-417; networkx
 * all_pairs_dijkstra_path_length with cutoff): A[s,t] = 1 iff the shortest-path length
 * from s to t is <= cutoff.  Undirected edges (edge_u[e], edge_v[e]) with weight
 * edge_w[e] ('length' for shortpath_weighted_layout) or unit weight when edge_w is NULL
 * ('shortpath').  keep_distances != 0 also keeps the dense f64 distance matrix
 * (inf where unreached) for safe_nbr_distances (self.node_distances, safe.py:417). */
int safe_nbr_shortpath(safe_ctx *ctx, int64_t n, int64_t n_edges, const int32_t *edge_u,
                       const int32_t *edge_v, const double *edge_w, double cutoff,
                       int keep_distances, safe_nbr **out);

/* Membership supplied by the caller as the reference's own layout: self.neighborhoods,
 * int64 [n,n] row-major with entries in {0,1} (safe.py:387,430).  Any other value is
 * SAFE_E_VALUE. */
int safe_nbr_from_dense_i64(safe_ctx *ctx, const int64_t *a_host, int64_t n, safe_nbr **out);
/* Optional hint: the 2-D layout the membership was derived from ([n,2] row-major, the x/y node
 * attributes read at safepy/safe.py:393-396).  Only used to renumber nodes along a
 * space-filling curve for the block-sparse matrix-core kernel; results never depend on it.
 * safe_nbr_euclidean records its own xy; without a layout the order is Cuthill-McKee. */
int safe_nbr_set_layout(safe_nbr *nbr, const double *xy_host);
/* Number of stored 256-row x 32-column blocks of the block-sparse membership used by the
 * matrix-core permutation kernel (0 until that kernel has run once on the handle): the
 * algorithmic GEMM size for roofline reporting. */
int safe_nbr_block_count(const safe_nbr *nbr, int64_t *blocks);
/* Of those blocks' 32-row x 32-column pieces (8 per block), the ones that hold at least one member: the kernel issues
 * matrix-core instructions for these only, so the EXECUTED operation count of a call is pieces x 32 x 32 x columns x slices
 * x (permutations + 1) x 2 (roofline reporting; 0 until the kernel has run once on the handle). */
int safe_nbr_piece_count(const safe_nbr *nbr, int64_t *pieces);
int safe_nbr_destroy(safe_nbr *nbr);
int safe_nbr_info(const safe_nbr *nbr, int64_t *n, int64_t *nnz, int64_t *max_row_count);
/* self.neighborhoods in the reference layout: int64 [n,n] row-major (safe.py:430). */
int safe_nbr_to_dense_i64(safe_nbr *nbr, int64_t *out_host);
int safe_nbr_to_dense_i64_dev(safe_nbr *nbr, int64_t *out_dev);
/* np.sum(neighborhoods, axis=1) (safe.py:423). */
int safe_nbr_row_counts(safe_nbr *nbr, int64_t *out_host);
/* CSR view (row_ptr [n+1], col [nnz], ascending columns). */
int safe_nbr_csr(safe_nbr *nbr, int32_t *row_ptr_host, int32_t *col_host);
/* Dense f64 [n,n] distances of a shortest-path handle built with keep_distances
 * (inf where unreached).  SAFE_E_INVALID if distances were not kept. */
int safe_nbr_distances(safe_nbr *nbr, double *out_host);

/* The fused all-pairs kernel on its own (compute_node_distances for 'euclidean', and
 * the K1 roofline bench): xy_dev [n,2]; mask_out_dev int64 [n,n] and/or dist_out_dev
 * f64 [n,n]; either may be NULL.  Same arithmetic as safe_nbr_euclidean
 * (safepy/safe.py:397-399). */
int safe_euclidean_dense_dev(safe_ctx *ctx, const double *xy_dev, int64_t n, double nr,
                             int64_t *mask_out_dev, double *dist_out_dev);
/* Replaces calculate_edge_lengths (safepy/safe_io.py:311-333): length[e] =
 * sqrt(dx*dx + dy*dy) of the edge's end points, without the N x N detour. */
int safe_edge_lengths(safe_ctx *ctx, const double *xy_host, int64_t n, int64_t n_edges,
                      const int32_t *edge_u, const int32_t *edge_v, double *out_host);

/* --------------------------------------------------------------- attributes ---- */
/* self.node2attribute (safepy/safe_io.py:410): [n,m], f32 or f64, NaN = missing, with
 * element strides (row_stride, col_stride): (m,1) for C order, (1,n) for Fortran order.
 * The _host form uploads a copy; the _dev form borrows the caller's device buffer, which
 * must outlive the handle.  Additive (the reference's loader hands over f32, safe_io.py:361, and the
 * upload of a 0/1 matrix is most of a hypergeometric call, safe.py:556-608): the _host form also
 * takes SAFE_DTYPE_U8 -- one byte per value over the link, widened to f32 on the device. */
int safe_attr_create_host(safe_ctx *ctx, const void *b_host, int dtype, int64_t n, int64_t m,
                          int64_t row_stride, int64_t col_stride, safe_attr **out);
int safe_attr_create_dev(safe_ctx *ctx, const void *b_dev, int dtype, int64_t n, int64_t m,
                         int64_t row_stride, int64_t col_stride, safe_attr **out);
int safe_attr_destroy(safe_attr *attr);
/* read_attributes on the device (safepy/safe_io.py:386-390 `node2attribute.reindex(index=
 * node_label_order, fill_value=fill_value)` + `.values` at :410): uploads the file's
 * [n_labels, m] table (f32 or f64, C or Fortran order by element strides) once, gathers its
 * rows into network node order -- row_map_host[i] = table row of node i, -1 = label not in
 * the file (filled with fill_value), -2 = masked duplicate node (NaN, safe_io.py:394-404) --
 * and returns the aligned matrix as a device-resident handle, laid out C (out_order 0) or
 * Fortran (1).  If out_host is not NULL the aligned matrix (same dtype and order) is also
 * copied there: that is self.node2attribute. */
int safe_attr_reindex(safe_ctx *ctx, const void *table_host, int dtype, int64_t n_labels, int64_t m,
                      int64_t row_stride, int64_t col_stride, const int64_t *row_map_host, int64_t n,
                      double fill_value, int out_order, void *out_host, safe_attr **out);
/* The value census read_attributes logs (safepy/safe_io.py:426-429): #NaN, #zeros,
 * #positives, #negatives of the whole matrix in one pass. */
int safe_attr_value_counts(safe_attr *attr, int64_t *n_nan, int64_t *n_zero, int64_t *n_positive,
                           int64_t *n_negative);
/* background='network' (safepy/safe.py:449-451): NaN -> 0 in place on the device copy. */
int safe_attr_nan_to_zero(safe_attr *attr);
/* Copies the device matrix (its own dtype and order) to the host. */
int safe_attr_download(safe_attr *attr, void *out_host);
/* Whole-matrix facts compute_pvalues needs before dispatch (safepy/safe.py:453-463):
 *   n_other   = #(non-NaN values not in {0,1})          -> 'auto' rule (safe.py:461)
 *   max_nan_col = max over columns of the NaN count      -> >50% warning (safe.py:454-459)
 *   n_rows_with_value = #rows with >= 1 non-NaN value    -> hypergeometric N (safe.py:574-578)
 *   n_non_integer = #(non-NaN values that are not integers) */
int safe_attr_stats(safe_attr *attr, int64_t *n_other, int64_t *max_nan_col,
                    int64_t *n_rows_with_value, int64_t *n_non_integer);
/* row_has_value[i] = 1 iff row i has >= 1 non-NaN value: indx_vals of
 * safepy/safe_extras.py:51 and nodes_not_nan of safepy/safe.py:574. */
int safe_attr_row_flags(safe_attr *attr, uint8_t *out_host);
/* Override the row flags (attribute-sharded multi-GPU runs must use the flags of the
 * FULL matrix, not of the local column shard). */
int safe_attr_set_row_flags(safe_attr *attr, const uint8_t *flags_host);

/* ------------------------------------------------------------ permutations ---- */
/* The row-permutation stream of run_permutations (safepy/safe_extras.py:46-58):
 * np.random.seed(seed) (legacy MT19937 init_genrand) followed by num_permutations calls
 * of np.random.permutation(indx_vals) (masked-rejection Fisher-Yates), applied
 * cumulatively; movable_host[i] != 0 marks indx_vals.  has_seed == 0 seeds from OS
 * entropy (random_seed=None).  The result is the composed table cur[p][i] with
 * permuted_matrix_p = B[cur[p]], resident on the device.  Split of the work: one host thread runs
 * MT19937 and the rejection chain (how many words a shuffle consumes depends on its rejections:
 * the one sequential part) and ships the accepted swap targets; the swaps are replayed and the
 * tables composed on the device, chunk by chunk, while the enrichment kernels of earlier chunks run. */
int safe_perms_create(safe_ctx *ctx, int64_t n, const uint8_t *movable_host,
                      int64_t num_permutations, int has_seed, uint32_t seed, safe_perms **out);
int safe_perms_destroy(safe_perms *perms);
/* Copy table rows [p0,p1) to host as int32 [p1-p0, n] (tests). */
int safe_perms_read(safe_perms *perms, int64_t p0, int64_t p1, int32_t *out_host);
/* The same handle from a table the CALLER supplies instead of the legacy stream -- the `perm` of
 * safepy/safe_extras.py:58 produced elsewhere (an external RNG, a recorded run): perm_idx_host is int32
 * [num_permutations, n] row-major, COMPOSED like the tables above: row p is the index vector with
 * permuted_matrix_p = B[perm_idx[p]] (for the reference's cumulative in-place shuffle that is
 * cur_p = cur_{p-1}[...] of SURVEY A.3, not the single draw).  Every row must be a permutation of
 * 0..n-1 (SAFE_E_VALUE otherwise). */
int safe_perms_create_from_table(safe_ctx *ctx, int64_t n, int64_t num_permutations,
                                 const int32_t *perm_idx_host, safe_perms **out);
/* Permutations [p0, p1) of an existing handle as a handle of their own (device-to-device copy).  The permutation-axis
 * split of safepy/safe.py:489-519 (the reference hands each worker process num_permutations / processes permutations;
 * its workers reseed identically, so they repeat one another -- here every rank draws the ONE cumulative stream of
 * safe_extras.py:46-58 and tests its own range of it, and the ranks' counts add up to the single-process counts). */
int safe_perms_slice(safe_perms *perms, int64_t p0, int64_t p1, safe_perms **out);
/* UNSEEDED runs (random_seed=None -- the reference's default, safepy/safe.py:88; np.random.seed(None) at safe_extras.py:46 takes
 * OS entropy, so there is no stream to reproduce): the tables are generated ON THE DEVICE.  The reference's cumulative in-place
 * shuffles (safe_extras.py:58) make the composed tables i.i.d. uniform permutations of the movable rows, so each table row is
 * drawn directly: one lane per permutation runs Fisher-Yates in LDS with Philox4x32-10 words (key = `key`, counter =
 * permutation, word block) and unbiased bounded draws (Lemire's multiply-shift with rejection).  No host thread, nothing
 * sequential across permutations; every rank of a sharded run generates identical tables from the same key.  num movable rows
 * <= 65535.  The same arguments and the same handle as safe_perms_create otherwise; tables depend on (key, movable rows,
 * num_permutations index) only -- restated in oracle/safe_oracle.py for the tests. */
int safe_perms_create_device(safe_ctx *ctx, int64_t n, const uint8_t *movable_host, int64_t num_permutations,
                             uint64_t key, safe_perms **out);
/* One permutation stream per NODE (multi-GPU runs; the reference's workers each repeat the whole stream,
 * safepy/safe.py:489-519, 1339-1353).  The stream of safe_extras.py:46-58 is sequential, so a rank cannot draw "its part":
 * safe_ctx_share_stream attaches the context to a shared-memory ring named `name` (the same string on every rank of the
 * node, unique per job; capacity_bytes of chunk slots), local rank 0 being the node's producer.  safe_perms_create_shared
 * then behaves exactly like safe_perms_create -- same arguments, same tables -- but only the producer runs the draw
 * thread; it publishes each pipeline chunk's accepted swap targets (2 bytes each) and the other ranks block (futex, no CPU)
 * until a chunk is there, copy it and upload it; every rank replays the swaps and composes the tables on its own device.  The call is COLLECTIVE over the ranks of the node: same n, movable rows and permutation count,
 * same order of calls (checked: SAFE_E_VALUE otherwise; waits time out after SAFE_HIP_RING_TIMEOUT_S, default 120 s).  The
 * seed is the producer's.  When a chunk (128 x n x 2 bytes; 4 beyond 65535 rows) does not fit the ring twice, or the context shares no
 * stream, every rank silently draws for itself (decided from n and the capacity alone, hence identically everywhere). */
int safe_ctx_share_stream(safe_ctx *ctx, const char *name, int local_rank, int local_world, int64_t capacity_bytes);
int safe_ctx_unshare_stream(safe_ctx *ctx);
int safe_perms_create_shared(safe_ctx *ctx, int64_t n, const uint8_t *movable_host, int64_t num_permutations,
                             int has_seed, uint32_t seed, safe_perms **out);
/* Host-side timing of a handle's stream, ms: out5[0] = time the draw thread spent drawing, [1] = handle creation -> last
 * draw finished, [2] = creation -> last chunk's table kernels enqueued, [3] = time this rank blocked waiting for the
 * node's producer, [4] = role (0 own stream, 1 producer of a shared stream, 2 consumer, 3 generated on the device).  For
 * bench reporting. */
int safe_perms_timing(safe_perms *perms, double *out5);
/* The twin chain of a seeded handle (np.random.seed / np.random.permutation, safepy/safe_extras.py:46,58, drawn by two host
 * threads at once on a shared host: whichever finishes a pipeline chunk first publishes it): *twin_active = it runs for this
 * handle (opt-in: SAFE_HIP_DRAW_TWIN=1, single-process seeded calls with polling waits -- measured no better than one thread), *chunks = chunks published
 * so far, *chunks_won_by_twin = how many of them came from the second thread.  For bench reporting. */
int safe_perms_twin_stats(safe_perms *perms, int *twin_active, int64_t *chunks, int64_t *chunks_won_by_twin);
/* The ring by itself (host memory only, no device): what safe_perms_create_shared runs on, exported so that the
 * multi-process protocol can be tested on a host without GPUs.  local rank 0 creates, the others attach;
 * safe_ring_begin opens the next call on either side (producer: waits until every consumer has left the previous one;
 * consumer: waits for the producer's announcement and checks n / k / count / movable_hash against it);
 * safe_ring_publish / safe_ring_fetch move chunk `chunk` (in order); safe_ring_end leaves the call. */
int safe_ring_open(const char *name, int local_rank, int local_world, int64_t capacity_bytes, safe_ring **out);
int safe_ring_close(safe_ring *ring);
int safe_ring_begin(safe_ring *ring, int64_t n, int64_t k, int64_t count, uint64_t movable_hash, int64_t slot_bytes);
int safe_ring_publish(safe_ring *ring, int64_t chunk, const void *src_host, size_t bytes);
int safe_ring_fetch(safe_ring *ring, int64_t chunk, void *dst_host, size_t bytes);
int safe_ring_end(safe_ring *ring);
/* Host-only: the raw stream, for pinning against numpy (no device needed).  Writes
 * count permutations of values[0..n_items) back to back into out[count*n_items]. */
int safe_rng_permutations_host(uint32_t seed, const int64_t *values, int64_t n_items,
                               int64_t count, int64_t *out);

/* -------------------------------------------------------------- enrichment ---- */
/* Replaces compute_neighborhood_score (safepy/safe_extras.py:6-33).  Columns
 * [col0,col1) of the attribute handle; out_dev is f64 [n, col1-col0] row-major. */
int safe_score(safe_ctx *ctx, safe_nbr *nbr, safe_attr *attr, int score_type,
               int64_t col0, int64_t col1, double *out_dev);

/* Replaces run_permutations (safepy/safe_extras.py:36-70): counts_neg/counts_pos as
 * f64 [n, col1-col0] (number of permutations with permuted score <= / >= observed);
 * ns_dev (observed score) may be NULL. */
int safe_permtest_counts(safe_ctx *ctx, safe_nbr *nbr, safe_attr *attr, safe_perms *perms,
                         int score_type, int64_t col0, int64_t col1, double *ns_dev,
                         double *counts_neg_dev, double *counts_pos_dev);

/* Replaces compute_pvalues_by_randomization + the binarisation of compute_pvalues
 * (safepy/safe.py:496-554, 468-472; FDR branch excluded): all outputs f64
 * [n, col1-col0] row-major on the device, num_enriched_dev f64 [col1-col0].
 * nes_table_host: optional [P+1] table with nes_table[k] = -log10(k/P), k >= 1, and
 * nes_table[0] = -log10(1/P) evaluated by the caller's libm (NumPy) so NES matches the
 * caller's log10 bit for bit; NULL = use the library's log10. */
int safe_randomization(safe_ctx *ctx, safe_nbr *nbr, safe_attr *attr, safe_perms *perms,
                       int score_type, int sign_mode, double enrichment_threshold,
                       const double *nes_table_host, int64_t col0, int64_t col1,
                       double *ns_dev, double *pvalues_neg_dev, double *pvalues_pos_dev,
                       double *nes_dev, double *nes_binary_dev, double *num_enriched_dev);

/* Replaces compute_pvalues_by_hypergeom + the binarisation (safepy/safe.py:573-608,
 * 468-472; FDR branch excluded).  n_rows_with_value is the population size of
 * safe.py:578 (from safe_attr_stats of the FULL matrix). */
int safe_hypergeom(safe_ctx *ctx, safe_nbr *nbr, safe_attr *attr, double enrichment_threshold,
                   int64_t col0, int64_t col1, double *pvalues_pos_dev, double *nes_dev,
                   double *nes_binary_dev, double *num_enriched_dev);

/* multiple_testing=True (safepy/safe.py:536-542 and 599-605): Benjamini-Hochberg adjustment of
 * every row of the p-value matrices across its m attributes -- statsmodels'
 * fdrcorrection(row)[1], operation by operation -- IN PLACE on pvalues_neg_dev / pvalues_pos_dev
 * (f64 [n,m] row-major), then NES, nes_binary and the per-attribute counts recomputed from the
 * adjusted values (safe.py:546-554 / 608, 468-472).  num_permutations > 0: randomization form
 * (zero p-values become 1/num_permutations inside the logarithm, sign_mode combines the two
 * sides); num_permutations == 0: hypergeometric form (nes = -log10 pvalues_pos; pvalues_neg_dev
 * may be NULL).  A row needs all of its attributes: under attribute sharding this runs after the
 * p-value blocks have been gathered.  With num_permutations > 0 the p-values are normally
 * count / num_permutations (what safe_randomization and safe_outputs_from_* write): both matrices are
 * checked read-only, and if every entry is such a ratio the rows are adjusted from their histogram
 * over the counts, without a sort; otherwise (a caller's own values) they are sorted like the
 * hypergeometric form.  Nothing is modified before the form is chosen. */
int safe_fdr_adjust(safe_ctx *ctx, int64_t n, int64_t m, int64_t num_permutations, int sign_mode,
                    double enrichment_threshold, double *pvalues_neg_dev, double *pvalues_pos_dev, double *nes_dev,
                    double *nes_binary_dev, double *num_enriched_dev);

/* ---- consumers of nes_binary (SAFE.define_top_attributes / define_domains) ---------------- */
/* Connected components of the subgraph induced by the enriched nodes of each candidate
 * attribute (safepy/safe.py:640-655: nx.subgraph + nx.connected_components per attribute).
 * member_host: f64 [n, n_cols] row-major, > 0 = enriched (nes_binary[:, cols]); undirected edges
 * (edge_u[e], edge_v[e]).  labels_host: int32 [n_cols, n]: the smallest node id of the node's
 * component, -1 for nodes that are not enriched. */
int safe_enriched_components(safe_ctx *ctx, int64_t n, int64_t n_edges, const int32_t *edge_u, const int32_t *edge_v,
                             const double *member_host, int64_t n_cols, int32_t *labels_host);
/* scipy.spatial.distance.pdist(x, 'jaccard') as called inside linkage() at safepy/safe.py:673:
 * x_host f64 [m_top, n] row-major (non-zero = set), out_host f64 [m_top (m_top - 1) / 2] in SciPy's
 * condensed order; 0 where both profiles are empty. */
int safe_jaccard_condensed(safe_ctx *ctx, int64_t m_top, int64_t n, const double *x_host, double *out_host);

/* Counts of a whole call -> outputs (safepy/safe.py:528-554 and 468-472): counts_neg / counts_pos are #(S_p <= S_obs) /
 * #(S_p >= S_obs) as f64 [n, m] on the device (e.g. safe_permtest_counts results summed over the ranks of a
 * permutation-axis split), ns_dev the observed scores (NaN = no test; may be NULL).  Writes p-values, NES
 * (nes_table_host as in safe_randomization), nes_binary [n, m] and num_enriched [m].  Precondition: every count is a
 * whole number in [0, num_permutations] wherever the observed score is not NaN -- SAFE_E_VALUE otherwise (e.g. the
 * sum of ranks that each ran the full num_permutations). */
int safe_outputs_from_counts(safe_ctx *ctx, int64_t n, int64_t m, int64_t num_permutations, int sign_mode,
                             double enrichment_threshold, const double *nes_table_host, const double *counts_neg_dev,
                             const double *counts_pos_dev, const double *ns_dev, double *pvalues_neg_dev, double *pvalues_pos_dev,
                             double *nes_dev, double *nes_binary_dev, double *num_enriched_dev);
/* Multi-GPU exchange in integers (replaces gathering f64 NES blocks for the np.concatenate of
 * safepy/safe.py:1355): after safe_randomization / safe_permtest_counts on the bit-sliced or
 * matrix-core kernel, the raw counters of the call are still resident as
 * u32 [m][n_pad] = (#(S_p < S_obs) << 16 | #(S_p > S_obs)), attribute-major, so that a rank's
 * block is one contiguous slab.  safe_export_packed_counts copies them into dst_dev (NULL: only
 * report sizes); *layout is 0 / 1 for the two position orders, -1 if the last call kept no
 * counters (then exchange the f64 outputs instead).  After an all-gather of the slabs (every
 * rank holds the same membership handle, hence the same layout), safe_nes_from_packed_counts
 * turns [m_total][n_pad] counters into NES f64 [n, m_total] row-major with the arithmetic of
 * safepy/safe.py:532-554. */
int safe_export_packed_counts(safe_ctx *ctx, uint32_t *dst_dev, int64_t capacity, int64_t *n_pad, int64_t *m,
                              int *layout);
int safe_nes_from_packed_counts(safe_ctx *ctx, safe_nbr *nbr, const uint32_t *counts_dev, int layout, int64_t n_pad,
                                int64_t m, int64_t num_permutations, int sign_mode, const double *nes_table_host,
                                double *nes_dev);
/* The same exchange for EVERY matrix the reference's compute_pvalues leaves on the instance from the counters
 * (safepy/safe.py:532-554, 468-472): pvalues_neg, pvalues_pos, nes, nes_binary as f64 [n, m] row-major, any of them NULL =
 * not wanted.  One integer all-gather (4 bytes per node x attribute) then serves all four full matrices on every rank --
 * the "final RCCL all-gather of the p-value matrix" without moving 8-byte doubles or transposing blocks. */
int safe_outputs_from_packed_counts(safe_ctx *ctx, safe_nbr *nbr, const uint32_t *counts_dev, int layout, int64_t n_pad,
                                    int64_t m, int64_t num_permutations, int sign_mode, double enrichment_threshold,
                                    const double *nes_table_host, double *pvalues_neg_dev, double *pvalues_pos_dev,
                                    double *nes_dev, double *nes_binary_dev);
/* The same exchange in COLUMN CHUNKS (np.concatenate(axis=1) of safepy/safe.py:1355 is the one exchange of the reference's
 * sharded run, safe.py:1339-1355), optionally overlapped with the last launches of the permutation kernels.
 * safe_set_exchange_chunks(chunks, cols_per_chunk, cb, user) arms the following safe_randomization calls on this context
 * (chunks = 0: off): chunk k = this block's columns [k * cols_per_chunk, (k + 1) * cols_per_chunk), cols_per_chunk a
 * multiple of 64, 1..8 chunks covering the widest rank's block.  safe_export_packed_chunk(k, dst, capacity, stream) copies
 * chunk k's counters (u32 [columns][n_pad], the layout of safe_export_packed_counts; the rest of `capacity` is zeroed) on
 * `stream` -- after the call from the finished counters, so that every rank takes part in the same collectives whatever
 * kernel form its call took (the caller flags slabs that are not bit-sliced counters).
 * With SAFE_HIP_XCHG_TAIL=<fraction> in the environment and >= 2 chunks, the bit-sliced kernel also runs that last part of
 * the permutations as one launch per column chunk and calls cb(user) on the calling thread once every launch is enqueued --
 * before the call waits for them; safe_export_packed_chunk called from there makes `stream` wait until chunk k's counters
 * are final, so the all-gather of chunk k runs while the later chunks compute.  Off by default: at configs[1] it cost the
 * step more than the exchange it hides.  safe_packed_chunk_info tells what the last call did
 * (*chunks = 0: no column-chunked launches).  safe_outputs_from_packed_slabs derives the requested matrices from n_slabs
 * gathered slabs (slab r: slab_cols[r] columns of counters at slabs_dev + r * slab_stride, written to columns out_col0[r]...
 * of f64 [n, m_total] matrices), on `stream` (NULL: the context's), without waiting.  safe_randomization_plan: the counter
 * layout safe_randomization WOULD leave for this block (0 = bit-sliced kernel; -1 otherwise), so that ranks can settle the
 * form of the exchange in the collective they run before the kernels anyway. */
typedef void (*safe_enqueued_fn)(void *user);
int safe_set_exchange_chunks(safe_ctx *ctx, int chunks, int64_t cols_per_chunk, safe_enqueued_fn on_enqueued, void *user);
int safe_packed_chunk_info(safe_ctx *ctx, int *chunks, int64_t *bounds, int64_t *tail_permutations);
int safe_export_packed_chunk(safe_ctx *ctx, int chunk, uint32_t *dst_dev, int64_t capacity, void *stream);
/* The same slab with the counter pair packed to what the permutation count needs (the concatenated blocks of
 * safepy/safe.py:1355 carry counts of at most num_permutations, safe_extras.py:63-66): for num_permutations <= 1023 a pair is
 * 10 + 10 bits and two outputs travel in five bytes -- a column is n_pad / 2 words (low 32 bits of every 40) followed by
 * n_pad / 2 bytes (the high 8), 5 * n_pad / 8 words per column, 0.625 of the u32 slab.  capacity_words: u32 words at dst_dev
 * (the rest is zeroed).  safe_outputs_from_packed_slabs takes such slabs with SAFE_PACKED_NARROW added to `layout`
 * (slab_stride stays in u32 words).  More than 1023 permutations: use the u32 form. */
#define SAFE_PACKED_NARROW 16
int safe_export_packed_chunk_narrow(safe_ctx *ctx, int chunk, uint32_t *dst_dev, int64_t capacity_words, void *stream);
int safe_outputs_from_packed_slabs(safe_ctx *ctx, safe_nbr *nbr, const uint32_t *slabs_dev, int layout, int64_t n_pad, int n_slabs,
                                   int64_t slab_stride, const int64_t *slab_cols, const int64_t *out_col0, int64_t m_total,
                                   int64_t num_permutations, int sign_mode, double enrichment_threshold,
                                   const double *nes_table_host, double *pvalues_neg_dev, double *pvalues_pos_dev,
                                   double *nes_dev, double *nes_binary_dev, void *stream);
int safe_randomization_plan(safe_ctx *ctx, safe_nbr *nbr, safe_attr *attr, int64_t num_permutations, int score_type,
                            int *packed_layout);

/* The exchange step of the attribute-sharded path for hosts without their own collective library
 * (replaces np.concatenate(combined_nes, axis=1), safepy/safe.py:1355, and the process pool around it,
 * safe.py:1339-1353): a thin wrapper over RCCL (ncclAllGather over xGMI), loaded on first use.
 * One rank calls safe_comm_unique_id and hands the SAFE_COMM_ID_BYTES bytes to the others by any means (a
 * file, MPI, a socket); every rank then calls safe_comm_create with its own context (one GPU per rank).
 * safe_allgather_cols enqueues, on the context's stream, the all-gather of every rank's slab of
 * bytes_per_rank bytes at local_dev into all_dev (world_size slabs, rank order) -- e.g. the packed counters of
 * safe_export_packed_counts (attribute-major: a rank's column block IS one slab) or a transposed result
 * block; it returns without waiting (safe_ctx_sync).  SAFE_E_UNSUPPORTED if RCCL cannot be loaded. */
#define SAFE_COMM_ID_BYTES 128
int safe_comm_unique_id(char *id_out, size_t id_len);
int safe_comm_create(safe_ctx *ctx, int world_size, int rank, const char *id, size_t id_len, safe_comm **out);
int safe_comm_destroy(safe_comm *comm);
int safe_allgather_cols(safe_comm *comm, const void *local_dev, size_t bytes_per_rank, void *all_dev);

/* Number of i8 slices the last matrix-core permutation test ran with (2 / 4 / 6: the bits its columns need
 * on their fixed-point grid; 0 if that kernel has not run) -- for roofline reporting. */
int safe_last_mfma_slices(safe_ctx *ctx, int *slices);
/* The filtered form of that kernel (six-slice columns of 'sum' scores, np.dot of safepy/safe_extras.py:15 inside the
 * loop of :56-66): *core_slices = slices the matrix cores multiplied (3 when the filter ran: only the high digits; the
 * compares they cannot decide are settled exactly from the low digits), *undecided = how many compares that was
 * (negative: their list overflowed and the call was repeated with all six slices).  Counts are identical either way. */
int safe_last_mfma_filter(safe_ctx *ctx, int *core_slices, int64_t *undecided);
/* Diagnostics (bench.py's per-step probe): the number of hipMalloc / hipHostMalloc calls the library has made in this process.
 * Buffers are cached per context and per handle shape, so a repeated call of the same shape is expected to add none
 * (the reference allocates every [N, M] temporary anew on each pass: safe_extras.py:50-66). */
int safe_alloc_count(int64_t *calls);
/* Host placement of the seeded stream's sequential part (np.random.seed / np.random.permutation, safepy/safe_extras.py:46,58):
 * the CPUs the library's draw threads may run on (count = 0: wherever the process may).  For LAUNCHERS that place their own
 * threads (bench.py, run_batch): keeping the draw thread off the hardware threads that share a core with the launcher's
 * polling thread.  Applies to draw threads started after the call (one persistent thread per context, created at the
 * first seeded call).  The library never re-pins its caller's threads. */
int safe_set_draw_cpus(const int *cpus, int count);

/* Name and average duration (ms) of the dominant kernel of the last enrichment call,
 * measured with HIP events on the context stream (bench.py's roofline object). */
int safe_last_kernel_stats(safe_ctx *ctx, char *name, size_t name_len, double *avg_ms,
                           int64_t *launches);

/* Time (ms) during which at least one launch of that kernel was running in the last enrichment call: the union of the launches'
 * intervals.  Consecutive launches of the permutation kernels run on two streams and overlap, so launches x avg_ms exceeds it
 * (and can exceed the call).  Measurement only; no reference counterpart. */
int safe_last_kernel_busy_ms(safe_ctx *ctx, double *busy_ms);

#ifdef __cplusplus
}
#endif
#endif /* SAFE_HIP_H */
