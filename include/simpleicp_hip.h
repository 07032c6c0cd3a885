/*
 * simpleicp_hip.h -- C ABI of libsimpleicp_hip.so: the MI355X (gfx950) implementation of
 * the simpleICP inner loop.
 *
 * The reference (pglira/simpleICP, Python flavour) has NO FFI / plugin layer: its seam is
 * the Python API `SimpleICP.run` / `PointCloud` (python/simpleicp/__init__.py:12-14).  The
 * entry points below are what a ctypes binding for that seam needs; each one names the
 * reference operator (file:line under /root/reference) it stands in for.  The binding
 * itself is simpleicp_amd/_lib.py; INTEGRATION.md shows the stub a reference maintainer
 * would add.
 *
 * Conventions
 *   - every function returns SICP_OK (0) or a negative SICP_ERR_*; nothing throws across
 *     the ABI; sicp_last_error() returns a thread-local message for the last failure.
 *   - "host-or-device pointer": caller-owned, C-contiguous, valid only for the call; may be
 *     a host pointer (numpy) or a device pointer (e.g. torch.Tensor.data_ptr()) -- copies
 *     use hipMemcpyDefault.  Device memory created by the library is owned by the ctx.
 *   - calls are synchronous on return.  A ctx is not thread-safe; use one per thread/GPU.
 *   - point clouds are float64 (n,3) row-major, like PointCloud.X (pointcloud.py:81-84).
 *   - indices are int64, 0-based; in a sharded (multi-GPU) job they are GLOBAL indices:
 *     index_base (given at upload) + local row.
 *
 * Arithmetic contract (identical in oracle/sicp_oracle.c, which checks this library):
 *   (T) x' = fma(H02,z, fma(H01,y, H00*x)) + H03             (pointcloud.py:205-217)
 *   (D) d2 = fma(dz,dz, fma(dy,dy, dx*dx)), dx = p.x - q.x
 *   (K) neighbours ordered ascending by (d2, index); cKDTree's tie order is arbitrary
 *   (P) d  = (dx*nx + dy*ny) + dz*nz, no contraction, n upcast from float32
 *                                                            (corrpts.py:195-211)
 */
#ifndef SIMPLEICP_HIP_H
#define SIMPLEICP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 1: round 1.  2: + sicp_corr_*, sicp_estimate_parameters, sicp_comm_*, sicp_set_partition, sicp_cloud_set_planarity,
 * sicp_cloud_download_columns, sicp_match_work; sicp_timing_enable(ctx, 2); every upload resets the slot's planarity
 * column.  3: + sicp_comm_activate, sicp_comm_info, sicp_device_memory; SICP_K_XCHG; sicp_comm_init bounded + handshake.  4: + sicp_knn_work, sicp_cloud_download_both; sicp_estimate_normals runs the one-sweep k-NN + covariance kernel.  5: + sicp_match_deferred, sicp_tail_cycles; kind 6 of sicp_last_match_kernel.
 * 6: + SICP_XCHG_MIN_U64 / SICP_XCHG_MAX_U64 (asked of a registered callback with count = 0 first: a callback written for ABI 5 answers
 * non-zero and keeps the all-gather exchange), sicp_tail_selection, sicp_exchange_info.  7: + sicp_cloud_upload_start, sicp_cloud_upload_wait.  A binding checks sicp_abi_version() against the header it was written for. */
#define SICP_ABI_VERSION 7

#define SICP_OK               0
#define SICP_ERR_INVALID     -1   /* bad argument / wrong call order                         */
#define SICP_ERR_HIP         -2   /* HIP runtime failure (message has hipGetErrorString)     */
#define SICP_ERR_NO_DEVICE   -3   /* no usable gfx950 device                                 */
#define SICP_ERR_TOO_FEW     -4   /* < 6 correspondences left  (simpleicp.py:209-214)        */
#define SICP_ERR_NUMERIC     -5   /* normal matrix not positive definite / no progress       */
#define SICP_ERR_EXCHANGE    -6   /* the registered exchange callback failed                 */

#define SICP_FIX 0   /* fixed cloud   (pc1 in simpleicp.py:70-73) */
#define SICP_MOV 1   /* movable cloud (pc2)                        */

typedef struct sicp_ctx sicp_ctx;

/* ---- library / context ------------------------------------------------------------ */
int         sicp_abi_version(void);
const char *sicp_last_error(void);
int         sicp_device_count(int *count_out);
int         sicp_ctx_create(int device, sicp_ctx **ctx_out);
int         sicp_ctx_destroy(sicp_ctx *ctx);
int         sicp_ctx_device_name(sicp_ctx *ctx, char *buf, int buflen);

/* ---- point clouds : PointCloud (pointcloud.py:15-49) -------------------------------- */
/* Upload n points (host-or-device pointer) into slot SICP_FIX / SICP_MOV; replaces any
 * previous content.  index_base = global index of row 0 (0 unless the cloud is a shard). */
int sicp_cloud_upload(sicp_ctx *ctx, int slot, const double *xyz, int64_t n, int64_t index_base);
/* Same from three separate contiguous columns (how a DataFrame stores x, y, z once a column has been
 * assigned, pointcloud.py:215-217): they are copied straight into the column-wise device layout. */
int sicp_cloud_upload_columns(sicp_ctx *ctx, int slot, const double *x, const double *y, const double *z, int64_t n,
                              int64_t index_base);
/* The same upload BEHIND the caller (ABI 7): xyz (rows) or x, y, z (columns), the other(s) NULL.  Returns once the device arrays are
 * sized; a helper thread of the library moves the data on a stream of its own, so the caller can go on working on the OTHER slot
 * (SimpleICP.run builds the fixed cloud's grid and normals behind the movable cloud's upload, simpleicp.py:161-178).  The source
 * arrays must stay alive and unchanged until sicp_cloud_upload_wait -- or any other call naming the slot, which waits first and
 * returns the upload's error if it had one (then the slot is empty).  One at a time: a start on the other slot waits for the first to finish (whose verdict stays with its own slot).  Clouds
 * of at most 2^19 points and device pointers are uploaded on the spot. */
int sicp_cloud_upload_start(sicp_ctx *ctx, int slot, const double *xyz, const double *x, const double *y, const double *z,
                            int64_t n, int64_t index_base);
int sicp_cloud_upload_wait(sicp_ctx *ctx, int slot);
int sicp_cloud_size(sicp_ctx *ctx, int slot, int64_t *n_out);
/* PointCloud.transform_by_H (pointcloud.py:205-217): in-place, contract (T).  */
int sicp_cloud_transform(sicp_ctx *ctx, int slot, const double H[16]);
/* PointCloud.X (pointcloud.py:81-84): (n,3) row-major copy out. */
int sicp_cloud_download(sicp_ctx *ctx, int slot, double *xyz_out);
/* The same as three contiguous columns -- what `self[["x", "y", "z"]] = ...` leaves in the DataFrame after
 * transform_by_H (pointcloud.py:215-217): straight out of the column-wise device layout, no transpose on either side. */
int sicp_cloud_download_columns(sicp_ctx *ctx, int slot, double *x_out, double *y_out, double *z_out);
/* Both forms in one pass over the link: xyz_out (n, 3) row-major and / or the three columns (all three or none; xyz_out may be
 * null).  The columns are pulled through a pinned double buffer while host threads fan each chunk out (the row form is
 * transposed on the host) -- what PointCloud.transform_by_H needs after run() (simpleicp.py:316, pointcloud.py:205-217). */
int sicp_cloud_download_both(sicp_ctx *ctx, int slot, double *xyz_out, double *x_out, double *y_out, double *z_out);
/* The `planarity` column of a cloud that has one (CorrPts.reject_wrt_planarity also tests the MOVABLE cloud's
 * planarity of every matched point when pc2 carries that column, corrpts.py:158-163; NaN fails the test).
 * Only consulted for SICP_MOV by the iteration.  The column is indexed by GLOBAL point index and has n_global
 * entries (= the cloud size unless the cloud is a shard: every rank then holds the whole column, 4 B per point).
 *   rows == NULL : planarity is the dense column, m == n_global
 *   rows != NULL : m (row, value) pairs, NaN everywhere else (how estimate_normals leaves it: sparse)
 *   planarity == NULL : the cloud has no such column (also the state after every upload of the slot). */
int sicp_cloud_set_planarity(sicp_ctx *ctx, int slot, const int64_t *rows, const float *planarity, int64_t m,
                             int64_t n_global);

/* ---- nearest neighbours ------------------------------------------------------------- */
/* What the reference asks of scipy.spatial.cKDTree(...).query(q, k, p=2[, distance_upper_bound])
 * at corrpts.py:131-132 (k=1, cloud pre-transformed by H: simpleicp.py:188),
 * pointcloud.py:161-165 (k=1, strict upper bound) and pointcloud.py:185-186 (k=neighbors).
 *   H        : optional 4x4 row-major transform applied to the SEARCHED cloud on the fly
 *              (contract (T)); NULL = identity.  Only honoured for k == 1.
 *   max_dist : candidates need d2 < max_dist*max_dist (strict, like cKDTree); +inf = none.
 *   idx_out  : (Q,k) int64, ascending (d2, idx); -1 where fewer than k candidates exist.
 *   d2_out   : (Q,k) float64 SQUARED distances (+inf where idx == -1); may be NULL.
 * Results cover the LOCAL shard only (see sicp_set_exchange for multi-GPU). */
int sicp_knn(sicp_ctx *ctx, int slot, const double *q_xyz, int64_t Q, int k,
             const double *H, double max_dist, int64_t *idx_out, double *d2_out);

/* PointCloud.select_in_range (pointcloud.py:149-171) between two RESIDENT clouds -- the partial-overlap
 * pre-pass of SimpleICP.run (simpleicp.py:155-170) without moving a coordinate over the host link:
 * in_range_out[i] = 1 iff the nearest neighbour of point sel_idx[i] of cloud `query_slot` (of point i, Q
 * ignored, when sel_idx is NULL) among cloud `search_slot` transformed by H (NULL = identity) is closer
 * than max_range (strict, like cKDTree's distance_upper_bound).  Job-wide when an exchange is set. */
int sicp_select_in_range(sicp_ctx *ctx, int query_slot, int search_slot, const int64_t *sel_idx, int64_t Q,
                         const double *H, double max_range, uint8_t *in_range_out);

/* PointCloud.estimate_normals (pointcloud.py:173-203) for the rows sel_idx (LOCAL rows of
 * `slot`): k-NN among ALL points of the slot (self included), sample covariance (/(k-1)),
 * symmetric eigen-decomposition in fp64; normal = eigenvector of the smallest eigenvalue
 * with its largest-magnitude component made positive (np.linalg.eig's sign is arbitrary),
 * planarity = (l_mid - l_min) / l_max; both stored as float32 like the reference.
 *   normals_out (Q,3) float32, planarity_out (Q) float32, nn_idx_out (Q,k) int64 or NULL (on a binned cloud the neighbour
 *   lists are not even formed in memory unless asked for: one sweep per query, covariance from the winners' coordinates). */
int sicp_estimate_normals(sicp_ctx *ctx, int slot, const int64_t *sel_idx, int64_t Q, int k,
                          float *normals_out, float *planarity_out, int64_t *nn_idx_out);

/* ---- the ICP iteration (simpleicp.py:184-250) ---------------------------------------- */
/* Declare the selected fixed points and their attributes (simpleicp.py:173-178):
 *   sel_idx (Q) rows of the SICP_FIX slot, normals (Q,3) float32, planarity (Q) float32
 *   (NaN planarity = rejected, corrpts.py:153-155).  Stays resident across iterations.  */
int sicp_icp_setup(sicp_ctx *ctx, const int64_t *sel_idx, int64_t Q,
                   const float *normals, const float *planarity);

typedef struct sicp_iter_params {
    double x[6];            /* current estimate alpha1..3 [rad], tx,ty,tz: defines H for the
                               match AND the solver start (simpleicp.py:223-227,250)        */
    double obs[6];          /* rbp_observed_values, angles already in rad                   */
    double obs_weight[6];   /* rbp_observation_weights; +inf = parameter fixed              */
    double min_planarity;   /* corrpts.py:139-163                                           */
    double distance_weight; /* > 0; <= 0 or NaN = "None": 1/std(d)^2 (simpleicp.py:233-234) */
    int64_t max_lm_steps;   /* cap on solver steps (0 = default 100)                        */
} sicp_iter_params;

typedef struct sicp_iter_result {
    double  x[6];           /* new estimate (optimization.py:103-115)                       */
    double  H[16];          /* RigidBodyParameters.H (optimization.py:334-350)              */
    int64_t n_queries;      /* Q                                                            */
    int64_t n_planar;       /* survivors of reject_wrt_planarity                            */
    int64_t n_kept;         /* survivors of reject_wrt_point_to_plane_distances             */
    double  median, mad;    /* of the planarity survivors' distances (raw MAD, scale 1.0)   */
    double  dist_mean, dist_std;   /* kept point-to-plane distances BEFORE optimisation     */
    double  res_mean, res_std;     /* unweighted residuals AFTER optimisation (ddof = 0)    */
    double  weight_used;    /* distance weight actually applied                             */
    double  cost;           /* objective at x                                               */
    int64_t lm_steps;       /* accepted solver steps                                        */
    int64_t ne_evals;       /* fused normal-equation reductions launched                    */
} sicp_iter_result;

/* One full iteration on the GPU (simpleicp.py:186-250): match -- the EXACT nearest neighbour, order (K), of each of the
 * Q selected fixed points in the movable cloud under H(x); by default a pruned search on a uniform grid built once per
 * upload, queries pulled back by the rigid inverse of H (bit-identical to the brute-force scan, which runs instead for a
 * non-rigid H or on request) --, point-to-plane distances, planarity + raw-MAD rejection, then minimisation of the
 * reference's objective (optimization.py:65-124) by Levenberg-Marquardt on fused normal-equation reductions with the
 * 6x6 solves on the device (no host round trip inside the iteration; the operator sicp_estimate_parameters below is
 * the variant with the host-side solve).  A chain of length one of what sicp_icp_run enqueues. */
int sicp_icp_iterate(sicp_ctx *ctx, const sicp_iter_params *params, sicp_iter_result *result);

/* The whole iteration loop of simpleicp.py:184-261 in one call: repeats sicp_icp_iterate from
 * params->x, feeding every estimate into the next iteration, freezing an automatic distance
 * weight after the first iteration (simpleicp.py:229-234), until the reference's convergence test
 * (simpleicp.py:356-379: change of mean AND of std(ddof 0) of the residuals, in percent, both
 * < min_change; checked from the second iteration on) or max_iterations.  results[] receives one
 * entry per executed iteration, *iterations_out their number.  On SICP_ERR_TOO_FEW the failing
 * iteration's entry is filled (n_kept < 6) and counted. */
int sicp_icp_run(sicp_ctx *ctx, const sicp_iter_params *params, int64_t max_iterations, double min_change,
                 sicp_iter_result *results, int64_t *iterations_out);

/* State of the LAST iteration, each (Q)-sized in query order, host-or-device, any NULL:
 *   pc2_idx  matched movable index (corrpts.py:135), dist  distance before optimisation
 *   (corrpts.py:195-211), keep  1 = survived both rejections, residual  unweighted residual
 *   at the new estimate (0 where keep == 0).                                              */
int sicp_icp_get_state(sicp_ctx *ctx, int64_t *pc2_idx, double *dist, uint8_t *keep, double *residual);

/* estimate_parameter_uncertainties (optimization.py:126-170) at the last estimate; NaN for
 * fixed parameters. */
int sicp_icp_uncertainties(sicp_ctx *ctx, double sigma_out[6]);

/* Fused reduction exposed for parity tests: normal equations of the unweighted distance
 * residuals over the kept correspondences of the last iteration at parameters x:
 * out[0..20] upper triangle of J^T J (row-major), out[21..26] J^T r, out[27] sum r,
 * out[28] sum r^2, out[29] n. */
int sicp_icp_normal_equations(sicp_ctx *ctx, const double x[6], double out[30]);

/* ---- the iteration's operators one by one ------------------------------------------------ */
/* sicp_icp_iterate is the reference's loop body in one call.  A caller that drives the reference's classes itself
 * -- CorrPts(pc1, pc2).match() / .reject_wrt_planarity() / .reject_wrt_point_to_plane_distances(), then
 * SimpleICPOptimization(...).estimate_parameters() (simpleicp.py:190-227) -- binds these instead; they share the
 * iteration's kernels and its state (sicp_icp_setup declares pc1's selected points and normals; the movable cloud is
 * pc2.X_selected).  A correspondence is "alive" until a rejection drops it (the reference drops the row from its
 * DataFrame, corrpts.py:156,163,188); sicp_icp_get_state returns the alive mask as `keep`.
 *
 * CorrPts.match (corrpts.py:124-137) + __compute_point_to_plane_distances (corrpts.py:195-211): nearest movable
 * point of every selected fixed point under H (NULL = identity: the caller has transformed pc2 itself, as
 * simpleicp.py:188 does) and the signed point-to-plane distance to it; every correspondence is alive afterwards.
 *   pc2_idx_out (Q) int64 (global index), dist_out (Q) float64; either may be NULL. */
int sicp_corr_match(sicp_ctx *ctx, const double *H, int64_t *pc2_idx_out, double *dist_out);
/* CorrPts.reject_wrt_planarity (corrpts.py:139-163): alive &= planarity >= min_planarity (float32 compare, NaN
 * fails) for the correspondence's point in pc1 and in pc2.  The columns are handed over PER CORRESPONDENCE, (Q)
 * float32 in query order -- the reference's `pc.iloc[idx]["planarity"]`; NULL = that cloud has no such column and
 * is not tested (corrpts.py:151,158). */
int sicp_corr_reject_planarity(sicp_ctx *ctx, double min_planarity, const float *pc1_planarity,
                               const float *pc2_planarity, int64_t *n_alive_out);
/* CorrPts.reject_wrt_point_to_plane_distances (corrpts.py:165-188) over the alive correspondences: median (mean of
 * the two middle values), raw MAD (scale 1.0), alive &= |d - median| <= 3 MAD.  median / mad are NaN when nothing
 * was alive. */
int sicp_corr_reject_distances(sicp_ctx *ctx, double *median_out, double *mad_out, int64_t *n_alive_out);
/* SimpleICPOptimization.estimate_parameters (optimization.py:65-124) over the alive correspondences: minimises the
 * reference's objective from params->x (min_planarity is ignored here) and leaves the unweighted residuals for
 * sicp_icp_get_state, the estimate for sicp_icp_uncertainties.
 *   pc2_xyz : optional (Q,3) coordinates of the matched movable points as they are NOW (host-or-device) -- the
 *             reference reads pc2's coordinates when it optimises, after it has undone the match's transform
 *             (simpleicp.py:202); NULL = the coordinates the match saw.
 * SICP_ERR_TOO_FEW below 6 alive correspondences (simpleicp.py:209-214). */
int sicp_estimate_parameters(sicp_ctx *ctx, const sicp_iter_params *params, const double *pc2_xyz,
                             sicp_iter_result *result);

/* mathutils.py:39-68,81-93: parameters -> 4x4 row-major H (host only). */
int sicp_params_to_H(const double x[6], double H_out[16]);

/* ---- multi-GPU exchange hook (one process per GPU; collectives supplied by the host) -- */
/* Variant 1 -- collectives supplied by the host: the host binding (torch.distributed over RCCL/xGMI in
 * simpleicp_amd/dist.py) registers a callback the iteration calls at its exchange points.
 * All pointers handed to the callback are DEVICE pointers owned by the ctx.  The callback must either
 * ENQUEUE the collective in order on the library's stream (sicp_ctx_stream; e.g. under
 * torch.cuda.stream(torch.cuda.ExternalStream(ptr)) -- nothing then blocks the host), or finish it
 * (results visible in device memory) before returning, after synchronising that stream itself.
 * Return 0 on success.
 *   SICP_XCHG_ALLGATHER_F64 : a = send f64[count], b = recv f64[world*count]; fill b with every
 *                             rank's `a` in rank order (the library packs per-query
 *                             (d2, idx, x, y, z) records into `a` and afterwards reduces `b` to
 *                             the job-wide lexicographic (d2, idx) minimum with its own kernel).
 *   SICP_XCHG_SUM_F64       : a = f64[count]; replace in place by the sum over ranks.
 *   SICP_XCHG_MIN_U64, SICP_XCHG_MAX_U64 (ABI 6; optional) : a = uint64[count]; replace in place by the element-wise minimum /
 *                             maximum over ranks, the words compared as UNSIGNED integers.  With them cloud shards of many queries
 *                             (SICP_XCHG_KEYS_MIN_Q, default 32 768) find the job-wide winner by three reductions on 8-byte keys
 *                             -- min of the squared distance's bits, min of the index among the holders of that minimum, max of
 *                             the owner's coordinate bits against zeros -- instead of gathering 40 bytes per query AND RANK.
 *                             sicp_set_exchange asks once with count = 0 and a = NULL (no data, no collective: just return 0 if
 *                             the operation is served, non-zero if not); a callback that declines keeps the all-gather road. */
#define SICP_XCHG_ALLGATHER_F64 1
#define SICP_XCHG_SUM_F64       2
#define SICP_XCHG_MIN_U64       3
#define SICP_XCHG_MAX_U64       4
typedef int (*sicp_exchange_fn)(void *user, int what, void *a, void *b, void *c, int64_t count);
/* With an exchange registered, sicp_knn(k == 1) and the iteration's match return JOB-WIDE winners.
 * gn_shard: 0 = every rank reduces all correspondences (no collective in the solver),
 *           1 = rank r reduces slice r of the correspondences + ONE SUM exchange per evaluation: the 8x8 Gram block
 *               [J^T J | J^T r | sum r, sum r^2, n] (64 doubles) on the device solver (more than 2048 correspondences; below
 *               that the single-workgroup tail ignores the flag), the 30 sums of the host-side solve under SICP_SOLVE=host. */
int sicp_set_exchange(sicp_ctx *ctx, sicp_exchange_fn fn, void *user, int rank, int world, int gn_shard);
/* The same exchange issued by the library itself: one RCCL communicator per ctx (one process per GPU), collectives
 * enqueued on the ctx's stream between its kernels -- no host callback, nothing blocks (SURVEY 8b `sicp_comm_init`).
 * librccl is loaded on demand; it is not a link-time dependency.
 *   sicp_comm_unique_id : rank 0 obtains the 128-byte id (ncclGetUniqueId) and hands it to the other ranks by any
 *                         transport (simpleicp_amd/dist.py broadcasts it over torch.distributed);
 *   sicp_comm_init      : collective over all ranks (ncclCommInitRank); replaces a registered callback;
 *   sicp_comm_destroy   : back to single-GPU behaviour. */
int sicp_comm_unique_id(void *id128);
int sicp_comm_init(sicp_ctx *ctx, const void *id128, int rank, int world, int gn_shard);
int sicp_comm_destroy(sicp_ctx *ctx);
/* sicp_comm_init never blocks for good: the rendezvous (ncclCommInitRank) and a first small all-gather on the ctx's stream
 * are both awaited with a deadline (SICP_COMM_TIMEOUT_S, default 60 s) and what RCCL reports (ncclCommCount /
 * ncclCommUserRank) is checked against rank / world; any failure returns SICP_ERR_EXCHANGE and leaves the ctx without a
 * communicator, so the host can send every rank down another road together.  Inside a run a result that does not arrive within
 * twice that deadline (default 120 s) while collectives are in flight is SICP_ERR_EXCHANGE as well, not a hang.
 *   sicp_comm_activate : the communicator stays with the ctx between runs; on = 1 makes the ctx's searches and iterations
 *                        job-wide again (with this gn_shard), on = 0 parks it (single-GPU behaviour, communicator kept).
 *   sicp_comm_info     : out[0] exchange in force: 0 none, 1 host callback, 2 the library's RCCL communicator;
 *                        out[1] ranks and out[2] this rank -- for 2 as RCCL itself counts them (ncclCommCount /
 *                        ncclCommUserRank); out[3] partition; out[4] gn_shard; out[5] a communicator exists (active or parked). */
int sicp_comm_activate(sicp_ctx *ctx, int on, int gn_shard);
int sicp_comm_info(sicp_ctx *ctx, int out[6]);
/* What the chained iterations' exchanges did since sicp_icp_setup (ABI 6): out4[0] form of the LAST one -- 0 none yet, 1 all-gather of
 * 40-byte (d2, index, xyz) records + lexicographic minimum (cloud shards), 2 three all-reduces on 8-byte keys (cloud shards from
 * SICP_XCHG_KEYS_MIN_Q queries), 3 all-gather of the query slices' matched indices (query shards); out4[1] how many exchanges ran;
 * out4[2] the key exchange's threshold in queries (0 = never); out4[3] 1 if a registered callback serves SICP_XCHG_MIN_U64 / MAX_U64. */
int sicp_exchange_info(sicp_ctx *ctx, int64_t out4[4]);
/* free / total bytes of the ctx's device (hipMemGetInfo): what a host consults before it replicates a cloud on every rank */
int sicp_device_memory(sicp_ctx *ctx, int64_t *free_out, int64_t *total_out);
/* What the ranks shard:
 *   SICP_PART_CLOUD   (default) every rank holds a contiguous index range of the searched cloud and all Q queries; one
 *                     all-gather of per-query winners + lexicographic minimum per iteration;
 *   SICP_PART_QUERIES every rank holds the WHOLE searched cloud (index_base 0) and matches Q / world of the queries;
 *                     one all-gather of the slices' matched indices per iteration (8 bytes per query), no reduction.  Pays off when the match dominates
 *                     (Q >= ~1e5); needs the default grid search.  sicp_knn / sicp_select_in_range are then local. */
#define SICP_PART_CLOUD   0
#define SICP_PART_QUERIES 1
int sicp_set_partition(sicp_ctx *ctx, int mode);
/* the HIP stream (hipStream_t) every kernel and copy of this ctx is issued on */
int sicp_ctx_stream(sicp_ctx *ctx, void **stream_out);

/* Reduction used after the all-gather, exposed for tests: gathered = [world][Q][5] records
 * (d2, idx as int64 bits, x, y, z); outputs the lexicographic (d2, idx) minimum per query over the
 * records with idx >= 0 (idx -1 / d2 +inf / xyz 0 when none).  Host-or-device pointers. */
int sicp_lexmin_gathered(sicp_ctx *ctx, const double *gathered, int world, int64_t Q,
                         double *d2_out, int64_t *idx_out, double *xyz_out);

/* ---- .xyz text I/O (host only, multithreaded) --------------------------- */
/* What the reference's callers do with np.genfromtxt (python/simpleicp/tests/test_simpleicp.py:102-103)
 * and PointCloud.write_xyz / CorrPts.write_xyz (pointcloud.py:219-226, corrpts.py:213-237).
 * Data rows are the lines that start with a number; the first three columns are read.  Values are
 * parsed with strtod and printed with printf, i.e. identical to Python's float() / "%.3f". */
int sicp_xyz_count(const char *path, int64_t *rows_out);
int sicp_xyz_read(const char *path, double *xyz_out, int64_t capacity_rows, int64_t *rows_out, int threads);
/* decimals >= 0: "%.<decimals>f"; < 0: "%.18e" (np.savetxt default).  header may be NULL. */
int sicp_xyz_write(const char *path, const double *data, int64_t n, int cols, int decimals,
                   const char *header, int threads);

/* ---- kernel timing (HIP events on the library's own stream) -------------------------- */
#define SICP_K_KNN1     0   /* the 1-NN search of the match, whichever flavour ran (sicp_last_match_kernel)          */
#define SICP_K_KNNK     1   /* the k-NN search of estimate_normals                                                  */
#define SICP_K_NORMALEQ 2   /* the solver's launches: k_icp_tail (Q <= 2048: the whole tail) or k_lm_eval/k_lm_finish */
#define SICP_K_SELECT   3   /* distances + median / MAD selection + keep mask when they are launches of their own   */
#define SICP_K_XCHG     4   /* the multi-GPU exchange of an iteration: pack + collective + unpack / lexicographic minimum   */
#define SICP_K_COUNT    5
int sicp_timing_enable(sicp_ctx *ctx, int on);   /* 0 off, 1 kernel timing, 2 timing + the grid search's work tallies (sicp_match_work) */
/* which 1-NN flavour the last sicp_knn(k=1) / sicp_icp_iterate used: 0 exact scan, 1 filtered scan
 * with inline verification, 2 grid search, 3 filtered scan (VALU filter) with recorded candidates + fix-up kernel,
 * (4: the matrix-pipe filter of ABI <= 3, removed), 5 grid search with four / eight queries per wave, exact arithmetic on every
 * candidate, 6 grid search with four / eight queries per wave through the float32 filter (+ the exact kernel for what the filter
 * cannot decide) -- all return identical results */
int sicp_last_match_kernel(sicp_ctx *ctx, int *kind_out);
int sicp_timing_reset(sicp_ctx *ctx);
/* Work the pruned grid search did in its launches since sicp_timing_reset, counted by the kernel itself while
 * sicp_timing_enable(ctx, 2) is in force: out3[0] candidates evaluated (one 32-byte record read each), out3[1] non-empty grid rows
 * visited (two 4-byte offsets each), out3[2] launches -- the bytes the bench prices the search's roofline on. */
int sicp_match_work(sicp_ctx *ctx, uint64_t out3[3]);
/* ... and how many queries the float32-filtered search (kind 6) left to the exact kernel in those launches: ties within the
 * filter's margin and queries float32 cannot place.  (The filtered search reads a 16-byte record per candidate, plus the winner's
 * 32-byte record per query and pass.) */
int sicp_match_deferred(sicp_ctx *ctx, uint64_t *out);
/* The single-workgroup tail's own clock (shader cycles) over the phases of the LAST iteration it ran (Q <= 2048): out5[0] loading the
 * distances and verdicts the match left, [1] median + MAD selection, [2] keep mask + statistics, [3] the minimisation (residual +
 * Jacobian evaluations, 6x6 solves), [4] statistics of the residuals, convergence test, the record.  What the bench's latency model
 * (kernel boundaries + dependent memory round trips + these on-chip phases) is built from.  Zeros before the first such iteration. */
int sicp_tail_cycles(sicp_ctx *ctx, double out5[5]);
/* How the single-workgroup tail (Q <= 2048) found median and MAD (corrpts.py:165-188) in the LAST iteration it ran: out3[0] / out3[1]
 * range-histogram rounds spent on the median / the MAD -- 0 = read off a window around the previous iteration's value (the settled
 * iterations of a chained run; same keys, same order statistic, same bits) --, out3[2] iterations since sicp_icp_setup in which
 * both came from their windows. */
int sicp_tail_selection(sicp_ctx *ctx, int64_t out3[3]);
/* Work the one-sweep k-NN (sicp_estimate_normals, sicp_knn with k > 1 on a binned cloud) did since sicp_timing_reset, under
 * sicp_timing_enable(ctx, 2): out4[0] candidates read (one 32-byte record each), out4[1] sweeps (a query needs one when its first
 * ball holds k points), out4[2] queries that took the k-round extraction instead, out4[3] candidates inside their query's ball. */
int sicp_knn_work(sicp_ctx *ctx, uint64_t out4[4]);
int sicp_timing_get(sicp_ctx *ctx, int kernel, double *total_ms_out, int64_t *launches_out);

#ifdef __cplusplus
}
#endif
#endif /* SIMPLEICP_HIP_H */
