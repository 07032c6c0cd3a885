// sicp_internal.h -- shared between the host side (sicp_api.cpp) and the kernels (sicp_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sicp {

// rows 0..2 of a 4x4 row-major homogeneous transform (row 3 is 0 0 0 1)
struct Xf { double m[12]; };

constexpr int    KNN_BLOCK = 256;     // lanes per workgroup of the scan kernels (4 waves, 1 per SIMD)
constexpr int    KNN1_R    = 4;       // queries per lane in the 1-NN scan
constexpr int    TILE_PTS  = 1024;    // cloud points staged in LDS per step (24 KiB)
constexpr int    FS_TILE   = 512;     // points per LDS tile of the filtered scan (float4: 8 KiB, x2 buffers)
constexpr int    FS_R      = 8;       // queries per lane of the filtered scan
constexpr int    FS_G      = 8;       // points between two slow-path checks of the filtered scan
constexpr int    REJECT_MAX_Q = 16384;   // single-workgroup rejection: its uint64 keys live in LDS (128 KB)
constexpr int    QPAD      = 2048;    // query padding granule (covers R = 4 and R = 8 scan blocks)
constexpr int    NE_BLOCK  = 256;
constexpr int    NE_MAX_GRID = 1024;
constexpr long   STATS_MB_MIN_Q = 16384;  // above: mean / std over many workgroups (two launches) instead of one CU
#define SICP_PAD_COORD 1.0e300

constexpr int SOLVE_MAX_Q = 2048;   // single-launch tail (sicp_tail.hip): 8 staged Jacobian columns x 2048 x 8 B = 128 KiB of LDS
void launch_fill_f32(hipStream_t s, float *dst, long n, float v);
void launch_corr_planarity(hipStream_t s, uint8_t *alive, const float *pl1, const float *pl2, float min_planarity, long Q);
void launch_scatter_f32(hipStream_t s, float *dst, const int64_t *rows, const float *vals, long m);

// ---- device-chained iteration loop (sicp_tail.hip) ----
// Loop state that lives in device memory for a whole run: each tail launch starts from it and leaves the next
// iteration's start there, so the launches of consecutive iterations are enqueued without a host round trip.
struct IcpDev {
    double x[6];                  // estimate the next iteration starts from
    double sc[6];                 // sin, cos of x[0..2]
    Xf H, Hinv;                   // H(x) and its rigid inverse [R^T | -R^T t]: the transform of the next match
    double w;                     // distance weight; <= 0: automatic, frozen by the first iteration that runs
    double prev_mean, prev_std;   // residual statistics of the last completed iteration (convergence test)
    int done_iters;               // iterations completed in this run
    int stop;                     // the run is over (converged / failed): later launches of the chain exit at once
    int sel_m;                    // the single-workgroup tail's last selection: how many distances took part ...
    int pad;
    double sel_med, sel_mad;      // ... their median and MAD (sel_mad > 0: the next launch looks for both in a window around these first)
};
// what the match kernel of a chained iteration leaves besides the match itself (null dist = nothing): the point-to-plane
// distance and the planarity verdict per query, i.e. k_postmatch's output
struct PostMatch {
    const float *normals;         // (Q,3)
    const float *planarity;       // (Q)
    const float *pl2;             // movable cloud's planarity column by GLOBAL index, or null
    long pl2_n;
    float min_planarity;
    double *dist;                 // (Q) out
    uint8_t *flag;                // (Q) out: 1 = matched and planar enough in both clouds
    double *pack;                 // nullable, (Q, 5) out: the exchange's (d2, index bits, x, y, z) record of this rank's winner --
                                  // what k_pack_best would re-read 48 bytes per query for, in a launch of its own
    double *pack_idx;             // nullable, (Q) out: query shards' slim record, the matched index as int64 bits (k_pack_idx's output)
};
struct TailArgs {
    double obs[6], ow[6];
    double min_change;            // simpleicp.py:356-379, percent; < 0: no convergence test (single-iteration API)
    double seq;                   // completion ticket published with this iteration's record
    float min_planarity;
    int max_steps;
    int Q;
    int window;                   // k_icp_tail: 1 = try the window around the last launch's median / MAD first (sicp_tail.hip)
    const float *pl2;             // movable cloud's planarity column by GLOBAL index (corrpts.py:158-163), or null
    long pl2_n;
};
// per-iteration record the tail streams into pinned host memory (doubles):
// 0 n_planar, 1 median, 2 mad, 3 n_kept, 4 dist_mean, 5 dist_std, 6 w_used, 7 cost, 8 lm_steps, 9 ne_evals,
// 10..15 x, 16 res_mean, 17 res_std, 18 status, 19 converged, 20..49 normal equations at x, 50..54 phase cycles
constexpr int REC_STATUS = 18;      // 0 ok / 1 too few correspondences / 2 objective not finite / 3 skipped (run already over) / 4 a grid barrier timed out
constexpr int REC_CONVERGED = 19;
constexpr int REC_RESID_SLOT = 61;  // larger Q: which of the two residual buffers holds the accepted residuals
constexpr int REC_TICKET = 63;
constexpr int REC_DOUBLES = 64;
void launch_icp_tail(hipStream_t s, const double *qx, const double *qy, const double *qz, const float *normals,
                     const double *p2, const TailArgs &A, IcpDev *st, const double *dist, const uint8_t *flag,
                     uint8_t *keep, double *resid, double *rec);

// solver state of the multi-workgroup evaluation chain (sicp_lm.hip, Q > SOLVE_MAX_Q)
struct LmDev {
    double x[6], sc[6];           // accepted estimate, sin / cos of its angles
    double xt[6], sct[6];         // trial the next evaluation runs at
    double G[2][64];              // 8x8 Gram matrices of the rows [a0..a5 | r | 1]: [cur] accepted, [cur ^ 1] trial
    double stat[2][2];            // per slot: sum (r - shift), sum (r - shift)^2
    double shift;                 // the kept distances' mean: residual statistics are accumulated relative to it
    double w, cost, lambda, dxmax;
    int cur, first, tries, steps, evals, done, pad[2];
};
int  lm_eval_grid(long Q);
void launch_lm_eval(hipStream_t s, const double *qx, const double *qy, const double *qz, const float *normals, const double *p2,
                    const uint8_t *keep, long Q, const TailArgs &A, const IcpDev *st, LmDev *L, const double *stats, double *partial,
                    unsigned *ticket, double *resid0, double *resid1, int rank = 0, int world = 1, double *gsum = nullptr);
void launch_lm_all(hipStream_t s, const double *qx, const double *qy, const double *qz, const float *normals, const double *p2,
                   const uint8_t *keep, long Q, const TailArgs &A, IcpDev *st, LmDev *L, const double *rj4, const double *stats,
                   double *partial, void *bar, unsigned long long *bar_total, double *resid0, double *resid1, double *rec,
                   unsigned absent = 0);
size_t lm_bar_bytes();
void launch_lm_advance(hipStream_t s, const TailArgs &A, const IcpDev *st, LmDev *L, const double *stats, const double *gsum);
void launch_lm_finish(hipStream_t s, const double *qx, const double *qy, const double *qz, const float *normals, const double *p2,
                      const uint8_t *keep, long Q, const TailArgs &A, IcpDev *st, LmDev *L, const double *rj4, const double *stats,
                      double *resid0, double *resid1, double *rec);

// uniform grid over a cloud in its own frame (sicp_grid.hip)
struct GridGeom { double mn[3]; double h, inv_h; int dim[3]; };

void launch_cloud_stats(hipStream_t s, const double *x, const double *y, const double *z, long n, unsigned long long *out7);
void launch_cell_ids(hipStream_t s, const double *x, const double *y, const double *z, long n, const GridGeom &G,
                     uint32_t *ids, uint32_t *counts, unsigned long long *occupied);
void launch_cell_ids_tiled(hipStream_t s, const double *x, const double *y, const double *z, long n, const GridGeom &G,
                           uint32_t *ids, uint32_t *counts);       // G.dim[0], G.dim[1]: multiples of 8
void launch_window_probe(hipStream_t s, const double *x, const double *y, const double *z, long n, long every, const GridGeom &Gw,
                         const double wmax[3], uint32_t *counts, unsigned long long *out2);
long grid_scan_blocks(long n);
void launch_grid_scan(hipStream_t s, const uint32_t *in, long n, uint32_t *block_off, uint32_t *out, uint32_t *cursor);
void launch_grid_scan_sums(hipStream_t s, const uint32_t *in, long n, uint32_t *block_off, unsigned long long *sumsq);
void launch_grid_scan_rest(hipStream_t s, const uint32_t *in, long n, uint32_t *block_off, uint32_t *out, uint32_t *cursor);
void launch_scatter(hipStream_t s, const double *x, const double *y, const double *z, const uint32_t *ids, long n,
                    uint32_t *cursor, void *rec);
// rec: the cloud in cell order as packed 32-byte records (x, y, z, original row as int64 bits)
// cell_box (nullable): the cells' tight boxes (sicp_grid_dev.h): far searches trim their rows by them
// a second, coarse grid over the SAME points (clouds whose density varies by orders of magnitude): wide passes of the exact search run on it
struct GridLevel { GridGeom g; const uint32_t *cell_start; const void *rec; };
constexpr int NN_TIGHT = 1;      // prev_p2 is a bound to search in one go (the nearest point of a subsample), not an old match
constexpr int NN_APPROX = 2;     // the first hit is good enough: the caller wants a cloud point NEAR the query (a bound), not the nearest
void launch_grid_nn(hipStream_t s, const double *qx, const double *qy, const double *qz, long Q, const double *prev_p2,
                    const GridGeom &G, const uint32_t *cell_start, const void *rec, const Xf *H, const Xf *Hinv, double rmax,
                    double max_d2, int64_t idx_base, double *d2_out, int64_t *idx_out, double *p2_out, unsigned long long *work,
                    bool four_per_wave, const unsigned long long *cell_box = nullptr, const GridLevel *coarse = nullptr);
void launch_grid_nn_chained(hipStream_t s, const double *qx, const double *qy, const double *qz, long Q, const double *prev_p2,
                            const GridGeom &G, const uint32_t *cell_start, const void *rec, const IcpDev *st, double rmax,
                            int64_t idx_base, double *d2_out, int64_t *idx_out, double *p2_out, unsigned long long *work,
                            const uint32_t *order, bool four_per_wave, int flags = 0, const PostMatch *post = nullptr,
                            bool eight_per_wave = false, const unsigned long long *cell_box = nullptr, const GridLevel *coarse = nullptr);
void launch_grid_nn_redo(hipStream_t s, const double *qx, const double *qy, const double *qz, long Q, const double *prev_p2,
                         const GridGeom &G, const uint32_t *cell_start, const void *rec, const IcpDev *st, const Xf *H, const Xf *Hinv,
                         double rmax, double max_d2, int64_t idx_base, double *d2_out, int64_t *idx_out, double *p2_out,
                         unsigned long long *work, int flags, const PostMatch *post, const unsigned long long *cell_box,
                         const uint32_t *redo_list, const unsigned *redo_count, unsigned *redo_clear, const GridLevel *coarse = nullptr);
// sicp_gridf.hip: the grid's lazily built companions and the filtered many-queries search
void launch_recf(hipStream_t s, const void *rec, long n, const double c0[3], void *recf);
void launch_cell_boxes(hipStream_t s, const uint32_t *cell_start, const void *rec, long ncells, const GridGeom &G, unsigned long long *cell_box);
void launch_slot_queries(hipStream_t s, const double *qx, const double *qy, const double *qz, const uint32_t *order, const double *prev_p2,
                         long Q, void *qrec, void *pslot);
void launch_slot_bounds(hipStream_t s, const void *qrec, const int64_t *idx, const double *p2, long Q, void *pslot);
// far: the flavour with row batches, hit-driven culling and box trimming (a run's first iterations); otherwise the lean flavour,
// which marks the queries it cannot do in `state` (1) for a launch of the other flavour over the same slots
void launch_grid_nn16f(hipStream_t s, int lanes_per_query, bool far, const IcpDev *st, const void *qrec, void *pslot, long Q,
                       const GridGeom &G, const double c0[3], double eps_p, const uint32_t *cell_start,
                       const void *recf, const void *rec, bool xcd_order, const Xf *H,
                       const Xf *Hinv, double rmax, double max_d2, int64_t idx_base, double *d2_out, int64_t *idx_out, double *p2_out,
                       unsigned long long *work, int flags, uint8_t *state, uint32_t *redo_list, unsigned *redo_count);
void launch_stride_sample(hipStream_t s, const double *x, const double *y, const double *z, long n, long stride, long m, long mpad,
                          double *out);
void launch_scatter_order(hipStream_t s, const uint32_t *ids, long n, uint32_t *cursor, uint32_t *order);
void launch_grid_knn(hipStream_t s, const double *qx, const double *qy, const double *qz, long Q, int k, const GridGeom &G,
                     const uint32_t *cell_start, const void *rec, double rmax, int64_t idx_base, double *d2_out, int64_t *idx_out);

bool grid_knn_sweep_handles(int k);
void launch_grid_knn_sweep(hipStream_t s, const double *qx, const double *qy, const double *qz, const uint32_t *order, long Q, int k,
                           const GridGeom &G, double avg_per_cell, const uint32_t *cell_start, const void *rec, double rmax,
                           int64_t idx_base, double *d2_out, int64_t *idx_out, double *cov /* (Q, 6) scratch when normals are asked for */,
                           float *normals, float *planarity, unsigned long long *work, long batch_override = 0,
                           uint32_t *redo = nullptr /* Q + 1 words: enables the four-queries-per-wave kernel */, int group = 0 /* 0 auto, 1, 4 */);

void launch_pack_best(hipStream_t s, const double *d2, const int64_t *idx, const double *p2, long Q, double *rec);
// the job-wide winner by all-reduces on 8-byte keys (sicp_kernels.hip, "the same winner by three all-reduces")
void launch_xkey_d2(hipStream_t s, const double *d2, const int64_t *idx, long Q, unsigned long long *key);
void launch_xkey_idx(hipStream_t s, const double *d2, const int64_t *idx, const unsigned long long *gmin, long Q, unsigned long long *key);
void launch_xkey_xyz(hipStream_t s, const int64_t *idx, const double *p2, const unsigned long long *gidx, long Q, unsigned long long *xyz);
void launch_xkey_unpack(hipStream_t s, const unsigned long long *gmin, const unsigned long long *gidx, const unsigned long long *xyz,
                        long Q, double *d2, int64_t *idx, double *p2);
void launch_pack_idx(hipStream_t s, const int64_t *idx, long cnt, long per, double *out);
void launch_unpack_idx_postmatch(hipStream_t s, const double *gathered, long Q, const double *cx, const double *cy, const double *cz,
                                 int64_t idx_base, long n, const double *qx, const double *qy, const double *qz, const float *normals,
                                 const float *planarity, float min_planarity, const float *pl2, long pl2_n, const IcpDev *st,
                                 int64_t *idx, double *p2, double *dist, uint8_t *flag);
void launch_lexmin_gathered(hipStream_t s, const double *g, int world, long Q, double *d2, int64_t *idx, double *p2);
void launch_lexmin_postmatch(hipStream_t s, const double *g, int world, long Q, const double *qx, const double *qy, const double *qz,
                             const float *normals, const float *planarity, float min_planarity, const float *pl2, long pl2_n,
                             const IcpDev *st, double *d2, int64_t *idx, double *p2, double *dist, uint8_t *flag);
size_t reject_select_scratch_bytes();
long resident_blocks(const void *kernel, int threads);   // blocks the current device holds at once (grid-barrier kernels)
hipError_t hsel_state_init(hipStream_t s, void *state);
hipError_t reject_by_select_one_launch(hipStream_t s, const double *dist, const uint8_t *flag, long Q, uint8_t *keep, double *out4,
                                       double *out3, void *state, unsigned long long *bar_total, double *partial, double *host_out,
                                       double seq, const IcpDev *st, unsigned absent = 0, bool use_prior = false);
void launch_aos_to_soa(hipStream_t s, const double *aos, long n, long npad, double *x, double *y, double *z);
void launch_found_mask(hipStream_t s, const int64_t *idx, long Q, uint8_t *out);
void launch_pad_fill(hipStream_t s, double *x, double *y, double *z, long n, long npad);
void launch_pack_chunks(hipStream_t s, const double *x, const double *y, const double *z, long n, long CH, double *out);
void launch_soa_to_aos(hipStream_t s, const double *x, const double *y, const double *z, long n, double *aos);
void launch_transform(hipStream_t s, double *x, double *y, double *z, long n, const Xf &H);
void launch_gather_queries(hipStream_t s, const double *x, const double *y, const double *z, const int64_t *sel,
                           long Q, long qpad, double *qx, double *qy, double *qz);
void launch_aos_queries(hipStream_t s, const double *aos, long Q, long qpad, double *qx, double *qy, double *qz);
void launch_knn1_scan(hipStream_t s, const double *qx, const double *qy, const double *qz, int qpad, int qblocks,
                      const double *px, const double *py, const double *pz, long npad, int tile_step,
                      int tiles_per_chunk, int nchunks, const Xf *H, double *part_d2, uint32_t *part_idx);
void launch_knn1_fscan(hipStream_t s, int block, const double *qx, const double *qy, const double *qz, int qpad, long Q,
                       int qblocks, const double *bound, const double *px, const double *py, const double *pz,
                       int ntiles, int nparts, const Xf *H, double rmax, double *part_d2, uint32_t *part_idx);
int  fscan_blocks_per_cu(int block);
void launch_knn1_frec(hipStream_t s, int block, const double *qx, const double *qy, const double *qz, long Q, int qblocks,
                      const double *bound, const double *px, const double *py, const double *pz, int ntiles, int nparts,
                      const Xf *H, double rmax, uint32_t *hit_cnt, uint32_t *hit_list, uint32_t cap);
void launch_knn1_fixup(hipStream_t s, const double *qx, const double *qy, const double *qz, long Q, const double *px,
                       const double *py, const double *pz, const Xf *H, const uint32_t *hit_cnt, const uint32_t *hit_list,
                       uint32_t cap, uint32_t group, double max_d2, int64_t idx_base, double *d2_out, int64_t *idx_out,
                       double *p2_out, uint32_t *overflow);
int  frec_blocks_per_cu(int block);
void launch_bound_prev(hipStream_t s, const double *qx, const double *qy, const double *qz, const double *p2, long Q,
                       long qpad, const Xf &H, double *bound);
void launch_knn1_reduce(hipStream_t s, const double *part_d2, const uint32_t *part_idx, int nparts, int qpad, long Q,
                        double max_d2, int64_t idx_base, const double *px, const double *py, const double *pz,
                        double *d2_out, int64_t *idx_out, double *p2_out);
void launch_knnk_pass(hipStream_t s, int K, const double *qx, const double *qy, const double *qz, int qpad, long Q,
                      const double *px, const double *py, const double *pz, long npad, int chunk_pts, int nchunks,
                      const double *floor_d2_in, const uint32_t *floor_idx_in, double *part_d2, uint32_t *part_idx,
                      int kout, int col0, int kstride, int64_t idx_base, double *d2_out, int64_t *idx_out,
                      double *floor_d2_out, uint32_t *floor_idx_out);
void launch_normals(hipStream_t s, const double *px, const double *py, const double *pz, const int64_t *nn, long Q, int k,
                    int64_t idx_base, float *normals, float *planarity);
void launch_postmatch(hipStream_t s, const double *qx, const double *qy, const double *qz, const float *normals,
                      const float *planarity, const double *p2, const int64_t *idx, long Q, const Xf &H,
                      float min_planarity, const float *pl2, long pl2_n, double *dist, uint8_t *flag, const IcpDev *st = nullptr);
// st (nullable): loop state of a chained run -- H comes from it (postmatch) and every kernel exits at once when the run is over
// use_prior: out4 holds the (m, median, mad, kept) an earlier launch left for the SAME correspondences' last iteration -- both
// statistics are looked for in a window around them first (any values are safe: a window that misses falls back)
void launch_reject(hipStream_t s, const double *dist, const uint8_t *flag, long Q, uint8_t *keep, double *out4,
                   const IcpDev *st = nullptr, double *out3 = nullptr, bool use_prior = false);
void launch_stats(hipStream_t s, const double *v, const uint8_t *keep, long Q, double *out3, const double *also4 = nullptr,
                  double *host_out = nullptr, double seq = 0.0, double *partial = nullptr, unsigned *ticket = nullptr,
                  const IcpDev *st = nullptr);
int  ne_grid_for(long count);
void launch_normal_eq(hipStream_t s, const double *qx, const double *qy, const double *qz, const float *normals,
                      const double *p2, const uint8_t *keep, long lo, long hi, const double H12[12], const double dR[27],
                      double *partial, unsigned *ticket, double *out30, double *resid, double *host_out = nullptr,
                      double seq = 0.0);

}  // namespace sicp
