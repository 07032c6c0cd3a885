// sicp_host.h -- what the host-side translation units of libsimpleicp_hip share: the context, the clouds and their grids, the error
// plumbing, and the internal functions that cross files.  sicp_api.cpp: context, errors, timing, small exports; sicp_clouds.cpp: uploads,
// downloads, grid builds; sicp_search.cpp: the 1-NN / k-NN drivers and their exports; sicp_icp.cpp: the iteration loop, the solvers,
// the operator-by-operator road; sicp_comm.cpp: RCCL loader, exchanges, communicator exports.
#ifndef SICP_HOST_H
#define SICP_HOST_H

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types and prototypes only: librccl is loaded on demand (sicp_comm_init), never linked
#include <dlfcn.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/simpleicp_hip.h"
#include "sicp_internal.h"

using namespace sicp;

#define SICP_EXPORT extern "C" __attribute__((visibility("default")))

namespace sicph {

int fail(int code, const char *fmt, ...);

#define HIPCHK(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(SICP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),     \
                        __FILE__, __LINE__);                                                     \
    } while (0)

#define CHK(expr)                  \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != SICP_OK) return rc_; \
    } while (0)

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;   // elements
    int reserve(size_t n)
    {
        if (n <= cap) return SICP_OK;
        if (p) { HIPCHK(hipFree(p)); p = nullptr; cap = 0; }
        HIPCHK(hipMalloc((void **)&p, n * sizeof(T)));
        cap = n;
        return SICP_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct Grid {
    bool valid = false;
    GridGeom g;
    long ncells = 0;
    double avg_per_cell = 0;
    double target_used = 0;          // points per occupied cell the build aimed at (grid_build: rebuilt when the regime changes)
    bool cap_limited = false;        // the cell table's size limit, not the points-per-cell target, set the cell size
    double pointwise_occupancy = 0;  // sum c^2 / n over the cells: how many points share the cell of an average point
    bool nonuniform = false;         // the cell size was set by the points' own view (dense core), not by the average: wide balls cross
                                     // thousands of its rows -- such a cloud gets a coarse twin (Cloud::coarse_grid)
    DevBuf<uint32_t> cell_start;     // ncells + 1
    DevBuf<double> rec;              // the cloud in cell order: packed 32-byte records (x, y, z, local row as int64 bits)
    // companions, built the first time a search wants them (grid_companions) and dropped with the grid:
    DevBuf<float> recf;              // the cloud in cell order as 16-byte float32 records relative to c0 (the filtered many-queries search)
    DevBuf<unsigned long long> cell_box;   // the cells' tight boxes + counts (far searches trim their rows by them)
    bool recf_valid = false, box_valid = false;
    double c0[3] = {0, 0, 0}, eps_p = 0;   // float32 frame: centre of the cloud's box; 6e-8 x the largest |coordinate - c0|
    bool filter_ok = false;          // float32 can hold the cloud (half extents below 1e15)
};

struct Cloud {
    int64_t n = 0, npad = 0, idx_base = 0;
    double rmax = 0.0;    // largest point norm (error bounds of the filtered / grid searches)
    double bb_lo[3] = {0, 0, 0}, bb_hi[3] = {0, 0, 0};   // bounding box (measured with rmax in the upload's one statistics pass)
    Grid grid;
    DevBuf<float> pl;     // `planarity` column by GLOBAL index (pl_n entries; 0 = the cloud has no such column)
    int64_t pl_n = 0;
    DevBuf<double> xyz;   // x[npad] | y[npad] | z[npad]
    // every 64th point with a grid of its own (built on demand for a cold chained search): the nearest SUBSAMPLE point is a cloud
    // point, so its distance bounds the answer -- one cheap search hands the real one a radius instead of a doubling ladder
    DevBuf<double> sub_xyz; int64_t sub_n = 0, sub_npad = 0;
    Grid sub_grid;
    Grid coarse_grid;     // ALL points again in cells 8 x as wide, only for clouds whose grid is `nonuniform`: the exact search's wide passes
    const double *x() const { return xyz.p; }
    const double *y() const { return xyz.p + npad; }
    const double *z() const { return xyz.p + 2 * npad; }
    double *x() { return xyz.p; }
    double *y() { return xyz.p + npad; }
    double *z() { return xyz.p + 2 * npad; }
};

// RCCL entry points, resolved the first time a communicator is asked for (single-GPU users never load the library)
struct Rccl {
    void *h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    std::string why;               // why the library is unusable (dlerror is read ONCE, where it is fresh)
};

struct EventPair { hipEvent_t a, b; int kernel; };
constexpr int REC_RING = 16;     // records in flight + being read

inline long round_up(long v, long g) { return (v + g - 1) / g * g; }
inline double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline bool is_observed(double w) { return w > 0 && std::isfinite(w); }
inline void H16_to_Xf(const double H[16], Xf *o) { for (int i = 0; i < 12; ++i) o->m[i] = H[i]; }

}  // namespace sicph
using namespace sicph;

// an upload running behind its caller (sicp_cloud_upload_start): the helper thread and what it left
struct BgUpload {
    std::thread th;
    bool active = false;
    int rc = SICP_OK;
    std::string err;
};

struct sicp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;   // background uploads: their DMA, layout kernel and statistics pass
    BgUpload bg[2];                // by slot; at most one is active
    DevBuf<double> stage_bg;       // ... their AoS staging block
    DevBuf<double> bg_small;       // ... their statistics scratch (7 words) and its pinned mirror
    double *h_bg = nullptr;
    hipDeviceProp_t prop;
    Cloud cloud[2];
    DevBuf<double> stage;          // AoS staging for uploads / downloads / query sets
    // scan workspace
    DevBuf<double> part_d2;
    DevBuf<uint32_t> part_idx;
    DevBuf<double> kq;             // SoA queries of sicp_knn: qx|qy|qz
    DevBuf<double> k_d2;           // (Q,k) results
    DevBuf<int64_t> k_idx;
    DevBuf<double> floor_d2;
    DevBuf<uint32_t> floor_idx;
    DevBuf<double> bound;          // per-query upper bound of the NN distance (filtered scan)
    DevBuf<double> x_send, x_recv; // exchange records: [Q][5] and [world][Q][5]
    int fs_blocks_per_cu[2] = {0, 0};   // occupancy of k_knn1_fscan<128>, <256>
    int fr_blocks_per_cu[2] = {0, 0};   // occupancy of k_knn1_frec<128>, <256>
    int fscan_variant = 0;         // SICP_FSCAN = record (default: VALU filter, candidates recorded) | inline
    long fscan_cap = 0;            // SICP_FSCAN_CAP: recorded groups per query (tests force overflow with tiny values)
    DevBuf<uint32_t> hit_cnt, hit_list;
    int knn1_mode = 0;             // SICP_KNN1 = exact | filter | grid: force one 1-NN flavour (A/B + tests); 0 = auto
    DevBuf<uint32_t> g_ids, g_counts, g_cursor, g_blk;    // grid build scratch: cell ids, histogram, scatter cursors, scan partials
    DevBuf<unsigned long long> match_work;                // [0] candidates evaluated, [1] grid rows visited, [2] launches (instrumented runs); [4..7] the k-NN sweep's tallies
    DevBuf<unsigned long long> rj_keys;   // large-Q rejection scratch: Q keys + the selection state
    bool have_prev_match = false;  // m_p2 holds last iteration's winners (bound source)
    DevBuf<uint32_t> q_order;      // large query sets: the queries [q_order_lo, +q_order_cnt) in cell order (search locality)
    long q_order_lo = -1, q_order_cnt = 0;
    long order_min_q = 32768;      // SICP_ORDER_MIN_Q: from this many queries per launch on (0: never)
    DevBuf<uint32_t> k_order;      // the queries of a k-NN / normals call in cell order
    DevBuf<int64_t> k_sel;         // sicp_estimate_normals: selected rows, normals and planarity before they leave
    DevBuf<float> k_nv, k_pl;
    DevBuf<double> k_cov;          // (Q, 6) covariances between the k-NN sweep and the eigen step
    DevBuf<uint32_t> k_redo;       // [0] count, [1..] slots the four-queries-per-wave sweep left to the one-query-per-wave kernel
    int knn_group = 0;             // SICP_KNN_GROUP = 1 / 4: queries per wave of the k-NN sweep (0: chosen per launch)
    long knn_batch = 0;            // SICP_KNN_BATCH: queries a wave of the one-sweep k-NN works through (0: chosen per launch)
    bool knn_sweep = true;         // SICP_KNN_SWEEP=0: k extraction rounds (k_grid_knn) + k_normals instead of the one-sweep kernel
    DevBuf<double> bound_p2, bound_d2;   // cold search: nearest subsample point per query (coordinates = the bound) + scratch
    DevBuf<int64_t> bound_idx;
    int coarse_iters = 1;          // chained iterations (from a cold start) whose search is bounded by the subsample's (more than one helped nowhere)
    long coarse_min_n = 262144;    // ... for clouds of at least this many points
    long nn16_min_q = 5120;        // SICP_NN16_MIN_Q: from this many queries per launch on the grid search runs four queries per wave.  Round 5 measured
                                   // the crossover at 8 192 -- when the four-per-wave kernel still needed k_postmatch's launch behind it; with its own
                                   // distance epilogue (round 6, below 65 536 queries) it wins from ~5 000: match per launch at 4 096 / 6 000 / 8 191
                                   // queries 13.4 / 16.0 / 19.2 us with one wave per query, 15.3 / 14.5 / 15.6 with four (profiles/r6/nn16_crossover.txt)
    int nn16_filter = 2;           // SICP_NN16 = exact (0: k_grid_nn16) | far (1: the filtered search, one flavour) | near (2, default: the lean
                                   // flavour first, the full one for what it leaves)
    int dl_threads = 16;                  // host threads of sicp_cloud_download_both: one feeds the link, the others fan the chunks out into the caller's arrays
                                          // (round 6: with 15 of them the call waits 4.8 of its 5.3 ms for the link -- profiles/r6/download_parts_ring_packed_streamed.txt)
    int grid_cap_nonuniform_log2 = 27;    // log2 of the cell table's limit for clouds whose points crowd a few cells
    bool grid_pointwise = true;    // the cell size follows the POINT-weighted occupancy (false: the average over occupied cells only -- round 4's rule)
    long nn16f_min_q = 196608;     // SICP_NN16F_MIN_Q: from this many queries per launch on the many-queries search goes through the float32 filter
    bool nn16f_min_q_forced = false;   // (the environment named it: no per-cloud adjustment)
                                   // (below: its two extra launches cost more than the filter saves on a machine that is not full)
    double far_move = 0.75;        // SICP_FAR_MOVE: the lean flavour goes first once the estimate moves by less than this many cells per iteration
    bool upload_staged = true;     // small uploads go through the pinned buffer (false: every upload a DMA straight out of the caller's arrays)
    bool use_boxes = false;        // SICP_BOXES=1: far searches trim their rows by the cells' tight boxes.  OFF by default: measured (profiles/r5), the
                                   // boxes cut 14-30 % of the candidates and never a microsecond -- DESIGN.md section 4
    bool boxes_always = false;     // SICP_BOXES=2: ... and stand-alone searches of a handful of queries build them too (tests)
    DevBuf<double> q_slot, p_slot; // filtered search: queries (x, y, z, index) and their last matches in SLOT order (32 bytes each)
    long slot_lo = -1, slot_cnt = 0;
    bool slot_ordered = false;
    DevBuf<double> kq_slot, kp_slot;   // ... of a stand-alone search (sicp_knn k = 1, sicp_select_in_range, the operators): the run's stay untouched
    DevBuf<uint8_t> nn_state;      // by slot: 1 = the lean flavour left this query to the full one
    DevBuf<uint32_t> nn_redo;      // [0], [1] counters (alternating by launch), [2..] queries left to the exact kernel
    int nn_parity = 0;
    double last_move = 0.0;        // displacement at the cloud's edge the last completed iteration caused (far / lean flavour choice)
    int last_match_kernel = 0;     // 0 exact scan, 1 filtered scan (inline), 2 grid, 3 filtered scan (record + fix-up), 5 grid, four queries per wave (exact), 6 grid, float32 filter
    // ICP state (selected fixed points and per-iteration products)
    int64_t Q = 0, qpad = 0;
    DevBuf<double> q;              // qx|qy|qz [qpad]
    DevBuf<float> normals, planarity;
    DevBuf<int64_t> m_idx;         // matched movable index (global)
    DevBuf<double> m_d2, m_p2, dist, resid;
    DevBuf<uint8_t> flag, keep;
    DevBuf<double> small;          // [0..3] reject out, [4..6] stats out, [8..37] normal equations
    DevBuf<double> ne_partial;
    DevBuf<unsigned> ticket;
    double *h_small = nullptr;     // pinned mirror of `small`
    double *h_dl = nullptr;        // pinned buffers of sicp_cloud_download_both (a ring of 4 x 3 x 256 Ki doubles) and of the staged uploads (2 x 3 x 512 Ki), on first use
    hipEvent_t dl_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool have_iter = false;
    bool have_corr = false;        // sicp_corr_match has run: m_idx / m_p2 / dist hold its correspondences, `keep` the alive mask
    DevBuf<float> corr_pl;         // per-correspondence planarity columns handed to sicp_corr_reject_planarity: pc1 [Q] | pc2 [Q]
    double last_x[6] = {0}, last_w = 1.0, last_obs[6] = {0}, last_ow[6] = {0};
    double last_ne[30] = {0};      // normal equations at last_x (fused path caches them)
    double last_tail_cycles[5] = {0};   // k_icp_tail's own clock over its phases, last iteration (sicp_tail_cycles)
    long last_sel_rounds[2] = {0, 0}, sel_window_hits = 0;   // ... how it found median / MAD (sicp_tail_selection)
    bool have_last_ne = false;
    int solve_mode = 0;            // SICP_SOLVE = fused | host (A/B + tests); 0 = auto
    bool grid_target_forced = false;   // (kept for a caller that pins the target: every grid uses it then)
    double grid_target = 16.0;     // points per occupied grid cell the cell size aims at (
                                   // measured flat from 12 to 32, 5-20 % slower below 8: fewer, longer rows win)
    bool host_trace = false;       // SICP_SOLVE_TRACE=host: per-iteration host timings on stderr
    bool solve_trace = false;      // SICP_SOLVE_TRACE: the fused kernel's cycle counters on stderr
    bool trace_sel = false, trace_eval = false;   // SICP_SOLVE_TRACE=sel / eval: the fine splits of a trace build
    long solve_seq = 0;            // completion tickets of the fused kernel
    DevBuf<IcpDev> icp_dev;        // device-resident loop state of a chained run (sicp_tail.hip)
    DevBuf<LmDev> lm_dev;          // solver state of the multi-workgroup evaluation chain (sicp_lm.hip)
    LmDev *h_lm = nullptr;         // pinned staging of it
    DevBuf<double> resid2;         // second residual buffer of that chain (trial / accepted alternate)
    int resid_slot = 0;            // which buffer holds the last iteration's accepted residuals
    int lm_evals = 4;              // evaluations enqueued per iteration for Q > SOLVE_MAX_Q (k_lm_finish completes the rest)
    double *h_rec = nullptr;       // pinned ring of per-iteration records the tail kernel streams to the host
    IcpDev *h_state = nullptr;     // pinned staging of the loop state
    DevBuf<unsigned long long> lm_bar_buf;   // grid barrier of the one-launch minimisation (zeroed when allocated)
    unsigned long long lm_bar = 0;     // what its launches have added to the counter so far
    bool lm_one_launch = true;         // SICP_LM=launches: one launch per evaluation + finish (A/B; always with a sharded reduction)
    unsigned long long hsel_bar = 0;   // what the one-launch rejection's launches have added to its barrier counter so far
    int test_barrier_fault = 0;    // SICP_TEST_BARRIER_FAULT = 1 / 2 (tests only): the rejection's / the solver's grid barrier expects a block that never comes
    bool tail_window = true;       // SICP_TAIL_WINDOW=0: the single-workgroup tail never looks for median / MAD in a window around the last iteration's (A/B, tests)
    bool reject_prior = false;     // c->small[0..3] holds the last chained k_reject launch's statistics for the current setup
    bool hsel_window = true;       // SICP_HSEL_WINDOW=0: never the windowed (three-barrier) form of the large-Q rejection
    long hsel_run_launches = 0;    // chained rejection launches since the last setup (the window needs two of them behind it)
    bool hsel_dirty = false;
    int nn_group = 0;              // SICP_NN_GROUP=8|16: lanes per query of the many-queries search (0: chosen per launch)
    int chain_depth = 4;           // iterations enqueued ahead of the last record read (two are not enough, four are: profiles/r2/ab_chain_depth.txt)
    // exchange: an RCCL communicator of the library's own (sicp_comm_init) or a host callback (sicp_set_exchange)
    sicp_exchange_fn xfn = nullptr;
    void *xuser = nullptr;
    int last_xchg_form = 0; long xchg_count = 0;   // sicp_exchange_info: 1 records all-gather, 2 key all-reduces, 3 query slices
    bool xfn_u64 = false;          // the callback serves SICP_XCHG_MIN_U64 / MAX_U64 (asked once, at registration)
    ncclComm_t comm = nullptr;
    bool comm_active = false;      // a communicator stays with the ctx between runs (sicp_comm_activate): building one costs ~0.1-1 s
    int comm_rank = 0, comm_world = 1;
    long xkeys_min_q = 32768;      // SICP_XCHG_KEYS_MIN_Q: cloud shards from this many queries on exchange their winners by three all-reduces on
                                   // 8-byte keys instead of an all-gather of 40-byte records (0: never; the library's own communicator only)
    double xchg_timeout_s = 120.0; // a record that does not arrive within this while collectives are in flight = SICP_ERR_EXCHANGE, not a hang
    DevBuf<double> lm_gsum;        // sharded 6x6 reduction on the device solver: this rank's 8x8 Gram block, summed over ranks in place
    bool resid_sharded = false;    // ... after which only this rank's slice of the residuals is current (recomputed on demand)
    int rank = 0, world = 1, gn_shard = 0;
    int partition = SICP_PART_CLOUD;   // what is sharded over the ranks: the searched cloud or the queries
    bool collective() const { return xfn != nullptr || (comm != nullptr && comm_active); }
    // timing
    bool timing = false;
    bool count_work = false;       // sicp_timing_enable(ctx, 2): the grid search also tallies its candidates / rows
    std::vector<EventPair> pending, pool;
    double t_ms[SICP_K_COUNT] = {0};
    int64_t t_n[SICP_K_COUNT] = {0};
};

// from how many queries per launch the float32 filter takes a cloud's searches: a nonuniform cloud's alternative is one wave per query
// (the terrestrial stand-in at 100 000 queries: 0.60 ms per match against 0.53 through the filter), a uniform cloud's the four-per-wave kernel
inline long filter_min_q(const sicp_ctx *c, bool nonuniform)
{
    return (nonuniform && !c->nn16f_min_q_forced && c->nn16f_min_q > 65536) ? 65536 : c->nn16f_min_q;
}
namespace sicph {

struct Timed {
    sicp_ctx *c; EventPair ev; bool on;
    Timed(sicp_ctx *ctx, int kernel) : c(ctx), on(ctx->timing)
    {
        if (!on) return;
        if (!c->pool.empty()) { ev = c->pool.back(); c->pool.pop_back(); }
        else {
            // timing-only events: no system-scope fence (cache write-back + invalidate) at every record --
            // the default flavour cost 13 us per ICP iteration on the stream it was measuring
            (void)hipEventCreateWithFlags(&ev.a, hipEventDisableSystemFence);
            (void)hipEventCreateWithFlags(&ev.b, hipEventDisableSystemFence);
        }
        ev.kernel = kernel;
        (void)hipEventRecord(ev.a, c->stream);
    }
    ~Timed()
    {
        if (!on) return;
        (void)hipEventRecord(ev.b, c->stream);
        c->pending.push_back(ev);
    }
};


Rccl *rccl();
void euler_R(const double a[3], double R[9]);
void euler_dR(const double a[3], double dR[27]);
void params_to_H12(const double x[6], double H12[12]);
bool spd_solve(int m, double *A, double *b);
int sync(sicp_ctx *c);
void collect_ready(sicp_ctx *c);
void abandon_exchange(sicp_ctx *c);
int wait_ticket(sicp_ctx *c, const double *flag_word, double seq);
int all_gather_f64(sicp_ctx *c, double *send, double *recv, long count);
int all_reduce_sum_f64(sicp_ctx *c, double *buf, long count);
int exchange_best(sicp_ctx *c, double *d2, int64_t *idx, double *p2, long Q);
int exchange_best_chained(sicp_ctx *c, const TailArgs &A, long Q, bool packed_by_match);
bool exchange_by_keys(const sicp_ctx *c, long Q);
int exchange_best_keys_chained(sicp_ctx *c, const TailArgs &A, long Q);
int all_reduce_u64(sicp_ctx *c, unsigned long long *buf, long count, bool take_max);
long query_slice(const sicp_ctx *c, long Q, long *lo);
int exchange_query_slices_idx(sicp_ctx *c, const TailArgs &A, long Q, bool packed_by_match);
void plan_chunks(const sicp_ctx *c, long npad, long qblocks, size_t bytes_per_chunk_row, int *chunk_pts, int *nchunks);
int reject_select(sicp_ctx *c, long Q, double *host_out, double seq, const IcpDev *st);
int reset_barrier_state(sicp_ctx *c);
int barrier_timed_out(sicp_ctx *c);
int check_slot(sicp_ctx *c, int slot, bool need_data);
int check_rows(const int64_t *rows, int64_t m, int64_t n, const char *what);
double key_to_double(unsigned long long k);
int subsample_build(sicp_ctx *c, int slot);
int grid_coarse_level(sicp_ctx *c, int slot, GridLevel *lv, const GridLevel **out);
int grid_companions(sicp_ctx *c, const Cloud &cl, Grid &gr, long n, bool want_recf, bool want_box);
int points_order_build(sicp_ctx *c, const double *qx, const double *qy, const double *qz, long cnt, double h, long max_cells,
                       DevBuf<uint32_t> &order);
int query_order_build(sicp_ctx *c, long lo, long cnt, double h);
bool rigid_inverse(const Xf &H, Xf *inv);
double smax3(const Xf &H);
int knn1_device(sicp_ctx *c, int slot, const double *qsoa, long Q, long qpad, const Xf *H, double max_dist,
                const double *prev_p2, double *d2_out, int64_t *idx_out, double *p2_out);
bool knnk_uses_grid(const sicp_ctx *c, const Cloud &cl, long Q);
int normal_eq_host(sicp_ctx *c, const double x[6], bool write_resid, bool allow_shard, double out[30]);
double objective(const double ne[30], double w, const double x[6], const double obs[6], const double ow[6]);
int grid_build_arrays(sicp_ctx *c, const Cloud &cl, const double *X, const double *Y, const double *Z, long n, Grid &gr, double target = 0.0,
                      double h_forced = 0.0);
int grid_build(sicp_ctx *c, int slot, long icp_queries = -1);
int knnk_device(sicp_ctx *c, int slot, const double *qsoa, long Q, long qpad, int k, double *d2_out, int64_t *idx_out,
                float *normals_out = nullptr, float *planarity_out = nullptr, bool *fused = nullptr);
int cloud_stats(sicp_ctx *c, int slot);
int upload_join(sicp_ctx *c, int slot);

}  // namespace sicph

#endif
