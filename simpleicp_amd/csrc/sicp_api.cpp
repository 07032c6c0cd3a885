// sicp_api.cpp -- host side of libsimpleicp_hip.so: C ABI (include/simpleicp_hip.h), device
// memory, kernel sequencing of one ICP iteration, and the host 6x6 Levenberg-Marquardt solve
// that the fused GPU reductions feed.  No CPU compute fallback exists: without a gfx950 device
// sicp_ctx_create fails with SICP_ERR_NO_DEVICE.
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

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

}  // namespace

// error sink for the host-only translation units (sicp_io.cpp)
int sicp_io_fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

namespace {

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
Rccl &rccl_state() { static Rccl R; return R; }
Rccl *rccl()
{
    Rccl &R = rccl_state();
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            R.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (R.h) break;
            const char *e = dlerror();
            R.why += std::string(R.why.empty() ? "" : "; ") + (e ? e : "dlopen failed");
        }
        if (R.h) {
            R.GetUniqueId = (decltype(R.GetUniqueId))dlsym(R.h, "ncclGetUniqueId");
            R.CommInitRank = (decltype(R.CommInitRank))dlsym(R.h, "ncclCommInitRank");
            R.CommDestroy = (decltype(R.CommDestroy))dlsym(R.h, "ncclCommDestroy");
            R.AllGather = (decltype(R.AllGather))dlsym(R.h, "ncclAllGather");
            R.AllReduce = (decltype(R.AllReduce))dlsym(R.h, "ncclAllReduce");
            R.GetErrorString = (decltype(R.GetErrorString))dlsym(R.h, "ncclGetErrorString");
            R.CommCount = (decltype(R.CommCount))dlsym(R.h, "ncclCommCount");
            R.CommUserRank = (decltype(R.CommUserRank))dlsym(R.h, "ncclCommUserRank");
            R.CommAbort = (decltype(R.CommAbort))dlsym(R.h, "ncclCommAbort");
            if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.AllGather || !R.AllReduce || !R.GetErrorString ||
                !R.CommCount || !R.CommUserRank || !R.CommAbort) { R.h = nullptr; R.why = "librccl lacks an entry point this library needs"; }
        }
    }
    return R.h ? &R : nullptr;
}

struct EventPair { hipEvent_t a, b; int kernel; };
constexpr int REC_RING = 16;     // records in flight + being read

long round_up(long v, long g) { return (v + g - 1) / g * g; }

// mathutils.py:39-68 : R = Rx(a1) * Ry(a2) * Rz(a3) written out
void euler_R(const double a[3], double R[9])
{
    const double c1 = std::cos(a[0]), s1 = std::sin(a[0]);
    const double c2 = std::cos(a[1]), s2 = std::sin(a[1]);
    const double c3 = std::cos(a[2]), s3 = std::sin(a[2]);
    R[0] = c2 * c3;                 R[1] = -c2 * s3;                R[2] = s2;
    R[3] = c1 * s3 + s1 * s2 * c3;  R[4] = c1 * c3 - s1 * s2 * s3;  R[5] = -s1 * c2;
    R[6] = s1 * s3 - c1 * s2 * c3;  R[7] = s1 * c3 + c1 * s2 * s3;  R[8] = c1 * c2;
}

// analytic partial derivatives of R w.r.t. the three Euler angles
void euler_dR(const double a[3], double dR[27])
{
    const double c1 = std::cos(a[0]), s1 = std::sin(a[0]);
    const double c2 = std::cos(a[1]), s2 = std::sin(a[1]);
    const double c3 = std::cos(a[2]), s3 = std::sin(a[2]);
    double *A = dR, *B = dR + 9, *C = dR + 18;
    A[0] = 0; A[1] = 0; A[2] = 0;
    A[3] = -s1 * s3 + c1 * s2 * c3;  A[4] = -s1 * c3 - c1 * s2 * s3;  A[5] = -c1 * c2;
    A[6] = c1 * s3 + s1 * s2 * c3;   A[7] = c1 * c3 - s1 * s2 * s3;   A[8] = -s1 * c2;
    B[0] = -s2 * c3;       B[1] = s2 * s3;        B[2] = c2;
    B[3] = s1 * c2 * c3;   B[4] = -s1 * c2 * s3;  B[5] = s1 * s2;
    B[6] = -c1 * c2 * c3;  B[7] = c1 * c2 * s3;   B[8] = -c1 * s2;
    C[0] = -c2 * s3;                 C[1] = -c2 * c3;                  C[2] = 0;
    C[3] = c1 * c3 - s1 * s2 * s3;   C[4] = -c1 * s3 - s1 * s2 * c3;   C[5] = 0;
    C[6] = s1 * c3 + c1 * s2 * s3;   C[7] = -s1 * s3 + c1 * s2 * c3;   C[8] = 0;
}

void params_to_H12(const double x[6], double H12[12])
{
    double R[9];
    euler_R(x, R);
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) H12[4 * r + c] = R[3 * r + c];
        H12[4 * r + 3] = x[3 + r];
    }
}

// in-place Cholesky solve of an m x m SPD system (m <= 6); returns false if not SPD
bool spd_solve(int m, double *A, double *b)
{
    for (int j = 0; j < m; ++j) {
        double s = A[j * m + j];
        for (int k = 0; k < j; ++k) s -= A[j * m + k] * A[j * m + k];
        if (!(s > 0.0) || !std::isfinite(s)) return false;
        const double l = std::sqrt(s);
        A[j * m + j] = l;
        for (int i = j + 1; i < m; ++i) {
            double t = A[i * m + j];
            for (int k = 0; k < j; ++k) t -= A[i * m + k] * A[j * m + k];
            A[i * m + j] = t / l;
        }
    }
    for (int i = 0; i < m; ++i) {
        double t = b[i];
        for (int k = 0; k < i; ++k) t -= A[i * m + k] * b[k];
        b[i] = t / A[i * m + i];
    }
    for (int i = m - 1; i >= 0; --i) {
        double t = b[i];
        for (int k = i + 1; k < m; ++k) t -= A[k * m + i] * b[k];
        b[i] = t / A[i * m + i];
    }
    return true;
}

bool is_observed(double w) { return w > 0 && std::isfinite(w); }

}  // namespace

struct sicp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
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
    long nn16_min_q = 8192;        // SICP_NN16_MIN_Q: from this many queries per launch on the grid search runs four queries per wave (measured on 10 M
                                   // points: steady match 10.2 us against 35.7 at 16 384 queries, 19.9 / 69.7 at 32 768 -- one wave per query stops
                                   // being latency-bound at ~4 000 queries; its fused distance epilogue is worth a launch, ~4 us)
    int nn16_filter = 2;           // SICP_NN16 = exact (0: k_grid_nn16) | far (1: the filtered search, one flavour) | near (2, default: the lean
                                   // flavour first, the full one for what it leaves)
    bool grid_pointwise = true;    // SICP_GRID_POINTWISE=0: the cell size follows the average over occupied cells only (A/B)
    long nn16f_min_q = 196608;     // SICP_NN16F_MIN_Q: from this many queries per launch on the many-queries search goes through the float32 filter
                                   // (below: its two extra launches cost more than the filter saves on a machine that is not full)
    double far_move = 0.75;        // SICP_FAR_MOVE: the lean flavour goes first once the estimate moves by less than this many cells per iteration
    bool upload_staged = true;     // SICP_UPLOAD_STAGED=0: every upload is a DMA straight out of the caller's arrays (A/B)
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
    double *h_dl = nullptr;        // pinned double buffer of sicp_cloud_download_both (2 x 3 x 512 Ki doubles), on first use
    hipEvent_t dl_ev[2] = {nullptr, nullptr};
    bool have_iter = false;
    bool have_corr = false;        // sicp_corr_match has run: m_idx / m_p2 / dist hold its correspondences, `keep` the alive mask
    DevBuf<float> corr_pl;         // per-correspondence planarity columns handed to sicp_corr_reject_planarity: pc1 [Q] | pc2 [Q]
    double last_x[6] = {0}, last_w = 1.0, last_obs[6] = {0}, last_ow[6] = {0};
    double last_ne[30] = {0};      // normal equations at last_x (fused path caches them)
    double last_tail_cycles[5] = {0};   // k_icp_tail's own clock over its phases, last iteration (sicp_tail_cycles)
    bool have_last_ne = false;
    int solve_mode = 0;            // SICP_SOLVE = fused | host (A/B + tests); 0 = auto
    bool grid_target_forced = false;   // SICP_GRID_TARGET given: every grid uses it
    double grid_target = 16.0;     // points per occupied grid cell the cell size aims at (SICP_GRID_TARGET overrides;
                                   // measured flat from 12 to 32, 5-20 % slower below 8: fewer, longer rows win)
    bool host_trace = false;       // SICP_HOST_TRACE: per-iteration host timings on stderr
    bool solve_trace = false;      // SICP_SOLVE_TRACE: the fused kernel's cycle counters on stderr
    long solve_seq = 0;            // completion tickets of the fused kernel
    DevBuf<IcpDev> icp_dev;        // device-resident loop state of a chained run (sicp_tail.hip)
    DevBuf<LmDev> lm_dev;          // solver state of the multi-workgroup evaluation chain (sicp_lm.hip)
    LmDev *h_lm = nullptr;         // pinned staging of it
    DevBuf<double> resid2;         // second residual buffer of that chain (trial / accepted alternate)
    int resid_slot = 0;            // which buffer holds the last iteration's accepted residuals
    int lm_evals = 4;              // evaluations enqueued per iteration for Q > SOLVE_MAX_Q (SICP_LM_EVALS; k_lm_finish completes the rest)
    double *h_rec = nullptr;       // pinned ring of per-iteration records the tail kernel streams to the host
    IcpDev *h_state = nullptr;     // pinned staging of the loop state
    DevBuf<unsigned long long> lm_bar_buf;   // grid barrier of the one-launch minimisation (zeroed when allocated)
    unsigned long long lm_bar = 0;     // what its launches have added to the counter so far
    bool lm_one_launch = true;         // SICP_LM=launches: one launch per evaluation + finish (A/B; always with a sharded reduction)
    unsigned long long hsel_bar = 0;   // what the one-launch rejection's launches have added to its barrier counter so far
    int test_barrier_fault = 0;    // SICP_TEST_BARRIER_FAULT = 1 / 2 (tests only): the rejection's / the solver's grid barrier expects a block that never comes
    bool hsel_window = true;       // SICP_HSEL_WINDOW=0: never the windowed (three-barrier) form of the large-Q rejection
    long hsel_run_launches = 0;    // chained rejection launches since the last setup (the window needs two of them behind it)
    bool hsel_dirty = false;
    int nn_group = 0;              // SICP_NN_GROUP=8|16: lanes per query of the many-queries search (0: chosen per launch)
    int chain_depth = 4;           // iterations enqueued ahead of the last record read (SICP_CHAIN_DEPTH)
    // exchange: an RCCL communicator of the library's own (sicp_comm_init) or a host callback (sicp_set_exchange)
    sicp_exchange_fn xfn = nullptr;
    void *xuser = nullptr;
    ncclComm_t comm = nullptr;
    bool comm_active = false;      // a communicator stays with the ctx between runs (sicp_comm_activate): building one costs ~0.1-1 s
    int comm_rank = 0, comm_world = 1;
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

namespace {

int sync(sicp_ctx *c)
{
    HIPCHK(hipStreamSynchronize(c->stream));
    for (auto &p : c->pending) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, p.a, p.b));
        c->t_ms[p.kernel] += ms;
        c->t_n[p.kernel] += 1;
        c->pool.push_back(p);
    }
    c->pending.clear();
    return SICP_OK;
}

// non-blocking: fold in the event pairs whose kernels have finished (used on the polling path so
// that kernel timing does not add a stream synchronisation to every iteration)
void collect_ready(sicp_ctx *c)
{
    size_t w = 0;
    for (size_t i = 0; i < c->pending.size(); ++i) {
        EventPair p = c->pending[i];
        float ms = 0;
        if (hipEventQuery(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            c->t_ms[p.kernel] += ms; c->t_n[p.kernel] += 1; c->pool.push_back(p);
        } else {
            c->pending[w++] = p;
        }
    }
    c->pending.resize(w);
}

// Kernels that hand a few doubles to the host write them into pinned + mapped memory and then publish a
// sequence number there: polling that word sees the result a few microseconds before the end-of-kernel
// signal and spares a copy + hipStreamSynchronize per hand-over.  Falls back to a stream wait.
// An exchange that will never complete (a rank left the job, the ranks' collectives went out of step): give the communicator up,
// so that the collective -- and every chained kernel queued behind it on the ctx's stream -- ends instead of wedging each later
// hipStreamSynchronize (sicp_comm_destroy, sicp_ctx_destroy and sicp_icp_get_state all start with one).  ncclCommAbort makes the
// in-flight collective return; kernels behind it then run on garbage and finish.  Should the stream still not drain (a callback
// exchange: its collective is torch's, not ours to abort), the ctx moves to a fresh stream and the wedged one is left behind.
void abandon_exchange(sicp_ctx *c)
{
    if (c->comm) { (void)rccl()->CommAbort(c->comm); c->comm = nullptr; }
    c->comm_active = false;
    c->xfn = nullptr; c->xuser = nullptr;
    c->rank = 0; c->world = 1; c->gn_shard = 0;
    const auto t0 = std::chrono::steady_clock::now();
    bool drained = false;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 5.0) {
        if (hipStreamQuery(c->stream) != hipErrorNotReady) { drained = true; break; }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    (void)hipGetLastError();
    if (!drained) {
        hipStream_t fresh = nullptr;
        if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) == hipSuccess) c->stream = fresh;    // (the old one is leaked on purpose)
        c->pending.clear();                                   // their events sit on the abandoned stream
    }
    c->have_iter = false; c->have_corr = false; c->have_prev_match = false;
    c->hsel_dirty = true;                                     // whatever the interrupted launches left in the selection state
}

int wait_ticket(sicp_ctx *c, const double *flag_word, double seq)
{
    volatile const double *flag = flag_word;
    bool seen = false;
    for (long spin = 0; spin < 4000000L; ++spin) {
        if (*flag == seq) { seen = true; break; }
        __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (!seen && c->collective()) {
        // collectives are enqueued between the kernels: a rank that left the job (or a rank whose launches went out of step)
        // would leave this stream waiting forever -- give up with an error instead of hanging the process
        const auto t0 = std::chrono::steady_clock::now();
        while (*flag != seq) {
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) return fail(SICP_ERR_HIP, "hipStreamQuery: %s", hipGetErrorString(q));
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->xchg_timeout_s) {
                const int rank = c->rank, world = c->world;
                abandon_exchange(c);
                return fail(SICP_ERR_EXCHANGE, "no result after %.0f s with a multi-GPU exchange in flight (rank %d of %d): a rank left "
                                               "the job or the ranks' collectives are out of step (SICP_XCHG_TIMEOUT_S); the "
                                               "communicator was aborted, the context is single-GPU again",
                            c->xchg_timeout_s, rank, world);
            }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        seen = *flag == seq;
    }
    if (!seen) return sync(c);
    if (c->timing) collect_ready(c);
    return SICP_OK;
}

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

// job-wide winner per query: pack (d2, idx, xyz) records, all-gather through the host's callback
// (torch.distributed over RCCL), reduce lexicographically on the device -- one collective per call
// recv[world][count] <- every rank's send[count], enqueued in order on the library's stream
int all_gather_f64(sicp_ctx *c, double *send, double *recv, long count)
{
    if (c->comm && c->comm_active) {
        const ncclResult_t r = rccl()->AllGather(send, recv, (size_t)count, ncclDouble, c->comm, c->stream);
        if (r != ncclSuccess) return fail(SICP_ERR_EXCHANGE, "ncclAllGather failed: %s", rccl()->GetErrorString(r));
        return SICP_OK;
    }
    // no host wait: the callback enqueues the collective in order on this stream (or synchronises itself)
    if (c->xfn(c->xuser, SICP_XCHG_ALLGATHER_F64, send, recv, nullptr, count) != 0)
        return fail(SICP_ERR_EXCHANGE, "exchange callback (ALLGATHER_F64) failed");
    return SICP_OK;
}
int all_reduce_sum_f64(sicp_ctx *c, double *buf, long count)
{
    if (c->comm && c->comm_active) {
        const ncclResult_t r = rccl()->AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, c->comm, c->stream);
        if (r != ncclSuccess) return fail(SICP_ERR_EXCHANGE, "ncclAllReduce failed: %s", rccl()->GetErrorString(r));
        return SICP_OK;
    }
    if (c->xfn(c->xuser, SICP_XCHG_SUM_F64, buf, nullptr, nullptr, count) != 0)
        return fail(SICP_ERR_EXCHANGE, "exchange callback (SUM_F64) failed");
    return SICP_OK;
}

// cloud shards: job-wide winner per query = lexicographic minimum over the ranks' local winners
int exchange_best(sicp_ctx *c, double *d2, int64_t *idx, double *p2, long Q)
{
    if (!c->collective() || c->partition != SICP_PART_CLOUD) return SICP_OK;
    Timed t(c, SICP_K_XCHG);
    CHK(c->x_send.reserve((size_t)5 * Q));
    CHK(c->x_recv.reserve((size_t)5 * Q * c->world));
    launch_pack_best(c->stream, d2, idx, p2, Q, c->x_send.p);
    HIPCHK(hipGetLastError());
    CHK(all_gather_f64(c, c->x_send.p, c->x_recv.p, 5 * Q));
    launch_lexmin_gathered(c->stream, c->x_recv.p, c->world, Q, d2, idx, p2);
    HIPCHK(hipGetLastError());
    return SICP_OK;
}

// The same behind the match of a chained ICP iteration, in two launches fewer: the match kernel's winning lanes left the packed
// records themselves (PostMatch::pack), and ONE kernel takes the lexicographic minimum over the ranks and forms the
// point-to-plane distance + planarity verdict (k_postmatch's work) from it.
int exchange_best_chained(sicp_ctx *c, const TailArgs &A, long Q, bool packed_by_match)
{
    Timed t(c, SICP_K_XCHG);
    CHK(c->x_recv.reserve((size_t)5 * Q * c->world));
    if (!packed_by_match) {
        launch_pack_best(c->stream, c->m_d2.p, c->m_idx.p, c->m_p2.p, Q, c->x_send.p);
        HIPCHK(hipGetLastError());
    }
    CHK(all_gather_f64(c, c->x_send.p, c->x_recv.p, 5 * Q));
    launch_lexmin_postmatch(c->stream, c->x_recv.p, c->world, Q, c->q.p, c->q.p + c->qpad, c->q.p + 2 * c->qpad, c->normals.p,
                            c->planarity.p, A.min_planarity, A.pl2, A.pl2_n, c->icp_dev.p, c->m_d2.p, c->m_idx.p, c->m_p2.p,
                            c->dist.p, c->flag.p);
    HIPCHK(hipGetLastError());
    return SICP_OK;
}

// query shards (cloud replicated): rank r matched queries [r * per, (r + 1) * per); the slices are gathered in rank
// order, which IS query order, so every rank ends up with all Q results
long query_slice(const sicp_ctx *c, long Q, long *lo)
{
    const long per = (Q + c->world - 1) / c->world;
    *lo = std::min<long>(Q, per * c->rank);
    return std::min<long>(Q, *lo + per) - *lo;
}
// ... gathered slim: 8 bytes per query (the matched index) instead of the 40-byte (d2, idx, xyz) record -- the cloud is replicated,
// so every rank looks the coordinates up itself and forms distance + verdict in the same pass (k_postmatch's work)
int exchange_query_slices_idx(sicp_ctx *c, const TailArgs &A, long Q, bool packed_by_match)
{
    const Cloud &cl = c->cloud[SICP_MOV];
    const long per = (Q + c->world - 1) / c->world;
    long lo; const long cnt = query_slice(c, Q, &lo);
    Timed t(c, SICP_K_XCHG);
    CHK(c->x_send.reserve((size_t)per));
    CHK(c->x_recv.reserve((size_t)per * c->world));
    if (packed_by_match) {
        // the match kernel's winning lanes wrote the slice's entries; the padding behind a short last slice reads "no match"
        if (per > cnt) HIPCHK(hipMemsetAsync(c->x_send.p + cnt, 0xff, (size_t)(per - cnt) * sizeof(double), c->stream));
    } else {
        launch_pack_idx(c->stream, c->m_idx.p + lo, cnt, per, c->x_send.p);
        HIPCHK(hipGetLastError());
    }
    CHK(all_gather_f64(c, c->x_send.p, c->x_recv.p, per));
    launch_unpack_idx_postmatch(c->stream, c->x_recv.p, Q, cl.x(), cl.y(), cl.z(), cl.idx_base, cl.n, c->q.p, c->q.p + c->qpad,
                                c->q.p + 2 * c->qpad, c->normals.p, c->planarity.p, A.min_planarity, A.pl2, A.pl2_n, c->icp_dev.p,
                                c->m_idx.p, c->m_p2.p, c->dist.p, c->flag.p);
    HIPCHK(hipGetLastError());
    return SICP_OK;
}

// how the scanned cloud is cut into chunks so the grid fills 256 CUs several times over
void plan_chunks(const sicp_ctx *c, long npad, long qblocks, size_t bytes_per_chunk_row, int *chunk_pts, int *nchunks)
{
    const long target_blocks = 8L * c->prop.multiProcessorCount;
    long want = (target_blocks + qblocks - 1) / qblocks;
    const long tiles = npad / TILE_PTS;
    if (want > tiles) want = tiles;
    const size_t budget = (size_t)2 << 30;   // partial-result workspace cap: 2 GiB
    while (want > 1 && (size_t)want * bytes_per_chunk_row > budget) want = (want + 1) / 2;
    if (want < 1) want = 1;
    long tiles_per_chunk = (tiles + want - 1) / want;
    *chunk_pts = (int)(tiles_per_chunk * TILE_PTS);
    *nchunks = (int)((tiles + tiles_per_chunk - 1) / tiles_per_chunk);
}

// median / MAD rejection + keep mask + kept statistics for Q > REJECT_MAX_Q: ONE launch with grid barriers
int reject_select(sicp_ctx *c, long Q, double *host_out, double seq, const IcpDev *st)
{
    const size_t words = (reject_select_scratch_bytes() + 7) / 8;
    if (c->rj_keys.cap < words) {
        CHK(c->rj_keys.reserve(words));
        HIPCHK(hsel_state_init(c->stream, c->rj_keys.p));                    // the one-launch form keeps its state clean from here on
        c->hsel_bar = 0;
    }
    if (c->hsel_dirty) {                                                     // interrupted launches may have left anything: start clean
        HIPCHK(hsel_state_init(c->stream, c->rj_keys.p));
        c->hsel_bar = 0; c->hsel_dirty = false;
    }
    const hipError_t e = reject_by_select_one_launch(c->stream, c->dist.p, c->flag.p, Q, c->keep.p, c->small.p, c->small.p + 4, c->rj_keys.p,
                                                     &c->hsel_bar, c->ne_partial.p, host_out, seq, st, c->test_barrier_fault == 1 ? 1u : 0u,
                                                     c->hsel_window && st != nullptr && c->hsel_run_launches >= 2);
    if (st) ++c->hsel_run_launches;
    if (e != hipSuccess) return fail(SICP_ERR_HIP, "rejection by digit selection failed: %s", hipGetErrorString(e));
    return SICP_OK;
}

// the grid barriers' state (arrival counters, generation, error word) of both one-launch kernels, as new
int reset_barrier_state(sicp_ctx *c)
{
    if (c->rj_keys.p) { HIPCHK(hsel_state_init(c->stream, c->rj_keys.p)); c->hsel_bar = 0; c->hsel_dirty = false; }
    if (c->lm_bar_buf.p) { HIPCHK(hipMemsetAsync(c->lm_bar_buf.p, 0, lm_bar_bytes(), c->stream)); c->lm_bar = 0; }
    return sync(c);
}

// a host-read rejection whose launch could not meet itself at a grid barrier (k_hsel_all reports a negative count)
int barrier_timed_out(sicp_ctx *c)
{
    (void)hipStreamSynchronize(c->stream);
    CHK(reset_barrier_state(c));
    return fail(SICP_ERR_HIP, "a device-wide barrier of the rejection timed out (blocks not co-resident: is another process using the "
                              "GPU?); the barrier state was reset");
}

int check_slot(sicp_ctx *c, int slot, bool need_data)
{
    if (!c) return fail(SICP_ERR_INVALID, "null ctx");
    if (slot != SICP_FIX && slot != SICP_MOV) return fail(SICP_ERR_INVALID, "slot must be SICP_FIX or SICP_MOV");
    if (need_data && c->cloud[slot].n <= 0) return fail(SICP_ERR_INVALID, "cloud slot %d is empty", slot);
    return SICP_OK;
}

// rows handed over the ABI index device gathers: every one must be a row of the cloud (host-or-device pointer)
int check_rows(const int64_t *rows, int64_t m, int64_t n, const char *what)
{
    hipPointerAttribute_t at;
    std::vector<int64_t> tmp;
    if (hipPointerGetAttributes(&at, rows) == hipSuccess && at.type == hipMemoryTypeDevice) {
        tmp.resize((size_t)m);
        HIPCHK(hipMemcpy(tmp.data(), rows, (size_t)m * sizeof(int64_t), hipMemcpyDeviceToHost));
        rows = tmp.data();
    } else (void)hipGetLastError();                       // plain host memory is "invalid value" to the query: not an error
    for (int64_t i = 0; i < m; ++i)
        if (rows[i] < 0 || rows[i] >= n) return fail(SICP_ERR_INVALID, "%s[%lld] = %lld is not a row of the cloud (%lld points)", what,
                                                     (long long)i, (long long)rows[i], (long long)n);
    return SICP_OK;
}

void H16_to_Xf(const double H[16], Xf *o) { for (int i = 0; i < 12; ++i) o->m[i] = H[i]; }

double key_to_double(unsigned long long k)
{
    const unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    double v; std::memcpy(&v, &b, sizeof v); return v;
}

// bins the cloud of `slot` once (own frame); see sicp_grid.hip
int grid_build_arrays(sicp_ctx *c, const Cloud &cl, const double *X, const double *Y, const double *Z, long n, Grid &gr, double target = 0.0,
                      double h_forced = 0.0);

// icp_queries: how many queries per launch the MATCH of an ICP run is about to send (its caller passes the rank's own count); -1: any
// other search (sicp_knn, sicp_select_in_range, normals, the operators) -- those take the grid as it is and never rebuild one.
int grid_build(sicp_ctx *c, int slot, long icp_queries = -1)
{
    Cloud &cl = c->cloud[slot];
    if (icp_queries < 0 && cl.grid.valid) return SICP_OK;
    // Points per occupied cell.  Large query sets pay for candidates (the machine is full: 1 M queries in 10 M points take 0.61 ms
    // per match at 16 per cell, 0.51 at 8, 0.62 at 4), the one-wave-per-query search of a few queries pays for round trips
    // and likes its rows long -- so the movable cloud of a run with many correspondences is binned finer.
    double target = c->grid_target;
    if (!c->grid_target_forced && slot == SICP_MOV && icp_queries >= c->nn16_min_q && c->nn16_min_q > 0) target = 0.5 * target;
    // a grid binned for the other regime (the same clouds first registered with 1000 correspondences, then with a million) is
    // rebuilt -- by the ICP match only: ~1 ms per 10 M points once, against 0.1 ms per iteration of a million queries
    if (cl.grid.valid && cl.grid.target_used > 0 && cl.grid.target_used != target) cl.grid.valid = false;
    cl.grid.target_used = target;
    return grid_build_arrays(c, cl, cl.x(), cl.y(), cl.z(), cl.n, cl.grid, target);
}

// the cloud's subsample (every SUB_STRIDE-th point) and its grid
constexpr long SUB_STRIDE = 64;       // (measured 64 / 16 / 8: 152 / 128 / 122 candidates per query of the cold search -- the subsample's own search pays the difference back)
int subsample_build(sicp_ctx *c, int slot)
{
    Cloud &cl = c->cloud[slot];
    if (cl.sub_grid.valid) return SICP_OK;
    cl.sub_n = (cl.n + SUB_STRIDE - 1) / SUB_STRIDE;
    cl.sub_npad = round_up(cl.sub_n, 1024);
    CHK(cl.sub_xyz.reserve((size_t)3 * cl.sub_npad));
    launch_stride_sample(c->stream, cl.x(), cl.y(), cl.z(), cl.n, SUB_STRIDE, cl.sub_n, cl.sub_npad, cl.sub_xyz.p);
    HIPCHK(hipGetLastError());
    return grid_build_arrays(c, cl, cl.sub_xyz.p, cl.sub_xyz.p + cl.sub_npad, cl.sub_xyz.p + 2 * cl.sub_npad, cl.sub_n, cl.sub_grid);      // (points per cell of this grid: 4 / 8 / 16 measured equal)
}

// the coarse twin of a nonuniform grid (all points, cells 8 x as wide); null when the cloud needs none
int grid_coarse_level(sicp_ctx *c, int slot, GridLevel *lv, const GridLevel **out)
{
    Cloud &cl = c->cloud[slot];
    *out = nullptr;
    if (!cl.grid.valid || !cl.grid.nonuniform) return SICP_OK;
    CHK(grid_build_arrays(c, cl, cl.x(), cl.y(), cl.z(), cl.n, cl.coarse_grid, 0.0, 8.0 * cl.grid.g.h));
    lv->g = cl.coarse_grid.g; lv->cell_start = cl.coarse_grid.cell_start.p; lv->rec = cl.coarse_grid.rec.p;
    *out = lv;
    return SICP_OK;
}

// bins n points (columns X, Y, Z, inside cl's bounding box) once; see sicp_grid.hip
int grid_build_arrays(sicp_ctx *c, const Cloud &cl, const double *X, const double *Y, const double *Z, long n, Grid &gr, double target_in,
                      double h_forced)
{
    if (gr.valid) return SICP_OK;
    gr.cap_limited = false;
    gr.recf_valid = false; gr.box_valid = false;
    gr.nonuniform = false;
    double mn[3], ex[3], vol = 1.0; int deff = 0;
    for (int a = 0; a < 3; ++a) {
        mn[a] = cl.bb_lo[a];
        ex[a] = cl.bb_hi[a] - cl.bb_lo[a];
        if (!(ex[a] >= 0) || !std::isfinite(ex[a])) return fail(SICP_ERR_INVALID, "cloud has non-finite coordinates");
        if (ex[a] > 0) { vol *= ex[a]; ++deff; }
    }
    const double target = target_in > 0.0 ? target_in : c->grid_target;       // points per occupied cell
    // Dense cell array cap: 2^27 cells (512 MiB of offsets), more for clouds that are worth it -- the box of a 100 M-point SURFACE
    // is mostly empty layers, and at 2^27 cells its occupied ones held 25 points (128 candidates per 1-NN query where 10 M points
    // pay 48): six cells per point, at most 2^30 (4 GiB of offsets + as much again of build scratch, on a 288 GB device).
    long cap = 1L << 27;
    if (6 * n > cap) cap = std::min<long>(6 * n, 1L << 30);
    {
        // ... and never more than the device can spare: a cell costs 12 bytes of table + build scratch (+ 8 of tight boxes when a far
        // search asks for them) -- a quarter of what is free, at least 2^22 cells (ranks sharing a device, smaller devices)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const long fit = (long)(free_b / 4 / 20);
            cap = std::max<long>(1L << 22, std::min(cap, fit));
        }
    }
    double h = deff ? std::pow(vol * target / (double)n, 1.0 / deff) : 1.0;
    if (!(h > 0) || !std::isfinite(h)) h = 1.0;
    unsigned long long *d_cnt = (unsigned long long *)(c->small.p + 54);      // 2 u64
    // Data on a surface / curve fills far fewer cells than the volume estimate assumes.  Large clouds: measure
    // the occupancy in a small central window of the box (1/8 of every extent) at the candidate cell size and
    // correct it -- a coalesced read of the cloud per probe instead of a full trial binning with random atomics.
    bool probed = false;                             // the window probes settled on this h: no full-cloud occupancy check
    if (h_forced > 0.0) h = h_forced;                // (a coarse twin: the caller names the cell size, nothing is probed or adjusted)
    if (n >= 262144 && deff > 0 && !(h_forced > 0.0)) {
        const long every = 4;                        // a quarter of the cloud: cells of ~16 points still hold ~4 sampled ones
        double wlo[3], whi[3];
        for (int a = 0; a < 3; ++a) {
            const double mid = mn[a] + 0.5 * ex[a], half = ex[a] > 0 ? ex[a] / 16.0 : 1.0;
            wlo[a] = mid - half; whi[a] = mid + half;
        }
        double h_prev = 0, avg_prev = 0;
        for (int probe = 0; probe < 4; ++probe) {
            GridGeom W;
            long wc = 1;
            for (int a = 0; a < 3; ++a) {
                W.mn[a] = wlo[a];
                double d = std::floor((whi[a] - wlo[a]) / h) + 1.0;
                if (d > 4096.0) d = 4096.0;
                W.dim[a] = (int)d; wc *= (long)W.dim[a];
            }
            if (wc > (1L << 24)) break;                                   // window grid too fine to probe: keep h
            W.h = h; W.inv_h = 1.0 / h;
            CHK(c->g_counts.reserve((size_t)wc + 1));
            HIPCHK(hipMemsetAsync(c->g_counts.p, 0, ((size_t)wc + 1) * sizeof(uint32_t), c->stream));
            HIPCHK(hipMemsetAsync(d_cnt, 0, 2 * sizeof(unsigned long long), c->stream));
            launch_window_probe(c->stream, X, Y, Z, n, every, W, whi, c->g_counts.p, d_cnt);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(c->h_small + 54, d_cnt, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            CHK(sync(c));
            unsigned long long res[2]; std::memcpy(res, c->h_small + 54, sizeof res);
            if (res[0] < 4096 || res[1] == 0) break;                      // (nearly) empty window: no evidence, keep h
            // points per occupied cell of the FULL cloud: the sample misses a cell of k points with probability
            // ~exp(-k / every) -- negligible around the target
            const double avg = (double)res[0] * (double)every / (double)res[1];
            if (avg <= 1.5 * target && avg >= target / 1.5) { probed = true; break; }
            // occupancy ~ h^D: D from the last two probes once there are two, the embedding dimension before
            double D = deff;
            if (h_prev > 0 && avg_prev > 0 && avg != avg_prev) {
                D = std::log(avg / avg_prev) / std::log(h / h_prev);
                if (!(D > 0.5)) D = 0.5;
                if (D > 3.0) D = 3.0;
            }
            h_prev = h; avg_prev = avg;
            h *= std::pow(target / avg, 1.0 / D);
        }
    }
    CHK(c->g_ids.reserve(n));
    GridGeom G;
    long ncells = 1;
    for (int attempt = 0;; ++attempt) {
        bool capped = false;
        for (;;) {
            ncells = 1;
            for (int a = 0; a < 3; ++a) {
                double d = std::floor(ex[a] / h) + 1.0;
                if (d > 2.0e9) d = 2.0e9;
                G.dim[a] = (int)d; ncells *= (long)G.dim[a];
                if (ncells > (1L << 40)) ncells = 1L << 40;
            }
            if (ncells <= cap) break;
            h *= std::cbrt((double)ncells / (double)cap) * 1.02;
            capped = true;
        }
        gr.cap_limited = capped;                         // cells are coarser than the target asked for
        for (int a = 0; a < 3; ++a) G.mn[a] = mn[a];
        G.h = h; G.inv_h = 1.0 / h;
        CHK(c->g_counts.reserve((size_t)ncells + 1));
        CHK(c->g_blk.reserve((size_t)grid_scan_blocks(ncells) + 1));
        HIPCHK(hipMemsetAsync(c->g_counts.p, 0, ((size_t)ncells + 1) * sizeof(uint32_t), c->stream));
        HIPCHK(hipMemsetAsync(d_cnt, 0, 2 * sizeof(unsigned long long), c->stream));
        launch_cell_ids(c->stream, X, Y, Z, n, G, c->g_ids.p, c->g_counts.p, probed ? nullptr : d_cnt);
        // first step of the offsets' scan; it also leaves sum c^2 over the cells: sum c^2 / n = the occupancy of the cell an average
        // POINT lives in.  On a scan whose density falls like 1 / r^2 that is thousands where the average over occupied cells says 16
        // -- and it is what a query, itself a point of such a cloud, pays for.  The cell size follows it (down to the table's limit).
        launch_grid_scan_sums(c->stream, c->g_counts.p, ncells, c->g_blk.p, d_cnt + 1);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(c->h_small + 54, d_cnt, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        CHK(sync(c));
        unsigned long long res2[2]; std::memcpy(res2, c->h_small + 54, sizeof res2);
        const double pw = (double)res2[1] / (double)std::max<long>(n, 1);
        gr.avg_per_cell = probed ? target : (double)n / (double)std::max<unsigned long long>(res2[0], 1);
        // still far too coarse (small clouds are not probed; windows can mislead): shrink and bin again
        if (h_forced > 0.0) { gr.pointwise_occupancy = pw; break; }
        if (!probed && gr.avg_per_cell > 3 * target && attempt < 4 && ncells < cap / 2) { h *= std::sqrt(target / gr.avg_per_cell); continue; }
        // the points' own view: shrink until an average point shares its cell with a few times the target (occupancy ~ h^2 on a surface);
        // not below the table's limit (a binning that was capped stands), at most six rounds
        if (c->grid_pointwise && pw > 4.0 * target && !capped && attempt < 6) {
            const double f = std::sqrt(2.0 * target / pw);
            h *= f < 0.3 ? 0.3 : (f > 0.8 ? 0.8 : f);
            probed = false;                              // (the window's evidence is overruled: measure the plain occupancy too from here on)
            gr.nonuniform = true;
            continue;
        }
        gr.pointwise_occupancy = pw;
        break;
    }
    gr.g = G; gr.ncells = ncells;
    // offsets = exclusive scan of the histogram (entry ncells = n), then the counting-sort scatter writes every
    // point once, as a packed record, into its cell's range
    CHK(gr.cell_start.reserve((size_t)ncells + 1));
    CHK(c->g_cursor.reserve((size_t)ncells + 1));
    CHK(gr.rec.reserve((size_t)4 * n));
    launch_grid_scan_rest(c->stream, c->g_counts.p, ncells, c->g_blk.p, gr.cell_start.p, c->g_cursor.p);
    launch_scatter(c->stream, X, Y, Z, c->g_ids.p, n, c->g_cursor.p, gr.rec.p);
    HIPCHK(hipGetLastError());
    CHK(sync(c));
    gr.valid = true;
    return SICP_OK;
}

// The companions of a grid that exists: float32 records for the filtered search, tight boxes for far searches.  Each is one pass
// (0.08 ms / 0.2 ms per 10 M points) paid by the first search that wants it.
int grid_companions(sicp_ctx *c, const Cloud &cl, Grid &gr, long n, bool want_recf, bool want_box)
{
    if (!gr.valid) return fail(SICP_ERR_INVALID, "internal: grid companions before the grid");
    if (want_recf && !gr.recf_valid) {
        double half = 0.0;
        for (int a = 0; a < 3; ++a) {
            gr.c0[a] = 0.5 * (cl.bb_lo[a] + cl.bb_hi[a]);
            half = std::max(half, std::max(cl.bb_hi[a] - gr.c0[a], gr.c0[a] - cl.bb_lo[a]));
        }
        gr.filter_ok = std::isfinite(half) && half < 1.0e15;
        gr.eps_p = 6.0e-8 * half;
        if (gr.filter_ok) {
            CHK(gr.recf.reserve((size_t)4 * n));
            launch_recf(c->stream, gr.rec.p, n, gr.c0, gr.recf.p);
            HIPCHK(hipGetLastError());
        }
        gr.recf_valid = true;
    }
    if (want_box && !gr.box_valid) {
        CHK(gr.cell_box.reserve((size_t)gr.ncells));
        launch_cell_boxes(c->stream, gr.cell_start.p, gr.rec.p, gr.ncells, gr.g, gr.cell_box.p);
        HIPCHK(hipGetLastError());
        gr.box_valid = true;
    }
    return SICP_OK;
}

// Permutation of `cnt` points (columns qx, qy, qz) by cell of a grid over their own bounding box (cell size h, grown until the
// table has at most max_cells cells): order[slot] = point.  Only the ORDER in which waves take the queries changes -- every
// result still lands at the query's own index -- so neighbouring waves walk the same rows of the searched cloud's grid.
int points_order_build(sicp_ctx *c, const double *qx, const double *qy, const double *qz, long cnt, double h, long max_cells,
                       DevBuf<uint32_t> &order)
{
    unsigned long long *d_st = (unsigned long long *)(c->small.p + 40);       // 7 u64
    unsigned long long h_init[7] = {~0ull, ~0ull, ~0ull, 0ull, 0ull, 0ull, 0ull};
    HIPCHK(hipMemcpyAsync(d_st, h_init, sizeof h_init, hipMemcpyHostToDevice, c->stream));
    launch_cloud_stats(c->stream, qx, qy, qz, cnt, d_st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->h_small + 40, d_st, 7 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    CHK(sync(c));
    unsigned long long hk[7]; std::memcpy(hk, c->h_small + 40, sizeof hk);
    GridGeom G;
    double ex[3];
    for (int a = 0; a < 3; ++a) { G.mn[a] = key_to_double(hk[a]); ex[a] = key_to_double(hk[3 + a]) - G.mn[a]; }
    if (!(h > 0) || !std::isfinite(h)) h = 1.0;
    long ncells = 1;
    for (;;) {
        ncells = 1;
        for (int a = 0; a < 3; ++a) {
            double d = std::floor(ex[a] / h) + 1.0;
            if (!(d >= 1.0)) d = 1.0;
            if (d > 1.0e6) d = 1.0e6;
            G.dim[a] = (int)d;
            if (a < 2) G.dim[a] = (G.dim[a] + 7) & ~7;        // cells are numbered in 8 x 8 (x, y) tiles: k_cell_ids_tiled
            ncells *= (long)G.dim[a];
            if (ncells > (1L << 40)) ncells = 1L << 40;
        }
        if (ncells <= max_cells) break;
        h *= 1.3;
    }
    G.h = h; G.inv_h = 1.0 / h;
    CHK(c->g_ids.reserve(cnt));
    CHK(c->g_counts.reserve((size_t)ncells + 1));
    CHK(c->g_cursor.reserve((size_t)ncells + 1));
    CHK(c->g_blk.reserve((size_t)grid_scan_blocks(ncells) + 1));
    CHK(order.reserve(cnt));
    HIPCHK(hipMemsetAsync(c->g_counts.p, 0, ((size_t)ncells + 1) * sizeof(uint32_t), c->stream));
    launch_cell_ids_tiled(c->stream, qx, qy, qz, cnt, G, c->g_ids.p, c->g_counts.p);
    launch_grid_scan(c->stream, c->g_counts.p, ncells, c->g_blk.p, nullptr, c->g_cursor.p);
    launch_scatter_order(c->stream, c->g_ids.p, cnt, c->g_cursor.p, order.p);
    HIPCHK(hipGetLastError());
    return SICP_OK;
}

// the ICP queries [lo, lo + cnt) in cell order (cell size: the cloud grid's, the two frames differ by a near-rigid H), once per setup
int query_order_build(sicp_ctx *c, long lo, long cnt, double h)
{
    if (c->q_order_lo == lo && c->q_order_cnt == cnt) return SICP_OK;
    CHK(points_order_build(c, c->q.p + lo, c->q.p + c->qpad + lo, c->q.p + 2 * c->qpad + lo, cnt, h, 1L << 25, c->q_order));
    c->q_order_lo = lo; c->q_order_cnt = cnt;
    return SICP_OK;
}

// H (rows 0..2) rigid to working precision?  Then Hinv = [R^T | -R^T t].
bool rigid_inverse(const Xf &H, Xf *inv)
{
    double e = 0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        double s = 0; for (int k = 0; k < 3; ++k) s += H.m[4 * k + i] * H.m[4 * k + j];
        e = std::max(e, std::fabs(s - (i == j ? 1.0 : 0.0)));
    }
    if (!(e < 1e-13)) return false;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) inv->m[4 * i + j] = H.m[4 * j + i];
        inv->m[4 * i + 3] = -(H.m[i] * H.m[3] + H.m[4 + i] * H.m[7] + H.m[8 + i] * H.m[11]);
    }
    return true;
}

// largest singular value of the 3x3 part of H (so |Hp| <= smax*|p| + |t| for ANY affine H)
double smax3(const Xf &H)
{
    double A[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        A[i][j] = 0; for (int k = 0; k < 3; ++k) A[i][j] += H.m[4 * k + i] * H.m[4 * k + j];   // A = M^T M
    }
    for (int sweep = 0; sweep < 32; ++sweep) {
        const double off = std::fabs(A[0][1]) + std::fabs(A[0][2]) + std::fabs(A[1][2]);
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) {
            if (A[p][q] == 0.0) continue;
            const double th = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
            const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
            const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c, apq = A[p][q];
            A[p][p] -= t * apq; A[q][q] += t * apq; A[p][q] = A[q][p] = 0.0;
            const int r = 3 - p - q;
            const double arp = A[r][p], arq = A[r][q];
            A[r][p] = A[p][r] = c * arp - sn * arq;
            A[r][q] = A[q][r] = sn * arp + c * arq;
        }
    }
    const double l = std::max(A[0][0], std::max(A[1][1], A[2][2]));
    return std::sqrt(std::max(l, 0.0)) * (1.0 + 1e-9);
}

// 1-NN of SoA queries (qx|qy|qz with stride qpad) in a slot; results in device buffers.
//   prev_p2 : optional (Q,3) coordinates of a cloud point per query (last iteration's match): its
//             exact distance under H is the filter bound; otherwise a strided-subsample exact
//             pre-pass provides one.  Either way the answer equals the plain brute-force scan's.
int knn1_device(sicp_ctx *c, int slot, const double *qsoa, long Q, long qpad, const Xf *H, double max_dist,
                const double *prev_p2, double *d2_out, int64_t *idx_out, double *p2_out)
{
    Cloud &cl = c->cloud[slot];
    const double max_d2 = max_dist * max_dist;
    const long tiles = cl.npad / TILE_PTS;
    const int cus = c->prop.multiProcessorCount;
    double rmax_t = cl.rmax;
    if (H) rmax_t = smax3(*H) * cl.rmax + std::sqrt(H->m[3] * H->m[3] + H->m[7] * H->m[7] + H->m[11] * H->m[11]);
    rmax_t *= (1.0 + 1e-9);
    // ---- pruned exact search on the static grid (rigid H only) ----
    Xf Hinv;
    const bool rigid = !H || rigid_inverse(*H, &Hinv);
    // the grid build hands 32-bit item counts to the device sort/scan primitives
    const bool big = (cl.n > 65536 || (double)cl.n * (double)Q > 1.0e9) && cl.n < (1LL << 31);
    if (rigid && (c->knn1_mode == 3 || (c->knn1_mode == 0 && big))) {
        CHK(grid_build(c, slot));
        Grid &gr = cl.grid;
        GridLevel coarse_lv; const GridLevel *coarse = nullptr;
        CHK(grid_coarse_level(c, slot, &coarse_lv, &coarse));
        // (a nonuniform cloud: one wave per query -- 64 rows per batch and the coarse grid for wide passes -- until the filtered search takes over)
        const bool four = Q >= c->nn16_min_q && !gr.nonuniform;
        c->last_match_kernel = four ? 5 : 2;
        // large query sets: through the float32 filter (sicp_gridf.hip), what it leaves (ties within its margin) through the exact
        // kernel -- the same answers
        if (Q >= c->nn16_min_q && Q >= c->nn16f_min_q && c->nn16_filter != 0 && Q < (1L << 31)) {
            CHK(grid_companions(c, cl, gr, cl.n, true, c->use_boxes));
            if (gr.filter_ok) {
                CHK(c->kq_slot.reserve((size_t)4 * Q)); CHK(c->kp_slot.reserve((size_t)4 * Q));
                CHK(c->nn_state.reserve((size_t)Q));
                if (c->nn_redo.cap < (size_t)Q + 2) {
                    CHK(c->nn_redo.reserve((size_t)Q + 2));
                    HIPCHK(hipMemsetAsync(c->nn_redo.p, 0, 2 * sizeof(uint32_t), c->stream));
                }
                launch_slot_queries(c->stream, qsoa, qsoa + qpad, qsoa + 2 * qpad, nullptr, prev_p2, Q, c->kq_slot.p, c->kp_slot.p);
                const unsigned long long *cbox = c->use_boxes ? gr.cell_box.p : nullptr;
                unsigned *tie_cnt = c->nn_redo.p + c->nn_parity, *tie_clear = c->nn_redo.p + (c->nn_parity ^ 1);
                uint32_t *tie_list = c->nn_redo.p + 2;
                unsigned long long *wk = c->count_work ? c->match_work.p : nullptr;
                c->last_match_kernel = 6;
                Timed t(c, SICP_K_KNN1);
                const bool all_far = c->nn16_filter == 1;
                if (!all_far)
                    launch_grid_nn16f(c->stream, 16, false, nullptr, c->kq_slot.p, c->kp_slot.p, Q, gr.g, gr.c0, gr.eps_p, gr.cell_start.p,
                                      gr.recf.p, gr.rec.p, false, H, H ? &Hinv : nullptr, cl.rmax, max_d2, cl.idx_base, d2_out,
                                      idx_out, p2_out, wk, 0, c->nn_state.p, tie_list, tie_cnt);
                launch_grid_nn16f(c->stream, 16, true, nullptr, c->kq_slot.p, c->kp_slot.p, Q, gr.g, gr.c0, gr.eps_p, gr.cell_start.p,
                                  gr.recf.p, gr.rec.p, false, H, H ? &Hinv : nullptr, cl.rmax, max_d2, cl.idx_base, d2_out, idx_out,
                                  p2_out, wk, 0, all_far ? nullptr : c->nn_state.p, tie_list, tie_cnt);
                launch_grid_nn_redo(c->stream, qsoa, qsoa + qpad, qsoa + 2 * qpad, Q, p2_out ? p2_out : prev_p2, gr.g, gr.cell_start.p, gr.rec.p,
                                    nullptr, H, H ? &Hinv : nullptr, cl.rmax, max_d2, cl.idx_base, d2_out, idx_out, p2_out, wk,
                                    p2_out ? NN_TIGHT : 0, nullptr, cbox, tie_list, tie_cnt, tie_clear, coarse);
                c->nn_parity ^= 1;
                HIPCHK(hipGetLastError());
                return SICP_OK;
            }
        }
        // (stand-alone searches of a few queries do not pay for the boxes of a cloud: SICP_BOXES=2 builds them anyway -- tests)
        const bool boxes = c->use_boxes && (c->boxes_always || Q >= 4096);
        if (boxes) CHK(grid_companions(c, cl, gr, cl.n, false, true));
        {
            Timed t(c, SICP_K_KNN1);
            launch_grid_nn(c->stream, qsoa, qsoa + qpad, qsoa + 2 * qpad, Q, prev_p2, gr.g, gr.cell_start.p, gr.rec.p, H,
                           H ? &Hinv : nullptr, cl.rmax, max_d2, cl.idx_base, d2_out, idx_out, p2_out,
                           c->count_work ? c->match_work.p : nullptr, four, boxes ? gr.cell_box.p : nullptr, coarse);
        }
        HIPCHK(hipGetLastError());
        return SICP_OK;
    }
    const bool small = cl.n <= 262144;                       // launch-bound anyway: one exact pass
    const bool filter_ok = std::isfinite(rmax_t) && rmax_t < 1e18;   // squares must fit float32
    if (c->knn1_mode == 1 || (small && c->knn1_mode != 2) || !filter_ok) {
        c->last_match_kernel = 0;
        const long qblocks = (Q + KNN_BLOCK * KNN1_R - 1) / (KNN_BLOCK * KNN1_R);
        long want = std::max<long>(1, (8L * cus + qblocks - 1) / qblocks);
        want = std::min(want, tiles);
        const int tpc = (int)((tiles + want - 1) / want);
        const int nchunks = (int)((tiles + tpc - 1) / tpc);
        CHK(c->part_d2.reserve((size_t)nchunks * qpad));
        CHK(c->part_idx.reserve((size_t)nchunks * qpad));
        {
            Timed t(c, SICP_K_KNN1);
            launch_knn1_scan(c->stream, qsoa, qsoa + qpad, qsoa + 2 * qpad, (int)qpad, (int)qblocks, cl.x(), cl.y(),
                             cl.z(), cl.npad, tpc, tpc, nchunks, H, c->part_d2.p, c->part_idx.p);
        }
        launch_knn1_reduce(c->stream, c->part_d2.p, c->part_idx.p, nchunks, (int)qpad, Q, max_d2, cl.idx_base, cl.x(),
                           cl.y(), cl.z(), d2_out, idx_out, p2_out);
        HIPCHK(hipGetLastError());
        return SICP_OK;
    }

    // ---- bound ----
    c->last_match_kernel = 1;
    CHK(c->bound.reserve(qpad));
    if (prev_p2) {
        launch_bound_prev(c->stream, qsoa, qsoa + qpad, qsoa + 2 * qpad, prev_p2, Q, qpad, *H, c->bound.p);
    } else {
        const int step = 64;                                  // every 64th 1024-point tile: 1.6 % of the cloud
        const long qblocks = (Q + KNN_BLOCK * KNN1_R - 1) / (KNN_BLOCK * KNN1_R);
        const int nsub = (int)((tiles + step - 1) / step);
        CHK(c->part_d2.reserve((size_t)nsub * qpad));
        CHK(c->part_idx.reserve((size_t)nsub * qpad));
        launch_knn1_scan(c->stream, qsoa, qsoa + qpad, qsoa + 2 * qpad, (int)qpad, (int)qblocks, cl.x(), cl.y(), cl.z(),
                         cl.npad, step, 1, nsub, H, c->part_d2.p, c->part_idx.p);
        launch_knn1_reduce(c->stream, c->part_d2.p, c->part_idx.p, nsub, (int)qpad, Q,
                           std::numeric_limits<double>::infinity(), 0, cl.x(), cl.y(), cl.z(), c->bound.p, nullptr, nullptr);
    }
    // ---- filtered scan: fill the chip exactly once with resident blocks ----
    const int blk = (Q > 1024) ? 256 : 128;                  // 8 queries per lane either way
    const long qblocks = (Q + blk * FS_R - 1) / (blk * FS_R);
    const int ftiles = (int)(cl.npad / FS_TILE);
    // (a) record + fix-up: the streaming kernel carries no FP64 state; exact work in a second, tiny kernel.  (The filter runs on the
    //     vector ALU: the FP32 matrix-pipe form measured slower -- profiles/r2/README.md -- and was removed in round 4.)
    if (c->fscan_variant != 1) {
        int &bpr = c->fr_blocks_per_cu[blk == 256];
        if (bpr == 0) bpr = frec_blocks_per_cu(blk);
        long nparts = std::max<long>(1, ((long)cus * bpr) / qblocks);
        nparts = std::min<long>(nparts, ftiles);
        uint32_t cap = c->fscan_cap > 0 ? (uint32_t)c->fscan_cap
                                        : (uint32_t)std::max<long>(32, std::min<long>(4096, (256L << 20) / qpad));
        CHK(c->hit_cnt.reserve((size_t)qpad + 4));
        CHK(c->hit_list.reserve((size_t)qpad * cap));
        HIPCHK(hipMemsetAsync(c->hit_cnt.p, 0, ((size_t)qpad + 4) * sizeof(uint32_t), c->stream));
        uint32_t *d_over = c->hit_cnt.p + qpad;
        {
            Timed t(c, SICP_K_KNN1);
            launch_knn1_frec(c->stream, blk, qsoa, qsoa + qpad, qsoa + 2 * qpad, Q, (int)qblocks, c->bound.p, cl.x(), cl.y(),
                             cl.z(), ftiles, (int)nparts, H, rmax_t, c->hit_cnt.p, c->hit_list.p, cap);
        }
        launch_knn1_fixup(c->stream, qsoa, qsoa + qpad, qsoa + 2 * qpad, Q, cl.x(), cl.y(), cl.z(), H, c->hit_cnt.p,
                          c->hit_list.p, cap, (uint32_t)FS_G, max_d2, cl.idx_base, d2_out, idx_out, p2_out, d_over);
        HIPCHK(hipGetLastError());
        uint32_t *h_over = (uint32_t *)(c->h_small + 62);
        HIPCHK(hipMemcpyAsync(h_over, d_over, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        CHK(sync(c));
        if (*h_over == 0) { c->last_match_kernel = 3; return SICP_OK; }
        // some query's candidate list overflowed (poor bound): fall through to the self-contained kernel
    }
    // (b) self-contained variant: exact re-evaluation inside the scan (tightens its own threshold)
    int &bpc = c->fs_blocks_per_cu[blk == 256];
    if (bpc == 0) bpc = fscan_blocks_per_cu(blk);
    long nparts = std::max<long>(1, ((long)cus * bpc) / qblocks);
    nparts = std::min<long>(nparts, ftiles);
    CHK(c->part_d2.reserve((size_t)nparts * qpad));
    CHK(c->part_idx.reserve((size_t)nparts * qpad));
    {
        Timed t(c, SICP_K_KNN1);
        launch_knn1_fscan(c->stream, blk, qsoa, qsoa + qpad, qsoa + 2 * qpad, (int)qpad, Q, (int)qblocks, c->bound.p, cl.x(),
                          cl.y(), cl.z(), ftiles, (int)nparts, H, rmax_t, c->part_d2.p, c->part_idx.p);
    }
    launch_knn1_reduce(c->stream, c->part_d2.p, c->part_idx.p, (int)nparts, (int)qpad, Q, max_d2, cl.idx_base, cl.x(),
                       cl.y(), cl.z(), d2_out, idx_out, p2_out);
    HIPCHK(hipGetLastError());
    return SICP_OK;
}

// does the k-NN of Q queries in this cloud go through the grid?  (the grid build hands 32-bit item counts to its scans)
bool knnk_uses_grid(const sicp_ctx *c, const Cloud &cl, long Q)
{
    // (the brute-force k-NN keeps a sorted list per lane: 1000 queries x 44 k points x k = 40 -- the Webots pair -- took it 20.8 ms, the
    // binning of such a cloud plus the one-sweep search take well under a millisecond: only clouds of a few thousand points stay there)
    const bool big = (cl.n >= 4096 || (double)cl.n * (double)Q > 1.0e9) && cl.n < (1LL << 31);
    return c->knn1_mode == 3 || (c->knn1_mode == 0 && big);
}

// k-NN (k >= 2, or k == 1 without transform) of SoA queries; (Q,k) device outputs.  With normals_out / planarity_out the grid
// path's one-sweep kernel also forms covariance + normal + planarity of every query's neighbourhood (pointcloud.py:188-203) and
// sets *fused; d2_out / idx_out may then be null (nothing but the normals leaves the kernel).  Otherwise *fused stays false and
// the caller runs k_normals on the indices.
int knnk_device(sicp_ctx *c, int slot, const double *qsoa, long Q, long qpad, int k, double *d2_out, int64_t *idx_out,
                float *normals_out = nullptr, float *planarity_out = nullptr, bool *fused = nullptr)
{
    if (fused) *fused = false;
    Cloud &cl = c->cloud[slot];
    if (knnk_uses_grid(c, cl, Q)) {                            // pruned search on the slot's grid
        CHK(grid_build(c, slot));
        Grid &gr = cl.grid;
        if (c->knn_sweep && grid_knn_sweep_handles(k)) {       // one sweep per query (k <= 128)
            const uint32_t *order = nullptr;
            if (c->order_min_q > 0 && Q >= c->order_min_q) {
                CHK(points_order_build(c, qsoa, qsoa + qpad, qsoa + 2 * qpad, Q, 2.0 * gr.g.h, 1L << 22, c->k_order));
                order = c->k_order.p;
            }
            if (normals_out) CHK(c->k_cov.reserve((size_t)6 * Q));
            CHK(c->k_redo.reserve((size_t)Q + 1));
            Timed t(c, SICP_K_KNNK);
            launch_grid_knn_sweep(c->stream, qsoa, qsoa + qpad, qsoa + 2 * qpad, order, Q, k, gr.g, gr.avg_per_cell, gr.cell_start.p,
                                  gr.rec.p, cl.rmax, cl.idx_base, d2_out, idx_out, c->k_cov.p, normals_out, planarity_out,
                                  c->count_work ? c->match_work.p + 4 : nullptr, c->knn_batch, c->k_redo.p, c->knn_group);
            if (fused) *fused = normals_out != nullptr;
        } else {
            Timed t(c, SICP_K_KNNK);
            launch_grid_knn(c->stream, qsoa, qsoa + qpad, qsoa + 2 * qpad, Q, k, gr.g, gr.cell_start.p, gr.rec.p, cl.rmax,
                            cl.idx_base, d2_out, idx_out);
        }
        HIPCHK(hipGetLastError());
        return SICP_OK;
    }
    int done = 0;
    bool floor_valid = false;
    while (done < k) {
        const int rem = k - done;
        const int K = rem <= 8 ? 8 : rem <= 16 ? 16 : rem <= 32 ? 32 : 64;
        const int kout = rem < K ? rem : K;
        int chunk_pts, nchunks;
        plan_chunks(c, cl.npad, (Q + KNN_BLOCK - 1) / KNN_BLOCK, (size_t)qpad * K * 12, &chunk_pts, &nchunks);
        CHK(c->part_d2.reserve((size_t)nchunks * qpad * K));
        CHK(c->part_idx.reserve((size_t)nchunks * qpad * K));
        const bool more = done + kout < k;
        if (more || floor_valid) { CHK(c->floor_d2.reserve(qpad)); CHK(c->floor_idx.reserve(qpad)); }
        {
            Timed t(c, SICP_K_KNNK);
            launch_knnk_pass(c->stream, K, qsoa, qsoa + qpad, qsoa + 2 * qpad, (int)qpad, Q, cl.x(), cl.y(), cl.z(),
                             cl.npad, chunk_pts, nchunks, floor_valid ? c->floor_d2.p : nullptr,
                             floor_valid ? c->floor_idx.p : nullptr, c->part_d2.p, c->part_idx.p, kout, done, k,
                             cl.idx_base, d2_out, idx_out, more ? c->floor_d2.p : nullptr,
                             more ? c->floor_idx.p : nullptr);
        }
        HIPCHK(hipGetLastError());
        floor_valid = more;
        done += kout;
    }
    return SICP_OK;
}

// fused reduction at parameters x over [lo,hi) -> host out[30] (sums over ranks if sharded)
int normal_eq_host(sicp_ctx *c, const double x[6], bool write_resid, bool allow_shard, double out[30])
{
    double H12[12], dR[27];
    params_to_H12(x, H12);
    euler_dR(x, dR);
    long lo = 0, hi = c->Q;
    const bool shard = allow_shard && c->gn_shard && c->collective();
    if (shard) {
        const long per = (c->Q + c->world - 1) / c->world;
        lo = std::min<long>(c->Q, per * c->rank);
        hi = std::min<long>(c->Q, lo + per);
    }
    double *d_out = c->small.p + 8;
    double *h_ne = c->h_small + 128;                      // pinned: [0..29] sums, [31] ticket
    const double seq = (double)(++c->solve_seq);
    {
        Timed t(c, SICP_K_NORMALEQ);
        launch_normal_eq(c->stream, c->q.p, c->q.p + c->qpad, c->q.p + 2 * c->qpad, c->normals.p, c->m_p2.p, c->keep.p,
                         lo, hi, H12, dR, c->ne_partial.p, c->ticket.p, d_out, write_resid ? c->resid.p : nullptr,
                         shard ? nullptr : h_ne, seq);
    }
    HIPCHK(hipGetLastError());
    if (!shard) {
        CHK(wait_ticket(c, h_ne + 31, seq));
        std::memcpy(out, h_ne, 30 * sizeof(double));
        return SICP_OK;
    }
    if (shard) CHK(all_reduce_sum_f64(c, d_out, 30));
    HIPCHK(hipMemcpyAsync(c->h_small + 8, d_out, 30 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    CHK(sync(c));
    std::memcpy(out, c->h_small + 8, 30 * sizeof(double));
    return SICP_OK;
}

double objective(const double ne[30], double w, const double x[6], const double obs[6], const double ow[6])
{
    double cst = w * w * ne[28];
    for (int j = 0; j < 6; ++j)
        if (is_observed(ow[j])) { const double e = ow[j] * (x[j] - obs[j]); cst += e * e; }
    return cst;
}

}  // namespace

// ------------------------------------------------------------------------------------------
SICP_EXPORT int sicp_abi_version(void) { return SICP_ABI_VERSION; }
SICP_EXPORT const char *sicp_last_error(void) { return g_err.c_str(); }

SICP_EXPORT int sicp_device_count(int *count_out)
{
    if (!count_out) return fail(SICP_ERR_INVALID, "count_out is null");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count_out = n;
    return SICP_OK;
}

SICP_EXPORT int sicp_ctx_create(int device, sicp_ctx **ctx_out)
{
    if (!ctx_out) return fail(SICP_ERR_INVALID, "ctx_out is null");
    *ctx_out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(SICP_ERR_NO_DEVICE, "no HIP device visible: libsimpleicp_hip has no CPU path");
    if (device < 0 || device >= n) return fail(SICP_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    HIPCHK(hipSetDevice(device));
    sicp_ctx *c = new sicp_ctx();
    c->device = device;
    if (hipGetDeviceProperties(&c->prop, device) != hipSuccess) { delete c; return fail(SICP_ERR_HIP, "hipGetDeviceProperties failed"); }
    if (std::strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
        const std::string arch = c->prop.gcnArchName;
        delete c;
        return fail(SICP_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code only", device, arch.c_str());
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(SICP_ERR_HIP, "hipStreamCreate failed"); }
    if (hipHostMalloc((void **)&c->h_small, 256 * sizeof(double), hipHostMallocMapped) != hipSuccess) { delete c; return fail(SICP_ERR_HIP, "hipHostMalloc failed"); }
    int rc = c->small.reserve(128);
    if (rc == SICP_OK) rc = c->ne_partial.reserve((size_t)NE_MAX_GRID * 64);
    if (rc == SICP_OK) rc = c->ticket.reserve(4);
    if (rc == SICP_OK) rc = c->icp_dev.reserve(1);
    if (rc == SICP_OK) rc = c->lm_dev.reserve(1);
    if (rc == SICP_OK && hipHostMalloc((void **)&c->h_lm, sizeof(LmDev), hipHostMallocDefault) != hipSuccess) rc = SICP_ERR_HIP;
    if (rc == SICP_OK) rc = c->match_work.reserve(8);
    if (rc == SICP_OK && hipMemsetAsync(c->match_work.p, 0, 8 * sizeof(unsigned long long), c->stream) != hipSuccess) rc = SICP_ERR_HIP;
    if (rc == SICP_OK && hipHostMalloc((void **)&c->h_rec, (size_t)REC_RING * REC_DOUBLES * sizeof(double), hipHostMallocMapped) != hipSuccess) rc = SICP_ERR_HIP;
    if (rc == SICP_OK && hipHostMalloc((void **)&c->h_state, sizeof(IcpDev), hipHostMallocDefault) != hipSuccess) rc = SICP_ERR_HIP;
    if (rc == SICP_OK) std::memset(c->h_rec, 0, (size_t)REC_RING * REC_DOUBLES * sizeof(double));
    if (rc == SICP_OK && hipMemsetAsync(c->ticket.p, 0, 4 * sizeof(unsigned), c->stream) != hipSuccess) rc = SICP_ERR_HIP;
    if (rc != SICP_OK) { sicp_ctx_destroy(c); return rc; }
    if (const char *e = std::getenv("SICP_LM_EVALS")) { const int d = std::atoi(e); if (d >= 0 && d <= 32) c->lm_evals = d; }
    if (const char *e = std::getenv("SICP_ORDER_MIN_Q")) c->order_min_q = std::atol(e);
    if (const char *e = std::getenv("SICP_KNN_SWEEP")) c->knn_sweep = std::atoi(e) != 0;
    if (const char *e = std::getenv("SICP_KNN_BATCH")) c->knn_batch = std::atol(e);
    if (const char *e = std::getenv("SICP_KNN_GROUP")) c->knn_group = std::atoi(e);
    if (const char *e = std::getenv("SICP_NN16_MIN_Q")) c->nn16_min_q = std::atol(e);
    if (const char *e = std::getenv("SICP_NN16")) c->nn16_filter = !std::strcmp(e, "exact") ? 0 : !std::strcmp(e, "far") ? 1 : 2;
    if (const char *e = std::getenv("SICP_BOXES")) { c->use_boxes = std::atoi(e) != 0; c->boxes_always = std::atoi(e) >= 2; }
    if (const char *e = std::getenv("SICP_UPLOAD_STAGED")) c->upload_staged = std::atoi(e) != 0;
    if (const char *e = std::getenv("SICP_GRID_POINTWISE")) c->grid_pointwise = std::atoi(e) != 0;
    if (const char *e = std::getenv("SICP_NN16F_MIN_Q")) c->nn16f_min_q = std::atol(e);
    if (const char *e = std::getenv("SICP_FAR_MOVE")) { const double t = std::atof(e); if (t >= 0) c->far_move = t; }
    if (const char *e = std::getenv("SICP_COARSE_MIN_N")) c->coarse_min_n = std::atol(e);
    if (const char *e = std::getenv("SICP_LM")) c->lm_one_launch = std::strcmp(e, "launches") != 0;
    if (const char *e = std::getenv("SICP_HSEL_WINDOW")) c->hsel_window = std::atoi(e) != 0;
    if (const char *e = std::getenv("SICP_TEST_BARRIER_FAULT")) c->test_barrier_fault = std::atoi(e);
    if (const char *e = std::getenv("SICP_NN_GROUP")) { const int v = std::atoi(e); if (v == 8 || v == 16) c->nn_group = v; }
    if (const char *e = std::getenv("SICP_XCHG_TIMEOUT_S")) { const double v = std::atof(e); if (v > 0) c->xchg_timeout_s = v; }
    if (const char *e = std::getenv("SICP_CHAIN_DEPTH")) { const int d = std::atoi(e); if (d >= 1 && d < REC_RING) c->chain_depth = d; }
    if (const char *e = std::getenv("SICP_FSCAN")) c->fscan_variant = !std::strcmp(e, "inline") ? 1 : 0;
    if (const char *e = std::getenv("SICP_FSCAN_CAP")) c->fscan_cap = std::atol(e);
    if (const char *e = std::getenv("SICP_GRID_TARGET")) { const double t = std::atof(e); if (t >= 0.25 && t <= 1024) { c->grid_target = t; c->grid_target_forced = true; } }
    c->host_trace = std::getenv("SICP_HOST_TRACE") != nullptr;
    c->solve_trace = std::getenv("SICP_SOLVE_TRACE") != nullptr;
    if (const char *e = std::getenv("SICP_SOLVE")) c->solve_mode = !std::strcmp(e, "fused") ? 1 : !std::strcmp(e, "host") ? 2 : 0;
    if (const char *e = std::getenv("SICP_KNN1"))
        c->knn1_mode = !std::strcmp(e, "exact") ? 1 : !std::strcmp(e, "filter") ? 2 : !std::strcmp(e, "grid") ? 3 : 0;
    *ctx_out = c;
    return SICP_OK;
}

SICP_EXPORT int sicp_ctx_destroy(sicp_ctx *c)
{
    if (!c) return SICP_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) { (void)rccl()->CommDestroy(c->comm); c->comm = nullptr; }
    if (c->h_dl) { (void)hipHostFree(c->h_dl); c->h_dl = nullptr; }
    for (auto &e : c->dl_ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    for (auto &p : c->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto &p : c->pool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto &cl : c->cloud) { cl.xyz.release(); cl.pl.release(); cl.grid.cell_start.release(); cl.grid.rec.release();
                                 cl.sub_xyz.release(); cl.sub_grid.cell_start.release(); cl.sub_grid.rec.release(); }
    c->bound_p2.release(); c->bound_d2.release(); c->bound_idx.release(); c->q_order.release(); c->k_order.release(); c->k_sel.release(); c->k_nv.release(); c->k_pl.release(); c->k_cov.release(); c->k_redo.release(); c->g_ids.release(); c->g_counts.release(); c->g_cursor.release(); c->g_blk.release(); c->match_work.release(); c->rj_keys.release();
    c->stage.release(); c->part_d2.release(); c->part_idx.release(); c->kq.release(); c->k_d2.release();
    c->k_idx.release(); c->floor_d2.release(); c->floor_idx.release(); c->bound.release(); c->hit_cnt.release(); c->hit_list.release(); c->x_send.release(); c->x_recv.release(); c->q.release(); c->normals.release();
    c->planarity.release(); c->m_idx.release(); c->m_d2.release(); c->m_p2.release(); c->dist.release();
    c->resid.release(); c->flag.release(); c->keep.release(); c->small.release(); c->ne_partial.release();
    c->lm_bar_buf.release(); c->lm_gsum.release(); c->ticket.release(); c->icp_dev.release(); c->lm_dev.release(); c->resid2.release();
    c->corr_pl.release();
    if (c->h_lm) (void)hipHostFree(c->h_lm);
    if (c->h_small) (void)hipHostFree(c->h_small);
    if (c->h_rec) (void)hipHostFree(c->h_rec);
    if (c->h_state) (void)hipHostFree(c->h_state);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SICP_OK;
}

SICP_EXPORT int sicp_ctx_device_name(sicp_ctx *c, char *buf, int buflen)
{
    if (!c || !buf || buflen <= 0) return fail(SICP_ERR_INVALID, "bad arguments");
    std::snprintf(buf, (size_t)buflen, "%s (%s, %d CUs)", c->prop.name, c->prop.gcnArchName, c->prop.multiProcessorCount);
    return SICP_OK;
}

// ------------------------------------------------------------------------------------------
namespace {
// shared by the two upload flavours: validates, sizes the padded SoA arrays
int upload_begin(sicp_ctx *c, int slot, int64_t n, int64_t index_base)
{
    CHK(check_slot(c, slot, false));
    if (n <= 0) return fail(SICP_ERR_INVALID, "cloud must have at least one point");
    if (n >= (int64_t)0xffffffffLL) return fail(SICP_ERR_INVALID, "at most 2^32-2 points per GPU shard");
    HIPCHK(hipSetDevice(c->device));
    Cloud &cl = c->cloud[slot];
    cl.n = n; cl.npad = round_up(n, TILE_PTS); cl.idx_base = index_base;
    cl.grid.valid = false; cl.sub_grid.valid = false; cl.coarse_grid.valid = false;
    cl.pl_n = 0;                                       // a new cloud has no planarity column until one is set
    CHK(cl.xyz.reserve((size_t)3 * cl.npad));
    return SICP_OK;
}

// ... and finishes: ONE statistics pass gives the largest norm (rounding-error bounds of the filtered / grid searches;
// a non-finite cloud is refused like cKDTree would) and the bounding box the grid build starts from
int cloud_stats(sicp_ctx *c, int slot)
{
    Cloud &cl = c->cloud[slot];
    unsigned long long *d_st = (unsigned long long *)(c->small.p + 40);       // 7 u64
    unsigned long long h_init[7] = {~0ull, ~0ull, ~0ull, 0ull, 0ull, 0ull, 0ull};
    HIPCHK(hipMemcpyAsync(d_st, h_init, sizeof h_init, hipMemcpyHostToDevice, c->stream));
    launch_cloud_stats(c->stream, cl.x(), cl.y(), cl.z(), cl.n, d_st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->h_small + 40, d_st, 7 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    CHK(sync(c));
    unsigned long long hk[7]; std::memcpy(hk, c->h_small + 40, sizeof hk);
    double nn; std::memcpy(&nn, &hk[6], sizeof nn);
    if (!std::isfinite(nn)) {
        cl.n = 0;                                    // like cKDTree (pointcloud.py:161,185): no search structure over NaN / inf
        return fail(SICP_ERR_INVALID, "cloud has non-finite coordinates (NaN / inf, or |p|^2 overflows a double)");
    }
    cl.rmax = std::sqrt(nn) * (1.0 + 1e-12);
    for (int a = 0; a < 3; ++a) { cl.bb_lo[a] = key_to_double(hk[a]); cl.bb_hi[a] = key_to_double(hk[3 + a]); }
    if (slot == SICP_MOV) c->have_prev_match = false;
    return SICP_OK;
}
int upload_end(sicp_ctx *c, int slot) { return cloud_stats(c, slot); }

// Small and medium clouds go through the library's own pinned double buffer: a DMA straight out of the caller's pageable array makes
// the runtime pin that address range first, and for a range it has not seen before that costs 10-20 ms whatever the size (measured:
// Webots' two 1 MB uploads took 13-22 ms on fresh arrays, 0.2 ms on recycled addresses).  A host copy into pinned memory costs
// ~0.1 ms per MB and always the same.  Rows are transposed (or columns copied) by the host on the way, chunk ch + 1 while chunk ch
// is on the link.  Above UPLOAD_STAGED_MAX points the pinning is the smaller price.
constexpr int64_t UPLOAD_STAGED_MAX = 1 << 19;       // (one chunk: ~1.5 ms of host copy at most)
int upload_staged(sicp_ctx *c, Cloud &cl, const double *xyz, const double *x, const double *y, const double *z, int64_t n)
{
    const long CH = 1L << 19;                                 // (the download's buffers: 2 x 3 x 512 Ki doubles)
    if (!c->h_dl) HIPCHK(hipHostMalloc((void **)&c->h_dl, (size_t)2 * 3 * CH * sizeof(double), hipHostMallocDefault));
    if (!c->dl_ev[0]) { HIPCHK(hipEventCreateWithFlags(&c->dl_ev[0], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->dl_ev[1], hipEventDisableTiming)); }
    const long nchunks = (n + CH - 1) / CH;
    for (long ch = 0; ch < nchunks; ++ch) {
        const long lo = ch * CH, m = std::min<long>(CH, n - lo);
        double *b = c->h_dl + (size_t)(ch & 1) * 3 * CH;
        if (ch >= 2) HIPCHK(hipEventSynchronize(c->dl_ev[ch & 1]));       // the DMA that last read this buffer
        if (xyz) {
            const double *src = xyz + 3 * lo;
            for (long i = 0; i < m; ++i) { b[i] = src[3 * i]; b[CH + i] = src[3 * i + 1]; b[2 * CH + i] = src[3 * i + 2]; }
        } else {
            std::memcpy(b, x + lo, (size_t)m * sizeof(double));
            std::memcpy(b + CH, y + lo, (size_t)m * sizeof(double));
            std::memcpy(b + 2 * CH, z + lo, (size_t)m * sizeof(double));
        }
        HIPCHK(hipMemcpyAsync(cl.x() + lo, b, (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(cl.y() + lo, b + CH, (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(cl.z() + lo, b + 2 * CH, (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipEventRecord(c->dl_ev[ch & 1], c->stream));
    }
    launch_pad_fill(c->stream, cl.x(), cl.y(), cl.z(), n, cl.npad);
    HIPCHK(hipGetLastError());
    return SICP_OK;
}
}  // namespace

SICP_EXPORT int sicp_cloud_upload(sicp_ctx *c, int slot, const double *xyz, int64_t n, int64_t index_base)
{
    if (!xyz) return fail(SICP_ERR_INVALID, "xyz is null");
    CHK(upload_begin(c, slot, n, index_base));
    Cloud &cl = c->cloud[slot];
    if (n <= UPLOAD_STAGED_MAX && c->upload_staged) {
        CHK(upload_staged(c, cl, xyz, nullptr, nullptr, nullptr, n));
        return upload_end(c, slot);
    }
    CHK(c->stage.reserve((size_t)3 * n));
    HIPCHK(hipMemcpyAsync(c->stage.p, xyz, (size_t)3 * n * sizeof(double), hipMemcpyDefault, c->stream));
    launch_aos_to_soa(c->stream, c->stage.p, n, cl.npad, cl.x(), cl.y(), cl.z());
    HIPCHK(hipGetLastError());
    return upload_end(c, slot);
}

SICP_EXPORT int sicp_cloud_upload_columns(sicp_ctx *c, int slot, const double *x, const double *y, const double *z, int64_t n,
                                          int64_t index_base)
{
    if (!x || !y || !z) return fail(SICP_ERR_INVALID, "x / y / z is null");
    CHK(upload_begin(c, slot, n, index_base));
    Cloud &cl = c->cloud[slot];
    if (n <= UPLOAD_STAGED_MAX && c->upload_staged) {
        CHK(upload_staged(c, cl, nullptr, x, y, z, n));
        return upload_end(c, slot);
    }
    // the device layout is column-wise already: three copies straight into place, no staging, no transpose
    HIPCHK(hipMemcpyAsync(cl.x(), x, (size_t)n * sizeof(double), hipMemcpyDefault, c->stream));
    HIPCHK(hipMemcpyAsync(cl.y(), y, (size_t)n * sizeof(double), hipMemcpyDefault, c->stream));
    HIPCHK(hipMemcpyAsync(cl.z(), z, (size_t)n * sizeof(double), hipMemcpyDefault, c->stream));
    launch_pad_fill(c->stream, cl.x(), cl.y(), cl.z(), n, cl.npad);
    HIPCHK(hipGetLastError());
    return upload_end(c, slot);
}

SICP_EXPORT int sicp_cloud_size(sicp_ctx *c, int slot, int64_t *n_out)
{
    CHK(check_slot(c, slot, false));
    if (!n_out) return fail(SICP_ERR_INVALID, "n_out is null");
    *n_out = c->cloud[slot].n;
    return SICP_OK;
}

SICP_EXPORT int sicp_cloud_transform(sicp_ctx *c, int slot, const double H[16])
{
    CHK(check_slot(c, slot, true));
    if (!H) return fail(SICP_ERR_INVALID, "H is null");
    HIPCHK(hipSetDevice(c->device));
    Xf X; H16_to_Xf(H, &X);
    Cloud &cl = c->cloud[slot];
    launch_transform(c->stream, cl.x(), cl.y(), cl.z(), cl.n, X);
    HIPCHK(hipGetLastError());
    cl.grid.valid = false; cl.sub_grid.valid = false; cl.coarse_grid.valid = false;
    return cloud_stats(c, slot);                          // new bounding box / largest norm (also the synchronisation point)
}

SICP_EXPORT int sicp_cloud_set_planarity(sicp_ctx *c, int slot, const int64_t *rows, const float *planarity, int64_t m,
                                         int64_t n_global)
{
    CHK(check_slot(c, slot, true));
    Cloud &cl = c->cloud[slot];
    if (!planarity) { cl.pl_n = 0; return SICP_OK; }
    if (n_global < cl.idx_base + cl.n) return fail(SICP_ERR_INVALID, "n_global is smaller than the cloud");
    if (m < 0 || (!rows && m != n_global)) return fail(SICP_ERR_INVALID, "a dense planarity column needs n_global values");
    HIPCHK(hipSetDevice(c->device));
    CHK(cl.pl.reserve((size_t)n_global));
    if (!rows) {
        HIPCHK(hipMemcpyAsync(cl.pl.p, planarity, (size_t)m * sizeof(float), hipMemcpyDefault, c->stream));
        cl.pl_n = n_global;
        return sync(c);
    }
    CHK(check_rows(rows, m, n_global, "planarity rows"));
    DevBuf<int64_t> d_rows; DevBuf<float> d_vals;
    int rc = d_rows.reserve((size_t)std::max<int64_t>(m, 1));
    if (rc == SICP_OK) rc = d_vals.reserve((size_t)std::max<int64_t>(m, 1));
    auto body = [&]() -> int {
        HIPCHK(hipMemcpyAsync(d_rows.p, rows, (size_t)m * sizeof(int64_t), hipMemcpyDefault, c->stream));
        HIPCHK(hipMemcpyAsync(d_vals.p, planarity, (size_t)m * sizeof(float), hipMemcpyDefault, c->stream));
        launch_fill_f32(c->stream, cl.pl.p, n_global, std::numeric_limits<float>::quiet_NaN());
        launch_scatter_f32(c->stream, cl.pl.p, d_rows.p, d_vals.p, m);
        HIPCHK(hipGetLastError());
        return sync(c);
    };
    if (rc == SICP_OK) rc = body();
    (void)hipStreamSynchronize(c->stream);
    d_rows.release(); d_vals.release();
    if (rc == SICP_OK) cl.pl_n = n_global;
    return rc;
}

SICP_EXPORT int sicp_cloud_download(sicp_ctx *c, int slot, double *xyz_out)
{
    CHK(check_slot(c, slot, true));
    if (!xyz_out) return fail(SICP_ERR_INVALID, "xyz_out is null");
    HIPCHK(hipSetDevice(c->device));
    Cloud &cl = c->cloud[slot];
    CHK(c->stage.reserve((size_t)3 * cl.n));
    launch_soa_to_aos(c->stream, cl.x(), cl.y(), cl.z(), cl.n, c->stage.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(xyz_out, c->stage.p, (size_t)3 * cl.n * sizeof(double), hipMemcpyDefault, c->stream));
    return sync(c);
}

SICP_EXPORT int sicp_cloud_download_columns(sicp_ctx *c, int slot, double *x_out, double *y_out, double *z_out)
{
    CHK(check_slot(c, slot, true));
    if (!x_out || !y_out || !z_out) return fail(SICP_ERR_INVALID, "x_out / y_out / z_out is null");
    HIPCHK(hipSetDevice(c->device));
    Cloud &cl = c->cloud[slot];
    HIPCHK(hipMemcpyAsync(x_out, cl.x(), (size_t)cl.n * sizeof(double), hipMemcpyDefault, c->stream));
    HIPCHK(hipMemcpyAsync(y_out, cl.y(), (size_t)cl.n * sizeof(double), hipMemcpyDefault, c->stream));
    HIPCHK(hipMemcpyAsync(z_out, cl.z(), (size_t)cl.n * sizeof(double), hipMemcpyDefault, c->stream));
    return sync(c);
}

// The cloud as (n, 3) rows AND as three columns in ONE pass over the link (the Python mirror's transform_by_H needs both:
// run() returns the rows, the DataFrame keeps the columns -- simpleicp.py:316, pointcloud.py:205-217).  Two plain downloads into
// pageable memory cost 2 x 11-21 ms per 10 M points (the copy engine waits for the host's staging copies and page faults).
// Here the columns are pulled chunk by chunk into a pinned double buffer at link speed while host threads fan the previous chunk
// out into both destinations (the row form is a transpose the host does from the pinned chunk: nothing crosses the link twice).
SICP_EXPORT int sicp_cloud_download_both(sicp_ctx *c, int slot, double *xyz_out, double *x_out, double *y_out, double *z_out)
{
    CHK(check_slot(c, slot, true));
    if (!xyz_out && !(x_out && y_out && z_out)) return fail(SICP_ERR_INVALID, "no destination");
    if ((x_out || y_out || z_out) && !(x_out && y_out && z_out)) return fail(SICP_ERR_INVALID, "x_out / y_out / z_out: all or none");
    HIPCHK(hipSetDevice(c->device));
    Cloud &cl = c->cloud[slot];
    const long n = cl.n, CH = 1L << 19;                       // 512 Ki points = 12 MiB per chunk
    if (!c->h_dl) HIPCHK(hipHostMalloc((void **)&c->h_dl, (size_t)2 * 3 * CH * sizeof(double), hipHostMallocDefault));
    if (!c->dl_ev[0]) { HIPCHK(hipEventCreateWithFlags(&c->dl_ev[0], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->dl_ev[1], hipEventDisableTiming)); }
    const long nchunks = (n + CH - 1) / CH;
    // host threads that fan a chunk out: as many as this process may actually run on (cgroup / affinity limits, not the machine's
    // core count), at most 8, and none for clouds that are one chunk's worth of microseconds
    unsigned T = 1;
    {
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0) T = (unsigned)CPU_COUNT(&set);
        else T = std::thread::hardware_concurrency();
        T = T < 2 ? 1 : (T > 8 ? 8 : T);
        if (n < (1L << 16)) T = 1;
    }
    // waiting: a few polite spins, then sleep -- a spinner must not starve the thread it waits for in a one-CPU container
    auto wait_until = [](auto &&cond) {
        for (int spins = 0; !cond(); ++spins) {
            if (spins < 256) std::this_thread::yield();
            else std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    };
    auto enqueue = [&](long ch) -> int {
        const long lo = ch * CH, m = std::min(CH, n - lo);
        double *b = c->h_dl + (size_t)(ch & 1) * 3 * CH;
        HIPCHK(hipMemcpyAsync(b, cl.x() + lo, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(b + CH, cl.y() + lo, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(b + 2 * CH, cl.z() + lo, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipEventRecord(c->dl_ev[ch & 1], c->stream));
        return SICP_OK;
    };
    // workers: chunk `ready` is in its pinned buffer; worker t fans out its share and counts itself in `done`
    std::atomic<long> ready{-1}, done{0};
    std::atomic<bool> quit{false};
    auto work = [&](unsigned t) {
        for (long ch = 0; ch < nchunks; ++ch) {
            wait_until([&] { return ready.load(std::memory_order_acquire) >= ch || quit.load(); });
            if (quit.load()) return;
            const long lo = ch * CH, m = std::min(CH, n - lo);
            const long a = m * t / T, e = m * (t + 1) / T;
            const double *b = c->h_dl + (size_t)(ch & 1) * 3 * CH;
            if (x_out) {
                std::memcpy(x_out + lo + a, b + a, (size_t)(e - a) * sizeof(double));
                std::memcpy(y_out + lo + a, b + CH + a, (size_t)(e - a) * sizeof(double));
                std::memcpy(z_out + lo + a, b + 2 * CH + a, (size_t)(e - a) * sizeof(double));
            }
            if (xyz_out) {
                double *o = xyz_out + 3 * (lo + a);
                for (long i = a; i < e; ++i) { o[0] = b[i]; o[1] = b[CH + i]; o[2] = b[2 * CH + i]; o += 3; }
            }
            done.fetch_add(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    try {
        for (unsigned t = 1; t < T; ++t) pool.emplace_back(work, t);
    } catch (...) {
        // no thread to be had (resource limits): nothing has been copied yet -- send the ones that started home and do it alone
        // (an exception must not cross the C ABI)
        quit.store(true);
        for (auto &th : pool) th.join();
        pool.clear();
        quit.store(false);
        T = 1;
    }
    int rc = nchunks > 0 ? enqueue(0) : SICP_OK;
    for (long ch = 0; ch < nchunks && rc == SICP_OK; ++ch) {
        // the buffer chunk ch + 1 lands in was chunk ch - 1's: every worker must be through with it
        wait_until([&] { return done.load(std::memory_order_acquire) >= (long)(T - 1) * ch; });
        if (ch + 1 < nchunks) rc = enqueue(ch + 1);
        if (rc == SICP_OK && hipEventSynchronize(c->dl_ev[ch & 1]) != hipSuccess) rc = fail(SICP_ERR_HIP, "hipEventSynchronize failed");
        if (rc != SICP_OK) break;
        ready.store(ch, std::memory_order_release);
        // this thread is worker 0 of the chunk
        {
            const long lo = ch * CH, m = std::min(CH, n - lo);
            const long a = 0, e = m / T;
            const double *b = c->h_dl + (size_t)(ch & 1) * 3 * CH;
            if (x_out) {
                std::memcpy(x_out + lo + a, b + a, (size_t)(e - a) * sizeof(double));
                std::memcpy(y_out + lo + a, b + CH + a, (size_t)(e - a) * sizeof(double));
                std::memcpy(z_out + lo + a, b + 2 * CH + a, (size_t)(e - a) * sizeof(double));
            }
            if (xyz_out) {
                double *o = xyz_out + 3 * (lo + a);
                for (long i = a; i < e; ++i) { o[0] = b[i]; o[1] = b[CH + i]; o[2] = b[2 * CH + i]; o += 3; }
            }
        }
        // (the next round's wait covers the other workers; after the last chunk the joins do)
        if (ch + 1 == nchunks) wait_until([&] { return done.load(std::memory_order_acquire) >= (long)(T - 1) * nchunks; });
    }
    if (rc != SICP_OK) quit.store(true);
    for (auto &th : pool) th.join();
    if (rc != SICP_OK) { (void)hipStreamSynchronize(c->stream); return rc; }
    return sync(c);
}

// ------------------------------------------------------------------------------------------
SICP_EXPORT int sicp_knn(sicp_ctx *c, int slot, const double *q_xyz, int64_t Q, int k, const double *H, double max_dist,
                         int64_t *idx_out, double *d2_out)
{
    CHK(check_slot(c, slot, true));
    if (!q_xyz || !idx_out) return fail(SICP_ERR_INVALID, "q_xyz / idx_out is null");
    if (Q <= 0) return fail(SICP_ERR_INVALID, "Q must be > 0");
    if (k < 1) return fail(SICP_ERR_INVALID, "k must be >= 1");
    if (k > 1 && (H || std::isfinite(max_dist)))
        return fail(SICP_ERR_INVALID, "H / max_dist are only supported for k == 1");
    if (std::isnan(max_dist) || max_dist < 0) return fail(SICP_ERR_INVALID, "max_dist must be >= 0");
    HIPCHK(hipSetDevice(c->device));
    const long qpad = round_up(Q, QPAD);
    CHK(c->kq.reserve((size_t)3 * qpad));
    CHK(c->stage.reserve((size_t)3 * Q));
    CHK(c->k_d2.reserve((size_t)Q * k));
    CHK(c->k_idx.reserve((size_t)Q * k));
    HIPCHK(hipMemcpyAsync(c->stage.p, q_xyz, (size_t)3 * Q * sizeof(double), hipMemcpyDefault, c->stream));
    launch_aos_queries(c->stream, c->stage.p, Q, qpad, c->kq.p, c->kq.p + qpad, c->kq.p + 2 * qpad);
    if (k == 1) {
        Xf X;
        if (H) H16_to_Xf(H, &X);
        CHK(knn1_device(c, slot, c->kq.p, Q, qpad, H ? &X : nullptr, max_dist, nullptr, c->k_d2.p, c->k_idx.p, nullptr));
        CHK(exchange_best(c, c->k_d2.p, c->k_idx.p, nullptr, Q));
    } else {
        CHK(knnk_device(c, slot, c->kq.p, Q, qpad, k, c->k_d2.p, c->k_idx.p));
    }
    HIPCHK(hipMemcpyAsync(idx_out, c->k_idx.p, (size_t)Q * k * sizeof(int64_t), hipMemcpyDefault, c->stream));
    if (d2_out) HIPCHK(hipMemcpyAsync(d2_out, c->k_d2.p, (size_t)Q * k * sizeof(double), hipMemcpyDefault, c->stream));
    return sync(c);
}

SICP_EXPORT int sicp_select_in_range(sicp_ctx *c, int query_slot, int search_slot, const int64_t *sel_idx, int64_t Q,
                                     const double *H, double max_range, uint8_t *in_range_out)
{
    CHK(check_slot(c, query_slot, true));
    CHK(check_slot(c, search_slot, true));
    if (!in_range_out) return fail(SICP_ERR_INVALID, "in_range_out is null");
    if (query_slot == search_slot) return fail(SICP_ERR_INVALID, "query and search slot must differ");
    if (std::isnan(max_range) || max_range < 0) return fail(SICP_ERR_INVALID, "max_range must be >= 0");
    Cloud &qc = c->cloud[query_slot];
    if (!sel_idx) Q = qc.n;
    if (Q <= 0) return fail(SICP_ERR_INVALID, "Q must be > 0");
    if (qc.idx_base != 0) return fail(SICP_ERR_INVALID, "the query cloud must not be a shard");
    HIPCHK(hipSetDevice(c->device));
    const long qpad = round_up(Q, QPAD);
    CHK(c->kq.reserve((size_t)3 * qpad));
    CHK(c->k_d2.reserve((size_t)Q));
    CHK(c->k_idx.reserve((size_t)Q));
    DevBuf<int64_t> sel; DevBuf<uint8_t> mask;
    int rc = mask.reserve(Q);
    if (rc == SICP_OK && sel_idx) rc = sel.reserve(Q);
    auto body = [&]() -> int {
        if (sel_idx) {
            CHK(check_rows(sel_idx, Q, qc.n, "sel_idx"));
            HIPCHK(hipMemcpyAsync(sel.p, sel_idx, (size_t)Q * sizeof(int64_t), hipMemcpyDefault, c->stream));
        }
        launch_gather_queries(c->stream, qc.x(), qc.y(), qc.z(), sel_idx ? sel.p : nullptr, Q, qpad, c->kq.p, c->kq.p + qpad,
                              c->kq.p + 2 * qpad);
        Xf X;
        if (H) H16_to_Xf(H, &X);
        CHK(knn1_device(c, search_slot, c->kq.p, Q, qpad, H ? &X : nullptr, max_range, nullptr, c->k_d2.p, c->k_idx.p, nullptr));
        CHK(exchange_best(c, c->k_d2.p, c->k_idx.p, nullptr, Q));
        launch_found_mask(c->stream, c->k_idx.p, Q, mask.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(in_range_out, mask.p, (size_t)Q, hipMemcpyDefault, c->stream));
        return sync(c);
    };
    if (rc == SICP_OK) rc = body();
    (void)hipStreamSynchronize(c->stream);
    sel.release(); mask.release();
    return rc;
}

SICP_EXPORT int sicp_estimate_normals(sicp_ctx *c, int slot, const int64_t *sel_idx, int64_t Q, int k, float *normals_out,
                                      float *planarity_out, int64_t *nn_idx_out)
{
    CHK(check_slot(c, slot, true));
    if (!sel_idx || !normals_out || !planarity_out) return fail(SICP_ERR_INVALID, "null argument");
    if (Q <= 0) return fail(SICP_ERR_INVALID, "Q must be > 0");
    if (k < 2) return fail(SICP_ERR_INVALID, "neighbors must be >= 2");
    Cloud &cl = c->cloud[slot];
    if (k > cl.n) return fail(SICP_ERR_INVALID, "neighbors (%d) exceeds the number of points (%lld)", k, (long long)cl.n);
    CHK(check_rows(sel_idx, Q, cl.n, "sel_idx"));
    HIPCHK(hipSetDevice(c->device));
    const long qpad = round_up(Q, QPAD);
    CHK(c->kq.reserve((size_t)3 * qpad));
    // the one-sweep kernel keeps the neighbours on chip: the (Q, k) index / distance arrays exist only when the caller wants them
    const bool sweep = c->knn_sweep && knnk_uses_grid(c, cl, Q) && grid_knn_sweep_handles(k);
    const bool want_lists = !sweep || nn_idx_out;
    if (want_lists) { CHK(c->k_d2.reserve((size_t)Q * k)); CHK(c->k_idx.reserve((size_t)Q * k)); }
    // (scratch kept with the ctx: a hipMalloc / hipFree pair per call costs more than the kernels at Q = 1000)
    DevBuf<int64_t> &sel = c->k_sel; DevBuf<float> &nv = c->k_nv, &pl = c->k_pl;
    int rc = sel.reserve(Q);
    if (rc == SICP_OK) rc = nv.reserve((size_t)3 * Q);
    if (rc == SICP_OK) rc = pl.reserve(Q);
    auto body = [&]() -> int {
        HIPCHK(hipMemcpyAsync(sel.p, sel_idx, (size_t)Q * sizeof(int64_t), hipMemcpyDefault, c->stream));
        launch_gather_queries(c->stream, cl.x(), cl.y(), cl.z(), sel.p, Q, qpad, c->kq.p, c->kq.p + qpad, c->kq.p + 2 * qpad);
        bool fused = false;
        CHK(knnk_device(c, slot, c->kq.p, Q, qpad, k, want_lists ? c->k_d2.p : nullptr, want_lists ? c->k_idx.p : nullptr, nv.p, pl.p,
                        &fused));
        if (!fused) launch_normals(c->stream, cl.x(), cl.y(), cl.z(), c->k_idx.p, Q, k, cl.idx_base, nv.p, pl.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(normals_out, nv.p, (size_t)3 * Q * sizeof(float), hipMemcpyDefault, c->stream));
        HIPCHK(hipMemcpyAsync(planarity_out, pl.p, (size_t)Q * sizeof(float), hipMemcpyDefault, c->stream));
        if (nn_idx_out) HIPCHK(hipMemcpyAsync(nn_idx_out, c->k_idx.p, (size_t)Q * k * sizeof(int64_t), hipMemcpyDefault, c->stream));
        return sync(c);
    };
    if (rc == SICP_OK) rc = body();
    if (rc != SICP_OK) (void)hipStreamSynchronize(c->stream);
    return rc;
}

// ------------------------------------------------------------------------------------------
SICP_EXPORT int sicp_icp_setup(sicp_ctx *c, const int64_t *sel_idx, int64_t Q, const float *normals, const float *planarity)
{
    CHK(check_slot(c, SICP_FIX, true));
    if (!sel_idx || !normals || !planarity) return fail(SICP_ERR_INVALID, "null argument");
    if (Q <= 0) return fail(SICP_ERR_INVALID, "Q must be > 0");
    HIPCHK(hipSetDevice(c->device));
    Cloud &cl = c->cloud[SICP_FIX];
    CHK(check_rows(sel_idx, Q, cl.n, "sel_idx"));
    c->Q = Q; c->qpad = round_up(Q, QPAD);
    CHK(c->q.reserve((size_t)3 * c->qpad));
    CHK(c->normals.reserve((size_t)3 * Q)); CHK(c->planarity.reserve(Q));
    CHK(c->m_idx.reserve(Q)); CHK(c->m_d2.reserve(Q)); CHK(c->m_p2.reserve((size_t)3 * Q));
    CHK(c->dist.reserve(Q)); CHK(c->resid.reserve(Q)); CHK(c->flag.reserve(Q)); CHK(c->keep.reserve(Q));
    if (Q > SOLVE_MAX_Q) CHK(c->resid2.reserve(Q));
    c->resid_slot = 0; c->resid_sharded = false;
    HIPCHK(hipMemcpyAsync(c->m_idx.p, sel_idx, (size_t)Q * sizeof(int64_t), hipMemcpyDefault, c->stream));
    launch_gather_queries(c->stream, cl.x(), cl.y(), cl.z(), c->m_idx.p, Q, c->qpad, c->q.p, c->q.p + c->qpad, c->q.p + 2 * c->qpad);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->normals.p, normals, (size_t)3 * Q * sizeof(float), hipMemcpyDefault, c->stream));
    HIPCHK(hipMemcpyAsync(c->planarity.p, planarity, (size_t)Q * sizeof(float), hipMemcpyDefault, c->stream));
    c->have_iter = false;
    c->have_corr = false;
    c->have_prev_match = false;
    c->q_order_lo = -1; c->q_order_cnt = 0;
    c->hsel_run_launches = 0;
    return sync(c);
}

namespace {

int too_few(long long n)
{
    return fail(SICP_ERR_TOO_FEW, "Too few correspondences! At least 6 correspondences are needed to estimate the 6 "
                                  "rigid body transformation parameters. The current number of correspondences is %lld.", n);
}

int check_iter_args(sicp_ctx *c, const sicp_iter_params *P)
{
    if (c->Q <= 0) return fail(SICP_ERR_INVALID, "call sicp_icp_setup first");
    CHK(check_slot(c, SICP_MOV, true));
    for (int j = 0; j < 6; ++j)
        if (std::isnan(P->obs_weight[j]) || P->obs_weight[j] < 0) return fail(SICP_ERR_INVALID, "obs_weight[%d] must be >= 0", j);
    return SICP_OK;
}

// does this configuration run the single-launch tail (sicp_tail.hip) with the loop state on the device?
// (a sharded 6x6 reduction -- gn_shard -- runs there as well: one all-reduce of the 8x8 Gram block per evaluation)
bool device_tail(const sicp_ctx *c) { return c->solve_mode != 2; }

// ---- iterations enqueued back to back, loop state on the device --------------------------------------------------
// Q <= SOLVE_MAX_Q: match + ONE tail launch per iteration (sicp_tail.hip).  Larger Q: match, distances, rejection,
// statistics, `lm_evals` multi-workgroup evaluations whose last block advances the solver, and a finishing launch
// (sicp_lm.hip).  Either way the last kernel of an iteration reads the estimate it starts from out of the
// device-resident loop state and leaves the next one there (with H(x), its inverse, the frozen weight, the
// convergence verdict); with the grid search the match kernel takes its transform from that state too, so
// `chain_depth` iterations are in flight ahead of the record the host is reading and nothing waits for a host
// round trip.  Launches after the end of the run (converged / failed) see the stop flag and exit at once.
// min_change < 0: no convergence test.
int run_device_tail(sicp_ctx *c, const sicp_iter_params *P0, int64_t max_it, double min_change, sicp_iter_result *results,
                    int64_t *done_out)
{
    const long Q = c->Q;
    Cloud &cl = c->cloud[SICP_MOV];
    *done_out = 0;
    if (max_it <= 0) return SICP_OK;
    c->have_corr = false;
    c->resid_sharded = false;
    // the pruned exact search on the static grid serves every rigid H, i.e. every H(x) of the loop
    const bool grid = (c->knn1_mode == 0 || c->knn1_mode == 3) && cl.n < (1LL << 31);
    if (grid) {
        long lo0 = 0, cnt0 = Q;
        if (c->collective() && c->partition == SICP_PART_QUERIES) cnt0 = query_slice(c, Q, &lo0);      // (this rank's share selects the kernel)
        CHK(grid_build(c, SICP_MOV, cnt0));
    }
    const bool small_q = Q <= SOLVE_MAX_Q;
    const int depth = !grid ? 1 : small_q ? c->chain_depth : std::min(c->chain_depth, 2);

    IcpDev &hs = *c->h_state;
    std::memset(&hs, 0, sizeof hs);
    double H12[12];
    params_to_H12(P0->x, H12);
    for (int j = 0; j < 6; ++j) hs.x[j] = P0->x[j];
    for (int j = 0; j < 3; ++j) { hs.sc[2 * j] = std::sin(P0->x[j]); hs.sc[2 * j + 1] = std::cos(P0->x[j]); }
    for (int i = 0; i < 12; ++i) hs.H.m[i] = H12[i];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) hs.Hinv.m[4 * i + j] = H12[4 * j + i];
        hs.Hinv.m[4 * i + 3] = -(H12[i] * H12[3] + H12[4 + i] * H12[7] + H12[8 + i] * H12[11]);
    }
    hs.w = (P0->distance_weight > 0) ? P0->distance_weight : -1.0;
    HIPCHK(hipMemcpyAsync(c->icp_dev.p, &hs, sizeof hs, hipMemcpyHostToDevice, c->stream));
    if (!small_q) {
        LmDev &hl = *c->h_lm;
        std::memset(&hl, 0, sizeof hl);
        for (int j = 0; j < 6; ++j) { hl.x[j] = hl.xt[j] = hs.x[j]; hl.sc[j] = hl.sct[j] = hs.sc[j]; }
        hl.w = hs.w; hl.first = 1;
        HIPCHK(hipMemcpyAsync(c->lm_dev.p, &hl, sizeof hl, hipMemcpyHostToDevice, c->stream));
    }

    TailArgs A;
    for (int j = 0; j < 6; ++j) { A.obs[j] = P0->obs[j]; A.ow[j] = P0->obs_weight[j]; }
    A.min_change = min_change;
    A.min_planarity = (float)P0->min_planarity;
    A.max_steps = P0->max_lm_steps > 0 ? (int)P0->max_lm_steps : 100;
    A.Q = (int)Q;
    A.pl2 = cl.pl_n > 0 ? cl.pl.p : nullptr;
    A.pl2_n = cl.pl_n;

    double seqs[REC_RING];
    double xcur[6]; std::memcpy(xcur, P0->x, sizeof xcur);
    // how far the last completed iteration moved the estimate, as a displacement at the cloud's edge (translation + rotation x radius);
    // unknown (= far) until a cold run's first record is in, zero for a run that continues from an earlier match
    double last_move = c->have_prev_match ? c->last_move : std::numeric_limits<double>::infinity();
    int64_t launched = 0, completed = 0;
    const bool cold_start = !c->have_prev_match;      // no earlier match of these queries to bound the first searches
    bool over = false;
    int rc = SICP_OK;
    const bool htrace = c->host_trace;
    while (true) {
        while (launched < max_it && launched - completed < depth && !over) {
            const auto h0 = std::chrono::steady_clock::now();
            const double *prev = c->have_prev_match ? c->m_p2.p : nullptr;
            const bool qshard = c->collective() && c->partition == SICP_PART_QUERIES;
            bool post_done = false;             // distances + planarity verdicts already written by the match kernel
            bool packed = false;                // ... the exchange's packed records
            if (grid) {
                // (query shards: this rank searches its slice of the queries in the whole cloud, results land in
                // their place in the full arrays)
                long lo = 0, cnt = Q;
                if (qshard) cnt = query_slice(c, Q, &lo);
                c->last_match_kernel = (cnt >= c->nn16_min_q && (!cl.grid.nonuniform || cnt >= c->nn16f_min_q)) ? 5 : 2;
                const bool ordered = c->order_min_q > 0 && cnt >= c->order_min_q;
                if (ordered) CHK(query_order_build(c, lo, cnt, cl.grid.g.h));
                // A search without a useful bound (the run's first iterations: no previous match, or one made under an estimate
                // that was metres off) first asks the cloud's SUBSAMPLE for its nearest point: a cloud point, so a bound, and
                // close enough to the answer that the real search goes straight to that radius instead of doubling its way out.
                const bool coarse = cold_start && launched < c->coarse_iters && cl.n >= c->coarse_min_n && cnt > 0;
                if (coarse) {
                    CHK(subsample_build(c, SICP_MOV));
                    CHK(c->bound_p2.reserve((size_t)3 * Q)); CHK(c->bound_d2.reserve(Q)); CHK(c->bound_idx.reserve(Q));
                }
                // EIGHT queries per wave (8 lanes each) once the query set is large and cells are small: twice the independent
                // requests per wave in flight (0.69 -> 0.62 ms per 1 M queries on 10 M points); not with long rows (C5 sizes: the cell
                // table's limit leaves 25 points per cell, 8 lanes need twice the steps: 2.07 -> 2.53 ms per step) nor below ~200 k
                // queries (too few waves to fill the machine)
                const bool eight = c->nn_group ? c->nn_group == 8 : (cnt >= 196608 && !cl.grid.cap_limited && cl.grid.avg_per_cell <= 20.0);
                GridLevel coarse_lv; const GridLevel *coarse_grid = nullptr;
                CHK(grid_coarse_level(c, SICP_MOV, &coarse_lv, &coarse_grid));
                // (a nonuniform cloud: one wave per query -- 64 rows per batch, the coarse grid for wide passes -- until the filtered search takes over)
                const bool many_q = cnt >= c->nn16_min_q && (!cl.grid.nonuniform || cnt >= c->nn16f_min_q);
                // far searches (a run's first iterations) trim their rows by the tight boxes of the cells; large query sets are
                // searched through the float32 filter (sicp_gridf.hip) when float32 can hold the cloud
                const bool boxes = c->use_boxes && cnt > 0;
                bool filt = many_q && c->nn16_filter != 0 && cnt > 0 && cnt >= c->nn16f_min_q;
                if (filt || boxes) CHK(grid_companions(c, cl, cl.grid, cl.n, filt, boxes));
                if (filt && coarse) CHK(grid_companions(c, cl, cl.sub_grid, cl.sub_n, true, false));
                if (filt && (!cl.grid.filter_ok || (coarse && !cl.sub_grid.filter_ok))) filt = false;
                const unsigned long long *cbox = boxes ? cl.grid.cell_box.p : nullptr;
                if (filt) {
                    // queries and their last matches in slot order (once per setup: a new setup, a new cloud or another slice start afresh)
                    if (!c->have_prev_match || c->slot_lo != lo || c->slot_cnt != cnt || c->slot_ordered != ordered) {
                        CHK(c->q_slot.reserve((size_t)4 * cnt)); CHK(c->p_slot.reserve((size_t)4 * cnt));
                        CHK(c->nn_state.reserve((size_t)cnt));
                        if (c->nn_redo.cap < (size_t)cnt + 2) {
                            CHK(c->nn_redo.reserve((size_t)cnt + 2));
                            HIPCHK(hipMemsetAsync(c->nn_redo.p, 0, 2 * sizeof(uint32_t), c->stream));
                        }
                        launch_slot_queries(c->stream, c->q.p + lo, c->q.p + c->qpad + lo, c->q.p + 2 * c->qpad + lo,
                                            ordered ? c->q_order.p : nullptr, nullptr, cnt, c->q_slot.p, c->p_slot.p);
                        HIPCHK(hipGetLastError());
                        c->slot_lo = lo; c->slot_cnt = cnt; c->slot_ordered = ordered;
                    }
                    c->last_match_kernel = 6;
                    const int lanes = eight ? 8 : 16;
                    unsigned *tie_cnt = c->nn_redo.p + c->nn_parity, *tie_clear = c->nn_redo.p + (c->nn_parity ^ 1);
                    uint32_t *tie_list = c->nn_redo.p + 2;
                    unsigned long long *wk = c->count_work ? c->match_work.p : nullptr;
                    const double inf = std::numeric_limits<double>::infinity();
                    Timed t(c, SICP_K_KNN1);
                    // cold: the subsample's nearest point (any point near the query: NN_APPROX) is left in the slot as the bound ...
                    if (coarse)
                        launch_grid_nn16f(c->stream, lanes, true, c->icp_dev.p, c->q_slot.p, c->p_slot.p, cnt, cl.sub_grid.g, cl.sub_grid.c0,
                                          cl.sub_grid.eps_p, cl.sub_grid.cell_start.p, cl.sub_grid.recf.p, cl.sub_grid.rec.p, ordered,
                                          nullptr, nullptr, cl.rmax, inf, 0, nullptr, nullptr, nullptr, nullptr, NN_APPROX, nullptr,
                                          tie_list, tie_cnt);
                    // ... and the search proper goes straight to that radius (NN_TIGHT).  A cold search is a far search for every
                    // query: the full flavour takes all slots.  Later the lean flavour goes first and marks what it cannot do.
                    // (the estimate still moves by a cell or so per iteration: most searches are wide -- the lean flavour would only find
                    // that out and hand them on; judged from the last iterations the host has seen: the chain runs ahead of it)
                    const bool all_far = coarse || c->nn16_filter == 1 || !(last_move <= c->far_move * cl.grid.g.h);
                    if (!all_far)
                        launch_grid_nn16f(c->stream, lanes, false, c->icp_dev.p, c->q_slot.p, c->p_slot.p, cnt, cl.grid.g, cl.grid.c0,
                                          cl.grid.eps_p, cl.grid.cell_start.p, cl.grid.recf.p, cl.grid.rec.p, ordered, nullptr, nullptr,
                                          cl.rmax, inf, cl.idx_base, c->m_d2.p + lo, c->m_idx.p + lo, c->m_p2.p + 3 * lo, wk, 0,
                                          c->nn_state.p, tie_list, tie_cnt);
                    launch_grid_nn16f(c->stream, lanes, true, c->icp_dev.p, c->q_slot.p, c->p_slot.p, cnt, cl.grid.g, cl.grid.c0,
                                      cl.grid.eps_p, cl.grid.cell_start.p, cl.grid.recf.p, cl.grid.rec.p, ordered, nullptr, nullptr,
                                      cl.rmax, inf, cl.idx_base, c->m_d2.p + lo, c->m_idx.p + lo, c->m_p2.p + 3 * lo, wk,
                                      (coarse ? NN_TIGHT : 0), all_far ? nullptr : c->nn_state.p, tie_list, tie_cnt);
                    // ties within the filter's margin (and queries float32 cannot place): the exact kernel, from the by-query
                    // arrays (the previous match bounds them; in a cold iteration nothing does: they search outwards)
                    // (the filtered kernels left every such query's approximate winner -- or "none" -- in the by-query match array)
                    launch_grid_nn_redo(c->stream, c->q.p + lo, c->q.p + c->qpad + lo, c->q.p + 2 * c->qpad + lo, cnt,
                                        c->m_p2.p + 3 * lo, cl.grid.g, cl.grid.cell_start.p, cl.grid.rec.p,
                                        c->icp_dev.p, nullptr, nullptr, cl.rmax, inf, cl.idx_base, c->m_d2.p + lo, c->m_idx.p + lo,
                                        c->m_p2.p + 3 * lo, wk, NN_TIGHT, nullptr, cbox, tie_list, tie_cnt, tie_clear, coarse_grid);
                    c->nn_parity ^= 1;
                } else {
                Timed t(c, SICP_K_KNN1);
                if (coarse)
                    launch_grid_nn_chained(c->stream, c->q.p + lo, c->q.p + c->qpad + lo, c->q.p + 2 * c->qpad + lo, cnt, nullptr,
                                           cl.sub_grid.g, cl.sub_grid.cell_start.p, cl.sub_grid.rec.p, c->icp_dev.p, cl.rmax, 0,
                                           c->bound_d2.p + lo, c->bound_idx.p + lo, c->bound_p2.p + 3 * lo, nullptr,
                                           ordered ? c->q_order.p : nullptr, many_q, NN_APPROX);
                // without an exchange the match is final when its kernel ends: the winning lanes leave the point-to-plane
                // distance and the planarity verdict too (what k_postmatch would re-read 72 bytes per correspondence for)
                // (only in the one-wave-per-query flavour: with four queries per wave at the register limit the epilogue's late
                // loads cost the search more than k_postmatch's launch -- match 693 -> 758 us at 1 M queries, measured)
                post_done = !c->collective() && !many_q;
                // behind a cloud-shard exchange the winning lanes leave the exchange's packed record instead (no k_pack_best launch)
                // (query shards: the slim record, the matched index alone -- no k_pack_idx launch)
                const bool pack = c->collective() && !qshard, pack_idx = c->collective() && qshard;
                if (pack) CHK(c->x_send.reserve((size_t)5 * Q));
                if (pack_idx) CHK(c->x_send.reserve((size_t)((Q + c->world - 1) / c->world)));
                PostMatch pm = {c->normals.p, c->planarity.p, A.pl2, A.pl2_n, A.min_planarity, post_done ? c->dist.p : nullptr,
                                post_done ? c->flag.p : nullptr, pack ? c->x_send.p : nullptr, pack_idx ? c->x_send.p : nullptr};
                packed = (pack || pack_idx) && cnt > 0;
                if (cnt > 0)
                    launch_grid_nn_chained(c->stream, c->q.p + lo, c->q.p + c->qpad + lo, c->q.p + 2 * c->qpad + lo, cnt,
                                           coarse ? c->bound_p2.p + 3 * lo : (prev ? prev + 3 * lo : nullptr), cl.grid.g,
                                           cl.grid.cell_start.p, cl.grid.rec.p, c->icp_dev.p, cl.rmax, cl.idx_base, c->m_d2.p + lo,
                                           c->m_idx.p + lo, c->m_p2.p + 3 * lo, c->count_work ? c->match_work.p : nullptr,
                                           ordered ? c->q_order.p : nullptr, many_q, (coarse ? NN_TIGHT : 0),
                                           (post_done || pack || pack_idx) ? &pm : nullptr, eight, cbox, coarse_grid);
                }
            } else if (qshard) {
                return fail(SICP_ERR_INVALID, "query shards need the grid search (SICP_KNN1 forces another kernel)");
            } else {
                // brute-force flavours take H by value: one iteration in flight, H from the last record
                params_to_H12(xcur, H12);
                Xf X; for (int i = 0; i < 12; ++i) X.m[i] = H12[i];
                CHK(knn1_device(c, SICP_MOV, c->q.p, Q, c->qpad, &X, std::numeric_limits<double>::infinity(), prev, c->m_d2.p,
                                c->m_idx.p, c->m_p2.p));
            }
            HIPCHK(hipGetLastError());
            c->have_prev_match = true;          // (after an exchange: the job-wide winner's coordinates -- still a valid bound)
            if (qshard) { CHK(exchange_query_slices_idx(c, A, Q, packed)); post_done = true; }      // (distances + verdicts formed by the unpack)
            else if (c->collective() && c->partition == SICP_PART_CLOUD) {
                CHK(c->x_send.reserve((size_t)5 * Q));
                CHK(exchange_best_chained(c, A, Q, packed)); post_done = true;                 // (... by the lexicographic minimum's kernel)
            }
            A.seq = (double)(++c->solve_seq);
            seqs[launched % REC_RING] = A.seq;
            double *rec = c->h_rec + (launched % REC_RING) * REC_DOUBLES;
            const double *qx = c->q.p, *qy = c->q.p + c->qpad, *qz = c->q.p + 2 * c->qpad;
            Xf unused = {};
            if (small_q) {
                if (!post_done)
                    launch_postmatch(c->stream, qx, qy, qz, c->normals.p, c->planarity.p, c->m_p2.p, c->m_idx.p, Q, unused,
                                     A.min_planarity, A.pl2, A.pl2_n, c->dist.p, c->flag.p, c->icp_dev.p);
                Timed t(c, SICP_K_NORMALEQ);
                launch_icp_tail(c->stream, qx, qy, qz, c->normals.p, c->m_p2.p, A, c->icp_dev.p, c->dist.p, c->flag.p, c->keep.p,
                                c->resid.p, rec);
            } else {
                // distances + rejections (corrpts.py:139-211), kept-distance statistics, then the solver chain
                if (Q <= REJECT_MAX_Q) {
                    Timed t(c, SICP_K_SELECT);
                    // distances + flags by the whole machine (the match kernel's epilogue, or k_postmatch behind an exchange), then
                    // selection + keep mask + statistics by one workgroup on the 9 bytes per correspondence it still has to read
                    if (!post_done)
                        launch_postmatch(c->stream, qx, qy, qz, c->normals.p, c->planarity.p, c->m_p2.p, c->m_idx.p, Q, unused,
                                         A.min_planarity, A.pl2, A.pl2_n, c->dist.p, c->flag.p, c->icp_dev.p);
                    launch_reject(c->stream, c->dist.p, c->flag.p, Q, c->keep.p, c->small.p, c->icp_dev.p, c->small.p + 4);
                } else {
                    if (!post_done)
                        launch_postmatch(c->stream, qx, qy, qz, c->normals.p, c->planarity.p, c->m_p2.p, c->m_idx.p, Q, unused,
                                         A.min_planarity, A.pl2, A.pl2_n, c->dist.p, c->flag.p, c->icp_dev.p);
                    {
                        // median / MAD by digit selection over many workgroups, keep mask + kept statistics in one more pass
                        Timed t(c, SICP_K_SELECT);
                        CHK(reject_select(c, Q, nullptr, 0.0, c->icp_dev.p));
                    }
                }
                {
                    // gn_shard: every rank evaluates its slice of the correspondences, ONE all-reduce adds the
                    // 8x8 Gram blocks (J^T J, J^T r, sum r, sum r^2, n) up, a one-wave launch advances the replicated solver
                    const bool shard = c->gn_shard && c->collective();
                    if (shard) CHK(c->lm_gsum.reserve(64));
                    c->resid_sharded = shard;
                    Timed t(c, SICP_K_NORMALEQ);
                    if (c->lm_one_launch && !shard) {
                        const size_t words = (lm_bar_bytes() + 7) / 8;
                        if (c->lm_bar_buf.cap < words) {
                            CHK(c->lm_bar_buf.reserve(words));
                            HIPCHK(hipMemsetAsync(c->lm_bar_buf.p, 0, words * 8, c->stream));
                            c->lm_bar = 0;
                        }
                        launch_lm_all(c->stream, qx, qy, qz, c->normals.p, c->m_p2.p, c->keep.p, Q, A, c->icp_dev.p, c->lm_dev.p, c->small.p,
                                      c->small.p + 4, c->ne_partial.p, c->lm_bar_buf.p, &c->lm_bar, c->resid.p, c->resid2.p, rec,
                                      c->test_barrier_fault == 2 ? 1u : 0u);
                    } else {
                        for (int e = 0; e < c->lm_evals; ++e) {
                            launch_lm_eval(c->stream, qx, qy, qz, c->normals.p, c->m_p2.p, c->keep.p, Q, A, c->icp_dev.p, c->lm_dev.p,
                                           c->small.p + 4, c->ne_partial.p, c->ticket.p, c->resid.p, c->resid2.p,
                                           shard ? c->rank : 0, shard ? c->world : 1, shard ? c->lm_gsum.p : nullptr);
                            if (shard) {
                                CHK(all_reduce_sum_f64(c, c->lm_gsum.p, 64));
                                launch_lm_advance(c->stream, A, c->icp_dev.p, c->lm_dev.p, c->small.p + 4, c->lm_gsum.p);
                            }
                        }
                        launch_lm_finish(c->stream, qx, qy, qz, c->normals.p, c->m_p2.p, c->keep.p, Q, A, c->icp_dev.p, c->lm_dev.p,
                                         c->small.p, c->small.p + 4, c->resid.p, c->resid2.p, rec);
                    }
                }
            }
            HIPCHK(hipGetLastError());
            ++launched;
            if (htrace) {
                const auto h1 = std::chrono::steady_clock::now();
                std::fprintf(stderr, "[host] iteration %lld enqueued in %.1f us\n", (long long)launched,
                             std::chrono::duration<double, std::micro>(h1 - h0).count());
            }
        }
        if (completed == launched) break;
        const double *o = c->h_rec + (completed % REC_RING) * REC_DOUBLES;
        CHK(wait_ticket(c, o + REC_TICKET, seqs[completed % REC_RING]));
        const int status = (int)o[REC_STATUS];
        if (status == 3) { ++completed; over = true; continue; }        // launched after the end of the run: not an iteration
        if (status == 4) {
            // a one-launch kernel could not meet itself at its grid barrier (its blocks were not all resident: CUs held by another
            // process, a paused queue).  The launches behind it see the stop flag; start the barrier state afresh so that the next
            // run is not poisoned by this one (the error word is sticky on the device by design: every later phase must see it)
            ++completed; over = true;
            (void)hipStreamSynchronize(c->stream);
            CHK(reset_barrier_state(c));
            c->have_iter = false;
            rc = fail(SICP_ERR_HIP, "a device-wide barrier of iteration %lld timed out (blocks not co-resident: is another process "
                                    "using the GPU?); the run was stopped and the barrier state reset", (long long)completed);
            continue;
        }
        sicp_iter_result &R = results[*done_out];
        std::memset(&R, 0, sizeof R);
        R.n_queries = Q; R.n_planar = (int64_t)o[0]; R.median = o[1]; R.mad = o[2]; R.n_kept = (int64_t)o[3];
        R.dist_mean = o[4]; R.dist_std = o[5];
        for (int j = 0; j < 6; ++j) R.x[j] = o[10 + j];
        ++completed; ++*done_out;
        c->have_iter = true;
        c->have_last_ne = false;
        std::memcpy(c->last_x, R.x, sizeof c->last_x);
        if (status == 1 || R.n_kept < 6) { rc = too_few((long long)R.n_kept); over = true; continue; }
        if (status != 0) { rc = fail(SICP_ERR_NUMERIC, "objective is not finite"); over = true; continue; }
        R.weight_used = o[6]; R.cost = o[7]; R.lm_steps = (int64_t)o[8]; R.ne_evals = (int64_t)o[9];
        R.res_mean = o[16]; R.res_std = o[17];
        params_to_H12(R.x, R.H);
        R.H[12] = 0; R.H[13] = 0; R.H[14] = 0; R.H[15] = 1;
        {
            double dt = 0, da = 0;
            for (int j = 0; j < 3; ++j) { da += (R.x[j] - xcur[j]) * (R.x[j] - xcur[j]); dt += (R.x[3 + j] - xcur[3 + j]) * (R.x[3 + j] - xcur[3 + j]); }
            last_move = std::sqrt(dt) + std::sqrt(da) * cl.rmax;
            c->last_move = last_move;                    // (a host-driven loop -- one iteration per call -- carries it from call to call)
        }
        std::memcpy(xcur, R.x, sizeof xcur);
        c->last_w = R.weight_used;
        std::memcpy(c->last_obs, P0->obs, sizeof c->last_obs);
        std::memcpy(c->last_ow, P0->obs_weight, sizeof c->last_ow);
        std::memcpy(c->last_ne, o + 20, sizeof c->last_ne);
        c->have_last_ne = true;
        c->resid_slot = small_q ? 0 : (int)o[REC_RESID_SLOT];
        if (small_q) std::memcpy(c->last_tail_cycles, o + 50, 5 * sizeof(double));
        if (c->solve_trace && small_q)
            std::fprintf(stderr, "[tail] cycles: load+dist %.0f select %.0f (median %.0f in %.0f rounds, MAD %.0f in %.0f) keep %.0f lm %.0f "
                                 "(%lld evals %.0f, %lld steps, solves %.0f, accept %.0f) final %.0f\n",
                         o[50], o[51], o[55], o[56], o[57], o[58], o[52], o[53], (long long)R.ne_evals, o[59], (long long)R.lm_steps, o[60], o[62], o[54]);
        if (c->solve_trace && small_q && std::getenv("SICP_SEL_TRACE"))      // (a -DSICP_SEL_FINE_TRACE build: build.build_variant)
            std::fprintf(stderr, "[sel] median: atomics+barrier %.0f fold+barrier %.0f scan+pick %.0f (more rounds %.0f) gather+barrier %.0f rank %.0f | "
                                 "MAD: %.0f %.0f %.0f (%.0f) %.0f %.0f\n", o[38], o[39], o[40], o[41], o[42], o[43], o[44], o[45], o[46], o[47], o[48], o[49]);
        if (c->solve_trace && small_q && std::getenv("SICP_EVAL_TRACE"))     // (a -DSICP_EVAL_FINE_TRACE build)
            std::fprintf(stderr, "[eval] rows + LDS writes %.0f barrier %.0f MFMA Gram %.0f block write + barrier %.0f fold %.0f\n", o[38], o[39], o[40], o[41], o[42]);
        if (o[REC_CONVERGED] != 0.0) over = true;
    }
    return rc;
}

// ---- optimisation: optimization.py:65-124 as LM on fused 6x6 reductions over the rows of `keep`, from P->x ----
// Expects R->dist_std (kept distances, for the automatic weight); fills the solver's part of R, the residuals at the
// optimum (c->resid) and the state sicp_icp_uncertainties reads.
int host_lm_solve(sicp_ctx *c, const sicp_iter_params *P, sicp_iter_result *R)
{
    const long Q = c->Q;
    int nfree = 0, freeidx[6];
    for (int j = 0; j < 6; ++j)
        if (std::isfinite(P->obs_weight[j])) freeidx[nfree++] = j;
    double *h_st = c->h_small + 160;                      // pinned: [4..6] n / mean / std, [15] ticket
    double w = P->distance_weight;
    if (!(w > 0)) w = 1.0 / (R->dist_std * R->dist_std);   // simpleicp.py:233-234
    R->weight_used = w;
    const double *obs = P->obs, *ow = P->obs_weight;
    double x[6]; std::memcpy(x, P->x, sizeof x);
    double ne[30];
    CHK(normal_eq_host(c, x, false, true, ne)); R->ne_evals++;
    double cost = objective(ne, w, x, obs, ow);
    double lambda = 0.0;
    const int max_steps = P->max_lm_steps > 0 ? (int)P->max_lm_steps : 100;
    for (int it = 0; it < max_steps && nfree > 0; ++it) {
        double N[36], g[6];
        int t = 0;
        for (int u = 0; u < 6; ++u) for (int v = u; v < 6; ++v) { N[u * 6 + v] = N[v * 6 + u] = w * w * ne[t++]; }
        for (int u = 0; u < 6; ++u) g[u] = w * w * ne[21 + u];
        for (int j = 0; j < 6; ++j)
            if (is_observed(ow[j])) { N[j * 6 + j] += ow[j] * ow[j]; g[j] += ow[j] * ow[j] * (x[j] - obs[j]); }
        bool accepted = false, converged = false;
        double xn[6], nen[30], costn = cost, dxmax = 0;
        for (int tries = 0; tries < 40; ++tries) {
            double A[36], b[6];
            for (int u = 0; u < nfree; ++u) {
                for (int v = 0; v < nfree; ++v) A[u * nfree + v] = N[freeidx[u] * 6 + freeidx[v]];
                A[u * nfree + u] += lambda * N[freeidx[u] * 6 + freeidx[u]];
                b[u] = -g[freeidx[u]];
            }
            if (!spd_solve(nfree, A, b)) { lambda = lambda > 0 ? lambda * 10 : 1e-6; continue; }
            std::memcpy(xn, x, sizeof x);
            dxmax = 0;
            for (int u = 0; u < nfree; ++u) { xn[freeidx[u]] += b[u]; dxmax = std::max(dxmax, std::fabs(b[u])); }
            {
                double xm = 0; for (int j = 0; j < 6; ++j) xm = std::max(xm, std::fabs(x[j]));
                if (lambda == 0.0 && dxmax <= 1e-10 * (1.0 + xm)) { converged = true; break; }   // see k_icp_solve
            }
            CHK(normal_eq_host(c, xn, false, true, nen)); R->ne_evals++;
            costn = objective(nen, w, xn, obs, ow);
            if (costn <= cost * (1 + 1e-12) || dxmax < 1e-15) { accepted = true; break; }
            lambda = lambda > 0 ? lambda * 10 : 1e-6;
        }
        if (converged || !accepted) break;
        std::memcpy(x, xn, sizeof x); std::memcpy(ne, nen, sizeof ne);
        cost = costn;
        lambda = lambda > 0 ? lambda * 0.1 : 0.0;
        if (lambda < 1e-12) lambda = 0.0;
        R->lm_steps++;
        double xmax = 0; for (int j = 0; j < 6; ++j) xmax = std::max(xmax, std::fabs(x[j]));
        if (dxmax <= 1e-13 * (1.0 + xmax)) break;
    }
    if (!std::isfinite(cost)) return fail(SICP_ERR_NUMERIC, "objective is not finite");

    // ---- residuals at the optimum (optimization.py:117-124) + their mean/std (simpleicp.py:356-379) ----
    CHK(normal_eq_host(c, x, true, false, ne)); R->ne_evals++;
    cost = objective(ne, w, x, obs, ow);
    const double seq = (double)(++c->solve_seq);
    launch_stats(c->stream, c->resid.p, c->keep.p, Q, c->small.p + 4, nullptr, h_st, seq, c->ne_partial.p, c->ticket.p);
    HIPCHK(hipGetLastError());
    CHK(wait_ticket(c, h_st + 15, seq));
    R->res_mean = h_st[5]; R->res_std = h_st[6];
    R->cost = cost;
    std::memcpy(R->x, x, sizeof x);
    params_to_H12(x, R->H);
    R->H[12] = 0; R->H[13] = 0; R->H[14] = 0; R->H[15] = 1;
    std::memcpy(c->last_x, x, sizeof x);
    c->last_w = w;
    std::memcpy(c->last_obs, obs, sizeof c->last_obs);
    std::memcpy(c->last_ow, ow, sizeof c->last_ow);
    return SICP_OK;
}

// ---- larger Q (or a sharded 6x6 reduction): multi-kernel tail, LM loop on the host ---------------------------
int iterate_host_lm(sicp_ctx *c, const sicp_iter_params *P, sicp_iter_result *R)
{
    std::memset(R, 0, sizeof *R);
    const long Q = c->Q;
    c->resid_slot = 0; c->resid_sharded = false;
    c->have_corr = false;
    // ---- match: simpleicp.py:188-202, corrpts.py:124-137 (transform fused into the scan) ----
    double H12[12];
    params_to_H12(P->x, H12);
    Xf X; for (int i = 0; i < 12; ++i) X.m[i] = H12[i];
    CHK(knn1_device(c, SICP_MOV, c->q.p, Q, c->qpad, &X, std::numeric_limits<double>::infinity(),
                    c->have_prev_match ? c->m_p2.p : nullptr, c->m_d2.p, c->m_idx.p, c->m_p2.p));
    c->have_prev_match = true;              // (after an exchange: the job-wide winner's coordinates -- still a valid bound)
    CHK(exchange_best(c, c->m_d2.p, c->m_idx.p, c->m_p2.p, Q));
    c->have_last_ne = false;
    // ---- distances + rejections: corrpts.py:139-211 ----
    launch_postmatch(c->stream, c->q.p, c->q.p + c->qpad, c->q.p + 2 * c->qpad, c->normals.p, c->planarity.p, c->m_p2.p,
                     c->m_idx.p, Q, X, (float)P->min_planarity, c->cloud[SICP_MOV].pl_n > 0 ? c->cloud[SICP_MOV].pl.p : nullptr,
                     c->cloud[SICP_MOV].pl_n, c->dist.p, c->flag.p);
    double *h_st = c->h_small + 160;                      // pinned: [0..3] rejection, [4..6] n / mean / std, [15] ticket
    double seq = (double)(++c->solve_seq);
    {
        Timed t(c, SICP_K_SELECT);
        if (Q > REJECT_MAX_Q) {
            // one workgroup cannot chew a million distances: exact order statistics by multi-workgroup digit selection,
            // keep mask and kept-distance statistics in its last pass
            CHK(c->ne_partial.reserve((size_t)NE_MAX_GRID * 64));
            CHK(reject_select(c, Q, h_st, seq, nullptr));
        } else {
            launch_reject(c->stream, c->dist.p, c->flag.p, Q, c->keep.p, c->small.p);
            launch_stats(c->stream, c->dist.p, c->keep.p, Q, c->small.p + 4, c->small.p, h_st, seq, c->ne_partial.p, c->ticket.p);
        }
    }
    HIPCHK(hipGetLastError());
    CHK(wait_ticket(c, h_st + 15, seq));
    if (h_st[0] < 0.0) return barrier_timed_out(c);
    R->n_queries = Q;
    R->n_planar = (int64_t)h_st[0];
    R->median = h_st[1]; R->mad = h_st[2];
    R->n_kept = (int64_t)h_st[3];
    R->dist_mean = h_st[5]; R->dist_std = h_st[6];
    c->have_iter = true;
    std::memcpy(c->last_x, P->x, sizeof c->last_x);
    if (R->n_kept < 6) {
        std::memcpy(R->x, P->x, sizeof R->x);
        return too_few((long long)R->n_kept);
    }
    return host_lm_solve(c, P, R);
}

}  // namespace

SICP_EXPORT int sicp_icp_iterate(sicp_ctx *c, const sicp_iter_params *P, sicp_iter_result *R)
{
    if (!c || !P || !R) return fail(SICP_ERR_INVALID, "null argument");
    CHK(check_iter_args(c, P));
    HIPCHK(hipSetDevice(c->device));
    if (!device_tail(c)) return iterate_host_lm(c, P, R);
    std::memset(R, 0, sizeof *R);
    int64_t done = 0;
    return run_device_tail(c, P, 1, -1.0, R, &done);
}

SICP_EXPORT int sicp_icp_run(sicp_ctx *c, const sicp_iter_params *P0, int64_t max_iterations, double min_change,
                             sicp_iter_result *results, int64_t *iterations_out)
{
    if (!c || !P0 || !results || !iterations_out) return fail(SICP_ERR_INVALID, "null argument");
    *iterations_out = 0;
    if (max_iterations <= 0) return SICP_OK;
    CHK(check_iter_args(c, P0));
    HIPCHK(hipSetDevice(c->device));
    if (device_tail(c)) {
        if (std::isnan(min_change)) min_change = 0.0;
        // (a failing iteration's entry carries the estimate it started from: the tail kernel records it)
        return run_device_tail(c, P0, max_iterations, min_change < 0 ? 0.0 : min_change, results, iterations_out);
    }
    sicp_iter_params P = *P0;
    auto change = [](double now, double before) {          // simpleicp.py:361-365
        if (before == 0) return now == 0 ? 0.0 : std::numeric_limits<double>::infinity();
        return std::fabs((now - before) / before * 100.0);
    };
    for (int64_t it = 0; it < max_iterations; ++it) {
        sicp_iter_result &R = results[it];
        const int rc = iterate_host_lm(c, &P, &R);
        *iterations_out = it + 1;
        if (rc != SICP_OK) return rc;
        std::memcpy(P.x, R.x, sizeof P.x);
        if (!(P.distance_weight > 0)) P.distance_weight = R.weight_used;
        if (it > 0 && change(R.res_mean, results[it - 1].res_mean) < min_change &&
            change(R.res_std, results[it - 1].res_std) < min_change)
            break;
    }
    return SICP_OK;
}

SICP_EXPORT int sicp_icp_get_state(sicp_ctx *c, int64_t *pc2_idx, double *dist, uint8_t *keep, double *residual)
{
    if (!c) return fail(SICP_ERR_INVALID, "null ctx");
    if (!c->have_iter && !c->have_corr) return fail(SICP_ERR_INVALID, "no iteration has run yet");
    HIPCHK(hipSetDevice(c->device));
    const size_t Q = (size_t)c->Q;
    if (pc2_idx) HIPCHK(hipMemcpyAsync(pc2_idx, c->m_idx.p, Q * sizeof(int64_t), hipMemcpyDefault, c->stream));
    if (dist) HIPCHK(hipMemcpyAsync(dist, c->dist.p, Q * sizeof(double), hipMemcpyDefault, c->stream));
    if (keep) HIPCHK(hipMemcpyAsync(keep, c->keep.p, Q * sizeof(uint8_t), hipMemcpyDefault, c->stream));
    if (residual && c->resid_sharded && c->have_iter) {
        // the sharded reduction left only this rank's slice of the residuals current: one pass over all of them at the estimate
        double ne[30];
        CHK(normal_eq_host(c, c->last_x, true, false, ne));
        c->resid_slot = 0; c->resid_sharded = false;
    }
    if (residual) HIPCHK(hipMemcpyAsync(residual, c->resid_slot ? c->resid2.p : c->resid.p, Q * sizeof(double), hipMemcpyDefault, c->stream));
    return sync(c);
}

SICP_EXPORT int sicp_icp_normal_equations(sicp_ctx *c, const double x[6], double out[30])
{
    if (!c || !x || !out) return fail(SICP_ERR_INVALID, "null argument");
    if (!c->have_iter) return fail(SICP_ERR_INVALID, "no iteration has run yet");
    HIPCHK(hipSetDevice(c->device));
    return normal_eq_host(c, x, false, false, out);
}

SICP_EXPORT int sicp_icp_uncertainties(sicp_ctx *c, double sigma_out[6])
{
    if (!c || !sigma_out) return fail(SICP_ERR_INVALID, "null argument");
    if (!c->have_iter) return fail(SICP_ERR_INVALID, "no iteration has run yet");
    HIPCHK(hipSetDevice(c->device));
    double ne[30];
    if (c->have_last_ne) std::memcpy(ne, c->last_ne, sizeof ne);
    else CHK(normal_eq_host(c, c->last_x, false, false, ne));
    const double w = c->last_w, *ow = c->last_ow, *obs = c->last_obs, *x = c->last_x;
    int freeidx[6], m = 0, nobs = 0;
    for (int j = 0; j < 6; ++j) { sigma_out[j] = std::numeric_limits<double>::quiet_NaN(); if (std::isfinite(ow[j])) freeidx[m++] = j; }
    // optimization.py:154-159: N = A^T diag(w) A with LINEAR weights, s0^2 = v^T P v / (n_obs - n_prm)
    double N[36]; int t = 0;
    for (int u = 0; u < 6; ++u) for (int v = u; v < 6; ++v) { N[u * 6 + v] = N[v * 6 + u] = w * ne[t++]; }
    double vPv = w * ne[28];
    for (int j = 0; j < 6; ++j)
        if (is_observed(ow[j])) { N[j * 6 + j] += ow[j]; const double e = x[j] - obs[j]; vPv += ow[j] * e * e; ++nobs; }
    const double s02 = vPv / ((ne[29] + nobs) - m);
    for (int u = 0; u < m; ++u) {
        double A[36], b[6];
        for (int a = 0; a < m; ++a) { for (int q = 0; q < m; ++q) A[a * m + q] = N[freeidx[a] * 6 + freeidx[q]]; b[a] = (a == u) ? 1.0 : 0.0; }
        if (!spd_solve(m, A, b)) return fail(SICP_ERR_NUMERIC, "normal matrix is not positive definite");
        sigma_out[freeidx[u]] = std::sqrt(s02 * b[u]);
    }
    return SICP_OK;
}

// ------------------------------------------------------------------------------------------
// The iteration's operators one by one (CorrPts / SimpleICPOptimization as the reference's callers drive them,
// simpleicp.py:190-227): the kernels of the multi-kernel iteration behind separate entry points.  The alive mask of
// the correspondences lives in `keep`.
namespace {

int check_corr(sicp_ctx *c)
{
    if (!c) return fail(SICP_ERR_INVALID, "null ctx");
    if (!c->have_corr) return fail(SICP_ERR_INVALID, "call sicp_corr_match first");
    return SICP_OK;
}

// count / mean / std of the alive correspondences' distances -> pinned h_st[4..6] (and the rejection's out4 -> h_st[0..3])
int corr_alive_stats(sicp_ctx *c, const double *also4, double **h_st_out)
{
    double *h_st = c->h_small + 160;
    const double seq = (double)(++c->solve_seq);
    launch_stats(c->stream, c->dist.p, c->keep.p, c->Q, c->small.p + 4, also4, h_st, seq, c->ne_partial.p, c->ticket.p);
    HIPCHK(hipGetLastError());
    CHK(wait_ticket(c, h_st + 15, seq));
    *h_st_out = h_st;
    return SICP_OK;
}

}  // namespace

SICP_EXPORT int sicp_corr_match(sicp_ctx *c, const double *H, int64_t *pc2_idx_out, double *dist_out)
{
    if (!c) return fail(SICP_ERR_INVALID, "null ctx");
    if (c->Q <= 0) return fail(SICP_ERR_INVALID, "call sicp_icp_setup first");
    CHK(check_slot(c, SICP_MOV, true));
    HIPCHK(hipSetDevice(c->device));
    const long Q = c->Q;
    Xf X = {{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}};           // identity: contract (T) then returns the coordinates unchanged
    if (H) H16_to_Xf(H, &X);
    c->have_corr = false;
    CHK(knn1_device(c, SICP_MOV, c->q.p, Q, c->qpad, &X, std::numeric_limits<double>::infinity(),
                    c->have_prev_match ? c->m_p2.p : nullptr, c->m_d2.p, c->m_idx.p, c->m_p2.p));
    c->have_prev_match = true;
    CHK(exchange_best(c, c->m_d2.p, c->m_idx.p, c->m_p2.p, Q));
    c->have_last_ne = false;
    c->have_iter = false;                                     // no estimate belongs to these correspondences yet
    c->resid_slot = 0; c->resid_sharded = false;
    // distances (contract (P)); the flags of this launch are not used: nothing is rejected yet
    launch_postmatch(c->stream, c->q.p, c->q.p + c->qpad, c->q.p + 2 * c->qpad, c->normals.p, c->planarity.p, c->m_p2.p,
                     c->m_idx.p, Q, X, -std::numeric_limits<float>::infinity(), nullptr, 0, c->dist.p, c->flag.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(c->keep.p, 1, (size_t)Q, c->stream));
    HIPCHK(hipMemsetAsync(c->resid.p, 0, (size_t)Q * sizeof(double), c->stream));
    if (pc2_idx_out) HIPCHK(hipMemcpyAsync(pc2_idx_out, c->m_idx.p, (size_t)Q * sizeof(int64_t), hipMemcpyDefault, c->stream));
    if (dist_out) HIPCHK(hipMemcpyAsync(dist_out, c->dist.p, (size_t)Q * sizeof(double), hipMemcpyDefault, c->stream));
    CHK(sync(c));
    c->have_corr = true;
    return SICP_OK;
}

SICP_EXPORT int sicp_corr_reject_planarity(sicp_ctx *c, double min_planarity, const float *pc1_planarity,
                                           const float *pc2_planarity, int64_t *n_alive_out)
{
    CHK(check_corr(c));
    if (std::isnan(min_planarity)) return fail(SICP_ERR_INVALID, "min_planarity is NaN");
    HIPCHK(hipSetDevice(c->device));
    const long Q = c->Q;
    CHK(c->corr_pl.reserve((size_t)2 * Q));
    float *d1 = pc1_planarity ? c->corr_pl.p : nullptr, *d2 = pc2_planarity ? c->corr_pl.p + Q : nullptr;
    if (d1) HIPCHK(hipMemcpyAsync(d1, pc1_planarity, (size_t)Q * sizeof(float), hipMemcpyDefault, c->stream));
    if (d2) HIPCHK(hipMemcpyAsync(d2, pc2_planarity, (size_t)Q * sizeof(float), hipMemcpyDefault, c->stream));
    launch_corr_planarity(c->stream, c->keep.p, d1, d2, (float)min_planarity, Q);
    HIPCHK(hipGetLastError());
    double *h_st;
    CHK(corr_alive_stats(c, nullptr, &h_st));
    if (n_alive_out) *n_alive_out = (int64_t)h_st[4];
    return SICP_OK;
}

SICP_EXPORT int sicp_corr_reject_distances(sicp_ctx *c, double *median_out, double *mad_out, int64_t *n_alive_out)
{
    CHK(check_corr(c));
    HIPCHK(hipSetDevice(c->device));
    const long Q = c->Q;
    // the selection kernels read the candidates' mask and write the survivors' into distinct buffers
    HIPCHK(hipMemcpyAsync(c->flag.p, c->keep.p, (size_t)Q, hipMemcpyDeviceToDevice, c->stream));
    double *h_st = c->h_small + 160;
    {
        Timed t(c, SICP_K_SELECT);
        if (Q > REJECT_MAX_Q) {
            const double seq = (double)(++c->solve_seq);
            CHK(reject_select(c, Q, h_st, seq, nullptr));
            HIPCHK(hipGetLastError());
            CHK(wait_ticket(c, h_st + 15, seq));
            if (h_st[0] < 0.0) return barrier_timed_out(c);
        } else {
            launch_reject(c->stream, c->dist.p, c->flag.p, Q, c->keep.p, c->small.p);
            CHK(corr_alive_stats(c, c->small.p, &h_st));
        }
    }
    if (median_out) *median_out = h_st[1];
    if (mad_out) *mad_out = h_st[2];
    if (n_alive_out) *n_alive_out = (int64_t)h_st[3];
    return SICP_OK;
}

SICP_EXPORT int sicp_estimate_parameters(sicp_ctx *c, const sicp_iter_params *P, const double *pc2_xyz, sicp_iter_result *R)
{
    if (!P || !R) return fail(SICP_ERR_INVALID, "null argument");
    CHK(check_corr(c));
    for (int j = 0; j < 6; ++j)
        if (std::isnan(P->obs_weight[j]) || P->obs_weight[j] < 0) return fail(SICP_ERR_INVALID, "obs_weight[%d] must be >= 0", j);
    HIPCHK(hipSetDevice(c->device));
    const long Q = c->Q;
    std::memset(R, 0, sizeof *R);
    if (pc2_xyz) {
        HIPCHK(hipMemcpyAsync(c->m_p2.p, pc2_xyz, (size_t)3 * Q * sizeof(double), hipMemcpyDefault, c->stream));
        c->have_prev_match = false;                           // no longer points of the searched cloud: not a search bound
    }
    double *h_st;
    CHK(corr_alive_stats(c, nullptr, &h_st));
    R->n_queries = Q;
    R->n_kept = (int64_t)h_st[4];
    R->n_planar = R->n_kept;
    R->median = R->mad = std::numeric_limits<double>::quiet_NaN();
    R->dist_mean = h_st[5]; R->dist_std = h_st[6];
    c->resid_slot = 0; c->resid_sharded = false;
    c->have_last_ne = false;
    std::memcpy(R->x, P->x, sizeof R->x);
    if (R->n_kept < 6) return too_few((long long)R->n_kept);
    CHK(host_lm_solve(c, P, R));
    c->have_iter = true;
    return SICP_OK;
}

SICP_EXPORT int sicp_params_to_H(const double x[6], double H_out[16])
{
    if (!x || !H_out) return fail(SICP_ERR_INVALID, "null argument");
    params_to_H12(x, H_out);
    H_out[12] = 0; H_out[13] = 0; H_out[14] = 0; H_out[15] = 1;
    return SICP_OK;
}

SICP_EXPORT int sicp_set_exchange(sicp_ctx *c, sicp_exchange_fn fn, void *user, int rank, int world, int gn_shard)
{
    if (!c) return fail(SICP_ERR_INVALID, "null ctx");
    if (world < 1 || rank < 0 || rank >= world) return fail(SICP_ERR_INVALID, "bad rank/world");
    if (world > 1 && !fn) return fail(SICP_ERR_INVALID, "world > 1 needs an exchange callback");
    c->comm_active = false;                                // a callback replaces the library's own communicator (which stays parked)
    c->xfn = fn; c->xuser = user; c->rank = rank; c->world = world; c->gn_shard = gn_shard ? 1 : 0;
    return SICP_OK;
}

SICP_EXPORT int sicp_comm_unique_id(void *id128)
{
    if (!id128) return fail(SICP_ERR_INVALID, "null argument");
    Rccl *R = rccl();
    if (!R) return fail(SICP_ERR_EXCHANGE, "librccl could not be loaded: %s", rccl_state().why.c_str());
    ncclUniqueId id;
    const ncclResult_t r = R->GetUniqueId(&id);
    if (r != ncclSuccess) return fail(SICP_ERR_EXCHANGE, "ncclGetUniqueId failed: %s", R->GetErrorString(r));
    static_assert(sizeof id == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(id128, &id, sizeof id);
    return SICP_OK;
}

SICP_EXPORT int sicp_comm_destroy(sicp_ctx *c)
{
    if (!c) return fail(SICP_ERR_INVALID, "null ctx");
    if (c->comm) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        (void)rccl()->CommDestroy(c->comm);
        c->comm = nullptr;
    }
    c->comm_active = false;
    if (!c->xfn) { c->rank = 0; c->world = 1; c->gn_shard = 0; }
    return SICP_OK;
}

namespace {
// ncclCommInitRank is a rendezvous: it returns when EVERY rank has called it.  A rank that never does (it crashed, it took
// another code path) would leave the callers blocked for good, so the call runs on a helper thread and the caller waits for it
// with a deadline; on a timeout the helper stays behind (there is no handle to abort yet) and the caller reports an error --
// simpleicp_amd/dist.py then sends every rank to the torch.distributed callback exchange together.
struct CommInit {
    std::mutex m; std::condition_variable cv;
    bool done = false;
    ncclResult_t r = ncclSuccess;
    ncclComm_t comm = nullptr;
    bool abandoned = false;        // the caller's deadline passed: whoever gets a communicator now must give it up
};
}  // namespace

SICP_EXPORT int sicp_comm_init(sicp_ctx *c, const void *id128, int rank, int world, int gn_shard)
{
    if (!c || !id128) return fail(SICP_ERR_INVALID, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(SICP_ERR_INVALID, "bad rank/world");
    Rccl *R = rccl();
    if (!R) return fail(SICP_ERR_EXCHANGE, "librccl could not be loaded: %s", rccl_state().why.c_str());
    CHK(sicp_comm_destroy(c));
    HIPCHK(hipSetDevice(c->device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    double timeout_s = 60.0;
    if (const char *e = std::getenv("SICP_COMM_TIMEOUT_S")) { const double v = std::atof(e); if (v > 0) timeout_s = v; }
    auto job = std::make_shared<CommInit>();
    const int device = c->device;
    std::thread([job, R, id, rank, world, device] {
        (void)hipSetDevice(device);
        ncclComm_t comm = nullptr;
        const ncclResult_t r = R->CommInitRank(&comm, world, id, rank);
        std::lock_guard<std::mutex> g(job->m);
        job->r = r; job->comm = comm; job->done = true;
        if (job->abandoned && r == ncclSuccess && comm) { (void)R->CommAbort(comm); job->comm = nullptr; }    // nobody is waiting any more
        job->cv.notify_all();
    }).detach();
    {
        std::unique_lock<std::mutex> g(job->m);
        if (!job->cv.wait_for(g, std::chrono::duration<double>(timeout_s), [&] { return job->done; })) {
            job->abandoned = true;                                // (under the lock: the helper aborts what it gets, should it ever return)
            return fail(SICP_ERR_EXCHANGE, "ncclCommInitRank did not return within %.0f s (rank %d of %d): not every rank joined "
                                           "(SICP_COMM_TIMEOUT_S)", timeout_s, rank, world);
        }
        if (job->r != ncclSuccess) return fail(SICP_ERR_EXCHANGE, "ncclCommInitRank failed: %s", R->GetErrorString(job->r));
        c->comm = job->comm;
    }
    // what the communicator says about itself must be what the caller said (a mixed-up id would pair the wrong processes)
    int n = 0, r = -1;
    if (R->CommCount(c->comm, &n) != ncclSuccess || R->CommUserRank(c->comm, &r) != ncclSuccess || n != world || r != rank) {
        (void)R->CommAbort(c->comm); c->comm = nullptr;
        return fail(SICP_ERR_EXCHANGE, "RCCL communicator reports rank %d of %d, expected %d of %d", r, n, rank, world);
    }
    c->comm_rank = r; c->comm_world = n;
    // handshake: one small all-gather on the ctx's stream, awaited with a deadline -- the first collective is where a transport
    // problem (a link that does not come up, a peer that cannot be mapped) shows, and it must show here, not inside a run
    CHK(c->x_send.reserve(8)); CHK(c->x_recv.reserve((size_t)8 * world));
    std::vector<double> h((size_t)8 * world, -1.0);
    for (int j = 0; j < 8; ++j) h[j] = 1000.0 * rank + j;
    HIPCHK(hipMemcpyAsync(c->x_send.p, h.data(), 8 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    c->comm_active = true;
    int rc = all_gather_f64(c, c->x_send.p, c->x_recv.p, 8);
    if (rc == SICP_OK) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) { rc = fail(SICP_ERR_HIP, "hipStreamQuery: %s", hipGetErrorString(q)); break; }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
                rc = fail(SICP_ERR_EXCHANGE, "the first RCCL all-gather did not complete within %.0f s (rank %d of %d)", timeout_s, rank, world);
                break;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
    }
    if (rc == SICP_OK) {
        if (hipMemcpy(h.data(), c->x_recv.p, (size_t)8 * world * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(SICP_ERR_HIP, "hipMemcpy after the handshake failed");
        for (int k = 0; rc == SICP_OK && k < world; ++k)
            for (int j = 0; j < 8; ++j)
                if (h[(size_t)8 * k + j] != 1000.0 * k + j) { rc = fail(SICP_ERR_EXCHANGE, "RCCL handshake: slot %d holds %g, not rank %d's words", k, h[(size_t)8 * k + j], k); break; }
    }
    if (rc != SICP_OK) {
        (void)R->CommAbort(c->comm); c->comm = nullptr; c->comm_active = false;
        return rc;
    }
    c->xfn = nullptr; c->xuser = nullptr;
    c->rank = rank; c->world = world; c->gn_shard = gn_shard ? 1 : 0;
    return SICP_OK;
}

// The communicator stays with the ctx between runs; a run switches its use on, the end of the run off (a standalone
// PointCloud operator on the same ctx must not issue a collective the other ranks never join).
SICP_EXPORT int sicp_comm_activate(sicp_ctx *c, int on, int gn_shard)
{
    if (!c) return fail(SICP_ERR_INVALID, "null ctx");
    if (on && !c->comm) return fail(SICP_ERR_INVALID, "no communicator: call sicp_comm_init first");
    if (on) {
        c->xfn = nullptr; c->xuser = nullptr;
        c->comm_active = true; c->rank = c->comm_rank; c->world = c->comm_world; c->gn_shard = gn_shard ? 1 : 0;
    } else {
        c->comm_active = false;
        if (!c->xfn) { c->rank = 0; c->world = 1; c->gn_shard = 0; }
    }
    return SICP_OK;
}

SICP_EXPORT int sicp_comm_info(sicp_ctx *c, int out[6])
{
    if (!c || !out) return fail(SICP_ERR_INVALID, "null argument");
    out[0] = c->xfn ? 1 : (c->comm && c->comm_active) ? 2 : 0;          // 0 none, 1 host callback, 2 the library's RCCL communicator
    out[1] = c->world; out[2] = c->rank; out[3] = c->partition; out[4] = c->gn_shard;
    out[5] = c->comm ? 1 : 0;                                            // a communicator exists (active or parked)
    if (out[0] == 2) {
        // as RCCL itself counts them, not as the caller declared them
        int n = 0, r = -1;
        if (rccl()->CommCount(c->comm, &n) != ncclSuccess || rccl()->CommUserRank(c->comm, &r) != ncclSuccess)
            return fail(SICP_ERR_EXCHANGE, "ncclCommCount / ncclCommUserRank failed");
        out[1] = n; out[2] = r;
    }
    return SICP_OK;
}

SICP_EXPORT int sicp_device_memory(sicp_ctx *c, int64_t *free_out, int64_t *total_out)
{
    if (!c || !free_out || !total_out) return fail(SICP_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(c->device));
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    *free_out = (int64_t)f; *total_out = (int64_t)t;
    return SICP_OK;
}

SICP_EXPORT int sicp_set_partition(sicp_ctx *c, int mode)
{
    if (!c) return fail(SICP_ERR_INVALID, "null ctx");
    if (mode != SICP_PART_CLOUD && mode != SICP_PART_QUERIES) return fail(SICP_ERR_INVALID, "mode must be SICP_PART_CLOUD or SICP_PART_QUERIES");
    c->partition = mode;
    return SICP_OK;
}

SICP_EXPORT int sicp_lexmin_gathered(sicp_ctx *c, const double *gathered, int world, int64_t Q, double *d2_out,
                                     int64_t *idx_out, double *xyz_out)
{
    if (!c || !gathered || !d2_out || !idx_out || world < 1 || Q < 1) return fail(SICP_ERR_INVALID, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    DevBuf<double> g, d2, xyz; DevBuf<int64_t> idx;
    int rc = g.reserve((size_t)5 * Q * world);
    if (rc == SICP_OK) rc = d2.reserve(Q);
    if (rc == SICP_OK) rc = xyz.reserve((size_t)3 * Q);
    if (rc == SICP_OK) rc = idx.reserve(Q);
    auto body = [&]() -> int {
        HIPCHK(hipMemcpyAsync(g.p, gathered, (size_t)5 * Q * world * sizeof(double), hipMemcpyDefault, c->stream));
        launch_lexmin_gathered(c->stream, g.p, world, Q, d2.p, idx.p, xyz.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(d2_out, d2.p, (size_t)Q * sizeof(double), hipMemcpyDefault, c->stream));
        HIPCHK(hipMemcpyAsync(idx_out, idx.p, (size_t)Q * sizeof(int64_t), hipMemcpyDefault, c->stream));
        if (xyz_out) HIPCHK(hipMemcpyAsync(xyz_out, xyz.p, (size_t)3 * Q * sizeof(double), hipMemcpyDefault, c->stream));
        return sync(c);
    };
    if (rc == SICP_OK) rc = body();
    (void)hipStreamSynchronize(c->stream);
    g.release(); d2.release(); xyz.release(); idx.release();
    return rc;
}

SICP_EXPORT int sicp_ctx_stream(sicp_ctx *c, void **stream_out)
{
    if (!c || !stream_out) return fail(SICP_ERR_INVALID, "null argument");
    *stream_out = (void *)c->stream;
    return SICP_OK;
}

SICP_EXPORT int sicp_timing_enable(sicp_ctx *c, int on)
{
    if (!c) return fail(SICP_ERR_INVALID, "null ctx");
    c->timing = on != 0;
    c->count_work = on == 2;
    return SICP_OK;
}
SICP_EXPORT int sicp_last_match_kernel(sicp_ctx *c, int *kind_out)
{
    if (!c || !kind_out) return fail(SICP_ERR_INVALID, "null argument");
    *kind_out = c->last_match_kernel;
    return SICP_OK;
}
SICP_EXPORT int sicp_match_work(sicp_ctx *c, uint64_t out3[3])
{
    if (!c || !out3) return fail(SICP_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(out3, c->match_work.p, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    return sync(c);
}
SICP_EXPORT int sicp_match_deferred(sicp_ctx *c, uint64_t *out)
{
    if (!c || !out) return fail(SICP_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(out, c->match_work.p + 3, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    return sync(c);
}
SICP_EXPORT int sicp_tail_cycles(sicp_ctx *c, double out5[5])
{
    if (!c || !out5) return fail(SICP_ERR_INVALID, "null argument");
    std::memcpy(out5, c->last_tail_cycles, sizeof c->last_tail_cycles);
    return SICP_OK;
}
SICP_EXPORT int sicp_knn_work(sicp_ctx *c, uint64_t out4[4])
{
    if (!c || !out4) return fail(SICP_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(out4, c->match_work.p + 4, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    return sync(c);
}
SICP_EXPORT int sicp_timing_reset(sicp_ctx *c)
{
    if (!c) return fail(SICP_ERR_INVALID, "null ctx");
    if (!c->pending.empty()) CHK(sync(c));
    HIPCHK(hipMemsetAsync(c->match_work.p, 0, 8 * sizeof(unsigned long long), c->stream));
    for (int i = 0; i < SICP_K_COUNT; ++i) { c->t_ms[i] = 0; c->t_n[i] = 0; }
    return SICP_OK;
}
SICP_EXPORT int sicp_timing_get(sicp_ctx *c, int kernel, double *total_ms_out, int64_t *launches_out)
{
    if (!c || kernel < 0 || kernel >= SICP_K_COUNT) return fail(SICP_ERR_INVALID, "bad arguments");
    if (!c->pending.empty()) CHK(sync(c));          // fold in whatever is still in flight
    if (total_ms_out) *total_ms_out = c->t_ms[kernel];
    if (launches_out) *launches_out = c->t_n[kernel];
    return SICP_OK;
}
