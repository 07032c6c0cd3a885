// sicp_icp.cpp -- the ICP iteration: setup, the device-chained loop, the host-side solver road, state and uncertainties, and the
// operator-by-operator exports (CorrPts.*, estimate_parameters).  Split from sicp_api.cpp (round 5).
#include "sicp_host.h"

namespace sicph {

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


}  // namespace sicph

namespace sicph {

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


}  // namespace sicph

namespace sicph {

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


}  // namespace sicph

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
#ifndef SICP_TEST_REVERT_SLOT_FIX         // (tests/test_gpu_fuzz.py builds a variant WITHOUT this line to show that the call-sequence fuzz finds round 5's stale-slot bug)
    c->slot_cnt = -1;                  // (the filtered search's slot-ordered copies of the queries: other queries now)
#endif
    c->q_order_lo = -1; c->q_order_cnt = 0;
    c->hsel_run_launches = 0;
    c->sel_window_hits = 0; c->last_sel_rounds[0] = c->last_sel_rounds[1] = 0;
    c->last_xchg_form = 0; c->xchg_count = 0;
    c->reject_prior = false;
    return sync(c);
}

namespace sicph {

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
    A.window = c->tail_window ? 1 : 0;
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
                c->last_match_kernel = (cnt >= c->nn16_min_q && (!cl.grid.nonuniform || cnt >= filter_min_q(c, true))) ? 5 : 2;
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
                const bool many_q = cnt >= c->nn16_min_q && (!cl.grid.nonuniform || cnt >= filter_min_q(c, true));
                // far searches (a run's first iterations) trim their rows by the tight boxes of the cells; large query sets are
                // searched through the float32 filter (sicp_gridf.hip) when float32 can hold the cloud
                const bool boxes = c->use_boxes && cnt > 0;
                bool filt = many_q && c->nn16_filter != 0 && cnt > 0 && cnt >= filter_min_q(c, cl.grid.nonuniform);
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
                    // (a nonuniform cloud -- cells of hundreds of points next to empty ones -- hands most of its queries on as well: measured on the
                    // terrestrial stand-in at 1 M queries, 3.00 ms with the lean flavour first, 2.54 ms without)
                    const bool all_far = coarse || c->nn16_filter == 1 || cl.grid.nonuniform || !(last_move <= c->far_move * cl.grid.g.h);
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
                // ... except below ~65 536 queries, where the machine is not full: there the epilogue costs the search ~0.07 us per
                // 1 000 queries and k_postmatch's launch 5 us (profiles/r6: 10 000 correspondences, 63.6 -> 58.9 us per iteration)
                post_done = !c->collective() && (!many_q || cnt < 65536);
                // behind a cloud-shard exchange the winning lanes leave the exchange's packed record instead (no k_pack_best launch)
                // (query shards: the slim record, the matched index alone -- no k_pack_idx launch)
                const bool pack = c->collective() && !qshard && !exchange_by_keys(c, Q), pack_idx = c->collective() && qshard;
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
            if (qshard) { CHK(exchange_query_slices_idx(c, A, Q, packed)); post_done = true; c->last_xchg_form = 3; ++c->xchg_count; }      // (distances + verdicts formed by the unpack)
            else if (c->collective() && c->partition == SICP_PART_CLOUD) {
                CHK(c->x_send.reserve((size_t)5 * Q));
                const bool by_keys = exchange_by_keys(c, Q);
                if (by_keys) CHK(exchange_best_keys_chained(c, A, Q));                          // (many queries: all-reduces on 8-byte keys)
                else CHK(exchange_best_chained(c, A, Q, packed));                              // (... by the lexicographic minimum's kernel)
                // the filtered search keeps its bounds by slot and wrote THIS rank's winner there: make it the job-wide one
                if (c->last_match_kernel == 6 && c->slot_cnt == Q)
                    launch_slot_bounds(c->stream, c->q_slot.p, c->m_idx.p, c->m_p2.p, Q, c->p_slot.p);
                post_done = true;
                c->last_xchg_form = by_keys ? 2 : 1; ++c->xchg_count;
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
                    launch_reject(c->stream, c->dist.p, c->flag.p, Q, c->keep.p, c->small.p, c->icp_dev.p, c->small.p + 4,
                                  c->tail_window && c->reject_prior);
                    c->reject_prior = true;               // (c->small[0..3] now holds this iteration's statistics: the next launch's window)
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
        if (small_q) {
            std::memcpy(c->last_tail_cycles, o + 50, 5 * sizeof(double));
            c->last_sel_rounds[0] = (long)o[56]; c->last_sel_rounds[1] = (long)o[58];
            if (o[56] == 0.0 && o[58] == 0.0) ++c->sel_window_hits;
        }
        if (c->solve_trace && small_q)
            std::fprintf(stderr, "[tail] cycles: load+dist %.0f select %.0f (median %.0f in %.0f rounds, MAD %.0f in %.0f) keep %.0f lm %.0f "
                                 "(%lld evals %.0f, %lld steps, solves %.0f, accept %.0f) final %.0f\n",
                         o[50], o[51], o[55], o[56], o[57], o[58], o[52], o[53], (long long)R.ne_evals, o[59], (long long)R.lm_steps, o[60], o[62], o[54]);
        if (c->solve_trace && small_q && c->trace_sel)      // (a -DSICP_SEL_FINE_TRACE build: build.build_variant)
            std::fprintf(stderr, "[sel] median: atomics+barrier %.0f fold+barrier %.0f scan+pick %.0f (more rounds %.0f) gather+barrier %.0f rank %.0f | "
                                 "MAD: %.0f %.0f %.0f (%.0f) %.0f %.0f\n", o[38], o[39], o[40], o[41], o[42], o[43], o[44], o[45], o[46], o[47], o[48], o[49]);
        if (c->solve_trace && small_q && c->trace_eval)     // (a -DSICP_EVAL_FINE_TRACE build)
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

}  // namespace sicph

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
namespace sicph {

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

}  // namespace sicph

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
        c->slot_cnt = -1;                                     // (... nor are the bounds kept by slot: the operator route may set have_prev_match again)
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

