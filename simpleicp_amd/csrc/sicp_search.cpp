// sicp_search.cpp -- the drivers of the 1-NN and k-NN searches (which kernel, on which structure, with which bound) and the exports that are
// searches: sicp_knn, sicp_select_in_range, sicp_estimate_normals.  Split from sicp_api.cpp (round 5).
#include "sicp_host.h"

namespace sicph {

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


}  // namespace sicph

namespace sicph {

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
        if (Q >= c->nn16_min_q && Q >= filter_min_q(c, gr.nonuniform) && c->nn16_filter != 0 && Q < (1L << 31)) {
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
                const bool all_far = c->nn16_filter == 1 || gr.nonuniform;
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
                float *normals_out, float *planarity_out, bool *fused)
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

}  // namespace sicph

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

