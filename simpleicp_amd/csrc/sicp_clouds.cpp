// sicp_clouds.cpp -- clouds in HBM: uploads, downloads, transform, the planarity column, and everything that bins a cloud (the grid, its
// subsample, its coarse twin, its companions, the orders queries are taken in).  Split from sicp_api.cpp (round 5).
#include "sicp_host.h"

#include <emmintrin.h>

namespace sicph {

// icp_queries: how many queries per launch the MATCH of an ICP run is about to send (its caller passes the rank's own count); -1: any
// other search (sicp_knn, sicp_select_in_range, normals, the operators) -- those take the grid as it is and never rebuild one.
int grid_build(sicp_ctx *c, int slot, long icp_queries)
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
    if (cl.grid.valid && cl.grid.target_used > 0 && cl.grid.target_used != target) {
        cl.grid.valid = false;
        // (its coarse twin is 8 x the OLD cell size -- k_grid_nn switches to it at r > 4 h and assumes 8 x the new h -- and may no longer
        // be wanted at all: rebuilt on demand, released when the new grid is not nonuniform: ADVICE r5)
        cl.coarse_grid.valid = false;
    }
    cl.grid.target_used = target;
    CHK(grid_build_arrays(c, cl, cl.x(), cl.y(), cl.z(), cl.n, cl.grid, target));
    if (!cl.grid.nonuniform && !cl.coarse_grid.valid && cl.coarse_grid.rec.p) {      // a second 32-byte-per-point copy nobody reads
        cl.coarse_grid.rec.release(); cl.coarse_grid.cell_start.release();
    }
    return SICP_OK;
}

// the cloud's subsample (every SUB_STRIDE-th point) and its grid
constexpr long SUB_STRIDE = 64;       // (measured 64 / 16 / 8: 152 / 128 / 122 candidates per query of the cold search -- the subsample's own search pays the difference back;
                                      //  again in round 6 at the headline workload, 64 / 32 / 16: the cold first match 27.4 / 27.1 / 27.8 us -- while the clouds are metres apart the answer itself is far)
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
    // (a cloud whose POINTS crowd a few cells -- see the point-weighted occupancy below -- may take a larger table than its size alone
    // earns: c->grid_cap_nonuniform_log2)
    long cap_nu = std::max(cap, 1L << c->grid_cap_nonuniform_log2);
    {
        // ... and never more than the device can spare: a cell costs 12 bytes of table + build scratch (+ 8 of tight boxes when a far
        // search asks for them) -- a quarter of what is free, at least 2^22 cells (ranks sharing a device, smaller devices)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const long fit = (long)(free_b / 4 / 20);
            cap = std::max<long>(1L << 22, std::min(cap, fit));
            cap_nu = std::max(cap, std::min(cap_nu, fit));
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
    double pw_before = 0.0, h_before = 0.0;             // point-weighted occupancy and cell size before the last shrink
    int shrinks = 0;
    bool pw_settled = false;
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
        // (not once the point-weighted rule has settled on a size: shrinking again would undo its take-back -- ADVICE r5)
        if (!probed && !pw_settled && gr.avg_per_cell > 3 * target && attempt < 4 && ncells < cap / 2) { h *= std::sqrt(target / gr.avg_per_cell); continue; }
        // the points' own view: shrink until an average point shares its cell with a few times the target (occupancy ~ h^2 on a surface);
        // not below the table's limit (a binning that was capped stands), at most six rounds
        if (c->grid_pointwise && !pw_settled) {
            // (the last shrink bought next to nothing: what shares cells is COINCIDENT points, not density -- no cell size separates
            // those, and smaller cells only make every ball span more rows: back to the size before, and that stands)
            // (the table's limit stopped the shrink while an average point still shares its cell with many: such a cloud gets the larger table)
            if (capped && cap < cap_nu && pw > 4.0 * target && attempt < 8) {
                cap = cap_nu;
                const double f = std::sqrt(2.0 * target / pw);
                pw_before = pw; h_before = h; ++shrinks;
                h *= f < 0.3 ? 0.3 : (f > 0.8 ? 0.8 : f);
                probed = false; gr.nonuniform = true;
                continue;
            }
            if (pw_before > 0.0 && pw > 0.7 * pw_before) { h = h_before; pw_settled = true; gr.nonuniform = shrinks > 1; probed = false; continue; }
            if (pw > 4.0 * target && !capped && attempt < 8) {
                const double f = std::sqrt(2.0 * target / pw);
                pw_before = pw; h_before = h; ++shrinks;
                h *= f < 0.3 ? 0.3 : (f > 0.8 ? 0.8 : f);
                probed = false;                          // (the window's evidence is overruled: measure the plain occupancy too from here on)
                gr.nonuniform = true;
                continue;
            }
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


}  // namespace sicph

namespace sicph {
// An upload that runs behind its caller (sicp_cloud_upload_start): joined -- and its verdict delivered -- by the first call that names
// the slot (check_slot), by sicp_cloud_upload_wait, by the next background upload and by sicp_ctx_destroy.
int upload_join(sicp_ctx *c, int slot)
{
    BgUpload &b = c->bg[slot];
    if (!b.active) return SICP_OK;
    if (b.th.joinable()) b.th.join();
    b.active = false;
    if (b.rc != SICP_OK) {
        c->cloud[slot].n = 0;                          // (nothing usable arrived)
        return fail(b.rc, "%s", b.err.c_str());
    }
    return SICP_OK;
}

// shared by the upload flavours: validates, sizes the padded SoA arrays
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
    // (a new movable cloud: earlier matches are not its points -- neither the by-query ones nor those the filtered search keeps by slot,
    // which an operator-route match in between would not rebuild)
    if (slot == SICP_MOV) { c->have_prev_match = false; c->slot_cnt = -1; }
    CHK(cl.xyz.reserve((size_t)3 * cl.npad));
    return SICP_OK;
}

// ... and finishes: ONE statistics pass gives the largest norm (rounding-error bounds of the filtered / grid searches;
// a non-finite cloud is refused like cKDTree would) and the bounding box the grid build starts from.
// (s, d_st, h_st: the stream and the 7-word scratch, device + pinned -- the ctx's own, or a background upload's)
int cloud_stats_on(sicp_ctx *c, int slot, hipStream_t s, unsigned long long *d_st, double *h_st)
{
    Cloud &cl = c->cloud[slot];
    unsigned long long h_init[7] = {~0ull, ~0ull, ~0ull, 0ull, 0ull, 0ull, 0ull};
    HIPCHK(hipMemcpyAsync(d_st, h_init, sizeof h_init, hipMemcpyHostToDevice, s));
    launch_cloud_stats(s, cl.x(), cl.y(), cl.z(), cl.n, d_st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h_st, d_st, 7 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (s == c->stream) CHK(sync(c)); else HIPCHK(hipStreamSynchronize(s));
    unsigned long long hk[7]; std::memcpy(hk, h_st, sizeof hk);
    double nn; std::memcpy(&nn, &hk[6], sizeof nn);
    if (!std::isfinite(nn)) {
        cl.n = 0;                                    // like cKDTree (pointcloud.py:161,185): no search structure over NaN / inf
        return fail(SICP_ERR_INVALID, "cloud has non-finite coordinates (NaN / inf, or |p|^2 overflows a double)");
    }
    cl.rmax = std::sqrt(nn) * (1.0 + 1e-12);
    for (int a = 0; a < 3; ++a) { cl.bb_lo[a] = key_to_double(hk[a]); cl.bb_hi[a] = key_to_double(hk[3 + a]); }
    return SICP_OK;
}
int cloud_stats(sicp_ctx *c, int slot)
{
    if (slot == SICP_MOV) { c->have_prev_match = false; c->slot_cnt = -1; }       // (sicp_cloud_transform comes here too: the points moved)
    return cloud_stats_on(c, slot, c->stream, (unsigned long long *)(c->small.p + 40), c->h_small + 40);
}
int upload_end(sicp_ctx *c, int slot) { return cloud_stats(c, slot); }

// Small and medium clouds go through the library's own pinned double buffer: a DMA straight out of the caller's pageable array makes
// the runtime pin that address range first, and for a range it has not seen before that costs 10-20 ms whatever the size (measured:
// Webots' two 1 MB uploads took 13-22 ms on fresh arrays, 0.2 ms on recycled addresses).  A host copy into pinned memory costs
// ~0.1 ms per MB and always the same.  Rows are transposed (or columns copied) by the host on the way, chunk ch + 1 while chunk ch
// is on the link.  Above UPLOAD_STAGED_MAX points the pinning is the smaller price.
constexpr int64_t UPLOAD_STAGED_MAX = 1 << 19;       // (one chunk: ~1.5 ms of host copy at most)
constexpr long DL_CH = 1L << 19;          // the download's chunks: 512 Ki points = 12 MiB
constexpr int DL_RING = 4;                // ... in flight: the pinned block is DL_RING x 3 x DL_CH doubles (48 MiB), on first use
int upload_staged(sicp_ctx *c, Cloud &cl, const double *xyz, const double *x, const double *y, const double *z, int64_t n)
{
    const long CH = DL_CH;
    if (!c->h_dl) HIPCHK(hipHostMalloc((void **)&c->h_dl, (size_t)DL_RING * 3 * DL_CH * sizeof(double), hipHostMallocDefault));      // (the download's ring; two of its buffers serve here)
    for (auto &e : c->dl_ev) if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
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

// a large cloud's way to the device: straight out of the caller's arrays (rows through an AoS staging block on the device, columns into
// place), on stream s
int upload_direct(sicp_ctx *c, Cloud &cl, const double *xyz, const double *x, const double *y, const double *z, int64_t n, hipStream_t s,
                  DevBuf<double> &stage)
{
    if (xyz) {
        CHK(stage.reserve((size_t)3 * n));
        HIPCHK(hipMemcpyAsync(stage.p, xyz, (size_t)3 * n * sizeof(double), hipMemcpyDefault, s));
        launch_aos_to_soa(s, stage.p, n, cl.npad, cl.x(), cl.y(), cl.z());
    } else {
        // the device layout is column-wise already: three copies straight into place, no staging, no transpose
        HIPCHK(hipMemcpyAsync(cl.x(), x, (size_t)n * sizeof(double), hipMemcpyDefault, s));
        HIPCHK(hipMemcpyAsync(cl.y(), y, (size_t)n * sizeof(double), hipMemcpyDefault, s));
        HIPCHK(hipMemcpyAsync(cl.z(), z, (size_t)n * sizeof(double), hipMemcpyDefault, s));
        launch_pad_fill(s, cl.x(), cl.y(), cl.z(), n, cl.npad);
    }
    HIPCHK(hipGetLastError());
    return SICP_OK;
}

int upload_now(sicp_ctx *c, int slot, const double *xyz, const double *x, const double *y, const double *z, int64_t n, int64_t index_base)
{
    CHK(upload_begin(c, slot, n, index_base));
    Cloud &cl = c->cloud[slot];
    if (n <= UPLOAD_STAGED_MAX && c->upload_staged) CHK(upload_staged(c, cl, xyz, x, y, z, n));
    else CHK(upload_direct(c, cl, xyz, x, y, z, n, c->stream, c->stage));
    return upload_end(c, slot);
}

// The same upload on a helper thread with a stream, a staging block and statistics scratch of its own: a DMA out of pageable memory
// holds the calling thread until it is done (4.5 ms per 10 M points, profiles/r6/pageable_async.txt), and run() has the fixed cloud's
// grid and normals to build meanwhile.  Small clouds (the staged road shares the download's pinned ring) are uploaded on the spot.
int upload_start(sicp_ctx *c, int slot, const double *xyz, const double *x, const double *y, const double *z, int64_t n, int64_t index_base)
{
    if (!c) return fail(SICP_ERR_INVALID, "null ctx");
    if (slot != SICP_FIX && slot != SICP_MOV) return fail(SICP_ERR_INVALID, "slot must be SICP_FIX or SICP_MOV");
    // one at a time (they share the stream and the scratch): the other slot's helper is waited for -- its verdict stays with ITS slot
    if (c->bg[1 - slot].active && c->bg[1 - slot].th.joinable()) c->bg[1 - slot].th.join();
    if (n <= UPLOAD_STAGED_MAX && c->upload_staged) return upload_now(c, slot, xyz, x, y, z, n, index_base);
    CHK(upload_begin(c, slot, n, index_base));
    if (!c->copy_stream) HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (!c->h_bg) HIPCHK(hipHostMalloc((void **)&c->h_bg, 8 * sizeof(double), hipHostMallocDefault));
    CHK(c->bg_small.reserve(8));
    if (xyz) CHK(c->stage_bg.reserve((size_t)3 * n));                    // (allocations stay on the calling thread)
    BgUpload &b = c->bg[slot];
    b.rc = SICP_OK; b.err.clear(); b.active = true;
    try {
        b.th = std::thread([c, slot, xyz, x, y, z, n, &b] {
            auto body = [&]() -> int {
                HIPCHK(hipSetDevice(c->device));
                CHK(upload_direct(c, c->cloud[slot], xyz, x, y, z, n, c->copy_stream, c->stage_bg));
                return cloud_stats_on(c, slot, c->copy_stream, (unsigned long long *)c->bg_small.p, c->h_bg);
            };
            b.rc = body();
            if (b.rc != SICP_OK) { b.err = sicp_last_error(); (void)hipStreamSynchronize(c->copy_stream); }
        });
    } catch (...) {
        // no thread to be had: upload here and now (an exception must not cross the C ABI)
        b.active = false;
        CHK(upload_direct(c, c->cloud[slot], xyz, x, y, z, n, c->stream, c->stage));
        return upload_end(c, slot);
    }
    return SICP_OK;
}
}  // namespace sicph

SICP_EXPORT int sicp_cloud_upload(sicp_ctx *c, int slot, const double *xyz, int64_t n, int64_t index_base)
{
    if (!xyz) return fail(SICP_ERR_INVALID, "xyz is null");
    return upload_now(c, slot, xyz, nullptr, nullptr, nullptr, n, index_base);
}

SICP_EXPORT int sicp_cloud_upload_columns(sicp_ctx *c, int slot, const double *x, const double *y, const double *z, int64_t n,
                                          int64_t index_base)
{
    if (!x || !y || !z) return fail(SICP_ERR_INVALID, "x / y / z is null");
    return upload_now(c, slot, nullptr, x, y, z, n, index_base);
}

SICP_EXPORT int sicp_cloud_upload_start(sicp_ctx *c, int slot, const double *xyz, const double *x, const double *y, const double *z,
                                        int64_t n, int64_t index_base)
{
    if (!xyz && !(x && y && z)) return fail(SICP_ERR_INVALID, "xyz, or x / y / z");
    if (xyz && (x || y || z)) return fail(SICP_ERR_INVALID, "xyz OR x / y / z, not both");
    hipPointerAttribute_t at;
    const void *first = xyz ? (const void *)xyz : (const void *)x;
    if (hipPointerGetAttributes(&at, first) == hipSuccess && at.type == hipMemoryTypeDevice)
        return upload_now(c, slot, xyz, x, y, z, n, index_base);           // (device memory: nothing holds the caller up)
    (void)hipGetLastError();
    return upload_start(c, slot, xyz, x, y, z, n, index_base);
}

SICP_EXPORT int sicp_cloud_upload_wait(sicp_ctx *c, int slot)
{
    return check_slot(c, slot, false);                 // (joins; hands over the upload's own verdict)
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
// Here the columns are pulled chunk by chunk into a ring of pinned buffers at link speed while host threads fan the chunks that have
// landed out into both destinations (the row form is a transpose the host does from the pinned chunk: nothing crosses the link
// twice).  Round 6: a ring of DL_RING chunks and no meeting of the threads per chunk -- worker t copies ITS rows of every chunk as
// soon as the chunk is there, and a slow worker only holds up the reuse of a buffer four chunks later.  The two-buffer form met all
// 16 threads at every chunk and slept 50 us at a time in between: 10.5 ms per 10 M points of which 0.4 were spent waiting for the
// link (profiles/r6/download_parts_before.txt).  And the device packs the chunks first (x | y | z of a chunk back to back, 0.1 ms):
// one 12 MiB piece per chunk for the copy engine instead of three small ones (45 -> 55 GB/s, profiles/r6/d2h_rate.txt).
namespace sicph {
// Streaming copy: count doubles from src to dst with non-temporal stores.  The destination is 240 + 240 MB that nobody reads soon: plain
// stores would first pull every line in (480 MB of reads the fan-out does not need) and push the pinned ring -- which the copy
// engine is writing -- out of the caches the threads read it from; with them the link ran at 44 GB/s beside the threads, 52 alone.
static inline void stream_copy(double *dst, const double *src, long count)
{
    long i = 0;
    if (((uintptr_t)dst & 15) && count > 0) { _mm_stream_si64((long long *)dst, *(const long long *)src); i = 1; }
    for (; i + 1 < count; i += 2) _mm_stream_pd(dst + i, _mm_loadu_pd(src + i));
    if (i < count) _mm_stream_si64((long long *)(dst + i), *(const long long *)(src + i));
}

// rows [a, e) of a pinned chunk (columns CH apart) into the caller's columns and / or rows starting at point `lo`
static inline void fan_out(const double *b, long CH, long lo, long a, long e, double *xyz_out, double *x_out, double *y_out, double *z_out)
{
    if (e <= a) return;
    if (x_out) {
        stream_copy(x_out + lo + a, b + a, e - a);
        stream_copy(y_out + lo + a, b + CH + a, e - a);
        stream_copy(z_out + lo + a, b + 2 * CH + a, e - a);
    }
    if (xyz_out) {
        long long *o = (long long *)(xyz_out + 3 * (lo + a));
        const long long *bx = (const long long *)b, *by = (const long long *)(b + CH), *bz = (const long long *)(b + 2 * CH);
        for (long i = a; i < e; ++i) { _mm_stream_si64(o, bx[i]); _mm_stream_si64(o + 1, by[i]); _mm_stream_si64(o + 2, bz[i]); o += 3; }
    }
    _mm_sfence();                                       // (the streamed stores are visible before the caller is told the chunk is done)
}
}  // namespace sicph

SICP_EXPORT int sicp_cloud_download_both(sicp_ctx *c, int slot, double *xyz_out, double *x_out, double *y_out, double *z_out)
{
    CHK(check_slot(c, slot, true));
    if (!xyz_out && !(x_out && y_out && z_out)) return fail(SICP_ERR_INVALID, "no destination");
    if ((x_out || y_out || z_out) && !(x_out && y_out && z_out)) return fail(SICP_ERR_INVALID, "x_out / y_out / z_out: all or none");
    HIPCHK(hipSetDevice(c->device));
    Cloud &cl = c->cloud[slot];
    const long n = cl.n, CH = std::min<long>(DL_CH, round_up(n, 1024));      // (a small cloud is one chunk of its own size)
    if (!c->h_dl) HIPCHK(hipHostMalloc((void **)&c->h_dl, (size_t)DL_RING * 3 * DL_CH * sizeof(double), hipHostMallocDefault));
    for (auto &e : c->dl_ev) if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const long nchunks = (n + CH - 1) / CH;
    // host threads that fan the chunks out: as many as this process may actually run on (cgroup / affinity limits, not the machine's
    // core count), at most c->dl_threads, and none for clouds that are one chunk's worth of microseconds
    unsigned T = 1;
    {
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0) T = (unsigned)CPU_COUNT(&set);
        else T = std::thread::hardware_concurrency();
        const unsigned most = (unsigned)c->dl_threads;
        T = T < 2 ? 1 : (T > most ? most : T);
        if (n < (1L << 16)) T = 1;
    }
    // waiting: spin while the wait is short (a chunk is ~0.12 ms of link time), then yield, and sleep only when nothing has moved for
    // milliseconds -- a spinner must not starve the thread it waits for in a one-CPU container
    auto wait_until = [](auto &&cond) {
        for (long spins = 0; !cond(); ++spins) {
            if (spins < 4096) __builtin_ia32_pause();
            else if (spins < 4096 + 4096) std::this_thread::yield();
            else std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    };
    CHK(c->stage.reserve((size_t)3 * nchunks * CH));
    launch_pack_chunks(c->stream, cl.x(), cl.y(), cl.z(), n, CH, c->stage.p);
    HIPCHK(hipGetLastError());
    auto enqueue = [&](long ch) -> int {
        const long lo = ch * CH, m = std::min(CH, n - lo);
        double *b = c->h_dl + (size_t)(ch % DL_RING) * 3 * CH;
        HIPCHK(hipMemcpyAsync(b, c->stage.p + (size_t)3 * ch * CH, (size_t)(2 * CH + m) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipEventRecord(c->dl_ev[ch % DL_RING], c->stream));
        return SICP_OK;
    };
    // chunk `ready` and all before it are in their pinned buffers; parts[ch % DL_RING] counts the workers that are through with chunk ch
    // (it only ever grows: chunk ch is done when it reaches W * (ch / DL_RING + 1))
    unsigned W = T > 1 ? T - 1 : 1;                          // copying threads: with helpers, this thread only feeds the link
    std::atomic<long> ready{-1};
    std::atomic<long> parts[DL_RING];
    for (auto &p : parts) p.store(0);
    std::atomic<bool> quit{false};
    auto work = [&](unsigned t) {
        for (long ch = 0; ch < nchunks; ++ch) {
            wait_until([&] { return ready.load(std::memory_order_acquire) >= ch || quit.load(std::memory_order_relaxed); });
            if (quit.load(std::memory_order_relaxed)) return;
            const long lo = ch * CH, m = std::min(CH, n - lo);
            fan_out(c->h_dl + (size_t)(ch % DL_RING) * 3 * CH, CH, lo, m * t / W, m * (t + 1) / W, xyz_out, x_out, y_out, z_out);
            parts[ch % DL_RING].fetch_add(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    if (T > 1) {
        try {
            for (unsigned t = 0; t < W; ++t) pool.emplace_back(work, t);
        } catch (...) {
            // no thread to be had (resource limits): nothing has been copied yet -- send the ones that started home and do it alone
            // (an exception must not cross the C ABI)
            quit.store(true);
            for (auto &th : pool) th.join();
            pool.clear();
            quit.store(false);
            T = 1; W = 1;
        }
    }
    auto chunk_done = [&](long ch) { return parts[ch % DL_RING].load(std::memory_order_acquire) >= (long)W * (ch / DL_RING + 1); };
    int rc = SICP_OK;
    double t_workers = 0, t_link = 0;                        // SICP_SOLVE_TRACE=host: where this thread waited
    const double t_begin = c->host_trace ? wall_ms() : 0.0;
    long issued = 0;
    for (long ch = 0; ch < nchunks && rc == SICP_OK; ++ch) {
        // keep the link fed: every chunk whose buffer is free again (the first DL_RING have fresh ones)
        while (rc == SICP_OK && issued < nchunks && issued < ch + DL_RING && (issued < DL_RING || chunk_done(issued - DL_RING))) rc = enqueue(issued++);
        if (rc != SICP_OK) break;
        if (issued <= ch) {                                  // (only when the copying threads are DL_RING chunks behind)
            const double t0 = c->host_trace ? wall_ms() : 0.0;
            wait_until([&] { return chunk_done(ch - DL_RING); });
            if (c->host_trace) t_workers += wall_ms() - t0;
            --ch; continue;
        }
        const double t0 = c->host_trace ? wall_ms() : 0.0;
        if (hipEventSynchronize(c->dl_ev[ch % DL_RING]) != hipSuccess) { rc = fail(SICP_ERR_HIP, "hipEventSynchronize failed"); break; }
        if (c->host_trace) t_link += wall_ms() - t0;
        ready.store(ch, std::memory_order_release);
        if (pool.empty()) {                                  // alone: this thread copies the chunk out itself
            const long lo = ch * CH, m = std::min(CH, n - lo);
            fan_out(c->h_dl + (size_t)(ch % DL_RING) * 3 * CH, CH, lo, 0, m, xyz_out, x_out, y_out, z_out);
            parts[ch % DL_RING].fetch_add(1, std::memory_order_release);
        }
    }
    if (rc != SICP_OK) quit.store(true);
    const double t_join = c->host_trace ? wall_ms() : 0.0;
    for (auto &th : pool) th.join();
    if (c->host_trace)
        std::fprintf(stderr, "[sicp] download_both: %ld chunks, %u copying threads, %.2f ms: waited %.2f for the link, %.2f for a free buffer, %.2f for the last chunks\n",
                     nchunks, pool.empty() ? 1u : W, wall_ms() - t_begin, t_link, t_workers, wall_ms() - t_join);
    if (rc != SICP_OK) { (void)hipStreamSynchronize(c->stream); return rc; }
    return sync(c);
}

