// sicp_grid.hip -- pruned EXACT 1-NN / k-NN on a static uniform grid.
//
// The searched (movable) cloud never moves in its own frame, so it is binned ONCE per upload:
//   k_cloud_stats   bounding box + largest norm in one pass (one atomic per block and quantity)
//   cell size       from the bounding volume and, for surfaces / curves, from two histograms of a strided SAMPLE
//                   (occupancy at h and h/2 gives the data's box-counting dimension: no full-cloud trial pass)
//   k_cell_ids      cell id per point + histogram              (the only pass with random atomics)
//   scan            exclusive prefix sum of the histogram = cell offsets (three small kernels, no library)
//   k_scatter       counting-sort scatter: every point is written ONCE, as a packed 32-byte record
//                   (x, y, z, original index), into its cell's range
// Per iteration each query is pulled back into the cloud's frame with the rigid inverse of H and ONE WAVE
// enumerates the cell rows that intersect a ball around it; every candidate is evaluated with the exact
// arithmetic contract (T)+(D) from its original coordinates and compared lexicographically on (d2, original
// index) -- so the order of the points inside a cell does not matter.  The ball radius comes from an exact upper
// bound of the answer (previous iteration's match re-evaluated under the new H) or from an expanding search;
// termination needs best <= radius minus a slack that covers the rounding of H^-1 q and the non-orthogonality of
// the floating-point R, so the result equals the brute-force scan's, bit for bit (tests compare the two on
// full-size inputs).  The reference instead rebuilds a cKDTree on the transformed cloud every iteration
// (corrpts.py:131, simpleicp.py:188-202).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cmath>
#include <map>
#include <mutex>
#include <algorithm>
#include <stdint.h>

#include "sicp_internal.h"
#include "sicp_lanes.h"
#include "sicp_grid_dev.h"
#include "sicp_normals.h"

namespace sicp {

static inline unsigned cdiv(long a, long b) { return (unsigned)((a + b - 1) / b); }

__device__ __forceinline__ unsigned long long okey(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}

__device__ __forceinline__ double wmin_d(double v)
{
    v = fmin(v, lane_xor_f64<32>(v)); v = fmin(v, lane_xor_f64<16>(v)); v = fmin(v, lane_xor_f64<8>(v));
    v = fmin(v, lane_xor_f64<4>(v));  v = fmin(v, lane_xor_f64<2>(v));  v = fmin(v, lane_xor_f64<1>(v));
    return v;
}

// One pass over a cloud: out[0..2] = min keys, out[3..5] = max keys (ordered-uint64 image of the doubles),
// out[6] = bits of the largest squared norm (a NaN sticks: the upload rejects non-finite clouds).
// Wave reductions are register moves, the block folds in LDS: 7 atomics per BLOCK.
__global__ __launch_bounds__(256) void k_cloud_stats(const double *__restrict__ x, const double *__restrict__ y,
                                                     const double *__restrict__ z, long n, unsigned long long *__restrict__ out)
{
    __shared__ double red[4][8];
    double lo[3] = {__builtin_inf(), __builtin_inf(), __builtin_inf()};
    double nh[3] = {__builtin_inf(), __builtin_inf(), __builtin_inf()};      // minus the maxima
    double m = 0.0;
    bool bad = false;
    // four points per lane and step: twelve independent loads in flight (a plain grid-stride loop is latency-bound)
    const long stride = (long)gridDim.x * 1024;
    for (long base = (long)blockIdx.x * 1024 + threadIdx.x; base < n; base += stride) {
        double v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = base + 256 * u;
            const long ic = i < n ? i : n - 1;                // clamped: a repeated point changes no statistic
            v[u][0] = x[ic]; v[u][1] = y[ic]; v[u][2] = z[ic];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { lo[a] = fmin(lo[a], v[u][a]); nh[a] = fmin(nh[a], -v[u][a]); }
            const double nn = fma(v[u][2], v[u][2], fma(v[u][1], v[u][1], v[u][0] * v[u][0]));
            bad = bad || !(nn < __builtin_inf());             // NaN or inf
            m = fmax(m, nn);
        }
    }
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = wmin_d(lo[a]); nh[a] = wmin_d(nh[a]); }
    m = -wmin_d(-m);
    const bool anybad = __ballot(bad) != 0ull;
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { red[wid][a] = lo[a]; red[wid][3 + a] = nh[a]; }
        red[wid][6] = m; red[wid][7] = anybad ? 1.0 : 0.0;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        const int a = threadIdx.x;
        if (a < 6) {
            const double v = fmin(fmin(red[0][a], red[1][a]), fmin(red[2][a], red[3][a]));
            if (a < 3) atomicMin(out + a, okey(v)); else atomicMax(out + a, okey(-v));
        } else {
            double v = fmax(fmax(red[0][6], red[1][6]), fmax(red[2][6], red[3][6]));
            if (red[0][7] + red[1][7] + red[2][7] + red[3][7] > 0.0) v = __builtin_nan("");
            // non-negative doubles order like their bit patterns; a NaN's pattern is above every finite one
            atomicMax(out + 6, (unsigned long long)__double_as_longlong(v));
        }
    }
}

// cell id of every point + histogram; OCC: also count the cells that receive their first point (needs the
// returning flavour of the atomic: only small clouds, whose cell size no probe has checked, ask for it)
template <bool OCC>
__global__ __launch_bounds__(256) void k_cell_ids(const double *__restrict__ x, const double *__restrict__ y,
                                                  const double *__restrict__ z, long n, GridGeom G,
                                                  uint32_t *__restrict__ ids, uint32_t *__restrict__ counts,
                                                  unsigned long long *__restrict__ occupied)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    bool first = false;
    if (i < n) {
        const uint32_t id = cell_of(G, x[i], y[i], z[i]);
        ids[i] = id;
        if (OCC) first = atomicAdd(counts + id, 1u) == 0u;
        else atomicAdd(counts + id, 1u);
    }
    if (OCC) {
        const unsigned long long b = __ballot(first);
        if ((threadIdx.x & 63) == 0 && b) atomicAdd(occupied, (unsigned long long)__popcll((long long)b));
    }
}

// Cell ids for ORDERING queries (points_order_build): cells are numbered tile by tile -- 8 x 8 cells in (x, y) -- instead of row by
// row, so that consecutive slots of the order cover a compact patch and not a strip one cell wide: the waves an XCD runs side by
// side then share the rows their balls reach into (the halo of a 64-cell square is a sixth of it, of a strip more than half).
// G.dim[0], G.dim[1] are multiples of 8 here.
__global__ __launch_bounds__(256) void k_cell_ids_tiled(const double *__restrict__ x, const double *__restrict__ y,
                                                        const double *__restrict__ z, long n, GridGeom G,
                                                        uint32_t *__restrict__ ids, uint32_t *__restrict__ counts)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int cx = cell_coord(x[i], G.mn[0], G.inv_h, G.dim[0]);
    const int cy = cell_coord(y[i], G.mn[1], G.inv_h, G.dim[1]);
    const int cz = cell_coord(z[i], G.mn[2], G.inv_h, G.dim[2]);
    const uint32_t tile = ((uint32_t)cz * (uint32_t)(G.dim[1] >> 3) + (uint32_t)(cy >> 3)) * (uint32_t)(G.dim[0] >> 3) + (uint32_t)(cx >> 3);
    const uint32_t id = tile * 64u + (uint32_t)((cy & 7) * 8 + (cx & 7));
    ids[i] = id;
    atomicAdd(counts + id, 1u);
}

// Occupancy probe for the cell-size choice: histogram of the points of a SAMPLE (every `every`-th 1024-point chunk of
// the cloud: coalesced, a fraction of the traffic) that fall inside a small window of the bounding box (G describes
// the window's own grid); out[0] += sampled points inside, out[1] += cells that got a first point
__global__ __launch_bounds__(256) void k_window_probe(const double *__restrict__ x, const double *__restrict__ y,
                                                      const double *__restrict__ z, long n, long every, GridGeom G, double wx,
                                                      double wy, double wz, uint32_t *__restrict__ counts,
                                                      unsigned long long *__restrict__ out)
{
    __shared__ unsigned long long part[4][2];
    unsigned long long inside = 0, firsts = 0;
    const long chunks = (n + 1023) / 1024;
    for (long ch = (long)blockIdx.x * every; ch < chunks; ch += (long)gridDim.x * every) {
        const long base = ch * 1024 + threadIdx.x;
        double v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = base + 256 * u;
            const long ic = i < n ? i : n - 1;
            v[u][0] = x[ic]; v[u][1] = y[ic]; v[u][2] = z[ic];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double px = v[u][0], py = v[u][1], pz = v[u][2];
            const bool in = base + 256 * u < n && px >= G.mn[0] && px < wx && py >= G.mn[1] && py < wy && pz >= G.mn[2] && pz < wz;
            if (in) { inside += 1; firsts += atomicAdd(counts + cell_of(G, px, py, pz), 1u) == 0u ? 1 : 0; }
        }
    }
    inside = wsum_u64(inside); firsts = wsum_u64(firsts);
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6][0] = inside; part[threadIdx.x >> 6][1] = firsts; }
    __syncthreads();
    if (threadIdx.x < 2) {                     // one atomic per block and counter
        const unsigned long long t = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        if (t) atomicAdd(out + threadIdx.x, t);
    }
}

// ---- exclusive prefix sum of the histogram (cell offsets), three launches, no library --------------------------
constexpr int SCAN_ITEMS = 2048;            // per block: 256 lanes x 8
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned *total)      // 256 lanes; returns the lane's exclusive prefix
{
    __shared__ unsigned wsum4[4];
    const unsigned incl = wscan_u32(v);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 63) wsum4[wid] = incl;
    __syncthreads();
    unsigned before = 0;
    for (int w = 0; w < wid; ++w) before += wsum4[w];
    *total = wsum4[0] + wsum4[1] + wsum4[2] + wsum4[3];
    return before + incl - v;
}
// (sumsq, nullable: += sum of the squares of the counts -- sum c^2 / n is the occupancy of the cell an average POINT lives in, what a
// cloud whose density varies by orders of magnitude must be binned for: the plain average over occupied cells hides its dense core)
__global__ __launch_bounds__(256) void k_scan_sums(const uint32_t *__restrict__ in, long n, uint32_t *__restrict__ block_sum,
                                                   unsigned long long *__restrict__ sumsq)
{
    const long base = (long)blockIdx.x * SCAN_ITEMS + (long)threadIdx.x * 8;
    unsigned s = 0;
    unsigned long long q = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (base + k < n) { const unsigned v = in[base + k]; s += v; q += (unsigned long long)v * v; }
    unsigned total;
    (void)block_excl_scan(s, &total);
    if (threadIdx.x == 0) block_sum[blockIdx.x] = total;
    if (sumsq) {
        __shared__ unsigned long long qs[4];
        q = wsum_u64(q);
        if ((threadIdx.x & 63) == 0) qs[threadIdx.x >> 6] = q;
        __syncthreads();
        if (threadIdx.x == 0) { const unsigned long long t = (qs[0] + qs[1]) + (qs[2] + qs[3]); if (t) atomicAdd(sumsq, t); }
    }
}
__global__ __launch_bounds__(256) void k_scan_blocks(uint32_t *__restrict__ block_sum, long nblocks)      // ONE workgroup, in place -> exclusive
{
    __shared__ unsigned carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (long base = 0; base < nblocks; base += 256) {
        const long i = base + threadIdx.x;
        const unsigned v = i < nblocks ? block_sum[i] : 0u;
        unsigned total;
        const unsigned ex = block_excl_scan(v, &total);
        const unsigned carry = carry_s;
        if (i < nblocks) block_sum[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
}
// out[i] = sum of in[0..i) for i in [0, n]  (entry n = the grand total); cursor receives a copy (either may be null)
__global__ __launch_bounds__(256) void k_scan_final(const uint32_t *__restrict__ in, long n, const uint32_t *__restrict__ block_off,
                                                    uint32_t *__restrict__ out, uint32_t *__restrict__ cursor)
{
    const long base = (long)blockIdx.x * SCAN_ITEMS + (long)threadIdx.x * 8;
    unsigned v[8], s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = base + k < n ? in[base + k] : 0u; s += v[k]; }
    unsigned total;
    unsigned run = block_off[blockIdx.x] + block_excl_scan(s, &total);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (base + k <= n) { if (out) out[base + k] = run; if (cursor && base + k < n) cursor[base + k] = run; }
        run += v[k];
    }
}

// counting-sort scatter: point i goes to the next free slot of its cell as a packed record (x, y, z, index bits)
__global__ __launch_bounds__(256) void k_scatter(const double *__restrict__ x, const double *__restrict__ y,
                                                 const double *__restrict__ z, const uint32_t *__restrict__ ids, long n,
                                                 uint32_t *__restrict__ cursor, double4 *__restrict__ rec)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t pos = atomicAdd(cursor + ids[i], 1u);
    rec[pos] = make_double4(x[i], y[i], z[i], __longlong_as_double((long long)i));
}

// every `stride`-th point of a cloud (column layout in, column layout out): the subsample whose nearest point bounds a cold search
__global__ __launch_bounds__(256) void k_stride_sample(const double *__restrict__ x, const double *__restrict__ y,
                                                       const double *__restrict__ z, long n, long stride, long m, long mpad,
                                                       double *__restrict__ out)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= mpad) return;
    const bool in = i < m;
    const long j = in ? i * stride : 0;
    out[i] = in ? x[j] : 1e300; out[mpad + i] = in ? y[j] : 1e300; out[2 * mpad + i] = in ? z[j] : 1e300;
}

// the same counting sort for QUERIES, keeping only the permutation: order[slot] = query (large query sets are searched
// in cell order so that waves running side by side read the same rows of the cloud's grid)
__global__ __launch_bounds__(256) void k_scatter_order(const uint32_t *__restrict__ ids, long n, uint32_t *__restrict__ cursor,
                                                       uint32_t *__restrict__ order)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    order[atomicAdd(cursor + ids[i], 1u)] = (uint32_t)i;
}

// ------------------------------------------------------------------------------------
// one wave per query
// ------------------------------------------------------------------------------------
// CHAINED: the launch belongs to a run whose iterations are enqueued back to back -- H and its inverse come from
// the device-resident loop state the previous tail launch left (sicp_tail.hip), and the launch exits at once
// when that tail declared the run over.
//
// Search of one pass: lane r of the wave fetches row r's contiguous point range (the cells of one (cy, cz) row are
// contiguous in the cell order); the non-empty rows are then visited FOUR AT A TIME -- their ranges are broadcast
// with v_readlane (uniform row index: no LDS crossbar), lane l takes point l of each row, and the four 32-byte
// record loads are in flight together.  A pass costs two dependent memory round trips (offsets, records).
template <bool XFORM, bool CHAINED, bool EXT /* the options below the line are live: tight boxes, a coarse twin grid, a list of queries.
    A latency-bound kernel pays for every scalar register it spills: the plain instantiation (EXT = false) compiles them all away --
    with them in, the steady match of the headline workload took 8.3 us instead of 6.5 (172 spilled scalar registers against 38) */>
// (parameter order: the eight pointers the first instructions need come first -- preloaded into SGPRs at wave start,
// -amdgpu-kernarg-preload-count=16, instead of a kernarg load every dependent load would queue behind)
__global__ __launch_bounds__(256) void k_grid_nn(
    const IcpDev *__restrict__ st,
    const double *__restrict__ qx, const double *__restrict__ qy, const double *__restrict__ qz,
    const double *__restrict__ prev_p2 /* nullable: (Q,3) a cloud point per query (last match) -> its exact
                                          distance under H bounds the answer */,
    const uint32_t *__restrict__ cell_start, const double4 *__restrict__ rec,
    const uint32_t *__restrict__ order /* nullable: queries in cell order (grid size is a multiple of 8 then) */,
    long Q, GridGeom G, Xf H, Xf Hinv, double rmax, double max_d2, int64_t idx_base,
    double *__restrict__ d2_out, int64_t *__restrict__ idx_out, double *__restrict__ p2_out,
    unsigned long long *__restrict__ work /* nullable: [0] candidates, [1] rows, [2] launches */,
    int flags /* NN_TIGHT: prev_p2 is a bound to search in one go (the nearest point of a subsample), not an old match;
                 NN_APPROX: the first hit is good enough (the caller wants A cloud point near the query -- a bound --, not the nearest) */,
    PostMatch post /* chained match of an ICP iteration: the winning lane also leaves the point-to-plane distance and the
                      planarity verdict (corrpts.py:139-163,195-211) -- it holds the matched point, the query and H already */,
    const unsigned long long *__restrict__ cell_box_arg /* nullable: the cells' tight boxes (sicp_grid_dev.h) -- far searches trim their rows */,
    GridGeom G2, const uint32_t *__restrict__ cell_start2_arg /* nullable: no coarse grid */, const double4 *__restrict__ rec2,
    const uint32_t *__restrict__ redo_list_arg /* nullable: ONLY the queries listed here (what the filtered many-queries kernel left) */,
    const unsigned *__restrict__ redo_count /* entries of redo_list */,
    unsigned *__restrict__ redo_clear /* nullable: the counter the NEXT search's kernels add to -- cleared here (nobody touches it
                                         before that search is launched: stream order) */)
{
    const unsigned long long *const cell_box = EXT ? cell_box_arg : nullptr;
    const uint32_t *const cell_start2 = EXT ? cell_start2_arg : nullptr;
    const uint32_t *const redo_list = EXT ? redo_list_arg : nullptr;
    const int lane = threadIdx.x & 63;
    const int tight = flags & NN_TIGHT;
    const bool approx = (flags & NN_APPROX) != 0;
  auto one = [&](const long q) {
    const double ax = qx[q], ay = qy[q], az = qz[q];      // (issued before the loop state is waited for)
    double px0 = 0, py0 = 0, pz0 = 0;
    if (prev_p2) { px0 = prev_p2[3 * q]; py0 = prev_p2[3 * q + 1]; pz0 = prev_p2[3 * q + 2]; }
    float pnx = 0.f, pny = 0.f, pnz = 0.f, ppl = 0.f;     // the query's normal and planarity (wave-uniform; in flight during the search)
    if (post.dist) { pnx = post.normals[3 * q]; pny = post.normals[3 * q + 1]; pnz = post.normals[3 * q + 2]; ppl = post.planarity[q]; }
    if (CHAINED) {
        H = st->H; Hinv = st->Hinv;
        if (st->stop) return;
    }
    double cxq = ax, cyq = ay, czq = az;                  // query in the cloud's own frame
    if (XFORM) xf(Hinv, ax, ay, az, cxq, cyq, czq);
    // covers rounding of H^-1 q and |R^T R - I| ~ 1e-16: distances in the two frames agree to
    // ~1e-15 * scale; 1e-12 * scale leaves three orders of magnitude
    const double scale = rmax + (fabs(cxq) + fabs(cyq) + fabs(czq)) + 1.0;     // (1-norm: an upper bound of |q| is all the slack needs)
    const double slack = 1e-12 * scale;
    double r_lim = (max_d2 < __builtin_inf()) ? sqrt(max_d2) * (1.0 + 1e-12) + slack : __builtin_inf();
    bool lim_is_bound = false;                            // r_lim is the distance to a cloud point: that ball is never empty
    if (prev_p2) {
        double X = px0, Y = py0, Z = pz0;
        if (XFORM) { double u, v, w; xf(H, X, Y, Z, u, v, w); X = u; Y = v; Z = w; }
        const double dx = X - ax, dy = Y - ay, dz = Z - az;
        const double bnd = fma(dz, dz, fma(dy, dy, dx * dx));
        if (bnd < __builtin_inf()) {
            const double rb = sqrt(bnd) * (1.0 + 1e-12) + slack;
            if (rb < r_lim) { r_lim = rb; lim_is_bound = true; }
        }
    }
    // a bound that spans many cells (first iterations: the estimate still moves by metres) is not searched in one
    // go: start small and let the first hit shrink the ball
    double r = 0.75 * G.h;
    if (r > r_lim || (tight && r_lim < __builtin_inf())) r = r_lim;

    double best = __builtin_inf(), bx = 0, by = 0, bz = 0;
    uint32_t bidx = 0xffffffffu;
    unsigned long long n_cand = 0, n_rows = 0;
    bool last = false;
    for (int pass = 0; pass < 4096; ++pass) {                 // (ends by itself: the radius doubles until it hits, then one more pass)
        // A cloud whose density varies by orders of magnitude is binned for its dense core (cells of centimetres) -- and a query whose
        // answer lies metres away would walk ten thousand rows of that grid.  Such clouds carry a second, COARSE grid over the same
        // points (cells 8 x as wide): a pass whose ball spans more than a few fine cells runs on it.  Both grids hold every point, so
        // which one a pass reads changes what it costs, never what it finds.  (One query per wave: the choice is wave-uniform.)
        const bool cp = cell_start2 != nullptr && r > 4.0 * G.h;
        const GridGeom &Gp = cp ? G2 : G;
        const uint32_t *__restrict__ csp = cp ? cell_start2 : cell_start;
        const double4 *__restrict__ rcp = cp ? rec2 : rec;
        int lo[3], hi[3];
        const double c3[3] = {cxq, cyq, czq};
        bool all = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double fl = floor((c3[a] - r - Gp.mn[a]) * Gp.inv_h - 1e-6);
            const double fh = floor((c3[a] + r - Gp.mn[a]) * Gp.inv_h + 1e-6);
            lo[a] = fl < 0.0 ? 0 : (fl > (double)(Gp.dim[a] - 1) ? Gp.dim[a] - 1 : (int)fl);
            hi[a] = fh < 0.0 ? 0 : (fh > (double)(Gp.dim[a] - 1) ? Gp.dim[a] - 1 : (int)fh);
            // whole axis covered <=> the ball reaches past both faces of the box
            all = all && (fl <= 0.0) && (fh >= (double)(Gp.dim[a] - 1));
        }
        best = __builtin_inf(); bidx = 0xffffffffu;
        const int ny = hi[1] - lo[1] + 1, nz = hi[2] - lo[2] + 1;
        const long nrows = (long)ny * nz;
        // candidates of up to four rows: ranges are wave-uniform, lane l takes record l (+ 64, ...) of each row
        auto scan_rows = [&](const uint32_t (&rbv)[4], const uint32_t (&rlv)[4]) {
            uint32_t longest = rlv[0] > rlv[1] ? rlv[0] : rlv[1];
            { const uint32_t t2 = rlv[2] > rlv[3] ? rlv[2] : rlv[3]; longest = longest > t2 ? longest : t2; }
            for (uint32_t o = 0; o < longest; o += 64) {              // (rows longer than a wave: dense cells, duplicates)
                double4 P[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    ok[u] = o + (uint32_t)lane < rlv[u];
                    P[u] = rcp[ok[u] ? rbv[u] + o + (uint32_t)lane : 0u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (!ok[u]) continue;
                    double X = P[u].x, Y = P[u].y, Z = P[u].z;
                    if (XFORM) { double a2, b2, c2; xf(H, X, Y, Z, a2, b2, c2); X = a2; Y = b2; Z = c2; }
                    const double dx = X - ax, dy = Y - ay, dz = Z - az;
                    const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
                    const uint32_t oi = (uint32_t)__double_as_longlong(P[u].w);
                    if (d2 < best || (d2 == best && oi < bidx)) { best = d2; bidx = oi; bx = P[u].x; by = P[u].y; bz = P[u].z; }
                }
                if (work) { for (int u = 0; u < 4; ++u) n_cand += ok[u] ? 1 : 0; }
            }
        };
        // The ball, not its bounding cube: a row (cy, cz) is needed only if its (y, z) rectangle comes within r of the query,
        // and then only the cells within sqrt(r^2 - lb^2) of it along x.  (`all`: the cube covers the whole grid and the pass
        // ends the search whatever it finds -- then every row is taken in full.)
        const double r2 = r * r, etol = 1e-6 * Gp.h;
        double cull2 = __builtin_inf();                           // rows farther than this cannot hold the answer (set by hits)
        const float inv_ny = 1.0f / (float)ny;
        const bool few_rows = nrows < (1L << 22);
        // row rr of the pass's block -> its cells [xl, xh] within the ball (the hit's, once there is one), its record range
        auto row_range = [&](long rr, uint32_t &b, uint32_t &len, double &lb2, long &row, int &cy, int &cz, int &xl, int &xh) {
            b = 0; len = 0;
            int oy, oz;
            row_split(rr, ny, inv_ny, few_rows, oy, oz);
            cy = lo[1] + oy; cz = lo[2] + oz;
            row = ((long)cz * Gp.dim[1] + cy) * Gp.dim[0];
            xl = lo[0]; xh = hi[0];
            lb2 = 0.0;
            if (!all) {
                const double yl = Gp.mn[1] + (double)cy * Gp.h, zl = Gp.mn[2] + (double)cz * Gp.h;
                const double dy = fmax(fmax(yl - etol - cyq, cyq - (yl + Gp.h + etol)), 0.0);
                const double dz = fmax(fmax(zl - etol - czq, czq - (zl + Gp.h + etol)), 0.0);
                lb2 = fma(dy, dy, dz * dz);
                const double rem = fmin(r2, cull2) - lb2;
                if (rem >= 0.0) {
                    // half-width along x, rounded up (float sqrt + margin; the cell tolerance covers the rest)
                    const double hw = (rem < 1e-30 ? 1e-15 : (double)(sqrtf((float)rem) * 1.000001f)) + etol;
                    const double fl = floor((cxq - hw - Gp.mn[0]) * Gp.inv_h - 1e-6);
                    const double fh = floor((cxq + hw - Gp.mn[0]) * Gp.inv_h + 1e-6);
                    const int tl = fl < 0.0 ? 0 : (fl > (double)(Gp.dim[0] - 1) ? Gp.dim[0] - 1 : (int)fl);
                    const int th = fh < 0.0 ? 0 : (fh > (double)(Gp.dim[0] - 1) ? Gp.dim[0] - 1 : (int)fh);
                    xl = tl > xl ? tl : xl; xh = th < xh ? th : xh;
                } else {
                    xh = xl - 1;                                  // outside the ball
                }
            }
            if (xh >= xl && lb2 <= cull2) {
                b = csp[row + xl];
                len = csp[row + xh + 1] - b;
            }
        };
        for (long rb = 0; rb < nrows; rb += 64) {
            uint32_t b = 0, len = 0;
            double lb2 = __builtin_inf();
            long row = 0; int cy = 0, cz = 0, xl = 0, xh = -1;
            if (rb + lane < nrows) {
                row_range(rb + lane, b, len, lb2, row, cy, cz, xl, xh);
                // (a later batch of a far search: an earlier batch's hit already bounds the answer)
                if (cell_box && !cp && len > 0 && cull2 < __builtin_inf()) box_trim_row(cell_box, Gp, row, cy, cz, xl, xh, cxq, cyq, czq, cull2, etol, b, len);
            }
            unsigned long long todo = __ballot(len > 0);          // rows of this batch that hold points
            if (work && len > 0) n_rows += 1;                     // (per-lane tallies, summed once at the end)
            if (__popcll((long long)todo) > 4) {
                // many rows (a wide ball: cold start, far query): nearest row first, then drop the rows its hit rules out
                unsigned long long key = len > 0 ? (unsigned long long)__double_as_longlong(lb2) : ~0ull, mk = key;
                { unsigned long long o;
                  o = lane_xor64<32>(mk); mk = o < mk ? o : mk;  o = lane_xor64<16>(mk); mk = o < mk ? o : mk;
                  o = lane_xor64<8>(mk);  mk = o < mk ? o : mk;  o = lane_xor64<4>(mk);  mk = o < mk ? o : mk;
                  o = lane_xor64<2>(mk);  mk = o < mk ? o : mk;  o = lane_xor64<1>(mk);  mk = o < mk ? o : mk; }
                const int j = __ffsll((long long)__ballot(len > 0 && key == mk)) - 1;
                const uint32_t rbv[4] = {(uint32_t)__builtin_amdgcn_readlane((int)b, j), 0u, 0u, 0u};
                const uint32_t rlv[4] = {(uint32_t)__builtin_amdgcn_readlane((int)len, j), 0u, 0u, 0u};
                todo &= ~(1ull << j);
                scan_rows(rbv, rlv);
                double wb = best;
                { double o;
                  o = lane_xor_f64<32>(wb); wb = o < wb ? o : wb;  o = lane_xor_f64<16>(wb); wb = o < wb ? o : wb;
                  o = lane_xor_f64<8>(wb);  wb = o < wb ? o : wb;  o = lane_xor_f64<4>(wb);  wb = o < wb ? o : wb;
                  o = lane_xor_f64<2>(wb);  wb = o < wb ? o : wb;  o = lane_xor_f64<1>(wb);  wb = o < wb ? o : wb; }
                if (wb < __builtin_inf()) {
                    const double rbnd = sqrt(wb) * (1.0 + 1e-12) + slack;
                    const double c2 = rbnd * rbnd;
                    if (c2 < cull2) {
                        cull2 = c2;
                        // the rows still to do: their cells within the HIT's ball, trimmed by the cells' tight boxes
                        if (cell_box && !cp && ((todo >> lane) & 1ull)) {
                            row_range(rb + lane, b, len, lb2, row, cy, cz, xl, xh);
                            if (len > 0) box_trim_row(cell_box, Gp, row, cy, cz, xl, xh, cxq, cyq, czq, cull2, etol, b, len);
                        }
                    }
                    todo &= __ballot(len > 0 && lb2 <= cull2);
                }
            }
            while (todo) {
                // up to four rows per step: ranges by register broadcast, one record per lane and row
                uint32_t rbv[4], rlv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    rbv[u] = 0; rlv[u] = 0;
                    if (todo) {
                        const int j = __ffsll((long long)todo) - 1;
                        todo &= todo - 1ull;
                        rbv[u] = (uint32_t)__builtin_amdgcn_readlane((int)b, j);
                        rlv[u] = (uint32_t)__builtin_amdgcn_readlane((int)len, j);
                    }
                }
                scan_rows(rbv, rlv);
            }
        }
        // wave-wide lexicographic (d2, original index) minimum: DPP butterfly, every lane ends up with it;
        // the lane that found it keeps the coordinates
        const double lbest = best; const uint32_t lidx = bidx;
#define SICP_LEXMIN_STEP(J)                                                                       \
        {                                                                                         \
            const double od = lane_xor_f64<J>(best);                                              \
            const uint32_t oi = lane_xor32<J>(bidx);                                              \
            if (od < best || (od == best && oi < bidx)) { best = od; bidx = oi; }                 \
        }
        SICP_LEXMIN_STEP(32) SICP_LEXMIN_STEP(16) SICP_LEXMIN_STEP(8) SICP_LEXMIN_STEP(4) SICP_LEXMIN_STEP(2) SICP_LEXMIN_STEP(1)
#undef SICP_LEXMIN_STEP
        const bool found = bidx != 0xffffffffu;
        const bool winner = found && lbest == best && lidx == bidx;         // exactly one lane (indices are unique)
        // sqrt(best) + margin <= r, tested on the squares (no sqrt, no division).  The relative margin is HALF the one a
        // follow-up radius carries (r = sqrt(best) * (1 + 1e-12) + slack below), so the pass after a shrink terminates.
        const double r_eff = (r - slack) * (1.0 - 5e-13);
        const double r_eff2 = r_eff > 0.0 ? r_eff * r_eff * (1.0 - 1e-15) : -1.0;
        const bool done = (found && best <= r_eff2)           // nothing outside the ball can beat or tie it
                          || all || r >= r_lim                // searched everything that may qualify
                          || last                             // this ball was sized to hold the previous pass's hit: it holds the answer
                          || (approx && found);               // any cloud point will do
        if (done) {
            const bool ok = found && (best < max_d2);
            if (winner || (!found && lane == 0)) {
                d2_out[q] = ok ? best : __builtin_inf();
                const int64_t m = ok ? idx_base + (int64_t)bidx : (int64_t)-1;
                idx_out[q] = m;
                if (p2_out) {
                    p2_out[3 * q]     = ok ? bx : 0.0;
                    p2_out[3 * q + 1] = ok ? by : 0.0;
                    p2_out[3 * q + 2] = ok ? bz : 0.0;
                }
                if (post.dist) post_match(post, H, q, m, ok ? bx : 0.0, ok ? by : 0.0, ok ? bz : 0.0, ax, ay, az, pnx, pny, pnz, ppl);
                if (post.pack) {
                    double *r5 = post.pack + 5 * q;
                    r5[0] = ok ? best : __builtin_inf(); r5[1] = __longlong_as_double((long long)m);
                    r5[2] = ok ? bx : 0.0; r5[3] = ok ? by : 0.0; r5[4] = ok ? bz : 0.0;
                }
                if (post.pack_idx) post.pack_idx[q] = __longlong_as_double((long long)m);
            }
            break;
        }
        r = found ? sqrt(best) * (1.0 + 1e-12) + slack : 2.0 * r;
        last = found;
        if (r > r_lim) r = r_lim;
    }
    (void)lim_is_bound;
    if (work) {
        n_cand = wsum_u64(n_cand); n_rows = wsum_u64(n_rows);
        if (lane == 0) { atomicAdd(work, n_cand); atomicAdd(work + 1, n_rows); }
        if (q == 0 && lane == 0 && !redo_list) atomicAdd(work + 2, 1ull);
    }
  };
    if (redo_list) {
        // the waves share the list: wave w takes entries w, w + W, ...  (the launch cannot know how long the list is)
        const long cnt = (long)redo_count[0], nw = (long)gridDim.x * (blockDim.x >> 6);
        for (long i = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < cnt; i += nw) one((long)redo_list[i]);
        if (redo_clear && blockIdx.x == 0 && threadIdx.x == 0) *redo_clear = 0u;
        return;
    }
    long q = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (order) {
        // workgroups are dealt round-robin to the 8 XCDs: give each XCD one contiguous eighth of the ordered queries, so that
        // the rows a neighbourhood of queries shares are fetched into ONE L2
        const long per_xcd = gridDim.x >> 3;
        q = ((long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)) * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (q >= Q) return;
        q = order[q];
    }
    if (q >= Q) return;                                   // whole wave leaves together
    one(q);
}

// ------------------------------------------------------------------------------------
// FOUR queries per wave (16 lanes each) -- the throughput flavour of k_grid_nn for large query sets.  With one wave
// per query the search is bound by VALU issue once the machine is full (about 590 instructions per query of which
// most lanes execute a handful usefully: a ball holds ~70 candidates in 2-3 rows); sharing the wave between four
// queries divides the bookkeeping by four.  Same algorithm, same arithmetic, same answers: the 16 lanes of a group
// are one DPP row, so the lexicographic minimum stays a register butterfly; row ranges travel inside the group by
// ds_bpermute (per-group source lane: no uniform readlane).  Groups of a wave run in lock step and idle once done.
// ------------------------------------------------------------------------------------
#ifndef SICP_NN16_OCC
#define SICP_NN16_OCC 4                     // waves per SIMD the register budget is set for (A/B builds: build.build_variant)
#endif
template <bool XFORM, bool CHAINED, int GS /* lanes per query: 16 (four queries per wave) or 8 (eight) */>
__global__ __launch_bounds__(256, SICP_NN16_OCC) void k_grid_nn16(
    const IcpDev *__restrict__ st, const double *__restrict__ qx, const double *__restrict__ qy, const double *__restrict__ qz,
    const double *__restrict__ prev_p2, const uint32_t *__restrict__ cell_start, const double4 *__restrict__ rec,
    const uint32_t *__restrict__ order, long Q, GridGeom G,
    Xf H, Xf Hinv, double rmax, double max_d2, int64_t idx_base,
    double *__restrict__ d2_out, int64_t *__restrict__ idx_out, double *__restrict__ p2_out,
    unsigned long long *__restrict__ work, int tight, PostMatch post)
{
    constexpr int GPW = 64 / GS;                       // queries per wave
    constexpr unsigned GMASK = GS == 16 ? 0xffffu : 0xffu;
    const int lane = threadIdx.x & 63, gl = lane & (GS - 1), gbase = lane & (64 - GS);
    long blk = blockIdx.x;
    if (order) blk = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);     // one contiguous eighth per XCD
    const long slot = (blk * 4 + (threadIdx.x >> 6)) * GPW + lane / GS;
    const bool active = slot < Q;
    if (!__any(active)) return;
    const long q = active ? (order ? (long)order[slot] : slot) : 0;
    const double ax = qx[q], ay = qy[q], az = qz[q];
    double px0 = 0, py0 = 0, pz0 = 0;
    if (prev_p2) { px0 = prev_p2[3 * q]; py0 = prev_p2[3 * q + 1]; pz0 = prev_p2[3 * q + 2]; }
    if (CHAINED) {
        H = st->H; Hinv = st->Hinv;
        if (st->stop) return;
    }
    double cxq = ax, cyq = ay, czq = az;
    if (XFORM) xf(Hinv, ax, ay, az, cxq, cyq, czq);
    const double scale = rmax + (fabs(cxq) + fabs(cyq) + fabs(czq)) + 1.0;     // (1-norm: an upper bound of |q| is all the slack needs)
    const double slack = 1e-12 * scale;
    double r_lim = (max_d2 < __builtin_inf()) ? sqrt(max_d2) * (1.0 + 1e-12) + slack : __builtin_inf();
    if (prev_p2) {
        double X = px0, Y = py0, Z = pz0;
        if (XFORM) { double u, v, w; xf(H, X, Y, Z, u, v, w); X = u; Y = v; Z = w; }
        const double dx = X - ax, dy = Y - ay, dz = Z - az;
        const double bnd = fma(dz, dz, fma(dy, dy, dx * dx));
        if (bnd < __builtin_inf()) {
            const double rb = sqrt(bnd) * (1.0 + 1e-12) + slack;
            if (rb < r_lim) r_lim = rb;
        }
    }
    double r = 0.75 * G.h;
    if (r > r_lim || (tight && r_lim < __builtin_inf())) r = r_lim;

    bool done = !active, last = false;
    unsigned long long n_cand = 0, n_rows = 0;
    for (int pass = 0; pass < 4096 && __any(!done); ++pass) {
        int lo[3], hi[3];
        const double c3[3] = {cxq, cyq, czq};
        bool all = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double fl = floor((c3[a] - r - G.mn[a]) * G.inv_h - 1e-6);
            const double fh = floor((c3[a] + r - G.mn[a]) * G.inv_h + 1e-6);
            lo[a] = fl < 0.0 ? 0 : (fl > (double)(G.dim[a] - 1) ? G.dim[a] - 1 : (int)fl);
            hi[a] = fh < 0.0 ? 0 : (fh > (double)(G.dim[a] - 1) ? G.dim[a] - 1 : (int)fh);
            all = all && (fl <= 0.0) && (fh >= (double)(G.dim[a] - 1));
        }
        double best = __builtin_inf();
        uint32_t bidx = 0xffffffffu, bpos = 0;         // (record position: the group's winner re-reads its coordinates at the end)
        const int ny = hi[1] - lo[1] + 1, nz = hi[2] - lo[2] + 1;
        const long nrows = done ? 0 : (long)ny * nz;
        // candidates of up to two rows per group: lane gl of a group takes record gl (+ 16, ...) of each of its rows
        auto scan_rows = [&](const uint32_t (&rbv)[2], const uint32_t (&rlv)[2]) {
            const uint32_t longest = rlv[0] > rlv[1] ? rlv[0] : rlv[1];
            for (uint32_t o = 0; __any(o < longest); o += GS) {
                double4 P[2];
                bool ok[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    ok[u] = o + (uint32_t)gl < rlv[u];
                    P[u] = rec[ok[u] ? rbv[u] + o + (uint32_t)gl : 0u];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (!ok[u]) continue;
                    double X = P[u].x, Y = P[u].y, Z = P[u].z;
                    if (XFORM) { double a2, b2, c2; xf(H, X, Y, Z, a2, b2, c2); X = a2; Y = b2; Z = c2; }
                    const double dx = X - ax, dy = Y - ay, dz = Z - az;
                    const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
                    const uint32_t oi = (uint32_t)__double_as_longlong(P[u].w);
                    if (d2 < best || (d2 == best && oi < bidx)) { best = d2; bidx = oi; bpos = rbv[u] + o + (uint32_t)gl; }
                }
                if (work) { for (int u = 0; u < 2; ++u) n_cand += ok[u] ? 1 : 0; }
            }
        };
        // the ball, not its bounding cube (see k_grid_nn); once a hit bounds the answer the ball is the hit's, not the pass's
        const double r2 = r * r, etol = 1e-6 * G.h;
        double cull2 = __builtin_inf();
        const float inv_ny = 1.0f / (float)ny;
        const bool few_rows = nrows < (1L << 22);
        auto row_range = [&](long rr, uint32_t &b, uint32_t &len, double &lb2) {
            b = 0; len = 0;
            int oy, oz;
            row_split(rr, ny, inv_ny, few_rows, oy, oz);
            const int cy = lo[1] + oy, cz = lo[2] + oz;
            const long row = ((long)cz * G.dim[1] + cy) * G.dim[0];
            int xl = lo[0], xh = hi[0];
            lb2 = 0.0;
            if (!all) {
                const double yl = G.mn[1] + (double)cy * G.h, zl = G.mn[2] + (double)cz * G.h;
                const double dy = fmax(fmax(yl - etol - cyq, cyq - (yl + G.h + etol)), 0.0);
                const double dz = fmax(fmax(zl - etol - czq, czq - (zl + G.h + etol)), 0.0);
                lb2 = fma(dy, dy, dz * dz);
                const double rem = fmin(r2, cull2) - lb2;
                if (rem >= 0.0) {
                    const double hw = (rem < 1e-30 ? 1e-15 : (double)(sqrtf((float)rem) * 1.000001f)) + etol;
                    const double fl = floor((cxq - hw - G.mn[0]) * G.inv_h - 1e-6);
                    const double fh = floor((cxq + hw - G.mn[0]) * G.inv_h + 1e-6);
                    const int tl = fl < 0.0 ? 0 : (fl > (double)(G.dim[0] - 1) ? G.dim[0] - 1 : (int)fl);
                    const int th = fh < 0.0 ? 0 : (fh > (double)(G.dim[0] - 1) ? G.dim[0] - 1 : (int)fh);
                    xl = tl > xl ? tl : xl; xh = th < xh ? th : xh;
                } else {
                    xh = xl - 1;
                }
            }
            if (xh >= xl) {
                b = cell_start[row + xl];
                len = cell_start[row + xh + 1] - b;
            }
        };
        for (long rb = 0; __any(rb < nrows); rb += GS) {
            uint32_t b = 0, len = 0;
            double lb2 = __builtin_inf();
            if (rb + gl < nrows) row_range(rb + gl, b, len, lb2);
            unsigned todo = (unsigned)(__ballot(len > 0) >> gbase) & GMASK;      // this group's rows that hold points
            if (work && len > 0) n_rows += 1;
            const bool many = __popc(todo) > 4;
            if (__any(many)) {
                // groups with many rows: nearest row first, then drop the rows its hit rules out and shrink the others'
                // x ranges to the hit's ball (one more look at the cell offsets, a fraction of the candidates)
                unsigned long long key = len > 0 ? (unsigned long long)__double_as_longlong(lb2) : ~0ull, mk = key;
                { unsigned long long o;
                  if constexpr (GS == 16) { o = lane_xor64<8>(mk);  mk = o < mk ? o : mk; }  o = lane_xor64<4>(mk);  mk = o < mk ? o : mk;
                  o = lane_xor64<2>(mk);  mk = o < mk ? o : mk;  o = lane_xor64<1>(mk);  mk = o < mk ? o : mk; }
                const unsigned geq = (unsigned)(__ballot(len > 0 && key == mk) >> gbase) & GMASK;
                const int j = (many && geq) ? __ffs((int)geq) - 1 : 0;
                const uint32_t vb = (uint32_t)__shfl((int)b, gbase + j), vl = (uint32_t)__shfl((int)len, gbase + j);
                const uint32_t rbv[2] = {many ? vb : 0u, 0u};
                const uint32_t rlv[2] = {many ? vl : 0u, 0u};
                if (many) todo &= ~(1u << j);
                scan_rows(rbv, rlv);
                double gb = best;
                { double o;
                  if constexpr (GS == 16) { o = lane_xor_f64<8>(gb);  gb = o < gb ? o : gb; }  o = lane_xor_f64<4>(gb);  gb = o < gb ? o : gb;
                  o = lane_xor_f64<2>(gb);  gb = o < gb ? o : gb;  o = lane_xor_f64<1>(gb);  gb = o < gb ? o : gb; }
                if (many && gb < __builtin_inf()) {
                    const double rbnd = sqrt(gb) * (1.0 + 1e-12) + slack;
                    const double c2 = rbnd * rbnd;
                    if (c2 < cull2) {
                        cull2 = c2;
                        if (((todo >> gl) & 1u) && rb + gl < nrows) row_range(rb + gl, b, len, lb2);
                    }
                }
                todo &= (unsigned)(__ballot(len > 0 && lb2 <= cull2) >> gbase) & GMASK;
            }
            while (__any(todo != 0u)) {
                uint32_t rbv[2], rlv[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const bool has = todo != 0u;
                    const int j = has ? __ffs((int)todo) - 1 : 0;
                    todo &= todo - 1u;
                    const uint32_t vb = (uint32_t)__shfl((int)b, gbase + j), vl = (uint32_t)__shfl((int)len, gbase + j);
                    rbv[u] = has ? vb : 0u; rlv[u] = has ? vl : 0u;
                }
                scan_rows(rbv, rlv);
            }
        }
        // lexicographic (d2, original index) minimum over the group's 16 lanes (one DPP row)
        const double lbest = best; const uint32_t lidx = bidx;
#define SICP_LEXMIN_STEP(J)                                                                       \
        {                                                                                         \
            const double od = lane_xor_f64<J>(best);                                              \
            const uint32_t oi = lane_xor32<J>(bidx);                                              \
            if (od < best || (od == best && oi < bidx)) { best = od; bidx = oi; }                 \
        }
        if constexpr (GS == 16) SICP_LEXMIN_STEP(8)
        SICP_LEXMIN_STEP(4) SICP_LEXMIN_STEP(2) SICP_LEXMIN_STEP(1)
#undef SICP_LEXMIN_STEP
        if (!done) {
            const bool found = bidx != 0xffffffffu;
            const bool winner = found && lbest == best && lidx == bidx;
            const double r_eff = (r - slack) * (1.0 - 5e-13);
            const double r_eff2 = r_eff > 0.0 ? r_eff * r_eff * (1.0 - 1e-15) : -1.0;
            const bool fin = (found && best <= r_eff2) || all || r >= r_lim || last;
            if (fin) {
                const bool ok = found && (best < max_d2);
                if (winner || (!found && gl == 0)) {
                    d2_out[q] = ok ? best : __builtin_inf();
                    idx_out[q] = ok ? idx_base + (int64_t)bidx : (int64_t)-1;
                    if (post.pack_idx) post.pack_idx[q] = __longlong_as_double(ok ? (long long)(idx_base + (int64_t)bidx) : -1ll);
                    if (p2_out || post.dist || post.pack) {
                        double4 W = make_double4(0.0, 0.0, 0.0, 0.0);
                        if (ok) W = rec[bpos];
                        if (p2_out) { p2_out[3 * q] = W.x; p2_out[3 * q + 1] = W.y; p2_out[3 * q + 2] = W.z; }
                        if (post.pack) {
                            double *r5 = post.pack + 5 * q;
                            r5[0] = ok ? best : __builtin_inf(); r5[1] = __longlong_as_double(ok ? (long long)(idx_base + (int64_t)bidx) : -1ll);
                            r5[2] = W.x; r5[3] = W.y; r5[4] = W.z;
                        }
                        // (normal and planarity are fetched here, not up front: this flavour lives at its register limit, and a
                        // full machine hides the round trip)
                        if (post.dist)
                            post_match(post, H, q, ok ? idx_base + (int64_t)bidx : (int64_t)-1, W.x, W.y, W.z, ax, ay, az,
                                       post.normals[3 * q], post.normals[3 * q + 1], post.normals[3 * q + 2], post.planarity[q]);
                    }
                }
                done = true;
            } else {
                r = found ? sqrt(best) * (1.0 + 1e-12) + slack : 1.4142135623730951 * r;
                last = found;
                if (r > r_lim) r = r_lim;
            }
        }
    }
    if (work) {
        n_cand = wsum_u64(n_cand); n_rows = wsum_u64(n_rows);
        if (lane == 0) { atomicAdd(work, n_cand); atomicAdd(work + 1, n_rows); }
        if (slot == 0 && gl == 0) atomicAdd(work + 2, 1ull);
    }
}

// ------------------------------------------------------------------------------------
// k nearest neighbours on the grid (estimate_normals, pointcloud.py:185-186): one wave per query,
// cloud in its own frame (no transform).  The ball radius grows until its cells hold >= k points
// and the k-th distance found fits inside the ball; neighbours are then extracted one per round
// as the lexicographic (d2, original index) minimum above the previous one -- k rounds over a few
// hundred L2-resident candidates, no per-lane lists, any k.  Same answer as the brute-force k-NN.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_grid_knn(
    const double *__restrict__ qx, const double *__restrict__ qy, const double *__restrict__ qz, long Q, int k,
    GridGeom G, const uint32_t *__restrict__ cell_start, const double4 *__restrict__ rec,
    double rmax, int64_t idx_base, double *__restrict__ d2_out, int64_t *__restrict__ idx_out)
{
    const int lane = threadIdx.x & 63;
    const long q = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (q >= Q) return;
    const double ax = qx[q], ay = qy[q], az = qz[q];
    const double scale = rmax + sqrt(fma(az, az, fma(ay, ay, ax * ax))) + 1.0;
    const double slack = 1e-12 * scale;
    const double c3[3] = {ax, ay, az};
    double r = 0.5 * G.h;
    bool final_pass = false;
    for (int pass = 0; pass < 80; ++pass) {
        int lo[3], hi[3];
        bool all = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double fl = floor((c3[a] - r - G.mn[a]) * G.inv_h - 1e-6);
            const double fh = floor((c3[a] + r - G.mn[a]) * G.inv_h + 1e-6);
            lo[a] = fl < 0.0 ? 0 : (fl > (double)(G.dim[a] - 1) ? G.dim[a] - 1 : (int)fl);
            hi[a] = fh < 0.0 ? 0 : (fh > (double)(G.dim[a] - 1) ? G.dim[a] - 1 : (int)fh);
            all = all && (fl <= 0.0) && (fh >= (double)(G.dim[a] - 1));
        }
        const int ny = hi[1] - lo[1] + 1, nz = hi[2] - lo[2] + 1;
        const long nrows = (long)ny * nz;
        // how many points do these cells hold?
        unsigned long long cnt = 0;
        for (long rr = lane; rr < nrows; rr += 64) {
            const int cy = lo[1] + (int)(rr % ny), cz = lo[2] + (int)(rr / ny);
            const long row = ((long)cz * G.dim[1] + cy) * G.dim[0];
            cnt += cell_start[row + hi[0] + 1] - cell_start[row + lo[0]];
        }
        cnt = wsum_u64(cnt);
        if (cnt < (unsigned long long)k && !all) { r *= 2.0; continue; }

        // k extraction rounds over the candidate cells
        double fd = -1.0; uint32_t fi = 0;                 // exclusive lexicographic floor
        bool first = true;
        double dk = __builtin_inf();
        for (int j = 0; j < k; ++j) {
            double best = __builtin_inf(); uint32_t bidx = 0xffffffffu;
            for (long rr = 0; rr < nrows; ++rr) {
                const int cy = lo[1] + (int)(rr % ny), cz = lo[2] + (int)(rr / ny);
                const long row = ((long)cz * G.dim[1] + cy) * G.dim[0];
                const uint32_t b = cell_start[row + lo[0]], e = cell_start[row + hi[0] + 1];
                for (uint32_t i = b + lane; i < e; i += 64) {
                    const double4 P = rec[i];
                    const double dx = P.x - ax, dy = P.y - ay, dz = P.z - az;
                    const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
                    const uint32_t oi = (uint32_t)__double_as_longlong(P.w);
                    const bool above = first || d2 > fd || (d2 == fd && oi > fi);
                    if (above && (d2 < best || (d2 == best && oi < bidx))) { best = d2; bidx = oi; }
                }
            }
#define SICP_LEXMIN_STEP(J)                                                                       \
            {                                                                                     \
                const double od = lane_xor_f64<J>(best);                                          \
                const uint32_t oi = lane_xor32<J>(bidx);                                          \
                if (od < best || (od == best && oi < bidx)) { best = od; bidx = oi; }             \
            }
            SICP_LEXMIN_STEP(32) SICP_LEXMIN_STEP(16) SICP_LEXMIN_STEP(8) SICP_LEXMIN_STEP(4) SICP_LEXMIN_STEP(2) SICP_LEXMIN_STEP(1)
#undef SICP_LEXMIN_STEP
            const bool ok = bidx != 0xffffffffu;
            if (final_pass || all) {
                if (lane == 0) {
                    idx_out[q * k + j] = ok ? idx_base + (int64_t)bidx : (int64_t)-1;
                    if (d2_out) d2_out[q * k + j] = ok ? best : __builtin_inf();
                }
            }
            if (!ok) { dk = __builtin_inf(); fd = __builtin_inf(); fi = 0xffffffffu; first = false; continue; }
            fd = best; fi = bidx; first = false; dk = best;
        }
        if (final_pass || all) break;
        const double r_eff = (r - slack) / (1.0 + 1e-12);
        if (dk < __builtin_inf() && sqrt(dk) <= r_eff) {
            // the k-th neighbour lies inside the ball: nothing outside can enter the list; emit
            final_pass = true;                              // same r, this time writing the outputs
            continue;
        }
        r = (dk < __builtin_inf()) ? sqrt(dk) * (1.0 + 1e-12) + slack : 2.0 * r;
        final_pass = dk < __builtin_inf();
    }
}

// ------------------------------------------------------------------------------------
// k nearest neighbours in ONE sweep, covariance and eigen-decomposition in the same launch (estimate_normals,
// pointcloud.py:185-203).  k_grid_knn above walks its candidate cells 2 x k times and hands (Q, k) indices to k_normals,
// which gathers every neighbour again from the coordinate columns: 17.9 GB of HBM traffic per 1 M queries for 0.36 GB of
// algorithmic bytes (round 3).  Here:
//   * a wave takes one query at a time and sweeps the cells of a ball around it ONCE (rows culled to the ball as in
//     k_grid_nn, four rows' records in flight together); candidates within the pass's radius -- the only ones that can be
//     among the k nearest IF the ball holds k points -- are compacted into LDS with their coordinates (ballot + mbcnt);
//   * ball holds >= k survivors: every survivor counts the survivors lexicographically below it on (d2, original index)
//     -- its rank, exact and unique -- and the k smallest leave in rank order; fewer: the radius grows towards ~1.8 k
//     expected points and the sweep repeats; more than the LDS holds: the radius shrinks, and data that defeats that
//     (thousands of coincident points) takes the k-round extraction of k_grid_knn over the same cells;
//   * the neighbours' coordinates are still in LDS: six lanes form the mean and the six covariance sums in the operation
//     order of oracle/sicp_oracle.c:orc_normals (sequential in rank order), nothing is gathered again and no index goes to memory unless
//     the caller asked for it;
//   * a wave works through `batch` queries that are neighbours in space (the caller hands the queries in cell order, tile by
//     tile): their cells are in L1 / L2, and the k-th distances met so far (smoothed) set the next starting radius, so a cloud
//     whose density varies does not pay shrinking or growing sweeps at every query;
//   * the six covariance sums go to memory (48 B per query) and k_cov_normals runs the Jacobi eigen-solver with one lane per
//     query: done by the one lane of the wave that holds a query's sums it would cost more than the search (measured: 3.4 ms per
//     1 M queries against 1.8 ms for everything else), and batching it inside this kernel ties the batch size to it -- 64 queries
//     per wave put 8 x the distinct cells in flight per XCD and doubled the HBM traffic (1.84 GB against 0.9 GB).
// Same answers as k_grid_knn + k_normals, bit for bit (tests run both).
// ------------------------------------------------------------------------------------
struct KnnKey { double d2; uint32_t idx; uint32_t pad; };

__device__ __forceinline__ void wave_lds_sync()
{
    // LDS operations of one wave execute in order: only the compiler has to be kept from moving them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int KS_MAXK = 128;                // largest k of the sweep kernel (above: k_grid_knn + k_normals)
// LDS of a wave, in doubles: keys 2 + coordinates 3 per survivor slot | the k winners' coordinates in rank order | the k-th distance
__host__ __device__ constexpr int ks_wave_doubles(int cap, int kpad) { return 5 * cap + 3 * kpad + 2; }
__device__ __forceinline__ double rdlane_f64(double v, int l)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

template <int NS /* survivor slots per lane: the LDS of a wave holds 64 * NS */>
__global__ __launch_bounds__(256, 4) void k_grid_knn_sweep(
    const double *__restrict__ qx, const double *__restrict__ qy, const double *__restrict__ qz,
    const uint32_t *__restrict__ order /* nullable: queries in cell order (grid size is a multiple of 8 then) */,
    const uint32_t *__restrict__ cell_start, const double4 *__restrict__ rec,
    long Q, int k, int batch, GridGeom G, double rmax, double r_first, int64_t idx_base,
    double *__restrict__ d2_out /* nullable */, int64_t *__restrict__ idx_out /* nullable */,
    double *__restrict__ cov_out /* nullable: (Q, 6) by SLOT: upper triangle of every neighbourhood's sample covariance (NaN: fewer than k points) */,
    unsigned long long *__restrict__ work /* nullable: [0] candidates read, [1] sweeps, [2] queries on the k-round path, [3] survivors */,
    const uint32_t *__restrict__ redo_list /* nullable: only the slots listed here (what k_grid_knn_sweep4 left for this kernel) */,
    const unsigned *__restrict__ redo_count)
{
    constexpr int CAP = 64 * NS;
    extern __shared__ double ks_lds[];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kpad = (k + 7) & ~7;
    double *wbase = ks_lds + (size_t)wid * ks_wave_doubles(CAP, kpad);
    KnnKey *keys = (KnnKey *)wbase;
    double *xyz = wbase + 2 * CAP;
    double *nbr = wbase + 5 * CAP;                    // [k][3]
    double *dk_slot = nbr + 3 * kpad;

    long blk = blockIdx.x;
    if (order && !redo_list) blk = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);       // one contiguous eighth per XCD
    int nb;
    uint32_t my_slot = 0;                             // lane b: the b-th slot this wave works on
    if (redo_list) {
        // the waves share the list: wave w takes entries w, w + W, w + 2 W, ... (at most 64: the launch has Q / 64 waves)
        const long W = (long)gridDim.x * 4, wv = blk * 4 + wid, cnt = (long)*redo_count;
        if (wv >= cnt) return;
        const long mine = (cnt - wv + W - 1) / W;
        nb = (int)(mine < 64 ? mine : 64);
        if (lane < nb) my_slot = redo_list[wv + (long)lane * W];
    } else {
        const long slot0 = (blk * 4 + wid) * (long)batch;
        if (slot0 >= Q) return;
        nb = (int)((Q - slot0 < (long)batch) ? (Q - slot0) : (long)batch);
        if (lane < nb) my_slot = (uint32_t)(slot0 + lane);
    }
    uint32_t my_q = 0;
    if (lane < nb) my_q = order ? order[my_slot] : my_slot;
    const double inv_km1 = 1.0 / (double)(k - 1);
    unsigned long long n_cand = 0, n_sweeps = 0, n_slow = 0, n_surv = 0;
    const double etol = 1e-6 * G.h;
    // squared k-th distance the next starting radius is derived from: the caller's estimate (cell occupancy) until the data says otherwise
    double dk_est = (r_first * r_first) * (1.0 / (1.35 * 1.35));

    for (int b = 0; b < nb; ++b) {
        const long q = (long)(uint32_t)__builtin_amdgcn_readlane((int)my_q, b);
        const double ax = qx[q], ay = qy[q], az = qz[q];
        const double scale = rmax + (fabs(ax) + fabs(ay) + fabs(az)) + 1.0;       // (1-norm: an upper bound of |q| is all the slack needs)
        const double slack = 1e-12 * scale;
        const double c3[3] = {ax, ay, az};
        double r = (double)(1.35f * sqrtf((float)dk_est)) + slack;       // (a starting radius: any value is correct)
        if (!(r >= 0.25 * r_first)) r = 0.25 * r_first;                   // (coincident points: never start at zero)
        int shrinks = 0, sweeps = 0;
        int kk = 0;                                   // neighbours found (k, or every point of the cloud if it holds fewer)
        for (int pass = 0; pass < 4096; ++pass) {
            int lo[3], hi[3];
            bool all = true;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double fl = floor((c3[a] - r - G.mn[a]) * G.inv_h - 1e-6);
                const double fh = floor((c3[a] + r - G.mn[a]) * G.inv_h + 1e-6);
                lo[a] = fl < 0.0 ? 0 : (fl > (double)(G.dim[a] - 1) ? G.dim[a] - 1 : (int)fl);
                hi[a] = fh < 0.0 ? 0 : (fh > (double)(G.dim[a] - 1) ? G.dim[a] - 1 : (int)fh);
                all = all && (fl <= 0.0) && (fh >= (double)(G.dim[a] - 1));
            }
            const int ny = hi[1] - lo[1] + 1, nz = hi[2] - lo[2] + 1;
            const long nrows = (long)ny * nz;
            // a candidate at d2 <= thr lies inside the ball with room for every rounding: no point of a culled cell ties or beats it
            const double r_eff = (r - slack) * (1.0 - 1.5e-12);       // (<= (r - slack) / (1 + 1e-12), without the division)
            const double thr = all ? __builtin_inf() : (r_eff > 0.0 ? r_eff * r_eff : -1.0);
            const double r2 = r * r;
            unsigned ns = 0;                          // survivors so far (wave-uniform)
            n_sweeps += 1; ++sweeps;
            // (a ball spans a handful of rows: lane (y + 8 z) takes row (y, z) -- no division; wider balls enumerate their rows in full)
            const bool few = ny <= 8 && nz <= 8;
            for (long rb = 0; rb < (few ? 1L : nrows); rb += 64) {
                uint32_t rbeg = 0, rlen = 0;
                int oy = lane & 7, oz = lane >> 3;
                bool have = oy < ny && oz < nz;
                if (!few) {
                    have = rb + lane < nrows;
                    row_split(rb + lane, ny, 1.0f / (float)ny, nrows < (1L << 22), oy, oz);
                }
                if (have) {
                    const int cy = lo[1] + oy, cz = lo[2] + oz;
                    const long row = ((long)cz * G.dim[1] + cy) * G.dim[0];
                    int xl = lo[0], xh = hi[0];
                    if (!all) {
                        const double yl = G.mn[1] + (double)cy * G.h, zl = G.mn[2] + (double)cz * G.h;
                        const double dy = fmax(fmax(yl - etol - ay, ay - (yl + G.h + etol)), 0.0);
                        const double dz = fmax(fmax(zl - etol - az, az - (zl + G.h + etol)), 0.0);
                        const double rem = r2 - fma(dy, dy, dz * dz);
                        if (rem >= 0.0) {
                            const double hw = (rem < 1e-30 ? 1e-15 : (double)(sqrtf((float)rem) * 1.000001f)) + etol;
                            const double fl = floor((ax - hw - G.mn[0]) * G.inv_h - 1e-6);
                            const double fh = floor((ax + hw - G.mn[0]) * G.inv_h + 1e-6);
                            const int tl = fl < 0.0 ? 0 : (fl > (double)(G.dim[0] - 1) ? G.dim[0] - 1 : (int)fl);
                            const int th = fh < 0.0 ? 0 : (fh > (double)(G.dim[0] - 1) ? G.dim[0] - 1 : (int)fh);
                            xl = tl > xl ? tl : xl; xh = th < xh ? th : xh;
                        } else {
                            xh = xl - 1;
                        }
                    }
                    if (xh >= xl) {
                        rbeg = cell_start[row + xl];
                        rlen = cell_start[row + xh + 1] - rbeg;
                    }
                }
                unsigned long long todo = __ballot(rlen > 0);
                while (todo) {
                    uint32_t rbv[4], rlv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        rbv[u] = 0; rlv[u] = 0;
                        if (todo) {
                            const int j = __ffsll((long long)todo) - 1;
                            todo &= todo - 1ull;
                            rbv[u] = (uint32_t)__builtin_amdgcn_readlane((int)rbeg, j);
                            rlv[u] = (uint32_t)__builtin_amdgcn_readlane((int)rlen, j);
                        }
                    }
                    uint32_t longest = rlv[0] > rlv[1] ? rlv[0] : rlv[1];
                    { const uint32_t t2 = rlv[2] > rlv[3] ? rlv[2] : rlv[3]; longest = longest > t2 ? longest : t2; }
                    for (uint32_t o = 0; o < longest; o += 64) {
                        double4 P[4];
                        bool ok[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            ok[u] = o + (uint32_t)lane < rlv[u];
                            P[u] = rec[ok[u] ? rbv[u] + o + (uint32_t)lane : 0u];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const double dx = P[u].x - ax, dy = P[u].y - ay, dz = P[u].z - az;
                            const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
                            const bool sv = ok[u] && d2 <= thr;
                            const unsigned long long m = __ballot(sv);
                            if (m) {
                                const unsigned pos = ns + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                                if (sv && pos < (unsigned)CAP) {
                                    keys[pos].d2 = d2; keys[pos].idx = (uint32_t)__double_as_longlong(P[u].w);
                                    xyz[3 * pos] = P[u].x; xyz[3 * pos + 1] = P[u].y; xyz[3 * pos + 2] = P[u].z;
                                }
                                ns += (unsigned)__popcll((long long)m);
                            }
                            if (work) n_cand += ok[u] ? 1 : 0;
                        }
                    }
                }
            }
            if (ns > (unsigned)CAP && !all && shrinks < 3) {
                // far more than k points in the ball: aim at ~2 k of them (a count that grows like r^2 is the slow case)
                double f = sqrt(2.0 * (double)k / (double)ns);
                r *= f < 0.3 ? 0.3 : (f > 0.8 ? 0.8 : f);
                ++shrinks;
                continue;
            }
            if (ns < (unsigned)k && !all) {
                // too few: aim at ~1.8 k (count ~ r^2 on a surface), at least a quarter more, at most twice the radius
                double f = ns ? sqrt(1.8 * (double)k / (double)ns) : 2.0;
                r *= f < 1.25 ? 1.25 : (f > 2.0 ? 2.0 : f);
                continue;
            }
            ns = (unsigned)__builtin_amdgcn_readfirstlane((int)ns);       // (wave-uniform by construction: tell the compiler)
            n_surv += ns;
            kk = ns < (unsigned)k ? (int)ns : k;
            if (ns <= (unsigned)CAP) {
                // ---- rank by counting: survivor e's rank = number of survivors below it on (d2, index) ----
                wave_lds_sync();
                const int slots = ns <= 64u ? 1 : NS;  // (nearly always one survivor per lane at most)
                double md[NS]; uint32_t mi[NS]; unsigned rk[NS];
#pragma unroll
                for (int t = 0; t < NS; ++t) {
                    const unsigned e = (unsigned)lane + 64u * t;
                    const bool has = e < ns;
                    md[t] = has ? keys[e].d2 : __builtin_inf();
                    mi[t] = has ? keys[e].idx : 0xffffffffu;
                    rk[t] = 0;
                }
                if (slots == 1) {
                    for (unsigned j = 0; j < ns; ++j) {
                        const double od = keys[j].d2; const uint32_t oi = keys[j].idx;    // (same address in every lane: a broadcast)
                        rk[0] += (unsigned)((od < md[0]) | ((od == md[0]) & (oi < mi[0])));
                    }
                } else {
                    for (unsigned j = 0; j < ns; ++j) {
                        const double od = keys[j].d2; const uint32_t oi = keys[j].idx;
#pragma unroll
                        for (int t = 0; t < NS; ++t) rk[t] += (unsigned)((od < md[t]) | ((od == md[t]) & (oi < mi[t])));
                    }
                }
#pragma unroll
                for (int t = 0; t < NS; ++t) {
                    const unsigned e = (unsigned)lane + 64u * t;
                    if (t < slots && e < ns && rk[t] < (unsigned)k) {
                        const unsigned rr = rk[t];
                        nbr[3 * rr] = xyz[3 * e]; nbr[3 * rr + 1] = xyz[3 * e + 1]; nbr[3 * rr + 2] = xyz[3 * e + 2];
                        if (rr == (unsigned)(k - 1)) *dk_slot = md[t];
                        if (idx_out) idx_out[q * k + rr] = idx_base + (int64_t)mi[t];
                        if (d2_out) d2_out[q * k + rr] = md[t];
                    }
                }
            } else {
                // ---- the ball holds more than the LDS does and would not shrink (coincident points), or the whole grid was
                // asked for: k extraction rounds over the pass's cells (k_grid_knn's search); final, the ball holds k points ----
                n_slow += 1;
                double fd = -1.0; uint32_t fi = 0;
                bool first = true;
                kk = 0;
                for (int j = 0; j < k; ++j) {
                    double best = __builtin_inf(); uint32_t bidx = 0xffffffffu, bpos = 0;
                    for (long rr = 0; rr < nrows; ++rr) {
                        const int cy = lo[1] + (int)(rr % ny), cz = lo[2] + (int)(rr / ny);
                        const long row = ((long)cz * G.dim[1] + cy) * G.dim[0];
                        const uint32_t bb = cell_start[row + lo[0]], ee = cell_start[row + hi[0] + 1];
                        for (uint32_t i = bb + lane; i < ee; i += 64) {
                            const double4 P = rec[i];
                            const double dx = P.x - ax, dy = P.y - ay, dz = P.z - az;
                            const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
                            const uint32_t oi = (uint32_t)__double_as_longlong(P.w);
                            const bool above = first || d2 > fd || (d2 == fd && oi > fi);
                            if (above && (d2 < best || (d2 == best && oi < bidx))) { best = d2; bidx = oi; bpos = i; }
                        }
                    }
#define SICP_LEXMIN_STEP(J)                                                                       \
                    {                                                                             \
                        const double od = lane_xor_f64<J>(best);                                  \
                        const uint32_t oi = lane_xor32<J>(bidx), op = lane_xor32<J>(bpos);        \
                        if (od < best || (od == best && oi < bidx)) { best = od; bidx = oi; bpos = op; } \
                    }
                    SICP_LEXMIN_STEP(32) SICP_LEXMIN_STEP(16) SICP_LEXMIN_STEP(8) SICP_LEXMIN_STEP(4) SICP_LEXMIN_STEP(2) SICP_LEXMIN_STEP(1)
#undef SICP_LEXMIN_STEP
                    if (bidx == 0xffffffffu) break;         // the cloud holds fewer than k points
                    const double4 W = rec[bpos];
                    if (lane == 0) {
                        if (idx_out) idx_out[q * k + j] = idx_base + (int64_t)bidx;
                        if (d2_out) d2_out[q * k + j] = best;
                        nbr[3 * j] = W.x; nbr[3 * j + 1] = W.y; nbr[3 * j + 2] = W.z;
                        if (j == k - 1) *dk_slot = best;
                    }
                    fd = best; fi = bidx; first = false; kk = j + 1;
                }
            }
            break;
        }
        // a cloud with fewer than k points: the missing neighbours read -1 / inf
        for (int j = kk + lane; j < k; j += 64) {
            if (idx_out) idx_out[q * k + j] = (int64_t)-1;
            if (d2_out) d2_out[q * k + j] = __builtin_inf();
        }
        wave_lds_sync();
        // the next query of the batch is a neighbour in space: its radius follows the k-th distances met so far (one k-th distance
        // alone scatters by a third; a query that had to resize its ball resets the estimate)
        if (kk == k) {
            const double dk = *dk_slot;
            dk_est = sweeps > 1 ? dk : 0.75 * dk_est + 0.25 * dk;
        }
        if (cov_out) {
            // mean and covariance sums in rank order, one sum per lane (the order pointcloud.py:188-190 is restated in by the oracle)
            double mean = 0.0;
            if (lane < 3 && kk == k) {
                for (int s = 0; s < k; ++s) mean += nbr[3 * s + lane];
                mean /= (double)k;
            }
            const double m0 = rdlane_f64(mean, 0), m1 = rdlane_f64(mean, 1), m2 = rdlane_f64(mean, 2);
            if (lane < 6) {
                double cv = 0.0;
                if (kk == k) {
                    const int a = lane < 3 ? 0 : (lane < 5 ? 1 : 2);
                    const int c = lane < 3 ? lane : (lane < 5 ? lane - 2 : 2);
                    const double ma = a == 0 ? m0 : (a == 1 ? m1 : m2), mc = c == 0 ? m0 : (c == 1 ? m1 : m2);
                    for (int s = 0; s < k; ++s) cv = fma(nbr[3 * s + a] - ma, nbr[3 * s + c] - mc, cv);
                    cv *= inv_km1;
                } else {
                    cv = __builtin_nan("");
                }
                // (by SLOT: consecutive queries of the order write consecutive rows)
                cov_out[6 * (long)(uint32_t)__builtin_amdgcn_readlane((int)my_slot, b) + lane] = cv;
            }
        }
        wave_lds_sync();                              // (the next sweep overwrites the survivors)
    }
    if (work) {
        n_cand = wsum_u64(n_cand);
        if (lane == 0) { atomicAdd(work, n_cand); atomicAdd(work + 1, n_sweeps); atomicAdd(work + 2, n_slow); atomicAdd(work + 3, n_surv); }
    }
}

// ------------------------------------------------------------------------------------
// The same sweep with FOUR queries per wave (16 lanes each), for the common case only.  k_grid_knn_sweep above spends ~690
// vector instructions per query, most of them bookkeeping that 64 lanes execute for ONE query (a ball holds ~90 candidates in
// ~5 rows, ~19 of them inside); here a group of 16 lanes owns a query -- rows one per lane, a row's records 16 at a time,
// survivors compacted by the group's 16 bits of the ballot, ranked by the group in lock step with the other three, mean and
// covariance on its lanes 0..5 -- and the wave pays that bookkeeping once for four.  A group handles exactly what needs no loop
// around it: ONE pass at its starting radius whose ball spans at most 16 rows, not the whole grid, and holds between k and 64
// points.  Anything else (0.5 % of the queries of a typical cloud: a ball that came out short; dense clusters; coincident points;
// tiny clouds) is written to a list and done by k_grid_knn_sweep in a second launch -- same arithmetic, same answers.
// k <= 32.  LDS of a group: 64 survivors (key 16 B + coordinates 24 B), the k winners in rank order, the k-th distance.
// ------------------------------------------------------------------------------------
constexpr int KG_CAP = 64, KG_MAXK = 32;
__host__ __device__ constexpr int kg_group_doubles(int kpad) { return 2 * KG_CAP + 3 * kpad + 2; }

__global__ __launch_bounds__(256, 5) void k_grid_knn_sweep4(
    const double *__restrict__ qx, const double *__restrict__ qy, const double *__restrict__ qz,
    const uint32_t *__restrict__ order, const uint32_t *__restrict__ cell_start, const double4 *__restrict__ rec,
    long Q, int k, int batch /* <= 16 */, GridGeom G, double rmax, double r_first, int64_t idx_base,
    double *__restrict__ d2_out, int64_t *__restrict__ idx_out, double *__restrict__ cov_out,
    uint32_t *__restrict__ redo_list, unsigned *__restrict__ redo_count, unsigned long long *__restrict__ work)
{
    extern __shared__ double ks_lds[];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, gl = lane & 15, gi = lane >> 4, gbase = lane & 48;
    const int kpad = (k + 7) & ~7;
    double *wbase = ks_lds + (size_t)(wid * 4 + gi) * kg_group_doubles(kpad);
    KnnKey *keys = (KnnKey *)wbase;               // (.pad: the candidate's record -- the k winners fetch their coordinates again, from L1 / L2;
    double *nbr = wbase + 2 * KG_CAP;             //  coordinates of all 64 survivors in LDS would cost two of five waves per SIMD)
    double *dk_slot = nbr + 3 * kpad;

    long blk = blockIdx.x;
    if (order) blk = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);       // one contiguous eighth per XCD
    const long slot0 = ((blk * 4 + wid) * 4 + gi) * (long)batch;                            // this GROUP's first slot
    const int nb = slot0 >= Q ? 0 : (int)((Q - slot0 < (long)batch) ? (Q - slot0) : (long)batch);
    if (!__any(nb > 0)) return;
    uint32_t my_q = 0;
    if (gl < nb) my_q = order ? order[slot0 + gl] : (uint32_t)(slot0 + gl);
    const double inv_km1 = 1.0 / (double)(k - 1);
    const double etol = 1e-6 * G.h;
    double dk_est = (r_first * r_first) * (1.0 / (1.35 * 1.35));
    unsigned long long n_cand = 0, n_act = 0, n_defer = 0, n_surv = 0;

    for (int b = 0; b < batch; ++b) {
        const bool active = b < nb;
        const long q = active ? (long)(uint32_t)__shfl((int)my_q, gbase + b) : 0;
        const double ax = qx[q], ay = qy[q], az = qz[q];
        const double scale = rmax + (fabs(ax) + fabs(ay) + fabs(az)) + 1.0;
        const double slack = 1e-12 * scale;
        const double c3[3] = {ax, ay, az};
        double r = (double)(1.35f * sqrtf((float)dk_est)) + slack;
        if (!(r >= 0.25 * r_first)) r = 0.25 * r_first;
        int lo[3], hi[3];
        bool all = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double fl = floor((c3[a] - r - G.mn[a]) * G.inv_h - 1e-6);
            const double fh = floor((c3[a] + r - G.mn[a]) * G.inv_h + 1e-6);
            lo[a] = fl < 0.0 ? 0 : (fl > (double)(G.dim[a] - 1) ? G.dim[a] - 1 : (int)fl);
            hi[a] = fh < 0.0 ? 0 : (fh > (double)(G.dim[a] - 1) ? G.dim[a] - 1 : (int)fh);
            all = all && (fl <= 0.0) && (fh >= (double)(G.dim[a] - 1));
        }
        const int ny = hi[1] - lo[1] + 1, nz = hi[2] - lo[2] + 1;
        const long nrows = (long)ny * nz;
        bool mine = active && !all && nrows <= 16;                 // this group does the query itself (so far)
        const double r_eff = (r - slack) * (1.0 - 1.5e-12);
        const double thr = r_eff > 0.0 ? r_eff * r_eff : -1.0;
        const double r2 = r * r;
        // ---- rows: lane gl takes row gl of the ball's (ny x nz) block ----
        uint32_t rbeg = 0, rlen = 0;
        if (mine && gl < (int)nrows) {
            int oy, oz;
            row_split((long)gl, ny, 1.0f / (float)ny, true, oy, oz);
            const int cy = lo[1] + oy, cz = lo[2] + oz;
            const long row = ((long)cz * G.dim[1] + cy) * G.dim[0];
            int xl = lo[0], xh = hi[0];
            const double yl = G.mn[1] + (double)cy * G.h, zl = G.mn[2] + (double)cz * G.h;
            const double dy = fmax(fmax(yl - etol - ay, ay - (yl + G.h + etol)), 0.0);
            const double dz = fmax(fmax(zl - etol - az, az - (zl + G.h + etol)), 0.0);
            const double rem = r2 - fma(dy, dy, dz * dz);
            if (rem >= 0.0) {
                const double hw = (rem < 1e-30 ? 1e-15 : (double)(sqrtf((float)rem) * 1.000001f)) + etol;
                const double fl = floor((ax - hw - G.mn[0]) * G.inv_h - 1e-6);
                const double fh = floor((ax + hw - G.mn[0]) * G.inv_h + 1e-6);
                const int tl = fl < 0.0 ? 0 : (fl > (double)(G.dim[0] - 1) ? G.dim[0] - 1 : (int)fl);
                const int th = fh < 0.0 ? 0 : (fh > (double)(G.dim[0] - 1) ? G.dim[0] - 1 : (int)fh);
                xl = tl > xl ? tl : xl; xh = th < xh ? th : xh;
            } else {
                xh = xl - 1;
            }
            if (xh >= xl) {
                rbeg = cell_start[row + xl];
                rlen = cell_start[row + xh + 1] - rbeg;
            }
        }
        // ---- sweep: the group's non-empty rows one after the other, 16 records at a time; survivors compacted into its LDS ----
        unsigned todo = (unsigned)(__ballot(rlen > 0) >> gbase) & 0xffffu;
        unsigned ns = 0;
        while (__any(todo != 0u)) {
            const bool has = todo != 0u;
            const int j = has ? __ffs((int)todo) - 1 : 0;
            todo &= todo - 1u;                                                  // (0 stays 0)
            const uint32_t vb = (uint32_t)__shfl((int)rbeg, gbase + j);
            const uint32_t vl = has ? (uint32_t)__shfl((int)rlen, gbase + j) : 0u;
            for (uint32_t o = 0; __any(o < vl); o += 16) {
                const bool ok = o + (uint32_t)gl < vl;
                const double4 P = rec[ok ? vb + o + (uint32_t)gl : 0u];
                const double dx = P.x - ax, dy = P.y - ay, dz = P.z - az;
                const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
                const bool sv = ok && d2 <= thr;
                const unsigned long long m = __ballot(sv);
                if (m) {
                    const unsigned mg = (unsigned)(m >> gbase) & 0xffffu;
                    const unsigned pos = ns + (unsigned)__popc(mg & ((1u << gl) - 1u));
                    if (sv && pos < (unsigned)KG_CAP) {
                        keys[pos].d2 = d2; keys[pos].idx = (uint32_t)__double_as_longlong(P.w); keys[pos].pad = vb + o + (uint32_t)gl;
                    }
                    ns += (unsigned)__popc(mg);
                }
                if (work) n_cand += ok ? 1 : 0;
            }
        }
        mine = mine && ns >= (unsigned)k && ns <= (unsigned)KG_CAP;
        if (active && !mine && gl == 0) {
            // not a one-pass case: the one-query-per-wave kernel does this slot in the next launch
            redo_list[atomicAdd(redo_count, 1u)] = (uint32_t)(slot0 + b);
        }
        if (work && gl == 0) { n_act += active ? 1 : 0; n_defer += (active && !mine) ? 1 : 0; n_surv += mine ? ns : 0; }
        // ---- rank by counting inside the group (entries gl, gl + 16, gl + 32, gl + 48), the four groups in lock step ----
        wave_lds_sync();
        const unsigned nsm = mine ? ns : 0u;
        unsigned nmax = nsm;
        { unsigned o = lane_xor32<16>(nmax); nmax = o > nmax ? o : nmax; o = lane_xor32<32>(nmax); nmax = o > nmax ? o : nmax; }
        nmax = (unsigned)__builtin_amdgcn_readfirstlane((int)nmax);
        double md[4]; uint32_t mi[4], mp[4]; unsigned rk[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const unsigned e = (unsigned)gl + 16u * t;
            const bool hs = e < nsm;
            md[t] = hs ? keys[e].d2 : __builtin_inf();
            mi[t] = hs ? keys[e].idx : 0xffffffffu;
            mp[t] = hs ? keys[e].pad : 0u;
            rk[t] = 0;
        }
        if (nmax <= 32u) {
            for (unsigned j = 0; j < nmax; ++j) {
                const double od = keys[j].d2; const uint32_t oi = keys[j].idx;      // (one address per group)
                const unsigned in = j < nsm ? 1u : 0u;
                rk[0] += in & (unsigned)((od < md[0]) | ((od == md[0]) & (oi < mi[0])));
                rk[1] += in & (unsigned)((od < md[1]) | ((od == md[1]) & (oi < mi[1])));
            }
        } else {
            for (unsigned j = 0; j < nmax; ++j) {
                const double od = keys[j].d2; const uint32_t oi = keys[j].idx;
                const unsigned in = j < nsm ? 1u : 0u;
#pragma unroll
                for (int t = 0; t < 4; ++t) rk[t] += in & (unsigned)((od < md[t]) | ((od == md[t]) & (oi < mi[t])));
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const unsigned e = (unsigned)gl + 16u * t;
            if (e < nsm && rk[t] < (unsigned)k) {
                const unsigned rr = rk[t];
                const double4 W = rec[mp[t]];
                nbr[3 * rr] = W.x; nbr[3 * rr + 1] = W.y; nbr[3 * rr + 2] = W.z;
                if (rr == (unsigned)(k - 1)) *dk_slot = md[t];
                if (idx_out) idx_out[q * k + rr] = idx_base + (int64_t)mi[t];
                if (d2_out) d2_out[q * k + rr] = md[t];
            }
        }
        wave_lds_sync();
        if (mine) dk_est = 0.75 * dk_est + 0.25 * *dk_slot;
        if (cov_out) {
            double mean = 0.0;
            if (gl < 3 && mine) {
                for (int s = 0; s < k; ++s) mean += nbr[3 * s + gl];
                mean /= (double)k;
            }
            const double m0 = __shfl(mean, gbase), m1 = __shfl(mean, gbase + 1), m2 = __shfl(mean, gbase + 2);
            if (gl < 6 && mine) {
                const int a = gl < 3 ? 0 : (gl < 5 ? 1 : 2);
                const int c = gl < 3 ? gl : (gl < 5 ? gl - 2 : 2);
                const double ma = a == 0 ? m0 : (a == 1 ? m1 : m2), mc = c == 0 ? m0 : (c == 1 ? m1 : m2);
                double cv = 0.0;
                for (int s = 0; s < k; ++s) cv = fma(nbr[3 * s + a] - ma, nbr[3 * s + c] - mc, cv);
                cov_out[6 * (slot0 + b) + gl] = cv * inv_km1;
            }
        }
        wave_lds_sync();
    }
    if (work) {
        n_cand = wsum_u64(n_cand); n_act = wsum_u64(n_act); n_defer = wsum_u64(n_defer); n_surv = wsum_u64(n_surv);
        if (lane == 0) { atomicAdd(work, n_cand); atomicAdd(work + 1, n_act - n_defer); atomicAdd(work + 3, n_surv); }
    }
}

// covariance -> normal + planarity (pointcloud.py:192-203), one lane per query
__global__ __launch_bounds__(64) void k_cov_normals(const double *__restrict__ cov /* by slot of the order */,
                                                    const uint32_t *__restrict__ order /* nullable: slot = query */, long Q,
                                                    float *__restrict__ normals, float *__restrict__ planarity)
{
    const long slot = (long)blockIdx.x * 64 + threadIdx.x;
    if (slot >= Q) return;
    const long q = order ? (long)order[slot] : slot;
    double c6[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) c6[i] = cov[6 * slot + i];
    float nrm[3] = {__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")}, pl = __builtin_nanf("");
    if (c6[0] == c6[0]) normal_from_cov(c6, nrm, &pl);                       // (NaN: a cloud with fewer than k points)
    normals[3 * q] = nrm[0]; normals[3 * q + 1] = nrm[1]; normals[3 * q + 2] = nrm[2];
    planarity[q] = pl;
}

// ------------------------------------------------------------------------------------
// median / raw-MAD rejection for LARGE Q (corrpts.py:165-188): exact order statistics by digit SELECTION over many
// workgroups on the order-preserving uint64 image of the distances -- keys are formed on the fly from (dist, flag),
// nothing is sorted, nothing but ~35 KB of state is written -- everything chained on the stream without a host
// round trip.  out4 = (m, median, mad, n_kept), out3 = (n, mean, std) of the kept distances.
//
// One statistic = up to six digit passes (12 + 12 + 12 + 12 + 12 + 4 bits from the top) and one finishing launch.
// A pass histograms the digit of every key that still matches the prefix (LDS, wave-aggregated: distances share sign
// and exponent, whole waves hit one bin), adds its non-empty bins to the global histogram, and the LAST block to
// arrive (agent-scope ticket) picks the bin that holds the wanted rank.  As soon as that bin holds <= HS_CAP keys the
// statistic is `done`: the remaining pass launches exit at once (they are enqueued anyway: no host decision inside a
// chained iteration), and the finishing launch collects the few survivors, ranks them exactly and takes the mean of
// the two middle values.  Real distances need two passes (sign/exponent, then 12 mantissa bits leave ~Q/4096 keys).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ double oval64(unsigned long long k)
{
    const unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

constexpr int HS_BINS = 4096, HS_CAP = 256, HS_PASSES = 6, HS_UNROLL = 4;

// ------------------------------------------------------------------------------------
// The rejection in ONE launch: the passes of both statistics, the two finishing steps and the keep / statistics pass are phases
// of a single kernel separated by GRID BARRIERS, so only the passes the data needs are executed (real distances: two per
// statistic) and nothing is dispatched in between (rounds 2-3 kept a launch-per-phase form next to it -- 6 + 1 + 6 + 1 + 1 kernels of
// which 8 exit at once, ~4 us apiece -- as a cross-check; round 4 removed it: the windowed form below, the general form and the
// oracle check each other now).
//
//   * every block is resident at once (at most one block per CU is asked for, 256 lanes, 16 KiB of LDS), so the phases can meet
//     at the grid barrier of sicp_lanes.h (fence-free: everything blocks tell each other here travels in agent-scope atomics;
//     barrier numbers grow over the life of the state buffer, the host hands every launch the number it starts from);
//   * a pass needs ONE barrier: blocks add their local histograms to hist[p % 3], meet, and then EVERY block reads the complete
//     histogram and picks the bin itself (same integers, same answer) -- no second meeting to broadcast the pick.  The buffer two
//     passes ahead, hist[(p + 2) % 3], is wiped right after the barrier of pass p: it was last read before barrier p was entered
//     and is next added to after barrier p + 1;
//   * polling is bounded (about two seconds): a launch that cannot meet itself flags an error and ends instead of hanging
//     the queue.
// ------------------------------------------------------------------------------------
// ---- the WINDOWED form of the two selections: three barriers instead of seven ---------------------------------------------------
// From the second iteration of a run on, median and MAD are where the last iteration left them, give or take a little.  So:
//   sweep 1   every block histograms the distances over H3_NB linear bins of a window med' -+ 1.5 MAD' around the previous
//             median (two more bins catch what lies below / above); the bins are added up in global memory        -- barrier --
//             every block reads the whole histogram: the bin that holds the median's rank, and -- counting outwards from that
//             bin on both sides until half the keys are inside -- two SHELLS of five bins each in which the MAD's rank must end;
//   sweep 2   the keys of the median's bin and of the two shells are collected (raw distances)                    -- barrier --
//             every block sorts the median bin's keys in LDS and reads the exact median off them; forms |d - median| of the
//             shells' keys, sorts those, and reads the MAD off them -- then CHECKS the premise: every key between the shells
//             is nearer to the median than the MAD, every key outside is at least as far (binning is monotone in d, so probe
//             values placed in the shells' end bins bound the two groups);
//   sweep 3   keep mask + kept statistics, as before                                                             -- barrier --
// Exact: the answers are order statistics of the same keys, and a premise that does not hold (the window missed: an estimate
// that still moves, a distribution with a hole at the MAD) sends every block -- they all see the same numbers -- to the general
// digit selection above, at the cost of the phases spent.  To keep that rare the window is only tried when the last two launches
// of this run agree on the MAD to 20 % and on the median to 0.3 MAD.
constexpr int H3_NB = 4096, H3_CAP = 2048;
constexpr int HS_GRID_MAX = 256;                     // blocks of the one-launch rejection (one per CU at most)
constexpr int H3_BCAP_M = 32, H3_BCAP_S = 96;       // keys ONE block may contribute: from the median's bin / from the shells
// zoned flush (more than H3_ZONED_Q correspondences: every block fills nearly every bin, and a million additions to the global
// histogram cost more than the barriers saved): only the bins where the median's bin and the shells are EXPECTED -- the window is
// centred on the last median and 3 MAD wide, so bin 2049 and 2049 -+ 1365 -- are added up bin by bin, what lies between as four sums
constexpr long H3_ZONED_Q = 262144;
constexpr int H3_BM0 = H3_NB / 2 + 1, H3_JS0 = H3_NB / 3, H3_ZH = 64;
constexpr int H3_ZM_LO = H3_BM0 - H3_ZH, H3_ZM_HI = H3_BM0 + H3_ZH;
constexpr int H3_ZL_LO = H3_BM0 - H3_JS0 - H3_ZH - 6, H3_ZL_HI = H3_BM0 - H3_JS0 + H3_ZH + 6;
constexpr int H3_ZR_LO = H3_BM0 + H3_JS0 - H3_ZH - 6, H3_ZR_HI = H3_BM0 + H3_JS0 + H3_ZH + 6;
__device__ __forceinline__ int h3_region(int b)      // 0..3: a coarse region, -1: inside a zone
{
    if (b < H3_ZL_LO) return 0;
    if (b <= H3_ZL_HI) return -1;
    if (b < H3_ZM_LO) return 1;
    if (b <= H3_ZM_HI) return -1;
    if (b < H3_ZR_LO) return 2;
    if (b <= H3_ZR_HI) return -1;
    return 3;
}
// Barriers ONE launch may go through, counted on its longest road: a window that is tried, passes its analysis, collects, and then
// misses on the keys (3: histogram, lists, the meeting before the wipe) + the general form after it (per statistic HS_PASSES digit
// passes and one collecting sweep) + the keep / statistics sweep.  The host advances the barrier counter by exactly this much per
// launch (barrier numbers are absolute): a launch that could take one more would leave the next launch's first barrier open.
constexpr int HS_WINDOW_BARRIERS = 3, HS_STAT_BARRIERS = HS_PASSES + 1, HS_FINAL_BARRIERS = 0;      // (the fold behind the last sweep is a ticket, not a meeting)
constexpr int HS_MAXB = HS_WINDOW_BARRIERS + 2 * HS_STAT_BARRIERS + HS_FINAL_BARRIERS;
static_assert(HS_MAXB == 17, "k_hsel_all's worst case: 3 (window tried and missed late) + 2 x (6 passes + 1)");
struct HselAll {
    GridBar bar;                   // (sicp_lanes.h) all zero when the buffer is new
    unsigned long long nxt[2];     // per statistic: smallest key above the prefix interval (~0 between launches; see hsel_state_init)
    unsigned ncand[2];             // per statistic: candidates appended (0 between launches)
    unsigned hist[3][HS_BINS];     // all zero between launches
    unsigned long long cand[2][HS_CAP];
    // windowed form
    double prior[2][2];            // (median, MAD) of the last two launches, [1] the latest
    unsigned n_prior;              // how many of them this run has produced (the host restarts the count with every setup)
    unsigned ticket;               // blocks that have finished their last sweep (0 between launches): the last one folds the partial sums
    unsigned pad3[2];
    unsigned whist[H3_NB + 2];     // all zero between launches
    unsigned wcoarse[4];           // zoned flush (large Q): keys below / between / above the three zones; zero between launches
    unsigned wcnt[HS_GRID_MAX][2]; // per block: median-bin keys, shell keys it found (rewritten by every launch that collects)
    double wcand[HS_GRID_MAX][H3_BCAP_M + H3_BCAP_S];     // ... and the keys themselves (raw distances)
};

// Order statistics of n <= H3_CAP doubles in LDS without sorting them (a bitonic sort of 2048 keys cost 54 us here: strided LDS
// traffic and 66 barriers): 256 linear bins between the smallest and the largest key, the bin that holds `rank`, its <= 256 keys
// ranked by counting.  v = key of 0-based rank `rank`; v2 = key of rank + 1 (want2; rank + 1 < n).  False when a bin holds more
// than 256 keys (thousands of equal keys: the caller falls back).  256 threads; hb: >= 260 words, sm: >= 264 doubles of LDS.
__device__ bool h3_pick(const double *a, int n, long rank, bool want2, unsigned *hb, double *sm, double &v, double &v2)
{
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    double lo = __builtin_inf(), nh = __builtin_inf();
    for (int i = tid; i < n; i += 256) { lo = fmin(lo, a[i]); nh = fmin(nh, -a[i]); }
    lo = wmin_d(lo); nh = wmin_d(nh);
    if (lane == 0) { sm[256 + wid] = lo; sm[260 + wid] = nh; }
    for (int i = tid; i < 260; i += 256) hb[i] = 0u;
    __syncthreads();
    lo = fmin(fmin(sm[256], sm[257]), fmin(sm[258], sm[259]));
    const double hi = -fmin(fmin(sm[260], sm[261]), fmin(sm[262], sm[263]));
    const double inv = hi > lo ? 255.0 / (hi - lo) : 0.0;
    auto bin = [&](double x) { const int b = (int)((x - lo) * inv); return b > 255 ? 255 : b; };       // monotone in x
    for (int i = tid; i < n; i += 256) atomicAdd(&hb[bin(a[i])], 1u);
    __syncthreads();
    if (wid == 0) {
        // lane l owns bins 4 l .. 4 l + 3
        const unsigned h0 = hb[4 * lane], h1 = hb[4 * lane + 1], h2 = hb[4 * lane + 2], h3 = hb[4 * lane + 3];
        const unsigned mine = h0 + h1 + h2 + h3, incl = wscan_u32(mine);
        const unsigned before = incl - mine;
        if ((long)before <= rank && rank < (long)incl) {
            unsigned acc = before; int j = 0; unsigned c = h0;
            if (rank >= (long)(acc + h0)) { acc += h0; j = 1; c = h1;
                if (rank >= (long)(acc + h1)) { acc += h1; j = 2; c = h2;
                    if (rank >= (long)(acc + h2)) { acc += h2; j = 3; c = h3; } } }
            hb[256] = (unsigned)(4 * lane + j); hb[257] = acc; hb[258] = c;
        }
        if (lane == 0) hb[259] = 0u;
    }
    __syncthreads();
    const int bs = (int)hb[256];
    const long below = hb[257];
    const unsigned cnt = hb[258];
    if (cnt > 256u) return false;
    double above = __builtin_inf();                       // smallest key of the later bins
    for (int i = tid; i < n; i += 256) {
        const double x = a[i];
        const int b = bin(x);
        if (b == bs) sm[atomicAdd(&hb[259], 1u)] = x;
        else if (b > bs) above = fmin(above, x);
    }
    above = wmin_d(above);
    __syncthreads();
    if (lane == 0) sm[256 + wid] = above;
    double mine = 0.0;
    unsigned r = 0;
    if ((unsigned)tid < cnt) {
        mine = sm[tid];
        for (unsigned j = 0; j < cnt; ++j) { const double o = sm[j]; r += (o < mine || (o == mine && j < (unsigned)tid)) ? 1u : 0u; }
    }
    __syncthreads();
    above = fmin(fmin(sm[256], sm[257]), fmin(sm[258], sm[259]));
    const long t = rank - below;
    if ((unsigned)tid < cnt && (long)r == t) sm[264] = mine;
    if ((unsigned)tid < cnt && (long)r == t + 1) sm[265] = mine;
    __syncthreads();
    v = sm[264];
    v2 = want2 ? (t + 1 < (long)cnt ? sm[265] : above) : v;
    __syncthreads();
    return true;
}
__device__ __forceinline__ int h3_bin(double d, double lo, double inv_bw)
{
    // monotone non-decreasing in d (IEEE subtraction, multiplication by a positive constant, min, floor): keys of a lower bin are
    // below the keys of a higher one -- the only property the exactness argument uses
    if (d < lo) return 0;
    const double t = fmin((d - lo) * inv_bw, (double)H3_NB);
    return 1 + (int)t;                                    // 1 .. H3_NB inside the window, H3_NB + 1 above it
}

template <int HSU /* keys per lane and sweep step: all their loads are in flight together (one block per CU has only its own
                       waves to hide the latency behind: 4 at up to ~260 k correspondences, 16 at a million) */>
__global__ __launch_bounds__(256) void k_hsel_all(const double *__restrict__ dist, const uint8_t *__restrict__ flag, long Q,
                                                  HselAll *__restrict__ S, unsigned long long bar_base, uint8_t *__restrict__ keep,
                                                  double *__restrict__ partial /*[3][NE_MAX_GRID]*/, double *__restrict__ out4,
                                                  double *__restrict__ out3, double *__restrict__ host_out, double seq,
                                                  const IcpDev *__restrict__ st, unsigned absent /* test hook: see grid_barrier */,
                                                  int use_prior /* the windowed form may be tried (launch >= 3 of a run) */)
{
    __shared__ unsigned hist[HS_BINS + 8];
    __shared__ unsigned scan[4];
    __shared__ unsigned long long sc[HS_CAP];
    __shared__ double wk[H3_CAP];                    // windowed form: candidate keys being sorted
    __shared__ int h3i[12];
    __shared__ double lcand[H3_BCAP_M + H3_BCAP_S];  // this block's candidates on their way out
    __shared__ double lsm[272];                      // h3_pick's scratch
    __shared__ unsigned lcnt[2];
    __shared__ unsigned creg[4][4];
    __shared__ unsigned long long pnx[4], pick[2];
    __shared__ unsigned long long sel[3];            // picked by the bin's owner: prefix, rank inside it, its count
    __shared__ double red[4][3];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const unsigned g = gridDim.x;
    int nb = 0;                                       // barriers this block has gone through
    if (st && st->stop) {                             // the run is over: leave, but leave the counter where the next launch expects it
        return;
    }
    const long stride = (long)g * (256 * HSU);
    double val[2] = {0.0, 0.0};                       // median, MAD
    unsigned long long m_first = 0;
    bool have = false;                                // the windowed form delivered both statistics
#ifdef SICP_HSEL_DEBUG
    long long tq[16]; int nq = 0;
#define SICP_TQ() tq[nq++] = clock64()
#else
#define SICP_TQ()
#endif
    SICP_TQ();
    // ---- windowed form (see the comment above HselAll) ----
    const double pm1 = S->prior[1][0], pd1 = S->prior[1][1], pm0 = S->prior[0][0], pd0 = S->prior[0][1];
    const unsigned n_prior = S->n_prior;              // (written by the previous launch's block 0: a kernel boundary ago)
    const bool zoned = Q > H3_ZONED_Q;
    const double tol_mad = zoned ? 0.01 : 0.2, tol_med = zoned ? 0.01 : 0.3;      // (zones are +-64 bins = +-0.047 MAD wide)
    bool try_w = use_prior && g <= (unsigned)HS_GRID_MAX && n_prior >= 2u && pd1 > 0.0 && pd1 < __builtin_inf() &&
                 fabs(pd1 - pd0) <= tol_mad * pd1 && fabs(pm1 - pm0) <= tol_med * pd1;
#ifdef SICP_HSEL_DEBUG
    if (blockIdx.x == 0 && tid == 0 && !try_w)
        printf("[hsel3] not tried: use_prior %d n_prior %u prior med %.6g mad %.6g (before: %.6g %.6g)\n", use_prior, n_prior, pm1, pd1, pm0, pd0);
#endif
    if (try_w) {
        const double wlo = pm1 - 1.5 * pd1, inv_bw = (double)H3_NB / (3.0 * pd1);
        for (int i = tid; i < H3_NB + 2; i += 256) hist[i] = 0;
        __syncthreads();
        for (long base = (long)blockIdx.x * (256 * HSU); base < Q; base += stride) {
            double d[HSU]; uint8_t f[HSU];
#pragma unroll
            for (int u = 0; u < HSU; ++u) {
                const long i = base + u * 256 + tid;
                f[u] = i < Q ? flag[i] : (uint8_t)0;
                d[u] = i < Q ? dist[i] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < HSU; ++u) if (f[u]) atomicAdd(&hist[h3_bin(d[u], wlo, inv_bw)], 1u);
        }
        __syncthreads();
        if (!zoned) {
            for (int i = tid; i < H3_NB + 2; i += 256) if (hist[i]) atomicAdd(&S->whist[i], hist[i]);
        } else {
            unsigned cr[4] = {0u, 0u, 0u, 0u};
            for (int i = tid; i < H3_NB + 2; i += 256) {
                const unsigned v = hist[i];
                const int rg = h3_region(i);
                if (rg < 0) { if (v) atomicAdd(&S->whist[i], v); }
                else cr[rg] += v;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { cr[k] = (unsigned)wsum_u64(cr[k]); if (lane == 0) creg[wid][k] = cr[k]; }
            __syncthreads();
            if (tid < 4) { const unsigned t = (creg[0][tid] + creg[1][tid]) + (creg[2][tid] + creg[3][tid]); if (t) atomicAdd(&S->wcoarse[tid], t); }
        }
        SICP_TQ();
        grid_barrier(&S->bar, bar_base + (unsigned long long)(++nb), absent);
        SICP_TQ();
        // every block: the complete histogram as inclusive prefix sums in LDS (thread t owns bins 16 t .. 16 t + 15, thread 0 the
        // last two as well), then one thread finds the median's bin and the shells.  (Zoned flush: a coarse region's keys are
        // booked on its LAST bin -- prefix sums are then exact on that bin and inside the zones, which is where they are read.)
        unsigned h[16], mine = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int bi = 16 * tid + j;
            // (zoned flush: bins outside the zones were never added to -- 256 blocks need not ask the memory side for 3 700 zeros each)
            h[j] = (!zoned || h3_region(bi) < 0) ? __hip_atomic_load(&S->whist[bi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            if (zoned) {
                const int rg = bi == H3_ZL_LO - 1 ? 0 : (bi == H3_ZM_LO - 1 ? 1 : (bi == H3_ZR_LO - 1 ? 2 : -1));
                if (rg >= 0) h[j] += __hip_atomic_load(&S->wcoarse[rg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            mine += h[j];
        }
        const unsigned incl = wscan_u32(mine);
        if (lane == 63) scan[wid] = incl;
        __syncthreads();
        unsigned run = incl - mine;
        for (int w = 0; w < wid; ++w) run += scan[w];
#pragma unroll
        for (int j = 0; j < 16; ++j) { run += h[j]; hist[16 * tid + j] = run; }
        __syncthreads();
        if (tid == 0) {
            unsigned e = hist[H3_NB - 1];
            e += __hip_atomic_load(&S->whist[H3_NB], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); hist[H3_NB] = e;
            e += __hip_atomic_load(&S->whist[H3_NB + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (zoned) e += __hip_atomic_load(&S->wcoarse[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // (region 3 ends with the last bin)
            hist[H3_NB + 1] = e;
            const long m = e, r = m ? (m - 1) / 2 : 0;
            int ok = m > 0 ? 1 : 0, bm = 0, js = 0;
            if (ok) {
                int a = 0, b = H3_NB + 1;                   // first bin whose inclusive prefix exceeds r
                while (a < b) { const int c = (a + b) >> 1; if ((long)hist[c] > r) b = c; else a = c + 1; }
                bm = a;
                ok = bm >= 1 && bm <= H3_NB && (hist[bm] - hist[bm - 1]) <= (unsigned)H3_CAP;
                if (zoned) ok = ok && bm > H3_ZM_LO && bm <= H3_ZM_HI;       // (bm - 1 may be the region's last bin: exact there too)
            }
            if (ok) {
                // smallest radius js (in bins) with at least r + 1 keys in bins [bm - js, bm + js]; zoned: among the radii whose
                // both ends -- and the shells and probes around them, 3 bins either way -- lie inside the outer zones
                const int jmax = (bm - 1 < H3_NB - bm) ? bm - 1 : H3_NB - bm;
                int c_lo = 0, c_hi = jmax;
                if (zoned) {
                    c_lo = (H3_ZR_LO - bm > bm - 1 - H3_ZL_HI ? H3_ZR_LO - bm : bm - 1 - H3_ZL_HI) + 4;
                    c_hi = (H3_ZR_HI - bm < bm - 1 - H3_ZL_LO ? H3_ZR_HI - bm : bm - 1 - H3_ZL_LO) - 4;
                    ok = c_lo >= 4 && c_lo < c_hi && c_hi <= jmax && (long)(hist[bm + c_lo] - hist[bm - c_lo - 1]) <= r &&
                         (long)(hist[bm + c_hi] - hist[bm - c_hi - 1]) > r;
                }
                int a = c_lo, b = c_hi + 1;
                while (ok && a < b) { const int c = (a + b) >> 1; if ((long)(hist[bm + c] - hist[bm - c - 1]) > r) b = c; else a = c + 1; }
                js = a;
                ok = ok && js >= 4 && js + 2 <= jmax;
                if (ok) {
                    // shells: the five bins at distance js - 2 .. js + 2 from the median's bin, on either side
                    const unsigned c1 = hist[bm - js + 2] - hist[bm - js - 3], c2 = hist[bm + js + 2] - hist[bm + js - 3];
                    ok = c1 + c2 <= (unsigned)H3_CAP;
                }
            }
            h3i[0] = ok; h3i[1] = bm; h3i[2] = js;
            h3i[3] = ok ? (int)(hist[bm] - hist[bm - 1]) : 0;                       // keys in the median's bin
            h3i[4] = ok ? (int)hist[bm - 1] : 0;                                      // keys below it
            h3i[5] = ok ? (int)(hist[bm + js - 3] - hist[bm - js + 2]) : 0;           // keys strictly between the shells
            h3i[6] = ok ? (int)((hist[bm - js + 2] - hist[bm - js - 3]) + (hist[bm + js + 2] - hist[bm + js - 3])) : 0;
            h3i[7] = (int)(m & 0x7fffffff);
        }
        __syncthreads();
        SICP_TQ();
        bool okw = h3i[0] != 0;
        const int bm = h3i[1], js = h3i[2], cnt_m = h3i[3], below_m = h3i[4], c_in = h3i[5], c_sh = h3i[6];
        const long mw = h3i[7], rw = mw ? (mw - 1) / 2 : 0;
        if (okw) {
            SICP_TQ();
            // ---- sweep 2: the median bin's keys, the shells' keys, the smallest key above the median's bin.  A block lists what it
            //      finds in LDS and writes the lists into a region of its own: no counter every block appends to (two thousand
            //      same-address additions at a million correspondences would serialise at ~20 ns apiece) ----
            if (tid < 2) lcnt[tid] = 0u;
            __syncthreads();
            unsigned long long nxt = ~0ull;
            for (long base = (long)blockIdx.x * (256 * HSU); base < Q; base += stride) {
                double d[HSU]; uint8_t f[HSU];
#pragma unroll
                for (int u = 0; u < HSU; ++u) {
                    const long i = base + u * 256 + tid;
                    f[u] = i < Q ? flag[i] : (uint8_t)0;
                    d[u] = i < Q ? dist[i] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < HSU; ++u) {
                    if (!f[u]) continue;
                    const int b = h3_bin(d[u], wlo, inv_bw);
                    if (b == bm) {
                        const unsigned pos = atomicAdd(&lcnt[0], 1u);
                        if (pos < (unsigned)H3_BCAP_M) lcand[pos] = d[u];
                    } else {
                        if (b > bm) { const unsigned long long k = okey(d[u]); nxt = k < nxt ? k : nxt; }
                        const int off = b < bm ? bm - b : b - bm;
                        if (off >= js - 2 && off <= js + 2) {
                            const unsigned pos = atomicAdd(&lcnt[1], 1u);
                            if (pos < (unsigned)H3_BCAP_S) lcand[H3_BCAP_M + pos] = d[u];
                        }
                    }
                }
            }
            { unsigned long long o;
              o = lane_xor64<32>(nxt); nxt = o < nxt ? o : nxt;  o = lane_xor64<16>(nxt); nxt = o < nxt ? o : nxt;
              o = lane_xor64<8>(nxt);  nxt = o < nxt ? o : nxt;  o = lane_xor64<4>(nxt);  nxt = o < nxt ? o : nxt;
              o = lane_xor64<2>(nxt);  nxt = o < nxt ? o : nxt;  o = lane_xor64<1>(nxt);  nxt = o < nxt ? o : nxt; }
            if (lane == 0) pnx[wid] = nxt;
            __syncthreads();
            if (tid == 0) {
                unsigned long long tn = pnx[0];
                for (int w = 1; w < 4; ++w) tn = pnx[w] < tn ? pnx[w] : tn;
                if (tn != ~0ull) atomicMin(&S->nxt[0], tn);
            }
            {   // this block's lists -> its region (counts beyond the capacity say so: the gather below then misses)
                const unsigned nm = lcnt[0], nsh = lcnt[1];
                if (tid < 2) __hip_atomic_store(&S->wcnt[blockIdx.x][tid], tid ? nsh : nm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned km = nm < (unsigned)H3_BCAP_M ? nm : (unsigned)H3_BCAP_M, ks = nsh < (unsigned)H3_BCAP_S ? nsh : (unsigned)H3_BCAP_S;
                if ((unsigned)tid < km) __hip_atomic_store(&S->wcand[blockIdx.x][tid], lcand[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (tid >= H3_BCAP_M && (unsigned)(tid - H3_BCAP_M) < ks)
                    __hip_atomic_store(&S->wcand[blockIdx.x][tid], lcand[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            SICP_TQ();
            grid_barrier(&S->bar, bar_base + (unsigned long long)(++nb), absent);
            SICP_TQ();
            // ---- every block gathers every block's lists: thread t owns block t's (at most 256 blocks) ----
            unsigned my_m = 0, my_s = 0;
            if ((unsigned)tid < g) {
                my_m = __hip_atomic_load(&S->wcnt[tid][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                my_s = __hip_atomic_load(&S->wcnt[tid][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const bool fits = __syncthreads_and(my_m <= (unsigned)H3_BCAP_M && my_s <= (unsigned)H3_BCAP_S) != 0;
            unsigned tot_m, tot_s;
            const unsigned off_m = block_excl_scan(my_m, &tot_m);
            __syncthreads();
            const unsigned off_s = block_excl_scan(my_s, &tot_s);
            okw = fits && tot_m == (unsigned)cnt_m && tot_s == (unsigned)c_sh;
            if (okw)
                for (unsigned i = 0; i < my_m; ++i) wk[off_m + i] = __hip_atomic_load(&S->wcand[tid][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            SICP_TQ();
            const long t = rw - below_m;                                   // the median's rank inside its bin
            double ka = 0.0, kb = 0.0;
            if (okw) okw = h3_pick(wk, cnt_m, t, !(mw & 1) && t + 1 < cnt_m, hist, lsm, ka, kb);
            if (okw) {
            if (!(mw & 1) && t + 1 >= cnt_m) kb = oval64(__hip_atomic_load(&S->nxt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const double med = (ka + kb) / 2.0;
            __syncthreads();
            // ---- exact MAD from the shells' keys.  The premise -- every key between the shells is nearer to the median than the
            //      MAD, every key outside at least as far -- is checked with four probe values, one just inside either end of either
            //      shell: binning is monotone, so a probe that h3_bin puts into a shell's first (last) bin is above (below) every
            //      key of the bins before (after) it, and |x - median| is monotone in x on either side of the median ----
            for (unsigned i = 0; i < my_s; ++i)
                wk[off_s + i] = fabs(__hip_atomic_load(&S->wcand[tid][H3_BCAP_M + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - med);
            __syncthreads();
            const double bw = 3.0 * pd1 / (double)H3_NB;
            const double xl_out = wlo + ((double)(bm - js - 3) + 0.02) * bw, xl_in = wlo + ((double)(bm - js + 2) - 0.02) * bw;
            const double xr_in = wlo + ((double)(bm + js - 3) + 0.02) * bw, xr_out = wlo + ((double)(bm + js + 2) - 0.02) * bw;
            const long jr = rw - c_in;                                     // the MAD's rank among the shells' keys, if the premise holds
            okw = jr >= 0 && jr + ((mw & 1) ? 0 : 1) < c_sh &&
                  h3_bin(xl_out, wlo, inv_bw) >= bm - js - 2 && h3_bin(xl_in, wlo, inv_bw) <= bm - js + 2 &&
                  h3_bin(xr_in, wlo, inv_bw) >= bm + js - 2 && h3_bin(xr_out, wlo, inv_bw) <= bm + js + 2 &&
                  xl_in < med && med < xr_in;                              // (an even count whose upper middle key lies far out)
            double v = 0.0, v2 = 0.0;
            if (okw) okw = h3_pick(wk, c_sh, jr, !(mw & 1), hist, lsm, v, v2);
            if (okw) {
                const double t_in = fmax(fabs(xl_in - med), fabs(xr_in - med));      // >= |d - median| of every key between the shells
                const double t_out = fmin(fabs(xl_out - med), fabs(xr_out - med));   // <= |d - median| of every key outside them
                okw = t_in < v && v2 <= t_out;
                if (okw) { val[0] = med; val[1] = (v + v2) / 2.0; m_first = (unsigned long long)mw; have = true; }
            }
            }
            __syncthreads();
        }
        // Missed (every block sees the same numbers, so every block is here): meet once more, so that nobody still reads what is
        // wiped and reset below.  (A hit needs no extra meeting: the histogram was last read before the second barrier, and the
        // counters are reset where the launch ends, as always.)
        if (!have) grid_barrier(&S->bar, bar_base + (unsigned long long)(++nb), absent);
        for (unsigned i = blockIdx.x * 256u + (unsigned)tid; i < (unsigned)(H3_NB + 2); i += g * 256u)
            __hip_atomic_store(&S->whist[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (blockIdx.x == 0 && tid < 4) __hip_atomic_store(&S->wcoarse[tid], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!have && blockIdx.x == 0 && tid == 0)      // (the general form touches it only in its collecting sweeps, behind a barrier of its own)
            __hip_atomic_store(&S->nxt[0], ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int p = 0;                                        // running pass index over both statistics: hist[p % 3]
    for (int which = 0; which < 2 && !have; ++which) {
        const double ctr = which ? val[0] : 0.0;
        unsigned long long prefix = 0, rank = 0, m = 0;
        unsigned cnt = 0;
        int fixed = 0;
        bool done = false;
        for (int pass = 0; pass < HS_PASSES && !done; ++pass, ++p) {
            const int bits = pass < 5 ? 12 : 4, shift = pass < 5 ? 52 - 12 * pass : 0;
            const unsigned mask = (1u << bits) - 1u;
            unsigned *gh = S->hist[p % 3];
            for (int i = tid; i < HS_BINS; i += 256) hist[i] = 0;
            __syncthreads();
            for (long base = (long)blockIdx.x * (256 * HSU); base < Q; base += stride) {      // block-uniform trip count
                double d[HSU]; uint8_t f[HSU];
#pragma unroll
                for (int u = 0; u < HSU; ++u) {
                    const long i = base + u * 256 + tid;
                    f[u] = i < Q ? flag[i] : (uint8_t)0;
                    d[u] = i < Q ? dist[i] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < HSU; ++u) {
                    const unsigned long long k = f[u] ? okey(which ? fabs(d[u] - ctr) : d[u]) : ~0ull;
                    bool act = f[u] && (pass == 0 || (k >> (shift + bits)) == (prefix >> (shift + bits)));
                    const unsigned bin = (unsigned)(k >> shift) & mask;
#pragma unroll
                    for (int r = 0; r < 2; ++r) {       // distances share sign and exponent: whole waves hit one bin
                        const unsigned long long am = __ballot(act);
                        if (am == 0) break;
                        const int leader = __ffsll((long long)am) - 1;
                        const unsigned b0 = (unsigned)__builtin_amdgcn_readlane((int)bin, leader);
                        const unsigned long long same = __ballot(act && bin == b0);
                        if (lane == leader) atomicAdd(&hist[b0], (unsigned)__popcll(same));
                        act = act && bin != b0;
                    }
                    if (act) atomicAdd(&hist[bin], 1u);
                }
            }
            __syncthreads();
            for (int i = tid; i < HS_BINS; i += 256) if (hist[i]) atomicAdd(&gh[i], hist[i]);
            grid_barrier(&S->bar, bar_base + (unsigned long long)(++nb), absent);
            // every block picks the bin itself: thread t owns bins 16t .. 16t+15 of the complete histogram
            unsigned h[16], mine = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) { h[j] = __hip_atomic_load(&gh[16 * tid + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); mine += h[j]; }
            const unsigned incl = wscan_u32(mine);
            if (lane == 63) scan[wid] = incl;
            __syncthreads();
            unsigned before = 0;
            for (int w = 0; w < wid; ++w) before += scan[w];
            const unsigned long long total = (unsigned long long)scan[0] + scan[1] + scan[2] + scan[3];
            if (pass == 0) { m = total; rank = m ? (m - 1) / 2 : 0; }
            unsigned long long acc = before + incl - mine;
            if (m > 0 && rank >= acc && rank < acc + mine) {
                int j = 0;
                while (rank >= acc + h[j]) { acc += h[j]; ++j; }
                sel[0] = prefix | ((unsigned long long)(16 * tid + j) << shift);
                sel[1] = rank - acc;
                sel[2] = h[j];
            }
            __syncthreads();
            if (m > 0) {
                prefix = sel[0]; rank = sel[1]; cnt = (unsigned)sel[2];
                fixed = 64 - shift;
                done = cnt <= (unsigned)HS_CAP || pass == HS_PASSES - 1;
            } else { done = true; cnt = 0; fixed = 0; }
            // wipe this block's share of the buffer two passes ahead (see the header)
            {
                unsigned *z = S->hist[(p + 2) % 3];
                for (unsigned i = blockIdx.x * 256u + (unsigned)tid; i < (unsigned)HS_BINS; i += g * 256u)
                    __hip_atomic_store(&z[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();                              // (sel / scan are rewritten by the next pass)
        }
        // ---- finish: survivors of the prefix interval, the smallest key above it ----
        const bool collect = cnt <= (unsigned)HS_CAP;     // otherwise every bit is fixed: the interval is ONE value
        const unsigned long long hi = fixed >= 64 ? prefix : (prefix | (~0ull >> fixed));
        unsigned long long nxt = ~0ull;
        if (m > 0) {
            for (long base = (long)blockIdx.x * (256 * HSU); base < Q; base += stride) {
                double d[HSU]; uint8_t f[HSU];
#pragma unroll
                for (int u = 0; u < HSU; ++u) {
                    const long i = base + u * 256 + tid;
                    f[u] = i < Q ? flag[i] : (uint8_t)0;
                    d[u] = i < Q ? dist[i] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < HSU; ++u) {
                    if (!f[u]) continue;
                    const unsigned long long k = okey(which ? fabs(d[u] - ctr) : d[u]);
                    if (k > hi) nxt = k < nxt ? k : nxt;
                    else if (collect && k >= prefix) {
                        const unsigned pos = atomicAdd(&S->ncand[which], 1u);
                        if (pos < (unsigned)HS_CAP) __hip_atomic_store(&S->cand[which][pos], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
        { unsigned long long o;
          o = lane_xor64<32>(nxt); nxt = o < nxt ? o : nxt;  o = lane_xor64<16>(nxt); nxt = o < nxt ? o : nxt;
          o = lane_xor64<8>(nxt);  nxt = o < nxt ? o : nxt;  o = lane_xor64<4>(nxt);  nxt = o < nxt ? o : nxt;
          o = lane_xor64<2>(nxt);  nxt = o < nxt ? o : nxt;  o = lane_xor64<1>(nxt);  nxt = o < nxt ? o : nxt; }
        if (lane == 0) pnx[wid] = nxt;
        __syncthreads();
        if (tid == 0) {
            unsigned long long tn = pnx[0];
            for (int w = 1; w < 4; ++w) tn = pnx[w] < tn ? pnx[w] : tn;
            if (tn != ~0ull) atomicMin(&S->nxt[which], tn);
        }
        grid_barrier(&S->bar, bar_base + (unsigned long long)(++nb), absent);
        // every block ranks the survivors itself
        const unsigned long long above = __hip_atomic_load(&S->nxt[which], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 2) pick[tid] = prefix;                  // (single-value interval: both middles are that value unless ...)
        if (collect && m > 0) {
            if (tid < (int)cnt) sc[tid] = __hip_atomic_load(&S->cand[which][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (tid < (int)cnt) {
                const unsigned long long k = sc[tid];
                unsigned r = 0;
                for (unsigned j = 0; j < cnt; ++j) { const unsigned long long o = sc[j]; r += (o < k || (o == k && j < (unsigned)tid)) ? 1u : 0u; }
                if (r == rank) pick[0] = k;
                if (r == rank + 1) pick[1] = k;
            }
        }
        __syncthreads();
        {
            const unsigned long long ka = pick[0];
            unsigned long long kb = ka;
            if (!(m & 1)) kb = (rank + 1 < cnt) ? pick[1] : above;   // even count: the next value up, inside the interval or just above it
            val[which] = m > 0 ? (oval64(ka) + oval64(kb)) / 2.0 : __builtin_nan("");
        }
        if (which == 0) m_first = m;
        // the buffer of this statistic's last pass is the one nothing has wiped yet (every block read it before the barrier above)
        if (p > 0) {
            unsigned *z = S->hist[(p - 1) % 3];
            for (unsigned i = blockIdx.x * 256u + (unsigned)tid; i < (unsigned)HS_BINS; i += g * 256u)
                __hip_atomic_store(&z[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    SICP_TQ();
    // ---- keep mask + count / mean / std of the kept distances (sums relative to the median) ----
    const double med = val[0], bound = 3 * val[1];
    double n = 0, s1 = 0, s2 = 0;
    for (long base = (long)blockIdx.x * (256 * HSU); base < Q; base += stride) {
        double d[HSU]; uint8_t f[HSU];
#pragma unroll
        for (int u = 0; u < HSU; ++u) {
            const long i = base + u * 256 + tid;
            f[u] = i < Q ? flag[i] : (uint8_t)0;
            d[u] = i < Q ? dist[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < HSU; ++u) {
            const long i = base + u * 256 + tid;
            const double e = d[u] - med;
            const bool k = f[u] && fabs(e) <= bound;
            if (i < Q) keep[i] = k ? 1 : 0;
            if (k) { n += 1.0; s1 += e; s2 += e * e; }
        }
    }
    n = wsum(n); s1 = wsum(s1); s2 = wsum(s2);
    if (lane == 0) { red[wid][0] = n; red[wid][1] = s1; red[wid][2] = s2; }
    __syncthreads();
    if (tid < 3)
        __hip_atomic_store(&partial[(long)tid * NE_MAX_GRID + blockIdx.x], (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    SICP_TQ();
    // No last meeting: whoever finishes LAST (a ticket, nobody waits) folds the blocks' partial sums -- every other block is then past
    // its last use of the shared words, which is what the barrier that stood here (~6 us of dependent device-scope round trips at
    // 256 blocks) was for.
    if (nb > HS_MAXB && tid == 0) __hip_atomic_store(&S->bar.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (budget exceeded: never, by the count above HselAll)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int is_last;
    if (tid == 0) is_last = __hip_atomic_fetch_add(&S->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g - 1u ? 1 : 0;
    __syncthreads();
    SICP_TQ();
#ifdef SICP_HSEL_DEBUG
    if (is_last && tid == 0 && have)
        printf("[hsel3-t] sweep1+flush %lld | B1 %lld | analysis %lld | sweep2+lists %lld (incl. stamp) | B2 %lld | gather %lld | sorts+premise %lld | sweep3 %lld | ticket %lld\n",
               tq[1] - tq[0], tq[2] - tq[1], tq[3] - tq[2], tq[5] - tq[3], tq[6] - tq[5], tq[7] - tq[6], tq[8] - tq[7], tq[9] - tq[8], tq[10] - tq[9]);
#endif
    if (is_last) {
        if (tid == 0) __hip_atomic_store(&S->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-armed for the next launch
        if (wid < 3) {
            double t = 0;
            for (unsigned blk = lane; blk < g; blk += 64)
                t += __hip_atomic_load(&partial[(long)wid * NE_MAX_GRID + blk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            t = wsum(t);
            if (lane == 0) red[0][wid] = t;
        }
        __syncthreads();
        if (tid == 0) {
            const bool bad = __hip_atomic_load(&S->bar.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            const double cnt = bad ? 0.0 : red[0][0], mu = red[0][1] / cnt;
            const double var = red[0][2] / cnt - mu * mu;
            const double mean = med + mu, sd = sqrt(var > 0.0 ? var : 0.0);
            out4[0] = bad ? -1.0 : (double)m_first;       // (negative: this launch's barrier failed -- the solver's finish reports it)
            out4[1] = bad ? __builtin_nan("") : med; out4[2] = bad ? __builtin_nan("") : val[1]; out4[3] = cnt;
            out3[0] = cnt; out3[1] = mean; out3[2] = sd;
            if (host_out) {
                host_out[0] = out4[0]; host_out[1] = out4[1]; host_out[2] = out4[2]; host_out[3] = cnt;
                host_out[4] = cnt; host_out[5] = mean; host_out[6] = sd;
                __threadfence_system();
                __hip_atomic_store(host_out + 15, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            // what the next launch's window is centred on (kept for two launches: it is only tried when they agree)
            if (!bad && val[1] == val[1]) {
                S->prior[0][0] = S->prior[1][0]; S->prior[0][1] = S->prior[1][1];
                S->prior[1][0] = med; S->prior[1][1] = val[1];
                // (a window that was tried and missed costs its phases: the next two launches take the general form)
                S->n_prior = (try_w && !have) ? 1u : (n_prior < 2u ? n_prior + 1u : 2u);
            } else {
                S->n_prior = 0u;
            }
            // leave the state as the next launch expects it (every other block is past its last use of these words)
            __hip_atomic_store(&S->nxt[0], ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&S->nxt[1], ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&S->ncand[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&S->ncand[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Blocks of `kernel` (block size `threads`, no dynamic LDS) the CURRENT device holds at once: CUs x blocks per CU, looked up once
// per device and kernel -- a process may own contexts on devices of different sizes or partition modes, and a grid barrier
// launched with more blocks than are co-resident cannot meet.
long resident_blocks(const void *kernel, int threads)
{
    static std::mutex m;
    static std::map<std::pair<int, const void *>, long> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    std::lock_guard<std::mutex> g(m);
    auto it = cache.find({dev, kernel});
    if (it != cache.end()) return it->second;
    int cus = 0, per_cu = 0;
    long r = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) == hipSuccess && cus >= 1 && per_cu >= 1)
        r = (long)cus * per_cu;
    cache[{dev, kernel}] = r;
    return r;
}

size_t reject_select_scratch_bytes() { return sizeof(HselAll); }

// a new (or re-used) state buffer of the one-launch form: all zero, the two `nxt` words at ~0
hipError_t hsel_state_init(hipStream_t s, void *state)
{
    hipError_t e = hipMemsetAsync(state, 0, reject_select_scratch_bytes(), s);
    if (e != hipSuccess) return e;
    return hipMemsetAsync(&((HselAll *)state)->nxt[0], 0xff, 2 * sizeof(unsigned long long), s);
}

// One-launch form.  `state` must have been zeroed (hipMemset) when it was allocated and is left clean by every launch;
// *bar_total is the host's running count of what the launches on this buffer have added to its barrier counter.
hipError_t reject_by_select_one_launch(hipStream_t s, const double *dist, const uint8_t *flag, long Q, uint8_t *keep, double *out4,
                                       double *out3, void *state, unsigned long long *bar_total, double *partial, double *host_out,
                                       double seq, const IcpDev *st, unsigned absent, bool use_prior)
{
    const long cap = 256;
    // every block must be resident at once (grid barrier): never more blocks than the device can hold (a partitioned device has
    // far fewer CUs than 256)
    const int hsu = Q >= 900000 ? 16 : (Q >= 450000 ? 8 : 4);
    const void *fn = hsu == 16 ? (const void *)k_hsel_all<16> : hsu == 8 ? (const void *)k_hsel_all<8> : (const void *)k_hsel_all<4>;
    const long resident = resident_blocks(fn, 256);
    const unsigned g = (unsigned)std::max<long>(1, std::min<long>(std::min<long>(cap, resident), (Q + 256 * hsu - 1) / (256 * hsu)));
#define SICP_HSEL_LAUNCH(U)                                                                                                       \
    hipLaunchKernelGGL((k_hsel_all<U>), dim3(g), dim3(256), 0, s, dist, flag, Q, (HselAll *)state, *bar_total, keep, partial, out4, out3, \
                       host_out, seq, st, absent, use_prior ? 1 : 0)
    if (hsu == 16) SICP_HSEL_LAUNCH(16);
    else if (hsu == 8) SICP_HSEL_LAUNCH(8);
    else SICP_HSEL_LAUNCH(4);
#undef SICP_HSEL_LAUNCH
    *bar_total += (unsigned long long)HS_MAXB;
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------

void launch_cloud_stats(hipStream_t s, const double *x, const double *y, const double *z, long n, unsigned long long *out7)
{
    long g = (n + 1023) / 1024;
    if (g > 1024) g = 1024;             // 7 same-address atomics per block: few blocks, long grid-stride loops
    hipLaunchKernelGGL(k_cloud_stats, dim3((unsigned)g), dim3(256), 0, s, x, y, z, n, out7);
}

void launch_cell_ids(hipStream_t s, const double *x, const double *y, const double *z, long n, const GridGeom &G,
                     uint32_t *ids, uint32_t *counts, unsigned long long *occupied)
{
    if (occupied) hipLaunchKernelGGL(k_cell_ids<true>, dim3(cdiv(n, 256)), dim3(256), 0, s, x, y, z, n, G, ids, counts, occupied);
    else hipLaunchKernelGGL(k_cell_ids<false>, dim3(cdiv(n, 256)), dim3(256), 0, s, x, y, z, n, G, ids, counts, occupied);
}

void launch_cell_ids_tiled(hipStream_t s, const double *x, const double *y, const double *z, long n, const GridGeom &G,
                           uint32_t *ids, uint32_t *counts)
{
    hipLaunchKernelGGL(k_cell_ids_tiled, dim3(cdiv(n, 256)), dim3(256), 0, s, x, y, z, n, G, ids, counts);
}

void launch_window_probe(hipStream_t s, const double *x, const double *y, const double *z, long n, long every, const GridGeom &Gw,
                         const double wmax[3], uint32_t *counts, unsigned long long *out2)
{
    const long chunks = (n + 1023) / 1024;
    long g = (chunks + every - 1) / every;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(k_window_probe, dim3((unsigned)g), dim3(256), 0, s, x, y, z, n, every, Gw, wmax[0], wmax[1], wmax[2],
                       counts, out2);
}

long grid_scan_blocks(long n) { return (n + 1 + SCAN_ITEMS - 1) / SCAN_ITEMS; }

// out[0..n] = exclusive prefix sums of in[0..n) (entry n = total); block_off: grid_scan_blocks(n) words of scratch
void launch_grid_scan(hipStream_t s, const uint32_t *in, long n, uint32_t *block_off, uint32_t *out, uint32_t *cursor)
{
    launch_grid_scan_sums(s, in, n, block_off, nullptr);
    launch_grid_scan_rest(s, in, n, block_off, out, cursor);
}
// the same in two steps: the block sums first (with the sum of squares of the counts, for a caller that still has to decide whether
// this binning stands), the rest once it does
void launch_grid_scan_sums(hipStream_t s, const uint32_t *in, long n, uint32_t *block_off, unsigned long long *sumsq)
{
    hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)grid_scan_blocks(n)), dim3(256), 0, s, in, n, block_off, sumsq);
}
void launch_grid_scan_rest(hipStream_t s, const uint32_t *in, long n, uint32_t *block_off, uint32_t *out, uint32_t *cursor)
{
    const long nb = grid_scan_blocks(n);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, s, block_off, nb);
    hipLaunchKernelGGL(k_scan_final, dim3((unsigned)nb), dim3(256), 0, s, in, n, (const uint32_t *)block_off, out, cursor);
}

void launch_scatter(hipStream_t s, const double *x, const double *y, const double *z, const uint32_t *ids, long n,
                    uint32_t *cursor, void *rec)
{
    hipLaunchKernelGGL(k_scatter, dim3(cdiv(n, 256)), dim3(256), 0, s, x, y, z, ids, n, cursor, (double4 *)rec);
}

void launch_grid_nn(hipStream_t s, const double *qx, const double *qy, const double *qz, long Q, const double *prev_p2,
                    const GridGeom &G, const uint32_t *cell_start, const void *rec, const Xf *H, const Xf *Hinv, double rmax,
                    double max_d2, int64_t idx_base, double *d2_out, int64_t *idx_out, double *p2_out, unsigned long long *work,
                    bool four_per_wave, const unsigned long long *cell_box, const GridLevel *coarse)
{
    const dim3 grid(cdiv(Q, 4)), block(256);
    Xf id = {};
    const GridGeom G2 = coarse ? coarse->g : G;
    const uint32_t *cs2 = coarse ? coarse->cell_start : nullptr;
    const double4 *rec2 = coarse ? (const double4 *)coarse->rec : nullptr;
    const bool ext = cell_box != nullptr || coarse != nullptr;
    const uint32_t *no_list = nullptr;
    unsigned *no_count = nullptr;
    if (four_per_wave) {
        const dim3 g16(cdiv(Q, 16));
        if (H)
            hipLaunchKernelGGL((k_grid_nn16<true, false, 16>), g16, block, 0, s, (const IcpDev *)nullptr, qx, qy, qz, prev_p2, cell_start, (const double4 *)rec, (const uint32_t *)nullptr, Q, G, *H, *Hinv, rmax, max_d2, idx_base, d2_out, idx_out, p2_out, work, 0, PostMatch{});
        else
            hipLaunchKernelGGL((k_grid_nn16<false, false, 16>), g16, block, 0, s, (const IcpDev *)nullptr, qx, qy, qz, prev_p2, cell_start, (const double4 *)rec, (const uint32_t *)nullptr, Q, G, id, id, rmax, max_d2, idx_base, d2_out, idx_out, p2_out, work, 0, PostMatch{});
        return;
    }
    if (H)
        if (ext) hipLaunchKernelGGL((k_grid_nn<true, false, true>), grid, block, 0, s, (const IcpDev *)nullptr, qx, qy, qz, prev_p2, cell_start, (const double4 *)rec, (const uint32_t *)nullptr, Q, G, *H,
                           *Hinv, rmax, max_d2, idx_base, d2_out, idx_out, p2_out, work, 0, PostMatch{}, cell_box, G2, cs2, rec2, no_list, no_count, no_count);
        else hipLaunchKernelGGL((k_grid_nn<true, false, false>), grid, block, 0, s, (const IcpDev *)nullptr, qx, qy, qz, prev_p2, cell_start, (const double4 *)rec, (const uint32_t *)nullptr, Q, G, *H,
                           *Hinv, rmax, max_d2, idx_base, d2_out, idx_out, p2_out, work, 0, PostMatch{}, cell_box, G2, cs2, rec2, no_list, no_count, no_count);
    else
        if (ext) hipLaunchKernelGGL((k_grid_nn<false, false, true>), grid, block, 0, s, (const IcpDev *)nullptr, qx, qy, qz, prev_p2, cell_start, (const double4 *)rec, (const uint32_t *)nullptr, Q, G, id,
                           id, rmax, max_d2, idx_base, d2_out, idx_out, p2_out, work, 0, PostMatch{}, cell_box, G2, cs2, rec2, no_list, no_count, no_count);
        else hipLaunchKernelGGL((k_grid_nn<false, false, false>), grid, block, 0, s, (const IcpDev *)nullptr, qx, qy, qz, prev_p2, cell_start, (const double4 *)rec, (const uint32_t *)nullptr, Q, G, id,
                           id, rmax, max_d2, idx_base, d2_out, idx_out, p2_out, work, 0, PostMatch{}, cell_box, G2, cs2, rec2, no_list, no_count, no_count);
}

// the match of a chained iteration: transform taken from the loop state on the device
void launch_grid_nn_chained(hipStream_t s, const double *qx, const double *qy, const double *qz, long Q, const double *prev_p2,
                            const GridGeom &G, const uint32_t *cell_start, const void *rec, const IcpDev *st, double rmax,
                            int64_t idx_base, double *d2_out, int64_t *idx_out, double *p2_out, unsigned long long *work,
                            const uint32_t *order, bool four_per_wave, int flags, const PostMatch *post, bool eight_per_wave,
                            const unsigned long long *cell_box, const GridLevel *coarse)
{
    Xf id = {};
    const GridGeom G2 = coarse ? coarse->g : G;
    const uint32_t *cs2 = coarse ? coarse->cell_start : nullptr;
    const double4 *rec2 = coarse ? (const double4 *)coarse->rec : nullptr;
    PostMatch pm = {};
    if (post) pm = *post;
    if (four_per_wave) {
        unsigned g16 = cdiv(Q, 16);
        if (order) g16 = (g16 + 7u) & ~7u;
        if (eight_per_wave) {
            unsigned g8 = cdiv(Q, 32);
            if (order) g8 = (g8 + 7u) & ~7u;
            hipLaunchKernelGGL((k_grid_nn16<true, true, 8>), dim3(g8), dim3(256), 0, s, st, qx, qy, qz, prev_p2, cell_start, (const double4 *)rec, order, Q, G, id, id, rmax, (double)__builtin_inf(), idx_base, d2_out, idx_out, p2_out, work, flags & NN_TIGHT, pm);
            return;
        }
        hipLaunchKernelGGL((k_grid_nn16<true, true, 16>), dim3(g16), dim3(256), 0, s, st, qx, qy, qz, prev_p2, cell_start, (const double4 *)rec, order, Q, G, id, id, rmax, (double)__builtin_inf(), idx_base, d2_out, idx_out, p2_out, work, flags & NN_TIGHT, pm);
        return;
    }
    unsigned g = cdiv(Q, 4);
    if (order) g = (g + 7u) & ~7u;
    // (the plain instantiation unless an option is live: see the kernel's EXT parameter)
    if (cell_box != nullptr || coarse != nullptr)
        hipLaunchKernelGGL((k_grid_nn<true, true, true>), dim3(g), dim3(256), 0, s, st, qx, qy, qz, prev_p2, cell_start, (const double4 *)rec, order, Q, G, id, id, rmax, (double)__builtin_inf(), idx_base, d2_out, idx_out, p2_out, work, flags, pm,
                           cell_box, G2, cs2, rec2, (const uint32_t *)nullptr, (const unsigned *)nullptr, (unsigned *)nullptr);
    else
        hipLaunchKernelGGL((k_grid_nn<true, true, false>), dim3(g), dim3(256), 0, s, st, qx, qy, qz, prev_p2, cell_start, (const double4 *)rec, order, Q, G, id, id, rmax, (double)__builtin_inf(), idx_base, d2_out, idx_out, p2_out, work, flags, pm,
                           cell_box, G2, cs2, rec2, (const uint32_t *)nullptr, (const unsigned *)nullptr, (unsigned *)nullptr);
}

// The exact one-wave-per-query search over a LIST of queries: what the filtered many-queries kernel (sicp_gridf.hip) would not
// answer itself.  The launch cannot know how long the list is -- a few hundred waves share it; an empty list costs a launch that
// exits at once.  st: the chain's loop state (then H, Hinv are ignored), else H (nullable: no transform).
void launch_grid_nn_redo(hipStream_t s, const double *qx, const double *qy, const double *qz, long Q, const double *prev_p2,
                         const GridGeom &G, const uint32_t *cell_start, const void *rec, const IcpDev *st, const Xf *H, const Xf *Hinv,
                         double rmax, double max_d2, int64_t idx_base, double *d2_out, int64_t *idx_out, double *p2_out,
                         unsigned long long *work, int flags, const PostMatch *post, const unsigned long long *cell_box,
                         const uint32_t *redo_list, const unsigned *redo_count, unsigned *redo_clear, const GridLevel *coarse)
{
    Xf id = {};
    const GridGeom G2 = coarse ? coarse->g : G;
    const uint32_t *cs2 = coarse ? coarse->cell_start : nullptr;
    const double4 *rec2 = coarse ? (const double4 *)coarse->rec : nullptr;
    PostMatch pm = {};
    if (post) pm = *post;
    long want = Q / 256;                                   // ~1.5 % of the queries at one per wave before the waves loop
    const unsigned g = (unsigned)(want < 8 ? 8 : (want > 1024 ? 1024 : want));
    const uint32_t *no_order = nullptr;
    if (st)
        hipLaunchKernelGGL((k_grid_nn<true, true, true>), dim3(g), dim3(256), 0, s, st, qx, qy, qz, prev_p2, cell_start, (const double4 *)rec, no_order, Q, G, id, id, rmax, max_d2, idx_base, d2_out, idx_out, p2_out, work, flags, pm, cell_box, G2, cs2, rec2, redo_list, redo_count, redo_clear);
    else if (H)
        hipLaunchKernelGGL((k_grid_nn<true, false, true>), dim3(g), dim3(256), 0, s, st, qx, qy, qz, prev_p2, cell_start, (const double4 *)rec, no_order, Q, G, *H, *Hinv, rmax, max_d2, idx_base, d2_out, idx_out, p2_out, work, flags, pm, cell_box, G2, cs2, rec2, redo_list, redo_count, redo_clear);
    else
        hipLaunchKernelGGL((k_grid_nn<false, false, true>), dim3(g), dim3(256), 0, s, st, qx, qy, qz, prev_p2, cell_start, (const double4 *)rec, no_order, Q, G, id, id, rmax, max_d2, idx_base, d2_out, idx_out, p2_out, work, flags, pm, cell_box, G2, cs2, rec2, redo_list, redo_count, redo_clear);
}

void launch_stride_sample(hipStream_t s, const double *x, const double *y, const double *z, long n, long stride, long m, long mpad,
                          double *out)
{
    hipLaunchKernelGGL(k_stride_sample, dim3(cdiv(mpad, 256)), dim3(256), 0, s, x, y, z, n, stride, m, mpad, out);
}

void launch_scatter_order(hipStream_t s, const uint32_t *ids, long n, uint32_t *cursor, uint32_t *order)
{
    hipLaunchKernelGGL(k_scatter_order, dim3(cdiv(n, 256)), dim3(256), 0, s, ids, n, cursor, order);
}

void launch_grid_knn(hipStream_t s, const double *qx, const double *qy, const double *qz, long Q, int k, const GridGeom &G,
                     const uint32_t *cell_start, const void *rec, double rmax, int64_t idx_base, double *d2_out, int64_t *idx_out)
{
    hipLaunchKernelGGL(k_grid_knn, dim3(cdiv(Q, 4)), dim3(256), 0, s, qx, qy, qz, Q, k, G, cell_start, (const double4 *)rec,
                       rmax, idx_base, d2_out, idx_out);
}

bool grid_knn_sweep_handles(int k) { return k >= 1 && k <= KS_MAXK; }

// the one-sweep k-NN (k <= KS_MAXK); with `cov` ((Q, 6) doubles of scratch) also covariance + eigen-decomposition -> normals,
// planarity.  avg_per_cell: points per occupied cell of the grid (sets the first radius: a ball that is expected to hold ~1.8 k
// points of a surface)
void launch_grid_knn_sweep(hipStream_t s, const double *qx, const double *qy, const double *qz, const uint32_t *order, long Q, int k,
                           const GridGeom &G, double avg_per_cell, const uint32_t *cell_start, const void *rec, double rmax,
                           int64_t idx_base, double *d2_out, int64_t *idx_out, double *cov, float *normals, float *planarity,
                           unsigned long long *work, long batch_override, uint32_t *redo /* Q + 1 words of scratch, or null */, int group)
{
    const double ppc = avg_per_cell >= 1.0 ? avg_per_cell : 1.0;
    double r_first = 1.35 * G.h * std::sqrt((double)k / (3.141592653589793 * ppc));
    if (!(r_first > 0.0) || !std::isfinite(r_first)) r_first = G.h;
    const int kpad = (k + 7) & ~7;
    double *cov_out = normals ? cov : nullptr;
    // four queries per wave for the common case + the one-query-per-wave kernel for what it leaves (k <= 32, enough queries to
    // fill the machine with groups); otherwise the one-query-per-wave kernel for everything
    const bool four = redo && k <= KG_MAXK && (group == 4 || (group == 0 && Q >= 32768));
    const uint32_t *redo_list = nullptr;
    const unsigned *redo_count = nullptr;
    if (four) {
        // slots per group: what a workgroup's 16 groups hold at once is the L2's working set -- measured at 1 M queries on 10 M points
        // (time / HBM bytes fetched): 1 -> 0.96 ms / 0.89 GB, 2 -> 0.92 / 0.93, 4 -> 0.92 / 1.60, 8 -> 0.99 / 3.0, 16 -> 1.05
        long batch = batch_override > 0 ? batch_override : 2;
        batch = batch < 1 ? 1 : (batch > 16 ? 16 : batch);
        unsigned g4 = cdiv(Q, 16 * batch);
        if (order) g4 = (g4 + 7u) & ~7u;
        (void)hipMemsetAsync(redo, 0, sizeof(uint32_t), s);
        hipLaunchKernelGGL(k_grid_knn_sweep4, dim3(g4), dim3(256), 16 * kg_group_doubles(kpad) * sizeof(double), s, qx, qy, qz, order, cell_start,
                           (const double4 *)rec, Q, k, (int)batch, G, rmax, r_first, idx_base, d2_out, idx_out, cov_out, redo + 1,
                           (unsigned *)redo, work);
        redo_list = redo + 1; redo_count = (const unsigned *)redo;
    }
    // a wave stays with a neighbourhood for a few queries once there are enough waves to fill the machine several times over
    long batch = batch_override > 0 ? batch_override : (Q >= 131072 ? 8 : (Q >= 32768 ? 4 : 1));
    batch = batch < 1 ? 1 : (batch > 64 ? 64 : batch);
    unsigned g = four ? cdiv(Q, 256) : cdiv(Q, 4 * batch);                 // (redo: Q / 64 waves cover the list whatever its length)
    if (order && !four) g = (g + 7u) & ~7u;
#define SICP_KS_LAUNCH(NS)                                                                                                     \
    hipLaunchKernelGGL((k_grid_knn_sweep<NS>), dim3(g), dim3(256), 4 * ks_wave_doubles(64 * NS, kpad) * sizeof(double), s, qx, qy, qz, order, \
                       cell_start, (const double4 *)rec, Q, k, (int)batch, G, rmax, r_first, idx_base, d2_out, idx_out, cov_out, work,   \
                       redo_list, redo_count)
    if (k <= 32) SICP_KS_LAUNCH(2);
    else if (k <= 64) SICP_KS_LAUNCH(4);
    else SICP_KS_LAUNCH(8);
#undef SICP_KS_LAUNCH
    if (cov_out) hipLaunchKernelGGL(k_cov_normals, dim3(cdiv(Q, 64)), dim3(64), 0, s, (const double *)cov_out, order, Q, normals, planarity);
}

}  // namespace sicp
