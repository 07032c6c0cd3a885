// sicp_grid.hip -- pruned EXACT 1-NN on a static uniform grid.
//
// The searched (movable) cloud never moves in its own frame, so it is binned ONCE per upload:
// cell id per point -> stable radix sort (hipCUB; a library primitive used only in this one-off
// build) -> cell offsets by histogram + exclusive scan -> coordinates gathered into cell order.
// Per iteration each query is pulled back into the cloud's frame with the rigid inverse of H and
// ONE WAVE enumerates the cells that intersect a ball around it; every candidate is evaluated with
// the exact arithmetic contract (T)+(D) from its original coordinates and compared
// lexicographically on (d2, original index).  The ball radius comes from an exact upper bound of
// the answer (previous iteration's match re-evaluated under the new H) or from an expanding search;
// termination needs best <= radius minus a slack that covers the rounding of H^-1 q and the
// non-orthogonality of the floating-point R, so the result equals the brute-force scan's, bit for
// bit (tests compare the two on full-size inputs).  The reference instead rebuilds a cKDTree on
// the transformed cloud every iteration (corrpts.py:131, simpleicp.py:188-202).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "sicp_internal.h"
#include "sicp_lanes.h"

namespace sicp {

static inline unsigned cdiv(long a, long b) { return (unsigned)((a + b - 1) / b); }

__device__ __forceinline__ unsigned long long okey(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}

// out[0..2] = min keys, out[3..5] = max keys (ordered-uint64 image of the doubles)
__global__ void k_bbox(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z, long n,
                       unsigned long long *__restrict__ out)
{
    double lo[3] = {__builtin_inf(), __builtin_inf(), __builtin_inf()};
    double hi[3] = {-__builtin_inf(), -__builtin_inf(), -__builtin_inf()};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double v[3] = {x[i], y[i], z[i]};
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = v[a] < lo[a] ? v[a] : lo[a]; hi[a] = v[a] > hi[a] ? v[a] : hi[a]; }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double l = __shfl_down(lo[a], off, 64), h = __shfl_down(hi[a], off, 64);
            lo[a] = l < lo[a] ? l : lo[a]; hi[a] = h > hi[a] ? h : hi[a];
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(out + a, okey(lo[a])); atomicMax(out + 3 + a, okey(hi[a])); }
    }
}

__device__ __forceinline__ int cell_coord(double v, double mn, double inv_h, int dim)
{
    int c = (int)floor((v - mn) * inv_h);
    return c < 0 ? 0 : (c >= dim ? dim - 1 : c);
}

__global__ void k_cell_ids(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                           long n, GridGeom G, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
                           uint32_t *__restrict__ counts)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int cx = cell_coord(x[i], G.mn[0], G.inv_h, G.dim[0]);
    const int cy = cell_coord(y[i], G.mn[1], G.inv_h, G.dim[1]);
    const int cz = cell_coord(z[i], G.mn[2], G.inv_h, G.dim[2]);
    const uint32_t id = ((uint32_t)cz * G.dim[1] + cy) * G.dim[0] + cx;
    keys[i] = id; vals[i] = (uint32_t)i;
    atomicAdd(counts + id, 1u);
}

__global__ void k_gather_sorted(const double *__restrict__ x, const double *__restrict__ y,
                                const double *__restrict__ z, const uint32_t *__restrict__ sidx, long n,
                                double *__restrict__ sx, double *__restrict__ sy, double *__restrict__ sz)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = sidx[i];
    sx[i] = x[s]; sy[i] = y[s]; sz[i] = z[s];
}

// number of non-empty cells (to judge the cell size)
__global__ void k_count_nonempty(const uint32_t *__restrict__ counts, long ncells, unsigned long long *out)
{
    unsigned long long c = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < ncells; i += (long)gridDim.x * blockDim.x)
        c += counts[i] ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// ------------------------------------------------------------------------------------
// one wave per query
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void xf(const Xf &H, double x, double y, double z, double &ox, double &oy, double &oz)
{
    double t;
    t = H.m[0] * x;  t = fma(H.m[1], y, t);  t = fma(H.m[2], z, t);   ox = t + H.m[3];
    t = H.m[4] * x;  t = fma(H.m[5], y, t);  t = fma(H.m[6], z, t);   oy = t + H.m[7];
    t = H.m[8] * x;  t = fma(H.m[9], y, t);  t = fma(H.m[10], z, t);  oz = t + H.m[11];
}

// CHAINED: the launch belongs to a run whose iterations are enqueued back to back -- H and its inverse come from
// the device-resident loop state the previous tail launch left (sicp_tail.hip), and the launch exits at once
// when that tail declared the run over.
template <bool XFORM, bool CHAINED>
__global__ __launch_bounds__(256) void k_grid_nn(
    const double *__restrict__ qx, const double *__restrict__ qy, const double *__restrict__ qz, long Q,
    const double *__restrict__ prev_p2 /* nullable: (Q,3) a cloud point per query (last match) -> its exact
                                          distance under H bounds the answer; saves the expanding search */,
    GridGeom G, const uint32_t *__restrict__ cell_start, const double *__restrict__ sx,
    const double *__restrict__ sy, const double *__restrict__ sz, const uint32_t *__restrict__ sidx,
    Xf H, Xf Hinv, double rmax, double max_d2, int64_t idx_base,
    double *__restrict__ d2_out, int64_t *__restrict__ idx_out, double *__restrict__ p2_out,
    const IcpDev *__restrict__ st)
{
    const int lane = threadIdx.x & 63;
    const long q = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (q >= Q) return;                                   // whole wave leaves together
    const double ax = qx[q], ay = qy[q], az = qz[q];      // (issued before the loop state is waited for)
    if (CHAINED) {
        H = st->H; Hinv = st->Hinv;
        if (st->stop) return;
    }
    double cxq = ax, cyq = ay, czq = az;                  // query in the cloud's own frame
    if (XFORM) xf(Hinv, ax, ay, az, cxq, cyq, czq);
    // covers rounding of H^-1 q and |R^T R - I| ~ 1e-16: distances in the two frames agree to
    // ~1e-15 * scale; 1e-12 * scale leaves three orders of magnitude
    const double scale = rmax + sqrt(fma(czq, czq, fma(cyq, cyq, cxq * cxq))) + 1.0;
    const double slack = 1e-12 * scale;
    const double r_cap = (max_d2 < __builtin_inf()) ? sqrt(max_d2) * (1.0 + 1e-12) + slack : __builtin_inf();
    double bnd = __builtin_inf();
    if (prev_p2) {
        double X = prev_p2[3 * q], Y = prev_p2[3 * q + 1], Z = prev_p2[3 * q + 2];
        if (XFORM) { double u, v, w; xf(H, X, Y, Z, u, v, w); X = u; Y = v; Z = w; }
        const double dx = X - ax, dy = Y - ay, dz = Z - az;
        bnd = fma(dz, dz, fma(dy, dy, dx * dx));
    }
    const bool has_bound = bnd < __builtin_inf();
    double r = has_bound ? sqrt(bnd) * (1.0 + 1e-12) + slack : 0.5 * G.h;
    if (r > r_cap) r = r_cap;

    double best = __builtin_inf();
    uint32_t bidx = 0xffffffffu, bpos = 0;
    for (int pass = 0; pass < 64; ++pass) {
        int lo[3], hi[3];
        const double c3[3] = {cxq, cyq, czq};
        bool all = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double fl = floor((c3[a] - r - G.mn[a]) * G.inv_h - 1e-6);
            const double fh = floor((c3[a] + r - G.mn[a]) * G.inv_h + 1e-6);
            lo[a] = fl < 0.0 ? 0 : (fl > (double)(G.dim[a] - 1) ? G.dim[a] - 1 : (int)fl);
            hi[a] = fh < 0.0 ? 0 : (fh > (double)(G.dim[a] - 1) ? G.dim[a] - 1 : (int)fh);
            // whole axis covered <=> the ball reaches past both faces of the box
            all = all && (fl <= 0.0) && (fh >= (double)(G.dim[a] - 1));
        }
        best = __builtin_inf(); bidx = 0xffffffffu; bpos = 0;
        const int ny = hi[1] - lo[1] + 1, nz = hi[2] - lo[2] + 1;
        const long nrows = (long)ny * nz;
        // cells of one (cy, cz) row are contiguous in the sorted order: lane r fetches row r's point
        // range, a wave scan turns up to 64 ranges into one flat candidate list, and the lanes stride
        // over it -- two dependent memory round trips per batch instead of two per row
        for (long rb = 0; rb < nrows; rb += 64) {
            uint32_t b = 0, len = 0;
            if (rb + lane < nrows) {
                const long rr = rb + lane;
                const int cy = lo[1] + (int)(rr % ny), cz = lo[2] + (int)(rr / ny);
                const long row = ((long)cz * G.dim[1] + cy) * G.dim[0];
                b = cell_start[row + lo[0]];
                len = cell_start[row + hi[0] + 1] - b;
            }
            const uint32_t incl = wscan_u32(len);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            // four flat candidates per lane per round: their (row, offset) searches and coordinate loads
            // are independent, so four memory round trips overlap (wave-uniform trip count: the
            // shuffles need all lanes)
            for (uint32_t base = 0; base < total; base += 256) {
                uint32_t pos[4]; bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t k = base + 64 * u + lane;
                    int r0 = 0;                                 // first row whose inclusive offset exceeds k
#pragma unroll
                    for (int step = 32; step > 0; step >>= 1) {
                        const uint32_t v = __shfl(incl, r0 + step - 1, 64);
                        if (v <= k) r0 += step;
                    }
                    r0 = r0 > 63 ? 63 : r0;
                    const uint32_t rbeg = __shfl(b, r0, 64), ri = __shfl(incl, r0, 64), rl = __shfl(len, r0, 64);
                    ok[u] = k < total;
                    pos[u] = ok[u] ? rbeg + (k - (ri - rl)) : 0u;
                }
                double X[4], Y[4], Z[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { X[u] = sx[pos[u]]; Y[u] = sy[pos[u]]; Z[u] = sz[pos[u]]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (!ok[u]) continue;
                    double px = X[u], py = Y[u], pz = Z[u];
                    if (XFORM) { double a, bq, cq; xf(H, px, py, pz, a, bq, cq); px = a; py = bq; pz = cq; }
                    const double dx = px - ax, dy = py - ay, dz = pz - az;
                    const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
                    if (d2 <= best) {
                        const uint32_t oi = sidx[pos[u]];
                        if (d2 < best || oi < bidx) { best = d2; bidx = oi; bpos = pos[u]; }
                    }
                }
            }
        }
        // wave-wide lexicographic (d2, original index) minimum: DPP butterfly, every lane ends up with it
#define SICP_LEXMIN_STEP(J)                                                                       \
        {                                                                                         \
            const double od = lane_xor_f64<J>(best);                                              \
            const uint32_t oi = lane_xor32<J>(bidx), op = lane_xor32<J>(bpos);                    \
            if (od < best || (od == best && oi < bidx)) { best = od; bidx = oi; bpos = op; }      \
        }
        SICP_LEXMIN_STEP(32) SICP_LEXMIN_STEP(16) SICP_LEXMIN_STEP(8) SICP_LEXMIN_STEP(4) SICP_LEXMIN_STEP(2) SICP_LEXMIN_STEP(1)
#undef SICP_LEXMIN_STEP
        const bool found = bidx != 0xffffffffu;
        const double r_eff = (r - slack) / (1.0 + 1e-12);
        if (found && sqrt(best) <= r_eff) break;          // nothing outside the ball can beat or tie it
        if (all || r >= r_cap || has_bound) break;        // searched everything that may qualify
        r = found ? sqrt(best) * (1.0 + 1e-12) + slack : 2.0 * r;
        if (r > r_cap) r = r_cap;
    }
    if (lane == 0) {
        const bool ok = (bidx != 0xffffffffu) && (best < max_d2);
        d2_out[q] = ok ? best : __builtin_inf();
        idx_out[q] = ok ? idx_base + (int64_t)bidx : (int64_t)-1;
        if (p2_out) {
            p2_out[3 * q]     = ok ? sx[bpos] : 0.0;
            p2_out[3 * q + 1] = ok ? sy[bpos] : 0.0;
            p2_out[3 * q + 2] = ok ? sz[bpos] : 0.0;
        }
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
    GridGeom G, const uint32_t *__restrict__ cell_start, const double *__restrict__ sx,
    const double *__restrict__ sy, const double *__restrict__ sz, const uint32_t *__restrict__ sidx,
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
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
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
                    const double dx = sx[i] - ax, dy = sy[i] - ay, dz = sz[i] - az;
                    const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
                    if (d2 <= best && (first || d2 >= fd)) {
                        const uint32_t oi = sidx[i];
                        const bool above = first || d2 > fd || oi > fi;
                        if (above && (d2 < best || oi < bidx)) { best = d2; bidx = oi; }
                    }
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double od = __shfl_xor(best, off, 64);
                const uint32_t oi = __shfl_xor(bidx, off, 64);
                if (od < best || (od == best && oi < bidx)) { best = od; bidx = oi; }
            }
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
// median / raw-MAD rejection for LARGE Q (corrpts.py:165-188): exact order statistics by radix
// SELECTION over many workgroups on the order-preserving uint64 image of the distances, everything
// chained on the stream without a host round trip.  out4 = (m, median, mad, n_kept) like k_reject.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ double oval64(unsigned long long k)
{
    const unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// one atomic per BLOCK: 16 k waves adding to one word serialise in L2 for hundreds of microseconds at Q = 1 M
__device__ __forceinline__ void block_add_u64(unsigned long long c, unsigned long long *dst)
{
    __shared__ unsigned long long part[4];
    c = wsum_u64(c);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = (part[0] + part[1]) + (part[2] + part[3]);
        if (t) atomicAdd(dst, t);
    }
}

// keys of flagged distances (or of |d - med| when center != null), ~0 for the rest; counts the flagged
__global__ __launch_bounds__(256) void k_reject_keys(const double *__restrict__ dist, const uint8_t *__restrict__ flag, long Q,
                                                     const double *__restrict__ center, unsigned long long *__restrict__ keys,
                                                     unsigned long long *__restrict__ count)
{
    unsigned long long c = 0;
    const double ctr = center ? center[0] : 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < Q; i += (long)gridDim.x * 256) {
        const bool f = flag[i] != 0;
        const double v = center ? fabs(dist[i] - ctr) : dist[i];
        keys[i] = f ? okey(v) : ~0ull;
        c += f ? 1 : 0;
    }
    if (count) block_add_u64(c, count);
}

__global__ __launch_bounds__(256) void k_reject_keep(const double *__restrict__ dist, const uint8_t *__restrict__ flag, long Q,
                                                     const double *__restrict__ med_mad, uint8_t *__restrict__ keep,
                                                     unsigned long long *__restrict__ kept)
{
    unsigned long long c = 0;
    const double med = med_mad[0], bound = 3 * med_mad[1];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < Q; i += (long)gridDim.x * 256) {
        const uint8_t k = (flag[i] && fabs(dist[i] - med) <= bound) ? 1 : 0;
        keep[i] = k; c += k;
    }
    block_add_u64(c, kept);
}

__global__ void k_reject_finish(const unsigned long long *__restrict__ counts /*[0]=m,[1]=kept*/,
                                const double *__restrict__ med_mad, double *__restrict__ out4)
{
    out4[0] = (double)counts[0]; out4[1] = med_mad[0]; out4[2] = med_mad[1]; out4[3] = (double)counts[1];
}

// ---- exact order statistics over many workgroups: 11-bit-digit radix selection ------------------------
// State (device): st[0] = key prefix selected so far, st[1] = rank inside it, st[2] = #keys <= the selected
// key, st[3] = smallest key above it; hist = 2048 global bins; one ticket.  Every pass is one launch: each
// block histograms the keys that still match the prefix in LDS (wave-aggregated: distances share their
// sign/exponent bits, so whole waves hit one bin in the early passes), adds its non-empty bins to the
// global histogram, and the LAST block to arrive picks the bin holding the rank, extends the prefix and
// clears the histogram for the next launch.  Six passes (5 x 11 + 9 bits) read the keys six times --
// against two full 64-bit device sorts before (0.75 ms at Q = 1 M).
constexpr int RSEL_BINS = 2048;
struct RselState { unsigned long long st[8]; unsigned hist[RSEL_BINS]; unsigned ticket; };

__device__ __forceinline__ bool last_block_arrives(unsigned *ticket, int *is_last_lds)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *is_last_lds = (t == gridDim.x - 1) ? 1 : 0;
        if (*is_last_lds) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return *is_last_lds != 0;
}

__global__ __launch_bounds__(256) void k_rsel_pass(const unsigned long long *__restrict__ keys, long Q, int pass,
                                                   RselState *__restrict__ S, const unsigned long long *__restrict__ count)
{
    __shared__ unsigned hist[RSEL_BINS];
    __shared__ unsigned scan[4];               // wave totals of the last block's prefix scan
    __shared__ int is_last;
    const int tid = threadIdx.x, lane = tid & 63;
    const int shift = pass < 5 ? 53 - 11 * pass : 0, bits = pass < 5 ? 11 : 9;
    const unsigned mask = (1u << bits) - 1u;
    const unsigned long long prefix = S->st[0];
    for (int i = tid; i < RSEL_BINS; i += 256) hist[i] = 0;
    __syncthreads();
    const long stride = (long)gridDim.x * 256;
    for (long base = (long)blockIdx.x * 256; base < Q; base += stride) {        // wave-uniform trip count
        const long i = base + tid;
        const unsigned long long k = i < Q ? keys[i] : ~0ull;
        bool act = k != ~0ull && (pass == 0 || (k >> (shift + bits)) == (prefix >> (shift + bits)));
        const unsigned bin = (unsigned)(k >> shift) & mask;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
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
    __syncthreads();
    for (int i = tid; i < RSEL_BINS; i += 256) if (hist[i]) atomicAdd(&S->hist[i], hist[i]);
    if (!last_block_arrives(&S->ticket, &is_last)) return;
    // thread t owns bins 8t..8t+7 of the complete histogram
    unsigned h[8], mine = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { h[j] = S->hist[8 * tid + j]; mine += h[j]; S->hist[8 * tid + j] = 0; }
    // exclusive prefix of the 256 per-thread counts: DPP scan inside each wave + the three wave totals before it
    const unsigned incl = wscan_u32(mine);
    if (lane == 63) scan[tid >> 6] = incl;
    if (tid == 0) S->ticket = 0;
    const long m = (long)count[0];
    const unsigned long long rank = pass == 0 ? (unsigned long long)((m - 1) / 2) : S->st[1];   // (read by all BEFORE the
    __syncthreads();                                                                            //  owner of the bin rewrites it)
    unsigned before = 0;
    for (int w = 0; w < (tid >> 6); ++w) before += scan[w];
    const unsigned excl = before + incl - mine;
    unsigned long long acc = excl;
    if (m > 0 && rank >= acc && rank < acc + mine) {
        int j = 0;
        while (rank >= acc + h[j]) { acc += h[j]; ++j; }
        S->st[0] = prefix | ((unsigned long long)(8 * tid + j) << shift);
        S->st[1] = rank - acc;
        if (pass == 5) { S->st[2] = 0; S->st[3] = ~0ull; }
    }
}

// second middle value + their mean: dst[0] = np.median of the keys' values (NaN when there are none)
__global__ __launch_bounds__(256) void k_rsel_finish(const unsigned long long *__restrict__ keys, long Q,
                                                     RselState *__restrict__ S, const unsigned long long *__restrict__ count,
                                                     double *__restrict__ dst)
{
    __shared__ int is_last;
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned long long ka = S->st[0];
    unsigned long long le = 0, nxt = ~0ull;
    for (long i = (long)blockIdx.x * 256 + tid; i < Q; i += (long)gridDim.x * 256) {
        const unsigned long long k = keys[i];
        if (k <= ka) le += 1; else nxt = k < nxt ? k : nxt;
    }
    le = wsum_u64(le);
    { unsigned long long o;
      o = lane_xor64<32>(nxt); nxt = o < nxt ? o : nxt;  o = lane_xor64<16>(nxt); nxt = o < nxt ? o : nxt;
      o = lane_xor64<8>(nxt);  nxt = o < nxt ? o : nxt;  o = lane_xor64<4>(nxt);  nxt = o < nxt ? o : nxt;
      o = lane_xor64<2>(nxt);  nxt = o < nxt ? o : nxt;  o = lane_xor64<1>(nxt);  nxt = o < nxt ? o : nxt; }
    __shared__ unsigned long long ple[4], pnx[4];
    if (lane == 0) { ple[tid >> 6] = le; pnx[tid >> 6] = nxt; }
    __syncthreads();
    if (tid == 0) {
        const unsigned long long tl = (ple[0] + ple[1]) + (ple[2] + ple[3]);
        unsigned long long tn = pnx[0];
        for (int w = 1; w < 4; ++w) tn = pnx[w] < tn ? pnx[w] : tn;
        if (tl) atomicAdd(&S->st[2], tl);
        if (tn != ~0ull) atomicMin(&S->st[3], tn);
    }
    if (!last_block_arrives(&S->ticket, &is_last)) return;
    if (tid == 0) {
        const long m = (long)count[0];
        const long r = (m - 1) / 2;
        const unsigned long long kb = ((m & 1) || (long)S->st[2] >= r + 2) ? ka : S->st[3];
        dst[0] = m > 0 ? (oval64(ka) + oval64(kb)) / 2.0 : __builtin_nan("");
        S->st[0] = 0; S->st[1] = 0; S->ticket = 0;
    }
}

size_t reject_select_scratch_bytes() { return sizeof(RselState); }

// scratch: keys (Q u64), state (reject_select_scratch_bytes), small (4 u64/doubles: m, kept, med, mad)
hipError_t reject_by_select(hipStream_t s, const double *dist, const uint8_t *flag, long Q, uint8_t *keep, double *out4,
                            unsigned long long *keys, void *state, unsigned long long *small)
{
    unsigned long long *counts = small;            // [0] m, [1] kept
    double *med_mad = (double *)(small + 2);       // [0] median, [1] mad
    RselState *S = (RselState *)state;
    hipError_t e = hipMemsetAsync(small, 0, 4 * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(state, 0, sizeof(RselState), s);
    if (e != hipSuccess) return e;
    const unsigned g = (unsigned)std::min<long>(2048, (Q + 255) / 256);
    const unsigned gs = (unsigned)std::min<long>(512, (Q + 1023) / 1024);
    for (int stat = 0; stat < 2; ++stat) {
        hipLaunchKernelGGL(k_reject_keys, dim3(g), dim3(256), 0, s, dist, flag, Q, stat ? (const double *)med_mad : nullptr, keys,
                           stat ? (unsigned long long *)nullptr : counts);
        for (int pass = 0; pass < 6; ++pass)
            hipLaunchKernelGGL(k_rsel_pass, dim3(gs), dim3(256), 0, s, (const unsigned long long *)keys, Q, pass, S,
                               (const unsigned long long *)counts);
        hipLaunchKernelGGL(k_rsel_finish, dim3(gs), dim3(256), 0, s, (const unsigned long long *)keys, Q, S,
                           (const unsigned long long *)counts, med_mad + stat);
    }
    hipLaunchKernelGGL(k_reject_keep, dim3(g), dim3(256), 0, s, dist, flag, Q, (const double *)med_mad, keep, counts + 1);
    hipLaunchKernelGGL(k_reject_finish, dim3(1), dim3(1), 0, s, counts, (const double *)med_mad, out4);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------

void launch_bbox(hipStream_t s, const double *x, const double *y, const double *z, long n, unsigned long long *out6)
{
    long g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_bbox, dim3((unsigned)g), dim3(256), 0, s, x, y, z, n, out6);
}

void launch_cell_ids(hipStream_t s, const double *x, const double *y, const double *z, long n, const GridGeom &G,
                     uint32_t *keys, uint32_t *vals, uint32_t *counts)
{
    hipLaunchKernelGGL(k_cell_ids, dim3(cdiv(n, 256)), dim3(256), 0, s, x, y, z, n, G, keys, vals, counts);
}

void launch_count_nonempty(hipStream_t s, const uint32_t *counts, long ncells, unsigned long long *out)
{
    long g = (ncells + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_count_nonempty, dim3((unsigned)g), dim3(256), 0, s, counts, ncells, out);
}

size_t grid_sort_temp_bytes(long n, int bits)
{
    size_t t = 0;
    uint32_t *p = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t, p, p, p, p, (int)n, 0, bits, (hipStream_t)0);
    return t;
}
size_t grid_scan_temp_bytes(long ncells)
{
    size_t t = 0;
    uint32_t *p = nullptr;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t, p, p, (int)ncells, (hipStream_t)0);
    return t;
}
hipError_t grid_sort(hipStream_t s, void *tmp, size_t tmp_bytes, const uint32_t *k_in, uint32_t *k_out,
                     const uint32_t *v_in, uint32_t *v_out, long n, int bits)
{
    return hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, k_in, k_out, v_in, v_out, (int)n, 0, bits, s);
}
hipError_t grid_scan(hipStream_t s, void *tmp, size_t tmp_bytes, const uint32_t *in, uint32_t *out, long ncells)
{
    return hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, in, out, (int)ncells, s);
}

void launch_gather_sorted(hipStream_t s, const double *x, const double *y, const double *z, const uint32_t *sidx, long n,
                          double *sx, double *sy, double *sz)
{
    hipLaunchKernelGGL(k_gather_sorted, dim3(cdiv(n, 256)), dim3(256), 0, s, x, y, z, sidx, n, sx, sy, sz);
}

void launch_grid_nn(hipStream_t s, const double *qx, const double *qy, const double *qz, long Q, const double *prev_p2,
                    const GridGeom &G, const uint32_t *cell_start, const double *sx, const double *sy, const double *sz,
                    const uint32_t *sidx, const Xf *H, const Xf *Hinv, double rmax, double max_d2, int64_t idx_base,
                    double *d2_out, int64_t *idx_out, double *p2_out)
{
    const dim3 grid(cdiv(Q, 4)), block(256);
    Xf id = {};
    if (H)
        hipLaunchKernelGGL((k_grid_nn<true, false>), grid, block, 0, s, qx, qy, qz, Q, prev_p2, G, cell_start, sx, sy, sz, sidx, *H,
                           *Hinv, rmax, max_d2, idx_base, d2_out, idx_out, p2_out, (const IcpDev *)nullptr);
    else
        hipLaunchKernelGGL((k_grid_nn<false, false>), grid, block, 0, s, qx, qy, qz, Q, prev_p2, G, cell_start, sx, sy, sz, sidx, id,
                           id, rmax, max_d2, idx_base, d2_out, idx_out, p2_out, (const IcpDev *)nullptr);
}

// the match of a chained iteration: transform taken from the loop state on the device
void launch_grid_nn_chained(hipStream_t s, const double *qx, const double *qy, const double *qz, long Q, const double *prev_p2,
                            const GridGeom &G, const uint32_t *cell_start, const double *sx, const double *sy, const double *sz,
                            const uint32_t *sidx, const IcpDev *st, double rmax, int64_t idx_base, double *d2_out,
                            int64_t *idx_out, double *p2_out)
{
    Xf id = {};
    hipLaunchKernelGGL((k_grid_nn<true, true>), dim3(cdiv(Q, 4)), dim3(256), 0, s, qx, qy, qz, Q, prev_p2, G, cell_start, sx, sy, sz,
                       sidx, id, id, rmax, (double)__builtin_inf(), idx_base, d2_out, idx_out, p2_out, st);
}

void launch_grid_knn(hipStream_t s, const double *qx, const double *qy, const double *qz, long Q, int k, const GridGeom &G,
                     const uint32_t *cell_start, const double *sx, const double *sy, const double *sz, const uint32_t *sidx,
                     double rmax, int64_t idx_base, double *d2_out, int64_t *idx_out)
{
    hipLaunchKernelGGL(k_grid_knn, dim3(cdiv(Q, 4)), dim3(256), 0, s, qx, qy, qz, Q, k, G, cell_start, sx, sy, sz, sidx,
                       rmax, idx_base, d2_out, idx_out);
}

}  // namespace sicp
