// sicp_grid_dev.h -- device helpers shared by the grid kernels (sicp_grid.hip: the exact searches, the k-NN sweeps; sicp_gridf.hip: the
// filtered many-queries search, the cells' tight boxes).  Device code only.
#ifndef SICP_GRID_DEV_H
#define SICP_GRID_DEV_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sicp_internal.h"
#include "sicp_lanes.h"

namespace sicp {

__device__ __forceinline__ int cell_coord(double v, double mn, double inv_h, int dim)
{
    int c = (int)floor((v - mn) * inv_h);
    return c < 0 ? 0 : (c >= dim ? dim - 1 : c);
}
__device__ __forceinline__ uint32_t cell_of(const GridGeom &G, double x, double y, double z)
{
    const int cx = cell_coord(x, G.mn[0], G.inv_h, G.dim[0]);
    const int cy = cell_coord(y, G.mn[1], G.inv_h, G.dim[1]);
    const int cz = cell_coord(z, G.mn[2], G.inv_h, G.dim[2]);
    return ((uint32_t)cz * G.dim[1] + cy) * G.dim[0] + cx;
}

// contract (T): rows 0..2 of H applied to a point, this operation order everywhere
__device__ __forceinline__ void xf(const Xf &H, double x, double y, double z, double &ox, double &oy, double &oz)
{
    double t;
    t = H.m[0] * x;  t = fma(H.m[1], y, t);  t = fma(H.m[2], z, t);   ox = t + H.m[3];
    t = H.m[4] * x;  t = fma(H.m[5], y, t);  t = fma(H.m[6], z, t);   oy = t + H.m[7];
    t = H.m[8] * x;  t = fma(H.m[9], y, t);  t = fma(H.m[10], z, t);  oz = t + H.m[11];
}

// row rr of a ny-wide block of grid rows -> (rr % ny, rr / ny) without the 64-bit integer division the plain expressions compile
// to (~150 instructions per lane and pass: a tenth of the four-queries-per-wave search's issue budget).  Exact for rr < 2^22
// (float quotient within one of the truth, then corrected); balls with more rows take the division.
__device__ __forceinline__ void row_split(long rr, int ny, float inv_ny, bool small, int &oy, int &oz)
{
    if (small) {
        unsigned q = (unsigned)(((float)(unsigned)rr + 0.5f) * inv_ny);
        int r = (int)(unsigned)rr - (int)(q * (unsigned)ny);
        if (r < 0) { q -= 1u; r += ny; } else if (r >= ny) { q += 1u; r -= ny; }
        oy = r; oz = (int)q;
    } else {
        oy = (int)(rr % ny); oz = (int)(rr / ny);
    }
}

// What k_postmatch computes (sicp_kernels.hip), by the lane that holds the winner: signed point-to-plane distance of the matched
// point under H, contract (P), and the planarity verdict of both clouds -- the same expressions, so the same bits.
__device__ __forceinline__ void post_match(const PostMatch &post, const Xf &H, long q, int64_t m, double px, double py, double pz,
                                           double ax, double ay, double az, float nx, float ny, float nz, float pl)
{
    double X, Y, Z;
    xf(H, px, py, pz, X, Y, Z);
    const double a = (X - ax) * (double)nx, b = (Y - ay) * (double)ny, c = (Z - az) * (double)nz;
    post.dist[q] = (a + b) + c;
    bool f = m >= 0 && pl >= post.min_planarity;
    if (f && post.pl2) f = m < post.pl2_n && post.pl2[m] >= post.min_planarity;          // corrpts.py:158-163 (NaN fails)
    post.flag[q] = f ? 1 : 0;
}

// ---- tight boxes of the cells ("per-tile AABB culling") -------------------------------------------------------------
// One 64-bit word per cell of the dense table: the bounding box of the cell's points, each face quantised OUTWARDS to 1/256 of the
// cell size relative to the cell's own cube (bytes 0..2 low faces x y z, bytes 3..5 high faces), and the number of points in the
// cell (bits 48..63, saturating: 0xffff = "that many or more", such a cell is never skipped).  A far search -- the first iterations
// of a run, when the estimate is still metres off -- must open every cell whose CUBE comes within the best distance found so far; a
// cube of side h around eight points of a surface is mostly empty, and the lower bound from the points' own box prunes about half
// of what the cubes let through (measured on the bench terrain, iteration 0: 98 -> 44 candidates per query at 8 points per cell).
constexpr unsigned BOX_COUNT_SAT = 0xffffu;
__device__ __forceinline__ unsigned box_count(unsigned long long w) { return (unsigned)(w >> 48); }

// squared distance from (qx, qy, qz) to the box `w` of cell (cx, cy, cz): a lower bound of the distance to every point of the cell
// (faces moved outwards by etol, the tolerance every cell test of the grid searches carries)
__device__ __forceinline__ double box_lb2(unsigned long long w, const GridGeom &G, int cx, int cy, int cz, double qx, double qy,
                                          double qz, double etol)
{
    const double s = G.h * (1.0 / 256.0);
    const double ox = G.mn[0] + (double)cx * G.h, oy = G.mn[1] + (double)cy * G.h, oz = G.mn[2] + (double)cz * G.h;
    const unsigned lo = (unsigned)w, hi = (unsigned)(w >> 24);
    const double lx = ox + (double)(lo & 0xffu) * s - etol,          hx = ox + (double)((hi & 0xffu) + 1u) * s + etol;
    const double ly = oy + (double)((lo >> 8) & 0xffu) * s - etol,   hy = oy + (double)(((hi >> 8) & 0xffu) + 1u) * s + etol;
    const double lz = oz + (double)((lo >> 16) & 0xffu) * s - etol,  hz = oz + (double)(((hi >> 16) & 0xffu) + 1u) * s + etol;
    const double dx = fmax(fmax(lx - qx, qx - hx), 0.0), dy = fmax(fmax(ly - qy, qy - hy), 0.0), dz = fmax(fmax(lz - qz, qz - hz), 0.0);
    return fma(dz, dz, fma(dy, dy, dx * dx));
}

// Trims the cells [xl, xh] of grid row `row` (= ((cz * dim1 + cy) * dim0); record range [b, b + len)) from both ends: a cell whose
// points' box lies farther than sqrt(cull2) from the query cannot hold the answer (nor a tie: cull2 carries the caller's
// margins), and dropping it from an END keeps the row's records one contiguous range.  Up to BOX_TRIM cells per end are looked
// at -- their words are loaded together, one round trip -- which is the whole row for all but the widest balls.
constexpr int BOX_TRIM = 4;
__device__ __forceinline__ void box_trim_row(const unsigned long long *__restrict__ cell_box, const GridGeom &G, long row, int cy, int cz,
                                             int xl, int xh, double qx, double qy, double qz, double cull2, double etol,
                                             uint32_t &b, uint32_t &len)
{
    const int n = xh - xl + 1;
    if (n <= 0 || len == 0u) return;
    // left end: cells xl + i; right end: cells xh - i that the left end does not look at (rows wider than BOX_TRIM cells)
    unsigned long long wl[BOX_TRIM], wr[BOX_TRIM];
#pragma unroll
    for (int i = 0; i < BOX_TRIM; ++i) {
        wl[i] = i < n ? cell_box[row + xl + i] : 0ull;
        wr[i] = (n > BOX_TRIM && xh - i >= xl + BOX_TRIM) ? cell_box[row + xh - i] : 0ull;
    }
    bool gone_l[BOX_TRIM], gone_r[BOX_TRIM];
    unsigned cnt_l[BOX_TRIM], cnt_r[BOX_TRIM];
#pragma unroll
    for (int i = 0; i < BOX_TRIM; ++i) {
        // an empty cell goes for free; a cell whose count saturated never goes (its share of the row's records is unknown)
        cnt_l[i] = box_count(wl[i]);
        gone_l[i] = i < n && (cnt_l[i] == 0u || (cnt_l[i] != BOX_COUNT_SAT && box_lb2(wl[i], G, xl + i, cy, cz, qx, qy, qz, etol) > cull2));
        cnt_r[i] = box_count(wr[i]);
        gone_r[i] = n > BOX_TRIM && xh - i >= xl + BOX_TRIM &&
                    (cnt_r[i] == 0u || (cnt_r[i] != BOX_COUNT_SAT && box_lb2(wr[i], G, xh - i, cy, cz, qx, qy, qz, etol) > cull2));
    }
    unsigned drop = 0, dropr = 0;
    int lead = 0;
    bool open = true;
#pragma unroll
    for (int i = 0; i < BOX_TRIM; ++i) {
        open = open && gone_l[i];
        if (open) { drop += cnt_l[i]; ++lead; }
    }
    if (n <= BOX_TRIM) {
        if (lead >= n) { len = 0u; return; }                // every cell of the row was looked at and none stays
        open = true;                                         // the same cells from the other end (one of them stays: the loop stops there)
#pragma unroll
        for (int i = BOX_TRIM - 1; i >= 0; --i) {
            if (i >= n) continue;
            open = open && gone_l[i];
            if (open) dropr += cnt_l[i];
        }
    } else {
        open = true;
#pragma unroll
        for (int i = 0; i < BOX_TRIM; ++i) {
            open = open && gone_r[i];
            if (open) dropr += cnt_r[i];
        }
    }
    // (counts are exact below the saturation value and a saturated cell is never dropped: drop + dropr < len whenever a cell stays)
    if (drop + dropr >= len) { len = 0u; return; }
    b += drop; len -= drop + dropr;
}

}  // namespace sicp

#endif
