// sicp_gridf.hip -- the many-queries 1-NN search with a float32 FILTER in the cloud's own frame, and the grid's two lazily built
// companions it reads: the cloud as 16-byte float records and the cells' tight boxes (sicp_grid_dev.h).
//
// Replaces, like k_grid_nn16 before it, CorrPts.match of the reference (corrpts.py:124-137: cKDTree rebuild + query per iteration)
// for large query sets; the answers stay the brute-force scan's, bit for bit.
//
// Why.  k_grid_nn16 (sicp_grid.hip) evaluates every candidate with the exact contract: 32-byte record, H applied in float64 (12
// operations), distance (6), lexicographic compare -- ~34 vector instructions and 32 bytes per candidate, 120 registers, 4 waves per
// SIMD, and the counters say 77 % of its wave-cycles wait for memory (profiles/r4).  But the query has already been pulled back into
// the cloud's frame (to find its cells), and in THAT frame a candidate's distance needs no transform.  So here:
//   * candidates are 16-byte records: (x, y, z) as float32 relative to the centre of the cloud's box, and the distance to the
//     pulled-back query is formed in float32 -- 3 subtractions, 1 multiplication, 2 fused multiply-adds;
//   * a lane keeps the smallest and second-smallest value it met and where the smallest was; the group's winner is the exact
//     nearest neighbour IF no other value of the group lies within T of the smallest, where T/2 bounds |float32 value - contract
//     distance| for every candidate no farther than the pass's radius (derivation at filter_margin below).  That is decided by
//     counting, not assumed;
//   * the winner -- ONE candidate per pass and query -- is then evaluated with the exact contract from its 32-byte record; that value
//     decides termination, bounds the next pass and is what the caller gets;
//   * a query whose smallest value is not alone within T (0.3 % of the queries of a 1 km cloud; all of them on data with
//     coincident points, or with coordinates that float32 cannot hold) is not answered here: it goes on a list that k_grid_nn, the
//     exact one-wave-per-query kernel, works off in the launch behind this one.  Same arithmetic as ever for those, so the same bits.
// The point-to-plane distances and planarity verdicts of an ICP iteration are k_postmatch's here (a launch of its own: fused into a
// search at its register limit they cost the search more than the launch, measured in round 4), as are the exchange's records.
// (The cells' tight boxes, sicp_grid_dev.h, are NOT used here: with candidates this cheap -- 16 bytes, 13 instructions -- trimming rows
// by them cut 14-30 % of the candidates of a cold search and made it slower, profiles/r5.  They remain an option of the exact kernel.)
#include <hip/hip_runtime.h>
#include <cmath>
#include <stdint.h>

#include "sicp_internal.h"
#include "sicp_lanes.h"
#include "sicp_grid_dev.h"

namespace sicp {

static inline unsigned cdivf(long a, long b) { return (unsigned)((a + b - 1) / b); }

// ---- the grid's companions ---------------------------------------------------------------------------------------------------------
// records -> (x - c0x, y - c0y, z - c0z) as float32 + the original row's low 32 bits (unused by the filter: it names candidates by
// their position in cell order, which is also where the 32-byte record of the winner is)
__global__ __launch_bounds__(256) void k_recf(const double4 *__restrict__ rec, long n, double c0x, double c0y, double c0z,
                                              float4 *__restrict__ recf)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double4 P = rec[i];
    recf[i] = make_float4((float)(P.x - c0x), (float)(P.y - c0y), (float)(P.z - c0z),
                          __uint_as_float((unsigned)(unsigned long long)__double_as_longlong(P.w)));
}

// one thread per cell of the dense table: the box of the cell's points, quantised outwards (layout: sicp_grid_dev.h)
__global__ __launch_bounds__(256) void k_cell_boxes(const uint32_t *__restrict__ cell_start, const double4 *__restrict__ rec, long ncells,
                                                    GridGeom G, unsigned long long *__restrict__ cell_box)
{
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= ncells) return;
    const uint32_t s0 = cell_start[c], s1 = cell_start[c + 1];
    if (s1 <= s0) { cell_box[c] = 0ull; return; }
    double lo[3] = {__builtin_inf(), __builtin_inf(), __builtin_inf()}, hi[3] = {-__builtin_inf(), -__builtin_inf(), -__builtin_inf()};
    for (uint32_t i = s0; i < s1; ++i) {
        const double4 P = rec[i];
        lo[0] = fmin(lo[0], P.x); hi[0] = fmax(hi[0], P.x);
        lo[1] = fmin(lo[1], P.y); hi[1] = fmax(hi[1], P.y);
        lo[2] = fmin(lo[2], P.z); hi[2] = fmax(hi[2], P.z);
    }
    const long rowlen = G.dim[0], plane = (long)G.dim[0] * G.dim[1];
    const int cz = (int)(c / plane), cy = (int)((c - (long)cz * plane) / rowlen), cx = (int)(c - (long)cz * plane - (long)cy * rowlen);
    const int cc[3] = {cx, cy, cz};
    unsigned long long w = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double o = G.mn[a] + (double)cc[a] * G.h;
        // floor() of both faces: the decoded box is [o + ql s - etol, o + (qh + 1) s + etol], s = h / 256 (box_lb2)
        double ql = floor((lo[a] - o) * G.inv_h * 256.0), qh = floor((hi[a] - o) * G.inv_h * 256.0);
        ql = ql < 0.0 ? 0.0 : (ql > 255.0 ? 255.0 : ql);
        qh = qh < 0.0 ? 0.0 : (qh > 255.0 ? 255.0 : qh);
        w |= (unsigned long long)(unsigned)ql << (8 * a);
        w |= (unsigned long long)(unsigned)qh << (24 + 8 * a);
    }
    const uint32_t cnt = s1 - s0;
    w |= (unsigned long long)(cnt < BOX_COUNT_SAT ? cnt : BOX_COUNT_SAT) << 48;
    cell_box[c] = w;
}

void launch_recf(hipStream_t s, const void *rec, long n, const double c0[3], void *recf)
{
    hipLaunchKernelGGL(k_recf, dim3(cdivf(n, 256)), dim3(256), 0, s, (const double4 *)rec, n, c0[0], c0[1], c0[2], (float4 *)recf);
}
void launch_cell_boxes(hipStream_t s, const uint32_t *cell_start, const void *rec, long ncells, const GridGeom &G, unsigned long long *cell_box)
{
    hipLaunchKernelGGL(k_cell_boxes, dim3(cdivf(ncells, 256)), dim3(256), 0, s, cell_start, (const double4 *)rec, ncells, G, cell_box);
}

// ---- the filter's margin ------------------------------------------------------------------------------------------------------------
// v = float32 value of a candidate p for a query whose pulled-back image is q':  v = fl32(dx^2 + dy^2 + dz^2), dx = fl32(pf.x - qf.x),
// pf = fl32(p - c0), qf = fl32(q' - c0).  Against the contract distance d2 = |fl(H p) - q|^2 of the same candidate:
//   * per coordinate |dx - (p.x - q'.x)| <= e_c := eps_p + eps_q + 2^-24 |dx|, eps_q = 2^-24 |q' - c0|, eps_p = 2^-24 |p - c0| <= 2^-24
//     min(the cloud's half extent, |q' - c0| + A) for a candidate no farther than A from the query (both with the float64
//     subtraction's rounding thrown in: 6.0e-8 instead of 5.96e-8);
//   * |sum dx^2 - |p - q'|^2| <= e_c (2 sqrt(3) d + 3 e_c)  (Cauchy-Schwarz on sum |dx|), the three float32 roundings of the sum add
//     at most 2^-22 d^2;
//   * |p - q'| and the contract distance agree to `slack` (1e-12 x scale; sicp_grid.hip, the same slack the exact kernels use for the
//     same purpose: H^-1 q is rounded, R^T R = I holds to 1e-16), so their squares to 2 d slack + slack^2.
// For a candidate no farther than A:  |v - d2| <= E(A) := e_c (3.5 A + 3 e_c) + 2.4e-7 A^2 + 2.02 A slack + slack^2,  e_c = eps + 6e-8 A.
// E grows with A.  Let w be the candidate with the smallest v and suppose every other value of the group exceeds v_w + 2 E(A), A >= the
// contract distance of w.  A candidate c with d2(c) <= d2(w) would be no farther than A either, so v_c <= d2(c) + E <= d2(w) + E <=
// v_w + 2 E: contradiction.  Hence d2(c) > d2(w) for every c != w: w is the exact nearest neighbour and no index tie-break is needed.
// A is the pass's radius: the pass only ends the search when the winner lies inside it.
__device__ __forceinline__ float filter_margin(double A, double eps, double slack)
{
    const double e_c = eps + 6.0e-8 * A;
    const double E = e_c * (3.5 * A + 3.0 * e_c) + 2.4e-7 * A * A + 2.02 * A * slack + slack * slack;
    const double T = 2.0 * E * 1.001;
    // (rounded up; never zero: a candidate that ties with the best must count as "within T")
    return T < 1.0e30 ? (float)T * 1.0001f + 1.0e-37f : __builtin_inff();
}

#ifndef SICP_NN16F_OCC
#define SICP_NN16F_OCC 5                    // waves per SIMD the NEAR flavour's register budget is set for (A/B builds: build.build_variant)
#endif
#ifndef SICP_NN16F_FAR_OCC
#define SICP_NN16F_FAR_OCC 4                // ... the FAR flavour's
#endif

struct FilterGeom {
    double c0[3];                           // origin of the float32 coordinates (centre of the cloud's box)
    double eps_p;                           // 6e-8 x the largest |coordinate - c0| of the cloud
};

// queries in SLOT order (the order waves take them in: cell order, one contiguous eighth per XCD): coordinates + the query's own
// index in ONE 32-byte record, and next to it the slot's last match (the bound of the next search) -- so that nothing the search
// starts from is behind an index indirection: order[slot] -> q -> qx[q], prev_p2[3 q] were two dependent round trips of a kernel
// that is bound by (queries in flight) x (round trips per query) x latency and by nothing else (profiles/r4: 77 % of wave-cycles waiting)
__global__ __launch_bounds__(256) void k_slot_queries(const double *__restrict__ qx, const double *__restrict__ qy, const double *__restrict__ qz,
                                                      const uint32_t *__restrict__ order /* nullable: slot = query */,
                                                      const double *__restrict__ prev_p2 /* nullable: (Q, 3) a cloud point per query */, long Q,
                                                      double4 *__restrict__ qrec, double4 *__restrict__ pslot)
{
    const long slot = (long)blockIdx.x * 256 + threadIdx.x;
    if (slot >= Q) return;
    const long q = order ? (long)order[slot] : slot;
    qrec[slot] = make_double4(qx[q], qy[q], qz[q], __longlong_as_double((long long)q));
    // (no match yet: x = inf says "no bound")
    pslot[slot] = prev_p2 ? make_double4(prev_p2[3 * q], prev_p2[3 * q + 1], prev_p2[3 * q + 2], 0.0) : make_double4(__builtin_inf(), 0.0, 0.0, 0.0);
}
void launch_slot_queries(hipStream_t s, const double *qx, const double *qy, const double *qz, const uint32_t *order, const double *prev_p2,
                         long Q, void *qrec, void *pslot)
{
    hipLaunchKernelGGL(k_slot_queries, dim3(cdivf(Q, 256)), dim3(256), 0, s, qx, qy, qz, order, prev_p2, Q, (double4 *)qrec, (double4 *)pslot);
}

// Behind a cloud-shard exchange: the slot's bound becomes the JOB-WIDE winner (the exchange left it in the by-query arrays).  The search
// itself wrote this rank's own winner there -- on a rank whose shard lies elsewhere that is a far-away point, and every later search
// of the slot would start from its radius (lean pass, full pass, the exact kernel: ADVICE r5); any cloud point is a valid bound, the
// nearest one anybody holds is the useful one.
__global__ __launch_bounds__(256) void k_slot_bounds(const double4 *__restrict__ qrec, const int64_t *__restrict__ idx,
                                                     const double *__restrict__ p2, long Q, double4 *__restrict__ pslot)
{
    const long slot = (long)blockIdx.x * 256 + threadIdx.x;
    if (slot >= Q) return;
    const long q = (long)__double_as_longlong(qrec[slot].w);
    pslot[slot] = idx[q] >= 0 ? make_double4(p2[3 * q], p2[3 * q + 1], p2[3 * q + 2], 0.0) : make_double4(__builtin_inf(), 0.0, 0.0, 0.0);
}
void launch_slot_bounds(hipStream_t s, const void *qrec, const int64_t *idx, const double *p2, long Q, void *pslot)
{
    hipLaunchKernelGGL(k_slot_bounds, dim3(cdivf(Q, 256)), dim3(256), 0, s, (const double4 *)qrec, idx, p2, Q, (double4 *)pslot);
}

// FAR = false: the flavour of a run's steady state -- a ball of a few rows, one batch of them, no hit-driven culling, no boxes; a
//   query that turns out to need more (more rows than the group has lanes, more than four of them non-empty) is left to the exact
//   kernel like a tie.  That is what the registers of five waves per SIMD pay for.
// FAR = true: everything (a run's first iterations, searches with a distance limit): row batches, nearest row first, the hit's ball.
template <int GS /* lanes per query: 16 (four queries per wave) or 8 (eight) */, bool FAR>
__global__ __launch_bounds__(256, FAR ? SICP_NN16F_FAR_OCC : SICP_NN16F_OCC) void k_grid_nn16f(
    const IcpDev *__restrict__ st, const double4 *__restrict__ qrec /* by slot: x, y, z, query index */,
    double4 *__restrict__ pslot /* by slot: in = a cloud point whose distance bounds the answer (x = inf: none); out = this search's match */,
    const uint32_t *__restrict__ cell_start, const float4 *__restrict__ recf, const double4 *__restrict__ rec,
    long Q, GridGeom G, FilterGeom F,
    Xf H, Xf Hinv, int has_H /* without a chain state: is there a transform at all */, double rmax, double max_d2, int64_t idx_base,
    double *__restrict__ d2_out /* nullable with idx_out, p2_out: a search whose only product is the bound it leaves in pslot */,
    int64_t *__restrict__ idx_out, double *__restrict__ p2_out,
    unsigned long long *__restrict__ work /* nullable: [0] candidates, [1] rows, [2] launches, [3] queries left to the exact kernel */,
    int flags, int xcd_order /* workgroups take the slots one contiguous eighth per XCD */,
    uint8_t *__restrict__ state /* nullable, by slot.  FAR = false: out -- 1 = this query needs the other flavour, 0 = dealt with;
                                   FAR = true: in -- only the slots marked 1 are searched */,
    uint32_t *__restrict__ redo_list, unsigned *__restrict__ redo_count)
{
    constexpr int GPW = 64 / GS;
    constexpr unsigned GMASK = GS == 16 ? 0xffffu : 0xffu;
    const int lane = threadIdx.x & 63, gl = lane & (GS - 1), gbase = lane & (64 - GS);
    long blk = blockIdx.x;
    if (xcd_order) blk = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);     // one contiguous eighth per XCD
    const long slot = (blk * 4 + (threadIdx.x >> 6)) * GPW + lane / GS;
    bool active = slot < Q;
    if (FAR && state) active = active && state[active ? slot : 0] == (uint8_t)1;
    if (!__any(active)) return;
    const bool tight = (flags & NN_TIGHT) != 0, approx = (flags & NN_APPROX) != 0;
    double cxq, cyq, czq, slack, r_lim;
    uint32_t q;
    {
        const double4 A = qrec[active ? slot : 0];
        const double4 B = pslot[active ? slot : 0];
        if (st) {
            H = st->H; Hinv = st->Hinv; has_H = 1;
            if (st->stop) return;
        }
        q = (uint32_t)(unsigned long long)__double_as_longlong(A.w);
        cxq = A.x; cyq = A.y; czq = A.z;
        if (has_H) xf(Hinv, A.x, A.y, A.z, cxq, cyq, czq);
        const double scale = rmax + (fabs(cxq) + fabs(cyq) + fabs(czq)) + 1.0;     // (1-norm: an upper bound of |q| is all the slack needs)
        slack = 1e-12 * scale;
        r_lim = (max_d2 < __builtin_inf()) ? sqrt(max_d2) * (1.0 + 1e-12) + slack : __builtin_inf();
        double X = B.x, Y = B.y, Z = B.z;
        if (has_H) { double u, v, w; xf(H, X, Y, Z, u, v, w); X = u; Y = v; Z = w; }
        const double dx = X - A.x, dy = Y - A.y, dz = Z - A.z;
        const double bnd = fma(dz, dz, fma(dy, dy, dx * dx));
        if (B.x < __builtin_inf() && bnd < __builtin_inf()) {
            const double rb = sqrt(bnd) * (1.0 + 1e-12) + slack;
            if (rb < r_lim) r_lim = rb;
        }
    }
    double r = 0.75 * G.h;
    if (r > r_lim || (tight && r_lim < __builtin_inf())) r = r_lim;
    // the query in the filter's frame; eps: what float32 loses on a coordinate of the cloud and on one of the query
    float fqx, fqy, fqz;
    double qinf;                              // largest |coordinate| of the query in the filter's frame
    bool defer;
    {
        const double rqx = cxq - F.c0[0], rqy = cyq - F.c0[1], rqz = czq - F.c0[2];
        fqx = (float)rqx; fqy = (float)rqy; fqz = (float)rqz;
        qinf = fmax(fabs(rqx), fmax(fabs(rqy), fabs(rqz)));
        // a query float32 cannot place (1e15 and beyond: squares would overflow) is left to the exact kernel at once
        defer = active && !(qinf < 1.0e15);
    }
    bool done = !active || defer, last = false;
    if (approx) defer = false;                // (a search for a bound answers nobody: such a query simply gets no bound)
    bool unplaced = defer;                    // left to the exact kernel WITHOUT an approximate winner: it gets no bound
    bool far_defer = false;
    unsigned n_cand = 0, n_rows = 0;
    for (int pass = 0; pass < 4096 && __any(!done); ++pass) {
        int lo[3], hi[3];
        bool all = true;
        {
            const double c3[3] = {cxq, cyq, czq};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double fl = floor((c3[a] - r - G.mn[a]) * G.inv_h - 1e-6);
                const double fh = floor((c3[a] + r - G.mn[a]) * G.inv_h + 1e-6);
                lo[a] = fl < 0.0 ? 0 : (fl > (double)(G.dim[a] - 1) ? G.dim[a] - 1 : (int)fl);
                hi[a] = fh < 0.0 ? 0 : (fh > (double)(G.dim[a] - 1) ? G.dim[a] - 1 : (int)fh);
                all = all && (fl <= 0.0) && (fh >= (double)(G.dim[a] - 1));
            }
        }
        // The winner only ends the search when it lies inside the pass's ball, i.e. no farther than r: the margin for radius r
        // decides whether it is alone.  (`all`: the pass is final whatever it finds -- the farthest point of the cloud is then the
        // bound: |q'| + rmax.)
        // eps: what float32 loses on a coordinate of the query (6e-8 qinf) and on one of a candidate no farther than A from it
        // (6e-8 (qinf + A): such a candidate's coordinates are within A of the query's) -- never more than the cloud-wide figure
        const double A = all ? (fabs(cxq) + fabs(cyq) + fabs(czq)) + rmax : r;
        const double eps = 6.0e-8 * qinf + fmin(F.eps_p, 6.0e-8 * (qinf + A));
        const float T = filter_margin(A, eps, slack);
        float v1 = __builtin_inff(), v2 = __builtin_inff();     // smallest and second-smallest value this lane met in this pass
        uint32_t bpos = 0;                                       // where the smallest is (position in cell order)
        const int ny = hi[1] - lo[1] + 1, nz = hi[2] - lo[2] + 1;
        const long nrows = done ? 0 : (long)ny * nz;
        // candidates of up to two rows per group, TWO records per lane and row in flight: lane gl of a group takes records gl and
        // gl + GS (+ 2 GS, ...) of each of its rows -- a ball of the steady state (two or three rows of one to three cells) is one step
        auto scan_rows = [&](const uint32_t (&rbv)[2], const uint32_t (&rlv)[2]) {
            const uint32_t longest = rlv[0] > rlv[1] ? rlv[0] : rlv[1];
            for (uint32_t o = 0; __any(o < longest); o += 2 * GS) {
                float4 P[4];
                uint32_t at[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t i = o + (uint32_t)gl + (u & 2 ? (uint32_t)GS : 0u);
                    ok[u] = i < rlv[u & 1];
                    at[u] = ok[u] ? rbv[u & 1] + i : 0u;
                    P[u] = recf[at[u]];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float dx = P[u].x - fqx, dy = P[u].y - fqy, dz = P[u].z - fqz;
                    float v = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                    v = ok[u] ? v : __builtin_inff();
                    const bool first = v < v1;
                    v2 = first ? v1 : (v < v2 ? v : v2);
                    bpos = first ? at[u] : bpos;
                    v1 = first ? v : v1;
                }
                if (work) { for (int u = 0; u < 4; ++u) n_cand += ok[u] ? 1u : 0u; }
            }
        };
        // the ball, not its bounding cube (see k_grid_nn); once a hit bounds the answer the ball is the hit's, not the pass's
        const double r2 = r * r, etol = 1e-6 * G.h;
        double cull2 = __builtin_inf();
        const float inv_ny = 1.0f / (float)ny;
        const bool few_rows = nrows < (1L << 22);
        // row rr of the pass's block -> its cells [xl, xh] within the ball (cull2: the hit's ball), its record range
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
                const double rem = (FAR ? fmin(r2, cull2) : r2) - lb2;
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
        bool far_query = false;                  // (FAR = false) this query needs the other flavour
        // a ball of thousands of rows (a query far from everything, on a grid binned for a dense core): the exact kernel's business --
        // a wave per query takes 64 rows per batch, and such clouds give it a coarse grid for exactly these passes
        const bool huge = FAR && !done && !approx && nrows > 2048;
        if constexpr (!FAR) {
            uint32_t b = 0, len = 0;
            far_query = !done && nrows > (long)GS;
            if (!far_query && gl < (int)nrows) {
                double lb2;
                row_range((long)gl, b, len, lb2);
            }
            unsigned todo = (unsigned)(__ballot(len > 0) >> gbase) & GMASK;
            if (work && len > 0) n_rows += 1u;
            if (__popc(todo) > 4) { far_query = true; todo = 0u; }
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
        } else {
            // TWO rows per lane and batch: their offsets are in flight together -- a wide ball's cost is the number of dependent
            // batches (offsets, then records, per batch), not its candidates (measured: 10 batches of 8 rows made the cold search of
            // 1 M queries 1.7 ms where its candidates account for 1.2)
            const long nrows_s = huge ? 0 : nrows;            // (its lanes simply have no rows: the wave's other groups go on)
            for (long rb = 0; __any(rb < nrows_s); rb += 2 * GS) {
                uint32_t b[2] = {0u, 0u}, len[2] = {0u, 0u};
                double lb2[2] = {__builtin_inf(), __builtin_inf()};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const long rr = rb + (long)u * GS + gl;
                    if (rr < nrows_s) row_range(rr, b[u], len[u], lb2[u]);
                }
                // bit j of the group's mask: row of lane j % GS, slot j / GS
                unsigned todo = ((unsigned)(__ballot(len[0] > 0) >> gbase) & GMASK) | (((unsigned)(__ballot(len[1] > 0) >> gbase) & GMASK) << GS);
                if (work) n_rows += (len[0] > 0 ? 1u : 0u) + (len[1] > 0 ? 1u : 0u);
                // row j of the group -> its record range (two crossbar moves per slot: the slot is uniform per group, not per wave)
                auto fetch = [&](int j, uint32_t &vb, uint32_t &vl) {
                    const int src = gbase + (j & (GS - 1));
                    const uint32_t b0 = (uint32_t)__shfl((int)b[0], src), b1 = (uint32_t)__shfl((int)b[1], src);
                    const uint32_t l0 = (uint32_t)__shfl((int)len[0], src), l1 = (uint32_t)__shfl((int)len[1], src);
                    vb = j >= GS ? b1 : b0; vl = j >= GS ? l1 : l0;
                };
                const bool many = __popc(todo) > 4;
                if (__any(many)) {
                    // groups with many rows (a far search): nearest row first, then drop the rows its hit rules out and shrink the
                    // others' x ranges to the hit's ball
                    const unsigned long long k0 = len[0] > 0 ? (unsigned long long)__double_as_longlong(lb2[0]) : ~0ull;
                    const unsigned long long k1 = len[1] > 0 ? (unsigned long long)__double_as_longlong(lb2[1]) : ~0ull;
                    unsigned long long mk = k0 < k1 ? k0 : k1;
                    { unsigned long long o;
                      if constexpr (GS == 16) { o = lane_xor64<8>(mk);  mk = o < mk ? o : mk; }  o = lane_xor64<4>(mk);  mk = o < mk ? o : mk;
                      o = lane_xor64<2>(mk);  mk = o < mk ? o : mk;  o = lane_xor64<1>(mk);  mk = o < mk ? o : mk; }
                    const unsigned geq = ((unsigned)(__ballot(len[0] > 0 && k0 == mk) >> gbase) & GMASK) |
                                         (((unsigned)(__ballot(len[1] > 0 && k1 == mk) >> gbase) & GMASK) << GS);
                    const int j = (many && geq) ? __ffs((int)geq) - 1 : 0;
                    uint32_t vb, vl;
                    fetch(j, vb, vl);
                    const uint32_t rbv[2] = {many ? vb : 0u, 0u};
                    const uint32_t rlv[2] = {many ? vl : 0u, 0u};
                    if (many) todo &= ~(1u << j);
                    scan_rows(rbv, rlv);
                    float gb = v1;
                    { float o;
                      if constexpr (GS == 16) { o = __uint_as_float(lane_xor32<8>(__float_as_uint(gb)));  gb = o < gb ? o : gb; }
                      o = __uint_as_float(lane_xor32<4>(__float_as_uint(gb)));  gb = o < gb ? o : gb;
                      o = __uint_as_float(lane_xor32<2>(__float_as_uint(gb)));  gb = o < gb ? o : gb;
                      o = __uint_as_float(lane_xor32<1>(__float_as_uint(gb)));  gb = o < gb ? o : gb; }
                    if (many && gb < __builtin_inff()) {
                        // gb + T bounds the own-frame squared distance of a cloud point from above, with room for the difference
                        // between the frames on both sides (T = 2 E): nothing beyond it can beat or tie the answer
                        const double c2 = ((double)gb + (double)T) * (1.0 + 1e-6);
                        // (the rows the hit's ball no longer reaches are dropped below; the others keep the x ranges of the pass's ball.
                        // Round 6: narrowing them to the hit's ball too -- the geometry and the offsets of every remaining row a second
                        // time -- saved 8 % of the candidates and cost more: this kernel is bound by the instructions it issues, a
                        // candidate is 13 of them.  Cold search of 1 M queries 1.97 -> 1.82 ms, profiles/r6/cold_search_no_second_geometry_pass.txt)
                        if (c2 < cull2) cull2 = c2;
                    }
                    todo &= ((unsigned)(__ballot(len[0] > 0 && lb2[0] <= cull2) >> gbase) & GMASK) |
                            (((unsigned)(__ballot(len[1] > 0 && lb2[1] <= cull2) >> gbase) & GMASK) << GS);
                }
                while (__any(todo != 0u)) {
                    uint32_t rbv[2], rlv[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const bool has = todo != 0u;
                        const int j = has ? __ffs((int)todo) - 1 : 0;
                        todo &= todo - 1u;
                        uint32_t vb, vl;
                        fetch(j, vb, vl);
                        rbv[u] = has ? vb : 0u; rlv[u] = has ? vl : 0u;
                    }
                    scan_rows(rbv, rlv);
                }
            }
        }
        // ---- the group's smallest value, and is it alone within T? ----
        float gmin = v1;
        { float o;
          if constexpr (GS == 16) { o = __uint_as_float(lane_xor32<8>(__float_as_uint(gmin)));  gmin = o < gmin ? o : gmin; }
          o = __uint_as_float(lane_xor32<4>(__float_as_uint(gmin)));  gmin = o < gmin ? o : gmin;
          o = __uint_as_float(lane_xor32<2>(__float_as_uint(gmin)));  gmin = o < gmin ? o : gmin;
          o = __uint_as_float(lane_xor32<1>(__float_as_uint(gmin)));  gmin = o < gmin ? o : gmin; }
        const bool found = gmin < __builtin_inff();
        // (rounded UP: the float32 addition may lose up to 6e-8 (gmin + T) -- near the cloud's centre that is a tenth of T -- and a
        // candidate that is within the margin must count as within it: ADVICE r5)
        const float lim = (gmin + T) * 1.0000003f;
        const unsigned near1 = (unsigned)(__ballot(found && v1 <= lim) >> gbase) & GMASK;       // lanes whose smallest is within T
        const unsigned near2 = (unsigned)(__ballot(found && v2 <= lim) >> gbase) & GMASK;       // ... whose second-smallest is too
        const bool alone = __popc(near1) == 1 && near2 == 0u;
        const unsigned eq = (unsigned)(__ballot(found && v1 == gmin) >> gbase) & GMASK;         // (a lane that holds the smallest value)
        const uint32_t wpos = (uint32_t)__shfl((int)bpos, gbase + (eq ? __ffs((int)eq) - 1 : 0));
        if (far_query) { far_defer = true; done = true; }
        if (huge) { defer = true; unplaced = true; done = true; }
        if (!done) {
            // exact contract distance of the winner (every lane of the group computes it: one address, one request)
            double best = __builtin_inf();
            double4 W = make_double4(0.0, 0.0, 0.0, 0.0);
            const double4 A = qrec[slot];
            if (found) {
                W = rec[wpos];
                double X = W.x, Y = W.y, Z = W.z;
                if (has_H) { double a2, b2, c2; xf(H, X, Y, Z, a2, b2, c2); X = a2; Y = b2; Z = c2; }
                const double dx = X - A.x, dy = Y - A.y, dz = Z - A.z;
                best = fma(dz, dz, fma(dy, dy, dx * dx));
            }
            const double r_eff = (r - slack) * (1.0 - 5e-13);
            const double r_eff2 = r_eff > 0.0 ? r_eff * r_eff * (1.0 - 1e-15) : -1.0;
            const bool fin = (found && best <= r_eff2) || all || r >= r_lim || last || (approx && found);
            if (fin) {
                if (found && !alone && !approx) {
                    // a tie within the margin: the exact kernel answers this query -- from the approximate winner, a cloud point
                    // within the margin of the answer: its search goes straight to that radius
                    defer = true;
                    if (gl == 1 && p2_out) { p2_out[3 * (long)q] = W.x; p2_out[3 * (long)q + 1] = W.y; p2_out[3 * (long)q + 2] = W.z; }
                } else {
                    const uint32_t bidx = found ? (uint32_t)(unsigned long long)__double_as_longlong(W.w) : 0u;
                    const bool ok = found && (best < max_d2);
                    const int64_t m = ok ? idx_base + (int64_t)bidx : (int64_t)-1;
                    if (!ok) W = make_double4(0.0, 0.0, 0.0, 0.0);
                    if (gl == 0 && d2_out) { d2_out[q] = ok ? best : __builtin_inf(); idx_out[q] = m; }
                    if (gl == 1 && p2_out) { p2_out[3 * (long)q] = W.x; p2_out[3 * (long)q + 1] = W.y; p2_out[3 * (long)q + 2] = W.z; }
                    // the bound of this slot's next search (a search that found nothing leaves none)
                    if (gl == 2) pslot[slot] = make_double4(ok ? W.x : __builtin_inf(), W.y, W.z, 0.0);
                }
                done = true;
            } else {
                // any cloud point's exact distance bounds the answer: the winner's does, alone or not
                r = found ? sqrt(best) * (1.0 + 1e-12) + slack : 1.4142135623730951 * r;
                last = found;
                if (r > r_lim) r = r_lim;
            }
        }
    }
    // (a query left to another kernel keeps its old bound in pslot: a cloud point still, so still a bound)
    if (!FAR && state && active && gl == 0) state[slot] = far_defer ? (uint8_t)1 : (uint8_t)0;
    // (a query float32 cannot place goes to the exact kernel with no bound at all: x = inf says so)
    if (defer && unplaced && gl == 1 && p2_out) p2_out[3 * (long)q] = __builtin_inf();
    {
        // the exact kernel's list: one addition per WAVE (data with coincident points defers every query: a million additions to
        // one word would cost milliseconds)
        const unsigned long long dm = __ballot(defer && gl == 0);
        if (dm) {
            unsigned base = 0;
            if (lane == __ffsll((long long)dm) - 1) base = atomicAdd(redo_count, (unsigned)__popcll((long long)dm));
            base = (unsigned)__builtin_amdgcn_readlane((int)base, __ffsll((long long)dm) - 1);
            if (defer && gl == 0) redo_list[base + (unsigned)__popcll((long long)(dm & ((1ull << lane) - 1ull)))] = q;
        }
    }
    if (work) {
        const unsigned long long c = wsum_u64((unsigned long long)n_cand), rws = wsum_u64((unsigned long long)n_rows);
        const unsigned long long n_def = wsum_u64((defer && gl == 0) ? 1ull : 0ull);
        if (lane == 0) { atomicAdd(work, c); atomicAdd(work + 1, rws); if (n_def) atomicAdd(work + 3, n_def); }
        if (slot == 0 && gl == 0 && !(FAR && state)) atomicAdd(work + 2, 1ull);      // (a search = one count: the full flavour behind the lean one is the same search)
    }
}

void launch_grid_nn16f(hipStream_t s, int lanes_per_query, bool far, const IcpDev *st, const void *qrec, void *pslot, long Q,
                       const GridGeom &G, const double c0[3], double eps_p, const uint32_t *cell_start,
                       const void *recf, const void *rec, bool xcd_order, const Xf *H,
                       const Xf *Hinv, double rmax, double max_d2, int64_t idx_base, double *d2_out, int64_t *idx_out, double *p2_out,
                       unsigned long long *work, int flags, uint8_t *state, uint32_t *redo_list, unsigned *redo_count)
{
    Xf id = {};
    FilterGeom F;
    for (int a = 0; a < 3; ++a) F.c0[a] = c0[a];
    F.eps_p = eps_p;
    const int has_H = H ? 1 : 0;
    unsigned g = lanes_per_query == 8 ? cdivf(Q, 32) : cdivf(Q, 16);
    if (xcd_order) g = (g + 7u) & ~7u;
#define SICP_NN16F_LAUNCH(GS_, FAR_)                                                                                                       \
    hipLaunchKernelGGL((k_grid_nn16f<GS_, FAR_>), dim3(g), dim3(256), 0, s, st, (const double4 *)qrec, (double4 *)pslot, cell_start,       \
                       (const float4 *)recf, (const double4 *)rec, Q, G, F, H ? *H : id, Hinv ? *Hinv : id, has_H, rmax, max_d2, \
                       idx_base, d2_out, idx_out, p2_out, work, flags, xcd_order ? 1 : 0, state, redo_list, redo_count)
    if (lanes_per_query == 8) { if (far) SICP_NN16F_LAUNCH(8, true); else SICP_NN16F_LAUNCH(8, false); }
    else { if (far) SICP_NN16F_LAUNCH(16, true); else SICP_NN16F_LAUNCH(16, false); }
#undef SICP_NN16F_LAUNCH
}

}  // namespace sicp
