// sicp_lm.hip -- the Levenberg-Marquardt minimisation of one ICP iteration for Q > SOLVE_MAX_Q, entirely on the
// device and chained like the small-Q tail (optimization.py:65-124,172-288; simpleicp.py:229-234,356-379):
//
//   k_lm_eval    one evaluation of the normal equations at the trial estimate, over many workgroups: every block
//                stages its rows [a0..a5 | r - shift | 1] in LDS and forms their 8x8 Gram matrix on the FP64 matrix
//                pipe (v_mfma_f64_16x16x4_f64, A = B), block partials go to memory, and the LAST block to arrive
//                (agent-scope release / acquire ticket) folds them in a fixed order and advances the solver's state
//                machine (accept / reject the trial, 6x6 LDL^T step, next trial or done) in device memory;
//   k_lm_finish  one workgroup: completes the minimisation itself in the rare case the enqueued evaluations did not
//                (so correctness never depends on how many were enqueued), then residual statistics, the convergence
//                test, the iteration's record (pinned host memory + ticket) and the next iteration's start.
//
// The host enqueues  match, distances, rejection, statistics, k_lm_eval x E, k_lm_finish  per iteration and several
// iterations ahead; evaluations that find the solver done (or the run over) exit at once.  Nothing waits for the host.
// Residuals are accumulated relative to a shift (the kept distances' mean) so that their variance comes out of the
// same pass without cancellation: no separate statistics launches.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <stdint.h>

#include "sicp_internal.h"
#include "sicp_lanes.h"
#include "sicp_solver.h"

namespace sicp {

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int LB = 512;                     // lanes per block, one correspondence per lane and chunk
constexpr int LCH = 2;                      // chunks staged per round
constexpr int LW = LB / 64;                 // waves per block

struct LmShared {
    double ja[LCH * LB][8];                 // this round's rows
    double gp[LW][2][64];                   // per-wave Gram blocks
    double gb[64];                          // this block's Gram (shifted residual column)
    double G[2][64];                        // last block / finish: accepted and trial Gram, unshifted
    double sc[16];                          // broadcast scalars
    int is_last;
};

// rows of the correspondences  chunk*LB + lane  for chunk = first, first + step, ... < nchunks, their Gram matrix
// (shifted residual column) summed over the block into S.gb; writes the plain residuals of the trial
__device__ void block_gram(LmShared &S, const double *__restrict__ qx, const double *__restrict__ qy, const double *__restrict__ qz,
                           const float *__restrict__ normals, const double *__restrict__ p2, const uint8_t *__restrict__ keep, long Q,
                           const double (&x)[6], const double (&sc)[6], double shift, long first, long step, long nchunks,
                           double *__restrict__ resid_t)
{
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    double H[12];
    euler_H(x, sc, H);
    const double s1 = sc[0], c1 = sc[1], s2 = sc[2], c2 = sc[3];
    const double w3y = -s1 * c2, w3z = c1 * c2;
    v4d acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
    // LCH chunks of LB correspondences per round: their loads are in flight together (a block is alone on its CU -- one
    // block per CU keeps the ticket cheap -- so memory latency has to be covered inside the block), one barrier pair per round
    for (long ch0 = first * LCH; ch0 < nchunks; ch0 += step * LCH) {
        double a[LCH][8], r[LCH];
        bool in[LCH];
#pragma unroll
        for (int c = 0; c < LCH; ++c) {
            const long i = (ch0 + c) * LB + tid;
            in[c] = i < Q;
            r[c] = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) a[c][k] = 0.0;
            // (everything a correspondence needs is asked for together with its verdict -- one round trip, not the verdict first and
            // the rest behind it: the quarter of the loads a rejected correspondence wastes costs less than the dependent round)
            const long ic = in[c] ? i : 0;
            const bool kp = keep[ic] != 0;
            const double px = p2[3 * ic], py = p2[3 * ic + 1], pz = p2[3 * ic + 2];
            const float fx = normals[3 * ic], fy = normals[3 * ic + 1], fz = normals[3 * ic + 2];
            const double q0 = qx[ic], q1 = qy[ic], q2 = qz[ic];
            if (in[c] && kp) {
                double X, Y, Z;
                xfm(H, px, py, pz, X, Y, Z);
                r[c] = pdist(X - q0, Y - q1, Z - q2, fx, fy, fz);
                const double nx = fx, ny = fy, nz = fz;
                const double ux = X - x[3], uy = Y - x[4], uz = Z - x[5];          // R p
                const double cx = uy * nz - uz * ny, cy = uz * nx - ux * nz, cz = ux * ny - uy * nx;   // (R p) x n
                a[c][0] = cx;
                a[c][1] = c1 * cy + s1 * cz;
                a[c][2] = s2 * cx + w3y * cy + w3z * cz;
                a[c][3] = nx; a[c][4] = ny; a[c][5] = nz;
                a[c][6] = r[c] - shift;
                a[c][7] = 1.0;
            }
        }
        __syncthreads();                                  // the previous round's rows have been consumed
#pragma unroll
        for (int c = 0; c < LCH; ++c) {
            const long i = (ch0 + c) * LB + tid;
            if (in[c]) resid_t[i] = r[c];
            double2 *row = reinterpret_cast<double2 *>(&S.ja[c * LB + tid][0]);
            row[0] = make_double2(a[c][0], a[c][1]); row[1] = make_double2(a[c][2], a[c][3]);
            row[2] = make_double2(a[c][4], a[c][5]); row[3] = make_double2(a[c][6], a[c][7]);
        }
        __syncthreads();
        // Gram product on the FP64 matrix pipe (layout: see sicp_tail.hip::eval_ne); chunks beyond the end hold zero rows
#pragma unroll
        for (int c = 0; c < LCH; ++c) {
            const double *src = &S.ja[c * LB + wid * 64 + 4 * ((lane >> 3) & 1) + (lane >> 4)][lane & 7];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double v = src[0], u = src[64];
                src += 128;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(u, u, acc2, 0, 0, 0);
            }
        }
    }
    acc += acc2;
    {
        const int n = lane & 15;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int m = (lane >> 4) + 4 * rg;
            if ((m >> 3) == (n >> 3)) S.gp[wid][m >> 3][(m & 7) * 8 + (n & 7)] = acc[rg];
        }
    }
    __syncthreads();
    if (tid < 64) {
        double g = 0.0;
#pragma unroll
        for (int w = 0; w < LW; ++w) g += S.gp[w][0][tid] + S.gp[w][1][tid];
        S.gb[tid] = g;
    }
    __syncthreads();
}

// The solver's state machine, one transition: the trial's Gram matrix (shifted residual column) is in S.gb.
// Run by the 64 lanes of wave 0 (uniform values; lane 0 stores the scalars, every lane its Gram entry).
// Same acceptance rules as the single-launch tail and the host solver.
__device__ void lm_advance(LmShared &S, LmDev *__restrict__ L, const TailArgs &A, const double *__restrict__ stats, double shift)
{
    const int lane = threadIdx.x;                          // caller guarantees threadIdx.x < 64
    int cur = L->cur;
    // un-shift the residual column: sum a r = sum a r' + s sum a, sum r = sum r' + s n, sum r^2 = sum r'^2 + 2 s sum r' + n s^2
    {
        const double g = S.gb[lane];
        const int u = lane >> 3, v = lane & 7;
        const double n = S.gb[7 * 8 + 7], s1r = S.gb[6 * 8 + 7];
        double out = g;
        if (v == 6 && u < 6) out = g + shift * S.gb[u * 8 + 7];
        if (u == 6 && v == 7) out = g + shift * n;
        if (u == 6 && v == 6) out = g + 2.0 * shift * s1r + n * shift * shift;
        S.G[cur ^ 1][lane] = out;
        S.G[cur][lane] = L->G[cur][lane];
    }
    const double S1 = S.gb[6 * 8 + 7], S2 = S.gb[6 * 8 + 6];
    double x[6], sc[6], xt[6], sct[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) { x[j] = L->x[j]; sc[j] = L->sc[j]; xt[j] = L->xt[j]; sct[j] = L->sct[j]; }
    double w = L->w;
    if (!(w > 0)) { const double dstd = stats[2]; w = 1.0 / (dstd * dstd); }          // simpleicp.py:233-234 (frozen afterwards)
    double cost = L->cost, lambda = L->lambda, dxmax = L->dxmax;
    int first = L->first, tries = L->tries, steps = L->steps, evals = L->evals + 1, done = 0;
    int nfree = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) nfree += (A.ow[j] < __builtin_inf()) ? 1 : 0;

    const double costn = objective(S.G[cur ^ 1], w, xt, A);
    if (first || costn <= cost * (1 + 1e-12) || dxmax < 1e-15) {
#pragma unroll
        for (int j = 0; j < 6; ++j) { x[j] = xt[j]; sc[j] = sct[j]; }
        cur ^= 1; cost = costn; tries = 0;
        L->G[cur][lane] = S.G[cur][lane];
        if (lane == 0) { L->stat[cur][0] = S1; L->stat[cur][1] = S2; }
        if (!first) {
            lambda = lambda > 0 ? lambda * 0.1 : 0.0;
            if (lambda < 1e-12) lambda = 0.0;
            ++steps;
            double xmax = 0.0;
#pragma unroll
            for (int j = 0; j < 6; ++j) xmax = fmax(xmax, fabs(x[j]));
            if (dxmax <= 1e-13 * (1.0 + xmax)) done = 1;
        }
        first = 0;
        if (steps >= A.max_steps || nfree == 0) done = 1;
    } else {
        lambda = lambda > 0 ? lambda * 10 : 1e-6;
        if (++tries >= 40) done = 1;
    }
    if (!done) {
        bool ok = false;
        double dstep[6];
        for (; tries < 40; ++tries) {
            ok = lm_step(S.G[cur], w, x, lambda, A, dstep);
            dxmax = 0.0;
#pragma unroll
            for (int j = 0; j < 6; ++j) { xt[j] = x[j] + dstep[j]; dxmax = fmax(dxmax, fabs(dstep[j])); }
            ok = ok && (dxmax < __builtin_inf());
            if (ok) break;
            lambda = lambda > 0 ? lambda * 10 : 1e-6;
        }
        double xm = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) xm = fmax(xm, fabs(x[j]));
        if (!ok) done = 1;
        else if (lambda == 0.0 && dxmax <= 1e-10 * (1.0 + xm)) done = 1;      // x is the minimiser to 1e-10 (the reference stops at 1e-8)
        else {
#pragma unroll
            for (int j = 0; j < 3; ++j) sincos_step(xt[j], dstep[j], sc[2 * j], sc[2 * j + 1], sct[2 * j], sct[2 * j + 1]);
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 6; ++j) { L->x[j] = x[j]; L->sc[j] = sc[j]; L->xt[j] = xt[j]; L->sct[j] = sct[j]; }
        L->w = w; L->cost = cost; L->lambda = lambda; L->dxmax = dxmax; L->shift = shift;
        L->cur = cur; L->first = first; L->tries = tries; L->steps = steps; L->evals = evals; L->done = done;
    }
}

}  // namespace

__global__ __launch_bounds__(LB) void k_lm_eval(
    const double *__restrict__ qx, const double *__restrict__ qy, const double *__restrict__ qz,
    const float *__restrict__ normals, const double *__restrict__ p2, const uint8_t *__restrict__ keep, long Q, TailArgs A,
    const IcpDev *__restrict__ st, LmDev *__restrict__ L, const double *__restrict__ stats /* n, mean, std of the kept distances */,
    double *__restrict__ partial /* [gridDim.x][64] */, unsigned *__restrict__ ticket, double *__restrict__ resid0,
    double *__restrict__ resid1, long chunk_lo, long chunk_hi /* this rank's LB-row chunks (all of them: 0, nchunks) */,
    double *__restrict__ gsum /* != NULL: leave the folded Gram here instead of advancing the solver (k_lm_advance does, after
                                 the ranks' sums have been added up) */)
{
    __shared__ LmShared S;
    if (st->stop || L->done || stats[0] < 6.0) return;             // uniform: run over / solver finished / too few correspondences
    const int tid = threadIdx.x;
    double x[6], sc[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) { x[j] = L->xt[j]; sc[j] = L->sct[j]; }
    const int slot = L->cur ^ 1;
    const double shift = L->first ? stats[1] : L->shift;
    // (block b starts at chunk_lo + b * LCH: block_gram takes its first round from `first * LCH`, so the range's start is
    // folded into the arrays' base -- chunk_lo is a multiple of LCH)
    block_gram(S, qx, qy, qz, normals, p2, keep, Q, x, sc, shift, chunk_lo / LCH + blockIdx.x, gridDim.x, chunk_hi, slot ? resid1 : resid0);
    if (tid < 64) partial[(long)blockIdx.x * 64 + tid] = S.gb[tid];
    // publish this block's partial, then take a ticket (stores -> vmcnt(0) -> barrier -> one-lane agent release ->
    // ticket; last arriver: agent acquire -> plain loads)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        S.is_last = (t == gridDim.x - 1) ? 1 : 0;
        if (S.is_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!S.is_last) return;
    // fold the block partials (fixed order for a given grid): the block's waves take eight partials each per step --
    // 8 * LW rows of 512 B in flight -- lane t sums entry t; a one-wave fold of 1024 partials cost more than the evaluation
    {
        const int wid = tid >> 6, lane = tid & 63;
        double s8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (unsigned b0 = 0; b0 < gridDim.x; b0 += 8 * LW) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned b = b0 + (unsigned)(wid * 8 + k);
                if (b < gridDim.x) s8[k] += partial[(long)b * 64 + lane];
            }
        }
        S.gp[wid][0][lane] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    }
    __syncthreads();
    if (tid < 64) {
        double g = 0.0;
#pragma unroll
        for (int w = 0; w < LW; ++w) g += S.gp[w][0][tid];
        S.gb[tid] = g;
        if (tid == 0) *ticket = 0;                          // re-arm for the next launch on this stream
        if (gsum) gsum[tid] = g;                            // sharded reduction: the job-wide sum comes first
        else lm_advance(S, L, A, stats, shift);
    }
}

// Sharded 6x6 reduction (gn_shard; SURVEY 8e step 3, the north star's "RCCL all-reduce of the 6x6 ATA / ATb and residual
// stats"): every rank evaluated its slice of the correspondences (k_lm_eval with gsum), ONE ncclAllReduce(sum) added the
// 64-double Gram blocks up -- the same bits on every rank --, and this one-wave launch advances the (replicated) solver.
__global__ __launch_bounds__(64) void k_lm_advance(TailArgs A, const IcpDev *__restrict__ st, LmDev *__restrict__ L,
                                                   const double *__restrict__ stats, const double *__restrict__ gsum)
{
    __shared__ LmShared S;
    if (st->stop || L->done || stats[0] < 6.0) return;             // the evaluation in front of the collective exited on the same test
    const double shift = L->first ? stats[1] : L->shift;
    S.gb[threadIdx.x] = gsum[threadIdx.x];
    lm_advance(S, L, A, stats, shift);
}

// One workgroup.  rj4: (m, median, mad, n_kept) of the rejection; stats: (n, mean, std) of the kept distances.
__device__ __forceinline__ void lm_finish_body(
    LmShared &S, double *out /* LDS, 64 */, const double *__restrict__ qx, const double *__restrict__ qy, const double *__restrict__ qz,
    const float *__restrict__ normals, const double *__restrict__ p2, const uint8_t *__restrict__ keep, long Q, const TailArgs &A,
    IcpDev *__restrict__ st, LmDev *__restrict__ L, const double *__restrict__ rj4, const double *__restrict__ stats,
    double *__restrict__ resid0, double *__restrict__ resid1, double *__restrict__ rec, bool barrier_failed = false)
{
    const int tid = threadIdx.x;
    if (tid < 64) out[tid] = 0.0;
    // a grid barrier of this iteration gave up waiting (its blocks were not all resident: another process holds CUs, a profiler
    // paused the queue) -- here in the minimisation, or in the rejection (k_hsel_all says so with a negative count): whatever was
    // folded is incomplete.  Report it and end the run; the host resets the barrier state.
    if (!st->stop && (barrier_failed || rj4[0] < 0.0)) {
        if (tid == 0) {
            rec[REC_STATUS] = 4.0;
            st->stop = 1;
            __threadfence_system();
            __hip_atomic_store(rec + REC_TICKET, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    if (st->stop) {
        if (tid == 0) {
            rec[REC_STATUS] = 3.0;
            __threadfence_system();
            __hip_atomic_store(rec + REC_TICKET, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    const double nk = stats[0];
    const bool too_few = nk < 6.0;
    // the enqueued evaluations normally finish the minimisation; if they did not, finish it here (one workgroup over
    // all correspondences per evaluation: slow, rare, and it keeps the result independent of how many were enqueued)
    if (!too_few) {
        const long nchunks = (Q + LB - 1) / LB;
        while (!__hip_atomic_load(&L->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            double x[6], sc[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) { x[j] = L->xt[j]; sc[j] = L->sct[j]; }
            const int slot = L->cur ^ 1;
            const double shift = L->first ? stats[1] : L->shift;
            block_gram(S, qx, qy, qz, normals, p2, keep, Q, x, sc, shift, 0, 1, nchunks, slot ? resid1 : resid0);
            if (tid < 64) lm_advance(S, L, A, stats, shift);
            __threadfence();
            __syncthreads();
        }
    }
    __syncthreads();
    if (tid >= 64) return;
    // ---- wave 0: residual statistics, convergence test, record, next iteration's start ----
    const int cur = L->cur;
    double x[6], sc[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) { x[j] = too_few ? st->x[j] : L->x[j]; sc[j] = too_few ? st->sc[j] : L->sc[j]; }
    const double gn = L->G[cur][7 * 8 + 7], S1 = L->stat[cur][0], S2 = L->stat[cur][1];
    const double rmean = L->shift + S1 / gn;
    const double rvar = (S2 - S1 * S1 / gn) / gn;
    const double rstd = sqrt(rvar > 0.0 ? rvar : 0.0);
    const double cost = L->cost, w = L->w;
    const bool finite = too_few || cost < __builtin_inf();
    const int done_iters = st->done_iters;
    bool conv = false;
    if (!too_few && A.min_change >= 0.0 && done_iters > 0 && finite) {
        const double pm = st->prev_mean, ps = st->prev_std;
        const double cm = pm == 0.0 ? (rmean == 0.0 ? 0.0 : __builtin_inf()) : fabs((rmean - pm) / pm * 100.0);
        const double cs = ps == 0.0 ? (rstd == 0.0 ? 0.0 : __builtin_inf()) : fabs((rstd - ps) / ps * 100.0);
        conv = cm < A.min_change && cs < A.min_change;
    }
    if (tid < 30 && !too_few) {
        int u = 0, v = 0;
        if (tid < 21) { int t = tid; while (t >= 6 - u) { t -= 6 - u; ++u; } v = u + t; }
        else if (tid < 27) { u = tid - 21; v = 6; }
        else if (tid == 27) { u = 6; v = 7; }
        else if (tid == 28) { u = 6; v = 6; }
        else { u = 7; v = 7; }
        out[20 + tid] = L->G[cur][u * 8 + v];
    }
    if (tid == 0) {
        out[0] = rj4[0]; out[1] = rj4[1]; out[2] = rj4[2]; out[3] = rj4[3]; out[4] = stats[1]; out[5] = stats[2];
#pragma unroll
        for (int j = 0; j < 6; ++j) out[10 + j] = x[j];
        if (too_few) out[REC_STATUS] = 1.0;
        else {
            out[6] = w; out[7] = cost; out[8] = L->steps; out[9] = L->evals;
            out[16] = rmean; out[17] = rstd;
            out[REC_STATUS] = finite ? 0.0 : 2.0;
            out[REC_CONVERGED] = conv ? 1.0 : 0.0;
            out[REC_RESID_SLOT] = cur;
        }
    }
    // (tid < 64; the ticket word is overwritten last, below.  System-scope write-through stores, drained, then the ticket: no fence --
    // the host reads nothing else, and a system-scope release would write this XCD's L2 back on the iteration's critical path:
    // sicp_tail.hip, flush_rec)
    __hip_atomic_store(rec + tid, out[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {
        __hip_atomic_store(rec + REC_TICKET, A.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // the next iteration's start
        if (too_few || !finite) { st->stop = 1; }
        else {
            double Hn[12];
            euler_H(x, sc, Hn);
#pragma unroll
            for (int j = 0; j < 6; ++j) { st->x[j] = x[j]; st->sc[j] = sc[j]; }
#pragma unroll
            for (int j = 0; j < 12; ++j) st->H.m[j] = Hn[j];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < 3; ++j) st->Hinv.m[4 * i + j] = Hn[4 * j + i];
                st->Hinv.m[4 * i + 3] = -(Hn[i] * Hn[3] + Hn[4 + i] * Hn[7] + Hn[8 + i] * Hn[11]);
            }
            st->w = w; st->prev_mean = rmean; st->prev_std = rstd;
            st->done_iters = done_iters + 1;
            if (conv) st->stop = 1;
            // and the solver's: trial = the new estimate, always accepted
#pragma unroll
            for (int j = 0; j < 6; ++j) { L->xt[j] = x[j]; L->sct[j] = sc[j]; }
            L->w = w; L->cost = 0.0; L->lambda = 0.0; L->dxmax = 0.0;
            L->first = 1; L->tries = 0; L->steps = 0; L->evals = 0; L->done = 0;
        }
    }
}

__global__ __launch_bounds__(LB) void k_lm_finish(
    const double *__restrict__ qx, const double *__restrict__ qy, const double *__restrict__ qz,
    const float *__restrict__ normals, const double *__restrict__ p2, const uint8_t *__restrict__ keep, long Q, TailArgs A,
    IcpDev *__restrict__ st, LmDev *__restrict__ L, const double *__restrict__ rj4, const double *__restrict__ stats,
    double *__restrict__ resid0, double *__restrict__ resid1, double *__restrict__ rec)
{
    __shared__ LmShared S;
    __shared__ double out[64];
    lm_finish_body(S, out, qx, qy, qz, normals, p2, keep, Q, A, st, L, rj4, stats, resid0, resid1, rec);
}

// ------------------------------------------------------------------------------------
// The whole minimisation of an iteration in ONE launch (default without a sharded reduction): evaluations are phases of a single
// kernel separated by grid barriers (sicp_lanes.h), so exactly the evaluations the solver needs are run -- the
// launch-per-evaluation form enqueues `lm_evals` of them plus the finish and the unused ones exit at ~4 us apiece.
//   * one barrier per evaluation: blocks write their 8x8 partials (two alternating buffers), meet, and then EVERY block folds
//     all partials in the same fixed order and advances its OWN copy of the solver state (LDS) -- same numbers, same decisions,
//     no second meeting to broadcast the next trial;
//   * after the last evaluation block 0 stores the solver state and does what k_lm_finish does (residual statistics,
//     convergence test, record + ticket, next iteration's start; and the completion loop in the never-seen case that
//     LM_MAXB evaluations were not enough).
// Same arithmetic in the same order as the launch-per-evaluation form with the same grid: bit-identical results.
// ------------------------------------------------------------------------------------
constexpr int LM_MAXB = 24;
static_assert(4 * 8 * LW >= 256, "the fold of the block partials covers lm_eval_grid's cap in four steps");
// AHEAD: the instantiation for grids of more than 8 * LW blocks (its fold keeps 32 loads per lane in flight: registers the small grids'
// instantiation does better without -- with both forms in ONE kernel the minimisation of 10 000 correspondences took 28 us instead of 23.5)
template <bool AHEAD>
__global__ __launch_bounds__(LB) void k_lm_all(
    const double *__restrict__ qx, const double *__restrict__ qy, const double *__restrict__ qz,
    const float *__restrict__ normals, const double *__restrict__ p2, const uint8_t *__restrict__ keep, long Q, TailArgs A,
    IcpDev *__restrict__ st, LmDev *__restrict__ L, const double *__restrict__ rj4, const double *__restrict__ stats,
    double *__restrict__ partial /* [2][gridDim.x][64] */, GridBar *__restrict__ B, unsigned long long bar_base,
    double *__restrict__ resid0, double *__restrict__ resid1, double *__restrict__ rec, unsigned absent /* test hook: see grid_barrier */)
{
    __shared__ LmShared S;
    __shared__ LmDev Ls;
    __shared__ double out[64];
    const int tid = threadIdx.x;
    int nb = 0;
    if (st->stop || stats[0] < 6.0 || rj4[0] < 0.0) {
        // run over, too few correspondences, or a rejection whose barrier failed: nothing to minimise -- block 0 still reports
        // (as k_lm_finish does)
        if (blockIdx.x == 0) lm_finish_body(S, out, qx, qy, qz, normals, p2, keep, Q, A, st, L, rj4, stats, resid0, resid1, rec);
        return;
    }
    {   // every block starts from the same solver state
        const double *src = reinterpret_cast<const double *>(L);
        double *dst = reinterpret_cast<double *>(&Ls);
        for (int i = tid; i < (int)(sizeof(LmDev) / sizeof(double)); i += LB) dst[i] = src[i];
    }
    __syncthreads();
    const long nchunks = (Q + LB - 1) / LB;
    const unsigned g = gridDim.x;
    while (!Ls.done && nb < LM_MAXB) {
        double x[6], sc[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) { x[j] = Ls.xt[j]; sc[j] = Ls.sct[j]; }
        const int slot = Ls.cur ^ 1;
        const double shift = Ls.first ? stats[1] : Ls.shift;
        block_gram(S, qx, qy, qz, normals, p2, keep, Q, x, sc, shift, blockIdx.x, g, nchunks, slot ? resid1 : resid0);
        double *mypart = partial + ((long)(nb & 1) * g + blockIdx.x) * 64;
        if (tid < 64) __hip_atomic_store(&mypart[tid], S.gb[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ++nb;
        grid_barrier(B, bar_base + (unsigned long long)nb, absent);
        // fold the block partials in the launch-per-evaluation form's order: the block's waves take eight partials each per step
        {
            const double *all = partial + (long)((nb - 1) & 1) * g * 64;
            const int wid = tid >> 6, lane = tid & 63;
            double s8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if constexpr (!AHEAD) {
                for (unsigned b0 = 0; b0 < g; b0 += 8 * LW) {          // (one step here: g <= 8 * LW)
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const unsigned b = b0 + (unsigned)(wid * 8 + k);
                        if (b < g) s8[k] += __hip_atomic_load(&all[(long)b * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            } else {
                // more blocks than one step covers (up to 256: four steps): every partial of this wave's share is asked for before
                // the first is added -- 32 loads per lane in flight instead of four dependent rounds of eight; the additions keep their
                // order (same bits).  Minimisation at Q = 1 M: 72 -> 66 us per iteration.  (profiles/r6/q_sweep_lm_fold_loads_in_flight.txt)
                double pv[4][8];
#pragma unroll
                for (int st4 = 0; st4 < 4; ++st4)
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const unsigned b = (unsigned)st4 * 8u * LW + (unsigned)(wid * 8 + k);
                        pv[st4][k] = b < g ? __hip_atomic_load(&all[(long)b * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
                    }
#pragma unroll
                for (int st4 = 0; st4 < 4; ++st4)
#pragma unroll
                    for (int k = 0; k < 8; ++k) { const unsigned b = (unsigned)st4 * 8u * LW + (unsigned)(wid * 8 + k); if (b < g) s8[k] += pv[st4][k]; }
            }
            S.gp[wid][0][lane] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
        }
        __syncthreads();
        if (tid < 64) {
            double gsum = 0.0;
#pragma unroll
            for (int w = 0; w < LW; ++w) gsum += S.gp[w][0][tid];
            S.gb[tid] = gsum;
            lm_advance(S, &Ls, A, stats, shift);
        }
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        // the finish works on this block's LDS copy of the solver state (it reads what the loop left and leaves the next iteration's
        // start there), and only then does the state go to memory -- the next launch reads it, a kernel boundary away.  (Round 5
        // stored the state first and had all 512 lanes execute __threadfence() so that the finish could read it back: an L2
        // write-back + invalidate on the iteration's critical path.)
        const bool bad = __hip_atomic_load(&B->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        lm_finish_body(S, out, qx, qy, qz, normals, p2, keep, Q, A, st, &Ls, rj4, stats, resid0, resid1, rec, bad);
        __syncthreads();
        double *dst = reinterpret_cast<double *>(L);
        const double *src = reinterpret_cast<const double *>(&Ls);
        for (int i = tid; i < (int)(sizeof(LmDev) / sizeof(double)); i += LB) dst[i] = src[i];
    }
}

int lm_eval_grid(long Q)
{
    // (one block per CU: same-address ticket atomics serialise at ~20 ns apiece across the 8 XCDs)
    const long cap = 256;                             // (the folds below take four steps of 8 * LW partials: static_assert)
    long g = ((Q + LB - 1) / LB + LCH - 1) / LCH;     // one round of LCH chunks per block, up to the cap
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

void launch_lm_eval(hipStream_t s, const double *qx, const double *qy, const double *qz, const float *normals, const double *p2,
                    const uint8_t *keep, long Q, const TailArgs &A, const IcpDev *st, LmDev *L, const double *stats, double *partial,
                    unsigned *ticket, double *resid0, double *resid1, int rank, int world, double *gsum)
{
    // this rank's share of the LB-row chunks, in whole rounds of LCH chunks (world == 1: all of them)
    const long nchunks = (Q + LB - 1) / LB;
    const long rounds = (nchunks + LCH - 1) / LCH, per = (rounds + world - 1) / world;
    const long lo = std::min(rounds, per * rank) * LCH, hi = std::min(nchunks, std::min(rounds, per * (rank + 1)) * LCH);
    const long mine = std::max(0L, hi - lo) * LB;
    hipLaunchKernelGGL(k_lm_eval, dim3(lm_eval_grid(mine)), dim3(LB), 0, s, qx, qy, qz, normals, p2, keep, Q, A, st, L, stats,
                       partial, ticket, resid0, resid1, lo, hi, gsum);
}

// one launch for the whole minimisation; *bar_total: what the launches on `bar` have added to its counter so far
void launch_lm_all(hipStream_t s, const double *qx, const double *qy, const double *qz, const float *normals, const double *p2,
                   const uint8_t *keep, long Q, const TailArgs &A, IcpDev *st, LmDev *L, const double *rj4, const double *stats,
                   double *partial, void *bar, unsigned long long *bar_total, double *resid0, double *resid1, double *rec, unsigned absent)
{
    // every block must be resident at once (grid barrier): never more blocks than the device can hold
    const bool ahead = lm_eval_grid(Q) > 8 * LW;
    const long resident = ahead ? resident_blocks((const void *)k_lm_all<true>, LB) : resident_blocks((const void *)k_lm_all<false>, LB);
    const unsigned g = (unsigned)std::min<long>(lm_eval_grid(Q), resident);
    if (ahead && g > 8u * LW)
        hipLaunchKernelGGL(k_lm_all<true>, dim3(g), dim3(LB), 0, s, qx, qy, qz, normals, p2, keep, Q, A, st, L, rj4, stats, partial,
                           (GridBar *)bar, *bar_total, resid0, resid1, rec, absent);
    else
        hipLaunchKernelGGL(k_lm_all<false>, dim3(std::min(g, 8u * LW)), dim3(LB), 0, s, qx, qy, qz, normals, p2, keep, Q, A, st, L, rj4, stats, partial,
                           (GridBar *)bar, *bar_total, resid0, resid1, rec, absent);
    *bar_total += (unsigned long long)LM_MAXB;
}
size_t lm_bar_bytes() { return sizeof(GridBar); }

void launch_lm_advance(hipStream_t s, const TailArgs &A, const IcpDev *st, LmDev *L, const double *stats, const double *gsum)
{
    hipLaunchKernelGGL(k_lm_advance, dim3(1), dim3(64), 0, s, A, st, L, stats, gsum);
}

void launch_lm_finish(hipStream_t s, const double *qx, const double *qy, const double *qz, const float *normals, const double *p2,
                      const uint8_t *keep, long Q, const TailArgs &A, IcpDev *st, LmDev *L, const double *rj4, const double *stats,
                      double *resid0, double *resid1, double *rec)
{
    hipLaunchKernelGGL(k_lm_finish, dim3(1), dim3(LB), 0, s, qx, qy, qz, normals, p2, keep, Q, A, st, L, rj4, stats, resid0, resid1, rec);
}

}  // namespace sicp
