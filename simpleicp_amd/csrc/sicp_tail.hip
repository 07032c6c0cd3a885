// sicp_tail.hip -- everything after the match of one ICP iteration in ONE launch of ONE workgroup
// (Q <= SOLVE_MAX_Q), chained on the device: the kernel takes the estimate it starts from out of the
// device-resident loop state (IcpDev) and leaves the next one there (together with H(x), its inverse, the
// frozen distance weight and the convergence verdict), so the host can enqueue the iterations of a run back to
// back -- match, tail, match, tail, ... -- and only reads the per-iteration records the kernel streams into
// pinned memory.
//
//   (point-to-plane distances + planarity flags arrive with the matches: corrpts.py:139-163,195-211 is the match kernel's
//    epilogue, k_grid_nn / post_match, or k_postmatch behind a multi-GPU exchange)
//   median / raw-MAD rejection                            corrpts.py:165-188
//   kept-distance statistics, automatic weight            simpleicp.py:229-234
//   Levenberg-Marquardt on fused 6x6 normal equations     optimization.py:65-124,172-288
//   residual statistics + convergence test                simpleicp.py:356-379
//
// At Q ~ 1000 everything here is latency, nothing is bandwidth -- and that includes INSTRUCTION FETCH: a
// single workgroup runs every instruction once or a few times from a cold instruction cache, so the code is
// written to be small (loops instead of unrolled copies, one call site per building block):
//   * every correspondence lives in registers for the whole kernel (one global round trip, up front);
//   * exact order statistics by RANGE-HISTOGRAM SELECTION instead of sorting: 256 bins over the current key
//     interval, the bin holding the wanted rank becomes the next interval (1-3 rounds on real distances), the
//     last <= 8 keys are ranked with one ballot;
//   * the 29 normal-equation sums are ONE Gram product on the FP64 matrix pipe: rows [a0..a5 | r | 1] staged in
//     LDS, v_mfma_f64_16x16x4_f64 with A = B (two 8x8 Gram blocks per instruction, 8 correspondences each);
//   * the 6x6 solve runs redundantly in every lane (LDL^T in registers): no cross-lane traffic, no barrier;
//   * record and loop state leave the kernel as ONE store instruction each (lanes of wave 0).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sicp_internal.h"
#include "sicp_lanes.h"
#include "sicp_solver.h"

namespace sicp {

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long okey(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double oval(unsigned long long k)
{
    const unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

constexpr int TB = 256;                    // 4 waves, one per SIMD: each may use the whole 512-entry register file
                                           // (measured in round 6 with 8 waves: two waves share a SIMD's 512 registers, the LM state spills
                                           // to scratch and the iteration goes 24.3 -> 28.3 us: profiles/r6/q_sweep_tail_512_lanes.txt)
                                           // (the uniform solver state and up to 8 correspondences per lane live there)
constexpr int NW = TB / 64;
constexpr int HC = 16;                     // privatised histogram copies of a selection round
constexpr int CAND_MAX = 8;                // keys left when a selection stops binning and ranks them directly
constexpr int WCAP = 64;                   // keys a selection window may hold (one per lane of the wave that ranks them)
constexpr unsigned long long NOKEY = ~0ull;
constexpr int ST_DOUBLES = sizeof(IcpDev) / sizeof(double);
static_assert(sizeof(IcpDev) % sizeof(double) == 0 && ST_DOUBLES <= 64, "the loop state leaves the kernel as one 64-lane store");

#ifdef SICP_EVAL_FINE_TRACE
#define SICP_ET(i) if (threadIdx.x == 0) S.evt[i] = clock64();
#else
#define SICP_ET(i)
#endif

struct TailShared {
    double ja[SOLVE_MAX_Q][8];             // staged rows [a0..a5 | r | 1] of the kept correspondences (128 KiB at Q = 2048)
    double gp[NW][2][64];                  // per-wave Gram blocks (matrix-pipe form)
    double gw[2][NW][32];                  // per-wave sums of an evaluation (two evaluations' worth)
    double gf[NW][2][64];                  // per-wave copies of the reduced Gram matrix: current estimate / trial
    double red[2][NW][4];                  // block sums: one slot per call site, no reuse hazards
    double dmm[NW][2];                     // per-wave (min, -max) of the flagged distances
    unsigned long long wmin[NW];           // per-wave smallest member key above a selection's final interval
#ifdef SICP_SEL_FINE_TRACE
    long long selt[2][8]; int selw;        // cycle stamps inside the two selections (trace build only)
#endif
#ifdef SICP_EVAL_FINE_TRACE
    long long evt[8];                      // cycle stamps inside the last evaluation (trace build only)
#endif
    unsigned tot[256];                     // folded histogram of a selection round
    unsigned long long wc[2][NW][WCAP];    // windowed selection: per-wave candidate keys (one buffer per statistic: no barrier between
    alignas(16) unsigned wci[2][NW];       //  the median's readers and the MAD's writers), how many (read as one uint4),
    alignas(16) unsigned wbl[2][NW];       //  and how many member keys lie below the window
    unsigned long long cand[CAND_MAX];
    unsigned ncand;
    unsigned wcnt[NW];
    double out[64];                        // the record on its way out (wave 0)
    double out2[64];                       // the next iteration's loop state on its way out (wave 1, at the same time)
};

// block-wide sums of NV values per lane; every lane returns with the totals (fixed order)
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double (*buf)[4])
{
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NV; ++i) { v[i] = wsum(v[i]); if (lane == 0) buf[wid][i] = v[i]; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double t = (buf[0][i] + buf[1][i]) + (buf[2][i] + buf[3][i]);
        if constexpr (NW == 8) t += (buf[4][i] + buf[5][i]) + (buf[6][i] + buf[7][i]);
        v[i] = t;
    }
}

__device__ __forceinline__ unsigned long long wmin_u64(unsigned long long v)
{
    unsigned long long o;
    o = lane_xor64<32>(v); v = o < v ? o : v;  o = lane_xor64<16>(v); v = o < v ? o : v;
    o = lane_xor64<8>(v);  v = o < v ? o : v;  o = lane_xor64<4>(v);  v = o < v ? o : v;
    o = lane_xor64<2>(v);  v = o < v ? o : v;  o = lane_xor64<1>(v);  v = o < v ? o : v;
    return v;
}
__device__ __forceinline__ double wmin_f64(double v)
{
    v = fmin(v, lane_xor_f64<32>(v)); v = fmin(v, lane_xor_f64<16>(v)); v = fmin(v, lane_xor_f64<8>(v));
    v = fmin(v, lane_xor_f64<4>(v));  v = fmin(v, lane_xor_f64<2>(v));  v = fmin(v, lane_xor_f64<1>(v));
    return v;
}
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

// Exact order statistics of the block's member keys (EPT per lane, NOKEY = not a member; m >= 1 members):
// ka = key of rank r (0-based), kb = key of rank r + 1 when want2 (else ka).  Every lane gets both.
//   [lo, hi] : an interval that contains every member key;
//   round    : 256 bins of width 2^sh over [lo, hi]; the bin that holds the rank becomes the next [lo, hi] (integer
//              arithmetic on the order-preserving keys: monotone, exact).  Real distances share their leading bits, so
//              whole waves hit one bin in the first round -- and same-address LDS atomics cost ~10-20 cycles PER LANE.
//              The histogram is therefore kept in HC privatised copies (lane & 15 picks one, rows padded to 257 words:
//              at most 4 lanes of an instruction can meet); 256 lanes then fold one bin each, and every wave scans the
//              totals itself;
//   end      : the bin holds <= CAND_MAX keys (ranked with one ballot: lane 8 i + j compares keys i and j) or is a
//              single key value (duplicates: quantised clouds).
// hc (HC x 257 words, carved out of the row staging area) and S.ncand must be zero on entry and are zero again on exit.
template <int EPT>
__device__ __forceinline__ void block_select(TailShared &S, unsigned *hc, const unsigned long long (&k)[EPT], long r, bool want2,
                                             unsigned long long lo, unsigned long long hi, unsigned long long &ka,
                                             unsigned long long &kb, int &rounds_out)
{
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    unsigned long long below = 0;            // members with key < lo
    unsigned cs = 0;                         // members inside [lo, hi] once the loop ends
    int sh = 0;
    unsigned *mycopy = hc + (lane & (HC - 1)) * 257;
#ifdef SICP_SEL_FINE_TRACE
#define SICP_ST(i) if (tid == 0) S.selt[S.selw][i] = clock64();
#else
#define SICP_ST(i)
#endif
    SICP_ST(0)
#pragma unroll 1
    for (int round = 0; round < 10; ++round) {
        const unsigned long long range = hi - lo;
        sh = range < 256ull ? 0 : (64 - __clzll((long long)range)) - 8;
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if (k[e] != NOKEY && k[e] >= lo && k[e] <= hi) atomicAdd(&mycopy[(unsigned)((k[e] - lo) >> sh)], 1u);
        __syncthreads();
        if (round == 0) { SICP_ST(1) }
        if (TB == 256 || tid < 256) {   // lane t folds bin t over the copies and leaves them clean for the next round
            unsigned tot = 0;
#pragma unroll
            for (int c = 0; c < HC; ++c) { tot += hc[c * 257 + tid]; hc[c * 257 + tid] = 0u; }
            S.tot[tid] = tot;
        }
        __syncthreads();
        if (round == 0) { SICP_ST(2) }
        // every wave scans the 256 totals on its own (lane l owns bins 4l..4l+3)
        const uint4 h4 = *reinterpret_cast<const uint4 *>(&S.tot[4 * lane]);
        const unsigned mine = h4.x + h4.y + h4.z + h4.w;
        const unsigned incl = wscan_u32(mine);
        const unsigned long long t = (unsigned long long)r - below;           // rank inside [lo, hi]
        const unsigned long long gt = __ballot((unsigned long long)incl > t);
        const int L = __ffsll((long long)gt) - 1;                               // the lane whose bins hold rank t
        const unsigned eL = (unsigned)__builtin_amdgcn_readlane((int)(incl - mine), L);
        const unsigned a0 = (unsigned)__builtin_amdgcn_readlane((int)h4.x, L), a1 = (unsigned)__builtin_amdgcn_readlane((int)h4.y, L);
        const unsigned a2 = (unsigned)__builtin_amdgcn_readlane((int)h4.z, L), a3 = (unsigned)__builtin_amdgcn_readlane((int)h4.w, L);
        unsigned acc = eL; int j = 0; cs = a0;
        if (t >= (unsigned long long)acc + a0) { acc += a0; j = 1; cs = a1;
            if (t >= (unsigned long long)acc + a1) { acc += a1; j = 2; cs = a2;
                if (t >= (unsigned long long)acc + a2) { acc += a2; j = 3; cs = a3; } } }
        const unsigned s = 4u * (unsigned)L + (unsigned)j;
        below += acc;
        lo = lo + ((unsigned long long)s << sh);
        if (sh > 0) { const unsigned long long top = lo + ((1ull << sh) - 1ull); hi = top < hi ? top : hi; } else hi = lo;
        rounds_out = round + 1;
        if (round == 0) { SICP_ST(3) }
        if (cs <= (unsigned)CAND_MAX || sh == 0) break;
    }
    SICP_ST(4)
    const unsigned long long t = (unsigned long long)r - below;                 // rank inside the final interval
    const bool need_above = want2 && t + 1 >= cs;                                // the next rank lies above the final interval
    if (sh == 0 || lo == hi) {
        ka = lo; kb = lo;                                                         // one key value (cs copies of it)
    } else {
        // <= CAND_MAX members left: list them (at most 8 lanes touch the counter), then rank all pairs with one ballot
        // (measured and not adopted: listing them per wave by ballot compaction, no atomics -- the gather got 100 cycles cheaper
        // and the ranking 500 dearer, its lane-dependent LDS addresses are worth more than the atomics' return trip)
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if (k[e] != NOKEY && k[e] >= lo && k[e] <= hi) { const unsigned slot = atomicAdd(&S.ncand, 1u); if (slot < (unsigned)CAND_MAX) S.cand[slot] = k[e]; }
        __syncthreads();
        SICP_ST(5)
        if (tid == 0) S.ncand = 0u;             // (every gather atomic is behind the barrier; the next use is barriers away)
        const int ci = lane >> 3, cj = lane & 7;
        const unsigned long long vi = S.cand[ci], vj = S.cand[cj];
        const bool before = ci != cj && (unsigned)ci < cs && (unsigned)cj < cs && (cj < ci ? vj <= vi : vj < vi);   // key j sorts before key i
        const unsigned long long M = __ballot(before);
        const unsigned rk = (unsigned)__popcll((long long)((M >> (8 * (lane & 7))) & 0xffull));   // lane i < 8: rank of key i
        const unsigned long long mine = S.cand[lane & 7];
        const bool valid = lane < 8 && (unsigned)lane < cs;
        const unsigned long long ha = __ballot(valid && (unsigned long long)rk == t);
        ka = readlane_u64(mine, __ffsll((long long)ha) - 1);
        kb = ka;
        if (want2 && !need_above) {
            const unsigned long long hb = __ballot(valid && (unsigned long long)rk == t + 1);
            kb = readlane_u64(mine, __ffsll((long long)hb) - 1);
        }
    }
    if (need_above) {
        // the smallest member key greater than hi: wave minimum (register moves), then one slot per wave
        unsigned long long nx = NOKEY;
#pragma unroll
        for (int e = 0; e < EPT; ++e) if (k[e] != NOKEY && k[e] > hi) nx = k[e] < nx ? k[e] : nx;
        nx = wmin_u64(nx);
        if (lane == 0) S.wmin[wid] = nx;
        __syncthreads();
        unsigned long long b = S.wmin[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) { const unsigned long long a = S.wmin[w]; b = a < b ? a : b; }
        kb = b;
    }
    SICP_ST(6)
}


// ---- order statistics from a WINDOW (the selection of a settled run) --------------------------------------------------------
// From a run's fourth or fifth iteration on, median and MAD of the distances are where the last iteration left them, give or take
// a few per cent of a MAD (profiles/r6/sel_dynamics.txt) -- and the range-histogram selection above spends 4 k cycles per statistic
// finding that out again.  So the loop state carries both (IcpDev::sel_*), and a launch first looks at a window around each: every
// wave counts its member keys BELOW the window and lists those INSIDE it (ballot compaction, no atomics), one barrier, then every
// wave ranks the <= 64 listed keys itself (lane i holds key i and counts the keys that sort before it: register broadcasts, no
// LDS).  If the wanted rank (ranks) falls inside the list, that IS the exact order statistic -- same keys, same total order as
// block_select, so the same bits; if not (the estimate still moves, the window overflows) block_select runs as before.
// window_collect: call before a barrier;  window_pick: after it (false = the window missed).
template <int EPT>
__device__ __forceinline__ void window_collect(TailShared &S, int buf, const unsigned long long (&k)[EPT], unsigned long long wlo,
                                               unsigned long long whi)
{
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // a lane's own counts first, ONE wave prefix sum for both (inside: low half, below: high half; <= 64 * EPT each), then the
    // lane writes its keys behind its predecessors' (a ballot + bit count per key put four dependent vector -> scalar -> vector
    // round trips on a single wave's critical path: 1.2 k cycles against 0.25 k)
    unsigned cin = 0, cbel = 0;
    bool inw[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const bool mem = k[e] != NOKEY;
        inw[e] = mem && k[e] >= wlo && k[e] <= whi;
        cin += inw[e] ? 1u : 0u;
        cbel += (mem && k[e] < wlo) ? 1u : 0u;
    }
    const unsigned incl = wscan_u32(cin | (cbel << 16));
    unsigned slot = (incl & 0xffffu) - cin;
#pragma unroll
    for (int e = 0; e < EPT; ++e)
        if (inw[e]) { if (slot < (unsigned)WCAP) S.wc[buf][wid][slot] = k[e]; ++slot; }
    if (lane == 63) { S.wci[buf][wid] = incl & 0xffffu; S.wbl[buf][wid] = incl >> 16; }
}
__device__ __forceinline__ bool window_pick(const TailShared &S, int buf, long r, bool want2, unsigned long long &ka, unsigned long long &kb)
{
    static_assert(NW == 4 || NW == 8, "the waves' counts are read four at a time");
    const int lane = threadIdx.x & 63;
    unsigned n = 0, my_w = 0, my_base = 0;
    long below = 0;
#pragma unroll
    for (int w4 = 0; w4 < NW; w4 += 4) {
        const uint4 ci = *reinterpret_cast<const uint4 *>(&S.wci[buf][w4]), bl = *reinterpret_cast<const uint4 *>(&S.wbl[buf][w4]);
        const unsigned cc[4] = {ci.x, ci.y, ci.z, ci.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if ((unsigned)lane >= n && (unsigned)lane < n + cc[j]) { my_w = (unsigned)(w4 + j); my_base = n; }
            n += cc[j];
        }
        below += (long)((bl.x + bl.y) + (bl.z + bl.w));
    }
    long t = r - below;                                          // rank inside the window
    if (n > (unsigned)WCAP || t < 0 || t + (want2 ? 1 : 0) >= (long)n) return false;
    unsigned long long mine = NOKEY;
    if ((unsigned)lane < n) mine = S.wc[buf][my_w][(unsigned)lane - my_base];
    // the key of rank t among the n listed ones, by QUICKSELECT on the total order (key, lane) with wave-uniform bookkeeping: the
    // pivot is the first lane still in play (the list is in no particular order), one ballot counts the keys before it -- a
    // handful of rounds of ~10 instructions where counting every key against every other took n rounds
    unsigned long long active = n >= 64u ? ~0ull : ((1ull << n) - 1ull);
    int pa = 0;
#pragma unroll 1
    for (;;) {
        const int p = __ffsll((long long)active) - 1;
        const unsigned long long pv = readlane_u64(mine, p);
        const unsigned long long less = __ballot(mine < pv || (mine == pv && lane < p)) & active;
        const long c = (long)__popcll((long long)less);
        if (c == t) { ka = pv; pa = p; break; }
        if (c > t) active = less;
        else { active &= ~less & ~(1ull << p); t -= c + 1; }
    }
    kb = ka;
    if (want2) {
        // the successor of (ka, pa) in that order: the smallest listed key behind it
        const bool behind = (unsigned)lane < n && (mine > ka || (mine == ka && lane > pa));
        kb = wmin_u64(behind ? mine : NOKEY);
    }
    return true;
}

template <int EPT>
struct Corr {                                       // one lane's correspondences, register resident
    double px[EPT], py[EPT], pz[EPT], qx[EPT], qy[EPT], qz[EPT];
    float nx[EPT], ny[EPT], nz[EPT];
    bool keep[EPT];
};

// ---- the 30 sums of an evaluation without the matrix pipe ------------------------------------------------------------
// On gfx950 v_mfma_f64_16x16x4_f64 holds the pipe for 64 cycles (FP64 matrix peak = FP64 vector peak), and the Gram form wastes
// half of every instruction on the two off-diagonal 8x8 blocks: 32 instructions = 2.3 k cycles per evaluation, plus the 64 bytes
// per correspondence staged through LDS to transpose rows into operands (fine trace: rows + LDS writes 2.3 k, Gram 2.3 k of an
// evaluation's 5.8 k).  Instead: every lane accumulates the 30 distinct sums (21 of J^T J, 6 of J^T r, sum r, sum r^2, n) of its own
// correspondences with plain FMAs (28 per correspondence), and the wave adds them up with a HALVING butterfly -- at distance 32 a
// lane pair splits the 30 values, each keeps 15 and receives the partner's share of those (v_permlane32_swap does the split, the
// exchange and leaves two registers to add), then 8, 4, 2, 1: 15 + 8 + 4 + 2 + 1 + 1 = 31 additions instead of 30 x 6, and
// lane l ends up owning ONE finished sum.  No rows in LDS, one barrier.
__device__ __forceinline__ double swap_add32(double lo, double hi)
{
    // lanes 0..31 get lo(l) + lo(l + 32), lanes 32..63 get hi(l) + hi(l - 32)
    const unsigned l0 = (unsigned)__double2loint(lo), l1 = (unsigned)__double2hiint(lo);
    const unsigned h0 = (unsigned)__double2loint(hi), h1 = (unsigned)__double2hiint(hi);
    const v2u_t a = __builtin_amdgcn_permlane32_swap(l0, h0, false, false), b = __builtin_amdgcn_permlane32_swap(l1, h1, false, false);
    return __hiloint2double((int)b.x, (int)a.x) + __hiloint2double((int)b.y, (int)a.y);
}
__device__ __forceinline__ double swap_add16(double lo, double hi)
{
    // lanes with (l & 16) == 0 get lo(l) + lo(l + 16), the others hi(l) + hi(l - 16)
    const unsigned l0 = (unsigned)__double2loint(lo), l1 = (unsigned)__double2hiint(lo);
    const unsigned h0 = (unsigned)__double2loint(hi), h1 = (unsigned)__double2hiint(hi);
    const v2u_t a = __builtin_amdgcn_permlane16_swap(l0, h0, false, false), b = __builtin_amdgcn_permlane16_swap(l1, h1, false, false);
    return __hiloint2double((int)b.x, (int)a.x) + __hiloint2double((int)b.y, (int)a.y);
}
template <int J>
__device__ __forceinline__ double split_add(double lo, double hi)
{
    // lanes with (l & J) == 0 get lo(l) + lo(l ^ J), the others hi(l) + hi(l ^ J)
    if constexpr (J == 32) return swap_add32(lo, hi);
    else if constexpr (J == 16) return swap_add16(lo, hi);
    else {
        const bool up = (threadIdx.x & J) != 0;
        const double keep = up ? hi : lo, send = up ? lo : hi;
        return keep + lane_xor_f64<J>(send);
    }
}
// g[0..N) -> g[0..(N+1)/2): lanes with (l & J) == 0 keep the first half, the others the second (a missing last entry is 0)
template <int N, int J>
__device__ __forceinline__ void halve(double (&g)[30])
{
    constexpr int H = (N + 1) / 2;
#pragma unroll
    for (int i = 0; i < H; ++i) g[i] = split_add<J>(g[i], i + H < N ? g[i + H] : 0.0);
}
// index (0..29) of the sum lane l owns after halve<30,32>, <15,16>, <8,8>, <4,4>, <2,2>, <1,1>; -1: none
__device__ __forceinline__ int owned_sum(int lane)
{
    if (lane & 1) return -1;
    const int low = ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
    if ((lane & 16) && low == 7) return -1;                 // the second half of 15 has only 7 entries
    return ((lane >> 5) & 1) * 15 + ((lane >> 4) & 1) * 8 + low;
}
// position of G[u][v] in the list of 30: upper triangle of J^T J row by row (21), J^T r (6), sum r, sum r^2, n; -1: not kept
// (G[k][7] = sum a_k, which nothing in this kernel reads)
__device__ __forceinline__ int sum_index(int u, int v)
{
    if (u > v) { const int t = u; u = v; v = t; }
    if (v < 6) return u * 6 - u * (u - 1) / 2 + (v - u);
    if (v == 6) return u < 6 ? 21 + u : 28;
    return u == 6 ? 27 : (u == 7 ? 29 : -1);
}

// Normal equations of the unweighted residuals at x over the kept correspondences, as the 8x8 matrix G of the rows
// [a0..a5 | r | 1] (J^T J = G[0..5][0..5], J^T r = G[.][6], sum r = G[6][7], sum r^2 = G[6][6], n = G[7][7]; G[k][7] is not
// formed), left in this wave's LDS slot S.gf[wave][slot].  `parity` alternates between consecutive evaluations (two buffers for
// the waves' sums: a wave can be at most one evaluation ahead of another).
template <int EPT>
__device__ __forceinline__ void eval_ne(TailShared &S, const double (&x)[6], const double (&sc)[6], const Corr<EPT> &C,
                                        double (&rr)[EPT], int slot, int parity)
{
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    SICP_ET(0)
    double H[12];
    euler_H(x, sc, H);
    const double s1 = sc[0], c1 = sc[1], s2 = sc[2], c2 = sc[3];
    const double w3y = -s1 * c2, w3z = c1 * c2;
    double g[30];
#pragma unroll
    for (int i = 0; i < 30; ++i) g[i] = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        double a[7] = {0, 0, 0, 0, 0, 0, 0};
        if (C.keep[e]) {
            double X, Y, Z;
            xfm(H, C.px[e], C.py[e], C.pz[e], X, Y, Z);
            a[6] = pdist(X - C.qx[e], Y - C.qy[e], Z - C.qz[e], C.nx[e], C.ny[e], C.nz[e]);
            const double nx = C.nx[e], ny = C.ny[e], nz = C.nz[e];
            const double ux = X - x[3], uy = Y - x[4], uz = Z - x[5];          // R p
            const double cx = uy * nz - uz * ny, cy = uz * nx - ux * nz, cz = ux * ny - uy * nx;   // (R p) x n
            a[0] = cx;
            a[1] = c1 * cy + s1 * cz;
            a[2] = s2 * cx + w3y * cy + w3z * cz;
            a[3] = nx; a[4] = ny; a[5] = nz;
        }
        rr[e] = a[6];
        int t = 0;
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
            for (int v = u; v < 6; ++v) { g[t] = fma(a[u], a[v], g[t]); ++t; }
#pragma unroll
        for (int u = 0; u < 6; ++u) g[21 + u] = fma(a[u], a[6], g[21 + u]);
        g[27] += a[6];
        g[28] = fma(a[6], a[6], g[28]);
        g[29] += C.keep[e] ? 1.0 : 0.0;
    }
    SICP_ET(1)
    halve<30, 32>(g); halve<15, 16>(g); halve<8, 8>(g); halve<4, 4>(g); halve<2, 2>(g); halve<1, 1>(g);
    SICP_ET(2)
    const int mine = owned_sum(lane);
    if (mine >= 0) S.gw[parity][wid][mine] = g[0];
    SICP_ET(3)
    __syncthreads();
    SICP_ET(4)
    // every wave adds the four waves' sums itself (fixed order) and parks the matrix in its own LDS slot
    const int idx = sum_index(lane >> 3, lane & 7);
    double v = 0.0;
    if (idx >= 0) {
        v = (S.gw[parity][0][idx] + S.gw[parity][1][idx]) + (S.gw[parity][2][idx] + S.gw[parity][3][idx]);
        if constexpr (NW == 8) v += (S.gw[parity][4][idx] + S.gw[parity][5][idx]) + (S.gw[parity][6][idx] + S.gw[parity][7][idx]);
    }
    S.gf[wid][slot][lane] = v;
    SICP_ET(5)
}

// Normal equations of the unweighted residuals at x over the kept correspondences as the 8x8 Gram matrix G of the
// rows [a0..a5 | r | 1] (J^T J = G[0..5][0..5], J^T r = G[.][6], sum r = G[6][7], sum r^2 = G[6][6], n = G[7][7]),
// left in this wave's LDS slot S.gf[wave][slot] -- the sums are uniform, registers are not spent on them.
// Jacobian of r = n.(R p + t - p1) w.r.t. the Euler angles through the instantaneous axes
// w1 = e_x, w2 = Rx e_y = (0, c1, s1), w3 = Rx Ry e_z = (s2, -s1 c2, c1 c2)  (R = Rx Ry Rz, mathutils.py:39-68):
// a_k = w_k . ((R p) x n).
template <int EPT>
__device__ __forceinline__ void eval_ne_mfma(TailShared &S, const double (&x)[6], const double (&sc)[6], const Corr<EPT> &C,
                                             double (&rr)[EPT], int slot)
{
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    SICP_ET(0)
    double H[12];
    euler_H(x, sc, H);
    const double s1 = sc[0], c1 = sc[1], s2 = sc[2], c2 = sc[3];
    const double w3y = -s1 * c2, w3z = c1 * c2;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (C.keep[e]) {
            double X, Y, Z;
            xfm(H, C.px[e], C.py[e], C.pz[e], X, Y, Z);
            a[6] = pdist(X - C.qx[e], Y - C.qy[e], Z - C.qz[e], C.nx[e], C.ny[e], C.nz[e]);
            const double nx = C.nx[e], ny = C.ny[e], nz = C.nz[e];
            const double ux = X - x[3], uy = Y - x[4], uz = Z - x[5];          // R p
            const double cx = uy * nz - uz * ny, cy = uz * nx - ux * nz, cz = ux * ny - uy * nx;   // (R p) x n
            a[0] = cx;
            a[1] = c1 * cy + s1 * cz;
            a[2] = s2 * cx + w3y * cy + w3z * cz;
            a[3] = nx; a[4] = ny; a[5] = nz;
            a[7] = 1.0;
        }
        rr[e] = a[6];
        double2 *row = reinterpret_cast<double2 *>(&S.ja[tid + e * TB][0]);
        row[0] = make_double2(a[0], a[1]); row[1] = make_double2(a[2], a[3]);
        row[2] = make_double2(a[4], a[5]); row[3] = make_double2(a[6], a[7]);
    }
    SICP_ET(1)
    __syncthreads();
    SICP_ET(2)
    // Gram product on the FP64 matrix pipe.  v_mfma_f64_16x16x4_f64: A[m = lane&15][k = lane>>4], B[k][n = lane&15],
    // one f64 per lane each.  Lane l supplies component (l & 7) of correspondence  base + 8 j + 4 ((l >> 3) & 1) + (l >> 4)
    // as BOTH operands: rows / columns 0..7 of D accumulate the Gram block of four correspondences, rows / columns
    // 8..15 that of four more; the two off-diagonal 8x8 blocks are garbage and ignored.  Two chains: the pipe overlaps them.
    v4d acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
    {
        const double *src = &S.ja[wid * (EPT * 64) + 4 * ((lane >> 3) & 1) + (lane >> 4)][lane & 7];
#pragma unroll
        for (int j = 0; j < EPT * 4; ++j) {                  // (reading the whole batch first was measured SLOWER: v_mfma_f64_16x16x4 holds
            const double v = src[0], u = src[64];            //  the pipe for 64 cycles on gfx950, the interleaved LDS reads hide behind it)
            src += 128;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(u, u, acc2, 0, 0, 0);
        }
        acc += acc2;
    }
    SICP_ET(3)
    // D: col = lane & 15, row = (lane >> 4) + 4 * reg
    {
        const int n = lane & 15;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int m = (lane >> 4) + 4 * rg;
            if ((m >> 3) == (n >> 3)) S.gp[wid][m >> 3][(m & 7) * 8 + (n & 7)] = acc[rg];
        }
    }
    __syncthreads();
    SICP_ET(4)
    // every wave folds the partial blocks itself and parks the result in its own LDS slot (a wave's LDS
    // operations are ordered: no barrier between this write and the reads that follow)
    double g = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) g += S.gp[w][0][lane] + S.gp[w][1][lane];
    S.gf[wid][slot][lane] = g;
    SICP_ET(5)
}

// wave 0: the words parked in S.out leave in ONE store instruction.  The record goes to pinned host memory and the host polls its
// ticket word instead of waiting for the end-of-kernel signal: record words as system-scope write-through stores, drained
// (vmcnt(0)), then the ticket as one more such store.  No fence: the host reads nothing else this kernel wrote.  (Round 5 issued
// __threadfence_system() AND a release store; measured side by side in round 6, profiles/r6: the same 13.0 us per launch -- the
// kernel's end writes the L2 back anyway -- so the simpler form stays.)
__device__ __forceinline__ void flush_state(TailShared &S, double *dst, int count)
{
    const int lane = threadIdx.x & 63;
    if (lane < count) dst[lane] = S.out2[lane];
}
__device__ __forceinline__ void flush_rec(TailShared &S, double *rec, int count)
{
    const int lane = threadIdx.x;
    if (lane < count) __hip_atomic_store(rec + lane, S.out[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void publish(double *rec, double seq)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
        __hip_atomic_store(rec + REC_TICKET, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace

template <int EPT>
// (parameter order: the eight pointers the first instructions need come first -- they are preloaded into SGPRs when the wave
// starts, -amdgpu-kernarg-preload-count=16, instead of being fetched by a load the whole kernel would wait behind)
__global__ __launch_bounds__(TB, 1) void k_icp_tail(
    IcpDev *__restrict__ st,
    const double *__restrict__ dist /* point-to-plane distances and planarity verdicts of this iteration's matches: left by */,
    const uint8_t *__restrict__ flag /* the match kernel's winning lanes (or by k_postmatch behind a multi-GPU exchange)   */,
    const double *__restrict__ p2, const double *__restrict__ qx, const double *__restrict__ qy, const double *__restrict__ qz,
    const float *__restrict__ normals, uint8_t *__restrict__ keep, double *__restrict__ resid, double *__restrict__ rec, TailArgs A)
{
    __shared__ TailShared S;
    const int tid = threadIdx.x, wid = tid >> 6;
    const int Q = A.Q;
    long long tk[6]; tk[0] = clock64();
    // ---- loop state + this lane's correspondences: ONE global round trip (every load is issued before the first
    //      barrier and before the stop flag is looked at).  The rejection needs only the distances and verdicts -- 9 bytes per
    //      correspondence, asked for FIRST (loads return in order); the 60 bytes per correspondence the solver works on arrive
    //      while the order statistics are being taken ----
    double d[EPT];
    uint8_t fb[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * TB;
        const int ic = i < Q ? i : Q - 1;                 // clamped: every lane loads, lanes past Q are masked below
        d[e] = dist[ic]; fb[e] = flag[ic];
    }
    double x[6], sc[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) { x[j] = st->x[j]; sc[j] = st->sc[j]; }
    const double w_state = st->w, prev_mean = st->prev_mean, prev_std = st->prev_std;
    const int done_iters = st->done_iters;
    const int stop = st->stop;
    const double pmed = st->sel_med, pmad = st->sel_mad;               // the last launch's median and MAD (0: none) ...
    const int pcnt = st->sel_m;                                        // ... of this many distances
    Corr<EPT> C;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * TB;
        const int ic = i < Q ? i : Q - 1;
        C.px[e] = p2[3 * ic]; C.py[e] = p2[3 * ic + 1]; C.pz[e] = p2[3 * ic + 2];
        C.qx[e] = qx[ic]; C.qy[e] = qy[ic]; C.qz[e] = qz[ic];
        C.nx[e] = normals[3 * ic]; C.ny[e] = normals[3 * ic + 1]; C.nz[e] = normals[3 * ic + 2];
    }
    unsigned *hc = reinterpret_cast<unsigned *>(&S.ja[0][0]);      // the row staging area is idle until the LM phase
    for (int i = tid; i < HC * 257; i += TB) hc[i] = 0u;
    if (tid == 0) S.ncand = 0u;
    if (tid < 64) S.out[tid] = 0.0;
    if (stop) {
        // the run ended in an earlier launch (converged / failed): nothing to do but tell the host
        if (tid == 0) __hip_atomic_store(rec + REC_STATUS, 3.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (wid == 0) publish(rec, A.seq);
        return;
    }

    bool fl[EPT];
    unsigned long long key[EPT];
    unsigned nflag = 0;
    double dmn = __builtin_inf(), dmx = -__builtin_inf();
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * TB;
        C.keep[e] = false; key[e] = NOKEY;
        fl[e] = i < Q && fb[e] != 0;
        if (i >= Q) d[e] = 0.0;
        if (fl[e]) { key[e] = okey(d[e]); dmn = fmin(dmn, d[e]); dmx = fmax(dmx, d[e]); }
        nflag += (unsigned)__popcll((long long)__ballot(fl[e]));
    }
    // a settled run: both statistics are looked for in a window around the last launch's first (window_collect).  Half widths in
    // units of the MAD, sized for ~40 of the `pcnt` keys (a normal density holds 0.27 n keys per MAD at its median, 0.43 n of the
    // absolute deviations at theirs)
    bool win = A.window != 0 && pmad > 0.0 && pmad < __builtin_inf() && pcnt > 0;
    const double hw_med = pmad * fmin(0.25, 75.0 / (double)pcnt), hw_mad = pmad * fmin(0.25, 47.0 / (double)pcnt);
    if (win) window_collect<EPT>(S, 0, key, okey(pmed - hw_med), okey(pmed + hw_med));
    // survivors of the planarity test: wave counts + one barrier.  The RANGE of their distances (two wave reductions, the histogram
    // selection's first interval) is only formed when that selection runs: a settled run reads both statistics off its windows
    bool have_range = !win;
    if (have_range) { dmn = wmin_f64(dmn); dmx = wmin_f64(-dmx); }
    if ((tid & 63) == 0) { S.wcnt[wid] = nflag; if (have_range) { S.dmm[wid][0] = dmn; S.dmm[wid][1] = dmx; } }
    __syncthreads();
    long m = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) m += S.wcnt[w];
    if (have_range) {
#pragma unroll
        for (int w = 0; w < NW; ++w) { dmn = fmin(dmn, S.dmm[w][0]); dmx = fmin(dmx, S.dmm[w][1]); }
    }
    tk[1] = clock64();

    if (m == 0) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) { const int i = tid + e * TB; if (i < Q) { keep[i] = 0; resid[i] = 0.0; } }
        if (tid == 0) {
            S.out[1] = __builtin_nan(""); S.out[2] = __builtin_nan("");
            for (int k = 0; k < 6; ++k) S.out[10 + k] = x[k];
            S.out[REC_STATUS] = 1.0;
            st->stop = 1;
        }
        if (wid == 0) { flush_rec(S, rec, REC_TICKET); publish(rec, A.seq); }
        return;
    }

    // ---- median / raw MAD (corrpts.py:165-188): np.median = mean of the two middle values ----
    double med = 0.0, mad = 0.0;
    int rounds[2] = {0, 0};
    long long tsel = 0;
#pragma unroll 1
    for (int which = 0; which < 2; ++which) {
        if (which == 1) {
#pragma unroll
            for (int e = 0; e < EPT; ++e) if (fl[e]) key[e] = okey(fabs(d[e] - med));
            tsel = clock64();
            if (win) {
                window_collect<EPT>(S, 1, key, okey(fmax(pmad - hw_mad, 0.0)), okey(pmad + hw_mad));
                __syncthreads();
            }
        }
        unsigned long long ka, kb;
        int nr = 0;
#ifdef SICP_SEL_FINE_TRACE
        if (tid == 0) S.selw = which;
        __syncthreads();
#endif
        // (one miss ends the attempts of this launch: a median that moved takes the MAD with it)
        if (win) win = window_pick(S, which, (m - 1) / 2, (m & 1) == 0, ka, kb);
        if (!win) {
            if (!have_range) {
                // the window missed: the distances' range after all (dmn / dmx still hold this lane's own: d is untouched)
                dmn = wmin_f64(dmn); dmx = wmin_f64(-dmx);
                if ((tid & 63) == 0) { S.dmm[wid][0] = dmn; S.dmm[wid][1] = dmx; }
                __syncthreads();
#pragma unroll
                for (int w = 0; w < NW; ++w) { dmn = fmin(dmn, S.dmm[w][0]); dmx = fmin(dmx, S.dmm[w][1]); }
                have_range = true;
            }
            // the interval that holds every key: the distances' own, or (|d - med| is monotone in d on either side of med) what follows from it
            unsigned long long klo = okey(dmn), khi = okey(-dmx);
            if (which == 1) { const double u = fabs(dmn - med), v = fabs(-dmx - med); klo = okey(0.0); khi = okey(u > v ? u : v); }
            block_select<EPT>(S, hc, key, (m - 1) / 2, (m & 1) == 0, klo, khi, ka, kb, nr);
        }
        const double mid = (oval(ka) + oval(kb)) / 2.0;
        if (which == 0) { med = mid; rounds[0] = nr; } else { mad = mid; rounds[1] = nr; }
    }
    const double bound = 3 * mad;
    tk[2] = clock64();

    // ---- keep mask.  The kept distances' count / mean / std (simpleicp.py:233-234) are only needed NOW when the
    //      weight is still automatic (first iteration of such a run): then one pass over deviations from the median
    //      (a shift within a few MAD of the mean: var = (S2 - S1^2 / n) / n loses nothing to cancellation).  Otherwise
    //      they come for free out of the first evaluation's Gram matrix below (r = d at the start estimate). ----
    const bool need_w = !(w_state > 0);
    double v3[3] = {0.0, 0.0, 0.0};
    unsigned nkeep = 0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * TB;
        const double dev = d[e] - med;
        const bool kq = fl[e] && fabs(dev) <= bound;
        C.keep[e] = kq;
        if (i < Q) keep[i] = kq ? 1 : 0;
        if (kq) { v3[0] += 1.0; v3[1] += dev; v3[2] = fma(dev, dev, v3[2]); }
        nkeep += (unsigned)__popcll((long long)__ballot(kq));
    }
    double nk, dmean = 0.0, dstd = 0.0;
    if (need_w) {
        block_sum<3>(v3, S.red[0]);
        nk = v3[0]; dmean = med + v3[1] / nk;
        const double dvar = (v3[2] - v3[1] * v3[1] / nk) / nk;
        dstd = sqrt(dvar > 0.0 ? dvar : 0.0);
    } else {
        if ((tid & 63) == 0) S.wcnt[wid] = nkeep;        // (its earlier content was consumed two barriers ago)
        __syncthreads();
        unsigned nkw = (S.wcnt[0] + S.wcnt[1]) + (S.wcnt[2] + S.wcnt[3]);
        if constexpr (NW == 8) nkw += (S.wcnt[4] + S.wcnt[5]) + (S.wcnt[6] + S.wcnt[7]);
        nk = (double)nkw;
    }
    if (tid == 0) { S.out[0] = (double)m; S.out[1] = med; S.out[2] = mad; S.out[3] = nk; S.out[4] = dmean; S.out[5] = dstd; }
    if (nk < 6.0) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) { const int i = tid + e * TB; if (i < Q) resid[i] = 0.0; }
        if (tid == 0) {
            for (int k = 0; k < 6; ++k) S.out[10 + k] = x[k];
            S.out[REC_STATUS] = 1.0;
            st->stop = 1;
        }
        if (wid == 0) { flush_rec(S, rec, REC_TICKET); publish(rec, A.seq); }
        return;
    }
    const double w = need_w ? 1.0 / (dstd * dstd) : w_state;          // simpleicp.py:233-234 (frozen afterwards)
    tk[3] = clock64();

    // ---- Levenberg-Marquardt on the fused 6x6 reductions (same acceptance rules as the host solver);
    //      one evaluation site: the first evaluation is a trial that is always accepted ----
    int nfree = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) nfree += (A.ow[j] < __builtin_inf()) ? 1 : 0;
    double xn[6], scn[6], rr[EPT], rrn[EPT];
#pragma unroll
    for (int j = 0; j < 6; ++j) { xn[j] = x[j]; scn[j] = sc[j]; }
    int steps = 0, evals = 0, cur = 1, tries = 0;
    bool first = true;
    double d0n = 1.0, d0s1 = 0.0, d0s2 = 0.0;
    double cost = 0.0, lambda = 0.0, dxmax = 0.0;
    // finer split of the solver loop (-DSICP_TAIL_FINE_TRACE: each reading drains the LDS queue, ~100 cycles -- off by default;
    // measured at C4: evaluation 5.35 k cycles, acceptance 1.05 k, 6x6 solve 2.3 k, trial angles + loop 2.2 k per round)
#ifdef SICP_TAIL_FINE_TRACE
#define SICP_FT(x) x
#else
#define SICP_FT(x)
#endif
    long long t_eval = 0, t_step = 0, t_acc = 0;
#pragma unroll 1
    for (;;) {
        SICP_FT(const long long te0 = clock64();)
#ifdef SICP_TAIL_MFMA_GRAM
        eval_ne_mfma<EPT>(S, xn, scn, C, rrn, cur ^ 1); ++evals;      // (its first barrier orders it after the last one's LDS reads)
#else
        eval_ne<EPT>(S, xn, scn, C, rrn, cur ^ 1, evals & 1); ++evals;
#endif
        SICP_FT(const long long te1 = clock64(); t_eval += te1 - te0;)
        const double costn = objective(S.gf[wid][cur ^ 1], w, xn, A);
        if (first) {
            // r = d at the start estimate: sum r, sum r^2, n of this evaluation ARE the kept distances' statistics (turned into
            // mean / std where the record is written: two divisions and a root are not on the solver's path)
            const double *G0 = S.gf[wid][cur ^ 1];
            d0n = G0[7 * 8 + 7]; d0s1 = G0[6 * 8 + 7]; d0s2 = G0[6 * 8 + 6];
        }
        if (first || costn <= cost * (1 + 1e-12) || dxmax < 1e-15) {          // 1e-12: rounding noise of the sums
#pragma unroll
            for (int j = 0; j < 6; ++j) { x[j] = xn[j]; sc[j] = scn[j]; }
#pragma unroll
            for (int e = 0; e < EPT; ++e) rr[e] = rrn[e];
            cur ^= 1; cost = costn; tries = 0;
            if (!first) {
                lambda = lambda > 0 ? lambda * 0.1 : 0.0;
                if (lambda < 1e-12) lambda = 0.0;
                ++steps;
                double xmax = 0.0;
#pragma unroll
                for (int j = 0; j < 6; ++j) xmax = fmax(xmax, fabs(x[j]));
                if (dxmax <= 1e-13 * (1.0 + xmax)) break;
            }
            first = false;
            if (steps >= A.max_steps || nfree == 0) break;
        } else {
            lambda = lambda > 0 ? lambda * 10 : 1e-6;
            if (++tries >= 40) break;
        }
        // next trial from (x, lambda)
        bool ok = false;
        double dstep[6];
        SICP_FT(const long long ts0 = clock64(); t_acc += ts0 - te1;)
#pragma unroll 1
        for (; tries < 40; ++tries) {
            ok = lm_step(S.gf[wid][cur], w, x, lambda, A, dstep);
            dxmax = 0.0;
#pragma unroll
            for (int j = 0; j < 6; ++j) { xn[j] = x[j] + dstep[j]; dxmax = fmax(dxmax, fabs(dstep[j])); }
            ok = ok && (dxmax < __builtin_inf());
            if (ok) break;
            lambda = lambda > 0 ? lambda * 10 : 1e-6;
        }
        SICP_FT(t_step += clock64() - ts0;)
        if (!ok) break;
        double xm = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) xm = fmax(xm, fabs(x[j]));
        // the undamped Gauss-Newton step from x is below 1e-10: x is the minimiser to that accuracy (the reference stops at 1e-8)
        if (lambda == 0.0 && dxmax <= 1e-10 * (1.0 + xm)) break;
#pragma unroll
        for (int j = 0; j < 3; ++j) sincos_step(xn[j], dstep[j], sc[2 * j], sc[2 * j + 1], scn[2 * j], scn[2 * j + 1]);
    }
    tk[4] = clock64();

    // ---- residuals at the optimum (rr belongs to x: rejected trials only wrote rrn); their mean / std (ddof 0) come
    //      out of the accepted evaluation's Gram matrix: sum r, sum r^2, n (at the optimum |mean| is well below std, so
    //      sum r^2 / n - mean^2 keeps its digits) ----
    const double *G = S.gf[wid][cur];
    const double gn = G[7 * 8 + 7];
    const double rmean = G[6 * 8 + 7] / gn;
    const double rvar = G[6 * 8 + 6] / gn - rmean * rmean;
    const double rstd = sqrt(rvar > 0.0 ? rvar : 0.0);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * TB;
        if (i < Q) resid[i] = rr[e];
    }
    const bool finite = cost < __builtin_inf();
    // convergence test of simpleicp.py:356-379 on (mean, std) of this and the previous iteration's residuals
    bool conv = false;
    if (A.min_change >= 0.0 && done_iters > 0 && finite) {
        const double cm = prev_mean == 0.0 ? (rmean == 0.0 ? 0.0 : __builtin_inf()) : fabs((rmean - prev_mean) / prev_mean * 100.0);
        const double cs = prev_std == 0.0 ? (rstd == 0.0 ? 0.0 : __builtin_inf()) : fabs((rstd - prev_std) / prev_std * 100.0);
        conv = cm < A.min_change && cs < A.min_change;
    }
    if (wid > 1) return;
    if (wid == 1) {
        // ---- wave 1: the next iteration's start (estimate, its sin / cos, H(x) and the rigid inverse [R^T | -R^T t]) leaves as one
        //      store -- while wave 0 assembles and publishes the record (every lane of every wave holds the same estimate) ----
        if (tid == 64) {
            IcpDev n;
            double Hn[12];
            euler_H(x, sc, Hn);
#pragma unroll
            for (int j = 0; j < 6; ++j) { n.x[j] = x[j]; n.sc[j] = sc[j]; }
#pragma unroll
            for (int j = 0; j < 12; ++j) n.H.m[j] = Hn[j];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < 3; ++j) n.Hinv.m[4 * i + j] = Hn[4 * j + i];
                n.Hinv.m[4 * i + 3] = -(Hn[i] * Hn[3] + Hn[4 + i] * Hn[7] + Hn[8 + i] * Hn[11]);
            }
            n.w = w; n.prev_mean = rmean; n.prev_std = rstd;
            n.done_iters = done_iters + 1; n.stop = (conv || !finite) ? 1 : 0; n.sel_m = (int)m; n.pad = 0;
            n.sel_med = med; n.sel_mad = mad;
            const double *src = reinterpret_cast<const double *>(&n);
#pragma unroll
            for (int j = 0; j < ST_DOUBLES; ++j) S.out2[j] = src[j];
        }
        flush_state(S, reinterpret_cast<double *>(st), ST_DOUBLES);
        return;
    }
    // ---- wave 0: the record (pinned host memory) ----
    if (tid < 30) {
        // record layout of the 30 sums: 21 upper-triangle entries of J^T J (row-major), 6 of J^T r, sum r, sum r^2, n
        int u = 0, v = 0;
        if (tid < 21) { int t = tid; while (t >= 6 - u) { t -= 6 - u; ++u; } v = u + t; }
        else if (tid < 27) { u = tid - 21; v = 6; }
        else if (tid == 27) { u = 6; v = 7; }
        else if (tid == 28) { u = 6; v = 6; }
        else { u = 7; v = 7; }
        S.out[20 + tid] = G[u * 8 + v];
    }
    if (tid == 0) {
        if (!need_w) {
            const double mean0 = d0s1 / d0n, var0 = d0s2 / d0n - mean0 * mean0;
            S.out[4] = mean0; S.out[5] = sqrt(var0 > 0.0 ? var0 : 0.0);
        }
        S.out[6] = w; S.out[7] = cost; S.out[8] = steps; S.out[9] = evals;
#pragma unroll
        for (int j = 0; j < 6; ++j) S.out[10 + j] = x[j];
        S.out[16] = rmean; S.out[17] = rstd;
        S.out[REC_STATUS] = finite ? 0.0 : 2.0;
        S.out[REC_CONVERGED] = conv ? 1.0 : 0.0;
        tk[5] = clock64();
        for (int k = 0; k < 5; ++k) S.out[50 + k] = (double)(tk[k + 1] - tk[k]);
        S.out[59] = (double)t_eval; S.out[60] = (double)t_step; S.out[62] = (double)t_acc;
        S.out[55] = (double)(tsel - tk[1]); S.out[56] = rounds[0]; S.out[57] = (double)(tk[2] - tsel); S.out[58] = rounds[1];
#ifdef SICP_EVAL_FINE_TRACE
        for (int i = 0; i < 5; ++i) S.out[38 + i] = (double)(S.evt[i + 1] - S.evt[i]);
#endif
#ifdef SICP_SEL_FINE_TRACE
        // (trace build: the splits of both selections overwrite the last twelve normal-equation sums of the record)
        for (int wsel = 0; wsel < 2; ++wsel)
            for (int i = 0; i < 6; ++i) S.out[38 + 6 * wsel + i] = (double)(S.selt[wsel][i + 1] - S.selt[wsel][i]);
#endif

    }
    flush_rec(S, rec, REC_TICKET);
    publish(rec, A.seq);
}

void launch_icp_tail(hipStream_t s, const double *qx, const double *qy, const double *qz, const float *normals,
                     const double *p2, const TailArgs &A, IcpDev *st, const double *dist, const uint8_t *flag,
                     uint8_t *keep, double *resid, double *rec)
{
    // correspondences per lane: the register-resident copy is sized to the problem
    if (A.Q <= TB)
        hipLaunchKernelGGL(k_icp_tail<1>, dim3(1), dim3(TB), 0, s, st, dist, flag, p2, qx, qy, qz, normals, keep, resid, rec, A);
    else if (A.Q <= 2 * TB)
        hipLaunchKernelGGL(k_icp_tail<2>, dim3(1), dim3(TB), 0, s, st, dist, flag, p2, qx, qy, qz, normals, keep, resid, rec, A);
    else if (A.Q <= 4 * TB)
        hipLaunchKernelGGL(k_icp_tail<4>, dim3(1), dim3(TB), 0, s, st, dist, flag, p2, qx, qy, qz, normals, keep, resid, rec, A);
    else
        hipLaunchKernelGGL(k_icp_tail<8>, dim3(1), dim3(TB), 0, s, st, dist, flag, p2, qx, qy, qz, normals, keep, resid, rec, A);
}

}  // namespace sicp
