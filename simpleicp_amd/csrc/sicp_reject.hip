// sicp_reject.hip -- median / raw-MAD rejection of one ICP iteration by ONE workgroup whose keys live in REGISTERS
// (corrpts.py:165-188: np.median = mean of the two middle values; scipy median_abs_deviation with scale 1.0; keep |d - median| <=
// 3 MAD), the keep mask and the kept distances' count / mean / std (simpleicp.py:233-234) -- for up to REJECT_MAX_Q (16 384)
// correspondences: the step between the single-launch tail (sicp_tail.hip, Q <= 2048) and the many-workgroup selection
// (k_hsel_all, sicp_grid.hip), and the operator CorrPts.reject_wrt_point_to_plane_distances at any such size.
//
// Round 5's kernel (k_reject, keys staged in 128 KB of LDS, 16 privatised histogram copies) took 15 us at 2 049 correspondences and
// 40 us at 16 384 -- 47 % of that iteration.  What it paid for: (i) same-address LDS atomics -- real distances share their leading
// bits, so the first histogram round puts half the keys into two or three bins, and with 16 copies shared by 16 waves four lanes of
// every instruction met in one word; (ii) three to four histogram rounds per statistic down to <= 8 keys; (iii) every pass reading
// its keys back from LDS.  Here:
//   * 1 024 lanes hold up to 16 order-preserving keys each in registers: a pass over the keys is arithmetic, LDS is free for
//   * 32 privatised histogram copies (at most two lanes of an instruction can meet), folded by all 1 024 lanes;
//   * binning stops as soon as the wanted rank's bin holds <= 64 keys (usually after ONE round): those are listed per wave by a
//     prefix sum (no atomics) and the wanted rank is read off them by quickselect with wave-uniform bookkeeping (ballot counts);
//   * in a chained run both statistics are first looked for in a WINDOW around the last launch's values (the caller says when the
//     output buffer still holds them): once the estimate has settled that is the whole selection -- one pass, one barrier, one
//     quickselect per statistic.  A window that misses costs one barrier; the result is the exact order statistic either way
//     (same keys, same total order), so the bits do not depend on which road was taken.
// What is left is THROUGHPUT: at 10 000 correspondences every element-wise pass (keys, lists, keep mask, sums) costs ~10 k cycles -- a
// CU issues 64 lanes per clock -- 46 k in all with the windows, 70 k without (scripts/ubench/reject_trace.hip, profiles/r6).  Measured
// and rejected: the same work spread over ceil(Q / 1024) workgroups meeting at a flat grid barrier once per statistic (once per
// histogram round without a window): a barrier is ~5 us whatever the number of blocks (six dependent device-scope round trips),
// 39 us per iteration at 10 000 against 28 here (profiles/r6/q_sweep_few_workgroup_rejection_rejected.txt).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sicp_internal.h"
#include "sicp_lanes.h"

namespace sicp {

namespace {

constexpr int RB = 1024, RW = RB / 64;       // lanes, waves
constexpr int RHC = 32;                      // privatised histogram copies (rows padded to 257 words)
constexpr int RCAP = 64, RSEG = 64;          // keys a candidate list may hold in all / per wave (one wave may hold them all)
constexpr unsigned long long RNOKEY = ~0ull;

__device__ __forceinline__ unsigned long long rkey(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double rval(unsigned long long k)
{
    const unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}
__device__ __forceinline__ unsigned long long rmin_u64(unsigned long long v)
{
    unsigned long long o;
    o = lane_xor64<32>(v); v = o < v ? o : v;  o = lane_xor64<16>(v); v = o < v ? o : v;
    o = lane_xor64<8>(v);  v = o < v ? o : v;  o = lane_xor64<4>(v);  v = o < v ? o : v;
    o = lane_xor64<2>(v);  v = o < v ? o : v;  o = lane_xor64<1>(v);  v = o < v ? o : v;
    return v;
}
__device__ __forceinline__ double rmin_f64(double v)
{
    v = fmin(v, lane_xor_f64<32>(v)); v = fmin(v, lane_xor_f64<16>(v)); v = fmin(v, lane_xor_f64<8>(v));
    v = fmin(v, lane_xor_f64<4>(v));  v = fmin(v, lane_xor_f64<2>(v));  v = fmin(v, lane_xor_f64<1>(v));
    return v;
}
__device__ __forceinline__ unsigned long long rlane_u64(unsigned long long v, int l)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

struct RejShared {
    unsigned hc[RHC * 257];                  // histogram copies (33 KB)
    alignas(16) unsigned part[4][256];       // quarter folds of a round
    unsigned long long wc[2][RW][RSEG];      // per-wave candidate lists, two buffers (a list's readers and the next list's writers
    alignas(16) unsigned wci[2][RW];         //  are at most one barrier apart), how many per wave,
    alignas(16) unsigned wbl[2][RW];         //  and how many member keys lie below the listed interval
    unsigned long long wmin[2][RW];
    double red[RW][4];
    double dmm[RW][2];
    unsigned wcnt[RW];
};

// the member keys inside [wlo, whi] -> this wave's list, those below counted (all lanes call it; a barrier must follow)
// (K(e): element e's key -- the stored key of its distance, or the key of its absolute deviation formed from that on the fly)
template <int EPT, class KeyOf>
__device__ __forceinline__ void list_interval(RejShared &S, int buf, KeyOf K, unsigned long long wlo, unsigned long long whi)
{
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned cin = 0, cbel = 0, inm = 0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        // (RNOKEY is above every interval the callers ask for: no separate membership test)
        const unsigned long long k = K(e);
        const bool inw = k >= wlo && k <= whi;
        inm |= inw ? (1u << e) : 0u;
        cin += inw ? 1u : 0u;
        cbel += k < wlo ? 1u : 0u;
    }
    const unsigned incl = wscan_u32(cin | (cbel << 16));           // one wave prefix sum for both counts (<= 1 024 each)
    unsigned slot = (incl & 0xffffu) - cin;
#pragma unroll
    for (int e = 0; e < EPT; ++e)
        if ((inm >> e) & 1u) { if (slot < (unsigned)RSEG) S.wc[buf][wid][slot] = K(e); ++slot; }
    if (lane == 63) { S.wci[buf][wid] = incl & 0xffffu; S.wbl[buf][wid] = incl >> 16; }
}

// After the barrier: the keys of rank r and (want2) r + 1 among ALL member keys, read off the listed interval.
//   false: rank r is not inside the list (or the list overflowed) -- nothing returned;
//   above: want2 and rank r + 1 is the smallest member key ABOVE the interval -- ka is set, the caller finds kb.
__device__ __forceinline__ bool pick_listed(const RejShared &S, int buf, long r, bool want2, unsigned long long &ka,
                                            unsigned long long &kb, bool &above)
{
    const int lane = threadIdx.x & 63;
    unsigned n = 0, seg_max = 0, my_w = 0, my_base = 0;
    long below = 0;
#pragma unroll
    for (int w4 = 0; w4 < RW; w4 += 4) {
        const uint4 c4 = *reinterpret_cast<const uint4 *>(&S.wci[buf][w4]), b4 = *reinterpret_cast<const uint4 *>(&S.wbl[buf][w4]);
        const unsigned cc[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if ((unsigned)lane >= n && (unsigned)lane < n + cc[j]) { my_w = (unsigned)(w4 + j); my_base = n; }
            n += cc[j]; seg_max = cc[j] > seg_max ? cc[j] : seg_max;
        }
        below += (long)((b4.x + b4.y) + (b4.z + b4.w));
    }
    long t = r - below;
    above = false;
    if (n > (unsigned)RCAP || seg_max > (unsigned)RSEG || t < 0 || t >= (long)n) return false;
    unsigned long long mine = RNOKEY;
    if ((unsigned)lane < n) mine = S.wc[buf][my_w][(unsigned)lane - my_base];
    above = want2 && t + 1 >= (long)n;
    // quickselect on the total order (key, lane): the pivot is the first lane still in play, one ballot counts the keys before it
    unsigned long long active = n >= 64u ? ~0ull : ((1ull << n) - 1ull);
    int pa = 0;
#pragma unroll 1
    for (;;) {
        const int p = __ffsll((long long)active) - 1;
        const unsigned long long pv = rlane_u64(mine, p);
        const unsigned long long less = __ballot(mine < pv || (mine == pv && lane < p)) & active;
        const long c = (long)__popcll((long long)less);
        if (c == t) { ka = pv; pa = p; break; }
        if (c > t) active = less;
        else { active &= ~less & ~(1ull << p); t -= c + 1; }
    }
    kb = ka;
    if (want2 && !above) {
        const bool behind = (unsigned)lane < n && (mine > ka || (mine == ka && lane > pa));
        kb = rmin_u64(behind ? mine : RNOKEY);
    }
    return true;
}

// the smallest member key above `hi` (every lane gets it; one barrier inside)
template <int EPT, class KeyOf>
__device__ __forceinline__ unsigned long long min_above(RejShared &S, int buf, KeyOf K, unsigned long long hi)
{
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long nx = RNOKEY;
#pragma unroll
    for (int e = 0; e < EPT; ++e) { const unsigned long long k = K(e); if (k != RNOKEY && k > hi) nx = k < nx ? k : nx; }
    nx = rmin_u64(nx);
    if (lane == 0) S.wmin[buf][wid] = nx;
    __syncthreads();
    unsigned long long b = S.wmin[buf][0];
#pragma unroll
    for (int w = 1; w < RW; ++w) { const unsigned long long a = S.wmin[buf][w]; b = a < b ? a : b; }
    return b;
}

// Exact order statistics of the block's member keys (RNOKEY = not a member; m >= 1 of them): ka = key of rank r (0-based), kb = key
// of rank r + 1 when want2 (else ka).
//   wlo <= whi : an interval to try FIRST (the caller's guess of where the rank lies); wlo > whi: none;
//   [lo, hi]   : an interval that contains every member key -- narrowed by 256-bin histogram rounds (integer arithmetic on the
//                order-preserving keys: monotone, exact) until the wanted rank's bin holds <= RCAP keys or is a single key value.
// S.hc is zero on entry and on exit; `buf` alternates between the two statistics of a launch.
template <int EPT, class KeyOf>
__device__ __forceinline__ void select_rank(RejShared &S, int buf, KeyOf K, long r, bool want2,
                                            unsigned long long wlo, unsigned long long whi, bool listed /* list_interval(buf, wlo, whi) has
                                            been called and a barrier passed */, unsigned long long lo, unsigned long long hi,
                                            unsigned long long &ka, unsigned long long &kb)
{
    const int tid = threadIdx.x, lane = tid & 63;
    bool above = false;
    if (wlo <= whi) {
        if (!listed) { list_interval<EPT>(S, buf, K, wlo, whi); __syncthreads(); }
        if (pick_listed(S, buf, r, want2, ka, kb, above)) {
            if (above) kb = min_above<EPT>(S, buf, K, whi);
            return;
        }
        __syncthreads();                      // (the list is read: the general road below writes the same buffer)
    }
    unsigned long long below = 0;
    unsigned cs = 0;
    int sh = 0;
    unsigned *mycopy = S.hc + (lane & (RHC - 1)) * 257;
#pragma unroll 1
    for (int round = 0; round < 10; ++round) {
        const unsigned long long range = hi - lo;
        sh = range < 256ull ? 0 : (64 - __clzll((long long)range)) - 8;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const unsigned long long k = K(e);
            if (k != RNOKEY && k >= lo && k <= hi) atomicAdd(&mycopy[(unsigned)((k - lo) >> sh)], 1u);
        }
        __syncthreads();
        {   // lane t folds bin (t & 255) over a quarter of the copies and leaves them clean
            const int bin = tid & 255, q4 = tid >> 8;
            unsigned tot = 0;
#pragma unroll
            for (int c = 0; c < RHC / 4; ++c) { unsigned *w = &S.hc[(q4 * (RHC / 4) + c) * 257 + bin]; tot += *w; *w = 0u; }
            S.part[q4][bin] = tot;
        }
        __syncthreads();
        // every wave scans the 256 totals on its own (lane l owns bins 4l..4l+3)
        uint4 h4 = *reinterpret_cast<const uint4 *>(&S.part[0][4 * lane]);
#pragma unroll
        for (int q4 = 1; q4 < 4; ++q4) {
            const uint4 o = *reinterpret_cast<const uint4 *>(&S.part[q4][4 * lane]);
            h4.x += o.x; h4.y += o.y; h4.z += o.z; h4.w += o.w;
        }
        const unsigned mine = h4.x + h4.y + h4.z + h4.w;
        const unsigned incl = wscan_u32(mine);
        const unsigned long long t = (unsigned long long)r - below;
        const unsigned long long gt = __ballot((unsigned long long)incl > t);
        const int L = __ffsll((long long)gt) - 1;
        const unsigned eL = (unsigned)__builtin_amdgcn_readlane((int)(incl - mine), L);
        const unsigned a0 = (unsigned)__builtin_amdgcn_readlane((int)h4.x, L), a1 = (unsigned)__builtin_amdgcn_readlane((int)h4.y, L);
        const unsigned a2 = (unsigned)__builtin_amdgcn_readlane((int)h4.z, L), a3 = (unsigned)__builtin_amdgcn_readlane((int)h4.w, L);
        unsigned acc = eL; int j = 0; cs = a0;
        if (t >= (unsigned long long)acc + a0) { acc += a0; j = 1; cs = a1;
            if (t >= (unsigned long long)acc + a1) { acc += a1; j = 2; cs = a2;
                if (t >= (unsigned long long)acc + a2) { acc += a2; j = 3; cs = a3; } } }
        below += acc;
        lo = lo + ((unsigned long long)(4u * (unsigned)L + (unsigned)j) << sh);
        if (sh > 0) { const unsigned long long top = lo + ((1ull << sh) - 1ull); hi = top < hi ? top : hi; } else hi = lo;
        if (cs <= (unsigned)RCAP || sh == 0) break;
    }
    const unsigned long long t = (unsigned long long)r - below;
    if (sh == 0 && cs > (unsigned)RCAP) {
        // one key value, more copies of it than a list holds (quantised clouds, planes at exact offsets)
        ka = lo; kb = lo;
        if (want2 && t + 1 >= cs) kb = min_above<EPT>(S, buf, K, hi);
        return;
    }
    // <= RCAP keys left (counted: the lists cannot overflow unless one wave holds more than RSEG of them -- then once more, halved)
    list_interval<EPT>(S, buf, K, lo, hi);
    __syncthreads();
    static_assert(RSEG >= RCAP, "a wave's list holds everything the last interval can");
    (void)pick_listed(S, buf, r, want2, ka, kb, above);            // (<= RCAP keys, the rank among them: cannot miss)
    if (above) kb = min_above<EPT>(S, buf, K, hi);
}

}  // namespace

// out4[0] = m (planarity survivors), [1] = median, [2] = MAD, [3] = n_kept;  out3 (nullable) = n, mean, std of the kept distances.
// use_prior: out4 still holds what the last launch left for the same correspondences' previous iteration.
#ifdef SICP_REJECT_TRACE
#define SICP_RT(i) tr[i] = clock64();
#else
#define SICP_RT(i)
#endif
template <int EPT>
__global__ __launch_bounds__(RB) void k_reject_reg(const double *__restrict__ dist, const uint8_t *__restrict__ flag, int n,
                                                   uint8_t *__restrict__ keep, double *__restrict__ out4, double *__restrict__ out3,
                                                   const IcpDev *__restrict__ st, int use_prior)
{
    __shared__ RejShared S;
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
#ifdef SICP_REJECT_TRACE
    long long tr[8];
#endif
    SICP_RT(0)
    // every load first: distances + verdicts (9 bytes per correspondence), the loop state's stop flag, the last launch's statistics
    double d[EPT];
    uint8_t fb[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * RB;
        const int ic = i < n ? i : n - 1;
        d[e] = dist[ic]; fb[e] = flag[ic];
    }
    const int stop = st ? st->stop : 0;
    const double pcnt = use_prior ? out4[0] : 0.0, pmed = use_prior ? out4[1] : 0.0, pmad = use_prior ? out4[2] : 0.0;
    for (int i = tid; i < RHC * 257; i += RB) S.hc[i] = 0u;
    if (stop) return;

    // a lane keeps ONLY the order-preserving keys of its distances (the distance is the key's exact inverse image): 16 elements per
    // lane are 32 registers of a 128-register budget
    unsigned long long k0[EPT];
    unsigned cnt = 0;
    double dmn = __builtin_inf(), dmx = -__builtin_inf();
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * RB;
        const bool f = i < n && fb[e] != 0;
        k0[e] = f ? rkey(d[e]) : RNOKEY;
        if (f) { dmn = fmin(dmn, d[e]); dmx = fmax(dmx, d[e]); }
        cnt += (unsigned)__popcll((long long)__ballot(f));
    }
    auto key_d = [&](int e) -> unsigned long long { return k0[e]; };
    SICP_RT(1)
    // the window of a settled run: half widths in units of the MAD sized for ~40 of the keys (a normal density holds 0.27 n keys per
    // MAD at its median, 0.43 n of the absolute deviations at theirs)
    const bool prior = pmad > 0.0 && pmad < __builtin_inf() && pcnt >= 1.0 && pmed == pmed;
    const double hw_med = pmad * fmin(0.25, 75.0 / pcnt), hw_mad = pmad * fmin(0.25, 47.0 / pcnt);
    if (prior) list_interval<EPT>(S, 0, key_d, rkey(pmed - hw_med), rkey(pmed + hw_med));
    dmn = rmin_f64(dmn); dmx = rmin_f64(-dmx);
    if (lane == 0) { S.wcnt[wid] = cnt; S.dmm[wid][0] = dmn; S.dmm[wid][1] = dmx; }
    __syncthreads();
    long m = 0;
#pragma unroll
    for (int w = 0; w < RW; ++w) { m += S.wcnt[w]; dmn = fmin(dmn, S.dmm[w][0]); dmx = fmin(dmx, S.dmm[w][1]); }
    if (m == 0) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) { const int i = tid + e * RB; if (i < n) keep[i] = 0; }
        if (tid == 0) {
            out4[0] = 0; out4[1] = __builtin_nan(""); out4[2] = __builtin_nan(""); out4[3] = 0;
            if (out3) { out3[0] = 0; out3[1] = __builtin_nan(""); out3[2] = __builtin_nan(""); }
        }
        return;
    }
    const long r = (m - 1) / 2;
    const bool want2 = (m & 1) == 0;
    unsigned long long ka, kb;
    SICP_RT(2)
    select_rank<EPT>(S, 0, key_d, r, want2, prior ? rkey(pmed - hw_med) : 1ull, prior ? rkey(pmed + hw_med) : 0ull, true, rkey(dmn), rkey(-dmx), ka, kb);
    const double med = (rval(ka) + rval(kb)) / 2.0;
    SICP_RT(3)
    // |d - med| (monotone in d on either side of med: its range follows from the distances' own)
    auto key_a = [&](int e) -> unsigned long long { return k0[e] != RNOKEY ? rkey(fabs(rval(k0[e]) - med)) : RNOKEY; };
    {
        const double u = fabs(dmn - med), v = fabs(-dmx - med);
        select_rank<EPT>(S, 1, key_a, r, want2, prior ? rkey(fmax(pmad - hw_mad, 0.0)) : 1ull, prior ? rkey(pmad + hw_mad) : 0ull, false,
                         rkey(0.0), rkey(u > v ? u : v), ka, kb);
    }
    const double mad = (rval(ka) + rval(kb)) / 2.0;
    SICP_RT(4)
    const double bound = 3 * mad;
    // keep mask + statistics of the kept distances, one pass over deviations from the median (a shift within a few MAD of the
    // mean: var = (S2 - S1^2 / n) / n loses nothing to cancellation)
    double v3[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * RB;
        const double dev = rval(k0[e]) - med;
        const bool kq = k0[e] != RNOKEY && fabs(dev) <= bound;
        if (i < n) keep[i] = kq ? 1 : 0;
        if (kq) { v3[0] += 1.0; v3[1] += dev; v3[2] = fma(dev, dev, v3[2]); }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) { v3[j] = wsum(v3[j]); if (lane == 0) S.red[wid][j] = v3[j]; }
    __syncthreads();
    SICP_RT(5)
    if (tid == 0) {
        double t[3] = {0.0, 0.0, 0.0};
        for (int w = 0; w < RW; ++w) for (int j = 0; j < 3; ++j) t[j] += S.red[w][j];
        out4[0] = (double)m; out4[1] = med; out4[2] = mad; out4[3] = t[0];
        if (out3) {
            const double var = (t[2] - t[1] * t[1] / t[0]) / t[0];
            out3[0] = t[0]; out3[1] = med + t[1] / t[0]; out3[2] = sqrt(var > 0.0 ? var : 0.0);
        }
#ifdef SICP_REJECT_TRACE
        SICP_RT(6)
        for (int i = 0; i < 6; ++i) out4[40 + i] = (double)(tr[i + 1] - tr[i]);     // (trace build: the caller's buffer holds 48 doubles)
#endif
    }
}

void launch_reject(hipStream_t s, const double *dist, const uint8_t *flag, long Q, uint8_t *keep, double *out4, const IcpDev *st,
                   double *out3, bool use_prior)
{
    const int n = (int)Q, up = use_prior ? 1 : 0;
    // keys per lane sized to the problem (the register-resident copy)
    if (Q <= 1 * RB) hipLaunchKernelGGL(k_reject_reg<1>, dim3(1), dim3(RB), 0, s, dist, flag, n, keep, out4, out3, st, up);
    else if (Q <= 2 * RB) hipLaunchKernelGGL(k_reject_reg<2>, dim3(1), dim3(RB), 0, s, dist, flag, n, keep, out4, out3, st, up);
    else if (Q <= 4 * RB) hipLaunchKernelGGL(k_reject_reg<4>, dim3(1), dim3(RB), 0, s, dist, flag, n, keep, out4, out3, st, up);
    else if (Q <= 8 * RB) hipLaunchKernelGGL(k_reject_reg<8>, dim3(1), dim3(RB), 0, s, dist, flag, n, keep, out4, out3, st, up);
    else hipLaunchKernelGGL(k_reject_reg<16>, dim3(1), dim3(RB), 0, s, dist, flag, n, keep, out4, out3, st, up);
}
static_assert(REJECT_MAX_Q <= 16 * RB, "the largest instantiation holds REJECT_MAX_Q keys");

}  // namespace sicp
