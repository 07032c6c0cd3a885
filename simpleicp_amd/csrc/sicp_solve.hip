// sicp_solve.hip -- everything after the match of one ICP iteration in ONE launch of ONE workgroup
// (Q <= SOLVE_MAX_Q): point-to-plane distances + planarity flag (corrpts.py:139-163,195-211),
// median / raw-MAD rejection (corrpts.py:165-188), kept-distance statistics (simpleicp.py:233-234),
// the Levenberg-Marquardt minimisation of optimization.py:65-124 on fused 6x6 normal-equation
// reductions -- including the 6x6 solves, done wave-parallel on the device -- and the residual
// statistics of simpleicp.py:356-379.  At Q ~ 1000 the whole tail of the iteration is latency, not
// bandwidth: one launch + one 512-byte read-back replaces ~15 launches and ~12 host round trips.
// (Large Q keeps the multi-block path: k_reject / k_stats / k_normal_eq + host solve.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sicp_internal.h"
#include "sicp_lanes.h"

namespace sicp {

namespace {

__device__ __forceinline__ void xfm(const Xf &H, double x, double y, double z, double &ox, double &oy, double &oz)
{
    double t;
    t = H.m[0] * x;  t = fma(H.m[1], y, t);  t = fma(H.m[2], z, t);   ox = t + H.m[3];
    t = H.m[4] * x;  t = fma(H.m[5], y, t);  t = fma(H.m[6], z, t);   oy = t + H.m[7];
    t = H.m[8] * x;  t = fma(H.m[9], y, t);  t = fma(H.m[10], z, t);  oz = t + H.m[11];
}
__device__ __forceinline__ double pdist(double dx, double dy, double dz, float nx, float ny, float nz)
{
    const double a = dx * (double)nx, b = dy * (double)ny, c = dz * (double)nz;
    return (a + b) + c;
}
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

struct Shared {
    union {
        unsigned long long key[3 * SOLVE_MAX_Q];   // order-statistics scratch: two exchange buffers + the sorted keys
        double ja[7][SOLVE_MAX_Q];             // staged rows [a0..a5 | r] of the kept correspondences (LM phase)
    };
    double red[SOLVE_BLOCK / 64][32];      // per-wave partials
    double ne[2][32];                      // normal equations: current / trial
    double sc[6];                          // sin, cos of the three trial angles
    double dx[8];                          // LM step + ok flag
    double bc[4];                          // broadcast scalars
};

// block-wide sum of up to NV values per thread -> s.red, folded by the first NV threads into dst[]
template <int NV>
__device__ void block_sum(Shared &s, const double (&v)[NV], double *dst)
{
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const double t = wsum(v[i]);
        if (lane == 0) s.red[wid][i] = t;
    }
    __syncthreads();
    if (tid < NV) {
        double t = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s.red[w][tid];
        dst[tid] = t;
    }
    __syncthreads();
}

// Register-resident bitonic sort of N = EPT * SOLVE_BLOCK keys: thread t holds the keys at positions
// t + e * SOLVE_BLOCK.  A stage with partner distance J exchanges
//   J >= SOLVE_BLOCK      inside the thread (its own registers),
//   64 <= J < SOLVE_BLOCK through LDS (double-buffered: one barrier per stage),
//   J < 64                with DPP / permlane-swap register moves -- no LDS, no barrier.
// Of the 55 stages of a 1024-key sort only 9 touch LDS (the in-LDS version paid a write, a read and a
// barrier in every one).  The network is unrolled at compile time (SIZE, J are template parameters) so
// every lane exchange is a fixed instruction.  merge_only: the input already is bitonic.
template <int EPT, int SIZE, int J>
__device__ __forceinline__ void bitonic_stage(unsigned long long *lds, unsigned long long (&k)[EPT], int &buf)
{
    constexpr int N = EPT * SOLVE_BLOCK;
    const int tid = threadIdx.x;
    if constexpr (J >= SOLVE_BLOCK) {
        constexpr int M = J / SOLVE_BLOCK;                       // partner register: e ^ M
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if ((e & M) == 0 && (e | M) < EPT) {
                const int i = tid + e * SOLVE_BLOCK;
                const bool up = (i & SIZE) == 0;
                const unsigned long long a = k[e], b = k[e | M];
                if ((a > b) == up) { k[e] = b; k[e | M] = a; }
            }
    } else if constexpr (J >= 64) {
        unsigned long long *cur = lds + buf * N;
#pragma unroll
        for (int e = 0; e < EPT; ++e) cur[tid + e * SOLVE_BLOCK] = k[e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + e * SOLVE_BLOCK;
            const unsigned long long o = cur[i ^ J];
            const bool take_min = ((i & J) == 0) == ((i & SIZE) == 0);
            k[e] = take_min ? (o < k[e] ? o : k[e]) : (o > k[e] ? o : k[e]);
        }
        buf ^= 1;                                                  // next LDS stage writes the other half
    } else {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + e * SOLVE_BLOCK;
            const unsigned long long o = lane_xor64<J>(k[e]);
            const bool take_min = ((i & J) == 0) == ((i & SIZE) == 0);
            k[e] = take_min ? (o < k[e] ? o : k[e]) : (o > k[e] ? o : k[e]);
        }
    }
}

template <int EPT, int SIZE, int J>
__device__ __forceinline__ void bitonic_merge_steps(unsigned long long *lds, unsigned long long (&k)[EPT], int &buf)
{
    bitonic_stage<EPT, SIZE, J>(lds, k, buf);
    if constexpr (J > 1) bitonic_merge_steps<EPT, SIZE, J / 2>(lds, k, buf);
}

template <int EPT, int SIZE>
__device__ __forceinline__ void bitonic_sizes(unsigned long long *lds, unsigned long long (&k)[EPT], int &buf)
{
    bitonic_merge_steps<EPT, SIZE, SIZE / 2>(lds, k, buf);
    if constexpr (SIZE < EPT * SOLVE_BLOCK) bitonic_sizes<EPT, SIZE * 2>(lds, k, buf);
}

template <int EPT>
__device__ void bitonic_regs(unsigned long long *lds /* 2 * EPT * SOLVE_BLOCK words */, unsigned long long (&k)[EPT],
                             bool merge_only)
{
    constexpr int N = EPT * SOLVE_BLOCK;
    int buf = 0;
    if (merge_only) bitonic_merge_steps<EPT, N, N / 2>(lds, k, buf);
    else            bitonic_sizes<EPT, 2>(lds, k, buf);
}

// normal equations of the unweighted residuals at x over the kept correspondences.
// Phase 1: each lane turns ITS correspondences (held in registers for the whole kernel) into the
// Jacobian row [a0..a5] and residual r and parks them in LDS (zeros when rejected).  Phase 2: the 29
// sums are 29 dot products of LDS columns, dealt to the waves; each costs ONE wave reduction (30
// block-wide reductions of per-lane accumulators serialise on the LDS permute pipe -- measured 25k
// cycles per evaluation against ~4k here).
__constant__ unsigned char kPairU[29] = {0,0,0,0,0,0, 1,1,1,1,1, 2,2,2,2, 3,3,3, 4,4, 5,  0,1,2,3,4,5, 6, 6};
__constant__ unsigned char kPairV[29] = {0,1,2,3,4,5, 1,2,3,4,5, 2,3,4,5, 3,4,5, 4,5, 5,  6,6,6,6,6,6, 7, 6};

template <int EPT>                                  // correspondences per thread: 1, 2 or 4
struct Corr {                                       // one thread's correspondences, register resident
    double px[EPT], py[EPT], pz[EPT], qx[EPT], qy[EPT], qz[EPT];
    float nx[EPT], ny[EPT], nz[EPT];
    bool keep[EPT];
};

// sc = (sin a1, cos a1, sin a2, cos a2, sin a3, cos a3) of the angles in x.
// Jacobian of r = n.(R p + t - p1) w.r.t. the Euler angles without the 27 entries of dR/dalpha:
// d(Rp)/dalpha_k = w_k x (Rp) with the instantaneous axes w1 = e_x, w2 = Rx e_y = (0, c1, s1),
// w3 = Rx Ry e_z = (s2, -s1 c2, c1 c2)  (R = Rx Ry Rz, mathutils.py:39-68), hence
// a_k = n.(w_k x Rp) = w_k.(Rp x n): one cross product per correspondence and five constants.
__device__ long long g_prof[4];     // trace only (SICP_SOLVE_TRACE): cycles in eval phase 1 / barrier / phase 2 / barrier

template <int EPT>
__device__ void eval_ne(Shared &s, int Q, const double x[6], const double sc[6], const Corr<EPT> &C, double nk, double *dst,
                        double *__restrict__ resid)
{
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    const long long e0 = clock64();
    const double s1 = sc[0], c1 = sc[1], s2 = sc[2], c2 = sc[3], s3 = sc[4], c3 = sc[5];
    Xf H;
    H.m[0] = c2 * c3;                 H.m[1] = -c2 * s3;                H.m[2] = s2;        H.m[3] = x[3];
    H.m[4] = c1 * s3 + s1 * s2 * c3;  H.m[5] = c1 * c3 - s1 * s2 * s3;  H.m[6] = -s1 * c2;  H.m[7] = x[4];
    H.m[8] = s1 * s3 - c1 * s2 * c3;  H.m[9] = s1 * c3 + c1 * s2 * s3;  H.m[10] = c1 * c2;  H.m[11] = x[5];
    const double w3y = -s1 * c2, w3z = c1 * c2;

#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * SOLVE_BLOCK;
        if (i < Q) {
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
#pragma unroll
            for (int c = 0; c < 7; ++c) s.ja[c][i] = a[c];
            if (resid) resid[i] = a[6];
        }
    }
    const long long e1 = clock64();
    __syncthreads();
    const long long e2 = clock64();
    // wave w owns sums w, w+8, w+16, w+24: accumulated together so the LDS reads overlap
    constexpr int NW = SOLVE_BLOCK / 64;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    int pu[4], pv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int p = wid + NW * k; pu[k] = p < 29 ? kPairU[p] : 0; pv[k] = p < 29 ? kPairV[p] : 0; }
#pragma unroll 2
    for (int i = lane; i < Q; i += 64) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double a = s.ja[pu[k]][i];
            const double b = (pv[k] == 7) ? 1.0 : s.ja[pv[k] == 7 ? 0 : pv[k]][i];
            acc[k] = fma(a, b, acc[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = wsum(acc[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int p = wid + NW * k; if (p < 29) dst[p] = acc[k]; }
    }
    if (tid == 0) dst[29] = nk;
    const long long e3 = clock64();
    __syncthreads();
    if (tid == 0) { g_prof[0] += e1 - e0; g_prof[1] += e2 - e1; g_prof[2] += e3 - e2; g_prof[3] += clock64() - e3; }
}

// sin/cos of (a + d) from sin/cos of a: exact addition theorem with a short Taylor series for the
// small step d (|d| <= 0.25 rad: d^17/17! < 2e-25); larger steps take the library routine.
// Keeps the LM loop free of double-precision sincos calls (hundreds of dependent instructions each).
__device__ __forceinline__ void sincos_step(double a_new, double d, double sa, double ca, double &sn, double &cn)
{
    if (fabs(d) > 0.25) { sincos(a_new, &sn, &cn); return; }
    const double d2 = d * d;
    double sd = 1.0 / 1307674368000.0;                       // 1/15!
    sd = fma(sd, d2, -1.0 / 6227020800.0);                   // 1/13!
    sd = fma(sd, d2, 1.0 / 39916800.0);
    sd = fma(sd, d2, -1.0 / 362880.0);
    sd = fma(sd, d2, 1.0 / 5040.0);
    sd = fma(sd, d2, -1.0 / 120.0);
    sd = fma(sd, d2, 1.0 / 6.0);
    sd = d - d * d2 * sd;                                     // sin d
    double cd = 1.0 / 20922789888000.0;                      // 1/16!
    cd = fma(cd, d2, -1.0 / 87178291200.0);                  // 1/14!
    cd = fma(cd, d2, 1.0 / 479001600.0);
    cd = fma(cd, d2, -1.0 / 3628800.0);
    cd = fma(cd, d2, 1.0 / 40320.0);
    cd = fma(cd, d2, -1.0 / 720.0);
    cd = fma(cd, d2, 1.0 / 24.0);
    cd = 1.0 - d2 * (0.5 - d2 * cd);                          // cos d
    sn = fma(sa, cd, ca * sd);
    cn = fma(ca, cd, -(sa * sd));
}

__device__ __forceinline__ bool observed(double w) { return w > 0 && w < __builtin_inf(); }

__device__ double objective(const double *ne, double w, const double x[6], const SolveArgs &A)
{
    double c = w * w * ne[28];
#pragma unroll
    for (int j = 0; j < 6; ++j)
        if (observed(A.ow[j])) { const double e = A.ow[j] * (x[j] - A.obs[j]); c += e * e; }
    return c;
}

// wave 0: Gauss-Jordan on the 6x7 augmented system held one element per lane (lane = 7*row + col);
// fixed parameters get an identity row/column.  Writes s.dx[0..5] and s.dx[6] = 1 on success.
__device__ void lm_solve(Shared &s, const SolveArgs &A, const double *ne, double w, const double x[6], double lambda)
{
    const int lane = threadIdx.x;                 // caller guarantees threadIdx.x < 64
    const int i = lane / 7, j = lane % 7;
    double a = 0.0;
    if (lane < 42) {
        const bool fi = !(A.ow[i] < __builtin_inf());          // parameter i fixed (weight = inf)
        if (j < 6) {
            const bool fj = !(A.ow[j] < __builtin_inf());
            if (fi || fj) a = (i == j) ? 1.0 : 0.0;
            else {
                const int u = i < j ? i : j, v = i < j ? j : i;
                const int t = u * 6 - (u * (u - 1)) / 2 + (v - u);   // index in the upper triangle
                a = w * w * ne[t];
                if (i == j) { if (observed(A.ow[i])) a += A.ow[i] * A.ow[i]; a += lambda * a; }
            }
        } else {
            if (fi) a = 0.0;
            else {
                double g = w * w * ne[21 + i];
                if (observed(A.ow[i])) g += A.ow[i] * A.ow[i] * (x[i] - A.obs[i]);
                a = -g;
            }
        }
    }
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double piv = __shfl(a, k * 7 + k, 64);
        const double rk = __shfl(a, k * 7 + (lane < 42 ? j : 0), 64);
        const double fac = __shfl(a, (lane < 42 ? i : 0) * 7 + k, 64);
        ok = ok && (piv > 0.0) && (piv < __builtin_inf());
        const double t = rk / piv;
        if (lane < 42) a = (i == k) ? t : (a - fac * t);
    }
    if (lane < 42 && j == 6) s.dx[i] = a;
    if (lane == 0) s.dx[6] = ok ? 1.0 : 0.0;
}

}  // namespace

// out layout (doubles): 0 n_planar, 1 median, 2 mad, 3 n_kept, 4 dist_mean, 5 dist_std, 6 w_used, 7 cost,
// 8 lm_steps, 9 ne_evals, 10..15 x, 16 res_mean, 17 res_std, 18 status (0 ok / 1 too few / 2 numeric),
// 20..49 normal equations at x
template <int EPT>
__global__ __launch_bounds__(SOLVE_BLOCK) void k_icp_solve(
    const double *__restrict__ qx, const double *__restrict__ qy, const double *__restrict__ qz,
    const float *__restrict__ normals, const float *__restrict__ planarity, const double *__restrict__ p2,
    const int64_t *__restrict__ idx, SolveArgs A, double *__restrict__ dist, uint8_t *__restrict__ flag,
    uint8_t *__restrict__ keep, double *__restrict__ resid, double *__restrict__ out)
{
    __shared__ Shared s;
    const int tid = threadIdx.x;
    const long Q = A.Q;
    long long tk[6]; tk[0] = clock64();          // phase stamps (shader clock), reported in out[50..54]

    // ---- distances + planarity flag; every thread keeps its correspondences in registers ----
    Corr<EPT> C;
    double cnt[1] = {0.0};
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const long i = tid + (long)e * SOLVE_BLOCK;
        C.keep[e] = false;
        C.px[e] = C.py[e] = C.pz[e] = C.qx[e] = C.qy[e] = C.qz[e] = 0.0; C.nx[e] = C.ny[e] = C.nz[e] = 0.f;
        if (i < Q) {
            C.px[e] = p2[3 * i]; C.py[e] = p2[3 * i + 1]; C.pz[e] = p2[3 * i + 2];
            C.qx[e] = qx[i]; C.qy[e] = qy[i]; C.qz[e] = qz[i];
            C.nx[e] = normals[3 * i]; C.ny[e] = normals[3 * i + 1]; C.nz[e] = normals[3 * i + 2];
            double X, Y, Z;
            xfm(A.H, C.px[e], C.py[e], C.pz[e], X, Y, Z);
            dist[i] = pdist(X - C.qx[e], Y - C.qy[e], Z - C.qz[e], C.nx[e], C.ny[e], C.nz[e]);
            const int64_t mi = idx[i];
            bool fb = mi >= 0 && planarity[i] >= A.min_planarity;
            if (fb && A.pl2) fb = mi < A.pl2_n && A.pl2[mi] >= A.min_planarity;      // corrpts.py:158-163 (NaN fails)
            const uint8_t f = fb ? 1 : 0;
            flag[i] = f; cnt[0] += f;
        }
    }
    block_sum<1>(s, cnt, s.bc);
    const long m = (long)s.bc[0];
    __syncthreads();
    if (m == 0) {
        for (long i = tid; i < Q; i += blockDim.x) keep[i] = 0;
        if (tid == 0) {
            for (int k = 0; k < 50; ++k) out[k] = 0;
            out[1] = __builtin_nan(""); out[2] = __builtin_nan(""); out[18] = 1;
            __threadfence_system();
            __hip_atomic_store(out + 55, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    tk[1] = clock64();
    // ---- median / raw MAD: exact order statistics; keys sorted in registers (bitonic_regs) ----
    constexpr int NS = EPT * SOLVE_BLOCK;
    unsigned long long kk[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const long i = tid + (long)e * SOLVE_BLOCK;
        kk[e] = (i < Q && flag[i]) ? okey(dist[i]) : ~0ull;
    }
    __syncthreads();
    bitonic_regs<EPT>(s.key, kk, false);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e) s.key[2 * NS + tid + e * SOLVE_BLOCK] = kk[e];      // sorted order: position = t + e*BLOCK
    __syncthreads();
    const double med = (oval(s.key[2 * NS + (m - 1) / 2]) + oval(s.key[2 * NS + m / 2])) / 2.0;
    // |d - med| over the SORTED distances falls towards the median and rises after it (the unflagged
    // sentinels stay at the top): a bitonic sequence, so one merge (log2 n stages) sorts it
#pragma unroll
    for (int e = 0; e < EPT; ++e) if (kk[e] != ~0ull) kk[e] = okey(fabs(oval(kk[e]) - med));
    __syncthreads();
    bitonic_regs<EPT>(s.key, kk, true);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e) s.key[2 * NS + tid + e * SOLVE_BLOCK] = kk[e];
    __syncthreads();
    const double mad = (oval(s.key[2 * NS + (m - 1) / 2]) + oval(s.key[2 * NS + m / 2])) / 2.0;
    __syncthreads();
    const double bound = 3 * mad;
    tk[2] = clock64();
    // ---- keep mask + mean of kept distances, then their std (two-pass, ddof 0) ----
    double v2[2] = {0.0, 0.0};
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const long i = tid + (long)e * SOLVE_BLOCK;
        if (i < Q) {
            const uint8_t k = (flag[i] && fabs(dist[i] - med) <= bound) ? 1 : 0;
            keep[i] = k; C.keep[e] = k != 0;
            if (k) { v2[0] += 1.0; v2[1] += dist[i]; }
        }
    }
    block_sum<2>(s, v2, s.bc);
    const double nk = s.bc[0], dmean = s.bc[1] / s.bc[0];
    __syncthreads();
    double v1[1] = {0.0};
    for (long i = tid; i < Q; i += blockDim.x) if (keep[i]) { const double e = dist[i] - dmean; v1[0] += e * e; }
    block_sum<1>(s, v1, s.bc + 2);
    const double dstd = sqrt(s.bc[2] / nk);
    if (tid == 0) { out[0] = (double)m; out[1] = med; out[2] = mad; out[3] = nk; out[4] = dmean; out[5] = dstd; }
    if (nk < 6.0) {
        if (tid == 0) {
            for (int k = 6; k < 50; ++k) out[k] = 0;
            for (int k = 0; k < 6; ++k) out[10 + k] = A.x0[k];
            out[18] = 1;
            __threadfence_system();
            __hip_atomic_store(out + 55, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    const double w = (A.w > 0) ? A.w : 1.0 / (dstd * dstd);          // simpleicp.py:233-234

    tk[3] = clock64();
    // ---- Levenberg-Marquardt on the fused 6x6 reductions ----
    int nfree = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) nfree += (A.ow[j] < __builtin_inf()) ? 1 : 0;
    double x[6], xn[6], sc[6], scn[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) { x[j] = A.x0[j]; sc[j] = A.sc0[j]; }
    int cur = 0, steps = 0, evals = 0;
    eval_ne(s, (int)Q, x, sc, C, nk, s.ne[cur], nullptr); ++evals;
    double cost = objective(s.ne[cur], w, x, A);
    double lambda = 0.0;
    bool rows_current = true;        // s.ja / s.ne[cur] describe x (no rejected trial since)
    long long t_solve = 0;           // cycles spent in the 6x6 solves (trace only)
    for (int it = 0; it < A.max_steps && nfree > 0; ++it) {
        bool accepted = false, converged = false;
        double costn = cost, dxmax = 0.0;
        for (int tries = 0; tries < 40; ++tries) {
            const long long ts0 = clock64();
            if (tid < 64) lm_solve(s, A, s.ne[cur], w, x, lambda);
            __syncthreads();
            t_solve += clock64() - ts0;
            const bool ok = s.dx[6] != 0.0;
            dxmax = 0.0;
            double dstep[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) { const double d = s.dx[j]; dstep[j] = d; xn[j] = x[j] + d; dxmax = fmax(dxmax, fabs(d)); }
            __syncthreads();                                   // s.dx consumed before the next solve rewrites it
            if (!ok || !(dxmax < __builtin_inf())) { lambda = lambda > 0 ? lambda * 10 : 1e-6; continue; }
#pragma unroll
            for (int j = 0; j < 3; ++j) sincos_step(xn[j], dstep[j], sc[2 * j], sc[2 * j + 1], scn[2 * j], scn[2 * j + 1]);
            double xm = 0.0;
#pragma unroll
            for (int j = 0; j < 6; ++j) xm = fmax(xm, fabs(x[j]));
            if (lambda == 0.0 && dxmax <= 1e-10 * (1.0 + xm)) {
                // the undamped Gauss-Newton step from x is below 1e-10: x is the minimiser to that accuracy
                // (the reference stops at 1e-8); stop here -- rows and sums staged in LDS describe x
                converged = true; break;
            }
            eval_ne(s, (int)Q, xn, scn, C, nk, s.ne[cur ^ 1], nullptr); ++evals;
            costn = objective(s.ne[cur ^ 1], w, xn, A);
            if (costn <= cost * (1 + 1e-12) || dxmax < 1e-15) { accepted = true; rows_current = true; break; }   // 1e-12: rounding noise of the sums
            rows_current = false;
            lambda = lambda > 0 ? lambda * 10 : 1e-6;
        }
        if (converged || !accepted) break;
#pragma unroll
        for (int j = 0; j < 6; ++j) { x[j] = xn[j]; sc[j] = scn[j]; }
        cur ^= 1; cost = costn;
        lambda = lambda > 0 ? lambda * 0.1 : 0.0;
        if (lambda < 1e-12) lambda = 0.0;
        ++steps;
        double xmax = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) xmax = fmax(xmax, fabs(x[j]));
        if (dxmax <= 1e-13 * (1.0 + xmax)) break;
    }
    // ---- residuals at the optimum + their mean / std ----
    __syncthreads();
    tk[4] = clock64();
    if (rows_current) {
        // the last evaluation was at x: its residual column is still staged in LDS
        for (int i = tid; i < (int)Q; i += blockDim.x) resid[i] = s.ja[6][i];
    } else {
        eval_ne(s, (int)Q, x, sc, C, nk, s.ne[cur], resid); ++evals;
        cost = objective(s.ne[cur], w, x, A);
    }
    const double rmean = s.ne[cur][27] / s.ne[cur][29];
    double v3[1] = {0.0};
    for (long i = tid; i < Q; i += blockDim.x) if (keep[i]) { const double e = resid[i] - rmean; v3[0] += e * e; }
    block_sum<1>(s, v3, s.bc + 3);
    if (tid < 30) out[20 + tid] = s.ne[cur][tid];
    if (tid == 0) {
        out[6] = w; out[7] = cost; out[8] = steps; out[9] = evals;
        for (int j = 0; j < 6; ++j) out[10 + j] = x[j];
        out[16] = rmean; out[17] = sqrt(s.bc[3] / s.ne[cur][29]);
        out[18] = (cost < __builtin_inf()) ? 0.0 : 2.0;
        tk[5] = clock64();
        for (int k = 0; k < 5; ++k) out[50 + k] = (double)(tk[k + 1] - tk[k]);
        out[56] = (double)t_solve;
        for (int k = 0; k < 4; ++k) { out[57 + k] = (double)g_prof[k]; g_prof[k] = 0; }
    }
    // completion ticket for the host, which polls this pinned word instead of waiting for the
    // end-of-kernel signal: all result words first (system-scope fence), then the sequence number
    __syncthreads();
    if (tid == 0) {
        __threadfence_system();
        __hip_atomic_store(out + 55, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

void launch_icp_solve(hipStream_t st, const double *qx, const double *qy, const double *qz, const float *normals,
                      const float *planarity, const double *p2, const int64_t *idx, const SolveArgs &A, double *dist,
                      uint8_t *flag, uint8_t *keep, double *resid, double *out)
{
    // correspondences per thread: the register-resident copy is sized to the problem
    if (A.Q <= SOLVE_BLOCK)
        hipLaunchKernelGGL(k_icp_solve<1>, dim3(1), dim3(SOLVE_BLOCK), 0, st, qx, qy, qz, normals, planarity, p2, idx, A, dist,
                           flag, keep, resid, out);
    else if (A.Q <= 2 * SOLVE_BLOCK)
        hipLaunchKernelGGL(k_icp_solve<2>, dim3(1), dim3(SOLVE_BLOCK), 0, st, qx, qy, qz, normals, planarity, p2, idx, A, dist,
                           flag, keep, resid, out);
    else
        hipLaunchKernelGGL(k_icp_solve<4>, dim3(1), dim3(SOLVE_BLOCK), 0, st, qx, qy, qz, normals, planarity, p2, idx, A, dist,
                           flag, keep, resid, out);
}

}  // namespace sicp
