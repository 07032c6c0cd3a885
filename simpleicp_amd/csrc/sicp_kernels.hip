// sicp_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the simpleICP inner loop.
//
// Compile with -ffp-contract=off: every fused multiply-add below is an explicit fma() so the
// arithmetic contract (T)/(D)/(P) of include/simpleicp_hip.h holds bit-for-bit against
// oracle/sicp_oracle.c.
//
// Data layout in HBM (all owned by the ctx, see sicp_internal.h):
//   cloud slot : SoA  x[npad] | y[npad] | z[npad]  float64, npad = n rounded up to TILE_PTS,
//                pad rows hold SICP_PAD_COORD (their squared distance overflows to +inf and
//                can never win a strict `<`), so no scan kernel needs a tail branch.
//   queries    : SoA  qx|qy|qz [qpad] float64, qpad = Q rounded up to 1024.
//   partials   : [nchunks][qpad] (d2 f64, idx u32) -- one row per chunk of the scanned cloud.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>

#include "sicp_internal.h"
#include "sicp_lanes.h"
#include "sicp_normals.h"

namespace sicp {

// ------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void xform(const Xf &H, double x, double y, double z,
                                      double &ox, double &oy, double &oz)
{
    double t;
    t = H.m[0] * x;  t = fma(H.m[1], y, t);  t = fma(H.m[2], z, t);   ox = t + H.m[3];
    t = H.m[4] * x;  t = fma(H.m[5], y, t);  t = fma(H.m[6], z, t);   oy = t + H.m[7];
    t = H.m[8] * x;  t = fma(H.m[9], y, t);  t = fma(H.m[10], z, t);  oz = t + H.m[11];
}

__device__ __forceinline__ double plane_dist(double dx, double dy, double dz, float nx, float ny, float nz)
{
    const double a = dx * (double)nx;
    const double b = dy * (double)ny;
    const double c = dz * (double)nz;
    return (a + b) + c;
}

// (every lane gets the total; DPP butterfly, see sicp_lanes.h -- a __shfl_down chain costs ~1500 cycles)
__device__ __forceinline__ double wave_sum(double v) { return wsum(v); }

// ------------------------------------------------------------------------------------
// cloud upload: AoS (n,3) -> padded SoA                       PointCloud ctor, pointcloud.py:15-49
// ------------------------------------------------------------------------------------
__global__ void k_aos_to_soa(const double *__restrict__ aos, long n, long npad,
                             double *__restrict__ x, double *__restrict__ y, double *__restrict__ z)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npad) return;
    if (i < n) { x[i] = aos[3 * i]; y[i] = aos[3 * i + 1]; z[i] = aos[3 * i + 2]; }
    else       { x[i] = SICP_PAD_COORD; y[i] = SICP_PAD_COORD; z[i] = SICP_PAD_COORD; }
}

__global__ void k_pad_fill(double *__restrict__ x, double *__restrict__ y, double *__restrict__ z, long n, long npad)
{
    const long i = n + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npad) { x[i] = SICP_PAD_COORD; y[i] = SICP_PAD_COORD; z[i] = SICP_PAD_COORD; }
}

__global__ void k_soa_to_aos(const double *__restrict__ x, const double *__restrict__ y,
                             const double *__restrict__ z, long n, double *__restrict__ aos)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    aos[3 * i] = x[i]; aos[3 * i + 1] = y[i]; aos[3 * i + 2] = z[i];
}

// the cloud in chunks of CH points, each chunk its x | y | z columns back to back: ONE contiguous piece per chunk for the copy engine
// (sicp_cloud_download_both: three pieces of 2 MiB per chunk moved 45 GB/s, one of 12 MiB moves 55 -- profiles/r6/d2h_rate.txt)
__global__ void k_pack_chunks(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z, long n, long CH,
                              double *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long ch = i / CH, r = i - ch * CH;
    double *o = out + 3 * ch * CH + r;
    o[0] = x[i]; o[CH] = y[i]; o[2 * CH] = z[i];
}

// PointCloud.transform_by_H, pointcloud.py:205-217 -- in place, HBM-bound (48 B/point).
__global__ void k_transform(double *__restrict__ x, double *__restrict__ y, double *__restrict__ z, long n, Xf H)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double X, Y, Z;
    xform(H, x[i], y[i], z[i], X, Y, Z);
    x[i] = X; y[i] = Y; z[i] = Z;
}

// gather rows of a SoA cloud into SoA queries (selected fixed points)
__global__ void k_gather_queries(const double *__restrict__ x, const double *__restrict__ y,
                                 const double *__restrict__ z, const int64_t *__restrict__ sel, long Q, long qpad,
                                 double *__restrict__ qx, double *__restrict__ qy, double *__restrict__ qz)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= qpad) return;
    if (i < Q) { const int64_t s = sel ? sel[i] : (int64_t)i; qx[i] = x[s]; qy[i] = y[s]; qz[i] = z[s]; }
    else       { qx[i] = 0.0; qy[i] = 0.0; qz[i] = 0.0; }
}

// select_in_range's verdict (pointcloud.py:165-169): a neighbour exists below the strict bound
__global__ void k_found_mask(const int64_t *__restrict__ idx, long Q, uint8_t *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Q) out[i] = idx[i] >= 0 ? 1 : 0;
}

__global__ void k_aos_queries(const double *__restrict__ aos, long Q, long qpad,
                              double *__restrict__ qx, double *__restrict__ qy, double *__restrict__ qz)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= qpad) return;
    if (i < Q) { qx[i] = aos[3 * i]; qy[i] = aos[3 * i + 1]; qz[i] = aos[3 * i + 2]; }
    else       { qx[i] = 0.0; qy[i] = 0.0; qz[i] = 0.0; }
}

// ------------------------------------------------------------------------------------
// K1: brute-force 1-NN scan              CorrPts.match, corrpts.py:124-137 (+ simpleicp.py:188)
//
// lane  = R queries held in registers (query-stationary);
// block = 256 lanes -> 256*R queries, scans one chunk of the cloud;
// the cloud streams through LDS in TILE_PTS-point tiles: coalesced 8-B/lane global loads,
// the H transform is applied ONCE per point on the way into LDS (fused transform_by_H, the
// reference's O(N) pandas pass per iteration disappears), then every lane reads the tile by
// LDS broadcast (all lanes same address: conflict-free).
// Ascending scan + strict `<` == lexicographic (d2, idx) minimum within the chunk.
// ------------------------------------------------------------------------------------
template <int R, bool XFORM>
__global__ __launch_bounds__(KNN_BLOCK) void k_knn1_scan(
    const double *__restrict__ qx, const double *__restrict__ qy, const double *__restrict__ qz, int qpad,
    const double *__restrict__ px, const double *__restrict__ py, const double *__restrict__ pz,
    long npad, int tile_step, int tiles_per_chunk, Xf H, double *__restrict__ part_d2,
    uint32_t *__restrict__ part_idx)
{
    __shared__ double sx[TILE_PTS], sy[TILE_PTS], sz[TILE_PTS];
    const int tid = threadIdx.x;
    const long q0 = (long)blockIdx.x * (KNN_BLOCK * R);

    double ax[R], ay[R], az[R], best[R];
    uint32_t bidx[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long q = q0 + r * KNN_BLOCK + tid;
        ax[r] = qx[q]; ay[r] = qy[q]; az[r] = qz[q];
        best[r] = __builtin_inf(); bidx[r] = 0xffffffffu;
    }

    // chunk = tiles [y*tile_step, y*tile_step + tiles_per_chunk): contiguous cover when the two are
    // equal, a strided subsample (bound pre-pass of the filtered scan) when tile_step is larger
    const long c0 = (long)blockIdx.y * tile_step * TILE_PTS;
    long c1 = c0 + (long)tiles_per_chunk * TILE_PTS;
    if (c1 > npad) c1 = npad;

    for (long t0 = c0; t0 < c1; t0 += TILE_PTS) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < TILE_PTS / KNN_BLOCK; ++u) {
            const int s = u * KNN_BLOCK + tid;
            double X = px[t0 + s], Y = py[t0 + s], Z = pz[t0 + s];
            if (XFORM) { double a, b, c; xform(H, X, Y, Z, a, b, c); X = a; Y = b; Z = c; }
            sx[s] = X; sy[s] = Y; sz[s] = Z;
        }
        __syncthreads();
        const uint32_t base = (uint32_t)t0;
#pragma unroll 4
        for (int j = 0; j < TILE_PTS; ++j) {
            const double X = sx[j], Y = sy[j], Z = sz[j];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double dx = X - ax[r], dy = Y - ay[r], dz = Z - az[r];
                const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
                if (d2 < best[r]) { best[r] = d2; bidx[r] = base + (uint32_t)j; }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long q = q0 + r * KNN_BLOCK + tid;
        part_d2[(long)blockIdx.y * qpad + q] = best[r];
        part_idx[(long)blockIdx.y * qpad + q] = bidx[r];
    }
}

// ------------------------------------------------------------------------------------
// K1f: FILTERED brute-force 1-NN scan -- same result as K1, ~4x fewer VALU cycles per pair.
//
// The exact test costs 6 FP64 ops + compare/select per pair (FP64 VALU runs at 16 lanes/clk).
// Here every pair first goes through a conservative FP32 filter in expanded form
//        s = |p|^2 - 2 q.p          (3 v_fma_f32 + 1 v_cmp_f32, 32 lanes/clk)
// against a per-query threshold  thr >= bound - |q|^2 + M,  where `bound` is the exact squared
// distance to SOME cloud point (previous iteration's match, or a strided-subsample pre-pass)
// and M = 6 * 2^-24 * (rmax + |q|)^2 bounds the FP32 rounding error of s rigorously
// (inputs rounded to f32: u(|p|^2 + 4|q||p|); 3 fma roundings: 3u(|p|^2 + 2|q||p|);
// sum <= 5u(|p|+|q|)^2, u = 2^-24).  Any point whose exact d2 is <= bound therefore passes, and
// only passing pairs (a few dozen per query for the whole scan) are re-evaluated with the exact
// FP64 contract (T)+(D) from the original coordinates.  The final answer is the lexicographic
// (d2, idx) minimum over exactly-evaluated candidates == the plain brute-force answer, bit for bit.
// Predicates are OR-ed in scalar registers over a group of FS_G points, so the hot loop has no
// per-pair branch or select.  Tiles are double-buffered in LDS as float4 (x, y, z, |p|^2).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float filter_threshold(double bound, double qq, double rmax, double margin = 6.0)
{
    if (!(bound < __builtin_inf())) return __builtin_inff();
    const double sN = rmax + sqrt(qq);
    const double M = (margin * 5.9604644775390625e-08 * 1.0001) * sN * sN;
    const float f = (float)((bound - qq) + M);
    return f + fabsf(f) * 1.1920929e-07f + 1.0e-37f;      // round up past the f64->f32 conversion
}

template <int BLOCK, bool XFORM>
__global__ __launch_bounds__(BLOCK) void k_knn1_fscan(
    const double *__restrict__ qx, const double *__restrict__ qy, const double *__restrict__ qz, int qpad, long Q,
    const double *__restrict__ bound,
    const double *__restrict__ px, const double *__restrict__ py, const double *__restrict__ pz,
    int ntiles, Xf H, double rmax, double *__restrict__ part_d2, uint32_t *__restrict__ part_idx)
{
    constexpr int R = FS_R;
    __shared__ float4 tile[2][FS_TILE];
    const int tid = threadIdx.x;
    const long q0 = (long)blockIdx.x * (BLOCK * R);

    float m2x[R], m2y[R], m2z[R], thr[R];
    double best[R];
    uint32_t bidx[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long q = q0 + r * BLOCK + tid;
        const double x = qx[q], y = qy[q], z = qz[q];
        const double qq = fma(z, z, fma(y, y, x * x));
        m2x[r] = (float)(-2.0 * x); m2y[r] = (float)(-2.0 * y); m2z[r] = (float)(-2.0 * z);
        thr[r] = (q < Q) ? filter_threshold(bound[q], qq, rmax) : -__builtin_inff();      // padding lanes never hit
        best[r] = __builtin_inf(); bidx[r] = 0xffffffffu;
    }

    const int t_lo = (int)((long)blockIdx.y * ntiles / gridDim.y);
    const int t_hi = (int)((long)(blockIdx.y + 1) * ntiles / gridDim.y);

    constexpr int PER = FS_TILE / BLOCK;
    double lx[PER], ly[PER], lz[PER];
    auto gload = [&](int t) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const long g = (long)t * FS_TILE + u * BLOCK + tid;
            lx[u] = px[g]; ly[u] = py[g]; lz[u] = pz[g];
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            double X = lx[u], Y = ly[u], Z = lz[u];
            if (XFORM) { double a, b, c; xform(H, X, Y, Z, a, b, c); X = a; Y = b; Z = c; }
            const double pp = fma(Z, Z, fma(Y, Y, X * X));
            tile[buf][u * BLOCK + tid] = make_float4((float)X, (float)Y, (float)Z, (float)pp);
        }
    };
    if (t_lo < t_hi) { gload(t_lo); lstore(0); }
    int cur = 0;
    for (int t = t_lo; t < t_hi; ++t) {
        __syncthreads();                     // tile[cur] complete; nobody still reads tile[cur^1]
        const bool more = (t + 1 < t_hi);
        if (more) gload(t + 1);              // HBM latency hides under this tile's compute
        const uint32_t tbase = (uint32_t)t * FS_TILE;
        for (int g0 = 0; g0 < FS_TILE; g0 += FS_G) {
            // group minimum of s per query, then ONE compare per query: the per-pair cost is
            // 3 fma + 1/2 min3; the lane masks are OR-ed in scalar registers
            float gm[R];
            {
                const float4 P = tile[cur][g0];
#pragma unroll
                for (int r = 0; r < R; ++r) gm[r] = fmaf(m2x[r], P.x, fmaf(m2y[r], P.y, fmaf(m2z[r], P.z, P.w)));
            }
#pragma unroll
            for (int g = 1; g < FS_G; ++g) {
                const float4 P = tile[cur][g0 + g];
#pragma unroll
                for (int r = 0; r < R; ++r)
                    gm[r] = fminf(gm[r], fmaf(m2x[r], P.x, fmaf(m2y[r], P.y, fmaf(m2z[r], P.z, P.w))));
            }
            unsigned long long hit = 0ull;
#pragma unroll
            for (int r = 0; r < R; ++r) hit |= __builtin_amdgcn_ballot_w64(gm[r] < thr[r]);
            if (hit != 0ull) {
                // rare: re-run the group, evaluate passing pairs exactly from the original coordinates
#pragma unroll 1
                for (int g = 0; g < FS_G; ++g) {
                    const float4 P = tile[cur][g0 + g];
                    const uint32_t id = tbase + (uint32_t)(g0 + g);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const float s = fmaf(m2x[r], P.x, fmaf(m2y[r], P.y, fmaf(m2z[r], P.z, P.w)));
                        if (s < thr[r]) {
                            double X = px[id], Y = py[id], Z = pz[id];
                            if (XFORM) { double a, b, c; xform(H, X, Y, Z, a, b, c); X = a; Y = b; Z = c; }
                            const long q = q0 + r * BLOCK + tid;
                            const double ax = qx[q], ay = qy[q], az = qz[q];
                            const double dx = X - ax, dy = Y - ay, dz = Z - az;
                            const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
                            if (d2 < best[r]) {          // ascending scan: ties keep the lower index
                                best[r] = d2; bidx[r] = id;
                                const double qq = fma(az, az, fma(ay, ay, ax * ax));
                                thr[r] = fminf(thr[r], filter_threshold(d2, qq, rmax));
                            }
                        }
                    }
                }
            }
        }
        if (more) lstore(cur ^ 1);
        cur ^= 1;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long q = q0 + r * BLOCK + tid;
        part_d2[(long)blockIdx.y * qpad + q] = best[r];
        part_idx[(long)blockIdx.y * qpad + q] = bidx[r];
    }
}

// ------------------------------------------------------------------------------------
// K1r: the filtered scan with the exact work split off -- the streaming kernel only RECORDS which
// (query, 8-point group) pairs pass the conservative FP32 filter (a few dozen per query for the whole
// cloud), so its hot loop carries no FP64 state at all (110 instead of 170 VGPRs: twice the
// occupancy); k_knn1_fixup then evaluates the recorded groups with the exact FP64 contract, one wave
// per query, and takes the lexicographic (d2, idx) minimum.  Same answer as K1 / K1f.  A query whose
// list overflows `cap` is reported and the caller reruns the self-contained K1f kernel.
// ------------------------------------------------------------------------------------
template <int BLOCK, bool XFORM>
__global__ __launch_bounds__(BLOCK) void k_knn1_frec(
    const double *__restrict__ qx, const double *__restrict__ qy, const double *__restrict__ qz, long Q,
    const double *__restrict__ bound,
    const double *__restrict__ px, const double *__restrict__ py, const double *__restrict__ pz,
    int ntiles, Xf H, double rmax, uint32_t *__restrict__ hit_cnt, uint32_t *__restrict__ hit_list, uint32_t cap)
{
    constexpr int R = FS_R;
    __shared__ float4 tile[2][FS_TILE];
    const int tid = threadIdx.x;
    const long q0 = (long)blockIdx.x * (BLOCK * R);

    float m2x[R], m2y[R], m2z[R], thr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long q = q0 + r * BLOCK + tid;
        const double x = qx[q], y = qy[q], z = qz[q];
        const double qq = fma(z, z, fma(y, y, x * x));
        m2x[r] = (float)(-2.0 * x); m2y[r] = (float)(-2.0 * y); m2z[r] = (float)(-2.0 * z);
        thr[r] = (q < Q) ? filter_threshold(bound[q], qq, rmax) : -__builtin_inff();   // padding lanes never hit
    }
    const int t_lo = (int)((long)blockIdx.y * ntiles / gridDim.y);
    const int t_hi = (int)((long)(blockIdx.y + 1) * ntiles / gridDim.y);

    constexpr int PER = FS_TILE / BLOCK;
    double lx[PER], ly[PER], lz[PER];
    auto gload = [&](int t) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const long g = (long)t * FS_TILE + u * BLOCK + tid;
            lx[u] = px[g]; ly[u] = py[g]; lz[u] = pz[g];
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            double X = lx[u], Y = ly[u], Z = lz[u];
            if (XFORM) { double a, b, c; xform(H, X, Y, Z, a, b, c); X = a; Y = b; Z = c; }
            const double pp = fma(Z, Z, fma(Y, Y, X * X));
            tile[buf][u * BLOCK + tid] = make_float4((float)X, (float)Y, (float)Z, (float)pp);
        }
    };
    if (t_lo < t_hi) { gload(t_lo); lstore(0); }
    int cur = 0;
    for (int t = t_lo; t < t_hi; ++t) {
        __syncthreads();
        const bool more = (t + 1 < t_hi);
        if (more) gload(t + 1);
        const uint32_t tbase = (uint32_t)t * FS_TILE;
        for (int g0 = 0; g0 < FS_TILE; g0 += FS_G) {
            float gm[R];
            {
                const float4 P = tile[cur][g0];
#pragma unroll
                for (int r = 0; r < R; ++r) gm[r] = fmaf(m2x[r], P.x, fmaf(m2y[r], P.y, fmaf(m2z[r], P.z, P.w)));
            }
#pragma unroll
            for (int g = 1; g < FS_G; ++g) {
                const float4 P = tile[cur][g0 + g];
#pragma unroll
                for (int r = 0; r < R; ++r)
                    gm[r] = fminf(gm[r], fmaf(m2x[r], P.x, fmaf(m2y[r], P.y, fmaf(m2z[r], P.z, P.w))));
            }
            unsigned long long hit = 0ull;
#pragma unroll
            for (int r = 0; r < R; ++r) hit |= __builtin_amdgcn_ballot_w64(gm[r] < thr[r]);
            if (hit != 0ull) {
                const uint32_t gid = (tbase + (uint32_t)g0) / FS_G;
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (gm[r] < thr[r]) {
                        const long q = q0 + r * BLOCK + tid;
                        const uint32_t slot = atomicAdd(hit_cnt + q, 1u);
                        if (slot < cap) hit_list[(size_t)q * cap + slot] = gid;
                    }
            }
        }
        if (more) lstore(cur ^ 1);
        cur ^= 1;
    }
}

template <bool XFORM>
__global__ __launch_bounds__(256) void k_knn1_fixup(
    const double *__restrict__ qx, const double *__restrict__ qy, const double *__restrict__ qz, long Q,
    const double *__restrict__ px, const double *__restrict__ py, const double *__restrict__ pz, Xf H,
    const uint32_t *__restrict__ hit_cnt, const uint32_t *__restrict__ hit_list, uint32_t cap, uint32_t group /* points per
    recorded entry: FS_G for the VALU filter, 1 for the matrix-pipe filter */, double max_d2,
    int64_t idx_base, double *__restrict__ d2_out, int64_t *__restrict__ idx_out, double *__restrict__ p2_out,
    uint32_t *__restrict__ overflow)
{
    const int lane = threadIdx.x & 63;
    const long q = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (q >= Q) return;
    const uint32_t cnt = hit_cnt[q];
    const uint32_t n = cnt < cap ? cnt : cap;
    if (cnt > cap && lane == 0) atomicAdd(overflow, 1u);
    const double ax = qx[q], ay = qy[q], az = qz[q];
    double best = __builtin_inf();
    uint32_t bidx = 0xffffffffu;
    const uint32_t total = n * group;                      // candidates: `group` points per recorded entry
    for (uint32_t k = lane; k < total; k += 64) {
        const uint32_t id = hit_list[(size_t)q * cap + k / group] * group + k % group;
        double X = px[id], Y = py[id], Z = pz[id];
        if (XFORM) { double a, b, c; xform(H, X, Y, Z, a, b, c); X = a; Y = b; Z = c; }
        const double dx = X - ax, dy = Y - ay, dz = Z - az;
        const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
        if (d2 < best || (d2 == best && id < bidx)) { best = d2; bidx = id; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double od = __shfl_xor(best, off, 64);
        const uint32_t oi = __shfl_xor(bidx, off, 64);
        if (od < best || (od == best && oi < bidx)) { best = od; bidx = oi; }
    }
    if (lane == 0) {
        const bool ok = (bidx != 0xffffffffu) && (best < max_d2);
        d2_out[q] = ok ? best : __builtin_inf();
        idx_out[q] = ok ? idx_base + (int64_t)bidx : (int64_t)-1;
        if (p2_out) {
            p2_out[3 * q]     = ok ? px[bidx] : 0.0;
            p2_out[3 * q + 1] = ok ? py[bidx] : 0.0;
            p2_out[3 * q + 2] = ok ? pz[bidx] : 0.0;
        }
    }
}

// exact squared distance to the previous iteration's match under the NEW transform: an upper
// bound of the new nearest-neighbour distance that costs nothing (the point is in the cloud).
__global__ void k_bound_prev(const double *__restrict__ qx, const double *__restrict__ qy,
                             const double *__restrict__ qz, const double *__restrict__ p2, long Q, long qpad, Xf H,
                             double *__restrict__ bound)
{
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= qpad) return;
    if (q >= Q) { bound[q] = __builtin_inf(); return; }
    double X, Y, Z;
    xform(H, p2[3 * q], p2[3 * q + 1], p2[3 * q + 2], X, Y, Z);
    const double dx = X - qx[q], dy = Y - qy[q], dz = Z - qz[q];
    bound[q] = fma(dz, dz, fma(dy, dy, dx * dx));
}

// chunk partials -> per-query winner (ascending chunk order + strict `<` keeps the lowest
// index on ties), strict upper bound (pointcloud.py:163-167), gather of the winner's
// ORIGINAL coordinates (what the optimiser consumes, optimization.py:172-211).
__global__ __launch_bounds__(1024) void k_knn1_reduce(
    const double *__restrict__ part_d2, const uint32_t *__restrict__ part_idx, int nparts, int qpad, long Q,
    double max_d2, int64_t idx_base, const double *__restrict__ px, const double *__restrict__ py,
    const double *__restrict__ pz, double *__restrict__ d2_out, int64_t *__restrict__ idx_out,
    double *__restrict__ p2_out /* (Q,3) row-major, may be null */)
{
    // wave w of the block reads partial rows w, w+16, ... for 64 consecutive queries (coalesced),
    // then the 16 waves combine through LDS; all comparisons lexicographic on (d2, idx)
    __shared__ double sd[16][64];
    __shared__ uint32_t si[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long q = (long)blockIdx.x * 64 + tx;
    double best = __builtin_inf();
    uint32_t bi = 0xffffffffu;
    if (q < Q) {
        for (int c = ty; c < nparts; c += 16) {
            const double d = part_d2[(long)c * qpad + q];
            const uint32_t i = part_idx[(long)c * qpad + q];
            if (d < best || (d == best && i < bi)) { best = d; bi = i; }
        }
    }
    sd[ty][tx] = best; si[ty][tx] = bi;
    __syncthreads();
    if (ty != 0 || q >= Q) return;
    for (int w = 1; w < 16; ++w) {
        const double d = sd[w][tx];
        const uint32_t i = si[w][tx];
        if (d < best || (d == best && i < bi)) { best = d; bi = i; }
    }
    const bool ok = (bi != 0xffffffffu) && (best < max_d2);
    if (d2_out) d2_out[q] = ok ? best : __builtin_inf();
    if (idx_out) idx_out[q] = ok ? (idx_base + (int64_t)bi) : (int64_t)-1;
    if (p2_out) {
        p2_out[3 * q]     = ok ? px[bi] : 0.0;
        p2_out[3 * q + 1] = ok ? py[bi] : 0.0;
        p2_out[3 * q + 2] = ok ? pz[bi] : 0.0;
    }
}

// ------------------------------------------------------------------------------------
// K2: brute-force k-NN scan                       estimate_normals, pointcloud.py:185-186
//
// One query per lane, its K best (d2, idx) kept SORTED in registers (static indexing only:
// the bubble is fully unrolled).  The insertion branch is taken ~K*ln(n/K) times per lane,
// the scan itself is the same LDS-broadcast loop as K1.  A per-query lexicographic floor
// lets the host ask for neighbours K+1..2K in a second pass (k > 64).
// ------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(KNN_BLOCK) void k_knnk_scan(
    const double *__restrict__ qx, const double *__restrict__ qy, const double *__restrict__ qz, int qpad,
    const double *__restrict__ px, const double *__restrict__ py, const double *__restrict__ pz,
    long npad, int chunk_pts, const double *__restrict__ floor_d2, const uint32_t *__restrict__ floor_idx,
    double *__restrict__ part_d2, uint32_t *__restrict__ part_idx)
{
    __shared__ double sx[TILE_PTS], sy[TILE_PTS], sz[TILE_PTS];
    const int tid = threadIdx.x;
    const long q = (long)blockIdx.x * KNN_BLOCK + tid;
    const double ax = qx[q], ay = qy[q], az = qz[q];
    const bool has_floor = floor_d2 != nullptr;
    const double fd = has_floor ? floor_d2[q] : -1.0;
    const uint32_t fi = has_floor ? floor_idx[q] : 0u;

    double d[K];
    uint32_t ix[K];
#pragma unroll
    for (int s = 0; s < K; ++s) { d[s] = __builtin_inf(); ix[s] = 0xffffffffu; }

    const long c0 = (long)blockIdx.y * chunk_pts;
    long c1 = c0 + chunk_pts;
    if (c1 > npad) c1 = npad;

    for (long t0 = c0; t0 < c1; t0 += TILE_PTS) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < TILE_PTS / KNN_BLOCK; ++u) {
            const int s = u * KNN_BLOCK + tid;
            sx[s] = px[t0 + s]; sy[s] = py[t0 + s]; sz[s] = pz[t0 + s];
        }
        __syncthreads();
        const uint32_t base = (uint32_t)t0;
        for (int j = 0; j < TILE_PTS; ++j) {
            const double dx = sx[j] - ax, dy = sy[j] - ay, dz = sz[j] - az;
            const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
            const uint32_t id = base + (uint32_t)j;
            bool take = d2 < d[K - 1];
            if (has_floor) take = take && ((d2 > fd) || (d2 == fd && id > fi));
            if (take) {
                d[K - 1] = d2; ix[K - 1] = id;
#pragma unroll
                for (int s = K - 1; s > 0; --s) {
                    const bool sw = d[s] < d[s - 1];   // strict: equal d2 keeps the lower index first
                    const double lo = sw ? d[s] : d[s - 1], hi = sw ? d[s - 1] : d[s];
                    const uint32_t li = sw ? ix[s] : ix[s - 1], hi_i = sw ? ix[s - 1] : ix[s];
                    d[s - 1] = lo; d[s] = hi; ix[s - 1] = li; ix[s] = hi_i;
                }
            }
        }
    }
    const long o = ((long)blockIdx.y * qpad + q) * K;
#pragma unroll
    for (int s = 0; s < K; ++s) { part_d2[o + s] = d[s]; part_idx[o + s] = ix[s]; }
}

// merge the per-chunk sorted lists of one query; writes `kout` neighbours starting at
// column `col0` of the (Q, kstride) outputs, and the new floor for a following pass.
template <int K>
__global__ void k_knnk_merge(const double *__restrict__ part_d2, const uint32_t *__restrict__ part_idx,
                             int nchunks, int qpad, long Q, int kout, int col0, int kstride, int64_t idx_base,
                             double *__restrict__ d2_out, int64_t *__restrict__ idx_out,
                             double *__restrict__ floor_d2, uint32_t *__restrict__ floor_idx)
{
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    double d[K];
    uint32_t ix[K];
#pragma unroll
    for (int s = 0; s < K; ++s) { d[s] = __builtin_inf(); ix[s] = 0xffffffffu; }
    for (int c = 0; c < nchunks; ++c) {
        const long o = ((long)c * qpad + q) * K;
        for (int e = 0; e < K; ++e) {
            const double d2 = part_d2[o + e];
            if (!(d2 < d[K - 1])) break;          // lists are sorted: nothing further can enter
            d[K - 1] = d2; ix[K - 1] = part_idx[o + e];
#pragma unroll
            for (int s = K - 1; s > 0; --s) {
                const bool sw = d[s] < d[s - 1];
                const double lo = sw ? d[s] : d[s - 1], hi = sw ? d[s - 1] : d[s];
                const uint32_t li = sw ? ix[s] : ix[s - 1], hi_i = sw ? ix[s - 1] : ix[s];
                d[s - 1] = lo; d[s] = hi; ix[s - 1] = li; ix[s] = hi_i;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < K; ++s) {
        if (s < kout) {
            const bool ok = ix[s] != 0xffffffffu;
            if (d2_out) d2_out[q * kstride + col0 + s] = ok ? d[s] : __builtin_inf();
            idx_out[q * kstride + col0 + s] = ok ? idx_base + (int64_t)ix[s] : (int64_t)-1;
        }
    }
    if (floor_d2) {
        // last emitted neighbour becomes the exclusive lower bound of the next pass
        double fd = -1.0; uint32_t fi = 0;
#pragma unroll
        for (int s = 0; s < K; ++s) if (s == kout - 1) { fd = d[s]; fi = ix[s]; }
        floor_d2[q] = fd; floor_idx[q] = fi;
    }
}

// ------------------------------------------------------------------------------------
// K3: covariance + symmetric 3x3 eigen-decomposition     pointcloud.py:188-198
// one query per lane, fp64, same operation order as oracle/sicp_oracle.c:orc_normals.
// ------------------------------------------------------------------------------------
__global__ void k_normals(const double *__restrict__ px, const double *__restrict__ py,
                          const double *__restrict__ pz, const int64_t *__restrict__ nn, long Q, int k,
                          int64_t idx_base, float *__restrict__ normals, float *__restrict__ planarity)
{
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    const int64_t *row = nn + q * k;
    double m[3] = {0, 0, 0};
    for (int s = 0; s < k; ++s) {
        const int64_t i = row[s] - idx_base;
        m[0] += px[i]; m[1] += py[i]; m[2] += pz[i];
    }
    m[0] /= (double)k; m[1] /= (double)k; m[2] /= (double)k;
    double C[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int s = 0; s < k; ++s) {
        const int64_t i = row[s] - idx_base;
        const double d0 = px[i] - m[0], d1 = py[i] - m[1], d2 = pz[i] - m[2];
        C[0][0] = fma(d0, d0, C[0][0]); C[0][1] = fma(d0, d1, C[0][1]); C[0][2] = fma(d0, d2, C[0][2]);
        C[1][1] = fma(d1, d1, C[1][1]); C[1][2] = fma(d1, d2, C[1][2]); C[2][2] = fma(d2, d2, C[2][2]);
    }
    const double inv = 1.0 / (double)(k - 1);
    const double c6[6] = {C[0][0] * inv, C[0][1] * inv, C[0][2] * inv, C[1][1] * inv, C[1][2] * inv, C[2][2] * inv};
    float nrm[3], pl;
    normal_from_cov(c6, nrm, &pl);
    normals[3 * q] = nrm[0]; normals[3 * q + 1] = nrm[1]; normals[3 * q + 2] = nrm[2];
    planarity[q] = pl;
}

// ------------------------------------------------------------------------------------
// K5: point-to-plane distances of the fresh matches + planarity flag
//     corrpts.py:195-211, corrpts.py:139-163
// flag: 1 = passes the planarity test (float32 compare, NaN fails) and has a match
// ------------------------------------------------------------------------------------
__global__ void k_postmatch(const double *__restrict__ qx, const double *__restrict__ qy,
                            const double *__restrict__ qz, const float *__restrict__ normals,
                            const float *__restrict__ planarity, const double *__restrict__ p2,
                            const int64_t *__restrict__ idx, long Q, Xf H, float min_planarity,
                            const float *__restrict__ pl2 /* movable cloud's column by global index, or null */,
                            long pl2_n, double *__restrict__ dist, uint8_t *__restrict__ flag,
                            const IcpDev *__restrict__ st /* nullable: chained run -> its H, its stop flag */)
{
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (st) { if (st->stop) return; H = st->H; }
    if (q >= Q) return;
    double X, Y, Z;
    xform(H, p2[3 * q], p2[3 * q + 1], p2[3 * q + 2], X, Y, Z);
    dist[q] = plane_dist(X - qx[q], Y - qy[q], Z - qz[q], normals[3 * q], normals[3 * q + 1], normals[3 * q + 2]);
    const int64_t m = idx[q];
    bool f = m >= 0 && planarity[q] >= min_planarity;
    if (f && pl2) f = m < pl2_n && pl2[m] >= min_planarity;          // corrpts.py:158-163 (NaN fails)
    flag[q] = f ? 1 : 0;
}

// CorrPts.reject_wrt_planarity as an operator of its own (corrpts.py:139-163): a correspondence stays alive when the
// planarity of its point in pc1 -- and in pc2, iff that cloud has the column -- reaches the threshold (NaN fails).
// pl1 / pl2 are per CORRESPONDENCE (the reference's `pc.iloc[idx]["planarity"]`); a null column is not tested.
__global__ void k_corr_planarity(uint8_t *__restrict__ alive, const float *__restrict__ pl1, const float *__restrict__ pl2,
                                 float min_planarity, long Q)
{
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    bool f = alive[q] != 0;
    if (f && pl1) f = pl1[q] >= min_planarity;
    if (f && pl2) f = pl2[q] >= min_planarity;
    alive[q] = f ? 1 : 0;
}

__global__ void k_fill_f32(float *__restrict__ dst, long n, float v)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = v;
}
__global__ void k_scatter_f32(float *__restrict__ dst, const int64_t *__restrict__ rows, const float *__restrict__ vals, long m)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) dst[rows[i]] = vals[i];
}

// (K6, the median / raw-MAD rejection by one workgroup -- corrpts.py:165-188 --, lives in sicp_reject.hip)

// ------------------------------------------------------------------------------------
// masked mean / population std (two-pass, np.std ddof=0)     simpleicp.py:233-234,356-379
// single workgroup; out[0]=n out[1]=mean out[2]=std
// ------------------------------------------------------------------------------------
// host_out (nullable, pinned + mapped): [0..3] = also4 (the rejection's m / median / mad / n_kept, written by the
// launch before on this stream), [4..6] = n / mean / std, then the completion ticket [15] = seq -- the host polls it
// instead of paying a copy + stream synchronisation.
__global__ __launch_bounds__(1024) void k_stats(const double *__restrict__ v, const uint8_t *__restrict__ keep,
                                                long Q, double *__restrict__ out, const double *__restrict__ also4,
                                                double *__restrict__ host_out, double seq, const IcpDev *__restrict__ st)
{
    __shared__ double red[16];
    __shared__ double bc[2];
    if (st && st->stop) return;
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    double s = 0, n = 0;
    for (long i = tid; i < Q; i += blockDim.x) if (keep[i]) { s += v[i]; n += 1; }
    s = wave_sum(s); n = wave_sum(n);
    if (lane == 0) red[wid] = s;
    __syncthreads();
    if (tid == 0) { double t = 0; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w]; bc[0] = t; }
    __syncthreads();
    if (lane == 0) red[wid] = n;
    __syncthreads();
    if (tid == 0) { double t = 0; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w]; bc[1] = t; }
    __syncthreads();
    const double cnt = bc[1];
    const double mean = bc[0] / cnt;
    double ss = 0;
    for (long i = tid; i < Q; i += blockDim.x) if (keep[i]) { const double e = v[i] - mean; ss += e * e; }
    ss = wave_sum(ss);
    __syncthreads();
    if (lane == 0) red[wid] = ss;
    __syncthreads();
    if (tid == 0) {
        double t = 0; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
        const double sd = sqrt(t / cnt);
        out[0] = cnt; out[1] = mean; out[2] = sd;
        if (host_out) {
            if (also4) { host_out[0] = also4[0]; host_out[1] = also4[1]; host_out[2] = also4[2]; host_out[3] = also4[3]; }
            host_out[4] = cnt; host_out[5] = mean; host_out[6] = sd;
            __threadfence_system();
            __hip_atomic_store(host_out + 15, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// The same statistics over many workgroups (Q in the 10^5..10^6 range, where one CU would spend hundreds of
// microseconds streaming the vector twice): PHASE 0 = count and mean, PHASE 1 = deviation from that mean
// (two-pass like np.std).  Block partials are folded by the last block to arrive (agent-scope ticket, as in
// k_normal_eq), lanes striding over the blocks + a butterfly: a fixed order for a given grid.
template <int PHASE>
__global__ __launch_bounds__(256) void k_stats_mb(const double *__restrict__ v, const uint8_t *__restrict__ keep, long Q,
                                                  double *__restrict__ out, double *__restrict__ partial /*[2][NE_MAX_GRID]*/,
                                                  unsigned *__restrict__ ticket, const double *__restrict__ also4,
                                                  double *__restrict__ host_out, double seq, const IcpDev *__restrict__ st)
{
    __shared__ double red[4][2];
    __shared__ int is_last;
    if (st && st->stop) return;
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    const double mean = PHASE == 1 ? out[1] : 0.0;
    double a = 0, b = 0;
    for (long i = (long)blockIdx.x * 256 + tid; i < Q; i += (long)gridDim.x * 256)
        if (keep[i]) {
            if (PHASE == 0) { a += v[i]; b += 1; }
            else { const double e = v[i] - mean; a += e * e; }
        }
    a = wsum(a); b = wsum(b);
    if (lane == 0) { red[wid][0] = a; red[wid][1] = b; }
    __syncthreads();
    if (tid < 2) partial[(long)tid * NE_MAX_GRID + blockIdx.x] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (t == gridDim.x - 1) ? 1 : 0;
        if (is_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!is_last) return;
    if (wid < 2) {
        double s = 0;
        for (unsigned blk = lane; blk < gridDim.x; blk += 64) s += partial[(long)wid * NE_MAX_GRID + blk];
        s = wsum(s);
        if (lane == 0) red[0][wid] = s;
    }
    __syncthreads();
    if (tid == 0) {
        *ticket = 0;
        if (PHASE == 0) { out[0] = red[0][1]; out[1] = red[0][0] / red[0][1]; }
        else {
            const double cnt = out[0], sd = sqrt(red[0][0] / cnt);
            out[2] = sd;
            if (host_out) {
                if (also4) { host_out[0] = also4[0]; host_out[1] = also4[1]; host_out[2] = also4[2]; host_out[3] = also4[3]; }
                host_out[4] = cnt; host_out[5] = out[1]; host_out[6] = sd;
                __threadfence_system();
                __hip_atomic_store(host_out + 15, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// K7: fused residual + rejection mask + 6x6 normal-equation reduction
//     optimization.py:172-288 (residual), c++/src/corrpts.cpp:110-156 (row layout twin)
//
// Per kept correspondence: p = R(x) p2 + t (contract T), r = n.(p - p1) (contract P),
// J = [n.(dR1 p2), n.(dR2 p2), n.(dR3 p2), n];  accumulates the 21 upper-triangle
// entries of J^T J, the 6 of J^T r, sum r, sum r^2, n -- 30 doubles -- in registers,
// wave reduction by DPP/shuffle, one LDS hop per wave, block partials to global; the
// LAST block (agent-scope release/acquire ticket) folds the partials in fixed order, so
// the result is deterministic for a given grid.  72 B and ~75 flop per correspondence:
// HBM/latency-bound by construction.
// Optionally writes the residual vector (0 where masked).
// ------------------------------------------------------------------------------------
struct NeArgs {
    Xf H;            // R(x) | t
    double dR[27];   // dR/dalpha1, dR/dalpha2, dR/dalpha3 (row-major 3x3 each)
};

__global__ __launch_bounds__(NE_BLOCK) void k_normal_eq(
    const double *__restrict__ qx, const double *__restrict__ qy, const double *__restrict__ qz,
    const float *__restrict__ normals, const double *__restrict__ p2, const uint8_t *__restrict__ keep,
    long lo, long hi, NeArgs A, double *__restrict__ partial /*[30][NE_MAX_GRID]*/, unsigned *__restrict__ ticket,
    double *__restrict__ out /*[30]*/, double *__restrict__ resid /* nullable, (Q) */,
    double *__restrict__ host_out /* nullable, pinned + mapped: [0..29] = out, ticket [31] = seq */, double seq)
{
    __shared__ double red[NE_BLOCK / 64][32];
    __shared__ int is_last;
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    double acc[30];
#pragma unroll
    for (int i = 0; i < 30; ++i) acc[i] = 0.0;

    for (long i = lo + (long)blockIdx.x * NE_BLOCK + tid; i < hi; i += (long)gridDim.x * NE_BLOCK) {
        const bool k = keep[i] != 0;
        double r = 0.0;
        if (k) {
            const double x = p2[3 * i], y = p2[3 * i + 1], z = p2[3 * i + 2];
            double X, Y, Z;
            xform(A.H, x, y, z, X, Y, Z);
            const float fx = normals[3 * i], fy = normals[3 * i + 1], fz = normals[3 * i + 2];
            r = plane_dist(X - qx[i], Y - qy[i], Z - qz[i], fx, fy, fz);
            const double nx = fx, ny = fy, nz = fz;
            double a[6];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double *D = A.dR + 9 * c;
                const double gx = D[0] * x + D[1] * y + D[2] * z;
                const double gy = D[3] * x + D[4] * y + D[5] * z;
                const double gz = D[6] * x + D[7] * y + D[8] * z;
                a[c] = nx * gx + ny * gy + nz * gz;
            }
            a[3] = nx; a[4] = ny; a[5] = nz;
            int t = 0;
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int v = u; v < 6; ++v) { acc[t] += a[u] * a[v]; ++t; }
#pragma unroll
            for (int u = 0; u < 6; ++u) acc[21 + u] += a[u] * r;
            acc[27] += r; acc[28] += r * r; acc[29] += 1.0;
        }
        if (resid) resid[i] = r;
    }
#pragma unroll
    for (int i = 0; i < 30; ++i) {
        const double s = wave_sum(acc[i]);
        if (lane == 0) red[wid][i] = s;
    }
    __syncthreads();
    if (tid < 30) {
        double s = 0;
#pragma unroll
        for (int w = 0; w < NE_BLOCK / 64; ++w) s += red[w][tid];
        partial[(long)tid * NE_MAX_GRID + blockIdx.x] = s;       // column-major: the fold below reads it coalesced
    }
    // publish this block's partial, then take a ticket (guide G16: stores -> vmcnt(0) ->
    // barrier -> one-lane agent release -> ticket; last arriver: agent acquire -> plain loads)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (t == gridDim.x - 1) ? 1 : 0;
        if (is_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (is_last) {
        // fold the block partials: wave w takes columns w, w+4, ...; its lanes stride over the blocks and a
        // butterfly adds them up -- a fixed order for a given grid, and no 1024-long dependent chain
        for (int col = wid; col < 30; col += NE_BLOCK / 64) {
            double s = 0;
            for (unsigned b = lane; b < gridDim.x; b += 64) s += partial[(long)col * NE_MAX_GRID + b];
            s = wsum(s);
            if (lane == 0) { out[col] = s; if (host_out) host_out[col] = s; }
        }
        if (tid == 0) *ticket = 0;   // re-arm for the next launch on this stream
        if (host_out) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                     // is_last is block-uniform
            if (tid == 0) {
                __threadfence_system();
                __hip_atomic_store(host_out + 31, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// multi-GPU exchange helpers: per-query records (d2, idx bits, x, y, z) <-> job-wide winner
// ------------------------------------------------------------------------------------
__global__ void k_pack_best(const double *__restrict__ d2, const int64_t *__restrict__ idx,
                            const double *__restrict__ p2, long Q, double *__restrict__ rec)
{
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    rec[5 * q] = d2[q];
    rec[5 * q + 1] = __longlong_as_double((long long)idx[q]);
    rec[5 * q + 2] = p2 ? p2[3 * q] : 0.0; rec[5 * q + 3] = p2 ? p2[3 * q + 1] : 0.0; rec[5 * q + 4] = p2 ? p2[3 * q + 2] : 0.0;
}

__global__ void k_lexmin_gathered(const double *__restrict__ g, int world, long Q, double *__restrict__ d2,
                                  int64_t *__restrict__ idx, double *__restrict__ p2)
{
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    double bd = __builtin_inf(), bx = 0, by = 0, bz = 0;
    int64_t bi = -1;
    for (int r = 0; r < world; ++r) {
        const double *rec = g + ((long)r * Q + q) * 5;
        const double d = rec[0];
        const int64_t i = (int64_t)__double_as_longlong(rec[1]);
        if (i >= 0 && (bi < 0 || d < bd || (d == bd && i < bi))) { bd = d; bi = i; bx = rec[2]; by = rec[3]; bz = rec[4]; }
    }
    d2[q] = bi >= 0 ? bd : __builtin_inf();
    idx[q] = bi;
    if (p2) { p2[3 * q] = bx; p2[3 * q + 1] = by; p2[3 * q + 2] = bz; }
}

// the two in one launch for a chained iteration behind a cloud-shard exchange: job-wide winner per query (the single-GPU rule:
// lexicographic minimum on (d2, global index) over the ranks' records), then what k_postmatch does with it -- signed
// point-to-plane distance under the run's current H and the planarity verdict (corrpts.py:139-163,195-211)
__global__ void k_lexmin_postmatch(const double *__restrict__ g, int world, long Q, const double *__restrict__ qx,
                                   const double *__restrict__ qy, const double *__restrict__ qz, const float *__restrict__ normals,
                                   const float *__restrict__ planarity, float min_planarity, const float *__restrict__ pl2, long pl2_n,
                                   const IcpDev *__restrict__ st, double *__restrict__ d2, int64_t *__restrict__ idx,
                                   double *__restrict__ p2, double *__restrict__ dist, uint8_t *__restrict__ flag)
{
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (st->stop) return;
    if (q >= Q) return;
    const Xf H = st->H;
    double bd = __builtin_inf(), bx = 0, by = 0, bz = 0;
    int64_t bi = -1;
    for (int r = 0; r < world; ++r) {
        const double *rec = g + ((long)r * Q + q) * 5;
        const double d = rec[0];
        const int64_t i = (int64_t)__double_as_longlong(rec[1]);
        if (i >= 0 && (bi < 0 || d < bd || (d == bd && i < bi))) { bd = d; bi = i; bx = rec[2]; by = rec[3]; bz = rec[4]; }
    }
    d2[q] = bi >= 0 ? bd : __builtin_inf();
    idx[q] = bi;
    p2[3 * q] = bx; p2[3 * q + 1] = by; p2[3 * q + 2] = bz;
    double X, Y, Z;
    xform(H, bx, by, bz, X, Y, Z);
    dist[q] = plane_dist(X - qx[q], Y - qy[q], Z - qz[q], normals[3 * q], normals[3 * q + 1], normals[3 * q + 2]);
    bool f = bi >= 0 && planarity[q] >= min_planarity;
    if (f && pl2) f = bi < pl2_n && pl2[bi] >= min_planarity;        // corrpts.py:158-163 (NaN fails)
    flag[q] = f ? 1 : 0;
}

// ---- query shards (every rank holds the WHOLE searched cloud): the exchange carries nothing but the matched index ----
// 8 bytes per query instead of the 40-byte (d2, idx, xyz) record: the coordinates are this rank's own to look up, the squared
// distance is not used after the match, and the point-to-plane distance is formed from the looked-up point anyway.
__global__ void k_pack_idx(const int64_t *__restrict__ idx, long cnt, long per, double *__restrict__ out)
{
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < per) out[j] = __longlong_as_double(j < cnt ? (long long)idx[j] : -1ll);
}
// gathered[q] = bits of query q's matched global index (slices in rank order = query order).  Writes the index, looks the
// point up in the cloud, and does k_postmatch's work on it (distance under the chained run's H, planarity verdict).
__global__ void k_unpack_idx_postmatch(const double *__restrict__ gathered, long Q, const double *__restrict__ cx,
                                       const double *__restrict__ cy, const double *__restrict__ cz, int64_t idx_base, long n,
                                       const double *__restrict__ qx, const double *__restrict__ qy, const double *__restrict__ qz,
                                       const float *__restrict__ normals, const float *__restrict__ planarity, float min_planarity,
                                       const float *__restrict__ pl2, long pl2_n, const IcpDev *__restrict__ st,
                                       int64_t *__restrict__ idx, double *__restrict__ p2, double *__restrict__ dist,
                                       uint8_t *__restrict__ flag)
{
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (st->stop || q >= Q) return;
    const int64_t m = (int64_t)__double_as_longlong(gathered[q]);
    const long row = (long)(m - idx_base);
    const bool ok = m >= 0 && row >= 0 && row < n;
    const double px = ok ? cx[row] : 0.0, py = ok ? cy[row] : 0.0, pz = ok ? cz[row] : 0.0;
    idx[q] = ok ? m : (int64_t)-1;
    p2[3 * q] = px; p2[3 * q + 1] = py; p2[3 * q + 2] = pz;
    const Xf H = st->H;
    double X, Y, Z;
    xform(H, px, py, pz, X, Y, Z);
    dist[q] = plane_dist(X - qx[q], Y - qy[q], Z - qz[q], normals[3 * q], normals[3 * q + 1], normals[3 * q + 2]);
    bool f = ok && planarity[q] >= min_planarity;
    if (f && pl2) f = m < pl2_n && pl2[m] >= min_planarity;          // corrpts.py:158-163 (NaN fails)
    flag[q] = f ? 1 : 0;
}

// ------------------------------------------------------------------------------------
// launchers (host)
// ------------------------------------------------------------------------------------
static inline unsigned cdiv(long a, long b) { return (unsigned)((a + b - 1) / b); }

// ---- the same winner by three all-reduces on 8-byte keys (cloud shards, many queries) ------------------------------
// An all-gather hands every rank 40 bytes per query and RANK; a ring all-reduce moves ~2 x the vector whatever the rank count:
//   1. min over the ranks of the squared distance's bit pattern (non-negative doubles order like their bits; ~0 = no match here);
//   2. min over the ranks of the matched index, offered only by the ranks whose distance IS that minimum (~0 otherwise): together the
//      lexicographic (d2, index) minimum -- indices are global and the shards disjoint, so exactly one rank owns the winner;
//   3. max over the ranks of the winner's coordinates as BIT PATTERNS, the owner's against zeros: bit-exact (a sum would turn -0.0
//      into +0.0).
// 8 + 8 + 24 bytes per query, three collectives; below ~32 768 queries the single all-gather's latency wins.
__global__ void k_xkey_d2(const double *__restrict__ d2, const int64_t *__restrict__ idx, long Q, unsigned long long *__restrict__ key)
{
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    key[q] = idx[q] >= 0 ? (unsigned long long)__double_as_longlong(d2[q]) : ~0ull;
}
__global__ void k_xkey_idx(const double *__restrict__ d2, const int64_t *__restrict__ idx, const unsigned long long *__restrict__ gmin,
                           long Q, unsigned long long *__restrict__ key)
{
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    const bool mine = idx[q] >= 0 && (unsigned long long)__double_as_longlong(d2[q]) == gmin[q];
    key[q] = mine ? (unsigned long long)idx[q] : ~0ull;
}
__global__ void k_xkey_xyz(const int64_t *__restrict__ idx, const double *__restrict__ p2, const unsigned long long *__restrict__ gidx,
                           long Q, unsigned long long *__restrict__ xyz)
{
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    // (the index alone names the owner: a rank whose local winner has the winning index holds the winning point)
    const bool own = idx[q] >= 0 && (unsigned long long)idx[q] == gidx[q];
#pragma unroll
    for (int a = 0; a < 3; ++a) xyz[3 * q + a] = own ? (unsigned long long)__double_as_longlong(p2[3 * q + a]) : 0ull;
}
__global__ void k_xkey_unpack(const unsigned long long *__restrict__ gmin, const unsigned long long *__restrict__ gidx,
                              const unsigned long long *__restrict__ xyz, long Q, double *__restrict__ d2, int64_t *__restrict__ idx,
                              double *__restrict__ p2)
{
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    const bool any = gidx[q] != ~0ull;
    d2[q] = any ? __longlong_as_double((long long)gmin[q]) : __builtin_inf();
    idx[q] = any ? (int64_t)gidx[q] : (int64_t)-1;
#pragma unroll
    for (int a = 0; a < 3; ++a) p2[3 * q + a] = any ? __longlong_as_double((long long)xyz[3 * q + a]) : 0.0;
}
void launch_xkey_d2(hipStream_t s, const double *d2, const int64_t *idx, long Q, unsigned long long *key)
{
    hipLaunchKernelGGL(k_xkey_d2, dim3(cdiv(Q, 256)), dim3(256), 0, s, d2, idx, Q, key);
}
void launch_xkey_idx(hipStream_t s, const double *d2, const int64_t *idx, const unsigned long long *gmin, long Q, unsigned long long *key)
{
    hipLaunchKernelGGL(k_xkey_idx, dim3(cdiv(Q, 256)), dim3(256), 0, s, d2, idx, gmin, Q, key);
}
void launch_xkey_xyz(hipStream_t s, const int64_t *idx, const double *p2, const unsigned long long *gidx, long Q, unsigned long long *xyz)
{
    hipLaunchKernelGGL(k_xkey_xyz, dim3(cdiv(Q, 256)), dim3(256), 0, s, idx, p2, gidx, Q, xyz);
}
void launch_xkey_unpack(hipStream_t s, const unsigned long long *gmin, const unsigned long long *gidx, const unsigned long long *xyz,
                        long Q, double *d2, int64_t *idx, double *p2)
{
    hipLaunchKernelGGL(k_xkey_unpack, dim3(cdiv(Q, 256)), dim3(256), 0, s, gmin, gidx, xyz, Q, d2, idx, p2);
}

void launch_pack_best(hipStream_t s, const double *d2, const int64_t *idx, const double *p2, long Q, double *rec)
{
    hipLaunchKernelGGL(k_pack_best, dim3(cdiv(Q, 256)), dim3(256), 0, s, d2, idx, p2, Q, rec);
}
void launch_pack_idx(hipStream_t s, const int64_t *idx, long cnt, long per, double *out)
{
    if (per > 0) hipLaunchKernelGGL(k_pack_idx, dim3(cdiv(per, 256)), dim3(256), 0, s, idx, cnt, per, out);
}
void launch_unpack_idx_postmatch(hipStream_t s, const double *gathered, long Q, const double *cx, const double *cy, const double *cz,
                                 int64_t idx_base, long n, const double *qx, const double *qy, const double *qz, const float *normals,
                                 const float *planarity, float min_planarity, const float *pl2, long pl2_n, const IcpDev *st,
                                 int64_t *idx, double *p2, double *dist, uint8_t *flag)
{
    hipLaunchKernelGGL(k_unpack_idx_postmatch, dim3(cdiv(Q, 256)), dim3(256), 0, s, gathered, Q, cx, cy, cz, idx_base, n, qx, qy, qz,
                       normals, planarity, min_planarity, pl2, pl2_n, st, idx, p2, dist, flag);
}
void launch_lexmin_postmatch(hipStream_t s, const double *g, int world, long Q, const double *qx, const double *qy, const double *qz,
                             const float *normals, const float *planarity, float min_planarity, const float *pl2, long pl2_n,
                             const IcpDev *st, double *d2, int64_t *idx, double *p2, double *dist, uint8_t *flag)
{
    hipLaunchKernelGGL(k_lexmin_postmatch, dim3(cdiv(Q, 256)), dim3(256), 0, s, g, world, Q, qx, qy, qz, normals, planarity,
                       min_planarity, pl2, pl2_n, st, d2, idx, p2, dist, flag);
}
void launch_lexmin_gathered(hipStream_t s, const double *g, int world, long Q, double *d2, int64_t *idx, double *p2)
{
    hipLaunchKernelGGL(k_lexmin_gathered, dim3(cdiv(Q, 256)), dim3(256), 0, s, g, world, Q, d2, idx, p2);
}

void launch_aos_to_soa(hipStream_t s, const double *aos, long n, long npad, double *x, double *y, double *z)
{
    hipLaunchKernelGGL(k_aos_to_soa, dim3(cdiv(npad, 256)), dim3(256), 0, s, aos, n, npad, x, y, z);
}
void launch_pad_fill(hipStream_t s, double *x, double *y, double *z, long n, long npad)
{
    if (npad > n) hipLaunchKernelGGL(k_pad_fill, dim3(cdiv(npad - n, 256)), dim3(256), 0, s, x, y, z, n, npad);
}
void launch_soa_to_aos(hipStream_t s, const double *x, const double *y, const double *z, long n, double *aos)
{
    if (n > 0) hipLaunchKernelGGL(k_soa_to_aos, dim3(cdiv(n, 256)), dim3(256), 0, s, x, y, z, n, aos);
}
void launch_pack_chunks(hipStream_t s, const double *x, const double *y, const double *z, long n, long CH, double *out)
{
    if (n > 0) hipLaunchKernelGGL(k_pack_chunks, dim3(cdiv(n, 256)), dim3(256), 0, s, x, y, z, n, CH, out);
}
void launch_transform(hipStream_t s, double *x, double *y, double *z, long n, const Xf &H)
{
    if (n > 0) hipLaunchKernelGGL(k_transform, dim3(cdiv(n, 256)), dim3(256), 0, s, x, y, z, n, H);
}
void launch_gather_queries(hipStream_t s, const double *x, const double *y, const double *z, const int64_t *sel,
                           long Q, long qpad, double *qx, double *qy, double *qz)
{
    hipLaunchKernelGGL(k_gather_queries, dim3(cdiv(qpad, 256)), dim3(256), 0, s, x, y, z, sel, Q, qpad, qx, qy, qz);
}
void launch_found_mask(hipStream_t s, const int64_t *idx, long Q, uint8_t *out)
{
    if (Q > 0) hipLaunchKernelGGL(k_found_mask, dim3(cdiv(Q, 256)), dim3(256), 0, s, idx, Q, out);
}
void launch_aos_queries(hipStream_t s, const double *aos, long Q, long qpad, double *qx, double *qy, double *qz)
{
    hipLaunchKernelGGL(k_aos_queries, dim3(cdiv(qpad, 256)), dim3(256), 0, s, aos, Q, qpad, qx, qy, qz);
}

void launch_knn1_scan(hipStream_t s, const double *qx, const double *qy, const double *qz, int qpad, int qblocks,
                      const double *px, const double *py, const double *pz, long npad, int tile_step,
                      int tiles_per_chunk, int nchunks, const Xf *H, double *part_d2, uint32_t *part_idx)
{
    const dim3 grid(qblocks, nchunks), block(KNN_BLOCK);
    Xf id = {};
    if (H)
        hipLaunchKernelGGL((k_knn1_scan<KNN1_R, true>), grid, block, 0, s, qx, qy, qz, qpad, px, py, pz, npad,
                           tile_step, tiles_per_chunk, *H, part_d2, part_idx);
    else
        hipLaunchKernelGGL((k_knn1_scan<KNN1_R, false>), grid, block, 0, s, qx, qy, qz, qpad, px, py, pz, npad,
                           tile_step, tiles_per_chunk, id, part_d2, part_idx);
}

template <int BLOCK>
static void fscan_launch(hipStream_t s, const double *qx, const double *qy, const double *qz, int qpad, long Q, int qblocks,
                         const double *bound, const double *px, const double *py, const double *pz, int ntiles,
                         int nparts, const Xf *H, double rmax, double *part_d2, uint32_t *part_idx)
{
    const dim3 grid(qblocks, nparts), block(BLOCK);
    Xf id = {};
    if (H)
        hipLaunchKernelGGL((k_knn1_fscan<BLOCK, true>), grid, block, 0, s, qx, qy, qz, qpad, Q, bound, px, py, pz, ntiles,
                           *H, rmax, part_d2, part_idx);
    else
        hipLaunchKernelGGL((k_knn1_fscan<BLOCK, false>), grid, block, 0, s, qx, qy, qz, qpad, Q, bound, px, py, pz, ntiles,
                           id, rmax, part_d2, part_idx);
}

// BLOCK = 128 (1024 queries per block) or 256 (2048 queries per block)
void launch_knn1_fscan(hipStream_t s, int block, const double *qx, const double *qy, const double *qz, int qpad, long Q,
                       int qblocks, const double *bound, const double *px, const double *py, const double *pz,
                       int ntiles, int nparts, const Xf *H, double rmax, double *part_d2, uint32_t *part_idx)
{
    if (block == 256) fscan_launch<256>(s, qx, qy, qz, qpad, Q, qblocks, bound, px, py, pz, ntiles, nparts, H, rmax, part_d2, part_idx);
    else              fscan_launch<128>(s, qx, qy, qz, qpad, Q, qblocks, bound, px, py, pz, ntiles, nparts, H, rmax, part_d2, part_idx);
}

template <int BLOCK>
static void frec_launch(hipStream_t s, const double *qx, const double *qy, const double *qz, long Q, int qblocks,
                        const double *bound, const double *px, const double *py, const double *pz, int ntiles,
                        int nparts, const Xf *H, double rmax, uint32_t *hit_cnt, uint32_t *hit_list, uint32_t cap)
{
    const dim3 grid(qblocks, nparts), block(BLOCK);
    Xf id = {};
    if (H)
        hipLaunchKernelGGL((k_knn1_frec<BLOCK, true>), grid, block, 0, s, qx, qy, qz, Q, bound, px, py, pz, ntiles, *H, rmax,
                           hit_cnt, hit_list, cap);
    else
        hipLaunchKernelGGL((k_knn1_frec<BLOCK, false>), grid, block, 0, s, qx, qy, qz, Q, bound, px, py, pz, ntiles, id, rmax,
                           hit_cnt, hit_list, cap);
}

void launch_knn1_frec(hipStream_t s, int block, const double *qx, const double *qy, const double *qz, long Q, int qblocks,
                      const double *bound, const double *px, const double *py, const double *pz, int ntiles, int nparts,
                      const Xf *H, double rmax, uint32_t *hit_cnt, uint32_t *hit_list, uint32_t cap)
{
    if (block == 256) frec_launch<256>(s, qx, qy, qz, Q, qblocks, bound, px, py, pz, ntiles, nparts, H, rmax, hit_cnt, hit_list, cap);
    else              frec_launch<128>(s, qx, qy, qz, Q, qblocks, bound, px, py, pz, ntiles, nparts, H, rmax, hit_cnt, hit_list, cap);
}

void launch_knn1_fixup(hipStream_t s, const double *qx, const double *qy, const double *qz, long Q, const double *px,
                       const double *py, const double *pz, const Xf *H, const uint32_t *hit_cnt, const uint32_t *hit_list,
                       uint32_t cap, uint32_t group, double max_d2, int64_t idx_base, double *d2_out, int64_t *idx_out,
                       double *p2_out, uint32_t *overflow)
{
    const dim3 grid(cdiv(Q, 4)), block(256);
    Xf id = {};
    if (H)
        hipLaunchKernelGGL((k_knn1_fixup<true>), grid, block, 0, s, qx, qy, qz, Q, px, py, pz, *H, hit_cnt, hit_list, cap, group,
                           max_d2, idx_base, d2_out, idx_out, p2_out, overflow);
    else
        hipLaunchKernelGGL((k_knn1_fixup<false>), grid, block, 0, s, qx, qy, qz, Q, px, py, pz, id, hit_cnt, hit_list, cap, group,
                           max_d2, idx_base, d2_out, idx_out, p2_out, overflow);
}


int frec_blocks_per_cu(int block)
{
    int nb = 0;
    hipError_t e = (block == 256)
        ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_knn1_frec<256, true>, 256, 0)
        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_knn1_frec<128, true>, 128, 0);
    if (e != hipSuccess || nb < 1) nb = 2;
    return nb > 16 ? 16 : nb;
}

int fscan_blocks_per_cu(int block)
{
    int nb = 0;
    hipError_t e = (block == 256)
        ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_knn1_fscan<256, true>, 256, 0)
        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_knn1_fscan<128, true>, 128, 0);
    if (e != hipSuccess || nb < 1) nb = 2;
    return nb > 16 ? 16 : nb;
}

void launch_bound_prev(hipStream_t s, const double *qx, const double *qy, const double *qz, const double *p2, long Q,
                       long qpad, const Xf &H, double *bound)
{
    hipLaunchKernelGGL(k_bound_prev, dim3(cdiv(qpad, 256)), dim3(256), 0, s, qx, qy, qz, p2, Q, qpad, H, bound);
}

void launch_knn1_reduce(hipStream_t s, const double *part_d2, const uint32_t *part_idx, int nparts, int qpad, long Q,
                        double max_d2, int64_t idx_base, const double *px, const double *py, const double *pz,
                        double *d2_out, int64_t *idx_out, double *p2_out)
{
    hipLaunchKernelGGL(k_knn1_reduce, dim3(cdiv(Q, 64)), dim3(1024), 0, s, part_d2, part_idx, nparts, qpad, Q, max_d2,
                       idx_base, px, py, pz, d2_out, idx_out, p2_out);
}

template <int K>
static void knnk_pass(hipStream_t s, const double *qx, const double *qy, const double *qz, int qpad, long Q,
                      const double *px, const double *py, const double *pz, long npad, int chunk_pts, int nchunks,
                      const double *floor_d2_in, const uint32_t *floor_idx_in, double *part_d2, uint32_t *part_idx,
                      int kout, int col0, int kstride, int64_t idx_base, double *d2_out, int64_t *idx_out,
                      double *floor_d2_out, uint32_t *floor_idx_out)
{
    hipLaunchKernelGGL((k_knnk_scan<K>), dim3(cdiv(Q, KNN_BLOCK), nchunks), dim3(KNN_BLOCK), 0, s, qx, qy, qz, qpad, px,
                       py, pz, npad, chunk_pts, floor_d2_in, floor_idx_in, part_d2, part_idx);
    hipLaunchKernelGGL((k_knnk_merge<K>), dim3(cdiv(Q, 64)), dim3(64), 0, s, part_d2, part_idx, nchunks, qpad, Q, kout,
                       col0, kstride, idx_base, d2_out, idx_out, floor_d2_out, floor_idx_out);
}

void launch_knnk_pass(hipStream_t s, int K, const double *qx, const double *qy, const double *qz, int qpad, long Q,
                      const double *px, const double *py, const double *pz, long npad, int chunk_pts, int nchunks,
                      const double *floor_d2_in, const uint32_t *floor_idx_in, double *part_d2, uint32_t *part_idx,
                      int kout, int col0, int kstride, int64_t idx_base, double *d2_out, int64_t *idx_out,
                      double *floor_d2_out, uint32_t *floor_idx_out)
{
#define SICP_KNNK_CASE(KK)                                                                                          \
    case KK:                                                                                                        \
        knnk_pass<KK>(s, qx, qy, qz, qpad, Q, px, py, pz, npad, chunk_pts, nchunks, floor_d2_in, floor_idx_in,      \
                      part_d2, part_idx, kout, col0, kstride, idx_base, d2_out, idx_out, floor_d2_out,              \
                      floor_idx_out);                                                                               \
        break;
    switch (K) {
        SICP_KNNK_CASE(8)
        SICP_KNNK_CASE(16)
        SICP_KNNK_CASE(32)
        SICP_KNNK_CASE(64)
    }
#undef SICP_KNNK_CASE
}

void launch_normals(hipStream_t s, const double *px, const double *py, const double *pz, const int64_t *nn, long Q, int k,
                    int64_t idx_base, float *normals, float *planarity)
{
    hipLaunchKernelGGL(k_normals, dim3(cdiv(Q, 64)), dim3(64), 0, s, px, py, pz, nn, Q, k, idx_base, normals, planarity);
}

void launch_postmatch(hipStream_t s, const double *qx, const double *qy, const double *qz, const float *normals,
                      const float *planarity, const double *p2, const int64_t *idx, long Q, const Xf &H,
                      float min_planarity, const float *pl2, long pl2_n, double *dist, uint8_t *flag, const IcpDev *st)
{
    hipLaunchKernelGGL(k_postmatch, dim3(cdiv(Q, 256)), dim3(256), 0, s, qx, qy, qz, normals, planarity, p2, idx, Q, H,
                       min_planarity, pl2, pl2_n, dist, flag, st);
}

void launch_corr_planarity(hipStream_t s, uint8_t *alive, const float *pl1, const float *pl2, float min_planarity, long Q)
{
    if (Q > 0) hipLaunchKernelGGL(k_corr_planarity, dim3(cdiv(Q, 256)), dim3(256), 0, s, alive, pl1, pl2, min_planarity, Q);
}

void launch_fill_f32(hipStream_t s, float *dst, long n, float v)
{
    if (n > 0) hipLaunchKernelGGL(k_fill_f32, dim3(cdiv(n, 256)), dim3(256), 0, s, dst, n, v);
}
void launch_scatter_f32(hipStream_t s, float *dst, const int64_t *rows, const float *vals, long m)
{
    if (m > 0) hipLaunchKernelGGL(k_scatter_f32, dim3(cdiv(m, 256)), dim3(256), 0, s, dst, rows, vals, m);
}

void launch_stats(hipStream_t s, const double *v, const uint8_t *keep, long Q, double *out3, const double *also4,
                  double *host_out, double seq, double *partial, unsigned *ticket, const IcpDev *st)
{
    if (partial && ticket && Q > STATS_MB_MIN_Q) {
        const int g = (int)std::min<long>(NE_MAX_GRID, (Q + 1023) / 1024);
        hipLaunchKernelGGL(k_stats_mb<0>, dim3(g), dim3(256), 0, s, v, keep, Q, out3, partial, ticket, also4, host_out, seq, st);
        hipLaunchKernelGGL(k_stats_mb<1>, dim3(g), dim3(256), 0, s, v, keep, Q, out3, partial, ticket, also4, host_out, seq, st);
        return;
    }
    hipLaunchKernelGGL(k_stats, dim3(1), dim3(1024), 0, s, v, keep, Q, out3, also4, host_out, seq, st);
}

int ne_grid_for(long count)
{
    long g = (count + NE_BLOCK - 1) / NE_BLOCK;
    if (g < 1) g = 1;
    if (g > NE_MAX_GRID) g = NE_MAX_GRID;
    return (int)g;
}

void launch_normal_eq(hipStream_t s, const double *qx, const double *qy, const double *qz, const float *normals,
                      const double *p2, const uint8_t *keep, long lo, long hi, const double H12[12], const double dR[27],
                      double *partial, unsigned *ticket, double *out30, double *resid, double *host_out, double seq)
{
    NeArgs A;
    for (int i = 0; i < 12; ++i) A.H.m[i] = H12[i];
    for (int i = 0; i < 27; ++i) A.dR[i] = dR[i];
    hipLaunchKernelGGL(k_normal_eq, dim3(ne_grid_for(hi - lo)), dim3(NE_BLOCK), 0, s, qx, qy, qz, normals, p2, keep, lo,
                       hi, A, partial, ticket, out30, resid, host_out, seq);
}

}  // namespace sicp
