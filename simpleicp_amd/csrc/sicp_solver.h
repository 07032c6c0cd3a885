// sicp_solver.h -- device-side building blocks of the Levenberg-Marquardt solver shared by the single-launch tail
// (sicp_tail.hip, Q <= SOLVE_MAX_Q) and the multi-workgroup evaluation chain (sicp_lm.hip, larger Q):
// contracts (T) and (P), Euler rotation, sin / cos stepping, objective, and the in-lane 6x6 LDL^T step on the 8x8 Gram
// matrix G of the rows [a0..a5 | r | 1]  (J^T J = G[0..5][0..5], J^T r = G[.][6], sum r = G[6][7], sum r^2 = G[6][6],
// n = G[7][7]; optimization.py:65-124,172-288).  Device code only; everything has internal linkage.
#ifndef SICP_SOLVER_H
#define SICP_SOLVER_H

#include <hip/hip_runtime.h>

#include "sicp_internal.h"

namespace sicp {
namespace {

__device__ __forceinline__ void xfm(const double (&H)[12], double x, double y, double z, double &ox, double &oy, double &oz)
{
    double t;
    t = H[0] * x;  t = fma(H[1], y, t);  t = fma(H[2], z, t);   ox = t + H[3];
    t = H[4] * x;  t = fma(H[5], y, t);  t = fma(H[6], z, t);   oy = t + H[7];
    t = H[8] * x;  t = fma(H[9], y, t);  t = fma(H[10], z, t);  oz = t + H[11];
}
__device__ __forceinline__ double pdist(double dx, double dy, double dz, float nx, float ny, float nz)
{
    const double a = dx * (double)nx, b = dy * (double)ny, c = dz * (double)nz;
    return (a + b) + c;
}
// sin/cos of (a + d) from sin/cos of a: exact addition theorem with a short Taylor series for the
// small step d (|d| <= 0.25 rad: d^17/17! < 2e-25); larger steps take the library routine.
__device__ __attribute__((noinline)) double2 sincos_cold(double a)      // one out-of-line copy, results in registers
{
    double s, c;
    sincos(a, &s, &c);
    return make_double2(s, c);
}

__device__ __forceinline__ void sincos_step(double a_new, double d, double sa, double ca, double &sn, double &cn)
{
    if (fabs(d) > 0.25) { const double2 r = sincos_cold(a_new); sn = r.x; cn = r.y; return; }
    const double d2 = d * d;
    double sd = 1.0 / 1307674368000.0;
    sd = fma(sd, d2, -1.0 / 6227020800.0);
    sd = fma(sd, d2, 1.0 / 39916800.0);
    sd = fma(sd, d2, -1.0 / 362880.0);
    sd = fma(sd, d2, 1.0 / 5040.0);
    sd = fma(sd, d2, -1.0 / 120.0);
    sd = fma(sd, d2, 1.0 / 6.0);
    sd = d - d * d2 * sd;
    double cd = 1.0 / 20922789888000.0;
    cd = fma(cd, d2, -1.0 / 87178291200.0);
    cd = fma(cd, d2, 1.0 / 479001600.0);
    cd = fma(cd, d2, -1.0 / 3628800.0);
    cd = fma(cd, d2, 1.0 / 40320.0);
    cd = fma(cd, d2, -1.0 / 720.0);
    cd = fma(cd, d2, 1.0 / 24.0);
    cd = 1.0 - d2 * (0.5 - d2 * cd);
    sn = fma(sa, cd, ca * sd);
    cn = fma(ca, cd, -(sa * sd));
}

__device__ __forceinline__ bool observed(double w) { return w > 0 && w < __builtin_inf(); }

__device__ __forceinline__ void euler_H(const double (&x)[6], const double (&sc)[6], double (&H)[12])
{
    const double s1 = sc[0], c1 = sc[1], s2 = sc[2], c2 = sc[3], s3 = sc[4], c3 = sc[5];
    H[0] = c2 * c3;                 H[1] = -c2 * s3;                H[2] = s2;        H[3] = x[3];
    H[4] = c1 * s3 + s1 * s2 * c3;  H[5] = c1 * c3 - s1 * s2 * s3;  H[6] = -s1 * c2;  H[7] = x[4];
    H[8] = s1 * s3 - c1 * s2 * c3;  H[9] = s1 * c3 + c1 * s2 * s3;  H[10] = c1 * c2;  H[11] = x[5];
}

__device__ __forceinline__ double objective(const double *G, double w, const double (&x)[6], const TailArgs &A)
{
    double c = w * w * G[6 * 8 + 6];
#pragma unroll
    for (int j = 0; j < 6; ++j)
        if (observed(A.ow[j])) { const double e = A.ow[j] * (x[j] - A.obs[j]); c += e * e; }
    return c;
}

// LM step from the normal equations: (N + lambda diag N) dx = -g in every lane (LDL^T in registers, in place on
// the lower triangle); parameters with an infinite observation weight are fixed (identity row / column).
// Returns false when a pivot is not positive and finite.
__device__ __forceinline__ bool lm_step(const double *G, double w, const double (&x)[6], double lambda, const TailArgs &A,
                                        double (&dx)[6])
{
    const double w2 = w * w;
    double M[6][6], b[6];                                  // only M[i][j], j <= i, is used
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const bool fi = !(A.ow[i] < __builtin_inf());
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            const bool fj = !(A.ow[j] < __builtin_inf());
            double a = w2 * G[j * 8 + i];
            if (i == j) { if (observed(A.ow[i])) a += A.ow[i] * A.ow[i]; a += lambda * a; }
            M[i][j] = (fi || fj) ? (i == j ? 1.0 : 0.0) : a;
        }
        double g = w2 * G[i * 8 + 6];
        if (observed(A.ow[i])) g += A.ow[i] * A.ow[i] * (x[i] - A.obs[i]);
        b[i] = fi ? 0.0 : -g;
    }
    // M = L D L^T: afterwards M[i][j] (j < i) = L[i][j], M[j][j] = D[j].  The pivots' reciprocals are formed ONCE (hardware
    // estimate + two Newton steps: full precision, a third of the latency of the IEEE division sequence, and this chain is
    // on every lane's critical path) and reused by the back-substitution.
    bool ok = true;
    double inv[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        // v[k] = L[j][k] * D[k]: formed once per column, every product below is ONE fma
        double v[6] = {0, 0, 0, 0, 0, 0};
        double d = M[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) { v[k] = M[j][k] * M[k][k]; d = fma(-M[j][k], v[k], d); }
        ok = ok && (d > 0.0) && (d < __builtin_inf());
        M[j][j] = d;
        double y = __builtin_amdgcn_rcp(d);
        y = fma(fma(-d, y, 1.0), y, y);
        y = fma(fma(-d, y, 1.0), y, y);
        inv[j] = y;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double t = M[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) t = fma(-M[i][k], v[k], t);
            M[i][j] = t * y;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int k = 0; k < i; ++k) b[i] = fma(-M[i][k], b[k], b[i]);
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double t = b[i] * inv[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) t = fma(-M[k][i], dx[k], t);
        dx[i] = t;
    }
    return ok;
}


}  // namespace
}  // namespace sicp

#endif
