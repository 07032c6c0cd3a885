// sicp_normals.h -- covariance -> normal + planarity (pointcloud.py:188-203), shared by the two kernels that end in it:
// k_normals (neighbour indices in, sicp_kernels.hip) and k_grid_knn_sweep (the fused k-NN + covariance sweep, sicp_grid.hip).
// Same operation order as oracle/sicp_oracle.c:orc_normals (cyclic Jacobi in fp64, smallest eigenvalue's vector, largest
// component positive, float32 store), so both kernels give the same bits for the same covariance.
#ifndef SICP_NORMALS_H
#define SICP_NORMALS_H

#include <hip/hip_runtime.h>

namespace sicp {

__device__ inline void jacobi3(double a[3][3], double v[3][3])
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; ++sweep) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        if (off == 0.0) break;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = (pq == 2) ? 1 : 0;
            const int q = (pq == 0) ? 1 : 2;
            const int r = 3 - p - q;
            if (a[p][q] == 0.0) continue;
            const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            const double apq = a[p][q];
            a[p][p] -= t * apq; a[q][q] += t * apq; a[p][q] = 0.0; a[q][p] = 0.0;
            const double arp = a[r][p], arq = a[r][q];
            a[r][p] = c * arp - s * arq; a[p][r] = a[r][p];
            a[r][q] = s * arp + c * arq; a[q][r] = a[r][q];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double vip = v[i][p], viq = v[i][q];
                v[i][p] = c * vip - s * viq;
                v[i][q] = s * vip + c * viq;
            }
        }
    }
}

// c6 = upper triangle (00, 01, 02, 11, 12, 22) of the sample covariance, already divided by k - 1
__device__ inline void normal_from_cov(const double c6[6], float nrm[3], float *planarity)
{
    double C[3][3];
    C[0][0] = c6[0]; C[0][1] = c6[1]; C[0][2] = c6[2]; C[1][1] = c6[3]; C[1][2] = c6[4]; C[2][2] = c6[5];
    C[1][0] = C[0][1]; C[2][0] = C[0][2]; C[2][1] = C[1][2];
    double V[3][3];
    jacobi3(C, V);
    const double w[3] = {C[0][0], C[1][1], C[2][2]};
    int lo = 0, hi = 0;
#pragma unroll
    for (int c = 1; c < 3; ++c) { if (w[c] < w[lo]) lo = c; if (w[c] > w[hi]) hi = c; }
    if (lo == hi) { lo = 2; hi = 0; }
    const int mid = 3 - lo - hi;
    double wl = 0, wm = 0, wh = 0, n0 = 0, n1 = 0, n2 = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (c == lo) { wl = w[c]; n0 = V[0][c]; n1 = V[1][c]; n2 = V[2][c]; }
        if (c == mid) wm = w[c];
        if (c == hi) wh = w[c];
    }
    int big = 0; double bigv = fabs(n0);
    if (fabs(n1) > bigv) { big = 1; bigv = fabs(n1); }
    if (fabs(n2) > bigv) { big = 2; }
    const double lead = (big == 0) ? n0 : (big == 1) ? n1 : n2;
    if (lead < 0) { n0 = -n0; n1 = -n1; n2 = -n2; }
    nrm[0] = (float)n0; nrm[1] = (float)n1; nrm[2] = (float)n2;
    *planarity = (float)((wm - wl) / wh);
}

}  // namespace sicp

#endif
