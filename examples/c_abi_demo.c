/* A plain C99 host of libsimpleicp_hip.so: the boundary is a C ABI (include/simpleicp_hip.h), so any language with a C FFI
 * can sit where simpleICP's Python run() sits.  Registers a synthetic surface against a rigidly moved, independently
 * sampled copy of itself and prints the estimated parameters.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/c_abi_demo.c -Lsimpleicp_amd -lsimpleicp_hip -lm -Wl,-rpath,$PWD/simpleicp_amd -o c_abi_demo
 *   ./c_abi_demo [points] [correspondences]
 * Exit code 0 on success, 3 when no gfx950 device is visible (the library has no CPU path), 1 on any other error. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "simpleicp_hip.h"

static double urand(unsigned long long *s)
{
    *s = *s * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(*s >> 11) / 9007199254740992.0;
}

static void surface(double *xyz, long n, double side, unsigned long long seed)
{
    for (long i = 0; i < n; ++i) {
        const double x = urand(&seed) * side, y = urand(&seed) * side;
        xyz[3 * i] = x - side / 2; xyz[3 * i + 1] = y - side / 2;
        xyz[3 * i + 2] = 2.0 * sin(x / 5.0) * cos(y / 7.0) + 0.5 * sin(x / 1.3 + 1.0) * sin(y / 1.7);
    }
}

#define CHECK(call)                                                                     \
    do {                                                                                \
        const int rc_ = (call);                                                         \
        if (rc_ != SICP_OK) {                                                           \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, sicp_last_error());           \
            if (ctx) {                                                                  \
                sicp_ctx_destroy(ctx);                                                  \
            }                                                                           \
            return rc_ == SICP_ERR_NO_DEVICE ? 3 : 1;                                   \
        }                                                                               \
    } while (0)

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 200000, want = argc > 2 ? atol(argv[2]) : 1000;
    sicp_ctx *ctx = NULL;
    printf("ABI version %d\n", sicp_abi_version());
    CHECK(sicp_ctx_create(0, &ctx));

    const double side = sqrt((double)n / 10.0);
    double *fix = malloc(sizeof(double) * 3 * n), *mov = malloc(sizeof(double) * 3 * n);
    surface(fix, n, side, 1); surface(mov, n, side, 2);
    /* movable = H_true^-1 (independent sampling of the same surface) */
    const double x_true[6] = {0.004, -0.003, 0.006, 0.05, -0.04, 0.02};
    double H[16];
    CHECK(sicp_params_to_H(x_true, H));
    for (long i = 0; i < n; ++i) {          /* p' = R^T (p - t) */
        const double p[3] = {mov[3 * i] - H[3], mov[3 * i + 1] - H[7], mov[3 * i + 2] - H[11]};
        for (int r = 0; r < 3; ++r) mov[3 * i + r] = H[r] * p[0] + H[4 + r] * p[1] + H[8 + r] * p[2];
    }
    CHECK(sicp_cloud_upload(ctx, SICP_FIX, fix, n, 0));
    CHECK(sicp_cloud_upload(ctx, SICP_MOV, mov, n, 0));

    /* select_n_points (pointcloud.py:132-147): equidistant rows, half-to-even rounding */
    long Q = 0;
    int64_t *sel = malloc(sizeof(int64_t) * want);
    for (long k = 0; k < want; ++k) {
        const int64_t i = (int64_t)nearbyint((double)k * (double)(n - 1) / (double)(want - 1));
        if (Q == 0 || sel[Q - 1] != i) sel[Q++] = i;
    }
    float *normals = malloc(sizeof(float) * 3 * Q), *planarity = malloc(sizeof(float) * Q);
    CHECK(sicp_estimate_normals(ctx, SICP_FIX, sel, Q, 10, normals, planarity, NULL));
    CHECK(sicp_icp_setup(ctx, sel, Q, normals, planarity));

    sicp_iter_params P;
    for (int j = 0; j < 6; ++j) { P.x[j] = 0.0; P.obs[j] = 0.0; P.obs_weight[j] = 0.0; }
    P.min_planarity = 0.3; P.distance_weight = 1.0; P.max_lm_steps = 0;
    sicp_iter_result R[100];
    int64_t its = 0;
    CHECK(sicp_icp_run(ctx, &P, 100, 1.0, R, &its));        /* the whole loop of simpleicp.py:184-261 behind one call */
    double err = 0.0;
    for (int j = 0; j < 6; ++j) err = fmax(err, fabs(R[its - 1].x[j] - x_true[j]));
    printf("%ld points, %ld correspondences, %lld iterations, %lld kept, residual std %.5f\n", n, Q, (long long)its,
           (long long)R[its - 1].n_kept, R[its - 1].res_std);
    printf("x = %.6f %.6f %.6f %.5f %.5f %.5f   max |x - x_true| = %.2e\n", R[its - 1].x[0], R[its - 1].x[1], R[its - 1].x[2],
           R[its - 1].x[3], R[its - 1].x[4], R[its - 1].x[5], err);

    /* The same iterations operator by operator, the way the reference's own loop drives its classes (simpleicp.py:190-227):
     * CorrPts.match, .reject_wrt_planarity, .reject_wrt_point_to_plane_distances, SimpleICPOptimization.estimate_parameters. */
    sicp_iter_params Po = P;
    sicp_iter_result Ro;
    int64_t alive = 0;
    for (int64_t it = 0; it < its; ++it) {
        double Hx[16], median, mad;
        CHECK(sicp_params_to_H(Po.x, Hx));
        CHECK(sicp_corr_match(ctx, Hx, NULL, NULL));
        CHECK(sicp_corr_reject_planarity(ctx, P.min_planarity, planarity, NULL, &alive));
        CHECK(sicp_corr_reject_distances(ctx, &median, &mad, &alive));
        CHECK(sicp_estimate_parameters(ctx, &Po, NULL, &Ro));
        for (int j = 0; j < 6; ++j) Po.x[j] = Ro.x[j];
    }
    double diff = 0.0;
    for (int j = 0; j < 6; ++j) diff = fmax(diff, fabs(Ro.x[j] - R[its - 1].x[j]));
    printf("operator by operator: %lld alive in the last iteration, max |x - x_run| = %.2e\n", (long long)alive, diff);
    sicp_ctx_destroy(ctx);
    free(fix); free(mov); free(sel); free(normals); free(planarity);
    return (err < 5e-3 && diff < 1e-6 && llabs((long long)(alive - R[its - 1].n_kept)) <= 2) ? 0 : 1;
}
