// Where does the one-workgroup rejection spend its time?  The kernel of simpleicp_amd/csrc/sicp_reject.hip built with its cycle stamps,
// on 10 000 synthetic distances (normal + 5 % outliers), first without a prior (histogram rounds), then with the last launch's.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSICP_REJECT_TRACE -I simpleicp_amd/csrc scripts/ubench/reject_trace.hip -o scripts/ubench/reject_trace
#include "../../simpleicp_amd/csrc/sicp_reject.hip"
#include <cstdio>
#include <random>
#include <vector>
int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 10000;
    std::mt19937_64 g(1);
    std::normal_distribution<double> N(0.0, 0.02);
    std::uniform_real_distribution<double> U(-2.0, 2.0);
    std::vector<double> d(n); std::vector<uint8_t> f(n);
    for (int i = 0; i < n; ++i) { d[i] = (i % 20 == 0) ? U(g) : N(g); f[i] = (i % 5) != 0; }
    double *dd, *o4; uint8_t *df, *dk;
    hipMalloc(&dd, n * 8); hipMalloc(&df, n); hipMalloc(&dk, n); hipMalloc(&o4, 64 * 8);
    hipMemcpy(dd, d.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(df, f.data(), n, hipMemcpyHostToDevice);
    hipMemset(o4, 0, 64 * 8);
    const char *names[6] = {"loads+keys", "count+list+barrier", "median", "MAD", "keep+sums", "write"};
    for (int rep = 0; rep < 6; ++rep) {
        const bool prior = rep >= 2;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, 0);
        sicp::launch_reject(0, dd, df, n, dk, o4, nullptr, o4 + 4, prior);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double h[48]; hipMemcpy(h, o4, sizeof h, hipMemcpyDeviceToHost);
        printf("n %d %s: %.1f us  m %.0f median %.6e mad %.6e kept %.0f | cycles:", n, prior ? "window " : "general", ms * 1e3, h[0], h[1], h[2], h[3]);
        for (int i = 0; i < 6; ++i) printf(" %s %.0f", names[i], h[40 + i]);
        printf("\n");
    }
    return 0;
}
