// d2h_rate.hip -- device-to-pinned-host rate by piece size and number of streams (what feeds sicp_cloud_download_both's ring).
// hipcc --offload-arch=gfx950 -O2 -o d2h_rate d2h_rate.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    const size_t bytes = (size_t)240 << 20;
    char *d, *h;
    CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 1, bytes));
    CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    hipStream_t s[4];
    for (auto &x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    for (int dir = 0; dir < 2; ++dir)
        for (size_t piece : {bytes, (size_t)16 << 20, (size_t)4 << 20, (size_t)2 << 20, (size_t)1 << 20})
            for (int ns : {1, 2, 4}) {
                double best = 1e9;
                for (int rep = 0; rep < 3; ++rep) {
                    const double t0 = now();
                    int k = 0;
                    for (size_t o = 0; o < bytes; o += piece, ++k) {
                        const size_t m = piece < bytes - o ? piece : bytes - o;
                        if (dir == 0) CK(hipMemcpyAsync(h + o, d + o, m, hipMemcpyDeviceToHost, s[k % ns]));
                        else CK(hipMemcpyAsync(d + o, h + o, m, hipMemcpyHostToDevice, s[k % ns]));
                    }
                    for (int i = 0; i < ns; ++i) CK(hipStreamSynchronize(s[i]));
                    const double dt = now() - t0;
                    if (dt < best) best = dt;
                }
                printf("%s pieces of %6.1f MiB on %d stream(s): %.2f ms = %.1f GB/s\n", dir ? "H2D" : "D2H", piece / 1048576.0, ns, best, bytes / best / 1e6);
            }
    return 0;
}
