// pageable_async.hip -- does hipMemcpyAsync out of pageable memory return before the copy is done (can the calling thread launch other
// work behind it), and does a kernel on another stream run beside it?   hipcc --offload-arch=gfx950 -O2 -o pageable_async pageable_async.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void spin(double *p, long n, int rounds)
{
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = p[i];
    for (int r = 0; r < rounds; ++r) v = v * 1.0000001 + 1e-9;
    p[i] = v;
}
int main()
{
    const size_t bytes = (size_t)240 << 20;
    double *h = (double *)malloc(bytes), *h2 = (double *)malloc(bytes);
    memset(h, 1, bytes); memset(h2, 1, bytes);
    double *d, *d2, *w;
    hipMalloc(&d, bytes); hipMalloc(&d2, bytes); hipMalloc(&w, (size_t)64 << 20);
    hipMemset(w, 0, (size_t)64 << 20);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const long wn = ((size_t)64 << 20) / 8;
    for (int rep = 0; rep < 4; ++rep) {
        double t0 = now();
        hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s1);
        double t1 = now();
        hipStreamSynchronize(s1);
        double t2 = now();
        printf("rep %d: hipMemcpyAsync returned after %.2f ms, done after %.2f ms\n", rep, t1 - t0, t2 - t0);
    }
    // a kernel of ~3 ms alone
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        spin<<<(wn + 255) / 256, 256, 0, s2>>>(w, wn, 3000);
        hipStreamSynchronize(s2);
        printf("kernel alone: %.2f ms\n", now() - t0);
    }
    // copy on a helper thread + kernel on this one
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now(), tc = 0;
        std::thread th([&] { hipMemcpyAsync(d2, h2, bytes, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1); tc = now() - t0; });
        spin<<<(wn + 255) / 256, 256, 0, s2>>>(w, wn, 3000);
        hipStreamSynchronize(s2);
        double tk = now() - t0;
        th.join();
        printf("helper thread copy %.2f ms, kernel beside it %.2f ms, both %.2f ms\n", tc, tk, now() - t0);
    }
    // fresh (never pinned) source each time
    for (int rep = 0; rep < 2; ++rep) {
        double *f = (double *)malloc(bytes); memset(f, 2, bytes);
        double t0 = now();
        hipMemcpyAsync(d, f, bytes, hipMemcpyHostToDevice, s1);
        double t1 = now();
        hipStreamSynchronize(s1);
        printf("fresh source: returned after %.2f ms, done after %.2f ms\n", t1 - t0, now() - t0);
        free(f);
    }
    return 0;
}
