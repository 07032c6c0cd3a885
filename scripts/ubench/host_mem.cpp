// host_mem.cpp -- what the host side of a 240 MB download costs: first touch of fresh pages (with / without transparent huge pages) and
// the fan-out of a warm source into rows + columns by T threads.   g++ -O2 -pthread -o host_mem host_mem.cpp && ./host_mem
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class F> static double par(unsigned T, F f)
{
    const double t0 = now();
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) th.emplace_back(f, t);
    f(0u);
    for (auto &x : th) x.join();
    return (now() - t0) * 1e3;
}

int main()
{
    const long n = 10'000'000;
    const size_t bytes = (size_t)3 * n * sizeof(double);
    double *src = (double *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    memset(src, 1, bytes);
    for (int huge = 0; huge < 2; ++huge)
        for (unsigned T : {1u, 4u, 8u, 16u, 32u}) {
            double *rows = (double *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            double *cols = (double *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (huge) { madvise(rows, bytes, MADV_HUGEPAGE); madvise(cols, bytes, MADV_HUGEPAGE); }
            auto fan = [&](unsigned t) {
                const long a = n * t / T, e = n * (t + 1) / T;
                memcpy(cols + a, src + a, (size_t)(e - a) * 8);
                memcpy(cols + n + a, src + n + a, (size_t)(e - a) * 8);
                memcpy(cols + 2 * n + a, src + 2 * n + a, (size_t)(e - a) * 8);
                double *o = rows + 3 * a;
                for (long i = a; i < e; ++i) { o[0] = src[i]; o[1] = src[n + i]; o[2] = src[2 * n + i]; o += 3; }
            };
            const double cold = par(T, fan), warm = par(T, fan);
            printf("huge=%d threads=%2u: fan-out into fresh pages %.2f ms, into touched pages %.2f ms\n", huge, T, cold, warm);
            const double t0 = now();
            munmap(rows, bytes); munmap(cols, bytes);
            printf("                     munmap of both %.2f ms\n", (now() - t0) * 1e3);
        }
    return 0;
}
