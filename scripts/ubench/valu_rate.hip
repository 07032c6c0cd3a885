// VALU issue-rate microbenchmark (gfx950): v_fma_f32 vs v_pk_fma_f32 vs v_fma_f64 vs v_min3_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITERS 4096
#define NACC 16

__global__ void k_fma32(float *out, float a, float b) {
    float acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x + i;
    for (int it = 0; it < ITERS; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
    float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_pkfma32(float *out, float a, float b) {
    v2f acc[NACC]; v2f A = {a, a * 1.0001f}, B = {b, b * 0.999f};
    for (int i = 0; i < NACC; ++i) acc[i] = (v2f){(float)threadIdx.x + i, (float)i};
    for (int it = 0; it < ITERS; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_elementwise_fma(acc[i], A, B);
    v2f s = {0, 0}; for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
__global__ void k_fma64(float *out, double a, double b) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x + i;
    for (int it = 0; it < ITERS; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}
__global__ void k_min3(float *out, float a, float b) {
    float acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x + i;
    for (int it = 0; it < ITERS; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fminf(fminf(acc[i], a + it), b - i);   // expect v_min3_f32
    float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_pkmin(float *out, float a, float b) {
    v2f acc[NACC]; v2f A = {a, a * 1.0001f};
    for (int i = 0; i < NACC; ++i) acc[i] = (v2f){(float)threadIdx.x + i, (float)i};
    for (int it = 0; it < ITERS; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) { acc[i] = __builtin_elementwise_min(acc[i], A); A.x += 1.0f; }
    v2f s = {0, 0}; for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
template <class F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4 * 4);
    const int blocks = 256 * 8, threads = 256;   // 8 blocks/CU = 8 waves/SIMD
    const double waves = (double)blocks * threads / 64, ins = waves * ITERS * NACC;
    struct { const char *n; float ms; double flop; } r[5];
    r[0] = {"v_fma_f32   ", timeit([&] { hipLaunchKernelGGL(k_fma32, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f); }), 2};
    r[1] = {"v_pk_fma_f32", timeit([&] { hipLaunchKernelGGL(k_pkfma32, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f); }), 4};
    r[2] = {"v_fma_f64   ", timeit([&] { hipLaunchKernelGGL(k_fma64, dim3(blocks), dim3(threads), 0, 0, out, 1.0001, 0.5); }), 2};
    r[3] = {"v_min3_f32  ", timeit([&] { hipLaunchKernelGGL(k_min3, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f); }), 0};
    r[4] = {"v_pk_min_f32", timeit([&] { hipLaunchKernelGGL(k_pkmin, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f); }), 0};
    for (auto &x : r) {
        const double per_simd_cycle = ins / 1024.0 / (x.ms * 1e-3);   // wave-instr per SIMD per second
        printf("%s %8.3f ms  %7.2f G wave-instr/s/SIMD (=> %.2f cycles/instr at 2.4 GHz)  %7.1f TFLOP/s\n", x.n, x.ms,
               per_simd_cycle / 1e9, 2.4e9 / per_simd_cycle, ins * 64 * x.flop / (x.ms * 1e-3) / 1e12);
    }
    return 0;
}
