// gather_calib.hip -- what does rocprofv3's FETCH_SIZE report for the access patterns of the grid kernels?
// MI355X_MICROARCH.md calibrates the counter for wide streaming reads only (it reports HALF their bytes on gfx950) and says other
// widths are uncalibrated.  The grid searches gather 32-byte records (k_grid_nn*), 16-byte records (k_grid_nn16f) and 4 / 8-byte
// table entries at random positions of tables far larger than L2 + Infinity Cache.  Every kernel here reads a KNOWN number of bytes
// that way (positions from a hash of the lane's index: no index array in the traffic); run it under
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace -d <dir> -- scripts/ubench/gather_calib
// and divide.  scripts/summarize_profile.py applies the factors recorded in profiles/r5/README.md.
//     hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/gather_calib scripts/ubench/gather_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

// streaming: every lane reads 16 bytes, consecutive lanes consecutive addresses (the guide's calibrated case)
__global__ __launch_bounds__(256) void k_stream16(const float4 *__restrict__ a, long n, float *__restrict__ sink)
{
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 123.456f) sink[0] = s;
}
// gathers: `per` random records per lane out of nrec
template <typename T>
__global__ __launch_bounds__(256) void k_gather(const T *__restrict__ a, long nrec, int per, float *__restrict__ sink)
{
    const uint64_t id = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    float s = 0.f;
    for (int j = 0; j < per; ++j) {
        const T v = a[mix(id * 1315423911ull + (uint64_t)j * 2654435761ull) % (uint64_t)nrec];
        s += ((const float *)&v)[0];
    }
    if (s == 123.456f) sink[0] = s;
}
// a run of 8 consecutive 32-byte records at a random position (what a lane group reads from one grid row)
__global__ __launch_bounds__(256) void k_gather_run32(const double4 *__restrict__ a, long nrec, int per, float *__restrict__ sink)
{
    const uint64_t id = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t grp = id >> 3, l = id & 7;
    float s = 0.f;
    for (int j = 0; j < per; ++j) {
        const double4 v = a[(mix(grp * 1315423911ull + (uint64_t)j * 2654435761ull) % (uint64_t)(nrec - 8)) + l];
        s += (float)v.x;
    }
    if (s == 123.456f) sink[0] = s;
}

int main()
{
    const size_t bytes = (size_t)4 << 30;                       // 4 GiB table: 16 x the Infinity Cache
    void *buf; float *sink;
    CHK(hipMalloc(&buf, bytes)); CHK(hipMalloc((void **)&sink, 64));
    CHK(hipMemset(buf, 1, bytes));
    const long lanes = 64L << 20;                               // 64 Mi lanes
    const int per = 2;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_stream16, dim3(8192), dim3(256), 0, 0, (const float4 *)buf, (long)(bytes / 16), sink);
        hipLaunchKernelGGL(k_gather<double4>, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, (const double4 *)buf, (long)(bytes / 32), per, sink);
        hipLaunchKernelGGL(k_gather<float4>, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, (const float4 *)buf, (long)(bytes / 16), per, sink);
        hipLaunchKernelGGL(k_gather<uint2>, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, (const uint2 *)buf, (long)(bytes / 8), per, sink);
        hipLaunchKernelGGL(k_gather<uint32_t>, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, (const uint32_t *)buf, (long)(bytes / 4), per, sink);
        hipLaunchKernelGGL(k_gather_run32, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, (const double4 *)buf, (long)(bytes / 32), per, sink);
    }
    CHK(hipDeviceSynchronize());
    std::printf("bytes requested per launch: k_stream16 %zu | k_gather<double4> %ld | k_gather<float4> %ld | k_gather<uint2> %ld | k_gather<uint32_t> %ld | k_gather_run32 %ld\n",
                bytes, lanes * per * 32, lanes * per * 16, lanes * per * 8, lanes * per * 4, lanes * per * 32);
    return 0;
}
