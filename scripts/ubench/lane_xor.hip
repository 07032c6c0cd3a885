#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned dpp(unsigned v, int) { return v; }
template <int CTRL> __device__ __forceinline__ unsigned dppmov(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
// value held by lane (lane ^ J)
template <int J> __device__ __forceinline__ unsigned xor_lane(unsigned v) {
    if constexpr (J == 1) return dppmov<0xB1>(v);
    else if constexpr (J == 2) return dppmov<0x4E>(v);
    else if constexpr (J == 4) return dppmov<0x1B>(dppmov<0x141>(v));
    else if constexpr (J == 8) return dppmov<0x141>(dppmov<0x140>(v));
    else if constexpr (J == 16) { v2u r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return (threadIdx.x & 16) ? r.x : r.y; }
    else { v2u r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return (threadIdx.x & 32) ? r.x : r.y; }
}
__global__ void k(unsigned *out) {
    const unsigned v = threadIdx.x * 3 + 7;
    out[0 * 64 + threadIdx.x] = xor_lane<1>(v);
    out[1 * 64 + threadIdx.x] = xor_lane<2>(v);
    out[2 * 64 + threadIdx.x] = xor_lane<4>(v);
    out[3 * 64 + threadIdx.x] = xor_lane<8>(v);
    out[4 * 64 + threadIdx.x] = xor_lane<16>(v);
    out[5 * 64 + threadIdx.x] = xor_lane<32>(v);
}
int main() {
    unsigned *d, h[6 * 64]; hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const int J[6] = {1, 2, 4, 8, 16, 32}; int bad = 0;
    for (int s = 0; s < 6; ++s) for (int l = 0; l < 64; ++l) if (h[s * 64 + l] != (unsigned)((l ^ J[s]) * 3 + 7)) { if (bad < 10) printf("J=%d lane %d got %u want %u\n", J[s], l, h[s*64+l], (l ^ J[s]) * 3 + 7); ++bad; }
    printf("bad=%d\n", bad); return bad != 0;
}
