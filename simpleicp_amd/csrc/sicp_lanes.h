// sicp_lanes.h -- register-speed cross-lane moves for wave64 on gfx950 (device code only).
//
// __shfl* compiles to ds_bpermute (the LDS crossbar): ~250 cycles per step in a dependent chain.  A fixed
// lane ^ J exchange needs no crossbar: DPP quad_perm for J = 1, 2; J = 4 = row_half_mirror (i^7) then quad
// reverse (i^3); J = 8 = row_mirror (i^15) then row_half_mirror; J = 16 / 32 = gfx950's
// v_permlane16_swap / v_permlane32_swap.  (scripts/ubench/lane_xor.hip checks all six on the device.)
// Blocks are one-dimensional everywhere, so threadIdx.x & 63 is the lane.
#ifndef SICP_LANES_H
#define SICP_LANES_H

#include <hip/hip_runtime.h>

namespace sicp {

typedef unsigned v2u_t __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ unsigned dpp_mov(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

// value of v held by lane (lane ^ J), J in {1, 2, 4, 8, 16, 32}
template <int J>
__device__ __forceinline__ unsigned lane_xor32(unsigned v)
{
    if constexpr (J == 1) return dpp_mov<0xB1>(v);
    else if constexpr (J == 2) return dpp_mov<0x4E>(v);
    else if constexpr (J == 4) return dpp_mov<0x1B>(dpp_mov<0x141>(v));
    else if constexpr (J == 8) return dpp_mov<0x141>(dpp_mov<0x140>(v));
    else if constexpr (J == 16) { const v2u_t r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return (threadIdx.x & 16) ? r.x : r.y; }
    else { const v2u_t r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return (threadIdx.x & 32) ? r.x : r.y; }
}

template <int J>
__device__ __forceinline__ unsigned long long lane_xor64(unsigned long long v)
{
    const unsigned lo = lane_xor32<J>((unsigned)v), hi = lane_xor32<J>((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

template <int J>
__device__ __forceinline__ double lane_xor_f64(double v)
{
    return __longlong_as_double((long long)lane_xor64<J>((unsigned long long)__double_as_longlong(v)));
}

// wave-wide sum by a butterfly: every lane ends up with the total, in a fixed order
__device__ __forceinline__ double wsum(double v)
{
    v += lane_xor_f64<32>(v);
    v += lane_xor_f64<16>(v);
    v += lane_xor_f64<8>(v);
    v += lane_xor_f64<4>(v);
    v += lane_xor_f64<2>(v);
    v += lane_xor_f64<1>(v);
    return v;
}

// wave-wide inclusive prefix sum (lane l gets v[0] + ... + v[l]): row_shr 1/2/4/8 inside each row of 16 lanes
// (missing sources read 0), then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3
__device__ __forceinline__ unsigned wscan_u32(unsigned v)
{
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}

__device__ __forceinline__ unsigned long long wsum_u64(unsigned long long v)
{
    v += lane_xor64<32>(v);
    v += lane_xor64<16>(v);
    v += lane_xor64<8>(v);
    v += lane_xor64<4>(v);
    v += lane_xor64<2>(v);
    v += lane_xor64<1>(v);
    return v;
}


// ---- grid barrier for kernels whose blocks are all resident at once (at most one block per CU is launched) -------------------
// Arrivals are counted at one address, the release is published at another: the waiting blocks poll `gen` (a load nobody
// writes until the barrier opens), so they do not queue up behind -- and slow down -- the arriving blocks' read-modify-writes on
// `bar` (256 pollers on the counter itself made a phase 40 % slower than a kernel boundary).  The counter only ever grows: a
// launch adds exactly gridDim.x * MAXB to it (blocks top their share up when they leave, grid_barrier_leave), the host hands
// every launch the value it starts from, barrier k of a launch waits for  base + gridDim.x * (k + 1)  -- no reset between
// launches.  Polling is bounded (about two seconds): a launch that cannot meet itself flags an error and goes on instead of
// hanging the queue.
struct GridBar {
    unsigned long long bar;        // arrivals, monotone over the life of the buffer
    unsigned long long pad0[7];
    unsigned long long gen;        // the last barrier target reached
    unsigned long long pad1[7];
    unsigned error;                // a wait timed out (sticky)
    unsigned pad2[15];
};
__device__ __forceinline__ void grid_barrier(GridBar *B, unsigned long long target)
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long mine = __hip_atomic_fetch_add(&B->bar, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
        if (mine >= target) {
            // last to arrive: open (a later barrier's opener can only run after this one's waiters have left: gen never moves back)
            __hip_atomic_store(&B->gen, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            long spins = 0;
            while (__hip_atomic_load(&B->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1L << 21)) { __hip_atomic_store(&B->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
// a block that leaves after `done` of its launch's `budget` barriers
__device__ __forceinline__ void grid_barrier_leave(GridBar *B, int done, int budget)
{
    if (threadIdx.x == 0 && done < budget)
        __hip_atomic_fetch_add(&B->bar, (unsigned long long)(budget - done), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace sicp

#endif
