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
// What three measured versions taught (Q = 1 M, 256 blocks, a phase = one pass over 9 MB):
//   * one counter that everybody adds to AND polls: the pollers queue up in front of the arrivals -- a phase cost 21 us against
//     11 us for a kernel boundary;
//   * release / acquire FENCES in the barrier (buffer_wbl2 / buffer_inv: every block writes its L2 back and drops it, 256 times
//     per barrier) cost more than the barrier -- so there are none: blocks exchange data through agent-scope atomic loads and
//     stores (performed at the coherence point), and whatever they read again in the next phase stays in their L2;
//   * 256 same-address read-modify-writes serialise at ~20 ns apiece (5 us): arrivals are counted in 8 groups (blockIdx & 7, a
//     line each) whose last member reports to a top counter -- 32 + 8 in a row instead of 256.
// The counters wrap to zero at their last arrival (atomicInc), the last arriver of all then publishes the barrier's number in
// `gen`, the only word the waiting blocks poll.  Barrier numbers grow monotonically over the life of the buffer: the host hands
// every launch the number it starts from (launch index * the kernel's barrier budget) -- nothing to reset between launches, and a
// launch that leaves early leaves nothing behind.  Polling is bounded (about two seconds): a launch that cannot meet itself
// flags an error and goes on instead of hanging the queue.
struct GridBar {
    struct Line { unsigned v; unsigned pad[31]; };
    Line sub[8];                   // arrivals per group, 0 between barriers
    Line top;                      // groups that are complete, 0 between barriers
    unsigned long long gen;        // number of the last barrier every block reached
    unsigned long long pad1[15];
    unsigned error;                // a wait timed out: stays set for the rest of the buffer's life; the host reports it and starts over from a zeroed buffer (reset_barrier_state)
    unsigned pad2[31];
};
// all threads call it; every cross-block datum published before must have been written with agent-scope atomics
// `absent`: blocks the barrier is told to expect beyond those launched (0 in the product; tests/test_gpu_kernels.py asks for one
// to see the timeout path end in an error instead of a hang)
__device__ __forceinline__ void grid_barrier(GridBar *B, unsigned long long number, unsigned absent = 0u)
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        // (up to 64 blocks report to ONE group: the second level costs a dependent round trip that the serialisation of so few
        // arrivals does not)
        const unsigned g = gridDim.x + absent, flat = g <= 64u ? 1u : 0u, c = flat ? 0u : (blockIdx.x & 7u);
        const unsigned n_c = flat ? g : ((g + 7u - c) >> 3), groups = flat ? 1u : 8u;
        bool opener = false;
        // arrivals are counted with a WRAPPING increment (atomicInc: back to 0 at the last arrival), so the last arriver has nothing
        // to reset before it reports upwards -- round 5 stored 0 and drained that store first: one dependent device-scope round trip
        // per level, ~1 us each (profiles/r6)
        if (atomicInc(&B->sub[c].v, n_c - 1u) == n_c - 1u) {
            if (flat || atomicInc(&B->top.v, groups - 1u) == groups - 1u) {
                __hip_atomic_store(&B->gen, number, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                opener = true;
            }
        }
        if (!opener) {
            long spins = 0;
            while (__hip_atomic_load(&B->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < number) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > (1L << 21)) { __hip_atomic_store(&B->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                // a barrier of this buffer has already given up: the launch is lost, do not wait two seconds at each of its phases
                if ((spins & 1023) == 0 && __hip_atomic_load(&B->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            }
        }
    }
    asm volatile("" ::: "memory");
    __syncthreads();
}

}  // namespace sicp

#endif
