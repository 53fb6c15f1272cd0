// gather.hpp - device code of the post launch (propose_kernels.hip): one wave copies one picked candidate's descriptor
// and rewrites into pinned host memory.
#pragma once
#include "kernels.hpp"

namespace lvbgpu
{

// this wave's memory operations acknowledged (performed at the level their scope names) before what follows: all a
// hand-over through agent- / system-scope atomics needs - see atomics_acknowledged() in fitch_kernels.hip
__device__ __forceinline__ void gather_acknowledged()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// pick j by one wave (lane = 0..63): word by word, as system-scope stores (written through to the host)
__device__ __forceinline__ void gather_one_pick(const GatherArgs &a, uint32_t j, uint32_t lane)
{
    const uint32_t g = a.pick_idx[j];
    const ProposalInfo pi = a.info[g];
    uint32_t *dst = reinterpret_cast<uint32_t *>(a.out + (size_t)j * a.out_stride);
    constexpr uint32_t PI_WORDS = sizeof(ProposalInfo) / 4u;
    static_assert(sizeof(ProposalInfo) % 4u == 0 && sizeof(lvbgpu_edit_dev) == 12, "records are copied word by word");
    if (lane < PI_WORDS)
        __hip_atomic_store(dst + lane, reinterpret_cast<const uint32_t *>(a.info + g)[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    uint32_t ne = pi.overflow ? 0u : (uint32_t)pi.n_edits;
    if (ne > a.stride_e)
        ne = a.stride_e;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(a.edits + (size_t)g * a.stride_e);
    for (uint32_t i = lane; i < 3u * ne; i += 64u)
        __hip_atomic_store(dst + PI_WORDS + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    gather_acknowledged();
    if (lane == 0 && atomicAdd(a.arrived, 1u) == a.k - 1u)
    {
        __hip_atomic_store(a.arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.flag, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

} // namespace lvbgpu
