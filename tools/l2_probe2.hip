// tools/l2_probe2.hip - does L1 residency of hot rows raise the load ceiling?  (see l2_probe.hip)
// hotfrac of the loads go to `nhot` rows (the top of the tree in real programs); cold loads
// optionally non-temporal (bypass L1) so they do not evict the hot lines.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int RING, bool NT>
__global__ __launch_bounds__(256) void probe(const uint4 *rows, uint32_t stride4, uint32_t nrows, uint32_t ntiles,
                                             uint32_t ngroups, uint32_t B, uint32_t ntok, uint32_t nhot,
                                             uint32_t hot_per_256, uint4 *sink)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nblk = gridDim.x;
    const uint32_t pos = (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3);
    const uint32_t item = pos * 4 + wave;
    if (item >= B * ngroups)
        return;
    const uint32_t group = item / B, cand = item - group * B;
    const uint32_t t0 = group * ntiles / ngroups, t1 = (group + 1) * ntiles / ngroups;
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint32_t seed = cand * 2654435761u + 12345u;
    auto ld = [&](const uint4 *base, uint32_t &s) -> uint4 {
        s = s * 1664525u + 1013904223u;
        const bool hot = ((s >> 24) & 255u) < hot_per_256;
        const uint32_t r = hot ? (s >> 8) % nhot : nhot + (s >> 8) % (nrows - nhot);
        const uint4 *p = base + (size_t)r * stride4;
        if (NT && !hot)
        {
            typedef unsigned int v4u __attribute__((ext_vector_type(4)));
            const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p));
            return make_uint4(v.x, v.y, v.z, v.w);
        }
        return *p;
    };
    for (uint32_t t = t0; t < t1; t++)
    {
        const uint4 *base = rows + t * 64u + lane;
        uint32_t s = seed;
        uint4 ring[RING];
#pragma unroll
        for (int q = 0; q < RING; q++)
            ring[q] = ld(base, s);
        for (uint32_t j = 0; j + RING <= ntok; j += RING)
        {
#pragma unroll
            for (int q = 0; q < RING; q++)
            {
                acc.x ^= ring[q].x; acc.y ^= ring[q].y; acc.z ^= ring[q].z; acc.w ^= ring[q].w;
                ring[q] = ld(base, s);
            }
        }
#pragma unroll
        for (int q = 0; q < RING; q++) { acc.x ^= ring[q].x; acc.y ^= ring[q].y; acc.z ^= ring[q].z; acc.w ^= ring[q].w; }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u)
        sink[0] = acc;
}

template <bool NT>
double run(const uint4 *d, uint32_t stride4, uint32_t nrows, uint32_t ntiles, uint32_t ngroups, uint32_t B,
           uint32_t ntok, uint32_t nhot, uint32_t hot256, uint4 *sink)
{
    uint32_t nblk = ((B * ngroups + 3) / 4 + 7) & ~7u;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int i = 0; i < 5; i++)
        hipLaunchKernelGGL((probe<4, NT>), dim3(nblk), dim3(256), 0, 0, d, stride4, nrows, ntiles, ngroups, B, ntok, nhot, hot256, sink);
    (void)hipEventRecord(a);
    const int reps = 50;
    for (int i = 0; i < reps; i++)
        hipLaunchKernelGGL((probe<4, NT>), dim3(nblk), dim3(256), 0, 0, d, stride4, nrows, ntiles, ngroups, B, ntok, nhot, hot256, sink);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    const double loads = (double)B * ntiles * (ntok / 4 * 4 + 4);
    return loads * 1024.0 / (ms / reps * 1e-3) / 1e12;
}

int main()
{
    const uint32_t nrows = 997, ntiles = 25, stride4 = ntiles * 64, B = 4096, ntok = 24;
    uint4 *d, *sink;
    (void)hipMalloc(&d, (size_t)nrows * stride4 * 16);
    (void)hipMalloc(&sink, 16);
    (void)hipMemset(d, 0x5a, (size_t)nrows * stride4 * 16);
    for (uint32_t ngroups : {25u, 9u})
        for (uint32_t nhot : {8u, 16u, 32u})
            for (uint32_t hot256 : {0u, 100u, 136u, 200u})
                printf("groups %2u nhot %2u hot %.2f: plain %.1f TB/s   cold-nt %.1f TB/s\n", ngroups, nhot, hot256 / 256.0,
                       run<false>(d, stride4, nrows, ntiles, ngroups, B, ntok, nhot, hot256, sink),
                       run<true>(d, stride4, nrows, ntiles, ngroups, B, ntok, nhot, hot256, sink));
    return 0;
}
