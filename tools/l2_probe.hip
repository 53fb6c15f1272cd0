// tools/l2_probe.hip - what read bandwidth does the walk's access pattern get from the XCD L2s?
// Same shape as fitch_walk: 997 rows x 25 tiles x 1 KiB, each wave reads `ntok` pseudo-random rows
// of its tile group with `RING` loads in flight, XOR-accumulates (1 VALU per 16 B), no other work.
// Build: hipcc --offload-arch=gfx950 -O3 tools/l2_probe.hip -o gpurun_out/l2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int RING>
__global__ __launch_bounds__(256) void probe(const uint4 *rows, uint32_t stride4, uint32_t nrows, uint32_t ntiles,
                                             uint32_t ngroups, uint32_t B, uint32_t ntok, uint4 *sink)
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
    for (uint32_t t = t0; t < t1; t++)
    {
        const uint4 *base = rows + t * 64u + lane;
        uint32_t s = seed;
        uint4 ring[RING];
#pragma unroll
        for (int q = 0; q < RING; q++)
        {
            s = s * 1664525u + 1013904223u;
            ring[q] = base[(size_t)((s >> 8) % nrows) * stride4];
        }
        for (uint32_t j = 0; j + RING <= ntok; j += RING)
        {
#pragma unroll
            for (int q = 0; q < RING; q++)
            {
                acc.x ^= ring[q].x; acc.y ^= ring[q].y; acc.z ^= ring[q].z; acc.w ^= ring[q].w;
                s = s * 1664525u + 1013904223u;
                ring[q] = base[(size_t)((s >> 8) % nrows) * stride4];
            }
        }
#pragma unroll
        for (int q = 0; q < RING; q++) { acc.x ^= ring[q].x; acc.y ^= ring[q].y; acc.z ^= ring[q].z; acc.w ^= ring[q].w; }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u)
        sink[0] = acc;
}

template <int RING>
double run(const uint4 *d, uint32_t stride4, uint32_t nrows, uint32_t ntiles, uint32_t ngroups, uint32_t B,
           uint32_t ntok, uint4 *sink)
{
    uint32_t nblk = ((B * ngroups + 3) / 4 + 7) & ~7u;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 5; i++)
        hipLaunchKernelGGL(probe<RING>, dim3(nblk), dim3(256), 0, 0, d, stride4, nrows, ntiles, ngroups, B, ntok, sink);
    hipEventRecord(a);
    const int reps = 50;
    for (int i = 0; i < reps; i++)
        hipLaunchKernelGGL(probe<RING>, dim3(nblk), dim3(256), 0, 0, d, stride4, nrows, ntiles, ngroups, B, ntok, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double loads = (double)B * ntiles * (ntok / RING * RING + RING);
    return loads * 1024.0 / (ms / reps * 1e-3) / 1e12;
}

int main()
{
    const uint32_t nrows = 997, ntiles = 25, stride4 = ntiles * 64, B = 4096, ntok = 24;
    uint4 *d, *sink;
    hipMalloc(&d, (size_t)nrows * stride4 * 16);
    hipMalloc(&sink, 16);
    hipMemset(d, 0x5a, (size_t)nrows * stride4 * 16);
    for (uint32_t ngroups : {25u, 13u, 9u, 5u})
    {
        printf("groups %2u: ring2 %.1f TB/s  ring4 %.1f TB/s  ring8 %.1f TB/s\n", ngroups,
               run<2>(d, stride4, nrows, ntiles, ngroups, B, ntok, sink), run<4>(d, stride4, nrows, ntiles, ngroups, B, ntok, sink),
               run<8>(d, stride4, nrows, ntiles, ngroups, B, ntok, sink));
    }
    return 0;
}
