// tools/l2_probe3.hip - which per-token cost keeps fitch_walk below the pure-load ceiling?
// Same access pattern as l2_probe.hip (ring of 4), plus per load: NV dependent VALU ops, NS scalar
// ALU ops and NB never-taken scalar branches (inline asm, so the compiler keeps them).
// Build: hipcc --offload-arch=gfx950 -O3 tools/l2_probe3.hip -o tools/l2_probe3.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NV, int NS, int NB>
__global__ __launch_bounds__(256) void probe(const uint4 *rows, uint32_t stride4, uint32_t nrows, uint32_t ntiles,
                                             uint32_t ngroups, uint32_t B, uint32_t ntok, uint4 *sink)
{
    constexpr int RING = 4;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nblk = gridDim.x;
    const uint32_t pos = (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3);
    const uint32_t item = pos * 4 + wave;
    if (item >= B * ngroups)
        return;
    const uint32_t group = item / B, cand = item - group * B;
    const uint32_t t0 = group * ntiles / ngroups, t1 = (group + 1) * ntiles / ngroups;
    uint4 acc = make_uint4(1, 2, 3, 4);
    uint32_t seed = cand * 2654435761u + 12345u;
    uint32_t sdummy = cand;
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
#pragma unroll
                for (int v = 0; v < NV; v++)
                    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xd4" : "+v"(acc.x) : "v"(acc.y), "v"(ring[q].z));
#pragma unroll
                for (int k = 0; k < NS; k++)
                    asm volatile("s_add_u32 %0, %0, 1" : "+s"(sdummy) : : "scc");
#pragma unroll
                for (int k = 0; k < NB; k++)
                    asm volatile("s_cmp_eq_u32 %0, -1\n\ts_cbranch_scc1 1f\n1:" : : "s"(sdummy) : "scc");
                s = s * 1664525u + 1013904223u;
                ring[q] = base[(size_t)((s >> 8) % nrows) * stride4];
            }
        }
#pragma unroll
        for (int q = 0; q < RING; q++) { acc.x ^= ring[q].x; acc.y ^= ring[q].y; acc.z ^= ring[q].z; acc.w ^= ring[q].w; }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u && sdummy == 77u)
        sink[0] = acc;
}

template <int NV, int NS, int NB>
double run(const uint4 *d, uint32_t stride4, uint32_t nrows, uint32_t ntiles, uint32_t ngroups, uint32_t B,
           uint32_t ntok, uint4 *sink)
{
    uint32_t nblk = ((B * ngroups + 3) / 4 + 7) & ~7u;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int i = 0; i < 5; i++)
        hipLaunchKernelGGL((probe<NV, NS, NB>), dim3(nblk), dim3(256), 0, 0, d, stride4, nrows, ntiles, ngroups, B, ntok, sink);
    (void)hipEventRecord(a);
    const int reps = 50;
    for (int i = 0; i < reps; i++)
        hipLaunchKernelGGL((probe<NV, NS, NB>), dim3(nblk), dim3(256), 0, 0, d, stride4, nrows, ntiles, ngroups, B, ntok, sink);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    const double loads = (double)B * ntiles * (ntok / 4 * 4 + 4);
    return loads * 1024.0 / (ms / reps * 1e-3) / 1e12;
}

int main()
{
    const uint32_t nrows = 997, ntiles = 25, stride4 = ntiles * 64, B = 4096, ntok = 24, G = 13;
    uint4 *d, *sink;
    (void)hipMalloc(&d, (size_t)nrows * stride4 * 16);
    (void)hipMalloc(&sink, 16);
    (void)hipMemset(d, 0x5a, (size_t)nrows * stride4 * 16);
#define R(NV, NS, NB) printf("VALU+%-2d SALU+%-2d BR+%d : %.1f TB/s\n", NV, NS, NB, run<NV, NS, NB>(d, stride4, nrows, ntiles, G, B, ntok, sink));
    R(0, 0, 0) R(4, 0, 0) R(8, 0, 0) R(12, 0, 0) R(16, 0, 0)
    R(8, 4, 0) R(8, 8, 0) R(8, 16, 0)
    R(8, 0, 1) R(8, 0, 2) R(8, 0, 4)
    R(8, 4, 2) R(8, 8, 2) R(0, 8, 2) R(0, 16, 4)
    return 0;
}
