// tools/l2_probe4.hip - does the SKEW of the walk's row reads (53 % of an SPR batch's reads go to the 64 rows nearest
// the root) lower what the L2 -> CU path delivers?  The pattern of l2_probe.hip (one wave per
// (tile, candidate), ring of 4 x 1 KiB loads, 8 dependent VALU ops per load), with a share HOTP/256 of the reads
// redirected to the first NHOT rows.
// Build: hipcc --offload-arch=gfx950 -O3 tools/l2_probe4.hip -o tools/l2_probe4.bin
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void probe(const uint4 *rows, uint32_t stride4, uint32_t nrows, uint32_t ntiles,
                                             uint32_t ngroups, uint32_t B, uint32_t ntok, uint32_t nhot, uint32_t hotp,
                                             uint4 *sink)
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
    const uint32_t seed = cand * 2654435761u + 12345u;
    auto pick = [&](uint32_t &s) {
        s = s * 1664525u + 1013904223u;
        const uint32_t r = s >> 8;
        const uint32_t row = ((r & 255u) < hotp) ? (r >> 8) % nhot : (r >> 8) % nrows;
        return (size_t)row * stride4;
    };
    for (uint32_t t = t0; t < t1; t++)
    {
        const uint4 *base = rows + t * 64u + lane;
        uint32_t s = seed;
        uint4 ring[RING];
#pragma unroll
        for (int q = 0; q < RING; q++)
            ring[q] = base[pick(s)];
        for (uint32_t j = 0; j + RING <= ntok; j += RING)
        {
#pragma unroll
            for (int q = 0; q < RING; q++)
            {
                acc.x ^= ring[q].x; acc.y ^= ring[q].y; acc.z ^= ring[q].z; acc.w ^= ring[q].w;
#pragma unroll
                for (int v = 0; v < 8; v++)
                    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xd4" : "+v"(acc.x) : "v"(acc.y), "v"(ring[q].z));
                ring[q] = base[pick(s)];
            }
        }
#pragma unroll
        for (int q = 0; q < RING; q++) { acc.x ^= ring[q].x; acc.y ^= ring[q].y; acc.z ^= ring[q].z; acc.w ^= ring[q].w; }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u)
        sink[0] = acc;
}

double run(const uint4 *d, uint32_t stride4, uint32_t nrows, uint32_t ntiles, uint32_t ngroups, uint32_t B, uint32_t ntok,
           uint32_t nhot, uint32_t hotp, uint4 *sink)
{
    uint32_t nblk = ((B * ngroups + 3) / 4 + 7) & ~7u;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int i = 0; i < 5; i++)
        hipLaunchKernelGGL(probe, dim3(nblk), dim3(256), 0, 0, d, stride4, nrows, ntiles, ngroups, B, ntok, nhot, hotp, sink);
    (void)hipEventRecord(a);
    const int reps = 50;
    for (int i = 0; i < reps; i++)
        hipLaunchKernelGGL(probe, dim3(nblk), dim3(256), 0, 0, d, stride4, nrows, ntiles, ngroups, B, ntok, nhot, hotp, sink);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    const double loads = (double)B * ntiles * (ntok / 4 * 4 + 4);
    return loads * 1024.0 / (ms / reps * 1e-3) / 1e12;
}

int main()
{
    const uint32_t nrows = 997, ntiles = 25, stride4 = ntiles * 64, ntok = 24;
    uint4 *d, *sink;
    (void)hipMalloc(&d, (size_t)nrows * stride4 * 16);
    (void)hipMalloc(&sink, 16);
    (void)hipMemset(d, 0x5a, (size_t)nrows * stride4 * 16);
    for (uint32_t B : {4096u, 16384u})
    {
        const uint32_t G = B == 4096u ? 25u : 5u;
        printf("B = %u, %u groups\n", B, G);
        for (int rep = 0; rep < 2; rep++)
        {
            printf("  uniform                 : %.1f TB/s\n", run(d, stride4, nrows, ntiles, G, B, ntok, 1, 0, sink));
            printf("  53 %% to 64 rows         : %.1f TB/s\n", run(d, stride4, nrows, ntiles, G, B, ntok, 64, 136, sink));
            printf("  53 %% to 16 rows         : %.1f TB/s\n", run(d, stride4, nrows, ntiles, G, B, ntok, 16, 136, sink));
            printf("  25 %% to 4 rows          : %.1f TB/s\n", run(d, stride4, nrows, ntiles, G, B, ntok, 4, 64, sink));
            printf("  90 %% to 64 rows         : %.1f TB/s\n", run(d, stride4, nrows, ntiles, G, B, ntok, 64, 230, sink));
            printf("  100 %% to 8 rows (L1)    : %.1f TB/s\n", run(d, stride4, nrows, ntiles, G, B, ntok, 8, 256, sink));
        }
    }
    return 0;
}
