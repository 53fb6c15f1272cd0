// tools/l2_probe5.hip - (main) does the kind of allocation decide whether the L2s keep the rows across launches?
// (kernel of l2_probe4.hip) does the SKEW of the walk's row reads (53 % of an SPR batch's reads go to the 64 rows nearest
// the root) lower what the L2 -> CU path delivers?  The pattern of l2_probe.hip (one wave per
// (tile, candidate), ring of 4 x 1 KiB loads, 8 dependent VALU ops per load), with a share HOTP/256 of the reads
// redirected to the first NHOT rows.
// Build: hipcc --offload-arch=gfx950 -O3 tools/l2_probe5.hip -o tools/l2_probe4.bin
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
    // does the KIND of allocation decide whether the XCD L2s keep the rows from one launch to the next?
    const uint32_t nrows = 997, ntiles = 25, stride4 = ntiles * 64, ntok = 24;
    const size_t bytes = (size_t)nrows * stride4 * 16;
    uint4 *sink;
    (void)hipMalloc(&sink, 16);
    const char *names[3] = {"hipMalloc", "hipExtMallocWithFlags(Finegrained)", "hipExtMallocWithFlags(Uncached)"};
    const unsigned flags[3] = {hipDeviceMallocDefault, hipDeviceMallocFinegrained, hipDeviceMallocUncached};
    for (int k = 0; k < 3; k++)
    {
        uint4 *d = nullptr;
        hipError_t e = k == 0 ? hipMalloc(&d, bytes) : hipExtMallocWithFlags((void **)&d, bytes, flags[k]);
        if (e != hipSuccess)
        {
            printf("%s: %s\n", names[k], hipGetErrorString(e));
            continue;
        }
        (void)hipMemset(d, 0x5a, bytes);
        (void)hipDeviceSynchronize();
        for (int rep = 0; rep < 2; rep++)
            printf("%-36s B = 4096: %.1f TB/s   B = 16384: %.1f TB/s\n", names[k], run(d, stride4, nrows, ntiles, 25, 4096, ntok, 1, 0, sink),
                   run(d, stride4, nrows, ntiles, 5, 16384, ntok, 1, 0, sink));
        (void)hipFree(d);
    }
    return 0;
}
