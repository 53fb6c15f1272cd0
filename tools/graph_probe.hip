// tools/graph_probe.hip - what does a step's launch chain cost the HOST as four stream launches (two streams, two
// events) and as ONE graph launch with the kernel nodes' parameters updated every step?  (annealing step of one chain:
// commit walk | table rebuild on a side stream, then generator, then scoring walk)
//   hipcc --offload-arch=gfx950 -O2 tools/graph_probe.hip -o gpurun_out/graph_probe && gpurun_out/graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Args { unsigned long long *out; unsigned int spin, tag; unsigned int pad[40]; }; // ~ a WalkArgs-sized struct
__global__ void work(const Args a)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < a.spin) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) a.out[a.tag & 7] = t0;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t s, side;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    unsigned long long *out;
    CK(hipMalloc(&out, 64));
    volatile unsigned int *flag;
    CK(hipHostMalloc((void **)&flag, 64, hipHostMallocMapped));
    const int steps = 2000;
    const unsigned int spin = 300; // 100 MHz clock: 3 us per kernel
    auto args = [&](unsigned tag) { Args a{}; a.out = out; a.spin = spin; a.tag = tag; return a; };
    // ---- streams: commit (main) | rebuild (side), gen waits for both, walk
    for (int rep = 0; rep < 2; rep++)
    {
        double host = 0;
        CK(hipStreamSynchronize(s));
        const double t0 = now();
        for (int i = 0; i < steps; i++)
        {
            const double h0 = now();
            CK(hipEventRecord(e0, s));
            CK(hipStreamWaitEvent(side, e0, 0));
            hipLaunchKernelGGL(work, dim3(32), dim3(256), 0, side, args(i));
            CK(hipEventRecord(e1, side));
            hipLaunchKernelGGL(work, dim3(32), dim3(256), 0, s, args(i + 1));
            CK(hipStreamWaitEvent(s, e1, 0));
            hipLaunchKernelGGL(work, dim3(256), dim3(1024), 0, s, args(i + 2));
            hipLaunchKernelGGL(work, dim3(512), dim3(1024), 0, s, args(i + 3));
            host += now() - h0;
            CK(hipStreamSynchronize(s)); // (the library polls a flag instead; same for both variants here)
        }
        const double dt = now() - t0;
        if (rep)
            printf("streams: %.1f us per step, host enqueue %.1f us\n", dt / steps * 1e6, host / steps * 1e6);
    }
    // ---- one graph: the same four nodes, parameters updated per step
    hipGraph_t g;
    CK(hipGraphCreate(&g, 0));
    hipGraphNode_t n[4];
    Args a4[4] = {args(0), args(1), args(2), args(3)};
    void *kp[4][1] = {{&a4[0]}, {&a4[1]}, {&a4[2]}, {&a4[3]}};
    hipKernelNodeParams p[4];
    const dim3 grids[4] = {dim3(32), dim3(32), dim3(256), dim3(512)}, blocks[4] = {dim3(256), dim3(256), dim3(1024), dim3(1024)};
    for (int k = 0; k < 4; k++)
    {
        p[k] = hipKernelNodeParams{};
        p[k].func = (void *)work;
        p[k].gridDim = grids[k];
        p[k].blockDim = blocks[k];
        p[k].kernelParams = kp[k];
    }
    CK(hipGraphAddKernelNode(&n[0], g, nullptr, 0, &p[0]));       // rebuild
    CK(hipGraphAddKernelNode(&n[1], g, nullptr, 0, &p[1]));       // commit
    hipGraphNode_t dep01[2] = {n[0], n[1]};
    CK(hipGraphAddKernelNode(&n[2], g, dep01, 2, &p[2]));          // gen after both
    CK(hipGraphAddKernelNode(&n[3], g, &n[2], 1, &p[3]));          // walk
    hipGraphExec_t ge;
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int variant = 0; variant < 2; variant++) // 0: all four nodes updated; 1: none (parameters through memory)
        for (int rep = 0; rep < 2; rep++)
        {
            double host = 0;
            CK(hipStreamSynchronize(s));
            const double t0 = now();
            for (int i = 0; i < steps; i++)
            {
                const double h0 = now();
                if (variant == 0)
                    for (int k = 0; k < 4; k++)
                    {
                        a4[k].tag = i + k;
                        p[k].gridDim = dim3(grids[k].x + (i & 1)); // the batch size changes too
                        CK(hipGraphExecKernelNodeSetParams(ge, n[k], &p[k]));
                    }
                CK(hipGraphLaunch(ge, s));
                host += now() - h0;
                CK(hipStreamSynchronize(s));
            }
            const double dt = now() - t0;
            if (rep)
                printf("graph (%s): %.1f us per step, host enqueue %.1f us\n", variant == 0 ? "4 nodes updated" : "launch only",
                       dt / steps * 1e6, host / steps * 1e6);
        }
    // ---- plain: four launches in one stream, no events
    for (int rep = 0; rep < 2; rep++)
    {
        double host = 0;
        CK(hipStreamSynchronize(s));
        const double t0 = now();
        for (int i = 0; i < steps; i++)
        {
            const double h0 = now();
            hipLaunchKernelGGL(work, dim3(32), dim3(256), 0, s, args(i));
            hipLaunchKernelGGL(work, dim3(32), dim3(256), 0, s, args(i + 1));
            hipLaunchKernelGGL(work, dim3(256), dim3(1024), 0, s, args(i + 2));
            hipLaunchKernelGGL(work, dim3(512), dim3(1024), 0, s, args(i + 3));
            host += now() - h0;
            CK(hipStreamSynchronize(s));
        }
        const double dt = now() - t0;
        if (rep)
            printf("one stream, four launches: %.1f us per step, host enqueue %.1f us\n", dt / steps * 1e6, host / steps * 1e6);
    }
    return 0;
}
