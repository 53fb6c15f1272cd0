// Can the NEXT batch's generator run in the tail of THIS batch's walk?  Stream A: a long kernel W (4096 workgroups that spin
// ~20 us each: ~16 rounds on 256 CUs, then a tail) and, behind an event of stream B, a second long kernel W'.  Stream B: a
// one-wave GATE kernel that waits for a word W's workgroup number `trigger` stores when it starts (workgroups are dealt in
// order: by then only the launch's last rounds remain), then a short kernel G (256 workgroups x 16 waves spinning ~8 us: the
// generator's shape), then the event.  Measured with the device's wall clock: when G started and ended relative to W's last
// stamp, and how long after W's last stamp W' began - against the same three kernels in ONE stream (W, G, W').
//   hipcc --offload-arch=gfx950 -O2 tools/tail_gate_probe.hip -o tools/tail_gate_probe.bin && gpurun -- ./tools/tail_gate_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void walk_like(unsigned long long *stamps, unsigned ticks, unsigned *flag, unsigned trigger, unsigned seq)
{
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0)
    {
        stamps[2 * blockIdx.x] = t0;
        if (flag && blockIdx.x == trigger)
            __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (unsigned trips = 0; trips < (1u << 20) && wall_clock64() - t0 < ticks; trips++)
        __builtin_amdgcn_s_sleep(16);
    if (threadIdx.x == 0)
        stamps[2 * blockIdx.x + 1] = wall_clock64();
}
__global__ void gate(const unsigned *flag, unsigned seq)
{
    for (unsigned trips = 0; trips < (1u << 22) && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq; trips++)
        __builtin_amdgcn_s_sleep(32);
}

#define CHK(x)                                                                                                         \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e = (x);                                                                                            \
        if (e != hipSuccess)                                                                                           \
        {                                                                                                              \
            printf("%s: %s\n", #x, hipGetErrorString(e));                                                              \
            return 1;                                                                                                  \
        }                                                                                                              \
    } while (0)

static void span(const std::vector<unsigned long long> &h, unsigned n, unsigned long long &first, unsigned long long &last)
{
    first = ~0ull;
    last = 0;
    for (unsigned i = 0; i < n; i++)
    {
        first = h[2 * i] < first ? h[2 * i] : first;
        last = h[2 * i + 1] > last ? h[2 * i + 1] : last;
    }
}

int main()
{
    const unsigned NW = 4096, NG = 256;
    unsigned long long *d_w, *d_g, *d_w2;
    unsigned *d_flag;
    CHK(hipMalloc(&d_w, NW * 16));
    CHK(hipMalloc(&d_w2, NW * 16));
    CHK(hipMalloc(&d_g, NG * 16));
    CHK(hipMalloc(&d_flag, 64));
    CHK(hipMemset(d_flag, 0, 64));
    hipStream_t a, b;
    CHK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipEvent_t ev;
    CHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    std::vector<unsigned long long> hw(2 * NW), hg(2 * NG), hw2(2 * NW);
    unsigned seq = 0;
    for (int mode = 0; mode <= 3; mode++) // 0: one stream; 1-3: gated side stream, trigger 3072 / 2048 / 1024 workgroups before W's end
        for (int rep = 0; rep < 3; rep++)
        {
            seq++;
            const unsigned trigger = mode == 0 ? NW : NW - 1024u * (4u - (unsigned)mode);
            hipLaunchKernelGGL(walk_like, dim3(NW), dim3(256), 0, a, d_w, 2000u, mode ? d_flag : nullptr, trigger, seq);
            if (mode == 0)
                hipLaunchKernelGGL(walk_like, dim3(NG), dim3(1024), 40 * 1024, a, d_g, 800u, nullptr, 0u, 0u);
            else
            {
                hipLaunchKernelGGL(gate, dim3(1), dim3(64), 0, b, d_flag, seq);
                hipLaunchKernelGGL(walk_like, dim3(NG), dim3(1024), 40 * 1024, b, d_g, 800u, nullptr, 0u, 0u);
                CHK(hipEventRecord(ev, b));
                CHK(hipStreamWaitEvent(a, ev, 0));
            }
            hipLaunchKernelGGL(walk_like, dim3(NW), dim3(256), 0, a, d_w2, 2000u, nullptr, 0u, 0u);
            CHK(hipStreamSynchronize(a));
            CHK(hipStreamSynchronize(b));
            CHK(hipMemcpy(hw.data(), d_w, NW * 16, hipMemcpyDeviceToHost));
            CHK(hipMemcpy(hg.data(), d_g, NG * 16, hipMemcpyDeviceToHost));
            CHK(hipMemcpy(hw2.data(), d_w2, NW * 16, hipMemcpyDeviceToHost));
            unsigned long long w0, w1, g0, g1, x0, x1;
            span(hw, NW, w0, w1);
            span(hg, NG, g0, g1);
            span(hw2, NW, x0, x1);
            printf("mode %d (trigger at workgroup %u): W %.1f us; G from %+.1f to %+.1f us relative to W's end (%.1f us long); W' starts %+.1f us after "
                   "W's end; W start -> W' end %.1f us\n",
                   mode, trigger, (w1 - w0) / 100.0, ((long long)g0 - (long long)w1) / 100.0, ((long long)g1 - (long long)w1) / 100.0,
                   (g1 - g0) / 100.0, ((long long)x0 - (long long)w1) / 100.0, (x1 - w0) / 100.0);
        }
    return 0;
}
