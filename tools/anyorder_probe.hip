// Does hipExtAnyOrderLaunch let a kernel start before the kernel in front of it (same stream) has finished?  (hip_ext.h
// says the flag is "not supported on AMD GFX9xx boards"; measured, not assumed.)
// A: 4096 workgroups that each spin ~20 us (a launch of ~16 rounds on 256 CUs); B: one workgroup that stamps the clock.
// Printed: when B started relative to A's first and last stamps, with and without the flag.
//   hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o /tmp/anyorder_probe && /tmp/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

__global__ void spin_kernel(unsigned long long *stamps, unsigned ticks)
{
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0)
        stamps[2 * blockIdx.x] = t0;
    for (unsigned trips = 0; trips < (1u << 20) && wall_clock64() - t0 < ticks; trips++)
        __builtin_amdgcn_s_sleep(16);
    if (threadIdx.x == 0)
        stamps[2 * blockIdx.x + 1] = wall_clock64();
}
__global__ void stamp_kernel(unsigned long long *out)
{
    if (threadIdx.x == 0)
        out[0] = wall_clock64();
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

int main()
{
    const unsigned NB = 4096;
    unsigned long long *d_stamps, *d_b;
    CHK(hipMalloc(&d_stamps, NB * 16));
    CHK(hipMalloc(&d_b, 8));
    hipStream_t s;
    CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<unsigned long long> h(2 * NB);
    for (int flag = 0; flag <= 1; flag++)
        for (int rep = 0; rep < 3; rep++)
        {
            hipLaunchKernelGGL(spin_kernel, dim3(NB), dim3(256), 0, s, d_stamps, 2000u); // 20 us at 100 MHz
            void *args[] = {&d_b};
            CHK(hipExtLaunchKernel(reinterpret_cast<const void *>(&stamp_kernel), dim3(1), dim3(64), args, 0, s, nullptr, nullptr,
                                   flag ? hipExtAnyOrderLaunch : 0));
            CHK(hipStreamSynchronize(s));
            unsigned long long b = 0;
            CHK(hipMemcpy(h.data(), d_stamps, NB * 16, hipMemcpyDeviceToHost));
            CHK(hipMemcpy(&b, d_b, 8, hipMemcpyDeviceToHost));
            unsigned long long first = ~0ull, last = 0;
            for (unsigned i = 0; i < NB; i++)
            {
                first = h[2 * i] < first ? h[2 * i] : first;
                last = h[2 * i + 1] > last ? h[2 * i + 1] : last;
            }
            printf("any-order %d: A ran %.1f us; B started %.1f us after A's first stamp, %.1f us %s A's last\n", flag,
                   (last - first) / 100.0, ((long long)b - (long long)first) / 100.0, (b > last ? b - last : last - b) / 100.0,
                   b > last ? "AFTER" : "BEFORE");
        }
    return 0;
}
