// clock_probe.hip — rate of wall_clock64() (s_memrealtime) against HIP events: a one-wave kernel spins for 2,000,000 ticks
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long long ticks, long long* out) { const long long t0 = wall_clock64(); long long t; do { t = wall_clock64(); } while (t - t0 < ticks); out[0] = t - t0; }
int main() {
    long long* out; hipMalloc(&out, 8);
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, 1000LL, out); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, 2000000LL, out); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("hipDeviceAttributeWallClockRate %d kHz; 2,000,000 ticks took %.3f ms -> %.1f MHz\n", rate, ms, 2000.0 / ms);
    return 0;
}
