// clock_probe.hip — which device clocks tick on this box: wall_clock64() (s_memrealtime, 100 MHz), clock64() / s_memtime
// (shader clock), read twice around a dependent delay.   hipcc --offload-arch=gfx950 -O2 -o clock_probe clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned long long* out, int spin) {
    unsigned long long w0 = wall_clock64(), c0 = clock64(), r0 = __builtin_readcyclecounter();
    float x = (float)threadIdx.x;
    for (int i = 0; i < spin; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
    unsigned long long w1 = wall_clock64(), c1 = clock64(), r1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        out[0] = w0; out[1] = w1; out[2] = c0; out[3] = c1; out[4] = r0; out[5] = r1;
        out[6] = (unsigned long long)x;
    }
}
int main() {
    unsigned long long* d;
    unsigned long long h[7];
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 1;
    for (int spin : {1000, 100000}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, spin);
        if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        printf("spin %6d: wall_clock64 %llu -> %llu (d %llu) | clock64 %llu -> %llu (d %llu) | readcyclecounter d %llu\n", spin, h[0],
               h[1], h[1] - h[0], h[2], h[3], h[3] - h[2], h[5] - h[4]);
    }
    int rate = 0;
    hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    printf("hipDeviceAttributeWallClockRate = %d kHz\n", rate);
    return 0;
}
