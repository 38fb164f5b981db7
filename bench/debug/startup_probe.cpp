// startup_probe.cpp — where a short-lived HIP program's fixed cost goes (debug aid for bin/hw5's wall time)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include "../../include/nbody_amd.h"
static double ms(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
int main() {
    auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    hipGetDeviceCount(&n);
    printf("hipGetDeviceCount      %7.1f ms (devices %d)\n", ms(t0), n);
    hipSetDevice(0);
    hipFree(nullptr);
    printf("hipSetDevice+hipFree0  %7.1f ms\n", ms(t0));
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    printf("stream create          %7.1f ms\n", ms(t0));
    void* p;
    hipMalloc(&p, 1 << 20);
    printf("first hipMalloc        %7.1f ms\n", ms(t0));
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("hipGetDeviceProperties %7.1f ms\n", ms(t0));
    {
        auto t1 = std::chrono::steady_clock::now();
        hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        printf("  second stream create   +%.2f ms\n", ms(t1)); t1 = std::chrono::steady_clock::now();
        hipEvent_t e; hipEventCreate(&e);
        printf("  event create           +%.2f ms\n", ms(t1)); t1 = std::chrono::steady_clock::now();
        void* d; hipMalloc(&d, 200 * 14 * 8 + 512);
        printf("  small hipMalloc        +%.2f ms\n", ms(t1)); t1 = std::chrono::steady_clock::now();
        void* h; hipHostMalloc(&h, 256);
        printf("  small hipHostMalloc    +%.2f ms\n", ms(t1)); t1 = std::chrono::steady_clock::now();
        void* h2; hipHostMalloc(&h2, 256);
        printf("  second hipHostMalloc   +%.2f ms\n", ms(t1)); t1 = std::chrono::steady_clock::now();
        hipHostFree(h); hipHostFree(h2); hipFree(d);
        printf("  frees                  +%.2f ms\n", ms(t1)); t1 = std::chrono::steady_clock::now();
        hipStreamDestroy(s2); hipEventDestroy(e);
        printf("  stream/event destroy   +%.2f ms\n", ms(t1));
    }
    nb_config cfg;
    nb_config_default(&cfg);
    cfg.n = 200;
    nb_context* c = nullptr;
    nb_create(&c, &cfg);
    printf("nb_create #1           %7.1f ms\n", ms(t0));
    nb_context* c2 = nullptr;
    nb_create(&c2, &cfg);
    printf("nb_create #2           %7.1f ms\n", ms(t0));
    double q[3][200] = {}, v[3][200] = {}, m[200];
    for (int i = 0; i < 200; ++i) { q[0][i] = i * 1e9; m[i] = 1e20; }
    nb_set_state(c, q[0], q[1], q[2], v[0], v[1], v[2], m, nullptr);
    printf("nb_set_state           %7.1f ms\n", ms(t0));
    nb_step(c, 1, 1);
    printf("first nb_step (module) %7.1f ms\n", ms(t0));
    nb_step(c, 2, 1000);
    printf("1000 more steps        %7.1f ms\n", ms(t0));
    nb_destroy(c);
    nb_destroy(c2);
    printf("destroy                %7.1f ms\n", ms(t0));
    return 0;
}
