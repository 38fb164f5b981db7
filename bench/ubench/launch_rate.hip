// launch_rate.hip — what one dependent kernel launch costs on this box as a function of the kernarg size and of how the
// launches are issued (eager back-to-back vs a hipGraph of N kernel nodes replayed).  Decides how the per-step scenario
// engine (K2, csrc/nbody_kernels_f64.hip) should be driven: it is bound by this, not by arithmetic.
//   hipcc --offload-arch=gfx950 -O3 -o launch_rate launch_rate.hip && ./launch_rate [graph_nodes=2000 [quick]]
// `quick` runs one configuration only (64-byte kernarg, one workgroup): the repro used to bracket the node count from which
// rocprofv3's kernel tracing crashes inside hipGraphLaunch (profiles/r03_graph_trace_limit.txt).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int BYTES>
struct Args {
    double* p;
    int step;
    char pad[BYTES - 16];
};

template <int BYTES>
__global__ __launch_bounds__(256) void touch(Args<BYTES> a) {
    // a minimal dependent chain: one load, one store (like a step kernel's state), nothing else
    const int i = blockIdx.x * 256 + threadIdx.x;
    a.p[i] = a.p[i] + (double)a.pad[BYTES - 17] + 1.0;
}

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e = (x);                                                   \
        if (e != hipSuccess) {                                                \
            printf("%s failed: %s\n", #x, hipGetErrorString(e));              \
            return 1;                                                         \
        }                                                                     \
    } while (0)

static int g_nodes = 2000;

template <int BYTES>
int run(hipStream_t s, double* buf, int blocks, int n) {
    Args<BYTES> a{};
    a.p = buf;
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(touch<BYTES>, dim3(blocks), dim3(256), 0, s, a);
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) {
        a.step = i;
        hipLaunchKernelGGL(touch<BYTES>, dim3(blocks), dim3(256), 0, s, a);
    }
    auto t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(s));
    auto t2 = std::chrono::steady_clock::now();
    const double issue = std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
    const double total = std::chrono::duration<double, std::micro>(t2 - t0).count() / n;
    // graph: 2000 kernel nodes captured once, replayed n/2000 times
    const int G = g_nodes;
    hipGraph_t graph;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < G; ++i) {
        a.step = i;
        hipLaunchKernelGGL(touch<BYTES>, dim3(blocks), dim3(256), 0, s, a);
    }
    CK(hipStreamEndCapture(s, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(exec, s));
    CK(hipStreamSynchronize(s));
    auto g0 = std::chrono::steady_clock::now();
    for (int r = 0; r < n / G; ++r) CK(hipGraphLaunch(exec, s));
    CK(hipStreamSynchronize(s));
    auto g1 = std::chrono::steady_clock::now();
    const double graph_us = std::chrono::duration<double, std::micro>(g1 - g0).count() / (n / G * G);
    printf("kernarg %4d B, %3d workgroups: eager issue %.2f us/launch, eager total %.2f us/launch, graph(%d nodes) %.2f us/launch\n",
           BYTES, blocks, issue, total, G, graph_us);
    (void)hipGraphExecDestroy(exec);
    (void)hipGraphDestroy(graph);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1) g_nodes = atoi(argv[1]);
    if (g_nodes < 1 || g_nodes > 40000) return 2;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    double* buf;
    CK(hipMalloc(&buf, 1024 * 256 * sizeof(double)));
    CK(hipMemset(buf, 0, 1024 * 256 * sizeof(double)));
    const int n = 40000;
    if (argc > 2) return run<64>(s, buf, 1, n);
    for (int blocks : {1, 64, 256, 1024}) {
        if (run<64>(s, buf, blocks, n)) return 1;
        if (run<256>(s, buf, blocks, n)) return 1;
        if (run<1024>(s, buf, blocks, n)) return 1;
        if (run<2048>(s, buf, blocks, n)) return 1;
    }
    return 0;
}
