// main_nbody_bench.cpp — a compiled host of the C ABI for large N (no Python, no torch):
//     bin/nbody_bench [N=1048576] [steps=10] [warmup=2] [precision: f32|f32acc64|f64]
// Generates the synthetic bodies of SURVEY §8(d) (same splitmix64 stream as nbody_amd/synthetic.py), uploads them with
// nb_set_state, advances `steps` steps with nb_step_timed (HIP events on the context's stream) and prints pairs/s.
// Shows what a C/C++ caller of libnbody_amd looks like and gives rocprofv3 a target without an interpreter in front.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/nbody_amd.h"

static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline double u01(uint64_t i, int k) { return (double)(splitmix64(42 + 7 * i + (uint64_t)k) >> 11) * (1.0 / 9007199254740992.0); }

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : (1L << 20);
    const int steps = argc > 2 ? atoi(argv[2]) : 10;
    const int warmup = argc > 3 ? atoi(argv[3]) : 2;
    const char* prec = argc > 4 ? argv[4] : "f32";
    nb_config cfg;
    nb_config_default(&cfg);
    cfg.n = (int32_t)n;
    cfg.precision = !strcmp(prec, "f64") ? NB_F64 : !strcmp(prec, "f32acc64") ? NB_F32_ACC64 : NB_F32;
    cfg.dt = 1e-4;  // G, eps stay the reference's values
    std::vector<double> q(3 * n), v(3 * n), m(n);
    for (long i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) q[k * n + i] = 2.0 * u01(i, k) - 1.0;
        for (int k = 0; k < 3; ++k) v[k * n + i] = (2.0 * u01(i, 3 + k) - 1.0) * 1e-3;
        m[i] = (0.5 + u01(i, 6)) / ((double)n * cfg.G);
    }
    nb_context* ctx = nullptr;
    int rc = nb_create(&ctx, &cfg);
    if (rc) {
        fprintf(stderr, "nb_create: %s (%s)\n", nb_strerror(rc), ctx ? nb_last_error(ctx) : "");
        return 2;
    }
    rc = nb_set_state(ctx, &q[0], &q[n], &q[2 * n], &v[0], &v[n], &v[2 * n], m.data(), nullptr);
    if (!rc && warmup > 0) rc = nb_step(ctx, 1, warmup);
    float ms = 0;
    if (!rc) rc = nb_step_timed(ctx, 1 + warmup, steps, &ms);
    if (rc) {
        fprintf(stderr, "step failed: %s (%s)\n", nb_strerror(rc), nb_last_error(ctx));
        return 2;
    }
    const double pairs = (double)n * (double)(n - 1);
    printf("{\"n\": %ld, \"precision\": \"%s\", \"steps\": %d, \"ms_per_step\": %.4f, \"pairs_per_s\": %.6e, "
           "\"tflops_20flop\": %.2f}\n", n, prec, steps, ms, pairs / (ms * 1e-3), pairs / (ms * 1e-3) * 20 / 1e12);
    nb_destroy(ctx);
    return 0;
}
