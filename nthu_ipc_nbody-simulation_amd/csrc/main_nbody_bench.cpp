// main_nbody_bench.cpp — a compiled host of the C ABI for large N (no Python, no torch):
//     bin/nbody_bench [N=1048576] [steps=10] [warmup=2] [precision: f32|f32acc64|f64] [gpus=0] [overlap=0] [exchange=rccl] [pairs=shared] [deadline_s=0]
// gpus >= 1 runs the index-sharded stepper (nb_sharded_*: this one process drives GPUs 0..gpus-1; per step the GPUs share
// the unordered pairs of the system — K1s, one reduce-scatter of partial forces — and all-gather the positions in place;
// pairs=ordered: every GPU evaluates every ordered pair of its own targets (K1), all-gather only; overlap=1 = two-phase
// ordered-pair step hiding the gather); gpus = 0 the plain context.
// exchange: rccl | copy (peer copies on the copy engines, NB_SHARDED_COPY_EXCHANGE) | copy-one-gpu (the same with all
// `gpus` ranks on device 0 — every P > 1 line of the host runs on a one-GPU box; the ranks then share the chip) | host (the
// all-gather through a pinned host array, NB_SHARDED_HOST_EXCHANGE: no peer-to-peer involved) | host-one-gpu.
// Generates the synthetic bodies of SURVEY §8(d) (same splitmix64 stream as nbody_amd/synthetic.py), uploads them with
// nb_set_state, advances `steps` steps with nb_step_timed (HIP events on the context's stream) and prints pairs/s.
// Shows what a C/C++ caller of libnbody_amd looks like and gives rocprofv3 a target without an interpreter in front.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/nbody_amd_ext.h"  // nb_sharded_* (includes nbody_amd.h)

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
    const int gpus = argc > 5 ? atoi(argv[5]) : 0;
    const int overlap = argc > 6 ? atoi(argv[6]) : 0;
    const char* exchange = argc > 7 ? argv[7] : "rccl";
    const bool ordered = argc > 8 && !strcmp(argv[8], "ordered");
    const double deadline = argc > 9 ? atof(argv[9]) : 0.0;  // > 0: every wait of the sharded host bounded (nb_sharded_set_deadline)
    const bool one_gpu = !strcmp(exchange, "copy-one-gpu") || !strcmp(exchange, "host-one-gpu");
    const bool host = !strncmp(exchange, "host", 4);
    const bool copy = !strncmp(exchange, "copy", 4);
    if (!copy && !host && strcmp(exchange, "rccl")) {
        fprintf(stderr, "exchange must be rccl, copy, copy-one-gpu, host or host-one-gpu\n");
        return 2;
    }
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
    const double pairs = (double)n * (double)(n - 1);
    if (gpus >= 1) {
        if (cfg.precision == NB_F64) {
            fprintf(stderr, "the sharded stepper runs the fp32 modes\n");
            return 2;
        }
        std::vector<int> devs(gpus);
        for (int g = 0; g < gpus; ++g) devs[g] = one_gpu ? 0 : g;
        nb_sharded* sh = nullptr;
        int rc = nb_sharded_create(&sh, devs.data(), gpus, n, cfg.precision, cfg.G, cfg.eps, cfg.dt,
                                   (overlap ? NB_SHARDED_OVERLAP : 0) | (copy ? NB_SHARDED_COPY_EXCHANGE : 0) | (host ? NB_SHARDED_HOST_EXCHANGE : 0) |
                                       (ordered ? NB_SHARDED_ORDERED_PAIRS : 0));
        if (!rc && deadline > 0) rc = nb_sharded_set_deadline(sh, deadline);
        if (!rc) rc = nb_sharded_set_state(sh, &q[0], &q[n], &q[2 * n], &v[0], &v[n], &v[2 * n], m.data());
        if (!rc && warmup > 0) rc = nb_sharded_step(sh, warmup);
        double ms = 0;
        if (!rc) rc = nb_sharded_step_timed(sh, steps, &ms);
        if (rc) {
            fprintf(stderr, "sharded run failed: %s (%s)\n", nb_strerror(rc), nb_sharded_last_error(sh));
            if (sh) nb_sharded_destroy(sh);
            return 2;
        }
        int tpl = 0, js = 0, wg = 0;
        int64_t per = 0;
        nb_sharded_info(sh, nullptr, &per, &tpl, &js, &wg);
        printf("{\"n\": %ld, \"precision\": \"%s\", \"gpus\": %d, \"exchange\": \"%s\", \"overlap\": %d, \"steps\": %d, \"ms_per_step\": %.4f, "
               "\"pairs_per_s\": %.6e, \"tflops_20flop\": %.2f, \"targets_per_gpu\": %lld, \"plan\": [%d, %d, %d], \"kernel\": \"%s\"}\n",
               n, prec, gpus, exchange, overlap, steps, ms, pairs / (ms * 1e-3), pairs / (ms * 1e-3) * 20 / 1e12, (long long)per, tpl, js,
               wg, nb_sharded_kernel_name(sh));
        nb_sharded_destroy(sh);
        return 0;
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
    printf("{\"n\": %ld, \"precision\": \"%s\", \"steps\": %d, \"ms_per_step\": %.4f, \"pairs_per_s\": %.6e, "
           "\"tflops_20flop\": %.2f}\n", n, prec, steps, ms, pairs / (ms * 1e-3), pairs / (ms * 1e-3) * 20 / 1e12);
    nb_destroy(ctx);
    return 0;
}
