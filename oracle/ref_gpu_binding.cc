// ref_gpu_binding.cc — the binding of INTEGRATION.md §2, executed (test infrastructure only).
//
// oracle/Makefile compiles /root/reference/samples/nbody.cc *where it lies* as position-independent code, weakens its own
// definition of `run_step` (samples/nbody.cc:51-89) in the object file, and links the object with this file and
// libnbody_amd.so into oracle/_ref/nbody_gpu.  The result is the REFERENCE'S OWN main() — its argument check, read_input,
// the Problem 1 and Problem 2 loops, write_output (nbody.cc:91-146) — whose every `run_step(step, n, qx, ...)` call
// (nbody.cc:116,129) lands in the definition below: the same signature, the arithmetic on the MI355X through the C ABI.
// tests/test_gpu_f64_parity.py::test_reference_main_drives_the_gpu_step runs it on testcases/b20.in and compares lines 1-2
// with the golden output (line 3 is the sample's `// TODO`, nbody.cc:140-143).  No reference source text is reproduced here.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../include/nbody_amd.h"

static nb_context* g_ctx = nullptr;
static int g_n = 0;

static void release() {
    if (g_ctx) nb_destroy(g_ctx);
    g_ctx = nullptr;
}

void run_step(int step, int n, std::vector<double>& qx, std::vector<double>& qy, std::vector<double>& qz,
              std::vector<double>& vx, std::vector<double>& vy, std::vector<double>& vz, const std::vector<double>& m,
              const std::vector<std::string>& type) {
    std::vector<uint8_t> dev((size_t)n);
    for (int i = 0; i < n; i++) dev[(size_t)i] = type[(size_t)i] == "device";  // nbody.cc:62
    if (!g_ctx || g_n != n) {
        release();
        nb_config c;
        nb_config_default(&c);  // param:: of nbody.cc:9-20
        c.n = n;
        if (nb_create(&g_ctx, &c) != NB_OK) {
            fprintf(stderr, "nb_create: %s\n", g_ctx ? nb_last_error(g_ctx) : "no context");
            abort();
        }
        g_n = n;
        atexit(release);
    }
    // nb_run_step = nb_set_state + nb_step(step, 1) + nb_get_state in one call (one synchronisation; no upload while the vectors
    // still hold what the previous call returned, which is what main()'s loops do, nbody.cc:114-122,127-138)
    if (nb_run_step(g_ctx, step, qx.data(), qy.data(), qz.data(), vx.data(), vy.data(), vz.data(), m.data(), dev.data())) {
        fprintf(stderr, "GPU run_step failed: %s\n", nb_last_error(g_ctx));
        abort();
    }
}
