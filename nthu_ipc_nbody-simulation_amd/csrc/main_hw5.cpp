// main_hw5.cpp — the drop-in CLI:  hw5 <input> <output>   (samples/nbody.cc:91-94,145 ; hw5.cu:532-535,606,618)
//
// Same argv, same input and output formats, same three answers; all arithmetic runs on the MI355X through the
// C ABI of libnbody_amd (nb_solve).  NB_DEVICES=0,1,... (optional) spreads the independent scenarios P1, P2 and
// the per-device P3 runs over several GPUs, as the reference does over its two (hw5.cu:564-567,587-588).
// Tuning / test hooks of the driver come from the environment HERE and travel to the library as nb_solve_options:
//   NB_SOLVE_ENGINE=steps|persistent  NB_SOLVE_STREAMS=merged|split  NB_SOLVE_MAX_BATCH=2..8  NB_SOLVE_P3_PARALLEL=k
//   NB_GRAPH_CHUNK=<even 2..4000> (shorter replayed graphs: more replay boundaries in a short run, for tests)  NB_SOLVE_HANDOFF=host
// (the library itself reads only NB_SOLVE_TRACE=1: a timeline on stderr).
// <input> may also be the binary form of the same data (an NBODYST2 state file with planet/asteroid recorded, see
// include/nbody_amd.h; bin/nbconv converts) — recognised by its magic, the text format stays the default.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/nbody_amd_ext.h"  // nb_solve_ex and its options (includes nbody_amd.h)
#include "nbody_io.h"

int main(int argc, char** argv) {
    if (argc != 3) {
        throw std::runtime_error("must supply 2 arguments");  // uncaught -> abort, like the reference
    }
    nbio::Input in;
    if (nbio::is_state_file(argv[1])) {
        if (!nbio::read_state_input(argv[1], in)) {
            fprintf(stderr, "hw5: cannot read state file %s: %s\n", argv[1], nbio::state_input_error());
            return 1;
        }
    } else if (!nbio::read_input(argv[1], in)) {
        fprintf(stderr, "hw5: cannot read %s\n", argv[1]);
        return 1;
    }
    std::vector<int> gpus;
    if (const char* env = getenv("NB_DEVICES")) {
        std::string s(env);
        size_t pos = 0;
        while (pos < s.size()) {
            size_t e = s.find(',', pos);
            if (e == std::string::npos) e = s.size();
            if (e > pos) gpus.push_back(atoi(s.substr(pos, e - pos).c_str()));
            pos = e + 1;
        }
    }
    nb_solve_options opt{};
    if (const char* e = getenv("NB_SOLVE_ENGINE")) opt.engine = !strcmp(e, "steps") ? 1 : (!strcmp(e, "persistent") && in.n <= 128) ? 2 : 0;
    if (const char* e = getenv("NB_SOLVE_STREAMS")) opt.streams = !strcmp(e, "merged") ? 1 : !strcmp(e, "split") ? 2 : 0;
    if (const char* e = getenv("NB_SOLVE_MAX_BATCH")) opt.max_batch = atoi(e);
    if (const char* e = getenv("NB_SOLVE_P3_PARALLEL")) opt.p3_parallel = atoi(e);
    if (const char* e = getenv("NB_GRAPH_CHUNK")) opt.graph_chunk = atoi(e);
    if (const char* e = getenv("NB_SOLVE_HANDOFF")) opt.handoff = !strcmp(e, "host") ? NB_HANDOFF_HOST_STAGED : NB_HANDOFF_AUTO;
    nb_answer ans{};
    int rc = nb_solve_ex(in.n, in.planet, in.asteroid, in.qx.data(), in.qy.data(), in.qz.data(), in.vx.data(),
                         in.vy.data(), in.vz.data(), in.m.data(), in.is_device.data(),
                         gpus.empty() ? nullptr : gpus.data(), (int)gpus.size(), &opt, &ans);
    if (rc != NB_OK) {
        fprintf(stderr, "hw5: nb_solve failed: %s (%s)\n", nb_strerror(rc), nb_last_error(nullptr));
        return 2;
    }
    if (!nbio::write_output(argv[2], ans.min_dist, ans.hit_time_step, ans.gravity_device_id, ans.missile_cost)) {
        fprintf(stderr, "hw5: cannot write %s\n", argv[2]);
        return 1;
    }
    // The answer is on disk (write_output closes the file) and nb_solve has released everything it created: leave without
    // running the HIP runtime's exit handlers — they take 60-90 ms of a 0.5-1.6 s program (profiles/r02_startup_probe.txt)
    // and free only what the kernel driver reclaims at process exit anyway.
    // (NB_HW5_CLEAN_EXIT=1 returns normally instead: profilers write their files from exit handlers.)
    if (getenv("NB_HW5_CLEAN_EXIT")) return 0;
    fflush(nullptr);
    std::_Exit(0);
}
