// nbody_io_state.cpp — the binary input variant of the CLI: an NBODYST2 state file carrying what the
// text format carries (nbody.cc:22-39: n, planet, asteroid, then q, v, m, type per body — `type` reduced to the
// `device` predicate, the only type with semantics, nbody.cc:62,110).  Reads through the C ABI (nb_read_state_file).
#include "../../include/nbody_amd.h"
#include "nbody_io.h"

#include <cstdio>

namespace nbio {

static char why[256] = {0};
const char* state_input_error() { return why; }

bool read_state_input(const char* filename, Input& in) {
    nb_state_header h;
    why[0] = 0;
    if (nb_read_state_file(filename, &h, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) != NB_OK) {
        snprintf(why, sizeof why, "%s", nb_last_error(nullptr));
        return false;
    }
    if (h.n > 0x7fffffff || h.planet < 0 || h.asteroid < 0 || h.planet >= h.n || h.asteroid >= h.n) {
        snprintf(why, sizeof why, "no planet/asteroid recorded (a plain checkpoint is not a program input)");
        return false;
    }
    // the program solves the reference's problem: the step-0 fp64 input under param::'s constants (nbody.cc:10-13).  A
    // mid-run or fp32 checkpoint, or one written under other constants, would be solved as something it is not.
    nb_config ref;
    nb_config_default(&ref);
    if (h.step != 0 || h.precision != NB_F64 || h.G != ref.G || h.eps != ref.eps || h.dt != ref.dt) {
        snprintf(why, sizeof why, "not a step-0 fp64 input under the reference's constants: %s%s%s%s%s",
                 h.step != 0 ? "step != 0 " : "", h.precision != NB_F64 ? "precision != fp64 " : "",
                 h.G != ref.G ? "G differs " : "", h.eps != ref.eps ? "eps differs " : "", h.dt != ref.dt ? "dt differs" : "");
        return false;
    }
    const size_t n = (size_t)h.n;
    in.n = (int)h.n;
    in.planet = h.planet;
    in.asteroid = h.asteroid;
    for (auto* v : {&in.qx, &in.qy, &in.qz, &in.vx, &in.vy, &in.vz, &in.m}) v->assign(n, 0.0);
    in.is_device.assign(n, 0);
    if (nb_read_state_file(filename, &h, h.n, in.qx.data(), in.qy.data(), in.qz.data(), in.vx.data(), in.vy.data(),
                           in.vz.data(), in.m.data(), in.is_device.data()) != NB_OK) {
        snprintf(why, sizeof why, "%s", nb_last_error(nullptr));
        return false;
    }
    in.type.assign(n, std::string("body"));
    for (size_t i = 0; i < n; ++i)
        if (in.is_device[i]) in.type[i] = "device";
    return true;
}

}  // namespace nbio
