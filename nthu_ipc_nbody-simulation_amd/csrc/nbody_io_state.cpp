// nbody_io_state.cpp — the binary input variant of the CLI: an NBODYST2 state file carrying what the
// text format carries (nbody.cc:22-39: n, planet, asteroid, then q, v, m, type per body — `type` reduced to the
// `device` predicate, the only type with semantics, nbody.cc:62,110).  Reads through the C ABI (nb_read_state_file).
#include "../../include/nbody_amd.h"
#include "nbody_io.h"

namespace nbio {

bool read_state_input(const char* filename, Input& in) {
    nb_state_header h;
    if (nb_read_state_file(filename, &h, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) != NB_OK)
        return false;
    if (h.n > 0x7fffffff || h.planet < 0 || h.asteroid < 0 || h.planet >= h.n || h.asteroid >= h.n) return false;
    const size_t n = (size_t)h.n;
    in.n = (int)h.n;
    in.planet = h.planet;
    in.asteroid = h.asteroid;
    for (auto* v : {&in.qx, &in.qy, &in.qz, &in.vx, &in.vy, &in.vz, &in.m}) v->assign(n, 0.0);
    in.is_device.assign(n, 0);
    if (nb_read_state_file(filename, &h, h.n, in.qx.data(), in.qy.data(), in.qz.data(), in.vx.data(), in.vy.data(),
                           in.vz.data(), in.m.data(), in.is_device.data()) != NB_OK)
        return false;
    in.type.assign(n, std::string("body"));
    for (size_t i = 0; i < n; ++i)
        if (in.is_device[i]) in.type[i] = "device";
    return true;
}

}  // namespace nbio
