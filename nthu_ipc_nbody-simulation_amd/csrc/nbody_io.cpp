#include "nbody_io.h"

#include <cstdio>
#include <cstring>
#include <fstream>

namespace nbio {

bool read_input(const char* filename, Input& in) {
    std::ifstream fin(filename);
    if (!fin) return false;
    if (!(fin >> in.n >> in.planet >> in.asteroid) || in.n < 0) return false;
    const size_t n = (size_t)in.n;
    for (auto* v : {&in.qx, &in.qy, &in.qz, &in.vx, &in.vy, &in.vz, &in.m}) v->assign(n, 0.0);
    in.is_device.assign(n, 0);
    in.type.assign(n, std::string());
    for (size_t i = 0; i < n; ++i) {
        // operator>> parses doubles correctly rounded, as the reference's reader does (nbody.cc:37)
        if (!(fin >> in.qx[i] >> in.qy[i] >> in.qz[i] >> in.vx[i] >> in.vy[i] >> in.vz[i] >> in.m[i] >> in.type[i]))
            return false;
        in.is_device[i] = in.type[i] == "device";
    }
    return true;
}

bool is_state_file(const char* filename) {
    FILE* f = fopen(filename, "rb");
    if (!f) return false;
    char magic[8] = {0};
    const bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, "NBODYST", 7) == 0;
    fclose(f);
    return ok;
}

bool write_output(const char* filename, double min_dist, int hit_time_step, int gravity_device_id,
                  double missile_cost) {
    FILE* f = fopen(filename, "w");
    if (!f) return false;
    // std::scientific << setprecision(16) prints exactly what %.16e prints
    fprintf(f, "%.16e\n%d\n%d %.16e\n", min_dist, hit_time_step, gravity_device_id, missile_cost);
    return fclose(f) == 0;
}

}  // namespace nbio
