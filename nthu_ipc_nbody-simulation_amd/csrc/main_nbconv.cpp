// main_nbconv.cpp —  nbconv <input.txt> <output.nbst>
// Converts the reference's text input (nbody.cc:22-39) into the binary NBODYST2 form bin/hw5 also accepts: same bodies,
// same planet/asteroid, doubles stored exactly (the text form prints 17 significant digits, so both parse to the same
// values).  Text parsing of 2^24 bodies (~3 GB) is impractical; the binary file is 57 bytes per body.
#include <cstdio>

#include "../../include/nbody_amd.h"
#include "nbody_io.h"

int main(int argc, char** argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: nbconv <input.txt> <output.nbst>\n");
        return 2;
    }
    nbio::Input in;
    if (!nbio::read_input(argv[1], in)) {
        fprintf(stderr, "nbconv: cannot read %s\n", argv[1]);
        return 1;
    }
    nb_config cfg;
    nb_config_default(&cfg);  // the reference's param:: values (nbody.cc:10-13)
    nb_state_header h{};
    h.n = in.n;
    h.precision = NB_F64;
    h.step = 0;
    h.planet = in.planet;
    h.asteroid = in.asteroid;
    h.G = cfg.G;
    h.eps = cfg.eps;
    h.dt = cfg.dt;
    int rc = nb_write_state_file(argv[2], &h, in.qx.data(), in.qy.data(), in.qz.data(), in.vx.data(), in.vy.data(),
                                 in.vz.data(), in.m.data(), in.is_device.data());
    if (rc != NB_OK) {
        fprintf(stderr, "nbconv: %s (%s)\n", nb_strerror(rc), nb_last_error(nullptr));
        return 1;
    }
    return 0;
}
