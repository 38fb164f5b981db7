// io_check.cpp — host-only exerciser of nbody_io (no GPU): parses an input file the way bin/hw5 does and prints a digest;
// optionally writes an output file from given answers.  Built with -fsanitize=address,undefined by `make asan` and run over
// all testcase inputs by tests/test_oracle_cpu.py (the sanitizer pass the reference did with cuda-memcheck, hw5.cu:631-642).
#include <cstdio>
#include <cstdlib>

#include "nbody_io.h"

int main(int argc, char** argv) {
    if (argc < 2) return 64;
    nbio::Input in;
    if (!nbio::read_input(argv[1], in)) {
        fprintf(stderr, "io_check: cannot read %s\n", argv[1]);
        return 1;
    }
    int ndev = 0;
    double sum = 0;
    for (int i = 0; i < in.n; ++i) {
        ndev += in.is_device[i];
        sum += in.qx[i] + in.qy[i] + in.qz[i] + in.vx[i] + in.vy[i] + in.vz[i] + in.m[i];
    }
    printf("%d %d %d %d %.17g\n", in.n, in.planet, in.asteroid, ndev, sum);
    if (argc == 7) return nbio::write_output(argv[2], atof(argv[3]), atoi(argv[4]), atoi(argv[5]), atof(argv[6])) ? 0 : 1;
    return 0;
}
