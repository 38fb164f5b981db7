// ref_shim.cc — C-callable door into the REFERENCE's own run_step (test infrastructure only).
//
// oracle/Makefile compiles /root/reference/samples/nbody.cc *where it lies* together with this
// file into oracle/_ref/libnbody_ref.so.  The reference translation unit defines, with external
// linkage, `run_step` (samples/nbody.cc:51-54) and `param::gravity_device_mass` (nbody.cc:14-16);
// this shim only declares their prototypes and marshals plain arrays into the std::vector /
// std::string arguments they take.  No reference source text is reproduced here.
//
// Used to (a) pin oracle/nbody_oracle.c bit-for-bit against the real implementation
// (tests/test_oracle_cpu.py, tests/golden/make_kats.py) and (b) as bench.py's
// cpu_baseline of kind "reference".
#include <cstdint>
#include <string>
#include <vector>

// prototypes of the reference's symbols (samples/nbody.cc:51-54 and :14)
void run_step(int step, int n, std::vector<double>& qx, std::vector<double>& qy, std::vector<double>& qz,
              std::vector<double>& vx, std::vector<double>& vy, std::vector<double>& vz, const std::vector<double>& m,
              const std::vector<std::string>& type);
namespace param {
double gravity_device_mass(double m0, double t);
}

extern "C" {

// Advance `count` steps with step indices first_step .. first_step+count-1, in place.
int ref_run_steps(int first_step, int count, int n, double* qx, double* qy, double* qz, double* vx, double* vy,
                  double* vz, const double* m, const uint8_t* is_device) {
    std::vector<double> vqx(qx, qx + n), vqy(qy, qy + n), vqz(qz, qz + n);
    std::vector<double> vvx(vx, vx + n), vvy(vy, vy + n), vvz(vz, vz + n), vm(m, m + n);
    std::vector<std::string> type(n);
    for (int i = 0; i < n; i++) type[i] = (is_device && is_device[i]) ? "device" : "body";
    for (int s = 0; s < count; s++) run_step(first_step + s, n, vqx, vqy, vqz, vvx, vvy, vvz, vm, type);
    for (int i = 0; i < n; i++) {
        qx[i] = vqx[i]; qy[i] = vqy[i]; qz[i] = vqz[i];
        vx[i] = vvx[i]; vy[i] = vvy[i]; vz[i] = vvz[i];
    }
    return 0;
}

double ref_gravity_device_mass(double m0, double t) { return param::gravity_device_mass(m0, t); }

}  // extern "C"
