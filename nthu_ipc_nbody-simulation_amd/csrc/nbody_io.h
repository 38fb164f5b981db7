// nbody_io.h — the reference's two file formats (samples/nbody.cc:22-49 ; hw5.cu:86-141), host side.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace nbio {

struct Input {
    int n = 0, planet = 0, asteroid = 0;
    std::vector<double> qx, qy, qz, vx, vy, vz, m;
    std::vector<uint8_t> is_device;  // type[i] == "device" (nbody.cc:62,110) — the only type with semantics
    std::vector<std::string> type;
};

// "n planet asteroid" then n x "qx qy qz vx vy vz m type", whitespace separated (nbody.cc:27,37).
// Like the reference there is no validation beyond what the stream gives; returns false if the file cannot
// be opened or is truncated.
bool read_input(const char* filename, Input& in);

// binary form of the same input: an NBODYST1/2 state file (include/nbody_amd.h).  is_state_file looks at the magic only;
// read_state_input needs planet/asteroid recorded in the header (version 2) and links against libnbody_amd.
bool is_state_file(const char* filename);
bool read_state_input(const char* filename, Input& in);
const char* state_input_error();  // why the last read_state_input returned false

// three lines, scientific with 16 digits after the point (digits10 + 1): nbody.cc:43-48 == hw5.cu:135-140
bool write_output(const char* filename, double min_dist, int hit_time_step, int gravity_device_id,
                  double missile_cost);

}  // namespace nbio
