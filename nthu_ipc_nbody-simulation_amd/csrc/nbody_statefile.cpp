// nbody_statefile.cpp — NBODYST1/2 binary state files behind the C ABI (nb_save_state, nb_load_state, nb_state_file_info,
// nb_read_state_file, nb_write_state_file): checkpoint / large-N input.  The reference has only the text format
// (samples/nbody.cc:22-49) and an in-memory snapshot (hw5.cu:265-287); layout in include/nbody_amd.h.
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>
#include <memory>
#include <new>
#include <string>

#include "nbody_internal.h"

using namespace nbi;

namespace {
// version 1 (round 1): magic "NBODYST1", int64 n, int32 precision, int32 step, double G, eps, dt            (48 bytes)
// version 2:           magic "NBODYST2", uint32 byte-order mark, int32 precision, int64 n, int32 step,
//                      int32 planet, int32 asteroid, int32 reserved, double G, eps, dt                      (64 bytes)
struct StateHeaderV1 {
    char magic[8];
    int64_t n;
    int32_t precision;
    int32_t step;
    double G, eps, dt;
};
struct StateHeaderV2 {
    char magic[8];
    uint32_t bom;
    int32_t precision;
    int64_t n;
    int32_t step;
    int32_t planet, asteroid, reserved;
    double G, eps, dt;
};
static_assert(sizeof(StateHeaderV1) == 48 && sizeof(StateHeaderV2) == 64, "on-disk layout");
const char kMagic1[8] = {'N', 'B', 'O', 'D', 'Y', 'S', 'T', '1'};
const char kMagic2[8] = {'N', 'B', 'O', 'D', 'Y', 'S', 'T', '2'};
constexpr uint32_t kBom = 0x01020304u;

struct FileCloser {
    void operator()(FILE* f) const {
        if (f) fclose(f);
    }
};
using FilePtr = std::unique_ptr<FILE, FileCloser>;

// header of either version; the stream is left at the first body array
int read_header(FILE* f, nb_state_header* h) {
    char magic[8];
    if (fread(magic, 8, 1, f) != 1) return NB_ERR_IO;
    memset(h, 0, sizeof *h);
    if (memcmp(magic, kMagic1, 8) == 0) {
        StateHeaderV1 v;
        if (fread(&v.n, sizeof v - 8, 1, f) != 1) return NB_ERR_IO;
        h->n = v.n; h->precision = v.precision; h->step = v.step;
        h->planet = h->asteroid = -1;
        h->G = v.G; h->eps = v.eps; h->dt = v.dt;
    } else if (memcmp(magic, kMagic2, 8) == 0) {
        StateHeaderV2 v;
        if (fread(&v.bom, sizeof v - 8, 1, f) != 1) return NB_ERR_IO;
        if (v.bom != kBom) return set_error(NB_ERR_IO, "state file written with another byte order");
        h->n = v.n; h->precision = v.precision; h->step = v.step;
        h->planet = v.planet; h->asteroid = v.asteroid;
        h->G = v.G; h->eps = v.eps; h->dt = v.dt;
    } else {
        return set_error(NB_ERR_IO, "not an NBODYST1/NBODYST2 state file");
    }
    if (h->n <= 0 || h->precision < NB_F64 || h->precision > NB_F32_ACC64) return set_error(NB_ERR_IO, "corrupt state header");
    // the body arrays must be there before anyone sizes a buffer from n: 7 doubles + the device byte per body
    struct stat st;
    const long at = ftell(f);
    if (at < 0 || fstat(fileno(f), &st) != 0) return set_error(NB_ERR_IO, "cannot stat state file");
    if (S_ISREG(st.st_mode) && (h->n > (int64_t)((st.st_size - at) / 57)))
        return set_error(NB_ERR_IO, "truncated state file (header announces more bodies than the file holds)");
    return NB_OK;
}

int write_state(const char* path, const nb_state_header* h, const double* const q[6], const double* m,
                const uint8_t* is_device) {
    if (h->n <= 0) return NB_ERR_INVALID;
    const size_t n = (size_t)h->n;
    StateHeaderV2 v{};
    memcpy(v.magic, kMagic2, 8);
    v.bom = kBom;
    v.precision = h->precision;
    v.n = h->n;
    v.step = h->step;
    v.planet = h->planet;
    v.asteroid = h->asteroid;
    v.G = h->G; v.eps = h->eps; v.dt = h->dt;
    // a checkpoint replaces the previous one: written beside it, flushed to the disk, then renamed over it, so that a
    // run killed in the middle of a write (956 MB at N = 2^24) still finds the earlier file whole
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return set_error(NB_ERR_IO, "cannot open state file for writing");
    bool ok = fwrite(&v, sizeof v, 1, f) == 1;
    for (int k = 0; k < 6 && ok; ++k) ok = fwrite(q[k], sizeof(double), n, f) == n;
    ok = ok && fwrite(m, sizeof(double), n, f) == n;
    if (ok && is_device) ok = fwrite(is_device, 1, n, f) == n;
    else if (ok) {
        std::vector<uint8_t> z(n, 0);
        ok = fwrite(z.data(), 1, n, f) == n;
    }
    ok = ok && fflush(f) == 0 && fsync(fileno(f)) == 0;
    ok = (fclose(f) == 0) && ok;
    if (!ok) {
        (void)remove(tmp.c_str());
        return set_error(NB_ERR_IO, "short write to state file");
    }
    if (rename(tmp.c_str(), path) != 0) {
        (void)remove(tmp.c_str());
        return set_error(NB_ERR_IO, "cannot move the finished state file into place");
    }
    return NB_OK;
}

int read_state_impl(const char* path, nb_state_header* hdr, int64_t capacity, double* qx, double* qy, double* qz,
                    double* vx, double* vy, double* vz, double* m, uint8_t* is_device) {
    if (!path || !hdr) return NB_ERR_INVALID;
    FilePtr f(fopen(path, "rb"));
    if (!f) return set_error(NB_ERR_IO, "cannot open state file");
    if (int rc = read_header(f.get(), hdr)) return rc;
    if (!qx && !qy && !qz && !vx && !vy && !vz && !m && !is_device) return NB_OK;  // header only
    if (!qx || !qy || !qz || !vx || !vy || !vz || !m) return NB_ERR_INVALID;
    if (capacity < hdr->n) return set_error(NB_ERR_INVALID, "arrays too small for the bodies in the state file");
    const size_t n = (size_t)hdr->n;
    double* arr[7] = {qx, qy, qz, vx, vy, vz, m};
    for (double* a : arr)
        if (fread(a, sizeof(double), n, f.get()) != n) return set_error(NB_ERR_IO, "truncated state file");
    if (is_device) {
        if (fread(is_device, 1, n, f.get()) != n) return set_error(NB_ERR_IO, "truncated state file");
    }
    return NB_OK;
}

}  // namespace

extern "C" {

int nb_state_file_info(const char* path, int64_t* n, int* precision, int* step) {
    nb_state_header h;
    if (int rc = read_state_impl(path, &h, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr)) return rc;
    if (n) *n = h.n;
    if (precision) *precision = h.precision;
    if (step) *step = h.step;
    return NB_OK;
}

int nb_read_state_file(const char* path, nb_state_header* hdr, int64_t capacity, double* qx, double* qy, double* qz,
                       double* vx, double* vy, double* vz, double* m, uint8_t* is_device) {
    return read_state_impl(path, hdr, capacity, qx, qy, qz, vx, vy, vz, m, is_device);
}

int nb_write_state_file(const char* path, const nb_state_header* hdr, const double* qx, const double* qy,
                        const double* qz, const double* vx, const double* vy, const double* vz, const double* m,
                        const uint8_t* is_device) {
    if (!path || !hdr || !qx || !qy || !qz || !vx || !vy || !vz || !m) return NB_ERR_INVALID;
    if (hdr->precision < NB_F64 || hdr->precision > NB_F32_ACC64) return NB_ERR_INVALID;
    const double* q[6] = {qx, qy, qz, vx, vy, vz};
    try {
        return write_state(path, hdr, q, m, is_device);
    } catch (...) {
        return NB_ERR_NOMEM;
    }
}

int nb_save_state(nb_context* c, const char* path, int step) {
    if (!c || !path) return NB_ERR_INVALID;
    if (!c->have_state) return NB_ERR_STATE;
    try {
        const size_t n = (size_t)c->n;
        std::vector<double> buf(6 * n);
        if (int rc = nb_get_state(c, &buf[0], &buf[n], &buf[2 * n], &buf[3 * n], &buf[4 * n], &buf[5 * n])) return rc;
        nb_state_header h{};
        h.n = c->n;
        h.precision = c->cfg.precision;
        h.step = step;
        h.planet = h.asteroid = -1;  // a context does not know the scenario's bodies (nb_write_state_file records them)
        h.G = c->cfg.G;
        h.eps = c->cfg.eps;
        h.dt = c->cfg.dt;
        const double* q[6] = {&buf[0], &buf[n], &buf[2 * n], &buf[3 * n], &buf[4 * n], &buf[5 * n]};
        int rc = write_state(path, &h, q, c->m_host.data(), c->dev_host.data());
        if (rc) snprintf(c->err, sizeof c->err, "%s", nb_last_error(nullptr));
        return rc;
    } catch (...) {
        return NB_ERR_NOMEM;
    }
}

int nb_load_state(nb_context* c, const char* path, int* step) {
    if (!c || !path) return NB_ERR_INVALID;
    try {
        nb_state_header h;
        if (int rc = read_state_impl(path, &h, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr)) {
            snprintf(c->err, sizeof c->err, "%s", nb_last_error(nullptr));
            return rc;
        }
        // a checkpoint resumes the run it was taken from: same system size, arithmetic and integration parameters.
        // (nb_read_state_file + nb_set_state is the explicit route for loading a state under other parameters.)
        if (h.n != c->n || h.precision != c->cfg.precision || h.G != c->cfg.G || h.eps != c->cfg.eps || h.dt != c->cfg.dt) {
            snprintf(c->err, sizeof c->err,
                     "state file (n=%lld precision=%d G=%g eps=%g dt=%g) does not match the context (n=%d precision=%d "
                     "G=%g eps=%g dt=%g)", (long long)h.n, h.precision, h.G, h.eps, h.dt, c->n, c->cfg.precision, c->cfg.G,
                     c->cfg.eps, c->cfg.dt);
            return NB_ERR_INVALID;
        }
        const size_t n = (size_t)c->n;
        std::vector<double> buf(7 * n);
        std::vector<uint8_t> dev(n);
        if (int rc = read_state_impl(path, &h, (int64_t)n, &buf[0], &buf[n], &buf[2 * n], &buf[3 * n], &buf[4 * n], &buf[5 * n],
                                     &buf[6 * n], dev.data())) {
            snprintf(c->err, sizeof c->err, "%s", nb_last_error(nullptr));
            return rc;
        }
        if (step) *step = h.step;
        return nb_set_state(c, &buf[0], &buf[n], &buf[2 * n], &buf[3 * n], &buf[4 * n], &buf[5 * n], &buf[6 * n], dev.data());
    } catch (...) {
        return NB_ERR_NOMEM;
    }
}

}  // extern "C"
