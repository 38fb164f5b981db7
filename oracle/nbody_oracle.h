/*
 * nbody_oracle.h — interface of the CPU oracle (TEST INFRASTRUCTURE ONLY; see nbody_oracle.c).
 * Loaded through ctypes by oracle/oracle.py, which only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg import.
 */
#ifndef NBODY_ORACLE_H
#define NBODY_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int n_steps;
    double dt, eps, G, planet_radius, missile_speed;
} orc_params;

typedef struct {
    int n, planet, asteroid;
    double *qx, *qy, *qz, *vx, *vy, *vz, *m; /* one 7n block, qx is the base */
    uint8_t* is_device;
} orc_system;

typedef struct {
    double min_dist;
    int hit_time_step;
    int gravity_device_id;
    double missile_cost;
} orc_result;

typedef struct {
    int device;       /* original file index, -1 = unused slot */
    int arrival_step; /* -2 = missile never arrives before the hit */
    int feasible;
    int fail_step; /* step of the planet hit when infeasible */
    double cost;
} orc_p3_detail;

extern const orc_params ORC_REFERENCE_PARAMS;

double orc_gravity_device_mass(double m0, double t);
double orc_missile_cost(double t);
void orc_effective_mass(int step, int n, const double* m, const uint8_t* is_device, double dt, double* m_eff);
void orc_accel_rows(int n, const double* qx, const double* qy, const double* qz, const double* m_eff, double G,
                    double eps, int i0, int i1, double* ax, double* ay, double* az, double* abs_sum);
/* the same for a list of target rows: outputs[r] belongs to rows[r]; bit-identical per row, OpenMP over the list */
void orc_accel_rows_at(int n, const double* qx, const double* qy, const double* qz, const double* m_eff, double G, double eps,
                       const int* rows, int k, double* ax, double* ay, double* az, double* abs_sum);
void orc_run_step(int step, int n, double* qx, double* qy, double* qz, double* vx, double* vy, double* vz,
                  const double* m, const uint8_t* is_device, const orc_params* p, double* scratch);
int orc_read_input(const char* filename, orc_system* s);
int orc_write_output(const char* filename, const orc_result* r);
int orc_system_copy(orc_system* dst, const orc_system* src);
void orc_system_free(orc_system* s);
double orc_problem1(const orc_system* in, const orc_params* p);
void orc_problem23(const orc_system* in, const orc_params* p, orc_result* out, orc_p3_detail* detail, int max_detail);
int orc_problem3_from_zero(const orc_system* in, const orc_params* p, int device, double* cost, int* arrival_step);
int orc_solve_file(const char* in_path, const char* out_path);

#ifdef __cplusplus
}
#endif
#endif
