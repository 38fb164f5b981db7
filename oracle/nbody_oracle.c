/*
 * nbody_oracle.c — CPU restatement of the reference's direct N-body path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product library
 * (nthu_ipc_nbody-simulation_amd/csrc -> libnbody_amd.so, bin/hw5) never links, loads or
 * calls anything in this directory.
 *
 * Parity status: PINNED.  Gated in the build container against
 *   (1) the reference's 12 golden outputs testcases/b*.out (all three lines), and
 *   (2) the reference's own run_step compiled from /root/reference/samples/nbody.cc
 *       (oracle/_ref/libnbody_ref.so, see oracle/Makefile + oracle/ref_shim.cc): bit-identical
 *       q,v after steps 1, 2 and 1000 (tests/test_oracle_cpu.py; fixtures in tests/golden/).
 *
 * Each function cites the reference lines it follows (paths relative to /root/reference).
 * Arithmetic is kept operation-for-operation (association order, pow(x,1.5), three divides)
 * so that trajectories, not only the printed results, are bit-identical to the reference when
 * this file is compiled without FMA contraction (-ffp-contract=off, no -march).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nbody_oracle.h"

/* ---- constants: samples/nbody.cc:10-13,17-19 ; hw5.cu:51-54,65-67 ---- */
const orc_params ORC_REFERENCE_PARAMS = {
    /* n_steps        */ 200000,
    /* dt             */ 60.0,
    /* eps            */ 1e-3,
    /* G              */ 6.674e-11,
    /* planet_radius  */ 1e7,
    /* missile_speed  */ 1e6,
};

/* samples/nbody.cc:14-16  param::gravity_device_mass */
double orc_gravity_device_mass(double m0, double t) { return m0 + 0.5 * m0 * fabs(sin(t / 6000)); }

/* samples/nbody.cc:19  param::get_missile_cost */
double orc_missile_cost(double t) { return 1e5 + 1e3 * t; }

/*
 * Effective masses for the step with index `step`:  samples/nbody.cc:61-64.
 * The reference re-evaluates gravity_device_mass inside the j-loop; the value does not depend on i,
 * so hoisting it is bit-identical.
 */
void orc_effective_mass(int step, int n, const double* m, const uint8_t* is_device, double dt, double* m_eff) {
    for (int j = 0; j < n; j++) {
        double mj = m[j];
        if (is_device && is_device[j]) mj = orc_gravity_device_mass(mj, step * dt);
        m_eff[j] = mj;
    }
}

/*
 * Accelerations of target rows [i0, i1):  samples/nbody.cc:56-74 (accel phase of run_step).
 * j ascending, j == i skipped, dist3 = pow(r2 + eps*eps, 1.5), a += G*mj*dx/dist3 evaluated left to right.
 * abs_sum (optional, may be NULL): Σ_j |G mj d / dist3| per component-free magnitude, used by the parity tests
 * to scale fp32 tolerances (the net force on a uniform cloud cancels heavily).
 * Rows are independent, so the OpenMP split over i does not change any bit of the result.
 */
static inline void accel_row(int n, const double* qx, const double* qy, const double* qz, const double* m_eff, double G,
                             double eps, int i, double* ax, double* ay, double* az, double* abs_sum) {
    double sx = 0, sy = 0, sz = 0, sa = 0;
    for (int j = 0; j < n; j++) {
        if (j == i) continue;
        double mj = m_eff[j];
        double dx = qx[j] - qx[i];
        double dy = qy[j] - qy[i];
        double dz = qz[j] - qz[i];
        double dist3 = pow(dx * dx + dy * dy + dz * dz + eps * eps, 1.5);
        sx += G * mj * dx / dist3;
        sy += G * mj * dy / dist3;
        sz += G * mj * dz / dist3;
        if (abs_sum) sa += G * mj * sqrt(dx * dx + dy * dy + dz * dz) / dist3;
    }
    *ax = sx;
    *ay = sy;
    *az = sz;
    if (abs_sum) *abs_sum = sa;
}

void orc_accel_rows(int n, const double* qx, const double* qy, const double* qz, const double* m_eff, double G,
                    double eps, int i0, int i1, double* ax, double* ay, double* az, double* abs_sum) {
#pragma omp parallel for schedule(static) if ((long)(i1 - i0) * n >= 65536) /* tiny systems: fork/join costs more than the rows */
    for (int i = i0; i < i1; i++)
        accel_row(n, qx, qy, qz, m_eff, G, eps, i, ax + (i - i0), ay + (i - i0), az + (i - i0), abs_sum ? abs_sum + (i - i0) : NULL);
}

/*
 * The same for an arbitrary LIST of target rows (the spot checks of large systems take a few rows from every shard): row r of
 * the outputs belongs to target rows[r].  The same per-row loop, so every bit equals orc_accel_rows'; OpenMP over the list.
 */
void orc_accel_rows_at(int n, const double* qx, const double* qy, const double* qz, const double* m_eff, double G, double eps,
                       const int* rows, int k, double* ax, double* ay, double* az, double* abs_sum) {
#pragma omp parallel for schedule(dynamic, 1) if ((long)k * n >= 65536)
    for (int r = 0; r < k; r++)
        accel_row(n, qx, qy, qz, m_eff, G, eps, rows[r], ax + r, ay + r, az + r, abs_sum ? abs_sum + r : NULL);
}

/*
 * One step:  samples/nbody.cc:51-89  run_step.
 * accel (all i, from the old positions) -> v += a*dt (all i) -> q += v*dt (all i, new v).
 * scratch: 4*n doubles (ax, ay, az, m_eff).
 */
void orc_run_step(int step, int n, double* qx, double* qy, double* qz, double* vx, double* vy, double* vz,
                  const double* m, const uint8_t* is_device, const orc_params* p, double* scratch) {
    double* ax = scratch;
    double* ay = scratch + n;
    double* az = scratch + 2 * (size_t)n;
    double* me = scratch + 3 * (size_t)n;
    orc_effective_mass(step, n, m, is_device, p->dt, me);
    orc_accel_rows(n, qx, qy, qz, me, p->G, p->eps, 0, n, ax, ay, az, NULL);
    for (int i = 0; i < n; i++) { /* nbody.cc:77-81 */
        vx[i] += ax[i] * p->dt;
        vy[i] += ay[i] * p->dt;
        vz[i] += az[i] * p->dt;
    }
    for (int i = 0; i < n; i++) { /* nbody.cc:84-88 */
        qx[i] += vx[i] * p->dt;
        qy[i] += vy[i] * p->dt;
        qz[i] += vz[i] * p->dt;
    }
}

/* ---- input / output:  samples/nbody.cc:22-49 ---- */

void orc_system_free(orc_system* s) {
    if (!s) return;
    free(s->qx);
    free(s->is_device);
    memset(s, 0, sizeof *s);
}

static int sys_alloc(orc_system* s, int n) {
    double* blk = (double*)calloc((size_t)7 * n, sizeof(double));
    uint8_t* dev = (uint8_t*)calloc((size_t)n, 1);
    if (!blk || !dev) {
        free(blk);
        free(dev);
        return -1;
    }
    s->n = n;
    s->qx = blk;
    s->qy = blk + n;
    s->qz = blk + 2 * (size_t)n;
    s->vx = blk + 3 * (size_t)n;
    s->vy = blk + 4 * (size_t)n;
    s->vz = blk + 5 * (size_t)n;
    s->m = blk + 6 * (size_t)n;
    s->is_device = dev;
    return 0;
}

int orc_system_copy(orc_system* dst, const orc_system* src) {
    if (sys_alloc(dst, src->n)) return -1;
    dst->planet = src->planet;
    dst->asteroid = src->asteroid;
    const size_t B = (size_t)src->n * sizeof(double); /* the source's seven vectors need not be one block */
    memcpy(dst->qx, src->qx, B);
    memcpy(dst->qy, src->qy, B);
    memcpy(dst->qz, src->qz, B);
    memcpy(dst->vx, src->vx, B);
    memcpy(dst->vy, src->vy, B);
    memcpy(dst->vz, src->vz, B);
    memcpy(dst->m, src->m, B);
    memcpy(dst->is_device, src->is_device, (size_t)src->n);
    return 0;
}

/* samples/nbody.cc:22-39 read_input: "n planet asteroid" then n x "qx qy qz vx vy vz m type" (operator>>). */
int orc_read_input(const char* filename, orc_system* s) {
    FILE* f = fopen(filename, "r");
    if (!f) return -1;
    int n, planet, asteroid;
    if (fscanf(f, "%d %d %d", &n, &planet, &asteroid) != 3 || n < 0) {
        fclose(f);
        return -2;
    }
    if (sys_alloc(s, n)) {
        fclose(f);
        return -3;
    }
    s->planet = planet;
    s->asteroid = asteroid;
    for (int i = 0; i < n; i++) {
        char type[64];
        if (fscanf(f, "%lf %lf %lf %lf %lf %lf %lf %63s", &s->qx[i], &s->qy[i], &s->qz[i], &s->vx[i], &s->vy[i],
                   &s->vz[i], &s->m[i], type) != 8) {
            fclose(f);
            orc_system_free(s);
            return -2;
        }
        s->is_device[i] = (strcmp(type, "device") == 0); /* nbody.cc:62,110 */
    }
    fclose(f);
    return 0;
}

/* samples/nbody.cc:41-49 write_output: scientific, precision digits10+1 = 16. */
int orc_write_output(const char* filename, const orc_result* r) {
    FILE* f = fopen(filename, "w");
    if (!f) return -1;
    fprintf(f, "%.16e\n%d\n%d %.16e\n", r->min_dist, r->hit_time_step, r->gravity_device_id, r->missile_cost);
    fclose(f);
    return 0;
}

static double dist2(const orc_system* s, int i, int j) {
    double dx = s->qx[i] - s->qx[j];
    double dy = s->qy[i] - s->qy[j];
    double dz = s->qz[i] - s->qz[j];
    return dx * dx + dy * dy + dz * dz;
}

/* Problem 1:  samples/nbody.cc:106-122.  devices massless; min over steps 0..n_steps of |q_p - q_a|. */
double orc_problem1(const orc_system* in, const orc_params* p) {
    orc_system s;
    if (orc_system_copy(&s, in)) return NAN;
    double* scratch = (double*)malloc((size_t)4 * s.n * sizeof(double));
    for (int i = 0; i < s.n; i++)
        if (s.is_device[i]) s.m[i] = 0; /* nbody.cc:109-113 */
    double min_dist = INFINITY;
    for (int step = 0; step <= p->n_steps; step++) {
        if (step > 0) orc_run_step(step, s.n, s.qx, s.qy, s.qz, s.vx, s.vy, s.vz, s.m, s.is_device, p, scratch);
        double d = sqrt(dist2(&s, s.planet, s.asteroid)); /* nbody.cc:118-121 */
        if (d < min_dist) min_dist = d;
    }
    free(scratch);
    orc_system_free(&s);
    return min_dist;
}

/*
 * Problems 2 and 3.
 * P2: samples/nbody.cc:124-138 — first step with d2 < planet_radius^2 (strict), else -2.
 * P3: defined only by hw5.cu (the sample prints -999): for each device d, on the P2 trajectory,
 *   hw5.cu:289-309 missile_cost_gpu, evaluated after the step's update, in this order:
 *     (i)  planet–asteroid d2 < R^2            -> device d fails;
 *     (ii) m[d] != 0 and planet–device d2 < ((missile_speed*dt)*step)^2
 *                                             -> cost = 1e5 + 1e3*(step+1)*dt, m[d] = 0 from the next step on.
 *   hw5.cu:512,598-601: answer = feasible device of smallest cost (strict <), original file index; none -> -1 0;
 *   hw5.cu:547-548,568: no P2 hit -> -1 0.
 * Until its missile arrives, device d's trajectory IS the P2 trajectory, so each device resumes from a snapshot
 * taken at its arrival step (hw5.cu:265-287,482-489) — arithmetic identical to restarting from step 0
 * (orc_problem3_from_zero below does exactly that and the tests compare the two).
 */
void orc_problem23(const orc_system* in, const orc_params* p, orc_result* out, orc_p3_detail* detail, int max_detail) {
    orc_system s;
    orc_system_copy(&s, in);
    int n = s.n;
    double* scratch = (double*)malloc((size_t)4 * n * sizeof(double));
    int ndev = 0;
    for (int i = 0; i < n; i++) ndev += s.is_device[i];
    int* dev = (int*)malloc(sizeof(int) * (ndev + 1));
    int* arrival = (int*)malloc(sizeof(int) * (ndev + 1));
    orc_system* snap = (orc_system*)calloc((size_t)ndev + 1, sizeof(orc_system));
    for (int i = 0, k = 0; i < n; i++)
        if (s.is_device[i]) {
            dev[k] = i;
            arrival[k] = -2;
            k++;
        }
    const double R2 = p->planet_radius * p->planet_radius;
    int hit = -2;
    for (int step = 0; step <= p->n_steps; step++) {
        if (step > 0) orc_run_step(step, n, s.qx, s.qy, s.qz, s.vx, s.vy, s.vz, s.m, s.is_device, p, scratch);
        if (dist2(&s, s.planet, s.asteroid) < R2) { /* nbody.cc:134-137 */
            hit = step;
            break;
        }
        for (int k = 0; k < ndev; k++) { /* hw5.cu:265-287 */
            if (arrival[k] != -2 || s.m[dev[k]] == 0) continue; /* hw5.cu:299 m[d] != 0 */
            double md = (p->missile_speed * p->dt) * step;
            if (dist2(&s, s.planet, dev[k]) < md * md) {
                arrival[k] = step;
                orc_system_copy(&snap[k], &s);
            }
        }
    }
    out->hit_time_step = hit;
    out->gravity_device_id = -1;
    out->missile_cost = 0;
    if (hit != -2) {
        double best = INFINITY;
        for (int k = 0; k < ndev; k++) {
            int feasible = 0, fail_step = -2;
            double cost = INFINITY;
            if (arrival[k] != -2) {
                orc_system* t = &snap[k];
                cost = orc_missile_cost((arrival[k] + 1) * p->dt); /* hw5.cu:305 */
                t->m[dev[k]] = 0;                                   /* hw5.cu:306 */
                feasible = 1;
                for (int step = arrival[k] + 1; step <= p->n_steps; step++) {
                    orc_run_step(step, n, t->qx, t->qy, t->qz, t->vx, t->vy, t->vz, t->m, t->is_device, p, scratch);
                    if (dist2(t, t->planet, t->asteroid) < R2) { /* hw5.cu:295-298 */
                        feasible = 0;
                        fail_step = step;
                        break;
                    }
                }
            } else {
                fail_step = hit;
            }
            if (detail && k < max_detail) {
                detail[k].device = dev[k];
                detail[k].arrival_step = arrival[k];
                detail[k].feasible = feasible;
                detail[k].fail_step = fail_step;
                detail[k].cost = cost;
            }
            if (feasible && cost < best) { /* hw5.cu:512 */
                best = cost;
                out->gravity_device_id = dev[k];
                out->missile_cost = cost;
            }
        }
    }
    if (detail && hit == -2) /* no hit: nothing to prevent, but report what was observed */
        for (int k = 0; k < ndev && k < max_detail; k++) {
            detail[k].device = dev[k];
            detail[k].arrival_step = arrival[k];
            detail[k].feasible = 0;
            detail[k].fail_step = -2;
            detail[k].cost = arrival[k] != -2 ? orc_missile_cost((arrival[k] + 1) * p->dt) : INFINITY;
        }
    if (detail)
        for (int k = ndev; k < max_detail; k++) detail[k].device = -1;
    for (int k = 0; k < ndev; k++) orc_system_free(&snap[k]);
    free(snap);
    free(dev);
    free(arrival);
    free(scratch);
    orc_system_free(&s);
}

/*
 * P3 for ONE device restarted from step 0 (the literal SURVEY Appendix A-3 definition; hw5.cu:289-309 applied to
 * every step).  Returns 1 if feasible; *cost / *arrival_step receive the missile cost and arrival step.
 */
int orc_problem3_from_zero(const orc_system* in, const orc_params* p, int device, double* cost, int* arrival_step) {
    orc_system s;
    orc_system_copy(&s, in);
    double* scratch = (double*)malloc((size_t)4 * s.n * sizeof(double));
    const double R2 = p->planet_radius * p->planet_radius;
    int feasible = 1;
    *cost = INFINITY;
    *arrival_step = -2;
    for (int step = 0; step <= p->n_steps; step++) {
        if (step > 0) orc_run_step(step, s.n, s.qx, s.qy, s.qz, s.vx, s.vy, s.vz, s.m, s.is_device, p, scratch);
        if (dist2(&s, s.planet, s.asteroid) < R2) {
            feasible = 0;
            break;
        }
        if (s.m[device] != 0) {
            double md = (p->missile_speed * p->dt) * step;
            if (dist2(&s, s.planet, device) < md * md) {
                *cost = orc_missile_cost((step + 1) * p->dt);
                *arrival_step = step;
                s.m[device] = 0;
            }
        }
    }
    if (*arrival_step == -2) feasible = 0;
    free(scratch);
    orc_system_free(&s);
    return feasible;
}

/* Whole program:  samples/nbody.cc:91-146 main (P1, P2) + hw5.cu main (P3). */
int orc_solve_file(const char* in_path, const char* out_path) {
    orc_system s;
    memset(&s, 0, sizeof s);
    int rc = orc_read_input(in_path, &s);
    if (rc) return rc;
    orc_result r;
    r.min_dist = orc_problem1(&s, &ORC_REFERENCE_PARAMS);
    orc_problem23(&s, &ORC_REFERENCE_PARAMS, &r, NULL, 0);
    rc = orc_write_output(out_path, &r);
    orc_system_free(&s);
    return rc;
}

#ifdef ORC_MAIN
/* CLI with the reference's contract: prog <in> <out> (samples/nbody.cc:91-94). */
int main(int argc, char** argv) {
    if (argc != 3) {
        fprintf(stderr, "must supply 2 arguments\n");
        abort();
    }
    return orc_solve_file(argv[1], argv[2]) ? 1 : 0;
}
#endif
