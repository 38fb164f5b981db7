/*
 * nbody_amd.h — C-ABI of libnbody_amd.so, the MI355X (gfx950) drop-in for the reference's hot path:
 * the all-pairs gravitational step `run_step` (samples/nbody.cc:51-89 ; hw5.cu:159-215 + 231-239).
 *
 * The reference has no FFI: its boundary is the CLI `prog <in> <out>` (samples/nbody.cc:91-94,145) whose
 * main() calls run_step(step, n, qx,qy,qz, vx,vy,vz, m, type) once per step (nbody.cc:116,129).  This header
 * is that call hoisted behind a C ABI (plain pointers and sizes, no C++/torch types), plus the scenario
 * drivers that main() wraps around it (nbody.cc:106-138 ; hw5.cu:322-530) so that small systems need no
 * host round trip per step.  bin/hw5 (csrc/main_hw5.cpp) is the CLI drop-in built on it; INTEGRATION.md
 * shows the two-line change that makes the reference's own main() call it.
 *
 * This file is the run_step boundary and nothing else: lifecycle, state, nb_step / nb_accel, the scenario drivers,
 * nb_solve and the state files.  What a host needs only when it owns device memory, collectives or several GPUs itself
 * — raw launches, the shared-pairs launches, nb_sharded_*, nb_solve_ex's options, the pair-schedule self-test — lives in
 * include/nbody_amd_ext.h (since ABI 5; the library exports both).
 *
 * Conventions
 *  - every function returns int: 0 = NB_OK, <0 = nb_status error; no exception crosses the boundary;
 *  - host arrays are caller-owned SoA `double[n]`, exactly run_step's vectors; the library copies and never
 *    keeps a host pointer after return;
 *  - the library owns device memory and its HIP stream; a context is bound to ONE GPU, is not thread-safe,
 *    distinct contexts may be driven from distinct host threads (how hw5.cu uses its GPUs: hw5.cu:564-567);
 *  - there is NO CPU fallback: without a HIP device nb_create fails with NB_ERR_NO_DEVICE.
 */
#ifndef NBODY_AMD_H
#define NBODY_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NB_ABI_VERSION 5

typedef enum nb_status {
    NB_OK = 0,
    NB_ERR_INVALID = -1,   /* bad argument (NULL, n <= 0, unknown enum, eps == 0 in an fp32 mode ...) */
    NB_ERR_NO_DEVICE = -2, /* no usable HIP device / bad ordinal */
    NB_ERR_HIP = -3,       /* a HIP call failed; nb_last_error() has the text */
    NB_ERR_STATE = -4,     /* call sequence error (e.g. step before set_state) */
    NB_ERR_NOMEM = -5,
    NB_ERR_IO = -6
} nb_status;

/* arithmetic the step is computed in */
typedef enum nb_precision {
    NB_F64 = 0,      /* everything fp64 — the testcases (|q| ~ 3e20 m; fp32 cannot represent them) */
    NB_F32 = 1,      /* fp32 state and pair arithmetic — large synthetic N */
    NB_F32_ACC64 = 2 /* fp32 pair arithmetic, per-tile partial sums added into fp64; fp64 q,v masters */
} nb_precision;

/* param:: of the reference — samples/nbody.cc:9-20 ; hw5.cu:50-67 */
typedef struct nb_config {
    int32_t n;         /* bodies */
    int32_t precision; /* nb_precision */
    int32_t device;    /* HIP device ordinal */
    int32_t f64_large_min; /* NB_F64: from this many bodies on, nb_step/nb_accel use the large-n kernel; 0 = default (12288 with eps > 0: every unordered pair once; 32768 otherwise) */
    int32_t f64_split;     /* NB_F64: lanes of a wave that share one target in the step kernel; 0 = auto, else a power of two <= 64 */
    int32_t flags;         /* nb_config_flags, 0 = defaults (ABI 3 carried a measurement knob here, ABI 4 required 0) */
    double G;   /* 6.674e-11 */
    double eps; /* 1e-3  (Plummer softening; r2 + eps*eps) */
    double dt;  /* 60 */
} nb_config;
typedef enum nb_config_flags {
    NB_CFG_ORDERED_PAIRS = 1 /* fp32 modes: nb_step / nb_accel evaluate every ORDERED pair (kernel K1, 0.3-1 KB of workspace per
                                body) even where the default applies.  Default from 28672 bodies on: every UNORDERED pair once
                                (kernel K1s, 1.35x faster), which needs a pair-slot workspace — 1.7 GB at n = 2^20, beyond that
                                720 B per body: 2.5 GB at 2^22, 10 GB at 2^24 (K1's source slices: 288-1056 B per body).  That workspace
                                is allocated by the FIRST nb_step / nb_accel, not by nb_create; if the device cannot give it
                                (it would take more than 3/4 of the free memory) the step goes in batches that fit (below), and
                                if even those do not, or hipMalloc fails, the context falls back to K1 for its lifetime;
                                nb_last_error(ctx) says so once either way — it is not an error */
} nb_config_flags;
/* fp32 modes, OR-ed into nb_config.flags: the caller's own limit for that workspace, in GiB (1..65535; 0 = none).  Within a
 * limit K1s steps in BATCHES of superblocks — several launches whose reducers add up a running force — instead of one launch:
 * more launches for less memory, at no measurable cost (profiles/r05_workspace_cap_ab.txt).  The default is already frugal
 * (one launch up to 2 GiB of slots — n = 2^20: 1.7 GB — then batches within 720 B per body: 2.5 GB at 2^22, 10 GB at 2^24);
 * without a limit the library shrinks the batches by itself when even that would take more than 3/4 of the free device
 * memory; only when not even batches of 16 superblocks fit (~600 B per body) does the context fall back to K1 */
#define NB_CFG_WORKSPACE_GIB(g) (((g) & 0xffff) << 8)

/* scenario drivers — the loops main() runs around run_step */
typedef enum nb_scenario_kind {
    NB_SCN_MIN_DIST = 0,  /* Problem 1 monitor: min over steps of |q_planet - q_asteroid|   nbody.cc:114-122 */
    NB_SCN_FIRST_HIT = 1, /* Problem 2: first step with d2 < R^2, stop there               nbody.cc:127-138;
                             also records each watched device's missile-arrival step + a
                             device-side snapshot of (q,v) at it                            hw5.cu:265-287 */
    NB_SCN_MISSILE = 2    /* Problem 3, one device: hit test, then arrival -> cost, m[d]=0  hw5.cu:289-309 */
} nb_scenario_kind;

#define NB_MAX_WATCH 16

typedef struct nb_scenario {
    int32_t kind;        /* nb_scenario_kind */
    int32_t first_step;  /* index of the state currently loaded (0 = the input); monitors run on it first */
    int32_t last_step;   /* inclusive; reference: n_steps = 200000 */
    int32_t planet;      /* body indices (file order) */
    int32_t asteroid;
    int32_t n_watch;               /* devices watched for missile arrival (FIRST_HIT: up to NB_MAX_WATCH; MISSILE: 0 or 1 —
                                      one device is destroyed per run, hw5.cu:289-309; more is NB_ERR_INVALID) */
    int32_t watch[NB_MAX_WATCH];   /* their body indices */
    int32_t sync_every;            /* host polls the hit flag every this many steps (hw5.cu:72: 2000); <=0 -> 2000 */
    int32_t engine;                /* 0 = auto; 1 = one launch per step (any n; long runs replay a hipGraph of launches);
                                      2 = whole step loop inside one persistent single-workgroup launch (n <= 128) */
    int32_t flags;                 /* NB_SCN_NO_SNAPSHOT: FIRST_HIT records arrival steps but keeps no (q,v) snapshots */
    int32_t graph_chunk;           /* per-step engine: launches per replayed hipGraph; 0 = 1000, else even, 2..4000.  A short
                                      chunk puts many replay boundaries (fill node, advance node, host poll, follower start)
                                      into a short run: that is what tests use it for; it does not make a replay traceable */
    double planet_radius;          /* 1e7   nbody.cc:17 */
    double missile_speed;          /* 1e6   nbody.cc:18 */
} nb_scenario;
#define NB_SCN_NO_SNAPSHOT 1
#define NB_SCN_EAGER 2 /* per-step engine: issue every launch from the host even for long runs (default: runs of >= 4000
                          steps replay a captured hipGraph of 1000 launches, the host looks at the monitors once per replay) */

typedef struct nb_scenario_result {
    double min_dist2;                     /* MIN_DIST: min squared planet–asteroid distance (sqrt on the host) */
    int32_t hit_step;                     /* FIRST_HIT / MISSILE: first hit step, -2 = none */
    int32_t steps_done;                   /* index of the last state computed: last_step, or the hit step when a hit ended it */
    int32_t arrival_step[NB_MAX_WATCH];   /* per watched device, -2 = never (before the hit) */
    double missile_cost[NB_MAX_WATCH];    /* 1e5 + 1e3*(arrival+1)*dt   hw5.cu:305 ; nbody.cc:19 */
} nb_scenario_result;

typedef struct nb_context nb_context;

/* ---- lifecycle ---- */
int nb_abi_version(void);
int nb_device_count(int* count);
int nb_config_default(nb_config* cfg); /* fills the reference's param:: values, F64, device 0 */
int nb_create(nb_context** out, const nb_config* cfg);
int nb_destroy(nb_context* ctx);
const char* nb_strerror(int code);
/* text of the last failure on this context; ctx == NULL: of the last failed context-free call made by the calling
 * thread (state files, nb_solve; nbody_amd_ext.h: raw launches, nb_sharded_create) */
const char* nb_last_error(const nb_context* ctx);

/* ---- state: the seven vectors of run_step + the `type[j]=="device"` predicate (nbody.cc:62) ---- */
int nb_set_state(nb_context* ctx, const double* qx, const double* qy, const double* qz, const double* vx,
                 const double* vy, const double* vz, const double* m, const uint8_t* is_device /* may be NULL */);
int nb_get_state(nb_context* ctx, double* qx, double* qy, double* qz, double* vx, double* vy, double* vz);
int nb_set_mass(nb_context* ctx, int index, double m); /* e.g. Problem 1's m[device] = 0, nbody.cc:109-113 */

/* ---- the hot path ---- */
/* run_step(step, ...) for step = first_step .. first_step+count-1 (nbody.cc:51-89): accelerations from the old
 * positions, v += a*dt, q += v*dt.  The step index feeds the device-mass law m0 + 0.5*m0*|sin(step*dt/6000)|. */
int nb_step(nb_context* ctx, int first_step, int count);
/* run_step(step, n, qx, qy, qz, vx, vy, vz, m, type) itself — samples/nbody.cc:51-54, the call main() makes at nbody.cc:116,129 —
 * as ONE entry: the state in the caller's seven arrays goes to the GPU, step `step` runs, the new q, v come back in place
 * (m and is_device are only read; is_device may be NULL).  Equal, bit for bit, to nb_set_state + nb_step(ctx, step, 1) +
 * nb_get_state, with one synchronisation instead of three — and without the upload when the arrays still hold what the
 * previous nb_run_step returned (the library compares them with its copy of that download; any other call on the context
 * in between makes it upload again), which is the reference's own loop: ~30 us per step for the testcases instead of ~70 */
int nb_run_step(nb_context* ctx, int step, double* qx, double* qy, double* qz, double* vx, double* vy, double* vz,
                const double* m, const uint8_t* is_device /* may be NULL */);
/* accelerations only (nbody.cc:56-74), no update; outputs double[n] each */
int nb_accel(nb_context* ctx, int step, double* ax, double* ay, double* az);
/* as nb_step, and reports the mean GPU time of one step's launches in milliseconds, measured with HIP events
 * recorded on the context's own stream around the `count` steps */
int nb_step_timed(nb_context* ctx, int first_step, int count, float* ms_per_step);

/* ---- scenario drivers (monitors evaluated on the GPU, no per-step host round trip) ----
 * After a scenario that ends in a hit the context's (q,v) are unspecified — the reference discards that state too
 * (nbody.cc:136 breaks out of the loop); reload with nb_set_state / nb_restore_snapshot / nb_load_state. */
int nb_run_scenario(nb_context* ctx, const nb_scenario* scn, nb_scenario_result* res);
/* `count` (<= 8) scenarios of equally sized NB_F64 systems on ONE GPU, one launch stream for all of them, each with its
 * own context (state), step range and monitors — hw5.cu's one-thread-per-device Problem-3 loop (hw5.cu:587-588) without a
 * launch stream per device.  Persistent engine (n <= 128): ONE launch, workgroup k runs scenario k to its end.  Per-step
 * engine: one launch per step serves all of them in lock step (blockIdx.y), replayed from a hipGraph for long runs.  All
 * scenario kinds (FIRST_HIT with snapshots included); every scns[k].engine must agree; results[k] as nb_run_scenario
 * would give for (ctxs[k], scns[k]). */
int nb_run_scenarios_batched(nb_context** ctxs, const nb_scenario* scns, nb_scenario_result* results, int count);
/* load the (q,v) snapshot that the last FIRST_HIT scenario on `src` took at watched device `watch_slot`'s
 * missile arrival into `dst` (same n, precision F64, same GPU or not); hw5.cu:482-484 */
int nb_restore_snapshot(nb_context* dst, nb_context* src, int watch_slot);

/* ---- binary state files (checkpoint / large-N input; the reference only has the text format, nbody.cc:22-49,
 *      and an in-memory snapshot, hw5.cu:265-287).  Layout, little endian, version 2:
 *        char magic[8] = "NBODYST2"; uint32 byte-order mark 0x01020304; int32 precision; int64 n; int32 step;
 *        int32 planet; int32 asteroid; int32 reserved; double G, eps, dt;                              (64 bytes)
 *        double q[3][n]; double v[3][n]; double m[n]; uint8 is_device[n]
 *      q,v are the fp64 masters (F64 / F32_ACC64) or the widened fp32 state (F32); planet/asteroid = the header of the
 *      text format (nbody.cc:27), -1 when the file is a plain checkpoint.  Version-1 files ("NBODYST1", 48-byte header
 *      without byte-order mark and planet/asteroid) are still read.  bin/hw5 accepts such a file as <input>. ---- */
typedef struct nb_state_header {
    int64_t n;
    int32_t precision; /* nb_precision of the run that wrote it */
    int32_t step;      /* index of the state */
    int32_t planet;    /* body indices of the scenario, -1 = not recorded */
    int32_t asteroid;
    double G, eps, dt;
} nb_state_header;
int nb_save_state(nb_context* ctx, const char* path, int step);
/* resumes a checkpoint: n, precision, G, eps and dt of the file must equal the context's (NB_ERR_INVALID otherwise,
 * nb_last_error(ctx) says which); to start a run from a state file under other parameters use nb_read_state_file +
 * nb_set_state */
int nb_load_state(nb_context* ctx, const char* path, int* step);
int nb_state_file_info(const char* path, int64_t* n, int* precision, int* step);
/* host-array access to a state file (no GPU involved).  Reading: all array pointers NULL -> header only; otherwise the
 * seven double arrays must hold `capacity` >= n elements each (is_device may be NULL). */
int nb_read_state_file(const char* path, nb_state_header* hdr, int64_t capacity, double* qx, double* qy, double* qz,
                       double* vx, double* vy, double* vz, double* m, uint8_t* is_device);
int nb_write_state_file(const char* path, const nb_state_header* hdr, const double* qx, const double* qy,
                        const double* qz, const double* vx, const double* vy, const double* vz, const double* m,
                        const uint8_t* is_device /* may be NULL */);

/* ---- whole reference program: P1, P2, P3 (nbody.cc:106-146 ; hw5.cu:532-606) ----
 * n <= 128: all 2 + D scenarios (P1, P2, one Problem-3 run per gravity device) start at step 0 in ONE persistent launch
 * per GPU, a workgroup per scenario, at most 8 per launch; scenarios beyond that are queued cheapest-first (ascending
 * missile-arrival step, hw5.cu:574-585) and dropped once they cannot beat a feasible device (hw5.cu:490-493).  `devices`
 * spreads the scenarios over several GPUs (the reference's task parallelism, hw5.cu:564-567,587-588).
 * n > 128: the per-step engine instead — P1, P2 and the Problem-3 runs each replay their own graph of launches on their
 * own stream (one shared graph per GPU up to 256 bodies); a Problem-3 run starts from the snapshot P2 takes at its
 * missile's arrival (hw5.cu:265-287,482-489) as soon as P2's monitor shows it, one per GPU at a time in arrival order.
 * Tuning and test hooks travel in nb_solve_options (nb_solve_ex, nbody_amd_ext.h); the library reads ONE environment variable,
 * NB_SOLVE_TRACE=1: a timeline of the driver's phases on stderr, no change of behaviour.  bin/hw5 maps its NB_SOLVE_* /
 * NB_GRAPH_CHUNK environment onto the options (csrc/main_hw5.cpp). */
typedef struct nb_answer {
    double min_dist;
    int32_t hit_time_step;
    int32_t gravity_device_id;
    double missile_cost;
} nb_answer;
int nb_solve(int n, int planet, int asteroid, const double* qx, const double* qy, const double* qz,
             const double* vx, const double* vy, const double* vz, const double* m, const uint8_t* is_device,
             const int* devices /* HIP ordinals to spread scenarios over */, int n_devices, nb_answer* out);
#ifdef __cplusplus
}
#endif
#endif /* NBODY_AMD_H */
