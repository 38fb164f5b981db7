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

#define NB_ABI_VERSION 4

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
    int32_t f64_large_min; /* NB_F64: from this many bodies on, nb_step/nb_accel use the large-n kernel; 0 = default (32768) */
    int32_t f64_split;     /* NB_F64: lanes of a wave that share one target in the step kernel; 0 = auto, else a power of two <= 64 */
    int32_t reserved;      /* must be 0 (ABI 3 carried a measurement knob here; it lives in nbody_amd_debug.h now) */
    double G;   /* 6.674e-11 */
    double eps; /* 1e-3  (Plummer softening; r2 + eps*eps) */
    double dt;  /* 60 */
} nb_config;

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
 * thread (raw launches, state files, nb_solve, nb_sharded_create) */
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
 * Tuning and test hooks travel in nb_solve_options (nb_solve_ex); the library reads ONE environment variable,
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
typedef struct nb_solve_options { /* zero-initialise; every 0 = the default nb_solve uses */
    int32_t max_batch;   /* persistent engine: scenarios per launch, 2..8 (default 8); fewer queues the other devices */
    int32_t engine;      /* 0 = by system size; 1 = per-step engine (graph replay); 2 = persistent engine (n <= 128) */
    int32_t streams;     /* per-step engine: 0 = by system size; 1 = one shared graph per GPU; 2 = a stream per scenario */
    int32_t p3_parallel; /* per-step engine: Problem-3 runs at a time; 0 = one per listed GPU (hw5.cu:587-588) */
    int32_t graph_chunk; /* per-step engine: launches per replayed graph; 0 = 1000, else even, 2..4000 */
    int32_t handoff;     /* nb_solve_handoff: how P2's arrival snapshot reaches a Problem-3 run (hw5.cu:482-484) */
    int32_t reserved[2];
} nb_solve_options;
typedef enum nb_solve_handoff {
    NB_HANDOFF_AUTO = 0,       /* device copy on the same GPU ordinal, through host memory between different ones */
    NB_HANDOFF_HOST_STAGED = 1 /* through host memory whenever the run sits on another entry of `devices` than P2, even if
                                  both entries name the same GPU: executes the cross-GPU path on a one-GPU box */
} nb_solve_handoff;
int nb_solve_ex(int n, int planet, int asteroid, const double* qx, const double* qy, const double* qz,
                const double* vx, const double* vy, const double* vz, const double* m, const uint8_t* is_device,
                const int* devices, int n_devices, const nb_solve_options* options /* NULL = defaults */, nb_answer* out);

/* ---- raw launches on caller-owned HBM (device pointers + a hipStream_t as void*) ----
 * For hosts that own device memory and the exchange step themselves (one process per GPU with
 * torch.distributed/RCCL: bench.py, nbody_amd.distributed).  fp32 body record = float4 {x, y, z, G*m}.
 *
 *   src      float4[n_src]   all source bodies (the gathered array every rank holds)
 *   tgt_off  first target index in src; targets are src[tgt_off .. tgt_off+n_tgt)   (unless `tgt` is given)
 *   out      float4[n_src]   the OTHER (ping-pong) gathered array; only [tgt_off, tgt_off+n_tgt) is written
 *   vel      float4[n_tgt]   this rank's velocities, in place ({vx,vy,vz,unused})
 *   F32_ACC64 additionally keeps fp64 masters: pos64/vel64 = double4[n_tgt] ({x,y,z,G*m} / {vx,vy,vz,0})
 */
typedef struct nb_launch_f32 {
    const void* src;
    void* out;
    void* vel;
    void* pos64; /* NULL unless acc64 */
    void* vel64; /* NULL unless acc64 */
    void* acc;   /* nb_launch_accel_f32 only: float4[n_tgt] {ax,ay,az,0} (acc64: double4[n_tgt]) */
    void* workspace; /* optional scratch for source slicing (partial + running sums); NULL -> never slice */
    int64_t workspace_bytes; /* its size; must be >= nb_workspace_bytes_f32() (18 records per target: running sum,
                                compensation, 16 partial-sum slots) or the sources are not sliced.  A larger one — up to 66
                                records — gives a launch as many slots, so that a step of up to 64 slices is ONE force launch +
                                ONE reducer instead of j_split/16 of each (results are bit for bit the same).  The running sum
                                and its compensation are records 0 and 1 whatever the size, so the FIRST / MIDDLE / LAST
                                launches of one step may pass different sizes of the same buffer */
    int64_t n_src;
    int64_t tgt_off;
    int64_t n_tgt;
    float eps2;
    float dt;
    int32_t acc64;            /* 0 = NB_F32, 1 = NB_F32_ACC64 */
    int32_t targets_per_lane; /* 0 = auto; 2, 4 or 8 (packed pairs of targets per lane) */
    int32_t j_split;          /* 0 = auto; 1..1024 source slices (~1 MiB each when auto): workgroups sharing a target
                                 block each take one slice; 16 slices per launch, partial sums folded by a reducer */
    int32_t source_path;      /* 0 = auto; 1 = sources through the LDS tile; 2 = sources through scalar loads/SGPRs (both: every
                                 ORDERED pair, kernel K1); 3 = every UNORDERED pair once, Newton's third law (kernel K1s: the
                                 sources travel through the wave by DPP rotation) — needs the whole system in this one launch
                                 (n_tgt == n_src, tgt_off 0, phase WHOLE), n_src >= 49152 and a workspace of
                                 nb_workspace_bytes_sym_f32(); auto picks it whenever that holds and nothing else is forced;
                                 j_split then = workgroups per 4096-body superblock (0 = auto) */
    int32_t wg_size;          /* 0 = auto; 256, 512 (targets_per_lane 8) or 1024 (targets_per_lane 4) */
    int32_t phase;            /* nb_launch_phase: a step may be cut into several launches over disjoint source ranges
                                 (own shard while the all-gather of the other shards is still in flight);
                                 the running sums live in `workspace` between them (required unless NB_PHASE_WHOLE) */
    int64_t src_begin;        /* sources of this launch: src[src_begin .. src_end); 0,0 = all n_src.  src_begin must be */
    int64_t src_end;          /* a multiple of 256, src_end a multiple of 256 or n_src */
    const void* tgt;          /* NULL: the targets are src[tgt_off .. tgt_off+n_tgt).  Otherwise float4[n_tgt], the targets'
                                 own records, for hosts whose sources travel in blocks (ring pass: `src` is the block in
                                 hand, tgt_off then only places the result in `out`, and may exceed n_src) */
} nb_launch_f32;
typedef enum nb_launch_phase {
    NB_PHASE_WHOLE = 0,  /* the whole step in one call: start the sums, run the epilogue */
    NB_PHASE_FIRST = 1,  /* start the sums, keep them in the workspace */
    NB_PHASE_LAST = 2,   /* continue the sums, then the epilogue (accelerations out / kick-drift) */
    NB_PHASE_MIDDLE = 3  /* continue the sums, keep them */
} nb_launch_phase;
int nb_launch_step_f32(const nb_launch_f32* a, void* hip_stream);  /* force + fused kick-drift */
int nb_launch_accel_f32(const nb_launch_f32* a, void* hip_stream); /* force only -> a->acc */
/* name of the kernel symbol the two launches above resolve to for these arguments (for matching rocprofv3 rows) */
const char* nb_kernel_name_f32(const nb_launch_f32* a, int accel_only);
/* the register blocking, source split and workgroup size the launches above will use for these arguments */
int nb_plan_f32(const nb_launch_f32* a, int* targets_per_lane, int* j_split, int* wg_size);
/* workspace size that allows source slicing for n_tgt targets: 18 records per target (2 + 16 slots) */
int64_t nb_workspace_bytes_f32(int64_t n_tgt, int acc64);
/* workspace size that lets a whole-system launch of n bodies use K1s (source_path 3): three floats per body and
 * superblock round plus three (six with acc64) per workgroup of a superblock — 12 B x (n/8192 + 8) per body: 1.7 GB at
 * n = 2^20, 26 GB at 2^22; larger systems are stepped in batches of superblocks with a running force behind the slots: 52 GB
 * at 2^23 and 2^24, 107 GB at 2^26; 0 = K1s does not apply to this n (fewer than 49152 bodies, or no batch fits 128 GiB) */
int64_t nb_workspace_bytes_sym_f32(int64_t n, int acc64);

/* ---- K1s over several GPUs, for hosts that own the collectives themselves (one process per GPU: nbody_amd.distributed).
 * The GPUs share the UNORDERED pairs of the system: rank r = tgt_off / n_tgt of P = n_src / n_tgt takes the 4096-body
 * superblocks of its shard against the half of the system behind each.  Needs whole superblocks per shard
 * (n_src % (P * 4096) == 0) and n_src >= 49152: nb_workspace_bytes_shared_pairs_f32 answers 0 otherwise (use the ordered
 * launches above).  Per step and rank:
 *   nb_launch_pair_forces_f32   a->acc = float4[n_src] (double4 with acc64): this rank's partial force on ALL bodies
 *   reduce-scatter (sum) of a->acc over the ranks -> the force on the rank's own shard
 *   nb_launch_kick_drift_f32    a->acc = that force as float4[parts][n_tgt] (double4 with acc64), the pieces added in order
 *                               (1 after a reduce-scatter; P when the host gathered the ranks' pieces itself); kick + drift
 *                               of [tgt_off, tgt_off + n_tgt) into a->out / a->vel (pos64 / vel64), as nb_launch_step_f32 does
 *   all-gather of the positions, as with the ordered launches */
int nb_launch_pair_forces_f32(const nb_launch_f32* a, void* hip_stream);
int nb_launch_kick_drift_f32(const nb_launch_f32* a, int parts, void* hip_stream);
int64_t nb_workspace_bytes_shared_pairs_f32(int64_t n_src, int ranks, int acc64);
/* host-only replay of K1s' pair schedule for n bodies on `ranks` GPUs of n_cus compute units (ranks 1 = the one-GPU launch),
 * with the index arithmetic the kernels share: every unordered pair of 4096-body superblocks met exactly once in every tile
 * phase over all ranks and workgroups, no slot region written twice, the reducer's slot list equal to what was written.
 * Needs no GPU — it is how the 8-GPU shapes are checked on machines that have one or none.  NB_OK, or NB_ERR_STATE with the
 * first inconsistency in msg */
int nb_selftest_pair_schedule(int64_t n, int n_cus, int ranks, int acc64, char* msg, int msg_len);

/* ---- index-sharded multi-GPU stepping: ONE process, P GPUs of a node, RCCL over xGMI (csrc/nbody_sharded.cpp) ----
 * The reference's only multi-GPU use is task parallelism (hw5.cu:564-567,587-588); this is the data-parallel scheme of
 * GPU r owns targets [r*n/P, (r+1)*n/P) (velocities, fp64 masters), every GPU holds all positions twice
 * (ping-pong float4[n] {x,y,z,G*m}).  Per step and GPU, on its own stream:
 *   default (whole 4096-body superblocks per shard, n >= 131072, no overlap): its share of the UNORDERED pairs of the system
 *     (kernel K1s) -> a partial force on all n bodies -> ONE ncclReduceScatter (sum) to the shard owners -> kick-drift;
 *   otherwise / NB_SHARDED_ORDERED_PAIRS: every ordered pair of its own targets with the kick-drift fused (kernel K1);
 *   then ONE in-place ncclAllGather(sendbuff = recvbuff + r*4n/P, ncclFloat) of the positions.
 * RCCL is loaded (dlopen) by the first nb_sharded_create.  The same scheme with one process per GPU: nbody_amd.distributed
 * (torch.distributed).  n must be divisible by n_devices; precision NB_F32 or NB_F32_ACC64; with one device the trajectory
 * equals nb_step's bit for bit. */
typedef struct nb_sharded nb_sharded;
#define NB_SHARDED_OVERLAP 1 /* two-phase step: own-shard sources while the all-gather of the other shards is in flight
                                on a second stream, remote sources after it; n/P must be a multiple of 256 */
#define NB_SHARDED_COPY_EXCHANGE 2 /* the per-step all-gather as P-1 peer copies per GPU (hipMemcpyPeerAsync on the exchange
                                stream: SDMA engines over xGMI, no CU taken from the force kernel, RCCL not loaded) instead of
                                ncclAllGather.  The only exchange that accepts an ordinal more than once in `devices` — ranks
                                sharing a GPU, each with its own streams and arrays — which is how a one-GPU box executes the
                                P > 1 host logic (tests/test_gpu_sharded_native.py) */
#define NB_SHARDED_ORDERED_PAIRS 4 /* every GPU evaluates every ordered pair of its targets (kernel K1) even where the default
                                applies: when every shard is a whole number of 4096-body superblocks, n >= 49152 and the step is
                                not overlapped, the GPUs share the UNORDERED pairs of the system instead (kernel K1s: GPU r takes
                                the superblocks of its shard against the half of the system behind each), which leaves every GPU
                                with a partial force on all n bodies — one reduce-scatter per step (ncclReduceScatter, or peer
                                copies + an ordered sum with NB_SHARDED_COPY_EXCHANGE) in front of the kick-drift and the all-gather */
int nb_sharded_create(nb_sharded** out, const int* devices, int n_devices, int64_t n, int precision, double G,
                      double eps, double dt, int flags);
int nb_sharded_destroy(nb_sharded* s);
const char* nb_sharded_last_error(const nb_sharded* s); /* s == NULL: the calling thread's last failed create */
/* host arrays of ALL n bodies, as nb_set_state / nb_get_state (no `device` bodies in the fp32 modes) */
int nb_sharded_set_state(nb_sharded* s, const double* qx, const double* qy, const double* qz, const double* vx,
                         const double* vy, const double* vz, const double* m);
int nb_sharded_get_state(nb_sharded* s, double* qx, double* qy, double* qz, double* vx, double* vy, double* vz);
int nb_sharded_step(nb_sharded* s, int count); /* `count` run_steps of the whole system; returns with all GPUs idle */
/* as nb_sharded_step, and reports the host wall time per step in milliseconds (all GPUs idle on both sides) */
int nb_sharded_step_timed(nb_sharded* s, int count, double* ms_per_step);
/* as nb_sharded_step_timed, and additionally reports — per rank — the mean GPU time of one step's launch sequence
 * (force kernels + reducers), from HIP events recorded on that rank's own compute stream around the launches of every step,
 * the exchange excluded (with NB_SHARDED_OVERLAP the span contains the wait for the gathered remote shards between the
 * own-shard phase and the remote phases).  kernel_ms: float[n_devices].  count <= 1024 (two events per step and rank). */
int nb_sharded_step_profiled(nb_sharded* s, int count, double* wall_ms_per_step, float* kernel_ms);
/* shard size and the launch plan each GPU uses for a whole step (any pointer may be NULL) */
int nb_sharded_info(const nb_sharded* s, int* n_devices, int64_t* targets_per_device, int* targets_per_lane,
                    int* j_split, int* wg_size);
/* symbol name of the force kernel a rank's step launches (for matching rocprofv3 rows, like nb_kernel_name_f32) */
const char* nb_sharded_kernel_name(const nb_sharded* s);
/* who rank `rank` is: which GPU it drives (ordinal, PCI bus id, UUID, name), which targets it owns, and what its exchange
 * is — for RCCL straight from the rank's communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice), so that
 * "the collective ran over N ranks on N distinct GPUs" is a fact read back from RCCL, not an echo of the arguments */
typedef enum nb_sharded_exchange { NB_EXCHANGE_RCCL = 1, NB_EXCHANGE_COPY = 2 } nb_sharded_exchange;
typedef struct nb_sharded_rank {
    int32_t device;        /* HIP ordinal */
    int32_t compute_units;
    int64_t first_target;  /* owns targets [first_target, first_target + targets) */
    int64_t targets;
    int32_t exchange;      /* nb_sharded_exchange */
    int32_t comm_ranks;    /* RCCL: ncclCommCount of this rank's communicator; copy exchange: n_devices */
    int32_t comm_rank;     /* RCCL: ncclCommUserRank; copy exchange: rank */
    int32_t comm_device;   /* RCCL: ncclCommCuDevice; copy exchange: device */
    char pci_bus_id[16];   /* "0000:05:00.0" */
    char uuid[36];         /* hipDeviceGetUuid, 32 hex digits */
    char name[64];
} nb_sharded_rank;
int nb_sharded_rank_info(const nb_sharded* s, int rank, nb_sharded_rank* out);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_AMD_H */
