/*
 * nbody_amd_ext.h — the EXTENDED surface of libnbody_amd.so, for hosts that are more than a caller of run_step:
 *   nb_solve_ex            nb_solve with its tuning / test options                                   (csrc/nbody_solve.cpp)
 *   nb_launch_*_f32        raw launches on caller-owned HBM: one process per GPU that owns device memory and the
 *                          collectives itself (nbody_amd.distributed, bench.py)                      (csrc/nbody_launch.cpp)
 *   shared-pairs launches  K1s over several GPUs behind the caller's own reduce-scatter, and the host-only replay of the
 *                          pair schedule                                                             (csrc/nbody_launch.cpp)
 *   nb_sharded_*           index-sharded multi-GPU stepping, ONE process driving P GPUs over RCCL    (csrc/nbody_sharded.cpp)
 * The run_step boundary itself — what INTEGRATION.md §2 binds: lifecycle, state, nb_step / nb_accel, scenarios, nb_solve,
 * state files — is include/nbody_amd.h (included here).  The reference has none of this (its only multi-GPU use is task
 * parallelism, hw5.cu:564-567,587-588); every entry below says what it adds and which reference lines it stands next to.
 * Same conventions as the core header: int status returns, plain pointers and sizes, no exception crosses the boundary.
 */
#ifndef NBODY_AMD_EXT_H
#define NBODY_AMD_EXT_H

#include "nbody_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- nb_solve with options (the whole reference program, nbody.cc:106-146 ; hw5.cu:532-606: see nb_solve) ---- */
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

/* symbol name of the force kernel nb_step / nb_accel of this fp32 context launch (for matching rocprofv3 rows, and to see
 * whether the context runs K1s or — NB_CFG_ORDERED_PAIRS, fewer than 28672 bodies, or its pair-slot workspace could not be
 * had — K1).  Asks for the workspace like the first step would; "" for NB_F64 contexts */
const char* nb_context_kernel_name(nb_context* ctx);

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
                                 (n_tgt == n_src, tgt_off 0, phase WHOLE), n_src >= 28672 and a workspace of
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
/* workspace size with which a whole-system launch of n bodies runs K1s (source_path 3) in its FASTEST shape — a smaller one, down
 * to roughly 50 slots of 12 n bytes, still runs K1s, in batches of superblocks that fit it (about a percent slower per doubling
 * of the batch count); below that the launch falls to K1.  Up to 2 GiB one launch with a slot per superblock round — three
 * floats per body and round plus three (six with acc64) per workgroup of a superblock: 12 B x (n/8192 + 16) per body, 1.8 GB at
 * n = 2^20 — beyond that batches of 32 superblocks within 720 B per body: 2.5 GB at 2^22, 5 GB at 2^23, 10 GB at 2^24, 40 GB at
 * 2^26 (rounds 1-4: a slot per round up to 32 GiB — 26 GB at 2^22 — then 52 GB; the smaller shapes measure 0.8 % FASTER);
 * 0 = K1s does not apply to this n (fewer than 28672 bodies) */
int64_t nb_workspace_bytes_sym_f32(int64_t n, int acc64);

/* ---- K1s over several GPUs, for hosts that own the collectives themselves (one process per GPU: nbody_amd.distributed).
 * The GPUs share the UNORDERED pairs of the system: rank r = tgt_off / n_tgt of P = n_src / n_tgt takes the 4096-body
 * superblocks of its shard against the half of the system behind each.  Needs whole superblocks per shard
 * (n_src % (P * 4096) == 0), n_src >= 28672 and n_src^2 / P >= 1.1e9 (below, a rank's ordered-pair step is faster than its share):
 * nb_workspace_bytes_shared_pairs_f32 answers 0 otherwise (use the ordered
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
/* the launch shape nb_launch_pair_forces_f32 uses for one rank of `ranks` on the CURRENT device (its compute-unit count picks the
 * workgroup count): 4096-body superblocks a rank owns, workgroups per superblock, and the sub-launches a rank's share goes out
 * in (the grid of one launch = superblocks / sub-launches x workgroups per superblock; 1 sub-launch unless a slot per
 * superblock would exceed 2 GiB / 720 B per body: configs[3] over 8 GPUs = 4 x 32 superblocks, configs[4] = 16 x 32).
 * NB_ERR_INVALID when the ranks cannot share the pairs (nb_workspace_bytes_shared_pairs_f32 answers 0).  What a
 * host reports as its plan — bench.py — instead of re-deriving it (any pointer may be NULL) */
int nb_plan_shared_pairs_f32(int64_t n_src, int ranks, int acc64, int* superblocks_per_rank, int* workgroups_per_superblock,
                             int* sub_launches);
/* host-only replay of K1s' pair schedule for n bodies on `ranks` GPUs of n_cus compute units (ranks 1 = the one-GPU launch),
 * with the index arithmetic the kernels share: every unordered pair of 4096-body superblocks met exactly once in every tile
 * phase over all ranks and workgroups, no slot region written twice, the reducer's slot list equal to what was written.
 * Needs no GPU — it is how the 8-GPU shapes are checked on machines that have one or none.  NB_OK, or NB_ERR_STATE with the
 * first inconsistency in msg */
int nb_selftest_pair_schedule(int64_t n, int n_cus, int ranks, int acc64, char* msg, int msg_len);
/* the same for ONE GPU whose K1s workspace is limited to workspace_bytes (a raw launch handed a smaller workspace than
 * nb_workspace_bytes_sym_f32, nb_config's NB_CFG_WORKSPACE_GIB, or a device short of memory): the batches of superblocks chosen
 * for that budget must cover every pair once and fit it */
int nb_selftest_pair_schedule_within(int64_t n, int n_cus, int acc64, int64_t workspace_bytes, char* msg, int msg_len);

/* ---- index-sharded multi-GPU stepping: ONE process, P GPUs of a node, RCCL over xGMI (csrc/nbody_sharded.cpp) ----
 * The reference's only multi-GPU use is task parallelism (hw5.cu:564-567,587-588); this is the data-parallel scheme of
 * GPU r owns targets [r*n/P, (r+1)*n/P) (velocities, fp64 masters), every GPU holds all positions twice
 * (ping-pong float4[n] {x,y,z,G*m}).  Per step and GPU, on its own stream:
 *   default (whole 4096-body superblocks per shard, n >= 28672, no overlap): its share of the UNORDERED pairs of the system
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
                                applies: when every shard is a whole number of 4096-body superblocks, n >= 28672 and the step is
                                not overlapped, the GPUs share the UNORDERED pairs of the system instead (kernel K1s: GPU r takes
                                the superblocks of its shard against the half of the system behind each), which leaves every GPU
                                with a partial force on all n bodies — one reduce-scatter per step (ncclReduceScatter, or peer
                                copies + an ordered sum with NB_SHARDED_COPY_EXCHANGE) in front of the kick-drift and the all-gather */
#define NB_SHARDED_HOST_EXCHANGE 8 /* last resort for a node whose peer-to-peer path does not work: the all-gather through ONE pinned
                                host array — every GPU downloads its slot, the host waits for all of them, every GPU uploads the
                                others' — only per-device copies, no peer mapping, no cross-device event, RCCL not loaded.  Ordered
                                pairs only (kernel K1), not overlapped; accepts an ordinal more than once like the copy exchange.
                                Costs a host round trip per step and 16 n bytes over PCIe per GPU and step (2.4 ms of a 30 ms step at n = 2^20, P = 8) */
int nb_sharded_create(nb_sharded** out, const int* devices, int n_devices, int64_t n, int precision, double G,
                      double eps, double dt, int flags);
int nb_sharded_destroy(nb_sharded* s);
/* Bound every wait of this system: with seconds > 0 the host never blocks inside the HIP runtime — it polls (hipEventQuery /
 * hipStreamQuery) and gives up when ONE step has not finished `seconds` after the host started waiting for it (at most 16 steps
 * are in flight; a wait for an upload, download or the final drain of the streams gets the same allowance).  The call in
 * progress then returns NB_ERR_HIP with nb_sharded_last_error() = "... timed out after N s: <what was waited for>"; the system
 * is dead afterwards (every later call returns NB_ERR_STATE) and nb_sharded_destroy gives the GPUs one more allowance to drain
 * before it abandons — does not free — what they may still be using.  For hosts that must hand a verdict to someone (bench.py's
 * multi-GPU legs: a collective that never completes — a wedged ncclReduceScatter, a peer copy over a dead link — ends the
 * child process with a message instead of hanging it until its parent's timeout).  0 = wait as long as it takes (default) */
int nb_sharded_set_deadline(nb_sharded* s, double seconds);
const char* nb_sharded_last_error(const nb_sharded* s); /* s == NULL: the calling thread's last failed create */
/* host arrays of ALL n bodies, as nb_set_state / nb_get_state (no `device` bodies in the fp32 modes) */
int nb_sharded_set_state(nb_sharded* s, const double* qx, const double* qy, const double* qz, const double* vx,
                         const double* vy, const double* vz, const double* m);
int nb_sharded_get_state(nb_sharded* s, double* qx, double* qy, double* qz, double* vx, double* vy, double* vz);
int nb_sharded_step(nb_sharded* s, int count); /* `count` run_steps of the whole system; returns with all GPUs idle */
/* checkpoint / resume of a sharded run: ONE NBODYST2 state file for the whole system (the format of nb_save_state; q, v = the fp64
 * masters or the widened fp32 state, m exactly as nb_sharded_set_state got it), written beside the previous one and renamed over
 * it.  nb_sharded_load_state refuses a file whose n, precision, G, eps or dt differ from the system's; a resumed run continues
 * bit for bit.  (The reference has no files, only its in-memory Problem-3 snapshot, hw5.cu:265-287.) */
int nb_sharded_save_state(nb_sharded* s, const char* path, int step);
int nb_sharded_load_state(nb_sharded* s, const char* path, int* step /* may be NULL */);
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
typedef enum nb_sharded_exchange { NB_EXCHANGE_RCCL = 1, NB_EXCHANGE_COPY = 2, NB_EXCHANGE_HOST = 3 } nb_sharded_exchange;
typedef struct nb_sharded_rank {
    int32_t device;        /* HIP ordinal */
    int32_t compute_units;
    int64_t first_target;  /* owns targets [first_target, first_target + targets) */
    int64_t targets;
    int32_t exchange;      /* nb_sharded_exchange */
    int32_t comm_ranks;    /* RCCL: ncclCommCount of this rank's communicator; copy / host exchange: n_devices */
    int32_t comm_rank;     /* RCCL: ncclCommUserRank; copy / host exchange: rank */
    int32_t comm_device;   /* RCCL: ncclCommCuDevice; copy / host exchange: device */
    char pci_bus_id[16];   /* "0000:05:00.0" */
    char uuid[36];         /* hipDeviceGetUuid, 32 hex digits */
    char name[64];
} nb_sharded_rank;
int nb_sharded_rank_info(const nb_sharded* s, int rank, nb_sharded_rank* out);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_AMD_EXT_H */
