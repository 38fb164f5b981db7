// nbody_kernels.h — internal interface between the C-ABI layer (nbody_capi.cpp) and the gfx950 kernels.
// Not installed; the public surface is include/nbody_amd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#ifndef NB_STEP_STAMPS
#define NB_STEP_STAMPS 0  // 1: the per-step fp64 kernel records clock stamps (instrumented build, `make stamps`)
#endif

namespace nbk {

constexpr int WG = 256;    // threads per workgroup: 4 wave64, one per SIMD of a CU
constexpr int TILE = 256;  // source bodies staged in LDS per tile (one per thread, one coalesced 16-B load each)
constexpr int MAX_WATCH = 16;

// ---------------------------------------------------------------- fp32 / fp32+fp64acc (large N)
struct F32Args {
    const float4* src;  // [n_src] {x,y,z,G*m}
    const float4* tgt;  // [n_tgt] the targets' own records: src + tgt_off (launch_f32 fills it in when null), or a separate
                        // array when the sources of this launch are a block that travels (ring pass)
    float4* out;        // [n_src] other ping-pong array; [tgt_off, tgt_off+n_tgt) written
    float4* vel;        // [n_tgt]
    double4* pos64;     // [n_tgt] fp64 masters (ACC64 only)
    double4* vel64;     // [n_tgt]
    void* acc;          // accel-only output
    void* partial;      // source-slice workspace, records float4 (double4 if acc64):
                        //   [2][n_tgt] running sum / compensation + [slots][n_tgt] partial sums of one launch
    long n_src, tgt_off, n_tgt;
    long src_begin, src_end;  // sources of THIS launch sequence: [src_begin, src_end) (src_begin a multiple of 256);
                              // 0,0 = all n_src.  A step may be cut into several such phases (own shard first, the
                              // gathered remote shards later): the running sum in the workspace carries across them
    int slots;                // partial-sum slots the workspace holds = blockIdx.y extent of one split launch: 0 -> the default
                              // SLICES_PER_LAUNCH; a larger workspace (up to MAX_SLICES_PER_LAUNCH) lets a step with more
                              // slices go out as ONE launch + ONE reducer instead of js/16 of each
    int phase;                // F32_PHASE_* below
    long tiles_per_slice;  // SPLIT launches: 256-source tiles per source slice ...
    int slice0;            // ... and the index of the slice blockIdx.y == 0 works on
    float eps2, dt;
};
// The fp32 kernels evaluate the self pair and the zero-mass padding as d = 0 times G*m*rinv^3: that is exactly +0 only while
// rinv^3 = eps2^-1.5 is finite in fp32 (eps2 >= 4.4e-26) and eps2 is a normal number (v_rsq_f32 flushes denormals); below
// it every body would get 0 * inf = NaN.  Refused at the ABI: eps >= 1e-12.
constexpr float F32_EPS2_MIN = 1e-24f;
constexpr int F32_PHASE_WHOLE = 0;   // all of the step: start the sums, run the epilogue
constexpr int F32_PHASE_FIRST = 1;   // start the sums, keep them in the workspace
constexpr int F32_PHASE_LAST = 2;    // continue the sums, then the epilogue (store accelerations / kick-drift)
constexpr int F32_PHASE_MIDDLE = 3;  // continue the sums, keep them
constexpr int SLICES_PER_LAUNCH = 16;  // default blockIdx.y extent of one split launch = partial-sum slots in the workspace
constexpr int MAX_SLICES_PER_LAUNCH = 64;
constexpr int MAX_JSPLIT = 1024;       // source slices per step (processed SLICES_PER_LAUNCH at a time)
// ---- K1s (nbody_kernels_f32_sym.hip): every unordered pair once — Newton's third law
constexpr int SYM_P = 4, SYM_WGS = 512;            // packed target pairs per lane, threads per workgroup (8 waves)
constexpr int SYM_SB = SYM_WGS * 2 * SYM_P;        // superblock: 4096 bodies, the targets one workgroup holds in registers
#ifndef NB_SYM_MIN_N
#define NB_SYM_MIN_N 28672
#endif
constexpr long SYM_MIN_N = NB_SYM_MIN_N;           // 7 superblocks.  K1s has a floor per step (a workgroup does at least SYM_MIN_PHASES tile
                                                   // phases of ~42 us), K1 grows with n^2.  Round 5, with 8 phases (floor 0.33 ms): K1 wins by
                                                   // 18 % at 32768 bodies, K1s by 26 % at 36864 (profiles/r05_small_n_sym_ab.txt; rounds 3-4 started
                                                   // K1s at 49152); with 4 phases (floor 0.18 ms, profiles/r05_sym_min_phases_ab.txt) and K1's
                                                   // modelled slices: K1 wins by 20 % at 24499, K1s by 22 % at 28672, 24 % at 32768, 33 % at 36864
#ifndef NB_SYM_SHARE_MIN
#define NB_SYM_SHARE_MIN 1.1e9
#endif
constexpr double SYM_SHARE_MIN_N2_PER_RANK = NB_SYM_SHARE_MIN;  // several GPUs share the unordered pairs from n^2 / P >= this on (sym_sharded_ok)
constexpr size_t SYM_MAX_WORKSPACE = (size_t)128 << 30;  // absolute ceiling of a K1s workspace (what it takes by default: sym_batch_budget)
struct F32SymShape {  // who computes what in one launch
    int B;         // superblocks covering the system
    int b0, nb;    // I-superblocks this launch owns: [b0, b0 + nb)   (one GPU: 0, B)
    int chunks;    // workgroups per I-superblock: the tile phases of its 1 + rounds work units are cut evenly among them
    int by_super;  // slot of a finished superblock pair: 0 = round - 1 (one GPU: B/2 slots), 1 = I-superblock - b0 (several
                   // GPUs share the pairs: nb slots)
    long npad;     // B * SYM_SB: bodies per slot plane (a slot = three planes x, y, z of npad floats)
    int cus;       // compute units the chunk count was chosen for
};
__host__ __device__ inline int sym_own_slots(const F32SymShape& s, bool acc64) { return s.chunks * (acc64 ? 2 : 1); }
// slots: own sums per chunk | finished superblock pairs (by round, or by I-superblock) | per chunk the second part of the
// round that straddles it and the chunk before (the work is cut at tile-phase granularity)
__host__ __device__ inline int sym_tail_slot(const F32SymShape& s, bool acc64, int chunk) {
    return sym_own_slots(s, acc64) + (s.by_super ? s.nb : s.B / 2) + chunk;
}
__host__ __device__ inline int sym_total_slots(const F32SymShape& s, bool acc64) { return sym_tail_slot(s, acc64, s.chunks); }

// ---- the pair schedule, shared by the force kernel, the reducer and the host-side self-test (nb_selftest_pair_schedule)
constexpr int SYM_NT = SYM_SB / 128;  // tile phases of one work unit (a wave meets one 128-source tile per phase)
#ifndef NB_SYM_MIN_PHASES
#define NB_SYM_MIN_PHASES 4
#endif
// tile phases a workgroup executes at least: bounds the workgroups per superblock, i.e. how well a system of few superblocks fills the
// chip.  8 until round 5 (a quarter of a work unit); the pair schedule holds for any value (a workgroup has at most one piece that does
// not start its round whatever its length: nb_selftest_pair_schedule, builds with 4 and 2) and 4 measures 24 % faster at 36864
// bodies (0.342 -> 0.277 ms: 0.51 -> 0.63 of peak), 32 % for K1s-f64 at 16384, the same from 40960 on; 2 adds nothing where K1s is
// used (profiles/r05_sym_min_phases_ab.txt)
constexpr int SYM_MIN_PHASES = NB_SYM_MIN_PHASES;
// rounds of I-superblock b: J = b + r (mod B), r = 1 .. (B-1)/2, plus r = B/2 for the lower half when B is even
__host__ __device__ inline int sym_rounds(int B, int b) { return (B - 1) / 2 + ((B % 2 == 0 && b < B / 2) ? 1 : 0); }
// the phases [q_lo, q_hi) of superblock b's work (unit u = q / SYM_NT: 0 = the diagonal block, r = round r) that
// workgroup `chunk` executes.  The cut points are those of a superblock with the most rounds, the same for every b.
__host__ __device__ inline void sym_chunk_range(const F32SymShape& s, int b, int chunk, long* q_lo, long* q_hi) {
    const long Q = (long)SYM_NT * (1 + s.B / 2), q_end = (long)SYM_NT * (1 + sym_rounds(s.B, b));
    const long lo = chunk * Q / s.chunks, hi = (chunk + 1) * Q / s.chunks;
    *q_lo = lo < q_end ? lo : q_end;
    *q_hi = hi < q_end ? hi : q_end;
}
// the piece that starts at phase q: unit u, phases [ph0, ph1) of it; returns the phase the next piece starts at
__host__ __device__ inline long sym_piece(long q, long q_hi, int* u, int* ph0, int* ph1) {
    *u = (int)(q / SYM_NT);
    *ph0 = (int)(q % SYM_NT);
    *ph1 = (q_hi - q) < (SYM_NT - *ph0) ? *ph0 + (int)(q_hi - q) : SYM_NT;
    return q + (*ph1 - *ph0);
}
// slot of the image a piece of round u (u >= 1) leaves: the second part of a round that straddles two workgroups goes to
// the tail slot of the workgroup that starts with it
__host__ __device__ inline int sym_piece_slot(const F32SymShape& s, bool acc64, int b, int chunk, int u, int ph0) {
    return ph0 ? sym_tail_slot(s, acc64, chunk) : sym_own_slots(s, acc64) + (s.by_super ? b - s.b0 : u - 1);
}
// every slot of this launch that holds a contribution for the bodies of superblock J, in the order the reducer adds them
template <class F>
__host__ __device__ inline void sym_for_each_slot_of(const F32SymShape& s, bool acc64, int J, F&& f) {
    const int B = s.B, own = sym_own_slots(s, acc64);
    if (J >= s.b0 && J < s.b0 + s.nb)
        for (int c = 0; c < own; ++c) f(c);  // own sums (fp64 sums: the float and the float of its remainder)
    if (s.by_super) {  // a slot per producing I-superblock of this launch
        for (int k = 0; k < s.nb; ++k) {
            const int b = s.b0 + k, r = ((J - b) % B + B) % B;
            if (r >= 1 && r <= sym_rounds(B, b)) f(own + k);
        }
    } else {  // a slot per round: producer b = J - r
        for (int r = 1; r <= B / 2; ++r) {
            const int b = ((J - r) % B + B) % B;
            if (b >= s.b0 && b < s.b0 + s.nb && r <= sym_rounds(B, b)) f(own + r - 1);
        }
    }
    // second parts of the rounds that straddle two workgroups: chunk c starts inside round q_c / SYM_NT (the same for every
    // superblock), whose producer for superblock J is J - that round
    const long Q = (long)SYM_NT * (1 + B / 2);
    for (int c = 1; c < s.chunks; ++c) {
        const long q = c * Q / s.chunks;
        const int r = (int)(q / SYM_NT);
        if (q % SYM_NT == 0 || r == 0) continue;  // cut at a unit boundary, or inside the diagonal block (no image)
        const int b = ((J - r) % B + B) % B;
        if (b >= s.b0 && b < s.b0 + s.nb && r <= sym_rounds(B, b)) f(sym_tail_slot(s, acc64, c));
    }
}
F32SymShape sym_shape(long n, int n_cus, int b0 = 0, int nb = 0, int force_chunks = 0, int sb = SYM_SB);
size_t sym_workspace_bytes(const F32SymShape& s, bool acc64);
// How much workspace K1s takes by default (round 5: linear in n, no longer n^2).  A slot per superblock ROUND lets the whole
// system go out as ONE launch — n^2/8192 x 12 B of slots: 1.7 GB at n = 2^20, 26 GB at 2^22, 103 GB at 2^23.  Rounds 1-4 did
// that up to 32 GiB and cut larger systems into batches of I-superblocks within 64 GiB.  Measured this round (same box,
// alternating, profiles/r05_workspace_cap_ab.txt): batches cost NOTHING — n = 2^22 in 32 batches of 32 superblocks (8
// workgroups each: every launch still fills the chip) with 2.4 GB of slots steps in 2774.2 ms against 2771.7-2775.1 ms for the
// one launch with 26 GB; every batch's reducer adds its slots to a running force (32 B per body and batch: 0.02 % of the
// step).  So: one launch while its slots fit SYM_WHOLE_WORKSPACE (2 GiB: up to ~1.2e6 bodies, the shape every measurement at
// the metric's N was made with), beyond that the largest batches within 720 B per body — 48 / 56 slots of 12 B (fp32 /
// fp64 sums: 32 superblocks x 8 workgroups, or 16 x 16) + the running force: 3.0 GB at 2^22, 12 GB at 2^24, linear in n.
// -DNB_SYM_WHOLE_GIB=32 -DNB_SYM_BATCH_FLOOR_GIB=64 rebuilds rounds 1-4's shapes for an A/B.
#ifndef NB_SYM_WHOLE_GIB
#define NB_SYM_WHOLE_GIB 2
#endif
#ifndef NB_SYM_BATCH_FLOOR_GIB
#define NB_SYM_BATCH_FLOOR_GIB 0
#endif
constexpr size_t SYM_WHOLE_WORKSPACE = (size_t)NB_SYM_WHOLE_GIB << 30;
constexpr size_t SYM_BYTES_PER_BODY = 832;  // the BUDGET: 64 slots + the running force + margin.  What a 256-CU part then takes is 592 B
                                            // (fp32 sums: 48 slots) / 704 B (fp64 sums: 56 slots) per body — the "720 B per body" of the
                                            // documentation; 304 CUs need 62 slots with fp64 sums (found by the hypothesis self-test)
inline size_t sym_batch_budget(long npad) {  // of a batched step, and of the sub-launches of one rank's share of a multi-GPU step
    size_t b = (size_t)SYM_BYTES_PER_BODY * (size_t)npad;
    if (b < SYM_WHOLE_WORKSPACE) b = SYM_WHOLE_WORKSPACE;
    if (b < ((size_t)NB_SYM_BATCH_FLOOR_GIB << 30)) b = (size_t)NB_SYM_BATCH_FLOOR_GIB << 30;
    return b;
}
// One GPU: larger systems go in batches of `nb` I-superblocks — each a launch like one rank of a multi-GPU step, a slot per
// I-superblock of the batch — whose reducers add up a running force kept behind the slots; the last one runs the epilogue.
// bytes = workspace needed.
struct F32SymBatches {
    int nb = 0;       // I-superblocks per batch (= B when count == 1)
    int count = 0;    // 0: K1s does not apply (too small, or no batch fits SYM_MAX_WORKSPACE)
    size_t bytes = 0;
};
// budget = 0: the defaults above (the fastest shape: as few, as large launches as 32 / 64 GiB allow).  budget > 0 (round 5): the
// workspace the caller actually has — one launch if a slot per round fits it, else the largest batch of whole rounds of
// workgroups (failing that a half, a quarter ... of the CU count, down to 16 superblocks) whose slots and running force do:
// memory for speed, about a percent per doubling of the batch count (profiles/r05_workspace_cap_ab.txt); count == 0 if not
// even 16 superblocks per batch fit
F32SymBatches sym_batches(long n, int n_cus, bool acc64, size_t budget = 0);
// a launch that leaves a partial force (mode 2: one rank of a multi-GPU step): its I-superblocks go in sub-launches of
// sym_sub_batch(s) superblocks when a slot for each of them would not fit sym_batch_budget (configs[3] over 8 GPUs: 128
// superblocks per rank = 6.6 GB -> 4 x 32 in 2.4 GB; configs[4]: 512 per rank = 103 GB -> 16 x 32 in 11 GB);
// sym_partial_workspace_bytes = the workspace such a launch needs
int sym_sub_batch(const F32SymShape& s, bool acc64);
// superblocks [b0, b0 + nb) of launch s as a launch of their own.  max_bytes > 0: with no more workgroups per superblock than its
// slots fit into max_bytes (a short last sub-launch would otherwise be cut into up to 64 workgroups per superblock and want more
// own / tail slots than the full-sized ones before it)
F32SymShape sym_sub_shape(const F32SymShape& s, int b0, int nb, bool acc64 = false, size_t max_bytes = 0);
// batch k of a one-GPU step that goes in kb.count batches (the same rule for its last, shorter batch)
F32SymShape sym_batch_shape(long n, int n_cus, const F32SymBatches& kb, int k, bool acc64);
size_t sym_partial_workspace_bytes(const F32SymShape& s, bool acc64);
// mode 0: force + kick-drift of the whole system; 1: accelerations out; 2: this launch's partial force out (a.acc:
// float4[n] / double4[n]) for the reduce-scatter of a multi-GPU step.  a.partial = the slot workspace.
int launch_f32_sym(const F32Args& a, const F32SymShape& s, bool acc64, int mode, hipStream_t stream);  // hipError_t
// the whole system on one GPU, in kb.count launches (mode 0 or 1)
int launch_f32_sym_batched(const F32Args& a, const F32SymBatches& kb, int n_cus, bool acc64, int mode, hipStream_t stream);
// kick-drift of targets [tgt_off, tgt_off + n_tgt) from accelerations arriving as a.acc[parts][n_tgt], added in order
int launch_kick_drift_f32(const F32Args& a, bool acc64, int parts, hipStream_t stream);
// several GPUs sharing the pairs of one system: every GPU owns whole superblocks (n % (P * SYM_SB) == 0), n >= SYM_MIN_N,
// slot workspace within SYM_MAX_WORKSPACE; *shape_of_rank0 = the launch shape of rank 0 (rank r: b0 = r * nb)
bool sym_sharded_ok(long n, int P, int n_cus, bool acc64, F32SymShape* shape_of_rank0);

struct F32Plan {
    int targets_per_lane = 4;  // 2, 4 or 8 (one, two or four packed pairs per lane)
    int j_split = 1;           // source slices: workgroups sharing one target block, each over 1/j_split of the sources
    bool sgpr_sources = true;  // sources via scalar loads into SGPRs (default) instead of the LDS tile
    int wg_size = 256;         // threads per workgroup: 256, 512 (R = 8) or 1024 (R = 4); LDS path: 256
    bool symmetric = false;    // K1s instead of K1 (whole-system launches with a slot workspace; plan_symmetric)
    F32SymShape sym{};         // shape of the (first) launch
    F32SymBatches sym_batches{};
    int sym_cus = 256;
};
// K1s for this launch?  Needs the whole system as targets AND sources in one launch (n_tgt == n_src, no phases, no
// travelling target block), SYM_MIN_N bodies or more and a workspace of sym_workspace_bytes.  source_path: 0 = auto (yes
// when eligible), 3 = required (false if not eligible, the plan is left alone), 1 / 2 = never.
bool plan_symmetric(F32Plan& p, long n_tgt, long n_src, bool whole, size_t workspace_bytes, bool acc64, int n_cus,
                    int source_path, int force_chunks);
// source_path: 0 = auto (SGPR), 1 = LDS tile, 2 = SGPR (3 = K1s, see plan_symmetric) ; force_wg: 0 = auto
F32Plan plan_f32(long n_tgt, long n_src, int n_cus, int force_tpl, int force_js, bool have_workspace,
                 int source_path = 0, int force_wg = 0, int max_slices = MAX_SLICES_PER_LAUNCH);
int launch_f32(const F32Args& a, const F32Plan& plan, bool acc64, bool accel_only, hipStream_t stream);  // hipError_t
const char* kernel_name_f32(const F32Plan& plan, bool acc64, bool accel_only);

// ---------------------------------------------------------------- fp64 (testcases, n <= a few thousand)
// The fp64 kernels drop the `j == i` compare when the self pair adds +0 by itself — d = 0 times G*m_j * (eps2)^-1.5, which
// needs that product finite: eps2 >= 1e-160 leaves 1e240 x G*m up to 1e68.  Below (eps = 0 included) the SELFCHECK
// instantiations run, which skip the self pair explicitly like the reference (nbody.cc:59).
constexpr double F64_EPS2_MIN = 1e-160;
struct F64Monitor {  // device-resident scenario state, written by workgroup 0 only
    double min_d2;
    int hit_step;
    int arrival_step[MAX_WATCH];
    int pad;
};

struct F64Ctl {  // device-resident control word of a graph-driven scenario: a captured launch cannot carry its step index
    int base_step;  // the launch with offset t in the graph computes step base_step + t
    int active;     // 0: the slot is dormant (e.g. a Problem-3 run whose missile has not arrived yet)
};

struct F64Scenario {  // by-value kernel argument
    int kind;         // nb_scenario_kind, or -1 = no monitor (plain nb_step)
    int planet, asteroid;
    int n_watch;
    int watch[MAX_WATCH];
    int destroy_on_arrival;  // MISSILE: watched device's mass becomes 0 from the step after its arrival
    double R2;               // planet_radius^2
    double missile_dstep;    // missile_speed * dt  (distance per step index; hw5.cu:274,303)
};

struct F64Args {
    const double* qin;   // [3][n] planes x,y,z : state after step-1
    double* qout;        // [3][n] state after `step`
    double* v;           // [3][n] in place
    const double* m;     // [n]
    const double* coef;  // [n] 0.5 for `device` bodies else 0 : m_eff = m + (coef*m)*fst
    double* acc_out;     // [3][n] accel-only mode (update skipped) or nullptr
    double* snap_q;      // [n_watch][3][n] snapshots at missile arrival (FIRST_HIT) or nullptr
    double* snap_v;
    F64Monitor* mon;
    int n;
    int step;       // index of the step this launch computes (monitors run on state step-1 first)
    int do_update;  // 0 = monitor-only launch for the final state
    double fst;     // |sin(step*dt/6000)| computed on the host (glibc, as the CPU reference does)
    double G, eps2, dt;
    F64Scenario scn;
    // graph-driven stepping (ctl != nullptr): step = ctl->base_step + t instead of `step`, |sin| of that step from
    // `fst_chunk[t]` instead of `fst` — a per-scenario array the graph's own fill node rewrites for the next replay from the
    // host-computed (glibc) table, so that the value sits at an address known at capture time and its load does not wait
    // for the control word; the update runs while step <= last_step, step == last_step + 1 is the monitor-only launch for
    // the final state, later launches (and dormant slots) return at once
    const F64Ctl* ctl;
    const double* fst_chunk;
    int t, last_step;
#if NB_STEP_STAMPS
    // measurement hook of the instrumented build (libnbody_amd_stamps.so, nb_enable_step_stamps): thread 0 of workgroup 0
    // writes the 100 MHz wall clock at kernel entry and after its last store into stamps[2k], stamps[2k+1], k = (t-1 for a
    // graph node, step for an eager launch) mod stamp_slots — per-launch duration and launch-to-launch gap of the replayed
    // chain.  Not in the product build: the extra kernarg words and the branch cost the latency-bound step 7-13 %
    // (profiles/r03_step_stamps_cost.txt).
    unsigned long long* stamps;
    int stamp_slots;
#endif
};
int launch_f64(const F64Args& a, int S, hipStream_t stream);  // S = lanes sharing one target (1..64, pow2)
constexpr int MAX_BATCH = 8;
struct F64BatchArgs {  // up to MAX_BATCH independent systems of the same n, one step each per launch (blockIdx.y)
    F64Args item[MAX_BATCH];  // item[k].n == 0 marks an idle slot
    int count;
};
int launch_f64_batched(const F64BatchArgs& b, int n, int S, hipStream_t stream);
struct F64CtlBatch {
    F64Ctl* ctl[MAX_BATCH];
    double* fst_chunk[MAX_BATCH];  // [chunk + 2] per slot: fst_chunk[t] = |sin| of step base_step + t
    int count;
};
int launch_ctl_advance(const F64CtlBatch& b, int by, hipStream_t stream);  // base_step += by for the active slots
// fst_chunk[k][t] = table[base_step_k + (active_k ? by : 0) + t] for t = 0 .. chunk + 1 (index clamped to the table):
// with by = chunk the values of the NEXT replay (the node runs before launch_ctl_advance), with by = 0 a (re)fill
int launch_fst_fill(const F64CtlBatch& b, int by, int chunk, const double* table, int table_len, hipStream_t stream);
int auto_split_f64(int n, int n_cus);

// K1-f64: fp64 force + kick-drift for LARGE n (plain nb_step / nb_accel from F64_LARGE_MIN bodies up): sources broadcast
// from SGPRs like the fp32 K1, R = 2 targets per lane, source slices over blockIdx.y with a reducer
constexpr int F64_LARGE_MIN = 32768;  // measured crossover with K2 ~ 2e4 bodies (profiles/r01_f64_step_timing.txt)
struct F64LargeArgs {
    const double* q;    // [3][n] state after step-1
    double* qout;       // [3][n]
    double* v;          // [3][n] in place
    const double* m;    // [n]
    const double* coef; // [n]
    double* gm;         // [n] scratch: G*m_eff of this step (written by the launch sequence itself)
    double* acc_out;    // [3][n] accel-only, or nullptr
    double* partial;    // [j_split][3][n] partial sums (j_split > 1)
    int n;
    int j_split;
    double fst, G, eps2, dt;
    double* sym_slots;  // K1s-f64: pair-slot workspace of sym64_workspace_bytes, or nullptr (K1-f64)
};
int launch_f64_large(const F64LargeArgs& a, hipStream_t stream);
int plan_f64_large_slices(int n, int n_cus);
// K1s-f64 (nbody_kernels_f64_sym.hip): the fp64 member with every unordered pair evaluated once — the scheme of K1s with one
// travelling source per lane (a wave meets 64-source tiles), R = 4 targets per lane, superblocks of 2048 bodies, everything
// (register sums, LDS image, slots) in fp64.  Taken for eps > 0 when a.sym_slots (the slot workspace, sym64_workspace_bytes)
// is given.
constexpr int SYM64_SB = 2048;
#ifndef NB_SYM64_MIN_SB
#define NB_SYM64_MIN_SB 6
#endif
// K1s-f64 from this many superblocks on: 6 = 12288 bodies since the end of round 5 (round 4: 16; mid-round 5, with workgroups of at
// least 8 tile phases: 8).  With 4-phase workgroups (A/B build, profiles/r05_f64_mid_n_sweep.txt, last table): 12288 bodies 0.128 ms
// against K2's best 0.137, 14336 0.130 against 0.194, 10240 0.122 against 0.105 (K2 wins).  Mid-round, measured with an A/B build
// (-DNB_SYM64_MIN_SB=4, bench/f64_mid_n_sweep.py, profiles/r05_f64_mid_n_sweep.txt): K1s-f64 has a floor of ~0.21 ms per step
// (few workgroups) and passes the per-step kernel K2 between 14336 and 16384 bodies — 0.218 ms against 0.231 at 16384, 0.294
// against 0.50 at 24576, 0.42 against 0.80 at 30720
constexpr int SYM64_MIN_SB = NB_SYM64_MIN_SB;
constexpr size_t SYM64_MAX_WORKSPACE = (size_t)64 << 30;  // ceiling; by default 0.42 GB at n = 2^18 (one launch), then 1440 B per body
size_t sym64_workspace_bytes(int n, int n_cus);  // 0: not applicable (fewer than SYM64_MIN_SB superblocks of 2048 bodies)
int launch_f64_large_sym(const F64LargeArgs& a, int n_cus, hipStream_t stream);

// K3: whole scenario of a small system (n <= SMALL_N_MAX) in ONE persistent single-workgroup launch
constexpr int SMALL_N_MAX = 128;
struct F64SmallArgs {
    double* q;           // [3][n] in/out: state `first_step` in, last computed state out
    double* v;           // [3][n] in/out
    const double* m;     // [n]
    const double* coef;  // [n]
    const double* fst;   // [>= last_step + 3] |sin(step*dt/6000)| by step index, host-computed (glibc)
    double* snap_q;      // [n_watch][3][n] or nullptr
    double* snap_v;
    F64Monitor* mon;     // in/out (carries min_d2 / hit / arrivals across launches)
    int* steps_done;     // out: index of the last state computed
    int n;
    int first_step;      // index of the state in q,v
    int last_step;       // inclusive
    int final_monitor;   // evaluate the monitor on state last_step too (last launch of a scenario)
    double G, eps2, dt;
    F64Scenario scn;
};
int launch_f64_small(const F64SmallArgs& a, hipStream_t stream);
struct F64SmallBatchArgs {  // one workgroup per scenario (blockIdx.x); item[k].n == 0 marks a finished slot
    F64SmallArgs item[MAX_BATCH];
    int count;
};
int launch_f64_small_batched(const F64SmallBatchArgs& b, int n, hipStream_t stream);

}  // namespace nbk
