// nbody_internal.h — what the translation units behind the C ABI share (not installed; the public surface is
// include/nbody_amd.h + include/nbody_amd_ext.h, the kernel interface nbody_kernels.h):
//   nbody_capi.cpp       contexts, state, nb_step / nb_accel                          run_step call sites, nbody.cc:116,129
//   nbody_launch.cpp     raw launches on caller-owned HBM (nb_launch_*_f32, shared-pairs pair)
//   nbody_scenario.cpp   scenario drivers: persistent engine, per-step engine, graph replay, the Problem-3 follower queue
//                                                                                     nbody.cc:114-138 ; hw5.cu:366-404,489-508
//   nbody_solve.cpp      nb_solve: the whole program                                  nbody.cc:91-146 ; hw5.cu:532-606
//   nbody_statefile.cpp  NBODYST1/2 binary state files
//   nbody_sharded.cpp    index-sharded multi-GPU stepping (self-contained)
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../../include/nbody_amd_ext.h"  // (includes nbody_amd.h)
#include "nbody_kernels.h"

#ifndef NB_ABI_DEBUG
#define NB_ABI_DEBUG NB_STEP_STAMPS  // the instrumented build also exports include/nbody_amd_debug.h
#endif

struct nb_context {
    nb_config cfg;
    int n = 0;
    int n_cus = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = true;  // false: borrowed from another context of the same GPU (nb_solve: a stream costs ~8 ms to
                              // create, and only the leader of a launch stream ever enqueues on it)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool have_state = false;
    char err[512] = {0};

    // F64: SoA planes in HBM, exactly run_step's vectors: q[3][n] (ping-pong), v[3][n], m[n], coef[n]
    double* q[2] = {nullptr, nullptr};
    int cur = 0;
    double* v = nullptr;
    double* m = nullptr;
    double* coef = nullptr;
    double* acc = nullptr;  // [3][n] scratch for nb_accel
    nbk::F64Monitor* mon = nullptr;
    nbk::F64Monitor* mon_host = nullptr;  // pinned
    double* snap_q = nullptr;             // [n_watch][3][n]
    double* snap_v = nullptr;
    int snap_slots = 0;
    int snap_arrival[NB_MAX_WATCH];  // per snapshot slot: arrival step of the last FIRST_HIT scenario, -2 = holds nothing
    int split = 1;
    double* gm_large = nullptr;       // K1-f64 (n > F64_LARGE_MIN): G*m_eff scratch [n]
    double* partial_large = nullptr;  // ... and partial sums [slices][3][n]
    int slices_large = 1;
    double* sym64_slots = nullptr;    // K1s-f64: pair slots (eps > 0 and sym64_workspace_bytes(n) > 0), else K1-f64 runs
    double* fst_dev = nullptr;  // |sin(step*dt/6000)| table, steps 0 .. fst_len-1, host-computed (glibc)
    int fst_len = 0;
    double* fst_chunk = nullptr;  // graph-driven stepping: |sin| of steps base .. base + chunk + 1, refilled per replay
    int fst_chunk_len = 0;
    int* done_dev = nullptr;
    int* done_host = nullptr;         // pinned
    nbk::F64Ctl* ctl_host = nullptr;  // pinned staging copy of *ctl
    nbk::F64Ctl* ctl = nullptr;  // graph-driven stepping: {base step, active} read by every launch of a replayed graph
    unsigned long long* stamps = nullptr;  // nb_enable_step_stamps: [2 * stamp_slots] device words, else null
    int stamp_slots = 0;
    void* arena = nullptr;       // F64: ONE device allocation behind q, v, m, coef, acc, mon, done_dev, ctl ...
    void* host_arena = nullptr;  // ... and one pinned allocation behind mon_host, done_host (a context costs two
                                 // allocations instead of ten: nb_solve creates 2 + D of them per program run)
    std::vector<double> m_host;
    std::vector<uint8_t> dev_host;
    std::vector<double> stage_host;  // NB_F64, n <= 65536: nb_set_state / nb_get_state staging (3 / 2 copies instead of 8 / 6)
    bool stage_fresh = false;        // stage_host[0 .. 6n) is what nb_run_step last downloaded AND no other call has touched
                                     // the context since (every entry point passes through bind(), which clears this)

    // F32 / F32_ACC64: float4 {x,y,z,G*m} ping-pong, float4 velocities, optional double4 masters
    float4* pos[2] = {nullptr, nullptr};
    float4* vel = nullptr;
    double4* pos64 = nullptr;
    double4* vel64 = nullptr;
    void* acc32 = nullptr;
    void* pinned = nullptr;                        // two pinned staging halves (nb_set_state / nb_get_state of the fp32 modes)
    size_t pin_half = 0;                           // ... of this many bytes each (<= 16 MiB)
    hipEvent_t pin_ev[2] = {nullptr, nullptr};     // ... and the event that says a half's copies are done
    void* partial = nullptr;  // workspace: source slices [2 + partial_slots][n] float4 (double4 for ACC64) — allocated by
                              // nb_create — or, from the first nb_step / nb_accel of a system of SYM_MIN_N bodies or more, the
                              // (larger) pair-slot workspace of the symmetric kernel K1s, which holds the slices too
    int partial_slots = 0;
    size_t partial_bytes = 0;
    size_t k1_bytes = 0;      // what K1's source slices need ([2 + partial_slots][n] records; 0: the plan does not slice)
    size_t sym_bytes = 0;     // what K1s needs (0: does not apply, NB_CFG_ORDERED_PAIRS, or given up)
    bool sym_tried = false;   // the pair-slot workspace was asked for once (lazily: ensure_sym_workspace)
};

namespace nbi {

// ---- errors: per context, or per host thread for calls without one (read with nb_last_error(NULL))
char* thread_error();  // the calling thread's buffer, 512 bytes
int set_error(int code, const char* text);
int fail_hip(nb_context* c, hipError_t e, const char* what);
#define NB_HIP(ctx, call)                                           \
    do {                                                            \
        hipError_t e_ = (call);                                     \
        if (e_ != hipSuccess) return nbi::fail_hip(ctx, e_, #call); \
    } while (0)

int bind(nb_context* c);  // hipSetDevice(the context's GPU)

// |sin(step*dt/6000)| with glibc, the value samples/nbody.cc:15,63 feeds gravity_device_mass
inline double fst_of(int step, double dt) { return std::fabs(std::sin(step * dt / 6000)); }

template <class T>
void free_dev(T*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

// `borrowed`: use this stream (of another context on the same GPU, which must outlive this one) instead of creating one
int create_context(nb_context** out, const nb_config* cfg, hipStream_t borrowed, int cu_mask = 0);
nbk::F64Args base_args(nb_context* c, int step);  // one plain fp64 step launch of this context

// NB_SOLVE_TRACE=1 in the environment: timeline of the drivers' phases on stderr.  The ONLY environment variable the
// library reads; it changes no behaviour.
bool trace_enabled();

// The pinned monitor / done words are written by device-to-host copies that may still be in flight for a LATER replay
// while the host looks at them: read them through an atomic load so that the compiler can neither hoist nor tear it.
// (What is read is always usable: the words only move forward — first hit, first arrival — and a visible value was
// written by a replay that is complete.)
inline int load_word(const int* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }

// ---- scenario unit (nbody_scenario.cpp)
struct GraphGroup;
struct GraphSlot {
    nb_context* c = nullptr;
    const nb_scenario* scn = nullptr;
    nbk::F64Scenario sc{};
    bool snap = false;
    int base = 0;        // index of the state the slot's buffers hold (host mirror of ctl.base_step)
    bool active = true;  // false: dormant follower
    int done_at = -1;    // >= 0: finished; index of the last state computed
    int inflight = 0;    // replays enqueued with this slot active and not yet collected
    bool cancelled = false;  // follower dropped because a device that arrived earlier turned out feasible
    int gpu_slot = 0;        // index into the caller's device list (two slots may name the same ordinal)
    // follower: a MISSILE run that starts from the snapshot which slot `parent_slot` of `parent` (a FIRST_HIT scenario
    // with snapshots) takes when the missile of its watched device `parent_watch` arrives (hw5.cu:265-287,482-489)
    GraphGroup* parent = nullptr;
    int parent_slot = -1, parent_watch = -1;
};

struct GraphGroup {  // the scenarios that share one stream and one replayed graph
    std::vector<GraphSlot> slots;
    nb_context* lead = nullptr;  // owns the stream and the |sin| table, reports errors
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};  // end of the replays in flight (even / odd)
    int launched = 0, collected = 0;
    int chunk = 0;  // steps per replay (even: the ping-pong buffers are back in place after a chunk); 0 = default
    nbk::F64CtlBatch cb{};  // control words and per-replay |sin| arrays of the slots (valid once prepared)
    bool prepared = false;
    ~GraphGroup();
    bool running() const;
    bool anything_active() const;
};

struct FollowerPolicy {
    int parallel = 1 << 30;          // Problem-3 runs at a time (the reference: one per GPU, hw5.cu:587-588)
    bool stage_through_host = false;  // hand P2's snapshot to a follower on another device SLOT through host memory even
                                      // when both slots are the same physical GPU (the cross-GPU path of hw5.cu:482-484)
};
// all groups to completion, one host thread
int run_groups_graph(std::vector<GraphGroup*>& groups, const FollowerPolicy& policy);
bool valid_graph_chunk(int chunk);

nbk::F64Scenario device_scenario(const nb_context* c, const nb_scenario* s);
void reset_monitor_host(nb_context* c);
void fill_result(nb_context* c, const nb_scenario* s, const nbk::F64Scenario& sc, int steps_done, nb_scenario_result* res);

}  // namespace nbi
