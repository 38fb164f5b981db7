// nbody_solve.cpp — nb_solve / nb_solve_ex: the whole reference program, main() of samples/nbody.cc:91-146 and
// hw5.cu:532-606 — Problem 1 (min distance, devices massless), Problem 2 (first hit), Problem 3 (cheapest gravity device
// whose destruction avoids the hit) — on top of the scenario drivers (nbody_scenario.cpp).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <new>
#include <thread>
#include <vector>

#include "nbody_internal.h"

using namespace nbk;
using namespace nbi;

namespace {

struct CtxDeleter {
    void operator()(nb_context* c) const {
        if (c) nb_destroy(c);
    }
};
using CtxPtr = std::unique_ptr<nb_context, CtxDeleter>;

struct CtxList {  // destroyed newest first: a context that borrows a stream dies before the context that owns it
    std::vector<CtxPtr> v;
    ~CtxList() {
        while (!v.empty()) v.pop_back();
    }
};

struct ThreadJoiner {  // no exception may leave joinable threads behind (std::terminate)
    std::vector<std::thread>& ts;
    ~ThreadJoiner() {
        for (auto& t : ts)
            if (t.joinable()) t.join();
    }
};

struct SolveSlot {  // one scenario of the program: P1, P2 or the Problem-3 run of one device
    nb_scenario scn{};
    nb_scenario_result res{};
    int device_k = -1;  // MISSILE: index into the device list
    bool zero_devices = false;  // P1: devices massless (nbody.cc:109-113)
    int rc = NB_OK;
    bool ran = false;
    char err[256] = {0};
};

struct SolveInput {
    int n, planet, asteroid;
    const double *qx, *qy, *qz, *vx, *vy, *vz, *m;
    const uint8_t* is_device;
    const std::vector<double>* m_no_devices;
    void (*stamp)(const char*);  // NB_SOLVE_TRACE timeline, or nullptr
};

// all scenarios of `slots` on one GPU: contexts + one batched launch stream, at most `cap` scenarios at a time
void run_group(const SolveInput& in, int gpu, const std::vector<SolveSlot*>& slots, int cap) {
    try {
        for (size_t at = 0; at < slots.size(); at += (size_t)cap) {
            const int cnt = (int)std::min(slots.size() - at, (size_t)cap);
            CtxList list;
            std::vector<CtxPtr>& owned = list.v;
            nb_context* ctxs[MAX_BATCH];
            nb_scenario scns[MAX_BATCH];
            nb_scenario_result ress[MAX_BATCH];
            int rc = NB_OK;
            for (int k = 0; k < cnt && !rc; ++k) {
                SolveSlot* s = slots[at + k];
                nb_config cfg;
                nb_config_default(&cfg);
                cfg.n = in.n;
                cfg.device = gpu;
                nb_context* c = nullptr;
                // one launch stream serves the whole batch: only its leader creates a stream
                rc = create_context(&c, &cfg, k > 0 && ctxs[0] ? ctxs[0]->stream : nullptr);
                owned.emplace_back(c);
                if (!rc)
                    rc = nb_set_state(c, in.qx, in.qy, in.qz, in.vx, in.vy, in.vz,
                                      s->zero_devices ? in.m_no_devices->data() : in.m, in.is_device);
                ctxs[k] = c;
                scns[k] = s->scn;
            }
            const nb_context* failed = (rc && !owned.empty()) ? owned.back().get() : nullptr;  // set-up failure
            if (in.stamp) in.stamp("contexts created, state uploaded");
            if (!rc) {
                rc = nb_run_scenarios_batched(ctxs, scns, ress, cnt);
                if (in.stamp) in.stamp("batched scenarios returned");
                if (rc) failed = ctxs[0];  // the batch reports through its leader
            }
            for (int k = 0; k < cnt; ++k) {
                if (failed) snprintf(slots[at + k]->err, sizeof slots[at + k]->err, "%s", nb_last_error(failed));
                slots[at + k]->rc = rc;
                slots[at + k]->res = ress[k];
                slots[at + k]->ran = true;
            }
            if (rc) return;
        }
    } catch (...) {
        for (SolveSlot* s : slots)
            if (!s->ran) s->rc = NB_ERR_NOMEM;
    }
}

// groups[g] runs on gpus[g]: inline for one GPU, one host thread per GPU otherwise (hw5.cu:564-567,587-588)
void run_groups(const SolveInput& in, const std::vector<int>& gpus, const std::vector<std::vector<SolveSlot*>>& groups,
                int cap) {
    std::vector<std::thread> ts;
    ThreadJoiner join{ts};
    for (size_t g = 1; g < groups.size(); ++g)
        if (!groups[g].empty()) ts.emplace_back([&, g] { run_group(in, gpus[g], groups[g], cap); });
    if (!groups.empty() && !groups[0].empty()) run_group(in, gpus[0], groups[0], cap);
}

int solve_impl(int n, int planet, int asteroid, const double* qx, const double* qy, const double* qz, const double* vx,
               const double* vy, const double* vz, const double* m, const uint8_t* is_device, const int* devices,
               int n_devices, const nb_solve_options& opt, nb_answer* out) {
    if (n <= 0 || !qx || !qy || !qz || !vx || !vy || !vz || !m || !out) return NB_ERR_INVALID;
    if (opt.max_batch && (opt.max_batch < 2 || opt.max_batch > MAX_BATCH)) return set_error(NB_ERR_INVALID, "nb_solve_options.max_batch: 0 or 2..8");
    if (opt.engine < 0 || opt.engine > 2 || (opt.engine == 2 && n > SMALL_N_MAX)) return set_error(NB_ERR_INVALID, "nb_solve_options.engine: 0, 1, or 2 with n <= 128");
    if (opt.streams < 0 || opt.streams > 2 || opt.p3_parallel < 0) return set_error(NB_ERR_INVALID, "nb_solve_options.streams / p3_parallel");
    if (opt.graph_chunk && !valid_graph_chunk(opt.graph_chunk)) return set_error(NB_ERR_INVALID, "nb_solve_options.graph_chunk: 0 or even, 2..4000");
    if (opt.handoff < NB_HANDOFF_AUTO || opt.handoff > NB_HANDOFF_HOST_STAGED) return set_error(NB_ERR_INVALID, "nb_solve_options.handoff");
    if (planet < 0 || planet >= n || asteroid < 0 || asteroid >= n) return NB_ERR_INVALID;
    int ndev_gpu = 0;
    if (nb_device_count(&ndev_gpu) != NB_OK) return NB_ERR_NO_DEVICE;
    std::vector<int> gpus;
    if (devices && n_devices > 0) gpus.assign(devices, devices + n_devices);
    else gpus.push_back(0);
    for (int g : gpus)
        if (g < 0 || g >= ndev_gpu) return NB_ERR_NO_DEVICE;
    const size_t G = gpus.size();

    std::vector<int> dev_idx;
    std::vector<double> m_no_devices(m, m + n);
    for (int i = 0; i < n; ++i)
        if (is_device && is_device[i]) {
            dev_idx.push_back(i);
            m_no_devices[(size_t)i] = 0.0;
        }
    const size_t D = dev_idx.size();
    if (D > NB_MAX_WATCH) return NB_ERR_INVALID;
    const bool trace = trace_enabled();  // stderr timeline of the driver's phases
    static thread_local std::chrono::steady_clock::time_point t_start;
    t_start = std::chrono::steady_clock::now();
    auto stamp_fn = +[](const char* what) {
        fprintf(stderr, "[nb_solve] %8.1f ms  %s\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(), what);
    };
    auto stamp = [&](const char* what) {
        if (trace) stamp_fn(what);
    };
    const SolveInput in{n, planet, asteroid, qx, qy, qz, vx, vy, vz, m, is_device, &m_no_devices, trace ? stamp_fn : nullptr};
    stamp("HIP runtime up, input checked");

    const int cap = opt.max_batch ? opt.max_batch : MAX_BATCH;  // scenarios per launch stream; fewer = the queueing path

    const int n_steps = 200000;  // nbody.cc:10
    auto base_scn = [&](int kind) {
        nb_scenario s{};
        s.kind = kind;
        s.first_step = 0;
        s.last_step = n_steps;
        s.planet = planet;
        s.asteroid = asteroid;
        s.sync_every = 2000;    // hw5.cu:72
        s.planet_radius = 1e7;  // nbody.cc:17
        s.missile_speed = 1e6;  // nbody.cc:18
        return s;
    };

    // The scenarios of the program: P1 (devices massless, nbody.cc:109-122), P2 (nbody.cc:124-138) and one Problem-3 run per
    // gravity device (hw5.cu:289-309).  Two engines, chosen by system size below:
    //  * persistent (n <= 128): every scenario starts at step 0 in ONE launch with a workgroup per scenario — until its
    //    missile arrives a device's run IS the P2 trajectory, so nothing waits for P2's snapshots (hw5.cu:265-287,482-489)
    //    and the critical path of the whole program is one 200 000-step scenario;
    //  * per-step (n > 128): replayed graphs of launches, a stream per scenario, Problem-3 runs started from P2's arrival
    //    snapshots in arrival order.
    // Several GPUs each take a share of the scenarios (the reference's task parallelism, hw5.cu:564-567,587-588).
    std::vector<SolveSlot> slots(2 + D);
    slots[0].scn = base_scn(NB_SCN_MIN_DIST);
    slots[0].zero_devices = true;
    slots[1].scn = base_scn(NB_SCN_FIRST_HIT);
    slots[1].scn.n_watch = (int)D;  // arrival steps order the devices that do not fit the first wave (hw5.cu:574-585)
    for (size_t k = 0; k < D; ++k) slots[1].scn.watch[k] = dev_idx[k];
    slots[1].scn.flags = NB_SCN_NO_SNAPSHOT;
    for (size_t k = 0; k < D; ++k) {
        SolveSlot& s = slots[2 + k];
        s.scn = base_scn(NB_SCN_MISSILE);
        s.scn.n_watch = 1;
        s.scn.watch[0] = dev_idx[k];
        s.device_k = (int)k;
    }
    // nb_solve_options.engine overrides the choice by system size (tests run small systems through both)
    const bool per_step = opt.engine ? opt.engine == 1 : n > SMALL_N_MAX;
    if (per_step) {
        for (SolveSlot& sl : slots) sl.scn.engine = 1;
        // Per-step engine: one stream + replayed graph each for P1, for P2, and for the Problem-3 runs (at most two
        // streams of those per GPU: four hardware queues).  A Problem-3 run is dormant until P2's monitor reports the
        // missile's arrival at its device; it then starts from the snapshot P2 took at that step (hw5.cu:265-287,
        // 482-489) at most one replay (1000 steps) behind, so the whole program ends one replay after P1 does.
        // Independent streams keep the scenarios out of phase — one's latency-bound launch prologue overlaps another's
        // pair loop — which a lock-step batch of large systems cannot (profiles/r02_scenario_batch_timing.txt).
        CtxList list;
        std::vector<CtxPtr>& owned = list.v;
        auto make = [&](int gpu, bool zero_devices, nb_context** c, hipStream_t borrowed) -> int {
            nb_config cfg;
            nb_config_default(&cfg);
            cfg.n = n;
            cfg.device = gpu;
            int rc = create_context(c, &cfg, borrowed);
            owned.emplace_back(*c);
            if (rc) return set_error(rc, *c ? nb_last_error(*c) : "nb_create");
            rc = nb_set_state(*c, qx, qy, qz, vx, vy, vz, zero_devices ? m_no_devices.data() : m, is_device);
            if (rc) return set_error(rc, nb_last_error(*c));
            reset_monitor_host(*c);
            return NB_OK;
        };
        slots[1].scn.flags = 0;  // P2 keeps the arrival snapshots its followers start from
        // Streams: systems of a few hundred bodies leave most of the chip idle and their launch chain does not lengthen
        // when several scenarios share a launch (measured flat up to 5 at n = 200), so everything on a GPU goes into ONE
        // graph; from ~256 bodies on a lock-step batch pays for every member (n = 1024: 5.8 / 6.9 / 8.1 / 9.2 us per step
        // for 1 / 2 / 3 / 4 scenarios) and P1, P2 and the Problem-3 runs get a stream each instead
        // (profiles/r02_scenario_batch_timing.txt).  nb_solve_options.streams overrides.
        const bool merged = opt.streams ? opt.streams == 1 : n <= 256;
        std::vector<GraphGroup> groups(merged ? 3 * G : 2 + 2 * G);
        std::vector<nb_context*> cs(slots.size(), nullptr);
        // a launch stream's leader creates the stream (~8 ms each); the other scenarios of that stream borrow it
        if (int rc = make(gpus[0], true, &cs[0], nullptr)) return rc;
        if (int rc = make(gpus[1 % G], false, &cs[1], (merged && G == 1) ? cs[0]->stream : nullptr)) return rc;
        GraphGroup* p2_group = nullptr;
        int p2_slot = 0;
        for (size_t i = 0; i < 2; ++i) {
            GraphGroup& g = merged ? groups[3 * (i % G)] : groups[i];
            if (!g.lead) g.lead = cs[i];
            GraphSlot sl;
            sl.c = cs[i];
            sl.scn = &slots[i].scn;
            sl.gpu_slot = (int)(i % G);
            if (i == 1) { p2_group = &g; p2_slot = (int)g.slots.size(); }
            g.slots.push_back(sl);
        }
        for (size_t k = 0; k < D; ++k) {
            const size_t gi = (2 + k) % G;  // GPU of this device's run
            GraphGroup* g = nullptr;
            if (merged) {  // the GPU's shared graph while it has room (8 scenarios per launch), then two overflow graphs
                for (size_t j = 0; j < 3 && !g; ++j)
                    if (groups[3 * gi + j].slots.size() < (size_t)MAX_BATCH) g = &groups[3 * gi + j];
            } else {
                g = &groups[2 + 2 * gi + (k / G) % 2];  // one of the GPU's two follower streams
                if (g->slots.size() >= (size_t)MAX_BATCH) g = nullptr;
            }
            if (!g) return set_error(NB_ERR_INVALID, "too many gravity devices per stream");
            if (int rc = make(gpus[gi], false, &cs[2 + k], g->lead ? g->lead->stream : nullptr)) return rc;
            if (!g->lead) g->lead = cs[2 + k];
            GraphSlot f;
            f.c = cs[2 + k];
            f.scn = &slots[2 + k].scn;
            f.gpu_slot = (int)gi;
            f.active = false;
            f.parent = p2_group;
            f.parent_slot = p2_slot;
            f.parent_watch = (int)k;
            g->slots.push_back(f);
        }
        stamp("contexts created, state uploaded");
        std::vector<GraphGroup*> live;
        for (GraphGroup& g : groups)
            if (g.lead) {
                g.chunk = opt.graph_chunk;  // 0 = default
                live.push_back(&g);
            }
        // Problem-3 runs at a time: one per GPU, like the reference's one worker thread per GPU (hw5.cu:587-588); the
        // others wait their turn in arrival order.  nb_solve_options.p3_parallel overrides (e.g. 16 = all at once).
        FollowerPolicy policy;
        policy.parallel = opt.p3_parallel ? opt.p3_parallel : (int)G;
        policy.stage_through_host = opt.handoff == NB_HANDOFF_HOST_STAGED;
        if (int rc = run_groups_graph(live, policy)) return set_error(rc, nb_last_error(live[0]->lead));
        stamp(merged ? "graph-driven scenarios done (one stream per GPU)" : "graph-driven scenarios done (stream per scenario)");
        for (GraphGroup* g : live)
            for (GraphSlot& gs : g->slots) {
                SolveSlot& sl = slots[(size_t)(std::find(cs.begin(), cs.end(), gs.c) - cs.begin())];
                fill_result(gs.c, gs.scn, device_scenario(gs.c, gs.scn), gs.done_at, &sl.res);
                sl.ran = true;
            }
        out->min_dist = std::sqrt(slots[0].res.min_dist2);  // nbody.cc:121 takes min of sqrt; sqrt is monotone
        out->hit_time_step = slots[1].res.hit_step;
        out->gravity_device_id = -1;
        out->missile_cost = 0;
        if (slots[1].res.hit_step == -2) return NB_OK;  // no collision: nothing to prevent (hw5.cu:547-548,568)
        int best_arrival = std::numeric_limits<int>::max();
        for (size_t k = 0; k < D; ++k) {
            const nb_scenario_result& r = slots[2 + k].res;
            // feasible: the missile arrived (before the P2 hit, else the run never started) and no hit followed
            // (hw5.cu:512: strict <; cost is monotone in the arrival step)
            if (r.hit_step == -2 && r.arrival_step[0] != -2 && r.steps_done == n_steps && r.arrival_step[0] < best_arrival) {
                best_arrival = r.arrival_step[0];
                out->gravity_device_id = dev_idx[k];
                out->missile_cost = r.missile_cost[0];
            }
        }
        return NB_OK;
    }

    const size_t first_wave = std::min(slots.size(), G * (size_t)cap);
    {
        std::vector<std::vector<SolveSlot*>> groups(G);
        for (size_t i = 0; i < first_wave; ++i) groups[i % G].push_back(&slots[i]);
        run_groups(in, gpus, groups, cap);
    }
    stamp("first wave of scenarios done");
    if (slots[0].rc) return set_error(slots[0].rc, slots[0].err);
    if (slots[1].rc) return set_error(slots[1].rc, slots[1].err);
    out->min_dist = std::sqrt(slots[0].res.min_dist2);  // nbody.cc:121 takes min of sqrt; sqrt is monotone
    out->hit_time_step = slots[1].res.hit_step;
    out->gravity_device_id = -1;
    out->missile_cost = 0;
    if (slots[1].res.hit_step == -2) return NB_OK;  // no collision: nothing to prevent (hw5.cu:547-548,568)

    // Problem 3 (hw5.cu:568-602): answer = feasible device of least cost (strict <, hw5.cu:512); cost is monotone in
    // the arrival step, so a device can only improve on a feasible one by arriving earlier
    int best_arrival = std::numeric_limits<int>::max();
    int rc3 = NB_OK;
    auto account = [&](const SolveSlot& s) {
        if (s.rc) { rc3 = set_error(s.rc, s.err); return; }
        if (!s.ran) return;
        // feasible: the missile arrived (before the P2 hit, hence before any hit of this run) and no hit followed
        if (s.res.hit_step == -2 && s.res.arrival_step[0] != -2 && s.res.arrival_step[0] < best_arrival) {
            best_arrival = s.res.arrival_step[0];
            out->gravity_device_id = dev_idx[(size_t)s.device_k];
            out->missile_cost = s.res.missile_cost[0];
        }
    };
    for (size_t i = 2; i < first_wave; ++i) account(slots[i]);

    // devices beyond the first wave: cheapest first — ascending arrival step on the P2 trajectory (hw5.cu:574-585) —
    // and stop as soon as no remaining device can beat a feasible one (PROBLEM3_BREAK, hw5.cu:490-493)
    std::vector<size_t> rest;
    for (size_t i = first_wave; i < slots.size(); ++i)
        if (slots[1].res.arrival_step[slots[i].device_k] != -2) rest.push_back(i);  // never arrives before the hit: fails
    // (stable: devices with equal arrival steps stay in index order, so a tie resolves to the lowest device index here as
    // in the per-step engine's queue and in the strict < of account())
    std::stable_sort(rest.begin(), rest.end(), [&](size_t a, size_t b) {
        return slots[1].res.arrival_step[slots[a].device_k] < slots[1].res.arrival_step[slots[b].device_k];
    });
    size_t at = 0;
    while (at < rest.size() && !rc3) {
        std::vector<std::vector<SolveSlot*>> groups(G);
        size_t taken = 0;
        for (; at < rest.size() && taken < G * (size_t)cap; ++at) {
            if (slots[1].res.arrival_step[slots[rest[at]].device_k] >= best_arrival) { at = rest.size(); break; }
            groups[taken % G].push_back(&slots[rest[at]]);
            ++taken;
        }
        if (!taken) break;
        run_groups(in, gpus, groups, cap);
        for (auto& g : groups)
            for (SolveSlot* s : g) account(*s);
    }
    return rc3;
}

}  // namespace

extern "C" {

int nb_solve_ex(int n, int planet, int asteroid, const double* qx, const double* qy, const double* qz,
                const double* vx, const double* vy, const double* vz, const double* m, const uint8_t* is_device,
                const int* devices, int n_devices, const nb_solve_options* options, nb_answer* out) {
    try {
        const nb_solve_options defaults{};
        return solve_impl(n, planet, asteroid, qx, qy, qz, vx, vy, vz, m, is_device, devices, n_devices,
                          options ? *options : defaults, out);
    } catch (...) {  // bad_alloc / system_error from the host-side containers and threads
        return NB_ERR_NOMEM;
    }
}

int nb_solve(int n, int planet, int asteroid, const double* qx, const double* qy, const double* qz,
             const double* vx, const double* vy, const double* vz, const double* m, const uint8_t* is_device,
             const int* devices, int n_devices, nb_answer* out) {
    return nb_solve_ex(n, planet, asteroid, qx, qy, qz, vx, vy, vz, m, is_device, devices, n_devices, nullptr, out);
}

}  // extern "C"
