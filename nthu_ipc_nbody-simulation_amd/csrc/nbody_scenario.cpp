// nbody_scenario.cpp — the scenario drivers of the C ABI: the loops the reference's main() runs around run_step
//   nb_run_scenario / nb_run_scenarios_batched  <- P1/P2 loops, t_problem_12/_3   samples/nbody.cc:114-138 ; hw5.cu:366-404,489-508
// Engines: K3 (whole step loop in one persistent launch, n <= 128), eager per-step launches, and the graph-driven
// per-step engine with its Problem-3 follower queue (hw5.cu:490-493,574-596) that nb_solve (nbody_solve.cpp) drives.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <functional>
#include <future>
#include <limits>
#include <new>
#include <vector>

#include "nbody_internal.h"

using namespace nbk;
using namespace nbi;

namespace {

int check_scenario(const nb_context* c, const nb_scenario* s) {
    if (c->cfg.precision != NB_F64) return NB_ERR_INVALID;
    if (!c->have_state) return NB_ERR_STATE;
    if (s->kind < NB_SCN_MIN_DIST || s->kind > NB_SCN_MISSILE) return NB_ERR_INVALID;
    if (s->n_watch < 0 || s->n_watch > NB_MAX_WATCH) return NB_ERR_INVALID;
    // one device is destroyed per Problem-3 run (hw5.cu:289-309): the kernels keep a single dead body
    if (s->kind == NB_SCN_MISSILE && s->n_watch > 1) return NB_ERR_INVALID;
    if (s->planet < 0 || s->planet >= c->n || s->asteroid < 0 || s->asteroid >= c->n) return NB_ERR_INVALID;
    if (s->last_step < s->first_step) return NB_ERR_INVALID;
    for (int k = 0; k < s->n_watch; ++k)
        if (s->watch[k] < 0 || s->watch[k] >= c->n) return NB_ERR_INVALID;
    if (s->engine < 0 || s->engine > 2) return NB_ERR_INVALID;
    if (s->engine == 2 && c->n > SMALL_N_MAX) return NB_ERR_INVALID;
    if (s->graph_chunk && !valid_graph_chunk(s->graph_chunk)) return NB_ERR_INVALID;
    return NB_OK;
}

}  // namespace

F64Scenario nbi::device_scenario(const nb_context* c, const nb_scenario* s) {
    F64Scenario sc{};
    sc.kind = s->kind;
    sc.planet = s->planet;
    sc.asteroid = s->asteroid;
    sc.n_watch = (s->kind == NB_SCN_MIN_DIST) ? 0 : s->n_watch;
    for (int k = 0; k < sc.n_watch; ++k) sc.watch[k] = s->watch[k];
    sc.destroy_on_arrival = (s->kind == NB_SCN_MISSILE);
    sc.R2 = s->planet_radius * s->planet_radius;      // nbody.cc:134
    sc.missile_dstep = s->missile_speed * c->cfg.dt;  // hw5.cu:274
    return sc;
}

namespace {

bool wants_snapshots(const nb_scenario* s) {
    return s->kind == NB_SCN_FIRST_HIT && s->n_watch > 0 && !(s->flags & NB_SCN_NO_SNAPSHOT);
}

int ensure_snapshots(nb_context* c, int n_watch) {
    if (c->snap_slots >= n_watch) return NB_OK;
    const size_t n = (size_t)c->n;
    free_dev(c->snap_q);
    free_dev(c->snap_v);
    c->snap_slots = 0;
    NB_HIP(c, hipMalloc(&c->snap_q, (size_t)n_watch * 3 * n * sizeof(double)));
    NB_HIP(c, hipMalloc(&c->snap_v, (size_t)n_watch * 3 * n * sizeof(double)));
    c->snap_slots = n_watch;
    return NB_OK;
}

// K3 reads |sin(step*dt/6000)| by step index from a host-computed (glibc) table and prefetches two steps ahead
int ensure_fst_table(nb_context* c, int last_step) {
    const int need = last_step + 3;
    if (c->fst_len < need) {
        free_dev(c->fst_dev);
        c->fst_len = 0;
        std::vector<double> tab((size_t)need);
        for (int k = 0; k < need; ++k) tab[(size_t)k] = fst_of(k, c->cfg.dt);
        NB_HIP(c, hipMalloc(&c->fst_dev, (size_t)need * sizeof(double)));
        // (never the legacy stream: another host thread may be capturing a graph on its own context's stream)
        NB_HIP(c, hipMemcpyAsync(c->fst_dev, tab.data(), (size_t)need * sizeof(double), hipMemcpyHostToDevice, c->stream));
        NB_HIP(c, hipStreamSynchronize(c->stream));
        c->fst_len = need;
    }
    return NB_OK;
}

// graph-driven stepping: the per-scenario array a replay's launches read their |sin| from (see F64Args.fst_chunk)
int ensure_fst_chunk(nb_context* err, nb_context* c, int chunk) {
    const int need = chunk + 2;
    if (c->fst_chunk_len < need) {
        free_dev(c->fst_chunk);
        c->fst_chunk_len = 0;
        NB_HIP(err, hipMalloc(&c->fst_chunk, (size_t)need * sizeof(double)));
        c->fst_chunk_len = need;
    }
    return NB_OK;
}

// K3 reports the index of the last state it computed through a device word + its pinned host copy (part of the arenas)
int ensure_done_word(nb_context* c) { return (c->done_dev && c->done_host) ? NB_OK : NB_ERR_STATE; }

}  // namespace

void nbi::reset_monitor_host(nb_context* c) {
    F64Monitor* mh = c->mon_host;
    mh->min_d2 = std::numeric_limits<double>::infinity();
    mh->hit_step = -2;
    for (int k = 0; k < MAX_WATCH; ++k) mh->arrival_step[k] = -2;
    for (int k = 0; k < NB_MAX_WATCH; ++k) c->snap_arrival[k] = -2;
}

namespace {

// `err` = the context that reports a HIP failure (the batch leader when several contexts share a stream)
int reset_monitor(nb_context* err, nb_context* c, hipStream_t stream) {
    F64Monitor* mh = c->mon_host;
    reset_monitor_host(c);
    NB_HIP(err, hipMemcpyAsync(c->mon, mh, sizeof(F64Monitor), hipMemcpyHostToDevice, stream));
    return NB_OK;
}

}  // namespace

void nbi::fill_result(nb_context* c, const nb_scenario* s, const F64Scenario& sc, int steps_done, nb_scenario_result* res) {
    const F64Monitor* mh = c->mon_host;
    memset(res, 0, sizeof *res);
    res->min_dist2 = mh->min_d2;
    res->hit_step = mh->hit_step;
    res->steps_done = steps_done;
    for (int k = 0; k < NB_MAX_WATCH; ++k) {
        res->arrival_step[k] = (k < sc.n_watch) ? mh->arrival_step[k] : -2;
        res->missile_cost[k] = (res->arrival_step[k] != -2)
                                   ? 1e5 + 1e3 * ((res->arrival_step[k] + 1) * c->cfg.dt)  // hw5.cu:305 ; nbody.cc:19
                                   : 0.0;
        if (wants_snapshots(s)) c->snap_arrival[k] = res->arrival_step[k];  // which snapshot slots hold a state
    }
}

namespace {

F64SmallArgs small_args(nb_context* c, const F64Scenario& sc, bool want_snap, const double* fst_table, int at, int to,
                        int last_step) {
    F64SmallArgs k{};
    k.q = c->q[c->cur];
    k.v = c->v;
    k.m = c->m;
    k.coef = c->coef;
    k.fst = fst_table;
    k.snap_q = want_snap ? c->snap_q : nullptr;
    k.snap_v = want_snap ? c->snap_v : nullptr;
    k.mon = c->mon;
    k.steps_done = c->done_dev;
    k.n = c->n;
    k.first_step = at;
    k.last_step = to;
    k.final_monitor = (to == last_step);
    k.G = c->cfg.G;
    k.eps2 = c->cfg.eps * c->cfg.eps;
    k.dt = c->cfg.dt;
    k.scn = sc;
    return k;
}

constexpr int SMALL_CHUNK = 50000;  // K3: steps per launch, so that the host can stop relaunching after a hit
constexpr int GRAPH_CHUNK_DEFAULT = 1000;  // K2, graph-driven: steps per replay (even: the ping-pong buffers are back
                                           // in place after a chunk); nb_scenario.graph_chunk / nb_solve_options.graph_chunk
                                           // override it per run (shorter graphs for tracing tools)
}  // namespace
bool nbi::valid_graph_chunk(int chunk) { return chunk >= 2 && chunk <= 4000 && chunk % 2 == 0; }
namespace {
constexpr int GRAPH_MIN_STEPS = 4000;  // shorter ranges are launched eagerly: capture + instantiate would cost more

int run_batched_impl(nb_context** ctxs, const nb_scenario* scns, nb_scenario_result* results, int count);

int run_scenario_impl(nb_context* c, const nb_scenario* s, nb_scenario_result* res) {
    if (int rc = check_scenario(c, s)) return rc;
    if (int rc = bind(c)) return rc;
    const F64Scenario sc = device_scenario(c, s);
    const bool want_snap = wants_snapshots(s);
    if (want_snap)
        if (int rc = ensure_snapshots(c, sc.n_watch)) return rc;
    if (int rc = reset_monitor(c, c, c->stream)) return rc;
    NB_HIP(c, hipStreamSynchronize(c->stream));
    F64Monitor* mh = c->mon_host;

    const bool small_engine = (s->engine == 2) || (s->engine == 0 && c->n <= SMALL_N_MAX);
    if (small_engine) {
        // K3: the whole step loop inside one single-workgroup kernel, in chunks so the host can stop after a hit
        if (int rc = ensure_fst_table(c, s->last_step)) return rc;
        if (int rc = ensure_done_word(c)) return rc;
        int at = s->first_step;
        bool first = true;
        while (first || at < s->last_step) {
            first = false;
            const F64SmallArgs k = small_args(c, sc, want_snap, c->fst_dev, at, std::min(s->last_step, at + SMALL_CHUNK),
                                              s->last_step);
            NB_HIP(c, (hipError_t)launch_f64_small(k, c->stream));
            NB_HIP(c, hipMemcpyAsync(mh, c->mon, sizeof(F64Monitor), hipMemcpyDeviceToHost, c->stream));
            NB_HIP(c, hipMemcpyAsync(c->done_host, c->done_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            NB_HIP(c, hipStreamSynchronize(c->stream));
            at = *c->done_host;
            if (mh->hit_step != -2 || at < k.last_step) break;
        }
        fill_result(c, s, sc, at, res);
        return NB_OK;
    }

    if (!(s->flags & NB_SCN_EAGER) && s->last_step - s->first_step >= GRAPH_MIN_STEPS)
        return run_batched_impl(&c, s, res, 1);  // graph replay of the (batched) step kernel with one slot

    const int sync_every = s->sync_every > 0 ? s->sync_every : 2000;  // hw5.cu:72
    const bool can_stop = s->kind != NB_SCN_MIN_DIST;
    bool stopped = false;
    int step = s->first_step + 1;
    for (; step <= s->last_step; ++step) {
        F64Args a = base_args(c, step);
        a.scn = sc;
        a.snap_q = want_snap ? c->snap_q : nullptr;
        a.snap_v = want_snap ? c->snap_v : nullptr;
        NB_HIP(c, (hipError_t)launch_f64(a, c->split, c->stream));
        c->cur ^= 1;
        if (can_stop && (step % sync_every == sync_every - 1)) {  // hw5.cu:398-402
            NB_HIP(c, hipMemcpyAsync(&mh->hit_step, &c->mon->hit_step, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            NB_HIP(c, hipStreamSynchronize(c->stream));
            if (mh->hit_step != -2) {
                stopped = true;
                break;
            }
        }
    }
    if (!stopped) {  // monitor of the final state (index last_step): nbody.cc's loop runs step <= n_steps
        F64Args a = base_args(c, s->last_step + 1);
        a.scn = sc;
        a.do_update = 0;
        a.snap_q = want_snap ? c->snap_q : nullptr;
        a.snap_v = want_snap ? c->snap_v : nullptr;
        NB_HIP(c, (hipError_t)launch_f64(a, c->split, c->stream));
    }
    NB_HIP(c, hipMemcpyAsync(mh, c->mon, sizeof(F64Monitor), hipMemcpyDeviceToHost, c->stream));
    NB_HIP(c, hipStreamSynchronize(c->stream));
    // a hit ends the scenario at the state it was seen in, whichever poll noticed it (the launches after it returned at once)
    (void)stopped;
    fill_result(c, s, sc, (can_stop && mh->hit_step != -2) ? mh->hit_step : s->last_step, res);
    return NB_OK;
}

// ---------------------------------------------------------------- graph-driven stepping (the per-step engine, K2)
// One eager launch costs the HOST 3.1-3.7 us on this platform (bench/ubench/launch_rate.hip, profiles/r02_launch_rate.txt)
// — more than a step of a few-hundred-body system takes on the GPU — while a hipGraph of kernel nodes replays at
// 1.5-2.0 us per node with no host work at all.  A captured launch cannot carry its step index, so the scenario keeps a
// control word {base step, active} in HBM: the node with offset t computes step base + t, reads |sin| from the
// host-computed table, runs the monitor-only launch at last_step + 1 and returns at once beyond it (or while the slot is
// dormant); a one-thread node at the end of the graph advances base by the chunk length.  The host replays the graph,
// copies the monitors back and looks at them once per chunk (where hw5.cu polls every 2000 steps, hw5.cu:398-402).

}  // namespace

nbi::GraphGroup::~GraphGroup() {
    if (lead) (void)hipSetDevice(lead->cfg.device);
    if (lead && lead->stream) (void)hipStreamSynchronize(lead->stream);  // error paths: nothing of ours in flight
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    for (hipEvent_t e : ev)
        if (e) (void)hipEventDestroy(e);
}
bool nbi::GraphGroup::running() const {
    for (const GraphSlot& s : slots)
        if (s.done_at < 0) return true;
    return false;
}
bool nbi::GraphGroup::anything_active() const {
    for (const GraphSlot& s : slots)
        if (s.done_at < 0 && s.active) return true;
    return false;
}

namespace {

int upload_ctl(nb_context* err, GraphSlot& s, hipStream_t stream) {
    *s.c->ctl_host = F64Ctl{s.base, s.active ? 1 : 0};  // pinned; rewritten only with the same values while in flight
    NB_HIP(err, hipMemcpyAsync(s.c->ctl, s.c->ctl_host, sizeof(F64Ctl), hipMemcpyHostToDevice, stream));
    return NB_OK;
}

// monitors, control words, tables, and the captured graph of g.chunk batched launches + the advance node
int group_prepare(GraphGroup& g) {
    nb_context* c0 = g.lead;
    if (int rc = bind(c0)) return rc;
    hipStream_t stream = c0->stream;
    int max_last = 0;
    for (GraphSlot& s : g.slots) {
        s.sc = device_scenario(s.c, s.scn);
        s.snap = wants_snapshots(s.scn);
        if (s.snap)
            if (int rc = ensure_snapshots(s.c, s.sc.n_watch)) { snprintf(c0->err, sizeof c0->err, "%s", s.c->err); return rc; }
        max_last = std::max(max_last, s.scn->last_step);
        NB_HIP(c0, hipStreamSynchronize(s.c->stream));  // earlier work of this context (uploads) is complete
        if (int rc = reset_monitor(c0, s.c, stream)) return rc;
        if (int rc = upload_ctl(c0, s, stream)) return rc;
    }
    if (int rc = ensure_fst_table(c0, max_last)) return rc;  // indices up to last_step + 1 are read
    if (!g.chunk) g.chunk = GRAPH_CHUNK_DEFAULT;
    const int count = (int)g.slots.size();
    g.cb = F64CtlBatch{};
    g.cb.count = count;
    for (int b = 0; b < count; ++b) {
        nb_context* c = g.slots[(size_t)b].c;
        if (int rc = ensure_fst_chunk(c0, c, g.chunk)) return rc;
        g.cb.ctl[b] = c->ctl;
        g.cb.fst_chunk[b] = c->fst_chunk;
    }
    // the first replay's |sin| values, from the control words uploaded above (stream order)
    NB_HIP(c0, (hipError_t)launch_fst_fill(g.cb, 0, g.chunk, c0->fst_dev, c0->fst_len, stream));
    NB_HIP(c0, hipStreamSynchronize(stream));

    const auto t_prep = std::chrono::steady_clock::now();
    // relaxed mode: the capture restricts neither this thread's nor other host threads' HIP calls on OTHER streams
    // (distinct contexts may be driven from distinct threads); nothing but the launches below touches `stream` meanwhile
    NB_HIP(c0, hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed));
    hipError_t bad = hipSuccess;
    const int chunk = g.chunk;
    for (int t = 0; t < chunk && bad == hipSuccess; ++t) {
        F64BatchArgs args{};
        args.count = count;
        for (int b = 0; b < count; ++b) {
            GraphSlot& s = g.slots[(size_t)b];
            nb_context* c = s.c;
            F64Args a{};
            a.qin = c->q[c->cur ^ (t & 1)];
            a.qout = c->q[c->cur ^ (t & 1) ^ 1];
            a.v = c->v;
            a.m = c->m;
            a.coef = c->coef;
            a.snap_q = s.snap ? c->snap_q : nullptr;
            a.snap_v = s.snap ? c->snap_v : nullptr;
            a.mon = c->mon;
            a.n = c->n;
            a.do_update = 1;  // (full grid; the kernel decides from the control word)
            a.G = c->cfg.G;
            a.eps2 = c->cfg.eps * c->cfg.eps;
            a.dt = c->cfg.dt;
            a.scn = s.sc;
            a.ctl = c->ctl;
            a.fst_chunk = c->fst_chunk;
            a.t = t + 1;  // state index base + t  ->  step base + t + 1
#if NB_STEP_STAMPS
            a.stamps = c->stamps;  // measurement hook of the instrumented build, null unless nb_enable_step_stamps
            a.stamp_slots = c->stamp_slots;
#endif
            a.last_step = s.scn->last_step;
            args.item[b] = a;
        }
        bad = (hipError_t)launch_f64_batched(args, c0->n, c0->split, stream);
    }
    // end of a replay: the next replay's |sin| values (read against the OLD base step), then the base steps move on
    if (bad == hipSuccess) bad = (hipError_t)launch_fst_fill(g.cb, chunk, chunk, c0->fst_dev, c0->fst_len, stream);
    if (bad == hipSuccess) bad = (hipError_t)launch_ctl_advance(g.cb, chunk, stream);
    hipError_t e = hipStreamEndCapture(stream, &g.graph);
    if (bad != hipSuccess) return fail_hip(c0, bad, "capturing the step graph");
    if (e != hipSuccess) return fail_hip(c0, e, "hipStreamEndCapture");
    const auto t_cap = std::chrono::steady_clock::now();
    NB_HIP(c0, hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
    for (hipEvent_t& e : g.ev) NB_HIP(c0, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (trace_enabled())
        fprintf(stderr, "[graph] %d slots x %d launches: capture %.2f ms, instantiate %.2f ms\n", count, chunk,
                std::chrono::duration<double, std::milli>(t_cap - t_prep).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_cap).count());
    g.prepared = true;
    return NB_OK;
}

// one replay = g.chunk steps of every active slot, then the monitors travel to their pinned host copies
int group_launch(GraphGroup& g) {
    nb_context* c0 = g.lead;
    if (int rc = bind(c0)) return rc;
    if (!g.prepared)
        if (int rc = group_prepare(g)) return rc;
    NB_HIP(c0, hipGraphLaunch(g.exec, c0->stream));
    for (GraphSlot& s : g.slots)
        if (s.done_at < 0 && s.active) {
            NB_HIP(c0, hipMemcpyAsync(s.c->mon_host, s.c->mon, sizeof(F64Monitor), hipMemcpyDeviceToHost, c0->stream));
            ++s.inflight;
        }
    NB_HIP(c0, hipEventRecord(g.ev[g.launched & 1], c0->stream));
    ++g.launched;
    return NB_OK;
}

// the oldest replay in flight has finished: host mirrors of the control words, and which slots have ended.  The pinned
// monitors may already hold what a LATER replay wrote — they only ever move forward (min, first hit, first arrival), and
// a value that is visible here was written by a replay that is complete (the copy is stream-ordered behind it).
int group_collect(GraphGroup& g) {
    nb_context* c0 = g.lead;
    if (int rc = bind(c0)) return rc;
    NB_HIP(c0, hipEventSynchronize(g.ev[g.collected & 1]));
    ++g.collected;
    for (GraphSlot& s : g.slots) {
        if (s.inflight <= 0) continue;
        --s.inflight;
        if (s.done_at >= 0) continue;  // ended at an earlier replay: the launches of this one returned at once
        const int before = s.base, last = s.scn->last_step;
        s.base += g.chunk;  // what nbody_ctl_advance did
        const int hit = load_word(&s.c->mon_host->hit_step);
        if (s.scn->kind != NB_SCN_MIN_DIST && hit != -2) {
            s.done_at = hit;  // the launch after state `hit` saw it and every later one returned at once
        } else if (s.base > last) {  // steps before+1 .. last were taken, and the monitor-only launch at last + 1 has run
            s.done_at = last;
            s.c->cur ^= (last - before) & 1;  // an odd number of updates leaves the state in the other buffer
        }
    }
    return NB_OK;
}

// start one dormant follower from its parent's arrival snapshot (hw5.cu:482-489)
int activate_follower(GraphGroup& g, GraphSlot& f, int arr, const FollowerPolicy& policy) {
    nb_context* c0 = g.lead;
    GraphSlot& p = f.parent->slots[(size_t)f.parent_slot];
    const size_t n = (size_t)f.c->n, B = 3 * n * sizeof(double);
    const double* sq = p.c->snap_q + (size_t)f.parent_watch * 3 * n;
    const double* sv = p.c->snap_v + (size_t)f.parent_watch * 3 * n;
    const bool device_copy = p.c->cfg.device == f.c->cfg.device && !(policy.stage_through_host && p.gpu_slot != f.gpu_slot);
    if (trace_enabled())
        fprintf(stderr, "[followers] device body %d starts at step %d from P2's snapshot: %s (device slot %d -> %d)\n",
                f.scn->watch[0], arr, device_copy ? "device copy" : "host-staged", p.gpu_slot, f.gpu_slot);
    if (device_copy) {  // the parent's replay that took the snapshot is complete (see group_collect)
        if (int rc = bind(c0)) return rc;
        NB_HIP(c0, hipMemcpyAsync(f.c->q[f.c->cur], sq, B, hipMemcpyDeviceToDevice, c0->stream));
        NB_HIP(c0, hipMemcpyAsync(f.c->v, sv, B, hipMemcpyDeviceToDevice, c0->stream));
    } else {  // another GPU (hw5.cu:482-484 uploads P2's snapshot on the other device): through the host.  The copies
              // travel on the streams the two contexts enqueue on — the parent's group stream, which is past the replay
              // that took the snapshot (see group_collect), and this group's stream, which the next replay follows.
        std::vector<double> hq(3 * n), hv(3 * n);
        if (int rc = bind(p.c)) return rc;
        NB_HIP(c0, hipMemcpyAsync(hq.data(), sq, B, hipMemcpyDeviceToHost, p.c->stream));
        NB_HIP(c0, hipMemcpyAsync(hv.data(), sv, B, hipMemcpyDeviceToHost, p.c->stream));
        NB_HIP(c0, hipStreamSynchronize(p.c->stream));
        if (int rc = bind(c0)) return rc;
        NB_HIP(c0, hipMemcpyAsync(f.c->q[f.c->cur], hq.data(), B, hipMemcpyHostToDevice, c0->stream));
        NB_HIP(c0, hipMemcpyAsync(f.c->v, hv.data(), B, hipMemcpyHostToDevice, c0->stream));
        NB_HIP(c0, hipStreamSynchronize(c0->stream));  // host staging buffers die here
    }
    f.base = arr;
    f.active = true;
    if (int rc = bind(c0)) return rc;
    if (int rc = upload_ctl(c0, f, c0->stream)) return rc;
    // its |sin| values for the next replay start at the arrival step (a group not yet prepared fills all slots when it is)
    if (g.prepared) NB_HIP(c0, (hipError_t)launch_fst_fill(g.cb, 0, g.chunk, c0->fst_dev, c0->fst_len, c0->stream));
    return NB_OK;
}

// The Problem-3 work queue (hw5.cu:490-493,574-596) over the followers of all groups: candidates are the devices whose
// missile has arrived on the parent (P2) trajectory, cheapest first = ascending arrival step; at most `parallel` of them
// run at a time (the reference: one per GPU); a run that ends feasible cancels every candidate that arrived later —
// it cannot cost less (PROBLEM3_BREAK) — and a run that ends in a hit hands its place to the next candidate.
int schedule_followers(std::vector<GraphGroup*>& groups, const FollowerPolicy& policy) {
    const int parallel = policy.parallel;
    struct Cand { GraphGroup* g; GraphSlot* f; int arr; };
    std::vector<Cand> waiting;
    int active = 0, best = std::numeric_limits<int>::max();
    for (GraphGroup* g : groups)
        for (GraphSlot& f : g->slots) {
            if (!f.parent) continue;
            GraphSlot& p = f.parent->slots[(size_t)f.parent_slot];
            const int arr = load_word(&p.c->mon_host->arrival_step[f.parent_watch]);
            if (f.done_at >= 0) {
                if (f.active && !f.cancelled && f.done_at == f.scn->last_step && load_word(&f.c->mon_host->hit_step) == -2)
                    best = std::min(best, arr);  // ended feasible
            } else if (f.active) {
                ++active;
            } else if (arr != -2) {
                waiting.push_back(Cand{g, &f, arr});
            } else if (p.done_at >= 0) {
                f.done_at = f.scn->first_step;  // the parent ended before this missile arrived: never starts
            }
        }
    for (GraphGroup* g : groups)  // nothing that arrived after a feasible device can beat it
        for (GraphSlot& f : g->slots) {
            if (!f.parent || f.done_at >= 0) continue;
            const int arr = load_word(&f.parent->slots[(size_t)f.parent_slot].c->mon_host->arrival_step[f.parent_watch]);
            if (arr != -2 && arr > best) {
                f.cancelled = true;
                f.done_at = f.active ? std::min(f.base, f.scn->last_step) : f.scn->first_step;
                if (f.active) {  // stop it on the device too: its launches in later replays return at once (as for a dormant slot)
                    --active;
                    f.active = false;
                    if (int rc = bind(g->lead)) return rc;
                    if (int rc = upload_ctl(g->lead, f, g->lead->stream)) return rc;
                }
            }
        }
    std::stable_sort(waiting.begin(), waiting.end(), [](const Cand& a, const Cand& b) { return a.arr < b.arr; });
    for (const Cand& c : waiting) {
        if (c.f->done_at >= 0) continue;  // cancelled above
        if (active >= parallel) break;
        if (int rc = activate_follower(*c.g, *c.f, c.arr, policy)) return rc;
        ++active;
    }
    return NB_OK;
}

// all groups to completion, one host thread: every running group keeps up to two replays in flight (the second is
// enqueued while the first executes, so neither the host's enqueue work nor its look at the monitors idles the GPU)
}  // namespace

int nbi::run_groups_graph(std::vector<GraphGroup*>& groups, const FollowerPolicy& policy) {
    for (;;) {
        bool progressed = false;
        // enqueueing a replay costs the host about a millisecond per 1000 nodes: groups enqueue side by side (graphs are
        // captured serially, on the calling thread, the first time round)
        std::vector<GraphGroup*> due;
        for (GraphGroup* g : groups)
            if (g->running() && g->anything_active() && g->launched - g->collected < 2) {
                if (!g->prepared)
                    if (int rc = group_prepare(*g)) return rc;
                due.push_back(g);
            }
        if (!due.empty()) {
            std::vector<std::future<int>> side;
            for (size_t k = 1; k < due.size(); ++k) side.push_back(std::async(std::launch::async, group_launch, std::ref(*due[k])));
            int rc = group_launch(*due[0]);
            for (auto& f : side) {
                const int r = f.get();
                if (!rc) rc = r;
            }
            if (rc) return rc;
            progressed = true;
        }
        for (GraphGroup* g : groups)
            if (g->launched > g->collected && (g->launched - g->collected == 2 || !g->running() || !g->anything_active() ||
                                               !progressed)) {
                if (int rc = group_collect(*g)) return rc;
                progressed = true;
            }
        bool dormant_left = false, inflight = false;
        if (int rc = schedule_followers(groups, policy)) return rc;
        for (GraphGroup* g : groups) {
            for (const GraphSlot& s : g->slots) dormant_left |= (s.done_at < 0);
            inflight |= g->launched > g->collected;
        }
        if (!dormant_left && !inflight) break;
        if (!progressed && !inflight) {  // only dormant followers whose parents have all ended: cannot wake any more
            for (GraphGroup* g : groups)
                for (GraphSlot& s : g->slots)
                    if (s.done_at < 0) s.done_at = s.scn->first_step;
            break;
        }
    }
    return NB_OK;
}

namespace {

// Several scenarios of equally sized systems on one GPU, all driven by ONE stream (that of ctxs[0]):
//  * small systems (K3): one launch whose workgroup k runs scenario k to its end, relaunched per SMALL_CHUNK steps
//    for the slots still running;
//  * otherwise (K2): lock step, ONE launch per step serves all of them (blockIdx.y), each with its own state, step
//    index, |sin| and monitor — what hw5.cu does with one host thread + launch stream per scenario
//    (hw5.cu:564-567,587-588), without the streams contending for the command processor.
int run_batched_impl(nb_context** ctxs, const nb_scenario* scns, nb_scenario_result* results, int count) {
    if (!ctxs || !scns || !results || count <= 0 || count > MAX_BATCH) return NB_ERR_INVALID;
    nb_context* c0 = ctxs[0];
    if (!c0) return NB_ERR_INVALID;
    for (int b = 0; b < count; ++b) {
        nb_context* c = ctxs[b];
        if (!c || c->n != c0->n || c->cfg.device != c0->cfg.device) return NB_ERR_INVALID;
        if (c->cfg.dt != c0->cfg.dt || scns[b].engine != scns[0].engine) return NB_ERR_INVALID;
        if (int rc = check_scenario(c, &scns[b])) return rc;
        for (int b2 = 0; b2 < b; ++b2)
            if (ctxs[b2] == c) return NB_ERR_INVALID;
    }
    if (int rc = bind(c0)) return rc;
    hipStream_t stream = c0->stream;

    F64Scenario sc[MAX_BATCH];
    bool snap[MAX_BATCH];
    int done_at[MAX_BATCH];  // -1 while running; else the index of the last state computed
    for (int b = 0; b < count; ++b) {
        nb_context* c = ctxs[b];
        sc[b] = device_scenario(c, &scns[b]);
        snap[b] = wants_snapshots(&scns[b]);
        if (snap[b])
            if (int rc = ensure_snapshots(c, sc[b].n_watch)) { snprintf(c0->err, sizeof c0->err, "%s", c->err); return rc; }
        done_at[b] = -1;
        NB_HIP(c0, hipStreamSynchronize(c->stream));  // earlier work of this context (uploads) is complete
        if (int rc = reset_monitor(c0, c, stream)) return rc;
    }
    NB_HIP(c0, hipStreamSynchronize(stream));

    const bool small_engine = (scns[0].engine == 2) || (scns[0].engine == 0 && c0->n <= SMALL_N_MAX);
    if (small_engine) {
        int max_last = 0, at[MAX_BATCH];
        for (int b = 0; b < count; ++b) {
            max_last = std::max(max_last, scns[b].last_step);
            at[b] = scns[b].first_step;
            if (int rc = ensure_done_word(ctxs[b])) { snprintf(c0->err, sizeof c0->err, "%s", ctxs[b]->err); return rc; }
        }
        const bool trace = trace_enabled();
        const auto t_k3 = std::chrono::steady_clock::now();
        auto lap = [&](const char* what, int a0, int a1) {
            if (trace)
                fprintf(stderr, "[K3 batch] %8.1f ms  %s %d %d\n",
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_k3).count(), what, a0, a1);
        };
        if (int rc = ensure_fst_table(c0, max_last)) return rc;  // same dt everywhere: one table serves the batch
        lap("|sin| table ready, entries", max_last + 3, 0);
        int running = count;
        while (running > 0) {
            F64SmallBatchArgs args{};
            args.count = count;
            int to[MAX_BATCH];
            for (int b = 0; b < count; ++b) {
                if (done_at[b] >= 0) continue;  // finished slot: item[b].n stays 0
                // each workgroup runs to the end of its own scenario (it stops by itself at a hit): chunking the launch would
                // make every scenario wait for the slowest one of the batch at each chunk boundary
                to[b] = scns[b].last_step;
                args.item[b] = small_args(ctxs[b], sc[b], snap[b], c0->fst_dev, at[b], to[b], scns[b].last_step);
            }
            NB_HIP(c0, (hipError_t)launch_f64_small_batched(args, c0->n, stream));
            for (int b = 0; b < count; ++b) {
                if (done_at[b] >= 0) continue;
                NB_HIP(c0, hipMemcpyAsync(ctxs[b]->mon_host, ctxs[b]->mon, sizeof(F64Monitor), hipMemcpyDeviceToHost, stream));
                NB_HIP(c0, hipMemcpyAsync(ctxs[b]->done_host, ctxs[b]->done_dev, sizeof(int), hipMemcpyDeviceToHost, stream));
            }
            NB_HIP(c0, hipStreamSynchronize(stream));
            lap("launch returned, slots still running before it", running, 0);
            for (int b = 0; b < count; ++b) {
                if (done_at[b] >= 0) continue;
                at[b] = *ctxs[b]->done_host;
                lap("  slot reached state", b, at[b]);
                if (ctxs[b]->mon_host->hit_step != -2 || at[b] < to[b] || at[b] >= scns[b].last_step) {
                    done_at[b] = at[b];
                    --running;
                }
            }
        }
        for (int b = 0; b < count; ++b) fill_result(ctxs[b], &scns[b], sc[b], done_at[b], &results[b]);
        return NB_OK;
    }

    // long runs: replay a captured graph of launches instead of issuing every launch from the host
    int longest = 0;
    bool eager = false;
    for (int b = 0; b < count; ++b) {
        longest = std::max(longest, scns[b].last_step - scns[b].first_step);
        eager |= (scns[b].flags & NB_SCN_EAGER) != 0;
    }
    if (!eager && longest >= GRAPH_MIN_STEPS) {
        GraphGroup g;
        g.lead = c0;
        g.chunk = scns[0].graph_chunk;  // 0 = default
        g.slots.resize((size_t)count);
        for (int b = 0; b < count; ++b) {
            g.slots[(size_t)b].c = ctxs[b];
            g.slots[(size_t)b].scn = &scns[b];
            g.slots[(size_t)b].base = scns[b].first_step;
        }
        std::vector<GraphGroup*> one{&g};
        if (int rc = run_groups_graph(one, FollowerPolicy{})) return rc;
        for (int b = 0; b < count; ++b) fill_result(ctxs[b], &scns[b], g.slots[(size_t)b].sc, g.slots[(size_t)b].done_at, &results[b]);
        return NB_OK;
    }

    const int sync_every = scns[0].sync_every > 0 ? scns[0].sync_every : 2000;
    int running = count;
    for (int t = 1; running > 0; ++t) {  // t-th step of every scenario still running
        F64BatchArgs args{};
        args.count = count;
        for (int b = 0; b < count; ++b) {
            if (done_at[b] >= 0) continue;  // idle slot: item[b].n stays 0
            nb_context* c = ctxs[b];
            const int step = scns[b].first_step + t;
            F64Args a = base_args(c, step);
            a.scn = sc[b];
            a.snap_q = snap[b] ? c->snap_q : nullptr;
            a.snap_v = snap[b] ? c->snap_v : nullptr;
            if (step > scns[b].last_step) {  // the state last_step exists: only its monitor is left
                a.do_update = 0;
                done_at[b] = scns[b].last_step;
                --running;
            }
            args.item[b] = a;
            if (a.do_update) c->cur ^= 1;
        }
        NB_HIP(c0, (hipError_t)launch_f64_batched(args, c0->n, c0->split, stream));
        if (t % sync_every == sync_every - 1) {  // poll the hit flags (hw5.cu:398-402,503-507)
            for (int b = 0; b < count; ++b)
                if (done_at[b] < 0 && scns[b].kind != NB_SCN_MIN_DIST)
                    NB_HIP(c0, hipMemcpyAsync(&ctxs[b]->mon_host->hit_step, &ctxs[b]->mon->hit_step, sizeof(int),
                                              hipMemcpyDeviceToHost, stream));
            NB_HIP(c0, hipStreamSynchronize(stream));
            for (int b = 0; b < count; ++b)
                if (done_at[b] < 0 && scns[b].kind != NB_SCN_MIN_DIST && ctxs[b]->mon_host->hit_step != -2) {
                    done_at[b] = scns[b].first_step + t;
                    --running;
                }
        }
    }
    for (int b = 0; b < count; ++b)
        NB_HIP(c0, hipMemcpyAsync(ctxs[b]->mon_host, ctxs[b]->mon, sizeof(F64Monitor), hipMemcpyDeviceToHost, stream));
    NB_HIP(c0, hipStreamSynchronize(stream));
    for (int b = 0; b < count; ++b) {
        const int hit = ctxs[b]->mon_host->hit_step;  // as above: a hit ends the scenario at the state it was seen in
        if (scns[b].kind != NB_SCN_MIN_DIST && hit != -2) done_at[b] = hit;
        fill_result(ctxs[b], &scns[b], sc[b], done_at[b], &results[b]);
    }
    return NB_OK;
}

}  // namespace

extern "C" {

int nb_run_scenario(nb_context* c, const nb_scenario* s, nb_scenario_result* res) {
    if (!c || !s || !res) return NB_ERR_INVALID;
    try {
        return run_scenario_impl(c, s, res);
    } catch (...) {  // std::bad_alloc from the host-side tables: nothing crosses the C boundary
        return NB_ERR_NOMEM;
    }
}

int nb_run_scenarios_batched(nb_context** ctxs, const nb_scenario* scns, nb_scenario_result* results, int count) {
    try {
        // only the leader of the launch stream passes through bind(): every context's state is about to change, so none may
        // keep treating nb_run_step's last download as a mirror of it
        for (int k = 0; ctxs && k < count && k < MAX_BATCH; ++k)
            if (ctxs[k]) ctxs[k]->stage_fresh = false;
        return run_batched_impl(ctxs, scns, results, count);
    } catch (...) {
        return NB_ERR_NOMEM;
    }
}

}  // extern "C"
