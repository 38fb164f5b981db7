// nbody_sharded.cpp — index-sharded multi-GPU stepping behind the C ABI (nb_sharded_*): ONE process drives P GPUs of a
// node, one HIP stream per GPU, one in-place RCCL all-gather of float4 positions per GPU per step over xGMI.
//
// The reference never shards bodies: each of its 2 GPUs holds the whole system and runs a different scenario
// (hw5.cu:564-567,587-588).  This is the data-parallel scheme the north star adds:
//   rank r of P owns targets [r*N/P, (r+1)*N/P): their velocities (and fp64 masters) live only on GPU r;
//   every GPU holds ALL positions twice (ping-pong float4[N] {x,y,z,G*m}) because every target needs every source;
//   per step and GPU: force + fused kick-drift (samples/nbody.cc:56-88) of the own targets from array `cur`, written into
//   the own slot of array `nxt`, then ncclAllGather(sendbuff = nxt + r*N/P, recvbuff = nxt) — the in-place form.
// With overlap a step is cut into phases: the own shard's sources first — they are final as soon as the
// previous step's kernel has written them — while the all-gather of the other shards is still in flight on a second
// stream; the remote sources follow once it has landed.  The running sums wait in the workspace between the phases.
//
// RCCL is loaded with dlopen when the first sharded system is created, so bin/hw5 and single-GPU users of libnbody_amd
// never pay for (or depend on) librccl.  nbody_amd.distributed is the second host of the same scheme: one process per
// GPU with torch.distributed.
//
// Two exchanges implement the all-gather behind the same stream/event protocol (`exchange_rccl`, `exchange_copy`):
//   RCCL (default)            ncclAllGather in a group call — a kernel per GPU that occupies CUs while it runs;
//   NB_SHARDED_COPY_EXCHANGE  P-1 peer copies per GPU (hipMemcpyPeerAsync: the SDMA engines push the shards over xGMI,
//                             no CU is taken from the force kernel, nothing to load).  Each destination PULLS the other
//                             ranks' slots on its own exchange stream once the owner's `stepped` event has fired.  It is
//                             also the only exchange that accepts the same ordinal more than once in `devices` (ranks
//                             that share a GPU), which is how a one-GPU box runs every P > 1 line of step_once.
//   NB_SHARDED_HOST_EXCHANGE  the last resort (round 5): every GPU downloads its slot into one pinned host array, the host
//                             waits for all of them, every GPU uploads the other slots.  Only per-device operations — no
//                             peer mapping, no cross-device event, no library — for a node whose P2P path is broken; ordered
//                             pairs only (a host-side sum of P partial forces would cost more than the step).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "../../include/nbody_amd_ext.h"
#include "nbody_kernels.h"

using namespace nbk;

namespace {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclReduceScatter) ReduceScatter = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
};

// loaded once per process; never unloaded (communicators may outlive any one sharded system)
const RcclApi* rccl(char* err, size_t errlen) {
    static RcclApi api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
        }
        if (api.handle) {
            api.CommInitAll = (decltype(api.CommInitAll))dlsym(api.handle, "ncclCommInitAll");
            api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.handle, "ncclCommDestroy");
            api.AllGather = (decltype(api.AllGather))dlsym(api.handle, "ncclAllGather");
            api.ReduceScatter = (decltype(api.ReduceScatter))dlsym(api.handle, "ncclReduceScatter");
            api.GroupStart = (decltype(api.GroupStart))dlsym(api.handle, "ncclGroupStart");
            api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.handle, "ncclGroupEnd");
            api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.handle, "ncclGetErrorString");
            api.CommCount = (decltype(api.CommCount))dlsym(api.handle, "ncclCommCount");
            api.CommUserRank = (decltype(api.CommUserRank))dlsym(api.handle, "ncclCommUserRank");
            api.CommCuDevice = (decltype(api.CommCuDevice))dlsym(api.handle, "ncclCommCuDevice");
        }
    }
    if (!api.handle || !api.CommInitAll || !api.CommDestroy || !api.AllGather || !api.ReduceScatter || !api.GroupStart || !api.GroupEnd ||
        !api.GetErrorString || !api.CommCount || !api.CommUserRank || !api.CommCuDevice) {
        snprintf(err, errlen, "cannot load RCCL (librccl.so.1): %s", api.handle ? "missing symbol" : dlerror());
        return nullptr;
    }
    return &api;
}

struct Rank {
    int device = 0, n_cus = 256;
    int64_t lo = 0;
    hipStream_t stream = nullptr, comm_stream = nullptr;
    hipEvent_t stepped = nullptr, gathered = nullptr;
    float4* pos[2] = {nullptr, nullptr};
    float4* vel = nullptr;
    double4* pos64 = nullptr;
    double4* vel64 = nullptr;
    void* ws = nullptr;
    int ws_slots = 0;  // partial-sum slots of `ws` (>= 16; up to 64 when the plan cuts the sources into that many slices)
    size_t ws_bytes = 0;
    // symmetric step (K1s, several GPUs share the pairs): this GPU's partial force on ALL n bodies, and the pieces of the
    // force on its own shard as they arrive from the reduce-scatter (one piece) or from the peers' copies (P pieces)
    void* fpart = nullptr;
    void* facc = nullptr;
    hipEvent_t forced = nullptr;  // copy exchange: fpart is complete
    ncclComm_t comm = nullptr;
    // bounded waits (nb_sharded_set_deadline): `done[i % STEPS_IN_FLIGHT]` is recorded behind step i's last enqueue on every
    // stream the step used; the host waits for it — with the deadline — before it enqueues step i + STEPS_IN_FLIGHT
    std::vector<hipEvent_t> done;
};
constexpr int STEPS_IN_FLIGHT = 16;

}  // namespace

struct nb_sharded {
    int P = 0;
    int64_t n = 0, per = 0;
    int precision = NB_F32;
    int flags = 0;
    double G = 0, eps = 0, dt = 0;
    std::vector<Rank> rank;
    int cur = 0;
    bool have_state = false;
    bool gather_pending = false;  // overlap: the all-gather of pos[cur] is still in flight on the comm streams
    bool sym = false;    // every unordered pair once: K1s on every GPU + a reduce-scatter of the partial forces per step
    F32SymShape shape{};  // of rank 0 (rank r: b0 = r * nb)
    bool ready = false;  // creation went through: streams, buffers and (RCCL) communicators exist
    const RcclApi* api = nullptr;  // null with NB_SHARDED_COPY_EXCHANGE
    double deadline_s = 0;   // > 0: no wait blocks inside the runtime; a step gets this long once the host waits for it
    long enqueued = 0;       // steps enqueued since the streams were last drained (the `done` ring is indexed by it)
    long awaited = 0;        // ... of which the first `awaited` are known to have finished
    bool dead = false;       // a wait timed out: the GPUs may still be busy with what was enqueued; nothing may be enqueued
    float4* host_stage[2] = {nullptr, nullptr};  // NB_SHARDED_HOST_EXCHANGE: pinned float4[n] per ping-pong array
    std::vector<double> m_host;                  // the masses as nb_sharded_set_state got them (checkpoints carry them exactly)
    char err[512] = {0};
};

namespace {

thread_local char g_err[512] = {0};

int fail(nb_sharded* s, int code, const char* what, const char* detail) {
    snprintf(s ? s->err : g_err, sizeof g_err, "%s: %s", what, detail);
    return code;
}

#define SH_HIP(s, call)                                                                \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) return fail(s, NB_ERR_HIP, #call, hipGetErrorString(e_)); \
    } while (0)
#define SH_NCCL(s, call)                                                                      \
    do {                                                                                      \
        ncclResult_t r_ = (call);                                                             \
        if (r_ != ncclSuccess) return fail(s, NB_ERR_HIP, #call, (s)->api->GetErrorString(r_)); \
    } while (0)

// ---- bounded waits.  Without a deadline: the runtime's blocking calls, as always.  With one: poll, sleeping 50 us between
// looks once the first thousand have failed (a step of the bench is milliseconds: the poll costs nothing against it), and
// give the awaited thing `deadline_s` from the moment the host starts to wait for it.
template <class Query>
int wait_until(nb_sharded* s, Query&& query, const char* what) {
    const auto t0 = std::chrono::steady_clock::now();
    for (long spin = 0;; ++spin) {
        const hipError_t e = query();
        if (e == hipSuccess) return NB_OK;
        if (e != hipErrorNotReady) return fail(s, NB_ERR_HIP, what, hipGetErrorString(e));
        (void)hipGetLastError();  // hipErrorNotReady is sticky in the last-error slot on some runtimes: not an error here
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > s->deadline_s) {
            s->dead = true;
            char detail[160];
            snprintf(detail, sizeof detail, "timed out after %.3g s: %s", s->deadline_s, what);
            return fail(s, NB_ERR_HIP, "exchange timed out", detail);
        }
        if (spin > 1000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}
int wait_stream(nb_sharded* s, hipStream_t st, const char* what) {
    if (s->deadline_s <= 0) SH_HIP(s, hipStreamSynchronize(st));
    else return wait_until(s, [&] { return hipStreamQuery(st); }, what);
    return NB_OK;
}
int wait_event(nb_sharded* s, hipEvent_t ev, const char* what) {
    if (s->deadline_s <= 0) SH_HIP(s, hipEventSynchronize(ev));
    else return wait_until(s, [&] { return hipEventQuery(ev); }, what);
    return NB_OK;
}

bool acc64(const nb_sharded* s) { return s->precision == NB_F32_ACC64; }
bool overlapped(const nb_sharded* s) { return (s->flags & NB_SHARDED_OVERLAP) && s->P > 1; }
bool copy_exchange(const nb_sharded* s) { return (s->flags & NB_SHARDED_COPY_EXCHANGE) != 0; }
bool host_exchange(const nb_sharded* s) { return (s->flags & NB_SHARDED_HOST_EXCHANGE) != 0; }
bool uses_rccl(const nb_sharded* s) { return !copy_exchange(s) && !host_exchange(s); }
size_t force_rec(const nb_sharded* s) { return acc64(s) ? sizeof(double4) : sizeof(float4); }

// slots of the source-slice workspace: the documented minimum of 16, or as many as the whole-step plan has slices (up to
// 64) so that a step is ONE force launch + ONE reducer per phase instead of js/16 of each — N = 2^20 over 8 ranks
// (64 slices): 29.7 instead of 30.5 ms per step and rank (profiles/r03_shard_slots.txt); at most 8 GiB
int workspace_slots(const nb_sharded* s, int n_cus) {
    const int js = plan_f32(s->per, s->n, n_cus, 0, 0, true).j_split;
    const int64_t rec = nb_workspace_bytes_f32(s->per, acc64(s)) / (SLICES_PER_LAUNCH + 2);
    const int64_t cap = ((int64_t)8 << 30) / rec - 2;
    return (int)std::max<int64_t>(SLICES_PER_LAUNCH, std::min<int64_t>(std::min<int64_t>(js, MAX_SLICES_PER_LAUNCH), cap));
}
int64_t workspace_bytes(const nb_sharded* s, int slots) {
    return nb_workspace_bytes_f32(s->per, acc64(s)) / (SLICES_PER_LAUNCH + 2) * (slots + 2);
}

// one phase of rank r's step: sources [src_begin, src_end) of the gathered array `cur`
int launch_phase(nb_sharded* s, Rank& k, int64_t src_begin, int64_t src_end, int phase) {
    F32Args a{};
    a.src = k.pos[s->cur];
    a.out = k.pos[s->cur ^ 1];
    a.vel = k.vel;
    a.pos64 = k.pos64;
    a.vel64 = k.vel64;
    a.partial = k.ws;
    a.slots = k.ws_slots;
    a.n_src = s->n;
    a.tgt_off = k.lo;
    a.n_tgt = s->per;
    a.src_begin = src_begin;
    a.src_end = src_end;
    a.phase = phase;
    a.eps2 = (float)(s->eps * s->eps);
    a.dt = (float)s->dt;
    const long cover = (src_begin || src_end) ? (long)(src_end - src_begin) : (long)s->n;
    F32Plan plan = plan_f32(s->per, cover > 0 ? cover : 1, k.n_cus, 0, 0, k.ws != nullptr);
    // one GPU holds the whole system: the symmetric kernel K1s, exactly as nb_step picks it
    if (s->P == 1 && !(s->flags & NB_SHARDED_ORDERED_PAIRS))
        (void)plan_symmetric(plan, s->per, s->n, phase == F32_PHASE_WHOLE, k.ws_bytes, acc64(s), k.n_cus, 0, 0);
    SH_HIP(s, (hipError_t)launch_f32(a, plan, acc64(s), false, k.stream));
    return NB_OK;
}

// ---- the exchange step: all-gather of pos[nxt], every rank's own slot into every other rank's copy of the array.
// Protocol shared by both implementations (ov = overlapped step): the gather runs on the rank's exchange stream — its
// compute stream, or with ov its comm_stream, which first waits for the rank's `stepped` event (own slot written) and
// afterwards records `gathered` (remote slots landed) for the next step's remote-source phases to wait on.

int exchange_rccl(nb_sharded* s, int nxt, bool ov) {
    if (ov)
        for (Rank& k : s->rank) {
            SH_HIP(s, hipSetDevice(k.device));
            SH_HIP(s, hipEventRecord(k.stepped, k.stream));
            SH_HIP(s, hipStreamWaitEvent(k.comm_stream, k.stepped, 0));
        }
    SH_NCCL(s, s->api->GroupStart());
    for (Rank& k : s->rank) {
        float* buf = (float*)k.pos[nxt];
        ncclResult_t r = s->api->AllGather(buf + 4 * k.lo, buf, (size_t)(4 * s->per), ncclFloat, k.comm,
                                           ov ? k.comm_stream : k.stream);
        if (r != ncclSuccess) {
            (void)s->api->GroupEnd();
            return fail(s, NB_ERR_HIP, "ncclAllGather", s->api->GetErrorString(r));
        }
    }
    SH_NCCL(s, s->api->GroupEnd());
    if (ov)
        for (Rank& k : s->rank) {
            SH_HIP(s, hipSetDevice(k.device));
            SH_HIP(s, hipEventRecord(k.gathered, k.comm_stream));
        }
    return NB_OK;
}

// Copy-engine all-gather: destination rank q pulls slot r from rank r's array once r's step kernels have finished.
// Buffer reuse needs no further event: rank r next writes this slot of this array two steps later, behind an exchange
// that waited for q's `stepped` of the step in between, which q's stream reaches only after these copies (stream order,
// or with ov through q's `gathered`).
int exchange_copy(nb_sharded* s, int nxt, bool ov) {
    for (Rank& k : s->rank) {
        SH_HIP(s, hipSetDevice(k.device));
        SH_HIP(s, hipEventRecord(k.stepped, k.stream));
    }
    const size_t bytes = (size_t)s->per * sizeof(float4);
    for (Rank& q : s->rank) {
        SH_HIP(s, hipSetDevice(q.device));
        hipStream_t xs = ov ? q.comm_stream : q.stream;
        if (ov) SH_HIP(s, hipStreamWaitEvent(xs, q.stepped, 0));
        for (Rank& r : s->rank) {
            if (&r == &q) continue;
            SH_HIP(s, hipStreamWaitEvent(xs, r.stepped, 0));
            if (r.device == q.device)
                SH_HIP(s, hipMemcpyAsync(q.pos[nxt] + r.lo, r.pos[nxt] + r.lo, bytes, hipMemcpyDeviceToDevice, xs));
            else
                SH_HIP(s, hipMemcpyPeerAsync(q.pos[nxt] + r.lo, q.device, r.pos[nxt] + r.lo, r.device, bytes, xs));
        }
        if (ov) SH_HIP(s, hipEventRecord(q.gathered, xs));
    }
    return NB_OK;
}

// Host-staged all-gather: D2H of every GPU's own slot into the pinned array of this ping-pong side, a (bounded) host wait
// for all of them, H2D of the other slots on every GPU.  The host array is reused two steps later, behind another such wait,
// which every upload of this step has passed by then.
int exchange_host(nb_sharded* s, int nxt) {
    float4* H = s->host_stage[nxt];
    const size_t N = (size_t)s->n, per = (size_t)s->per;
    for (Rank& k : s->rank) {
        SH_HIP(s, hipSetDevice(k.device));
        SH_HIP(s, hipMemcpyAsync(H + k.lo, k.pos[nxt] + k.lo, per * sizeof(float4), hipMemcpyDeviceToHost, k.stream));
    }
    for (Rank& k : s->rank) {
        SH_HIP(s, hipSetDevice(k.device));
        if (int rc = wait_stream(s, k.stream, "a GPU's step and the download of its shard (host-staged exchange)")) return rc;
    }
    for (Rank& q : s->rank) {
        SH_HIP(s, hipSetDevice(q.device));
        const size_t lo = (size_t)q.lo, hi = lo + per;
        if (lo > 0) SH_HIP(s, hipMemcpyAsync(q.pos[nxt], H, lo * sizeof(float4), hipMemcpyHostToDevice, q.stream));
        if (hi < N) SH_HIP(s, hipMemcpyAsync(q.pos[nxt] + hi, H + hi, (N - hi) * sizeof(float4), hipMemcpyHostToDevice, q.stream));
    }
    return NB_OK;
}

// ---- symmetric step: the unordered pairs of the whole system are shared by the GPUs (GPU r owns the I-superblocks of its
// shard and meets the B/2 superblocks behind each of them, K1s); every GPU ends up with a partial force on ALL n bodies;
// a reduce-scatter hands every shard the sum; the owner kicks and drifts; then the all-gather of positions as always.
int launch_rank_sym(nb_sharded* s, Rank& k, int r) {
    F32Args a{};
    a.src = k.pos[s->cur];
    a.partial = k.ws;
    a.acc = k.fpart;
    a.n_src = s->n;
    a.tgt_off = k.lo;
    a.n_tgt = s->per;
    a.eps2 = (float)(s->eps * s->eps);
    F32SymShape sh = s->shape;
    sh.b0 = r * sh.nb;
    SH_HIP(s, (hipError_t)launch_f32_sym(a, sh, acc64(s), 2, k.stream));
    return NB_OK;
}

int kick_drift_rank(nb_sharded* s, Rank& k, int parts) {
    F32Args a{};
    a.src = k.pos[s->cur];
    a.out = k.pos[s->cur ^ 1];
    a.vel = k.vel;
    a.pos64 = k.pos64;
    a.vel64 = k.vel64;
    a.acc = k.facc;
    a.n_src = s->n;
    a.tgt_off = k.lo;
    a.n_tgt = s->per;
    a.dt = (float)s->dt;
    SH_HIP(s, (hipError_t)launch_kick_drift_f32(a, acc64(s), parts, k.stream));
    return NB_OK;
}

// the partial forces of all GPUs -> the force on every GPU's own shard
int exchange_forces(nb_sharded* s) {
    const size_t rec = force_rec(s);
    if (!copy_exchange(s)) {
        SH_NCCL(s, s->api->GroupStart());
        for (Rank& k : s->rank) {
            ncclResult_t r = s->api->ReduceScatter(k.fpart, k.facc, (size_t)(4 * s->per), acc64(s) ? ncclDouble : ncclFloat,
                                                   ncclSum, k.comm, k.stream);
            if (r != ncclSuccess) {
                (void)s->api->GroupEnd();
                return fail(s, NB_ERR_HIP, "ncclReduceScatter", s->api->GetErrorString(r));
            }
        }
        SH_NCCL(s, s->api->GroupEnd());
        for (Rank& k : s->rank) {
            SH_HIP(s, hipSetDevice(k.device));
            if (int rc = kick_drift_rank(s, k, 1)) return rc;
        }
        return NB_OK;
    }
    // copy engines: destination q pulls its shard of every GPU's partial force (its own included) into P pieces, which the
    // kick-drift kernel adds in rank order.  fpart of rank r is rewritten by r's next force reducer, which r's stream reaches
    // only behind the position exchange that waits for q's `stepped` — recorded after these copies.
    for (Rank& k : s->rank) {
        SH_HIP(s, hipSetDevice(k.device));
        SH_HIP(s, hipEventRecord(k.forced, k.stream));
    }
    const size_t bytes = (size_t)s->per * rec;
    for (Rank& q : s->rank) {
        SH_HIP(s, hipSetDevice(q.device));
        for (size_t r = 0; r < s->rank.size(); ++r) {
            Rank& src = s->rank[r];
            if (&src != &q) SH_HIP(s, hipStreamWaitEvent(q.stream, src.forced, 0));
            char* dst = (char*)q.facc + r * bytes;
            const char* from = (const char*)src.fpart + (size_t)q.lo * rec;
            if (src.device == q.device) SH_HIP(s, hipMemcpyAsync(dst, from, bytes, hipMemcpyDeviceToDevice, q.stream));
            else SH_HIP(s, hipMemcpyPeerAsync(dst, q.device, from, src.device, bytes, q.stream));
        }
        if (int rc = kick_drift_rank(s, q, s->P)) return rc;
    }
    return NB_OK;
}

// nb_sharded_step_profiled: per rank one pair of timing events around the launch sequence of every step
struct StepEvents {
    std::vector<std::vector<hipEvent_t>> begin, end;  // [rank][step]
};

// rank k's launches of one step
int launch_rank(nb_sharded* s, Rank& k, bool ov) {
    if (!ov) return launch_phase(s, k, 0, 0, F32_PHASE_WHOLE);
    // own shard first: final since this GPU's previous launch (or the initial upload); no exchange needed
    const int64_t lo = k.lo, hi = k.lo + s->per;
    if (int rc = launch_phase(s, k, lo, hi, F32_PHASE_FIRST)) return rc;
    if (s->gather_pending) SH_HIP(s, hipStreamWaitEvent(k.stream, k.gathered, 0));  // the other shards have landed
    if (lo > 0)
        if (int rc = launch_phase(s, k, 0, lo, hi < s->n ? F32_PHASE_MIDDLE : F32_PHASE_LAST)) return rc;
    if (hi < s->n)
        if (int rc = launch_phase(s, k, hi, s->n, F32_PHASE_LAST)) return rc;
    return NB_OK;
}

// bounded waits: step `i` of the steps enqueued since the last drain has finished on every GPU
int await_step(nb_sharded* s, long i) {
    for (Rank& k : s->rank) {
        SH_HIP(s, hipSetDevice(k.device));
        if (int rc = wait_event(s, k.done[(size_t)(i % STEPS_IN_FLIGHT)], "a step of the sharded system (kernels + exchange)")) return rc;
    }
    return NB_OK;
}

int step_once(nb_sharded* s, StepEvents* ev = nullptr, size_t step = 0) {
    const bool ov = overlapped(s);
    if (s->deadline_s > 0)  // never more than STEPS_IN_FLIGHT steps ahead of the GPUs: the oldest one gets the deadline
        while (s->enqueued - s->awaited >= STEPS_IN_FLIGHT) {
            if (int rc = await_step(s, s->awaited)) return rc;
            ++s->awaited;
        }
    for (size_t r = 0; r < s->rank.size(); ++r) {
        Rank& k = s->rank[r];
        SH_HIP(s, hipSetDevice(k.device));
        if (ev) SH_HIP(s, hipEventRecord(ev->begin[r][step], k.stream));
        if (int rc = s->sym ? launch_rank_sym(s, k, (int)r) : launch_rank(s, k, ov)) return rc;
        if (ev) SH_HIP(s, hipEventRecord(ev->end[r][step], k.stream));
    }
    if (s->sym)
        if (int rc = exchange_forces(s)) return rc;
    // exchange: every GPU contributes its own slot of the array its kernels have just written
    const int nxt = s->cur ^ 1;
    if (int rc = host_exchange(s) ? exchange_host(s, nxt) : copy_exchange(s) ? exchange_copy(s, nxt, ov) : exchange_rccl(s, nxt, ov)) return rc;
    s->gather_pending = ov;
    s->cur = nxt;
    if (s->deadline_s > 0) {  // (an overlapped step's gather sits on the comm stream: the NEXT step's kernels wait for it, and
                              // the final drain looks at both streams)
        for (Rank& k : s->rank) {
            SH_HIP(s, hipSetDevice(k.device));
            SH_HIP(s, hipEventRecord(k.done[(size_t)(s->enqueued % STEPS_IN_FLIGHT)], k.stream));
        }
        ++s->enqueued;
    }
    return NB_OK;
}

int sync_all(nb_sharded* s) {
    if (s->deadline_s > 0)  // step by step, each with its own allowance; then whatever else sits on the streams
        for (; s->awaited < s->enqueued; ++s->awaited)
            if (int rc = await_step(s, s->awaited)) return rc;
    for (Rank& k : s->rank) {
        SH_HIP(s, hipSetDevice(k.device));
        if (int rc = wait_stream(s, k.stream, "a GPU's compute stream to drain")) return rc;
        if (k.comm_stream)
            if (int rc = wait_stream(s, k.comm_stream, "a GPU's exchange stream to drain")) return rc;
    }
    s->enqueued = s->awaited = 0;
    s->gather_pending = false;
    return NB_OK;
}

void release(nb_sharded* s) {
    if (s->dead) {
        // a wait timed out.  One more allowance for the GPUs to drain; if they do not, everything they may still be reading or
        // writing — buffers, streams, events, communicators — is abandoned rather than freed (hipFree and ncclCommDestroy
        // would block on the wedged work for ever); the process is about to report the failure and exit
        bool idle = true;
        for (Rank& k : s->rank) {
            (void)hipSetDevice(k.device);
            for (hipStream_t st : {k.stream, k.comm_stream})
                if (st && wait_stream(s, st, "the streams to drain before the system is destroyed") != NB_OK) idle = false;
            if (!idle) break;
        }
        if (!idle) return;
    }
    for (Rank& k : s->rank) {
        (void)hipSetDevice(k.device);
        if (k.stream) (void)hipStreamSynchronize(k.stream);
        if (k.comm_stream) (void)hipStreamSynchronize(k.comm_stream);
        for (hipEvent_t e : k.done)
            if (e) (void)hipEventDestroy(e);
        if (k.comm && s->api) (void)s->api->CommDestroy(k.comm);
        for (void* p : {(void*)k.pos[0], (void*)k.pos[1], (void*)k.vel, (void*)k.pos64, (void*)k.vel64, k.ws, k.fpart, k.facc})
            if (p) (void)hipFree(p);
        if (k.forced) (void)hipEventDestroy(k.forced);
        if (k.stepped) (void)hipEventDestroy(k.stepped);
        if (k.gathered) (void)hipEventDestroy(k.gathered);
        if (k.stream) (void)hipStreamDestroy(k.stream);
        if (k.comm_stream) (void)hipStreamDestroy(k.comm_stream);
    }
    for (float4*& h : s->host_stage)  // (every stream that copied from / into them is gone)
        if (h) { (void)hipHostFree(h); h = nullptr; }
}

int create_impl(nb_sharded** out, const int* devices, int n_devices, int64_t n, int precision, double G, double eps,
                double dt, int flags) {
    if (!out) return NB_ERR_INVALID;
    *out = nullptr;
    if (!devices || n_devices <= 0 || n_devices > 64 || n <= 0) return fail(nullptr, NB_ERR_INVALID, "nb_sharded_create", "bad argument");
    if (precision != NB_F32 && precision != NB_F32_ACC64) return fail(nullptr, NB_ERR_INVALID, "nb_sharded_create", "precision must be NB_F32 or NB_F32_ACC64");
    if (!((float)(eps * eps) >= F32_EPS2_MIN)) return fail(nullptr, NB_ERR_INVALID, "nb_sharded_create", "fp32 kernels need eps >= 1e-12 (eps^2 a normal fp32 number with a finite inverse cube)");
    if (n % n_devices) return fail(nullptr, NB_ERR_INVALID, "nb_sharded_create", "n must be divisible by the number of devices");
    // overlap cuts the sources at shard boundaries: they must fall on whole 256-body tiles
    if ((flags & NB_SHARDED_HOST_EXCHANGE) && (flags & (NB_SHARDED_OVERLAP | NB_SHARDED_COPY_EXCHANGE)))
        return fail(nullptr, NB_ERR_INVALID, "nb_sharded_create", "the host-staged exchange is neither overlapped nor combined with the copy exchange");
    if ((flags & NB_SHARDED_OVERLAP) && n_devices > 1 && (n / n_devices) % TILE)
        return fail(nullptr, NB_ERR_INVALID, "nb_sharded_create", "overlap needs n / devices to be a multiple of 256");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return NB_ERR_NO_DEVICE;
    for (int i = 0; i < n_devices; ++i) {
        if (devices[i] < 0 || devices[i] >= ndev) return NB_ERR_NO_DEVICE;
        for (int j = 0; j < i; ++j)
            if (devices[j] == devices[i] && !(flags & (NB_SHARDED_COPY_EXCHANGE | NB_SHARDED_HOST_EXCHANGE)))
                return fail(nullptr, NB_ERR_INVALID, "nb_sharded_create", "a device is listed twice (RCCL: one rank per GPU; NB_SHARDED_COPY_EXCHANGE lets ranks share a GPU)");
    }
    nb_sharded* s = new (std::nothrow) nb_sharded();
    if (!s) return NB_ERR_NOMEM;
    *out = s;  // returned even on failure so the caller can read nb_sharded_last_error, then nb_sharded_destroy
    s->P = n_devices;
    s->n = n;
    s->per = n / n_devices;
    s->precision = precision;
    s->flags = flags;
    s->G = G;
    s->eps = eps;
    s->dt = dt;
    if (uses_rccl(s)) {
        s->api = rccl(s->err, sizeof s->err);
        if (!s->api) return NB_ERR_HIP;
    }
    s->rank.resize((size_t)n_devices);
    const size_t N = (size_t)n, per = (size_t)s->per;
    {   // every unordered pair once when the shards are whole superblocks (and nobody asked for the ordered-pair kernel
        // or the two-phase step, which cuts the sources of K1 into ranges)
        int cus0 = 256;
        if (hipDeviceGetAttribute(&cus0, hipDeviceAttributeMultiprocessorCount, devices[0]) != hipSuccess) cus0 = 256;
        s->sym = !(flags & NB_SHARDED_ORDERED_PAIRS) && !overlapped(s) && !host_exchange(s) &&
                 sym_sharded_ok(n, n_devices, cus0, acc64(s), &s->shape);
    }
    for (int r = 0; r < n_devices; ++r) {
        Rank& k = s->rank[(size_t)r];
        k.device = devices[r];
        k.lo = (int64_t)r * s->per;
        SH_HIP(s, hipSetDevice(k.device));
        SH_HIP(s, hipDeviceGetAttribute(&k.n_cus, hipDeviceAttributeMultiprocessorCount, k.device));
        SH_HIP(s, hipStreamCreateWithFlags(&k.stream, hipStreamNonBlocking));
        if (overlapped(s)) {
            SH_HIP(s, hipStreamCreateWithFlags(&k.comm_stream, hipStreamNonBlocking));
            SH_HIP(s, hipEventCreateWithFlags(&k.gathered, hipEventDisableTiming));
        }
        if (overlapped(s) || copy_exchange(s)) SH_HIP(s, hipEventCreateWithFlags(&k.stepped, hipEventDisableTiming));
        SH_HIP(s, hipMalloc(&k.pos[0], N * sizeof(float4)));
        SH_HIP(s, hipMalloc(&k.pos[1], N * sizeof(float4)));
        SH_HIP(s, hipMalloc(&k.vel, per * sizeof(float4)));
        if (acc64(s)) {
            SH_HIP(s, hipMalloc(&k.pos64, per * sizeof(double4)));
            SH_HIP(s, hipMalloc(&k.vel64, per * sizeof(double4)));
        }
        // the workspace lets the step slice the sources (and carry sums between phases): allocate it whenever the plan
        // would slice, and always for overlap
        if (overlapped(s) || plan_f32(s->per, s->n, k.n_cus, 0, 0, true).j_split > 1) {
            k.ws_slots = workspace_slots(s, k.n_cus);
            k.ws_bytes = (size_t)workspace_bytes(s, k.ws_slots);
        }
    }
    // Every unordered pair once (K1s) wants more memory than the ordered-pair step: pair slots (n^2-ish: 1.7 GB for one GPU at
    // n = 2^20, 2.4 GB per GPU of 8 at 2^22, 11 GB at 2^24) and, with several GPUs, a partial force on all n bodies.  It is a
    // preference, not a requirement: if any GPU cannot give it — more than 3/4 of its free memory, or hipMalloc fails — every
    // GPU gives back what it got and the system steps with ordered pairs (K1); `note` (nb_sharded_last_error after a
    // successful create) says so.  Ranks that share a GPU (copy exchange) see each other's allocations in the free figure.
    const bool want_one = n_devices == 1 && n >= SYM_MIN_N && !(flags & NB_SHARDED_ORDERED_PAIRS);
    if (s->sym || want_one) {
        const char* why = nullptr;
        size_t need = 0, free_b = 0, total_b = 0;
        for (Rank& k : s->rank) {
            SH_HIP(s, hipSetDevice(k.device));
            size_t ws = 0;
            if (want_one) {
                const F32SymBatches kb = sym_batches(n, k.n_cus, acc64(s));
                if (kb.count >= 1 && kb.bytes <= SYM_MAX_WORKSPACE) ws = kb.bytes;
            } else ws = sym_partial_workspace_bytes(s->shape, acc64(s));
            if (!ws) { why = "does not apply"; break; }
            ws = std::max(ws, k.ws_bytes);
            const size_t fp = s->sym ? N * force_rec(s) : 0, fa = s->sym ? (copy_exchange(s) ? (size_t)n_devices : 1) * per * force_rec(s) : 0;
            need = ws + fp + fa;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
            else if ((double)need > 0.75 * (double)free_b) {
                // one GPU holding the whole system: batches of superblocks that fit what is free (memory for speed)
                const F32SymBatches kb = want_one ? sym_batches(n, k.n_cus, acc64(s), (size_t)(0.75 * (double)free_b)) : F32SymBatches{};
                if (kb.count < 1) { why = "more than 3/4 of a GPU's free memory"; break; }
                snprintf(s->err, sizeof s->err, "note: %.1f GB free: the unordered-pair kernel (K1s) steps in %d batches of superblocks with a "
                         "%.1f GB workspace instead of one launch with %.1f GB", free_b / 1e9, kb.count, kb.bytes / 1e9, need / 1e9);
                ws = std::max(kb.bytes, k.ws_bytes);
                need = ws;
            }
            if (hipMalloc(&k.ws, ws) != hipSuccess || (fp && hipMalloc(&k.fpart, fp) != hipSuccess) || (fa && hipMalloc(&k.facc, fa) != hipSuccess)) {
                (void)hipGetLastError();
                why = "hipMalloc failed";
                break;
            }
            k.ws_bytes = ws;
            if (s->sym && copy_exchange(s)) SH_HIP(s, hipEventCreateWithFlags(&k.forced, hipEventDisableTiming));
        }
        if (why) {
            for (Rank& k : s->rank) {
                (void)hipSetDevice(k.device);
                for (void** p : {&k.ws, &k.fpart, &k.facc})
                    if (*p) { (void)hipFree(*p); *p = nullptr; }
                k.ws_bytes = k.ws_slots ? (size_t)workspace_bytes(s, k.ws_slots) : 0;
            }
            if (strcmp(why, "does not apply"))
                snprintf(s->err, sizeof s->err, "note: the unordered-pair step (K1s) needs %.1f GB per GPU (%s, %.1f GB free): stepping with "
                         "ordered pairs (K1) instead", need / 1e9, why, free_b / 1e9);
            s->sym = false;
        }
    }
    for (Rank& k : s->rank) {
        SH_HIP(s, hipSetDevice(k.device));
        if (k.ws_bytes && !k.ws) SH_HIP(s, hipMalloc(&k.ws, k.ws_bytes));
    }
    if (host_exchange(s)) {
        // one pinned array per ping-pong side, visible to every device's copy engine
        for (float4*& h : s->host_stage) SH_HIP(s, hipHostMalloc((void**)&h, N * sizeof(float4), hipHostMallocPortable));
    } else if (copy_exchange(s)) {
        // direct xGMI copies between distinct GPUs; without peer access the runtime stages through the host, which is
        // slower but still correct, so a refusal here is not an error
        for (Rank& q : s->rank)
            for (Rank& r : s->rank)
                if (q.device != r.device) {
                    SH_HIP(s, hipSetDevice(q.device));
                    hipError_t e = hipDeviceEnablePeerAccess(r.device, 0);
                    if (e != hipSuccess) (void)hipGetLastError();  // already enabled / not supported
                }
    } else {
        std::vector<ncclComm_t> comms((size_t)n_devices);
        SH_NCCL(s, s->api->CommInitAll(comms.data(), n_devices, devices));
        for (int r = 0; r < n_devices; ++r) s->rank[(size_t)r].comm = comms[(size_t)r];
    }
    s->ready = true;
    return NB_OK;
}

int set_state_impl(nb_sharded* s, const double* qx, const double* qy, const double* qz, const double* vx,
                   const double* vy, const double* vz, const double* m) {
    const size_t N = (size_t)s->n, per = (size_t)s->per;
    std::vector<float4> p(N), v(per);
    for (size_t i = 0; i < N; ++i)  // G*m folded in fp64, rounded once (as nb_set_state does)
        p[i] = make_float4((float)qx[i], (float)qy[i], (float)qz[i], (float)(s->G * m[i]));
    std::vector<double4> p64, v64;
    if (acc64(s)) { p64.resize(per); v64.resize(per); }
    for (Rank& k : s->rank) {
        SH_HIP(s, hipSetDevice(k.device));
        const size_t lo = (size_t)k.lo;
        for (size_t i = 0; i < per; ++i) v[i] = make_float4((float)vx[lo + i], (float)vy[lo + i], (float)vz[lo + i], 0.f);
        // both ping-pong arrays get every body once: the slots other GPUs own are refreshed by the all-gather, and the
        // G*m column never changes
        // (stream-ordered, never the legacy stream: another host thread may be capturing a graph)
        SH_HIP(s, hipMemcpyAsync(k.pos[0], p.data(), N * sizeof(float4), hipMemcpyHostToDevice, k.stream));
        SH_HIP(s, hipMemcpyAsync(k.pos[1], p.data(), N * sizeof(float4), hipMemcpyHostToDevice, k.stream));
        SH_HIP(s, hipMemcpyAsync(k.vel, v.data(), per * sizeof(float4), hipMemcpyHostToDevice, k.stream));
        if (int rc = wait_stream(s, k.stream, "the upload of the state")) return rc;  // `v` is refilled for the next GPU
        if (acc64(s)) {
            for (size_t i = 0; i < per; ++i) {
                p64[i] = make_double4(qx[lo + i], qy[lo + i], qz[lo + i], s->G * m[lo + i]);
                v64[i] = make_double4(vx[lo + i], vy[lo + i], vz[lo + i], 0.0);
            }
            SH_HIP(s, hipMemcpyAsync(k.pos64, p64.data(), per * sizeof(double4), hipMemcpyHostToDevice, k.stream));
            SH_HIP(s, hipMemcpyAsync(k.vel64, v64.data(), per * sizeof(double4), hipMemcpyHostToDevice, k.stream));
            if (int rc = wait_stream(s, k.stream, "the upload of the state")) return rc;
        }
    }
    s->cur = 0;
    s->gather_pending = false;
    s->have_state = true;
    s->m_host.assign(m, m + N);
    return NB_OK;
}

int get_state_impl(nb_sharded* s, double* qx, double* qy, double* qz, double* vx, double* vy, double* vz) {
    if (int rc = sync_all(s)) return rc;
    const size_t N = (size_t)s->n, per = (size_t)s->per;
    if (acc64(s)) {
        std::vector<double4> p(per), v(per);
        for (Rank& k : s->rank) {
            SH_HIP(s, hipSetDevice(k.device));
            SH_HIP(s, hipMemcpyAsync(p.data(), k.pos64, per * sizeof(double4), hipMemcpyDeviceToHost, k.stream));
            SH_HIP(s, hipMemcpyAsync(v.data(), k.vel64, per * sizeof(double4), hipMemcpyDeviceToHost, k.stream));
            if (int rc = wait_stream(s, k.stream, "the download of the state")) return rc;
            const size_t lo = (size_t)k.lo;
            for (size_t i = 0; i < per; ++i) {
                qx[lo + i] = p[i].x; qy[lo + i] = p[i].y; qz[lo + i] = p[i].z;
                vx[lo + i] = v[i].x; vy[lo + i] = v[i].y; vz[lo + i] = v[i].z;
            }
        }
        return NB_OK;
    }
    // every GPU holds all positions after the all-gather: read them from the last one (the most remote from rank 0's
    // own writes, so a broken exchange shows up in the result), velocities from their owners
    std::vector<float4> p(N), v(per);
    Rank& last = s->rank.back();
    SH_HIP(s, hipSetDevice(last.device));
    SH_HIP(s, hipMemcpyAsync(p.data(), last.pos[s->cur], N * sizeof(float4), hipMemcpyDeviceToHost, last.stream));
    if (int rc = wait_stream(s, last.stream, "the download of the state")) return rc;
    for (size_t i = 0; i < N; ++i) { qx[i] = p[i].x; qy[i] = p[i].y; qz[i] = p[i].z; }
    for (Rank& k : s->rank) {
        SH_HIP(s, hipSetDevice(k.device));
        SH_HIP(s, hipMemcpyAsync(v.data(), k.vel, per * sizeof(float4), hipMemcpyDeviceToHost, k.stream));
        if (int rc = wait_stream(s, k.stream, "the download of the state")) return rc;
        const size_t lo = (size_t)k.lo;
        for (size_t i = 0; i < per; ++i) { vx[lo + i] = v[i].x; vy[lo + i] = v[i].y; vz[lo + i] = v[i].z; }
    }
    return NB_OK;
}

}  // namespace

extern "C" {

int nb_sharded_create(nb_sharded** out, const int* devices, int n_devices, int64_t n, int precision, double G,
                      double eps, double dt, int flags) {
    try {
        return create_impl(out, devices, n_devices, n, precision, G, eps, dt, flags);
    } catch (...) {
        return NB_ERR_NOMEM;
    }
}

int nb_sharded_destroy(nb_sharded* s) {
    if (!s) return NB_ERR_INVALID;
    release(s);
    delete s;
    return NB_OK;
}

const char* nb_sharded_last_error(const nb_sharded* s) { return s ? s->err : g_err; }

int nb_sharded_set_deadline(nb_sharded* s, double seconds) {
    if (!s || !(seconds >= 0)) return NB_ERR_INVALID;
    if (!s->ready || s->dead) return NB_ERR_STATE;
    if (int rc = sync_all(s)) return rc;  // under the old rule; the step counters restart at zero
    try {
        if (seconds > 0)
            for (Rank& k : s->rank) {
                SH_HIP(s, hipSetDevice(k.device));
                k.done.reserve(STEPS_IN_FLIGHT);
                while ((int)k.done.size() < STEPS_IN_FLIGHT) {
                    hipEvent_t e = nullptr;
                    SH_HIP(s, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    k.done.push_back(e);
                }
            }
    } catch (...) {
        return NB_ERR_NOMEM;
    }
    s->deadline_s = seconds;
    return NB_OK;
}

int nb_sharded_set_state(nb_sharded* s, const double* qx, const double* qy, const double* qz, const double* vx,
                         const double* vy, const double* vz, const double* m) {
    if (!s || !qx || !qy || !qz || !vx || !vy || !vz || !m) return NB_ERR_INVALID;
    if (!s->ready || s->dead) return NB_ERR_STATE;  // creation failed half way / a wait timed out
    try {
        return set_state_impl(s, qx, qy, qz, vx, vy, vz, m);
    } catch (...) {
        return NB_ERR_NOMEM;
    }
}

int nb_sharded_get_state(nb_sharded* s, double* qx, double* qy, double* qz, double* vx, double* vy, double* vz) {
    if (!s || !qx || !qy || !qz || !vx || !vy || !vz) return NB_ERR_INVALID;
    if (!s->have_state || s->dead) return NB_ERR_STATE;
    try {
        return get_state_impl(s, qx, qy, qz, vx, vy, vz);
    } catch (...) {
        return NB_ERR_NOMEM;
    }
}

// ---- checkpoints of a sharded run (configs[4] runs for hours; the reference has only its in-memory Problem-3
// snapshot, hw5.cu:265-287): ONE NBODYST2 file for the whole system, the same format nb_save_state writes and bin/hw5 reads
int nb_sharded_save_state(nb_sharded* s, const char* path, int step) {
    if (!s || !path) return NB_ERR_INVALID;
    if (!s->have_state || s->dead) return NB_ERR_STATE;
    try {
        const size_t n = (size_t)s->n;
        std::vector<double> buf(6 * n);
        if (int rc = get_state_impl(s, &buf[0], &buf[n], &buf[2 * n], &buf[3 * n], &buf[4 * n], &buf[5 * n])) return rc;
        nb_state_header h{};
        h.n = s->n;
        h.precision = s->precision;
        h.step = step;
        h.planet = h.asteroid = -1;
        h.G = s->G;
        h.eps = s->eps;
        h.dt = s->dt;
        const int rc = nb_write_state_file(path, &h, &buf[0], &buf[n], &buf[2 * n], &buf[3 * n], &buf[4 * n], &buf[5 * n],
                                           s->m_host.data(), nullptr);
        if (rc) snprintf(s->err, sizeof s->err, "nb_sharded_save_state: %s", nb_last_error(nullptr));
        return rc;
    } catch (...) {
        return NB_ERR_NOMEM;
    }
}

int nb_sharded_load_state(nb_sharded* s, const char* path, int* step) {
    if (!s || !path) return NB_ERR_INVALID;
    if (!s->ready || s->dead) return NB_ERR_STATE;
    try {
        nb_state_header h;
        int rc = nb_read_state_file(path, &h, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        if (rc) {
            snprintf(s->err, sizeof s->err, "nb_sharded_load_state: %s", nb_last_error(nullptr));
            return rc;
        }
        // a checkpoint resumes the run it was taken from (as nb_load_state): same size, arithmetic and integration parameters
        if (h.n != s->n || h.precision != s->precision || h.G != s->G || h.eps != s->eps || h.dt != s->dt) {
            snprintf(s->err, sizeof s->err, "state file (n=%lld precision=%d G=%g eps=%g dt=%g) does not match the system (n=%lld "
                     "precision=%d G=%g eps=%g dt=%g)", (long long)h.n, h.precision, h.G, h.eps, h.dt, (long long)s->n, s->precision,
                     s->G, s->eps, s->dt);
            return NB_ERR_INVALID;
        }
        const size_t n = (size_t)s->n;
        std::vector<double> buf(7 * n);
        rc = nb_read_state_file(path, &h, (int64_t)n, &buf[0], &buf[n], &buf[2 * n], &buf[3 * n], &buf[4 * n], &buf[5 * n], &buf[6 * n],
                                nullptr);
        if (rc) {
            snprintf(s->err, sizeof s->err, "nb_sharded_load_state: %s", nb_last_error(nullptr));
            return rc;
        }
        if (step) *step = h.step;
        return set_state_impl(s, &buf[0], &buf[n], &buf[2 * n], &buf[3 * n], &buf[4 * n], &buf[5 * n], &buf[6 * n]);
    } catch (...) {
        return NB_ERR_NOMEM;
    }
}

int nb_sharded_step(nb_sharded* s, int count) {
    if (!s || count < 0) return NB_ERR_INVALID;
    if (!s->have_state || s->dead) return NB_ERR_STATE;
    for (int i = 0; i < count; ++i)
        if (int rc = step_once(s)) return rc;
    return sync_all(s);
}

int nb_sharded_step_timed(nb_sharded* s, int count, double* ms_per_step) {
    if (!s || count <= 0 || !ms_per_step) return NB_ERR_INVALID;
    if (!s->have_state || s->dead) return NB_ERR_STATE;
    if (int rc = sync_all(s)) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < count; ++i)
        if (int rc = step_once(s)) return rc;
    if (int rc = sync_all(s)) return rc;
    *ms_per_step = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / count;
    return NB_OK;
}

static int step_profiled_impl(nb_sharded* s, int count, double* wall_ms_per_step, float* kernel_ms) {
    if (int rc = sync_all(s)) return rc;
    const size_t P = s->rank.size(), K = (size_t)count;
    StepEvents ev;
    ev.begin.assign(P, std::vector<hipEvent_t>(K, nullptr));
    ev.end.assign(P, std::vector<hipEvent_t>(K, nullptr));
    auto destroy = [&]() {
        for (size_t r = 0; r < P; ++r) {
            (void)hipSetDevice(s->rank[r].device);
            for (size_t i = 0; i < K; ++i) {
                if (ev.begin[r][i]) (void)hipEventDestroy(ev.begin[r][i]);
                if (ev.end[r][i]) (void)hipEventDestroy(ev.end[r][i]);
            }
        }
    };
    auto run = [&]() -> int {
        for (size_t r = 0; r < P; ++r) {
            SH_HIP(s, hipSetDevice(s->rank[r].device));
            for (size_t i = 0; i < K; ++i) {
                SH_HIP(s, hipEventCreate(&ev.begin[r][i]));
                SH_HIP(s, hipEventCreate(&ev.end[r][i]));
            }
        }
        const auto t0 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < K; ++i)
            if (int rc = step_once(s, &ev, i)) return rc;
        if (int rc = sync_all(s)) return rc;
        *wall_ms_per_step = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / count;
        for (size_t r = 0; r < P; ++r) {
            SH_HIP(s, hipSetDevice(s->rank[r].device));
            double sum = 0;
            for (size_t i = 0; i < K; ++i) {
                float ms = 0;
                SH_HIP(s, hipEventElapsedTime(&ms, ev.begin[r][i], ev.end[r][i]));
                sum += ms;
            }
            kernel_ms[r] = (float)(sum / count);
        }
        return NB_OK;
    };
    const int rc = run();
    if (rc) (void)sync_all(s);  // nothing may still be recording into the events
    destroy();
    return rc;
}

int nb_sharded_step_profiled(nb_sharded* s, int count, double* wall_ms_per_step, float* kernel_ms) {
    if (!s || count <= 0 || count > 1024 || !wall_ms_per_step || !kernel_ms) return NB_ERR_INVALID;
    if (!s->have_state || s->dead) return NB_ERR_STATE;
    try {
        return step_profiled_impl(s, count, wall_ms_per_step, kernel_ms);
    } catch (...) {
        return NB_ERR_NOMEM;
    }
}

int nb_sharded_rank_info(const nb_sharded* cs, int rank, nb_sharded_rank* out) {
    nb_sharded* s = const_cast<nb_sharded*>(cs);  // (error text only)
    if (!s || !out || rank < 0 || rank >= (int)s->rank.size() || !s->ready) return NB_ERR_INVALID;
    const Rank& k = s->rank[(size_t)rank];
    memset(out, 0, sizeof *out);
    out->device = k.device;
    out->compute_units = k.n_cus;
    out->first_target = k.lo;
    out->targets = s->per;
    if (copy_exchange(s) || host_exchange(s)) {
        out->exchange = host_exchange(s) ? NB_EXCHANGE_HOST : NB_EXCHANGE_COPY;
        out->comm_ranks = s->P;
        out->comm_rank = rank;
        out->comm_device = k.device;
    } else {
        out->exchange = NB_EXCHANGE_RCCL;
        int v = 0;
        SH_NCCL(s, s->api->CommCount(k.comm, &v));
        out->comm_ranks = v;
        SH_NCCL(s, s->api->CommUserRank(k.comm, &v));
        out->comm_rank = v;
        SH_NCCL(s, s->api->CommCuDevice(k.comm, &v));
        out->comm_device = v;
    }
    SH_HIP(s, hipDeviceGetPCIBusId(out->pci_bus_id, (int)sizeof out->pci_bus_id, k.device));
    hipUUID id;
    SH_HIP(s, hipDeviceGetUuid(&id, k.device));
    for (int i = 0; i < 16; ++i) snprintf(out->uuid + 2 * i, 3, "%02x", (unsigned)(unsigned char)id.bytes[i]);
    hipDeviceProp_t prop;
    SH_HIP(s, hipGetDeviceProperties(&prop, k.device));
    snprintf(out->name, sizeof out->name, "%s", prop.name);
    return NB_OK;
}

const char* nb_sharded_kernel_name(const nb_sharded* s) {
    if (!s || s->rank.empty()) return "";
    const Rank& k = s->rank[0];
    F32Plan p = plan_f32(s->per, s->n, k.n_cus, 0, 0, k.ws != nullptr);
    if (s->P == 1 && !(s->flags & NB_SHARDED_ORDERED_PAIRS))
        (void)plan_symmetric(p, s->per, s->n, true, k.ws_bytes, acc64(s), k.n_cus, 0, 0);
    if (s->sym) p.symmetric = true;
    return kernel_name_f32(p, acc64(s), false);
}

int nb_sharded_info(const nb_sharded* s, int* n_devices, int64_t* targets_per_device, int* targets_per_lane, int* j_split,
                    int* wg_size) {
    if (!s || s->rank.empty()) return NB_ERR_INVALID;
    const Rank& k = s->rank[0];
    F32Plan p = plan_f32(s->per, s->n, k.n_cus, 0, 0, k.ws != nullptr);
    if (s->P == 1 && !(s->flags & NB_SHARDED_ORDERED_PAIRS))
        (void)plan_symmetric(p, s->per, s->n, true, k.ws_bytes, acc64(s), k.n_cus, 0, 0);
    if (s->sym) {
        p.symmetric = true;
        p.sym = s->shape;
        p.targets_per_lane = 2 * SYM_P;
        p.wg_size = SYM_WGS;
        p.j_split = s->shape.chunks;
    }
    if (n_devices) *n_devices = s->P;
    if (targets_per_device) *targets_per_device = s->per;
    if (targets_per_lane) *targets_per_lane = p.targets_per_lane;
    if (j_split) *j_split = p.j_split;
    if (wg_size) *wg_size = p.wg_size;
    return NB_OK;
}

}  // extern "C"
