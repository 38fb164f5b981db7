// nbody_capi.cpp — the core of the C ABI in include/nbody_amd.h on top of the gfx950 kernels: contexts, state,
// nb_step / nb_accel (the run_step call sites, samples/nbody.cc:116,129).
// The raw launches on caller-owned HBM live in nbody_launch.cpp, the scenario drivers, nb_solve and the state files in
// nbody_scenario.cpp, nbody_solve.cpp, nbody_statefile.cpp (shared declarations: nbody_internal.h).
// There is no CPU compute path here: every entry point needs a HIP device and fails loudly without one.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "nbody_internal.h"
#if NB_ABI_DEBUG
#include "../../include/nbody_amd_debug.h"
#endif

using namespace nbk;
using namespace nbi;

namespace nbi {

// text of the last failure of a call that has no context (raw launches, state files, nb_solve): per host thread, read
// with nb_last_error(NULL)
char* thread_error() {
    static thread_local char g_err[512] = {0};
    return g_err;
}

int set_error(int code, const char* text) {
    snprintf(thread_error(), 512, "%s", text);
    return code;
}

int fail_hip(nb_context* c, hipError_t e, const char* what) {
    snprintf(c ? c->err : thread_error(), 512, "%s: %s", what, hipGetErrorString(e));
    return NB_ERR_HIP;
}

int bind(nb_context* c) {
    if (!c) return NB_ERR_INVALID;
    c->stage_fresh = false;  // whatever follows may change the device state: nb_run_step's download is no longer its mirror
    NB_HIP(c, hipSetDevice(c->cfg.device));
    return NB_OK;
}

bool trace_enabled() {
    static const bool on = getenv("NB_SOLVE_TRACE") != nullptr;
    return on;
}

}  // namespace nbi

namespace {

constexpr size_t STAGE_MAX_N = 65536;  // fp64 contexts up to this size move their state through one host staging block

// fp32 modes, systems of thousands to millions of bodies: the state crosses PCIe through a PINNED staging block of the context
// (two halves of 16 MiB, allocated on first use) in chunks — packing / unpacking one half on the host while the copy engine
// moves the other — instead of through pageable vectors (2-3 GB/s): nb_set_state + nb_get_state of 2^20 bodies 19.4 -> ... ms
constexpr size_t PIN_HALF_MAX = (size_t)16 << 20;
int pinned_stage(nb_context* c) {
    if (c->pinned) return NB_OK;
    // a half holds the largest per-body record set of a chunk (set_state of NB_F32_ACC64: 96 B); small systems take less
    const size_t want = ((size_t)c->n * 96 + 4095) / 4096 * 4096;
    c->pin_half = std::min(PIN_HALF_MAX, std::max<size_t>(want, 4096));
    NB_HIP(c, hipHostMalloc(&c->pinned, 2 * c->pin_half, hipHostMallocDefault));
    for (hipEvent_t& e : c->pin_ev) NB_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return NB_OK;
}

void release(nb_context* c) {
    free_dev(c->arena);  // q, v, m, coef, acc, mon, done_dev, ctl
    free_dev(c->snap_q); free_dev(c->snap_v); free_dev(c->fst_dev); free_dev(c->fst_chunk); free_dev(c->stamps);
    free_dev(c->gm_large); free_dev(c->partial_large); free_dev(c->sym64_slots);
    if (c->host_arena) (void)hipHostFree(c->host_arena);  // mon_host, done_host
    free_dev(c->pos[0]); free_dev(c->pos[1]); free_dev(c->vel); free_dev(c->pos64); free_dev(c->vel64);
    free_dev(c->acc32);
    free_dev(c->partial);
    if (c->pinned) (void)hipHostFree(c->pinned);
    for (hipEvent_t e : c->pin_ev)
        if (e) (void)hipEventDestroy(e);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream && c->owns_stream) (void)hipStreamDestroy(c->stream);
}

}  // namespace

F64Args nbi::base_args(nb_context* c, int step) {
    F64Args a{};
    a.qin = c->q[c->cur];
    a.qout = c->q[c->cur ^ 1];
    a.v = c->v;
    a.m = c->m;
    a.coef = c->coef;
    a.mon = c->mon;
    a.n = c->n;
    a.step = step;
    a.do_update = 1;
    a.fst = fst_of(step, c->cfg.dt);
    a.G = c->cfg.G;
    a.eps2 = c->cfg.eps * c->cfg.eps;
    a.dt = c->cfg.dt;
    a.scn.kind = -1;
#if NB_STEP_STAMPS
    a.stamps = c->stamps;
    a.stamp_slots = c->stamp_slots;
#endif
    return a;
}

namespace {

F64LargeArgs large_args(nb_context* c, int step) {
    F64LargeArgs a{};
    a.q = c->q[c->cur];
    a.qout = c->q[c->cur ^ 1];
    a.v = c->v;
    a.m = c->m;
    a.coef = c->coef;
    a.gm = c->gm_large;
    a.partial = c->partial_large;
    a.n = c->n;
    a.j_split = c->slices_large;
    a.sym_slots = c->sym64_slots;
    a.fst = fst_of(step, c->cfg.dt);
    a.G = c->cfg.G;
    a.eps2 = c->cfg.eps * c->cfg.eps;
    a.dt = c->cfg.dt;
    return a;
}

int step_f64(nb_context* c, int first_step, int count) {
    if (c->gm_large) {  // n >= F64_LARGE_MIN (or the config's override)
        for (int s = 0; s < count; ++s) {
            F64LargeArgs a = large_args(c, first_step + s);
            NB_HIP(c, (hipError_t)(a.sym_slots ? launch_f64_large_sym(a, c->n_cus, c->stream) : launch_f64_large(a, c->stream)));
            c->cur ^= 1;
        }
        return NB_OK;
    }
    for (int s = 0; s < count; ++s) {
        F64Args a = base_args(c, first_step + s);
        NB_HIP(c, (hipError_t)launch_f64(a, c->split, c->stream));
        c->cur ^= 1;
    }
    return NB_OK;
}

F32Args f32_args(nb_context* c) {
    F32Args a{};
    a.src = c->pos[c->cur];
    a.out = c->pos[c->cur ^ 1];
    a.vel = c->vel;
    a.pos64 = c->pos64;
    a.vel64 = c->vel64;
    a.acc = c->acc32;
    a.partial = c->partial;
    a.slots = c->partial_slots;
    a.n_src = c->n;
    a.tgt_off = 0;
    a.n_tgt = c->n;
    a.eps2 = (float)(c->cfg.eps * c->cfg.eps);
    a.dt = (float)c->cfg.dt;
    return a;
}

// K1s' pair-slot workspace, on the first step / accel that would use it (not in nb_create: 1.7 GB at n = 2^20, 10 GB at 2^24 is a heavy
// default for a context that may only hold state, and several contexts may share a GPU).  If the device cannot give it —
// more than 3/4 of what is free, or hipMalloc fails — the context keeps K1's workspace (288 B per body, allocated at
// creation) and evaluates every ordered pair for the rest of its life; nb_last_error(ctx) says so, no call fails.
void ensure_sym_workspace(nb_context* c) {
    if (c->sym_tried) return;
    c->sym_tried = true;
    // K1's source-slice workspace (k1_bytes: up to 66 records per body) is only needed when K1 runs: below SYM_MIN_N bodies it
    // was allocated by nb_create; above, it is allocated here — and only if K1s' own workspace is not to be had
    auto k1_workspace = [&]() {
        if (c->partial || !c->k1_bytes) return;
        if (hipMalloc(&c->partial, c->k1_bytes) != hipSuccess) {  // not fatal either: K1 then runs unsliced
            (void)hipGetLastError();
            c->partial = nullptr;
            c->partial_slots = 0;
            return;
        }
        c->partial_bytes = c->k1_bytes;
    };
    if (!c->sym_bytes) return k1_workspace();
    const bool acc64 = c->cfg.precision == NB_F32_ACC64;
    size_t need = c->sym_bytes;
    size_t free_b = 0, total_b = 0;
    const bool known = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
    if (!known) (void)hipGetLastError();
    void* ws = nullptr;
    const char* why = nullptr;
    int batches = 1;
    const size_t room = known ? (size_t)(0.75 * (double)free_b) : need;
    if (need > room) {
        // the default shape does not fit: smaller batches of superblocks that do (more launches for less memory, at no
        // measurable cost: profiles/r05_workspace_cap_ab.txt)
        const F32SymBatches kb = sym_batches(c->n, c->n_cus, acc64, room);
        if (kb.count >= 1 && kb.bytes <= room) {
            need = kb.bytes;
            batches = kb.count;
        } else {
            why = "more than 3/4 of the free device memory even in batches of 16 superblocks";
        }
    }
    if (!why && hipMalloc(&ws, need) != hipSuccess) {
        (void)hipGetLastError();  // not an error of this context: the fallback below is the answer
        ws = nullptr;
        why = "hipMalloc failed";
    }
    if (!ws) {
        snprintf(c->err, sizeof c->err, "note: the %.1f GB pair-slot workspace of the unordered-pair kernel (K1s) was not allocated (%s, "
                 "%.1f GB free): this context evaluates every ordered pair (K1) instead", need / 1e9, why, free_b / 1e9);
        c->sym_bytes = 0;
        return k1_workspace();
    }
    if (need < c->sym_bytes)
        snprintf(c->err, sizeof c->err, "note: %.1f GB free: the unordered-pair kernel (K1s) steps in %d batches of superblocks with a %.1f GB "
                 "workspace instead of its default %.1f GB", free_b / 1e9, batches, need / 1e9, c->sym_bytes / 1e9);
    c->partial = ws;
    c->partial_bytes = need;
    c->sym_bytes = need;
}

// the context's launch plan: K1s (every unordered pair once) when the system is large enough for it and its workspace
// could be had, else K1
F32Plan context_plan_f32(nb_context* c) {
    ensure_sym_workspace(c);
    F32Plan plan = plan_f32(c->n, c->n, c->n_cus, 0, 0, c->partial != nullptr, 0, 0,
                            c->partial_slots > 0 ? c->partial_slots : MAX_SLICES_PER_LAUNCH);
    if (c->sym_bytes)
        (void)plan_symmetric(plan, c->n, c->n, true, c->partial_bytes, c->cfg.precision == NB_F32_ACC64, c->n_cus, 0, 0);
    return plan;
}

int step_f32(nb_context* c, int count) {
    const bool acc64 = c->cfg.precision == NB_F32_ACC64;
    const F32Plan plan = context_plan_f32(c);
    for (int s = 0; s < count; ++s) {
        F32Args a = f32_args(c);
        NB_HIP(c, (hipError_t)launch_f32(a, plan, acc64, false, c->stream));
        c->cur ^= 1;
    }
    return NB_OK;
}

}  // namespace

extern "C" {

int nb_abi_version(void) { return NB_ABI_VERSION; }

int nb_device_count(int* count) {
    if (!count) return NB_ERR_INVALID;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess) ? n : 0;
    return (e == hipSuccess && n > 0) ? NB_OK : NB_ERR_NO_DEVICE;
}

int nb_config_default(nb_config* cfg) {
    if (!cfg) return NB_ERR_INVALID;
    memset(cfg, 0, sizeof *cfg);
    cfg->precision = NB_F64;
    cfg->device = 0;
    cfg->G = 6.674e-11;  // samples/nbody.cc:13
    cfg->eps = 1e-3;     // :12
    cfg->dt = 60;        // :11
    return NB_OK;
}

const char* nb_strerror(int code) {
    switch (code) {
        case NB_OK: return "ok";
        case NB_ERR_INVALID: return "invalid argument";
        case NB_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
        case NB_ERR_HIP: return "HIP runtime error (see nb_last_error)";
        case NB_ERR_STATE: return "call sequence error (state not set?)";
        case NB_ERR_NOMEM: return "out of memory";
        case NB_ERR_IO: return "I/O error";
    }
    return "unknown error";
}

const char* nb_last_error(const nb_context* ctx) { return ctx ? ctx->err : thread_error(); }

int nb_create(nb_context** out, const nb_config* cfg) { return create_context(out, cfg, nullptr); }

#if NB_ABI_DEBUG
int nb_create_cu_masked(nb_context** out, const nb_config* cfg, int cu_mask) {
    if (cu_mask < NB_CU_ALL || cu_mask > NB_CU_ODD) return NB_ERR_INVALID;
    return create_context(out, cfg, nullptr, cu_mask);
}
#endif

}  // extern "C"

// `borrowed`: use this stream (of another context on the same GPU, which must outlive this one) instead of creating one
// `cu_mask` != 0 (instrumented build only, nb_create_cu_masked): the stream is confined to half of the compute units
int nbi::create_context(nb_context** out, const nb_config* cfg, hipStream_t borrowed, int cu_mask) {
    if (!out || !cfg || cfg->n <= 0) return NB_ERR_INVALID;
    if (cfg->precision < NB_F64 || cfg->precision > NB_F32_ACC64) return NB_ERR_INVALID;
    // fp32 kernels evaluate the self pair as 0 * G*m*eps2^-1.5: eps^2 must be a normal fp32 number with a finite inverse cube
    if (cfg->precision != NB_F64 && !((float)(cfg->eps * cfg->eps) >= F32_EPS2_MIN))
        return set_error(NB_ERR_INVALID, "nb_create: the fp32 modes need eps >= 1e-12");
    if (cfg->flags & ~(NB_CFG_ORDERED_PAIRS | NB_CFG_WORKSPACE_GIB(0xffff))) return NB_ERR_INVALID;
    if (cfg->f64_split < 0 || cfg->f64_split > 64 || (cfg->f64_split & (cfg->f64_split - 1))) return NB_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return NB_ERR_NO_DEVICE;
    if (cfg->device < 0 || cfg->device >= ndev) return NB_ERR_NO_DEVICE;
    nb_context* c = new (std::nothrow) nb_context();
    if (!c) return NB_ERR_NOMEM;
    c->cfg = *cfg;
    c->n = cfg->n;
    for (int k = 0; k < NB_MAX_WATCH; ++k) c->snap_arrival[k] = -2;
    *out = c;  // returned even on failure so the caller can read nb_last_error, then nb_destroy
    NB_HIP(c, hipSetDevice(cfg->device));
    // (one attribute, not hipGetDeviceProperties: nb_solve creates 2 + D contexts per program run)
    NB_HIP(c, hipDeviceGetAttribute(&c->n_cus, hipDeviceAttributeMultiprocessorCount, cfg->device));
    if (borrowed) {
        c->stream = borrowed;
        c->owns_stream = false;
#if NB_ABI_DEBUG
    } else if (cu_mask != NB_CU_ALL) {  // bench/scenario_concurrency.py: two scenario streams that do not share CUs
        uint32_t mask[8];
        const int cus = std::min(256, c->n_cus), how = cu_mask;
        for (int w = 0; w < 8; ++w) mask[w] = 0;
        for (int i = 0; i < cus; ++i) {
            const bool on = how == NB_CU_LOW ? i < cus / 2 : how == NB_CU_HIGH ? i >= cus / 2 : how == NB_CU_EVEN ? !(i & 1) : (i & 1);
            if (on) mask[i >> 5] |= 1u << (i & 31);
        }
        NB_HIP(c, hipExtStreamCreateWithCUMask(&c->stream, 8, mask));
#endif
    } else {
        NB_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    }
    NB_HIP(c, hipEventCreate(&c->ev0));
    NB_HIP(c, hipEventCreate(&c->ev1));
    const size_t n = (size_t)c->n;
    if (cfg->precision == NB_F64) {
        // one arena: [q0 3n][q1 3n][v 3n][acc 3n][m n][coef n] doubles, then monitor, done word, graph control word
        const size_t planes = (14 * n * sizeof(double) + 255) / 256 * 256;
        NB_HIP(c, hipMalloc(&c->arena, planes + 256 + 256));
        double* base = (double*)c->arena;
        c->q[0] = base; c->q[1] = base + 3 * n; c->v = base + 6 * n; c->acc = base + 9 * n;
        c->m = base + 12 * n; c->coef = base + 13 * n;
        c->mon = (F64Monitor*)((char*)c->arena + planes);
        static_assert(sizeof(F64Monitor) <= 224, "monitor + done word share one 256-byte slot");
        c->done_dev = (int*)((char*)c->arena + planes + 224);
        c->ctl = (F64Ctl*)((char*)c->arena + planes + 256);
        NB_HIP(c, hipHostMalloc(&c->host_arena, 256));
        c->mon_host = (F64Monitor*)c->host_arena;
        c->done_host = (int*)((char*)c->host_arena + 224);
        c->ctl_host = (F64Ctl*)((char*)c->host_arena + 232);
        c->split = cfg->f64_split > 0 ? cfg->f64_split : auto_split_f64(c->n, c->n_cus);
        const int large_min = cfg->f64_large_min > 0 ? cfg->f64_large_min : F64_LARGE_MIN;
        // every unordered pair once (K1s-f64) where its slots are affordable and eps > 0 (the self pair then adds +0): from 12288
        // bodies on (SYM64_MIN_SB superblocks) unless the caller moved the threshold of the large path himself
        const bool sym64 = cfg->eps * cfg->eps >= F64_EPS2_MIN && sym64_workspace_bytes(c->n, c->n_cus) > 0;
        if (c->n >= large_min || (cfg->f64_large_min <= 0 && sym64)) {  // plain steps of a large fp64 system: K1s-f64, else K1-f64
            c->slices_large = plan_f64_large_slices(c->n, c->n_cus);
            NB_HIP(c, hipMalloc(&c->gm_large, n * sizeof(double)));
            if (c->slices_large > 1)
                NB_HIP(c, hipMalloc(&c->partial_large, (size_t)c->slices_large * 3 * n * sizeof(double)));
            if (sym64) NB_HIP(c, hipMalloc(&c->sym64_slots, sym64_workspace_bytes(c->n, c->n_cus)));
        }
    } else {
        NB_HIP(c, hipMalloc(&c->pos[0], n * sizeof(float4)));
        NB_HIP(c, hipMalloc(&c->pos[1], n * sizeof(float4)));
        NB_HIP(c, hipMalloc(&c->vel, n * sizeof(float4)));
        NB_HIP(c, hipMalloc(&c->acc32, n * sizeof(double4)));
        const int js = plan_f32(c->n, c->n, c->n_cus, 0, 0, true).j_split;
        if (js > 1) {  // the step launches will slice the sources: one partial-sum slot per slice of a launch (16 .. 64,
                       // within 8 GiB), so that a step is one force launch + one reducer (profiles/r03_shard_slots.txt)
            const size_t rec = cfg->precision == NB_F32_ACC64 ? sizeof(double4) : sizeof(float4);
            const long cap = (long)(((size_t)8 << 30) / (n * rec)) - 2;
            c->partial_slots = (int)std::max<long>(SLICES_PER_LAUNCH, std::min<long>(std::min<long>(js, MAX_SLICES_PER_LAUNCH), cap));
            c->k1_bytes = (size_t)(c->partial_slots + 2) * n * rec;
        }
        if (c->n >= SYM_MIN_N && !(cfg->flags & NB_CFG_ORDERED_PAIRS)) {
            // K1s: a slot per superblock round (1.6 GB at 2^20), in batches beyond 32 GiB of them — wanted, not yet allocated
            // (ensure_sym_workspace, on the first step)
            const size_t cap = (size_t)((cfg->flags >> 8) & 0xffff) << 30;  // NB_CFG_WORKSPACE_GIB: the caller's own limit
            F32SymBatches kb = sym_batches(c->n, c->n_cus, cfg->precision == NB_F32_ACC64);
            if (cap && kb.count >= 1 && kb.bytes > cap) kb = sym_batches(c->n, c->n_cus, cfg->precision == NB_F32_ACC64, cap);
            if (kb.count >= 1 && kb.bytes <= SYM_MAX_WORKSPACE) c->sym_bytes = kb.bytes;
        }
        if (c->k1_bytes && !c->sym_bytes) {  // K1 is what this context will run: its slices now (a K1s context asks at its first step)
            NB_HIP(c, hipMalloc(&c->partial, c->k1_bytes));
            c->partial_bytes = c->k1_bytes;
        }
        if (cfg->precision == NB_F32_ACC64) {
            NB_HIP(c, hipMalloc(&c->pos64, n * sizeof(double4)));
            NB_HIP(c, hipMalloc(&c->vel64, n * sizeof(double4)));
        }
    }
    return NB_OK;
}

extern "C" {

int nb_destroy(nb_context* ctx) {
    if (!ctx) return NB_ERR_INVALID;
    (void)hipSetDevice(ctx->cfg.device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    release(ctx);
    delete ctx;
    return NB_OK;
}

static int nb_set_state_impl(nb_context* c, const double* qx, const double* qy, const double* qz, const double* vx,
                 const double* vy, const double* vz, const double* m, const uint8_t* is_device) {
    if (!c || !qx || !qy || !qz || !vx || !vy || !vz || !m) return NB_ERR_INVALID;
    if (int rc = bind(c)) return rc;
    const size_t n = (size_t)c->n;
    c->m_host.assign(m, m + n);
    c->dev_host.assign(n, 0);
    if (is_device) c->dev_host.assign(is_device, is_device + n);
    if (c->cfg.precision == NB_F64) {
        std::vector<double> coef(n);
        for (size_t i = 0; i < n; ++i) coef[i] = c->dev_host[i] ? 0.5 : 0.0;  // nbody.cc:15
        double* q = c->q[0];
        c->cur = 0;
        const size_t B = n * sizeof(double);
        if (n <= STAGE_MAX_N) {
            // small systems (the testcases; a caller that round-trips the state every step): a transfer
            // costs ~6 us whatever its size, so the eight arrays travel as the three contiguous ranges of the arena they fill
            // — [q0: x y z], [v: x y z], [m, coef] — from one host staging block: 3 copies instead of 8
            std::vector<double>& st = c->stage_host;
            st.resize(8 * n);
            memcpy(&st[0], qx, B); memcpy(&st[n], qy, B); memcpy(&st[2 * n], qz, B);
            memcpy(&st[3 * n], vx, B); memcpy(&st[4 * n], vy, B); memcpy(&st[5 * n], vz, B);
            memcpy(&st[6 * n], m, B); memcpy(&st[7 * n], coef.data(), B);
            NB_HIP(c, hipMemcpyAsync(q, &st[0], 3 * B, hipMemcpyHostToDevice, c->stream));
            NB_HIP(c, hipMemcpyAsync(c->v, &st[3 * n], 3 * B, hipMemcpyHostToDevice, c->stream));
            NB_HIP(c, hipMemcpyAsync(c->m, &st[6 * n], 2 * B, hipMemcpyHostToDevice, c->stream));  // coef sits right behind m
            NB_HIP(c, hipStreamSynchronize(c->stream));
            c->have_state = true;
            return NB_OK;
        }
        NB_HIP(c, hipMemcpyAsync(q, qx, B, hipMemcpyHostToDevice, c->stream));
        NB_HIP(c, hipMemcpyAsync(q + n, qy, B, hipMemcpyHostToDevice, c->stream));
        NB_HIP(c, hipMemcpyAsync(q + 2 * n, qz, B, hipMemcpyHostToDevice, c->stream));
        NB_HIP(c, hipMemcpyAsync(c->v, vx, B, hipMemcpyHostToDevice, c->stream));
        NB_HIP(c, hipMemcpyAsync(c->v + n, vy, B, hipMemcpyHostToDevice, c->stream));
        NB_HIP(c, hipMemcpyAsync(c->v + 2 * n, vz, B, hipMemcpyHostToDevice, c->stream));
        NB_HIP(c, hipMemcpyAsync(c->m, m, B, hipMemcpyHostToDevice, c->stream));
        NB_HIP(c, hipMemcpyAsync(c->coef, coef.data(), B, hipMemcpyHostToDevice, c->stream));
        NB_HIP(c, hipStreamSynchronize(c->stream));  // host buffers are the caller's: done with them on return
    } else {
        for (size_t i = 0; i < n; ++i)
            if (c->dev_host[i]) return NB_ERR_INVALID;  // the device-mass law needs fp64 state (SURVEY A-4)
        if (int rc = pinned_stage(c)) return rc;
        const bool a64 = c->cfg.precision == NB_F32_ACC64;
        const size_t per_body = 2 * sizeof(float4) + (a64 ? 2 * sizeof(double4) : 0);
        const size_t chunk = c->pin_half / per_body;
        c->cur = 0;
        size_t k = 0;
        for (size_t i0 = 0; i0 < n; i0 += chunk, ++k) {
            const size_t cnt = std::min(chunk, n - i0);
            char* half = (char*)c->pinned + (k & 1) * c->pin_half;
            if (k >= 2) NB_HIP(c, hipEventSynchronize(c->pin_ev[k & 1]));  // the copies that read this half two chunks ago are done
            float4* p = (float4*)half;
            float4* v = p + cnt;
            double4* p64 = (double4*)(v + cnt);
            double4* v64 = p64 + cnt;
            for (size_t j = 0; j < cnt; ++j) {
                const size_t i = i0 + j;
                // G*m folded in fp64, rounded once (keeps G*m ~ 1e-10*m away from fp32 underflow)
                p[j] = make_float4((float)qx[i], (float)qy[i], (float)qz[i], (float)(c->cfg.G * m[i]));
                v[j] = make_float4((float)vx[i], (float)vy[i], (float)vz[i], 0.f);
                if (a64) {
                    p64[j] = make_double4(qx[i], qy[i], qz[i], c->cfg.G * m[i]);
                    v64[j] = make_double4(vx[i], vy[i], vz[i], 0.0);
                }
            }
            NB_HIP(c, hipMemcpyAsync(c->pos[0] + i0, p, cnt * sizeof(float4), hipMemcpyHostToDevice, c->stream));
            NB_HIP(c, hipMemcpyAsync(c->vel + i0, v, cnt * sizeof(float4), hipMemcpyHostToDevice, c->stream));
            if (a64) {
                NB_HIP(c, hipMemcpyAsync(c->pos64 + i0, p64, cnt * sizeof(double4), hipMemcpyHostToDevice, c->stream));
                NB_HIP(c, hipMemcpyAsync(c->vel64 + i0, v64, cnt * sizeof(double4), hipMemcpyHostToDevice, c->stream));
            }
            NB_HIP(c, hipEventRecord(c->pin_ev[k & 1], c->stream));
        }
        NB_HIP(c, hipStreamSynchronize(c->stream));
    }
    c->have_state = true;
    return NB_OK;
}

static int nb_get_state_impl(nb_context* c, double* qx, double* qy, double* qz, double* vx, double* vy, double* vz) {
    if (!c || !qx || !qy || !qz || !vx || !vy || !vz) return NB_ERR_INVALID;
    if (!c->have_state) return NB_ERR_STATE;
    if (int rc = bind(c)) return rc;
    const size_t n = (size_t)c->n;
    if (c->cfg.precision == NB_F64) {
        const double* q = c->q[c->cur];
        const size_t B = n * sizeof(double);
        if (n <= STAGE_MAX_N) {  // as nb_set_state: the two contiguous ranges q[cur], v in 2 copies instead of 6
            std::vector<double>& st = c->stage_host;
            st.resize(8 * n);
            NB_HIP(c, hipMemcpyAsync(&st[0], q, 3 * B, hipMemcpyDeviceToHost, c->stream));
            NB_HIP(c, hipMemcpyAsync(&st[3 * n], c->v, 3 * B, hipMemcpyDeviceToHost, c->stream));
            NB_HIP(c, hipStreamSynchronize(c->stream));
            memcpy(qx, &st[0], B); memcpy(qy, &st[n], B); memcpy(qz, &st[2 * n], B);
            memcpy(vx, &st[3 * n], B); memcpy(vy, &st[4 * n], B); memcpy(vz, &st[5 * n], B);
            return NB_OK;
        }
        NB_HIP(c, hipMemcpyAsync(qx, q, B, hipMemcpyDeviceToHost, c->stream));
        NB_HIP(c, hipMemcpyAsync(qy, q + n, B, hipMemcpyDeviceToHost, c->stream));
        NB_HIP(c, hipMemcpyAsync(qz, q + 2 * n, B, hipMemcpyDeviceToHost, c->stream));
        NB_HIP(c, hipMemcpyAsync(vx, c->v, B, hipMemcpyDeviceToHost, c->stream));
        NB_HIP(c, hipMemcpyAsync(vy, c->v + n, B, hipMemcpyDeviceToHost, c->stream));
        NB_HIP(c, hipMemcpyAsync(vz, c->v + 2 * n, B, hipMemcpyDeviceToHost, c->stream));
        NB_HIP(c, hipStreamSynchronize(c->stream));
    } else {
        if (int rc = pinned_stage(c)) return rc;
        const bool a64 = c->cfg.precision == NB_F32_ACC64;  // the fp64 masters are the state of that mode
        const size_t rec = a64 ? sizeof(double4) : sizeof(float4);
        const size_t chunk = c->pin_half / (2 * rec);
        const size_t chunks = (n + chunk - 1) / chunk;
        const char* dpos = a64 ? (const char*)c->pos64 : (const char*)c->pos[c->cur];
        const char* dvel = a64 ? (const char*)c->vel64 : (const char*)c->vel;
        auto issue = [&](size_t k) -> int {  // chunk k -> half k & 1 (unpacked two iterations ago)
            const size_t i0 = k * chunk, cnt = std::min(chunk, n - i0);
            char* half = (char*)c->pinned + (k & 1) * c->pin_half;
            NB_HIP(c, hipMemcpyAsync(half, dpos + i0 * rec, cnt * rec, hipMemcpyDeviceToHost, c->stream));
            NB_HIP(c, hipMemcpyAsync(half + cnt * rec, dvel + i0 * rec, cnt * rec, hipMemcpyDeviceToHost, c->stream));
            NB_HIP(c, hipEventRecord(c->pin_ev[k & 1], c->stream));
            return NB_OK;
        };
        if (int rc = issue(0)) return rc;
        for (size_t k = 0; k < chunks; ++k) {
            if (k + 1 < chunks)
                if (int rc = issue(k + 1)) return rc;
            NB_HIP(c, hipEventSynchronize(c->pin_ev[k & 1]));
            const size_t i0 = k * chunk, cnt = std::min(chunk, n - i0);
            const char* half = (const char*)c->pinned + (k & 1) * c->pin_half;
            if (a64) {
                const double4* p = (const double4*)half;
                const double4* v = p + cnt;
                for (size_t j = 0; j < cnt; ++j) {
                    const size_t i = i0 + j;
                    qx[i] = p[j].x; qy[i] = p[j].y; qz[i] = p[j].z;
                    vx[i] = v[j].x; vy[i] = v[j].y; vz[i] = v[j].z;
                }
            } else {
                const float4* p = (const float4*)half;
                const float4* v = p + cnt;
                for (size_t j = 0; j < cnt; ++j) {
                    const size_t i = i0 + j;
                    qx[i] = p[j].x; qy[i] = p[j].y; qz[i] = p[j].z;
                    vx[i] = v[j].x; vy[i] = v[j].y; vz[i] = v[j].z;
                }
            }
        }
    }
    return NB_OK;
}

int nb_set_mass(nb_context* c, int index, double m) {
    if (!c || index < 0 || index >= c->n) return NB_ERR_INVALID;
    if (!c->have_state) return NB_ERR_STATE;
    if (c->cfg.precision != NB_F64) return NB_ERR_INVALID;
    if (int rc = bind(c)) return rc;
    c->m_host[index] = m;
    NB_HIP(c, hipMemcpyAsync(c->m + index, &c->m_host[index], sizeof(double), hipMemcpyHostToDevice, c->stream));
    NB_HIP(c, hipStreamSynchronize(c->stream));
    return NB_OK;
}

#if NB_ABI_DEBUG  // include/nbody_amd_debug.h: exported by the instrumented build only
int nb_enable_step_stamps(nb_context* c, int slots) {
    if (!c || slots < 0 || slots > (1 << 20) || c->cfg.precision != NB_F64) return NB_ERR_INVALID;
    if (int rc = bind(c)) return rc;
    NB_HIP(c, hipStreamSynchronize(c->stream));
    free_dev(c->stamps);
    c->stamp_slots = 0;
    if (slots) {
        NB_HIP(c, hipMalloc(&c->stamps, (size_t)slots * 2 * sizeof(unsigned long long)));
        NB_HIP(c, hipMemsetAsync(c->stamps, 0, (size_t)slots * 2 * sizeof(unsigned long long), c->stream));
        NB_HIP(c, hipStreamSynchronize(c->stream));
        c->stamp_slots = slots;
    }
    return NB_OK;
}

int nb_read_step_stamps(nb_context* c, uint64_t* out, int slots) {
    if (!c || !out || slots <= 0 || slots > c->stamp_slots) return NB_ERR_INVALID;
    if (int rc = bind(c)) return rc;
    NB_HIP(c, hipMemcpyAsync(out, c->stamps, (size_t)slots * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    NB_HIP(c, hipStreamSynchronize(c->stream));
    return NB_OK;
}
#endif  // NB_ABI_DEBUG

const char* nb_context_kernel_name(nb_context* c) {
    if (!c || c->cfg.precision == NB_F64) return "";
    if (bind(c)) return "";
    return kernel_name_f32(context_plan_f32(c), c->cfg.precision == NB_F32_ACC64, false);
}

static int nb_run_step_impl(nb_context* c, int step, double* qx, double* qy, double* qz, double* vx, double* vy, double* vz,
                            const double* m, const uint8_t* is_device) {
    if (!c || !qx || !qy || !qz || !vx || !vy || !vz || !m) return NB_ERR_INVALID;
    const size_t n = (size_t)c->n;
    if (c->cfg.precision != NB_F64 || n > STAGE_MAX_N) {  // the plain sequence (large systems: the transfers are not the cost)
        if (int rc = nb_set_state(c, qx, qy, qz, vx, vy, vz, m, is_device)) return rc;
        if (int rc = nb_step(c, step, 1)) return rc;
        return nb_get_state(c, qx, qy, qz, vx, vy, vz);
    }
    const bool fresh = c->stage_fresh && c->have_state;  // (read before bind() clears it)
    if (int rc = bind(c)) return rc;
    const size_t B = n * sizeof(double);
    std::vector<double>& st = c->stage_host;
    st.resize(8 * n);
    // the host arrays still hold what the previous call returned?  then the GPU already has this state (the reference's own
    // loop never touches q, v between two run_step calls: nbody.cc:114-122,127-138)
    const bool same_qv = fresh && !memcmp(&st[0], qx, B) && !memcmp(&st[n], qy, B) && !memcmp(&st[2 * n], qz, B) &&
                         !memcmp(&st[3 * n], vx, B) && !memcmp(&st[4 * n], vy, B) && !memcmp(&st[5 * n], vz, B);
    bool same_m = c->have_state && c->m_host.size() == n && !memcmp(c->m_host.data(), m, B);
    if (same_m)
        for (size_t i = 0; i < n && same_m; ++i) same_m = c->dev_host[i] == (is_device ? is_device[i] : 0);
    if (!same_qv) {
        memcpy(&st[0], qx, B); memcpy(&st[n], qy, B); memcpy(&st[2 * n], qz, B);
        memcpy(&st[3 * n], vx, B); memcpy(&st[4 * n], vy, B); memcpy(&st[5 * n], vz, B);
        NB_HIP(c, hipMemcpyAsync(c->q[c->cur], &st[0], 3 * B, hipMemcpyHostToDevice, c->stream));
        NB_HIP(c, hipMemcpyAsync(c->v, &st[3 * n], 3 * B, hipMemcpyHostToDevice, c->stream));
    }
    if (!same_m) {
        c->m_host.assign(m, m + n);
        c->dev_host.assign(n, 0);
        if (is_device) c->dev_host.assign(is_device, is_device + n);
        memcpy(&st[6 * n], m, B);
        for (size_t i = 0; i < n; ++i) st[7 * n + i] = c->dev_host[i] ? 0.5 : 0.0;  // nbody.cc:15
        NB_HIP(c, hipMemcpyAsync(c->m, &st[6 * n], 2 * B, hipMemcpyHostToDevice, c->stream));  // coef sits right behind m
    }
    c->have_state = true;
    if (int rc = step_f64(c, step, 1)) return rc;
    NB_HIP(c, hipMemcpyAsync(&st[0], c->q[c->cur], 3 * B, hipMemcpyDeviceToHost, c->stream));
    NB_HIP(c, hipMemcpyAsync(&st[3 * n], c->v, 3 * B, hipMemcpyDeviceToHost, c->stream));
    NB_HIP(c, hipStreamSynchronize(c->stream));
    memcpy(qx, &st[0], B); memcpy(qy, &st[n], B); memcpy(qz, &st[2 * n], B);
    memcpy(vx, &st[3 * n], B); memcpy(vy, &st[4 * n], B); memcpy(vz, &st[5 * n], B);
    c->stage_fresh = true;
    return NB_OK;
}

int nb_run_step(nb_context* c, int step, double* qx, double* qy, double* qz, double* vx, double* vy, double* vz, const double* m,
                const uint8_t* is_device) {
    try {
        return nb_run_step_impl(c, step, qx, qy, qz, vx, vy, vz, m, is_device);
    } catch (...) {  // std::bad_alloc from the host staging vectors
        return NB_ERR_NOMEM;
    }
}

int nb_step(nb_context* c, int first_step, int count) {
    if (!c || count < 0) return NB_ERR_INVALID;
    if (!c->have_state) return NB_ERR_STATE;
    if (int rc = bind(c)) return rc;
    int rc = (c->cfg.precision == NB_F64) ? step_f64(c, first_step, count) : step_f32(c, count);
    if (rc) return rc;
    NB_HIP(c, hipStreamSynchronize(c->stream));
    return NB_OK;
}

int nb_step_timed(nb_context* c, int first_step, int count, float* ms_per_step) {
    if (!c || count <= 0 || !ms_per_step) return NB_ERR_INVALID;
    if (!c->have_state) return NB_ERR_STATE;
    if (int rc = bind(c)) return rc;
    NB_HIP(c, hipEventRecord(c->ev0, c->stream));
    int rc = (c->cfg.precision == NB_F64) ? step_f64(c, first_step, count) : step_f32(c, count);
    if (rc) return rc;
    NB_HIP(c, hipEventRecord(c->ev1, c->stream));
    NB_HIP(c, hipEventSynchronize(c->ev1));
    float ms = 0;
    NB_HIP(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *ms_per_step = ms / count;
    return NB_OK;
}

static int nb_accel_impl(nb_context* c, int step, double* ax, double* ay, double* az) {
    if (!c || !ax || !ay || !az) return NB_ERR_INVALID;
    if (!c->have_state) return NB_ERR_STATE;
    if (int rc = bind(c)) return rc;
    const size_t n = (size_t)c->n;
    if (c->cfg.precision == NB_F64) {
        if (c->gm_large) {
            F64LargeArgs a = large_args(c, step);
            a.acc_out = c->acc;
            NB_HIP(c, (hipError_t)(a.sym_slots ? launch_f64_large_sym(a, c->n_cus, c->stream) : launch_f64_large(a, c->stream)));
        } else {
            F64Args a = base_args(c, step);
            a.acc_out = c->acc;
            NB_HIP(c, (hipError_t)launch_f64(a, c->split, c->stream));
        }
        const size_t B = n * sizeof(double);
        NB_HIP(c, hipMemcpyAsync(ax, c->acc, B, hipMemcpyDeviceToHost, c->stream));
        NB_HIP(c, hipMemcpyAsync(ay, c->acc + n, B, hipMemcpyDeviceToHost, c->stream));
        NB_HIP(c, hipMemcpyAsync(az, c->acc + 2 * n, B, hipMemcpyDeviceToHost, c->stream));
        NB_HIP(c, hipStreamSynchronize(c->stream));
    } else {
        const bool acc64 = c->cfg.precision == NB_F32_ACC64;
        const F32Plan plan = context_plan_f32(c);  // first: it may replace the workspace (ensure_sym_workspace) the arguments point at
        F32Args a = f32_args(c);
        NB_HIP(c, (hipError_t)launch_f32(a, plan, acc64, true, c->stream));
        if (acc64) {
            std::vector<double4> h(n);
            NB_HIP(c, hipMemcpyAsync(h.data(), c->acc32, n * sizeof(double4), hipMemcpyDeviceToHost, c->stream));
            NB_HIP(c, hipStreamSynchronize(c->stream));
            for (size_t i = 0; i < n; ++i) { ax[i] = h[i].x; ay[i] = h[i].y; az[i] = h[i].z; }
        } else {
            std::vector<float4> h(n);
            NB_HIP(c, hipMemcpyAsync(h.data(), c->acc32, n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
            NB_HIP(c, hipStreamSynchronize(c->stream));
            for (size_t i = 0; i < n; ++i) { ax[i] = h[i].x; ay[i] = h[i].y; az[i] = h[i].z; }
        }
    }
    return NB_OK;
}

static int nb_restore_snapshot_impl(nb_context* dst, nb_context* src, int slot) {
    if (!dst || !src || slot < 0 || slot >= src->snap_slots) return NB_ERR_INVALID;
    if (dst->n != src->n || dst->cfg.precision != NB_F64 || src->cfg.precision != NB_F64) return NB_ERR_INVALID;
    if (src->snap_arrival[slot] == -2) {  // that device's missile never arrived: the slot is uninitialised memory
        snprintf(dst->err, sizeof dst->err, "snapshot slot %d holds no state (no missile arrival recorded)", slot);
        return NB_ERR_STATE;
    }
    const size_t n = (size_t)src->n;
    std::vector<double> q(3 * n), v(3 * n);
    if (int rc = bind(src)) return rc;
    NB_HIP(src, hipMemcpyAsync(q.data(), src->snap_q + (size_t)slot * 3 * n, 3 * n * sizeof(double), hipMemcpyDeviceToHost,
                               src->stream));
    NB_HIP(src, hipMemcpyAsync(v.data(), src->snap_v + (size_t)slot * 3 * n, 3 * n * sizeof(double), hipMemcpyDeviceToHost,
                               src->stream));
    NB_HIP(src, hipStreamSynchronize(src->stream));
    return nb_set_state(dst, q.data(), q.data() + n, q.data() + 2 * n, v.data(), v.data() + n, v.data() + 2 * n,
                        src->m_host.data(), src->dev_host.data());
}

int nb_set_state(nb_context* c, const double* qx, const double* qy, const double* qz, const double* vx,
                 const double* vy, const double* vz, const double* m, const uint8_t* is_device) {
    try {
        return nb_set_state_impl(c, qx, qy, qz, vx, vy, vz, m, is_device);
    } catch (...) {  // std::bad_alloc from the host staging vectors
        return NB_ERR_NOMEM;
    }
}

int nb_get_state(nb_context* c, double* qx, double* qy, double* qz, double* vx, double* vy, double* vz) {
    try {
        return nb_get_state_impl(c, qx, qy, qz, vx, vy, vz);
    } catch (...) {  // std::bad_alloc from the host staging vectors
        return NB_ERR_NOMEM;
    }
}

int nb_accel(nb_context* c, int step, double* ax, double* ay, double* az) {
    try {
        return nb_accel_impl(c, step, ax, ay, az);
    } catch (...) {  // std::bad_alloc from the host staging vectors
        return NB_ERR_NOMEM;
    }
}

int nb_restore_snapshot(nb_context* dst, nb_context* src, int slot) {
    try {
        return nb_restore_snapshot_impl(dst, src, slot);
    } catch (...) {  // std::bad_alloc from the host staging vectors
        return NB_ERR_NOMEM;
    }
}

}  // extern "C"
