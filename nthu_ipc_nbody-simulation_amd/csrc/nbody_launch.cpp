// nbody_launch.cpp — the raw launches of the C ABI on caller-owned HBM (device pointers + a hipStream_t): nb_launch_*_f32,
// their plan / kernel-name / workspace queries, and the two-call form of the step in which several GPUs share the unordered
// pairs of one system (nb_launch_pair_forces_f32 -> the host's reduce-scatter -> nb_launch_kick_drift_f32).  For hosts that
// own device memory and the collectives themselves: bench.py, nbody_amd.distributed (one process per GPU, torch + RCCL).
// The call site these replace is run_step at samples/nbody.cc:116,129 — one launch sequence per step and rank.
#include <algorithm>
#include <cstdint>
#include <vector>

#include "nbody_internal.h"

using namespace nbk;
using namespace nbi;

extern "C" {

// ---------------------------------------------------------------- raw launches on caller-owned HBM
static int check_launch(const nb_launch_f32* a, bool accel_only) {
    if (!a || !a->src || a->n_src <= 0 || a->n_tgt <= 0 || a->tgt_off < 0) return NB_ERR_INVALID;
    if (!a->tgt && a->tgt_off + a->n_tgt > a->n_src) return NB_ERR_INVALID;  // targets are a window of the sources
    if (!(a->eps2 >= F32_EPS2_MIN)) return set_error(NB_ERR_INVALID, "fp32 kernels need eps2 >= 1e-24 (eps >= 1e-12): the self pair is 0 * G*m*eps2^-1.5");
    if (accel_only ? !a->acc : (!a->out || (a->acc64 ? (!a->pos64 || !a->vel64) : !a->vel))) return NB_ERR_INVALID;
    const int r = a->targets_per_lane;
    if (r != 0 && r != 2 && r != 4 && r != 8) return NB_ERR_INVALID;
    if (a->j_split < 0 || a->j_split > MAX_JSPLIT) return NB_ERR_INVALID;
    if (a->j_split > 1 && !a->workspace) return NB_ERR_INVALID;
    if (a->source_path < 0 || a->source_path > 3) return NB_ERR_INVALID;
    if (a->wg_size != 0 && a->wg_size != 256 && a->wg_size != 512 && a->wg_size != 1024) return NB_ERR_INVALID;
    if (a->phase < NB_PHASE_WHOLE || a->phase > NB_PHASE_MIDDLE) return NB_ERR_INVALID;
    if (a->src_begin || a->src_end) {  // a sub-range of the sources: whole 256-body tiles, except at the very end
        if (a->src_begin < 0 || a->src_begin > a->src_end || a->src_end > a->n_src) return NB_ERR_INVALID;
        if (a->src_begin % TILE || (a->src_end % TILE && a->src_end != a->n_src)) return NB_ERR_INVALID;
    }
    if (a->phase != NB_PHASE_WHOLE && (!a->workspace || a->workspace_bytes < nb_workspace_bytes_f32(a->n_tgt, a->acc64)))
        return set_error(NB_ERR_INVALID, "a step cut into phases keeps its running sums in the workspace");
    return NB_OK;
}

// partial-sum slots the caller's workspace holds behind the running sum and its compensation (records 0 and 1; 18 records
// per target is the documented minimum; a larger workspace, up to 66 records, lets up to 64 slices go out in one launch)
static int workspace_slots(const nb_launch_f32* a) {
    if (!a->workspace || a->n_tgt <= 0) return 0;
    const size_t rec = a->acc64 ? sizeof(double4) : sizeof(float4);
    const long records = (long)((size_t)a->workspace_bytes / ((size_t)a->n_tgt * rec));
    return (int)std::min<long>(std::max<long>(records - 2, 0), MAX_SLICES_PER_LAUNCH);
}

static F32Plan resolve_plan(const nb_launch_f32* a) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    // slices are planned for the sources this launch covers (a phase of a step covers a sub-range)
    const long n_cover = (a->src_begin || a->src_end) ? std::max<long>(1, a->src_end - a->src_begin) : a->n_src;
    F32Plan p = plan_f32(a->n_tgt, n_cover, cus, a->targets_per_lane, a->j_split, a->workspace != nullptr,
                         a->source_path, a->wg_size, workspace_slots(a));  // (a small whole system: slices one launch can hold)
    // the caller's workspace must hold SLICES_PER_LAUNCH partial records + running sum + compensation per target
    if (workspace_slots(a) < SLICES_PER_LAUNCH) p.j_split = 1;
    // K1s (every unordered pair once): the whole system in one launch and a workspace of nb_workspace_bytes_sym_f32
    const bool whole = a->phase == NB_PHASE_WHOLE && !a->src_begin && !a->src_end && !a->tgt && a->tgt_off == 0;
    // (a forced register blocking, workgroup size or slice count asks for K1; with source_path 3 j_split = chunks)
    if (a->targets_per_lane == 0 && a->wg_size == 0 && (a->j_split == 0 || a->source_path == 3))
        (void)plan_symmetric(p, a->n_tgt, a->n_src, whole, a->workspace ? (size_t)a->workspace_bytes : 0, a->acc64 != 0, cus,
                             a->source_path, a->j_split);
    return p;
}

static int refuse_unmet_symmetric(const nb_launch_f32* a, const F32Plan& p) {
    if (a->source_path == 3 && !p.symmetric)
        return set_error(NB_ERR_INVALID, "source_path 3 (every unordered pair once) needs the whole system in one launch "
                         "(n_tgt == n_src, tgt_off 0, no phases), n_src >= 28672 and a workspace of nb_workspace_bytes_sym_f32");
    return NB_OK;
}

static F32Args to_args(const nb_launch_f32* a) {
    F32Args k{};
    k.src = (const float4*)a->src;
    k.tgt = (const float4*)a->tgt;  // null -> src + tgt_off
    k.out = (float4*)a->out;
    k.vel = (float4*)a->vel;
    k.pos64 = (double4*)a->pos64;
    k.vel64 = (double4*)a->vel64;
    k.acc = a->acc;
    k.partial = a->workspace;
    k.slots = workspace_slots(a);
    k.n_src = a->n_src;
    k.tgt_off = a->tgt_off;
    k.n_tgt = a->n_tgt;
    k.src_begin = a->src_begin;
    k.src_end = a->src_end;
    k.phase = a->phase;
    k.eps2 = a->eps2;
    k.dt = a->dt;
    return k;
}

int nb_launch_step_f32(const nb_launch_f32* a, void* hip_stream) {
    if (int rc = check_launch(a, false)) return rc;
    const F32Plan plan = resolve_plan(a);
    if (int rc = refuse_unmet_symmetric(a, plan)) return rc;
    hipError_t e = (hipError_t)launch_f32(to_args(a), plan, a->acc64 != 0, false, (hipStream_t)hip_stream);
    return e == hipSuccess ? NB_OK : fail_hip(nullptr, e, "nb_launch_step_f32");
}

int nb_launch_accel_f32(const nb_launch_f32* a, void* hip_stream) {
    if (int rc = check_launch(a, true)) return rc;
    const F32Plan plan = resolve_plan(a);
    if (int rc = refuse_unmet_symmetric(a, plan)) return rc;
    hipError_t e = (hipError_t)launch_f32(to_args(a), plan, a->acc64 != 0, true, (hipStream_t)hip_stream);
    return e == hipSuccess ? NB_OK : fail_hip(nullptr, e, "nb_launch_accel_f32");
}

const char* nb_kernel_name_f32(const nb_launch_f32* a, int accel_only) {
    if (!a) return "";
    return kernel_name_f32(resolve_plan(a), a->acc64 != 0, accel_only != 0);
}

int nb_plan_f32(const nb_launch_f32* a, int* targets_per_lane, int* j_split, int* wg_size) {
    if (!a || a->n_src <= 0 || a->n_tgt <= 0) return NB_ERR_INVALID;
    F32Plan p = resolve_plan(a);
    if (targets_per_lane) *targets_per_lane = p.targets_per_lane;
    if (j_split) *j_split = p.j_split;
    if (wg_size) *wg_size = p.wg_size;
    return NB_OK;
}

// ---- several GPUs sharing the unordered pairs of one system (hosts that own the collectives: nbody_amd.distributed)
static bool shared_pairs_shape(const nb_launch_f32* a, F32SymShape* sh) {
    if (!a || a->n_tgt <= 0 || a->n_src <= 0 || a->n_src % a->n_tgt || a->tgt_off % a->n_tgt) return false;
    const int P = (int)(a->n_src / a->n_tgt), rank = (int)(a->tgt_off / a->n_tgt);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (rank >= P || !sym_sharded_ok(a->n_src, P, cus, a->acc64 != 0, sh)) return false;
    sh->b0 = rank * sh->nb;
    return true;
}

int nb_launch_pair_forces_f32(const nb_launch_f32* a, void* hip_stream) {
    F32SymShape sh{};
    if (!a || !a->src || !a->acc || !a->workspace || !(a->eps2 >= F32_EPS2_MIN) || a->tgt || a->phase != NB_PHASE_WHOLE || a->src_begin ||
        a->src_end || !shared_pairs_shape(a, &sh))
        return set_error(NB_ERR_INVALID, "nb_launch_pair_forces_f32: the shard must be whole 4096-body superblocks of a system of "
                         ">= 28672 bodies (n_src = ranks * n_tgt, tgt_off = rank * n_tgt), with acc and a workspace");
    if ((size_t)a->workspace_bytes < sym_partial_workspace_bytes(sh, a->acc64 != 0))
        return set_error(NB_ERR_INVALID, "nb_launch_pair_forces_f32: workspace smaller than nb_workspace_bytes_shared_pairs_f32");
    hipError_t e = (hipError_t)launch_f32_sym(to_args(a), sh, a->acc64 != 0, 2, (hipStream_t)hip_stream);
    return e == hipSuccess ? NB_OK : fail_hip(nullptr, e, "nb_launch_pair_forces_f32");
}

int nb_launch_kick_drift_f32(const nb_launch_f32* a, int parts, void* hip_stream) {
    if (!a || !a->src || !a->out || !a->acc || a->n_tgt <= 0 || a->tgt_off < 0 || a->tgt_off + a->n_tgt > a->n_src || parts < 1 ||
        (a->acc64 ? (!a->pos64 || !a->vel64) : !a->vel))
        return NB_ERR_INVALID;
    hipError_t e = (hipError_t)launch_kick_drift_f32(to_args(a), a->acc64 != 0, parts, (hipStream_t)hip_stream);
    return e == hipSuccess ? NB_OK : fail_hip(nullptr, e, "nb_launch_kick_drift_f32");
}

int64_t nb_workspace_bytes_shared_pairs_f32(int64_t n_src, int ranks, int acc64) {
    if (ranks < 2 || n_src <= 0 || n_src % ranks) return 0;
    nb_launch_f32 a{};
    a.n_src = n_src;
    a.n_tgt = n_src / ranks;
    a.acc64 = acc64;
    F32SymShape sh{};
    return shared_pairs_shape(&a, &sh) ? (int64_t)sym_partial_workspace_bytes(sh, acc64 != 0) : 0;
}

int nb_plan_shared_pairs_f32(int64_t n_src, int ranks, int acc64, int* superblocks_per_rank, int* workgroups_per_superblock,
                             int* sub_launches) {
    if (ranks < 2 || n_src <= 0 || n_src % ranks) return NB_ERR_INVALID;
    nb_launch_f32 a{};
    a.n_src = n_src;
    a.n_tgt = n_src / ranks;
    a.acc64 = acc64;
    F32SymShape sh{};
    if (!shared_pairs_shape(&a, &sh)) return set_error(NB_ERR_INVALID, "nb_plan_shared_pairs_f32: these ranks cannot share the pairs of this system");
    // the grid of the (first) launch: a share that goes out in sub-launches is re-cut for the smaller superblock count
    const int sub = sym_sub_batch(sh, acc64 != 0);
    const F32SymShape first = sub >= sh.nb ? sh : sym_sub_shape(sh, sh.b0, sub);
    if (superblocks_per_rank) *superblocks_per_rank = sh.nb;
    if (workgroups_per_superblock) *workgroups_per_superblock = first.chunks;
    if (sub_launches) *sub_launches = (sh.nb + sub - 1) / sub;
    return NB_OK;
}

// Host-only replay of K1s' pair schedule for `n` bodies on `ranks` GPUs of `n_cus` compute units each (ranks = 1: the
// one-GPU launch) with the very index arithmetic the kernels use (sym_chunk_range / sym_piece / sym_piece_slot /
// sym_for_each_slot_of, nbody_kernels.h).  Checks, for shapes no box here can run (8 GPUs): every unordered pair of
// superblocks is met in all its tile phases exactly once over all ranks and workgroups, and every diagonal block once; no
// two workgroups of a launch write the same (slot, J-superblock) region; the slots the reducer adds for a superblock are
// exactly the ones that were written for it; the workspace is large enough.  0 = consistent; else NB_ERR_STATE with the first
// inconsistency in msg.
static int selftest_pair_schedule(int64_t n, int n_cus, int ranks, int acc64, size_t budget, char* msg, int msg_len) {
    auto say = [&](const char* fmt, long a = 0, long b = 0, long c = 0, long d = 0) {
        if (msg && msg_len > 0) snprintf(msg, (size_t)msg_len, fmt, a, b, c, d);
        return NB_ERR_STATE;
    };
    if (msg && msg_len > 0) msg[0] = 0;
    if (n <= 0 || n_cus <= 0 || ranks < 1) return NB_ERR_INVALID;
    const bool a64 = acc64 != 0;
    // the launches that together cover the system: one per rank; on one GPU one, or one per batch of I-superblocks
    std::vector<F32SymShape> launches;
    try {
        if (ranks == 1) {
            const F32SymBatches kb = sym_batches(n, n_cus, a64, budget);
            if (kb.count < 1) return say("K1s does not apply to %ld bodies", n);
            if (budget && kb.bytes > budget) return say("the batches need %ld bytes, more than the budget of %ld", (long)kb.bytes, (long)budget);
            if (kb.count == 1) launches.push_back(sym_shape(n, n_cus));
            else
                for (int k = 0; k < kb.count; ++k) {
                    launches.push_back(sym_batch_shape(n, n_cus, kb, k, a64));
                    if (sym_workspace_bytes(launches.back(), a64) + (size_t)launches.back().npad * (a64 ? 32 : 16) > kb.bytes)
                        return say("batch %ld needs more workspace than sym_batches reports", k);
                }
        } else {
            F32SymShape base{};
            if (!sym_sharded_ok(n, ranks, n_cus, a64, &base)) return say("the %ld ranks cannot share the pairs of %ld bodies", ranks, n);
            for (int r = 0; r < ranks; ++r) {  // a rank's launch, in sub-launches when its slots would not fit the budget
                F32SymShape mine = base;
                mine.b0 = r * base.nb;
                const int sub = sym_sub_batch(mine, a64);
                if (sub >= mine.nb) launches.push_back(mine);
                else
                    for (int b0 = mine.b0; b0 < mine.b0 + mine.nb; b0 += sub) {
                        launches.push_back(sym_sub_shape(mine, b0, b0 + sub <= mine.b0 + mine.nb ? sub : mine.b0 + mine.nb - b0, a64,
                                                         sym_workspace_bytes(sym_sub_shape(mine, mine.b0, sub), a64)));
                        if (sym_workspace_bytes(launches.back(), a64) > sym_partial_workspace_bytes(mine, a64))
                            return say("a sub-launch of rank %ld needs more workspace than sym_partial_workspace_bytes reports", r);
                    }
            }
        }
    } catch (...) {
        return NB_ERR_NOMEM;
    }
    const int B = launches[0].B;
    try {
        // phases[b * B + J]: bit mask of the tile phases in which I-superblock b has met superblock J (J == b: diagonal)
        std::vector<uint32_t> phases((size_t)B * B, 0u);
        for (size_t rank = 0; rank < launches.size(); ++rank) {
            const F32SymShape sh = launches[rank];
            const int slots = sym_total_slots(sh, a64);
            if ((size_t)slots * (size_t)sh.npad * 3 * sizeof(float) != sym_workspace_bytes(sh, a64)) return say("workspace size formula");
            std::vector<uint8_t> written((size_t)slots * B, 0);  // (slot, J-superblock) regions written by this launch
            for (int b = sh.b0; b < sh.b0 + sh.nb; ++b)
                for (int chunk = 0; chunk < sh.chunks; ++chunk) {
                    if (written[(size_t)(chunk * (a64 ? 2 : 1)) * B + b]++) return say("own slot of chunk %ld written twice for superblock %ld", chunk, b);
                    if (a64) written[(size_t)(2 * chunk + 1) * B + b]++;
                    long q, q_hi;
                    sym_chunk_range(sh, b, chunk, &q, &q_hi);
                    while (q < q_hi) {
                        int u, ph0, ph1;
                        q = sym_piece(q, q_hi, &u, &ph0, &ph1);
                        if (ph1 <= ph0 || ph1 > SYM_NT || u > sym_rounds(B, b)) return say("bad piece: unit %ld phases %ld..%ld of superblock %ld", u, ph0, ph1, b);
                        const int J = (b + u) % B;
                        for (int ph = ph0; ph < ph1; ++ph) {
                            uint32_t& m = phases[(size_t)b * B + J];
                            if (m & (1u << ph)) return say("superblock %ld meets %ld twice in phase %ld", b, J, ph);
                            m |= 1u << ph;
                        }
                        if (u == 0) continue;
                        const int slot = sym_piece_slot(sh, a64, b, chunk, u, ph0);
                        if (slot < sym_own_slots(sh, a64) || slot >= slots) return say("slot %ld out of range (superblock %ld, round %ld)", slot, b, u);
                        if (written[(size_t)slot * B + J]++) return say("slot %ld written twice for superblock %ld (by %ld, round %ld)", slot, J, b, u);
                    }
                }
            for (int J = 0; J < B; ++J) {  // the reducer's view of this launch
                std::vector<uint8_t> added((size_t)slots, 0);
                int bad = -1;
                sym_for_each_slot_of(sh, a64, J, [&](long slot) {
                    if (slot < 0 || slot >= slots || added[(size_t)slot]++) bad = (int)slot;
                });
                if (bad >= 0) return say("reducer adds slot %ld twice or out of range for superblock %ld (launch %ld)", bad, J, (long)rank);
                for (int sl = 0; sl < slots; ++sl)
                    if ((added[(size_t)sl] != 0) != (written[(size_t)sl * B + J] != 0))
                        return say("slot %ld of superblock %ld (launch %ld): written %ld but the reducer disagrees", sl, J, (long)rank, written[(size_t)sl * B + J]);
            }
        }
        const uint32_t all = SYM_NT >= 32 ? 0xffffffffu : ((1u << SYM_NT) - 1);
        for (int b = 0; b < B; ++b)
            for (int J = b; J < B; ++J) {
                const uint32_t f = phases[(size_t)b * B + J], r = phases[(size_t)J * B + b];
                if (J == b ? f != all : !((f == all && r == 0) || (f == 0 && r == all)))
                    return say("superblocks %ld and %ld: phase masks %ld / %ld (each unordered pair once, all phases)", b, J, (long)f, (long)r);
            }
    } catch (...) {
        return NB_ERR_NOMEM;
    }
    return NB_OK;
}

int nb_selftest_pair_schedule(int64_t n, int n_cus, int ranks, int acc64, char* msg, int msg_len) {
    return selftest_pair_schedule(n, n_cus, ranks, acc64, 0, msg, msg_len);
}

int nb_selftest_pair_schedule_within(int64_t n, int n_cus, int acc64, int64_t workspace_bytes, char* msg, int msg_len) {
    if (workspace_bytes <= 0) return NB_ERR_INVALID;
    return selftest_pair_schedule(n, n_cus, 1, acc64, (size_t)workspace_bytes, msg, msg_len);
}

int64_t nb_workspace_bytes_sym_f32(int64_t n, int acc64) {
    if (n < SYM_MIN_N) return 0;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const F32SymBatches kb = sym_batches(n, cus, acc64 != 0);
    if (kb.count < 1) return 0;
    size_t b = kb.bytes;
    if (kb.count == 1) {  // one launch: room for up to 8 workgroups per superblock (j_split with source_path 3)
        F32SymShape sh = sym_shape(n, cus);
        sh.chunks = std::max(sh.chunks, 8);
        b = sym_workspace_bytes(sh, acc64 != 0);
    }
    return b <= SYM_MAX_WORKSPACE ? (int64_t)b : 0;
}

int64_t nb_workspace_bytes_f32(int64_t n_tgt, int acc64) {
    return (int64_t)(SLICES_PER_LAUNCH + 2) * n_tgt * (int64_t)(acc64 ? sizeof(double4) : sizeof(float4));
}

}  // extern "C"
