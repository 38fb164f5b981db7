// nbody_kernels_f32.hip — K1: all-pairs force accumulation in fp32 (sources broadcast from SGPRs, or staged through an
// LDS tile) with the kick-drift update fused into its epilogue, hand-written for gfx950 (CDNA4, wave64).
//
// Replaces the reference's compute_accelerations_gpu (hw5.cu:159-215: one thread per (i,j) pair, three global
// fp64 atomics per pair) + update_positions_gpu (hw5.cu:231-239) + clear_a_gpu (hw5.cu:224-229), i.e. the
// accel/kick/drift phases of run_step (samples/nbody.cc:56-88), for large synthetic N.
//
// Design:
//  * owner-computes: a lane owns R whole target bodies in registers; no atomics, no global `a` array,
//    bitwise run-to-run reproducible.
//  * two source paths, same arithmetic (template parameter SGPR, chosen by plan_f32, both parity-tested):
//    - LDS: sources stream through LDS in tiles of 256 float4 {x,y,z,G*m}: every lane issues ONE coalesced
//      16-byte global load per tile (a wave reads 1 KiB contiguous); the tile is double-buffered so one
//      s_barrier per tile suffices and the next tile's load is in flight while the current one is consumed;
//      the inner loop reads the tile with wave-uniform (broadcast) ds_read_b128.
//    - SGPR (default): a source is the same for all 64 lanes, so it never needs a vector register or LDS at
//      all: batches of 16 bodies are fetched with scalar loads (s_load_dwordx16, through the scalar cache and
//      L2) into SGPRs and used directly as the broadcast operand of the packed VALU ops
//      (v_pk_add_f32 v, s[n:n+1], v op_sel_hi:[0,1]).  No ds_read and no barrier inside a batch: exactly 12 packed
//      VALU + 2 v_rsq_f32 per 2 pairs (hot block of the bench kernel in the gfx950 dump: 384 v_pk_*, 64 v_rsq_f32,
//      1 v_mov, ~12 SALU, and since round 3 no s_nop — see `interact`).  Measured (bench/ubench/force_variants.hip,
//      profiles/r01_force_variants.txt; bench.py --source-path): 55.5 % of peak vs 52 % for the LDS path.
//  * targets are held two-per-register-pair (ext_vector float2) so the loop is PACKED fp32:
//    per source and target pair 3 v_pk_add, 3 v_pk_fma, 2 v_rsq_f32, 3 v_pk_mul, 3 v_pk_fma
//    = 12 packed VALU + 2 transcendentals per 2 pairs.
//    Why packed (profiles/r01_ubench_valu_rate.txt): a packed op costs 4 SIMD-cycles for 128 lane-ops, a
//    scalar op 2 cycles for 64 — same arithmetic rate — but ONE wave can only issue an instruction every
//    ~5 cycles, so the scalar form needs >= 3 always-ready waves per SIMD to keep the pipe full (measured
//    40-43 cycles per 64 pairs at 4-8 waves) while the packed form gets there with 2 (33 cycles measured,
//    floor 12*2 + 8 = 32).  v_rsq_f32 costs 8 cycles and does not overlap other VALU work: the floor is
//    62 % of the 157.3 TFLOP/s fp32 peak in the 20-flop/pair convention.
//  * summation is two-level: the 256 contributions of a tile are summed in fp32 registers, then the tile's
//    partial is added to the running sum — compensated (Kahan) in NB_F32, in fp64 in NB_F32_ACC64 — so the
//    rounding error does not grow with sqrt(N) (plain fp32 running sums measured 2.6e-5 * sum|a_ij| at N=2^20).
//  * source slices (j-split): the source range is cut into slices and blockIdx.y of a launch walks 16 of them;
//    partial sums go to a workspace and nbody_reduce_update_f32 folds them into a running (compensated) sum, the
//    last fold doing the update.  Why (all measured, same-device A/B in profiles/r01_jsplit_search.txt): at N = 2^20
//    an R = 8 kernel has only 2^20/(64*8) = 2048 waves — two per SIMD, one single round of workgroups that all run
//    for the whole 250 ms.  Slicing multiplies the workgroups: the slice kernel needs no epilogue state (122 VGPRs
//    -> 4 waves per SIMD), the rounds of short workgroups let faster CUs take more work, and each round streams one
//    L2-resident megabyte: 54.9 % (1 slice) -> 57.4 % (2) -> 58.7 % (8) -> 58.9 % (16) of peak.  The same mechanism
//    gives a multi-GPU shard (N/P targets) enough workgroups to fill the chip.
//  * MFMA deliberately unused: the only GEMM-shaped reformulation (sum_j s_ij x_j - x_i sum_j s_ij) cancels
//    catastrophically for close pairs; the loop is rsqrt-bound VALU work.
//  * every workgroup streams the whole source array at the same pace, so a tile is fetched once per XCD and
//    then served by that XCD's L2 / the Infinity Cache to its other resident workgroups; no XCD-aware block
//    remap is needed (all blocks share all sources).
//  * kick-drift fused: v += a*dt ; q_new = q + v*dt written to the OTHER position array (ping-pong), because
//    other workgroups still read the old positions — the barrier the reference gets from its kernel boundary
//    between hw5.cu:371 and :375.
#include "nbody_f32_common.h"

#ifndef NB_K1_PAIR_GROUP
#define NB_K1_PAIR_GROUP 2  // pairs taken two at a time, stage by stage (see `interact`); 1 = pair after pair
#endif
#ifndef NB_K1_WAVES_512
#define NB_K1_WAVES_512 4   // waves per SIMD the sliced 512-thread kernels are compiled for: 4 = two workgroups per CU =
                            // at most 128 VGPRs.  The pair-grouped loop needs 148 when left alone (ONE workgroup per CU,
                            // measured 257 vs 237 ms/step); held to 128 it compiles without scratch and without s_nop.
#endif

namespace nbk {

constexpr int SGPR_BATCH = 8;  // bodies per scalar-load batch (32 SGPRs; two batches live = 64 of the ~100 SGPRs)

// P = target PAIRS per lane (R = 2P targets).  SPLIT: blockIdx.y selects a slice of the sources and the partial
// sums go to a.partial instead of the epilogue.  SGPR: sources via scalar loads instead of the LDS tile.
// WGS = workgroup size.  The SGPR path has no LDS tile to fill, so its workgroup can be as large as the register
// file allows: with ONE workgroup per CU and a barrier per 256 sources all waves of a CU stay within one tile
// of each other and every source line is fetched once per XCD (measured, profiles/
// r01_force_variants_wgsize_traffic.txt: 140 MB per step = 8 XCDs x 16.8 MB, the floor; two independent 256-thread
// workgroups per CU drift apart — oldest-wave-first issue arbitration — and fetch it twice, four fetch it 4x).
template <int P, bool ACC64, bool ACCEL_ONLY, bool SPLIT, bool SGPR, int WGS>
__global__ __launch_bounds__(WGS, (WGS == 1024 ? 4 : WGS == 512 ? (SPLIT ? NB_K1_WAVES_512 : 2) : (P >= 4 ? 2 : 4))) void nbody_force_f32(F32Args a) {
    static_assert(SGPR || WGS == TILE, "the LDS path stages one source per thread");
    __shared__ float4 tile[SGPR ? 1 : 2][SGPR ? 1 : TILE];
    constexpr int WG = WGS;  // shadows nbk::WG inside this kernel
    constexpr int R = 2 * P;
    const int t = threadIdx.x;
    const long base = (long)blockIdx.x * (WG * R);

    // target pair p of this lane = bodies base + (2p)*WG + t and base + (2p+1)*WG + t  (coalesced loads)
    v2f xi[P], yi[P], zi[P], gmi[P];
    v2f ax[P], ay[P], az[P];      // partial sums of the current tile
    v2f sx[P], sy[P], sz[P];      // NB_F32: running sums ...
    v2f cx[P], cy[P], cz[P];      // ... and their Kahan compensation
    double dax[R], day[R], daz[R];  // NB_F32_ACC64: running sums in fp64
#pragma unroll
    for (int p = 0; p < P; ++p) {
        float4 b[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            long i = base + (long)(2 * p + h) * WG + t;
            long ic = i < a.n_tgt ? i : a.n_tgt - 1;  // tail lanes recompute the last body, never store
            b[h] = a.tgt[ic];
        }
        xi[p] = (v2f){b[0].x, b[1].x}; yi[p] = (v2f){b[0].y, b[1].y};
        zi[p] = (v2f){b[0].z, b[1].z}; gmi[p] = (v2f){b[0].w, b[1].w};
        ax[p] = ay[p] = az[p] = splat(0.f);
        sx[p] = sy[p] = sz[p] = cx[p] = cy[p] = cz[p] = splat(0.f);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) dax[r] = day[r] = daz[r] = 0.0;

    const v2f eps2 = splat(a.eps2);

    // one source body against this lane's P target pairs: 12 packed VALU + 2 v_rsq_f32 per pair of targets.
    // The pairs are taken in groups of G = 2, stage by stage (both subtractions, both r2 chains, both rsq, ...): a packed
    // result is then never consumed by the very next VALU instruction.  Pair after pair (rounds 1-2) the compiler left
    // `sc = gm*rinv ; sc *= rinv2` adjacent and padded the gfx950 forwarding hazard with one s_nop per (source, pair) —
    // 32 per 8-source batch; grouped, the hot block has none (hipcc -S: 384 v_pk_*, 64 v_rsq_f32, 0 s_nop).  The grouped
    // loop holds 8 more temporaries: left alone it takes 148 VGPRs, which costs the second workgroup per CU and 8 % of
    // the rate (257 vs 237 ms/step); compiled for 4 waves per SIMD (NB_K1_WAVES_512) it fits 128 without scratch.
    // Same-device A/B of the four combinations (profiles/r03_k1_ab.txt): grouped + 4 waves 237.9 ms/step, pair-after-pair
    // 241.0 (either bound), grouped at 148 VGPRs 256.0 — adopted: +1.3 %.  In the stand-alone harness at equal occupancy
    // (bench/ubench/force_variants.hip ORDER 0/2, profiles/r03_pair_order_ab.txt) the grouping alone is worth +3.1 %;
    // groups of 4 (40 live temporaries) and the rinv2*rinv*gm chain order measured slower than the original.
    auto interact = [&](const float4 s) {
        const v2f qx = splat(s.x), qy = splat(s.y), qz = splat(s.z), gm = splat(s.w);
        constexpr int G = (NB_K1_PAIR_GROUP > 1 && P % 2 == 0) ? 2 : 1;
#pragma unroll
        for (int p0 = 0; p0 < P; p0 += G) {
            v2f dx[G], dy[G], dz[G], r2[G], rinv[G], sc[G];
#define NB_STAGE(stmt) _Pragma("unroll") for (int g = 0; g < G; ++g) { const int p = p0 + g; (void)p; stmt; }
            NB_STAGE(dx[g] = qx - xi[p])
            NB_STAGE(dy[g] = qy - yi[p])
            NB_STAGE(dz[g] = qz - zi[p])
            NB_STAGE(r2[g] = pk_fma(dx[g], dx[g], eps2))
            NB_STAGE(r2[g] = pk_fma(dy[g], dy[g], r2[g]))
            NB_STAGE(r2[g] = pk_fma(dz[g], dz[g], r2[g]))
            NB_STAGE(rinv[g] = ((v2f){__builtin_amdgcn_rsqf(r2[g].x), __builtin_amdgcn_rsqf(r2[g].y)}))  // v_rsq_f32 x2
            NB_STAGE(sc[g] = gm * rinv[g])
            NB_STAGE(rinv[g] = rinv[g] * rinv[g])
            NB_STAGE(sc[g] = sc[g] * rinv[g])  // G*m_j / (r2+eps2)^(3/2) ; the self pair gives sc*0 = +0
            NB_STAGE(ax[p] = pk_fma(dx[g], sc[g], ax[p]))
            NB_STAGE(ay[p] = pk_fma(dy[g], sc[g], ay[p]))
            NB_STAGE(az[p] = pk_fma(dz[g], sc[g], az[p]))
#undef NB_STAGE
        }
    };
    // second summation level: fold the partial of the last <= 256 sources into the running sum
    auto flush = [&]() {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (ACC64) {
                dax[2 * p] += (double)ax[p].x; dax[2 * p + 1] += (double)ax[p].y;
                day[2 * p] += (double)ay[p].x; day[2 * p + 1] += (double)ay[p].y;
                daz[2 * p] += (double)az[p].x; daz[2 * p + 1] += (double)az[p].y;
            } else {  // Kahan: y = part - c ; t = s + y ; c = (t - s) - y ; s = t   (no reassociation: not fast-math)
                v2f y, tt;
                y = ax[p] - cx[p]; tt = sx[p] + y; cx[p] = (tt - sx[p]) - y; sx[p] = tt;
                y = ay[p] - cy[p]; tt = sy[p] + y; cy[p] = (tt - sy[p]) - y; sy[p] = tt;
                y = az[p] - cz[p]; tt = sz[p] + y; cz[p] = (tt - sz[p]) - y; sz[p] = tt;
            }
            ax[p] = ay[p] = az[p] = splat(0.f);
        }
    };

    // this workgroup's slice of the launch's source range [src_begin, src_end), in whole tiles (the last slice takes the
    // ragged end)
    const long ntiles_all = (a.src_end + TILE - 1) / TILE;
    long k0 = a.src_begin / TILE, k1 = ntiles_all;
    if (SPLIT) {
        k0 += ((long)a.slice0 + blockIdx.y) * a.tiles_per_slice;
        k1 = k0 + a.tiles_per_slice < ntiles_all ? k0 + a.tiles_per_slice : ntiles_all;
        if (k0 > k1) k0 = k1;
    }

    if constexpr (SGPR) {
        const long j0 = k0 * TILE;
        const long j1 = k1 * TILE < a.src_end ? k1 * TILE : a.src_end;
        constexpr int U = SGPR_BATCH;
        const long jb = j0 + (j1 - j0) / U * U;  // end of the whole batches
        if (jb > j0) {
            // wave-uniform addresses on a read-only array: the compiler selects s_load_dwordx16; one batch is
            // requested ahead of the one being consumed.  (Two named SGPR sets with the loop unrolled by two — no s_waitcnt +
            // 16 s_mov_b64 at the loop end — measured +2.5 % on an unsliced launch in the harness and +0.2 % on the sliced
            // product launch, below the noise between devices: not adopted, profiles/r03_k1_sgpr_ab.txt.)
            float4 cur[U], nxt[U];
#pragma unroll
            for (int u = 0; u < U; ++u) cur[u] = a.src[j0 + u];
            for (long j = j0; j < jb; j += U) {
                const long jn = (j + U < jb) ? j + U : j0;  // the last prefetch wraps around (harmless re-read)
#pragma unroll
                for (int u = 0; u < U; ++u) nxt[u] = a.src[jn + u];
#pragma unroll
                for (int u = 0; u < U; ++u) interact(cur[u]);
#pragma unroll
                for (int u = 0; u < U; ++u) cur[u] = nxt[u];
                if ((((j - j0) + U) & (TILE - 1)) == 0) {
                    flush();
                    // trip counts are workgroup-uniform: keeps the waves on one L2-resident tile (sliced launch without it:
                    // 239.2 vs 238.9 ms/step, same device, 3 alternations — the barrier stays)
                    __syncthreads();
                }
            }
        }
        for (long j = jb; j < j1; ++j) interact(a.src[j]);  // ragged end, one body at a time
        flush();
    } else {
        auto load_src = [&](long k) -> float4 {
            long j = k * TILE + t;
            // bodies past the end are massless points at the origin: with eps2 > 0 they add exactly +0
            return j < a.src_end ? a.src[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        if (k0 < k1) {
            tile[0][t] = load_src(k0);
            __syncthreads();
        }
        for (long k = k0; k < k1; ++k) {
            const int cur = (int)((k - k0) & 1);
            float4 nxt;
            if (k + 1 < k1) nxt = load_src(k + 1);  // in flight while tile k is consumed
#pragma unroll 4
            for (int j = 0; j < TILE; ++j) interact(tile[cur][j]);  // wave-uniform address: broadcast ds_read_b128
            flush();
            if (k + 1 < k1) tile[cur ^ 1][t] = nxt;  // buffer cur^1 was last read before the previous barrier
            __syncthreads();
        }
    }

#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int p = r >> 1, h = r & 1;
        long i = base + (long)r * WG + t;
        if (i >= a.n_tgt) continue;
        if (SPLIT) {
            const long slot = (long)(blockIdx.y + 2) * a.n_tgt + i;  // records 0 and 1 hold the running sum / compensation
            if (ACC64) ((double4*)a.partial)[slot] = make_double4(dax[r], day[r], daz[r], 0.0);
            else ((float4*)a.partial)[slot] = make_float4(sx[p][h], sy[p][h], sz[p][h], 0.f);
        } else if (ACC64) {
            finish_target<true, ACCEL_ONLY>(a, i, dax[r], day[r], daz[r], xi[p][h], yi[p][h], zi[p][h], gmi[p][h]);
        } else {
            finish_target<false, ACCEL_ONLY>(a, i, sx[p][h], sy[p][h], sz[p][h], xi[p][h], yi[p][h], zi[p][h],
                                             gmi[p][h]);
        }
    }
}

// fold the partial sums of one split launch (a.partial[2 + gy][n_tgt]) into the running sum kept in FRONT of them in the
// workspace (records 0 and 1: sum and compensation — at an offset that does not depend on how many slots the caller's
// workspace holds, so the launches of one step may be given different workspace sizes); compensated in fp32, plain in
// fp64; the last fold of a step runs the epilogue instead of storing
template <bool ACC64, bool ACCEL_ONLY>
__global__ __launch_bounds__(WG) void nbody_reduce_update_f32(F32Args a, int gy, int first, int last) {
    const long i = (long)blockIdx.x * WG + threadIdx.x;
    if (i >= a.n_tgt) return;
    const long run_at = i, comp_at = a.n_tgt + i;  // in front of the launch's partial-sum slots
    if (ACC64) {
        double4* ws = (double4*)a.partial;
        double4 run = first ? make_double4(0, 0, 0, 0) : ws[run_at];
        for (int s = 0; s < gy; ++s) {
            const double4 p = ws[(long)(s + 2) * a.n_tgt + i];
            run.x += p.x; run.y += p.y; run.z += p.z;
        }
        if (last) {
            const float4 b = a.tgt[i];
            finish_target<true, ACCEL_ONLY>(a, i, run.x, run.y, run.z, b.x, b.y, b.z, b.w);
        } else {
            ws[run_at] = run;
        }
    } else {
        float4* ws = (float4*)a.partial;
        float4 run = first ? make_float4(0, 0, 0, 0) : ws[run_at];
        float4 c = first ? make_float4(0, 0, 0, 0) : ws[comp_at];
        for (int s = 0; s < gy; ++s) {  // Kahan, like the in-kernel second level
            const float4 p = ws[(long)(s + 2) * a.n_tgt + i];
            float y, t;
            y = p.x - c.x; t = run.x + y; c.x = (t - run.x) - y; run.x = t;
            y = p.y - c.y; t = run.y + y; c.y = (t - run.y) - y; run.y = t;
            y = p.z - c.z; t = run.z + y; c.z = (t - run.z) - y; run.z = t;
        }
        if (last) {
            const float4 b = a.tgt[i];
            finish_target<false, ACCEL_ONLY>(a, i, run.x, run.y, run.z, b.x, b.y, b.z, b.w);
        } else {
            ws[run_at] = run;
            ws[comp_at] = c;
        }
    }
}

template <int P, bool ACC64, bool ACCEL_ONLY, bool SGPR, int WGS>
static int launch_one(const F32Args& a0, int js, hipStream_t stream) {
    const long per_block = (long)WGS * 2 * P;
    const long blocks = (a0.n_tgt + per_block - 1) / per_block;
    if (blocks <= 0 || blocks > 0x7fffffffL) return (int)hipErrorInvalidValue;
    F32Args a = a0;
    if (!a.tgt) a.tgt = a.src + a.tgt_off;
    if (a.src_begin == 0 && a.src_end == 0) a.src_end = a.n_src;
    if (a.src_begin % TILE || a.src_begin < 0 || a.src_end > a.n_src || a.src_begin > a.src_end) return (int)hipErrorInvalidValue;
    if (js <= 1 && a.phase == F32_PHASE_WHOLE) {
        hipLaunchKernelGGL((nbody_force_f32<P, ACC64, ACCEL_ONLY, false, SGPR, WGS>), dim3((unsigned)blocks), dim3(WGS), 0,
                           stream, a);
        return (int)hipGetLastError();
    }
    // source slices and/or a step cut into phases: partial sums go through the workspace
    if (!a.partial) return (int)hipErrorInvalidValue;
    if (js < 1) js = 1;
    const bool starts = a.phase == F32_PHASE_WHOLE || a.phase == F32_PHASE_FIRST;
    const bool ends = a.phase == F32_PHASE_WHOLE || a.phase == F32_PHASE_LAST;
    const long ntiles = (a.src_end - a.src_begin + TILE - 1) / TILE;
    a.tiles_per_slice = (ntiles + js - 1) / js;
    if (a.tiles_per_slice < 1) a.tiles_per_slice = 1;  // an empty range still runs its reducer (phase bookkeeping)
    const long rblocks = (a.n_tgt + WG - 1) / WG;
    if (a.slots <= 0) a.slots = SLICES_PER_LAUNCH;
    if (a.slots > MAX_SLICES_PER_LAUNCH) a.slots = MAX_SLICES_PER_LAUNCH;
    for (int s0 = 0; s0 < js; s0 += a.slots) {  // `slots` slices per launch; the running sum carries across launches
        const int gy = js - s0 < a.slots ? js - s0 : a.slots;
        a.slice0 = s0;
        hipLaunchKernelGGL((nbody_force_f32<P, ACC64, ACCEL_ONLY, true, SGPR, WGS>), dim3((unsigned)blocks, (unsigned)gy),
                           dim3(WGS), 0, stream, a);
        if (hipError_t e = hipGetLastError()) return (int)e;
        hipLaunchKernelGGL((nbody_reduce_update_f32<ACC64, ACCEL_ONLY>), dim3((unsigned)rblocks), dim3(WG), 0, stream, a, gy,
                           (int)(starts && s0 == 0), (int)(ends && s0 + gy >= js));
        if (hipError_t e = hipGetLastError()) return (int)e;
    }
    return (int)hipSuccess;
}

template <int P, bool SGPR, int WGS>
static int launch_p(const F32Args& a, int js, bool acc64, bool accel_only, hipStream_t stream) {
    if (acc64) return accel_only ? launch_one<P, true, true, SGPR, WGS>(a, js, stream) : launch_one<P, true, false, SGPR, WGS>(a, js, stream);
    return accel_only ? launch_one<P, false, true, SGPR, WGS>(a, js, stream) : launch_one<P, false, false, SGPR, WGS>(a, js, stream);
}

// valid (source path, workgroup size, targets per lane) combinations; anything else is refused
static int launch_f32_sym_batched_or_whole(const F32Args& a, const F32Plan& plan, bool acc64, bool accel_only, hipStream_t stream) {
    if (plan.sym_batches.count > 1) return launch_f32_sym_batched(a, plan.sym_batches, plan.sym_cus, acc64, accel_only ? 1 : 0, stream);
    return launch_f32_sym(a, plan.sym, acc64, accel_only ? 1 : 0, stream);
}

int launch_f32(const F32Args& a, const F32Plan& plan, bool acc64, bool accel_only, hipStream_t stream) {
    if (plan.symmetric)  // K1s: every unordered pair once (in several launches when the system is too large for a slot per round)
        return launch_f32_sym_batched_or_whole(a, plan, acc64, accel_only, stream);
    const int R = plan.targets_per_lane, js = plan.j_split, wg = plan.wg_size;
    if (!plan.sgpr_sources) {
        if (wg != 256) return (int)hipErrorInvalidValue;
        if (R == 2) return launch_p<1, false, 256>(a, js, acc64, accel_only, stream);
        if (R == 4) return launch_p<2, false, 256>(a, js, acc64, accel_only, stream);
        if (R == 8) return launch_p<4, false, 256>(a, js, acc64, accel_only, stream);
        return (int)hipErrorInvalidValue;
    }
    if (wg == 1024 && R == 4) return launch_p<2, true, 1024>(a, js, acc64, accel_only, stream);
    if (wg == 512 && R == 8) return launch_p<4, true, 512>(a, js, acc64, accel_only, stream);
    if (wg == 256 && R == 2) return launch_p<1, true, 256>(a, js, acc64, accel_only, stream);
    if (wg == 256 && R == 4) return launch_p<2, true, 256>(a, js, acc64, accel_only, stream);
    if (wg == 256 && R == 8) return launch_p<4, true, 256>(a, js, acc64, accel_only, stream);
    return (int)hipErrorInvalidValue;
}

const char* kernel_name_f32(const F32Plan& plan, bool acc64, bool accel_only) {
    static char buf[8][112];
    static int slot = 0;
    char* s = buf[slot++ & 7];
    if (plan.symmetric) {
        snprintf(s, 112, "nbody_force_sym_f32<%s>", acc64 ? "true" : "false");
        return s;
    }
    snprintf(s, 112, "nbody_force_f32<%d, %s, %s, %s, %s, %d>", plan.targets_per_lane / 2, acc64 ? "true" : "false",
             accel_only ? "true" : "false", plan.j_split > 1 ? "true" : "false", plan.sgpr_sources ? "true" : "false",
             plan.wg_size);
    return s;
}

// Workgroup shape, register blocking and source slicing for n_tgt targets against n_src sources on n_cus CUs
// (measured: profiles/r01_force_variants*.txt, profiles/r01_jsplit_search.txt, bench.py --targets-per-lane/--j-split/--wg-size):
//  * with a workspace: 512-thread workgroups, R = 8 targets per lane (4 packed pairs, ~150 VGPRs, one workgroup =
//    2 waves per SIMD per CU) and source slices of <= 65 536 bodies (1 MiB, L2-resident): 58.9 % of peak at N = 2^20;
//    the slice count is raised further when the targets alone cannot give every CU a workgroup (multi-GPU shards);
//  * without a workspace (no partial sums possible): one 1024-thread workgroup per CU, R = 4, whole source range,
//    a barrier per 256 sources to keep the CU's waves on the same lines (55.6 %);
//  * small systems: 256-thread workgroups, R = 4 or 2;  LDS path: 256 threads (one source per thread per tile).
// Source slices of a small system or shard (n_tgt < 131072: 256-thread workgroups of 1024 targets, SGPR-fed), round 5.
// Every slice count 1..64 was measured at 31 sizes from 1024 to 126976 bodies (bench/k1_small_n_model.py,
// profiles/r05_k1_small_n_model.txt) and a two-parameter model of the launch reproduces the best choice within 9 % at every one
// of them (exactly at 26): a workgroup that meets t source tiles costs 0.66 + t tile-times (prologue + epilogue = 0.66 of a
// tile), and a CU that carries k such workgroups at once runs each 1 + 0.7 (k - 1) times slower — an extra co-resident
// workgroup is cheaper than a second round, a half-filled extra round is not.  Rounds 2-4 took 16 slices at all of these
// sizes: up to 50 % slower at 6144-10240 and 20480-26624 bodies, 2-3x slower below 4096.  `max_slices`: what one launch's
// workspace holds (a second launch is outside the model).
#ifndef NB_K1_SLICE_MODEL
#define NB_K1_SLICE_MODEL 1  // 0: rounds 2-5's rule (16 slices / a power of two) — the A/B build of bench/debug/acc64_small_n_plan_ab.py
#endif
static long small_system_slices(long bx, long ntiles, int n_cus, long max_slices) {
    long best_js = 1;
    double best = 1e300;
    for (long js = 1; js <= max_slices && js <= ntiles; ++js) {
        const long t = (ntiles + js - 1) / js, eff = (ntiles + t - 1) / t;
        if (eff != js) continue;  // the same cut as `eff` slices with empty workgroups behind it
        const long k = (bx * eff + n_cus - 1) / n_cus;
        const double cost = (1.0 + 0.7 * (double)(k - 1)) * (0.66 + (double)t);
        if (cost < best) {  // ties: the smaller count
            best = cost;
            best_js = js;
        }
    }
    return best_js;
}

F32Plan plan_f32(long n_tgt, long n_src, int n_cus, int force_tpl, int force_js, bool have_workspace, int source_path,
                 int force_wg, int max_slices) {
    F32Plan p;
    p.sgpr_sources = source_path != 1;
    const long ntiles = (n_src + TILE - 1) / TILE;
    int wg = force_wg, R = force_tpl;
    if (!p.sgpr_sources) wg = 256;
    if (wg != 256 && wg != 512 && wg != 1024) wg = 0;
    if (R != 2 && R != 4 && R != 8) R = 0;
    const bool many = p.sgpr_sources && n_tgt >= 4096L * 32;  // enough targets for 4096-target workgroups (N = 32768: 512x8 2.1e12, 256x4 3.9e12 pairs/s)
    if (wg == 0 && R == 0) {
        if (many && have_workspace) { wg = 512; R = 8; }
        else if (many && n_tgt >= 4096L * n_cus) { wg = 1024; R = 4; }
        else { wg = 256; R = n_tgt < 1024 ? 2 : 4; }
    } else if (wg == 0) {
        wg = (many && R == 8) ? 512 : (many && R == 4 && n_tgt >= 4096L * n_cus) ? 1024 : 256;
    } else if (R == 0) {
        R = wg == 1024 ? 4 : wg == 512 ? 8 : (n_tgt < 1024 ? 2 : 4);
        if (!p.sgpr_sources && n_tgt >= 4096L * 8 && have_workspace) R = 8;  // LDS path, sliced: 55.3 % vs 54.4 % for R = 4
    }
    if ((wg == 1024 && R != 4) || (wg == 512 && R != 8)) wg = 256;  // beyond 256 threads only (1024,4) and (512,8) exist
    p.wg_size = wg;
    p.targets_per_lane = R;
    const long bx = (n_tgt + (long)wg * R - 1) / ((long)wg * R);
    long js = force_js;
    if (js <= 0) {
        js = 1;
        if (have_workspace) {
            // locality: slices of <= 512 tiles (131 072 bodies = 2 MiB of float4, half an XCD's L2)
            while (js * 512 < ntiles && js < MAX_JSPLIT) js <<= 1;
            // granularity: at least eight rounds of 512-thread workgroups (two fit a CU at 122 VGPRs), or the equivalent
            // for 256-thread ones — N = 2^18: 4 slices 55.3 %, 16 slices 58.2 %; N = 2^20: 2 -> 57.4 %, 8 -> 58.7 %
            const long want = 8 * (wg >= 512 ? 1L : 2L) * n_cus;
            while (bx * js < want && js < MAX_JSPLIT && js * 2 * 8 <= ntiles) js <<= 1;  // keep >= 8 tiles per slice
            // small systems (up to 2^16 bodies: too few workgroups either way): ONE launch of 16 slices + one reducer
            // measured best at every size — N = 4096 (one tile per slice): 0.0388 ms/step vs 0.0516 at 8 slices; 8192: 0.062
            // vs 0.071 at 32; 16384: 0.102 / 0.116; 65536: 1.008 / 1.025 (profiles/r02_small_n_jsplit.txt).  Below 16 tiles
            // a slice keeps at least two of them.
            if (n_tgt < 4096L * 32) {
                while (js < 16 && js * 2 <= ntiles && (ntiles >= 16 || js * 2 * 2 <= ntiles)) js <<= 1;
                if (n_src < 4096L * 32 && js > SLICES_PER_LAUNCH) js = SLICES_PER_LAUNCH;  // (measured for n_src = n_tgt)
                // the SGPR path: the measured model above instead of the fixed 16 (whole systems) / powers of two (shards: a
                // rank of 2, 4, 8 of 32768 .. 524288 bodies measured the same way, the same two parameters within 3 % of the best
                // where the old rule was up to 2x off — 65536 / 8: 0.30 -> 0.15 ms, 81920 / 4: 0.65 -> 0.41; profiles/
                // r05_k1_small_n_model.txt).  Slices stay L2-sized (<= 512 tiles), which one launch of 64 can do up to 8.4e6 sources.
                const long slots = max_slices >= SLICES_PER_LAUNCH ? max_slices : SLICES_PER_LAUNCH;
                if (NB_K1_SLICE_MODEL && p.sgpr_sources && wg == 256 && n_cus > 0 && ntiles <= 512L * slots)
                    js = small_system_slices(bx, ntiles, n_cus, slots);
            }
        }
    }
    if (js > MAX_JSPLIT) js = MAX_JSPLIT;
    if (js > ntiles) js = ntiles > 0 ? ntiles : 1;
    if (!have_workspace) js = 1;
    p.j_split = (int)js;
    return p;
}

}  // namespace nbk
