// nbody_kernels_f32_sym.hip — K1s: the all-pairs fp32 force with Newton's third law, hand-written for gfx950 (CDNA4,
// wave64): every UNORDERED pair of bodies is evaluated once and serves both of them.
//
// The reference evaluates every ordered pair (samples/nbody.cc:57-73: for i, for j != i; hw5.cu:159-215: a thread per
// (i,j)); so does K1 (nbody_kernels_f32.hip): 12 packed VALU + v_rsq_f32 per pair = 32 SIMD cycles per 64 pairs, an issue
// ceiling of 62 % of the fp32 peak in the 20-flop convention, of which K1 reaches 96 %.  The distance, the rsqrt and the
// inverse cube of a pair are the same for both bodies: evaluating the pair once costs
//     3 sub, 3 fma (r2), v_rsq_f32, 2 mul (rinv^3), 2 mul (x G*m_j, x G*m_i), 3 fma (a_i), 3 fma (a_j)
//   = 16 packed VALU + 2 v_rsq_f32 per two packed pairs = FOUR interactions  ->  20 SIMD cycles per 64 interactions
//   (21.75 with the 14 DPP moves per 8 such sets that rotate the sources: a ceiling of 0.92 of peak).
// The obstacle on a GPU is a_j: the partial sum of a source is spread over the lanes that hold its partners (the reference
// throws atomics at it, hw5.cu:211-213).  Here it is solved systolically inside a wave, with no atomics anywhere:
//   * a lane owns P = 4 packed pairs of TARGETS in registers (as in K1) and one packed pair of SOURCES that travels:
//     {x,y,z,G*m} and the three accumulators of the travelling pair move one lane per step (v_mov_b32_dpp wave_ror:1 —
//     14 moves per step against 8 x 16 packed ops), so after 64 steps every lane's 8 targets have met the wave's 128
//     sources, and every source's accumulator is back in its home lane holding the sum over the wave's 512 targets;
//   * per step and target pair two packed sets: (iA,j0),(iB,j1) and — the travelling pair's halves swapped by op_sel,
//     which costs no instruction — (iA,j1),(iB,j0); the two sets advance stage by stage so that no packed result feeds
//     the very next instruction (hot block of the gfx950 dump: 128 v_pk_*, 16 v_rsq_f32, 14 v_mov_b32_dpp, 2 s_nop);
//   * a workgroup (8 waves) owns a SUPERBLOCK of SB = 4096 bodies.  It meets another superblock in 32 phases, wave w
//     taking 128-source tile (phase + 4w) mod 32, and adds its travelling sums into an LDS image of that superblock's
//     accelerations by plain read-modify-write: the tiles of one phase are distinct, phases are separated by a barrier,
//     so the order of the additions — and with it every bit of the result — is fixed (bitwise reproducible, like K1);
//   * superblock pairs: I-superblock b takes J = b + r (mod B) for r = 1 .. (B-1)/2, plus r = B/2 for b < B/2 when B is
//     even — every unordered pair of superblocks once — and its own diagonal block without the symmetric half.  The
//     tile phases of these 1 + rounds work units are cut evenly among `chunks` workgroups (a round that straddles two of
//     them leaves its second part in a tail slot).  The LDS image of a finished pair goes to a slot of
//     a partial-sum workspace, the workgroup's own sums (two-level: 128 contributions in fp32 registers, then Kahan /
//     fp64 running sums, as in K1) to another; nbody_reduce_sym_f32 adds the slots of a body in a fixed order and runs
//     the fused kick-drift epilogue (samples/nbody.cc:76-88), or hands the per-GPU partial force to the host's
//     reduce-scatter when several GPUs share the pairs.
//   * consecutive superblocks run on one XCD (workgroups are dealt to the XCDs round-robin, so the block index is
//     remapped): at any time the 32 workgroups of an XCD read a window of 32 consecutive superblocks that slides by one
//     per round — the source reads are L2 hits instead of fabric traffic.
// Measured at N = 2^20 on one MI355X (bench/ubench/sym_force.hip, profiles/r04_sym_force_ubench.txt): 176.6 ms per step
// = 6.2e12 pairs/s = 79 % of the 157.3 TFLOP/s peak at 20 flop per pair, against 234 ms = 59 % for K1; error against the
// fp64 oracle 3.4e-8 * sum|a_ij| (K1: 2.8e-8).
// MFMA stays unused (north star): the loop is rsqrt-bound packed-VALU work; nothing here is a dense contraction.
#include <type_traits>

#include "nbody_f32_common.h"

namespace nbk {

namespace {

__device__ __forceinline__ v2f swp(v2f a) { return __builtin_shufflevector(a, a, 1, 0); }  // folds into op_sel
__device__ __forceinline__ float rot1(float x) {  // lane l <- lane l-1, wave-wide
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x13C /* wave_ror:1 */, 0xf, 0xf, true));
}
__device__ __forceinline__ v2f rot(v2f v) { return (v2f){rot1(v.x), rot1(v.y)}; }

}  // namespace

// rotation steps at which a wave steps down from priority 3 to 2, 1, 0 (a two-wave model of the arbiter puts the optimum near
// 30 / 50 / 61: long first stretch, short last one, so that the wave that finishes first leaves the other only a few steps)
constexpr int SYM_PRIO_STEP_1 = 30, SYM_PRIO_STEP_2 = 50, SYM_PRIO_STEP_3 = 61;
// Second summation level of NB_F32 (the sums over more than one 128-source tile): fp64 registers since round 5 — state, slots
// and outputs stay fp32.  Rounds 1-4 carried Kahan-compensated fp32 pairs there; the same registers as doubles are the
// NB_F32_ACC64 kernel's loop, which measured FASTER than the Kahan one on one box, alternating (168.3 against 169.9 ms per step
// at N = 2^20, profiles/r04b_priority_ab.txt; the round-5 A/B of exactly this switch: profiles/r05_sums64_ab.txt), and a
// fp64 sum of fp32 terms is exact where Kahan is only nearly so.  -DNB_SYM_F32_KAHAN=1 builds the old form for that A/B.
#ifndef NB_SYM_F32_KAHAN
#define NB_SYM_F32_KAHAN 0
#endif
constexpr int P = SYM_P, WGS = SYM_WGS, NW = WGS / 64, R = 2 * P, SB = SYM_SB, NT = SYM_NT;
static_assert(NT == NW * P && (NT & (NT - 1)) == 0, "tiles per superblock");

template <bool ACC64>
__global__ __launch_bounds__(WGS, 1) void nbody_force_sym_f32(F32Args a, F32SymShape sh) {
    constexpr bool SUMS64 = ACC64 || !NB_SYM_F32_KAHAN;  // second-level sums in fp64 registers
    __shared__ float lds[3][SB];  // image of the J-superblock's accelerations (this workgroup's share of them)
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int B = sh.B;
    // block -> (chunk, I-superblock): consecutive superblocks of one chunk share an XCD
    const int G = (int)gridDim.x;
    const int g = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int chunk = g / sh.nb, b = sh.b0 + g % sh.nb;
    const long n = a.n_src;
    const long ibase = (long)b * SB;
    const float4 nobody = make_float4(0.f, 0.f, 0.f, 0.f);  // past the end: massless, adds exactly +0 (eps2 > 0)
    auto body = [&](long i) -> float4 { return i < n ? a.src[i] : nobody; };

    v2f xi[P], yi[P], zi[P], gi[P];
    v2f ax[P], ay[P], az[P];      // sums over the tile in hand
    v2f sx[P], sy[P], sz[P];      // -DNB_SYM_F32_KAHAN: running sums ...
    v2f cx[P], cy[P], cz[P];      // ... and their Kahan compensation
    double dax[R], day[R], daz[R];  // running sums in fp64
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const float4 b0 = body(ibase + (long)(2 * p) * WGS + t), b1 = body(ibase + (long)(2 * p + 1) * WGS + t);
        xi[p] = (v2f){b0.x, b1.x}; yi[p] = (v2f){b0.y, b1.y}; zi[p] = (v2f){b0.z, b1.z}; gi[p] = (v2f){b0.w, b1.w};
        ax[p] = ay[p] = az[p] = sx[p] = sy[p] = sz[p] = cx[p] = cy[p] = cz[p] = splat(0.f);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) dax[r] = day[r] = daz[r] = 0.0;
    const v2f eps2 = splat(a.eps2);

    auto flush = [&]() {  // second summation level, as in K1
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (SUMS64) {
                dax[2 * p] += (double)ax[p].x; dax[2 * p + 1] += (double)ax[p].y;
                day[2 * p] += (double)ay[p].x; day[2 * p + 1] += (double)ay[p].y;
                daz[2 * p] += (double)az[p].x; daz[2 * p + 1] += (double)az[p].y;
            } else {
                v2f y, tt;
                y = ax[p] - cx[p]; tt = sx[p] + y; cx[p] = (tt - sx[p]) - y; sx[p] = tt;
                y = ay[p] - cy[p]; tt = sy[p] + y; cy[p] = (tt - sy[p]) - y; sy[p] = tt;
                y = az[p] - cz[p]; tt = sz[p] + y; cz[p] = (tt - sz[p]) - y; sz[p] = tt;
            }
            ax[p] = ay[p] = az[p] = splat(0.f);
        }
    };

    // one tile of 128 sources (j0 = this lane's body of the first 64, j1 of the second) against the lane's 8 targets:
    // 64 rotation steps.  SYM: the travelling pair collects its half of every interaction too.
    auto tile_pass = [&](const float4 j0, const float4 j1, auto sym, v2f& ajx, v2f& ajy, v2f& ajz) {
        constexpr bool SYM = decltype(sym)::value;
        v2f xj = (v2f){j0.x, j1.x}, yj = (v2f){j0.y, j1.y}, zj = (v2f){j0.z, j1.z}, gj = (v2f){j0.w, j1.w};
        ajx = ajy = ajz = splat(0.f);
        // The SIMD's arbiter serves its waves strictly oldest first: of the two waves that share a SIMD the older one would
        // run its 64 steps at the pace of a lone wave (850 cycles per step, it cannot issue back to back) with the younger
        // one filling its gaps — and the younger one would then finish alone, nobody filling ITS gaps, while the older one
        // waits at the phase barrier: 756 cycles per step and SIMD, measured, instead of the 696 the instructions need
        // (bench/ubench/sym_force variants 8 and 11, profiles/r04_sym_loop_stamps.txt).  Lowering the own priority as the
        // pass advances hands the SIMD to whichever wave is behind: they leapfrog and finish within a few steps of each other.
        int s = 0;
#pragma unroll 1
        for (int seg = 0; seg < 4; ++seg) {
        int s_end = 64;
        if (seg == 0) { __builtin_amdgcn_s_setprio(3); s_end = SYM_PRIO_STEP_1; }
        else if (seg == 1) { __builtin_amdgcn_s_setprio(2); s_end = SYM_PRIO_STEP_2; }
        else if (seg == 2) { __builtin_amdgcn_s_setprio(1); s_end = SYM_PRIO_STEP_3; }
        else __builtin_amdgcn_s_setprio(0);
#pragma unroll 1
        for (; s < s_end; ++s) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                // g = 0: (iA, j0), (iB, j1)        g = 1: (iA, j1), (iB, j0)
                v2f dx[2], dy[2], dz[2], r2[2], rinv[2], r3[2], si[2], sj[2];
                dx[0] = xj - xi[p];      dx[1] = swp(xj) - xi[p];
                dy[0] = yj - yi[p];      dy[1] = swp(yj) - yi[p];
                dz[0] = zj - zi[p];      dz[1] = swp(zj) - zi[p];
#define NB_ST(stmt) _Pragma("unroll") for (int g = 0; g < 2; ++g) { stmt; }
                NB_ST(r2[g] = pk_fma(dx[g], dx[g], eps2))
                NB_ST(r2[g] = pk_fma(dy[g], dy[g], r2[g]))
                NB_ST(r2[g] = pk_fma(dz[g], dz[g], r2[g]))
                NB_ST(rinv[g] = ((v2f){__builtin_amdgcn_rsqf(r2[g].x), __builtin_amdgcn_rsqf(r2[g].y)}))  // v_rsq_f32 x2
                NB_ST(r3[g] = rinv[g] * rinv[g])
                NB_ST(r3[g] = r3[g] * rinv[g])
                si[0] = gj * r3[0];      si[1] = swp(gj) * r3[1];
                if (SYM) { NB_ST(sj[g] = gi[p] * r3[g]) }
                NB_ST(ax[p] = pk_fma(dx[g], si[g], ax[p]))
                NB_ST(ay[p] = pk_fma(dy[g], si[g], ay[p]))
                NB_ST(az[p] = pk_fma(dz[g], si[g], az[p]))
                if (SYM) {  // reaction on the sources; sj[1].x belongs to pair (iA, j1): it goes to the accumulator's .y
                    ajx = pk_fma(-dx[0], sj[0], ajx); ajy = pk_fma(-dy[0], sj[0], ajy); ajz = pk_fma(-dz[0], sj[0], ajz);
                    ajx = pk_fma(-swp(dx[1]), swp(sj[1]), ajx); ajy = pk_fma(-swp(dy[1]), swp(sj[1]), ajy);
                    ajz = pk_fma(-swp(dz[1]), swp(sj[1]), ajz);
                }
#undef NB_ST
            }
            xj = rot(xj); yj = rot(yj); zj = rot(zj); gj = rot(gj);
            if (SYM) { ajx = rot(ajx); ajy = rot(ajy); ajz = rot(ajz); }
        }
        }
    };
    using Yes = std::true_type;
    using No = std::false_type;

    // this workgroup's share of the I-superblock's work: unit 0 = the diagonal block, unit r = round r, NT tile phases
    // each; the phases of all units, in order, are cut into `chunks` equal ranges (the same cut points for every
    // superblock: those with one round fewer simply end earlier).  A round that straddles two workgroups leaves two
    // partial images: the first part in the round's regular slot, the second in the TAIL slot of the workgroup that
    // starts with it — one tail slot per chunk suffices because, for a given chunk, it is the same round for every
    // superblock, so the images land on different J-superblocks of that slot.
    long q, q_hi;
    sym_chunk_range(sh, b, chunk, &q, &q_hi);
    v2f ajx, ajy, ajz;
    bool lds_clean = false;
    while (q < q_hi) {
        int u, ph0, ph1;
        q = sym_piece(q, q_hi, &u, &ph0, &ph1);
        if (u == 0) {  // the superblock against itself, without the symmetric half; the self pair adds exactly +0
            float4 j0 = body(ibase + (long)ph0 * 128 + lane), j1 = body(ibase + (long)ph0 * 128 + 64 + lane);
            for (int k = ph0; k < ph1; ++k) {
                const long nb_ = ibase + (long)((k + 1) & (NT - 1)) * 128;
                const float4 n0 = body(nb_ + lane), n1 = body(nb_ + 64 + lane);  // next tile, in flight during this one
                tile_pass(j0, j1, No{}, ajx, ajy, ajz);
                flush();
                j0 = n0; j1 = n1;
            }
            continue;
        }
        if (!lds_clean) {
#pragma unroll
            for (int k = 0; k < R; ++k) lds[0][k * WGS + t] = lds[1][k * WGS + t] = lds[2][k * WGS + t] = 0.f;
            __syncthreads();
            lds_clean = true;
        }
        const int J = (b + u) % B;
        const long jbase = (long)J * SB;
        const int tile0 = (ph0 + w * P) & (NT - 1);
        float4 j0 = body(jbase + (long)tile0 * 128 + lane), j1 = body(jbase + (long)tile0 * 128 + 64 + lane);
        for (int ph = ph0; ph < ph1; ++ph) {
            const int tile = (ph + w * P) & (NT - 1);
            const long nb_ = jbase + (long)((tile + 1) & (NT - 1)) * 128;
            const float4 n0 = body(nb_ + lane), n1 = body(nb_ + 64 + lane);  // next phase's tile, in flight during this one
            tile_pass(j0, j1, Yes{}, ajx, ajy, ajz);
            flush();
            const int e = tile * 128 + lane;  // distinct tiles per wave within a phase: plain read-modify-write
            lds[0][e] += ajx.x; lds[0][e + 64] += ajx.y;
            lds[1][e] += ajy.x; lds[1][e + 64] += ajy.y;
            lds[2][e] += ajz.x; lds[2][e + 64] += ajz.y;
            j0 = n0; j1 = n1;
            __syncthreads();  // phases must not overlap: the next one touches tiles other waves have just updated
        }
        const int slot = sym_piece_slot(sh, ACC64, b, chunk, u, ph0);
        float* out = (float*)a.partial + (long)slot * 3 * sh.npad + jbase;  // a slot = three planes x, y, z of npad floats
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int e = k * WGS + t;
            out[e] = lds[0][e]; out[sh.npad + e] = lds[1][e]; out[2 * sh.npad + e] = lds[2][e];
            lds[0][e] = lds[1][e] = lds[2][e] = 0.f;  // by the thread that wrote it out: clean for the next pair
        }
        __syncthreads();
    }
    // own sums -> slot `chunk` (NB_F32_ACC64: two slots, the value split into a float and the float of the remainder;
    // NB_F32: the fp64 sum rounded once to the fp32 the slots carry)
    float* own = (float*)a.partial + ibase;
    const long plane = sh.npad;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int p = r >> 1, h = r & 1;
        const long i = (long)r * WGS + t;
        if (ACC64) {
            const float hx = (float)dax[r], hy = (float)day[r], hz = (float)daz[r];
            float* hi = own + (long)(2 * chunk) * 3 * plane + i;
            float* lo = own + (long)(2 * chunk + 1) * 3 * plane + i;
            hi[0] = hx; hi[plane] = hy; hi[2 * plane] = hz;
            lo[0] = (float)(dax[r] - (double)hx); lo[plane] = (float)(day[r] - (double)hy); lo[2 * plane] = (float)(daz[r] - (double)hz);
        } else {
            float* o = own + (long)chunk * 3 * plane + i;
            if (SUMS64) { o[0] = (float)dax[r]; o[plane] = (float)day[r]; o[2 * plane] = (float)daz[r]; }
            else { o[0] = sx[p][h]; o[plane] = sy[p][h]; o[2 * plane] = sz[p][h]; }
        }
    }
}

// add the slots that hold a contribution of THIS launch for body i, in a fixed order, then the epilogue.
// MODE 0: kick-drift; 1: accelerations out; 2: the launch's partial force out (float4 / double4 [n], for a reduce-scatter
// over the GPUs that share the pairs)
// `carry` (float4[n], double4 with ACC64; may alias the MODE 2 output): sums of earlier batches of the same step to start from
template <bool ACC64, int MODE>
__global__ __launch_bounds__(WG) void nbody_reduce_sym_f32(F32Args a, F32SymShape sh, const void* carry) {
    const long i = (long)blockIdx.x * WG + threadIdx.x;
    if (i >= a.n_src) return;
    const int J = (int)(i / SB);
    const float* ws = (const float*)a.partial + i;  // slot s, component c of this body: ws[(3 s + c) * npad]
    const long plane = sh.npad;
    constexpr bool SUMS64 = ACC64 || !NB_SYM_F32_KAHAN;    // the slots of a body are added in fp64 (NB_F32: rounded once at the end)
    double dx = 0, dy = 0, dz = 0;
    float rx = 0, ry = 0, rz = 0, kx = 0, ky = 0, kz = 0;   // -DNB_SYM_F32_KAHAN: compensated fp32
    if (carry) {
        if (ACC64) { const double4 c = ((const double4*)carry)[i]; dx = c.x; dy = c.y; dz = c.z; }
        else { const float4 c = ((const float4*)carry)[i]; rx = c.x; ry = c.y; rz = c.z; dx = c.x; dy = c.y; dz = c.z; }
    }
    auto add = [&](long slot) {
        const float* p = ws + slot * 3 * plane;
        const float x = p[0], y = p[plane], z = p[2 * plane];
        if (SUMS64) { dx += (double)x; dy += (double)y; dz += (double)z; return; }
        float u, v;
        u = x - kx; v = rx + u; kx = (v - rx) - u; rx = v;
        u = y - ky; v = ry + u; ky = (v - ry) - u; ry = v;
        u = z - kz; v = rz + u; kz = (v - rz) - u; rz = v;
    };
    sym_for_each_slot_of(sh, ACC64, J, add);
    if (SUMS64 && !ACC64) { rx = (float)dx; ry = (float)dy; rz = (float)dz; }
    if (MODE == 2) {
        if (ACC64) ((double4*)a.acc)[i] = make_double4(dx, dy, dz, 0.0);
        else ((float4*)a.acc)[i] = make_float4(rx, ry, rz, 0.f);
        return;
    }
    const float4 me = a.src[i];
    if (ACC64) finish_target<true, MODE == 1>(a, i, dx, dy, dz, me.x, me.y, me.z, me.w);
    else finish_target<false, MODE == 1>(a, i, rx, ry, rz, me.x, me.y, me.z, me.w);
}

// kick-drift of targets [tgt_off, tgt_off + n_tgt) from accelerations that arrive in `parts` pieces — a.acc[parts][n_tgt]
// (float4, or double4 with ACC64), added in index order: one piece after an RCCL reduce-scatter, one per GPU when the host
// gathered the GPUs' partial forces with peer copies.  The epilogue of a step whose pairs were shared by several GPUs.
template <bool ACC64>
__global__ __launch_bounds__(WG) void nbody_kick_drift_f32(F32Args a, int parts) {
    const long i = (long)blockIdx.x * WG + threadIdx.x;
    if (i >= a.n_tgt) return;
    const float4 me = a.src[a.tgt_off + i];
    if (ACC64) {
        double x = 0, y = 0, z = 0;
        for (int q = 0; q < parts; ++q) {
            const double4 f = ((const double4*)a.acc)[(long)q * a.n_tgt + i];
            x += f.x; y += f.y; z += f.z;
        }
        finish_target<true, false>(a, i, x, y, z, me.x, me.y, me.z, me.w);
    } else {
#if NB_SYM_F32_KAHAN
        float rx = 0, ry = 0, rz = 0, kx = 0, ky = 0, kz = 0;
        for (int q = 0; q < parts; ++q) {
            const float4 f = ((const float4*)a.acc)[(long)q * a.n_tgt + i];
            float u, v;
            u = f.x - kx; v = rx + u; kx = (v - rx) - u; rx = v;
            u = f.y - ky; v = ry + u; ky = (v - ry) - u; ry = v;
            u = f.z - kz; v = rz + u; kz = (v - rz) - u; rz = v;
        }
#else
        double x = 0, y = 0, z = 0;  // fp64, like every other second-level sum of the fp32 mode; rounded once
        for (int q = 0; q < parts; ++q) {
            const float4 f = ((const float4*)a.acc)[(long)q * a.n_tgt + i];
            x += (double)f.x; y += (double)f.y; z += (double)f.z;
        }
        const float rx = (float)x, ry = (float)y, rz = (float)z;
#endif
        finish_target<false, false>(a, i, rx, ry, rz, me.x, me.y, me.z, me.w);
    }
}

bool plan_symmetric(F32Plan& p, long n_tgt, long n_src, bool whole, size_t workspace_bytes, bool acc64, int n_cus,
                    int source_path, int force_chunks) {
    if (source_path != 0 && source_path != 3) return false;
    if (!whole || n_tgt != n_src || n_src < SYM_MIN_N) return false;
    F32SymBatches kb = sym_batches(n_src, n_cus, acc64);
    // a workspace smaller than the fastest shape wants: the batches that fit what the caller has (memory for speed)
    if (kb.count >= 1 && kb.bytes > workspace_bytes && !force_chunks) kb = sym_batches(n_src, n_cus, acc64, workspace_bytes);
    F32SymShape s = sym_shape(n_src, n_cus, 0, kb.count > 1 ? kb.nb : 0, kb.count > 1 ? 0 : force_chunks);
    if (kb.count == 1) kb.bytes = sym_workspace_bytes(s, acc64);  // (a forced chunk count changes the own and tail slots)
    if (kb.count < 1 || kb.bytes > SYM_MAX_WORKSPACE || workspace_bytes < kb.bytes) return false;
    p.symmetric = true;
    p.sym = s;
    p.sym_batches = kb;
    p.sym_cus = n_cus;
    p.targets_per_lane = 2 * SYM_P;
    p.wg_size = SYM_WGS;
    p.j_split = s.chunks;
    p.sgpr_sources = false;
    return true;
}

// One 512-thread workgroup fits a CU, so workgroups run in rounds of n_cus and a partly filled last round is idle time:
// take the chunk count that minimises  ceil(nb * c / n_cus) / c  (time in units of one superblock's work), with 0.4 % per
// extra chunk for the reloaded targets and the shorter runs between slot writes (measured at N = 2^20: 1 / 2 / 4 / 8 chunks =
// 177.1 / 178.1 / 178.5 / 182.8 ms).  nb = 256: 1;  128: 2;  32 (an eighth of 2^20): 8;  48: 16 (768 workgroups = 3 full
// rounds; 6 would be 288 = one round and an eighth);  366: 2.
static int sym_choose_chunks(int nb, int B, int n_cus, int force_chunks) {
    // a chunk holds at least 8 tile phases (a quarter of a work unit): every chunk then has at most one piece that does not
    // start its round — the one its tail slot is for
    const int c_max = (int)((long)SYM_NT * (1 + B / 2) / SYM_MIN_PHASES);
    int c = force_chunks;
    if (c <= 0) {
        double best = 1e30;
        c = 1;
        for (int k = 1; k <= 64 && k <= c_max; ++k) {
            const long rounds = ((long)nb * k + n_cus - 1) / n_cus;
            const double cost = (double)rounds / k * (1.0 + 0.004 * (k - 1));
            if (cost < best - 1e-12) { best = cost; c = k; }
        }
    }
    if (c > c_max) c = c_max;
    return c < 1 ? 1 : c;
}

F32SymShape sym_shape(long n, int n_cus, int b0, int nb, int force_chunks, int sb) {
    F32SymShape s{};
    s.B = (int)((n + sb - 1) / sb);
    s.npad = (long)s.B * sb;
    if (nb <= 0) { b0 = 0; nb = s.B; }
    s.b0 = b0;
    s.nb = nb;
    s.by_super = nb < s.B;  // several launches (GPUs, batches) share the pairs: nb slots instead of B/2
    s.cus = n_cus > 0 ? n_cus : 256;
    s.chunks = sym_choose_chunks(nb, s.B, s.cus, force_chunks);
    return s;
}

static void fit_chunks(F32SymShape& t, bool acc64, size_t max_bytes) {
    while (max_bytes && t.chunks > 1 && sym_workspace_bytes(t, acc64) > max_bytes) --t.chunks;
}

F32SymShape sym_sub_shape(const F32SymShape& s, int b0, int nb, bool acc64, size_t max_bytes) {
    F32SymShape t = s;
    t.b0 = b0;
    t.nb = nb;
    t.by_super = 1;
    t.chunks = sym_choose_chunks(nb, s.B, s.cus, 0);
    fit_chunks(t, acc64, max_bytes);
    return t;
}

int sym_sub_batch(const F32SymShape& s, bool acc64) {
    const size_t budget = sym_batch_budget(s.npad);
    if (sym_workspace_bytes(s, acc64) <= budget) return s.nb;
    int nb = s.nb;
    while (nb > 16 && sym_workspace_bytes(sym_sub_shape(s, s.b0, nb), acc64) > budget) nb = (nb + 1) / 2;
    return nb;
}

size_t sym_partial_workspace_bytes(const F32SymShape& s, bool acc64) {
    const int nb = sym_sub_batch(s, acc64);
    if (nb >= s.nb) return sym_workspace_bytes(s, acc64);
    return sym_workspace_bytes(sym_sub_shape(s, s.b0, nb), acc64);  // (a shorter last sub-launch is fitted into the same bytes)
}

size_t sym_workspace_bytes(const F32SymShape& s, bool acc64) {
    return (size_t)sym_total_slots(s, acc64) * (size_t)s.npad * 3 * sizeof(float);  // a slot = 3 planes of npad floats
}

static int launch_sym_pass(const F32Args& a, const F32SymShape& sh, bool acc64, int mode, const void* carry, hipStream_t stream) {
    const unsigned G = (unsigned)sh.nb * (unsigned)sh.chunks;
    const unsigned rb = (unsigned)((a.n_src + WG - 1) / WG);
    if (acc64) hipLaunchKernelGGL(nbody_force_sym_f32<true>, dim3(G), dim3(WGS), 0, stream, a, sh);
    else hipLaunchKernelGGL(nbody_force_sym_f32<false>, dim3(G), dim3(WGS), 0, stream, a, sh);
    if (hipError_t e = hipGetLastError()) return (int)e;
#define NB_RED(A, M) hipLaunchKernelGGL((nbody_reduce_sym_f32<A, M>), dim3(rb), dim3(WG), 0, stream, a, sh, carry)
    if (acc64) { if (mode == 0) NB_RED(true, 0); else if (mode == 1) NB_RED(true, 1); else NB_RED(true, 2); }
    else { if (mode == 0) NB_RED(false, 0); else if (mode == 1) NB_RED(false, 1); else NB_RED(false, 2); }
#undef NB_RED
    return (int)hipGetLastError();
}

int launch_f32_sym(const F32Args& a0, const F32SymShape& sh, bool acc64, int mode, hipStream_t stream) {
    if (!a0.src || !a0.partial || a0.n_src <= 0 || sh.B <= 0 || sh.nb <= 0 || sh.chunks <= 0 || mode < 0 || mode > 2)
        return (int)hipErrorInvalidValue;
    if (sh.b0 < 0 || sh.b0 + sh.nb > sh.B) return (int)hipErrorInvalidValue;
    if (mode != 2 && (sh.nb != sh.B || a0.tgt_off != 0 || a0.n_tgt != a0.n_src)) return (int)hipErrorInvalidValue;
    F32Args a = a0;
    a.tgt = a.src;
    const int sub = mode == 2 ? sym_sub_batch(sh, acc64) : sh.nb;
    if (sub >= sh.nb) return launch_sym_pass(a, sh, acc64, mode, nullptr, stream);
    // a partial-force launch whose slots would not fit the budget: sub-launches, each adding to the partial force
    const size_t fit = sym_workspace_bytes(sym_sub_shape(sh, sh.b0, sub), acc64);
    for (int b0 = sh.b0; b0 < sh.b0 + sh.nb; b0 += sub) {
        const int nb = b0 + sub <= sh.b0 + sh.nb ? sub : sh.b0 + sh.nb - b0;
        if (int e = launch_sym_pass(a, sym_sub_shape(sh, b0, nb, acc64, fit), acc64, 2, b0 > sh.b0 ? a.acc : nullptr, stream)) return e;
    }
    return (int)hipSuccess;
}

// ---- one GPU, system too large for a slot per round (B/2 slots of n bodies): the I-superblocks go in BATCHES of `nb`, each a
// launch like one rank of a multi-GPU step (a slot per I-superblock of the batch) whose reducer adds the batch's slots to a
// running force F[n] kept behind the slots in the workspace; the last batch's reducer runs the epilogue from the total.
F32SymBatches sym_batches(long n, int n_cus, bool acc64, size_t budget) {
    F32SymBatches k{};
    if (n < SYM_MIN_N) return k;
    const F32SymShape whole = sym_shape(n, n_cus);
    const size_t whole_budget = budget ? budget : SYM_WHOLE_WORKSPACE;
    const size_t batch_budget = budget ? budget : sym_batch_budget(whole.npad);
    if (sym_workspace_bytes(whole, acc64) <= whole_budget) {
        k.nb = whole.B;
        k.count = 1;
        k.bytes = sym_workspace_bytes(whole, acc64);
        return k;
    }
    // the largest batch of whole rounds of workgroups (a multiple of the CU count; failing that a half, a quarter ... of it)
    // whose slots — of the first and of the smaller last batch, which may be cut into more chunks — fit the budget
    const size_t frec = acc64 ? sizeof(double4) : sizeof(float4);
    auto bytes_of = [&](int nb) {  // (a shorter last batch is fitted into the same bytes: sym_batch_shape)
        return sym_workspace_bytes(sym_shape(n, n_cus, 0, nb, 0), acc64) + (size_t)whole.npad * frec;
    };
    auto take = [&](int nb, size_t limit) {
        if (nb < 1 || nb >= whole.B || bytes_of(nb) > limit) return false;
        k.nb = nb;
        k.count = (whole.B + nb - 1) / nb;
        k.bytes = bytes_of(nb);
        return true;
    };
    for (int nb = (whole.B / n_cus) * n_cus; nb >= n_cus; nb -= n_cus)
        if (take(nb, batch_budget)) return k;
    for (int nb = n_cus / 2; nb >= 16; nb /= 2)
        if (take(nb, batch_budget)) return k;
    return k;  // count == 0: not even 16 superblocks per batch fit
}

F32SymShape sym_batch_shape(long n, int n_cus, const F32SymBatches& kb, int k, bool acc64) {
    const int B = (int)((n + SB - 1) / SB);
    const int b0 = k * kb.nb, nb = b0 + kb.nb <= B ? kb.nb : B - b0;
    F32SymShape sh = sym_shape(n, n_cus, b0, nb, 0);
    sh.by_super = 1;  // (a last batch of all B superblocks cannot happen: count > 1)
    const size_t frec = acc64 ? sizeof(double4) : sizeof(float4);
    fit_chunks(sh, acc64, kb.bytes - (size_t)sh.npad * frec);  // the running force sits behind the slots
    return sh;
}

int launch_f32_sym_batched(const F32Args& a0, const F32SymBatches& kb, int n_cus, bool acc64, int mode, hipStream_t stream) {
    if (!a0.src || !a0.partial || a0.n_src <= 0 || kb.count < 1 || kb.nb < 1 || mode < 0 || mode > 1) return (int)hipErrorInvalidValue;
    if (a0.tgt_off != 0 || a0.n_tgt != a0.n_src) return (int)hipErrorInvalidValue;
    if (kb.count == 1) return launch_f32_sym(a0, sym_shape(a0.n_src, n_cus), acc64, mode, stream);
    const int B = (int)((a0.n_src + SB - 1) / SB);
    F32Args a = a0;
    a.tgt = a.src;
    const size_t frec = acc64 ? sizeof(double4) : sizeof(float4);
    const long npad = (long)B * SB;
    void* F = (char*)a.partial + (kb.bytes - (size_t)npad * frec);  // the running force, behind the slots of the largest batch
    void* user_acc = a.acc;
    for (int k = 0; k < kb.count; ++k) {
        const F32SymShape sh = sym_batch_shape(a.n_src, n_cus, kb, k, acc64);
        const bool last = k == kb.count - 1;
        a.acc = last ? user_acc : F;
        if (int e = launch_sym_pass(a, sh, acc64, last ? mode : 2, k ? F : nullptr, stream)) return e;
    }
    return (int)hipSuccess;
}

int launch_kick_drift_f32(const F32Args& a, bool acc64, int parts, hipStream_t stream) {
    if (!a.src || !a.out || !a.acc || a.n_tgt <= 0 || parts < 1) return (int)hipErrorInvalidValue;
    const unsigned rb = (unsigned)((a.n_tgt + WG - 1) / WG);
    if (acc64) hipLaunchKernelGGL(nbody_kick_drift_f32<true>, dim3(rb), dim3(WG), 0, stream, a, parts);
    else hipLaunchKernelGGL(nbody_kick_drift_f32<false>, dim3(rb), dim3(WG), 0, stream, a, parts);
    return (int)hipGetLastError();
}

// do `n` bodies split over `P` GPUs by index let every GPU own whole superblocks, and is the slot workspace affordable?
bool sym_sharded_ok(long n, int P, int n_cus, bool acc64, F32SymShape* shape_of_rank0) {
    if (P < 2 || n < SYM_MIN_N || n % ((long)P * SYM_SB)) return false;
    // a rank's share of K1s has the same floor as a whole launch (~0.33 ms: a workgroup does at least 8 tile phases) while its K1
    // step shrinks with n^2 / P: measured per rank (bench/shard_pairs_ab.py, profiles/r05_small_n_sym_ab.txt) K1 wins at
    // n = 65536 over 8 and over 4 ranks and at 40960 over 2, K1s at 98304 / 8, 81920 / 4, 49152 / 2: the crossover is n^2 / P ~ 1.1e9
    if ((double)n * (double)n < SYM_SHARE_MIN_N2_PER_RANK * (double)P) return false;
    const int B = (int)(n / SYM_SB);
    const F32SymShape s = sym_shape(n, n_cus, 0, B / P, 0);
    if (sym_partial_workspace_bytes(s, acc64) > SYM_MAX_WORKSPACE) return false;
    if (shape_of_rank0) *shape_of_rank0 = s;
    return true;
}

}  // namespace nbk
