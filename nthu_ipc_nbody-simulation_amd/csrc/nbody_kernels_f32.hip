// nbody_kernels_f32.hip — K1: LDS-tiled all-pairs force accumulation in fp32 with the kick-drift update fused
// into its epilogue, hand-written for gfx950 (CDNA4, wave64).
//
// Replaces the reference's compute_accelerations_gpu (hw5.cu:159-215: one thread per (i,j) pair, three global
// fp64 atomics per pair) + update_positions_gpu (hw5.cu:231-239) + clear_a_gpu (hw5.cu:224-229), i.e. the
// accel/kick/drift phases of run_step (samples/nbody.cc:56-88), for large synthetic N.
//
// Design:
//  * owner-computes: a lane owns R whole target bodies (R = 1,2,4) in registers; no atomics, no global `a`
//    array, bitwise run-to-run reproducible.
//  * sources stream through LDS in tiles of 256 float4 {x,y,z,G*m}: every lane issues ONE coalesced 16-byte
//    global load per tile (a wave reads 1 KiB contiguous), the tile is double-buffered so one s_barrier per
//    tile suffices and the next tile's HBM/L2 load is in flight while the current one is consumed.
//  * the inner loop reads the tile with wave-uniform (broadcast) ds_read_b128, conflict-free, and feeds R
//    interactions per read: 3 sub, 3 fma, v_rsq_f32, 3 mul, 3 fma = 12 VALU + 1 transcendental per pair.
//    Measured issue costs on MI355X (profiles/r01_ubench_valu_rate.txt): fp32 VALU 2 cycles per wave64
//    instruction per SIMD with >= 2 resident waves, v_rsq_f32 8 cycles, no overlap between them -> floor of
//    32 SIMD-cycles per 64 pairs = 62 % of the 157.3 TFLOP/s fp32 peak in the 20-flop/pair convention.
//  * MFMA deliberately unused: the only GEMM-shaped reformulation (sum_j s_ij x_j - x_i sum_j s_ij) cancels
//    catastrophically for close pairs; the loop is rsqrt-bound VALU work.
//  * every workgroup streams the whole source array at the same pace, so a tile is fetched from HBM /
//    Infinity Cache once per XCD and then served by that XCD's L2 to its other ~127 resident workgroups:
//    HBM traffic ~ 8 x 16 B x N per step; no XCD-aware block remap is needed (all blocks share all sources).
//  * kick-drift fused: v += a*dt ; q_new = q + v*dt written to the OTHER position array (ping-pong), because
//    other workgroups still read the old positions — the barrier the reference gets from its kernel boundary
//    between hw5.cu:371 and :375.
//  * NB_F32_ACC64: fp32 pair arithmetic; each tile's 256 partial sums are added into fp64 accumulators
//    (3R v_add_f64 per 256R pairs) and q,v masters are integrated in fp64.
#include "nbody_kernels.h"

namespace nbk {

template <int R, bool ACC64, bool ACCEL_ONLY>
__global__ __launch_bounds__(WG, (R == 4 ? 4 : 8)) void nbody_force_f32(F32Args a) {
    __shared__ float4 tile[2][TILE];
    const int t = threadIdx.x;
    const long base = (long)blockIdx.x * (WG * R);

    float xi[R], yi[R], zi[R], gmi[R];
    float ax[R], ay[R], az[R];
    double dax[R], day[R], daz[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        long i = base + (long)r * WG + t;
        long ic = i < a.n_tgt ? i : a.n_tgt - 1;  // clamp: tail lanes recompute the last body, never store
        float4 p = a.src[a.tgt_off + ic];
        xi[r] = p.x; yi[r] = p.y; zi[r] = p.z; gmi[r] = p.w;
        ax[r] = ay[r] = az[r] = 0.f;
        dax[r] = day[r] = daz[r] = 0.0;
    }

    const long ntiles = (a.n_src + TILE - 1) / TILE;
    const float eps2 = a.eps2;
    auto load_src = [&](long k) -> float4 {
        long j = k * TILE + t;
        // bodies past the end are massless points at the origin: with eps2 > 0 they add exactly +0
        return j < a.n_src ? a.src[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    };

    tile[0][t] = load_src(0);
    __syncthreads();

    for (long k = 0; k < ntiles; ++k) {
        const int cur = (int)(k & 1);
        float4 nxt;
        if (k + 1 < ntiles) nxt = load_src(k + 1);  // in flight while tile k is consumed

#pragma unroll 8
        for (int j = 0; j < TILE; ++j) {
            const float4 s = tile[cur][j];  // wave-uniform address: broadcast ds_read_b128
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float dx = s.x - xi[r];
                float dy = s.y - yi[r];
                float dz = s.z - zi[r];
                float r2 = __builtin_fmaf(dx, dx, eps2);
                r2 = __builtin_fmaf(dy, dy, r2);
                r2 = __builtin_fmaf(dz, dz, r2);
                float rinv = __builtin_amdgcn_rsqf(r2);  // v_rsq_f32, 1 ulp
                float rinv2 = rinv * rinv;
                float sc = s.w * rinv;
                sc = sc * rinv2;  // G*m_j / (r2+eps2)^(3/2) ; the self pair gives sc*0 = +0
                ax[r] = __builtin_fmaf(dx, sc, ax[r]);
                ay[r] = __builtin_fmaf(dy, sc, ay[r]);
                az[r] = __builtin_fmaf(dz, sc, az[r]);
            }
        }
        if (ACC64) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                dax[r] += (double)ax[r]; day[r] += (double)ay[r]; daz[r] += (double)az[r];
                ax[r] = ay[r] = az[r] = 0.f;
            }
        }
        if (k + 1 < ntiles) tile[cur ^ 1][t] = nxt;  // buffer cur^1 was last read before the previous barrier
        __syncthreads();
    }

#pragma unroll
    for (int r = 0; r < R; ++r) {
        long i = base + (long)r * WG + t;
        if (i >= a.n_tgt) continue;
        if (ACCEL_ONLY) {
            if (ACC64) ((double4*)a.acc)[i] = make_double4(dax[r], day[r], daz[r], 0.0);
            else ((float4*)a.acc)[i] = make_float4(ax[r], ay[r], az[r], 0.f);
        } else if (ACC64) {
            // kick then drift on the fp64 masters (samples/nbody.cc:76-88), fp32 copy for the next step's sources
            const double dt = (double)a.dt;
            double4 v = a.vel64[i];
            double4 p = a.pos64[i];
            v.x += dax[r] * dt; v.y += day[r] * dt; v.z += daz[r] * dt;
            p.x += v.x * dt; p.y += v.y * dt; p.z += v.z * dt;
            a.vel64[i] = v;
            a.pos64[i] = p;
            a.out[a.tgt_off + i] = make_float4((float)p.x, (float)p.y, (float)p.z, gmi[r]);
        } else {
            const float dt = a.dt;
            float4 v = a.vel[i];
            v.x = __builtin_fmaf(ax[r], dt, v.x);
            v.y = __builtin_fmaf(ay[r], dt, v.y);
            v.z = __builtin_fmaf(az[r], dt, v.z);
            a.vel[i] = v;
            a.out[a.tgt_off + i] = make_float4(__builtin_fmaf(v.x, dt, xi[r]), __builtin_fmaf(v.y, dt, yi[r]),
                                               __builtin_fmaf(v.z, dt, zi[r]), gmi[r]);
        }
    }
}

template <int R, bool ACC64, bool ACCEL_ONLY>
static int launch_one(const F32Args& a, hipStream_t stream) {
    long per_block = (long)WG * R;
    long blocks = (a.n_tgt + per_block - 1) / per_block;
    if (blocks <= 0 || blocks > 0x7fffffffL) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL((nbody_force_f32<R, ACC64, ACCEL_ONLY>), dim3((unsigned)blocks), dim3(WG), 0, stream, a);
    return (int)hipGetLastError();
}

int launch_f32(const F32Args& a, int R, bool acc64, bool accel_only, hipStream_t stream) {
#define NBK_CASE(RR)                                                                      \
    case RR:                                                                              \
        if (acc64) return accel_only ? launch_one<RR, true, true>(a, stream) : launch_one<RR, true, false>(a, stream); \
        return accel_only ? launch_one<RR, false, true>(a, stream) : launch_one<RR, false, false>(a, stream);
    switch (R) {
        NBK_CASE(1)
        NBK_CASE(2)
        NBK_CASE(4)
    }
#undef NBK_CASE
    return (int)hipErrorInvalidValue;
}

const char* kernel_name_f32(int R, bool acc64, bool accel_only) {
    static const char* names[3][2][2] = {
        {{"nbody_force_f32<1, false, false>", "nbody_force_f32<1, false, true>"},
         {"nbody_force_f32<1, true, false>", "nbody_force_f32<1, true, true>"}},
        {{"nbody_force_f32<2, false, false>", "nbody_force_f32<2, false, true>"},
         {"nbody_force_f32<2, true, false>", "nbody_force_f32<2, true, true>"}},
        {{"nbody_force_f32<4, false, false>", "nbody_force_f32<4, false, true>"},
         {"nbody_force_f32<4, true, false>", "nbody_force_f32<4, true, true>"}}};
    int ri = R == 1 ? 0 : R == 2 ? 1 : 2;
    return names[ri][acc64 ? 1 : 0][accel_only ? 1 : 0];
}

// Enough workgroups to give every SIMD >= 2 waves (the fp32 VALU needs two resident waves per SIMD to issue at
// its full 2-cycle rate) while amortising each LDS read over as many targets as the register file allows.
int auto_targets_per_lane(long n_tgt, int n_cus) {
    long wg2 = 2L * n_cus;
    if (n_tgt >= wg2 * WG * 4) return 4;
    if (n_tgt >= wg2 * WG * 2) return 2;
    return 1;
}

}  // namespace nbk
