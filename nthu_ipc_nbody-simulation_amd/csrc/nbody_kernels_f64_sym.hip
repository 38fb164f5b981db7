// nbody_kernels_f64_sym.hip — K1s-f64: the fp64 all-pairs force of large systems (NB_F64, n >= 32768) with every UNORDERED
// pair evaluated once — Newton's third law — hand-written for gfx950.  The fp64 sibling of K1s (nbody_kernels_f32_sym.hip);
// replaces, for eps > 0, K1-f64 (nbody_kernels_f64.hip: every ordered pair, 17 fp64 VALU + v_rsq_f64 per pair = 84 SIMD cycles
// per 64 pairs) in run_step's accel phase (samples/nbody.cc:56-74; hw5.cu:159-215 with its three fp64 atomics per pair).
//
// One pair evaluation serves both bodies: 3 add, 3 fma, v_rsq_f64 + 5 (Halley), 2 mul (y^3), 2 mul (x G*m_j, x G*m_i),
// 3 fma (a_i), 3 fma (a_j) = 21 fp64 VALU (4 cycles) + 16 = 100 cycles per 64 pair evaluations = 50 per 64 interactions.
// The scheme is K1s': a lane owns R = 4 targets in registers and ONE source that travels — {x, y, z, G*m} and its three
// accumulators, 7 doubles = 14 v_mov_b32_dpp wave_ror:1 per step, so 57 cycles per 64 interactions with the moves — a
// workgroup (8 waves) owns a superblock of 2048 bodies and meets another in 32 phases of 64-source tiles, adding its
// travelling sums into an fp64 LDS image (48 KB) by plain read-modify-write (distinct tiles per phase, a barrier between
// phases: bitwise reproducible); the pair schedule, the cut into chunks and the slots are K1s' own (sym_chunk_range,
// sym_piece, sym_piece_slot, sym_for_each_slot_of in nbody_kernels.h — the host self-test covers them), with slots of three
// fp64 planes.  G*m_eff of the step comes from nbody_gm_f64 (the device-mass law, nbody.cc:14-16), the reducer sums a body's
// slots and does the non-contracted kick-drift (nbody.cc:76-88) like K1-f64's.
#include <type_traits>

#include "nbody_kernels.h"

namespace nbk {

namespace {

constexpr int R = 4, WGS = 512, NW = WGS / 64, SB = SYM64_SB, NT = SYM_NT;
static_assert(SB == WGS * R && SB / 64 == NT && NT == NW * R, "superblock = 32 tiles of 64 sources, 4 per wave and phase");

__device__ __forceinline__ double rsqrt_fast64(double x) {  // as nbody_kernels_f64.hip: v_rsq_f64 + one Halley step
    double y = __builtin_amdgcn_rsq(x);
    double e = __builtin_fma(-x * y, y, 1.0);
    double t = __builtin_fma(0.375, e, 0.5);
    return __builtin_fma(y * e, t, y);
}
__device__ __forceinline__ double rot(double v) {  // lane l <- lane l-1, wave-wide: two v_mov_b32_dpp wave_ror:1
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x13C, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x13C, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

}  // namespace

__global__ __launch_bounds__(WGS, 1) void nbody_force_sym_f64(F64LargeArgs a, F32SymShape sh) {
    __shared__ double lds[3][SB];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int B = sh.B, n = a.n;
    const int G = (int)gridDim.x;
    const int g = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int chunk = g / sh.nb, b = sh.b0 + g % sh.nb;
    const long ibase = (long)b * SB;
    const double* __restrict__ qx = a.q;
    const double* __restrict__ qy = a.q + n;
    const double* __restrict__ qz = a.q + 2 * (size_t)n;
    const double* __restrict__ gm = a.gm;
    // past the end: a massless body at the origin — adds exactly +0 (eps2 > 0)
    auto X = [&](long i) { return i < n ? qx[i] : 0.0; };
    auto Y = [&](long i) { return i < n ? qy[i] : 0.0; };
    auto Z = [&](long i) { return i < n ? qz[i] : 0.0; };
    auto M = [&](long i) { return i < n ? gm[i] : 0.0; };

    double xi[R], yi[R], zi[R], gi[R], ax[R], ay[R], az[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long i = ibase + (long)r * WGS + t;
        xi[r] = X(i); yi[r] = Y(i); zi[r] = Z(i); gi[r] = M(i);
        ax[r] = ay[r] = az[r] = 0.0;
    }
    const double eps2 = a.eps2;

    // one tile of 64 sources (this lane's: xj..gj) against the lane's 4 targets: 64 rotation steps
    auto tile_pass = [&](double xj, double yj, double zj, double gj, auto sym, double& ajx, double& ajy, double& ajz) {
        constexpr bool SYM = decltype(sym)::value;
        ajx = ajy = ajz = 0.0;
        // waves that share a SIMD are served strictly oldest first; stepping the own priority down as the pass advances lets
        // the wave that is behind catch up instead of finishing alone (see the fp32 kernel, nbody_kernels_f32_sym.hip)
        int s = 0;
#pragma unroll 1
        for (int seg = 0; seg < 4; ++seg) {
        int s_end = 64;
        if (seg == 0) { __builtin_amdgcn_s_setprio(3); s_end = 30; }
        else if (seg == 1) { __builtin_amdgcn_s_setprio(2); s_end = 50; }
        else if (seg == 2) { __builtin_amdgcn_s_setprio(1); s_end = 61; }
        else __builtin_amdgcn_s_setprio(0);
#pragma unroll 1
        for (; s < s_end; ++s) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double dx = xj - xi[r], dy = yj - yi[r], dz = zj - zi[r];
                const double r2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, __builtin_fma(dx, dx, eps2)));
                const double y = rsqrt_fast64(r2);
                const double y3 = y * y * y;
                const double si = gj * y3;
                ax[r] = __builtin_fma(dx, si, ax[r]); ay[r] = __builtin_fma(dy, si, ay[r]); az[r] = __builtin_fma(dz, si, az[r]);
                if (SYM) {
                    const double sj = gi[r] * y3;
                    ajx = __builtin_fma(-dx, sj, ajx); ajy = __builtin_fma(-dy, sj, ajy); ajz = __builtin_fma(-dz, sj, ajz);
                }
            }
            xj = rot(xj); yj = rot(yj); zj = rot(zj); gj = rot(gj);
            if (SYM) { ajx = rot(ajx); ajy = rot(ajy); ajz = rot(ajz); }
        }
        }
    };
    using Yes = std::true_type;
    using No = std::false_type;

    long q, q_hi;
    sym_chunk_range(sh, b, chunk, &q, &q_hi);
    double ajx, ajy, ajz;
    bool lds_clean = false;
    while (q < q_hi) {
        int u, ph0, ph1;
        q = sym_piece(q, q_hi, &u, &ph0, &ph1);
        if (u == 0) {  // the superblock against itself, without the symmetric half; the self pair adds exactly +0
            for (int k = ph0; k < ph1; ++k) {
                const long j = ibase + (long)k * 64 + lane;
                tile_pass(X(j), Y(j), Z(j), M(j), No{}, ajx, ajy, ajz);
            }
            continue;
        }
        if (!lds_clean) {
#pragma unroll
            for (int k = 0; k < R; ++k) lds[0][k * WGS + t] = lds[1][k * WGS + t] = lds[2][k * WGS + t] = 0.0;
            __syncthreads();
            lds_clean = true;
        }
        const int J = (b + u) % B;
        const long jbase = (long)J * SB;
        for (int ph = ph0; ph < ph1; ++ph) {
            const int tile = (ph + w * R) & (NT - 1);
            const long j = jbase + (long)tile * 64 + lane;
            tile_pass(X(j), Y(j), Z(j), M(j), Yes{}, ajx, ajy, ajz);
            const int e = tile * 64 + lane;  // distinct tiles per wave within a phase: plain read-modify-write
            lds[0][e] += ajx; lds[1][e] += ajy; lds[2][e] += ajz;
            __syncthreads();
        }
        const int slot = sym_piece_slot(sh, false, b, chunk, u, ph0);
        double* out = a.sym_slots + (long)slot * 3 * sh.npad + jbase;  // a slot = three fp64 planes of npad bodies
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int e = k * WGS + t;
            out[e] = lds[0][e]; out[sh.npad + e] = lds[1][e]; out[2 * sh.npad + e] = lds[2][e];
            lds[0][e] = lds[1][e] = lds[2][e] = 0.0;
        }
        __syncthreads();
    }
    double* own = a.sym_slots + (long)chunk * 3 * sh.npad + ibase;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long i = (long)r * WGS + t;
        own[i] = ax[r]; own[sh.npad + i] = ay[r]; own[2 * sh.npad + i] = az[r];
    }
}

// MODE 0: kick-drift; 1: accelerations out; 2: this launch's sums out as a running force F[3][npad] (a step that goes in
// batches of superblocks: every batch's reducer adds its slots to F, the last one runs the epilogue from the total).
// `carry` = F of the batches before (may alias the MODE 2 output), or null.
template <int MODE>
__global__ __launch_bounds__(WG) void nbody_reduce_sym_f64(F64LargeArgs a, F32SymShape sh, const double* carry, double* running) {
    const int i = blockIdx.x * WG + threadIdx.x, n = a.n;
    if (i >= n) return;
    const double* ws = a.sym_slots + i;
    const long plane = sh.npad;
    double ax = 0, ay = 0, az = 0;
    if (carry) { ax = carry[i]; ay = carry[plane + i]; az = carry[2 * plane + i]; }
    sym_for_each_slot_of(sh, false, i / SB, [&](long slot) {
        const double* p = ws + slot * 3 * plane;
        ax += p[0]; ay += p[plane]; az += p[2 * plane];
    });
    if (MODE == 2) {
        running[i] = ax; running[plane + i] = ay; running[2 * plane + i] = az;
    } else if (MODE == 1) {
        a.acc_out[i] = ax; a.acc_out[n + i] = ay; a.acc_out[2 * (size_t)n + i] = az;
    } else {  // kick, drift (nbody.cc:76-88), non-contracted like K2 and K1-f64
        const double vx = __dadd_rn(a.v[i], __dmul_rn(ax, a.dt));
        const double vy = __dadd_rn(a.v[n + i], __dmul_rn(ay, a.dt));
        const double vz = __dadd_rn(a.v[2 * (size_t)n + i], __dmul_rn(az, a.dt));
        a.v[i] = vx; a.v[n + i] = vy; a.v[2 * (size_t)n + i] = vz;
        a.qout[i] = __dadd_rn(a.q[i], __dmul_rn(vx, a.dt));
        a.qout[n + i] = __dadd_rn(a.q[n + i], __dmul_rn(vy, a.dt));
        a.qout[2 * (size_t)n + i] = __dadd_rn(a.q[2 * (size_t)n + i], __dmul_rn(vz, a.dt));
    }
}

// (defined in nbody_kernels_f64.hip)
__global__ void nbody_gm_f64(const double* __restrict__ m, const double* __restrict__ coef, double* __restrict__ gm, int n,
                             double fst, double G);

// ---- workspace and batches: as the fp32 kernel's since round 5 (nbody_kernels.h: sym_batch_budget) — one launch with a slot per
// superblock round while that fits 2 GiB (n <= ~6e5: 0.42 GB at n = 2^18), beyond it batches of I-superblocks (a slot per
// superblock of the batch, by_super) within 1440 B per body whose reducers add up a running force F[3][npad] kept behind the
// slots.  Rounds 4's cap of 8 GiB for the one-launch shape (n <= 1.1e6, K1-f64 beyond: 1.5x slower) is gone.
namespace {

constexpr size_t SYM64_BYTES_PER_BODY = 1440;  // 56 slots of 3 doubles + the running force
size_t shape_bytes64(const F32SymShape& s) { return (size_t)sym_total_slots(s, false) * (size_t)s.npad * 3 * sizeof(double); }

struct Sym64Batches {
    int nb = 0, count = 0;  // count == 0: K1s-f64 does not apply
    size_t bytes = 0;
};

Sym64Batches sym64_batches(int n, int n_cus) {
    Sym64Batches k{};
    if (n < SYM64_MIN_SB * SB) return k;
    const F32SymShape whole = sym_shape(n, n_cus, 0, 0, 0, SB);
    if (shape_bytes64(whole) <= SYM_WHOLE_WORKSPACE) {
        k.nb = whole.B;
        k.count = 1;
        k.bytes = shape_bytes64(whole);
        return k;
    }
    size_t budget = SYM64_BYTES_PER_BODY * (size_t)whole.npad;
    if (budget < SYM_WHOLE_WORKSPACE) budget = SYM_WHOLE_WORKSPACE;
    const size_t force = (size_t)whole.npad * 3 * sizeof(double);
    auto take = [&](int nb) {
        if (nb < 1 || nb >= whole.B) return false;
        const size_t b = shape_bytes64(sym_shape(n, n_cus, 0, nb, 0, SB)) + force;
        if (b > budget || b > SYM64_MAX_WORKSPACE) return false;
        k.nb = nb;
        k.count = (whole.B + nb - 1) / nb;
        k.bytes = b;
        return true;
    };
    for (int nb = (whole.B / n_cus) * n_cus; nb >= n_cus; nb -= n_cus)
        if (take(nb)) return k;
    for (int nb = n_cus / 2; nb >= 16; nb /= 2)
        if (take(nb)) return k;
    return k;
}

F32SymShape sym64_batch_shape(int n, int n_cus, const Sym64Batches& kb, int k) {
    const int B = (n + SB - 1) / SB;
    const int b0 = k * kb.nb, nb = b0 + kb.nb <= B ? kb.nb : B - b0;
    F32SymShape sh = sym_shape(n, n_cus, b0, nb, 0, SB);
    sh.by_super = 1;
    const size_t room = kb.bytes - (size_t)sh.npad * 3 * sizeof(double);  // a shorter last batch: no more workgroups per superblock
    while (sh.chunks > 1 && shape_bytes64(sh) > room) --sh.chunks;        // than its slots fit into the same bytes
    return sh;
}

}  // namespace

size_t sym64_workspace_bytes(int n, int n_cus) { return sym64_batches(n, n_cus).bytes; }

int launch_f64_large_sym(const F64LargeArgs& a, int n_cus, hipStream_t stream) {
    if (!a.sym_slots || !a.gm || !(a.eps2 >= F64_EPS2_MIN) || a.n < SYM64_MIN_SB * SB) return (int)hipErrorInvalidValue;
    const Sym64Batches kb = sym64_batches(a.n, n_cus);
    if (kb.count < 1) return (int)hipErrorInvalidValue;
    const unsigned b1 = (unsigned)((a.n + WG - 1) / WG);
    hipLaunchKernelGGL(nbody_gm_f64, dim3(b1), dim3(WG), 0, stream, a.m, a.coef, a.gm, a.n, a.fst, a.G);
    if (kb.count == 1) {
        const F32SymShape sh = sym_shape(a.n, n_cus, 0, 0, 0, SB);
        hipLaunchKernelGGL(nbody_force_sym_f64, dim3((unsigned)(sh.nb * sh.chunks)), dim3(WGS), 0, stream, a, sh);
        if (hipError_t e = hipGetLastError()) return (int)e;
        if (a.acc_out) hipLaunchKernelGGL(nbody_reduce_sym_f64<1>, dim3(b1), dim3(WG), 0, stream, a, sh, (const double*)nullptr, (double*)nullptr);
        else hipLaunchKernelGGL(nbody_reduce_sym_f64<0>, dim3(b1), dim3(WG), 0, stream, a, sh, (const double*)nullptr, (double*)nullptr);
        return (int)hipGetLastError();
    }
    const long npad = (long)((a.n + SB - 1) / SB) * SB;
    double* F = a.sym_slots + (kb.bytes / sizeof(double) - (size_t)npad * 3);  // the running force, behind the slots of the largest batch
    for (int k = 0; k < kb.count; ++k) {
        const F32SymShape sh = sym64_batch_shape(a.n, n_cus, kb, k);
        hipLaunchKernelGGL(nbody_force_sym_f64, dim3((unsigned)(sh.nb * sh.chunks)), dim3(WGS), 0, stream, a, sh);
        if (hipError_t e = hipGetLastError()) return (int)e;
        const double* carry = k ? F : nullptr;
        if (k + 1 < kb.count) hipLaunchKernelGGL(nbody_reduce_sym_f64<2>, dim3(b1), dim3(WG), 0, stream, a, sh, carry, F);
        else if (a.acc_out) hipLaunchKernelGGL(nbody_reduce_sym_f64<1>, dim3(b1), dim3(WG), 0, stream, a, sh, carry, (double*)nullptr);
        else hipLaunchKernelGGL(nbody_reduce_sym_f64<0>, dim3(b1), dim3(WG), 0, stream, a, sh, carry, (double*)nullptr);
        if (hipError_t e = hipGetLastError()) return (int)e;
    }
    return (int)hipSuccess;
}

}  // namespace nbk
