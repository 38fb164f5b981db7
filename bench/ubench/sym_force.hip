// sym_force.hip — experiment harness (measurement tool, not product code): the all-pairs fp32 force with Newton's third
// law, hand-written for gfx950.  The product kernel K1 evaluates every ORDERED pair (12 packed VALU + v_rsq_f32 per pair);
// the distance, the rsqrt and the inverse cube of a pair serve BOTH bodies, so evaluating every UNORDERED pair once costs
//   3 sub, 3 fma, v_rsq_f32, 2 mul (rinv^3), 2 mul (x G*m_j, x G*m_i), 3 fma (a_i), 3 fma (a_j) = 16 VALU + 1 rsq per 2 pairs
// = 8 VALU + 1/2 rsq per ordered pair.  The obstacle on a GPU is a_j: a source's partial sum is spread over the lanes that
// hold its partners.  Here it is solved systolically inside a wave:
//   * a lane owns P packed pairs of targets (registers, as K1) and one packed pair of SOURCES that travels: {x,y,z,G*m}
//     and the accumulators of the travelling pair rotate one lane per step (v_mov_b32_dpp wave_ror:1 — 14 moves per step
//     against 2P x 16 packed ops), so after 64 steps every lane's 2P targets have met the wave's 128 sources and every
//     source's accumulator is back home holding the sum over the wave's 128 P targets;
//   * per step and target pair two packed sets: (iA,j0),(iB,j1) and — operand halves swapped by op_sel, no instruction —
//     (iA,j1),(iB,j0);
//   * a workgroup (NW waves) owns a superblock of SB = 128 P NW bodies; it meets another superblock in NT = P NW phases,
//     wave w taking tile (phase + w P) mod NT, and adds its travelling sums into an LDS image of that superblock's
//     accelerations (plain read-modify-write: tiles are distinct within a phase, phases are separated by a barrier, so the
//     order of the additions — and with it every bit of the result — is fixed);
//   * superblock pairs: workgroup b takes (b, b+r mod B) for r = 1 .. (B-1)/2, plus r = B/2 for b < B/2 when B is even:
//     every unordered pair of superblocks once; the LDS image goes to slot r of a partial-sum workspace, the workgroup's own
//     sums (including the diagonal block, done without the symmetric half) to slot 0; a reducer adds the slots.
// Usage: sym_force [n=1048576] [reps=3] [P=4] [WGS=512] [variant=0] [chunks=1]
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o sym_force sym_force.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f splat(float x) { return (v2f){x, x}; }
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f swp(v2f a) { return __builtin_shufflevector(a, a, 1, 0); }  // folds into op_sel
__device__ __forceinline__ float rot1(float x) {  // lane l <- lane l-1 (wave-wide rotate by one)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x13C, 0xf, 0xf, true));
}
__device__ __forceinline__ v2f rot(v2f v) { return (v2f){rot1(v.x), rot1(v.y)}; }
__device__ __forceinline__ float rowrot1(float x) {  // within 16-lane rows only (timing experiment V = 3: WRONG results)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x121, 0xf, 0xf, true));
}
__device__ __forceinline__ float bperm1(int addr, float x) {  // value of the lane whose number x 4 is `addr`: the LDS crossbar, not the VALU
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, x)));
}
__device__ __forceinline__ v2f bperm(int addr, v2f v) { return (v2f){bperm1(addr, v.x), bperm1(addr, v.y)}; }
__device__ __forceinline__ v2f rowrot(v2f v) { return (v2f){rowrot1(v.x), rowrot1(v.y)}; }

struct SymArgs {
    const float4* src;  // [n] {x,y,z,G*m}
    float4* partial;    // [B/2 + 1][n] slot 0: own workgroup's sums; slot r: from the workgroup r superblocks before
    long n;             // multiple of SB
    int B;              // superblocks
    float eps2;
    unsigned long long* clk;  // V = 8: per wave {cycles inside the rotation loops, cycles of the whole kernel, rotation steps}
};

// V = 0: as the product kernel.  V = 1: the travelling POSITIONS come from an LDS copy of the wave's tile (two ds_read_b128
// per step, issued one step ahead) instead of 8 of the 14 DPP moves; the accumulators still rotate.  V = 2: no second
// summation level (sums stay in the tile registers) and compiled for two workgroups per CU — not a product candidate
// (accuracy), the upper bound of what 4 waves per SIMD would buy.
#ifndef SYM_STAMPS
#define SYM_STAMPS 0  // -DSYM_STAMPS=1: the stamps of V = 8 in every variant
#endif
template <int P, int WGS, int V>
__global__ __launch_bounds__(WGS, V == 2 ? 4 : 1) void sym_force(SymArgs a) {
    constexpr int NW = WGS / 64, R = 2 * P, SB = WGS * R, NT = SB / 128;
    static_assert(NT == NW * P, "tiles per superblock");
    __shared__ float lds[3][SB];
    __shared__ float4 jt[V == 1 ? NW : 1][128];  // V = 1: this wave's tile
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int B = a.B;
    // consecutive superblocks on one XCD (workgroups go to the XCDs round-robin): at round r the 32 workgroups of an XCD
    // read a window of 32 consecutive superblocks that slides by one per round — L2 hits instead of fabric reads
    const int nx = 8;
    const int G = (int)gridDim.x, C = G / B;
    const int g = (G % nx == 0) ? (int)(blockIdx.x % nx) * (G / nx) + (int)(blockIdx.x / nx) : (int)blockIdx.x;
    const int chunk = g / B, b = g % B;
    const long ibase = (long)b * SB;

    v2f xi[P], yi[P], zi[P], gi[P];
    v2f ax[P], ay[P], az[P], sx[P], sy[P], sz[P], cx[P], cy[P], cz[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const float4 b0 = a.src[ibase + (long)(2 * p) * WGS + t], b1 = a.src[ibase + (long)(2 * p + 1) * WGS + t];
        xi[p] = (v2f){b0.x, b1.x}; yi[p] = (v2f){b0.y, b1.y}; zi[p] = (v2f){b0.z, b1.z}; gi[p] = (v2f){b0.w, b1.w};
        ax[p] = ay[p] = az[p] = sx[p] = sy[p] = sz[p] = cx[p] = cy[p] = cz[p] = splat(0.f);
    }
    const v2f eps2 = splat(a.eps2);
    // V = 8: where the time goes — shader-clock stamps around every 64-step rotation loop and around the kernel
    if (V == 9 && w < NW / 2) __builtin_amdgcn_s_setprio(3);  // V = 9: the two waves of a SIMD (w, w + 4) at different priorities
    unsigned long long loop_cycles = 0, loop_steps = 0, k0 = 0;
    if (V == 8 || SYM_STAMPS) k0 = __builtin_amdgcn_s_memtime();
    auto flush = [&]() {
        if (V == 2) return;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            v2f y, tt;
            y = ax[p] - cx[p]; tt = sx[p] + y; cx[p] = (tt - sx[p]) - y; sx[p] = tt;
            y = ay[p] - cy[p]; tt = sy[p] + y; cy[p] = (tt - sy[p]) - y; sy[p] = tt;
            y = az[p] - cz[p]; tt = sz[p] + y; cz[p] = (tt - sz[p]) - y; sz[p] = tt;
            ax[p] = ay[p] = az[p] = splat(0.f);
        }
    };

    // one tile of 128 sources against this lane's 2P targets, 64 rotation steps.  SYM: the sources collect their half too.
    auto tile_pass = [&](const float4 j0, const float4 j1, auto sym, v2f& ajx, v2f& ajy, v2f& ajz) {
        constexpr bool SYM = decltype(sym)::value;
        v2f xj = (v2f){j0.x, j1.x}, yj = (v2f){j0.y, j1.y}, zj = (v2f){j0.z, j1.z}, gj = (v2f){j0.w, j1.w};
        ajx = ajy = ajz = splat(0.f);
        float4 n0 = j0, n1 = j1;
        unsigned long long c0 = 0;
        if (V == 8 || SYM_STAMPS) c0 = __builtin_amdgcn_s_memtime();
        if (V == 1) {  // (wave-private rows of jt: no barrier, the wave runs in lock step)
            jt[w][lane] = j0; jt[w][64 + lane] = j1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            n0 = jt[w][(lane - 1) & 63]; n1 = jt[w][64 + ((lane - 1) & 63)];
        }
#ifndef SYM_ROT_UNROLL
#define SYM_ROT_UNROLL 1  // -DSYM_ROT_UNROLL=2: half the taken branches of the rotation loop (A/B: +0.4 %)
#endif
        // V = 11: the SIMD's arbiter serves its waves strictly oldest first, so of the two waves that share a SIMD the older
        // one runs its 64 steps at the speed of a lone wave while the younger one only fills its gaps — and then finishes
        // alone, with nobody to fill ITS gaps, while the older one waits at the phase barrier.  Lowering the own priority as
        // the pass advances (s_setprio 3, 2, 1, 0 from steps 0, B0, B1, B2) hands the SIMD to whichever wave is behind.
#ifndef SYM_PRIO_B0
#define SYM_PRIO_B0 30
#define SYM_PRIO_B1 50
#define SYM_PRIO_B2 61
#endif
        int s = 0;
#pragma unroll 1
        for (int seg = 0; seg < ((V == 11 || V == 12) ? 4 : 1); ++seg) {
        int s_end = 64;
        if (V == 11 || V == 12) {
            if (seg == 0) { __builtin_amdgcn_s_setprio(3); s_end = SYM_PRIO_B0; }
            else if (seg == 1) { __builtin_amdgcn_s_setprio(2); s_end = SYM_PRIO_B1; }
            else if (seg == 2) { __builtin_amdgcn_s_setprio(1); s_end = SYM_PRIO_B2; }
            else { __builtin_amdgcn_s_setprio(0); }
        }
#pragma unroll SYM_ROT_UNROLL
        for (; s < s_end; ++s) {
            v2f nxj = xj, nyj = yj, nzj = zj, ngj = gj;
            if (V == 12) {  // V = 12: the travelling POSITIONS of the next step come through ds_bpermute_b32, requested now (a whole
                            // step of latency to hide), 8 LDS-pipe instructions instead of 8 of the 14 VALU-pipe DPP moves
                const int from = ((lane - 1) & 63) * 4;
                nxj = bperm(from, xj); nyj = bperm(from, yj); nzj = bperm(from, zj); ngj = bperm(from, gj);
            }
            float4 m0 = n0, m1 = n1;
            if (V == 1) { m0 = jt[w][(lane - s - 2) & 63]; m1 = jt[w][64 + ((lane - s - 2) & 63)]; }  // for step s + 2
            if (V == 5) {
#pragma unroll
                for (int p0 = 0; p0 < P; p0 += 2) {  // FOUR packed sets in flight: pairs p0, p0 + 1, straight and swapped
                    v2f dx[4], dy[4], dz[4], r2[4], rinv[4], r3[4], si[4], sj[4];
#define ST4(stmt) _Pragma("unroll") for (int g = 0; g < 4; ++g) { const int p = p0 + (g >> 1); const bool sw = g & 1; (void)p; (void)sw; stmt; }
                    ST4(dx[g] = (sw ? swp(xj) : xj) - xi[p])
                    ST4(dy[g] = (sw ? swp(yj) : yj) - yi[p])
                    ST4(dz[g] = (sw ? swp(zj) : zj) - zi[p])
                    ST4(r2[g] = pk_fma(dx[g], dx[g], eps2))
                    ST4(r2[g] = pk_fma(dy[g], dy[g], r2[g]))
                    ST4(r2[g] = pk_fma(dz[g], dz[g], r2[g]))
                    ST4(rinv[g] = ((v2f){__builtin_amdgcn_rsqf(r2[g].x), __builtin_amdgcn_rsqf(r2[g].y)}))
                    ST4(r3[g] = rinv[g] * rinv[g])
                    ST4(r3[g] = r3[g] * rinv[g])
                    ST4(si[g] = (sw ? swp(gj) : gj) * r3[g])
                    if (SYM) { ST4(sj[g] = gi[p] * r3[g]) }
                    ST4(ax[p] = pk_fma(dx[g], si[g], ax[p]))
                    ST4(ay[p] = pk_fma(dy[g], si[g], ay[p]))
                    ST4(az[p] = pk_fma(dz[g], si[g], az[p]))
                    if (SYM) {
                        ST4(ajx = sw ? pk_fma(-swp(dx[g]), swp(sj[g]), ajx) : pk_fma(-dx[g], sj[g], ajx))
                        ST4(ajy = sw ? pk_fma(-swp(dy[g]), swp(sj[g]), ajy) : pk_fma(-dy[g], sj[g], ajy))
                        ST4(ajz = sw ? pk_fma(-swp(dz[g]), swp(sj[g]), ajz) : pk_fma(-dz[g], sj[g], ajz))
                    }
#undef ST4
                }
            } else
#pragma unroll
            for (int p = 0; p < P; ++p) {
                // two packed sets advanced stage by stage so that no packed result feeds the very next VALU instruction:
                //   g = 0: (iA, j0), (iB, j1)        g = 1: (iA, j1), (iB, j0) — the travelling pair's halves swapped (op_sel)
                v2f dx[2], dy[2], dz[2], r2[2], rinv[2], r3[2], si[2], sj[2];
                dx[0] = xj - xi[p];      dx[1] = swp(xj) - xi[p];
                dy[0] = yj - yi[p];      dy[1] = swp(yj) - yi[p];
                dz[0] = zj - zi[p];      dz[1] = swp(zj) - zi[p];
#define ST(stmt) _Pragma("unroll") for (int g = 0; g < 2; ++g) { stmt; }
                ST(r2[g] = pk_fma(dx[g], dx[g], eps2))
                ST(r2[g] = pk_fma(dy[g], dy[g], r2[g]))
                ST(r2[g] = pk_fma(dz[g], dz[g], r2[g]))
                if (V == 10) { ST(rinv[g] = r2[g]) }  // timing only: no v_rsq_f32 (WRONG results)
                else { ST(rinv[g] = ((v2f){__builtin_amdgcn_rsqf(r2[g].x), __builtin_amdgcn_rsqf(r2[g].y)})) }
                ST(r3[g] = rinv[g] * rinv[g])
                ST(r3[g] = r3[g] * rinv[g])
                si[0] = gj * r3[0];      si[1] = swp(gj) * r3[1];
                if (SYM) { ST(sj[g] = gi[p] * r3[g]) }
                ST(ax[p] = pk_fma(dx[g], si[g], ax[p]))
                ST(ay[p] = pk_fma(dy[g], si[g], ay[p]))
                ST(az[p] = pk_fma(dz[g], si[g], az[p]))
                if (SYM) {  // sj[1].x belongs to pair (iA, j1): it goes to the accumulator's .y
                    ajx = pk_fma(-dx[0], sj[0], ajx); ajy = pk_fma(-dy[0], sj[0], ajy); ajz = pk_fma(-dz[0], sj[0], ajz);
                    ajx = pk_fma(-swp(dx[1]), swp(sj[1]), ajx); ajy = pk_fma(-swp(dy[1]), swp(sj[1]), ajy);
                    ajz = pk_fma(-swp(dz[1]), swp(sj[1]), ajz);
                }
#undef ST
            }
            if (V == 1) {
                xj = (v2f){n0.x, n1.x}; yj = (v2f){n0.y, n1.y}; zj = (v2f){n0.z, n1.z}; gj = (v2f){n0.w, n1.w};
                n0 = m0; n1 = m1;
            } else if (V == 3) {
                xj = rowrot(xj); yj = rowrot(yj); zj = rowrot(zj); gj = rowrot(gj);
            } else if (V == 4) {  // no rotation at all (timing experiment: WRONG results)
            } else if (V == 12) {
                xj = nxj; yj = nyj; zj = nzj; gj = ngj;
            } else {
                xj = rot(xj); yj = rot(yj); zj = rot(zj); gj = rot(gj);
            }
            if (SYM && V == 3) { ajx = rowrot(ajx); ajy = rowrot(ajy); ajz = rowrot(ajz); }
            else if (SYM && V != 4) { ajx = rot(ajx); ajy = rot(ajy); ajz = rot(ajz); }
        }
        }
        if (V == 8 || SYM_STAMPS) { loop_cycles += __builtin_amdgcn_s_memtime() - c0; loop_steps += 64; }
    };

    using T = std::true_type;
    using F = std::false_type;
    v2f ajx, ajy, ajz;

    // diagonal block: the superblock against itself, without the symmetric half (the self pair adds exactly +0: eps2 > 0)
    for (int k = 0; k < (chunk == 0 ? NT : 0); ++k) {
        const long jb = ibase + (long)k * 128;
        tile_pass(a.src[jb + lane], a.src[jb + 64 + lane], F{}, ajx, ajy, ajz);
        flush();
    }

    const int rounds = (B - 1) / 2 + ((B % 2 == 0 && b < B / 2) ? 1 : 0);
#pragma unroll
    for (int k = 0; k < R; ++k) lds[0][k * WGS + t] = lds[1][k * WGS + t] = lds[2][k * WGS + t] = 0.f;
    __syncthreads();
    for (int r = 1 + chunk * rounds / C; r <= (chunk + 1) * rounds / C; ++r) {
        const int J = (b + r) % B;
        const long jbase = (long)J * SB;
        for (int ph = 0; ph < NT; ++ph) {
            const int tile = (ph + w * P) & (NT - 1);
            const long jb = jbase + (long)tile * 128;
            tile_pass(a.src[jb + lane], a.src[jb + 64 + lane], T{}, ajx, ajy, ajz);
            flush();
            const int e = tile * 128 + lane;
            if (V == 6) {  // timing only: private accumulation, no barrier (WRONG results)
                lds[0][t] += ajx.x + ajx.y + ajy.x + ajy.y + ajz.x + ajz.y;
            } else {
                lds[0][e] += ajx.x; lds[0][e + 64] += ajx.y;
                lds[1][e] += ajy.x; lds[1][e + 64] += ajy.y;
                lds[2][e] += ajz.x; lds[2][e + 64] += ajz.y;
                __syncthreads();
            }
        }
        float4* out = a.partial + (long)r * a.n + jbase;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int e = k * WGS + t;
            out[e] = make_float4(lds[0][e], lds[1][e], lds[2][e], 0.f);
            lds[0][e] = lds[1][e] = lds[2][e] = 0.f;
        }
        __syncthreads();
    }
    float4* own = a.partial + (long)(chunk == 0 ? 0 : B / 2 + chunk) * a.n;  // chunk 0: slot 0, the others behind the rounds
#pragma unroll
    for (int p = 0; p < P; ++p) {
        if (V == 2) { sx[p] = ax[p]; sy[p] = ay[p]; sz[p] = az[p]; }
        own[ibase + (long)(2 * p) * WGS + t] = make_float4(sx[p].x, sy[p].x, sz[p].x, 0.f);
        own[ibase + (long)(2 * p + 1) * WGS + t] = make_float4(sx[p].y, sy[p].y, sz[p].y, 0.f);
    }
    if ((V == 8 || SYM_STAMPS) && lane == 0) {
        unsigned long long* c = a.clk + ((long)blockIdx.x * NW + w) * 3;
        c[0] = loop_cycles; c[1] = __builtin_amdgcn_s_memtime() - k0; c[2] = loop_steps;
    }
}

// a[i] = sum of the slots that hold a contribution for body i (compensated)
__global__ void sym_reduce(const float4* partial, float4* acc, long n, int B, int SB, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int J = (int)(i / SB);
    const int full = (B - 1) / 2;
    const int slots = 1 + full + ((B % 2 == 0 && J >= B / 2) ? 1 : 0);
    float4 run = make_float4(0, 0, 0, 0), c = make_float4(0, 0, 0, 0);
    for (int s = 0; s < slots; ++s) {
        const float4 p = partial[(long)s * n + i];
        float y, t;
        y = p.x - c.x; t = run.x + y; c.x = (t - run.x) - y; run.x = t;
        y = p.y - c.y; t = run.y + y; c.y = (t - run.y) - y; run.y = t;
        y = p.z - c.z; t = run.z + y; c.z = (t - run.z) - y; run.z = t;
    }
    for (int k = 1; k < C; ++k) {  // own sums of the other chunks
        const float4 p = partial[(long)(B / 2 + k) * n + i];
        run.x += p.x; run.y += p.y; run.z += p.z;
    }
    acc[i] = run;
}

static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline double u01(uint64_t i, int k) { return (double)(splitmix64(42 + 7 * i + (uint64_t)k) >> 11) * (1.0 / 9007199254740992.0); }

template <int P, int WGS, int V>
static void run(long n, int reps, int C) {
    constexpr int SB = WGS * 2 * P;
    if (n % SB) { fprintf(stderr, "n must be a multiple of %d\n", SB); exit(2); }
    const int B = (int)(n / SB);
    const int nslots = B / 2 + 1 + C;
    std::vector<float4> h(n);
    const double G = 6.674e-11;
    for (long i = 0; i < n; ++i) {
        const double m = (0.5 + u01(i, 6)) / ((double)n * G);
        h[i] = make_float4((float)(2 * u01(i, 0) - 1), (float)(2 * u01(i, 1) - 1), (float)(2 * u01(i, 2) - 1), (float)(G * m));
    }
    float4 *src, *partial, *acc;
    CK(hipMalloc(&src, n * sizeof(float4)));
    CK(hipMalloc(&partial, (size_t)nslots * n * sizeof(float4)));
    CK(hipMalloc(&acc, n * sizeof(float4)));
    CK(hipMemcpy(src, h.data(), n * sizeof(float4), hipMemcpyHostToDevice));
    CK(hipMemset(partial, 0, (size_t)nslots * n * sizeof(float4)));
    unsigned long long* clk = nullptr;
    const long nwaves = (long)B * C * (WGS / 64);
    if (V == 8 || SYM_STAMPS) { CK(hipMalloc(&clk, nwaves * 3 * sizeof(unsigned long long))); CK(hipMemset(clk, 0, nwaves * 3 * sizeof(unsigned long long))); }
    SymArgs a{src, partial, n, B, 1e-6f, clk};
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    float best = 1e30f, best_red = 0;
    for (int k = 0; k < reps; ++k) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((sym_force<P, WGS, V>), dim3(B * C), dim3(WGS), 0, 0, a);
        CK(hipEventRecord(e1));
        hipLaunchKernelGGL(sym_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, partial, acc, n, B, SB, C);
        CK(hipEventRecord(e2));
        CK(hipEventSynchronize(e2));
        float ms = 0, ms2 = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipEventElapsedTime(&ms2, e1, e2));
        if (ms + ms2 < best + best_red) { best = ms; best_red = ms2; }
        printf("  rep %d: force %.3f ms, reduce %.3f ms\n", k, ms, ms2);
    }
    const double pairs = (double)n * (double)(n - 1);
    printf("V=%d C=%d P=%d WGS=%d SB=%d B=%d slots=%d (workspace %.2f GB): force %.3f ms + reduce %.3f ms -> %.4e pairs/s = %.3f of 157.3 TF at 20 flop/pair\n",
           V, C, P, WGS, SB, B, nslots, (double)nslots * n * 16 / 1e9, best, best_red, pairs / ((best + best_red) * 1e-3),
           pairs * 20 / ((best + best_red) * 1e-3) / 157.3e12);
    if (V == 8 || SYM_STAMPS) {  // per wave: cycles per rotation step inside the loop, and the loop's share of the wave's life
        std::vector<unsigned long long> c(nwaves * 3);
        CK(hipMemcpy(c.data(), clk, c.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::vector<double> per_step, share;
        for (long i = 0; i < nwaves; ++i) if (c[3 * i + 2]) {
            per_step.push_back((double)c[3 * i] / (double)c[3 * i + 2]);
            share.push_back((double)c[3 * i] / (double)c[3 * i + 1]);
        }
        std::sort(per_step.begin(), per_step.end()); std::sort(share.begin(), share.end());
        const size_t m = per_step.size();
        auto q = [&](const std::vector<double>& v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
        hipFuncAttributes fa;
        CK(hipFuncGetAttributes(&fa, (const void*)sym_force<P, WGS, V>));
        printf("  stamps (%zu waves, %d VGPRs): shader-clock cycles per rotation step of one wave, inside the loop: min %.1f  p10 %.1f  p25 %.1f"
               "  median %.1f  p75 %.1f  p90 %.1f  max %.1f  (waves sharing a SIMD x 696 at best);\n"
               "  share of a wave's life inside the rotation loops: min %.4f  median %.4f  max %.4f\n",
               m, fa.numRegs, per_step[0], q(per_step, .1), q(per_step, .25), per_step[m / 2], q(per_step, .75), q(per_step, .9), per_step[m - 1],
               share[0], share[m / 2], share[m - 1]);
        CK(hipFree(clk));
    }
    // check 16 rows against fp64 on the host
    std::vector<float4> out(n);
    CK(hipMemcpy(out.data(), acc, n * sizeof(float4), hipMemcpyDeviceToHost));
    double worst = 0;
    for (int c = 0; c < 16; ++c) {
        const long i = (c * (n / 16) + (c * 37) % (n / 16)) % n;
        double s[3] = {0, 0, 0}, sa = 0;
        for (long j = 0; j < n; ++j) {
            if (j == i) continue;
            const double dx = (double)h[j].x - h[i].x, dy = (double)h[j].y - h[i].y, dz = (double)h[j].z - h[i].z;
            const double r2 = dx * dx + dy * dy + dz * dz + 1e-6;
            const double f = (double)h[j].w / (r2 * std::sqrt(r2));
            s[0] += f * dx; s[1] += f * dy; s[2] += f * dz;
            sa += f * std::sqrt(dx * dx + dy * dy + dz * dz);
        }
        const double e = std::max({std::fabs(out[i].x - s[0]), std::fabs(out[i].y - s[1]), std::fabs(out[i].z - s[2])}) / sa;
        worst = std::max(worst, e);
    }
    printf("  16 rows vs fp64: max err / sum|a_ij| = %.3e\n", worst);
    CK(hipFree(src)); CK(hipFree(partial)); CK(hipFree(acc));
}

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : (1L << 20);
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    const int P = argc > 3 ? atoi(argv[3]) : 4;
    const int wgs = argc > 4 ? atoi(argv[4]) : 512;
    const int V = argc > 5 ? atoi(argv[5]) : 0;
    const int C = argc > 6 ? atoi(argv[6]) : 1;
    if (P == 4 && wgs == 512 && V == 0) run<4, 512, 0>(n, reps, C);
    else if (P == 4 && wgs == 512 && V == 1) run<4, 512, 1>(n, reps, C);
    else if (P == 4 && wgs == 512 && V == 2) run<4, 512, 2>(n, reps, C);
    else if (P == 4 && wgs == 512 && V == 3) run<4, 512, 3>(n, reps, C);
    else if (P == 4 && wgs == 512 && V == 5) run<4, 512, 5>(n, reps, C);
    else if (P == 4 && wgs == 512 && V == 6) run<4, 512, 6>(n, reps, C);
    else if (P == 4 && wgs == 512 && V == 4) run<4, 512, 4>(n, reps, C);
    else if (P == 4 && wgs == 512 && V == 8) run<4, 512, 8>(n, reps, C);
    else if (P == 4 && wgs == 512 && V == 9) run<4, 512, 9>(n, reps, C);
    else if (P == 4 && wgs == 512 && V == 10) run<4, 512, 10>(n, reps, C);
    else if (P == 4 && wgs == 512 && V == 11) run<4, 512, 11>(n, reps, C);
    else if (P == 4 && wgs == 512 && V == 12) run<4, 512, 12>(n, reps, C);
    else if (P == 4 && wgs == 256 && V == 0) run<4, 256, 0>(n, reps, C);
    else if (P == 2 && wgs == 512 && V == 0) run<2, 512, 0>(n, reps, C);
    else if (P == 2 && wgs == 1024 && V == 0) run<2, 1024, 0>(n, reps, C);
    else { fprintf(stderr, "unsupported P/WGS/V\n"); return 2; }
    return 0;
}
