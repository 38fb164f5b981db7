// force_variants.hip — experiment harness for the K1 inner loop (measurement tool, not product code).
//
// Times alternative formulations of the fp32 all-pairs loop on the same synthetic input and checks each against
// variant 0's accelerations, so that a change only moves into csrc/nbody_kernels_f32.hip with a measured reason.
//   V_LDS      the product loop: LDS tile, broadcast ds_read_b128, packed pairs, source-major order
//   V_STAGE    same data path, but U sources x P pairs advanced stage by stage (dependent ops >= U*P apart)
//   V_SMEM     no LDS: sources are wave-uniform, fetched with scalar loads (s_load_dwordx4..16) into SGPRs,
//              software-prefetched one batch ahead (compiler-scheduled)
//   V_SMEM_PF  the same with the s_load / s_waitcnt placed by hand (inline asm, explicit SGPR ping-pong)
//   V_SMEM_AB  V_SMEM with the batch loop unrolled by two over two named SGPR sets A/B (compiler-scheduled loads): each set is
//              reloaded right after it was consumed, one whole batch ahead of its use, and nothing is copied at the loop end
//              (V_SMEM ends every batch with s_waitcnt + 16 s_mov_b64 from the prefetch registers into the working set)
//   template knobs: WGS (workgroup size), SYNC (barrier per 256 sources), STAG (de-phase waves sharing a SIMD),
//   PF (touch-load L2 prefetch distance), js argument of run() (source slices over blockIdx.y)
// Results of every round of experiments: profiles/r01_force_variants*.txt, profiles/r01_jsplit_search.txt
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o force_variants force_variants.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f splat(float x) { return (v2f){x, x}; }
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

constexpr int WG = 256, TILE = 256;
enum { V_LDS = 0, V_STAGE = 1, V_SMEM = 2, V_SMEM_PF = 3, V_SMEM_AB = 4 };
typedef float v16f __attribute__((ext_vector_type(16)));
// 4 bodies (64 B) into 16 SGPRs; completion is NOT tracked by the compiler: pair with sload_wait() before use
__device__ __forceinline__ v16f sload16(const float4* p) { v16f v; asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(v) : "s"(p)); return v; }
__device__ __forceinline__ void sload_wait(v16f& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a)); }
// as above, and additionally ordered AFTER the last write of `dep` (an accumulator of the batch being consumed), so the
// scheduler cannot hoist the wait above the compute it is meant to overlap with
__device__ __forceinline__ void sload_wait_after(v16f& a, v2f& dep) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+v"(dep)); }
__device__ __forceinline__ void sload_wait2(v16f& a, v16f& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b)); }


template <int P>
struct Acc {
    v2f ax[P], ay[P], az[P], sx[P], sy[P], sz[P], cx[P], cy[P], cz[P];
    __device__ void init() {
#pragma unroll
        for (int p = 0; p < P; ++p) ax[p] = ay[p] = az[p] = sx[p] = sy[p] = sz[p] = cx[p] = cy[p] = cz[p] = splat(0.f);
    }
    __device__ void flush() {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            v2f y, tt;
            y = ax[p] - cx[p]; tt = sx[p] + y; cx[p] = (tt - sx[p]) - y; sx[p] = tt;
            y = ay[p] - cy[p]; tt = sy[p] + y; cy[p] = (tt - sy[p]) - y; sy[p] = tt;
            y = az[p] - cz[p]; tt = sz[p] + y; cz[p] = (tt - sz[p]) - y; sz[p] = tt;
            ax[p] = ay[p] = az[p] = splat(0.f);
        }
    }
};

// ORDER (round 3, the s_nop question): how one source meets the lane's P target pairs
//   0  the product loop of rounds 1-2: pair after pair; the compiler leaves `sc = gm*rinv ; sc *= rinv2` adjacent and
//      pads the packed-result forwarding hazard with one s_nop per (source, pair)
//   1  same pair-after-pair order, mul chain rinv2 = rinv*rinv ; r3 = rinv2*rinv ; sc = r3*gm (SGPR operand last)
//   2  pairs in groups of G = 2: every stage of the chain for both pairs before the next stage (dependent packed ops are
//      never adjacent; 16 more temporaries live)
//   3  G = 4: all P pairs stage by stage (40 temporaries)
template <int P, int ORDER = 0>
__device__ __forceinline__ void interact(const float4 s, const v2f* xi, const v2f* yi, const v2f* zi, v2f eps2, Acc<P>& A) {
    const v2f qx = splat(s.x), qy = splat(s.y), qz = splat(s.z), gm = splat(s.w);
    if constexpr (ORDER == 1) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            v2f dx = qx - xi[p], dy = qy - yi[p], dz = qz - zi[p];
            v2f r2 = pk_fma(dx, dx, eps2);
            r2 = pk_fma(dy, dy, r2);
            r2 = pk_fma(dz, dz, r2);
            v2f rinv = (v2f){__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
            v2f r3 = rinv * rinv;
            r3 = r3 * rinv;
            v2f sc = r3 * gm;
            A.ax[p] = pk_fma(dx, sc, A.ax[p]);
            A.ay[p] = pk_fma(dy, sc, A.ay[p]);
            A.az[p] = pk_fma(dz, sc, A.az[p]);
        }
        return;
    }
    if constexpr (ORDER >= 2) {
        constexpr int G = ORDER == 2 ? 2 : 4;
        static_assert(P % G == 0, "");
#pragma unroll
        for (int p0 = 0; p0 < P; p0 += G) {
            v2f dx[G], dy[G], dz[G], r2[G], ri[G], sc[G];
#define STG(stmt) _Pragma("unroll") for (int g = 0; g < G; ++g) { const int p = p0 + g; (void)p; stmt; }
            STG(dx[g] = qx - xi[p])
            STG(dy[g] = qy - yi[p])
            STG(dz[g] = qz - zi[p])
            STG(r2[g] = pk_fma(dx[g], dx[g], eps2))
            STG(r2[g] = pk_fma(dy[g], dy[g], r2[g]))
            STG(r2[g] = pk_fma(dz[g], dz[g], r2[g]))
            STG(ri[g] = ((v2f){__builtin_amdgcn_rsqf(r2[g].x), __builtin_amdgcn_rsqf(r2[g].y)}))
            STG(sc[g] = gm * ri[g])
            STG(ri[g] = ri[g] * ri[g])
            STG(sc[g] = sc[g] * ri[g])
            STG(A.ax[p] = pk_fma(dx[g], sc[g], A.ax[p]))
            STG(A.ay[p] = pk_fma(dy[g], sc[g], A.ay[p]))
            STG(A.az[p] = pk_fma(dz[g], sc[g], A.az[p]))
#undef STG
        }
        return;
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
        v2f dx = qx - xi[p], dy = qy - yi[p], dz = qz - zi[p];
        v2f r2 = pk_fma(dx, dx, eps2);
        r2 = pk_fma(dy, dy, r2);
        r2 = pk_fma(dz, dz, r2);
        v2f rinv = (v2f){__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
        v2f rinv2 = rinv * rinv;
        v2f sc = gm * rinv;
        sc = sc * rinv2;
        A.ax[p] = pk_fma(dx, sc, A.ax[p]);
        A.ay[p] = pk_fma(dy, sc, A.ay[p]);
        A.az[p] = pk_fma(dz, sc, A.az[p]);
    }
}

// U sources x P pairs, stage by stage
template <int P, int U>
__device__ __forceinline__ void interact_staged(const float4* s, const v2f* xi, const v2f* yi, const v2f* zi, v2f eps2,
                                                Acc<P>& A) {
    v2f dx[U][P], dy[U][P], dz[U][P], r2[U][P], ri[U][P], sc[U][P];
#define ST(stmt) _Pragma("unroll") for (int u = 0; u < U; ++u) _Pragma("unroll") for (int p = 0; p < P; ++p) { stmt; }
    ST(dx[u][p] = splat(s[u].x) - xi[p])
    ST(dy[u][p] = splat(s[u].y) - yi[p])
    ST(dz[u][p] = splat(s[u].z) - zi[p])
    ST(r2[u][p] = pk_fma(dx[u][p], dx[u][p], eps2))
    ST(r2[u][p] = pk_fma(dy[u][p], dy[u][p], r2[u][p]))
    ST(r2[u][p] = pk_fma(dz[u][p], dz[u][p], r2[u][p]))
    ST(ri[u][p] = ((v2f){__builtin_amdgcn_rsqf(r2[u][p].x), __builtin_amdgcn_rsqf(r2[u][p].y)}))
    ST(sc[u][p] = splat(s[u].w) * ri[u][p])
    ST(ri[u][p] = ri[u][p] * ri[u][p])
    ST(sc[u][p] = sc[u][p] * ri[u][p])
    ST(A.ax[p] = pk_fma(dx[u][p], sc[u][p], A.ax[p]))
    ST(A.ay[p] = pk_fma(dy[u][p], sc[u][p], A.ay[p]))
    ST(A.az[p] = pk_fma(dz[u][p], sc[u][p], A.az[p]))
#undef ST
}

template <int P, int VAR, int U, int MINW, int WGS = 256, bool SYNC = false, int STAG = 0, int PF = 0, int ORDER = 0>
__global__ __launch_bounds__(WGS, MINW) void force(const float4* __restrict__ src_all, long n_src_all, long n_tgt,
                                                  float4* __restrict__ acc_all, float eps2s) {
    const long n_src = n_src_all / gridDim.y;
    const float4* __restrict__ src = src_all + (long)blockIdx.y * n_src;
    float4* __restrict__ acc = acc_all + (long)blockIdx.y * n_tgt;
    __shared__ float4 tile[2][TILE];
    constexpr int R = 2 * P;
    const int t = threadIdx.x;
    constexpr int WG = WGS;
    const long base = (long)blockIdx.x * (WG * R);
    v2f xi[P], yi[P], zi[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        float4 b0 = src_all[min(base + (long)(2 * p) * WG + t, n_tgt - 1)];
        float4 b1 = src_all[min(base + (long)(2 * p + 1) * WG + t, n_tgt - 1)];
        xi[p] = (v2f){b0.x, b1.x}; yi[p] = (v2f){b0.y, b1.y}; zi[p] = (v2f){b0.z, b1.z};
    }
    Acc<P> A;
    A.init();
    const v2f eps2 = splat(eps2s);
    const long ntiles = n_src / TILE;  // harness: n_src multiple of TILE

    if constexpr (VAR == V_SMEM_PF) {
        // batches of 4 sources (one s_load_dwordx16 = 64 B) in two SGPR sets A/B, explicit ping-pong: the load of
        // the next batch is issued before the current one is consumed, and waited for (tied "+s" operand, so
        // no use can be scheduled above the wait) only afterwards.  U counts sources per half iteration.
        static_assert(U == 4, "");
        auto body = [&](const v16f& c) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                interact<P>(make_float4(c[4 * u], c[4 * u + 1], c[4 * u + 2], c[4 * u + 3]), xi, yi, zi, eps2, A);
        };
        v16f a = sload16(src), b;
        sload_wait(a);
#pragma unroll 1
        for (long j = 0; j < n_src; j += 8) {
            b = sload16(src + j + 4);
            body(a);
            sload_wait_after(b, A.az[P - 1]);
            a = sload16(src + ((j + 8 < n_src) ? j + 8 : 0));
            body(b);
            sload_wait_after(a, A.az[P - 1]);
            if (((j + 8) & (TILE - 1)) == 0) A.flush();
        }
    } else if constexpr (VAR == V_SMEM_AB) {
        float4 SA[U], SB[U];
#pragma unroll
        for (int u = 0; u < U; ++u) SA[u] = src[u];
        for (long j = 0; j < n_src; j += 2 * U) {  // harness: n_src a multiple of 2U
#pragma unroll
            for (int u = 0; u < U; ++u) SB[u] = src[j + U + u];
#pragma unroll
            for (int u = 0; u < U; ++u) interact<P, ORDER>(SA[u], xi, yi, zi, eps2, A);
            const long jn = (j + 2 * U < n_src) ? j + 2 * U : 0;
#pragma unroll
            for (int u = 0; u < U; ++u) SA[u] = src[jn + u];
#pragma unroll
            for (int u = 0; u < U; ++u) interact<P, ORDER>(SB[u], xi, yi, zi, eps2, A);
            if (((j + 2 * U) & (TILE - 1)) == 0) {
                A.flush();
                if (SYNC) __syncthreads();
            }
        }
    } else if constexpr (VAR == V_SMEM) {
        // sources straight from memory with wave-uniform addresses (scalar loads), one batch prefetched ahead
        float4 cur[U], nxt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = src[u];
        for (long j = 0; j < n_src; j += U) {
            const long jn = (j + U < n_src) ? j + U : 0;
#pragma unroll
            for (int u = 0; u < U; ++u) nxt[u] = src[jn + u];
            if (PF) {  // touch the line PF bytes ahead with a vector load whose result is never used: pulls it into L2
                const long jp = (j + PF / 16 < n_src) ? j + PF / 16 : 0;
                float junk;
                asm volatile("global_load_dword %0, %1, off" : "=v"(junk) : "v"(src + jp) : "memory");
            }
#pragma unroll
            for (int u = 0; u < U; ++u) interact<P, ORDER>(cur[u], xi, yi, zi, eps2, A);
#pragma unroll
            for (int u = 0; u < U; ++u) cur[u] = nxt[u];
            if (((j + U) & (TILE - 1)) == 0) {
                A.flush();
                if (SYNC) __syncthreads();  // keep the workgroup's waves within one tile of each other (L2 locality)
                if (STAG) {  // de-phase the waves that share a SIMD (waves w, w+4, w+8, w+12) by STAG*64 cycles each
                    const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
                    if (g == 1) __builtin_amdgcn_s_sleep(STAG);
                    if (g == 2) __builtin_amdgcn_s_sleep(2 * STAG);
                    if (g == 3) __builtin_amdgcn_s_sleep(3 * STAG);
                }
            }
        }
    } else {
        tile[0][t] = src[t];
        __syncthreads();
        for (long k = 0; k < ntiles; ++k) {
            const int cur = (int)(k & 1);
            float4 nxt;
            if (k + 1 < ntiles) nxt = src[(k + 1) * TILE + t];
            if (VAR == V_LDS) {
#pragma unroll U
                for (int j = 0; j < TILE; ++j) interact<P>(tile[cur][j], xi, yi, zi, eps2, A);
            } else {
#pragma unroll 1
                for (int j = 0; j < TILE; j += U) {
                    float4 s[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) s[u] = tile[cur][j + u];
                    interact_staged<P, U>(s, xi, yi, zi, eps2, A);
                }
            }
            A.flush();
            if (k + 1 < ntiles) tile[cur ^ 1][t] = nxt;
            __syncthreads();
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        long i = base + (long)r * WG + t;
        if (i < n_tgt) acc[i] = make_float4(A.sx[r >> 1][r & 1], A.sy[r >> 1][r & 1], A.sz[r >> 1][r & 1], 0.f);
    }
}

static std::vector<float4> ref_acc;

template <int P, int VAR, int U, int MINW, int WGS = 256, bool SYNC = false, int STAG = 0, int PF = 0, int ORDER = 0>
static void run(const char* name, const float4* d_src, long n_src, long n_tgt, float4* d_acc, int js = 1) {
    const long blocks = (n_tgt + WGS * 2 * P - 1) / (WGS * 2 * P);
    auto launch = [&] { hipLaunchKernelGGL((force<P, VAR, U, MINW, WGS, SYNC, STAG, PF, ORDER>), dim3(blocks, js), dim3(WGS), 0, 0, d_src, n_src, n_tgt, d_acc, 1e-6f); };
    launch();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    const int reps = 5;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms); sum += ms;
    }
    std::vector<float4> h(n_tgt);
    CK(hipMemcpy(h.data(), d_acc, n_tgt * sizeof(float4), hipMemcpyDeviceToHost));
    double maxrel = 0;
    if (ref_acc.empty()) ref_acc = h;
    if (js > 1) h = ref_acc;  // slices are not combined in the harness: timing only
    for (long i = 0; i < n_tgt; ++i) {
        double d = std::fabs(h[i].x - ref_acc[i].x) + std::fabs(h[i].y - ref_acc[i].y) + std::fabs(h[i].z - ref_acc[i].z);
        double s = std::fabs(ref_acc[i].x) + std::fabs(ref_acc[i].y) + std::fabs(ref_acc[i].z) + 1e-30;
        maxrel = std::max(maxrel, d / s);
    }
    double pairs = (double)n_tgt * n_src;
    double cyc = best * 1e-3 * 2.4e9 * 1024.0 / (pairs / 64.0);
    printf("%-34s blocks=%5ld best %8.3f ms avg %8.3f ms  %.3e pairs/s  %5.1f%% of 157.3TF  ~%5.2f cyc/64pairs@2.4GHz  maxdiff_vs_v0 %.2e\n",
           name, blocks, best, sum / reps, pairs / (best * 1e-3), pairs / (best * 1e-3) * 20 / 157.3e12 * 100, cyc, maxrel);
}

int main(int argc, char** argv) {
    const long n_tgt = 1 << 20, n_src = (argc > 1) ? atol(argv[1]) : (1 << 17);
    std::vector<float4> h(n_tgt);
    unsigned long long x = 42;
    auto rnd = [&] { x += 0x9E3779B97F4A7C15ull; unsigned long long z = x; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                     z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31; return (double)(z >> 11) * (1.0 / 9007199254740992.0); };
    for (long i = 0; i < n_tgt; ++i) h[i] = make_float4(2 * rnd() - 1, 2 * rnd() - 1, 2 * rnd() - 1, (0.5 + rnd()) / n_tgt);
    float4 *d_src, *d_acc;
    CK(hipMalloc(&d_src, n_tgt * sizeof(float4))); CK(hipMalloc(&d_acc, 16 * n_tgt * sizeof(float4)));
    CK(hipMemcpy(d_src, h.data(), n_tgt * sizeof(float4), hipMemcpyHostToDevice));
    printf("n_tgt=%ld n_src=%ld\n", n_tgt, n_src);
    // round 3b: product shape (pairs of two, 512-thread workgroups compiled for 4 waves per SIMD = 128 VGPRs), loop-end
    // copies vs A/B SGPR sets, batch of 8 and 16 sources; each twice
    for (int rep = 0; rep < 2; ++rep) {
        run<4, V_SMEM, 8, 4, 512, true, 0, 0, 2>("SMEM    U=8  order2 4w (product)", d_src, n_src, n_tgt, d_acc);
        run<4, V_SMEM_AB, 8, 4, 512, true, 0, 0, 2>("SMEM_AB U=8  order2 4w", d_src, n_src, n_tgt, d_acc);
        run<4, V_SMEM, 16, 4, 512, true, 0, 0, 2>("SMEM    U=16 order2 4w", d_src, n_src, n_tgt, d_acc);
        run<4, V_SMEM_AB, 4, 4, 512, true, 0, 0, 2>("SMEM_AB U=4  order2 4w", d_src, n_src, n_tgt, d_acc);
        run<4, V_SMEM, 8, 4, 512, true, 0, 0, 2>("SMEM    U=8  order2 4w js16", d_src, n_src, n_tgt, d_acc, 16);
        run<4, V_SMEM_AB, 8, 4, 512, true, 0, 0, 2>("SMEM_AB U=8  order2 4w js16", d_src, n_src, n_tgt, d_acc, 16);
    }
    return 0;
}
