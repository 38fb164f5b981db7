// valu_rate.hip — gfx950 VALU issue-rate microbenchmark (measurement tool, not product code).
//
// Purpose: settle which fp32 "peak" the all-pairs inner loop is priced against on MI355X:
// cycles per wave64 instruction per SIMD for v_fma_f32 / v_pk_fma_f32 / v_rsq_f32 / fp64 ops,
// as a function of resident waves per SIMD.  Cycles come from s_memtime inside the kernel
// (shader clock), wall time from hipEvents (gives the effective clock under load).
//
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstring>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

enum Op { FMA32, PKFMA32, MUL32, PKMUL32, ADD32, PKADD32, RSQ32, RCP32, SQRT32,
          MIX_PAIR_PK, MIX_PAIR_SC, FMA64, MUL64, ADD64, RSQ64, RCP64, SQRT64, DIVSCALE64, DIVFMAS64,
          DIVFIXUP64, LDS128_FMA, DPP_WAVE_ROR, DPP_ROW_ROR, DPP_BETWEEN_PK, NOPS };
static const char* op_name[] = { "v_fma_f32", "v_pk_fma_f32", "v_mul_f32", "v_pk_mul_f32", "v_add_f32",
  "v_pk_add_f32", "v_rsq_f32", "v_rcp_f32", "v_sqrt_f32", "mix:12pk+2rsq(2 pairs)", "mix:12sc+1rsq(1 pair)",
  "v_fma_f64", "v_mul_f64", "v_add_f64", "v_rsq_f64", "v_rcp_f64", "v_sqrt_f64", "v_div_scale_f64",
  "v_div_fmas_f64", "v_div_fixup_f64", "ds_read_b128+12pk+2rsq", "v_mov_b32_dpp wave_ror:1", "v_mov_b32_dpp row_ror:1",
  "8 v_pk_fma + 1 dpp wave_ror", "" };

constexpr int U = 16;      // independent chains per lane
constexpr int ITERS = 32768;
struct WaveRec { unsigned long long c0, c1, r0, r1; unsigned hw, xcc; };

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, WaveRec* rec, float seed) {
  __shared__ float4 tile[256];
  tile[threadIdx.x] = make_float4(seed + threadIdx.x, seed, seed * 2, seed * 3);
  __syncthreads();
  float a[U]; v2f p[U]; double d[U];
  for (int i = 0; i < U; ++i) { a[i] = seed + i; p[i] = (v2f){seed + i, seed - i}; d[i] = (double)seed + i; }
  float b = seed * 0.5f, c = seed * 0.25f; v2f pb = (v2f){b, c}; v2f pc = (v2f){c, b};
  double db = 1.0 + seed * 1e-3, dc = seed * 1e-6;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int it = 0; it < ITERS; ++it) {
    if constexpr (OP == MIX_PAIR_PK || OP == LDS128_FMA) {
      // the shape of one source body against two targets per lane: 3 sub, 3 fma, 2 rsq, 3 mul, 3 fma.
      // Four source bodies are processed stage by stage so dependent instructions sit >= 4 apart
      // (no compiler-inserted s_nop hazard pads between the asm statements).
      constexpr int GN = U / 4;
      v2f sx[GN], sy[GN], sz[GN], gm[GN], dx[GN], dy[GN], dz[GN], r2[GN], ri[GN], s3[GN];
#pragma unroll
      for (int g = 0; g < GN; ++g) {
        sx[g] = pb; sy[g] = pc; sz[g] = pb; gm[g] = pc;
        if constexpr (OP == LDS128_FMA) {
          float4 s = tile[(it * GN + g) & 255];
          sx[g] = (v2f){s.x, s.x}; sy[g] = (v2f){s.y, s.y}; sz[g] = (v2f){s.z, s.z}; gm[g] = (v2f){s.w, s.w};
        }
      }
#define STAGE(stmt) _Pragma("unroll") for (int g = 0; g < GN; ++g) { stmt; }
      STAGE(asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dx[g]) : "v"(sx[g]), "v"(p[4 * g])))
      STAGE(asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dy[g]) : "v"(sy[g]), "v"(p[4 * g + 1])))
      STAGE(asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dz[g]) : "v"(sz[g]), "v"(p[4 * g + 2])))
      STAGE(asm volatile("v_pk_fma_f32 %0, %1, %1, %2" : "=v"(r2[g]) : "v"(dx[g]), "v"(pc)))
      STAGE(asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(r2[g]) : "v"(dy[g])))
      STAGE(asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(r2[g]) : "v"(dz[g])))
      STAGE(asm volatile("v_rsq_f32 %0, %1" : "=v"(ri[g].x) : "v"(r2[g].x)))
      STAGE(asm volatile("v_rsq_f32 %0, %1" : "=v"(ri[g].y) : "v"(r2[g].y)))
      STAGE(asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(s3[g]) : "v"(ri[g])))
      STAGE(asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(ri[g]) : "v"(ri[g]), "v"(gm[g])))
      STAGE(asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(s3[g]) : "v"(s3[g]), "v"(ri[g])))
      STAGE(asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[4 * g + 3]) : "v"(dx[g]), "v"(s3[g])))
      STAGE(asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[(4 * g + 7) % U]) : "v"(dy[g]), "v"(s3[g])))
      STAGE(asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[(4 * g + 11) % U]) : "v"(dz[g]), "v"(s3[g])))
    } else if constexpr (OP == MIX_PAIR_SC) {
      constexpr int GN = U / 4;
      float dx[GN], dy[GN], dz[GN], r2[GN], ri[GN], s3[GN];
      STAGE(asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dx[g]) : "v"(b), "v"(a[4 * g])))
      STAGE(asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dy[g]) : "v"(c), "v"(a[4 * g + 1])))
      STAGE(asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dz[g]) : "v"(b), "v"(a[4 * g + 2])))
      STAGE(asm volatile("v_fma_f32 %0, %1, %1, %2" : "=v"(r2[g]) : "v"(dx[g]), "v"(c)))
      STAGE(asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(r2[g]) : "v"(dy[g])))
      STAGE(asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(r2[g]) : "v"(dz[g])))
      STAGE(asm volatile("v_rsq_f32 %0, %1" : "=v"(ri[g]) : "v"(r2[g])))
      STAGE(asm volatile("v_mul_f32 %0, %1, %1" : "=v"(s3[g]) : "v"(ri[g])))
      STAGE(asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ri[g]) : "v"(ri[g]), "v"(c)))
      STAGE(asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s3[g]) : "v"(s3[g]), "v"(ri[g])))
      STAGE(asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[4 * g + 3]) : "v"(dx[g]), "v"(s3[g])))
      STAGE(asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[(4 * g + 7) % U]) : "v"(dy[g]), "v"(s3[g])))
      STAGE(asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[(4 * g + 11) % U]) : "v"(dz[g]), "v"(s3[g])))
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if constexpr (OP == FMA32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[u]) : "v"(b), "v"(c));
        if constexpr (OP == PKFMA32) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[u]) : "v"(pb), "v"(pc));
        if constexpr (OP == MUL32) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[u]) : "v"(b));
        if constexpr (OP == PKMUL32) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[u]) : "v"(pb));
        if constexpr (OP == ADD32) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[u]) : "v"(b));
        if constexpr (OP == PKADD32) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[u]) : "v"(pb));
        if constexpr (OP == DPP_WAVE_ROR) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[u]));
        if constexpr (OP == DPP_ROW_ROR) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[u]));
        if constexpr (OP == DPP_BETWEEN_PK) {  // the product's ratio: 8 packed ops per DPP move, different registers
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[u]) : "v"(pb), "v"(pc));
          if ((u & 7) == 7) asm volatile("v_mov_b32_dpp %0, %0 wave_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[u]));
        }
        if constexpr (OP == RSQ32) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[u]));
        if constexpr (OP == RCP32) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[u]));
        if constexpr (OP == SQRT32) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[u]));
        if constexpr (OP == FMA64) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[u]) : "v"(db), "v"(dc));
        if constexpr (OP == MUL64) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(d[u]) : "v"(db));
        if constexpr (OP == ADD64) asm volatile("v_add_f64 %0, %1, %0" : "+v"(d[u]) : "v"(dc));
        if constexpr (OP == RSQ64) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[u]));
        if constexpr (OP == RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[u]));
        if constexpr (OP == SQRT64) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d[u]));
        if constexpr (OP == DIVSCALE64) asm volatile("v_div_scale_f64 %0, vcc, %1, %1, %0" : "+v"(d[u]) : "v"(db) : "vcc");
        if constexpr (OP == DIVFMAS64) asm volatile("v_div_fmas_f64 %0, %1, %2, %0" : "+v"(d[u]) : "v"(db), "v"(dc) : "vcc");
        if constexpr (OP == DIVFIXUP64) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(d[u]) : "v"(db), "v"(dc));
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  float s = 0; double sd = 0;
  for (int i = 0; i < U; ++i) { s += a[i] + p[i].x + p[i].y; sd += d[i]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)sd;
  if ((threadIdx.x & 63) == 0) {
    WaveRec w; w.c0 = t0; w.c1 = t1; w.r0 = r0; w.r1 = r1;
    w.hw = __builtin_amdgcn_s_getreg(63492); w.xcc = __builtin_amdgcn_s_getreg(63508);
    rec[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = w;
  }
}

template <int OP>
static void run(int ncu, int instr_per_iter_per_lane_chain, float* d_out, WaveRec* d_rec) {
  const int wps_list[] = {1, 2, 3, 4, 6, 8};
  for (int wps : wps_list) {
    // one 256-thread block = 4 waves; wps blocks per CU.  Placement is whatever the dispatcher does: it is
    // recorded (HW_ID) and reported as min/max waves per SIMD.
    int blocks = ncu * wps;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<OP><<<blocks, 256>>>(d_out, d_rec, 1.25f);   // warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k<OP><<<blocks, 256>>>(d_out, d_rec, 1.25f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<WaveRec> rec(blocks * 4);
    CK(hipMemcpy(rec.data(), d_rec, rec.size() * sizeof(WaveRec), hipMemcpyDeviceToHost));
    std::map<unsigned long long, int> per_simd;
    unsigned long long rmin = ~0ull, rmax = 0; double clk_sum = 0; std::vector<double> cyc;
    for (auto& w : rec) {
      unsigned long long key = ((unsigned long long)(w.xcc & 0xf) << 32) | (w.hw & 0xff30u);  // se,sh,cu,simd
      per_simd[key]++;
      rmin = std::min(rmin, w.r0); rmax = std::max(rmax, w.r1);
      clk_sum += (double)(w.c1 - w.c0) / (double)(w.r1 - w.r0) * 0.1;  // GHz (realtime ticks are 100 MHz)
      cyc.push_back((double)(w.c1 - w.c0));
    }
    std::sort(cyc.begin(), cyc.end());
    int mn = 1 << 30, mx = 0; for (auto& kv : per_simd) { mn = std::min(mn, kv.second); mx = std::max(mx, kv.second); }
    double ghz = clk_sum / rec.size();
    double span_us = (double)(rmax - rmin) * 0.01;
    double n_inst = (double)ITERS * instr_per_iter_per_lane_chain;       // wave-instructions per wave
    double simd_inst = n_inst * rec.size() / (ncu * 4.0);                // per SIMD if spread evenly
    double cyc_per_inst = span_us * 1e3 * ghz / simd_inst;               // SIMD-cycles per wave-instruction
    printf("%-24s blk/CU=%d simds_used=%4zu waves/SIMD[min,max]=[%d,%d] span=%8.1f us wall=%8.1f us clk=%.3f GHz "
           "wave-cyc/inst(med)=%6.2f  SIMD cyc/inst=%6.3f  lane-ops/clk/CU=%6.1f\n",
           op_name[OP], wps, per_simd.size(), mn, mx, span_us, ms * 1e3, ghz, cyc[cyc.size() / 2] / n_inst,
           cyc_per_inst, 4.0 * 64.0 / cyc_per_inst);
  }
}

int main(int argc, char** argv) {
  const bool only_dpp = argc > 1 && !strcmp(argv[1], "dpp");
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int ncu = prop.multiProcessorCount;
  printf("device %s  CUs=%d  clockRate=%d kHz  arch=%s\n", prop.name, ncu, prop.clockRate, prop.gcnArchName);
  float* d_out; WaveRec* d_cyc;
  CK(hipMalloc(&d_out, (size_t)ncu * 8 * 256 * 4)); CK(hipMalloc(&d_cyc, (size_t)ncu * 8 * 4 * sizeof(WaveRec)));
  if (only_dpp) {  // what a cross-lane move costs (K1s rotates its travelling sources with 14 of them per step)
    run<PKFMA32>(ncu, U, d_out, d_cyc);
    run<DPP_WAVE_ROR>(ncu, U, d_out, d_cyc);
    run<DPP_ROW_ROR>(ncu, U, d_out, d_cyc);
    run<DPP_BETWEEN_PK>(ncu, U + U / 8, d_out, d_cyc);
    return 0;
  }
  run<FMA32>(ncu, U, d_out, d_cyc);
  run<PKFMA32>(ncu, U, d_out, d_cyc);
  run<MUL32>(ncu, U, d_out, d_cyc);
  run<PKMUL32>(ncu, U, d_out, d_cyc);
  run<ADD32>(ncu, U, d_out, d_cyc);
  run<PKADD32>(ncu, U, d_out, d_cyc);
  run<RSQ32>(ncu, U, d_out, d_cyc);
  run<RCP32>(ncu, U, d_out, d_cyc);
  run<SQRT32>(ncu, U, d_out, d_cyc);
  run<MIX_PAIR_PK>(ncu, 14 * (U / 4), d_out, d_cyc);
  run<MIX_PAIR_SC>(ncu, 13 * (U / 4), d_out, d_cyc);
  run<LDS128_FMA>(ncu, 14 * (U / 4), d_out, d_cyc);
  run<FMA64>(ncu, U, d_out, d_cyc);
  run<MUL64>(ncu, U, d_out, d_cyc);
  run<ADD64>(ncu, U, d_out, d_cyc);
  run<RSQ64>(ncu, U, d_out, d_cyc);
  run<RCP64>(ncu, U, d_out, d_cyc);
  run<SQRT64>(ncu, U, d_out, d_cyc);
  run<DIVSCALE64>(ncu, U, d_out, d_cyc);
  run<DIVFMAS64>(ncu, U, d_out, d_cyc);
  run<DIVFIXUP64>(ncu, U, d_out, d_cyc);
  return 0;
}
