// nbody_f32_common.h — what the two fp32 force kernels (nbody_kernels_f32.hip: every ordered pair, sources broadcast;
// nbody_kernels_f32_sym.hip: every unordered pair once, sources travelling through the wave) share: packed-fp32 helpers
// and the epilogue of a step.  Device code only; included by those two files.
#pragma once
#include "nbody_kernels.h"

namespace nbk {

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f splat(float x) { return (v2f){x, x}; }
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// ---- shared epilogue: store accelerations, or kick + drift (samples/nbody.cc:76-88) ----
template <bool ACC64, bool ACCEL_ONLY, typename ACC_T>
__device__ __forceinline__ void finish_target(const F32Args& a, long i, ACC_T ax, ACC_T ay, ACC_T az, float xi,
                                              float yi, float zi, float gmi) {
    if (ACCEL_ONLY) {
        if (ACC64) ((double4*)a.acc)[i] = make_double4((double)ax, (double)ay, (double)az, 0.0);
        else ((float4*)a.acc)[i] = make_float4((float)ax, (float)ay, (float)az, 0.f);
    } else if (ACC64) {
        const double dt = (double)a.dt;
        double4 v = a.vel64[i];
        double4 p = a.pos64[i];
        v.x += (double)ax * dt; v.y += (double)ay * dt; v.z += (double)az * dt;
        p.x += v.x * dt; p.y += v.y * dt; p.z += v.z * dt;
        a.vel64[i] = v;
        a.pos64[i] = p;
        a.out[a.tgt_off + i] = make_float4((float)p.x, (float)p.y, (float)p.z, gmi);
    } else {
        const float dt = a.dt;
        float4 v = a.vel[i];
        v.x = __builtin_fmaf((float)ax, dt, v.x);
        v.y = __builtin_fmaf((float)ay, dt, v.y);
        v.z = __builtin_fmaf((float)az, dt, v.z);
        a.vel[i] = v;
        a.out[a.tgt_off + i] = make_float4(__builtin_fmaf(v.x, dt, xi), __builtin_fmaf(v.y, dt, yi),
                                           __builtin_fmaf(v.z, dt, zi), gmi);
    }
}

}  // namespace nbk
