/*
 * nbody_amd_debug.h — measurement hooks of the INSTRUMENTED build (libnbody_amd_stamps.so, `make stamps`,
 * compiled with -DNB_ABI_DEBUG=1).  Not part of the product ABI: libnbody_amd.so does not export these symbols, and
 * the product's latency-bound fp64 step kernel carries no instrumentation.  What the reference does with its
 * __debug_printf / nvprof recipes (hw5.cu:22-48,617-669) is done here from the kernel's own clock.
 */
#ifndef NBODY_AMD_DEBUG_H
#define NBODY_AMD_DEBUG_H

#include "nbody_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* With slots > 0 every step launch of this NB_F64 context that does work records the GPU's 100 MHz wall clock at kernel
 * entry and after its last store (workgroup 0) into slot k = (node index within a replayed graph, or the step index for
 * eager launches) mod slots; nb_read_step_stamps copies {entry, exit} pairs out (exit = 0: a monitor-only launch).  Gives
 * the per-launch duration and the launch-to-launch gap of a replayed hipGraph, which rocprofv3 1.1 cannot trace
 * (bench/replay_stamps.py, profiles/r03_replay_stamps.txt).  slots = 0 switches it off again. */
int nb_enable_step_stamps(nb_context* ctx, int slots);
int nb_read_step_stamps(nb_context* ctx, uint64_t* out /* [2*slots] */, int slots);

/* nb_create with the context's stream confined to half of the compute units (measurements of concurrent scenario
 * streams, bench/scenario_concurrency.py) */
typedef enum nb_cu_mask { NB_CU_ALL = 0, NB_CU_LOW = 1, NB_CU_HIGH = 2, NB_CU_EVEN = 3, NB_CU_ODD = 4 } nb_cu_mask;
int nb_create_cu_masked(nb_context** out, const nb_config* cfg, int cu_mask);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_AMD_DEBUG_H */
