#!/bin/bash
# bench/sums64_ab.sh — same-box, alternating A/B of K1s' second summation level in NB_F32: Kahan-compensated fp32 pairs (rounds
# 1-4; library built with `make LIB=bench/ab/kahan/libnbody_amd.so EXTRA=-DNB_SYM_F32_KAHAN=1 lib`) against fp64 registers
# (round 5, the in-tree library).  bin/nbody_bench finds its library through RUNPATH, which LD_LIBRARY_PATH precedes.
# env: ROUNDS (3), BODIES (1048576), STEPS (10)
cd "$(dirname "$0")/.." || exit 1
ROUNDS=${ROUNDS:-3}; BODIES=${BODIES:-1048576}; STEPS=${STEPS:-10}
for r in $(seq "$ROUNDS"); do
  echo "== kahan f32";  LD_LIBRARY_PATH=bench/ab/kahan bin/nbody_bench "$BODIES" "$STEPS" 2 f32 || exit 1
  echo "== sums64 f32"; bin/nbody_bench "$BODIES" "$STEPS" 2 f32 || exit 1
  echo "== sums64 f32acc64"; bin/nbody_bench "$BODIES" "$STEPS" 2 f32acc64 || exit 1
done
